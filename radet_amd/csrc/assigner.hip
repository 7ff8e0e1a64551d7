// Visibility-guided positive-sample assigner on the GPU, one workgroup per image.
// Replaces LabelAssignment.__call__ / generate_candidate_cell / cal_sample_pro / random_sample
// (radet/datasets/pipelines/label_assignment.py:57-201), which runs in NumPy inside DataLoader workers.
//
// Bit-exact contract: the gts are visited in ascending box area; per gt the candidate test and the mask
// gather run in parallel (ordered wave-ballot compaction keeps NumPy's index order); the float32
// normaliser reproduces NumPy's pairwise summation; the draw is numpy's legacy
// RandomState.choice(p=..., replace = n < positive_num) -- float64 sequential cumsum, normalised cdf,
// searchsorted(side='right'), first-occurrence de-duplication -- or, with random_sample_by_distance=False, the integer draws
// of choice() WITHOUT p (randint / permutation: masked rejection sampling) -- driven by the host-supplied stream of the
// RandomState's raw 32-bit outputs (a uniform is two words: genrand_res53), so results equal the reference's for the same seed.
#include "common.h"
#include "../../include/radet_hip.h"

#define ASG_MAXG 256
#define ASG_MAXK 64

struct AsgLevels {
    int n;
    int h[RADET_MAX_SEG], w[RADET_MAX_SEG], stride[RADET_MAX_SEG];
    int pt_off[RADET_MAX_SEG + 1];
    float lo[RADET_MAX_SEG], hi[RADET_MAX_SEG];
};

// exclusive prefix over a 256-thread block (ordered), returns total through `total`
__device__ __forceinline__ int block_excl_scan_256(int flag, int* s_wave, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int v = s_wave[i];
        if (i < wave) base += v;
        tot += v;
    }
    total = tot;
    return base + within;
}

// NumPy's pairwise float32 summation (numpy/core/src/umath/loops_utils.h.src, pairwise_sum), iteratively
__device__ float np_pairwise_sum(const float* a, int n) {
    // explicit stack of (offset, length); results combined in post-order
    int st_off[32], st_len[32], st_state[32];
    float st_val[32];
    int sp = 0;
    st_off[0] = 0; st_len[0] = n; st_state[0] = 0; st_val[0] = 0.f;
    float ret = 0.f;
    while (sp >= 0) {
        const int off = st_off[sp], len = st_len[sp];
        if (st_state[sp] == 0) {
            if (len < 8) {
                float res = 0.f;
                for (int i = 0; i < len; ++i) res += a[off + i];
                ret = res;
                --sp;
            } else if (len <= 128) {
                float r[8];
                for (int j = 0; j < 8; ++j) r[j] = a[off + j];
                int i;
                for (i = 8; i < len - (len % 8); i += 8)
                    for (int j = 0; j < 8; ++j) r[j] += a[off + i + j];
                float res = ((r[0] + r[1]) + (r[2] + r[3])) + ((r[4] + r[5]) + (r[6] + r[7]));
                for (; i < len; ++i) res += a[off + i];
                ret = res;
                --sp;
            } else {
                int n2 = len / 2;
                n2 -= n2 % 8;
                st_state[sp] = 1;
                ++sp;
                st_off[sp] = off; st_len[sp] = n2; st_state[sp] = 0;
            }
        } else if (st_state[sp] == 1) {
            int n2 = len / 2;
            n2 -= n2 % 8;
            st_val[sp] = ret;
            st_state[sp] = 2;
            ++sp;
            st_off[sp] = off + n2; st_len[sp] = len - n2; st_state[sp] = 0;
        } else {
            ret = st_val[sp] + ret;
            --sp;
        }
    }
    return ret;
}

__device__ __forceinline__ int upper_bound_d(const double* cdf, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (cdf[mid] <= x) lo = mid + 1; else hi = mid;
    }
    return lo;
}

// one double of RandomState.random_sample() from two consecutive MT19937 outputs (genrand_res53)
__device__ __forceinline__ double uniform_from_words(const uint32_t* w) {
    return ((double)(w[0] >> 5) * 67108864.0 + (double)(w[1] >> 6)) / 9007199254740992.0;
}
// legacy bounded integer in [0, rng] (numpy random_interval / the masked path of randint): words & mask until <= rng.
// Returns the new stream position, or -1 when the stream is exhausted.
__device__ __forceinline__ int bounded_from_words(const uint32_t* w, int pos, int U, unsigned rng, unsigned* out) {
    if (rng == 0) { *out = 0; return pos; }
    unsigned mask = rng;
    mask |= mask >> 1; mask |= mask >> 2; mask |= mask >> 4; mask |= mask >> 8; mask |= mask >> 16;
    while (true) {
        if (pos >= U) return -1;
        const unsigned v = w[pos++] & mask;
        if (v <= rng) { *out = v; return pos; }
    }
}

// MT = uint8_t: visible instance masks (GenerateDistanceMap(with_gt_mask=True)); float: the per-box distance maps of
// the mask-free sampler (MBD / GDT transforms, loading.py:586-645), read as np.float32 like label_assignment.py:85-92
template <class MT>
__global__ __launch_bounds__(256) void assign_kernel(const float* __restrict__ gt_boxes, const int* __restrict__ gt_off,
                                                     const MT* __restrict__ masks, int H, int W,
                                                     const uint32_t* __restrict__ rng_words, int U, const AsgLevels L,
                                                     int K0, int flags, float neg_thr, int64_t* __restrict__ p2g_all,
                                                     float* __restrict__ pw_all, int* __restrict__ used_out,
                                                     char* __restrict__ ws_all, size_t ws_per_image) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = L.pt_off[L.n];
    const int g0 = gt_off[b], G = gt_off[b + 1] - g0;
    int64_t* p2g = p2g_all + (size_t)b * N;
    float* pw = pw_all + (size_t)b * N;
    const uint32_t* u = rng_words + (size_t)b * U;      // this image's stream of raw 32-bit outputs; s_i[0] = words consumed
    char* w = ws_all + (size_t)b * ws_per_image;
    double* p64 = (double*)w; w += (size_t)N * 8;
    double* cdf = (double*)w; w += (size_t)N * 8;
    int* cand_idx = (int*)w; w += (size_t)N * 4;
    float* cand_p = (float*)w; w += (size_t)N * 4;
    int* nn_idx = (int*)w; w += (size_t)N * 4;
    float* nn_p = (float*)w;

    __shared__ int order[ASG_MAXG];
    __shared__ float area[ASG_MAXG];
    __shared__ int s_wave[4];
    __shared__ float s_red[4];
    __shared__ int s_i[8];
    __shared__ float s_f[4];
    __shared__ int chosen[ASG_MAXK], found[ASG_MAXK], uniq_idx[ASG_MAXK], uniq_cnt[ASG_MAXK];
    __shared__ int s_lvl[RADET_MAX_SEG];        // candidates per level (adapt_positive_num)
    // flags (label_assignment.py:30-46, 88-131): bit 0 balance_sample (fewer non-negative candidates than positive_num: draw
    // positive_num with replacement; off: take them all once), bit 1 multiply_samplepro_for_weight (weight = count x the
    // candidate's clipped map value), bit 2 adapt_positive_num (positive_num per gt from the candidates' anchor sizes)
    // bit 3: random_sample_by_distance = False: np.random.choice without p -- randint(0, n, K) when drawing with replacement,
    // permutation(n)[:K] (Fisher-Yates from the top, one bounded integer per position) otherwise
    const bool balance = (flags & 1) != 0, mulpro = (flags & 2) != 0, adapt = (flags & 4) != 0, uniform = (flags & 8) != 0;
    __shared__ double xs[ASG_MAXK];

    for (int p = tid; p < N; p += 256) { p2g[p] = -1; pw[p] = 1.f; }
    if (tid == 0) {
        for (int g = 0; g < G && g < ASG_MAXG; ++g) {
            const float* bb = gt_boxes + (size_t)(g0 + g) * 4;
            area[g] = (bb[2] - bb[0]) * (bb[3] - bb[1]);
            order[g] = g;
        }
        for (int i = 1; i < G && i < ASG_MAXG; ++i) {   // stable insertion sort, ascending area
            const int oi = order[i];
            const float ai = area[oi];
            int j = i - 1;
            while (j >= 0 && area[order[j]] > ai) { order[j + 1] = order[j]; --j; }
            order[j + 1] = oi;
        }
        s_i[0] = 0;  // 32-bit words of the stream consumed
    }
    __syncthreads();
    if (G > ASG_MAXG) { if (tid == 0) used_out[b] = -2; return; }

    for (int oi = 0; oi < G; ++oi) {
        const int g = order[oi];
        const float* bb = gt_boxes + (size_t)(g0 + g) * 4;
        const float x1 = bb[0], y1 = bb[1], x2 = bb[2], y2 = bb[3];
        const MT* mk = masks + (size_t)(g0 + g) * H * W;
        // ---- candidates, ordered
        int nc = 0;
        float pmax = 0.f;
        if (adapt) {
            if (tid < RADET_MAX_SEG) s_lvl[tid] = 0;
            __syncthreads();
        }
        for (int c0 = 0; c0 < N; c0 += 256) {
            const int p = c0 + tid;
            int flag = 0;
            float prob = 0.f;
            if (p < N) {
                int l = 0;
#pragma unroll
                for (int i = 1; i < RADET_MAX_SEG; ++i)
                    if (i < L.n && p >= L.pt_off[i]) l = i;
                const int pix = p - L.pt_off[l];
                const int iy = pix / L.w[l], ix = pix - iy * L.w[l];
                const float cx = (float)(ix * L.stride[l]), cy = (float)(iy * L.stride[l]);
                const float left = cx - x1, right = x2 - cx, top = cy - y1, bottom = y2 - cy;
                const float mn = fminf(fminf(left, top), fminf(right, bottom));
                const float mx = fmaxf(fmaxf(left, top), fmaxf(right, bottom));
                if (mn > 0.01f && mx >= L.lo[l] && mx <= L.hi[l] && p2g[p] == -1) {
                    flag = 1;
                    if (adapt) atomicAdd(&s_lvl[l], 1);          // (integer counts: order independent)
                    prob = fmaxf((float)mk[(size_t)(iy * L.stride[l]) * W + ix * L.stride[l]], 1e-8f);
                }
            }
            int t;
            const int pos = block_excl_scan_256(flag, s_wave, t);
            if (flag) { cand_idx[nc + pos] = p; cand_p[nc + pos] = prob; pmax = fmaxf(pmax, prob); }
            nc += t;
        }
        if (nc == 0) continue;   // uniform across the block
        pmax = wave_max(pmax);
        __syncthreads();
        if ((tid & 63) == 0) s_red[tid >> 6] = pmax;
        __syncthreads();
        pmax = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        const float thr = neg_thr * pmax;
        // positive_num of this gt.  adapt_cal_k (label_assignment.py:88-95): over the anchor sizes present among the candidates
        // (ascending = level order), ratio (float64) x exp((object_size - size) / (2 size)) (float32), summed in order,
        // x positive_num, rounded half up.  expf: numpy's float32 exp may differ in the last bit -- it only matters within an
        // ulp of a .5 boundary of the product
        int K = K0;
        if (adapt) {
            if (tid == 0) {
                const float obj = fmaxf(x2 - x1, y2 - y1);
                double dk = 0.0;
                for (int l = 0; l < L.n; ++l)
                    if (s_lvl[l] > 0) {
                        const float sz = (float)(8 * L.stride[l]);
                        dk += ((double)s_lvl[l] / (double)nc) * (double)expf((obj - sz) / (2.f * sz));
                    }
                s_i[4] = (int)((double)K0 * dk + 0.5);
            }
            __syncthreads();
            K = s_i[4];
            if (K > ASG_MAXK) { if (tid == 0) used_out[b] = -3; return; }
        }
        // ---- non-negative subset (p > thr), ordered
        int n = 0;
        for (int c0 = 0; c0 < nc; c0 += 256) {
            const int i = c0 + tid;
            const int flag = (i < nc) && (cand_p[i] > thr);
            int t;
            const int pos = block_excl_scan_256(flag, s_wave, t);
            if (flag) { nn_idx[n + pos] = cand_idx[i]; nn_p[n + pos] = cand_p[i]; }
            n += t;
        }
        __syncthreads();
        if (tid == 0) s_f[0] = np_pairwise_sum(nn_p, n);
        __syncthreads();
        const float ssum = s_f[0];
        for (int i = tid; i < n; i += 256) p64[i] = (double)(nn_p[i] / ssum);
        __syncthreads();
        // ---- fewer non-negative candidates than positive_num and no balancing: all of them, once each (no draw)
        if (n < K && !balance) {
            for (int i = tid; i < n; i += 256) { p2g[nn_idx[i]] = g + 1; pw[nn_idx[i]] = mulpro ? nn_p[i] : 1.f; }
            __syncthreads();
            continue;
        }
        // ---- numpy legacy choice
        const bool replace = n < K;
        if (uniform) {
            // choice(a = n, size = K, replace) without p (label_assignment.py:111, 118): sequential by construction
            if (tid == 0) {
                int pos = s_i[0];
                if (replace) {
                    for (int k = 0; k < K && pos >= 0; ++k) {
                        unsigned v;
                        pos = bounded_from_words(u, pos, U, (unsigned)(n - 1), &v);
                        chosen[k] = (int)v;
                    }
                } else {
                    int* perm = (int*)cdf;                       // scratch of N doubles: the permutation being shuffled
                    for (int i = 0; i < n; ++i) perm[i] = i;
                    for (int i = n - 1; i >= 1 && pos >= 0; --i) {
                        unsigned v;
                        pos = bounded_from_words(u, pos, U, (unsigned)i, &v);
                        if (pos >= 0) { const int t = perm[i]; perm[i] = perm[v]; perm[v] = t; }
                    }
                    if (pos >= 0)
                        for (int k = 0; k < K; ++k) chosen[k] = perm[k];
                }
                s_i[1] = pos < 0 ? 1 : 0;
                if (pos >= 0) s_i[0] = pos;
            }
            __syncthreads();
            if (s_i[1]) { if (tid == 0) used_out[b] = -1; return; }
        } else if (replace) {
            if (tid == 0) {
                double acc = 0.0;
                for (int i = 0; i < n; ++i) { acc += p64[i]; cdf[i] = acc; }
                int us = s_i[0];
                if (us + 2 * K > U) s_i[1] = 1;
                else { s_i[1] = 0; for (int k = 0; k < K; ++k) xs[k] = uniform_from_words(u + us + 2 * k); s_i[0] = us + 2 * K; }
            }
            __syncthreads();
            if (s_i[1]) { if (tid == 0) used_out[b] = -1; return; }
            const double last = cdf[n - 1];
            __syncthreads();
            for (int i = tid; i < n; i += 256) cdf[i] = cdf[i] / last;
            __syncthreads();
            if (tid < K) chosen[tid] = upper_bound_d(cdf, n, xs[tid]);
            __syncthreads();
        } else {
            if (tid == 0) s_i[2] = 0;   // n_uniq
            __syncthreads();
            while (true) {
                const int n_uniq = s_i[2];
                if (n_uniq >= K) break;
                const int kk = K - n_uniq;
                if (tid == 0) {
                    const int us = s_i[0];
                    if (us + 2 * kk > U) s_i[1] = 1;
                    else {
                        s_i[1] = 0;
                        for (int k = 0; k < kk; ++k) xs[k] = uniform_from_words(u + us + 2 * k);
                        s_i[0] = us + 2 * kk;
                        for (int k = 0; k < n_uniq; ++k) p64[found[k]] = 0.0;
                        double acc = 0.0;
                        for (int i = 0; i < n; ++i) { acc += p64[i]; cdf[i] = acc; }
                    }
                }
                __syncthreads();
                if (s_i[1]) { if (tid == 0) used_out[b] = -1; return; }
                const double last = cdf[n - 1];
                __syncthreads();
                for (int i = tid; i < n; i += 256) cdf[i] = cdf[i] / last;
                __syncthreads();
                if (tid < kk) chosen[tid] = upper_bound_d(cdf, n, xs[tid]);
                __syncthreads();
                if (tid == 0) {
                    int nu = n_uniq;
                    for (int k = 0; k < kk; ++k) {
                        const int v = chosen[k];
                        bool dup = false;
                        for (int q = 0; q < k; ++q) dup |= (chosen[q] == v);
                        if (!dup) found[nu++] = v;
                    }
                    s_i[2] = nu;
                }
                __syncthreads();
            }
            if (tid < K) chosen[tid] = found[tid];
            __syncthreads();
        }
        // ---- multiplicities of the drawn indices
        if (tid == 0) {
            int nu = 0;
            for (int k = 0; k < K; ++k) {
                const int v = chosen[k];
                int q = 0;
                for (; q < nu; ++q) if (uniq_idx[q] == v) break;
                if (q == nu) { uniq_idx[nu] = v; uniq_cnt[nu] = 1; ++nu; } else ++uniq_cnt[q];
            }
            s_i[3] = nu;
        }
        __syncthreads();
        for (int i = tid; i < n; i += 256) { p2g[nn_idx[i]] = 0; pw[nn_idx[i]] = 0.f; }
        __syncthreads();
        if (tid < s_i[3]) {
            p2g[nn_idx[uniq_idx[tid]]] = g + 1;
            pw[nn_idx[uniq_idx[tid]]] = mulpro ? (float)uniq_cnt[tid] * nn_p[uniq_idx[tid]] : (float)uniq_cnt[tid];
        }
        __syncthreads();
    }
    if (tid == 0) used_out[b] = s_i[0];
}

extern "C" size_t radet_assign_ws_bytes(int B, int N) { return (size_t)B * (((size_t)N * 32 + 255) / 256 * 256); }

template <class MT>
static int assign_impl(const float* gt_boxes, const int* gt_off, const MT* masks, int H, int W, const uint32_t* rng_words, int U,
                       const int* level_desc, const float* regress_ranges, int nlvl, int B, int positive_num, int flags,
                       float neg_threshold, int64_t* p2g, float* pw, int* used, void* ws, void* stream) {
    if (nlvl < 1 || nlvl > RADET_MAX_SEG || positive_num < 1 || positive_num > ASG_MAXK || B < 1 || (flags & ~15)) return RADET_ERR_ARG;
    AsgLevels L;
    L.n = nlvl;
    int pt = 0;
    for (int l = 0; l < nlvl; ++l) {
        L.h[l] = level_desc[3 * l]; L.w[l] = level_desc[3 * l + 1]; L.stride[l] = level_desc[3 * l + 2];
        L.lo[l] = regress_ranges[2 * l]; L.hi[l] = regress_ranges[2 * l + 1];
        L.pt_off[l] = pt;
        pt += L.h[l] * L.w[l];
    }
    L.pt_off[nlvl] = pt;
    for (int l = nlvl; l < RADET_MAX_SEG; ++l) { L.h[l] = 1; L.w[l] = 1; L.stride[l] = 1; L.lo[l] = 0.f; L.hi[l] = 0.f; }
    const size_t per = ((size_t)pt * 32 + 255) / 256 * 256;
    hipLaunchKernelGGL(assign_kernel<MT>, dim3(B), dim3(256), 0, (hipStream_t)stream, gt_boxes, gt_off, masks, H, W, rng_words,
                       U, L, positive_num, flags, neg_threshold, p2g, pw, used, (char*)ws, per);
    return radet_check_launch();
}

extern "C" int radet_assign_points(const float* gt_boxes, const int* gt_off, const uint8_t* masks, int H, int W,
                                   const uint32_t* rng_words, int U, const int* level_desc, const float* regress_ranges,
                                   int nlvl, int B, int positive_num, int flags, float neg_threshold, int64_t* p2g, float* pw,
                                   int* used, void* ws, void* stream) {
    return assign_impl<uint8_t>(gt_boxes, gt_off, masks, H, W, rng_words, U, level_desc, regress_ranges, nlvl, B, positive_num,
                                flags, neg_threshold, p2g, pw, used, ws, stream);
}

extern "C" int radet_assign_points_f(const float* gt_boxes, const int* gt_off, const float* distance_maps, int H, int W,
                                     const uint32_t* rng_words, int U, const int* level_desc, const float* regress_ranges,
                                     int nlvl, int B, int positive_num, int flags, float neg_threshold, int64_t* p2g, float* pw,
                                     int* used, void* ws, void* stream) {
    return assign_impl<float>(gt_boxes, gt_off, distance_maps, H, W, rng_words, U, level_desc, regress_ranges, nlvl, B,
                              positive_num, flags, neg_threshold, p2g, pw, used, ws, stream);
}
