// Inference post-processing on the GPU (the reference copies candidates to the host and runs a
// single-threaded C++ loop: radet_head.py:149-153, ops/vote/vote_ext.cpp).
//
//  radet_decode_candidates : per (level, image) workgroup: sigmoid > thr, exact top-k by 3-pass radix
//                            select on the score bits, ordered wave-ballot compaction, TBLR decode + clamp.
//  radet_nms               : per image workgroup (1024 threads): LDS bitonic sort on 64-bit keys
//                            (label | score desc | index), per-label greedy clustering by one wavefront
//                            each (64 IoUs per step, in the reference's fp32 operation order), second
//                            sort of the cluster heads, then the score-weighted 1-sigma box vote with
//                            strictly sequential fp32 sums (bit-exact w.r.t. vote_ext.cpp).
//  Modes: 0 vote_nms, 1 global_vote_nms (vote_ext.cpp:70-353), 2 cluster_nms (cluster_ext.cpp:4-87),
//         3 class-aware hard NMS with mmcv.ops.batched_nms semantics.
#include "common.h"
#include <stdlib.h>
#include "../../include/radet_hip.h"

struct DecLevels {
    int n;
    int h[RADET_MAX_SEG], w[RADET_MAX_SEG], stride[RADET_MAX_SEG];
    int row_off[RADET_MAX_SEG + 1];
};

__device__ __forceinline__ float sigmoidf_(float x) { return 1.f / (1.f + expf(-x)); }

// ordered compaction helper: returns the exclusive prefix of `flag` over the 1024-thread block and the total
__device__ __forceinline__ int block_excl_scan_1024(int flag, int* s_wave, int& total) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const unsigned long long b = __ballot(flag);
    const int within = __popcll(b & ((1ull << lane) - 1ull));
    __syncthreads();
    if (lane == 0) s_wave[wave] = __popcll(b);
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        const int v = s_wave[i];
        if (i < wave) base += v;
        tot += v;
    }
    total = tot;
    return base + within;
}

#define DEC_LIST_CAP 16384   // candidates per (level, image) the LDS list of the fast decode path holds (128 KiB)

struct DecCtx {
    const float* reg_u; const float* iou; const float* scale_factor;
    float* lv_boxes; float* lv_scores; float* lv_ctr; int64_t* lv_labels;
    int C, wl, stride_i, r0, b;
    float stride, sc, H, W;
    size_t obase;
};

// TBLR decode + clamp (+ rescale) of flattened candidate i with score s into output slot `slot` of its level region
__device__ __forceinline__ void decode_one(const DecCtx& d, int i, float s, int slot) {
    const int pt = i / d.C, c = i - pt * d.C;
    const int iy = pt / d.wl, ix = pt - iy * d.wl;
    const float cx = (float)(ix * d.stride_i), cy = (float)(iy * d.stride_i);
    const float4 u4 = *reinterpret_cast<const float4*>(d.reg_u + (size_t)(d.r0 + pt) * 4);
    const float hw8 = 8.f * d.stride;
    const float top = fmaxf(u4.x * d.sc, 0.f) * 0.125f * hw8, bottom = fmaxf(u4.y * d.sc, 0.f) * 0.125f * hw8;
    const float left = fmaxf(u4.z * d.sc, 0.f) * 0.125f * hw8, right = fmaxf(u4.w * d.sc, 0.f) * 0.125f * hw8;
    float x1 = cx - left, y1 = cy - top, x2 = cx + right, y2 = cy + bottom;
    x1 = fminf(fmaxf(x1, 0.f), d.W); x2 = fminf(fmaxf(x2, 0.f), d.W);
    y1 = fminf(fmaxf(y1, 0.f), d.H); y2 = fminf(fmaxf(y2, 0.f), d.H);
    if (d.scale_factor) {
        x1 /= d.scale_factor[d.b * 4 + 0]; y1 /= d.scale_factor[d.b * 4 + 1];
        x2 /= d.scale_factor[d.b * 4 + 2]; y2 /= d.scale_factor[d.b * 4 + 3];
    }
    const size_t o = d.obase + slot;
    *reinterpret_cast<float4*>(d.lv_boxes + o * 4) = make_float4(x1, y1, x2, y2);
    d.lv_scores[o] = s;
    d.lv_ctr[o] = sigmoidf_(d.iou[d.r0 + pt]);
    d.lv_labels[o] = c;
}

// Radix-select step by one wavefront: the bin (scanning from the highest bin down) in which the cumulative count
// reaches `remaining`, and how many are still needed inside that bin.  Lane l owns nb/64 consecutive bins (highest
// first), so the serial part is nb/64 + log2(64) steps instead of a nb-step dependent LDS chain.
__device__ __forceinline__ void find_kth_bin(const int* hist, int nb, int remaining, int* s_sel) {
    const int lane = threadIdx.x & 63;
    const int per = nb >> 6;
    const int top = nb - 1 - lane * per;               // my bins: top, top-1, ..., top-per+1
    int mine = 0;
    for (int j = 0; j < per; ++j) mine += hist[top - j];
    int incl = mine;
#pragma unroll
    for (int o = 1; o < 64; o <<= 1) {
        const int v = __shfl_up(incl, o, 64);
        if (lane >= o) incl += v;
    }
    const unsigned long long hit = __ballot(incl >= remaining);
    const int owner = hit ? __ffsll((long long)hit) - 1 : 63;
    if (lane == owner) {
        int acc = incl - mine, bin = top;
        for (int j = 0; j < per; ++j, --bin) {
            if (acc + hist[bin] >= remaining || j == per - 1) break;
            acc += hist[bin];
        }
        s_sel[0] = bin;
        s_sel[1] = remaining - acc;
    }
}

__global__ __launch_bounds__(1024) void decode_kernel(const float* __restrict__ cls, const float* __restrict__ reg_u,
                                                      const float* __restrict__ iou, const float* __restrict__ scales,
                                                      const DecLevels L, int B, int C, float score_thr, int nms_pre,
                                                      const float* __restrict__ img_hw,
                                                      const float* __restrict__ scale_factor,
                                                      float* __restrict__ lv_boxes, float* __restrict__ lv_scores,
                                                      float* __restrict__ lv_ctr, int64_t* __restrict__ lv_labels,
                                                      int* __restrict__ lv_count, int force_slow) {
    const int l = blockIdx.x, b = blockIdx.y;
    const int tid = threadIdx.x;
    const int hw = L.h[l] * L.w[l];
    const int r0 = L.row_off[l] + b * hw;
    const int total = hw * C;
    const float* s_in = cls + (size_t)r0 * C;
    __shared__ int hist[2048];
    __shared__ int s_wave[16];
    __shared__ int s_sel[4];

    // ---- count candidates: every wavefront owns a contiguous index range (so that ordered compaction needs no
    //      block-wide scan per 1024 elements, only one 16-entry prefix)
    const int lane = tid & 63, wave = tid >> 6;
    const int per_wave = (((total + 15) / 16) + 63) / 64 * 64;
    const int wb = wave * per_wave;
    const int we = wb + per_wave < total ? wb + per_wave : total;
    int wc = 0;
    constexpr int DU = 8;                         // independent loads in flight per lane (the loop is latency bound)
    for (int i0 = wb; i0 < we; i0 += 64 * DU) {
        float xv[DU];
#pragma unroll
        for (int u = 0; u < DU; ++u) {
            const int i = i0 + u * 64 + lane;
            xv[u] = i < we ? s_in[i] : -INFINITY;  // sigmoid(-inf) = 0: never a candidate
        }
#pragma unroll
        for (int u = 0; u < DU; ++u) wc += __popcll(__ballot(sigmoidf_(xv[u]) > score_thr));
    }
    if (lane == 0) s_wave[wave] = wc;
    __syncthreads();
    int tot = 0, wbase = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        if (i < wave) wbase += s_wave[i];
        tot += s_wave[i];
    }
    __syncthreads();

    if (tot <= DEC_LIST_CAP && !force_slow) {
        // ---- fast path: the candidates (a few % of the scores) are compacted once, in index order, into an LDS list
        //      of (index, score bits); the exact top-k selection and the decode then only touch that list
        extern __shared__ __attribute__((aligned(16))) unsigned long long dec_list[];
        int run = wbase;
        {
            for (int i0 = wb; i0 < we; i0 += 64 * DU) {
                float xv[DU];
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    const int i = i0 + u * 64 + lane;
                    xv[u] = i < we ? s_in[i] : -INFINITY;
                }
#pragma unroll
                for (int u = 0; u < DU; ++u) {
                    const int i = i0 + u * 64 + lane;
                    const float sv = sigmoidf_(xv[u]);
                    const bool pass = sv > score_thr;
                    const unsigned long long bal = __ballot(pass);
                    if (pass) dec_list[run + __popcll(bal & ((1ull << lane) - 1ull))] =
                        ((unsigned long long)(unsigned)i << 32) | (unsigned long long)__float_as_uint(sv);
                    run += __popcll(bal);
                }
            }
        }
        __syncthreads();
        DecCtx dc;
        dc.reg_u = reg_u; dc.iou = iou; dc.scale_factor = scale_factor;
        dc.lv_boxes = lv_boxes; dc.lv_scores = lv_scores; dc.lv_ctr = lv_ctr; dc.lv_labels = lv_labels;
        dc.C = C; dc.wl = L.w[l]; dc.stride_i = L.stride[l]; dc.r0 = r0; dc.b = b;
        dc.stride = (float)L.stride[l]; dc.sc = scales[l]; dc.H = img_hw[b * 2 + 0]; dc.W = img_hw[b * 2 + 1];
        dc.obase = ((size_t)b * L.n + l) * nms_pre;
        if (tot <= nms_pre) {
            for (int t = tid; t < tot; t += 1024)
                decode_one(dc, (int)(dec_list[t] >> 32), __uint_as_float((unsigned)dec_list[t]), t);
            if (tid == 0) lv_count[b * L.n + l] = tot;
            return;
        }
        // exact k-th largest score: 3-pass radix select on the (positive) float bits of the list
        unsigned prefix = 0u, pmask = 0u;
        int remaining = nms_pre;
        const int shifts[3] = {21, 10, 0};
        const int widths[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
            __syncthreads();
            const int sh = shifts[pass];
            const unsigned bm = (1u << widths[pass]) - 1u;
            for (int t = tid; t < tot; t += 1024) {
                const unsigned u = (unsigned)dec_list[t];
                if ((u & pmask) == prefix) atomicAdd(&hist[(u >> sh) & bm], 1);
            }
            __syncthreads();
            if (tid < 64) find_kth_bin(hist, (int)bm + 1, remaining, s_sel);
            __syncthreads();
            prefix |= ((unsigned)s_sel[0]) << sh;
            pmask |= bm << sh;
            remaining = s_sel[1];
            __syncthreads();
        }
        const unsigned thr_bits = prefix;
        const int need_eq = remaining;
        // ordered selection over the list: scores above the threshold + the first need_eq equal to it (index order)
        const int lper = (((tot + 15) / 16) + 63) / 64 * 64;
        const int lb = wave * lper;
        const int le = lb + lper < tot ? lb + lper : tot;
        int ngt = 0, neq = 0;
        for (int t0 = lb; t0 < le; t0 += 64) {
            const int t = t0 + lane;
            const unsigned u = t < le ? (unsigned)dec_list[t] : 0u;
            ngt += __popcll(__ballot(t < le && u > thr_bits));
            neq += __popcll(__ballot(t < le && u == thr_bits));
        }
        int* s_gt = hist;            // reuse: [16] + [16]
        int* s_eq = hist + 16;
        if (lane == 0) { s_gt[wave] = ngt; s_eq[wave] = neq; }
        __syncthreads();
        int gt_before = 0, eq_before = 0;
#pragma unroll
        for (int i = 0; i < 16; ++i)
            if (i < wave) { gt_before += s_gt[i]; eq_before += s_eq[i]; }
        for (int t0 = lb; t0 < le; t0 += 64) {
            const int t = t0 + lane;
            const unsigned long long e = t < le ? dec_list[t] : 0ull;
            const unsigned u = (unsigned)e;
            const bool gt = t < le && u > thr_bits, eq = t < le && u == thr_bits;
            const unsigned long long bgt = __ballot(gt), beq = __ballot(eq);
            const unsigned long long lt = (1ull << lane) - 1ull;
            const int g0 = gt_before + __popcll(bgt & lt), e0 = eq_before + __popcll(beq & lt);
            if (gt || (eq && e0 < need_eq))
                decode_one(dc, (int)(e >> 32), __uint_as_float(u), g0 + (e0 < need_eq ? e0 : need_eq));
            gt_before += __popcll(bgt);
            eq_before += __popcll(beq);
        }
        if (tid == 0) lv_count[b * L.n + l] = nms_pre;
        return;
    }
    // ---- slow path (more than DEC_LIST_CAP scores above the threshold): multi-pass over the raw scores
    const int k = tot < nms_pre ? tot : nms_pre;
    // ---- exact k-th largest score by radix select on the (positive) float bits
    unsigned thr_bits = 0u;   // select bits > thr_bits, plus the first `need_eq` with bits == thr_bits
    int need_eq = 0;
    if (tot > nms_pre) {
        unsigned prefix = 0u, pmask = 0u;
        int remaining = k;
        const int shifts[3] = {21, 10, 0};
        const int widths[3] = {11, 11, 10};
        for (int pass = 0; pass < 3; ++pass) {
            for (int i = tid; i < 2048; i += 1024) hist[i] = 0;
            __syncthreads();
            const int sh = shifts[pass];
            const unsigned bm = (1u << widths[pass]) - 1u;
            for (int i = tid; i < total; i += 1024) {
                const float s = sigmoidf_(s_in[i]);
                if (s > score_thr) {
                    const unsigned u = __float_as_uint(s);
                    if ((u & pmask) == prefix) atomicAdd(&hist[(u >> sh) & bm], 1);
                }
            }
            __syncthreads();
            if (tid < 64) find_kth_bin(hist, (int)bm + 1, remaining, s_sel);
            __syncthreads();
            prefix |= ((unsigned)s_sel[0]) << sh;
            pmask |= bm << sh;
            remaining = s_sel[1];
            __syncthreads();
        }
        thr_bits = prefix;
        need_eq = remaining;
    }
    // ---- ordered compaction + decode
    const float stride = (float)L.stride[l];
    const float sc = scales[l];
    const float H = img_hw[b * 2 + 0], W = img_hw[b * 2 + 1];
    const size_t obase = ((size_t)b * L.n + l) * nms_pre;
    int written = 0, eq_taken = 0;
    for (int c0 = 0; c0 < total; c0 += 1024) {
        const int i = c0 + tid;
        float s = 0.f;
        int sel = 0, is_eq = 0;
        if (i < total) {
            s = sigmoidf_(s_in[i]);
            if (s > score_thr) {
                if (tot <= nms_pre) sel = 1;
                else {
                    const unsigned u = __float_as_uint(s);
                    if (u > thr_bits) sel = 1;
                    else if (u == thr_bits) is_eq = 1;
                }
            }
        }
        if (tot > nms_pre) {
            int teq;
            const int eq_rank = block_excl_scan_1024(is_eq, s_wave, teq);
            if (is_eq && eq_taken + eq_rank < need_eq) sel = 1;
            eq_taken += teq;
        }
        int tsel;
        const int pos = block_excl_scan_1024(sel, s_wave, tsel);
        if (sel) {
            const int pt = i / C, c = i - pt * C;
            const int iy = pt / L.w[l], ix = pt - iy * L.w[l];
            const float cx = (float)(ix * L.stride[l]), cy = (float)(iy * L.stride[l]);
            const float4 u4 = *reinterpret_cast<const float4*>(reg_u + (size_t)(r0 + pt) * 4);
            const float hw8 = 8.f * stride;
            const float top = fmaxf(u4.x * sc, 0.f) * 0.125f * hw8, bottom = fmaxf(u4.y * sc, 0.f) * 0.125f * hw8;
            const float left = fmaxf(u4.z * sc, 0.f) * 0.125f * hw8, right = fmaxf(u4.w * sc, 0.f) * 0.125f * hw8;
            float x1 = cx - left, y1 = cy - top, x2 = cx + right, y2 = cy + bottom;
            x1 = fminf(fmaxf(x1, 0.f), W); x2 = fminf(fmaxf(x2, 0.f), W);
            y1 = fminf(fmaxf(y1, 0.f), H); y2 = fminf(fmaxf(y2, 0.f), H);
            if (scale_factor) {
                x1 /= scale_factor[b * 4 + 0]; y1 /= scale_factor[b * 4 + 1];
                x2 /= scale_factor[b * 4 + 2]; y2 /= scale_factor[b * 4 + 3];
            }
            const size_t o = obase + written + pos;
            *reinterpret_cast<float4*>(lv_boxes + o * 4) = make_float4(x1, y1, x2, y2);
            lv_scores[o] = s;
            lv_ctr[o] = sigmoidf_(iou[r0 + pt]);
            lv_labels[o] = c;
        }
        written += tsel;
    }
    if (tid == 0) lv_count[b * L.n + l] = written;
}

// concatenate the per-level regions of every image into one compact candidate list
__global__ __launch_bounds__(256) void compact_levels_kernel(const float* __restrict__ lv_boxes,
                                                             const float* __restrict__ lv_scores,
                                                             const float* __restrict__ lv_ctr,
                                                             const int64_t* __restrict__ lv_labels,
                                                             const int* __restrict__ lv_count, int nlvl, int nms_pre,
                                                             float* __restrict__ boxes, float* __restrict__ scores,
                                                             float* __restrict__ ctr, int64_t* __restrict__ labels,
                                                             int* __restrict__ count) {
    const int b = blockIdx.x;
    const int cap = nlvl * nms_pre;
    int off = 0;
    for (int l = 0; l < nlvl; ++l) {
        const int n = lv_count[b * nlvl + l];
        const size_t src = ((size_t)b * nlvl + l) * nms_pre, dst = (size_t)b * cap + off;
        for (int i = threadIdx.x; i < n; i += 256) {
            *reinterpret_cast<float4*>(boxes + (dst + i) * 4) = *reinterpret_cast<const float4*>(lv_boxes + (src + i) * 4);
            scores[dst + i] = lv_scores[src + i];
            ctr[dst + i] = lv_ctr[src + i];
            labels[dst + i] = lv_labels[src + i];
        }
        off += n;
    }
    if (threadIdx.x == 0) count[b] = off;
}

extern "C" size_t radet_decode_ws_bytes(int B, int nlvl, int nms_pre) {
    const size_t n = (size_t)B * nlvl * nms_pre;
    return n * (16 + 4 + 4 + 8) + (size_t)B * nlvl * 4 + 256;
}

extern "C" int radet_decode_candidates(const float* cls, const float* reg_u, const float* iou, const float* scales,
                                       const int* level_desc, int nlvl, int B, int num_classes, float score_thr,
                                       int nms_pre, const float* img_hw, const float* scale_factor, float* cand_boxes,
                                       float* cand_scores, float* cand_ctr, int64_t* cand_labels, int* cand_count,
                                       void* ws, void* stream) {
    if (nlvl < 1 || nlvl > RADET_MAX_SEG || nms_pre < 1) return RADET_ERR_ARG;
    DecLevels L;
    L.n = nlvl;
    int row = 0;
    for (int l = 0; l < nlvl; ++l) {
        L.h[l] = level_desc[3 * l]; L.w[l] = level_desc[3 * l + 1]; L.stride[l] = level_desc[3 * l + 2];
        L.row_off[l] = row;
        row += B * L.h[l] * L.w[l];
    }
    L.row_off[nlvl] = row;
    const size_t n = (size_t)B * nlvl * nms_pre;
    char* p = (char*)ws;
    float* lv_boxes = (float*)p; p += n * 16;
    int64_t* lv_labels = (int64_t*)p; p += n * 8;
    float* lv_scores = (float*)p; p += n * 4;
    float* lv_ctr = (float*)p; p += n * 4;
    int* lv_count = (int*)p;
    hipStream_t st = (hipStream_t)stream;
    static bool dec_attr = false;
    if (!dec_attr) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                DEC_LIST_CAP * 8) != hipSuccess)
            return RADET_ERR_LAUNCH;
        dec_attr = true;
    }
    hipLaunchKernelGGL(decode_kernel, dim3(nlvl, B), dim3(1024), DEC_LIST_CAP * 8, st, cls, reg_u, iou, scales, L, B, num_classes,
                       score_thr, nms_pre, img_hw, scale_factor, lv_boxes, lv_scores, lv_ctr, lv_labels, lv_count,
                       getenv("RADET_DECODE_SLOW") ? 1 : 0);   // test hook: force the multi-pass path
    hipLaunchKernelGGL(compact_levels_kernel, dim3(B), dim3(256), 0, st, lv_boxes, lv_scores, lv_ctr, lv_labels, lv_count,
                       nlvl, nms_pre, cand_boxes, cand_scores, cand_ctr, cand_labels, cand_count);
    return radet_check_launch();
}

// ================================================================================================ NMS
__device__ __forceinline__ unsigned sortable_desc(float f) {
    unsigned u = __float_as_uint(f);
    u = (u & 0x80000000u) ? ~u : (u | 0x80000000u);   // ascending order-preserving map
    return ~u;                                         // descending
}

// bitonic sort of m (power of two) 64-bit keys in LDS, ascending; 1024 threads
__device__ void bitonic_sort_u64(unsigned long long* keys, int m) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (m >> 1); t += 1024) {
                const int i = ((t / j) * (j << 1)) + (t % j);
                const int p = i + j;
                const bool up = ((i & k) == 0);
                const unsigned long long a = keys[i], b = keys[p];
                if ((a > b) == up) { keys[i] = b; keys[p] = a; }
            }
            __syncthreads();
        }
    }
}

// the same network over keys in GLOBAL memory (more than 8192 candidates per image do not fit the LDS sort): one
// 1024-thread workgroup, a barrier per step (workgroup-visible global stores); m <= 65536.  A rarely taken path --
// the detector's own configurations stay below 8192 -- so it is written for exactness, not speed.
__device__ void bitonic_sort_u64_global(unsigned long long* __restrict__ keys, int m) {
    const int tid = threadIdx.x;
    for (int k = 2; k <= m; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            for (int t = tid; t < (m >> 1); t += 1024) {
                const int i = ((t / j) * (j << 1)) + (t % j);
                const int p = i + j;
                const bool up = ((i & k) == 0);
                const unsigned long long a = keys[i], b = keys[p];
                if ((a > b) == up) { keys[i] = b; keys[p] = a; }
            }
            __threadfence_block();
            __syncthreads();
        }
    }
}

__device__ __forceinline__ float iou_ref(float x1i, float y1i, float x2i, float y2i, float area_i, float x1j,
                                         float y1j, float x2j, float y2j) {
    const float xl = fmaxf(x1j, x1i), yt = fmaxf(y1j, y1i);
    const float xr = fminf(x2j, x2i), yb = fminf(y2j, y2i);
    const float iw = fmaxf(0.f, xr - xl), ih = fmaxf(0.f, yb - yt);
    const float inter = iw * ih;
    const float area_j = (x2j - x1j) * (y2j - y1j);
    return inter / (area_j + area_i - inter);
}

#define NMS_KREG 1   // label segments of up to 64 * NMS_KREG boxes are clustered out of registers

struct NmsWs {   // per-image global workspace (cap entries each unless noted)
    float* bx;   // [4][cap] sorted coordinates (mode 3: offset coordinates)
    float* vs;   // adjusted vote scores
    float* cs;   // cluster scores
    int* lab;    // labels
    int* oidx;   // original indices
    int* head;   // head position of each sorted position (-1: dropped)
    int* hpos;   // head positions in output order
    int* seg;    // [cap + 1] label-segment starts (sorted positions), seg[nseg] = n
    int* big;    // [cap / (64 * NMS_KREG) + 1] indices of the segments longer than 64 * NMS_KREG
    int* misc;   // [8]: 0 nseg, 1 nbig, 2 nheads, 3 K
    unsigned long long* gkeys;  // [pow2 >= cap] sort keys in global memory (cap > NMS_LDS_SORT only)
    unsigned long long* mask;   // [cap][(cap + 63) / 64] suppression bits of the crowded segments (row = sorted position)
};
#define NMS_LDS_SORT 8192       // candidates per image the LDS sorts / the register-resident resolve pass hold
#define NMS_MAX_CAP 65536       // 16-bit positions in the sort keys
__host__ __device__ inline size_t nms_gkeys_bytes(int cap) {
    if (cap <= NMS_LDS_SORT) return 0;
    size_t m = 1;
    while (m < (size_t)cap) m <<= 1;
    return m * 8;
}

__device__ __forceinline__ NmsWs nms_ws(char* ws_all, size_t ws_per_image, int b, int cap) {
    char* w = ws_all + (size_t)b * ws_per_image;
    NmsWs ws;
    ws.bx = (float*)w; w += (size_t)cap * 16;
    ws.vs = (float*)w; w += (size_t)cap * 4;
    ws.cs = (float*)w; w += (size_t)cap * 4;
    ws.lab = (int*)w; w += (size_t)cap * 4;
    ws.oidx = (int*)w; w += (size_t)cap * 4;
    ws.head = (int*)w; w += (size_t)cap * 4;
    ws.hpos = (int*)w; w += (size_t)cap * 4;
    ws.seg = (int*)w; w += (size_t)(cap + 8) * 4;
    ws.big = (int*)w; w += (size_t)(cap / (64 * NMS_KREG) + 8) * 4;
    ws.misc = (int*)w; w += 64;
    char* a16 = ws_all + (size_t)b * ws_per_image + (((size_t)(w - (ws_all + (size_t)b * ws_per_image)) + 15) & ~(size_t)15);
    ws.gkeys = (unsigned long long*)a16;
    ws.mask = (unsigned long long*)(a16 + nms_gkeys_bytes(cap));
    return ws;
}

// The pipeline is five launches so that ONE image's post-processing spreads over many CUs (the reference is a
// single-threaded host loop; a one-workgroup-per-image kernel leaves 248 of 256 CUs idle at batch 8):
//   nms_sort_kernel     (1 WG / image)            sort by (label, score desc, index), label segments
//   nms_small_kernel    (32 WG x 4 waves / image) greedy clustering of segments <= 64*NMS_KREG boxes, one wave each
//   nms_big_kernel      (1 WG / crowded segment)  blocked exact greedy clustering of longer segments
//   nms_heads_kernel    (1 WG / image)            order the cluster heads, mode 2 / 3 outputs
//   nms_vote_kernel     (1 wave / output box)     score-weighted 1-sigma vote, strictly sequential fp32 sums

// ---- 1. sort + segments
__global__ __launch_bounds__(1024) void nms_sort_kernel(const float* __restrict__ boxes, const float* __restrict__ cscores,
                                                        const float* __restrict__ vscores, const int64_t* __restrict__ labels,
                                                        const int* __restrict__ counts, int cap, int mode,
                                                        char* __restrict__ ws_all, size_t ws_per_image) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* lkeys = reinterpret_cast<unsigned long long*>(smem);  // [m] when m <= NMS_LDS_SORT
    int* s_misc = reinterpret_cast<int*>(smem + (size_t)8192 * 8);            // [64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = counts[b];
    int m = 1;
    while (m < n) m <<= 1;
    if (m < 2) m = 2;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    const bool in_lds = m <= NMS_LDS_SORT;                                    // uniform
    unsigned long long* keys = in_lds ? lkeys : ws.gkeys;
    const float* bsrc = boxes + (size_t)b * cap * 4;
    const float* csrc = cscores + (size_t)b * cap;
    const float* vsrc = vscores + (size_t)b * cap;
    const int64_t* lsrc = labels + (size_t)b * cap;

    // mode 3: class offset = label * (max coordinate + 1)
    float offs_unit = 0.f;
    if (mode == 3) {
        float mx = -INFINITY;
        for (int i = tid; i < n * 4; i += 1024) mx = fmaxf(mx, bsrc[i]);
        mx = wave_max(mx);
        __shared__ float s_mx[16];
        if (lane == 0) s_mx[wave] = mx;
        __syncthreads();
        mx = s_mx[0];
#pragma unroll
        for (int i = 1; i < 16; ++i) mx = fmaxf(mx, s_mx[i]);
        offs_unit = mx + 1.0f;
        __syncthreads();
    }
    // sort by (label asc, score desc, index asc)
    for (int i = tid; i < m; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < n) {
            const unsigned long long lab = (unsigned long long)(lsrc[i] & 0x7FFF);
            key = (lab << 48) | ((unsigned long long)sortable_desc(csrc[i]) << 16) | (unsigned long long)i;
        }
        keys[i] = key;
    }
    __threadfence_block();
    __syncthreads();
    if (in_lds) bitonic_sort_u64(lkeys, m);
    else bitonic_sort_u64_global(ws.gkeys, m);
    for (int i = tid; i < n; i += 1024) {
        const int o = (int)(keys[i] & 0xFFFFull);
        const int lab = (int)lsrc[o];
        const float4 bb = *reinterpret_cast<const float4*>(bsrc + (size_t)o * 4);
        float off = 0.f;
        if (mode == 3) off = (float)lab * offs_unit;
        ws.bx[0 * cap + i] = bb.x + off; ws.bx[1 * cap + i] = bb.y + off;
        ws.bx[2 * cap + i] = bb.z + off; ws.bx[3 * cap + i] = bb.w + off;
        ws.vs[i] = vsrc[o];
        ws.cs[i] = csrc[o];
        ws.lab[i] = lab;
        ws.oidx[i] = o;
        ws.head[i] = -1;
    }
    __threadfence_block();
    __syncthreads();
    // label segments (ordered compaction of the segment starts)
    int nseg_total = 0;
    for (int c0 = 0; c0 < n; c0 += 1024) {
        const int i = c0 + tid;
        const int is_start = (i < n) && (i == 0 || ws.lab[i] != ws.lab[i - 1]);
        int t;
        const int pos = block_excl_scan_1024(is_start, s_misc, t);
        if (is_start) ws.seg[nseg_total + pos] = i;
        nseg_total += t;
    }
    if (tid == 0) { ws.seg[nseg_total] = n; ws.misc[0] = nseg_total; }
    __threadfence_block();
    __syncthreads();
    // list of the crowded segments
    int nbig = 0;
    for (int c0 = 0; c0 < nseg_total; c0 += 1024) {
        const int sgi = c0 + tid;
        const int is_big = (sgi < nseg_total) && (ws.seg[sgi + 1] - ws.seg[sgi] > 64 * NMS_KREG);
        int t;
        const int pos = block_excl_scan_1024(is_big, s_misc, t);
        if (is_big) ws.big[nbig + pos] = sgi;
        nbig += t;
    }
    if (tid == 0) ws.misc[1] = nbig;
}

// ---- 2. short label segments: one wavefront each, boxes and suppressed bits in registers; the head box is
//         broadcast with readlane -> no memory access inside the greedy loop except the fire-and-forget head[]
//         stores.  Same IoU arithmetic and visiting order as the reference's loop (vote_ext.cpp:95-146).
__global__ __launch_bounds__(256) void nms_small_kernel(int cap, int mode, float thr, int iou_enable, float sigma,
                                                        char* __restrict__ ws_all, size_t ws_per_image) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    const int nseg = ws.misc[0];
    const int nwaves = gridDim.x * 4;
    for (int s = blockIdx.x * 4 + (threadIdx.x >> 6); s < nseg; s += nwaves) {
        const int p0 = ws.seg[s], p1 = ws.seg[s + 1];
        const int cnt = p1 - p0;
        if (cnt > 64 * NMS_KREG) continue;                   // nms_big_kernel
        float X1[NMS_KREG], Y1[NMS_KREG], X2[NMS_KREG], Y2[NMS_KREG];
        unsigned supm = 0;
#pragma unroll
        for (int k = 0; k < NMS_KREG; ++k) {
            const int jl = k * 64 + lane;
            const bool v = jl < cnt;
            const int j = v ? p0 + jl : p0;
            X1[k] = ws.bx[j]; Y1[k] = ws.bx[cap + j]; X2[k] = ws.bx[2 * cap + j]; Y2[k] = ws.bx[3 * cap + j];
            if (!v) supm |= 1u << k;
        }
        const int kmax = (cnt + 63) >> 6;
        bool label_done = false;
        for (int i = 0; i < cnt; ++i) {
            const int ki = i >> 6, li = i & 63;
            if ((__shfl(supm, li, 64) >> ki) & 1u) continue;
            if (mode == 1 && label_done) break;              // the rest of the label is dropped (head stays -1)
            float sx1 = X1[0], sy1 = Y1[0], sx2 = X2[0], sy2 = Y2[0];
#pragma unroll
            for (int q = 1; q < NMS_KREG; ++q)
                if (ki == q) { sx1 = X1[q]; sy1 = Y1[q]; sx2 = X2[q]; sy2 = Y2[q]; }
            const float x1 = __shfl(sx1, li, 64), y1 = __shfl(sy1, li, 64);
            const float x2 = __shfl(sx2, li, 64), y2 = __shfl(sy2, li, 64);
            const float area_i = (x2 - x1) * (y2 - y1);
            if (lane == li) { supm |= 1u << ki; ws.head[p0 + i] = p0 + i; }
            label_done = true;
#pragma unroll
            for (int k = 0; k < NMS_KREG; ++k) {
                if (k < ki || k >= kmax) continue;           // uniform: blocks behind the head / past the segment
                const int jl = k * 64 + lane;
                if (jl > i && !((supm >> k) & 1u)) {
                    const float iou = iou_ref(x1, y1, x2, y2, area_i, X1[k], Y1[k], X2[k], Y2[k]);
                    if (iou > thr) {
                        supm |= 1u << k;
                        ws.head[p0 + jl] = p0 + i;
                        if (iou_enable && mode <= 1) {
                            const float f = -(1 - iou) * (1 - iou) / sigma;
                            ws.vs[p0 + jl] = ws.vs[p0 + jl] * expf(f);
                        }
                    }
                }
            }
        }
    }
}

// ---- 3. crowded label segments: blocked greedy NMS with the reference's exact sequential result.
// Per round: (1) wave 0 collects the next <= 64 unsuppressed positions (ordered), (2) resolves the greedy order
// among them in registers (readlane broadcast, one IoU per lane and step) -> the round's true heads, (3) all 16
// wavefronts test every later unsuppressed box against those heads IN ORDER and stop at the first IoU > thr, which
// is exactly the head the sequential loop would have assigned.  Two barriers per 64 candidates instead of a serial
// pass per head.  Coordinates staged in LDS (<= 4096 boxes, else read through L2).
__global__ __launch_bounds__(1024) void nms_big_kernel(int cap, int mode, float thr, int iou_enable, float sigma,
                                                       char* __restrict__ ws_all, size_t ws_per_image) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    float* lx = reinterpret_cast<float*>(smem);                     // [4][4096]
    volatile unsigned char* sup = smem + (size_t)65536;             // [8192] suppressed flags (segment-local)
    int* bmisc = reinterpret_cast<int*>(smem + 65536 + 8192);       // [0]=nc, [1]=nh, [2]=batch_end, [3]=stop
    int* bcand = bmisc + 8;                                         // [64] candidate positions
    int* bhead = bcand + 64;                                        // [64] true-head positions
    float* bhx = reinterpret_cast<float*>(bhead + 64);              // [4][64] true-head coordinates
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    if ((int)blockIdx.x >= ws.misc[1]) return;
    const int s = ws.big[blockIdx.x];
    const int p0 = ws.seg[s], p1 = ws.seg[s + 1];
    const int cnt = p1 - p0;
    const bool in_lds = cnt <= 4096;
    for (int t = tid; t < cnt; t += 1024) {
        sup[t] = 0;
        if (in_lds) {
            lx[t] = ws.bx[p0 + t]; lx[4096 + t] = ws.bx[cap + p0 + t];
            lx[8192 + t] = ws.bx[2 * cap + p0 + t]; lx[12288 + t] = ws.bx[3 * cap + p0 + t];
        }
    }
    __syncthreads();
    const float* c0 = in_lds ? lx : ws.bx + p0;
    const int cs = in_lds ? 4096 : cap;
    int pos = 0;
    bool first_round = true;
    while (pos < cnt) {
        if (wave == 0) {
            // (1) next <= 64 unsuppressed positions at or after pos
            int nc = 0, scan = pos, last = pos;
            while (nc < 64 && scan < cnt) {
                const int probe = scan + lane;
                const bool al = probe < cnt && !sup[probe];
                const unsigned long long bal = __ballot(al);
                const int rank = nc + __popcll(bal & ((1ull << lane) - 1ull));
                if (al && rank < 64) bcand[rank] = probe;
                const int got = __popcll(bal);
                if (nc + got >= 64) {                     // this window holds the 64th candidate: the batch ends right after it
                    const int need = 64 - nc;             // position of the need-th set bit of bal
                    unsigned long long t = bal;
                    for (int q = 1; q < need; ++q) t &= t - 1ull;
                    last = scan + __ffsll((long long)t);
                    nc = 64;
                } else {
                    nc += got;
                    scan += 64;
                    last = scan < cnt ? scan : cnt;
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            // (2) greedy order among the candidates
            const bool have = lane < nc;
            const int cp = have ? bcand[lane] : 0;
            const float bx1 = c0[cp], by1 = c0[cs + cp], bx2 = c0[2 * cs + cp], by2 = c0[3 * cs + cp];
            unsigned long long alive = nc >= 64 ? ~0ull : ((1ull << nc) - 1ull);
            int nh = 0, stop = 0;
            for (int t = 0; t < nc; ++t) {
                if (!((alive >> t) & 1ull)) continue;
                if (mode == 1 && !(first_round && nh == 0)) { stop = 1; break; }   // global vote: one head per label
                const float x1 = __shfl(bx1, t, 64), y1 = __shfl(by1, t, 64);
                const float x2 = __shfl(bx2, t, 64), y2 = __shfl(by2, t, 64);
                const float area_i = (x2 - x1) * (y2 - y1);
                const int hp = __shfl(cp, t, 64);
                bool kill = false;
                float iou = 0.f;
                if (have && lane > t && ((alive >> lane) & 1ull)) {
                    iou = iou_ref(x1, y1, x2, y2, area_i, bx1, by1, bx2, by2);
                    kill = iou > thr;
                }
                if (kill) {
                    sup[cp] = 1;
                    ws.head[p0 + cp] = p0 + hp;
                    if (iou_enable && mode <= 1) {
                        const float f = -(1 - iou) * (1 - iou) / sigma;
                        ws.vs[p0 + cp] = ws.vs[p0 + cp] * expf(f);
                    }
                }
                alive &= ~__ballot(kill);
                if (lane == t) {
                    sup[cp] = 1;
                    ws.head[p0 + cp] = p0 + cp;
                    bhead[nh] = cp;
                    bhx[nh] = bx1; bhx[64 + nh] = by1; bhx[128 + nh] = bx2; bhx[192 + nh] = by2;
                }
                ++nh;
            }
            if (lane == 0) { bmisc[0] = nc; bmisc[1] = nh; bmisc[2] = last; bmisc[3] = stop; }
        }
        __syncthreads();
        const int nh = bmisc[1], batch_end = bmisc[2], stop = bmisc[3];
        // (3) every later box against this round's heads, in head order
        for (int j = batch_end + tid; j < cnt; j += 1024) {
            if (sup[j]) continue;
            const float jx1 = c0[j], jy1 = c0[cs + j], jx2 = c0[2 * cs + j], jy2 = c0[3 * cs + j];
            for (int q = 0; q < nh; ++q) {
                const float x1 = bhx[q], y1 = bhx[64 + q], x2 = bhx[128 + q], y2 = bhx[192 + q];
                const float iou = iou_ref(x1, y1, x2, y2, (x2 - x1) * (y2 - y1), jx1, jy1, jx2, jy2);
                if (iou > thr) {
                    sup[j] = 1;
                    ws.head[p0 + j] = p0 + bhead[q];
                    if (iou_enable && mode <= 1) {
                        const float f = -(1 - iou) * (1 - iou) / sigma;
                        ws.vs[p0 + j] = ws.vs[p0 + j] * expf(f);
                    }
                    break;
                }
            }
        }
        __syncthreads();
        pos = batch_end;
        first_round = false;
        if (stop || (mode == 1 && nh > 0)) break;            // global vote: the label's only head has been applied
    }
}

// ---- 3'. crowded label segments, two passes (default; RADET_NMS_BLOCKED=1 selects nms_big_kernel above):
//   nms_mask_kernel     every CU: bit (i, j) = IoU(box i, box j) > thr for j > i inside a crowded segment, 64 x 64 bits
//                       per wavefront -- the same iou_ref(i = earlier box, j) the sequential loop evaluates;
//   nms_resolve_kernel  1 WG / crowded segment: the greedy pass over the rows needs bit operations only (a position is
//                       a head iff no earlier head's row has its bit; head[j] = the first head whose row has bit j),
//                       then the vote-score decay of every suppressed box from its recomputed IoU with its head.
// Same heads, head assignment and decay factors as the sequential reference loop; the IoU work (n^2 / 2 per segment,
// one CU-bound workgroup in nms_big_kernel) spreads over the whole device.
__global__ __launch_bounds__(256) void nms_mask_kernel(int cap, float thr, char* __restrict__ ws_all, size_t ws_per_image) {
    const int b = blockIdx.y, lane = threadIdx.x & 63;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    const int nbig = ws.misc[1];
    const int capw = (cap + 63) >> 6;
    const int w = blockIdx.x * 4 + (threadIdx.x >> 6), step = gridDim.x * 4;
    int base = 0;                                             // tiles of the segments before this one (rotates the start)
    for (int k = 0; k < nbig; ++k) {
        const int s = ws.big[k];
        const int p0 = ws.seg[s], cnt = ws.seg[s + 1] - p0;
        const int nb = (cnt + 63) >> 6, T = nb * (nb + 1) / 2;
        for (int t = (w + step - base % step) % step; t < T; t += step) {
            int rb = 0, rem = t;
            while (rem >= nb - rb) { rem -= nb - rb; ++rb; }
            const int cb = rb + rem;
            const int r = rb * 64 + lane, c = cb * 64 + lane;
            const int rr = p0 + (r < cnt ? r : 0), cc = p0 + (c < cnt ? c : 0);
            const float x1 = ws.bx[rr], y1 = ws.bx[cap + rr], x2 = ws.bx[2 * cap + rr], y2 = ws.bx[3 * cap + rr];
            const float cx1 = ws.bx[cc], cy1 = ws.bx[cap + cc], cx2 = ws.bx[2 * cap + cc], cy2 = ws.bx[3 * cap + cc];
            const float area_i = (x2 - x1) * (y2 - y1);
            unsigned long long bits = 0ull;
            const int qn = min(64, cnt - cb * 64);
            for (int q = 0; q < qn; ++q) {
                const float jx1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cx1), q));     // q is uniform
                const float jy1 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cy1), q));
                const float jx2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cx2), q));
                const float jy2 = __uint_as_float(__builtin_amdgcn_readlane(__float_as_uint(cy2), q));
                const float iou = iou_ref(x1, y1, x2, y2, area_i, jx1, jy1, jx2, jy2);
                if (iou > thr && cb * 64 + q > r) bits |= 1ull << q;
            }
            if (r < cnt) ws.mask[(size_t)(p0 + r) * capw + cb] = bits;
        }
        base += T;
    }
}

#define NMS_RROWS 32     // mask rows per chunk of the resolve pass
__global__ __launch_bounds__(256) void nms_resolve_kernel(int cap, int mode, int iou_enable, float sigma,
                                                          char* __restrict__ ws_all, size_t ws_per_image) {
    __shared__ unsigned long long rows[NMS_RROWS][128];     // one chunk of mask rows (<= 8192 / 64 words each)
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    if ((int)blockIdx.x >= ws.misc[1]) return;
    const int s = ws.big[blockIdx.x];
    const int p0 = ws.seg[s], cnt = ws.seg[s + 1] - p0;
    if (cnt > NMS_LDS_SORT) return;                          // nms_resolve_any_kernel
    const int capw = (cap + 63) >> 6, nbw = (cnt + 63) >> 6;
    // thread t stages words (t & 127) of rows (t >> 7) + 2 k of the chunk: 16 loads in flight per thread
    unsigned long long st[NMS_RROWS / 2];
    auto fetch = [&](int chunk) {
#pragma unroll
        for (int k = 0; k < NMS_RROWS / 2; ++k) {
            const int r = chunk + (tid >> 7) + 2 * k, wd = tid & 127;
            // words left of the diagonal are never written by the mask pass (and never needed): zero
            st[k] = (r < cnt && wd < nbw && wd >= (r >> 6)) ? ws.mask[(size_t)(p0 + r) * capw + wd] : 0ull;
        }
    };
    unsigned long long sup0 = 0ull, sup1 = 0ull;            // wave 0: lane l holds the suppressed bits of words l, l + 64
    bool stop = false;
    fetch(0);
    for (int chunk = 0; chunk < cnt; chunk += NMS_RROWS) {
#pragma unroll
        for (int k = 0; k < NMS_RROWS / 2; ++k) rows[(tid >> 7) + 2 * k][tid & 127] = st[k];
        __syncthreads();
        if (chunk + NMS_RROWS < cnt) fetch(chunk + NMS_RROWS);      // in flight while wave 0 resolves this chunk
        if (wave == 0 && !stop) {
            const int nr = min(NMS_RROWS, cnt - chunk);
            // the chunk's rows into registers first (all LDS reads in flight together): the greedy chain below is serial,
            // an LDS round trip per head would be most of its time
            unsigned long long q0[NMS_RROWS], q1[NMS_RROWS];
#pragma unroll
            for (int r = 0; r < NMS_RROWS; ++r) { q0[r] = rows[r][lane]; q1[r] = rows[r][lane + 64]; }
#pragma unroll
            for (int r = 0; r < NMS_RROWS; ++r) {
                if (r < nr && !stop) {                             // (no break / continue: the loop must unroll, q0 / q1 are registers)
                    const int i = chunk + r, wd = i >> 6;
                    const unsigned long long sw = wd < 64 ? sup0 : sup1;
                    const unsigned lo = __builtin_amdgcn_readlane((unsigned)sw, wd & 63);
                    const unsigned hi = __builtin_amdgcn_readlane((unsigned)(sw >> 32), wd & 63);
                    const unsigned long long cur = ((unsigned long long)hi << 32) | lo;
                    if (!((cur >> (i & 63)) & 1ull)) {              // not suppressed by an earlier head: head i --
                        // everything its row marks and nobody claimed before belongs to it
                        const unsigned long long r0 = q0[r], r1 = q1[r];
                        unsigned long long n0 = r0 & ~sup0, n1 = r1 & ~sup1;
                        sup0 |= r0; sup1 |= r1;
                        if (lane == 0) ws.head[p0 + i] = p0 + i;
                        while (n0) { const int j = lane * 64 + __ffsll((long long)n0) - 1; ws.head[p0 + j] = p0 + i; n0 &= n0 - 1ull; }
                        while (n1) { const int j = (lane + 64) * 64 + __ffsll((long long)n1) - 1; ws.head[p0 + j] = p0 + i; n1 &= n1 - 1ull; }
                        if (mode == 1) stop = true;                 // global vote: one head per label, the rest is dropped
                    }
                }
            }
        }
        __syncthreads();
    }
    // vote-score decay of the suppressed boxes (the factor the sequential loop applies when it assigns the head)
    if (iou_enable && mode <= 1) {
        __threadfence();
        __syncthreads();
        for (int j = tid; j < cnt; j += 256) {
            const int h = ws.head[p0 + j];
            if (h < 0 || h == p0 + j) continue;
            const float x1 = ws.bx[h], y1 = ws.bx[cap + h], x2 = ws.bx[2 * cap + h], y2 = ws.bx[3 * cap + h];
            const int jj = p0 + j;
            const float iou = iou_ref(x1, y1, x2, y2, (x2 - x1) * (y2 - y1), ws.bx[jj], ws.bx[cap + jj], ws.bx[2 * cap + jj],
                                      ws.bx[3 * cap + jj]);
            const float f = -(1 - iou) * (1 - iou) / sigma;
            ws.vs[jj] = ws.vs[jj] * expf(f);
        }
    }
}

// Segments of more than 8192 boxes (cap > 8192 only): the same greedy pass over the mask rows with the suppressed bits in
// LDS instead of registers -- rows are visited in order, a row whose bit is clear is a head and ORs its row into the
// suppressed set (one barrier per HEAD, none per suppressed row).  Same heads / head assignment / decay as the pass above.
__global__ __launch_bounds__(256) void nms_resolve_any_kernel(int cap, int mode, int iou_enable, float sigma,
                                                              char* __restrict__ ws_all, size_t ws_per_image) {
    __shared__ unsigned long long sup[NMS_MAX_CAP / 64];
    const int b = blockIdx.y, tid = threadIdx.x;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    if ((int)blockIdx.x >= ws.misc[1]) return;
    const int s = ws.big[blockIdx.x];
    const int p0 = ws.seg[s], cnt = ws.seg[s + 1] - p0;
    if (cnt <= NMS_LDS_SORT) return;                         // nms_resolve_kernel
    const int capw = (cap + 63) >> 6, nbw = (cnt + 63) >> 6;
    for (int w = tid; w < nbw; w += 256) sup[w] = 0ull;
    __syncthreads();
    for (int i = 0; i < cnt; ++i) {
        if ((sup[i >> 6] >> (i & 63)) & 1ull) continue;      // uniform: every thread reads the same word
        if (tid == 0) ws.head[p0 + i] = p0 + i;
        const unsigned long long* row = ws.mask + (size_t)(p0 + i) * capw;
        for (int w = (i >> 6) + tid; w < nbw; w += 256) {    // (words left of the diagonal are never written)
            const unsigned long long r = row[w];
            unsigned long long nw = r & ~sup[w];
            sup[w] |= r;
            while (nw) { ws.head[p0 + w * 64 + __ffsll((long long)nw) - 1] = p0 + i; nw &= nw - 1ull; }
        }
        __syncthreads();
        if (mode == 1) break;                                // global vote: one head per label, the rest is dropped
    }
    if (iou_enable && mode <= 1) {
        __threadfence();
        __syncthreads();
        for (int j = tid; j < cnt; j += 256) {
            const int h = ws.head[p0 + j];
            if (h < 0 || h == p0 + j) continue;
            const float x1 = ws.bx[h], y1 = ws.bx[cap + h], x2 = ws.bx[2 * cap + h], y2 = ws.bx[3 * cap + h];
            const int jj = p0 + j;
            const float iou = iou_ref(x1, y1, x2, y2, (x2 - x1) * (y2 - y1), ws.bx[jj], ws.bx[cap + jj], ws.bx[2 * cap + jj],
                                      ws.bx[3 * cap + jj]);
            const float f = -(1 - iou) * (1 - iou) / sigma;
            ws.vs[jj] = ws.vs[jj] * expf(f);
        }
    }
}

// ---- 4. order the cluster heads (score desc, original index asc); mode 2 / 3 outputs
__global__ __launch_bounds__(1024) void nms_heads_kernel(const float* __restrict__ boxes, const int* __restrict__ counts,
                                                         int cap, int mode, int max_out, float* __restrict__ out_boxes,
                                                         float* __restrict__ out_scores, int64_t* __restrict__ out_labels,
                                                         int* __restrict__ out_count, int64_t* __restrict__ aux0,
                                                         int64_t* __restrict__ aux1, char* __restrict__ ws_all,
                                                         size_t ws_per_image) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    unsigned long long* lkeys = reinterpret_cast<unsigned long long*>(smem);  // [m] when m <= NMS_LDS_SORT
    int* s_misc = reinterpret_cast<int*>(smem + (size_t)8192 * 8);            // [64]
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int n = counts[b];
    int m = 1;
    while (m < n) m <<= 1;
    if (m < 2) m = 2;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    const bool in_lds = m <= NMS_LDS_SORT;
    unsigned long long* keys = in_lds ? lkeys : ws.gkeys;
    const float* bsrc = boxes + (size_t)b * cap * 4;
    for (int i = tid; i < m; i += 1024) {
        unsigned long long key = ~0ull;
        if (i < n && ws.head[i] == i)
            key = ((unsigned long long)sortable_desc(ws.cs[i]) << 32) | ((unsigned long long)ws.oidx[i] << 16) |
                  (unsigned long long)i;
        keys[i] = key;
    }
    __threadfence_block();
    __syncthreads();
    if (in_lds) bitonic_sort_u64(lkeys, m);
    else bitonic_sort_u64_global(ws.gkeys, m);
    {
        int c = 0;
        for (int i = tid; i < n; i += 1024) c += keys[i] != ~0ull ? 1 : 0;
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) c += __shfl_xor(c, o, 64);
        if (lane == 0) s_misc[wave] = c;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int i = 0; i < 16; ++i) t += s_misc[i];
            s_misc[33] = t;
        }
        __syncthreads();
    }
    const int nheads = s_misc[33];
    const int K = (max_out > 0 && nheads > max_out) ? max_out : nheads;
    if (tid == 0) { out_count[b] = K; ws.misc[2] = nheads; ws.misc[3] = K; }
    for (int r = tid; r < nheads; r += 1024) ws.hpos[r] = (int)(keys[r] & 0xFFFFull);
    __threadfence_block();
    __syncthreads();

    if (mode == 2) {
        // instance ids (rank of the head) for every box, cluster size at the head
        int64_t* inst = aux0 + (size_t)b * cap;
        int64_t* num = aux1 + (size_t)b * cap;
        for (int i = tid; i < n; i += 1024) num[ws.oidx[i]] = 0;
        __threadfence_block();
        __syncthreads();
        for (int r = tid; r < nheads; r += 1024) {
            const int h = ws.hpos[r];
            int p1 = h + 1;
            while (p1 < n && ws.lab[p1] == ws.lab[h]) ++p1;
            int cntm = 0;
            for (int j = h; j < p1; ++j)
                if (ws.head[j] == h) { inst[ws.oidx[j]] = r; ++cntm; }
            num[ws.oidx[h]] = cntm;
        }
        return;
    }
    if (mode == 3) {
        int64_t* keep = aux0 + (size_t)b * max_out;
        for (int r = tid; r < K; r += 1024) {
            const int h = ws.hpos[r];
            const int o = ws.oidx[h];
            const float4 bb = *reinterpret_cast<const float4*>(bsrc + (size_t)o * 4);
            *reinterpret_cast<float4*>(out_boxes + ((size_t)b * max_out + r) * 4) = bb;
            out_scores[(size_t)b * max_out + r] = ws.cs[h];
            out_labels[(size_t)b * max_out + r] = ws.lab[h];
            keep[r] = o;
        }
    }
}

// ---- 5. vote: one wavefront per output box.  The wave streams the head's label segment in 64-box chunks
// (coalesced), compacts the cluster members in order into LDS, and lanes 0..3 (one per coordinate) run the
// reference's strictly sequential fp32 sums over them (vote_ext.cpp:8-35,148-200): three passes (mean, sigma,
// 1-sigma window), bit-exact.
__global__ __launch_bounds__(64) void nms_vote_kernel(const int* __restrict__ counts, int cap, int max_out,
                                                      float* __restrict__ out_boxes, float* __restrict__ out_scores,
                                                      int64_t* __restrict__ out_labels, char* __restrict__ ws_all,
                                                      size_t ws_per_image) {
    __shared__ float m_vs[64], m_cs[64], m_x[4][64];
    const int b = blockIdx.y, r = blockIdx.x, lane = threadIdx.x;
    const NmsWs ws = nms_ws(ws_all, ws_per_image, b, cap);
    if (r >= ws.misc[3]) return;
    const int n = counts[b];
    const int out_cap = max_out > 0 ? max_out : cap;
    const int h = ws.hpos[r];
    const int lab = ws.lab[h];
    // end of the head's label segment: binary search in the segment starts
    int p1;
    {
        int lo = 0, hi = ws.misc[0];           // seg[lo] <= h < seg[hi]
        while (hi - lo > 1) {
            const int mid = (lo + hi) >> 1;
            if (ws.seg[mid] <= h) lo = mid; else hi = mid;
        }
        p1 = ws.seg[hi];
    }
    (void)n;
    const int d = lane & 3;
    float ssum = 0.f, v = 0.f, sig = 0.f, fs = 0.f, fv = 0.f, mx = -INFINITY;
    for (int pass = 0; pass < 3; ++pass) {
        for (int c0 = h; c0 < p1; c0 += 64) {
            const int j = c0 + lane;
            const bool mem = j < p1 && ws.head[j] == h;
            const unsigned long long bal = __ballot(mem);
            const int cntm = __popcll(bal);
            if (cntm == 0) continue;
            const int rank = __popcll(bal & ((1ull << lane) - 1ull));
            if (mem) {
                m_vs[rank] = ws.vs[j];
                m_cs[rank] = ws.cs[j];
                m_x[0][rank] = ws.bx[j]; m_x[1][rank] = ws.bx[cap + j];
                m_x[2][rank] = ws.bx[2 * cap + j]; m_x[3][rank] = ws.bx[3 * cap + j];
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            if (lane < 4) {
                if (pass == 0) {
                    for (int q = 0; q < cntm; ++q) { ssum += m_vs[q]; v += m_vs[q] * m_x[d][q]; }
                } else if (pass == 1) {
                    for (int q = 0; q < cntm; ++q) sig += m_vs[q] * (m_x[d][q] - v) * (m_x[d][q] - v);
                } else {
                    for (int q = 0; q < cntm; ++q) {
                        const float x = m_x[d][q];
                        if ((v - sig <= x) & (x <= v + sig)) { fv += m_vs[q] * x; fs += m_vs[q]; }
                        mx = fmaxf(mx, m_cs[q]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        if (pass == 0) v = v / ssum;
        else if (pass == 1) sig = sqrtf(sig / ssum);
    }
    if (lane < 4) {
        out_boxes[((size_t)b * out_cap + r) * 4 + d] = fv / fs;
        if (d == 0) {
            out_scores[(size_t)b * out_cap + r] = mx;
            out_labels[(size_t)b * out_cap + r] = lab;
        }
    }
}

static size_t nms_ws_per_image(int cap) {
    return ((size_t)cap * (16 + 4 * 6) + (size_t)(cap + 8) * 4 + (size_t)(cap / (64 * NMS_KREG) + 8) * 4 + 64 +
            nms_gkeys_bytes(cap) + (size_t)cap * ((cap + 63) / 64) * 8 + 16 + 255) / 256 * 256;
}

extern "C" size_t radet_nms_ws_bytes(int B, int cap) { return (size_t)B * nms_ws_per_image(cap); }

extern "C" int radet_nms(const float* boxes, const float* cluster_scores, const float* vote_scores,
                         const int64_t* labels, const int* counts, int B, int cap, int mode, float iou_thr,
                         int iou_enable, float sigma, int max_out, float* out_boxes, float* out_scores,
                         int64_t* out_labels, int* out_count, int64_t* aux0, int64_t* aux1, void* ws, void* stream) {
    if (cap < 1 || cap > NMS_MAX_CAP || mode < 0 || mode > 3 || B < 1) return RADET_ERR_ARG;
    if (mode == 3 && max_out <= 0) return RADET_ERR_ARG;
    const size_t smem_sort = (size_t)8192 * 8 + 64 * 4;
    const size_t smem_big = (size_t)65536 + 8192 + (8 + 64 + 64 + 256) * 4;
    static bool attr_set = false;
    if (!attr_set) {
        if (hipFuncSetAttribute(reinterpret_cast<const void*>(nms_sort_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem_sort) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(nms_heads_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem_sort) != hipSuccess ||
            hipFuncSetAttribute(reinterpret_cast<const void*>(nms_big_kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                (int)smem_big) != hipSuccess)
            return RADET_ERR_LAUNCH;
        attr_set = true;
    }
    hipStream_t st = (hipStream_t)stream;
    const size_t wpi = nms_ws_per_image(cap);
    hipLaunchKernelGGL(nms_sort_kernel, dim3(B), dim3(1024), smem_sort, st, boxes, cluster_scores, vote_scores, labels,
                       counts, cap, mode, (char*)ws, wpi);
    hipLaunchKernelGGL(nms_small_kernel, dim3(32, B), dim3(256), 0, st, cap, mode, iou_thr, iou_enable, sigma, (char*)ws, wpi);
    static const bool blocked = [] { const char* e = getenv("RADET_NMS_BLOCKED"); return e && e[0] == '1'; }();
    if (blocked && cap <= NMS_LDS_SORT) {
        hipLaunchKernelGGL(nms_big_kernel, dim3(cap / (64 * NMS_KREG) + 1, B), dim3(1024), smem_big, st, cap, mode, iou_thr,
                           iou_enable, sigma, (char*)ws, wpi);
    } else {
        hipLaunchKernelGGL(nms_mask_kernel, dim3(B >= 8 ? 64 : (B >= 2 ? 128 : 256), B), dim3(256), 0, st, cap, iou_thr, (char*)ws, wpi);
        hipLaunchKernelGGL(nms_resolve_kernel, dim3(cap / (64 * NMS_KREG) + 1, B), dim3(256), 0, st, cap, mode, iou_enable,
                           sigma, (char*)ws, wpi);
        if (cap > NMS_LDS_SORT)       // a label segment longer than the register-resident pass holds: at most cap / 8192 of them
            hipLaunchKernelGGL(nms_resolve_any_kernel, dim3(cap / (64 * NMS_KREG) + 1, B), dim3(256), 0, st, cap, mode,
                               iou_enable, sigma, (char*)ws, wpi);
    }
    hipLaunchKernelGGL(nms_heads_kernel, dim3(B), dim3(1024), smem_sort, st, boxes, counts, cap, mode, max_out, out_boxes,
                       out_scores, out_labels, out_count, aux0, aux1, (char*)ws, wpi);
    if (mode <= 1)
        hipLaunchKernelGGL(nms_vote_kernel, dim3(max_out > 0 ? max_out : cap, B), dim3(64), 0, st, counts, cap, max_out,
                           out_boxes, out_scores, out_labels, (char*)ws, wpi);
    return radet_check_launch();
}
