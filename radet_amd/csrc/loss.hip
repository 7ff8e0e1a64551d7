// Head loss on the GPU: targets -> sigmoid focal (dense) -> positive-sample GIoU / IoU-BCE, forward values and
// gradients w.r.t. the head outputs in three stream-ordered launches, no host synchronisation
// (the reference syncs twice: nonzero() and `if num_pos > 0`, radet_head.py:245-261).
//
//   prep   : 1 workgroup; labels per row (incl. the "ignore -> last gt label" quirk, radet_head.py:388-390),
//            ordered compaction of the positive rows, num_pos = sum of their weights.
//   focal  : HBM-streaming over [R, C]; per-workgroup partial sums (deterministic two-stage reduction).
//   pos    : 1 workgroup over the P positive rows: TBLR encode/decode, aligned IoU (detached weight),
//            GIoU loss, BCE-with-logits on the IoU logit, analytic backward incl. Scale/ReLU of atss_reg.
#include "common.h"
#include "../../include/radet_hip.h"

struct LossLevels {
    int n;
    int h[RADET_MAX_SEG], w[RADET_MAX_SEG], stride[RADET_MAX_SEG];
    int row_off[RADET_MAX_SEG + 1];  // level-major row offsets (B * cumulative h*w)
    int pt_off[RADET_MAX_SEG + 1];   // per-image point offsets (cumulative h*w)
};

struct RowInfo { int lvl, n, pix, pt; float cx, cy, stride; };

__device__ __forceinline__ RowInfo decode_row(const LossLevels& L, int r) {
    RowInfo o;
    int l = 0;
#pragma unroll
    for (int i = 1; i < RADET_MAX_SEG; ++i)
        if (i < L.n && r >= L.row_off[i]) l = i;
    const int hw = L.h[l] * L.w[l];
    const int local = r - L.row_off[l];
    o.lvl = l;
    o.n = local / hw;
    o.pix = local - o.n * hw;
    o.pt = L.pt_off[l] + o.pix;
    const int iy = o.pix / L.w[l], ix = o.pix - iy * L.w[l];
    o.stride = (float)L.stride[l];
    o.cx = (float)(ix * L.stride[l]);
    o.cy = (float)(iy * L.stride[l]);
    return o;
}

// ws layout (ints): [0] = P, [1] = num_pos as float bits, [2] = focal partial count, [16 .. 16+R) = positive rows,
// [16+R .. 16+2R) = focal partials (float bits)
#define WS_HDR 16

// Pass 1 (one thread per row, R / 256 workgroups): label and TBLR target of every row, the workgroup's positive rows
// compacted in row order into its own slice of the scratch list, its positive count and weight sum.
// Pass 2 (one workgroup): exclusive scan of the per-workgroup counts, slices copied to their final place (ascending
// rows, as the single-workgroup version produced them), weight sums added in a fixed tree.
// The former single 1024-thread workgroup walked 25 rows per thread: ~10 k instructions per thread on ONE CU, 70 us
// between the head forward and the loss kernels with the rest of the chip idle.
#define LP_BLK 256
__device__ __forceinline__ int* lp_scratch(int* ws, int R) { return ws + WS_HDR + R + 1024 + 16; }     // see radet_head_loss_ws_ints
__device__ __forceinline__ int lp_blocks(int R) { return (R + LP_BLK - 1) / LP_BLK; }

__global__ __launch_bounds__(LP_BLK) void loss_prep_rows_kernel(const int64_t* __restrict__ gt_labels,
                                                                const float* __restrict__ gt_boxes,
                                                                const int* __restrict__ gt_off,
                                                                const int64_t* __restrict__ p2g,
                                                                const float* __restrict__ pw, const LossLevels L, int B,
                                                                int N, int R, int num_classes,
                                                                int64_t* __restrict__ labels_out,
                                                                float* __restrict__ tgt_out, int* __restrict__ ws) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = blockIdx.x * LP_BLK + tid;
    bool pos = false;
    float w = 0.f;
    if (r < R) {
        const RowInfo ri = decode_row(L, r);
        const int G = gt_off[ri.n + 1] - gt_off[ri.n];
        const int64_t g = p2g[(size_t)ri.n * N + ri.pt];
        int64_t lab = num_classes;
        if (G > 0 && g > -1) lab = gt_labels[gt_off[ri.n] + (g == 0 ? G - 1 : (int)g - 1)];
        if (labels_out) labels_out[r] = lab;
        if (tgt_out) {
            float4 t = make_float4(0.f, 0.f, 0.f, 0.f);
            if (G > 0 && g > 0) {
                const float* gb = gt_boxes + (size_t)(gt_off[ri.n] + (int)g - 1) * 4;
                const float hw8 = 8.f * ri.stride;
                t.x = (ri.cy - gb[1]) / hw8 / 0.125f;
                t.y = (gb[3] - ri.cy) / hw8 / 0.125f;
                t.z = (ri.cx - gb[0]) / hw8 / 0.125f;
                t.w = (gb[2] - ri.cx) / hw8 / 0.125f;
            }
            *reinterpret_cast<float4*>(tgt_out + (size_t)r * 4) = t;
        }
        if (lab >= 0 && lab < num_classes) {
            pos = true;
            w = pw[(size_t)ri.n * N + ri.pt];
        }
    }
    const unsigned long long bal = __ballot(pos);
    const int before = __popcll(bal & ((1ull << lane) - 1ull));
    const float wsum = wave_sum(w);
    __shared__ int wc[LP_BLK / 64];
    __shared__ float ww[LP_BLK / 64];
    if (lane == 0) { wc[wave] = __popcll(bal); ww[wave] = wsum; }
    __syncthreads();
    int base = 0;
    for (int i = 0; i < wave; ++i) base += wc[i];
    int* scratch = lp_scratch(ws, R);
    const int nb = lp_blocks(R);
    if (pos) scratch[blockIdx.x * LP_BLK + base + before] = r;
    if (tid == 0) {
        scratch[nb * LP_BLK + blockIdx.x] = (wc[0] + wc[1]) + (wc[2] + wc[3]);
        scratch[nb * LP_BLK + nb + blockIdx.x] = __float_as_int((ww[0] + ww[1]) + (ww[2] + ww[3]));
    }
}

__global__ __launch_bounds__(1024) void loss_prep_scan_kernel(int R, int* __restrict__ ws) {
    const int tid = threadIdx.x;
    const int nb = lp_blocks(R);
    const int* scratch = lp_scratch(ws, R);
    const int* bcnt = scratch + nb * LP_BLK;
    const int* bw = bcnt + nb;
    const int per = (nb + 1023) / 1024;
    const int b0 = tid * per, b1 = min(nb, b0 + per);
    int cnt = 0;
    float wsum = 0.f;
    for (int b = b0; b < b1; ++b) {
        cnt += bcnt[b];
        wsum += __int_as_float(bw[b]);
    }
    __shared__ int scnt[1024];
    __shared__ float swt[1024];
    scnt[tid] = cnt;
    swt[tid] = wsum;
    __syncthreads();
    // inclusive scan of counts (Hillis-Steele), tree sum of weights
    for (int o = 1; o < 1024; o <<= 1) {
        const int v = tid >= o ? scnt[tid - o] : 0;
        __syncthreads();
        scnt[tid] += v;
        __syncthreads();
    }
    for (int o = 512; o > 0; o >>= 1) {
        if (tid < o) swt[tid] += swt[tid + o];
        __syncthreads();
    }
    int pos = scnt[tid] - cnt;
    for (int b = b0; b < b1; ++b) {
        const int n = bcnt[b];
        for (int i = 0; i < n; ++i) ws[WS_HDR + pos + i] = scratch[b * LP_BLK + i];
        pos += n;
    }
    if (tid == 0) {
        ws[0] = scnt[1023];
        ws[1] = __float_as_int(swt[0]);
    }
}

__device__ __forceinline__ float softplus_f(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

// one thread per (row, class) element
__global__ __launch_bounds__(256) void focal_kernel(const float* __restrict__ cls, const int64_t* __restrict__ gt_labels,
                                                    const int* __restrict__ gt_off, const int64_t* __restrict__ p2g,
                                                    const float* __restrict__ pw, const LossLevels L, int B, int N,
                                                    int R, int C, float alpha, float gamma,
                                                    const float* __restrict__ grad_scale, float* __restrict__ dcls,
                                                    int dcls_ld, int* __restrict__ ws) {
    const float num_pos = __int_as_float(ws[1]);
    const float avg = num_pos + (float)B;
    const float gs = (grad_scale ? grad_scale[0] : 1.f) / avg;
    const size_t total = (size_t)R * C;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / C);
        const int c = (int)(i - (size_t)r * C);
        const RowInfo ri = decode_row(L, r);
        const int G = gt_off[ri.n + 1] - gt_off[ri.n];
        const int64_t g = p2g[(size_t)ri.n * N + ri.pt];
        int64_t lab = C;
        if (G > 0 && g > -1) lab = gt_labels[gt_off[ri.n] + (g == 0 ? G - 1 : (int)g - 1)];
        const float wt = pw[(size_t)ri.n * N + ri.pt];
        const float x = cls[i];
        const float p = 1.f / (1.f + expf(-x));
        float loss, dx;
        if (lab == c) {
            const float q = 1.f - p;
            const float logp = -softplus_f(-x);
            const float mod = gamma == 2.f ? q * q : powf(q, gamma);
            loss = -alpha * mod * logp;
            dx = alpha * mod * (gamma * p * logp - q);
        } else {
            const float log1mp = -softplus_f(x);
            const float mod = gamma == 2.f ? p * p : powf(p, gamma);
            loss = -(1.f - alpha) * mod * log1mp;
            dx = (1.f - alpha) * mod * (p - gamma * (1.f - p) * log1mp);
        }
        acc += loss * wt;
        dcls[(size_t)r * dcls_ld + c] = dx * wt * gs;
    }
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) ws[WS_HDR + R + blockIdx.x] = __float_as_int((red[0] + red[1]) + (red[2] + red[3]));
}

struct PosTerms { float w, one_minus_giou, bce, pwt; };

__device__ __forceinline__ float sel_gt(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }  // d max(a,b)/da
__device__ __forceinline__ float sel_lt(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }  // d min(a,b)/da

template <int NT>
__device__ __forceinline__ float block_sum(float v, float* red) {
    // deterministic tree over NT threads
    const int tid = threadIdx.x;
    __syncthreads();
    red[tid] = v;
    __syncthreads();
    for (int o = NT / 2; o > 0; o >>= 1) {
        if (tid < o) red[tid] += red[tid + o];
        __syncthreads();
    }
    const float r = red[0];
    __syncthreads();
    return r;
}

__global__ __launch_bounds__(256) void pos_loss_kernel(const float* __restrict__ reg_u, const float* __restrict__ iou_logit,
                                                       const float* __restrict__ scales,
                                                       const float* __restrict__ gt_boxes,
                                                       const int* __restrict__ gt_off, const int64_t* __restrict__ p2g,
                                                       const float* __restrict__ pw, const LossLevels L, int B, int N,
                                                       int R, float lbw, float eps, const float* __restrict__ grad_scale,
                                                       float* __restrict__ losses, float* __restrict__ dreg_u,
                                                       int dreg_ld, float* __restrict__ diou, int diou_ld,
                                                       float* __restrict__ dscales, const int* __restrict__ ws,
                                                       int n_focal_partials, int postact) {
    __shared__ float red[256];
    const int tid = threadIdx.x;
    const int P = ws[0];
    const float num_pos = __int_as_float(ws[1]);
    // --- finalise the classification loss from the focal partials
    {
        double a = 0.0;
        for (int k = tid; k < n_focal_partials; k += 256) a += (double)__int_as_float(ws[WS_HDR + R + k]);
        const float s = block_sum<256>((float)a, red);
        if (tid == 0) losses[0] = s / (num_pos + (float)B);
    }
    if (!(num_pos > 0.f)) {
        if (tid == 0) { losses[1] = 0.f; losses[2] = 0.f; }
        if (tid < L.n) dscales[tid] = 0.f;
        return;
    }
    float sw = 0.f, swl = 0.f, sb = 0.f, sp = 0.f;
    for (int pass = 0; pass < 2; ++pass) {
        float inv_sw = 0.f, inv_sp = 0.f, g1 = 1.f, g2 = 1.f;
        float ds[RADET_MAX_SEG];
#pragma unroll
        for (int l = 0; l < RADET_MAX_SEG; ++l) ds[l] = 0.f;
        if (pass == 1) {
            inv_sw = 1.f / sw;
            inv_sp = 1.f / sp;
            if (grad_scale) { g1 = grad_scale[1]; g2 = grad_scale[2]; }
        }
        for (int k = tid; k < P; k += 256) {
            const int r = ws[WS_HDR + k];
            const RowInfo ri = decode_row(L, r);
            const int64_t g = p2g[(size_t)ri.n * N + ri.pt];
            const float pwt = pw[(size_t)ri.n * N + ri.pt];
            const float sc = scales[ri.lvl];
            const float4 u = *reinterpret_cast<const float4*>(reg_u + (size_t)r * 4);
            // prediction (top, bottom, left, right) after Scale + ReLU
            const float v0 = u.x * sc, v1 = u.y * sc, v2 = u.z * sc, v3 = u.w * sc;
            const float p0 = postact ? v0 : fmaxf(v0, 0.f), p1 = postact ? v1 : fmaxf(v1, 0.f);
            const float p2 = postact ? v2 : fmaxf(v2, 0.f), p3 = postact ? v3 : fmaxf(v3, 0.f);
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
            if (g > 0) {
                const float* gb = gt_boxes + (size_t)(gt_off[ri.n] + (int)g - 1) * 4;
                const float hw8 = 8.f * ri.stride;
                t0 = (ri.cy - gb[1]) / hw8 / 0.125f;
                t1 = (gb[3] - ri.cy) / hw8 / 0.125f;
                t2 = (ri.cx - gb[0]) / hw8 / 0.125f;
                t3 = (gb[2] - ri.cx) / hw8 / 0.125f;
            }
            const float hw8 = 8.f * ri.stride;
            // decode: d = tblr * normalizer * (h|w)
            const float px1 = ri.cx - p2 * 0.125f * hw8, py1 = ri.cy - p0 * 0.125f * hw8;
            const float px2 = ri.cx + p3 * 0.125f * hw8, py2 = ri.cy + p1 * 0.125f * hw8;
            const float tx1 = ri.cx - t2 * 0.125f * hw8, ty1 = ri.cy - t0 * 0.125f * hw8;
            const float tx2 = ri.cx + t3 * 0.125f * hw8, ty2 = ri.cy + t1 * 0.125f * hw8;
            const float area1 = (px2 - px1) * (py2 - py1), area2 = (tx2 - tx1) * (ty2 - ty1);
            const float ltx = fmaxf(px1, tx1), lty = fmaxf(py1, ty1), rbx = fminf(px2, tx2), rby = fminf(py2, ty2);
            const float dw = rbx - ltx, dh = rby - lty;
            const float iw = fmaxf(dw, 0.f), ih = fmaxf(dh, 0.f);
            const float I = iw * ih;
            const float Uraw = area1 + area2 - I;
            const float U = fmaxf(Uraw, eps);
            const float iou = I / U;
            const float eltx = fminf(px1, tx1), elty = fminf(py1, ty1), erbx = fmaxf(px2, tx2), erby = fmaxf(py2, ty2);
            const float dew = erbx - eltx, deh = erby - elty;
            const float ew = fmaxf(dew, 0.f), eh = fmaxf(deh, 0.f);
            const float Eraw = ew * eh;
            const float E = fmaxf(Eraw, eps);
            const float giou = iou - (E - U) / E;
            const float wgt = fmaxf(iou, 1e-12f) * pwt;
            const float x = iou_logit[r];
            if (pass == 0) {
                sw += wgt;
                swl += wgt * (1.f - giou);
                sb += pwt * (fmaxf(x, 0.f) - x * iou + log1pf(expf(-fabsf(x))));
                sp += pwt;
            } else {
                // d loss_bbox / d giou
                const float gg = -lbw * wgt * inv_sw * g1;
                const float g_E = -gg * U / (E * E);
                float g_U = gg / E - gg * I / (U * U);
                float g_I = gg / U;
                const float g_Uraw = g_U * sel_gt(Uraw, eps);
                const float g_area1 = g_Uraw;
                g_I -= g_Uraw;
                const float g_dw = g_I * ih * (dw >= 0.f ? 1.f : 0.f);
                const float g_dh = g_I * iw * (dh >= 0.f ? 1.f : 0.f);
                const float g_Eraw = g_E * sel_gt(Eraw, eps);
                const float g_dew = g_Eraw * eh * (dew >= 0.f ? 1.f : 0.f);
                const float g_deh = g_Eraw * ew * (deh >= 0.f ? 1.f : 0.f);
                float gx1 = -g_dw * sel_gt(px1, tx1) - g_dew * sel_lt(px1, tx1) - g_area1 * (py2 - py1);
                float gy1 = -g_dh * sel_gt(py1, ty1) - g_deh * sel_lt(py1, ty1) - g_area1 * (px2 - px1);
                float gx2 = g_dw * sel_lt(px2, tx2) + g_dew * sel_gt(px2, tx2) + g_area1 * (py2 - py1);
                float gy2 = g_dh * sel_lt(py2, ty2) + g_deh * sel_gt(py2, ty2) + g_area1 * (px2 - px1);
                // decode backward: top<-y1(-), bottom<-y2, left<-x1(-), right<-x2 ; times normalizer*size
                const float k8 = 0.125f * hw8;
                const float gp0 = -gy1 * k8, gp1 = gy2 * k8, gp2 = -gx1 * k8, gp3 = gx2 * k8;
                // postact: reg_u already is the head's bbox_pred (after Scale + ReLU, the tensor RADetHead.loss receives),
                // so the gradient is taken w.r.t. that tensor and the ReLU mask does not apply
                const float gv0 = (postact || p0 > 0.f) ? gp0 : 0.f, gv1 = (postact || p1 > 0.f) ? gp1 : 0.f;
                const float gv2 = (postact || p2 > 0.f) ? gp2 : 0.f, gv3 = (postact || p3 > 0.f) ? gp3 : 0.f;
                float4 du;
                du.x = gv0 * sc; du.y = gv1 * sc; du.z = gv2 * sc; du.w = gv3 * sc;
                *reinterpret_cast<float4*>(dreg_u + (size_t)r * dreg_ld) = du;
                const float dsv = (gv0 * u.x + gv1 * u.y) + (gv2 * u.z + gv3 * u.w);
#pragma unroll
                for (int l = 0; l < RADET_MAX_SEG; ++l)
                    if (l == ri.lvl) ds[l] += dsv;
                const float sig = 1.f / (1.f + expf(-x));
                diou[(size_t)r * diou_ld] = pwt * (sig - iou) * inv_sp * g2;
            }
        }
        if (pass == 0) {
            sw = block_sum<256>(sw, red);
            swl = block_sum<256>(swl, red);
            sb = block_sum<256>(sb, red);
            sp = block_sum<256>(sp, red);
            if (tid == 0) {
                losses[1] = lbw * swl / sw;
                losses[2] = sb / sp;
            }
        } else {
#pragma unroll
            for (int l = 0; l < RADET_MAX_SEG; ++l) {
                if (l < L.n) {
                    const float s = block_sum<256>(ds[l], red);
                    if (tid == 0) dscales[l] = s;
                }
            }
        }
    }
}

__global__ void zero_sparse_kernel(float* __restrict__ dreg, int dreg_ld, float* __restrict__ diou, int diou_ld, int R) {
    const int r = blockIdx.x * blockDim.x + threadIdx.x;
    if (r < R) {
        *reinterpret_cast<float4*>(dreg + (size_t)r * dreg_ld) = make_float4(0.f, 0.f, 0.f, 0.f);
        diou[(size_t)r * diou_ld] = 0.f;
    }
}

static int fill_levels(LossLevels* L, const int* level_desc, int nlvl, int B) {
    if (nlvl < 1 || nlvl > RADET_MAX_SEG) return RADET_ERR_ARG;
    L->n = nlvl;
    int row = 0, pt = 0;
    for (int l = 0; l < nlvl; ++l) {
        L->h[l] = level_desc[3 * l];
        L->w[l] = level_desc[3 * l + 1];
        L->stride[l] = level_desc[3 * l + 2];
        L->row_off[l] = row;
        L->pt_off[l] = pt;
        row += B * L->h[l] * L->w[l];
        pt += L->h[l] * L->w[l];
    }
    L->row_off[nlvl] = row;
    L->pt_off[nlvl] = pt;
    for (int l = nlvl; l < RADET_MAX_SEG; ++l) { L->h[l] = 1; L->w[l] = 1; L->stride[l] = 1; }
    return RADET_OK;
}

#define FOCAL_BLOCKS 1024

// header | positive rows [R] | focal partials [FOCAL_BLOCKS] | pad 16 | per-workgroup positive lists [nb * 256] | counts
// [nb] | weight sums [nb]   (nb = ceil(R / 256): lp_scratch() / loss_prep_*_kernel)
static_assert(FOCAL_BLOCKS == 1024, "lp_scratch() hard-codes the focal partial count");
extern "C" int radet_head_loss_ws_ints(int R) {
    const int nb = (R + LP_BLK - 1) / LP_BLK;
    return WS_HDR + R + FOCAL_BLOCKS + 16 + nb * LP_BLK + 2 * nb;
}

extern "C" int radet_head_loss(const float* cls, const float* reg_u, const float* iou, const float* scales,
                               const float* gt_boxes, const int64_t* gt_labels, const int* gt_off, const int64_t* p2g,
                               const float* pw, const int* level_desc, int nlvl, int B, int num_classes, float alpha,
                               float gamma, float loss_bbox_weight, float giou_eps, const float* grad_scale,
                               float* losses, float* dcls, int dcls_ld, float* dreg_u, int dreg_ld, float* diou,
                               int diou_ld, float* dscales, int64_t* labels_out, float* bbox_targets_out, int flags,
                               int* ws, void* stream) {
    LossLevels L;
    int rc = fill_levels(&L, level_desc, nlvl, B);
    if (rc) return rc;
    const int R = L.row_off[nlvl], N = L.pt_off[nlvl];
    if (dcls_ld < num_classes || dreg_ld < 4 || (dreg_ld & 3) || diou_ld < 1) return RADET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(loss_prep_rows_kernel, dim3((R + LP_BLK - 1) / LP_BLK), dim3(LP_BLK), 0, st, gt_labels, gt_boxes,
                       gt_off, p2g, pw, L, B, N, R, num_classes, labels_out, bbox_targets_out, ws);
    hipLaunchKernelGGL(loss_prep_scan_kernel, dim3(1), dim3(1024), 0, st, R, ws);
    const size_t total = (size_t)R * num_classes;
    int fb = (int)((total + 255) / 256);
    if (fb > FOCAL_BLOCKS) fb = FOCAL_BLOCKS;
    hipLaunchKernelGGL(focal_kernel, dim3(fb), dim3(256), 0, st, cls, gt_labels, gt_off, p2g, pw, L, B, N, R, num_classes,
                       alpha, gamma, grad_scale, dcls, dcls_ld, ws);
    hipLaunchKernelGGL(zero_sparse_kernel, dim3((R + 255) / 256), dim3(256), 0, st, dreg_u, dreg_ld, diou, diou_ld, R);
    hipLaunchKernelGGL(pos_loss_kernel, dim3(1), dim3(256), 0, st, reg_u, iou, scales, gt_boxes, gt_off, p2g, pw, L, B, N,
                       R, loss_bbox_weight, giou_eps, grad_scale, losses, dreg_u, dreg_ld, diou, diou_ld, dscales, ws,
                       fb, flags & 1);
    return radet_check_launch();
}

__global__ void scale_relu_kernel(const float4* __restrict__ u, const float* __restrict__ scales, float4* __restrict__ out,
                                  const LossLevels L, int R) {
    for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < R; r += gridDim.x * blockDim.x) {
        int l = 0;
#pragma unroll
        for (int i = 1; i < RADET_MAX_SEG; ++i)
            if (i < L.n && r >= L.row_off[i]) l = i;
        const float s = scales[l];
        const float4 v = u[r];
        out[r] = make_float4(fmaxf(v.x * s, 0.f), fmaxf(v.y * s, 0.f), fmaxf(v.z * s, 0.f), fmaxf(v.w * s, 0.f));
    }
}

extern "C" int radet_scale_relu(const float* reg_u, const float* scales, float* out, const int* level_desc, int nlvl,
                                int B, void* stream) {
    LossLevels L;
    int rc = fill_levels(&L, level_desc, nlvl, B);
    if (rc) return rc;
    const int R = L.row_off[nlvl];
    hipLaunchKernelGGL(scale_relu_kernel, dim3((R + 255) / 256), dim3(256), 0, (hipStream_t)stream, (const float4*)reg_u,
                       scales, (float4*)out, L, R);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ anchors
__global__ void anchors_kernel(float4* __restrict__ out, const LossLevels L, int base_scale) {
    const int N = L.pt_off[L.n];
    for (int p = blockIdx.x * blockDim.x + threadIdx.x; p < N; p += gridDim.x * blockDim.x) {
        int l = 0;
#pragma unroll
        for (int i = 1; i < RADET_MAX_SEG; ++i)
            if (i < L.n && p >= L.pt_off[i]) l = i;
        const int pix = p - L.pt_off[l];
        const int iy = pix / L.w[l], ix = pix - iy * L.w[l];
        const float half = 0.5f * (float)(base_scale * L.stride[l]);
        const float cx = (float)(ix * L.stride[l]), cy = (float)(iy * L.stride[l]);
        out[p] = make_float4(cx - half, cy - half, cx + half, cy + half);
    }
}

extern "C" int radet_grid_anchors(float* out, const int* level_desc, int nlvl, int octave_base_scale, void* stream) {
    LossLevels L;
    int rc = fill_levels(&L, level_desc, nlvl, 1);
    if (rc) return rc;
    const int N = L.pt_off[nlvl];
    hipLaunchKernelGGL(anchors_kernel, dim3((N + 255) / 256), dim3(256), 0, (hipStream_t)stream, (float4*)out, L,
                       octave_base_scale);
    return radet_check_launch();
}
