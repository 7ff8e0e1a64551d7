// Stand-alone box / loss operators behind the registered classes of the reference API (bbox_overlaps / BboxOverlaps2D,
// TBLRBBoxCoder, FocalLoss, GIoULoss, CrossEntropyLoss(use_sigmoid), multiclass_nms' score filter).  Inside the detector
// the same arithmetic runs fused (loss.hip / decode_nms.hip); these entry points serve callers that use the pieces on
// their own.  All of them are HBM-streaming kernels: one pass over the inputs, coalesced 16-byte box loads, coalesced
// stores, deterministic two-stage sums (per-workgroup partials in a fixed order, fp64 combine).
//
// Arithmetic order follows the reference's PyTorch expressions operation by operation (compiled with
// -ffp-contract=off), so IoU / TBLR results equal the PyTorch-CPU ones bit for bit.
#include "common.h"
#include "../../include/radet_hip.h"

// ------------------------------------------------------------------------------------------------ overlaps
// iou2d_calculator.py:116-158
struct BoxA { float x1, y1, x2, y2, area; };

__device__ __forceinline__ BoxA load_box(const float* p) {
    const float4 v = *reinterpret_cast<const float4*>(p);
    BoxA b;
    b.x1 = v.x; b.y1 = v.y; b.x2 = v.z; b.y2 = v.w;
    b.area = (v.z - v.x) * (v.w - v.y);
    return b;
}

template <int MODE>
__device__ __forceinline__ float overlap_of(const BoxA& a, const BoxA& b, float eps) {
    const float ltx = fmaxf(a.x1, b.x1), lty = fmaxf(a.y1, b.y1);
    const float rbx = fminf(a.x2, b.x2), rby = fminf(a.y2, b.y2);
    const float w = fmaxf(rbx - ltx, 0.f), h = fmaxf(rby - lty, 0.f);
    const float overlap = w * h;
    float uni = MODE == 1 ? a.area : (a.area + b.area) - overlap;
    uni = fmaxf(uni, eps);
    const float iou = overlap / uni;
    if (MODE != 2) return iou;
    const float ex1 = fminf(a.x1, b.x1), ey1 = fminf(a.y1, b.y1);
    const float ex2 = fmaxf(a.x2, b.x2), ey2 = fmaxf(a.y2, b.y2);
    const float ew = fmaxf(ex2 - ex1, 0.f), eh = fmaxf(ey2 - ey1, 0.f);
    const float earea = fmaxf(ew * eh, eps);
    return iou - (earea - uni) / earea;
}

#define OV_ROWS 64     // rows of bboxes1 per workgroup (staged in LDS, read as broadcasts)
#define OV_COLS 256    // columns per workgroup = one per thread; a wavefront stores 64 consecutive floats (256 B)

// grid (ceil(N / 256), ceil(M / 64), batch): thread = one column box in registers, loop over the tile's rows
template <int MODE>
__global__ __launch_bounds__(OV_COLS) void overlaps_matrix_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                                  float* __restrict__ out, int M, int N, float eps) {
    __shared__ BoxA rows[OV_ROWS];
    const int batch = blockIdx.z;
    b1 += (size_t)batch * M * 4;
    b2 += (size_t)batch * N * 4;
    out += (size_t)batch * M * N;
    const int r0 = blockIdx.y * OV_ROWS;
    const int nr = min(OV_ROWS, M - r0);
    if ((int)threadIdx.x < nr) rows[threadIdx.x] = load_box(b1 + (size_t)(r0 + threadIdx.x) * 4);
    __syncthreads();
    const int col = blockIdx.x * OV_COLS + threadIdx.x;
    if (col >= N) return;
    const BoxA c = load_box(b2 + (size_t)col * 4);
    float* o = out + (size_t)r0 * N + col;
#pragma unroll 4
    for (int r = 0; r < nr; ++r) o[(size_t)r * N] = overlap_of<MODE>(rows[r], c, eps);
}

template <int MODE>
__global__ __launch_bounds__(256) void overlaps_aligned_kernel(const float* __restrict__ b1, const float* __restrict__ b2,
                                                               float* __restrict__ out, size_t n, float eps) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256)
        out[i] = overlap_of<MODE>(load_box(b1 + i * 4), load_box(b2 + i * 4), eps);
}

extern "C" int radet_bbox_overlaps(const float* bboxes1, const float* bboxes2, float* out, int batch, int M, int N, int mode,
                                   int aligned, float eps, void* stream) {
    if (batch < 0 || M < 0 || N < 0 || mode < 0 || mode > 2 || (aligned && M != N)) return RADET_ERR_ARG;
    if (batch == 0 || M == 0 || N == 0) return RADET_OK;
    hipStream_t st = (hipStream_t)stream;
    if (aligned) {
        const size_t n = (size_t)batch * M;
        const int blocks = (int)((n + 255) / 256 < 4096 ? (n + 255) / 256 : 4096);
        if (mode == 0) hipLaunchKernelGGL(overlaps_aligned_kernel<0>, dim3(blocks), dim3(256), 0, st, bboxes1, bboxes2, out, n, eps);
        else if (mode == 1) hipLaunchKernelGGL(overlaps_aligned_kernel<1>, dim3(blocks), dim3(256), 0, st, bboxes1, bboxes2, out, n, eps);
        else hipLaunchKernelGGL(overlaps_aligned_kernel<2>, dim3(blocks), dim3(256), 0, st, bboxes1, bboxes2, out, n, eps);
        return radet_check_launch();
    }
    const int gy = (M + OV_ROWS - 1) / OV_ROWS;
    if (gy > 65535 || batch > 65535) return RADET_ERR_ARG;
    const dim3 grid((N + OV_COLS - 1) / OV_COLS, gy, batch);
    if (mode == 0) hipLaunchKernelGGL(overlaps_matrix_kernel<0>, grid, dim3(OV_COLS), 0, st, bboxes1, bboxes2, out, M, N, eps);
    else if (mode == 1) hipLaunchKernelGGL(overlaps_matrix_kernel<1>, grid, dim3(OV_COLS), 0, st, bboxes1, bboxes2, out, M, N, eps);
    else hipLaunchKernelGGL(overlaps_matrix_kernel<2>, grid, dim3(OV_COLS), 0, st, bboxes1, bboxes2, out, M, N, eps);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------------ TBLR coder
// tblr_bbox_coder.py:71-172
struct Norm4 { float v[4]; };

__global__ __launch_bounds__(256) void tblr_encode_kernel(const float4* __restrict__ priors, const float4* __restrict__ gts,
                                                          float4* __restrict__ out, int n, Norm4 nm, int by_wh) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 p = priors[i], g = gts[i];
        const float cx = (p.x + p.z) / 2.f, cy = (p.y + p.w) / 2.f;
        float top = cy - g.y, bottom = g.w - cy, left = cx - g.x, right = g.z - cx;
        if (by_wh) {
            const float w = p.z - p.x, h = p.w - p.y;
            top /= h; bottom /= h; left /= w; right /= w;
        }
        out[i] = make_float4(top / nm.v[0], bottom / nm.v[1], left / nm.v[2], right / nm.v[3]);
    }
}

__global__ __launch_bounds__(256) void tblr_decode_kernel(const float4* __restrict__ priors, const float4* __restrict__ tblr,
                                                          float4* __restrict__ out, int n, Norm4 nm, int by_wh, float max_h,
                                                          float max_w, int clip) {
    for (int i = blockIdx.x * 256 + threadIdx.x; i < n; i += gridDim.x * 256) {
        const float4 p = priors[i], t = tblr[i];
        float top = t.x * nm.v[0], bottom = t.y * nm.v[1], left = t.z * nm.v[2], right = t.w * nm.v[3];
        const float cx = (p.x + p.z) / 2.f, cy = (p.y + p.w) / 2.f;
        if (by_wh) {
            const float w = p.z - p.x, h = p.w - p.y;
            top *= h; bottom *= h; left *= w; right *= w;
        }
        float x1 = cx - left, x2 = cx + right, y1 = cy - top, y2 = cy + bottom;
        if (clip) {
            x1 = fminf(fmaxf(x1, 0.f), max_w); y1 = fminf(fmaxf(y1, 0.f), max_h);
            x2 = fminf(fmaxf(x2, 0.f), max_w); y2 = fminf(fmaxf(y2, 0.f), max_h);
        }
        out[i] = make_float4(x1, y1, x2, y2);
    }
}

static inline int grid_for(size_t n, int cap = 4096) {
    const size_t b = (n + 255) / 256;
    return (int)(b < (size_t)cap ? (b ? b : 1) : (size_t)cap);
}

extern "C" int radet_tblr_encode(const float* priors, const float* gts, float* out, int n, const float* normalizer4,
                                 int normalize_by_wh, void* stream) {
    if (n < 0 || !normalizer4) return RADET_ERR_ARG;
    if (n == 0) return RADET_OK;
    Norm4 nm;
    for (int i = 0; i < 4; ++i) nm.v[i] = normalizer4[i];
    hipLaunchKernelGGL(tblr_encode_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float4*)priors,
                       (const float4*)gts, (float4*)out, n, nm, normalize_by_wh);
    return radet_check_launch();
}

extern "C" int radet_tblr_decode(const float* priors, const float* tblr, float* out, int n, const float* normalizer4,
                                 int normalize_by_wh, float max_h, float max_w, int clip, void* stream) {
    if (n < 0 || !normalizer4) return RADET_ERR_ARG;
    if (n == 0) return RADET_OK;
    Norm4 nm;
    for (int i = 0; i < 4; ++i) nm.v[i] = normalizer4[i];
    hipLaunchKernelGGL(tblr_decode_kernel, dim3(grid_for(n)), dim3(256), 0, (hipStream_t)stream, (const float4*)priors,
                       (const float4*)tblr, (float4*)out, n, nm, normalize_by_wh, max_h, max_w, clip);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------------ elementwise losses
// Common contract (losses/utils.py:24-51 `weight_reduce_loss`): elem = loss(x) * weight; either the elements are the
// result (reduction 'none') or their sum, scaled by the caller (mean / avg_factor / loss_weight) in radet_loss_finalize.
// weight_cols: 0 = no weight, 1 = one weight per row, C = one per element.  In the forward kernels `scale` multiplies
// the stored elements only (loss_weight under reduction 'none'); the partial sums stay unscaled.
#define LOSS_PARTIALS 1024

__device__ __forceinline__ float softplus_(float x) { return fmaxf(x, 0.f) + log1pf(expf(-fabsf(x))); }

__device__ __forceinline__ float weight_at(const float* w, int wcols, size_t row, int c, int C) {
    if (wcols == 0) return 1.f;
    return wcols == 1 ? w[row] : w[row * C + c];
}

__device__ __forceinline__ void block_partial(float acc, float* partials) {
    __shared__ float red[4];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0 && partials) partials[blockIdx.x] = (red[0] + red[1]) + (red[2] + red[3]);
}

// the scalar that multiplies the upstream gradient in the backward kernels
__device__ __forceinline__ float bwd_scale(const float* grad_scalar, const float* avg_factor, float scale) {
    float s = scale;
    if (grad_scalar) s *= grad_scalar[0];
    if (avg_factor) s /= avg_factor[0];
    return s;
}

// sigmoid focal loss, mmcv.ops.sigmoid_focal_loss semantics (focal_loss.py:44-86): target = class index per row,
// any index outside [0, C) is background
template <bool BWD>
__global__ __launch_bounds__(256) void focal_elem_kernel(const float* __restrict__ x, const int64_t* __restrict__ target,
                                                         const float* __restrict__ weight, int wcols, size_t N, int C,
                                                         float gamma, float alpha, float* __restrict__ out,
                                                         float* __restrict__ partials, const float* __restrict__ grad_elem,
                                                         const float* __restrict__ grad_scalar,
                                                         const float* __restrict__ avg_factor, float scale) {
    const size_t total = N * C;
    const float gs = BWD ? bwd_scale(grad_scalar, avg_factor, scale) : 0.f;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C;
        const int c = (int)(i - r * C);
        const float v = x[i];
        const float p = 1.f / (1.f + expf(-v));
        const float wt = weight_at(weight, wcols, r, c, C);
        const bool pos = target[r] == (int64_t)c;
        float loss, dx;
        if (pos) {
            const float q = 1.f - p, logp = -softplus_(-v);
            const float mod = gamma == 2.f ? q * q : powf(q, gamma);
            loss = -alpha * mod * logp;
            dx = alpha * mod * (gamma * p * logp - q);
        } else {
            const float log1mp = -softplus_(v);
            const float mod = gamma == 2.f ? p * p : powf(p, gamma);
            loss = -(1.f - alpha) * mod * log1mp;
            dx = (1.f - alpha) * mod * (p - gamma * (1.f - p) * log1mp);
        }
        if (BWD) {
            out[i] = dx * wt * (grad_elem ? grad_elem[i] * gs : gs);
        } else {
            const float e = loss * wt;
            if (out) out[i] = e * scale;
            acc += e;
        }
    }
    if (!BWD) block_partial(acc, partials);
}

// binary cross entropy with logits against float targets (cross_entropy_loss.py:57-92, F.binary_cross_entropy_with_logits)
template <bool BWD>
__global__ __launch_bounds__(256) void bce_elem_kernel(const float* __restrict__ x, const float* __restrict__ target,
                                                       const float* __restrict__ weight, int wcols, size_t N, int C,
                                                       float* __restrict__ out, float* __restrict__ partials,
                                                       const float* __restrict__ grad_elem, const float* __restrict__ grad_scalar,
                                                       const float* __restrict__ avg_factor, float scale) {
    const size_t total = N * C;
    const float gs = BWD ? bwd_scale(grad_scalar, avg_factor, scale) : 0.f;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const size_t r = i / C;
        const int c = (int)(i - r * C);
        const float v = x[i], t = target[i];
        const float wt = weight_at(weight, wcols, r, c, C);
        if (BWD) {
            const float sig = 1.f / (1.f + expf(-v));
            out[i] = (sig - t) * wt * (grad_elem ? grad_elem[i] * gs : gs);
        } else {
            const float e = ((fmaxf(v, 0.f) - v * t) + log1pf(expf(-fabsf(v)))) * wt;
            if (out) out[i] = e * scale;
            acc += e;
        }
    }
    if (!BWD) block_partial(acc, partials);
}

// GIoU loss 1 - giou(pred, target) per aligned pair (iou_loss.py:82-98) and its gradient w.r.t. pred
__device__ __forceinline__ float sel_gt_(float a, float b) { return a > b ? 1.f : (a == b ? 0.5f : 0.f); }
__device__ __forceinline__ float sel_lt_(float a, float b) { return a < b ? 1.f : (a == b ? 0.5f : 0.f); }

template <bool BWD>
__global__ __launch_bounds__(256) void giou_elem_kernel(const float* __restrict__ pred, const float* __restrict__ target,
                                                        const float* __restrict__ weight, size_t N, float eps,
                                                        float* __restrict__ out, float* __restrict__ partials,
                                                        const float* __restrict__ grad_elem, const float* __restrict__ grad_scalar,
                                                        const float* __restrict__ avg_factor, float scale) {
    const float gs = BWD ? bwd_scale(grad_scalar, avg_factor, scale) : 0.f;
    float acc = 0.f;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < N; i += (size_t)gridDim.x * 256) {
        const BoxA a = load_box(pred + i * 4), b = load_box(target + i * 4);
        const float wt = weight ? weight[i] : 1.f;
        if (!BWD) {
            const float e = (1.f - overlap_of<2>(a, b, eps)) * wt;
            if (out) out[i] = e * scale;
            acc += e;
            continue;
        }
        // d(1 - giou)/d(pred), same case analysis as autograd over the reference expression (ties: max/min split 1/2,
        // clamp(min=0) passes the gradient at 0, torch.max(x, eps) splits it at equality)
        const float ltx = fmaxf(a.x1, b.x1), lty = fmaxf(a.y1, b.y1), rbx = fminf(a.x2, b.x2), rby = fminf(a.y2, b.y2);
        const float dw = rbx - ltx, dh = rby - lty;
        const float iw = fmaxf(dw, 0.f), ih = fmaxf(dh, 0.f);
        const float I = iw * ih;
        const float Uraw = (a.area + b.area) - I;
        const float U = fmaxf(Uraw, eps);
        const float ex1 = fminf(a.x1, b.x1), ey1 = fminf(a.y1, b.y1), ex2 = fmaxf(a.x2, b.x2), ey2 = fmaxf(a.y2, b.y2);
        const float dew = ex2 - ex1, deh = ey2 - ey1;
        const float ew = fmaxf(dew, 0.f), eh = fmaxf(deh, 0.f);
        const float Eraw = ew * eh;
        const float E = fmaxf(Eraw, eps);
        const float gg = -wt * (grad_elem ? grad_elem[i] * gs : gs);      // d/d giou
        const float g_E = -gg * U / (E * E);
        const float g_U = gg / E - gg * I / (U * U);
        float g_I = gg / U;
        const float g_Uraw = g_U * sel_gt_(Uraw, eps);
        const float g_area1 = g_Uraw;
        g_I -= g_Uraw;
        const float g_dw = g_I * ih * (dw >= 0.f ? 1.f : 0.f), g_dh = g_I * iw * (dh >= 0.f ? 1.f : 0.f);
        const float g_Eraw = g_E * sel_gt_(Eraw, eps);
        const float g_dew = g_Eraw * eh * (dew >= 0.f ? 1.f : 0.f), g_deh = g_Eraw * ew * (deh >= 0.f ? 1.f : 0.f);
        const float aw = a.x2 - a.x1, ah = a.y2 - a.y1;
        float4 g;
        g.x = -g_dw * sel_gt_(a.x1, b.x1) - g_dew * sel_lt_(a.x1, b.x1) - g_area1 * ah;
        g.y = -g_dh * sel_gt_(a.y1, b.y1) - g_deh * sel_lt_(a.y1, b.y1) - g_area1 * aw;
        g.z = g_dw * sel_lt_(a.x2, b.x2) + g_dew * sel_gt_(a.x2, b.x2) + g_area1 * ah;
        g.w = g_dh * sel_lt_(a.y2, b.y2) + g_deh * sel_gt_(a.y2, b.y2) + g_area1 * aw;
        *reinterpret_cast<float4*>(out + i * 4) = g;
    }
    if (!BWD) block_partial(acc, partials);
}

__global__ __launch_bounds__(256) void loss_finalize_kernel(const float* __restrict__ partials, int n,
                                                            const float* __restrict__ avg_factor, float scale,
                                                            float* __restrict__ out) {
    __shared__ double red[256];
    double a = 0.0;
    for (int k = threadIdx.x; k < n; k += 256) a += (double)partials[k];
    red[threadIdx.x] = a;
    __syncthreads();
    for (int o = 128; o > 0; o >>= 1) {
        if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        float s = (float)red[0];
        if (avg_factor) s = s / avg_factor[0];
        out[0] = s * scale;
    }
}

extern "C" int radet_loss_partials(size_t n_elem) { return grid_for(n_elem, LOSS_PARTIALS); }

extern "C" int radet_loss_finalize(const float* partials, int npartials, const float* avg_factor, float scale, float* out,
                                   void* stream) {
    if (npartials < 0) return RADET_ERR_ARG;
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(256), 0, (hipStream_t)stream, partials, npartials, avg_factor, scale, out);
    return radet_check_launch();
}

static inline bool bad_wcols(int wcols, int C) { return !(wcols == 0 || wcols == 1 || wcols == C); }

extern "C" int radet_sigmoid_focal_loss(const float* logits, const int64_t* target, const float* weight, int weight_cols,
                                        size_t N, int C, float gamma, float alpha, float* loss_elem, float elem_scale,
                                        float* partials, void* stream) {
    if (C < 1 || bad_wcols(weight_cols, C)) return RADET_ERR_ARG;
    hipLaunchKernelGGL(focal_elem_kernel<false>, dim3(grid_for(N * C, LOSS_PARTIALS)), dim3(256), 0, (hipStream_t)stream, logits,
                       target, weight, weight_cols, N, C, gamma, alpha, loss_elem, partials, nullptr, nullptr, nullptr, elem_scale);
    return radet_check_launch();
}

extern "C" int radet_sigmoid_focal_loss_bwd(const float* logits, const int64_t* target, const float* weight, int weight_cols,
                                            size_t N, int C, float gamma, float alpha, const float* grad_elem,
                                            const float* grad_scalar, const float* avg_factor, float scale, float* dlogits,
                                            void* stream) {
    if (C < 1 || bad_wcols(weight_cols, C)) return RADET_ERR_ARG;
    if (N == 0) return RADET_OK;
    hipLaunchKernelGGL(focal_elem_kernel<true>, dim3(grid_for(N * C)), dim3(256), 0, (hipStream_t)stream, logits, target, weight,
                       weight_cols, N, C, gamma, alpha, dlogits, nullptr, grad_elem, grad_scalar, avg_factor, scale);
    return radet_check_launch();
}

extern "C" int radet_bce_logits_loss(const float* logits, const float* target, const float* weight, int weight_cols, size_t N,
                                     int C, float* loss_elem, float elem_scale, float* partials, void* stream) {
    if (C < 1 || bad_wcols(weight_cols, C)) return RADET_ERR_ARG;
    hipLaunchKernelGGL(bce_elem_kernel<false>, dim3(grid_for(N * C, LOSS_PARTIALS)), dim3(256), 0, (hipStream_t)stream, logits,
                       target, weight, weight_cols, N, C, loss_elem, partials, nullptr, nullptr, nullptr, elem_scale);
    return radet_check_launch();
}

extern "C" int radet_bce_logits_loss_bwd(const float* logits, const float* target, const float* weight, int weight_cols,
                                         size_t N, int C, const float* grad_elem, const float* grad_scalar,
                                         const float* avg_factor, float scale, float* dlogits, void* stream) {
    if (C < 1 || bad_wcols(weight_cols, C)) return RADET_ERR_ARG;
    if (N == 0) return RADET_OK;
    hipLaunchKernelGGL(bce_elem_kernel<true>, dim3(grid_for(N * C)), dim3(256), 0, (hipStream_t)stream, logits, target, weight,
                       weight_cols, N, C, dlogits, nullptr, grad_elem, grad_scalar, avg_factor, scale);
    return radet_check_launch();
}

extern "C" int radet_giou_loss(const float* pred, const float* target, const float* weight, size_t N, float eps,
                               float* loss_elem, float elem_scale, float* partials, void* stream) {
    hipLaunchKernelGGL(giou_elem_kernel<false>, dim3(grid_for(N, LOSS_PARTIALS)), dim3(256), 0, (hipStream_t)stream, pred, target,
                       weight, N, eps, loss_elem, partials, nullptr, nullptr, nullptr, elem_scale);
    return radet_check_launch();
}

extern "C" int radet_giou_loss_bwd(const float* pred, const float* target, const float* weight, size_t N, float eps,
                                   const float* grad_elem, const float* grad_scalar, const float* avg_factor, float scale,
                                   float* dpred, void* stream) {
    if (N == 0) return RADET_OK;
    hipLaunchKernelGGL(giou_elem_kernel<true>, dim3(grid_for(N)), dim3(256), 0, (hipStream_t)stream, pred, target, weight, N, eps,
                       dpred, nullptr, grad_elem, grad_scalar, avg_factor, scale);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------------ score filter
// multiclass_nms' `valid_mask = scores > score_thr; inds = valid_mask.nonzero()` (bbox_nms.py:54-56): ordered stream
// compaction by one workgroup (ballot prefix inside a wavefront, running offsets across wavefronts), no atomics, so
// the index list comes out ascending like torch.nonzero.  count[0] = number of selected entries.
__global__ __launch_bounds__(1024) void threshold_compact_kernel(const float* __restrict__ scores, size_t n, float thr,
                                                                 int64_t* __restrict__ idx, int* __restrict__ count) {
    __shared__ int wcnt[16];
    __shared__ int base_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) base_s = 0;
    __syncthreads();
    for (size_t c0 = 0; c0 < n; c0 += 1024) {
        const size_t i = c0 + tid;
        const bool sel = i < n && scores[i] > thr;
        const unsigned long long m = __ballot(sel);
        if (lane == 0) wcnt[wv] = __popcll(m);
        __syncthreads();
        int off = base_s;
        for (int k = 0; k < wv; ++k) off += wcnt[k];
        if (sel) idx[off + __popcll(m & ((1ull << lane) - 1ull))] = (int64_t)i;
        __syncthreads();
        if (tid == 0) {
            int t = 0;
            for (int k = 0; k < 16; ++k) t += wcnt[k];
            base_s += t;
        }
        __syncthreads();
    }
    if (tid == 0) count[0] = base_s;
}

extern "C" int radet_threshold_compact(const float* scores, size_t n, float thr, int64_t* idx, int* count, void* stream) {
    hipLaunchKernelGGL(threshold_compact_kernel, dim3(1), dim3(1024), 0, (hipStream_t)stream, scores, n, thr, idx, count);
    return radet_check_launch();
}
