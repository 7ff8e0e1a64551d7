// Non-GEMM layers of the conv stack (HBM-bound; coalesced 16-byte accesses, LDS reductions):
//   stem 7x7/2 conv + folded BN + ReLU (NCHW image in -> NHWC out)      resnet.py:558-570,622-630
//   maxpool 3x3/2                                                          resnet.py:570,630
//   weight fold (BN scale into OHWI weights, + transposed copy for dgrad) / grad unfold
//   GroupNorm(32)+ReLU forward / backward over multi-level buffers         atss_head.py:60-76
//   nearest upsample-add forward / backward (FPN top-down path)            fpn.py:182-191
#include "common.h"
#include <type_traits>
#include "../../include/radet_hip.h"

// ------------------------------------------------------------------------------------------ stem
// 7x7 / 2 conv (3 -> 64) + folded BN + ReLU as an implicit GEMM on the matrix cores: M = output pixels, N = 64 channels,
// K = 7 * 7 * 3 = 147 (padded to 148 with a zero weight row).  One workgroup = 8 x 32 output pixels of one image: the
// 21 x 69 x 3 input patch and the [K][64] weights sit in LDS; wave w owns output rows 2w, 2w + 1 (two 32-pixel M tiles)
// x both 32-channel halves = 4 accumulators, so a k pair costs 2 + 2 LDS reads for 4 v_mfma_f32_32x32x2_f32.
// The A operand is gathered straight from the patch: element (pixel, k = (ky, kx, c)) = patch[c][2 oy + ky][2 ox + kx],
// i.e. lane base + a per-k constant (two constants per k pair, selected by the lane's k half).  The former VALU kernel
// (64 accumulators per thread, 147 x 64 FMAs each) took 265 us of the step at 21 TFLOP/s.
#define STEM_TR 8
#define STEM_TC 32
#define STEM_S 72                     // patch row stride (floats)
#define STEM_LDW 65                   // weight row stride: transposing LDS writes and the B reads are conflict free
__host__ __device__ constexpr int stem_aoff(int k) {      // patch offset of reduction index k = (ky * 7 + kx) * 3 + c
    return k >= 147 ? stem_aoff(146) : (k % 3) * ((2 * STEM_TR + 5) * STEM_S) + (k / 21) * STEM_S + (k / 3) % 7;
}

template <class T>
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ img, const float* __restrict__ wf,
                                                   const float* __restrict__ bias, T* __restrict__ y, int H,
                                                   int W, int Ho, int Wo, int B, unsigned* y_amax = nullptr) {
    float am = 0.f;                          // largest output this thread stores (y_amax: the tensor's amax slot, common.h "h2")
    constexpr int PR = 2 * STEM_TR + 5, PC = 2 * STEM_TC + 5, CS = PR * STEM_S, KP = 148;
    __shared__ float sw[KP * STEM_LDW];          // [k][n]
    __shared__ float sx[3 * CS];                 // [c][patch row][patch col]
    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // wf is [o][ky][kx][c] (OHWI, folded): coalesced read, transposed LDS write (bank = (k + o) % 32: no conflicts)
    // both fill loops keep 8 independent (clamped, unconditional) loads in flight per thread: with a conditional load per
    // iteration the compiler emits load -> wait -> LDS write, one L2 round trip per element
    constexpr int NWF = 147 * 64, NPX = 3 * PR * PC;
    for (int i0 = tid; i0 < NWF; i0 += 256 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u;
            v[u] = wf[i < NWF ? i : NWF - 1];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u;
            if (i < NWF) {
                const int o = i / 147, k = i - o * 147;
                sw[k * STEM_LDW + o] = v[u];
            }
        }
    }
    if (tid < 64) sw[147 * STEM_LDW + tid] = 0.f;
    // persistent workgroups: the 37 KiB of transposed weights are staged once per workgroup, not once per tile (1200 tiles
    // at bs 4: 45 MB of L2 reads and as many index computations as the patches themselves)
    const int tiles_x = (Wo + STEM_TC - 1) / STEM_TC, tiles_y = (Ho + STEM_TR - 1) / STEM_TR;
    const int ntiles = tiles_x * tiles_y * B;
    const float bv0 = bias[li], bv1 = bias[32 + li];
  for (int tile = blockIdx.x; tile < ntiles; tile += gridDim.x) {
    const int n = tile / (tiles_x * tiles_y);
    const int trem = tile - n * tiles_x * tiles_y;
    const int oy0 = (trem / tiles_x) * STEM_TR, ox0 = (trem % tiles_x) * STEM_TC;
    const int iy0 = oy0 * 2 - 3, ix0 = ox0 * 2 - 3;
    __syncthreads();                                              // every wave is done with the previous patch
    const float* imgn = img + (size_t)n * 3 * H * W;
    for (int i0 = tid; i0 < NPX; i0 += 256 * 8) {
        float v[8];
        int dst[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = i0 + 256 * u;
            const int ic = i < NPX ? i : NPX - 1;
            const int c = ic / (PR * PC);
            const int rem = ic - c * PR * PC;
            const int yy = rem / PC, xx = rem - yy * PC;
            const int iy = iy0 + yy, ix = ix0 + xx;
            const bool in = iy >= 0 && iy < H && ix >= 0 && ix < W;
            const float t = imgn[((size_t)c * H + (in ? iy : 0)) * W + (in ? ix : 0)];
            v[u] = in ? t : 0.f;
            dst[u] = i < NPX ? c * CS + yy * STEM_S + xx : -1;
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (dst[u] >= 0) sx[dst[u]] = v[u];
    }
    __syncthreads();
    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const float* a0 = &sx[(2 * (2 * wave)) * STEM_S + 2 * li];          // output row 2w, column li
    const float* a1 = a0 + 2 * STEM_S;                                   // output row 2w + 1
    const float* b0 = &sw[lh * STEM_LDW + li];
    // operands of k pair kk + 1 are read before the MFMAs of pair kk (the LDS latency hides behind 4 MFMAs)
    int off = lh ? stem_aoff(1) : stem_aoff(0);
    float x0 = a0[off], x1 = a1[off], w0 = b0[0], w1 = b0[32];
#pragma unroll
    for (int kk = 0; kk < KP / 2; ++kk) {
        float nx0 = 0.f, nx1 = 0.f, nw0 = 0.f, nw1 = 0.f;
        if (kk + 1 < KP / 2) {
            off = lh ? stem_aoff(2 * kk + 3) : stem_aoff(2 * kk + 2);
            nx0 = a0[off]; nx1 = a1[off];
            nw0 = b0[(2 * kk + 2) * STEM_LDW]; nw1 = b0[(2 * kk + 2) * STEM_LDW + 32];
        }
        acc[0][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, w0, acc[0][0], 0, 0, 0);
        acc[0][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x0, w1, acc[0][1], 0, 0, 0);
        acc[1][0] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, w0, acc[1][0], 0, 0, 0);
        acc[1][1] = __builtin_amdgcn_mfma_f32_32x32x2f32(x1, w1, acc[1][1], 0, 0, 0);
        x0 = nx0; x1 = nx1; w0 = nw0; w1 = nw1;
    }
    // D layout: lane column li = channel within the 32-channel half, register r = pixel (r & 3) + 8 (r >> 2) + 4 lh
    const bool interior = ox0 + STEM_TC <= Wo && oy0 + STEM_TR <= Ho;       // uniform: unguarded stores (no exec juggling,
#pragma unroll                                                                // no conservative waits between them)
    for (int i = 0; i < 2; ++i) {
        const int oy = oy0 + 2 * wave + i;
        T* row = y + ((size_t)(n * Ho + (oy < Ho ? oy : 0)) * Wo + ox0 + 4 * lh) * 64 + li;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const float bv = j ? bv1 : bv0;
            if (interior) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float v = fmaxf(acc[i][j][r] + bv, 0.f);
                    row[((r & 3) + 8 * (r >> 2)) * 64 + j * 32] = (T)v;
                    am = fmaxf(am, v);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int ox = ox0 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                    if (oy < Ho && ox < Wo) {
                        const float v = fmaxf(acc[i][j][r] + bv, 0.f);
                        row[((r & 3) + 8 * (r >> 2)) * 64 + j * 32] = (T)v;
                        am = fmaxf(am, v);
                    }
                }
            }
        }
    }
  }
  if (y_amax) radet_amax_publish(am, y_amax);                  // (persistent workgroups: one atomic per wave and launch)
}

extern "C" int radet_stem_conv_bn_relu_a(const float* img_nchw, const float* wf_ohwi, const float* bias, float* y_nhwc,
                                         int B, int H, int W, void* y_amax, void* stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int ntiles = ((Wo + STEM_TC - 1) / STEM_TC) * ((Ho + STEM_TR - 1) / STEM_TR) * B;
    dim3 grid(ntiles < 512 ? ntiles : 512);                      // 2 workgroups per CU (57 KiB of LDS each)
    hipLaunchKernelGGL(stem_kernel<float>, grid, dim3(256), 0, (hipStream_t)stream, img_nchw, wf_ohwi, bias, y_nhwc, H, W,
                       Ho, Wo, B, (unsigned*)y_amax);
    return radet_check_launch();
}
extern "C" int radet_stem_conv_bn_relu(const float* img_nchw, const float* wf_ohwi, const float* bias, float* y_nhwc,
                                       int B, int H, int W, void* stream) {
    return radet_stem_conv_bn_relu_a(img_nchw, wf_ohwi, bias, y_nhwc, B, H, W, nullptr, stream);
}

extern "C" int radet_stem_conv_bn_relu_h(const float* img_nchw, const float* wf_ohwi, const float* bias, void* y_nhwc,
                                         int B, int H, int W, void* stream) {
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const int ntiles = ((Wo + STEM_TC - 1) / STEM_TC) * ((Ho + STEM_TR - 1) / STEM_TR) * B;
    dim3 grid(ntiles < 512 ? ntiles : 512);
    hipLaunchKernelGGL(stem_kernel<__bf16>, grid, dim3(256), 0, (hipStream_t)stream, img_nchw, wf_ohwi, bias,
                       (__bf16*)y_nhwc, H, W, Ho, Wo, B);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ maxpool
template <class T>
__global__ void maxpool_kernel(const T* __restrict__ x, T* __restrict__ y, int B, int H, int W, int C4,
                               int Ho, int Wo, unsigned* amax = nullptr, _Float16* yq = nullptr, unsigned* yq_amax = nullptr,
                               const unsigned* x_amax = nullptr) {
    float am = 0.f;                          // largest |y| stored by this thread (amax: the output's h2 slot, common.h)
    // yq: the output once more as fp16 plane pairs, scaled by the power of two of amax(x) (a max-pool's outputs are inputs)
    float qs = 1.f, qs2 = 2048.f;
    if (yq) {
        const unsigned bits = radet_amax_read(x_amax);
        if (blockIdx.x == 0 && threadIdx.x == 0) radet_amax_store(yq_amax, bits);
        const int e = radet_h2_exp(bits);
        qs = radet_pow2(e); qs2 = radet_pow2(e + 11);
    }
    // one thread = TWO horizontally adjacent outputs of one channel quad: their 3 x 3 windows share a column, 15 loads
    // instead of 18 (all unconditional: an out-of-range tap is clamped onto the border pixel, which is inside the window
    // already -- max is idempotent -- instead of skipped: a skipped load is a branch and a wait per tap)
    const int Wp = (Wo + 1) / 2;
    const size_t total = (size_t)B * Ho * Wp * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int oxp = (int)(p % Wp);
        p /= Wp;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const int ox = 2 * oxp;
        float4 v[15];
#pragma unroll
        for (int r = 0; r < 3; ++r) {
            const int iy = min(max(oy * 2 - 1 + r, 0), H - 1);
#pragma unroll
            for (int q = 0; q < 5; ++q) {
                const int ix = min(max(ox * 2 - 1 + q, 0), W - 1);
                v[r * 5 + q] = ld4(x, ((size_t)(n * H + iy) * W + ix) * C4 + c);
            }
        }
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            if (ox + h >= Wo) break;
            // (the second output's window is columns 2 .. 4; when its right column is out of range the clamp makes it a copy of
            // an in-range column of the same window)
            float4 m = v[2 * h];
#pragma unroll
            for (int r = 0; r < 3; ++r)
#pragma unroll
                for (int q = 0; q < 3; ++q) {
                    const float4 u = v[r * 5 + 2 * h + q];
                    m.x = fmaxf(m.x, u.x); m.y = fmaxf(m.y, u.y); m.z = fmaxf(m.z, u.z); m.w = fmaxf(m.w, u.w);
                }
            st4(y, ((size_t)(n * Ho + oy) * Wo + ox + h) * C4 + c, m);
            if constexpr (std::is_same<T, float>::value) {
                if (yq) st4_pairs(yq, (size_t)(n * Ho + oy) * Wo + ox + h, C4 * 4, c, m, qs, qs2);
            }
            am = fmaxf(fmaxf(am, fmaxf(fabsf(m.x), fabsf(m.y))), fmaxf(fabsf(m.z), fabsf(m.w)));
        }
    }
    if (amax) radet_amax_publish(am, amax);
}

// _a variants (this one and radet_upsample_add_a / _bwd_a, radet_relu_bwd_a below): the same kernels, which also raise the
// output tensor's amax slot (4 bytes, device) to the largest magnitude they store -- the scale source of the fp16 hi / lo
// conv arithmetic (common.h "h2").  The slot is NOT reset here.
extern "C" int radet_maxpool3x3s2_a(const float* x, float* y, int B, int H, int W, int C, void* y_amax, void* stream) {
    if (C % 4) return RADET_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * Ho * ((Wo + 1) / 2) * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4, Ho, Wo,
                       (unsigned*)y_amax);
    return radet_check_launch();
}
extern "C" int radet_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, void* stream) {
    return radet_maxpool3x3s2_a(x, y, B, H, W, C, nullptr, stream);
}
// ... and the output once more as fp16 plane pairs yq (rows [2][C], C % 32 == 0) scaled by the power of two of x's amax slot
// (x_amax; its bits are stored to yq_amax)
extern "C" int radet_maxpool3x3s2_q(const float* x, float* y, int B, int H, int W, int C, void* y_amax, void* yq, void* yq_amax,
                                    const void* x_amax, void* stream) {
    if (C % 32 || yq == nullptr || yq_amax == nullptr || x_amax == nullptr) return RADET_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * Ho * ((Wo + 1) / 2) * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(maxpool_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, H, W, C / 4, Ho, Wo,
                       (unsigned*)y_amax, (_Float16*)yq, (unsigned*)yq_amax, (const unsigned*)x_amax);
    return radet_check_launch();
}

extern "C" int radet_maxpool3x3s2_h(const void* x, void* y, int B, int H, int W, int C, void* stream) {
    if (C % 4) return RADET_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * Ho * ((Wo + 1) / 2) * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(maxpool_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)x, (__bf16*)y,
                       B, H, W, C / 4, Ho, Wo);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ trainable stem (frozen_stages = -1)
// Backward of max-pool 3x3 / 2 (pad 1) fused with the stem's ReLU: ds[n, y, x, c] = [s > 0] * the sum of dpool over the (at
// most four) windows that contain (y, x) and whose FIRST maximum in row-major scan order over their in-range positions is
// (y, x) -- the element torch.nn.MaxPool2d routes the gradient to (resnet.py:570,630 + autograd).  Gather form: no atomics, one
// writer per element.
__global__ void maxpool_bwd_relu_kernel(const float* __restrict__ s, const float* __restrict__ dpool, float* __restrict__ ds,
                                        int B, int H, int W, int C4, int Ho, int Wo) {
    const size_t total = (size_t)B * H * W * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int x = (int)(p % W);
        p /= W;
        const int y = (int)(p % H);
        const int n = (int)(p / H);
        const float4 v = ld4(s, i);
        float4 g = make_float4(0.f, 0.f, 0.f, 0.f);
        for (int oy = y >> 1; oy <= min((y + 1) >> 1, Ho - 1); ++oy)
            for (int ox = x >> 1; ox <= min((x + 1) >> 1, Wo - 1); ++ox) {
                bool seen = false;
                bool fx = true, fy = true, fz = true, fw = true;          // (y, x) is the first maximum of this window, per channel
                for (int r = 0; r < 3; ++r) {
                    const int iy = oy * 2 - 1 + r;
                    if (iy < 0 || iy >= H) continue;
                    for (int q = 0; q < 3; ++q) {
                        const int ix = ox * 2 - 1 + q;
                        if (ix < 0 || ix >= W) continue;
                        if (iy == y && ix == x) { seen = true; continue; }
                        const float4 u = ld4(s, ((size_t)(n * H + iy) * W + ix) * C4 + c);
                        if (!seen) { fx &= u.x < v.x; fy &= u.y < v.y; fz &= u.z < v.z; fw &= u.w < v.w; }   // earlier: strictly less
                        else       { fx &= u.x <= v.x; fy &= u.y <= v.y; fz &= u.z <= v.z; fw &= u.w <= v.w; }
                    }
                }
                const float4 d = ld4(dpool, ((size_t)(n * Ho + oy) * Wo + ox) * C4 + c);
                if (fx) g.x += d.x;
                if (fy) g.y += d.y;
                if (fz) g.z += d.z;
                if (fw) g.w += d.w;
            }
        st4(ds, i, make_float4(v.x > 0.f ? g.x : 0.f, v.y > 0.f ? g.y : 0.f, v.z > 0.f ? g.z : 0.f, v.w > 0.f ? g.w : 0.f));
    }
}

extern "C" int radet_maxpool3x3s2_bwd_relu(const float* s, const float* dpool, float* ds, int B, int H, int W, int C, void* stream) {
    if (C % 4 || B <= 0 || H <= 0 || W <= 0) return RADET_ERR_ARG;
    const int Ho = (H - 1) / 2 + 1, Wo = (W - 1) / 2 + 1;
    const size_t total = (size_t)B * H * W * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(maxpool_bwd_relu_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, s, dpool, ds, B, H, W, C / 4, Ho, Wo);
    return radet_check_launch();
}

// Weight gradient of the 7x7 / 2 stem conv (3 -> 64) straight from the NCHW image: slabs[split][o][tap][c] = sum over the
// split's output pixels of ds[pixel][o] * img[n, c, 2 y - 3 + ky, 2 x - 3 + kx] (zero outside the image), and the column sums of
// ds (the folded BN's shift gradient) -- the layout `radet_unfold_grads` reduces.  5.8 GFLOP per step at bs 4 and off every
// BOP config's path (they freeze the stem), so a plain VALU kernel: a workgroup walks its pixels 16 at a time (ds rows and
// the 147-element input patches in LDS), thread (o, kg) keeps 40 of the 147 products of output channel o in registers and
// reads the patch as broadcast float4s.  One writer per slab element, pixels in ascending order: deterministic.
#define SW_PIX 16
#define SW_K 160            // 147 patch elements padded to four groups of 40
__global__ __launch_bounds__(256) void stem_wgrad_kernel(const float* __restrict__ img, const float* __restrict__ ds,
                                                         float* __restrict__ slabs, float* __restrict__ dbias_partials, int B, int H,
                                                         int W, int Ho, int Wo, int chunks_per_split) {
    __shared__ __attribute__((aligned(16))) float sds[SW_PIX][64];
    __shared__ __attribute__((aligned(16))) float sp[SW_PIX][SW_K];
    const int t = threadIdx.x, o = t & 63, kg = t >> 6;
    const long npix = (long)B * Ho * Wo;
    const long p0 = (long)blockIdx.x * chunks_per_split * SW_PIX;
    const long p1 = p0 + (long)chunks_per_split * SW_PIX < npix ? p0 + (long)chunks_per_split * SW_PIX : npix;
    float acc[40];
#pragma unroll
    for (int j = 0; j < 40; ++j) acc[j] = 0.f;
    float bsum = 0.f;
    for (long base = p0; base < p1; base += SW_PIX) {
        {   // ds tile: 16 pixels x 64 channels, one float4 per thread
            const int px = t >> 4, c4 = t & 15;
            const long pix = base + px;
            const float4 v = pix < p1 ? ld4(ds, (size_t)pix * 16 + c4) : make_float4(0.f, 0.f, 0.f, 0.f);
            *reinterpret_cast<float4*>(&sds[px][4 * c4]) = v;
        }
        for (int e = t; e < SW_PIX * SW_K; e += 256) {      // input patches: element k = tap * 3 + c
            const int px = e / SW_K, k = e - px * SW_K;
            const long pix = base + px;
            float v = 0.f;
            if (k < 147 && pix < p1) {
                const int tap = k / 3, c = k - 3 * tap, ky = tap / 7, kx = tap - 7 * ky;
                const int x = (int)(pix % Wo), y = (int)((pix / Wo) % Ho), n = (int)(pix / ((long)Wo * Ho));
                const int iy = 2 * y - 3 + ky, ix = 2 * x - 3 + kx;
                if (iy >= 0 && iy < H && ix >= 0 && ix < W) v = img[((size_t)(n * 3 + c) * H + iy) * W + ix];
            }
            sp[px][k] = v;
        }
        __syncthreads();
#pragma unroll 4
        for (int px = 0; px < SW_PIX; ++px) {
            const float d = sds[px][o];
            if (kg == 0) bsum += d;
            const float4* pp = reinterpret_cast<const float4*>(&sp[px][40 * kg]);
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const float4 u = pp[j];
                acc[4 * j] += d * u.x; acc[4 * j + 1] += d * u.y; acc[4 * j + 2] += d * u.z; acc[4 * j + 3] += d * u.w;
            }
        }
        __syncthreads();
    }
    float* out = slabs + ((size_t)blockIdx.x * 64 + o) * 147;
#pragma unroll
    for (int j = 0; j < 40; ++j)
        if (40 * kg + j < 147) out[40 * kg + j] = acc[j];
    if (kg == 0 && dbias_partials != nullptr) dbias_partials[(size_t)blockIdx.x * 64 + o] = bsum;
}

extern "C" int radet_stem_wgrad_splits(int B, int H, int W) {
    const int Ho = (H + 2 * 3 - 7) / 2 + 1, Wo = (W + 2 * 3 - 7) / 2 + 1;
    const long chunks = ((long)B * Ho * Wo + SW_PIX - 1) / SW_PIX;
    long S = (chunks + 31) / 32;
    if (S < 1) S = 1;
    if (S > 1024) S = 1024;
    return (int)S;
}

extern "C" int radet_stem_wgrad(const float* img_nchw, const float* ds, float* slabs, float* dbias_partials, int B, int H, int W,
                                int S, void* stream) {
    if (B <= 0 || H <= 0 || W <= 0 || S < 1) return RADET_ERR_ARG;
    const int Ho = (H + 2 * 3 - 7) / 2 + 1, Wo = (W + 2 * 3 - 7) / 2 + 1;
    const long chunks = ((long)B * Ho * Wo + SW_PIX - 1) / SW_PIX;
    const int cps = (int)((chunks + S - 1) / S);
    hipLaunchKernelGGL(stem_wgrad_kernel, dim3(S), dim3(256), 0, (hipStream_t)stream, img_nchw, ds, slabs, dbias_partials, B, H, W,
                       Ho, Wo, cps);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ fold / unfold
// fold: for every conv in the table, s[o] = gamma*rsqrt(var+eps) (1 without BN):
//   wf[o][t][c] = s[o] * w[o][c][t] ; wft[c][t][o] = same value ; bias_f[o] = beta - mean*s | conv bias | 0
// One work item = (o-tile of 32, c-tile of 32 * (9 / KT) channels): the OIHW slab [32][c-tile * KT] (up to 288 floats per output
// channel) is read with contiguous runs, transposed through LDS and written as contiguous runs of both OHWI and [c][t][o].
// (Round 6: a 1 x 1 conv's item used to be 32 x 32 weights = 4 KiB per load -> barrier -> store -> store round, which left the
// pass bound by those round trips -- 236 us for 100 MB read + 200 MB written -- while it shared the device with the frozen part of
// the forward pass; skipping the fold altogether made the step 0.16 ms shorter.  Items of up to 32 x 288 whatever the taps.)
#define FOLD_TO 32
#define FOLD_TC 32
#define FOLD_UNR 8
// folded weight store: element `col` of row `row` (row length ld): fp32, bf16 (w16 = 1) or bf16 plane triple (w16 = 2:
// rows of ld / 32 groups [hi | mid | lo] x 32 channels, see common.h; ld % 32 == 0)
__device__ __forceinline__ void stw(float* p, size_t row_, int col, int ld, float v, int w16, float s = 1.f, float s2 = 2048.f) {
    const unsigned row = (unsigned)row_;  // (a conv's folded weights are far below 2^31 elements: 32-bit offsets)
    if (w16 == 3) {                       // fp16 plane pair (common.h "h2"), scaled by the conv's power of two
        unsigned h, l;
        radet_split2(v, 0.f, s, s2, h, l);
        unsigned short* q = reinterpret_cast<unsigned short*>(p) + (row * 2u * (unsigned)ld + (unsigned)radet_pair_off(col));
        q[0] = (unsigned short)(h & 0xFFFFu);
        q[32] = (unsigned short)(l & 0xFFFFu);
    } else if (w16 == 2) {
        const unsigned h = __float_as_uint(v) & 0xFFFF0000u;
        const float r = v - __uint_as_float(h);
        const unsigned m = __float_as_uint(r) & 0xFFFF0000u;
        const unsigned l = __float_as_uint(r - __uint_as_float(m));
        unsigned short* q = reinterpret_cast<unsigned short*>(p) + (row * 3u * (unsigned)ld + (unsigned)radet_plane_off(col));
        q[0] = (unsigned short)(h >> 16);
        q[32] = (unsigned short)(m >> 16);
        q[64] = (unsigned short)(l >> 16);
    } else if (w16) reinterpret_cast<__bf16*>(p)[row * (unsigned)ld + (unsigned)col] = (__bf16)v;
    else p[row * (unsigned)ld + (unsigned)col] = v;
}
__global__ __launch_bounds__(256) void fold_kernel(const RadetConvDesc* __restrict__ table) {
    const RadetConvDesc d = table[blockIdx.y];
    const int KT = d.kh * d.kw;
    __shared__ float tile[FOLD_TO][FOLD_TC * 9 + 1];
    __shared__ float ssc[FOLD_TO];
    const int tid = threadIdx.x;
    float wmax = 0.f;                       // largest |folded weight| this thread stores -> d.w_amax (fp32 / bf16 stores; the plane
                                            // pairs of w16 = 3 need the maximum BEFORE they are written: fold_amax_kernel)
    float ps = 1.f, ps2 = 2048.f;           // w16 = 3 / wfq: the power-of-two scale of this conv's plane pairs
    if (d.w16 == 3 || d.wfq) {
        const int e = radet_h2_exp(radet_amax_read(reinterpret_cast<const unsigned*>(d.w_amax)));
        ps = radet_pow2(e); ps2 = radet_pow2(e + 11);
    }
    if (KT > 9) {   // 7x7 stem: small, keep the simple element-wise path
        const size_t total = (size_t)d.cout * d.cin * KT;
        for (size_t i = (size_t)blockIdx.x * 256 + tid; i < total; i += (size_t)gridDim.x * 256) {
            const int c = (int)(i % d.cin);
            const size_t oc = i / d.cin;
            const int t = (int)(oc % KT);
            const int o = (int)(oc / KT);
            float s = 1.f;
            if (d.bn_gamma) s = d.bn_gamma[o] * (1.0f / sqrtf(d.bn_var[o] + d.eps));
            const float v = d.w[((size_t)o * d.cin + c) * KT + t] * s;
            wmax = fmaxf(wmax, fabsf(v));
            stw(d.wf, oc, c, d.cin, v, d.w16, ps, ps2);
            if (d.wfq) stw(d.wfq, oc, c, d.cin, v, 3, ps, ps2);
            if (d.wft) stw(d.wft, (size_t)c * KT + t, d.wft_off + o, d.wft_ld ? d.wft_ld : d.cout, v, d.w16, ps, ps2);
        }
    } else {
        const int TC = FOLD_TC * (9 / KT);                 // channels per item: TC * KT <= 288 floats per output channel
        const int tiles_o = (d.cout + FOLD_TO - 1) / FOLD_TO, tiles_c = (d.cin + TC - 1) / TC;
        const int ld_t = d.wft_ld ? d.wft_ld : d.cout;
        for (int item = blockIdx.x; item < tiles_o * tiles_c; item += gridDim.x) {
            const int o0 = (item / tiles_c) * FOLD_TO, c0 = (item % tiles_c) * TC;
            const int nc = min(TC, d.cin - c0), no = min(FOLD_TO, d.cout - o0);
            const int run = nc * KT;                      // contiguous floats per output channel
            __syncthreads();
            if (tid < FOLD_TO) {
                float s = 1.f;
                const int o = o0 + tid;
                if (d.bn_gamma && o < d.cout) s = d.bn_gamma[o] * (1.0f / sqrtf(d.bn_var[o] + d.eps));
                ssc[tid] = s;
            }
            // (index arithmetic: quotients by the item's run-time extents through a float reciprocal -- exact for these ranges
            // (i < 9216, divisors <= 288: the error of (i + 0.5) * (1 / n) is far below the 0.5 / n margin) -- and 32-bit offsets:
            // seven integer divisions and the 64-bit address products per weight were most of this kernel's time)
            const float inv_run = 1.0f / (float)run, inv_nc = 1.0f / (float)nc, inv_no = 1.0f / (float)no, inv_kt = 1.0f / (float)KT;
            auto fdiv = [](int i, float inv) { return (int)(((float)i + 0.5f) * inv); };
            const float* const wsrc = d.w + ((size_t)o0 * d.cin + c0) * KT;          // row oo of the slab: + oo * cin * KT
            const unsigned row_in = (unsigned)d.cin * (unsigned)KT;
            for (int i0 = tid; i0 < no * run; i0 += FOLD_UNR * 256) {    // FOLD_UNR loads in flight per thread
                float v[FOLD_UNR];
#pragma unroll
                for (int u = 0; u < FOLD_UNR; ++u) {
                    const int i = min(i0 + 256 * u, no * run - 1);
                    const int oo = fdiv(i, inv_run), k = i - oo * run;      // k = c_local*KT + t
                    v[u] = wsrc[(unsigned)oo * row_in + (unsigned)k];
                }
#pragma unroll
                for (int u = 0; u < FOLD_UNR; ++u) {
                    const int i = i0 + 256 * u;
                    if (i < no * run) {
                        const int oo = fdiv(i, inv_run), k = i - oo * run;
                        tile[oo][k] = v[u];
                    }
                }
            }
            __syncthreads();
            // OHWI: wf[o][t][c0 + cl]  (runs of nc floats)
            for (int i = tid; i < no * run; i += 256) {
                const int oo = fdiv(i, inv_run), r = i - oo * run;
                const int t = fdiv(r, inv_nc), cl = r - t * nc;
                const float v = tile[oo][cl * KT + t] * ssc[oo];
                wmax = fmaxf(wmax, fabsf(v));
                stw(d.wf, (size_t)(o0 + oo) * KT + t, c0 + cl, d.cin, v, d.w16, ps, ps2);
                if (d.wfq) stw(d.wfq, (size_t)(o0 + oo) * KT + t, c0 + cl, d.cin, v, 3, ps, ps2);
            }
            // [c][t][o0 + oo]  (runs of no floats)
            if (d.wft) {
                for (int i = tid; i < no * run; i += 256) {
                    const int r = fdiv(i, inv_no), oo = i - r * no;      // r = cl*KT + t
                    const int cl = fdiv(r, inv_kt), t = r - cl * KT;
                    stw(d.wft, (size_t)(c0 + cl) * KT + t, d.wft_off + o0 + oo, ld_t, tile[oo][r] * ssc[oo], d.w16, ps, ps2);
                }
            }
        }
    }
    if (d.w_amax && d.w16 != 3 && !d.wfq) radet_amax_publish(wmax, reinterpret_cast<unsigned*>(d.w_amax));     // (uniform branch)
    if (blockIdx.x == 0 && d.bias_f) {
        float bmax = 0.f;
        for (int o = tid; o < d.cout; o += 256) {
            float b = 0.f;
            if (d.bn_gamma) {
                const float s = d.bn_gamma[o] * (1.0f / sqrtf(d.bn_var[o] + d.eps));
                b = d.bn_beta[o] - d.bn_mean[o] * s;
            } else if (d.bias) b = d.bias[o];
            d.bias_f[o] = b;
            bmax = fmaxf(bmax, fabsf(b));
        }
        if (d.bias_amax) radet_amax_publish(bmax, reinterpret_cast<unsigned*>(d.bias_amax));        // (all threads reach this)
    }
}

// amax slots of the folded weights (RadetConvDesc.w_amax, optional): largest |s[o] w[o][c][t]| of each conv -- the scale of
// the fp16 hi / lo arithmetic's weight operand (common.h "h2"; wf and wft hold the same values).  Zeroed, then raised by
// atomicMax (order independent): by fold_kernel as it stores fp32 / bf16 weights, by the pass below for the convs whose
// weights are stored as plane pairs (w16 = 3: the scale has to be known before the first store).
__global__ void fold_amax_zero_kernel(const RadetConvDesc* __restrict__ table, int nconv) {      // one wave per conv
    const RadetConvDesc& d = table[blockIdx.x];
    if (d.w_amax) reinterpret_cast<unsigned*>(d.w_amax)[threadIdx.x * RADET_AMAX_STRIDE] = 0u;
    if (d.w_l1) reinterpret_cast<unsigned*>(d.w_l1)[threadIdx.x * RADET_AMAX_STRIDE] = 0u;
    if (d.bias_amax) reinterpret_cast<unsigned*>(d.bias_amax)[threadIdx.x * RADET_AMAX_STRIDE] = 0u;
    if (d.w_l1t) reinterpret_cast<unsigned*>(d.w_l1t)[threadIdx.x * RADET_AMAX_STRIDE] = 0u;
}
__global__ __launch_bounds__(256) void fold_amax_kernel(const RadetConvDesc* __restrict__ table) {
    const RadetConvDesc d = table[blockIdx.y];
    // (the other convs' w_amax slots are raised by fold_kernel itself).  w_l1 (optional): the largest L1 norm of a folded
    // output channel, max_o sum_{c,t} |wf[o][t][c]| -- |conv output| <= amax(x) * that: the bound an epilogue that writes
    // fp16 plane pairs scales them with (conv_common.h).  One wave sums a whole channel in a fixed order.
    const int K = d.cin * d.kh * d.kw;                         // OIHW: K contiguous weights per output channel
    if (d.w_l1t) {
        // largest L1 norm of an input channel, max_c sum_{o,t} |wf[o][t][c]|: bounds the dgrad's output (its GEMM sums over
        // o and t).  A workgroup owns 64 consecutive channels: thread (rg, cl) sums rows o = rg, rg + 4, ... of channel cl --
        // a wave reads 64 neighbouring channels of one OIHW row (coalesced) -- and the four row groups are added in a fixed order
        const int KT = d.kh * d.kw;
        __shared__ float part[4][64];
        const int cl = threadIdx.x & 63, rg = threadIdx.x >> 6;
        float l1t = 0.f;
        for (int c0 = blockIdx.x * 64; c0 < d.cin; c0 += gridDim.x * 64) {
            const int c = c0 + cl;
            float sc_ = 0.f;
            if (c < d.cin)
                for (int o = rg; o < d.cout; o += 4) {
                    float s = 1.f;
                    if (d.bn_gamma) s = d.bn_gamma[o] * (1.0f / sqrtf(d.bn_var[o] + d.eps));
                    const float* w = d.w + ((size_t)o * d.cin + c) * KT;
                    float so = 0.f;
                    for (int t = 0; t < KT; ++t) so += fabsf(w[t] * s);
                    sc_ += so;
                }
            __syncthreads();
            part[rg][cl] = sc_;
            __syncthreads();
            if (rg == 0) l1t = fmaxf(l1t, (part[0][cl] + part[1][cl]) + (part[2][cl] + part[3][cl]));
        }
        radet_amax_publish(l1t, reinterpret_cast<unsigned*>(d.w_l1t));
    }
    if (d.w_amax == nullptr || (d.w16 != 3 && !d.wfq && !d.w_l1)) return;
    // One WAVE per output channel (round 6; a workgroup per channel with two barriers each made cout / 32 dependent load ->
    // reduce -> barrier rounds per workgroup: 64 of them for a 2048-channel conv, the 102 us of this pass), its K weights in
    // a fixed order: lane l sums k = l, l + 64, ..., then the xor butterfly -- the same sum in every run.
    float m = 0.f, l1max = 0.f;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int o = blockIdx.x * 4 + wave; o < d.cout; o += gridDim.x * 4) {
        float s = 1.f;
        if (d.bn_gamma) s = d.bn_gamma[o] * (1.0f / sqrtf(d.bn_var[o] + d.eps));
        const float* w = d.w + (size_t)o * K;
        float mo = 0.f, so = 0.f;
        for (int k0 = lane; k0 < K; k0 += 64 * 8) {              // 8 loads in flight per lane
            float v[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) v[u] = w[min(k0 + 64 * u, K - 1)];
#pragma unroll
            for (int u = 0; u < 8; ++u)
                if (k0 + 64 * u < K) {
                    const float a = fabsf(v[u] * s);            // (the product fold_kernel stores)
                    mo = fmaxf(mo, a);
                    so += a;
                }
        }
        m = fmaxf(m, mo);
        if (d.w_l1) l1max = fmaxf(l1max, wave_sum(so));           // (uniform)
    }
    if (d.w16 == 3 || d.wfq) radet_amax_publish(m, reinterpret_cast<unsigned*>(d.w_amax));
    if (d.w_l1) radet_amax_publish(l1max, reinterpret_cast<unsigned*>(d.w_l1));
}

extern "C" int radet_fold_weights(const RadetConvDesc* table_dev, int nconv, void* stream) {
    if (nconv <= 0) return RADET_OK;
    hipLaunchKernelGGL(fold_amax_zero_kernel, dim3(nconv), dim3(RADET_AMAX_WORDS), 0, (hipStream_t)stream, table_dev, nconv);
    hipLaunchKernelGGL(fold_amax_kernel, dim3(64, nconv), dim3(256), 0, (hipStream_t)stream, table_dev);
    hipLaunchKernelGGL(fold_kernel, dim3(128, nconv), dim3(256), 0, (hipStream_t)stream, table_dev);
    return radet_check_launch();
}

// unfold: one block per (conv, o).  dwf_sum[o][t][c] = sum over wgrad splits (fixed order); then
//   dw[o][c][t] = s[o]*dwf_sum ; ds[o] = <dwf_sum[o], w[o]> ; db[o] = sum of bias partials
//   BN: dgamma = rstd*(ds - db*mean) ; dbeta = db.  plain bias: dbias = db.
// The [t][c] row is staged in LDS so that slab reads, weight reads and OIHW writes are all contiguous.
__global__ __launch_bounds__(256) void unfold_kernel(const RadetConvDesc* __restrict__ table) {
    const RadetConvDesc d = table[blockIdx.y];
    const int o = blockIdx.x;
    if (o >= d.cout || d.dw == nullptr) return;
    const int KT = d.kh * d.kw;
    const int K = d.cin * KT;
    const size_t slab = (size_t)d.cout * K;
    float s = 1.f, rstd = 0.f;
    if (d.bn_gamma) {
        rstd = 1.0f / sqrtf(d.bn_var[o] + d.eps);
        s = d.bn_gamma[o] * rstd;
    }
    __shared__ float row[2304 + 8];
    float dot = 0.f;
    for (int k0 = 0; k0 < K; k0 += 2304) {   // chunk = whole taps x channels span of <= 2304 (KT*256)
        const int kn = min(2304, K - k0);
        // k0 is a multiple of KT*... only when K > 2304; handle generally by index math below
        __syncthreads();
        for (int i = threadIdx.x; i < kn; i += 256) {        // i + k0 = t*cin + c  (OHWI inner index)
            // slabs summed in split order (as before: same bits), but four loads in flight per thread: with the split
            // count a run-time value the plain loop is load -> wait -> add, one HBM round trip per slab
            const float* p = d.dwf_slabs + (size_t)o * K + k0 + i;
            float g = 0.f;
            int sp = 0;
            for (; sp + 4 <= d.nsplit; sp += 4) {
                const float v0 = p[(size_t)sp * slab], v1 = p[(size_t)(sp + 1) * slab];
                const float v2 = p[(size_t)(sp + 2) * slab], v3 = p[(size_t)(sp + 3) * slab];
                g = (((g + v0) + v1) + v2) + v3;
            }
            for (; sp < d.nsplit; ++sp) g += p[(size_t)sp * slab];
            row[i] = g;
        }
        __syncthreads();
        if (K <= 2304) {
            // whole row resident: emit OIHW order j = c*KT + t contiguously
            for (int j0 = threadIdx.x; j0 < K; j0 += 4 * 256) {       // 4 weight loads in flight per thread
                float wv[4];
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 256 * u;
                    wv[u] = d.w[(size_t)o * K + (j < K ? j : threadIdx.x)];
                }
#pragma unroll
                for (int u = 0; u < 4; ++u) {
                    const int j = j0 + 256 * u;
                    if (j < K) {
                        const int c = j / KT, t = j - c * KT;
                        const float g = row[t * d.cin + c];
                        dot += g * wv[u];
                        d.dw[(size_t)o * K + j] = g * s;
                    }
                }
            }
        } else {
            for (int i = threadIdx.x; i < kn; i += 256) {
                const int q = k0 + i;
                const int t = q / d.cin, c = q - t * d.cin;
                const size_t wi = ((size_t)o * d.cin + c) * KT + t;
                dot += row[i] * d.w[wi];
                d.dw[wi] = row[i] * s;
            }
        }
    }
    __shared__ float red[4];
    dot = wave_sum(dot);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = dot;
    __syncthreads();
    if (threadIdx.x == 0) {
        const float ds = (red[0] + red[1]) + (red[2] + red[3]);
        float db = 0.f;
        if (d.dbias_partials)
            for (int sp = 0; sp < d.nsplit; ++sp) db += d.dbias_partials[sp * d.cout + o];
        if (d.bn_gamma) {
            if (d.dgamma) d.dgamma[o] = rstd * (ds - db * d.bn_mean[o]);
            if (d.dbeta) d.dbeta[o] = db;
        } else if (d.dbias) d.dbias[o] = db;
    }
}

extern "C" int radet_unfold_grads(const RadetConvDesc* table_dev, int nconv, int max_cout, void* stream) {
    if (nconv <= 0) return RADET_OK;
    hipLaunchKernelGGL(unfold_kernel, dim3(max_cout, nconv), dim3(256), 0, (hipStream_t)stream, table_dev);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ GroupNorm
// Work unit = chunk of up to GN_CH pixels of one (level, image).  C = 256 channels, 32 groups of 8.
// A block (256 threads) covers 4 pixel rows x 64 float4 columns per step.
#define GN_CH 64
#define GN_UNR 4      // pixel rows whose loads are in flight together per thread

struct GnChunk { int seg, n, first_pix, npix, row0, chunk_in_img, nchunks_img, part_base; };

__device__ __forceinline__ GnChunk gn_decode(const RadetSegs& segs, int B, int bid) {
    GnChunk c;
    int base = 0, pbase = 0;
    c.seg = 0; c.n = 0; c.first_pix = 0; c.npix = 0; c.row0 = 0; c.chunk_in_img = 0; c.nchunks_img = 1; c.part_base = 0;
    for (int l = 0; l < segs.nseg; ++l) {
        const int hw = segs.s[l].Ho * segs.s[l].Wo;
        const int nch = (hw + GN_CH - 1) / GN_CH;
        if (bid < base + B * nch) {
            const int local = bid - base;
            c.seg = l;
            c.n = local / nch;
            c.chunk_in_img = local - c.n * nch;
            c.nchunks_img = nch;
            c.first_pix = c.chunk_in_img * GN_CH;
            c.npix = min(GN_CH, hw - c.first_pix);
            c.row0 = segs.s[l].row_begin + c.n * hw + c.first_pix;
            c.part_base = pbase + c.n * nch;
            return c;
        }
        base += B * nch;
        pbase += B * nch;
    }
    return c;
}

static int gn_total_chunks(const RadetSegs& segs, int B) {
    int t = 0;
    for (int l = 0; l < segs.nseg; ++l) t += B * ((segs.s[l].Ho * segs.s[l].Wo + GN_CH - 1) / GN_CH);
    return t;
}

// partial[chunk][32 groups][2] = (sum, sumsq) over the chunk's pixels x 8 channels
// one or two independent tensors of the same geometry per launch (blockIdx.y): cls / reg tower of one layer
template <class T>
struct GnFwdSet {
    const T* z;
    float* partial;
    const float *gamma, *beta;
    T* y;             // may be null when yp is given
    float* stats;
    __bf16* yp;       // optional second output: y as bf16 plane triples (rows [3][256], common.h) for the conv GEMMs
    // fp16 hi / lo arithmetic (common.h "h2"): yp holds fp16 plane PAIRS (rows [2][256]) scaled by the power of two of a BOUND
    // on |y| that every block derives from gamma / beta alone (|zhat| <= sqrt(n - 1) for the n values of a group), written to
    // yq_amax before any consumer runs; y_amax (fp32 output y) / zhat_amax are raised to the largest |y| / |zhat| seen
    // (zeroed by the statistics kernel; zhat_amax bounds the backward pass's dz)
    unsigned* yq_amax;
    unsigned* y_amax;
    unsigned* zhat_amax;
};
template <class T>
struct GnFwdArgs {
    GnFwdSet<T> p[2];
};

template <class T>
__global__ __launch_bounds__(256) void gn_stats_kernel(const GnFwdArgs<T> args, const RadetSegs segs, int B) {
    const T* __restrict__ z = args.p[blockIdx.y].z;
    float* __restrict__ partial = args.p[blockIdx.y].partial;
    const GnChunk c = gn_decode(segs, B, blockIdx.x);
    const int tid = threadIdx.x;
    if (blockIdx.x == 0 && tid < RADET_AMAX_WORDS) {   // (the apply kernel, which raises them, starts after this kernel has finished)
        if (args.p[blockIdx.y].y_amax) args.p[blockIdx.y].y_amax[tid * RADET_AMAX_STRIDE] = 0u;
        if (args.p[blockIdx.y].zhat_amax) args.p[blockIdx.y].zhat_amax[tid * RADET_AMAX_STRIDE] = 0u;
    }
    const int col = tid & 63, prow = tid >> 6;
    float s = 0.f, ss = 0.f;
    // 4 pixel rows per iteration, their loads issued together: left to the compiler the loop ran one 16-byte load
    // at a time per lane (load -> vmcnt(0) -> add), far too little in flight for an HBM-bound kernel
    for (int p0 = prow; p0 < c.npix; p0 += GN_UNR * 4) {
        float4 v[GN_UNR];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int p = p0 + 4 * u;
            v[u] = ld4(z, (size_t)(c.row0 + (p < c.npix ? p : prow)) * 64 + col);
        }
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            if (p0 + 4 * u < c.npix) {
                s += (v[u].x + v[u].y) + (v[u].z + v[u].w);
                ss += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
            }
        }
    }
    // the two float4 columns of a group are neighbouring lanes
    s += __shfl_xor(s, 1, 64);
    ss += __shfl_xor(ss, 1, 64);
    __shared__ float red[4][32][2];
    if ((col & 1) == 0) { red[prow][col >> 1][0] = s; red[prow][col >> 1][1] = ss; }
    __syncthreads();
    if (tid < 64) {
        const int g = tid >> 1, k = tid & 1;
        partial[((size_t)blockIdx.x * 32 + g) * 2 + k] = (red[0][g][k] + red[1][g][k]) + (red[2][g][k] + red[3][g][k]);
    }
}

// Per-(image, group) totals of the chunk partials: 256 threads = 8 interleaved parts x 32 groups, fp64, combined in a
// fixed order (deterministic).  A single-thread-per-group loop here was a 75-step dependent load chain in front of
// every apply block (37 us of a 45 us kernel).
__device__ __forceinline__ void gn_reduce_partials(const float* __restrict__ part, int part_base, int nchunks,
                                                   double* out_a, double* out_b) {
    __shared__ double pa[8][32], pb[8][32];
    const int tid = threadIdx.x, g = tid & 31, q = tid >> 5;
    double a = 0.0, b = 0.0;
    for (int k = q; k < nchunks; k += 8) {
        a += (double)part[((size_t)(part_base + k) * 32 + g) * 2 + 0];
        b += (double)part[((size_t)(part_base + k) * 32 + g) * 2 + 1];
    }
    pa[q][g] = a;
    pb[q][g] = b;
    __syncthreads();
    if (tid < 32) {
        double sa = 0.0, sb = 0.0;
#pragma unroll
        for (int i = 0; i < 8; ++i) { sa += pa[i][tid]; sb += pb[i][tid]; }
        *out_a = sa;
        *out_b = sb;
    }
}

// y = relu?((z - mean) * rstd * gamma + beta); the chunk-0 block of each image also publishes (mean, rstd)
template <class T>
__global__ __launch_bounds__(256) void gn_apply_kernel(const GnFwdArgs<T> args, const RadetSegs segs, int B, float eps,
                                                       int relu) {
    const T* __restrict__ z = args.p[blockIdx.y].z;
    const float* __restrict__ partial = args.p[blockIdx.y].partial;
    const float* __restrict__ gamma = args.p[blockIdx.y].gamma;
    const float* __restrict__ beta = args.p[blockIdx.y].beta;
    T* __restrict__ y = args.p[blockIdx.y].y;
    __bf16* __restrict__ yp = args.p[blockIdx.y].yp;
    float* __restrict__ stats = args.p[blockIdx.y].stats;
    unsigned* const yq_amax = args.p[blockIdx.y].yq_amax;
    unsigned* const y_amax = args.p[blockIdx.y].y_amax;
    unsigned* const zhat_amax = args.p[blockIdx.y].zhat_amax;
    const GnChunk c = gn_decode(segs, B, blockIdx.x);
    const int tid = threadIdx.x;
    __shared__ float sm[32], sr[32];
    float qs = 1.f, qs2 = 2048.f;                  // plane pairs: 2^e, 2^(e + 11) of the bound on |y|
    if (yq_amax) {
        __shared__ float bnd[8];
        const float g1 = wave_max(fabsf(gamma[tid])), b1 = wave_max(fabsf(beta[tid]));
        if ((tid & 63) == 0) { bnd[tid >> 6] = g1; bnd[4 + (tid >> 6)] = b1; }
        __syncthreads();
        int nmax = 1;
        for (int l = 0; l < segs.nseg; ++l) nmax = max(nmax, segs.s[l].Ho * segs.s[l].Wo * 8);
        const float gmax = fmaxf(fmaxf(bnd[0], bnd[1]), fmaxf(bnd[2], bnd[3]));
        const float bmax = fmaxf(fmaxf(bnd[4], bnd[5]), fmaxf(bnd[6], bnd[7]));
        const unsigned bits = __float_as_uint(gmax * sqrtf((float)(nmax - 1)) + bmax);
        if (blockIdx.x == 0 && tid == 0) radet_amax_store(yq_amax, bits);
        const int e = radet_h2_exp(bits);
        qs = radet_pow2(e); qs2 = radet_pow2(e + 11);
    }
    float ymax = 0.f, zmax = 0.f;
    double s = 0.0, ss = 0.0;
    gn_reduce_partials(partial, c.part_base, c.nchunks_img, &s, &ss);
    if (tid < 32) {
        const double cnt = (double)segs.s[c.seg].Ho * segs.s[c.seg].Wo * 8.0;
        const double mean = s / cnt;
        double var = ss / cnt - mean * mean;
        if (var < 0.0) var = 0.0;
        const float rstd = (float)(1.0 / sqrt(var + (double)eps));
        sm[tid] = (float)mean;
        sr[tid] = rstd;
    }
    __syncthreads();
    const int col = tid & 63, prow = tid >> 6;
    const int g = col >> 1;
    const float mean = sm[g], rstd = sr[g];
    const float4 gm = *reinterpret_cast<const float4*>(gamma + col * 4);
    const float4 bt = *reinterpret_cast<const float4*>(beta + col * 4);
    for (int p0 = prow; p0 < c.npix; p0 += GN_UNR * 4) {
        float4 vv[GN_UNR];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int p = p0 + 4 * u;
            vv[u] = ld4(z, (size_t)(c.row0 + (p < c.npix ? p : prow)) * 64 + col);
        }
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int p = p0 + 4 * u;
            if (p >= c.npix) continue;
            const float4 v = vv[u];
            float4 r;
            r.x = (v.x - mean) * rstd * gm.x + bt.x;
            r.y = (v.y - mean) * rstd * gm.y + bt.y;
            r.z = (v.z - mean) * rstd * gm.z + bt.z;
            r.w = (v.w - mean) * rstd * gm.w + bt.w;
            if (relu) { r.x = fmaxf(r.x, 0.f); r.y = fmaxf(r.y, 0.f); r.z = fmaxf(r.z, 0.f); r.w = fmaxf(r.w, 0.f); }
            if (y) st4(y, (size_t)(c.row0 + p) * 64 + col, r);
            if (yp) {
                if (yq_amax) st4_pairs(reinterpret_cast<_Float16*>(yp), (size_t)(c.row0 + p), 256, col, r, qs, qs2);
                else st4_planes(yp, (size_t)(c.row0 + p), 256, col, r);
            }
            ymax = fmaxf(fmaxf(ymax, fmaxf(fabsf(r.x), fabsf(r.y))), fmaxf(fabsf(r.z), fabsf(r.w)));
            zmax = fmaxf(fmaxf(zmax, fmaxf(fabsf((v.x - mean) * rstd), fabsf((v.y - mean) * rstd))),
                         fmaxf(fabsf((v.z - mean) * rstd), fabsf((v.w - mean) * rstd)));
        }
    }
    if (y_amax) radet_amax_publish(ymax, y_amax);
    if (zhat_amax) radet_amax_publish(zmax, zhat_amax);
    if (c.chunk_in_img == 0 && tid < 32) {
        // stats layout: [(seg, n)][32][2]; (seg, n) linear id = sum_{l<seg} B + n
        int lin = c.n;
        for (int l = 0; l < c.seg; ++l) lin += B;
        stats[((size_t)lin * 32 + tid) * 2 + 0] = sm[tid];
        stats[((size_t)lin * 32 + tid) * 2 + 1] = sr[tid];
    }
}

template <class T>
static int gn_relu_fwd_impl(const T* z, const float* gamma, const float* beta, T* y, float* stats,
                            float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc,
                            int nseg, void* stream, const GnFwdSet<T>* second = nullptr, __bf16* yp = nullptr,
                            unsigned* yq_amax = nullptr, unsigned* y_amax = nullptr, unsigned* zhat_amax = nullptr) {
    if (C != 256 || groups != 32 || (y == nullptr && yp == nullptr)) return RADET_ERR_ARG;
    RadetSegs segs;
    if (nseg < 1 || nseg > RADET_MAX_SEG) return RADET_ERR_ARG;
    segs.nseg = nseg;
    for (int l = 0; l < nseg; ++l) {
        const int* d = seg_desc + 6 * l;
        segs.s[l].Hi = d[0]; segs.s[l].Wi = d[1]; segs.s[l].Ho = d[2]; segs.s[l].Wo = d[3];
        segs.s[l].in_row_off = d[4]; segs.s[l].row_begin = d[5]; segs.s[l].row_end = d[5] + B * d[2] * d[3];
    }
    const int chunks = gn_total_chunks(segs, B);
    hipStream_t st = (hipStream_t)stream;
    GnFwdArgs<T> args;
    args.p[0] = GnFwdSet<T>{z, partial_ws, gamma, beta, y, stats, yp, yq_amax, y_amax, zhat_amax};
    args.p[1] = second ? *second : args.p[0];
    const int sets = second ? 2 : 1;
    hipLaunchKernelGGL(gn_stats_kernel<T>, dim3(chunks, sets), dim3(256), 0, st, args, segs, B);
    hipLaunchKernelGGL(gn_apply_kernel<T>, dim3(chunks, sets), dim3(256), 0, st, args, segs, B, eps, relu);
    return radet_check_launch();
}

// GroupNorm + ReLU of two tensors of the same geometry in one pair of launches (cls_convs[i] / reg_convs[i]: two
// separate launches on two streams cost a fork and a join, 10-17 us of idle device each, around 23 us of kernels)
extern "C" int radet_gn_relu_fwd_pair(const float* z0, const float* gamma0, const float* beta0, float* y0, float* stats0,
                                      float* partial_ws0, const float* z1, const float* gamma1, const float* beta1,
                                      float* y1, float* stats1, float* partial_ws1, int B, int C, int groups, float eps,
                                      int relu, const int* seg_desc, int nseg, void* stream) {
    const GnFwdSet<float> second{z1, partial_ws1, gamma1, beta1, y1, stats1, nullptr};
    return gn_relu_fwd_impl<float>(z0, gamma0, beta0, y0, stats0, partial_ws0, B, C, groups, eps, relu, seg_desc, nseg,
                                   stream, &second);
}

extern "C" int radet_gn_relu_fwd_pair_h(const void* z0, const float* gamma0, const float* beta0, void* y0, float* stats0,
                                        float* partial_ws0, const void* z1, const float* gamma1, const float* beta1,
                                        void* y1, float* stats1, float* partial_ws1, int B, int C, int groups, float eps,
                                        int relu, const int* seg_desc, int nseg, void* stream) {
    const GnFwdSet<__bf16> second{(const __bf16*)z1, partial_ws1, gamma1, beta1, (__bf16*)y1, stats1, nullptr};
    return gn_relu_fwd_impl<__bf16>((const __bf16*)z0, gamma0, beta0, (__bf16*)y0, stats0, partial_ws0, B, C, groups, eps,
                                    relu, seg_desc, nseg, stream, &second);
}

extern "C" int radet_gn_relu_fwd(const float* z, const float* gamma, const float* beta, float* y, float* stats,
                                 float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc,
                                 int nseg, void* stream) {
    return gn_relu_fwd_impl<float>(z, gamma, beta, y, stats, partial_ws, B, C, groups, eps, relu, seg_desc, nseg, stream);
}

extern "C" int radet_gn_relu_fwd_h(const void* z, const float* gamma, const float* beta, void* y, float* stats,
                                   float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc,
                                   int nseg, void* stream) {
    return gn_relu_fwd_impl<__bf16>((const __bf16*)z, gamma, beta, (__bf16*)y, stats, partial_ws, B, C, groups, eps, relu,
                                    seg_desc, nseg, stream);
}

// fp32 in, outputs as fp32 (y, may be NULL) and / or bf16 plane triples (yp, may be NULL): the tower activations are read
// by conv GEMMs only, which take plane operands (radet_conv2d_igemm tile_override 0x2000000)
extern "C" int radet_gn_relu_fwd_p(const float* z, const float* gamma, const float* beta, float* y, void* yp, float* stats,
                                   float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc,
                                   int nseg, void* stream) {
    return gn_relu_fwd_impl<float>(z, gamma, beta, y, stats, partial_ws, B, C, groups, eps, relu, seg_desc, nseg, stream,
                                   nullptr, (__bf16*)yp);
}
extern "C" int radet_gn_relu_fwd_pair_p(const float* z0, const float* gamma0, const float* beta0, float* y0, void* yp0,
                                        float* stats0, float* partial_ws0, const float* z1, const float* gamma1,
                                        const float* beta1, float* y1, void* yp1, float* stats1, float* partial_ws1, int B,
                                        int C, int groups, float eps, int relu, const int* seg_desc, int nseg, void* stream) {
    if (y1 == nullptr && yp1 == nullptr) return RADET_ERR_ARG;
    const GnFwdSet<float> second{z1, partial_ws1, gamma1, beta1, y1, stats1, (__bf16*)yp1};
    return gn_relu_fwd_impl<float>(z0, gamma0, beta0, y0, stats0, partial_ws0, B, C, groups, eps, relu, seg_desc, nseg,
                                   stream, &second, (__bf16*)yp0);
}

// fp16 hi / lo arithmetic (common.h "h2"): fp32 in; outputs y (fp32, may be NULL; y_amax, may be NULL, is raised to its largest
// magnitude) and / or yq (fp16 plane pairs, rows [2][256], may be NULL; yq_amax receives the bound on |y| whose power of two
// scales them); zhat_amax (may be NULL) is raised to the largest |zhat| -- radet_gn_relu_bwd_q's bound on dz uses it.
extern "C" int radet_gn_relu_fwd_q(const float* z, const float* gamma, const float* beta, float* y, void* yq, float* stats,
                                   float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc,
                                   int nseg, void* stream, void* y_amax, void* yq_amax, void* zhat_amax) {
    if (yq != nullptr && yq_amax == nullptr) return RADET_ERR_ARG;
    return gn_relu_fwd_impl<float>(z, gamma, beta, y, stats, partial_ws, B, C, groups, eps, relu, seg_desc, nseg, stream,
                                   nullptr, (__bf16*)yq, yq ? (unsigned*)yq_amax : nullptr, (unsigned*)y_amax,
                                   (unsigned*)zhat_amax);
}
extern "C" int radet_gn_relu_fwd_pair_q(const float* z0, const float* gamma0, const float* beta0, float* y0, void* yq0,
                                        float* stats0, float* partial_ws0, void* y_amax0, void* yq_amax0, void* zhat_amax0,
                                        const float* z1, const float* gamma1, const float* beta1, float* y1, void* yq1,
                                        float* stats1, float* partial_ws1, void* y_amax1, void* yq_amax1, void* zhat_amax1,
                                        int B, int C, int groups, float eps, int relu, const int* seg_desc, int nseg,
                                        void* stream) {
    if ((y1 == nullptr && yq1 == nullptr) || (yq0 != nullptr && yq_amax0 == nullptr) || (yq1 != nullptr && yq_amax1 == nullptr))
        return RADET_ERR_ARG;
    const GnFwdSet<float> second{z1, partial_ws1, gamma1, beta1, y1, stats1, (__bf16*)yq1, yq1 ? (unsigned*)yq_amax1 : nullptr,
                                 (unsigned*)y_amax1, (unsigned*)zhat_amax1};
    return gn_relu_fwd_impl<float>(z0, gamma0, beta0, y0, stats0, partial_ws0, B, C, groups, eps, relu, seg_desc, nseg,
                                   stream, &second, (__bf16*)yq0, yq0 ? (unsigned*)yq_amax0 : nullptr, (unsigned*)y_amax0,
                                   (unsigned*)zhat_amax0);
}

extern "C" int radet_gn_workspace_floats(int B, const int* seg_desc, int nseg) {
    int t = 0;
    for (int l = 0; l < nseg; ++l) t += B * ((seg_desc[6 * l + 2] * seg_desc[6 * l + 3] + GN_CH - 1) / GN_CH);
    return t * (64 + 512);  // group partials (fwd/bwd) + per-channel partials (bwd)
}

// backward pass 1: g = dy * [y > 0] (y recomputed);  per chunk: group sums (sum g*gamma, sum g*gamma*xhat),
// channel sums (sum g*xhat -> dgamma, sum g -> dbeta)
template <class T>
__global__ __launch_bounds__(256) void gn_bwd_stats_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float* __restrict__ gpart,
                                                           float* __restrict__ cpart, const RadetSegs segs, int B,
                                                           int relu) {
    const GnChunk c = gn_decode(segs, B, blockIdx.x);
    const int tid = threadIdx.x;
    const int col = tid & 63, prow = tid >> 6;
    const int g = col >> 1;
    int lin = c.n;
    for (int l = 0; l < c.seg; ++l) lin += B;
    const float mean = stats[((size_t)lin * 32 + g) * 2 + 0];
    const float rstd = stats[((size_t)lin * 32 + g) * 2 + 1];
    const float4 gm = *reinterpret_cast<const float4*>(gamma + col * 4);
    const float4 bt = *reinterpret_cast<const float4*>(beta + col * 4);
    float s1 = 0.f, s2 = 0.f;
    float4 cg = make_float4(0.f, 0.f, 0.f, 0.f), cb = make_float4(0.f, 0.f, 0.f, 0.f);
    for (int p0 = prow; p0 < c.npix; p0 += GN_UNR * 4) {
        float4 vv[GN_UNR], dd[GN_UNR];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int p = p0 + 4 * u;
            const size_t o = (size_t)(c.row0 + (p < c.npix ? p : prow)) * 64 + col;
            vv[u] = ld4(z, o);
            dd[u] = ld4(dy, o);
        }
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            if (p0 + 4 * u >= c.npix) continue;
            const float4 v = vv[u];
            float4 d = dd[u];
            float4 xh;
            xh.x = (v.x - mean) * rstd; xh.y = (v.y - mean) * rstd; xh.z = (v.z - mean) * rstd; xh.w = (v.w - mean) * rstd;
            if (relu) {
                if (xh.x * gm.x + bt.x <= 0.f) d.x = 0.f;
                if (xh.y * gm.y + bt.y <= 0.f) d.y = 0.f;
                if (xh.z * gm.z + bt.z <= 0.f) d.z = 0.f;
                if (xh.w * gm.w + bt.w <= 0.f) d.w = 0.f;
            }
            cg.x += d.x * xh.x; cg.y += d.y * xh.y; cg.z += d.z * xh.z; cg.w += d.w * xh.w;
            cb.x += d.x; cb.y += d.y; cb.z += d.z; cb.w += d.w;
            const float a0 = d.x * gm.x, a1 = d.y * gm.y, a2 = d.z * gm.z, a3 = d.w * gm.w;
            s1 += (a0 + a1) + (a2 + a3);
            s2 += (a0 * xh.x + a1 * xh.y) + (a2 * xh.z + a3 * xh.w);
        }
    }
    s1 += __shfl_xor(s1, 1, 64);
    s2 += __shfl_xor(s2, 1, 64);
    __shared__ float red[4][32][2];
    __shared__ float4 cred[4][64][2];
    if ((col & 1) == 0) { red[prow][g][0] = s1; red[prow][g][1] = s2; }
    cred[prow][col][0] = cg;
    cred[prow][col][1] = cb;
    __syncthreads();
    if (tid < 64) {
        const int gg = tid >> 1, k = tid & 1;
        gpart[((size_t)blockIdx.x * 32 + gg) * 2 + k] = (red[0][gg][k] + red[1][gg][k]) + (red[2][gg][k] + red[3][gg][k]);
    }
    if (tid < 128) {
        const int cc = tid & 63, k = tid >> 6;
        float4 a = cred[0][cc][k], b = cred[1][cc][k], c2 = cred[2][cc][k], d2 = cred[3][cc][k];
        float4 r;
        r.x = (a.x + b.x) + (c2.x + d2.x); r.y = (a.y + b.y) + (c2.y + d2.y);
        r.z = (a.z + b.z) + (c2.z + d2.z); r.w = (a.w + b.w) + (c2.w + d2.w);
        // cpart[chunk][2][256]
        *reinterpret_cast<float4*>(cpart + ((size_t)blockIdx.x * 2 + k) * 256 + cc * 4) = r;
    }
}

// final per-channel reduce over chunks: dgamma[c], dbeta[c].  32 workgroups x (16 row groups x 16 columns) over the
// [nchunks][2][256] partials, fixed summation order (8 workgroups x 4 row groups made a 100-step chain per thread on
// the critical path of every tower layer's backward).
// Runs as the LAST 32 workgroups of the apply launch (it only needs the stats pass's partials): as a launch of its own it sat
// on the tower's backward chain between the GroupNorm and the next dgrad GEMM, 8 us + a kernel boundary per layer.
__device__ __forceinline__ void gn_bwd_param_body(const float* __restrict__ cpart, int nchunks, float* __restrict__ dgamma,
                                                  float* __restrict__ dbeta, int bx) {
    const int lc = threadIdx.x & 15, rg = threadIdx.x >> 4;
    const int col = bx * 16 + lc;   // 0..511 = k*256 + c
    float a = 0.f;
    // same order of additions as a plain loop, with 8 loads in flight (a plain loop is one round trip per chunk row: 26
    // dependent loads per thread on the tower's backward chain)
    for (int k0 = rg; k0 < nchunks; k0 += 16 * 8) {
        float v[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int k = k0 + 16 * u;
            v[u] = cpart[(size_t)(k < nchunks ? k : rg) * 512 + col];
        }
#pragma unroll
        for (int u = 0; u < 8; ++u)
            if (k0 + 16 * u < nchunks) a += v[u];
    }
    __shared__ float red[16][16];
    red[rg][lc] = a;
    __syncthreads();
    if (rg == 0) {
        float v = 0.f;
#pragma unroll
        for (int i = 0; i < 16; ++i) v += red[i][lc];
        if (col < 256) dgamma[col] = v; else dbeta[col - 256] = v;
    }
}

// backward pass 2: dz = rstd * (g*gamma - m1 - xhat*m2), m1/m2 = group means of g*gamma, g*gamma*xhat
template <class T>
__global__ __launch_bounds__(256) void gn_bwd_apply_kernel(const T* __restrict__ dy, const T* __restrict__ z,
                                                           const float* __restrict__ stats,
                                                           const float* __restrict__ gamma,
                                                           const float* __restrict__ beta,
                                                           const float* __restrict__ gpart, T* __restrict__ dz,
                                                           __bf16* __restrict__ dzp, const RadetSegs segs, int B,
                                                           int relu, const unsigned* dy_amax, const unsigned* zhat_amax,
                                                           unsigned* dzq_amax, const float* __restrict__ cpart, int nchunks,
                                                           float* __restrict__ dgamma, float* __restrict__ dbeta) {
    if ((int)blockIdx.x >= nchunks) {                            // (uniform) the parameter-gradient reduction: workgroups nchunks .. + 31
        gn_bwd_param_body(cpart, nchunks, dgamma, dbeta, (int)blockIdx.x - nchunks);
        return;
    }
    const GnChunk c = gn_decode(segs, B, blockIdx.x);
    const int tid = threadIdx.x;
    __shared__ float m1s[32], m2s[32];
    // fp16 plane pairs (dzq_amax given; common.h "h2"): every block derives the same bound on |dz| from quantities that are
    // complete before this kernel starts -- |dz| = rstd |g gamma - m1 - zhat m2| <= rstd_max gamma_max dy_amax (2 + zhat_max),
    // because |m1| <= gamma_max dy_amax and |m2| <= gamma_max dy_amax mean|zhat| <= gamma_max dy_amax -- and scales by its
    // power of two; block 0 publishes the bound for the GEMMs that read the pairs
    float qs = 1.f, qs2 = 2048.f;
    if (dzq_amax) {
        __shared__ float bnd[8];
        int ngrp = 0;
        for (int l = 0; l < segs.nseg; ++l) ngrp += B * 32;
        float rmax = 0.f;
        for (int i = tid; i < ngrp; i += 256) rmax = fmaxf(rmax, stats[(size_t)i * 2 + 1]);
        rmax = wave_max(rmax);
        const float g1 = wave_max(fabsf(gamma[tid]));
        if ((tid & 63) == 0) { bnd[tid >> 6] = rmax; bnd[4 + (tid >> 6)] = g1; }
        __syncthreads();
        const float rstd_max = fmaxf(fmaxf(bnd[0], bnd[1]), fmaxf(bnd[2], bnd[3]));
        const float gmax = fmaxf(fmaxf(bnd[4], bnd[5]), fmaxf(bnd[6], bnd[7]));
        int nmax = 1;
        for (int l = 0; l < segs.nseg; ++l) nmax = max(nmax, segs.s[l].Ho * segs.s[l].Wo * 8);
        const float zh = zhat_amax ? __uint_as_float(radet_amax_read(zhat_amax)) : sqrtf((float)(nmax - 1));
        const unsigned bits = __float_as_uint(rstd_max * gmax * __uint_as_float(radet_amax_read(dy_amax)) * (2.0f + zh));
        if (blockIdx.x == 0 && tid == 0) radet_amax_store(dzq_amax, bits);
        const int e = radet_h2_exp(bits);
        qs = radet_pow2(e); qs2 = radet_pow2(e + 11);
    }
    double a = 0.0, b = 0.0;
    gn_reduce_partials(gpart, c.part_base, c.nchunks_img, &a, &b);
    if (tid < 32) {
        const double cnt = (double)segs.s[c.seg].Ho * segs.s[c.seg].Wo * 8.0;
        m1s[tid] = (float)(a / cnt);
        m2s[tid] = (float)(b / cnt);
    }
    __syncthreads();
    const int col = tid & 63, prow = tid >> 6;
    const int g = col >> 1;
    int lin = c.n;
    for (int l = 0; l < c.seg; ++l) lin += B;
    const float mean = stats[((size_t)lin * 32 + g) * 2 + 0];
    const float rstd = stats[((size_t)lin * 32 + g) * 2 + 1];
    const float m1 = m1s[g], m2 = m2s[g];
    const float4 gm = *reinterpret_cast<const float4*>(gamma + col * 4);
    const float4 bt = *reinterpret_cast<const float4*>(beta + col * 4);
    for (int p0 = prow; p0 < c.npix; p0 += GN_UNR * 4) {
        float4 vv[GN_UNR], dd[GN_UNR];
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int p = p0 + 4 * u;
            const size_t o = (size_t)(c.row0 + (p < c.npix ? p : prow)) * 64 + col;
            vv[u] = ld4(z, o);
            dd[u] = ld4(dy, o);
        }
#pragma unroll
        for (int u = 0; u < GN_UNR; ++u) {
            const int p = p0 + 4 * u;
            if (p >= c.npix) continue;
            const float4 v = vv[u];
            float4 d = dd[u];
            float4 xh;
            xh.x = (v.x - mean) * rstd; xh.y = (v.y - mean) * rstd; xh.z = (v.z - mean) * rstd; xh.w = (v.w - mean) * rstd;
            if (relu) {
                if (xh.x * gm.x + bt.x <= 0.f) d.x = 0.f;
                if (xh.y * gm.y + bt.y <= 0.f) d.y = 0.f;
                if (xh.z * gm.z + bt.z <= 0.f) d.z = 0.f;
                if (xh.w * gm.w + bt.w <= 0.f) d.w = 0.f;
            }
            float4 r;
            r.x = rstd * (d.x * gm.x - m1 - xh.x * m2);
            r.y = rstd * (d.y * gm.y - m1 - xh.y * m2);
            r.z = rstd * (d.z * gm.z - m1 - xh.z * m2);
            r.w = rstd * (d.w * gm.w - m1 - xh.w * m2);
            if (dz) st4(dz, (size_t)(c.row0 + p) * 64 + col, r);
            if (dzp) {                                                         // planes for the dgrad / wgrad GEMMs
                if (dzq_amax) st4_pairs(reinterpret_cast<_Float16*>(dzp), (size_t)(c.row0 + p), 256, col, r, qs, qs2);
                else st4_planes(dzp, (size_t)(c.row0 + p), 256, col, r);
            }
        }
    }
}

template <class T>
static int gn_relu_bwd_impl(const T* dy, const T* z, const float* stats, const float* gamma,
                            const float* beta, T* dz, float* dgamma, float* dbeta, float* partial_ws, int B,
                            int C, int groups, int relu, const int* seg_desc, int nseg, void* stream,
                            __bf16* dzp = nullptr, const unsigned* dy_amax = nullptr, const unsigned* zhat_amax = nullptr,
                            unsigned* dzq_amax = nullptr) {
    if (C != 256 || groups != 32 || (dz == nullptr && dzp == nullptr)) return RADET_ERR_ARG;
    RadetSegs segs;
    if (nseg < 1 || nseg > RADET_MAX_SEG) return RADET_ERR_ARG;
    segs.nseg = nseg;
    for (int l = 0; l < nseg; ++l) {
        const int* d = seg_desc + 6 * l;
        segs.s[l].Hi = d[0]; segs.s[l].Wi = d[1]; segs.s[l].Ho = d[2]; segs.s[l].Wo = d[3];
        segs.s[l].in_row_off = d[4]; segs.s[l].row_begin = d[5]; segs.s[l].row_end = d[5] + B * d[2] * d[3];
    }
    const int chunks = gn_total_chunks(segs, B);
    float* gpart = partial_ws;
    float* cpart = partial_ws + (size_t)chunks * 64;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(gn_bwd_stats_kernel<T>, dim3(chunks), dim3(256), 0, st, dy, z, stats, gamma, beta, gpart, cpart,
                       segs, B, relu);
    hipLaunchKernelGGL(gn_bwd_apply_kernel<T>, dim3(chunks + 32), dim3(256), 0, st, dy, z, stats, gamma, beta, gpart, dz, dzp,
                       segs, B, relu, dy_amax, zhat_amax, dzq_amax, cpart, chunks, dgamma, dbeta);
    return radet_check_launch();
}

extern "C" int radet_gn_relu_bwd(const float* dy, const float* z, const float* stats, const float* gamma,
                                 const float* beta, float* dz, float* dgamma, float* dbeta, float* partial_ws, int B,
                                 int C, int groups, int relu, const int* seg_desc, int nseg, void* stream) {
    return gn_relu_bwd_impl<float>(dy, z, stats, gamma, beta, dz, dgamma, dbeta, partial_ws, B, C, groups, relu, seg_desc,
                                   nseg, stream);
}

// fp32 in, dz as fp32 (may be NULL) and / or bf16 plane triples (dzp, may be NULL)
extern "C" int radet_gn_relu_bwd_p(const float* dy, const float* z, const float* stats, const float* gamma,
                                   const float* beta, float* dz, void* dzp, float* dgamma, float* dbeta, float* partial_ws,
                                   int B, int C, int groups, int relu, const int* seg_desc, int nseg, void* stream) {
    return gn_relu_bwd_impl<float>(dy, z, stats, gamma, beta, dz, dgamma, dbeta, partial_ws, B, C, groups, relu, seg_desc,
                                   nseg, stream, (__bf16*)dzp);
}

// fp16 hi / lo arithmetic: dz as fp32 (may be NULL) and / or fp16 plane pairs dzq (rows [2][256]) scaled by the power of two of
// a bound on |dz| built from dy_amax (the amax slot of dy, required with dzq), zhat_amax (from radet_gn_relu_fwd_q; NULL: the
// hard bound sqrt(n - 1)) and the forward statistics; the bound is written to dzq_amax
extern "C" int radet_gn_relu_bwd_q(const float* dy, const float* z, const float* stats, const float* gamma,
                                   const float* beta, float* dz, void* dzq, float* dgamma, float* dbeta, float* partial_ws,
                                   int B, int C, int groups, int relu, const int* seg_desc, int nseg, void* stream,
                                   const void* dy_amax, const void* zhat_amax, void* dzq_amax) {
    if (dzq != nullptr && (dy_amax == nullptr || dzq_amax == nullptr)) return RADET_ERR_ARG;
    return gn_relu_bwd_impl<float>(dy, z, stats, gamma, beta, dz, dgamma, dbeta, partial_ws, B, C, groups, relu, seg_desc,
                                   nseg, stream, (__bf16*)dzq, (const unsigned*)dy_amax, (const unsigned*)zhat_amax,
                                   dzq ? (unsigned*)dzq_amax : nullptr);
}

extern "C" int radet_gn_relu_bwd_h(const void* dy, const void* z, const float* stats, const float* gamma,
                                   const float* beta, void* dz, float* dgamma, float* dbeta, float* partial_ws, int B,
                                   int C, int groups, int relu, const int* seg_desc, int nseg, void* stream) {
    return gn_relu_bwd_impl<__bf16>((const __bf16*)dy, (const __bf16*)z, stats, gamma, beta, (__bf16*)dz, dgamma, dbeta,
                                    partial_ws, B, C, groups, relu, seg_desc, nseg, stream);
}

// ------------------------------------------------------------------------------------------ upsample-add
__device__ __forceinline__ int nearest_src(int dst, float scale, int in_size) {
    const int s = (int)floorf((float)dst * scale);
    return s < in_size - 1 ? s : in_size - 1;
}

// dst[n, oy, ox, :] += src[n, nearest(oy), nearest(ox), :]      (F.interpolate(mode='nearest', size=...))
template <class T>
__global__ void upsample_add_kernel(T* __restrict__ dst, const T* __restrict__ src, int B, int Ho, int Wo,
                                    int Hi, int Wi, int C4, float sy, float sx, unsigned* amax = nullptr) {
    float am = 0.f;
    const size_t total = (size_t)B * Ho * Wo * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ox = (int)(p % Wo);
        p /= Wo;
        const int oy = (int)(p % Ho);
        const int n = (int)(p / Ho);
        const int iy = nearest_src(oy, sy, Hi), ix = nearest_src(ox, sx, Wi);
        const float4 s = ld4(src, ((size_t)(n * Hi + iy) * Wi + ix) * C4 + c);
        float4 d = ld4(dst, i);
        d.x += s.x; d.y += s.y; d.z += s.z; d.w += s.w;
        st4(dst, i, d);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(d.x), fabsf(d.y))), fmaxf(fabsf(d.z), fabsf(d.w)));
    }
    if (amax) radet_amax_publish(am, amax);
}

// dsrc[n, iy, ix, :] += sum over dst pixels that read (iy, ix)
template <class T>
__global__ void upsample_add_bwd_kernel(T* __restrict__ dsrc, const T* __restrict__ ddst, int B, int Ho,
                                        int Wo, int Hi, int Wi, int C4, float sy, float sx, unsigned* amax = nullptr) {
    float am = 0.f;
    const size_t total = (size_t)B * Hi * Wi * C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C4);
        size_t p = i / C4;
        const int ix = (int)(p % Wi);
        p /= Wi;
        const int iy = (int)(p % Hi);
        const int n = (int)(p / Hi);
        const int oy_lo = max(0, (int)((float)iy / sy) - 2), oy_hi = min(Ho - 1, (int)((float)(iy + 1) / sy) + 2);
        const int ox_lo = max(0, (int)((float)ix / sx) - 2), ox_hi = min(Wo - 1, (int)((float)(ix + 1) / sx) + 2);
        float4 a = ld4(dsrc, i);
        for (int oy = oy_lo; oy <= oy_hi; ++oy) {
            if (nearest_src(oy, sy, Hi) != iy) continue;
            for (int ox = ox_lo; ox <= ox_hi; ++ox) {
                if (nearest_src(ox, sx, Wi) != ix) continue;
                const float4 d = ld4(ddst, ((size_t)(n * Ho + oy) * Wo + ox) * C4 + c);
                a.x += d.x; a.y += d.y; a.z += d.z; a.w += d.w;
            }
        }
        st4(dsrc, i, a);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(a.x), fabsf(a.y))), fmaxf(fabsf(a.z), fabsf(a.w)));
    }
    if (amax) radet_amax_publish(am, amax);
}

extern "C" int radet_upsample_add_a(float* dst, const float* src, int B, int Ho, int Wo, int Hi, int Wi, int C,
                                    void* dst_amax, void* stream) {
    if (C % 4) return RADET_ERR_ARG;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(upsample_add_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dst, src, B, Ho, Wo, Hi,
                       Wi, C / 4, (float)Hi / (float)Ho, (float)Wi / (float)Wo, (unsigned*)dst_amax);
    return radet_check_launch();
}
extern "C" int radet_upsample_add(float* dst, const float* src, int B, int Ho, int Wo, int Hi, int Wi, int C,
                                  void* stream) {
    return radet_upsample_add_a(dst, src, B, Ho, Wo, Hi, Wi, C, nullptr, stream);
}

extern "C" int radet_upsample_add_h(void* dst, const void* src, int B, int Ho, int Wo, int Hi, int Wi, int C, void* stream) {
    if (C % 4) return RADET_ERR_ARG;
    const size_t total = (size_t)B * Ho * Wo * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(upsample_add_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (__bf16*)dst,
                       (const __bf16*)src, B, Ho, Wo, Hi, Wi, C / 4, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return radet_check_launch();
}

extern "C" int radet_upsample_add_bwd_a(float* dsrc, const float* ddst, int B, int Ho, int Wo, int Hi, int Wi, int C,
                                        void* dsrc_amax, void* stream) {
    if (C % 4) return RADET_ERR_ARG;
    const size_t total = (size_t)B * Hi * Wi * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(upsample_add_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dsrc, ddst, B, Ho,
                       Wo, Hi, Wi, C / 4, (float)Hi / (float)Ho, (float)Wi / (float)Wo, (unsigned*)dsrc_amax);
    return radet_check_launch();
}
extern "C" int radet_upsample_add_bwd(float* dsrc, const float* ddst, int B, int Ho, int Wo, int Hi, int Wi, int C,
                                      void* stream) {
    return radet_upsample_add_bwd_a(dsrc, ddst, B, Ho, Wo, Hi, Wi, C, nullptr, stream);
}

extern "C" int radet_upsample_add_bwd_h(void* dsrc, const void* ddst, int B, int Ho, int Wo, int Hi, int Wi, int C,
                                        void* stream) {
    if (C % 4) return RADET_ERR_ARG;
    const size_t total = (size_t)B * Hi * Wi * (C / 4);
    const int blocks = (int)((total + 255) / 256 < 4096 ? (total + 255) / 256 : 4096);
    hipLaunchKernelGGL(upsample_add_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (__bf16*)dsrc,
                       (const __bf16*)ddst, B, Ho, Wo, Hi, Wi, C / 4, (float)Hi / (float)Ho, (float)Wi / (float)Wo);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ relu backward
// dx = (dy (+ addend)) * [act > 0]
template <class T>
__global__ void relu_bwd_kernel(const T* __restrict__ dy, const T* __restrict__ addend,
                                const T* __restrict__ act, T* __restrict__ dx, size_t n4, unsigned* amax = nullptr) {
    float am = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        float4 v = ld4(dy, i);
        if (addend) { const float4 a = ld4(addend, i); v.x += a.x; v.y += a.y; v.z += a.z; v.w += a.w; }
        const float4 m = ld4(act, i);
        v.x = m.x > 0.f ? v.x : 0.f; v.y = m.y > 0.f ? v.y : 0.f; v.z = m.z > 0.f ? v.z : 0.f; v.w = m.w > 0.f ? v.w : 0.f;
        st4(dx, i, v);
        am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    if (amax) radet_amax_publish(am, amax);
}
extern "C" int radet_relu_bwd_a(const float* dy, const float* addend, const float* act, float* dx, size_t n, void* dx_amax,
                                void* stream) {
    if (n % 4) return RADET_ERR_ARG;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(relu_bwd_kernel<float>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, dy, addend, act, dx, n4,
                       (unsigned*)dx_amax);
    return radet_check_launch();
}
extern "C" int radet_relu_bwd(const float* dy, const float* addend, const float* act, float* dx, size_t n, void* stream) {
    return radet_relu_bwd_a(dy, addend, act, dx, n, nullptr, stream);
}
// largest magnitude of n floats (n % 4 == 0) -> raises *amax (for tensors whose producer does not track it; not reset here)
__global__ void absmax_kernel(const float* __restrict__ x, size_t n4, unsigned* amax) {
    float am = 0.f;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x) {
        const float4 v = reinterpret_cast<const float4*>(x)[i];
        am = fmaxf(fmaxf(am, fmaxf(fabsf(v.x), fabsf(v.y))), fmaxf(fabsf(v.z), fabsf(v.w)));
    }
    radet_amax_publish(am, amax);
}
extern "C" int radet_amax_slot_words(void) { return RADET_AMAX_WORDS * RADET_AMAX_STRIDE; }
extern "C" int radet_absmax(const float* x, size_t n, void* amax, void* stream) {
    if (n % 4 || amax == nullptr) return RADET_ERR_ARG;
    if (n == 0) return RADET_OK;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 2048 ? (n4 + 255) / 256 : 2048);
    hipLaunchKernelGGL(absmax_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, n4, (unsigned*)amax);
    return radet_check_launch();
}
extern "C" int radet_relu_bwd_h(const void* dy, const void* addend, const void* act, void* dx, size_t n, void* stream) {
    if (n % 4) return RADET_ERR_ARG;
    const size_t n4 = n / 4;
    const int blocks = (int)((n4 + 255) / 256 < 4096 ? (n4 + 255) / 256 : 4096);
    hipLaunchKernelGGL(relu_bwd_kernel<__bf16>, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)dy,
                       (const __bf16*)addend, (const __bf16*)act, (__bf16*)dx, n4);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ fp32 <-> bf16 rows
// dst[r][dst_off + c] = src[r][src_off + c] for c < ncols (row strides in elements); used at the fp32 boundaries of the
// bf16-storage mode (loss gradients -> bf16 dgrad / wgrad inputs, bf16 features -> fp32 module outputs)
template <class TS, class TD>
__global__ void convert_rows_kernel(const TS* __restrict__ src, TD* __restrict__ dst, size_t rows, int ncols, int src_ld,
                                    int src_off, int dst_ld, int dst_off) {
    const size_t total = rows * (size_t)ncols;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / ncols;
        const int c = (int)(i - r * ncols);
        dst[r * dst_ld + dst_off + c] = (TD)(float)src[r * src_ld + src_off + c];
    }
}
extern "C" int radet_convert_rows(const void* src, void* dst, size_t rows, int ncols, int src_ld, int src_off, int dst_ld,
                                  int dst_off, int to_bf16, void* stream) {
    if (rows == 0 || ncols <= 0) return RADET_OK;
    const size_t total = rows * (size_t)ncols;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    if (to_bf16)
        hipLaunchKernelGGL((convert_rows_kernel<float, __bf16>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, (__bf16*)dst, rows, ncols, src_ld, src_off, dst_ld, dst_off);
    else
        hipLaunchKernelGGL((convert_rows_kernel<__bf16, float>), dim3(blocks), dim3(256), 0, (hipStream_t)stream,
                           (const __bf16*)src, (float*)dst, rows, ncols, src_ld, src_off, dst_ld, dst_off);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ fp32 rows -> bf16 plane triples
// dst rows [3][C] bf16 (hi | mid | lo, common.h) from src rows of C fp32 (row stride src_ld floats); and back.
__global__ void split_planes_kernel(const float* __restrict__ src, __bf16* __restrict__ dst, size_t rows, int C4, int src_ld4) {
    const size_t total = rows * (size_t)C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C4;
        const int c = (int)(i - r * C4);
        st4_planes(dst, r, C4 * 4, c, reinterpret_cast<const float4*>(src)[r * src_ld4 + c]);
    }
}
__global__ void merge_planes_kernel(const __bf16* __restrict__ src, float* __restrict__ dst, size_t rows, int C4, int dst_ld4) {
    const size_t total = rows * (size_t)C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C4;
        const int c = (int)(i - r * C4);
        reinterpret_cast<float4*>(dst)[r * dst_ld4 + c] = ld4_planes(src, r, C4 * 4, c);
    }
}
extern "C" int radet_split_planes(const float* src, void* dst, size_t rows, int C, int src_ld, void* stream) {
    if (C <= 0 || (C & 31) || (src_ld & 3) || src_ld < C) return RADET_ERR_ARG;
    if (rows == 0) return RADET_OK;
    const size_t total = rows * (size_t)(C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (__bf16*)dst, rows, C / 4,
                       src_ld / 4);
    return radet_check_launch();
}
extern "C" int radet_merge_planes(const void* src, float* dst, size_t rows, int C, int dst_ld, void* stream) {
    if (C <= 0 || (C & 31) || (dst_ld & 3) || dst_ld < C) return RADET_ERR_ARG;
    if (rows == 0) return RADET_OK;
    const size_t total = rows * (size_t)(C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(merge_planes_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const __bf16*)src, dst, rows,
                       C / 4, dst_ld / 4);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ fp32 rows -> fp16 plane pairs
// dst rows [2][C] fp16 (32-channel groups [hi x 32 | lo x 32], common.h "h2") = src scaled by 2^radet_h2_exp(*src_amax)
// (src_amax: the amax slot of src -- raised by the kernels that wrote src, or any bound on its largest magnitude); the
// same bits are copied to dst_amax for the GEMMs that read the pairs.  merge: the inverse (tests / API boundary).
__global__ void split_pairs_kernel(const float* __restrict__ src, _Float16* __restrict__ dst, size_t rows, int C4, int src_ld4,
                                   const unsigned* __restrict__ src_amax, unsigned* __restrict__ dst_amax) {
    const unsigned bits = radet_amax_read(src_amax);
    const int e = radet_h2_exp(bits);
    const float s = radet_pow2(e), s2 = radet_pow2(e + 11);
    if (blockIdx.x == 0 && threadIdx.x == 0) radet_amax_store(dst_amax, bits);
    const size_t total = rows * (size_t)C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C4;
        const int c = (int)(i - r * C4);
        st4_pairs(dst, r, C4 * 4, c, reinterpret_cast<const float4*>(src)[r * src_ld4 + c], s, s2);
    }
}
__global__ void merge_pairs_kernel(const _Float16* __restrict__ src, float* __restrict__ dst, size_t rows, int C4, int dst_ld4,
                                   const unsigned* __restrict__ amax) {
    const float inv = radet_pow2(-radet_h2_exp(radet_amax_read(amax)));
    const size_t total = rows * (size_t)C4;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const size_t r = i / C4;
        const int c = (int)(i - r * C4);
        reinterpret_cast<float4*>(dst)[r * dst_ld4 + c] = ld4_pairs(src, r, C4 * 4, c, inv);
    }
}
extern "C" int radet_split_pairs(const float* src, void* dst, size_t rows, int C, int src_ld, const void* src_amax,
                                 void* dst_amax, void* stream) {
    if (C <= 0 || (C & 31) || (src_ld & 3) || src_ld < C || src_amax == nullptr || dst_amax == nullptr) return RADET_ERR_ARG;
    if (rows == 0) return RADET_OK;
    const size_t total = rows * (size_t)(C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(split_pairs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, src, (_Float16*)dst, rows, C / 4,
                       src_ld / 4, (const unsigned*)src_amax, (unsigned*)dst_amax);
    return radet_check_launch();
}
extern "C" int radet_merge_pairs(const void* src, float* dst, size_t rows, int C, int dst_ld, const void* amax, void* stream) {
    if (C <= 0 || (C & 31) || (dst_ld & 3) || dst_ld < C || amax == nullptr) return RADET_ERR_ARG;
    if (rows == 0) return RADET_OK;
    const size_t total = rows * (size_t)(C / 4);
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(merge_pairs_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, (const _Float16*)src, dst, rows,
                       C / 4, dst_ld / 4, (const unsigned*)amax);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ layout helpers
// NCHW <-> NHWC for the drop-in module API (the fast path never needs them except for the image,
// which the stem reads directly).
__global__ void nchw_to_nhwc_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int c = (int)(i % C);
        size_t p = i / C;
        const int hw = (int)(p % HW);
        const int n = (int)(p / HW);
        y[i] = x[((size_t)n * C + c) * HW + hw];
    }
}
__global__ void nhwc_to_nchw_kernel(const float* __restrict__ x, float* __restrict__ y, int B, int C, int HW) {
    const size_t total = (size_t)B * C * HW;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int hw = (int)(i % HW);
        size_t p = i / HW;
        const int c = (int)(p % C);
        const int n = (int)(p / C);
        y[i] = x[((size_t)n * HW + hw) * C + c];
    }
}
extern "C" int radet_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    const size_t total = (size_t)B * C * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nchw_to_nhwc_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, C, H * W);
    return radet_check_launch();
}
extern "C" int radet_nhwc_to_nchw(const float* x, float* y, int B, int C, int H, int W, void* stream) {
    const size_t total = (size_t)B * C * H * W;
    const int blocks = (int)((total + 255) / 256 < 8192 ? (total + 255) / 256 : 8192);
    hipLaunchKernelGGL(nhwc_to_nchw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, x, y, B, C, H * W);
    return radet_check_launch();
}
