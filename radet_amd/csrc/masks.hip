// Instance-mask path feeding the visibility-guided assigner, on the GPU: nearest resize -> flip -> pad of a stack
// of u8 bitmaps in ONE pass, plus the loader's per-mask normalisation (mask / mask.max()).
// Replaces radet/core/mask/structures.py:253-303 (BitmapMasks.rescale / resize / flip / pad, i.e. mmcv.imresize
// (cv2.INTER_NEAREST) / np.flip / np.pad per mask on the host) and radet/datasets/pipelines/loading.py:419-422.
// HBM-bound byte work: every output byte is written once (4 per thread, one 32-bit store when the row allows),
// every source byte is read at most ~once (rows are walked contiguously).
#include "common.h"
#include "radet_hip.h"

__global__ __launch_bounds__(256) void mask_max_kernel(const uint8_t* __restrict__ m, unsigned* __restrict__ mx, size_t hw) {
    const int g = blockIdx.y;
    const uint8_t* p = m + (size_t)g * hw;
    unsigned v = 0;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < hw; i += (size_t)gridDim.x * 256) v = max(v, (unsigned)p[i]);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = max(v, (unsigned)__shfl_xor((int)v, o, 64));
    if ((threadIdx.x & 63) == 0 && v) atomicMax(mx + g, v);
}

// cv2.resize(INTER_NEAREST): src index = min(floor(dst index * (1 / (dst / src))), src - 1), in double
__device__ __forceinline__ int nn_src(int d, double inv, int n) {
    const int s = (int)floor((double)d * inv);
    return s < n - 1 ? s : n - 1;
}

__global__ __launch_bounds__(256) void mask_transform_kernel(const uint8_t* __restrict__ src, uint8_t* __restrict__ dst,
                                                             const unsigned* __restrict__ norm_max, int Hs, int Ws, int Hr,
                                                             int Wr, int Hd, int Wd, double ify, double ifx, int flip,
                                                             int pad_val) {
    const int g = blockIdx.z, y = blockIdx.y;
    const int x0 = (blockIdx.x * 256 + threadIdx.x) * 4;
    if (x0 >= Wd) return;
    const uint8_t* sp = src + (size_t)g * Hs * Ws;
    const unsigned mx = norm_max ? norm_max[g] : 0u;
    unsigned out[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int x = x0 + j;
        unsigned v = (unsigned)pad_val & 0xFFu;
        if (y < Hr && x < Wr) {
            const int yr = (flip & 2) ? Hr - 1 - y : y;       // flip acts on the resized image
            const int xr = (flip & 1) ? Wr - 1 - x : x;
            v = sp[(size_t)nn_src(yr, ify, Hs) * Ws + nn_src(xr, ifx, Ws)];
            // (mask / mask.max()).astype(u8): 1 where the value equals the mask's maximum, else 0; an all-zero mask
            // is 0 / 0 = NaN -> 0 after the cast
            if (norm_max) v = (mx != 0u && v == mx) ? 1u : 0u;
        }
        out[j] = v;
    }
    uint8_t* dp = dst + ((size_t)g * Hd + y) * Wd + x0;
    if ((Wd & 3) == 0) {
        *reinterpret_cast<unsigned*>(dp) = out[0] | (out[1] << 8) | (out[2] << 16) | (out[3] << 24);
    } else {
#pragma unroll
        for (int j = 0; j < 4; ++j)
            if (x0 + j < Wd) dp[j] = (uint8_t)out[j];
    }
}

extern "C" int radet_mask_max(const uint8_t* masks, uint32_t* maxes, int G, size_t hw, void* stream) {
    if (G <= 0 || hw == 0) return RADET_ERR_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (hipMemsetAsync(maxes, 0, sizeof(uint32_t) * G, st) != hipSuccess) return RADET_ERR_LAUNCH;
    int bx = (int)((hw + 256 * 16 - 1) / (256 * 16));
    if (bx > 64) bx = 64;
    hipLaunchKernelGGL(mask_max_kernel, dim3(bx, G), dim3(256), 0, st, masks, maxes, hw);
    return radet_check_launch();
}

extern "C" int radet_mask_transform(const uint8_t* src, uint8_t* dst, const uint32_t* norm_max, int G, int Hs, int Ws,
                                    int Hr, int Wr, int Hd, int Wd, int flip, int pad_val, void* stream) {
    if (G <= 0 || Hs <= 0 || Ws <= 0 || Hr <= 0 || Wr <= 0 || Hd < Hr || Wd < Wr || flip < 0 || flip > 3 || Hd > 65535)
        return RADET_ERR_ARG;
    // OpenCV computes inv_scale = dsize / ssize and then 1. / inv_scale (not ssize / dsize)
    const double ify = 1.0 / ((double)Hr / (double)Hs), ifx = 1.0 / ((double)Wr / (double)Ws);
    hipLaunchKernelGGL(mask_transform_kernel, dim3((Wd + 1023) / 1024, Hd, G), dim3(256), 0, (hipStream_t)stream, src, dst,
                       norm_max, Hs, Ws, Hr, Wr, Hd, Wd, ify, ifx, flip, pad_val);
    return radet_check_launch();
}
