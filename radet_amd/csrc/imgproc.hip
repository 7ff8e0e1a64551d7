// Image pre/post-processing around the MBD / GDT box-to-distance transforms of the mask-free sampler
// (radet/ops/bbox2distance/bbox2distance_wrapper.py:80-93, 118-130, 170-181: cv2.resize, cv2.GaussianBlur, cv2.cvtColor,
// cv2.Sobel, cv2.addWeighted), batched over the box crops of an image: crops are packed back to back, a descriptor row
// per crop = {pixel offset, height, width}; grid = (pixel tiles, crops), one output pixel per thread.
//
// cv2 is neither in the reference tree nor in this image, so these kernels restate OpenCV's published algorithms
// (generic C++ paths): PARITY UNPINNED against cv2 itself; the arithmetic is pinned by oracle/imgproc.py (NumPy, same
// formulas) and cross-checked against scipy.ndimage there.  Chosen semantics:
//   * resize, 8-bit: INTER_LINEAR with half-pixel centres, 11-bit fixed-point coefficients (cvRound(f * 2048)), the
//     horizontal pass kept as integers, the vertical pass ((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2;
//   * resize, float / double: same coordinates, float coefficients, products summed in the element type;
//   * GaussianBlur 9x9, sigma 0 -> 0.3 * ((9 - 1) * 0.5 - 1) + 0.8 = 1.7, float kernel normalised in double,
//     BORDER_REFLECT_101, symmetric row / column passes in float, result rounded to nearest even;
//   * sobel edge map (GDT_box2distance.sobel_extract_edge): 3x3 Gaussian (1 2 1) / 4 in exact integers rounded half up,
//     RGB2GRAY = (4899 c0 + 9617 c1 + 1868 c2 + 8192) >> 14, 3x3 Sobel d/dx and d/dy (REFLECT_101) in float,
//     |0.5 gx + 0.5 gy| divided by the crop's maximum.
#include "common.h"
#include "../../include/radet_hip.h"

struct Crop { int off, h, w; };

__device__ __forceinline__ Crop load_crop(const int* desc, int n) {
    Crop c;
    c.off = desc[3 * n]; c.h = desc[3 * n + 1]; c.w = desc[3 * n + 2];
    return c;
}

__device__ __forceinline__ int reflect101(int p, int n) {      // cv::borderInterpolate(BORDER_REFLECT_101)
    if (n == 1) return 0;
    while (p < 0 || p >= n) p = p < 0 ? -p : 2 * n - 2 - p;
    return p;
}

// source index + fraction of destination index d (cv::resize, INTER_LINEAR): f = (d + 0.5) * scale - 0.5 in float
__device__ __forceinline__ void lin_coord(int d, double scale, int n, bool clamp_frac, int* s, float* f) {
    float fx = (float)(((double)d + 0.5) * scale - 0.5);
    int sx = (int)floorf(fx);
    fx -= (float)sx;
    if (clamp_frac) {                                           // x direction: coefficients are reset at the borders
        if (sx < 0) { fx = 0.f; sx = 0; }
        if (sx >= n - 1) { fx = 0.f; sx = n - 1; }
    }
    *s = sx;
    *f = fx;
}

__device__ __forceinline__ int clip_row(int y, int n) { return y < 0 ? 0 : (y < n ? y : n - 1); }

__device__ __forceinline__ int coef_fix(float c) {             // saturate_cast<short>(c * 2048): round to nearest even
    int v = __float2int_rn(c * 2048.f);
    return v > 32767 ? 32767 : (v < -32768 ? -32768 : v);
}

template <int C>
__global__ __launch_bounds__(256) void resize_u8_kernel(const uint8_t* __restrict__ src, const int* __restrict__ sdesc,
                                                        uint8_t* __restrict__ dst, const int* __restrict__ ddesc) {
    const Crop s = load_crop(sdesc, blockIdx.y), d = load_crop(ddesc, blockIdx.y);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.h * d.w) return;
    const int dy = p / d.w, dx = p - dy * d.w;
    const double scale_x = 1.0 / ((double)d.w / (double)s.w), scale_y = 1.0 / ((double)d.h / (double)s.h);
    int sx, sy;
    float fx, fy;
    lin_coord(dx, scale_x, s.w, true, &sx, &fx);
    lin_coord(dy, scale_y, s.h, false, &sy, &fy);
    const int a0 = coef_fix(1.f - fx), a1 = coef_fix(fx), b0 = coef_fix(1.f - fy), b1 = coef_fix(fy);
    const int y0 = clip_row(sy, s.h), y1 = clip_row(sy + 1, s.h);
    const int x1 = sx + 1 < s.w ? sx + 1 : sx;                  // a1 == 0 whenever sx is the last column
    const uint8_t* r0 = src + ((size_t)s.off + (size_t)y0 * s.w) * C;
    const uint8_t* r1 = src + ((size_t)s.off + (size_t)y1 * s.w) * C;
    uint8_t* o = dst + ((size_t)d.off + p) * C;
#pragma unroll
    for (int c = 0; c < C; ++c) {
        const int h0 = (int)r0[sx * C + c] * a0 + (int)r0[x1 * C + c] * a1;
        const int h1 = (int)r1[sx * C + c] * a0 + (int)r1[x1 * C + c] * a1;
        const int v = (((b0 * (h0 >> 4)) >> 16) + ((b1 * (h1 >> 4)) >> 16) + 2) >> 2;
        o[c] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

template <class T>
__global__ __launch_bounds__(256) void resize_f_kernel(const T* __restrict__ src, const int* __restrict__ sdesc,
                                                       T* __restrict__ dst, const int* __restrict__ ddesc) {
    const Crop s = load_crop(sdesc, blockIdx.y), d = load_crop(ddesc, blockIdx.y);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= d.h * d.w) return;
    const int dy = p / d.w, dx = p - dy * d.w;
    const double scale_x = 1.0 / ((double)d.w / (double)s.w), scale_y = 1.0 / ((double)d.h / (double)s.h);
    int sx, sy;
    float fx, fy;
    lin_coord(dx, scale_x, s.w, true, &sx, &fx);
    lin_coord(dy, scale_y, s.h, false, &sy, &fy);
    const float a0 = 1.f - fx, a1 = fx, b0 = 1.f - fy, b1 = fy;
    const int y0 = clip_row(sy, s.h), y1 = clip_row(sy + 1, s.h);
    const int x1 = sx + 1 < s.w ? sx + 1 : sx;
    const T* r0 = src + (size_t)s.off + (size_t)y0 * s.w;
    const T* r1 = src + (size_t)s.off + (size_t)y1 * s.w;
    const T h0 = r0[sx] * (T)a0 + r0[x1] * (T)a1;
    const T h1 = r1[sx] * (T)a0 + r1[x1] * (T)a1;
    dst[(size_t)d.off + p] = h0 * (T)b0 + h1 * (T)b1;
}

struct Gauss9 { float k[5]; };   // k[0] = centre tap, k[i] = taps at distance i

__global__ __launch_bounds__(256) void gauss9_row_kernel(const uint8_t* __restrict__ src, const int* __restrict__ desc,
                                                         float* __restrict__ tmp, Gauss9 g) {
    const Crop c = load_crop(desc, blockIdx.y);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= c.h * c.w) return;
    const int y = p / c.w, x = p - y * c.w;
    const uint8_t* row = src + ((size_t)c.off + (size_t)y * c.w) * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float s = g.k[0] * (float)row[x * 3 + ch];
#pragma unroll
        for (int i = 1; i <= 4; ++i)
            s += g.k[i] * ((float)row[reflect101(x + i, c.w) * 3 + ch] + (float)row[reflect101(x - i, c.w) * 3 + ch]);
        tmp[((size_t)c.off + p) * 3 + ch] = s;
    }
}

__global__ __launch_bounds__(256) void gauss9_col_kernel(const float* __restrict__ tmp, const int* __restrict__ desc,
                                                         uint8_t* __restrict__ dst, Gauss9 g) {
    const Crop c = load_crop(desc, blockIdx.y);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= c.h * c.w) return;
    const int y = p / c.w, x = p - y * c.w;
    const float* base = tmp + (size_t)c.off * 3;
#pragma unroll
    for (int ch = 0; ch < 3; ++ch) {
        float s = g.k[0] * base[((size_t)y * c.w + x) * 3 + ch];
#pragma unroll
        for (int i = 1; i <= 4; ++i)
            s += g.k[i] * (base[((size_t)reflect101(y + i, c.h) * c.w + x) * 3 + ch] +
                           base[((size_t)reflect101(y - i, c.h) * c.w + x) * 3 + ch]);
        const int v = __float2int_rn(s);
        dst[((size_t)c.off + p) * 3 + ch] = (uint8_t)(v < 0 ? 0 : (v > 255 ? 255 : v));
    }
}

// 3x3 Gaussian (1 2 1)(1 2 1)^T / 16 per channel, rounded half up, then the 8-bit RGB2GRAY of OpenCV
__global__ __launch_bounds__(256) void blur3_gray_kernel(const uint8_t* __restrict__ src, const int* __restrict__ desc,
                                                         uint8_t* __restrict__ gray) {
    const Crop c = load_crop(desc, blockIdx.y);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= c.h * c.w) return;
    const int y = p / c.w, x = p - y * c.w;
    const uint8_t* img = src + (size_t)c.off * 3;
    int acc[3] = {0, 0, 0};
#pragma unroll
    for (int j = -1; j <= 1; ++j) {
        const int yy = reflect101(y + j, c.h), wy = j == 0 ? 2 : 1;
#pragma unroll
        for (int i = -1; i <= 1; ++i) {
            const int xx = reflect101(x + i, c.w), wgt = wy * (i == 0 ? 2 : 1);
            const uint8_t* px = img + ((size_t)yy * c.w + xx) * 3;
            acc[0] += wgt * px[0]; acc[1] += wgt * px[1]; acc[2] += wgt * px[2];
        }
    }
    const int b0 = (acc[0] + 8) >> 4, b1 = (acc[1] + 8) >> 4, b2 = (acc[2] + 8) >> 4;
    gray[(size_t)c.off + p] = (uint8_t)((b0 * 4899 + b1 * 9617 + b2 * 1868 + (1 << 13)) >> 14);
}

// edge = |0.5 * sobel_x + 0.5 * sobel_y|; the crop's maximum is collected as the bit pattern of a non-negative float
// (monotone in the value, so an integer atomicMax gives the exact maximum independent of the arrival order)
__global__ __launch_bounds__(256) void sobel_kernel(const uint8_t* __restrict__ gray, const int* __restrict__ desc,
                                                    float* __restrict__ edge, unsigned* __restrict__ maxbits) {
    const Crop c = load_crop(desc, blockIdx.y);
    const int p = blockIdx.x * 256 + threadIdx.x;
    float e = 0.f;
    if (p < c.h * c.w) {
        const int y = p / c.w, x = p - y * c.w;
        const uint8_t* g = gray + (size_t)c.off;
        float v[3][3];
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int i = 0; i < 3; ++i)
                v[j][i] = (float)g[(size_t)reflect101(y + j - 1, c.h) * c.w + reflect101(x + i - 1, c.w)];
        const float gx = (v[0][2] - v[0][0]) + 2.f * (v[1][2] - v[1][0]) + (v[2][2] - v[2][0]);
        const float gy = (v[2][0] - v[0][0]) + 2.f * (v[2][1] - v[0][1]) + (v[2][2] - v[0][2]);
        e = fabsf(gx * 0.5f + gy * 0.5f);
        edge[(size_t)c.off + p] = e;
    }
    e = wave_max(e);
    if ((threadIdx.x & 63) == 0 && e > 0.f) atomicMax(maxbits + blockIdx.y, __float_as_uint(e));
}

__global__ __launch_bounds__(256) void edge_norm_kernel(float* __restrict__ edge, const int* __restrict__ desc,
                                                        const unsigned* __restrict__ maxbits) {
    const Crop c = load_crop(desc, blockIdx.y);
    const int p = blockIdx.x * 256 + threadIdx.x;
    if (p >= c.h * c.w) return;
    edge[(size_t)c.off + p] = edge[(size_t)c.off + p] / __uint_as_float(maxbits[blockIdx.y]);   // 0 / 0 = nan like np
}

__global__ void zero_u32_kernel(unsigned* p, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = 0u;
}

static inline dim3 crop_grid(int max_px, int n) { return dim3((max_px + 255) / 256, n); }

extern "C" int radet_resize_linear_u8(const uint8_t* src, const int* src_desc, uint8_t* dst, const int* dst_desc, int ncrop,
                                      int max_dst_px, int channels, void* stream) {
    if (ncrop < 0 || max_dst_px < 0 || (channels != 1 && channels != 3)) return RADET_ERR_ARG;
    if (ncrop == 0 || max_dst_px == 0) return RADET_OK;
    if (channels == 3)
        hipLaunchKernelGGL(resize_u8_kernel<3>, crop_grid(max_dst_px, ncrop), dim3(256), 0, (hipStream_t)stream, src, src_desc, dst, dst_desc);
    else
        hipLaunchKernelGGL(resize_u8_kernel<1>, crop_grid(max_dst_px, ncrop), dim3(256), 0, (hipStream_t)stream, src, src_desc, dst, dst_desc);
    return radet_check_launch();
}

extern "C" int radet_resize_linear_f(const void* src, const int* src_desc, void* dst, const int* dst_desc, int ncrop,
                                     int max_dst_px, int is_f64, void* stream) {
    if (ncrop < 0 || max_dst_px < 0) return RADET_ERR_ARG;
    if (ncrop == 0 || max_dst_px == 0) return RADET_OK;
    if (is_f64)
        hipLaunchKernelGGL(resize_f_kernel<double>, crop_grid(max_dst_px, ncrop), dim3(256), 0, (hipStream_t)stream,
                           (const double*)src, src_desc, (double*)dst, dst_desc);
    else
        hipLaunchKernelGGL(resize_f_kernel<float>, crop_grid(max_dst_px, ncrop), dim3(256), 0, (hipStream_t)stream,
                           (const float*)src, src_desc, (float*)dst, dst_desc);
    return radet_check_launch();
}

extern "C" int radet_gaussian_blur9_u8(const uint8_t* src, const int* desc, uint8_t* dst, float* tmp, const float* kernel5,
                                       int ncrop, int max_px, void* stream) {
    if (ncrop < 0 || max_px < 0 || !kernel5) return RADET_ERR_ARG;
    if (ncrop == 0 || max_px == 0) return RADET_OK;
    Gauss9 g;
    for (int i = 0; i < 5; ++i) g.k[i] = kernel5[i];
    hipLaunchKernelGGL(gauss9_row_kernel, crop_grid(max_px, ncrop), dim3(256), 0, (hipStream_t)stream, src, desc, tmp, g);
    hipLaunchKernelGGL(gauss9_col_kernel, crop_grid(max_px, ncrop), dim3(256), 0, (hipStream_t)stream, tmp, desc, dst, g);
    return radet_check_launch();
}

extern "C" int radet_sobel_edge(const uint8_t* src, const int* desc, float* edge, uint8_t* gray_ws, uint32_t* max_ws, int ncrop,
                                int max_px, void* stream) {
    if (ncrop < 0 || max_px < 0) return RADET_ERR_ARG;
    if (ncrop == 0 || max_px == 0) return RADET_OK;
    hipStream_t st = (hipStream_t)stream;
    hipLaunchKernelGGL(zero_u32_kernel, dim3((ncrop + 255) / 256), dim3(256), 0, st, max_ws, ncrop);
    hipLaunchKernelGGL(blur3_gray_kernel, crop_grid(max_px, ncrop), dim3(256), 0, st, src, desc, gray_ws);
    hipLaunchKernelGGL(sobel_kernel, crop_grid(max_px, ncrop), dim3(256), 0, st, gray_ws, desc, edge, max_ws);
    hipLaunchKernelGGL(edge_norm_kernel, crop_grid(max_px, ncrop), dim3(256), 0, st, edge, desc, max_ws);
    return radet_check_launch();
}
