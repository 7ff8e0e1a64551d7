// Shared device/host helpers for the radet_amd HIP kernels (gfx950 / CDNA4 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define RADET_MAX_SEG 8

// One pyramid level (or a plain tensor when nseg == 1) of a row-concatenated NHWC buffer:
// rows [row_begin, row_end) of the "output side" hold B images of Ho x Wo pixels; the matching
// "input side" level starts at input row in_row_off and has Hi x Wi pixels per image.
struct RadetSeg {
    int row_end;     // exclusive end row (cumulative over segments) on the output side
    int row_begin;   // first output row of this segment
    int in_row_off;  // first input row of this segment
    int Hi, Wi, Ho, Wo;
};

struct RadetSegs {
    int nseg;
    RadetSeg s[RADET_MAX_SEG];
};

#define RADET_OK 0
#define RADET_ERR_ARG -1
#define RADET_ERR_LAUNCH -2

static inline int radet_check_launch() {
    hipError_t e = hipGetLastError();
    return e == hipSuccess ? RADET_OK : RADET_ERR_LAUNCH;
}

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ double wave_sum_d(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
    return v;
}

// XCD-aware bijective remap of a 1-D grid: blocks that land on the same XCD (bid % 8) get a
// contiguous chunk of tile ids, so neighbouring tiles share that XCD's L2.
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7;
    const int xcd = bid & 7, k = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + k;
}

// ---- activation element access, fp32 or bf16 storage (4 consecutive elements per call; i4 = index of the group)
typedef float rd_f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 rd_bf16x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ unsigned radet_pack_bf16(float lo, float hi) {          // v_cvt_pk_bf16_f32, round to nearest even
    const rd_f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, rd_bf16x2));
}
template <class T> __device__ __forceinline__ float4 ld4(const T* p, size_t i4);
template <> __device__ __forceinline__ float4 ld4<float>(const float* p, size_t i4) {
    return reinterpret_cast<const float4*>(p)[i4];
}
template <> __device__ __forceinline__ float4 ld4<__bf16>(const __bf16* p, size_t i4) {
    const uint2 u = reinterpret_cast<const uint2*>(p)[i4];
    return make_float4(__uint_as_float(u.x << 16), __uint_as_float(u.x & 0xFFFF0000u), __uint_as_float(u.y << 16),
                       __uint_as_float(u.y & 0xFFFF0000u));
}
template <class T> __device__ __forceinline__ void st4(T* p, size_t i4, float4 v);
template <> __device__ __forceinline__ void st4<float>(float* p, size_t i4, float4 v) {
    reinterpret_cast<float4*>(p)[i4] = v;
}
template <> __device__ __forceinline__ void st4<__bf16>(__bf16* p, size_t i4, float4 v) {
    reinterpret_cast<uint2*>(p)[i4] = make_uint2(radet_pack_bf16(v.x, v.y), radet_pack_bf16(v.z, v.w));
}

// ---- bf16 plane triples: an fp32 value x is stored as three bf16 numbers hi + mid + lo == x EXACTLY (hi = the top 16
// bits of x, mid = the top 16 bits of x - hi, lo = x - hi - mid: 8 significand bits each, truncation, both residuals are
// exact in fp32).  A row of C channels (C % 32 == 0) is C / 32 groups of 192 bytes: [hi of 32 channels | mid | lo] -- the
// three planes of a 32-channel group are contiguous, so the K = 32-channel stage of a conv GEMM reads 192 contiguous bytes
// per row (1.5 cache lines; plane-major rows would be three half-used lines: the L2 -> L1 path is the bound of those
// loaders, measured in tools/micro/fill_probe.hip).  The conv GEMMs read the planes straight into
// v_mfma_f32_32x32x16_bf16 (conv_igemm.hip, P3); producers split ONCE per element here instead of once per use inside the
// GEMMs' K loops.
__host__ __device__ __forceinline__ size_t radet_plane_off(int c) {      // element offset of channel c's hi value in its row
    return (size_t)(c >> 5) * 96 + (c & 31);                              // (mid: + 32, lo: + 64)
}
__device__ __forceinline__ void radet_split3(const float4 v, uint2& hi, uint2& mid, uint2& lo) {
    const float x[4] = {v.x, v.y, v.z, v.w};
    unsigned h[4], m[4], l[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        h[i] = __float_as_uint(x[i]) & 0xFFFF0000u;
        const float r = x[i] - __uint_as_float(h[i]);
        m[i] = __float_as_uint(r) & 0xFFFF0000u;
        l[i] = __float_as_uint(r - __uint_as_float(m[i]));
    }
    hi = make_uint2((h[0] >> 16) | h[1], (h[2] >> 16) | h[3]);
    mid = make_uint2((m[0] >> 16) | m[1], (m[2] >> 16) | m[3]);
    lo = make_uint2((l[0] >> 16) | (l[1] & 0xFFFF0000u), (l[2] >> 16) | (l[3] & 0xFFFF0000u));
}
// 4 consecutive channels (col4 * 4 ..) of plane row `row` (C channels per row)
__device__ __forceinline__ void st4_planes(__bf16* p, size_t row, int C, int col4, const float4 v) {
    uint2 hi, mid, lo;
    radet_split3(v, hi, mid, lo);
    uint2* q = reinterpret_cast<uint2*>(p + row * 3 * (size_t)C + radet_plane_off(col4 * 4));
    q[0] = hi;
    q[8] = mid;
    q[16] = lo;
}
__device__ __forceinline__ float4 ld4_planes(const __bf16* p, size_t row, int C, int col4) {
    const uint2* q = reinterpret_cast<const uint2*>(p + row * 3 * (size_t)C + radet_plane_off(col4 * 4));
    const uint2 h = q[0], m = q[8], l = q[16];
    float4 r;        // (hi + mid) + lo: both additions exact
    r.x = (__uint_as_float(h.x << 16) + __uint_as_float(m.x << 16)) + __uint_as_float(l.x << 16);
    r.y = (__uint_as_float(h.x & 0xFFFF0000u) + __uint_as_float(m.x & 0xFFFF0000u)) + __uint_as_float(l.x & 0xFFFF0000u);
    r.z = (__uint_as_float(h.y << 16) + __uint_as_float(m.y << 16)) + __uint_as_float(l.y << 16);
    r.w = (__uint_as_float(h.y & 0xFFFF0000u) + __uint_as_float(m.y & 0xFFFF0000u)) + __uint_as_float(l.y & 0xFFFF0000u);
    return r;
}
