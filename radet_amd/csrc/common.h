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
