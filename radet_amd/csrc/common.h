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
