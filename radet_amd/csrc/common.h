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

// Touch every 64-byte line of the kernel-argument segment at once.  The compiler fetches by-value argument structs field by
// field where they are first used, each fetch behind its own s_waitcnt: a chain of scalar-cache misses (the segment was just
// written by the command processor) at the head of every workgroup.  One batch of independent s_load_dword + one wait turns
// the chain into a single miss latency; the later field loads hit the scalar cache.  BYTES = sizeof(the argument struct).
template <int BYTES>
__device__ __forceinline__ void radet_kernarg_warm() {
    typedef const char __attribute__((address_space(4))) * kptr_t;
    const kptr_t kp = (kptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    constexpr int N = (BYTES + 63) / 64;
    unsigned d[N];
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("s_load_dword %0, %1, %2" : "=s"(d[i]) : "s"(kp), "n"(i * 64));
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
    for (int i = 0; i < N; ++i) asm volatile("" ::"s"(d[i]));       // (the destination registers stay reserved until here)
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

// ---- fp16 plane pairs ("h2", round 5): an fp32 value x of a tensor whose largest magnitude is known (or bounded) by `amax`
// is stored / fed to the matrix cores as TWO fp16 numbers  hi = fp16(t),  lo = fp16((t - hi) * 2^11),  t = x * 2^e,
// e = radet_h2_exp(amax) (an exact power-of-two scale that puts the largest |t| into [2^14, 2^15): hi never overflows).
// Both conversions round to nearest even; t - hi is exact in fp32, so x = 2^-e (hi + 2^-11 lo) up to the rounding of lo:
// <= 2^-23 |x| for every element within 2^-28 of the tensor's largest (hi normal); smaller elements keep an ABSOLUTE
// error of 2^-36 2^-e, i.e. 2^-51 of the largest element.  A product of two such numbers is formed as
// hi hi' + 2^-11 (hi lo' + lo hi') -- three v_mfma_f32_32x32x16_f16 into two fp32 accumulators instead of the six bf16
// plane products of the round-2 scheme; the dropped lo lo' term is <= 2^-24 relative.  `amax` lives in a 4-byte device slot
// next to the tensor: fp32 tensors get theirs from the producing kernel's epilogue (atomicMax of |y| bit patterns, order
// independent), plane tensors from a bound their producer computes before it writes (GroupNorm), see DESIGN.md 3.
__host__ __device__ __forceinline__ int radet_h2_exp(unsigned amax_bits) {
    const int ex = (int)((amax_bits >> 23) & 0xFFu);          // biased exponent of the largest magnitude
    if (ex == 0 || ex == 255) return 0;                       // all zero (or denormal) / not finite: no scaling
    const int e = 141 - ex;                                   // 14 - (ex - 127)
    return e > 100 ? 100 : e;                                 // (2^(e + 11) must stay a normal float)
}
__host__ __device__ __forceinline__ float radet_pow2(int e) {           // -126 <= e <= 127
#if defined(__HIP_DEVICE_COMPILE__)
    return __uint_as_float((unsigned)(e + 127) << 23);
#else
    union { unsigned u; float f; } c; c.u = (unsigned)(e + 127) << 23; return c.f;
#endif
}
// two values -> packed (hi0 | hi1 << 16), (lo0 | lo1 << 16); s = 2^e, s2 = 2^(e + 11).  3 VALU operations per element:
// v_mul (u = x s2), v_fma_mixlo/hi_f16 (hi = fp16(x s)), v_fma_mixlo/hi_f16 (lo = fp16(u - 2048 hi): one rounding)
__device__ __forceinline__ void radet_split2(float x0, float x1, float s, float s2, unsigned& hi, unsigned& lo) {
    const float u0 = x0 * s2, u1 = x1 * s2;
    unsigned h, l;
    asm("v_fma_mixlo_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "=v"(h) : "v"(x0), "s"(s));
    asm("v_fma_mixhi_f16 %0, %1, %2, 0 op_sel:[0,0,0] op_sel_hi:[0,0,0]" : "+v"(h) : "v"(x1), "s"(s));
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(l) : "v"(h), "s"(-2048.0f), "v"(u0));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(l) : "v"(h), "s"(-2048.0f), "v"(u1));
    hi = h;
    lo = l;
}
// the value a pair stands for, in units of 2^-e: hi + lo / 2048 (exact in fp32: 11 + 11 significand bits)
__device__ __forceinline__ float radet_pair_value(unsigned short hi, unsigned short lo) {
    return (float)__builtin_bit_cast(_Float16, hi) + (float)__builtin_bit_cast(_Float16, lo) * (1.0f / 2048.0f);
}
// A row of C channels (C % 32 == 0) is C / 32 groups of 128 bytes: [hi of 32 channels | lo of 32 channels] -- one cache
// line per 32-channel K stage and row, 4 bytes per element like the fp32 tensor it replaces.
__host__ __device__ __forceinline__ size_t radet_pair_off(int c) {       // element offset of channel c's hi value in its row
    return (size_t)(c >> 5) * 64 + (c & 31);                              // (lo: + 32)
}
__device__ __forceinline__ void st4_pairs(_Float16* p, size_t row, int C, int col4, const float4 v, float s, float s2) {
    unsigned h0, l0, h1, l1;
    radet_split2(v.x, v.y, s, s2, h0, l0);
    radet_split2(v.z, v.w, s, s2, h1, l1);
    uint2* q = reinterpret_cast<uint2*>(p + row * 2 * (size_t)C + radet_pair_off(col4 * 4));
    q[0] = make_uint2(h0, h1);
    q[8] = make_uint2(l0, l1);
}
__device__ __forceinline__ float4 ld4_pairs(const _Float16* p, size_t row, int C, int col4, float inv_s) {
    const uint2* q = reinterpret_cast<const uint2*>(p + row * 2 * (size_t)C + radet_pair_off(col4 * 4));
    const uint2 h = q[0], l = q[8];
    float4 r;
    r.x = radet_pair_value((unsigned short)(h.x & 0xFFFFu), (unsigned short)(l.x & 0xFFFFu)) * inv_s;
    r.y = radet_pair_value((unsigned short)(h.x >> 16), (unsigned short)(l.x >> 16)) * inv_s;
    r.z = radet_pair_value((unsigned short)(h.y & 0xFFFFu), (unsigned short)(l.y & 0xFFFFu)) * inv_s;
    r.w = radet_pair_value((unsigned short)(h.y >> 16), (unsigned short)(l.y >> 16)) * inv_s;
    return r;
}
// An amax slot is RADET_AMAX_WORDS (64) words = 256 bytes; its value is the LARGEST of the words.  Bit patterns of
// non-negative floats order like unsigned integers and atomicMax is order independent, so the slot's final value -- and with
// it every scale derived from it -- is the same in every run.  Why 64 words: the ~3000 waves of a launch's first round finish
// together and each raises the slot once; on ONE address these device-scope atomics serialise (measured: a fixed +35 us per
// launch on the 76 800-row layer1 convs; 64 words in two cache lines: still +10-16 us), spread by (workgroup, wave) over 64
// words in 64 different 128-byte lines they do not (+0.5-3 us, tools/bench_h2.py).  On ONE word a look before the
// read-modify-write helped (the words only grow); on 64 lines it costs more than it saves -- the load is a round trip at the
// tail of every wave, the atomic is fire-and-forget: 8.45 against 8.55 ms per step without the look.
#ifndef RADET_AMAX_WORDS
#define RADET_AMAX_WORDS 64
#endif
#ifndef RADET_AMAX_STRIDE
#define RADET_AMAX_STRIDE 32           // distance between the words of a slot, in words: one 128-byte line per word (see above)
#endif
// the slot's value; call with all 64 lanes of the wave active (every lane loads one word).  Wave-uniform.
// In two halves, so that a GEMM can put the (cold: every launch starts on invalidated L2s) load at its very top and the
// reduction behind the wait for its first tiles: radet_amax_load issues the gather, radet_amax_reduce is four DPP maxima
// inside the rows of 16 lanes + four v_readlane (a ds_bpermute butterfly is six dependent LDS round trips).
__device__ __forceinline__ unsigned radet_amax_load(const unsigned* slot) {
    return slot[(threadIdx.x & (RADET_AMAX_WORDS - 1)) * RADET_AMAX_STRIDE];
}
__device__ __forceinline__ unsigned radet_amax_reduce(unsigned v) {
    auto mx = [](unsigned a, unsigned b) { return a > b ? a : b; };
    v = mx(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0xB1, 0xF, 0xF, true));       // quad_perm [1, 0, 3, 2]
    v = mx(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x4E, 0xF, 0xF, true));       // quad_perm [2, 3, 0, 1]
    v = mx(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x141, 0xF, 0xF, true));      // row_half_mirror
    v = mx(v, (unsigned)__builtin_amdgcn_mov_dpp((int)v, 0x140, 0xF, 0xF, true));      // row_mirror: every lane of a row holds the row's maximum
    const unsigned r0 = (unsigned)__builtin_amdgcn_readlane((int)v, 0), r1 = (unsigned)__builtin_amdgcn_readlane((int)v, 16);
    const unsigned r2 = (unsigned)__builtin_amdgcn_readlane((int)v, 32), r3 = (unsigned)__builtin_amdgcn_readlane((int)v, 48);
    return mx(mx(r0, r1), mx(r2, r3));
}
__device__ __forceinline__ unsigned radet_amax_read(const unsigned* slot) { return radet_amax_reduce(radet_amax_load(slot)); }
// largest magnitude seen by this wave (m >= 0 in every lane, all 64 lanes active): the bit patterns of non-negative floats order
// like the floats, so the wave maximum is the DPP reduction below on the bits
__device__ __forceinline__ void radet_amax_publish(float m, unsigned* slot) {
    const unsigned bits = radet_amax_reduce(__float_as_uint(m > 0.f ? m : 0.f));       // (NaN, -0: not published, as fmaxf would)
    if ((threadIdx.x & 63) == 0 && bits != 0u) {
        unsigned* w = slot + ((blockIdx.x * 8u + (threadIdx.x >> 6)) & (RADET_AMAX_WORDS - 1)) * RADET_AMAX_STRIDE;
        atomicMax(w, bits);          // result unused: a fire-and-forget L2 atomic -- nothing at the wave's tail waits for it
    }
}
// a producer that KNOWS the value (a bound computed before writing): word 0, the other words stay zero
__device__ __forceinline__ void radet_amax_store(unsigned* slot, unsigned bits) { slot[0] = bits; }
