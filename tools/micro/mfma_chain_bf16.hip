// Sustained v_mfma_f32_32x32x16_bf16 rate with 1 / 2 / 4 independent accumulator chains per wave (dependent
// back-to-back accumulation vs interleaved chains), 1..2 waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

template <int CH>
__global__ __launch_bounds__(256) void chain(float* out, int iters, unsigned seed) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    u32x4 ua = {seed + threadIdx.x, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u}, ub = {0x3f803f80u, seed, 0x3f803f80u, 0x3f803f80u};
    const bf16x8 a = __builtin_bit_cast(bf16x8, ua), b = __builtin_bit_cast(bf16x8, ub);
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

// the same loop with 8 different pseudo-random operand pairs per lane in rotation (realistic toggle rate of the data path;
// the constant-operand loop above is the best case for power) and the shader clock measured around it
template <int CH>
__global__ __launch_bounds__(256) void chain_rnd(float* out, int iters, unsigned seed, long long* clk) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    bf16x8 a[8], b[8];
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
    for (int k = 0; k < 8; ++k) {
        u32x4 ua, ub;
        for (int q = 0; q < 4; ++q) {
            x = x * 1664525u + 1013904223u; ua[q] = (x & 0x807F807Fu) | 0x3F003F00u;     // bf16 pairs in [0.5, 1), random sign / mantissa
            x = x * 1664525u + 1013904223u; ub[q] = (x & 0x807F807Fu) | 0x3F003F00u;
        }
        a[k] = __builtin_bit_cast(bf16x8, ua); b[k] = __builtin_bit_cast(bf16x8, ub);
    }
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c)
                acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a[(u * CH + c) & 7], b[(u * CH + c + 3) & 7], acc[c], 0, 0, 0);
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int CH>
void run_rnd(int w) {
    float* out; long long* clk;
    const int grid = 256 * w, iters = 20000;
    hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&clk, 16);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    chain_rnd<CH><<<grid, 256>>>(out, 100, 1, clk);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(s);
        chain_rnd<CH><<<grid, 256>>>(out, iters, 1, clk);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)grid * 4 * iters * 24 * (2.0 * 32 * 32 * 16);
    printf("bf16 32x32x16, random operands: chains=%d waves/SIMD=%d: %.2f ms  %.0f TFLOP/s  s_memtime/s_memrealtime = %.2f\n", CH, w,
           best, fl / best / 1e9, (double)h[0] / h[1]);
    hipFree(out); hipFree(clk);
}

template <int CH>
void run(int w) {
    float* out;
    const int grid = 256 * w, iters = 20000;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    chain<CH><<<grid, 256>>>(out, 100, 1);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(s);
        chain<CH><<<grid, 256>>>(out, iters, 1);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double fl = (double)grid * 4 * iters * 24 * (2.0 * 32 * 32 * 16);
    printf("bf16 32x32x16: chains=%d waves/SIMD=%d: %.2f ms  %.0f TFLOP/s\n", CH, w, best, fl / best / 1e9);
    hipFree(out);
}
int main() {
    for (int w = 1; w <= 2; ++w) { run<1>(w); run<4>(w); }
    for (int w = 1; w <= 2; ++w) { run_rnd<1>(w); run_rnd<4>(w); }
    return 0;
}
