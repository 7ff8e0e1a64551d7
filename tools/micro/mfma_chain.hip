// Micro-benchmark: sustained v_mfma_f32_32x32x2_f32 rate with 1 / 2 / 4 independent accumulator chains per wave and
// 1..4 waves per SIMD.   hipcc --offload-arch=gfx950 -O3 -o mfma_chain mfma_chain.hip && ./mfma_chain
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int CH>
__global__ __launch_bounds__(256) void chain(float* out, int iters, float a0, float b0) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    float a = a0 + threadIdx.x, b = b0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 16 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) acc[c] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[c], 0, 0, 0);
    }
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int CH>
void run(int wgs_per_cu) {
    float* out;
    const int grid = 256 * wgs_per_cu, iters = 20000;
    hipMalloc(&out, grid * 256 * sizeof(float));
    hipEvent_t s, e;
    hipEventCreate(&s); hipEventCreate(&e);
    chain<CH><<<grid, 256>>>(out, 100, 1.f, 1.f);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(s);
        chain<CH><<<grid, 256>>>(out, iters, 1.f, 1.f);
        hipEventRecord(e);
        hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    const double fl = (double)grid * 4 * iters * 16 * (2.0 * 32 * 32 * 2);
    printf("chains=%d waves/SIMD=%d: %.2f ms  %.1f TFLOP/s\n", CH, wgs_per_cu, best, fl / best / 1e9);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) { run<1>(w); run<2>(w); run<4>(w); }
    return 0;
}
