// Shader clock actually held under load: s_memtime (shader clock counter) against s_memrealtime (100 MHz) around
//   (a) a register-only MFMA loop, (b) MFMA + LDS fragment reads, (c) MFMA + LDS reads + streaming global->LDS loads,
// on every CU at once (256 x W workgroups).   hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ src, float* out, long long* stamps, int iters, size_t span) {
    __shared__ __attribute__((aligned(16))) float tile[4][4096];      // MODE 2 / 3 / 4: 2 / 3 / 4 stages in flight
    constexpr int STG = MODE < 2 ? 2 : (MODE == 5 ? 3 : MODE);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int i = tid; i < 16384; i += 256) (&tile[0][0])[i] = 1.f + i;
    __syncthreads();
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const size_t base = (size_t)blockIdx.x * 65536 + wave * 256 + lane * 4, lim = span - 8192;
    for (int it = 0; it < iters; ++it) {
        if (MODE >= 2) {
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (base + (size_t)it * 4096) % lim), (lptr_t)(&tile[it % STG][wave * 256]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(src + (base + (size_t)it * 4096 + 2048) % lim), (lptr_t)(&tile[it % STG][1024 + wave * 256]), 16, 0, 0);
        }
        f32x4 a, b;
        if (MODE >= 1) {
            a = *reinterpret_cast<const f32x4*>(&tile[(it + 1) % STG][(lane * 4 + wave * 256) & 4095]);
            b = *reinterpret_cast<const f32x4*>(&tile[(it + 1) % STG][(2048 + lane * 4 + wave * 256) & 4095]);
        } else {
            a = f32x4{1.f, 2.f, 3.f, 4.f}; b = a;
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.x, b.x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.y, b.y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.z, b.z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a.w, b.w, acc, 0, 0, 0);
        }
        if (MODE == 2) { asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); __syncthreads(); }
        if (MODE == 3) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); __syncthreads(); }
        if (MODE == 4) { asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); __syncthreads(); }
        if (MODE == 5) { asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); }      // 3 stages, wave-private tiles: no barrier
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + tid] = s;
    if (tid == 0 && (blockIdx.x & 63) == 0) { stamps[(blockIdx.x >> 6) * 2] = t1 - t0; stamps[(blockIdx.x >> 6) * 2 + 1] = r1 - r0; }
}

template <int MODE>
void run(int w, const char* what) {
    const int grid = 256 * w, iters = 40000 / w;
    const size_t span = (size_t)64 << 20;       // floats: 256 MiB source
    float *src, *out; long long* st;
    hipMalloc(&src, span * 4); hipMalloc(&out, (size_t)grid * 256 * 4); hipMalloc(&st, 64 * 16);
    hipMemset(src, 0, span * 4);
    probe<MODE><<<grid, 256>>>(src, out, st, 100, span);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    hipEventRecord(s);
    probe<MODE><<<grid, 256>>>(src, out, st, iters, span);
    hipEventRecord(e); hipEventSynchronize(e);
    float ms; hipEventElapsedTime(&ms, s, e);
    long long h[2]; hipMemcpy(h, st, 16, hipMemcpyDeviceToHost);
    const double fl = (double)grid * 4 * iters * 16 * 4096.0;
    printf("%-44s %d WG/CU: %.1f TFLOP/s, s_memtime / s_memrealtime = %.3f (x 100 MHz)\n", what, w, fl / ms / 1e9, (double)h[0] / h[1]);
    hipFree(src); hipFree(out); hipFree(st);
}
int main() {
    for (int w = 1; w <= 4; w *= 2) {
        run<0>(w, "MFMA only");
        run<1>(w, "MFMA + LDS reads");
        run<2>(w, "MFMA + LDS reads + global->LDS, 2 stages");
        run<3>(w, "MFMA + LDS reads + global->LDS, 3 stages");
        run<4>(w, "MFMA + LDS reads + global->LDS, 4 stages");
        run<5>(w, "same, 3 stages, no barrier (wave-private)");
    }
    return 0;
}
