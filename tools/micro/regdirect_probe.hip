// Probe for a barrier-free GEMM inner loop: every wave loads its own 32 x 32 (A) and 32 x 32 (B) fp32 operand slices of a
// K stage (BK = 32) straight into registers with 8 global_load_dwordx4, NST stages ahead of the 16 MFMAs that consume
// them; no LDS, no barrier, the compiler places the vmcnt waits.  Streams `rows` x K matrices (A rows shared by the two
// waves of a row pair, like a 64 x 64 tile).   hipcc --offload-arch=gfx950 -O3 -o rd regdirect_probe.hip && ./rd
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NST>
__global__ __launch_bounds__(256) void probe(const float* __restrict__ A, const float* __restrict__ B, float* out, int K, int tiles_m) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    const int wm = wave >> 1, wn = wave & 1;
    const int tm = blockIdx.x % tiles_m, tn = blockIdx.x / tiles_m;
    const float* ap = A + (size_t)(tm * 64 + wm * 32 + li) * K + 4 * lh;
    const float* bp = B + (size_t)(tn * 64 + wn * 32 + li) * K + 4 * lh;
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    f32x4 a[NST][4], b[NST][4];
    auto load = [&](int st, int k0) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            a[st][s] = *reinterpret_cast<const f32x4*>(ap + k0 + 8 * s);
            b[st][s] = *reinterpret_cast<const f32x4*>(bp + k0 + 8 * s);
        }
    };
    auto mma = [&](int st) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][s].x, b[st][s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][s].y, b[st][s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][s].z, b[st][s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[st][s].w, b[st][s].w, acc, 0, 0, 0);
        }
    };
    const int nK = K / 32;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) load(st, 32 * st);
    for (int it = 0; it < nK; it += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            const int nxt = it + u + NST - 1;
            if (nxt < nK) load((u + NST - 1) % NST, 32 * nxt);
            if (it + u < nK) mma(u);
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 256 + tid] = s;
}

template <int NST>
void run(int M, int N, int K) {
    float *A, *B, *out;
    hipMalloc(&A, (size_t)M * K * 4); hipMalloc(&B, (size_t)N * K * 4);
    hipMemset(A, 0, (size_t)M * K * 4); hipMemset(B, 0, (size_t)N * K * 4);
    const int tiles = (M / 64) * (N / 64);
    hipMalloc(&out, (size_t)tiles * 256 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(s);
        for (int q = 0; q < 10; ++q) probe<NST><<<tiles, 256>>>(A, B, out, K, M / 64);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms / 10 < best) best = ms / 10;
    }
    printf("M=%5d N=%4d K=%4d, %d stages in registers, %4d tiles: %7.1f us  %6.1f TFLOP/s\n", M, N, K, NST, tiles, best * 1e3,
           2.0 * M * N * K / best / 1e9);
    hipFree(A); hipFree(B); hipFree(out);
}
int main() {
    run<2>(4800 / 64 * 64, 256, 1024); run<3>(4800 / 64 * 64, 256, 1024); run<4>(4800 / 64 * 64, 256, 1024);
    run<3>(4096 * 4, 256, 1024); run<4>(4096 * 4, 256, 1024);       // exactly 1024 tiles: 4 per CU
    run<3>(4096, 256, 1024); run<4>(4096, 256, 1024);               // exactly 256 tiles: 1 per CU
    run<3>(4096, 256, 2304); run<4>(4096, 256, 2304);
    run<3>(25600, 256, 2304); run<4>(25600, 256, 2304);
    return 0;
}
