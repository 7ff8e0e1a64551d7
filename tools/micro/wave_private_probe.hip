// Probe: barrier-free K loop with wave-private LDS staging.  Every wave DMA-loads its own A (32 rows x 32 k) and B
// (32 rows x 32 k) fp32 slices of a K stage (8 global_load_lds_dwordx4 = 8 KiB), NST stages ahead, reads its fragments
// back with ds_read_b128 and runs 16 v_mfma_f32_32x32x2_f32; the only synchronisation is the wave's own vmcnt.
// Workgroup = 2 waves (one 64 x 32 output block); grid = 512 x w -> 1 x w waves per SIMD.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int NST, bool BARRIER>
__global__ __launch_bounds__(128) void probe(const float* __restrict__ A, const float* __restrict__ B, float* out, int K, int rowsA) {
    __shared__ __attribute__((aligned(16))) float buf[NST][2][2048];       // [stage][wave][A 1024 | B 1024]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, li = lane & 31, lh = lane >> 5;
    // DMA lane mapping: a wave load = 8 rows x 128 B; lane -> row (lane / 8), 16-byte slot (lane % 8)
    const int blk = blockIdx.x;
    const float* ap = A + ((size_t)((blk * 64 + wave * 32) % rowsA) + lane / 8) * K + 4 * (lane % 8);
    const float* bp = B + ((size_t)((blk * 7 % 4) * 64) + lane / 8) * K + 4 * (lane % 8);
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    auto issue = [&](int st, int k0) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            __builtin_amdgcn_global_load_lds((gptr_t)(ap + (size_t)q * 8 * K + k0), (lptr_t)(&buf[st][wave][q * 256]), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((gptr_t)(bp + (size_t)q * 8 * K + k0), (lptr_t)(&buf[st][wave][1024 + q * 256]), 16, 0, 0);
        }
    };
    const unsigned base = (unsigned)(size_t)(lptr_t)(&buf[0][wave][0]) + (unsigned)(li * 128 + lh * 16);
    auto compute = [&](int st) {
        f32x4 a[4], b[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            asm volatile("ds_read_b128 %0, %1" : "=v"(a[s]) : "v"(base + st * 16384 + s * 32) : "memory");
            asm volatile("ds_read_b128 %0, %1" : "=v"(b[s]) : "v"(base + st * 16384 + 4096 + s * 32) : "memory");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            asm volatile("" : "+v"(a[s]), "+v"(b[s]));
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].x, b[s].x, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].y, b[s].y, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].z, b[s].z, acc, 0, 0, 0);
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(a[s].w, b[s].w, acc, 0, 0, 0);
        }
    };
    const int nK = K / 32;
#pragma unroll
    for (int st = 0; st < NST - 1; ++st) issue(st, 32 * st);
    for (int it = 0; it < nK; it += NST) {
#pragma unroll
        for (int u = 0; u < NST; ++u) {
            if (it + u < nK) {
                const int nxt = it + u + NST - 1;
                if (nxt < nK) {
                    issue((u + NST - 1) % NST, 32 * nxt);
                    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(8 * (NST - 1)) : "memory");
                } else {
                    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                }
                if (BARRIER) __syncthreads();
                compute(u);
                if (BARRIER) __syncthreads();
            }
        }
    }
    float s = 0;
    for (int r = 0; r < 16; ++r) s += acc[r];
    out[blockIdx.x * 128 + tid] = s;
}

template <int NST, bool BARRIER>
void run(int grid, int K) {
    const int rowsA = 65536;
    float *A, *B, *out;
    hipMalloc(&A, (size_t)(rowsA + 64) * K * 4); hipMalloc(&B, (size_t)320 * K * 4);
    hipMemset(A, 0, (size_t)(rowsA + 64) * K * 4); hipMemset(B, 0, (size_t)320 * K * 4);
    hipMalloc(&out, (size_t)grid * 128 * 4);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    float best = 1e9;
    for (int r = 0; r < 6; ++r) {
        hipEventRecord(s);
        for (int q = 0; q < 10; ++q) probe<NST, BARRIER><<<grid, 128>>>(A, B, out, K, rowsA);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms / 10 < best) best = ms / 10;
    }
    const double fl = (double)grid * 2 * (K / 32) * 16 * 4096.0;
    printf("grid %4d (%d waves/SIMD) K=%4d %d stages %s: %7.1f us  %6.1f TFLOP/s\n", grid, grid / 512, K, NST,
           BARRIER ? "barrier/stage" : "no barrier   ", best * 1e3, fl / best / 1e9);
    hipFree(A); hipFree(B); hipFree(out);
}
int main() {
    for (int g = 512; g <= 2048; g *= 2) {
        run<2, true>(g, 1024); run<2, false>(g, 1024); run<3, false>(g, 1024); run<4, false>(g, 1024);
    }
    run<3, false>(512, 2304); run<3, false>(1024, 2304);
    return 0;
}
