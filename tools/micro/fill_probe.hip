// Global -> LDS fill throughput on gfx950: LDS-DMA (global_load_lds_dwordx4) vs global_load_dwordx4 + ds_write_b128, for the
// row-chunk gather patterns of the conv GEMM loaders (each wave instruction moves 1 KiB: 64 lanes x 16 B; a "row chunk" of
// CH bytes is read by CH / 16 consecutive lanes, rows are `stride` bytes apart).  Prints GB/s per CU and TB/s aggregate.
//   hipcc --offload-arch=gfx950 -O3 tools/micro/fill_probe.hip -o tools/_probe/fill_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

template <int MODE, int L, int NT>
__global__ __launch_bounds__(NT) void fill(const char* __restrict__ src, float* out, int iters, int chunk, int stride,
                                           long window_rows, int rows_per_wg_iter) {
    __shared__ __attribute__((aligned(16))) float lds[2][NT / 64 * L * 256];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int lpr = chunk / 16;                    // lanes per row chunk
    const int rpi = 64 / lpr;                      // rows per wave instruction
    long row0 = ((long)blockIdx.x * 7919) % window_rows;
    float4 acc = {0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
        const int buf = it & 1;
        float4 v[L];
#pragma unroll
        for (int k = 0; k < L; ++k) {
            long r = row0 + (long)(wave * L + k) * rpi + lane / lpr;
            if (r >= window_rows) r -= window_rows;
            const char* p = src + r * stride + (long)((it % (stride / chunk)) * chunk) + (lane % lpr) * 16;
            if (MODE == 0)
                __builtin_amdgcn_global_load_lds((gptr_t)p, (lptr_t)(&lds[buf][(wave * L + k) * 256]), 16, 0, 0);
            else
                v[k] = *reinterpret_cast<const float4*>(p);
        }
        if (MODE == 1) {
#pragma unroll
            for (int k = 0; k < L; ++k) *reinterpret_cast<float4*>(&lds[buf][(wave * L + k) * 256 + lane * 4]) = v[k];
        }
        if ((it % (stride / chunk)) == stride / chunk - 1) {
            row0 += rows_per_wg_iter;
            if (row0 >= window_rows) row0 -= window_rows;
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    acc.x = lds[0][threadIdx.x] + lds[1][threadIdx.x];
    out[blockIdx.x * NT + threadIdx.x] = acc.x;
}

template <int MODE, int L, int NT>
void run(const char* name, const char* src, float* out, int wgs_per_cu, int chunk, int stride, long window_rows) {
    const int iters = 2000;
    const int grid = 256 * wgs_per_cu;
    hipEvent_t a, b;
    (void)hipEventCreate(&a); (void)hipEventCreate(&b);
    const int rows_per_iter = NT / 64 * L * (64 / (chunk / 16));
    fill<MODE, L, NT><<<grid, NT>>>(src, out, 50, chunk, stride, window_rows, rows_per_iter);
    hipDeviceSynchronize();
    hipEventRecord(a);
    fill<MODE, L, NT><<<grid, NT>>>(src, out, iters, chunk, stride, window_rows, rows_per_iter);
    hipEventRecord(b);
    hipEventSynchronize(b);
    float ms;
    hipEventElapsedTime(&ms, a, b);
    const double bytes = (double)grid * iters * (NT / 64) * L * 1024.0;
    printf("%-6s L=%d NT=%d wg/CU=%d chunk=%4d stride=%5d window=%7ld rows (%6.1f MB): %7.1f GB/s/CU  %6.2f TB/s  (%5.1f B/clk/CU @2.4GHz)\n",
           name, L, NT, wgs_per_cu, chunk, stride, window_rows, window_rows * (double)stride / 1e6, bytes / ms / 1e6 / 256,
           bytes / ms / 1e9, bytes / ms / 1e6 / 256 / 2.4);
}

int main() {
    const long maxrows = 400000;
    char* src; float* out;
    hipMalloc(&src, maxrows * 2048);
    hipMemset(src, 1, maxrows * 2048);
    hipMalloc(&out, 256 * 8 * 512 * 4);
    for (long window : {2000L, 50000L, 400000L}) {           // ~3 MB (L2), 77 MB (Infinity Cache), 614 MB (HBM)
        for (int chunk : {32, 64, 128, 512}) {
            const int stride = 1536;
            run<0, 6, 256>("dma", src, out, 1, chunk, stride, window);
            run<0, 6, 256>("dma", src, out, 2, chunk, stride, window);
            run<0, 6, 256>("dma", src, out, 3, chunk, stride, window);
            run<0, 6, 512>("dma", src, out, 1, chunk, stride, window);
            run<0, 9, 512>("dma", src, out, 1, chunk, stride, window);
            run<1, 6, 256>("vgpr", src, out, 1, chunk, stride, window);
            run<1, 6, 256>("vgpr", src, out, 3, chunk, stride, window);
            run<1, 6, 512>("vgpr", src, out, 1, chunk, stride, window);
            run<1, 9, 512>("vgpr", src, out, 1, chunk, stride, window);
        }
    }
    return 0;
}
