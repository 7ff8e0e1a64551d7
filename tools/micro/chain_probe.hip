// What would ONE persistent launch per bottleneck block save over three dependent launches?  (round-4 verdict, item 2.)
// A chain of dependent phases -- every workgroup reads a 16-KiB slice another workgroup (another XCD) wrote in the phase before,
// does `iters` x 8 MFMAs, writes its own slice -- run (A) as one kernel launch per phase on one stream, (B) as ONE launch whose
// workgroups stay resident and meet at a grid-wide barrier between phases: agent-scope release (L2 write-back) + arrival counter
// + spin + agent-scope acquire (L2 invalidate), i.e. the cache work of a kernel boundary without the dispatch.  Results are
// checked (a stale read shows up as a wrong sum).  (A) - (B) per phase is the upper bound of what a persistent block can win per
// removed boundary -- before it pays for anything else (tile loops, co-residency limit of 512 workgroups, ...).
//   hipcc --offload-arch=gfx950 -O3 tools/micro/chain_probe.hip -o tools/_probe/chain_probe && tools/_probe/chain_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

#define SLICE 4096          // floats per workgroup slice (16 KiB)

typedef float f32x4 __attribute__((ext_vector_type(4)));
// WT: the slices travel with system-coherent accesses (sc0 sc1: stores write through the XCD's L2, loads bypass it), so that the
// barrier needs no L2 write-back / invalidate -- what the split-K reduction inside the GEMM launches does for its partial tiles.
template <bool WT>
__device__ __forceinline__ float phase_body(const float* __restrict__ in, float* __restrict__ out, int wg, int nwg, int iters) {
    const int src = (wg * 7 + 3) % nwg;                                   // another workgroup's slice of the previous phase
    const float4* s = reinterpret_cast<const float4*>(in + (size_t)src * SLICE);
    float4 v[4];
    if (WT) {
        f32x4 t[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1" : "=v"(t[i]) : "v"(s + threadIdx.x + 256 * i) : "memory");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = make_float4(t[i][0], t[i][1], t[i][2], t[i][3]);
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i) v[i] = s[threadIdx.x + 256 * i];
    }
    f32x16 acc[2];
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(v[i & 3].x * 0.f + 1.f); b[i] = (_Float16)(v[(i + 1) & 3].y * 0.f); }
    for (int it = 0; it < iters; ++it)
#pragma unroll
        for (int u = 0; u < 8; ++u) acc[u & 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, acc[u & 1], 0, 0, 0);
    float z = 0.f;                                                        // (zero: b is zero; keeps the MFMAs alive)
    for (int c = 0; c < 2; ++c)
        for (int r = 0; r < 16; ++r) z += acc[c][r];
    float4* d = reinterpret_cast<float4*>(out + (size_t)wg * SLICE);
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        float4 w = v[i];
        w.x += 1.f + z; w.y += 1.f; w.z += 1.f; w.w += 1.f;
        if (WT) {
            f32x4 t = {w.x, w.y, w.z, w.w};
            asm volatile("global_store_dwordx4 %0, %1, off sc0 sc1" ::"v"(d + threadIdx.x + 256 * i), "v"(t) : "memory");
        } else {
            d[threadIdx.x + 256 * i] = w;
        }
    }
    if (WT) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");              // the write-through stores have reached memory
    return z;
}
__global__ __launch_bounds__(256) void phase_kernel(const float* in, float* out, int iters) {
    phase_body<false>(in, out, blockIdx.x, gridDim.x, iters);
}
__global__ __launch_bounds__(256) void persistent_wt_kernel(float* buf0, float* buf1, int phases, int iters, unsigned* counter) {
    float* in = buf0;
    float* out = buf1;
    for (int p = 0; p < phases; ++p) {
        phase_body<true>(in, out, blockIdx.x, gridDim.x, iters);
        __syncthreads();
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);          // (no cache maintenance)
            const unsigned want = (unsigned)(p + 1) * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        float* t = in; in = out; out = t;
    }
}
__global__ __launch_bounds__(256) void persistent_kernel(float* buf0, float* buf1, int phases, int iters, unsigned* counter) {
    float* in = buf0;
    float* out = buf1;
    for (int p = 0; p < phases; ++p) {
        phase_body<false>(in, out, blockIdx.x, gridDim.x, iters);
        __syncthreads();                                                  // (all stores of the workgroup issued)
        if (threadIdx.x == 0) {
            __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);          // write-back, then arrive
            const unsigned want = (unsigned)(p + 1) * gridDim.x;
            while (__hip_atomic_load(counter, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < want) __builtin_amdgcn_s_sleep(2);
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                // every wave drops its stale lines
        float* t = in; in = out; out = t;
    }
}

int main() {
    const int phases = 48;
    float *b0, *b1;
    unsigned* counter;
    hipMalloc(&b0, 512 * SLICE * sizeof(float));
    hipMalloc(&b1, 512 * SLICE * sizeof(float));
    hipMalloc(&counter, 4);
    hipEvent_t t0, t1;
    hipEventCreate(&t0); hipEventCreate(&t1);
    std::vector<float> h(512 * SLICE);
    printf("%d dependent phases; per phase every workgroup reads 16 KiB another one wrote, runs iters x 8 MFMAs, writes 16 KiB\n", phases);
    for (int nwg : {300, 512})
        for (int iters : {0, 60, 260, 1000}) {
            double ms[3];
            bool ok[3];
            for (int mode = 0; mode < 3; ++mode) {
                float best = 1e9f;
                for (int rep = 0; rep < 5; ++rep) {
                    hipMemset(b0, 0, 512 * SLICE * sizeof(float));
                    hipMemset(counter, 0, 4);
                    hipDeviceSynchronize();
                    hipEventRecord(t0, 0);
                    if (mode == 0)
                        for (int p = 0; p < phases; ++p) phase_kernel<<<nwg, 256>>>(p & 1 ? b1 : b0, p & 1 ? b0 : b1, iters);
                    else if (mode == 1)
                        persistent_kernel<<<nwg, 256>>>(b0, b1, phases, iters, counter);
                    else
                        persistent_wt_kernel<<<nwg, 256>>>(b0, b1, phases, iters, counter);
                    hipEventRecord(t1, 0);
                    hipDeviceSynchronize();
                    float t;
                    hipEventElapsedTime(&t, t0, t1);
                    best = t < best ? t : best;
                }
                hipMemcpy(h.data(), b0, (size_t)nwg * SLICE * sizeof(float), hipMemcpyDeviceToHost);     // 48 phases: result in b0
                ok[mode] = true;
                for (size_t i = 0; i < (size_t)nwg * SLICE; ++i) ok[mode] &= h[i] == (float)phases;
                ms[mode] = best;
            }
            printf("  %3d workgroups, iters %4d: launch per phase %6.2f us (%s) | persistent, barrier with L2 write-back + invalidate %6.2f us (%s) | "
                   "persistent, write-through stores + L2-bypassing loads, plain barrier %6.2f us (%s)\n",
                   nwg, iters, ms[0] * 1e3 / phases, ok[0] ? "ok" : "WRONG", ms[1] * 1e3 / phases, ok[1] ? "ok" : "WRONG",
                   ms[2] * 1e3 / phases, ok[2] ? "ok" : "WRONG");
        }
    return 0;
}
