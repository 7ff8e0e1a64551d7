// Per-launch floor of back-to-back kernels in one stream, and the cost of dependent global-load round trips.
#include <hip/hip_runtime.h>
#include <cstdio>
__global__ void empty_k() {}
template <int DEPTH>
__global__ void chain_k(const int* __restrict__ tab, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    int v = i % n;
#pragma unroll
    for (int d = 0; d < DEPTH; ++d) v = tab[v];            // dependent loads (table of indices)
    out[i] = (float)v;
}
template <class F>
float time_it(F f, int reps = 50) {
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    for (int i = 0; i < 10; ++i) f();
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 5; ++r) {
        hipEventRecord(s);
        for (int i = 0; i < reps; ++i) f();
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms / reps < best) best = ms / reps;
    }
    return best * 1e3f;
}
int main() {
    const int n = 64 << 20;                                   // 256 MiB of indices: cold in L2 for most touches
    int* tab; float* out;
    hipMalloc(&tab, (size_t)n * 4); hipMalloc(&out, (size_t)n * 4);
    int* h = (int*)malloc((size_t)n * 4);
    for (int i = 0; i < n; ++i) h[i] = (int)(((long)i * 1000003L + 12345) % n);
    hipMemcpy(tab, h, (size_t)n * 4, hipMemcpyHostToDevice);
    printf("empty kernel, 1 WG:        %.2f us\n", time_it([&] { empty_k<<<1, 64>>>(); }));
    printf("empty kernel, 1200 WGs:    %.2f us\n", time_it([&] { empty_k<<<1200, 256>>>(); }));
    printf("store only, 300 WGs:       %.2f us\n", time_it([&] { chain_k<0><<<300, 256>>>(tab, out, n); }));
    printf("1 dependent load, 300 WGs: %.2f us\n", time_it([&] { chain_k<1><<<300, 256>>>(tab, out, n); }));
    printf("2 dependent loads:         %.2f us\n", time_it([&] { chain_k<2><<<300, 256>>>(tab, out, n); }));
    printf("4 dependent loads:         %.2f us\n", time_it([&] { chain_k<4><<<300, 256>>>(tab, out, n); }));
    printf("8 dependent loads:         %.2f us\n", time_it([&] { chain_k<8><<<300, 256>>>(tab, out, n); }));
    return 0;
}
