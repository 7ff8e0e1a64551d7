// Stand-in for an 8-rank RCCL all-reduce of one gradient bucket on ONE GPU (tools/comm_emulation.py): `wgs` workgroups stay
// resident for `target_us` (what the ring needs at the assumed bus bandwidth) and move what a rank's share of the ring moves
// through HBM -- 2 * 7/8 of the bucket read and as much written.  Not part of the product library.
//   hipcc --offload-arch=gfx950 -O3 -shared -fPIC tools/micro/comm_standin.hip -o tools/_probe/libcomm_standin.so
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void standin_kernel(float4* buf, size_t n4, size_t half4, long long target_ticks) {
    const long long t0 = __builtin_amdgcn_s_memrealtime();            // 100 MHz
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t b = blockIdx.x * per, e = b + per < n4 ? b + per : n4;
    for (size_t i = b + threadIdx.x; i < e; i += 256) {
        float4 v = buf[i];
        v.x += 1.0f;
        buf[half4 + i] = v;                                           // (second half of the scratch buffer)
    }
    while (__builtin_amdgcn_s_memrealtime() - t0 < target_ticks) __builtin_amdgcn_s_sleep(64);
}
extern "C" int comm_standin(void* buf, size_t bytes, int wgs, double target_us, void* stream) {
    const size_t n4 = bytes / 16;
    hipLaunchKernelGGL(standin_kernel, dim3(wgs), dim3(256), 0, (hipStream_t)stream, (float4*)buf, n4, n4, (long long)(target_us * 100.0));
    return hipGetLastError() == hipSuccess ? 0 : -2;
}
