#include <hip/hip_runtime.h>
#include <stdio.h>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(short* out) {
    __shared__ short lds[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) lds[i] = (short)i;
    __syncthreads();
    // each lane points at 4 consecutive shorts: lane l -> elements 4l..4l+3
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)(lds + threadIdx.x * 4));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = v[j];
}
int main() {
    short* d; hipMalloc(&d, 256 * 2);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d);
    short h[256]; hipMemcpy(h, d, 512, hipMemcpyDeviceToHost);
    for (int l = 0; l < 64; ++l) printf("lane %2d: %4d %4d %4d %4d\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    return 0;
}
