// Optimiser step on flat fp32 arenas (HBM-streaming, 16-byte accesses):
//   global L2 norm (two-stage deterministic reduction) -> clip coefficient -> AdamW update, fused.
// Replaces torch.nn.utils.clip_grad_norm_ + torch.optim.AdamW as driven by mmcv's OptimizerHook
// (configs/base/default_runtime.py:1-19; radet/apis/train.py:87-126 in the reference).
#include "common.h"
#include "../../include/radet_hip.h"

__global__ __launch_bounds__(256) void sqnorm_kernel(const float* __restrict__ g, size_t n, float* __restrict__ partials) {
    double acc = 0.0;
    const size_t n4 = n / 4;
    const float4* g4 = reinterpret_cast<const float4*>(g);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        const float4 v = g4[i];
        acc += (double)((v.x * v.x + v.y * v.y) + (v.z * v.z + v.w * v.w));
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) acc += (double)(g[i] * g[i]);
    __shared__ double red[4];
    acc = wave_sum_d(acc);
    if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) partials[blockIdx.x] = (float)((red[0] + red[1]) + (red[2] + red[3]));
}

extern "C" int radet_sqnorm_partials(const float* g, size_t n, float* partials, int npartials, void* stream) {
    if (npartials < 1 || npartials > 4096) return RADET_ERR_ARG;
    hipLaunchKernelGGL(sqnorm_kernel, dim3(npartials), dim3(256), 0, (hipStream_t)stream, g, n, partials);
    return radet_check_launch();
}

__global__ __launch_bounds__(256) void adamw_kernel(float* __restrict__ p, const float* __restrict__ g,
                                                    float* __restrict__ m, float* __restrict__ v, size_t n, float lr,
                                                    float b1, float b2, float eps, float wd, float bc1, float bc2_sqrt,
                                                    float max_norm, float grad_div, const float* __restrict__ partials,
                                                    int npartials, float* __restrict__ grad_norm_out) {
    __shared__ float s_coef;
    if (threadIdx.x < 64) {
        double a = 0.0;
        for (int k = threadIdx.x; k < npartials; k += 64) a += (double)partials[k];
        a = wave_sum_d(a);
        if (threadIdx.x == 0) {
            const float norm = (float)sqrt(a) / grad_div;
            float coef = 1.f;
            if (max_norm > 0.f) {
                coef = max_norm / (norm + 1e-6f);
                if (coef > 1.f) coef = 1.f;
            }
            s_coef = coef / grad_div;
            if (blockIdx.x == 0 && grad_norm_out) *grad_norm_out = norm;
        }
    }
    __syncthreads();
    const float coef = s_coef;
    const float step = lr / bc1;
    const float decay = 1.f - lr * wd;
    const size_t n4 = n / 4;
    float4* p4 = reinterpret_cast<float4*>(p);
    const float4* g4 = reinterpret_cast<const float4*>(g);
    float4* m4 = reinterpret_cast<float4*>(m);
    float4* v4 = reinterpret_cast<float4*>(v);
#define ADAMW_ONE(P, G, M, V)                                 \
    {                                                         \
        const float gg = (G) * coef;                          \
        (M) = b1 * (M) + (1.f - b1) * gg;                     \
        (V) = b2 * (V) + (1.f - b2) * gg * gg;                \
        const float den = sqrtf(V) / bc2_sqrt + eps;          \
        (P) = (P) * decay - step * ((M) / den);               \
    }
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += (size_t)gridDim.x * 256) {
        float4 pp = p4[i], mm = m4[i], vv = v4[i];
        const float4 gq = g4[i];
        ADAMW_ONE(pp.x, gq.x, mm.x, vv.x)
        ADAMW_ONE(pp.y, gq.y, mm.y, vv.y)
        ADAMW_ONE(pp.z, gq.z, mm.z, vv.z)
        ADAMW_ONE(pp.w, gq.w, mm.w, vv.w)
        p4[i] = pp; m4[i] = mm; v4[i] = vv;
    }
    if (blockIdx.x == 0)
        for (size_t i = n4 * 4 + threadIdx.x; i < n; i += 256) {
            float pp = p[i], mm = m[i], vv = v[i];
            ADAMW_ONE(pp, g[i], mm, vv)
            p[i] = pp; m[i] = mm; v[i] = vv;
        }
}

extern "C" int radet_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1,
                                float beta2, float eps, float weight_decay, int step, float max_norm, float grad_div,
                                const float* partials, int npartials, float* grad_norm_out, void* stream) {
    if (step < 1 || grad_div <= 0.f) return RADET_ERR_ARG;
    const float bc1 = 1.f - powf(beta1, (float)step);
    const float bc2 = 1.f - powf(beta2, (float)step);
    const size_t n4 = n / 4;
    int blocks = (int)((n4 + 255) / 256);
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(adamw_kernel, dim3(blocks), dim3(256), 0, (hipStream_t)stream, p, g, m, v, n, lr, beta1, beta2, eps,
                       weight_decay, bc1, sqrtf(bc2), max_norm, grad_div, partials, npartials, grad_norm_out);
    return radet_check_launch();
}
