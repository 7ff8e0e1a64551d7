// Device probe for the fp16 hi / lo arithmetic (common.h "h2"): (1) radet_split2 against a host restatement (rounding mode,
// denormals), (2) what v_mfma_f32_32x32x16_f16 does with denormal operands, (3) sustained rate + shader clock of the f16
// MFMA on random operands next to the bf16 one (the 3-product scheme issues half the MFMAs of the 6-product one: is an
// f16 MFMA as fast, and does it draw the same power?).
//   hipcc --offload-arch=gfx950 -O3 -I radet_amd/csrc tools/micro/h2_probe.hip -o tools/_probe/h2_probe
#include "common.h"
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));

__global__ void split_kernel(const float* x, unsigned* hi, unsigned* lo, int n, float s, float s2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) radet_split2(x[2 * i], x[2 * i + 1], s, s2, hi[i], lo[i]);
}

// one 32x32x16 product: A row i = a[i][0..15], B col j = b[j][0..15] (lane (i, h) holds k = 8h..8h+7)
__global__ void mfma_kernel(const _Float16* a, const _Float16* b, float* c) {
    const int lane = threadIdx.x, i = lane & 31, h = lane >> 5;
    f16x8 av, bv;
    for (int k = 0; k < 8; ++k) { av[k] = a[i * 16 + 8 * h + k]; bv[k] = b[i * 16 + 8 * h + k]; }
    f32x16 acc;
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(av, bv, acc, 0, 0, 0);
    for (int r = 0; r < 16; ++r) c[((r & 3) + 8 * (r >> 2) + 4 * h) * 32 + i] = acc[r];
}

template <int F16, int CH>
__global__ __launch_bounds__(256) void chain_rnd(float* out, int iters, unsigned seed, long long* clk) {
    f32x16 acc[CH];
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
    u32x4 a[8], b[8];
    unsigned x = seed * 2654435761u + threadIdx.x * 40503u + blockIdx.x;
    for (int k = 0; k < 8; ++k)
        for (int q = 0; q < 4; ++q) {
            // values in [0.5, 1) with random sign and mantissa: bf16 0x3F00 | 7 bits, f16 0x3800 | 10 bits
            x = x * 1664525u + 1013904223u; a[k][q] = F16 ? ((x & 0x83FF83FFu) | 0x38003800u) : ((x & 0x807F807Fu) | 0x3F003F00u);
            x = x * 1664525u + 1013904223u; b[k][q] = F16 ? ((x & 0x83FF83FFu) | 0x38003800u) : ((x & 0x807F807Fu) | 0x3F003F00u);
        }
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int u = 0; u < 24 / CH; ++u)
#pragma unroll
            for (int c = 0; c < CH; ++c) {
                if constexpr (F16)
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, a[(u * CH + c) & 7]),
                                                                    __builtin_bit_cast(f16x8, b[(u * CH + c + 3) & 7]), acc[c], 0, 0, 0);
                else
                    acc[c] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, a[(u * CH + c) & 7]),
                                                                     __builtin_bit_cast(bf16x8, b[(u * CH + c + 3) & 7]), acc[c], 0, 0, 0);
            }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int c = 0; c < CH; ++c)
        for (int r = 0; r < 16; ++r) s += acc[c][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (blockIdx.x == 0 && threadIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int F16, int CH>
static void run_rnd(int w) {
    float* out; long long* clk;
    const int grid = 256 * w, iters = 20000;
    hipMalloc(&out, grid * 256 * sizeof(float)); hipMalloc(&clk, 16);
    hipEvent_t s, e; hipEventCreate(&s); hipEventCreate(&e);
    chain_rnd<F16, CH><<<grid, 256>>>(out, 100, 1, clk);
    hipDeviceSynchronize();
    float best = 1e9;
    for (int r = 0; r < 3; ++r) {
        hipEventRecord(s);
        chain_rnd<F16, CH><<<grid, 256>>>(out, iters, 1, clk);
        hipEventRecord(e); hipEventSynchronize(e);
        float ms; hipEventElapsedTime(&ms, s, e);
        if (ms < best) best = ms;
    }
    long long h[2]; hipMemcpy(h, clk, 16, hipMemcpyDeviceToHost);
    const double fl = (double)grid * 4 * iters * 24 * (2.0 * 32 * 32 * 16);
    printf("%s 32x32x16, random operands: chains=%d waves/SIMD=%d: %.2f ms  %.0f TFLOP/s  s_memtime/s_memrealtime = %.2f\n",
           F16 ? "f16 " : "bf16", CH, w, best, fl / best / 1e9, (double)h[0] / h[1]);
    hipFree(out); hipFree(clk);
}

static void host_split(float x, float s, float s2, _Float16& hi, _Float16& lo) {
    const float t = x * s;
    hi = (_Float16)t;                                             // x86 F16C / soft-float: round to nearest even, denormals kept
    const double r = (double)x * (double)s2 - 2048.0 * (double)(float)hi;   // exact
    lo = (_Float16)r;
}

int main() {
    // ---- (1) split
    std::vector<float> x;
    unsigned seed = 12345u;
    auto rnd = [&]() { seed = seed * 1664525u + 1013904223u; return seed; };
    for (int e = -40; e <= 14; ++e)
        for (int k = 0; k < 512; ++k) {
            const float m = 1.0f + (float)(rnd() >> 9) / 8388608.0f;           // [1, 2), 23 random bits
            x.push_back(std::ldexp((rnd() & 1) ? m : -m, e));
        }
    // ties and edge values
    const float edge[] = {0.f, -0.f, 1.0f, 1.0f + 1.0f / 2048.f, 1.0f + 3.0f / 2048.f, 32767.99f, 32752.0f, 32760.0f, 6.1035156e-5f,
                          6.0e-5f, 5.9604645e-8f, 2.9802322e-8f, 1e-9f, -32767.0f};
    for (float v : edge) x.push_back(v);
    if (x.size() & 1) x.push_back(0.f);
    const int n = (int)x.size() / 2;
    float* dx; unsigned *dh, *dl;
    hipMalloc(&dx, x.size() * 4); hipMalloc(&dh, n * 4); hipMalloc(&dl, n * 4);
    hipMemcpy(dx, x.data(), x.size() * 4, hipMemcpyHostToDevice);
    split_kernel<<<(n + 255) / 256, 256>>>(dx, dh, dl, n, 1.0f, 2048.0f);
    std::vector<unsigned> hh(n), hl(n);
    hipMemcpy(hh.data(), dh, n * 4, hipMemcpyDeviceToHost);
    hipMemcpy(hl.data(), dl, n * 4, hipMemcpyDeviceToHost);
    int bad_hi = 0, bad_lo = 0;
    double worst_rel = 0, worst_rel_normal = 0;
    for (size_t i = 0; i < x.size(); ++i) {
        const unsigned short gh = (unsigned short)(i & 1 ? hh[i / 2] >> 16 : hh[i / 2] & 0xFFFF);
        const unsigned short gl = (unsigned short)(i & 1 ? hl[i / 2] >> 16 : hl[i / 2] & 0xFFFF);
        _Float16 eh, el;
        host_split(x[i], 1.0f, 2048.0f, eh, el);
        unsigned short ehb, elb;
        memcpy(&ehb, &eh, 2); memcpy(&elb, &el, 2);
        if (ehb != gh && !(x[i] == 0.f)) { if (bad_hi < 5) printf("hi mismatch x=%a dev=%04x host=%04x\n", x[i], gh, ehb); ++bad_hi; }
        if (elb != gl && !(x[i] == 0.f)) { if (bad_lo < 5) printf("lo mismatch x=%a dev=%04x host=%04x\n", x[i], gl, elb); ++bad_lo; }
        _Float16 fh, fl; memcpy(&fh, &gh, 2); memcpy(&fl, &gl, 2);
        const double v = (double)(float)fh + (double)(float)fl / 2048.0;
        if (x[i] != 0.f) {
            const double rel = std::fabs(v - (double)x[i]) / std::fabs((double)x[i]);
            if (rel > worst_rel) worst_rel = rel;
            if (std::fabs(x[i]) >= 6.1035156e-5f && rel > worst_rel_normal) worst_rel_normal = rel;
        }
    }
    printf("split: %zu values, hi mismatches %d, lo mismatches %d, worst relative error %.3g (2^%.1f); hi normal: %.3g (2^%.1f)\n",
           x.size(), bad_hi, bad_lo, worst_rel, std::log2(worst_rel), worst_rel_normal, std::log2(worst_rel_normal));

    // ---- (2) MFMA with denormal operands: a[i][k] = 2^-20 (f16 denormal), b = 2^10 -> every c = 16 * 2^-10 if kept, 0 if flushed
    std::vector<_Float16> ha(32 * 16), hb(32 * 16);
    for (auto& v : ha) v = (_Float16)9.5367431640625e-07f;     // 2^-20
    for (auto& v : hb) v = (_Float16)1024.0f;
    _Float16 *da, *db; float* dc;
    hipMalloc(&da, 1024); hipMalloc(&db, 1024); hipMalloc(&dc, 4096);
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(da, db, dc);
    std::vector<float> hc(1024);
    hipMemcpy(hc.data(), dc, 4096, hipMemcpyDeviceToHost);
    printf("mfma f16, denormal A (2^-20) x 2^10, K = 16: c[0][0] = %g (kept: %g, flushed: 0)\n", hc[0], 16 * std::ldexp(1.0, -10));
    // products near the fp32 denormal range and large sums: 16 * (2^14)^2 = 2^32
    for (auto& v : ha) v = (_Float16)16384.0f;
    for (auto& v : hb) v = (_Float16)16384.0f;
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(da, db, dc);
    hipMemcpy(hc.data(), dc, 4096, hipMemcpyDeviceToHost);
    printf("mfma f16, 2^14 x 2^14, K = 16: c[0][0] = %g (expected %g)\n", hc[0], 16 * std::ldexp(1.0, 28));
    // exactness of a 16-term product sum: integers with 11-bit operands, alternating signs
    for (int i = 0; i < 32; ++i)
        for (int k = 0; k < 16; ++k) { ha[i * 16 + k] = (_Float16)(float)(1025 + 31 * k + i); hb[i * 16 + k] = (_Float16)(float)((k & 1 ? -1 : 1) * (2047 - 17 * k - i)); }
    hipMemcpy(da, ha.data(), 1024, hipMemcpyHostToDevice); hipMemcpy(db, hb.data(), 1024, hipMemcpyHostToDevice);
    mfma_kernel<<<1, 64>>>(da, db, dc);
    hipMemcpy(hc.data(), dc, 4096, hipMemcpyDeviceToHost);
    int inexact = 0;
    for (int i = 0; i < 32; ++i)
        for (int j = 0; j < 32; ++j) {
            double ref = 0;
            for (int k = 0; k < 16; ++k) ref += (double)(float)ha[i * 16 + k] * (double)(float)hb[j * 16 + k];
            if ((double)hc[i * 32 + j] != (double)(float)ref) ++inexact;
        }
    printf("mfma f16, 16-term sums of 22-bit products: %d of 1024 differ from the correctly rounded sum\n", inexact);

    // ---- (3) rates
    for (int w = 1; w <= 2; ++w) { run_rnd<0, 4>(w); run_rnd<1, 4>(w); }
    run_rnd<0, 2>(2); run_rnd<1, 2>(2);
    return 0;
}
