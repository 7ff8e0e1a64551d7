// 3x3 / stride 1 / pad 1 convolution on plane operands from an LDS PATCH (the head towers: forward and dgrad).
//
// Why: the implicit-GEMM kernels stage one (pixel row, 32-channel) slice per tap, i.e. every input row travels L2 -> LDS
// nine times per output-column tile.  Measured with the MFMAs switched off, that fill traffic alone takes as long as the
// MFMAs (tools/sf_ablate.py: 2.8 GB per launch at ~7 TB/s out of L2 / Infinity Cache for 81 MB of tensors): the plateau of
// every large 3x3 GEMM here (0.46-0.50 of the bf16 pipe) is the fill path, not the matrix cores.  This kernel keeps the
// input PATCH of its output pixels in LDS and reads the nine taps as nine shifted fragment reads:
//
//   * a workgroup owns 4 blocks of 4 x 8 output pixels (any level / image of the row-concatenated pyramid: a block table
//     built once per geometry, 96.6 % of the MFMA rows are real pixels at 640 x 480) x 128 output channels; 4 waves = 2 block
//     pairs x 2 column halves, 2 x 2 accumulator blocks of 32 x 32 per wave;
//   * K runs over 16-channel sub-chunks; per sub-chunk the 6 x 10 pixel patch of every block goes global -> registers ->
//     LDS once ([block][plane][row][pixel][2 x 8 channels], 23 KiB, double buffered, fetched a whole sub-chunk = 9 taps
//     ahead) -- 1.9 fetches per input row instead of 9 x (Cout / 128);
//   * per (sub-chunk, tap) stage the 128 x 16-channel weight slice (12 KiB, L2 resident: 3.5 MB per conv) goes through a
//     two-deep register ring into a two-deep LDS ring, three stages ahead of its use;
//   * per stage and wave: 12 ds_read_b128 (A: 2 blocks x 3 planes at the tap's offset, B: 2 x 3) + 24
//     v_mfma_f32_32x32x16_bf16 (the 6 plane products of a fp32-accurate product), one barrier.
//   LDS 69 KiB -> two workgroups per CU.  16-byte slots are XOR-swizzled by the row parity (patch) / bit 3 of the output
//   channel (weights) so that all fragment reads are bank-conflict free.
// Arithmetic and results: the same six plane products per K = 16 slice as the plane-operand implicit GEMM (TAG bit 4), K
// order channel-major then tap -- the accumulation order differs from the implicit GEMM's only in rounding.
#include "conv_common.h"

#ifndef RADET_PATCH_ABL
#define RADET_PATCH_ABL 0     // experiments (compile time): 1 no loader work in the loop, 2 no MFMAs, 4 fragment reads of stage 0 only, 8 no barriers
#endif

struct PatchBlock { int base_row, H, W, yx; };   // 4 x 8 output pixels at (y0 = yx >> 16, x0 = yx & 0xFFFF) of one (level, image)

struct PatchArgs {
    const char* x;        // plane rows [*][3 Cin] bf16
    const char* w;        // plane rows [(n * 9 + tap)][3 Cin] bf16
    const float* bias;    // [Cout] or null
    const float* addend;  // [rows][Cout] or null
    float* y;             // [rows][Cout]
    const PatchBlock* blocks;
    int nblocks, Cin, Cout, flip;     // flip: weight tap of spatial tap s is 8 - s (dgrad of a 3x3 / 1 conv)
};

static __device__ __attribute__((aligned(16))) unsigned char radet_patch_zero[64];

__global__ __launch_bounds__(256, 2) void conv3x3_patch_kernel(const PatchArgs a) {
    constexpr int NB = 4, BN = 128;
    constexpr int P_SLOTS = NB * 3 * 60 * 2;               // 1440 16-byte slots per patch buffer
    constexpr int P_BYTES = P_SLOTS * 16, W_BYTES = 3 * BN * 2 * 16;
    constexpr int NPS = (P_SLOTS + 255) / 256, NWS = 3;
    __shared__ __attribute__((aligned(16))) unsigned char Ps[2][P_BYTES];
    __shared__ __attribute__((aligned(16))) unsigned char Ws[2][W_BYTES];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int n0 = blockIdx.y * BN;
    const int b0 = blockIdx.x * NB;
    const size_t xrow = (size_t)6 * a.Cin;                 // bytes per plane row
    const int nsub = a.Cin / 16;

    // ---- loader side
    const char* psrc[NPS];
    int pmul[NPS];
#pragma unroll
    for (int q = 0; q < NPS; ++q) {
        const int s = tid + 256 * q;
        psrc[q] = reinterpret_cast<const char*>(radet_patch_zero);
        pmul[q] = 0;
        if (s < P_SLOTS) {
            const int hs = s & 1, t1 = s >> 1;
            const int px = t1 % 10, t2 = t1 / 10;
            const int r = t2 % 6, t3 = t2 / 6;
            const int pl = t3 % 3, b = t3 / 3;
            if (b0 + b < a.nblocks) {
                const PatchBlock B = a.blocks[b0 + b];
                const int iy = (B.yx >> 16) + r - 1, ix = (B.yx & 0xFFFF) + px - 1;
                if (iy >= 0 && iy < B.H && ix >= 0 && ix < B.W) {
                    psrc[q] = a.x + (size_t)(B.base_row + iy * B.W + ix) * xrow + pl * 64 + (hs ^ (r & 1)) * 16;
                    pmul[q] = 1;
                }
            }
        }
    }
    const char* wsrc[NWS];
    int wmul[NWS];
#pragma unroll
    for (int q = 0; q < NWS; ++q) {
        const int s = tid + 256 * q;
        const int hs = s & 1, n = (s >> 1) & (BN - 1), pl = s >> 8;
        const bool ok = n0 + n < a.Cout;
        wsrc[q] = ok ? a.w + (size_t)(n0 + n) * 9 * xrow + pl * 64 + (hs ^ ((n >> 3) & 1)) * 16
                     : reinterpret_cast<const char*>(radet_patch_zero);
        wmul[q] = ok ? 1 : 0;
    }
    typedef unsigned u32x4v __attribute__((ext_vector_type(4)));
    constexpr int RING = 4;                                // register ring of weight stages: loads run RING + 1 stages ahead of their use
    u32x4v pr[NPS], wr[RING][NWS];
    auto load_patch = [&](int c16) {
        const int off = (c16 >> 1) * 192 + (c16 & 1) * 32;
#pragma unroll
        for (int q = 0; q < NPS; ++q) pr[q] = *reinterpret_cast<const u32x4v*>(psrc[q] + pmul[q] * off);
    };
    auto write_patch = [&](int buf) {
#pragma unroll
        for (int q = 0; q < NPS; ++q)
            if (tid + 256 * q < P_SLOTS) *reinterpret_cast<u32x4v*>(&Ps[buf][(tid + 256 * q) * 16]) = pr[q];
    };
    auto load_w = [&](auto rc, int g) {                    // stage g = (sub-chunk g / 9, tap g % 9) -> register set rc
        constexpr int R = decltype(rc)::value;
        const int c16 = g / 9, tap = g - 9 * c16;
        const int wt = a.flip ? 8 - tap : tap;
        const size_t off = (size_t)wt * xrow + (c16 >> 1) * 192 + (c16 & 1) * 32;
#pragma unroll
        for (int q = 0; q < NWS; ++q) wr[R][q] = *reinterpret_cast<const u32x4v*>(wsrc[q] + wmul[q] * off);
    };
    auto write_w = [&](auto rc, int buf) {
        constexpr int R = decltype(rc)::value;
#pragma unroll
        for (int q = 0; q < NWS; ++q) *reinterpret_cast<u32x4v*>(&Ws[buf][(tid + 256 * q) * 16]) = wr[R][q];
    };

    // ---- reader side
    const int py = li >> 3, pxl = li & 7;
    unsigned aoff[2], boff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) aoff[i] = (unsigned)((((wm * 2 + i) * 3) * 60 + py * 10 + pxl) * 32);
    const unsigned half_e = (unsigned)((lh ^ (py & 1)) * 16), half_o = (unsigned)((lh ^ (py & 1) ^ 1) * 16);
#pragma unroll
    for (int j = 0; j < 2; ++j) boff[j] = (unsigned)(((((wn * 2 + j) * 32 + li) * 2) + (lh ^ ((li >> 3) & 1))) * 16);

    f32x16 acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    const int nstage = 9 * nsub;
    bf16x8 fa[3][2], fb[3][2];
    std::integral_constant<int, 0> R0;
    std::integral_constant<int, 1> R1;
    // prologue: patch 0 and weight stage 0 in LDS; weight stages 1 .. RING in flight in the register ring
    load_patch(0);
    load_w(R0, 0);
    write_patch(0);
    write_w(R0, 0);
    static_for<1, RING + 1>([&](auto gc) { load_w(std::integral_constant<int, decltype(gc)::value % RING>{}, decltype(gc)::value); });
    __syncthreads();

    // one (sub-chunk, tap) stage; G4 = stage index g mod 4 (compile time: the loop below is unrolled over four sub-chunks = 36
    // stages), LDS weight buffer g & 1, register-ring slot of stage g + 1 = (g + 1) % RING
    auto stage = [&](auto g4c, auto tapc, int g, int c16) {
        constexpr int G4 = decltype(g4c)::value, TAP = decltype(tapc)::value;
        constexpr int PAR = G4 & 1, SLOT = (G4 + 1) % RING;
        constexpr int DY = TAP / 3, DX = TAP % 3;
        if constexpr (!(RADET_PATCH_ABL & 1)) {
        // weight stage g + 1 (loaded RING stages ago) -> the LDS buffer stage g - 1 released; stage g + 1 + RING into its registers
        if (g + 1 < nstage) write_w(std::integral_constant<int, SLOT>{}, PAR ^ 1);
        if (g + 1 + RING < nstage) load_w(std::integral_constant<int, SLOT>{}, g + 1 + RING);
        if constexpr (TAP == 0) {
            if (c16 + 1 < nsub) load_patch(c16 + 1);       // next sub-chunk's patch: nine stages to arrive
        }
        if constexpr (TAP == 5) {
            if (c16 + 1 < nsub) write_patch((c16 + 1) & 1);
        }
        }
        const unsigned char* Pb = &Ps[c16 & 1][0];
        const unsigned char* Wb = &Ws[PAR][0];
        if (!(RADET_PATCH_ABL & 4) || g == 0)
#pragma unroll
        for (int p = 0; p < 3; ++p) {
#pragma unroll
            for (int i = 0; i < 2; ++i)
                fa[p][i] = *reinterpret_cast<const bf16x8*>(Pb + aoff[i] + p * 60 * 32 + (DY * 10 + DX) * 32 + ((DY & 1) ? half_o : half_e));
#pragma unroll
            for (int j = 0; j < 2; ++j) fb[p][j] = *reinterpret_cast<const bf16x8*>(Wb + boff[j] + p * BN * 32);
        }
#pragma unroll
        for (int t = 0; t < 6; ++t) {                      // terms: hi hi, mid hi, hi mid, mid mid, lo hi, hi lo
            const int pa = (t == 1 || t == 3) ? 1 : (t == 4 ? 2 : 0), pb = (t == 2 || t == 3) ? 1 : (t == 5 ? 2 : 0);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    if constexpr (!(RADET_PATCH_ABL & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[pa][i], fb[pb][j], acc[i][j], 0, 0, 0);
                    else asm volatile("" :: "v"(fa[pa][i]), "v"(fb[pb][j]));
        }
        if constexpr (!(RADET_PATCH_ABL & 8)) __syncthreads();
    };
    for (int c16 = 0; c16 < nsub; c16 += 4) {
        static_for<0, 36>([&](auto sc) {
            constexpr int S = decltype(sc)::value;          // stage within the group of four sub-chunks
            stage(std::integral_constant<int, S & 3>{}, std::integral_constant<int, S % 9>{}, 9 * c16 + S, c16 + S / 9);
        });
    }

    // ---- epilogue: D layout col = lane & 31 (output channel), row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5) (pixel of the block)
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int b = b0 + wm * 2 + i;
        if (b >= a.nblocks) continue;
        const PatchBlock B = a.blocks[b];
        const int y0 = B.yx >> 16, x0 = B.yx & 0xFFFF;
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + (wn * 2 + j) * 32 + li;
            const bool cok = col < a.Cout;
            const float bv = (a.bias && cok) ? a.bias[col] : 0.f;
            size_t o[16];
            bool ok[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int p = (r & 3) + 8 * (r >> 2) + 4 * lh;
                const int yy = y0 + (p >> 3), xx = x0 + (p & 7);
                ok[r] = cok && yy < B.H && xx < B.W;
                o[r] = ok[r] ? (size_t)(B.base_row + yy * B.W + xx) * a.Cout + col : 0;
            }
            float av[16];
            if (a.addend) {
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = a.addend[o[r]];
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v = acc[i][j][r] + bv;
                if (a.addend) v += av[r];
                if (ok[r]) a.y[o[r]] = v;
            }
        }
    }
}

extern "C" int radet_conv3x3_patch_p(const void* x, const void* w, const float* bias, const float* addend, float* y,
                                     const int* blocks_dev, int nblocks, int Cin, int Cout, int flip, void* stream) {
    if (Cin % 64 != 0 || Cin < 64 || Cout <= 0 || nblocks <= 0 || blocks_dev == nullptr) return RADET_ERR_ARG;
    PatchArgs a;
    a.x = (const char*)x; a.w = (const char*)w; a.bias = bias; a.addend = addend; a.y = y;
    a.blocks = reinterpret_cast<const PatchBlock*>(blocks_dev);
    a.nblocks = nblocks; a.Cin = Cin; a.Cout = Cout; a.flip = flip;
    hipLaunchKernelGGL(conv3x3_patch_kernel, dim3((nblocks + 3) / 4, (Cout + 127) / 128), dim3(256), 0, (hipStream_t)stream, a);
    return radet_check_launch();
}
