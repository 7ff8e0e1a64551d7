// fp32 implicit-GEMM convolution on CDNA4 matrix cores (v_mfma_f32_32x32x2_f32), NHWC.
//
//   forward / dgrad :  Y[m, n] = sum_{tap, c} X[gather(m, tap), c] * W[n, tap, c]   (+ fused epilogue)
//   wgrad           :  dW[o, tap, c] = sum_{m in split} dY[m, o] * X[gather(m, tap), c]   (per-split slabs)
//
// m runs over the rows of a row-concatenated multi-level NHWC buffer (RadetSegs), so the five
// pyramid levels that share the head's weights are ONE launch.  Replaces the cuDNN / torch conv
// calls behind radet/models/backbones/resnet.py:260-299, necks/fpn.py:170-221 and
// dense_heads/atss_head.py:118-145 of the reference.
//
// Tiling: 256 threads = 4 waves; block tile BM x BN, K step 16 or 32 (one tap, 16 / 32 channels).  Tiles go
// global -> LDS by LDS-DMA (global_load_lds_dwordx4) into unpadded rows whose 16-byte slots are XOR-swizzled on the
// source side, so that the 16-byte fragment reads are bank-conflict free.  A lane (i = l&31, h = l>>5) fetches 4
// consecutive k for its row with one ds_read_b128 and feeds 4 MFMAs; the K order inside a step is permuted (lower
// half-wave takes k 0-3 / 8-11, upper 4-7 / 12-15), which is legal because A and B use the same permutation.
// 2 or 3 LDS stages: one barrier per K step, the next step's loads in flight under 8*TM*TN MFMAs.
#include "common.h"
#include "../../include/radet_hip.h"
#include <stdlib.h>
#include <type_traits>

#include "conv_common.h"

#include "conv_igemm_kernel.h"
#include "conv_wgrad_kernel.h"

// ------------------------------------------------------------------------------------------ predictor 3x3 from an LDS patch
// The predictor convs of the head (3x3, 256 -> 21 / 4 / 1 channels over all B * 6400 pyramid positions) are bound by the
// delivery of their A operand in the implicit-GEMM kernel: every pixel row is fetched 9 times (once per tap) into LDS for
// a 32-column tile.  Here a workgroup owns an 8 x 16 block of output pixels of one (level, image) and walks the channels
// in chunks of 16: the 10 x 18 input patch of the chunk (zero outside the image) and the [9][32][16] weight slice go
// global -> LDS once (LDS-DMA, XOR-swizzled 16-byte slots as in the implicit-GEMM kernel), and the nine taps are nine
// fragment reads at shifted patch positions: 1.4 fetches per pixel row instead of 9.  Arithmetic: fp32 operands split
// into three bf16 planes in registers, 6 plane products per K = 16 on the bf16 matrix cores (as TAG bit 3 above).
// Up to two convs of the same input share a launch (reg + iou: output columns [0, c0) -> y0, [c0, c0 + c1) -> y1).
struct PredTile { int base_row, H, W, yx; };      // rows of this (level, image) start at base_row; yx = (tile y << 16) | tile x
struct PredArgs {
    const float* x;
    const float* w[2];        // OHWI [c][9][Cin]
    const float* bias[2];
    float* y[2];              // [rows][c]
    int c[2];
    const PredTile* tiles;
    int Cin;
};

__global__ __launch_bounds__(512) void pred3x3_patch_kernel(const PredArgs a) {
    constexpr int TH = 8, TW = 16, PW = TW + 2, PP = (TH + 2) * PW;          // 180 patch pixels
    constexpr int PROWS = 192, WROWS = 9 * 32;                                // 64-byte rows: patch (padded), weights
    constexpr int A_BYTES = PROWS * 64, BUF_BYTES = (PROWS + WROWS) * 64;     // 12 + 18 KiB per chunk buffer
    constexpr int N_INSTR = (PROWS + WROWS) / 16;                             // 30 wave loads of 1 KiB per chunk
    constexpr int PWL = (N_INSTR + 7) / 8;                                    // <= 4 per wave
    // two chunk buffers (a third, loads two chunks ahead, measured the same at B = 4 and costs the second workgroup per CU)
    __shared__ __attribute__((aligned(64))) float S[2][BUF_BYTES / 4];

    // 8 waves = two per SIMD: wave quad 0 runs taps 0-4, quad 1 taps 5-8 of the same 4 x 32 pixels, so that one wave's
    // operand split (VALU) runs next to the other's MFMA chain (a lone wave runs them one after the other: 670 cycles per
    // tap against 190 of MFMA work); the two partial accumulators are added through LDS at the end
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wq = wave & 3, grp = wave >> 2;
    const int li = lane & 31, lh = lane >> 5;
    const PredTile T = a.tiles[blockIdx.x];
    const int y0 = (T.yx >> 16) * TH, x0 = (T.yx & 0xFFFF) * TW;
    const int c0 = a.c[0], ctot = a.c[0] + a.c[1];

    // writer side: load k of this wave is wave load i = wave + 8k; it fills rows 16 i + lane / 4, slot lane % 4
    const float* src[PWL];
    int inc[PWL];
    const int kq = (lane & 3) ^ ((lane >> 4) & 3);                            // the k-quad this lane's slot holds
#pragma unroll
    for (int k = 0; k < PWL; ++k) {
        const int i = wave + 8 * k;
        const int r = i * 16 + (lane >> 2);
        src[k] = radet_zero_page + lane * 4;
        inc[k] = 0;
        if (i < PROWS / 16) {
            const int py = r / PW, px = r - py * PW;
            const int iy = y0 + py - 1, ix = x0 + px - 1;
            if (r < PP && iy >= 0 && iy < T.H && ix >= 0 && ix < T.W) {
                src[k] = a.x + (size_t)(T.base_row + iy * T.W + ix) * a.Cin + 4 * kq;
                inc[k] = 16;
            }
        } else if (i < N_INSTR) {
            const int rb = r - PROWS, tap = rb >> 5, n = rb & 31;
            if (n < ctot) {
                const int sel = n < c0 ? 0 : 1;
                src[k] = a.w[sel] + (size_t)((n - (sel ? c0 : 0)) * 9 + tap) * a.Cin + 4 * kq;
                inc[k] = 16;
            }
        }
    }
    auto issue = [&](int buf) {
#pragma unroll
        for (int k = 0; k < PWL; ++k) {
            const int i = wave + 8 * k;
            if (i < N_INSTR) {
                __builtin_amdgcn_global_load_lds((gptr_t)src[k], (lptr_t)(&S[buf][i * 256]), 16, 0, 0);
                src[k] += inc[k];
            }
        }
    };

    // reader side: this lane's output pixel is (2 wq + li / 16, li % 16); tap (dy, dx) reads patch pixel + dy * 18 + dx
    const unsigned s_base = (unsigned)(size_t)(lptr_t)(&S[0][0]);
    const int tap0 = grp * 5, ntap = grp ? 4 : 5;
    unsigned aa0[5], aa1[5];
    {
        const int p0 = (2 * wq + (li >> 4)) * PW + (li & 15);
#pragma unroll
        for (int u = 0; u < 5; ++u) {
            const int t = tap0 + (u < ntap ? u : 0);
            const int pp = p0 + (t / 3) * PW + (t % 3);
            const unsigned off = (unsigned)(pp * 64 + (((2 * lh) ^ ((pp >> 2) & 3)) << 4));
            aa0[u] = s_base + off;
            aa1[u] = s_base + (off ^ 16u);
        }
    }
    const unsigned boff = (unsigned)(li * 64 + (((2 * lh) ^ ((li >> 2) & 3)) << 4)) + (unsigned)(tap0 * 2048);
    const unsigned bb0 = s_base + A_BYTES + boff, bb1 = s_base + A_BYTES + (boff ^ 16u);

    f32x16 acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;

    const int nch = a.Cin / 16;
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    auto chunk = [&](auto bufc, int c) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr int BO = BUF * BUF_BYTES;
        if (c + 1 < nch) issue(BUF ^ 1);
        f32x4 fa[2][2], fb[2][2];
        lds_read128<BO>(fa[0][0], aa0[0]);
        lds_read128<BO>(fa[0][1], aa1[0]);
        lds_read128<BO>(fb[0][0], bb0);
        lds_read128<BO>(fb[0][1], bb1);
        static_for<0, 5>([&](auto uc) {
            constexpr int u = decltype(uc)::value;
            constexpr int pp = u & 1;
            if (u < ntap) {                                        // uniform per wave
                if (u + 1 < ntap) {
                    if constexpr (u + 1 < 5) {
                        lds_read128<BO>(fa[pp ^ 1][0], aa0[u + 1]);
                        lds_read128<BO>(fa[pp ^ 1][1], aa1[u + 1]);
                        lds_read128<BO + (u + 1) * 2048>(fb[pp ^ 1][0], bb0);
                        lds_read128<BO + (u + 1) * 2048>(fb[pp ^ 1][1], bb1);
                    }
                    lds_wait<4>();
                } else {
                    lds_wait<0>();
                }
                asm volatile("" : "+v"(fa[pp][0]), "+v"(fa[pp][1]), "+v"(fb[pp][0]), "+v"(fb[pp][1]));
                bf16x8 ah, am, al, bh, bm, bl;
                split3_bf16(fa[pp][0], fa[pp][1], ah, am, al);
                split3_bf16(fb[pp][0], fb[pp][1], bh, bm, bl);
                mfma_x3(acc, ah, am, al, bh, bm, bl);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int c = 0; c < nch; c += 2) {
        chunk(std::integral_constant<int, 0>{}, c);
        if (c + 1 < nch) chunk(std::integral_constant<int, 1>{}, c + 1);
    }

    // the tap groups' partial sums: quad 1 -> LDS -> quad 0 (all loads have landed and been consumed: S is free)
    float* red = &S[0][0];
    if (grp == 1) {
#pragma unroll
        for (int r = 0; r < 16; ++r) red[(wq * 16 + r) * 64 + lane] = acc[r];
    }
    __syncthreads();
    if (grp == 1) return;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] += red[(wq * 16 + r) * 64 + lane];

    // epilogue: accumulator r of this lane = pixel (r & 3) + 8 (r >> 2) + 4 lh of the wave's 32, output column li
    if (li < ctot) {
        const int sel = li < c0 ? 0 : 1;
        const int col = li - (sel ? c0 : 0), cn = a.c[sel];
        const float bv = a.bias[sel] ? a.bias[sel][col] : 0.f;
        float* yo = a.y[sel];
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2) + 4 * lh;
            const int oy = y0 + 2 * wq + (m >> 4), ox = x0 + (m & 15);
            if (oy < T.H && ox < T.W) yo[(size_t)(T.base_row + oy * T.W + ox) * cn + col] = acc[r] + bv;
        }
    }
}

// tiles_dev: [ntiles] PredTile; second conv optional (w1 = null / c1 = 0)
extern "C" int radet_pred3x3_patch(const float* x, int Cin, const int* tiles_dev, int ntiles, const float* w0,
                                   const float* bias0, float* y0, int c0, const float* w1, const float* bias1, float* y1,
                                   int c1, void* stream) {
    if (x == nullptr || tiles_dev == nullptr || w0 == nullptr || y0 == nullptr || Cin <= 0 || Cin % 16 != 0 || c0 < 1 ||
        c1 < 0 || c0 + c1 > 32 || (c1 > 0 && (w1 == nullptr || y1 == nullptr)))
        return RADET_ERR_ARG;
    if (ntiles <= 0) return RADET_OK;
    PredArgs a;
    a.x = x; a.Cin = Cin; a.tiles = reinterpret_cast<const PredTile*>(tiles_dev);
    a.w[0] = w0; a.bias[0] = bias0; a.y[0] = y0; a.c[0] = c0;
    a.w[1] = c1 > 0 ? w1 : w0; a.bias[1] = c1 > 0 ? bias1 : nullptr; a.y[1] = c1 > 0 ? y1 : y0; a.c[1] = c1;
    hipLaunchKernelGGL(pred3x3_patch_kernel, dim3(ntiles), dim3(512), 0, (hipStream_t)stream, a);
    return radet_check_launch();
}

// ------------------------------------------------------------------------------------------ wgrad, bf16 storage
// dW[o, tap, c] = sum_m dy[m, o] * x[g(m, tap), c] with bf16 dy / x in HBM, v_mfma_f32_32x32x16_bf16, fp32 slabs.
// The MFMA wants, per lane, 8 consecutive PIXELS (k) of one channel, but memory is [pixel][channel]: the tiles are
// brought in by LDS-DMA as 128-byte sub-tiles of [4 pixels][16 channels] (lane -> source address is free, so the
// image is built for the read), and the operands are fetched with gfx950's transposing LDS read
// ds_read_b64_tr_b16: within a 16-lane group lane m supplies the address of sub-tile bytes 8m..8m+7 and receives
// column m, i.e. 4 consecutive pixels of its channel; two reads = one MFMA operand (probed on the device:
// tools/_probe, lane l <- elements (l&15) + 16 j + 64 (l>>4)).
typedef short s16x4v __attribute__((ext_vector_type(4)));
typedef short s16x8v __attribute__((ext_vector_type(8)));

template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgradh_kernel(const WgradArgs a) {
    constexpr int BP = 32, NW = 4;                          // pixels per stage (two K = 16 MFMA steps)
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int CBA = BM / 16, CBB = BN / 16;             // 16-channel sub-tile columns
    constexpr int A_INSTR = BP * BM * 2 / 1024, B_INSTR = BP * BN * 2 / 1024;
    constexpr int N_INSTR = A_INSTR + B_INSTR;
    constexpr int PER_WAVE = (N_INSTR + NW - 1) / NW;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BP * BM];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][BP * BN];
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy);
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(a.x);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int KT = a.KH * a.KW;
    const int tilesO = (a.Cout + BM - 1) / BM;
    const int tilesC = (a.Cin + BN - 1) / BN;
    const int tilesPerSplit = tilesO * tilesC * KT;
    int id = blockIdx.x;
    const int split = id / tilesPerSplit;
    id -= split * tilesPerSplit;
    const int to = id % tilesO;
    id /= tilesO;
    const int tc = id % tilesC;
    const int tap = id / tilesC;
    const int o0 = to * BM, c0 = tc * BN;
    const int* tab_tap = a.rowtab ? a.rowtab + (size_t)tap * a.Mp : nullptr;

    const int p_begin = split * a.chunks_per_split * 16;    // chunks_per_split counts 16-pixel chunks
    int p_end = p_begin + a.chunks_per_split * 16;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + BP - 1) / BP : 0;

    // writer side: lane -> (sub-tile, pixel row, 8-channel half) of every wave load it issues
    const int l_blk = lane >> 3, l_prow = (lane & 7) >> 1, l_half = lane & 1;
    int brow[PER_WAVE];                                     // gather rows of the NEXT stage (x-tile loads)
    bool bok[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int bi = wave + k * NW - A_INSTR;
        brow[k] = -1;
        bok[k] = false;
        if (bi >= 0 && bi < B_INSTR) {
            const int blk = bi * 8 + l_blk;
            const int m = p_begin + 4 * (blk / CBB) + l_prow;
            brow[k] = tab_tap ? tab_tap[m < a.Mp ? m : a.Mp - 1] : m;   // unconditional (clamped) load, masked at use
            bok[k] = m < p_end;
        }
    }
    auto issue_stage = [&](int it, int buf) {                 // order: x tiles, dy tiles, next gather rows (see conv_wgradg)
        // the gather rows were fetched one stage ago and drained by the barrier's vmcnt(0), which the compiler cannot see: left
        // alone it puts a vmcnt(0) in front of every x-tile load that reads brow[k] -- and from the second one on that wait
        // covers the LDS-DMA load issued just before it: the pieces of a stage went out one round trip apart.  One wait here
        // (free), and the rows are plain registers afterwards.
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) asm volatile("" : "+v"(brow[k]));
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int blk = bi * 8 + l_blk;
                const int c = c0 + 16 * (blk % CBB) + 8 * l_half;
                radet_lds_load16(xh, (bok[k] && brow[k] >= 0 && c < a.Cin), (size_t)((size_t)brow[k] * a.Cin + c), (lptr_t)(&Bs[buf][bi * 512]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int blk = ins * 8 + l_blk;
                const int m = p0 + 4 * (blk / CBA) + l_prow;
                const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
                radet_lds_load16(dyh, (m < p_end && o < a.Cout), (size_t)((size_t)m * a.ld_dy + o), (lptr_t)(&As[buf][ins * 512]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int blk = (ins - A_INSTR) * 8 + l_blk;
                const int m = p0 + BP + 4 * (blk / CBB) + l_prow;
                brow[k] = tab_tap ? tab_tap[m < a.Mp ? m : a.Mp - 1] : m;
                bok[k] = m < p_end;
            }
        }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    float bsum = 0.f;
    const bool want_bias = a.dbias_partials != nullptr && tap == 0 && tc == 0;

    // reader side (per lane): 16-lane group g16 -> channel sub-tile, m -> bytes 8m of the sub-tile, lh -> pixel half
    const int g16 = (lane >> 4) & 1, m16 = lane & 15;

    // the transposing read is not ordered against in-flight LDS-DMA by the compiler: every wave drains its own
    // loads (vmcnt(0)) before the barrier that publishes the stage
    // per-lane LDS byte addresses of the transposing reads (sub-tile row 2 * lh of a 4-row group, channel sub-tile of the
    // wave tile + g16, bytes 8 * m16 of the sub-tile)
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(((2 * lh) * CBA + wm * TM * 2 + g16) * 128 + m16 * 8);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)(((2 * lh) * CBB + wn * TN * 2 + g16) * 128 + m16 * 8);
    if (nIt > 0) issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < nIt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nIt) issue_stage(it + 1, buf ^ 1);
        // transposing reads as inline asm (+ hand-placed lgkmcnt waits): behind the builtin the compiler waits vmcnt(0)
        // for the LDS-DMA prefetch just issued before the first read of the stage (see conv_wgradg)
        {
            const unsigned ab = a_thr + (unsigned)buf * (BP * BM * 2), bb = b_thr + (unsigned)buf * (BP * BN * 2);
            s16x4v al[2][TM], ah[2][TM], bl[2][TN], bh[2][TN];
            auto read_ks = [&](auto ksc, int pp) {
                constexpr int ks = decltype(ksc)::value;
                static_for<0, TM>([&](auto ic) {
                    constexpr int off = ((4 * ks) * CBA + decltype(ic)::value * 2) * 128;
                    lds_read_tr16<off>(al[pp][decltype(ic)::value], ab);
                    lds_read_tr16<off + CBA * 128>(ah[pp][decltype(ic)::value], ab);
                });
                static_for<0, TN>([&](auto jc) {
                    constexpr int off = ((4 * ks) * CBB + decltype(jc)::value * 2) * 128;
                    lds_read_tr16<off>(bl[pp][decltype(jc)::value], bb);
                    lds_read_tr16<off + CBB * 128>(bh[pp][decltype(jc)::value], bb);
                });
            };
            read_ks(std::integral_constant<int, 0>{}, 0);
            static_for<0, BP / 16>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, pp = ks & 1;
                if constexpr (ks + 1 < BP / 16) {
                    read_ks(std::integral_constant<int, ks + 1>{}, pp ^ 1);
                    lds_wait<2 * (TM + TN)>();
                } else {
                    lds_wait<0>();
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) { asm volatile("" : "+v"(al[pp][i])); asm volatile("" : "+v"(ah[pp][i])); }
#pragma unroll
                for (int j = 0; j < TN; ++j) { asm volatile("" : "+v"(bl[pp][j])); asm volatile("" : "+v"(bh[pp][j])); }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        const s16x8v af = __builtin_shufflevector(al[pp][i], ah[pp][i], 0, 1, 2, 3, 4, 5, 6, 7);
                        const s16x8v bf = __builtin_shufflevector(bl[pp][j], bh[pp][j], 0, 1, 2, 3, 4, 5, 6, 7);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af),
                                                                            __builtin_bit_cast(bf16x8, bf), acc[i][j], 0, 0, 0);
                    }
            });
        }
        if (want_bias && tid < BM) {                        // column sums of dy, pixel order
            const int cb = tid >> 4, cc = tid & 15;
#pragma unroll
            for (int p = 0; p < BP; ++p)
                bsum += (float)reinterpret_cast<const __bf16*>(&As[buf][0])[((p >> 2) * CBA + cb) * 64 + (p & 3) * 16 + cc];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (want_bias && tid < BM && o0 + tid < a.Cout) a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = bsum;
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int c = c0 + (wn * TN + j) * 32 + li;
            if (c >= a.Cin) continue;
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int o = o0 + (wm * TM + i) * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (o >= a.Cout) continue;
                out[((size_t)o * KT + tap) * a.Cin + c] = acc[i][j][t];
            }
        }
}

// ------------------------------------------------------------------------------------------ wgrad, all 9 taps, bf16 storage
// conv_wgrad9g_kernel's tiling (256 output channels x 32 input channels x 9 taps per 8-wave workgroup, the dy tile
// loaded once for all taps) with conv_wgradh_kernel's data path (bf16 [4 pixels][16 channels] sub-tiles by LDS-DMA,
// ds_read_b64_tr_b16 operands, v_mfma_f32_32x32x16_bf16).
__global__ __launch_bounds__(512) void conv_wgrad9h_kernel(const WgradArgs a) {
    constexpr int BP = 32, NW = 8, BM = 256, BC = 32, KT = 9;
    constexpr int CBA = BM / 16, CBB = BC / 16;
    constexpr int A_INSTR = BP * BM * 2 / 1024;             // 16
    constexpr int B_TAP = BP * BC * 2 / 1024;               // 2 wave loads per tap tile
    constexpr int B_INSTR = KT * B_TAP;                     // 18
    constexpr int N_INSTR = A_INSTR + B_INSTR;
    constexpr int PER_WAVE = (N_INSTR + NW - 1) / NW;       // 5
    __shared__ __attribute__((aligned(16))) unsigned short As[2][BP * BM];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][KT * BP * BC];
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy);
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(a.x);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int tilesO = (a.Cout + BM - 1) / BM;
    const int tilesC = a.Cin / BC;
    const int tilesPerSplit = tilesO * tilesC;
    int id = blockIdx.x;
    const int split = id / tilesPerSplit;
    id -= split * tilesPerSplit;
    const int to = id % tilesO, tc = id / tilesO;
    const int o0 = to * BM, c0 = tc * BC;

    const int p_begin = split * a.chunks_per_split * 16;
    int p_end = p_begin + a.chunks_per_split * 16;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + BP - 1) / BP : 0;

    const int l_blk = lane >> 3, l_prow = (lane & 7) >> 1, l_half = lane & 1;
    // x-tile load bi = tap * B_TAP + half: sub-tiles blk = half * 8 + l_blk of the tap's [8 pq][2 cb] grid
    int brow[PER_WAVE];
    bool bok[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int bi = wave + k * NW - A_INSTR;
        brow[k] = -1;
        bok[k] = false;
        if (bi >= 0 && bi < B_INSTR) {
            const int blk = (bi % B_TAP) * 8 + l_blk;
            const int m = p_begin + 4 * (blk / CBB) + l_prow;
            brow[k] = a.rowtab[(size_t)(bi / B_TAP) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
            bok[k] = m < p_end;
        }
    }
    auto issue_stage = [&](int it, int buf) {                // order: x tiles, dy tiles, next gather rows (see conv_wgradg)
        // the gather rows were fetched one stage ago and drained by the barrier's vmcnt(0), which the compiler cannot see: left
        // alone it puts a vmcnt(0) in front of every x-tile load that reads brow[k] -- and from the second one on that wait
        // covers the LDS-DMA load issued just before it: the pieces of a stage went out one round trip apart.  One wait here
        // (free), and the rows are plain registers afterwards.
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) asm volatile("" : "+v"(brow[k]));
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int blk = (bi % B_TAP) * 8 + l_blk;
                const int c = c0 + 16 * (blk % CBB) + 8 * l_half;
                radet_lds_load16(xh, (bok[k] && brow[k] >= 0), (size_t)((size_t)brow[k] * a.Cin + c), (lptr_t)(&Bs[buf][bi * 512]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int blk = ins * 8 + l_blk;
                const int m = p0 + 4 * (blk / CBA) + l_prow;
                const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
                radet_lds_load16(dyh, (m < p_end && o < a.Cout), (size_t)((size_t)m * a.ld_dy + o), (lptr_t)(&As[buf][ins * 512]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int blk = (bi % B_TAP) * 8 + l_blk;
                const int m = p0 + BP + 4 * (blk / CBB) + l_prow;
                brow[k] = a.rowtab[(size_t)(bi / B_TAP) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
                bok[k] = m < p_end;
            }
        }
    };

    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool want_bias = a.dbias_partials != nullptr && tc == 0;
    const int g16 = (lane >> 4) & 1, m16 = lane & 15;
    // per-lane LDS byte addresses of the (inline-asm) transposing reads, see conv_wgradh
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(((2 * lh) * CBA + wave * 2 + g16) * 128 + m16 * 8);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)(((2 * lh) * CBB + g16) * 128 + m16 * 8);

    if (nIt > 0) issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < nIt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nIt) issue_stage(it + 1, buf ^ 1);
        const unsigned ab = a_thr + (unsigned)buf * (BP * BM * 2), bb = b_thr + (unsigned)buf * (KT * BP * BC * 2);
        static_for<0, BP / 16>([&](auto ksc) {
            constexpr int ks = decltype(ksc)::value;
            s16x4v alo, ahi, blo[2], bhi[2];
            lds_read_tr16<(4 * ks) * CBA * 128>(alo, ab);
            lds_read_tr16<(4 * ks) * CBA * 128 + CBA * 128>(ahi, ab);
            lds_read_tr16<(4 * ks) * CBB * 128>(blo[0], bb);
            lds_read_tr16<(4 * ks) * CBB * 128 + CBB * 128>(bhi[0], bb);
            static_for<0, KT>([&](auto tc_) {
                constexpr int t = decltype(tc_)::value, pp = t & 1;
                if constexpr (t + 1 < KT) {
                    lds_read_tr16<(t + 1) * BP * BC * 2 + (4 * ks) * CBB * 128>(blo[pp ^ 1], bb);
                    lds_read_tr16<(t + 1) * BP * BC * 2 + (4 * ks) * CBB * 128 + CBB * 128>(bhi[pp ^ 1], bb);
                    lds_wait<2>();
                } else {
                    lds_wait<0>();
                }
                asm volatile("" : "+v"(alo)); asm volatile("" : "+v"(ahi));
                asm volatile("" : "+v"(blo[pp])); asm volatile("" : "+v"(bhi[pp]));
                const bf16x8 af = __builtin_bit_cast(bf16x8, __builtin_shufflevector(alo, ahi, 0, 1, 2, 3, 4, 5, 6, 7));
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(
                    af, __builtin_bit_cast(bf16x8, __builtin_shufflevector(blo[pp], bhi[pp], 0, 1, 2, 3, 4, 5, 6, 7)), acc[t], 0, 0, 0);
            });
        });
        if (want_bias && tid < BM) {
            const int cb = tid >> 4, cc = tid & 15;
#pragma unroll
            for (int p = 0; p < BP; ++p)
                bsum += (float)reinterpret_cast<const __bf16*>(&As[buf][0])[((p >> 2) * CBA + cb) * 64 + (p & 3) * 16 + cc];
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (want_bias && tid < BM && o0 + tid < a.Cout) a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = bsum;
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
    const int c = c0 + li;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (o < a.Cout) out[((size_t)o * KT + t) * a.Cin + c] = acc[t][r];
        }
}

// ------------------------------------------------------------------------------------------ wgrad, all 9 taps, plane operands
// conv_wgrad9h_kernel's tiling and data path (256 output channels x 32 input channels x 9 taps per 8-wave workgroup,
// [4 pixels][16 channels] bf16 sub-tiles by LDS-DMA, ds_read_b64_tr_b16 operands) for operands that arrive as bf16 plane
// triples: dy rows [3][ld_dy], x rows [3][Cin] (hi | mid | lo, hi + mid + lo = the fp32 value exactly; written once by the
// producers, see radet_split_planes / the GroupNorm kernels).  Per tap the 6 plane products of conv_igemmg_kernel's X3 / P3
// modes; no operand split in the loop.  16 pixels (one K = 16 MFMA step) per stage: 3 planes x (8 KiB dy + 9 KiB x) = 51
// KiB per stage, two stages; 54 MFMAs per wave between barriers, two waves per SIMD.
// (a 4-wave / 128-channel variant, two workgroups per CU, measured 201 vs 184 us on the tower shape: not kept)
__global__ __launch_bounds__(512) void conv_wgrad9p_kernel(const WgradArgs a) {
    constexpr int BP = 16, NW = 8, BM = 256, BC = 32, KT = 9;
    constexpr int CBA = BM / 16, CBB = BC / 16;
    constexpr int A_PL = BP * BM, B_PL = BP * BC;           // bf16 elements per dy plane tile / per (tap, plane) x tile
    constexpr int A_Q = A_PL * 2 / 1024;                    // wave loads per dy plane tile: 8
    constexpr int A_INSTR = 3 * A_Q;                        // 24
    constexpr int B_INSTR = KT * 3;                         // 27: one wave load per (tap, plane)
    constexpr int N_INSTR = A_INSTR + B_INSTR;
    constexpr int PER_WAVE = (N_INSTR + NW - 1) / NW;       // 7
    static_assert(B_PL * 2 == 1024, "one wave load per x tile");
    __shared__ __attribute__((aligned(16))) unsigned short As[2][3 * A_PL];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][KT * 3 * B_PL];
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy);
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(a.x);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane & 31, lh = lane >> 5;

    const int tilesO = (a.Cout + BM - 1) / BM;
    const int tilesC = a.Cin / BC;
    const int tilesPerSplit = tilesO * tilesC;
    // XCD-aware order: the channel tiles of one pixel split run next to each other on ONE XCD, so that its L2 serves the
    // dy rows to all of them (grid order, tc fastest, put channel tile k of EVERY split on XCD k: each XCD then streamed
    // the whole dy tensor -- 4.1x the algorithmic bytes at the fabric, profiles/round3_pmc_hbm_traffic*.txt)
    int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int split = id / tilesPerSplit;
    id -= split * tilesPerSplit;
    const int to = id % tilesO, tc = id / tilesO;
    const int o0 = to * BM, c0 = tc * BC;

    const int p_begin = split * a.chunks_per_split * 16;
    int p_end = p_begin + a.chunks_per_split * 16;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + BP - 1) / BP : 0;

    const int l_blk = lane >> 3, l_prow = (lane & 7) >> 1, l_half = lane & 1;
    // x-tile load bi = tap * 3 + plane: the 8 sub-tiles [4 pixel quads][2 channel blocks] of that tap and plane
    int brow[PER_WAVE];
    bool bok[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int bi = wave + k * NW - A_INSTR;
        brow[k] = -1;
        bok[k] = false;
        if (bi >= 0 && bi < B_INSTR) {
            const int m = p_begin + 4 * (l_blk / CBB) + l_prow;
            brow[k] = a.rowtab[(size_t)(bi / 3) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
            bok[k] = m < p_end;
        }
    }
    auto issue_stage = [&](int it, int buf) {                // order: x tiles, dy tiles, next gather rows (see conv_wgradg)
        // the gather rows were fetched one stage ago and drained by the barrier's vmcnt(0), which the compiler cannot see: left
        // alone it puts a vmcnt(0) in front of every x-tile load that reads brow[k] -- and from the second one on that wait
        // covers the LDS-DMA load issued just before it: the pieces of a stage went out one round trip apart.  One wait here
        // (free), and the rows are plain registers afterwards.
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) asm volatile("" : "+v"(brow[k]));
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int c = c0 + 16 * (l_blk % CBB) + 8 * l_half;
                radet_lds_load16(xh, (bok[k] && brow[k] >= 0), (size_t)((size_t)brow[k] * 3 * a.Cin + radet_plane_off(c) + 32 * (bi % 3)), (lptr_t)(&Bs[buf][bi * B_PL]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int pl = ins / A_Q, blk = (ins % A_Q) * 8 + l_blk;
                const int m = p0 + 4 * (blk / CBA) + l_prow;
                const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
                radet_lds_load16(dyh, (m < p_end && o < a.Cout), (size_t)((size_t)m * 3 * a.ld_dy + radet_plane_off(o) + 32 * pl), (lptr_t)(&As[buf][ins * 512]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int m = p0 + BP + 4 * (l_blk / CBB) + l_prow;
                brow[k] = a.rowtab[(size_t)(bi / 3) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
                bok[k] = m < p_end;
            }
        }
    };

    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;
    const bool want_bias = a.dbias_partials != nullptr && tc == 0;
    const int g16 = (lane >> 4) & 1, m16 = lane & 15;
    // per-lane LDS byte addresses of the (inline-asm) transposing reads, see conv_wgradh: pixel quad 2 lh (+ 1), channel
    // sub-tile of the wave's 32 channels + g16, bytes 8 m16 of the sub-tile
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(((2 * lh) * CBA + wave * 2 + g16) * 128 + m16 * 8);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)(((2 * lh) * CBB + g16) * 128 + m16 * 8);

    if (nIt > 0) issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int it = 0; it < nIt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nIt) issue_stage(it + 1, buf ^ 1);
        const unsigned ab = a_thr + (unsigned)buf * (3 * A_PL * 2), bb = b_thr + (unsigned)buf * (KT * 3 * B_PL * 2);
        s16x4v al[3], ah[3], bl[2][3], bh[2][3];
        static_for<0, 3>([&](auto pc) {
            constexpr int pl = decltype(pc)::value;
            lds_read_tr16<pl * A_PL * 2>(al[pl], ab);
            lds_read_tr16<pl * A_PL * 2 + CBA * 128>(ah[pl], ab);
        });
        static_for<0, 3>([&](auto pc) {
            constexpr int pl = decltype(pc)::value;
            lds_read_tr16<pl * B_PL * 2>(bl[0][pl], bb);
            lds_read_tr16<pl * B_PL * 2 + CBB * 128>(bh[0][pl], bb);
        });
        bf16x8 af[3];
        static_for<0, KT>([&](auto tc_) {
            constexpr int t = decltype(tc_)::value, pp = t & 1;
            if constexpr (t + 1 < KT) {
                static_for<0, 3>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    lds_read_tr16<((t + 1) * 3 + pl) * B_PL * 2>(bl[pp ^ 1][pl], bb);
                    lds_read_tr16<((t + 1) * 3 + pl) * B_PL * 2 + CBB * 128>(bh[pp ^ 1][pl], bb);
                });
                lds_wait<6>();
            } else {
                lds_wait<0>();
            }
            if constexpr (t == 0) {
#pragma unroll
                for (int pl = 0; pl < 3; ++pl) {
                    asm volatile("" : "+v"(al[pl])); asm volatile("" : "+v"(ah[pl]));
                    af[pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(al[pl], ah[pl], 0, 1, 2, 3, 4, 5, 6, 7));
                }
            }
            bf16x8 bf[3];
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
                asm volatile("" : "+v"(bl[pp][pl])); asm volatile("" : "+v"(bh[pp][pl]));
                bf[pl] = __builtin_bit_cast(bf16x8, __builtin_shufflevector(bl[pp][pl], bh[pp][pl], 0, 1, 2, 3, 4, 5, 6, 7));
            }
            mfma_x3(acc[t], af[0], af[1], af[2], bf[0], bf[1], bf[2]);
            __builtin_amdgcn_sched_barrier(0);
        });
        if (want_bias && tid < BM) {                        // column sums of dy, pixel order; (hi + mid) + lo is exact
            const int cb = tid >> 4, cc = tid & 15;
            const __bf16* ap = reinterpret_cast<const __bf16*>(&As[buf][0]);
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                const int e = ((p >> 2) * CBA + cb) * 64 + (p & 3) * 16 + cc;
                bsum += ((float)ap[e] + (float)ap[A_PL + e]) + (float)ap[2 * A_PL + e];
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (want_bias && tid < BM && o0 + tid < a.Cout) a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = bsum;
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
    const int c = c0 + li;
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int o = o0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
            if (o < a.Cout) out[((size_t)o * KT + t) * a.Cin + c] = acc[t][r];
        }
}

// ------------------------------------------------------------------------------------------ gather table
__global__ void gather_table_kernel(int* __restrict__ tab, const RadetSegs segs, int M, int Mp, int KH, int KW, int so,
                                    int sr, int off, int div) {
    const int m = blockIdx.x * blockDim.x + threadIdx.x;
    if (m >= Mp) return;
    const PixCtx pc = decode_pixel(segs, m, M, so, off);
    for (int t = 0; t < KH * KW; ++t) {
        const int r = t / KW, q = t - r * KW;
        tab[(size_t)t * Mp + m] = m < M ? gather_row(pc, r, q, sr, div) : -1;
    }
}

// ------------------------------------------------------------------------------------------ host
static int fill_segs(RadetSegs* out, const int* seg_desc, int nseg, int B, int out_is_o) {
    // seg_desc: nseg x 6 ints: {Hi, Wi, Ho, Wo, in_row_off, out_row_off}; rows per level = B*Ho*Wo
    if (nseg < 1 || nseg > RADET_MAX_SEG) return RADET_ERR_ARG;
    out->nseg = nseg;
    for (int l = 0; l < nseg; ++l) {
        const int* d = seg_desc + 6 * l;
        RadetSeg& s = out->s[l];
        s.Hi = d[0]; s.Wi = d[1]; s.Ho = d[2]; s.Wo = d[3];
        s.in_row_off = d[4];
        s.row_begin = d[5];
        s.row_end = d[5] + B * d[2] * d[3];
        if (l > 0 && s.row_begin != out->s[l - 1].row_end) return RADET_ERR_ARG;
    }
    (void)out_is_o;
    return RADET_OK;
}

// Experiment switches (environment, read once) of the launch heuristics; none is needed in normal operation.
struct RadetSwitches {
    bool no_tail_split, wgrad9_bm128, no_wgrad9, no_splitk;
    int wgrad_tile64_m, dbg_wgrad;
    long wgrad9_blocks, wgrad_blocks;
};
static const RadetSwitches& radet_switches() {
    static const RadetSwitches s = [] {
        RadetSwitches r;
        auto on = [](const char* n) { return getenv(n) != nullptr; };
        auto num = [](const char* n) { const char* e = getenv(n); return e ? atol(e) : 0L; };
        r.no_tail_split = on("RADET_NO_TAIL_SPLIT");
        r.no_splitk = on("RADET_NO_SPLITK");
        r.wgrad9_bm128 = on("RADET_WGRAD9_BM128");
        r.no_wgrad9 = on("RADET_NO_WGRAD9");
        r.wgrad_tile64_m = (int)num("RADET_WGRAD_TILE64_M");
        r.dbg_wgrad = (int)num("RADET_DBG_WGRAD");
        r.wgrad9_blocks = num("RADET_WGRAD9_BLOCKS");
        r.wgrad_blocks = num("RADET_WGRAD_BLOCKS");
        return r;
    }();
    return s;
}

// plane-operand instantiations (TAG bit 4): K step 16 units (32 channels) with 2 or 3 LDS stages, or 8 units with 2 or 4;
// a configuration whose tiles would not fit the CU's 160 KiB of LDS falls back to the next smaller one
template <int BM, int BN, int WM, int WN>
static void launch_p3(const ConvArgs& a, hipStream_t st, int tag, int bk, int stages, int tiles) {
    constexpr int NT = WM * WN * 64;
    constexpr int STG16 = 3 * (BM + BN) * 16 * 4;                                   // LDS bytes per stage
    constexpr int LDS_MAX = 160 * 1024;
#define RADET_LAUNCH_P3(TAGV, BKV, NSV) \
    hipLaunchKernelGGL((conv_igemmg_kernel<BM, BN, WM, WN, TAGV, BKV, NSV>), dim3(tiles, a.sk), dim3(NT), 0, st, a)
    if constexpr (3 * STG16 <= LDS_MAX) {
        if (stages >= 3) { RADET_LAUNCH_P3(16, 16, 3); return; }
    }
    if constexpr (2 * STG16 <= LDS_MAX) {
        if (tag & 1) RADET_LAUNCH_P3(17, 16, 2); else RADET_LAUNCH_P3(16, 16, 2);
    }
#undef RADET_LAUNCH_P3
}

template <int BM, int BN, int WM, int WN, bool P3ONLY = false>
static void launch_igemm(const ConvArgs& a_in, hipStream_t st, int tag, int bk, size_t ws_floats, int stages, int skw) {
    ConvArgs a = a_in;
    const int tiles = igemm_plan<BM, BN>(a, tag, bk, ws_floats, skw, radet_switches().no_tail_split);
    if (tag & 16) {
        launch_p3<BM, BN, WM, WN>(a, st, tag, bk, stages, tiles);
        return;
    }
    if (tag & 32) {                       // K-divided 4-wave tile (fp32 operands split in registers): K step 64 / 32
        if constexpr (BM == 64 && BN == 64 && WM * WN == 4) {
#define RADET_LAUNCH_KW(BKV) \
    do { if (tag & 1) hipLaunchKernelGGL((conv_igemmg_kernel<BM, BN, WM, WN, 41, BKV, 2>), dim3(tiles, a.sk), dim3(256), 0, st, a); \
         else hipLaunchKernelGGL((conv_igemmg_kernel<BM, BN, WM, WN, 40, BKV, 2>), dim3(tiles, a.sk), dim3(256), 0, st, a); } while (0)
            if (bk == 64) RADET_LAUNCH_KW(64); else RADET_LAUNCH_KW(32);
#undef RADET_LAUNCH_KW
        }
        return;
    }
#define RADET_LAUNCH_IGEMM(K, TAGV, BKV) hipLaunchKernelGGL((K<BM, BN, WM, WN, TAGV, BKV>), dim3(tiles, a.sk), dim3(256), 0, st, a)
    if constexpr (!P3ONLY) {
        if (a.sk_wgs > 0) {                                            // stream-K: tag 0, 2 stages
            if (bk == 32) hipLaunchKernelGGL((conv_igemmg_kernel<BM, BN, WM, WN, 0, 32, 2, true>), dim3(tiles, 1), dim3(256), 0, st, a);
            else hipLaunchKernelGGL((conv_igemmg_kernel<BM, BN, WM, WN, 0, 16, 2, true>), dim3(tiles, 1), dim3(256), 0, st, a);
        } else if (stages >= 3 && (tag < 2 || (tag & 8))) {
#define RADET_LAUNCH_IGEMM3(TAGV, BKV) hipLaunchKernelGGL((conv_igemmg_kernel<BM, BN, WM, WN, TAGV, BKV, 3>), dim3(tiles, a.sk), dim3(256), 0, st, a)
            if (tag & 8) { if (tag & 1) RADET_LAUNCH_IGEMM3(9, 32); else RADET_LAUNCH_IGEMM3(8, 32); }
            else if (bk == 32) { if (tag) RADET_LAUNCH_IGEMM3(1, 32); else RADET_LAUNCH_IGEMM3(0, 32); }
            else          { if (tag) RADET_LAUNCH_IGEMM3(1, 16); else RADET_LAUNCH_IGEMM3(0, 16); }
#undef RADET_LAUNCH_IGEMM3
        } else if (tag & 8) {                                          // bf16 x 3 planes (fp32 tensors), K step 32
            if (tag & 1) RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 9, 32); else RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 8, 32);
        } else if (bk == 32) {
            switch (tag) {
                case 0: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 0, 32); break;
                case 1: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 1, 32); break;
                case 2: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 2, 32); break;
                case 3: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 3, 32); break;
                case 4: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 4, 32); break;
                default: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 5, 32); break;
            }
        } else {
            switch (tag) {
                case 0: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 0, 16); break;
                case 1: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 1, 16); break;
                case 2: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 2, 16); break;
                case 3: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 3, 16); break;
                case 4: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 4, 16); break;
                default: RADET_LAUNCH_IGEMM(conv_igemmg_kernel, 5, 16); break;
            }
        }
    }
#undef RADET_LAUNCH_IGEMM
}

static long igemm_tiles(int M, int N, int choice) {
    const int bm = (choice == 3 || choice == 7 || choice == 8) ? 64 : (choice == 6 ? 256 : 128);
    const int bn = (choice == 1 || choice == 5 || choice == 6) ? 128 : (choice == 4 ? 32 : 64);
    return (long)((M + bm - 1) / bm) * ((N + bn - 1) / bn);
}

// efficiency model: wave quantisation over 256 CUs x tile padding waste x intrinsic tile efficiency
static double tile_score(int M, int N, int bm, int bn, double intrinsic) {
    const long tm = (M + bm - 1) / bm, tn = (N + bn - 1) / bn;
    const long blocks = tm * tn;
    const double quant = (double)blocks / (256.0 * ((blocks + 255) / 256));
    const double pad = ((double)M * N) / ((double)tm * bm * tn * bn);
    return quant * pad * intrinsic;
}

// rows per tap of a gather table: whole 256-row tiles (the largest block tile reads table rows [m0, m0 + 256) unguarded)
extern "C" int radet_gather_table_rows(int M) { return (M + 255) / 256 * 256; }

extern "C" int radet_build_gather_table(int* table, int B, int KH, int KW, int so, int sr, int off, int div,
                                        const int* seg_desc, int nseg, void* stream) {
    RadetSegs segs;
    int rc = fill_segs(&segs, seg_desc, nseg, B, 1);
    if (rc) return rc;
    if (segs.s[0].row_begin != 0) return RADET_ERR_ARG;
    const int M = segs.s[nseg - 1].row_end;
    const int Mp = radet_gather_table_rows(M);
    hipLaunchKernelGGL(gather_table_kernel, dim3((Mp + 255) / 256), dim3(256), 0, (hipStream_t)stream, table, segs, M, Mp,
                       KH, KW, so, sr, off, div);
    return radet_check_launch();
}

static int igemm_impl(const float* x, const float* w, const float* bias, const float* addend, const float* mask,
                      float* y, const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                      int tile_override, float* splitk_ws, size_t splitk_ws_floats, const int* out_rows,
                      const int* tap_ids, int kt_w, void* stream, const ConvPtrs* second = nullptr,
                      const int* cls = nullptr, const RadetScales* sc = nullptr);

// Two independent convolutions of identical geometry (the cls- and reg-tower layers of the shared head) as ONE
// launch: twice the tiles per launch halves the wave-quantisation loss on 256 CUs and the launch count.
extern "C" int radet_conv2d_igemm_pair(const float* x0, const float* w0, const float* bias0, const float* addend0,
                                       const float* mask0, float* y0, const float* x1, const float* w1,
                                       const float* bias1, const float* addend1, const float* mask1, float* y1,
                                       const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                                       int tile_override, float* splitk_ws, size_t splitk_ws_floats, void* stream) {
    ConvPtrs second;
    second.x = x1; second.w = w1; second.bias = bias1; second.addend = addend1; second.mask = mask1; second.y = y1;
    second.xs = second.ws = nullptr; second.ys = nullptr;
    second.yq = nullptr; second.yqs = nullptr; second.xt = second.wl1 = second.bs = second.as = nullptr;
    return igemm_impl(x0, w0, bias0, addend0, mask0, y0, gather_table, M, Cin, Cout, KH, KW, relu, tile_override,
                      splitk_ws, splitk_ws_floats, nullptr, nullptr, 0, stream, &second);
}

extern "C" int radet_conv2d_igemm(const float* x, const float* w, const float* bias, const float* addend,
                                  const float* mask, float* y, const int* gather_table, int M, int Cin, int Cout,
                                  int KH, int KW, int relu, int tile_override, float* splitk_ws, size_t splitk_ws_floats,
                                  void* stream) {
    return igemm_impl(x, w, bias, addend, mask, y, gather_table, M, Cin, Cout, KH, KW, relu, tile_override, splitk_ws,
                      splitk_ws_floats, nullptr, nullptr, 0, stream);
}

// The same entry points with the amax slots of the fp16 hi / lo arithmetic (tile_override 0x8000000) and / or of the output
// (sc->y_amax: raised to the largest |y| stored, any arithmetic).  sc is a HOST struct of device pointers.
extern "C" int radet_conv2d_igemm_s(const float* x, const float* w, const float* bias, const float* addend,
                                    const float* mask, float* y, const int* gather_table, int M, int Cin, int Cout,
                                    int KH, int KW, int relu, int tile_override, float* splitk_ws, size_t splitk_ws_floats,
                                    void* stream, const RadetScales* sc) {
    return igemm_impl(x, w, bias, addend, mask, y, gather_table, M, Cin, Cout, KH, KW, relu, tile_override, splitk_ws,
                      splitk_ws_floats, nullptr, nullptr, 0, stream, nullptr, nullptr, sc);
}

extern "C" int radet_conv2d_igemm_pair_s(const float* x0, const float* w0, const float* bias0, const float* addend0,
                                         const float* mask0, float* y0, const float* x1, const float* w1,
                                         const float* bias1, const float* addend1, const float* mask1, float* y1,
                                         const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                                         int tile_override, float* splitk_ws, size_t splitk_ws_floats, void* stream,
                                         const RadetScales* sc) {
    ConvPtrs second;
    second.x = x1; second.w = w1; second.bias = bias1; second.addend = addend1; second.mask = mask1; second.y = y1;
    second.xs = second.ws = nullptr; second.ys = nullptr;
    second.yq = nullptr; second.yqs = nullptr; second.xt = second.wl1 = second.bs = second.as = nullptr;
    return igemm_impl(x0, w0, bias0, addend0, mask0, y0, gather_table, M, Cin, Cout, KH, KW, relu, tile_override,
                      splitk_ws, splitk_ws_floats, nullptr, nullptr, 0, stream, &second, nullptr, sc);
}

// Tap-subset variant: GEMM rows are a subset of the output rows (out_rows[m] = real output row) that share the
// same set of contributing taps (tap_ids[t] = index into the weight's KTw taps); table is [ntaps][Mp].
// Used for the dgrad of strided convs: one launch per parity class does only the non-zero work.
extern "C" int radet_conv2d_igemm_taps_s(const float* x, const float* w, const float* addend, const float* mask, float* y,
                                         const int* gather_table, const int* out_rows, const int* tap_ids_host, int ntaps,
                                         int kt_w, int M, int Cin, int Cout, int tile_override, float* splitk_ws,
                                         size_t splitk_ws_floats, void* stream, const RadetScales* sc) {
    if (ntaps < 1 || ntaps > 16 || kt_w < ntaps || out_rows == nullptr || tap_ids_host == nullptr) return RADET_ERR_ARG;
    return igemm_impl(x, w, nullptr, addend, mask, y, gather_table, M, Cin, Cout, ntaps, 1, 0, tile_override, splitk_ws,
                      splitk_ws_floats, out_rows, tap_ids_host, kt_w, stream, nullptr, nullptr, sc);
}
extern "C" int radet_conv2d_igemm_taps(const float* x, const float* w, const float* addend, const float* mask, float* y,
                                       const int* gather_table, const int* out_rows, const int* tap_ids_host, int ntaps,
                                       int kt_w, int M, int Cin, int Cout, int tile_override, float* splitk_ws,
                                       size_t splitk_ws_floats, void* stream) {
    return radet_conv2d_igemm_taps_s(x, w, addend, mask, y, gather_table, out_rows, tap_ids_host, ntaps, kt_w, M, Cin, Cout,
                                     tile_override, splitk_ws, splitk_ws_floats, stream, nullptr);
}

// Class variant: ALL parity classes of a strided dgrad in one grid.  GEMM rows are the output rows sorted by class, each
// class padded to a multiple of 128 rows (out_rows = -1 on the pad rows, table entries -1); class c owns rows
// [cls_start[c], cls_start[c + 1]) (cls_start[0] = 0, the last class ends at M), runs cls_ntaps[c] <= 4 taps and its tap
// t reads weight tap tap_ids_host[4 c + t]; table is [max ntaps][M].  Order the classes by taps, most first.
extern "C" int radet_conv2d_igemm_classes_s(const float* x, const float* w, const float* addend, const float* mask, float* y,
                                            const int* gather_table, const int* out_rows, const int* tap_ids_host,
                                            const int* cls_ntaps, const int* cls_start, int ncls, int kt_w, int M, int Cin,
                                            int Cout, int tile_override, float* splitk_ws, size_t splitk_ws_floats,
                                            void* stream, const RadetScales* sc) {
    if (ncls < 1 || ncls > 4 || M % 128 != 0 || out_rows == nullptr || tap_ids_host == nullptr || cls_ntaps == nullptr ||
        cls_start == nullptr || cls_start[0] != 0)
        return RADET_ERR_ARG;
    int cls[9] = {ncls, 0, 0, 0, 0, 0, 0, 0, 0}, tids[16] = {0}, kmax = 0;
    for (int c = 0; c < ncls; ++c) {
        if (cls_ntaps[c] < 1 || cls_ntaps[c] > 4 || cls_start[c] % 128 != 0 || cls_start[c] >= M ||
            (c > 0 && cls_start[c] <= cls_start[c - 1]))
            return RADET_ERR_ARG;
        cls[1 + c] = cls_ntaps[c]; cls[5 + c] = cls_start[c];
        if (cls_ntaps[c] > kmax) kmax = cls_ntaps[c];
        for (int t = 0; t < cls_ntaps[c]; ++t) {
            if (tap_ids_host[4 * c + t] < 0 || tap_ids_host[4 * c + t] >= kt_w) return RADET_ERR_ARG;
            tids[4 * c + t] = tap_ids_host[4 * c + t];
        }
    }
    return igemm_impl(x, w, nullptr, addend, mask, y, gather_table, M, Cin, Cout, kmax, 1, 0, tile_override, splitk_ws,
                      splitk_ws_floats, out_rows, tids, kt_w, stream, nullptr, cls, sc);
}
extern "C" int radet_conv2d_igemm_classes(const float* x, const float* w, const float* addend, const float* mask, float* y,
                                          const int* gather_table, const int* out_rows, const int* tap_ids_host,
                                          const int* cls_ntaps, const int* cls_start, int ncls, int kt_w, int M, int Cin,
                                          int Cout, int tile_override, float* splitk_ws, size_t splitk_ws_floats,
                                          void* stream) {
    return radet_conv2d_igemm_classes_s(x, w, addend, mask, y, gather_table, out_rows, tap_ids_host, cls_ntaps, cls_start, ncls,
                                        kt_w, M, Cin, Cout, tile_override, splitk_ws, splitk_ws_floats, stream, nullptr);
}

static int igemm_impl(const float* x, const float* w, const float* bias, const float* addend, const float* mask,
                      float* y, const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                      int tile_override, float* splitk_ws, size_t splitk_ws_floats, const int* out_rows,
                      const int* tap_ids, int kt_w, void* stream, const ConvPtrs* second, const int* cls,
                      const RadetScales* sc) {
    const int h16 = (tile_override >> 11) & 1;                 // 0x800: bf16 storage, 0x10000: fp32 output from bf16 inputs
    // 0x2000000: x and w are bf16 plane triples (x rows [3][Cin], w [Cout][taps][3][Cin]; hi + mid + lo = the fp32 value);
    // y / addend / mask / bias stay fp32.  +0x4000000: K step of 16 instead of 32 channels
    const int p3 = ((tile_override >> 25) & 1) && !h16;
    if (h16 || p3) {
        if (Cin % 32 != 0) return RADET_ERR_ARG;               // 16 channel pairs per K step at least
        Cin /= 2;                                              // K is counted in channel pairs (4-byte units) from here on
    }
    // (gather_table may be NULL for a 1 x 1 / stride 1 conv whose input and output rows coincide: GEMM row m reads row m)
    if (Cin % 16 != 0 || Cin <= 0 || Cout <= 0 || M <= 0 || KH * KW > 16 ||
        (gather_table == nullptr && (KH * KW != 1 || out_rows != nullptr || cls != nullptr)))
        return RADET_ERR_ARG;
    ConvArgs a;
    a.io = h16 ? (((tile_override >> 16) & 1) ? 2 : 1) : 0;
    a.out_rows = out_rows;
    for (int t = 0; t < 16; ++t) a.tap_ids[t] = tap_ids ? (t < KH * KW ? tap_ids[t] : 0) : t;
    a.cls_nt = 0; a.cls_b[0] = a.cls_b[1] = a.cls_b[2] = 0x7fffffff;
    if (cls != nullptr) {                                      // cls = {ncls, ntaps[4], start[4]}: class launch
        for (int t = 0; t < 16; ++t) a.tap_ids[t] = tap_ids[t];
        for (int c = 0; c < cls[0]; ++c) {
            a.cls_nt |= cls[1 + c] << (4 * c);
            if (c > 0) a.cls_b[c - 1] = cls[5 + c];
        }
    }
    a.KTw = kt_w > 0 ? kt_w : KH * KW;
    { static const int dbg = getenv("RADET_DBG_IGEMM") ? atoi(getenv("RADET_DBG_IGEMM")) : 0; a.dbg = dbg; }
    // 0x10000000: `mask` is an fp16 plane-pair tensor (rows [2][Cout]) -- the ReLU mask of an activation that exists only as
    // pairs (fp32 tensors only)
    a.maskq = ((tile_override >> 28) & 1) && mask != nullptr;
    if (a.maskq && (h16 || Cout % 32 != 0 || second != nullptr)) return RADET_ERR_ARG;
    // y may be NULL when the output is wanted as plane pairs only (RadetScales.yq): the launch then stores no fp32 tensor
    if (y == nullptr && (sc == nullptr || sc->yq == nullptr)) return RADET_ERR_ARG;
    a.p[0].x = x; a.p[0].w = w; a.p[0].bias = bias; a.p[0].addend = addend; a.p[0].mask = mask; a.p[0].y = y;
    a.p[0].xs = sc ? (const unsigned*)sc->x_amax : nullptr;
    a.p[0].ws = sc ? (const unsigned*)sc->w_amax : nullptr;
    a.p[0].ys = sc ? (unsigned*)sc->y_amax : nullptr;
    a.p[0].yq = sc ? (_Float16*)sc->yq : nullptr;
    a.p[0].yqs = sc ? (unsigned*)sc->yq_amax : nullptr;
    a.p[0].xt = sc ? (const unsigned*)sc->x_true_amax : nullptr;
    a.p[0].wl1 = sc ? (const unsigned*)sc->w_l1 : nullptr;
    a.p[0].bs = (sc && bias) ? (const unsigned*)sc->bias_amax : nullptr;
    a.p[0].as = (sc && addend) ? (const unsigned*)sc->addend_amax : nullptr;
    if (a.p[0].yq != nullptr) {              // pair copy of the output: fp32 outputs, whole 32-channel groups, every slot it needs
        if (h16 || Cout % 32 != 0 || a.p[0].yqs == nullptr || a.p[0].xt == nullptr || a.p[0].wl1 == nullptr ||
            (bias && a.p[0].bs == nullptr) || (addend && a.p[0].as == nullptr) || second != nullptr)
            return RADET_ERR_ARG;
    }
    a.p[1] = a.p[0];
    a.groups = 1;
    if (second != nullptr) {
        a.p[1] = *second;
        a.p[1].yq = nullptr; a.p[1].yqs = nullptr; a.p[1].xt = a.p[1].wl1 = a.p[1].bs = a.p[1].as = nullptr;
        a.p[1].xs = sc ? (const unsigned*)sc->x1_amax : nullptr;
        a.p[1].ws = sc ? (const unsigned*)sc->w1_amax : nullptr;
        a.p[1].ys = sc ? (unsigned*)sc->y1_amax : nullptr;
        a.groups = 2;
    }
    a.rowtab = gather_table;
    a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW;
    a.relu = relu;
    a.M = M;
    a.Mp = radet_gather_table_rows(M);
    hipStream_t st = (hipStream_t)stream;
    int tag = h16 ? (4 | ((tile_override >> 8) & 1))
                  : (((tile_override >> 8) & 1) | (((tile_override >> 10) & 1) << 1));   // 0x100 symbol tag, 0x400 bf16 math
    int bk = ((tile_override >> 9) & 1) ? 32 : 16;
    if (Cin % 32 != 0) bk = 16;
    // 0x1000000: fp32 tensors, products from three bf16 planes per operand (6 bf16 MFMAs per K = 16 step); K step 32
    const bool x3 = ((tile_override >> 24) & 1) && !h16 && !p3 && !((tile_override >> 10) & 1) && Cin % 32 == 0;
    if (x3) { bk = 32; tag |= 8; }
    if (p3) { tag = 16 | ((tile_override >> 8) & 1); bk = 16; }
    // 0x8000000 (with 0x1000000 or 0x2000000): fp16 hi / lo arithmetic -- operands scaled by the power of two their amax
    // slots give, two fp16 planes each, three f16 MFMAs per K = 16 step (common.h "h2"); with 0x2000000 the operands ARRIVE
    // as fp16 plane pairs.  Needs the slots (radet_conv2d_igemm_s).  Without a plane-capable K (Cin % 32) the launch falls
    // back to the native fp32 MFMA like 0x1000000 does.
    const bool h2 = ((tile_override >> 27) & 1) && (x3 || p3);
    if (h2) {
        for (int g = 0; g < a.groups; ++g)
            if (a.p[g].xs == nullptr || a.p[g].ws == nullptr) return RADET_ERR_ARG;
        tag |= 64;
    }
    int choice = tile_override & 0xFF;
    if (p3 && h2 && ((tile_override >> 19) & 1) && (choice == 5 || choice == 6)) {
        // 0x80000: row-interleaved plane pairs for the 8-wave tiles (TAG bit 7 with bit 4): one wave load fetches both planes
        // of 8 tile rows (128 contiguous bytes per row) instead of one plane of 16 -- the launch counts K in channels again
        // (a pair row is as long as the fp32 row) and its stage is 32 four-byte units
        Cin *= 2;
        a.Cin = Cin;
        if (Cin % 32 != 0) return RADET_ERR_ARG;
        tag |= 128; bk = 32;
    }
    if (choice == 7 && p3 && h2) {
        // K-divided 64 x 64 tile on fp16 plane PAIRS (round 6): a pair row has the fp32 row's byte length, so the launch is
        // the fp32 K-divided one (K counted in channels again, 64-channel stages) with the reader's pair flag (TAG bit 7)
        Cin *= 2;
        a.Cin = Cin;
        if (Cin % 64 != 0 || ((tile_override >> 20) & 7) || a.groups != 1) return RADET_ERR_ARG;
        tag = 8 | 32 | 64 | 128; bk = 64;
    } else
    if (choice == 7 || choice == 8) {                          // 64 x 64 tiles whose four waves divide the K step (see TAG bit 5):
        // 7: four k-groups of a 64-channel stage; 8: two k-groups x two column halves of a 32-channel stage
        const int kbk = choice == 7 ? 64 : 32;
        if (!x3 || Cin % kbk != 0 || ((tile_override >> 20) & 7)) return RADET_ERR_ARG;
        tag |= 32; bk = kbk;
    } else
    if (choice > 4 && !p3) return RADET_ERR_ARG;               // the 8-wave tiles exist for plane operands only
    if (choice == 6 && cls != nullptr) return RADET_ERR_ARG;   // class boundaries are multiples of 128 rows
    if (choice <= 0) {
        if (Cout <= 32) choice = 4;
        else {
            const double s1 = tile_score(a.M, Cout, 128, 128, 1.00);
            const double s2 = tile_score(a.M, Cout, 128, 64, 0.96);
            const double s3 = tile_score(a.M, Cout, 64, 64, 0.90);
            choice = 1;
            double best = s1;
            if (s2 > best) { best = s2; choice = 2; }
            if (s3 > best) { best = s3; choice = 3; }
        }
    }
    // split-K for launches that cannot fill 256 CUs twice over (low-M stages): each split keeps >= 8 K stages
    const int nK = KH * KW * (Cin / bk);
    int sk = 1;
    // 0x20000: 3 LDS stages (launches that run alone); 0x40000: 4 (the fp16 hi / lo K-divided tile 8 only)
    const int stages3 = ((tile_override >> 18) & 1) ? 4 : (((tile_override >> 17) & 1) ? 3 : 2);
    const int skw = cls ? 0 : (tile_override >> 20) & 7;     // 0x100000 * w: stream-K, w workgroups per CU
    const int sk_force = skw ? 1 : (tile_override >> 12) & 0xF;
    const long tiles = igemm_tiles(a.M, Cout, choice) * a.groups;
    if (splitk_ws != nullptr && a.groups == 1 && !radet_switches().no_splitk) {
        if (sk_force) sk = sk_force;
        else if (tiles < 384) {
            sk = (int)((512 + tiles - 1) / tiles);
            if (sk > nK / 8) sk = nK / 8;
            if (sk > 8) sk = 8;
        }
        if (sk < 1) sk = 1;
    }
    a.sk = sk;
    a.it_per_split = (nK + sk - 1) / sk;
    // workspace = RADET_SPLIT_COUNTERS arrival tickets (ints, zero between launches) followed by the partial tiles
    a.counters = (int*)splitk_ws;
    a.partial = nullptr;
    if (splitk_ws != nullptr && splitk_ws_floats > RADET_SPLIT_COUNTERS) {
        a.partial = splitk_ws + RADET_SPLIT_COUNTERS;
        splitk_ws_floats -= RADET_SPLIT_COUNTERS;
    } else {
        a.sk = 1;
        a.it_per_split = nK;
        splitk_ws_floats = 0;
    }
    if (tag & 64) {                                            // (stream-K bits are ignored, as for the bf16-plane arithmetic)
        if (!radet_launch_igemm_h2(choice, a, st, tag, bk, splitk_ws_floats, stages3, radet_switches().no_tail_split))
            return RADET_ERR_ARG;
        return radet_check_launch();
    }
    switch (choice) {
        case 1: launch_igemm<128, 128, 2, 2>(a, st, tag, bk, splitk_ws_floats, stages3, skw); break;
        case 2: launch_igemm<128, 64, 2, 2>(a, st, tag, bk, splitk_ws_floats, stages3, skw); break;
        case 3: launch_igemm<64, 64, 2, 2>(a, st, tag, bk, splitk_ws_floats, stages3, skw); break;
        case 4: launch_igemm<128, 32, 4, 1>(a, st, tag, bk, splitk_ws_floats, stages3, skw); break;
        case 5: launch_igemm<128, 128, 2, 4, true>(a, st, tag, bk, splitk_ws_floats, stages3, skw); break;   // 8 waves
        case 6: launch_igemm<256, 128, 4, 2, true>(a, st, tag, bk, splitk_ws_floats, stages3, skw); break;   // 8 waves
        case 7: case 8: launch_igemm<64, 64, 2, 2, true>(a, st, tag, bk, splitk_ws_floats, 2, skw); break;   // K-divided
        default: return RADET_ERR_ARG;
    }
    return radet_check_launch();
}

template <int BM, int BN, int WM, int WN>
static void launch_wgrad(const WgradArgs& a, hipStream_t st) {
    const int tiles = ((a.Cout + BM - 1) / BM) * ((a.Cin + BN - 1) / BN) * a.KH * a.KW * a.S;
    const bool bp32 = a.bp32 != 0;                                        // 32 pixels per LDS stage (half the barriers)
    if (a.math == 2 && bp32 && BM >= 64) hipLaunchKernelGGL((conv_wgradg_kernel<BM, BN, WM, WN, 2, 32>), dim3(tiles), dim3(256), 0, st, a);
    else if (a.math == 2) hipLaunchKernelGGL((conv_wgradg_kernel<BM, BN, WM, WN, 2>), dim3(tiles), dim3(256), 0, st, a);
    else if (a.math == 1) hipLaunchKernelGGL((conv_wgradg_kernel<BM, BN, WM, WN, 1>), dim3(tiles), dim3(256), 0, st, a);
    else if (bp32 && BM >= 64) hipLaunchKernelGGL((conv_wgradg_kernel<BM, BN, WM, WN, 0, 32>), dim3(tiles), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_wgradg_kernel<BM, BN, WM, WN, 0>), dim3(tiles), dim3(256), 0, st, a);
}

static int wgrad9_bm(int Cout) { return (Cout >= 256 && !radet_switches().wgrad9_bm128) ? 256 : 128; }

// all-taps kernel: pays off for 3x3 convs with >= 256 output channels and a long pixel dimension (head towers,
// FPN P3): measured 96.6 vs 84.5 TFLOP/s on the tower shape; for the short-M backbone stages the one-tap kernel
// with its finer tile grid stays ahead (tools/bench_conv.py)
static bool use_wgrad9(int M, int Cin, int Cout, int KH, int KW) {
    return KH == 3 && KW == 3 && Cin % 32 == 0 && Cout >= 256 && M >= 16384 && !radet_switches().no_wgrad9;
}


// Tile of the one-tap wgrad kernel.  Short pixel dimensions (M) would need many K-splits to fill the GPU with
// 128x128 tiles, and every split costs a weight-sized slab write + read; a 64x64 tile grid has 4x the tiles, so a
// quarter of the splits, at a modest loss of MFMA efficiency -> preferred when M is short.
static void wgrad_tile(int M, int Cout, int Cin, int KT, int* bm, int* bn) {
    if (Cout <= 32) { *bm = 32; *bn = 128; return; }
    if (Cout <= 64 || Cin <= 64) { *bm = 64; *bn = 64; return; }
    *bm = 128; *bn = 128;
    const long tiles128 = (long)((Cout + 127) / 128) * ((Cin + 127) / 128) * KT;
    const long s128 = (448 + tiles128 - 1) / tiles128;
    const int mthr = radet_switches().wgrad_tile64_m;   // measured: no net gain on R50 640x480 (kept as a switch)
    if (M <= mthr && s128 >= 4) { *bm = 64; *bn = 64; }
}

// Number of pixel splits the wgrad launcher will use (callers size the slab buffer with it).
extern "C" int radet_conv2d_wgrad_splits(int M, int Cin, int Cout, int KH, int KW) {
    int bm, bn;
    wgrad_tile(M, Cout, Cin, KH * KW, &bm, &bn);
    long tiles = (long)((Cout + bm - 1) / bm) * ((Cin + bn - 1) / bn) * KH * KW;
    if (use_wgrad9(M, Cin, Cout, KH, KW)) {
        const int b9 = wgrad9_bm(Cout);
        tiles = (long)((Cout + b9 - 1) / b9) * (Cin / 32);
    }
    const int chunks = (M + 15) / 16;
    // pick S in the 2..4 blocks-per-CU range whose block count quantises best onto 256 CUs
    long lo = (448 + tiles - 1) / tiles, hi = (1024 + tiles - 1) / tiles;
    if (use_wgrad9(M, Cin, Cout, KH, KW)) {
        // 8-wave workgroups: one per CU already gives 2 waves per SIMD, and every extra split costs a full
        // weight-sized slab write + read in unfold
        lo = hi = (256 + tiles - 1) / tiles;
        if (radet_switches().wgrad9_blocks) lo = hi = (radet_switches().wgrad9_blocks + tiles / 2) / tiles;
    }
    if (radet_switches().wgrad_blocks) lo = hi = (radet_switches().wgrad_blocks + tiles / 2) / tiles;
    if (lo < 1) lo = 1;
    long S = lo;
    double best = -1.0;
    for (long c = lo; c <= hi; ++c) {
        const long blocks = tiles * c;
        const double eff = (double)blocks / (256.0 * ((blocks + 255) / 256)) - 0.002 * (double)(c - lo);
        if (eff > best) { best = eff; S = c; }
    }
    const long maxS = (chunks + 7) / 8;          // at least 8 stages (128 pixels) per block
    if (S > maxS) S = maxS;
    if (S < 1) S = 1;
    if (S > 64) S = 64;
    return (int)S;
}

static int wgrad_impl(const float* dy, const float* x, float* slabs, float* dbias_partials, const int* gather_table, int M,
                      int Cin, int Cout, int ld_dy, int KH, int KW, int S, int flags, void* stream, const RadetScales* sc);
extern "C" int radet_conv2d_wgrad(const float* dy, const float* x, float* slabs, float* dbias_partials,
                                  const int* gather_table, int M, int Cin, int Cout, int ld_dy, int KH, int KW, int S,
                                  int flags, void* stream) {
    return wgrad_impl(dy, x, slabs, dbias_partials, gather_table, M, Cin, Cout, ld_dy, KH, KW, S, flags, stream, nullptr);
}
// with the amax slots of the fp16 hi / lo arithmetic (flags 0x1000): sc->x_amax = the slot of dy, sc->w_amax = the slot of x
extern "C" int radet_conv2d_wgrad_s(const float* dy, const float* x, float* slabs, float* dbias_partials,
                                    const int* gather_table, int M, int Cin, int Cout, int ld_dy, int KH, int KW, int S,
                                    int flags, void* stream, const RadetScales* sc) {
    return wgrad_impl(dy, x, slabs, dbias_partials, gather_table, M, Cin, Cout, ld_dy, KH, KW, S, flags, stream, sc);
}
static int wgrad_impl(const float* dy, const float* x, float* slabs, float* dbias_partials, const int* gather_table, int M,
                      int Cin, int Cout, int ld_dy, int KH, int KW, int S, int flags, void* stream, const RadetScales* sc) {
    // dy rows must be 16-byte aligned and hold whole float4s for every real channel (pad small heads with zeros)
    if (Cin % 4 != 0 || S < 1 || ld_dy < Cout || (ld_dy & 3) || ((Cout + 3) / 4) * 4 > ld_dy || M <= 0 ||
        (gather_table == nullptr && KH * KW != 1))
        return RADET_ERR_ARG;
    if ((flags & 2) && ((Cout + 7) / 8) * 8 > ld_dy) return RADET_ERR_ARG;   // whole 8-channel groups per dy row
    WgradArgs a;
    a.ld_dy = ld_dy;
    a.dy = dy; a.x = x; a.slabs = slabs; a.dbias_partials = dbias_partials;
    a.rowtab = gather_table;
    a.Cin = Cin; a.Cout = Cout; a.KH = KH; a.KW = KW;
    a.M = M;
    a.Mp = radet_gather_table_rows(M);
    a.S = S;
    a.dbg = radet_switches().dbg_wgrad;
    a.math = (flags & 1) ? 1 : (((flags >> 8) & 1) && !(flags & 2) ? 2 : 0);   // 0x100: fp32 products from 3 bf16 planes
    a.bp32 = (flags >> 7) & 1;
    a.dys = sc ? (const unsigned*)sc->x_amax : nullptr;
    a.xss = sc ? (const unsigned*)sc->w_amax : nullptr;
    const int chunks = (a.M + 15) / 16;
    a.chunks_per_split = (chunks + S - 1) / S;
    hipStream_t st = (hipStream_t)stream;
    if (flags & 0x1000) {
        // fp16 hi / lo arithmetic (common.h "h2"): fp32 tensors split in registers (one-tap tiles), or, with 0x200, operands
        // that arrive as fp16 plane pairs (all-taps kernel, 3 x 3 only); needs both amax slots
        if ((flags & 3) || a.dys == nullptr || a.xss == nullptr) return RADET_ERR_ARG;
        a.math = 3;
        int bm = 0, bn = 0;
        if (!(flags & 0x200)) {
            wgrad_tile(M, Cout, Cin, KH * KW, &bm, &bn);
            if (bm != 32 && ((flags >> 4) & 3) == 1) bm = bn = 128;
            if (bm != 32 && ((flags >> 4) & 3) == 2) bm = bn = 64;
            if (bm != 32 && ((flags >> 4) & 3) == 3) { bm = 128; bn = 64; }
        }
        return radet_launch_wgrad_h2(a, flags, bm, bn, st);
    }
    if (flags & 0x200) {   // plane operands: dy rows [3][ld_dy] bf16, x rows [3][Cin] bf16 (hi | mid | lo)
        if ((ld_dy & 31) || (Cin & 31) || (flags & 3)) return RADET_ERR_ARG;
        if (KH == 3 && KW == 3 && Cin % 32 == 0) {
            const int tiles9 = ((Cout + 255) / 256) * (Cin / 32) * S;
            hipLaunchKernelGGL(conv_wgrad9p_kernel, dim3(tiles9), dim3(512), 0, st, a);
            return radet_check_launch();
        }
        return RADET_ERR_ARG;
    }
    if (flags & 2) {   // bf16 storage: dy / x are bf16 (ld_dy, Cin in elements; 16-byte aligned rows)
        if ((ld_dy & 7) || (Cin & 7)) return RADET_ERR_ARG;
        if (use_wgrad9(M, Cin, Cout, KH, KW) && wgrad9_bm(Cout) == 256 && !(flags & 0x40) && !radet_switches().no_wgrad9) {
            const int tiles9 = ((Cout + 255) / 256) * (Cin / 32) * S;
            hipLaunchKernelGGL(conv_wgrad9h_kernel, dim3(tiles9), dim3(512), 0, st, a);
            return radet_check_launch();
        }
        int bm, bn;
        wgrad_tile(M, Cout, Cin, KH * KW, &bm, &bn);
        if (bm != 32 && ((flags >> 4) & 3) == 1) bm = bn = 128;
        if (bm != 32 && ((flags >> 4) & 3) != 1) bm = bn = 64;
        const int tiles = ((Cout + bm - 1) / bm) * ((Cin + bn - 1) / bn) * KH * KW * S;
        if (bm == 32) hipLaunchKernelGGL((conv_wgradh_kernel<32, 128, 1, 4>), dim3(tiles), dim3(256), 0, st, a);
        else if (bm == 64) hipLaunchKernelGGL((conv_wgradh_kernel<64, 64, 2, 2>), dim3(tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgradh_kernel<128, 128, 2, 2>), dim3(tiles), dim3(256), 0, st, a);
        return radet_check_launch();
    }
    if (use_wgrad9(M, Cin, Cout, KH, KW) && !(flags & 0x40) && (a.math == 0 || wgrad9_bm(Cout) == 256)) {
        if (wgrad9_bm(Cout) == 256) {
            const int tiles = ((Cout + 255) / 256) * (Cin / 32) * S;
            if (a.math == 2) hipLaunchKernelGGL((conv_wgrad9g_kernel<8, 2>), dim3(tiles), dim3(512), 0, st, a);
            else if (a.math == 1) hipLaunchKernelGGL((conv_wgrad9g_kernel<8, 1>), dim3(tiles), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((conv_wgrad9g_kernel<8, 0>), dim3(tiles), dim3(512), 0, st, a);
        } else {
            const int tiles = ((Cout + 127) / 128) * (Cin / 32) * S;
            hipLaunchKernelGGL((conv_wgrad9g_kernel<4, 0>), dim3(tiles), dim3(256), 0, st, a);
        }
        return radet_check_launch();
    }
    int bm, bn;
    wgrad_tile(M, Cout, Cin, KH * KW, &bm, &bn);
    if (bm != 32 && ((flags >> 4) & 3) == 1) bm = bn = 128;      // autotuned tile (radet_amd/kernels.py)
    if (bm != 32 && ((flags >> 4) & 3) == 2) bm = bn = 64;
    // 128 (output channels) x 64 (input channels): two accumulators per wave -- three operand splits per two MFMA blocks
    // instead of two per block; offered to the tuner with the bf16-plane arithmetic, where the 64 x 64 tile is VALU-bound
    if (bm != 32 && ((flags >> 4) & 3) == 3 && a.math == 2) { bm = 128; bn = 64; }
    // 0x400 / 0x800 (bf16-plane arithmetic, 64 x 64 tile): the four waves divide a 64-pixel stage four ways / a 32-pixel
    // stage two ways (x two column halves) and share the operand splits (see wgradg_body, KD)
    if (bm == 64 && a.math == 2 && (flags & 0xC00)) {
        const int tiles = ((a.Cout + 63) / 64) * ((a.Cin + 63) / 64) * a.KH * a.KW * a.S;
        if (flags & 0x400) hipLaunchKernelGGL((conv_wgradg_kernel<64, 64, 2, 2, 2, 64, 4>), dim3(tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgradg_kernel<64, 64, 2, 2, 2, 32, 2>), dim3(tiles), dim3(256), 0, st, a);
    } else
    if (bm == 32) launch_wgrad<32, 128, 1, 4>(a, st);
    else if (bm == 64) launch_wgrad<64, 64, 2, 2>(a, st);
    else if (bn == 64) {
        const int tiles = ((a.Cout + 127) / 128) * ((a.Cin + 63) / 64) * a.KH * a.KW * a.S;
        if (a.bp32) hipLaunchKernelGGL((conv_wgradg_kernel<128, 64, 2, 2, 2, 32>), dim3(tiles), dim3(256), 0, st, a);
        else hipLaunchKernelGGL((conv_wgradg_kernel<128, 64, 2, 2, 2>), dim3(tiles), dim3(256), 0, st, a);
    } else launch_wgrad<128, 128, 2, 2>(a, st);
    return radet_check_launch();
}

// Grouped one-tap wgrad (see conv_wgradg_group_kernel): jobs[i] describes one conv exactly like the arguments of
// radet_conv2d_wgrad.  flags bit 0: bf16 math mode; bits 4-5: tile (1 = 128x128, otherwise 64x64).
extern "C" int radet_conv2d_wgrad_group(const RadetWgradJob* jobs, int njobs, int flags, void* stream) {
    if (njobs < 1 || njobs > WG_MAX || jobs == nullptr) return RADET_ERR_ARG;
    const int bm = ((flags >> 4) & 3) == 1 ? 128 : 64;
    WgradGroup g;
    g.n = njobs;
    int total = 0;
    for (int i = 0; i < njobs; ++i) {
        const RadetWgradJob& j = jobs[i];
        if (j.Cin % 4 != 0 || j.S < 1 || j.ld_dy < j.Cout || (j.ld_dy & 3) || j.M <= 0 || (j.gather_table == nullptr && j.KH * j.KW != 1) ||
            j.Cout % bm != 0 || j.Cin % bm != 0)
            return RADET_ERR_ARG;
        WgradArgs& a = g.p[i];
        a.dy = j.dy; a.x = j.x; a.slabs = j.slabs; a.dbias_partials = j.dbias_partials; a.rowtab = j.gather_table;
        a.M = j.M; a.Mp = radet_gather_table_rows(j.M); a.Cin = j.Cin; a.Cout = j.Cout; a.KH = j.KH; a.KW = j.KW;
        a.ld_dy = j.ld_dy; a.S = j.S; a.dbg = 0; a.math = flags & 1; a.bp32 = 0; a.dys = a.xss = nullptr;
        const int chunks = (j.M + 15) / 16;
        a.chunks_per_split = (chunks + j.S - 1) / j.S;
        g.begin[i] = total;
        total += (j.Cout / bm) * (j.Cin / bm) * j.KH * j.KW * j.S;
    }
    g.begin[njobs] = total;
    hipStream_t st = (hipStream_t)stream;
    if (bm == 128) {
        if (flags & 1) hipLaunchKernelGGL((conv_wgradg_group_kernel<128, 128, 2, 2, 1>), dim3(total), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((conv_wgradg_group_kernel<128, 128, 2, 2, 0>), dim3(total), dim3(256), 0, st, g);
    } else {
        if (flags & 1) hipLaunchKernelGGL((conv_wgradg_group_kernel<64, 64, 2, 2, 1>), dim3(total), dim3(256), 0, st, g);
        else hipLaunchKernelGGL((conv_wgradg_group_kernel<64, 64, 2, 2, 0>), dim3(total), dim3(256), 0, st, g);
    }
    return radet_check_launch();
}
