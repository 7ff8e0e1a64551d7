// Weight-gradient kernel templates on fp32 tensors (one tap per workgroup, all nine taps), shared by conv_igemm.hip and
// conv_wgrad_h2.hip.  gfx950 only.
#pragma once
#include "conv_igemm_kernel.h"
// ------------------------------------------------------------------------------------------ wgrad
struct WgradArgs {
    const float* dy;  // [M][Cout]
    const float* x;   // input rows [*, Cin]
    float* slabs;     // [S][Cout][KH*KW][Cin]
    float* dbias_partials;  // [S][Cout] column sums of dy (bias / BN-shift gradient) or null
    const int* rowtab;      // [KH*KW][Mp] gather table (same as the forward conv's)
    int M, Mp, Cin, Cout, KH, KW;
    int ld_dy;        // row stride of dy (>= Cout; padded gradient buffers)
    int S;            // pixel splits
    int chunks_per_split;  // 16-pixel chunks per split
    int dbg;          // experiments only (RADET_DBG_WGRAD): 1 = skip global loads after the first stage
    int math;         // 0: fp32 MFMA; 1: operands rounded to bf16, fp32 accumulate (LDS-DMA kernels only)
    int bp32;         // one-tap fp32 kernel: 32 instead of 16 pixels per stage (flags bit 7; chosen by the autotuner)
    const unsigned* dys;    // math 3 (fp16 hi / lo arithmetic, common.h "h2"): amax slots of dy and of x
    const unsigned* xss;
};

// ------------------------------------------------------------------------------------------ wgrad, all 9 taps
// 3x3 convs: one workgroup owns a (128 output-channel) x (32 input-channel) tile of ALL nine taps for its pixel
// split.  The dy tile is loaded once per stage instead of once per tap, and the nine shifted x tiles overlap
// in L1 (they read the same 3x(16+2) pixel rows), so the L2->LDS traffic per MAC drops ~2.4x against the
// one-tap kernel above, and 72 MFMAs (9 taps x 8 K steps) run between barriers instead of 32.
// Wave w owns output channels [32w, 32w+32): 9 accumulator tiles (144 AGPRs), one A fragment feeds 9 MFMAs.
// The dy tile and the nine x tiles go global -> LDS directly
// (global_load_lds_dwordx4: each wave instruction lands 1 KiB lane-linearly, which is exactly one 256-channel dy
// row or eight 32-channel x rows of the unpadded tiles).  No staging VGPRs next to the 144 accumulators, no ds_write
// pass; padding / out-of-range rows are read from a zero page.


template <int NW, int MATH>   // MATH 1: bf16 operands (rounded from the fp32 tiles), fp32 accumulate
__global__ __launch_bounds__(NW * 64) void conv_wgrad9g_kernel(const WgradArgs a) {
    radet_kernarg_warm<sizeof(WgradArgs)>();
    constexpr int BP = 16, BM = 32 * NW, BC = 32, KT = 9, NT = NW * 64;
    constexpr int A_INSTR = BP * BM * 4 / 1024;             // wave instructions per dy tile (1 KiB each)
    constexpr int B_INSTR = KT * BP * BC * 4 / 1024;        // 18
    constexpr int N_INSTR = A_INSTR + B_INSTR;
    constexpr int PER_WAVE = (N_INSTR + NW - 1) / NW;
    constexpr int ROWS_A = 1024 / (BM * 4);                 // dy rows per instruction (1 for BM=256, 2 for BM=128)
    __shared__ __attribute__((aligned(16))) float As[2][BP * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][KT * BP * BC];

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

    const int p_begin = split * a.chunks_per_split * BP;
    int p_end = p_begin + a.chunks_per_split * BP;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + BP - 1) / BP : 0;

    // gather-table rows of the NEXT stage for this wave's x-tile instructions (loaded one stage ahead; unconditional,
    // clamped loads whose validity is applied at use -- see conv_wgradg for why)
    int brow[PER_WAVE];
    bool bok[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int ins = wave + k * NW - A_INSTR;             // x-tile instruction index (tap, half)
        brow[k] = -1;
        bok[k] = false;
        if (ins >= 0 && ins < B_INSTR) {
            const int m = p_begin + (ins & 1) * 8 + (lane >> 3);
            brow[k] = a.rowtab[(size_t)(ins >> 1) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
            bok[k] = m < p_end;
        }
    }
    auto issue_stage = [&](int it, int buf) {                // order: x tiles (consume brow), dy tiles, next gather rows
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
                radet_lds_load16(a.x, (bok[k] && brow[k] >= 0), (size_t)((size_t)brow[k] * a.Cin + c0 + (lane & 7) * 4), (lptr_t)(&Bs[buf][bi * 256]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int j = ins * ROWS_A + (ROWS_A == 1 ? 0 : (lane * 4) / BM);
                const int o = o0 + (lane * 4) % BM;
                const int m = p0 + j;
                radet_lds_load16(a.dy, (m < p_end && o < a.Cout), (size_t)((size_t)m * a.ld_dy + o), (lptr_t)(&As[buf][ins * 256]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int m = p0 + BP + (bi & 1) * 8 + (lane >> 3);
                brow[k] = a.rowtab[(size_t)(bi >> 1) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
                bok[k] = m < p_end;
            }
        }
    };

    f32x16 acc[KT];
#pragma unroll
    for (int t = 0; t < KT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bsum = 0.f;                                        // column sum of dy for o = o0 + tid % BM (bias gradient)
    const bool want_bias = a.dbias_partials != nullptr && tc == 0;

    if (nIt > 0) issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    // fragment reads are inline asm (see conv_igemmg_kernel): the compiler would put a vmcnt(0) wait on the in-flight
    // LDS-DMA loads of the other buffer in front of every ds_read it can see
    const unsigned a_addr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)((lh * BM + wave * 32 + li) * 4);
    const unsigned b_addr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)((lh * BC + li) * 4);
    const unsigned a_addr4 = a_addr + (unsigned)(3 * lh * BM * 4), b_addr4 = b_addr + (unsigned)(3 * lh * BC * 4);  // row 4*lh
    const unsigned a_addr8 = a_addr + (unsigned)(7 * lh * BM * 4), b_addr8 = b_addr + (unsigned)(7 * lh * BC * 4);  // row 8*lh
    auto stage = [&](auto bufc, int it) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr int AO = BUF * BP * BM * 4, BO = BUF * KT * BP * BC * 4;
        if (want_bias) {                                     // plain LDS reads: keep them ahead of the next loads
            constexpr int RPT = BP * BM / NT;                // rows per thread: 8
#pragma unroll
            for (int j = 0; j < RPT; ++j) bsum += As[BUF][((tid / BM) * RPT + j) * BM + (tid % BM)];
        }
        if (it + 1 < nIt) issue_stage(it + 1, BUF ^ 1);
        if constexpr (MATH == 1) {
            // v_mfma_f32_32x32x8_bf16_1k: lane (i, h) holds pixels 4h..4h+3 of an 8-pixel group for its channel
            static_for<0, BP / 8>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                float a4[4], b4[KT][4];
                static_for<0, 4>([&](auto rc) {
                    constexpr int r = decltype(rc)::value;
                    lds_read32<AO + (8 * g + r) * BM * 4>(a4[r], a_addr4);
                    static_for<0, KT>([&](auto t) {
                        lds_read32<BO + (decltype(t)::value * BP + 8 * g + r) * BC * 4>(b4[decltype(t)::value][r], b_addr4);
                    });
                });
                lds_wait<0>();
#pragma unroll
                for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(a4[r]));
#pragma unroll
                for (int t = 0; t < KT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) asm volatile("" : "+v"(b4[t][r]));
                const s16x4 ab = cvt_bf16x4(a4[0], a4[1], a4[2], a4[3]);
#pragma unroll
                for (int t = 0; t < KT; ++t)
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab, cvt_bf16x4(b4[t][0], b4[t][1], b4[t][2], b4[t][3]),
                                                                      acc[t], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            return;
        }
        if constexpr (MATH == 2) {
            // fp32-accurate products on the bf16 matrix cores: lane (i, h) holds pixels 8h .. 8h+7 of the stage's 16 for
            // its channel; the dy fragment is split once, the x fragment of every tap as it arrives (one tap ahead)
            float a8[8], b8[2][8];
            static_for<0, 8>([&](auto ec) { lds_read32<AO + decltype(ec)::value * BM * 4>(a8[decltype(ec)::value], a_addr8); });
            static_for<0, 8>([&](auto ec) { lds_read32<BO + decltype(ec)::value * BC * 4>(b8[0][decltype(ec)::value], b_addr8); });
            lds_wait<8>();
#pragma unroll
            for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(a8[e]));
            bf16x8 ah, am, al;
            split3_bf16(a8, ah, am, al);
            static_for<0, KT>([&](auto tc_) {
                constexpr int t = decltype(tc_)::value, pp = t & 1;
                if constexpr (t + 1 < KT) {
                    static_for<0, 8>([&](auto ec) {
                        lds_read32<BO + ((t + 1) * BP + decltype(ec)::value) * BC * 4>(b8[pp ^ 1][decltype(ec)::value], b_addr8);
                    });
                    lds_wait<8>();
                } else {
                    lds_wait<0>();
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(b8[pp][e]));
                bf16x8 bh, bm, bl;
                split3_bf16(b8[pp], bh, bm, bl);
                mfma_x3(acc[t], ah, am, al, bh, bm, bl);
                __builtin_amdgcn_sched_barrier(0);
            });
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
            return;
        }
        float af[2], bf[2][KT];
        lds_read32<AO>(af[0], a_addr);
        static_for<0, KT>([&](auto t) { lds_read32<BO + decltype(t)::value * BP * BC * 4>(bf[0][decltype(t)::value], b_addr); });
        static_for<0, BP / 2>([&](auto kc) {
            constexpr int kk = decltype(kc)::value, pp = kk & 1;
            if constexpr (kk + 1 < BP / 2) {
                lds_read32<AO + 2 * (kk + 1) * BM * 4>(af[pp ^ 1], a_addr);
                static_for<0, KT>([&](auto t) {
                    lds_read32<BO + (decltype(t)::value * BP + 2 * (kk + 1)) * BC * 4>(bf[pp ^ 1][decltype(t)::value], b_addr);
                });
                lds_wait<KT + 1>();
            } else {
                lds_wait<0>();
            }
            asm volatile("" : "+v"(af[pp]));
#pragma unroll
            for (int t = 0; t < KT; ++t) asm volatile("" : "+v"(bf[pp][t]));
#pragma unroll
            for (int t = 0; t < KT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pp], bf[pp][t], acc[t], 0, 0, 0);
            __builtin_amdgcn_sched_barrier(0);
        });
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    for (int it = 0; it < nIt; it += 2) {
        stage(std::integral_constant<int, 0>{}, it);
        if (it + 1 < nIt) stage(std::integral_constant<int, 1>{}, it + 1);
    }

    if (want_bias) {
        constexpr int GROUPS = NT / BM;                      // 2
        __syncthreads();
        As[0][tid] = bsum;
        __syncthreads();
        if (tid < BM && o0 + tid < a.Cout) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < GROUPS; ++g) t += As[0][g * BM + tid];
            a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = t;
        }
    }
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

// ------------------------------------------------------------------------------------------ wgrad, one tap, LDS-DMA
// One (tap, 64x64 or 128x128 output x input channel tile, pixel split) per workgroup; the dy / x tiles are brought in
// by global_load_lds (see conv_wgrad9g_kernel): the unpadded
// [pixel][channel] tiles are lane-linear images of 1-KiB wave loads, so no staging registers and no ds_write pass.
// KD > 1 (MATH 2 only): the waves divide the pixels of a stage as well as the tile -- BP = 16 KD pixels per stage, wave
// (kg, nh) accumulates ALL BM rows x its BN / (4 / KD) columns over pixel group kg, and the KD partial tiles are added through
// LDS after the loop (as TAG bit 5 of conv_igemmg_kernel: the operand splits and fragment reads of a wave serve TMA x TNA
// accumulator blocks instead of one)
template <int BM, int BN, int WM, int WN, int MATH, int BP = 16, int KD = 1>
__device__ __forceinline__ void wgradg_body(const WgradArgs& a, int id) {
    constexpr int NW = 4;                                   // BP = pixels per stage (16 or 32); splits count 16-pixel chunks
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr bool KW = KD > 1;
    constexpr int WNK = NW / KD;
    constexpr int TMA = KW ? BM / 32 : TM, TNA = KW ? BN / (32 * WNK) : TN;
    constexpr bool H2 = MATH == 3;                          // fp16 hi / lo arithmetic: 3 f16 MFMAs into an accumulator pair
    static_assert(!KW || ((MATH == 2 || MATH == 3) && BP == 16 * KD && TM == 1 && TN == 1 && TMA * TNA == KD), "pixel-divided tile");
    constexpr int A_INSTR = BP * BM * 4 / 1024, B_INSTR = BP * BN * 4 / 1024;
    constexpr int N_INSTR = A_INSTR + B_INSTR;
    constexpr int PER_WAVE = (N_INSTR + NW - 1) / NW;
    constexpr int RA = 256 / BM, RB = 256 / BN;             // tile rows per wave instruction
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) float As[2][BP * BM];
    __shared__ __attribute__((aligned(16))) float Bs[2][BP * BN];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;

    const int KT = a.KH * a.KW;
    const int tilesO = (a.Cout + BM - 1) / BM;
    const int tilesC = (a.Cin + BN - 1) / BN;
    const int tilesPerSplit = tilesO * tilesC * KT;
    const int split = id / tilesPerSplit;
    id -= split * tilesPerSplit;
    const int to = id % tilesO;
    id /= tilesO;
    const int tc = id % tilesC;
    const int tap = id / tilesC;
    const int o0 = to * BM, c0 = tc * BN;
    const int* tab_tap = a.rowtab ? a.rowtab + (size_t)tap * a.Mp : nullptr;     // (null: 1 x 1 / stride 1, pixel m reads row m)

    const int p_begin = split * a.chunks_per_split * 16;
    int p_end = p_begin + a.chunks_per_split * 16;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + BP - 1) / BP : 0;

    int brow[PER_WAVE];                                     // gather rows of the NEXT stage (x-tile instructions)
    bool bok[PER_WAVE];
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int bi = wave + k * NW - A_INSTR;
        brow[k] = -1;
        bok[k] = false;
        if (bi >= 0 && bi < B_INSTR) {
            const int m = p_begin + bi * RB + (lane * 4) / BN;
            brow[k] = tab_tap ? tab_tap[m < a.Mp ? m : a.Mp - 1] : m;   // unconditional (clamped) load; rows >= p_end are masked at use
            bok[k] = m < p_end;
        }
    }
    // Issue order inside a stage: (1) the x-tile loads, which consume the gather rows fetched one stage earlier, (2) the
    // dy-tile loads, (3) the gather rows of the next stage.  The compiler cannot see that the rows loaded in the previous
    // iteration were already drained by the barrier's vmcnt(0) and waits (vmcnt(0)) before their first use: placed first,
    // that wait is free; placed after a dy-tile load (the former order) it stalled every stage on its own prefetch.
    auto issue_stage = [&](int it, int buf) {
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
                const int c = c0 + (lane * 4) % BN;
                radet_lds_load16(a.x, (bok[k] && brow[k] >= 0 && c < a.Cin), (size_t)((size_t)brow[k] * a.Cin + c), (lptr_t)(&Bs[buf][bi * 256]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int m = p0 + ins * RA + (lane * 4) / BM;
                const int o = o0 + (lane * 4) % BM;
                radet_lds_load16(a.dy, (m < p_end && o < a.Cout), (size_t)((size_t)m * a.ld_dy + o), (lptr_t)(&As[buf][ins * 256]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                // unconditional (clamped) load; rows >= p_end are masked at use
                const int m = p0 + BP + (ins - A_INSTR) * RB + (lane * 4) / BN;
                brow[k] = tab_tap ? tab_tap[m < a.Mp ? m : a.Mp - 1] : m;
                bok[k] = m < p_end;
            }
        }
    };

    f32x16 acc[TMA][TNA];
    f32x16 acc1[H2 ? TMA : 1][H2 ? TNA : 1];
#pragma unroll
    for (int i = 0; i < TMA; ++i)
#pragma unroll
        for (int j = 0; j < TNA; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    H2Scale sdy = {1.f, 1.f, 1.f}, sxx = {1.f, 1.f, 1.f};
    if constexpr (H2) {
#pragma unroll
        for (int i = 0; i < TMA; ++i)
#pragma unroll
            for (int j = 0; j < TNA; ++j)
#pragma unroll
                for (int t = 0; t < 16; ++t) acc1[i][j][t] = 0.f;
    }
    unsigned raw_dy = 0u, raw_x = 0u;                          // slot gathers in front of the first tile loads, reduced behind them
    if constexpr (H2) { raw_dy = h2_scale_load(a.dys); raw_x = h2_scale_load(a.xss); }
    float bsum = 0.f;
    const int kg = wave % KD, nh = wave / KD;
    const bool want_bias = a.dbias_partials != nullptr && tap == 0 && tc == 0;

    // per-lane LDS byte addresses of the operand reads: pixel row lh of a k pair, channel (wave tile) * 32 + li
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + 4u * (unsigned)(lh * BM + wm * TM * 32 + li);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + 4u * (unsigned)(lh * BN + wn * TN * 32 + li);
    if (nIt > 0) issue_stage(0, 0);
    __syncthreads();
    if constexpr (H2) { sdy = h2_scale_finish(raw_dy); sxx = h2_scale_finish(raw_x); }
    for (int it = 0; it < nIt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nIt) issue_stage(it + 1, buf ^ 1);
        if constexpr (KW) {
            // pixel group kg of the stage, all TMA x TNA blocks of this wave's column group
            const unsigned ab = (unsigned)(size_t)(lptr_t)(&As[0][0]) + 4u * (unsigned)(lh * BM + li) + (unsigned)buf * (BP * BM * 4) +
                                7u * lh * BM * 4 + (unsigned)(kg * 16 * BM * 4);
            const unsigned bb = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + 4u * (unsigned)(lh * BN + nh * TNA * 32 + li) +
                                (unsigned)buf * (BP * BN * 4) + 7u * lh * BN * 4 + (unsigned)(kg * 16 * BN * 4);
            float a8[TMA][8], b8[TNA][8];
            static_for<0, TMA>([&](auto ic) {
                static_for<0, 8>([&](auto ec) {
                    lds_read32<(decltype(ec)::value * BM + decltype(ic)::value * 32) * 4>(a8[decltype(ic)::value][decltype(ec)::value], ab);
                });
            });
            static_for<0, TNA>([&](auto jc) {
                static_for<0, 8>([&](auto ec) {
                    lds_read32<(decltype(ec)::value * BN + decltype(jc)::value * 32) * 4>(b8[decltype(jc)::value][decltype(ec)::value], bb);
                });
            });
            lds_wait<0>();
#pragma unroll
            for (int i = 0; i < TMA; ++i)
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(a8[i][e]));
#pragma unroll
            for (int j = 0; j < TNA; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(b8[j][e]));
            if constexpr (H2) {
                f16x8 ah[TMA], al[TMA], bh[TNA], bl[TNA];
#pragma unroll
                for (int i = 0; i < TMA; ++i) split2_f16(a8[i], sdy, ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < TNA; ++j) split2_f16(b8[j], sxx, bh[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TMA; ++i)
#pragma unroll
                    for (int j = 0; j < TNA; ++j) mfma_h2(acc[i][j], acc1[i][j], ah[i], al[i], bh[j], bl[j]);
            } else {
            bf16x8 ah[TMA], am[TMA], al[TMA], bh[TNA], bm[TNA], bl[TNA];
#pragma unroll
            for (int i = 0; i < TMA; ++i) split3_bf16(a8[i], ah[i], am[i], al[i]);
#pragma unroll
            for (int j = 0; j < TNA; ++j) split3_bf16(b8[j], bh[j], bm[j], bl[j]);
#pragma unroll
            for (int i = 0; i < TMA; ++i)
#pragma unroll
                for (int j = 0; j < TNA; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (MATH == 2 || MATH == 3) {
            // fp32-accurate products on the bf16 matrix cores (see conv_igemmg_kernel, X3): lane (i, h) holds pixels
            // 8h .. 8h+7 of every 16-pixel group for its channel
            const unsigned ab = a_thr + (unsigned)buf * (BP * BM * 4) + 7u * lh * BM * 4;
            const unsigned bb = b_thr + (unsigned)buf * (BP * BN * 4) + 7u * lh * BN * 4;
            static_for<0, BP / 16>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                float a8[TM][8], b8[TN][8];
                static_for<0, TM>([&](auto ic) {
                    static_for<0, 8>([&](auto ec) {
                        lds_read32<((16 * g + decltype(ec)::value) * BM + decltype(ic)::value * 32) * 4>(a8[decltype(ic)::value][decltype(ec)::value], ab);
                    });
                });
                static_for<0, TN>([&](auto jc) {
                    static_for<0, 8>([&](auto ec) {
                        lds_read32<((16 * g + decltype(ec)::value) * BN + decltype(jc)::value * 32) * 4>(b8[decltype(jc)::value][decltype(ec)::value], bb);
                    });
                });
                lds_wait<0>();
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(a8[i][e]));
#pragma unroll
                for (int j = 0; j < TN; ++j)
#pragma unroll
                    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(b8[j][e]));
                if constexpr (H2) {
                    f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) split2_f16(a8[i], sdy, ah[i], al[i]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) split2_f16(b8[j], sxx, bh[j], bl[j]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) mfma_h2(acc[i][j], acc1[i][j], ah[i], al[i], bh[j], bl[j]);
                } else {
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) split3_bf16(a8[i], ah[i], am[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) split3_bf16(b8[j], bh[j], bm[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        } else if constexpr (MATH == 1) {
#pragma unroll
            for (int g = 0; g < BP / 8; ++g) {
                s16x4 ab[TM], bb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
                    const float* q = &As[buf][(8 * g + 4 * lh) * BM + (wm * TM + i) * 32 + li];
                    ab[i] = cvt_bf16x4(q[0], q[BM], q[2 * BM], q[3 * BM]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
                    const float* q = &Bs[buf][(8 * g + 4 * lh) * BN + (wn * TN + j) * 32 + li];
                    bb[j] = cvt_bf16x4(q[0], q[BN], q[2 * BN], q[3 * BN]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab[i], bb[j], acc[i][j], 0, 0, 0);
            }
        } else {
            // The operand reads are inline asm with hand-placed lgkmcnt waits: behind plain LDS loads the compiler puts
            // s_waitcnt vmcnt(0) (it cannot prove that the LDS-DMA just issued targets the OTHER buffer), which made every
            // stage wait for its own prefetch before the first MFMA -- load and compute of a workgroup ran back to back.
            const unsigned ab = a_thr + (unsigned)buf * (BP * BM * 4), bb = b_thr + (unsigned)buf * (BP * BN * 4);
            float af[2][TM], bf[2][TN];
            static_for<0, TM>([&](auto ic) { lds_read32<decltype(ic)::value * 128>(af[0][decltype(ic)::value], ab); });
            static_for<0, TN>([&](auto jc) { lds_read32<decltype(jc)::value * 128>(bf[0][decltype(jc)::value], bb); });
            static_for<0, BP / 2>([&](auto kc) {
                constexpr int kk = decltype(kc)::value, pp = kk & 1;
                if constexpr (kk + 1 < BP / 2) {
                    static_for<0, TM>([&](auto ic) {
                        lds_read32<(2 * (kk + 1) * BM + decltype(ic)::value * 32) * 4>(af[pp ^ 1][decltype(ic)::value], ab);
                    });
                    static_for<0, TN>([&](auto jc) {
                        lds_read32<(2 * (kk + 1) * BN + decltype(jc)::value * 32) * 4>(bf[pp ^ 1][decltype(jc)::value], bb);
                    });
                    lds_wait<TM + TN>();
                } else {
                    lds_wait<0>();
                }
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[pp][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bf[pp][j]));
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pp][i], bf[pp][j], acc[i][j], 0, 0, 0);
            });
        }
        if (want_bias) {
            constexpr int G = 256 / BM, RPT = BP / G;       // row groups, rows per thread
#pragma unroll
            for (int j = 0; j < RPT; ++j) bsum += As[buf][((tid / BM) * RPT + j) * BM + (tid % BM)];
        }
        __syncthreads();
    }
    if (want_bias) {
        constexpr int G = 256 / BM;
        __syncthreads();
        As[0][tid] = bsum;
        __syncthreads();
        if (tid < BM && o0 + tid < a.Cout) {
            float t = 0.f;
#pragma unroll
            for (int g = 0; g < G; ++g) t += As[0][g * BM + tid];
            a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = t;
        }
    }
    if constexpr (H2) {
#pragma unroll
        for (int i = 0; i < TMA; ++i)
#pragma unroll
            for (int j = 0; j < TNA; ++j) h2_combine(acc[i][j], acc1[i][j], sdy.inv, sxx.inv);
    }
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
    if constexpr (KW) {
        // KD partial tiles of a column group -> one (see conv_igemmg_kernel): wave (kg, nh) keeps block kg, ships the others
        constexpr int SLOTS_A = 2 * BP * BM / 1024, SLOTS_B = 2 * BP * BN / 1024;
        static_assert(NW * (KD - 1) <= SLOTS_A + SLOTS_B, "partial blocks fit the stage buffers");
        float* const pa = &As[0][0];
        float* const pb = &Bs[0][0];
        auto slot_ptr = [&](int sender_wave, int blk) {
            const int sk_ = sender_wave % KD;
            const int slot = sender_wave * (KD - 1) + blk - (blk > sk_ ? 1 : 0);
            return reinterpret_cast<f32x4*>(slot < SLOTS_A ? pa + slot * 1024 : pb + (slot - SLOTS_A) * 1024) + lane;
        };
        __syncthreads();                                        // (the bias column sums above read As[0])
        f32x16 res;
        static_for<0, KD>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const f32x16& v = acc[b / TNA][b % TNA];
            if (kg == b) {
                res = v;
            } else {
                f32x4* d = slot_ptr(wave, b);
#pragma unroll
                for (int q = 0; q < 4; ++q) d[q * 64] = f32x4{v[4 * q], v[4 * q + 1], v[4 * q + 2], v[4 * q + 3]};
            }
        });
        __syncthreads();
#pragma unroll
        for (int sg = 0; sg < KD; ++sg) {
            if (sg != kg) {
                const f32x4* d = slot_ptr(nh * KD + sg, kg);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    const f32x4 t = d[q * 64];
                    res[4 * q] += t.x; res[4 * q + 1] += t.y; res[4 * q + 2] += t.z; res[4 * q + 3] += t.w;
                }
            }
        }
        const int c = c0 + (nh * TNA + kg % TNA) * 32 + li;
        if (c < a.Cin) {
#pragma unroll
            for (int t = 0; t < 16; ++t) {
                const int o = o0 + (kg / TNA) * 32 + (t & 3) + 8 * (t >> 2) + 4 * lh;
                if (o < a.Cout) out[((size_t)o * KT + tap) * a.Cin + c] = res[t];
            }
        }
    } else {
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
}

template <int BM, int BN, int WM, int WN, int MATH, int BP = 16, int KD = 1>
__global__ __launch_bounds__(256) void conv_wgradg_kernel(const WgradArgs a) {
    radet_kernarg_warm<sizeof(WgradArgs)>();
    wgradg_body<BM, BN, WM, WN, MATH, BP, KD>(a, blockIdx.x);
}

// Grouped launch: up to WG_MAX independent weight-gradient GEMMs (the convs of one backbone stage / of the neck, all
// off the critical path of the backward chain) in ONE grid.  Each conv alone is a 150-1000 workgroup launch whose
// ramp-up and tail leave most of the 256 CUs idle, and filling the chip per conv needs many pixel splits (every
// split = one more weight-sized slab written here and read again by the reduction); a group keeps every CU busy with a
// few long workgroups per conv instead.  Problem descriptors travel in the kernel argument segment.
#define WG_MAX 32
struct WgradGroup {
    int n;
    int begin[WG_MAX + 1];      // first workgroup of problem i; begin[n] = grid size
    WgradArgs p[WG_MAX];
};

template <int BM, int BN, int WM, int WN, int MATH>
__global__ __launch_bounds__(256) void conv_wgradg_group_kernel(const WgradGroup g) {
    int pi = 0;
    for (int i = 1; i < g.n; ++i)
        if ((int)blockIdx.x >= g.begin[i]) pi = i;      // uniform: scalar compares on kernel arguments
    wgradg_body<BM, BN, WM, WN, MATH>(g.p[pi], (int)blockIdx.x - g.begin[pi]);
}

