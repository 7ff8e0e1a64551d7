// fp16 hi / lo arithmetic (common.h "h2"): the weight-gradient kernels -- the all-taps kernels on fp16 plane pairs
// (conv_wgrad9q_kernel and its round-6 variants), the one-tap pair kernel, and the launcher of the in-register-split one-tap
// instantiations of conv_wgrad_kernel.h.  Split from conv_h2.hip in round 6 so that the two halves compile side by side.
// gfx950 only.  Replaces cuDNN's weight gradients behind radet/models/backbones/resnet.py:260-299, necks/fpn.py:170-221,
// dense_heads/atss_head.py:118-145.
#include "common.h"
#include "../../include/radet_hip.h"
#include <stdlib.h>
#include <type_traits>

#include "conv_igemm_kernel.h"
#include "conv_wgrad_kernel.h"

// ------------------------------------------------------------------------------------------ wgrad, all 9 taps, fp16 plane pairs
// dW[o, tap, c] = sum_m dy[m, o] x[g(m, tap), c] with dy rows [2][ld_dy] and x rows [2][Cin] fp16 (32-channel groups
// [hi x 32 | lo x 32], written once by the GroupNorm kernels / radet_split_pairs).  conv_wgrad9p_kernel's data path
// ([4 pixels][16 channels] sub-tiles by LDS-DMA, ds_read_b64_tr_b16 operands) with two planes instead of three and an
// accumulator PAIR per tap: 18 accumulator blocks do not fit the registers of a wave that owns all nine taps of a 32 x 32
// block (8 waves = 256 registers each), so a workgroup owns 128 output x 32 input channels and a 32-channel output group is
// shared by TWO waves, taps 0-4 and 5-8 (160 / 128 accumulator registers).  Per 16-pixel stage: 8 KiB of dy + 18 KiB of x by
// LDS-DMA, 15 / 12 MFMAs per wave between barriers.
// SUB: 16-pixel sub-stages per pipeline stage (one wait + barrier per SUB x 16 pixels).  With one 8-wave workgroup per CU a
// 16-pixel stage is 0.19 us of MFMA work per wave behind ~1 us of load latency + barrier; two sub-stages per barrier amortise
// that (104 KiB of LDS: the workgroup owns the CU anyway).
#if RADET_P3_DBG
// (ablation builds only) in-kernel time stamps of workgroup 0: [wave 0 / wave 4][iteration][point] shader clocks
__device__ long long radet_dbg_clk[2][64][4];
extern "C" int radet_dbg_clk_read(long long* host, int n) {
    return (int)hipMemcpyFromSymbol(host, HIP_SYMBOL(radet_dbg_clk), (size_t)n * sizeof(long long));
}
#define RADET_STAMP(it, pt) do { if (blockIdx.x == 0 && (wave == 0 || wave == 4) && (it) < 64) { \
    const long long t_ = __builtin_amdgcn_s_memtime(); asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); \
    if (lane == 0) radet_dbg_clk[wave >> 2][(it)][(pt)] = t_; } } while (0)
#else
#define RADET_STAMP(it, pt) do { } while (0)
#endif
template <int SUB, bool SPREAD = false, bool TGLOAD = false>
__global__ __launch_bounds__(512) void conv_wgrad9q_kernel(const WgradArgs a) {
    radet_kernarg_warm<sizeof(WgradArgs)>();
    constexpr int BP = 16, NW = 8, BM = 128, BC = 32, KT = 9;
    constexpr int CBA = BM / 16, CBB = BC / 16;
    constexpr int A_PL = BP * BM, B_PL = BP * BC;           // fp16 elements per dy plane tile / per (tap, plane) x tile
    constexpr int A_Q = A_PL * 2 / 1024;                    // wave loads per dy plane tile: 4
    constexpr int A_INSTR = 2 * A_Q;                        // 8
    constexpr int B_INSTR = KT * 2;                         // 18: one wave load per (tap, plane)
    constexpr int N_INSTR = A_INSTR + B_INSTR;              // 26
    // TGLOAD (round 6): ALL tile loads are issued by the four waves of tap group 1 (taps 5-8: four taps against five, so each has
    // six MFMAs per stage less to issue), seven per sub-stage each; the waves of tap group 0 -- one per SIMD, next to one of the
    // others -- never stand in the vector-memory issue queue and keep the matrix pipe busy meanwhile
    constexpr int LW = TGLOAD ? 4 : NW;                     // waves that issue loads
    constexpr int PER_WAVE = (N_INSTR + LW - 1) / LW;       // 4 (7) loads per wave and sub-stage
    static_assert(B_PL * 2 == 1024, "one wave load per x tile");
    __shared__ __attribute__((aligned(16))) unsigned short As[2 * SUB][2 * A_PL];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2 * SUB][KT * 2 * B_PL];
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy);
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(a.x);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = wave & 3, tg = wave >> 2;                // output-channel group, tap group (0: taps 0-4, 1: taps 5-8)
    const int lbase = TGLOAD ? (wave >= 4 ? wave - 4 : N_INSTR) : wave;       // first load of this wave (N_INSTR: none)
    const int li = lane & 31, lh = lane >> 5;

    const int tilesO = (a.Cout + BM - 1) / BM;
    const int tilesC = a.Cin / BC;
    const int tilesPerSplit = tilesO * tilesC;
    // XCD-aware order: the channel tiles of one pixel split run next to each other on ONE XCD (conv_wgrad9p_kernel)
    int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);
    const int split = id / tilesPerSplit;
    id -= split * tilesPerSplit;
    const int to = id % tilesO, tc = id / tilesO;
    const int o0 = to * BM, c0 = tc * BC;

    const int p_begin = split * a.chunks_per_split * 16;
    int p_end = p_begin + a.chunks_per_split * 16;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + SUB * BP - 1) / (SUB * BP) : 0;

    const int l_blk = lane >> 3, l_prow = (lane & 7) >> 1, l_half = lane & 1;
    // x-tile load bi = tap * 2 + plane: the 8 sub-tiles [4 pixel quads][2 channel blocks] of that tap and plane
    int brow[SUB][PER_WAVE];
    bool bok[SUB][PER_WAVE];
#pragma unroll
    for (int s = 0; s < SUB; ++s)
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int bi = lbase + k * LW - A_INSTR;
            brow[s][k] = -1;
            bok[s][k] = false;
            if (bi >= 0 && bi < B_INSTR) {
                const int m = p_begin + s * BP + 4 * (l_blk / CBB) + l_prow;
                brow[s][k] = a.rowtab[(size_t)(bi / 2) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
                bok[s][k] = m < p_end;
            }
        }
    // the loads of stage `it` into buffer `buf`, as individually issuable pieces: piece (s, k) = the k-th wave load of sub-stage s
    // (ins = wave + 8 k: a dy tile for ins < 8, else the x tile of (tap, plane) ins - 8), then the gather rows of the stage after
    auto pin_rows = [&]() {
        // the gather rows were fetched one stage ago and drained by the barrier's vmcnt(0), which the compiler cannot see: left
        // alone it puts a vmcnt(0) in front of every x-tile load that reads brow[k] -- and from the second one on that wait
        // covers the LDS-DMA load issued just before it: the pieces of a stage went out one round trip apart.  One wait here
        // (free), and the rows are plain registers afterwards.
#pragma unroll
        for (int s = 0; s < SUB; ++s)
#pragma unroll
            for (int k = 0; k < PER_WAVE; ++k) asm volatile("" : "+v"(brow[s][k]));
    };
    auto issue_piece = [&](int it, int buf, auto sc, auto kc) {
        constexpr int s = decltype(sc)::value, k = decltype(kc)::value;
        const int p0 = p_begin + (it * SUB + s) * BP;
        const int ins = lbase + k * LW;
        if (ins >= A_INSTR && ins < N_INSTR) {
            const int bi = ins - A_INSTR;
            const int c = c0 + 16 * (l_blk % CBB) + 8 * l_half;
            radet_lds_load16(xh, (bok[s][k] && brow[s][k] >= 0), (size_t)((size_t)brow[s][k] * 2 * a.Cin + radet_pair_off(c) + 32 * (bi % 2)), (lptr_t)(&Bs[buf * SUB + s][bi * B_PL]));
        } else if (ins < A_INSTR) {
            const int pl = ins / A_Q, blk = (ins % A_Q) * 8 + l_blk;
            const int m = p0 + 4 * (blk / CBA) + l_prow;
            const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
            radet_lds_load16(dyh, (m < p_end && o < a.Cout), (size_t)((size_t)m * 2 * a.ld_dy + radet_pair_off(o) + 32 * pl), (lptr_t)(&As[buf * SUB + s][ins * 512]));
        }
    };
    auto next_rows = [&](int it) {
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            const int p0 = p_begin + (it * SUB + s) * BP;
#pragma unroll
            for (int k = 0; k < PER_WAVE; ++k) {
                const int ins = lbase + k * LW;
                if (ins >= A_INSTR && ins < N_INSTR) {
                    const int bi = ins - A_INSTR;
                    const int m = p0 + SUB * BP + 4 * (l_blk / CBB) + l_prow;
                    brow[s][k] = a.rowtab[(size_t)(bi / 2) * a.Mp + (m < a.Mp ? m : a.Mp - 1)];
                    bok[s][k] = m < p_end;
                }
            }
        }
    };
    auto issue_stage = [&](int it, int buf) {                // everything at once: x tiles, dy tiles (per sub-stage), next gather rows
        pin_rows();
        static_for<0, SUB>([&](auto sc) {
            static_for<1, PER_WAVE>([&](auto kc) { issue_piece(it, buf, sc, kc); });
            issue_piece(it, buf, sc, std::integral_constant<int, 0>{});
        });
        next_rows(it);
    };

    constexpr int NTAP = 5;                                  // accumulator pairs per wave (tap group 1 leaves the last one idle)
    f32x16 acc[NTAP], acc1[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acc1[t][r] = 0.f; }
    const unsigned raw_dy = h2_scale_load(a.dys), raw_x = h2_scale_load(a.xss);   // reduced behind the first tiles' wait
    float bsum = 0.f;
    const bool want_bias = a.dbias_partials != nullptr && tc == 0;
    const int g16 = (lane >> 4) & 1, m16 = lane & 15;
    const int tap0 = tg * 5, ntap = tg ? 4 : 5;
    // per-lane LDS byte addresses of the (inline-asm) transposing reads, see conv_wgradh: pixel quad 2 lh (+ 1), channel
    // sub-tile of the wave's 32 channels + g16, bytes 8 m16 of the sub-tile
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(((2 * lh) * CBA + og * 2 + g16) * 128 + m16 * 8);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)(((2 * lh) * CBB + g16) * 128 + m16 * 8) +
                           (unsigned)(tap0 * 2 * B_PL * 2);

    if (nIt > 0) issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const H2Scale sdy = h2_scale_finish(raw_dy), sxx = h2_scale_finish(raw_x);
    for (int it = 0; it < nIt; ++it) {
        const int buf = it & 1;
        RADET_STAMP(it, 0);
        const bool do_issue = it + 1 < nIt && (!RADET_P3_DBG || !(a.dbg & 1));      // (ablation: RADET_DBG_WGRAD bits 1 / 2 / 4)
        // SPREAD (round 6): the loads of the next stage go out ONE AT A TIME behind the MFMAs of a tap instead of all at the head
        // of the iteration.  In-kernel time stamps (s_memtime in a -DRADET_P3_DBG=1 build) put the ISSUE of the eight wave loads
        // of an iteration at 1300-1450 of its 4600 clocks -- all eight waves of the CU stand in the vector-memory issue queue
        // at the same time right behind the barrier, and the matrix pipe idles until the first of them gets through
        if constexpr (SPREAD) { if (do_issue) pin_rows(); }
        else if (do_issue) issue_stage(it + 1, buf ^ 1);
        RADET_STAMP(it, 1);
        static_for<0, SUB>([&](auto sc_) {
        constexpr int s = decltype(sc_)::value;
        const unsigned ab = a_thr + (unsigned)(buf * SUB + s) * (2 * A_PL * 2), bb = b_thr + (unsigned)(buf * SUB + s) * (KT * 2 * B_PL * 2);
        s16x4v_ al[2], ah[2], bl[2][2], bh[2][2];
        static_for<0, 2>([&](auto pc) {
            constexpr int pl = decltype(pc)::value;
            lds_read_tr16<pl * A_PL * 2>(al[pl], ab);
            lds_read_tr16<pl * A_PL * 2 + CBA * 128>(ah[pl], ab);
        });
        static_for<0, 2>([&](auto pc) {
            constexpr int pl = decltype(pc)::value;
            lds_read_tr16<pl * B_PL * 2>(bl[0][pl], bb);
            lds_read_tr16<pl * B_PL * 2 + CBB * 128>(bh[0][pl], bb);
        });
        f16x8 af[2];
        static_for<0, NTAP>([&](auto tc_) {
            constexpr int t = decltype(tc_)::value, pp = t & 1;
            if (t < ntap) {                                      // uniform per wave
                if (t + 1 < ntap) {
                    if constexpr (t + 1 < NTAP) {
                        if (!RADET_P3_DBG || !(a.dbg & 4))
                        static_for<0, 2>([&](auto pc) {
                            constexpr int pl = decltype(pc)::value;
                            lds_read_tr16<((t + 1) * 2 + pl) * B_PL * 2>(bl[pp ^ 1][pl], bb);
                            lds_read_tr16<((t + 1) * 2 + pl) * B_PL * 2 + CBB * 128>(bh[pp ^ 1][pl], bb);
                        });
                    }
                    lds_wait<4>();
                } else {
                    lds_wait<0>();
                }
                if constexpr (t == 0) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        asm volatile("" : "+v"(al[pl])); asm volatile("" : "+v"(ah[pl]));
                        af[pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(al[pl], ah[pl], 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                }
                f16x8 bf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    asm volatile("" : "+v"(bl[pp][pl])); asm volatile("" : "+v"(bh[pp][pl]));
                    bf[pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(bl[pp][pl], bh[pp][pl], 0, 1, 2, 3, 4, 5, 6, 7));
                }
                if (!RADET_P3_DBG || !(a.dbg & 2)) mfma_h2(acc[t], acc1[t], af[0], af[1], bf[0], bf[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if constexpr (SPREAD && t < PER_WAVE) {               // piece t of sub-stage s of the next stage (x tiles first)
                if (do_issue) issue_piece(it + 1, buf ^ 1, sc_, std::integral_constant<int, (t + 1) % PER_WAVE>{});
                __builtin_amdgcn_sched_barrier(0);
            }
        });
        if (want_bias && tid < BM) {                        // column sums of dy, pixel order, in units of 2^-e (scaled back below)
            // (inline-asm LDS reads: behind plain ones the compiler drains vmcnt, i.e. waits for the loads just issued)
            unsigned vh[BP], vl[BP];
            const unsigned sa = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)((((tid >> 4) & 7) * 64 + (tid & 15)) * 2) +
                                (unsigned)(buf * SUB + s) * (2 * A_PL * 2);
            static_for<0, BP>([&](auto pc) {
                constexpr int p = decltype(pc)::value, e = ((p >> 2) * CBA) * 64 + (p & 3) * 16;
                lds_read_u16<e * 2>(vh[p], sa);
                lds_read_u16<(A_PL + e) * 2>(vl[p], sa);
            });
            lds_wait<0>();
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                asm volatile("" : "+v"(vh[p]), "+v"(vl[p]));
                bsum += radet_pair_value((unsigned short)vh[p], (unsigned short)vl[p]);
            }
        }
        });
        if constexpr (SPREAD) { if (do_issue) next_rows(it + 1); }
        RADET_STAMP(it, 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        RADET_STAMP(it, 3);
        __syncthreads();
    }
    if (want_bias && tid < BM && o0 + tid < a.Cout) a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = bsum * sdy.inv;
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
    const int c = c0 + li;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        if (t < ntap) {
            h2_combine(acc[t], acc1[t], sdy.inv, sxx.inv);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + og * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < a.Cout) out[((size_t)o * KT + tap0 + t) * a.Cin + c] = acc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad, all 9 taps, pairs, deep pipeline
// conv_wgrad9q_kernel with D stage buffers instead of two (round 6).  An ablation of that kernel on the tower shape (RADET_DBG_WGRAD
// bits in a -DRADET_P3_DBG=1 build: 119 us; without tile loads 93, without MFMAs 99, without fragment reads 111, without all
// three 64) says that more than half of it is the SKELETON: every iteration ends in `s_waitcnt vmcnt(0)` + barrier behind loads
// and gather-table look-ups it issued itself, i.e. one global round trip (~1.2 us) per 32 pixels whatever else happens, with
// one 8-wave workgroup per CU and nothing to switch to.  Here:
//   * the rows of the gather table this workgroup needs (9 taps x its pixel split, as 16-bit row - pixel differences: a
//     unit-stride 3 x 3 with padding 1 reads within one image row of the pixel) are copied to LDS once, so the look-ups are
//     LDS reads (lgkmcnt) and no vector-memory result has to come back in the loop: in-order return would tie the wait for a
//     look-up to every tile load issued before it;
//   * stage i + D - 1 is issued while stage i is computed; a stage's loads are waited for with `vmcnt((D - 2) x loads per
//     stage of this wave)` (every wave issues a fixed 4 or 3 per stage) and published with the bare barrier
//     (radet_pipe_barrier, conv_common.h); one 16-pixel stage per barrier;
//   * the bias column sums read their dy values with inline-asm LDS reads (behind plain LDS loads the compiler drains vmcnt).
// Same products in the same order as conv_wgrad9q_kernel: bit-identical.  LDS: D x 26 KiB + 29 KiB of table (D = 5: 159 KiB).
#define RADET_W9D_TCAP 1664            // pixels per split the LDS copy of the table holds (the launcher falls back beyond)
template <int D>
__global__ __launch_bounds__(512) void conv_wgrad9d_kernel(const WgradArgs a) {
    radet_kernarg_warm<sizeof(WgradArgs)>();
    constexpr int BP = 16, NW = 8, BM = 128, BC = 32, KT = 9, TCAP = RADET_W9D_TCAP;
    constexpr int CBA = BM / 16, CBB = BC / 16;
    constexpr int A_PL = BP * BM, B_PL = BP * BC;           // fp16 elements per dy plane tile / per (tap, plane) x tile
    constexpr int A_Q = A_PL * 2 / 1024;                    // wave loads per dy plane tile: 4
    constexpr int A_INSTR = 2 * A_Q;                        // 8
    constexpr int B_INSTR = KT * 2;                         // 18: one wave load per (tap, plane)
    constexpr int N_INSTR = A_INSTR + B_INSTR;              // 26
    constexpr int PER_WAVE = (N_INSTR + NW - 1) / NW;       // 4 (waves 0, 1) or 3 loads per wave and stage
    static_assert(N_INSTR == 3 * NW + 2 && D >= 3, "waves 0 and 1 issue four loads per stage, the others three");
    __shared__ __attribute__((aligned(16))) unsigned short As[D][2 * A_PL];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[D][KT * 2 * B_PL];
    __shared__ short tabL[KT * TCAP + 16];
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy);
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(a.x);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = wave & 3, tg = wave >> 2;                // output-channel group, tap group (0: taps 0-4, 1: taps 5-8)
    const int li = lane & 31, lh = lane >> 5;

    const int tilesO = (a.Cout + BM - 1) / BM;
    const int tilesC = a.Cin / BC;
    const int tilesPerSplit = tilesO * tilesC;
    int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);    // (the channel tiles of one pixel split next to each other on one XCD)
    const int split = id / tilesPerSplit;
    id -= split * tilesPerSplit;
    const int to = id % tilesO, tc = id / tilesO;
    const int o0 = to * BM, c0 = tc * BC;

    const int p_begin = split * a.chunks_per_split * 16;
    int p_end = p_begin + a.chunks_per_split * 16;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + BP - 1) / BP : 0;

    const unsigned raw_dy = h2_scale_load(a.dys), raw_x = h2_scale_load(a.xss);   // (gathers; reduced behind the table's barrier)
    // the table rows of this split -> LDS (row - pixel, or -32768 for a padding tap)
    for (int j = tid; j < p_end - p_begin; j += NW * 64) {
        int r[KT];
#pragma unroll
        for (int t = 0; t < KT; ++t) r[t] = a.rowtab[(size_t)t * a.Mp + p_begin + j];
#pragma unroll
        for (int t = 0; t < KT; ++t) tabL[t * TCAP + j] = r[t] < 0 ? (short)-32768 : (short)(r[t] - (p_begin + j));
    }
    const int l_blk = lane >> 3, l_prow = (lane & 7) >> 1, l_half = lane & 1;
    const int pix = 4 * (l_blk / CBB) + l_prow;             // this lane's pixel of a stage in the x-tile loads
    unsigned tb[PER_WAVE];                                   // LDS byte address of its table entry for stage 0, per x load
#pragma unroll
    for (int k = 0; k < PER_WAVE; ++k) {
        const int bi = wave + k * NW - A_INSTR;
        tb[k] = (unsigned)(size_t)(lptr_t)(&tabL[0]) + (unsigned)((((bi >= 0 && bi < B_INSTR) ? bi / 2 : 0) * TCAP + pix) * 2);
    }
    int drow[PER_WAVE];                                      // look-ups of the stage that is issued next
    auto lookup = [&](int stage) {
        static_for<1, PER_WAVE>([&](auto kc) { lds_read_i16<0>(drow[decltype(kc)::value], tb[decltype(kc)::value] + (unsigned)(stage * BP * 2)); });
    };
    auto issue_stage = [&](int it, int buf) {                // x tiles (their look-ups were read one issue ago), dy tiles, next look-ups
        lds_wait<0>();
#pragma unroll
        for (int k = 1; k < PER_WAVE; ++k) asm volatile("" : "+v"(drow[k]));
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 1; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < N_INSTR) {                                 // (uniform: k = 3 exists for waves 0 and 1)
                const int bi = ins - A_INSTR;
                const int c = c0 + 16 * (l_blk % CBB) + 8 * l_half;
                const int m = p0 + pix;
                const bool ok = drow[k] != -32768 && m < p_end;
                radet_lds_load16(xh, ok, (size_t)((size_t)(m + drow[k]) * 2 * a.Cin + radet_pair_off(c) + 32 * (bi % 2)), (lptr_t)(&Bs[buf][bi * B_PL]));
            }
        }
        {
            const int ins = wave;                                // k = 0: every wave owns one dy load
            const int pl = ins / A_Q, blk = (ins % A_Q) * 8 + l_blk;
            const int m = p0 + 4 * (blk / CBA) + l_prow;
            const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
            radet_lds_load16(dyh, (m < p_end && o < a.Cout), (size_t)((size_t)m * 2 * a.ld_dy + radet_pair_off(o) + 32 * pl), (lptr_t)(&As[buf][ins * 512]));
        }
        if (it + 1 < nIt) lookup(it + 1);                        // (uniform)
    };

    constexpr int NTAP = 5;                                  // accumulator pairs per wave (tap group 1 leaves the last one idle)
    f32x16 acc[NTAP], acc1[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acc1[t][r] = 0.f; }
    float bsum = 0.f;
    const bool want_bias = a.dbias_partials != nullptr && tc == 0;
    const int g16 = (lane >> 4) & 1, m16 = lane & 15;
    const int tap0 = tg * 5, ntap = tg ? 4 : 5;
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(((2 * lh) * CBA + og * 2 + g16) * 128 + m16 * 8);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)(((2 * lh) * CBB + g16) * 128 + m16 * 8) +
                           (unsigned)(tap0 * 2 * B_PL * 2);
    const unsigned s_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)((((tid >> 4) & 7) * 64 + (tid & 15)) * 2);   // bias: channel tid of pixel 0

    __syncthreads();                                             // the table is complete (nothing else is in flight yet)
    const H2Scale sdy = h2_scale_finish(raw_dy), sxx = h2_scale_finish(raw_x);     // (drained with the table's loads)
    lookup(0);
#pragma unroll
    for (int k = 0; k < D - 1; ++k)
        if (k < nIt) issue_stage(k, k);
    int cb_ = 0, ib_ = D - 1;                                    // buffer of the stage computed / issued next
    for (int it = 0; it < nIt; ++it) {
        // stage `it` has landed once at most the D - 2 stages issued behind it are outstanding (in-order return; the last stages
        // of the split: plain wait)
        if (it + D - 2 < nIt) {
            if (wave < 2) vm_wait<(D - 2) * 4>(); else vm_wait<(D - 2) * 3>();
        } else {
            vm_wait<0>();
        }
        if (!RADET_P3_DBG || !(a.dbg & 8)) radet_pipe_barrier();        // ... for every wave, and buffer ib_ (stage it - 1) is free
        if (it + D - 1 < nIt && (!RADET_P3_DBG || !(a.dbg & 1))) issue_stage(it + D - 1, ib_);      // (ablation: RADET_DBG_WGRAD bits 1 / 2 / 4 / 8)
        const unsigned ab = a_thr + (unsigned)cb_ * (2 * A_PL * 2), bb = b_thr + (unsigned)cb_ * (KT * 2 * B_PL * 2);
        s16x4v_ al[2], ah[2], bl[2][2], bh[2][2];
        static_for<0, 2>([&](auto pc) {
            constexpr int pl = decltype(pc)::value;
            lds_read_tr16<pl * A_PL * 2>(al[pl], ab);
            lds_read_tr16<pl * A_PL * 2 + CBA * 128>(ah[pl], ab);
        });
        static_for<0, 2>([&](auto pc) {
            constexpr int pl = decltype(pc)::value;
            lds_read_tr16<pl * B_PL * 2>(bl[0][pl], bb);
            lds_read_tr16<pl * B_PL * 2 + CBB * 128>(bh[0][pl], bb);
        });
        f16x8 af[2];
        static_for<0, NTAP>([&](auto tc_) {
            constexpr int t = decltype(tc_)::value, pp = t & 1;
            if (t < ntap) {                                      // uniform per wave
                if (t + 1 < ntap) {
                    if constexpr (t + 1 < NTAP) {
                        if (!RADET_P3_DBG || !(a.dbg & 4))
                        static_for<0, 2>([&](auto pc) {
                            constexpr int pl = decltype(pc)::value;
                            lds_read_tr16<((t + 1) * 2 + pl) * B_PL * 2>(bl[pp ^ 1][pl], bb);
                            lds_read_tr16<((t + 1) * 2 + pl) * B_PL * 2 + CBB * 128>(bh[pp ^ 1][pl], bb);
                        });
                    }
                    lds_wait<4>();
                } else {
                    lds_wait<0>();
                }
                if constexpr (t == 0) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        asm volatile("" : "+v"(al[pl])); asm volatile("" : "+v"(ah[pl]));
                        af[pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(al[pl], ah[pl], 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                }
                f16x8 bf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    asm volatile("" : "+v"(bl[pp][pl])); asm volatile("" : "+v"(bh[pp][pl]));
                    bf[pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(bl[pp][pl], bh[pp][pl], 0, 1, 2, 3, 4, 5, 6, 7));
                }
                if (!RADET_P3_DBG || !(a.dbg & 2)) mfma_h2(acc[t], acc1[t], af[0], af[1], bf[0], bf[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (want_bias && tid < BM) {                        // column sums of dy, pixel order, in units of 2^-e (scaled back below)
            unsigned vh[BP], vl[BP];
            const unsigned sa = s_thr + (unsigned)cb_ * (2 * A_PL * 2);
            static_for<0, BP>([&](auto pc) {
                constexpr int p = decltype(pc)::value, e = ((p >> 2) * CBA) * 64 + (p & 3) * 16;
                lds_read_u16<e * 2>(vh[p], sa);
                lds_read_u16<(A_PL + e) * 2>(vl[p], sa);
            });
            lds_wait<0>();
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                asm volatile("" : "+v"(vh[p]), "+v"(vl[p]));
                bsum += radet_pair_value((unsigned short)vh[p], (unsigned short)vl[p]);
            }
        }
        cb_ = cb_ + 1 == D ? 0 : cb_ + 1;
        ib_ = ib_ + 1 == D ? 0 : ib_ + 1;
    }
    if (want_bias && tid < BM && o0 + tid < a.Cout) a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = bsum * sdy.inv;
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
    const int c = c0 + li;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        if (t < ntap) {
            h2_combine(acc[t], acc1[t], sdy.inv, sxx.inv);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + og * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < a.Cout) out[((size_t)o * KT + tap0 + t) * a.Cin + c] = acc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad, all 9 taps, pairs, x as shifted windows
// conv_wgrad9q_kernel for unit-stride 3 x 3 convs with padding 1 (round 6).  There the x tile of a 16-pixel stage is nine gathered
// copies of (almost) the same rows -- tap (r, q) of pixel m reads the row that tap (r, 1) of pixel m + q - 1 reads, unless
// column(m) + q - 1 leaves the image, where the tap is padding -- and every tile arrives as 64-byte pieces (one plane of one
// 32-channel group of one pixel): 26 wave loads = 416 half cache lines per stage.  That kernel is bound by the NUMBER of those
// requests, not by their bytes or by the matrix pipe (0.34 of it): a first version of this kernel that fetched 15 instead of 26
// KiB per stage, but the x part in 32-byte pieces, was no faster.  Here everything arrives as whole lines:
//   x   per tap ROW r ONE segment of 18 pixels -- rows g(m - 1 .. m + 16, tap (r, 1)) -- each pixel as the 128 contiguous
//       bytes [hi x 32 | lo x 32] of its 32-channel group: 54 lines (7 wave loads) per stage.  Tap (r, q) reads the window that
//       starts q pixels in: a ds_read_b64_tr_b16 takes a per-lane address, so a window is the same read 128 q bytes further on.
//   dy  each pixel as the 512 contiguous bytes of the tile's four 32-channel groups: 64 lines (8 wave loads) per stage.
// 118 lines instead of 416 half lines.  An LDS-DMA wave load writes 1 KiB lane-linearly, so the pixel pitch is 128 / 512 bytes
// -- a multiple of the 256-byte bank row: the 16-byte slots of a pixel are XOR-swizzled by its index on the SOURCE side (x:
// the plane bit by bit 1 of the pixel slot; dy: slot bits 2-3 by the pixel's low bits) so that the four pixels of a
// transposing read fall into different banks; the readers apply the same XOR (a per-lane constant).
// Where a tap is padding although its neighbour's row exists (q = 0 at column 0, q = 2 at the last column: the window would
// deliver the end of the previous / start of the next image row) the operand's pixel is zeroed in registers: two table
// look-ups per pixel and stage (taps (1, 0) and (1, 2): -1 there and only there), one ballot, 16-bit masks over the packed
// fragment, built on the scalar unit -- only in stages that contain such a pixel (wave-uniform branch).  Same products in the
// same order as conv_wgrad9q_kernel: bit-identical results (tests/test_gpu_kernels.py).
template <int SUB>
__global__ __launch_bounds__(512) void conv_wgrad9r_kernel(const WgradArgs a) {
    radet_kernarg_warm<sizeof(WgradArgs)>();
    constexpr int BP = 16, NW = 8, BM = 128, BC = 32, KT = 9;
    constexpr int A_EL = BP * 256;                          // fp16 elements of the dy image: 16 pixels x 512 bytes
    constexpr int A_INSTR = 8;                              // wave loads of it (two pixels each)
    constexpr int XPX = 18, B_UNITS = 3 * XPX;              // pixels per tap-row segment; 128-byte pixel units of the x image: 54
    constexpr int B_INSTR = (B_UNITS + 7) / 8;              // 7 wave loads (eight pixel units each)
    constexpr int B_EL = B_INSTR * 512;                     // fp16 elements of the x image (7 KiB)
    constexpr int ROWB = XPX * 128;                         // bytes per tap-row segment
    static_assert(A_INSTR == NW && B_INSTR <= NW && ROWB % 256 == 0, "one dy and at most one x load per wave; segments start on a bank row");
    // (256-byte alignment: the readers' XORs and the bank analysis above assume images that start on a bank row)
    __shared__ __attribute__((aligned(256))) unsigned short As[2 * SUB][A_EL];
    __shared__ __attribute__((aligned(256))) unsigned short Bs[2 * SUB][B_EL];
    const unsigned short* dyh = reinterpret_cast<const unsigned short*>(a.dy);
    const unsigned short* xh = reinterpret_cast<const unsigned short*>(a.x);

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int og = wave & 3, tg = wave >> 2;                // output-channel group, tap group (0: taps 0-4, 1: taps 5-8)
    const int li = lane & 31, lh = lane >> 5;

    const int tilesO = (a.Cout + BM - 1) / BM;
    const int tilesC = a.Cin / BC;
    const int tilesPerSplit = tilesO * tilesC;
    int id = xcd_remap((int)blockIdx.x, (int)gridDim.x);    // (the channel tiles of one pixel split next to each other on one XCD)
    const int split = id / tilesPerSplit;
    id -= split * tilesPerSplit;
    const int to = id % tilesO, tc = id / tilesO;
    const int o0 = to * BM, c0 = tc * BC;

    const int p_begin = split * a.chunks_per_split * 16;
    int p_end = p_begin + a.chunks_per_split * 16;
    if (p_end > a.M) p_end = a.M;
    const int nIt = p_begin < p_end ? (p_end - p_begin + SUB * BP - 1) / (SUB * BP) : 0;

    // this lane's share of the wave's x load: pixel unit 8 wave + lane / 8 = (tap row, pixel slot), LDS slot lane % 8 of its 128
    // bytes <- source slot (lane % 8) ^ 4 [pixel slot bit 1]  (slot = plane * 4 + 16-channel block * 2 + 8-channel half)
    const int xU = 8 * wave + (lane >> 3);
    const int xr = xU / XPX, xpx = xU - xr * XPX;
    const bool xuse = xU < B_UNITS;
    const int xtab = (xr * 3 + 1) * a.Mp;                   // tap (r, 1) of the segment's tap row (table offsets: [9][Mp] ints)
    const unsigned xcol = (unsigned)((c0 >> 5) * 64 + (((lane & 7) ^ ((xpx & 2) << 1)) << 3));     // element offset inside a pair row
    const int mtab = ((lane & 16) ? 5 : 3) * a.Mp;          // padding masks: lanes 0-15 tap (1, 0) of the stage's pixels, lanes 16-31 tap (1, 2)
    auto xrow_of = [&](int p0) {                             // source row of this lane's unit for a stage that starts at pixel p0, or -1
        const int m = p0 + xpx - 1;
        const int r = a.rowtab[xtab + (m < 0 ? 0 : (m < a.Mp ? m : a.Mp - 1))];
        return (xuse && m >= 0 && m < a.M) ? r : -1;
    };
    // ... and of its dy load: pixel 2 wave + lane / 32, LDS slot lane % 32 of its 512 bytes <- source slot (lane % 32) ^ (pixel % 4) << 2
    // (slot = 32-channel group * 8 + plane * 4 + 16-channel block * 2 + 8-channel half)
    const int ypx = 2 * wave + (lane >> 5);
    const int yslot = (lane & 31) ^ ((ypx & 3) << 2);
    const bool yuse = o0 + 32 * (yslot >> 3) + 16 * ((yslot >> 1) & 1) + 8 * (yslot & 1) < a.Cout;
    const unsigned ycol = (unsigned)((o0 >> 5) * 64 + yslot * 8);
    int brow[SUB], vmn[SUB];
#pragma unroll
    for (int s = 0; s < SUB; ++s) { brow[s] = xrow_of(p_begin + s * BP); vmn[s] = -1; }
    auto issue_stage = [&](int it, int buf) {                // order: x image, dy image, then the look-ups of later stages
        // (brow was fetched one stage ago and drained by the barrier's vmcnt(0), which the compiler cannot see: one free wait here
        // instead of one in front of the load that consumes it, see conv_wgrad9q_kernel)
#pragma unroll
        for (int s = 0; s < SUB; ++s) asm volatile("" : "+v"(brow[s]));
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            const int p0 = p_begin + (it * SUB + s) * BP;
            if (wave < B_INSTR)                                  // (uniform)
                radet_lds_load16(xh, (brow[s] >= 0), (size_t)((size_t)brow[s] * 2 * a.Cin + xcol), (lptr_t)(&Bs[buf * SUB + s][wave * 512]));
            const int m = p0 + ypx;
            radet_lds_load16(dyh, (yuse && m < p_end), (size_t)((size_t)m * 2 * a.ld_dy + ycol), (lptr_t)(&As[buf * SUB + s][wave * 512]));
            const int pm = p0 + (lane & 15);
            vmn[s] = a.rowtab[mtab + (pm < a.Mp ? pm : a.Mp - 1)];      // this stage's padding masks (used when the stage is computed)
            brow[s] = xrow_of(p0 + SUB * BP);                // the next stage's rows
        }
    };

    constexpr int NTAP = 5;                                  // accumulator pairs per wave (tap group 1 leaves the last one idle)
    f32x16 acc[NTAP], acc1[NTAP];
#pragma unroll
    for (int t = 0; t < NTAP; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) { acc[t][r] = 0.f; acc1[t][r] = 0.f; }
    const unsigned raw_dy = h2_scale_load(a.dys), raw_x = h2_scale_load(a.xss);   // reduced behind the first tiles' wait
    float bsum = 0.f;
    const bool want_bias = a.dbias_partials != nullptr && tc == 0;
    const int g16 = (lane >> 4) & 1, m16 = lane & 15;
    const int tap0 = tg * 5, ntap = tg ? 4 : 5;
    // per-lane LDS byte addresses of the transposing reads (lane m16 of a 16-lane group supplies bytes 8 (m16 % 4) .. + 7 of pixel
    // m16 / 4 of the window and receives channel m16 of its four pixels).  dy: pixel 8 lh + m16 / 4 (+ 4: immediate), slot
    // (og, plane, g16, m16 % 4 / 2) ^ pixel % 4 << 2 -- the XOR reaches the plane bit, so one address per plane.
    unsigned a_thr[2];
    {
        const int pq = m16 >> 2;
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            const int slot = (og * 8 + pl * 4 + g16 * 2 + ((m16 & 3) >> 1)) ^ (pq << 2);
            a_thr[pl] = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)((8 * lh + pq) * 512 + slot * 16 + (m16 & 1) * 8);
        }
    }
    // x: window start q = 0, 1, 2; pixel slot p = 8 lh + q + m16 / 4 (+ 4: immediate, bit 1 of p unchanged), plane 0 at slot bit 2 =
    // bit 1 of p (plane 1: the address ^ 64)
    unsigned b_thr[3];
#pragma unroll
    for (int q = 0; q < 3; ++q) {
        const int pq = 8 * lh + q + (m16 >> 2);
        b_thr[q] = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)(pq * 128 + ((pq >> 1) & 1) * 64 + g16 * 32 + (m16 & 3) * 8);
    }

    if (nIt > 0) issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const H2Scale sdy = h2_scale_finish(raw_dy), sxx = h2_scale_finish(raw_x);
    typedef unsigned u32x2_ __attribute__((ext_vector_type(2)));
    for (int it = 0; it < nIt; ++it) {
        const int buf = it & 1;
        // padding masks of this stage: bit j of the low half = tap (1, 0) of pixel j exists, bit 16 + j = tap (1, 2) does
        unsigned bal[SUB];
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
            asm volatile("" : "+v"(vmn[s]));
            bal[s] = (unsigned)__builtin_amdgcn_ballot_w64(vmn[s] >= 0);
        }
        if (it + 1 < nIt) issue_stage(it + 1, buf ^ 1);
#pragma unroll
        for (int s = 0; s < SUB; ++s) {
        const unsigned ao = (unsigned)(buf * SUB + s) * (A_EL * 2), bo = (unsigned)(buf * SUB + s) * (B_EL * 2);
#ifdef RADET_W9R_NOMASK
        const bool need = false;
#else
        const bool need = bal[s] != 0xFFFFFFFFu;                 // (uniform) a pixel of this stage sits at an image border
#endif
        // masks over the four VGPRs of a fragment (pixels 8 lh + 2 v, + 1): built from the uniform ballot for both halves of the
        // wave (scalar unit), one select per register
        unsigned mk[2][4];
        if (need) {
            auto ex = [](unsigned b8, int v) { return (((b8 >> (2 * v)) & 1u) ? 0x0000FFFFu : 0u) | (((b8 >> (2 * v + 1)) & 1u) ? 0xFFFF0000u : 0u); };
#pragma unroll
            for (int v = 0; v < 4; ++v) {
                const unsigned l0 = ex(bal[s], v), l1 = ex(bal[s] >> 8, v), r0 = ex(bal[s] >> 16, v), r1 = ex(bal[s] >> 24, v);
                mk[0][v] = lh ? l1 : l0;
                mk[1][v] = lh ? r1 : r0;
            }
        }
        s16x4v_ al[2], ah[2], bl[2][2], bh[2][2];
#pragma unroll
        for (int pl = 0; pl < 2; ++pl) {
            lds_read_tr16<0>(al[pl], a_thr[pl] + ao);
            lds_read_tr16<4 * 512>(ah[pl], a_thr[pl] + ao);
        }
        f16x8 af[2];
        // tap (r, q) = tap0 + t: the window of segment row r that starts q pixels in
        auto read_tap = [&](int t, s16x4v_ (&lo)[2], s16x4v_ (&hi)[2]) {
            const int tap = tap0 + t, q = tap % 3;               // (uniform)
            const unsigned bt = (q == 0 ? b_thr[0] : (q == 1 ? b_thr[1] : b_thr[2])) + bo + (unsigned)((tap / 3) * ROWB);
            lds_read_tr16<0>(lo[0], bt);
            lds_read_tr16<4 * 128>(hi[0], bt);
            lds_read_tr16<0>(lo[1], bt ^ 64u);
            lds_read_tr16<4 * 128>(hi[1], bt ^ 64u);
        };
        read_tap(0, bl[0], bh[0]);
        static_for<0, NTAP>([&](auto tc_) {
            constexpr int t = decltype(tc_)::value, pp = t & 1;
            if (t < ntap) {                                      // uniform per wave
                if (t + 1 < ntap) {
                    if constexpr (t + 1 < NTAP) read_tap(t + 1, bl[pp ^ 1], bh[pp ^ 1]);
                    lds_wait<4>();
                } else {
                    lds_wait<0>();
                }
                if constexpr (t == 0) {
#pragma unroll
                    for (int pl = 0; pl < 2; ++pl) {
                        asm volatile("" : "+v"(al[pl])); asm volatile("" : "+v"(ah[pl]));
                        af[pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(al[pl], ah[pl], 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                }
                const int q = (tap0 + t) % 3;                    // (uniform)
                f16x8 bf[2];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
                    asm volatile("" : "+v"(bl[pp][pl])); asm volatile("" : "+v"(bh[pp][pl]));
                    if (need && q != 1) {                        // (uniform) zero the pixels whose tap is padding
                        const int sd = q == 0 ? 0 : 1;
                        u32x2_ lo = __builtin_bit_cast(u32x2_, bl[pp][pl]), hi = __builtin_bit_cast(u32x2_, bh[pp][pl]);
                        lo.x &= sd ? mk[1][0] : mk[0][0]; lo.y &= sd ? mk[1][1] : mk[0][1];
                        hi.x &= sd ? mk[1][2] : mk[0][2]; hi.y &= sd ? mk[1][3] : mk[0][3];
                        bl[pp][pl] = __builtin_bit_cast(s16x4v_, lo); bh[pp][pl] = __builtin_bit_cast(s16x4v_, hi);
                    }
                    bf[pl] = __builtin_bit_cast(f16x8, __builtin_shufflevector(bl[pp][pl], bh[pp][pl], 0, 1, 2, 3, 4, 5, 6, 7));
                }
                mfma_h2(acc[t], acc1[t], af[0], af[1], bf[0], bf[1]);
            }
            __builtin_amdgcn_sched_barrier(0);
        });
        if (want_bias && tid < BM) {                        // column sums of dy, pixel order, in units of 2^-e (scaled back below)
            const unsigned short* ap = &As[buf * SUB + s][0];
            const int sl = (tid >> 5) * 8 + ((tid >> 4) & 1) * 2 + ((tid >> 3) & 1);      // hi slot of this channel, un-swizzled
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                const int e = p * 256 + ((sl ^ ((p & 3) << 2)) << 3) + (tid & 7);
                bsum += radet_pair_value(ap[e], ap[e ^ 32]);                             // (lo plane: slot ^ 4)
            }
        }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (want_bias && tid < BM && o0 + tid < a.Cout) a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = bsum * sdy.inv;
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
    const int c = c0 + li;
#pragma unroll
    for (int t = 0; t < NTAP; ++t) {
        if (t < ntap) {
            h2_combine(acc[t], acc1[t], sdy.inv, sxx.inv);
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int o = o0 + og * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                if (o < a.Cout) out[((size_t)o * KT + tap0 + t) * a.Cin + c] = acc[t][r];
            }
        }
    }
}

// ------------------------------------------------------------------------------------------ wgrad, one tap, fp16 plane pairs
// conv_wgradh_kernel's data path (the bf16-storage one-tap kernel: [4 pixels][16 channels] sub-tiles by LDS-DMA,
// ds_read_b64_tr_b16 operands, 32 pixels = two K = 16 MFMA steps per stage) for operands that arrive as fp16 plane pairs: per
// (block pair, K step) the three plane products of mfma_h2 into an accumulator pair, no operand work in the loop.  For the
// backbone / neck weight gradients, whose fp32 operands are split into pairs once per tensor (radet_split_pairs on the
// weight-gradient stream) instead of once per use in registers: the in-register one-tap kernel needs 8 ds_read_b32 per
// 8-pixel fragment and 24 VALU operations per fragment, this one 2 transposing reads and none.
template <int BM, int BN, int WM, int WN>
__global__ __launch_bounds__(256) void conv_wgradq_kernel(const WgradArgs a) {
    radet_kernarg_warm<sizeof(WgradArgs)>();
    constexpr int BP = 32, NW = 4;
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int CBA = BM / 16, CBB = BN / 16;             // 16-channel sub-tile columns
    constexpr int A_PL = BP * BM, B_PL = BP * BN;           // fp16 elements per plane tile
    constexpr int A_Q = A_PL * 2 / 1024, B_Q = B_PL * 2 / 1024;     // wave loads per plane tile
    constexpr int A_INSTR = 2 * A_Q, B_INSTR = 2 * B_Q;
    constexpr int N_INSTR = A_INSTR + B_INSTR;
    constexpr int PER_WAVE = (N_INSTR + NW - 1) / NW;
    static_assert(WM * WN == 4, "4 waves");
    __shared__ __attribute__((aligned(16))) unsigned short As[2][2 * A_PL];
    __shared__ __attribute__((aligned(16))) unsigned short Bs[2][2 * B_PL];
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
            const int blk = (bi % B_Q) * 8 + l_blk;
            const int m = p_begin + 4 * (blk / CBB) + l_prow;
            brow[k] = tab_tap ? tab_tap[m < a.Mp ? m : a.Mp - 1] : m;       // unconditional (clamped) load, masked at use
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
                const int pl = bi / B_Q, blk = (bi % B_Q) * 8 + l_blk;
                const int c = c0 + 16 * (blk % CBB) + 8 * l_half;
                radet_lds_load16(xh, (bok[k] && brow[k] >= 0 && c < a.Cin), (size_t)((size_t)brow[k] * 2 * a.Cin + radet_pair_off(c) + 32 * pl), (lptr_t)(&Bs[buf][bi * 512]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int pl = ins / A_Q, blk = (ins % A_Q) * 8 + l_blk;
                const int m = p0 + 4 * (blk / CBA) + l_prow;
                const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
                radet_lds_load16(dyh, (m < p_end && o < a.Cout), (size_t)((size_t)m * 2 * a.ld_dy + radet_pair_off(o) + 32 * pl), (lptr_t)(&As[buf][ins * 512]));
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int blk = ((ins - A_INSTR) % B_Q) * 8 + l_blk;
                const int m = p0 + BP + 4 * (blk / CBB) + l_prow;
                brow[k] = tab_tap ? tab_tap[m < a.Mp ? m : a.Mp - 1] : m;
                bok[k] = m < p_end;
            }
        }
    };

    f32x16 acc[TM][TN], acc1[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) { acc[i][j][t] = 0.f; acc1[i][j][t] = 0.f; }
    const unsigned raw_dy = h2_scale_load(a.dys), raw_x = h2_scale_load(a.xss);   // reduced behind the first tiles' wait
    float bsum = 0.f;
    const bool want_bias = a.dbias_partials != nullptr && tap == 0 && tc == 0;
    const int g16 = (lane >> 4) & 1, m16 = lane & 15;
    // per-lane LDS byte addresses of the transposing reads (sub-tile row 2 * lh of a 4-row group, channel sub-tile of the
    // wave tile + g16, bytes 8 * m16 of the sub-tile), see conv_wgradh_kernel
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(((2 * lh) * CBA + wm * TM * 2 + g16) * 128 + m16 * 8);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)(((2 * lh) * CBB + wn * TN * 2 + g16) * 128 + m16 * 8);
    if (nIt > 0) issue_stage(0, 0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const H2Scale sdy = h2_scale_finish(raw_dy), sxx = h2_scale_finish(raw_x);
    for (int it = 0; it < nIt; ++it) {
        const int buf = it & 1;
        if (it + 1 < nIt) issue_stage(it + 1, buf ^ 1);
        {
            const unsigned ab = a_thr + (unsigned)buf * (2 * A_PL * 2), bb = b_thr + (unsigned)buf * (2 * B_PL * 2);
            s16x4v_ al[2][2][TM], ah[2][2][TM], bl[2][2][TN], bh[2][2][TN];       // [fragment set][plane][block]
            auto read_ks = [&](auto ksc, int pp) {
                constexpr int ks = decltype(ksc)::value;
                static_for<0, 2>([&](auto pc) {
                    constexpr int pl = decltype(pc)::value;
                    static_for<0, TM>([&](auto ic) {
                        constexpr int off = pl * A_PL * 2 + ((4 * ks) * CBA + decltype(ic)::value * 2) * 128;
                        lds_read_tr16<off>(al[pp][pl][decltype(ic)::value], ab);
                        lds_read_tr16<off + CBA * 128>(ah[pp][pl][decltype(ic)::value], ab);
                    });
                    static_for<0, TN>([&](auto jc) {
                        constexpr int off = pl * B_PL * 2 + ((4 * ks) * CBB + decltype(jc)::value * 2) * 128;
                        lds_read_tr16<off>(bl[pp][pl][decltype(jc)::value], bb);
                        lds_read_tr16<off + CBB * 128>(bh[pp][pl][decltype(jc)::value], bb);
                    });
                });
            };
            // (the next K step's reads go out ahead of this one's MFMAs only while they fit lgkmcnt's 4 bits)
            constexpr bool AHEAD = 4 * (TM + TN) <= 15;
            read_ks(std::integral_constant<int, 0>{}, 0);
            static_for<0, BP / 16>([&](auto ksc) {
                constexpr int ks = decltype(ksc)::value, pp = AHEAD ? (ks & 1) : 0;
                if constexpr (AHEAD && ks + 1 < BP / 16) {
                    read_ks(std::integral_constant<int, ks + 1>{}, pp ^ 1);
                    lds_wait<AHEAD ? 4 * (TM + TN) : 0>();
                } else {
                    lds_wait<0>();
                }
                f16x8 af[2][TM], bf[2][TN];
#pragma unroll
                for (int pl = 0; pl < 2; ++pl) {
#pragma unroll
                    for (int i = 0; i < TM; ++i) {
                        asm volatile("" : "+v"(al[pp][pl][i])); asm volatile("" : "+v"(ah[pp][pl][i]));
                        af[pl][i] = __builtin_bit_cast(f16x8, __builtin_shufflevector(al[pp][pl][i], ah[pp][pl][i], 0, 1, 2, 3, 4, 5, 6, 7));
                    }
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        asm volatile("" : "+v"(bl[pp][pl][j])); asm volatile("" : "+v"(bh[pp][pl][j]));
                        bf[pl][j] = __builtin_bit_cast(f16x8, __builtin_shufflevector(bl[pp][pl][j], bh[pp][pl][j], 0, 1, 2, 3, 4, 5, 6, 7));
                    }
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mfma_h2(acc[i][j], acc1[i][j], af[0][i], af[1][i], bf[0][j], bf[1][j]);
                if constexpr (!AHEAD && ks + 1 < BP / 16) read_ks(std::integral_constant<int, ks + 1>{}, 0);
            });
        }
        if (want_bias && tid < BM) {                        // column sums of dy, pixel order, in units of 2^-e (scaled back below)
            const int cb = tid >> 4, cc = tid & 15;
            const unsigned short* ap = &As[buf][0];
#pragma unroll
            for (int p = 0; p < BP; ++p) {
                const int e = ((p >> 2) * CBA + cb) * 64 + (p & 3) * 16 + cc;
                bsum += radet_pair_value(ap[e], ap[A_PL + e]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }
    if (want_bias && tid < BM && o0 + tid < a.Cout) a.dbias_partials[(size_t)split * a.Cout + o0 + tid] = bsum * sdy.inv;
    float* out = a.slabs + (size_t)split * a.Cout * KT * a.Cin;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            h2_combine(acc[i][j], acc1[i][j], sdy.inv, sxx.inv);
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

// the h2 weight-gradient launches behind radet_conv2d_wgrad_s: flags 0x1000 (fp32 tensors, split in registers: the one-tap
// tiles; bits 4-5 / 7 / 10-11 select tile / 32-pixel stages / pixel-divided tiles as for the bf16-plane arithmetic) or
// 0x1000 | 0x200 (fp16 plane pairs: the all-taps kernel above)
int radet_launch_wgrad_h2(const WgradArgs& a, int flags, int bm, int bn, hipStream_t st) {
    if (flags & 0x200) {
        if ((a.ld_dy & 31) || (a.Cin & 31)) return RADET_ERR_ARG;
        if (a.KH == 3 && a.KW == 3 && !(flags & 0x40)) {               // all nine taps per workgroup
            const int tiles9 = ((a.Cout + 127) / 128) * (a.Cin / 32) * a.S;
            if ((flags & 0x2000) && a.chunks_per_split * 16 <= RADET_W9D_TCAP) {      // unit stride, padding 1: deep pipeline, table in LDS
                hipLaunchKernelGGL(conv_wgrad9d_kernel<5>, dim3(tiles9), dim3(512), 0, st, a);
                return radet_check_launch();
            }
            if (flags & 0x4000) {                                        // (experiment) unit stride, padding 1: x as shifted windows
                hipLaunchKernelGGL(conv_wgrad9r_kernel<2>, dim3(tiles9), dim3(512), 0, st, a);
                return radet_check_launch();
            }
            // (experiments, DESIGN.md 7: RADET_WGRAD9_SUB=1 one 16-pixel sub-stage per barrier; _SPREAD=1 the next stage's loads one
            // by one behind the taps' MFMAs; _TGLOAD=1 all loads from the waves of tap group 1)
            static const int sub = getenv("RADET_WGRAD9_SUB") ? atoi(getenv("RADET_WGRAD9_SUB")) : 2;
            static const int spread = getenv("RADET_WGRAD9_SPREAD") ? atoi(getenv("RADET_WGRAD9_SPREAD")) : 0;
            static const int tgload = getenv("RADET_WGRAD9_TGLOAD") ? atoi(getenv("RADET_WGRAD9_TGLOAD")) : 0;
            if (tgload) hipLaunchKernelGGL((conv_wgrad9q_kernel<2, false, true>), dim3(tiles9), dim3(512), 0, st, a);
            else if (spread) hipLaunchKernelGGL((conv_wgrad9q_kernel<2, true>), dim3(tiles9), dim3(512), 0, st, a);
            else if (sub == 1) hipLaunchKernelGGL((conv_wgrad9q_kernel<1, false>), dim3(tiles9), dim3(512), 0, st, a);
            else hipLaunchKernelGGL((conv_wgrad9q_kernel<2, false>), dim3(tiles9), dim3(512), 0, st, a);
            return radet_check_launch();
        }
        // one tap per workgroup (0x40, or not a 3 x 3): bits 4-5 = 1: 128 x 128 tile, otherwise 64 x 64
        const int KTq = a.KH * a.KW;
        if (((flags >> 4) & 3) == 1) {
            const int tiles = ((a.Cout + 127) / 128) * ((a.Cin + 127) / 128) * KTq * a.S;
            hipLaunchKernelGGL((conv_wgradq_kernel<128, 128, 2, 2>), dim3(tiles), dim3(256), 0, st, a);
        } else {
            const int tiles = ((a.Cout + 63) / 64) * ((a.Cin + 63) / 64) * KTq * a.S;
            hipLaunchKernelGGL((conv_wgradq_kernel<64, 64, 2, 2>), dim3(tiles), dim3(256), 0, st, a);
        }
        return radet_check_launch();
    }
    const int KT = a.KH * a.KW;
    // experiment (RADET_WGRAD_LDS_PAD = bytes of unused dynamic LDS per workgroup): fewer weight-gradient workgroups fit a CU,
    // so the dgrad chain's workgroups find LDS there -- see DESIGN.md 7 for what it measured
    static const int pad = getenv("RADET_WGRAD_LDS_PAD") ? atoi(getenv("RADET_WGRAD_LDS_PAD")) : 0;
    if (bm == 64 && (flags & 0xC00)) {
        const int tiles = ((a.Cout + 63) / 64) * ((a.Cin + 63) / 64) * KT * a.S;
        if (flags & 0x400) hipLaunchKernelGGL((conv_wgradg_kernel<64, 64, 2, 2, 3, 64, 4>), dim3(tiles), dim3(256), pad, st, a);
        else hipLaunchKernelGGL((conv_wgradg_kernel<64, 64, 2, 2, 3, 32, 2>), dim3(tiles), dim3(256), pad, st, a);
        return radet_check_launch();
    }
    const int tiles = ((a.Cout + bm - 1) / bm) * ((a.Cin + bn - 1) / bn) * KT * a.S;
#define RADET_WG_H2(BMV, BNV, WMV, WNV) \
    do { if (a.bp32 && BMV >= 64) hipLaunchKernelGGL((conv_wgradg_kernel<BMV, BNV, WMV, WNV, 3, 32>), dim3(tiles), dim3(256), pad, st, a); \
         else hipLaunchKernelGGL((conv_wgradg_kernel<BMV, BNV, WMV, WNV, 3>), dim3(tiles), dim3(256), pad, st, a); } while (0)
    if (bm == 32) RADET_WG_H2(32, 128, 1, 4);
    else if (bm == 64) RADET_WG_H2(64, 64, 2, 2);
    else if (bn == 64) RADET_WG_H2(128, 64, 2, 2);
    else RADET_WG_H2(128, 128, 2, 2);
#undef RADET_WG_H2
    return radet_check_launch();
}
