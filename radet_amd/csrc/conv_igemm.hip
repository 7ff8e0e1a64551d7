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

__device__ __attribute__((aligned(16))) float radet_zero_page[512];

// Implicit-GEMM kernel: the A (gathered pixels) and B (weights) tiles go global -> LDS with
// global_load_lds_dwordx4, no staging registers and no ds_write pass.  A wave load writes 1 KiB lane-linearly, so
// the LDS rows are unpadded [row][BK]; bank conflicts of the 16-byte fragment reads are avoided by an XOR swizzle
// of the 16-byte slot inside a row, applied on the SOURCE side: the lane that fills slot q of tile row r fetches
// k-quad q ^ swz(r), swz(r) = (r / (64 / BK)) % (BK / 4); the reader of k-quad kq looks in slot kq ^ swz(r).
// Stage order per K step: issue the next stage's loads into the other buffer, then fragment reads (software
// pipelined one 8-wide k slice ahead) + MFMAs on the current buffer, then vmcnt(0) + barrier.
// TAG only changes the kernel's symbol name: TAG=1 marks the head-tower GEMM family (M = B*6400, N = 256,
// K = 2304) so that rocprofv3 --stats reports it on its own line (bench.py's roofline kernel).
template <int BM, int BN, int WM, int WN, int TAG, int BK, int NSTG = 2, bool SK = false>
__global__ __launch_bounds__(WM * WN * 64) void conv_igemmg_kernel(const ConvArgs a) {
    constexpr int NW = WM * WN;           // waves per workgroup: 4, or 8 (plane-operand tiles that own a whole CU's LDS)
    constexpr int F4 = BK / 4;            // 16-byte slots per tile row
    constexpr int RPI = 64 / F4;          // tile rows per wave load
    constexpr int RPB = 64 / BK;          // tile rows per 256 bytes of LDS
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    constexpr int A_INSTR = BM / RPI, B_INSTR = BN / RPI;
    constexpr int A_PW = (A_INSTR + NW - 1) / NW, B_PW = (B_INSTR + NW - 1) / NW;
    constexpr int NS = BK / 8;
    constexpr bool BF16 = (TAG & 2) != 0;                     // TAG bit 0: profiling symbol, bit 1: bf16 math mode
    constexpr bool H16 = (TAG & 4) != 0;                      // bit 2: bf16 storage (a 16-byte slot = 8 bf16 = one MFMA operand)
    // bit 3: fp32 tensors, fp32-accurate products on the bf16 matrix cores: every operand is split into three bf16 planes
    // in registers and 6 of the 9 plane products (everything above 2^-24 relative) are accumulated by
    // v_mfma_f32_32x32x16_bf16, which retires 16x the MACs per cycle of v_mfma_f32_32x32x2_f32
    constexpr bool X3 = (TAG & 8) != 0;
    // bit 4: the operands ARRIVE as bf16 plane triples (x rows [3][Cin] bf16 = hi | mid | lo with hi + mid + lo == the fp32
    // value exactly, written once by the producer of the tensor; weights [Cout][taps][3][Cin]): the same 6 plane products as
    // X3, but no operand split anywhere in the K loop -- it is ds_read_b128 + v_mfma only.  Byte geometry per plane = the
    // bf16-storage path (K counted in channel pairs, a 16-byte LDS slot = 8 bf16 = one MFMA operand); outputs stay fp32
    constexpr bool P3 = (TAG & 16) != 0;
    constexpr int NPL = P3 ? 3 : 1;
    // bit 5 (with bit 3): the waves divide the K step as well as the tile.  A stage of BK = 16 KD channels is cut into KD
    // k-groups; wave (kg, nh) accumulates ALL BM rows x its BN / WNK columns over k-group kg, and the KD partial tiles of a
    // column group are added through LDS after the K loop.  A wave's operand splits (VALU work) and fragment reads (LDS
    // bandwidth) then serve TMA x TNA accumulator blocks instead of one: (TMA + TNA) splits per 6 TMA TNA MFMAs -- for the
    // 64 x 64 tile 1 split per 6 MFMAs with KD = 4 (2 x 2 blocks per wave) or 1.5 with KD = 2 (2 x 1), against 2 for the
    // 2 x 2-wave tile whose waves each split one A and one B fragment per 6 MFMAs
    constexpr bool KW = (TAG & 32) != 0;
    constexpr int KD = KW ? BK / 16 : 1;                                  // k-groups per stage
    constexpr int WNK = NW / KD;                                          // column groups of waves
    constexpr int TMA = KW ? BM / 32 : TM, TNA = KW ? BN / (32 * WNK) : TN;      // accumulator blocks of a wave
    static_assert(!KW || (X3 && !SK && NW == 4 && KD * WNK == NW && TM == 1 && TN == 1 && TMA * TNA == KD && NSTG == 2),
                  "K-divided tile: one 32 x 32 block per wave after the reduction");
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(!P3 || BK == 16, "plane rows are laid out in 32-channel groups: one group per K stage");
    // NSTG LDS stages: loads run NSTG - 1 K steps ahead of the MFMAs.  3 stages hide more L2 latency (+8 % on the
    // tower GEMM running alone) but cost LDS occupancy, which loses when dgrad and wgrad kernels share the CUs: used
    // for forward launches only (tile_override 0x20000), chosen per shape by the autotuner
    __shared__ __attribute__((aligned(16))) float As[NSTG][NPL * BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[NSTG][NPL * BN * BK];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int li = lane & 31, lh = lane >> 5;
    const int xld = NPL * a.Cin;          // row stride of x / of one weight tap, in 4-byte units

    const int tilesN = (a.Cout + BN - 1) / BN;
    const int tilesG = ((a.M + BM - 1) / BM) * tilesN;
    const int KT = a.KH * a.KW;
    const int cpt = a.Cin / BK;
    // stream-K: this workgroup walks its share [cur, end_it) of the launch's K stages, tile after tile (virtual index v:
    // the workgroups of one XCD own consecutive tiles); otherwise exactly one (tile, K range) per workgroup
    constexpr bool streamk = SK;                           // stream-K instantiations only: the loop costs registers
    const int vwg = streamk ? ((int)blockIdx.x & 7) * (a.sk_wgs >> 3) + ((int)blockIdx.x >> 3) : 0;
    int cur = streamk ? streamk_start(a, vwg) : 0;
    const int end_it = streamk ? cur + a.sk_base + (vwg < a.sk_rem ? 1 : 0) : 1;
    const int first_tile = cur / (KT * cpt);
  do {
    const bool tail = !streamk && (int)blockIdx.x >= a.n_full;
    const int tail_slot = tail ? (int)blockIdx.x - a.n_full : 0;
    // (class launches keep the grid order: classes are sorted by taps, heaviest first, and spread over the XCDs)
    int id = streamk ? cur / (KT * cpt)
                     : (tail ? a.n_full + tail_slot / a.sk_tail : (a.cls_nt ? (int)blockIdx.x : xcd_remap(blockIdx.x, a.n_full)));
    const int sk_tile = id;
    const int grp = id >= tilesG ? 1 : 0;
    id -= grp * tilesG;
    ConvPtrs P = a.p[grp];
    P.y = pin_sgpr(P.y); P.bias = pin_sgpr(P.bias); P.addend = pin_sgpr(P.addend); P.mask = pin_sgpr(P.mask);
    EpiArgs epi;
    epi.M = pin_sgpr(a.M); epi.Cout = pin_sgpr(a.Cout); epi.relu = pin_sgpr(a.relu); epi.io = pin_sgpr(a.io);
    epi.out_rows = pin_sgpr(a.out_rows);
    epi.partial = pin_sgpr(a.partial); epi.counters = pin_sgpr(a.counters);
    epi.sk_base = pin_sgpr(a.sk_base); epi.sk_rem = pin_sgpr(a.sk_rem);
    // split episode of this workgroup: (number of splits, split-tile index, my split)
    const int nsplit = pin_sgpr(tail ? a.sk_tail : a.sk);
    const int ctile = pin_sgpr(tail ? tail_slot / a.sk_tail : sk_tile);
    const int zsplit = pin_sgpr(tail ? tail_slot % a.sk_tail : (int)blockIdx.y);
    const int m0 = (id / tilesN) * BM;
    const int n0 = (id % tilesN) * BN;
    int KTt = KT, tbase = 0;                                 // taps of this tile, its slice of tap_ids
    if (!streamk && a.cls_nt) {
        const int cls = (m0 >= a.cls_b[0] ? 1 : 0) + (m0 >= a.cls_b[1] ? 1 : 0) + (m0 >= a.cls_b[2] ? 1 : 0);
        KTt = (a.cls_nt >> (4 * cls)) & 15;
        tbase = 4 * cls;
    }

    const int per = tail ? a.it_per_tail : a.it_per_split;
    const int it0 = streamk ? cur - sk_tile * KT * cpt : (tail ? tail_slot % a.sk_tail : (int)blockIdx.y) * per;
    int nK = KTt * cpt - it0;
    if (streamk) {
        if (nK > end_it - cur) nK = end_it - cur;
    } else if (nK > per) {
        nK = per;
    }

    // writer side: this lane fills slot (lane % F4) of tile row ins * RPI + lane / F4 of every load it issues
    const int lrow = lane / F4;
    // K order: tap-major (all channel chunks of a tap, then the next tap), or -- plane operands (dbg bit 3: tap-major) --
    // channel-major (the taps of one channel chunk back to back: the shifted re-reads of an input row are then a few stages
    // apart instead of a whole channel sweep, i.e. they hit the XCD's L2 instead of the Infinity Cache)
    const bool cmaj = P3 && (!RADET_P3_DBG || !(a.dbg & 8));
    int ld_tap = cmaj ? it0 % KTt : it0 / cpt, ld_c0 = cmaj ? (it0 / KTt) * BK : (it0 - ld_tap * cpt) * BK;
    int arow[A_PW], akq[A_PW];
    const float* wp[B_PW];
#pragma unroll
    for (int k = 0; k < A_PW; ++k) {
        const int r = (wave + NW * k) * RPI + lrow;
        akq[k] = 4 * ((lane % F4) ^ ((r / RPB) % F4));
        arow[k] = (nK > 0 && wave + NW * k < A_INSTR) ? a.rowtab[(size_t)ld_tap * a.Mp + m0 + r] : -1;
    }
#pragma unroll
    for (int k = 0; k < B_PW; ++k) {
        const int r = (wave + NW * k) * RPI + lrow;
        const int n = n0 + r;
        wp[k] = (wave + NW * k < B_INSTR && n < a.Cout)
                    ? P.w + (size_t)n * a.KTw * xld + 4 * ((lane % F4) ^ ((r / RPB) % F4)) : nullptr;
    }
    constexpr bool A_FULL = A_INSTR % NW == 0, B_FULL = B_INSTR % NW == 0;   // every wave owns A_PW / B_PW loads
    int wtap = nK > 0 ? a.tap_ids[tbase + ld_tap] : 0;
    // the loads of one K stage as individually issuable pieces (piece q < NPIECE: plane p of this wave's k-th A load, then
    // of its k-th B load), so that the plane-operand loop can spread them between its MFMAs; advance_stage() moves the
    // (tap, channel chunk) cursor and fetches the next gather rows
    constexpr int NPIECE = NPL * (A_PW + B_PW);
    // plane operands: per load k a base pointer (row start, or the zero page for padding rows) and a mask that cancels the
    // stage offset on padding rows -- a select between two LOADS per piece costs exec-mask juggling and a branch each
    const float* abase[A_PW];
    const float* wbase[B_PW];
    unsigned amask[A_PW], wmask[B_PW];
    constexpr bool BMASK = P3 || KW;
    int pc0 = 0, pwt = 0;                 // (channel chunk, weight tap) of the stage whose pieces are being issued
    auto set_abase = [&]() {
        pc0 = ld_c0; pwt = wtap;
        if constexpr (BMASK) {
#pragma unroll
            for (int k = 0; k < A_PW; ++k) {
                amask[k] = arow[k] >= 0 ? 0xFFFFFFFFu : 0u;
                const unsigned long long real = (unsigned long long)(P.x + (size_t)(arow[k] & (int)amask[k]) * xld + akq[k]);
                const unsigned long long zero = (unsigned long long)(radet_zero_page + lane * 4);
                const unsigned long long m = (unsigned long long)(long long)(int)amask[k];
                abase[k] = (const float*)((real & m) | (zero & ~m));
            }
        }
    };
    if constexpr (BMASK) {
#pragma unroll
        for (int k = 0; k < B_PW; ++k) {
            wmask[k] = wp[k] ? 0xFFFFFFFFu : 0u;
            wbase[k] = wp[k] ? wp[k] : radet_zero_page + lane * 4;
        }
    }
    set_abase();
    auto issue_piece = [&](int buf, auto qc) {
        constexpr int q = decltype(qc)::value;
        if constexpr (q < NPL * A_PW) {
            constexpr int k = q / NPL, p = q % NPL;
            const int ins = wave + NW * k;
            if (A_FULL || ins < A_INSTR) {
                // plane rows: 32-channel groups of [hi | mid | lo] x 16 units -> chunk ld_c0 starts at unit 3 * ld_c0
                const float* src;
                if constexpr (BMASK) src = abase[k] + ((unsigned)(NPL * pc0 + p * BK) & amask[k]);
                else src = arow[k] >= 0 ? P.x + (size_t)arow[k] * xld + NPL * pc0 + akq[k] + p * BK : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&As[buf][p * BM * BK + ins * 256]), 16, 0, 0);
            }
        } else {
            constexpr int k = (q - NPL * A_PW) / NPL, p = (q - NPL * A_PW) % NPL;
            const int ins = wave + NW * k;
            if (B_FULL || ins < B_INSTR) {
                const float* src;
                if constexpr (BMASK) src = wbase[k] + ((unsigned)(pwt * xld + NPL * pc0 + p * BK) & wmask[k]);
                else src = wp[k] ? wp[k] + pwt * xld + NPL * pc0 + p * BK : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&Bs[buf][p * BN * BK + ins * 256]), 16, 0, 0);
            }
        }
    };
    auto advance_stage = [&]() {
        if (cmaj) {
            if (++ld_tap == KTt) { ld_tap = 0; ld_c0 += BK; }
            wtap = a.tap_ids[tbase + ld_tap];
#pragma unroll
            for (int k = 0; k < A_PW; ++k)
                if (A_FULL || wave + NW * k < A_INSTR)
                    arow[k] = a.rowtab[(size_t)ld_tap * a.Mp + m0 + (wave + NW * k) * RPI + lrow];
            return;                     // (abase follows in refresh_abase(), right before the next stage's first piece)
        }
        ld_c0 += BK;
        if (ld_c0 == a.Cin) {
            ld_c0 = 0;
            ++ld_tap;
            if (ld_tap < KTt) {
                wtap = a.tap_ids[tbase + ld_tap];
#pragma unroll
                for (int k = 0; k < A_PW; ++k)
                    if (A_FULL || wave + NW * k < A_INSTR)
                        arow[k] = a.rowtab[(size_t)ld_tap * a.Mp + m0 + (wave + NW * k) * RPI + lrow];
            }
        }
    };
    auto issue_stage = [&](int buf) {
        set_abase();
        static_for<0, NPIECE>([&](auto qc) { issue_piece(buf, qc); });
        advance_stage();
    };

    f32x16 acc[TMA][TNA];
#pragma unroll
    for (int i = 0; i < TMA; ++i)
#pragma unroll
        for (int j = 0; j < TNA; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    // vmcnt(LOADS) = "everything except the newest stage's loads has landed" (in-order return); only valid when every
    // wave owns exactly A_PW + B_PW loads per stage
    constexpr int LOADS = (A_FULL && B_FULL) ? NPL * (A_PW + B_PW) : 0;
    // prologue: stages 0 .. NSTG-2 in flight, stage 0 landed
    if (nK > 0) issue_stage(0);
    if (NSTG >= 3 && nK >= NSTG - 1) {
        issue_stage(1);
        if constexpr (NSTG >= 4) issue_stage(2);
        asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (NSTG - 2)) : "memory");
    } else {
        if (NSTG >= 3 && nK > 1) issue_stage(1);          // short K range: fewer stages, plain wait
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    }
    __syncthreads();

    // reader side: tile rows wm*TM*32 + i*32 + li; (i*32) % (RPB*F4) == 0 so swz only depends on li.
    // The fragment reads are inline asm: the compiler would otherwise order every ds_read behind a vmcnt(0) wait on
    // the in-flight LDS-DMA loads (it cannot prove they target the other buffer) and serialise load and compute.
    const int rswz = (li / RPB) % F4;
    unsigned aaddr[NS], baddr[NS];
    {
        const unsigned a_lds = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)((wm * TM * 32 + li) * BK * 4);
        const unsigned b_lds = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)((wn * TN * 32 + li) * BK * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            aaddr[s] = a_lds + 16u * (unsigned)((2 * s + lh) ^ rswz);
            baddr[s] = b_lds + 16u * (unsigned)((2 * s + lh) ^ rswz);
        }
    }
    auto stage = [&](auto bufc, int it) {
        constexpr int BUF = decltype(bufc)::value;
        constexpr int AO = BUF * BM * BK * 4, BO = BUF * BN * BK * 4, RO = 32 * BK * 4;
        if (it + NSTG - 1 < nK) issue_stage((BUF + NSTG - 1) % NSTG);
        f32x4 af[X3 ? NS : 2][TM], bf[X3 ? NS : 2][TN];
        auto read_s = [&](int s, int pp) {
            lds_read128<AO>(af[pp][0], aaddr[s]);
            if constexpr (TM > 1) lds_read128<AO + RO>(af[pp][TM - 1], aaddr[s]);
            lds_read128<BO>(bf[pp][0], baddr[s]);
            if constexpr (TN > 1) lds_read128<BO + RO>(bf[pp][TN - 1], baddr[s]);
        };
        if constexpr (X3) {
            // all fragment reads of the stage up front; slices 2g, 2g + 1 are the 8 k values per lane of one K = 16 MFMA
            // (the lane -> k assignment only has to be the same for A and B)
#pragma unroll
            for (int s = 0; s < NS; ++s) read_s(s, s);
            static_for<0, NS / 2>([&](auto gc) {
                constexpr int g = decltype(gc)::value;
                lds_wait<(NS - 2 * g - 2) * (TM + TN)>();
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[2 * g][i]), "+v"(af[2 * g + 1][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bf[2 * g][j]), "+v"(bf[2 * g + 1][j]));
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) split3_bf16(af[2 * g][i], af[2 * g + 1][i], ah[i], am[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) split3_bf16(bf[2 * g][j], bf[2 * g + 1][j], bh[j], bm[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
                // nothing may move across this point: without it the compiler hoists the NEXT waits (incl. the stage's
                // closing vmcnt(0)) above this group's split + MFMAs and the wave waits for its own prefetch first
                __builtin_amdgcn_sched_barrier(0);
            });
        } else {
        read_s(0, 0);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            const int pp = s & 1;
            if (s + 1 < NS) {
                read_s(s + 1, pp ^ 1);
                lds_wait<TM + TN>();
            } else {
                lds_wait<0>();
            }
#pragma unroll
            for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(af[pp][i]));
#pragma unroll
            for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(bf[pp][j]));
            if constexpr (H16) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, af[pp][i]),
                                                                            __builtin_bit_cast(bf16x8, bf[pp][j]), acc[i][j], 0, 0, 0);
            } else if constexpr (BF16) {
                s16x4 ab[TM], bb[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) ab[i] = cvt_bf16x4(af[pp][i].x, af[pp][i].y, af[pp][i].z, af[pp][i].w);
#pragma unroll
                for (int j = 0; j < TN; ++j) bb[j] = cvt_bf16x4(bf[pp][j].x, bf[pp][j].y, bf[pp][j].z, bf[pp][j].w);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x8bf16_1k(ab[i], bb[j], acc[i][j], 0, 0, 0);
            } else {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pp][i].x, bf[pp][j].x, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pp][i].y, bf[pp][j].y, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pp][i].z, bf[pp][j].z, acc[i][j], 0, 0, 0);
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x2f32(af[pp][i].w, bf[pp][j].w, acc[i][j], 0, 0, 0);
                    }
            }
            __builtin_amdgcn_sched_barrier(0);                // keep the MFMAs of slice s ahead of the next waits
        }
        }
        // stage it+1 has landed once at most the NSTG-2 stages issued after it are outstanding (in-order return);
        // on the last stages of the range fewer are in flight: plain wait
        if (NSTG >= 3 && it + NSTG - 1 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (NSTG - 2)) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    };
    if constexpr (KW) {
        const int kg = wave % KD, nh = wave / KD;
        const unsigned ka = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(li * BK * 4);
        const unsigned kb = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)((nh * TNA * 32 + li) * BK * 4);
        unsigned kaa[2], kba[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            kaa[h] = ka + 16u * (unsigned)((4 * kg + 2 * h + lh) ^ rswz);
            kba[h] = kb + 16u * (unsigned)((4 * kg + 2 * h + lh) ^ rswz);
        }
        f32x4 fa[2][2][TMA], fb[2][2][TNA];              // [fragment set][k half][block]
        constexpr int NRD = 2 * (TMA + TNA);
        // fragment read r of buffer BUF into fragment set PP: A blocks, then B blocks, k half 0 then 1
        auto read_one = [&](auto bufc, auto ppc, auto rc) {
            constexpr int BUF = decltype(bufc)::value, PP = decltype(ppc)::value, r = decltype(rc)::value;
            constexpr int AO = BUF * BM * BK * 4, BO = BUF * BN * BK * 4, RO = 32 * BK * 4;
            constexpr int h = r / (TMA + TNA), e = r % (TMA + TNA);
            if constexpr (e < TMA) lds_read128<AO + e * RO>(fa[PP][h][e], kaa[h]);
            else lds_read128<BO + (e - TMA) * RO>(fb[PP][h][e - TMA], kba[h]);
        };
        auto pin = [&](auto ppc) {
            constexpr int PP = decltype(ppc)::value;
            (void)fa; (void)fb;            // (named outside the asm operands: clang does not capture through those alone)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
#pragma unroll
                for (int i = 0; i < TMA; ++i) asm volatile("" : "+v"(fa[PP][h][i]));
#pragma unroll
                for (int j = 0; j < TNA; ++j) asm volatile("" : "+v"(fb[PP][h][j]));
            }
        };
        // two stages, several workgroups per CU: loads of stage it + 1 at the head of stage it, the fragment reads of
        // stage it + 1 right behind the barrier that publishes it
        auto stage_kw = [&](auto bufc, auto ppc, int it) {
            constexpr int BUF = decltype(bufc)::value, PP = decltype(ppc)::value;
            if (it + 1 < nK) issue_stage(BUF ^ 1);
            lds_wait<0>();
            pin(ppc);
            bf16x8 ah[TMA], am[TMA], al[TMA], bh[TNA], bm[TNA], bl[TNA];
#pragma unroll
            for (int i = 0; i < TMA; ++i) split3_bf16(fa[PP][0][i], fa[PP][1][i], ah[i], am[i], al[i]);
#pragma unroll
            for (int j = 0; j < TNA; ++j) split3_bf16(fb[PP][0][j], fb[PP][1][j], bh[j], bm[j], bl[j]);
#pragma unroll
            for (int i = 0; i < TMA; ++i)
#pragma unroll
                for (int j = 0; j < TNA; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
            __builtin_amdgcn_sched_barrier(0);
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __syncthreads();
        };
        if (nK > 0) static_for<0, NRD>([&](auto rc) { read_one(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rc); });
        for (int it = 0; it < nK; it += 2) {
            stage_kw(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, it);
            if (it + 1 < nK) {
                static_for<0, NRD>([&](auto rc) { read_one(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, rc); });
                stage_kw(std::integral_constant<int, 1>{}, std::integral_constant<int, 1>{}, it + 1);
                if (it + 2 < nK) static_for<0, NRD>([&](auto rc) { read_one(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rc); });
            }
        }
    } else if constexpr (P3) {
        // Plane operands: per K = 16 slice 3 (TM + TN) fragment reads (one ds_read_b128 = the 8 bf16 of one plane a lane
        // feeds to v_mfma_f32_32x32x16_bf16) and 6 TM TN MFMAs, nothing else.  Everything that is not an MFMA is spread
        // BETWEEN the MFMAs: a slice is six groups of TM TN MFMAs (one plane product each), and behind each group go a few
        // of the fragment reads of the next slice and -- in the last slice of a stage -- of the tile loads that refill the
        // buffer released by the stage's barrier.  These tiles own the CU's LDS (one workgroup per CU, two waves per SIMD in
        // lockstep): issued in a block after the barrier, the loads of both waves idle the SIMD's matrix pipe together.
        // The pipeline is rotated by one slice: the barrier that ends stage `it` sits in front of the MFMAs of its last
        // slice, whose fragments are in registers already.
        f32x4 fa[2][3][TM], fb[2][3][TN];
        constexpr int NRD = 3 * (TM + TN);
        // fragment read r of slice s of buffer BUF into fragment set pp: order A hi, B hi, A mid, B mid, A lo, B lo
        auto read_one = [&](auto bufc, auto sc, auto ppc, auto rc) {
            constexpr int BUF = decltype(bufc)::value, s = decltype(sc)::value, pp = decltype(ppc)::value, r = decltype(rc)::value;
            constexpr int AO = BUF * NPL * BM * BK * 4, BO = BUF * NPL * BN * BK * 4, RO = 32 * BK * 4;
            constexpr int pl = r / (TM + TN), e = r % (TM + TN);
            // (the buffer offset goes into the address register: a ds_read immediate holds 16 bits)
            if constexpr (e < TM) lds_read128<pl * BM * BK * 4 + e * RO>(fa[pp][pl][e], aaddr[s] + (unsigned)AO);
            else lds_read128<pl * BN * BK * 4 + (e - TM) * RO>(fb[pp][pl][e - TM], baddr[s] + (unsigned)BO);
        };
        // the six MFMA groups of fragment set PP; behind group g: reads [g NRD / 5, (g + 1) NRD / 5) of the next fragment set
        // (none behind the last group: they would not be back by the next slice) and, with LD, the pieces of the refill
        auto slice = [&](auto ppc, auto rbufc, auto rsc, bool do_read, auto ldc, int ld_buf, bool do_load) {
            constexpr int PP = decltype(ppc)::value;
            constexpr bool LD = decltype(ldc)::value;
#pragma unroll
            for (int pl = 0; pl < 3; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[PP][pl][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[PP][pl][j]));
            }
            static_for<0, 6>([&](auto tc) {
                constexpr int t = decltype(tc)::value;          // terms: hi hi, mid hi, hi mid, mid mid, lo hi, hi lo
                constexpr int pa = t == 1 || t == 3 ? 1 : (t == 4 ? 2 : 0), pb = t == 2 || t == 3 ? 1 : (t == 5 ? 2 : 0);
                if (!RADET_P3_DBG || !(a.dbg & 2)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PP][pa][i]),
                                                                            __builtin_bit_cast(bf16x8, fb[PP][pb][j]), acc[i][j], 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (LD) {
                    if (do_load) {
                        if constexpr (t == 0) set_abase();
                        static_for<t * NPIECE / 6, (t + 1) * NPIECE / 6>([&](auto qc) { issue_piece(ld_buf, qc); });
                        if constexpr (t == 5) advance_stage();
                    }
                }
                if constexpr (t < 5) {
                    if (do_read)
                        static_for<t * NRD / 5, (t + 1) * NRD / 5>([&](auto rc) { read_one(rbufc, rsc, std::integral_constant<int, PP ^ 1>{}, rc); });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto stage_p3 = [&](auto bufc, int it) {
            constexpr int BUF = decltype(bufc)::value;
            static_for<0, NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int PP = s & 1;                                      // parity of the global slice index (NS even)
                lds_wait<0>();                                                 // fragment set PP has arrived
                if constexpr (s + 1 < NS) {
                    slice(std::integral_constant<int, PP>{}, bufc, std::integral_constant<int, s + 1>{},
                          !RADET_P3_DBG || !(a.dbg & 4), std::false_type{}, 0, false);
                } else {
                    // (this wave has no read of buffer BUF in flight any more)
                    if (NSTG >= 3 && it + NSTG - 1 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (NSTG - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    __syncthreads();                                           // stage it + 1 landed, buffer BUF released
                    slice(std::integral_constant<int, PP>{}, std::integral_constant<int, (BUF + 1) % NSTG>{},
                          std::integral_constant<int, 0>{}, it + 1 < nK && (!RADET_P3_DBG || !(a.dbg & 4)), std::true_type{}, BUF,
                          it + NSTG < nK && (!RADET_P3_DBG || !(a.dbg & 1)));
                }
            });
        };
        static_assert(NS % 2 == 0, "fragment double buffer: compile-time slice parity");
        if (nK > 0) {
            if (NSTG - 1 < nK) issue_stage(NSTG - 1);
            static_for<0, NRD>([&](auto rc) {
                read_one(std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, std::integral_constant<int, 0>{}, rc);
            });
        }
        for (int it = 0; it < nK; it += NSTG) {
            stage_p3(std::integral_constant<int, 0>{}, it);
            if (it + 1 < nK) stage_p3(std::integral_constant<int, 1>{}, it + 1);
            if constexpr (NSTG >= 3)
                if (it + 2 < nK) stage_p3(std::integral_constant<int, 2>{}, it + 2);
            if constexpr (NSTG >= 4)
                if (it + 3 < nK) stage_p3(std::integral_constant<int, 3>{}, it + 3);
        }
    } else
    for (int it = 0; it < nK; it += NSTG) {
        stage(std::integral_constant<int, 0>{}, it);
        if (it + 1 < nK) stage(std::integral_constant<int, 1>{}, it + 1);
        if constexpr (NSTG >= 3)
            if (it + 2 < nK) stage(std::integral_constant<int, 2>{}, it + 2);
        if constexpr (NSTG >= 4)
            if (it + 3 < nK) stage(std::integral_constant<int, 3>{}, it + 3);
    }
    if constexpr (SK) {
        cur += nK;
        bool fin = nK == KT * cpt;                                // whole tile: plain epilogue
        if (!fin)
            fin = streamk_publish<BM, BN, WM, WN>(epi, acc, sk_tile, nK, KT * cpt, vwg, sk_tile == first_tile ? 0 : 1,
                                                  reinterpret_cast<volatile int*>(&As[0][0]));
        if (fin) igemm_store<BM, BN, WM, WN>(epi, P, acc, m0, n0, 1, 0, 0, wm, wn, li, lh, nullptr);
    } else if constexpr (KW) {
        // the KD partial tiles of a column group -> one: wave (kg, nh) keeps its block kg (row block kg / TNA, column block
        // kg % TNA of the group) and ships its other KD - 1 blocks through LDS (4 KiB each, slot (sender wave, block) in the
        // stage buffers, which every wave has left behind the closing barrier of the last stage); sum order fixed: own block +
        // the partners in ascending k-group order
        constexpr int SLOTS_A = NSTG * BM * BK / 1024, SLOTS_B = NSTG * BN * BK / 1024;
        static_assert(NW * (KD - 1) <= SLOTS_A + SLOTS_B, "partial blocks fit the stage buffers");
        const int kg = wave % KD, nh = wave / KD;
        float* const pa = &As[0][0];
        float* const pb = &Bs[0][0];
        auto slot_ptr = [&](int sender_wave, int blk) {
            const int sk_ = sender_wave % KD;
            const int slot = sender_wave * (KD - 1) + blk - (blk > sk_ ? 1 : 0);
            return reinterpret_cast<f32x4*>(slot < SLOTS_A ? pa + slot * 1024 : pb + (slot - SLOTS_A) * 1024) + lane;
        };
        f32x16 out[1][1];
        static_for<0, KD>([&](auto bc) {
            constexpr int b = decltype(bc)::value;
            const f32x16& v = acc[b / TNA][b % TNA];
            if (kg == b) {
                out[0][0] = v;
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
                    out[0][0][4 * q] += t.x; out[0][0][4 * q + 1] += t.y; out[0][0][4 * q + 2] += t.z; out[0][0][4 * q + 3] += t.w;
                }
            }
        }
        __syncthreads();                                            // (igemm_store reuses the head of As for its ticket)
        igemm_store<BM, BN, WM, WN>(epi, P, out, m0, n0, nsplit, ctile, zsplit, kg / TNA, nh * TNA + kg % TNA, li, lh,
                                    reinterpret_cast<volatile int*>(&As[0][0]));
        cur = end_it;
    } else {
        igemm_store<BM, BN, WM, WN>(epi, P, acc, m0, n0, nsplit, ctile, zsplit, wm, wn, li, lh,
                                    reinterpret_cast<volatile int*>(&As[0][0]));
        cur = end_it;
    }
  } while (SK && cur < end_it);
}

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
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const float* src = (bok[k] && brow[k] >= 0) ? a.x + (size_t)brow[k] * a.Cin + c0 + (lane & 7) * 4
                                                            : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&Bs[buf][bi * 256]), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int j = ins * ROWS_A + (ROWS_A == 1 ? 0 : (lane * 4) / BM);
                const int o = o0 + (lane * 4) % BM;
                const int m = p0 + j;
                const float* src = (m < p_end && o < a.Cout) ? a.dy + (size_t)m * a.ld_dy + o : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&As[buf][ins * 256]), 16, 0, 0);
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
    static_assert(!KW || (MATH == 2 && BP == 16 * KD && TM == 1 && TN == 1 && TMA * TNA == KD), "pixel-divided tile");
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
    const int* tab_tap = a.rowtab + (size_t)tap * a.Mp;

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
            brow[k] = tab_tap[m < a.Mp ? m : a.Mp - 1];         // unconditional (clamped) load; rows >= p_end are masked at use
            bok[k] = m < p_end;
        }
    }
    // Issue order inside a stage: (1) the x-tile loads, which consume the gather rows fetched one stage earlier, (2) the
    // dy-tile loads, (3) the gather rows of the next stage.  The compiler cannot see that the rows loaded in the previous
    // iteration were already drained by the barrier's vmcnt(0) and waits (vmcnt(0)) before their first use: placed first,
    // that wait is free; placed after a dy-tile load (the former order) it stalled every stage on its own prefetch.
    auto issue_stage = [&](int it, int buf) {
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int c = c0 + (lane * 4) % BN;
                const float* src = (bok[k] && brow[k] >= 0 && c < a.Cin) ? a.x + (size_t)brow[k] * a.Cin + c
                                                                         : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&Bs[buf][bi * 256]), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int m = p0 + ins * RA + (lane * 4) / BM;
                const int o = o0 + (lane * 4) % BM;
                const float* src = (m < p_end && o < a.Cout) ? a.dy + (size_t)m * a.ld_dy + o : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&As[buf][ins * 256]), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                // unconditional (clamped) load; rows >= p_end are masked at use
                const int m = p0 + BP + (ins - A_INSTR) * RB + (lane * 4) / BN;
                brow[k] = tab_tap[m < a.Mp ? m : a.Mp - 1];
                bok[k] = m < p_end;
            }
        }
    };

    f32x16 acc[TMA][TNA];
#pragma unroll
    for (int i = 0; i < TMA; ++i)
#pragma unroll
        for (int j = 0; j < TNA; ++j)
#pragma unroll
            for (int t = 0; t < 16; ++t) acc[i][j][t] = 0.f;
    float bsum = 0.f;
    const int kg = wave % KD, nh = wave / KD;
    const bool want_bias = a.dbias_partials != nullptr && tap == 0 && tc == 0;

    // per-lane LDS byte addresses of the operand reads: pixel row lh of a k pair, channel (wave tile) * 32 + li
    const unsigned a_thr = (unsigned)(size_t)(lptr_t)(&As[0][0]) + 4u * (unsigned)(lh * BM + wm * TM * 32 + li);
    const unsigned b_thr = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + 4u * (unsigned)(lh * BN + wn * TN * 32 + li);
    if (nIt > 0) issue_stage(0, 0);
    __syncthreads();
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
            bf16x8 ah[TMA], am[TMA], al[TMA], bh[TNA], bm[TNA], bl[TNA];
#pragma unroll
            for (int i = 0; i < TMA; ++i) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(a8[i][e]));
                split3_bf16(a8[i], ah[i], am[i], al[i]);
            }
#pragma unroll
            for (int j = 0; j < TNA; ++j) {
#pragma unroll
                for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(b8[j][e]));
                split3_bf16(b8[j], bh[j], bm[j], bl[j]);
            }
#pragma unroll
            for (int i = 0; i < TMA; ++i)
#pragma unroll
                for (int j = 0; j < TNA; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
            __builtin_amdgcn_sched_barrier(0);
        } else if constexpr (MATH == 2) {
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
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(a8[i][e]));
                    split3_bf16(a8[i], ah[i], am[i], al[i]);
                }
#pragma unroll
                for (int j = 0; j < TN; ++j) {
#pragma unroll
                    for (int e = 0; e < 8; ++e) asm volatile("" : "+v"(b8[j][e]));
                    split3_bf16(b8[j], bh[j], bm[j], bl[j]);
                }
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
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
    const int* tab_tap = a.rowtab + (size_t)tap * a.Mp;

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
            brow[k] = tab_tap[m < a.Mp ? m : a.Mp - 1];         // unconditional (clamped) load, masked at use
            bok[k] = m < p_end;
        }
    }
    auto issue_stage = [&](int it, int buf) {                 // order: x tiles, dy tiles, next gather rows (see conv_wgradg)
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int blk = bi * 8 + l_blk;
                const int c = c0 + 16 * (blk % CBB) + 8 * l_half;
                const void* src = (bok[k] && brow[k] >= 0 && c < a.Cin) ? (const void*)(xh + (size_t)brow[k] * a.Cin + c)
                                                                        : (const void*)(radet_zero_page + lane * 4);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&Bs[buf][bi * 512]), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int blk = ins * 8 + l_blk;
                const int m = p0 + 4 * (blk / CBA) + l_prow;
                const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
                const void* src = (m < p_end && o < a.Cout) ? (const void*)(dyh + (size_t)m * a.ld_dy + o)
                                                            : (const void*)(radet_zero_page + lane * 4);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&As[buf][ins * 512]), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int blk = (ins - A_INSTR) * 8 + l_blk;
                const int m = p0 + BP + 4 * (blk / CBB) + l_prow;
                brow[k] = tab_tap[m < a.Mp ? m : a.Mp - 1];
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
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int blk = (bi % B_TAP) * 8 + l_blk;
                const int c = c0 + 16 * (blk % CBB) + 8 * l_half;
                const void* src = (bok[k] && brow[k] >= 0) ? (const void*)(xh + (size_t)brow[k] * a.Cin + c)
                                                           : (const void*)(radet_zero_page + lane * 4);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&Bs[buf][bi * 512]), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int blk = ins * 8 + l_blk;
                const int m = p0 + 4 * (blk / CBA) + l_prow;
                const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
                const void* src = (m < p_end && o < a.Cout) ? (const void*)(dyh + (size_t)m * a.ld_dy + o)
                                                            : (const void*)(radet_zero_page + lane * 4);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&As[buf][ins * 512]), 16, 0, 0);
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
        const int p0 = p_begin + it * BP;
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins >= A_INSTR && ins < N_INSTR) {
                const int bi = ins - A_INSTR;
                const int c = c0 + 16 * (l_blk % CBB) + 8 * l_half;
                const void* src = (bok[k] && brow[k] >= 0)
                                      ? (const void*)(xh + (size_t)brow[k] * 3 * a.Cin + radet_plane_off(c) + 32 * (bi % 3))
                                      : (const void*)(radet_zero_page + lane * 4);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&Bs[buf][bi * B_PL]), 16, 0, 0);
            }
        }
#pragma unroll
        for (int k = 0; k < PER_WAVE; ++k) {
            const int ins = wave + k * NW;
            if (ins < A_INSTR) {
                const int pl = ins / A_Q, blk = (ins % A_Q) * 8 + l_blk;
                const int m = p0 + 4 * (blk / CBA) + l_prow;
                const int o = o0 + 16 * (blk % CBA) + 8 * l_half;
                const void* src = (m < p_end && o < a.Cout)
                                      ? (const void*)(dyh + (size_t)m * 3 * a.ld_dy + radet_plane_off(o) + 32 * pl)
                                      : (const void*)(radet_zero_page + lane * 4);
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&As[buf][ins * 512]), 16, 0, 0);
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
    const int T = a.groups * ((a.M + BM - 1) / BM) * ((a.Cout + BN - 1) / BN);
    a.sk_wgs = a.sk_base = a.sk_rem = 0;
    if (skw > 0 && a.partial != nullptr && tag == 0 && BM * BN <= 128 * 64) {
        // stream-K with skw workgroups per CU: needs at least one K stage per workgroup, a ticket per tile and two
        // partial-tile slots per workgroup; otherwise the plain launch below
        const long I = (long)T * a.KH * a.KW * (a.Cin / bk);
        const int G = 256 * skw;
        if (I >= G && T <= RADET_SPLIT_COUNTERS && (size_t)G * 2 * BM * BN <= ws_floats && T % G != 0) {
            a.sk = 1;
            a.it_per_split = a.KH * a.KW * (a.Cin / bk);
            a.sk_wgs = G;
            a.sk_base = (int)(I / G);
            a.sk_rem = (int)(I % G);
        }
    }
    // split-K partial tiles are tile-local [tile][z][BM][BN]: shrink the split until they (and the tickets) fit
    while (a.sk > 1 && ((size_t)T * a.sk * BM * BN > ws_floats || T > RADET_SPLIT_COUNTERS)) --a.sk;
    if (a.sk != a_in.sk) {
        const int nKs0 = a.KH * a.KW * (a.Cin / bk);
        a.it_per_split = (nKs0 + a.sk - 1) / a.sk;
    }
    a.n_full = T; a.sk_tail = 1; a.it_per_tail = a.it_per_split;
    const int nKs = a.KH * a.KW * (a.Cin / bk);
    const int rem = T % 256;
    // only for long K loops: on short kernels the extra epilogue launch costs more than the idle tail
    if (a.sk_wgs == 0 && a.cls_nt == 0 && a.sk == 1 && a.partial != nullptr && T > 256 && rem > 0 && rem <= 160 && nKs * bk >= 512 &&
        !radet_switches().no_tail_split) {
        int skt = 256 / rem;
        if (skt > 8) skt = 8;
        if (skt > nKs / 8) skt = nKs / 8;
        while (skt > 1 && ((size_t)rem * skt * BM * BN > ws_floats || rem > RADET_SPLIT_COUNTERS)) --skt;
        if (skt >= 2) {
            a.n_full = T - rem;
            a.sk_tail = skt;
            a.it_per_tail = (nKs + skt - 1) / skt;
        }
    }
    const int tiles = a.sk_wgs > 0 ? a.sk_wgs : a.n_full + (T - a.n_full) * a.sk_tail;
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
                      const int* cls = nullptr);

// Two independent convolutions of identical geometry (the cls- and reg-tower layers of the shared head) as ONE
// launch: twice the tiles per launch halves the wave-quantisation loss on 256 CUs and the launch count.
extern "C" int radet_conv2d_igemm_pair(const float* x0, const float* w0, const float* bias0, const float* addend0,
                                       const float* mask0, float* y0, const float* x1, const float* w1,
                                       const float* bias1, const float* addend1, const float* mask1, float* y1,
                                       const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                                       int tile_override, float* splitk_ws, size_t splitk_ws_floats, void* stream) {
    ConvPtrs second;
    second.x = x1; second.w = w1; second.bias = bias1; second.addend = addend1; second.mask = mask1; second.y = y1;
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

// Tap-subset variant: GEMM rows are a subset of the output rows (out_rows[m] = real output row) that share the
// same set of contributing taps (tap_ids[t] = index into the weight's KTw taps); table is [ntaps][Mp].
// Used for the dgrad of strided convs: one launch per parity class does only the non-zero work.
extern "C" int radet_conv2d_igemm_taps(const float* x, const float* w, const float* addend, const float* mask, float* y,
                                       const int* gather_table, const int* out_rows, const int* tap_ids_host, int ntaps,
                                       int kt_w, int M, int Cin, int Cout, int tile_override, float* splitk_ws,
                                       size_t splitk_ws_floats, void* stream) {
    if (ntaps < 1 || ntaps > 16 || kt_w < ntaps || out_rows == nullptr || tap_ids_host == nullptr) return RADET_ERR_ARG;
    return igemm_impl(x, w, nullptr, addend, mask, y, gather_table, M, Cin, Cout, ntaps, 1, 0, tile_override, splitk_ws,
                      splitk_ws_floats, out_rows, tap_ids_host, kt_w, stream);
}

// Class variant: ALL parity classes of a strided dgrad in one grid.  GEMM rows are the output rows sorted by class, each
// class padded to a multiple of 128 rows (out_rows = -1 on the pad rows, table entries -1); class c owns rows
// [cls_start[c], cls_start[c + 1]) (cls_start[0] = 0, the last class ends at M), runs cls_ntaps[c] <= 4 taps and its tap
// t reads weight tap tap_ids_host[4 c + t]; table is [max ntaps][M].  Order the classes by taps, most first.
extern "C" int radet_conv2d_igemm_classes(const float* x, const float* w, const float* addend, const float* mask, float* y,
                                          const int* gather_table, const int* out_rows, const int* tap_ids_host,
                                          const int* cls_ntaps, const int* cls_start, int ncls, int kt_w, int M, int Cin,
                                          int Cout, int tile_override, float* splitk_ws, size_t splitk_ws_floats,
                                          void* stream) {
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
                      splitk_ws_floats, out_rows, tids, kt_w, stream, nullptr, cls);
}

static int igemm_impl(const float* x, const float* w, const float* bias, const float* addend, const float* mask,
                      float* y, const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                      int tile_override, float* splitk_ws, size_t splitk_ws_floats, const int* out_rows,
                      const int* tap_ids, int kt_w, void* stream, const ConvPtrs* second, const int* cls) {
    const int h16 = (tile_override >> 11) & 1;                 // 0x800: bf16 storage, 0x10000: fp32 output from bf16 inputs
    // 0x2000000: x and w are bf16 plane triples (x rows [3][Cin], w [Cout][taps][3][Cin]; hi + mid + lo = the fp32 value);
    // y / addend / mask / bias stay fp32.  +0x4000000: K step of 16 instead of 32 channels
    const int p3 = ((tile_override >> 25) & 1) && !h16;
    if (h16 || p3) {
        if (Cin % 32 != 0) return RADET_ERR_ARG;               // 16 channel pairs per K step at least
        Cin /= 2;                                              // K is counted in channel pairs (4-byte units) from here on
    }
    if (Cin % 16 != 0 || Cin <= 0 || Cout <= 0 || M <= 0 || gather_table == nullptr || KH * KW > 16) return RADET_ERR_ARG;
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
    a.p[0].x = x; a.p[0].w = w; a.p[0].bias = bias; a.p[0].addend = addend; a.p[0].mask = mask; a.p[0].y = y;
    a.p[1] = a.p[0];
    a.groups = 1;
    if (second != nullptr) { a.p[1] = *second; a.groups = 2; }
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
    int choice = tile_override & 0xFF;
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
    const int stages3 = ((tile_override >> 17) & 1) ? 3 : 2;   // 0x20000: 3 LDS stages (launches that run alone)
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

extern "C" int radet_conv2d_wgrad(const float* dy, const float* x, float* slabs, float* dbias_partials,
                                  const int* gather_table, int M, int Cin, int Cout, int ld_dy, int KH, int KW, int S,
                                  int flags, void* stream) {
    // dy rows must be 16-byte aligned and hold whole float4s for every real channel (pad small heads with zeros)
    if (Cin % 4 != 0 || S < 1 || ld_dy < Cout || (ld_dy & 3) || ((Cout + 3) / 4) * 4 > ld_dy || M <= 0 ||
        gather_table == nullptr)
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
    const int chunks = (a.M + 15) / 16;
    a.chunks_per_split = (chunks + S - 1) / S;
    hipStream_t st = (hipStream_t)stream;
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
        if (j.Cin % 4 != 0 || j.S < 1 || j.ld_dy < j.Cout || (j.ld_dy & 3) || j.M <= 0 || j.gather_table == nullptr ||
            j.Cout % bm != 0 || j.Cin % bm != 0)
            return RADET_ERR_ARG;
        WgradArgs& a = g.p[i];
        a.dy = j.dy; a.x = j.x; a.slabs = j.slabs; a.dbias_partials = j.dbias_partials; a.rowtab = j.gather_table;
        a.M = j.M; a.Mp = radet_gather_table_rows(j.M); a.Cin = j.Cin; a.Cout = j.Cout; a.KH = j.KH; a.KW = j.KW;
        a.ld_dy = j.ld_dy; a.S = j.S; a.dbg = 0; a.math = flags & 1; a.bp32 = 0;
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
