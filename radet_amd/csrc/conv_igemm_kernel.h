// conv_igemmg_kernel: the implicit-GEMM kernel template, shared by the translation units that instantiate it
// (conv_igemm.hip: native fp32 / bf16 / bf16-plane arithmetic; conv_h2.hip: the fp16 hi / lo arithmetic).  gfx950 only.
#pragma once
#include "conv_common.h"
// (one zero page per translation unit: device symbols are not shared across objects without -fgpu-rdc)
static __device__ __attribute__((aligned(16))) float radet_zero_page[512];

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
    constexpr int NS = ((TAG & 128) && (TAG & 16)) ? 2 : BK / 8;   // K = 8-wide k slices per stage (row-interleaved pairs: two K = 16 slices of a 32-channel group)
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
    // bit 6 (with bit 3, bits 3 + 5, or bit 4): fp16 hi / lo arithmetic (common.h "h2") -- two fp16 planes per operand
    // instead of three bf16 ones, 3 v_mfma_f32_32x32x16_f16 per K = 16 step into an accumulator PAIR (hi hi' | hi lo' + lo hi')
    // instead of 6 bf16 MFMAs into one; operands scaled by the exact power of two their amax slots give (ConvPtrs::xs / ws),
    // pair combined and un-scaled right behind the K loop.  With bit 4 the planes arrive as fp16 pairs: rows of 32-channel
    // groups [hi x 32 | lo x 32], 4 bytes per element.
    constexpr bool H2 = (TAG & 64) != 0;
    static_assert(!H2 || X3 || P3, "fp16 hi / lo arithmetic: in-register split (bit 3) or plane pairs (bit 4)");
    // bit 7 with bit 4 (round 6): ROW-INTERLEAVED plane pairs.  A stage of the plane-pair tiles is one 32-channel group = 128
    // contiguous bytes [hi x 32 | lo x 32] of every tile row; fetched plane by plane (two 64-byte pieces per row in two wave
    // loads) the global -> LDS path runs at its 64-byte-chunk rate, 52-55 GB/s per CU for one 8-wave workgroup per CU, against
    // 73-84 GB/s for 128-byte chunks (profiles/round3_fill_probe.txt) -- and the tower GEMM's 48 KiB per stage at 52 GB/s are
    // 0.92 us next to 0.66 us of MFMA issue.  With this bit a wave load fetches 8 rows x 128 bytes (both planes of a row in
    // one line-sized piece): the loaders and LDS images are those of an fp32 tile with a 32-deep K step (BK = 32: 128-byte
    // rows, 8 swizzled 16-byte slots), the reader takes slots 2 s + lh (hi) and 4 + 2 s + lh (lo) of K = 16 slice s
    constexpr bool RI = P3 && (TAG & 128) != 0;
    static_assert(!RI || (H2 && BK == 32), "row-interleaved pairs: fp16 hi / lo planes, 128-byte tile rows");
    constexpr int NPLC = P3 ? (H2 ? 2 : 3) : 1;              // planes of an operand (compute side)
    constexpr int NPL = RI ? 1 : NPLC;                       // plane tiles per LDS stage / loads per row (loader side)
    // bit 5 (with bit 3): the waves divide the K step as well as the tile.  A stage of BK = 16 KD channels is cut into KD
    // k-groups; wave (kg, nh) accumulates ALL BM rows x its BN / WNK columns over k-group kg, and the KD partial tiles of a
    // column group are added through LDS after the K loop.  A wave's operand splits (VALU work) and fragment reads (LDS
    // bandwidth) then serve TMA x TNA accumulator blocks instead of one: (TMA + TNA) splits per 6 TMA TNA MFMAs -- for the
    // 64 x 64 tile 1 split per 6 MFMAs with KD = 4 (2 x 2 blocks per wave) or 1.5 with KD = 2 (2 x 1), against 2 for the
    // 2 x 2-wave tile whose waves each split one A and one B fragment per 6 MFMAs
    constexpr bool KW = (TAG & 32) != 0;
    // bit 7 (with bits 3 + 5 + 6, round 6): K-divided tile on fp16 plane PAIRS.  A pair row is as long as the fp32 row (4 bytes
    // per channel: 32-channel groups [hi x 32 | lo x 32]), so a 64-channel stage is the same 256 contiguous bytes per tile row and
    // the fp32 loaders / LDS images are used unchanged; only the reader differs -- wave kg takes the hi and the lo 16-byte slot of
    // ITS 16 channels (group kg / 2, half kg % 2) straight into the MFMA: no operand split, the same two ds_read_b128 per block
    constexpr bool PQ = (TAG & 128) != 0 && (TAG & 32) != 0;    // (bit 7 with bit 4 instead of bit 5: row-interleaved pairs, below)
    static_assert(!PQ || (KW && H2 && BK == 64), "pair operands in the K-divided tile: 64-channel stages");
    constexpr int KD = KW ? BK / 16 : 1;                                  // k-groups per stage
    constexpr int WNK = NW / KD;                                          // column groups of waves
    constexpr int TMA = KW ? BM / 32 : TM, TNA = KW ? BN / (32 * WNK) : TN;      // accumulator blocks of a wave
    static_assert(!KW || (X3 && !SK && NW == 4 && KD * WNK == NW && TM == 1 && TN == 1 && TMA * TNA == KD && NSTG >= 2 && NSTG <= 4),
                  "K-divided tile: one 32 x 32 block per wave after the reduction");
    static_assert(NW == 4 || NW == 8, "4 or 8 waves");
    static_assert(!P3 || BK == 16 || RI, "plane rows are laid out in 32-channel groups: one group per K stage");
    // NSTG LDS stages: loads run NSTG - 1 K steps ahead of the MFMAs.  3 stages hide more L2 latency (+8 % on the
    // tower GEMM running alone) but cost LDS occupancy, which loses when dgrad and wgrad kernels share the CUs: used
    // for forward launches only (tile_override 0x20000), chosen per shape by the autotuner
    __shared__ __attribute__((aligned(16))) float As[NSTG][NPL * BM * BK];
    __shared__ __attribute__((aligned(16))) float Bs[NSTG][NPL * BN * BK];

    radet_kernarg_warm<sizeof(ConvArgs)>();
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
    epi.out_rows = pin_sgpr(a.out_rows); epi.maskq = pin_sgpr(a.maskq);
    epi.partial = pin_sgpr(a.partial); epi.counters = pin_sgpr(a.counters);
    epi.sk_base = pin_sgpr(a.sk_base); epi.sk_rem = pin_sgpr(a.sk_rem);
    epi.qs = 1.f; epi.qs2 = 2048.f;
    // amax slots (h2 arithmetic: the operands' scales; pair copy of the output: the terms of its bound): the gathers are issued
    // HERE, in front of everything, and reduced behind the prologue's wait for the first tiles -- every launch starts on cold
    // L2s, and a reduction in this place kept the tile loads of each workgroup waiting for the slot's round trip (1.5-2.5 us)
    unsigned raw_xs = 0u, raw_ws = 0u;
    if constexpr ((TAG & 64) != 0) { raw_xs = h2_scale_load(P.xs); raw_ws = h2_scale_load(P.ws); }
    PairScaleRaw raw_q = {0u, 0u, 0u, 0u};
    const bool pair_copy = P.yq != nullptr;                      // (uniform)
    if (pair_copy) raw_q = igemm_pair_scale_load(P);
    const bool pair_store = id == 0 && (int)blockIdx.y == 0 && !SK;
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
    }
    {
        // first gather rows: UNCONDITIONAL loads from clamped indices, selected afterwards -- with the load inside the
        // condition every one of the A_PW loads sat behind its own branch and its own vmcnt(0): four dependent round trips
        // at the head of every workgroup.  (no table: a 1 x 1 / stride 1 conv, GEMM row m reads input row m)
        int araw[A_PW];
        if (a.rowtab) {                                          // (uniform)
            const int* const tab = a.rowtab + (size_t)(nK > 0 ? ld_tap : 0) * a.Mp;
#pragma unroll
            for (int k = 0; k < A_PW; ++k) {
                const int m = m0 + (wave + NW * k) * RPI + lrow;
                araw[k] = tab[m < a.Mp ? m : a.Mp - 1];
            }
        } else {
#pragma unroll
            for (int k = 0; k < A_PW; ++k) {
                const int m = m0 + (wave + NW * k) * RPI + lrow;
                araw[k] = m < a.M ? m : -1;
            }
        }
#pragma unroll
        for (int k = 0; k < A_PW; ++k) arow[k] = (nK > 0 && (A_INSTR % NW == 0 || wave + NW * k < A_INSTR)) ? araw[k] : -1;
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
    // BL (round 6, plane-pair tiles): the tile loads as `buffer_load_dwordx4 ... lds` -- a 32-bit offset VGPR per lane against an
    // SGPR resource of the tensor instead of a 64-bit address pair, padding rows as out-of-range offsets (the buffer returns
    // zeros: no zero page, no pointer select).  The ablation of the tower GEMM (tools/dbg_tower_h2.py) puts the ISSUE of its 48
    // LDS-DMA wave loads per stage at a third of the launch: they add to the MFMA time instead of hiding under it
    constexpr bool BL = RADET_BUFLDS != 0;                  // (every instantiation; RADET_BUFLDS=0 builds the former loads for comparison)
    __amdgpu_buffer_rsrc_t rs_x, rs_w;
    unsigned aoffb[A_PW], woffb[B_PW];
    if constexpr (BL) {
        // (range 4 GiB - 256 B: the out-of-range marker is offset 0xFFFFFFFF; the launchers refuse tensors beyond that)
        rs_x = radet_rsrc(P.x);
        rs_w = radet_rsrc(P.w);
    }
    int pc0 = 0, pwt = 0;                 // (channel chunk, weight tap) of the stage whose pieces are being issued
    auto set_abase = [&]() {
        pc0 = ld_c0; pwt = wtap;
        if constexpr (BL) {
#pragma unroll
            for (int k = 0; k < A_PW; ++k) {
                amask[k] = arow[k] >= 0 ? 0xFFFFFFFFu : 0u;
                aoffb[k] = ((unsigned)(arow[k] & (int)amask[k]) * (unsigned)xld + (unsigned)akq[k]) * 4u;
            }
        } else
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
    if constexpr (BMASK || BL) {
#pragma unroll
        for (int k = 0; k < B_PW; ++k) {
            wmask[k] = wp[k] ? 0xFFFFFFFFu : 0u;
            wbase[k] = wp[k] ? wp[k] : radet_zero_page + lane * 4;
            if constexpr (BL) woffb[k] = wp[k] ? (unsigned)((const char*)wp[k] - (const char*)P.w) : 0u;
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
                if constexpr (BL) {
                    const unsigned voff = (aoffb[k] + 4u * (unsigned)(NPL * pc0 + p * BK)) | ~amask[k];     // padding row: out of range -> zeros
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_x, (lptr_t)(&As[buf][p * BM * BK + ins * 256]), 16, (int)voff, 0, 0, 0);
                } else {
                const float* src;
                if constexpr (BMASK) src = abase[k] + ((unsigned)(NPL * pc0 + p * BK) & amask[k]);
                else src = arow[k] >= 0 ? P.x + (size_t)arow[k] * xld + NPL * pc0 + akq[k] + p * BK : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&As[buf][p * BM * BK + ins * 256]), 16, 0, 0);
                }
            }
        } else {
            constexpr int k = (q - NPL * A_PW) / NPL, p = (q - NPL * A_PW) % NPL;
            const int ins = wave + NW * k;
            if (B_FULL || ins < B_INSTR) {
                if constexpr (BL) {
                    const unsigned voff = (woffb[k] + 4u * (unsigned)(pwt * xld + NPL * pc0 + p * BK)) | ~wmask[k];
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_w, (lptr_t)(&Bs[buf][p * BN * BK + ins * 256]), 16, (int)voff, 0, 0, 0);
                } else {
                const float* src;
                if constexpr (BMASK) src = wbase[k] + ((unsigned)(pwt * xld + NPL * pc0 + p * BK) & wmask[k]);
                else src = wp[k] ? wp[k] + pwt * xld + NPL * pc0 + p * BK : radet_zero_page + lane * 4;
                __builtin_amdgcn_global_load_lds((gptr_t)src, (lptr_t)(&Bs[buf][p * BN * BK + ins * 256]), 16, 0, 0);
                }
            }
        }
    };
    auto advance_stage = [&]() {
        if (cmaj) {
            if (++ld_tap == KTt) { ld_tap = 0; ld_c0 += BK; }
            wtap = a.tap_ids[tbase + ld_tap];
            if (a.rowtab) {
#pragma unroll
                for (int k = 0; k < A_PW; ++k)
                    if (A_FULL || wave + NW * k < A_INSTR)
                        arow[k] = a.rowtab[(size_t)ld_tap * a.Mp + m0 + (wave + NW * k) * RPI + lrow];
            }
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
    f32x16 acc1[H2 ? TMA : 1][H2 ? TNA : 1];                 // h2: the cross terms hi lo' + lo hi' (scaled by 2^11)
#pragma unroll
    for (int i = 0; i < TMA; ++i)
#pragma unroll
        for (int j = 0; j < TNA; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    if constexpr (H2) {
#pragma unroll
        for (int i = 0; i < TMA; ++i)
#pragma unroll
            for (int j = 0; j < TNA; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc1[i][j][r] = 0.f;
    }
    H2Scale sx = {1.f, 1.f, 1.f}, sw = {1.f, 1.f, 1.f};       // power-of-two scales of x and w (set behind the prologue's wait)

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
    if constexpr (H2) { sx = h2_scale_finish(raw_xs); sw = h2_scale_finish(raw_ws); }
    if (pair_copy) {                                             // pair copy of the output: its scale, from the bound
        float qs, qs2;
        igemm_pair_scale(P, raw_q, pair_store, qs, qs2);
        epi.qs = pin_sgpr(qs); epi.qs2 = pin_sgpr(qs2);
    }

    // reader side: tile rows wm*TM*32 + i*32 + li; (i*32) % (RPB*F4) == 0 so swz only depends on li.
    // The fragment reads are inline asm: the compiler would otherwise order every ds_read behind a vmcnt(0) wait on
    // the in-flight LDS-DMA loads (it cannot prove they target the other buffer) and serialise load and compute.
    const int rswz = (li / RPB) % F4;
    unsigned aaddr[NS], baddr[NS];
    unsigned aaddr_lo[RI ? NS : 1], baddr_lo[RI ? NS : 1];
    {
        const unsigned a_lds = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)((wm * TM * 32 + li) * BK * 4);
        const unsigned b_lds = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)((wn * TN * 32 + li) * BK * 4);
#pragma unroll
        for (int s = 0; s < NS; ++s) {
            aaddr[s] = a_lds + 16u * (unsigned)((2 * s + lh) ^ rswz);
            baddr[s] = b_lds + 16u * (unsigned)((2 * s + lh) ^ rswz);
            if constexpr (RI) {                                  // the lo plane's slots of the same row
                aaddr_lo[s] = a_lds + 16u * (unsigned)((4 + 2 * s + lh) ^ rswz);
                baddr_lo[s] = b_lds + 16u * (unsigned)((4 + 2 * s + lh) ^ rswz);
            }
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
                if constexpr (H2) {
                    f16x8 ah[TM], al[TM], bh[TN], bl[TN];
#pragma unroll
                    for (int i = 0; i < TM; ++i) split2_f16(af[2 * g][i], af[2 * g + 1][i], sx, ah[i], al[i]);
#pragma unroll
                    for (int j = 0; j < TN; ++j) split2_f16(bf[2 * g][j], bf[2 * g + 1][j], sw, bh[j], bl[j]);
#pragma unroll
                    for (int i = 0; i < TM; ++i)
#pragma unroll
                        for (int j = 0; j < TN; ++j) mfma_h2(acc[i][j], acc1[i][j], ah[i], al[i], bh[j], bl[j]);
                } else {
                bf16x8 ah[TM], am[TM], al[TM], bh[TN], bm[TN], bl[TN];
#pragma unroll
                for (int i = 0; i < TM; ++i) split3_bf16(af[2 * g][i], af[2 * g + 1][i], ah[i], am[i], al[i]);
#pragma unroll
                for (int j = 0; j < TN; ++j) split3_bf16(bf[2 * g][j], bf[2 * g + 1][j], bh[j], bm[j], bl[j]);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
                }
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
        radet_pipe_barrier();
    };
    if constexpr (KW) {
        const int kg = wave % KD, nh = wave / KD;
        const unsigned ka = (unsigned)(size_t)(lptr_t)(&As[0][0]) + (unsigned)(li * BK * 4);
        const unsigned kb = (unsigned)(size_t)(lptr_t)(&Bs[0][0]) + (unsigned)((nh * TNA * 32 + li) * BK * 4);
        unsigned kaa[2], kba[2];
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            // fp32 rows: k half h of this wave's 16 channels; pair rows (PQ): plane h (0 hi, 1 lo) of them -- slot
            // 8 (kg / 2) + 4 h + 2 (kg % 2) + lh of the 16 slots of a 64-channel row
            const int slot = PQ ? (8 * (kg >> 1) + 4 * h + 2 * (kg & 1) + lh) : (4 * kg + 2 * h + lh);
            kaa[h] = ka + 16u * (unsigned)(slot ^ rswz);
            kba[h] = kb + 16u * (unsigned)(slot ^ rswz);
        }
        f32x4 fa[2][2][TMA], fb[2][2][TNA];              // [fragment set][k half][block]
        constexpr int NRD = 2 * (TMA + TNA);
        // fragment read r of stage buffer `buf` (run time: NSTG buffers in rotation) into fragment set PP (compile time: the
        // sets alternate): A blocks, then B blocks, k half 0 then 1
        auto read_one = [&](int buf, auto ppc, auto rc) {
            constexpr int PP = decltype(ppc)::value, r = decltype(rc)::value;
            constexpr int RO = 32 * BK * 4;
            constexpr int h = r / (TMA + TNA), e = r % (TMA + TNA);
            if constexpr (e < TMA) lds_read128<e * RO>(fa[PP][h][e], kaa[h] + (unsigned)(buf * (BM * BK * 4)));
            else lds_read128<(e - TMA) * RO>(fb[PP][h][e - TMA], kba[h] + (unsigned)(buf * (BN * BK * 4)));
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
        // NSTG stages, several workgroups per CU: loads of stage it + NSTG - 1 at the head of stage it (into the buffer stage
        // it - 1 has left), the fragment reads of stage it + 1 right behind the barrier that publishes it.  With more than two
        // stages a stage waits for everything but the NSTG - 2 newest stages' loads (in-order return; round 5: the K-divided
        // tiles are bound by the L2 -> LDS latency of their stages, profiles/round3_pmc_stalls.txt)
        auto stage_kw = [&](int buf, auto ppc, int it) {
            constexpr int PP = decltype(ppc)::value;
            if (it + NSTG - 1 < nK) issue_stage(buf + NSTG - 1 >= NSTG ? buf - 1 : buf + NSTG - 1);
            lds_wait<0>();
            pin(ppc);
            if constexpr (H2) {
                f16x8 ah[TMA], al[TMA], bh[TNA], bl[TNA];
                if constexpr (PQ) {                              // the fragments ARE the planes
#pragma unroll
                    for (int i = 0; i < TMA; ++i) { ah[i] = __builtin_bit_cast(f16x8, fa[PP][0][i]); al[i] = __builtin_bit_cast(f16x8, fa[PP][1][i]); }
#pragma unroll
                    for (int j = 0; j < TNA; ++j) { bh[j] = __builtin_bit_cast(f16x8, fb[PP][0][j]); bl[j] = __builtin_bit_cast(f16x8, fb[PP][1][j]); }
                } else {
#pragma unroll
                for (int i = 0; i < TMA; ++i) split2_f16(fa[PP][0][i], fa[PP][1][i], sx, ah[i], al[i]);
#pragma unroll
                for (int j = 0; j < TNA; ++j) split2_f16(fb[PP][0][j], fb[PP][1][j], sw, bh[j], bl[j]);
                }
#pragma unroll
                for (int i = 0; i < TMA; ++i)
#pragma unroll
                    for (int j = 0; j < TNA; ++j) mfma_h2(acc[i][j], acc1[i][j], ah[i], al[i], bh[j], bl[j]);
            } else {
            bf16x8 ah[TMA], am[TMA], al[TMA], bh[TNA], bm[TNA], bl[TNA];
#pragma unroll
            for (int i = 0; i < TMA; ++i) split3_bf16(fa[PP][0][i], fa[PP][1][i], ah[i], am[i], al[i]);
#pragma unroll
            for (int j = 0; j < TNA; ++j) split3_bf16(fb[PP][0][j], fb[PP][1][j], bh[j], bm[j], bl[j]);
#pragma unroll
            for (int i = 0; i < TMA; ++i)
#pragma unroll
                for (int j = 0; j < TNA; ++j) mfma_x3(acc[i][j], ah[i], am[i], al[i], bh[j], bm[j], bl[j]);
            }
            __builtin_amdgcn_sched_barrier(0);
            if (NSTG >= 3 && it + NSTG - 1 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (NSTG - 2)) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            radet_pipe_barrier();
        };
        static_assert(NSTG == 2 || LOADS > 0, "deep K-divided pipelines count on every wave owning the same loads per stage");
        int buf = 0;
        auto next_buf = [&]() { buf = buf + 1 == NSTG ? 0 : buf + 1; };
        if (nK > 0) static_for<0, NRD>([&](auto rc) { read_one(0, std::integral_constant<int, 0>{}, rc); });
        for (int it = 0; it < nK; it += 2) {
            stage_kw(buf, std::integral_constant<int, 0>{}, it);
            next_buf();
            if (it + 1 < nK) {
                static_for<0, NRD>([&](auto rc) { read_one(buf, std::integral_constant<int, 1>{}, rc); });
                stage_kw(buf, std::integral_constant<int, 1>{}, it + 1);
                next_buf();
                if (it + 2 < nK) static_for<0, NRD>([&](auto rc) { read_one(buf, std::integral_constant<int, 0>{}, rc); });
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
        f32x4 fa[2][NPLC][TM], fb[2][NPLC][TN];
        constexpr int NRD = NPLC * (TM + TN);
        constexpr int NT = H2 ? 3 : 6;                         // plane products (MFMA groups) per slice
        // fragment read r of slice s of buffer BUF into fragment set pp: order A hi, B hi, A mid, B mid, A lo, B lo
        auto read_one = [&](auto bufc, auto sc, auto ppc, auto rc) {
            constexpr int BUF = decltype(bufc)::value, s = decltype(sc)::value, pp = decltype(ppc)::value, r = decltype(rc)::value;
            constexpr int AO = BUF * NPL * BM * BK * 4, BO = BUF * NPL * BN * BK * 4, RO = 32 * BK * 4;
            // (h2: the lo plane's reads go B first -- group 1 = hi lo' needs B lo, group 2 = lo hi' A lo: see the waits in slice())
            constexpr int pl = r / (TM + TN), e = (H2 && pl == 1) ? (r % (TM + TN) + TM) % (TM + TN) : r % (TM + TN);
            // (the buffer offset goes into the address register: a ds_read immediate holds 16 bits)
            if constexpr (RI) {                                  // both planes in the row: another slot, not another tile
                if constexpr (e < TM) lds_read128<e * RO>(fa[pp][pl][e], (pl ? aaddr_lo[s] : aaddr[s]) + (unsigned)AO);
                else lds_read128<(e - TM) * RO>(fb[pp][pl][e - TM], (pl ? baddr_lo[s] : baddr[s]) + (unsigned)BO);
            } else
            if constexpr (e < TM) lds_read128<pl * BM * BK * 4 + e * RO>(fa[pp][pl][e], aaddr[s] + (unsigned)AO);
            else lds_read128<pl * BN * BK * 4 + (e - TM) * RO>(fb[pp][pl][e - TM], baddr[s] + (unsigned)BO);
        };
        // the six MFMA groups of fragment set PP; behind group g: reads [g NRD / 5, (g + 1) NRD / 5) of the next fragment set
        // (none behind the last group: they would not be back by the next slice) and, with LD, the pieces of the refill
        auto slice = [&](auto ppc, auto rbufc, auto rsc, bool do_read, auto ldc, int ld_buf, bool do_load) {
            constexpr int PP = decltype(ppc)::value;
            constexpr bool LD = decltype(ldc)::value;
#pragma unroll
            for (int pl = 0; pl < NPLC; ++pl) {
#pragma unroll
                for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[PP][pl][i]));
#pragma unroll
                for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[PP][pl][j]));
            }
            static_for<0, NT>([&](auto tc) {
                constexpr int t = decltype(tc)::value;          // terms: hi hi, mid hi, hi mid, mid mid, lo hi, hi lo
                constexpr int pa = H2 ? (t == 2 ? 1 : 0) : (t == 1 || t == 3 ? 1 : (t == 4 ? 2 : 0));      // (h2: hi hi, hi lo, lo hi)
                constexpr int pb = H2 ? (t == 1 ? 1 : 0) : (t == 2 || t == 3 ? 1 : (t == 5 ? 2 : 0));
                // (h2: the slice started with only the hi fragments waited for -- the lo fragments, read behind them, are due now;
                // the reads issued behind group 0 may stay in flight)
                // in order: [A hi, B hi | B lo | A lo] of this set, then the NRD / 2 reads issued behind each earlier group
                // (the fragments a wait releases are pinned BEHIND it: an MFMA has no other tie to the wait, and the compiler hoisted
                // the first MFMA of groups 1 / 2 above it -- a lo fragment could be consumed before its ds_read had returned: rare
                // last-bit differences from run to run in the 64 x 64 tile, whose group is a single MFMA; round 6)
                if constexpr (H2 && t == 1) {
                    if (do_read) lds_wait<TM + NRD / (NT - 1)>(); else lds_wait<TM>();
#pragma unroll
                    for (int j = 0; j < TN; ++j) asm volatile("" : "+v"(fb[PP][1][j]));
                }
                if constexpr (H2 && t == 2) {
                    if (do_read) lds_wait<2 * (NRD / (NT - 1))>(); else lds_wait<0>();
#pragma unroll
                    for (int i = 0; i < TM; ++i) asm volatile("" : "+v"(fa[PP][1][i]));
                }
                if (!RADET_P3_DBG || !(a.dbg & 2)) {
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j) {
                        if constexpr (H2) {
                            f32x16& dst = t == 0 ? acc[i][j] : acc1[i][j];
                            dst = __builtin_amdgcn_mfma_f32_32x32x16_f16(__builtin_bit_cast(f16x8, fa[PP][pa][i]),
                                                                         __builtin_bit_cast(f16x8, fb[PP][pb][j]), dst, 0, 0, 0);
                        } else {
                            acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, fa[PP][pa][i]),
                                                                                __builtin_bit_cast(bf16x8, fb[PP][pb][j]), acc[i][j], 0, 0, 0);
                        }
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (LD) {
                    if (do_load) {
                        if constexpr (t == 0) set_abase();
                        static_for<t * NPIECE / NT, (t + 1) * NPIECE / NT>([&](auto qc) { issue_piece(ld_buf, qc); });
                        if constexpr (t == NT - 1) advance_stage();
                    }
                }
                if constexpr (t < NT - 1) {
                    if (do_read)
                        static_for<t * NRD / (NT - 1), (t + 1) * NRD / (NT - 1)>([&](auto rc) { read_one(rbufc, rsc, std::integral_constant<int, PP ^ 1>{}, rc); });
                }
                __builtin_amdgcn_sched_barrier(0);
            });
        };
        auto stage_p3 = [&](auto bufc, int it) {
            constexpr int BUF = decltype(bufc)::value;
            static_for<0, NS>([&](auto sc) {
                constexpr int s = decltype(sc)::value;
                constexpr int PP = s & 1;                                      // parity of the global slice index (NS even)
                // fragment set PP has arrived -- h2: its hi fragments (read first); group 0 = hi hi' starts on them
                if constexpr (H2 && s + 1 < NS) lds_wait<NRD / 2>(); else lds_wait<0>();
                if constexpr (s + 1 < NS) {
                    slice(std::integral_constant<int, PP>{}, bufc, std::integral_constant<int, s + 1>{},
                          !RADET_P3_DBG || !(a.dbg & 4), std::false_type{}, 0, false);
                } else {
                    // (this wave has no read of buffer BUF in flight any more)
                    if (NSTG >= 3 && it + NSTG - 1 < nK) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LOADS * (NSTG - 2)) : "memory");
                    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                    radet_pipe_barrier();                                      // stage it + 1 landed, buffer BUF released
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
    if constexpr (H2) {                                             // accumulator pair -> the product in fp32 units
#pragma unroll
        for (int i = 0; i < TMA; ++i)
#pragma unroll
            for (int j = 0; j < TNA; ++j) h2_combine(acc[i][j], acc1[i][j], sx.inv, sw.inv);
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


// Launch plan of one implicit-GEMM call for a BM x BN tile: stream-K partition, split-K that fits the workspace, tail split
// of the T % 256 left-over tiles.  Fills the schedule fields of `a`; returns the number of workgroups along x.
template <int BM, int BN>
static int igemm_plan(ConvArgs& a, int tag, int bk, size_t ws_floats, int skw, bool no_tail_split) {
    const int sk_in = a.sk;
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
    if (a.sk != sk_in) {
        const int nKs0 = a.KH * a.KW * (a.Cin / bk);
        a.it_per_split = (nKs0 + a.sk - 1) / a.sk;
    }
    a.n_full = T; a.sk_tail = 1; a.it_per_tail = a.it_per_split;
    const int nKs = a.KH * a.KW * (a.Cin / bk);
    const int rem = T % 256;
    // only for long K loops: on short kernels the extra epilogue launch costs more than the idle tail
    if (a.sk_wgs == 0 && a.cls_nt == 0 && a.sk == 1 && a.partial != nullptr && T > 256 && rem > 0 && rem <= 160 && nKs * bk >= 512 &&
        !no_tail_split) {
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
    return a.sk_wgs > 0 ? a.sk_wgs : a.n_full + (T - a.n_full) * a.sk_tail;
}
