// Pieces shared by the implicit-GEMM kernels (conv_igemm.hip; kept apart so that experimental tiles can be built as their own
// translation units against the same epilogue / split-K code): launch arguments, the fused epilogue,
// the in-launch split-K reduction, the exact fp32 -> three-bf16-planes split.  gfx950 only.
#pragma once
#include "common.h"
#include "../../include/radet_hip.h"
#include <stdlib.h>
#include <type_traits>

#define RADET_SPLIT_COUNTERS 16384     // arrival tickets at the head of the split workspace
#ifndef RADET_P3_DBG
#define RADET_P3_DBG 0                 // 1: ConvArgs::dbg ablation switches of the plane-operand loop are live (experiments)
#endif

struct ConvPtrs {
    const float* x;       // input rows [*, Cin]
    const float* w;       // [Cout][KH*KW][Cin]
    const float* bias;    // [Cout] or null
    const float* addend;  // [M][Cout] or null (added before relu / mask)
    const float* mask;    // [M][Cout] or null: out = mask > 0 ? out : 0   (ReLU backward)
    float* y;             // [M][Cout]
    // fp16 hi / lo arithmetic (common.h "h2"): amax slots of x and w (bit pattern of the largest magnitude, or of a bound on
    // it; the operands are scaled by 2^radet_h2_exp(slot) before they are split / were scaled when their planes were written)
    // and the slot this launch raises to the largest |y| it stores (null: not tracked)
    const unsigned* xs;
    const unsigned* ws;
    unsigned* ys;
    // optional second output: y once more as fp16 plane pairs (rows [2][Cout], Cout % 32 == 0; fp32 outputs only) -- the x
    // operand of the next conv, so that it runs without an operand split in its K loop.  The pairs' power-of-two scale must be
    // known BEFORE the first store, i.e. before y's largest magnitude is: it comes from a bound that is complete when the
    // launch starts,  |y| <= amax(x) * max_n sum_k |w[n][k]|  +  max|bias|  +  amax(addend)   (ReLU and mask only shrink),
    // built from the TRUE largest magnitude of x (xt: the slot its producer raised, not the bound its own pairs were scaled
    // with -- bounds do not compound from layer to layer), the weights' largest channel L1 norm (wl1, from the fold) and the
    // slots of bias / addend (bs / as, null: absent).  The bound's bit pattern is stored to yqs for the consumer.
    _Float16* yq;
    unsigned* yqs;
    const unsigned* xt;
    const unsigned* wl1;
    const unsigned* bs;
    const unsigned* as;
};

struct ConvArgs {
    ConvPtrs p[2];        // 1 or 2 independent problems of identical geometry in one launch (cls / reg tower)
    int groups;
    const int* rowtab;    // [KH*KW][Mp] input row of (output row, tap) or -1 (built once per geometry)
    float* partial;       // split-K / tail split: tile-local partial tiles [split tile][z][BM][BN]
    int* counters;        // arrival tickets, one per split tile (zero before the launch, left zero by it)
    const int* out_rows;  // optional [M]: output row of GEMM row m (parity-class dgrad of strided convs)
    int tap_ids[16];      // weight tap index of table tap t (identity unless a tap subset is used)
    int KTw;              // taps in the weight tensor (row stride of w is KTw*Cin)
    int M, Mp, Cin, Cout, KH, KW;
    int relu;
    int sk, it_per_split; // K-stage range of block z = blockIdx.y: [z*it_per_split, min(nK, (z+1)*it_per_split))
    // tail split: tiles [0, n_full) run whole; the T % 256 left-over tiles (which would otherwise occupy a mostly
    // empty last round on the 256 CUs) are cut sk_tail ways along K, their partial tiles reduced by a second pass
    int n_full, sk_tail, it_per_tail;
    // io = 0: fp32 tensors.  io = 1 / 2 (bf16 storage): x, w, addend, mask are bf16 and Cin counts PAIRS of channels
    // (a 4-byte unit, so the loaders and LDS layouts are those of the fp32 kernel); y is bf16 (1) or fp32 (2)
    int io;
    // stream-K: sk_wgs persistent workgroups share the T * nK K-stages of the launch evenly (workgroup v owns stages
    // [v * sk_base + min(v, sk_rem), ...) across tile boundaries); tiles cut by a boundary are reduced in the launch
    int sk_wgs, sk_base, sk_rem;
    // class launch (all parity classes of a strided dgrad in one grid): GEMM rows [cls_b[c-1], cls_b[c]) belong to class
    // c (boundaries are multiples of 128, pad rows have out_rows = -1), class c runs (cls_nt >> 4c) & 15 taps and its
    // tap t reads weight tap tap_ids[4c + t]; rowtab is [max taps][Mp].  cls_nt = 0: one tap list for all rows.
    int cls_nt, cls_b[3];
    int dbg;              // experiments (RADET_DBG_IGEMM, plane-operand kernels): 1 no tile loads after the prologue, 2 no MFMAs,
                          // 4 no fragment reads after the first
    int maskq;            // 1: `mask` is an fp16 plane-pair tensor (rows [2][Cout]: an activation that exists ONLY as pairs, round 6)
};

__device__ __forceinline__ float ld_act(const float* p, size_t o, int io) {
    return io ? (float)reinterpret_cast<const __bf16*>(p)[o] : p[o];
}
__device__ __forceinline__ void st_out(float* p, size_t o, float v, int io) {
    if (io == 1) reinterpret_cast<__bf16*>(p)[o] = (__bf16)v;      // v_cvt_pk_bf16_f32: round to nearest even
    else p[o] = v;
}


typedef const __attribute__((address_space(1))) void* gptr_t;
typedef __attribute__((address_space(3))) void* lptr_t;

// ---- global -> LDS tile loads (LDS-DMA, 1 KiB per wave instruction).  Round 6: issued as `buffer_load_dwordx4 ... lds` -- one
// 32-bit offset VGPR per lane against an SGPR resource of the tensor, a lane without a source (padding row, channel beyond
// the tensor) gets an out-of-range offset and the buffer unit returns zeros -- instead of `global_load_lds_dwordx4` with a
// 64-bit address pair per lane and a select against a zero page.  An ablation of the tower GEMM (tools/dbg_tower_h2.py) had
// put the ISSUE of these loads, not their latency, at a third of the launch; with the buffer form the tower GEMM takes 90
// instead of 98 us alone, the K-divided backbone tile 42.0 instead of 43.6 us in the step, the bf16-storage step 4.95 instead
// of 5.07 ms (bit-identical results).  RADET_BUFLDS=0 builds the former loads.  Tensors are addressed up to 4 GiB - 256 B.
#ifndef RADET_BUFLDS
#define RADET_BUFLDS 1
#endif
__device__ __forceinline__ __amdgpu_buffer_rsrc_t radet_rsrc(const void* base) {
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, 0xFFFFFF00u, 0x00020000);
}
// (a macro, not a function template: the host pass cannot substitute the LDS-pointer parameter)
#if RADET_BUFLDS
#define radet_lds_load16(base, ok, elem_off, dst) \
    __builtin_amdgcn_raw_ptr_buffer_load_lds(radet_rsrc(base), (dst), 16, \
        (int)((ok) ? (unsigned)((elem_off) * sizeof(*(base))) : 0xFFFFFFFFu), 0, 0, 0)
#else
#define radet_lds_load16(base, ok, elem_off, dst) \
    __builtin_amdgcn_global_load_lds((gptr_t)((ok) ? (const void*)((base) + (elem_off)) : (const void*)(radet_zero_page + (threadIdx.x & 63) * 4)), \
                                     (dst), 16, 0, 0)
#endif

// The barrier of a software-pipelined K loop: the bare instruction behind the loop's own waits.  `__syncthreads()` is a
// workgroup-scope fence + s_barrier, and for the fence the compiler drains EVERY outstanding memory operation in front of it
// (`s_waitcnt vmcnt(0) lgkmcnt(0)`) -- directly behind a hand-placed `s_waitcnt vmcnt(N)` that was meant to leave the newest
// stages' tile loads in flight.  The 3- and 4-stage pipelines of rounds 3-5 therefore ran as 2-stage ones with idle LDS
// ("a third stage changes nothing"); found in round 6 in the ISA of an NSTG = 3 instantiation.
// Contract of the caller: (1) it has waited for the tile loads it is about to publish (vmcnt) -- an LDS-DMA load has written
// its LDS bytes when the counter drops; (2) no LDS read of the buffer that will be refilled behind the barrier is outstanding
// (lgkmcnt(0) here: the fragment reads are consumed before the barrier in every loop); (3) the LDS accesses around it are
// volatile inline asm / side-effecting builtins, which the compiler does not move across the (side-effecting) barrier builtin;
// the empty asm statements keep plain memory accesses on their side as well.
__device__ __forceinline__ void radet_pipe_barrier() {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}

__device__ __forceinline__ int find_seg(const RadetSegs& s, int m) {
    int l = 0;
#pragma unroll
    for (int i = 0; i < RADET_MAX_SEG - 1; ++i)
        if (i < s.nseg - 1 && m >= s.s[i].row_end) l = i + 1;
    return l;
}

struct PixCtx {  // decoded output pixel, ready for the per-tap gather
    int by, bx;  // oy*so + off, ox*so + off
    int Hi, Wi;  // Hi == 0 marks an out-of-range row
    int base;    // input row of (n, 0, 0)
};

__device__ __forceinline__ PixCtx decode_pixel(const RadetSegs& segs, int m, int M, int so, int off) {
    PixCtx p;
    p.Hi = 0; p.Wi = 0; p.by = 0; p.bx = 0; p.base = 0;
    if (m < M) {
        const int l = find_seg(segs, m);
        const RadetSeg& sg = segs.s[l];
        const int local = m - sg.row_begin;
        const int hw = sg.Ho * sg.Wo;
        const int n = local / hw;
        const int rem = local - n * hw;
        const int oy = rem / sg.Wo;
        const int ox = rem - oy * sg.Wo;
        p.by = oy * so + off;
        p.bx = ox * so + off;
        p.Hi = sg.Hi;
        p.Wi = sg.Wi;
        p.base = sg.in_row_off + n * sg.Hi * sg.Wi;
    }
    return p;
}

// returns the input row index for (pixel, tap) or -1
__device__ __forceinline__ int gather_row(const PixCtx& p, int r, int q, int sr, int div) {
    int iy = p.by + r * sr, ix = p.bx + q * sr;
    if (div > 1) {
        if ((iy % div) != 0 || (ix % div) != 0) return -1;
        iy /= div;
        ix /= div;
    }
    if (iy < 0 || iy >= p.Hi || ix < 0 || ix >= p.Wi) return -1;
    return p.base + iy * p.Wi + ix;
}

// Fused epilogue of one block tile: y = relu?(acc + bias (+ addend)) masked by mask > 0.  All loads of an accumulator
// tile (output-row indirection, residual, ReLU mask) are issued back to back BEFORE anything waits for them: written as
// one loop with per-element branches the compiler emitted load -> s_waitcnt vmcnt(0) -> branch -> load -> ... for every
// one of the 16 accumulator registers, i.e. up to 48 fully serialised L2 / HBM round trips per tile (44 % of the wave
// cycles of the K = 256 residual layers were spent parked there).  Out-of-range rows / columns read a clamped address
// and are dropped at the store; the storage type IO (0 fp32, 1 bf16, 2 bf16 in / fp32 out) is a compile-time branch.
template <int IO>
__device__ __forceinline__ float ld_act_t(const float* p, size_t o) {
    if constexpr (IO != 0) return (float)reinterpret_cast<const __bf16*>(p)[o];
    else return p[o];
}
template <int IO>
__device__ __forceinline__ void st_out_t(float* p, size_t o, float v) {
    if constexpr (IO == 1) reinterpret_cast<__bf16*>(p)[o] = (__bf16)v;   // v_cvt_pk_bf16_f32: round to nearest even
    else p[o] = v;
}

// Everything the code after the K loop needs from the argument struct, loaded before the loop and pinned in SGPRs:
// left to the compiler these become s_loads (each behind its own wait) between the last barrier and the first store.
struct EpiArgs {
    float qs, qs2;        // ConvPtrs::yq: 2^e, 2^(e + 11) of the output bound (igemm_pair_scale)
    int M, Cout, relu, io;
    int maskq;            // ConvArgs::maskq
    const int* out_rows;
    float* partial;       // split-K / stream-K workspace (ConvArgs::partial, ::counters)
    int* counters;
    int sk_base, sk_rem;  // stream-K partition
};
template <class T>
__device__ __forceinline__ T pin_sgpr(T v) {
    asm volatile("" : "+s"(v));
    return v;
}

template <int BM, int BN, int WM, int WN, int IO>
__device__ __forceinline__ void igemm_epilogue(const EpiArgs& a, const ConvPtrs& P,
                                               f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)], int m0, int n0, int wm, int wn,
                                               int li, int lh) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    const bool has_add = P.addend != nullptr, has_mask = P.mask != nullptr, has_rows = a.out_rows != nullptr;   // uniform
    const bool interior = !has_rows && m0 + BM <= a.M && n0 + BN <= a.Cout;                                        // uniform
    float amax = 0.f;                                            // largest |y| this lane stores (P.ys: the tensor's amax slot)
#pragma unroll
    for (int i = 0; i < TM; ++i) {
        size_t obase[16];
        bool rvalid[16];
        if (has_rows) {
            int orow[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                orow[r] = a.out_rows[row < a.M ? row : 0];
                rvalid[r] = row < a.M && orow[r] >= 0;             // (-1: pad row between the classes of a class launch)
                if (orow[r] < 0) orow[r] = 0;
            }
#pragma unroll
            for (int r = 0; r < 16; ++r) obase[r] = (size_t)orow[r] * a.Cout;
        } else {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = m0 + (wm * TM + i) * 32 + (r & 3) + 8 * (r >> 2) + 4 * lh;
                rvalid[r] = row < a.M;
                obase[r] = (size_t)(rvalid[r] ? row : 0) * a.Cout;
            }
        }
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int col = n0 + (wn * TN + j) * 32 + li;
            const bool cvalid = col < a.Cout;
            const int cc = cvalid ? col : 0;
            const float bv = P.bias ? P.bias[cc] : 0.f;
            float av[16], mv[16];
            if (has_add) {
#pragma unroll
                for (int r = 0; r < 16; ++r) av[r] = ld_act_t<IO>(P.addend, obase[r] + cc);
            }
            if (has_mask) {
                if (IO == 0 && a.maskq) {
                    // the mask tensor exists only as fp16 plane pairs (hi + 2^-11 lo of the scaled value): positive iff hi > 0,
                    // or hi == 0 and lo > 0 (an element below fp16's range in BOTH halves is zero for its consumers as well)
                    const unsigned short* mq = reinterpret_cast<const unsigned short*>(P.mask) + radet_pair_off(cc);
                    unsigned short mh[16], ml[16];
#pragma unroll
                    for (int r = 0; r < 16; ++r) { mh[r] = mq[2 * obase[r]]; ml[r] = mq[2 * obase[r] + 32]; }
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const unsigned sel = (mh[r] & 0x7FFFu) ? mh[r] : ml[r];
                        mv[r] = ((sel & 0x7FFFu) != 0u && (sel & 0x8000u) == 0u) ? 1.f : 0.f;
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) mv[r] = ld_act_t<IO>(P.mask, obase[r] + cc);
                }
            }
            float out[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) {                      // straight-line arithmetic: one wait for the loads above
                float v = acc[i][j][r] + bv;
                if (has_add) v += av[r];
                if (a.relu) v = fmaxf(v, 0.f);
                if (has_mask) v = mv[r] > 0.f ? v : 0.f;
                out[r] = v;
            }
            // stores last, with nothing left in flight that they would have to wait for (on gfx9-class hardware a
            // store behind a conservative vmcnt(0) also waits for the store before it); interior tiles store unguarded
            if constexpr (IO == 0) {
                if (P.yq) {
                    // The same values as fp16 plane pairs.  A lane holds ONE channel of 16 rows; a row's 32-channel group is 128
                    // bytes [hi x 32 | lo x 32].  Neighbouring lanes swap halves (one DPP move per row pair) so that every lane
                    // stores 4 bytes: even lanes the hi halves of channels (c, c + 1), odd lanes the lo halves of (c - 1, c) --
                    // one store instruction per row covers the whole 128-byte group, like the fp32 store next to it.
                    const bool odd = li & 1;
                    unsigned* const q = reinterpret_cast<unsigned*>(reinterpret_cast<unsigned short*>(P.yq) +
                                                                    radet_pair_off(cc & ~1) + (odd ? 32 : 0));
#pragma unroll
                    for (int r = 0; r < 16; r += 2) {
                        unsigned h, l;
                        radet_split2(out[r], out[r + 1], a.qs, a.qs2, h, l);     // h = hi(row r) | hi(row r + 1) << 16, l alike
                        const unsigned send = odd ? h : l;                         // what the neighbour stores
                        const unsigned recv = (unsigned)__builtin_amdgcn_mov_dpp((int)send, 0xB1, 0xF, 0xF, true);   // lanes 2k <-> 2k + 1
                        const unsigned lo_ch = odd ? recv : h, hi_ch = odd ? l : recv;                     // channel c & ~1, channel c | 1
                        const unsigned w0 = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x05040100u);              // row r:     lo_ch[15:0] | hi_ch[15:0] << 16
                        const unsigned w1 = __builtin_amdgcn_perm(hi_ch, lo_ch, 0x07060302u);              // row r + 1: the upper halves
                        if (interior || (cvalid && rvalid[r])) q[obase[r]] = w0;
                        if (interior || (cvalid && rvalid[r + 1])) q[obase[r + 1]] = w1;
                    }
                }
            }
            const bool has_y = P.y != nullptr;                  // (uniform; null: the output exists only as the pairs above)
            if (interior) {
                if (has_y) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) st_out_t<IO>(P.y, obase[r] + cc, out[r]);
                }
                if (P.ys) {
#pragma unroll
                    for (int r = 0; r < 16; ++r) amax = fmaxf(amax, fabsf(out[r]));
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r)
                    if (cvalid && rvalid[r]) {
                        if (has_y) st_out_t<IO>(P.y, obase[r] + cc, out[r]);
                        amax = fmaxf(amax, fabsf(out[r]));
                    }
            }
        }
    }
    if (P.ys) radet_amax_publish(amax, P.ys);                      // (uniform branch; one atomicMax per wave)
}

// Split episodes (split-K over the whole grid, or the K-split left-over tiles of a tail split) are reduced INSIDE the
// launch: every workgroup of a tile writes its raw partial tile (tile-local layout), publishes it with one agent-scope
// release and draws an arrival ticket; the workgroup that draws the last ticket acquires, re-reads all partial tiles in
// split order z = 0, 1, ... (so the sum does not depend on the arrival order: deterministic, and equal to what a
// separate reduction pass would produce) and runs the fused epilogue.  This replaces 60-80 reduction launches per
// train step that sat between dependent GEMMs on the critical chain.  `ws` is any LDS word all waves are done with.
// partial tile = register image: [wave][i][j][lane][16 accumulator floats] -> every lane moves 64 contiguous bytes with
// 16-byte accesses.  The stores are write-through (sc1), so publishing needs no L2 write-back.
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void partial_write(float* dst, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)]) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
    typedef float f32x4_ __attribute__((ext_vector_type(4)));
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(dst, 0, BM * BN * 4, 0x00020000);
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const int off = ((((wave * TM + i) * TN + j) * 64 + lane) * 16) * 4;
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4_ v = {acc[i][j][4 * q], acc[i][j][4 * q + 1], acc[i][j][4 * q + 2], acc[i][j][4 * q + 3]};
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rs, off + 16 * q, 0, 16);   // aux 16 = sc1
            }
        }
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void partial_add(const float* pz, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)]) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), lane = threadIdx.x & 63;
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j) {
            const float4* src = reinterpret_cast<const float4*>(pz + (((wave * TM + i) * TN + j) * 64 + lane) * 16);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 v = src[q];
                acc[i][j][4 * q] += v.x; acc[i][j][4 * q + 1] += v.y; acc[i][j][4 * q + 2] += v.z; acc[i][j][4 * q + 3] += v.w;
            }
        }
}

template <class A>
__device__ __forceinline__ int streamk_start(const A& a, int v) { return v * a.sk_base + (v < a.sk_rem ? v : a.sk_rem); }
template <class A>
__device__ __forceinline__ int streamk_owner(const A& a, int it) {       // workgroup that owns K-stage `it`
    const int big = (a.sk_base + 1) * a.sk_rem;
    return it < big ? it / (a.sk_base + 1) : a.sk_rem + (it - big) / a.sk_base;
}

// Stream-K: a workgroup's K-stage range [cur, cur + nseg) inside tile `tile` does not cover the tile.  Every contributor
// writes its partial tile to its own slot (workgroup v has at most two cut tiles: the one its range starts in -> slot 0,
// the one it ends in -> slot 1) and adds its stage count to the tile's counter; the contributor that completes the
// count (nKs) re-reads ALL partials in K order (ascending workgroup), so the sum is independent of the arrival order.
// Returns true in the workgroup that has to run the epilogue (acc = the reduced tile).
template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ bool streamk_publish(const EpiArgs& a, f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)],
                                                int tile, int nseg, int nKs, int v, int slot, volatile int* ws) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    partial_write<BM, BN, WM, WN>(a.partial + (size_t)(2 * v + slot) * BM * BN, acc);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0)
        ws[0] = __hip_atomic_fetch_add(&a.counters[tile], nseg, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    __syncthreads();
    const bool last = ws[0] + nseg == nKs;
    __syncthreads();                                           // ws is LDS tile memory: the next segment's loads may reuse it
    if (!last) return false;
    // every wave acquires for itself (agent scope: invalidates this CU's L1 before the partial tiles of other
    // workgroups, published with write-through stores + a drained vmcnt + the ticket, are read with plain loads)
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    if (threadIdx.x == 0)
        __hip_atomic_store(&a.counters[tile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
    __syncthreads();
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
    const int t0 = tile * nKs;
    const int v0 = streamk_owner(a, t0), v1 = streamk_owner(a, t0 + nKs - 1);
    for (int u = v0; u <= v1; ++u) {
        const int sl = streamk_start(a, u) >= t0 ? 0 : 1;      // starts inside this tile -> its first cut tile
        partial_add<BM, BN, WM, WN>(a.partial + (size_t)(2 * u + sl) * BM * BN, acc);
    }
    return true;
}

template <int BM, int BN, int WM, int WN>
__device__ __forceinline__ void igemm_store(const EpiArgs& e, const ConvPtrs& P,
                                            f32x16 (&acc)[BM / (WM * 32)][BN / (WN * 32)], int m0, int n0, int nsplit,
                                            int ctile, int z, int wm, int wn, int li, int lh, volatile int* ws) {
    constexpr int TM = BM / (WM * 32), TN = BN / (WN * 32);
    // epilogue: D layout col = lane & 31, row = (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)
    if (nsplit > 1) {
        // (a release fence here instead of write-through stores flushes every dirty line of the XCD's L2, i.e. the
        // outputs of all concurrently running kernels: measured slower than the separate reduction launches it replaced)
        float* base = e.partial + (size_t)ctile * nsplit * BM * BN;
        partial_write<BM, BN, WM, WN>(base + (size_t)z * BM * BN, acc);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0)
            ws[0] = __hip_atomic_fetch_add(&e.counters[ctile], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        if (ws[0] != nsplit - 1) return;                       // uniform: not the last arriver of this tile
        // every wave acquires for itself (see streamk_publish)
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
        if (threadIdx.x == 0)
            __hip_atomic_store(&e.counters[ctile], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // ready for the next launch
        __syncthreads();
#pragma unroll
        for (int i = 0; i < TM; ++i)
#pragma unroll
            for (int j = 0; j < TN; ++j)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;
        for (int zz = 0; zz < nsplit; ++zz) partial_add<BM, BN, WM, WN>(base + (size_t)zz * BM * BN, acc);
    }
    if (e.io == 0) igemm_epilogue<BM, BN, WM, WN, 0>(e, P, acc, m0, n0, wm, wn, li, lh);
    else if (e.io == 1) igemm_epilogue<BM, BN, WM, WN, 1>(e, P, acc, m0, n0, wm, wn, li, lh);
    else igemm_epilogue<BM, BN, WM, WN, 2>(e, P, acc, m0, n0, wm, wn, li, lh);
}

typedef float f32x4 __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read128(f32x4& d, unsigned addr) {
    asm volatile("ds_read_b128 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
// bf16 math mode: operands are rounded (RNE) to bf16 on the way from LDS to the matrix core, accumulation stays
// fp32 (v_mfma_f32_32x32x8_bf16_1k: lane (i, h) holds k = 4h..4h+3 -- the same 4 consecutive k a ds_read_b128 of an
// fp32 tile row delivers, so the fp32 LDS layouts are used unchanged); activations / weights stay fp32 in HBM
typedef short s16x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x2 __attribute__((ext_vector_type(2)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
__device__ __forceinline__ unsigned cvt_pk_bf16(float lo, float hi) {    // one v_cvt_pk_bf16_f32 (RNE)
    const f32x2 v = {lo, hi};
    return __builtin_bit_cast(unsigned, __builtin_convertvector(v, bf16x2));
}
__device__ __forceinline__ s16x4 cvt_bf16x4(float a, float b, float c, float d) {
    const u32x2 r = {cvt_pk_bf16(a, b), cvt_pk_bf16(c, d)};
    return __builtin_bit_cast(s16x4, r);
}

template <int OFF>
__device__ __forceinline__ void lds_read32(float& d, unsigned addr) {
    asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_u16(unsigned& d, unsigned addr) {
    asm volatile("ds_read_u16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int OFF>
__device__ __forceinline__ void lds_read_i16(int& d, unsigned addr) {
    asm volatile("ds_read_i16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void vm_wait() {
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
typedef short s16x4v_ __attribute__((ext_vector_type(4)));
template <int OFF>
__device__ __forceinline__ void lds_read_tr16(s16x4v_& d, unsigned addr) {   // gfx950 transposing LDS read (see conv_wgradh)
    asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(d) : "v"(addr), "n"(OFF) : "memory");
}
template <int N>
__device__ __forceinline__ void lds_wait() {
    asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(N) : "memory");
}

// fp32 operand -> three bf16 planes, exact: x = hi + mid + lo where hi / mid / lo are the top 8 / next 8 / last 8 bits of
// the significand (truncation; each residual x - hi is exact in fp32).  Two 4-float fragments (8 k values of one lane)
// become three bf16x8 MFMA operands; v_perm_b32 packs the high halves of two dwords.
typedef unsigned u32x4_ __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void split3_bf16(const float (&a)[8], bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    typedef float f32x2_ __attribute__((ext_vector_type(2)));
    u32x4_ h, m, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {            // pairs: the two residual subtractions are one v_pk_add_f32 each
        const f32x2_ a2 = {a[2 * q], a[2 * q + 1]};
        const unsigned ua0 = __float_as_uint(a2.x), ua1 = __float_as_uint(a2.y);
        const f32x2_ h2 = {__uint_as_float(ua0 & 0xFFFF0000u), __uint_as_float(ua1 & 0xFFFF0000u)};
        const f32x2_ r2 = a2 - h2;
        const unsigned ur0 = __float_as_uint(r2.x), ur1 = __float_as_uint(r2.y);
        const f32x2_ m2 = {__uint_as_float(ur0 & 0xFFFF0000u), __uint_as_float(ur1 & 0xFFFF0000u)};
        const f32x2_ l2 = r2 - m2;
        h[q] = __builtin_amdgcn_perm(ua1, ua0, 0x07060302u);
        m[q] = __builtin_amdgcn_perm(ur1, ur0, 0x07060302u);
        l[q] = __builtin_amdgcn_perm(__float_as_uint(l2.y), __float_as_uint(l2.x), 0x07060302u);
    }
    hi = __builtin_bit_cast(bf16x8, h);
    mid = __builtin_bit_cast(bf16x8, m);
    lo = __builtin_bit_cast(bf16x8, l);
}
__device__ __forceinline__ void split3_bf16(const f32x4& f0, const f32x4& f1, bf16x8& hi, bf16x8& mid, bf16x8& lo) {
    const float a[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    split3_bf16(a, hi, mid, lo);
}
// acc += A x B with fp32-accurate products from the planes of A and B (6 of the 9 plane products).  Order: the planes
// that are ready first -- hi x hi needs one v_perm per pair, the mid / lo planes two / four more operations -- so that the
// rest of the split runs in the shadow of the first MFMAs (the accumulator is fp32: the order of these six terms moves
// the result by less than its last bit).
__device__ __forceinline__ void mfma_x3(f32x16& acc, const bf16x8& ah, const bf16x8& am, const bf16x8& al,
                                        const bf16x8& bh, const bf16x8& bm, const bf16x8& bl) {
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(am, bm, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(al, bh, acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bl, acc, 0, 0, 0);
}


// ---- fp16 hi / lo arithmetic (common.h "h2"): fp32-accurate products from TWO fp16 planes per operand, three
// v_mfma_f32_32x32x16_f16 per K = 16 step and accumulator pair instead of the six bf16 plane products above.
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
struct H2Scale {              // of one operand tensor: s = 2^e, s2 = 2^(e + 11), inv = 2^-e
    float s, s2, inv;
};
__device__ __forceinline__ H2Scale h2_scale_bits(unsigned amax_bits) {
    const int e = radet_h2_exp(amax_bits);
    H2Scale r;
    r.s = radet_pow2(e); r.s2 = radet_pow2(e + 11); r.inv = radet_pow2(-e);
    return r;
}
__device__ __forceinline__ H2Scale h2_scale(const unsigned* slot) {          // (all lanes active: kernel prologue)
    return h2_scale_bits(radet_amax_read(slot));
}
// the two halves of h2_scale for kernels that overlap the slot's (cold) load with their first tile loads: h2_scale_load at the
// top of the kernel, h2_scale_finish behind the prologue's wait (the empty volatile asm keeps the reduction from being
// scheduled in front of that wait)
__device__ __forceinline__ unsigned h2_scale_load(const unsigned* slot) { return radet_amax_load(slot); }
__device__ __forceinline__ H2Scale h2_scale_finish(unsigned raw) {
    asm volatile("" : "+v"(raw));
    return h2_scale_bits(radet_amax_reduce(raw));
}
// 8 k values of one lane -> the hi and lo operands of v_mfma_f32_32x32x16_f16 (24 VALU operations; split3_bf16: 44)
__device__ __forceinline__ void split2_f16(const float (&a)[8], const H2Scale& sc, f16x8& hi, f16x8& lo) {
    u32x4_ h, l;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        unsigned hq, lq;
        radet_split2(a[2 * q], a[2 * q + 1], sc.s, sc.s2, hq, lq);
        h[q] = hq; l[q] = lq;
    }
    hi = __builtin_bit_cast(f16x8, h);
    lo = __builtin_bit_cast(f16x8, l);
}
__device__ __forceinline__ void split2_f16(const f32x4& f0, const f32x4& f1, const H2Scale& sc, f16x8& hi, f16x8& lo) {
    const float a[8] = {f0.x, f0.y, f0.z, f0.w, f1.x, f1.y, f1.z, f1.w};
    split2_f16(a, sc, hi, lo);
}
// acc0 += hi hi', acc1 += hi lo' + lo hi' (the scaled lo planes carry a factor 2^11: see h2_combine)
__device__ __forceinline__ void mfma_h2(f32x16& acc0, f32x16& acc1, const f16x8& ah, const f16x8& al, const f16x8& bh,
                                        const f16x8& bl) {
    acc0 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bh, acc0, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah, bl, acc1, 0, 0, 0);
    acc1 = __builtin_amdgcn_mfma_f32_32x32x16_f16(al, bh, acc1, 0, 0, 0);
}
// scale of the fp16 pair copy of the output (ConvPtrs::yq), from quantities that are complete when the launch starts; call with
// all lanes active.  `store`: this workgroup publishes the bound (one per problem).
struct PairScaleRaw { unsigned xt, wl1, bs, as; };
__device__ __forceinline__ PairScaleRaw igemm_pair_scale_load(const ConvPtrs& P) {           // (P.yq != nullptr)
    PairScaleRaw r;
    r.xt = radet_amax_load(P.xt); r.wl1 = radet_amax_load(P.wl1);
    r.bs = P.bs ? radet_amax_load(P.bs) : 0u;
    r.as = P.as ? radet_amax_load(P.as) : 0u;
    return r;
}
__device__ __forceinline__ void igemm_pair_scale(const ConvPtrs& P, PairScaleRaw r, bool store, float& qs, float& qs2) {
    asm volatile("" : "+v"(r.xt), "+v"(r.wl1), "+v"(r.bs), "+v"(r.as));
    float bound = __uint_as_float(radet_amax_reduce(r.xt)) * __uint_as_float(radet_amax_reduce(r.wl1));
    bound += __uint_as_float(radet_amax_reduce(r.bs));           // (0 when the launch has no bias / addend)
    bound += __uint_as_float(radet_amax_reduce(r.as));
    bound *= 1.0001f;                                            // (fp32 rounding of the accumulation and of this sum)
    const unsigned bits = __float_as_uint(bound);
    if (store && threadIdx.x == 0) radet_amax_store(P.yqs, bits);
    const int e = radet_h2_exp(bits);
    qs = radet_pow2(e);
    qs2 = radet_pow2(e + 11);
}

// the product in the operands' own units: (acc0 + 2^-11 acc1) 2^-ea 2^-eb (two multiplications: either factor alone is a
// normal float, their product need not be)
__device__ __forceinline__ void h2_combine(f32x16& acc0, const f32x16& acc1, float inv_a, float inv_b) {
#pragma unroll
    for (int r = 0; r < 16; ++r) acc0[r] = (fmaf(acc1[r], 1.0f / 2048.0f, acc0[r]) * inv_a) * inv_b;
}

// ---- launchers of the fp16 hi / lo instantiations (conv_h2.hip, conv_wgrad_h2.hip: their own translation units, compiled next to conv_igemm.hip)
bool radet_launch_igemm_h2(int choice, const ConvArgs& a, hipStream_t st, int tag, int bk, size_t ws_floats, int stages,
                           bool no_tail_split);
struct WgradArgs;
int radet_launch_wgrad_h2(const WgradArgs& a, int flags, int bm, int bn, hipStream_t st);
