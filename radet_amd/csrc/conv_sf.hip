// fp32 implicit-GEMM convolution, split-at-fill tile (tile choice 9 of radet_conv2d_igemm): see the kernel comment.
#include "conv_common.h"

static __device__ __attribute__((aligned(16))) float radet_zero_page[512];

// ------------------------------------------------------------------------------------------ split-at-fill tile
// fp32 tensors, products from bf16 planes (as TAG bit 3), but every operand element is split ONCE PER WORKGROUP on its way
// into LDS instead of once per consuming wave on its way out: the tiles go global -> registers (two dwordx4 per 8 channels)
// -> split3 -> three ds_write_b128 (hi / mid / lo plane images [plane][row][32 channels] bf16, 16-byte slots XOR-swizzled by
// (row >> 2) & 3 so that both the writes and the fragment reads are bank-conflict free), and the K loop is the
// plane-operand loop: per K = 16 slice 3 (TM + TN) ds_read_b128 + 6 TM TN v_mfma_f32_32x32x16_bf16, no VALU on the
// fragments.  Against the 2 x 2-wave register-split tile of the same size: half the vector-ALU work per MFMA (each A row
// used to be split by both waves of its tile row, each B row by both of its column) and half the LDS fill per MFMA of the
// 64 x 64 K-divided tiles.  One LDS stage (48 KiB for 128 x 128) + the next stage's raw fp32 tile in 32 registers: two
// workgroups per CU, the second one's MFMAs cover this one's barrier / write phase; the split of the prefetched tile is
// interleaved with the MFMAs of the second k slice.  Same epilogue / split-K / tail-split / class-launch machinery.
template <int BM, int BN, int TAG>
__global__ __launch_bounds__(256, 2) void conv_igemm_sf_kernel(const ConvArgs a) {
    constexpr int BK = 32, WM = 2, WN = 2;
    constexpr int TM = BM / 64, TN = BN / 64;
    constexpr int AU = BM * 4 / 256, BU = BN * 4 / 256;       // (row, 8-channel octet) units per thread and stage
    __shared__ __attribute__((aligned(16))) unsigned char As[3 * BM * 64];
    __shared__ __attribute__((aligned(16))) unsigned char Bs[3 * BN * 64];

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int li = lane & 31, lh = lane >> 5;
    const int tilesN = (a.Cout + BN - 1) / BN;
    const int tilesG = ((a.M + BM - 1) / BM) * tilesN;
    const int KT = a.KH * a.KW;
    const int cpt = a.Cin / BK;
    const bool tail = (int)blockIdx.x >= a.n_full;
    const int tail_slot = tail ? (int)blockIdx.x - a.n_full : 0;
    int id = tail ? a.n_full + tail_slot / a.sk_tail : (a.cls_nt ? (int)blockIdx.x : xcd_remap(blockIdx.x, a.n_full));
    const int sk_tile = id;
    const int grp = id >= tilesG ? 1 : 0;
    id -= grp * tilesG;
    ConvPtrs P = a.p[grp];
    P.y = pin_sgpr(P.y); P.bias = pin_sgpr(P.bias); P.addend = pin_sgpr(P.addend); P.mask = pin_sgpr(P.mask);
    EpiArgs epi;
    epi.M = pin_sgpr(a.M); epi.Cout = pin_sgpr(a.Cout); epi.relu = pin_sgpr(a.relu); epi.io = 0;
    epi.out_rows = pin_sgpr(a.out_rows);
    epi.partial = pin_sgpr(a.partial); epi.counters = pin_sgpr(a.counters);
    epi.sk_base = 0; epi.sk_rem = 0;
    const int nsplit = pin_sgpr(tail ? a.sk_tail : a.sk);
    const int ctile = pin_sgpr(tail ? tail_slot / a.sk_tail : sk_tile);
    const int zsplit = pin_sgpr(tail ? tail_slot % a.sk_tail : (int)blockIdx.y);
    const int m0 = (id / tilesN) * BM;
    const int n0 = (id % tilesN) * BN;
    int KTt = KT, tbase = 0;
    if (a.cls_nt) {
        const int cls = (m0 >= a.cls_b[0] ? 1 : 0) + (m0 >= a.cls_b[1] ? 1 : 0) + (m0 >= a.cls_b[2] ? 1 : 0);
        KTt = (a.cls_nt >> (4 * cls)) & 15;
        tbase = 4 * cls;
    }
    const int per = tail ? a.it_per_tail : a.it_per_split;
    const int it0 = zsplit * per;
    int nK = KTt * cpt - it0;
    if (nK > per) nK = per;

    // loader: this thread owns octet `oct` (8 channels) of tile rows r0 + 64 k
    const int r0 = tid >> 2, oct = tid & 3;
    int ld_tap = it0 / cpt, ld_c0 = (it0 - ld_tap * cpt) * BK;
    int arow[AU];
    const float* wrow[BU];
#pragma unroll
    for (int k = 0; k < AU; ++k) arow[k] = nK > 0 ? a.rowtab[(size_t)ld_tap * a.Mp + m0 + r0 + 64 * k] : -1;
#pragma unroll
    for (int k = 0; k < BU; ++k) {
        const int n = n0 + r0 + 64 * k;
        wrow[k] = n < a.Cout ? P.w + (size_t)n * a.KTw * a.Cin + 8 * oct : nullptr;
    }
    int wtap = nK > 0 ? a.tap_ids[tbase + ld_tap] : 0;
    const float* zero = radet_zero_page + lane * 8;
    f32x4 raw[2 * (AU + BU)];
    auto pick = [&](const float* real, bool ok) {          // address select instead of a conditional load (no exec juggling)
        const unsigned long long m = ok ? ~0ull : 0ull;
        return reinterpret_cast<const f32x4*>(((unsigned long long)real & m) | ((unsigned long long)zero & ~m));
    };
    auto issue = [&]() {                        // the stage at the cursor -> raw registers; then move the cursor
#pragma unroll
        for (int k = 0; k < AU; ++k) {
            const f32x4* src = pick(P.x + (size_t)(arow[k] < 0 ? 0 : arow[k]) * a.Cin + ld_c0 + 8 * oct, arow[k] >= 0);
            raw[2 * k] = src[0];
            raw[2 * k + 1] = src[1];
        }
#pragma unroll
        for (int k = 0; k < BU; ++k) {
            const f32x4* src = pick(wrow[k] + (size_t)wtap * a.Cin + ld_c0, wrow[k] != nullptr);
            raw[2 * (AU + k)] = src[0];
            raw[2 * (AU + k) + 1] = src[1];
        }
        ld_c0 += BK;
        if (ld_c0 == a.Cin) {
            ld_c0 = 0;
            ++ld_tap;
            if (ld_tap < KTt) {
                wtap = a.tap_ids[tbase + ld_tap];
#pragma unroll
                for (int k = 0; k < AU; ++k) arow[k] = a.rowtab[(size_t)ld_tap * a.Mp + m0 + r0 + 64 * k];
            }
        }
    };
    bf16x8 pl[AU + BU][3];
    auto split_unit = [&](int u) { split3_bf16(raw[2 * u], raw[2 * u + 1], pl[u][0], pl[u][1], pl[u][2]); };
    const unsigned wslot = (unsigned)(r0 * 64 + ((oct ^ ((r0 >> 2) & 3)) * 16));      // (row + 64 k keeps (row >> 2) & 3)
    auto write_planes = [&]() {
#pragma unroll
        for (int k = 0; k < AU; ++k)
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(As + p * BM * 64 + k * 64 * 64 + wslot) = pl[k][p];
#pragma unroll
        for (int k = 0; k < BU; ++k)
#pragma unroll
            for (int p = 0; p < 3; ++p) *reinterpret_cast<bf16x8*>(Bs + p * BN * 64 + k * 64 * 64 + wslot) = pl[AU + k][p];
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int i = 0; i < TM; ++i)
#pragma unroll
        for (int j = 0; j < TN; ++j)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[i][j][r] = 0.f;

    if (nK > 0) {
        issue();
#pragma unroll
        for (int u = 0; u < AU + BU; ++u) split_unit(u);
        write_planes();
        if (nK > 1) issue();
    }
    __syncthreads();

    // reader: rows (wm TM + i) 32 + li of A, (wn TN + j) 32 + li of B; k slice s = octets 2 s + lh
    const unsigned rsw = (unsigned)((li >> 2) & 3);
    const unsigned abase = (unsigned)((wm * TM * 32 + li) * 64), bbase = (unsigned)((wn * TN * 32 + li) * 64);
    // Software pipeline of iteration `it` (planes of stage it in LDS, raw tile of stage it + 1 in flight since the middle
    // of iteration it - 1): fragment reads of both k slices; MFMAs of slice 0; MFMAs of slice 1 with the split of the raw
    // tile spread between them (a wave issues in order: a VALU block in front of or behind the MFMAs idles the matrix pipe
    // for its 700 cycles); loads of stage it + 2 into the freed raw registers; barrier; planes of stage it + 1 -> LDS;
    // barrier.  The loads have the barrier / write phase and a whole slice of MFMAs to arrive.
    const int dbg = a.dbg;
    bf16x8 fa[2][3][TM], fb[2][3][TN];
    for (int it = 0; it < nK; ++it) {
        if (!(dbg & 4) || it == 0)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            const unsigned so = ((unsigned)(2 * s + lh) ^ rsw) * 16;
#pragma unroll
            for (int p = 0; p < 3; ++p) {
#pragma unroll
                for (int i = 0; i < TM; ++i) fa[s][p][i] = *reinterpret_cast<const bf16x8*>(As + p * BM * 64 + abase + i * 32 * 64 + so);
#pragma unroll
                for (int j = 0; j < TN; ++j) fb[s][p][j] = *reinterpret_cast<const bf16x8*>(Bs + p * BN * 64 + bbase + j * 32 * 64 + so);
            }
        }
#pragma unroll
        for (int s = 0; s < 2; ++s) {
#pragma unroll
            for (int t = 0; t < 6; ++t) {           // terms: hi hi, mid hi, hi mid, mid mid, lo hi, hi lo
                const int pa = (t == 1 || t == 3) ? 1 : (t == 4 ? 2 : 0), pb = (t == 2 || t == 3) ? 1 : (t == 5 ? 2 : 0);
#pragma unroll
                for (int i = 0; i < TM; ++i)
#pragma unroll
                    for (int j = 0; j < TN; ++j)
                        if (!(dbg & 2)) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fa[s][pa][i], fb[s][pb][j], acc[i][j], 0, 0, 0);
            }
            if (s == 0) {
                // slice 0: its MFMAs carry the second slice's fragment reads between them
#pragma unroll
                for (int q = 0; q < 6 * TM * TN; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
                if (!(dbg & 1))
#pragma unroll
                for (int u = 0; u < AU + BU; ++u) split_unit(u);       // (of a stale tile on the last iteration: harmless)
            } else {
#pragma unroll
                for (int q = 0; q < 6 * TM * TN; ++q) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 1);
                    __builtin_amdgcn_sched_group_barrier(0x002, (44 * (AU + BU) + 6 * TM * TN - 1) / (6 * TM * TN), 1);
                }
                // (the planes are only consumed behind the barrier: left alone the compiler sinks the whole split there)
#pragma unroll
                for (int u = 0; u < AU + BU; ++u)
#pragma unroll
                    for (int p = 0; p < 3; ++p) asm volatile("" : "+v"(pl[u][p]));
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        const bool more = it + 1 < nK;
        if (it + 2 < nK && !(dbg & 1)) issue();
        __syncthreads();                           // every wave is done with this stage's planes
        if (more && !(dbg & 1)) write_planes();
        __syncthreads();
    }
    igemm_store<BM, BN, WM, WN>(epi, P, acc, m0, n0, nsplit, ctile, zsplit, wm, wn, li, lh, reinterpret_cast<volatile int*>(As));
}

void radet_launch_igemm_sf(const ConvArgs& a, int tiles, int tagged, hipStream_t st) {
    if (tagged) hipLaunchKernelGGL((conv_igemm_sf_kernel<128, 128, 1>), dim3(tiles, a.sk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_igemm_sf_kernel<128, 128, 0>), dim3(tiles, a.sk), dim3(256), 0, st, a);
}
