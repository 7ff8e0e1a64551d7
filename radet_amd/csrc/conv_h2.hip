// fp16 hi / lo arithmetic (common.h "h2", round 5): the instantiations of the implicit-GEMM / weight-gradient templates that
// form fp32-accurate products from TWO fp16 planes per operand -- three v_mfma_f32_32x32x16_f16 per K = 16 step into an
// accumulator pair, instead of the six bf16 plane products of the round-2 scheme.  A translation unit of its own so that it
// compiles next to conv_igemm.hip (the weight-gradient kernels of this arithmetic: conv_wgrad_h2.hip).  gfx950 only.
// Replaces cuDNN behind radet/models/backbones/resnet.py:260-299, necks/fpn.py:170-221, dense_heads/atss_head.py:118-145.
#include "common.h"
#include "../../include/radet_hip.h"
#include <stdlib.h>
#include <type_traits>

#include "conv_igemm_kernel.h"

// ------------------------------------------------------------------------------------------ implicit GEMM
// tag = 64 | 8 (fp32 operands split in registers, K step 32) [| 32: K-divided 64 x 64 tiles], or 64 | 16 (operands arrive
// as fp16 plane pairs, K step 32 channels); bit 0: profiling symbol.
template <int BM, int BN, int WM, int WN, int KIND>      // KIND 0: in-register split, 1: K-divided, 2: plane pairs
static void launch_h2(const ConvArgs& a_in, hipStream_t st, int tag, int bk, size_t ws_floats, int stages, bool no_tail_split) {
    ConvArgs a = a_in;
    const int tiles = igemm_plan<BM, BN>(a, tag, bk, ws_floats, 0, no_tail_split);
    constexpr int NT = WM * WN * 64;
#define RADET_H2(TAGV, BKV, NSV) \
    hipLaunchKernelGGL((conv_igemmg_kernel<BM, BN, WM, WN, TAGV, BKV, NSV>), dim3(tiles, a.sk), dim3(NT), 0, st, a)
    if constexpr (KIND == 0) {
        if (stages >= 3) { if (tag & 1) RADET_H2(73, 32, 3); else RADET_H2(72, 32, 3); }
        else { if (tag & 1) RADET_H2(73, 32, 2); else RADET_H2(72, 32, 2); }
    } else if constexpr (KIND == 1) {
        // (untagged symbols only: the K-divided tiles serve the backbone / neck, the profiling tag marks the head towers)
        if (tag & 128) RADET_H2(232, 64, 2);                         // ... on fp16 plane pairs (round 6): 64-channel stages only
        else if (bk == 64) RADET_H2(104, 64, 2);                     // 32 KiB per stage: two stages, two workgroups per CU
        else if (stages >= 4) RADET_H2(104, 32, 4);                  // 16 KiB per stage: loads up to three stages ahead
        else if (stages == 3) RADET_H2(104, 32, 3);
        else RADET_H2(104, 32, 2);
    } else {
        if constexpr (WM * WN == 8) {                                // row-interleaved pairs (TAG bit 7): the 8-wave tiles
            if (tag & 128) { if (tag & 1) RADET_H2(209, 32, 2); else RADET_H2(208, 32, 2); return; }
        }
        constexpr int STG = 2 * (BM + BN) * 16 * 4;                  // LDS bytes per stage: two planes x 64 bytes per tile row
        if constexpr (3 * STG <= 160 * 1024) {
            if (stages >= 3) { if (tag & 1) RADET_H2(81, 16, 3); else RADET_H2(80, 16, 3); return; }
        }
        if (tag & 1) RADET_H2(81, 16, 2); else RADET_H2(80, 16, 2);
    }
#undef RADET_H2
}

bool radet_launch_igemm_h2(int choice, const ConvArgs& a, hipStream_t st, int tag, int bk, size_t ws_floats, int stages,
                           bool no_tail_split) {
    if (tag & 16) {                                                  // plane pairs: the 8-wave tiles (+ the 4-wave tiles 1-3)
        switch (choice) {
            case 1: launch_h2<128, 128, 2, 2, 2>(a, st, tag & ~1, bk, ws_floats, stages, no_tail_split); return true;
            case 2: launch_h2<128, 64, 2, 2, 2>(a, st, tag & ~1, bk, ws_floats, stages, no_tail_split); return true;
            case 3: launch_h2<64, 64, 2, 2, 2>(a, st, tag & ~1, bk, ws_floats, stages, no_tail_split); return true;
            case 5: launch_h2<128, 128, 2, 4, 2>(a, st, tag, bk, ws_floats, stages, no_tail_split); return true;
            case 6: launch_h2<256, 128, 4, 2, 2>(a, st, tag, bk, ws_floats, stages, no_tail_split); return true;
            default: return false;
        }
    }
    switch (choice) {
        case 1: launch_h2<128, 128, 2, 2, 0>(a, st, tag, bk, ws_floats, stages, no_tail_split); return true;
        case 2: launch_h2<128, 64, 2, 2, 0>(a, st, tag, bk, ws_floats, stages, no_tail_split); return true;
        case 3: launch_h2<64, 64, 2, 2, 0>(a, st, tag, bk, ws_floats, stages, no_tail_split); return true;
        case 4: launch_h2<128, 32, 4, 1, 0>(a, st, tag, bk, ws_floats, stages, no_tail_split); return true;
        case 7: case 8: launch_h2<64, 64, 2, 2, 1>(a, st, tag, bk, ws_floats, stages, no_tail_split); return true;
        default: return false;
    }
}

