// Plane-pair tower tile with 128 x 64 WAVE tiles (round 6 experiment, tile_override 9): 256 x 128 block tile, FOUR waves (one per
// SIMD), 8 accumulator pairs = 256 accumulator registers per wave -- compiled WITHOUT -amdgpu-mfma-vgpr-form so that the
// accumulators may live in AGPRs (a wave that is alone on its SIMD owns 512 registers).  Per 32-channel stage a workgroup reads
// 4 x (4 + 2) x 2 planes x 2 slices x 1 KiB = 96 KiB of fragments instead of the 8-wave tile's 128 KiB for the same 192 MFMAs.
// Replaces cuDNN behind radet/models/dense_heads/atss_head.py:118-145 (same template as conv_h2.hip: conv_igemm_kernel.h).
#include "common.h"
#include "../../include/radet_hip.h"
#include <stdlib.h>
#include <type_traits>

#include "conv_igemm_kernel.h"

bool radet_launch_igemm_h2_big(const ConvArgs& a_in, hipStream_t st, int tag, int bk, size_t ws_floats, bool no_tail_split) {
    ConvArgs a = a_in;
    const int tiles = igemm_plan<256, 128>(a, tag, bk, ws_floats, 0, no_tail_split);
    if (tag & 1) hipLaunchKernelGGL((conv_igemmg_kernel<256, 128, 2, 2, 81, 16, 2>), dim3(tiles, a.sk), dim3(256), 0, st, a);
    else hipLaunchKernelGGL((conv_igemmg_kernel<256, 128, 2, 2, 80, 16, 2>), dim3(tiles, a.sk), dim3(256), 0, st, a);
    return true;
}
