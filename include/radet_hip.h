/* radet_hip.h -- C ABI of libradet_hip.so, the MI355X (gfx950) implementation of RADet's detector hot path.
 *
 * Every entry point: plain pointers (DEVICE memory unless stated "host"), sizes, and a `void* stream`
 * (hipStream_t; NULL = default stream).  Stream-ordered, no allocation, no host synchronisation
 * inside; workspaces are supplied by the caller.  Return value: 0 = ok, -1 = bad argument, -2 = launch
 * failure.  Activations are NHWC fp32, several pyramid levels concatenated row-wise ("multi-level
 * buffer"): level l holds B images of Ho_l x Wo_l pixels starting at row out_row_off_l.
 *
 * seg_desc (host): nseg x 6 ints {Hi, Wi, Ho, Wo, in_row_off, out_row_off} per level.
 *
 * Each function names the reference interface it replaces (paths relative to /root/reference).
 */
#ifndef RADET_HIP_H
#define RADET_HIP_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- convolution stack: replaces torch.nn.Conv2d/BatchNorm2d(eval)/ReLU/residual-add behind
 *      radet/models/backbones/resnet.py:260-299,622-637, necks/fpn.py:170-221,
 *      dense_heads/atss_head.py:118-145 (cuDNN in the reference) ------------------------------- */

/* One convolution's parameter bundle for the batched fold / unfold kernels (device pointers). */
typedef struct RadetConvDesc {
    const float* w;        /* [Cout][Cin][KH][KW]  parameter (OIHW, state-dict layout) */
    const float* bias;     /* [Cout] conv bias or NULL */
    const float* bn_gamma; /* [Cout] or NULL (no BN) */
    const float* bn_beta;
    const float* bn_mean;
    const float* bn_var;
    float* wf;             /* [Cout][KH*KW][Cin]   folded weights (OHWI) for forward */
    float* wft;            /* [Cin][KH*KW][Cout]   transposed copy for dgrad, or NULL */
    float* bias_f;         /* [Cout] folded bias (BN shift or conv bias or 0) */
    const float* dwf_slabs;      /* [nsplit][Cout][KH*KW][Cin] wgrad partial slabs */
    const float* dbias_partials; /* [nsplit][Cout] or NULL */
    float* dw;             /* [Cout][Cin][KH][KW] gradient (OIHW) or NULL (frozen) */
    float* dbias;          /* [Cout] or NULL */
    float* dgamma;         /* [Cout] or NULL */
    float* dbeta;          /* [Cout] or NULL */
    int cout, cin, kh, kw;
    int nsplit;
    float eps;
    int wft_ld;            /* row stride of wft's last dim (>= Cout; zero-padded K for small heads), 0 = Cout */
    int wft_off;           /* column offset inside that padded row */
    int w16;               /* 1: wf / wft are bf16 buffers (bf16-storage mode); 2: bf16 plane triples (rows [3][Cin] resp.
                              [3][wft_ld], see "planes" below); 3: fp16 plane pairs (rows [2][Cin] resp. [2][wft_ld], see
                              "plane pairs" below; needs w_amax); bias_f stays fp32 */
    float* w_amax;         /* 64 words or NULL: amax slot (see RadetScales) of the folded weights, largest |wf|; zeroed and raised
                              by radet_fold_weights -- the weight operand's scale in the fp16 hi / lo arithmetic */
    float* wfq;            /* optional: a second copy of wf as fp16 plane pairs (rows [2][Cin], scaled by w_amax's power of two),
                              the weight operand of forward launches whose x arrives as plane pairs; or NULL */
    float* w_l1;           /* amax-style slot or NULL: largest L1 norm of a folded output channel, max_o sum |wf[o][.][.]| --
                              |conv output| <= amax(x) * it (+ |bias| + |addend|): the bound RadetScales.yq is scaled with */
    float* bias_amax;      /* amax-style slot or NULL: largest |bias_f| */
    float* w_l1t;          /* amax-style slot or NULL (round 6): largest L1 norm of a folded INPUT channel, max_c sum_{o,t} |wf[o][t][c]|
                              -- |dgrad output| <= amax(dy) * it: the bound a dgrad launch scales its pair output with */
} RadetConvDesc;

/* ---- amax slots (round 5, "fp16 hi / lo arithmetic").  The default fp32 conv arithmetic forms fp32-accurate products
 * from TWO fp16 numbers per operand element: t = x 2^e, hi = fp16(t), lo = fp16((t - hi) 2^11), and
 * x x' = 2^-(e + e') (hi hi' + 2^-11 (hi lo' + lo hi')) -- three v_mfma_f32_32x32x16_f16 per K = 16 step instead of the six
 * bf16 plane products of 0x1000000 (the dropped lo lo' term is <= 2^-24 relative; x = 2^-e (hi + 2^-11 lo) to 2^-23
 * relative for every element within 2^-27 of the tensor's largest).  fp16 has 5 exponent bits, so every operand tensor is
 * scaled by an exact power of two 2^e that puts its largest magnitude into [2^14, 2^15).  e comes from the tensor's "amax
 * slot": RADET_AMAX_WORDS = 64 32-bit words, 128 bytes apart (radet_amax_slot_words() = 2048 words = 8 KiB per slot, the
 * rest unused), of device memory whose LARGEST word is the bit pattern of a
 * non-negative float that is >= every |element| (the largest magnitude, or any bound on it), e = 141 - biased_exponent(slot)
 * (0 for a zero slot).  Slots are maintained by the kernels that WRITE a tensor: they raise one of the 64 words (chosen by
 * workgroup and wave: thousands of waves raising ONE address serialise) with atomicMax on the bit pattern -- order
 * independent, hence deterministic -- in conv epilogues (RadetScales.y_amax), the _a variants of the elementwise kernels and
 * radet_absmax; or they store a bound they can derive before writing into word 0 (GroupNorm: radet_gn_relu_fwd_q / _bwd_q,
 * radet_split_pairs; the other words must be zero).  The caller zeroes a slot before the first kernel of a step that
 * raises it.  Host struct of device pointers (each to a 64-word slot): */
#define RADET_AMAX_WORDS 64
typedef struct RadetScales {
    const void* x_amax;    /* slot of the GEMM's x operand (dgrad: dy; wgrad: dy) */
    const void* w_amax;    /* slot of the w operand (RadetConvDesc.w_amax; wgrad: the slot of x) */
    void* y_amax;          /* slot raised to the largest |y| stored, or NULL (works with every arithmetic) */
    const void* x1_amax;   /* the same for the second problem of a pair launch */
    const void* w1_amax;
    void* y1_amax;
    /* optional pair copy of the output (first problem only; fp32 y, Cout % 32 == 0): yq = y once more as fp16 plane pairs
     * (rows [2][Cout]), scaled by the power of two of the bound  amax(x) * L1max(w) + max|bias| + amax(addend)  -- complete
     * before the launch starts, unlike y's own largest magnitude -- which is stored to yq_amax.  x_true_amax: the slot x's
     * producer RAISED (for a plane-pair x: not the bound its pairs were scaled with, or the bounds of consecutive layers would
     * multiply); w_l1: RadetConvDesc.w_l1; bias_amax / addend_amax: NULL when the launch has no bias / addend.  The next
     * conv reads yq as its x operand (+0x2000000 | 0x8000000) and needs no operand split in its K loop. */
    void* yq;
    void* yq_amax;
    const void* x_true_amax;
    const void* w_l1;
    const void* bias_amax;
    const void* addend_amax;
} RadetScales;

/* Gather table of one conv geometry: table[tap][Mp] = input row feeding (output row m, tap) or -1 (padding /
 * stride hole); Mp = radet_gather_table_rows(M).  Forward / wgrad: (so, sr, off, div) = (stride, 1, -pad, 1) with
 * seg_desc rows = conv outputs.  dgrad: (1, -1, pad, stride) with input/output roles of seg_desc swapped.
 * Built once per (B, H, W); removes every integer division from the GEMM main loops. */
int radet_gather_table_rows(int M);
int radet_build_gather_table(int* table, int B, int KH, int KW, int so, int sr, int off, int div, const int* seg_desc,
                             int nseg, void* stream);
/* Implicit-GEMM conv on MFMA: y[m,n] = sum_{tap,c} x[table[tap][m], c] * w[n][tap][c].  Forward: w = wf.
 * dgrad: x = dy, w = wft (Cin/Cout swapped, dgrad table).
 * Epilogue: y = acc + bias[n] (+ addend[m,n]) ; relu ; then y = mask[m,n] > 0 ? y : 0.
 * tile_override: 0 = heuristic, 1..4 = tile config; +0x100 = tagged kernel symbol (profiling); +0x200 = K step 32;
 * +0x800 = bf16 storage (x, w, addend, mask are bf16 tensors, Cin % 32 == 0; y bf16, or fp32 with +0x10000);
 * +0x400 = bf16 math mode (operands rounded RNE to bf16 between LDS and the matrix core, fp32 accumulate, fp32
 * tensors in HBM -- the arithmetic of mmcv's fp16 wrapper, `apis/train.py:113-117`, in bf16); bits 12-15 force
 * a split-K factor; +0x20000 = 3 LDS stages (fp32, launches that run alone on the device); + (w << 20), w = 1..7 =
 * stream-K schedule with w persistent workgroups per CU (fp32, untagged symbol, tiles 2-4): the K stages of the whole
 * launch are shared evenly, tiles cut by a share boundary are reduced inside the launch in K order (deterministic);
 * plain launch when there is less than one K stage per workgroup; +0x1000000 = fp32 tensors, products formed on the
 * bf16 matrix cores from an exact three-way bf16 split of every fp32 operand (x = hi + mid + lo, 8 significand bits
 * each; 6 of the 9 plane products -- everything above 2^-24 relative -- through v_mfma_f32_32x32x16_bf16, fp32
 * accumulate; Cin % 32 == 0, otherwise the native fp32 MFMA is used), with two more tiles: 7 / 8 = 64 x 64 tiles whose four
 * waves divide the K step (four 16-channel k-groups of a 64-channel stage, Cin % 64 == 0 / two k-groups x two column halves of
 * a 32-channel stage) and add their partial tiles through LDS -- every operand element is split once per workgroup
 * (-1 for these tiles without 0x1000000, with stream-K bits, or when Cin does not divide); +0x2000000 = the same arithmetic with operands
 * that ARRIVE as bf16 plane triples (x rows [3][Cin] bf16, w [Cout][taps][3][Cin] bf16 -- "planes" below; y, addend, mask,
 * bias fp32; Cin % 32 == 0): no operand split in the K loop; tiles 1..4 as above plus 5 = 128 x 128 and 6 = 256 x 128 with
 * 8 waves; K step 32 channels, or 16 with +0x4000000; +0x20000 = one more LDS stage.  splitk_ws (may be NULL): workspace of splitk_ws_floats floats whose first 16384 words are arrival
 * tickets that must be ZERO before the first launch (every launch leaves them zero); when given, launches with too few
 * tiles for 256 CUs split the K loop (<= 8 ways, or only the left-over tiles of the last round) and the workgroup that
 * arrives last at a tile sums the partial tiles in split order and applies the epilogue -- one launch, deterministic.
 * One workspace per stream: concurrent launches must not share it. */
int radet_conv2d_igemm(const float* x, const float* w, const float* bias, const float* addend, const float* mask,
                       float* y, const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                       int tile_override, float* splitk_ws, size_t splitk_ws_floats, void* stream);
/* Two independent convolutions of identical geometry (cls / reg tower layers of the shared head) in ONE launch */
/* ... with amax slots: tile_override +0x8000000 (together with 0x1000000: fp32 tensors split in registers, tiles 1-4, 7, 8;
 * or with 0x2000000: operands arrive as fp16 plane pairs, tiles 5 / 6) selects the fp16 hi / lo arithmetic described at
 * RadetScales; x_amax and w_amax are required then.  y_amax alone may be used with any arithmetic. */
int radet_conv2d_igemm_s(const float* x, const float* w, const float* bias, const float* addend, const float* mask,
                         float* y, const int* gather_table, int M, int Cin, int Cout, int KH, int KW, int relu,
                         int tile_override, float* splitk_ws, size_t splitk_ws_floats, void* stream, const RadetScales* sc);
int radet_conv2d_igemm_pair_s(const float* x0, const float* w0, const float* bias0, const float* addend0,
                              const float* mask0, float* y0, const float* x1, const float* w1, const float* bias1,
                              const float* addend1, const float* mask1, float* y1, const int* gather_table, int M,
                              int Cin, int Cout, int KH, int KW, int relu, int tile_override, float* splitk_ws,
                              size_t splitk_ws_floats, void* stream, const RadetScales* sc);
int radet_conv2d_igemm_taps_s(const float* x, const float* w, const float* addend, const float* mask, float* y,
                              const int* gather_table, const int* out_rows, const int* tap_ids_host, int ntaps, int kt_w,
                              int M, int Cin, int Cout, int tile_override, float* splitk_ws, size_t splitk_ws_floats,
                              void* stream, const RadetScales* sc);
int radet_conv2d_igemm_classes_s(const float* x, const float* w, const float* addend, const float* mask, float* y,
                                 const int* gather_table, const int* out_rows, const int* tap_ids_host,
                                 const int* cls_ntaps, const int* cls_start, int ncls, int kt_w, int M, int Cin, int Cout,
                                 int tile_override, float* splitk_ws, size_t splitk_ws_floats, void* stream,
                                 const RadetScales* sc);
int radet_conv2d_igemm_pair(const float* x0, const float* w0, const float* bias0, const float* addend0,
                            const float* mask0, float* y0, const float* x1, const float* w1, const float* bias1,
                            const float* addend1, const float* mask1, float* y1, const int* gather_table, int M, int Cin,
                            int Cout, int KH, int KW, int relu, int tile_override, float* splitk_ws,
                            size_t splitk_ws_floats, void* stream);
/* Tap-subset variant for the dgrad of strided convs: GEMM row m writes output row out_rows[m]; only `ntaps` taps
 * (tap_ids_host[t] = tap index inside the weight's kt_w taps) contribute; gather_table is [ntaps][Mp]. One launch
 * per stride-parity class performs only the non-zero multiply-adds (no bias / relu; addend + mask supported). */
int radet_conv2d_igemm_taps(const float* x, const float* w, const float* addend, const float* mask, float* y,
                            const int* gather_table, const int* out_rows, const int* tap_ids_host, int ntaps, int kt_w,
                            int M, int Cin, int Cout, int tile_override, float* splitk_ws, size_t splitk_ws_floats,
                            void* stream);
/* Class variant: ALL stride-parity classes of a strided dgrad in ONE launch (replaces one radet_conv2d_igemm_taps
 * launch per class; reference: the dgrad torch autograd runs for the stride-2 convs of mmdet ResNet / FPN extra levels).
 * GEMM rows = output rows sorted by class, each class padded to a multiple of 128 rows (out_rows = -1 and table = -1 on
 * the pad rows); class c owns rows [cls_start[c], cls_start[c+1]) (cls_start[0] = 0, the last class ends at M, M % 128
 * == 0), runs cls_ntaps[c] <= 4 taps, its tap t reads weight tap tap_ids_host[4c + t]; gather_table is
 * [max ntaps][radet_gather_table_rows(M)].  Order the classes by taps, most first (the grid keeps that order). */
int radet_conv2d_igemm_classes(const float* x, const float* w, const float* addend, const float* mask, float* y,
                               const int* gather_table, const int* out_rows, const int* tap_ids_host, const int* cls_ntaps,
                               const int* cls_start, int ncls, int kt_w, int M, int Cin, int Cout, int tile_override,
                               float* splitk_ws, size_t splitk_ws_floats, void* stream);
/* Predictor head convs (3x3, stride 1, pad 1, <= 32 output channels; reference: atss_cls / atss_reg / atss_iou of
 * radet/models/dense_heads/radet_head.py:_init_layers, applied per level in forward_single) as a direct convolution from an
 * LDS patch: x [rows][Cin] fp32 over all pyramid levels (rows of (level, image) contiguous, row-major H x W), w OHWI
 * [c][9][Cin], y [rows][c].  tiles_dev: ntiles x {base_row, H, W, (tile_y << 16) | tile_x} (int32), one 8 x 16 block of
 * output pixels each.  A second conv on the same input may share the launch (w1 / bias1 / y1 / c1; c0 + c1 <= 32).
 * Arithmetic: fp32 products from three bf16 planes per operand (as tile_override 0x1000000).  Cin % 16 == 0. */
int radet_pred3x3_patch(const float* x, int Cin, const int* tiles_dev, int ntiles, const float* w0, const float* bias0,
                        float* y0, int c0, const float* w1, const float* bias1, float* y1, int c1, void* stream);
/* wgrad: slabs[s][o][tap][c] = sum over pixel split s of dy[m,o] * x[table[tap][m],c];
 * optional dbias_partials[s][o] = column sums of dy.  S from radet_conv2d_wgrad_splits.
 * flags bit 0: bf16 math mode (as tile_override 0x400 of radet_conv2d_igemm); bit 8 (0x100): fp32 products from three bf16
 * planes per operand (as tile_override 0x1000000); bits 4-5: tile override of the one-tap
 * kernel (1 = 128x128, 2 = 64x64; S is then the caller's choice); bit 6: never use the all-taps kernel; bit 7: 32
 * instead of 16 pixels per LDS stage in the one-tap fp32 kernel; bits 10 / 11 (0x400 / 0x800, with 0x100 and the 64x64 tile):
 * the four waves divide a 64-pixel stage four ways / a 32-pixel stage two ways and share the operand splits;
 * bit 1: bf16 storage -- dy and x are bf16 tensors (ld_dy / Cin in elements, multiples of 8), slabs stay fp32;
 * bit 9 (0x200): dy and x are bf16 plane triples (dy rows [3][ld_dy], x rows [3][Cin]; 3x3 convs with Cin % 32 == 0). */
int radet_conv2d_wgrad_splits(int M, int Cin, int Cout, int KH, int KW);
int radet_conv2d_wgrad(const float* dy, const float* x, float* slabs, float* dbias_partials, const int* gather_table,
                       int M, int Cin, int Cout, int ld_dy, int KH, int KW, int S, int flags, void* stream);
/* ... with amax slots: flags +0x1000 = fp16 hi / lo arithmetic (see RadetScales; sc->x_amax = the slot of dy, sc->w_amax = the
 * slot of x, both required): fp32 tensors split in registers with the one-tap tiles (bits 4-5, 7, 10-11 as for 0x100), or,
 * with +0x200, dy rows [2][ld_dy] / x rows [2][Cin] fp16 plane pairs (3x3 convs, Cin % 32 == 0, ld_dy % 32 == 0:
 * conv_wgrad9q_kernel, 128 output x 32 input channels x 9 taps per workgroup).  +0x2000 (with +0x200, 3x3): the gather table
 * is that of a unit-stride conv with padding 1 -- tap (r, q) of pixel m reads what tap (r, 1) of pixel m + q - 1 reads, or
 * padding: row - pixel fits 16 bits -- and the launch runs conv_wgrad9d_kernel (five stage buffers, loads four stages ahead,
 * the table rows of a pixel split copied to LDS; splits of more than 1664 pixels fall back to conv_wgrad9q_kernel).  +0x4000
 * (same geometry, an experiment): the nine taps take shifted windows of three row segments in LDS instead of nine gathered
 * tiles (conv_wgrad9r_kernel).  All three form the same sums in the same order: bit-identical results. */
int radet_conv2d_wgrad_s(const float* dy, const float* x, float* slabs, float* dbias_partials, const int* gather_table,
                         int M, int Cin, int Cout, int ld_dy, int KH, int KW, int S, int flags, void* stream,
                         const RadetScales* sc);
/* Grouped wgrad: up to 32 independent weight-gradient GEMMs (one-tap kernel) in ONE launch -- the convs of a backbone
 * stage / of the neck, whose individual grids are too short to fill 256 CUs.  Each job = the arguments of
 * radet_conv2d_wgrad (host array; Cout and Cin multiples of the tile).  flags bit 0: bf16 math mode; bits 4-5 = 1:
 * 128x128 tiles (default 64x64). */
typedef struct RadetWgradJob {
    const float* dy;
    const float* x;
    float* slabs;
    float* dbias_partials;
    const int* gather_table;
    int M, Cin, Cout, ld_dy, KH, KW, S;
} RadetWgradJob;
int radet_conv2d_wgrad_group(const RadetWgradJob* jobs, int njobs, int flags, void* stream);
int radet_fold_weights(const RadetConvDesc* table_dev, int nconv, void* stream);
int radet_unfold_grads(const RadetConvDesc* table_dev, int nconv, int max_cout, void* stream);
/* stem: 7x7/2 conv (3->64) + folded BN + ReLU, NCHW image in, NHWC out (resnet.py:558-570,627-629) */
int radet_stem_conv_bn_relu(const float* img_nchw, const float* wf_ohwi, const float* bias, float* y_nhwc, int B,
                            int H, int W, void* stream);
int radet_maxpool3x3s2(const float* x, float* y, int B, int H, int W, int C, void* stream); /* resnet.py:570,630 */
/* Trainable stem (ResNet(frozen_stages=-1): conv1 / bn1 get gradients, resnet.py:572-588).  Backward of the max-pool fused with
 * the stem's ReLU: ds = [s > 0] * (dpool routed to the first maximum of each 3x3 / 2 window in row-major order -- the element
 * torch's MaxPool2d backward picks); s = stem output [B,H,W,C], dpool [B,Ho,Wo,C], fp32, C % 4 == 0. */
int radet_maxpool3x3s2_bwd_relu(const float* s, const float* dpool, float* ds, int B, int H, int W, int C, void* stream);
/* Weight gradient of the 7x7 / 2 stem conv from the NCHW image: slabs[S][64][49][3] (pixel splits, the layout radet_unfold_grads
 * reduces) and dbias_partials[S][64] (column sums of ds, may be NULL); ds = gradient w.r.t. the stem's pre-activation
 * [B*Ho*Wo, 64].  S from radet_stem_wgrad_splits (or any S >= 1). */
int radet_stem_wgrad_splits(int B, int H, int W);
int radet_stem_wgrad(const float* img_nchw, const float* ds, float* slabs, float* dbias_partials, int B, int H, int W, int S,
                     void* stream);

/* ---- GroupNorm(32, 256) + ReLU over a multi-level buffer (atss_head.py:32,60-76 via mmcv ConvModule) */
int radet_gn_workspace_floats(int B, const int* seg_desc, int nseg);
int radet_gn_relu_fwd(const float* z, const float* gamma, const float* beta, float* y, float* stats /*[nseg*B][32][2]*/,
                      float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc, int nseg,
                      void* stream);
int radet_gn_relu_bwd(const float* dy, const float* z, const float* stats, const float* gamma, const float* beta,
                      float* dz, float* dgamma, float* dbeta, float* partial_ws, int B, int C, int groups, int relu,
                      const int* seg_desc, int nseg, void* stream);

/* ---- FPN top-down: dst += nearest_upsample(src) and its adjoint (fpn.py:182-191) */
int radet_upsample_add(float* dst, const float* src, int B, int Ho, int Wo, int Hi, int Wi, int C, void* stream);
int radet_upsample_add_bwd(float* dsrc, const float* ddst, int B, int Ho, int Wo, int Hi, int Wi, int C, void* stream);
/* ReLU backward for a junction with no producing GEMM: dx = (dy + addend?) * [act > 0]; n % 4 == 0 */
int radet_relu_bwd(const float* dy, const float* addend, const float* act, float* dx, size_t n, void* stream);
/* bf16-storage variants (activation tensors are bf16; parameters, statistics and workspaces stay fp32) and the row
 * converter used at the fp32 boundaries (loss gradients, module outputs) */
int radet_stem_conv_bn_relu_h(const float* img_nchw, const float* wf_ohwi, const float* bias, void* y_nhwc, int B, int H,
                              int W, void* stream);
int radet_maxpool3x3s2_h(const void* x, void* y, int B, int H, int W, int C, void* stream);
/* Two tensors of one geometry (cls_convs[i].gn / reg_convs[i].gn, atss_head.py:60-76) in one pair of launches. */
int radet_gn_relu_fwd_pair(const float* z0, const float* gamma0, const float* beta0, float* y0, float* stats0,
                           float* partial_ws0, const float* z1, const float* gamma1, const float* beta1, float* y1,
                           float* stats1, float* partial_ws1, int B, int C, int groups, float eps, int relu,
                           const int* seg_desc, int nseg, void* stream);
int radet_gn_relu_fwd_pair_h(const void* z0, const float* gamma0, const float* beta0, void* y0, float* stats0,
                             float* partial_ws0, const void* z1, const float* gamma1, const float* beta1, void* y1,
                             float* stats1, float* partial_ws1, int B, int C, int groups, float eps, int relu,
                             const int* seg_desc, int nseg, void* stream);
int radet_gn_relu_fwd_h(const void* z, const float* gamma, const float* beta, void* y, float* stats, float* partial_ws,
                        int B, int C, int groups, float eps, int relu, const int* seg_desc, int nseg, void* stream);
int radet_gn_relu_bwd_h(const void* dy, const void* z, const float* stats, const float* gamma, const float* beta, void* dz,
                        float* dgamma, float* dbeta, float* partial_ws, int B, int C, int groups, int relu,
                        const int* seg_desc, int nseg, void* stream);
int radet_upsample_add_h(void* dst, const void* src, int B, int Ho, int Wo, int Hi, int Wi, int C, void* stream);
int radet_upsample_add_bwd_h(void* dsrc, const void* ddst, int B, int Ho, int Wo, int Hi, int Wi, int C, void* stream);
int radet_relu_bwd_h(const void* dy, const void* addend, const void* act, void* dx, size_t n, void* stream);
int radet_convert_rows(const void* src, void* dst, size_t rows, int ncols, int src_ld, int src_off, int dst_ld,
                       int dst_off, int to_bf16, void* stream);
/* ---- bf16 plane triples ("planes"): the operand format of the default fp32 arithmetic's conv GEMMs.  An fp32 value x is
 * stored as three bf16 numbers hi + mid + lo == x exactly (8 significand bits each, truncation); a row of C channels is
 * [3][C] bf16 = hi | mid | lo, rows back to back (6 bytes per element).  The producers of a tensor that only conv GEMMs read
 * (GroupNorm+ReLU outputs of the head towers, their gradients, folded weights with RadetConvDesc.w16 = 2) write planes once;
 * radet_conv2d_igemm (+0x2000000) and radet_conv2d_wgrad (flags 0x200) multiply them on the bf16 matrix cores with fp32
 * accumulation (6 of the 9 plane products, as +0x1000000) -- same results as splitting the fp32 operands inside the GEMM,
 * without the per-use VALU work.  Replaces nothing in the reference by itself: it is the storage format behind the convs of
 * resnet.py:260-299 / fpn.py:170-221 / atss_head.py:118-145.  C % 8 == 0, row strides in floats, % 4 == 0. */
int radet_split_planes(const float* src, void* dst_planes, size_t rows, int C, int src_ld, void* stream);
int radet_merge_planes(const void* src_planes, float* dst, size_t rows, int C, int dst_ld, void* stream);
/* ---- fp16 plane pairs ("plane pairs"): the operand format of the fp16 hi / lo arithmetic for tensors that only conv GEMMs
 * read.  A row of C channels (C % 32 == 0) is C / 32 groups of 128 bytes [hi x 32 | lo x 32] fp16 -- 4 bytes per element,
 * one cache line per 32-channel K stage and row -- holding x 2^e split as described at RadetScales, e from the tensor's amax
 * slot.  radet_split_pairs scales by the slot of src and copies the slot's bits to dst_amax; radet_merge_pairs is the
 * inverse (tests / API boundary).  radet_conv2d_igemm_s (+0x8000000 + 0x2000000) and radet_conv2d_wgrad_s (0x1000 + 0x200)
 * multiply them with three v_mfma_f32_32x32x16_f16 per K = 16 step and no operand work in their K loops. */
int radet_split_pairs(const float* src, void* dst_pairs, size_t rows, int C, int src_ld, const void* src_amax, void* dst_amax,
                      void* stream);
int radet_merge_pairs(const void* src_pairs, float* dst, size_t rows, int C, int dst_ld, const void* amax, void* stream);
/* GroupNorm + ReLU with plane-pair outputs.  forward: y (fp32, may be NULL; y_amax, may be NULL, is raised to its largest
 * magnitude) and / or yq (plane pairs, may be NULL) scaled by the power of two of a bound on |y| that needs no pass over the
 * data -- max|gamma| sqrt(n - 1) + max|beta| for the n values of a group -- which is stored to yq_amax; zhat_amax (may be
 * NULL) is raised to the largest normalised magnitude.  backward: dz (fp32, may be NULL) and / or dzq (plane pairs) scaled by
 * the bound rstd_max max|gamma| dy_amax (2 + zhat_max), stored to dzq_amax; dy_amax = the amax slot of dy (required with
 * dzq), zhat_amax from the forward call (NULL: sqrt(n - 1)). */
int radet_gn_relu_fwd_q(const float* z, const float* gamma, const float* beta, float* y, void* yq, float* stats,
                        float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc, int nseg,
                        void* stream, void* y_amax, void* yq_amax, void* zhat_amax);
int radet_gn_relu_fwd_pair_q(const float* z0, const float* gamma0, const float* beta0, float* y0, void* yq0, float* stats0,
                             float* partial_ws0, void* y_amax0, void* yq_amax0, void* zhat_amax0, const float* z1,
                             const float* gamma1, const float* beta1, float* y1, void* yq1, float* stats1,
                             float* partial_ws1, void* y_amax1, void* yq_amax1, void* zhat_amax1, int B, int C, int groups,
                             float eps, int relu, const int* seg_desc, int nseg, void* stream);
int radet_gn_relu_bwd_q(const float* dy, const float* z, const float* stats, const float* gamma, const float* beta, float* dz,
                        void* dzq, float* dgamma, float* dbeta, float* partial_ws, int B, int C, int groups, int relu,
                        const int* seg_desc, int nseg, void* stream, const void* dy_amax, const void* zhat_amax,
                        void* dzq_amax);
/* Elementwise kernels that also raise the amax slot of the tensor they write (not reset here), and a stand-alone pass for
 * tensors whose producer does not (n % 4 == 0). */
int radet_maxpool3x3s2_a(const float* x, float* y, int B, int H, int W, int C, void* y_amax, void* stream);
/* the max-pool with its output once more as fp16 plane pairs (scaled by x's amax slot, whose bits go to yq_amax), and the
 * stem with an amax slot for its output */
int radet_maxpool3x3s2_q(const float* x, float* y, int B, int H, int W, int C, void* y_amax, void* yq, void* yq_amax,
                         const void* x_amax, void* stream);
int radet_stem_conv_bn_relu_a(const float* img_nchw, const float* wf_ohwi, const float* bias, float* y_nhwc, int B, int H, int W,
                              void* y_amax, void* stream);
int radet_upsample_add_a(float* dst, const float* src, int B, int Ho, int Wo, int Hi, int Wi, int C, void* dst_amax,
                         void* stream);
int radet_upsample_add_bwd_a(float* dsrc, const float* ddst, int B, int Ho, int Wo, int Hi, int Wi, int C, void* dsrc_amax,
                             void* stream);
int radet_relu_bwd_a(const float* dy, const float* addend, const float* act, float* dx, size_t n, void* dx_amax, void* stream);
int radet_absmax(const float* x, size_t n, void* amax, void* stream);
int radet_amax_slot_words(void);   /* 32-bit words to allocate (and zero) per amax slot */
/* GroupNorm + ReLU with plane outputs: y / dz as fp32 (may be NULL) and / or as planes (yp / dzp, may be NULL) */
int radet_gn_relu_fwd_p(const float* z, const float* gamma, const float* beta, float* y, void* yp, float* stats,
                        float* partial_ws, int B, int C, int groups, float eps, int relu, const int* seg_desc, int nseg,
                        void* stream);
int radet_gn_relu_fwd_pair_p(const float* z0, const float* gamma0, const float* beta0, float* y0, void* yp0, float* stats0,
                             float* partial_ws0, const float* z1, const float* gamma1, const float* beta1, float* y1,
                             void* yp1, float* stats1, float* partial_ws1, int B, int C, int groups, float eps, int relu,
                             const int* seg_desc, int nseg, void* stream);
int radet_gn_relu_bwd_p(const float* dy, const float* z, const float* stats, const float* gamma, const float* beta,
                        float* dz, void* dzp, float* dgamma, float* dbeta, float* partial_ws, int B, int C, int groups,
                        int relu, const int* seg_desc, int nseg, void* stream);
int radet_nchw_to_nhwc(const float* x, float* y, int B, int C, int H, int W, void* stream);
int radet_nhwc_to_nchw(const float* x, float* y, int B, int C, int H, int W, void* stream);

/* ---- head loss: targets + sigmoid focal + GIoU + IoU-BCE, forward and gradients in one pass.
 *      Replaces RADetHead.get_targets/_get_target_single/loss (dense_heads/radet_head.py:173-392),
 *      TBLRBBoxCoder.encode/decode (core/bbox/coder/tblr_bbox_coder.py:71-172), bbox_overlaps aligned
 *      (core/bbox/iou_calculators/iou2d_calculator.py:116-158), FocalLoss/GIoULoss/CrossEntropyLoss
 *      (models/losses/{focal,iou,cross_entropy}_loss.py) and mmcv.ops.sigmoid_focal_loss.
 *  Rows are level-major: level l rows [off_l, off_l + B*h_l*w_l), image-major inside, row-major pixels.
 *  cls [R,C] logits; reg_u [R,4] = atss_reg output BEFORE Scale/ReLU; iou [R] logits; scales [nlvl].
 *  gt_boxes [sumG,4], gt_labels i64 [sumG], gt_off i32 [B+1]; p2g i64 [B,N], pw f32 [B,N] (N = points/img).
 *  level_desc (host): nlvl x 3 ints {h, w, stride}.  grad_scale: 3 device floats (upstream grads) or NULL.
 *  Outputs: losses[3] = {loss_cls, loss_bbox, loss_iou}; dcls [R,dcls_ld]; dreg_u [R,dreg_ld]; diou [R,diou_ld]
 *  (row strides let the gradients land in zero-padded buffers that the dgrad GEMM consumes directly; the
 *  sparse dreg_u / diou rows of non-positive points are zero-filled here); dscales [nlvl];
 *  optional dumps (may be NULL): labels_out i64 [R], bbox_targets_out [R,4].
 *  flags bit 0: reg_u holds the head's bbox_pred AFTER Scale + ReLU (the tensor `RADetHead.loss` receives,
 *  radet_head.py:173-181; pass scales = 1): dreg_u is then the gradient w.r.t. that tensor (no ReLU mask).
 *  ws: int32 workspace of radet_head_loss_ws_ints(R) ints. */
int radet_head_loss_ws_ints(int R);
int radet_head_loss(const float* cls, const float* reg_u, const float* iou, const float* scales,
                    const float* gt_boxes, const int64_t* gt_labels, const int* gt_off, const int64_t* p2g,
                    const float* pw, const int* level_desc, int nlvl, int B, int num_classes, float alpha,
                    float gamma, float loss_bbox_weight, float giou_eps, const float* grad_scale, float* losses,
                    float* dcls, int dcls_ld, float* dreg_u, int dreg_ld, float* diou, int diou_ld, float* dscales,
                    int64_t* labels_out, float* bbox_targets_out, int flags, int* ws, void* stream);
/* bbox_pred = relu(reg_u * scale_level) materialised for the module API (atss_head.py:143, radet_head.py:29) */
int radet_scale_relu(const float* reg_u, const float* scales, float* out, const int* level_desc, int nlvl, int B,
                     void* stream);

/* ---- inference: threshold / per-level top-k / TBLR decode (radet_head.py:55-146, atss_head.py:325-387) */
/* For each image b: candidates written to cand_* [B, cap] (cap = nlvl * nms_pre), count in cand_count[b].
 * Per level: scores = sigmoid(cls) > score_thr, top nms_pre by score (ties: lower flat index first),
 * boxes decoded with clamp to (img_h, img_w) and divided by scale_factor[b] (4 floats per image). */
size_t radet_decode_ws_bytes(int B, int nlvl, int nms_pre);
int radet_decode_candidates(const float* cls, const float* reg_u, const float* iou, const float* scales,
                            const int* level_desc, int nlvl, int B, int num_classes, float score_thr, int nms_pre,
                            const float* img_hw /*[B,2]*/, const float* scale_factor /*[B,4] or NULL*/,
                            float* cand_boxes, float* cand_scores, float* cand_ctr, int64_t* cand_labels,
                            int* cand_count, void* ws, void* stream);

/* ---- NMS family (radet/ops/vote/vote_ext.cpp:70-353, radet/ops/cluster/cluster_ext.cpp:4-87,
 *      mmcv.ops.batched_nms as used at radet_head.py:160). Batched over images: inputs [B, cap] with
 *      counts[b] valid entries.  mode: 0 vote, 1 global_vote, 2 cluster (ids), 3 hard batched NMS.
 *      Outputs (capacity max_out per image, heads in descending cluster score):
 *        out_boxes [B,max_out,4], out_scores [B,max_out], out_labels i64 [B,max_out], out_count [B];
 *      mode 2: instance_id i64 [B,cap], cluster_num i64 [B,cap]; mode 3: keep i64 [B,max_out] (input index).
 *      ws: workspace of radet_nms_ws_bytes(B, cap) bytes.  cap <= 65536: up to 8192 candidates per image the sorts run in
 *      LDS and a crowded label's greedy pass out of registers; above, the sorts run in global memory and label segments
 *      longer than 8192 boxes take a general pass (same results; the reference ops have no size limit,
 *      cluster_ext.cpp:4-87 / vote_ext.cpp:70-207). */
size_t radet_nms_ws_bytes(int B, int cap);
int radet_nms(const float* boxes, const float* cluster_scores, const float* vote_scores, const int64_t* labels,
              const int* counts, int B, int cap, int mode, float iou_thr, int iou_enable, float sigma, int max_out,
              float* out_boxes, float* out_scores, int64_t* out_labels, int* out_count, int64_t* aux0, int64_t* aux1,
              void* ws, void* stream);

/* ---- visibility-guided positive-sample assigner (radet/datasets/pipelines/label_assignment.py:57-201).
 *      One image per workgroup.  masks u8 [sumG, H, W]; rng_words u32 [B, U] = the next U raw 32-bit outputs of the image's
 *      RandomState (MT19937; a random_sample() is two words); out p2g i64 [B,N], pw f32 [B,N], used i32 [B] (words
 *      consumed; -1 = stream exhausted, -2 = more than 256 gts, -3 = an adapted positive_num above 64).
 *      flags: bit 0 balance_sample, bit 1 multiply_samplepro_for_weight, bit 2 adapt_positive_num, bit 3 NOT
 *      random_sample_by_distance (uniform integer draws: randint / permutation) -- the constructor arguments of
 *      label_assignment.py:30-46; the BOP configs use flags = 1.  ws: radet_assign_ws_bytes(B, N) bytes. */
size_t radet_assign_ws_bytes(int B, int N);
int radet_assign_points(const float* gt_boxes, const int* gt_off, const uint8_t* masks, int H, int W,
                        const uint32_t* rng_words, int U, const int* level_desc, const float* regress_ranges /*host, nlvl x 2*/,
                        int nlvl, int B, int positive_num, int flags, float neg_threshold, int64_t* p2g, float* pw, int* used,
                        void* ws, void* stream);

/* same with float per-box distance maps f32 [sumG, H, W] (mask-free sampler: MBD / GDT output mapped into the image,
 * radet/datasets/pipelines/loading.py:586-645, read as np.float32 by label_assignment.py:85-92) */
int radet_assign_points_f(const float* gt_boxes, const int* gt_off, const float* distance_maps, int H, int W,
                          const uint32_t* rng_words, int U, const int* level_desc, const float* regress_ranges /*host, nlvl x 2*/,
                          int nlvl, int B, int positive_num, int flags, float neg_threshold, int64_t* p2g, float* pw, int* used,
                          void* ws, void* stream);

/* ---- box-to-distance transforms of the mask-free sampler (GenerateDistanceMap(with_gt_mask=False)):
 *      pybind ops bbox2distance_ext.{MBD, GDT} (radet/ops/bbox2distance/bbox2distance_ext.cpp:127-133, 225-236), batched
 *      over box crops.  img_desc_dev[n][5] = (pixel offset of the crop in the packed arrays, h, w, first seed, seeds);
 *      images u8 [px][3] (HWC), dmap f64 [px], cost / dist f32 [px]; seeds as int32.  Bit-identical to the
 *      reference's sequential raster scans. */
size_t radet_mbd_ws_bytes(size_t total_px);
int radet_mbd(const uint8_t* images, const int* img_desc_dev, int nimg, const int* seeds_x, const int* seeds_y, float alpha,
              int niter, int base_size, double* dmap, size_t total_px, void* ws, void* stream);
int radet_gdt(const float* cost, const int* img_desc_dev, int nimg, const int* seeds_x, const int* seeds_y, float* dist,
              void* ws_labels /* int32 [px] */, void* stream);

/* ---- image processing around those transforms (radet/ops/bbox2distance/bbox2distance_wrapper.py:80-93, 118-130,
 *      170-181: cv2.resize / cv2.GaussianBlur / cv2.cvtColor / cv2.Sobel / cv2.addWeighted), batched over packed crops:
 *      *_desc (device) = ncrop x 3 ints {pixel offset, height, width}; max_*_px = the largest crop's pixel count (grid).
 *      OpenCV's generic algorithms restated (cv2 is absent: parity unpinned against cv2, pinned by oracle/imgproc.py):
 *      8-bit INTER_LINEAR with 11-bit fixed-point coefficients, float / double INTER_LINEAR, 9x9 Gaussian (sigma 1.7,
 *      kernel5 (host) = centre tap + 4 taps, BORDER_REFLECT_101, round to nearest even; tmp = 3 floats per pixel), and
 *      the Sobel edge map (3x3 Gaussian, RGB2GRAY, |0.5 d/dx + 0.5 d/dy| / max; gray_ws 1 byte per pixel, max_ws
 *      one word per crop). */
int radet_resize_linear_u8(const uint8_t* src, const int* src_desc, uint8_t* dst, const int* dst_desc, int ncrop,
                           int max_dst_px, int channels, void* stream);
int radet_resize_linear_f(const void* src, const int* src_desc, void* dst, const int* dst_desc, int ncrop, int max_dst_px,
                          int is_f64, void* stream);
int radet_gaussian_blur9_u8(const uint8_t* src, const int* desc, uint8_t* dst, float* tmp, const float* kernel5, int ncrop,
                            int max_px, void* stream);
int radet_sobel_edge(const uint8_t* src, const int* desc, float* edge, uint8_t* gray_ws, uint32_t* max_ws, int ncrop,
                     int max_px, void* stream);

/* ---- instance-mask path feeding the assigner: BitmapMasks.rescale / resize / flip / pad
 *      (core/mask/structures.py:253-303: mmcv.imresize = cv2.INTER_NEAREST, np.flip, np.pad per mask) fused into one
 *      pass over a [G,Hs,Ws] u8 stack, and LoadAnnotations._load_bop_masks' normalisation (loading.py:419-422).
 *      dst[g][y][x] = y < Hr && x < Wr ? norm(src[g][nn(flip_y(y))][nn(flip_x(x))]) : pad_val, dst is [G,Hd,Wd];
 *      flip: 0 none, 1 horizontal, 2 vertical, 3 diagonal; norm_max (from radet_mask_max) or NULL = no normalisation */
int radet_mask_max(const uint8_t* masks, uint32_t* maxes /* [G] */, int G, size_t hw, void* stream);
int radet_mask_transform(const uint8_t* src, uint8_t* dst, const uint32_t* norm_max, int G, int Hs, int Ws, int Hr, int Wr,
                         int Hd, int Wd, int flip, int pad_val, void* stream);

/* ---- stand-alone box / loss operators behind the registered classes (used on their own; inside the detector the same
 *      arithmetic runs fused in radet_head_loss / radet_decode_candidates) ------------------------------------------ */
/* bbox_overlaps / BboxOverlaps2D (radet/core/bbox/iou_calculators/iou2d_calculator.py:43-159): boxes [batch, M, 4] and
 * [batch, N, 4]; mode 0 iou, 1 iof, 2 giou; aligned: out [batch, M] (M == N) else the M x N matrix out [batch, M, N].
 * Operation order = the reference's PyTorch expressions, results equal PyTorch-CPU fp32 bit for bit. */
int radet_bbox_overlaps(const float* bboxes1, const float* bboxes2, float* out, int batch, int M, int N, int mode,
                        int aligned, float eps, void* stream);
/* TBLRBBoxCoder.encode / decode = bboxes2tblr / tblr2bboxes (radet/core/bbox/coder/tblr_bbox_coder.py:71-172).
 * normalizer4 (host): the 4 normalisation factors (a scalar normalizer repeated); decode clamps to
 * [0, max_w] x [0, max_h] when clip != 0 (clip_border and max_shape given). */
int radet_tblr_encode(const float* priors, const float* gts, float* out, int n, const float* normalizer4,
                      int normalize_by_wh, void* stream);
int radet_tblr_decode(const float* priors, const float* tblr, float* out, int n, const float* normalizer4,
                      int normalize_by_wh, float max_h, float max_w, int clip, void* stream);
/* Elementwise losses with `weight_reduce_loss` semantics (radet/models/losses/utils.py:24-51).  Forward: optional
 * loss_elem = loss * weight * elem_scale per element, optional partials[radet_loss_partials(n_elem)] = per-workgroup sums that
 * radet_loss_finalize adds in a fixed order: out[0] = sum / avg_factor[0] (device scalar, may be NULL) * scale.
 * weight_cols: 0 none, 1 per row, C per element.  Backward: d(input) = dloss/dinput * weight * g with
 * g = grad_elem[i] * scale (reduction 'none') or grad_scalar[0] * scale / avg_factor[0] (reduced); grad_elem,
 * grad_scalar, avg_factor may each be NULL.
 *   sigmoid focal  : mmcv.ops.sigmoid_focal_loss as wrapped by radet/models/losses/focal_loss.py:44-86 (target = class
 *                    index per row, anything outside [0, C) = background)
 *   bce with logits: radet/models/losses/cross_entropy_loss.py:57-92 (use_sigmoid=True), float targets [N, C]
 *   giou           : radet/models/losses/iou_loss.py:82-98, boxes [N, 4], weight [N] or NULL, gradient w.r.t. pred */
int radet_loss_partials(size_t n_elem);
int radet_loss_finalize(const float* partials, int npartials, const float* avg_factor, float scale, float* out,
                        void* stream);
int radet_sigmoid_focal_loss(const float* logits, const int64_t* target, const float* weight, int weight_cols, size_t N,
                             int C, float gamma, float alpha, float* loss_elem, float elem_scale, float* partials,
                             void* stream);
int radet_sigmoid_focal_loss_bwd(const float* logits, const int64_t* target, const float* weight, int weight_cols, size_t N,
                                 int C, float gamma, float alpha, const float* grad_elem, const float* grad_scalar,
                                 const float* avg_factor, float scale, float* dlogits, void* stream);
int radet_bce_logits_loss(const float* logits, const float* target, const float* weight, int weight_cols, size_t N, int C,
                          float* loss_elem, float elem_scale, float* partials, void* stream);
int radet_bce_logits_loss_bwd(const float* logits, const float* target, const float* weight, int weight_cols, size_t N,
                              int C, const float* grad_elem, const float* grad_scalar, const float* avg_factor, float scale,
                              float* dlogits, void* stream);
int radet_giou_loss(const float* pred, const float* target, const float* weight, size_t N, float eps, float* loss_elem,
                    float elem_scale, float* partials, void* stream);
int radet_giou_loss_bwd(const float* pred, const float* target, const float* weight, size_t N, float eps,
                        const float* grad_elem, const float* grad_scalar, const float* avg_factor, float scale, float* dpred,
                        void* stream);
/* multiclass_nms' score filter (radet/core/post_processing/bbox_nms.py:54-56): idx[0..count) = ascending indices i
 * with scores[i] > thr (torch.nonzero order); one workgroup, ordered ballot-prefix compaction. */
int radet_threshold_compact(const float* scores, size_t n, float thr, int64_t* idx, int* count, void* stream);

/* ---- anchors (core/anchor/anchor_generator.py:206-271): [sum h*w, 4], centre (j*stride, i*stride), side 8*stride */
int radet_grid_anchors(float* out, const int* level_desc, int nlvl, int octave_base_scale, void* stream);

/* ---- optimiser step on flat arenas: global L2 grad-norm clip + AdamW (torch.optim.AdamW +
 *      clip_grad_norm_ as driven by mmcv OptimizerHook; configs/base/default_runtime.py:1-19) */
int radet_sqnorm_partials(const float* g, size_t n, float* partials, int npartials, void* stream);
int radet_adamw_step(float* p, const float* g, float* m, float* v, size_t n, float lr, float beta1, float beta2,
                     float eps, float weight_decay, int step, float max_norm, float grad_div,
                     const float* partials, int npartials, float* grad_norm_out, void* stream);

/* ---- launch tape (round 6): the host side of a steady-state train step as ONE call.  The reference drives its step from
 *      Python, one autograd node and one cuDNN / ATen launch at a time (mmcv's EpochBasedRunner.train -> model.train_step ->
 *      radet/models/detectors/base.py:218-253 -> optimizer hook); here a step is ~250 C-ABI calls on four HIP streams, and
 *      issuing them from Python costs ~19 us each -- 4.5-4.9 ms of host time per step, which bounds the bf16-storage step and
 *      would bound every step on a node whose ranks share the host.  A tape is a host array of RadetTapeOp recorded from one
 *      eager step (radet_amd/tape.py): every entry point of this header that was called, with its argument values, and the
 *      event record / wait operations between the streams.  radet_tape_replay issues ops [first, last) in order -- exactly the
 *      calls the eager step made, so results are bit-identical -- without re-entering Python.  Arguments that change from step
 *      to step (input / target pointers, learning rate, step number) are patched in the array by the owner before the call.
 *      Not a hipGraph: the launches stay ordinary stream-ordered launches (graphs measured slower here), the
 *      gradient exchange stays in Python between two replayed segments. */
typedef struct RadetTapeOp {
    int32_t kind;        /* 0: call thunk `fn` with args; 1: hipEventRecord(event, stream); 2: hipStreamWaitEvent(stream, event) */
    int32_t fn;          /* kind 0: index returned by radet_tape_fn_index */
    void* stream;        /* kind 1 / 2 (host handle: hipStream_t) */
    void* event;         /* kind 1 / 2 (host handle: hipEvent_t) */
    uint64_t args[32];   /* kind 0: one 64-bit word per argument in declaration order: pointers / size_t as is, int sign-extended,
                            float as its bit pattern in the low 32 bits */
} RadetTapeOp;
int radet_tape_fn_index(const char* name);    /* index of an `int radet_*(...)` entry point of this header, -1 if unknown (host) */
/* issues ops[first .. last); stops at the first op that fails: returns its code and stores its index in *failed (host, may be
 * NULL); 0 when all were issued */
int radet_tape_replay(const RadetTapeOp* ops, int first, int last, int* failed);
/* stream-ordered helpers a taped step uses instead of torch's fill / copy (hipMemsetAsync / hipMemcpyAsync device to device) */
int radet_fill_zero(void* dst, size_t nbytes, void* stream);
/* a HIP stream whose kernels may only run on the compute units whose bit is set in cu_mask (host array of nwords 32-bit words, bit i
 * of word w = CU 32 w + i; hipExtStreamCreateWithCUMask): the engine's weight-gradient streams can be kept off a share of the CUs
 * so that the dependent dgrad chain always finds free ones (experiment switch RADET_WGRAD_CU_MASK).  *stream_out: hipStream_t (host) */
int radet_stream_create_cumask(const uint32_t* cu_mask, int nwords, void** stream_out);
int radet_copy_d2d(void* dst, const void* src, size_t nbytes, void* stream);

#ifdef __cplusplus
}
#endif
#endif
