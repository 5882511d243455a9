/*
 * disyolo.h -- C ABI of the MI355X (gfx950) kernel library for the DIS-YOLO hot path.
 *
 * The reference (ZHANGKEON/DIS-YOLO) has no FFI/plugin interface: its hot path is a
 * TensorFlow-1.x graph (yolo/yolo3_net_pos.py) driven through tf.Session.run.  Each
 * entry point below replaces the TF op call sites named in its comment (file:line in
 * the reference checkout).  A maintainer binds them with ctypes.
 *
 * Conventions
 *   - extern "C", plain pointers + sizes; no torch / C++ types.
 *   - every pointer is a DEVICE pointer unless stated; `stream` is a hipStream_t
 *     passed as void*.  Calls are stream-ordered, never synchronise, never allocate
 *     (graph-capture safe); scratch comes from caller-provided workspaces.
 *   - return 0 on success, a negative DISYOLO_E_* code otherwise.  Nothing throws or
 *     exits.  disyolo_last_error() returns a static message for the calling thread.
 *   - activations are NHWC bf16; logits / score maps / losses / statistics are f32;
 *     master weights are f32 HWIO [k,k,Cin,Cout] (the reference checkpoint layout,
 *     yolo/yolo3_net_pos.py:112,118); the MFMA kernels read a packed bf16 copy made
 *     by disyolo_pack_weights.
 */
#ifndef DISYOLO_H
#define DISYOLO_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define DISYOLO_OK 0
#define DISYOLO_E_ARG (-1)      /* invalid argument / unsupported shape */
#define DISYOLO_E_WORKSPACE (-2) /* workspace too small */
#define DISYOLO_E_HIP (-3)      /* a HIP runtime call failed */

#define DISYOLO_GRAD_LD 32      /* channel pitch of the head-logit / score-map gradients */
#define DISYOLO_ROI_MAX 16      /* RoI slots per image in the mask-loss RoI table       */
#define DISYOLO_ROI_W 12        /* int32 words per RoI: gy0..3, gx0..3, gt_row, area, valid, 0 */

int disyolo_version(void);
const char* disyolo_last_error(void);

/* ---- convolution (replaces tf.nn.conv2d + bias_add + folded batch_normalization +
 *      leaky_relu + residual add + resize_nearest_neighbor/concat feeding a 1x1 conv:
 *      yolo/yolo3_net_pos.py:125-129,142-145,150,290-291,325-326,386-387,401-402) ---- */
enum {
  DISYOLO_CONV_LEAKY = 1,       /* y = max(alpha*y, y) after scale/shift            */
  DISYOLO_CONV_OUT_F32 = 2,     /* y is f32 (head logits / score maps), else bf16    */
  DISYOLO_CONV_STATS = 4,       /* also emit per-channel (sum, sum of squares) of the
                                   raw accumulators into `stats` (training BN, :90)  */
  DISYOLO_CONV_BN_BWD_STATS = 8,/* y is the (now final) gradient wrt the OUTPUT of a batch-
                                   normalised layer: also emit that layer's batch-norm
                                   backward sums, per channel over this block's pixels,
                                   (sum g, sum g*xhat) with g = y*act'(bn_x*bn_scale+bn_shift),
                                   xhat = (bn_x-bn_mean)*bn_rstd, from the bf16 values
                                   stored to y, into `bn_partials` -- the column reduction
                                   disyolo_bn_act_bwd would otherwise start with.  Only the
                                   3x3 patch kernels have this epilogue (tile ids 16-18, 24,
                                   25: disyolo_conv2d_bn_bwd_stats_ok tells), bf16 y,
                                   Cout % 8 == 0 (TF autodiff of :68-107)                */
  DISYOLO_CONV_BN_FUSED = 16,   /* with DISYOLO_CONV_STATS: training-mode batch norm INSIDE the launch
                                   (tf.nn.moments + batch_normalization + the moving-average assigns +
                                   leaky_relu, :90-107, in the conv's epilogue).  The blocks that share a
                                   channel tile exchange their statistics rows through `stats` within the
                                   launch (write-through stores, an arrival counter in `cluster_sync`,
                                   every block sums the rows in the same fixed order), then write BOTH
                                   y = the conv output (bf16, what the backward pass keeps) and y_act =
                                   leaky(y*scale + shift) computed from the bf16-rounded y -- bit for bit
                                   what disyolo_bn_finalize + disyolo_bn_act_fwd produce from y (up to the
                                   f64 summation order of the statistics rows) -- and one block per channel
                                   tile writes bn_out_* and updates the moving statistics.  Needs every
                                   block of the launch resident at once: disyolo_conv2d_bn_fused_ok tells;
                                   never run two such launches concurrently on one device (each waits for
                                   its own blocks: with both half-resident neither completes; the wait is
                                   bounded and reports through disyolo_cluster_sync_error)             */
  DISYOLO_CONV_BN_BWD_FUSED = 32,/* with the bn_* fields of DISYOLO_CONV_BN_BWD_STATS: the whole batch-norm
                                   backward of the target layer inside the data-gradient conv that makes its
                                   output gradient final: the sums are exchanged within the launch like the
                                   forward statistics, then y receives dx of the TARGET's conv output
                                   (scale*(g - mean(g) - xhat*mean(g*xhat))) instead of the gradient wrt its
                                   activation, and one block per channel tile writes bn_dgamma / bn_dbeta.
                                   disyolo_conv2d_bn_fused_ok tells; same residency rule                */
};

typedef struct disyolo_conv_desc {
  int32_t B, H, W;        /* source-0 spatial size                                   */
  int32_t C0, C1;         /* channels of source 0 / source 1 (C1 = 0: single source) */
  int32_t Ho, Wo, Cout;   /* output size                                              */
  int32_t ksize, stride;  /* 1 or 3; 1 or 2                                           */
  int32_t pad_t, pad_l;   /* TF 'SAME' leading pads (asymmetric for k=3,s=2)          */
  int32_t in_div;         /* 1 = forward gather.  s>1 = transposed gather for the
                             data-gradient of a stride-s conv: a tap contributes only
                             where (yo*stride + kh - pad_t) is divisible by in_div    */
  int32_t flags;          /* DISYOLO_CONV_*                                           */
  float alpha;            /* leaky slope                                              */
  int32_t tile;           /* 0 = the launcher's heuristic; else a tile code a caller-side
                             tuner pins (what YOLONet.autotune does): low byte = id
                             (1-12 GEMM block tiles 64x64 ... 256x128, 13-15 the same
                             with two K groups per block, 16/17 the 3x3 stride-1 patch
                             kernel with 8/4 waves, 18 = 16 with 32 instead of 64 output
                             channels per block, 20 = the persistent streaming form of
                             the patch kernel for 3x3 stride-1 layers with exactly 32
                             input channels, 21 = the streaming 1x1 kernel (stride 1,
                             Cout <= 64, C0 + C1 <= 192); an id that does not cover the shape
                             falls back), bit 8 = force K depth 32, bit 9 = the tile's
                             alternative pipeline depth                              */
  const void* x0;         /* bf16 [B,H,W,C0]                                          */
  const void* x1;         /* bf16 [B,H/2,W/2,C1]: nearest-upsampled x2 and concatenated
                             after x0 along channels (1x1 convs only), or NULL        */
  const void* w;          /* bf16 packed [Cout][ksize*ksize*(C0+C1)], k = (kh,kw,ci) */
  const float* scale;     /* [Cout] or NULL (=1)                                      */
  const float* shift;     /* [Cout] or NULL (=0)  (folded BN beta / conv bias)        */
  const void* residual;   /* bf16 [B,Ho,Wo,Cout] added after the activation, or NULL  */
  void* y;                /* bf16 or f32 [B,Ho,Wo,Cout]                               */
  float* stats;           /* f32 [disyolo_conv2d_stats_rows][Cout][2] or NULL         */
  /* DISYOLO_CONV_BN_BWD_STATS only (else ignored): */
  const void* bn_x;       /* bf16 [B,Ho,Wo,Cout]: the conv output the target layer normalised */
  const float* bn_scale;  /* [Cout] gamma*rstd                                         */
  const float* bn_shift;  /* [Cout] beta - mean*gamma*rstd                             */
  const float* bn_mean;   /* [Cout] batch mean                                         */
  const float* bn_rstd;   /* [Cout] 1/sqrt(var+eps)                                    */
  float* bn_partials;     /* f32 [disyolo_conv2d_stats_rows][Cout][2]                  */
  float bn_alpha;         /* the target layer's leaky slope                            */
  /* DISYOLO_CONV_BN_FUSED / DISYOLO_CONV_BN_BWD_FUSED only (else ignored): */
  float bn_decay, bn_eps; /* moving-average decay (0.997), variance epsilon            */
  void* y_act;            /* bf16 [B,Ho,Wo,Cout]: the activation (BN_FUSED)            */
  const float* bn_gamma;  /* [Cout]                                                    */
  const float* bn_beta;   /* [Cout]                                                    */
  float* bn_moving_mean;  /* [Cout] updated in place (or NULL)                         */
  float* bn_moving_var;   /* [Cout] updated in place (or NULL)                         */
  float* bn_out_scale;    /* [Cout] gamma*rstd              } what the backward pass   */
  float* bn_out_shift;    /* [Cout] beta - mean*gamma*rstd  } of this layer reads      */
  float* bn_out_mean;     /* [Cout] batch mean                                         */
  float* bn_out_rstd;     /* [Cout] 1/sqrt(var + eps)                                  */
  float* bn_dgamma;       /* [Cout] (BN_BWD_FUSED)                                     */
  float* bn_dbeta;        /* [Cout] (BN_BWD_FUSED)                                     */
  uint32_t* cluster_sync; /* disyolo_cluster_sync_words(Cout) words, zeroed ONCE by the caller
                             (the launches leave it zero); one buffer per layer and direction  */
} disyolo_conv_desc;

/* Host-side (no device work): the contour extraction of the dataset pre-processing, cv2.findContours(img,
 * cv2.RETR_TREE, cv2.CHAIN_APPROX_NONE) of pre_process.py:78-86 -- Suzuki-Abe border following, OpenCV's point
 * order and contour numbering (cv2 not available in the build environment: parity unpinned).  binary: h x w bytes,
 * non-zero = foreground.  points_xy int32 [n_points][2] (x, y); contour_start int32 [n_contours + 1]; hierarchy
 * int32 [n_contours][4] = next, previous, first child, parent.  Too-small buffers: DISYOLO_E_WORKSPACE with the
 * needed sizes in *n_contours / *n_points (call once with max 0 to size them). */
int disyolo_find_contours(const uint8_t* binary, int h, int w, int32_t* points_xy, int64_t max_points,
                          int32_t* contour_start, int32_t* hierarchy, int max_contours, int* n_contours,
                          int64_t* n_points);
/* sizeof(disyolo_conv_desc) as this library was built: a binding checks its mirror against it */
size_t disyolo_conv_desc_size(void);
/* 1 when a call with this descriptor runs a kernel that can emit DISYOLO_CONV_BN_BWD_STATS: the patch kernels and (round 6)
 * the GEMM tiles that carry that epilogue (1x1: 64x64, 64x128, 128x64, 96x128, 128x128; 1x1 and 3x3: 192x128), in_div 1 */
int disyolo_conv2d_bn_bwd_stats_ok(const disyolo_conv_desc* d);
/* 1 when a call with this descriptor (shape and tile as given) can run batch norm inside the launch -- the backward form
 * when DISYOLO_CONV_BN_BWD_FUSED is set in its flags, the forward form otherwise (DISYOLO_CONV_BN_FUSED need not be set):
 * a kernel that has the epilogue, bf16 y, Cout % 8 == 0, and a grid that is resident at once on this device (blocks <=
 * occupancy x compute units, queried from the runtime) with statistics rows few enough to be summed by every block */
int disyolo_conv2d_bn_fused_ok(const disyolo_conv_desc* d);
/* words of a `cluster_sync` buffer for a layer with Cout channels (any tile); the last word is the error word */
int disyolo_cluster_sync_words(int Cout);
/* 0, or the code a bounded in-launch wait left in the buffer's error word (device memory is read with a blocking copy:
 * call it at a synchronisation point, e.g. when the losses are fetched) */
int disyolo_cluster_sync_error(const uint32_t* cluster_sync, int Cout);
/* rows of the `stats` partial buffer a call with this descriptor writes */
int disyolo_conv2d_stats_rows(const disyolo_conv_desc* d);
int disyolo_conv2d_fwd(const disyolo_conv_desc* d, void* stream);
/* tile configuration the launcher picks for this descriptor: returns its id (> 0) and the
 * block tile (pixels x channels), K depth and pipeline stages -- the template arguments of the
 * conv_igemm_kernel instance that will run (ids 16/17: the conv_halo_kernel patch, 64
 * channels, 32-channel slices, 2 stages); bench.py attributes time per instance with it */
int disyolo_conv2d_tile(const disyolo_conv_desc* d, int* bm, int* bn, int* bk, int* stages);

/* first layer (Cin=3, k=3, s=1; yolo/yolo3_net_pos.py:159): f32 NHWC image in, exact f32
 * FMA with the f32 HWIO weights, folded BN + leaky, bf16 out. */
int disyolo_conv_first_fwd(const float* images, const float* w_hwio, const float* scale,
                           const float* shift, void* y_bf16, int B, int H, int W, int Cout,
                           float alpha, void* stream);

/* conv1 + conv2 in one launch when both run in inference mode (locked / inference net): act2 = leaky(bn2(conv3x3 stride 2
 * (leaky(bn1(conv3x3(images)))))), conv_bn 'convolutional1' + 'convolutional2', yolo/yolo3_net_pos.py:159-167 -- conv1's output (one consumer, the largest tensor of the
 * network) is never written.  images f32 NHWC [B,H,W,3]; w1 f32 HWIO [3,3,3,32]; w2 bf16 packed [64][9*32] (pack_weights);
 * folded BN scale / shift per layer; y bf16 [B,H/2,W/2,64].  conv1 runs on the bf16 matrix cores with hi/lo-split operands
 * (relative error 2^-16 per product against exact f32, before the rounding to bf16).  disyolo_conv12_fused_ok: 1 when the
 * sizes are covered (H/2 a multiple of 8, W/2 of 16). */
int disyolo_conv12_fused_ok(int B, int H, int W);
int disyolo_conv12_fused_fwd(const float* images, const float* w1_hwio, const float* scale1, const float* shift1,
                             const void* w2_packed, const float* scale2, const float* shift2, void* y_bf16, int B, int H,
                             int W, float alpha, void* stream);

/* The two HBM-bound [1x1 -> 32] -> [3x3 32 -> 64] chains of the half-resolution maps in one launch each, batch norms in
 * inference mode (folded scale / shift, leaky alpha); the 32- and 64-channel intermediates stay on chip:
 *   post 0  (C0 = 64, C1 = 0): y bf16 [B,H,W,64] = leaky(bnB(conv3x3(leaky(bnA(conv1x1(x0)))))) + x0 -- the first residual
 *           block (conv_bn 'convolutional3' + res_conv_bn 'convolutional4'), yolo/yolo3_net_pos.py:169-176;
 *   post 1  (C0 = 64, C1 = 32): y f32 [B,H,W,9] = conv1x1(leaky(bnB(conv3x3(leaky(bnA(conv1x1([x0, up2(x1)]))))))) + biasC -- the
 *           mask head (conv_bn 'convolutional80', 'convolutional81', conv 'convolutional82'), yolo/yolo3_net_pos.py:404-412; x1 bf16 [B,H/2,W/2,32] is read at (y/2, x/2).
 * x0 bf16 NHWC; wA packed [32][C0+C1], wB packed [64][9*32], wC packed [9][64] (pack_weights).  _ok: 1 when covered
 * (those two shapes, H a multiple of 8, W of 16). */
int disyolo_block32_fused_ok(int B, int H, int W, int C0, int C1, int post);
int disyolo_block32_fused_fwd(const void* x0, const void* x1, int C0, int C1, const void* wA, const float* scaleA,
                              const float* shiftA, const void* wB, const float* scaleB, const float* shiftB, int post,
                              const void* wC, const float* biasC, void* y, int B, int H, int W, float alpha, void* stream);

/* A residual block of the quarter-resolution maps in one launch, batch norms in inference mode: y bf16 [B,H,W,128] =
 * leaky(bnB(conv3x3(leaky(bnA(conv1x1(x)))))) + x with x bf16 [B,H,W,128], wA packed [64][128], wB packed [128][9*64]
 * (conv_bn + res_conv_bn, yolo/yolo3_net_pos.py:184-201, conv6+7 and conv8+9); the 64-channel intermediate stays in LDS.  _ok: C0 == 128, H a
 * multiple of 8, W of 16. */
int disyolo_block64_fused_ok(int B, int H, int W, int C0);
int disyolo_block64_fused_fwd(const void* x, const void* wA, const float* scaleA, const float* shiftA, const void* wB,
                              const float* scaleB, const float* shiftB, void* y, int B, int H, int W, int C0, float alpha,
                              void* stream);

/* Data gradient of a 3x3 stride-2 SAME conv over an even-sized input (TF autodiff of tf.nn.conv2d wrt its input,
 * train_yolo3_mask.py:55; conv2 / conv5 of the network, yolo/yolo3_net_pos.py:167,192) as one 2x2-tap conv over dy with a
 * depth-to-space store: dx bf16 [B, 2 Hdy, 2 Wdy, C] (+= residual when given) from dy bf16 [B, Hdy, Wdy, Cdy] and wq =
 * disyolo_pack_quad(w) (bf16 [4 C][9 Cdy], rebuilt from the f32 HWIO master [3][3][C][Cdy] whenever it changes).
 * The generic path (disyolo_conv2d_fwd with in_div = 2) gives the same values; this one is for the shallow layers
 * (C <= 64), where the generic tiles are bound by their fixed cost.  _ok: Cdy % 64 == 0, C in {32, 64}. */
int disyolo_pack_quad(const float* w_hwio, void* wq, int C, int Cdy, void* stream);
int disyolo_dgrad_s2_quad_ok(int B, int Hdy, int Wdy, int Cdy, int C);
int disyolo_dgrad_s2_quad(const void* dy, const void* wq, void* dx, const void* residual, int B, int Hdy, int Wdy, int Cdy,
                          int C, void* stream);

/* weight gradient (TF autodiff of tf.nn.conv2d wrt filters; train_yolo3_mask.py:55):
 * dw[kh,kw,ci,co] (f32 HWIO, overwritten) = sum_m xcol[m,(kh,kw,ci)] * dy[m,co].
 * Uses d->x0/x1 (the layer input, same gather as forward) and `dy` bf16 [B*Ho*Wo, dy_ld]
 * (dy_ld >= Cout, multiple of 8; columns >= Cout are ignored).
 * workspace: disyolo_conv2d_wgrad_workspace(d, opts) bytes.
 * The descriptor's `tile` field is NOT read (it pins the forward / data-gradient tile of the same GEMM shape);
 * `opts` = 0 for the product path, or DISYOLO_WGRAD_* bits for tuning and per-kernel timing. */
enum {
  DISYOLO_WGRAD_IM2COL = 1,       /* force the im2col kernel where the tap-fused 3x3 kernel would run   */
  DISYOLO_WGRAD_PARTIAL_ONLY = 2, /* launch only the partial-sum kernel (dw is NOT written if it splits) */
  DISYOLO_WGRAD_REDUCE_ONLY = 4,  /* launch only the slab reduction (workspace must hold the slabs)      */
  DISYOLO_WGRAD_STAGES_MASK = 0x30 /* im2col kernel pipeline depth: (n << 4), n = 1..3 -> 2..4 stages    */
};
size_t disyolo_conv2d_wgrad_workspace(const disyolo_conv_desc* d, int opts);
int disyolo_conv2d_wgrad(const disyolo_conv_desc* d, const void* dy, int dy_ld, float* dw,
                         void* workspace, size_t workspace_bytes, int opts, void* stream);
/* same for the first layer (f32 image input, Cin = 3): dw f32 [3,3,3,Cout] */
/* ---- fp8 (OCP e4m3) forward path of the inference-mode layers (BASELINE.json configs[4]; the reference
 * is f32: yolo/yolo3_net_pos.py:132-146 conv_bn with lock=True) ----
 * y = leaky(acc * escale[c] + eshift[c]) [+ residual_fp8 * residual_scale], acc = sum of e4m3 products in
 * f32 (v_mfma_f32_16x16x32_fp8_fp8).  desc: x0 = e4m3 NHWC input, geometry / alpha / LEAKY flag as for
 * conv2d_fwd (w, scale, shift, residual, y of the descriptor are ignored); w_fp8 = e4m3 [Cout][k*k*Cin]
 * (pack_weights_fp8); escale = s_in * s_w * folded-BN scale, eshift = folded-BN shift (f32 [Cout]).
 * Outputs: y_fp8 = e4m3 of y / out_scale and/or y_bf16 = bf16 of y (either may be NULL).
 * Cin a power of two >= 32, Cout a multiple of 16, no fused concat. */
int disyolo_conv2d_fp8_fwd(const disyolo_conv_desc* d, const void* w_fp8, const float* escale,
                           const float* eshift, const void* residual_fp8, float residual_scale,
                           void* y_fp8, float out_scale, void* y_bf16, void* stream);
/* first layer (f32 image, exact f32 FMA, folded BN + leaky) with the output stored as e4m3 of y / out_scale */
int disyolo_conv_first_fwd_fp8(const float* images, const float* w_hwio, const float* scale, const float* shift,
                               void* y_fp8, float out_scale, int B, int H, int W, int Cout, float alpha,
                               void* stream);
/* e4m3 of x / scale (x bf16, or f32 when x_is_f32; n % 8 == 0, saturating at +-448); back to f32 * scale */
int disyolo_quant_fp8(const void* x, int x_is_f32, void* y_fp8, int64_t n, float scale, void* stream);
int disyolo_dequant_fp8(const void* x_fp8, float* y, int64_t n, float scale, void* stream);
/* HWIO f32 weights -> e4m3 [Cout][k*k*Cin] of w / scale */
int disyolo_pack_weights_fp8(const float* w_hwio, void* w_fp8, int ksize, int Cin, int Cout,
                             float scale, void* stream);

/* which kernel the launcher picks for this descriptor: kind 0 = im2col kernel (tile_n = channel tile),
 * 1 = tap-fused 3x3 kernel, stride 1 or 2 (tile_n = channel tile, ring = input ring slots); splits = pixel splits
 * (f32 slabs summed by slab_reduce when > 1).  opts: DISYOLO_WGRAD_* as passed to disyolo_conv2d_wgrad. */
int disyolo_conv2d_wgrad_plan(const disyolo_conv_desc* d, int opts, int* kind, int* tile_n, int* ring, int* splits);
/* weight gradient of the first layer (yolo/yolo3_net_pos.py:159: 3 -> 32 filters, 3x3, stride 1): images f32 [B,H,W,3], dy bf16
 * [B,H,W,Cout], dw f32 [3][3][3][Cout].  Cout == 32 runs on the matrix cores with the 27 (tap, channel) pairs as the M axis (the image
 * is rounded to bf16 on its way into LDS); other Cout (27 * Cout <= 1024) by direct f32 accumulation. */
size_t disyolo_conv_first_wgrad_workspace(int B, int H, int W, int Cout);
int disyolo_conv_first_wgrad(const float* images, const void* dy, float* dw, int B, int H, int W,
                             int Cout, void* workspace, size_t workspace_bytes, void* stream);

/* first-layer helpers for the MFMA weight gradient: images f32 [pixels,3] -> bf16 [pixels,8]
 * (zero padded channels), and a pitched f32 copy used to drop the padded rows of its result */
int disyolo_image_pad8(const float* images, void* out_bf16, int64_t pixels, void* stream);
int disyolo_copy2d_f32(const float* src, float* dst, int rows, int cols, int src_ld, int dst_ld,
                       void* stream);

/* f32 HWIO master -> packed bf16 operands.  w_fwd [Cout][k*k*Cin] (may be NULL); w_dgrad
 * (may be NULL) [Cin][k*k*cout_pad] with taps flipped and zero columns for co >= Cout, i.e.
 * the forward operand of the data-gradient conv over a dy padded to cout_pad channels. */
int disyolo_pack_weights(const float* w_hwio, void* w_fwd, void* w_dgrad, int ksize, int Cin,
                         int Cout, int cout_pad, void* stream);

/* the same for many layers in ONE launch (re-pack after every optimizer step): the caller
 * builds a table with pack_table_build (host memory, pack_table_bytes(njobs) bytes), copies it
 * to the device once, and calls pack_all each step */
typedef struct disyolo_pack_job {
  const float* w_hwio;
  void* w_fwd;
  void* w_dgrad;      /* or NULL */
  int32_t ksize, Cin, Cout, cout_pad;
} disyolo_pack_job;
size_t disyolo_pack_table_bytes(int njobs);
int disyolo_pack_table_build(const disyolo_pack_job* jobs, int njobs, void* host_table, int* total_blocks);
int disyolo_pack_all(const void* device_table, int njobs, int total_blocks, void* stream);

/* ---- batch normalisation (yolo/yolo3_net_pos.py:71-107) ---- */
/* stats partials -> batch mean / population variance; scale = gamma*rsqrt(var+eps),
 * shift = beta - mean*scale; moving <- decay*moving + (1-decay)*batch (:93-96). */
int disyolo_bn_finalize(const float* stats, int rows, int C, int64_t count, const float* gamma,
                        const float* beta, float* moving_mean, float* moving_var, float decay,
                        float eps, float* scale, float* shift, float* mean, float* rstd,
                        void* stream);
/* SyncBN building blocks (data-parallel option; the reference is single-GPU): the phases of
 * bn_finalize / bn_act_bwd as separate calls, so that a caller can add the per-channel f64 sums [C][2] up over
 * the ranks (one small all-reduce) between them.  Forward: conv partials -> bn_partial_sums -> (all-reduce) ->
 * bn_finalize_sums(count = elements per channel over ALL ranks).  Backward: bn_bwd_reduce -> (all-reduce of a
 * copy) -> bn_bwd_apply_sums: dgamma/dbeta stay this rank's sums (the gradient exchange adds the ranks up), dx
 * uses the global ones.  With one rank and no all-reduce the results equal bn_finalize / bn_act_bwd. */
int disyolo_bn_partial_sums(const float* partials, int rows, int C, double* sums, void* stream);
int disyolo_bn_finalize_sums(const double* sums, int C, int64_t count, const float* gamma, const float* beta,
                             float* moving_mean, float* moving_var, float decay, float eps, float* scale,
                             float* shift, float* mean, float* rstd, void* stream);
int disyolo_bn_bwd_reduce_rows(int64_t rows, int C);   /* workspace of bn_bwd_reduce: rows x C x 2 floats */
int disyolo_bn_bwd_reduce(const void* dy, const void* x, const float* scale, const float* shift,
                          const float* mean, const float* rstd, int64_t rows, int C, float alpha, double* sums,
                          void* workspace, size_t workspace_bytes, void* stream);
int disyolo_bn_bwd_apply_sums(const void* dy, const void* x, const float* scale, const float* shift,
                              const float* mean, const float* rstd, const double* local_sums,
                              const double* global_sums, int64_t count, void* dx, float* dgamma, float* dbeta,
                              int64_t rows, int C, float alpha, void* shortcut_grad, int shortcut_accumulate,
                              void* workspace, size_t workspace_bytes, void* stream);
/* (sum, sum of squares) partials of a materialised bf16 [rows,C] conv output, in the layout
 * bn_finalize consumes: stats f32 [disyolo_colstats_rows(rows,C)][C][2] */
int disyolo_colstats_rows(int64_t rows, int C);
int disyolo_colstats(const void* x, float* stats, int64_t rows, int C, void* stream);
/* locked / inference BN folded to scale/shift from the moving statistics (:81,:101) */
int disyolo_bn_fold(const float* gamma, const float* beta, const float* moving_mean,
                    const float* moving_var, float eps, float* scale, float* shift, int C,
                    void* stream);
/* y = leaky(x*scale + shift) [+ residual], bf16 [rows,C] */
int disyolo_bn_act_fwd(const void* x, const float* scale, const float* shift, const void* residual,
                       void* y, int64_t rows, int C, float alpha, void* stream);
/* backward of y = leaky(gamma*xhat + beta), training statistics.  dy, x bf16 [rows,C];
 * writes dx bf16 and dgamma/dbeta f32.  workspace: disyolo_bn_act_bwd_workspace bytes.
 * shortcut_grad (bf16 [rows,C] or NULL): the gradient buffer of the layer a residual shortcut comes from
 * (res_conv_bn, :148-151): it receives dy (shortcut_accumulate = 0) or dy + itself (1) in the same pass that
 * reads dy for dx -- what disyolo_add_bf16(dy, shortcut_grad) would do in a launch of its own. */
size_t disyolo_bn_act_bwd_workspace(int64_t rows, int C);
int disyolo_bn_act_bwd(const void* dy, const void* x, const float* scale, const float* shift,
                       const float* mean, const float* rstd, void* dx, float* dgamma, float* dbeta,
                       int64_t rows, int C, float alpha, void* shortcut_grad, int shortcut_accumulate,
                       void* workspace, size_t workspace_bytes, void* stream);
/* the same when the per-channel sums (sum g, sum g*xhat) over disjoint row sets are already in
 * `partials` f32 [part_rows][C][2]: written by the data-gradient conv that produced dy, with
 * DISYOLO_CONV_BN_BWD_STATS (part_rows = disyolo_conv2d_stats_rows of that conv). */
size_t disyolo_bn_act_bwd_partials_workspace(int C);
int disyolo_bn_act_bwd_partials(const void* dy, const void* x, const float* scale, const float* shift,
                                const float* mean, const float* rstd, void* dx, float* dgamma,
                                float* dbeta, int64_t rows, int C, float alpha, const float* partials,
                                int part_rows, void* shortcut_grad, int shortcut_accumulate, void* workspace,
                                size_t workspace_bytes, void* stream);

/* ---- small data-movement ops of the backward pass ---- */
/* dst[b,y,x,c] = sum of the 2x2 block of src (gradient of resize_nearest_neighbor x2),
 * reading channels [c_off, c_off+C) of a src row of src_C channels; bf16. */
int disyolo_upsample2x_bwd(const void* src, void* dst, int B, int Hs, int Ws, int src_C, int c_off,
                           int C, int accumulate, void* stream);
/* column sums of a bf16 [rows,C] matrix -> f32 out[0:out_C] (bias gradient; out_C <= C) */
size_t disyolo_colsum_workspace(int64_t rows, int C);
int disyolo_colsum(const void* x, float* out, int64_t rows, int C, int out_C, void* workspace,
                   size_t workspace_bytes, void* stream);

/* ---- detection decode + filter (interpret_output / filter_detections,
 *      yolo/yolo3_net_pos.py:465-628,940-952) ---- */
/* logits: three f32 tensors [B,g,g,3,5+C] in the reference's scale order (S/8, S/16, S/32
 * grids).  anchors: host pointer to 18 floats (w,h)x9 in pixels.  Writes detections f32
 * [B,max_det,6] rows (y1,x1,y2,x2,classid,score), score-descending, zero padded, and
 * det_count int32 [B].  workspace: disyolo_detect_workspace bytes. */
size_t disyolo_detect_workspace(int B, int S, int num_class);
int disyolo_detect(const float* logits3, const float* logits2, const float* logits1, int B, int S,
                   int num_class, const float* anchors_host, const float* clip_window,
                   float obj_thresh, float nms_thresh, int max_det, float* detections,
                   int32_t* det_count, void* workspace, size_t workspace_bytes, void* stream);

/* ---- YOLO loss + gradient (loss_yolo, yolo/yolo3_net_pos.py:631-747) ---- */
/* labels: f32 [B,g,g,3,5+C] per scale (yolo3, yolo2, yolo1); true_boxes f32 [B,20,5].
 * dlogits: bf16 [B,g,g,32] per scale = d(loss)/d(logits) (already divided by B), rows padded
 * with zeros from 3*(5+C) to 32 channels (DISYOLO_GRAD_LD) so they feed the MFMA convs.
 * losses f32[8]: obj, noobj, class, xy, wh, (conf, coord, yolo_total). */
size_t disyolo_yolo_loss_workspace(int B, int S, int num_class);
int disyolo_yolo_loss(const float* const logits[3], const float* const labels[3],
                      const float* true_boxes, int max_boxes, int B, int S, int num_class,
                      const float* anchors_host, float ignore_thresh, const float scales[4],
                      void* const dlogits[3], float* losses, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- position-sensitive RoI assembly (yolo/yolo3_net_pos.py:750-938) ---- */
/* RoI selection for the mask loss (:757-796): detections [B,max_det,6], true_boxes [B,G,5],
 * perm_det int32 [B,max_det] / perm_gt int32 [B,G] replace tf.random_shuffle (:781-782).
 * Writes rois int32 [B,DISYOLO_ROI_MAX,DISYOLO_ROI_W] (positive RoIs first: the k+1 bin edges
 * per axis on the S/2 grid after tf.round, assigned GT row, pixel area) and roi_count [B]. */
int disyolo_mask_rois(const float* detections, int max_det, const float* true_boxes, int G,
                      const int32_t* perm_det, const int32_t* perm_gt, int B, int map_size,
                      int n_det, int n_gt, float iou_thresh, int32_t* rois, int32_t* roi_count,
                      void* stream);
/* tf.random_shuffle replacement (:781-782): uniformly random permutations perm_det int32
 * [B,n_det], perm_gt int32 [B,n_gt] from a counter-based hash of (seed, *step_counter, image);
 * step_counter (device int64, may be NULL) makes every replay of a recorded step reshuffle */
int disyolo_shuffle_perm(int32_t* perm_det, int n_det, int32_t* perm_gt, int n_gt, int B, uint32_t seed,
                         const int64_t* step_counter, void* stream);
/* masked BCE over assembled logits + gradient wrt the score maps (:799-858).
 * score f32 [B,Sm,Sm,k*k]; true_masks uint8 [B,G,2Sm,2Sm]; dscore bf16 [B,Sm,Sm,32] (padded);
 * loss f32[1] = mask_scale * mean_b(mean_r(sum BCE / area)). */
size_t disyolo_psroi_loss_workspace(int B, int map_size);
int disyolo_psroi_loss(const float* score, const uint8_t* true_masks, int G, const int32_t* rois,
                       const int32_t* roi_count, int B, int map_size, int k, float mask_scale,
                       void* dscore, float* loss, void* workspace, size_t workspace_bytes,
                       void* stream);
/* inference assembly (val_test, :862-938): masks f32 [B,max_det,Sm,Sm] = sigmoid(selected
 * channel) inside the box, 0.5 outside; keep int32 [B,max_det] = 1 for rows with h>0 and w>0. */
int disyolo_psroi_assemble(const float* score, const float* detections, int B, int max_det,
                           int map_size, int k, float* masks, int32_t* keep, void* stream);

/* evaluate()'s per-image mask post-processing (calculate_test_map.py:237-262), the caller side
 * of `evaluation`: for each of the image's n detections the crop [cy1:cy2, cx1:cx2] of its
 * size x size f32 mask (disyolo_psroi_assemble output) is resized to (y2-y1) x (x2-x1) with
 * cv2.resize(INTER_LINEAR) semantics, thresholded (> 0.5) and pasted at [y1:y2, x1:x2] of an
 * image_h x image_w mask.  rects: int32 [n][8] = cy1,cx1,cy2,cx2, y1,x1,y2,x2 (host side:
 * np.around(box*size) and correct_yolo_boxes, :121-138,244-251); a detection with an empty crop
 * or destination contributes nothing.  full_masks uint8 [n][image_h][image_w] or NULL; merged
 * uint8 [image_h][image_w] = class id + 1 of the LAST detection covering the pixel (:258-266). */
int disyolo_mask_paste(const float* masks, int n, int size, const int32_t* rects,
                       const int32_t* classids, int image_h, int image_w, uint8_t* full_masks,
                       uint8_t* merged, void* stream);
/* image_read of the test / validation drivers (calculate_test_map.py:149-176; utils/val_data.py:36-63):
 * rgb uint8 [image_h, image_w, 3] (device) -> out f32 [size, size, 3] (device): aspect-preserving
 * bilinear resize (cv2.resize INTER_LINEAR on the float32 image) centred in the letter box, padding
 * 127, / 255.  window_host (host, may be NULL) receives the clip window [top, left, bottom, right]. */
int disyolo_letterbox(const uint8_t* rgb, int image_h, int image_w, float* out, int size,
                      float* window_host, void* stream);
/* semantic-segmentation accuracy of evaluate (calculate_test_map.py:303-346): adds the 4x4 pixel
 * confusion counts of two uint8 class maps (0 = background, 1..3 = classes; other values are
 * ignored) to conf int64 [16], conf[true*4 + pred]; the caller zeroes conf before the first image
 * and forms IoU_c = n_cc / (row_c + column_c - n_cc) at the end. */
int disyolo_confusion16(const uint8_t* true_map, const uint8_t* pred_map, int64_t n, int64_t* conf,
                        void* stream);

/* ---- optimizer (tf.train.AdamOptimizer.minimize, train_yolo3_mask.py:55) ---- */
/* TF-form Adam on a flat f32 arena; elements [0, n_decay) also receive the gradient of the
 * l2 regulariser (grad += l2*w; yolo/yolo3_net_pos.py:38).  step t >= 1. */
int disyolo_adam_step(float* w, const float* grad, float* m, float* v, int64_t n, int64_t n_decay,
                      float lr, float beta1, float beta2, float eps, float l2, int64_t t,
                      float grad_scale, void* stream);
/* same with the step count kept in device memory: uses t = *step_counter + 1 and then
 * increments the counter (nothing about the step lives on the host: replayable) */
int disyolo_adam_step_dev(float* w, const float* grad, float* m, float* v, int64_t n,
                          int64_t n_decay, float lr, float beta1, float beta2, float eps, float l2,
                          int64_t* step_counter, float grad_scale, void* stream);
/* same, with the learning rate read from device memory too (lr_dev f32[1]: a recorded step follows
 * the schedule of train_yolo3_mask.py:130-141 by rewriting one float), and -- when reg_loss_out is
 * not NULL -- the value of the l2 term of total_loss (yolo/yolo3_net_pos.py:38,61),
 * 0.5*l2*sum(w[0:n_decay]^2) of the weights BEFORE the update, produced by the same sweep
 * (workspace: disyolo_adam_fused_workspace(n) bytes; deterministic fixed-order partial sums) */
size_t disyolo_adam_fused_workspace(int64_t n);
int disyolo_adam_step_fused(float* w, const float* grad, float* m, float* v, int64_t n,
                            int64_t n_decay, const float* lr_dev, float beta1, float beta2, float eps,
                            float l2, int64_t* step_counter, float grad_scale, float* reg_loss_out,
                            void* workspace, size_t workspace_bytes, void* stream);
/* The same update as sweeps over slices of the variables + one finish (train_yolo3_mask.py:55 is one
 * op over all variables; the slices only choose WHEN each part of it runs): a recorded step sweeps a
 * slice as soon as its gradients are final, beside the rest of the backward pass.  Every sweep uses
 * t = *step_counter + 1 and writes disyolo_adam_sweep_parts(n) l2 partial sums to `parts` (NULL: none;
 * n_decay = how many leading elements of THIS slice are regularised); the finish sums the nparts
 * partials of all sweeps in order into reg_loss_out (NULL: skip) and increments the counter. */
int disyolo_adam_sweep_parts(int64_t n);
int disyolo_adam_sweep(float* w, const float* grad, float* m, float* v, int64_t n, int64_t n_decay,
                       const float* lr_dev, float beta1, float beta2, float eps, float l2,
                       const int64_t* step_counter, float grad_scale, float* parts, void* stream);
int disyolo_adam_finish(int64_t* step_counter, const float* parts, int nparts, float l2,
                        float* reg_loss_out, void* stream);
/* the finish that also files the step's total loss (get_total_loss, yolo/yolo3_net_pos.py:61: the four YOLO terms
 * [losses8[7]] + the mask term + the l2 term) into ring[t % ring_len], t = the counter before its increment.  The
 * reference fetches total_loss with every sess.run (train_yolo3_mask.py:216) and only accumulates it (:218); a host that
 * reads the ring every ring_len steps or less gets the same values without joining the device once per step.
 * reg_loss_in: the l2 term when this finish sums none (reg_loss_out NULL); NULL = 0. */
int disyolo_adam_finish_record(int64_t* step_counter, const float* parts, int nparts, float l2, float* reg_loss_out,
                               const float* losses8, const float* mask_loss, const float* reg_loss_in, float* ring,
                               int ring_len, void* stream);
/* 0.5*l2*sum(w[0:n]^2) -> out f32[1] (only needed when the loss value is logged) */
size_t disyolo_l2_workspace(int64_t n);
int disyolo_l2_loss(const float* w, int64_t n, float l2, float* out, void* workspace,
                    size_t workspace_bytes, void* stream);

/* dst (+)= src over n bf16 elements (n % 8 == 0): gradient accumulation at the residual
 * shortcuts (yolo/yolo3_net_pos.py:150) */
int disyolo_add_bf16(const void* src, void* dst, int64_t n, int accumulate, void* stream);

/* ---- training-data pipeline (utils/train_data.py:44-276, 321-531): the pixel work of defect_train.get() ----
 * polygon_mask: one annotated instance = npoly polygons (vertices px/py f32, polygon k = [poly_start[k],
 * poly_start[k+1]), poly_is_out[k] = 1 'out' / 0 'in' = hole) -> uint8 mask [image_h, image_w]:
 * skimage.draw.polygon per polygon in order, holes cleared, vertex pixels set (load_mask, :321-338). */
int disyolo_polygon_mask(const float* px, const float* py, const int32_t* poly_start,
                         const int32_t* poly_is_out, int npoly, int image_h, int image_w, uint8_t* mask,
                         void* stream);
/* apply_random_scale_and_crop + flip (:392-397, 428-434, 446-464): src resized to new_w x new_h (cv2
 * INTER_LINEAR), placed with its corner at (dx, dy) of the size x size frame (negative = cropped),
 * flip 1 none / 2 horizontal / 3 vertical.  is_mask = 0: src uint8 RGB [H,W,3], pad 127, dst uint8
 * [size,size,3]; is_mask = 1: src uint8 0/1 [H,W] resized as float32, pad 0, np.around, dst uint8 0/1. */
int disyolo_aug_place(const uint8_t* src, int is_mask, int image_h, int image_w, uint8_t* dst, int size,
                      int new_w, int new_h, int dx, int dy, int flip, void* stream);
/* a whole batch's placements in one launch (round 6).  jobs: DEVICE array of njobs entries; per-pixel arithmetic = disyolo_aug_place's
 * (image: OpenCV's 8-bit fixed-point bilinear path, pad 127; mask: float bilinear, pad 0, half-to-even rounding to 0 / 1).
 * Replaces nothing in the reference (its loader is host code, utils/train_data.py:376-444); it replaces ~25 launches per batch here. */
typedef struct disyolo_place_job {
  const uint8_t* src;      /* image [image_h, image_w, 3] or mask [image_h, image_w], uint8 */
  uint8_t* dst;            /* [size, size, 3] or [size, size] */
  int32_t is_mask, image_h, image_w, new_w, new_h, dx, dy, flip;
} disyolo_place_job;
int disyolo_aug_place_batch(const disyolo_place_job* jobs, int njobs, int size, void* stream);
/* add_salt_pepper_noise (:511-525): pixels (rows[i], cols[i]) <- 1 for i < nsalt, then <- 0 for the next npepper */
int disyolo_aug_salt_pepper(uint8_t* image, int size, const int32_t* rows, const int32_t* cols, int nsalt,
                            int npepper, void* stream);
/* change_light (:527-535): L of HLS scaled by coeff, clipped */
int disyolo_aug_change_light(uint8_t* image, int size, double coeff, void* stream);
/* linearmotion_blur3C (:466-494), line length 3: angle 0/45/90/135, line_type 0 full / 1 right / 2 left */
int disyolo_aug_motion_blur3(const uint8_t* src, uint8_t* dst, int size, int angle, int line_type, void* stream);
/* image.astype(float32) / 255.0 (:411-413) */
int disyolo_aug_to_float(const uint8_t* image, float* out, int64_t n, void* stream);

/* ---- host helper of the checkpoint reader / writer (dis-yolo_amd/checkpoint.py; train_yolo3_mask.py:58,
 * 221-226 Saver.save / calculate_test_map.py:184-185 Saver.restore): CRC-32C (Castagnoli) of n bytes,
 * continuing from `crc` (0 to start) -- TensorFlow tensor bundles store it masked per tensor and per
 * index block.  Runs on the host; no device work. */
uint32_t disyolo_crc32c(const void* data, size_t n, uint32_t crc);

/* ---- command lists: the native step executor ----
 * Between cmdlist_begin and cmdlist_end every launch entry point above, called from the
 * recording thread, appends itself to the list (arguments captured by value; device buffers
 * and workspaces must outlive the list) instead of launching.  cmdlist_run replays commands
 * [first,last) on `stream` in one call -- the replacement for tf.Session.run's executor on
 * this path (train_yolo3_mask.py:216), and capturable into a hipGraph. */
void* disyolo_cmdlist_create(void);
void disyolo_cmdlist_destroy(void* list);
int disyolo_cmdlist_begin(void* list);
int disyolo_cmdlist_end(void);
int disyolo_cmdlist_size(void* list);
/* packets a replay puts on `lane`: what = 0 launches, 1 event records, 2 event waits (an event packet costs its stream ~3 us) */
int disyolo_cmdlist_count(void* list, int what, int lane);
/* four lanes: 0 = the stream passed to cmdlist_run, 1..3 = side streams owned by the list (3 = the gradient exchange).
 * set_lane selects the lane of the launches recorded next; sync(from,to) makes lane `to` wait
 * for what lane `from` has recorded so far.  Both are no-ops outside a recording.  Every
 * cmdlist_run range forks the side lane after the caller's stream and joins it at the end. */
int disyolo_cmdlist_set_lane(int lane);
int disyolo_cmdlist_sync(int from_lane, int to_lane);
/* finer edge: mark(lane) remembers this point of `lane` (returns the mark's id, -1 outside a recording);
 * wait(mark, lane) makes `lane` wait for that point only -- not for what the marked lane recorded later */
int disyolo_cmdlist_mark(int lane);
int disyolo_cmdlist_wait(int mark, int lane);
/* named marks that outlive a replay: wait_slot waits for the point the slot was LAST marked at -- earlier in this replay
 * or in the previous replay of the list (no wait on the first).  Lets a step's side-lane tail overlap the next step's
 * main-lane start with per-tensor dependencies (YOLONet.build_program(overlap_tail=True)).  slot 0..15. */
int disyolo_cmdlist_mark_slot(int lane, int slot);
int disyolo_cmdlist_wait_slot(int slot, int lane);
int disyolo_cmdlist_run(void* list, int first, int last, void* stream);
/* same with explicit fork/join control (flags bit 0 = fork at the start, bit 1 = join at the
 * end) for a step replayed in several ranges, and the side lane's hipStream_t so a caller can
 * order other work (an RCCL all-reduce) after the side lane only */
int disyolo_cmdlist_run_ex(void* list, int first, int last, void* stream, int flags);
void* disyolo_cmdlist_side_stream(void* list);
void* disyolo_cmdlist_lane_stream(void* list, int lane);   /* lane 1..3 (side_stream = lane 1) */
/* the side lanes are streams of one process-wide pool, created when a list first uses them; lanes_reserve(mask) (bit i =
 * lane i) creates them NOW on the current device -- before torch.distributed's NCCL backend (whose stream pool takes
 * the hardware queues: a lane created afterwards may share the caller's stream's queue, 2.5x the step time) */
int disyolo_lanes_reserve(int mask);
/* how the side lanes of the current device were chosen, one text line per lane that exists (priority, probed or not,
 * candidates tried, whether the fallback was taken, the probe's times): bench.py puts it into its line so that a run that
 * fell back to an unmeasured stream can be told from one that did not */
int disyolo_lanes_report(char* buf, int size);

/* ---- data-parallel gradient exchange as COMMANDS of the step (csrc/comm.hip) ----
 * The reference trains on one GPU (yolo/config.py:18; the op being distributed is train_yolo3_mask.py:55-56,
 * AdamOptimizer.minimize(total_loss)): every loss term is a per-image sum averaged over the batch
 * (yolo/yolo3_net_pos.py:692-726,858), so N ranks with equal local batches exchange ONE thing per step -- the sum of
 * their gradients.  These entry points put that exchange on RCCL (xGMI) without leaving the command list: inside a
 * recording a collective is a command of the current lane like any launch.
 * comm_load: dlopen RCCL (path NULL = default search; the host binding passes the copy the process already maps);
 *   no link-time dependency.  comm_unique_id (rank 0) -> 128 opaque bytes, exchanged by the host out of band ->
 *   comm_init on every rank (a collective; the current HIP device is the rank's GPU) -> an opaque communicator.
 * dtype: 0 = f32, 1 = bf16, 2 = f64.  Every rank must issue the same collectives in the same order. */
int disyolo_comm_load(const char* rccl_path, int* version);
int disyolo_comm_unique_id(void* id128);
int disyolo_comm_init(const void* id128, int rank, int nranks, void** comm);
int disyolo_comm_destroy(void* comm);
int disyolo_comm_allreduce_sum(void* comm, void* buf, int64_t count, int dtype, void* stream);
int disyolo_comm_reduce_scatter_sum(void* comm, const void* send, void* recv, int64_t recvcount, int dtype, void* stream);
int disyolo_comm_all_gather(void* comm, const void* send, void* recv, int64_t sendcount, int dtype, void* stream);
/* wire-format conversions of a gradient bucket (bf16 on the links, round to nearest even): any n > 0 */
int disyolo_cast_f32_bf16(const void* src_f32, void* dst_bf16, int64_t n, void* stream);
int disyolo_cast_bf16_f32(const void* src_bf16, void* dst_f32, int64_t n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* DISYOLO_H */
