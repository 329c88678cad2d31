/* icl_hip.h — C ABI of libicl_hip.so, the gfx950 (MI355X) kernels behind the ICL hot path.
 *
 * The reference (zhuye98/ICL) is pure PyTorch and has no FFI of its own; each entry point below
 * replaces the ATen operator the reference dispatches at the cited call site
 * (paths relative to /root/reference/code).  INTEGRATION.md shows the ctypes binding.
 *
 * Conventions
 *  - every pointer is a DEVICE pointer to fp32 (unless typed otherwise) owned by the caller;
 *    the library allocates nothing and keeps no state; `ws` arguments are caller-provided scratch
 *    whose size comes from the matching *_ws_bytes() query;
 *  - tensors are dense NC[D]HW unless a batch stride (in elements) is given;
 *  - `stream` is a hipStream_t; launches are asynchronous on it, no device synchronisation inside;
 *  - return 0 on success, a negative code on error (-1 bad argument, -2 launch failure);
 *    icl_last_error() returns a thread-local message.  Nothing throws across the boundary.
 */
#ifndef ICL_HIP_H
#define ICL_HIP_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

int icl_abi_version(void);
const char* icl_last_error(void);
/* name of the convolution kernel instantiation picked by the most recent icl_conv3d_* call on this thread (profiling aid) */
const char* icl_last_kernel_name(void);

/* ---- Conv3d k=3 pad=1 / k=1, stride 1 (networks/utils.py:104,107; networks/unet_3D_icl.py:65,178,196,327).
 * Weights are used in packed form Wp[taps][KP][NP]: mode 0 (forward) K=Cin,N=Cout; mode 1 (dgrad) K=Cout,N=Cin
 * with taps flipped.  KP = round_up(K,4), NP = round_up(N,16). */
int64_t icl_conv3d_packed_elems(int cout, int cin, int ks, int mode);
int icl_conv3d_pack_weights(const float* w, float* wp, int cout, int cin, int ks, int mode, void* stream);
/* both packings (mode 0 into wp_fwd, mode 1 into wp_dgrad) in one launch — a training step needs both of every weight */
int icl_conv3d_pack_weights_both(const float* w, float* wp_fwd, float* wp_dgrad, int cout, int cin, int ks, void* stream);
/* Both packings of `count` weights in one launch per 32 weights (arrays of device pointers and of Cout / Cin / kernel size): the
 * trainer packs every convolution weight of the model once per step (they only change in the optimiser), not once per call. */
int icl_conv3d_pack_weights_multi(const void* const* w, void* const* wp_fwd, void* const* wp_dgrad, const int32_t* cout, const int32_t* cin,
                                  const int32_t* ks, int count, void* stream);
/* y[n, 0:cout] = conv(x[n, 0:cin], Wp) + bias (bias may be NULL).  dgrad: call with Wp packed in mode 1,
 * x = dY, cin/cout swapped, bias NULL.  Batch strides in elements; channel stride is D*H*W.
 * ws: icl_conv3d_fwd_ws_bytes(...) bytes (0 for most shapes): launches with too few output tiles to fill the chip
 * split their Cin range over workgroups, each split writes a partial-output slab there and a fixed-order reduction adds
 * them (bitwise reproducible); ws may be NULL, the launch then runs unsplit.
 * Arithmetic: fp32 operands, fp32 accumulation, fp32 result.  3x3x3 layers with Cin % 16 == 0, W % 4 == 0 and >= 24^3 voxels
 * form their products on the bf16 matrix pipe from exact three-way bf16 splits of both operands (six MFMA terms per product,
 * csrc/kernels/conv_bf16x3.h).  Accuracy, measured against fp64: the OUTPUTS are as close as those of
 * v_mfma_f32_16x16x4_f32 (mean |error| 2.3e-7 vs 3.2e-7 relative), but v_mfma_f32_16x16x32_bf16 does not round its 32-product
 * partial sums to nearest: sums of same-sign products carry a coherent offset of -0.36 * 2^-24 (relative) that the fp32 kernels do
 * not have.  Logits, maps, losses, gradient norms and every compared gradient are unaffected: on the most cancellation-heavy tensor of the
 * step (the sampled 13,824^2 mlp2 gradient) the two paths differ by 1.2e-3 while the quantity itself is only defined to 1.6e-2 (an input
 * perturbation of 1e-7 moves it that far on the fp32 kernels; round 4, profiles/r4_mlp2_grad_sensitivity.txt).  The split weights live in
 * ws (counted by icl_conv3d_fwd_ws_bytes; without ws the fp32 MFMA kernels run).  Environment: ICL_CONV_SPLIT=0 selects the fp32
 * MFMA kernels for every shape. */
int64_t icl_conv3d_fwd_ws_bytes(int n, int cin, int cout, int d, int h, int w, int ks);
/* Split-product convolution with the weight split hoisted out of the call (the weights only change in the optimiser step):
 * icl_conv3d_split_weights_multi turns up to `count` packed weights wp[i] (icl_conv3d_pack_weights, mode 0 or 1; Cin % 16 == 0) into
 * their three bf16 planes wsplit[i] (icl_conv3d_split_ws_bytes(cin, cout) bytes each, 16-byte aligned; 0 = not eligible) in ONE
 * launch; icl_conv3d_fwd_presplit is icl_conv3d_fwd on such planes.  It returns 1 — nothing launched — when the shape does not run on
 * the split-product kernel (volume below 24^3, W % 4 != 0, ICL_CONV_SPLIT=0): the caller then uses icl_conv3d_fwd with the fp32 pack. */
int64_t icl_conv3d_split_ws_bytes(int cin, int cout);
int icl_conv3d_split_weights_multi(const void* const* wp, void* const* wsplit, const int32_t* cin, const int32_t* cout, int count,
                                   void* stream);
int icl_conv3d_fwd_presplit(const float* x, const void* wsplit, const float* bias, float* y, int n, int cin, int cout, int d, int h, int w,
                            int64_t x_bstride, int64_t y_bstride, void* stream);
/* Round 6: the deep levels (rows of 12 or 24 voxels: 18 / 72 output tiles for a batch of two) fill the chip by splitting the channel
 * chunks of a tile over workgroups; every slice writes raw partial sums into a slab of ws and a fixed-order reduction adds them (+ bias)
 * into y — bitwise reproducible.  icl_conv3d_fwd_presplit_ws_bytes: the bytes of ws the launcher's plan needs for this shape (0: the
 * launch is not split; icl_conv3d_fwd_presplit does the same work).  icl_conv3d_fwd_presplit_ws = icl_conv3d_fwd_presplit with that
 * workspace; the 12^3 level (4 x 4 x 12 tiles, Cout % 32 == 0) runs on the split-product kernel only through it.  A split launch hands no
 * statistics to the normalisation (icl_conv3d_fwd_stats_slots returns 0 for such shapes).  Reference: nn.Conv3d of UnetConv3 at the
 * conv3 / conv4 / up_concat4 / up_concat3 levels (/root/reference/code/networks/unet_3D.py:41-47,53-54, networks/utils.py:104,107) and
 * its input gradient.  Environment: ICL_CONV_SPLIT_KSPLIT=0 never splits (the 12^3 level then stays on the fp32 MFMA kernels). */
int64_t icl_conv3d_fwd_presplit_ws_bytes(int n, int cin, int cout, int d, int h, int w);
int icl_conv3d_fwd_presplit_ws(const float* x, const void* wsplit, const float* bias, float* y, void* ws, int n, int cin, int cout, int d,
                               int h, int w, int64_t x_bstride, int64_t y_bstride, void* stream);
/* Convolution + the InstanceNorm statistics of its output from the epilogue's registers (reference: Conv3d -> InstanceNorm3d,
 * /root/reference/code/networks/utils.py:104-105, :107-108) — the stand-alone statistics pass of icl_norm_fwd re-reads the whole output.
 * icl_conv3d_fwd_stats_slots: the number of (count, mean, M2) summaries per (sample, channel) the launch will write (= its workgroup
 * count), or 0 when the shape cannot produce them (not on the split-product kernel, or a workgroup's tile list would span two samples):
 * the caller then runs icl_conv3d_fwd_presplit + icl_norm_fwd.  stats: n * cout * slots * 3 floats, every element written.
 * icl_conv3d_fwd_presplit_stats returns 1 — nothing launched — under the same conditions as icl_conv3d_fwd_presplit. */
int icl_conv3d_fwd_stats_slots(int n, int cin, int cout, int d, int h, int w);
int icl_conv3d_fwd_presplit_stats(const float* x, const void* wsplit, const float* bias, float* y, float* stats, int n, int cin, int cout,
                                  int d, int h, int w, int64_t x_bstride, int64_t y_bstride, void* stream);
int icl_conv3d_fwd(const float* x, const float* wp, const float* bias, float* y, void* ws, int n, int cin, int cout, int d, int h,
                   int w, int ks, int64_t x_bstride, int64_t y_bstride, void* stream);
/* gw[cout][cin][taps] = sum over batch and voxels; ws >= icl_conv3d_wgrad_ws_bytes(n,cin,cout,ks) bytes (one packed
 * partial-sum slab per workgroup row, reduced in a fixed order: bitwise reproducible).  gbias (may be NULL) [cout] = sum of gy. */
int64_t icl_conv3d_wgrad_ws_bytes(int n, int cin, int cout, int ks);
int icl_conv3d_wgrad(const float* x, const float* gy, float* gw, float* gbias, void* ws, int n, int cin, int cout, int d,
                     int h, int w, int ks, int64_t x_bstride, int64_t gy_bstride, void* stream);
/* The two halves of icl_conv3d_wgrad apart (round 6): `_slabs` runs the weight-gradient kernel only and leaves its per-workgroup partial
 * sums in ws (*nslabs of them); `_reduce_multi` sums the slabs of up to any number of such calls (HOST arrays of `count` entries) into
 * their [Cout, Cin, k, k, k] gradients in ONE launch, each in the order icl_conv3d_wgrad uses (bit-identical).  The weight gradient is
 * a leaf of the backward pass (only the optimiser reads it, utils.py:104-107 / trainer :113-115), so a step scope queues the sums and
 * runs them once at the end of backward (ops.DeferredWgradReduce) instead of one small launch behind each of 23 weight-gradient kernels. */
int icl_conv3d_wgrad_slabs(const float* x, const float* gy, void* ws, int n, int cin, int cout, int d, int h, int w, int ks,
                           int64_t x_bstride, int64_t gy_bstride, int32_t* nslabs, void* stream);
int icl_conv3d_wgrad_reduce_multi(const void* const* ws, void* const* gw, const int32_t* cout, const int32_t* cin, const int32_t* ks,
                                  const int32_t* nslabs, int count, void* stream);

/* ---- InstanceNorm3d(+ReLU) (networks/utils.py:105-106,108-109) and BatchNorm3d(+ReLU)
 * (networks/unet_3D_icl.py:325-340).  mode 0 = instance (group = (n,c)), 1 = batch (group = c).
 * use_batch_stats 0 = normalise with the given mean/rstd... (eval-mode BatchNorm: mean=running_mean,
 * rstd computed from running_var by icl_rstd_from_var).  act 0 none / 1 ReLU / 2 LeakyReLU(0.01).  gamma/beta/running_* may be NULL. */
int64_t icl_norm_ws_bytes(int n, int c, int64_t s);
int icl_norm_fwd(const float* x, float* y, float* mean, float* rstd, const float* gamma, const float* beta,
                 float* running_mean, float* running_var, int n, int c, int64_t s, int mode, int use_batch_stats,
                 int act, float eps, float momentum, void* ws, void* stream);
int icl_norm_bwd(const float* gy, const float* x, const float* mean, const float* rstd, const float* gamma,
                 const float* beta, float* gx, float* dgamma, float* dbeta, int n, int c, int64_t s, int mode,
                 int use_batch_stats, int act, void* ws, void* stream);
/* Same, with a residual branch: y = act(norm(x) + res) and, backward, gres = act'(.) * gy next to gx.  This is the tail of
 * MONAI 1.0.1 UnetResBlock.forward ("out = norm2(conv2(.)); out += residual; out = lrelu(out)") used by every encoder/decoder
 * block of SwinUNETR (networks/swinunetr_icl.py:123-223).  res / gres may be NULL (then identical to icl_norm_fwd/_bwd). */
/* icl_norm_res_fwd with batch statistics taken from `part` — nslots (count, mean, M2) summaries per (sample, channel) row, layout
 * part[(row * nslots + slot) * 3], as icl_conv3d_fwd_presplit_stats writes them — instead of a statistics pass over x. */
int icl_norm_fwd_given_stats(const float* x, const float* res, float* y, float* mean, float* rstd, const float* gamma, const float* beta,
                             float* running_mean, float* running_var, int n, int c, int64_t s, int mode, int act, float eps,
                             float momentum, const float* part, int nslots, void* stream);
int icl_norm_res_fwd(const float* x, const float* res, float* y, float* mean, float* rstd, const float* gamma, const float* beta,
                     float* running_mean, float* running_var, int n, int c, int64_t s, int mode, int use_batch_stats,
                     int act, float eps, float momentum, void* ws, void* stream);
int icl_norm_res_bwd(const float* gy, const float* x, const float* res, const float* mean, const float* rstd, const float* gamma,
                     const float* beta, float* gx, float* gres, float* dgamma, float* dbeta, int n, int c, int64_t s, int mode,
                     int use_batch_stats, int act, void* ws, void* stream);
int icl_rstd_from_var(const float* var, float* rstd, int c, float eps, void* stream);

/* ---- Deferred InstanceNorm3d + ReLU (round 5).  Reference: Conv3d -> InstanceNorm3d -> ReLU in UnetConv3
 * (/root/reference/code/networks/utils.py:107-109) whose output feeds MaxPool3d / the skip concatenation / nn.Upsample / the `final`
 * convolution (unet_3D_icl.py:41-53,116-117, utils.py:264,276).  The convolution stores its RAW output and (count, mean, M2) summaries
 * (icl_conv3d_fwd_presplit_stats); icl_norm_finalize_stats turns them into mean / rstd (for icl_norm_bwd) and the per-(sample, channel)
 * pair ss = (scale, shift) = (rstd, -mean * rstd); the consumers below apply relu(fma(x, scale, shift)) while they load x, so the
 * normalised tensor is never written.  ss: [n * c][2] floats. */
int icl_norm_finalize_stats(const float* part, int n, int c, int nslots, float eps, float* mean, float* rstd, float* ss, void* stream);
/* y = relu(fma(x, scale, shift)) written out (a consumer that cannot apply the deferred normalisation on load); s % 4 == 0 */
int icl_norm_apply(const float* x, const float* ss, float* y, int n, int c, int64_t s, void* stream);
/* MaxPool3d(2) of the normalised tensor from the raw one (values and argmax of relu(fma(x, scale, shift))) */
int icl_maxpool2_fwd_norm(const float* x, const float* ss, float* y, uint8_t* idx, int64_t nc, int dout, int hout, int wout, int pool_depth,
                          void* stream);
/* out = cat([skip, upsample2x(deep)], 1) (utils.py:264,276): skip [n, cs, d, h, w], deep [n, cd, d/2, h/2, w/2], out [n, cs + cd, d, h, w];
 * skip_ss / deep_ss: that source is raw with deferred normalisation (NULL: used as stored).  d, h even, w % 4 == 0. */
int icl_upsample2x_concat_norm(const float* skip, const float* skip_ss, const float* deep, const float* deep_ss, float* out, int n, int cs,
                               int cd, int d, int h, int w, void* stream);

/* ---- MaxPool3d(2) (networks/unet_3D_icl.py:41-53) and MaxPool2d(2) (networks/unet_icl.py:64, pool_depth = 1 on a
 * D = 1 volume); idx = uint8 argmax within the pool_depth x 2 x 2 window. */
int icl_maxpool2_fwd(const float* x, float* y, uint8_t* idx, int64_t nc, int dout, int hout, int wout, int pool_depth, void* stream);
int icl_maxpool2_bwd(const float* gy, const uint8_t* idx, float* gx, int64_t nc, int dout, int hout, int wout, int pool_depth, void* stream);
/* gx = add + maxpool2 backward of gy: the gradient of a tensor that feeds both a max-pool and a skip connection
 * (/root/reference/code/networks/unet_3D_icl.py:100-116) in one pass.  add: [n][c] planes of the un-pooled extent, batches add_bstride
 * elements apart (a channel slice of the concat gradient). */
int icl_maxpool2_bwd_add(const float* gy, const uint8_t* idx, const float* add, float* gx, int n, int c, int dout, int hout, int wout,
                         int pool_depth, int64_t add_bstride, void* stream);

/* ---- trilinear / bilinear (D = 1) resize (networks/utils.py:264; networks/unet_icl.py:84-85; utils/losses.py:245,263,281,292).
 * align_corners=0: source scale in/out (what ATen uses for size= and for scale_factor=2); 1: (in-1)/(out-1). */
int icl_trilinear_fwd(const float* x, float* y, int n, int c, int di, int hi, int wi, int dout, int hout, int wout,
                      int64_t y_bstride, int align_corners, void* stream);
/* backward = three separable one-axis gathers (x, y, z); ws >= icl_trilinear_bwd_ws_bytes(...) */
int64_t icl_trilinear_bwd_ws_bytes(int n, int c, int di, int hi, int wi, int dout, int hout, int wout);
int icl_trilinear_bwd(const float* gy, float* gx, void* ws, int n, int c, int di, int hi, int wi, int dout, int hout, int wout,
                      int64_t gy_bstride, int align_corners, void* stream);

/* ---- strided row copy (torch.cat([skip, up], 1), networks/utils.py:276) */
int icl_copy_rows(const float* src, float* dst, int64_t rows, int64_t row_elems, int64_t src_stride, int64_t dst_stride,
                  void* stream);

/* ---- out = [a ; b] of na + nb floats (multiples of 4, 16-byte aligned); a NULL half is written as zeros.  The gradient of the batch
 * split of the ICL forward (train_inherent_consistent_unet_3D_BraTS.py:103-104 slices outputs[:labeled_bs] / [labeled_bs:]; the unlabeled
 * half of the features alone feeds uscl, unet_3D_icl.py:124-131) in one launch. */
int icl_concat2(const float* a, int64_t na, const float* b, int64_t nb, float* out, void* stream);

/* ---- depthwise Conv3d k=3 pad=1 groups=C, no bias (networks/unet_3D_icl.py:320-323).
 * w is [C][27]; flip=1 applies the transposed stencil (input gradient). gw [C][27] is overwritten (per-chunk partial sums in `ws`,
 * icl_dwconv3_wgrad_ws_bytes, added in a fixed order). */
int icl_dwconv3_fwd(const float* x, const float* w, float* y, int n, int c, int d, int h, int wd, int flip, void* stream);
int64_t icl_dwconv3_wgrad_ws_bytes(int n, int c, int d, int h, int wd);
int icl_dwconv3_wgrad(const float* x, const float* gy, float* gw, void* ws, int n, int c, int d, int h, int wd, void* stream);

/* ---- 1x1x1 Conv3d with a handful of channels on a small volume (SeparableConv3d.pointwise and attn_convs1 of the aligner,
 * networks/unet_3D_icl.py:196,327): y[b][o][v] = bias[o] + sum_i w[o * w_ostride + i * w_istride] * x[b][i][v], x [n, cin, s],
 * y [n, cout, s], s % 4 == 0, bias may be NULL.  The input gradient is the same call with the strides swapped. */
int icl_conv1x1_small(const float* x, const float* w, const float* bias, float* y, int n, int cin, int cout, int64_t s, int w_ostride,
                      int w_istride, void* stream);
/* the same with nn.Dropout(p) folded in (the mask of icl_dropout, same seed protocol; unet_3D_icl.py:67-68,117: `final(dropout2(up1))`):
 * mask_mode 1: the INPUT is the dropped tensor (forward); 2: the OUTPUT is (the pair's input gradient).  Returns 1 (nothing launched) for
 * n * s < 65536: run icl_dropout and icl_conv1x1_small then.  icl_conv1x1_wgrad_dropout: the pair's weight gradient from the un-dropped x. */
int icl_conv1x1_dropout(const float* x, const float* w, const float* bias, float* y, int n, int cin, int cout, int64_t s, int w_ostride,
                        int w_istride, int mask_mode, uint32_t seed, float p, const uint32_t* seed_dev, void* stream);
int icl_conv1x1_wgrad_dropout(const float* x, const float* gy, float* gw, float* gbias, void* ws, int n, int cin, int cout, int64_t s,
                              int64_t gy_bstride, uint32_t seed, float p, const uint32_t* seed_dev, void* stream);
/* the same pair with x given raw + ss (deferred InstanceNorm + ReLU of `up1`, then dropout2, then `final`; p = 0: no mask): forward
 * (n * s >= 65536) and weight gradient; the input gradient is icl_conv1x1_dropout(mask_mode 2) as before. */
int icl_conv1x1_dropout_norm(const float* x, const float* ss, const float* w, const float* bias, float* y, int n, int cin, int cout, int64_t s,
                             int w_ostride, int w_istride, uint32_t seed, float p, const uint32_t* seed_dev, void* stream);
int icl_conv1x1_wgrad_dropout_norm(const float* x, const float* ss, const float* gy, float* gw, float* gbias, void* ws, int n, int cin,
                                   int cout, int64_t s, int64_t gy_bstride, uint32_t seed, float p, const uint32_t* seed_dev, void* stream);

/* ---- nn.Dropout(p) (networks/unet_3D_icl.py:67-68,110,116): y = keep ? x/(1-p) : 0 with a counter-based
 * mask keyed by (seed, element index); calling it again with the same seed on dY is the backward.  seed_dev (may be
 * NULL) is a device-resident counter folded into the seed so that replays of a captured hipGraph draw fresh masks. */
int icl_dropout(const float* x, float* y, int64_t n, uint32_t seed, float p, const uint32_t* seed_dev, void* stream);
/* DropPath / stochastic depth (timm, MONAI; networks/unet_3D_icl.py:253, swinunet_icl.py:215): one keep/drop decision per `group`
 * consecutive elements (group = elements per sample), kept samples scaled by 1/(1-p).  Same counter-based mask as icl_dropout. */
int icl_drop_path(const float* x, float* y, int64_t n, int64_t group, uint32_t seed, float p, const uint32_t* seed_dev, void* stream);
/* y = res + DropPath(x): the residual update  x + drop_path(branch)  of the Class_Decoder / Swin blocks (unet_3D_icl.py:262-267,
 * swinunetr_icl.py:871-884) in one pass; res may be NULL (plain icl_drop_path) or alias x (x + drop_path(x), :263). */
int icl_drop_path_add(const float* x, const float* res, float* y, int64_t n, int64_t group, uint32_t seed, float p,
                      const uint32_t* seed_dev, void* stream);

/* ---- fused softmax + Dice / CE / soft-Dice / MSE reductions (utils/losses.py:22-59,68-90,200-231 and the
 * CrossEntropyLoss at train_inherent_consistent_unet_3D_BraTS.py:107).  a is [B,nc,S] logits (or probabilities when
 * a_is_prob); the target is int64 labels [B,S] (modes 0,1) or a second logit tensor b [B,nc,S] (modes 2,3):
 *   mode 0 hard Dice        out = {0, dice}          mode 1 CE + hard Dice   out = {ce, dice}
 *   mode 2 soft Dice        out = {0, dice}          mode 3 softmax MSE      out = {mse, 0}
 * stats: ICL_LOSS_STATS_FLOATS(nc) floats — the 3*nc+1 reduced sums (kept for backward) followed by per-workgroup partials,
 * summed in a fixed order (no atomics: the loss is bit-reproducible and nothing needs zeroing); gout0 / gout1: upstream grads of out[0] / out[1], one float each, NULL when
 * that output does not reach the loss; ga: gradient w.r.t. a.  weight (per-class Dice weights) may be NULL.  nc <= 16. */
#define ICL_LOSS_MAX_BLOCKS 1024
#define ICL_LOSS_STATS_FLOATS(nc) ((3 * (nc) + 1) * (1 + ICL_LOSS_MAX_BLOCKS))
int icl_loss_fwd(const float* a, const float* b, const int64_t* labels, const float* weight, float* stats, float* out,
                 int batch, int nc, int64_t s, int mode, int a_is_prob, void* stream);
int icl_loss_bwd(const float* a, const float* b, const int64_t* labels, const float* weight, const float* stats,
                 const float* gout0, const float* gout1, float* ga, int batch, int nc, int64_t s, int mode, int a_is_prob,
                 void* stream);
/* All loss terms of a step in one statistics launch + one finalize launch (icl_loss_fwd_multi) and one gradient launch
 * (icl_loss_bwd_multi): job j is call j of icl_loss_fwd / icl_loss_bwd — same blocks, same partial sums, same order (bit-identical).
 * The trainer's objective (train_inherent_consistent_unet_3D_BraTS.py:105-112) is ten such terms.  `jobs`: HOST array of <= 12;
 * per job the arguments of the single-term calls (stats: ICL_LOSS_STATS_FLOATS(nc) floats, out: 2 floats; gout0 / gout1 / ga: backward). */
typedef struct IclLossJob {
  const float* a; const float* b; const int64_t* labels; const float* weight;
  float* stats; float* out;
  const float* gout0; const float* gout1; float* ga;
  int64_t s;
  int batch, nc, mode, a_is_prob;
} IclLossJob;
int icl_loss_fwd_multi(const IclLossJob* jobs, int count, void* stream);
int icl_loss_bwd_multi(const IclLossJob* jobs, int count, void* stream);

/* ---- aligner token operators (networks/unet_3D_icl.py:244-315)
 * LayerNorm over the last axis c of [rows, c] (nn.LayerNorm, eps 1e-5); mean/rstd [rows] are saved for the backward;
 * dgamma/dbeta (may both be NULL) are overwritten: row-chunk partial sums go to `ws` (icl_layernorm_bwd_ws_bytes) and are added in
 * a fixed order (no float atomics).  dgamma NULL with ws given: only the partials are written — ws = part_g [chunks][c] followed by
 * part_b [chunks][c], chunks = ws bytes / (8 c) — and the caller sums them later (icl_colsum_multi; ICLTrainer does that for all
 * LayerNorm layers of a step in one launch).  GELU is the exact erf form (nn.GELU default). */
int icl_layernorm_fwd(const float* x, const float* gamma, const float* beta, float* y, float* mean, float* rstd, int64_t rows,
                      int c, float eps, void* stream);
int64_t icl_layernorm_bwd_ws_bytes(int64_t rows, int c);
int icl_layernorm_bwd(const float* gy, const float* x, const float* gamma, const float* mean, const float* rstd, float* gx,
                      float* dgamma, float* dbeta, void* ws, int64_t rows, int c, void* stream);
int icl_gelu_fwd(const float* x, float* y, int64_t n, void* stream);
int icl_gelu_bwd(const float* gy, const float* x, float* gx, int64_t n, void* stream);
/* Prototype cross-attention of Query_Attention.forward (:283-297): q [B,h,nc,d] (reshape-quirk layout of fc_q's output),
 * kv [B,N,2,h,d] (fc_kv output).  logits [B,nc,h,N] = scale*q.k (the pre-softmax map, in the class-major layout the reference returns it in: `attn1.permute(0, 2, 1, 3)`, :296),
 * out [B,h,nc,d] = softmax_N(logits) @ v, stats [B,h,nc,2] = (row max, sum exp).  d in {8,16}, nc <= 16.
 * Backward: gout / glog (either may be NULL) are the upstream gradients of out / logits. */
int icl_attn_fwd(const float* q, const float* kv, float* logits, float* out, float* stats, int b, int h, int nc, int n, int d,
                 float scale, void* stream);
int icl_attn_bwd(const float* q, const float* kv, const float* logits, const float* stats, const float* out, const float* gout,
                 const float* glog, float* gq, float* gkv, int b, int h, int nc, int n, int d, float scale, void* stream);
/* the same with a workspace (icl_attn_bwd_ws_bytes): dQ is summed from per-token-chunk shares written by the dK / dV pass instead of a second
 * walk over K and V by one workgroup per (sample, head, class) row; ws == NULL is icl_attn_bwd */
int64_t icl_attn_bwd_ws_bytes(int b, int h, int nc, int n, int d);
int icl_attn_bwd_ws(const float* q, const float* kv, const float* logits, const float* stats, const float* out, const float* gout,
                    const float* glog, float* gq, float* gkv, void* ws, int b, int h, int nc, int n, int d, float scale, void* stream);

/* out_i[c] = sum_r g_i[r * cols_i + c] for `count` small row-major matrices, one launch per 48 of them: the bias gradients of the
 * aligner / Swin Linear layers (unet_3D_icl.py:244-315) collected over a backward pass. */
int icl_colsum_multi(const void* const* g, void* const* out, const int32_t* rows, const int32_t* cols, int count, void* stream);
/* out[j] = sum_i w[j * n_in + i] * *in[i]: n_in <= 16 device scalars (pointers in a HOST array), n_out <= 16 results, w a HOST array.
 * The scalar arithmetic between the loss terms of an ICL step (weighted sums, means over three scales) in one launch; its backward is
 * the same call with the transposed weights. */
int icl_scalar_combine(const void* const* in, const float* w, int n_in, int n_out, float* out, void* stream);

/* ---- token <-> window order of the Swin blocks (pad + roll + window_partition, window_reverse + roll + crop:
 * networks/swinunetr_icl.py:825-866, networks/swinunet_icl.py:256-283): out[b][m][0..c) = idx[m] >= 0 ? src[b][idx[m]][0..c) : 0,
 * src [b, s, c], out [b, m, c], c % 4 == 0.  The inverse direction and both gradients are gathers with the inverse index. */
int icl_gather_rows(const float* src, const int32_t* idx, float* out, int64_t b, int64_t s, int64_t m, int c, void* stream);
/* out[b][m] = src[b][idx2[m][0]] + src[b][idx2[m][1]] (an index of -1 is absent).  Gradient of icl_gather_rows when the
 * index lists a source row up to twice or not at all: PatchMerging's eight strided slices repeat two and omit two
 * (networks/swinunetr_icl.py:953-961; the 2-D merge networks/swinunet_icl.py:320-326 is a plain permutation). */
int icl_gather_rows_sum2(const float* src, const int32_t* idx2, float* out, int64_t b, int64_t s, int64_t m, int c, void* stream);

/* ---- 3^3 convolutions on tiny volumes (<= 6^3 voxels, hundreds of channels: `encoder10`/`decoder5` of SwinUNETR,
 * networks/swinunetr_icl.py:163-183, and the U-Net `center`, networks/unet_3D_icl.py:54) are skinny GEMMs over the weights:
 * cols [n*S, c*27] with cols[b*S+v][ci*27+tap] = x[b][ci][v+offset(tap)] (zero padded), y = cols * W[cout, c*27]^T on the
 * library GEMM.  col2im3 is the exact transpose (input gradient from d(cols)). */
int icl_im2col3(const float* x, float* cols, int n, int c, int d, int h, int w, void* stream);
/* planes[b][ci*27 + tap][v] = x[b][ci][v + offset(tap)] (zero outside), w % 4 == 0: the first convolution's weight gradient
 * (Cin = 1: networks/unet_3D.py:27, utils.py:104) as icl_conv1x1_wgrad over 27 shifted planes instead of a Cin-padded implicit GEMM. */
int icl_im2col3_planes(const float* x, float* planes, int n, int c, int d, int h, int w, void* stream);

/* The same layer — nn.Conv3d(1, Cout <= 16, 3, padding 1), the first convolution of unet_3D(in_channels=1) (networks/unet_3D.py:27,
 * networks/utils.py:104) — as dedicated kernels (kernels/conv_cin1.h): K = 27, both directions are HBM streams of the Cout-channel tensor.
 * x [N][1][D][H][W] (sample stride x_bstride), w the nn.Conv3d weight [Cout][1][3][3][3], y / gy [N][Cout][D][H][W] (sample stride
 * y_bstride / gy_bstride), gw like w.  The weight gradient needs W % 4 == 0, gy 16-byte aligned and a workspace of
 * icl_conv3d_cin1_wgrad_ws_bytes bytes; partial sums are added in a fixed order. */
int icl_conv3d_cin1_fwd(const float* x, const float* w, const float* bias, float* y, int n, int cout, int d, int h, int wd, int64_t x_bstride,
                        int64_t y_bstride, void* stream);
int64_t icl_conv3d_cin1_wgrad_ws_bytes(int n, int d, int h);
int icl_conv3d_cin1_wgrad(const float* x, const float* gy, float* gw, void* ws, int n, int cout, int d, int h, int wd, int64_t x_bstride,
                          int64_t gy_bstride, void* stream);
int icl_col2im3(const float* g, float* dx, int n, int c, int d, int h, int w, void* stream);

/* ---- ConvTranspose3d(kernel 2, stride 2, no bias) = GEMM + depth-to-space (MONAI UnetrUpBlock.transp_conv in SwinUNETR,
 * networks/swinunetr_icl.py:186-230): yt [n, d*h*w, cout*8] (column co*8 + i*4 + j*2 + k) -> out[b][co][2z+i][2y+j][2x+k];
 * out_bstride lets the result land in the channel slice of the concat buffer that UnetrUpBlock builds next.
 * icl_space_to_depth2 is the inverse (gradient back to the GEMM layout). */
int icl_depth_to_space2(const float* yt, float* out, int n, int d, int h, int w, int cout, int64_t out_bstride, void* stream);
int icl_space_to_depth2(const float* g, float* gt, int n, int d, int h, int w, int cout, int64_t g_bstride, void* stream);

/* ---- nn.Linear forward / input gradient and the other dense fp32 products of the path, on the fp32 matrix cores
 * (csrc/kernels/gemm.h).  Replaces F.linear at networks/unet_3D_icl.py:283-315 (fc_q, fc_kv, proj, MLP.fc1/fc2 and the token-axis
 * Class_Decoder.mlp2, :258-268), query_convs (:197), the token projection (:212), the Swin qkv / proj / MLP / PatchMerging linears
 * (networks/swinunetr_icl.py:703-750,812,946-976) and their autograd input gradients.
 *   icl_linear_fwd    y[rows, out] = act(x[rows, in] W[out, in]^T + bias)     act 0 none, 1 exact-erf GELU (MLP.act, :307); bias may be NULL
 *   icl_linear_dgrad  gx[rows, in] = gy[rows, out] W[out, in]
 *   icl_linear_wgrad_small  dw[out, in] = gy^T x for few rows (tall inputs: icl_linear_wgrad above)
 * rows <= 32 with >= 2^20 weights (the 13,824^2 / 1,728^2 mlp2 matrices): the weight matrix is streamed from HBM exactly once
 * (HBM-bound); everything else runs on LDS-staged 64x64 wave tiles, split over K when the output has too few tiles to fill the chip.
 * ws: icl_linear_ws_bytes(rows, in, out, kind) bytes, kind 0 fwd / 1 dgrad / 2 wgrad_small / 3 dgrad_sgd (may be 0; NULL then allowed). */
int64_t icl_linear_ws_bytes(int64_t rows, int in, int out, int kind);
int icl_linear_fwd(const float* x, const float* w, const float* bias, float* y, void* ws, int64_t rows, int in, int out, int act,
                   void* stream);
int icl_linear_dgrad(const float* gy, const float* w, float* gx, void* ws, int64_t rows, int in, int out, void* stream);
/* gx = gy W AND dW = gy^T x of a Linear layer that sees <= 32 rows (the aligner's query-side layers, unet_3D_icl.py:283-315) in one launch;
 * returns 1 (nothing launched) when the shapes do not take the weight-streaming path: call icl_linear_dgrad + icl_linear_wgrad_small then */
int icl_linear_bwd_small(const float* gy, const float* w, const float* x, float* gx, float* dw, void* ws, int64_t rows, int in, int out,
                         void* stream);
int icl_linear_wgrad_small(const float* gy, const float* x, float* dw, void* ws, int64_t rows, int in, int out, void* stream);
/* Backward + optimiser step of a big skinny Linear in ONE pass over its weight (single rank, the weight used once per step):
 * gx[rows, in] = gy[rows, out] W_old, then W / momentum are updated from the factors of dW = gy^T x with the SGD rule of
 * icl_sgd_step_factored (train_inherent_consistent_unet_3D_BraTS.py:85-86,113-115 on Class_Decoder.mlp2, unet_3D_icl.py:258-268).
 * rows <= 32, in / out multiples of 4; ws: icl_linear_ws_bytes(rows, in, out, 3) bytes. */
int icl_linear_dgrad_sgd(const float* gy, const float* x, float* w, float* mom, float* gx, void* ws, int64_t rows, int in, int out,
                         float lr, float momentum, float weight_decay, int first, const float* lr_dev, void* stream);
/* General (batched) product C[b] = act(A[b] B[b] + bias[n]), C [m, n] with row pitch ldc; act = activation (0 none, 1 GELU)
 * + 16 when the bias is indexed by the output row m instead of the column (1x1x1 convolution: rows are channels).  a_kcontig 1: A is row-major [m][lda]
 * (k contiguous), 0: A is given transposed [k][lda] (m contiguous); b_kcontig 1: B is [n][ldb] (k contiguous, "W^T"), 0: B is
 * [k][ldb].  Batch strides in elements.  Used for ConvTranspose3d(k=2, s=2) as one product (MONAI UnetrUpBlock.transp_conv,
 * networks/swinunetr_icl.py:178-225: A = x[b] as [Cin][S] transposed view, B = W [Cin][Cout*8]) and its gradients, and for the
 * 3^3 convolutions on <= 6^3 voxels through im2col (networks/utils.py:104,107 at the `center` level).
 * ws: icl_gemm_ws_bytes(m, n, k, batch) bytes (split-K slabs, summed in a fixed order); with ws NULL the product runs unsplit. */
int64_t icl_gemm_ws_bytes(int64_t m, int n, int k, int batch);
int icl_gemm(const float* a, const float* b, float* c, const float* bias, void* ws, int64_t m, int n, int k, int64_t lda, int64_t ldb,
             int64_t ldc, int a_kcontig, int b_kcontig, int act, int batch, int64_t a_bstride, int64_t b_bstride, int64_t c_bstride,
             void* stream);

/* ---- nn.Linear weight/bias gradient for tall token matrices (qkv / proj / MLPBlock linears of the Swin stages,
 * networks/swinunetr_icl.py:703,705,812; PatchEmbed; the k2s2 transposed convolutions written as GEMMs):
 * dw[o][i] = sum_r gy[r][o] * x[r][i], db[o] = sum_r gy[r][o] (db may be NULL); gy [rows, o], x [rows, i] row-major.
 * Rows are split over many waves (fp32 MFMA), partial blocks go to ws (icl_linear_wgrad_ws_bytes) and are summed in a fixed order. */
int64_t icl_linear_wgrad_ws_bytes(int64_t rows, int o, int i);
int icl_linear_wgrad(const float* gy, const float* x, float* dw, float* db, void* ws, int64_t rows, int o, int i, void* stream);

/* Weight/bias gradient of a 1x1x1 convolution on a big volume (channel-major operands, s = D*H*W voxels per sample, s % 4 == 0):
 * gw[cout][cin] = sum_{b,v} gy[b][co][v] * x[b][ci][v]; gbias may be NULL.  Same row-split / slab scheme as icl_linear_wgrad. */
int64_t icl_conv1x1_wgrad_ws_bytes(int n, int64_t s, int cin, int cout);
int icl_conv1x1_wgrad(const float* x, const float* gy, float* gw, float* gbias, void* ws, int n, int cin, int cout, int64_t s,
                      int64_t x_bstride, int64_t gy_bstride, void* stream);

/* ---- Swin window attention: WindowAttention.forward core, networks/swinunetr_icl.py:728-747 —
 *   attn = softmax((q*scale) @ k^T + relative_position_bias [+ shift mask]);  out = attn @ v   per (window, head).
 * head_dim 16 (3-D SwinUNETR, n <= 352) or 32 (2-D Swin-UNet 7x7 windows, networks/swinunet_icl.py:120-155, n <= 96).
 * qkv  [B_, n, 3, heads, head_dim]  output of the qkv Linear (B_ = batch * nW windows, window id of row b_ = b_ % nW)
 * bias [heads, n, npad], npad = n rounded up to 16, pad columns = -1e30 (icl_window_attn_bias_elems floats; the caller gathers
 *      relative_position_bias_table[relative_position_index[:n,:n]] into it, :733-737)
 * regions int32 [nW, n] or NULL: region id of every token of the rolled volume; query/key pairs with different ids get -100
 *      (the dense attn_mask of compute_mask, :979-1016, is exactly -100 * (id_i != id_j))
 * out [B_, n, heads*head_dim];  lse [B_, heads, n] = log-sum-exp of every score row (saved for backward).
 * Backward: dqkv (layout of qkv) and, if dbias != NULL, dbias [chunks, heads, n, npad]: one slab per slice of the B_ windows
 * (icl_window_attn_bwd_chunks), to be summed by icl_relpos_bias_bwd_sum. */
int64_t icl_window_attn_bias_elems(int n, int heads);
/* bias[h][i][j] = table[index[i*idx_stride + j]][h] (j < n), -1e30 (n <= j < npad): the gather of :733-737 into the padded layout;
 * index = the int64 relative_position_index buffer [idx_stride, idx_stride] (343 x 343), table [table_rows, heads].
 * Backward: icl_relpos_bias_bwd_sum below. */
int icl_relpos_bias_fwd(const float* table, const int64_t* index, float* bias, int n, int heads, int idx_stride, void* stream);
int icl_window_attn_fwd(const float* qkv, const float* bias, const int32_t* regions, float* out, float* lse, int b_, int n, int heads,
                        int nw, int head_dim, float scale, void* stream);
int icl_window_attn_bwd(const float* qkv, const float* bias, const int32_t* regions, const float* out, const float* lse,
                        const float* dout, float* dqkv, float* dbias, int b_, int n, int heads, int nw, int head_dim, float scale,
                        void* stream);
/* d(bias) of icl_window_attn_bwd arrives as icl_window_attn_bwd_chunks(...) slabs [chunk][heads][n][npad] (one per slice of windows,
 * plain stores, every used element written: no zero fill).  icl_relpos_bias_bwd_sum adds them into the table gradient in a FIXED order —
 * inv: the padded positions i * npad + j of index[:n, :n] sorted by table row (stable), offs[t] .. offs[t + 1] the entries of row t —
 * so the gradient of relative_position_bias_table (swinunetr_icl.py:733-737) is bit-reproducible run to run.  dbias is used as scratch
 * (the slabs are first added, in order, into slab 0). */
int icl_window_attn_bwd_chunks(int b_, int n, int heads, int nw, int head_dim);
int icl_relpos_bias_bwd_sum(float* dbias, int chunks, const int32_t* inv, const int32_t* offs, float* dtable, int64_t table_rows, int n,
                            int heads, void* stream);

/* ---- on-device training augmentation: RandomRotFlip -> RandomCrop -> ToTensor of the 3-D trainers
 * (dataloaders/brats2019.py:80-147,177-189; composed at train_inherent_consistent_unet_3D_BraTS.py:66-73) as one gather over
 * volumes resident in HBM.  images[b] / labels[b]: HOST arrays of device pointers to the fp32 [n0,n1,n2] volume and its uint8
 * label; params: HOST int32 [batch][11] = {n0, n1, n2, k (np.rot90 count on axes 0,1), flip axis (0, 1, -1 none), zero padding
 * p0, p1, p2 (both sides), crop origin c0, c1, c2 (in the padded, rotated, flipped volume)}, drawn by the caller with the
 * reference's numpy call sequence.  image_out [batch,1,o0,o1,o2] fp32, label_out [batch,o0,o1,o2] int64.  batch <= 16. */
int icl_crop_rotflip(const void* const* images, const void* const* labels, const int32_t* params, int batch, float* image_out,
                     int64_t* label_out, int o0, int o1, int o2, void* stream);

/* ---- fused SGD(momentum, weight decay) step, torch.optim.SGD semantics (train_inherent_consistent_unet_3D_BraTS.py:85-86,115):
 * d = g + wd*p; m = first ? d : momentum*m + d; p -= lr*m.  The multi form takes HOST arrays of device pointers.
 * lr_dev (may be NULL): when given, the learning rate is read from this device scalar instead of `lr` (hipGraph replay). */
int icl_sgd_step(float* p, const float* g, float* m, int64_t n, float lr, float momentum, float weight_decay, int first,
                 const float* lr_dev, void* stream);
/* The same step for a Linear weight p [n, k] whose gradient is still factored, dW[n][k] = sum_{r < rows} g[r][n] * x[r][k]
 * (g [rows, n] = dL/dy, x [rows, k] = the layer input): the 13,824^2 token-axis MLP weights of Class_Decoder.mlp2
 * (networks/unet_3D_icl.py:258,267) are updated in one pass over p and m without ever forming the 764 MB gradient. */
int icl_sgd_step_factored(float* p, float* m, const float* g, const float* x, int rows, int n, int k, float lr, float momentum,
                          float weight_decay, int first, const float* lr_dev, void* stream);
/* The same update for MANY factor rows (gathered factors of a data-parallel step, nc = 16: 64 ... 1536 rows): d = g^T x on the bf16
 * matrix pipe from exact three-way bf16 splits of both factors (six MFMA terms per product, fp32 accumulation — fp32 accuracy), so
 * that the update stays an HBM stream (0.86 -> ~0.6 ms at 128 rows on a 13,824^2 matrix).  ws: icl_sgd_factored_split_ws_bytes. */
int64_t icl_sgd_factored_split_ws_bytes(int rows, int n, int k);
int icl_sgd_step_factored_split(float* p, float* m, const float* g, const float* x, void* ws, int rows, int n, int k, float lr,
                                float momentum, float weight_decay, int first, const float* lr_dev, void* stream);
/* icl_sgd_step_factored for <= 16 factor rows as a NARROW persistent launch: `workgroups` workgroups of 1,024 threads (one CU each) walk the
 * matrix, so that the 16 B / weight stream can run under the under-filled launches of the deep backward levels without taking their CUs
 * (FusedSGD.update_placement "deep"; same sums in the same order as icl_sgd_step_factored: bit-identical).  Returns 1 (nothing launched)
 * for more than 16 rows. */
int icl_sgd_step_factored_narrow(float* p, float* m, const float* g, const float* x, int rows, int n, int k, float lr, float momentum,
                                 float weight_decay, int first, const float* lr_dev, int workgroups, void* stream);
int icl_sgd_step_multi(void* const* p, const void* const* g, void* const* m, const int64_t* n, int count, float lr,
                       float momentum, float weight_decay, int first, const float* lr_dev, void* stream);

/* ---- the aligner's QUERY chain as fused stages (round 6; csrc/kernels/qchain.h).  Replaces, per resolution level of
 * InherentConsistent, the operator-by-operator mirror of Class_Decoder's query half — norm1_query -> attn.fc_q -> [read-out of
 * softmax(QK^T)V: icl_attn_fwd / icl_attn_bwd_ws] -> attn.proj -> q + drop_path(q) -> norm2 -> mlp.fc1 -> GELU -> mlp.fc2 ->
 * q + drop_path(.) -> query_convs (networks/unet_3D_icl.py:258-264, 283-297, 299-315, 197/220-222) — on tensors of <= 32 rows
 * (batch * classes) by <= 256 channels: one stage launch = rows -> row prologue -> slice of a product with ONE weight matrix ->
 * element epilogue -> rows.  The struct is plain C (pointers, sizes; device pointers unless stated).
 *   trans 0: y[R][N] = epi(pro(x)[R][K] W[N][K]^T + bias)     (a Linear's forward; K % 4 == 0, K <= 1024)
 *   trans 1: y[R][N] = epi(pro(x)[R][K] W[K][N])               (its input gradient; N % 4 == 0, K <= 1024)
 *   trans 2: y[R][K] = epi(pro(x))                             (no product; epi 5: y[nc][K] = sum over samples, R <= 8)
 *   pro 0 none | 1 LayerNorm(pa = gamma, pb = beta; so0 <- xhat, so1 <- rstd, so2 <- result) | 2 GELU (so2 <- result) |
 *       3 LayerNorm backward (x = d(normalised), pa = xhat, pb = rstd, pc = gamma, + pd residual rows, x (1 + f0) if pro_dp; so2 <- result) |
 *       4 rows x f1 (so2 <- result)
 *   epi 0 none | 1 x (1 + f0) | 2 ea[r][n] + f1 * v | 3 v * gelu'(ea[r][n]) | 4 + ea rows on [ea_r0, ea_r1), + eb rows on [eb_r0, eb_r1) |
 *       5 (trans 2) sum over the batch
 *   f0 / f1: the per-sample drop-path factors of the two residual sites (keep ? 1/(1-p) : 0; thresh 0, scale 1 = inactive = 1),
 *   same hash as icl_drop_path (sample index as the counter), dp_seed_dev as there.  R <= 32; sample of row r = r / nc.
 *   x_rows (0 = R): row r reads x row r % x_rows (a [1, nc, C] query broadcast over the batch). */
typedef struct IclQcStage {
  const float* x; int x_rows;
  const float* w; const float* bias; float* y;
  int R, K, N, trans, nc, npw;       /* npw: filled in by the launcher */
  int pro, epi;
  const float* pa; const float* pb; const float* pc; const float* pd; int pro_dp;
  float* so0; float* so1; float* so2;
  const float* ea; const float* eb; int ea_r0, ea_r1, eb_r0, eb_r1;
  uint32_t dp_seed[2], dp_thresh[2]; float dp_scale[2]; const uint32_t* dp_seed_dev;
  float eps;
} IclQcStage;
int icl_qchain_stage(const IclQcStage* stage, void* stream);
/* Every parameter gradient of a level in ONE launch (they are leaves of the chain).  Job arrays are HOST arrays of `count` (<= 12) entries:
 * kind 0: dw[N][K] = g[R][N]^T x[R][K], db[N] = column sums of g (db[i] may be NULL)  — Linear weight / bias;
 * kind 1: dw[K] = sum_r g[r][k] x[r][k], db[K] = sum_r g[r][k]                         — LayerNorm gamma / beta from (d(normalised), xhat). */
int icl_qchain_wgrad(const void* const* g, const void* const* x, void* const* dw, void* const* db, const int32_t* rows, const int32_t* n,
                     const int32_t* k, const int32_t* kind, int count, void* stream);

#ifdef __cplusplus
}
#endif
#endif
