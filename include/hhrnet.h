/*
 * hhrnet.h -- C-ABI of the MI355X-native HigherHRNet forward + associative-embedding decode.
 *
 * The reference (thawro/pytorch-human-pose) has no FFI: its plug-in points are Python
 * classes.  Each entry point below names the reference interface it stands in for
 * (paths relative to the reference's src/); the Python shims in
 * pytorch-human-pose_amd/keypoints/ bind them with ctypes.
 *
 * Conventions: every function returning int returns 0 on success, non-zero on error with
 * a thread-local message in hh_last_error().  All *device* pointers are plain HIP device
 * addresses (e.g. torch.Tensor.data_ptr()); `stream` is a hipStream_t passed as void*
 * (NULL = the legacy default stream).  Calls are asynchronous on `stream` unless noted.
 * A handle is not re-entrant; different handles may be used from different threads.
 */
#ifndef HHRNET_H
#define HHRNET_H
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define HH_ABI_VERSION 3
#define HH_DTYPE_BF16 1 /* bf16 MFMA operands, fp32 accumulate, bf16 NHWC activations */
#define HH_DTYPE_FP8 2  /* OCP e4m3 MFMA operands (v_mfma_f32_32x32x64_f8f6f4), fp32 accumulate, e4m3 NHWC activations with one
                           scale per tensor, e4m3 weights with one scale per output channel; needs hh_calibrate (BASELINE.json
                           configs[4]; no reference precedent: the reference infers in fp32, keypoints/model.py:79-83)     */

typedef struct hh_net hh_net;
typedef struct hh_decoder hh_decoder;

int hh_abi_version(void);
const char *hh_last_error(void);

/* ------------------------------------------------------------------ network forward
 * hh_create: HigherHRNet(num_kpts, C).__init__ -- keypoints/architectures/higher_hrnet.py:47-64
 * (backbone: hrnet.py:342-376).  Parameters are addressed by the reference's state-dict
 * key names (1810 keys for W32).                                                        */
hh_net *hh_create(int num_kpts, int C, int dtype);
void hh_destroy(hh_net *net);

/* ClassificationHRNet(C, num_classes) -- classification/architectures/hrnet.py:64-74 (BASELINE.json configs[0]):
 * the same backbone with a 4-scale last fusion + ClassificationHead.  Parameters/keys as the reference's
 * state dict (hh_num_params/hh_load_weights/hh_finalize work on the handle); forward writes logits [B,num_classes]. */
hh_net *hh_create_classifier(int C, int num_classes, int dtype);
int hh_forward_classifier(hh_net *net, const float *images, int B, int H, int W, float *logits, void *stream);

/* state_dict() introspection: key names and shapes in the reference's order. */
int hh_num_params(const hh_net *net);
const char *hh_param_name(const hh_net *net, int index);
int hh_param_shape(const hh_net *net, int index, int64_t shape[4]); /* returns ndim (0 for scalars) */

/* load_state_dict(): base/model.py:155-175. `host` is fp32 (int64 counters are ignored);
 * unknown names and shape mismatches are errors (strict=True semantics).                */
int hh_load_weights(hh_net *net, const char *name, const float *host, const int64_t *shape, int ndim);
/* Folds eval-mode BatchNorm (eps 1e-5) into the preceding conv, packs bf16 kernel-layout
 * weights and uploads them.  Fails if any parameter was never loaded.  Synchronous.     */
int hh_finalize(hh_net *net);

/* fp8 handles only: sets the per-tensor activation scales from `rounds` (0 = 2) forwards over a calibration batch
 * (images [B,3,H,W] fp32 NCHW, device).  Every layer's output maximum is taken from its fp32 epilogue values and mapped to
 * 240 of e4m3's 448; maxima accumulate over calls until the next hh_finalize.  Synchronous.  hh_forward on an fp8 handle
 * fails until this has run.                                                                                              */
int hh_calibrate(hh_net *net, const float *images, int B, int H, int W, int rounds, void *stream);

/* The host-side OCP e4m3fn codec the fp8 weight packer uses (round to nearest even, saturating at +-448, NaN = 0x7f);
 * exported so that it can be checked against an independent implementation without a GPU.                                */
int hh_e4m3_encode(const float *x, int64_t n, unsigned char *out);
int hh_e4m3_decode(const unsigned char *x, int64_t n, float *out);

/* Allocates the activation workspace for inputs up to [B,3,H,W] (H, W multiples of 32).
 * Synchronous; hh_forward calls it implicitly when the shape grows.                     */
int hh_reserve(hh_net *net, int B, int H, int W);
int64_t hh_workspace_bytes(const hh_net *net);

/* HigherHRNet.forward: higher_hrnet.py:66-81.
 *   images          [B,3,H,W]        fp32 NCHW (device)
 *   init_heatmaps   [B,2K,H/4,W/4]   fp32 NCHW (device): stage-0 heatmaps = [:, :K], tags = [:, K:]
 *   deconv_heatmaps [B,K,H/2,W/2]    fp32 NCHW (device): stage-1 heatmaps
 * use_graph != 0 replays a cached hipGraph when (pointers, shape) repeat.               */
int hh_forward(hh_net *net, const float *images, int B, int H, int W, float *init_heatmaps, float *deconv_heatmaps,
               int use_graph, void *stream);

/* Multi-lane execution (default on): independent resolution branches / fusion outputs are launched on
 * internal HIP streams forked from and joined back to `stream` with events.  0 = everything on `stream`. */
int hh_set_multi_lane(hh_net *net, int enable);

/* Static check of the multi-lane schedule (no GPU needed): every read-after-write, write-after-read and write-after-write
 * pair of ops on different lanes must be ordered by a join/dependency edge.  0 = no hazard.                              */
int hh_debug_check_plan(const hh_net *net);

/* Algorithmic conv/deconv FLOPs (2*MACs) of one forward at this shape -- SURVEY.md §8d.  */
double hh_forward_flops(const hh_net *net, int B, int H, int W);

/* Live per-launch timing for bench.py's roofline line: when enabled, hh_forward runs eagerly and brackets
 * every convolution launch with HIP events recorded on `stream`.  hh_profile_get(i) returns the i-th
 * launch since hh_profile_enable: kernel instantiation index, algorithmic FLOPs (2*MACs) and bytes (input + output
 * (+ residual) + weights, each once: no halo re-reads) of that launch,
 * elapsed milliseconds, and the state-dict prefix of the layer.  hh_conv_config describes an
 * instantiation as {KS, S, KC, NT, WC, PT, TW}.  `ms` = hipEventElapsedTime of the start / stop events the launch itself
 * was given (hipExtLaunchKernelGGL): the runtime fills them from the dispatch packet's begin / end timestamps, the same
 * clock pair rocprofv3's kernel trace reports.  `kernel_ms` is first-workgroup-start to last-workgroup-end read by the
 * kernel itself from the device wall clock (hipDeviceAttributeWallClockRate): shorter, it leaves out the dispatch ramp
 * and the end-of-kernel write-back; only with hh_profile_enable(net, 2) -- the same-address atomics that stamp it lengthen
 * each launch by 3-5 us as the dispatch timestamps see it, so mode 1 (events only) is the one to quote -- else -1.     */
int hh_profile_enable(hh_net *net, int enable);
int hh_profile_count(const hh_net *net);
int hh_profile_get(hh_net *net, int index, int *cfg, double *flops, double *bytes, float *ms, float *kernel_ms, const char **layer);
/* mode 2 only, kernels that stamp it (the fused 32-channel block): the core clock workgroup 0 of launch `index` ran at, from
 * s_memtime / s_memrealtime deltas inside the kernel; 0 = not stamped.  Under load the chip holds this well below its 2.4 GHz. */
int hh_profile_clock(hh_net *net, int index, double *ghz);
int hh_conv_config(int cfg, int out[7]);
/* 1 if instantiation `cfg` runs its K loop on two LDS buffers (conv_mfma.hip, DB), 0 if not, -1 for an unknown index */
int hh_conv_config_double_buffered(int cfg);

/* Kernel micro-benchmark used by tools/conv_bench.py (not on the hot path): `iters` back-to-back launches of
 * convolution instantiation `cfg` on random bf16 data, HIP-event timed; returns ms per launch.        */
int hh_debug_conv_bench(int cfg, int B, int Hin, int Win, int cin, int cout, int with_res, int relu, int iters,
                        float *ms_per_launch, unsigned long long *stamps16, int ref_cfg, float *max_diff);

int hh_debug_bb_bench(int B, int H, int W, int iters, float *ms_per_launch, unsigned long long *stamps64);
/* the two fused 32-channel block kernels (tile form / producer-consumer form) on the same input: output difference and time */
int hh_debug_bb_compare(int B, int H, int W, int iters, float *max_diff, float *ms_classic, float *ms_pc);

/* Debug taps (parity tests): when enabled, hh_forward copies selected intermediate
 * activations; hh_tap_read converts one to fp32 NCHW on the host. Names follow the
 * reference module paths, e.g. "stages.2.blocks.3#1" = output 1 of backbone.stages[2].blocks[3]. */
int hh_set_taps(hh_net *net, int enable);
int hh_num_taps(const hh_net *net);
const char *hh_tap_name(const hh_net *net, int index);
int hh_tap_shape(const hh_net *net, int index, int64_t shape[4]); /* N,C,H,W of the last forward */
int hh_tap_read(hh_net *net, int index, float *host_nchw);        /* synchronous */

/* InferenceKeypointsModel.prepare_input on the device (keypoints/model.py:70-76; resize-align warp of
 * base/transforms/utils.py:89-97): `image_hwc` uint8 RGB [h,w,3] (device), `dst_to_src` the INVERSE of the 2x3 affine
 * that get_affine_transform returns (cv2.warpAffine maps every destination pixel back; hh_invert_affine), output fp32 NCHW [3,H,W] = Normalize(ToTensor(warpAffine(image))).
 * The warp is OpenCV 4.9's 8-bit INTER_LINEAR path restated (fixed-point coordinates and weights, see hh_warp_affine_u8);
 * parity with cv2 itself is UNPINNED (no cv2 in the build or run images): checked against oracle/transforms.py. */
int hh_preprocess_u8(const unsigned char *image_hwc, int h, int w, const double dst_to_src[6], float *out_nchw, int H, int W,
                     const float mean[3], const float stdv[3], void *stream);

/* The same for a batch of raw images of any sizes in ONE launch (the batched caller behind the reference's per-image `__call__`,
 * bin/eval.py:18-49): the images lie in one device buffer, image i at `images_base + descs[i].offset` with its own size and affine;
 * `descs_dev` is a DEVICE array (the caller ships it with the pixels in the same host->device copy); out_nchw is [n,3,H,W]. */
typedef struct hh_image_desc {
    long long offset;     /* bytes from images_base */
    int h, w;             /* raw image size */
    double dst_to_src[6]; /* as in hh_preprocess_u8 */
} hh_image_desc;
int hh_preprocess_u8_batch(const unsigned char *images_base, const hh_image_desc *descs_dev, int n, float *out_nchw, int H, int W,
                           const float mean[3], const float stdv[3], void *stream);

/* Flip test-time augmentation, keypoints/model.py:85-94 (COCO_FLIP_INDEX: keypoints/transforms.py:11).
 * hh_flip_images: out = flip(images, W axis), fp32 NCHW.
 * hh_flip_merge : hm[b,k] = (hm[b,k] + flip_w(hm_flipped[b, perm[k]])) / 2  in place for `hm`
 *                 (K channels each, batch strides in elements), and
 *                 tags_out[b,k] = flip_w(tags_flipped[b, perm[k]]).                       */
int hh_flip_images(const float *images, float *out, int B, int C, int H, int W, void *stream);
int hh_flip_merge(float *hm, int64_t hm_bstride, const float *hm_flipped, int64_t hmf_bstride, const float *tags_flipped,
                  int64_t tf_bstride, float *tags_out, int64_t to_bstride, const int32_t *perm_host, int B, int K,
                  int h, int w, void *stream);

/* ------------------------------------------------------------------ decode
 * hh_decoder_create: MPPEHeatmapParser(num_kpts, max_num_people, det_thr, tag_thr)
 * -- keypoints/grouping.py:67-78.  Thresholds are doubles because the reference compares
 * float32 scores / float64 distances against Python floats.                             */
hh_decoder *hh_decoder_create(int num_kpts, int max_people, double det_thr, double tag_thr);
void hh_decoder_destroy(hh_decoder *dec);
int hh_decoder_reserve(hh_decoder *dec, int B, int H, int W, int E); /* H, W = full (model-input) resolution */
/* By default hh_decode skips the NMS / top-k work of every 16x16-pixel block whose averaged half-resolution source values cannot
 * exceed det_thr (and only keeps peaks above it): such pixels cannot contribute a candidate that survives match_by_tag's
 * `score > det_thr` filter (grouping.py:98-102), so joints / scores / num_people are unchanged, bit for bit; the no-group
 * fallback's top-1 candidates are then recomputed from the maps for the flagged images.  What changes is the candidate list
 * itself (sub-threshold entries are missing): enable = 1 takes the exhaustive path (the stage average materialised, every
 * 60x60 tile processed), which hh_decoder_read_topk (the reference's full top_k) needs.                                    */
int hh_decoder_set_exact_topk(hh_decoder *dec, int enable);

/* InferenceKeypointsResult.from_preds aggregation + MPPEHeatmapParser.parse, batched:
 * keypoints/results.py:225-238 + keypoints/grouping.py:252-283.
 *   hm_q   [B,K,hq,wq]   fp32, batch stride hm_q_bstride elements (channel-slice views allowed)
 *   hm_h   [B,K,2hq,2wq] fp32
 *   tags_q E pointers (host array of device pointers), each [B,K,hq,wq]
 * Outputs (device): joints [B,max_people,K,3+E] (x, y, score, tag...; zero rows = no person),
 * scores [B,max_people], num_people [B], flags [B] (may be NULL).  Bit-exact with the reference on identical inputs
 * (ties between equal candidate scores are ordered by ascending pixel index).
 * flags[b] bits: HH_DECODE_FALLBACK = no group was formed and the one person returned is the best-candidate pseudo-person
 * of grouping.py:262-269 (the reference's arrays are float64 there with the score 0.01 as a double; the device arrays stay
 * float32 and the host shim widens them); HH_DECODE_SOLVER_GUARD = the assignment solver stopped at its iteration guard,
 * the image's result is invalid and the caller must raise.                                                           */
#define HH_DECODE_FALLBACK 1
#define HH_DECODE_SOLVER_GUARD 2
int hh_decode(hh_decoder *dec, const float *hm_q, int64_t hm_q_bstride, const float *hm_h, int64_t hm_h_bstride,
              const float *const *tags_q, const int64_t *tags_bstride, int E, int B, int hq, int wq, int adjust,
              int refine, float *joints, float *scores, int32_t *num_people, int32_t *flags, void *stream);

/* MPPEHeatmapParser.parse on explicit full-resolution maps (grouping.py:252-283):
 *   hm_full [B,K,H,W] fp32, tags_full [B,K,H,W,E] fp32 (both contiguous).              */
int hh_parse(hh_decoder *dec, const float *hm_full, const float *tags_full, int E, int B, int H, int W, int adjust,
             int refine, float *joints, float *scores, int32_t *num_people, int32_t *flags, void *stream);

/* Training loss of keypoints/loss.py, each fused with its gradient (all pointers device memory, fp32).
 * `scratch`: >= max(1024, 2*B) doubles.  Sums are taken in double in a fixed order (results do not depend on the launch).
 *
 * hh_loss_heatmaps = HeatmapsLoss.forward (loss.py:12-16): *loss = mean((pred - target)^2 * mask[:,None]);
 *   pred [B,K,h,w] with batch stride pred_bstride (a channel slice of a wider tensor is fine), target [B,K,h,w] and
 *   mask [B,h,w] contiguous; if grad != NULL, grad[b,k] (batch stride grad_bstride) = d loss / d pred.
 * hh_loss_ae_grouping = AEGroupingLoss.forward (loss.py:20-61): push_pull[0] = push, [1] = pull, both / batch size
 *   (calculate_loss, loss.py:90-92, scales them by 1e-3 afterwards); tags [B,K,h,w]; joints [B,P,K,3] int32 (x, y, vis)
 *   padded to P people, num_people[b] of them valid, x in [0,w), y in [0,h) wherever vis > 0 (the caller checks);
 *   if grad != NULL, push_scale * d push + pull_scale * d pull is ADDED to grad (zero it first).                    */
int hh_loss_heatmaps(const float *pred, int64_t pred_bstride, const float *target, const float *mask, int B, int K, int h, int w,
                     float *loss, float *grad, int64_t grad_bstride, double *scratch, void *stream);
int hh_loss_ae_grouping(const float *tags, int64_t tags_bstride, const int32_t *joints, const int32_t *num_people, int B, int P, int K,
                        int h, int w, float *push_pull, float *grad, int64_t grad_bstride, float push_scale, float pull_scale,
                        double *scratch, void *stream);

/* Building blocks of the training step (keypoints/module.py:43-71), assembled into the net's training forward / backward
 * by keypoints/train_net.py with torch autograd as the tape.  Activations are NHWC bf16 [B,H,W,C] (= torch channels_last),
 * parameters fp32, all device pointers.
 *
 * hh_conv2d: y = act(conv(x, w) + bias (+ res)) with the CURRENT fp32 weights w [cout][cin][ks][ks] (packed on the device
 *   each call), ks in {1,2,3} (2x2: stride 1), stride in {1,2}; pad_y / pad_x = top / left zero padding, -1 = (ks-1)/2
 *   (the output keeps the input size at stride 1, so a 2x2 kernel with pad 0 pads bottom/right instead).  mode 1 = data gradient of the stride-1 conv with these
 *   weights: x is dL/dy [B,H,W,cout], y is dL/dx [B,H,W,cin] (the same kernel with rotated, transposed weights).
 *   mode 2 = data gradient of the 3x3 stride-2 conv: x is dL/dy [B,H,W,cout], y is dL/dx [B,2H,2W,cin] (four
 *   output-parity phases, each a 2x2 conv over dL/dy).
 *   Input channels (of the conv that runs) % 16 == 0, output channels % 8 == 0; workspace: hh_conv2d_workspace_bytes.
 * hh_bn_train_forward = nn.BatchNorm2d in training mode on [P = B*H*W, C] (+ residual, + ReLU): batch mean and biased
 *   variance, y = act(gamma * (x - mean) * invstd + beta (+ res)); mean / invstd are kept for the backward.
 *   scratch: 256 * C * 2 doubles.  (The running statistics are updated by the caller: plain torch arithmetic on C floats.)
 * hh_bn_train_backward: dx, dgamma, dbeta (and dres = the gradient after the ReLU mask, if dres != NULL).
 * hh_bn_train_backward_plain: the same for a BatchNorm that had no residual input, without its stored output y: no ReLU needs
 *   nothing of it, and with ReLU the mask y > 0 is recomputed from x (gamma * (x - mean) * invstd + beta > 0, evaluated by the
 *   function the forward evaluated) -- one tensor pass less in each of the two kernels.
 * hh_conv2d_wgrad: dw [cout][cin][ks][ks] fp32 = dL/dW of y = conv(x, W) (padding (ks-1)/2) from x [B,H,W,cin] and
 *   dy [B,Ho,Wo,cout]; 3x3 stride 1/2 and 1x1 stride 1, channel counts % 8 == 0.  A GEMM contracted over pixels on MFMA
 *   (operands read from LDS with the transposing ds_read_b64_tr_b16), partial sums reduced in a fixed order.          */
int64_t hh_conv2d_workspace_bytes(int cin, int cout, int ks, int mode);
int64_t hh_conv2d_wgrad_workspace_bytes(int B, int H, int W, int cin, int cout, int ks, int stride);
int hh_conv2d_wgrad(const void *x, const void *dy, int B, int H, int W, int cin, int cout, int ks, int stride, int pad_y, int pad_x, float *dw,
                    void *workspace, void *stream);
int hh_conv2d(const void *x, int B, int H, int W, int cin, const float *w, int cout, int ks, int stride, int mode, int pad_y, int pad_x,
              const float *bias, const void *res, int relu, void *y, void *workspace, void *stream);
/* The same convolution with weights packed ahead of it.  A training step packs ~700 weight sets (forward layout and
 * data-gradient layout of every conv); as separate launches that is ~700 tiny dependent kernels whose launch gaps cost more
 * than the packing.  hh_pack_conv_weights_batch packs n weight sets in ONE launch: w[i] fp32 [cout][cin][ks][ks] (device
 * pointers in a HOST array), packed[i] device buffers of hh_conv2d_packed_elems(cin, cout, ks, stride, mode) bf16 elements,
 * shapes host int32 [n][5] = cout, cin, ks, stride, mode (modes as in hh_conv2d), descs_dev a device scratch of
 * n * 4 * 64 bytes.  hh_conv2d_packed = hh_conv2d on such a buffer (bias NULL or a multiple of 32 long).  The packed copy
 * is valid until the fp32 weights change (the optimizer step).                                                            */
int64_t hh_conv2d_packed_elems(int cin, int cout, int ks, int stride, int mode);
int hh_pack_conv_weights_batch(int n, const float *const *w, void *const *packed, const int32_t *shapes, void *descs_dev, void *stream);
int hh_conv2d_packed(const void *x, int B, int H, int W, int cin, const void *w_packed, int cout, int ks, int stride, int mode, int pad_y,
                     int pad_x, const float *bias, const void *res, int relu, void *y, void *stream);
int hh_bn_train_forward(const void *x, int64_t P, int C, const float *gamma, const float *beta, float eps, const void *res, int relu,
                        void *y, float *mean, float *invstd, double *scratch, void *stream);
int hh_bn_train_backward(const void *x, const void *y, const void *dy, int64_t P, int C, const float *mean, const float *invstd,
                         const float *gamma, int relu, void *dx, void *dres, float *dgamma, float *dbeta, double *scratch, void *stream);
int hh_bn_train_backward_plain(const void *x, const void *dy, int64_t P, int C, const float *mean, const float *invstd, const float *gamma,
                               const float *beta, int relu, void *dx, float *dgamma, float *dbeta, double *scratch, void *stream);

/* FusionLayer's sum in the training step (hrnet.py:214-229: `sum_j f_ij(x_j)` then ReLU, with nn.Upsample(nearest) on the
 * low-resolution terms, hrnet.py:200-205): out = act(sum_j term_j[b, y >> shift_j, x >> shift_j, :]) over 1..4 NHWC bf16
 * terms [B, H >> shift_j, W >> shift_j, C] (shift 0 first), summed in fp32 -- the upsampled tensors are never materialised.
 * Backward: g = dy * (out > 0) [B,H,W,C] is the gradient of every shift-0 term (g may be NULL when relu == 0: then it is dy
 * itself); dup[j] [B, H >> s, W >> s, C] = the 2^s x 2^s block sums of g for the nup upsampled terms.                     */
int hh_fusion_sum_forward(const void *const *terms, const int *shifts, int nterms, int B, int H, int W, int C, int relu, void *out, void *stream);
int hh_fusion_sum_backward(const void *dy, const void *out, int relu, int B, int H, int W, int C, void *g, void *const *dup, const int *up_shift,
                           int nup, void *stream);

/* SyncBatchNorm (src/base/model.py:42-44: `to_DDP(..., use_batchnorm=True)` converts every BatchNorm2d, the reference
 * trainer's default, trainer.py:44,253; experiments/keypoints/higher_hrnet_32.yaml:17 turns it off): the two passes above split around their one exchange step.
 *   hh_bn_train_stats:          sums[2c], sums[2c+1] = sum x, sum x^2 over THIS rank's P pixels (doubles).
 *   -- the caller all-reduces (SUM) sums and the pixel count over the ranks (RCCL) --
 *   hh_bn_train_normalize:      mean / invstd from the global sums and count, then the same apply pass.
 *   hh_bn_train_backward_stats: sums = sum g, sum g * xhat of this rank (g = dy after the ReLU mask); dbeta / dgamma = the
 *                               same local sums as floats (parameter gradients are averaged by DDP like all others).
 *   -- all-reduce (SUM) sums --
 *   hh_bn_train_backward_apply: dx (and dres) from the global sums and count.
 * With count == P and no exchange the results equal hh_bn_train_forward / hh_bn_train_backward.  scratch: 256*C*2 doubles. */
int hh_bn_train_stats(const void *x, int64_t P, int C, double *sums, double *scratch, void *stream);
int hh_bn_train_normalize(const void *x, int64_t P, int C, const double *sums, double count, const float *gamma, const float *beta, float eps,
                          const void *res, int relu, void *y, float *mean, float *invstd, void *stream);
int hh_bn_train_backward_stats(const void *x, const void *y, const void *dy, int64_t P, int C, const float *mean, const float *invstd, int relu,
                               double *sums, float *dgamma, float *dbeta, double *scratch, void *stream);
int hh_bn_train_backward_apply(const void *x, const void *y, const void *dy, int64_t P, int C, const float *mean, const float *invstd,
                               const float *gamma, int relu, const double *sums, double count, void *dx, void *dres, double *scratch,
                               void *stream);

/* Multi-scale test-time augmentation (BASELINE.json configs[3]; an extension: the reference only calls its resize helper
 * with scale 1, keypoints/model.py:73): dst[B,K,H,W] (+)= weight * bilinear(src[B,K,h,w] -> HxW) with the arithmetic of
 * F.interpolate(mode="bilinear", align_corners=False); init != 0 overwrites dst.  Batch strides in elements.            */
int hh_resize_accumulate(const float *src, int64_t src_bstride, int B, int K, int h, int w, float *dst, int64_t dst_bstride, int H,
                         int W, float weight, int init, void *stream);

/* Candidates of the last hh_decode/hh_parse call (MPPEHeatmapParser.top_k, grouping.py:147-170),
 * copied to host: tags_k [B,K,max_people,E], coords_k [B,K,max_people,2] (x,y), scores_k [B,K,max_people].
 * Synchronous; for parity tests.  Needs the exhaustive candidate lists: hh_decoder_set_exact_topk(dec, 1) before the decode. */
int hh_decoder_read_topk(hh_decoder *dec, float *tags_k, int32_t *coords_k, float *scores_k);

/* Test hook: the matcher's assignment solver (munkres 1.1.4's step machine, grouping.py:55-59, as one wavefront) alone on one
 * square float64 cost matrix in host memory, n <= 32: star[i] = the column assigned to row i.  Pad a rectangular problem with zeros
 * as munkres.pad_matrix does.  */
int hh_debug_munkres(const double *cost, int n, int32_t *star);

/* get_affine_transform(center, scale, rot=0, output_size, inverse) of base/transforms/utils.py:25-57 -> the 2x3 matrix
 * (row-major, 6 doubles) cv2.getAffineTransform returns for the reference's three float32 point pairs: the 6x6 system solved
 * by OpenCV's own LU in float64.  Only scale[0] enters (the reference ignores scale_h).  Host-side. */
int hh_get_affine_transform(double center_x, double center_y, double scale_w, double dst_w, double dst_h, int inverse, double out[6]);

/* The destination -> source matrix cv2.warpAffine builds from a forward 2x3 matrix (no WARP_INVERSE_MAP), same operation
 * order; what hh_preprocess_u8 / hh_warp_affine_u8 take as `dst_to_src`.  Host-side. */
int hh_invert_affine(const double m[6], double out[6]);

/* cv2.warpAffine(image, M, (W, H)) itself, uint8 HWC in and out on the device (resize_align_multi_scale,
 * base/transforms/utils.py:89-97): OpenCV's fixed-point bilinear (10-bit coordinates, 5-bit fractions, int16 weights summing to
 * 32768, (v + 2^14) >> 15), constant-0 border. */
int hh_warp_affine_u8(const unsigned char *image_hwc, int h, int w, const double dst_to_src[6], unsigned char *out_hwc, int H, int W,
                      void *stream);

/* get_final_kpts_coords / transform_coords: keypoints/results.py:158-171,189-201 with
 * get_affine_transform(inverse=True, rot=0) (hh_get_affine_transform). Host-side,
 * float64: xy_out[i] = M @ (xy_in[i], 1) as affine_transform (base/transforms/utils.py:5-8). */
int hh_transform_coords(const float *xy_in, int n, double center_x, double center_y, double scale_w, double dst_w,
                        double dst_h, double *xy_out);

#ifdef __cplusplus
}
#endif
#endif /* HHRNET_H */
