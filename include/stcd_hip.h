/* stcd_hip.h -- C ABI of the MI355X (gfx950) bi-temporal change-detection engine.
 *
 * The reference (VCISwang/STCD) has NO native interface: its hot path sits behind a Python
 * torch.nn.Module duck-type built by a factory:
 *     define_G(args)            /root/reference/models/networks.py:138-215
 *     Module(input_nbr,label_nbr).forward(x1,x2) -> logits
 *                               /root/reference/models/SiamUnet_diff.py:13,94-181
 *                               /root/reference/models/SiamUnet_conc.py:94-181
 *                               /root/reference/models/SiamUnet_sub.py:94-180
 *                               /root/reference/models/SNUNet.py:63-152
 *     loss(logits,label)        /root/reference/models/losses.py:6-21 (cross_entropy), :24-34 (cd_loss)
 * This header is the boundary a maintainer would bind instead (ctypes stub: INTEGRATION.md): plain
 * pointers and sizes, no torch types.  All pointers are DEVICE pointers unless a parameter says "host".
 * Every entry point returns 0 on success; on failure it returns non-zero and stcd_last_error() gives
 * the reason (thread-local).  No exceptions cross the ABI.  An engine handle is not thread-safe: one
 * handle per process / per GPU (the reference pins nn.DataParallel to one device,
 * /root/reference/train_pse_cd.py:405-417).
 *
 * Ownership: the caller owns inputs, parameters, gradients, outputs and the workspace; the engine borrows
 * them for the duration of one call, enqueued on the caller's HIP stream.  The engine allocates no device
 * memory and never synchronises the device.
 */
#ifndef STCD_HIP_H
#define STCD_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define STCD_ABI_VERSION 2

/* network family: replaces the class chosen by define_G (networks.py:145-169) */
#define STCD_ARCH_DIFF 0   /* "SiamUnet_abs"  -> SiamUnet_diff  */
#define STCD_ARCH_CONC 1   /* "SiamUnet_conc" -> SiamUnet_conc  */
#define STCD_ARCH_SUB 2    /* "SiamUnet_sub"  -> SiamUnet_sub   */
#define STCD_ARCH_SNUNET 3 /* "SNUNet"        -> SNUNet_ECAM    */
#define STCD_ARCH_SEGCD 4  /* smp.SegCD(encoder_name="resnet50"): the model the scripts train (train_pse_cd.py:419-427,
                            * segmentation_models_pytorch/decoders/unet/model.py:267-332).  Its forward returns THREE maps
                            * (mask_t1, mask_t2, change): `logits` / `grad_logits` of stcd_forward / stcd_backward hold
                            * [3*batch, label_ch, H, W] floats in that order; H and W must be divisible by 32. */
/* the same class over the other plain ResNet encoders of the reference's registry
 * (segmentation_models_pytorch/encoders/resnet.py:126-171; blocks: models/resnet.py:37-75 BasicBlock, :78-124 Bottleneck) */
#define STCD_ARCH_SEGCD_R18 5  /* encoder_name="resnet18":  BasicBlock  [2, 2, 2, 2]  */
#define STCD_ARCH_SEGCD_R34 6  /* encoder_name="resnet34":  BasicBlock  [3, 4, 6, 3]  */
#define STCD_ARCH_SEGCD_R101 7 /* encoder_name="resnet101": Bottleneck  [3, 4, 23, 3] */
#define STCD_ARCH_SEGCD_R152 8 /* encoder_name="resnet152": Bottleneck  [3, 8, 36, 3] */
/* "Unet" -> Unet, the FC-EF network of the same paper (models/Unet.py:10-154; define_G name "Unet", models/networks.py:144-145): ONE
 * encoder stream over cat(x1, x2) (conv11 takes 2 * in_ch channels, in_ch <= 4), skips = the stream's own activations, decoder and
 * state_dict layout of SiamUnet_diff; forward returns the logits tensor. */
#define STCD_ARCH_FCEF 9
/* "SiamUnet_cross_conc" -> SiamUnet_cross_conc (models/SiamUnet_crossconc.py:35-212; define_G name, models/networks.py:152-153): the
 * FC-Siam encoder / decoder with a cross_conc block on every skip (:11-33: the two dates' channels interleaved into a grouped 3x3
 * conv (C groups of 2 -> 1), BatchNorm, ReLU, a 3x3 conv C -> C, BatchNorm, ReLU); forward returns [logits]. */
#define STCD_ARCH_XCONC 10
/* smp.UnetSeg (decoders/unet/model.py:109-171), the single-image ResNet UNet train_sup.py:303 trains: the same encoder /
 * decoder / head on ONE image batch (x2 of stcd_forward is ignored; BatchNorm over the whole batch; logits [batch, label_ch,
 * H, W]).  STCD_ARCH_UNETSEG + k, k = 0..4: resnet50, resnet18, resnet34, resnet101, resnet152 (the order of the ids above). */
#define STCD_ARCH_UNETSEG 16
/* smp.FFCTLCD (decoders/unet/model.py:335-423; the commented alternative of train_pse_cd.py:419 / train_stcd.py:637): shared
 * encoder on both dates, the shared decoder + head on |f1 - f2| (feature level), f1 and f2 -- three decoder passes with their
 * own BatchNorm batch statistics, in that order --, change = min(head(dec(|f1 - f2|)), |mask_t1 - mask_t2|).  Same outputs as
 * STCD_ARCH_SEGCD ([3*batch, label_ch, H, W] = mask_t1, mask_t2, change).  STCD_ARCH_FFCTLCD + k, k as above. */
#define STCD_ARCH_FFCTLCD 32

/* ChangeFormerV6 (/root/reference/models/ChangeFormer.py:1669-1701; define_G name "ChangeFormerV6", models/networks.py:195-196;
 * BASELINE.json configs[4]): hierarchical transformer encoder shared by both dates + MLP / conv-difference decoder.  Its forward
 * returns FIVE maps [p_c4, p_c3, p_c2, p_c1, cp] (ChangeFormer.py:1578-1622): `logits` of stcd_forward holds them back to back
 * (stcd_cf_output_info gives offset / size of each; stcd_output_floats the total), `grad_logits` of stcd_backward has the same
 * layout and only the gradient of the LAST map (cp) is propagated (the reference's loss uses G_pred[-1], models/trainer.py:311).
 * H and W must be divisible by 32.  stcd_create(STCD_ARCH_CHANGEFORMER, ...) builds the V6 configuration with embed_dim 256;
 * stcd_create_changeformer takes any configuration of the same class family. */
#define STCD_ARCH_CHANGEFORMER 64

/* arithmetic / storage type of activations. Parameters, gradients, BN statistics, logits: always fp32. */
#define STCD_DTYPE_F32 0  /* parity mode: fp32 storage, fp32 FMA */
#define STCD_DTYPE_BF16 1 /* production: bf16 storage, MFMA with fp32 accumulation */

typedef struct stcd_engine stcd_engine;

typedef struct stcd_tensor_info {
    char name[64];    /* reference state_dict key, e.g. "conv43d.weight" */
    int32_t ndim;
    int64_t shape[4]; /* reference shape, e.g. (256,128,3,3) */
    int64_t offset;   /* element offset into the flat fp32 parameter / gradient buffer */
    int64_t numel;
} stcd_tensor_info;

typedef struct stcd_bn_info {
    char name[64];           /* module name, e.g. "bn11": buffers are name.running_mean / name.running_var */
    int32_t channels;
    int32_t calls_per_forward; /* 2 for the shared encoder (T1 then T2), 1 otherwise: num_batches_tracked increment */
    int64_t offset;          /* running_mean at offset, running_var at offset+channels, in the flat BN buffer */
} stcd_bn_info;

typedef struct stcd_dropout_info {
    char name[64];   /* module name, e.g. "do11" */
    int32_t rows;    /* 2*batch for encoder layers (T1 rows then T2 rows), batch for decoder layers */
    int32_t channels;
    int64_t offset;  /* element offset into the flat fp32 mask buffer; mask[row*channels + c] in {0, 1/(1-p)} */
} stcd_dropout_info;

/* ChangeFormerV6.__init__ / EncoderTransformer_v3.__init__ arguments (ChangeFormer.py:1671-1691, 1344-1348) */
typedef struct stcd_cf_config {
    int32_t in_ch, out_ch;          /* input_nc, output_nc */
    int32_t embed_dims[4];          /* [64, 128, 320, 512] */
    int32_t depths[4];              /* [3, 3, 4, 3] */
    int32_t num_heads[4];           /* [1, 2, 4, 8] */
    int32_t sr_ratios[4];           /* [8, 4, 2, 1] */
    int32_t mlp_ratio;              /* 4 */
    int32_t embedding_dim;          /* embed_dim = 256: decoder width */
    int32_t patch1, patch;          /* patch_embed1: k7 s4; patch_embed2..4: k = patch_size = 7, s2 */
    float drop_rate, attn_drop, drop_path_rate;   /* 0.1, 0.1, 0.1 */
    float diff_drop;                /* conv_diff's nn.Dropout(p=0.6), ChangeFormer.py:1143 */
} stcd_cf_config;

/* one Dropout / DropPath site of the configured ChangeFormer engine: masks are never stored -- element i of the site (linear index
 * over dims, the engine's NHWC / [image, head, query, key] order; encoder sites hold both dates: 2*batch images, date 1 first)
 * is kept iff (fmix32(i * 0x9E3779B1 + stcd_cf_site_seed(seed, site)) >> 8) >= floor(p * 2^24), and scaled by 1 / (1 - p). */
typedef struct stcd_cf_site {
    char name[96];                  /* e.g. "Tenc_x2.block1.0.attn.attn_drop", "TDec_x2.diff_c4.3" */
    int32_t ndim;
    int32_t dims[4];
    float p;
} stcd_cf_site;

const char* stcd_last_error(void);
int stcd_abi_version(void);

/* ---- engine lifecycle: replaces Module.__init__ (SiamUnet_diff.py:13-92, SNUNet.py:65-113) ---- */
int stcd_create(int arch, int in_ch, int label_ch, int dtype, stcd_engine** out);
int stcd_cf_default_config(stcd_cf_config* cfg);                     /* the ChangeFormerV6 values above */
int stcd_create_changeformer(const stcd_cf_config* cfg, int dtype, stcd_engine** out);
void stcd_destroy(stcd_engine* e);
/* floats of the `logits` / `grad_logits` buffers for the current configuration (every family) */
int64_t stcd_output_floats(const stcd_engine* e);
/* ChangeFormer: output map i (0..4 = p_c4, p_c3, p_c2, p_c1, cp), each [batch, out_ch, height, width] fp32 NCHW at `offset` floats */
int stcd_cf_output_info(const stcd_engine* e, int i, int64_t* offset, int* height, int* width);
int stcd_cf_num_sites(const stcd_engine* e);
int stcd_cf_site_get(const stcd_engine* e, int i, stcd_cf_site* out);
uint32_t stcd_cf_site_seed(uint64_t seed, int site);
/* change the element-wise dropout rates of a ChangeFormer engine (re-run stcd_configure afterwards); DropPath rates are fixed at
 * creation (they are a per-block schedule) */
int stcd_cf_set_drop_rates(stcd_engine* e, float drop_rate, float attn_drop, float diff_drop);
/* multi_scale_train (models/trainer.py:300-309: the loss is a weighted sum over ALL five predictions): on != 0 makes stcd_backward
 * propagate the gradients of the four auxiliary maps too -- `grad_logits` then carries d(loss)/d(p_c4 ... p_c1, cp) in the output
 * layout of stcd_cf_output_info (make_prediction, models/ChangeFormer.py:1151-1157: conv - ReLU - BatchNorm - conv per scale).  Off
 * (the default, multi_scale_train == "False", trainer.py:311): only cp's gradient is read and the heads' backward launches are not
 * part of the plan.  Changing the setting invalidates the plan: call stcd_configure (and re-query stcd_workspace_bytes) afterwards. */
int stcd_cf_set_aux_backward(stcd_engine* e, int on);
/* FC-Siam family (SiamUnet_diff / _conc / _sub / _cross_conc, Unet), SNUNet_ECAM and ChangeFormerV6: with the whole backward in ONE stcd_backward call
 * (stage -1) the grouped weight gradients go out as soon as their last operand exists, on a low-priority side stream of the engine
 * beside the rest of the backward chain, and are joined before the call's work on `hip_stream` ends; the FC-Siam decoder's grids and
 * ChangeFormer's LDS-DMA group are planned smaller for that.  on == 0 plans and runs them on the caller's
 * stream -- what a data-parallel caller wants, which calls stage 0 and stage 1 separately to all-reduce the decoder's gradients
 * (nn.DataParallel's role in /root/reference/models/networks.py:85-116) in between (staged calls never use the side stream).  Default on
 * (environment STCD_WGRAD_SIDE=0: off).  Changing the setting invalidates the plan: call stcd_configure afterwards. */
int stcd_set_wgrad_side(stcd_engine* e, int on);

/* ---- parameter / buffer enumeration in the reference's registration order (state_dict compatibility,
 *      /root/reference/models/trainer.py:138,183; basic_model.py:35) ---- */
int stcd_num_params(const stcd_engine* e);
int stcd_param_info(const stcd_engine* e, int i, stcd_tensor_info* info);
int64_t stcd_param_floats(const stcd_engine* e);
int stcd_num_bn(const stcd_engine* e);
int stcd_bn_info_get(const stcd_engine* e, int i, stcd_bn_info* info);
int64_t stcd_bn_floats(const stcd_engine* e);

/* ---- shape binding: batch = image PAIRS per call; height/width as the reference accepts (any >= 16;
 *      ReplicationPad2d branch, SiamUnet_diff.py:149, for sizes not divisible by 16) ---- */
int stcd_configure(stcd_engine* e, int batch, int height, int width);
int64_t stcd_workspace_bytes(const stcd_engine* e);
int stcd_num_dropout(const stcd_engine* e);
int stcd_dropout_info_get(const stcd_engine* e, int i, stcd_dropout_info* info);
int64_t stcd_dropout_floats(const stcd_engine* e);
/* Dropout2d probability (reference: 0.2, SiamUnet_diff.py:20); 0 disables dropout in training mode. */
int stcd_set_dropout_p(stcd_engine* e, float p);

/* ---- forward: replaces Module.forward(x1,x2) (SiamUnet_diff.py:94-181) ----
 * x1,x2      fp32 NCHW [batch,in_ch,H,W]
 * params     flat fp32 parameters (layout: stcd_param_info)
 * bn_running flat fp32 running stats (layout: stcd_bn_info_get); updated in place when training != 0
 * dropout_masks  nullable; when given (training only) these masks are used verbatim (parity tests);
 *            when NULL and training, masks are drawn on-device from (dropout_seed) by a counter hash
 * logits     fp32 NCHW [batch,label_ch,H,W]
 */
int stcd_forward(stcd_engine* e, const float* x1, const float* x2, const float* params, float* bn_running,
                 const float* dropout_masks, uint64_t dropout_seed, int training, float* logits,
                 void* workspace, void* hip_stream);

/* ---- backward: replaces autograd through the module (loss.backward(), trainer.py:313) ----
 * Must follow a training-mode stcd_forward on the same workspace.  grads (flat fp32, layout of params) is
 * OVERWRITTEN with d loss / d param.  stage: -1 = whole backward; 0 = decoder half (its gradients are final
 * when it returns, so a data-parallel caller may start reducing them), 1 = encoder half.
 */
int stcd_backward(stcd_engine* e, const float* grad_logits, const float* params, float* grads,
                  void* workspace, int stage, void* hip_stream);
/* Inference loops: every stcd_forward repacks the filters from `params` into the workspace (the optimizer rewrites them each
 * training step).  A caller that KNOWS the parameters are unchanged declares it with a non-zero tag: while the tag, the workspace
 * and the parameter pointer stay the same the repack is skipped (0.3 ms of a 3.5 ms SegCD eval forward).  tag 0 (default): always
 * repack.  Changing the parameters without changing the tag is the caller's error. */
int stcd_set_weights_tag(stcd_engine* e, uint64_t tag);
/* [begin,end) element range of the flat gradient buffer that stage 0 finalises (the rest belongs to stage 1). */
int stcd_grad_stage_range(const stcd_engine* e, int stage, int64_t* begin, int64_t* end);

/* ---- test introspection: the named activation / gradient tensors of the CURRENT configuration inside the workspace, so a
 *      parity test can check every layer of a deep network IN PLACE (layer output against a convolution of the layer's own
 *      stored input, weight gradient against the stored input and output gradient ...) at per-op tolerance, independent of how
 *      rounding differences grow through the depth.  Filled for the STCD_ARCH_SEGCD* families ("<conv name>.in|.Y|.A|.dY|.dIn|.res")
 *      and the FC-Siam families ("<conv name>.in|.Y|.A.g0|.A.g1|.dY"; the activation per date); 0 tensors for SNUNet.  NHWC: element (n, y, x, ch) at offset_bytes + (((n*h + y)*w + x)*ld + ch) * elem_size.
 *      stcd_set_debug bit 0 (before stcd_configure): every layer writes its input gradient to a buffer of its own (the
 *      producer gathers it) instead of in place into the producer's gradient tensor, so ".dIn" survives the backward. */
typedef struct stcd_ws_tensor {
    char name[96];
    int64_t offset_bytes;
    int n, h, w, c, ld;
    int dtype;
} stcd_ws_tensor;
int stcd_set_debug(stcd_engine* e, int flags);
int stcd_ws_tensor_count(const stcd_engine* e);
int stcd_ws_tensor_get(const stcd_engine* e, int index, stcd_ws_tensor* out);

/* ---- measurement aid (bench.py): when enabled, every launch of the engine's kernel classes is bracketed by a
 *      hipEvent pair on the caller's stream; stcd_profile_read sums one class (it synchronises those events) and
 *      reports the ALGORITHMIC work of the same launches (each tensor counted once).
 *      Classes: 0 conv/dgrad  1 wgrad  2 bn-stats  3 bn+relu+dropout(+pool)  4 bn-backward reduce
 *               5 bn-backward apply  6 pool/fusion backward  7 packing. */
#define STCD_PROFILE_CLASSES 8
int stcd_profile_enable(stcd_engine* e, int on);
int stcd_profile_read(stcd_engine* e, int klass, double* total_ms, int64_t* launches, double* flops, double* bytes);
/* the same records grouped by kernel NAME (a substring of the name rocprofv3 prints, e.g. "k_conv_mfma<4>") */
int stcd_profile_num_kernels(const stcd_engine* e);
int stcd_profile_kernel(stcd_engine* e, int i, char* name, int name_cap, double* total_ms, int64_t* launches, double* flops,
                        double* bytes);

/* ---- losses, fused forward+backward: replace cross_entropy (losses.py:6-21) and
 *      cd_loss(sigmoid(x),y) == BCE_DICE (losses.py:24-34; train_pse_cd.py:227-228,436-462) ----
 * loss_out: one fp32 on the device.  dlogits nullable.  scratch: >= stcd_loss_scratch_bytes() bytes.
 */
int64_t stcd_loss_scratch_bytes(void);
int stcd_loss_ce(const float* logits, const int64_t* target, int batch, int classes, int64_t hw, int ignore_index,
                 float* loss_out, float* dlogits, void* scratch, void* hip_stream);
/* from_logits bit 0 set: x are logits, the loss is cd_loss(sigmoid(x), y) and dx is d loss / d logits (fused form used by
 * the script-shaped loop); clear: x are probabilities, exactly cd_loss(x, y) with dx = d loss / d probability.
 * bit 1 set: the Dice term alone (class Dice, train_pse_cd.py:436-447), without the BCE term. */
int stcd_loss_bce_dice(const float* x, const float* target, int64_t numel, int from_logits, float* loss_out, float* dx,
                       void* scratch, void* hip_stream);
/* contrastive loss of the semi-supervised stage (replaces contrastive_loss, /root/reference/train_stcd.py:334-385):
 * pred fp32 [2*numel_half] PROBABILITIES, first half = change prediction of the real pairs (cd), second half = of the
 * pseudo pairs (pse); labels int64 [numel_half] each.  loss = masked MSE(pse, cd) over (cd_label == pse_label) + masked
 * MSE(pse, |cd - 1|) over the rest, each divided by (count + 1e-8); dpred (nullable) = d loss / d pred, both halves. */
int stcd_loss_contrastive(const float* pred, const int64_t* cd_label, const int64_t* pse_label, int64_t numel_half, float* loss_out,
                          float* dpred, void* scratch, void* hip_stream);
/* ---- metric: replaces SegmentationMetric.genConfusionMatrix (train_pse_cd.py:361-368) without the
 *      per-step .cpu() sync (train_pse_cd.py:231).  cm[2*label+pred] += count; cm is 4 int64 on the device.
 *      pred = argmax over classes (classes==2) or logit > 0 (classes==1, i.e. sigmoid > 0.5). */
int stcd_confusion_update(const float* logits, const int64_t* target, int batch, int classes, int64_t hw,
                          int64_t* cm, void* hip_stream);

/* ---- optimizer: one launch of torch.optim.Adam (decoupled == 0: weight decay added to the gradient) or
 *      torch.optim.AdamW (decoupled != 0) over the flat parameter / gradient buffers, amsgrad off
 *      (replaces optimizer_G.step(): models/trainer.py:46-50,312-314; train_pse_cd.py:431,239).
 *      step = 1-based update count (bias correction); exp_avg / exp_avg_sq: fp32 state, zero-initialised. */
int stcd_adam_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, int64_t step,
                   double lr, double beta1, double beta2, double eps, double weight_decay, int decoupled, void* hip_stream);

/* The same update for a CAPTURED training step (hipGraph): the step-dependent scalars come from device memory, so one captured launch
 * serves every replay.  stcd_adam_hyper fills a HOST array of 8 floats for (step, lr, ...) with exactly the scalars stcd_adam_step
 * forms (the caller keeps it in pinned memory and makes its copy to hyper_dev8 part of the graph); stcd_adam_step_dev reads them. */
int stcd_adam_hyper(int64_t step, double lr, double beta1, double beta2, double eps, double weight_decay, float* hyper_host8);
int stcd_adam_step_dev(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t numel, const float* hyper_dev8,
                       int decoupled, void* hip_stream);

/* ---- pseudo-change pair synthesis on the device (replaces the file-based assembly of data/dataset.py:468-482 plus
 *      ToTensor/Normalize :499-500 and the paired cutout :24-57; the reference holds NO generator arithmetic, so the
 *      blend below is this library's own specification -- oracle/pseudo_ref.py restates it, parity is unpinned).
 *      img_a, donor: uint8 [B,H,W,3] (HWC, device); mask: uint8 [B,H,W] (>= 1 means building); change: uint8 [B]
 *      (tile is in the change list); alpha: nullable fp32 [B] blend strength (NULL = 1: donor replaces A inside the
 *      mask, as the reference's in-painted file does); erase_xywh: nullable int32 [B,4] cutout rectangle (w == 0: none),
 *      filled with per-pixel uniform values from `seed`, identical in A and B, label 255 there (host mean3/std3).
 *      x1, x2: fp32 [B,3,H,W] normalised; c_label / s_label_a / s_label_b: int64 [B,H,W] (the last two nullable). */
int stcd_pseudo_pair(const uint8_t* img_a, const uint8_t* donor, const uint8_t* mask, const uint8_t* change, const float* alpha,
                     const int32_t* erase_xywh, uint64_t seed, int batch, int height, int width, const float* mean3,
                     const float* std3, float* x1, float* x2, int64_t* c_label, int64_t* s_label_a, int64_t* s_label_b,
                     void* hip_stream);

/* ---- photometric augmentation on the device: counterpart of the host-side PIL/torchvision path of the datasets
 *      (/root/reference/data/dataset.py:488-495 ColorJitter(0.5,0.5,0.5,0.25) w.p. 0.5, RandomGrayscale(0.2), blur() :120-124)
 *      on normalised fp32 NCHW images [n_images,3,H,W] already on the device (e.g. the x1/x2 of stcd_pseudo_pair).
 *      params: device fp32 [n_images][8] = {jitter_on, brightness, contrast, saturation, hue, gray_on, sigma (0: no blur), 0};
 *      torchvision's float functional formulas, fixed order brightness -> contrast -> saturation -> hue, then grayscale, then a
 *      separable Gaussian (radius ceil(3 sigma), replicated edges).  The reference holds no device arithmetic for this:
 *      parity is unpinned beyond the per-op formulas (oracle/pseudo_ref.py restates them; tests pin them to PIL).
 *      out must not alias x; scratch >= stcd_augment_scratch_bytes() bytes. */
int64_t stcd_augment_scratch_bytes(int n_images, int height, int width);
int stcd_augment(const float* x, const float* params, int n_images, int height, int width, const float* mean3, const float* std3,
                 float* out, void* scratch, int64_t scratch_bytes, void* hip_stream);

/* ---- per-op entry points (NHWC, activation dtype per `dtype`); used by the parity tests.
 *      Geometry is the engine's generic "tap list" convolution: see DESIGN.md section 3. ---- */
typedef struct stcd_conv_geom {
    int32_t n, hi, wi, ci, ldi;         /* input: images, spatial, channels (K per tap), pixel stride (elements) */
    int32_t hm, wm, in_stride;          /* output-position grid per image; input pixel = m*in_stride + tap */
    int32_t ho, wo, out_stride, oy0, ox0; /* output buffer dims; output pixel = m*out_stride + (oy0,ox0) */
    int32_t co, ldo;                    /* output channels, output pixel stride (elements) */
    int32_t ntaps;
    int8_t dy[9], dx[9];
    int8_t pad_[2];
} stcd_conv_geom;

/* w: fp32 [ntaps][ci][co]; bias: fp32 [co] or NULL; impl: 0 = reference FMA kernel, 1 = MFMA, the kernel the engine would
 * pick (bf16 only: small-channel persistent kernel, resident-filter kernel, tap-list GEMM kernel, else the generic one),
 * 2 = generic MFMA kernel, 3 = resident-halo kernel (wide 3x3 layers, opt-in in the engine), 6 = LDS-DMA kernel (Ci % 64 == 0,
 * Co % 256 == 0, an even number >= 4 of (64-channel chunk, tap) K-tiles: 256 positions x 256 channels per persistent 8-wave block,
 * `buffer_load ... lds` staging with counted vmcnt -- what the engine runs for ChangeFormer's 256 -> 256 decoder-head layers,
 * /root/reference/models/ChangeFormerBaseNetworks.py:85-120) */
int stcd_op_conv(int dtype, int impl, const stcd_conv_geom* g, const void* in, const float* w, const float* bias,
                 void* out, void* scratch, int64_t scratch_bytes, void* hip_stream);
/* dw: fp32 [ntaps][ci][co], overwritten; impl: 0 = reference FMA kernel, 1 = MFMA tile kernel (4: its 64 x 32-channel tile
 * allowed for Ci >= 64, as the SNUNet engine runs it; 5: its 64 x 64-channel tile, an opt-in variant), 3 = MFMA position-GEMM kernel
 * (one tap, Ci >= 64, Co >= 64: what the engine runs for 1x1 convs and the phases of 2x2 stride-2 transposed convs),
 * 7 = LDS-DMA kernel (stride-1 full-map tap lists with Ci % 256 == 0, Co % 256 == 0, maps >= 64 wide; dY plain or one sub-pixel
 * phase of a map twice as large: block = (position split, tap, 256 x 256 channel tile) -- ChangeFormer's decoder-head layers) */
int stcd_op_wgrad(int dtype, int impl, const stcd_conv_geom* g, const void* in, const void* dout, float* dw,
                  void* scratch, int64_t scratch_bytes, void* hip_stream);
int64_t stcd_op_scratch_bytes(const stcd_conv_geom* g);

/* ---- per-op entry points of the normalisation / pooling / fusion kernels (NHWC, activation dtype per `dtype`,
 *      channels a power of two >= 8); each runs exactly the launch sequence the engine runs for that layer step.
 *      n = images (date-0 images first, then date-1 images when groups == 2: BatchNorm statistics stay per date,
 *      /root/reference/models/SiamUnet_diff.py:99 and :123 call the same bn module once per date). ---- */
typedef struct stcd_map_geom {
    int32_t n, h, w, c, groups;
} stcd_map_geom;
int64_t stcd_op_ew_scratch_bytes(const stcd_map_geom* g);
/* a = dropout_mask * relu(BatchNorm2d(y)) (nn.BatchNorm2d + F.relu + nn.Dropout2d, SiamUnet_diff.py:19-20,99), optional
 * pool = F.max_pool2d(a, 2, 2) (:101).  training != 0: batch statistics per group, running stats updated in place
 * (momentum 0.1, unbiased variance; group 0 then group 1); else running statistics.  mask: nullable fp32 [n][c].
 * stat (out): fp32 [groups][4][c] = mean, 1/sqrt(var+eps), scale, shift -- what the backward needs. */
int stcd_op_bn_act(int dtype, const stcd_map_geom* g, const void* y, int ldy, const float* gamma, const float* beta,
                   float* running_mean, float* running_var, const float* mask, int relu, int training, void* a, int lda,
                   void* pool, int ldp, float* stat, void* scratch, int64_t scratch_bytes, void* hip_stream);
/* the same for the last conv of an encoder level (groups == 2, training, ReLU), both dates in one pass, also writing the
 * bi-temporal skip fused = |a1 - a2| (fuse_mode 0, SiamUnet_diff.py:150) or a2 - a1 (1, SiamUnet_sub.py:150): [n/2] images.
 * a == NULL: the activations themselves are not written (only pool and fused; see stcd_op_skip_bwd) */
int stcd_op_bn_act_pair(int dtype, const stcd_map_geom* g, const void* y, int ldy, const float* gamma, const float* beta,
                        float* running_mean, float* running_var, const float* mask, int fuse_mode, void* a, int lda, void* pool,
                        int ldp, void* fused, int ldf, float* stat, void* scratch, int64_t scratch_bytes, void* hip_stream);
/* backward of stcd_op_bn_act (training): da -> dy (may alias da), dgamma, dbeta summed over the groups */
int stcd_op_bn_act_bwd(int dtype, const stcd_map_geom* g, const void* da, int ldda, const void* y, int ldy, const float* stat,
                       const float* mask, int relu, void* dy, int lddy, float* dgamma, float* dbeta, void* scratch,
                       int64_t scratch_bytes, void* hip_stream);
int stcd_op_maxpool(int dtype, const stcd_map_geom* g, const void* a, int lda, void* pool, int ldp, void* hip_stream);
/* da (+)= dpool routed to the first maximum of each 2x2 window (torch's tie rule) */
int stcd_op_maxpool_bwd(int dtype, const stcd_map_geom* g, const void* a, int lda, const void* dpool, int ldp, void* da, int ldda,
                        int accumulate, void* hip_stream);
/* skip fusion of the two dates (groups == 2): d[n/2 images] = |a1 - a2| (mode 0) or a2 - a1 (mode 1), and its gradient */
/* F.max_pool2d(x, 3, 2, 1) of the ResNet stem (models/resnet.py:176) and its gradient (SegCD): g describes the INPUT map (h, w even);
 * pool / dpool are [n, h/2, w/2, c]; idx (n*(h/2)*(w/2)*c bytes) receives / provides the winning window position of every element */
int stcd_op_maxpool3(int dtype, const stcd_map_geom* g, const void* a, int lda, void* pool, int ldp, void* idx, void* hip_stream);
int stcd_op_maxpool3_bwd(int dtype, const stcd_map_geom* g, const void* idx, const void* dpool, int ldp, void* da, int ldda, void* hip_stream);
/* cross_conc's grouped 3x3 convolution (models/SiamUnet_crossconc.py:14-18,24-29: the two dates' channels interleaved, one group per
 * channel pair) on the two dates as they sit stacked in the batch dimension (g->groups == 2, a: [2 * n, h, w, c]; out / dout: [n, h, w, c];
 * w: the reference tensor [c][2][3][3], b: [c] or NULL); the backward writes both dates' gradients (da like a) and dw. */
int64_t stcd_op_pairdw_scratch_bytes(const stcd_map_geom* g);
int stcd_op_pairdw(int dtype, const stcd_map_geom* g, const void* a, int lda, const float* w, const float* b, void* out, int ldo,
                   void* hip_stream);
int stcd_op_pairdw_bwd(int dtype, const stcd_map_geom* g, const void* a, int lda, const void* dout, int lddo, const float* w, void* da,
                       int ldda, float* dw, void* scratch, int64_t scratch_bytes, void* hip_stream);
int stcd_op_fuse(int dtype, int mode, const stcd_map_geom* g, const void* a, int lda, void* d, int ldd, void* hip_stream);
int stcd_op_fuse_bwd(int dtype, int mode, const stcd_map_geom* g, const void* a, int lda, const void* dd, int ldd, void* da, int ldda,
                     void* hip_stream);
/* nn.ReplicationPad2d((0, w - w0, 0, h - h0)) in place on a [n,h,w] map whose valid region is h0 x w0
 * (SiamUnet_diff.py:149), and its gradient (the replicas' gradients fold into the edge pixel) */
int stcd_op_rep_pad(int dtype, const stcd_map_geom* g, void* d, int ld, int h0, int w0, void* hip_stream);
int stcd_op_rep_pad_bwd(int dtype, const stcd_map_geom* g, void* dd, int ld, int h0, int w0, void* hip_stream);
/* backward of an encoder level's last conv in one pass (groups == 2): da = max-pool gradient of dpool + skip-fusion
 * gradient of dd, then the BatchNorm backward of it: dy, dgamma, dbeta.  a == NULL: the activations were not stored
 * (stcd_op_bn_act_pair with a == NULL; the engine's default plan for diff / sub) and are recomputed from y, stat and mask with
 * the forward's arithmetic -- channels a multiple of 8 and a power of two <= 256 */
int stcd_op_skip_bwd(int dtype, int mode, const stcd_map_geom* g, const void* a, int lda, const void* y, int ldy, const void* dd,
                     int ldd, const void* dpool, int ldp, const float* stat, const float* mask, void* da, int ldda, void* dy,
                     int lddy, float* dgamma, float* dbeta, void* scratch, int64_t scratch_bytes, void* hip_stream);

/* ---- per-op entry points of the ChangeFormer kernels (contiguous tensors: tokens [rows, c] / maps [n, h, w, c] NHWC, activation dtype
 *      per `dtype`); each runs the launch sequence the engine runs for that step.  Dropout inside an op: site 0 of `seed`
 *      (stcd_cf_site above; DropPath of stcd_op_cf_resid_drop: site 1), p == 0 disables it. ---- */
int64_t stcd_op_cf_scratch_bytes(int64_t rows, int channels, int n, int q_tokens, int kv_tokens);
/* OverlapPatchEmbed.proj / Attention.sr as GEMMs (ChangeFormer.py:207-208, 315): col [n*ho*wo, ldc], column ci*k*k + ky*k + kx */
int stcd_op_cf_im2col(int dtype, const void* x, void* col, int ldc, int n, int h, int w, int c, int k, int stride, int pad, void* hip_stream);
int stcd_op_cf_col2im(int dtype, const void* dcol, int ldc, void* dx, int n, int h, int w, int c, int k, int stride, int pad, int accumulate,
                      void* hip_stream);
/* nn.LayerNorm(c, eps) per row (ChangeFormer.py:209,317,478,486); stats fp32 [rows][2] = mean, 1/sqrt(var + eps) */
int stcd_op_cf_layernorm(int dtype, const void* x, void* y, const float* gamma, const float* beta, float* stats, int64_t rows, int c, float eps,
                         void* hip_stream);
/* dx = [add] + LayerNorm-backward(dy [+ dy2]); dgamma, dbeta fp32 [c] overwritten */
int stcd_op_cf_layernorm_bwd(int dtype, const void* dy, const void* dy2, const void* x, const float* stats, const float* gamma, const void* add,
                             void* dx, float* dgamma, float* dbeta, void* scratch, int64_t rows, int c, void* hip_stream);
int stcd_op_cf_colsum(int dtype, const void* x, int64_t rows, int c, float* out, void* scratch, void* hip_stream);
/* Attention.forward between the q / kv projections and proj (ChangeFormer.py:347-354): q [n, q_tokens, heads*d], kv [n, kv_tokens, 2*heads*d]
 * (k then v, head-major channels), out [n, q_tokens, heads*d], lse fp32 [n, heads, q_tokens]; scale = d^-0.5; attn_drop probability p */
int stcd_op_cf_attention(int dtype, const void* q, const void* kv, void* out, float* lse, int n, int q_tokens, int kv_tokens, int heads, int d,
                         float p, uint64_t seed, void* hip_stream);
int64_t stcd_op_cf_attention_scratch_bytes(int n, int q_tokens, int kv_tokens, int heads, int d);
int stcd_op_cf_attention_bwd(int dtype, const void* q, const void* kv, const void* out, const void* dout, const float* lse, void* dq, void* dkv,
                             void* scratch, int n, int q_tokens, int kv_tokens, int heads, int d, float p, uint64_t seed, void* hip_stream);
/* Mlp between fc1 and fc2 (ChangeFormer.py:289-292, 517-523): u = depthwise3x3(h) + b, a = dropout(GELU(u)); w = [ch][1][3][3] */
int stcd_op_cf_dwgelu(int dtype, const void* h, void* u, void* a, const float* w, const float* b, int n, int height, int width, int ch, float p,
                      uint64_t seed, void* hip_stream);
int64_t stcd_op_cf_dwgelu_scratch_bytes(int n, int height, int width, int ch);
/* da is overwritten with the gated gradient; dh = d(h); dw [ch][9], db [ch] overwritten */
int stcd_op_cf_dwgelu_bwd(int dtype, const void* h, const void* u, void* da, void* dh, const float* w, float* dw, float* db, void* scratch, int n,
                          int height, int width, int ch, float p, uint64_t seed, void* hip_stream);
/* Block.forward's x + drop_path(dropout(y)) (ChangeFormer.py:505-509) over [n, rows_per_img, c]; backward != 0: out = d(y) from y = d(out) */
int stcd_op_cf_resid_drop(int dtype, const void* x, const void* y, void* out, int n, int64_t rows_per_img, int c, float p, float path_p,
                          uint64_t seed, int backward, void* hip_stream);
/* F.interpolate(mode="bilinear", align_corners=False) (ChangeFormer.py:1585,1591) [n,h,w,c] -> [n,out_h,out_w,c]; backward != 0: src is the
 * gradient of the large map, dst receives the gradient of the small one; accumulate != 0: dst += */
int stcd_op_cf_bilinear(int dtype, const void* src, void* dst, int n, int h, int w, int out_h, int out_w, int c, int accumulate, int backward,
                        void* hip_stream);
/* op: 0 PReLU(x; alpha[0]) | 1 x * dropout(p) | 2 relu(x) | 3 x * [y > 0] | 4 alpha_f * x + beta_f * y (y nullable) */
int stcd_op_cf_elementwise(int dtype, int op, const void* x, const void* y, void* out, int64_t rows, int c, const float* alpha, float alpha_f,
                           float beta_f, float p, uint64_t seed, void* hip_stream);
/* dy = dz * (y > 0 ? 1 : alpha[0]); dalpha[0] = sum dz * y * [y <= 0] */
int stcd_op_cf_prelu_bwd(int dtype, const void* dz, const void* y, void* dy, const float* alpha, float* dalpha, void* scratch, int64_t rows, int c,
                         void* hip_stream);

#ifdef __cplusplus
}
#endif
#endif /* STCD_HIP_H */
