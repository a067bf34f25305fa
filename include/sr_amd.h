/*
 * sr_amd.h -- C ABI of libsr_amd.so: the MI355X (gfx950) kernels under the RRDBNet / FDConv /
 * RealESRGANer Python API of cdrookie/Super_resolution.
 *
 * The reference has no FFI on this path: it is Python on torch operators.  Each entry point
 * below therefore names the torch call (reference file:line) whose work it takes over.  Paths
 * are relative to /root/reference; "FDConv.py" is
 * FDConv-main/FDConv-main/FDConv_detection/mmdet_custom/FDConv.py and "RRDBNet" is
 * basicsr==1.4.2 basicsr/archs/rrdbnet_arch.py (not vendored; call site
 * train/train_realESRGAN_fdconv_1.py:100).
 *
 * Conventions
 *   - every pointer is DEVICE memory on the current HIP device unless it says "host";
 *   - activations are NHWC ("pixel-major, channels innermost") or channel-blocked, in `dtype`
 *     (SR_F32 or SR_BF16); see sr_view for the addressing;
 *   - no entry point allocates, frees or synchronises; work is enqueued on `stream`
 *     (a hipStream_t passed as void*); scratch memory is supplied by the caller;
 *     the exceptions are the *_create / *_destroy pairs which build small lookup tables;
 *   - return value: 0 = SR_OK, negative = error (see sr_strerror); nothing is launched on error.
 */
#ifndef SR_AMD_H
#define SR_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SR_OK 0
#define SR_ERR_ARG (-1)       /* bad argument / unsupported shape */
#define SR_ERR_WORKSPACE (-2) /* caller's scratch buffer too small */
#define SR_ERR_HIP (-3)       /* a HIP runtime call failed (see sr_last_hip_error) */
#define SR_ERR_STATE (-4)     /* object not prepared for this shape */

#define SR_F32 0
#define SR_BF16 1

#define SR_MAX_SEG 8 /* max tensors in a virtual channel concat */

/* A view of `C` channels of an activation tensor.  Two layouts:
 *   NHWC (cbs = 0):  element (b, pixel, c), cc = c0 + c, lives at  b*bs + pixel*ld + cc  (pixel = h*W + w, in
 *     elements); ld = channels of the underlying tensor, bs = H*W*ld (bs = 0 selects exactly that).
 *   channel-blocked [B][C/cb][H][W][cb] (cbs != 0), cb = ld a power of two >= 8:
 *     b*bs + (cc / cb)*cbs + pixel*cb + (cc % cb),  cbs = cb*H*W, bs = C*H*W.
 *     The executor keeps trunk tensors in blocks of 32 channels (a conv K-chunk is then one contiguous 64-byte
 *     run per pixel, a tile row one contiguous run) and the 2x / 4x tensors in blocks of 8 (an image row of 8
 *     channels is contiguous for the FFT passes). */
typedef struct {
    void* p;     /* base of the underlying tensor */
    int32_t ld;  /* elements between consecutive pixels */
    int32_t c0;  /* first channel of the view */
    int32_t C;   /* channels in the view */
    int32_t cbs; /* elements between consecutive channel blocks (0: NHWC) */
    int64_t bs;  /* elements between consecutive samples (0: NHWC, H*W*ld) */
} sr_view;

/* The input of a conv: a virtual concat (torch.cat in RRDBNet's dense block) of up to
 * SR_MAX_SEG views.  When `bands` is set every view also has its three FBM low-pass planes
 * (same shape, at the conv's resolution).  `ups` = 1 reads x through a nearest 2x upsample
 * (F.interpolate(scale_factor=2, mode='nearest') in RRDBNet.forward): x is then (H/2, W/2). */
typedef struct {
    int32_t nseg;
    int32_t bands;
    int32_t ups;
    int32_t _pad;
    sr_view x[SR_MAX_SEG];
    sr_view l2[SR_MAX_SEG];
    sr_view l4[SR_MAX_SEG];
    sr_view l8[SR_MAX_SEG];
} sr_concat;

/* Epilogue of a conv:  y = acc + bias ; activation ; y = y*s1 + r1 ; y = y*s2 + r2
 * (RRDBNet: lrelu(conv(x)); ResidualDenseBlock x5*0.2+x; RRDB out*0.2+x; feat + body_feat).
 * activation: slope != NULL: y < 0 ? slope[channel] * y : y  (nn.PReLU(num_parameters=cout) of SRVGGNetCompact,
 * srvgg_arch.py:38,50; zeros give ReLU, a constant LeakyReLU); else lrelu ? leaky_relu(y, 0.2) : none. */
typedef struct {
    const float* bias; /* [Cout] or NULL */
    int32_t lrelu;
    int32_t _pad;
    sr_view r1;  /* p == NULL: unused */
    float s1;
    float _pad1;
    sr_view r2;
    float s2;
    float _pad2;
    const float* slope; /* [Cout] per-channel negative slopes or NULL */
} sr_epilogue;

/* Parameters of one conv layer of RRDBNet, as float32 device pointers straight into the
 * nn.Module's parameter storage (state_dict layout of SURVEY 8b), plus two derived buffers. */
typedef struct {
    int32_t cin, cout;
    int32_t active;   /* 1: FDConv path (has dft_weight); 0: plain nn.Conv2d (FDConv.py:641-642) */
    int32_t loc_taps; /* KSM_Local.conv kernel size (FDConv.py:282) */
    const float* bias;     /* (cout) or NULL */
    const void* wpacked;   /* plain: weight repacked by sr_pack_conv_weight to [9][cout][cin] in dtype */
    const float* basis;    /* active: sr_fdconv_basis output [4][cout][cin][9] */
    const float* fc_w;     /* KSM_Global.fc.weight (16,cin) */
    const float *bn_w, *bn_b, *bn_rm, *bn_rv; /* KSM_Global.bn.* (16) */
    const float *relu_s, *relu_b;             /* KSM_Global.relu.{scale,bias} (1) */
    const float *ch_w, *ch_b;                 /* channel_fc (cin,16),(cin) */
    const float *fl_w, *fl_b;                 /* filter_fc  (cout,16),(cout) */
    const float *sp_w, *sp_b;                 /* spatial_fc (9,16),(9) */
    const float *kn_w, *kn_b;                 /* kernel_fc  (4,16),(4) */
    const float* gate_w[3];                   /* FBM.freq_weight_conv_list.k.weight (1,cin,3,3) */
    const float* gate_b[3];                   /* ....bias (1) */
    const float* loc_w;                       /* KSM_Local.conv.weight (cout*9,1,taps) */
    const float *ln_w, *ln_b;                 /* KSM_Local.norm.{weight,bias} (cin) */
    int64_t wpacked_lo_off;                   /* plain: elements from wpacked to the low-order parts (0: none) */
} sr_layer_params;

const char* sr_strerror(int code);
int sr_last_hip_error(void); /* hipError_t of the last failed HIP call on this thread */
int sr_version(void);

/* ---- layout ------------------------------------------------------------------------------
 * NCHW (torch default) <-> NHWC.  pack folds basicsr's pixel_unshuffle(x, r) (RRDBNet.forward,
 * scale 2 -> r=2, scale 1 -> r=4; r=1: none): dst is (B, H/r, W/r, C*r*r), channel c*r*r+dy*r+dx. */
int sr_pack_nchw(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C, int H, int W, int r,
                 void* stream);
/* The same layout change with the channel count padded to `cpad` (a multiple of 8; zeros beyond C), bf16 destination
 * (B, H, W, cpad): a 1- or 3-channel image as one 32-channel K-chunk of the MFMA convolution kernels.  Replaces the
 * implicit NCHW read of the first nn.Conv2d of SRVGGNetCompact (Real-ESRGAN-master/realesrgan/archs/srvgg_arch.py:34, 56-58). */
int sr_pack_nchw_padded(const void* src, int src_dtype, void* dst, int B, int C, int H, int W, int cpad, void* stream);
int sr_unpack_nhwc(const void* src, int src_dtype, void* dst, int dst_dtype, int B, int C, int H, int W,
                   void* stream);
/* nn.Conv2d weight (cout,cin,3,3) f32 -> [9][cout][cin] in `dtype` (what the conv kernels read); dst_lo (may be NULL)
 * receives w - dst in the same layout (split weights, see sr_fdconv_weightgen) */
int sr_pack_conv_weight(const float* w, void* dst, void* dst_lo, int dtype, int cout, int cin, void* stream);

/* ---- FBM low-pass (FDConv.py:427-457: rfft2 -> 3 box masks -> irfft2) ----------------------
 * A plan holds the twiddles, the nearest-resized masks (F.interpolate at FDConv.py:446) and the
 * kept-column lists for one (H, W).  create/destroy allocate; everything else does not. */
typedef struct sr_fbm_plan sr_fbm_plan;
int sr_fbm_plan_create(int H, int W, sr_fbm_plan** out);
void sr_fbm_plan_destroy(sr_fbm_plan* plan);
/* host-only (no HIP call): the resized masks as HOST arrays rows (3*H bytes), cols (3*(W/2+1) bytes) */
int sr_fbm_masks_host(int H, int W, uint8_t* rows_host, uint8_t* cols_host);
size_t sr_fbm_workspace_bytes(const sr_fbm_plan* plan, int B, int C);
/* x: (B,H,W,C) view, or (B,H/2,W/2,C) when ups=1.  Writes L_k = lowpass_k(x) for k=2,4,8 into
 * l2/l4/l8 (B,H,W,C) and, if pooled != NULL, the spatial mean of every channel
 * (F.adaptive_avg_pool2d at FDConv.py:643) into pooled[b*pooled_ld + c] (float).
 * A band whose view has p == NULL is not produced (at least one must be given): the adjoint of the band mix in the
 * training path needs one band per call, and the synthesis passes then run for that band only. */
int sr_fbm_lowpass(const sr_fbm_plan* plan, int dtype, int B, sr_view x, int ups, sr_view l2, sr_view l4, sr_view l8,
                   float* pooled, int pooled_ld, void* workspace, size_t workspace_bytes, void* stream);

/* The same low-pass with NO allocation anywhere and no caller workspace.  No counterpart in the reference: it exists for
 * callers that cannot allocate (code under HIP graph capture, tight serving loops, harnesses that forbid hipMalloc).
 * The intermediates of the passes live in a device array of SR_STATIC_SCRATCH_BYTES that the library holds on every
 * GPU it runs on until the process ends; shapes whose sr_fbm_workspace_bytes exceed it return SR_ERR_WORKSPACE
 * (64 MiB: e.g. 1 x 64 channels x 384 x 384 or 4 x 32 x 256 x 256).  Calls that overlap on one GPU share the array
 * and race: issue them on one stream.  The plan must have been created beforehand (plan creation allocates). */
#define SR_STATIC_SCRATCH_BYTES (64u << 20)
size_t sr_fbm_static_scratch_bytes(void);
int sr_fbm_lowpass_static(const sr_fbm_plan* plan, int dtype, int B, sr_view x, int ups, sr_view l2, sr_view l4, sr_view l8,
                          float* pooled, int pooled_ld, void* stream);

/* ---- FBM gates (FDConv.py:460-461): g_k = 2*sigmoid(conv3x3(x; gate_w[k]) + gate_b[k]) ------
 * gates: float (B,H,W,4) = {g1,g2,g3,unused}. */
int sr_fbm_gates(int dtype, int B, int H, int W, const sr_concat* in, const sr_layer_params* lp, float* gates,
                 void* stream);

/* The same gates, accumulated segment by segment (bf16 activations only).  The gate conv is linear in its input
 * channels, so inside a dense block (torch.cat of x, x1..x4 feeding conv1..conv5) every tensor is read ONCE, when it
 * is produced, for all the layers that will consume it, instead of once per consuming layer:
 *   logits_j += conv3x3(seg; gate_w_j[:, c0:c0+C])   for each consumer j,
 * and the consumer for which `seg` is the last input gets gates = 2*sigmoid(logits) written out.
 * No counterpart call in the reference (FDConv.py:460-461 recomputes the conv over the whole concat per layer). */
#define SR_MAX_GATE_CONSUMERS 8
typedef struct {
    const float* w[3]; /* gate weights of this consumer at the segment's first channel: [C][3*3] per k */
    const float* b[3]; /* gate biases (used when first != 0) */
    float* logits;     /* (B,H,W,3) float accumulator of this consumer; read unless first, written unless gates != NULL */
    float* gates;      /* non-NULL: seg is this consumer's last segment -> write 2*sigmoid(logits) here (B,H,W,4) */
    int32_t first;     /* non-zero: seg is this consumer's first segment (start from the bias) */
} sr_gate_consumer;
int sr_fbm_gates_accumulate(int dtype, int B, int H, int W, sr_view seg, int nconsumers, const sr_gate_consumer* consumers,
                            void* stream);

/* ---- FDConv weight path --------------------------------------------------------------------
 * basis: the inverse real 2-D DFT (torch.fft.irfft2 at FDConv.py:683) of the alpha-scaled
 * spectrum restricted to each of the 4 frequency groups (FDConv.py:661-680), rearranged to
 * [4][cout][cin][3*3].  gmap: uint8 (3cout, 3cin/2+1) group ids.  Run when dft_weight changes. */
size_t sr_fdconv_basis_workspace_bytes(int cout, int cin);
int sr_fdconv_basis(const float* dft_weight, const uint8_t* gmap, float alpha, int cout, int cin, float* basis,
                    void* workspace, size_t workspace_bytes, void* stream);
/* per-sample weights W_b = sa*ca*fa*hr*sum_g ka_g*basis_g (FDConv.py:644-652,692-697) from the
 * pooled means (B, pooled_ld) -> wdyn [B][9][cout][cin] in dtype. */
size_t sr_fdconv_weightgen_workspace_bytes(int B, int cin, int cout);
int sr_fdconv_weightgen(int dtype, int B, const float* pooled, int pooled_ld, const sr_layer_params* lp, int train,
                        float momentum, void* wdyn, void* wdyn_lo, void* workspace, size_t workspace_bytes, void* stream);
/* train = 0: KSM_Global's BatchNorm on its running statistics (module.eval()).  train != 0: on the statistics of the batch
 * of B pooled vectors (nn.BatchNorm2d in train(), FDConv.py:245-258 under model.train() at
 * train/train_realESRGAN_fdconv_1.py:108); lp->bn_rm / lp->bn_rv are then UPDATED IN PLACE with `momentum` (torch: 0.1;
 * unbiased variance), B must be 2..512 (torch raises for one value per channel).  The caller bumps num_batches_tracked.
 * workspace: afterwards holds, per sample, the modulation vectors as floats  ca[cin] | ln[cin] | fa[cout] | sa[9] | ka[4]
 * (sample stride = (2*cin + cout + 13 + 3) & ~3 floats) -- KSM_Global's four attentions and KSM_Local's LayerNorm output. */
/* wdyn_lo (may be NULL): the same layout, receives w - wdyn, i.e. the part of each weight that rounding to `dtype` lost.
 * In bf16 the pair (wdyn, wdyn_lo) carries 16 significant bits; sr_conv3x3 multiplies by both when given w_lo_offset.
 * Why: a rounding error in a weight is coherent over every pixel it touches, and through 349 stacked layers with
 * data-dependent weights it -- not the bf16 activations -- dominated the bf16 output error. */

/* ---- 3x3 convolution, stride 1, zero pad 1 (F.conv2d at FDConv.py:699 / nn.Conv2d.forward) ---
 * weights: [9][cout][cin] per sample; w_bstride = elements between samples (0: shared weights);
 * w_lo_offset != 0: weights + w_lo_offset holds the low-order parts of the same weights (see sr_fdconv_weightgen).
 * With in->bands the A operand is FBM(x) = g1*x + (g2-g1)*L2 + (g3-g2)*L4 + (1-g3)*L8
 * (FDConv.py:456-479) formed on the fly from `gates`.  out: (B,H,W,cout) view. */
int sr_conv3x3(int dtype, int B, int H, int W, const sr_concat* in, const float* gates, const void* weights,
               long w_bstride, long w_lo_offset, int cout, const sr_epilogue* ep, sr_view out, void* stream);

/* ---- network ends (bandwidth-bound) ----------------------------------------------------------
 * SRVGGNetCompact's tail (Real-ESRGAN-master/realesrgan/archs/srvgg_arch.py:65-68): dst (B,C,H*r,W*r) NCHW =
 * nn.PixelShuffle(r)(src) + F.interpolate(base, scale_factor=r, mode='nearest'); src: (B,H,W,>=C*r*r) view in
 * src_dtype (the last conv's output), base: the network input (B,C,H,W) NCHW in io_dtype. */
int sr_pixel_shuffle_add(sr_view src, int src_dtype, const void* base, void* dst, int io_dtype, int B, int C, int H, int W,
                         int r, void* stream);
/* F.interpolate(x, size=(Hd,Wd), mode='bilinear', align_corners=False) on `planes` = B*C planes (NCHW)
 * (train/train_realESRGAN_fdconv_1.py:115, train/test_realESRGAN_fdconv.py:51).  pre_sigmoid: nn.Sigmoid applied to the
 * source first (CustomRRDBNet head, train/train_realESRGAN_fdconv.py:111-115); post_expm1: dst = max(expm1(v*post_scale), 0)
 * (de-normalisation of train/test_realESRGAN_fdconv.py:55-57). */
int sr_bilinear_resize(const void* src, int src_dtype, void* dst, int dst_dtype, long planes, int Hs, int Ws, int Hd, int Wd,
                       int pre_sigmoid, int post_expm1, float post_scale, void* stream);
/* loss = mean(w * |bilinear(act(pred), (Hd,Wd)) - target|): nn.L1Loss of fdconv_1.py:116 (weighted = 0) or the weighted L1
 * of train_realESRGAN_fdconv.py:124-126 (w = clamp(10 * target, 0.1, 1)); float32.  loss: one float on the device.
 * grad_pred (may be NULL): d loss / d pred, (planes,Hs,Ws), overwritten.  workspace: sr_resize_l1_workspace_bytes(). */
size_t sr_resize_l1_workspace_bytes(void);
int sr_resize_l1_loss(const float* pred, const float* target, long planes, int Hs, int Ws, int Hd, int Wd, int weighted,
                      int pre_sigmoid, float* loss, float* grad_pred, void* workspace, size_t workspace_bytes, void* stream);

/* ---- perceptual (VGG-feature L1) loss of the Real-ESRGAN recipe -------------------------------------------------------
 * (Real-ESRGAN-master/options/train_realesrgan_x4plus.yml:133-149: basicsr PerceptualLoss, vgg19, features before ReLU at
 * conv1_2, conv2_2, conv3_4, conv4_4, conv5_4 with weights 0.1, 0.1, 1, 1, 1, criterion l1, use_input_norm; BASELINE
 * configs[3]).  The 3x3 convolutions of the extractor run on sr_conv3x3 (forward and, with transposed weights, backward:
 * the extractor is frozen, so no weight gradient exists); these are the pieces between them, all on contiguous NHWC
 * tensors in `dtype`:
 *   sr_relu_pool           dst (B,Ho,Wo,C) = max_pool2d(relu(src (B,Ho*pool,Wo*pool,C)), pool), pool = 1 (ReLU only) or 2
 *   sr_relu_pool_backward  dx = its backward at src given dy (+ `add`, may be NULL: a tapped layer's own loss gradient)
 *   sr_l1_feature_loss     *loss += weight * mean|a - b| (float atomics); grad_a (may be NULL) = weight * sign(a - b) / n
 *   sr_vgg_prep            adjoint = 0: src float image (B,Cs,H,W), Cs = 1 (grey, repeated to RGB) or 3 -> dst (B,H,W,3)
 *                          in dtype, (x - mean[c]) / std[c] (use_input_norm; both NULL: no normalisation);
 *                          adjoint != 0: dst holds the gradient (B,H,W,3), src RECEIVES the float gradient (B,Cs,H,W)
 *   sr_bilinear_resize_backward  adjoint of sr_bilinear_resize (float32): dsrc (planes,Hs,Ws) OVERWRITTEN from ddst
 *                          (planes,Hd,Wd) -- the perceptual term is taken on the prediction resized to the target's size. */
int sr_relu_pool(int dtype, int B, int Ho, int Wo, int C, int pool, const void* src, void* dst, void* stream);
int sr_relu_pool_backward(int dtype, int B, int Ho, int Wo, int C, int pool, const void* src, const void* dy, const void* add, void* dx,
                          void* stream);
int sr_l1_feature_loss(int dtype, long n, const void* a, const void* b, float weight, float* loss, void* grad_a, void* stream);
int sr_vgg_prep(int dtype, int B, int Cs, int H, int W, const float* src, void* dst, const float* mean3, const float* std3, int adjoint,
                void* stream);
int sr_bilinear_resize_backward(const float* ddst, float* dsrc, long planes, int Hs, int Ws, int Hd, int Wd, void* stream);

/* RealESRGANer.enhance's image ends (Real-ESRGAN-master/realesrgan/utils.py:197-216 and :225-254), so that the
 * host <-> device transfers carry the INTEGER image (a 16384^2 RGB result is 0.8 GB as uint8, 3.2 GB as float32):
 * sr_image_to_planes: src (H,W,channels) HWC device image, src_kind 0 = uint8, 1 = uint16, 2 = float32 ->
 *   dst (3,H,W) float planes, plane j = (float)src[..., cj] / full  (img.astype(np.float32) / 255 | 65535 at :197-203 and the
 *   channel shuffles of cv2.cvtColor at :204-216: BGR -> RGB = (2,1,0), gray -> RGB = (0,0,0), alpha = (3,3,3)).
 * sr_image_from_planes: src = R, G, B planes of the network output (element strides plane_stride / row_stride: the
 *   post_process crop is a strided view) -> dst (H,W,dst_channels) HWC uint8 (sixteen = 0) / uint16 (sixteen = 1) image,
 *   channels dst_c0.. : clamp_(0, 1) (:225), RGB -> BGR (:226) or, gray != 0, COLOR_BGR2GRAY (:228, 0.114 B + 0.587 G + 0.299 R),
 *   x 255 | 65535, round half to even, cast (:251-254). */
int sr_image_to_planes(const void* src, int src_kind, int H, int W, int channels, int c0, int c1, int c2, float full, float* dst,
                       void* stream);
int sr_image_from_planes(const void* src, int src_dtype, int H, int W, long plane_stride, long row_stride, int gray, int sixteen,
                         void* dst, int dst_channels, int dst_c0, void* stream);

/* ---- training: backward of one conv / FDConv layer ------------------------------------------
 * The reference obtains these from torch autograd through FDConv.py:640-722 (loss.backward() at
 * train/train_realESRGAN_fdconv_1.py:117).  Layer forward:  y = bias + conv3x3(a, W_b),  a = FBM(x) (see sr_conv3x3).
 *
 * sr_conv_weight_transpose: wt[b][8-t][i][o] = w[b][t][o][i]  ([B][9][cout][cin] -> [B][9][cin][cout], taps mirrored): with
 *   it  da = sr_conv3x3(dy, wt)  is the gradient wrt the conv operand (adjoint of F.conv2d at FDConv.py:699). */
int sr_conv_weight_transpose(const void* w, void* wt, int dtype, int B, int cout, int cin, void* stream);
/* dw[b][t][o][i] = sum_p dy[b,p,o] * a[b, p + off(t), i]  (float32, OVERWRITTEN), a formed from `in` / `gates` exactly as
 * sr_conv3x3 forms it.  per_sample = 0: one [9][cout][cin] gradient summed over the batch (nn.Conv2d.weight.grad of a plain
 * layer); != 0: [B][9][cout][cin] (gradient wrt the per-sample dynamic weights, input of the small weight path's backward).
 * dbias (may be NULL): [cout] = sum over batch and pixels of dy (bias.grad), OVERWRITTEN.  Sums are fp32 atomics: the
 * order of the partial sums, hence the last bits, vary from run to run. */
int sr_conv3x3_wgrad(int dtype, int B, int H, int W, const sr_concat* in, const float* gates, sr_view dy, int cout,
                     int per_sample, float* dw, float* dbias, void* stream);
/* backward of the band mix a = g1 x + (g2-g1) L2 + (g3-g2) L4 + (1-g3) L8 (FDConv.py:456-479) given da:
 *   dx = g1 da, dl2 = (g2-g1) da, dl4 = (g3-g2) da, dl8 = (1-g3) da   (all (B,H,W,C) views in dtype)
 *   dlogits (B,H,W,4) float = {dz1, dz2, dz3, 0}: gradients wrt the gate-conv outputs z_k (g_k = 2 sigmoid(z_k)),
 *   dz1 = sum_c da (x - L2) * g1 (1 - g1/2), dz2 with (L2 - L4), dz3 with (L4 - L8). */
int sr_fbm_backward(int dtype, int B, int H, int W, sr_view x, sr_view l2, sr_view l4, sr_view l8, const float* gates, sr_view da,
                    sr_view dx, sr_view dl2, sr_view dl4, sr_view dl8, float* dlogits, int accumulate, void* dlogits_bf16x32,
                    void* stream);
/* accumulate != 0: the four branch gradients are ADDED to what dx / dl2 / dl4 / dl8 hold -- a tensor of a dense block feeds
 * several layers (torch.cat in basicsr's ResidualDenseBlock), the consumers' branch gradients are summed in place and the
 * adjoint low-pass then runs once per produced tensor on the sums (the low-pass is linear).
 * dlogits_bf16x32 (may be NULL): also receives the logit gradients as a (B,H,W,32) bf16 tensor {dz1, dz2, dz3, 0 ...}: the
 * dy operand with which sr_conv3x3_wgrad (cout = 32) computes the gate convs' weight and bias gradients on the matrix cores. */
/* backward of the three gate convs (FDConv.py:460-461) and assembly of the layer's input gradient:
 *   dgate_w [3][cin][9], dgate_b [3] (float32, OVERWRITTEN; same atomics remark as sr_conv3x3_wgrad); both NULL: not
 *   computed (the bf16 training path gets them from sr_conv3x3_wgrad with the logit gradients in the place of dy)
 *   dx_out = dx_direct + a2 + a4 + a8 + conv3x3^T(dlogits; gate_w) + dpooled[b][c] / (H W)
 * a_k = sr_fbm_lowpass plane k of dl_k (the box low-pass is self-adjoint: a real symmetric mask between a unitary
 * transform pair); dpooled (B, dpooled_ld) float or NULL: gradient wrt the pooled means (F.adaptive_avg_pool2d at
 * FDConv.py:643) that the weight path's backward returns.  dx_out may alias dx_direct.
 * a2, a4, a8 with p == NULL (all three): those terms are left out (a tensor with several consumers collects them once,
 * after its last consumer, through sr_grad_combine). */
int sr_fbm_gates_backward(int dtype, int B, int H, int W, sr_view x, const float* dlogits, const sr_layer_params* lp,
                          sr_view dx_direct, sr_view a2, sr_view a4, sr_view a8, const float* dpooled, int dpooled_ld,
                          sr_view dx_out, float* dgate_w, float* dgate_b, void* stream);

/* out = mask * sum_j scales[j] * terms[j]  over (B,H,W,C) views, nterms <= 8;  mask = 1 where act > 0, else `slope`
 * (act.p == NULL: no mask).  The sum of a tensor's gradient contributions -- direct branch, adjoint low-pass planes,
 * residual paths (x5 * 0.2 + x, out * 0.2 + x of basicsr's RDB / RRDB) -- with the backward of the LeakyReLU(0.2) that
 * produced the tensor folded in (autograd's AddBackward / LeakyReluBackward nodes in the reference). */
int sr_grad_combine(int dtype, int B, int H, int W, int C, int nterms, const sr_view* terms, const float* scales, sr_view act,
                    float slope, const float* bc, int bc_ld, float bc_scale, sr_view out, void* stream);
/* bc (may be NULL): float (B, bc_ld) values added as bc_scale * bc[b][c] to every pixel before the mask: the summed
 * pooled-mean gradients of a tensor's consumers (backward of F.adaptive_avg_pool2d at FDConv.py:643, bc_scale = 1 / (H W)).
 *
 * The three transposed gate convs of a layer (backward of FDConv.py:460 with respect to its input) as one 32 -> cin
 * convolution for sr_conv3x3 over the (B,H,W,32) bf16 logit gradients of sr_fbm_backward:
 * hi / lo [9][cin][32] bf16 = mirrored taps of gate_w[k] in input channel k < 3, zeros elsewhere (w = hi + lo). */
int sr_pack_gate_weight_transposed(const sr_layer_params* lp, void* hi, void* lo, void* stream);
/* adjoint == 0: dst (B,2Hs,2Ws,C) = F.interpolate(src (B,Hs,Ws,C), scale_factor=2, mode='nearest') (RRDBNet.forward);
 * adjoint != 0: dst (B,Hs,Ws,C) = 2 x 2 sums of src (B,2Hs,2Ws,C) (its backward). */
int sr_upsample2x(int dtype, int B, int Hs, int Ws, int C, sr_view src, sr_view dst, int adjoint, void* stream);

/* backward of the weight path: gradients of sr_fdconv_weightgen's output with respect to every parameter it reads and to
 * the pooled means.  The reference obtains them from torch autograd through KSM_Global (FDConv.py:245-258), KSM_Local
 * (FDConv.py:302-321), the spectral scatter + irfft2 (FDConv.py:661-684) and the aggregation (FDConv.py:692-697).
 *   dw     [B][9][cout][cin] float: gradient with respect to the per-sample weights (sr_conv3x3_wgrad, per_sample != 0)
 *   mods   the workspace sr_fdconv_weightgen left behind for the same B, pooled and parameters (ca | ln | fa | sa | ka)
 *   gmap   uint8 (3 cout, 3 cin / 2 + 1) frequency-group ids, alpha as for sr_fdconv_basis; lp->basis must be current
 *   train  as in the forward call (BatchNorm on batch statistics; the running statistics are NOT touched here)
 *   grads  one float32 output per parameter, each OVERWRITTEN, shapes of the state_dict entries:
 *          dft_weight (3 cout, 3 cin / 2 + 1, 2) -- by linearity  d dft[r][c] = alpha w_c / (9 cout cin) rfft2(S_G(r,c))[r][c]
 *          with S_g = sum_b ka[b][g] dA[b] (adjoint of the C2R transform, w_c = 1 on the self-conjugate columns, else 2)
 *   dpooled (B, dpooled_ld) float: gradient with respect to the pooled means (both KSM branches); OVERWRITTEN, or with
 *          accumulate != 0 ADDED to (a tensor of a dense block feeds several layers).
 * Sums over the batch inside wp_elem / wp_ln are float atomics (last bits vary from run to run). */
typedef struct {
    float* dft_weight;
    float* fc_w;
    float *bn_w, *bn_b;
    float *relu_s, *relu_b;
    float *ch_w, *ch_b, *fl_w, *fl_b, *sp_w, *sp_b, *kn_w, *kn_b;
    float* loc_w;
    float *ln_w, *ln_b;
} sr_weight_path_grads;
size_t sr_fdconv_weightgen_backward_workspace_bytes(int B, int cin, int cout);
int sr_fdconv_weightgen_backward(int B, const float* pooled, int pooled_ld, const sr_layer_params* lp, int train, const float* mods,
                                 const float* dft_weight, const uint8_t* gmap, float alpha, const float* dw,
                                 const sr_weight_path_grads* grads, float* dpooled, int dpooled_ld, int accumulate,
                                 void* workspace, size_t workspace_bytes, void* stream);

/* ---- whole network (RRDBNet.forward) ------------------------------------------------------- */
typedef struct sr_rrdbnet sr_rrdbnet;
int sr_rrdbnet_create(int num_in_ch, int num_out_ch, int scale, int num_feat, int num_block, int num_grow_ch,
                      sr_rrdbnet** out);
void sr_rrdbnet_destroy(sr_rrdbnet* net);
int sr_rrdbnet_num_layers(const sr_rrdbnet* net);
/* layer order = state_dict order: conv_first, body.i.rdbj.conv1..5, conv_body, conv_up1, conv_up2, conv_hr, conv_last */
int sr_rrdbnet_set_layer(sr_rrdbnet* net, int index, const sr_layer_params* lp);
/* builds the FBM plans for (H,W),(2H,2W),(4H,4W) of an input (B,C,Hin,Win); allocates tables */
int sr_rrdbnet_prepare(sr_rrdbnet* net, int Hin, int Win);
size_t sr_rrdbnet_workspace_bytes(const sr_rrdbnet* net, int dtype, int B, int Hin, int Win);
/* x: NCHW (B,num_in_ch,Hin,Win) in io_dtype; y: NCHW (B,num_out_ch,Hin*scale... ) in io_dtype */
int sr_rrdbnet_forward(sr_rrdbnet* net, int dtype, int io_dtype, int B, int Hin, int Win, const void* x, void* y,
                       void* workspace, size_t workspace_bytes, void* stream);
/* dtype SR_BF16: the FDConv layers' matrix-core operands are fp16 inside the executor -- one fp16 weight per MAC plus the 9-tap sum of
 * the rounding remainders on the centre tap, the band-mixed activations packed to fp16 (saturating) -- with float32 accumulation;
 * SR_AMD_OPERAND_F16=0 in the environment: bf16 hi + lo weight pairs with a bf16 operand (what sr_fdconv_weightgen + sr_conv3x3
 * compute when called one by one).  Tensors in memory are bf16 either way. */
/* per-phase kernel-time attribution for bench.py: enables hipEvent timing around kernel groups
 * on the next forward; read back (ms) after a stream sync.  groups (arrays of 8): 0 conv (trunk) 1 lowpass (trunk)
 * 2 gates 3 weightgen 4 layout 5 head convs (conv_up1, conv_up2, conv_hr, conv_last) 6 lowpass of the head tensors
 * 7 unused.  enable: 0 off, 1 every group, 2 the conv groups only (an event pair costs ~4 us of stream
 * time: 11 ms per config-2 forward with every group, 3 ms with the conv groups alone) */
int sr_rrdbnet_profile(sr_rrdbnet* net, int enable);
int sr_rrdbnet_profile_read(sr_rrdbnet* net, float* ms8, long* launches8);

#ifdef __cplusplus
}
#endif
#endif /* SR_AMD_H */
