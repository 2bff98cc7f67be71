/* cryovit_hip.h -- C ABI of libcryovit_hip.so (gfx950 / MI355X).
 *
 * The reference (VivianDLi/CryoVIT) is pure Python and has NO FFI boundary of its own: its plug-in
 * surface is Hydra `_target_` instantiation plus two duck-typed Python protocols.  This
 * header is therefore the boundary a maintainer would bind with ctypes to replace the third-party
 * arithmetic the reference delegates to torch / xformers / the dinov2 hub model:
 *
 *   cvx_preprocess_patches + cvx_vit_encode
 *                     replace   model.forward_features(vec)["x_norm_patchtokens"] + reshape/permute/half
 *                               src/cryovit/run/dino_features.py:53-61  and the CPU bicubic resize of
 *                               src/cryovit/datasets/vit_dataset.py:117-123 (fused into the first kernel)
 *   cvx_head_forward (= cvx_gemm_bf16 with fp16 operands, cvx_groupnorm_f16, cvx_conv3d_f16, cvx_conv3_out_fused)
 *                     replaces  CryoVIT.forward_volume + sigmoid          src/cryovit/models/cryovit.py:36-49
 *                               and the masked Dice reductions            src/cryovit/models/base_model.py:99-110,
 *                                                                         src/cryovit/models/metrics.py:30-43
 *
 * Conventions: every function returns 0 on success and a negative code on failure
 * (cvx_last_error() gives the message, thread-local); all pointers named *_dev / in descriptors are DEVICE
 * pointers owned by the caller; nothing allocates, frees or synchronises inside a call (graph-capturable);
 * every launch goes to the caller's hipStream_t.  One handle per stream / GPU; handles are not thread-safe.
 * The op-level entry points (cvx_gemm_bf16 ... cvx_dice_sums) are the kernels the two high-level calls are
 * built from; they are exported so the parity tests can check each kernel against the CPU oracle.
 */
#ifndef CRYOVIT_HIP_H
#define CRYOVIT_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct ihipStream_t* hipStream_t; /* same as <hip/hip_runtime_api.h> */

#define CVX_OK 0
#define CVX_ERR_ARG (-1)
#define CVX_ERR_HIP (-2)

const char* cvx_last_error(void);
int cvx_version(void);
/* name of the device the library would launch on ("gfx950...") or "" when no HIP device is visible */
int cvx_device_arch(char* buf, int buflen);
/* Process-global tuning switches for in-process A/B measurements (tools/, tests): every default is the shipped, measured-best setting and
 * no product path sets one.  Returns non-zero (cvx_last_error) for an unknown name or a value the library was not built with (the product
 * library holds the shipped kernel variants only; the rejected ones live in the -DCVX_ABLATION build).  Names: use_gemm256, gemm256_variant,
 * gemm_stagger, gemm_tail_split, gemm_tail_max, gemm_tail_tile, gemm_tail_deep, gemm_resid_reverse, gemm_resid_stagger, tile_group_l,
 * attn_variant, attn_xcd_remap, attn_mfma_prio, attn_half_tile, win_attn_prefetch, win_attn_x32, conv_halo, conv_wide, convt_small,
 * ln_policy -- each is described where it is defined (csrc/gemm.hip, attention.hip, hiera.hip) and in DESIGN.md s.4 / s.8. */
int cvx_set_option(const char* name, int value);
/* diagnostic: per-wave cycle sums {load, load-barrier, mma, mma-barrier} x 8 waves written by gemm256 variant 20 */
int cvx_debug_read_gemm256(unsigned long long* out32);
/* diagnostic: cycle stamps of the persistent tile kernel (variant 29, -DCVX_ABLATION builds only; zeros otherwise):
 * [wave group 2][tile 8][6] = K-loop start, K-loop end, epilogue start, epilogue end, next tile released, first K tile done */
int cvx_debug_read_gemm256p(unsigned long long* out96);

/* ---------------------------------------------------------------------------------------------------
 * Dense GEMM  C[M,N] = A[M,K] * W[N,K]^T, bf16 operands (K contiguous, leading dims in elements),
 * fp32 accumulation on v_mfma_f32_16x16x32_bf16, fused epilogue.  A must be allocated (not necessarily
 * valid) up to a multiple of 256 rows; W / bias / gamma are packed to n_pad (multiple of 16; 128 for
 * SWIGLU / VT) rows and k_pad (multiple of 64) columns.  Replaces torch.nn.Linear / Conv2d(14,14) /
 * Conv3d(k=1) / ConvTranspose3d calls made by the hub ViT and by cryovit/models/cryovit.py:18-34,74-77.
 * ------------------------------------------------------------------------------------------------- */
enum cvx_epilogue {
    CVX_EPI_BF16 = 0,      /* out bf16 [M][ldc]       = acc + bias                                   */
    CVX_EPI_BF16_GELU = 1, /* out bf16 [M][ldc]       = gelu_erf(acc + bias)                         */
    CVX_EPI_SWIGLU = 2,    /* out bf16 [M][ldc], N/2 cols = silu(a) * b   (W rows interleaved by 8)  */
    CVX_EPI_RESID = 3,     /* out fp32 [M][ldc]      += gamma * (acc + bias)     (LayerScale+residual)*/
    CVX_EPI_PATCH = 4,     /* out fp32 token stream: row slice*ntp+tok0+p = acc + bias + pos[1+p]     */
    CVX_EPI_VT = 5,        /* out bf16 V^T [slice][head][64][kp] = acc + bias   (for cvx_attention)   */
    CVX_EPI_CONVT = 6,     /* out bf16 [D][2H][2W][cout] pixel-shuffle of N = 4*cout, optional GELU   */
    CVX_EPI_F32 = 7,       /* out fp32 [M][ldc]       = gamma * (acc + bias)   (written, not accumulated)  */
    CVX_EPI_RESID_HL = 8   /* the residual stream as a bf16 PAIR: x = hi + lo (out = hi, out2 = lo, both bf16 [M][ldc]);
                              x += gamma * (acc + bias); hi = bf16(x), lo = bf16(x - hi); stat_part[n / 64][m] = (sum, sum of
                              squares) of the new x over 64-column slots (N a multiple of 64, N <= n_pad).  hi is the next GEMM's A
                              operand: see ln_rowstat                                                            */
};

enum { CVX_DTYPE_BF16 = 0, CVX_DTYPE_F16 = 1 };

typedef struct cvx_gemm_desc {
    int epilogue;
    const void* a; long lda;   /* bf16 [M_alloc][lda]  */
    const void* w; long ldw;   /* bf16 [n_pad][ldw]    */
    long m, n, n_pad, k_pad;   /* valid rows / cols; padded N, K */
    void* out; long ldc;
    const float* bias;         /* [n_pad] */
    const float* gamma;        /* RESID: [n_pad] */
    const float* pos; long ldpos; /* PATCH: fp32 [1+npatch][ldpos] */
    int npatch, ntp, tok0;     /* PATCH: patches per slice, padded tokens per slice, first patch token */
    int heads, kp;             /* VT: heads, padded key count (multiple of 64) */
    int H, W, cout, act;       /* CONVT: input plane size, C_out, act (0 none / 1 GELU) */
    int dtype;                 /* CVX_DTYPE_BF16 (0, default): bf16 operands / 16-bit outputs; CVX_DTYPE_F16: fp16 operands
                                  and outputs (BF16, BF16_GELU and CONVT epilogues only: the segmentation head) */
    int convt_up_z;            /* CONVT: 0 = kernel/stride (1,2,2), N = 4*C_out (the head); 1 = (2,2,2), N = 8*C_out with
                                  n = ((iz*2+i)*2+j)*C_out + o, output [2D][2H][2W][C_out] (UNet3D upconv, unet3d.py:166-170) */
    /* LayerNorm folded into the GEMM (BF16, BF16_GELU, SWIGLU, VT; bf16 operands): A = bf16(x) un-normalised (the hi array), W
     * packed as bf16(W * ln_gamma), bias = fp32 [2][n_pad]: b' = b + W ln_beta, then cs[n] = sum_k W'[n][k];
     * ln_rowstat = fp32 [M][2] = (rstd, -mean * rstd) per row (cvx_rowstat_finalize / cvx_split_stream).  The epilogue then
     * starts from  rstd * acc + (-mean * rstd) * cs[n] + b'[n]  instead of acc + bias[n].  NULL: plain epilogue. */
    const float* ln_rowstat;
    /* RESID_HL: lo array, fp32 partial row sums [n_pad / 64][stat_rows][2], stat_rows >= M rounded up to 256 */
    void* out2; float* stat_part; long stat_rows;
} cvx_gemm_desc;

int cvx_gemm_bf16(const cvx_gemm_desc* d, hipStream_t stream);

/* Measurement hook: while set, every cvx_gemm_bf16 launch with this epilogue (also inside cvx_vit_encode) is bracketed by
 * hipEventRecord(start[i]) / hipEventRecord(stop[i]) on the launch stream, i = 0 .. capacity-1.  NULL arrays disable it. */
int cvx_set_gemm_event_hook(int epilogue, void** start_events, void** stop_events, int capacity);
int cvx_get_gemm_event_count(void);

/* Dilated 3x3x3 "same" convolution, dilation (dil,1,1), channels-last FP16 volume in[D][H][W][C] ->
 * out[D][H][W][cout] fp16 = act(conv + bias), as an implicit GEMM (K = tap*C + c) on v_mfma_f32_16x16x32_f16 (the head
 * stores fp16: the reference runs it under fp16 autocast, and fp16 storage keeps the logits within 1e-2 of the fp32 CPU
 * path where bf16 storage costs 0.2 -- DESIGN.md s.2).  w is fp16 [n_pad][k_pad] with
 * k = ((kz*3+ky)*3+kx)*C + c.  zero_page: >= 16 zero bytes on the device (source of padded taps).
 * Replaces nn.Conv3d(c1,c2,3,padding="same",dilation=(d,1,1)) -- cryovit/models/cryovit.py:70-73,30-33. */
typedef struct cvx_conv3d_desc {
    const void* in; const void* w; const float* bias; const void* zero_page; void* out;
    int C, D, H, W, dil, cout, n_pad, k_pad, act;
} cvx_conv3d_desc;
int cvx_conv3d_f16(const cvx_conv3d_desc* d, hipStream_t stream);

/* nn.Conv3d(C, cout, 2, stride=2) (UNet3D's pooling convolution, models/unet3d.py:127-131) on the same descriptor: in fp16
 * [D][H][W][C] (D, H, W even, C a power of two >= 8) -> out fp16 [D/2][H/2][W/2][cout] = act(conv + bias); w fp16 [n_pad][8*C]
 * with k = ((iz*2+iy)*2+ix)*C + c; dil is ignored. */
int cvx_conv2s2_f16(const cvx_conv3d_desc* d, hipStream_t stream);

/* UNet3D helpers (csrc/unet.hip).  cvx_concat_channels_f16: out[v][0..Ca) = a[v][:], out[v][Ca..Ca+Cb) = b[v][:] (torch.cat along
 * channels, unet3d.py:64; Ca, Cb multiples of 8).  cvx_pointwise_out_f16: the 1x1x1 output layer + clip + sigmoid
 * (unet3d.py:45,69-71,96): logits / probs fp32 [nvox] (either nullable) from in fp16 [nvox][C], w fp32 [C], C <= 64, a multiple of 8. */
int cvx_concat_channels_f16(const void* a, int Ca, const void* b, int Cb, void* out, long nvox, hipStream_t stream);
int cvx_pointwise_out_f16(const void* in, const float* w, float bias, float* logits, float* probs, long nvox, int C, hipStream_t stream);

/* LayerNorm over the last dim of an fp32 token stream -> bf16 (GEMM operand).  eps inside the sqrt.
 * x fp32 [rows][ldx], out bf16 [rows][ldo].  Replaces nn.LayerNorm(C, eps=1e-6) in the hub ViT blocks. */
int cvx_layernorm_bf16(const float* x, long ldx, const float* w, const float* b, void* out, long ldo, long rows,
                       int C, float eps, hipStream_t stream);

/* Multi-head attention, head_dim 64, no mask/dropout:  O = softmax(Q K^T) V  per (slice, head).
 *   qk   bf16 [slices*ntp (+64 rows slack)][ldqk] : columns [0,C) = Q pre-scaled by head_dim^-0.5 * log2(e)
 *                                                  (scores in log2 units: the kernel uses exp2),
 *                                                  [C,2C) = K, head h at columns h*64..h*64+63
 *   vt   bf16 [slices][heads][64][kp]  (kp = ntok rounded up to 64; columns >= ntok must be finite)
 *   out  bf16 [slices*ntp][ldo], head h at columns h*64...
 * ntok = valid tokens per slice (keys >= ntok are masked), ntp = padded tokens per slice (multiple of 8).
 * Replaces xformers.memory_efficient_attention inside the hub model (run/dino_features.py:58). */
int cvx_attention_bf16(const void* qk, long ldqk, const void* vt, void* out, long ldo, int slices, int heads,
                       int ntok, int ntp, int kp, hipStream_t stream);

/* The same attention reading V ROW-MAJOR from the buffer that holds Q and K: qkv bf16 [slices*ntp (+64 rows slack)][ld], columns
 * [0,C) = Q (log2 units), [C,2C) = K, [2C,3C) = V, C = heads*64 -- the output of ONE qkv GEMM with the plain row-major epilogue
 * (no V^T GEMM launch, no vt buffer).  ld % 64 == 0.  The kernel transposes V fragments on the LDS read (ds_read_b64_tr_b16). */
int cvx_attention_qkv_bf16(const void* qkv, long ld, void* out, long ldo, int slices, int heads, int ntok, int ntp,
                           hipStream_t stream);

/* Pre-processing fused with im2col: raw slices [b][H][W] (u8 -> /255, or f32), edge-pad to x16, bicubic
 * x14/16 (A = -0.75, align_corners = False, clamped taps), cut into 14x14 patches of ONE channel (the 3
 * input channels are identical copies -- vit_dataset.py:117-118 -- so the patch-embed weight is summed over
 * channels at pack time): out bf16 [b*hp*wp][k_pad], k = py*14+px, zero padded to k_pad.
 * Replaces VITDataset._dino_transform (vit_dataset.py:90-123) + the unfold inside Conv2d(3,C,14,14). */
int cvx_preprocess_patches(const void* slices, int is_u8, int b, int H, int W, void* out, int k_pad,
                           hipStream_t stream);

/* cls / register / padding rows of the token stream: x[s*ntp + 0] = cls + pos[0]; rows 1..n_reg = registers;
 * rows ntok..ntp-1 = 0.  (patch rows are written by CVX_EPI_PATCH.) */
int cvx_init_tokens(float* x, long ldx, const float* cls_pos0, const float* reg, int n_reg, int slices, int ntok,
                    int ntp, int C, hipStream_t stream);

/* Final LayerNorm + drop cls/registers + layout transform (run/dino_features.py:58-61):
 *   feats_f16  (nullable) fp16 [C][d_total][hp][wp], slices written at depth d0..d0+slices-1
 *   feats_cl   (nullable) fp16 [slices][hp][wp][C]   channels-last copy for the segmentation head (same rounding as feats_f16)
 *   tokens_f32 (nullable) fp32 [slices][hp*wp][C]    "x_norm_patchtokens" of the encoder protocol */
int cvx_final_norm_features(const float* x, long ldx, const float* w, const float* b, float eps, int slices,
                            int ntp, int tok0, int hp, int wp, int C, void* feats_f16, long d_total, long d0,
                            void* feats_cl, float* tokens_f32, hipStream_t stream);

/* The ViT path keeps its residual stream as a PAIR of bf16 arrays, x = hi + lo (hi = bf16(x) doubles as the A operand of the
 * next GEMM, which applies the LayerNorm in its epilogue: cvx_gemm_desc.ln_rowstat; lo = bf16(x - hi)).
 * cvx_final_norm_features_hl: cvx_final_norm_features reading that pair (ld in elements of either array).
 * cvx_split_stream: fp32 rows -> (hi, lo) and rowstat[row] = (rstd, -mean * rstd) of nn.LayerNorm(C, eps) over the row: the
 *   hand-over from the patch-embedding GEMM (fp32) to the first block.
 * cvx_rowstat_finalize: the partial row sums a CVX_EPI_RESID_HL GEMM left in stat_part[nslot = C / 64][part_rows][2] ->
 *   rowstat[rows][2] for the next GEMM. */
int cvx_final_norm_features_hl(const void* xh, const void* xl, long ld, const float* w, const float* b, float eps, int slices,
                               int ntp, int tok0, int hp, int wp, int C, void* feats_f16, long d_total, long d0,
                               void* feats_cl, float* tokens_f32, hipStream_t stream);
int cvx_split_stream(const float* x, long ldx, void* xh, void* xl, long ld, float* rowstat, long rows, int C, float eps,
                     hipStream_t stream);
int cvx_rowstat_finalize(const float* part, int nslot, long part_rows, float* rowstat, long rows, int C, float eps,
                         hipStream_t stream);
/* (hi, lo) -> fp32 rows, x = hi + lo (exact): the hand-over from a folded stretch of a stream to kernels that take fp32. */
int cvx_merge_stream(const void* xh, const void* xl, long ld, float* x, long ldx, long rows, int C, hipStream_t stream);

/* im2col of already-resized 3-channel images x fp32 [b][3][Hi][Wi] (Hi, Wi multiples of 14) for the
 * encoder-protocol entry point forward_features(x) (run/dino_features.py:58): out bf16 [b*hp*wp][k_pad],
 * k = c*196 + py*14 + px (the flattening of Conv2d(3,C,14,14).weight), zero padded to k_pad >= 588. */
int cvx_im2col_patches(const float* x, int b, int Hi, int Wi, void* out, int k_pad, hipStream_t stream);

/* fp16 [C][D][h][w] (the HDF5 `dino_features` layout) -> fp16 channels-last [D][h][w][C] (a transpose: exact) */
int cvx_features_to_channels_last(const void* feats_f16, void* out_cl, int C, long nvox, hipStream_t stream);

/* GroupNorm over a channels-last fp16 volume x[nvox][C], G groups (<= 512), biased variance, eps inside sqrt
 * (nn.GroupNorm(G, C, eps=1e-3) -- cryovit.py:69).  Three launches: per-block partial sums, fixed-order reduction (no
 * atomics: results are bitwise reproducible) that also leaves the per-channel affine coefficients, apply -> fp16 out.
 * stats: fp32 scratch of 2*G*(1 + CVX_GN_BLOCKS) floats (stats[0..2G) = sum | sum of squares per group after the call;
 * block partials and the 2*C coefficients behind them). */
#define CVX_GN_BLOCKS 1024
#define CVX_GN_MAX_GROUPS 512
int cvx_groupnorm_f16(const void* x, const float* w, const float* b, void* out, float* stats, long nvox, int C,
                       int G, float eps, hipStream_t stream);
/* The same with an activation fused into the apply pass (act: 0 none, 1 exact GELU).  G = C is
 * nn.InstanceNorm3d(C, eps, affine=True) of the UNet3D baseline (models/unet3d.py:19-25,127-140). */
int cvx_groupnorm_act_f16(const void* x, const float* w, const float* b, void* out, float* stats, long nvox, int C,
                           int G, float eps, int act, hipStream_t stream);
/* The same writing into a column block of a WIDER channels-last buffer: out points at the block's first column, rows are ldo
 * elements apart (ldo >= C, a multiple of 8); out2_dense (nullable) receives a second, dense [nvox][C] copy.  torch.cat along the
 * channels of a UNet3D synthesis block (models/unet3d.py:64) then costs no pass of its own: both of its inputs are outputs of this op. */
int cvx_groupnorm_act_strided_f16(const void* x, const float* w, const float* b, void* out, long ldo, void* out2_dense, float* stats,
                                   long nvox, int C, int G, float eps, int act, hipStream_t stream);

/* Last layer of the head at full resolution, channels-last fp16 in[D][H][W][8]:
 *   conv3x3x3(8->1, w fp32 [27][8] tap-major) + bias, clip(+-5) -> logits fp32 (nullable), sigmoid -> probs fp32
 *   (nullable), and masked Dice partial sums (labels int8 nullable; dice must be zeroed by the caller):
 *   dice[0] += sum(y*p_hat), dice[1] += sum(y), dice[2] += sum(p_hat) over labels > -1, p_hat = (p >= mask_threshold): the
 *   threshold of DiceMetric (configs/model/metrics/dice_metric.yaml: 0.5) and of the uint8 mask are ONE parameter.
 * Replaces output_layer.2 + clip + sigmoid (cryovit.py:33,39,49) and the reductions of
 * base_model.py:99-110 / metrics.py:36-41.  (output_layer.0 + GELU runs through cvx_conv3d_f16.)
 * scratch: >= 3*CVX_DICE_BLOCKS floats (per-block partial sums, reduced in a fixed order: reproducible).
 * mask (nullable): uint8 [D][H][W] = (p >= mask_threshold), the binary segmentation PredictionWriter stores
 * (src/cryovit/models/callbacks.py:100-102), written here so that only 1 byte per voxel leaves the GPU. */
#define CVX_DICE_BLOCKS 4096
int cvx_conv3_out_fused(const void* in, const float* w, float bias, float* logits, float* probs, const int8_t* labels,
                        float* dice, float* scratch, uint8_t* mask, float mask_threshold, int D, int H, int W,
                        hipStream_t stream);

/* Masked Dice partial sums over existing predictions (same definition as above, threshold thr). */
int cvx_dice_sums(const float* probs, const int8_t* labels, float* dice, long n, float thr, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Training-side pieces of the head (csrc/train.hip).
 *
 * cvx_dice_loss_forward -- replaces DiceLoss.forward (/root/reference/src/cryovit/models/losses.py:17-32) applied to the
 *   masked predictions of BaseModel._masked_predict (models/base_model.py:91-112): over the n voxels with label > -1,
 *   out4 = { I = sum y p, Sy = sum y, Sp = sum p, loss = 1 - 2 I / (Sy + Sp + 1e-3) }.  probs fp32 (16-B aligned), labels int8
 *   in {-1, 0, 1}; scratch: >= 3*CVX_DICE_BLOCKS floats.  Block partials + fixed-order finalize: bitwise reproducible.
 * cvx_dice_loss_backward -- what autograd derives for that expression: grad[i] = grad_out * (-2 y_i / den + 2 I / den^2) for
 *   label > -1, 0 elsewhere (den = Sy + Sp + 1e-3; sums4 = the forward's out4, on the device).  through_sigmoid = 1 continues
 *   through p = sigmoid(clip(logit, -5, 5)) (models/cryovit.py:39,49): * p (1 - p), and 0 where |logits[i]| >= 5 (logits
 *   nullable: the clipped logits the forward stored).
 * cvx_adamw_step -- one torch.optim.AdamW step (the optimizer of BaseModel.configure_optimizers, models/base_model.py:57-63)
 *   over flat fp32 arrays (16-B aligned), in place, same order of operations as torch's single-tensor path:
 *   p *= 1 - lr wd; m += (1 - b1)(g - m); v = b2 v + (1 - b2) g^2; p -= (lr / (1 - b1^t)) m / (sqrt(v) / sqrt(1 - b2^t) + eps).
 * --------------------------------------------------------------------------------------------------- */
int cvx_dice_loss_forward(const float* probs, const int8_t* labels, long n, float* scratch, float* out4, hipStream_t stream);
int cvx_dice_loss_backward(const float* probs, const float* logits, const int8_t* labels, long n, const float* sums4, float grad_out,
                           int through_sigmoid, float* grad, hipStream_t stream);
int cvx_adamw_step(float* p, const float* g, float* m, float* v, long n, double lr, double beta1, double beta2, double eps,
                   double weight_decay, int step, hipStream_t stream);
/* cvx_focal_loss_forward / _backward -- FocalLoss.forward (/root/reference/src/cryovit/models/losses.py:35-64), i.e.
 * torchvision.ops.sigmoid_focal_loss(y_pred, y_true, alpha = (n - sum y) / n, gamma, reduction = "mean") over the voxels with
 * label > -1 (x: the model output the reference passes as the loss's "logits"; labels int8).  out4 = {count, sum y, alpha, loss};
 * scratch >= 3*CVX_DICE_BLOCKS floats.  backward: grad[i] = grad_out / count * d loss_i / d x_i, alpha treated as a constant
 * (weight.item()), 0 where the label is -1. */
int cvx_focal_loss_forward(const float* x, const int8_t* labels, long n, float gamma, float* scratch, float* out4, hipStream_t stream);
int cvx_focal_loss_backward(const float* x, const int8_t* labels, long n, float gamma, const float* stats4, float grad_out, float* grad,
                            hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * Alternate encoder: SAM2.1 Hiera image encoder + FPN neck (BASELINE configs[4]).  These entry points, together with
 * cvx_gemm_bf16 (qkv / proj / MLP / patch-embed / lateral convs) and cvx_layernorm_bf16, replace
 * `self.model.image_encoder(flat_data)` and the resize in front of it -- SAM2.forward_features,
 * src/cryovit/models/sam2.py:190-209 -- whose outputs _sam_features stores as float16
 * (src/cryovit/run/dino_features.py:67-106).  Token rows are channels-last: row = (slice * G + y) * G + x.
 * ------------------------------------------------------------------------------------------------- */

/* Resize to S x S (bilinear, align_corners = False: the in-plane part of the reference's trilinear F.interpolate,
 * sam2.py:196-203; identity when H == W == S) and gather the 7x7 / stride 4 / pad 3 patches of the patch-embedding
 * conv as a bf16 GEMM operand out[slices*(S/4)^2][ldo], column c*49 + ky*7 + kx (zero outside the image).
 * mode 0: src uint8 [D][H][W] scaled by 1/255 and replicated to 3 channels (vit_dataset.py:86-88,136-137);
 * mode 1: float [D][H][W] replicated; mode 2: float [D][3][H][W]. */
int cvx_sam_patches(const void* src, int mode, int slices, int H, int W, int S, void* out, long ldo, hipStream_t stream);

/* Multi-head attention inside windows of a token grid (Hiera's MultiScaleAttention after window_partition):
 * keys/values k, v: bf16 rows on a grid x grid token grid (leading dimension ldkv, head h at column h*head_dim),
 * windows of window x window tokens (window == grid: global attention); queries q on a q_grid x q_grid grid with
 * q_window x q_window windows -- the same grid, or the 2x2-pooled one of a stage transition.  out rows follow the query
 * grid.  softmax(q k^T / sqrt(head_dim)) v per (slice, window, head).  window^2 % 16 == 0, head_dim % 4 == 0, <= 96. */
int cvx_window_attention_bf16(const void* q, long ldq, const void* k, const void* v, long ldkv, void* out, long ldo,
                              int slices, int heads, int head_dim, int grid, int window, int q_grid, int q_window,
                              hipStream_t stream);

/* 2x2 max pool over the token grid (Hiera's do_pool): rows on grid x grid -> rows on grid/2 x grid/2, C channels;
 * fp32 (is_bf16 = 0: the residual shortcut) or bf16 (the queries). */
int cvx_pool2x2(const void* in, long ldi, void* out, long ldo, int slices, int grid, int C, int is_bf16, hipStream_t stream);

/* fp32 rows -> bf16 rows (residual stream -> GEMM operand of the FPN lateral convs). */
int cvx_cast_bf16(const float* in, long ldi, void* out, long ldo, long rows, int C, hipStream_t stream);

/* One FPN level: out_f16[slice][c][y][x] = lateral[(slice*grid + y)*grid + x][c] (+ coarse[(slice*grid/2 + y/2)*grid/2
 * + x/2][c] when coarse != NULL: the nearest-upsampled top-down term), fp32 in, float16 out. */
int cvx_fpn_level_out(const float* lateral, const float* coarse, int slices, int C, int grid, void* out_f16,
                      hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * The whole DINOv2-with-registers encoder for one slice batch as ONE call: token init + patch-embed
 * GEMM, `depth` x { LN, QK GEMM, V^T GEMM, attention, proj GEMM (LayerScale+residual), LN, FFN-in GEMM (SwiGLU gate or
 * GELU), FFN-out GEMM (LayerScale+residual) }, final LN + feature layouts.  Replaces the hub model's forward_features
 * (run/dino_features.py:58).  All pointers are device buffers owned by the caller; weights are packed as documented for
 * cvx_gemm_bf16 (Q rows pre-scaled by log2(e)/8; W12 interleaved in blocks of 8 for the SwiGLU variant).
 * ------------------------------------------------------------------------------------------------- */
typedef struct cvx_vit_layer {
    const float *ln1_w, *ln1_b;
    const void* qk_w; const float* qk_b;      /* bf16 [rup(2C,128)][C], fp32 [rup(2C,128)] */
    const void* v_w; const float* v_b;        /* bf16 [rup(C,128)][C] */
    const void* proj_w; const float *proj_b, *ls1;
    const float *ln2_w, *ln2_b;
    const void* ffn1_w; const float* ffn1_b;  /* SwiGLU: bf16 [2*hid_pad][C] interleaved; MLP: bf16 [hid_pad][C] */
    const void* ffn2_w; const float *ffn2_b, *ls2; /* bf16 [rup(C,128)][hid_pad] */
} cvx_vit_layer;

typedef struct cvx_vit_desc {
    int dim, depth, heads, n_reg, ffn_swiglu, hid_pad;
    float ln_eps;
    int qkv_merged;                /* 1 (with ln_fold): ONE qkv GEMM per block -- layers[i].qk_w / qk_b hold all 3C rows (q | k | v), v_w / v_b
                                      are not read, ws.qk is [rows][3C] and ws.vt is not used; attention through cvx_attention_qkv_bf16 */
    int ln_fold;                   /* 1 (the product path): LayerNorms folded into the qk / v / ffn1 GEMMs and the residual stream
                                      kept as a bf16 (hi, lo) pair.  The layers then carry qk_w / v_w / ffn1_w = bf16(W * ln_gamma)
                                      and qk_b / v_b / ffn1_b = fp32 [2][n_pad] (b' | column sums, see cvx_gemm_desc.ln_rowstat);
                                      ln1_* / ln2_* are not read.  0: separate cvx_layernorm_bf16 passes over an fp32 stream */
    const float* pe_b;             /* fp32 [rup(C,128)] patch-embed bias */
    const float* reg;              /* fp32 [n_reg][C] register tokens */
    const float *norm_w, *norm_b;  /* final LayerNorm */
    const cvx_vit_layer* layers;   /* HOST array of `depth` entries (device pointers inside) */
} cvx_vit_desc;

typedef struct cvx_vit_ws {        /* device workspaces, rows = rup(b*ntp,256)+256 (ntp = tokens per slice rounded up to 8) */
    void* x;    /* fp32 [rows][C]: the residual stream (ln_fold = 0); ln_fold = 1: staging of the embedded tokens only, dead
                   after the split -- it may alias `hid` when hid_pad >= 2 C */
    void* xn;   /* bf16 [rows][C]  LayerNorm output (ln_fold = 0 only; may be NULL otherwise) */
    void* qk;   /* bf16 [rows][2C]       */
    void* vt;   /* bf16 [b][heads][64][kp], zero-initialised once (kp = tokens rounded up to 64) */
    void* ao;   /* bf16 [rows][C], zero-initialised once */
    void* hid;  /* bf16 [rows][hid_pad]  */
    /* ln_fold = 1: */
    void* xh;   /* bf16 [rows][C]  hi half of the residual stream = A operand of the qk / v / ffn1 GEMMs */
    void* xl;   /* bf16 [rows][C]  lo half */
    float* stat_part; /* fp32 [C / 64][rows][2] partial row sums */
    float* rowstat;   /* fp32 [rows][2] (rstd, -mean * rstd) */
} cvx_vit_ws;

/* patches: bf16 [rup(b*hp*wp,256)+256][patches_ld] from cvx_preprocess_patches (patches_ld = 256, pe_w = channel-summed
 * kernel [rup(C,128)][256]) or cvx_im2col_patches (patches_ld = 640, pe_w [rup(C,128)][640]); pos fp32 [1+hp*wp][C]
 * (interpolated position table), cls_pos0 fp32 [C] = cls_token + pos[0].  Outputs as in cvx_final_norm_features. */
int cvx_vit_encode(const cvx_vit_desc* vit, const cvx_vit_ws* ws, int b, int hp, int wp, const void* patches, long patches_ld,
                   const void* pe_w, const float* pos, const float* cls_pos0, void* feats_f16, long d_total, long d0,
                   void* feats_cl, float* tokens_f32, hipStream_t stream);

/* ---------------------------------------------------------------------------------------------------
 * The whole CryoVIT segmentation head as ONE call: Conv3d(c_in -> c0, k=1) + GELU, n_blocks x SynthesisBlock { GroupNorm(eps
 * 1e-3), dilated Conv3d + GELU, dilated Conv3d + GELU, ConvTranspose3d (1,2,2) + GELU }, Conv3d(8,8,3) + GELU, Conv3d(8,1,3),
 * clip(+-5), sigmoid, masked Dice sums and the thresholded uint8 segmentation.  Replaces CryoVIT.forward_volume / forward
 * (src/cryovit/models/cryovit.py:36-49), the reductions of base_model.py:99-110 / metrics.py:36-41 and PredictionWriter's
 * threshold (callbacks.py:100-102).  Weights packed as documented for cvx_gemm_bf16 / cvx_conv3d_f16, in FP16.
 * ------------------------------------------------------------------------------------------------- */
typedef struct cvx_head_block {
    int c1, c2, c3, d1, d2, groups;           /* SynthesisBlock(c1, c2, c3, d1, d2); groups = max(8, c1/8) */
    const float *gn_w, *gn_b;                 /* fp32 [c1] */
    const void* conv1_w; const float* conv1_b; int conv1_npad, conv1_kpad;   /* fp16 [npad][rup(27*c1,64)] */
    const void* conv2_w; const float* conv2_b; int conv2_npad, conv2_kpad;   /* fp16 [npad][rup(27*c2,64)] */
    const void* convt_w; const float* convt_b; int convt_npad, convt_kpad;   /* fp16 [npad(4*c3)][rup(c2,64)], row (i*2+j)*c3 + o */
} cvx_head_block;

typedef struct cvx_head_desc {
    int c_in, c0, c_tail, n_blocks;           /* c_tail must be 8 */
    const void* proj_w; const float* proj_b; int proj_npad, proj_kpad;
    const cvx_head_block* blocks;             /* HOST array of n_blocks entries (device pointers inside) */
    const void* out0_w; const float* out0_b; int out0_npad, out0_kpad;        /* Conv3d(8,8,3) */
    const float* out2_w; float out2_b;        /* Conv3d(8,1,3): fp32 [27][8] tap-major, scalar bias */
    const void* zero_page;                    /* >= 256 zero bytes */
} cvx_head_desc;

#define CVX_HEAD_MAX_BLOCKS 8
typedef struct cvx_head_ws {                  /* device workspaces, fp16 channels-last, rows = rup(voxels,256)+256 (+2048 elements) */
    void* act0;                               /* [D*h*w rows][c0] */
    void* gn[CVX_HEAD_MAX_BLOCKS];            /* block i input resolution rows x c1 */
    void* t1[CVX_HEAD_MAX_BLOCKS];            /* rows x c2 */
    void* t2[CVX_HEAD_MAX_BLOCKS];            /* rows x c2 */
    void* up[CVX_HEAD_MAX_BLOCKS];            /* 4*rows x c3 */
    void* mid;                                /* full-resolution rows x 8 */
    float* gn_stats;                          /* fp32 [2*G_max*(1+CVX_GN_BLOCKS)], G_max = 128 */
    float* dice_scratch;                      /* fp32 [3*CVX_DICE_BLOCKS] (only read when labels != NULL) */
} cvx_head_ws;

/* feats_cl: FP16 channels-last features [D*h*w (+pad rows)][c_in] (cvx_vit_encode's feats_cl or
 * cvx_features_to_channels_last).  Outputs at 2^n_blocks x the in-plane resolution, all nullable: logits / probs fp32,
 * dice fp32[3] (+=, needs labels int8), mask uint8 (probs >= mask_threshold). */
int cvx_head_forward(const cvx_head_desc* head, const cvx_head_ws* ws, const void* feats_cl, int D, int h, int w, float* logits,
                     float* probs, const int8_t* labels, float* dice, uint8_t* mask, float mask_threshold, hipStream_t stream);

#ifdef __cplusplus
}
#endif
#endif /* CRYOVIT_HIP_H */
