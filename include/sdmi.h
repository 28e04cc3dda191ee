/*
 * sdmi.h -- C ABI of libsdmi.so: the MI355X (gfx950) kernels behind the SlotDiffusion hot path.
 *
 * Drop-in boundary.  The reference has no native code: every entry
 * below replaces the cuDNN/cuBLAS work that a reference Python call site dispatches through
 * torch.nn.  The host side (slotdiffusion_amd/ python package) binds these with ctypes; INTEGRATION.md
 * shows the stub a reference maintainer would add.
 *
 * Conventions
 *   - plain C: POD argument structs of raw DEVICE pointers, int dims, dtype enums.  No torch types.
 *   - caller owns all memory (inputs, outputs, workspaces); the library never allocates.
 *   - every call is asynchronous on the hipStream_t passed in (passed as void*); no implicit sync.
 *   - return 0 on success, a negative SDMI_E* code otherwise; sdmi_last_error() gives the text.
 *   - activations are NHWC ("channels last"): [B][H][W][C], C contiguous.
 *   - conv / linear weights are [Cout][KH][KW][Cin] (K-contiguous), i.e. the bytes of a
 *     torch tensor of logical shape [Cout,Cin,KH,KW] held in torch.channels_last.
 */
#ifndef SDMI_H_
#define SDMI_H_

#ifdef __cplusplus
extern "C" {
#endif

#define SDMI_VERSION 100

enum { SDMI_F32 = 0, SDMI_BF16 = 1, SDMI_FP8 = 2 /* OCP e4m3fn; sdmi_igemm operands / sdmi_quant_fp8 output only */ };
enum { SDMI_ACT_NONE = 0, SDMI_ACT_RELU = 1, SDMI_ACT_SILU = 2, SDMI_ACT_GELU = 3 };
enum { SDMI_OK = 0, SDMI_EINVAL = -1, SDMI_ELAUNCH = -2, SDMI_EUNSUPPORTED = -3 };

int sdmi_version(void);
const char* sdmi_last_error(void);

/* ------------------------------------------------------------------------------------------
 * Implicit-GEMM convolution / GEMM on the matrix cores.
 *   out[m][n] = act( alpha * sum_k A[m][k] * W[n][k] + bias[n] + rowvec[b(m)][n] + residual[m][n] )
 * A is gathered on the fly from an NHWC tensor (im2col never materialised):
 *   m -> (b, oy, ox);  k -> (kh, kw, ci);  A[m][k] = in[b][oy*stride+kh-pad_t][ox*stride+kw-pad_l][ci]
 *   (zero outside the image; with ups=1 the input is read through a virtual nearest x2 upsample).
 * A plain GEMM is the 1x1 case (H=W=1, B=M): rows of A have pitch `lda`.
 * Replaces: nn.Conv2d / nn.Linear / einsum call sites, e.g.
 *   video_based/models/unet/unet.py:222,249,255-259,408,542 (ResBlock/in/out convs),
 *   unet.py:108-121,165-179 (Up/Downsample), attention.py:44-48,175-180 (Linear),
 *   vqvae/modules.py:23-48,71-91 (VQ-VAE convs, asymmetric pad), resnet.py:12-35.
 * dtype: SDMI_BF16 -> v_mfma_f32_32x32x16_bf16; SDMI_F32 -> v_mfma_f32_32x32x2_f32 (exact fp32);
 *        SDMI_FP8 (e4m3fn operands from sdmi_quant_fp8, the BASELINE "fp8 MFMA UNet" configuration) ->
 *        v_mfma_scale_f32_32x32x64_f8f6f4 with unit block scales; the per-tensor quantisation scales
 *        of A and W are undone through `alpha` = 1 / (scale_a * scale_w).
 * Accumulation is always fp32.  Epilogue tensors bias/rowvec are fp32.
 * split_k > 1: partial sums go to `workspace` (fp32, split_k*M*N) and a second kernel reduces.
 * batch > 1 (1x1 only): grid.z batches with element strides sa/sw/sc/sr (attention GEMMs).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* a;        /* NHWC activations, `dtype` */
  const void* w;        /* [N][ldw] weights, `dtype` */
  void* out;            /* [M][ldc], `out_dtype` */
  const float* bias;    /* [N] (or [M] when bias_m) or NULL */
  const float* rowvec;  /* [B][N] per-image vector (time embedding) or NULL */
  const void* residual; /* [M][ldr] `out_dtype` or NULL */
  float* workspace;     /* split-K partials or NULL */
  int dtype, out_dtype;
  int M, N, K;          /* K = KH*KW*Cin */
  int lda, ldw, ldc, ldr;
  int B, H, W, Cin, Ho, Wo, KH, KW, stride, pad_t, pad_l, ups;
  int act;
  float alpha;
  int bias_m;
  int split_k;
  int batch;
  int ldrv;             /* row pitch of rowvec (>= N) */
  int zins;             /* >1: input is read through virtual zero-insertion upsampling by zins
                           (data-gradient of a stride-`zins` conv: flipped weights, stride=1) */
  long long sa, sw, sc, sr;
  /* Sub-sampled output placement (osy > 0): output row m = (b, oy, ox) is stored at pixel
   * (oy*osy + ooy, ox*osx + oox) of a [B][oH][oW][ldc] tensor instead of row m.  The data gradient
   * of a stride-s convolution is computed as s*s plain stride-1 convolutions over dy -- one per
   * input-pixel parity, each with the filter taps of that parity -- written interleaved this
   * way (no multiply-adds on inserted zeros, unlike `zins`).  No residual / split-K in this mode. */
  int oH, oW, osy, osx, ooy, oox;
  /* Fused LayerNorm prologue (inference; 1x1 / linear only): the GEMM runs on the RAW rows x of A
   * with weights pre-multiplied by the norm's gamma, W'[n][k] = W[n][k] * gamma[k]; the kernel takes
   * the row statistics from the operand registers and finishes
   *   out[m][n] = act( rstd[m] * (acc[m][n] - mean[m] * ln_colsum[n]) + bias[n] )
   * with ln_colsum[n] = sum_k W'[n][k] (of the ROUNDED operand) and bias[n] = sum_k W[n][k] beta[k]
   * (+ the layer's own bias), mean / rstd over the K entries of row m (eps = ln_eps).  Replaces
   * nn.LayerNorm + nn.Linear pairs: attention.py:238-240,246-251 (norm1/2/3 -> to_q/k/v, ff). */
  const float* ln_colsum; /* [N] (GEGLU: [2N]) or NULL */
  float ln_eps;
  /* geglu != 0: W has 2N rows (value rows 0..N-1, gate rows N..2N-1), bias / ln_colsum 2N entries;
   * out[m][n] = value[m][n] * gelu(gate[m][n]), N columns (attention.py:44-48 GEGLU.forward). */
  int geglu;
  /* optional extra A sources appended along K (1x1 / linear, or a stride-1 "same" convolution): the
   * first K1 = KH*KW*Cin columns of A come from `a` as usual, columns [K1, K2) are a2[m][k - K1] (row
   * pitch lda2) and columns [K2, K) a3[m][k - K2] (row pitch lda3; K2 = 0: no third source) -- i.e. 1x1
   * taps at the output pixel.  K1, K2 - K1, K - K2 are multiples of the K tile (64 elements).  Uses:
   * the skip concat's 1x1 convolution reads its two inputs in place; a ResBlock's second 3x3
   * convolution and its 1x1 skip convolution run as ONE GEMM ([im2col(h) | x] or [im2col(h) | xa | xb],
   * weights concatenated along K; unet.py:255-259,268-269). */
  const void* a2;
  int lda2, K1;
  const void* a3;
  int lda3, K2;
  /* softmax8 = S in 1..16 (with ln_colsum): the epilogue finishes with a softmax over every aligned group
   * of 8 (S <= 8) or 16 (S > 8: the video configurations' 11 / 15 slots) output columns of which the first S are
   * scores (the others are pads: treated as -inf, written as 0) -- the S slot scores of one attention head.  With batch > 1, ln_colsum / bias advance by s_colsum / s_bias entries per batch:
   * slot cross-attention as two per-image GEMMs (attention.py:182-206 with the 7 keys folded into the
   * query weights, the values into the output projection; see engine.UNetRunner.cross_fold). */
  int softmax8;
  long long s_colsum, s_bias;
  /* geglu only: if non-NULL, the pre-activation h [M][2N] (value columns, then gate columns; bias
   * included; out_dtype, row pitch ldc2) is stored as well and out is computed from the ROUNDED h, i.e.
   * exactly what nn.Linear followed by GEGLU's chunk / gelu / multiply gives -- the training forward
   * keeps h for the backward (attention.py:44-48) without a separate GEGLU launch. */
  void* out2;
  int ldc2;
  /* defer_epilogue != 0: when the launcher splits K (see sdmi_igemm_split_plan), the partials are LEFT in
   * `workspace` ([splits][M][N] fp32) and nothing is written to `out`: the reduction + bias / rowvec /
   * residual terms are finished by the next kernel's prologue -- sdmi_groupnorm's `part` source (the
   * GroupNorm that reads this convolution, unet.py:243-250) -- or by sdmi_splitk_finish.  Requires
   * act = none, ldc = N, no fused prologue / epilogue forms.  Ignored when K is not split. */
  int defer_epilogue;
  /* gn_part != NULL (bf16 out; M, N multiples of 128; Ho*Wo a power of two >= 32; N / gn_groups a power of two
   * <= 32; no split-K): the epilogue ALSO writes the GroupNorm statistics of its output -- per image b, 32-row
   * block k and group g the sum and the sum of squares of the ROUNDED outputs,
   *   gn_part[((b * (Ho*Wo/32) + k) * gn_groups + g) * 2 + {0, 1}],
   * each entry by exactly one wave (no atomics).  This is the `partial` operand of sdmi_groupnorm_apply with
   * nsplit = Ho*Wo/32: the GroupNorm behind the convolution (unet.py:243-250) needs no statistics pass. */
  float* gn_part;
  int gn_groups;
  /* alpha_dev != NULL (fp8 operands only): the effective alpha is alpha * (*alpha_dev), read on the device when the
   * kernel starts -- the 1 / scale of an operand that sdmi_fp8_quant_group quantised with a scale it derived from the
   * tensor's own amax in the same stream, so that a captured train step re-derives it at every replay (BASELINE
   * configs[4] "fp8 MFMA UNet": unet.py:271-285 with e4m3fn operands). */
  const float* alpha_dev;
  /* parity4 != 0 (batch = 4, KH = KW = 2, stride 1, osy = osx = 2, sa = sc = 0): the four parity convolutions of a
   * nearest-2x upsample + 3x3 convolution (unet.py:108-121 as 2x2 convolutions of the low-resolution input, one per
   * output parity) in ONE launch -- batch index z = 2 py + px selects the filter (w + z * sw), the padding
   * (pad_t = 1 - py, pad_l = 1 - px) and the placement (ooy = py, oox = px); the caller's pad_t / pad_l / ooy / oox are
   * ignored.  Bias-only epilogue like every sub-sampled output. */
  int parity4;
} SdmiGemmArgs;
int sdmi_igemm(const SdmiGemmArgs* a, void* stream);
/* Host-side query, no launch (stream ignored): the number of K slices sdmi_igemm would use for these
 * arguments (1 = no split-K).  A caller that wants the next kernel to finish the reduction sets
 * defer_epilogue when this returns > 1. */
int sdmi_igemm_split_plan(const SdmiGemmArgs* a, void* stream);
/* Stand-alone second stage of a deferred split-K launch: a->split_k = the slice count the launch used
 * (sdmi_igemm_split_plan), every other field as passed to sdmi_igemm.  Sums the partials in slice order
 * and applies alpha / bias / rowvec / residual / act -> out. */
int sdmi_splitk_finish(const SdmiGemmArgs* a, void* stream);

/* wgrad: dW[n][kh][kw][ci] = sum_m dY[m][n] * A[m][(kh,kw,ci)]  (fp32 output, [N][K] like W).
 * Replaces the weight-gradient half of Conv2d/Linear backward (torch autograd in the reference). */
typedef struct {
  const void* a;        /* NHWC activations (forward input), `dtype` */
  const void* dy;       /* [M][ldy] output gradient, `dtype` */
  float* dw;            /* [N][K] fp32 (accumulated into when accumulate != 0) */
  float* dbias;         /* [N] fp32 column sums of dy, or NULL */
  float* workspace;     /* split partials: splits*(N*K + N) floats (unused when splits == 1:
                           one launch, written straight into dw / dbias) */
  int dtype;
  int M, N, K, lda, ldy;
  int B, H, W, Cin, Ho, Wo, KH, KW, stride, pad_t, pad_l, ups;
  int splits;
  int accumulate;
  int defer_fold;   /* != 0 with splits > 1: leave the split partials in `workspace`; the caller folds
                       them later with sdmi_wgrad_fold_group (one launch for up to 16 layers) */
} SdmiWgradArgs;
int sdmi_wgrad(const SdmiWgradArgs* a, void* stream);
/* Fold the split partials of up to 16 earlier sdmi_wgrad(defer_fold = 1) launches into their dw / dbias
 * (+= when accumulate) in ONE launch.  `problems` (see SdmiWgradGroupArgs below) is a HOST array of the
 * same SdmiWgradArgs (dw, dbias, workspace, N, K, splits, accumulate are read); destinations must be
 * distinct within a call.  Results equal the immediate folds bit for bit (split order). */

/* Grouped weight gradients: up to 16 independent 1x1 / linear bf16 problems (the eight Linear layers
 * of a transformer block, attention.py:182-251) in ONE launch.  Each problem alone has too few output
 * tiles to fill 256 CUs and would pay an M-split + fold launch pair; together their tiles fill the
 * chip.  `problems` is a HOST array of n SdmiWgradArgs (each with its own splits / workspace as for
 * sdmi_wgrad; bf16, KH = KW = 1, N > 64, K > 64); problems with splits > 1 are folded by one shared
 * second launch.  Results equal n sdmi_wgrad calls bit for bit.
 * A group may instead hold fp32 problems (all of them; any N, K; 64 x 64 tiles of the exact-fp32 kernel): the Slot
 * Attention / predictor layers (slot_attention.py:57-106: M = images x slots rows) are 9 - 36 workgroups and ~20 us of
 * latency each when launched alone. */
typedef struct { const void* problems; int n; } SdmiWgradGroupArgs;
int sdmi_wgrad_group(const SdmiWgradGroupArgs* a, void* stream);
int sdmi_wgrad_fold_group(const SdmiWgradGroupArgs* a, void* stream);

/* Data gradient AND weight gradient of one conv / linear layer in ONE launch (bf16) -- the backward of
 * unet.py:271-285 (ResBlock convolutions), attention.py:182-206, 247-251 (Linear layers),
 * slot_attention.py:57-106, as torch autograd runs it behind `loss.backward()` (ldm.py:59-83).
 *   dgrad  HOST pointer to the SdmiGemmArgs of the data gradient exactly as sdmi_igemm would take it
 *          (a = dY, w = the flipped operand, out = dX; alpha / bias / residual epilogue only; no split-K,
 *          batch, sub-sampled output or fused epilogues; N > 64);
 *   wgrad  HOST pointer to the SdmiWgradArgs of the weight gradient as sdmi_wgrad would take it (N > 64,
 *          K > 64; 1x1 / linear, or a stride-1 same-size convolution; splits > 1
 *          leaves the M-split partials in `workspace` -- see `fold`);
 *   fold   optional HOST pointer to the SdmiWgradArgs of an EARLIER launch whose partials are complete
 *          (dw, dbias, workspace, N, K, splits, accumulate are read): folded into its dw / dbias by extra
 *          workgroups of this launch, in split order (bit-identical to sdmi_wgrad's own fold).  Must not
 *          target this launch's dw / dbias.  The last layer's partials are folded with
 *          sdmi_wgrad_fold_group.
 *   dgrad_cap  workgroups that walk the data-gradient tiles (0: what is left of two per CU).
 *   wgrad_tile 0 / 128: 128 x 128 output tiles of dW (sdmi_wgrad's choice); 64: 64 x 64 tiles for small
 *              layers (only next to 64 x 64 data-gradient tiles) -- more, shorter workgroups.
 * Results equal sdmi_igemm + sdmi_wgrad bit for bit with 128 x 128 tiles (same tile bodies, same split
 * order); with 64 x 64 tiles every dW element is still one fp32 accumulation chain over its rows in m
 * order, M-split partials folded in split order. */
typedef struct {
  const void* dgrad; const void* wgrad; const void* fold;
  int dgrad_cap, wgrad_tile;
} SdmiBwdPairArgs;
int sdmi_bwd_pair(const SdmiBwdPairArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * GroupNorm (+ fused activation) on NHWC, statistics in fp32.
 * Replaces GroupNorm32/Normalize + SiLU/ReLU/swish: unet/utils.py:120-139, unet.py:219-222,
 * 243-250, attention.py:77-79, vqvae/modules.py:12-14, resnet.py:8-9.
 * stats: [B][groups][2] fp32 (mean, rstd) written by sdmi_groupnorm_stats.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* x;       /* [B][HW][C] */
  void* y;             /* [B][HW][C] (may alias x) */
  const float* gamma;  /* [C] */
  const float* beta;   /* [C] */
  float* stats;        /* [B][groups][2] */
  float* partial;      /* workspace [B][nsplit][groups][2] */
  int dtype;
  int B, HW, C, groups;
  float eps;
  int act;
  int nsplit;
  const void* residual; /* optional [B][HW][C] added AFTER the affine, before act (ResNet block) */
  /* optional inverted dropout applied to the activated output (ResBlock: Dropout(SiLU(GN(h))),
   * unet.py:243-250): y = keep ? y / (1 - p) : 0 with the counter-based keep mask of
   * sdmi_dropout's family, keyed on (drop_seed + *drop_seed_dev, 16-byte vector index). */
  float drop_p;
  long long drop_seed;
  const long long* drop_seed_dev;
  /* optional (bf16 input): the activated output is ALSO quantised to e4m3fn bytes at y8 (same
   * [B][HW][C] indexing, y8_scale as in sdmi_quant_fp8) -- the operand of the fp8 convolution
   * behind the norm; y may then be NULL (inference: the bf16 copy has no other reader). */
  void* y8;
  float y8_scale;
  /* optional second source (inference): the input is the channel concatenation [x | x2] of two
   * tensors -- x [B][HW][C1], x2 [B][HW][C - C1] -- read in place (the UNet's skip concat,
   * unet.py:571-573, without materialising it).  C1 is a multiple of the 16-byte vector. */
  const void* x2;
  int C1;
  /* optional (bf16): the input is NOT x but the unfinished result of a split-K sdmi_igemm launched with
   * defer_epilogue -- part [part_splits][B*HW][C] fp32 partial sums; the kernel forms
   *   in = bf16( sum_k part[k] * part_alpha + part_bias[c] + part_rowvec[b][c] + part_residual[b][hw][c] )
   * (same order of operations and the same rounding as the GEMM's own split-K epilogue, so results are
   * bit-identical to the two launches it replaces), stores it at raw_out (the tensor's other readers: skip
   * branch, UNet concat) and normalises it.  x is not read.  With x2 the partials stand for the FIRST of the two
   * concatenated tensors ([B*HW][C1]: C1 replaces C in the layouts above). */
  const float* part;
  int part_splits;
  float part_alpha;
  const float* part_bias;     /* [C] or NULL */
  const float* part_rowvec;   /* [B][part_ldrv] or NULL */
  int part_ldrv;
  const void* part_residual;  /* [B][HW][C] bf16 or NULL */
  void* raw_out;              /* [B][HW][C] bf16 */
} SdmiGroupNormArgs;
int sdmi_groupnorm_stats(const SdmiGroupNormArgs* a, void* stream);
int sdmi_groupnorm_apply(const SdmiGroupNormArgs* a, void* stream);
/* Both passes in one call; images small enough to sit in one workgroup's registers
 * (HW * C * elem <= 64 KiB, nsplit == 1) take a single fused launch (x read once). */
int sdmi_groupnorm(const SdmiGroupNormArgs* a, void* stream);

typedef struct {
  const void* x;       /* forward input */
  const void* dy;      /* grad of the op output (after act) */
  void* dx;
  const float* gamma;
  const float* beta;
  const float* stats;  /* from forward */
  float* dgamma;       /* [C] fp32 (written) */
  float* dbeta;        /* [C] */
  float* partial;      /* workspace [B][nsplit][groups][2] + [B][nsplit'][C][2] */
  int dtype;
  int B, HW, C, groups;
  int act;
  int nsplit;
  const void* residual; /* forward residual (for act derivative) or NULL */
  void* dresidual;      /* optional: receives the gradient flowing to the residual branch */
  int accumulate;       /* != 0: dgamma/dbeta += (parameter used more than once per step) */
  const void* dextra0;  /* optional [B][HW][C]: gradients reaching x through OTHER consumers (the */
  const void* dextra1;  /* residual / skip / concat branches); dx = gn_bwd(dy) + dextra0 + dextra1 */
  float drop_p;         /* the forward's fused dropout (mask regenerated from the seed) */
  long long drop_seed;
  const long long* drop_seed_dev;
  float* dxsum;         /* optional [B][ld_dxsum]: dxsum[b][c] = sum over the image's pixels of dx[b][.][c]
                           (single-pass kernel only; sdmi_groupnorm_bwd_fused tells) -- the gradient of a
                           per-image row vector that was added to x (ResBlock time embedding, unet.py:268) */
  int ld_dxsum;
  int defer_colsum;     /* != 0: leave the per-image channel sums in `partial` ([B*nsplit][C][2]); the
                           caller folds them into dbeta / dgamma later with sdmi_colsum_group */
} SdmiGroupNormBwdArgs;
int sdmi_groupnorm_bwd(const SdmiGroupNormBwdArgs* a, void* stream);
/* number of [C][2] entries sdmi_groupnorm_bwd leaves in `partial` for this geometry (B for the
 * single-pass kernels, B * nsplit for the two-pass ones); no launch.  Returns the count (> 0). */
/* query: 1 if sdmi_groupnorm_bwd runs this geometry as the single-pass kernel (and fills dxsum) */
int sdmi_groupnorm_bwd_fused(const SdmiGroupNormBwdArgs* a, void* unused);
int sdmi_groupnorm_bwd_entries(const SdmiGroupNormBwdArgs* a, void* stream);

/* LayerNorm over the last dim (eps 1e-5 default): attention.py:238-240, savi.py:38-54. */
typedef struct {
  const void* x; void* y; const float* gamma; const float* beta;
  float* stats;        /* optional [rows][2] (mean, rstd) saved for backward, or NULL */
  int dtype; int rows, C, ldx, ldy; float eps;
} SdmiLayerNormArgs;
int sdmi_layernorm(const SdmiLayerNormArgs* a, void* stream);

typedef struct {
  const void* x; const void* dy; void* dx; const float* gamma; const float* stats;
  float* dgamma; float* dbeta;   /* [C] fp32, accumulated via per-block partials */
  float* partial;                /* workspace [nblk][C][2] */
  int dtype; int rows, C; int nblk; int accumulate;
  const void* dextra;            /* optional [rows][C]: dx = ln_bwd(dy) + dextra (residual branch of x) */
  int defer_colsum;              /* != 0: partials stay in `partial` ([nblk][C][2]) for sdmi_colsum_group */
} SdmiLayerNormBwdArgs;
int sdmi_layernorm_bwd(const SdmiLayerNormBwdArgs* a, void* stream);

/* Deferred folds of the normalisation backward passes: up to 64 independent column sums
 *   out0[c] += sum_e partial[e][c][0],  out1[c] += sum_e partial[e][c][1]   (e < nblk)
 * in ONE launch (a train step has ~190 of them, each a few microseconds of work behind a launch).
 * `items` is a HOST array of n SdmiColsumItem. */
typedef struct { const float* partial; float* out0; float* out1; int nblk, C; } SdmiColsumItem;
typedef struct { const void* items; int n; } SdmiColsumGroupArgs;
int sdmi_colsum_group(const SdmiColsumGroupArgs* a, void* stream);


/* ------------------------------------------------------------------------------------------
 * Multi-head attention, head_dim 32 (UNet self- and slot cross-attention).
 *   out[b][i][h*32+d] = sum_j softmax_j(scale * q_i . k_j) v_j[d]
 * Replaces CrossAttention.forward, unet/attention.py:182-206 (and the frozen DINO ViT's
 * ViTSelfAttention at head_dim 64, dino.py:30-41 through transformers).
 * q/k/v are [B][S][ld*] with head h at channel offset h*head_dim.  lse (optional): [B][heads][Sq]
 * log-sum-exp saved for the backward kernel.
 * Sequence lengths: bf16 with head_dim 32 / 64 -- any Skv (keys pass through LDS in chunks; forward,
 * and for head_dim 32 the backward too); fp32 and head_dim 48 -- Skv <= 400 (K/V whole in LDS), longer
 * sequences are composed on the host from sdmi_igemm + sdmi_softmax_rows (ops.attention_long).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* q; const void* k; const void* v; void* out; float* lse;
  int dtype; int B, heads, Sq, Skv; int ldq, ldk, ldv, ldo; float scale;
  int head_dim;        /* 0 or 32: UNet; 48: SAVi transformer predictor (predictor.py:20-44); 64: DINO ViT (bf16, forward) */
} SdmiAttnArgs;
int sdmi_attention(const SdmiAttnArgs* a, void* stream);

typedef struct {
  const void* q; const void* k; const void* v; const void* out; const void* dout; const float* lse;
  void* dq; void* dk; void* dv;
  int dtype; int B, heads, Sq, Skv; int ldq, ldk, ldv, ldo; float scale;
  int head_dim;
} SdmiAttnBwdArgs;
int sdmi_attention_bwd(const SdmiAttnBwdArgs* a, void* stream);

/* row softmax in place over [rows][cols] fp32/bf16 with scale (VQ-VAE AttnBlock,
 * vqvae/modules.py:141-143). */
typedef struct { void* x; int dtype; int rows, cols, ld; float scale; } SdmiSoftmaxArgs;
int sdmi_softmax_rows(const SdmiSoftmaxArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Fused Slot Attention: ALL iterations in one launch, one workgroup per image.
 * Replaces SlotAttentionWMask.forward's loop, img_based/models/sa_diffusion.py:40-68
 * (video twin savi_diffusion.py:40-68): q=Linear(LN(slots)); softmax over slots; +eps, renorm
 * over tokens; updates=attn^T v; GRUCell; residual MLP.  k,v come from the K/V projection GEMM.
 * seg [B][M][N] = last-iteration softmax (before +eps); slots_out [B][N][D] fp32.
 * trace (optional, training): per-iteration saved tensors for the backward kernel.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* k; const void* v;      /* [B][M][ldkv], `dtype` */
  const float* slots_in;             /* [B][N][D] or [N][D] when slots_bstride == 0 */
  float* slots_out;                  /* [B][N][D] */
  float* seg;                        /* [B][M][N] */
  const float *lnq_g, *lnq_b, *wq;   /* LN(D), [D][D] */
  const float *w_ih, *w_hh, *b_ih, *b_hh; /* GRUCell [3D][D], [3D] */
  const float *lnm_g, *lnm_b, *w1, *b1, *w2, *b2; /* MLP LN, [H][D], [H], [D][H], [D] */
  float* trace;                      /* NULL or [B][iters][SDMI_SA_TRACE_FLOATS(N,D,H)] */
  int dtype; int B, M, N, D, Hid, iters, ldkv; long long slots_bstride; float eps, scale;
} SdmiSlotAttnArgs;
int sdmi_slot_attention(const SdmiSlotAttnArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * VQ nearest code (expanded-square distance, first minimum): vqvae/quantize.py:85-94.
 * z [R][ldz] fp32 (first `dim` channels used) -> idx int64 [R], zq [R][ldz] (straight-through
 * form z + (e - z), pad channels zeroed).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const float* z; const float* codebook; long long* idx; float* zq;
  int R, dim, ldz, n_codes; float scale;
  /* optional: the latent is formed on load as (zc0 * z + zc1 * z2) / zdiv, the operations and their order those of
   * sdmi_lincomb (the sampler's x0 = (x_t - sigma_t eps) / alpha_t, dpm_solver.py:345-361, without a launch of its
   * own; only the first `dim` channels of z2 are read, so the pad channel of eps need not be zeroed). */
  const float* z2; float zc0, zc1, zdiv;
} SdmiVqArgs;
int sdmi_vq_nearest(const SdmiVqArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Small fused elementwise kernels.
 * ------------------------------------------------------------------------------------------ */
/* y = ((c0*x0 + c1*x1) + c2*(x2 - x3)) / div   (fp32; NULL operands skipped, x3 NULL -> c2*x2,
 * div == 0 -> no division).  Evaluated in exactly this order without FMA contraction so the
 * DPM-Solver++ updates round like the reference's op-by-op tensor arithmetic
 * (ddpm/dpm_solver.py:523-534, 665-668, 722-732, 815-831) and q-sample (ddpm/ddpm.py:161-165). */
typedef struct {
  float* y; const float* x0; const float* x1; const float* x2; const float* x3;
  float c0, c1, c2, div; long long n;
} SdmiLincombArgs;
int sdmi_lincomb(const SdmiLincombArgs* a, void* stream);
/* per-row coefficients: y[b][i] = ca[b]*x0[b][i] + cb[b]*x1[b][i] (q-sample with per-image t) */
typedef struct {
  float* y; const float* x0; const float* x1; const float* ca; const float* cb; int B; long long per;
} SdmiRowLincombArgs;
int sdmi_row_lincomb(const SdmiRowLincombArgs* a, void* stream);

/* layout / dtype conversion: src [B][C][H][W] fp32 (NCHW) <-> dst [B][H][W][Cpad] `dtype` */
typedef struct { const float* src; void* dst; int dtype; int B, C, H, W, Cpad; } SdmiNchwToNhwcArgs;
int sdmi_nchw_to_nhwc(const SdmiNchwToNhwcArgs* a, void* stream);
typedef struct { const void* src; float* dst; int dtype; int B, C, H, W, Cpad; } SdmiNhwcToNchwArgs;
int sdmi_nhwc_to_nchw(const SdmiNhwcToNchwArgs* a, void* stream);
/* generic cast/copy of a strided 2-D view: dst[r][c] = (dst_dtype) src[r][c] */
typedef struct {
  const void* src; void* dst; int src_dtype, dst_dtype; long long rows; int cols, lds, ldd;
  int zpad;      /* != 0: columns [cols, ldd) of dst are written with zeros (channel padding) */
} SdmiCast2dArgs;
int sdmi_cast2d(const SdmiCast2dArgs* a, void* stream);
/* Per-tensor fp8 quantisation of a strided 2-D view: dst[r][c] = e4m3fn(clamp(src[r][c] * scale, +-448)),
 * columns [cols, ldd) zero (K padding).  src fp32 / bf16; dst bytes, row pitch ldd (multiple of 16).
 * Feeds the SDMI_FP8 operands of sdmi_igemm (activations with a fixed scale, weights with
 * scale = 448 / amax at weight-preparation time). */
typedef struct {
  const void* src; void* dst; int src_dtype; long long rows; int cols, lds, ldd; float scale;
} SdmiQuantFp8Args;
int sdmi_quant_fp8(const SdmiQuantFp8Args* a, void* stream);
/* Per-tensor e4m3fn quantisation of a whole TABLE of tensors with scales derived on the device (no read-back):
 * for every descriptor d, amax_d = max |src_d|, scale_d = 448 / max(amax_d, 1e-12), dst_d = e4m3fn(src_d * scale_d),
 * inv_scale[d] = max(amax_d, 1e-12) / 448 (= SdmiGemmArgs.alpha_dev of the GEMMs that read dst_d).  One memset +
 * two launches for the table: the weights of the fp8 configuration are re-quantised after every optimiser step inside
 * the captured train step.  `descs` is a device array; workgroup b serves the descriptor with
 * block_begin <= b < next block_begin, a tensor takes ceil(n / 4096) workgroups; n is a multiple of 16. */
typedef struct { const void* src; void* dst; long long n; int block_begin; int pad_; } SdmiFp8Desc;
typedef struct {
  const void* descs; int n_desc; int src_dtype; int total_blocks;
  unsigned* amax_bits;   /* [n_desc] workspace (zeroed by the call): max |x| as fp32 bits */
  float* inv_scale;      /* [n_desc] out */
} SdmiFp8GroupArgs;
int sdmi_fp8_quant_group(const SdmiFp8GroupArgs* a, void* stream);
/* ------------------------------------------------------------------------------------------
 * Fused SpatialTransformer block (bf16 inference): GroupNorm -> proj_in -> [LayerNorm -> self-attention ->
 * + ; LayerNorm -> slot cross-attention -> + ; LayerNorm -> GEGLU feed-forward -> +] -> proj_out -> + x in TWO
 * launches (phase A: GroupNorm, proj_in, q | k | v; phase B: everything behind them), activations of a
 * workgroup's 64 token rows resident in LDS / registers.
 * Replaces SpatialTransformer.forward / BasicTransformerBlock._forward / CrossAttention.forward / FeedForward:
 * video_based/models/unet/attention.py:297-308, 247-251, 182-206, 44-65 (10-13 launches of sdmi_groupnorm,
 * sdmi_igemm, sdmi_attention per block otherwise).
 *   x, tok, out [B][S][C] bf16; qkv [B][S][3C] bf16 (tok / qkv: workspaces written by phase A, read by B).
 *   C = 256 or 384 (heads = C / 32, head dim 32); S = tokens per image, a multiple of `rows`, <= 256.
 *   wstream_a / wstream_b: weights pre-packed into per-wave unit streams (2 KB units = the LDS image of 16
 *     weight rows x 64 k, in the order the kernel consumes them; python: kern.WeightBank.st_pack).
 *   wstream_img [B][...]: the per-image operands of the folded slot cross-attention (kern.Kern.cross_prepare)
 *     in the same unit format; vec_img [B][256] fp32 = LayerNorm-fold column sums | biases of the score GEMM.
 *   vec_a fp32 [7C]: proj_in bias | LayerNorm-fold column sums of q, k, v | folded biases of q, k, v.
 *   vec_b fp32 [19C]: to_out bias | cross to_out bias | fold column sums of the GEGLU projection (8C) | its
 *     folded biases (8C) | bias of the merged (ff.net.2 ; proj_out) output.
 *   phase: 0 = both launches, 1 = A only, 2 = B only.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* x; void* tok; void* qkv; void* out;
  const float* gn_gamma; const float* gn_beta;
  const void* wstream_a; const float* vec_a;
  const void* wstream_b; const float* vec_b;
  const void* wstream_img; const float* vec_img;
  int B, S, C, slots, phase;        /* slots <= 8: 8-column score groups; 9..16 (C = 256 only: heads * 16 = 128 columns): 16 */
  float gn_eps, ln_eps, attn_scale;
  int rows;   /* token rows per workgroup: 0 / 64, or 32 (twice the workgroups: small grids, e.g. 64 images at 8^2) */
  /* ff_split = 2 (phase B): every 32 / 64 token rows are served by a PAIR of workgroups, each streaming half of the
   * feed-forward's hidden chunks (half the weight bytes per workgroup, twice the workgroups: grids that leave half the
   * chip idle).  The pair's fp32 partial outputs go to part [2][B*S][C] instead of `out`; out = part[0] + part[1] +
   * vec_b's output bias + x is finished by the next kernel exactly like a split-K convolution's second stage
   * (sdmi_groupnorm's part source, or sdmi_splitk_finish). */
  float* part; int ff_split;
} SdmiStBlockArgs;
int sdmi_st_block(const SdmiStBlockArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Training twin of the fused SpatialTransformer block (bf16): the forward pass of
 * SpatialTransformer.forward / BasicTransformerBlock._forward / CrossAttention.forward / FeedForward
 * (video_based/models/unet/attention.py:297-308, 247-251, 182-206, 44-65) in TWO launches that also store
 * what the backward pass reads -- instead of ~14 launches of sdmi_groupnorm / sdmi_igemm / sdmi_layernorm /
 * sdmi_attention per block.  Differences to sdmi_st_block (inference): plain weights (no LayerNorm fold, no
 * ff.net.2 / proj_out merge: the weights change every step), LayerNorm computed explicitly on the fp32
 * residual stream, slot cross-attention explicit (q projection, softmax over the slots per head, output
 * projection: gradients reach the slot keys / values).
 *   x [B][S][C] block input; out [B][S][C] block output.
 *   saved for backward (all bf16 [B][S][.] unless noted):
 *     hgn  = GroupNorm(x)                          gn_stats [B][32][2] fp32 (mean, rstd)
 *     tok  = proj_in(hgn)            n1 = LN1(tok)  st1 [B*S][2] fp32 (mean, rstd)
 *     qkv  [B][S][3C] = n1 Wqkv^T    a1 = self-attention output, lse1 [B][heads][S] fp32
 *     x1   = a1 Wo^T + bo + tok      n2 = LN2(x1)   st2
 *     q2   = n2 Wq2^T                a2 = slot cross-attention output, lse2 [B][heads][S] fp32
 *     x2   = a2 Wo2^T + bo2 + x1     n3 = LN3(x2)   st3
 *     h    [B][S][8C] = n3 W1^T + b1 (value | gate) g [B][S][4C] = value * gelu(gate)
 *     x3   = g Wff2^T + bff2 + x2    out = x3 Wpo^T + bpo + x
 *   kv2 [B][slots][ldkv] bf16: slot keys in columns [0, C), values in [C, 2C) (attn2.to_k / to_v of the context).
 *   wstream_a / wstream_b: the block's bf16 weights as per-wave unit streams (sdmi_st_pack, re-packed after
 *     every optimiser step): A = [proj_in | to_q | to_k | to_v], B = [attn1.to_out | attn2.to_q | attn2.to_out |
 *     per hidden chunk of 128: ff.net.0.proj value rows, gate rows, ff.net.2 k-chunk | proj_out].
 *   fp32 vectors straight from the parameter arena: gn_gamma / gn_beta, b_in, ln{1,2,3}_g / _b, b_o, b_o2,
 *     b_ff1 [8C], b_ff2, b_po.
 *   C = 256 or 384; S a multiple of `rows` (64 or 32), <= 256; slots <= 16; phase 0 = both, 1 = A, 2 = B.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* x; void* out;
  void* hgn; float* gn_stats;
  void* tok; void* n1; float* st1; void* qkv;
  void* a1; float* lse1;
  void* x1; void* n2; float* st2;
  void* q2; void* a2; float* lse2;
  void* x2; void* n3; float* st3;
  void* h; void* g; void* x3;
  const void* kv2; int ldkv;
  const void* wstream_a; const void* wstream_b;
  const float* gn_gamma; const float* gn_beta; const float* b_in;
  const float* ln1_g; const float* ln1_b; const float* b_o;
  const float* ln2_g; const float* ln2_b; const float* b_o2;
  const float* ln3_g; const float* ln3_b; const float* b_ff1; const float* b_ff2; const float* b_po;
  int B, S, C, slots, phase, rows;
  float gn_eps, ln_eps, attn_scale;
} SdmiStTrainArgs;
int sdmi_st_train_fwd(const SdmiStTrainArgs* a, void* stream);

/* Backward data path of the block above in three launches (the two attention backward passes run between them as
 * sdmi_attention_bwd; weight / bias gradients are sdmi_wgrad launches on the tensors stored here):
 *   phase 1 (B1): dout -> proj_out' -> dx3 ; ff.net.2' -> GEGLU' (h) -> dh ; ff.net.0.proj' -> LN3' (x2, st3) + dx3 -> dx2 ;
 *                 attn2.to_out' -> da2
 *   phase 2 (B2): dq2 -> attn2.to_q' -> LN2' (x1, st2) + dx2 -> dx1 ; attn1.to_out' -> da1
 *   phase 3 (A) : dqkv -> to_q|k|v' -> LN1' (tok, st1) + dx1 -> dtok ; proj_in' -> dhgn   (GroupNorm' = sdmi_groupnorm_bwd)
 * All row tensors bf16 [B][S][C] (dh [B][S][8C], dqkv [B][S][3C]); ln{1,2,3}_part: [B*S/rows][C] float2 partial column
 * sums (sum_rows dn * xhat, sum_rows dn) of this launch's workgroups for sdmi_colsum_group (dgamma, dbeta).
 * wstream_b1 / _b2 / _a: the TRANSPOSED weights as unit streams (sdmi_st_pack with rs = 1):
 *   b1 = [proj_out^T | per hidden chunk: ff.net.2^T rows, ff.net.0.proj^T value k-chunk, gate k-chunk | attn2.to_out^T],
 *   b2 = [attn2.to_q^T | attn1.to_out^T],  a = [to_q^T | to_k^T | to_v^T | proj_in^T]. */
typedef struct {
  const void* dout; const void* h; const void* x2; const float* st3; const float* ln3_g;
  void* dx3; void* dh; void* dx2; void* da2; float* ln3_part;
  const void* dq2; const void* x1; const float* st2; const float* ln2_g;
  void* dx1; void* da1; float* ln2_part;
  const void* dqkv; const void* tok; const float* st1; const float* ln1_g;
  void* dtok; void* dhgn; float* ln1_part;
  const void* wstream_b1; const void* wstream_b2; const void* wstream_a;
  int B, S, C, phase, rows;
} SdmiStTrainBwdArgs;
int sdmi_st_train_bwd(const SdmiStTrainBwdArgs* a, void* stream);

/* Weight units of the fused SpatialTransformer kernels from the (bf16) parameter arena: unit u = the XOR-swizzled
 * LDS image of 16 rows x 64 k of a matrix (physical 16-byte chunk p of row r holds logical chunk p ^ ((r >> 1) & 7)),
 * element (r, k) read at src + 2 * (r * rs + k * cs) bytes -- cs = 1: a row-major weight; rs = 1: its transpose (the
 * data-gradient streams).  `descs`: DEVICE array of n_units descriptors; one launch re-packs every stream of a model
 * after an optimiser step (inside the captured train step). */
typedef struct { const void* src; void* dst; int rs; int cs; } SdmiStPackDesc;
typedef struct { const void* descs; int n_units; } SdmiStPackArgs;
int sdmi_st_pack(const SdmiStPackArgs* a, void* stream);

/* Folded slot cross-attention of one transformer block in ONE launch (bf16 inference; attention.py:182-206,
 * 247-251: norm2 -> CrossAttention(slots) -> to_out + residual), per image b with the operands of
 * kern.Kern.cross_prepare (the slot keys folded into the query projection, the values into the output projection):
 *   P   = softmax over each head's `slots` scores of  LayerNorm(tok[b]) wq[b]^T + biasq[b]     [HW][R]
 *   out = P w2[b]^T + bias + tok[b]                                                            [HW][C]
 * tok / out [B][HW][C] bf16 (out may not alias tok); wq [B][R][ld_wq], w2 [B][C][ld_w2] bf16 with image strides
 * s_wq / s_w2 (elements); colsum [B][s_colsum], biasq [B][s_bias] fp32 -- the LayerNorm-fold terms of sdmi_igemm's
 * ln_colsum form (wq carries the norm's gamma); R = heads * 8 score columns, column 8 h + j = slot j of head h,
 * j >= slots are pads (probability 0); bias [C] fp32 or NULL.  One workgroup per 16 tokens of an image: replaces
 * the two batched sdmi_igemm launches (softmax8 epilogue + output projection) where an image has few tokens. */
typedef struct {
  const void* tok; void* out;
  const void* wq; const void* w2;
  const float* colsum; const float* biasq; const float* bias;
  int B, HW, C, R, slots;
  int ld_wq, ld_w2;
  long long s_wq, s_w2, s_colsum, s_bias;
  float ln_eps;
  /* packed != 0: wq / w2 are stored per image in MFMA-fragment order -- [tile of 16 rows][k step of 32][lane][8]
   * with lane = (k sub-group of 8) * 16 + row in tile, i.e. every 16-byte operand load of a wave is 1 KB of
   * consecutive memory (kern.cross_fold_pack); ld_wq / ld_w2 are ignored, s_wq / s_w2 still step images. */
  int packed;
} SdmiCrossFoldArgs;
int sdmi_cross_fold(const SdmiCrossFoldArgs* a, void* stream);

/* Head-expanded slot keys / values for the folded cross-attention (engine.UNetRunner.cross_fold):
 * kv [B][S][ldkv] holds K in columns [0, C) and V in [C, 2C); row h * gw + j (j < S <= gw) of
 * kexp / vexp [B][heads * gw][C] is slot j's key (times `scale`) / value restricted to the channels of
 * head h, zero elsewhere; rows j >= S are zero.  gw = group width: 8 (0 = 8) or 16. */
typedef struct {
  const void* kv; void* kexp; void* vexp; int dtype; int B, S, C, heads, ldkv; float scale;
  int gw;
} SdmiExpandHeadsArgs;
int sdmi_expand_heads(const SdmiExpandHeadsArgs* a, void* stream);

/* Batched gather of 2-byte elements: dst[b][i] = src[b][idx[i]], i < n (n a multiple of 8), idx a DEVICE array of n
 * 64-bit element offsets shared by all b; s_src / s_dst = batch pitches in elements.  Re-lays the per-image operands
 * of the folded slot cross-attention into unit-stream / MFMA-fragment order once per sampling call
 * (kern.Kern.cross_prepare / cross_block: replaces torch.index_select on the timed path). */
typedef struct { const void* src; void* dst; const long long* idx; long long n, s_src, s_dst; int B; } SdmiGatherRowsArgs;
int sdmi_gather_rows(const SdmiGatherRowsArgs* a, void* stream);

/* Small stream-ordered helpers that keep the training step free of framework kernels:
 *   sdmi_memset0 ...... zero `bytes` bytes (gradient arena, padded buffers) -- a memset node in a graph
 *   sdmi_scale_dev .... y = x * s[0], s a DEVICE scalar (upstream gradient of the loss)
 *   sdmi_counters_inc . ++step[0] (int) and ++seed[0] (long long), either may be NULL (Adam step,
 *                       dropout seed word: advanced inside the captured step)
 *   sdmi_draw_tn ...... the training draws of LDM.loss_function (ldm.py:65-69) in one launch:
 *                       t[b] ~ U{0..T-1} (int64) with tf[b] = (float) t[b], ca[b] = tab_a[t[b]],
 *                       cb[b] = tab_b[t[b]] (sqrt_alphas_bar / sqrt_one_minus_alphas_bar) and
 *                       noise ~ N(0,1) as NHWC fp32 [B][hw][4] (channel 3 = 0); counter-based
 *                       generator keyed on seed + *seed_dev (a graph replay draws new values). */
typedef struct { void* ptr; long long bytes; } SdmiMemsetArgs;
int sdmi_memset0(const SdmiMemsetArgs* a, void* stream);
typedef struct { const void* x; void* y; const float* s; int dtype; long long n; } SdmiScaleDevArgs;
int sdmi_scale_dev(const SdmiScaleDevArgs* a, void* stream);
typedef struct { int* step; long long* seed; } SdmiCountersArgs;
int sdmi_counters_inc(const SdmiCountersArgs* a, void* stream);
typedef struct {
  long long* t; float* tf; float* ca; float* cb; float* noise;
  const float* tab_a; const float* tab_b;
  int B, T; long long per;     /* per = h*w*4 floats of noise per image */
  long long seed; const long long* seed_dev;
} SdmiDrawTnArgs;
int sdmi_draw_tn(const SdmiDrawTnArgs* a, void* stream);

/* sinusoidal timestep embedding [cos | sin], fractional t (unet/utils.py:70-92) -> fp32 [B][dim],
 * optionally followed by nothing (MLP runs on the GEMM). */
typedef struct { const float* t; float* out; int B, dim; float max_period; } SdmiTimeEmbArgs;
int sdmi_timestep_embedding(const SdmiTimeEmbArgs* a, void* stream);

/* y = act(x) elementwise with dtype conversion (SiLU on the time embedding) */
typedef struct { const void* x; void* y; int src_dtype, dst_dtype, act; long long n; } SdmiActArgs;
int sdmi_act(const SdmiActArgs* a, void* stream);
/* GEGLU: y[r][c] = h[r][c] * gelu(h[r][C+c]) (attention.py:46-48); h [rows][2C] */
typedef struct { const void* h; void* y; int dtype; long long rows; int C; } SdmiGegluArgs;
int sdmi_geglu(const SdmiGegluArgs* a, void* stream);
typedef struct { const void* h; const void* dy; void* dh; int dtype; long long rows; int C; } SdmiGegluBwdArgs;
int sdmi_geglu_bwd(const SdmiGegluBwdArgs* a, void* stream);
/* y[b][p][c] = x[b][p][c] + pos[p][c] (SoftPositionEmbed, models/utils.py:60-63) */
typedef struct { const void* x; const float* pos; void* y; int dtype; int B; long long per; } SdmiAddPosArgs;
int sdmi_add_pos(const SdmiAddPosArgs* a, void* stream);
/* Plain-SA spatial-broadcast decoder (img_based/models/slot_attention.py:343-364):
 * y[g][r][c] = x[g][c] + pos[r][c]: every slot vector broadcast over the R decoder positions plus
 * the decoder position embedding (x, pos fp32; y in `dtype`). */
typedef struct { const float* x; const float* pos; void* y; int dtype; int G, R, C; } SdmiBroadcastPosArgs;
int sdmi_broadcast_pos(const SdmiBroadcastPosArgs* a, void* stream);
/* o [B*N][HW][ldo] (channels 0..2 rgb, 3 alpha logit) -> masks[b][n][p] = softmax_n(alpha),
 * recon[b][p][0..2] = sum_n rgb*mask (fp32, pitch 4, channel 3 = 0).  slot_attention.py:357-363. */
typedef struct { const void* o; float* recon; float* masks; int dtype; int B, N, HW, ldo; } SdmiSaCombineArgs;
int sdmi_sa_combine(const SdmiSaCombineArgs* a, void* stream);
/* backward: dout[bn][p][0..2] = mask*drecon, [3] = mask*(s_n - sum_m mask_m s_m), s_n = rgb_n.drecon */
typedef struct {
  const void* o; const float* masks; const float* drecon; void* dout; int dtype; int B, N, HW, ldo;
} SdmiSaCombineBwdArgs;
int sdmi_sa_combine_bwd(const SdmiSaCombineBwdArgs* a, void* stream);
/* channel concat of two NHWC tensors (skip connections, unet.py:572) */
typedef struct { const void* a; const void* b; void* y; int dtype; long long rows; int Ca, Cb; } SdmiConcatArgs;
int sdmi_concat_channels(const SdmiConcatArgs* a, void* stream);
/* eval-time mask path: bilinear x(H/h) upsample (align_corners=False) of seg [B][h*w][N] + argmax
 * over slots -> masks_up [B][N][H][W] fp32 (optional) and idx [B][H][W] int64
 * (sa_diffusion.py:170-180, test_seg.py argmax). */
typedef struct {
  const float* seg; float* up; long long* idx; int B, N, h, w, H, W;
} SdmiMaskUpArgs;
int sdmi_mask_upsample_argmax(const SdmiMaskUpArgs* a, void* stream);
/* mean squared error: out[0] = mean((a-b)^2) (deterministic two-stage), grad optional:
 * dpred = 2*(a-b)/n * gscale   (ldm.py:82) */
typedef struct {
  const void* pred; const float* target; float* out; void* dpred; float* partial;
  int dtype; long long n; int nblk; float gscale;
  int l1;        /* 1: mean |pred - target| (VQ-VAE reconstruction loss, vqvae/loss.py:27), dpred = sign * gscale / n */
  float oscale;  /* != 0: out[0] = mean * oscale (the 4/3 channel-pad correction of the NHWC pair) */
} SdmiMseArgs;
int sdmi_mse(const SdmiMseArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Backward helpers.
 * ------------------------------------------------------------------------------------------ */
/* dgrad operand: dst[ci][kh'][kw'][co] = src[co][KH-1-kh'][KW-1-kw'][ci]  (flip + transpose), so the
 * data gradient of a conv is sdmi_igemm on dY with this operand (pad' = K-1-pad, zins = stride). */
typedef struct { const void* src; void* dst; int dtype; int Cout, KH, KW, Cin, CoutPad; } SdmiPackDgradArgs;
/* dst row pitch CoutPad >= Cout (pad columns must be pre-zeroed by the caller) */
int sdmi_pack_dgrad(const SdmiPackDgradArgs* a, void* stream);
/* The same for a whole table of operands in ONE launch (every dgrad operand of the model is
 * rebuilt after each optimiser step).  `descs` is a device array of SdmiPackDesc; workgroup b
 * serves the descriptor with block_begin <= b < next block_begin; an operand takes
 * KH*KW * ceil(Cout/64) * ceil(Cin/64) workgroups (one 64x64 tile of one tap each). */
typedef struct {
  const void* src; void* dst;
  int Cout, KH, KW, Cin, CoutPad;
  int block_begin;
  /* optional tap selection (nkh > 0): dst holds only the taps kh' = kh0 + i*kstep (i < nkh),
   * kw' = kw0 + j*kstep (j < nkw) of the flipped filter, [Cin][nkh][nkw][CoutPad] -- the parity
   * sub-filters of a strided convolution's data gradient; the operand takes nkh*nkw tap slices. */
  int kh0, kw0, kstep, nkh, nkw;
} SdmiPackDesc;
typedef struct { const void* descs; int n_desc; int dtype; int total_blocks; } SdmiPackBatchArgs;
int sdmi_pack_dgrad_batch(const SdmiPackBatchArgs* a, void* stream);
/* out[g][n] = sum over the `rows_per` consecutive rows of group g of x[m][n] (fp32 out):
 * gradient of the per-image time-embedding row vector, bias gradients (rows_per = M). */
typedef struct { const void* x; float* out; int dtype; int groups, rows_per, N, ldx;
                 int ldo; /* row pitch of out (0 = N): results written into a column slice of a wider matrix */
} SdmiRowGroupSumArgs;
int sdmi_rowgroup_sum(const SdmiRowGroupSumArgs* a, void* stream);
/* y[b][y][x][c] = sum of the 2x2 block of x (backward of the nearest x2 upsample folded into a conv) */
typedef struct { const void* x; void* y; int dtype; int B, H, W, C; } SdmiPool2x2Args;
int sdmi_pool2x2_sum(const SdmiPool2x2Args* a, void* stream);
/* y = x + z elementwise in `dtype` (gradient accumulation at residual / skip joins) */
typedef struct { const void* x; const void* z; void* y; int dtype; long long n; } SdmiAddArgs;
int sdmi_add(const SdmiAddArgs* a, void* stream);
/* dst_i[0 .. count_i) += src_i[0 .. count_i) for up to 32 fp32 segments in ONE launch: the weight
 * gradient of a GEMM that was fused across parameters which are not adjacent in the gradient arena
 * (the 22 time-embedding projections of the UNet run as one GEMM; unet.py:233-236, 268) goes back
 * to its parameters this way.  `items` is a HOST array of n SdmiScatterItem; the launch carries the
 * table by value (graph-capturable). */
typedef struct { const float* src; float* dst; long long count; } SdmiScatterItem;
typedef struct { const void* items; int n; } SdmiScatterAddArgs;
int sdmi_scatter_add(const SdmiScatterAddArgs* a, void* stream);
/* Up to 32 plain device-to-device copies in ONE launch (`items`: HOST array of n SdmiCopyItem, table by
 * value): the fused operand / bias of a GEMM over parameters that are not adjacent in the arena is
 * re-gathered after every optimiser step this way (22 blocks -> 1 launch instead of 44). */
typedef struct { const void* src; void* dst; long long bytes; } SdmiCopyItem;
typedef struct { const void* items; int n; } SdmiCopyGroupArgs;
int sdmi_copy_group(const SdmiCopyGroupArgs* a, void* stream);
/* EMA of the denoiser weights (LitEma.forward, ddpm/ema.py:29-52):
 * shadow[i] -= one_minus_decay * (shadow[i] - p[i]) over a contiguous fp32 arena range. */
typedef struct { float* shadow; const float* p; long long n; float one_minus_decay; } SdmiEmaArgs;
int sdmi_ema_update(const SdmiEmaArgs* a, void* stream);
/* inverted dropout with a counter-based generator: y[i] = keep(seed, i) ? x[i] / (1-p) : 0, keep
 * derived from a 64-bit mix of (seed, i) so the backward pass regenerates the same mask from the
 * seed instead of storing it (ResBlock dropout, unet.py:246; p = 0.1 in every LDM config). */
typedef struct {
  const void* x; void* y; int dtype; long long n; float p; long long seed;
  const long long* seed_dev;   /* optional device word added to `seed` (HIP-graph replays advance it) */
} SdmiDropoutArgs;
int sdmi_dropout(const SdmiDropoutArgs* a, void* stream);
/* split of a channel concat: a[r][:Ca] = y[r][:Ca], b[r][:Cb] = y[r][Ca:] */
typedef struct { const void* y; void* a; void* b; int dtype; long long rows; int Ca, Cb; } SdmiSplitArgs;
int sdmi_split_channels(const SdmiSplitArgs* a, void* stream);
/* dx = dy * act'(x) (SiLU on the time embedding path) */
typedef struct { const void* x; const void* dy; void* dx; int dtype, act; long long n; } SdmiActBwdArgs;
int sdmi_act_bwd(const SdmiActBwdArgs* a, void* stream);

/* Training-mode Slot Attention (one iteration's streaming pass + its backward; the [B*N, D]
 * mat-vecs of the iteration run on sdmi_igemm / sdmi_wgrad).  sa_diffusion.py:40-58.
 *   attn [B][M][N] = softmax over slots;  den [B][N] = sum_m (attn+eps);  upd [B][N][D]. */
typedef struct {
  const void* k; const void* v; const float* q; float* attn; float* upd; float* den;
  int dtype; int B, M, N, D, ldkv; float eps, scale;
  float* workspace;   /* optional, B*ceil(M/64)*N*(D+1) floats: enables the token-tiled kernels
                         (one workgroup per 64/128 tokens + a finalize pass) for N <= 8 */
} SdmiSaAttendArgs;
int sdmi_sa_attend_fwd(const SdmiSaAttendArgs* a, void* stream);
typedef struct {
  const void* k; const void* v; const float* q; const float* attn; const float* upd;
  const float* den; const float* dupd; float* dq; void* dk; void* dv;
  int dtype; int B, M, N, D, ldkv; float eps, scale;
  float* workspace;   /* optional, B*ceil(M/64)*N*D floats (see SdmiSaAttendArgs) */
  int accumulate;     /* != 0: dk / dv += (k, v feed every iteration: the iterations' gradients are
                         summed in place instead of by separate accumulation kernels) */
} SdmiSaAttendBwdArgs;
int sdmi_sa_attend_bwd(const SdmiSaAttendBwdArgs* a, void* stream);
/* GRUCell gate arithmetic on gi = W_ih x + b_ih, gh = W_hh h + b_hh ([R][3D], gates r,z,n). */
typedef struct { const float* gi; const float* gh; const float* h; float* hout; int R, D; } SdmiGruGatesArgs;
int sdmi_gru_gates(const SdmiGruGatesArgs* a, void* stream);
typedef struct {
  const float* gi; const float* gh; const float* h; const float* dhout;
  float* dgi; float* dgh; float* dh; int R, D;
} SdmiGruGatesBwdArgs;
int sdmi_gru_gates_bwd(const SdmiGruGatesBwdArgs* a, void* stream);

/* ------------------------------------------------------------------------------------------
 * Optimiser: global-norm clip + Adam over a flat fp32 arena (two lr groups), and bf16 shadow
 * refresh.  Replaces torch.optim.Adam + clip_grad_norm_ (nerv trainer; vb/method.py:291-341).
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const void* g; float* partial; long long n; int nblk;
  int g_dtype;                 /* SDMI_F32 (default) or SDMI_BF16: the gradients as they came off a bf16 wire */
} SdmiSqSumArgs;
int sdmi_sqsum_partial(const SdmiSqSumArgs* a, void* stream);
typedef struct {
  float* p; const void* g; float* m; float* v; void* shadow_bf16; /* optional */
  const float* sq_partial; int nblk;     /* global norm^2 = sum(sq_partial[0..nblk)) */
  long long n; float lr, beta1, beta2, eps, clip; int step;
  const float* lr_dev;         /* optional device scalar overriding `lr` (graph-replayable schedule) */
  const int* step_dev;         /* optional device step counter overriding `step` */
  float gscale;                /* gradients (and the norm the clip sees) are g * gscale; 0 = 1.  Data parallel: g is
                                  the SUM over ranks as the all-reduce left it, gscale = 1 / world -- no separate
                                  averaging pass over the arena */
  int g_dtype;                 /* SDMI_F32 (default) or SDMI_BF16: `g` points at the bf16 wire buffer (same offsets
                                  as the fp32 gradient arena) -- no widening pass after a bf16 all-reduce */
} SdmiAdamArgs;
int sdmi_adam_clip(const SdmiAdamArgs* a, void* stream);


/* ------------------------------------------------------------------------------------------
 * Evaluation metrics on device (video_based/models/eval_utils.py:119-320).
 * The segmentation scores (ARI / FG-ARI, Hungarian mIoU, mBO) are all functions of the per-image
 * contingency table of (ground-truth id, predicted id) pairs, which is exact integer work:
 *   counts[b][c][k] = #{p : gt[b][p] == c and pred[b][p] == k}     (ids outside the table are skipped)
 * replaces the one_hot + einsum("bthwc,bthwk->bck") of adjusted_rand_index (eval_utils.py:148-157)
 * and the one_hot products of hungarian_miou / mean_best_overlap (238-290).  `counts` must be
 * zeroed by the caller.
 * ------------------------------------------------------------------------------------------ */
typedef struct {
  const int* gt; const int* pred; int* counts;     /* [B][P], [B][P], [B][Kg][Kp] */
  int B; long long P; int Kg, Kp;                   /* Kg * Kp <= 8192 */
} SdmiContingencyArgs;
int sdmi_contingency(const SdmiContingencyArgs* a, void* stream);
/* Per-image squared error for mse_metric / psnr_metric (eval_utils.py:75-92):
 *   partial[b][j] = sum over the j-th of `nchunk` equal slices of (x[b][i] - y[b][i])^2, in fp64
 * (the caller adds the nchunk partials of an image in index order). */
typedef struct {
  const float* x; const float* y; double* partial; int B; long long n; int nchunk;
  int mode;                      /* 0: (x-y)^2   1: |x-y| (L1 reconstruction loss, vqvae/loss.py:27) */
} SdmiSqErrArgs;
int sdmi_sqerr_rows(const SdmiSqErrArgs* a, void* stream);
/* Structural similarity (eval_utils.py:91-106 -> skimage.metrics.structural_similarity with
 * gaussian_weights=True, sigma=1.5, use_sample_covariance=False, data_range=L, channel_axis=0;
 * skimage is a third-party dependency absent here: the published algorithm, Wang et al. 2004, is
 * restated -- parity with skimage itself is unpinned).  Per plane p of x, y [P][H][W] fp32:
 *   out[p] = mean over the interior (5-pixel border cropped) of
 *            ((2 ux uy + C1)(2 vxy + C2)) / ((ux^2 + uy^2 + C1)(vx + vy + C2)),
 * ux, uy, vx, vy, vxy = 11x11 Gaussian-window moments (sigma 1.5, truncate 3.5), C1 = (0.01 L)^2,
 * C2 = (0.03 L)^2.  fp64 arithmetic; out [P] fp64. */
typedef struct { const float* x; const float* y; double* out; int P, H, W; float data_range; } SdmiSsimArgs;
int sdmi_ssim(const SdmiSsimArgs* a, void* stream);


/* ------------------------------------------------------------------------------------------
 * VQ-VAE stage-1 training helpers (vqvae/modules.py:113-154, quantize.py:98-107).
 * The AttnBlock is one head of width C over h*w tokens: its backward runs as batched GEMMs
 * (sdmi_igemm) around these two kernels.
 * ------------------------------------------------------------------------------------------ */
/* batched 2-D transpose dst[z][c][r] = src[z][r][c] (LDS tile transpose, both sides coalesced) */
typedef struct {
  const void* src; void* dst; int dtype; int Z, R, C;
  long long lds, ldd, ss, sd;                 /* row pitches and batch strides, in elements */
} SdmiTransposeArgs;
int sdmi_transpose2d(const SdmiTransposeArgs* a, void* stream);
/* softmax backward in place on dp: ds[r][j] = scale * p[r][j] * (dp[r][j] - sum_k dp[r][k] p[r][k]) */
typedef struct { const void* p; void* dp; int dtype; int rows, cols, ld; float scale; } SdmiSoftmaxBwdArgs;
int sdmi_softmax_rows_bwd(const SdmiSoftmaxBwdArgs* a, void* stream);
/* Straight-through quantizer backward (quantize.py:98-107, legacy loss):
 *   dz[r][d]          = dzq[r][d] + g * 2/n * (z - zq)            (g = d loss / d quant_loss, device scalar)
 *   dcode[idx[r]][d] += g * 2*beta/n * (zq - z)                    (fp32 atomics into the codebook gradient)
 * z / zq / dzq / dz are [R][ldz] fp32 rows using the first `dim` columns; n = R * dim. */
typedef struct {
  const float* z; const float* zq; const float* dzq; float* dz; float* dcode;
  const long long* idx; const float* g; long long R; int dim, ldz; float beta;
} SdmiVqBwdArgs;
int sdmi_vq_bwd(const SdmiVqBwdArgs* a, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* SDMI_H_ */
