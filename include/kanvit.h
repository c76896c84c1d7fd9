/*
 * kanvit.h -- C ABI of the MI355X (gfx950) hot-path library for the ViKANformer
 * reference (akshathmangudi/KAN-ViT).
 *
 * The reference has no native / FFI layer at all: its boundary is the Python
 * nn.Module surface.  This header is the boundary the
 * replacement adds UNDERNEATH that surface; each entry point below cites the
 * reference function whose arithmetic it replaces (paths relative to the
 * reference repository root).
 *
 * Conventions (all entry points)
 *   - plain C types only; every pointer is a BORROWED device pointer to fp32
 *     data owned by the caller (a torch tensor, a hipMalloc block, ...), which
 *     must stay alive until the work queued on `stream` has completed;
 *   - asynchronous on `stream` (a hipStream_t passed as void*; NULL = default
 *     stream); no internal synchronisation, no allocation, graph-capturable;
 *   - returns 0 on success, a negative KANVIT_E* code otherwise; never throws,
 *     never exits; kanvit_last_error() gives a thread-local message;
 *   - re-entrant and thread safe for distinct streams.
 */
#ifndef KANVIT_H
#define KANVIT_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The library is built with -fvisibility=hidden: only what this header declares is exported. */
#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility push(default)
#endif

#define KANVIT_ABI_VERSION 7

/* error codes */
#define KANVIT_OK 0
#define KANVIT_EINVAL (-22)      /* bad descriptor / unsupported shape          */
#define KANVIT_ENOMEM (-12)      /* workspace too small                         */
#define KANVIT_EDEVICE (-5)      /* HIP runtime error (no device, launch error) */

/* descriptor flags */
#define KANVIT_FLAG_BF16_MFMA 1  /* contract on the bf16 matrix cores (operands rounded to bf16, fp32 accumulate, fp32
                                    I/O); used for the bf16 configurations, never for the fp32 parity path.  Shapes
                                    the bf16 kernels do not cover silently use the exact fp32 kernels.               */

#define KANVIT_FLAG_UNIFORM_KNOTS 2 /* BSPLINE, spline_order 3: every knot row is g0 + j*h (the layout the reference builds
                                    at models/effkan.py:44-53 and never changes) -> closed-form cubic evaluation of
                                    the 4 non-zero bases instead of the Cox-de Boor recursion.  g0, h are read from
                                    the first two knots of each group.
                                    RBF with G = 8: the centres are c0 + j/rbf_inv_h (FastKAN's own grid, models/
                                    fastkan.py:22-27: linspace centres, denominator = their spacing) -> the eight
                                    Gaussians come from two exp anchors and a two-step recurrence instead of eight
                                    exps (relative error <= ~2e-6 where the anchor exp(-(t-2)^2) / exp(-(t-5)^2) is a
                                    normal number; beyond, values below 4e-24 are flushed to 0: absolute error
                                    <= 4e-24); the caller vouches for the layout.                                  */
#define KANVIT_FLAG_FUSED_LN 8   /* RBF (FastKAN): the LayerNorm in front of the spline path (models/fastkan.py:68) is formed IN the
                                    kernels: u = (x - mean) * rstd * gamma + beta over the I features of the group's x slice,
                                    eps = ln_eps.  bparams of a group = [centres(G) | gamma(I) | beta(I)].  The `u` argument
                                    of the three entry points then carries a statistics buffer float[M][x_group_mod][2]
                                    (mean, rstd per row and x slice): WRITTEN by kanvit_layer_fwd, read by the backward
                                    calls; `du` still receives d loss / d u (the caller applies the LayerNorm backward).
                                    Only shapes for which kanvit_layer_ln_fusable() returns 1.                          */
#define KANVIT_FLAG_SHARED_BPARAMS 4 /* groups that read the same x columns (q, k, v of a head) also have identical
                                    basis parameters, so one basis tile may serve all of them (as for the
                                    parameter-free families).  Honoured for BSPLINE.                                */

#define KANVIT_FLAG_SINE_DFREQ 16 /* SINE, kanvit_layer_bwd_weight only: the generated operand is x * cos(x f_g + p_ig) instead of
                                    sin(x f_g + p_ig), so the result is Q[(i,g)][o] = sum_m dy[m][o] x[m][i] cos(...) and
                                    d loss / d freq_g = sum_{i,o} w[(i,g)][o] Q[(i,g)][o] -- the frequency gradient of a layer
                                    whose INPUT gradient nobody needs (the patch embedding: its input is the image) at the
                                    cost of a second weight-gradient pass instead of the slower input-gradient contraction
                                    (models/sinekan.py:81-91; freq is trainable, sinekan.py:60).  Shapes the register
                                    weight-gradient kernels do not cover are refused (EINVAL).                          */

/* basis families: phi_g(x) generated on the fly, never stored in HBM */
#define KANVIT_LINEAR 0   /* phi = x                               nn.Linear in attention.py:136-142           */
#define KANVIT_CHEBY 1    /* T_g(tanh x), g = 0..degree            models/cheby.py:36-48                       */
#define KANVIT_BSPLINE 2  /* Cox-de Boor B-splines (+ silu base)   models/effkan.py:99-132,174-187             */
#define KANVIT_RBF 3      /* exp(-((u-c_g)/h)^2) (+ silu base)     models/fastkan.py:29-30,66-76               */
#define KANVIT_SINE 4     /* sin(x f_g + p_ig)                     models/sinekan.py:81-91                     */
#define KANVIT_FOURIER 5  /* cos(k x), sin(k x), k = 1..G          models/nfkan.py:36-52                       */

/*
 * One launch evaluates `groups` independent KAN layers that share M rows:
 *   y[m, g*O + o] = bias[g][o] + sum_i sum_j phi_j(x[m, (g % x_group_mod)*I + i]) * w[g][i*GP + j][o]
 * groups = 1 is a plain layer (patch embedding, model.py:146); groups = 3*H with
 * x_group_mod = H is the per-head q|k|v mapping of MSA (attention.py:191-197)
 * folded over the batch, group index = proj*H + head.
 *
 * GP (generated columns per input feature):
 *   LINEAR 1 | CHEBY G | BSPLINE G+has_base | RBF G+has_base | SINE G | FOURIER 2G
 * Packed weights w: [groups][I*GP][O] row-major, k = i*GP + j, with j ordered
 *   CHEBY   j = degree d                      (cheby_coeffs[i][o][d])
 *   BSPLINE j = basis 0..G-1, then base       (spline_weight*spline_scaler [o][i][j], base_weight[o][i])
 *   RBF     j = centre 0..G-1, then base      (spline_linear.weight[o][i*G+j], base_linear.weight[o][i])
 *   SINE    j = g                             (amplitudes[o][i][g])
 *   FOURIER j = c*G + (k-1), c = 0 cos, 1 sin (fouriercoeffs[c][o][i][k-1])
 * Basis parameters bparams: [groups][bparam_stride] floats per group
 *   BSPLINE knots[I][G+spline_order+1] | RBF centres[G] | SINE freq[G] then phase[I][G] | others: none (NULL)
 */
typedef struct kanvit_layer_desc {
    int32_t family;        /* KANVIT_*                                                      */
    int32_t groups;        /* independent layers in this launch (>= 1)                      */
    int32_t x_group_mod;   /* group g reads x columns [(g % x_group_mod)*I, +I)             */
    int32_t I;             /* input features per group                                      */
    int32_t O;             /* output features per group                                     */
    int32_t G;             /* cheby: degree+1; bspline: grid_size+spline_order; rbf: num_grids;
                              sine: grid_size; fourier: gridsize; linear: 1                 */
    int32_t spline_order;  /* BSPLINE only                                                  */
    int32_t has_base;      /* BSPLINE / RBF: extra silu(x) column per input feature         */
    float rbf_inv_h;       /* RBF: 1 / denominator                                          */
    int32_t flags;         /* KANVIT_FLAG_*                                                  */
    int64_t M;             /* rows                                                          */
    int64_t ldx;           /* row stride (floats) of x and dx                               */
    int64_t ldu;           /* row stride of u and du (RBF; group g uses columns [g*I, +I))  */
    int64_t ldy;           /* row stride of y and dy (group g uses columns [g*O, +O))       */
    int64_t bparam_stride; /* floats between consecutive groups in bparams                  */
    float ln_eps;          /* KANVIT_FLAG_FUSED_LN: epsilon of the fused LayerNorm          */
    int32_t reserved;
} kanvit_layer_desc;

/* 1 if KANVIT_FLAG_FUSED_LN may be set for this layer (register kernels cover forward and both gradients), else 0 */
int kanvit_layer_ln_fusable(const kanvit_layer_desc* d);

/* Backward of the fused LayerNorm (what autograd does for models/fastkan.py:68's nn.LayerNorm), given du from
 * kanvit_layer_bwd_input and the statistics kanvit_layer_fwd wrote:  dx[M][ldx] += LN-backward (IN PLACE, may be NULL),
 * dgamma / dbeta [groups][I] = column sums.  Deterministic (fixed-order partial sums in the workspace).              */
size_t kanvit_layer_ln_bwd_workspace(const kanvit_layer_desc* d);
int kanvit_layer_ln_bwd(const kanvit_layer_desc* d, const float* x, const float* stats, const float* bparams, const float* du,
                        float* dx, float* dgamma, float* dbeta, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused basis evaluation + coefficient contraction -------------------------------------
 * forward of models/cheby.py:36-48, models/effkan.py:174-187, models/fastkan.py:66-76 (the
 * LayerNorm of :68 is applied by the caller and passed as u; u = NULL means u = x),
 * models/nfkan.py:36-52, models/sinekan.py:81-91, and nn.Linear (attention.py:136-142).
 * With KANVIT_FLAG_FUSED_LN the LayerNorm is applied inside the kernel and `u` is the statistics buffer (see the flag). */
size_t kanvit_layer_fwd_workspace(const kanvit_layer_desc* d);   /* 0 unless KANVIT_FLAG_BF16_MFMA (repacked weights) */
int kanvit_layer_fwd(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,
                     const float* bparams, const float* bias, float* y, void* workspace, size_t workspace_bytes,
                     void* stream);

/* ---- backward w.r.t. the layer input (what torch.autograd derives for the reference) ------
 * dx[m, c*I + i] = sum over the groups g with g % x_group_mod == c of
 *                  sum_j phi_j'(x) * sum_o dy[m, g*O+o] * w[g][i*GP+j][o]     (written, not accumulated)
 * RBF: du[m, g*I+i] gets the spline-path gradient, dx the base-path gradient.
 * SINE: dparam receives per-row-tile partial sums of d loss / d freq, shape
 *       [kanvit_layer_dparam_tiles(d)][groups][G]; the caller sums over dim 0.                */
size_t kanvit_layer_bwd_input_workspace(const kanvit_layer_desc* d);   /* 0 unless KANVIT_FLAG_BF16_MFMA */
int kanvit_layer_bwd_input(const kanvit_layer_desc* d, const float* x, const float* u, const float* w,
                           const float* bparams, const float* dy, float* dx, float* du, float* dparam,
                           void* workspace, size_t workspace_bytes, void* stream);
int64_t kanvit_layer_dparam_tiles(const kanvit_layer_desc* d);
/* 1 if KANVIT_FLAG_SINE_DFREQ may be set for kanvit_layer_bwd_weight on this layer (SINE, a shape the register weight-gradient
 * kernels cover), else 0.  The flag is refused by kanvit_layer_fwd and kanvit_layer_bwd_input.  (ABI >= 6) */
int kanvit_layer_sine_dfreq_ok(const kanvit_layer_desc* d);

/* ---- backward w.r.t. the packed weights ---------------------------------------------------
 * dw[g][i*GP+j][o] = sum_m phi_j(x[m, ...i]) * dy[m, g*O+o]; split over row ranges into
 * partial slabs in `workspace`, then reduced in a fixed order (deterministic, no atomics).   */
size_t kanvit_layer_bwd_weight_workspace(const kanvit_layer_desc* d);
int kanvit_layer_bwd_weight(const kanvit_layer_desc* d, const float* x, const float* u, const float* bparams,
                            const float* dy, float* dw, void* workspace, size_t workspace_bytes, void* stream);

/* ---- fused patch embedding --------------------------------------
 * The patch-embedding layer of VisionTransformer.forward (model.py:144-152) with its prologue and epilogue inside the
 * kernel: the rows of x are gathered straight from the NCHW image batch -- row m = (image m / P, patch m % P),
 * feature i = (c, iy, ix) of the patch, i.e. model.py:111-126's patchify without the [B, P, I] staging tensor -- and the
 * output is written as the token sequence the first transformer block reads:
 *     y[b, prepend_rows + p, :] = layer(patch(b, p)) + pos[prepend_rows + p, :]
 *     y[b, 0, :]                = cls + pos[0, :]                    (prepend_rows = 1; model.py:150-152)
 * d describes the layer as for kanvit_layer_fwd with M = B*P rows, I = C*ph*pw, groups = 1; `ldy` is the row stride of y,
 * whose row count is B*(P + prepend_rows).  pos ([P + prepend_rows][O]) and cls ([O]) may be NULL (nothing added /
 * prepend_rows = 0).  Runs on the register-form kernels only: returns KANVIT_EINVAL for shapes they do not cover
 * (O % 32, a feature chunk that does not divide the patch width, FastKAN, which needs u = LayerNorm(x)) -- the caller
 * then uses patchify + kanvit_layer_fwd.  With KANVIT_FLAG_BF16_MFMA the bf16 register-form forward runs the same gather
 * (workspace as kanvit_layer_fwd_workspace(d) reports, ABI >= 7; without the flag no workspace is needed).
 *
 * kanvit_patch_embed_bwd_weight (ABI >= 7): the layer's weight gradient with the same gather on both operands --
 *     dw[i*GP + j][o] = sum_{b,p} phi_j(patch(b, p)[i]) * dy[b, prepend_rows + p, o]
 * x rows come from the NCHW images, dY rows from the token-sequence gradient [B][P + prepend_rows][ldy] as the first block's
 * backward leaves it (class-token rows are stepped over): no transient [B*P, I] patch matrix, no copy of dY.  d as above
 * (M = B*P, `ldy` = row stride of dy); KANVIT_FLAG_SINE_DFREQ as for kanvit_layer_bwd_weight.  Exact fp32: with
 * KANVIT_FLAG_BF16_MFMA (whose short MFMA phases cannot hide the row walker: measured slower than the copy it saves)
 * kanvit_patch_embed_bwd_weight_ok() answers 0.
 * kanvit_patch_embed_bwd_weight_ok() (pure host function) says whether the gathering kernels cover the layer: the patch
 * embeddings VisionTransformer builds (model.py:67-80) with ChebyKAN degree 4, SineKAN and FourierKAN at grid 28 (I, O
 * multiples of 32, M >= 256, 32-bit element offsets); otherwise (efficient-KAN: its weight-gradient kernels have no register
 * left for the row walker and measured slower with it; bf16 mode) the call returns KANVIT_EINVAL and the caller uses
 * patchify + kanvit_layer_bwd_weight. */
typedef struct kanvit_patch_desc {
    int32_t C, H, W;         /* image batch is [B][C][H][W], contiguous                                  */
    int32_t n_patches;       /* patches per side: patch = (H / n_patches) x (W / n_patches) pixels       */
    int32_t prepend_rows;    /* 0, or 1 = a class-token row in front of every image's patch tokens       */
    int32_t reserved;
} kanvit_patch_desc;
int kanvit_patch_embed_fwd(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const float* images, const float* w,
                           const float* bparams, const float* bias, const float* cls, const float* pos, float* y,
                           void* stream);
int kanvit_patch_embed_fwd_ws(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const float* images, const float* w,
                              const float* bparams, const float* bias, const float* cls, const float* pos, float* y,
                              void* workspace, size_t workspace_bytes, void* stream);
int kanvit_patch_embed_bwd_weight_ok(const kanvit_layer_desc* d, const kanvit_patch_desc* p);
size_t kanvit_patch_embed_bwd_weight_workspace(const kanvit_layer_desc* d, const kanvit_patch_desc* p);
int kanvit_patch_embed_bwd_weight(const kanvit_layer_desc* d, const kanvit_patch_desc* p, const float* images,
                                  const float* bparams, const float* dy, float* dw, void* workspace, size_t workspace_bytes,
                                  void* stream);

/* ---- multi-head attention core ------------------------------------------------------------
 * o = softmax(q k^T * scale) v per (batch, head); replaces attention.py:199-200 (MSA) and
 * utils.py:137-227 (FlashAttentionFunction.forward; lse is what it saves for backward).
 * Element (b, h, n, c) of q/k/v/o lives at base[b*stride_b + h*stride_h + n*stride_n + c];
 * lse is [B][H][N] contiguous.  D even, D <= KANVIT_ATTN_MAX_D; one head must fit the 160 KiB
 * LDS of a CU: N <= 256 for D <= 32, N <= 224 for D <= 64.                                   */
#define KANVIT_ATTN_MAX_N 256
#define KANVIT_ATTN_MAX_D 64
typedef struct kanvit_attn_desc {
    int32_t B, H, N, D;
    int32_t causal;         /* utils.py:177-180 with equal q/k lengths */
    float scale;            /* D^-1/2 in both callers */
    int32_t flags;          /* KANVIT_FLAG_BF16_MFMA: products on the bf16 matrix cores (needs D % 16 == 0) */
    int32_t reserved;
    int64_t q_stride_b, q_stride_h, q_stride_n;
    int64_t k_stride_b, k_stride_h, k_stride_n;
    int64_t v_stride_b, v_stride_h, v_stride_n;
    int64_t o_stride_b, o_stride_h, o_stride_n;   /* also the layout of do; dq/dk/dv use the q/k/v strides */
} kanvit_attn_desc;

int kanvit_attn_fwd(const kanvit_attn_desc* d, const float* q, const float* k, const float* v,
                    float* o, float* lse, void* stream);
/* backward of the above; replaces utils.py:229-295 (recompute p from q, k, lse).
 * workspace: kanvit_attn_bwd_workspace(d) bytes (holds rowsum(dO*O), "D" of utils.py:286). */
size_t kanvit_attn_bwd_workspace(const kanvit_attn_desc* d);
int kanvit_attn_bwd(const kanvit_attn_desc* d, const float* q, const float* k, const float* v,
                    const float* o, const float* lse, const float* d_o,
                    float* dq, float* dk, float* dv, void* workspace, size_t workspace_bytes, void* stream);

/* ---- general attention core: q_len != k_len and an attend-mask (ABI >= 7) -----------------------------------------
 * The rest of FlashAttentionFunction's domain (utils.py:134-295): independent query / key lengths (cross-attention through
 * FlashAttention(context=...), attention.py:59-109; utils.py:150-160), a boolean mask broadcastable to [b, h, q_len, k_len]
 * (a [b, k_len] key-padding mask is [b, 1, 1, k_len], utils.py:156-157; nonzero = attend).  `causal` (key j > query i dead)
 * needs k_len <= q_len: with k_len > q_len utils.py:169 shifts the diagonal so that the first k_len - q_len queries see no
 * key and the reference's answer depends on its bucket sizes -- KANVIT_EINVAL here.  d as for kanvit_attn_fwd with
 * d->N = q_len (q, o, do, dq: [B][H][q_len][D] through the q / o strides; k, v, dk, dv: [B][H][Nk][D] through the k / v
 * strides; lse [B][H][q_len]).  Exact fp32 (KANVIT_FLAG_BF16_MFMA is refused).  A query whose keys are all dead gets o = 0,
 * lse = -FLT_MAX, zero gradients.  Any q_len / k_len: the swept operand is walked in LDS chunks of 128 rows (running max / sum in
 * the forward, utils.py:199-221), so these entry points also serve self-attention heads too long for kanvit_attn_fwd's
 * one-head-per-work-group form (N > 224 at D = 64). */
typedef struct kanvit_attn_ext {
    int32_t Nk;              /* key / value length */
    int32_t reserved;
    const void* mask;        /* NULL, or bytes (torch.bool): element (b, h, i, j) at mask[b*sb + h*sh + i*sq + j*sk] */
    int64_t mask_stride_b, mask_stride_h, mask_stride_q, mask_stride_k;      /* in bytes = elements; 0 broadcasts */
} kanvit_attn_ext;
int kanvit_attn_x_fwd(const kanvit_attn_desc* d, const kanvit_attn_ext* e, const float* q, const float* k, const float* v,
                      float* o, float* lse, void* stream);
size_t kanvit_attn_x_bwd_workspace(const kanvit_attn_desc* d, const kanvit_attn_ext* e);
int kanvit_attn_x_bwd(const kanvit_attn_desc* d, const kanvit_attn_ext* e, const float* q, const float* k, const float* v,
                      const float* o, const float* lse, const float* d_o, float* dq, float* dk, float* dv, void* workspace,
                      size_t workspace_bytes, void* stream);

/* ---- fused feed-forward for the small geometries ---------------------------------------------
 * y = relu(x W1^T + b1) W2^T + b2, the TransformerBlock's nn.Sequential(Linear, ReLU(inplace), Linear) (model.py:25-29,36),
 * x[M][D], W1[F][D], b1[F], W2[D][F], b2[D] in nn.Linear's own layouts; fp32 products and sums.  Forward: one launch;
 * backward (recomputes the hidden activations): dx[M][D], dW1[F][D], db1[F], dW2[D][F], db2[D], deterministic.
 * Instantiated for the reference's default width only (D = 64, F = 256) and M <= kanvit_ff_small_max_rows().            */
int kanvit_ff_small_supported(int D, int F);
int64_t kanvit_ff_small_max_rows(void);
int kanvit_ff_small_fwd(int64_t M, int D, int F, const float* x, const float* w1, const float* b1, const float* w2, const float* b2,
                        float* y, void* stream);
size_t kanvit_ff_small_bwd_workspace(int64_t M, int D, int F);
int kanvit_ff_small_bwd(int64_t M, int D, int F, const float* x, const float* w1, const float* b1, const float* w2, const float* dy,
                        float* dx, float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, void* stream);
/* The same with the block's second LayerNorm fused in front (model.py:36  x = x + FF(LN2(x)),  x = x_in + delta):
 *   fwd: s = x + delta (delta may be NULL), y = FF(LayerNorm(s; gamma, beta, eps)); writes s[M][D], mean[M], rstd[M], y[M][D]
 *   bwd: ds = ds_in (may be NULL) + LayerNorm-backward(FF-backward(dy))  -- the gradient of x and of delta alike --
 *        dgamma[D], dbeta[D], dW1, db1, dW2, db2.  Workspace: kanvit_ff_small_bwd_workspace.                         */
int kanvit_lnff_small_fwd(int64_t M, int D, int F, float eps, const float* x, const float* delta, const float* gamma, const float* beta,
                          const float* w1, const float* b1, const float* w2, const float* b2, float* s, float* mean, float* rstd, float* y,
                          void* stream);
int kanvit_lnff_small_bwd(int64_t M, int D, int F, const float* s, const float* mean, const float* rstd, const float* gamma, const float* beta,
                          const float* w1, const float* b1, const float* w2, const float* dy, const float* ds_in, float* ds, float* dgamma,
                          float* dbeta, float* dw1, float* db1, float* dw2, float* db2, void* workspace, size_t workspace_bytes, void* stream);

/* ---- residual add + LayerNorm (the TransformerBlock assembly around the KAN / attention kernels) -------------------
 * Replaces the `x + ...` adds and the nn.LayerNorm calls of model.py:31-37 (TransformerBlock.forward) and their autograd
 * backward, two to four separate passes over [M, D] each in the reference, by one pass forward and one backward.
 *   forward : s = x + delta (delta may be NULL);  y = (s - mean) * rstd * gamma + beta  (biased variance, eps inside the
 *             square root: torch.nn.LayerNorm).  Writes y[M, D], mean[M], rstd[M] and, if xsum != NULL, s to xsum[M, D].
 *   backward: dx = LayerNorm backward of dy w.r.t. s, plus dres[M, D] if not NULL (the gradient that arrives on the
 *             residual stream); dgamma[D], dbeta[D] are fully written.  `xsum` is the s of the forward.
 * Rows are contiguous (row stride D), D % 4 == 0, 4 <= D <= 1024, pointers 16-byte aligned.                        */
int kanvit_addln_fwd(int64_t M, int D, float eps, const float* x, const float* delta, const float* gamma, const float* beta,
                     float* xsum, float* y, float* mean, float* rstd, void* stream);
size_t kanvit_addln_bwd_workspace(int64_t M, int D);
int kanvit_addln_bwd(int64_t M, int D, const float* xsum, const float* gamma, const float* mean, const float* rstd,
                     const float* dy, const float* dres, float* dx, float* dgamma, float* dbeta, void* workspace,
                     size_t workspace_bytes, void* stream);
/* The same two passes with bf16 tensors at the boundary (torch.autocast: the feed-forward's output `delta`, its input `y` and the
 * gradient `dy` arriving from it are bf16; the residual stream x / xsum / dres / dx stays fp32): *_bf16 != 0 says that the pointer
 * in front of it is bf16 [M][D]; dx_bf16, if not NULL, receives a second copy of dx rounded to bf16 (the gradient of a bf16
 * delta).  With all flags 0 and dx_bf16 NULL these ARE kanvit_addln_fwd / _bwd.  (ABI >= 6) */
int kanvit_addln_fwd_ex(int64_t M, int D, float eps, const float* x, const void* delta, int delta_bf16, const float* gamma, const float* beta,
                        float* xsum, void* y, int y_bf16, float* mean, float* rstd, void* stream);
int kanvit_addln_bwd_ex(int64_t M, int D, const float* xsum, const float* gamma, const float* mean, const float* rstd,
                        const void* dy, int dy_bf16, const float* dres, float* dx, void* dx_bf16, float* dgamma, float* dbeta, void* workspace,
                        size_t workspace_bytes, void* stream);

/* ---- ReLU backward + bias gradient of the feed-forward's first Linear in one pass ----------------
 * The TransformerBlock's nn.Sequential(Linear, ReLU(inplace), Linear) (model.py:25-29): what autograd runs for the ReLU and
 * for the first Linear's bias -- threshold_backward on the saved activation y, then a column sum of the result --
 *     dh[m][n] = y[m][n] > 0 ? dy[m][n] : 0,      dbias[n] = sum_m dh[m][n]
 * as one pass over [M, N] + an ordered reduce of per-row-band partial sums (deterministic).  dh may alias dy.
 * Row-major contiguous [M][N] fp32, N % 4 == 0, 16-byte aligned pointers; workspace: kanvit_relu_bwd_bias_workspace(M, N). */
size_t kanvit_relu_bwd_bias_workspace(int64_t M, int N);
int kanvit_relu_bwd_bias(int64_t M, int N, const float* dy, const float* y, float* dh, float* dbias, void* workspace,
                         size_t workspace_bytes, void* stream);
/* The same on bf16 tensors (torch.autocast): dy, y, dh are bf16 [M][N] (16-byte aligned, N % 4 == 0); dbias and the workspace stay
 * fp32 (sizes as above).  (ABI >= 6) */
int kanvit_relu_bwd_bias_bf16(int64_t M, int N, const void* dy_bf16, const void* y_bf16, void* dh_bf16, float* dbias, void* workspace,
                              size_t workspace_bytes, void* stream);

/* ---- three-term bf16 split image (opt-in "bf16x3" feed-forward mode, kanvit/dense.py) --------------------------------
 * v = hi + lo, hi = bf16(v), lo = bf16(v - hi).  Writes out[M][3K] (bf16) = [hi|hi|lo] (pattern 0) or [hi|lo|hi]
 * (pattern 1) of  v = x (+ bias[K]) (ReLU if relu) (zeroed where mask_hi[m][k] <= 0, a bf16 image with row stride
 * mask_ld elements): an fp32-accurate (~5e-6) product through ONE bf16 GEMM over a 3K-long axis, pattern-0 image times
 * pattern-1 image.  Replaces nothing in the reference (model.py:25-29 runs fp32 nn.Linear); K % 8 == 0.           */
int kanvit_split3_bf16(int64_t M, int K, const float* x, const float* bias, int relu, const void* mask_hi, int64_t mask_ld,
                       void* out, int pattern, void* stream);

/* ---- per-family named entry points (SURVEY.md section 8b naming) ----------------------------
 * kanvit_<family>_{fwd,bwd_input,bwd_weight} are kanvit_layer_* with d->family checked;
 * kanvit_<family>_qkv_* additionally require groups == 3 * x_group_mod (one launch for all
 * heads' q, k and v mappings).                                                                 */
#define KANVIT_DECLARE_FAMILY(name)                                                                           \
    int kanvit_##name##_fwd(const kanvit_layer_desc*, const float*, const float*, const float*, const float*, \
                            const float*, float*, void*, size_t, void*);                                      \
    int kanvit_##name##_bwd_input(const kanvit_layer_desc*, const float*, const float*, const float*,         \
                                  const float*, const float*, float*, float*, float*, void*, size_t, void*);  \
    int kanvit_##name##_bwd_weight(const kanvit_layer_desc*, const float*, const float*, const float*,        \
                                   const float*, float*, void*, size_t, void*);                               \
    int kanvit_##name##_qkv_fwd(const kanvit_layer_desc*, const float*, const float*, const float*,           \
                                const float*, const float*, float*, void*, size_t, void*);                    \
    int kanvit_##name##_qkv_bwd_input(const kanvit_layer_desc*, const float*, const float*, const float*,     \
                                      const float*, const float*, float*, float*, float*, void*, size_t,      \
                                      void*);                                                                 \
    int kanvit_##name##_qkv_bwd_weight(const kanvit_layer_desc*, const float*, const float*, const float*,    \
                                       const float*, float*, void*, size_t, void*);
KANVIT_DECLARE_FAMILY(linear)
KANVIT_DECLARE_FAMILY(cheby)
KANVIT_DECLARE_FAMILY(bspline)
KANVIT_DECLARE_FAMILY(rbf)
KANVIT_DECLARE_FAMILY(sine)
KANVIT_DECLARE_FAMILY(fourier)

/* ---- misc ---------------------------------------------------------------------------------- */
int kanvit_abi_version(void);
const char* kanvit_last_error(void);
/* Kernel-selection switches (KANVIT_NO_REG, KANVIT_NO_BF16, KANVIT_ATTN_V1, ...: fallback kernels for the parity tests and two
 * tuning knobs).  The environment is read ONCE, on first use of the library -- never on the launch path; kanvit_config()
 * returns the active set as "name=value ..." (all zero = the default kernels) so a measurement can say what it ran, and
 * kanvit_config_reload() re-reads the environment (test hook; not for use while launches are in flight).            */
const char* kanvit_config(void);
int kanvit_config_reload(void);
/* number of HIP devices visible, or a negative error code; does not initialise a context */
int kanvit_device_count(void);

#if defined(__GNUC__) || defined(__clang__)
#pragma GCC visibility pop
#endif

#ifdef __cplusplus
}
#endif
#endif /* KANVIT_H */
