/*
 * ldm_amd.h -- C-ABI of the MI355X (gfx950) latent-diffusion hot path.
 *
 * One shared library (latent-diffusion_amd/lib/libldm_amd.so), plain pointers and sizes only,
 * no torch types.  Every device entry point launches asynchronously on the HIP stream it is given
 * (a hipStream_t passed as void*; NULL = the default stream), never synchronises the host and never
 * allocates device memory.  All entry points return 0 on success or a negative LDM_E* code; the text
 * of the last failure on the calling thread is available from ldm_last_error().
 *
 * The reference (anenbergb/latent-diffusion, /root/reference) is pure Python: its "FFI" for this path
 * is the set of torch ops its two hot files call.  Each declaration cites the reference lines whose
 * arithmetic it replaces.  Layout conventions:
 *   - activations are NHWC ("pixels x channels" row-major matrices, the transformer's (B, HW, C) view
 *     is the same buffer), dtype LDM_BF16 or LDM_F32;
 *   - filters are K-contiguous: [Cout][kh][kw][Cin] (see ldm_pack_weight);
 *   - statistics, biases, norm affine parameters and scheduler tables are float32.
 */
#ifndef LDM_AMD_H
#define LDM_AMD_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDM_ABI_VERSION 3   /* 2: struct LdmConvGemm gained the LayerNorm-fold fields at its end; 3: glu_preact behind them */

enum { LDM_F32 = 0, LDM_BF16 = 1 };

enum {
  LDM_OK = 0,
  LDM_EINVAL = -1,   /* bad argument (shape / dtype / alignment / null pointer) */
  LDM_EUNSUPPORTED = -2,
  LDM_EHIP = -3      /* a HIP runtime call or kernel launch failed */
};

int ldm_abi_version(void);
const char* ldm_last_error(void);

/* ---------------------------------------------------------------- DDPM scheduler: host-side tables */

enum { LDM_SCHED_LINEAR = 0, LDM_SCHED_COSINE = 1, LDM_SCHED_SIGMOID = 2 };
enum { LDM_SPACING_LEADING = 0, LDM_SPACING_LINSPACE = 1, LDM_SPACING_TRAILING = 2 };

/* Seven float32 tables of T entries each, written to HOST buffers.
 * Replaces DDPMScheduler.__init__ (ldm/scheduling_ddpm.py:54-75) and get_beta_schedule (:308-357):
 * float64 betas -> float32, then sequential float32 cumprod.  Bit-exact with the reference on CPU. */
int ldm_ddpm_build_tables(int schedule, int T, float* betas, float* alphas_cumprod,
                          float* sqrt_alphas_cumprod, float* sqrt_one_minus_alphas_cumprod,
                          float* posterior_mean_coef1, float* posterior_mean_coef2, float* posterior_std);

/* Inference timestep list (int64, n entries, HOST buffer).
 * Replaces DDPMScheduler.set_timesteps (scheduling_ddpm.py:131-180).  LDM_EINVAL if n > T or n < 1. */
int ldm_ddpm_set_timesteps(int spacing, int T, int n, int steps_offset, int64_t* timesteps);

/* The five float32 scalars of one reverse step t -> s (s == -1: alpha_bar_s := 1):
 * coef = {sqrt(1-acp_t) [table], sqrt(acp_t) [table], c1, c2, std}.
 * Replaces the (B,1,1,1) coefficient arithmetic of predict_start_from_noise (:112-114) and
 * q_posterior_between (:244-255) + _extract_alpha_bar (:270-282).  Tables are HOST pointers. */
int ldm_ddpm_step_coefficients(const float* alphas_cumprod, const float* sqrt_alphas_cumprod,
                               const float* sqrt_one_minus_alphas_cumprod, int T, int t, int s,
                               float coef[5]);

/* ------------------------------------------------------------- DDPM scheduler: device elementwise */

/* prev = c1*x0 + c2*x_t + std*noise with x0 = (x_t - sigma*eps)/sqrt_acp; one coalesced pass.
 * eps and x_t may be LDM_F32 or LDM_BF16 (promoted to float32), noise/prev/x0 are float32
 * (scheduling_ddpm.py:207-214; results are float32 whatever the inputs are).  x0 may be NULL.
 * n = number of elements (any n >= 0).  Replaces DDPMScheduler.step (:185-215). */
int ldm_ddpm_step(const void* eps, int eps_dtype, const void* x_t, int x_dtype, const float* noise,
                  float* prev, float* x0, int64_t n, const float coef[5], void* stream);

/* Same update with classifier-free guidance folded in: eps = eps_u + g*(eps_c - eps_u)
 * (train_latent_diffusion.py:589-593), then the step above.  eps_u / eps_c share eps_dtype. */
int ldm_ddpm_step_cfg(const void* eps_u, const void* eps_c, int eps_dtype, float guidance,
                      const void* x_t, int x_dtype, const float* noise, float* prev, float* x0,
                      int64_t n, const float coef[5], void* stream);

/* x_t = sqrt_acp[t_b]*x0 + sqrt(1-acp)[t_b]*noise, per-sample int64 t (DEVICE), tables DEVICE float32.
 * x0/noise share `dtype`; out is float32 (the float32 (B,1,1,1) coefficients promote the result).
 * batch samples of per_sample elements each.  A timestep outside [0, T) -- a device assert in the reference's gather --
 * yields NaN for that whole sample (the table read itself is clamped, so it cannot fault).
 * Replaces DDPMScheduler.add_noise + extract (scheduling_ddpm.py:81-101, :285-305). */
int ldm_ddpm_add_noise(const void* x0, const void* noise, float* out, int dtype, const int64_t* t,
                       const float* sqrt_alphas_cumprod, const float* sqrt_one_minus_alphas_cumprod,
                       int T, int batch, int64_t per_sample, void* stream);

/* Training-side input preparation in one pass (train_latent_diffusion.py:1039-1064), float32:
 *   latents = scaling * (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps)   moments = (mean | logvar): [batch][2*channels][hw]
 *                                                                           (vae._encode output, :1036; diffusers
 *                                                                           DiagonalGaussianDistribution.sample, :1039-1040)
 *   noise  += noise_offset * offset[b][c]        if offset != NULL (:1043-1048; written back: it is the loss target)
 *   noisy   = sqrt_acp[t_b]*latents + sqrt(1-acp)[t_b]*(noise + perturbation*extra)   (extra may be NULL; :1049-1064)
 * eps, noise, extra, latents, noisy: [batch][channels][hw]; all random draws are the caller's.  t, sa, so as ldm_ddpm_add_noise. */
int ldm_train_input_prep(const float* moments, const float* eps, float* noise, const float* offset, const float* extra,
                         const int64_t* t, const float* sqrt_alphas_cumprod, const float* sqrt_one_minus_alphas_cumprod, int T,
                         int batch, int channels, int64_t hw, float scaling, float noise_offset, float perturbation,
                         float* latents, float* noisy, void* stream);

/* enc[b] = drop[b] ? null_emb : enc[b], in place: text-conditioning dropout (:1073-1079).  enc [batch][per_sample], null_emb
 * [per_sample] in dtype; drop: DEVICE bytes. */
int ldm_cond_dropout(void* enc, const void* null_emb, const unsigned char* drop, int dtype, int batch, int64_t per_sample,
                     void* stream);

/* N(0,1) noise for rows [row0, row0 + rows) of a batch whose rows have row_elems elements, float32, as a function of
 * (seed, step, GLOBAL row index, element) only: Philox4x32-10 keyed by seed, counter = (element / 4, row, step + 1), Box-Muller
 * on 23-bit uniforms.  Replaces `torch.randn_like(mean)` of DDPMScheduler.step (scheduling_ddpm.py:212, global RNG) and the
 * initial `randn` of generate (train_latent_diffusion.py:566-576) in the batch-sharded sampler (ldm/sampling.py): a rank draws
 * only its own rows and the values of a row do not depend on the number of ranks.  step = -1: the initial latents.
 * The reference's own noise stream (torch's global generator) is NOT reproduced; parity tests inject identical noise tensors. */
int ldm_gaussian_noise(uint64_t seed, int64_t step, int64_t row0, int64_t rows, int64_t row_elems, float* out, void* stream);

/* ------------------------------------------------------------------------- UNet: weight packing */

/* dst[o'][kh][kw][ci] = src[o][ci][kh][kw] (src float32 OIHW / [out,in]; dst dtype LDM_BF16 or LDM_F32).
 * glu_pairs != 0 interleaves the two halves of a GEGLU/SwiGLU projection in blocks of 16 rows
 * (o' = 32*(o/16) + o%16 for value rows, +16 for the matching gate rows) so that value and gate of one
 * output column land in the same lane of the MFMA epilogue (unet.py:787-790: value first half). */
int ldm_pack_weight(const float* src, void* dst, int dst_dtype, int cout, int cin, int kh, int kw,
                    int glu_pairs, void* stream);

/* Both derived layouts of one filter in one pass: fwd = ldm_pack_weight's [o'][kh][kw][ci] (may be NULL) and dgrad =
 * [ci][kh][kw][o'] with the taps rotated by 180 degrees and row pitch dgrad_ld >= cout (may be NULL): the filter with which
 * ldm_conv_gemm on dY computes the input gradient of the layer (autograd of nn.Conv2d / nn.Linear under loss.backward(),
 * train_latent_diffusion.py:1119).  Filters fused along the output dimension (q|k|v, the 22 time projections) are packed part
 * by part: fwd at its row offset, dgrad at its column offset of one matrix of pitch dgrad_ld.  kh*kw <= 9. */
int ldm_pack_weight_pair(const float* src, void* fwd, void* dgrad, int dst_dtype, int cout, int cin, int kh, int kw,
                         int glu_pairs, long long dgrad_ld, void* stream);

/* ldm_pack_weight_pair for MANY filters in one launch: the repack of every filter of the model that follows each optimizer step
 * of the reference's train loop (train_latent_diffusion.py:1123-1126; autocast's per-step weight casts in the reference).  `jobs`
 * is a DEVICE array of njobs descriptors, all bfloat16 destinations with the same tap count (1 or 9), cout, cin and dgrad_ld
 * multiples of 4, 8-byte aligned destinations; first_block ascending from 0 with first_block[i+1] = first_block[i] +
 * blocks_x[i] * blocks_y[i] (blocks_x = ceil(cout / 32), blocks_y = ceil(cin / 32)); total_blocks = their sum.  The caller
 * validates the descriptors (ldm/ops.py pack_weight_pair_batch does); the kernel trusts them. */
typedef struct LdmPackJob {
  const float* src;   /* float32 [cout][cin][taps] */
  void* fwd;          /* may be NULL */
  void* dgrad;        /* may be NULL */
  int64_t dgrad_ld;
  int32_t cout, cin, glu_pairs;
  int32_t blocks_x, blocks_y, first_block;
} LdmPackJob;
int ldm_pack_weight_pair_batch(const LdmPackJob* jobs_device, int njobs, int total_blocks, int taps, void* stream);

/* ------------------------------------------------------------------- UNet: implicit-GEMM conv/linear */

enum {
  LDM_EPI_BIAS = 1,        /* + bias[n] */
  LDM_EPI_ROWVEC = 2,      /* + rowvec[sample(m)][n]  (time-embedding add, unet.py:727-728) */
  LDM_EPI_RESIDUAL = 4,    /* + residual[m][n]        (unet.py:731, :593-594) */
  LDM_EPI_SILU = 8,        /* y = y*sigmoid(y)        (time_embedding SiLU, unet.py:31, :703) */
  LDM_EPI_GEGLU = 16,      /* out[m][j] = v*gelu_erf(g), paired columns (unet.py:787-790) */
  LDM_EPI_SWIGLU = 32,     /* out[m][j] = v*silu(g) (unet.py:765-768) */
  /* nn.LayerNorm (unet.py:550-551, :576) folded into the two GEMMs around it instead of a pass of its own (bf16 only):
   *   producer (the GEMM that writes x):  LDM_EPI_ROWSTATS also writes, per row and 80-column segment, the sum and the sum of
   *     squares of the outputs as stored -> rowstats[m][n / 80][2] (float32; n % 160 == 0; deterministic);
   *   consumer (the GEMM that reads LayerNorm(x)): LDM_EPI_LNFOLD runs on x itself with the filter pre-scaled by gamma,
   *     W'[n][k] = gamma[k] W[n][k], and finishes in the epilogue: y = rstd_m (x W'^T - mean_m * ln_colsum[n]) + bias[n], where
   *     ln_colsum[n] = sum_k W'[n][k] (of the rounded W') and `bias` already holds bias + W beta.  mean / rstd come from the
   *     producer's rowstats (ln_stats, ln_segments = producer's n / 80) with K = c0 columns and eps = ln_eps. */
  LDM_EPI_ROWSTATS = 64,
  LDM_EPI_LNFOLD = 128
};

typedef struct LdmConvGemm {
  /* A operand: one or two NHWC sources, virtually concatenated along channels (torch.cat at
   * unet.py:313/:377 is never materialised for the 1x1 shortcut).  c1 == 0: single source. */
  const void* a0;
  const void* a1;
  int32_t c0, c1;
  int32_t batch, hin, win;     /* stored spatial extent of the sources (1,1 for plain matrices) */
  int32_t hout, wout;          /* output spatial extent */
  int32_t ksize;               /* 1 or 3 (pad = ksize/2) */
  int32_t stride;              /* 1 or 2 (unet.py:218 downsample) */
  int32_t upsample;            /* 1: nearest-2x before the conv (unet.py:316-317, :385-386);
                                * 2: zero-insertion 2x (only even positions hold data): with the 180-degree-rotated,
                                *    in/out-transposed filter this is the input gradient of the stride-2 conv (:218) */
  /* B operand: packed filter [n][ksize*ksize*(c0+c1)] */
  const void* w;
  int32_t n;
  /* epilogue */
  int32_t epilogue;            /* LDM_EPI_* flags */
  const float* bias;           /* [n] float32 */
  const void* rowvec;          /* [batch][rowvec_stride] in dtype */
  int32_t rowvec_stride;
  const void* residual;        /* [m][n] in dtype */
  void* out;                   /* [m][n] (or [m][n/2] with GEGLU/SWIGLU) */
  int32_t dtype;               /* LDM_BF16 (MFMA bf16, fp32 accumulate) or LDM_F32 (exact fp32 MFMA) */
  int32_t out_dtype;           /* dtype of out; normally == dtype */
  /* Optional DEVICE scratch the call may use for split-K partial sums (small-M layers that cannot fill the chip
   * otherwise: the 8x8 level of the UNet).  NULL / 0: never split.  Contents are undefined afterwards; calls
   * sharing one workspace must be ordered on one stream.  The result is deterministic either way. */
  void* workspace;
  int64_t workspace_bytes;
  /* LayerNorm fold (ABI version 2; zero / NULL when unused) */
  float* rowstats;             /* LDM_EPI_ROWSTATS: out, [m][n / 80][2] float32 */
  const float* ln_stats;       /* LDM_EPI_LNFOLD: the rowstats the producer of the A operand wrote */
  const float* ln_colsum;      /* LDM_EPI_LNFOLD: [n] float32 */
  int32_t ln_segments;         /* LDM_EPI_LNFOLD: segments per row of ln_stats (= c0 / 80) */
  float ln_eps;                /* LDM_EPI_LNFOLD: LayerNorm eps (1e-5 in the reference) */
  /* Training forward of the GLU feed-forward (ABI version 3; NULL when unused): with LDM_EPI_GEGLU / LDM_EPI_SWIGLU and bf16
   * output, also store the biased pre-activation, [m][n] bf16 in the filter's paired column order (16 values, their 16
   * gates, ...) -- the tensor autograd keeps for the backward of unet.py:787-790 -- so the forward needs no second pass. */
  void* glu_preact;
} LdmConvGemm;

/* Dropout (unet.py:636-639, :707, :579: nn.Dropout(p) after the attention output projection, inside ResnetBlock.layers and the
 * feed-forward) has no kernel here: the reference's trainer and every default leave p = 0.0, where it is the identity; the
 * Python drop-in accepts dropout > 0 and runs such a model in eval mode (identity there, as in the reference); a training-mode
 * call, where the reference draws masks, raises NotImplementedError instead of silently ignoring them.
 *
 * out = epilogue(im2col(A) @ W^T).  M = batch*hout*wout rows.  Replaces nn.Conv2d 3x3/1x1 and nn.Linear
 * as called from ResnetBlock (unet.py:698-731), TransformerModel (:508,:521), Attention (:655-656,:683),
 * GEGLU (:787-790), feed_forward (:579), down/upsample (:218,:317), time_embedding (:29-33). */
int ldm_conv_gemm(const LdmConvGemm* desc, void* stream);

/* The (BM, BN, BK) tile ldm_conv_gemm would pick for this descriptor (host-only query; used by bench.py to
 * attribute measured launch times to kernel instantiations).  No counterpart in the reference. */
int ldm_conv_gemm_tile(const LdmConvGemm* desc, int tile[3]);

/* Weight gradient of the same layer: dw[n][ci][kh][kw] (+)= sum_m dy[m][n] * im2col(A)[m][(kh,kw,ci)], float32 in the
 * checkpoint layout (OIHW / [out,in]) -- what loss.backward() (train_latent_diffusion.py:1117) leaves in the .grad of
 * every nn.Conv2d / nn.Linear weight of ldm/unet.py.  `desc` describes the forward call (A operand, geometry, n, dtype;
 * w / epilogue / out are ignored); dy is [m][n] in desc->dtype.  accumulate != 0 adds to dw (gradient accumulation,
 * :1032).  `workspace`: DEVICE scratch of at least ldm_conv_wgrad_workspace_bytes(desc) bytes (fp32 partial tiles of
 * the pixel splits, folded in a fixed order: deterministic).  bf16 with channel counts in multiples of 8 runs on the
 * matrix cores; float32 (the exact parity path) and other channel counts run a plain-FMA kernel. */
long long ldm_conv_wgrad_workspace_bytes(const LdmConvGemm* desc);
int ldm_conv_wgrad(const LdmConvGemm* desc, const void* dy, float* dw, int accumulate, void* workspace,
                   long long workspace_bytes, void* stream);

/* --------------------------------------------------------------------------- UNet: normalisation */

/* Per-(sample, group) mean and rstd over an NHWC tensor (two-source virtual concat allowed).
 * stats[(b*groups+g)*2 + {0,1}] = {mean, rstd}.  eps 1e-5 in the reference everywhere.
 * Two deterministic stages (per-slab partial sums, then a fold in double); `workspace` is caller-provided
 * DEVICE scratch of at least ldm_groupnorm_workspace_floats(batch, groups) floats.
 * First half of nn.GroupNorm (unet.py:699, :705, :480, :113). */
int ldm_groupnorm_workspace_floats(int batch, int groups);
int ldm_groupnorm_stats(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw,
                        int groups, float eps, float* stats, float* workspace, void* stream);

/* out[b][p][c] = act((x-mean)*rstd*gamma[c]+beta[c]); act = SiLU if silu != 0.  Writes the (materialised)
 * concatenation when two sources are given.  Second half of nn.GroupNorm (+ nn.SiLU, unet.py:700,:706,:114). */
int ldm_groupnorm_apply(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw,
                        int groups, const float* stats, const float* gamma, const float* beta, int silu,
                        void* out, void* stream);

/* nn.GroupNorm (+ nn.SiLU) in one call (unet.py:699-700, :705-706, :480, :113-114): statistics, normalisation and the
 * (B, groups, 2) mean/rstd tensor the backward reads.  Feature maps whose (sample, group set) slice fits one workgroup
 * (bf16, the 32x32 / 16x16 / 8x8 levels) run as ONE kernel: one read and one write per element; larger ones run
 * ldm_groupnorm_stats + ldm_groupnorm_apply.  `workspace` as for ldm_groupnorm_stats. */
int ldm_groupnorm(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw, int groups, float eps,
                  const float* gamma, const float* beta, int silu, void* out, float* stats, float* workspace, void* stream);

/* ldm_groupnorm for callers that cannot allocate or lend device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc).  NO COUNTERPART IN THE REFERENCE.  Same arguments and bit-identical results as
 * ldm_groupnorm, minus the workspace: the partial sums of the three-kernel form go to a static device array inside the
 * library (LDM_GROUPNORM_NOALLOC_MAX_FLOATS floats = 4 MiB, held by every GPU the library runs on until the process ends).
 * The call makes no hipMalloc / hipFree and calls no allocating routine.  LDM_EINVAL if batch * groups exceeds
 * LDM_GROUPNORM_NOALLOC_MAX_FLOATS / 128 (= 8192; e.g. 256 samples x 32 groups).  All calls on one GPU share that GPU's
 * array: calls that overlap on one GPU (two streams) race; calls ordered on one stream do not.  Nothing in the array has to be
 * zero on entry (every word read is written earlier in the same call).  Graph capture: the FIRST call on a device looks the
 * array's address up (hipGetSymbolAddress, not a stream operation) and is refused with LDM_EINVAL while `stream` is capturing --
 * make one ordinary call per device before capturing; every later call is capturable. */
#define LDM_GROUPNORM_NOALLOC_MAX_FLOATS (1 << 20)
int ldm_groupnorm_noalloc(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw, int groups, float eps,
                          const float* gamma, const float* beta, int silu, void* out, float* stats, void* stream);

/* Row-wise LayerNorm over the last dimension (unet.py:550-551, :576), eps 1e-5. */
int ldm_layernorm(const void* x, int dtype, int64_t rows, int cols, float eps, const float* gamma,
                  const float* beta, void* out, void* stream);

/* ------------------------------------------------------------------ UNet: backward (HBM-bound kernels)
 * What loss.backward() (train_latent_diffusion.py:1117) computes for the non-GEMM layers of ldm/unet.py.  All
 * deterministic (fixed-order reductions); workspaces are caller-provided DEVICE scratch. */

/* out[b][h][w][c] = sum of the 2x2 block of x at (2h, 2w) (+ residual): gradient of F.interpolate(scale_factor=2,
 * mode="nearest") (unet.py:316, :385).  x: (batch, 2*hout, 2*wout, c) NHWC. */
int ldm_sum2x2(const void* x, const void* residual, void* out, int dtype, int batch, int hout, int wout, int c, void* stream);

/* out[s][n] (+)= sum_r x[s*rows_per_segment + r][n], float32: bias gradients (segments = 1 or summed by the caller) and
 * the gradient of the per-sample time-embedding add (unet.py:727-728; segments = batch). */
long long ldm_colsum_workspace_floats(long long segments, int n);
int ldm_colsum(const void* x, int dtype, long long segments, long long rows_per_segment, int n, float* out, int accumulate,
               float* workspace, void* stream);

/* GEGLU / SwiGLU in training form (unet.py:787-790 / :765-768): h = packed-order pre-activation [rows][2*ff] (every 32
 * columns = 16 values then their 16 gates, as ldm_pack_weight(glu_pairs) orders the projection); out[rows][ff] =
 * value * act(gate); backward: dh from dout.  gelu != 0: exact GELU, else SiLU. */
int ldm_glu_forward(const void* h, void* out, int dtype, long long rows, int ff, int gelu, void* stream);
int ldm_glu_backward(const void* h, const void* dout, void* dh, int dtype, long long rows, int ff, int gelu, void* stream);

/* out[r][c] = a[r*lda + c] + b[r*ldb + c], out dense [rows][cols]: where two gradient paths meet (skip connections,
 * unet.py:313/:377, :731) and neither producer can fuse the add.  Row strides in elements, so an operand may be the
 * channel slice of a concat gradient.  cols, lda, ldb multiples of 8. */
int ldm_add(const void* a, long long lda, const void* b, long long ldb, void* out, int dtype, long long rows, int cols,
            void* stream);

/* dx = dy * d/dx silu(x)  (time-embedding MLP, unet.py:31, :703). */
int ldm_silu_backward(const void* x, const void* dy, void* dx, int dtype, long long n, void* stream);

/* Backward of y = [SiLU](GroupNorm(x)) (unet.py:699-700, :705-706, :480, :113-114) on NHWC (two-source concat allowed):
 * dx (+ add, an optional second gradient arriving at x) for all c0+c1 channels, dgamma / dbeta (float32, (+)= when
 * accumulate).  stats: mean/rstd from ldm_groupnorm_stats.  dy, add, dx: [batch*hw][c0+c1] in dtype.  dx1 != NULL (two
 * sources only): the gradient of the concat (unet.py:313/:377) is split where it is produced -- dx [batch*hw][c0] for
 * the hidden state, dx1 [batch*hw][c1] for the skip connection. */
long long ldm_groupnorm_backward_workspace_floats(int batch, int channels, int groups);
int ldm_groupnorm_backward(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw, int groups,
                           const float* stats, const float* gamma, const float* beta, int silu, const void* dy,
                           const void* add, void* dx, void* dx1, float* dgamma, float* dbeta, int accumulate,
                           float* workspace, void* stream);

/* Backward of nn.LayerNorm (unet.py:550-551, :576): statistics are recomputed from x. */
long long ldm_layernorm_backward_workspace_floats(int cols);
int ldm_layernorm_backward(const void* x, const void* dy, int dtype, long long rows, int cols, float eps, const float* gamma,
                           const void* add, void* dx, float* dgamma, float* dbeta, int accumulate, float* workspace,
                           void* stream);

/* ------------------------------------------------------- training-step tail around the UNet
 * Tensor lists are DEVICE tables: per tensor its pointers (as 64-bit integers), per chunk -- at most
 * ldm_opt_chunk_elements() consecutive float32 elements of one tensor, one workgroup each -- the tensor index, the element
 * offset inside the tensor and the length. */
int ldm_opt_chunk_elements(void);

/* out[0] = L2 norm over all gradients, out[1] = min(1, max_norm / (norm + 1e-6)) (max_norm <= 0: 1): the reduction of
 * torch.nn.utils.clip_grad_norm_ (train_latent_diffusion.py:1123) without touching the gradients -- ldm_adamw_step applies the
 * scale while it reads them.  partial: nchunks floats of DEVICE scratch.  Deterministic (fixed-order folds). */
int ldm_grad_norm_clip(const unsigned long long* grads, int ntensors, const int* chunk_tensor, const long long* chunk_off,
                       const int* chunk_len, int nchunks, float max_norm, float* partial, float* out, void* stream);

/* One torch.optim.AdamW step (train_latent_diffusion.py:909-915, :1124: decoupled weight decay, bias correction, no amsgrad)
 * over all tensors in one launch: p, exp_avg, exp_avg_sq updated in place from g * clip[1] (clip: DEVICE pointer to the pair
 * written by ldm_grad_norm_clip, or NULL for no clipping).  step counts from 1.  float32 tensors; the hyper-parameters are
 * doubles because torch derives 1 - beta, 1 - lr*wd and the bias corrections from Python floats before rounding. */
int ldm_adamw_step(const unsigned long long* params, const unsigned long long* grads, const unsigned long long* exp_avg,
                   const unsigned long long* exp_avg_sq, int ntensors, const int* chunk_tensor, const long long* chunk_off,
                   const int* chunk_len, int nchunks, double lr, double beta1, double beta2, double eps, double weight_decay,
                   long long step, const float* clip, void* stream);

/* loss[0] = mean((pred - target)^2) in float32 (F.mse_loss(pred.float(), noise.float()), :1108) and, if dpred != NULL, its
 * gradient dpred = 2 (pred - target) / n in pred's dtype, in the same pass.  workspace: ldm_mse_workspace_floats(n) floats. */
long long ldm_mse_workspace_floats(long long n);
int ldm_mse_loss(const void* pred, int dtype, const float* target, long long n, float* loss, void* dpred, float* workspace,
                 void* stream);

/* -------------------------------------------------------------------------------- UNet: attention */

/* out[b][i][h*d + :] = softmax(q_h k_h^T / sqrt(d)) v_h, flash-style, no mask
 * (F.scaled_dot_product_attention at unet.py:678).  q: [batch][lq][q_stride], k/v: [batch][lk][kv_stride]
 * (row strides in elements; head h starts at column h*d), out: [batch][lq][out_stride]. */
int ldm_attention(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out,
                  int out_stride, int dtype, int batch, int heads, int lq, int lk, int d, void* stream);

/* Training form: also writes lse[batch][heads][lq] (float32) = log2 sum_k exp2(log2(e)/sqrt(d) * q.k), which the
 * backward needs to recompute the probabilities. */
int ldm_attention_train(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out,
                        int out_stride, int dtype, int batch, int heads, int lq, int lk, int d, float* lse, void* stream);

/* Backward of F.scaled_dot_product_attention (unet.py:678): dq, dk, dv from q, k, v, the forward output o, its
 * gradient d_o and lse.  Flash-style (the score matrix is recomputed tile by tile, never stored), two deterministic
 * kernels (no atomics).  All tensors (rows, heads*d) with the given row strides, head h in columns [h*d, (h+1)*d);
 * dk and dv share dkv_stride.  delta_workspace: batch*heads*lq floats.  bf16 with head dims 40/64/80/128/160 and
 * 16-byte-aligned operands runs on the matrix cores; float32 (the exact parity path) and other head dims <= 256 run
 * a plain-FMA kernel of the same structure. */
int ldm_attention_backward(const void* q, int q_stride, const void* k, const void* v, int kv_stride, const void* o,
                           int o_stride, const void* d_o, int do_stride, const float* lse, void* dq, int dq_stride,
                           void* dk, void* dv, int dkv_stride, float* delta_workspace, int dtype, int batch, int heads,
                           int lq, int lk, int d, void* stream);

/* ------------------------------------------------------------------------ UNet: first / last layers */

/* Sinusoidal timestep embedding, cos first, divisor dim/2-1 (unet.py:171-173 ->
 * diffusers get_timestep_embedding(flip_sin_to_cos=True, downscale_freq_shift=1)).
 * t: DEVICE float32 [batch] (already expanded); out [batch][dim] in dtype. */
int ldm_timestep_embedding(const float* t, int batch, int dim, void* out, int dtype, void* stream);

/* y = x * sigmoid(x) elementwise (nn.SiLU in front of ResnetBlock.time, unet.py:703).  Small tensors only:
 * every other SiLU of the model is fused into a GroupNorm or GEMM kernel. */
int ldm_silu(const void* x, void* y, int dtype, int64_t n, void* stream);

/* conv_in (unet.py:26,:135): 3x3 pad 1 over an NCHW input with few channels -> NHWC output.
 * x: [batch][cin][h][w] in x_dtype; w: float32 OIHW [cout][cin][3][3]; bias float32. */
int ldm_conv_in(const void* x, int x_dtype, const float* w, const float* bias, void* out, int out_dtype,
                int batch, int cin, int h, int wd, int cout, void* stream);

/* final conv (unet.py:115,:155): 3x3 pad 1 from an NHWC activation (already GroupNorm+SiLU'ed) to an NCHW
 * output with few channels.  w: float32 OIHW [cout][cin][3][3]. */
int ldm_conv_out(const void* x, int x_dtype, const float* w, const float* bias, void* out, int out_dtype,
                 int batch, int cin, int h, int wd, int cout, void* stream);

/* ---- VAE-decoder leg of `generate` (train_latent_diffusion.py:596-601).  The decoder itself is diffusers'
 * AutoencoderKL.decode (third-party, version unpinned in the reference's pyproject.toml:28, absent offline; layer stack in the
 * reference's VAE.md:119-157): everything but the three entry points below is the UNet's GroupNorm / conv-GEMM kernels. */

/* out[r][c] = softmax over c < valid_cols of scale * scores[r][c], zero for valid_cols <= c < cols.  The decoder's
 * mid-block attention (VAE.md:132-137) is one head of 512 channels: its scores are materialised by ldm_conv_gemm (float32
 * out), normalised here, and multiplied with V by a second ldm_conv_gemm.  scores float32 [rows][ld_in]; out in out_dtype
 * [rows][ld_out]; scale > 0. */
int ldm_softmax_rows(const float* scores, long long ld_in, long long rows, int cols, int valid_cols, float scale, void* out,
                     int out_dtype, long long ld_out, void* stream);

/* post_quant_conv (VAE.md:125): 1x1 conv over an NCHW tensor of at most 8 channels, float32 NCHW out.
 * x: [batch][cin][hw] in x_dtype; w: float32 [cout][cin]; bias float32 [cout]. */
int ldm_conv1x1_nchw(const void* x, int x_dtype, const float* w, const float* bias, float* out, int batch, int cin, int cout,
                     long long hw, void* stream);

/* train_latent_diffusion.py:600-601: out = uint8(255 * clamp(x / 2 + 0.5, 0, 1)), each intermediate rounded to dtype as
 * torch rounds it, the final cast truncating. */
int ldm_image_to_uint8(const void* x, int dtype, unsigned char* out, long long n, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* LDM_AMD_H */
