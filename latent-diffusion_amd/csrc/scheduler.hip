// DDPM scheduler: host-side table construction and the fused elementwise device kernels.
// Compiled with -ffp-contract=off: the reference evaluates every product and sum as a separately
// rounded float32 torch op, and the results of these kernels are meant to be bit-identical to it.
#include "common.h"

#include <type_traits>

#include <math.h>
#include <vector>

namespace ldm {

// ------------------------------------------------------------------------------- host: tables
// torch.linspace(start, end, n, dtype=float64) on CPU: symmetric evaluation around the midpoint.
static void linspace_f64(double start, double end, int n, double* out) {
  if (n == 1) { out[0] = start; return; }
  const double step = (end - start) / (double)(n - 1);
  const int half = n / 2;
  for (int i = 0; i < n; ++i) out[i] = i < half ? start + step * (double)i : end - step * (double)(n - i - 1);
}

static int betas_f64(int schedule, int T, std::vector<double>& betas) {
  betas.resize(T);
  if (schedule == LDM_SCHED_LINEAR) {  // scheduling_ddpm.py:308-315
    const double k = 1000.0 / (double)T;
    linspace_f64(k * 0.0001, k * 0.02, T, betas.data());
    return LDM_OK;
  }
  std::vector<double> g(T + 1), acp(T + 1);
  linspace_f64(0.0, (double)T, T + 1, g.data());
  for (int i = 0; i <= T; ++i) g[i] = g[i] / (double)T;
  if (schedule == LDM_SCHED_COSINE) {  // :318-328
    const double s = 0.008;
    for (int i = 0; i <= T; ++i) {
      const double c = cos((g[i] + s) / (1 + s) * M_PI * 0.5);
      acp[i] = c * c;
    }
  } else if (schedule == LDM_SCHED_SIGMOID) {  // :331-343 (start=-3, end=3, tau=1); v_start/v_end are float32 scalars there
    const float v_lo = 1.0f / (1.0f + expf(3.0f));
    const float v_hi = 1.0f / (1.0f + expf(-3.0f));
    const double denom = (double)(v_hi - v_lo);
    for (int i = 0; i <= T; ++i) {
      const double x = (g[i] * 6.0 + (-3.0)) / 1.0;
      const double sg = 1.0 / (1.0 + exp(-x));
      acp[i] = (-sg + (double)v_hi) / denom;
    }
  } else {
    set_error("unknown beta schedule %d", schedule);
    return LDM_EINVAL;
  }
  const double a0 = acp[0];
  for (int i = 0; i <= T; ++i) acp[i] = acp[i] / a0;
  for (int i = 0; i < T; ++i) {
    double b = 1.0 - (acp[i + 1] / acp[i]);
    b = b < 0.0 ? 0.0 : (b > 0.999 ? 0.999 : b);
    betas[i] = b;
  }
  return LDM_OK;
}

}  // namespace ldm

using namespace ldm;

extern "C" int ldm_ddpm_build_tables(int schedule, int T, float* betas, float* acp, float* sqrt_acp,
                                     float* sqrt_1m, float* pm1, float* pm2, float* pstd) {
  LDM_REQUIRE(T >= 1, "ldm_ddpm_build_tables: T must be >= 1 (got %d)", T);
  LDM_REQUIRE(betas && acp && sqrt_acp && sqrt_1m && pm1 && pm2 && pstd, "ldm_ddpm_build_tables: null output");
  std::vector<double> b64;
  int rc = betas_f64(schedule, T, b64);
  if (rc) return rc;
  // torch.cumprod on a CPU float32 tensor keeps its running product in the accumulation type (double)
  // and rounds each prefix to float32 on store (:59); the golden tables pin exactly that.
  double acc = 1.0;
  float run = 1.0f;
  for (int i = 0; i < T; ++i) {
    const float b = (float)b64[i];
    const float a = 1.0f - b;
    const float prev = run;            // alpha_bar_{t-1}, 1.0 for t = 0 (scheduling_ddpm.py:70)
    acc *= (double)a;
    run = (float)acc;
    betas[i] = b;
    acp[i] = run;
    sqrt_acp[i] = sqrtf(run);
    const float om = 1.0f - run;
    sqrt_1m[i] = sqrtf(om);
    pm1[i] = b * sqrtf(prev) / om;                 // :71
    pm2[i] = (1.0f - prev) * sqrtf(a) / om;        // :72
    float var = b * (1.0f - prev) / om;            // :74
    var = var < 1e-20f ? 1e-20f : var;             // clamp(min=1e-20) (:75)
    pstd[i] = sqrtf(var);
  }
  return LDM_OK;
}

extern "C" int ldm_ddpm_set_timesteps(int spacing, int T, int n, int steps_offset, int64_t* ts) {
  LDM_REQUIRE(ts != nullptr, "ldm_ddpm_set_timesteps: null output");
  LDM_REQUIRE(n >= 1, "ldm_ddpm_set_timesteps: num_inference_steps must be >= 1 (got %d)", n);
  LDM_REQUIRE(n <= T,
              "`num_inference_steps`: %d cannot be larger than `self.train_timesteps`: %d as the unet model "
              "trained with this scheduler can only handle maximal %d timesteps.", n, T, T);
  if (spacing == LDM_SPACING_LEADING) {  // scheduling_ddpm.py:164-169 (integer arithmetic throughout)
    const int64_t ratio = T / n;
    for (int i = 0; i < n; ++i) ts[i] = (int64_t)(n - 1 - i) * ratio + steps_offset;
  } else if (spacing == LDM_SPACING_LINSPACE) {  // :157-163, float32 linspace, round half to even, flip
    // torch.linspace default dtype float32: step computed in float32, symmetric evaluation
    const float start = 0.0f, end = (float)(T - 1);
    const float step = n > 1 ? (end - start) / (float)(n - 1) : 0.0f;
    const int half = n / 2;
    for (int i = 0; i < n; ++i) {
      const float v = n == 1 ? start : (i < half ? start + step * (float)i : end - step * (float)(n - i - 1));
      ts[n - 1 - i] = (int64_t)nearbyintf(v);
    }
  } else if (spacing == LDM_SPACING_TRAILING) {  // :170-175: arange(T, 0, -T/n) in float32, round, -1
    // torch.arange(T, 0, -T/n) is a float32 tensor filled by torch's CPU kernel: size ceil((0-T)/step) in
    // double; values in blocks of 16 as two 8-lane vectors, lane j of a vector that starts at index i0 being
    // float(double(float(T + step*i0)) + j*step), and the last (size mod 16) values float(T + step*i) from
    // double.  The lane arithmetic decides round-half cases (n = 48: 62.50001 -> 63, not 62.5 -> 62), so it
    // is reproduced exactly; checked against torch for every n <= T at T = 250, 1000, 4000.
    const double stepd = -((double)T / (double)n);
    const int64_t cnt = (int64_t)ceil((0.0 - (double)T) / stepd);
    // For some n the double division above yields n+1 entries (an extra ~1e-13 that becomes timestep -1);
    // the reference's sampling loop then fails on the out-of-range index.  Reject those n up front.
    LDM_REQUIRE(cnt == n, "ldm_ddpm_set_timesteps: 'trailing' spacing with n=%d yields %lld entries in the "
                "reference (float step rounding); choose another num_inference_steps", n, (long long)cnt);
    int i = 0;
    for (; i <= n - 16; i += 16) {
      for (int b = 0; b < 16; b += 8) {
        const float base = (float)((double)T + stepd * (double)(i + b));
        for (int j = 0; j < 8; ++j)
          ts[i + b + j] = (int64_t)nearbyintf((float)((double)base + (double)j * stepd)) - 1;
      }
    }
    for (; i < n; ++i) ts[i] = (int64_t)nearbyintf((float)((double)T + stepd * (double)i)) - 1;
  } else {
    set_error("%d is not supported. Please make sure to choose one of 'linspace', 'leading' or 'trailing'.", spacing);
    return LDM_EINVAL;
  }
  return LDM_OK;
}

extern "C" int ldm_ddpm_step_coefficients(const float* acp, const float* sqrt_acp, const float* sqrt_1m, int T,
                                          int t, int s, float coef[5]) {
  LDM_REQUIRE(acp && sqrt_acp && sqrt_1m && coef, "ldm_ddpm_step_coefficients: null pointer");
  LDM_REQUIRE(t >= 0 && t < T, "ldm_ddpm_step_coefficients: timestep %d outside [0, %d)", t, T);
  LDM_REQUIRE(s >= -1 && s < T, "ldm_ddpm_step_coefficients: previous timestep %d outside [-1, %d)", s, T);
  const float A_t = acp[t];
  const float A_s = s >= 0 ? acp[s] : 1.0f;     // scheduling_ddpm.py:270-282
  const float a_ts = A_t / A_s;                 // :247
  const float b_ts = 1.0f - a_ts;               // :248
  const float omt = 1.0f - A_t;
  const float oms = 1.0f - A_s;
  coef[0] = sqrt_1m[t];                         // :113
  coef[1] = sqrt_acp[t];                        // :112
  coef[2] = sqrtf(A_s) * b_ts / omt;            // :250
  coef[3] = sqrtf(a_ts) * oms / omt;            // :251
  float var = oms * b_ts / omt;                 // :254
  var = var < 1e-20f ? 1e-20f : var;
  coef[4] = sqrtf(var);                         // :255
  return LDM_OK;
}

// ---------------------------------------------------------------------------- device kernels
namespace ldm {

struct StepCoef { float sigma, a_sqrt, c1, c2, stdv; };

template <typename TE, typename TX, bool CFG, bool WRITE_X0>
__global__ void __launch_bounds__(256)
ddpm_step_kernel(const TE* __restrict__ eps_u, const TE* __restrict__ eps_c, float guidance,
                 const TX* __restrict__ x_t, const float* __restrict__ noise, float* __restrict__ prev,
                 float* __restrict__ x0_out, int64_t n, StepCoef k) {
  const int64_t nvec = n >> 3;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nvec; i += stride) {
    float e[8], x[8], z[8], p[8], o[8];
    load8(eps_u + i * 8, e);
    if (CFG) {
      float c[8];
      load8(eps_c + i * 8, c);
#pragma unroll
      for (int j = 0; j < 8; ++j) e[j] = e[j] + guidance * (c[j] - e[j]);  // train_latent_diffusion.py:592
    }
    load8(x_t + i * 8, x);
    load8(noise + i * 8, z);
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const float x0 = (x[j] - k.sigma * e[j]) / k.a_sqrt;   // scheduling_ddpm.py:114
      const float mean = k.c1 * x0 + k.c2 * x[j];            // :253
      p[j] = mean + k.stdv * z[j];                           // :212
      o[j] = x0;
    }
    store8(prev + i * 8, p);
    if (WRITE_X0) store8(x0_out + i * 8, o);
  }
  // tail (n not a multiple of 8)
  const int64_t tail0 = nvec << 3;
  const int64_t gi = tail0 + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (gi < n) {
    float e = Elem<TE>::ld(eps_u + gi);
    if (CFG) { const float c = Elem<TE>::ld(eps_c + gi); e = e + guidance * (c - e); }
    const float x = Elem<TX>::ld(x_t + gi);
    const float x0 = (x - k.sigma * e) / k.a_sqrt;
    const float mean = k.c1 * x0 + k.c2 * x;
    prev[gi] = mean + k.stdv * noise[gi];
    if (WRITE_X0) x0_out[gi] = x0;
  }
}

// All-float32 form of the two kernels above for the sizes where they are HBM streams (the sampling job's whole batch in
// one launch): one float4 per lane per stream, lanes contiguous (1 KiB per wave-instruction; in the 8-per-thread form a
// wave-instruction touches every other 16 bytes of 2 KiB), one quad per thread and no loop, non-temporal loads and stores
// (nothing here is read twice).  Same arithmetic, same order: bit-identical results.  Measured on (256, 4, 64, 64) with
// DRAM-resident operands (tools/probes/stream_bench.hip): 4.3 -> 4.8 TB/s from the access shape, -> 5.2-5.3 with the
// non-temporal hint.
typedef __attribute__((ext_vector_type(4))) float v4f_t;
__device__ __forceinline__ void ld4_stream(const float* p, float (&v)[4]) {
  const v4f_t r = __builtin_nontemporal_load(reinterpret_cast<const v4f_t*>(p));
  v[0] = r.x; v[1] = r.y; v[2] = r.z; v[3] = r.w;
}
__device__ __forceinline__ void st4_stream(float* p, const float (&v)[4]) {
  __builtin_nontemporal_store(v4f_t{v[0], v[1], v[2], v[3]}, reinterpret_cast<v4f_t*>(p));
}

template <bool CFG, bool WRITE_X0>
__global__ void __launch_bounds__(256)
ddpm_step_stream_kernel(const float* __restrict__ eps_u, const float* __restrict__ eps_c, float guidance,
                        const float* __restrict__ x_t, const float* __restrict__ noise, float* __restrict__ prev,
                        float* __restrict__ x0_out, int64_t n, StepCoef k) {
  const int64_t nq = n >> 2;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
    float e[4], x[4], z[4], p[4], o[4];
    ld4_stream(eps_u + i * 4, e);
    if (CFG) {
      float c[4];
      ld4_stream(eps_c + i * 4, c);
#pragma unroll
      for (int j = 0; j < 4; ++j) e[j] = e[j] + guidance * (c[j] - e[j]);
    }
    ld4_stream(x_t + i * 4, x);
    ld4_stream(noise + i * 4, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const float x0 = (x[j] - k.sigma * e[j]) / k.a_sqrt;
      const float mean = k.c1 * x0 + k.c2 * x[j];
      p[j] = mean + k.stdv * z[j];
      o[j] = x0;
    }
    st4_stream(prev + i * 4, p);
    if (WRITE_X0) st4_stream(x0_out + i * 4, o);
  }
  const int64_t gi = (nq << 2) + (int64_t)blockIdx.x * blockDim.x + threadIdx.x;   // tail (n not a multiple of 4)
  if (gi < n) {
    float e = eps_u[gi];
    if (CFG) { const float c = eps_c[gi]; e = e + guidance * (c - e); }
    const float x = x_t[gi];
    const float x0 = (x - k.sigma * e) / k.a_sqrt;
    const float mean = k.c1 * x0 + k.c2 * x;
    prev[gi] = mean + k.stdv * noise[gi];
    if (WRITE_X0) x0_out[gi] = x0;
  }
}

__global__ void __launch_bounds__(256)
ddpm_add_noise_stream_kernel(const float* __restrict__ x0, const float* __restrict__ noise, float* __restrict__ out,
                             const int64_t* __restrict__ t, const float* __restrict__ sa, const float* __restrict__ so,
                             int T_steps, int64_t per_sample) {
  const int b = blockIdx.y;
  int64_t tb = t[b];
  const bool bad_t = tb < 0 || tb >= T_steps;             // as in ddpm_add_noise_kernel: NaN-poisoned, never silently clamped
  tb = tb < 0 ? 0 : (tb >= T_steps ? T_steps - 1 : tb);
  const float a = bad_t ? __builtin_nanf("") : sa[tb], s = bad_t ? __builtin_nanf("") : so[tb];
  const float* xb = x0 + (int64_t)b * per_sample;
  const float* nb = noise + (int64_t)b * per_sample;
  float* ob = out + (int64_t)b * per_sample;
  const int64_t nq = per_sample >> 2;                     // per_sample % 4 == 0 (checked on the host)
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < nq; i += stride) {
    float x[4], z[4], o[4];
    ld4_stream(xb + i * 4, x);
    ld4_stream(nb + i * 4, z);
#pragma unroll
    for (int j = 0; j < 4; ++j) o[j] = a * x[j] + s * z[j];
    st4_stream(ob + i * 4, o);
  }
}

template <typename T>
__global__ void __launch_bounds__(256)
ddpm_add_noise_kernel(const T* __restrict__ x0, const T* __restrict__ noise, float* __restrict__ out,
                      const int64_t* __restrict__ t, const float* __restrict__ sa, const float* __restrict__ so,
                      int T_steps, int64_t per_sample, int vec_ok) {
  const int b = blockIdx.y;
  int64_t tb = t[b];
  // A timestep outside [0, T) makes the reference's gather fail with a device assert (scheduling_ddpm.py:301).  Here the
  // index is clamped so that the table read cannot fault and the sample's whole output is poisoned with NaN: a bad timestep
  // must not come back as plausible noisy latents.
  const bool bad_t = tb < 0 || tb >= T_steps;
  tb = tb < 0 ? 0 : (tb >= T_steps ? T_steps - 1 : tb);
  const float a = bad_t ? __builtin_nanf("") : sa[tb], s = bad_t ? __builtin_nanf("") : so[tb];
  const T* xb = x0 + (int64_t)b * per_sample;
  const T* nb = noise + (int64_t)b * per_sample;
  float* ob = out + (int64_t)b * per_sample;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int64_t tid = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (vec_ok) {
    const int64_t nvec = per_sample >> 3;
    for (int64_t i = tid; i < nvec; i += stride) {
      float x[8], z[8], o[8];
      load8(xb + i * 8, x);
      load8(nb + i * 8, z);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] = a * x[j] + s * z[j];  // scheduling_ddpm.py:100
      store8(ob + i * 8, o);
    }
  } else {
    for (int64_t i = tid; i < per_sample; i += stride)
      ob[i] = a * Elem<T>::ld(xb + i) + s * Elem<T>::ld(nb + i);
  }
}

// Training-side input preparation in one pass (train_latent_diffusion.py:1039-1064): posterior sample of the
// VAE's (mean | logvar) moments times the scaling factor, optional offset noise (one draw per sample and channel), optional
// input perturbation, then q(x_t | x_0).  All draws are inputs (the caller's torch generators).  float32 throughout, every
// product and sum rounded separately (-ffp-contract=off) like the torch ops it replaces.
//   latents = sf * (mean + exp(0.5 * clamp(logvar, -30, 20)) * eps)          diffusers DiagonalGaussianDistribution.sample
//   noise'  = noise + noise_offset * offset[b][c]                             (:1046)
//   noisy   = sqrt_acp[t_b] * latents + sqrt(1 - acp)[t_b] * (noise' + perturbation * extra)     (:1050, :1062)
__global__ void __launch_bounds__(256)
train_input_prep_kernel(const float* __restrict__ moments, const float* __restrict__ eps, float* __restrict__ noise,
                        const float* __restrict__ offset, const float* __restrict__ extra, const int64_t* __restrict__ t,
                        const float* __restrict__ sa, const float* __restrict__ so, int T_steps, int channels, int64_t hw,
                        float scaling, float noise_offset, float perturbation, float* __restrict__ latents,
                        float* __restrict__ noisy) {
  const int b = blockIdx.y;
  int64_t tb = t[b];
  const bool bad_t = tb < 0 || tb >= T_steps;             // as in ddpm_add_noise_kernel: NaN-poisoned, never silently clamped
  tb = tb < 0 ? 0 : (tb >= T_steps ? T_steps - 1 : tb);
  const float a = bad_t ? __builtin_nanf("") : sa[tb], s = bad_t ? __builtin_nanf("") : so[tb];
  const int64_t per_sample = (int64_t)channels * hw;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += stride) {
    const int c = (int)(i / hw);
    const int64_t o = (int64_t)b * per_sample + i;
    const float mean = moments[(int64_t)b * 2 * per_sample + i];
    float logvar = moments[(int64_t)b * 2 * per_sample + per_sample + i];
    logvar = fminf(fmaxf(logvar, -30.0f), 20.0f);
    const float std_ = expf(0.5f * logvar);
    const float lat = scaling * (mean + std_ * eps[o]);
    float nz = noise[o];
    if (offset != nullptr) {
      nz = nz + noise_offset * offset[b * channels + c];
      noise[o] = nz;                                   // the loss target is the offset noise (:1108 uses `noise`)
    }
    const float nin = extra != nullptr ? nz + perturbation * extra[o] : nz;
    latents[o] = lat;
    noisy[o] = a * lat + s * nin;
  }
}

// enc_h[b] = drop[b] ? null_emb : enc_h[b]   (text-conditioning dropout, :1076-1079), in place; T = bf16 or float
template <typename T>
__global__ void __launch_bounds__(256)
cond_dropout_kernel(T* __restrict__ enc, const T* __restrict__ null_emb, const unsigned char* __restrict__ drop, int64_t per_sample) {
  const int b = blockIdx.y;
  if (!drop[b]) return;
  T* e = enc + (int64_t)b * per_sample;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < per_sample; i += stride) e[i] = null_emb[i];
}

static int grid_for(int64_t work_items, int block) {
  int64_t g = (work_items + block - 1) / block;
  if (g < 1) g = 1;
  if (g > 2048) g = 2048;  // 256 CUs x 8 blocks, grid-stride beyond
  return (int)g;
}

template <typename TE, typename TX>
static int launch_step(const void* eu, const void* ec, bool cfg, float g, const void* x, const float* noise,
                       float* prev, float* x0, int64_t n, const float* coef, hipStream_t st) {
  StepCoef k{coef[0], coef[1], coef[2], coef[3], coef[4]};
  if constexpr (std::is_same<TE, float>::value && std::is_same<TX, float>::value) {
    // one quad per thread; the tail lanes live in the first block; grids past 65,536 blocks walk with a stride
    int64_t blocks = ((n >> 2) + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 65536) blocks = 65536;
    dim3 sgrid((unsigned)blocks), sblock(256);
#define LDM_LAUNCH_STREAM(CFG_, X0_)                                                                         \
    hipLaunchKernelGGL((ddpm_step_stream_kernel<CFG_, X0_>), sgrid, sblock, 0, st, (const float*)eu,        \
                       (const float*)ec, g, (const float*)x, noise, prev, x0, n, k)
    if (cfg) { if (x0) LDM_LAUNCH_STREAM(true, true); else LDM_LAUNCH_STREAM(true, false); }
    else     { if (x0) LDM_LAUNCH_STREAM(false, true); else LDM_LAUNCH_STREAM(false, false); }
#undef LDM_LAUNCH_STREAM
    LDM_HIP_CHECK_LAUNCH("ldm_ddpm_step");
    return LDM_OK;
  }
  // the tail lanes live in the first blocks; the grid must cover max(nvec, 8) work items
  const int64_t items = (n >> 3) > 8 ? (n >> 3) : 8;
  dim3 grid(grid_for(items, 256)), block(256);
#define LDM_LAUNCH_STEP(CFG_, X0_)                                                                     \
  hipLaunchKernelGGL((ddpm_step_kernel<TE, TX, CFG_, X0_>), grid, block, 0, st, (const TE*)eu,         \
                     (const TE*)ec, g, (const TX*)x, noise, prev, x0, n, k)
  if (cfg) { if (x0) LDM_LAUNCH_STEP(true, true); else LDM_LAUNCH_STEP(true, false); }
  else     { if (x0) LDM_LAUNCH_STEP(false, true); else LDM_LAUNCH_STEP(false, false); }
#undef LDM_LAUNCH_STEP
  LDM_HIP_CHECK_LAUNCH("ldm_ddpm_step");
  return LDM_OK;
}

static int step_common(const void* eu, const void* ec, bool cfg, int eps_dtype, float g, const void* x, int x_dtype,
                       const float* noise, float* prev, float* x0, int64_t n, const float* coef, void* stream) {
  LDM_REQUIRE(dtype_ok(eps_dtype) && dtype_ok(x_dtype), "ldm_ddpm_step: bad dtype (eps %d, x_t %d)", eps_dtype, x_dtype);
  LDM_REQUIRE(n >= 0, "ldm_ddpm_step: negative element count");
  if (n == 0) return LDM_OK;
  LDM_REQUIRE(eu && x && noise && prev && coef && (!cfg || ec), "ldm_ddpm_step: null pointer");
  LDM_REQUIRE(aligned16(eu) && aligned16(x) && aligned16(noise) && aligned16(prev) && (!x0 || aligned16(x0)) &&
              (!cfg || aligned16(ec)), "ldm_ddpm_step: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (eps_dtype == LDM_F32 && x_dtype == LDM_F32) return launch_step<float, float>(eu, ec, cfg, g, x, noise, prev, x0, n, coef, st);
  if (eps_dtype == LDM_BF16 && x_dtype == LDM_F32) return launch_step<bf16_t, float>(eu, ec, cfg, g, x, noise, prev, x0, n, coef, st);
  if (eps_dtype == LDM_F32 && x_dtype == LDM_BF16) return launch_step<float, bf16_t>(eu, ec, cfg, g, x, noise, prev, x0, n, coef, st);
  return launch_step<bf16_t, bf16_t>(eu, ec, cfg, g, x, noise, prev, x0, n, coef, st);
}

// ------------------------------------------------------------------------------- per-sample Gaussian noise
// Counter-based N(0,1) draws keyed by the GLOBAL sample index (noise must be a function of (seed, step, sample),
// not of the shard): Philox4x32-10 (Salmon et al., SC'11) with key = seed and counter = (block of four elements inside the
// sample, sample index lo / hi, step + 1); the four 32-bit outputs become two Box-Muller pairs.  A rank draws ONLY its rows;
// a row's values are the same on any rank of any world size.  oracle/noise_ref.py restates the same function in numpy.
__device__ __forceinline__ void philox4x32_10(uint32_t (&c)[4], uint32_t k0, uint32_t k1) {
#pragma unroll
  for (int r = 0; r < 10; ++r) {
    const uint64_t p0 = (uint64_t)0xD2511F53u * c[0], p1 = (uint64_t)0xCD9E8D57u * c[2];
    const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0, n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
    c[1] = (uint32_t)p1; c[3] = (uint32_t)p0; c[0] = n0; c[2] = n2;
    k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
  }
}

__global__ void __launch_bounds__(256)
gaussian_rows_kernel(uint32_t seed_lo, uint32_t seed_hi, uint32_t step1, int64_t row0, int64_t rows, int64_t row_elems,
                     float* __restrict__ out) {
  const int64_t quads = (row_elems + 3) >> 2;
  const int64_t total = rows * quads;
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
    const int64_t r = i / quads, q = i - r * quads;
    const uint64_t grow = (uint64_t)(row0 + r);
    uint32_t c[4] = {(uint32_t)q, (uint32_t)grow, (uint32_t)(grow >> 32), step1};
    philox4x32_10(c, seed_lo, seed_hi);
    float z[4];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
      // 23 random bits + 1/2 in (0, 1): exact in float32 on every host and device
      const float u0 = ((float)(c[2 * h] >> 9) + 0.5f) * 1.1920928955078125e-07f;
      const float u1 = ((float)(c[2 * h + 1] >> 9) + 0.5f) * 1.1920928955078125e-07f;
      const float rad = sqrtf(-2.0f * logf(u0));
      float sn, cs;
      sincosf(6.283185307179586f * u1, &sn, &cs);
      z[2 * h] = rad * cs;
      z[2 * h + 1] = rad * sn;
    }
    float* o = out + r * row_elems + q * 4;
    if (q * 4 + 4 <= row_elems && (row_elems & 3) == 0) {
      *reinterpret_cast<float4*>(o) = make_float4(z[0], z[1], z[2], z[3]);
    } else {
#pragma unroll
      for (int e = 0; e < 4; ++e)
        if (q * 4 + e < row_elems) o[e] = z[e];
    }
  }
}

}  // namespace ldm

extern "C" int ldm_ddpm_step(const void* eps, int eps_dtype, const void* x_t, int x_dtype, const float* noise,
                             float* prev, float* x0, int64_t n, const float coef[5], void* stream) {
  return step_common(eps, nullptr, false, eps_dtype, 0.0f, x_t, x_dtype, noise, prev, x0, n, coef, stream);
}

extern "C" int ldm_ddpm_step_cfg(const void* eps_u, const void* eps_c, int eps_dtype, float guidance, const void* x_t,
                                 int x_dtype, const float* noise, float* prev, float* x0, int64_t n,
                                 const float coef[5], void* stream) {
  return step_common(eps_u, eps_c, true, eps_dtype, guidance, x_t, x_dtype, noise, prev, x0, n, coef, stream);
}

extern "C" int ldm_ddpm_add_noise(const void* x0, const void* noise, float* out, int dtype, const int64_t* t,
                                  const float* sa, const float* so, int T, int batch, int64_t per_sample,
                                  void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_ddpm_add_noise: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && per_sample >= 0 && T >= 1, "ldm_ddpm_add_noise: bad sizes");
  if (batch == 0 || per_sample == 0) return LDM_OK;
  LDM_REQUIRE(batch <= 65535, "ldm_ddpm_add_noise: batch %d exceeds 65535", batch);
  LDM_REQUIRE(x0 && noise && out && t && sa && so, "ldm_ddpm_add_noise: null pointer");
  const int vec_ok = (per_sample % 8 == 0) && aligned16(x0) && aligned16(noise) && aligned16(out);
  const int64_t items = vec_ok ? per_sample / 8 : per_sample;
  int gx = grid_for(items, 256);
  if ((int64_t)gx * batch > 4096) gx = (int)((4096 + batch - 1) / batch);
  if (gx < 1) gx = 1;
  dim3 grid(gx, batch), block(256);
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LDM_F32 && vec_ok) {
    int64_t sx = ((per_sample >> 2) + 255) / 256;
    if (sx * batch > 65536) sx = (65536 + batch - 1) / batch;
    if (sx < 1) sx = 1;
    hipLaunchKernelGGL(ddpm_add_noise_stream_kernel, dim3((unsigned)sx, batch), block, 0, st, (const float*)x0, (const float*)noise,
                       (float*)out, t, sa, so, T, per_sample);
  } else if (dtype == LDM_F32)
    hipLaunchKernelGGL((ddpm_add_noise_kernel<float>), grid, block, 0, st, (const float*)x0, (const float*)noise,
                       (float*)out, t, sa, so, T, per_sample, vec_ok);
  else
    hipLaunchKernelGGL((ddpm_add_noise_kernel<bf16_t>), grid, block, 0, st, (const bf16_t*)x0, (const bf16_t*)noise,
                       (float*)out, t, sa, so, T, per_sample, vec_ok);
  LDM_HIP_CHECK_LAUNCH("ldm_ddpm_add_noise");
  return LDM_OK;
}

extern "C" int ldm_train_input_prep(const float* moments, const float* eps, float* noise, const float* offset, const float* extra,
                                    const int64_t* t, const float* sa, const float* so, int T, int batch, int channels, int64_t hw,
                                    float scaling, float noise_offset, float perturbation, float* latents, float* noisy,
                                    void* stream) {
  LDM_REQUIRE(batch >= 0 && channels >= 1 && hw >= 1 && T >= 1, "ldm_train_input_prep: bad sizes");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(batch <= 65535, "ldm_train_input_prep: batch %d exceeds 65535", batch);
  LDM_REQUIRE(moments && eps && noise && t && sa && so && latents && noisy, "ldm_train_input_prep: null pointer");
  int gx = grid_for((int64_t)channels * hw, 256);
  if ((int64_t)gx * batch > 4096) gx = (int)((4096 + batch - 1) / batch);
  if (gx < 1) gx = 1;
  hipLaunchKernelGGL(train_input_prep_kernel, dim3(gx, batch), dim3(256), 0, (hipStream_t)stream, moments, eps, noise, offset, extra, t,
                     sa, so, T, channels, hw, scaling, noise_offset, perturbation, latents, noisy);
  LDM_HIP_CHECK_LAUNCH("ldm_train_input_prep");
  return LDM_OK;
}

extern "C" int ldm_cond_dropout(void* enc, const void* null_emb, const unsigned char* drop, int dtype, int batch, int64_t per_sample,
                                void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_cond_dropout: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && per_sample >= 0, "ldm_cond_dropout: bad sizes");
  if (batch == 0 || per_sample == 0) return LDM_OK;
  LDM_REQUIRE(batch <= 65535, "ldm_cond_dropout: batch %d exceeds 65535", batch);
  LDM_REQUIRE(enc && null_emb && drop, "ldm_cond_dropout: null pointer");
  int gx = grid_for(per_sample, 256);
  if (gx > 64) gx = 64;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((cond_dropout_kernel<bf16_t>), dim3(gx, batch), dim3(256), 0, st, (bf16_t*)enc, (const bf16_t*)null_emb, drop, per_sample);
  else
    hipLaunchKernelGGL((cond_dropout_kernel<float>), dim3(gx, batch), dim3(256), 0, st, (float*)enc, (const float*)null_emb, drop, per_sample);
  LDM_HIP_CHECK_LAUNCH("ldm_cond_dropout");
  return LDM_OK;
}

extern "C" int ldm_gaussian_noise(uint64_t seed, int64_t step, int64_t row0, int64_t rows, int64_t row_elems, float* out,
                                  void* stream) {
  LDM_REQUIRE(rows >= 0 && row_elems >= 0 && row0 >= 0 && step >= -1 && step < 0xfffffffeLL, "ldm_gaussian_noise: bad sizes");
  if (rows == 0 || row_elems == 0) return LDM_OK;
  LDM_REQUIRE(out != nullptr && aligned16(out), "ldm_gaussian_noise: out must be a 16-byte aligned device pointer");
  LDM_REQUIRE((row_elems + 3) / 4 < (1LL << 32), "ldm_gaussian_noise: rows longer than 2^34 elements");
  const int64_t total = rows * ((row_elems + 3) / 4);
  int64_t blocks = (total + 255) / 256;
  if (blocks > 8192) blocks = 8192;
  hipLaunchKernelGGL(gaussian_rows_kernel, dim3((unsigned)blocks), dim3(256), 0, (hipStream_t)stream, (uint32_t)seed,
                     (uint32_t)(seed >> 32), (uint32_t)(step + 1), row0, rows, row_elems, out);
  LDM_HIP_CHECK_LAUNCH("ldm_gaussian_noise");
  return LDM_OK;
}
