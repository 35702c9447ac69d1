// Training-step tail around the UNet (train_latent_diffusion.py:1108-1126): noise-prediction MSE with its
// gradient, global gradient-norm clipping and AdamW over all 654 parameter tensors -- HBM-bound multi-tensor kernels.
//
// A tensor list is described by DEVICE tables the caller builds once per step (pointers change when gradients are fresh
// tensors): per tensor its pointers, per chunk (at most LDM_OPT_CHUNK elements of one tensor) the tensor index, the element
// offset and the length.  One workgroup per chunk; 16-byte accesses where the chunk start is 16-byte aligned (always: chunk
// offsets are multiples of LDM_OPT_CHUNK and torch allocations are 256-byte aligned), scalar tail otherwise.
//   * sum of squares: per-chunk partial (fixed in-block order) -> one workgroup folds the partials in index order and writes
//     {norm, clip scale = min(1, max_norm / (norm + 1e-6))} (torch.nn.utils.clip_grad_norm_'s formula).  Deterministic.
//   * AdamW: reads the clip scale from device memory (no host sync between the norm and the update), applies it to the
//     gradient on the fly, and performs torch.optim.AdamW's update (decoupled weight decay, bias correction) in one pass:
//     28 B per parameter (p, m, v read + written, g read) against >= 44 B for clip + unfused update.
#include "common.h"

namespace ldm {

constexpr int OPT_CHUNK = 16384;     // elements per chunk / workgroup (256 threads x 8 elements x 8 rounds)

struct OptTables {
  const unsigned long long* g;       // [ntensors] gradient pointers
  const unsigned long long* p;       // [ntensors] parameter pointers (AdamW only)
  const unsigned long long* m;       // [ntensors] exp_avg
  const unsigned long long* v;       // [ntensors] exp_avg_sq
  const int* chunk_tensor;           // [nchunks]
  const long long* chunk_off;        // [nchunks] element offset inside the tensor
  const int* chunk_len;              // [nchunks] <= OPT_CHUNK
};

__device__ __forceinline__ float block_sum_256(float s) {
  __shared__ float red[4];
  s = wave_sum(s);
  if ((threadIdx.x & 63) == 0) red[threadIdx.x >> 6] = s;
  __syncthreads();
  return (red[0] + red[1]) + (red[2] + red[3]);
}

__global__ void __launch_bounds__(256)
opt_sumsq_kernel(OptTables t, float* __restrict__ partial) {
  const int c = blockIdx.x;
  const float* g = reinterpret_cast<const float*>(t.g[t.chunk_tensor[c]]) + t.chunk_off[c];
  const int len = t.chunk_len[c];
  float s = 0.f;
  const bool vec = (reinterpret_cast<uintptr_t>(g) & 15) == 0;
  const int nv = vec ? len >> 2 : 0;
  for (int i = threadIdx.x; i < nv; i += 256) {
    const float4 x = reinterpret_cast<const float4*>(g)[i];
    s += (x.x * x.x + x.y * x.y) + (x.z * x.z + x.w * x.w);
  }
  for (int i = nv * 4 + threadIdx.x; i < len; i += 256) s += g[i] * g[i];
  s = block_sum_256(s);
  if (threadIdx.x == 0) partial[c] = s;
}

// out[0] = sqrt(sum partial) = total L2 norm, out[1] = clip scale
__global__ void __launch_bounds__(256)
opt_norm_finalize_kernel(const float* __restrict__ partial, int n, float max_norm, float* __restrict__ out) {
  double s = 0.0;                                 // fixed order: thread-strided, then the block tree
  for (int i = threadIdx.x; i < n; i += 256) s += (double)partial[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) {
    const float norm = (float)sqrt(red[0]);
    out[0] = norm;
    const float sc = max_norm / (norm + 1e-6f);
    out[1] = max_norm > 0.f ? (sc < 1.0f ? sc : 1.0f) : 1.0f;
  }
}

struct AdamWScalars {
  float lr, beta1, beta2, eps, weight_decay;
  float decay;                              // 1 - lr * weight_decay
  float step_size;                          // lr / bias_correction1
  float one_minus_beta1, one_minus_beta2;   // rounded from the double difference, as torch rounds its Python floats
  float bias_correction1;            // 1 - beta1^step
  float inv_sqrt_bias_correction2;   // 1 / sqrt(1 - beta2^step)
};

__device__ __forceinline__ void adamw_one(float& p, float g, float& m, float& v, const AdamWScalars& a, float step_size) {
  // torch.optim.AdamW (_single_tensor_adamw): p *= 1 - lr*wd; m = lerp(m, g, 1-b1); v = b2*v + (1-b2) g^2;
  // denom = sqrt(v)/sqrt(bc2) + eps; p -= (lr/bc1) * m / denom
  p *= a.decay;
  m = m + (g - m) * a.one_minus_beta1;
  v = v * a.beta2 + a.one_minus_beta2 * (g * g);
  const float denom = sqrtf(v) * a.inv_sqrt_bias_correction2 + a.eps;
  p -= step_size * (m / denom);
}

__global__ void __launch_bounds__(256)
opt_adamw_kernel(OptTables t, AdamWScalars a, const float* __restrict__ clip) {
  const int c = blockIdx.x;
  const int ti = t.chunk_tensor[c];
  const long long off = t.chunk_off[c];
  const int len = t.chunk_len[c];
  float* p = reinterpret_cast<float*>(t.p[ti]) + off;
  const float* g = reinterpret_cast<const float*>(t.g[ti]) + off;
  float* m = reinterpret_cast<float*>(t.m[ti]) + off;
  float* v = reinterpret_cast<float*>(t.v[ti]) + off;
  const float gs = clip != nullptr ? clip[1] : 1.0f;
  const float step_size = a.step_size;
  const bool vec = ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                     reinterpret_cast<uintptr_t>(v)) & 15) == 0;
  const int nv = vec ? len >> 2 : 0;
  for (int i = threadIdx.x; i < nv; i += 256) {
    float4 pp = reinterpret_cast<float4*>(p)[i], mm = reinterpret_cast<float4*>(m)[i], vv = reinterpret_cast<float4*>(v)[i];
    const float4 gg = reinterpret_cast<const float4*>(g)[i];
    adamw_one(pp.x, gg.x * gs, mm.x, vv.x, a, step_size);
    adamw_one(pp.y, gg.y * gs, mm.y, vv.y, a, step_size);
    adamw_one(pp.z, gg.z * gs, mm.z, vv.z, a, step_size);
    adamw_one(pp.w, gg.w * gs, mm.w, vv.w, a, step_size);
    reinterpret_cast<float4*>(p)[i] = pp;
    reinterpret_cast<float4*>(m)[i] = mm;
    reinterpret_cast<float4*>(v)[i] = vv;
  }
  for (int i = nv * 4 + threadIdx.x; i < len; i += 256) adamw_one(p[i], g[i] * gs, m[i], v[i], a, step_size);
}

// ---------------------------------------------------------------------------------------- MSE + gradient
constexpr int MSE_BLOCK = 2048;      // elements per workgroup

template <typename T>
__global__ void __launch_bounds__(256)
mse_partial_kernel(const T* __restrict__ pred, const float* __restrict__ target, long long n, float grad_scale,
                   T* __restrict__ dpred, float* __restrict__ partial) {
  const long long base = (long long)blockIdx.x * MSE_BLOCK;
  float s = 0.f;
  for (int k = threadIdx.x; k < MSE_BLOCK; k += 256) {
    const long long i = base + k;
    if (i < n) {
      const float d = Elem<T>::ld(pred + i) - target[i];
      s += d * d;
      if (dpred != nullptr) Elem<T>::st(dpred + i, d * grad_scale);
    }
  }
  s = block_sum_256(s);
  if (threadIdx.x == 0) partial[blockIdx.x] = s;
}

__global__ void __launch_bounds__(256)
mse_finalize_kernel(const float* __restrict__ partial, int nblocks, double inv_n, float* __restrict__ loss) {
  double s = 0.0;
  for (int i = threadIdx.x; i < nblocks; i += 256) s += (double)partial[i];
  __shared__ double red[256];
  red[threadIdx.x] = s;
  __syncthreads();
  for (int o = 128; o > 0; o >>= 1) {
    if ((int)threadIdx.x < o) red[threadIdx.x] += red[threadIdx.x + o];
    __syncthreads();
  }
  if (threadIdx.x == 0) loss[0] = (float)(red[0] * inv_n);
}

static int check_tables(const char* who, const void* g, const void* ct, const void* co, const void* cl, int ntensors, int nchunks) {
  LDM_REQUIRE(ntensors >= 0 && nchunks >= 0, "%s: negative counts", who);
  LDM_REQUIRE(nchunks == 0 || (g && ct && co && cl), "%s: null table", who);
  return LDM_OK;
}

}  // namespace ldm

using namespace ldm;

extern "C" int ldm_opt_chunk_elements(void) { return OPT_CHUNK; }

extern "C" int ldm_grad_norm_clip(const unsigned long long* grads, int ntensors, const int* chunk_tensor, const long long* chunk_off,
                                  const int* chunk_len, int nchunks, float max_norm, float* partial, float* out, void* stream) {
  int rc = check_tables("ldm_grad_norm_clip", grads, chunk_tensor, chunk_off, chunk_len, ntensors, nchunks);
  if (rc) return rc;
  LDM_REQUIRE(out != nullptr && (nchunks == 0 || partial != nullptr), "ldm_grad_norm_clip: null output");
  hipStream_t st = (hipStream_t)stream;
  OptTables t{grads, nullptr, nullptr, nullptr, chunk_tensor, chunk_off, chunk_len};
  if (nchunks > 0) {
    hipLaunchKernelGGL(opt_sumsq_kernel, dim3(nchunks), dim3(256), 0, st, t, partial);
    LDM_HIP_CHECK_LAUNCH("ldm_grad_norm_clip(partial)");
  }
  hipLaunchKernelGGL(opt_norm_finalize_kernel, dim3(1), dim3(256), 0, st, partial, nchunks, max_norm, out);
  LDM_HIP_CHECK_LAUNCH("ldm_grad_norm_clip(finalize)");
  return LDM_OK;
}

extern "C" int ldm_adamw_step(const unsigned long long* params, const unsigned long long* grads, const unsigned long long* exp_avg,
                              const unsigned long long* exp_avg_sq, int ntensors, const int* chunk_tensor, const long long* chunk_off,
                              const int* chunk_len, int nchunks, double lr, double beta1, double beta2, double eps, double weight_decay,
                              long long step, const float* clip, void* stream) {
  int rc = check_tables("ldm_adamw_step", grads, chunk_tensor, chunk_off, chunk_len, ntensors, nchunks);
  if (rc) return rc;
  LDM_REQUIRE(step >= 1, "ldm_adamw_step: step counts from 1 (got %lld)", step);
  LDM_REQUIRE(nchunks == 0 || (params && exp_avg && exp_avg_sq), "ldm_adamw_step: null table");
  if (nchunks == 0) return LDM_OK;
  AdamWScalars a;
  a.lr = (float)lr; a.beta1 = (float)beta1; a.beta2 = (float)beta2; a.eps = (float)eps; a.weight_decay = (float)weight_decay;
  a.decay = (float)(1.0 - lr * weight_decay);
  a.step_size = (float)(lr / (1.0 - pow(beta1, (double)step)));
  a.one_minus_beta1 = (float)(1.0 - beta1); a.one_minus_beta2 = (float)(1.0 - beta2);
  a.bias_correction1 = (float)(1.0 - pow(beta1, (double)step));
  a.inv_sqrt_bias_correction2 = (float)(1.0 / sqrt(1.0 - pow(beta2, (double)step)));
  OptTables t{grads, params, exp_avg, exp_avg_sq, chunk_tensor, chunk_off, chunk_len};
  hipLaunchKernelGGL(opt_adamw_kernel, dim3(nchunks), dim3(256), 0, (hipStream_t)stream, t, a, clip);
  LDM_HIP_CHECK_LAUNCH("ldm_adamw_step");
  return LDM_OK;
}

extern "C" long long ldm_mse_workspace_floats(long long n) { return (n + MSE_BLOCK - 1) / MSE_BLOCK; }

extern "C" int ldm_mse_loss(const void* pred, int dtype, const float* target, long long n, float* loss, void* dpred, float* workspace,
                            void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_mse_loss: bad dtype %d", dtype);
  LDM_REQUIRE(n >= 1, "ldm_mse_loss: empty input");
  LDM_REQUIRE(pred && target && loss && workspace, "ldm_mse_loss: null pointer");
  const long long nblocks = (n + MSE_BLOCK - 1) / MSE_BLOCK;
  LDM_REQUIRE(nblocks < (1LL << 31), "ldm_mse_loss: too many elements");
  hipStream_t st = (hipStream_t)stream;
  const float gs = (float)(2.0 / (double)n);       // d mean((p - t)^2) / dp
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((mse_partial_kernel<bf16_t>), dim3((unsigned)nblocks), dim3(256), 0, st, (const bf16_t*)pred, target, n, gs,
                       (bf16_t*)dpred, workspace);
  else
    hipLaunchKernelGGL((mse_partial_kernel<float>), dim3((unsigned)nblocks), dim3(256), 0, st, (const float*)pred, target, n, gs,
                       (float*)dpred, workspace);
  LDM_HIP_CHECK_LAUNCH("ldm_mse_loss(partial)");
  hipLaunchKernelGGL(mse_finalize_kernel, dim3(1), dim3(256), 0, st, workspace, (int)nblocks, 1.0 / (double)n, loss);
  LDM_HIP_CHECK_LAUNCH("ldm_mse_loss(finalize)");
  return LDM_OK;
}
