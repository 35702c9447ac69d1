// Kernels only the VAE-decoder leg of `generate` needs (train_latent_diffusion.py:596-601 around
// diffusers' AutoencoderKL.decode): row softmax for the single-head, 512-wide mid-block attention (whose scores are
// materialised by two plain GEMMs -- the head is wider than any flash tile), the 1x1 post_quant_conv on the NCHW latent,
// and the image post-processing to uint8.  Everything else in the decoder is the UNet's GroupNorm / implicit-GEMM kernels.
#include "common.h"

namespace ldm {

// ---------------------------------------------------------------------------------------- row softmax
// out[r][c] = softmax_c(scale * s[r][c]) over c < valid, 0 for valid <= c < cols.  One workgroup per row; the row is read
// three times (max, sum, write), the second and third from cache.  exp2 with the scale folded into log2(e).
template <typename TO>
__global__ void __launch_bounds__(256)
softmax_rows_kernel(const float* __restrict__ s, long long ld_in, int cols, int valid, float scale, TO* __restrict__ out,
                    long long ld_out) {
  __shared__ float red[4];
  const float* row = s + (size_t)blockIdx.x * ld_in;
  TO* orow = out + (size_t)blockIdx.x * ld_out;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const float k = scale * 1.44269504088896341f;

  float m = -INFINITY;
  for (int c = threadIdx.x; c < valid; c += 256) m = fmaxf(m, row[c]);
  m = wave_max(m);
  if (lane == 0) red[wave] = m;
  __syncthreads();
  m = fmaxf(fmaxf(red[0], red[1]), fmaxf(red[2], red[3]));
  __syncthreads();
  // scale > 0 is required by the host, so max(scale * s) = scale * max(s)
  float sum = 0.f;
  for (int c = threadIdx.x; c < valid; c += 256) sum += exp2f((row[c] - m) * k);
  sum = wave_sum(sum);
  if (lane == 0) red[wave] = sum;
  __syncthreads();
  sum = ((red[0] + red[1]) + red[2]) + red[3];
  const float inv = 1.0f / sum;
  for (int c = threadIdx.x; c < cols; c += 256) Elem<TO>::st(orow + c, c < valid ? exp2f((row[c] - m) * k) * inv : 0.f);
}

// ------------------------------------------------------------------------------ 1x1 conv on few NCHW channels
// out[b][co][p] = bias[co] + sum_ci w[co][ci] * x[b][ci][p]; float32 out.  cin, cout <= 8.
template <typename TX>
__global__ void __launch_bounds__(256)
conv1x1_nchw_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, float* __restrict__ out,
                    int batch, int cin, int cout, long long hw) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;   // over batch * hw
  if (i >= (long long)batch * hw) return;
  const long long b = i / hw, p = i - b * hw;
  float v[8];
  for (int ci = 0; ci < cin; ++ci) v[ci] = Elem<TX>::ld(x + (b * cin + ci) * hw + p);
  for (int co = 0; co < cout; ++co) {
    float acc = bias[co];
    for (int ci = 0; ci < cin; ++ci) acc += w[co * cin + ci] * v[ci];
    out[(b * cout + co) * hw + p] = acc;
  }
}

// ------------------------------------------------------------------------------------------ image -> uint8
// train_latent_diffusion.py:600-601: (255 * (x / 2 + 0.5).clamp(0, 1)).to(torch.uint8), every intermediate rounded to the
// tensor's own dtype as torch does (bf16 under autocast), the final cast truncating.
template <typename T> __device__ __forceinline__ float round_as(float v);
template <> __device__ __forceinline__ float round_as<float>(float v) { return v; }
template <> __device__ __forceinline__ float round_as<bf16_t>(float v) { return bf16_to_f32(f32_to_bf16(v)); }

template <typename T>
__global__ void __launch_bounds__(256) image_to_uint8_kernel(const T* __restrict__ x, uint8_t* __restrict__ out, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  float v = round_as<T>(Elem<T>::ld(x + i) * 0.5f);   // x / 2 is exact short of underflow
  v = round_as<T>(v + 0.5f);
  v = fminf(fmaxf(v, 0.0f), 1.0f);                    // clamp propagates NaN in torch; NaN -> 0 here like the uint8 cast of NaN
  v = round_as<T>(255.0f * v);
  out[i] = (uint8_t)(int)v;
}

}  // namespace ldm

using namespace ldm;

extern "C" int ldm_softmax_rows(const float* scores, long long ld_in, long long rows, int cols, int valid_cols, float scale,
                                void* out, int out_dtype, long long ld_out, void* stream) {
  LDM_REQUIRE(dtype_ok(out_dtype), "ldm_softmax_rows: bad dtype %d", out_dtype);
  LDM_REQUIRE(rows >= 0 && rows < (1LL << 31) && cols >= 1 && valid_cols >= 1 && valid_cols <= cols,
              "ldm_softmax_rows: bad sizes (rows=%lld cols=%d valid=%d)", rows, cols, valid_cols);
  LDM_REQUIRE(ld_in >= cols && ld_out >= cols, "ldm_softmax_rows: row pitch below the column count");
  LDM_REQUIRE(scale > 0.0f, "ldm_softmax_rows: scale must be positive");
  if (rows == 0) return LDM_OK;
  LDM_REQUIRE(scores && out, "ldm_softmax_rows: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (out_dtype == LDM_BF16)
    hipLaunchKernelGGL((softmax_rows_kernel<bf16_t>), dim3((unsigned)rows), dim3(256), 0, st, scores, ld_in, cols, valid_cols, scale,
                       (bf16_t*)out, ld_out);
  else
    hipLaunchKernelGGL((softmax_rows_kernel<float>), dim3((unsigned)rows), dim3(256), 0, st, scores, ld_in, cols, valid_cols, scale,
                       (float*)out, ld_out);
  LDM_HIP_CHECK_LAUNCH("ldm_softmax_rows");
  return LDM_OK;
}

extern "C" int ldm_conv1x1_nchw(const void* x, int x_dtype, const float* w, const float* bias, float* out, int batch, int cin,
                                int cout, long long hw, void* stream) {
  LDM_REQUIRE(dtype_ok(x_dtype), "ldm_conv1x1_nchw: bad dtype %d", x_dtype);
  LDM_REQUIRE(batch >= 0 && hw >= 1 && cin >= 1 && cin <= 8 && cout >= 1 && cout <= 8,
              "ldm_conv1x1_nchw: 1..8 channels each side (cin=%d cout=%d)", cin, cout);
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(x && w && bias && out, "ldm_conv1x1_nchw: null pointer");
  const long long n = (long long)batch * hw;
  LDM_REQUIRE((n + 255) / 256 < (1LL << 31), "ldm_conv1x1_nchw: too large");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (x_dtype == LDM_BF16)
    hipLaunchKernelGGL((conv1x1_nchw_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, w, bias, out, batch, cin, cout, hw);
  else
    hipLaunchKernelGGL((conv1x1_nchw_kernel<float>), grid, dim3(256), 0, st, (const float*)x, w, bias, out, batch, cin, cout, hw);
  LDM_HIP_CHECK_LAUNCH("ldm_conv1x1_nchw");
  return LDM_OK;
}

extern "C" int ldm_image_to_uint8(const void* x, int dtype, unsigned char* out, long long n, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_image_to_uint8: bad dtype %d", dtype);
  LDM_REQUIRE(n >= 0 && (n + 255) / 256 < (1LL << 31), "ldm_image_to_uint8: bad size");
  if (n == 0) return LDM_OK;
  LDM_REQUIRE(x && out, "ldm_image_to_uint8: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((image_to_uint8_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, out, n);
  else
    hipLaunchKernelGGL((image_to_uint8_kernel<float>), grid, dim3(256), 0, st, (const float*)x, out, n);
  LDM_HIP_CHECK_LAUNCH("ldm_image_to_uint8");
  return LDM_OK;
}
