// HBM-bound backward kernels of the UNet (everything that is not a GEMM or attention): column sums (bias and
// time-embedding gradients), GroupNorm(+SiLU) and LayerNorm backward, GEGLU / SwiGLU forward (training form, keeps the
// pre-activation) and backward, SiLU backward, and the 2x2 fold that is the gradient of nearest-2x upsampling.
// All streaming: 16-byte accesses, fp32 arithmetic, reductions in fixed order (deterministic).
#include "common.h"

namespace ldm {

// ------------------------------------------------------------------------------------- 2x2 fold
template <typename T>
__global__ void __launch_bounds__(256)
sum2x2_kernel(const T* __restrict__ x, const T* __restrict__ residual, T* __restrict__ out, int hout, int wout, int c, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;      // over (b, oh, ow, chunk)
  if (idx >= total) return;
  const int cpr = c >> 3;
  const int ch = (int)(idx % cpr);
  const long long pix = idx / cpr;
  const int ow = (int)(pix % wout);
  const long long t = pix / wout;
  const int oh = (int)(t % hout);
  const long long b = t / hout;
  const T* src = x + (((b * (2 * hout) + 2 * oh) * (2LL * wout)) + 2 * ow) * c + ch * 8;
  float v[8], a[8];
  load8(src, v);
  load8(src + c, a);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += a[e];
  load8(src + 2LL * wout * c, a);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += a[e];
  load8(src + 2LL * wout * c + c, a);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] += a[e];
  if (residual != nullptr) {
    load8(residual + pix * c + ch * 8, a);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += a[e];
  }
  store8(out + pix * c + ch * 8, v);
}

// ------------------------------------------------------------------------------------ column sums
// Two stages.  partial[(seg*slabs + slab)][n] = sum over the slab's rows; then a fold over the slabs.
// Block = 64 chunk-lanes (512 columns) x 4 row-lanes.
constexpr int CS_MAX_SLABS = 256;

// Partial column sums of one slab of rows.  A workgroup covers `cb` 8-column chunks (all of them when n <= 512, else 64 per
// blockIdx.z) x rp = 256 / cb rows per pass, so that every lane has work whatever n is (with a fixed 64 chunks per wave a
// 320-column matrix left 24 of 64 lanes idle), four rows in flight per thread; the rp partial rows meet in LDS in a fixed order.
template <typename T>
__global__ void __launch_bounds__(256)
colsum_partial_kernel(const T* __restrict__ x, long long rows_per_segment, int n, int slab_rows, int cb, float* __restrict__ partial) {
  __shared__ float red[256][8];
  const int slab = blockIdx.x, nslabs = gridDim.x;
  const long long seg = blockIdx.y;
  const int rp = 256 / cb;
  const int tp = threadIdx.x / cb, tc = threadIdx.x - tp * cb;
  const int chunk = blockIdx.z * cb + tc;
  const bool active = tp < rp && chunk * 8 < n;
  float s[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (active) {
    const long long r0 = (long long)slab * slab_rows;
    const long long r1 = r0 + slab_rows < rows_per_segment ? r0 + slab_rows : rows_per_segment;
    const T* base = x + (seg * rows_per_segment) * n + chunk * 8;
    long long r = r0 + tp;
    for (; r + 3 * rp < r1; r += 4 * rp) {          // 4 rows in flight per thread
      float v[4][8];
#pragma unroll
      for (int u = 0; u < 4; ++u) load8(base + (r + (long long)u * rp) * n, v[u]);
#pragma unroll
      for (int u = 0; u < 4; ++u)
#pragma unroll
        for (int e = 0; e < 8; ++e) s[e] += v[u][e];
    }
    for (; r < r1; r += rp) {
      float v[8];
      load8(base + r * n, v);
#pragma unroll
      for (int e = 0; e < 8; ++e) s[e] += v[e];
    }
  }
#pragma unroll
  for (int e = 0; e < 8; ++e) red[threadIdx.x][e] = s[e];
  __syncthreads();
  if (tp == 0 && active) {
    float o[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) o[e] = red[tc][e];
    for (int q = 1; q < rp; ++q)
#pragma unroll
      for (int e = 0; e < 8; ++e) o[e] += red[q * cb + tc][e];
    store8(partial + (seg * nslabs + slab) * n + chunk * 8, o);
  }
}

// out[seg][n] (+)= sum_slab partial[(seg*slabs + slab)][n].  64 columns x 4 slab groups per workgroup: a thread adds a contiguous
// quarter of the slabs in slab order, the four quarters meet in LDS in quarter order (fixed order: deterministic).
__global__ void __launch_bounds__(256)
colsum_fold_kernel(const float* __restrict__ partial, int nslabs, int n, long long total, int accumulate, float* __restrict__ out) {
  __shared__ float red[4][64];
  const int lane = threadIdx.x & 63, grp = threadIdx.x >> 6;
  const long long idx = (long long)blockIdx.x * 64 + lane;           // over seg*n
  float v = 0.f;
  if (idx < total) {
    const long long seg = idx / n;
    const int col = (int)(idx - seg * n);
    const int per = (nslabs + 3) / 4;
    const int s0 = grp * per, s1 = s0 + per < nslabs ? s0 + per : nslabs;
#pragma unroll 8
    for (int s = s0; s < s1; ++s) v += partial[(seg * nslabs + s) * n + col];
  }
  red[grp][lane] = v;
  __syncthreads();
  if (grp == 0 && idx < total) {
    float r = accumulate ? out[idx] : 0.f;
    r += ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
    out[idx] = r;
  }
}

// --------------------------------------------------------------------------------------- GLU / SiLU
// h is the pre-activation of w1_w2 in the packed (interleaved) column order: every 32 columns = 16 values then their 16
// gates (ldm_pack_weight glu_pairs).  out[m][16*blk + j] = h[m][32*blk + j] * act(h[m][32*blk + 16 + j]).
__device__ __forceinline__ float gelu_grad_f(float x) {   // d/dx [x * Phi(x)] = Phi(x) + x * phi(x)
  const float cdf = 0.5f * (1.0f + erff(x * 0.70710678118654752440f));
  return cdf + x * 0.39894228040143267794f * __expf(-0.5f * x * x);
}
__device__ __forceinline__ float silu_grad_f(float x) {
  const float sg = 1.0f / (1.0f + __expf(-x));
  return sg * (1.0f + x * (1.0f - sg));
}

template <typename T, bool GELU>
__global__ void __launch_bounds__(256)
glu_fwd_kernel(const T* __restrict__ h, T* __restrict__ out, int ff, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over (m, 8-column chunk of out)
  if (idx >= total) return;
  const int cpr = ff >> 3;
  const long long m = idx / cpr;
  const int ch = (int)(idx - m * cpr);
  const int col = ch * 8;                                   // output column
  const T* src = h + m * (2LL * ff) + (col >> 4) * 32 + (col & 15);
  float v[8], g[8];
  load8(src, v);
  load8(src + 16, g);
#pragma unroll
  for (int e = 0; e < 8; ++e) v[e] *= GELU ? gelu_erf_f(g[e]) : silu_f(g[e]);
  store8(out + m * ff + col, v);
}

template <typename T, bool GELU>
__global__ void __launch_bounds__(256)
glu_bwd_kernel(const T* __restrict__ h, const T* __restrict__ dout, T* __restrict__ dh, int ff, long long total) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= total) return;
  const int cpr = ff >> 3;
  const long long m = idx / cpr;
  const int ch = (int)(idx - m * cpr);
  const int col = ch * 8;
  const long long off = m * (2LL * ff) + (col >> 4) * 32 + (col & 15);
  float v[8], g[8], d[8], dv[8], dg[8];
  load8(h + off, v);
  load8(h + off + 16, g);
  load8(dout + m * ff + col, d);
#pragma unroll
  for (int e = 0; e < 8; ++e) {
    dv[e] = d[e] * (GELU ? gelu_erf_f(g[e]) : silu_f(g[e]));
    dg[e] = d[e] * v[e] * (GELU ? gelu_grad_f(g[e]) : silu_grad_f(g[e]));
  }
  store8(dh + off, dv);
  store8(dh + off + 16, dg);
}

// out[r][c] = a[r][lda..] + b[r][ldb..] (gradient accumulation where two paths meet and neither producer can fuse the
// add); the operands may be column slices of wider matrices (the two halves of a concat gradient)
template <typename T>
__global__ void __launch_bounds__(256)
add_kernel(const T* __restrict__ a, long long lda, const T* __restrict__ b, long long ldb, T* __restrict__ out, int cpr,
           long long nchunks) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= nchunks) return;
  const long long r = i / cpr;
  const int c = (int)(i - r * cpr) * 8;
  float x[8], y[8];
  load8(a + r * lda + c, x);
  load8(b + r * ldb + c, y);
#pragma unroll
  for (int e = 0; e < 8; ++e) x[e] += y[e];
  store8(out + i * 8, x);
}

// dx = dy * silu'(x)
template <typename T>
__global__ void __launch_bounds__(256)
silu_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, T* __restrict__ dx, long long n) {
  const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
  if (i >= n) return;
  Elem<T>::st(dx + i, Elem<T>::ld(dy + i) * silu_grad_f(Elem<T>::ld(x + i)));
}

// --------------------------------------------------------------------------------- GroupNorm backward
// y = act(gamma * xhat + beta), xhat = (x - mean) * rstd, act = SiLU or identity.  With dz = dy * act'(z):
//   A[b][c] = sum_pixels dz,  B[b][c] = sum_pixels dz * xhat            (stage 1 + fold)
//   dgamma[c] = sum_b B, dbeta[c] = sum_b A;  s1[b][g] = sum_{c in g} gamma A,  s2[b][g] = sum_{c in g} gamma B
//   dx = rstd * (dz * gamma - (s1 + xhat * s2) / n),  n = pixels * channels per group
constexpr int GB_THREADS = 512;    // 2560 channels (the widest concat) = 320 chunks, one per thread
constexpr int GB_MAX_SLABS = 64;

struct GnbGeom {
  int c0, c1, ctot, nchunks, ct, pix, cpg, groups, hw, slab_pixels;
};

template <typename T>
__device__ __forceinline__ const T* gnb_src(const T* a0, const T* a1, const GnbGeom& g, long long pixel, int c) {
  if (c < g.c0) return a0 + pixel * g.c0 + c;
  return a1 + pixel * g.c1 + (c - g.c0);
}

// dz of one 8-channel chunk of one pixel
__device__ __forceinline__ void gnb_dz(const float (&x)[8], const float (&dy)[8], const float (&sc)[8], const float (&sh)[8],
                                       const float (&mu)[8], const float (&rs)[8], int silu, float (&xh)[8], float (&dz)[8]) {
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    xh[j] = (x[j] - mu[j]) * rs[j];
    const float z = x[j] * sc[j] + sh[j];
    dz[j] = silu ? dy[j] * silu_grad_f(z) : dy[j];
  }
}

// one 8-channel chunk per thread (C <= 8 * GB_THREADS, checked on the host): thread (tp, tc) walks pixels tp, tp+pix, ...
template <typename T>
__global__ void __launch_bounds__(GB_THREADS)
gn_bwd_partial_kernel(const T* __restrict__ a0, const T* __restrict__ a1, const T* __restrict__ dy, GnbGeom g,
                      const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                      float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);   // [pix][nchunks*8][2]
  const int b = blockIdx.y, slab = blockIdx.x, nslabs = gridDim.x;
  const int tid = threadIdx.x;
  const int tp = tid / g.ct, tc = tid - tp * g.ct;
  const bool active = tp < g.pix && tc < g.nchunks;
  float sA[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f}, sB[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  if (active) {
    float sc[8], sh[8], mu[8], rs[8];
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      const int c = tc * 8 + j;
      const float* st = stats + ((size_t)b * g.groups + c / g.cpg) * 2;
      mu[j] = st[0]; rs[j] = st[1];
      sc[j] = st[1] * gamma[c];
      sh[j] = beta[c] - st[0] * sc[j];
    }
    const int p_begin = slab * g.slab_pixels;
    const int p_end = min(p_begin + g.slab_pixels, g.hw);
    for (int p = p_begin + tp; p < p_end; p += g.pix) {
      const long long pixel = (long long)b * g.hw + p;
      float x[8], d[8], xh[8], dz[8];
      load8(gnb_src(a0, a1, g, pixel, tc * 8), x);
      load8(dy + pixel * g.ctot + tc * 8, d);
      gnb_dz(x, d, sc, sh, mu, rs, silu, xh, dz);
#pragma unroll
      for (int j = 0; j < 8; ++j) { sA[j] += dz[j]; sB[j] += dz[j] * xh[j]; }
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      red[((size_t)tp * g.ctot + tc * 8 + j) * 2] = sA[j];
      red[((size_t)tp * g.ctot + tc * 8 + j) * 2 + 1] = sB[j];
    }
  }
  __syncthreads();
  for (int c = tid; c < g.ctot; c += GB_THREADS) {
    float a = 0.f, q = 0.f;
    for (int r = 0; r < g.pix; ++r) {
      a += red[((size_t)r * g.ctot + c) * 2];
      q += red[((size_t)r * g.ctot + c) * 2 + 1];
    }
    float* o = partial + (((size_t)b * nslabs + slab) * g.ctot + c) * 2;
    o[0] = a;
    o[1] = q;
  }
}

// One workgroup per (sample, range of groups): fold the slabs per channel (-> ab[b][c][2]), then the group sums
// (-> gsum[b][g][2]).  gridDim.y ranges of whole groups keep 4x more workgroups busy than one per sample; four slabs are
// in flight per thread.  Fixed summation order.
__global__ void __launch_bounds__(256)
gn_bwd_fold_kernel(const float* __restrict__ partial, int nslabs, int ctot, int groups, const float* __restrict__ gamma,
                   float* __restrict__ ab, float* __restrict__ gsum) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* ga = reinterpret_cast<float*>(smem_raw);    // [channels of this range][2] : gamma * A, gamma * B
  const int b = blockIdx.x;
  const int cpg = ctot / groups;
  const int gper = groups / gridDim.y;               // whole groups per workgroup (host: groups % gridDim.y == 0)
  const int g0 = blockIdx.y * gper;
  const int c_begin = g0 * cpg, c_end = (g0 + gper) * cpg;
  const float2* part2 = reinterpret_cast<const float2*>(partial);
  for (int c = c_begin + threadIdx.x; c < c_end; c += 256) {
    float a = 0.f, q = 0.f;
    int s = 0;
    for (; s + 3 < nslabs; s += 4) {
      const float2 v0 = part2[((size_t)b * nslabs + s) * ctot + c];
      const float2 v1 = part2[((size_t)b * nslabs + s + 1) * ctot + c];
      const float2 v2 = part2[((size_t)b * nslabs + s + 2) * ctot + c];
      const float2 v3 = part2[((size_t)b * nslabs + s + 3) * ctot + c];
      a = (((a + v0.x) + v1.x) + v2.x) + v3.x;
      q = (((q + v0.y) + v1.y) + v2.y) + v3.y;
    }
    for (; s < nslabs; ++s) {
      const float2 v = part2[((size_t)b * nslabs + s) * ctot + c];
      a += v.x;
      q += v.y;
    }
    ab[((size_t)b * ctot + c) * 2] = a;
    ab[((size_t)b * ctot + c) * 2 + 1] = q;
    ga[2 * (c - c_begin)] = gamma[c] * a;
    ga[2 * (c - c_begin) + 1] = gamma[c] * q;
  }
  __syncthreads();
  for (int g = threadIdx.x; g < gper; g += 256) {
    float s1 = 0.f, s2 = 0.f;
    for (int c = g * cpg; c < (g + 1) * cpg; ++c) { s1 += ga[2 * c]; s2 += ga[2 * c + 1]; }
    gsum[((size_t)b * groups + g0 + g) * 2] = s1;
    gsum[((size_t)b * groups + g0 + g) * 2 + 1] = s2;
  }
}

// dgamma[c] (+)= sum_b ab[b][c][1], dbeta[c] (+)= sum_b ab[b][c][0]
__global__ void __launch_bounds__(256)
gn_bwd_param_kernel(const float* __restrict__ ab, int batch, int ctot, int accumulate, float* __restrict__ dgamma,
                    float* __restrict__ dbeta) {
  const int c = blockIdx.x * 256 + threadIdx.x;
  if (c >= ctot) return;
  float a = 0.f, q = 0.f;
  for (int b = 0; b < batch; ++b) {
    a += ab[((size_t)b * ctot + c) * 2];
    q += ab[((size_t)b * ctot + c) * 2 + 1];
  }
  dgamma[c] = (accumulate ? dgamma[c] : 0.f) + q;
  dbeta[c] = (accumulate ? dbeta[c] : 0.f) + a;
}

template <typename T>
__global__ void __launch_bounds__(GB_THREADS)
gn_bwd_apply_kernel(const T* __restrict__ a0, const T* __restrict__ a1, const T* __restrict__ dy, GnbGeom g,
                    const float* __restrict__ stats, const float* __restrict__ gamma, const float* __restrict__ beta, int silu,
                    const float* __restrict__ gsum, const T* __restrict__ add, T* __restrict__ dx, T* __restrict__ dx1) {
  const int b = blockIdx.y, slab = blockIdx.x;
  const int tid = threadIdx.x;
  const int tp = tid / g.ct, tc = tid - tp * g.ct;
  if (tp >= g.pix || tc >= g.nchunks) return;
  float sc[8], sh[8], mu[8], rs[8], gm[8], k1[8], k2[8];
  const float inv_n = 1.0f / ((float)g.hw * (float)g.cpg);
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int c = tc * 8 + j;
    const int grp = c / g.cpg;
    const float* st = stats + ((size_t)b * g.groups + grp) * 2;
    mu[j] = st[0]; rs[j] = st[1];
    gm[j] = gamma[c];
    sc[j] = st[1] * gm[j];
    sh[j] = beta[c] - st[0] * sc[j];
    k1[j] = gsum[((size_t)b * g.groups + grp) * 2] * inv_n;
    k2[j] = gsum[((size_t)b * g.groups + grp) * 2 + 1] * inv_n;
  }
  const int p_begin = slab * g.slab_pixels;
  const int p_end = min(p_begin + g.slab_pixels, g.hw);
  for (int p = p_begin + tp; p < p_end; p += g.pix) {
    const long long pixel = (long long)b * g.hw + p;
    float x[8], d[8], xh[8], dz[8], o[8];
    load8(gnb_src(a0, a1, g, pixel, tc * 8), x);
    load8(dy + pixel * g.ctot + tc * 8, d);
    gnb_dz(x, d, sc, sh, mu, rs, silu, xh, dz);
#pragma unroll
    for (int j = 0; j < 8; ++j) o[j] = rs[j] * (dz[j] * gm[j] - (k1[j] + xh[j] * k2[j]));
    if (add != nullptr) {
      float a[8];
      load8(add + pixel * g.ctot + tc * 8, a);
#pragma unroll
      for (int j = 0; j < 8; ++j) o[j] += a[j];
    }
    // dx1 given: the two sources of the concat get their own dense gradients (c0 resp. c1 channels per pixel)
    if (dx1 == nullptr) store8(dx + pixel * g.ctot + tc * 8, o);
    else if (tc * 8 < g.c0) store8(dx + pixel * g.c0 + tc * 8, o);
    else store8(dx1 + pixel * g.c1 + (tc * 8 - g.c0), o);
  }
}

// ---------------------------------------------------------------------------------- LayerNorm backward
// One wave per row, rows grid-strided; a lane keeps CPL chunks of the row.  dx = rstd * (dy*gamma - m1 - xhat*m2)
// (+ add); the wave accumulates dy*xhat and dy per column over its rows; the workgroup's four waves are added in LDS and
// written to colpart[workgroup][cols][2].
template <typename T, int CPL>
__global__ void __launch_bounds__(256)
ln_bwd_kernel(const T* __restrict__ x, const T* __restrict__ dy, long long rows, int cols, float eps, const float* __restrict__ gamma,
              const T* __restrict__ add, T* __restrict__ dx, float* __restrict__ colpart) {
  const int lane = threadIdx.x & 63;
  const long long wave = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);
  const long long nwaves = (long long)gridDim.x * 4;
  const int nchunks = cols >> 3;
  float g[CPL][8], sg[CPL][8], sb[CPL][8];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int cc = lane + k * 64;
#pragma unroll
    for (int j = 0; j < 8; ++j) { g[k][j] = 0.f; sg[k][j] = 0.f; sb[k][j] = 0.f; }
    if (cc < nchunks) load8(gamma + cc * 8, g[k]);
  }
  const float inv_n = 1.0f / (float)cols;
  for (long long row = wave; row < rows; row += nwaves) {
    float xv[CPL][8], dv[CPL][8];
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = lane + k * 64;
      if (cc < nchunks) {
        load8(x + row * cols + cc * 8, xv[k]);
        load8(dy + row * cols + cc * 8, dv[k]);
      } else {
#pragma unroll
        for (int j = 0; j < 8; ++j) { xv[k][j] = 0.f; dv[k][j] = 0.f; }
      }
#pragma unroll
      for (int j = 0; j < 8; ++j) s += xv[k][j];
    }
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
      if (lane + k * 64 < nchunks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = xv[k][j] - mean; q += d * d; }
      }
    const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
    float m1 = 0.f, m2 = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) {
        const float xh = (xv[k][j] - mean) * rstd;
        const float dg = dv[k][j] * g[k][j];
        m1 += dg;
        m2 += dg * xh;
        sg[k][j] += dv[k][j] * xh;
        sb[k][j] += dv[k][j];
        xv[k][j] = xh;            // keep xhat
        dv[k][j] = dg;            // keep dy*gamma
      }
    m1 = wave_sum(m1) * inv_n;
    m2 = wave_sum(m2) * inv_n;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = lane + k * 64;
      if (cc < nchunks) {
        float o[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = rstd * (dv[k][j] - m1 - xv[k][j] * m2);
        if (add != nullptr) {
          float a[8];
          load8(add + row * cols + cc * 8, a);
#pragma unroll
          for (int j = 0; j < 8; ++j) o[j] += a[j];
        }
        store8(dx + row * cols + cc * 8, o);
      }
    }
  }
  // The four waves of the workgroup meet in LDS (waves 1..3 store, wave 0 adds them in wave order: fixed order) and ONE row of
  // column partials per workgroup goes out: a quarter of the rows ln_bwd_param_kernel has to read (it ran 23 us per call on
  // 4,096 rows of partials).
  extern __shared__ __attribute__((aligned(16))) unsigned char ln_smem[];
  float* red = reinterpret_cast<float*>(ln_smem);                    // [3][cols][2]
  const int wv = threadIdx.x >> 6;
  if (wv > 0) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = lane + k * 64;
      if (cc < nchunks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[((size_t)(wv - 1) * cols + cc * 8 + j) * 2] = sg[k][j];
          red[((size_t)(wv - 1) * cols + cc * 8 + j) * 2 + 1] = sb[k][j];
        }
      }
    }
  }
  __syncthreads();
  if (wv == 0) {
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = lane + k * 64;
      if (cc < nchunks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float a = sg[k][j], b = sb[k][j];
          for (int w = 0; w < 3; ++w) {
            a += red[((size_t)w * cols + cc * 8 + j) * 2];
            b += red[((size_t)w * cols + cc * 8 + j) * 2 + 1];
          }
          colpart[((size_t)blockIdx.x * cols + cc * 8 + j) * 2] = a;
          colpart[((size_t)blockIdx.x * cols + cc * 8 + j) * 2 + 1] = b;
        }
      }
    }
  }
}

// dgamma[c] (+)= sum_w colpart[w][c][0], dbeta[c] (+)= sum_w colpart[w][c][1].  Workgroup = 16 columns x 16 row-lanes:
// every load instruction covers whole 128-byte lines of colpart, four rows in flight per thread; fixed summation order.
__global__ void __launch_bounds__(256)
ln_bwd_param_kernel(const float* __restrict__ colpart, long long nwaves, int cols, int accumulate, float* __restrict__ dgamma,
                    float* __restrict__ dbeta) {
  __shared__ float2 red[16][17];
  const int cl = threadIdx.x & 15, wl = threadIdx.x >> 4;
  const int c = blockIdx.x * 16 + cl;
  const float2* part2 = reinterpret_cast<const float2*>(colpart);
  float a = 0.f, q = 0.f;
  if (c < cols) {
    long long w = wl;
    for (; w + 48 < nwaves; w += 64) {
      const float2 v0 = part2[(size_t)w * cols + c];
      const float2 v1 = part2[(size_t)(w + 16) * cols + c];
      const float2 v2 = part2[(size_t)(w + 32) * cols + c];
      const float2 v3 = part2[(size_t)(w + 48) * cols + c];
      a = (((a + v0.x) + v1.x) + v2.x) + v3.x;
      q = (((q + v0.y) + v1.y) + v2.y) + v3.y;
    }
    for (; w < nwaves; w += 16) {
      const float2 v = part2[(size_t)w * cols + c];
      a += v.x;
      q += v.y;
    }
  }
  red[wl][cl] = make_float2(a, q);
  __syncthreads();
  if (wl == 0 && c < cols) {
    float sa = 0.f, sq = 0.f;
#pragma unroll
    for (int k = 0; k < 16; ++k) { sa += red[k][cl].x; sq += red[k][cl].y; }
    dgamma[c] = (accumulate ? dgamma[c] : 0.f) + sa;
    dbeta[c] = (accumulate ? dbeta[c] : 0.f) + sq;
  }
}

static int gnb_geom(int c0, int c1, int hw, int groups, int batch, GnbGeom& g, int& nslabs, const char* who) {
  LDM_REQUIRE(c0 >= 8 && c1 >= 0 && c0 % 8 == 0 && c1 % 8 == 0, "%s: channel counts must be multiples of 8 (c0=%d c1=%d)", who, c0, c1);
  g.c0 = c0; g.c1 = c1; g.ctot = c0 + c1;
  LDM_REQUIRE(groups >= 1 && g.ctot % groups == 0, "%s: %d channels not divisible by %d groups", who, g.ctot, groups);
  g.nchunks = g.ctot / 8;
  LDM_REQUIRE(g.nchunks <= GB_THREADS, "%s: more than %d channels (%d)", who, 8 * GB_THREADS, g.ctot);
  g.ct = g.nchunks;
  g.pix = GB_THREADS / g.ct;
  g.cpg = g.ctot / groups;
  g.groups = groups;
  g.hw = hw;
  int want = (2048 + batch - 1) / (batch > 0 ? batch : 1);
  int by_work = (hw + 4 * g.pix - 1) / (4 * g.pix);
  nslabs = want < by_work ? want : by_work;
  if (nslabs > GB_MAX_SLABS) nslabs = GB_MAX_SLABS;
  if (nslabs < 1) nslabs = 1;
  g.slab_pixels = (hw + nslabs - 1) / nslabs;
  nslabs = (hw + g.slab_pixels - 1) / g.slab_pixels;
  return LDM_OK;
}

constexpr int LNB_BLOCKS = 1024;   // 4096 waves walk the rows

}  // namespace ldm

using namespace ldm;

extern "C" int ldm_sum2x2(const void* x, const void* residual, void* out, int dtype, int batch, int hout, int wout, int c, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_sum2x2: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && hout >= 1 && wout >= 1 && c >= 8 && c % 8 == 0, "ldm_sum2x2: bad sizes");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(x && out && aligned16(x) && aligned16(out) && aligned16(residual), "ldm_sum2x2: null or unaligned pointer");
  const long long total = (long long)batch * hout * wout * (c / 8);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((sum2x2_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)residual, (bf16_t*)out, hout, wout, c, total);
  else
    hipLaunchKernelGGL((sum2x2_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (const float*)residual, (float*)out, hout, wout, c, total);
  LDM_HIP_CHECK_LAUNCH("ldm_sum2x2");
  return LDM_OK;
}

extern "C" long long ldm_colsum_workspace_floats(long long segments, int n) { return segments * CS_MAX_SLABS * (long long)n; }

extern "C" int ldm_colsum(const void* x, int dtype, long long segments, long long rows_per_segment, int n, float* out,
                          int accumulate, float* workspace, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_colsum: bad dtype %d", dtype);
  LDM_REQUIRE(segments >= 0 && rows_per_segment >= 1 && n >= 8 && n % 8 == 0, "ldm_colsum: bad sizes");
  if (segments == 0) return LDM_OK;
  LDM_REQUIRE(segments <= 65535, "ldm_colsum: more than 65535 segments");
  LDM_REQUIRE(x && out && workspace && aligned16(x) && aligned16(workspace), "ldm_colsum: null or unaligned pointer");
  long long want = (4096 + segments - 1) / segments;
  long long by_work = (rows_per_segment + 63) / 64;
  int nslabs = (int)(want < by_work ? want : by_work);
  if (nslabs > CS_MAX_SLABS) nslabs = CS_MAX_SLABS;
  if (nslabs < 1) nslabs = 1;
  const int slab_rows = (int)((rows_per_segment + nslabs - 1) / nslabs);
  nslabs = (int)((rows_per_segment + slab_rows - 1) / slab_rows);
  hipStream_t st = (hipStream_t)stream;
  const int chunks = n / 8;
  const int cb = chunks >= 64 ? 64 : chunks;       // 8-column chunks per workgroup; 256 / cb rows per pass
  const dim3 grid(nslabs, (unsigned)segments, (chunks + cb - 1) / cb);
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((colsum_partial_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, rows_per_segment, n, slab_rows, cb, workspace);
  else
    hipLaunchKernelGGL((colsum_partial_kernel<float>), grid, dim3(256), 0, st, (const float*)x, rows_per_segment, n, slab_rows, cb, workspace);
  LDM_HIP_CHECK_LAUNCH("ldm_colsum(partial)");
  const long long total = segments * n;
  hipLaunchKernelGGL(colsum_fold_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, st, workspace, nslabs, n, total, accumulate, out);
  LDM_HIP_CHECK_LAUNCH("ldm_colsum(fold)");
  return LDM_OK;
}

extern "C" int ldm_glu_forward(const void* h, void* out, int dtype, long long rows, int ff, int gelu, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_glu_forward: bad dtype %d", dtype);
  LDM_REQUIRE(rows >= 0 && ff >= 16 && ff % 16 == 0, "ldm_glu_forward: ff must be a multiple of 16 (got %d)", ff);
  if (rows == 0) return LDM_OK;
  LDM_REQUIRE(h && out && aligned16(h) && aligned16(out), "ldm_glu_forward: null or unaligned pointer");
  const long long total = rows * (ff / 8);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == LDM_BF16) {
    if (gelu) hipLaunchKernelGGL((glu_fwd_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)h, (bf16_t*)out, ff, total);
    else hipLaunchKernelGGL((glu_fwd_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)h, (bf16_t*)out, ff, total);
  } else {
    if (gelu) hipLaunchKernelGGL((glu_fwd_kernel<float, true>), grid, dim3(256), 0, st, (const float*)h, (float*)out, ff, total);
    else hipLaunchKernelGGL((glu_fwd_kernel<float, false>), grid, dim3(256), 0, st, (const float*)h, (float*)out, ff, total);
  }
  LDM_HIP_CHECK_LAUNCH("ldm_glu_forward");
  return LDM_OK;
}

extern "C" int ldm_glu_backward(const void* h, const void* dout, void* dh, int dtype, long long rows, int ff, int gelu, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_glu_backward: bad dtype %d", dtype);
  LDM_REQUIRE(rows >= 0 && ff >= 16 && ff % 16 == 0, "ldm_glu_backward: ff must be a multiple of 16 (got %d)", ff);
  if (rows == 0) return LDM_OK;
  LDM_REQUIRE(h && dout && dh && aligned16(h) && aligned16(dout) && aligned16(dh), "ldm_glu_backward: null or unaligned pointer");
  const long long total = rows * (ff / 8);
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((total + 255) / 256));
  if (dtype == LDM_BF16) {
    if (gelu) hipLaunchKernelGGL((glu_bwd_kernel<bf16_t, true>), grid, dim3(256), 0, st, (const bf16_t*)h, (const bf16_t*)dout, (bf16_t*)dh, ff, total);
    else hipLaunchKernelGGL((glu_bwd_kernel<bf16_t, false>), grid, dim3(256), 0, st, (const bf16_t*)h, (const bf16_t*)dout, (bf16_t*)dh, ff, total);
  } else {
    if (gelu) hipLaunchKernelGGL((glu_bwd_kernel<float, true>), grid, dim3(256), 0, st, (const float*)h, (const float*)dout, (float*)dh, ff, total);
    else hipLaunchKernelGGL((glu_bwd_kernel<float, false>), grid, dim3(256), 0, st, (const float*)h, (const float*)dout, (float*)dh, ff, total);
  }
  LDM_HIP_CHECK_LAUNCH("ldm_glu_backward");
  return LDM_OK;
}

extern "C" int ldm_add(const void* a, long long lda, const void* b, long long ldb, void* out, int dtype, long long rows, int cols,
                       void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_add: bad dtype %d", dtype);
  LDM_REQUIRE(rows >= 0 && cols >= 8 && cols % 8 == 0 && lda >= cols && ldb >= cols && lda % 8 == 0 && ldb % 8 == 0,
              "ldm_add: cols and row strides must be multiples of 8 (rows=%lld cols=%d lda=%lld ldb=%lld)", rows, cols, lda, ldb);
  if (rows == 0) return LDM_OK;
  LDM_REQUIRE(a && b && out && aligned16(a) && aligned16(b) && aligned16(out), "ldm_add: null or unaligned pointer");
  hipStream_t st = (hipStream_t)stream;
  const long long chunks = rows * (cols / 8);
  const dim3 grid((unsigned)((chunks + 255) / 256));
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((add_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)a, lda, (const bf16_t*)b, ldb, (bf16_t*)out, cols / 8, chunks);
  else
    hipLaunchKernelGGL((add_kernel<float>), grid, dim3(256), 0, st, (const float*)a, lda, (const float*)b, ldb, (float*)out, cols / 8, chunks);
  LDM_HIP_CHECK_LAUNCH("ldm_add");
  return LDM_OK;
}

extern "C" int ldm_silu_backward(const void* x, const void* dy, void* dx, int dtype, long long n, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_silu_backward: bad dtype %d", dtype);
  if (n <= 0) return LDM_OK;
  LDM_REQUIRE(x && dy && dx, "ldm_silu_backward: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const dim3 grid((unsigned)((n + 255) / 256));
  if (dtype == LDM_BF16) hipLaunchKernelGGL((silu_bwd_kernel<bf16_t>), grid, dim3(256), 0, st, (const bf16_t*)x, (const bf16_t*)dy, (bf16_t*)dx, n);
  else hipLaunchKernelGGL((silu_bwd_kernel<float>), grid, dim3(256), 0, st, (const float*)x, (const float*)dy, (float*)dx, n);
  LDM_HIP_CHECK_LAUNCH("ldm_silu_backward");
  return LDM_OK;
}

extern "C" long long ldm_groupnorm_backward_workspace_floats(int batch, int channels, int groups) {
  return (long long)batch * GB_MAX_SLABS * channels * 2 + (long long)batch * channels * 2 + (long long)batch * groups * 2;
}

extern "C" int ldm_groupnorm_backward(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw, int groups,
                                      const float* stats, const float* gamma, const float* beta, int silu, const void* dy,
                                      const void* add, void* dx, void* dx1, float* dgamma, float* dbeta, int accumulate,
                                      float* workspace, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_groupnorm_backward: bad dtype %d", dtype);
  LDM_REQUIRE(dx1 == nullptr || (c1 > 0 && aligned16(dx1)), "ldm_groupnorm_backward: dx1 needs a second source and 16-byte alignment");
  LDM_REQUIRE(batch >= 0 && hw >= 1, "ldm_groupnorm_backward: bad sizes");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(batch <= 65535, "ldm_groupnorm_backward: batch %d exceeds 65535", batch);
  LDM_REQUIRE(a0 && stats && gamma && beta && dy && dx && dgamma && dbeta && workspace && (c1 == 0 || a1),
              "ldm_groupnorm_backward: null pointer");
  GnbGeom g;
  int nslabs;
  int rc = gnb_geom(c0, c1, hw, groups, batch, g, nslabs, "ldm_groupnorm_backward");
  if (rc) return rc;
  LDM_REQUIRE(aligned16(a0) && aligned16(dy) && aligned16(dx) && aligned16(add) && (c1 == 0 || aligned16(a1)),
              "ldm_groupnorm_backward: buffers must be 16-byte aligned");
  const size_t smem = (size_t)g.pix * g.ctot * 2 * sizeof(float);
  LDM_REQUIRE(smem <= 64 * 1024, "ldm_groupnorm_backward: %d channels need %zu bytes of LDS", g.ctot, smem);
  float* partial = workspace;
  float* ab = partial + (size_t)batch * GB_MAX_SLABS * g.ctot * 2;
  float* gsum = ab + (size_t)batch * g.ctot * 2;
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nslabs, batch);
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((gn_bwd_partial_kernel<bf16_t>), grid, dim3(GB_THREADS), smem, st, (const bf16_t*)a0, (const bf16_t*)a1,
                       (const bf16_t*)dy, g, stats, gamma, beta, silu, partial);
  else
    hipLaunchKernelGGL((gn_bwd_partial_kernel<float>), grid, dim3(GB_THREADS), smem, st, (const float*)a0, (const float*)a1,
                       (const float*)dy, g, stats, gamma, beta, silu, partial);
  LDM_HIP_CHECK_LAUNCH("ldm_groupnorm_backward(partial)");
  const int gsplit = groups % 4 == 0 ? 4 : 1;
  hipLaunchKernelGGL(gn_bwd_fold_kernel, dim3(batch, gsplit), dim3(256), (size_t)(g.ctot / gsplit) * 2 * sizeof(float), st, partial,
                     nslabs, g.ctot, groups, gamma, ab, gsum);
  LDM_HIP_CHECK_LAUNCH("ldm_groupnorm_backward(fold)");
  hipLaunchKernelGGL(gn_bwd_param_kernel, dim3((g.ctot + 255) / 256), dim3(256), 0, st, ab, batch, g.ctot, accumulate, dgamma, dbeta);
  LDM_HIP_CHECK_LAUNCH("ldm_groupnorm_backward(param)");
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((gn_bwd_apply_kernel<bf16_t>), grid, dim3(GB_THREADS), 0, st, (const bf16_t*)a0, (const bf16_t*)a1,
                       (const bf16_t*)dy, g, stats, gamma, beta, silu, gsum, (const bf16_t*)add, (bf16_t*)dx, (bf16_t*)dx1);
  else
    hipLaunchKernelGGL((gn_bwd_apply_kernel<float>), grid, dim3(GB_THREADS), 0, st, (const float*)a0, (const float*)a1,
                       (const float*)dy, g, stats, gamma, beta, silu, gsum, (const float*)add, (float*)dx, (float*)dx1);
  LDM_HIP_CHECK_LAUNCH("ldm_groupnorm_backward(apply)");
  return LDM_OK;
}

extern "C" long long ldm_layernorm_backward_workspace_floats(int cols) { return (long long)LNB_BLOCKS * 4 * cols * 2; }

extern "C" int ldm_layernorm_backward(const void* x, const void* dy, int dtype, long long rows, int cols, float eps, const float* gamma,
                                      const void* add, void* dx, float* dgamma, float* dbeta, int accumulate, float* workspace,
                                      void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_layernorm_backward: bad dtype %d", dtype);
  LDM_REQUIRE(rows >= 0 && cols >= 8 && cols % 8 == 0 && cols <= 2048, "ldm_layernorm_backward: cols must be a multiple of 8 in [8, 2048] (got %d)", cols);
  if (rows == 0) return LDM_OK;
  LDM_REQUIRE(x && dy && gamma && dx && dgamma && dbeta && workspace, "ldm_layernorm_backward: null pointer");
  LDM_REQUIRE(aligned16(x) && aligned16(dy) && aligned16(dx) && aligned16(add) && aligned16(gamma), "ldm_layernorm_backward: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  long long blocks = (rows + 3) / 4;
  if (blocks > LNB_BLOCKS) blocks = LNB_BLOCKS;
  const dim3 grid((unsigned)blocks);
  const size_t lds = (size_t)3 * cols * 2 * sizeof(float);          // <= 48 KB (cols <= 2048)
#define LDM_LNB(T, CPL) hipLaunchKernelGGL((ln_bwd_kernel<T, CPL>), grid, dim3(256), lds, st, (const T*)x, (const T*)dy, rows, cols, eps, gamma, (const T*)add, (T*)dx, workspace)
  if (dtype == LDM_BF16) {
    if (cols <= 512) LDM_LNB(bf16_t, 1); else if (cols <= 1024) LDM_LNB(bf16_t, 2); else LDM_LNB(bf16_t, 4);
  } else {
    if (cols <= 512) LDM_LNB(float, 1); else if (cols <= 1024) LDM_LNB(float, 2); else LDM_LNB(float, 4);
  }
#undef LDM_LNB
  LDM_HIP_CHECK_LAUNCH("ldm_layernorm_backward");
  hipLaunchKernelGGL(ln_bwd_param_kernel, dim3((cols + 15) / 16), dim3(256), 0, st, workspace, blocks, cols, accumulate, dgamma, dbeta);
  LDM_HIP_CHECK_LAUNCH("ldm_layernorm_backward(param)");
  return LDM_OK;
}
