// Weight gradient of the implicit-GEMM conv / linear layer on the gfx950 matrix cores.
//
//   dW[n][k] = sum_m dY[m][n] * A[m][k]        m = (sample, oh, ow), k = (kh, kw, ci), A = im2col(activation)
//
// The contraction runs over the ROWS of both operands, so both are staged row-major (one row = one output pixel) exactly
// as the forward kernel stages A -- LDS-DMA through buffer descriptors, im2col / stride / nearest-2x upsample / two-source
// concat folded into the per-lane source offset, padding zero-filled by the range check -- and the MFMA fragments are
// read TRANSPOSED from LDS with ds_read_b64_tr_b16 (the same read the attention kernel uses for V^T).  Both operands
// come through the same transposed read, so the permutation it applies to the contraction index cancels.
//
// One workgroup = 4 waves = a 128 (n) x 128 (k) tile of dW; it walks its share of the pixels 64 rows per step,
// double-buffered.  Every 16-byte chunk of a k-tile (8 consecutive k) is addressed on its own, so a k-tile may span
// taps and the concat boundary: channel counts only need to be multiples of 8.  The pixels are split over blockIdx.y (layers with few output tiles
// have the most pixels); every split writes an fp32 partial tile and wgrad_fold_kernel adds the splits in a fixed order
// and stores the checkpoint layout (OIHW), optionally accumulating into an existing gradient.  Deterministic.
#include "common.h"

#include <type_traits>

namespace ldm {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct WgradParams {
  const void* a0;
  const void* a1;
  const void* dy;
  float* partial;          // [splits][N][K]
  int c0, c1, ctot;
  int batch, hin, win, hout, wout;
  int ksize, stride, upsample;
  int M, N, K;
  int tiles_n, tiles_k;
  int steps_per_split;     // 64-row steps per blockIdx.y
  int splits;
};

__device__ __forceinline__ void wg_dma16(const void* base, unsigned bytes, unsigned char* lds, unsigned voff, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

constexpr int WG_TN = 128, WG_TK = 128, WG_MS = 64;      // dW tile and rows per step
constexpr int WG_ROWB = 256;                              // bytes per LDS row of either image (128 bf16)
constexpr int WG_IMG = WG_MS * WG_ROWB;                   // 16 KiB per image
constexpr int WG_STAGE = 2 * WG_IMG;                      // dY image then A image

__global__ void __launch_bounds__(256, 2)
conv_wgrad_bf16_kernel(const WgradParams p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  constexpr unsigned OOB = 0x80000000u;
  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wn = wave >> 1, wk = wave & 1;                // wave's 64 x 64 quadrant of the dW tile
  const int tn = blockIdx.x % p.tiles_n, tk = blockIdx.x / p.tiles_n;
  const int n0 = tn * WG_TN, k0 = tk * WG_TK;
  const int step_begin = blockIdx.y * p.steps_per_split;
  const int total_steps = (p.M + WG_MS - 1) / WG_MS;
  const int step_end = min(step_begin + p.steps_per_split, total_steps);
  const int nsteps = step_end - step_begin;

  const unsigned bytes_a0 = (unsigned)p.batch * p.hin * p.win * p.c0 * 2u;
  const unsigned bytes_a1 = (unsigned)p.batch * p.hin * p.win * (p.c1 ? p.c1 : p.c0) * 2u;
  const unsigned bytes_dy = (unsigned)p.M * p.N * 2u;

  // ---- staging geometry.  A wave issues pieces j = 0..3 of each image per step; piece (wave*4 + j) fills rows
  // 4*(wave*4+j) .. +3 (64 lanes x 16 B = 4 rows of 256 B).  Lane -> (row in piece, physical chunk); the logical chunk
  // it must fetch is phys ^ (row & 7) (XOR swizzle: the transposed reads of 8 rows then hit 8 different bank groups).
  const int lrow = lane >> 4, pchunk = lane & 15;
  const int hlim = p.upsample ? p.hin * 2 : p.hin;
  const int wlim = p.upsample ? p.win * 2 : p.win;
  const int pad = p.ksize >> 1;
  // per piece parity (row & 7 = 4*(j&1) + lrow): the lane's logical chunk, hence its k-group, is fixed for the kernel
  int a_kh[2], a_kw[2], a_coff[2], a_cs[2];
  bool a_second[2], a_kok[2];
  unsigned dy_voff[2];
#pragma unroll
  for (int par = 0; par < 2; ++par) {
    const int lchunk = pchunk ^ ((4 * par + lrow) & 7);
    const int k = k0 + lchunk * 8;                        // first of the 8 k of this chunk
    a_kok[par] = k < p.K;
    const int kg = a_kok[par] ? k : 0;
    const int tap = kg / p.ctot;
    int c = kg - tap * p.ctot;
    a_kh[par] = tap / p.ksize;
    a_kw[par] = tap - a_kh[par] * p.ksize;
    a_second[par] = c >= p.c0;
    if (a_second[par]) c -= p.c0;
    a_coff[par] = c;
    a_cs[par] = a_second[par] ? p.c1 : p.c0;
    const int n = n0 + lchunk * 8;
    dy_voff[par] = n < p.N ? (unsigned)n * 2u : OOB;      // + row * N * 2 below
  }
  // rows of this lane's four pieces at the first step, as (sample, oh, ow); advanced by 64 rows per step
  int r_b[4], r_oh[4], r_ow[4];
  const int hw_out = p.hout * p.wout;
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const int m = step_begin * WG_MS + 4 * (wave * 4 + j) + lrow;
    const int b = m / hw_out, rem = m - b * hw_out;
    r_b[j] = b;
    r_oh[j] = rem / p.wout;
    r_ow[j] = rem - r_oh[j] * p.wout;
  }
  const int d_ow = WG_MS % p.wout, d_ohr = WG_MS / p.wout;
  const int d_oh = d_ohr % p.hout, d_b = d_ohr / p.hout;

  auto stage = [&](int buf, int step) {
    unsigned char* sdy = smem + buf * WG_STAGE;
    unsigned char* sa = sdy + WG_IMG;
    const unsigned row0 = (unsigned)step * WG_MS;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
      const int par = j & 1;
      const int piece = wave * 4 + j;
      const unsigned m = row0 + 4 * piece + lrow;
      // dY rows: plain matrix; rows past M fall outside the descriptor and read as zeros
      const unsigned dvo = dy_voff[par] == OOB ? OOB : (m < (unsigned)p.M ? m * (unsigned)p.N * 2u + dy_voff[par] : OOB);
      wg_dma16(p.dy, bytes_dy, sdy + piece * 1024, dvo, 0);
      // A rows: im2col gather of this lane's k-group
      int ih = r_oh[j] * p.stride - pad + a_kh[par], iw = r_ow[j] * p.stride - pad + a_kw[par];
      const bool ok = a_kok[par] && m < (unsigned)p.M && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
      if (p.upsample) { ih >>= 1; iw >>= 1; }
      const unsigned avo = ok ? ((unsigned)((r_b[j] * p.hin + ih) * p.win + iw) * a_cs[par] + a_coff[par]) * 2u : OOB;
      if (a_second[par]) wg_dma16(p.a1, bytes_a1, sa + piece * 1024, avo, 0);
      else wg_dma16(p.a0, bytes_a0, sa + piece * 1024, avo, 0);
      // advance this piece's rows by one step
      r_ow[j] += d_ow;
      if (r_ow[j] >= p.wout) { r_ow[j] -= p.wout; r_oh[j] += 1; }
      r_oh[j] += d_oh;
      r_b[j] += d_b;
      if (r_oh[j] >= p.hout) { r_oh[j] -= p.hout; r_b[j] += 1; }
    }
  };

  f32x4_t acc[4][4];
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  // transposed fragment of a 16-column tile `ct` (columns ct*16 .. +15) over the 32 rows of contraction block st
  auto frag = [&](const unsigned char* img, int ct, int st) -> bf16x8_t {
    const int row = 32 * st + 4 * fq + tq;
    const int lchunk = 2 * ct + (tp >> 1);
    const unsigned char* a0 = img + row * WG_ROWB + ((lchunk ^ (row & 7)) << 4) + ((tp & 1) << 3);
    const unsigned char* a1 = a0 + 16 * WG_ROWB;          // (row + 16) & 7 == row & 7
    const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)a0);
    const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)a1);
    return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
  };
  auto compute = [&](int buf) {
    const unsigned char* sdy = smem + buf * WG_STAGE;
    const unsigned char* sa = sdy + WG_IMG;
#pragma unroll
    for (int st = 0; st < 2; ++st) {
      bf16x8_t fn[4], fk[4];
#pragma unroll
      for (int i = 0; i < 4; ++i) fn[i] = frag(sdy, wn * 4 + i, st);
#pragma unroll
      for (int j = 0; j < 4; ++j) fk[j] = frag(sa, wk * 4 + j, st);
#pragma unroll
      for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[i], fk[j], acc[i][j], 0, 0, 0);
    }
  };

  if (nsteps > 0) {
    stage(0, step_begin);
    __syncthreads();
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) stage((s + 1) & 1, step_begin + s + 1);
      compute(s & 1);
      __syncthreads();
    }
  }

  // ---- partial tile: C[row = n = fq*4 + r][col = k = fr]
  float* out = p.partial + (size_t)blockIdx.y * p.N * p.K;
#pragma unroll
  for (int i = 0; i < 4; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + (wn * 4 + i) * 16 + fq * 4 + r;
      if (n >= p.N) continue;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const int k = k0 + (wk * 4 + j) * 16 + fr;
        if (k < p.K) out[(size_t)n * p.K + k] = acc[i][j][r];
      }
    }
}

// dw[n][ci][tap] (+)= sum_s partial[s][n][tap*ctot + ci]   (fixed order; OIHW = the checkpoint layout)
__global__ void __launch_bounds__(256)
wgrad_fold_kernel(const float* __restrict__ partial, float* __restrict__ dw, int N, int K, int ctot, int taps, int splits,
                  int accumulate) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over dw elements, OIHW order
  if (idx >= (long long)N * K) return;
  const int n = (int)(idx / K), rem = (int)(idx - (long long)n * K);
  const int ci = rem / taps, tap = rem - ci * taps;
  const size_t src = (size_t)n * K + (size_t)tap * ctot + ci;
  float v = accumulate ? dw[idx] : 0.f;
  for (int s = 0; s < splits; ++s) v += partial[(size_t)s * N * K + src];
  dw[idx] = v;
}


// Plain-FMA weight gradient for either dtype and any channel count: the exact-float32 parity path.  One thread per
// dW element and pixel split; the same partial buffer and fold kernel as above.  Not a speed path.
constexpr int WGG_ROWS = 1024;   // pixels per split

template <typename T>
__global__ void __launch_bounds__(256)
conv_wgrad_generic_kernel(const WgradParams p) {
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;   // over [N][K]
  if (idx >= (long long)p.N * p.K) return;
  const int n = (int)(idx / p.K), k = (int)(idx - (long long)n * p.K);
  const int tap = k / p.ctot;
  int c = k - tap * p.ctot;
  const int kh = tap / p.ksize, kw = tap - kh * p.ksize;
  const bool second = c >= p.c0;
  if (second) c -= p.c0;
  const T* src = static_cast<const T*>(second ? p.a1 : p.a0);
  const int cs = second ? p.c1 : p.c0;
  const T* dy = static_cast<const T*>(p.dy);
  const int hlim = p.upsample ? p.hin * 2 : p.hin, wlim = p.upsample ? p.win * 2 : p.win;
  const int pad = p.ksize >> 1;
  const int m0 = blockIdx.y * WGG_ROWS, m1 = min(m0 + WGG_ROWS, p.M);
  const int hw_out = p.hout * p.wout;
  int b = m0 / hw_out, rem = m0 - b * hw_out;
  int oh = rem / p.wout, ow = rem - oh * p.wout;
  float acc = 0.f;
  for (int m = m0; m < m1; ++m) {
    int ih = oh * p.stride - pad + kh, iw = ow * p.stride - pad + kw;
    if (ih >= 0 && ih < hlim && iw >= 0 && iw < wlim) {
      if (p.upsample) { ih >>= 1; iw >>= 1; }
      acc += Elem<T>::ld(dy + (size_t)m * p.N + n) * Elem<T>::ld(src + ((size_t)(b * p.hin + ih) * p.win + iw) * cs + c);
    }
    if (++ow == p.wout) { ow = 0; if (++oh == p.hout) { oh = 0; ++b; } }
  }
  p.partial[(size_t)blockIdx.y * p.N * p.K + idx] = acc;
}

static bool wgrad_use_mfma(const LdmConvGemm* d) {
  return d->dtype == LDM_BF16 && (d->c0 + d->c1) % 8 == 0 && d->c0 % 8 == 0 && d->n % 8 == 0;
}

static void wgrad_plan(const LdmConvGemm* d, WgradParams& p) {
  p.c0 = d->c0; p.c1 = d->c1; p.ctot = d->c0 + d->c1;
  p.batch = d->batch; p.hin = d->hin; p.win = d->win; p.hout = d->hout; p.wout = d->wout;
  p.ksize = d->ksize; p.stride = d->stride; p.upsample = d->upsample;
  p.M = d->batch * d->hout * d->wout; p.N = d->n; p.K = d->ksize * d->ksize * p.ctot;
  p.tiles_n = (p.N + WG_TN - 1) / WG_TN;
  p.tiles_k = (p.K + WG_TK - 1) / WG_TK;
  if (!wgrad_use_mfma(d)) {
    p.splits = (p.M + WGG_ROWS - 1) / WGG_ROWS;
    if (p.splits < 1) p.splits = 1;
    p.steps_per_split = 0;
    return;
  }
  const int steps = (p.M + WG_MS - 1) / WG_MS;
  // enough workgroups for two per CU, at least 8 steps each
  int splits = (512 + p.tiles_n * p.tiles_k - 1) / (p.tiles_n * p.tiles_k);
  if (splits > steps / 8) splits = steps / 8;
  if (splits < 1) splits = 1;
  if (splits > 64) splits = 64;
  p.steps_per_split = (steps + splits - 1) / splits;
  p.splits = (steps + p.steps_per_split - 1) / p.steps_per_split;
}

}  // namespace ldm

using namespace ldm;

extern "C" long long ldm_conv_wgrad_workspace_bytes(const LdmConvGemm* d) {
  if (!d || d->batch <= 0) return 0;
  WgradParams p;
  wgrad_plan(d, p);
  return (long long)p.splits * p.N * p.K * 4;
}

extern "C" int ldm_conv_wgrad(const LdmConvGemm* d, const void* dy, float* dw, int accumulate, void* workspace,
                              long long workspace_bytes, void* stream) {
  LDM_REQUIRE(d != nullptr, "ldm_conv_wgrad: null descriptor");
  LDM_REQUIRE(dtype_ok(d->dtype), "ldm_conv_wgrad: bad dtype %d", d->dtype);
  LDM_REQUIRE(d->ksize == 1 || d->ksize == 3, "ldm_conv_wgrad: ksize must be 1 or 3 (got %d)", d->ksize);
  LDM_REQUIRE(d->stride == 1 || d->stride == 2, "ldm_conv_wgrad: stride must be 1 or 2 (got %d)", d->stride);
  LDM_REQUIRE(!(d->upsample && d->stride != 1), "ldm_conv_wgrad: upsample with stride 2 is not a thing");
  LDM_REQUIRE(d->batch >= 0 && d->hin >= 1 && d->win >= 1 && d->hout >= 1 && d->wout >= 1 && d->n >= 1 && d->c0 >= 1 && d->c1 >= 0,
              "ldm_conv_wgrad: bad sizes");
  const int pad = d->ksize / 2;
  const int hsrc = d->upsample ? 2 * d->hin : d->hin, wsrc = d->upsample ? 2 * d->win : d->win;
  LDM_REQUIRE(d->hout == (hsrc + 2 * pad - d->ksize) / d->stride + 1 && d->wout == (wsrc + 2 * pad - d->ksize) / d->stride + 1,
              "ldm_conv_wgrad: output extent %dx%d does not match input %dx%d (k=%d, stride=%d, upsample=%d)",
              d->hout, d->wout, d->hin, d->win, d->ksize, d->stride, d->upsample);
  LDM_REQUIRE(dw != nullptr, "ldm_conv_wgrad: null dw");
  WgradParams p;
  wgrad_plan(d, p);
  hipStream_t st = (hipStream_t)stream;
  if (d->batch == 0) {
    if (!accumulate) {
      hipError_t e = hipMemsetAsync(dw, 0, (size_t)p.N * p.K * 4, st);
      if (e != hipSuccess) { set_error("ldm_conv_wgrad: hipMemsetAsync: %s", hipGetErrorString(e)); return LDM_EHIP; }
    }
    return LDM_OK;
  }
  LDM_REQUIRE(d->a0 && dy && (d->c1 == 0 || d->a1), "ldm_conv_wgrad: null pointer");
  const bool mfma = wgrad_use_mfma(d);
  const long long pix = (long long)d->batch * d->hin * d->win;
  if (mfma) {
    LDM_REQUIRE(pix * d->c0 * 2 < (1LL << 31) && pix * d->c1 * 2 < (1LL << 31) && (long long)p.M * p.N * 2 < (1LL << 31),
                "ldm_conv_wgrad: an operand exceeds 2 GiB (32-bit buffer offsets)");
    LDM_REQUIRE(aligned16(d->a0) && (d->c1 == 0 || aligned16(d->a1)) && aligned16(dy), "ldm_conv_wgrad: operands must be 16-byte aligned");
  }
  LDM_REQUIRE(pix < (1LL << 31) && (long long)p.M < (1LL << 31) && p.splits <= 65535, "ldm_conv_wgrad: too many pixels");
  const long long need = (long long)p.splits * p.N * p.K * 4;
  LDM_REQUIRE(workspace != nullptr && workspace_bytes >= need, "ldm_conv_wgrad: workspace of %lld bytes needed (got %lld)", need,
              workspace_bytes);
  p.a0 = d->a0; p.a1 = d->c1 ? d->a1 : d->a0; p.dy = dy; p.partial = static_cast<float*>(workspace);
  if (mfma) {
    const int smem = 2 * WG_STAGE;
    static bool attr_done = false;
    if (!attr_done) {
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) { set_error("ldm_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
      attr_done = true;
    }
    dim3 grid(p.tiles_n * p.tiles_k, p.splits);
    hipLaunchKernelGGL(conv_wgrad_bf16_kernel, grid, dim3(256), smem, st, p);
  } else {
    dim3 grid((unsigned)(((long long)p.N * p.K + 255) / 256), p.splits);
    if (d->dtype == LDM_BF16) hipLaunchKernelGGL((conv_wgrad_generic_kernel<bf16_t>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_generic_kernel<float>), grid, dim3(256), 0, st, p);
  }
  LDM_HIP_CHECK_LAUNCH("ldm_conv_wgrad");
  const long long total = (long long)p.N * p.K;
  hipLaunchKernelGGL(wgrad_fold_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, st, p.partial, dw, p.N, p.K, p.ctot,
                     d->ksize * d->ksize, p.splits, accumulate);
  LDM_HIP_CHECK_LAUNCH("ldm_conv_wgrad(fold)");
  return LDM_OK;
}
