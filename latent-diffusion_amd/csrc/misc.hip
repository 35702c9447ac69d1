// Small kernels around the GEMM core: filter packing, sinusoidal timestep embedding, and the two
// degenerate convolutions of the UNet (conv_in: K = 36, final conv: N = 4) that do not fit an MFMA tile.
#include "common.h"

namespace ldm {

// dst[o'][kh][kw][ci] = src[o][ci][kh][kw]
template <typename T>
__global__ void __launch_bounds__(256)
pack_weight_kernel(const float* __restrict__ src, T* __restrict__ dst, int cout, int cin, int taps, int glu) {
  const int64_t total = (int64_t)cout * cin * taps;
  const int64_t stride = (int64_t)gridDim.x * blockDim.x;
  const int half = cout >> 1;
  for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const int ci = (int)(i % cin);
    const int64_t r = i / cin;
    const int tap = (int)(r % taps);
    const int op = (int)(r / taps);  // destination row
    int o = op;
    if (glu) {  // op = 32*blk + {0..15: value, 16..31: gate}
      const int blk = op >> 5, w = op & 31;
      o = w < 16 ? blk * 16 + w : half + blk * 16 + (w - 16);
    }
    Elem<T>::st(dst + i, src[((int64_t)o * cin + ci) * taps + tap]);
  }
}

// Both derived layouts of one filter from ONE coalesced read of the checkpoint tensor (training repacks every filter after
// every optimizer step: 859.5 M floats):
//   fwd  [o'][tap][ci]          (ldm_pack_weight's layout; o' = GLU-interleaved row order when glu)
//   dgrad[ci][taps-1-tap][o']   row pitch dgrad_ld (>= cout: several filters may share one fused dgrad matrix): the filter of
//                               the input-gradient pass -- in/out transposed, taps rotated by 180 degrees.
// One workgroup moves a 32 (o') x 32 (ci) x taps block through LDS; global runs are 32 x taps floats in, 32 elements out.
// VEC4 (bf16 only; cout, cin, dgrad_ld multiples of 4, destinations 8-byte aligned): a lane stores 4 packed elements.
// TAPS = 1 / 9 (0: any, from the argument): with the tap count known the trip counts are constants, so all of a thread's
// global loads (4 / 36) are in flight before the first LDS store -- with a run-time count the loop was one load, one
// store, one HBM latency per iteration, and the whole repack ran at 2 TB/s.
template <typename T, bool VEC4, int TAPS>
__device__ __forceinline__ void pack_weight_pair_body(const float* __restrict__ src, T* __restrict__ fwd, T* __restrict__ dgrad,
                                                      int cout, int cin, int taps_arg, int glu, long long dgrad_ld, int bx, int by) {
  extern __shared__ float tile[];                       // [32 o'][32*taps + 1]
  const int taps = TAPS ? TAPS : taps_arg;
  const int ost = 32 * taps + 1;
  const int o0 = bx * 32, c0 = by * 32;
  const int half = cout >> 1;
  const int run = 32 * taps;                            // floats per source row segment
  auto fetch = [&](int i) {
    const int ol = i / run, k = i - ol * run;           // k = (ci_local, tap) flattened as in the source
    const int op = o0 + ol;
    const int ci = c0 + k / taps;
    float v = 0.f;
    if (op < cout && ci < cin) {
      int o = op;
      if (glu) { const int blk = op >> 5, w = op & 31; o = w < 16 ? blk * 16 + w : half + blk * 16 + (w - 16); }
      v = src[((long long)o * cin + c0) * taps + k];
    }
    return v;
  };
  if constexpr (TAPS != 0) {
    constexpr int N = 32 * 32 * TAPS / 256;
    float v[N];
#pragma unroll
    for (int j = 0; j < N; ++j) v[j] = fetch(threadIdx.x + j * 256);
#pragma unroll
    for (int j = 0; j < N; ++j) {
      const int i = threadIdx.x + j * 256;
      const int ol = i / run;
      tile[ol * ost + (i - ol * run)] = v[j];
    }
  } else {
    for (int i = threadIdx.x; i < 32 * run; i += 256) {
      const int ol = i / run;
      tile[ol * ost + (i - ol * run)] = fetch(i);
    }
  }
  __syncthreads();
  if constexpr (VEC4) {
    if (fwd != nullptr) {
      for (int i = threadIdx.x; i < 8 * run; i += 256) { // (o', tap, ci_local / 4) order
        const int ol = i / (run >> 2), r = i - ol * (run >> 2);
        const int tap = r >> 3, cl = (r & 7) * 4;
        const int op = o0 + ol, ci = c0 + cl;
        if (op < cout && ci < cin) {
          const float* t = tile + ol * ost + cl * taps + tap;
          uint2 v;
          v.x = pack_bf16x2(t[0], t[taps]);
          v.y = pack_bf16x2(t[2 * taps], t[3 * taps]);
          *reinterpret_cast<uint2*>(fwd + ((long long)op * taps + tap) * cin + ci) = v;
        }
      }
    }
    if (dgrad != nullptr) {
      for (int i = threadIdx.x; i < 8 * run; i += 256) { // (ci_local, tap', o' / 4) order
        const int ol = (i & 7) * 4, r = i >> 3;
        const int cl = r / taps, tp = r - cl * taps;
        const int op = o0 + ol, ci = c0 + cl;
        if (op < cout && ci < cin) {
          const float* t = tile + ol * ost + cl * taps + (taps - 1 - tp);
          uint2 v;
          v.x = pack_bf16x2(t[0], t[ost]);
          v.y = pack_bf16x2(t[2 * ost], t[3 * ost]);
          *reinterpret_cast<uint2*>(dgrad + ((long long)ci * taps + tp) * dgrad_ld + op) = v;
        }
      }
    }
    return;
  }
  if (fwd != nullptr) {
    for (int i = threadIdx.x; i < 32 * run; i += 256) { // (o', tap, ci_local) order
      const int ol = i / run, r = i - ol * run;
      const int tap = r >> 5, cl = r & 31;
      const int op = o0 + ol, ci = c0 + cl;
      if (op < cout && ci < cin) Elem<T>::st(fwd + ((long long)op * taps + tap) * cin + ci, tile[ol * ost + cl * taps + tap]);
    }
  }
  if (dgrad != nullptr) {
    for (int i = threadIdx.x; i < 32 * run; i += 256) { // (ci_local, tap', o') order
      const int ol = i & 31, r = i >> 5;
      const int cl = r / taps, tp = r - cl * taps;
      const int op = o0 + ol, ci = c0 + cl;
      if (op < cout && ci < cin)
        Elem<T>::st(dgrad + ((long long)ci * taps + tp) * dgrad_ld + op, tile[ol * ost + cl * taps + (taps - 1 - tp)]);
    }
  }
}

template <typename T, bool VEC4, int TAPS>
__global__ void __launch_bounds__(256)
pack_weight_pair_kernel(const float* __restrict__ src, T* __restrict__ fwd, T* __restrict__ dgrad, int cout, int cin, int taps_arg,
                        int glu, long long dgrad_ld) {
  pack_weight_pair_body<T, VEC4, TAPS>(src, fwd, dgrad, cout, cin, taps_arg, glu, dgrad_ld, blockIdx.x, blockIdx.y);
}

// Every filter of a model in ONE launch (per tap count): the training step repacks all ~250 filters after each optimizer step,
// and as 250 launches of a few microseconds each that cost more on the host (and in launch boundaries) than in the kernels.
// Workgroup b finds its job by bisection over the jobs' first-block numbers (device table, ascending).
template <int TAPS>
__global__ void __launch_bounds__(256)
pack_weight_pair_batch_kernel(const LdmPackJob* __restrict__ jobs, int njobs) {
  const int b = blockIdx.x;
  int lo = 0, hi = njobs - 1;
  while (lo < hi) {
    const int mid = (lo + hi + 1) >> 1;
    if (jobs[mid].first_block <= b) lo = mid; else hi = mid - 1;
  }
  const LdmPackJob j = jobs[lo];
  const int local = b - j.first_block;
  const int by = local / j.blocks_x, bx = local - by * j.blocks_x;
  if (by >= j.blocks_y) return;
  pack_weight_pair_body<bf16_t, true, TAPS>(j.src, static_cast<bf16_t*>(j.fwd), static_cast<bf16_t*>(j.dgrad), j.cout, j.cin, TAPS,
                                            j.glu_pairs, j.dgrad_ld, bx, by);
}

template <typename T>
__global__ void timestep_embedding_kernel(const float* __restrict__ t, int batch, int dim, T* __restrict__ out) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  const int half = dim >> 1;
  if (i >= batch * dim) return;
  const int b = i / dim, c = i - b * dim;
  if (c >= 2 * half) { Elem<T>::st(out + i, 0.f); return; }  // odd dim: zero pad
  const int k = c < half ? c : c - half;
  // float32 throughout, in the order of diffusers' get_timestep_embedding
  const float exponent = (-9.210340371976184f * (float)k) / (float)(half - 1);
  const float arg = t[b] * expf(exponent);
  Elem<T>::st(out + i, c < half ? cosf(arg) : sinf(arg));  // flip_sin_to_cos: cosine half first
}

// conv_in: NCHW (few channels) -> NHWC, one thread per (pixel, 8 output channels).  The filter is staged once per
// block in LDS as [tap][cout] so that neighbouring threads (neighbouring channel chunks) read neighbouring words.
template <typename TX, typename TO, int CIN>
__global__ void __launch_bounds__(256)
conv_in_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, TO* __restrict__ out,
               int batch, int h, int wd, int cout, int cblk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // [CIN*9][cb]: this workgroup's range of output channels (blockIdx.y)
  constexpr int TAPS = CIN * 9;
  const int co0 = blockIdx.y * cblk;
  const int cb = cout - co0 < cblk ? cout - co0 : cblk;
  for (int i = threadIdx.x; i < TAPS * cb; i += 256) {
    const int co = i / TAPS, tap = i - co * TAPS;   // w is [cout][CIN][3][3]
    wl[tap * cb + co] = w[(size_t)co0 * TAPS + i];
  }
  __syncthreads();
  const unsigned chunks = cb >> 3;
  const unsigned total = (unsigned)batch * h * wd * chunks;    // < 2^31 (checked on the host): 32-bit divisions
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const unsigned pix = i / chunks;
    const int ch = (int)(i - pix * chunks);
    const unsigned prow = pix / (unsigned)wd;
    const int ow = (int)(pix - prow * wd);
    const int b = (int)(prow / (unsigned)h);
    const int oh = (int)(prow - (unsigned)b * h);
    float acc[8];
    load8(bias + co0 + ch * 8, acc);
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh + kh - 1;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = ow + kw - 1;
          float v = 0.f;
          if (ih >= 0 && ih < h && iw >= 0 && iw < wd) v = Elem<TX>::ld(x + (((int64_t)b * CIN + ci) * h + ih) * wd + iw);
          float wv[8];
          load8(wl + ((ci * 3 + kh) * 3 + kw) * cb + ch * 8, wv);
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[j] += v * wv[j];
        }
      }
    }
    store8(out + (size_t)pix * cout + co0 + ch * 8, acc);
  }
}

// conv_in, four horizontally adjacent pixels per thread: the 36 x 8 filter words a thread reads from LDS and the input
// values of a row serve four outputs (the one-pixel form spends its time on LDS filter reads: 72 ds_read_b128 per pixel
// and chunk against 288 FMAs).  Needs wd % 4 == 0; the input row segment [ow0 - 1, ow0 + 4] is read once per (ci, kh).
template <typename TX, typename TO, int CIN>
__global__ void __launch_bounds__(256)
conv_in4_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, TO* __restrict__ out,
                int batch, int h, int wd, int cout, int cblk) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // [CIN*9][cb]
  constexpr int TAPS = CIN * 9;
  const int co0 = blockIdx.y * cblk;
  const int cb = cout - co0 < cblk ? cout - co0 : cblk;
  for (int i = threadIdx.x; i < TAPS * cb; i += 256) {
    const int co = i / TAPS, tap = i - co * TAPS;   // w is [cout][CIN][3][3]
    wl[tap * cb + co] = w[(size_t)co0 * TAPS + i];
  }
  __syncthreads();
  const unsigned chunks = cb >> 3;
  const unsigned wq = (unsigned)wd >> 2;                           // pixel quads per row
  const unsigned total = (unsigned)batch * h * wq * chunks;        // < 2^31 (checked on the host)
  const unsigned stride = gridDim.x * blockDim.x;
  for (unsigned i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
    const unsigned quad = i / chunks;
    const int ch = (int)(i - quad * chunks);
    const unsigned prow = quad / wq;
    const int ow0 = (int)(quad - prow * wq) * 4;
    const int b = (int)(prow / (unsigned)h);
    const int oh = (int)(prow - (unsigned)b * h);
    float acc[4][8];
    {
      float bv[8];
      load8(bias + co0 + ch * 8, bv);
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[q][j] = bv[j];
    }
#pragma unroll
    for (int ci = 0; ci < CIN; ++ci) {
#pragma unroll
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh + kh - 1;
        float v[6];
#pragma unroll
        for (int e = 0; e < 6; ++e) {
          const int iw = ow0 + e - 1;
          v[e] = 0.f;
          if (ih >= 0 && ih < h && iw >= 0 && iw < wd) v[e] = Elem<TX>::ld(x + (((int64_t)b * CIN + ci) * h + ih) * wd + iw);
        }
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
          float wv[8];
          load8(wl + ((ci * 3 + kh) * 3 + kw) * cb + ch * 8, wv);
#pragma unroll
          for (int q = 0; q < 4; ++q)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[q][j] += v[q + kw] * wv[j];
        }
      }
    }
    const size_t pix0 = (size_t)prow * wd + ow0;
#pragma unroll
    for (int q = 0; q < 4; ++q) store8(out + (pix0 + q) * cout + co0 + ch * 8, acc[q]);
  }
}

// generic channel count (slow path)
template <typename TX, typename TO>
__global__ void __launch_bounds__(256)
conv_in_generic_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias,
                       TO* __restrict__ out, int batch, int cin, int h, int wd, int cout) {
  const int64_t total = (int64_t)batch * h * wd * cout;
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= total) return;
  const int co = (int)(i % cout);
  const int64_t pix = i / cout;
  const int ow = (int)(pix % wd);
  const int oh = (int)((pix / wd) % h);
  const int b = (int)(pix / ((int64_t)wd * h));
  float acc = bias[co];
  for (int ci = 0; ci < cin; ++ci)
    for (int kh = 0; kh < 3; ++kh) {
      const int ih = oh + kh - 1;
      if (ih < 0 || ih >= h) continue;
      for (int kw = 0; kw < 3; ++kw) {
        const int iw = ow + kw - 1;
        if (iw < 0 || iw >= wd) continue;
        acc += Elem<TX>::ld(x + (((int64_t)b * cin + ci) * h + ih) * wd + iw) * w[((co * cin + ci) * 3 + kh) * 3 + kw];
      }
    }
  Elem<TO>::st(out + i, acc);
}

// final conv: NHWC (many channels) -> NCHW (COUT <= 8 channels); one lane per output pixel, filter taps are
// wave-uniform (scalar loads), the activation row is streamed 16 bytes at a time.
template <typename TX, typename TO, int COUT>
__global__ void __launch_bounds__(256)
conv_out_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, TO* __restrict__ out,
                int batch, int cin, int h, int wd) {
  const int64_t npix = (int64_t)batch * h * wd;
  const int64_t pix = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (pix >= npix) return;
  const int ow = (int)(pix % wd);
  const int oh = (int)((pix / wd) % h);
  const int b = (int)(pix / ((int64_t)wd * h));
  float acc[COUT];
#pragma unroll
  for (int co = 0; co < COUT; ++co) acc[co] = bias[co];
  for (int kh = 0; kh < 3; ++kh) {
    const int ih = oh + kh - 1;
    if (ih < 0 || ih >= h) continue;
    for (int kw = 0; kw < 3; ++kw) {
      const int iw = ow + kw - 1;
      if (iw < 0 || iw >= wd) continue;
      const TX* row = x + (((int64_t)b * h + ih) * wd + iw) * cin;
      for (int c = 0; c < cin; c += 8) {
        float v[8];
        load8(row + c, v);
#pragma unroll
        for (int co = 0; co < COUT; ++co) {
          const float* wp = w + ((int64_t)(co * cin + c) * 3 + kh) * 3 + kw;
#pragma unroll
          for (int j = 0; j < 8; ++j) acc[co] += v[j] * wp[j * 9];
        }
      }
    }
  }
#pragma unroll
  for (int co = 0; co < COUT; ++co) Elem<TO>::st(out + (((int64_t)b * COUT + co) * h + oh) * wd + ow, acc[co]);
}

// Wide-image form of the final conv (the VAE decoder's 128 -> 3 at up to 768x768): 16 lanes per output pixel, each lane
// owning 8-channel chunks of the pixel's NHWC row, so a wave reads 4 neighbouring pixels as one contiguous run per tap; the
// filter sits in LDS as [tap][cin][COUT] (lanes of a pixel read neighbouring words, the four pixels broadcast); partial
// sums meet in a 16-lane butterfly.  Summation order is fixed.
template <typename TX, typename TO, int COUT>
__global__ void __launch_bounds__(256)
conv_out_lanes_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, TO* __restrict__ out,
                      int batch, int cin, int h, int wd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // [9][COUT][rowf]: chunk c of a row at c*8 + (c>>3)*4 floats, so the 16
  const int rowf = cin + (cin >> 6) * 4 + 4;       // chunks a 16-lane group reads at once fall into 16 different bank quads
  for (int i = threadIdx.x; i < 9 * cin * COUT; i += 256) {
    const int tap = i % 9, c = (i / 9) % cin, co = i / (9 * cin);   // w is [COUT][cin][3][3]: coalesced read
    wl[(tap * COUT + co) * rowf + c + (c >> 6) * 4] = w[i];
  }
  __syncthreads();
  const int sub = threadIdx.x & 15;
  const int chunks = cin >> 3;
  const unsigned npix = (unsigned)batch * h * wd;  // < 2^31 (checked on the host): 32-bit divisions
  const unsigned stride = gridDim.x * 16;
  for (unsigned base = blockIdx.x * 16; base < npix; base += stride) {
    const unsigned pix = base + (threadIdx.x >> 4);
    const bool live = pix < npix;                  // whole 16-lane groups stay in the loop for the butterfly
    const unsigned prow = pix / (unsigned)wd;
    const int ow = (int)(pix - prow * wd);
    const int b = (int)(prow / (unsigned)h);
    const int oh = (int)(prow - (unsigned)b * h);
    float acc[COUT];
#pragma unroll
    for (int co = 0; co < COUT; ++co) acc[co] = 0.f;
    if (live) {
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh + kh - 1;
        if (ih < 0 || ih >= h) continue;
        for (int kw = 0; kw < 3; ++kw) {
          const int iw = ow + kw - 1;
          if (iw < 0 || iw >= wd) continue;
          const TX* row = x + (((size_t)b * h + ih) * wd + iw) * cin;
          const float* wt = wl + (size_t)(kh * 3 + kw) * COUT * rowf;
          for (int ch = sub; ch < chunks; ch += 16) {
            float v[8];
            load8(row + ch * 8, v);
            const float* wp = wt + ch * 8 + (ch >> 3) * 4;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
              float wv[8];
              load8(wp + co * rowf, wv);
#pragma unroll
              for (int j = 0; j < 8; ++j) acc[co] += v[j] * wv[j];
            }
          }
        }
      }
    }
#pragma unroll
    for (int co = 0; co < COUT; ++co) {
#pragma unroll
      for (int o = 8; o > 0; o >>= 1) acc[co] += __shfl_xor(acc[co], o, 16);
    }
    if (live && sub < COUT) {
      float r = acc[0];
#pragma unroll
      for (int co = 1; co < COUT; ++co) r = sub == co ? acc[co] : r;
      Elem<TO>::st(out + (((size_t)b * COUT + sub) * h + oh) * wd + ow, r + bias[sub]);
    }
  }
}

// Final conv, four horizontally adjacent pixels per 16-lane group: the filter words a lane reads from LDS serve four
// outputs and the six input positions of a row serve three taps each (the one-pixel form issues 72 ds_read_b128 and 9
// 16-byte loads per pixel and chunk; this one 18 and 4.5).  Needs wd % 4 == 0.  Same summation order per output.
template <typename TX, typename TO, int COUT>
__global__ void __launch_bounds__(256)
conv_out_lanes4_kernel(const TX* __restrict__ x, const float* __restrict__ w, const float* __restrict__ bias, TO* __restrict__ out,
                       int batch, int cin, int h, int wd) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* wl = reinterpret_cast<float*>(smem_raw);  // [9][COUT][rowf], rows padded as in conv_out_lanes_kernel
  const int rowf = cin + (cin >> 6) * 4 + 4;
  for (int i = threadIdx.x; i < 9 * cin * COUT; i += 256) {
    const int tap = i % 9, c = (i / 9) % cin, co = i / (9 * cin);   // w is [COUT][cin][3][3]: coalesced read
    wl[(tap * COUT + co) * rowf + c + (c >> 6) * 4] = w[i];
  }
  __syncthreads();
  const int sub = threadIdx.x & 15;
  const int chunks = cin >> 3;
  const unsigned wq = (unsigned)wd >> 2;
  const unsigned nquad = (unsigned)batch * h * wq;   // < 2^31 (checked on the host)
  const unsigned stride = gridDim.x * 16;
  for (unsigned base = blockIdx.x * 16; base < nquad; base += stride) {
    const unsigned quad = base + (threadIdx.x >> 4);
    const bool live = quad < nquad;                  // whole 16-lane groups stay in the loop for the butterfly
    const unsigned prow = quad / wq;
    const int ow0 = (int)(quad - prow * wq) * 4;
    const int b = (int)(prow / (unsigned)h);
    const int oh = (int)(prow - (unsigned)b * h);
    float acc[4][COUT];
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int co = 0; co < COUT; ++co) acc[q][co] = 0.f;
    if (live) {
      for (int kh = 0; kh < 3; ++kh) {
        const int ih = oh + kh - 1;
        if (ih < 0 || ih >= h) continue;
        const TX* row = x + (((size_t)b * h + ih) * wd) * cin;
        for (int ch = sub; ch < chunks; ch += 16) {
          float v[6][8];
#pragma unroll
          for (int e = 0; e < 6; ++e) {
            const int iw = ow0 + e - 1;
            if (iw >= 0 && iw < wd) {
              load8(row + (size_t)iw * cin + ch * 8, v[e]);
            } else {
#pragma unroll
              for (int j = 0; j < 8; ++j) v[e][j] = 0.f;
            }
          }
#pragma unroll
          for (int kw = 0; kw < 3; ++kw) {
            const float* wp = wl + (size_t)(kh * 3 + kw) * COUT * rowf + ch * 8 + (ch >> 3) * 4;
#pragma unroll
            for (int co = 0; co < COUT; ++co) {
              float wv[8];
              load8(wp + co * rowf, wv);
#pragma unroll
              for (int q = 0; q < 4; ++q)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[q][co] += v[q + kw][j] * wv[j];
            }
          }
        }
      }
    }
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int co = 0; co < COUT; ++co) {
#pragma unroll
        for (int o = 8; o > 0; o >>= 1) acc[q][co] += __shfl_xor(acc[q][co], o, 16);
      }
    const int co_s = sub >> 2, q_s = sub & 3;
    if (live && co_s < COUT) {
      float r = 0.f;
#pragma unroll
      for (int q = 0; q < 4; ++q)
#pragma unroll
        for (int co = 0; co < COUT; ++co) r = (q == q_s && co == co_s) ? acc[q][co] : r;
      Elem<TO>::st(out + (((size_t)b * COUT + co_s) * h + oh) * wd + ow0 + q_s, r + bias[co_s]);
    }
  }
}

template <typename T>
__global__ void silu_kernel(const T* __restrict__ x, T* __restrict__ y, int64_t n) {
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) { const float v = Elem<T>::ld(x + i); Elem<T>::st(y + i, v / (1.0f + expf(-v))); }
}

static int blocks_for(int64_t n) { return (int)((n + 255) / 256); }

}  // namespace ldm

using namespace ldm;

extern "C" int ldm_pack_weight(const float* src, void* dst, int dst_dtype, int cout, int cin, int kh, int kw,
                               int glu_pairs, void* stream) {
  LDM_REQUIRE(dtype_ok(dst_dtype), "ldm_pack_weight: bad dtype %d", dst_dtype);
  LDM_REQUIRE(src && dst, "ldm_pack_weight: null pointer");
  LDM_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1, "ldm_pack_weight: bad sizes");
  LDM_REQUIRE(!glu_pairs || cout % 32 == 0, "ldm_pack_weight: glu_pairs needs cout %% 32 == 0 (got %d)", cout);
  const int64_t total = (int64_t)cout * cin * kh * kw;
  int grid = (int)((total + 255) / 256 > 65535 * 4 ? 65535 * 4 : (total + 255) / 256);
  hipStream_t st = (hipStream_t)stream;
  if (dst_dtype == LDM_BF16)
    hipLaunchKernelGGL((pack_weight_kernel<bf16_t>), dim3(grid), dim3(256), 0, st, src, (bf16_t*)dst, cout, cin, kh * kw, glu_pairs);
  else
    hipLaunchKernelGGL((pack_weight_kernel<float>), dim3(grid), dim3(256), 0, st, src, (float*)dst, cout, cin, kh * kw, glu_pairs);
  LDM_HIP_CHECK_LAUNCH("ldm_pack_weight");
  return LDM_OK;
}

extern "C" int ldm_pack_weight_pair(const float* src, void* fwd, void* dgrad, int dst_dtype, int cout, int cin, int kh, int kw,
                                    int glu_pairs, long long dgrad_ld, void* stream) {
  LDM_REQUIRE(dtype_ok(dst_dtype), "ldm_pack_weight_pair: bad dtype %d", dst_dtype);
  LDM_REQUIRE(src && (fwd || dgrad), "ldm_pack_weight_pair: null pointer");
  LDM_REQUIRE(cout >= 1 && cin >= 1 && kh >= 1 && kw >= 1 && kh * kw <= 9, "ldm_pack_weight_pair: bad sizes");
  LDM_REQUIRE(!glu_pairs || cout % 32 == 0, "ldm_pack_weight_pair: glu_pairs needs cout %% 32 == 0 (got %d)", cout);
  LDM_REQUIRE(dgrad == nullptr || dgrad_ld >= cout, "ldm_pack_weight_pair: dgrad row pitch %lld below cout %d", dgrad_ld, cout);
  const int taps = kh * kw;
  const dim3 grid((cout + 31) / 32, (cin + 31) / 32);
  LDM_REQUIRE(grid.y <= 65535, "ldm_pack_weight_pair: cin too large");
  const size_t smem = (size_t)32 * (32 * taps + 1) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  const bool vec4 = dst_dtype == LDM_BF16 && cout % 4 == 0 && cin % 4 == 0 && (dgrad == nullptr || dgrad_ld % 4 == 0) &&
                    (reinterpret_cast<uintptr_t>(fwd) & 7) == 0 && (reinterpret_cast<uintptr_t>(dgrad) & 7) == 0;
#define LDM_PACK_PAIR(T, V, TP) \
  hipLaunchKernelGGL((pack_weight_pair_kernel<T, V, TP>), grid, dim3(256), smem, st, src, (T*)fwd, (T*)dgrad, cout, cin, taps, glu_pairs, dgrad_ld)
  if (vec4) {
    if (taps == 9) LDM_PACK_PAIR(bf16_t, true, 9); else if (taps == 1) LDM_PACK_PAIR(bf16_t, true, 1); else LDM_PACK_PAIR(bf16_t, true, 0);
  } else if (dst_dtype == LDM_BF16) {
    LDM_PACK_PAIR(bf16_t, false, 0);
  } else {
    LDM_PACK_PAIR(float, false, 0);
  }
#undef LDM_PACK_PAIR
  LDM_HIP_CHECK_LAUNCH("ldm_pack_weight_pair");
  return LDM_OK;
}

extern "C" int ldm_pack_weight_pair_batch(const LdmPackJob* jobs_device, int njobs, int total_blocks, int taps, void* stream) {
  LDM_REQUIRE(njobs >= 0 && total_blocks >= 0, "ldm_pack_weight_pair_batch: bad sizes");
  if (njobs == 0 || total_blocks == 0) return LDM_OK;
  LDM_REQUIRE(jobs_device != nullptr, "ldm_pack_weight_pair_batch: null job table");
  LDM_REQUIRE(taps == 1 || taps == 9, "ldm_pack_weight_pair_batch: taps must be 1 or 9 (got %d)", taps);
  const size_t smem = (size_t)32 * (32 * taps + 1) * sizeof(float);
  hipStream_t st = (hipStream_t)stream;
  if (taps == 1) hipLaunchKernelGGL((pack_weight_pair_batch_kernel<1>), dim3(total_blocks), dim3(256), smem, st, jobs_device, njobs);
  else hipLaunchKernelGGL((pack_weight_pair_batch_kernel<9>), dim3(total_blocks), dim3(256), smem, st, jobs_device, njobs);
  LDM_HIP_CHECK_LAUNCH("ldm_pack_weight_pair_batch");
  return LDM_OK;
}

extern "C" int ldm_timestep_embedding(const float* t, int batch, int dim, void* out, int dtype, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_timestep_embedding: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && dim >= 4, "ldm_timestep_embedding: bad sizes");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(t && out, "ldm_timestep_embedding: null pointer");
  hipStream_t st = (hipStream_t)stream;
  const int n = batch * dim;
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((timestep_embedding_kernel<bf16_t>), dim3(blocks_for(n)), dim3(256), 0, st, t, batch, dim, (bf16_t*)out);
  else
    hipLaunchKernelGGL((timestep_embedding_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st, t, batch, dim, (float*)out);
  LDM_HIP_CHECK_LAUNCH("ldm_timestep_embedding");
  return LDM_OK;
}

template <typename TX, typename TO>
static int conv_in_dispatch(const void* x, const float* w, const float* bias, void* out, int batch, int cin, int h,
                            int wd, int cout, hipStream_t st) {
  const int64_t npix = (int64_t)batch * h * wd;
  if (cin == 4 && cout % 8 == 0 && aligned16(out) && aligned16(bias) && npix * (cout / 8) < (1LL << 31)) {
    const int cblk = cout <= 448 ? cout : 256;   // output channels per workgroup: 36 * cblk floats of LDS (<= 64 KB)
    const int ny = (cout + cblk - 1) / cblk;
    if (wd % 4 == 0 && npix >= 4096) {   // four pixels per thread (the summation order per output is the same)
      int64_t nb = (npix / 4 * (cblk / 8) + 255) / 256;
      if (nb > 2048) nb = 2048;
      hipLaunchKernelGGL((conv_in4_kernel<TX, TO, 4>), dim3((int)nb, ny), dim3(256), (size_t)36 * cblk * sizeof(float), st,
                         (const TX*)x, w, bias, (TO*)out, batch, h, wd, cout, cblk);
      LDM_HIP_CHECK_LAUNCH("ldm_conv_in");
      return LDM_OK;
    }
    int64_t nb = (npix * (cblk / 8) + 255) / 256;
    if (nb > 2048) nb = 2048;  // grid-stride: the LDS filter image is filled once per block
    hipLaunchKernelGGL((conv_in_kernel<TX, TO, 4>), dim3((int)nb, ny), dim3(256), (size_t)36 * cblk * sizeof(float), st, (const TX*)x,
                       w, bias, (TO*)out, batch, h, wd, cout, cblk);
  } else {
    hipLaunchKernelGGL((conv_in_generic_kernel<TX, TO>), dim3(blocks_for(npix * cout)), dim3(256), 0, st, (const TX*)x, w, bias,
                       (TO*)out, batch, cin, h, wd, cout);
  }
  LDM_HIP_CHECK_LAUNCH("ldm_conv_in");
  return LDM_OK;
}

extern "C" int ldm_conv_in(const void* x, int x_dtype, const float* w, const float* bias, void* out, int out_dtype,
                           int batch, int cin, int h, int wd, int cout, void* stream) {
  LDM_REQUIRE(dtype_ok(x_dtype) && dtype_ok(out_dtype), "ldm_conv_in: bad dtype");
  LDM_REQUIRE(batch >= 0 && cin >= 1 && h >= 1 && wd >= 1 && cout >= 1, "ldm_conv_in: bad sizes");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(x && w && bias && out, "ldm_conv_in: null pointer");
  LDM_REQUIRE((int64_t)batch * h * wd * cout / 256 < (1LL << 31), "ldm_conv_in: too large");
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == LDM_F32 && out_dtype == LDM_F32) return conv_in_dispatch<float, float>(x, w, bias, out, batch, cin, h, wd, cout, st);
  if (x_dtype == LDM_F32 && out_dtype == LDM_BF16) return conv_in_dispatch<float, bf16_t>(x, w, bias, out, batch, cin, h, wd, cout, st);
  if (x_dtype == LDM_BF16 && out_dtype == LDM_F32) return conv_in_dispatch<bf16_t, float>(x, w, bias, out, batch, cin, h, wd, cout, st);
  return conv_in_dispatch<bf16_t, bf16_t>(x, w, bias, out, batch, cin, h, wd, cout, st);
}

template <typename TX, typename TO>
static int conv_out_dispatch(const void* x, const float* w, const float* bias, void* out, int batch, int cin, int h,
                             int wd, int cout, hipStream_t st) {
  const int64_t npix = (int64_t)batch * h * wd;
  dim3 grid(blocks_for(npix)), block(256);
  const size_t lds = (size_t)9 * cout * (cin + (cin >> 6) * 4 + 4) * sizeof(float);
  if (npix >= 65536 && npix < (1LL << 31) && lds <= 65536 && cout <= 4) {   // large images: 16 lanes per pixel, filter in LDS
    const bool quad = wd % 4 == 0;                   // four pixels per 16-lane group
    int64_t nb = ((quad ? npix / 4 : npix) + 15) / 16;
    if (nb > 8192) nb = 8192;                        // grid-stride: the LDS filter image is filled once per workgroup
    const dim3 g((unsigned)nb);
#define LDM_CONV_OUT_LANES(CO) \
  if (quad) hipLaunchKernelGGL((conv_out_lanes4_kernel<TX, TO, CO>), g, block, lds, st, (const TX*)x, w, bias, (TO*)out, batch, cin, h, wd); \
  else hipLaunchKernelGGL((conv_out_lanes_kernel<TX, TO, CO>), g, block, lds, st, (const TX*)x, w, bias, (TO*)out, batch, cin, h, wd)
    switch (cout) {
      case 1: LDM_CONV_OUT_LANES(1); break;
      case 2: LDM_CONV_OUT_LANES(2); break;
      case 3: LDM_CONV_OUT_LANES(3); break;
      default: LDM_CONV_OUT_LANES(4); break;
    }
#undef LDM_CONV_OUT_LANES
    LDM_HIP_CHECK_LAUNCH("ldm_conv_out");
    return LDM_OK;
  }
#define LDM_CONV_OUT(CO) \
  hipLaunchKernelGGL((conv_out_kernel<TX, TO, CO>), grid, block, 0, st, (const TX*)x, w, bias, (TO*)out, batch, cin, h, wd)
  switch (cout) {
    case 1: LDM_CONV_OUT(1); break;
    case 2: LDM_CONV_OUT(2); break;
    case 3: LDM_CONV_OUT(3); break;
    case 4: LDM_CONV_OUT(4); break;
    case 8: LDM_CONV_OUT(8); break;
    default:
      set_error("ldm_conv_out: cout must be 1, 2, 3, 4 or 8 (got %d)", cout);
      return LDM_EUNSUPPORTED;
  }
#undef LDM_CONV_OUT
  LDM_HIP_CHECK_LAUNCH("ldm_conv_out");
  return LDM_OK;
}

extern "C" int ldm_conv_out(const void* x, int x_dtype, const float* w, const float* bias, void* out, int out_dtype,
                            int batch, int cin, int h, int wd, int cout, void* stream) {
  LDM_REQUIRE(dtype_ok(x_dtype) && dtype_ok(out_dtype), "ldm_conv_out: bad dtype");
  LDM_REQUIRE(batch >= 0 && cin >= 8 && cin % 8 == 0 && h >= 1 && wd >= 1 && cout >= 1, "ldm_conv_out: bad sizes (cin %% 8 == 0 required)");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(x && w && bias && out, "ldm_conv_out: null pointer");
  LDM_REQUIRE(aligned16(x), "ldm_conv_out: input must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  if (x_dtype == LDM_F32 && out_dtype == LDM_F32) return conv_out_dispatch<float, float>(x, w, bias, out, batch, cin, h, wd, cout, st);
  if (x_dtype == LDM_F32 && out_dtype == LDM_BF16) return conv_out_dispatch<float, bf16_t>(x, w, bias, out, batch, cin, h, wd, cout, st);
  if (x_dtype == LDM_BF16 && out_dtype == LDM_F32) return conv_out_dispatch<bf16_t, float>(x, w, bias, out, batch, cin, h, wd, cout, st);
  return conv_out_dispatch<bf16_t, bf16_t>(x, w, bias, out, batch, cin, h, wd, cout, st);
}

extern "C" int ldm_silu(const void* x, void* y, int dtype, int64_t n, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_silu: bad dtype %d", dtype);
  LDM_REQUIRE(n >= 0 && n / 256 < (1LL << 31), "ldm_silu: bad size");
  if (n == 0) return LDM_OK;
  LDM_REQUIRE(x && y, "ldm_silu: null pointer");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LDM_BF16) hipLaunchKernelGGL((silu_kernel<bf16_t>), dim3(blocks_for(n)), dim3(256), 0, st, (const bf16_t*)x, (bf16_t*)y, n);
  else hipLaunchKernelGGL((silu_kernel<float>), dim3(blocks_for(n)), dim3(256), 0, st, (const float*)x, (float*)y, n);
  LDM_HIP_CHECK_LAUNCH("ldm_silu");
  return LDM_OK;
}
