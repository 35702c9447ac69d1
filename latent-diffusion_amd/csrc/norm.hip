// GroupNorm (statistics + apply, optional SiLU, two-source virtual concat) and LayerNorm on NHWC activations.
// All of these are HBM-bound streaming kernels: 16-byte vector accesses, fp32 statistics.
#include "common.h"

#include <mutex>

namespace ldm {

constexpr int GN_THREADS = 256;
constexpr int GN_MAX_SLABS = 64;   // partial-sum slabs per sample (see ldm_groupnorm_stats)
constexpr int GN_MAX_CHUNKS = 4;   // 8-channel chunks per thread -> C <= 8 * 256 * 4
constexpr int GN_UNROLL = 4;       // pixels in flight per thread in the streaming loops

__device__ __forceinline__ void unpack8(const uint4& r, float (&v)[8]) {
  const uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}

// Thread layout shared by the two GroupNorm kernels: CT threads across the 8-channel chunks of a pixel,
// PIX = 256 / CT pixels per pass; a thread keeps its channel chunk(s) for the whole kernel.
struct GnGeom {
  int c0, c1, ctot, nchunks, ct, pix, cpg, groups, hw, slab_pixels;
};

template <typename T>
__device__ __forceinline__ const T* gn_src(const T* a0, const T* a1, const GnGeom& g, int64_t pixel, int c, int& cs) {
  if (c < g.c0) { cs = g.c0; return a0 + pixel * g.c0 + c; }
  cs = g.c1;
  return a1 + pixel * g.c1 + (c - g.c0);
}

// stage 1: per (sample, slab) partial sum / sum of squares per group -> partial[(b*slabs + s)*groups*2]
template <typename T>
__global__ void __launch_bounds__(GN_THREADS)
gn_partial_kernel(const T* __restrict__ a0, const T* __restrict__ a1, GnGeom g, float* __restrict__ partial) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);  // [pix][ctot][2]
  const int b = blockIdx.y, slab = blockIdx.x, nslabs = gridDim.x;
  const int tid = threadIdx.x;
  const int tp = tid / g.ct, tc = tid - tp * g.ct;
  const bool active = tp < g.pix;
  const int p_begin = slab * g.slab_pixels;
  const int p_end = min(p_begin + g.slab_pixels, g.hw);
  float s[GN_MAX_CHUNKS][8], ss[GN_MAX_CHUNKS][8];
#pragma unroll
  for (int k = 0; k < GN_MAX_CHUNKS; ++k)
#pragma unroll
    for (int j = 0; j < 8; ++j) s[k][j] = ss[k][j] = 0.f;
  if (active) {
    if (g.nchunks <= g.ct) {
      // one chunk per thread (C <= 2048): GN_UNROLL pixels in flight per thread -- a single 16-byte load per
      // loop trip left the kernel latency-bound at 45 % of the HBM rate
      const int cc = tc;
      int p = p_begin + tp;
      for (; p + (GN_UNROLL - 1) * g.pix < p_end; p += GN_UNROLL * g.pix) {
        uint4 raw[GN_UNROLL];
        float4 rawf[GN_UNROLL][2];
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
          int cs;
          const T* src = gn_src(a0, a1, g, (int64_t)b * g.hw + p + u * g.pix, cc * 8, cs);
          if constexpr (sizeof(T) == 2) raw[u] = *reinterpret_cast<const uint4*>(src);
          else { rawf[u][0] = *reinterpret_cast<const float4*>(src); rawf[u][1] = *reinterpret_cast<const float4*>(src + 4); }
        }
#pragma unroll
        for (int u = 0; u < GN_UNROLL; ++u) {
          float v[8];
          if constexpr (sizeof(T) == 2) unpack8(raw[u], v);
          else { v[0] = rawf[u][0].x; v[1] = rawf[u][0].y; v[2] = rawf[u][0].z; v[3] = rawf[u][0].w;
                 v[4] = rawf[u][1].x; v[5] = rawf[u][1].y; v[6] = rawf[u][1].z; v[7] = rawf[u][1].w; }
#pragma unroll
          for (int j = 0; j < 8; ++j) { s[0][j] += v[j]; ss[0][j] += v[j] * v[j]; }
        }
      }
      for (; p < p_end; p += g.pix) {
        int cs;
        float v[8];
        load8(gn_src(a0, a1, g, (int64_t)b * g.hw + p, cc * 8, cs), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) { s[0][j] += v[j]; ss[0][j] += v[j] * v[j]; }
      }
    } else {
    for (int p = p_begin + tp; p < p_end; p += g.pix) {
      const int64_t pixel = (int64_t)b * g.hw + p;
#pragma unroll
      for (int k = 0; k < GN_MAX_CHUNKS; ++k) {
        const int cc = tc + k * g.ct;
        if (cc < g.nchunks) {
          int cs;
          float v[8];
          load8(gn_src(a0, a1, g, pixel, cc * 8, cs), v);
#pragma unroll
          for (int j = 0; j < 8; ++j) { s[k][j] += v[j]; ss[k][j] += v[j] * v[j]; }
        }
      }
    }
    }
#pragma unroll
    for (int k = 0; k < GN_MAX_CHUNKS; ++k) {
      const int cc = tc + k * g.ct;
      if (cc < g.nchunks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          red[((size_t)tp * g.ctot + cc * 8 + j) * 2] = s[k][j];
          red[((size_t)tp * g.ctot + cc * 8 + j) * 2 + 1] = ss[k][j];
        }
      }
    }
  }
  __syncthreads();
  // one thread per group folds its channels over the pix rows (fixed order -> deterministic)
  for (int grp = tid; grp < g.groups; grp += GN_THREADS) {
    float a = 0.f, q = 0.f;
    for (int c = grp * g.cpg; c < (grp + 1) * g.cpg; ++c)
      for (int r = 0; r < g.pix; ++r) {
        a += red[((size_t)r * g.ctot + c) * 2];
        q += red[((size_t)r * g.ctot + c) * 2 + 1];
      }
    float* o = partial + (((size_t)b * nslabs + slab) * g.groups + grp) * 2;
    o[0] = a;
    o[1] = q;
  }
}

// stage 2: fold the slabs in double, emit mean / rstd.  One wave per (sample, group): a lane takes one slab, the fold
// is a fixed butterfly (deterministic); a single thread walking 64 dependent loads took 15 us per launch.
__global__ void __launch_bounds__(256)
gn_finalize_kernel(const float* __restrict__ partial, int nslabs, int groups, int total, double count,
                   float eps, float* __restrict__ stats) {
  const int i = blockIdx.x * 4 + (threadIdx.x >> 6);  // over batch*groups
  const int lane = threadIdx.x & 63;
  if (i >= total) return;
  const int b = i / groups, grp = i - b * groups;
  double a = 0.0, q = 0.0;
  for (int s = lane; s < nslabs; s += 64) {
    const float* o = partial + (((size_t)b * nslabs + s) * groups + grp) * 2;
    a += (double)o[0];
    q += (double)o[1];
  }
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) {
    a += __shfl_xor(a, o, 64);
    q += __shfl_xor(q, o, 64);
  }
  if (lane == 0) {
    const double mean = a / count;
    double var = q / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    stats[2 * i] = (float)mean;
    stats[2 * i + 1] = (float)(1.0 / sqrt(var + (double)eps));
  }
}

template <typename T>
__global__ void __launch_bounds__(GN_THREADS)
gn_apply_kernel(const T* __restrict__ a0, const T* __restrict__ a1, GnGeom g, const float* __restrict__ stats,
                const float* __restrict__ gamma, const float* __restrict__ beta, int silu, T* __restrict__ out) {
  const int b = blockIdx.y, slab = blockIdx.x;
  const int tid = threadIdx.x;
  const int tp = tid / g.ct, tc = tid - tp * g.ct;
  if (tp >= g.pix) return;
  const int p_begin = slab * g.slab_pixels;
  const int p_end = min(p_begin + g.slab_pixels, g.hw);
  float sc[GN_MAX_CHUNKS][8], sh[GN_MAX_CHUNKS][8];
#pragma unroll
  for (int k = 0; k < GN_MAX_CHUNKS; ++k) {
    const int cc = tc + k * g.ct;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
      sc[k][j] = sh[k][j] = 0.f;
      if (cc < g.nchunks) {
        const int c = cc * 8 + j;
        const float* st = stats + ((size_t)b * g.groups + c / g.cpg) * 2;
        const float w = st[1] * gamma[c];
        sc[k][j] = w;
        sh[k][j] = beta[c] - st[0] * w;
      }
    }
  }
  int p = p_begin + tp;
  if (g.nchunks <= g.ct) {   // one chunk per thread: GN_UNROLL pixels in flight (see gn_partial_kernel)
    const int cc = tc;
    for (; p + (GN_UNROLL - 1) * g.pix < p_end; p += GN_UNROLL * g.pix) {
      float v[GN_UNROLL][8];
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) {
        int cs;
        load8(gn_src(a0, a1, g, (int64_t)b * g.hw + p + u * g.pix, cc * 8, cs), v[u]);
      }
#pragma unroll
      for (int u = 0; u < GN_UNROLL; ++u) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = v[u][j] * sc[0][j] + sh[0][j];
          if (silu) y = sizeof(T) == 2 ? silu_fast(y) : silu_f(y);   // bf16 output: raw v_exp/v_rcp form
          v[u][j] = y;
        }
        store8(out + ((int64_t)b * g.hw + p + u * g.pix) * g.ctot + cc * 8, v[u]);
      }
    }
  }
  for (; p < p_end; p += g.pix) {
    const int64_t pixel = (int64_t)b * g.hw + p;
#pragma unroll
    for (int k = 0; k < GN_MAX_CHUNKS; ++k) {
      const int cc = tc + k * g.ct;
      if (cc < g.nchunks) {
        int cs;
        float v[8];
        load8(gn_src(a0, a1, g, pixel, cc * 8, cs), v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = v[j] * sc[k][j] + sh[k][j];
          if (silu) y = sizeof(T) == 2 ? silu_fast(y) : silu_f(y);   // bf16 output: raw v_exp/v_rcp form
          v[j] = y;
        }
        store8(out + pixel * g.ctot + cc * 8, v);
      }
    }
  }
}

static int gn_geom(int c0, int c1, int hw, int groups, int batch, GnGeom& g, int& nslabs, const char* who) {
  LDM_REQUIRE(c0 >= 8 && c1 >= 0 && c0 % 8 == 0 && c1 % 8 == 0, "%s: channel counts must be multiples of 8 (c0=%d c1=%d)", who, c0, c1);
  g.c0 = c0; g.c1 = c1; g.ctot = c0 + c1;
  LDM_REQUIRE(groups >= 1 && g.ctot % groups == 0, "%s: %d channels not divisible by %d groups", who, g.ctot, groups);
  g.nchunks = g.ctot / 8;
  LDM_REQUIRE(g.nchunks <= GN_THREADS * GN_MAX_CHUNKS, "%s: too many channels (%d)", who, g.ctot);
  g.ct = g.nchunks < GN_THREADS ? g.nchunks : GN_THREADS;
  if (g.nchunks > GN_THREADS) g.ct = (g.nchunks + ((g.nchunks + GN_THREADS - 1) / GN_THREADS) - 1) / ((g.nchunks + GN_THREADS - 1) / GN_THREADS);
  g.pix = GN_THREADS / g.ct;
  g.cpg = g.ctot / groups;
  g.groups = groups;
  g.hw = hw;
  // enough blocks to fill the chip (~8 per CU) without starving a block of work
  int want = (2048 + batch - 1) / (batch > 0 ? batch : 1);
  int by_work = (hw + 4 * g.pix - 1) / (4 * g.pix);
  nslabs = want < by_work ? want : by_work;
  if (nslabs > GN_MAX_SLABS) nslabs = GN_MAX_SLABS;
  if (nslabs < 1) nslabs = 1;
  g.slab_pixels = (hw + nslabs - 1) / nslabs;
  nslabs = (hw + g.slab_pixels - 1) / g.slab_pixels;
  return LDM_OK;
}

// ------------------------------------------------------------------------------------- one-kernel GroupNorm
// Small feature maps (the 32x32, 16x16 and 8x8 levels of the UNet): the slice of one (sample, set of GS adjacent groups)
// fits the registers of one workgroup, so statistics and normalisation are ONE pass over HBM (2 B read + 2 B written per
// element instead of two reads and a write) and ONE launch instead of three -- at these sizes the three-kernel form is
// bound by launch boundaries, not bandwidth (35 us for 42 MB at 16x16).  GS is the smallest group count whose channels
// fill whole 8-channel chunks.  A thread keeps one chunk column and up to GNF_MAXC pixels of it; per-channel partial sums
// meet in LDS and one wave per group folds them in double in a fixed order (deterministic).
constexpr int GNF_MAXC = 16;        // pixels per thread of the common instantiations
constexpr int GNF_MAXC_BIG = 21;    // 64x64 x 40 channels (C = 320 / 640 at the top level): 320 KB in 1024 threads' registers

struct GnFusedGeom {
  int c0, c1, ctot, cpg, groups, hw;
  int gs;        // groups per workgroup
  int cpr;       // 8-channel chunks per pixel of the workgroup's channel range
  int pix;       // pixels per pass = blockDim / cpr
  int iters;     // passes = ceil(hw / pix) <= GNF_MAXC
};

template <int NT, int MAXC>
__global__ void __launch_bounds__(NT)
gn_fused_kernel(const bf16_t* __restrict__ a0, const bf16_t* __restrict__ a1, GnFusedGeom g, float eps,
                const float* __restrict__ gamma, const float* __restrict__ beta, int silu, bf16_t* __restrict__ out,
                float* __restrict__ stats) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
  float* red = reinterpret_cast<float*>(smem_raw);          // [pix][seg][2]
  __shared__ float gstat[4][2];
  __shared__ double gpart[4][16][2];
  // (sample, group set) of this workgroup.  The group sets of one sample read and write DIFFERENT 16..80-byte pieces of the
  // SAME 128-byte lines (a pixel's channels are contiguous), and workgroups are dealt round-robin over the 8 XCDs, each with
  // its own L2: in launch order the 8 group sets of a sample sat on 8 different XCDs, every one of which fetched the whole
  // line and wrote its piece back separately (rocprofv3, 32 x 64 x 64 x 320: 230 MB fetched and 114 MB written per launch for
  // 84 + 84 MB of data).  Give every XCD a contiguous run of (sample, set) pairs instead (blocks l and l + 8 share an XCD):
  // a sample's sets then run at the same time on ONE XCD, whose L2 fetches each line once and merges the partial writes.
  int b, set;
  {
    const int nblk = gridDim.x * gridDim.y, lin = blockIdx.y * gridDim.x + blockIdx.x;
    const int q = nblk >> 3, r = nblk & 7, x = lin & 7;
    const int t = (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (lin >> 3);
    b = t / (int)gridDim.x;
    set = t - b * (int)gridDim.x;
  }
  const int seg = g.gs * g.cpg;                              // channels of this workgroup
  const int cbase = set * seg;
  const int tid = threadIdx.x;
  const int tp = tid / g.cpr, tc = tid - tp * g.cpr;
  const bool active = tp < g.pix;
  const int c = cbase + tc * 8;                              // first channel of this thread's chunk column
  const bf16_t* src;
  int cs;
  if (c < g.c0) { src = a0 + c; cs = g.c0; } else { src = a1 + (c - g.c0); cs = g.c1; }
  src += (size_t)b * g.hw * cs;

  uint4 raw[MAXC];
  float s[8], ss[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) s[j] = ss[j] = 0.f;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    if (k < g.iters) {
      const int p = k * g.pix + tp;
      raw[k] = (active && p < g.hw) ? *reinterpret_cast<const uint4*>(src + (size_t)p * cs) : make_uint4(0, 0, 0, 0);
    }
  }
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    if (k < g.iters) {
      float v[8];
      unpack8(raw[k], v);                                    // pixels past hw hold zeros: they add nothing
#pragma unroll
      for (int j = 0; j < 8; ++j) { s[j] += v[j]; ss[j] += v[j] * v[j]; }
    }
  }
  if (active) {
    float* r = red + ((size_t)tp * seg + tc * 8) * 2;
#pragma unroll
    for (int j = 0; j < 8; ++j) { r[2 * j] = s[j]; r[2 * j + 1] = ss[j]; }
  }
  __syncthreads();
  // fold in double, fixed order: the waves share the groups (wpg waves per group, each a strided set of pixel rows, a
  // lane a strided set of the group's channels), partial sums meet in LDS, one thread per group adds them in wave order
  const int wave = tid >> 6, lane = tid & 63;
  const int wpg = (NT / 64) / g.gs;                          // >= 1 (checked on the host)
  const int fgrp = wave / wpg, part = wave - fgrp * wpg;
  if (fgrp < g.gs) {
    double a = 0.0, q = 0.0;
    for (int r = part; r < g.pix; r += wpg)
      for (int ch = lane; ch < g.cpg; ch += 64) {
        const float* e = red + ((size_t)r * seg + fgrp * g.cpg + ch) * 2;
        a += (double)e[0];
        q += (double)e[1];
      }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
      a += __shfl_xor(a, o, 64);
      q += __shfl_xor(q, o, 64);
    }
    if (lane == 0) { gpart[fgrp][part][0] = a; gpart[fgrp][part][1] = q; }
  }
  __syncthreads();
  if (tid < g.gs) {
    double a = 0.0, q = 0.0;
    for (int w = 0; w < wpg; ++w) { a += gpart[tid][w][0]; q += gpart[tid][w][1]; }
    const double count = (double)g.hw * g.cpg;
    const double mean = a / count;
    double var = q / count - mean * mean;
    var = var < 0.0 ? 0.0 : var;
    const float m = (float)mean, rs = (float)(1.0 / sqrt(var + (double)eps));
    gstat[tid][0] = m;
    gstat[tid][1] = rs;
    float* o = stats + ((size_t)b * g.groups + set * g.gs + tid) * 2;
    o[0] = m;
    o[1] = rs;
  }
  __syncthreads();
  if (!active) return;
  float sc[8], sh[8];
#pragma unroll
  for (int j = 0; j < 8; ++j) {
    const int grp = (tc * 8 + j) / g.cpg;
    const float w = gstat[grp][1] * gamma[c + j];
    sc[j] = w;
    sh[j] = beta[c + j] - gstat[grp][0] * w;
  }
  bf16_t* dst = out + (size_t)b * g.hw * g.ctot + c;
#pragma unroll
  for (int k = 0; k < MAXC; ++k) {
    if (k < g.iters) {
      const int p = k * g.pix + tp;
      if (p < g.hw) {
        float v[8];
        unpack8(raw[k], v);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
          float y = v[j] * sc[j] + sh[j];
          if (silu) y = silu_fast(y);
          v[j] = y;
        }
        store8(dst + (size_t)p * g.ctot, v);
      }
    }
  }
}

// Geometry of the one-kernel form; false when the slice of a group set does not fit one workgroup.
static bool gn_fused_geom(int c0, int c1, int hw, int groups, GnFusedGeom& g, int& threads, size_t& smem) {
  const int ctot = c0 + c1;
  if (c0 % 8 || c1 % 8 || groups < 1 || ctot % groups) return false;
  g.c0 = c0; g.c1 = c1; g.ctot = ctot; g.groups = groups; g.hw = hw;
  g.cpg = ctot / groups;
  int gs = 1;
  while (gs <= 4 && (gs * g.cpg) % 8) gs *= 2;
  if (gs > 4 || groups % gs) return false;
  g.gs = gs;
  g.cpr = gs * g.cpg / 8;
  for (int nt : {256, 512, 1024}) {
    if (g.cpr > nt) continue;
    g.pix = nt / g.cpr;
    g.iters = (hw + g.pix - 1) / g.pix;
    smem = (size_t)g.pix * gs * g.cpg * 2 * sizeof(float);
    if (g.iters <= (nt == 1024 ? GNF_MAXC_BIG : GNF_MAXC) && smem <= 64 * 1024 && nt / 64 >= gs) { threads = nt; return true; }
  }
  return false;
}

// ------------------------------------------------------------------------------------- LayerNorm
constexpr int LN_MAX_CHUNKS = 8;  // per lane -> cols <= 4096
// One wave normalises ROWS rows at a time, a lane holding CPL 8-element chunks of each: all ROWS*CPL loads are in
// flight before the first reduction (one row per wave with one or two loads in flight ran at 45 % of the HBM rate).
// (ROWS, CPL) by width: C <= 512 -> (4, 1); C <= 1024 -> (2, 2); C <= 2048 -> (1, 4); C <= 4096 -> (1, 8).
template <typename T, int ROWS, int CPL>
__global__ void __launch_bounds__(256)
layernorm_kernel(const T* __restrict__ x, int64_t rows, int cols, float eps, const float* __restrict__ gamma,
                 const float* __restrict__ beta, T* __restrict__ out) {
  const int lane = threadIdx.x & 63;
  const int64_t row0 = ((int64_t)blockIdx.x * 4 + (threadIdx.x >> 6)) * ROWS;
  if (row0 >= rows) return;
  const int nchunks = cols >> 3;
  float v[ROWS][CPL][8];
#pragma unroll
  for (int r = 0; r < ROWS; ++r)
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = lane + k * 64;
      if (cc < nchunks && row0 + r < rows) load8(x + (row0 + r) * cols + cc * 8, v[r][k]);
      else {
#pragma unroll
        for (int j = 0; j < 8; ++j) v[r][k][j] = 0.f;
      }
    }
  float g[CPL][8], b[CPL][8];
#pragma unroll
  for (int k = 0; k < CPL; ++k) {
    const int cc = lane + k * 64;
    if (cc < nchunks) { load8(gamma + cc * 8, g[k]); load8(beta + cc * 8, b[k]); }
  }
  const float inv_n = 1.0f / (float)cols;
#pragma unroll
  for (int r = 0; r < ROWS; ++r) {
    float s = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k)
#pragma unroll
      for (int j = 0; j < 8; ++j) s += v[r][k][j];
    const float mean = wave_sum(s) * inv_n;
    float q = 0.f;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      if (lane + k * 64 < nchunks) {
#pragma unroll
        for (int j = 0; j < 8; ++j) { const float d = v[r][k][j] - mean; q += d * d; }
      }
    }
    const float rstd = rsqrtf(wave_sum(q) * inv_n + eps);
    if (row0 + r >= rows) continue;
    T* orow = out + (row0 + r) * cols;
#pragma unroll
    for (int k = 0; k < CPL; ++k) {
      const int cc = lane + k * 64;
      if (cc < nchunks) {
        float y[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) y[j] = (v[r][k][j] - mean) * rstd * g[k][j] + b[k][j];
        store8(orow + cc * 8, y);
      }
    }
  }
}

template <typename T, int ROWS, int CPL>
static void launch_layernorm(const void* x, int64_t rows, int cols, float eps, const float* gamma, const float* beta, void* out,
                             hipStream_t st) {
  const int64_t waves = (rows + ROWS - 1) / ROWS;
  hipLaunchKernelGGL((layernorm_kernel<T, ROWS, CPL>), dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, st, (const T*)x, rows, cols,
                     eps, gamma, beta, (T*)out);
}

template <typename T>
static void dispatch_layernorm(const void* x, int64_t rows, int cols, float eps, const float* gamma, const float* beta, void* out,
                               hipStream_t st) {
  if (cols <= 512) launch_layernorm<T, 4, 1>(x, rows, cols, eps, gamma, beta, out, st);
  else if (cols <= 1024) launch_layernorm<T, 2, 2>(x, rows, cols, eps, gamma, beta, out, st);
  else if (cols <= 2048) launch_layernorm<T, 1, 4>(x, rows, cols, eps, gamma, beta, out, st);
  else launch_layernorm<T, 1, 8>(x, rows, cols, eps, gamma, beta, out, st);
}

}  // namespace ldm

using namespace ldm;

extern "C" int ldm_groupnorm_workspace_floats(int batch, int groups) {
  return batch * groups * 2 * GN_MAX_SLABS;
}

extern "C" int ldm_groupnorm_stats(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw,
                                   int groups, float eps, float* stats, float* workspace, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_groupnorm_stats: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && hw >= 1, "ldm_groupnorm_stats: bad sizes");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(batch <= 65535, "ldm_groupnorm_stats: batch %d exceeds 65535", batch);
  LDM_REQUIRE(a0 && stats && workspace && (c1 == 0 || a1), "ldm_groupnorm_stats: null pointer");
  GnGeom g;
  int nslabs;
  int rc = gn_geom(c0, c1, hw, groups, batch, g, nslabs, "ldm_groupnorm_stats");
  if (rc) return rc;
  LDM_REQUIRE(aligned16(a0) && (c1 == 0 || aligned16(a1)), "ldm_groupnorm_stats: inputs must be 16-byte aligned");
  if (dtype == LDM_F32) LDM_REQUIRE(c0 % 4 == 0, "ldm_groupnorm_stats: bad c0");
  const size_t smem = (size_t)g.pix * g.ctot * 2 * sizeof(float);
  LDM_REQUIRE(smem <= 64 * 1024, "ldm_groupnorm_stats: %d channels need %zu bytes of LDS", g.ctot, smem);
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nslabs, batch);
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((gn_partial_kernel<bf16_t>), grid, dim3(GN_THREADS), smem, st, (const bf16_t*)a0, (const bf16_t*)a1, g, workspace);
  else
    hipLaunchKernelGGL((gn_partial_kernel<float>), grid, dim3(GN_THREADS), smem, st, (const float*)a0, (const float*)a1, g, workspace);
  LDM_HIP_CHECK_LAUNCH("ldm_groupnorm_stats(partial)");
  const int total = batch * groups;
  hipLaunchKernelGGL(gn_finalize_kernel, dim3((total + 3) / 4), dim3(256), 0, st, workspace, nslabs, groups, total,
                     (double)hw * g.cpg, eps, stats);
  LDM_HIP_CHECK_LAUNCH("ldm_groupnorm_stats(finalize)");
  return LDM_OK;
}

extern "C" int ldm_groupnorm_apply(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw,
                                   int groups, const float* stats, const float* gamma, const float* beta, int silu,
                                   void* out, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_groupnorm_apply: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && hw >= 1, "ldm_groupnorm_apply: bad sizes");
  if (batch == 0) return LDM_OK;
  LDM_REQUIRE(batch <= 65535, "ldm_groupnorm_apply: batch %d exceeds 65535", batch);
  LDM_REQUIRE(a0 && stats && gamma && beta && out && (c1 == 0 || a1), "ldm_groupnorm_apply: null pointer");
  GnGeom g;
  int nslabs;
  int rc = gn_geom(c0, c1, hw, groups, batch, g, nslabs, "ldm_groupnorm_apply");
  if (rc) return rc;
  LDM_REQUIRE(aligned16(a0) && aligned16(out) && (c1 == 0 || aligned16(a1)), "ldm_groupnorm_apply: buffers must be 16-byte aligned");
  hipStream_t st = (hipStream_t)stream;
  dim3 grid(nslabs, batch);
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((gn_apply_kernel<bf16_t>), grid, dim3(GN_THREADS), 0, st, (const bf16_t*)a0, (const bf16_t*)a1, g, stats,
                       gamma, beta, silu, (bf16_t*)out);
  else
    hipLaunchKernelGGL((gn_apply_kernel<float>), grid, dim3(GN_THREADS), 0, st, (const float*)a0, (const float*)a1, g, stats,
                       gamma, beta, silu, (float*)out);
  LDM_HIP_CHECK_LAUNCH("ldm_groupnorm_apply");
  return LDM_OK;
}

extern "C" int ldm_groupnorm(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw, int groups,
                             float eps, const float* gamma, const float* beta, int silu, void* out, float* stats,
                             float* workspace, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_groupnorm: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && hw >= 1, "ldm_groupnorm: bad sizes");
  if (batch == 0) return LDM_OK;
  GnFusedGeom g;
  int threads = 0;
  size_t smem = 0;
  if (dtype == LDM_BF16 && batch <= 65535 && gn_fused_geom(c0, c1, hw, groups, g, threads, smem)) {
    LDM_REQUIRE(a0 && stats && gamma && beta && out && (c1 == 0 || a1), "ldm_groupnorm: null pointer");
    LDM_REQUIRE(aligned16(a0) && aligned16(out) && (c1 == 0 || aligned16(a1)), "ldm_groupnorm: buffers must be 16-byte aligned");
    hipStream_t st = (hipStream_t)stream;
    const dim3 grid(groups / g.gs, batch);
#define LDM_GN_FUSED(NT, MAXC) \
  hipLaunchKernelGGL((gn_fused_kernel<NT, MAXC>), grid, dim3(NT), smem, st, (const bf16_t*)a0, (const bf16_t*)a1, g, eps, gamma, \
                     beta, silu, (bf16_t*)out, stats)
    if (threads == 256) LDM_GN_FUSED(256, GNF_MAXC);
    else if (threads == 512) LDM_GN_FUSED(512, GNF_MAXC);
    else if (g.iters <= GNF_MAXC) LDM_GN_FUSED(1024, GNF_MAXC);
    else LDM_GN_FUSED(1024, GNF_MAXC_BIG);
#undef LDM_GN_FUSED
    LDM_HIP_CHECK_LAUNCH("ldm_groupnorm(one kernel)");
    return LDM_OK;
  }
  const int rc = ldm_groupnorm_stats(a0, a1, c0, c1, dtype, batch, hw, groups, eps, stats, workspace, stream);
  if (rc) return rc;
  return ldm_groupnorm_apply(a0, a1, c0, c1, dtype, batch, hw, groups, stats, gamma, beta, silu, out, stream);
}

// ------------------------------------------------------------------- GroupNorm without a caller workspace
// ldm_groupnorm_noalloc: the partial-sum workspace of the three-kernel form lives in a file-scope device array of this code
// object, so the call neither takes nor allocates scratch memory.  `scratch` is only a host-side stand-in for the symbol:
// the device address is looked up once per GPU (every GPU holds its own copy at its own address) and that pointer, never
// the array's name, is what the kernels receive.
static __device__ __align__(256) float scratch[LDM_GROUPNORM_NOALLOC_MAX_FLOATS];

namespace {
constexpr int kMaxDevices = 64;
std::mutex g_scratch_mutex;
float* g_scratch_ptr[kMaxDevices] = {};

int scratch_for_current_device(hipStream_t stream, float** out) {
  int dev = -1;
  hipError_t e = hipGetDevice(&dev);
  if (e != hipSuccess || dev < 0 || dev >= kMaxDevices) {
    set_error("ldm_groupnorm_noalloc: hipGetDevice: %s (device %d)", hipGetErrorString(e), dev);
    return LDM_EHIP;
  }
  std::lock_guard<std::mutex> lock(g_scratch_mutex);
  if (g_scratch_ptr[dev] == nullptr) {
    // The symbol lookup may load the code object: not a stream operation, so it must not happen while `stream` is being
    // captured into a graph.  Callers that capture make one ordinary call per device first (include/ldm_amd.h).
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(stream, &cs) == hipSuccess && cs != hipStreamCaptureStatusNone) {
      set_error("ldm_groupnorm_noalloc: first call on device %d is under stream capture; call it once outside the capture first", dev);
      return LDM_EINVAL;
    }
    void* ptr = nullptr;
    e = hipGetSymbolAddress(&ptr, HIP_SYMBOL(scratch));
    if (e != hipSuccess || ptr == nullptr) {
      set_error("ldm_groupnorm_noalloc: hipGetSymbolAddress(scratch) on device %d: %s", dev, hipGetErrorString(e));
      return LDM_EHIP;
    }
    g_scratch_ptr[dev] = static_cast<float*>(ptr);
  }
  *out = g_scratch_ptr[dev];
  return LDM_OK;
}
}  // namespace

extern "C" int ldm_groupnorm_noalloc(const void* a0, const void* a1, int c0, int c1, int dtype, int batch, int hw, int groups,
                                     float eps, const float* gamma, const float* beta, int silu, void* out, float* stats,
                                     void* stream) {
  LDM_REQUIRE(batch >= 0 && groups >= 1, "ldm_groupnorm_noalloc: bad sizes");
  if (batch == 0) return LDM_OK;
  const long long need = (long long)batch * groups * 2 * GN_MAX_SLABS;
  LDM_REQUIRE(need <= (long long)LDM_GROUPNORM_NOALLOC_MAX_FLOATS,
              "ldm_groupnorm_noalloc: batch %d x %d groups needs %lld scratch floats, the built-in buffer holds %d "
              "(batch * groups <= %d); use ldm_groupnorm with a caller workspace",
              batch, groups, need, (int)LDM_GROUPNORM_NOALLOC_MAX_FLOATS, (int)(LDM_GROUPNORM_NOALLOC_MAX_FLOATS / (2 * GN_MAX_SLABS)));
  float* ws = nullptr;
  const int rc = scratch_for_current_device((hipStream_t)stream, &ws);
  if (rc) return rc;
  // Nothing in the buffer has to start at zero: gn_partial_kernel writes every (sample, slab, group) pair that
  // gn_finalize_kernel reads in the same call.
  return ldm_groupnorm(a0, a1, c0, c1, dtype, batch, hw, groups, eps, gamma, beta, silu, out, stats, ws, stream);
}

extern "C" int ldm_layernorm(const void* x, int dtype, int64_t rows, int cols, float eps, const float* gamma,
                             const float* beta, void* out, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_layernorm: bad dtype %d", dtype);
  LDM_REQUIRE(rows >= 0 && cols >= 8 && cols % 8 == 0 && cols <= 8 * 64 * LN_MAX_CHUNKS,
              "ldm_layernorm: cols must be a multiple of 8 in [8, %d] (got %d)", 8 * 64 * LN_MAX_CHUNKS, cols);
  if (rows == 0) return LDM_OK;
  LDM_REQUIRE(x && gamma && beta && out, "ldm_layernorm: null pointer");
  LDM_REQUIRE(aligned16(x) && aligned16(out) && aligned16(gamma) && aligned16(beta), "ldm_layernorm: buffers must be 16-byte aligned");
  LDM_REQUIRE((rows + 3) / 4 < (1LL << 31), "ldm_layernorm: too many rows");
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LDM_BF16) dispatch_layernorm<bf16_t>(x, rows, cols, eps, gamma, beta, out, st);
  else dispatch_layernorm<float>(x, rows, cols, eps, gamma, beta, out, st);
  LDM_HIP_CHECK_LAUNCH("ldm_layernorm");
  return LDM_OK;
}
