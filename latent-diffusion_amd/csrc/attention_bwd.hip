// Backward of the flash-style attention (no mask, scale 1/sqrt(d)), bf16 on v_mfma_f32_16x16x32_bf16.
//
// With c = log2(e)/sqrt(d), L[q] = log2 sum_k exp2(c s[q][k]) saved by the forward (ldm_attention_train) and
// Dl[q] = sum_j dO[q][j] O[q][j]:
//     P = exp2(c S - L),   dV = P^T dO,   dP = dO V^T,   dS = P o (dP - Dl) / sqrt(d),   dQ = dS K,   dK = dS^T Q.
// No atomics (deterministic): two kernels recompute S and dP.
//   * attn_bwd_dq_kernel: a workgroup owns 128 queries (a wave 32, as two 16-query groups; 256 and four at d = 40) and sweeps the keys in tiles
//     of 64 -- the forward kernel's shape.  Transposed orientation: S^T = K Q^T and dP^T = V dO^T give every lane one
//     query column, so L and Dl are per-lane scalars, and dS^T packed to bf16 is directly the B operand of
//     dQ^T += K^T dS^T (K^T fragments by transposed LDS reads of the same K image).
//   * attn_bwd_dkv_kernel: the mirror image.  A workgroup owns 128 keys (a wave 32) whose K and V fragments stay in
//     registers as B operands, and sweeps the queries in tiles of 64 staged in LDS (Q and dO images, L and Dl vectors):
//     S = Q K^T and dP = dO V^T give every lane one key column; P and dS packed to bf16 are the B operands of
//     dV^T += dO^T P and dK^T += Q^T dS (Q^T / dO^T by transposed reads).
// One LDS image per operand serves both the row reads (ds_read_b128, contraction over d) and the transposed reads
// (ds_read_b64_tr_b16, contraction over rows): row pitch 2*DK + 32 bytes.
// d = 40 carries -L and -Dl inside the score products (AbGeom::AUG): the loop is one exponential and one multiply per score.
#include "common.h"

namespace ldm {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ uint32_t ab_cvt_pk_bf16(float a, float b) { return pack_bf16x2(a, b); }

struct AttnBwdParams {
  const void* q; const void* k; const void* v; const void* o; const void* d_o;
  void* dq; void* dk; void* dv;
  int q_stride, kv_stride, o_stride, do_stride, dq_stride, dkv_stride;
  int batch, heads, lq, lk;
  float scale_log2e, scale;
  const float* lse;    // [batch][heads][lq]
  float* delta;        // [batch][heads][lq]  (written by the prep kernel)
};

template <int D>
struct AbGeom {
  static constexpr int KS = (D + 31) / 32;
  static constexpr int DK = KS * 32;
  static constexpr int DT = (D + 15) / 16;
  static constexpr int ROW = DK * 2 + 32;            // bytes per image row: 32 (mod 64) keeps the row reads (ds_read_b128, four
                                                     // 16-lane groups) AND the transposed reads (two 32-lane halves) conflict-free;
                                                     // + 16 made both 2-way (SQ_LDS_BANK_CONFLICT = half of SQ_LDS_IDX_ACTIVE)
  static constexpr int IMG = 64 * ROW;               // one 64-row image
  static constexpr int CPR = D / 8;
  static constexpr int NCH = (64 * CPR + 255) / 256;
  static constexpr int NBUF = D <= 80 ? 2 : 1;       // double-buffered tiles when they fit comfortably
  // Head dims whose padding to 32 leaves three spare columns (d = 40, 80) carry the per-row terms of the exponent INSIDE the
  // score products, as attention32.hip does with its running maximum: the operand swept through LDS and the one held in
  // registers get two extra columns (-hi(L), -lo(L)) x (1, 1), so the matrix core delivers c s - L (and dP - Dl) and the loop
  // keeps one v_exp_f32 and one multiply per score.  (The plain form spends 197 / 154 vector instructions per 64 x 32 tile of
  // the dQ / dK-dV kernel on 44 / 56 MFMAs and is bound by their issue, not by the matrix cores.)  hi + lo are two bf16
  // terms: L to 2^-17 relative, far below what a bf16 P resolves.
  static constexpr bool AUG = DK - D >= 3 && D <= 40;   // (d = 80 has the columns but not the registers: the dQ kernel spills)
  static constexpr int NGQ = D == 40 ? 4 : 2;     // 16-query groups per wave of the dQ kernel: four at d = 40 (every K / V fragment read
                                                  // from LDS feeds four MFMAs; 256 registers, two waves per SIMD: -11 % on the launch)
  static constexpr int NGK = D > 40 ? 1 : 2;      // 16-key groups per wave of the dK/dV kernel (one for the wide heads: the accumulators of
                                                  // two would spill; four at d = 40 need one wave per SIMD and ran 20 % slower)
  static constexpr int AUG_KS = D / 32, AUG_FQ = (D % 32) / 8;   // where column D sits in a register fragment (element 0)
};

constexpr float AB_MASK_BIG = 30000.0f;   // exponent offset of rows that must not contribute: exp2(-30000) == 0

// bf16 pair (-hi, -lo), hi + lo = x to 2^-17 relative (x finite)
__device__ __forceinline__ uint32_t ab_neg_split(float x) {
  const float hi = bf16_to_f32(f32_to_bf16(x));
  return pack_bf16x2(-hi, -(x - hi));
}
// fragment elements times a scalar, rounded back to bf16 (the forward rounds its scaled Q the same way)
__device__ __forceinline__ bf16x8_t ab_scale_frag(bf16x8_t z, float c) {
#pragma unroll
  for (int e = 0; e < 8; e += 2) {
    const uint32_t w = pack_bf16x2(bf16_to_f32((bf16_t)z[e]) * c, bf16_to_f32((bf16_t)z[e + 1]) * c);
    z[e] = (short)(w & 0xffffu);
    z[e + 1] = (short)(w >> 16);
  }
  return z;
}

// The wave-uniform tile offset goes into the DESCRIPTOR (base advanced, record count shrunk), not into the instruction's
// scalar-offset field: LLVM documents soffset as EXCLUDED from the range check of raw buffer loads, and the rows past lk
// of the last, ragged tile must come back as zeros (never as whatever lies behind this (sample, head)'s rows).
__device__ __forceinline__ uint4 ab_buffer_load16(const void* base, unsigned bytes, unsigned voff, unsigned soff) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const unsigned left = bytes - soff;   // callers fetch existing tiles only: soff < bytes (a clamped form compiles to a VALU saturating subtract and drags the descriptor into a waterfall loop)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(static_cast<const unsigned char*>(base) + soff), 0, (int)left, 0x00020000);
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// Dl[b][h][q] = sum_j dO[q][h*d + j] * O[q][h*d + j].  One thread per (b, q, 8-column chunk): consecutive threads read
// consecutive 16-byte chunks of a row (coalesced), the D/8 chunk products of a head are folded through LDS in a fixed
// order.  (One wave per (b, h, q) row with d = 40 kept 24 of 64 lanes idle and moved 84 MB in 90 us.)
template <int D>
__global__ void __launch_bounds__(256)
attn_bwd_prep_vec_kernel(const AttnBwdParams p) {
  constexpr int CPH = D / 8;                        // chunks per head
  constexpr int SPAN = (256 / CPH) * CPH;           // chunks per block: whole heads only (the fold stays inside the block)
  __shared__ float part[256 + CPH];
  const int cpr = p.heads * CPH;                    // chunks per row
  const long long total = (long long)p.batch * p.lq * cpr;
  const long long idx = (long long)blockIdx.x * SPAN + threadIdx.x;
  const bool mine = (int)threadIdx.x < SPAN && idx < total;
  float s = 0.f;
  if (mine) {
    const long long row = idx / cpr;                // b * lq + q
    const int ch = (int)(idx - row * cpr);
    const bf16_t* o = static_cast<const bf16_t*>(p.o) + row * p.o_stride + ch * 8;
    const bf16_t* g = static_cast<const bf16_t*>(p.d_o) + row * p.do_stride + ch * 8;
    float a[8], c[8];
    load8(o, a);
    load8(g, c);
#pragma unroll
    for (int e = 0; e < 8; ++e) s += a[e] * c[e];
  }
  part[threadIdx.x] = s;
  __syncthreads();
  // the thread of a head's first chunk adds its head's CPH partials, in chunk order
  if (mine) {
    const long long row = idx / cpr;
    const int ch = (int)(idx - row * cpr);
    if (ch % CPH == 0) {
      float t = 0.f;
#pragma unroll
      for (int k = 0; k < CPH; ++k) t += part[threadIdx.x + k];
      const int h = ch / CPH;
      const long long b = row / p.lq;
      const int qi = (int)(row - b * p.lq);
      p.delta[(b * p.heads + h) * p.lq + qi] = t;
    }
  }
}

// generic shape of the same (any d, unaligned): one wave per (b, h, q) row
__global__ void __launch_bounds__(256)
attn_bwd_prep_kernel(const AttnBwdParams p, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // over batch*heads*lq
  const int lane = threadIdx.x & 63;
  if (row >= (long long)p.batch * p.heads * p.lq) return;
  const int qi = (int)(row % p.lq);
  const long long bh = row / p.lq;
  const int h = (int)(bh % p.heads);
  const long long b = bh / p.heads;
  const bf16_t* o = static_cast<const bf16_t*>(p.o) + (b * p.lq + qi) * p.o_stride + h * d;
  const bf16_t* g = static_cast<const bf16_t*>(p.d_o) + (b * p.lq + qi) * p.do_stride + h * d;
  float s = 0.f;
  for (int j = lane; j < d; j += 64) s += bf16_to_f32(o[j]) * bf16_to_f32(g[j]);
  s = wave_sum(s);
  if (lane == 0) p.delta[row] = s;
}

// Stage a 64-row tile of one operand (rows row0.., head slice of D columns) into registers / commit to an LDS image.
template <int D>
struct TileStager {
  using G = AbGeom<D>;
  unsigned g_off[G::NCH];
  int l_off[G::NCH];
  __device__ __forceinline__ void init(int tid, int stride) {
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
      const int idx = c * 256 + tid;
      const int row = idx / G::CPR, ch = idx - row * G::CPR;
      const bool ok = idx < 64 * G::CPR;
      g_off[c] = ok ? ((unsigned)row * stride + ch * 8) * 2u : 0x80000000u;
      l_off[c] = ok ? row * G::ROW + ch * 16 : -1;
    }
  }
  __device__ __forceinline__ void fetch(const void* base, unsigned bytes, unsigned soff, uint4 (&r)[G::NCH]) const {
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) r[c] = ab_buffer_load16(base, bytes, g_off[c], soff);
  }
  __device__ __forceinline__ void commit(unsigned char* img, const uint4 (&r)[G::NCH]) const {
#pragma unroll
    for (int c = 0; c < G::NCH; ++c)
      if (l_off[c] >= 0) *reinterpret_cast<uint4*>(img + l_off[c]) = r[c];
  }
};

// fragments of a 64-row image
template <int D>
__device__ __forceinline__ bf16x8_t ab_row_frag(const unsigned char* img, int rb, int ks, int fr, int fq) {
  return *reinterpret_cast<const bf16x8_t*>(img + (rb * 16 + fr) * AbGeom<D>::ROW + (ks * 4 + fq) * 16);
}
template <int D>
__device__ __forceinline__ bf16x8_t ab_tr_frag(const unsigned char* img, int dt, int st, int fr, int fq) {
  const int tq = fr >> 2, tp = fr & 3;
  const unsigned char* a0 = img + ((2 * st) * 16 + 4 * fq + tq) * AbGeom<D>::ROW + (dt * 16 + 4 * tp) * 2;
  const unsigned char* a1 = a0 + 16 * AbGeom<D>::ROW;
  const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)a0);
  const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)a1);
  return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
}
// four accumulator tiles (contraction rows kb = 0..3, 4 values each) -> two B operands (st = 0, 1)
__device__ __forceinline__ void ab_pack(const float (&v)[4][4], bf16x8_t (&pb)[2]) {
#pragma unroll
  for (int st = 0; st < 2; ++st) {
    typedef __attribute__((ext_vector_type(4))) uint32_t u4;
    u4 w;
    w[0] = ab_cvt_pk_bf16(v[2 * st][0], v[2 * st][1]);
    w[1] = ab_cvt_pk_bf16(v[2 * st][2], v[2 * st][3]);
    w[2] = ab_cvt_pk_bf16(v[2 * st + 1][0], v[2 * st + 1][1]);
    w[3] = ab_cvt_pk_bf16(v[2 * st + 1][2], v[2 * st + 1][3]);
    pb[st] = __builtin_bit_cast(bf16x8_t, w);
  }
}

// B-operand fragments of 16 rows held in registers: lane (row fr, k-group fq) holds X[row][32*ks + 8*fq + 0..7]
template <int D>
__device__ __forceinline__ void ab_load_bfrag(const bf16_t* base, int row, int nrows, int stride, int fq, bf16x8_t (&f)[AbGeom<D>::KS]) {
#pragma unroll
  for (int ks = 0; ks < AbGeom<D>::KS; ++ks) {
    const int col = ks * 32 + fq * 8;
    bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
    if (row < nrows && col < D) z = *reinterpret_cast<const bf16x8_t*>(base + (size_t)row * stride + col);
    f[ks] = z;
  }
}

// ------------------------------------------------------------------------------------------ dQ
template <int D>
__global__ void __launch_bounds__(256, (D <= 80 ? 2 : 1))
attn_bwd_dq_kernel(const AttnBwdParams p) {
  using G = AbGeom<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NBUF x (K image, V image)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  constexpr int NGQ = G::NGQ;
  const int q0 = blockIdx.x * (64 * NGQ) + wave * (16 * NGQ);
  const bf16_t* qp = static_cast<const bf16_t*>(p.q) + (size_t)b * p.lq * p.q_stride + h * D;
  const bf16_t* gp = static_cast<const bf16_t*>(p.d_o) + (size_t)b * p.lq * p.do_stride + h * D;
  const bf16_t* kp = static_cast<const bf16_t*>(p.k) + (size_t)b * p.lk * p.kv_stride + h * D;
  const bf16_t* vp = static_cast<const bf16_t*>(p.v) + (size_t)b * p.lk * p.kv_stride + h * D;
  bf16_t* dqp = static_cast<bf16_t*>(p.dq) + (size_t)b * p.lq * p.dq_stride + h * D;
  const unsigned kv_bytes = ((unsigned)(p.lk - 1) * p.kv_stride + D) * 2u;

  for (int i = tid * 16; i < G::NBUF * 2 * G::IMG; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  bf16x8_t qf[NGQ][G::KS], gf[NGQ][G::KS];
  float Lq[NGQ], Dq[NGQ];
#pragma unroll
  for (int g = 0; g < NGQ; ++g) {
    const int qi = q0 + g * 16 + fr;
    ab_load_bfrag<D>(qp, qi, p.lq, p.q_stride, fq, qf[g]);
    ab_load_bfrag<D>(gp, qi, p.lq, p.do_stride, fq, gf[g]);
    const bool ok = qi < p.lq;
    Lq[g] = ok ? p.lse[((size_t)b * p.heads + h) * p.lq + qi] : INFINITY;   // invalid queries: P = 0
    Dq[g] = ok ? p.delta[((size_t)b * p.heads + h) * p.lq + qi] : 0.f;
    if constexpr (G::AUG) {
      // Q carries c and (-L, -1 for the mask column of K); dO carries -Dl: the K / V images hold 1.0 in those columns
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) qf[g][ks] = ab_scale_frag(qf[g][ks], p.scale_log2e);
      if (fq == G::AUG_FQ) {
        const uint32_t wl = ab_neg_split(ok ? Lq[g] : AB_MASK_BIG), wd = ab_neg_split(Dq[g]);
        qf[g][G::AUG_KS][0] = (short)(wl & 0xffffu);
        qf[g][G::AUG_KS][1] = (short)(wl >> 16);
        qf[g][G::AUG_KS][2] = (short)0xBF80;
        gf[g][G::AUG_KS][0] = (short)(wd & 0xffffu);
        gf[g][G::AUG_KS][1] = (short)(wd >> 16);
      }
    }
  }
  f32x4_t acc[NGQ][G::DT];
#pragma unroll
  for (int g = 0; g < NGQ; ++g)
#pragma unroll
    for (int t = 0; t < G::DT; ++t) acc[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  TileStager<D> stg;
  stg.init(tid, p.kv_stride);
  uint4 kreg[G::NCH], vreg[G::NCH];
  const int ntiles = (p.lk + 63) / 64;
  auto fetch = [&](int t) {
    const unsigned soff = (unsigned)t * 64u * p.kv_stride * 2u;
    stg.fetch(kp, kv_bytes, soff, kreg);
    stg.fetch(vp, kv_bytes, soff, vreg);
  };
  const int ragged = p.lk & 63;
  auto commit = [&](unsigned char* buf, int t) {
    stg.commit(buf, kreg);
    stg.commit(buf + G::IMG, vreg);
    // keys past lk of the last, ragged tile: +BIG in the mask column (times Q's -1); the buffer is not used again
    if (G::AUG && ragged != 0 && t == ntiles - 1 && tid < 64 && tid >= ragged)
      *reinterpret_cast<bf16_t*>(buf + tid * G::ROW + (D + 2) * 2) = (bf16_t)(__float_as_uint(AB_MASK_BIG) >> 16);
  };
  __syncthreads();
  if constexpr (G::AUG) {   // the constant columns D, D + 1 of every K and V image (a commit writes columns < D only)
    for (int i = tid; i < G::NBUF * 2 * 64; i += 256)
      *reinterpret_cast<uint32_t*>(smem + (i >> 6) * G::IMG + (i & 63) * G::ROW + D * 2) = 0x3F803F80u;
    __syncthreads();
  }
  fetch(0);
  commit(smem, 0);
  if (ntiles > 1) fetch(1);
  for (int t = 0; t < ntiles; ++t) {
    unsigned char* cur = smem + (G::NBUF == 2 ? (t & 1) : 0) * 2 * G::IMG;
    __syncthreads();
    if (G::NBUF == 2 && t + 1 < ntiles) {
      commit(smem + ((t + 1) & 1) * 2 * G::IMG, t + 1);
      if (t + 2 < ntiles) fetch(t + 2);
    }
    const unsigned char* kimg = cur;
    const unsigned char* vimg = cur + G::IMG;
    const int t0 = t * 64;
#pragma unroll
    for (int g = 0; g < NGQ; ++g) {
      f32x4_t s[4], dp[4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) {
        s[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
        dp[kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          s[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab_row_frag<D>(kimg, kb, ks, fr, fq), qf[g][ks], s[kb], 0, 0, 0);
          dp[kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab_row_frag<D>(vimg, kb, ks, fr, fq), gf[g][ks], dp[kb], 0, 0, 0);
        }
      }
      float ds[4][4];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
          if constexpr (G::AUG) {
            ds[kb][r] = __builtin_amdgcn_exp2f(s[kb][r]) * dp[kb][r];      // 1/sqrt(d) waits for the store
          } else {
            const bool valid = t0 + kb * 16 + fq * 4 + r < p.lk;
            const float pr = valid ? __builtin_amdgcn_exp2f(fmaf(s[kb][r], p.scale_log2e, -Lq[g])) : 0.f;
            ds[kb][r] = pr * (dp[kb][r] - Dq[g]) * p.scale;
          }
        }
      bf16x8_t pb[2];
      ab_pack(ds, pb);
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int st = 0; st < 2; ++st)
          acc[g][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab_tr_frag<D>(kimg, dt, st, fr, fq), pb[st], acc[g][dt], 0, 0, 0);
    }
    if (G::NBUF == 1 && t + 1 < ntiles) {
      __syncthreads();
      commit(smem, t + 1);
      if (t + 2 < ntiles) fetch(t + 2);
    }
  }
#pragma unroll
  for (int g = 0; g < NGQ; ++g) {
    const int qi = q0 + g * 16 + fr;
    if (qi >= p.lq) continue;
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) {
      const int col = dt * 16 + fq * 4;
      if (col < D) {
        if constexpr (G::AUG) acc[g][dt] *= p.scale;
        uint2 w;
        w.x = ab_cvt_pk_bf16(acc[g][dt][0], acc[g][dt][1]);
        w.y = ab_cvt_pk_bf16(acc[g][dt][2], acc[g][dt][3]);
        *reinterpret_cast<uint2*>(dqp + (size_t)qi * p.dq_stride + col) = w;
      }
    }
  }
}

// --------------------------------------------------------------------------------------- dK, dV
template <int D>
__global__ void __launch_bounds__(256, (D <= 128 && AbGeom<D>::NGK <= 2 ? 2 : 1))
attn_bwd_dkv_kernel(const AttnBwdParams p) {
  using G = AbGeom<D>;
  constexpr int NG = G::NGK;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];   // NBUF x (Q image, dO image, L[64], Dl[64])
  constexpr int BUF = 2 * G::IMG + 512;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  const int b = blockIdx.z, h = blockIdx.y;
  const int k0 = blockIdx.x * (64 * NG) + wave * (16 * NG);
  const bf16_t* qp = static_cast<const bf16_t*>(p.q) + (size_t)b * p.lq * p.q_stride + h * D;
  const bf16_t* gp = static_cast<const bf16_t*>(p.d_o) + (size_t)b * p.lq * p.do_stride + h * D;
  const bf16_t* kp = static_cast<const bf16_t*>(p.k) + (size_t)b * p.lk * p.kv_stride + h * D;
  const bf16_t* vp = static_cast<const bf16_t*>(p.v) + (size_t)b * p.lk * p.kv_stride + h * D;
  bf16_t* dkp = static_cast<bf16_t*>(p.dk) + (size_t)b * p.lk * p.dkv_stride + h * D;
  bf16_t* dvp = static_cast<bf16_t*>(p.dv) + (size_t)b * p.lk * p.dkv_stride + h * D;
  const float* lsep = p.lse + ((size_t)b * p.heads + h) * p.lq;
  const float* dlp = p.delta + ((size_t)b * p.heads + h) * p.lq;
  const unsigned q_bytes = ((unsigned)(p.lq - 1) * p.q_stride + D) * 2u;
  const unsigned g_bytes = ((unsigned)(p.lq - 1) * p.do_stride + D) * 2u;

  for (int i = tid * 16; i < G::NBUF * BUF; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);

  bf16x8_t kf[NG][G::KS], vf[NG][G::KS];
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int ki = k0 + g * 16 + fr;
    ab_load_bfrag<D>(kp, ki, p.lk, p.kv_stride, fq, kf[g]);
    ab_load_bfrag<D>(vp, ki, p.lk, p.kv_stride, fq, vf[g]);
    if constexpr (G::AUG) {   // K carries c; columns D, D + 1 of K and V are 1.0 (times the -L / -Dl pairs of the Q / dO images)
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) kf[g][ks] = ab_scale_frag(kf[g][ks], p.scale_log2e);
      if (fq == G::AUG_FQ) {
        kf[g][G::AUG_KS][0] = kf[g][G::AUG_KS][1] = (short)0x3F80;
        vf[g][G::AUG_KS][0] = vf[g][G::AUG_KS][1] = (short)0x3F80;
      }
    }
  }
  f32x4_t dk[NG][G::DT], dv[NG][G::DT];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int t = 0; t < G::DT; ++t) { dk[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; dv[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f}; }

  TileStager<D> sq, sg;
  sq.init(tid, p.q_stride);
  sg.init(tid, p.do_stride);
  uint4 qreg[G::NCH], greg[G::NCH];
  // L and Dl of the tile's 64 queries: wave 0 carries L, wave 1 Dl, by ONE unconditional buffer load each.  Queries past lq
  // read 0, and so do their Q and dO rows: P = 1 meets dO = 0 and dS = 1 * (0 - 0), nothing reaches dK / dV.  (As
  // `if (tid < 64) l = qi < lq ? lse[qi] : ...` the select's default overwrote a register that, for all the compiler
  // could prove, still had a load in flight: it put s_waitcnt vmcnt(0) right behind the tile loads just issued, wave 0 sat
  // out a memory latency per tile and the other three waves waited for it at the barrier -- 16 % of the kernel.)
  const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
  const float* const ld_base = wv == 0 ? lsep : dlp;
  const unsigned ld_off = wv < 2 ? (unsigned)lane * 4u : 0x80000000u;
  float ldreg = 0.f;
  const int ntiles = (p.lq + 63) / 64;
  auto fetch = [&](int t) {
    sq.fetch(qp, q_bytes, (unsigned)t * 64u * p.q_stride * 2u, qreg);
    sg.fetch(gp, g_bytes, (unsigned)t * 64u * p.do_stride * 2u, greg);
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(ld_base + t * 64), 0, (p.lq - t * 64) * 4, 0x00020000);
    ldreg = __uint_as_float((unsigned)__builtin_amdgcn_raw_buffer_load_b32(rs, (int)ld_off, 0, 0));
  };
  auto commit = [&](unsigned char* buf) {
    sq.commit(buf, qreg);
    sg.commit(buf + G::IMG, greg);
    if (wv < 2) {
      if constexpr (G::AUG) *reinterpret_cast<uint32_t*>(buf + wv * G::IMG + lane * G::ROW + D * 2) = ab_neg_split(ldreg);
      else reinterpret_cast<float*>(buf + 2 * G::IMG + wv * 256)[lane] = ldreg;
    }
  };
  __syncthreads();
  fetch(0);
  commit(smem);
  if (ntiles > 1) fetch(1);
  for (int t = 0; t < ntiles; ++t) {
    unsigned char* cur = smem + (G::NBUF == 2 ? (t & 1) : 0) * BUF;
    __syncthreads();
    if (G::NBUF == 2 && t + 1 < ntiles) {
      commit(smem + ((t + 1) & 1) * BUF);
      if (t + 2 < ntiles) fetch(t + 2);
    }
    const unsigned char* qimg = cur;
    const unsigned char* gimg = cur + G::IMG;
    const float* lt = reinterpret_cast<const float*>(cur + 2 * G::IMG);
    const float* dlt = reinterpret_cast<const float*>(cur + 2 * G::IMG + 256);
#pragma unroll
    for (int g = 0; g < NG; ++g) {
      // S[query rows qb*16 + fq*4 + r][key column fr] and dP alike
      float pv[4][4], ds[4][4];
#pragma unroll
      for (int qb = 0; qb < 4; ++qb) {
        f32x4_t s = {0.f, 0.f, 0.f, 0.f}, dp = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int ks = 0; ks < G::KS; ++ks) {
          s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab_row_frag<D>(qimg, qb, ks, fr, fq), kf[g][ks], s, 0, 0, 0);
          dp = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab_row_frag<D>(gimg, qb, ks, fr, fq), vf[g][ks], dp, 0, 0, 0);
        }
        if constexpr (G::AUG) {
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(s[r]);
            pv[qb][r] = pr;
            ds[qb][r] = pr * dp[r];                                         // 1/sqrt(d) waits for the store
          }
        } else {
          const float4 l4 = *reinterpret_cast<const float4*>(lt + qb * 16 + fq * 4);
          const float4 d4 = *reinterpret_cast<const float4*>(dlt + qb * 16 + fq * 4);
          const float lv[4] = {l4.x, l4.y, l4.z, l4.w}, dl[4] = {d4.x, d4.y, d4.z, d4.w};
#pragma unroll
          for (int r = 0; r < 4; ++r) {
            const float pr = __builtin_amdgcn_exp2f(fmaf(s[r], p.scale_log2e, -lv[r]));
            pv[qb][r] = pr;
            ds[qb][r] = pr * (dp[r] - dl[r]) * p.scale;
          }
        }
      }
      bf16x8_t pbp[2], pbs[2];
      ab_pack(pv, pbp);
      ab_pack(ds, pbs);
#pragma unroll
      for (int dt = 0; dt < G::DT; ++dt)
#pragma unroll
        for (int st = 0; st < 2; ++st) {
          dv[g][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab_tr_frag<D>(gimg, dt, st, fr, fq), pbp[st], dv[g][dt], 0, 0, 0);
          dk[g][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ab_tr_frag<D>(qimg, dt, st, fr, fq), pbs[st], dk[g][dt], 0, 0, 0);
        }
    }
    if (G::NBUF == 1 && t + 1 < ntiles) {
      __syncthreads();
      commit(smem);
      if (t + 2 < ntiles) fetch(t + 2);
    }
  }
#pragma unroll
  for (int g = 0; g < NG; ++g) {
    const int ki = k0 + g * 16 + fr;
    if (ki >= p.lk) continue;
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) {
      const int col = dt * 16 + fq * 4;
      if (col < D) {
        if constexpr (G::AUG) dk[g][dt] *= p.scale;
        uint2 w;
        w.x = ab_cvt_pk_bf16(dk[g][dt][0], dk[g][dt][1]);
        w.y = ab_cvt_pk_bf16(dk[g][dt][2], dk[g][dt][3]);
        *reinterpret_cast<uint2*>(dkp + (size_t)ki * p.dkv_stride + col) = w;
        w.x = ab_cvt_pk_bf16(dv[g][dt][0], dv[g][dt][1]);
        w.y = ab_cvt_pk_bf16(dv[g][dt][2], dv[g][dt][3]);
        *reinterpret_cast<uint2*>(dvp + (size_t)ki * p.dkv_stride + col) = w;
      }
    }
  }
}

template <int D>
static int launch_attn_bwd(const AttnBwdParams& p, hipStream_t st) {
  using G = AbGeom<D>;
  const int smem_dq = G::NBUF * 2 * G::IMG;
  const int smem_dkv = G::NBUF * (2 * G::IMG + 512);
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dq_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_dq);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_dkv_kernel<D>), hipFuncAttributeMaxDynamicSharedMemorySize, smem_dkv);
    if (e != hipSuccess) { set_error("ldm_attention_backward: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
    attr_done = true;
  }
  constexpr int CPH = D / 8, SPAN = (256 / CPH) * CPH;
  if ((reinterpret_cast<uintptr_t>(p.o) & 15) == 0 && p.o_stride % 8 == 0) {   // d_o alignment is checked by the caller
    const long long total = (long long)p.batch * p.lq * p.heads * CPH;
    hipLaunchKernelGGL((attn_bwd_prep_vec_kernel<D>), dim3((unsigned)((total + SPAN - 1) / SPAN)), dim3(256), 0, st, p);
  } else {
    const long long rows = (long long)p.batch * p.heads * p.lq;
    hipLaunchKernelGGL(attn_bwd_prep_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, D);
  }
  LDM_HIP_CHECK_LAUNCH("ldm_attention_backward(prep)");
  hipLaunchKernelGGL((attn_bwd_dq_kernel<D>), dim3((p.lq + 64 * G::NGQ - 1) / (64 * G::NGQ), p.heads, p.batch), dim3(256), smem_dq, st, p);
  LDM_HIP_CHECK_LAUNCH("ldm_attention_backward(dq)");
  hipLaunchKernelGGL((attn_bwd_dkv_kernel<D>), dim3((p.lk + 64 * G::NGK - 1) / (64 * G::NGK), p.heads, p.batch), dim3(256), smem_dkv, st, p);
  LDM_HIP_CHECK_LAUNCH("ldm_attention_backward(dkv)");
  return LDM_OK;
}

// ----------------------------------------------------------------------------------------------------------------
// Generic backward for any head dim <= 256 and either dtype (the exact-float32 parity path and the odd head dims of
// small configurations; the forward's attention_generic_kernel is its counterpart).  Same two-kernel, atomic-free
// structure, plain FMA arithmetic: a lane owns one row of the tile being swept for the score / dP dot products, then
// the wave switches to one output column per lane for the accumulation.  Not a speed path.
constexpr int GB_RPW = 4;       // rows (queries resp. keys) owned by a wave
constexpr int GB_MAXD = 256;

template <typename T>
__global__ void __launch_bounds__(256)
attn_bwd_prep_generic_kernel(const AttnBwdParams p, int d) {
  const long long row = (long long)blockIdx.x * 4 + (threadIdx.x >> 6);   // over batch*heads*lq
  const int lane = threadIdx.x & 63;
  if (row >= (long long)p.batch * p.heads * p.lq) return;
  const int qi = (int)(row % p.lq);
  const long long bh = row / p.lq;
  const int h = (int)(bh % p.heads);
  const long long b = bh / p.heads;
  const T* o = static_cast<const T*>(p.o) + (b * p.lq + qi) * p.o_stride + h * d;
  const T* g = static_cast<const T*>(p.d_o) + (b * p.lq + qi) * p.do_stride + h * d;
  float s = 0.f;
  for (int j = lane; j < d; j += 64) s += Elem<T>::ld(o + j) * Elem<T>::ld(g + j);
  s = wave_sum(s);
  if (lane == 0) p.delta[row] = s;
}

// SWEEP_KEYS: the workgroup owns 16 queries and sweeps the keys (dQ); else it owns 16 keys and sweeps the queries (dK, dV).
template <typename T, bool SWEEP_KEYS>
__global__ void __launch_bounds__(256)
attn_bwd_generic_kernel(const AttnBwdParams p, int d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int dp = d + 1;
  float* ta = reinterpret_cast<float*>(smem);             // [64][dp]  K tile      | Q tile
  float* tb = ta + 64 * dp;                               // [64][dp]  V tile      | dO tile
  float* ra = tb + 64 * dp;                               // [4][RPW][d]  own Q    | own K
  float* rb = ra + 4 * GB_RPW * d;                        // [4][RPW][d]  own dO   | own V
  float* ps = rb + 4 * GB_RPW * d;                        // [4][2][64]  per-wave P and dS of the current tile
  float* tl = ps + 4 * 2 * 64;                            // [2][64]     lse / delta of the query tile (dK/dV pass)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int own = SWEEP_KEYS ? p.lq : p.lk, swept = SWEEP_KEYS ? p.lk : p.lq;
  const int r0 = blockIdx.x * (4 * GB_RPW) + wave * GB_RPW;
  const T* qp = static_cast<const T*>(p.q) + (size_t)b * p.lq * p.q_stride + h * d;
  const T* kp = static_cast<const T*>(p.k) + (size_t)b * p.lk * p.kv_stride + h * d;
  const T* vp = static_cast<const T*>(p.v) + (size_t)b * p.lk * p.kv_stride + h * d;
  const T* gp = static_cast<const T*>(p.d_o) + (size_t)b * p.lq * p.do_stride + h * d;
  const float* lse = p.lse + ((size_t)b * p.heads + h) * p.lq;
  const float* dl = p.delta + ((size_t)b * p.heads + h) * p.lq;

  float* mya = ra + wave * GB_RPW * d;
  float* myb = rb + wave * GB_RPW * d;
  float* myp = ps + wave * 128;
  for (int i = lane; i < GB_RPW * d; i += 64) {
    const int rr = i / d, c = i - rr * d;
    const bool ok = r0 + rr < own;
    if (SWEEP_KEYS) {
      mya[i] = ok ? Elem<T>::ld(qp + (size_t)(r0 + rr) * p.q_stride + c) : 0.f;
      myb[i] = ok ? Elem<T>::ld(gp + (size_t)(r0 + rr) * p.do_stride + c) : 0.f;
    } else {
      mya[i] = ok ? Elem<T>::ld(kp + (size_t)(r0 + rr) * p.kv_stride + c) : 0.f;
      myb[i] = ok ? Elem<T>::ld(vp + (size_t)(r0 + rr) * p.kv_stride + c) : 0.f;
    }
  }
  constexpr int DPL = GB_MAXD / 64;
  float acc0[GB_RPW][DPL], acc1[GB_RPW][DPL];             // dQ | (dK, dV)
#pragma unroll
  for (int rr = 0; rr < GB_RPW; ++rr)
#pragma unroll
    for (int k = 0; k < DPL; ++k) { acc0[rr][k] = 0.f; acc1[rr][k] = 0.f; }

  for (int t0 = 0; t0 < swept; t0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * d; i += 256) {
      const int row = i / d, c = i - row * d;
      const bool ok = t0 + row < swept;
      if (SWEEP_KEYS) {
        ta[row * dp + c] = ok ? Elem<T>::ld(kp + (size_t)(t0 + row) * p.kv_stride + c) : 0.f;
        tb[row * dp + c] = ok ? Elem<T>::ld(vp + (size_t)(t0 + row) * p.kv_stride + c) : 0.f;
      } else {
        ta[row * dp + c] = ok ? Elem<T>::ld(qp + (size_t)(t0 + row) * p.q_stride + c) : 0.f;
        tb[row * dp + c] = ok ? Elem<T>::ld(gp + (size_t)(t0 + row) * p.do_stride + c) : 0.f;
      }
    }
    if (!SWEEP_KEYS && tid < 64) {
      const bool ok = t0 + tid < swept;
      tl[tid] = ok ? lse[t0 + tid] : 0.f;
      tl[64 + tid] = ok ? dl[t0 + tid] : 0.f;
    }
    __syncthreads();
    const bool lane_ok = t0 + lane < swept;
#pragma unroll
    for (int rr = 0; rr < GB_RPW; ++rr) {
      const bool row_ok = r0 + rr < own;                  // wave-uniform
      float s = 0.f, dpv = 0.f;
      for (int c = 0; c < d; ++c) {
        s += mya[rr * d + c] * ta[lane * dp + c];         // q.k either way round
        dpv += myb[rr * d + c] * tb[lane * dp + c];       // dO.v
      }
      float L, Dl;
      if (SWEEP_KEYS) { L = row_ok ? lse[r0 + rr] : 0.f; Dl = row_ok ? dl[r0 + rr] : 0.f; }
      else { L = tl[lane]; Dl = tl[64 + lane]; }
      const float pr = (lane_ok && row_ok) ? exp2f(s * p.scale_log2e - L) : 0.f;
      const float ds = pr * (dpv - Dl);
      myp[lane] = pr;
      myp[64 + lane] = ds;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        const int c = lane + 64 * k;
        if (c < d) {
          float a0 = acc0[rr][k], a1 = acc1[rr][k];
          if (SWEEP_KEYS) {
            for (int j = 0; j < 64; ++j) a0 += myp[64 + j] * ta[j * dp + c];            // dQ += dS K
          } else {
            for (int j = 0; j < 64; ++j) {
              a0 += myp[64 + j] * ta[j * dp + c];                                      // dK += dS^T Q
              a1 += myp[j] * tb[j * dp + c];                                           // dV += P^T dO
            }
          }
          acc0[rr][k] = a0; acc1[rr][k] = a1;
        }
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int rr = 0; rr < GB_RPW; ++rr) {
    if (r0 + rr >= own) continue;
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int c = lane + 64 * k;
      if (c >= d) continue;
      if (SWEEP_KEYS) {
        Elem<T>::st(static_cast<T*>(p.dq) + ((size_t)b * p.lq + r0 + rr) * p.dq_stride + h * d + c, acc0[rr][k] * p.scale);
      } else {
        Elem<T>::st(static_cast<T*>(p.dk) + ((size_t)b * p.lk + r0 + rr) * p.dkv_stride + h * d + c, acc0[rr][k] * p.scale);
        Elem<T>::st(static_cast<T*>(p.dv) + ((size_t)b * p.lk + r0 + rr) * p.dkv_stride + h * d + c, acc1[rr][k]);
      }
    }
  }
}

template <typename T>
static int launch_attn_bwd_generic(const AttnBwdParams& p, int d, hipStream_t st) {
  const size_t smem = ((size_t)2 * 64 * (d + 1) + 2 * 4 * GB_RPW * d + 4 * 2 * 64 + 2 * 64) * sizeof(float);
  if (smem > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_generic_kernel<T, true>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(attn_bwd_generic_kernel<T, false>),
                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("ldm_attention_backward: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
  }
  const long long rows = (long long)p.batch * p.heads * p.lq;
  hipLaunchKernelGGL((attn_bwd_prep_generic_kernel<T>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, st, p, d);
  LDM_HIP_CHECK_LAUNCH("ldm_attention_backward(generic prep)");
  hipLaunchKernelGGL((attn_bwd_generic_kernel<T, true>), dim3((p.lq + 4 * GB_RPW - 1) / (4 * GB_RPW), p.heads, p.batch), dim3(256),
                     smem, st, p, d);
  LDM_HIP_CHECK_LAUNCH("ldm_attention_backward(generic dq)");
  hipLaunchKernelGGL((attn_bwd_generic_kernel<T, false>), dim3((p.lk + 4 * GB_RPW - 1) / (4 * GB_RPW), p.heads, p.batch), dim3(256),
                     smem, st, p, d);
  LDM_HIP_CHECK_LAUNCH("ldm_attention_backward(generic dkv)");
  return LDM_OK;
}

}  // namespace ldm

using namespace ldm;

extern "C" int ldm_attention_backward(const void* q, int q_stride, const void* k, const void* v, int kv_stride, const void* o,
                                      int o_stride, const void* d_o, int do_stride, const float* lse, void* dq, int dq_stride,
                                      void* dk, void* dv, int dkv_stride, float* delta_workspace, int dtype, int batch,
                                      int heads, int lq, int lk, int d, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_attention_backward: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && heads >= 1 && lq >= 0 && lk >= 1 && d >= 1, "ldm_attention_backward: bad sizes");
  if (batch == 0 || lq == 0) return LDM_OK;
  LDM_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv && delta_workspace, "ldm_attention_backward: null pointer");
  const int hd = heads * d;
  LDM_REQUIRE(q_stride >= hd && kv_stride >= hd && o_stride >= hd && do_stride >= hd && dq_stride >= hd && dkv_stride >= hd,
              "ldm_attention_backward: row strides smaller than heads*d");
  LDM_REQUIRE(batch <= 65535 && heads <= 65535, "ldm_attention_backward: batch/heads exceed the grid limits");
  const bool vec_ok = aligned16(q) && aligned16(k) && aligned16(v) && aligned16(d_o) && q_stride % 8 == 0 && kv_stride % 8 == 0 &&
                      do_stride % 8 == 0 && (reinterpret_cast<uintptr_t>(dq) & 7) == 0 && (reinterpret_cast<uintptr_t>(dk) & 7) == 0 &&
                      (reinterpret_cast<uintptr_t>(dv) & 7) == 0 && dq_stride % 4 == 0 && dkv_stride % 4 == 0 &&
                      (long long)lk * kv_stride * 2 < (1LL << 31) && (long long)lq * q_stride * 2 < (1LL << 31) &&
                      (long long)lq * do_stride * 2 < (1LL << 31);
  AttnBwdParams p;
  p.q = q; p.k = k; p.v = v; p.o = o; p.d_o = d_o; p.dq = dq; p.dk = dk; p.dv = dv;
  p.q_stride = q_stride; p.kv_stride = kv_stride; p.o_stride = o_stride; p.do_stride = do_stride;
  p.dq_stride = dq_stride; p.dkv_stride = dkv_stride;
  p.batch = batch; p.heads = heads; p.lq = lq; p.lk = lk;
  p.scale = 1.0f / sqrtf((float)d);
  p.scale_log2e = p.scale * 1.4426950408889634f;
  p.lse = lse; p.delta = delta_workspace;
  hipStream_t st = (hipStream_t)stream;
  if (dtype == LDM_BF16 && vec_ok) {
    switch (d) {
      case 40: return launch_attn_bwd<40>(p, st);
      case 64: return launch_attn_bwd<64>(p, st);
      case 80: return launch_attn_bwd<80>(p, st);
      case 128: return launch_attn_bwd<128>(p, st);
      case 160: return launch_attn_bwd<160>(p, st);
      default: break;
    }
  }
  LDM_REQUIRE(d <= GB_MAXD, "ldm_attention_backward: head dim %d exceeds %d", d, GB_MAXD);
  return dtype == LDM_BF16 ? launch_attn_bwd_generic<bf16_t>(p, d, st) : launch_attn_bwd_generic<float>(p, d, st);
}
