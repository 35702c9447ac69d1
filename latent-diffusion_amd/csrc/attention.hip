// Flash-style attention (no mask, scale 1/sqrt(d)) for the UNet's self- and cross-attention layers.
//
// bf16 path (head dims that are multiples of 8 up to 160: 40 / 80 / 160 in the SD-1.5-shaped model):
//   one workgroup = 4 waves = 128 queries of one (sample, head); each wave owns 32 queries (two groups of 16) and
//   sweeps the keys in tiles of 64.  Both products run on v_mfma_f32_16x16x32_bf16 in the TRANSPOSED
//   orientation, S^T = K Q^T and O^T = V^T P^T: a lane then owns one query column, so the softmax is in-lane
//   except for one 4-way max, and the S^T accumulators are, after packing to bf16, directly the B operand of the
//   PV product (no LDS round trip for P).  V^T fragments come from ds_read_b64_tr_b16 (hardware transpose).
//   K/V tiles are register-staged into two LDS buffers: the loads of tile t+2 are issued, and tile t+1 is
//   committed, before the math of tile t; one barrier per tile.
// generic path (float32, or head dims the MFMA path does not take): one wave per query, fp32 math.
#include "common.h"

#include <type_traits>

namespace ldm {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

__device__ __forceinline__ uint32_t cvt_pk_bf16(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  f2 v = {a, b};
  b2 r = __builtin_convertvector(v, b2);
  return __builtin_bit_cast(uint32_t, r);
}

struct AttnParams {
  const void* q; const void* k; const void* v; void* out;
  int q_stride, kv_stride, out_stride;
  int batch, heads, lq, lk;
  float scale_log2e;  // (1/sqrt(d)) * log2(e)
  float scale;
  float* lse;         // optional [batch][heads][lq]: log2-sum-exp2 of the scaled scores (training: saved for the backward)
#ifdef LDM_GEMM_STAMPS
  unsigned long long* stamps;  // diagnostic build only (make stamps): 16 cycle stamps per workgroup
#endif
};

#ifdef LDM_GEMM_STAMPS
#define ATT_STAMP(i)                                                                                          \
  do {                                                                                                        \
    if (p.stamps != nullptr && threadIdx.x == 0)                                                              \
      p.stamps[((size_t)(blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x) * 16 + (i)] =         \
          __builtin_amdgcn_s_memtime();                                                                       \
  } while (0)
#else
#define ATT_STAMP(i) do {} while (0)
#endif

template <int D>
struct AttnGeom {
  static constexpr int KS = (D + 31) / 32;             // 32-deep contraction steps of Q K^T
  static constexpr int DK = KS * 32;
  static constexpr int DT = D / 16 + 1 - (D % 16 == 0 ? 1 : 0);  // 16-row tiles of O^T
  static constexpr int KROW = DK * 2 + 32;             // bytes per key row of the K image: 32 (mod 64) is conflict-free for the four 16-lane groups of ds_read_b128 (+16 was 2-way)
  static constexpr int VROW = (DT % 2 == 1 ? DT : DT + 1) * 32;  // odd multiple of 32 B: tr reads conflict-free
  static constexpr int K_BYTES = 64 * KROW;
  static constexpr int V_BYTES = 64 * VROW;
  static constexpr int BUF_BYTES = K_BYTES + V_BYTES;  // one key tile (K image then V image); two of them in LDS
  static constexpr int CPR = D / 8;                    // 16-byte chunks per row in global memory
  static constexpr int NCH = (64 * CPR + 255) / 256;   // chunks per thread per tile and operand
  // D not a multiple of 16: the V image has pad columns.  Column D is set to 1.0 once, so row D of O^T accumulates
  // sum_k P[k] on the matrix core and the softmax denominator costs no vector adds.
  static constexpr bool ONES = (D % 16) != 0;
  // D not a multiple of 32: the contraction of Q K^T has pad columns.  Column D of the K image is set to 1.0 once and
  // column D of the Q fragment holds -m (the running maximum), Q itself is pre-multiplied by log2(e)/sqrt(d): the
  // matrix core then delivers s*scale - m directly and the softmax needs no multiply-add per score.
  static constexpr bool AUG = (D % 32) != 0;
  static constexpr int AUG_KS = D / 32, AUG_FQ = (D % 32) / 8, AUG_E = D % 8;
};

__device__ __forceinline__ float round_to_bf16(float x) { return __uint_as_float(cvt_pk_bf16(x, 0.0f) << 16); }

// 16 bytes per lane from a raw buffer (base, bytes): `buffer_load_dwordx4 v, v, s[rsrc], s offen`.  voff is the
// per-lane byte offset, soff a wave-uniform one; an offset past `bytes` reads zeros (key rows past lk).
// The wave-uniform tile offset goes into the DESCRIPTOR (base advanced, record count shrunk), not into the instruction's
// scalar-offset field: LLVM documents soffset as EXCLUDED from the range check of raw buffer loads, and the rows past lk
// of the last, ragged tile must come back as zeros (never as whatever lies behind this (sample, head)'s rows).
__device__ __forceinline__ uint4 buffer_load16(const void* base, unsigned bytes, unsigned voff, unsigned soff) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const unsigned left = bytes - soff;   // callers fetch existing tiles only: soff < bytes (a clamped form compiles to a VALU saturating subtract and drags the descriptor into a waterfall loop)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(static_cast<const unsigned char*>(base) + soff), 0, (int)left, 0x00020000);
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
  return make_uint4(r[0], r[1], r[2], r[3]);
}

// D <= 80: two waves per SIMD (<= 256 registers), which also makes hipcc keep the MFMA accumulators in VGPRs
// (with 512 registers allowed it parks them in AGPRs and pays ~200 v_accvgpr moves per key tile for the rescale).
//
// The loop is vector-ALU bound at d = 40 (per 64-key tile a wave runs 28 MFMAs but 32 exponentials per lane), so
// everything else is kept off the vector ALU: K/V staging addresses are loop-invariant per-lane offsets plus a scalar
// tile offset (buffer loads; rows past lk read as zeros), the ragged-tile mask exists only in the last tile's code,
// the running maximum is only raised (and O rescaled) when some query's maximum grew by more than 2^8, and for
// d = 40 the denominator comes out of the PV product (AttnGeom::ONES).  One barrier per key tile: tile t+1 is
// committed to the other LDS buffer while tile t is being consumed.
// Maximum over the four 16-lane rows of a wave (the key quarters fq of one query column) on the vector ALU:
// v_permlane16_swap / v_permlane32_swap exchange rows between two registers, so the butterfly needs no trip through the
// LDS crossbar (ds_bpermute, ~100+ cycles of latency twice per query group and key tile, sat on the tile's critical path).
__device__ __forceinline__ float max_over_fq(float v) {
  const auto a = __builtin_amdgcn_permlane16_swap(__float_as_uint(v), __float_as_uint(v), false, false);
  const float m = fmaxf(__uint_as_float(a[0]), __uint_as_float(a[1]));
  const auto b = __builtin_amdgcn_permlane32_swap(__float_as_uint(m), __float_as_uint(m), false, false);
  return fmaxf(__uint_as_float(b[0]), __uint_as_float(b[1]));
}

template <int D>
__global__ void __launch_bounds__(256, (D <= 80 ? 2 : 1))
attention_bf16_kernel(const AttnParams p) {
  using G = AttnGeom<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  ATT_STAMP(0);

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int fr = lane & 15, fq = lane >> 4;
  // Workgroup -> (sample, head, query block): consecutive workgroup ids alternate over the 8 XCDs, so give every XCD a
  // contiguous run of the (sample, head, query block) order: the query blocks of one (sample, head) then share its
  // K/V through ONE L2 instead of fetching them into all eight (FETCH_SIZE was 4.5x the K/V bytes).
  const int nqb = gridDim.x;
  const int nblk = nqb * gridDim.y * gridDim.z;
  const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * nqb + blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
  const int work = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int bh = work / nqb;
  const int b = bh / p.heads, h = bh - b * p.heads;
  const int q0 = (work - bh * nqb) * 128 + wave * 32;

  const bf16_t* qp = static_cast<const bf16_t*>(p.q) + (size_t)b * p.lq * p.q_stride + h * D;
  const bf16_t* kp = static_cast<const bf16_t*>(p.k) + (size_t)b * p.lk * p.kv_stride + h * D;
  const bf16_t* vp = static_cast<const bf16_t*>(p.v) + (size_t)b * p.lk * p.kv_stride + h * D;
  bf16_t* op = static_cast<bf16_t*>(p.out) + (size_t)b * p.lq * p.out_stride + h * D;
  const unsigned kv_bytes = ((unsigned)(p.lk - 1) * p.kv_stride + D) * 2u;   // this (sample, head)'s rows, < 2 GiB (host)

  // zero both LDS images once: pad columns are never written again and must not hold NaN patterns
  for (int i = tid * 16; i < 2 * G::BUF_BYTES; i += 256 * 16)
    *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (G::ONES && tid < 128)
    *reinterpret_cast<bf16_t*>(smem + (tid >> 6) * G::BUF_BYTES + G::K_BYTES + (tid & 63) * G::VROW + D * 2) = 0x3F80;  // 1.0
  if (G::AUG && tid >= 128)
    *reinterpret_cast<bf16_t*>(smem + ((tid - 128) >> 6) * G::BUF_BYTES + (tid & 63) * G::KROW + D * 2) = 0x3F80;

  // Q^T fragments (B operand): lane (query fr, k-group fq) holds Q[query][32*ks + 8*fq + 0..7]
  bf16x8_t qf[2][G::KS];
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    const int qi = q0 + g * 16 + fr;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int col = ks * 32 + fq * 8;
      bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (qi < p.lq && col < D) z = *reinterpret_cast<const bf16x8_t*>(qp + (size_t)qi * p.q_stride + col);
      if (G::AUG) {  // Q * log2(e)/sqrt(d), rounded to bf16 again
#pragma unroll
        for (int e = 0; e < 8; e += 2) {
          const uint32_t w = cvt_pk_bf16(bf16_to_f32((bf16_t)z[e]) * p.scale_log2e, bf16_to_f32((bf16_t)z[e + 1]) * p.scale_log2e);
          z[e] = (short)(w & 0xffffu);
          z[e + 1] = (short)(w >> 16);
        }
      }
      qf[g][ks] = z;
    }
  }

  f32x4_t o[2][G::DT];
#pragma unroll
  for (int g = 0; g < 2; ++g)
#pragma unroll
    for (int t = 0; t < G::DT; ++t) o[g][t] = f32x4_t{0.f, 0.f, 0.f, 0.f};
  float m_run[2] = {G::AUG ? 0.f : -INFINITY, G::AUG ? 0.f : -INFINITY};   // AUG: column D of Q holds -m_run = 0
  float l_run[2] = {0.f, 0.f};

  // staging geometry of this thread: loop-invariant
  unsigned g_off[G::NCH];       // byte offset of the chunk inside a key tile (global), or out of range
  int k_lds[G::NCH], v_lds[G::NCH];
#pragma unroll
  for (int c = 0; c < G::NCH; ++c) {
    const int idx = c * 256 + tid;
    const int row = idx / G::CPR, ch = idx - row * G::CPR;
    const bool ok = idx < 64 * G::CPR;
    g_off[c] = ok ? ((unsigned)row * p.kv_stride + ch * 8) * 2u : 0x80000000u;
    k_lds[c] = ok ? row * G::KROW + ch * 16 : -1;
    v_lds[c] = G::K_BYTES + row * G::VROW + ch * 16;
  }
  uint4 kreg[G::NCH], vreg[G::NCH];
  auto fetch = [&](int t) {
    const unsigned soff = (unsigned)t * 64u * p.kv_stride * 2u;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
      kreg[c] = buffer_load16(kp, kv_bytes, g_off[c], soff);
      vreg[c] = buffer_load16(vp, kv_bytes, g_off[c], soff);
    }
  };
  auto commit = [&](unsigned char* buf) {
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) {
      if (k_lds[c] >= 0) {
        *reinterpret_cast<uint4*>(buf + k_lds[c]) = kreg[c];
        *reinterpret_cast<uint4*>(buf + v_lds[c]) = vreg[c];
      }
    }
  };

  const int tq = fr >> 2, tp = fr & 3;
  auto tile = [&](auto ragged_tag, const unsigned char* buf, int t0, bool first) {
    constexpr bool RAGGED = decltype(ragged_tag)::value;
    const unsigned char* klds = buf;
    const unsigned char* vlds = buf + G::K_BYTES;
    // ---- S^T = K Q^T : s[g][kb] holds keys kb*16 + 4*fq + r of query fr
    f32x4_t s[2][4];
#pragma unroll
    for (int g = 0; g < 2; ++g)
#pragma unroll
      for (int kb = 0; kb < 4; ++kb) s[g][kb] = f32x4_t{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int kb = 0; kb < 4; ++kb) {
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(klds + (kb * 16 + fr) * G::KROW + (ks * 4 + fq) * 16);
        s[0][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[0][ks], s[0][kb], 0, 0, 0);
        s[1][kb] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(kf, qf[1][ks], s[1][kb], 0, 0, 0);
      }
    }
    if constexpr (RAGGED) {  // keys past lk get -inf
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r)
          if (t0 + kb * 16 + fq * 4 + r >= p.lk) { s[0][kb][r] = -INFINITY; s[1][kb][r] = -INFINITY; }
    }

    // ---- online softmax, per query column; probabilities packed to bf16 as the B operand of P V
    bf16x8_t pb[2][2];
#pragma unroll
    for (int g = 0; g < 2; ++g) {
      float mx = s[g][0][0];
#pragma unroll
      for (int kb = 0; kb < 4; ++kb)
#pragma unroll
        for (int r = 0; r < 4; ++r) mx = fmaxf(mx, s[g][kb][r]);
      mx = max_over_fq(mx);
      float pv[4][4];
      if constexpr (G::AUG) {
        // s already is score*scale - m_run.  Lazy maximum: m_run is raised (and O rescaled) only when some query of
        // this group sees a value above 8 (p <= 2^8: harmless in fp32 and in bf16), or in the first tile, where it is
        // set to the tile maximum whatever its sign.  m_run stays bf16-representable so that the copy inside the Q
        // fragment is exact.  Wave-uniform branch.
        const bool raise = first || mx > 8.0f;
        if (__builtin_amdgcn_ballot_w64(raise) != 0) {
          const float m_new = round_to_bf16(m_run[g] + (raise ? mx : 0.f));
          const float d2 = m_new - m_run[g];
          const float alpha = __builtin_amdgcn_exp2f(-d2);
          m_run[g] = m_new;
          if (!G::ONES) l_run[g] *= alpha;
#pragma unroll
          for (int t2 = 0; t2 < G::DT; ++t2) {
            o[g][t2][0] *= alpha; o[g][t2][1] *= alpha; o[g][t2][2] *= alpha; o[g][t2][3] *= alpha;
          }
#pragma unroll
          for (int kb = 0; kb < 4; ++kb)
#pragma unroll
            for (int r = 0; r < 4; ++r) s[g][kb][r] -= d2;
          if (fq == G::AUG_FQ) qf[g][G::AUG_KS][G::AUG_E] = (short)(__float_as_uint(-m_new) >> 16);
        }
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) pv[kb][r] = __builtin_amdgcn_exp2f(s[g][kb][r]);   // raw v_exp_f32: arguments <= 8
      } else {
        const float ms = mx * p.scale_log2e;
        // Lazy maximum: probabilities are taken relative to m_run, which is only raised when some query of this
        // group saw its maximum grow by more than 8 (p <= 2^8: harmless in fp32 and in bf16).  Wave-uniform branch.
        if (__builtin_amdgcn_ballot_w64(ms > m_run[g] + 8.0f) != 0) {
          const float m_new = fmaxf(m_run[g], ms);
          const float alpha = __builtin_amdgcn_exp2f(m_run[g] - m_new);
          m_run[g] = m_new;
          if (!G::ONES) l_run[g] *= alpha;
#pragma unroll
          for (int t2 = 0; t2 < G::DT; ++t2) {
            o[g][t2][0] *= alpha; o[g][t2][1] *= alpha; o[g][t2][2] *= alpha; o[g][t2][3] *= alpha;
          }
        }
        const float mr = m_run[g];
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r)
            pv[kb][r] = __builtin_amdgcn_exp2f(fmaf(s[g][kb][r], p.scale_log2e, -mr));  // raw v_exp_f32: arguments <= 8
      }
      if (!G::ONES) {
        float psum = 0.f;
#pragma unroll
        for (int kb = 0; kb < 4; ++kb)
#pragma unroll
          for (int r = 0; r < 4; ++r) psum += pv[kb][r];
        l_run[g] += psum;
      }
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        // MFMA k index 8*fq + j  <->  key (2*st)*16 + 4*fq + j (j < 4), (2*st+1)*16 + 4*fq + (j-4) (j >= 4)
        typedef __attribute__((ext_vector_type(4))) uint32_t u4;
        u4 w;
        w[0] = cvt_pk_bf16(pv[2 * st][0], pv[2 * st][1]);
        w[1] = cvt_pk_bf16(pv[2 * st][2], pv[2 * st][3]);
        w[2] = cvt_pk_bf16(pv[2 * st + 1][0], pv[2 * st + 1][1]);
        w[3] = cvt_pk_bf16(pv[2 * st + 1][2], pv[2 * st + 1][3]);
        pb[g][st] = __builtin_bit_cast(bf16x8_t, w);
      }
    }

    // ---- O^T += V^T P^T : V^T fragments by transposed LDS reads (4 keys x 16 dims per 16-lane group)
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) {
#pragma unroll
      for (int st = 0; st < 2; ++st) {
        const unsigned char* a0 = vlds + ((2 * st) * 16 + 4 * fq + tq) * G::VROW + (dt * 16 + 4 * tp) * 2;
        const unsigned char* a1 = a0 + 16 * G::VROW;
        const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)a0);
        const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)a1);
        const bf16x8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
        o[0][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[0][st], o[0][dt], 0, 0, 0);
        o[1][dt] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(vf, pb[1][st], o[1][dt], 0, 0, 0);
      }
    }
  };

  const int ntiles = (p.lk + 63) / 64;
  fetch(0);
  commit(smem);
  if (ntiles > 1) fetch(1);
  ATT_STAMP(1);
  for (int t = 0; t < ntiles; ++t) {
    unsigned char* cur = smem + (t & 1) * G::BUF_BYTES;
    unsigned char* nxt = smem + ((t + 1) & 1) * G::BUF_BYTES;
    if (t == 8) ATT_STAMP(2);
    __syncthreads();   // tile t is in `cur` for every wave, and every wave is done reading `nxt` (tile t-1)
    if (t == 8) ATT_STAMP(3);
    if (t + 1 < ntiles) {
      __builtin_amdgcn_s_setprio(3);      // the few staging instructions go ahead of the other waves' math
      commit(nxt);                        // tile t+1: its loads were issued a whole tile ago
      if (t + 2 < ntiles) fetch(t + 2);   // in flight under the math below
      __builtin_amdgcn_s_setprio(0);
    }
    if (t == 8) ATT_STAMP(4);
    if (t + 1 == ntiles && (p.lk & 63) != 0) tile(std::true_type{}, cur, t * 64, t == 0);
    else tile(std::false_type{}, cur, t * 64, t == 0);
    if (t == 8) ATT_STAMP(7);
  }
  ATT_STAMP(8);

  // ---- normalise and store: lane holds dims dt*16 + 4*fq + 0..3 of query fr
#pragma unroll
  for (int g = 0; g < 2; ++g) {
    float l;
    if (G::ONES) {   // row D of O^T: tile D/16, lanes fq == (D%16)/4, register (D%16)%4
      l = __shfl(o[g][D / 16][(D % 16) % 4], ((D % 16) / 4) * 16 + fr, 64);
    } else {
      l = l_run[g];
      l += __shfl_xor(l, 16, 64);
      l += __shfl_xor(l, 32, 64);
    }
    const float inv = 1.0f / l;
    const int qi = q0 + g * 16 + fr;
    if (qi >= p.lq) continue;
    if (p.lse != nullptr && fq == 0) p.lse[((size_t)b * p.heads + h) * p.lq + qi] = m_run[g] + __log2f(l);
#pragma unroll
    for (int dt = 0; dt < G::DT; ++dt) {
      const int col = dt * 16 + fq * 4;
      if (col < D) {
        uint2 w;
        w.x = cvt_pk_bf16(o[g][dt][0] * inv, o[g][dt][1] * inv);
        w.y = cvt_pk_bf16(o[g][dt][2] * inv, o[g][dt][3] * inv);
        *reinterpret_cast<uint2*>(op + (size_t)qi * p.out_stride + col) = w;
      }
    }
  }
}

// ---------------------------------------------------------------------------------- generic path
// One wave per query (4 queries per wave in sequence, 16 per workgroup); K/V tiles of 64 keys in LDS as float.
constexpr int GEN_QPW = 4;
constexpr int GEN_MAXD = 256;

template <typename T>
__global__ void __launch_bounds__(256)
attention_generic_kernel(const AttnParams p, int d) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int dp = d + 1;                                   // padded row: conflict-free column walks
  float* kt = reinterpret_cast<float*>(smem);             // [64][dp]
  float* vt = kt + 64 * dp;                               // [64][dp]
  float* qs = vt + 64 * dp;                               // [4 waves][GEN_QPW][d]
  float* ps = qs + 4 * GEN_QPW * d;                       // [4 waves][64]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int b = blockIdx.z, h = blockIdx.y;
  const int q0 = blockIdx.x * (4 * GEN_QPW) + wave * GEN_QPW;
  const T* qp = static_cast<const T*>(p.q) + (size_t)b * p.lq * p.q_stride + h * d;
  const T* kp = static_cast<const T*>(p.k) + (size_t)b * p.lk * p.kv_stride + h * d;
  const T* vp = static_cast<const T*>(p.v) + (size_t)b * p.lk * p.kv_stride + h * d;
  T* op = static_cast<T*>(p.out) + (size_t)b * p.lq * p.out_stride + h * d;

  float* myq = qs + wave * GEN_QPW * d;
  float* myp = ps + wave * 64;
  for (int i = lane; i < GEN_QPW * d; i += 64) {
    const int qq = i / d, c = i - qq * d;
    myq[i] = (q0 + qq < p.lq) ? Elem<T>::ld(qp + (size_t)(q0 + qq) * p.q_stride + c) : 0.f;
  }
  constexpr int DPL = GEN_MAXD / 64;                      // output dims per lane
  float m_run[GEN_QPW], l_run[GEN_QPW], o[GEN_QPW][DPL];
#pragma unroll
  for (int qq = 0; qq < GEN_QPW; ++qq) {
    m_run[qq] = -INFINITY; l_run[qq] = 0.f;
#pragma unroll
    for (int k = 0; k < DPL; ++k) o[qq][k] = 0.f;
  }
  for (int t0 = 0; t0 < p.lk; t0 += 64) {
    __syncthreads();
    for (int i = tid; i < 64 * d; i += 256) {
      const int row = i / d, c = i - row * d;
      const bool ok = t0 + row < p.lk;
      kt[row * dp + c] = ok ? Elem<T>::ld(kp + (size_t)(t0 + row) * p.kv_stride + c) : 0.f;
      vt[row * dp + c] = ok ? Elem<T>::ld(vp + (size_t)(t0 + row) * p.kv_stride + c) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int qq = 0; qq < GEN_QPW; ++qq) {
      float sc = 0.f;
      for (int c = 0; c < d; ++c) sc += myq[qq * d + c] * kt[lane * dp + c];
      sc = (t0 + lane < p.lk) ? sc * p.scale : -INFINITY;
      const float m_new = fmaxf(m_run[qq], wave_max(sc));
      const float alpha = expf(m_run[qq] - m_new);
      const float e = expf(sc - m_new);
      l_run[qq] = l_run[qq] * alpha + wave_sum(e);
      m_run[qq] = m_new;
      myp[lane] = e;
      __builtin_amdgcn_wave_barrier();
#pragma unroll
      for (int k = 0; k < DPL; ++k) {
        const int c = lane + 64 * k;
        float acc = o[qq][k] * alpha;
        if (c < d)
          for (int j = 0; j < 64; ++j) acc += myp[j] * vt[j * dp + c];
        o[qq][k] = acc;
      }
      __builtin_amdgcn_wave_barrier();
    }
  }
#pragma unroll
  for (int qq = 0; qq < GEN_QPW; ++qq) {
    if (q0 + qq >= p.lq) continue;
    if (p.lse != nullptr && lane == 0)
      p.lse[((size_t)b * p.heads + h) * p.lq + q0 + qq] = (m_run[qq] + logf(l_run[qq])) * 1.4426950408889634f;
    const float inv = 1.0f / l_run[qq];
#pragma unroll
    for (int k = 0; k < DPL; ++k) {
      const int c = lane + 64 * k;
      if (c < d) Elem<T>::st(op + (size_t)(q0 + qq) * p.out_stride + c, o[qq][k] * inv);
    }
  }
}

// attention32.hip: the 32x32x16, software-pipelined kernel for small head dims (d = 40); true when it took the call
bool attention32_try(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out, int out_stride,
                     int batch, int heads, int lq, int lk, int d, float scale_log2e, float* lse, hipStream_t st, int* rc);

template <int D>
static int launch_mfma(const AttnParams& p, hipStream_t st) {
  using G = AttnGeom<D>;
  const int smem = 2 * G::BUF_BYTES;
  static bool attr_done = false;   // > 64 KiB of dynamic LDS needs the opt-in once per process
  if (smem > 64 * 1024 && !attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(attention_bf16_kernel<D>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { set_error("ldm_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
    attr_done = true;
  }
  dim3 grid((p.lq + 127) / 128, p.heads, p.batch);
  hipLaunchKernelGGL((attention_bf16_kernel<D>), grid, dim3(256), smem, st, p);
  LDM_HIP_CHECK_LAUNCH("ldm_attention(bf16 mfma)");
  return LDM_OK;
}

}  // namespace ldm

using namespace ldm;

#ifdef LDM_GEMM_STAMPS
static unsigned long long* g_att_stamp_buf = nullptr;
static long long g_att_stamp_blocks = 0;
extern "C" int ldm_debug_set_attention_stamps(unsigned long long* buf, long long blocks) {
  g_att_stamp_buf = buf;
  g_att_stamp_blocks = blocks;
  return LDM_OK;
}
#endif

static int attention_impl(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out,
                          int out_stride, int dtype, int batch, int heads, int lq, int lk, int d, float* lse, void* stream);

extern "C" int ldm_attention(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out,
                             int out_stride, int dtype, int batch, int heads, int lq, int lk, int d, void* stream) {
  return attention_impl(q, q_stride, k, v, kv_stride, out, out_stride, dtype, batch, heads, lq, lk, d, nullptr, stream);
}

extern "C" int ldm_attention_train(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out,
                                   int out_stride, int dtype, int batch, int heads, int lq, int lk, int d, float* lse,
                                   void* stream) {
  LDM_REQUIRE(lse != nullptr || batch == 0 || lq == 0, "ldm_attention_train: null lse");
  return attention_impl(q, q_stride, k, v, kv_stride, out, out_stride, dtype, batch, heads, lq, lk, d, lse, stream);
}

static int attention_impl(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out,
                          int out_stride, int dtype, int batch, int heads, int lq, int lk, int d, float* lse, void* stream) {
  LDM_REQUIRE(dtype_ok(dtype), "ldm_attention: bad dtype %d", dtype);
  LDM_REQUIRE(batch >= 0 && heads >= 1 && lq >= 0 && lk >= 1 && d >= 1, "ldm_attention: bad sizes");
  if (batch == 0 || lq == 0) return LDM_OK;
  LDM_REQUIRE(q && k && v && out, "ldm_attention: null pointer");
  LDM_REQUIRE(q_stride >= heads * d && kv_stride >= heads * d && out_stride >= heads * d, "ldm_attention: row strides smaller than heads*d");
  LDM_REQUIRE(batch <= 65535 && heads <= 65535, "ldm_attention: batch/heads exceed the grid limits");
  AttnParams p;
  p.q = q; p.k = k; p.v = v; p.out = out;
  p.q_stride = q_stride; p.kv_stride = kv_stride; p.out_stride = out_stride;
  p.batch = batch; p.heads = heads; p.lq = lq; p.lk = lk;
  p.scale = 1.0f / sqrtf((float)d);
  p.scale_log2e = p.scale * 1.4426950408889634f;
  p.lse = lse;
#ifdef LDM_GEMM_STAMPS
  p.stamps = (long long)((lq + 127) / 128) * heads * batch <= g_att_stamp_blocks ? g_att_stamp_buf : nullptr;
#endif
  hipStream_t st = (hipStream_t)stream;
  const bool vec_ok = aligned16(q) && aligned16(k) && aligned16(v) && (reinterpret_cast<uintptr_t>(out) & 7) == 0 &&
                      q_stride % 8 == 0 && kv_stride % 8 == 0 && out_stride % 4 == 0;
  if (dtype == LDM_BF16 && vec_ok && (long long)lk * kv_stride * 2 < (1LL << 31)) {
    static const bool v1 = getenv("LDM_ATTN_V1") != nullptr;      // tools/attention_bench.py: the 16x16x32 kernel for d = 40 (A/B)
    int rc = LDM_OK;
    if (!v1 && attention32_try(q, q_stride, k, v, kv_stride, out, out_stride, batch, heads, lq, lk, d, p.scale_log2e, lse, st, &rc))
      return rc;
    switch (d) {
      case 40: return launch_mfma<40>(p, st);
      case 64: return launch_mfma<64>(p, st);
      case 80: return launch_mfma<80>(p, st);
      case 128: return launch_mfma<128>(p, st);
      case 160: return launch_mfma<160>(p, st);
      default: break;
    }
  }
  LDM_REQUIRE(d <= GEN_MAXD, "ldm_attention: head dim %d exceeds %d", d, GEN_MAXD);
  const size_t smem = ((size_t)2 * 64 * (d + 1) + 4 * GEN_QPW * d + 4 * 64) * sizeof(float);
  dim3 grid((lq + 4 * GEN_QPW - 1) / (4 * GEN_QPW), heads, batch);
  if (smem > 64 * 1024) {
    auto kern = dtype == LDM_BF16 ? (const void*)attention_generic_kernel<bf16_t> : (const void*)attention_generic_kernel<float>;
    hipError_t e = hipFuncSetAttribute(kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)smem);
    if (e != hipSuccess) { set_error("ldm_attention: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
  }
  if (dtype == LDM_BF16)
    hipLaunchKernelGGL((attention_generic_kernel<bf16_t>), grid, dim3(256), smem, st, p, d);
  else
    hipLaunchKernelGGL((attention_generic_kernel<float>), grid, dim3(256), smem, st, p, d);
  LDM_HIP_CHECK_LAUNCH("ldm_attention(generic)");
  return LDM_OK;
}
