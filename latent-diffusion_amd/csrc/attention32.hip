// Flash-style attention for SMALL head dims (d = 40: the five 4,096-token self-attentions of the 64x64 level, 107 of the
// 126 attention GFLOP per image, unet.py:678) on v_mfma_f32_32x32x16_bf16, software-pipelined over key tiles.
//
// Why a second kernel.  At d = 40 the loop is bound by vector-instruction ISSUE, not by the matrix cores: per 32-query x
// 64-key tile a lane must issue 32 v_exp_f32 (8 cycles each) whatever the tiling, and an MFMA holds the SIMD's issue port
// for 8 of its cycles.  The 16x16x32 form (attention.hip) needs 28 MFMAs per wave and tile (224 issue cycles, K padded
// 40 -> 64); the 32x32x16 form needs 14 (112 issue cycles): its K granularity is 16, so Q K^T pads 40 (+1) only to 48, and
// a query's 64 scores sit in TWO lanes (l, l + 32) instead of four, which halves the cross-lane part of the row maximum.
// The second lever is overlap INSIDE a wave: the matrix core and the vector ALU are separate pipes, but a wave that runs
// Q K^T, then its softmax, then P V uses one at a time.  Here tile t+1's Q K^T is issued in the same basic block as tile
// t's exponentials and P V product, so the compiler interleaves them (two score sets live, 2 waves per SIMD).
//
// Layout (both products TRANSPOSED, as in attention.hip: S^T = K Q^T, O^T = V^T P^T, a lane owns one query column):
//   wave = 32 queries, lane l: query l & 31, half h = l >> 5;  workgroup = 4 waves = 128 queries of one (sample, head).
//   S^T accumulator (32 keys x 32 queries): register r holds key (r & 3) + 8 (r >> 2) + 4 h.  Registers 8 s .. 8 s + 7,
//   packed to bf16, ARE the B operand of k-step s of the next 32x32x16 MFMA (no lane movement, no LDS), provided the A
//   operand (V^T) lists its keys in the same order -- two ds_read_b64_tr_b16 per k-step do exactly that.
//   Column D of the K image is 1.0 and element D of every Q fragment holds -m (the running maximum), Q is pre-multiplied
//   by log2(e)/sqrt(d): the matrix core delivers s*scale - m and the softmax is one v_exp_f32 per score.  Column D + 1 of
//   the K image is 0 for real keys and +BIG for rows past lk (last, ragged tile), element D + 1 of Q is -1: masked scores
//   come out of the MFMA as -BIG, no masking code in the loop.  Column D of the V image is 1.0: row D of O^T accumulates
//   sum_k P[k], the softmax denominator, on the matrix core.
//   Lazy maximum (as attention.hip): m is raised, and O rescaled, only when a query's scores exceed m by more than 2^8;
//   the decision for tile t is taken before its exponentials (textbook order), in a wave-uniform cold branch.
#include "common.h"

#include <type_traits>

namespace ldm {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

struct Attn32Params {
  const void* q; const void* k; const void* v; void* out;
  int q_stride, kv_stride, out_stride;
  int batch, heads, lq, lk;
  float scale_log2e;
  float* lse;
};

template <int D>
struct Attn32Geom {
  static_assert(D % 8 == 0 && D >= 8 && D <= 56, "head dims 8..56 in multiples of 8");
  static constexpr int KS = (D + 2 + 15) / 16;       // 16-deep k-steps of Q K^T: dims + the -m column + the mask column
  static constexpr int DP = KS * 16;                 // padded dims of the K image
  static constexpr int MT = (D + 1 + 31) / 32;       // 32-row tiles of O^T: dims + the denominator row
  static constexpr int KROW = DP * 2 + 16;           // bytes per key row (+16: ds_read_b128 of 16 distinct rows is conflict-free)
  static constexpr int VROW = 192;                   // bytes per key row of the V image: >= MT * 64, and 4 rows x 64 B of a
                                                     // transposed read fall on 4 disjoint bank spans (192 = 48 dwords)
  static_assert(MT * 64 <= VROW, "V image row too short");
  static constexpr int K_BYTES = 64 * KROW, V_BYTES = 64 * VROW;
  static constexpr int LDS_BYTES = 2 * K_BYTES + 2 * V_BYTES;
  static constexpr int CPR = D / 8;                  // 16-byte chunks per row in global memory
  static constexpr int NCH = (64 * CPR + 255) / 256; // chunks per thread per tile and operand
  static constexpr int AUG_KS = D / 16, AUG_H = (D % 16) / 8;              // where dims D (and D + 1) sit in a Q fragment
  static constexpr int ONE_MT = D / 32, ONE_LOCAL = D % 32;                // where row D sits in the O^T accumulators
  static constexpr int ONE_H = (ONE_LOCAL >> 2) & 1, ONE_REG = (ONE_LOCAL & 3) + 4 * (ONE_LOCAL >> 3);
};

__device__ __forceinline__ uint32_t cvt_pk_bf16_32(float a, float b) {
  typedef __attribute__((ext_vector_type(2))) float f2;
  typedef __attribute__((ext_vector_type(2))) __bf16 b2;
  f2 v = {a, b};
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(v, b2));
}

// The wave-uniform tile offset goes into the DESCRIPTOR (base advanced, record count shrunk), not into the instruction's
// scalar-offset field: LLVM documents soffset as EXCLUDED from the range check of raw buffer loads, and the rows past lk
// of the last, ragged tile must come back as zeros (never as whatever lies behind this (sample, head)'s rows).
__device__ __forceinline__ uint4 buffer_load16_32(const void* base, unsigned bytes, unsigned voff, unsigned soff) {
  typedef __attribute__((ext_vector_type(4))) unsigned int u32x4;
  const unsigned left = bytes - soff;   // callers fetch existing tiles only: soff < bytes (a clamped form compiles to a VALU saturating subtract and drags the descriptor into a waterfall loop)
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(
      const_cast<unsigned char*>(static_cast<const unsigned char*>(base) + soff), 0, (int)left, 0x00020000);
  const u32x4 r = __builtin_amdgcn_raw_buffer_load_b128(rs, (int)voff, 0, 0);
  return make_uint4(r[0], r[1], r[2], r[3]);
}

constexpr float ATTN32_MASK_BIG = 30000.0f;   // score offset of keys past lk: exp2(-30000) == 0 in float32

template <int D>
__global__ void __launch_bounds__(256, 2)
attention32_bf16_kernel(const Attn32Params p) {
  using G = Attn32Geom<D>;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned char* const kbuf = smem;                          // 2 x K image
  unsigned char* const vbuf = smem + 2 * G::K_BYTES;         // 2 x V image

  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int qc = lane & 31, h = lane >> 5;
  // workgroup -> (sample, head, query block), every XCD a contiguous run (the query blocks of a (sample, head) share K/V in one L2)
  const int nqb = gridDim.x;
  const int nblk = nqb * gridDim.y * gridDim.z;
  const int lin = (blockIdx.z * gridDim.y + blockIdx.y) * nqb + blockIdx.x;
  const int xq = nblk >> 3, xr = nblk & 7, xcd = lin & 7;
  const int work = (xcd < xr ? xcd * (xq + 1) : xr * (xq + 1) + (xcd - xr) * xq) + (lin >> 3);
  const int bh = work / nqb;
  const int b = bh / p.heads, hd = bh - b * p.heads;
  const int q0 = (work - bh * nqb) * 128 + wave * 32;

  const bf16_t* qp = static_cast<const bf16_t*>(p.q) + (size_t)b * p.lq * p.q_stride + hd * D;
  const bf16_t* kp = static_cast<const bf16_t*>(p.k) + (size_t)b * p.lk * p.kv_stride + hd * D;
  const bf16_t* vp = static_cast<const bf16_t*>(p.v) + (size_t)b * p.lk * p.kv_stride + hd * D;
  bf16_t* op = static_cast<bf16_t*>(p.out) + (size_t)b * p.lq * p.out_stride + hd * D;
  const unsigned kv_bytes = ((unsigned)(p.lk - 1) * p.kv_stride + D) * 2u;   // this (sample, head)'s rows, < 2 GiB (host)
  const int ntiles = (p.lk + 63) >> 6;

  // zero the images once (pad columns are never written again), then the constant columns
  for (int i = tid * 16; i < G::LDS_BYTES; i += 256 * 16) *reinterpret_cast<uint4*>(smem + i) = make_uint4(0, 0, 0, 0);
  __syncthreads();
  if (tid < 128) {
    *reinterpret_cast<bf16_t*>(kbuf + (tid >> 6) * G::K_BYTES + (tid & 63) * G::KROW + D * 2) = 0x3F80;   // 1.0
  } else {
    *reinterpret_cast<bf16_t*>(vbuf + ((tid - 128) >> 6) * G::V_BYTES + (tid & 63) * G::VROW + D * 2) = 0x3F80;
  }

  // Q^T fragments (B operand): lane (query qc, half h) holds Q[query][16 ks + 8 h + 0..7] * log2(e)/sqrt(d), bf16
  bf16x8_t qf[G::KS];
  {
    const int qi = q0 + qc;
#pragma unroll
    for (int ks = 0; ks < G::KS; ++ks) {
      const int col = ks * 16 + h * 8;
      bf16x8_t z = {0, 0, 0, 0, 0, 0, 0, 0};
      if (qi < p.lq && col < D) z = *reinterpret_cast<const bf16x8_t*>(qp + (size_t)qi * p.q_stride + col);
#pragma unroll
      for (int e = 0; e < 8; e += 2) {
        const uint32_t w = cvt_pk_bf16_32(bf16_to_f32((bf16_t)z[e]) * p.scale_log2e, bf16_to_f32((bf16_t)z[e + 1]) * p.scale_log2e);
        z[e] = (short)(w & 0xffffu);
        z[e + 1] = (short)(w >> 16);
      }
      qf[ks] = z;
    }
    // element D: -m (0 for now); element D + 1: -1 (times the mask column of K)
    if (h == G::AUG_H) qf[G::AUG_KS][1] = (short)0xBF80;
  }

  // staging geometry of this thread: loop-invariant
  unsigned g_off[G::NCH];
  int k_lds[G::NCH], v_lds[G::NCH];
#pragma unroll
  for (int c = 0; c < G::NCH; ++c) {
    const int idx = c * 256 + tid;
    const int row = idx / G::CPR, ch = idx - row * G::CPR;
    const bool ok = idx < 64 * G::CPR;
    g_off[c] = ok ? ((unsigned)row * p.kv_stride + ch * 8) * 2u : 0x80000000u;
    k_lds[c] = ok ? row * G::KROW + ch * 16 : -1;
    v_lds[c] = row * G::VROW + ch * 16;
  }
  uint4 kreg[G::NCH], vreg[G::NCH];
  auto fetch_k = [&](int t) {
    const unsigned soff = (unsigned)t * 64u * p.kv_stride * 2u;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) kreg[c] = buffer_load16_32(kp, kv_bytes, g_off[c], soff);
  };
  auto fetch_v = [&](int t) {
    const unsigned soff = (unsigned)t * 64u * p.kv_stride * 2u;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c) vreg[c] = buffer_load16_32(vp, kv_bytes, g_off[c], soff);
  };
  const int ragged = p.lk & 63;
  auto commit_k = [&](int t) {
    unsigned char* buf = kbuf + (t & 1) * G::K_BYTES;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c)
      if (k_lds[c] >= 0) *reinterpret_cast<uint4*>(buf + k_lds[c]) = kreg[c];
    // rows past lk of the last, ragged tile: +BIG in the mask column (the buffer is not used again afterwards)
    if (ragged != 0 && t == ntiles - 1 && tid < 64 && tid >= ragged)
      *reinterpret_cast<bf16_t*>(buf + tid * G::KROW + (D + 1) * 2) = (bf16_t)(__float_as_uint(ATTN32_MASK_BIG) >> 16);
  };
  auto commit_v = [&](int t) {
    unsigned char* buf = vbuf + (t & 1) * G::V_BYTES;
#pragma unroll
    for (int c = 0; c < G::NCH; ++c)
      if (k_lds[c] >= 0) *reinterpret_cast<uint4*>(buf + v_lds[c]) = vreg[c];
  };

  // fragment addresses: K rows by ds_read_b128, V^T by transposed reads (4 keys x 16 dims per 16-lane group)
  const int k_frag = qc * G::KROW + h * 16;
  const int v_frag = (h * 4 + ((lane & 15) >> 2)) * G::VROW + (16 * ((lane >> 4) & 1) + 4 * (lane & 3)) * 2;

  auto qk = [&](int t, f32x16_t (&s)[2]) {
    const unsigned char* kl = kbuf + (t & 1) * G::K_BYTES + k_frag;
#pragma unroll
    for (int kb = 0; kb < 2; ++kb) {
      f32x16_t acc;
#pragma unroll
      for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
      for (int ks = 0; ks < G::KS; ++ks) {
        const bf16x8_t kf = *reinterpret_cast<const bf16x8_t*>(kl + kb * 32 * G::KROW + ks * 32);
        acc = __builtin_amdgcn_mfma_f32_32x32x16_bf16(kf, qf[ks], acc, 0, 0, 0);
      }
      s[kb] = acc;
    }
  };
  // maximum of a query's 64 scores of one tile (the two lanes that share the query exchange theirs)
  auto tile_max = [&](const f32x16_t (&s)[2]) {
    float mx = s[0][0];
#pragma unroll
    for (int kb = 0; kb < 2; ++kb)
#pragma unroll
      for (int r = 0; r < 16; ++r) mx = fmaxf(mx, s[kb][r]);
    const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(mx), __float_as_uint(mx), false, false);
    return fmaxf(__uint_as_float(sw[0]), __uint_as_float(sw[1]));
  };

  f32x16_t o[G::MT];
#pragma unroll
  for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
    for (int r = 0; r < 16; ++r) o[mt][r] = 0.f;
  float m_run = 0.f;                       // element D of Q holds -m_run = 0

  // prologue: K(0) in LDS, K(1) and V(0) in registers; then the peeled "iteration -1" computes Q K^T of tile 0
  fetch_k(0);
  commit_k(0);
  if (ntiles > 1) fetch_k(1);
  fetch_v(0);
  __syncthreads();
  if (ntiles > 1) commit_k(1);
  commit_v(0);
  if (ntiles > 2) fetch_k(2);
  if (ntiles > 1) fetch_v(1);
  f32x16_t s_a[2], s_b[2];
  qk(0, s_a);
  float mx = tile_max(s_a);

  // One key tile.  sc: scores of tile t (computed one step earlier), sn: receives the scores of tile t+1.
  auto step = [&](int t, f32x16_t (&sc)[2], f32x16_t (&sn)[2], auto has_next_tag) {
    constexpr bool HAS_NEXT = decltype(has_next_tag)::value;
    __syncthreads();   // K(t+1), V(t) are in LDS for every wave; every wave is done with K(t) and V(t-1)
    if (t + 2 < ntiles) commit_k(t + 2);
    if (t + 1 < ntiles) commit_v(t + 1);
    if (t + 3 < ntiles) fetch_k(t + 3);
    if (t + 2 < ntiles) fetch_v(t + 2);

    // ---- decision for tile t (before its exponentials): raise the running maximum?  Wave-uniform, cold after the first tiles.
    {
      const bool raise = (t == 0) || mx > 8.0f;
      if (__builtin_amdgcn_ballot_w64(raise) != 0) {
        // m_run stays bf16-representable: its copy inside the Q fragment must be exact
        const float m_new = __uint_as_float(cvt_pk_bf16_32(m_run + (raise ? mx : 0.f), 0.f) << 16);
        const float d2 = m_new - m_run;
        const float alpha = __builtin_amdgcn_exp2f(-d2);
        m_run = m_new;
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
          for (int r = 0; r < 16; ++r) o[mt][r] *= alpha;
#pragma unroll
        for (int kb = 0; kb < 2; ++kb)
#pragma unroll
          for (int r = 0; r < 16; ++r) sc[kb][r] -= d2;
        if (h == G::AUG_H) qf[G::AUG_KS][0] = (short)(__float_as_uint(-m_new) >> 16);
      }
    }

    // ---- one basic block: Q K^T of tile t+1 | exponentials + packing of tile t | P V of tile t | maximum of tile t+1
    if constexpr (HAS_NEXT) qk(t + 1, sn);
    bf16x8_t pb[4];
#pragma unroll
    for (int s2 = 0; s2 < 4; ++s2) {
      typedef __attribute__((ext_vector_type(4))) uint32_t u4;
      u4 w;
#pragma unroll
      for (int j = 0; j < 4; ++j) {
        const float e0 = __builtin_amdgcn_exp2f(sc[s2 >> 1][8 * (s2 & 1) + 2 * j]);       // raw v_exp_f32: arguments <= 8
        const float e1 = __builtin_amdgcn_exp2f(sc[s2 >> 1][8 * (s2 & 1) + 2 * j + 1]);
        w[j] = cvt_pk_bf16_32(e0, e1);
      }
      pb[s2] = __builtin_bit_cast(bf16x8_t, w);
    }
    {
      const unsigned char* vl = vbuf + (t & 1) * G::V_BYTES + v_frag;
#pragma unroll
      for (int s2 = 0; s2 < 4; ++s2) {
#pragma unroll
        for (int mt = 0; mt < G::MT; ++mt) {
          const unsigned char* a0 = vl + (16 * s2) * G::VROW + mt * 64;
          const bf16x4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)a0);
          const bf16x4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)(a0 + 8 * G::VROW));
          const bf16x8_t vf = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
          o[mt] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(vf, pb[s2], o[mt], 0, 0, 0);
        }
      }
    }
    if constexpr (HAS_NEXT) mx = tile_max(sn);
  };

  // two tiles per trip with the score sets swapping roles (no register copies); the last tile has no successor
  int t = 0;
  for (; t + 2 < ntiles; t += 2) {
    step(t, s_a, s_b, std::true_type{});
    step(t + 1, s_b, s_a, std::true_type{});
  }
  if (t + 1 < ntiles) {
    step(t, s_a, s_b, std::true_type{});
    step(t + 1, s_b, s_a, std::false_type{});
  } else {
    step(t, s_a, s_b, std::false_type{});
  }

  // ---- normalise and store: lane (query qc, half h) holds dims 32 mt + 8 g + 4 h + 0..3 in registers 4 g .. 4 g + 3
  float l = o[G::ONE_MT][G::ONE_REG];
  l = __shfl(l, qc + 32 * G::ONE_H, 64);
  const float inv = 1.0f / l;
  const int qi = q0 + qc;
  if (qi >= p.lq) return;
  if (p.lse != nullptr && h == 0) p.lse[((size_t)b * p.heads + hd) * p.lq + qi] = m_run + __log2f(l);
#pragma unroll
  for (int mt = 0; mt < G::MT; ++mt)
#pragma unroll
    for (int g = 0; g < 4; ++g) {
      const int col = 32 * mt + 8 * g + 4 * h;
      if (col < D) {
        uint2 w;
        w.x = cvt_pk_bf16_32(o[mt][4 * g] * inv, o[mt][4 * g + 1] * inv);
        w.y = cvt_pk_bf16_32(o[mt][4 * g + 2] * inv, o[mt][4 * g + 3] * inv);
        *reinterpret_cast<uint2*>(op + (size_t)qi * p.out_stride + col) = w;
      }
    }
}

template <int D>
static int launch_attention32(const Attn32Params& p, hipStream_t st) {
  using G = Attn32Geom<D>;
  dim3 grid((p.lq + 127) / 128, p.heads, p.batch);
  hipLaunchKernelGGL((attention32_bf16_kernel<D>), grid, dim3(256), G::LDS_BYTES, st, p);
  LDM_HIP_CHECK_LAUNCH("ldm_attention(bf16 mfma 32x32)");
  return LDM_OK;
}

// Entry for attention.hip's dispatcher: true when this kernel took the call.
bool attention32_try(const void* q, int q_stride, const void* k, const void* v, int kv_stride, void* out, int out_stride,
                     int batch, int heads, int lq, int lk, int d, float scale_log2e, float* lse, hipStream_t st, int* rc) {
  if (d != 40) return false;
  Attn32Params p;
  p.q = q; p.k = k; p.v = v; p.out = out;
  p.q_stride = q_stride; p.kv_stride = kv_stride; p.out_stride = out_stride;
  p.batch = batch; p.heads = heads; p.lq = lq; p.lk = lk;
  p.scale_log2e = scale_log2e;
  p.lse = lse;
  *rc = launch_attention32<40>(p, st);
  return true;
}

}  // namespace ldm
