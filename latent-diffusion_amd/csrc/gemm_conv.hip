// Implicit-GEMM convolution / linear layer on the gfx950 matrix cores.
//
//   out[m][n] = epilogue( sum_k A[m][k] * W[n][k] )      m = (sample, oh, ow), k = (kh, kw, ci)
//
// A is never materialised: every 16-byte chunk of a K-tile is fetched straight from the NHWC activation
// (one or two channel-concatenated sources, optional stride 2, optional nearest-2x upsample folded into the
// index) into LDS by LDS-DMA (buffer_load_dwordx4 ... lds), padding taps zero-filled by the buffer range check.  W is packed
// [n][kh][kw][ci], so both MFMA operands are K-contiguous.  bf16 inputs, fp32 accumulation on
// v_mfma_f32_16x16x32_bf16; the fp32 instantiation (exact fp32, v_mfma_f32_16x16x4_f32) serves the 1e-4 gate.
//
// LDS image of a tile: [rows][BK] bf16, 16-byte chunks XOR-swizzled by row; LDS-DMA writes lane-linear, so
// the swizzle is applied to the per-lane SOURCE chunk and again on the fragment read (same involution).
#include "common.h"

#include <type_traits>

namespace ldm {

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;

struct ConvGemmParams {
  const void* a0;
  const void* a1;
  int c0, c1, ctot;
  int batch, hin, win, hout, wout;
  int ksize, stride, upsample;
  const void* w;
  int M, N, K;
  int epilogue;
  const float* bias;
  const void* rowvec;
  int rowvec_stride;
  const void* residual;
  void* out;
  int out_f32;      // out dtype is float32 (else same as compute dtype)
  int tiles_m, tiles_n;
  int vec_epilogue; // 16-byte epilogue accesses are legal (alignment and n % 8 checked on the host)
  float* rowstats;        // LDM_EPI_ROWSTATS: [M][N / 80][2] partial (sum, sum of squares) of the rounded outputs per row
  const float* ln_stats;  // LDM_EPI_LNFOLD: the producer's rowstats of THIS call's A operand, [M][ln_nseg][2]
  int ln_nseg;
  const float* ln_s1;     //   [N]: sum_k W'[n][k] of the gamma-scaled filter this call multiplies with
  float ln_eps;
  void* glu_preact; // GLU epilogues, bf16: also store the biased pre-activation [M][N] (paired column order) -- the training forward
  int nt_out;       // bf16 output stored with the non-temporal hint: set by the host for outputs far larger than the caches
                    // (>= LDM_GEMM_NT_MB, default 128 MB: the 64x64 level's q|k|v and feed-forward outputs), which the next
                    // kernel reads from HBM either way; smaller outputs keep plain stores (their consumer finds them in the
                    // Infinity Cache: the 84 MB output of a 64x64-level conv was 3 % SLOWER with the hint)
  int splits;       // split-K: blockIdx.y takes K-tiles [y*nk/splits, (y+1)*nk/splits) and writes raw fp32 partial sums
  float* partial;   //          to partial[y][M][N]; splitk_reduce_kernel folds them and applies the epilogue
#ifdef LDM_GEMM_STAMPS
  unsigned long long* stamps;  // diagnostic build only (make stamps): 16 cycle stamps per workgroup
  int ablate;                  // diagnostic build only (LDM_GEMM_ABLATE): 1 = no LDS-DMA in the ring loop, 2 = no MFMA (fragment
                               // reads kept), 4 = neither fragment reads nor MFMA -- results are garbage, only the timing is read
#endif
};

// Diagnostic build (make stamps -> libldm_amd_stamps.so, tools/gemm_stamps.py): thread 0 of every workgroup records
// the shader clock at the phase boundaries of the bf16 kernel.  The product library compiles none of it.
#ifdef LDM_GEMM_STAMPS
#define LDM_STAMP(i)                                                                                      \
  do {                                                                                                    \
    if (p.stamps != nullptr && threadIdx.x == 0)                                                          \
      p.stamps[(size_t)blockIdx.x * 16 + (i)] = ((i) == 5 || (i) == 6) ? __builtin_amdgcn_s_memrealtime() : __builtin_amdgcn_s_memtime(); \
  } while (0)
#else
#define LDM_STAMP(i) do {} while (0)
#endif
#ifdef LDM_GEMM_STAMPS
#define LDM_ABLATE(bit) (p.ablate & (bit))
#else
#define LDM_ABLATE(bit) false
#endif

template <int CH>
__device__ __forceinline__ int swz(int row) {
  return CH == 8 ? (row & 7) : ((row >> 1) & 3);
}

// bijective XCD remap: blocks b and b+8 share an XCD; give every XCD a contiguous run of tiles
__device__ __forceinline__ int xcd_remap(int bid, int nblk) {
  const int q = nblk >> 3, r = nblk & 7, x = bid & 7;
  return (x < r ? x * (q + 1) : r * (q + 1) + (x - r) * q) + (bid >> 3);
}

// Raster order of the output tiles: bands of GROUP_M tile-rows, walked column by column.  The blocks in flight on
// one XCD (a contiguous run of tile indices after xcd_remap) then cover a GROUP_M x ~8 patch: its A tiles and
// filter tiles are fetched from HBM once and re-read from that XCD's L2, instead of streaming all of A once per
// tile-column (at K = 320 and N = 2560 that was 20 passes over an 84 MB activation).
constexpr int GROUP_M = 8;
__device__ __forceinline__ void tile_to_mn(int tile, int tiles_m, int tiles_n, int& tm, int& tn) {
  const int band = GROUP_M * tiles_n;
  const int g = tile / band, in = tile - g * band;
  const int first = g * GROUP_M;
  const int rows = min(tiles_m - first, GROUP_M);
  tn = in / rows;
  tm = first + (in - tn * rows);
}

// 16 bytes per lane from a raw buffer (base, bytes) straight into LDS: `buffer_load_dwordx4 v, s[rsrc], s offen lds`.
// voff is the per-lane byte offset, soff a wave-uniform one; an offset past `bytes` makes the hardware write zeros.
// (Kept in a __device__ function: the resource type does not exist in the host pass that emits the kernel stub.)
__device__ __forceinline__ void dma16_to_lds(const void* base, unsigned bytes, unsigned char* lds, unsigned voff, int soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

template <int BM, int BN, int BK, int WM, int WN, int STAGES, int PP = 0>
__global__ void __launch_bounds__(64 * WM * WN, (BM == 128 && BN == 128 && BK == 32) ? 3 : 2)   // 128x128x32: three workgroups per CU (short-K GLU layers)
conv_gemm_bf16_kernel(const ConvGemmParams p) {
  constexpr int NT = 64 * WM * WN;
  constexpr int CH = BK / 8;                 // 16-byte chunks per tile row
  constexpr int ROWB = BK * 2;               // bytes per tile row
  constexpr int A_BYTES = BM * ROWB, B_BYTES = BN * ROWB;
  constexpr int STAGE = A_BYTES + B_BYTES;
  constexpr int A_IT = BM * CH / NT, B_IT = (BN * CH + NT - 1) / NT;
  constexpr int B_FULL = BN * CH / NT;       // B staging passes every wave takes part in
  static_assert(BM * CH % NT == 0 && (BN * CH) % 64 == 0, "tile/threads mismatch");
  constexpr int WTM = BM / WM, WTN = BN / WN;
  constexpr int MI = WTM / 16, NI = WTN / 16, KS = BK / 32;
  static_assert(WTM % 16 == 0 && WTN % 16 == 0, "wave tile must be a multiple of 16");

  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  LDM_STAMP(0);
  LDM_STAMP(5);

  const int tid = threadIdx.x;
  const int lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int wm = wave / WN, wn = wave % WN;

  const int tile = xcd_remap(blockIdx.x, gridDim.x);
  int tm, tn;
  tile_to_mn(tile, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;

  // ---- LDM_EPI_LNFOLD: mean / rstd of this workgroup's rows of A from the producer's per-segment partial sums (fixed
  // order: deterministic); thread r owns row m0 + r.  The loads are ISSUED here, ahead of the first LDS-DMA, and folded
  // only after the prologue's DMA has been issued (ln_finish below): consumed at once, their memory latency came in front of
  // the tile's first operand fetch -- a tenth of the life of a K = 320 tile (GLU launches: +27 us with the fold on).
  constexpr int LN_MAX_SEG = 16;             // 80-column segments of a row: c0 <= 1280 (checked on the host)
  float2 ln_raw[LN_MAX_SEG];
  const bool ln_row = p.ln_stats != nullptr && tid < BM && m0 + tid < p.M;
  if (ln_row) {
    const float2* st = reinterpret_cast<const float2*>(p.ln_stats) + (size_t)(m0 + tid) * p.ln_nseg;
#pragma unroll
    for (int i = 0; i < LN_MAX_SEG; ++i) ln_raw[i] = i < p.ln_nseg ? st[i] : make_float2(0.f, 0.f);
  }
  float ln_mu = 0.f, ln_r = 0.f;
  auto ln_finish = [&]() {
    if (ln_row) {
      float sx = 0.f, sq = 0.f;
#pragma unroll
      for (int i = 0; i < LN_MAX_SEG; ++i) { sx += ln_raw[i].x; sq += ln_raw[i].y; }
      const float inv_k = 1.0f / (float)p.K;
      ln_mu = sx * inv_k;
      const float var = fmaxf(sq * inv_k - ln_mu * ln_mu, 0.f);
      ln_r = rsqrtf(var + p.ln_eps);
    }
  };

  // ---- operand staging: LDS-DMA through buffer descriptors.  Per K-tile a wave issues A_IT + B_IT
  // `buffer_load_dwordx4 ... offen lds` with a per-lane byte offset that stays fixed while the K loop walks the
  // channels of one (tap, source) and a SCALAR offset that advances by one K-tile: no vector ALU work per tile.
  // Padding taps, rows past M and filter rows past N carry an out-of-range offset: the descriptor's range check
  // makes the hardware write zeros to LDS for them.
  constexpr unsigned OOB = 0x80000000u;      // every buffer here is < 2 GiB (checked on the host)
  const unsigned bytes_a0 = (unsigned)p.batch * p.hin * p.win * p.c0 * 2u;
  const unsigned bytes_a1 = (unsigned)p.batch * p.hin * p.win * (p.c1 ? p.c1 : p.c0) * 2u;
  const unsigned bytes_w = (unsigned)p.N * p.K * 2u;

  const int cpos = tid % CH;                 // chunk position inside the LDS row this thread fills
  int a_pix[A_IT], a_ih[A_IT], a_iw[A_IT], a_sw[A_IT];
  const int hw_out = p.hout * p.wout;
  // pixel -> (sample, row, column): shifts when the output extents are powers of two (every level of a 64x64 or 32x32
  // latent), and none at all for a pointwise stride-1 layer, whose input pixel IS the output row (block-uniform tests;
  // the general path costs eight ~40-instruction integer divisions per thread before the first load is issued)
  const bool ext_pow2 = ((hw_out & (hw_out - 1)) | (p.wout & (p.wout - 1))) == 0;
  const int sh_hw = __builtin_ctz(hw_out), sh_w = __builtin_ctz(p.wout);
  const bool pointwise = p.ksize == 1 && p.stride == 1 && !p.upsample;
  auto sample_of = [&](int m) { return ext_pow2 ? m >> sh_hw : m / hw_out; };
  const int hlim = p.upsample ? p.hin * 2 : p.hin;
  const int wlim = p.upsample ? p.win * 2 : p.win;
  const int pad = p.ksize >> 1;
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int r = (it * NT + tid) / CH;
    const int m = m0 + r;
    a_sw[it] = (cpos ^ swz<CH>(r)) * 8;      // source element offset inside the K-tile
    if (m < p.M && pointwise) {
      a_pix[it] = m;
      a_ih[it] = a_iw[it] = 0;
    } else if (m < p.M) {
      const int b = sample_of(m), rem = m - b * hw_out;
      const int oh = ext_pow2 ? rem >> sh_w : rem / p.wout, ow = rem - oh * p.wout;
      a_pix[it] = b * p.hin * p.win;
      a_ih[it] = oh * p.stride - pad;
      a_iw[it] = ow * p.stride - pad;
    } else {
      a_pix[it] = -1;
      a_ih[it] = a_iw[it] = 0;
    }
  }
  unsigned b_voff[B_IT];
#pragma unroll
  for (int it = 0; it < B_IT; ++it) {
    const int r = (it * NT + tid) / CH;      // may run past BN in the last, partial pass (never staged then)
    const int n = n0 + r;
    b_voff[it] = n < p.N ? ((unsigned)n * p.K + (cpos ^ swz<CH>(r)) * 8) * 2u : OOB;
  }

  // this workgroup's K-tiles (all of them unless split-K)
  const int nk_all = p.K / BK;
  const int kt_begin = p.splits > 1 ? (int)((long long)blockIdx.y * nk_all / p.splits) : 0;
  const int kt_end = p.splits > 1 ? (int)((long long)(blockIdx.y + 1) * nk_all / p.splits) : nk_all;
  const int nk = kt_end - kt_begin;

  // K-loop iterator (block-uniform): tap -> source -> channel offset inside the source
  int it_tap = (kt_begin * BK) / p.ctot;
  int it_cc = kt_begin * BK - it_tap * p.ctot;
  int it_second = it_cc >= p.c0 ? 1 : 0;
  if (it_second) it_cc -= p.c0;
  bool it_fresh = true;
  unsigned a_voff[A_IT];
  // A K-tile is staged as NPIECE wave-instructions of 1 KiB (A_IT of A, then B_IT of the filter).  stage_begin fixes
  // the tile's addressing state and advances the iterator; stage_piece(i) issues one of them, so the ring loop can
  // spread them between its MFMAs.
  constexpr int NPIECE = A_IT + B_IT;
  const void* st_abase = p.a0;
  unsigned st_abytes = bytes_a0;
  int st_soff_a = 0, st_soff_b = 0;
  unsigned char* st_sa = smem;
  auto stage_begin = [&](int buf, int kt) {
    st_sa = smem + buf * STAGE;
    const int cs = it_second ? p.c1 : p.c0;
    if (it_fresh) {                          // new (tap, source): per-lane offsets change
      const int kh = it_tap / p.ksize, kw = it_tap - kh * p.ksize;
#pragma unroll
      for (int it = 0; it < A_IT; ++it) {
        int ih = a_ih[it] + kh, iw = a_iw[it] + kw;
        bool ok = a_pix[it] >= 0 && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
        if (p.upsample == 2) ok = ok && ((ih | iw) & 1) == 0;   // zero insertion: only even positions hold data
        if (p.upsample) { ih >>= 1; iw >>= 1; }
        a_voff[it] = ok ? ((unsigned)(a_pix[it] + ih * p.win + iw) * cs + a_sw[it]) * 2u : OOB;
      }
      it_fresh = false;
    }
    st_abase = it_second ? p.a1 : p.a0;
    st_abytes = it_second ? bytes_a1 : bytes_a0;
    st_soff_a = it_cc * 2;
    st_soff_b = kt * (BK * 2);
    // advance the iterator
    it_cc += BK;
    if (it_cc >= cs) {
      it_cc = 0;
      it_fresh = true;
      if (!it_second && p.c1 > 0) it_second = 1;
      else { it_second = 0; ++it_tap; }
    }
  };
  auto stage_piece = [&](auto piece_tag) {
    constexpr int PI = decltype(piece_tag)::value;
    if constexpr (PI < A_IT) {
      dma16_to_lds(st_abase, st_abytes, st_sa + (PI * NT + wave * 64) * 16, a_voff[PI], st_soff_a);
    } else if constexpr (PI < NPIECE) {
      constexpr int it = PI - A_IT;
      if (it < B_FULL || (it * NT + wave * 64) < BN * CH)   // wave-uniform
        dma16_to_lds(p.w, bytes_w, st_sa + A_BYTES + (it * NT + wave * 64) * 16, b_voff[it], st_soff_b);
    }
  };
  auto stage_all = [&](auto... tags) { (stage_piece(tags), ...); };
  auto stage = [&](int buf, int kt) {
    stage_begin(buf, kt);
    [&]<int... I>(std::integer_sequence<int, I...>) { stage_all(std::integral_constant<int, I>{}...); }
    (std::make_integer_sequence<int, NPIECE>{});
  };

  f32x4_t acc[MI][NI];
#pragma unroll
  for (int i = 0; i < MI; ++i)
#pragma unroll
    for (int j = 0; j < NI; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;

  // Ring loop body: the math of one K-tile with the LDS-DMA pieces of the tile two ahead issued between the MFMAs
  // (one piece every GAP MFMAs).  Issued in one burst after the barrier, the pieces of all 8 waves queue behind the
  // CU's ~38 B/clk L2->LDS path for ~1,200 cycles during which no wave can issue an MFMA; spread out, the matrix
  // cores run while the path drains.
  auto compute_and_stage = [&](int buf, bool do_stage) {
    const unsigned char* sa = smem + buf * STAGE;
    const unsigned char* sb = sa + A_BYTES;
    constexpr int NMFMA = KS * MI * NI;
    constexpr int GAP = NMFMA / (NPIECE + 1) > 0 ? NMFMA / (NPIECE + 1) : 1;
    [&]<int... KSI>(std::integer_sequence<int, KSI...>) {
      ([&] {
        constexpr int ks = KSI;
        bf16x8_t af[MI], bfr[NI];
        if (!LDM_ABLATE(4)) {
#pragma unroll
        for (int i = 0; i < MI; ++i) {
          const int row = wm * WTM + i * 16 + fr;
          af[i] = *reinterpret_cast<const bf16x8_t*>(sa + row * ROWB + (((ks * 4 + fq) ^ swz<CH>(row)) << 4));
        }
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int row = wn * WTN + j * 16 + fr;
          bfr[j] = *reinterpret_cast<const bf16x8_t*>(sb + row * ROWB + (((ks * 4 + fq) ^ swz<CH>(row)) << 4));
        }
        }
#ifdef LDM_GEMM_STAMPS
        if (LDM_ABLATE(2)) {
#pragma unroll
          for (int i = 0; i < MI; ++i) asm volatile("" ::"v"(af[i]));
#pragma unroll
          for (int j = 0; j < NI; ++j) asm volatile("" ::"v"(bfr[j]));
        }
#endif
        [&]<int... IJ>(std::integer_sequence<int, IJ...>) {
          ([&] {
            constexpr int i = IJ / NI, j = IJ % NI;
            constexpr int seq = ks * MI * NI + IJ;           // MFMA number inside the K-tile
            if (!LDM_ABLATE(6)) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
            if constexpr ((seq + 1) % GAP == 0 && (seq + 1) / GAP - 1 < NPIECE) {
              if (do_stage && !LDM_ABLATE(1)) {
                __builtin_amdgcn_sched_barrier(0);
                stage_piece(std::integral_constant<int, (seq + 1) / GAP - 1>{});
                __builtin_amdgcn_sched_barrier(0);
              }
            }
          }(), ...);
        }(std::make_integer_sequence<int, MI * NI>{});
      }(), ...);
    }(std::make_integer_sequence<int, KS>{});
  };

  auto compute = [&](int buf) {
    const unsigned char* sa = smem + buf * STAGE;
    const unsigned char* sb = sa + A_BYTES;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
      bf16x8_t af[MI], bfr[NI];
#pragma unroll
      for (int i = 0; i < MI; ++i) {
        const int row = wm * WTM + i * 16 + fr;
        af[i] = *reinterpret_cast<const bf16x8_t*>(sa + row * ROWB + (((ks * 4 + fq) ^ swz<CH>(row)) << 4));
      }
#pragma unroll
      for (int j = 0; j < NI; ++j) {
        const int row = wn * WTN + j * 16 + fr;
        bfr[j] = *reinterpret_cast<const bf16x8_t*>(sb + row * ROWB + (((ks * 4 + fq) ^ swz<CH>(row)) << 4));
      }
#pragma unroll
      for (int i = 0; i < MI; ++i)
#pragma unroll
        for (int j = 0; j < NI; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[i], bfr[j], acc[i][j], 0, 0, 0);
    }
  };

  LDM_STAMP(1);
  if constexpr (STAGES == 2) {
    stage(0, kt_begin);
    ln_finish();
    __syncthreads();  // waits vmcnt(0) for the LDS-DMA, then barrier
    LDM_STAMP(2);
    for (int kt = 0; kt < nk; ++kt) {
      const int cur = kt & 1;
      if (kt == 8) LDM_STAMP(8);
      if (kt + 1 < nk) stage(cur ^ 1, kt_begin + kt + 1);
      if (kt == 8) LDM_STAMP(9);
      compute(cur);
      if (kt == 8) LDM_STAMP(10);
      __syncthreads();
      if (kt == 8) LDM_STAMP(11);
    }
  } else {
    // 3-slot ring, ONE raw barrier per K-tile, the LDS-DMA of the tile after next stays in flight across it:
    //   wait (counted) until this wave's loads of tile kt have landed -> barrier (now every wave's have, and every
    //   wave is done reading tile kt-1) -> refill the slot of tile kt-1 with tile kt+2 -> math on tile kt.
    // vmcnt counts this wave's LDS-DMA in issue order; a wave issues A_IT + B_FULL (+1 for the waves that take
    // part in the partial last B pass) per tile.
    static_assert(STAGES == 3, "ring depth");
    const bool extra = B_IT > B_FULL && (B_FULL * NT + wave * 64) < BN * CH;
    if constexpr (PP < 2) {
      stage(0, kt_begin);
      if (nk > 1) stage(1, kt_begin + 1);
      ln_finish();
    }
    int slot = 0;
    if constexpr (PP) {
      // Ping-pong form of the ring: the waves of a workgroup are two groups of WM*WN/2, one wave of each per SIMD, that run
      // the same two phases per K-tile in opposite order:
      //   R = read all fragments of the K-tile from LDS into registers + issue this wave's share of the LDS-DMA of the tile
      //       after next;   M = the K-tile's MFMAs, nothing else in the way (PP == 2: the next staging arithmetic under them)
      //   group 0:  | R(kt)   M(kt)   | R(kt+1) M(kt+1) | ...          '|' = the ONE workgroup barrier per K-tile
      //   group 1:  | M(kt-1) R(kt)   | M(kt)   R(kt+1) | ...
      // so the matrix cores of a SIMD always have one wave that only feeds them while the other waits on LDS and on the
      // DMA issue path (in the plain ring both waves of a SIMD reach the barrier and the staging code together: ~2,200
      // cycles per K-tile for 1,280 of matrix-core time).  What the barrier in front of K-tile kt guarantees: every wave's DMA
      // of tile kt has landed (each waited for its own before it), and every wave has finished reading tile kt-1 (group 1's
      // R(kt-1) is the second half of the interval before it), whose slot the tile kt+2 issued after the barrier refills.
      // A first version had a second barrier between the two phases (phase = max(R, M) + barrier); without it an interval
      // is R + M + barrier: -3..7 % per launch on the 3x3 convolutions (profiles/r3_gemm_pingpong_one_barrier.txt).
      static_assert((WM * WN) % 2 == 0, "two wave groups");
      bf16x8_t af[KS][MI], bfr[KS][NI];
#ifdef LDM_GEMM_STAMPS
#pragma unroll
      for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
        for (int i = 0; i < MI; ++i) af[ks][i] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < NI; ++j) bfr[ks][j] = bf16x8_t{0, 0, 0, 0, 0, 0, 0, 0};
      }
#endif
      auto read_frags = [&](int buf) {
        const unsigned char* sa = smem + buf * STAGE;
        const unsigned char* sb = sa + A_BYTES;
        if (LDM_ABLATE(4)) return;
#pragma unroll
        for (int ks = 0; ks < KS; ++ks) {
#pragma unroll
          for (int i = 0; i < MI; ++i) {
            const int row = wm * WTM + i * 16 + fr;
            af[ks][i] = *reinterpret_cast<const bf16x8_t*>(sa + row * ROWB + (((ks * 4 + fq) ^ swz<CH>(row)) << 4));
          }
#pragma unroll
          for (int j = 0; j < NI; ++j) {
            const int row = wn * WTN + j * 16 + fr;
            bfr[ks][j] = *reinterpret_cast<const bf16x8_t*>(sb + row * ROWB + (((ks * 4 + fq) ^ swz<CH>(row)) << 4));
          }
        }
      };
      auto mfma_all = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        if (!LDM_ABLATE(2))
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      };
      auto pp_stage = [&](int buf, int kt) {
        if (LDM_ABLATE(8)) return;
        if (LDM_ABLATE(1)) { stage_begin(buf, kt); return; }
        stage(buf, kt);
      };
      // PP == 2: the per-tile staging arithmetic as straight-line code that runs UNDER the MFMAs of the M phase (the R phase
      // then only issues instructions that move data).  Per lane and staged row: the byte offset of the row's input pixel at
      // tap (0, 0) for each source, and one validity bit per tap; a tap is then a scalar byte shift and a bit test.
      unsigned pb0[A_IT], pb1[A_IT], tapmask[A_IT];
      if constexpr (PP >= 2) {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
          const int r = (it * NT + tid) / CH;
          const int m = m0 + r;
          int pix = 0;
          unsigned mask = 0;
          if (m < p.M && pointwise) {
            pix = m;
            mask = 1;
          } else if (m < p.M) {
            const int b = sample_of(m), rem = m - b * hw_out;
            const int oh = ext_pow2 ? rem >> sh_w : rem / p.wout, ow = rem - oh * p.wout;
            const int ih0 = oh * p.stride - pad, iw0 = ow * p.stride - pad;
            pix = (b * p.hin + ih0) * p.win + iw0;
            unsigned cols = 0;
            for (int kw = 0; kw < p.ksize; ++kw) cols |= ((unsigned)(iw0 + kw) < (unsigned)p.win ? 1u : 0u) << kw;
            for (int kh = 0; kh < p.ksize; ++kh)
              if ((unsigned)(ih0 + kh) < (unsigned)p.hin) mask |= cols << (kh * p.ksize);
          }
          const unsigned sw = (unsigned)((cpos ^ swz<CH>(r)) * 8);
          pb0[it] = ((unsigned)pix * (unsigned)p.c0 + sw) * 2u;
          pb1[it] = ((unsigned)pix * (unsigned)p.c1 + sw) * 2u;
          tapmask[it] = mask;
        }
      }
      auto pp_prepare = [&](int buf, int kt) {
        st_sa = smem + buf * STAGE;
        const int t = it_tap;
        const int kh = (t * 11) >> 5, kw = t - kh * p.ksize;      // t / 3 for t < 12 (ksize 1: t == 0)
        const bool sec = it_second != 0;
        const int cs = sec ? p.c1 : p.c0;
        const unsigned shift = (unsigned)((kh * p.win + kw) * cs) * 2u;
#pragma unroll
        for (int it = 0; it < A_IT; ++it) {
          const unsigned base = sec ? pb1[it] : pb0[it];
          a_voff[it] = ((tapmask[it] >> t) & 1u) ? base + shift : OOB;
        }
        st_abase = sec ? p.a1 : p.a0;
        st_abytes = sec ? bytes_a1 : bytes_a0;
        st_soff_a = it_cc * 2;
        st_soff_b = kt * (BK * 2);
        const int cc = it_cc + BK;
        const bool wrap = cc >= cs;
        const bool to_second = wrap && !sec && p.c1 > 0;
        // (kept on the scalar unit: without the readfirstlane the compiler turns the flag arithmetic into vector code and
        // every use of the tap index with it)
        it_tap = __builtin_amdgcn_readfirstlane(it_tap + ((wrap && !to_second) ? 1 : 0));
        it_second = __builtin_amdgcn_readfirstlane(wrap ? (to_second ? 1 : 0) : it_second);
        it_cc = __builtin_amdgcn_readfirstlane(wrap ? 0 : cc);
      };
      // Pins the results of pp_prepare where they are: the compiler otherwise sinks the whole computation to its use, the
      // DMA issue of the next R phase -- the phase it was taken out of.
      auto pp_pin = [&]() {
#pragma unroll
        for (int it = 0; it < A_IT; ++it) asm volatile("" : "+v"(a_voff[it]));
        asm volatile("" : "+s"(st_soff_a), "+s"(st_soff_b), "+s"(st_abytes), "+s"(it_tap), "+s"(it_cc), "+s"(it_second));
      };
      auto pp_issue = [&]() {
        [&]<int... I>(std::integer_sequence<int, I...>) { stage_all(std::integral_constant<int, I>{}...); }
        (std::make_integer_sequence<int, NPIECE>{});
      };
      // M phase of PP == 2: the staging arithmetic of the tile three ahead, then the MFMAs, in ONE basic block; the
      // scheduling groups below make the compiler place two scalar / one vector instruction of it behind every MFMA.
      // (Measured and dropped: issuing the last four MFMAs of an M phase after the barrier that ends it, so that the wave
      // reaches the barrier earlier -- +2.6 % on one shape, -1.4 % on others, whole step 31.89 against 31.76 ms.)
      auto mfma_and_prepare = [&](int buf, int kt) {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_setprio(1);
        pp_prepare(buf, kt);
#pragma unroll
        for (int ks = 0; ks < KS; ++ks)
#pragma unroll
          for (int i = 0; i < MI; ++i)
#pragma unroll
            for (int j = 0; j < NI; ++j)
              acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(af[ks][i], bfr[ks][j], acc[i][j], 0, 0, 0);
#pragma unroll
        for (int q = 0; q < KS * MI * NI; ++q) {
          __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);   // one MFMA
          __builtin_amdgcn_sched_group_barrier(0x004, 2, 0);   // two SALU
          __builtin_amdgcn_sched_group_barrier(0x002, 1, 0);   // one VALU
        }
        pp_pin();
        __builtin_amdgcn_s_setprio(0);
        __builtin_amdgcn_sched_barrier(0);
      };
      // all of this wave's LDS-DMA has landed except the batch issued last (when one more tile was staged)
      auto wait_older_dma = [&](bool newer_in_flight) {
        if (LDM_ABLATE(9)) return;
        if (newer_in_flight) {
          if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT + B_FULL + 1) : "memory");
          else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT + B_FULL) : "memory");
        } else {
          asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
      };
      auto barrier = [&]() {
        __builtin_amdgcn_sched_barrier(0);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        __builtin_amdgcn_sched_barrier(0);
      };
      if constexpr (PP >= 2) {
        pp_prepare(0, kt_begin);
        pp_issue();
        if (nk > 1) { pp_prepare(1, kt_begin + 1); pp_issue(); }
        ln_finish();
        pp_prepare(2, kt_begin + 2);          // the state R(0) issues from (never issued when nk < 3)
        pp_pin();
      }
      wait_older_dma(nk > 1);                 // tile 0 (tile 1 may still be in flight)
      barrier();
      LDM_STAMP(2);                           // (diagnostic build) first tile landed for every wave
      if (wave < WM * WN / 2) {
        for (int kt = 0; kt < nk; ++kt) {
          if (kt == 1) LDM_STAMP(12);          // (diagnostic build) starts of K-tiles 1, 2, 3: what a short-K tile's loop is made of
          if (kt == 2) LDM_STAMP(13);
          if (kt == 3) LDM_STAMP(14);
          if (kt == 8) LDM_STAMP(8);
          read_frags(slot);
          const bool more = kt + 2 < nk;
          if constexpr (PP >= 2) { if (more) pp_issue(); }
          else { if (more) pp_stage(slot == 0 ? 2 : slot - 1, kt_begin + kt + 2); }
          // (kept although no barrier follows: with the compiler's counted waits alone -- MFMAs starting as their fragments
          // arrive -- the step measured 0.1 ms slower)
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          if (kt == 8) LDM_STAMP(9);
          if (kt == 8) LDM_STAMP(10);
          if constexpr (PP >= 2) mfma_and_prepare(slot, kt_begin + kt + 3);
          else mfma_all();
          wait_older_dma(more);
          if (kt == 8) LDM_STAMP(11);
          barrier();
          slot = slot == 2 ? 0 : slot + 1;
        }
      } else {
        for (int kt = 0; kt < nk; ++kt) {
          if (kt > 0) {
            if constexpr (PP >= 2) mfma_and_prepare(slot == 0 ? 2 : slot - 1, kt_begin + kt + 2);
            else mfma_all();
          }
          read_frags(slot);
          const bool more = kt + 2 < nk;
          if constexpr (PP >= 2) { if (more) pp_issue(); }
          else { if (more) pp_stage(slot == 0 ? 2 : slot - 1, kt_begin + kt + 2); }
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          wait_older_dma(more);
          barrier();
          slot = slot == 2 ? 0 : slot + 1;
        }
        mfma_all();
      }
    } else
    for (int kt = 0; kt < nk; ++kt) {
      if (kt + 1 < nk) {
        if (extra) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT + B_FULL + 1) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(A_IT + B_FULL) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      if (kt == 8) LDM_STAMP(8);
      __builtin_amdgcn_s_barrier();
      asm volatile("" ::: "memory");
      if (kt == 8) LDM_STAMP(9);
      const bool more = kt + 2 < nk;
      if (more) stage_begin(slot == 0 ? 2 : slot - 1, kt_begin + kt + 2);
      if (kt == 8) LDM_STAMP(10);
      compute_and_stage(slot, more);
      if (kt == 8) LDM_STAMP(11);
      slot = slot == 2 ? 0 : slot + 1;
    }
    __syncthreads();  // tile buffers become the epilogue's staging area
  }

  LDM_STAMP(3);
  // ---- LDM_EPI_LNFOLD: row statistics and the column sums of the scaled filter go to LDS behind the epilogue's staging area
  constexpr int EPI_CST_BYTES = WM * WN * 16 * (WTN + 4) * 4;
  constexpr int EPI_LN_BYTES = BM * 8;
  constexpr int EPI_RST_BYTES = WM * WN * 16 * (WTN / 8) * 8;
  static_assert(EPI_CST_BYTES + EPI_LN_BYTES + EPI_RST_BYTES <= STAGES * STAGE, "epilogue scratch does not fit the tile buffers");
  float* const lnst = reinterpret_cast<float*>(smem + EPI_CST_BYTES);     // [BM][2]: rstd, -mean * rstd
  const bool has_ln = p.ln_stats != nullptr;
  if (has_ln) {
    if (tid < BM) { lnst[2 * tid] = ln_r; lnst[2 * tid + 1] = -ln_mu * ln_r; }
    __syncthreads();
  }
  // ---- epilogue: C[row = fq*4 + r][col = fr] per 16x16 tile
  const bool split = p.splits > 1;           // raw fp32 partial sums, epilogue applied by splitk_reduce_kernel
  const int epi = split ? 0 : p.epilogue;
  const bool glu = (epi & (LDM_EPI_GEGLU | LDM_EPI_SWIGLU)) != 0;
  const int nout = glu ? p.N >> 1 : p.N;
  const bf16_t* resid = static_cast<const bf16_t*>(p.residual);
  const bf16_t* rowv = static_cast<const bf16_t*>(p.rowvec);
  void* const outp = split ? static_cast<void*>(p.partial + (size_t)blockIdx.y * p.M * p.N) : p.out;
  const bool out_f32 = split || p.out_f32;
  if (p.vec_epilogue) {
    // Vector epilogue: each wave transposes its accumulators through LDS, 16 rows at a time, so that a lane owns 8
    // consecutive output columns: bias / time-embedding / residual operands and the result move as 16-byte accesses
    // (the scalar form below issues one 2-byte store per accumulator register and is store-issue bound at small K).
    // The chunk geometry is compile-time, the bias is loaded once per lane, and the residual / time-embedding
    // operands of 16-row step i+1 are requested before step i is transposed, so their L2 latency (the bulk of the
    // epilogue when it was paid 3x per step) hides behind the LDS round trip and the stores of the step before.
    auto run = [&](auto glu_tag, auto preact_tag) {
      constexpr bool GLU = decltype(glu_tag)::value;
      constexpr bool PREACT = decltype(preact_tag)::value;   // compile-time: as a run-time branch inside the unrolled steps it cost the inference GLU launches 11 %
      constexpr int LDC = WTN + 4;                                // floats per staged row (+4: spreads the 4 row groups)
      static_assert(WM * WN * 16 * LDC * 4 <= STAGES * STAGE, "epilogue staging does not fit the tile buffers");
      constexpr int CPRW = GLU ? WTN / 16 : WTN / 8;              // 8-column output chunks per staged row
      constexpr int NCHUNK = 16 * CPRW;
      constexpr int NPASS = (NCHUNK + 63) / 64;
      float* cst = reinterpret_cast<float*>(smem) + wave * (16 * LDC);
      const int ncol0 = n0 + wn * WTN;
      const int mrow0 = m0 + wm * WTM;

      int row[NPASS], scol[NPASS], ncol[NPASS], ocol[NPASS];
      bool cok[NPASS];
      float bv[NPASS][8], bg[GLU ? NPASS : 1][8];
#pragma unroll
      for (int ps = 0; ps < NPASS; ++ps) {
        const int c = lane + 64 * ps;
        row[ps] = c / CPRW;
        const int cc = c - row[ps] * CPRW;
        scol[ps] = GLU ? (cc >> 1) * 32 + (cc & 1) * 8 : cc * 8; // GLU: value columns; gates sit 16 further
        ncol[ps] = ncol0 + scol[ps];
        ocol[ps] = GLU ? (ncol0 >> 1) + cc * 8 : ncol[ps];
        cok[ps] = c < NCHUNK && ncol[ps] < p.N;
#pragma unroll
        for (int e = 0; e < 8; ++e) { bv[ps][e] = 0.f; if (GLU) bg[ps][e] = 0.f; }
        if (cok[ps] && (epi & LDM_EPI_BIAS)) {
          load8(p.bias + ncol[ps], bv[ps]);
          if (GLU) load8(p.bias + ncol[ps] + 16, bg[ps]);
        }
      }
      // LDM_EPI_LNFOLD: y = rstd * acc + (-mean * rstd) * colsum(W')[n] + bias'[n].  colsum lives in registers like the bias (one
      // load per lane before the loop); the two per-row factors come from LDS once per 16-row step.
      float s1v[NPASS][8], s1g[GLU ? NPASS : 1][8];
      if (has_ln) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
#pragma unroll
          for (int e = 0; e < 8; ++e) { s1v[ps][e] = 0.f; if (GLU) s1g[ps][e] = 0.f; }
          if (cok[ps]) {
            load8(p.ln_s1 + ncol[ps], s1v[ps]);
            if (GLU) load8(p.ln_s1 + ncol[ps] + 16, s1g[ps]);
          }
        }
      }
      const bool has_res = !GLU && (epi & LDM_EPI_RESIDUAL), has_row = !GLU && (epi & LDM_EPI_ROWVEC);
      // LDM_EPI_ROWSTATS (bf16 out, N a multiple of the 80-column wave tile): per 16-row step every lane leaves the sum and
      // sum of squares of its 8 rounded outputs in LDS, lanes 0..15 fold the row's CPRW chunks in a fixed order and write
      // one (sum, sumsq) pair per row and 80-column segment: the LayerNorm statistics of the next GEMM's A operand.
      const bool want_stats = !GLU && p.rowstats != nullptr;
      float* const rst = reinterpret_cast<float*>(smem + EPI_CST_BYTES + EPI_LN_BYTES) + wave * (16 * (WTN / 8) * 2);
      const int rs_seg = ncol0 / WTN, rs_nseg = p.N / WTN;
      uint4 rres[NPASS], rrow[NPASS];
      auto fetch = [&](int i, uint4 (&qres)[NPASS], uint4 (&qrow)[NPASS]) {
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int m = mrow0 + i * 16 + row[ps];
          qres[ps] = make_uint4(0, 0, 0, 0);
          qrow[ps] = make_uint4(0, 0, 0, 0);
          if (cok[ps] && m < p.M) {
            if (has_res) qres[ps] = *reinterpret_cast<const uint4*>(resid + (size_t)m * p.N + ncol[ps]);
            if (has_row) qrow[ps] = *reinterpret_cast<const uint4*>(rowv + (size_t)sample_of(m) * p.rowvec_stride + ncol[ps]);
          }
        }
      };
      auto add_bf16x8 = [](float (&v)[8], const uint4& q) {
        const uint32_t w[4] = {q.x, q.y, q.z, q.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
          v[2 * e] += __uint_as_float(w[e] << 16);
          v[2 * e + 1] += __uint_as_float(w[e] & 0xffff0000u);
        }
      };
      if (has_res || has_row) fetch(0, rres, rrow);
#pragma unroll
      for (int i = 0; i < MI; ++i) {
#pragma unroll
        for (int j = 0; j < NI; ++j)
#pragma unroll
          for (int r = 0; r < 4; ++r) cst[(fq * 4 + r) * LDC + j * 16 + fr] = acc[i][j][r];
        uint4 nres[NPASS], nrow[NPASS];
        if ((has_res || has_row) && i + 1 < MI) fetch(i + 1, nres, nrow);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
#pragma unroll
        for (int ps = 0; ps < NPASS; ++ps) {
          const int m = mrow0 + i * 16 + row[ps];
          if (!cok[ps] || m >= p.M) continue;
          float v[8];
          const float* src = cst + row[ps] * LDC + scol[ps];
          const float4 x0 = *reinterpret_cast<const float4*>(src);
          const float4 x1 = *reinterpret_cast<const float4*>(src + 4);
          v[0] = x0.x; v[1] = x0.y; v[2] = x0.z; v[3] = x0.w; v[4] = x1.x; v[5] = x1.y; v[6] = x1.z; v[7] = x1.w;
          float rr = 1.f, nm = 0.f;
          if (has_ln) {   // y = rstd * (x W'^T - mean * colsum(W')) + (bias + W beta): LayerNorm folded into this GEMM
            const float2 st = *reinterpret_cast<const float2*>(lnst + 2 * (wm * WTM + i * 16 + row[ps]));
            rr = st.x; nm = st.y;
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] = fmaf(rr, v[e], fmaf(nm, s1v[ps][e], bv[ps][e]));
          } else {
#pragma unroll
            for (int e = 0; e < 8; ++e) v[e] += bv[ps][e];
          }
          if constexpr (GLU) {
            const float4 g0 = *reinterpret_cast<const float4*>(src + 16);
            const float4 g1 = *reinterpret_cast<const float4*>(src + 20);
            float g[8] = {g0.x, g0.y, g0.z, g0.w, g1.x, g1.y, g1.z, g1.w};
            if (has_ln) {
#pragma unroll
              for (int e = 0; e < 8; ++e) g[e] = fmaf(rr, g[e], fmaf(nm, s1g[ps][e], bg[ps][e]));
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) g[e] += bg[ps][e];
            }
            if constexpr (PREACT) {   // what the backward differentiates: value and gate blocks as the plain GEMM lays them out
              // (non-temporal: nothing reads it before the backward pass)
              typedef __attribute__((ext_vector_type(4))) unsigned int v4u_t;
              bf16_t* pa = static_cast<bf16_t*>(p.glu_preact) + (size_t)m * p.N + ncol[ps];
              const v4u_t rv = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
              const v4u_t rg = {pack_bf16x2(g[0], g[1]), pack_bf16x2(g[2], g[3]), pack_bf16x2(g[4], g[5]), pack_bf16x2(g[6], g[7])};
              __builtin_nontemporal_store(rv, reinterpret_cast<v4u_t*>(pa));
              __builtin_nontemporal_store(rg, reinterpret_cast<v4u_t*>(pa + 16));
            }
            if (epi & LDM_EPI_GEGLU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= gelu_erf_fast(g[e]);
            } else {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] *= silu_fast(g[e]);
            }
          } else {
            if (has_row) add_bf16x8(v, rrow[ps]);
            if (epi & LDM_EPI_SILU) {
#pragma unroll
              for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
            }
            if (has_res) add_bf16x8(v, rres[ps]);
          }
          if (out_f32) store8(static_cast<float*>(outp) + (size_t)m * nout + ocol[ps], v);
          else {
            typedef __attribute__((ext_vector_type(4))) unsigned int v4u_t;
            const v4u_t r = {pack_bf16x2(v[0], v[1]), pack_bf16x2(v[2], v[3]), pack_bf16x2(v[4], v[5]), pack_bf16x2(v[6], v[7])};
            v4u_t* const dst = reinterpret_cast<v4u_t*>(static_cast<bf16_t*>(outp) + (size_t)m * nout + ocol[ps]);
            if (p.nt_out) __builtin_nontemporal_store(r, dst);      // only the store instruction differs
            else *dst = r;
          }
          if (want_stats) {
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int e = 0; e < 8; e += 2) {       // statistics of the values as stored (rounded to the output dtype)
              float a = v[e], b = v[e + 1];
              if (!out_f32) {
                const uint32_t w = pack_bf16x2(a, b);
                a = __uint_as_float(w << 16);
                b = __uint_as_float(w & 0xffff0000u);
              }
              sx += a + b;
              sq += a * a + b * b;
            }
            const int c = lane + 64 * ps;            // chunk index inside the 16 x WTN step: row * CPRW + chunk of the row
            *reinterpret_cast<float2*>(rst + 2 * c) = make_float2(sx, sq);
          }
        }
        if (want_stats) {
          asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
          const int m = mrow0 + i * 16 + lane;
          if (lane < 16 && m < p.M) {
            float sx = 0.f, sq = 0.f;
#pragma unroll
            for (int cidx = 0; cidx < CPRW; ++cidx) {
              const float2 t = *reinterpret_cast<const float2*>(rst + 2 * (lane * CPRW + cidx));
              sx += t.x;
              sq += t.y;
            }
            *reinterpret_cast<float2*>(p.rowstats + ((size_t)m * rs_nseg + rs_seg) * 2) = make_float2(sx, sq);
          }
        }
        if (i + 1 < MI) {
#pragma unroll
          for (int ps = 0; ps < NPASS; ++ps) { rres[ps] = nres[ps]; rrow[ps] = nrow[ps]; }
        }
        LDM_LOCKSTEP();   // the next step overwrites the staging rows other lanes have just read
      }
    };
    if (glu) {
      if constexpr (NI % 2 == 0) {
        if (p.glu_preact != nullptr) run(std::true_type{}, std::true_type{});
        else run(std::true_type{}, std::false_type{});
      }
    } else {
      run(std::false_type{}, std::false_type{});
    }
    LDM_STAMP(4);
    LDM_STAMP(6);
    return;
  }
#pragma unroll
  for (int i = 0; i < MI; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * WTM + i * 16 + fq * 4 + r;
      if (m >= p.M) continue;
      const int sample = (epi & LDM_EPI_ROWVEC) ? m / hw_out : 0;
      if (!glu) {
#pragma unroll
        for (int j = 0; j < NI; ++j) {
          const int n = n0 + wn * WTN + j * 16 + fr;
          if (n >= p.N) continue;
          float v = acc[i][j][r];
          if (epi & LDM_EPI_BIAS) v += p.bias[n];
          if (epi & LDM_EPI_ROWVEC) v += bf16_to_f32(rowv[(size_t)sample * p.rowvec_stride + n]);
          if (epi & LDM_EPI_SILU) v = silu_f(v);
          if (epi & LDM_EPI_RESIDUAL) v += bf16_to_f32(resid[(size_t)m * p.N + n]);
          if (out_f32) static_cast<float*>(outp)[(size_t)m * p.N + n] = v;
          else static_cast<bf16_t*>(outp)[(size_t)m * p.N + n] = f32_to_bf16(v);
        }
      } else {
        if constexpr (NI % 2 == 0) {
#pragma unroll
          for (int j = 0; j < NI; j += 2) {
            const int n = n0 + wn * WTN + j * 16 + fr;       // packed column of the value
            if (n + 16 >= p.N + 0 && n >= p.N) continue;
            if (n >= p.N) continue;
            float val = acc[i][j][r], gate = acc[i][j + 1][r];
            if (epi & LDM_EPI_BIAS) { val += p.bias[n]; gate += p.bias[n + 16]; }
            const float g = (epi & LDM_EPI_GEGLU) ? gelu_erf_f(gate) : silu_f(gate);
            const int jo = ((n0 + wn * WTN + j * 16) >> 1) + fr;  // column in the [M][N/2] output
            const float v = val * g;
            if (out_f32) static_cast<float*>(outp)[(size_t)m * nout + jo] = v;
            else static_cast<bf16_t*>(outp)[(size_t)m * nout + jo] = f32_to_bf16(v);
          }
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------- split-K fold
// out[m][n..n+8) = epilogue( sum_z partial[z][m][n..n+8) ), fixed z order (deterministic).  One lane per 8 columns.
__global__ void __launch_bounds__(256)
splitk_reduce_kernel(const ConvGemmParams p) {
  const int cpr = p.N >> 3;
  const long long idx = (long long)blockIdx.x * 256 + threadIdx.x;
  if (idx >= (long long)p.M * cpr) return;
  const int m = (int)(idx / cpr), n = (int)(idx - (long long)m * cpr) * 8;
  float v[8] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
  for (int z = 0; z < p.splits; ++z) {
    float t[8];
    load8(p.partial + ((size_t)z * p.M + m) * p.N + n, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
  const int epi = p.epilogue;
  float t[8];
  if (epi & LDM_EPI_BIAS) {
    load8(p.bias + n, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
  if (epi & LDM_EPI_ROWVEC) {
    load8(static_cast<const bf16_t*>(p.rowvec) + (size_t)(m / (p.hout * p.wout)) * p.rowvec_stride + n, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
  if (epi & LDM_EPI_SILU) {
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] = silu_fast(v[e]);
  }
  if (epi & LDM_EPI_RESIDUAL) {
    load8(static_cast<const bf16_t*>(p.residual) + (size_t)m * p.N + n, t);
#pragma unroll
    for (int e = 0; e < 8; ++e) v[e] += t[e];
  }
  if (p.out_f32) store8(static_cast<float*>(p.out) + (size_t)m * p.N + n, v);
  else store8(static_cast<bf16_t*>(p.out) + (size_t)m * p.N + n, v);
}

// ------------------------------------------------------------------------------------------- fp32
// Exact-fp32 implicit GEMM (v_mfma_f32_16x16x4_f32: bit-for-bit a k-ordered fmaf chain).  Used for the fp32
// parity gate, not for speed: register-staged 64x64x16 tiles, 4 waves, each a 32x32 quadrant.
template <int BM, int BN, int BK>
__global__ void __launch_bounds__(256)
conv_gemm_f32_kernel(const ConvGemmParams p) {
  constexpr int LDA = BK + 4;   // +4 floats: conflict-free column reads
  __shared__ float sa[BM * LDA];
  __shared__ float sb[BN * LDA];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const int wm = wave >> 1, wn = wave & 1;
  const int tile = blockIdx.x;
  int tm, tn;
  tile_to_mn(tile, p.tiles_m, p.tiles_n, tm, tn);
  const int m0 = tm * BM, n0 = tn * BN;
  const float* a0 = static_cast<const float*>(p.a0);
  const float* a1 = static_cast<const float*>(p.a1);
  const float* wq = static_cast<const float*>(p.w);
  const int hw_out = p.hout * p.wout;
  const int hlim = p.upsample ? p.hin * 2 : p.hin;
  const int wlim = p.upsample ? p.win * 2 : p.win;
  const int pad = p.ksize >> 1;

  // staging: BM x BK floats = BM*BK/4 float4 chunks; 256 threads -> (BM*BK/4)/256 per thread
  constexpr int CPR = BK / 4;                       // float4 chunks per row
  constexpr int A_IT = BM * CPR / 256, B_IT = BN * CPR / 256;
  static_assert(BM * CPR % 256 == 0 && BN * CPR % 256 == 0, "fp32 tile/threads mismatch");
  int a_pix[A_IT], a_ih[A_IT], a_iw[A_IT];
#pragma unroll
  for (int it = 0; it < A_IT; ++it) {
    const int r = (it * 256 + tid) / CPR;
    const int m = m0 + r;
    if (m < p.M) {
      const int b = m / hw_out, rem = m - b * hw_out;
      const int oh = rem / p.wout, ow = rem - oh * p.wout;
      a_pix[it] = b * p.hin * p.win;
      a_ih[it] = oh * p.stride - pad;
      a_iw[it] = ow * p.stride - pad;
    } else {
      a_pix[it] = -1; a_ih[it] = a_iw[it] = 0;
    }
  }
  const int cq = (tid % CPR) * 4;

  f32x4_t acc[2][2];
#pragma unroll
  for (int i = 0; i < 2; ++i)
#pragma unroll
    for (int j = 0; j < 2; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  const int fr = lane & 15, fq = lane >> 4;
  const int nk = p.K / BK;
  for (int kt = 0; kt < nk; ++kt) {
    const int k0 = kt * BK;
    const int tap = k0 / p.ctot;
    const int c = k0 - tap * p.ctot;
    const int kh = tap / p.ksize, kw = tap - kh * p.ksize;
    const bool second = c >= p.c0;
    const float* src = second ? a1 : a0;
    const int cs = second ? p.c1 : p.c0;
    const int cc = second ? c - p.c0 : c;
    float4 va[A_IT], vb[B_IT];
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      int ih = a_ih[it] + kh, iw = a_iw[it] + kw;
      bool ok = a_pix[it] >= 0 && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
      if (p.upsample == 2) ok = ok && ((ih | iw) & 1) == 0;
      if (p.upsample) { ih >>= 1; iw >>= 1; }
      va[it] = ok ? *reinterpret_cast<const float4*>(src + ((size_t)(a_pix[it] + ih * p.win + iw) * cs + cc + cq))
                  : make_float4(0.f, 0.f, 0.f, 0.f);
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int r = (it * 256 + tid) / CPR;
      const int n = n0 + r;
      vb[it] = n < p.N ? *reinterpret_cast<const float4*>(wq + (size_t)n * p.K + k0 + cq) : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();  // previous tile fully consumed
#pragma unroll
    for (int it = 0; it < A_IT; ++it) {
      const int r = (it * 256 + tid) / CPR;
      *reinterpret_cast<float4*>(&sa[r * LDA + cq]) = va[it];
    }
#pragma unroll
    for (int it = 0; it < B_IT; ++it) {
      const int r = (it * 256 + tid) / CPR;
      *reinterpret_cast<float4*>(&sb[r * LDA + cq]) = vb[it];
    }
    __syncthreads();
#pragma unroll
    for (int ks = 0; ks < BK / 4; ++ks) {
      float af[2], bf[2];
#pragma unroll
      for (int i = 0; i < 2; ++i) af[i] = sa[(wm * 32 + i * 16 + fr) * LDA + ks * 4 + fq];
#pragma unroll
      for (int j = 0; j < 2; ++j) bf[j] = sb[(wn * 32 + j * 16 + fr) * LDA + ks * 4 + fq];
#pragma unroll
      for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
          acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x4f32(af[i], bf[j], acc[i][j], 0, 0, 0);
    }
  }

  const int epi = p.epilogue;
  const bool glu = (epi & (LDM_EPI_GEGLU | LDM_EPI_SWIGLU)) != 0;
  const int nout = glu ? p.N >> 1 : p.N;
  const float* resid = static_cast<const float*>(p.residual);
  const float* rowv = static_cast<const float*>(p.rowvec);
  float* out = static_cast<float*>(p.out);
#pragma unroll
  for (int i = 0; i < 2; ++i) {
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int m = m0 + wm * 32 + i * 16 + fq * 4 + r;
      if (m >= p.M) continue;
      const int sample = (epi & LDM_EPI_ROWVEC) ? m / hw_out : 0;
      if (!glu) {
#pragma unroll
        for (int j = 0; j < 2; ++j) {
          const int n = n0 + wn * 32 + j * 16 + fr;
          if (n >= p.N) continue;
          float v = acc[i][j][r];
          if (epi & LDM_EPI_BIAS) v += p.bias[n];
          if (epi & LDM_EPI_ROWVEC) v += rowv[(size_t)sample * p.rowvec_stride + n];
          if (epi & LDM_EPI_SILU) v = v / (1.0f + expf(-v));
          if (epi & LDM_EPI_RESIDUAL) v += resid[(size_t)m * p.N + n];
          out[(size_t)m * p.N + n] = v;
        }
      } else {
        const int n = n0 + wn * 32 + fr;
        if (n < p.N) {
          float val = acc[i][0][r], gate = acc[i][1][r];
          if (epi & LDM_EPI_BIAS) { val += p.bias[n]; gate += p.bias[n + 16]; }
          const float g = (epi & LDM_EPI_GEGLU) ? 0.5f * gate * (1.0f + erff(gate * 0.70710678118654752440f))
                                                : gate / (1.0f + expf(-gate));
          out[(size_t)m * nout + ((n0 + wn * 32) >> 1) + fr] = val * g;
        }
      }
    }
  }
}

// ------------------------------------------------------------------------------------------ launch
#ifdef LDM_GEMM_STAMPS
static unsigned long long* g_stamp_buf = nullptr;
static long long g_stamp_blocks = 0;
#endif
template <int BM, int BN, int BK, int WM, int WN, int STAGES = 2, int PP = 0>
static int launch_bf16(ConvGemmParams& p, hipStream_t st) {
  p.tiles_m = (p.M + BM - 1) / BM;
  p.tiles_n = (p.N + BN - 1) / BN;
  const int smem = STAGES * (BM + BN) * BK * 2;
  auto kern = conv_gemm_bf16_kernel<BM, BN, BK, WM, WN, STAGES, PP>;
  static bool attr_done = false;  // opt in to > 64 KiB of dynamic LDS once per instantiation
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { set_error("ldm_conv_gemm: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
    attr_done = true;
  }
#ifdef LDM_GEMM_STAMPS
  p.stamps = (long long)p.tiles_m * p.tiles_n <= g_stamp_blocks && p.splits == 1 ? g_stamp_buf : nullptr;
#endif
  hipLaunchKernelGGL(kern, dim3(p.tiles_m * p.tiles_n, p.splits), dim3(64 * WM * WN), smem, st, p);
  LDM_HIP_CHECK_LAUNCH("ldm_conv_gemm(bf16)");
  if (p.splits > 1) {
    const long long chunks = (long long)p.M * (p.N / 8);
    hipLaunchKernelGGL(splitk_reduce_kernel, dim3((unsigned)((chunks + 255) / 256)), dim3(256), 0, st, p);
    LDM_HIP_CHECK_LAUNCH("ldm_conv_gemm(split-K fold)");
  }
  return LDM_OK;
}

// Which form of the 3-slot ring the 256-row tiles run: 2 = ping-pong with the staging arithmetic under the MFMAs (default),
// 1 = ping-pong with the ring's own staging code (the layers form 2 does not cover: folded upsampling), 0 = the plain ring.
// LDM_GEMM_PINGPONG overrides (read per call: A/B runs in one process, tools/gemm_bench.py --pp).
static int pingpong(const ConvGemmParams& p) {
  const char* e = getenv("LDM_GEMM_PINGPONG");
  const int v = e ? (atoi(e) > 2 ? 2 : atoi(e)) : 2;
  if (v >= 2 && p.upsample) return 1;   // form 2 keeps one tap shift per K-tile: not linear under the folded upsample
  return v;
}

// Expected efficiency of a (tile, split-K) choice: padding waste x chip fill x operand reuse x split-K overhead.
static double tile_score(const ConvGemmParams& p, int BM, int BN, int blocks_per_cu, int splits) {
  const double tm = (p.M + BM - 1) / BM, tn = (p.N + BN - 1) / BN;
  const double tiles = tm * tn;
  const double useful = ((double)p.M * p.N) / (tiles * BM * BN);          // padding waste
  const double slots = 256.0 * blocks_per_cu;
  const double blocks = tiles * splits;
  const double rounds = ceil(blocks / slots);
  const double fill = blocks / (rounds * slots);                           // tail / under-fill
  // operand reuse: 256x160 (8 waves, 3-slot ring, DMA issue spread between the MFMAs) measured 2-9 % ahead of 128x160
  // on the large layers and 256x128 behind 128x128 (profiles/r1_gemm_tile_sweep.txt); 64-row tiles stage 40 % more
  // bytes per FLOP.
  const double size_bonus = BM >= 256 ? (BN == 160 ? 1.04 : 0.97) : ((BM * BN >= 128 * 128) ? 1.0 : 0.85);
  // split-K: every split adds an fp32 round trip of the output through the fold kernel
  const double split_cost = 1.0 - 0.04 * (splits - 1);
  return useful * fill * size_bonus * split_cost;
}

// Tile choice of the bf16 path: BK = 64 when no K-tile straddles a tap or a concat boundary, else 32;
// among the BK = 64 tiles (and, when the caller lent a workspace, split-K factors) the best score; GLU epilogues
// need an even number of 16-column tiles per wave, i.e. BN = 128, and are never split.
static void pick_tile_bf16(const ConvGemmParams& p, bool glu, long long workspace_bytes, int& bm, int& bn, int& bk, int& splits) {
  splits = 1;
  const bool need160 = p.rowstats != nullptr;          // row statistics are written per 80-column wave tile: BN = 160 tiles only
  bool bk64 = p.ctot % 64 == 0 && p.c0 % 64 == 0;
  // GLU projections with five K-tiles or fewer (K <= 320: the 64x64 level's feed-forward): the tile's life is set-up, first
  // fill and a GELU-heavy epilogue, not its loop, so what pays is more workgroups per CU to overlap them: 128x128x32 fits
  // three (48 KB of LDS each on the 3-slot ring, 168 registers) where 128x128x64 fits two: -6 % per launch, -0.2 ms per step
  // (profiles/r3_glu_short_k_three_workgroups.txt), the 3-slot ring another -5 % / -0.15 ms.  Equal at K = 640, slower at K = 1280 and for the plain K = 320 layers.  LDM_GEMM_GLU_BK32=0: off (A/B).
  if (glu && p.K <= 320) {
    const char* e = getenv("LDM_GEMM_GLU_BK32");
    if (e == nullptr || atoi(e) != 0) bk64 = false;
  }
  if (!bk64) {
    bk = 32;
    if (p.M <= 64 * 256 && !glu) { bm = 64; bn = 64; } else { bm = 128; bn = 128; }
    return;
  }
  bk = 64;
  if (const char* force = getenv("LDM_GEMM_TILE")) {   // tuning knob (tools/gemm_bench.py): "BM,BN[,splits]" of a BK=64 tile
    int fm = 0, fn = 0, fs = 1;
    const int got = sscanf(force, "%d,%d,%d", &fm, &fn, &fs);
    if (got >= 2 && (fm == 64 || fm == 128 || fm == 256) && ((fn == 128 && !need160) || (fn == 160 && !glu))) {
      bm = fm; bn = fn;
      if (got == 3 && fs >= 1 && fs <= 16 && !glu && (long long)fs * p.M * p.N * 4 <= workspace_bytes && p.K / 64 >= fs) splits = fs;
      return;
    }
  }
  static const int cand[][3] = {{128, 128, 2}, {128, 160, 2}, {64, 128, 3}, {64, 160, 3}, {256, 128, 1}, {256, 160, 1}};
  static const int split_cand[] = {1, 2, 3, 4, 6, 8};
  const int nk = p.K / 64;
  double best = -1.0;
  for (const auto& c : cand) {
    if (glu && c[1] != 128) continue;
    if (need160 && c[1] != 160) continue;
    for (int sp : split_cand) {
      if (sp > 1 && (glu || nk / sp < 16 || (long long)sp * p.M * p.N * 4 > workspace_bytes)) continue;
      const double sc = tile_score(p, c[0], c[1], c[2], sp);
      if (sc > best) { best = sc; bm = c[0]; bn = c[1]; splits = sp; }
    }
  }
}

}  // namespace ldm

using namespace ldm;

#ifdef LDM_GEMM_STAMPS
extern "C" int ldm_debug_set_stamps(unsigned long long* buf, long long blocks) {
  g_stamp_buf = buf;
  g_stamp_blocks = blocks;
  return LDM_OK;
}
#endif

extern "C" int ldm_conv_gemm(const LdmConvGemm* d, void* stream) {
  LDM_REQUIRE(d != nullptr, "ldm_conv_gemm: null descriptor");
  LDM_REQUIRE(dtype_ok(d->dtype) && dtype_ok(d->out_dtype), "ldm_conv_gemm: bad dtype");
  LDM_REQUIRE(d->out_dtype == d->dtype || d->out_dtype == LDM_F32, "ldm_conv_gemm: out_dtype must equal dtype or be float32");
  LDM_REQUIRE(d->ksize == 1 || d->ksize == 3, "ldm_conv_gemm: ksize must be 1 or 3 (got %d)", d->ksize);
  LDM_REQUIRE(d->stride == 1 || d->stride == 2, "ldm_conv_gemm: stride must be 1 or 2 (got %d)", d->stride);
  LDM_REQUIRE(d->upsample >= 0 && d->upsample <= 2, "ldm_conv_gemm: upsample must be 0, 1 (nearest) or 2 (zero insertion)");
  LDM_REQUIRE(!(d->upsample && d->stride != 1), "ldm_conv_gemm: upsample with stride 2 is not a thing");
  LDM_REQUIRE(d->batch >= 0 && d->hin >= 1 && d->win >= 1 && d->hout >= 1 && d->wout >= 1 && d->n >= 1 && d->c0 >= 1 && d->c1 >= 0,
              "ldm_conv_gemm: bad sizes");
  const int pad = d->ksize / 2;
  const int hsrc = d->upsample ? 2 * d->hin : d->hin, wsrc = d->upsample ? 2 * d->win : d->win;
  {
    const int he = (hsrc + 2 * pad - d->ksize) / d->stride + 1, we = (wsrc + 2 * pad - d->ksize) / d->stride + 1;
    // zero insertion is the input gradient of a stride-2 conv, whose input may have had an odd extent (2*hin - 1)
    const bool ok = d->upsample == 2 ? ((d->hout == he || d->hout == he - 1) && (d->wout == we || d->wout == we - 1))
                                     : (d->hout == he && d->wout == we);
    LDM_REQUIRE(ok, "ldm_conv_gemm: output extent %dx%d does not match input %dx%d (k=%d, stride=%d, upsample=%d)",
                d->hout, d->wout, d->hin, d->win, d->ksize, d->stride, d->upsample);
  }
  const long long M64 = (long long)d->batch * d->hout * d->wout;
  LDM_REQUIRE(M64 < (1LL << 31) && (long long)d->batch * d->hin * d->win < (1LL << 31), "ldm_conv_gemm: too many pixels");
  if (M64 == 0) return LDM_OK;
  LDM_REQUIRE(d->a0 && d->w && d->out && (d->c1 == 0 || d->a1), "ldm_conv_gemm: null pointer");
  const int epi = d->epilogue;
  LDM_REQUIRE(!(epi & LDM_EPI_BIAS) || d->bias, "ldm_conv_gemm: LDM_EPI_BIAS without bias");
  LDM_REQUIRE(!(epi & LDM_EPI_ROWVEC) || (d->rowvec && d->rowvec_stride >= d->n), "ldm_conv_gemm: LDM_EPI_ROWVEC without rowvec");
  LDM_REQUIRE(!(epi & LDM_EPI_RESIDUAL) || d->residual, "ldm_conv_gemm: LDM_EPI_RESIDUAL without residual");
  const bool glu = (epi & (LDM_EPI_GEGLU | LDM_EPI_SWIGLU)) != 0;
  LDM_REQUIRE(!glu || (d->n % 32 == 0 && !(epi & (LDM_EPI_ROWVEC | LDM_EPI_RESIDUAL | LDM_EPI_SILU))),
              "ldm_conv_gemm: GLU epilogue needs n %% 32 == 0 and no rowvec/residual/silu");
  LDM_REQUIRE(!((epi & LDM_EPI_GEGLU) && (epi & LDM_EPI_SWIGLU)), "ldm_conv_gemm: GEGLU and SWIGLU are exclusive");

  ConvGemmParams p;
  p.a0 = d->a0; p.a1 = d->c1 ? d->a1 : d->a0;
  p.c0 = d->c0; p.c1 = d->c1; p.ctot = d->c0 + d->c1;
  p.batch = d->batch; p.hin = d->hin; p.win = d->win; p.hout = d->hout; p.wout = d->wout;
  p.ksize = d->ksize; p.stride = d->stride; p.upsample = d->upsample;
  p.w = d->w; p.M = (int)M64; p.N = d->n; p.K = d->ksize * d->ksize * p.ctot;
  p.epilogue = epi; p.bias = d->bias; p.rowvec = d->rowvec; p.rowvec_stride = d->rowvec_stride;
  p.residual = d->residual; p.out = d->out;
  p.out_f32 = d->out_dtype == LDM_F32;
  p.vec_epilogue = 0;
  p.splits = 1;
  p.partial = nullptr;
  p.rowstats = (epi & LDM_EPI_ROWSTATS) ? d->rowstats : nullptr;
  p.ln_stats = (epi & LDM_EPI_LNFOLD) ? d->ln_stats : nullptr;
  p.ln_nseg = d->ln_segments;
  p.ln_s1 = d->ln_colsum;
  p.ln_eps = d->ln_eps;
  p.glu_preact = d->glu_preact;
  {
    const char* e = getenv("LDM_GEMM_NT_MB");             // read per call (tests toggle it): 0 = never, 1 = nearly always
    const long long nt_bytes = (e ? atoll(e) : 128LL) << 20;
    const long long out_bytes = (long long)d->batch * d->hout * d->wout * (long long)d->n * (glu ? 1 : 2);
    p.nt_out = nt_bytes > 0 && d->out_dtype == LDM_BF16 && out_bytes >= nt_bytes;
  }
#ifdef LDM_GEMM_STAMPS
  p.stamps = nullptr;
  {
    static const int abl = [] { const char* e = getenv("LDM_GEMM_ABLATE"); return e ? atoi(e) : 0; }();
    p.ablate = abl;
  }
#endif
  hipStream_t st = (hipStream_t)stream;
  if (epi & (LDM_EPI_ROWSTATS | LDM_EPI_LNFOLD)) {
    LDM_REQUIRE(d->dtype == LDM_BF16, "ldm_conv_gemm: LDM_EPI_ROWSTATS / LDM_EPI_LNFOLD are bf16-only");
    if (epi & LDM_EPI_ROWSTATS)
      LDM_REQUIRE(d->rowstats && !glu && d->n % 160 == 0 && (reinterpret_cast<uintptr_t>(d->rowstats) & 7) == 0,
                  "ldm_conv_gemm: LDM_EPI_ROWSTATS needs a rowstats buffer, no GLU and n %% 160 == 0 (n=%d)", d->n);
    if (epi & LDM_EPI_LNFOLD)
      LDM_REQUIRE(d->ln_stats && d->ln_colsum && d->ln_segments >= 1 && d->ln_segments <= 16 && d->ksize == 1 && d->c1 == 0 &&
                      d->stride == 1 && !d->upsample && d->ln_eps > 0.f && (reinterpret_cast<uintptr_t>(d->ln_stats) & 7) == 0,
                  "ldm_conv_gemm: LDM_EPI_LNFOLD needs ln_stats / ln_colsum / ln_segments and a plain (1x1, single-source) layer");
    // the statistics are per 80-column segment of the tensor this call reads as A: any other count would read past the
    // statistics buffer (or normalise with another layer's sums)
    if (epi & LDM_EPI_LNFOLD)
      LDM_REQUIRE(d->ln_segments * 80 == d->c0, "ldm_conv_gemm: LDM_EPI_LNFOLD: %d segments of 80 columns do not cover c0 = %d",
                  d->ln_segments, d->c0);
  }

  if (d->dtype == LDM_F32) {
    LDM_REQUIRE(d->glu_preact == nullptr, "ldm_conv_gemm: glu_preact is bf16-only");
    LDM_REQUIRE(p.ctot % 16 == 0 && p.c0 % 16 == 0, "ldm_conv_gemm(f32): channel counts must be multiples of 16 (c0=%d c1=%d)", d->c0, d->c1);
    LDM_REQUIRE(aligned16(d->a0) && aligned16(p.a1) && aligned16(d->w), "ldm_conv_gemm: operands must be 16-byte aligned");
    p.tiles_m = (p.M + 63) / 64;
    p.tiles_n = (p.N + 63) / 64;
    LDM_REQUIRE(!glu || true, "");
    hipLaunchKernelGGL((conv_gemm_f32_kernel<64, 64, 16>), dim3(p.tiles_m * p.tiles_n), dim3(256), 0, st, p);
    LDM_HIP_CHECK_LAUNCH("ldm_conv_gemm(f32)");
    return LDM_OK;
  }

  LDM_REQUIRE(p.ctot % 32 == 0 && p.c0 % 32 == 0, "ldm_conv_gemm(bf16): channel counts must be multiples of 32 (c0=%d c1=%d)", d->c0, d->c1);
  {
    const long long pix = (long long)d->batch * d->hin * d->win;
    LDM_REQUIRE(pix * d->c0 * 2 < (1LL << 31) && pix * d->c1 * 2 < (1LL << 31) && (long long)p.N * p.K * 2 < (1LL << 31),
                "ldm_conv_gemm(bf16): an operand exceeds 2 GiB (32-bit buffer offsets)");
  }
  LDM_REQUIRE(aligned16(d->a0) && aligned16(p.a1) && aligned16(d->w), "ldm_conv_gemm: operands must be 16-byte aligned");
  {
    const int nout = glu ? p.N / 2 : p.N;
    const size_t osz = p.out_f32 ? 4 : 2;
    bool ok = p.N % 8 == 0 && nout % 8 == 0 && aligned16(p.out) && ((size_t)nout * osz) % 16 == 0;
    if (epi & LDM_EPI_BIAS) ok = ok && aligned16(p.bias);
    if (epi & LDM_EPI_ROWVEC) ok = ok && aligned16(p.rowvec) && p.rowvec_stride % 8 == 0;
    if (epi & LDM_EPI_RESIDUAL) ok = ok && aligned16(p.residual);
    if (glu) ok = ok && p.N % 32 == 0;
    p.vec_epilogue = ok ? 1 : 0;
  }
  LDM_REQUIRE(d->glu_preact == nullptr || (glu && p.vec_epilogue && d->out_dtype == LDM_BF16 && aligned16(d->glu_preact)),
              "ldm_conv_gemm: glu_preact needs a GLU epilogue, bf16 output, 16-byte aligned operands and n %% 32 == 0");
  LDM_REQUIRE(p.vec_epilogue || !(epi & (LDM_EPI_ROWSTATS | LDM_EPI_LNFOLD)),
              "ldm_conv_gemm: LDM_EPI_ROWSTATS / LDM_EPI_LNFOLD need 16-byte aligned operands and n %% 8 == 0");
  int bm, bn, bk;
  const bool ws_ok = d->workspace != nullptr && aligned16(d->workspace) && p.vec_epilogue &&
                     !(epi & (LDM_EPI_ROWSTATS | LDM_EPI_LNFOLD));       // the split-K fold kernel knows neither
  pick_tile_bf16(p, glu, ws_ok ? d->workspace_bytes : 0, bm, bn, bk, p.splits);
  p.partial = static_cast<float*>(d->workspace);
  // row statistics are laid out per 80-column wave tile of a BN = 160 tile (rowstats[m][n / 80][2]): any other tile would
  // write a different segment count than the caller allocated
  LDM_REQUIRE(!(epi & LDM_EPI_ROWSTATS) || (bn == 160 && bk == 64),
              "ldm_conv_gemm: LDM_EPI_ROWSTATS needs channel counts in multiples of 64 and n %% 160 == 0 (tile %dx%dx%d)", bm, bn, bk);
  const int key = bm * 1000000 + bn * 1000 + bk;
  switch (key) {
    case 64064032: return launch_bf16<64, 64, 32, 2, 2>(p, st);
    case 128128032: {
      // the short-K GLU tile on the 3-slot ring (48 KB of LDS: still three workgroups per CU, and the LDS-DMA of the tile after
      // next stays in flight across the ONE barrier per K-tile): -5 % on the 64x64 level's feed-forward (392 -> 373 us).
      // The same ring under 128x128x64 costs the second workgroup per CU: 286 -> 430 us.  LDM_GEMM_128_RING3=0: two slots (A/B).
      const char* e = getenv("LDM_GEMM_128_RING3");
      return (e == nullptr || atoi(e) != 0) ? launch_bf16<128, 128, 32, 2, 2, 3>(p, st) : launch_bf16<128, 128, 32, 2, 2>(p, st);
    }
    case 128128064: return launch_bf16<128, 128, 64, 2, 2>(p, st);
    case 64128064: return launch_bf16<64, 128, 64, 2, 2>(p, st);
    case 128160064: return launch_bf16<128, 160, 64, 2, 2>(p, st);
    case 64160064: return launch_bf16<64, 160, 64, 2, 2>(p, st);
    case 256160064:
      switch (pingpong(p)) {
        case 1: return launch_bf16<256, 160, 64, 4, 2, 3, 1>(p, st);
        case 2: return launch_bf16<256, 160, 64, 4, 2, 3, 2>(p, st);
        default: return launch_bf16<256, 160, 64, 4, 2, 3>(p, st);
      }
    case 256128064:   // (GLU layers and forced tiles: form 1 is not instantiated for it)
      return pingpong(p) == 2 ? launch_bf16<256, 128, 64, 4, 2, 3, 2>(p, st) : launch_bf16<256, 128, 64, 4, 2, 3>(p, st);
    default: break;
  }
  set_error("ldm_conv_gemm: no kernel for tile %dx%dx%d", bm, bn, bk);
  return LDM_EUNSUPPORTED;
}

extern "C" int ldm_conv_gemm_tile(const LdmConvGemm* d, int tile[3]) {
  LDM_REQUIRE(d && tile, "ldm_conv_gemm_tile: null pointer");
  if (d->dtype == LDM_F32) { tile[0] = 64; tile[1] = 64; tile[2] = 16; return LDM_OK; }
  ConvGemmParams p;
  p.c0 = d->c0; p.c1 = d->c1; p.ctot = d->c0 + d->c1;
  p.M = d->batch * d->hout * d->wout; p.N = d->n;
  p.K = d->ksize * d->ksize * p.ctot;
  p.rowstats = (d->epilogue & LDM_EPI_ROWSTATS) ? d->rowstats : nullptr;
  const bool glu = (d->epilogue & (LDM_EPI_GEGLU | LDM_EPI_SWIGLU)) != 0;
  const bool fold = (d->epilogue & (LDM_EPI_ROWSTATS | LDM_EPI_LNFOLD)) != 0;
  int splits;
  pick_tile_bf16(p, glu, d->workspace && !fold ? d->workspace_bytes : 0, tile[0], tile[1], tile[2], splits);
  return LDM_OK;
}
