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
#include <utility>

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
  int steps_per_split;     // steps (64 rows; 32 for the large-tile kernel) per blockIdx.y
  int splits;
  int cfg;                 // 0: 128x128 kernel; 1: <256,256>, 2: <320,256>, 3: <256,320> large tiles
};

__device__ __forceinline__ void wg_dma16(const void* base, unsigned bytes, unsigned char* lds, unsigned voff, unsigned soff) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, soff, 0, 0);
}

constexpr int WG_TN = 128, WG_TK = 128, WG_MS = 64;      // dW tile and rows per step
constexpr int WG_ROWB = 256;                              // bytes per LDS row of either image (128 bf16)
constexpr int WG_IMG = WG_MS * WG_ROWB;                   // 16 KiB per image
constexpr int WG_STAGE = 2 * WG_IMG;                      // dY image then A image

// TWO_SRC: the A operand is a two-source concat.  Compile-time, because the buffer descriptor of a lane's chunk would
// otherwise be lane-dependent and every LDS-DMA would sit in a readfirstlane waterfall loop.
template <bool TWO_SRC>
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
    a_second[par] = TWO_SRC && c >= p.c0;
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
      if (TWO_SRC && a_second[par]) wg_dma16(p.a1, bytes_a1, sa + piece * 1024, avo, 0);
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
  // One step: all fragments of this step's images go to registers first, THEN the LDS-DMA of the next step is issued,
  // then the MFMAs run with that DMA in flight.  (The compiler waits for every outstanding LDS-DMA in front of a
  // transposed LDS read -- it cannot tell which image the DMA writes -- so DMA issued before the reads would be waited
  // for at once and never overlap the math.)
  if (nsteps > 0) {
    stage(0, step_begin);
    for (int s = 0; s < nsteps; ++s) {
      __syncthreads();                                    // this step's images have landed; the other buffer is free
      const unsigned char* sdy = smem + (s & 1) * WG_STAGE;
      const unsigned char* sa = sdy + WG_IMG;
      bf16x8_t fn[2][4], fk[2][4];
#pragma unroll
      for (int st = 0; st < 2; ++st) {
#pragma unroll
        for (int i = 0; i < 4; ++i) fn[st][i] = frag(sdy, wn * 4 + i, st);
#pragma unroll
        for (int j = 0; j < 4; ++j) fk[st][j] = frag(sa, wk * 4 + j, st);
      }
      __builtin_amdgcn_sched_barrier(0);
      if (s + 1 < nsteps) stage((s + 1) & 1, step_begin + s + 1);
      __builtin_amdgcn_sched_barrier(0);
#pragma unroll
      for (int st = 0; st < 2; ++st)
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
          for (int j = 0; j < 4; ++j)
            acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(fn[st][i], fk[st][j], acc[i][j], 0, 0, 0);
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

// ----------------------------------------------------------------------------------------------------------------
// Large-tile kernel (the speed path).  The 128 x 128 tile above stages 64 flop per operand byte; the CU's L2->LDS path
// delivers ~28 B/clk, so it cannot pass ~45 % of the matrix-core peak.  Here a workgroup of 8 waves owns a T0 (n) x T1 (k)
// tile of dW with T0, T1 in {256, 320} (128-142 flop per staged byte; every N of the model is a multiple of 320 and so
// are the K of its 1x1 layers), walks the pixels 32 rows per step -- one MFMA contraction block -- and keeps the LDS-DMA
// of the next two steps in flight on a 3-slot ring with ONE barrier per step and counted vmcnt waits.  The transposed
// fragment reads are issued as inline asm: in front of the builtin the compiler waits for every outstanding LDS-DMA (it
// cannot tell which slot a DMA writes), which would serialise staging and math.
constexpr int W2_MS = 32;

// XOR swizzle of the 16-byte chunk index by row, chosen so that the 8 rows one 32-lane group of ds_read_b64_tr_b16
// touches (32 bytes = one chunk PAIR per row) land in 8 different 32-byte bank slots of the 256-byte LDS word:
//   512-byte rows (T = 256) all start at bank 0: flip the three pair bits with row & 7;
//   640-byte rows (T = 320) alternate between the two 128-byte halves: flip two pair bits with (row >> 1) & 3
//   (and stay inside aligned groups of 8 chunks: a row has 40).
template <int T>
__device__ __forceinline__ int w2_swz(int row) { return T == 256 ? (row & 7) << 1 : ((row >> 1) & 3) << 1; }

// FAST: stride 1, no upsample, and (1x1 or power-of-two extents) -- the per-step address arithmetic described below is
// compiled out of the other path entirely (its registers spilled, and every scratch reload waits for vmcnt(0), i.e. for
// all LDS-DMA in flight).  Single source only: concat layers run as two passes (wgrad_two_pass).
template <int T0, int T1, int W0, int W1, int MODE>   // MODE 0: general; 1: fast (see above); 2: fast and 1x1
__global__ void __launch_bounds__(512)
conv_wgrad_bf16_v2_kernel(const WgradParams p) {
  constexpr bool TWO_SRC = false;
  constexpr bool FAST = MODE != 0, ONE_TAP = MODE == 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  static_assert(W0 * W1 == 8, "8 waves");
  constexpr unsigned OOB = 0x80000000u;
  constexpr int F0 = T0 / W0 / 16, F1 = T1 / W1 / 16;     // 16-column fragments per wave along n and k
  constexpr int CPR0 = T0 / 8, CPR1 = T1 / 8;             // 16-byte chunks per image row
  constexpr int IMG0 = W2_MS * T0 * 2, IMG1 = W2_MS * T1 * 2;
  constexpr int SLOT = IMG0 + IMG1;
  constexpr int NI0 = IMG0 / 1024, NI1 = IMG1 / 1024, NI = NI0 + NI1;   // wave-wide DMA instructions per step
  constexpr int JMAX = (NI + 7) / 8;
  static_assert(IMG0 % 1024 == 0 && IMG1 % 1024 == 0, "whole DMA instructions");

  const int tid = threadIdx.x, lane = tid & 63;
  const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
  const int w0 = wave / W1, w1 = wave % W1;
  const int tn = blockIdx.x % p.tiles_n, tk = blockIdx.x / p.tiles_n;
  const int n0 = tn * T0, k0 = tk * T1;
  const int total_steps = (p.M + W2_MS - 1) / W2_MS;
  const int step_begin = blockIdx.y * p.steps_per_split;
  const int nsteps = min(step_begin + p.steps_per_split, total_steps) - step_begin;

  const unsigned bytes_a0 = (unsigned)p.batch * p.hin * p.win * p.c0 * 2u;
  const int hlim = p.upsample ? p.hin * 2 : p.hin, wlim = p.upsample ? p.win * 2 : p.win;
  const int pad = p.ksize >> 1;
  const float inv_w = 1.0f / (float)p.wout, inv_h = 1.0f / (float)p.hout;

  // ---- static staging geometry: instruction q = j*8 + wave of a step moves chunks q'*64 .. +63 of one image
  // (q' = q for the dY image, q - NI0 for the A image); lane -> (row, physical chunk); it fetches the logical chunk
  // phys ^ (row & 7), so its column group is fixed for the whole kernel.
  //
  // The staging must cost (almost) no vector ALU: with ~45 VALU instructions per DMA the address arithmetic of two waves
  // per SIMD took longer than the MFMAs.  So the buffer descriptors move instead of the lane offsets: per step the
  // scalar unit advances the descriptor base by 32 rows and shrinks num_records (rows past M fall out of range and read
  // zeros), and a lane's byte offset from that base is a kernel-long constant -- for dY outright, and for the im2col
  // gather of a stride-1 layer too, because there input pixel = output pixel + (kh-1)*W + (kw-1) is linear in the row
  // (base biased by W+1 pixels so the offset never goes negative).  Only the padding test of a 3x3 tap is per step:
  // (oh, ow) of the row by shift/mask when W and H are powers of two (float reciprocals otherwise), two unsigned
  // compares, one select.  Stride-2 and upsampled layers (six in the model) keep the general per-step arithmetic.
  constexpr bool fast = FAST;                             // hout == hin, wout == win
  const bool pow2 = FAST || ((p.wout & (p.wout - 1)) == 0 && (p.hout & (p.hout - 1)) == 0);
  const int wlog = 31 - __builtin_clz(p.wout);
  int s_row[JMAX];          // row of the step this lane stages in instruction j
  unsigned s_col[JMAX];     // byte offset from the step's descriptor base (dY, fast A) or channel byte offset inside the
                            // tap (general A); OOB when the column group lies past N / K
  int s_tap[JMAX];          // A: kh | kw << 2 | second source << 4
#pragma unroll
  for (int j = 0; j < JMAX; ++j) {
    const int q = j * 8 + wave;
    const bool isA = q >= NI0;
    const int c = (isA ? q - NI0 : q) * 64 + lane;
    const int cpr = isA ? CPR1 : CPR0;
    const int row = c / cpr, phys = c - row * cpr;
    const int col = (phys ^ (isA ? w2_swz<T1>(row) : w2_swz<T0>(row))) * 8;
    s_row[j] = row;
    s_tap[j] = 0;
    if (!isA) {
      s_col[j] = n0 + col < p.N ? ((unsigned)row * p.N + n0 + col) * 2u : OOB;
    } else {
      const int k = k0 + col;
      if (k < p.K) {
        const int tap = k / p.ctot;
        int ch = k - tap * p.ctot;
        const int kh = tap / p.ksize, kw = tap - kh * p.ksize;
        const bool second = TWO_SRC && ch >= p.c0;
        if (second) ch -= p.c0;
        const int cs = second ? p.c1 : p.c0;
        s_tap[j] = kh | (kw << 2) | (second ? 16 : 0);
        if (fast) s_col[j] = ((unsigned)(row + (kh - pad) * p.win + (kw - pad) + p.win + 1) * cs + ch) * 2u;
        else s_col[j] = (unsigned)ch * 2u;
      } else {
        s_col[j] = OOB;
      }
    }
  }
  // first pixel of the current staging step as (sample, oh, ow): wave-uniform, advanced by 32 rows per staged step
  // (general path and non-power-of-two extents only)
  const int hw_out = p.hout * p.wout;
  int g_b = 0, g_oh = 0, g_ow = 0;
  if constexpr (!FAST) {
    const long long m = (long long)step_begin * W2_MS;
    g_b = (int)(m / hw_out);
    const int rem = (int)(m - (long long)g_b * hw_out);
    g_oh = rem / p.wout;
    g_ow = rem - g_oh * p.wout;
  }
  const int d_ow = W2_MS % p.wout, d_ohr = W2_MS / p.wout;
  const int d_oh = d_ohr % p.hout, d_b = d_ohr / p.hout;
  const long long total_pix = (long long)p.batch * p.hin * p.win;

  auto rsrc = [](const void* base, long long bytes) {
    const int n = bytes < 0 ? 0 : bytes > 0x7fffffffLL ? 0x7fffffff : (int)bytes;
    return __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(base), 0, n, 0x00020000);
  };
  auto dma = [](__amdgpu_buffer_rsrc_t rs, unsigned char* lds, unsigned voff) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)lds, 16, voff, 0, 0, 0);
  };

  // Running descriptor state (wave-uniform): stage() is called for consecutive steps, so bases and remaining byte counts
  // advance by one step per call -- a handful of scalar adds.  (Recomputing them per call from the step index cost ~80
  // scalar instructions of 64-bit arithmetic and, with the branches around it, made a wave's serial staging code longer
  // than its 32 MFMAs: the matrix cores idled ~20 % of every step.)
  const long long dy_step = (long long)W2_MS * p.N * 2;
  const long long a_step = (long long)W2_MS * p.c0 * 2;    // fast path: a step is 32 pixels of the input, too
  const char* dy_ptr = static_cast<const char*>(p.dy) + (long long)step_begin * dy_step;
  long long dy_left = ((long long)p.M - (long long)step_begin * W2_MS) * p.N * 2;
  // fast path: base biased back by W+1 pixels (never dereferenced below the allocation: those lanes are padding)
  const long long pix_first = (long long)step_begin * W2_MS - p.win - 1;
  const char* a_ptr = static_cast<const char*>(p.a0) + pix_first * p.c0 * 2;
  long long a_left = (total_pix - pix_first) * p.c0 * 2;
  const __amdgpu_buffer_rsrc_t rs_a_general = rsrc(p.a0, bytes_a0);
  int st_row0 = step_begin * W2_MS;                        // first output row of the step being staged
  auto clamp31 = [](long long b) {                         // num_records of a descriptor, from 32-bit scalar compares
    const int hi = (int)(b >> 32);
    const unsigned lo = (unsigned)b;
    return hi < 0 ? 0 : (hi > 0 || lo > 0x7fffffffu) ? 0x7fffffff : (int)lo;
  };

  auto stage = [&](int slot, int /*step: consecutive from step_begin*/) {
    unsigned char* s0 = smem + slot * SLOT;
    const int row0 = st_row0;
    const __amdgpu_buffer_rsrc_t rs_dy = __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(dy_ptr), 0, clamp31(dy_left), 0x00020000);
    const __amdgpu_buffer_rsrc_t rs_a0 =
        FAST ? __builtin_amdgcn_make_buffer_rsrc(const_cast<char*>(a_ptr), 0, clamp31(a_left), 0x00020000) : rs_a_general;
    [&]<int... J>(std::integer_sequence<int, J...>) {
      ([&] {
        constexpr int j = J;
        const int q = j * 8 + wave;
        bool live = true;
        if constexpr ((j + 1) * 8 > NI) live = q < NI;                       // only the last pass can run past the images
        bool is_dy;
        if constexpr ((j + 1) * 8 <= NI0) is_dy = true;                      // whole pass inside the dY image
        else if constexpr (j * 8 >= NI0) is_dy = false;                      // whole pass inside the A image
        else is_dy = q < NI0;                                                // straddling pass: wave-uniform branch
        if (live && is_dy) {
          dma(rs_dy, s0 + q * 1024, s_col[j]);
        } else if (live) {
          unsigned char* dst = s0 + IMG0 + (q - NI0) * 1024;
          const int kh = s_tap[j] & 3, kw = (s_tap[j] >> 2) & 3;
          unsigned vo;
          if constexpr (ONE_TAP) {
            vo = s_col[j];
          } else {
            int oh, ow, b = 0;
            if (pow2) {
              const int m = row0 + s_row[j];
              ow = m & (p.wout - 1);
              oh = (m >> wlog) & (p.hout - 1);
              if (!fast) b = m >> (wlog + (31 - __builtin_clz(p.hout)));
            } else {    // exact float reciprocals: the operands are far below 2^20
              const int pw = g_ow + s_row[j];
              const int qw = (int)(((float)pw + 0.5f) * inv_w);
              ow = pw - qw * p.wout;
              const int ph = g_oh + qw;
              const int qh = (int)(((float)ph + 0.5f) * inv_h);
              oh = ph - qh * p.hout;
              b = g_b + qh;
            }
            if (fast) {
              const bool ok = (unsigned)(oh + kh - pad) < (unsigned)p.hin && (unsigned)(ow + kw - pad) < (unsigned)p.win;
              vo = ok ? s_col[j] : OOB;
            } else {
              const unsigned m = (unsigned)row0 + s_row[j];
              int ih = oh * p.stride - pad + kh, iw = ow * p.stride - pad + kw;
              const bool ok = s_col[j] != OOB && m < (unsigned)p.M && ih >= 0 && ih < hlim && iw >= 0 && iw < wlim;
              if (p.upsample) { ih >>= 1; iw >>= 1; }
              vo = ok ? (unsigned)((b * p.hin + ih) * p.win + iw) * (unsigned)p.c0 * 2u + s_col[j] : OOB;
            }
          }
          dma(rs_a0, dst, vo);
        }
      }(), ...);
    }(std::make_integer_sequence<int, JMAX>{});
    dy_ptr += dy_step; dy_left -= dy_step;
    if constexpr (FAST) { a_ptr += a_step; a_left -= a_step; }
    st_row0 += W2_MS;
    if constexpr (!FAST) {
      g_ow += d_ow;
      if (g_ow >= p.wout) { g_ow -= p.wout; g_oh += 1; }
      g_oh += d_oh;
      g_b += d_b;
      if (g_oh >= p.hout) { g_oh -= p.hout; g_b += 1; }
    }
  };

  f32x4_t acc[F0][F1];
#pragma unroll
  for (int i = 0; i < F0; ++i)
#pragma unroll
    for (int j = 0; j < F1; ++j) acc[i][j] = f32x4_t{0.f, 0.f, 0.f, 0.f};

  // transposed fragment addresses: lane -> (row 4*fq + tq, chunk 2*ct + (tp >> 1), half (tp & 1)); the second read of a
  // fragment is 16 rows further down, where (row + 16) & 7 == row & 7 leaves the swizzle unchanged
  const int fr = lane & 15, fq = lane >> 4, tq = fr >> 2, tp = fr & 3;
  const int frow = 4 * fq + tq;
  const unsigned lds_base = (unsigned)(uintptr_t)(__attribute__((address_space(3))) unsigned char*)smem;
  unsigned off0[F0], off1[F1];                            // byte offsets of this lane's fragments inside a slot
#pragma unroll
  for (int i = 0; i < F0; ++i) {
    const int lchunk = 2 * ((w0 * F0 + i)) + (tp >> 1);
    off0[i] = frow * (T0 * 2) + ((lchunk ^ w2_swz<T0>(frow)) << 4) + ((tp & 1) << 3);
  }
#pragma unroll
  for (int j = 0; j < F1; ++j) {
    const int lchunk = 2 * ((w1 * F1 + j)) + (tp >> 1);
    off1[j] = IMG0 + frow * (T1 * 2) + ((lchunk ^ w2_swz<T1>(frow)) << 4) + ((tp & 1) << 3);
  }

  // this wave's DMA instructions per step (vmcnt counts them in issue order)
  const int my_cnt = (NI - wave + 7) / 8;

  if (nsteps > 0) {
    stage(0, step_begin);
    if (nsteps > 1) stage(1, step_begin + 1);
    int slot = 0;
    for (int s = 0; s < nsteps; ++s) {
      if (s + 1 < nsteps) {
        if (my_cnt == JMAX) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JMAX) : "memory");
        else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(JMAX - 1) : "memory");
      } else {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
      }
      __builtin_amdgcn_s_barrier();                       // step s has landed for every wave; slot of step s-1 is free
      asm volatile("" ::: "memory");
      const unsigned sbase = lds_base + slot * SLOT;
      bf16x4_t lo0[F0], hi0[F0], lo1[F1], hi1[F1];
      // Fragment reads in the order the MFMAs need them -- the shorter list (needed by every group) first, then the longer
      // one group by group -- and counted waits: group g starts as soon as ITS reads are back (lgkmcnt counts in issue
      // order), while the reads of the later groups are still in flight.  One wait for everything left the matrix cores
      // idle for the ~400 cycles the CU's LDS needs to serve 8 waves x 24 transposed reads after each barrier.
      constexpr bool OUT0 = F0 >= F1;                      // outer loop over the longer fragment list
      constexpr int NOUT = OUT0 ? F0 : F1;
      auto read0 = [&](int i) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo0[i]) : "v"(sbase + off0[i]) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi0[i]) : "v"(sbase + off0[i]), "n"(16 * T0 * 2) : "memory");
      };
      auto read1 = [&](int j) {
        asm volatile("ds_read_b64_tr_b16 %0, %1" : "=v"(lo1[j]) : "v"(sbase + off1[j]) : "memory");
        asm volatile("ds_read_b64_tr_b16 %0, %1 offset:%2" : "=v"(hi1[j]) : "v"(sbase + off1[j]), "n"(16 * T1 * 2) : "memory");
      };
      if constexpr (OUT0) {
#pragma unroll
        for (int j = 0; j < F1; ++j) read1(j);
#pragma unroll
        for (int i = 0; i < F0; ++i) read0(i);
      } else {
#pragma unroll
        for (int i = 0; i < F0; ++i) read0(i);
#pragma unroll
        for (int j = 0; j < F1; ++j) read1(j);
      }
      if (s + 2 < nsteps) stage(slot == 0 ? 2 : slot - 1, step_begin + s + 2);
      bf16x8_t f0[F0], f1[F1];
      auto join = [](const bf16x4_t& lo, const bf16x4_t& hi) {
        return bf16x8_t{lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
      };
      [&]<int... G>(std::integer_sequence<int, G...>) {
        ([&] {
          // all but the reads of the groups after G have returned (scalar loads in between only make the wait stricter)
          constexpr int LEFT = 2 * (NOUT - 1 - G);         // the counter saturates at 15: a smaller bound only waits longer
          asm volatile("s_waitcnt lgkmcnt(%0)" ::"n"(LEFT < 15 ? LEFT : 15) : "memory");
          // tie the fragments to a point behind the wait (volatile asm keeps its order): no MFMA on them moves above it
          if constexpr (OUT0) {
            if constexpr (G == 0) {
#pragma unroll
              for (int j = 0; j < F1; ++j) { asm volatile("" : "+v"(lo1[j]), "+v"(hi1[j])); f1[j] = join(lo1[j], hi1[j]); }
            }
            asm volatile("" : "+v"(lo0[G]), "+v"(hi0[G]));
            f0[G] = join(lo0[G], hi0[G]);
#pragma unroll
            for (int j = 0; j < F1; ++j) acc[G][j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0[G], f1[j], acc[G][j], 0, 0, 0);
          } else {
            if constexpr (G == 0) {
#pragma unroll
              for (int i = 0; i < F0; ++i) { asm volatile("" : "+v"(lo0[i]), "+v"(hi0[i])); f0[i] = join(lo0[i], hi0[i]); }
            }
            asm volatile("" : "+v"(lo1[G]), "+v"(hi1[G]));
            f1[G] = join(lo1[G], hi1[G]);
#pragma unroll
            for (int i = 0; i < F0; ++i) acc[i][G] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(f0[i], f1[G], acc[i][G], 0, 0, 0);
          }
          __builtin_amdgcn_sched_barrier(0);
        }(), ...);
      }(std::make_integer_sequence<int, NOUT>{});
      slot = slot == 2 ? 0 : slot + 1;
    }
  }

  // ---- partial tile: C[row = n = fq*4 + r][col = k = fr]
  float* out = p.partial + (size_t)blockIdx.y * p.N * p.K;
#pragma unroll
  for (int i = 0; i < F0; ++i)
#pragma unroll
    for (int r = 0; r < 4; ++r) {
      const int n = n0 + (w0 * F0 + i) * 16 + fq * 4 + r;
      if (n >= p.N) continue;
#pragma unroll
      for (int j = 0; j < F1; ++j) {
        const int k = k0 + (w1 * F1 + j) * 16 + fr;
        if (k < p.K) out[(size_t)n * p.K + k] = acc[i][j][r];
      }
    }
}

// dw[n][c_off + ci][tap] (+)= sum_s partial[s][n][tap*csrc + ci]   (fixed order; OIHW = the checkpoint layout).  A concat
// layer's two sources are run as two passes: each fills its own channel range [c_off, c_off + csrc) of the ctot_dst channels.
template <int TAPS>   // 1 / 9: constant divisions; 0: any (run-time)
__global__ void __launch_bounds__(256)
wgrad_fold_kernel(const float* __restrict__ partial, float* __restrict__ dw, int N, int K, int csrc, int taps_arg, int splits,
                  int accumulate, int c_off, int ctot_dst) {
  const int taps = TAPS ? TAPS : taps_arg;
  const unsigned total = (unsigned)N * (unsigned)K;                    // < 2^31 (checked on the host)
  const unsigned idx = blockIdx.x * 256u + threadIdx.x;                // over this pass's dw elements, (n, ci, tap) order
  if (idx >= total) return;
  const unsigned n = idx / (unsigned)K, rem = idx - n * (unsigned)K;
  const unsigned ci = rem / (unsigned)taps, tap = rem - ci * taps;
  const float* src = partial + (size_t)n * K + (size_t)tap * csrc + ci;
  const size_t dst = ((size_t)n * ctot_dst + c_off + ci) * taps + tap;
  const size_t sstride = (size_t)total;
  float v = accumulate ? dw[dst] : 0.f;
  int s = 0;
  for (; s + 3 < splits; s += 4) {                                     // four splits in flight; fixed order
    const float p0 = src[(size_t)s * sstride], p1 = src[(size_t)(s + 1) * sstride];
    const float p2 = src[(size_t)(s + 2) * sstride], p3 = src[(size_t)(s + 3) * sstride];
    v = (((v + p0) + p1) + p2) + p3;
  }
  for (; s < splits; ++s) v += src[(size_t)s * sstride];
  dw[dst] = v;
}

// The 3x3 form of the fold, laid out for the reads (they are `splits` times the writes): a wave takes 64 consecutive input
// channels of one filter n, so each of its 9 x splits loads is one contiguous 256-byte run of the partial tile; the nine sums
// of a lane are the nine taps of one (n, ci), contiguous in OIHW, and go out through LDS as one contiguous run per wave.
// Same order of additions per element as wgrad_fold_kernel.
__global__ void __launch_bounds__(256)
wgrad_fold9_kernel(const float* __restrict__ partial, float* __restrict__ dw, int N, int K, int csrc, int splits, int accumulate,
                   int c_off, int ctot_dst) {
  __shared__ float tr[4][64 * 9];
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  const int n = blockIdx.y * 4 + wv, ci0 = blockIdx.x * 64, ci = ci0 + lane;
  const bool live = n < N && ci < csrc;
  const size_t total = (size_t)N * K;
  const size_t dst0 = ((size_t)(n < N ? n : 0) * ctot_dst + c_off + ci0) * 9;
  float v[9];
#pragma unroll
  for (int t = 0; t < 9; ++t) v[t] = (live && accumulate) ? dw[dst0 + lane * 9 + t] : 0.f;
  if (live) {
    const float* src = partial + (size_t)n * K + ci;
    int s = 0;
    for (; s + 1 < splits; s += 2) {
      float p0[9], p1[9];
#pragma unroll
      for (int t = 0; t < 9; ++t) { p0[t] = src[(size_t)s * total + t * csrc]; p1[t] = src[(size_t)(s + 1) * total + t * csrc]; }
#pragma unroll
      for (int t = 0; t < 9; ++t) v[t] = (v[t] + p0[t]) + p1[t];
    }
    if (s < splits) {
#pragma unroll
      for (int t = 0; t < 9; ++t) v[t] += src[(size_t)s * total + t * csrc];
    }
  }
#pragma unroll
  for (int t = 0; t < 9; ++t) tr[wv][lane * 9 + t] = v[t];
  __syncthreads();
  if (n < N) {
    const int run = min(64, csrc - ci0) * 9;
#pragma unroll
    for (int t = 0; t < 9; ++t) {
      const int o = t * 64 + lane;
      if (o < run) dw[dst0 + o] = tr[wv][o];
    }
  }
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

template <int T0, int T1, int W0, int W1, int MODE>
static int launch_wgrad_v2(const WgradParams& p, dim3 grid, hipStream_t st) {
  constexpr int smem = 3 * W2_MS * (T0 + T1) * 2;
  static bool attr_done = false;
  if (!attr_done) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_v2_kernel<T0, T1, W0, W1, MODE>),
                                       hipFuncAttributeMaxDynamicSharedMemorySize, smem);
    if (e != hipSuccess) { set_error("ldm_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
    attr_done = true;
  }
  hipLaunchKernelGGL((conv_wgrad_bf16_v2_kernel<T0, T1, W0, W1, MODE>), grid, dim3(512), smem, st, p);
  return LDM_OK;
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
  p.cfg = 0;
  if (!getenv("LDM_WGRAD_V1") && d->c1 == 0) {   // (a concat with c1 % 8 != 0 stays on the 128x128 kernel)
    // large tiles: the shape with the least padded work; one workgroup per CU, so aim at one full round of 256
    static const int shapes[3][2] = {{256, 256}, {320, 256}, {256, 320}};
    long long best = -1;
    // the 320-wide tiles of the general (strided / upsampled / non-power-of-two) path would spill: 256 x 256 there
    const bool pow2 = (d->wout & (d->wout - 1)) == 0 && (d->hout & (d->hout - 1)) == 0;
    const bool fastk = d->stride == 1 && !d->upsample && (d->ksize == 1 || pow2);
    for (int c = 0; c < (fastk ? 3 : 1); ++c) {
      const long long tn = (p.N + shapes[c][0] - 1) / shapes[c][0], tk = (p.K + shapes[c][1] - 1) / shapes[c][1];
      const long long padded = tn * shapes[c][0] * tk * shapes[c][1];
      if (best < 0 || padded < best) { best = padded; p.cfg = c + 1; p.tiles_n = (int)tn; p.tiles_k = (int)tk; }
    }
    const int steps = (p.M + W2_MS - 1) / W2_MS;
    const int tiles = p.tiles_n * p.tiles_k;
    int splits = tiles >= 256 ? 1 : 256 / tiles;
    if (splits > steps / 16) splits = steps / 16;
    if (splits < 1) splits = 1;
    p.steps_per_split = (steps + splits - 1) / splits;
    p.splits = (steps + p.steps_per_split - 1) / p.steps_per_split;
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

// A concat layer on the matrix-core path runs as two single-source passes (a lane-dependent source would put every
// LDS-DMA of the gather into a readfirstlane waterfall loop).
static bool wgrad_two_pass(const LdmConvGemm* d) { return d->c1 > 0 && wgrad_use_mfma(d) && d->c1 % 8 == 0 && !getenv("LDM_WGRAD_V1"); }

static LdmConvGemm wgrad_source(const LdmConvGemm* d, int which) {
  LdmConvGemm s = *d;
  if (which == 1) { s.a0 = d->a1; s.c0 = d->c1; }
  s.a1 = nullptr; s.c1 = 0;
  return s;
}

static int wgrad_run(const LdmConvGemm* d, const void* dy, float* dw, int accumulate, void* workspace, long long workspace_bytes,
                     void* stream, int c_off, int ctot_dst);

extern "C" long long ldm_conv_wgrad_workspace_bytes(const LdmConvGemm* d) {
  if (!d || d->batch <= 0) return 0;
  if (wgrad_two_pass(d)) {
    const LdmConvGemm s0 = wgrad_source(d, 0), s1 = wgrad_source(d, 1);
    const long long b0 = ldm_conv_wgrad_workspace_bytes(&s0), b1 = ldm_conv_wgrad_workspace_bytes(&s1);
    return b0 > b1 ? b0 : b1;
  }
  WgradParams p;
  wgrad_plan(d, p);
  return (long long)p.splits * p.N * p.K * 4;
}

extern "C" int ldm_conv_wgrad(const LdmConvGemm* d, const void* dy, float* dw, int accumulate, void* workspace,
                              long long workspace_bytes, void* stream) {
  LDM_REQUIRE(d != nullptr, "ldm_conv_wgrad: null descriptor");
  if (d->batch > 0 && wgrad_two_pass(d)) {
    LDM_REQUIRE(d->a1 != nullptr, "ldm_conv_wgrad: null pointer");
    const LdmConvGemm s0 = wgrad_source(d, 0), s1 = wgrad_source(d, 1);
    int rc = wgrad_run(&s0, dy, dw, accumulate, workspace, workspace_bytes, stream, 0, d->c0 + d->c1);
    if (rc) return rc;
    return wgrad_run(&s1, dy, dw, accumulate, workspace, workspace_bytes, stream, d->c0, d->c0 + d->c1);
  }
  return wgrad_run(d, dy, dw, accumulate, workspace, workspace_bytes, stream, 0, d->c0 + d->c1);
}

static int wgrad_run(const LdmConvGemm* d, const void* dy, float* dw, int accumulate, void* workspace, long long workspace_bytes,
                     void* stream, int c_off, int ctot_dst) {
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
      hipError_t e = hipMemsetAsync(dw, 0, (size_t)p.N * p.K * 4, st);   // (batch 0 never takes the two-pass route)
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
      hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e == hipSuccess)
        e = hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wgrad_bf16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, smem);
      if (e != hipSuccess) { set_error("ldm_conv_wgrad: hipFuncSetAttribute: %s", hipGetErrorString(e)); return LDM_EHIP; }
      attr_done = true;
    }
    dim3 grid(p.tiles_n * p.tiles_k, p.splits);
    if (p.cfg != 0) {
      int rc = LDM_OK;
      const bool pow2 = (d->wout & (d->wout - 1)) == 0 && (d->hout & (d->hout - 1)) == 0;
      const bool fastk = d->stride == 1 && !d->upsample && (d->ksize == 1 || pow2);
      const int mode = !fastk ? 0 : d->ksize == 1 ? 2 : 1;
      switch (p.cfg * 4 + mode) {
        case 4: rc = launch_wgrad_v2<256, 256, 2, 4, 0>(p, grid, st); break;
        case 5: rc = launch_wgrad_v2<256, 256, 2, 4, 1>(p, grid, st); break;
        case 6: rc = launch_wgrad_v2<256, 256, 2, 4, 2>(p, grid, st); break;
        case 9: rc = launch_wgrad_v2<320, 256, 2, 4, 1>(p, grid, st); break;
        case 10: rc = launch_wgrad_v2<320, 256, 2, 4, 2>(p, grid, st); break;
        case 13: rc = launch_wgrad_v2<256, 320, 4, 2, 1>(p, grid, st); break;
        case 14: rc = launch_wgrad_v2<256, 320, 4, 2, 2>(p, grid, st); break;
        default: set_error("ldm_conv_wgrad: internal: tile %d without the fast path", p.cfg); return LDM_EINVAL;
      }
      if (rc) return rc;
    } else if (d->c1) hipLaunchKernelGGL(conv_wgrad_bf16_kernel<true>, grid, dim3(256), smem, st, p);
    else hipLaunchKernelGGL(conv_wgrad_bf16_kernel<false>, grid, dim3(256), smem, st, p);
  } else {
    dim3 grid((unsigned)(((long long)p.N * p.K + 255) / 256), p.splits);
    if (d->dtype == LDM_BF16) hipLaunchKernelGGL((conv_wgrad_generic_kernel<bf16_t>), grid, dim3(256), 0, st, p);
    else hipLaunchKernelGGL((conv_wgrad_generic_kernel<float>), grid, dim3(256), 0, st, p);
  }
  LDM_HIP_CHECK_LAUNCH("ldm_conv_wgrad");
  const long long total = (long long)p.N * p.K;
  LDM_REQUIRE(total < (1LL << 31), "ldm_conv_wgrad: filter of %lld elements", total);
  const dim3 fgrid((unsigned)((total + 255) / 256));
  const int ftaps = d->ksize * d->ksize;
  if (ftaps == 9 && !getenv("LDM_WGRAD_FOLD_V1"))
    hipLaunchKernelGGL(wgrad_fold9_kernel, dim3((p.ctot + 63) / 64, (p.N + 3) / 4), dim3(256), 0, st, p.partial, dw, p.N, p.K, p.ctot, p.splits,
                       accumulate, c_off, ctot_dst);
  else if (ftaps == 9)
    hipLaunchKernelGGL(wgrad_fold_kernel<9>, fgrid, dim3(256), 0, st, p.partial, dw, p.N, p.K, p.ctot, ftaps, p.splits, accumulate, c_off, ctot_dst);
  else if (ftaps == 1)
    hipLaunchKernelGGL(wgrad_fold_kernel<1>, fgrid, dim3(256), 0, st, p.partial, dw, p.N, p.K, p.ctot, ftaps, p.splits, accumulate, c_off, ctot_dst);
  else
    hipLaunchKernelGGL(wgrad_fold_kernel<0>, fgrid, dim3(256), 0, st, p.partial, dw, p.N, p.K, p.ctot, ftaps, p.splits, accumulate, c_off, ctot_dst);
  LDM_HIP_CHECK_LAUNCH("ldm_conv_wgrad(fold)");
  return LDM_OK;
}
