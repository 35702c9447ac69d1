// Shared device/host helpers for the gfx950 latent-diffusion kernels.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdarg.h>

#include "../../include/ldm_amd.h"

// Marks a point where the lanes of ONE wave hand data to each other through LDS and rely on the wave executing in lockstep
// (the hardware needs no instruction there).  Expands to nothing; the host emulator of the test suite (tests/emu), whose
// lanes are fibers, turns it into a wave rendezvous.
#ifndef LDM_LOCKSTEP
#define LDM_LOCKSTEP() do {} while (0)
#endif

namespace ldm {

// ----------------------------------------------------------------------------- error plumbing
void set_error(const char* fmt, ...);

#define LDM_REQUIRE(cond, ...)                 \
  do {                                         \
    if (!(cond)) {                             \
      ::ldm::set_error(__VA_ARGS__);           \
      return LDM_EINVAL;                       \
    }                                          \
  } while (0)

#define LDM_HIP_CHECK_LAUNCH(what)                                                   \
  do {                                                                               \
    hipError_t e__ = hipGetLastError();                                              \
    if (e__ != hipSuccess) {                                                         \
      ::ldm::set_error("%s: launch failed: %s", what, hipGetErrorString(e__));       \
      return LDM_EHIP;                                                               \
    }                                                                                \
  } while (0)

// ----------------------------------------------------------------------------------- bf16 bits
typedef uint16_t bf16_t;  // raw storage

__device__ __forceinline__ float bf16_to_f32(bf16_t v) { return __uint_as_float(((uint32_t)v) << 16); }

// round-to-nearest-even, NaN stays NaN: the hardware conversion (v_cvt_pk_bf16_f32), one instruction per pair
typedef __attribute__((ext_vector_type(2))) __bf16 bf16x2_hw_t;
typedef __attribute__((ext_vector_type(2))) float f32x2_hw_t;

__device__ __forceinline__ uint32_t pack_bf16x2(float lo, float hi) {
  return __builtin_bit_cast(uint32_t, __builtin_convertvector(f32x2_hw_t{lo, hi}, bf16x2_hw_t));
}

__device__ __forceinline__ bf16_t f32_to_bf16(float f) { return (bf16_t)(pack_bf16x2(f, 0.0f) & 0xffffu); }

template <typename T>
struct Elem;
template <>
struct Elem<float> {
  static __device__ __forceinline__ float ld(const float* p) { return *p; }
  static __device__ __forceinline__ void st(float* p, float v) { *p = v; }
};
template <>
struct Elem<bf16_t> {
  static __device__ __forceinline__ float ld(const bf16_t* p) { return bf16_to_f32(*p); }
  static __device__ __forceinline__ void st(bf16_t* p, float v) { *p = f32_to_bf16(v); }
};

// 8-element vector load/store as float[8]
__device__ __forceinline__ void load8(const bf16_t* p, float (&v)[8]) {
  uint4 r = *reinterpret_cast<const uint4*>(p);
  uint32_t w[4] = {r.x, r.y, r.z, r.w};
#pragma unroll
  for (int i = 0; i < 4; ++i) {
    v[2 * i] = __uint_as_float(w[i] << 16);
    v[2 * i + 1] = __uint_as_float(w[i] & 0xffff0000u);
  }
}
__device__ __forceinline__ void load8(const float* p, float (&v)[8]) {
  float4 a = *reinterpret_cast<const float4*>(p);
  float4 b = *reinterpret_cast<const float4*>(p + 4);
  v[0] = a.x; v[1] = a.y; v[2] = a.z; v[3] = a.w;
  v[4] = b.x; v[5] = b.y; v[6] = b.z; v[7] = b.w;
}
__device__ __forceinline__ void store8(bf16_t* p, const float (&v)[8]) {
  uint4 r;
  r.x = pack_bf16x2(v[0], v[1]);
  r.y = pack_bf16x2(v[2], v[3]);
  r.z = pack_bf16x2(v[4], v[5]);
  r.w = pack_bf16x2(v[6], v[7]);
  *reinterpret_cast<uint4*>(p) = r;
}
__device__ __forceinline__ void store8(float* p, const float (&v)[8]) {
  *reinterpret_cast<float4*>(p) = make_float4(v[0], v[1], v[2], v[3]);
  *reinterpret_cast<float4*>(p + 4) = make_float4(v[4], v[5], v[6], v[7]);
}

__device__ __forceinline__ float silu_f(float x) { return x / (1.0f + __expf(-x)); }
__device__ __forceinline__ float gelu_erf_f(float x) { return 0.5f * x * (1.0f + erff(x * 0.70710678118654752440f)); }

// Epilogue forms for the bf16 path (results are rounded to bf16, 2^-9 relative): raw v_rcp_f32 / v_exp_f32, no
// denormal or range fix-up code.
//   silu_fast: exp2 overflow for x < -88 gives rcp(inf) = 0, i.e. the correct limit -0.
//   gelu_erf_fast: 0.5 x (1 + erf(x / sqrt 2)) with erf from Abramowitz & Stegun 7.1.26 (|abs error| <= 1.5e-7).
//     For x < 0 it uses 1 + erf(-z) = erfc(z) = poly(t) exp(-z^2) directly, so the tail has no cancellation.
__device__ __forceinline__ float silu_fast(float x) {
  return x * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * x));
}
__device__ __forceinline__ float gelu_erf_fast(float x) {
  const float z = fabsf(x) * 0.70710678118654752440f;
  const float t = __builtin_amdgcn_rcpf(fmaf(0.3275911f, z, 1.0f));
  float q = fmaf(t, 1.061405429f, -1.453152027f);
  q = fmaf(q, t, 1.421413741f);
  q = fmaf(q, t, -0.284496736f);
  q = fmaf(q, t, 0.254829592f);
  const float erfc_z = q * t * __builtin_amdgcn_exp2f(-1.44269504088896341f * z * z);
  return 0.5f * x * (x >= 0.0f ? 2.0f - erfc_z : erfc_z);
}

__device__ __forceinline__ float wave_sum(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
  return v;
}
__device__ __forceinline__ float wave_max(float v) {
#pragma unroll
  for (int o = 32; o > 0; o >>= 1) v = fmaxf(v, __shfl_xor(v, o, 64));
  return v;
}

static inline size_t dtype_size(int dt) { return dt == LDM_BF16 ? 2 : 4; }
static inline bool dtype_ok(int dt) { return dt == LDM_BF16 || dt == LDM_F32; }
static inline bool aligned16(const void* p) { return (reinterpret_cast<uintptr_t>(p) & 15) == 0; }

}  // namespace ldm
