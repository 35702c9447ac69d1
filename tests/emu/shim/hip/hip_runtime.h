// Stand-in for <hip/hip_runtime.h> when the kernel sources are compiled for the HOST against the SIMT emulator
// (tests/emu/emu.h).  Only what latent-diffusion_amd/csrc uses.  TEST INFRASTRUCTURE.
#pragma once
#include <algorithm>
#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <type_traits>
#include <utility>

#include "emu.h"

#define LDM_EMU 1
#define LDM_LOCKSTEP() ::emu::wave_barrier()
#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __launch_bounds__(...)
#define __shared__ static
#define __align__(n) __attribute__((aligned(n)))
#define HIP_SYMBOL(x) x

using emu::dim3;
#define threadIdx (::emu::me().tid)
#define blockIdx (::emu::g_block_idx)
#define blockDim (::emu::g_block_dim)
#define gridDim (::emu::g_grid_dim)

// ------------------------------------------------------------------------------------------- vector types
struct alignas(8) float2 { float x, y; };
struct alignas(16) float4 { float x, y, z, w; };
struct alignas(8) uint2 { unsigned x, y; };
struct alignas(16) uint4 { unsigned x, y, z, w; };
struct alignas(8) int2 { int x, y; };
struct alignas(16) int4 { int x, y, z, w; };
struct alignas(16) double2 { double x, y; };
static inline float2 make_float2(float x, float y) { return float2{x, y}; }
static inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }
static inline uint2 make_uint2(unsigned x, unsigned y) { return uint2{x, y}; }
static inline uint4 make_uint4(unsigned x, unsigned y, unsigned z, unsigned w) { return uint4{x, y, z, w}; }
static inline int2 make_int2(int x, int y) { return int2{x, y}; }
static inline int4 make_int4(int x, int y, int z, int w) { return int4{x, y, z, w}; }
static inline double2 make_double2(double x, double y) { return double2{x, y}; }

// ------------------------------------------------------------------------------------------------ scalars
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __int_as_float(int u) { float f; memcpy(&f, &u, 4); return f; }
static inline int __float_as_int(float f) { int u; memcpy(&u, &f, 4); return u; }
#define __expf(x) expf(x)
#define __logf(x) logf(x)
#define __log2f(x) log2f(x)
#define __exp2f(x) exp2f(x)
#define __sinf(x) sinf(x)
#define __cosf(x) cosf(x)
#define __powf(x, y) powf(x, y)
static inline float __fdividef(float a, float b) { return a / b; }
static inline float __frcp_rn(float a) { return 1.0f / a; }
static inline float rsqrtf(float x) { return 1.0f / sqrtf(x); }
static inline double rsqrt(double x) { return 1.0 / sqrt(x); }
static inline float __fsqrt_rn(float x) { return sqrtf(x); }
using std::max;
using std::min;
static inline long long min(long long a, int b) { return a < b ? a : b; }
static inline long long min(int a, long long b) { return a < b ? a : b; }
static inline long long max(long long a, int b) { return a > b ? a : b; }
static inline long long max(int a, long long b) { return a > b ? a : b; }

// soft-float hook for float -> __bf16 conversions (round to nearest even; what v_cvt_pk_bf16_f32 does)
extern "C" inline __bf16 __truncsfbf2(float f) {
  unsigned u;
  memcpy(&u, &f, 4);
  unsigned short h;
  if ((u & 0x7fffffffu) > 0x7f800000u) h = (unsigned short)((u >> 16) | 0x40);
  else h = (unsigned short)((u + 0x7fffu + ((u >> 16) & 1u)) >> 16);
  __bf16 r;
  memcpy(&r, &h, 2);
  return r;
}

// ---------------------------------------------------------------------------------------- cross-lane ops
#define EMU_NOINLINE __attribute__((noinline))

namespace emu {

template <typename A, typename B, typename C>
EMU_NOINLINE C mfma_f32_16x16x32_bf16(A a, B b, C c, int, int, int) {
  static_assert(sizeof(A) == 16 && sizeof(B) == 16 && sizeof(C) == 16, "operand sizes");
  struct { A a; B b; C c; } in{a, b, c};
  static_assert(offsetof(decltype(in), c) == 32 && offsetof(decltype(in), b) == 16, "layout the emulator reads");
  C out;
  wave_op(OP_MFMA_16x16x32_BF16, __builtin_return_address(0), &in, &out);
  return out;
}
template <typename A, typename B, typename C>
EMU_NOINLINE C mfma_f32_32x32x16_bf16(A a, B b, C c, int, int, int) {
  static_assert(sizeof(A) == 16 && sizeof(B) == 16 && sizeof(C) == 64, "operand sizes");
  struct { A a; B b; C c; } in{a, b, c};
  static_assert(offsetof(decltype(in), c) == 64 && offsetof(decltype(in), b) == 16, "layout the emulator reads");
  C out;
  wave_op(OP_MFMA_32x32x16_BF16, __builtin_return_address(0), &in, &out);
  return out;
}
template <typename C>
EMU_NOINLINE C mfma_f32_16x16x4f32(float a, float b, C c, int, int, int) {
  static_assert(sizeof(C) == 16, "operand sizes");
  struct { float a, b; C c; } in{a, b, c};
  static_assert(offsetof(decltype(in), c) == 16, "layout the emulator reads");
  C out;
  wave_op(OP_MFMA_16x16x4_F32, __builtin_return_address(0), &in, &out);
  return out;
}

template <typename T>
EMU_NOINLINE T shfl_any(T v, int src, int width, int mode) {
  static_assert(sizeof(T) <= 8 && std::is_trivially_copyable_v<T>, "shuffle operand");
  ShflIn in{0, src, width, mode};
  memcpy(&in.value, &v, sizeof(T));
  uint64_t out = 0;
  wave_op(OP_SHFL, __builtin_return_address(0), &in, &out);
  T r;
  memcpy(&r, &out, sizeof(T));
  return r;
}

EMU_NOINLINE inline int readfirstlane(int v) {
  uint32_t in = (uint32_t)v, out = 0;
  wave_op(OP_READFIRSTLANE, __builtin_return_address(0), &in, &out);
  return (int)out;
}
EMU_NOINLINE inline uint64_t ballot_w64(bool p) {
  int in = p ? 1 : 0;
  uint64_t out = 0;
  wave_op(OP_BALLOT, __builtin_return_address(0), &in, &out);
  return out;
}
typedef unsigned u32x2_emu __attribute__((ext_vector_type(2)));
EMU_NOINLINE inline u32x2_emu permlane32_swap(unsigned vdst, unsigned src, bool, bool) {
  unsigned in[2] = {vdst, src}, out[2] = {0, 0};
  wave_op(OP_PERMLANE32_SWAP, __builtin_return_address(0), in, out);
  return u32x2_emu{out[0], out[1]};
}
EMU_NOINLINE inline u32x2_emu permlane16_swap(unsigned vdst, unsigned src, bool, bool) {
  unsigned in[2] = {vdst, src}, out[2] = {0, 0};
  wave_op(OP_PERMLANE16_SWAP, __builtin_return_address(0), in, out);
  return u32x2_emu{out[0], out[1]};
}
typedef short i16x4_emu __attribute__((ext_vector_type(4)));
EMU_NOINLINE inline i16x4_emu ds_read_tr16_b64(const void* addr) {
  const void* in = addr;
  i16x4_emu out;
  wave_op(OP_DS_READ_TR16_B64, __builtin_return_address(0), &in, &out);
  return out;
}
// the inline-asm form: a 32-bit LDS byte address (the dynamic LDS base truncated to 32 bits + an offset)
template <typename V>
static inline void ds_read_tr16_b64_lds_addr(V& out, unsigned lds_addr, int off) {
  static_assert(sizeof(V) == 8, "ds_read_b64_tr_b16 result");
  const unsigned rel = lds_addr + (unsigned)off - (unsigned)(uintptr_t)dyn_smem();
  const i16x4_emu r = ds_read_tr16_b64(dyn_smem() + rel);
  memcpy(&out, &r, 8);
}
EMU_NOINLINE inline void wave_barrier() { wave_op(OP_WAVE_BARRIER, __builtin_return_address(0), nullptr, nullptr); }

// buffer resources: base + byte count; an access whose last byte lies past the count returns zeros
struct rsrc_t { const unsigned char* base; unsigned bytes; };
static inline rsrc_t make_buffer_rsrc(void* p, int /*stride*/, int num, int /*flags*/) {
  return rsrc_t{static_cast<const unsigned char*>(p), (unsigned)num};
}
// Range check as LLVM documents the raw buffer intrinsics: voffset (+ immediate) is checked against the record count, the
// scalar offset is EXCLUDED from the check -- an in-range voffset with a scalar offset that carries the access past the end
// reads past the end (here: into the guard page of a guarded allocation).  LDM_EMU_SOFF_CHECKED=1 switches to the other
// reading (scalar offset included in the check), to tell which kernels depend on it.
static inline bool soff_checked() {
  static const bool v = getenv("LDM_EMU_SOFF_CHECKED") != nullptr && getenv("LDM_EMU_SOFF_CHECKED")[0] == '1';
  return v;
}
static inline void raw_ptr_buffer_load_lds(rsrc_t rs, void* lds, int size, unsigned voff, int soff, int imm, int) {
  // LDS address = wave-uniform base + lane * size (the instruction writes lane-linear)
  const unsigned long long chk = (unsigned long long)voff + (unsigned)imm + (soff_checked() ? (unsigned)soff : 0u);
  const unsigned long long off = (unsigned long long)voff + (unsigned)imm + (unsigned)soff;
  unsigned char* dst = static_cast<unsigned char*>(lds) + me().lane * size;
  dma_to_lds(dst, chk + size <= rs.bytes ? rs.base + off : nullptr, size);
}
typedef unsigned u32x4_emu __attribute__((ext_vector_type(4)));
static inline u32x4_emu raw_buffer_load_b128(rsrc_t rs, int voff, int soff, int) {
  const unsigned long long chk = (unsigned long long)(unsigned)voff + (soff_checked() ? (unsigned)soff : 0u);
  const unsigned long long off = (unsigned long long)(unsigned)voff + (unsigned)soff;
  u32x4_emu r{0, 0, 0, 0};
  if (chk + 16 <= rs.bytes) memcpy(&r, rs.base + off, 16);
  return r;
}
static inline unsigned raw_buffer_load_b32(rsrc_t rs, int voff, int soff, int) {
  const unsigned long long chk = (unsigned long long)(unsigned)voff + (soff_checked() ? (unsigned)soff : 0u);
  const unsigned long long off = (unsigned long long)(unsigned)voff + (unsigned)soff;
  unsigned r = 0;
  if (chk + 4 <= rs.bytes) memcpy(&r, rs.base + off, 4);
  return r;
}

}  // namespace emu

#define __amdgpu_buffer_rsrc_t ::emu::rsrc_t
#define __builtin_amdgcn_make_buffer_rsrc ::emu::make_buffer_rsrc
#define __builtin_amdgcn_raw_ptr_buffer_load_lds ::emu::raw_ptr_buffer_load_lds
#define __builtin_amdgcn_raw_buffer_load_b128 ::emu::raw_buffer_load_b128
#define __builtin_amdgcn_raw_buffer_load_b32 ::emu::raw_buffer_load_b32
#define __builtin_amdgcn_mfma_f32_16x16x32_bf16 ::emu::mfma_f32_16x16x32_bf16
#define __builtin_amdgcn_mfma_f32_32x32x16_bf16 ::emu::mfma_f32_32x32x16_bf16
#define __builtin_amdgcn_mfma_f32_16x16x4f32 ::emu::mfma_f32_16x16x4f32
#define __builtin_amdgcn_readfirstlane ::emu::readfirstlane
#define __builtin_amdgcn_ballot_w64 ::emu::ballot_w64
#define __builtin_amdgcn_permlane32_swap ::emu::permlane32_swap
#define __builtin_amdgcn_permlane16_swap ::emu::permlane16_swap
#define __builtin_amdgcn_ds_read_tr16_b64_v4i16 ::emu::ds_read_tr16_b64
#define __builtin_amdgcn_wave_barrier ::emu::wave_barrier
#define __builtin_amdgcn_s_barrier ::emu::barrier
#define __builtin_amdgcn_sched_barrier(x) ((void)0)
#define __builtin_amdgcn_sched_group_barrier(a, b, c) ((void)0)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_s_memtime ::emu::clock_ticks
#define __builtin_amdgcn_s_sleep(n) ((void)0)
#define __builtin_amdgcn_s_memrealtime ::emu::clock_ticks
#define __builtin_amdgcn_exp2f(x) exp2f(x)
#define __builtin_amdgcn_rcpf(x) (1.0f / (x))

template <typename T>
static inline T __shfl(T v, int src, int width = 64) { return ::emu::shfl_any(v, src, width, 0); }
template <typename T>
static inline T __shfl_xor(T v, int mask, int width = 64) { return ::emu::shfl_any(v, mask, width, 1); }
template <typename T>
static inline T __shfl_down(T v, int d, int width = 64) { return ::emu::shfl_any(v, d, width, 2); }
template <typename T>
static inline T __shfl_up(T v, int d, int width = 64) { return ::emu::shfl_any(v, d, width, 3); }

static inline void __syncthreads() {
  ::emu::wait_vmcnt(0);
  ::emu::barrier();
}

// ---------------------------------------------------------------------------------------------- host API
typedef void* hipStream_t;
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
constexpr int hipFuncAttributeMaxDynamicSharedMemorySize = 8;
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline const char* hipGetErrorString(hipError_t) { return "emu"; }
static inline hipError_t hipFuncSetAttribute(const void*, int, int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
enum hipStreamCaptureStatus { hipStreamCaptureStatusNone = 0, hipStreamCaptureStatusActive = 1 };
static inline hipError_t hipStreamIsCapturing(hipStream_t, hipStreamCaptureStatus* s) { *s = hipStreamCaptureStatusNone; return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* p, int v, size_t n, hipStream_t) { memset(p, v, n); return hipSuccess; }
template <typename T>
static inline hipError_t hipGetSymbolAddress(void** p, T& sym) { *p = (void*)&sym; return hipSuccess; }

template <typename K, typename... Args>
static inline void emu_launch_kernel(K kern, dim3 grid, dim3 block, size_t smem, hipStream_t, Args... args) {
  ::emu::note_launch(reinterpret_cast<const void*>(kern));
  ::emu::launch(grid, block, smem, [&] { kern(args...); });
}
#define hipLaunchKernelGGL(kern, grid, block, smem, stream, ...) \
  emu_launch_kernel(kern, dim3(grid), dim3(block), smem, stream, ##__VA_ARGS__)
