// Kernels that exercise ONE emulated operation each (tests/test_emu_selftest_cpu.py compares them with numpy): they pin the
// emulator's reading of the operand layouts and of the LDS-DMA / wait counters, independently of the product kernels.
// Compiled only into libldm_emu.so (tests/emu/build_emu.py); written in the same HIP subset as latent-diffusion_amd/csrc.
#include <hip/hip_runtime.h>

typedef __attribute__((ext_vector_type(8))) short bf16x8_t;
typedef __attribute__((ext_vector_type(4))) short bf16x4_t;
typedef __attribute__((ext_vector_type(4))) float f32x4_t;
typedef __attribute__((ext_vector_type(16))) float f32x16_t;

// A [16][32] bf16 row-major, B [32][16] bf16 row-major, C/D [16][16] float
__global__ void st_mfma16_kernel(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
  const int l = threadIdx.x;
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)A[(l % 16) * 32 + (l / 16) * 8 + j];
    b[j] = (short)B[((l / 16) * 8 + j) * 16 + l % 16];
  }
  f32x4_t c;
  for (int r = 0; r < 4; ++r) c[r] = C[((l / 16) * 4 + r) * 16 + l % 16];
  c = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l / 16) * 4 + r) * 16 + l % 16] = c[r];
}

// A [32][16], B [16][32], C/D [32][32]
__global__ void st_mfma32_kernel(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
  const int l = threadIdx.x;
  bf16x8_t a, b;
  for (int j = 0; j < 8; ++j) {
    a[j] = (short)A[(l % 32) * 16 + (l / 32) * 8 + j];
    b[j] = (short)B[((l / 32) * 8 + j) * 32 + l % 32];
  }
  f32x16_t c;
  for (int r = 0; r < 16; ++r) c[r] = C[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32];
  c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) D[(8 * (r / 4) + 4 * (l / 32) + r % 4) * 32 + l % 32] = c[r];
}

// A [16][4] float, B [4][16] float, C/D [16][16]
__global__ void st_mfma_f32_kernel(const float* A, const float* B, const float* C, float* D) {
  const int l = threadIdx.x;
  f32x4_t c;
  for (int r = 0; r < 4; ++r) c[r] = C[((l / 16) * 4 + r) * 16 + l % 16];
  c = __builtin_amdgcn_mfma_f32_16x16x4f32(A[(l % 16) * 4 + l / 16], B[(l / 16) * 16 + l % 16], c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) D[((l / 16) * 4 + r) * 16 + l % 16] = c[r];
}

// src [16 rows][16 cols] of 16-bit values per 16-lane group (4 groups: [4][4][16] ... laid out [64 rows][16 cols]); the
// transposed read of group g takes rows 4g..4g+3: out[lane][q] = src[4g + q][lane % 16]
__global__ void st_tr16_kernel(const unsigned short* src, unsigned short* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  unsigned short* s = reinterpret_cast<unsigned short*>(smem);
  const int l = threadIdx.x;
  for (int i = l; i < 16 * 16; i += 64) s[i] = src[i];
  __syncthreads();
  const int g = l / 16, i = l % 16, q = i / 4, p = i % 4;
  const unsigned char* addr = smem + ((4 * g + q) * 16 + 4 * p) * 2;        // row q of the group's block, columns 4p..4p+3
  const bf16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) bf16x4_t*)addr);
  for (int e = 0; e < 4; ++e) out[l * 4 + e] = (unsigned short)v[e];
}

// out[0][lane], out[1][lane] = v_permlane32_swap(vdst = lane, src = 100 + lane); out[2], out[3] the same for permlane16
__global__ void st_permlane_kernel(unsigned* out) {
  const unsigned l = threadIdx.x;
  const auto a = __builtin_amdgcn_permlane32_swap(l, 100u + l, false, false);
  const auto b = __builtin_amdgcn_permlane16_swap(l, 100u + l, false, false);
  out[l] = a[0];
  out[64 + l] = a[1];
  out[128 + l] = b[0];
  out[192 + l] = b[1];
}

// Two LDS-DMA loads of 1 KiB each (16 B per lane) from src into LDS slots 0 and 1, then vmcnt(1): the FIRST has landed, the
// second may not have (it has not in the lazy model, it has in the eager one).  out[0..255] = slot 0, out[256..511] = slot 1
// as seen right after the counted wait; the LDS image starts as 0xA5 bytes.  A third load with an out-of-range voffset must
// deliver zeros (out[512..767], after vmcnt(0)).
__global__ void st_dma_kernel(const unsigned* src, unsigned bytes, unsigned* out) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int l = threadIdx.x;
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned*>(src), 0, (int)bytes, 0x00020000);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)smem, 16, l * 16, 0, 0, 0);
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 1024), 16, l * 16, 1024, 0, 0);
  asm volatile("s_waitcnt vmcnt(%0)" ::"n"(1) : "memory");
  const unsigned* s = reinterpret_cast<const unsigned*>(smem);
  for (int i = 0; i < 4; ++i) {
    out[l * 4 + i] = s[l * 4 + i];
    out[256 + l * 4 + i] = s[256 + l * 4 + i];
  }
  __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(smem + 2048), 16, 0x80000000u + l * 16, 0, 0, 0);
  asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
  for (int i = 0; i < 4; ++i) out[512 + l * 4 + i] = s[512 + l * 4 + i];
}

extern "C" {
void emu_selftest_mfma16(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
  hipLaunchKernelGGL(st_mfma16_kernel, dim3(1), dim3(64), 0, nullptr, A, B, C, D);
}
void emu_selftest_mfma32(const unsigned short* A, const unsigned short* B, const float* C, float* D) {
  hipLaunchKernelGGL(st_mfma32_kernel, dim3(1), dim3(64), 0, nullptr, A, B, C, D);
}
void emu_selftest_mfma_f32(const float* A, const float* B, const float* C, float* D) {
  hipLaunchKernelGGL(st_mfma_f32_kernel, dim3(1), dim3(64), 0, nullptr, A, B, C, D);
}
void emu_selftest_tr16(const unsigned short* src, unsigned short* out) {
  hipLaunchKernelGGL(st_tr16_kernel, dim3(1), dim3(64), 16 * 16 * 2, nullptr, src, out);
}
void emu_selftest_permlane(unsigned* out) { hipLaunchKernelGGL(st_permlane_kernel, dim3(1), dim3(64), 0, nullptr, out); }
void emu_selftest_dma(const unsigned* src, unsigned bytes, unsigned* out) {
  hipLaunchKernelGGL(st_dma_kernel, dim3(1), dim3(64), 3072, nullptr, src, bytes, out);
}
}
