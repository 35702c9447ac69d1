// Hardware probe (round 3, first GPU call): is the SCALAR offset of a raw buffer load part of the range check on gfx950?
// LLVM documents it as excluded; the attention kernels of rounds 1-2 relied on it being included (rows past lk of a ragged last
// key tile reading zeros) until the host emulator's strict reading flagged it.  The kernels no longer
// depend on the answer; this prints it so that the emulator's default can follow the hardware.
//   hipcc --offload-arch=gfx950 -O2 tools/probes/soffset_range_check.hip -o /tmp/soff && /tmp/soff
// The buffer holds 1024 dwords of 1.0f inside a 2048-dword allocation whose second half holds 2.0f; the descriptor covers the
// first 4096 bytes.  Lane l reads dword l with voffset = 4 l and soffset = 4096: in range by voffset alone, out of range once
// the scalar offset is added.  Output 0.0 -> the scalar offset IS checked; 2.0 -> it is not (the load went past num_records).
#include <hip/hip_runtime.h>
#include <stdio.h>

__global__ void probe(const float* base, float* out) {
  const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(base), 0, 4096, 0x00020000);
  out[threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(threadIdx.x * 4), 4096, 0));
  out[64 + threadIdx.x] = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, (int)(threadIdx.x * 4 + 4096), 0, 0));
}

int main() {
  float *buf, *out, h[2048], r[128];
  for (int i = 0; i < 2048; ++i) h[i] = i < 1024 ? 1.0f : 2.0f;
  if (hipMalloc(&buf, sizeof h) != hipSuccess || hipMalloc(&out, sizeof r) != hipSuccess) return 1;
  hipMemcpy(buf, h, sizeof h, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, buf, out);
  if (hipMemcpy(r, out, sizeof r, hipMemcpyDeviceToHost) != hipSuccess) return 2;
  printf("voffset in range + soffset past the end -> %.1f  (0.0: soffset is range-checked, 2.0: it is not)\n", r[0]);
  printf("voffset past the end, soffset 0          -> %.1f  (control: must be 0.0)\n", r[64]);
  return 0;
}
