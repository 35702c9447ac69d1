// Access-pattern probe for the scheduler step (3 float32 streams in, 2 out) on DRAM-resident, rotated operand sets.
// Build: hipcc -O3 --offload-arch=gfx950 -ffp-contract=off tools/probes/stream_bench.hip -o gpurun_out/stream_bench
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef __attribute__((ext_vector_type(4))) float v4f;
__device__ __forceinline__ float4 ntload(const float4* p) { const v4f v = __builtin_nontemporal_load(reinterpret_cast<const v4f*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
__device__ __forceinline__ void ntstore(const float4& a, float4* p) { __builtin_nontemporal_store(v4f{a.x, a.y, a.z, a.w}, reinterpret_cast<v4f*>(p)); }
struct K { float sigma, a_sqrt, c1, c2, stdv; };
__device__ __forceinline__ void step4(const float4& e, const float4& x, const float4& z, float4& p, float4& o, const K& k) {
  const float ee[4] = {e.x, e.y, e.z, e.w}, xx[4] = {x.x, x.y, x.z, x.w}, zz[4] = {z.x, z.y, z.z, z.w};
  float pp[4], oo[4];
#pragma unroll
  for (int j = 0; j < 4; ++j) {
    const float x0 = (xx[j] - k.sigma * ee[j]) / k.a_sqrt;
    pp[j] = k.c1 * x0 + k.c2 * xx[j] + k.stdv * zz[j];
    oo[j] = x0;
  }
  p = make_float4(pp[0], pp[1], pp[2], pp[3]);
  o = make_float4(oo[0], oo[1], oo[2], oo[3]);
}
// V0: the product kernel's pattern: a thread owns 8 consecutive floats (two float4 32 B apart between lanes), grid-stride
__global__ void __launch_bounds__(256) v0(const float4* e, const float4* x, const float4* z, float4* p, float4* o, long n4, K k) {
  const long nv = n4 >> 1;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < nv; i += (long)gridDim.x * 256) {
    float4 p0, o0, p1, o1;
    const float4 e0 = e[2 * i], e1 = e[2 * i + 1], x0 = x[2 * i], x1 = x[2 * i + 1], z0 = z[2 * i], z1 = z[2 * i + 1];
    step4(e0, x0, z0, p0, o0, k); step4(e1, x1, z1, p1, o1, k);
    p[2 * i] = p0; p[2 * i + 1] = p1; o[2 * i] = o0; o[2 * i + 1] = o1;
  }
}
// V1: lane-contiguous float4 (1 KiB per wave-instruction), U independent chunks per thread per iteration, grid-stride
template <int U, bool NT>
__global__ void __launch_bounds__(256) v1(const float4* e, const float4* x, const float4* z, float4* p, float4* o, long n4, K k) {
  const long span = (long)gridDim.x * 256;
  for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n4; i += span * U) {
    float4 ev[U], xv[U], zv[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * span;
      if (j < n4) {
        if (NT) { ev[u] = ntload(e + j); xv[u] = ntload(x + j); zv[u] = ntload(z + j); }
        else { ev[u] = e[j]; xv[u] = x[j]; zv[u] = z[j]; }
      }
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
      const long j = i + u * span;
      if (j < n4) {
        float4 pp, oo;
        step4(ev[u], xv[u], zv[u], pp, oo, k);
        if (NT) { ntstore(pp, p + j); ntstore(oo, o + j); }
        else { p[j] = pp; o[j] = oo; }
      }
    }
  }
}
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main() {
  const long n = 256L * 4 * 64 * 64, n4 = n / 4;
  const int NSET = 12;
  std::vector<float*> bufs;
  for (int s = 0; s < NSET * 5; ++s) { float* b; CK(hipMalloc(&b, n * 4)); CK(hipMemset(b, 0x3c, n * 4)); bufs.push_back(b); }
  K k{0.6f, 0.8f, 0.3f, 0.7f, 0.1f};
  hipEvent_t a, b; CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
  auto run = [&](const char* name, auto launch) -> int {
    for (int it = 0; it < 3; ++it) launch(it % NSET);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    const int iters = 36;
    for (int it = 0; it < iters; ++it) launch(it % NSET);
    CK(hipEventRecord(b)); CK(hipEventSynchronize(b));
    float ms; CK(hipEventElapsedTime(&ms, a, b));
    const double us = ms * 1e3 / iters;
    printf("%-44s %7.2f us  %7.1f GB/s\n", name, us, 5.0 * n * 4 / us / 1e3);
    return 0;
  };
#define ARGS(s) (const float4*)bufs[5 * (s)], (const float4*)bufs[5 * (s) + 1], (const float4*)bufs[5 * (s) + 2], (float4*)bufs[5 * (s) + 3], (float4*)bufs[5 * (s) + 4], n4, k
  for (int g : {1024, 2048, 4096, 16384}) {
    char nm[96];
    snprintf(nm, 96, "v0 8-per-thread            grid %5d", g); if (run(nm, [&](int s) { hipLaunchKernelGGL(v0, dim3(g), dim3(256), 0, 0, ARGS(s)); })) return 1;
    snprintf(nm, 96, "v1 contiguous U=1          grid %5d", g); if (run(nm, [&](int s) { hipLaunchKernelGGL((v1<1, false>), dim3(g), dim3(256), 0, 0, ARGS(s)); })) return 1;
    snprintf(nm, 96, "v1 contiguous U=2          grid %5d", g); if (run(nm, [&](int s) { hipLaunchKernelGGL((v1<2, false>), dim3(g), dim3(256), 0, 0, ARGS(s)); })) return 1;
    snprintf(nm, 96, "v1 contiguous U=4          grid %5d", g); if (run(nm, [&](int s) { hipLaunchKernelGGL((v1<4, false>), dim3(g), dim3(256), 0, 0, ARGS(s)); })) return 1;
    snprintf(nm, 96, "v1 contiguous U=2 nt       grid %5d", g); if (run(nm, [&](int s) { hipLaunchKernelGGL((v1<2, true>), dim3(g), dim3(256), 0, 0, ARGS(s)); })) return 1;
    snprintf(nm, 96, "v1 contiguous U=4 nt       grid %5d", g); if (run(nm, [&](int s) { hipLaunchKernelGGL((v1<4, true>), dim3(g), dim3(256), 0, 0, ARGS(s)); })) return 1;
  }
  return 0;
}
