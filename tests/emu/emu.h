// Host-side SIMT emulator for the gfx950 kernels of latent-diffusion_amd/csrc (TEST INFRASTRUCTURE: nothing under
// latent-diffusion_amd/ includes or links this; tests/emu/build_emu.py compiles the unmodified kernel sources against it).
//
// Execution model: one workgroup at a time, every thread of it a fiber with its own stack.  A fiber runs until it reaches a
// cross-lane operation (MFMA, shuffle, permlane swap, transposed LDS read, ballot, readfirstlane) or a workgroup barrier.
// When no lane of a wave can run, the lanes parked at the same operation (same kind, same call site) execute it together,
// with exactly those lanes as the EXEC mask; MFMA and ds_read_b64_tr_b16 insist on all 64 lanes, as the ISA does.
// LDS-DMA (buffer_load ... lds) is queued per lane and retired by s_waitcnt vmcnt(n) / __syncthreads, either as late as the
// counted waits allow ("lazy": a missing wait shows up as stale LDS data) or at issue ("eager": a write-after-read hazard on
// a ring slot shows up as corrupted operands).  Dynamic LDS ends on a guard page: a read or write past the size the launch
// asked for is a SIGSEGV, as is any access past a buffer from emu_alloc_guarded.
#pragma once
#include <cstddef>
#include <cstdint>
#include <functional>

namespace emu {

struct dim3 {
  unsigned x, y, z;
  constexpr dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};

struct LaneInfo {
  dim3 tid;
  int flat;   // linear thread index in the workgroup
  int lane;   // flat % 64
  int wave;   // flat / 64
};

const LaneInfo& me();
extern dim3 g_block_idx, g_block_dim, g_grid_dim;
unsigned char* dyn_smem();

enum Op : int {
  OP_MFMA_16x16x32_BF16 = 1,
  OP_MFMA_32x32x16_BF16,
  OP_MFMA_16x16x4_F32,
  OP_SHFL,
  OP_READFIRSTLANE,
  OP_BALLOT,
  OP_PERMLANE16_SWAP,
  OP_PERMLANE32_SWAP,
  OP_DS_READ_TR16_B64,
  OP_WAVE_BARRIER,
};

struct ShflIn { uint64_t value; int src; int width; int mode; };   // mode 0: index, 1: xor, 2: down, 3: up

void wave_op(Op op, const void* tag, const void* in, void* out);     // blocks the calling lane until the wave runs the op
void barrier();                                                       // s_barrier: no memory wait of its own
void dma_to_lds(void* lds_dst, const void* src_or_null, int bytes);   // one lane's share of a buffer_load ... lds
void wait_vmcnt(int n);                                               // at most n of this lane's LDS-DMAs stay in flight
void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body);
void note_launch(const void* kernel);      // per-kernel launch histogram (emu_launch_histogram)
uint64_t clock_ticks();

}  // namespace emu

extern "C" {
// 0: eager (a DMA lands when it is issued), 1: lazy (when a counted wait retires it).  Default 1.
void emu_set_dma_mode(int lazy);
// [ptr, ptr + bytes) with PROT_NONE pages on both sides; ptr + bytes is the end of the last mapped page (bytes is rounded up
// to 16 first), so the first byte past the end faults.  emu_free_guarded releases it.
void* emu_alloc_guarded(size_t bytes);
void emu_free_guarded(void* p);
long long emu_launch_count();
// "count<TAB>demangled kernel name" lines of every kernel launched since the last reset, most frequent first; returns the
// number of bytes written (the text is truncated to cap - 1 bytes and NUL-terminated).  reset != 0 clears the histogram.
long long emu_launch_histogram(char* buf, long long cap, int reset);
long long emu_waveop_count();
}
