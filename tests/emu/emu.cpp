// Host-side SIMT emulator: scheduler, cross-lane operations, LDS-DMA queue, guarded allocations.  See emu.h.
#include "emu.h"

#include <cxxabi.h>
#include <dlfcn.h>
#include <execinfo.h>
#include <signal.h>
#include <sys/mman.h>
#include <unistd.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <algorithm>
#include <map>
#include <string>
#include <vector>

namespace emu {

dim3 g_block_idx, g_block_dim, g_grid_dim;

namespace {

enum State { READY, WAVEOP, BARRIER, DONE };

struct Dma { void* dst; const void* src; int bytes; };

// Context switch: callee-saved registers on the fiber's own stack, the stack pointer in the Fiber (ucontext's swapcontext
// costs two sigprocmask system calls per switch; a wave operation switches 128 times).
extern "C" void emu_switch(void** save_sp, void* load_sp);
asm(R"(
.text
.globl emu_switch
.type emu_switch,@function
emu_switch:
  pushq %rbp
  pushq %rbx
  pushq %r12
  pushq %r13
  pushq %r14
  pushq %r15
  subq $8, %rsp
  stmxcsr (%rsp)
  fnstcw 4(%rsp)
  movq %rsp, (%rdi)
  movq %rsi, %rsp
  ldmxcsr (%rsp)
  fldcw 4(%rsp)
  addq $8, %rsp
  popq %r15
  popq %r14
  popq %r13
  popq %r12
  popq %rbx
  popq %rbp
  ret
.size emu_switch,.-emu_switch
)");

struct Fiber {
  void* sp = nullptr;
  LaneInfo info;
  State state = DONE;
  Op op;
  const void* tag;
  const void* in;
  void* out;
  std::deque<Dma> dma;
  unsigned char* stack = nullptr;
};

constexpr size_t STACK_BYTES = 512 * 1024;
constexpr int MAX_THREADS = 1024;

std::vector<Fiber> g_fibers;
void* g_main_sp = nullptr;
Fiber* g_cur = nullptr;
const std::function<void()>* g_body = nullptr;
unsigned char* g_smem_map = nullptr;
size_t g_smem_map_bytes = 0;
unsigned char* g_smem = nullptr;
int g_dma_lazy = 1;
long long g_launches = 0, g_waveops = 0;
uint64_t g_clock = 0;

[[noreturn]] void die(const char* what) {
  fprintf(stderr, "emu: %s (block %u,%u thread %d)\n", what, g_block_idx.x, g_block_idx.y, g_cur ? g_cur->info.flat : -1);
  fflush(stderr);
  abort();
}

void yield_to_main() { emu_switch(&g_cur->sp, g_main_sp); }

void do_dma(const Dma& d) {
  if (d.src) memcpy(d.dst, d.src, d.bytes);
  else memset(d.dst, 0, d.bytes);
}

void trampoline() {
  (*g_body)();
  Fiber* f = g_cur;
  while (!f->dma.empty()) { do_dma(f->dma.front()); f->dma.pop_front(); }
  f->state = DONE;
  yield_to_main();
  die("resumed a finished fiber");
}

inline float bf16_bits_to_f32(uint16_t v) {
  uint32_t u = (uint32_t)v << 16;
  float f;
  memcpy(&f, &u, 4);
  return f;
}

// sum_k a[k] * b[k] over n (a multiple of 8) floats: eight partial sums (the compiler turns them into one vector FMA per
// step), folded pairwise, then the accumulator is added -- float32 throughout, like the matrix core's accumulation
inline float dot_f32(const float* a, const float* b, int n, float c) {
  float p[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  for (int k = 0; k < n; k += 8)
    for (int j = 0; j < 8; ++j) p[j] += a[k + j] * b[k + j];
  return c + (((p[0] + p[4]) + (p[2] + p[6])) + ((p[1] + p[5]) + (p[3] + p[7])));
}

struct Mfma16In { alignas(16) uint16_t a[8]; alignas(16) uint16_t b[8]; alignas(16) float c[4]; };
struct Mfma32In { uint16_t a[8], b[8]; alignas(64) float c[16]; };   // as the shim's { bf16x8 a, b; f32x16 c; }
struct MfmaF32In { float a, b; alignas(16) float c[4]; };   // as the shim's { float a, b; f32x4 c; }

void need_full(int n, const char* what) {
  if (n != 64) die(what);
}

// lanes[0..n) are the fibers parked at this operation, in lane order
void run_op(Op op, Fiber** lanes, int n) {
  g_waveops++;
  switch (op) {
    case OP_MFMA_16x16x32_BF16: {
      need_full(n, "v_mfma_f32_16x16x32_bf16 with a partial EXEC mask");
      alignas(32) static float A[16][32], Bt[16][32];
      for (int l = 0; l < 64; ++l) {
        const Mfma16In* in = static_cast<const Mfma16In*>(lanes[l]->in);
        for (int j = 0; j < 8; ++j) {
          A[l % 16][(l / 16) * 8 + j] = bf16_bits_to_f32(in->a[j]);
          Bt[l % 16][(l / 16) * 8 + j] = bf16_bits_to_f32(in->b[j]);
        }
      }
      for (int l = 0; l < 64; ++l) {
        const Mfma16In* in = static_cast<const Mfma16In*>(lanes[l]->in);
        float* out = static_cast<float*>(lanes[l]->out);
        const int col = l % 16;
        for (int r = 0; r < 4; ++r) out[r] = dot_f32(A[(l / 16) * 4 + r], Bt[col], 32, in->c[r]);
      }
      break;
    }
    case OP_MFMA_32x32x16_BF16: {
      need_full(n, "v_mfma_f32_32x32x16_bf16 with a partial EXEC mask");
      alignas(32) static float A[32][16], Bt[32][16];
      for (int l = 0; l < 64; ++l) {
        const Mfma32In* in = static_cast<const Mfma32In*>(lanes[l]->in);
        for (int j = 0; j < 8; ++j) {
          A[l % 32][(l / 32) * 8 + j] = bf16_bits_to_f32(in->a[j]);
          Bt[l % 32][(l / 32) * 8 + j] = bf16_bits_to_f32(in->b[j]);
        }
      }
      for (int l = 0; l < 64; ++l) {
        const Mfma32In* in = static_cast<const Mfma32In*>(lanes[l]->in);
        float* out = static_cast<float*>(lanes[l]->out);
        const int col = l % 32;
        for (int r = 0; r < 16; ++r) out[r] = dot_f32(A[8 * (r / 4) + 4 * (l / 32) + (r % 4)], Bt[col], 16, in->c[r]);
      }
      break;
    }
    case OP_MFMA_16x16x4_F32: {
      need_full(n, "v_mfma_f32_16x16x4_f32 with a partial EXEC mask");
      static float A[16][4], B[4][16];
      for (int l = 0; l < 64; ++l) {
        const MfmaF32In* in = static_cast<const MfmaF32In*>(lanes[l]->in);
        A[l % 16][l / 16] = in->a;
        B[l / 16][l % 16] = in->b;
      }
      for (int l = 0; l < 64; ++l) {
        const MfmaF32In* in = static_cast<const MfmaF32In*>(lanes[l]->in);
        float* out = static_cast<float*>(lanes[l]->out);
        const int col = l % 16;
        for (int r = 0; r < 4; ++r) {
          const int row = (l / 16) * 4 + r;
          float s = in->c[r];
          for (int k = 0; k < 4; ++k) s = fmaf(A[row][k], B[k][col], s);
          out[r] = s;
        }
      }
      break;
    }
    case OP_SHFL: {
      const Fiber* by_lane[64] = {};
      for (int i = 0; i < n; ++i) by_lane[lanes[i]->info.lane] = lanes[i];
      for (int i = 0; i < n; ++i) {
        const ShflIn* in = static_cast<const ShflIn*>(lanes[i]->in);
        const int l = lanes[i]->info.lane;
        const int w = in->width;
        const int base = l / w * w;
        int s;
        if (in->mode == 0) s = base + ((in->src % w) + w) % w;
        else if (in->mode == 1) s = base + ((l - base) ^ in->src);
        else if (in->mode == 2) s = (l - base) + in->src < w ? l + in->src : l;
        else s = (l - base) - in->src >= 0 ? l - in->src : l;
        if (s < 0 || s >= 64 || (in->mode == 1 && s >= base + w)) s = l;
        if (!by_lane[s]) die("shuffle reads a lane that is not active at this call");
        *static_cast<uint64_t*>(lanes[i]->out) = static_cast<const ShflIn*>(by_lane[s]->in)->value;
      }
      break;
    }
    case OP_READFIRSTLANE: {
      const uint32_t v = *static_cast<const uint32_t*>(lanes[0]->in);
      for (int i = 0; i < n; ++i) *static_cast<uint32_t*>(lanes[i]->out) = v;
      break;
    }
    case OP_BALLOT: {
      uint64_t m = 0;
      for (int i = 0; i < n; ++i)
        if (*static_cast<const int*>(lanes[i]->in)) m |= 1ull << lanes[i]->info.lane;
      for (int i = 0; i < n; ++i) *static_cast<uint64_t*>(lanes[i]->out) = m;
      break;
    }
    case OP_PERMLANE32_SWAP: {
      // in: {vdst, src}; lanes 32..63 of vdst swap with lanes 0..31 of src
      need_full(n, "v_permlane32_swap with a partial EXEC mask (emulator is strict)");
      uint32_t vd[64], sr[64];
      for (int l = 0; l < 64; ++l) {
        const uint32_t* in = static_cast<const uint32_t*>(lanes[l]->in);
        vd[l] = in[0];
        sr[l] = in[1];
      }
      for (int l = 0; l < 64; ++l) {
        uint32_t* out = static_cast<uint32_t*>(lanes[l]->out);
        out[0] = l < 32 ? vd[l] : sr[l - 32];
        out[1] = l < 32 ? vd[l + 32] : sr[l];
      }
      break;
    }
    case OP_PERMLANE16_SWAP: {
      // rows of 16 lanes: odd rows of vdst swap with even rows of src (row 1 <-> row 0, row 3 <-> row 2)
      need_full(n, "v_permlane16_swap with a partial EXEC mask (emulator is strict)");
      uint32_t vd[64], sr[64];
      for (int l = 0; l < 64; ++l) {
        const uint32_t* in = static_cast<const uint32_t*>(lanes[l]->in);
        vd[l] = in[0];
        sr[l] = in[1];
      }
      for (int l = 0; l < 64; ++l) {
        uint32_t* out = static_cast<uint32_t*>(lanes[l]->out);
        const int row = l / 16;
        out[0] = (row & 1) ? sr[l - 16] : vd[l];
        out[1] = (row & 1) ? sr[l] : vd[l + 16];
      }
      break;
    }
    case OP_DS_READ_TR16_B64: {
      // per 16-lane group: lane 4q+p supplies the address of row q, columns 4p..4p+3; lane i receives column i, row q in
      // element q
      need_full(n, "ds_read_b64_tr_b16 with a partial EXEC mask");
      const unsigned char* addr[64];
      for (int l = 0; l < 64; ++l) {
        addr[l] = *static_cast<const unsigned char* const*>(lanes[l]->in);
        if (reinterpret_cast<uintptr_t>(addr[l]) & 7) die("ds_read_b64_tr_b16: address not 8-byte aligned");
      }
      for (int l = 0; l < 64; ++l) {
        uint16_t* out = static_cast<uint16_t*>(lanes[l]->out);
        const int g = l / 16, i = l % 16;
        for (int q = 0; q < 4; ++q) memcpy(&out[q], addr[16 * g + 4 * q + i / 4] + 2 * (i % 4), 2);
      }
      break;
    }
    case OP_WAVE_BARRIER:
      break;
    default:
      die("unknown wave operation");
  }
}

void run_block() {
  const int nthreads = (int)(g_block_dim.x * g_block_dim.y * g_block_dim.z);
  for (int t = 0; t < nthreads; ++t) {
    Fiber& f = g_fibers[t];
    f.info.flat = t;
    f.info.lane = t % 64;
    f.info.wave = t / 64;
    f.info.tid = dim3(t % g_block_dim.x, (t / g_block_dim.x) % g_block_dim.y, t / (g_block_dim.x * g_block_dim.y));
    f.state = READY;
    f.dma.clear();
    // initial frame for emu_switch: [mxcsr|fpcw] r15 r14 r13 r12 rbx rbp <return address = trampoline>, 16-byte aligned so
    // that the stack pointer is 8 modulo 16 on entry to trampoline, as after a call
    {
      uintptr_t top = reinterpret_cast<uintptr_t>(f.stack + STACK_BYTES) & ~uintptr_t(15);
      uint64_t* spp = reinterpret_cast<uint64_t*>(top);
      *--spp = 0;                                             // fake return address of trampoline (never returns)
      *--spp = reinterpret_cast<uint64_t>(&trampoline);       // ret target of the first switch
      for (int k = 0; k < 6; ++k) *--spp = 0;                 // rbp rbx r12 r13 r14 r15
      uint32_t csr[2] = {0x1F80u, 0x037Fu};                   // default MXCSR, default x87 control word
      --spp;
      memcpy(spp, csr, 8);
      f.sp = spp;
    }
  }
  const int nwaves = (nthreads + 63) / 64;
  for (;;) {
    bool ran = false;
    for (int t = 0; t < nthreads; ++t) {
      Fiber& f = g_fibers[t];
      if (f.state != READY) continue;
      g_cur = &f;
      emu_switch(&g_main_sp, f.sp);
      g_cur = nullptr;
      ran = true;
    }
    bool did = false;
    for (int w = 0; w < nwaves; ++w) {
      const int lo = w * 64, hi = std::min(nthreads, lo + 64);
      // partition the parked lanes by (op, call site)
      {   // the common case: the whole wave parked at one operation
        bool same = hi - lo == 64 && g_fibers[lo].state == WAVEOP;
        for (int t = lo + 1; same && t < hi; ++t)
          same = g_fibers[t].state == WAVEOP && g_fibers[t].op == g_fibers[lo].op && g_fibers[t].tag == g_fibers[lo].tag;
        if (same) {
          Fiber* all[64];
          for (int t = lo; t < hi; ++t) all[t - lo] = &g_fibers[t];
          run_op(g_fibers[lo].op, all, 64);
          for (int t = lo; t < hi; ++t) g_fibers[t].state = READY;
          did = true;
          continue;
        }
      }
      std::map<std::pair<int, const void*>, std::vector<Fiber*>> groups;
      for (int t = lo; t < hi; ++t)
        if (g_fibers[t].state == WAVEOP) groups[{(int)g_fibers[t].op, g_fibers[t].tag}].push_back(&g_fibers[t]);
      for (auto& kv : groups) {
        run_op((Op)kv.first.first, kv.second.data(), (int)kv.second.size());
        for (Fiber* f : kv.second) f->state = READY;
        did = true;
      }
    }
    if (did) continue;
    int at_barrier = 0, done = 0;
    for (int t = 0; t < nthreads; ++t) {
      at_barrier += g_fibers[t].state == BARRIER;
      done += g_fibers[t].state == DONE;
    }
    if (done == nthreads) break;
    if (at_barrier + done == nthreads && at_barrier > 0) {
      for (int t = 0; t < nthreads; ++t)
        if (g_fibers[t].state == BARRIER) g_fibers[t].state = READY;
      continue;
    }
    if (!ran) die("deadlock: no lane can run");
  }
}

// SIGSEGV report: which lane of which workgroup touched what (a guard page behind LDS or behind a guarded buffer), with a
// backtrace into the kernel source (-g1 line tables: addr2line -e libldm_emu.so <pc>)
void on_segv(int sig, siginfo_t* si, void*) {
  char buf[512];
  const unsigned char* a = static_cast<const unsigned char*>(si->si_addr);
  const char* where = "outside the emulator's regions";
  long rel = 0;
  if (g_smem_map && a >= g_smem_map && a < g_smem_map + g_smem_map_bytes) {
    where = "LDS image (offset from the dynamic LDS base)";
    rel = (long)(a - g_smem);
  }
  int n = snprintf(buf, sizeof buf, "emu: SIGSEGV at %p: %s %ld; block (%u,%u,%u) thread %d\n", (void*)a, where, rel,
                   g_block_idx.x, g_block_idx.y, g_block_idx.z, g_cur ? g_cur->info.flat : -1);
  if (write(2, buf, n)) {}
  void* bt[32];
  const int m = backtrace(bt, 32);
  backtrace_symbols_fd(bt, m, 2);
  signal(sig, SIG_DFL);
  raise(sig);
}

void install_segv_handler() {
  static bool done = false;
  if (done) return;
  done = true;
  static unsigned char alt[64 * 1024];
  stack_t ss{};
  ss.ss_sp = alt;
  ss.ss_size = sizeof alt;
  sigaltstack(&ss, nullptr);
  struct sigaction sa{};
  sa.sa_sigaction = on_segv;
  sa.sa_flags = SA_SIGINFO | SA_ONSTACK | SA_RESETHAND;
  sigaction(SIGSEGV, &sa, nullptr);
}

void ensure_fibers(int n) {
  if ((int)g_fibers.size() < n) {
    const size_t old = g_fibers.size();
    g_fibers.resize(n);
    for (size_t i = old; i < g_fibers.size(); ++i) {
      void* m = mmap(nullptr, STACK_BYTES + 4096, PROT_READ | PROT_WRITE, MAP_PRIVATE | MAP_ANONYMOUS | MAP_NORESERVE, -1, 0);
      if (m == MAP_FAILED) die("mmap of a fiber stack failed");
      mprotect(m, 4096, PROT_NONE);    // stack overflow = fault, not silent corruption
      g_fibers[i].stack = static_cast<unsigned char*>(m) + 4096;
    }
  }
}

}  // namespace

static std::map<const void*, long long> g_hist;
void note_launch(const void* kernel) { g_hist[kernel]++; }

const LaneInfo& me() { return g_cur->info; }
unsigned char* dyn_smem() { return g_smem; }
uint64_t clock_ticks() { return ++g_clock; }

void wave_op(Op op, const void* tag, const void* in, void* out) {
  Fiber* f = g_cur;
  f->op = op;
  f->tag = tag;
  f->in = in;
  f->out = out;
  f->state = WAVEOP;
  yield_to_main();
}

void barrier() {
  g_cur->state = BARRIER;
  yield_to_main();
}

void dma_to_lds(void* lds_dst, const void* src_or_null, int bytes) {
  const Dma d{lds_dst, src_or_null, bytes};
  if (g_dma_lazy) g_cur->dma.push_back(d);
  else do_dma(d);
}

void wait_vmcnt(int n) {
  Fiber* f = g_cur;
  while ((int)f->dma.size() > n) {
    do_dma(f->dma.front());
    f->dma.pop_front();
  }
}

void launch(dim3 grid, dim3 block, size_t dyn_smem_bytes, const std::function<void()>& body) {
  const int nthreads = (int)(block.x * block.y * block.z);
  if (nthreads < 1 || nthreads > MAX_THREADS) die("bad workgroup size");
  if (dyn_smem_bytes > 160 * 1024) die("more than 160 KiB of dynamic LDS");
  ensure_fibers(nthreads);
  if (getenv("LDM_EMU_SEGV_REPORT")) install_segv_handler();
  // dynamic LDS: ends on a guard page
  const size_t need = (dyn_smem_bytes + 15) & ~size_t(15);
  const size_t pages = (need + 4095) / 4096 * 4096;
  if (g_smem_map_bytes < pages + 8192) {
    if (g_smem_map) munmap(g_smem_map, g_smem_map_bytes);
    g_smem_map_bytes = 160 * 1024 + 4096 + 8192;
    g_smem_map = static_cast<unsigned char*>(mmap(nullptr, g_smem_map_bytes, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
    if (g_smem_map == MAP_FAILED) die("mmap of the LDS image failed");
  }
  mprotect(g_smem_map, g_smem_map_bytes, PROT_NONE);
  if (pages) mprotect(g_smem_map + 4096, pages, PROT_READ | PROT_WRITE);
  g_smem = g_smem_map + 4096 + pages - need;
  if (need) memset(g_smem, 0xA5, need);      // LDS starts out as garbage, not zeros
  g_grid_dim = grid;
  g_block_dim = block;
  g_body = &body;
  g_launches++;
  for (unsigned z = 0; z < grid.z; ++z)
    for (unsigned y = 0; y < grid.y; ++y)
      for (unsigned x = 0; x < grid.x; ++x) {
        g_block_idx = dim3(x, y, z);
        run_block();
      }
  g_body = nullptr;
}

}  // namespace emu

extern "C" {

void emu_set_dma_mode(int lazy) { emu::g_dma_lazy = lazy ? 1 : 0; }

struct GuardHeader { void* map; size_t map_bytes; };
static std::map<void*, GuardHeader> g_guarded;

void* emu_alloc_guarded(size_t bytes) {
  const size_t need = (bytes + 15) & ~size_t(15);
  const size_t pages = (need + 4095) / 4096 * 4096;
  const size_t total = pages + 2 * 4096;
  unsigned char* m = static_cast<unsigned char*>(mmap(nullptr, total, PROT_NONE, MAP_PRIVATE | MAP_ANONYMOUS, -1, 0));
  if (m == MAP_FAILED) return nullptr;
  if (pages) mprotect(m + 4096, pages, PROT_READ | PROT_WRITE);
  unsigned char* p = m + 4096 + pages - need;
  g_guarded[p] = GuardHeader{m, total};
  return p;
}

void emu_free_guarded(void* p) {
  auto it = g_guarded.find(p);
  if (it == g_guarded.end()) return;
  munmap(it->second.map, it->second.map_bytes);
  g_guarded.erase(it);
}

long long emu_launch_count() { return emu::g_launches; }

long long emu_launch_histogram(char* buf, long long cap, int reset) {
  std::vector<std::pair<long long, std::string>> rows;
  for (const auto& kv : emu::g_hist) {
    Dl_info info{};
    std::string name = "?";
    if (dladdr(kv.first, &info) && info.dli_sname) {
      int status = 0;
      char* dm = abi::__cxa_demangle(info.dli_sname, nullptr, nullptr, &status);
      name = status == 0 && dm ? dm : info.dli_sname;
      free(dm);
    }
    rows.emplace_back(kv.second, name);
  }
  std::sort(rows.begin(), rows.end(), [](const auto& a, const auto& b) { return a.first != b.first ? a.first > b.first : a.second < b.second; });
  std::string text;
  for (const auto& r : rows) text += std::to_string(r.first) + "\t" + r.second + "\n";
  if (reset) emu::g_hist.clear();
  if (cap <= 0) return 0;
  const long long n = std::min<long long>((long long)text.size(), cap - 1);
  memcpy(buf, text.data(), n);
  buf[n] = 0;
  return n;
}
long long emu_waveop_count() { return emu::g_waveops; }

}
