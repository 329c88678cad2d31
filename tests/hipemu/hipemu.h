// hipemu.h — TEST INFRASTRUCTURE: run the HIP kernel sources of icl_amd/csrc on the CPU.
//
// The build container has no GPU, so every kernel in icl_amd/csrc/kernels/*.h is written
// against a small device environment (csrc/device_env_hip.h for the product).  This header
// is the second implementation of that environment: each "GPU thread" is a fiber (own stack; see "Fiber switch" below),
// a workgroup is a set of fibers stepped cooperatively, __syncthreads() and the wave-wide
// collectives (shuffles, MFMA) are rendezvous points.  Blocks run sequentially per OS
// thread and are spread over OS threads.  It is slow and only meant for tiny shapes:
// tests/ uses it to check kernel index math, LDS tiling, MFMA lane maps and the autograd
// wrappers against the oracle before GPU minutes are spent.  Never part of the product.
//
// MFMA lane maps follow /opt/skills/guides/cdna_hip_programming.md §3 (f32 16x16x4:
// A[l&15][k=l>>4], B[k=l>>4][l&15], D col=l&15,row=(l>>4)*4+reg; fmaf chain in k order).
#pragma once
#include <atomic>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <thread>
#include <type_traits>
#include <ucontext.h>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
struct float2 { float x, y; };
struct float4 { float x, y, z, w; };
struct uint4 { unsigned x, y, z, w; };
struct uint2 { unsigned x, y; };
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }
static inline uint4 make_uint4(unsigned a, unsigned b, unsigned c, unsigned d) { return uint4{a, b, c, d}; }
static inline unsigned icl_bf16_rn_bits(float f) {      // fp32 -> bf16 bits, round to nearest even; NaN stays a (quiet) NaN
  uint32_t u;
  memcpy(&u, &f, 4);
  if ((u & 0x7fffffffu) > 0x7f800000u) return (u >> 16) | 0x40u;
  return (u + 0x7fffu + ((u >> 16) & 1u)) >> 16;
}
static inline unsigned icl_pack_bf16_rn(float lo, float hi) { return icl_bf16_rn_bits(lo) | (icl_bf16_rn_bits(hi) << 16); }
static inline float icl_med3(float a, float b, float c) { return fmaxf(fminf(a, b), fminf(fmaxf(a, b), c)); }
static inline unsigned icl_alignbit(unsigned hi, unsigned lo, unsigned sh) { return (unsigned)((((uint64_t)hi << 32) | lo) >> sh); }
static inline unsigned __float_as_uint(float f) { unsigned u; memcpy(&u, &f, 4); return u; }
static inline float __uint_as_float(unsigned u) { float f; memcpy(&f, &u, 4); return f; }
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
typedef float f32x4 __attribute__((vector_size(16)));
typedef float f32x16 __attribute__((vector_size(64)));
typedef void* hipStream_t;

#define __global__
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __shared__ static thread_local
#define __launch_bounds__(...)

namespace hipemu {

constexpr int kWave = 64;
constexpr size_t kStack = 256 * 1024;

// Fiber switch.  glibc's swapcontext / getcontext save and restore the signal mask with a system call each — a quarter of the CPU suite's
// time went there (23 of ~100 CPU-minutes in the kernel).  On x86-64 the switch is the textbook one instead: push the callee-saved
// registers and the two floating-point control words, exchange stack pointers, pop, return; a new fiber's stack is laid out so that the
// first switch "returns" into its entry function.  Elsewhere: ucontext as before.
#if defined(__x86_64__)
#define HIPEMU_ASM_SWITCH 1
struct Ctx { void* sp = nullptr; };
extern "C" void hipemu_switch(Ctx* from, Ctx* to);
__asm__(
    ".text\n"
    ".globl hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "    pushq %rbp\n    pushq %rbx\n    pushq %r12\n    pushq %r13\n    pushq %r14\n    pushq %r15\n"
    "    subq $8, %rsp\n    stmxcsr (%rsp)\n    fnstcw 4(%rsp)\n"
    "    movq %rsp, (%rdi)\n"
    "    movq (%rsi), %rsp\n"
    "    ldmxcsr (%rsp)\n    fldcw 4(%rsp)\n    addq $8, %rsp\n"
    "    popq %r15\n    popq %r14\n    popq %r13\n    popq %r12\n    popq %rbx\n    popq %rbp\n"
    "    ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");
inline void ctx_switch(Ctx* from, Ctx* to) { hipemu_switch(from, to); }
inline void ctx_make(Ctx* c, char* stack, size_t size, void (*entry)()) {
  uintptr_t top = ((uintptr_t)stack + size) & ~(uintptr_t)15;
  uint64_t* sp = (uint64_t*)(top - 16);      // the return address sits 16 below a 16-byte boundary: rsp % 16 == 8 at entry, as after a call
  *sp = (uint64_t)(uintptr_t)entry;
  for (int i = 0; i < 6; ++i) *--sp = 0;      // rbp, rbx, r12..r15
  --sp;
  uint32_t* cw = (uint32_t*)sp;
  cw[0] = 0x1F80;                             // mxcsr: all exceptions masked, round to nearest
  cw[1] = 0x037F;                             // x87 control word (low 16 bits are loaded)
  c->sp = sp;
}
#else
#define HIPEMU_ASM_SWITCH 0
struct Ctx { ucontext_t uc; };
inline void ctx_switch(Ctx* from, Ctx* to) { swapcontext(&from->uc, &to->uc); }
inline void ctx_make(Ctx* c, char* stack, size_t size, void (*entry)()) {
  getcontext(&c->uc);
  c->uc.uc_stack.ss_sp = stack;
  c->uc.uc_stack.ss_size = size;
  c->uc.uc_link = nullptr;
  makecontext(&c->uc, entry, 0);
}
#endif

struct Fiber {
  Ctx ctx;
  char* stack = nullptr;
  int state = 0;  // 0 runnable, 1 wait-block, 2 wait-wave, 3 done
  dim3 tid;
};

struct Worker {
  Ctx sched;
  std::vector<Fiber> fibers;
  int cur = -1;
  int nthreads = 0;
  const std::function<void()>* body = nullptr;
  std::vector<unsigned char> dyn;
  // wave exchange slots: [wave][lane][4 dwords]
  std::vector<uint32_t> xch;
};

inline thread_local Worker* W = nullptr;
inline thread_local dim3 g_threadIdx, g_blockIdx, g_blockDim, g_gridDim;

inline void trampoline() {
  Worker* w = W;
  (*w->body)();
  w->fibers[w->cur].state = 3;
  ctx_switch(&w->fibers[w->cur].ctx, &w->sched);
  __builtin_unreachable();      // a finished fiber is never scheduled again
}

inline void yield_state(int st) {
  Worker* w = W;
  Fiber& f = w->fibers[w->cur];
  f.state = st;
  ctx_switch(&f.ctx, &w->sched);
  g_threadIdx = f.tid;  // restored by scheduler too; belt and braces
}

inline void run_block(Worker* w, dim3 bid, dim3 block) {
  const int n = block.x * block.y * block.z;
  w->nthreads = n;
  if ((int)w->fibers.size() < n) {
    size_t old = w->fibers.size();
    w->fibers.resize(n);
    for (size_t i = old; i < (size_t)n; ++i) w->fibers[i].stack = (char*)malloc(kStack);
  }
  w->xch.assign((size_t)((n + kWave - 1) / kWave) * kWave * 4, 0u);
  g_blockIdx = bid;
  for (int i = 0; i < n; ++i) {
    Fiber& f = w->fibers[i];
    f.state = 0;
    f.tid = dim3(i % block.x, (i / block.x) % block.y, i / (block.x * block.y));
    ctx_make(&f.ctx, f.stack, kStack, trampoline);
  }
  for (;;) {
    bool progressed = false;
    int done = 0;
    for (int i = 0; i < n; ++i) {
      Fiber& f = w->fibers[i];
      if (f.state == 3) { ++done; continue; }
      if (f.state != 0) continue;
      w->cur = i;
      g_threadIdx = f.tid;
      ctx_switch(&w->sched, &f.ctx);
      progressed = true;
    }
    if (done == n) break;
    // release barriers whose participants have all arrived
    int live = 0, atblk = 0;
    for (int i = 0; i < n; ++i) {
      if (w->fibers[i].state != 3) ++live;
      if (w->fibers[i].state == 1) ++atblk;
    }
    if (live > 0 && atblk == live) {
      for (int i = 0; i < n; ++i) if (w->fibers[i].state == 1) w->fibers[i].state = 0;
      progressed = true;
    }
    for (int wv = 0; wv * kWave < n; ++wv) {
      int lo = wv * kWave, hi = std::min(n, lo + kWave), lv = 0, at = 0;
      for (int i = lo; i < hi; ++i) {
        if (w->fibers[i].state != 3) ++lv;
        if (w->fibers[i].state == 2) ++at;
      }
      if (lv > 0 && at == lv) {
        for (int i = lo; i < hi; ++i) if (w->fibers[i].state == 2) w->fibers[i].state = 0;
        progressed = true;
      }
    }
    if (!progressed) {
      fprintf(stderr, "hipemu: deadlock in block (%u,%u,%u): divergent barrier\n", bid.x, bid.y, bid.z);
      abort();
    }
  }
}

inline int emu_threads() {
  const char* e = getenv("HIPEMU_THREADS");
  int n = e ? atoi(e) : (int)std::thread::hardware_concurrency();
  return n < 1 ? 1 : n;
}

inline void launch(dim3 grid, dim3 block, size_t dyn_lds, const std::function<void()>& body) {
  const long nblk = (long)grid.x * grid.y * grid.z;
  if (nblk == 0) return;
  std::atomic<long> next{0};
  auto work = [&]() {
    Worker w;
    W = &w;
    w.body = &body;
    w.dyn.assign(dyn_lds + 64, 0);
    g_blockDim = block;
    g_gridDim = grid;
    for (;;) {
      long b = next.fetch_add(1);
      if (b >= nblk) break;
      dim3 bid((unsigned)(b % grid.x), (unsigned)((b / grid.x) % grid.y), (unsigned)(b / ((long)grid.x * grid.y)));
      run_block(&w, bid, block);
    }
    for (auto& f : w.fibers) free(f.stack);
    W = nullptr;
  };
  int nt = (int)std::min<long>(emu_threads(), nblk);
  if (nt <= 1) { work(); return; }
  std::vector<std::thread> th;
  for (int i = 0; i < nt; ++i) th.emplace_back(work);
  for (auto& t : th) t.join();
}

inline void* dyn_lds() {
  uintptr_t p = (uintptr_t)W->dyn.data();
  return (void*)((p + 63) & ~(uintptr_t)63);
}

inline int lane_linear() {
  const dim3& t = g_threadIdx;
  return (int)(t.x + g_blockDim.x * (t.y + g_blockDim.y * t.z));
}

// wave-wide exchange of up to 4 dwords per lane: publish, rendezvous, read peer, rendezvous
inline void wave_publish(const uint32_t* v, int nd) {
  int lin = lane_linear();
  memcpy(&W->xch[(size_t)lin * 4], v, nd * 4);
  yield_state(2);
}
inline const uint32_t* wave_peer(int src_lane) {
  int lin = lane_linear();
  int base = (lin / kWave) * kWave;
  return &W->xch[(size_t)(base + src_lane) * 4];
}
inline void wave_done() { yield_state(2); }

}  // namespace hipemu

#define threadIdx (hipemu::g_threadIdx)
#define blockIdx (hipemu::g_blockIdx)
#define blockDim (hipemu::g_blockDim)
#define gridDim (hipemu::g_gridDim)

static inline void __syncthreads() { hipemu::yield_state(1); }

template <typename T>
static inline T hipemu_shfl_from(T v, int src) {
  static_assert(sizeof(T) == 4, "4-byte shuffles only");
  uint32_t u;
  memcpy(&u, &v, 4);
  hipemu::wave_publish(&u, 1);
  uint32_t r = hipemu::wave_peer(src & 63)[0];
  hipemu::wave_done();
  T out;
  memcpy(&out, &r, 4);
  return out;
}
static inline int hipemu_lane() { return hipemu::lane_linear() & 63; }
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) {
  int l = hipemu_lane();
  return hipemu_shfl_from(v, (l & ~(width - 1)) | ((l ^ m) & (width - 1)));
}
template <typename T> static inline T __shfl_down(T v, int d, int width = 64) {
  int l = hipemu_lane();
  int s = (l & (width - 1)) + d;
  return hipemu_shfl_from(v, s < width ? l + d : l);
}
template <typename T> static inline T __shfl(T v, int src, int width = 64) {
  int l = hipemu_lane();
  return hipemu_shfl_from(v, (l & ~(width - 1)) | (src & (width - 1)));
}

// ds_read_b64_tr_b16 (device_env_hip.h): lane i of a 16-lane group receives, for q = 0..3, the 16-bit element i & 3 of the 8 bytes
// addressed by lane 4q + (i >> 2) of its group
static inline uint2 icl_lds_read_tr16_b64(const void* lds_ptr) {
  uint64_t a = (uint64_t)(uintptr_t)lds_ptr;
  uint32_t pub[2] = {(uint32_t)a, (uint32_t)(a >> 32)};
  hipemu::wave_publish(pub, 2);
  const int l = hipemu_lane(), g0 = l & ~15, i = l & 15;
  uint16_t e[4];
  for (int q = 0; q < 4; ++q) {
    const uint32_t* pa = hipemu::wave_peer(g0 + 4 * q + (i >> 2));
    const uint16_t* src = reinterpret_cast<const uint16_t*>((uintptr_t)(((uint64_t)pa[1] << 32) | pa[0]));
    e[q] = src[i & 3];
  }
  hipemu::wave_done();
  return make_uint2((uint32_t)e[0] | ((uint32_t)e[1] << 16), (uint32_t)e[2] | ((uint32_t)e[3] << 16));
}
struct icl_rsrc_t { const unsigned char* p; unsigned bytes; };
static inline icl_rsrc_t icl_make_rsrc(const void* p, unsigned bytes) { return icl_rsrc_t{(const unsigned char*)p, bytes}; }
static inline float icl_buffer_load_f32(icl_rsrc_t r, unsigned byte_off) {
  if ((uint64_t)byte_off + 4 > r.bytes) return 0.f;
  float f;
  memcpy(&f, r.p + byte_off, 4);
  return f;
}
static inline float icl_buffer_load_f32(icl_rsrc_t r, unsigned byte_off, unsigned uniform_off) {
  if ((uint64_t)byte_off + 4 > r.bytes) return 0.f;
  if ((uint64_t)byte_off + uniform_off + 4 > r.bytes) { fprintf(stderr, "hipemu: buffer load past the descriptor through the scalar offset\n"); abort(); }
  float f;
  memcpy(&f, r.p + byte_off + uniform_off, 4);
  return f;
}
static inline uint4 icl_buffer_load_u32x4(icl_rsrc_t r, unsigned byte_off) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if ((uint64_t)byte_off + 16 <= r.bytes) memcpy(&v, r.p + byte_off, 16);
  return v;
}
static inline uint4 icl_buffer_load_u32x4(icl_rsrc_t r, unsigned byte_off, unsigned uniform_off) {
  uint4 v = make_uint4(0u, 0u, 0u, 0u);
  if ((uint64_t)byte_off + 16 > r.bytes) return v;
  if ((uint64_t)byte_off + uniform_off + 16 > r.bytes) { fprintf(stderr, "hipemu: buffer load past the descriptor through the scalar offset\n"); abort(); }
  memcpy(&v, r.p + byte_off + uniform_off, 16);
  return v;
}
// LDS-DMA (device_env_hip.h): each lane's 16 bytes land at lds_wave_base + 16 lane.  A lane outside the extent aborts: the kernels
// must point such lanes at their zero block (the product does not rely on the hardware's range-check result for LDS-DMA).
static inline void icl_buffer_load_lds_b128(icl_rsrc_t r, void* lds_wave_base, unsigned byte_off, unsigned uniform_off) {
  if ((uint64_t)byte_off + uniform_off + 16 > r.bytes) { fprintf(stderr, "hipemu: LDS-DMA source outside the descriptor\n"); abort(); }
  memcpy((unsigned char*)lds_wave_base + 16 * hipemu_lane(), r.p + byte_off + uniform_off, 16);
}
#define ICL_WAIT_VMEM() ((void)0)
#define ICL_WAIT_VMCNT(n) ((void)0)
#define ICL_BARRIER_KEEP_VMEM() __syncthreads()
static inline float icl_fast_exp(float x) { return expf(x); }
static inline float4 icl_nt_load4(const float* p) { return *reinterpret_cast<const float4*>(p); }
static inline void icl_nt_store4(float* p, float4 v) { *reinterpret_cast<float4*>(p) = v; }
#define ICL_OPAQUE_INT(x) ((void)(x))
#define ICL_SCHED_BARRIER() ((void)0)
#define ICL_SCHED_GROUP(mask, n) ((void)0)
#define ICL_PIN4(u) ((void)(u))
#define ICL_PIN1(f) ((void)(f))
#define ICL_SETPRIO(p) ((void)0)
#define ICL_WAVE_UNIFORM(x) ((void)(x))
#define ICL_WAVE_SYNC() hipemu::yield_state(2)
static inline float atomicAdd(float* p, float v) {
  uint32_t* ip = (uint32_t*)p;
  uint32_t old = __atomic_load_n(ip, __ATOMIC_RELAXED), neu;
  float f;
  do {
    memcpy(&f, &old, 4);
    f += v;
    memcpy(&neu, &f, 4);
  } while (!__atomic_compare_exchange_n(ip, &old, neu, false, __ATOMIC_RELAXED, __ATOMIC_RELAXED));
  memcpy(&f, &old, 4);
  return f;
}
static inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
static inline unsigned atomicExch(unsigned* p, unsigned v) { return __atomic_exchange_n(p, v, __ATOMIC_RELAXED); }
static inline void __threadfence() { __atomic_thread_fence(__ATOMIC_SEQ_CST); }

// v_mfma_f32_16x16x4_f32: D(16x16) = A(16x4) * B(4x16) + C, exact fmaf chain over k = 0..3
static inline f32x4 icl_mfma_16x16x4(float a, float b, f32x4 c) {
  uint32_t u[2];
  memcpy(&u[0], &a, 4);
  memcpy(&u[1], &b, 4);
  hipemu::wave_publish(u, 2);
  int l = hipemu_lane();
  int col = l & 15;
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    int row = (l >> 4) * 4 + r;
    float acc = c[r];
    for (int k = 0; k < 4; ++k) {
      float av, bv;
      memcpy(&av, &hipemu::wave_peer(k * 16 + row)[0], 4);
      memcpy(&bv, &hipemu::wave_peer(k * 16 + col)[1], 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  hipemu::wave_done();
  return d;
}

// v_mfma_f32_16x16x32_bf16: lane l holds A[row l&15][k 8*(l>>4)+j], B[k 8*(l>>4)+j][col l&15] as 8 packed bf16 (low half of a dword =
// even element); D as icl_mfma_16x16x4.  Products of two bf16 are exact in fp32; the 32 products of one instruction are summed
// here in double and rounded to fp32 once together with C (the hardware's internal order is not documented; tests hold tolerances).
static inline f32x4 icl_mfma_16x16x32_bf16(uint4 a, uint4 b, f32x4 c) {
  uint32_t ua[4] = {a.x, a.y, a.z, a.w}, ub[4] = {b.x, b.y, b.z, b.w};
  const int l = hipemu_lane(), col = l & 15;
  uint32_t arow[4][4][4], bcol[4][4];      // [r][k block][dword], [k block][dword]
  hipemu::wave_publish(ua, 4);
  for (int r = 0; r < 4; ++r)
    for (int kb = 0; kb < 4; ++kb) memcpy(arow[r][kb], hipemu::wave_peer(kb * 16 + (l >> 4) * 4 + r), 16);
  hipemu::wave_done();
  hipemu::wave_publish(ub, 4);
  for (int kb = 0; kb < 4; ++kb) memcpy(bcol[kb], hipemu::wave_peer(kb * 16 + col), 16);
  hipemu::wave_done();
  auto bf = [](uint32_t dw, int odd) { uint32_t u = odd ? (dw & 0xffff0000u) : (dw << 16); float f; memcpy(&f, &u, 4); return f; };
  f32x4 d = c;
  for (int r = 0; r < 4; ++r) {
    double acc = c[r];
    for (int kb = 0; kb < 4; ++kb)
      for (int j = 0; j < 8; ++j) acc += (double)bf(arow[r][kb][j >> 1], j & 1) * (double)bf(bcol[kb][j >> 1], j & 1);
    d[r] = (float)acc;
  }
  return d;
}

// v_mfma_f32_32x32x2_f32: lane l holds A[i=l&31][k=l>>5], B[k=l>>5][j=l&31];
// D: col = l&31, row = (reg&3) + 8*(reg>>2) + 4*(l>>5)
static inline f32x16 icl_mfma_32x32x2(float a, float b, f32x16 c) {
  uint32_t u[2];
  memcpy(&u[0], &a, 4);
  memcpy(&u[1], &b, 4);
  hipemu::wave_publish(u, 2);
  int l = hipemu_lane();
  int col = l & 31;
  f32x16 d = c;
  for (int r = 0; r < 16; ++r) {
    int row = (r & 3) + 8 * (r >> 2) + 4 * (l >> 5);
    float acc = c[r];
    for (int k = 0; k < 2; ++k) {
      float av, bv;
      memcpy(&av, &hipemu::wave_peer(k * 32 + row)[0], 4);
      memcpy(&bv, &hipemu::wave_peer(k * 32 + col)[1], 4);
      acc = fmaf(av, bv, acc);
    }
    d[r] = acc;
  }
  hipemu::wave_done();
  return d;
}

#define ICL_DYN_LDS(type, name) type* name = (type*)hipemu::dyn_lds()
#define ICL_LAUNCH(kern, grid, block, lds, stream, ...) \
  hipemu::launch((grid), (block), (lds), [=]() { kern(__VA_ARGS__); })
#define ICL_MEMSET_ASYNC(ptr, val, bytes, stream) ((void)memset((ptr), (val), (bytes)))
#define ICL_SYMBOL_ADDRESS(pp, sym) (*(pp) = (void*)(sym), 0)
#define ICL_LAST_LAUNCH_ERROR() 0
#define ICL_ERROR_STRING(e) "hipemu"
#define ICL_SET_MAX_DYN_LDS(kern, bytes) ((void)0)
