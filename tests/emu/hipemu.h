// hipemu.h — a tiny single-threaded emulator of the HIP execution model, for DEVELOPMENT TESTS ONLY.
//
// It lets the .hip kernel sources of this repo be compiled by g++ and stepped on a CPU-only box so that
// kernel LOGIC (indexing, barriers, wave ballots/shuffles, look-back protocols) can be debugged without
// spending GPU minutes. It is NOT a product path and NOT a fallback: the product library
// (libmot_hip.so, built by hipcc for gfx950) never contains it, mot_create() fails without a GPU, and no
// parity claim rests on it — the parity tests are the `-m gpu` tests that run the real kernels.
//
// Model: blocks run one after another; the threads of a block are ucontext fibers scheduled round-robin;
// __syncthreads() and the wave-collective operations (64-lane waves) are rendezvous points.
// A fiber runs until its next rendezvous, so in the default (ascending) order a thread always sees what lower-numbered
// threads wrote before THEIR next rendezvous — a missing barrier between such a write and read goes unnoticed. The
// environment variable MOT_EMU_SCHED changes the order in which the fibers of a block are resumed: "rev" (descending) or
// "rand:<seed>" (a fresh permutation on every sweep). A kernel whose result depends on that order has a race (or relies on
// intra-wave lockstep without a wave-level synchronisation); the emulator tests are run in all three modes. Because IEEE
// fp32/fp64 +,-,*,/,sqrt are identical on x86-64 (no FMA contraction) and gfx950 (-ffp-contract=off,
// correctly rounded divide/sqrt), arithmetic results under emulation equal the device's.
#ifndef HIPEMU_H_
#define HIPEMU_H_
#ifndef MOT_HIPEMU
#define MOT_HIPEMU 1
#endif

#include <ucontext.h>
// Fiber switch. glibc's swapcontext saves and restores the signal mask with two system calls per switch — a rendezvous of a 512-thread
// workgroup is a thousand switches, and a third of the CPU suite's time was spent in rt_sigprocmask. On x86-64, outside the AddressSanitizer
// build (whose runtime must see fiber switches: it intercepts swapcontext), the switch is the six callee-saved registers, the two
// floating-point control words and the stack pointer.
#if defined(__x86_64__) && !defined(__SANITIZE_ADDRESS__) && !defined(HIPEMU_UCONTEXT)
#define HIPEMU_FAST_SWITCH 1
extern "C" void hipemu_switch(void** save_sp, void* const* load_sp);
__asm__(
    ".text\n"
    ".p2align 4\n"
    ".weak hipemu_switch\n"
    ".type hipemu_switch,@function\n"
    "hipemu_switch:\n"
    "  pushq %rbp\n  pushq %rbx\n  pushq %r12\n  pushq %r13\n  pushq %r14\n  pushq %r15\n"
    "  subq $8, %rsp\n  stmxcsr (%rsp)\n  fnstcw 4(%rsp)\n"
    "  movq %rsp, (%rdi)\n"
    "  movq (%rsi), %rsp\n"
    "  ldmxcsr (%rsp)\n  fldcw 4(%rsp)\n  addq $8, %rsp\n"
    "  popq %r15\n  popq %r14\n  popq %r13\n  popq %r12\n  popq %rbx\n  popq %rbp\n"
    "  ret\n"
    ".size hipemu_switch,.-hipemu_switch\n");
#endif
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <algorithm>
#include <functional>
#include <vector>

#define __global__
#define __device__
#define __host__
// every __shared__ variable lands in one ELF section, which launch() fills with 0xFF before each workgroup starts: on the GPU
// LDS holds whatever the previous workgroup left there, so a kernel that reads LDS it has not written must not get zeros here
#define __shared__ __attribute__((section("mot_lds"))) static
extern "C" __attribute__((visibility("hidden"))) char __start_mot_lds[];
extern "C" __attribute__((visibility("hidden"))) char __stop_mot_lds[];
#define __constant__ static const
#define __forceinline__ inline
#define __launch_bounds__(...)
#define __restrict__

struct dim3 { unsigned x, y, z; dim3(unsigned a = 1, unsigned b = 1, unsigned c = 1) : x(a), y(b), z(c) {} };
struct float4 { float x, y, z, w; };
struct float2 { float x, y; };
struct int2 { int x, y; };
struct int4 { int x, y, z, w; };
struct uint2 { unsigned x, y; };
struct uint4 { unsigned x, y, z, w; };
struct double2 { double x, y; };
static inline float4 make_float4(float a, float b, float c, float d) { return float4{a, b, c, d}; }
static inline float2 make_float2(float a, float b) { return float2{a, b}; }
static inline int2 make_int2(int a, int b) { return int2{a, b}; }
static inline int4 make_int4(int a, int b, int c, int d) { return int4{a, b, c, d}; }
static inline uint2 make_uint2(unsigned a, unsigned b) { return uint2{a, b}; }

typedef int hipError_t;
enum { hipSuccess = 0, hipErrorInvalidValue = 1, hipErrorNoDevice = 100 };
typedef void* hipStream_t;
typedef struct hipemu_event { double t; }* hipEvent_t;
enum hipMemcpyKind { hipMemcpyHostToHost, hipMemcpyHostToDevice, hipMemcpyDeviceToHost, hipMemcpyDeviceToDevice, hipMemcpyDefault };
struct hipDeviceProp_t { char name[256]; char gcnArchName[256]; int multiProcessorCount; size_t totalGlobalMem; };

#include <time.h>
static inline long long hipemu_now_ns();
namespace hipemu {
constexpr int kWave = 64;
constexpr size_t kStack = 128 * 1024;

struct Thread {
#ifdef HIPEMU_FAST_SWITCH
  void* ctx = nullptr;   // saved stack pointer
#else
  ucontext_t ctx;
#endif
  dim3 tid;
  int lin = 0;     // linear thread id in block
  bool done = false;
};
struct Wave {
  int alive = 0, arrived = 0, gen = 0;
  uint64_t buf[2][kWave];
  uint64_t pred[2];
};
struct State {
  dim3 grid, block, bid;
  int nthreads = 0, alive = 0, bar_count = 0, bar_gen = 0;
  std::vector<Thread> threads;
  std::vector<Wave> waves;
  std::vector<char> stacks;
#ifdef HIPEMU_FAST_SWITCH
  void* sched = nullptr;
#else
  ucontext_t sched;
#endif
  Thread* cur = nullptr;
  std::function<void()> body;
};
inline State& S() { static State s; return s; }

#ifdef HIPEMU_FAST_SWITCH
#define HIPEMU_SWAP(from, to) hipemu_switch(&(from), &(to))
#else
#define HIPEMU_SWAP(from, to) swapcontext(&(from), &(to))
#endif
inline void yield() { State& s = S(); HIPEMU_SWAP(s.cur->ctx, s.sched); }

inline void release_barrier_if_complete(State& s) {
  if (s.bar_count > 0 && s.bar_count >= s.alive) { s.bar_count = 0; s.bar_gen++; }
}
inline void release_wave_if_complete(Wave& w) {
  if (w.arrived > 0 && w.arrived >= w.alive) { w.arrived = 0; w.gen++; }
}
inline void trampoline() {
  State& s = S();
  s.body();
  s.cur->done = true;
  s.alive--;
  Wave& w = s.waves[s.cur->lin / kWave];
  w.alive--;
  release_barrier_if_complete(s);
  release_wave_if_complete(w);
  HIPEMU_SWAP(s.cur->ctx, s.sched);
}

inline void syncthreads() {
  State& s = S();
  int gen = s.bar_gen;
  s.bar_count++;
  release_barrier_if_complete(s);
  while (s.bar_gen == gen) yield();
}

// rendezvous of the live lanes of the calling wave; returns the generation's buffer index
inline int wave_rendezvous(Wave& w) {
  int gen = w.gen;
  w.arrived++;
  release_wave_if_complete(w);
  while (w.gen == gen) yield();
  return gen & 1;
}
inline Wave& my_wave() { State& s = S(); return s.waves[s.cur->lin / kWave]; }
inline int lane() { return S().cur->lin % kWave; }

inline uint64_t ballot(bool p) {
  Wave& w = my_wave();
  int b = w.gen & 1;
  if (w.arrived == 0) w.pred[b] = 0;
  if (p) w.pred[b] |= (1ull << lane());
  int r = wave_rendezvous(w);
  return w.pred[r];
}
template <typename T>
inline T shfl(T v, int src) {
  static_assert(sizeof(T) <= 8, "shfl type");
  Wave& w = my_wave();
  int b = w.gen & 1;
  uint64_t raw = 0; memcpy(&raw, &v, sizeof(T));
  w.buf[b][lane()] = raw;
  State& s = S();
  int base = (s.cur->lin / kWave) * kWave;
  int srcl = src & (kWave - 1);
  // a lane that has exited or does not exist returns the caller's own value. Decided BEFORE the rendezvous: a lane that
  // takes part in this exchange may run on and finish before the others get to read its slot.
  const bool absent = base + srcl >= s.nthreads || s.threads[base + srcl].done;
  int r = wave_rendezvous(w);
  if (absent) return v;
  T out; memcpy(&out, &w.buf[r][srcl], sizeof(T));
  return out;
}

// 0 = ascending, 1 = descending, 2 = random (seeded)
struct Sched { int mode = 0; uint64_t rng = 0x9E3779B97F4A7C15ull; };
inline Sched& sched_cfg() {
  static Sched c = [] {
    Sched k;
    const char* e = getenv("MOT_EMU_SCHED");
    if (e && !strncmp(e, "rev", 3)) k.mode = 1;
    else if (e && !strncmp(e, "rand", 4)) { k.mode = 2; const char* c = strchr(e, ':'); if (c) k.rng ^= strtoull(c + 1, nullptr, 10) * 0xD1342543DE82EF95ull + 1; }
    return k;
  }();
  return c;
}
inline uint64_t sched_next(Sched& c) { c.rng ^= c.rng << 13; c.rng ^= c.rng >> 7; c.rng ^= c.rng << 17; return c.rng; }

template <typename F>
inline void launch(dim3 grid, dim3 block, F&& f) {
  State& s = S();
  s.grid = grid; s.block = block;
  s.nthreads = (int)(block.x * block.y * block.z);
  if ((int)s.threads.size() < s.nthreads) s.threads.resize(s.nthreads);
  if (s.stacks.size() < (size_t)s.nthreads * kStack) s.stacks.resize((size_t)s.nthreads * kStack);
  int nw = (s.nthreads + kWave - 1) / kWave;
  s.waves.assign(nw, Wave());
  s.body = f;
  // workgroups run one after another; MOT_EMU_SCHED=rev / rand also reverses / shuffles the order in which they are started (the
  // GPU promises none: a kernel whose workgroups wait for each other by blockIdx instead of by an arrival ticket would hang here)
  const unsigned long nblocks = (unsigned long)grid.x * grid.y * grid.z;
  std::vector<unsigned long> border(nblocks);
  for (unsigned long i = 0; i < nblocks; i++) border[i] = sched_cfg().mode == 1 ? nblocks - 1 - i : i;
  if (sched_cfg().mode == 2) for (unsigned long i = nblocks; i > 1; i--) std::swap(border[i - 1], border[sched_next(sched_cfg()) % i]);
  for (unsigned long bi = 0; bi < nblocks; bi++) {
      {
        const unsigned bx = (unsigned)(border[bi] % grid.x), by = (unsigned)((border[bi] / grid.x) % grid.y), bz = (unsigned)(border[bi] / ((unsigned long)grid.x * grid.y));
        s.bid = dim3(bx, by, bz);
        memset(__start_mot_lds, 0xFF, (size_t)(__stop_mot_lds - __start_mot_lds));   // LDS arrives dirty
        s.alive = s.nthreads; s.bar_count = 0; s.bar_gen = 0;
        for (int w = 0; w < nw; w++) { s.waves[w] = Wave(); s.waves[w].alive = std::min(kWave, s.nthreads - w * kWave); }
        for (int t = 0; t < s.nthreads; t++) {
          Thread& th = s.threads[t];
          th.lin = t; th.done = false;
          th.tid = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
#ifdef HIPEMU_FAST_SWITCH
          {   // a fresh fiber: the frame hipemu_switch pops — control words, six registers, then `ret` into the trampoline with the stack as after a call
            uintptr_t top = (uintptr_t)(s.stacks.data() + (size_t)(t + 1) * kStack) & ~(uintptr_t)15;
            void** sp = (void**)(top - 72);
            const uint32_t cw[2] = {0x1F80u, 0x037Fu};   // MXCSR and x87 control word at their defaults
            memcpy(&sp[0], cw, 8);
            for (int r = 1; r <= 6; r++) sp[r] = nullptr;
            sp[7] = (void*)(void (*)())trampoline;          // at top - 16; the trampoline then runs with rsp = top - 8 (it never returns)
            sp[8] = nullptr;
            th.ctx = (void*)sp;
          }
#else
          getcontext(&th.ctx);
          th.ctx.uc_stack.ss_sp = s.stacks.data() + (size_t)t * kStack;
          th.ctx.uc_stack.ss_size = kStack;
          th.ctx.uc_link = &s.sched;
          makecontext(&th.ctx, (void (*)())trampoline, 0);
#endif
        }
        int live = s.nthreads;
        long spins = 0;
        Sched& sc = sched_cfg();
        std::vector<int> order(s.nthreads);
        for (int t = 0; t < s.nthreads; t++) order[t] = sc.mode == 1 ? s.nthreads - 1 - t : t;
        while (live > 0) {
          live = 0;
          if (sc.mode == 2) for (int t = s.nthreads - 1; t > 0; t--) std::swap(order[t], order[sched_next(sc) % (uint64_t)(t + 1)]);
          for (int k = 0; k < s.nthreads; k++) {
            const int t = order[k];
            Thread& th = s.threads[t];
            if (th.done) continue;
            s.cur = &th;
            HIPEMU_SWAP(s.sched, th.ctx);
            if (!th.done) live++;
          }
          if (++spins > 50000000L) { fprintf(stderr, "hipemu: block (%u,%u,%u) appears deadlocked\n", bx, by, bz); abort(); }
        }
      }
  }
  s.cur = nullptr;
}
}  // namespace hipemu

#define threadIdx (hipemu::S().cur->tid)
#define blockIdx (hipemu::S().bid)
#define blockDim (hipemu::S().block)
#define gridDim (hipemu::S().grid)
#define warpSize 64

#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  hipemu::launch((grid), (block), [=]() { kernel(__VA_ARGS__); })

static inline void __syncthreads() { hipemu::syncthreads(); }
static inline unsigned long long __ballot(int p) { return hipemu::ballot(p != 0); }
static inline int __any(int p) { return hipemu::ballot(p != 0) != 0; }
static inline int __all(int p) { hipemu::State& s = hipemu::S(); (void)s; unsigned long long b = hipemu::ballot(p != 0); unsigned long long a = hipemu::ballot(true); return b == a; }
template <typename T> static inline T __shfl(T v, int src, int width = 64) { (void)width; return hipemu::shfl(v, src); }
template <typename T> static inline T __shfl_xor(T v, int m, int width = 64) { (void)width; return hipemu::shfl(v, hipemu::lane() ^ m); }
template <typename T> static inline T __shfl_up(T v, unsigned d, int width = 64) { (void)width; int l = hipemu::lane(); return hipemu::shfl(v, l >= (int)d ? l - (int)d : l); }
template <typename T> static inline T __shfl_down(T v, unsigned d, int width = 64) { (void)width; int l = hipemu::lane(); return hipemu::shfl(v, l + (int)d < 64 ? l + (int)d : l); }
static inline int __popcll(unsigned long long v) { return __builtin_popcountll(v); }
static inline int __popc(unsigned v) { return __builtin_popcount(v); }
static inline int __ffsll(unsigned long long v) { return __builtin_ffsll((long long)v); }
static inline int __ffs(unsigned v) { return __builtin_ffs((int)v); }
static inline int __clz(int v) { return v ? __builtin_clz((unsigned)v) : 32; }
static inline int __clzll(long long v) { return v ? __builtin_clzll((unsigned long long)v) : 64; }
static inline unsigned __builtin_amdgcn_readlane(unsigned v, int l) { return hipemu::shfl(v, l); }
static inline unsigned __builtin_amdgcn_readfirstlane(unsigned v) { return hipemu::shfl(v, 0); }
static inline long long clock64() { return (long long)(hipemu_now_ns()); }
static inline int __lane_id() { return hipemu::lane(); }
static inline unsigned __builtin_amdgcn_mbcnt_lo(unsigned mask, unsigned base) { int l = hipemu::lane(); unsigned m = l >= 32 ? mask : (mask & ((1u << l) - 1)); return base + __builtin_popcount(m); }
static inline unsigned __builtin_amdgcn_mbcnt_hi(unsigned mask, unsigned base) { int l = hipemu::lane(); unsigned m = l <= 32 ? 0u : (mask & ((1u << (l - 32)) - 1)); return base + __builtin_popcount(m); }
static inline int __float_as_int(float f) { int i; memcpy(&i, &f, 4); return i; }
static inline float __int_as_float(int i) { float f; memcpy(&f, &i, 4); return f; }
static inline unsigned __float_as_uint(float f) { unsigned i; memcpy(&i, &f, 4); return i; }
static inline float __uint_as_float(unsigned i) { float f; memcpy(&f, &i, 4); return f; }
static inline long long __double_as_longlong(double d) { long long i; memcpy(&i, &d, 8); return i; }
static inline double __longlong_as_double(long long i) { double d; memcpy(&d, &i, 8); return d; }
static inline void __threadfence() {}
static inline void __threadfence_block() {}
static inline unsigned long long __umul64hi(unsigned long long a, unsigned long long b) { return (unsigned long long)(((unsigned __int128)a * b) >> 64); }
static inline int min(int a, int b) { return a < b ? a : b; }
static inline int max(int a, int b) { return a > b ? a : b; }
static inline void __builtin_amdgcn_s_sleep(int) { if (hipemu::S().cur) hipemu::yield(); }

// atomics (single OS thread: plain read-modify-write)
template <typename T> static inline T atomicAdd(T* p, T v) { T o = *p; *p = o + v; return o; }
template <typename T> static inline T atomicSub(T* p, T v) { T o = *p; *p = o - v; return o; }
template <typename T> static inline T atomicMin(T* p, T v) { T o = *p; if (v < o) *p = v; return o; }
template <typename T> static inline T atomicMax(T* p, T v) { T o = *p; if (v > o) *p = v; return o; }
template <typename T> static inline T atomicOr(T* p, T v) { T o = *p; *p = o | v; return o; }
template <typename T> static inline T atomicAnd(T* p, T v) { T o = *p; *p = o & v; return o; }
template <typename T> static inline T atomicExch(T* p, T v) { T o = *p; *p = v; return o; }
template <typename T> static inline T atomicCAS(T* p, T cmp, T v) { T o = *p; if (o == cmp) *p = v; return o; }
#define __ATOMIC_RELAXED_ 0
#define __HIP_MEMORY_SCOPE_AGENT 4
#define __HIP_MEMORY_SCOPE_WORKGROUP 2
#define __HIP_MEMORY_SCOPE_SYSTEM 5
#define __hip_atomic_load(p, order, scope) (*(p))
#define __hip_atomic_store(p, v, order, scope) (*(p) = (v))
#define __hip_atomic_fetch_add(p, v, order, scope) atomicAdd((p), (v))
#define __hip_atomic_fetch_min(p, v, order, scope) atomicMin((p), (v))
#define __hip_atomic_fetch_or(p, v, order, scope) atomicOr((p), (v))

// ---- host runtime (memory is plain host memory) ----
static inline const char* hipGetErrorString(hipError_t e) { return e == hipSuccess ? "hipSuccess" : "hipemu error"; }
static inline hipError_t hipGetLastError() { return hipSuccess; }
static inline hipError_t hipGetDeviceCount(int* n) { *n = 1; return hipSuccess; }
static inline hipError_t hipSetDevice(int) { return hipSuccess; }
static inline hipError_t hipGetDevice(int* d) { *d = 0; return hipSuccess; }
static inline hipError_t hipGetDeviceProperties(hipDeviceProp_t* p, int) { memset(p, 0, sizeof *p); strcpy(p->name, "hipemu"); strcpy(p->gcnArchName, "gfx950:emulated"); p->multiProcessorCount = 256; return hipSuccess; }
// device memory arrives poisoned (0xFF: NaN floats, -1 ints) so that reads of never-written memory show up in the tests
static inline hipError_t hipMalloc(void** p, size_t n) {   // 256-byte aligned like the device allocator (types with alignas(64) live in these buffers)
  const size_t m = ((n ? n : 1) + 255) & ~(size_t)255;
  *p = aligned_alloc(256, m); if (*p) memset(*p, 0xFF, m); return *p ? hipSuccess : 2; }
template <typename T> static inline hipError_t hipMalloc(T** p, size_t n) { return hipMalloc((void**)p, n); }
static inline hipError_t hipHostMalloc(void** p, size_t n, unsigned = 0) { return hipMalloc(p, n); }
template <typename T> static inline hipError_t hipHostMalloc(T** p, size_t n, unsigned f = 0) { return hipHostMalloc((void**)p, n, f); }
static inline hipError_t hipFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipHostFree(void* p) { free(p); return hipSuccess; }
static inline hipError_t hipMemcpy(void* d, const void* s, size_t n, hipMemcpyKind) { if (n) memmove(d, s, n); return hipSuccess; }
static inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind k, hipStream_t) { return hipMemcpy(d, s, n, k); }
static inline hipError_t hipMemset(void* d, int v, size_t n) { if (n) memset(d, v, n); return hipSuccess; }
static inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { return hipMemset(d, v, n); }
static inline hipError_t hipMemsetD32Async(void* d, int v, size_t count, hipStream_t) { int* q = (int*)d; for (size_t i = 0; i < count; i++) q[i] = v; return hipSuccess; }
static inline hipError_t hipStreamCreate(hipStream_t* s) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithFlags(hipStream_t* s, unsigned) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipStreamCreateWithPriority(hipStream_t* s, unsigned, int) { *s = (void*)1; return hipSuccess; }
static inline hipError_t hipDeviceGetStreamPriorityRange(int* least, int* greatest) { *least = 1; *greatest = -1; return hipSuccess; }
#define hipStreamNonBlocking 1
#define hipHostMallocDefault 0
static inline hipError_t hipStreamDestroy(hipStream_t) { return hipSuccess; }
static inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
static inline hipError_t hipDeviceSynchronize() { return hipSuccess; }
#include <time.h>
static inline long long hipemu_now_ns();
static inline double hipemu_now() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1e3 + ts.tv_nsec * 1e-6; }
static inline long long hipemu_now_ns() { timespec ts; clock_gettime(CLOCK_MONOTONIC, &ts); return ts.tv_sec * 1000000000ll + ts.tv_nsec; }
static inline hipError_t hipEventCreate(hipEvent_t* e) { *e = new hipemu_event{0}; return hipSuccess; }
constexpr unsigned hipEventDisableTiming = 2;
static inline hipError_t hipEventCreateWithFlags(hipEvent_t* e, unsigned) { *e = new hipemu_event{0}; return hipSuccess; }
static inline hipError_t hipStreamWaitEvent(hipStream_t, hipEvent_t, unsigned) { return hipSuccess; }  // the emulator is synchronous
static inline hipError_t hipEventDestroy(hipEvent_t e) { delete e; return hipSuccess; }
static inline hipError_t hipEventRecord(hipEvent_t e, hipStream_t) { e->t = hipemu_now(); return hipSuccess; }
static inline hipError_t hipEventSynchronize(hipEvent_t) { return hipSuccess; }
static inline hipError_t hipEventElapsedTime(float* ms, hipEvent_t a, hipEvent_t b) { *ms = (float)(b->t - a->t); return hipSuccess; }

// ---- optional: device-math-library stand-in (-DMOT_EMU_PERTURB, built by MOT_EMU_PERTURB=1) -------------------------------
// sin / cos / sincos / exp / atan2 / pow of the device math library are correctly rounded in most cases, not all: they may
// differ from glibc in the last bit. This mode moves each result of these functions by one ulp up, down or not at all
// (chosen by a hash of the argument bits, so the same call gives the same answer), to see on the CPU whether anything
// discrete — a box corner after its rounding to fp32, a gate decision, a track state — depends on such a bit.
#ifdef MOT_EMU_PERTURB
namespace hipemu {
inline double nudge(double r, double a, double b = 0.0) {
  uint64_t x, y; memcpy(&x, &a, 8); memcpy(&y, &b, 8);
  uint64_t h = (x ^ (y * 0x9E3779B97F4A7C15ull)) * 0xD1342543DE82EF95ull; h ^= h >> 29; h *= 0xBF58476D1CE4E5B9ull; h ^= h >> 32;
  const unsigned k = (unsigned)(h % 3u);
  if (!std::isfinite(r) || k == 0) return r;
  return std::nextafter(r, k == 1 ? INFINITY : -INFINITY);
}
inline double p_sin(double a) { return nudge(std::sin(a), a); }
inline double p_cos(double a) { return nudge(std::cos(a), a, 1.0); }
inline double p_exp(double a) { return nudge(std::exp(a), a, 2.0); }
inline double p_atan2(double a, double b) { return nudge(std::atan2(a, b), a, b); }
inline double p_pow(double a, double b) { return nudge(std::pow(a, b), a, b); }
inline void p_sincos(double a, double* s, double* c) { *s = p_sin(a); *c = p_cos(a); }
}  // namespace hipemu
#define sin(x) hipemu::p_sin(x)
#define cos(x) hipemu::p_cos(x)
#define exp(x) hipemu::p_exp(x)
#define atan2(y, x) hipemu::p_atan2((y), (x))
#define pow(x, y) hipemu::p_pow((x), (y))
#define sincos(x, s, c) hipemu::p_sincos((x), (s), (c))
#endif

#endif  // HIPEMU_H_
