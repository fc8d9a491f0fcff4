// hip_emu.h -- TEST INFRASTRUCTURE.  Whole kernels on the CPU: enough of the HIP programming model for the kernels of
// rpg_svo_amd/csrc that use a workgroup's barrier, LDS and the plain cross-lane moves to be compiled by the host compiler
// and launched through their own C-ABI entry points.  A launch runs the workgroups one after the other, each with one FIBER per
// work-item on the calling thread; __syncthreads is a barrier that work-items which have left the kernel drop out of (as
// exited waves do), __shared__ is `static` (one workgroup at a time), __shfl_up an exchange inside the 64 threads of a wave, atomicAdd
// a host atomic.  No timing, no memory model subtleties: what this checks is the kernels' LOGIC against the oracle.
// Include first in the test's translation unit (it defines SVO_HOST_MATH_TEST for the product headers).
#pragma once
#define SVO_HOST_MATH_TEST
#define SVO_HIP_EMU

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <map>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <memory>
#include <ucontext.h>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
typedef void* hipStream_t;
// memory: "device" memory is host memory, copies are memcpy, streams are synchronous
enum hipMemcpyKind { hipMemcpyHostToHost = 0, hipMemcpyHostToDevice = 1, hipMemcpyDeviceToHost = 2, hipMemcpyDeviceToDevice = 3, hipMemcpyDefault = 4 };
// (fresh "device" memory is poisoned -- NaN as a float, -1 as an integer -- so that a kernel reading what nobody wrote shows)
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); if (*p) std::memset(*p, 0xFF, n ? n : 1); return *p ? hipSuccess : 2; }
inline hipError_t hipMallocAsync(void** p, size_t n, hipStream_t) { return hipMalloc(p, n); }
inline hipError_t hipFree(void* p) { std::free(p); return hipSuccess; }
inline hipError_t hipFreeAsync(void* p, hipStream_t) { std::free(p); return hipSuccess; }
inline hipError_t hipMemcpyAsync(void* d, const void* s, size_t n, hipMemcpyKind, hipStream_t) { std::memcpy(d, s, n); return hipSuccess; }
inline hipError_t hipMemcpy2DAsync(void* d, size_t dpitch, const void* s, size_t spitch, size_t w, size_t h, hipMemcpyKind, hipStream_t) {
  for (size_t r = 0; r < h; ++r) std::memcpy(static_cast<char*>(d) + r * dpitch, static_cast<const char*>(s) + r * spitch, w);
  return hipSuccess;
}
inline hipError_t hipMemsetAsync(void* d, int v, size_t n, hipStream_t) { std::memset(d, v, n); return hipSuccess; }
inline hipError_t hipStreamSynchronize(hipStream_t) { return hipSuccess; }
struct alignas(16) uint4 {
  uint32_t x, y, z, w;
};
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
struct alignas(8) uint2 {
  uint32_t x, y;
};
inline uint2 make_uint2(uint32_t x, uint32_t y) { return uint2{x, y}; }
struct alignas(8) float2 {
  float x, y;
};
inline float2 make_float2(float x, float y) { return float2{x, y}; }
struct alignas(16) float4 {
  float x, y, z, w;
};
inline float4 make_float4(float x, float y, float z, float w) { return float4{x, y, z, w}; }

namespace svo_emu {

// One workgroup at a time, its work-items as FIBERS (ucontext) of the calling OS thread: a work-item runs until it
// reaches a barrier or an exchange, yields, and is resumed when every work-item that is still alive has arrived -- no OS
// scheduling, ~0.2 us per switch, so a launch of hundreds of workgroups of 256 work-items costs milliseconds.
struct Fiber {
  ucontext_t ctx;
  std::vector<char> stack;
  dim3 tid;
  unsigned flat = 0;
  bool done = false;
  int waiting_on = -1;           // -1 runnable; 0 the workgroup barrier; 1 .. WAVE_BARRIER-1: a cross-lane rendezvous (site id);
                                 // WAVE_BARRIER + w: the barrier of all live lanes of wave w
  unsigned long long slot = 0;   // value offered to a rendezvous
  int site = 0;                  // call site of the wave-wide collective the work-item stands at
  const struct Rendezvous* met = nullptr;  // the one this work-item was released from
  void* tsan = nullptr;          // (ThreadSanitizer builds: the fiber as the sanitizer knows it)
};

constexpr int WAVE_BARRIER = 1 << 24;

// What the lanes of one wave that met at one call site offered: who took part, and their values.
struct Rendezvous {
  unsigned long long mask = 0;
  unsigned long long val[64];
};

struct Block {
  std::deque<Fiber> fibers;      // (a deque: a suspended fiber's saved context must not move)
  std::map<std::pair<int, unsigned>, Rendezvous> met;  // (site, wave) -> the last meeting there
  ucontext_t scheduler;
  dim3 bid, bdim, gdim;
  unsigned alive = 0;
  std::vector<unsigned> wave_alive;
};

struct alignas(16) Lds16 { char b[16]; };
inline std::vector<Lds16> g_dyn_lds = [] { std::vector<Lds16> v; v.reserve(160 * 1024 / 16 + 1); return v; }();  // the running launch's dynamic LDS (one address for good)
inline Block* g_block = nullptr;   // (one OS thread runs the emulation)
inline Fiber* g_fiber = nullptr;
inline std::function<void()>* g_body = nullptr;

// Sanitizer builds (tests/emu_build.py with SVO_EMU_SANITIZE=address or thread; scripts/emu_sanitize.sh).
//  address: the kernels' loads and stores checked against the bounds of the buffers they were handed and of their LDS
//           arrays; the sanitizer is told about every change of stack.
//  thread:  every work-item is a fiber of its own to ThreadSanitizer, switches carry NO ordering, and the only ordering
//           between work-items is what the kernel asks for: __syncthreads (workgroup), the wave-wide collectives and
//           hand-overs (wave), the cross-lane moves and lane-group hand-overs (the lanes that meet at the call site).
//           A store to LDS or global memory by one work-item and a load or store of the same bytes by another with none
//           of these in between -- a missing barrier, or a hand-over that relies on lock step without telling the
//           compiler -- is reported as a data race with both source lines.  Workgroups are ordered one after the other
//           (they share the static LDS arrays here), so races BETWEEN workgroups are not looked for.
//           The emulator's own bookkeeping is exempt (SVO_EMU_NOTSAN).
#if defined(__has_feature)
#if __has_feature(address_sanitizer)
#define SVO_EMU_ASAN 1
extern "C" void __sanitizer_start_switch_fiber(void** fake_stack_save, const void* bottom, size_t size);
extern "C" void __sanitizer_finish_switch_fiber(void* fake_stack_save, const void** bottom_old, size_t* size_old);
#endif
#if __has_feature(thread_sanitizer)
#define SVO_EMU_TSAN 1
extern "C" void* __tsan_get_current_fiber(void);
extern "C" void* __tsan_create_fiber(unsigned flags);
extern "C" void __tsan_switch_to_fiber(void* fiber, unsigned flags);
extern "C" void __tsan_acquire(void* addr);
extern "C" void __tsan_release(void* addr);
extern "C" void AnnotateBenignRaceSized(const char* file, int line, const volatile void* mem, size_t size, const char* description);
// the static LDS arrays of a library (ThreadSanitizer builds put them into one section: see __shared__ below)
extern "C" char __start_svo_lds[] __attribute__((weak, visibility("hidden")));
extern "C" char __stop_svo_lds[] __attribute__((weak, visibility("hidden")));
#endif
#endif
// SVO_EMU_TSAN_BETWEEN_WORKGROUPS=1 (ThreadSanitizer builds): the workgroups of a launch are NOT ordered one after the other
// any more -- as on the device, where nothing orders them -- and the LDS arrays (every workgroup's own there, the same static
// arrays here) are exempt.  What is reported then is global memory that one workgroup writes and another
// reads or writes without an atomic: results that depend on the order the workgroups happen to run in.  (A work-item
// of one workgroup and the work-item of the same index of the next are the same sanitizer thread: not looked at.)
inline int schedule_mode() {  // 0: in order; 1: SVO_EMU_SCHEDULE=reverse; 2: =waves
  static const int m = [] {
    const char* e = std::getenv("SVO_EMU_SCHEDULE");
    return !e ? 0 : (e[0] == 'r' ? 1 : (e[0] == 'w' ? 2 : 0));
  }();
  return m;
}
inline bool between_workgroups_mode() {
  static const bool on = [] { const char* e = std::getenv("SVO_EMU_TSAN_BETWEEN_WORKGROUPS"); return e && e[0] == '1'; }();
  return on;
}
#ifdef SVO_EMU_TSAN
#define SVO_EMU_NOTSAN __attribute__((no_sanitize("thread"), noinline))
#else
#define SVO_EMU_NOTSAN
#endif
inline const void* g_sched_stack = nullptr;  // the scheduler's (= the calling thread's) stack, learnt at a fiber's first entry
inline size_t g_sched_stack_size = 0;
inline void* g_sched_tsan = nullptr;         // the calling thread, as ThreadSanitizer knows it
// what a barrier's arrivals publish and its departures pick up (addresses only; ThreadSanitizer keeps a clock per address)
inline char g_sync_block, g_sync_start, g_sync_exit, g_sync_wave[64], g_sync_site[16][8192];
SVO_EMU_NOTSAN inline void* sync_object(int id, unsigned flat) {
  if (id == 0) return &g_sync_block;
  if (id >= WAVE_BARRIER) return &g_sync_wave[(id - WAVE_BARRIER) & 63];
  return &g_sync_site[(flat / 64) & 15][id & 8191];
}

// from the running work-item back to the scheduler
SVO_EMU_NOTSAN inline void yield_to_scheduler() {
  Fiber* self = g_fiber;
#ifdef SVO_EMU_ASAN
  void* fake = nullptr;
  __sanitizer_start_switch_fiber(&fake, g_sched_stack, g_sched_stack_size);
#endif
#ifdef SVO_EMU_TSAN
  __tsan_switch_to_fiber(g_sched_tsan, 1u /* no ordering */);
#endif
  swapcontext(&self->ctx, &g_block->scheduler);
#ifdef SVO_EMU_ASAN
  __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
}

// a work-item's fiber lives as long as the emulator: it runs the kernel body of one workgroup after the other
SVO_EMU_NOTSAN inline void fiber_main() {
#ifdef SVO_EMU_ASAN
  __sanitizer_finish_switch_fiber(nullptr, &g_sched_stack, &g_sched_stack_size);
#endif
  for (;;) {
#ifdef SVO_EMU_TSAN
    __tsan_acquire(&g_sync_start);  // what the host and the workgroups before this one wrote
#endif
    (*g_body)();
#ifdef SVO_EMU_TSAN
    __tsan_release(&g_sync_exit);
#endif
    g_fiber->done = true;
    --g_block->alive;
    --g_block->wave_alive[g_fiber->flat / 64];
    yield_to_scheduler();
  }
}

// park the running work-item on barrier `id` until all live work-items of its scope have arrived
SVO_EMU_NOTSAN inline void barrier_wait(int id) {
  g_fiber->waiting_on = id;
#ifdef SVO_EMU_TSAN
  void* so = sync_object(id, g_fiber->flat);
  __tsan_release(so);
#endif
  yield_to_scheduler();
#ifdef SVO_EMU_TSAN
  __tsan_acquire(so);
#endif
}

SVO_EMU_NOTSAN inline void run_block(Block& b, std::function<void()>& body, size_t stack_bytes, bool first_of_launch = true,
                                     bool last_of_launch = true) {
  g_block = &b;
  g_body = &body;
#ifdef SVO_EMU_TSAN
  g_sched_tsan = __tsan_get_current_fiber();
  const bool xwg = between_workgroups_mode();
  if (!xwg || first_of_launch) __tsan_acquire(&g_sync_exit);   // the workgroup (or, between workgroups: the launch) before this one
  if (xwg) {  // LDS is every workgroup's own on the device and the same static arrays here: not looked at in this mode
    static bool lds_exempt = false;
    static const void* dyn_exempt = nullptr;
    if (!lds_exempt && __start_svo_lds && __stop_svo_lds > __start_svo_lds) {
      AnnotateBenignRaceSized(__FILE__, __LINE__, __start_svo_lds, (size_t)(__stop_svo_lds - __start_svo_lds), "LDS (between-workgroups mode)");
      lds_exempt = true;
    }
    if (!g_dyn_lds.empty() && dyn_exempt != g_dyn_lds.data()) {
      dyn_exempt = g_dyn_lds.data();
      AnnotateBenignRaceSized(__FILE__, __LINE__, g_dyn_lds.data(), g_dyn_lds.capacity() * sizeof(Lds16), "dynamic LDS (between-workgroups mode)");
    }
  }
  __tsan_release(&g_sync_start);
#endif
  const unsigned n = (unsigned)b.fibers.size();
  b.alive = n;
  b.wave_alive.assign((n + 63) / 64, 0);
  for (unsigned t = 0; t < n; ++t) {
    Fiber& f = b.fibers[t];
    f.flat = t;
    f.tid = dim3(t % b.bdim.x, (t / b.bdim.x) % b.bdim.y, t / (b.bdim.x * b.bdim.y));
    f.done = false;
    f.waiting_on = -1;
    ++b.wave_alive[t / 64];
    if (f.stack.size() != stack_bytes) {  // a new work-item: its fiber starts at the top of fiber_main's loop
      f.stack.resize(stack_bytes);
      getcontext(&f.ctx);
      f.ctx.uc_stack.ss_sp = f.stack.data();
      f.ctx.uc_stack.ss_size = f.stack.size();
      f.ctx.uc_link = &b.scheduler;
      makecontext(&f.ctx, fiber_main, 0);
#ifdef SVO_EMU_TSAN
      f.tsan = __tsan_create_fiber(0);
#endif
    }
  }
  while (b.alive > 0) {
    bool progressed = false;
    for (unsigned k = 0; k < n; ++k) {  // run every runnable work-item up to its next barrier
      // SVO_EMU_SCHEDULE=reverse: the work-items (and, in launch(), the workgroups) take their turns in the opposite order;
      // =waves: the waves in the opposite order, the lanes of a wave in order.  Nothing a kernel computes may depend on it.
      const unsigned t = schedule_mode() == 1 ? n - 1 - k : (schedule_mode() == 2 ? (((n - 1 - k) & ~63u) | (k & 63u)) : k);
      if (t >= n) continue;
      Fiber& f = b.fibers[t];
      if (f.done || f.waiting_on >= 0) continue;
      g_fiber = &f;
#ifdef SVO_EMU_ASAN
      void* fake = nullptr;
      __sanitizer_start_switch_fiber(&fake, f.stack.data(), f.stack.size());
#endif
#ifdef SVO_EMU_TSAN
      __tsan_switch_to_fiber(f.tsan, 1u /* no ordering */);
#endif
      swapcontext(&b.scheduler, &f.ctx);
#ifdef SVO_EMU_ASAN
      __sanitizer_finish_switch_fiber(fake, nullptr, nullptr);
#endif
      progressed = true;
    }
    if (progressed) continue;
    // nobody can run.  In this order: the workgroup barrier if every live work-item stands at it; the barriers of whole
    // waves that are complete; else, per wave, the cross-lane rendezvous at the EARLIEST call site -- the lanes that stand
    // there are the ones that execute it together (whoever was going to arrive has arrived: everybody else is blocked
    // further down or gone), and the lanes waiting further down must not go on before these have caught up.
    unsigned at_block = 0;
    std::vector<unsigned> at_wave(b.wave_alive.size(), 0);
    std::vector<int> first_site(b.wave_alive.size(), 0);
    for (unsigned t = 0; t < n; ++t) {
      const Fiber& f = b.fibers[t];
      if (f.done) continue;
      const size_t w = f.flat / 64;
      if (f.waiting_on == 0) ++at_block;
      else if (f.waiting_on >= WAVE_BARRIER) ++at_wave[w];
      else if (f.waiting_on > 0 && (first_site[w] == 0 || f.waiting_on < first_site[w])) first_site[w] = f.waiting_on;
    }
    if (b.alive > 0 && at_block == b.alive) {
      for (Fiber& f : b.fibers) if (!f.done && f.waiting_on == 0) f.waiting_on = -1;
      continue;
    }
    bool released = false;
    for (size_t w = 0; w < at_wave.size(); ++w)
      if (b.wave_alive[w] > 0 && at_wave[w] == b.wave_alive[w]) {
        for (Fiber& f : b.fibers) if (!f.done && f.flat / 64 == w && f.waiting_on >= WAVE_BARRIER) f.waiting_on = -1;
        released = true;
      }
    if (released) continue;
    b.met.clear();
    for (Fiber& f : b.fibers) {
      const size_t w = f.flat / 64;
      if (!f.done && f.waiting_on > 0 && f.waiting_on < WAVE_BARRIER && f.waiting_on == first_site[w]) {
        Rendezvous& r = b.met[std::make_pair(f.waiting_on, (unsigned)w)];
        r.mask |= 1ull << (f.flat & 63u);
        r.val[f.flat & 63u] = f.slot;
      }
    }
    for (Fiber& f : b.fibers) {
      const size_t w = f.flat / 64;
      if (!f.done && f.waiting_on > 0 && f.waiting_on < WAVE_BARRIER && f.waiting_on == first_site[w]) {
        f.met = &b.met[std::make_pair(f.waiting_on, (unsigned)w)];
        f.waiting_on = -1;
        released = true;
      }
    }
    if (released) continue;
    if (b.alive > 0) {
      std::fprintf(stderr, "hip_emu: deadlock (work-items stand at barriers that cannot complete); workgroup (%u,%u,%u), per wave: ",
                   b.bid.x, b.bid.y, b.bid.z);
      for (size_t w = 0; w < b.wave_alive.size(); ++w) {
        std::map<int, int> where;
        for (const Fiber& f : b.fibers)
          if (!f.done && f.flat / 64 == w) ++where[f.waiting_on];
        std::fprintf(stderr, "[wave %zu:", w);
        for (auto& kv : where)
          std::fprintf(stderr, " %d lanes at %s%d", kv.second, kv.first == 0 ? "syncthreads " : (kv.first >= WAVE_BARRIER ? "wave barrier " : "line "),
                       kv.first >= WAVE_BARRIER ? kv.first - WAVE_BARRIER : kv.first);
        std::fprintf(stderr, "] ");
      }
      std::fprintf(stderr, "\n");
      std::abort();
    }
  }
#ifdef SVO_EMU_TSAN
  if (!xwg || last_of_launch) __tsan_acquire(&g_sync_exit);  // the host reads what the kernel wrote
#endif
  (void)first_of_launch; (void)last_of_launch;
  g_block = nullptr;
  g_fiber = nullptr;
}

template <typename F>
SVO_EMU_NOTSAN void launch(dim3 grid, dim3 block, F&& body_in) {
  std::function<void()> body = body_in;
  const unsigned n = block.x * block.y * block.z;
  static thread_local Block blk;  // (stacks are kept between launches)
  blk.fibers.resize(n);
  blk.bdim = block;
  blk.gdim = grid;
  const size_t stack_bytes = 256 * 1024;
  const unsigned long long n_blocks = (unsigned long long)grid.x * grid.y * grid.z;
  for (unsigned long long i = 0; i < n_blocks; ++i) {
    const unsigned long long j = schedule_mode() ? n_blocks - 1 - i : i;  // (reversed schedules: the last workgroup first)
    blk.bid = dim3((unsigned)(j % grid.x), (unsigned)((j / grid.x) % grid.y), (unsigned)(j / ((unsigned long long)grid.x * grid.y)));
    run_block(blk, body, stack_bytes, i == 0, i + 1 == n_blocks);
  }
}

// the running work-item meets the other lanes of its wave that reach call site `site`; returns what they offered
template <typename T>
SVO_EMU_NOTSAN inline const Rendezvous& meet(int site, T v) {
  static_assert(sizeof(T) <= 8, "rendezvous slot");
  unsigned long long bits = 0;
  __builtin_memcpy(&bits, &v, sizeof(T));
  g_fiber->slot = bits;
  barrier_wait(site);
  return *g_fiber->met;
}
template <typename T>
SVO_EMU_NOTSAN inline T lane_value(const Rendezvous& r, int src_lane, T own) {  // the value lane `src_lane` offered; `own` if it did not take part
  if (src_lane < 0 || src_lane > 63 || !((r.mask >> src_lane) & 1ull)) return own;
  T out;
  __builtin_memcpy(&out, &r.val[src_lane], sizeof(T));
  return out;
}
inline int my_lane() { return (int)(g_fiber->flat & 63u); }
inline void wave_barrier() { barrier_wait(WAVE_BARRIER + (int)(g_fiber->flat / 64)); }  // every live lane of the wave
template <typename T>
inline T shfl_up(int site, T v, unsigned delta, int /*width*/) { return lane_value(meet(site, v), my_lane() - (int)delta, v); }
template <typename T>
inline T shfl(int site, T v, int src, int /*width*/) { return lane_value(meet(site, v), src & 63, v); }
template <typename T>
inline T shfl_xor(int site, T v, int mask, int /*width*/) { return lane_value(meet(site, v), my_lane() ^ mask, v); }
// a value from every live lane of the wave: collectives that every lane of the wave executes (ballot, readfirstlane,
// readlane, the permlane swaps)
template <typename T>
SVO_EMU_NOTSAN inline void wave_gather(T v, unsigned long long* mask, unsigned long long vals[64], int site = 0) {
  unsigned long long bits = 0;
  __builtin_memcpy(&bits, &v, sizeof(T));
  g_fiber->slot = bits;
  g_fiber->site = site;
  wave_barrier();
  const unsigned w0 = (g_fiber->flat / 64) * 64;
  *mask = 0;
  for (unsigned l = 0; l < 64 && w0 + l < g_block->fibers.size(); ++l) {
    const Fiber& f = g_block->fibers[w0 + l];
    if (f.done) continue;
    *mask |= 1ull << l;
    vals[l] = f.slot;
    if (f.site != site) {  // a collective of the whole wave that only part of the wave executes: SVO_BALLOT_ACTIVE & co. are for that
      std::fprintf(stderr, "hip_emu: lanes %u and %u of a wave stand at different wave-wide collectives (lines %d and %d)\n",
                   g_fiber->flat & 63u, l, site, f.site);
      std::abort();
    }
  }
  wave_barrier();  // everybody has read before anybody offers again
}
SVO_EMU_NOTSAN inline unsigned long long ballot(bool pred, int site = 0) {
  unsigned long long mask, vals[64], m = 0;
  wave_gather((unsigned long long)(pred ? 1 : 0), &mask, vals, site);
  for (int l = 0; l < 64; ++l)
    if (((mask >> l) & 1ull) && vals[l]) m |= 1ull << l;
  return m;
}
// the ballot of the lanes that are active in a divergent branch (SVO_BALLOT_ACTIVE): those that arrive
SVO_EMU_NOTSAN inline unsigned long long ballot_active(int site, bool pred) {
  const Rendezvous& r = meet(site, (unsigned long long)(pred ? 1 : 0));
  unsigned long long m = 0;
  for (int l = 0; l < 64; ++l)
    if (((r.mask >> l) & 1ull) && r.val[l]) m |= 1ull << l;
  return m;
}
template <typename T>
SVO_EMU_NOTSAN inline T readfirstlane(T v, int site = 0) {
  unsigned long long mask, vals[64];
  wave_gather(v, &mask, vals, site);
  T out;
  __builtin_memcpy(&out, &vals[__builtin_ctzll(mask)], sizeof(T));
  return out;
}
// the DPP controls the kernels use (bound_ctrl: a lane without a source gets 0)
SVO_EMU_NOTSAN inline int update_dpp(int site, int /*old*/, int v, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
  const Rendezvous& r = meet(site, v);
  const int l = my_lane();
  int src;
  switch (ctrl) {
    case 0x111: src = (l & 15) >= 1 ? l - 1 : -1; break;                  // row_shr:1
    case 0xB1: src = l ^ 1; break;                                        // quad_perm [1,0,3,2]
    case 0x4E: src = l ^ 2; break;                                        // quad_perm [2,3,0,1]
    case 0x141: src = (l & ~7) | (7 - (l & 7)); break;                    // row_half_mirror
    case 0x128: src = (l & ~15) | ((l + 8) & 15); break;                  // row_ror:8
    default: std::fprintf(stderr, "hip_emu: DPP control %#x not emulated\n", ctrl); std::abort();
  }
  return lane_value(r, src, 0);
}
inline uint32_t udot4(uint32_t a, uint32_t b, uint32_t c, bool /*clamp*/) {
  for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
}
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sel) {
  return (uint32_t)(((((uint64_t)hi) << 32) | (uint64_t)lo) >> (8u * (sel & 3u)));
}
// v_permlane32_swap / v_permlane16_swap: the upper half (odd 16-lane rows) of the first operand changes places with the
// lower half (even rows) of the second; both results come back (rpg_svo_amd/csrc/wave_reduce.h states what the kernels
// rely on: lanes < 32 see {a[l], a[l+32]}, lanes >= 32 {b[l-32], b[l]}).  Every lane of the wave executes it.
struct Pair32 {
  uint32_t v[2];
  uint32_t operator[](int i) const { return v[i]; }
};
SVO_EMU_NOTSAN inline Pair32 permlane_swap(uint32_t a, uint32_t b, int half, int site = 0) {
  unsigned long long mask, vals[64];
  wave_gather(((unsigned long long)b << 32) | a, &mask, vals, site);
  const int l = my_lane();
  const bool upper = (l & half) != 0;
  auto A = [&](int lane) { return ((mask >> lane) & 1ull) ? (uint32_t)vals[lane] : 0u; };
  auto B = [&](int lane) { return ((mask >> lane) & 1ull) ? (uint32_t)(vals[lane] >> 32) : 0u; };
  Pair32 r;
  if (!upper) { r.v[0] = A(l); r.v[1] = A(l + half); }
  else { r.v[0] = B(l - half); r.v[1] = B(l); }
  return r;
}
SVO_EMU_NOTSAN inline bool syncthreads_or(bool pred) {
  static int flag;          // (one workgroup at a time)
  barrier_wait(0);
  if (g_fiber->flat == 0) flag = 0;
  barrier_wait(0);
  if (pred) flag = 1;
  barrier_wait(0);
  return flag != 0;
}

SVO_EMU_NOTSAN inline int readlane(int v, int src, int site = 0) {
  unsigned long long mask, vals[64];
  wave_gather(v, &mask, vals, site);
  int out = 0;
  if ((mask >> (src & 63)) & 1ull) __builtin_memcpy(&out, &vals[src & 63], sizeof(int));
  return out;
}
inline uint32_t mbcnt(unsigned long long mask_part_shifted, uint32_t base) { return base + (uint32_t)__builtin_popcountll(mask_part_shifted); }

}  // namespace svo_emu

using std::max;
using std::min;
#define threadIdx (svo_emu::g_fiber->tid)
#define blockIdx (svo_emu::g_block->bid)
#define blockDim (svo_emu::g_block->bdim)
#define gridDim (svo_emu::g_block->gdim)
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__ static const
#define __global__
#ifdef SVO_EMU_TSAN
#define __shared__ static __attribute__((section("svo_lds")))
#else
#define __shared__ static
#endif
#define __launch_bounds__(...)
#define __syncthreads() (svo_emu::barrier_wait(0))
#define __shfl_up(...) svo_emu::shfl_up(__LINE__, __VA_ARGS__)
#define __shfl(...) svo_emu::shfl(__LINE__, __VA_ARGS__)
#define __shfl_xor(...) svo_emu::shfl_xor(__LINE__, __VA_ARGS__)
#define __ballot(p) svo_emu::ballot((p), __LINE__)
#define __builtin_amdgcn_ballot_w64(p) svo_emu::ballot((p), __LINE__)
#define SVO_BALLOT_ACTIVE(p) svo_emu::ballot_active(__LINE__, (p))
#define __builtin_amdgcn_readfirstlane(v) svo_emu::readfirstlane((v), __LINE__)
#define __builtin_amdgcn_update_dpp(...) svo_emu::update_dpp(__LINE__, __VA_ARGS__)
#define __builtin_amdgcn_mbcnt_lo(m, base) svo_emu::mbcnt((unsigned long long)(uint32_t)(m) & ((svo_emu::my_lane() >= 32 ? 0xffffffffull : ((1ull << svo_emu::my_lane()) - 1ull))), (base))
#define __builtin_amdgcn_mbcnt_hi(m, base) svo_emu::mbcnt(svo_emu::my_lane() > 32 ? ((unsigned long long)(uint32_t)(m) & ((1ull << (svo_emu::my_lane() - 32)) - 1ull)) : 0ull, (base))
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() svo_emu::wave_barrier()
#define SVO_WAVE_LDS_HANDOVER() svo_emu::wave_barrier()
#define SVO_LANES_LDS_HANDOVER() ((void)svo_emu::meet(__LINE__, 0))
#define SVO_WAVE_LDS_FENCE() svo_emu::wave_barrier()
#define SVO_LANES_LDS_FENCE() ((void)svo_emu::meet(__LINE__, 0))
#define __builtin_amdgcn_udot4(...) svo_emu::udot4(__VA_ARGS__)
#define __builtin_amdgcn_alignbyte(...) svo_emu::alignbyte(__VA_ARGS__)
#define __hip_atomic_store(p, v, order, scope) __atomic_store_n((p), (v), __ATOMIC_RELAXED)
#define __builtin_amdgcn_permlane32_swap(a, b, fi, bc) svo_emu::permlane_swap((a), (b), 32, __LINE__)
#define __builtin_amdgcn_permlane16_swap(a, b, fi, bc) svo_emu::permlane_swap((a), (b), 16, __LINE__)
#define __builtin_amdgcn_s_setprio(x) ((void)0)
#define __builtin_amdgcn_readlane(v, l) svo_emu::readlane((v), (l), __LINE__)
#define __builtin_amdgcn_rcp(x) (1.0 / (x))
#define __builtin_amdgcn_rsq(x) (1.0 / std::sqrt(x))
#define __syncthreads_or(p) svo_emu::syncthreads_or((p))
template <typename To, typename From>
inline To svo_bits(From v) {
  static_assert(sizeof(To) == sizeof(From), "bit cast");
  To r;
  std::memcpy(&r, &v, sizeof(To));
  return r;
}
inline float __int_as_float(int x) { return svo_bits<float>(x); }
inline float __uint_as_float(unsigned x) { return svo_bits<float>(x); }
inline int __float_as_int(float x) { return svo_bits<int>(x); }
inline unsigned __float_as_uint(float x) { return svo_bits<unsigned>(x); }
inline double __longlong_as_double(long long x) { return svo_bits<double>(x); }
inline long long __double_as_longlong(double x) { return svo_bits<long long>(x); }
#define __popcll(x) __builtin_popcountll(x)
#define __clz(x) ((x) ? __builtin_clz(x) : 32)
#define __ffsll(x) __builtin_ffsll(x)
inline int atomicAdd(int* p, int v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline unsigned atomicAdd(unsigned* p, unsigned v) { return __atomic_fetch_add(p, v, __ATOMIC_RELAXED); }
inline int atomicMin(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old > v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline int atomicMax(int* p, int v) {
  int old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
inline unsigned long long atomicMax(unsigned long long* p, unsigned long long v) {
  unsigned long long old = __atomic_load_n(p, __ATOMIC_RELAXED);
  while (old < v && !__atomic_compare_exchange_n(p, &old, v, true, __ATOMIC_RELAXED, __ATOMIC_RELAXED)) {}
  return old;
}
// dynamically sized LDS: one buffer per launch, sized by the launch's byte count
#define SVO_DYNAMIC_LDS(type, name) type* const name = reinterpret_cast<type*>(svo_emu::g_dyn_lds.data())
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  (svo_emu::g_dyn_lds.assign(((size_t)(shmem) + 15) / 16 + 1, svo_emu::Lds16{}), svo_emu::launch((grid), (block), [&] { kernel(__VA_ARGS__); }))
