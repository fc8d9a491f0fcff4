// hip_emu.h -- TEST INFRASTRUCTURE.  Whole kernels on the CPU: enough of the HIP programming model for the kernels of
// rpg_svo_amd/csrc that use a workgroup's barrier, LDS and the plain cross-lane moves to be compiled by the host compiler
// and launched through their own C-ABI entry points.  A launch runs the workgroups one after the other, each with one FIBER per
// work-item on the calling thread; __syncthreads is a barrier that work-items which have left the kernel drop out of (as
// exited waves do), __shared__ is `static` (one workgroup at a time), __shfl_up an exchange inside the 64 threads of a wave, atomicAdd
// a host atomic.  No timing, no memory model subtleties: what this checks is the kernels' LOGIC against the oracle.
// Include first in the test's translation unit (it defines SVO_HOST_MATH_TEST for the product headers).
#pragma once
#define SVO_HOST_MATH_TEST

#include <algorithm>
#include <cmath>
#include <cstdio>
#include <functional>
#include <cstdint>
#include <cstdlib>
#include <cstring>
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
inline hipError_t hipMalloc(void** p, size_t n) { *p = std::malloc(n ? n : 1); return *p ? hipSuccess : 2; }
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
  int waiting_on = -1;           // -1 runnable; 0 the workgroup barrier; 1 + w: the exchange barrier of wave w
  unsigned long long slot = 0;   // value offered to an exchange
};

struct Block {
  std::vector<Fiber> fibers;
  ucontext_t scheduler;
  dim3 bid, bdim, gdim;
  unsigned alive = 0;
  std::vector<unsigned> wave_alive;
};

inline Block* g_block = nullptr;   // (one OS thread runs the emulation)
inline Fiber* g_fiber = nullptr;
inline std::function<void()>* g_body = nullptr;

inline void fiber_main() {
  (*g_body)();
  g_fiber->done = true;
  --g_block->alive;
  --g_block->wave_alive[g_fiber->flat / 64];
  swapcontext(&g_fiber->ctx, &g_block->scheduler);
}

// park the running work-item on barrier `id` until all live work-items of its scope have arrived
inline void barrier_wait(int id) {
  g_fiber->waiting_on = id;
  swapcontext(&g_fiber->ctx, &g_block->scheduler);
}

inline void run_block(Block& b, std::function<void()>& body, size_t stack_bytes) {
  g_block = &b;
  g_body = &body;
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
    if (f.stack.size() != stack_bytes) f.stack.resize(stack_bytes);
    getcontext(&f.ctx);
    f.ctx.uc_stack.ss_sp = f.stack.data();
    f.ctx.uc_stack.ss_size = f.stack.size();
    f.ctx.uc_link = &b.scheduler;
    makecontext(&f.ctx, fiber_main, 0);
  }
  while (b.alive > 0) {
    bool progressed = false;
    for (unsigned t = 0; t < n; ++t) {  // run every runnable work-item up to its next barrier
      Fiber& f = b.fibers[t];
      if (f.done || f.waiting_on >= 0) continue;
      g_fiber = &f;
      swapcontext(&b.scheduler, &f.ctx);
      progressed = true;
    }
    // release the barriers every live work-item of the scope has reached
    unsigned at_block = 0;
    std::vector<unsigned> at_wave(b.wave_alive.size(), 0);
    for (unsigned t = 0; t < n; ++t) {
      const Fiber& f = b.fibers[t];
      if (f.done) continue;
      if (f.waiting_on == 0) ++at_block;
      else if (f.waiting_on > 0) ++at_wave[f.waiting_on - 1];
    }
    if (b.alive > 0 && at_block == b.alive) {
      for (Fiber& f : b.fibers) if (!f.done && f.waiting_on == 0) f.waiting_on = -1;
      progressed = true;
    }
    for (size_t w = 0; w < at_wave.size(); ++w)
      if (b.wave_alive[w] > 0 && at_wave[w] == b.wave_alive[w]) {
        for (Fiber& f : b.fibers) if (!f.done && f.waiting_on == (int)w + 1) f.waiting_on = -1;
        progressed = true;
      }
    if (!progressed && b.alive > 0) {
      std::fprintf(stderr, "hip_emu: deadlock (work-items wait on different barriers)\n");
      std::abort();
    }
  }
  g_block = nullptr;
  g_fiber = nullptr;
}

template <typename F>
void launch(dim3 grid, dim3 block, F&& body_in) {
  std::function<void()> body = body_in;
  const unsigned n = block.x * block.y * block.z;
  static thread_local Block blk;  // (stacks are kept between launches)
  blk.fibers.resize(n);
  blk.bdim = block;
  blk.gdim = grid;
  const size_t stack_bytes = 256 * 1024;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        blk.bid = dim3(bx, by, bz);
        run_block(blk, body, stack_bytes);
      }
}

template <typename T>
inline T shfl_up(T v, unsigned delta, int /*width*/) {
  static_assert(sizeof(T) <= 8, "exchange slot");
  Fiber& me = *g_fiber;
  Block& b = *g_block;
  unsigned long long bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  me.slot = bits;
  const int wave_id = 1 + (int)(me.flat / 64);
  barrier_wait(wave_id);   // every live lane of the wave has offered its value
  const unsigned lane = me.flat & 63u;
  const unsigned long long got = b.fibers[lane >= delta ? me.flat - delta : me.flat].slot;
  barrier_wait(wave_id);   // every lane has read before anyone offers again
  T r;
  std::memcpy(&r, &got, sizeof(T));
  return r;
}

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
#define __shared__ static
#define __launch_bounds__(...)
#define __syncthreads() (svo_emu::barrier_wait(0))
#define __shfl_up(...) svo_emu::shfl_up(__VA_ARGS__)
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
#define hipLaunchKernelGGL(kernel, grid, block, shmem, stream, ...) \
  svo_emu::launch((grid), (block), [&] { kernel(__VA_ARGS__); })
