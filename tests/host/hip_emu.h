// hip_emu.h -- TEST INFRASTRUCTURE.  Whole kernels on the CPU: enough of the HIP programming model for the kernels of
// rpg_svo_amd/csrc that use a workgroup's barrier, LDS and the plain cross-lane moves to be compiled by the host compiler
// and launched through their own C-ABI entry points.  A launch runs the workgroups one after the other, each as one host
// thread per work-item; __syncthreads is a barrier that threads which have left the kernel drop out of (as exited waves
// do), __shared__ is `static` (one workgroup at a time), __shfl_up an exchange inside the 64 threads of a wave, atomicAdd
// a host atomic.  No timing, no memory model subtleties: what this checks is the kernels' LOGIC against the oracle.
// Include first in the test's translation unit (it defines SVO_HOST_MATH_TEST for the product headers).
#pragma once
#define SVO_HOST_MATH_TEST

#include <algorithm>
#include <cmath>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <memory>
#include <mutex>
#include <thread>
#include <vector>

struct dim3 {
  unsigned x, y, z;
  dim3(unsigned x_ = 1, unsigned y_ = 1, unsigned z_ = 1) : x(x_), y(y_), z(z_) {}
};
typedef int hipError_t;
constexpr hipError_t hipSuccess = 0;
inline hipError_t hipGetLastError() { return hipSuccess; }
typedef void* hipStream_t;
struct alignas(16) uint4 {
  uint32_t x, y, z, w;
};
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }

namespace svo_emu {

// a barrier whose participants may leave for good
class Barrier {
 public:
  explicit Barrier(int n) : n_(n), waiting_(0), phase_(0) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    const unsigned long ph = phase_;
    if (++waiting_ >= n_) release();
    else cv_.wait(lk, [&] { return phase_ != ph; });
  }
  void drop() {
    std::unique_lock<std::mutex> lk(m_);
    --n_;
    if (n_ > 0 && waiting_ >= n_) release();
  }

 private:
  void release() {
    waiting_ = 0;
    ++phase_;
    cv_.notify_all();
  }
  std::mutex m_;
  std::condition_variable cv_;
  int n_, waiting_;
  unsigned long phase_;
};

struct Block {
  Barrier bar;
  std::vector<std::unique_ptr<Barrier>> wave_bar;
  std::vector<unsigned long long> slot;
  explicit Block(unsigned n) : bar((int)n), slot(n) {
    for (unsigned w = 0; w * 64 < n; ++w) wave_bar.emplace_back(new Barrier((int)std::min(64u, n - w * 64)));
  }
};

inline thread_local dim3 t_threadIdx, t_blockIdx, t_blockDim, t_gridDim;
inline thread_local Block* t_block = nullptr;
inline thread_local unsigned t_flat = 0;

template <typename F>
void launch(dim3 grid, dim3 block, F&& body) {
  const unsigned n = block.x * block.y * block.z;
  for (unsigned bz = 0; bz < grid.z; ++bz)
    for (unsigned by = 0; by < grid.y; ++by)
      for (unsigned bx = 0; bx < grid.x; ++bx) {
        Block blk(n);
        std::vector<std::thread> threads;
        threads.reserve(n);
        for (unsigned t = 0; t < n; ++t)
          threads.emplace_back([&, t] {
            t_block = &blk;
            t_flat = t;
            t_threadIdx = dim3(t % block.x, (t / block.x) % block.y, t / (block.x * block.y));
            t_blockIdx = dim3(bx, by, bz);
            t_blockDim = block;
            t_gridDim = grid;
            body();
            blk.wave_bar[t / 64]->drop();  // (an exited work-item no longer takes part in barriers)
            blk.bar.drop();
          });
        for (auto& th : threads) th.join();
      }
}

template <typename T>
inline T shfl_up(T v, unsigned delta, int /*width*/) {
  static_assert(sizeof(T) <= 8, "exchange slot");
  Block& b = *t_block;
  unsigned long long bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  b.slot[t_flat] = bits;
  Barrier& wb = *b.wave_bar[t_flat / 64];
  wb.wait();
  const unsigned lane = t_flat & 63u;
  const unsigned long long got = b.slot[lane >= delta ? t_flat - delta : t_flat];
  wb.wait();
  T r;
  std::memcpy(&r, &got, sizeof(T));
  return r;
}

}  // namespace svo_emu

using std::max;
using std::min;
#define threadIdx (svo_emu::t_threadIdx)
#define blockIdx (svo_emu::t_blockIdx)
#define blockDim (svo_emu::t_blockDim)
#define gridDim (svo_emu::t_gridDim)
#define __device__
#define __host__
#define __forceinline__ inline __attribute__((always_inline))
#define __constant__ static const
#define __global__
#define __shared__ static
#define __launch_bounds__(...)
#define __syncthreads() (svo_emu::t_block->bar.wait())
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
