// simt_emu.h -- TEST INFRASTRUCTURE.  Just enough of a wave for the kernels' group-of-eight code to run on the CPU: one host
// thread per lane of a group, the cross-lane moves (DPP quad permutes, row_shr:1, half mirror, __shfl, __shfl_xor) served
// by an exchange through a per-group array between two barriers, the same-wave LDS hand-over (fence + wave_barrier) by
// a barrier.  Lanes of a group follow the same control flow in the code under test (uniform trip counts, uniform
// branches around the moves), so every lane reaches every exchange; groups are independent.
// Include BEFORE the device headers, with SVO_HOST_MATH_TEST defined.
#pragma once
#include <algorithm>
#include <condition_variable>
#include <cstdint>
#include <cstring>
#include <mutex>

namespace svo_emu {

constexpr int GROUP = 8;

class Barrier {
 public:
  explicit Barrier(int n) : n_(n), waiting_(0), phase_(0) {}
  void wait() {
    std::unique_lock<std::mutex> lk(m_);
    const unsigned long ph = phase_;
    if (++waiting_ == n_) {
      waiting_ = 0;
      ++phase_;
      cv_.notify_all();
    } else {
      cv_.wait(lk, [&] { return phase_ != ph; });
    }
  }

 private:
  std::mutex m_;
  std::condition_variable cv_;
  int n_, waiting_;
  unsigned long phase_;
};

struct Group {
  Barrier bar;
  unsigned long long slot[GROUP];
  Group() : bar(GROUP) {}
};

struct Tid {
  unsigned x;
};
inline thread_local Group* t_group = nullptr;
inline thread_local int t_lane = 0;        // lane inside the group
inline thread_local unsigned t_thread = 0;  // threadIdx.x of the emulated workgroup
inline Tid tid() { return Tid{t_thread}; }

// value of lane `src` of the group (src < 0: the hardware's bound_ctrl zero)
template <typename T>
inline T from_lane(T v, int src) {
  static_assert(sizeof(T) <= 8, "exchange slot");
  unsigned long long bits = 0;
  std::memcpy(&bits, &v, sizeof(T));
  t_group->slot[t_lane] = bits;
  t_group->bar.wait();
  unsigned long long got = src >= 0 ? t_group->slot[src] : 0ull;
  t_group->bar.wait();
  T r;
  std::memcpy(&r, &got, sizeof(T));
  return src >= 0 ? r : T(0);
}

inline int update_dpp(int /*old*/, int v, int ctrl, int /*row_mask*/, int /*bank_mask*/, bool /*bound_ctrl*/) {
  switch (ctrl) {
    case 0x111: return from_lane(v, t_lane - 1);       // row_shr:1 (lane 0 of the group: the caller substitutes its carry)
    case 0xB1: return from_lane(v, t_lane ^ 1);        // quad_perm [1,0,3,2]
    case 0x4E: return from_lane(v, t_lane ^ 2);        // quad_perm [2,3,0,1]
    case 0x141: return from_lane(v, GROUP - 1 - t_lane);  // row_half_mirror
    default: __builtin_trap();
  }
}
template <typename T>
inline T shfl(T v, int src_lane_of_wave, int /*width*/) { return from_lane(v, src_lane_of_wave & (GROUP - 1)); }
template <typename T>
inline T shfl_xor(T v, int mask, int /*width*/) { return from_lane(v, t_lane ^ mask); }
template <typename T>
inline T shfl_up(T v, int delta, int /*width*/) { return from_lane(v, t_lane - delta); }
inline void lds_handover() { t_group->bar.wait(); }

inline uint32_t udot4(uint32_t a, uint32_t b, uint32_t c, bool /*clamp*/) {
  for (int k = 0; k < 4; ++k) c += ((a >> (8 * k)) & 0xffu) * ((b >> (8 * k)) & 0xffu);
  return c;
}
inline uint32_t alignbyte(uint32_t hi, uint32_t lo, uint32_t sel) {
  return (uint32_t)(((((uint64_t)hi) << 32) | (uint64_t)lo) >> (8u * (sel & 3u)));
}

}  // namespace svo_emu

using std::max;
using std::min;
struct alignas(16) uint4 {
  uint32_t x, y, z, w;
};
inline uint4 make_uint4(uint32_t x, uint32_t y, uint32_t z, uint32_t w) { return uint4{x, y, z, w}; }
#define threadIdx (svo_emu::tid())
#define __builtin_amdgcn_update_dpp(...) svo_emu::update_dpp(__VA_ARGS__)
#define __shfl(...) svo_emu::shfl(__VA_ARGS__)
#define __shfl_xor(...) svo_emu::shfl_xor(__VA_ARGS__)
#define __shfl_up(...) svo_emu::shfl_up(__VA_ARGS__)
#define __builtin_amdgcn_udot4(...) svo_emu::udot4(__VA_ARGS__)
#define __builtin_amdgcn_alignbyte(...) svo_emu::alignbyte(__VA_ARGS__)
#define __builtin_amdgcn_fence(...) ((void)0)
#define __builtin_amdgcn_wave_barrier() svo_emu::lds_handover()
