// emu_race_probe.cpp -- TEST INFRASTRUCTURE: the check of the checker for the ThreadSanitizer build of the emulator
// (tests/host/hip_emu.h, tests/test_emulated_sanitized.py).  One workgroup of two waves stores a value per work-item to
// LDS and reads a neighbour's: with the workgroup barrier in between (clean), with nothing (a race), with only the
// wave-wide hand-over although the neighbour of lane 63 lives in the other wave (a race: the bug class "the hand-over's
// scope is smaller than the exchange"), and with the wave-wide hand-over for a neighbour inside the wave (clean).
#include "hip_emu.h"

namespace {
template <int MODE>
__global__ void neighbour_sum(const int* in, int* out) {
  __shared__ int s[128];
  const int t = (int)threadIdx.x;
  s[t] = in[t];
  if (MODE == 0) __syncthreads();
  if (MODE == 2 || MODE == 3) SVO_WAVE_LDS_HANDOVER();
  const int nb = MODE == 3 ? ((t & ~63) | ((t + 1) & 63)) : ((t + 1) & 127);
  out[t] = s[t] + s[nb];
}
// two workgroups: the second one adds 1 to the neighbour's value -- read from the input (MODE 4: clean) or from what the FIRST
// workgroup wrote (MODE 5: nothing orders two workgroups of a launch; seen only with SVO_EMU_TSAN_BETWEEN_WORKGROUPS=1)
template <int MODE>
__global__ void two_workgroups(const int* in, int* out) {
  const int t = (int)threadIdx.x, b = (int)blockIdx.x;
  if (b == 0) out[t] = in[t];
  else out[128 + t] = (MODE == 5 ? out[(t + 1) & 127] : in[(t + 1) & 127]) + 1;
}
}  // namespace

extern "C" int probe_neighbour_sum(int mode, const int* in, int* out) {
  switch (mode) {
    case 0: hipLaunchKernelGGL(neighbour_sum<0>, dim3(1), dim3(128), 0, nullptr, in, out); break;
    case 1: hipLaunchKernelGGL(neighbour_sum<1>, dim3(1), dim3(128), 0, nullptr, in, out); break;
    case 2: hipLaunchKernelGGL(neighbour_sum<2>, dim3(1), dim3(128), 0, nullptr, in, out); break;
    case 3: hipLaunchKernelGGL(neighbour_sum<3>, dim3(1), dim3(128), 0, nullptr, in, out); break;
    case 4: hipLaunchKernelGGL(two_workgroups<4>, dim3(2), dim3(128), 0, nullptr, in, out); break;
    case 5: hipLaunchKernelGGL(two_workgroups<5>, dim3(2), dim3(128), 0, nullptr, in, out); break;
    default: return -1;
  }
  return 0;
}
