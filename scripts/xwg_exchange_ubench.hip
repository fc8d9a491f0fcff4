// xwg_exchange_ubench.hip -- what an exchange of eight f64 partials between the FOUR workgroups of a frame costs on
// gfx950, i.e. the price of splitting a 1000-patch frame of BASELINE configs[3] (64 frames: 64 of 256 CUs busy) over
// four CUs.  The protocol a split K1 would run once per Gauss-Newton iteration:
//     every workgroup writes its 8 partials + a flag to its slot of the frame's exchange block (agent scope), arrives
//     at a generation counter (atomic add, release), spins until all four have arrived (acquire), reads the four
//     slots in a fixed order (deterministic sums) -- no second barrier: every workgroup solves redundantly.
// Measured: microseconds per exchange with F frames x 4 workgroups in flight, for the two placements of a frame's
// workgroups -- consecutive ids (they land on four different XCDs: workgroup id % 8 is the XCD) and ids 8 apart (the same
// XCD, one L2) -- and, as the baseline, the same loop with a workgroup barrier instead of the exchange.
//   hipcc --offload-arch=gfx950 -O2 scripts/xwg_exchange_ubench.hip -o build/xwg_exchange_ubench && build/xwg_exchange_ubench
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHECK(x)                                                                      \
  do {                                                                                \
    hipError_t e_ = (x);                                                              \
    if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); std::exit(1); } \
  } while (0)

struct Block {  // one frame's exchange block (one 256-byte line per buffer half and part would be kinder; this is the naive layout)
  double part[2][4][8];
  unsigned count;  // generation counter: 4 arrivals per exchange
  unsigned timed_out;
  unsigned pad[30];
};

// mode 0: consecutive ids per frame; 1: ids 8 apart (same XCD); 2: no exchange, __syncthreads only
__global__ void __launch_bounds__(256) exchange_kernel(Block* blocks, int n_frames, int rounds, int mode, double* out, long long* cycles) {
  int frame, part;
  if (mode == 1) {
    // id = ((f / 8) * 4 + p) * 8 + f % 8
    frame = (int)(blockIdx.x / 32) * 8 + (int)(blockIdx.x % 8);
    part = (int)(blockIdx.x / 8) % 4;
  } else {
    frame = (int)blockIdx.x / 4;
    part = (int)blockIdx.x % 4;
  }
  if (frame >= n_frames) return;
  Block& b = blocks[frame];
  __shared__ double s_tot[8];
  double acc = 0.0;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    const int buf = r & 1;
    if (mode == 2) {
      __syncthreads();
      acc += 1.0;
      continue;
    }
    __syncthreads();  // (the workgroup's own reduction has happened: wave 0 publishes)
    if (threadIdx.x < 8) {
      const double v = (double)(part + 1) * (double)(threadIdx.x + 1) + (double)r;
      __hip_atomic_store(&b.part[buf][part][threadIdx.x], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    if (threadIdx.x == 0) {
      // (lanes 0-7 are one wave: their stores are ordered before this release by program order + the fence)
      __hip_atomic_fetch_add(&b.count, 1u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
      const unsigned target = 4u * (unsigned)(r + 1);
      unsigned spins = 0;
      while (__hip_atomic_load(&b.count, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_AGENT) < target) {
        if (++spins > (1u << 22)) {  // never hang the device: flag and leave
          __hip_atomic_store(&b.timed_out, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
      }
    }
    __syncthreads();
    if (threadIdx.x < 8) {
      double t = 0.0;
      for (int p = 0; p < 4; ++p) t += __hip_atomic_load(&b.part[buf][p][threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
      s_tot[threadIdx.x] = t;
    }
    __syncthreads();
    acc += s_tot[threadIdx.x & 7];
  }
  const long long t1 = (long long)__builtin_readcyclecounter();
  if (threadIdx.x == 0) {
    out[blockIdx.x] = acc;
    cycles[blockIdx.x] = t1 - t0;
  }
}

int main(int argc, char** argv) {
  const int rounds = argc > 1 ? std::atoi(argv[1]) : 200;
  std::printf("{\"rounds\": %d", rounds);
  const int frame_counts[3] = {1, 16, 64};
  for (int fi = 0; fi < 3; ++fi) {
    const int F = frame_counts[fi];
    for (int mode = 0; mode < 3; ++mode) {
      Block* d_blocks;
      double* d_out;
      long long* d_cyc;
      const int n_wg = mode == 1 ? ((F + 7) / 8) * 32 : 4 * F;
      CHECK(hipMalloc(&d_blocks, sizeof(Block) * (size_t)F));
      CHECK(hipMalloc(&d_out, sizeof(double) * (size_t)n_wg));
      CHECK(hipMalloc(&d_cyc, sizeof(long long) * (size_t)n_wg));
      hipEvent_t e0, e1;
      CHECK(hipEventCreate(&e0));
      CHECK(hipEventCreate(&e1));
      float best_ms = 1e30f;
      unsigned timed_out = 0;
      double check = 0.0;
      for (int rep = 0; rep < 5; ++rep) {
        CHECK(hipMemset(d_blocks, 0, sizeof(Block) * (size_t)F));
        CHECK(hipEventRecord(e0, 0));
        hipLaunchKernelGGL(exchange_kernel, dim3(n_wg), dim3(256), 0, 0, d_blocks, F, rounds, mode, d_out, d_cyc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best_ms) best_ms = ms;
        std::vector<Block> hb((size_t)F);
        CHECK(hipMemcpy(hb.data(), d_blocks, sizeof(Block) * (size_t)F, hipMemcpyDeviceToHost));
        for (int f = 0; f < F; ++f) timed_out |= hb[(size_t)f].timed_out;
        std::vector<double> ho((size_t)n_wg);
        CHECK(hipMemcpy(ho.data(), d_out, sizeof(double) * (size_t)n_wg, hipMemcpyDeviceToHost));
        check = ho[0];
      }
      std::vector<long long> hc((size_t)n_wg);
      CHECK(hipMemcpy(hc.data(), d_cyc, sizeof(long long) * (size_t)n_wg, hipMemcpyDeviceToHost));
      long long cmax = 0;
      for (int i = 0; i < n_wg; ++i) cmax = hc[(size_t)i] > cmax ? hc[(size_t)i] : cmax;
      // expected value of acc for lane 0 (column 0): sum over rounds of (1+2+3+4)*1 + 4 r = 10 + 4 r
      double expect = 0.0;
      for (int r = 0; r < rounds; ++r) expect += 10.0 + 4.0 * r;
      const char* names[3] = {"parts_on_four_xcds", "parts_on_one_xcd", "workgroup_barrier_only"};
      std::printf(",\n \"frames_%d_%s\": {\"us_per_exchange\": %.3f, \"kernel_ms\": %.4f, \"shader_cycles_per_exchange\": %.0f, \"timed_out\": %u, \"sum_ok\": %s}",
                  F, names[mode], 1e3 * best_ms / rounds, best_ms, (double)cmax / rounds, timed_out,
                  mode == 2 ? "true" : (check == expect ? "true" : "false"));
      CHECK(hipFree(d_blocks));
      CHECK(hipFree(d_out));
      CHECK(hipFree(d_cyc));
    }
  }
  std::printf("\n}\n");
  return 0;
}
