// xwg_exchange_ubench.hip -- what an exchange of eight f64 partials between the FOUR workgroups of a frame costs on
// gfx950, i.e. the price of splitting a 1000-patch frame of BASELINE configs[3] (64 frames: 64 of 256 CUs busy) over
// four CUs.  The protocol a split K1 would run once per Gauss-Newton iteration:
//     every workgroup writes its 8 partials + a flag to its slot of the frame's exchange block (agent scope), arrives
//     at a generation counter (atomic add, release), spins until all four have arrived (acquire), reads the four
//     slots in a fixed order (deterministic sums) -- no second barrier: every workgroup solves redundantly.
// Measured: microseconds per exchange with F frames x 4 workgroups in flight, for the two placements of a frame's
// workgroups -- consecutive ids (they land on four different XCDs: workgroup id % 8 is the XCD) and ids 8 apart (the same
// XCD, one L2) -- and, as the baseline, the same loop with a workgroup barrier instead of the exchange.
// Round 6 adds mode 3: the four parts of a frame on ONE XCD exchanging through that XCD's L2 only --
// no agent-scope release / acquire (which on a multi-XCD part goes past the L2), but sc0 stores and loads (workgroup scope
// in the ISA's terms: they bypass the CU's vector L1 and are served by the L2 the four CUs share), and no counter: lane l
// of a part stores {partial l, generation} as ONE 16-byte chunk, the 32 lanes (part, l) of every workgroup poll their
// chunk until it carries this round's generation (s_sleep back-off) -- data and flag arrive together, one store and one
// load round trip per exchange.  Placement relies on workgroup id % 8 being the XCD, as the library's XCD-aware launches do.
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

struct Chunk {  // mode 3: a partial and the generation it belongs to, written and read as one 16-byte access
  double v;
  unsigned long long gen;
};
struct BlockL2 {  // one frame: [buffer half][part][partial]
  Chunk c[2][4][8];
};
// STORE: 0: sc0 (the line stays in the XCD's L2), 1: sc1 (write-through: the agent-scope form, placement-independent)
template <int STORE>
__device__ __forceinline__ void store_sc0_x4(Chunk* p, double v, unsigned long long gen) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 w;
  w.x = (unsigned)__double_as_longlong(v); w.y = (unsigned)((unsigned long long)__double_as_longlong(v) >> 32);
  w.z = (unsigned)gen; w.w = (unsigned)(gen >> 32);
  // (s_nop: the store reads its 16 bytes of data for a few cycles after issue, and the compiler does not see the hazard inside asm)
  if (STORE == 0) asm volatile("global_store_dwordx4 %0, %1, off sc0\n\ts_nop 3" ::"v"(p), "v"(w) : "memory");
  else asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 3" ::"v"(p), "v"(w) : "memory");
}
// POLICY: the cache-policy bits of the polling load -- 0: sc0 (workgroup scope), 1: sc1 (agent scope), 2: sc0 sc1 (system scope)
template <int POLICY>
__device__ __forceinline__ Chunk load_sc0_x4(const Chunk* p) {
  typedef unsigned int u4 __attribute__((ext_vector_type(4)));
  u4 w;
  if (POLICY == 0) asm volatile("global_load_dwordx4 %0, %1, off sc0\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  else if (POLICY == 1) asm volatile("global_load_dwordx4 %0, %1, off sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  else asm volatile("global_load_dwordx4 %0, %1, off sc0 sc1\n\ts_waitcnt vmcnt(0)" : "=v"(w) : "v"(p) : "memory");
  Chunk c;
  c.v = __longlong_as_double((long long)(((unsigned long long)w.y << 32) | w.x));
  c.gen = ((unsigned long long)w.w << 32) | w.z;
  return c;
}

// modes 3-5: see the header (load policy sc0 / sc1 / sc0 sc1), stores sc0, ids 8 apart per frame (the same XCD);
// modes 6-7: stores sc1 (write-through) + loads sc1 -- the placement-independent form -- with ids 8 apart / consecutive ids
// (four XCDs).  xcc[block] = the XCC the block ran on (the census: is "id % 8 = XCD" what the dispatcher did?).
template <int POLICY, int STORE, bool CONSECUTIVE>
__global__ void __launch_bounds__(256) exchange_l2_kernel(BlockL2* blocks, unsigned* timed_out, int n_frames, int rounds, double* out, long long* cycles,
                                                          int* xcc) {
  const int frame = CONSECUTIVE ? (int)blockIdx.x / 4 : (int)(blockIdx.x / 32) * 8 + (int)(blockIdx.x % 8);
  const int part = CONSECUTIVE ? (int)blockIdx.x % 4 : (int)(blockIdx.x / 8) % 4;
  if (threadIdx.x == 0) {
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(id));
    xcc[blockIdx.x] = frame < n_frames ? (int)(id & 15u) : -1;
  }
  if (frame >= n_frames) return;
  BlockL2& b = blocks[frame];
  __shared__ double s_tot[8];
  double acc = 0.0;
  const long long t0 = (long long)__builtin_readcyclecounter();
  for (int r = 0; r < rounds; ++r) {
    const int buf = r & 1;
    const unsigned long long gen = (unsigned long long)(r + 1);
    __syncthreads();  // (the workgroup's own reduction has happened: wave 0 publishes)
    if (threadIdx.x < 8) {
      const double v = (double)(part + 1) * (double)(threadIdx.x + 1) + (double)r;
      store_sc0_x4<STORE>(&b.c[buf][part][threadIdx.x], v, gen);
    }
    if (threadIdx.x < 32) {  // lane = (p, l): poll the chunk of part p, partial l
      const int p = (int)threadIdx.x >> 3, l = (int)threadIdx.x & 7;
      Chunk c = load_sc0_x4<POLICY>(&b.c[buf][p][l]);
      unsigned spins = 0;
      while (c.gen != gen) {
        if (++spins > (1u << 12)) {  // never hang the device: flag and leave (a policy that never sees the store ends here)
          atomicOr(timed_out, 1u);
          break;
        }
        __builtin_amdgcn_s_sleep(1);
        c = load_sc0_x4<POLICY>(&b.c[buf][p][l]);
      }
      // parts in a fixed order (deterministic sums): lanes l, 8 + l, 16 + l, 24 + l of the wave hold them
      double t = c.v;
      const double t1 = __shfl(t, l + 8, 64), t2 = __shfl(t, l + 16, 64), t3 = __shfl(t, l + 24, 64);
      if (p == 0) s_tot[l] = ((t + t1) + t2) + t3;
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
    for (int mode = 0; mode < 8; ++mode) {
      Block* d_blocks;
      double* d_out;
      long long* d_cyc;
      const int n_wg = (mode == 1 || (mode >= 3 && mode != 7)) ? ((F + 7) / 8) * 32 : 4 * F;
      BlockL2* d_l2 = nullptr;
      unsigned* d_to = nullptr;
      int* d_xcc = nullptr;
      CHECK(hipMalloc(&d_xcc, sizeof(int) * (size_t)n_wg));
      if (mode >= 3) {
        CHECK(hipMalloc(&d_l2, sizeof(BlockL2) * (size_t)F));
        CHECK(hipMalloc(&d_to, sizeof(unsigned)));
      }
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
        if (mode >= 3) {
          CHECK(hipMemset(d_l2, 0, sizeof(BlockL2) * (size_t)F));
          CHECK(hipMemset(d_to, 0, sizeof(unsigned)));
        }
        CHECK(hipEventRecord(e0, 0));
        if (mode == 3) hipLaunchKernelGGL((exchange_l2_kernel<0, 0, false>), dim3(n_wg), dim3(256), 0, 0, d_l2, d_to, F, rounds, d_out, d_cyc, d_xcc);
        else if (mode == 4) hipLaunchKernelGGL((exchange_l2_kernel<1, 0, false>), dim3(n_wg), dim3(256), 0, 0, d_l2, d_to, F, rounds, d_out, d_cyc, d_xcc);
        else if (mode == 5) hipLaunchKernelGGL((exchange_l2_kernel<2, 0, false>), dim3(n_wg), dim3(256), 0, 0, d_l2, d_to, F, rounds, d_out, d_cyc, d_xcc);
        else if (mode == 6) hipLaunchKernelGGL((exchange_l2_kernel<1, 1, false>), dim3(n_wg), dim3(256), 0, 0, d_l2, d_to, F, rounds, d_out, d_cyc, d_xcc);
        else if (mode == 7) hipLaunchKernelGGL((exchange_l2_kernel<1, 1, true>), dim3(n_wg), dim3(256), 0, 0, d_l2, d_to, F, rounds, d_out, d_cyc, d_xcc);
        else hipLaunchKernelGGL(exchange_kernel, dim3(n_wg), dim3(256), 0, 0, d_blocks, F, rounds, mode, d_out, d_cyc);
        CHECK(hipEventRecord(e1, 0));
        CHECK(hipEventSynchronize(e1));
        float ms;
        CHECK(hipEventElapsedTime(&ms, e0, e1));
        if (rep > 0 && ms < best_ms) best_ms = ms;
        std::vector<Block> hb((size_t)F);
        CHECK(hipMemcpy(hb.data(), d_blocks, sizeof(Block) * (size_t)F, hipMemcpyDeviceToHost));
        for (int f = 0; f < F; ++f) timed_out |= hb[(size_t)f].timed_out;
        if (mode >= 3) {
          unsigned to = 0;
          CHECK(hipMemcpy(&to, d_to, sizeof(unsigned), hipMemcpyDeviceToHost));
          timed_out |= to;
        }
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
      const char* names[8] = {"parts_on_four_xcds", "parts_on_one_xcd", "workgroup_barrier_only", "parts_on_one_xcd_chunks_load_sc0",
                              "parts_on_one_xcd_chunks_load_sc1", "parts_on_one_xcd_chunks_load_sc0_sc1",
                              "ids_8_apart_chunks_store_sc1_load_sc1", "consecutive_ids_chunks_store_sc1_load_sc1"};
      // census (modes 3-7): the fraction of frames whose four parts reported the same XCC
      double same = -1.0;
      if (mode >= 3) {
        std::vector<int> hx((size_t)n_wg);
        CHECK(hipMemcpy(hx.data(), d_xcc, sizeof(int) * (size_t)n_wg, hipMemcpyDeviceToHost));
        int n_same = 0;
        for (int f = 0; f < F; ++f) {
          int ids[4];
          for (int q = 0; q < 4; ++q) ids[q] = hx[(size_t)(mode == 7 ? 4 * f + q : ((f / 8) * 4 + q) * 8 + f % 8)];
          n_same += ids[0] == ids[1] && ids[1] == ids[2] && ids[2] == ids[3];
        }
        same = (double)n_same / F;
      }
      std::printf(",\n \"frames_%d_%s\": {\"us_per_exchange\": %.3f, \"kernel_ms\": %.4f, \"shader_cycles_per_exchange\": %.0f, \"timed_out\": %u, \"sum_ok\": %s, \"frames_with_all_parts_on_one_xcc_frac\": %.3f}",
                  F, names[mode], 1e3 * best_ms / rounds, best_ms, (double)cmax / rounds, timed_out,
                  mode == 2 ? "true" : (check == expect ? "true" : "false"), same);
      std::fflush(stdout);
      CHECK(hipFree(d_blocks));
      if (d_l2) CHECK(hipFree(d_l2));
      if (d_to) CHECK(hipFree(d_to));
      CHECK(hipFree(d_xcc));
      CHECK(hipFree(d_out));
      CHECK(hipFree(d_cyc));
    }
  }
  std::printf("\n}\n");
  return 0;
}
