// pose_exchange.h -- the one exchange step of a camera rig on an 8 x MI355X node:
// one process per GPU, one svo::FrameHandlerMono (one camera) per process, and after every frame set an
// all-gather of the SE(3) results over RCCL / xGMI.  Nothing else ever crosses between the ranks:
// pyramids, maps and seeds stay local to their GPU.
//
// Plain C++11 host code.  Ranks find each other through the environment every launcher of
// one-process-per-GPU jobs exports (RANK, WORLD_SIZE, LOCAL_RANK, MASTER_ADDR; torchrun, mpirun
// wrappers, SLURM scripts): rank 0 creates the RCCL unique id and hands it to the others over a TCP
// socket on MASTER_ADDR : SVO_RIG_PORT (default MASTER_PORT + 17, or 29617).
#ifndef SVO_HIP_RIG_POSE_EXCHANGE_H_
#define SVO_HIP_RIG_POSE_EXCHANGE_H_

#include <cstddef>
#include <string>

namespace svo_hip {

// rank 0 -> every other rank: `bytes` bytes of `blob` (filled on rank 0, received elsewhere).  Blocking;
// throws std::runtime_error on timeout (seconds).  Exposed for tests: it needs no GPU.
void tcpBroadcast(int rank, int world, const std::string& addr, int port, void* blob, size_t bytes, int timeout_s = 60);

class PoseExchange {
 public:
  // reads RANK / WORLD_SIZE / LOCAL_RANK / MASTER_ADDR / MASTER_PORT / SVO_RIG_PORT; the process must
  // already have selected its GPU (svo_hip_set_device(LOCAL_RANK))
  static PoseExchange* fromEnvironment();
  PoseExchange(int rank, int world, const std::string& addr, int port);
  ~PoseExchange();
  int rank() const { return rank_; }
  int world() const { return world_; }
  // d_all[r * n .. r * n + n) <- rank r's d_local[0 .. n), doubles in device memory, enqueued on `stream`
  // (a hipStream_t as void*, like the rest of the C ABI): e.g. n = 12 for [R|t], 48 with the covariance
  void allGather(const double* d_local, double* d_all, int n, void* stream);

 private:
  int rank_, world_;
  void* comm_;  // ncclComm_t
};

}  // namespace svo_hip
#endif
