// tests/host/test_bootstrap.cpp -- TEST INFRASTRUCTURE: the TCP hand-over of the RCCL unique id
// (svo_hip::tcpBroadcast, rpg_svo_amd/host/rig/pose_exchange.cpp) between `world` processes, no GPU.
//   test_bootstrap <rank> <world> <port>   prints the received 128 bytes' checksum
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "pose_exchange.h"

int main(int argc, char** argv) {
  if (argc < 4) return 2;
  const int rank = std::atoi(argv[1]), world = std::atoi(argv[2]), port = std::atoi(argv[3]);
  unsigned char blob[128];
  std::memset(blob, 0, sizeof(blob));
  if (rank == 0)
    for (int i = 0; i < 128; ++i) blob[i] = (unsigned char)(37 * i + 11);
  try {
    svo_hip::tcpBroadcast(rank, world, "127.0.0.1", port, blob, sizeof(blob), 20);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "rank %d: %s\n", rank, e.what());
    return 1;
  }
  unsigned sum = 0;
  for (int i = 0; i < 128; ++i) sum = sum * 131u + blob[i];
  std::printf("%u\n", sum);
  return 0;
}
