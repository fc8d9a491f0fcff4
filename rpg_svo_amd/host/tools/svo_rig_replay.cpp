// svo_rig_replay -- BASELINE configs[4] in C++: a camera rig, one process per camera and GPU, each
// aligning ITS OWN image stream frame by frame through the C ABI (svo_hip_sparse_align, batch 1: frame
// k+1 needs frame k's pose), with an RCCL all-gather of the SE(3) results after every frame set
// (rpg_svo_amd/host/rig/pose_exchange.h).  Launch one process per GPU with RANK / WORLD_SIZE /
// LOCAL_RANK / MASTER_ADDR set, e.g.
//
//   python -m torch.distributed.run --no-python --nproc-per-node 8 --master-addr 127.0.0.1 build/svo_rig_replay [frames]
//
// Every rank renders the same synthetic scene with its own camera motion; rank 0 prints the rig rate
// and checks that every gathered pose is the translation its camera performed.
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <vector>

#include <svo_hip.h>

#include "pose_exchange.h"

#define CK(call)                                                                                   \
  do {                                                                                             \
    int rc_ = (call);                                                                              \
    if (rc_ < 0) { std::fprintf(stderr, "%s -> %s\n", #call, svo_hip_strerror(rc_)); return 1; }   \
  } while (0)

static double texture(double x, double y) {
  return 128.0 + 40.0 * std::sin(0.11 * x + 0.3) * std::cos(0.07 * y) + 35.0 * std::sin(0.023 * x * 0.9 + 0.031 * y) +
         25.0 * std::cos(0.19 * y + 0.05 * x) + 20.0 * std::sin(0.37 * x) * std::sin(0.29 * y + 1.0);
}

int main(int argc, char** argv) {
  const int n_frames = argc > 1 ? std::atoi(argv[1]) : 100;
  const char* lr = std::getenv("LOCAL_RANK");
  const int local_rank = lr ? std::atoi(lr) : 0;
  if (svo_hip_device_count() <= local_rank) { std::fprintf(stderr, "no HIP device for local rank %d\n", local_rank); return 2; }
  CK(svo_hip_set_device(local_rank));
  svo_hip::PoseExchange* ex = NULL;
  try {
    ex = svo_hip::PoseExchange::fromEnvironment();
  } catch (const std::exception& e) {
    std::fprintf(stderr, "svo_rig_replay: %s\n", e.what());
    return 3;
  }
  const int rank = ex->rank(), world = ex->world();

  // this camera: 752x480, the reference's default schedule (5 levels, 4 -> 2), 120 patches; the content
  // of frame k is the texture shifted by k * (sx, sy) pixels = a sideways translation at depth Z
  const int W = 752, H = 480, LEVELS = 5, N = 120;
  const double fx = 315.5, fy = 315.5, cx = 376, cy = 240, Z = 2.0;
  const double sx = 1.3 + 0.2 * rank, sy = -0.9 + 0.1 * rank;
  svo_hip_pyr_layout L;
  CK(svo_hip_pyr_layout_init(W, H, LEVELS, &L));
  void* stream = NULL;
  CK(svo_hip_stream_create(&stream));
  void* d_store = NULL;
  const size_t store_bytes = (size_t)svo_hip_pyr_store_bytes(&L, n_frames + 1);
  CK(svo_hip_malloc(&d_store, store_bytes));
  CK(svo_hip_memset(d_store, 0, store_bytes, stream));
  std::vector<uint8_t> img((size_t)W * H);
  void* d_stage = NULL;  // packed image on its way into the tiled store
  CK(svo_hip_malloc(&d_stage, (size_t)W * H));
  for (int k = 0; k <= n_frames; ++k) {
    for (int v = 0; v < H; ++v)
      for (int u = 0; u < W; ++u) img[(size_t)v * W + u] = (uint8_t)std::lround(texture(u - k * sx, v - k * sy));
    CK(svo_hip_pyramid_upload_build(&L, (uint8_t*)d_store, k, img.data(), W, SVO_HIP_HALFSAMPLE_AUTO, d_stage, stream));
    CK(svo_hip_stream_sync(stream));  // img and d_stage are reused by the next frame
  }

  std::vector<double> px((size_t)N * 2), xyz((size_t)N * 3), Tin(12, 0.0);
  for (int i = 0; i < N; ++i) {
    const double u = 90.0 + 38.0 * (i % 15), v = 70.0 + 42.0 * (i / 15);
    px[2 * i] = u; px[2 * i + 1] = v;
    xyz[3 * i] = (u - cx) / fx * Z; xyz[3 * i + 1] = (v - cy) / fy * Z; xyz[3 * i + 2] = Z;
  }
  Tin[0] = Tin[4] = Tin[8] = 1.0;
  void *d_px, *d_xyz, *d_Tin, *d_Tout, *d_all, *d_slots, *d_n, *d_ntr;
  CK(svo_hip_malloc(&d_px, px.size() * 8)); CK(svo_hip_malloc(&d_xyz, xyz.size() * 8));
  CK(svo_hip_malloc(&d_Tin, 96)); CK(svo_hip_malloc(&d_Tout, 96)); CK(svo_hip_malloc(&d_all, (size_t)world * 96));
  CK(svo_hip_malloc(&d_slots, (size_t)(n_frames + 1) * 4)); CK(svo_hip_malloc(&d_n, 4)); CK(svo_hip_malloc(&d_ntr, 4));
  std::vector<int32_t> slots(n_frames + 1);
  for (int k = 0; k <= n_frames; ++k) slots[k] = k;
  const int32_t n = N;
  CK(svo_hip_memcpy_h2d(d_px, px.data(), px.size() * 8, stream)); CK(svo_hip_memcpy_h2d(d_xyz, xyz.data(), xyz.size() * 8, stream));
  CK(svo_hip_memcpy_h2d(d_Tin, Tin.data(), 96, stream)); CK(svo_hip_memcpy_h2d(d_slots, slots.data(), slots.size() * 4, stream));
  CK(svo_hip_memcpy_h2d(d_n, &n, 4, stream));
  CK(svo_hip_stream_sync(stream));

  svo_hip_sia_params P = {fx, fy, cx, cy, /*max_level=*/4, /*min_level=*/2, /*n_iter=*/30, SVO_HIP_CAM_PINHOLE, /*eps=*/1e-6, {0, 0, 0, 0, 0}};
  std::vector<double> all((size_t)world * 12);
  double worst = 0;
  const auto t0 = std::chrono::steady_clock::now();
  for (int k = 0; k < n_frames; ++k) {
    CK(svo_hip_sparse_align(&L, (const uint8_t*)d_store, 1, (const int32_t*)d_slots + k, (const int32_t*)d_slots + k + 1, (const int32_t*)d_n, N,
                            (const double*)d_px, (const double*)d_xyz, NULL, &P, (const double*)d_Tin, (double*)d_Tout, NULL, (int32_t*)d_ntr,
                            NULL, NULL, NULL, stream));
    try {
      ex->allGather((const double*)d_Tout, (double*)d_all, 12, stream);  // the only exchange step of the rig
    } catch (const std::exception& e) {
      std::fprintf(stderr, "svo_rig_replay: %s\n", e.what());
      return 3;
    }
    CK(svo_hip_memcpy_d2h(all.data(), d_all, all.size() * 8, stream));
    CK(svo_hip_stream_sync(stream));  // the host consumes the rig's poses before the next frame set
    for (int r = 0; r < world; ++r) {
      const double tx = (1.3 + 0.2 * r) * Z / fx, ty = (-0.9 + 0.1 * r) * Z / fy;
      const double* T = &all[(size_t)r * 12];
      const double e = std::sqrt((T[9] - tx) * (T[9] - tx) + (T[10] - ty) * (T[10] - ty) + T[11] * T[11]);
      if (e > worst) worst = e;
    }
  }
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  if (rank == 0)
    std::printf("{\"cameras\": %d, \"frames_per_camera\": %d, \"rig_frames_per_s\": %.1f, \"us_per_frame_set\": %.1f, "
                "\"gather_bytes_per_frame_set\": %d, \"worst_translation_error_m\": %.3e}\n",
                world, n_frames, world * n_frames / sec, sec / n_frames * 1e6, world * 96, worst);
  delete ex;
  for (void* p : {d_px, d_xyz, d_Tin, d_Tout, d_all, d_slots, d_n, d_ntr, d_store}) svo_hip_free(p);
  svo_hip_stream_destroy(stream);
  return worst < 2e-3 ? 0 : 1;
}
