// examples/sparse_align_batch.cpp -- the C ABI of include/svo_hip.h from plain C++ (g++, no HIP
// headers, no torch, no reference headers): build image pyramids on the device, align a batch of
// (reference, current) problems, read the poses back.
//
// Scene: a fronto-parallel textured plane at depth Z seen by a pinhole camera; the current image
// is the reference image shifted by (sx, sy) pixels, which is what a camera translation of
// (-sx*Z/fx, -sy*Z/fy, 0) produces.  SparseImgAlign starts from the identity and must recover it.
//
//   g++ -std=c++11 -O2 -I include examples/sparse_align_batch.cpp -L rpg_svo_amd/lib -lsvo_hip
//       -Wl,-rpath,$PWD/rpg_svo_amd/lib -o build/sparse_align_batch   (one command line), then
//   build/sparse_align_batch [B]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

#include <svo_hip.h>

#define CK(call)                                                                          \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ < 0) { std::fprintf(stderr, "%s -> %s\n", #call, svo_hip_strerror(rc_)); return 1; } \
  } while (0)

static double texture(double x, double y) {  // smooth, non-periodic enough, gradients everywhere
  return 128.0 + 40.0 * std::sin(0.11 * x + 0.3) * std::cos(0.07 * y) + 35.0 * std::sin(0.023 * x * 0.9 + 0.031 * y) +
         25.0 * std::cos(0.19 * y + 0.05 * x) + 20.0 * std::sin(0.37 * x) * std::sin(0.29 * y + 1.0);
}

int main(int argc, char** argv) {
  const int B = argc > 1 ? std::atoi(argv[1]) : 1024;
  const int W = 640, H = 480, LEVELS = 4, N = 200;
  const double fx = 400, fy = 400, cx = 320, cy = 240, Z = 2.0, sx = 2.6, sy = -1.7;
  if (svo_hip_device_count() <= 0) { std::fprintf(stderr, "no HIP device: there is no CPU fallback\n"); return 2; }
  CK(svo_hip_set_device(0));

  // two images on the host, level 0 only
  std::vector<uint8_t> ref((size_t)W * H), cur((size_t)W * H);
  for (int v = 0; v < H; ++v)
    for (int u = 0; u < W; ++u) {
      ref[(size_t)v * W + u] = (uint8_t)std::lround(texture(u, v));
      cur[(size_t)v * W + u] = (uint8_t)std::lround(texture(u - sx, v - sy));  // content moved by (+sx, +sy)
    }
  // pyramid store with two slots; K0 fills them on the device
  svo_hip_pyr_layout L;
  CK(svo_hip_pyr_layout_init(W, H, LEVELS, &L));
  void* d_store = NULL;
  CK(svo_hip_malloc(&d_store, (size_t)svo_hip_pyr_store_bytes(&L, 2)));
  CK(svo_hip_memset(d_store, 0, (size_t)svo_hip_pyr_store_bytes(&L, 2), NULL));
  void* stream = NULL;
  CK(svo_hip_stream_create(&stream));
  // image -> packed device scratch (NULL: a stream-ordered temporary) -> one kernel: tiled level 0 + levels 1..
  CK(svo_hip_pyramid_upload_build(&L, (uint8_t*)d_store, 0, ref.data(), W, SVO_HIP_HALFSAMPLE_AUTO, NULL, stream));
  CK(svo_hip_pyramid_upload_build(&L, (uint8_t*)d_store, 1, cur.data(), W, SVO_HIP_HALFSAMPLE_AUTO, NULL, stream));

  // N features on a grid of the reference frame, all at depth Z: xyz_ref = f * range
  std::vector<double> px((size_t)B * N * 2), xyz((size_t)B * N * 3), Tin((size_t)B * 12, 0.0);
  std::vector<int32_t> rs(B, 0), cs(B, 1), n(B, N);
  for (int b = 0; b < B; ++b) {
    for (int i = 0; i < N; ++i) {
      const double u = 60.0 + 26.0 * (i % 20) + 0.3 * (b % 7), v = 60.0 + 38.0 * (i / 20) + 0.2 * (b % 5);
      const double x = (u - cx) / fx, y = (v - cy) / fy;
      px[((size_t)b * N + i) * 2] = u; px[((size_t)b * N + i) * 2 + 1] = v;
      xyz[((size_t)b * N + i) * 3] = x * Z; xyz[((size_t)b * N + i) * 3 + 1] = y * Z; xyz[((size_t)b * N + i) * 3 + 2] = Z;
    }
    Tin[(size_t)b * 12 + 0] = Tin[(size_t)b * 12 + 4] = Tin[(size_t)b * 12 + 8] = 1.0;  // identity prior
  }
  void *d_px, *d_xyz, *d_Tin, *d_Tout, *d_rs, *d_cs, *d_n, *d_ntr, *d_it;
  CK(svo_hip_malloc(&d_px, px.size() * 8)); CK(svo_hip_malloc(&d_xyz, xyz.size() * 8));
  CK(svo_hip_malloc(&d_Tin, Tin.size() * 8)); CK(svo_hip_malloc(&d_Tout, Tin.size() * 8));
  CK(svo_hip_malloc(&d_rs, B * 4)); CK(svo_hip_malloc(&d_cs, B * 4)); CK(svo_hip_malloc(&d_n, B * 4));
  CK(svo_hip_malloc(&d_ntr, B * 4)); CK(svo_hip_malloc(&d_it, (size_t)B * SVO_HIP_MAX_LEVELS * 4));
  CK(svo_hip_memcpy_h2d(d_px, px.data(), px.size() * 8, stream)); CK(svo_hip_memcpy_h2d(d_xyz, xyz.data(), xyz.size() * 8, stream));
  CK(svo_hip_memcpy_h2d(d_Tin, Tin.data(), Tin.size() * 8, stream));
  CK(svo_hip_memcpy_h2d(d_rs, rs.data(), B * 4, stream)); CK(svo_hip_memcpy_h2d(d_cs, cs.data(), B * 4, stream));
  CK(svo_hip_memcpy_h2d(d_n, n.data(), B * 4, stream));

  svo_hip_sia_params P = {fx, fy, cx, cy, /*max_level=*/3, /*min_level=*/0, /*n_iter=*/30, 0, /*eps=*/1e-6};
  auto run = [&]() {
    return svo_hip_sparse_align(&L, (const uint8_t*)d_store, B, (const int32_t*)d_rs, (const int32_t*)d_cs, (const int32_t*)d_n, N,
                                (const double*)d_px, (const double*)d_xyz, NULL, &P, (const double*)d_Tin, (double*)d_Tout, NULL,
                                (int32_t*)d_ntr, (int32_t*)d_it, NULL, NULL, stream);
  };
  CK(run());
  CK(svo_hip_stream_sync(stream));
  const int reps = 20;
  const auto t0 = std::chrono::steady_clock::now();
  for (int r = 0; r < reps; ++r) CK(run());
  CK(svo_hip_stream_sync(stream));
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();

  std::vector<double> Tout((size_t)B * 12);
  std::vector<int32_t> ntr(B);
  CK(svo_hip_memcpy_d2h(Tout.data(), d_Tout, Tout.size() * 8, stream));
  CK(svo_hip_memcpy_d2h(ntr.data(), d_ntr, B * 4, stream));
  CK(svo_hip_stream_sync(stream));
  // a point (x,y,Z) of the reference frame appears sx pixels further right in the current image:
  // p_cur = p_ref + t with t = (sx*Z/fx, sy*Z/fy, 0)
  const double tx = sx * Z / fx, ty = sy * Z / fy;
  double worst = 0;
  int min_tracked = N;
  for (int b = 0; b < B; ++b) {
    const double* T = &Tout[(size_t)b * 12];
    const double e = std::sqrt((T[9] - tx) * (T[9] - tx) + (T[10] - ty) * (T[10] - ty) + T[11] * T[11]) + std::fabs(T[0] - 1) + std::fabs(T[4] - 1) +
                     std::fabs(T[8] - 1);
    if (e > worst) worst = e;
    if (ntr[b] < min_tracked) min_tracked = ntr[b];
  }
  std::printf("B=%d problems x %d patches: %.0f frames/s; translation recovered (%.5f, %.5f, %.5f) vs (%.5f, %.5f, 0), worst error %.2e, "
              "min tracked %d\n", B, N, B * reps / sec, Tout[9], Tout[10], Tout[11], tx, ty, worst, min_tracked);
  for (void* p : {d_px, d_xyz, d_Tin, d_Tout, d_rs, d_cs, d_n, d_ntr, d_it, d_store}) svo_hip_free(p);
  svo_hip_stream_destroy(stream);
  if (!(worst < 2e-3) || min_tracked < N - 5) { std::fprintf(stderr, "FAILED\n"); return 1; }
  std::puts("OK");
  return 0;
}
