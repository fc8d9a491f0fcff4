// examples/reproject_map.cpp -- the map-mirror entry point of include/svo_hip.h (row N2) from plain C++ (g++, no HIP
// headers, no torch, no reference headers): fill a svo_hip_map through a patch, call svo_hip_reproject_map for one
// frame, read the visit list back and check it against a host walk written the way Reprojector::reprojectMap does it
// (svo/src/reprojector.cpp:64-142: keyframes closest first, every feature once, then the candidates; points binned into
// grid cells; per cell good before unknown before candidate points, otherwise in binning order).
//
//   g++ -std=c++11 -O2 -I include examples/reproject_map.cpp -L rpg_svo_amd/lib -lsvo_hip
//       -Wl,-rpath,$PWD/rpg_svo_amd/lib -o build/reproject_map   (one command line), then   build/reproject_map
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <svo_hip.h>

#define CK(call)                                                                          \
  do {                                                                                    \
    int rc_ = (call);                                                                     \
    if (rc_ < 0) { std::fprintf(stderr, "%s -> %s\n", #call, svo_hip_strerror(rc_)); return 1; } \
  } while (0)

template <typename T> static T* dev(const std::vector<T>& h, size_t n_min = 1) {
  void* p = NULL;
  const size_t n = std::max(h.size(), n_min);
  if (svo_hip_malloc(&p, n * sizeof(T)) < 0) std::exit(3);
  if (!h.empty() && svo_hip_memcpy_h2d(p, &h[0], h.size() * sizeof(T), NULL) < 0) std::exit(3);
  return static_cast<T*>(p);
}
template <typename T> static std::vector<T> host(const T* d, size_t n) {
  std::vector<T> h(n);
  if (n && svo_hip_memcpy_d2h(&h[0], d, n * sizeof(T), NULL) < 0) std::exit(3);
  svo_hip_stream_sync(NULL);
  return h;
}

static unsigned rng_state = 12345u;
static double uni() {  // [0, 1)
  rng_state = rng_state * 1664525u + 1013904223u;
  return (rng_state >> 8) * (1.0 / 16777216.0);
}

int main() {
  if (svo_hip_device_count() <= 0) { std::fprintf(stderr, "no HIP device: there is no CPU fallback\n"); return 2; }
  CK(svo_hip_set_device(0));
  const int W = 752, H = 480, CELL = 30, N_KF = 6, N_PTS = 600, N_CAND = 400;
  const int n_cols = (W + CELL - 1) / CELL, n_rows = (H + CELL - 1) / CELL, n_cells = n_cols * n_rows;
  svo_hip_camera cam;
  CK(svo_hip_camera_pinhole(W, H, 315.5, 315.5, 376.0, 240.0, 0, 0, 0, 0, 0, &cam));

  // frames: keyframes hovering 2 m above the plane z = 0 and looking down, the current frame last
  const int n_frames = N_KF + 1, cur = N_KF;
  std::vector<double> T((size_t)n_frames * 12, 0.0);
  std::vector<double> centre((size_t)n_frames * 3);
  for (int f = 0; f < n_frames; ++f) {
    const double c[3] = {f == cur ? 0.0 : 2.0 * uni() - 1.0, f == cur ? 0.0 : 2.0 * uni() - 1.0, 2.0};
    double* R = &T[(size_t)f * 12];
    R[0] = 1; R[4] = -1; R[8] = -1;  // R = diag(1, -1, -1)
    R[9] = -(R[0] * c[0]); R[10] = -(R[4] * c[1]); R[11] = -(R[8] * c[2]);
    std::memcpy(&centre[(size_t)f * 3], c, sizeof(c));
  }
  std::vector<int32_t> kf_rank(n_frames, -1);
  {  // the four closest keyframes overlap, closest first
    std::vector<std::pair<double, int> > d;
    for (int f = 0; f < N_KF; ++f) d.push_back(std::make_pair(std::hypot(centre[3 * f], centre[3 * f + 1]), f));
    std::sort(d.begin(), d.end());
    for (int r = 0; r < 4; ++r) kf_rank[d[r].second] = r;
  }

  // the map: points of the keyframes (good / unknown), each observed from 1-3 keyframes, then the candidates
  const int P = N_PTS + N_CAND;
  std::vector<double> pos(3 * P);
  std::vector<int32_t> type(P), order(P, 0), obs_begin(P), obs_count(P), index(P);
  std::vector<int32_t> o_frame, o_order, o_level, o_index;
  std::vector<uint8_t> o_type;
  std::vector<double> o_px, o_f, o_grad;
  std::vector<int> next_ord(N_KF, 0);
  for (int p = 0; p < P; ++p) {
    index[p] = p;
    pos[3 * p] = 5.0 * uni() - 2.5; pos[3 * p + 1] = 4.0 * uni() - 2.0; pos[3 * p + 2] = 0.05 * uni();
    type[p] = p < N_PTS ? (uni() < 0.5 ? 3 : 2) : 1;
    order[p] = p < N_PTS ? 0 : p - N_PTS;
    obs_begin[p] = (int32_t)o_frame.size();
    const int n_obs = p < N_PTS ? 1 + (int)(3 * uni()) : 1;
    int first = (int)(N_KF * uni());
    for (int k = 0; k < n_obs; ++k) {
      const int f = (first + k) % N_KF;
      o_index.push_back((int32_t)o_frame.size());
      o_frame.push_back(f);
      o_order.push_back(p < N_PTS ? next_ord[f]++ : -1);  // a candidate's feature is in no keyframe's list
      o_level.push_back(0);
      o_type.push_back(0);
      o_px.push_back(100.0); o_px.push_back(100.0);
      o_f.push_back(0.0); o_f.push_back(0.0); o_f.push_back(1.0);
      o_grad.push_back(1.0); o_grad.push_back(0.0);
    }
    obs_count[p] = (int32_t)o_frame.size() - obs_begin[p];
  }
  const int O = (int)o_frame.size();

  // device side: the resident map (empty) and the patch that fills it
  svo_hip_map map;
  std::memset(&map, 0, sizeof(map));
  map.n_points = P; map.n_obs = O;
  map.d_pos = dev(std::vector<double>(), 3 * P); map.d_type = dev(std::vector<int32_t>(), P);
  map.d_order = dev(std::vector<int32_t>(), P); map.d_obs_begin = dev(std::vector<int32_t>(), P);
  map.d_obs_count = dev(std::vector<int32_t>(), P); map.d_obs_frame = dev(std::vector<int32_t>(), O);
  map.d_obs_order = dev(std::vector<int32_t>(), O); map.d_obs_level = dev(std::vector<int32_t>(), O);
  map.d_obs_type = dev(std::vector<uint8_t>(), O); map.d_obs_px = dev(std::vector<double>(), 2 * O);
  map.d_obs_f = dev(std::vector<double>(), 3 * O); map.d_obs_grad = dev(std::vector<double>(), 2 * O);
  svo_hip_map_patch patch;
  std::memset(&patch, 0, sizeof(patch));
  patch.n_points = P; patch.n_obs = O;
  patch.d_index = dev(index); patch.d_pos = dev(pos); patch.d_type = dev(type); patch.d_order = dev(order);
  patch.d_obs_begin = dev(obs_begin); patch.d_obs_count = dev(obs_count);
  patch.d_obs_index = dev(o_index); patch.d_obs_order = dev(o_order);
  patch.obs.d_frame = dev(o_frame); patch.obs.d_level = dev(o_level); patch.obs.d_type = dev(o_type);
  patch.obs.d_px = dev(o_px); patch.obs.d_f = dev(o_f); patch.obs.d_grad = dev(o_grad);

  std::vector<int32_t> cell_order(n_cells), cell_rank(n_cells);
  for (int k = 0; k < n_cells; ++k) cell_order[k] = k;
  for (int k = n_cells - 1; k > 0; --k) std::swap(cell_order[k], cell_order[(int)(uni() * (k + 1))]);  // random_shuffle
  for (int i = 0; i < n_cells; ++i) cell_rank[cell_order[i]] = i;
  svo_hip_grid grid;
  grid.cell_size = CELL; grid.n_cols = n_cols; grid.n_rows = n_rows; grid.n_cells = n_cells;
  grid.d_cell_rank = dev(cell_rank);
  svo_hip_frames frames;
  frames.n_frames = n_frames; frames.reserved = 0; frames.d_slot = NULL; frames.d_T_f_w = dev(T);
  const int CAP = 2048;
  svo_hip_reprojection out;
  out.d_header = dev(std::vector<int32_t>(), SVO_HIP_REPROJ_HEADER);
  out.d_point_cell = dev(std::vector<int32_t>(), P); out.d_point_px = dev(std::vector<double>(), 2 * P);
  out.d_kf_count = dev(std::vector<int32_t>(), n_frames);
  out.d_visit_point = dev(std::vector<int32_t>(), CAP); out.d_visit_cell = dev(std::vector<int32_t>(), CAP);
  out.d_visit_trial = dev(std::vector<int32_t>(), CAP);
  out.d_trial_cur = dev(std::vector<int32_t>(), CAP); out.d_trial_pos = dev(std::vector<double>(), 3 * CAP);
  out.d_trial_obs_begin = dev(std::vector<int32_t>(), CAP); out.d_trial_obs_end = dev(std::vector<int32_t>(), CAP);
  out.d_trial_cell = dev(std::vector<int32_t>(), CAP); out.d_trial_px = dev(std::vector<double>(), 2 * CAP);
  int32_t* d_rank = dev(kf_rank);

  CK(svo_hip_reproject_map(&cam, &frames, cur, d_rank, &map, &patch, &grid, 0, 1 << 30, CAP, CAP, &out, NULL));
  CK(svo_hip_stream_sync(NULL));
  const std::vector<int32_t> header = host(out.d_header, SVO_HIP_REPROJ_HEADER);
  const int V = header[2], M = header[3];
  std::printf("status %d: %d points inside the frame, %d visits, %d trials, end cell %d of %d\n", header[0], header[1], V, M, header[4], n_cells);
  if (header[0] != 0) return 1;
  const std::vector<int32_t> vp = host(out.d_visit_point, V), vc = host(out.d_visit_cell, V), vt = host(out.d_visit_trial, V);
  const std::vector<int32_t> kfc = host(out.d_kf_count, n_frames);

  // ---- the host walk, the reference's way ---------------------------------------------------------------------------
  std::vector<std::vector<int> > cells(n_cells);
  std::vector<char> done(P, 0);
  std::vector<int> kf_in(n_frames, 0);
  auto project = [&](int p) -> int {  // Reprojector::reprojectPoint
    const double* R = &T[(size_t)cur * 12];
    const double x = R[0] * pos[3 * p] + R[9], y = R[4] * pos[3 * p + 1] + R[10], z = R[8] * pos[3 * p + 2] + R[11];
    const double u = cam.fx * x / z + cam.cx, v = cam.fy * y / z + cam.cy;
    if ((int)u >= 8 && (int)u < W - 8 && (int)v >= 8 && (int)v < H - 8) return (int)(v / CELL) * n_cols + (int)(u / CELL);
    return -1;
  };
  for (int r = 0; r < 4; ++r) {
    int f = -1;
    for (int i = 0; i < n_frames; ++i) if (kf_rank[i] == r) f = i;
    std::vector<std::pair<int, int> > fts;  // (position in fts_, point)
    for (int p = 0; p < N_PTS; ++p)
      for (int o = obs_begin[p]; o < obs_begin[p] + obs_count[p]; ++o)
        if (o_frame[o] == f) fts.push_back(std::make_pair(o_order[o], p));
    std::sort(fts.begin(), fts.end());
    for (size_t i = 0; i < fts.size(); ++i) {
      const int p = fts[i].second;
      if (done[p]) continue;
      done[p] = 1;
      const int k = project(p);
      if (k >= 0) { cells[k].push_back(p); ++kf_in[f]; }
    }
  }
  for (int p = N_PTS; p < P; ++p) {  // candidates, in list order
    const int k = project(p);
    if (k >= 0) cells[k].push_back(p);
  }
  std::vector<int> want_p, want_c;
  for (int i = 0; i < n_cells; ++i) {
    std::vector<int>& c = cells[cell_order[i]];
    std::stable_sort(c.begin(), c.end(), [&](int a, int b) { return type[a] > type[b]; });
    for (size_t k = 0; k < c.size(); ++k) { want_p.push_back(c[k]); want_c.push_back(i); }
  }
  bool ok = (int)want_p.size() == V;
  for (int v = 0; ok && v < V; ++v) ok = vp[v] == want_p[v] && vc[v] == want_c[v];
  for (int f = 0; ok && f < n_frames; ++f) ok = kfc[f] == kf_in[f];
  int with_view = 0;
  for (int v = 0; v < V; ++v) with_view += vt[v] >= 0;
  ok = ok && with_view == M;
  std::printf("visit list %s the host walk (%d candidates in %d cells, %d with a close view)\n", ok ? "equals" : "DIFFERS from", V,
              V ? want_c.back() + 1 : 0, M);
  std::printf(ok ? "OK\n" : "FAILED\n");
  return ok ? 0 : 1;
}
