// tests/host/test_device.cpp -- unit test of the product's C++ host layer (rpg_svo_amd/host/
// svo_hip_device.{h,cpp}) over the C ABI; built with plain g++ by tests/test_host_device_gpu.py and
// run on the GPU box.  Prints "ok <name>" per check, exits non-zero on the first failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

#include "svo_hip_device.h"

using namespace svo_hip;

#define CHECK(cond)                                                              \
  do {                                                                           \
    if (!(cond)) { std::fprintf(stderr, "FAILED %s:%d: %s\n", __FILE__, __LINE__, #cond); std::exit(1); } \
  } while (0)

static std::vector<uint8_t> image(int w, int h, int seed) {
  std::vector<uint8_t> v((size_t)w * h);
  unsigned s = 1234567u + 77u * (unsigned)seed;
  for (size_t i = 0; i < v.size(); ++i) { s = s * 1664525u + 1013904223u; v[i] = (uint8_t)(s >> 24); }
  return v;
}

int main() {
  const int W = 160, H = 120, LEVELS = 3;
  Device& dev = Device::forGeometry(W, H, LEVELS);
  CHECK(dev.configured() && dev.layout().w[0] == W && dev.layout().h[2] == H / 4);
  CHECK(&Device::forGeometry(W, H, LEVELS) == &dev);          // same geometry -> same context
  CHECK(&Device::forGeometry(W, H, 2) == &dev);               // fewer levels fit the same store
  Device& other = Device::forGeometry(W / 2, H / 2, LEVELS);  // another camera -> its own context
  CHECK(&other != &dev && other.layout().w[0] == W / 2);
  std::puts("ok contexts per geometry");

  // ---- arena: blocks, one upload, in-place fetch, download -------------------------------------
  Lane& lane = dev.lane(Device::LANE_TRACKING);
  Arena& a = lane.arena;
  a.reset();
  double* d_in; double* d_out; int32_t* d_idx;
  double* in = a.alloc<double>(1000, &d_in);
  int32_t* idx = a.alloc<int32_t>(7, &d_idx);
  CHECK(((uintptr_t)in & 255) == 0 && ((uintptr_t)idx & 255) == 0 && ((uintptr_t)d_in & 255) == 0);
  for (int i = 0; i < 1000; ++i) in[i] = 0.5 * i;
  for (int i = 0; i < 7; ++i) idx[i] = i * i;
  a.endInputs();
  double* out = a.alloc<double>(1000, &d_out);
  std::memset(out, 0, 1000 * sizeof(double));
  a.upload(lane.stream);
  check(svo_hip_memcpy_d2d(d_out, d_in, 1000 * sizeof(double), lane.stream), "d2d");
  check(svo_hip_memset(d_idx + 1, 0, 4, lane.stream), "memset");  // device-side change of an input block
  a.download(lane.stream);
  a.fetch(idx, 7, lane.stream);
  check(svo_hip_stream_sync(lane.stream), "sync");
  for (int i = 0; i < 1000; ++i) CHECK(out[i] == 0.5 * i);
  CHECK(idx[0] == 0 && idx[1] == 0 && idx[2] == 4 && idx[6] == 36);
  bool threw = false;
  try { int x; a.fetch(&x, 1, lane.stream); } catch (const Error&) { threw = true; }
  CHECK(threw);
  threw = false;
  try { a.reset(); double* d; a.alloc<double>((size_t)1 << 30, &d); } catch (const Error&) { threw = true; }
  CHECK(threw);  // overflow is an error, never a silent reallocation under live pointers
  a.reset();
  a.reserve((size_t)24 << 20);
  { double* d; double* h = a.alloc<double>((size_t)2 << 20, &d); CHECK(h != NULL && d != NULL); }
  a.reset();
  std::puts("ok arena");

  // ---- pyramid cache: hit, LRU eviction, pinning, re-upload --------------------------------------
  dev.configure(W, H, LEVELS, /*n_slots=*/3);
  const uint64_t up0 = dev.stats.uploads;
  std::vector<std::vector<uint8_t> > img;
  for (int i = 0; i < 5; ++i) img.push_back(image(W, H, i));
  dev.beginCall(Device::LANE_TRACKING);
  const int s0 = dev.slotOf(100, img[0].data(), W, Device::LANE_TRACKING);
  const int s1 = dev.slotOf(101, img[1].data(), W, Device::LANE_TRACKING);
  CHECK(s0 != s1 && dev.slotOf(100, img[0].data(), W, Device::LANE_TRACKING) == s0);  // hit
  CHECK(dev.stats.uploads - up0 == 2);
  const int s2 = dev.slotOf(102, img[2].data(), W, Device::LANE_TRACKING);
  CHECK(s2 != s0 && s2 != s1);
  threw = false;  // three frames pinned by the running call, pool of three: a fourth cannot come in
  try { dev.slotOf(103, img[3].data(), W, Device::LANE_TRACKING); } catch (const Error&) { threw = true; }
  CHECK(threw);
  dev.beginCall(Device::LANE_TRACKING);  // next call: nothing pinned, LRU (frame 100... touched last? no: 101) goes
  dev.slotOf(100, img[0].data(), W, Device::LANE_TRACKING);            // touch 100 -> 101 is the oldest
  const int s3 = dev.slotOf(103, img[3].data(), W, Device::LANE_TRACKING);
  CHECK(s3 == s1 && dev.stats.evictions >= 1);
  // the mapping lane pins independently
  dev.beginCall(Device::LANE_MAPPING);
  CHECK(dev.slotOf(102, img[2].data(), W, Device::LANE_MAPPING) == s2);
  dev.beginCall(Device::LANE_TRACKING);
  dev.slotOf(100, img[0].data(), W, Device::LANE_TRACKING);
  dev.slotOf(103, img[3].data(), W, Device::LANE_TRACKING);
  threw = false;  // 100, 103 pinned by tracking, 102 by mapping
  try { dev.slotOf(104, img[4].data(), W, Device::LANE_TRACKING); } catch (const Error&) { threw = true; }
  CHECK(threw);
  dev.beginCall(Device::LANE_MAPPING);  // mapping call over: 102 can go
  dev.beginCall(Device::LANE_TRACKING);
  const int s4 = dev.slotOf(104, img[4].data(), W, Device::LANE_TRACKING);
  CHECK(s4 == s2);
  // an evicted frame that is still alive on the host simply comes back, with the right pixels
  const int s1b = dev.slotOf(101, img[1].data(), W, Device::LANE_TRACKING);
  std::vector<uint8_t> back((size_t)W * H), lvl1((size_t)(W / 2) * (H / 2));
  check(svo_hip_pyramid_download_level(&dev.layout(), dev.store(), s1b, 0, back.data(), dev.lane(0).stream), "download");
  CHECK(back == img[1]);
  check(svo_hip_pyramid_download_level(&dev.layout(), dev.store(), s1b, 1, lvl1.data(), dev.lane(0).stream), "download");
  // level 1 was built on the device (K0): 160 % 16 == 0 -> SSE2 flavour (avg of avgs, rounding up)
  for (int y = 0; y < H / 2; ++y)
    for (int x = 0; x < W / 2; ++x) {
      const uint8_t* p = &img[1][(size_t)(2 * y) * W + 2 * x];
      const unsigned aa = (p[0] + p[W] + 1u) >> 1, bb = (p[1] + p[W + 1] + 1u) >> 1;
      CHECK(lvl1[(size_t)y * (W / 2) + x] == (uint8_t)((aa + bb + 1u) >> 1));
    }
  dev.forget(101);
  CHECK(dev.slotOf(105, img[0].data(), W, Device::LANE_TRACKING) == s1b);  // freed slot reused first
  std::puts("ok pyramid cache");

  // ---- workspace growth ------------------------------------------------------------------------------
  void* w1 = dev.workspace(lane, 10);
  CHECK(w1 != NULL && lane.workspace_bytes >= svo_hip_match_workspace_bytes(10));
  dev.workspace(lane, 200000);
  CHECK(lane.workspace_bytes >= svo_hip_match_workspace_bytes(200000));
  std::puts("ok workspace");
  // ---- several host threads (a rig of cameras on one GPU): each gets its own lane / stream ------------
  dev.configure(W, H, LEVELS, /*n_slots=*/16);
  {
    const int NT = 6, ROUNDS = 40;
    std::vector<int> bad(NT, 0);
    std::vector<Lane*> lanes(NT, (Lane*)NULL);
    std::vector<std::thread> th;
    for (int t = 0; t < NT; ++t)
      th.push_back(std::thread([&, t]() {
        try {
          Lane& l = dev.lane(Device::LANE_TRACKING);
          lanes[t] = &l;
          std::vector<uint8_t> mine = image(W, H, 100 + t);
          for (int r = 0; r < ROUNDS; ++r) {
            dev.beginCall(l);
            const int a0 = dev.slotOf(1000 + t, mine.data(), W, l);          // this thread's own frame
            const int a1 = dev.slotOf(2000 + (r + t) % 5, img[(r + t) % 5].data(), W, l);  // shared keyframes
            if (a0 == a1) bad[t] = 1;
            Arena& ar = l.arena;
            ar.reset();
            double* d_x; double* d_y;
            double* x = ar.alloc<double>(512, &d_x);
            for (int i = 0; i < 512; ++i) x[i] = t * 1000.0 + r + i;
            ar.endInputs();
            double* y = ar.alloc<double>(512, &d_y);
            ar.upload(l.stream);
            check(svo_hip_memcpy_d2d(d_y, d_x, 512 * sizeof(double), l.stream), "d2d");
            ar.download(l.stream);
            check(svo_hip_stream_sync(l.stream), "sync");
            for (int i = 0; i < 512; ++i) if (y[i] != t * 1000.0 + r + i) bad[t] = 2;
            std::vector<uint8_t> back0((size_t)W * H);
            check(svo_hip_pyramid_download_level(&dev.layout(), dev.store(), a0, 0, back0.data(), l.stream), "download");
            if (back0 != mine) bad[t] = 3;
          }
        } catch (const Error& e) {
          std::fprintf(stderr, "thread %d: %s\n", t, e.what());
          bad[t] = 4;
        }
      }));
    for (size_t t = 0; t < th.size(); ++t) th[t].join();
    for (int t = 0; t < NT; ++t) {
      CHECK(bad[t] == 0);
      for (int u = 0; u < t; ++u) CHECK(lanes[t] != lanes[u] && lanes[t]->stream != lanes[u]->stream);
    }
  }
  std::puts("ok concurrent lanes");
  std::puts("ALL OK");
  return 0;
}
