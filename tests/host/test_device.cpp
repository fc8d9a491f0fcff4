// tests/host/test_device.cpp -- unit test of the product's C++ host layer (rpg_svo_amd/host/
// svo_hip_device.{h,cpp}) over the C ABI; built with plain g++ by tests/test_host_device_gpu.py and
// run on the GPU box; the same source linked against tests/host/mock_svo_hip.cpp instead of libsvo_hip.so checks the
// host LOGIC where there is no GPU.  Prints "ok <name>" per check, exits non-zero on the first failure.
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
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

  // ---- where blocks live: hybrid (the default) keeps inputs in the mirror and makes outputs the pinned host block
  //      itself; mirrored copies everything both ways; mapped has no mirror (SVO_HIP_ARENA selects) ---------------------
  {
    const char* mode_env = std::getenv("SVO_HIP_ARENA");
    const std::string mode = mode_env ? mode_env : "hybrid";
    CHECK(a.mode() == (mode == "mapped" ? Arena::MAPPED : mode == "mirrored" ? Arena::MIRRORED : Arena::HYBRID));
    a.reset();
    int32_t *d_i, *d_o, *d_o2;
    int32_t* hi = a.alloc<int32_t>(64, &d_i);
    a.endInputs();
    const size_t inputs_end = a.used();
    int32_t* ho = a.alloc<int32_t>(64, &d_o);
    int32_t* ho2 = a.alloc<int32_t>(64, &d_o2);
    CHECK(((void*)hi == (void*)d_i) == (a.mode() == Arena::MAPPED));
    CHECK(((void*)ho == (void*)d_o) == (a.mode() != Arena::MIRRORED) && ((void*)ho2 == (void*)d_o2) == (a.mode() != Arena::MIRRORED));
    for (int i = 0; i < 64; ++i) { hi[i] = 3 * i; ho[i] = -1; ho2[i] = 7; }
    a.uploadAll(lane.stream);  // the inputs only: the in/out blocks are host memory already
    check(svo_hip_memcpy_d2d(d_o, d_i, 64 * sizeof(int32_t), lane.stream), "d2d");  // "a kernel writes a result"
    a.download(lane.stream);                                                          // nothing to copy
    a.downloadRange(inputs_end, a.used(), lane.stream);
    a.fetch(ho, 64, lane.stream);
    check(svo_hip_stream_sync(lane.stream), "sync");
    for (int i = 0; i < 64; ++i) CHECK(ho[i] == 3 * i && ho2[i] == 7);
    check(svo_hip_memset(d_i, 0, 8, lane.stream), "memset");  // device-side change of an INPUT block: fetch() copies it
    a.fetch(hi, 64, lane.stream);
    check(svo_hip_stream_sync(lane.stream), "sync");
    CHECK(hi[0] == 0 && hi[1] == 0 && hi[2] == 6);
    threw = false;
    try { a.downloadRange(8, a.used() + 1, lane.stream); } catch (const Error&) { threw = true; }
    CHECK(threw);
    a.reset();
  }
  std::puts("ok arena modes");

  // ---- work left running for the lane's next call: prediction and deferred second halves -----------------------
  {
    const Device::Stats st0 = dev.statsSnapshot();
    lane.spec.valid = true;          // a prediction nobody takes: the next call of the lane drains and drops it
    lane.spec.in_flight = true;
    lane.spec.stream = lane.stream;
    dev.beginCall(lane);
    CHECK(!lane.spec.valid && !lane.spec.in_flight);
    dev.countSpeculation(true);
    const Device::Stats st1 = dev.statsSnapshot();
    CHECK(st1.spec_misses == st0.spec_misses + 1 && st1.spec_hits == st0.spec_hits + 1 && st1.calls == st0.calls + 1);
    int ran = 0;
    lane.deferred = [&ran]() { ++ran; };
    dev.beginCall(lane);             // the second half of a deferred call runs before the arena is handed out again
    CHECK(ran == 1 && !lane.deferred);
    dev.beginCall(lane);
    CHECK(ran == 1);
    lane.deferred = [&ran]() { ran += 10; };
    dev.joinDeferred(Device::LANE_MAPPING);  // another lane: nothing of its own to join (and no lane is created for it)
    CHECK(ran == 1);
    dev.joinDeferred(Device::LANE_TRACKING);
    CHECK(ran == 11 && !lane.deferred);
    lane.deferred = [&ran]() { ran += 100; };
    Device::joinDeferredAll();
    CHECK(ran == 111);
    Device::joinDeferredAll();
    CHECK(ran == 111);
    dev.addStage(Device::STAGE_DEPTH_FILTER, 1.0, 2.0, 3.0, 4.0);
    dev.addStage(Device::STAGE_DEPTH_FILTER, 0.0, 5.0, 6.0, 0.0, false);  // the join of a deferred call: same sample
    const Device::Stats st2 = dev.statsSnapshot();
    CHECK(st2.n[Device::STAGE_DEPTH_FILTER] == st1.n[Device::STAGE_DEPTH_FILTER] + 1);
    CHECK(st2.device_us[Device::STAGE_DEPTH_FILTER] == st1.device_us[Device::STAGE_DEPTH_FILTER] + 7.0);
  }
  std::puts("ok prediction / deferred bookkeeping");

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
  // the mapping lane pins independently (and takes the frame over through the slot's event: uploads are published
  // when enqueued, not when complete)
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
