// Drop-in for THREE members of svo::DepthFilter (svo/include/svo/depth_filter.h): updateSeeds,
// updateSeed and computeTau.  Everything else of svo/src/depth_filter.cpp -- svo::Seed, the seed list,
// the keyframe queue, the mapping thread -- stays in the reference's own file: build that file minus
// these three definitions (scripts/strip_members.py) next to this one.
// DepthFilter::updateSeeds (depth_filter.cpp
// :197-291) -- visibility test, Matcher::findEpipolarMatchDirect (epipolar ZMSSD scan +
// sub-pixel alignment, matcher.cpp:179-321), triangulation, computeTau and the Bayesian
// updateSeed -- runs for ALL seeds in one batched call of svo_hip_update_seeds (K5) on the
// mapping lane's stream.  The list surgery the reference interleaves with the arithmetic
// (erase old / NaN seeds, create the Point of a converged seed and hand it to the callback,
// mark the detector grid) is replayed on the host from the per-seed status, in list order.
// With deferred mapping switched on (svo_hip::Device::deferredMapping(); only without the mapping thread) the call
// returns after enqueueing and the replay runs at the next join point -- see svo_hip_device.h.
#include <svo/depth_filter.h>

#include <algorithm>
#include <cmath>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/feature_detection.h>
#include <svo/frame.h>
#include <svo/matcher.h>
#include <svo/point.h>

#include "marshal.h"
#include "seed_store.h"

namespace svo {

namespace hip_dropin {
// Row N2, seeds: one resident store per DepthFilter (seed_store.h), alive as long as the filter (released by the
// drop-in's destructor below).  SVO_HIP_SEED_STORE=off flattens and ships the whole list per call, as rounds 1-4 did.
struct SeedStoreRegistry {
  std::mutex mut;
  std::map<const DepthFilter*, SeedStore*> all;
  uint64_t calls, records_sent, rebuilds;
  SeedStoreRegistry() : calls(0), records_sent(0), rebuilds(0) {}
};
static SeedStoreRegistry& seedStores() {
  static SeedStoreRegistry r;
  return r;
}
static SeedStore& seedStoreOf(const DepthFilter* df) {
  SeedStoreRegistry& r = seedStores();
  std::lock_guard<std::mutex> g(r.mut);
  SeedStore*& s = r.all[df];
  if (s == NULL) s = new SeedStore();
  return *s;
}
static void releaseSeedStore(const DepthFilter* df) {
  SeedStoreRegistry& r = seedStores();
  std::lock_guard<std::mutex> g(r.mut);
  std::map<const DepthFilter*, SeedStore*>::iterator it = r.all.find(df);
  if (it == r.all.end()) return;
  r.calls += it->second->stats.calls; r.records_sent += it->second->stats.records_sent; r.rebuilds += it->second->stats.rebuilds;
  delete it->second;
  r.all.erase(it);
}
static bool seedStoreOn() {
  static const bool on = [] { const char* v = std::getenv("SVO_HIP_SEED_STORE"); return !(v && std::string(v) == "off"); }();  // (on | verify)
  return on;
}
// ---- an update enqueued early (round 6) -------------------------------------------------------------------------------
// FrameHandlerMono::processFrame calls depth_filter_->addFrame -- hence, without the mapping thread, updateSeeds -- at its very
// end (frame_handler_mono.cpp:198), a few dozen microseconds of host work (optimizeStructure, tracking quality, scene depth,
// the keyframe decision) after pose_optimizer::optimizeGaussNewton has fixed the pose the update depends on.  The
// optimizer's drop-in therefore calls the mapping lane's `early_hook` as its last act: the update's kernels are enqueued
// then and run while the host finishes the frame; updateSeeds, when the reference calls it, finds them (nearly) done.  It
// TAKES the early update only for the same frame with the same pose and the same seed list, and only for a frame that has
// not become a keyframe; anything else -- a frame that fails, a keyframe (addKeyframe does not update), another call on
// the mapping lane -- DROPS it: the stream is drained, nothing is replayed into the list and the resident seed store is
// told to forget its shadow (the kernels have advanced the seeds' state in HBM: the next call re-sends the list).
// Two phases: the optimizer's drop-in calls the hook BEFORE it waits for the refinement the reprojector predicted (the host
// would only wait: the seed list's merge walk, the call's tables and their upload happen then -- phase 1, nothing of it
// depends on the pose) and AFTER it (phase 2: svo_hip_update_seeds_resident_pose, the pose by value); a phase 2 that finds
// nothing held does both at once.
struct EarlyUpdate {
  bool valid;      // the update's kernels are running (or done) on `stream`
  bool prepared;   // phase 1 only: tables marshalled and uploaded, `launch` holds the rest (the frame's pose is not final yet)
  std::function<void(const double*)> launch;  // (phase 2: the kernels, with the pose's frame-table row by value)
  int frame_id;
  double T[12];
  std::vector<int> ids;
  std::function<void()> replay;  // (needs seeds_mut_)
  SeedStore* store;
  void* stream;
  svo_hip::Lane* lane;           // where the hook is registered
  EarlyUpdate() : valid(false), prepared(false), frame_id(-1), store(NULL), stream(NULL), lane(NULL) {}
};
struct EarlyRegistry {
  std::mutex mut;
  std::map<const DepthFilter*, EarlyUpdate> all;
};
static EarlyRegistry& earlyUpdates() {
  static EarlyRegistry r;
  return r;
}
static EarlyUpdate& earlyOf(const DepthFilter* df) {
  EarlyRegistry& r = earlyUpdates();
  std::lock_guard<std::mutex> g(r.mut);
  return r.all[df];  // (node addresses are stable)
}
static thread_local int tl_early_phase = 0;  // updateSeeds is being called through the hook: phase 1 (marshal, upload, hold) or 2 (launch)

// calls / records sent / rebuilds, summed over the stores of the process (read-outs of the tests and the benchmark)
void seedStoreStats(uint64_t out[3]) {
  SeedStoreRegistry& r = seedStores();
  std::lock_guard<std::mutex> g(r.mut);
  out[0] = r.calls; out[1] = r.records_sent; out[2] = r.rebuilds;
  for (std::map<const DepthFilter*, SeedStore*>::const_iterator it = r.all.begin(); it != r.all.end(); ++it) {
    out[0] += it->second->stats.calls; out[1] += it->second->stats.records_sent; out[2] += it->second->stats.rebuilds;
  }
}
}  // namespace hip_dropin

// ---- the update, on the device ---------------------------------------------------------------
void DepthFilter::updateSeeds(FramePtr frame) {
  using namespace hip_dropin;
  const int phase = tl_early_phase;
  const bool early = phase != 0;  // enqueue only: called by the pose optimizer's drop-in through the lane's hook
  if (early && (thread_ != NULL || svo_hip::Device::deferredMapping() || frame->isKeyframe())) return;
  if (phase == 1) {  // (the flattened list carries the pose in its tables; SVO_HIP_EARLY_MAPPER=1phase: phase 2 does it all)
    static const bool one_phase = [] { const char* v = std::getenv("SVO_HIP_EARLY_MAPPER"); return v && std::string(v) == "1phase"; }();
    if (one_phase || !seedStoreOn()) return;
  }
  svo_hip::Device::joinDeferredAll();  // the previous frame's update writes into seeds_ first (takes seeds_mut_ itself)
  svo_hip::Device& dev = ensureDevice(*frame);
  const int L = svo_hip::Device::LANE_MAPPING;
  svo_hip::Lane& lane = dev.lane(L);
  EarlyUpdate& eu = earlyOf(this);
  {  // (nothing to do: do not touch the device)
    lock_t peek(seeds_mut_);
    if ((seeds_updating_halt_ || seeds_.empty()) && !eu.valid && !eu.prepared) return;
  }
  // Lock order: the lane, then the seed list -- the order the deferred closure below takes them in when a later call joins it
  // (it runs under the lane's mutex and locks seeds_mut_ itself).  The other way round here would be a lock-order inversion
  // (ThreadSanitizer names it), harmless only as long as every path joins before it locks.
  std::lock_guard<std::mutex> guard(lane.mut);
  lock_t lock(seeds_mut_);
  if (thread_ == NULL && !svo_hip::Device::deferredMapping() && svo_hip::Device::earlyMappingEnabled()) {
    if (!lane.early_hook || eu.lane != &lane) {  // (once per filter and lane)
      eu.lane = &lane;
      lane.early_hook = [this](const void* fp, const int ph) {
        struct Flag { explicit Flag(int p) { tl_early_phase = p; } ~Flag() { tl_early_phase = 0; } } flag(ph);
        updateSeeds(*static_cast<const FramePtr*>(fp));
      };
    }
  }
  if (phase == 2 && eu.prepared) {
    // phase 1 holds this frame's update: launch it with the pose the optimizer has just fixed -- or drop what is held
    if (eu.frame_id == frame->id_ && eu.ids.size() == seeds_.size() && !seeds_updating_halt_) {
      double T[12];
      poseToRt(frame->T_f_w_, T);
      std::function<void(const double*)> launch;
      launch.swap(eu.launch);
      eu.prepared = false;
      try {
        launch(T);
      } catch (...) {
        if (lane.early_drop) {
          std::function<void()> drop;
          drop.swap(lane.early_drop);
          try { drop(); } catch (...) {}
        }
        throw;
      }
      std::memcpy(eu.T, T, sizeof(T));
      eu.valid = true;
      dev.countEarlyTwoPhase();
      return;
    }
    if (lane.early_drop) {
      std::function<void()> drop;
      drop.swap(lane.early_drop);
      drop();
    }
    // (and the update in one go, below)
  }
  if (early && (eu.valid || eu.prepared)) return;  // (the hook called twice for a frame: nothing to add)
  if (!early && (eu.valid || eu.prepared)) {
    // an update of this filter is running on the stream (or held before its launch): this call's, or one to be dropped
    const size_t S_now = seeds_.size();
    bool same = eu.valid && eu.frame_id == frame->id_ && eu.ids.size() == S_now && !frame->isKeyframe() && !seeds_updating_halt_;
    if (same) {
      double T[12];
      poseToRt(frame->T_f_w_, T);
      same = std::memcmp(T, eu.T, sizeof(T)) == 0;
    }
    if (same) {
      size_t s = 0;
      for (std::list<Seed>::const_iterator it = seeds_.begin(); it != seeds_.end(); ++it, ++s)
        if (eu.ids[s] != it->id) { same = false; break; }
    }
    if (same) {
      eu.valid = false;
      lane.early_drop = nullptr;
      const double t0 = svo_hip::StageTimer::now();
      try {
        svo_hip::check(svo_hip_stream_sync(eu.stream), "svo_hip_stream_sync");
      } catch (...) {
        if (eu.store) eu.store->invalidate();
        throw;
      }
      const double t1 = svo_hip::StageTimer::now();
      std::function<void()> replay;
      replay.swap(eu.replay);
      replay();  // seeds_mut_ is held
      dev.addStage(svo_hip::Device::STAGE_DEPTH_FILTER, 0.0, t1 - t0, svo_hip::StageTimer::now() - t1, 0.0, false);
      dev.countEarlyMapping(true);
      return;
    }
    if (lane.early_drop) {
      std::function<void()> drop;
      drop.swap(lane.early_drop);
      drop();
    }
  }
  if (seeds_updating_halt_) return;  // the halt flag is honoured between launches
  const size_t S = seeds_.size();
  if (S == 0) return;
  dev.beginCall(L);
  svo_hip::StageTimer stage_timer(dev, lane, svo_hip::Device::STAGE_DEPTH_FILTER);
  svo_hip::Arena& a = lane.arena;
  a.reset();
  a.reserve(((size_t)1 << 16) + S * 256 + 4096 * 32);
  const bool resident = seedStoreOn();
  std::vector<int> ids(S);  // Seed::id, ascending along the list: how a later replay finds its seeds again
  std::vector<float> st;
  float *sa = NULL, *sb = NULL, *smu = NULL, *ss2 = NULL;
  int32_t *d_cur = NULL, *d_status; double *d_xyz, *d_px; float* d_state = NULL;
  svo_hip_frames ft;
  svo_hip_seeds seeds;
  FeatureColumns ftr;
  SeedStore::Call rc;
  // sync() commits the store's shadow before anything has reached the device: until the update has completed (or has been
  // handed to the deferred closure, which guards its own wait) every way out of this function -- slotOf / workspace /
  // launch / stream failures all throw -- must leave the shadow claiming nothing (SeedStore::invalidate).
  struct StoreGuard {
    SeedStore* store;
    StoreGuard() : store(NULL) {}
    ~StoreGuard() { if (store) store->invalidate(); }
    void done() { store = NULL; }
  } store_guard;
  if (resident) {
    store_guard.store = &seedStoreOf(this);
    // row N2: state and Feature of every seed stay in HBM; this call sends the slots in list order, the records of the
    // seeds it meets for the first time and the frame table (seed_store.h)
    rc = seedStoreOf(this).sync(seeds_, frame.get(), dev, L, a);
    size_t s = 0;
    for (std::list<Seed>::iterator it = seeds_.begin(); it != seeds_.end(); ++it, ++s) ids[s] = it->id;
    ft = rc.frames;
    a.endInputs();
    float* h_state = a.alloc<float>(4 * S, &d_state);  // a, b, mu, sigma2 after the update, dense in list order
    sa = h_state; sb = h_state + S; smu = h_state + 2 * S; ss2 = h_state + 3 * S;
  } else {
    FrameTable frames(dev, L);
    const int i_cur = frames.indexOf(frame.get());
    int32_t* d_batch; float *d_a, *d_b, *d_mu, *d_zr, *d_s2;
    int32_t* cur = a.alloc<int32_t>(S, &d_cur);
    int32_t* batch = a.alloc<int32_t>(S, &d_batch);
    float* szr = a.alloc<float>(S, &d_zr);
    ftr.alloc(a, S);
    // the seed state is updated in place by the kernel: it lives in the output block, is filled
    // below and travels both ways (uploadAll / download)
    st.resize(4 * S);
    sa = &st[0]; sb = &st[S]; smu = &st[2 * S]; ss2 = &st[3 * S];
    size_t s = 0;
    for (std::list<Seed>::iterator it = seeds_.begin(); it != seeds_.end(); ++it, ++s) {
      ids[s] = it->id;
      cur[s] = i_cur;
      batch[s] = it->batch_id;
      sa[s] = it->a; sb[s] = it->b; smu[s] = it->mu; szr[s] = it->z_range; ss2[s] = it->sigma2;
      ftr.set(s, frames.indexOf(it->ftr->frame), it->ftr);
    }
    frames.emit(a, &ft);
    a.endInputs();
    {
      float* h;
      h = a.alloc<float>(S, &d_a);  std::copy(sa, sa + S, h);  sa = h;
      h = a.alloc<float>(S, &d_b);  std::copy(sb, sb + S, h);  sb = h;
      h = a.alloc<float>(S, &d_mu); std::copy(smu, smu + S, h); smu = h;
      h = a.alloc<float>(S, &d_s2); std::copy(ss2, ss2 + S, h); ss2 = h;
    }
    seeds.d_a = d_a; seeds.d_b = d_b; seeds.d_mu = d_mu; seeds.d_z_range = d_zr; seeds.d_sigma2 = d_s2; seeds.d_batch_id = d_batch;
  }
  int32_t* status = a.alloc<int32_t>(S, &d_status);
  double* xyz = a.alloc<double>(3 * S, &d_xyz);
  double* px_cur = a.alloc<double>(2 * S, &d_px);

  svo_hip_depth_filter_options opt;
  opt.max_n_kfs = options_.max_n_kfs;
  opt.batch_counter = Seed::batch_counter;
  opt.seed_convergence_sigma2_thresh = options_.seed_convergence_sigma2_thresh;
  opt.align_1d = matcher_.options_.align_1d;
  opt.align_max_iter = matcher_.options_.align_max_iter;
  opt.max_epi_search_steps = (int32_t)matcher_.options_.max_epi_search_steps;
  opt.subpix_refinement = matcher_.options_.subpix_refinement;
  opt.epi_search_edgelet_filtering = matcher_.options_.epi_search_edgelet_filtering;
  opt.n_pyr_levels = Config::nPyrLevels();
  opt.epi_search_edgelet_max_angle = matcher_.options_.epi_search_edgelet_max_angle;
  const svo_hip_camera cam = cameraOf(frame->cam_);
  void* ws = dev.workspace(lane, (int)S);

  stage_timer.device(a.used());
  std::function<void(const double*)> held;  // phase 1: the launches that wait for the frame's pose
  if (resident) {
    a.upload(lane.stream);
    if (rc.patch.n > 0)
      svo_hip::check(svo_hip_seed_store_patch(&rc.patch, &rc.ftr, &rc.seeds, lane.stream), "svo_hip_seed_store_patch");
    if (phase == 1) {
      svo_hip::Device* const pd = &dev;
      svo_hip::Arena* const pa = &a;
      void* const st_ = lane.stream;
      const size_t ws_bytes = lane.workspace_bytes;
      const SeedStore::Call rcv = rc;
      held = [pd, pa, st_, ws_bytes, rcv, cam, ft, opt, S, d_status, d_xyz, d_px, d_state, ws](const double* T_cur) {
        svo_hip::check(svo_hip_update_seeds_resident_pose(&pd->layout(), pd->store(), &cam, &ft, rcv.cur_key, T_cur, (int)S, rcv.d_slot_of,
                                                          &rcv.ftr, &rcv.seeds, &opt, d_status, d_xyz, d_px, d_state, ws, ws_bytes, st_),
                       "svo_hip_update_seeds_resident_pose");
        pa->download(st_);
      };
    } else {
      svo_hip::check(svo_hip_update_seeds_resident(&dev.layout(), dev.store(), &cam, &ft, rc.cur_key, (int)S, rc.d_slot_of, &rc.ftr,
                                                   &rc.seeds, &opt, d_status, d_xyz, d_px, d_state, ws, lane.workspace_bytes, lane.stream),
                     "svo_hip_update_seeds_resident");
    }
  } else {
    a.uploadAll(lane.stream);
    svo_hip::check(svo_hip_update_seeds(&dev.layout(), dev.store(), &cam, &ft, (int)S, d_cur, &ftr.dev, &seeds, &opt, d_status, d_xyz,
                                        d_px, ws, lane.workspace_bytes, lane.stream),
                   "svo_hip_update_seeds");
  }
  if (phase != 1) a.download(lane.stream);

  // ---- replay of the list surgery, in list order (:216-219, :238-245, :255-290) ---------------
  // Seeds are found again by Seed::id (ascending along the list): in deferred mode the list may have lost seeds
  // (removeKeyframe, reset) or gained some at its end (initializeSeeds) before the replay runs.
  const bool is_kf = frame->isKeyframe();
  void* const stream = lane.stream;
  svo_hip::Device* const pdev = &dev;
  SeedStore* const vstore = resident && seedStoreOf(this).verifying() ? &seedStoreOf(this) : NULL;
  const std::function<void()> replay = [this, frame, is_kf, S, ids, status, sa, sb, smu, ss2, xyz, px_cur, vstore]() {
    size_t s = 0;
    for (std::list<Seed>::iterator it = seeds_.begin(); it != seeds_.end() && s < S;) {
      while (s < S && ids[s] < it->id) ++s;  // erased since the update was enqueued
      if (s == S) break;
      if (ids[s] != it->id) { ++it; continue; }
      const int st = status[s];
      const size_t k = s++;
      if (st == SVO_HIP_SEED_BEHIND || st == SVO_HIP_SEED_NOT_IN_FRAME) { ++it; continue; }
      if (st == SVO_HIP_SEED_ERASED_OLD) { it = seeds_.erase(it); continue; }
      it->a = sa[k]; it->b = sb[k]; it->mu = smu[k]; it->sigma2 = ss2[k];
      if (vstore) vstore->reported(it->id, sa[k], sb[k], smu[k], ss2[k]);
      if (st == SVO_HIP_SEED_NO_MATCH) { ++it; continue; }  // b was incremented on the device
      if (is_kf)  // the detector must not start new seeds next to a matched one
        feature_detector_->setGridOccpuancy(Vector2d(px_cur[2 * k], px_cur[2 * k + 1]));
      if (st == SVO_HIP_SEED_CONVERGED) {
        Point* point = new Point(Vector3d(xyz[3 * k], xyz[3 * k + 1], xyz[3 * k + 2]), it->ftr);
        it->ftr->point = point;
        seed_converged_cb_(point, it->sigma2);  // into the map's candidate list
        it = seeds_.erase(it);
      } else if (st == SVO_HIP_SEED_NAN) {
        SVO_WARN_STREAM("z_min is NaN");
        it = seeds_.erase(it);
      } else {
        ++it;
      }
    }
  };
  if (early) {
    // enqueued ahead of the reference's call: the kernels run while the host finishes the frame (see EarlyUpdate)
    stage_timer.unmarshal();  // this call's share is marshal + enqueue; taking it adds the wait and the replay
    eu.valid = phase != 1;
    eu.prepared = phase == 1;
    eu.launch.swap(held);
    eu.frame_id = frame->id_;
    poseToRt(frame->T_f_w_, eu.T);  // (phase 1: not the final pose yet -- phase 2 writes the one it launches with)
    eu.ids = ids;
    eu.replay = replay;
    eu.store = store_guard.store;
    eu.stream = stream;
    store_guard.done();
    EarlyUpdate* const peu = &eu;
    lane.early_drop = [peu, pdev]() {
      peu->valid = false;
      peu->prepared = false;
      peu->launch = nullptr;
      peu->replay = nullptr;
      pdev->countEarlyMapping(false);
      SeedStore* const st = peu->store;
      peu->store = NULL;
      if (st) st->invalidate();  // (first: whatever the wait below does, the shadow claims nothing)
      svo_hip::check(svo_hip_stream_sync(peu->stream), "svo_hip_stream_sync(early update dropped)");
    };
    return;
  }
  if (thread_ == NULL && svo_hip::Device::deferredMapping()) {
    // the caller is the tracking thread itself (addFrame without the mapping thread): leave the kernels running
    stage_timer.unmarshal();  // this call's share is marshal + enqueue; the join adds its wait and the replay
    SeedStore* const pstore = store_guard.store;  // (the filter's destructor joins before it releases the store)
    store_guard.done();
    lane.deferred = [this, replay, stream, pdev, pstore]() {
      const double t0 = svo_hip::StageTimer::now();
      try {
        svo_hip::check(svo_hip_stream_sync(stream), "svo_hip_stream_sync");
      } catch (...) {
        if (pstore) pstore->invalidate();  // the enqueued patch / update did not complete: host and device no longer agree
        throw;
      }
      const double t1 = svo_hip::StageTimer::now();
      lock_t relock(seeds_mut_);
      replay();
      pdev->addStage(svo_hip::Device::STAGE_DEPTH_FILTER, 0.0, t1 - t0, svo_hip::StageTimer::now() - t1, 0.0, false);
    };
    return;
  }
  dev.finish(lane);
  store_guard.done();
  stage_timer.unmarshal();
  replay();  // seeds_mut_ is held
}

// The reference's destructor (depth_filter.cpp:58-62) behind the join a deferred update needs: its closure writes into
// THIS filter's seed list and feature detector, and would otherwise run -- from the next beginCall() of the mapping lane
// or the next reprojectMap -- after they are gone.  (No-op unless SVO_HIP_MAPPER=deferred left an update pending.)
DepthFilter::~DepthFilter() {
  svo_hip::Device::joinDeferredAll();
  {  // an early update still pending writes nothing back; the hook must not outlive the filter
    hip_dropin::EarlyRegistry& r = hip_dropin::earlyUpdates();
    svo_hip::Lane* lane = NULL;
    {
      std::lock_guard<std::mutex> g(r.mut);
      std::map<const DepthFilter*, hip_dropin::EarlyUpdate>::iterator it = r.all.find(this);
      if (it != r.all.end()) lane = it->second.lane;
    }
    if (lane != NULL) {
      std::lock_guard<std::mutex> g(lane->mut);
      if (lane->early_drop) {
        std::function<void()> drop;
        drop.swap(lane->early_drop);
        try { drop(); } catch (...) {}
      }
      lane->early_hook = nullptr;
    }
    std::lock_guard<std::mutex> g(r.mut);
    r.all.erase(this);
  }
  stopThread();
  hip_dropin::releaseSeedStore(this);
  SVO_INFO_STREAM("DepthFilter destructed.");
}

// ---- the two static helpers, also on the device (single measurement) --------------------------
void DepthFilter::updateSeed(const float x, const float tau2, Seed* seed) {
  svo_hip::Device& dev = svo_hip::Device::instance();
  if (!dev.configured()) throw svo_hip::Error("DepthFilter::updateSeed: device context not configured yet");
  const int L = svo_hip::Device::LANE_MAPPING;
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
  dev.beginCall(lane);
  svo_hip::Arena& a = lane.arena;
  a.reset();
  float* d[7];
  float* h[7];
  for (int i = 0; i < 7; ++i) h[i] = a.alloc<float>(1, &d[i]);
  *h[0] = x; *h[1] = tau2; *h[2] = seed->a; *h[3] = seed->b; *h[4] = seed->mu; *h[5] = seed->z_range; *h[6] = seed->sigma2;
  a.endInputs();
  svo_hip_seeds sd;
  sd.d_a = d[2]; sd.d_b = d[3]; sd.d_mu = d[4]; sd.d_z_range = d[5]; sd.d_sigma2 = d[6]; sd.d_batch_id = NULL;
  a.upload(lane.stream);
  svo_hip::check(svo_hip_update_seed_batch(1, d[0], d[1], &sd, lane.stream), "svo_hip_update_seed_batch");
  for (int i = 2; i < 7; ++i) a.fetch(h[i], 1, lane.stream);
  svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  seed->a = *h[2]; seed->b = *h[3]; seed->mu = *h[4]; seed->sigma2 = *h[6];
}

double DepthFilter::computeTau(const SE3& T_ref_cur, const Vector3d& f, const double z, const double px_error_angle) {
  svo_hip::Device& dev = svo_hip::Device::instance();
  if (!dev.configured()) throw svo_hip::Error("DepthFilter::computeTau: device context not configured yet");
  const int L = svo_hip::Device::LANE_MAPPING;
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
  dev.beginCall(lane);
  svo_hip::Arena& a = lane.arena;
  a.reset();
  double *d_t, *d_f, *d_z, *d_tau;
  double* ht = a.alloc<double>(3, &d_t);
  double* hf = a.alloc<double>(3, &d_f);
  double* hz = a.alloc<double>(1, &d_z);
  const Vector3d t(T_ref_cur.translation());
  for (int k = 0; k < 3; ++k) { ht[k] = t[k]; hf[k] = f[k]; }
  *hz = z;
  a.endInputs();
  double* tau = a.alloc<double>(1, &d_tau);
  a.upload(lane.stream);
  svo_hip::check(svo_hip_compute_tau_batch(1, d_t, d_f, d_z, px_error_angle, d_tau, lane.stream), "svo_hip_compute_tau_batch");
  a.download(lane.stream);
  svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
  return *tau;
}

}  // namespace svo
