// seed_store.h -- row N2, seeds: the host side of the device-resident seed store (include/svo_hip.h:
// svo_hip_seed_store_patch / svo_hip_update_seeds_resident).
//
// DepthFilter::seeds_ (depth_filter.h:140) is a std::list<Seed>; rounds 1-4 flattened the whole list -- state and the
// Feature of every seed, 89 bytes each -- into the arena and shipped it both ways on every updateSeeds call.  The
// state is only ever changed by the update itself and a seed's Feature never changes, so both live in HBM now:
//   * a seed gets a SLOT of the store when the drop-in first meets it (the reference's own initializeSeeds appended
//     it, depth_filter.cpp:114-132) and keeps it until it leaves the list; new records travel once, as a patch;
//   * per call the host sends the slots in LIST ORDER (4 bytes per seed) and the frame table; the kernels read and
//     update the records in place and return status / state / points densely, in list order, for the replay of the
//     list surgery (depth_filter.cpp:216-219, 238-245, 255-290), which is unchanged.
// The store is told nothing by the reference's code.  It follows the list by what the list guarantees: Seed::id
// ascends along it (seeds are appended with a running counter and only ever erased), so ONE merge walk per call finds
// the seeds that left (the replay's own erasures, removeKeyframe, reset) and the ones that arrived.
// Feature::frame is stored as a KEY into a small table of the keyframes that still have seeds (key 0: the frame being
// processed), stable while the keyframe has seeds.
#ifndef SVO_HIP_DROPIN_SEED_STORE_H_
#define SVO_HIP_DROPIN_SEED_STORE_H_

#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <list>
#include <map>
#include <string>
#include <vector>

#include <svo/depth_filter.h>
#include <svo/feature.h>
#include <svo/frame.h>

#include "marshal.h"

namespace svo {
namespace hip_dropin {

class SeedStore {
 public:
  struct Stats {
    uint64_t calls, rebuilds, records_sent, seeds_released, invalidations, edits_found;
    Stats() : calls(0), rebuilds(0), records_sent(0), seeds_released(0), invalidations(0), edits_found(0) {}
  };
  Stats stats;

  SeedStore() : high_(0), cap_(0), verify_(false) {
    clearDevice();
    const char* v = std::getenv("SVO_HIP_SEED_STORE");
    verify_ = v && std::string(v) == "verify";
  }
  // verify mode: what the device reported for a seed after an update (the replay of depth_filter.cpp hands it over), to be
  // compared with the host's copy at the next call
  bool verifying() const { return verify_; }
  void reported(int id, float a, float b, float mu, float sigma2) {
    Reported& r = reported_[id];
    r.a = a; r.b = b; r.mu = mu; r.sigma2 = sigma2;
  }
  ~SeedStore() { releaseDevice(); }

  // What one call hands to the device (arena input blocks and the store's columns).
  struct Call {
    int S;                    // seeds in the list
    const int32_t* d_slot_of;  // [S] list order -> slot
    svo_hip_seed_patch patch;  // new records (n may be 0)
    svo_hip_frames frames;     // table indexed by key; `cur_key` is the frame being processed
    int cur_key;
    svo_hip_features ftr;      // the store's columns
    svo_hip_seeds seeds;
  };

  // Brings the shadow in step with `seeds` and writes this call's input blocks into the arena (before endInputs()).
  // Uploads nothing itself: the blocks travel with the arena's one H2D copy.
  Call sync(std::list<Seed>& seeds, const Frame* cur, svo_hip::Device& dev, int lane, svo_hip::Arena& a) {
    ++stats.calls;
    // The merge walk below relies on Seed::id ascending strictly along the list.  A host that resets Seed::seed_counter
    // (the reference never does; oracle/ref_driver.cpp does, between independent runs) would alias new seeds onto the
    // slots of old ones by id: checked on every call (one comparison per seed), and answered by forgetting the shadow.
    {
      bool ascending = true;
      int prev = 0;
      bool first = true;
      for (std::list<Seed>::const_iterator it = seeds.begin(); it != seeds.end(); ++it) {
        if (!first && it->id <= prev) { ascending = false; break; }
        prev = it->id; first = false;
      }
      // (ids below the largest id the shadow has seen, appended at the END of the list, are a counter reset too)
      if (ascending && !seeds.empty() && !ids_.empty() && seeds.back().id < ids_.back() &&
          (seeds.size() > ids_.size() || seeds.front().id < ids_.front()))
        ascending = false;
      if (!ascending) invalidate();
    }
    // ---- merge walk: the list against the shadow (both ascend in Seed::id) -------------------------------------
    new_ids_.clear(); new_slot_.clear(); new_key_.clear(); fresh_.clear();
    size_t k = 0;
    for (std::list<Seed>::iterator it = seeds.begin(); it != seeds.end(); ++it) {
      while (k < ids_.size() && ids_[k] < it->id) release(k++);
      if (k < ids_.size() && ids_[k] == it->id && !(verify_ && edited(*it))) {
        new_ids_.push_back(ids_[k]); new_slot_.push_back(slot_[k]); new_key_.push_back(key_[k]);
        ++k;
      } else if (k < ids_.size() && ids_[k] == it->id) {
        // SVO_HIP_SEED_STORE=verify: the host's state of a resident seed is not what the device last reported (an external
        // edit: the static DepthFilter::updateSeed(x, tau2, Seed*), user code): the seed keeps its slot, its record travels again
        new_ids_.push_back(ids_[k]); new_slot_.push_back(slot_[k]); new_key_.push_back(key_[k]);
        fresh_.push_back(std::make_pair(&*it, new_ids_.size() - 1));
        ++stats.edits_found;
        ++k;
      } else {
        const int32_t slot = allocSlot();
        const int32_t key = keyOf(it->ftr->frame);
        new_ids_.push_back(it->id); new_slot_.push_back(slot); new_key_.push_back(key);
        fresh_.push_back(std::make_pair(&*it, new_ids_.size() - 1));
      }
    }
    while (k < ids_.size()) release(k++);
    ids_.swap(new_ids_); slot_.swap(new_slot_); key_.swap(new_key_);
    const size_t S = ids_.size();
    if ((size_t)high_ > cap_) {
      // the columns are too small: new ones, and every record of the list travels again
      size_t c = 1024;
      while (c < (size_t)high_) c <<= 1;
      releaseDevice();
      allocDevice(c);
      ++stats.rebuilds;
      fresh_.clear();
      size_t i = 0;
      for (std::list<Seed>::iterator it = seeds.begin(); it != seeds.end(); ++it, ++i) fresh_.push_back(std::make_pair(&*it, i));
    }
    Call c;
    c.S = (int)S;
    // ---- slots in list order ---------------------------------------------------------------------------------------
    int32_t* d_slot_of;
    int32_t* h_slot_of = a.alloc<int32_t>(S ? S : 1, &d_slot_of);
    for (size_t i = 0; i < S; ++i) h_slot_of[i] = slot_[i];
    c.d_slot_of = d_slot_of;
    // ---- new records ---------------------------------------------------------------------------------------------------
    const size_t n = fresh_.size();
    std::memset(&c.patch, 0, sizeof(c.patch));
    c.patch.n = (int32_t)n;
    if (n) {
      int32_t *d_ps, *d_batch; float *d_a, *d_b, *d_mu, *d_zr, *d_s2;
      int32_t* ps = a.alloc<int32_t>(n, &d_ps);
      int32_t* batch = a.alloc<int32_t>(n, &d_batch);
      float* sa = a.alloc<float>(n, &d_a);
      float* sb = a.alloc<float>(n, &d_b);
      float* smu = a.alloc<float>(n, &d_mu);
      float* szr = a.alloc<float>(n, &d_zr);
      float* ss2 = a.alloc<float>(n, &d_s2);
      FeatureColumns col;
      col.alloc(a, n);
      for (size_t i = 0; i < n; ++i) {
        const Seed* s = fresh_[i].first;
        const size_t pos = fresh_[i].second;
        ps[i] = slot_[pos];
        batch[i] = s->batch_id;
        sa[i] = s->a; sb[i] = s->b; smu[i] = s->mu; szr[i] = s->z_range; ss2[i] = s->sigma2;
        col.set(i, key_[pos], s->ftr);
      }
      c.patch.d_slot = d_ps;
      c.patch.src_ftr = col.dev;
      c.patch.src_seeds.d_a = d_a; c.patch.src_seeds.d_b = d_b; c.patch.src_seeds.d_mu = d_mu;
      c.patch.src_seeds.d_z_range = d_zr; c.patch.src_seeds.d_sigma2 = d_s2; c.patch.src_seeds.d_batch_id = d_batch;
      stats.records_sent += n;
    }
    // ---- the frame table, indexed by key (key 0: the frame being processed) -----------------------------------------------
    const size_t K = kf_.size();
    int32_t* d_fs; double* d_T;
    int32_t* h_fs = a.alloc<int32_t>(K, &d_fs);
    double* h_T = a.alloc<double>(12 * K, &d_T);
    kf_[0] = cur;
    for (size_t key = 0; key < K; ++key) {
      const Frame* f = kf_[key] ? kf_[key] : cur;  // (a free key: nobody refers to it; any resident frame will do)
      const cv::Mat& img = f->img_pyr_[0];
      h_fs[key] = dev.slotOf(f->id_, img.data, (int)img.step.p[0], lane);
      poseToRt(f->T_f_w_, h_T + 12 * key);
    }
    c.frames.n_frames = (int32_t)K;
    c.frames.reserved = 0;
    c.frames.d_slot = d_fs;
    c.frames.d_T_f_w = d_T;
    c.cur_key = 0;
    c.ftr = ftr_;
    c.seeds = seeds_;
    return c;
  }

  size_t size() const { return ids_.size(); }
  const std::vector<int>& ids() const { return ids_; }

  // Forget what the shadow believes the device holds: every seed of the list is met "for the first time" by the next
  // sync() and its record travels again.  sync() commits the shadow (ids / slots / keys) BEFORE the patch kernel is
  // even enqueued, so whenever anything between sync() and the completion of the update fails -- slotOf, the workspace,
  // a launch, the stream's completion in the deferred closure -- the shadow would claim records the device never
  // received (stale slots, a garbage d_frame indexing the frame table).  The caller invalidates on every such path.
  // The device columns are kept (their capacity is still right); slots are handed out from 0 again.
  void invalidate() {
    reported_.clear();
    ids_.clear(); slot_.clear(); key_.clear(); free_.clear();
    kf_.clear(); kf_refs_.clear();
    high_ = 0;
    ++stats.invalidations;
  }

 private:
  struct Reported { float a, b, mu, sigma2; };
  bool edited(const Seed& s) const {
    std::map<int, Reported>::const_iterator r = reported_.find(s.id);
    if (r == reported_.end()) return false;  // never updated since it was sent: the record that travelled is the host's
    // (bit patterns: a NaN state compares equal to itself)
    return std::memcmp(&r->second.a, &s.a, sizeof(float)) != 0 || std::memcmp(&r->second.b, &s.b, sizeof(float)) != 0 ||
           std::memcmp(&r->second.mu, &s.mu, sizeof(float)) != 0 || std::memcmp(&r->second.sigma2, &s.sigma2, sizeof(float)) != 0;
  }
  int32_t allocSlot() {
    if (!free_.empty()) {
      const int32_t s = free_.back();
      free_.pop_back();
      return s;
    }
    return high_++;
  }
  void release(size_t k) {
    if (verify_) reported_.erase(ids_[k]);
    free_.push_back(slot_[k]);
    const int32_t key = key_[k];
    if (--kf_refs_[(size_t)key] == 0) kf_[(size_t)key] = NULL;
    ++stats.seeds_released;
  }
  int32_t keyOf(const Frame* f) {
    if (kf_.empty()) { kf_.push_back(NULL); kf_refs_.push_back(0); }  // key 0: the current frame of a call
    int32_t free_key = -1;
    for (size_t k = 1; k < kf_.size(); ++k) {
      if (kf_[k] == f) { ++kf_refs_[k]; return (int32_t)k; }
      if (kf_[k] == NULL && free_key < 0) free_key = (int32_t)k;
    }
    if (free_key < 0) { kf_.push_back(NULL); kf_refs_.push_back(0); free_key = (int32_t)kf_.size() - 1; }
    kf_[(size_t)free_key] = f;
    kf_refs_[(size_t)free_key] = 1;
    return free_key;
  }

  template <typename T> static T* devAlloc(size_t n) {
    void* p = NULL;
    svo_hip::check(svo_hip_malloc(&p, (n ? n : 1) * sizeof(T)), "svo_hip_malloc(seed store)");
    return static_cast<T*>(p);
  }
  void allocDevice(size_t c) {
    ftr_.d_frame = devAlloc<int32_t>(c); ftr_.d_level = devAlloc<int32_t>(c); ftr_.d_type = devAlloc<uint8_t>(c);
    ftr_.d_px = devAlloc<double>(2 * c); ftr_.d_f = devAlloc<double>(3 * c); ftr_.d_grad = devAlloc<double>(2 * c);
    seeds_.d_a = devAlloc<float>(c); seeds_.d_b = devAlloc<float>(c); seeds_.d_mu = devAlloc<float>(c);
    seeds_.d_z_range = devAlloc<float>(c); seeds_.d_sigma2 = devAlloc<float>(c); seeds_.d_batch_id = devAlloc<int32_t>(c);
    cap_ = c;
  }
  void clearDevice() {
    std::memset(&ftr_, 0, sizeof(ftr_));
    std::memset(&seeds_, 0, sizeof(seeds_));
    cap_ = 0;
  }
  void releaseDevice() {
    const void* p[12] = {ftr_.d_frame, ftr_.d_level, ftr_.d_type, ftr_.d_px, ftr_.d_f, ftr_.d_grad,
                         seeds_.d_a, seeds_.d_b, seeds_.d_mu, seeds_.d_z_range, seeds_.d_sigma2, seeds_.d_batch_id};
    for (int i = 0; i < 12; ++i)
      if (p[i]) svo_hip_free(const_cast<void*>(p[i]));
    clearDevice();
  }

  std::vector<int> ids_, new_ids_;            // Seed::id in list order
  std::vector<int32_t> slot_, new_slot_;      // ... its slot
  std::vector<int32_t> key_, new_key_;        // ... the key of its Feature::frame
  std::vector<std::pair<const Seed*, size_t> > fresh_;  // seeds met for the first time in this call, and their list position
  std::vector<int32_t> free_;
  int32_t high_;                              // slots handed out so far
  size_t cap_;                                // slots the device columns hold
  std::vector<const Frame*> kf_;              // key -> keyframe (NULL: free); [0] the current frame of the call
  std::vector<int> kf_refs_;                  // seeds per key
  svo_hip_features ftr_;
  svo_hip_seeds seeds_;
  bool verify_;                               // SVO_HIP_SEED_STORE=verify
  std::map<int, Reported> reported_;          // ... Seed::id -> the state the device reported last
};

}  // namespace hip_dropin
}  // namespace svo
#endif
