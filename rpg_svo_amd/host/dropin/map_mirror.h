// map_mirror.h -- row N2 on the host: keeps the device-resident mirror of the map (include/svo_hip.h: svo_hip_map) in
// step with the reference's pointer graph, for the drop-in body of Reprojector::reprojectMap.
//
// What reprojectMap reads, every frame (svo/src/reprojector.cpp:64-142): Map::keyframes_ -> Frame::fts_ ->
// Feature::point -> Point::{pos_, type_, obs_} and MapPointCandidates::candidates_ -- ~2000 list nodes.  The mirror
// holds those records in HBM; per frame the host sends only what changed:
//
//   * STRUCTURE changes when a keyframe is inserted / removed (frame_handler_mono.cpp:192-232: addFrameRef for every
//     feature, candidates promoted into the frame, the furthest keyframe and its points deleted).  Detected without a
//     walk -- the ids of Map::keyframes_ in order, and every keyframe's fts_.size() -- and answered with a REBUILD: one
//     walk of the keyframes and the candidate list (once per keyframe, ~25 frames apart on the reference's trace).
//   * POSITIONS change in FrameHandlerBase::optimizeStructure (Point::optimize, <= 20 of the points the last frame
//     observed): the points the previous call selected are re-read (~120 loads).
//   * TYPES / DELETIONS happen in this call's own cell loop (reprojector.cpp:165-180) and candidate loop (:108-123):
//     recorded as they are made.
//   * NEW CANDIDATES are appended to candidates_ by the depth filter's callback (map.cpp:213-218): the tail of the list
//     beyond the last known candidate is read under the list's mutex.
//
// Everything else the reference's control plane could do to these records without one of the above (bundle adjustment,
// which the reference compiles out by default; a host that edits the map from outside) is caught by
// SVO_HIP_MAP_MIRROR=verify, which re-walks the graph on every call and throws on the first difference -- the mode the
// tests run in.  SVO_HIP_MAP_MIRROR=off keeps the list-walking path of reprojector.cpp (also taken, per frame, whenever
// the map does not fit the mirror's limits or its graph is not the consistent one the mirror assumes).
#ifndef SVO_HIP_DROPIN_MAP_MIRROR_H_
#define SVO_HIP_DROPIN_MAP_MIRROR_H_

#include <cstdlib>
#include <cstring>
#include <iterator>
#include <list>
#include <string>
#include <unordered_map>
#include <vector>

#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/map.h>
#include <svo/point.h>

#include "marshal.h"

namespace svo {
namespace hip_dropin {

class MapMirror {
 public:
  typedef MapPointCandidates::PointCandidateList CandList;
  enum Mode { OFF = 0, ON = 1, VERIFY = 2 };
  static Mode mode() {
    static const Mode m = [] {
      const char* v = std::getenv("SVO_HIP_MAP_MIRROR");
      if (!v || std::string(v) == "on") return ON;
      if (std::string(v) == "off") return OFF;
      if (std::string(v) == "verify") return VERIFY;
      throw svo_hip::Error("SVO_HIP_MAP_MIRROR must be 'on', 'off' or 'verify'");
    }();
    return m;
  }

  struct Entry {
    Point* pt;
    double pos[3];
    int32_t type;        // Point::type_ as the device holds it; 0 = dead entry
    int32_t order;       // candidates: position key in candidates_
    int32_t obs_begin, obs_count;
    bool is_cand;
    CandList::iterator cand;  // valid while is_cand && type != 0
  };
  struct Stats {
    uint64_t calls, rebuilds, fallbacks, patched_points, patched_obs, second_batches;
    Stats() : calls(0), rebuilds(0), fallbacks(0), patched_points(0), patched_obs(0), second_batches(0) {}
  };

  MapMirror() : obs_sent_(0), live_cands_(0), next_order_(0), valid_(false), cap_points_(0), cap_obs_(0), cap_trials_(0),
                d_cell_rank_(NULL), d_point_px_(NULL), d_trial_cur_(NULL), d_trial_pos_(NULL), d_trial_obs_begin_(NULL),
                d_trial_obs_end_(NULL), d_trial_cell_(NULL) {
    std::memset(&dmap_, 0, sizeof(dmap_));
  }
  ~MapMirror() { releaseDevice(); }

  const std::vector<Entry>& entries() const { return pts_; }
  Feature* obsFeature(int32_t o) const { return obs_ftr_[(size_t)o]; }
  const std::vector<const Frame*>& frames() const { return frames_; }
  int frameIndex(const Frame* f) const {
    std::unordered_map<const Frame*, int>::const_iterator it = frame_index_.find(f);
    return it == frame_index_.end() ? -1 : it->second;
  }
  Stats stats;

  // ---- keeping the shadow in step ---------------------------------------------------------------------------------
  // false: the map cannot be mirrored as it is (limits, inconsistent graph): the caller takes its list-walking path
  bool sync(Map& map) {
    if (!valid_ || structureChanged(map)) {
      if (!rebuild(map)) { valid_ = false; return false; }
      valid_ = true;
      ++stats.rebuilds;
    } else {
      recheckWatched();
      if (!appendNewCandidates(map)) {
        if (!rebuild(map)) { valid_ = false; return false; }
        ++stats.rebuilds;
      }
    }
    if (mode() == VERIFY) verify(map);
    return true;
  }

  // The caller hands this frame to the list-walking path: what that path does to the map (types, deletions, the points
  // Point::optimize moves afterwards) goes unobserved, so the next call starts from a fresh walk.
  void invalidate() { valid_ = false; }

  // this call's own changes (reprojector.cpp:108-123, 165-180)
  void markType(int32_t e, int32_t type) { pts_[(size_t)e].type = type; touch(e); }
  void markDead(int32_t e) {
    Entry& x = pts_[(size_t)e];
    if (x.type == 0) return;
    if (x.is_cand) --live_cands_;
    x.type = 0;
    touch(e);
  }
  void watch(const std::vector<int32_t>& selected) { watch_ = selected; }
  // live candidates in list order (entries appended since the last rebuild are in list order by construction)
  const std::vector<int32_t>& candidateEntries() {
    size_t k = 0;  // drop the dead ones
    for (size_t i = 0; i < cand_entries_.size(); ++i)
      if (pts_[(size_t)cand_entries_[i]].type != 0) cand_entries_[k++] = cand_entries_[i];
    cand_entries_.resize(k);
    return cand_entries_;
  }

  // changes recorded since the last emitPatch() (a caller that enqueued a launch on the last patch asks whether the map
  // has moved on since)
  bool pending() const { return !dirty_.empty() || obs_sent_ != obs_ftr_.size(); }

  // ---- the device side --------------------------------------------------------------------------------------------
  size_t patchBytes() const { return dirty_.size() * 64 + (obs_ftr_.size() - obs_sent_) * 96 + 4096; }
  // device buffers for the current shadow (grown geometrically; growing re-sends everything) and the visiting ranks
  void ensureDevice(const std::vector<int>& cell_order, size_t max_trials) {
    if (pts_.size() > cap_points_ || obs_ftr_.size() > cap_obs_) {
      size_t cp = 4096, co = 8192;
      while (cp < pts_.size()) cp <<= 1;
      while (co < obs_ftr_.size()) co <<= 1;
      releaseMap();
      allocMap(cp, co);
      resendAll();
    }
    if (cell_order != cell_order_) {
      if (d_cell_rank_) svo_hip_free(d_cell_rank_);
      d_cell_rank_ = NULL;
      std::vector<int32_t> rank(cell_order.size());
      for (size_t i = 0; i < cell_order.size(); ++i) rank[(size_t)cell_order[i]] = (int32_t)i;
      void* p = NULL;
      svo_hip::check(svo_hip_malloc(&p, rank.size() * sizeof(int32_t)), "svo_hip_malloc(cell rank)");
      d_cell_rank_ = static_cast<int32_t*>(p);
      svo_hip::check(svo_hip_memcpy_h2d(d_cell_rank_, &rank[0], rank.size() * sizeof(int32_t), NULL), "cell rank upload");
      svo_hip::check(svo_hip_stream_sync(NULL), "svo_hip_stream_sync");
      cell_order_ = cell_order;
    }
    if (max_trials > cap_trials_) {
      releaseTrials();
      size_t c = 1024;
      while (c < max_trials) c <<= 1;
      d_trial_cur_ = devAlloc<int32_t>(c);
      d_trial_pos_ = devAlloc<double>(3 * c);
      d_trial_obs_begin_ = devAlloc<int32_t>(c);
      d_trial_obs_end_ = devAlloc<int32_t>(c);
      d_trial_cell_ = devAlloc<int32_t>(c);
      cap_trials_ = c;
    }
  }
  const int32_t* cellRank() const { return d_cell_rank_; }
  svo_hip_map deviceMap() const {
    svo_hip_map m = dmap_;
    m.n_points = (int32_t)pts_.size();
    m.n_obs = (int32_t)obs_ftr_.size();
    return m;
  }
  svo_hip_features deviceObs() const {
    svo_hip_features f;
    f.d_frame = dmap_.d_obs_frame; f.d_level = dmap_.d_obs_level; f.d_type = dmap_.d_obs_type;
    f.d_px = dmap_.d_obs_px; f.d_f = dmap_.d_obs_f; f.d_grad = dmap_.d_obs_grad;
    return f;
  }
  double* pointPx() const { return d_point_px_; }
  int32_t* trialCur() const { return d_trial_cur_; }
  double* trialPos() const { return d_trial_pos_; }
  int32_t* trialObsBegin() const { return d_trial_obs_begin_; }
  int32_t* trialObsEnd() const { return d_trial_obs_end_; }
  int32_t* trialCell() const { return d_trial_cell_; }

  // the pending changes as a svo_hip_map_patch in the arena's INPUT block (before endInputs()); clears them
  svo_hip_map_patch emitPatch(svo_hip::Arena& a) {
    svo_hip_map_patch p;
    std::memset(&p, 0, sizeof(p));
    const size_t n = dirty_.size(), m = obs_ftr_.size() - obs_sent_;
    p.n_points = (int32_t)n;
    p.n_obs = (int32_t)m;
    if (n) {
      int32_t *d_index, *d_type, *d_order, *d_ob, *d_oc; double* d_pos;
      int32_t* index = a.alloc<int32_t>(n, &d_index);
      double* pos = a.alloc<double>(3 * n, &d_pos);
      int32_t* type = a.alloc<int32_t>(n, &d_type);
      int32_t* order = a.alloc<int32_t>(n, &d_order);
      int32_t* ob = a.alloc<int32_t>(n, &d_ob);
      int32_t* oc = a.alloc<int32_t>(n, &d_oc);
      for (size_t i = 0; i < n; ++i) {
        const Entry& e = pts_[(size_t)dirty_[i]];
        index[i] = dirty_[i];
        pos[3 * i] = e.pos[0]; pos[3 * i + 1] = e.pos[1]; pos[3 * i + 2] = e.pos[2];
        type[i] = e.type; order[i] = e.order; ob[i] = e.obs_begin; oc[i] = e.obs_count;
        dirty_flag_[(size_t)dirty_[i]] = 0;
      }
      p.d_index = d_index; p.d_pos = d_pos; p.d_type = d_type; p.d_order = d_order; p.d_obs_begin = d_ob; p.d_obs_count = d_oc;
    }
    if (m) {
      int32_t *d_oi, *d_oo;
      int32_t* oi = a.alloc<int32_t>(m, &d_oi);
      int32_t* oo = a.alloc<int32_t>(m, &d_oo);
      FeatureColumns cols;
      cols.alloc(a, m);
      for (size_t i = 0; i < m; ++i) {
        const size_t o = obs_sent_ + i;
        oi[i] = (int32_t)o;
        oo[i] = obs_order_[o];
        cols.set(i, obs_frame_[o], obs_ftr_[o]);
      }
      p.d_obs_index = d_oi; p.d_obs_order = d_oo; p.obs = cols.dev;
    }
    stats.patched_points += n;
    stats.patched_obs += m;
    dirty_.clear();
    obs_sent_ = obs_ftr_.size();
    return p;
  }

 private:
  MapMirror(const MapMirror&);
  void touch(int32_t e) {
    if (!dirty_flag_[(size_t)e]) { dirty_flag_[(size_t)e] = 1; dirty_.push_back(e); }
  }
  void resendAll() {
    dirty_.clear();
    dirty_flag_.assign(pts_.size(), 0);
    for (size_t e = 0; e < pts_.size(); ++e) touch((int32_t)e);
    obs_sent_ = 0;
  }

  bool structureChanged(const Map& map) const {
    if (map.keyframes_.size() != kf_ids_.size()) return true;
    size_t i = 0;
    for (std::list<FramePtr>::const_iterator kf = map.keyframes_.begin(); kf != map.keyframes_.end(); ++kf, ++i)
      if ((*kf)->id_ != kf_ids_[i] || (*kf)->fts_.size() != kf_nfts_[i]) return true;
    return false;
  }

  int addFrame(const Frame* f) {
    std::unordered_map<const Frame*, int>::const_iterator it = frame_index_.find(f);
    if (it != frame_index_.end()) return it->second;
    const int idx = (int)frames_.size();
    frames_.push_back(f);
    frame_index_[f] = idx;
    return idx;
  }

  // Point::obs_ of entry e, in list order, appended to the observation records
  bool appendObs(Entry& e, const std::unordered_map<const Feature*, int>& ftr_order) {
    e.obs_begin = (int32_t)obs_ftr_.size();
    for (std::list<Feature*>::const_iterator o = e.pt->obs_.begin(); o != e.pt->obs_.end(); ++o) {
      const Feature* ftr = *o;
      if (ftr == NULL || ftr->frame == NULL) return false;
      std::unordered_map<const Feature*, int>::const_iterator fo = ftr_order.find(ftr);
      obs_ftr_.push_back(*o);
      obs_frame_.push_back(addFrame(ftr->frame));
      obs_order_.push_back(fo == ftr_order.end() ? -1 : fo->second);
    }
    e.obs_count = (int32_t)obs_ftr_.size() - e.obs_begin;
    return true;
  }

  bool rebuild(Map& map) {
    pts_.clear(); obs_ftr_.clear(); obs_frame_.clear(); obs_order_.clear();
    frames_.clear(); frame_index_.clear(); kf_ids_.clear(); kf_nfts_.clear();
    cand_entries_.clear(); watch_.clear();
    live_cands_ = 0; next_order_ = 0;
    std::unordered_map<const Feature*, int> ftr_order;
    std::unordered_map<const Point*, int32_t> entry_of;
    std::vector<int> appearances;  // per entry: features of keyframes that point at it
    for (std::list<FramePtr>::const_iterator kf = map.keyframes_.begin(); kf != map.keyframes_.end(); ++kf) {
      const Frame* f = kf->get();
      addFrame(f);
      kf_ids_.push_back(f->id_);
      kf_nfts_.push_back(f->fts_.size());
      if (f->fts_.size() >= 4096) return false;  // (the position in fts_ travels as 12 bits)
      int ord = 0;
      for (Features::const_iterator it = f->fts_.begin(); it != f->fts_.end(); ++it, ++ord) {
        ftr_order[*it] = ord;
        Point* pt = (*it)->point;
        if (pt == NULL) continue;
        // a keyframe's feature points at a live map point (map.cpp:75-99 keeps it so)
        if (pt->type_ != Point::TYPE_UNKNOWN && pt->type_ != Point::TYPE_GOOD) return false;
        std::unordered_map<const Point*, int32_t>::iterator e = entry_of.find(pt);
        if (e == entry_of.end()) {
          Entry x;
          x.pt = pt; x.is_cand = false; x.order = 0; x.obs_begin = x.obs_count = 0;
          x.type = pt->type_ == Point::TYPE_GOOD ? 3 : 2;
          for (int k = 0; k < 3; ++k) x.pos[k] = pt->pos_[k];
          entry_of[pt] = (int32_t)pts_.size();
          pts_.push_back(x);
          appearances.push_back(1);
        } else {
          ++appearances[(size_t)e->second];
        }
      }
    }
    for (size_t e = 0; e < pts_.size(); ++e) {
      if (!appendObs(pts_[e], ftr_order)) return false;
      // every keyframe feature that points at the point is one of the point's observations (point.cpp:60-63,
      // frame_handler_mono.cpp:197-199): the device finds "where the keyframe loop meets the point" in obs_
      int in_lists = 0;
      for (int32_t o = pts_[e].obs_begin; o < pts_[e].obs_begin + pts_[e].obs_count; ++o) {
        if (obs_order_[(size_t)o] < 0) continue;
        ++in_lists;
        if (obs_ftr_[(size_t)o]->point != pts_[e].pt) return false;
      }
      if (in_lists != appearances[e]) return false;
    }
    {
      boost::unique_lock<boost::mutex> lock(map.point_candidates_.mut_);
      CandList& cl = map.point_candidates_.candidates_;
      for (CandList::iterator c = cl.begin(); c != cl.end(); ++c)
        if (!appendCandidate(c, ftr_order)) return false;
    }
    if (pts_.size() > 8192 || frames_.size() + 1 > 64) return false;
    resendAll();
    return true;
  }

  bool appendCandidate(CandList::iterator c, const std::unordered_map<const Feature*, int>& ftr_order) {
    Point* pt = c->first;
    if (pt == NULL || pt->type_ != Point::TYPE_CANDIDATE || next_order_ >= 65000) return false;
    Entry x;
    x.pt = pt; x.is_cand = true; x.cand = c; x.type = 1; x.order = next_order_++;
    for (int k = 0; k < 3; ++k) x.pos[k] = pt->pos_[k];
    if (!appendObs(x, ftr_order)) return false;
    const int32_t e = (int32_t)pts_.size();
    pts_.push_back(x);
    cand_entries_.push_back(e);
    ++live_cands_;
    if (dirty_flag_.size() < pts_.size()) dirty_flag_.resize(pts_.size(), 0);
    touch(e);
    return true;
  }

  // the depth filter's callback appends to candidates_ (map.cpp:213-218): read the tail beyond the last candidate known
  bool appendNewCandidates(Map& map) {
    boost::unique_lock<boost::mutex> lock(map.point_candidates_.mut_);
    CandList& cl = map.point_candidates_.candidates_;
    const size_t n = cl.size();
    if (n < live_cands_) return false;  // somebody else erased: not an event the mirror follows
    CandList::iterator first_new = cl.end();
    for (size_t k = live_cands_; k < n; ++k) --first_new;
    // the node before the new ones is the last candidate the mirror knows
    const std::vector<int32_t>& live = candidateEntries();
    if (live.size() != live_cands_) return false;
    if (live_cands_ > 0) {
      if (first_new == cl.begin()) return false;
      CandList::iterator prev = first_new;
      --prev;
      if (prev != pts_[(size_t)live.back()].cand) return false;
    } else if (first_new != cl.begin()) {
      return false;
    }
    if (pts_.size() + (n - live_cands_) > 8192) return false;
    static const std::unordered_map<const Feature*, int> none;  // a candidate's Feature is in no keyframe's list yet
    for (CandList::iterator c = first_new; c != cl.end(); ++c)
      if (!appendCandidate(c, none)) return false;
    // (a candidate's observation may have brought a frame the table did not hold: the limits rebuild() enforces hold
    // on this path too -- false sends the caller through rebuild(), which hands the frame to the list-walking path)
    if (pts_.size() > 8192 || frames_.size() + 1 > 64) return false;
    return true;
  }

  // FrameHandlerBase::optimizeStructure moved some of the points the last frame observed (frame_handler_base.cpp:178-196)
  void recheckWatched() {
    for (size_t i = 0; i < watch_.size(); ++i) {
      Entry& e = pts_[(size_t)watch_[i]];
      if (e.type == 0) continue;
      const Point* pt = e.pt;
      if (pt->pos_[0] != e.pos[0] || pt->pos_[1] != e.pos[1] || pt->pos_[2] != e.pos[2]) {
        for (int k = 0; k < 3; ++k) e.pos[k] = pt->pos_[k];
        touch(watch_[i]);
      }
    }
  }

  // SVO_HIP_MAP_MIRROR=verify: the shadow against a fresh walk of the graph
  void verify(Map& map) {
    MapMirror fresh;
    if (!fresh.rebuild(map)) throw svo_hip::Error("MapMirror::verify: the map cannot be mirrored");
    // entries of the shadow may be dead or in another order (candidates are appended): compare by point
    std::unordered_map<const Point*, const Entry*> mine;
    size_t live = 0;
    for (size_t e = 0; e < pts_.size(); ++e)
      if (pts_[e].type != 0) { mine[pts_[e].pt] = &pts_[e]; ++live; }
    int32_t last_order = -1;
    size_t late = 0;  // candidates the mapping thread appended between this call's tail read and this walk
    for (size_t e = 0; e < fresh.pts_.size(); ++e) {
      const Entry& f = fresh.pts_[e];
      std::unordered_map<const Point*, const Entry*>::const_iterator it = mine.find(f.pt);
      if (it == mine.end()) {
        if (f.is_cand) { ++late; continue; }
        throw svo_hip::Error("MapMirror::verify: a point of the map is missing from the mirror");
      }
      if (late) throw svo_hip::Error("MapMirror::verify: a candidate is missing from the mirror");  // (late ones are the list's tail)
      const Entry& m = *it->second;
      if (m.type != f.type || m.pos[0] != f.pos[0] || m.pos[1] != f.pos[1] || m.pos[2] != f.pos[2] || m.obs_count != f.obs_count ||
          m.is_cand != f.is_cand)
        throw svo_hip::Error("MapMirror::verify: a point's record is stale (type " + std::to_string(m.type) + " / " + std::to_string(f.type) + ")");
      for (int32_t k = 0; k < f.obs_count; ++k) {
        const size_t om = (size_t)(m.obs_begin + k), of = (size_t)(f.obs_begin + k);
        if (obs_ftr_[om] != fresh.obs_ftr_[of] || obs_order_[om] != fresh.obs_order_[of] ||
            frames_[(size_t)obs_frame_[om]] != fresh.frames_[(size_t)fresh.obs_frame_[of]])
          throw svo_hip::Error("MapMirror::verify: an observation record is stale");
      }
      if (f.is_cand) {  // list order is what the order keys say
        if (m.order <= last_order) throw svo_hip::Error("MapMirror::verify: candidate order keys do not follow the list");
        last_order = m.order;
      }
    }
    if (live + late != fresh.pts_.size())
      throw svo_hip::Error("MapMirror::verify: " + std::to_string(live) + " live entries, the map has " + std::to_string(fresh.pts_.size()));
  }

  // ---- device buffers ------------------------------------------------------------------------------------------------
  template <typename T> static T* devAlloc(size_t n) {
    void* p = NULL;
    svo_hip::check(svo_hip_malloc(&p, (n ? n : 1) * sizeof(T)), "svo_hip_malloc(map mirror)");
    return static_cast<T*>(p);
  }
  void allocMap(size_t cp, size_t co) {
    dmap_.d_pos = devAlloc<double>(3 * cp);
    dmap_.d_type = devAlloc<int32_t>(cp);
    dmap_.d_order = devAlloc<int32_t>(cp);
    dmap_.d_obs_begin = devAlloc<int32_t>(cp);
    dmap_.d_obs_count = devAlloc<int32_t>(cp);
    dmap_.d_obs_frame = devAlloc<int32_t>(co);
    dmap_.d_obs_order = devAlloc<int32_t>(co);
    dmap_.d_obs_level = devAlloc<int32_t>(co);
    dmap_.d_obs_type = devAlloc<uint8_t>(co);
    dmap_.d_obs_px = devAlloc<double>(2 * co);
    dmap_.d_obs_f = devAlloc<double>(3 * co);
    dmap_.d_obs_grad = devAlloc<double>(2 * co);
    d_point_px_ = devAlloc<double>(2 * cp);
    cap_points_ = cp;
    cap_obs_ = co;
  }
  void releaseMap() {
    void* ps[] = {dmap_.d_pos, dmap_.d_type, dmap_.d_order, dmap_.d_obs_begin, dmap_.d_obs_count, dmap_.d_obs_frame, dmap_.d_obs_order,
                  dmap_.d_obs_level, dmap_.d_obs_type, dmap_.d_obs_px, dmap_.d_obs_f, dmap_.d_obs_grad, d_point_px_};
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); ++i)
      if (ps[i]) svo_hip_free(ps[i]);
    std::memset(&dmap_, 0, sizeof(dmap_));
    d_point_px_ = NULL;
    cap_points_ = cap_obs_ = 0;
  }
  void releaseTrials() {
    void* ps[] = {d_trial_cur_, d_trial_pos_, d_trial_obs_begin_, d_trial_obs_end_, d_trial_cell_};
    for (size_t i = 0; i < sizeof(ps) / sizeof(ps[0]); ++i)
      if (ps[i]) svo_hip_free(ps[i]);
    d_trial_cur_ = d_trial_obs_begin_ = d_trial_obs_end_ = d_trial_cell_ = NULL;
    d_trial_pos_ = NULL;
    cap_trials_ = 0;
  }
  void releaseDevice() {
    releaseMap();
    releaseTrials();
    if (d_cell_rank_) svo_hip_free(d_cell_rank_);
    d_cell_rank_ = NULL;
  }

  std::vector<Entry> pts_;
  std::vector<Feature*> obs_ftr_;     // observation record -> the reference's Feature
  std::vector<int32_t> obs_frame_;    //                    -> index into frames_
  std::vector<int32_t> obs_order_;    //                    -> position in its frame's fts_, -1: in no keyframe's list
  std::vector<const Frame*> frames_;  // the mirror's frame table: the map's keyframes in list order, then any other frame
  std::unordered_map<const Frame*, int> frame_index_;
  std::vector<int> kf_ids_;
  std::vector<size_t> kf_nfts_;
  std::vector<int32_t> dirty_;
  std::vector<char> dirty_flag_;
  size_t obs_sent_;                   // observation records [obs_sent_, size) are still to be sent
  std::vector<int32_t> watch_;
  std::vector<int32_t> cand_entries_;
  size_t live_cands_;
  int32_t next_order_;
  bool valid_;
  svo_hip_map dmap_;
  size_t cap_points_, cap_obs_, cap_trials_;
  std::vector<int> cell_order_;
  int32_t* d_cell_rank_;
  double* d_point_px_;
  int32_t* d_trial_cur_;
  double* d_trial_pos_;
  int32_t *d_trial_obs_begin_, *d_trial_obs_end_, *d_trial_cell_;
};

void mapMirrorStats(uint64_t out[6]);  // reprojector.cpp

}  // namespace hip_dropin
}  // namespace svo
#endif  // SVO_HIP_DROPIN_MAP_MIRROR_H_
