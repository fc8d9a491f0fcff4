// Drop-in for svo::Reprojector::reprojectMap (svo/include/svo/reprojector.h); reprojectCell, which only
// it called, goes away.  Constructor, grid set-up, resetGrid, reprojectPoint and the candidate comparator
// stay in the reference's own svo/src/reprojector.cpp: build that file minus reprojectMap / reprojectCell
// (scripts/strip_members.py) next to this one.
// The map bookkeeping stays on the host exactly as the
// reference has it (which keyframes overlap, which cell a point falls in, one match per
// cell in the shuffled cell order, the n_failed/n_succeeded counters and point deletion,
// reprojector.cpp:64-204); what moves to the MI355X is the expensive part, the
// Matcher::findMatchDirect trials (matcher.cpp:135-177: closest-view reference observation,
// affine warp, 10x10 warped template, inverse-compositional align2D/align1D).  The reference
// runs them one at a time and stops a cell at its first success; here EVERY binned candidate
// is matched in one batched launch (svo_hip_find_match_direct, K2+K3) and the per-cell
// selection then just reads the results.  A trial is a pure function of (point, frame), so
// the selected features are the same.
// The call also PREDICTS the next one: FrameHandlerMono::processFrame hands the frame straight to
// pose_optimizer::optimizeGaussNewton (frame_handler_mono.cpp:164-176), whose input is exactly the features selected
// here.  svo_hip_select_matches applies the selection rule on the device and svo_hip_pose_optimize is enqueued behind
// it; this function returns when the MATCH results have arrived and does its list surgery while the optimizer runs
// (svo_hip::Speculation; the pose optimizer's drop-in checks the prediction before it takes the result).
#include <svo/reprojector.h>

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <stdexcept>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/map.h>
#include <svo/point.h>

#include "frame_chain.h"
#include "map_mirror.h"
#include "marshal.h"

namespace svo {

namespace {
typedef std::pair<FramePtr, double> KfDist;
bool closerKf(const KfDist& a, const KfDist& b) { return a.second < b.second; }

// Point::getCloseViewObs (svo/src/point.cpp:97-117), statement for statement, except that Frame::pos() -- an SE3 inversion
// per call (frame.h:112) -- is looked up: every candidate of a frame asks for the positions of the same dozen keyframes,
// and the batch asks for ALL binned candidates where the reference asks only for the ones it gets to try.
// Reprojector::reprojectCell's cell.sort(pointQualityComparator) (:152; stable: std::list::sort is a merge sort).  Most
// cells hold one candidate or candidates of one type, and libstdc++'s list::sort builds 65 scratch lists before it looks
// at anything: a cell that is in order already -- no neighbour pair the comparator would swap -- is left alone.
// (Reprojector::Cell / Candidate are private: the types are deduced)
template <class CellList>
void sortCell(CellList& cell) {
  typedef typename CellList::value_type Cand;
  typename CellList::iterator a = cell.begin();
  if (a == cell.end()) return;
  typename CellList::iterator b = a;
  for (++b; b != cell.end(); ++a, ++b)
    if (b->pt->type_ > a->pt->type_) {
      cell.sort([](Cand& l, Cand& r) { return l.pt->type_ > r.pt->type_; });
      return;
    }
}

struct FramePositions {
  std::vector<std::pair<const Frame*, Vector3d> > known;
  Vector3d of(const Frame* f) {
    for (size_t i = 0; i < known.size(); ++i)
      if (known[i].first == f) return known[i].second;
    known.push_back(std::make_pair(f, f->pos()));
    return known.back().second;
  }
};
bool closeViewObs(const Point& pt, const Vector3d& framepos, FramePositions& positions, Feature*& ftr) {
  if (pt.obs_.empty()) return false;
  Vector3d obs_dir(framepos - pt.pos_); obs_dir.normalize();
  std::list<Feature*>::const_iterator min_it = pt.obs_.begin();
  double min_cos_angle = 0;
  for (std::list<Feature*>::const_iterator it = pt.obs_.begin(), ite = pt.obs_.end(); it != ite; ++it) {
    Vector3d dir(positions.of((*it)->frame) - pt.pos_); dir.normalize();
    const double cos_angle = obs_dir.dot(dir);
    if (cos_angle > min_cos_angle) {
      min_cos_angle = cos_angle;
      min_it = it;
    }
  }
  ftr = *min_it;
  return !(min_cos_angle < 0.5);  // observations more than 60 degrees apart are useless
}

// ---- row N2: the same call on the device-resident mirror of the map (map_mirror.h, svo_hip_reproject_map) -----------
// One mirror per svo::Map (a process may run several handlers).  An entry lives as long as the Reprojector that reads the map:
// the drop-in's destructor below releases it (the reference's Reprojector holds its Map by reference for its whole life),
// so a Map allocated later at the same address never inherits a shadow full of dangling Point* / Feature* / Frame*.
struct MirrorRegistry {
  std::mutex mut;
  std::map<const Map*, hip_dropin::MapMirror*> all;
};
MirrorRegistry& mirrors() {
  static MirrorRegistry r;
  return r;
}
hip_dropin::MapMirror& mirrorOf(const Map* map) {
  MirrorRegistry& r = mirrors();
  std::lock_guard<std::mutex> g(r.mut);
  hip_dropin::MapMirror*& m = r.all[map];
  if (m == NULL) m = new hip_dropin::MapMirror();
  return *m;
}
void releaseMirror(const Map* map) {
  MirrorRegistry& r = mirrors();
  std::lock_guard<std::mutex> g(r.mut);
  std::map<const Map*, hip_dropin::MapMirror*>::iterator it = r.all.find(map);
  if (it == r.all.end()) return;
  delete it->second;  // host vectors and the device buffers of the shadow (MapMirror::~MapMirror)
  r.all.erase(it);
}

// ---- one batch of the mirrored path: reproject_map -> match kernels -> selection -> predicted pose refinement ---------
// The blocks of a batch in the lane's arena, and the launches on them.  Two users: reprojectMapMirrored's own call (fill
// -> upload -> launch -> wait), and MirrorChain below, which adds the same blocks and launches to the call of
// SparseImgAlign::run's drop-in, behind K1 (frame_chain.h).
struct MirrorBatch {
  static const size_t V_CAP = 4096;
  // ---- what the batch is about (set by the user before allocInputs)
  svo_hip::Device* dev;
  svo_hip::Lane* lane;
  hip_dropin::MapMirror* mm;
  svo_hip_camera cam;
  int cell_size, n_cols, n_rows;
  size_t n_cells, first_cell, max_cells, T_CAP;
  int align_max_iter;
  int i_cur;
  bool predict;
  std::vector<int32_t> rank_of;  // per entry of the frame table: rank among the overlapping keyframes, or -1
  int frame_id;
  // ---- device addresses and their host views
  svo_hip_map_patch patch;
  int32_t* d_rank;
  svo_hip_frames ft;
  double* h_frame_T;  // host view of the frame table's poses (the chain re-reads nothing from it; kept for symmetry)
  double *d_sf, *d_spos; int32_t* d_slvl;
  svo_hip_reprojection out;
  int32_t* header;
  const int32_t *h_point_cell, *h_kf_count, *h_vp, *h_vc, *h_vt, *h_ok, *h_ref, *h_lvl;
  const double *h_px, *h_A;
  double* d_A; int32_t *d_ok, *d_ref, *d_lvl;
  double *d_T, *d_Cov, *d_stats; int32_t *d_nsel, *d_sel, *d_ran, *d_flag; uint8_t* d_has;
  volatile int32_t* flag;
  double* h_T;  // the predicted refinement's in/out pose block (host view)
  size_t inputs_end, match_end, results_begin;

  MirrorBatch() { std::memset(static_cast<void*>(&patch), 0, sizeof(patch)); clear(); }
  void clear() {
    dev = NULL; lane = NULL; mm = NULL; predict = false; d_rank = NULL; h_frame_T = NULL; d_sf = d_spos = NULL; d_slvl = NULL;
    std::memset(&out, 0, sizeof(out)); header = NULL; flag = NULL; d_flag = NULL; d_T = d_Cov = d_stats = NULL; h_T = NULL;
    d_nsel = d_sel = d_ran = NULL; d_has = NULL; inputs_end = match_end = results_begin = 0; frame_id = -1;
  }
  size_t arenaBytes(size_t n_tab) const {
    return ((size_t)1 << 17) + mm->patchBytes() + mm->entries().size() * 8 + T_CAP * 128 + V_CAP * 16 + n_tab * 128;
  }
  // inputs: what changed in the map, the ranks of the overlapping keyframes, the frame table (before Arena::endInputs())
  void allocInputs(svo_hip::Arena& a, const hip_dropin::FrameTable& frames) {
    patch = mm->emitPatch(a);
    int32_t* rank = a.alloc<int32_t>(rank_of.size(), &d_rank);
    std::copy(rank_of.begin(), rank_of.end(), rank);
    frames.emit(a, &ft);
    const size_t cap = (size_t)Config::maxFts() + 1;
    if (predict) {  // the predicted pose refinement's observations: gathered on the device, never read by the host
      a.alloc<double>(3 * cap, &d_sf);
      a.alloc<double>(3 * cap, &d_spos);
      a.alloc<int32_t>(cap, &d_slvl);
    }
  }
  // outputs the host reads (after Arena::endInputs()); sp: the lane's speculation record, filled when predict
  void allocOutputs(svo_hip::Arena& a, size_t n_tab, const FramePtr& frame, svo_hip::Speculation& sp, bool pose_from_device) {
    inputs_end = a.used();
    const size_t P = mm->entries().size();
    header = a.alloc<int32_t>(SVO_HIP_REPROJ_HEADER, &out.d_header);
    header[0] = -1;
    h_point_cell = a.alloc<int32_t>(P ? P : 1, &out.d_point_cell);
    h_kf_count = a.alloc<int32_t>(n_tab, &out.d_kf_count);
    h_vp = a.alloc<int32_t>(V_CAP, &out.d_visit_point);
    h_vc = a.alloc<int32_t>(V_CAP, &out.d_visit_cell);
    h_vt = a.alloc<int32_t>(V_CAP, &out.d_visit_trial);
    h_px = a.alloc<double>(2 * T_CAP, &out.d_trial_px);
    h_ok = a.alloc<int32_t>(T_CAP, &d_ok);
    h_ref = a.alloc<int32_t>(T_CAP, &d_ref);
    h_lvl = a.alloc<int32_t>(T_CAP, &d_lvl);
    h_A = a.alloc<double>(4 * T_CAP, &d_A);
    match_end = a.used();
    out.d_point_px = mm->pointPx();
    out.d_trial_cur = mm->trialCur(); out.d_trial_pos = mm->trialPos();
    out.d_trial_obs_begin = mm->trialObsBegin(); out.d_trial_obs_end = mm->trialObsEnd(); out.d_trial_cell = mm->trialCell();
    results_begin = match_end;
    if (predict) {
      const size_t cap = (size_t)Config::maxFts() + 1;
      sp.frame_id = frame->id_;
      sp.point.clear(); sp.px.clear(); sp.level.clear(); sp.trial.clear();
      results_begin = a.used();
      h_T = a.alloc<double>(12, &d_T);
      hip_dropin::poseToRt(frame->T_f_w_, h_T);  // (pose_from_device: overwritten on the stream, and T_init set when it is known)
      if (!pose_from_device) std::copy(h_T, h_T + 12, sp.T_init);
      sp.T = h_T;
      sp.n_sel = a.alloc<int32_t>(1, &d_nsel);
      sp.sel = a.alloc<int32_t>(cap, &d_sel);
      sp.has_point = a.alloc<uint8_t>(cap, &d_has);
      sp.Cov = a.alloc<double>(36, &d_Cov);
      sp.stats = a.alloc<double>(4, &d_stats);
      sp.ran = a.alloc<int32_t>(1, &d_ran);
      if (a.mode() != svo_hip::Arena::MIRRORED) {
        flag = a.alloc<int32_t>(1, &d_flag);
        *flag = 0;
      }
      sp.reproj_thresh = Config::poseOptimThresh();
      sp.n_iter = (int)Config::poseOptimNumIter();
    }
  }
  // reproject_map -> match kernels -> (predict, flag-capable arena) selection + pose refinement, on the lane's stream
  void launchMatch(svo_hip::Arena& a) {
    const svo_hip_map dmap = mm->deviceMap();
    const svo_hip_features dobs = mm->deviceObs();
    svo_hip_grid g;
    g.cell_size = cell_size; g.n_cols = n_cols; g.n_rows = n_rows; g.n_cells = (int32_t)n_cells;
    g.d_cell_rank = mm->cellRank();
    void* ws = dev->workspace(*lane, (int)T_CAP);
    int32_t* const d_M = out.d_header + 3;
    svo_hip::check(svo_hip_reproject_map(&cam, &ft, i_cur, d_rank, &dmap, &patch, &g, (int)first_cell,
                                         (int)std::min(max_cells, (size_t)1 << 30), (int)V_CAP, (int)T_CAP, &out, lane->stream),
                   "svo_hip_reproject_map");
    svo_hip::check(svo_hip_find_match_direct_indirect(&dev->layout(), dev->store(), &cam, &ft, (int)T_CAP, d_M, out.d_trial_cur,
                                                      out.d_trial_pos, out.d_trial_obs_begin, out.d_trial_obs_end, &dobs,
                                                      Config::nPyrLevels(), align_max_iter, out.d_trial_px, d_ok, d_ref, d_lvl, d_A,
                                                      NULL, ws, lane->workspace_bytes, lane->stream),
                   "svo_hip_find_match_direct_indirect");
    a.downloadRange(inputs_end, match_end, lane->stream);
  }
  // hybrid / mapped arena: the selection kernel stores `flag` when it starts, i.e. when the match kernels are through;
  // the host polls that, pose refinement follows on the same stream
  void launchPredictionSameStream(svo_hip::Speculation& sp) {
    const size_t cap = (size_t)Config::maxFts() + 1;
    int32_t* const d_M = out.d_header + 3;
    svo_hip::check(svo_hip_select_matches_indirect(&cam, (int)T_CAP, d_M, out.d_trial_cell, d_ok, out.d_trial_px, d_lvl,
                                                   out.d_trial_pos, Config::maxFts(), d_nsel, d_sel, d_sf, d_slvl, d_spos, d_has,
                                                   d_flag, 1, lane->stream),
                   "svo_hip_select_matches_indirect");
    svo_hip::check(svo_hip_pose_optimize_deferred(&cam, 1, d_nsel, (int)cap, d_sf, d_slvl, d_spos, d_has, sp.reproj_thresh,
                                                  sp.n_iter, d_T, d_Cov, d_stats, d_ran, lane->stream),
                   "svo_hip_pose_optimize_deferred");
    sp.stream = lane->stream;
    sp.in_flight = true;
  }
};

// ---- the chain behind SparseImgAlign::run (frame_chain.h) ---------------------------------------------------------------
// One per Reprojector, registered on the tracking lane of the thread that calls reprojectMap.  GridT: Reprojector::Grid
// (private: deduced).
struct ChainRegistry {
  std::mutex mut;
  std::map<const void*, hip_dropin::FrameChain*> all;  // Reprojector -> its chain
};
ChainRegistry& chains() {
  static ChainRegistry r;
  return r;
}

template <class GridT>
class MirrorChain : public hip_dropin::FrameChain {
 public:
  MirrorChain(Map& map, const GridT& grid, const Reprojector::Options& options, const int* align_max_iter)
      : map_(map), grid_(grid), options_(options), align_max_iter_(align_max_iter), state_(IDLE), hook_lane_(NULL), frames_(NULL), n_kf_(0), d_key_pos_(NULL), d_key_valid_(NULL), h_rank_(NULL),
        d_rank_host_(NULL), d_qt_(NULL),
        h_Tcomp_(NULL), d_Tcomp_(NULL), flag_k1_(NULL), d_flag_k1_(NULL), n_tab_(0), rebuilds_at_prepare_(0) {}

  svo_hip::Lane* hook_lane_;  // the lane the chain is registered on

  size_t outputBytesBound() const {
    // (MirrorBatch::allocOutputs at its largest: 8192 map entries, 64 frames, the trial capacity; + the chain's own words)
    const size_t T = mirrorTrialCap();
    return 8192 * 4 + 64 * 4 + 3 * MirrorBatch::V_CAP * 4 + T * (16 + 12 + 32) + ((size_t)Config::maxFts() + 1) * 8 + 64 * 256;
  }
  bool prepare(const FramePtr& ref, const FramePtr& cur, svo_hip::Device& dev, svo_hip::Lane& lane) {
    using namespace hip_dropin;
    state_ = IDLE;
    delete frames_;
    frames_ = NULL;
    if (!options_.find_match_direct || MapMirror::mode() == MapMirror::OFF || !cur->fts_.empty()) return false;
    if (lane.arena.mode() == svo_hip::Arena::MIRRORED) return false;  // (no host-visible signals)
    const size_t n_cells = grid_.cells.size();
    if (n_cells > (size_t)SVO_HIP_REPROJ_MAX_CELLS || options_.max_n_kfs > 16) return false;
    MapMirror& mm = mirrorOf(&map_);
    if (!mm.sync(map_)) return false;  // (the ordinary path will find the same and fall back)
    const size_t n_mirror = mm.frames().size();
    if (n_mirror + 2 > 64 || (size_t)dev.slots() < n_mirror + 4) return false;
    // The overlapping keyframes are ranked ON THE DEVICE, with the pose it forms from K1's result (svo_hip_frame_pose_compose):
    // what the host sends are the key points Map::getCloseKeyframes looks at, keyframes in map order = the first entries
    // of the mirror's frame table.  (Ranking them here with the prior pose -- the first version of the chain -- was right
    // on 78 % of the frames of a 600-frame sequence with ten keyframes in view: the order of two keyframes at nearly the
    // same distance flips with the 2 cm the pose moves.)
    n_kf_ = map_.keyframes_.size();
    if (n_kf_ > n_mirror) return false;
    key_pos_.assign(15 * n_kf_, 0.0);
    key_valid_.assign(5 * n_kf_, 0);
    {
      size_t i = 0;
      for (std::list<FramePtr>::const_iterator kf = map_.keyframes_.begin(); kf != map_.keyframes_.end(); ++kf, ++i) {
        if (mm.frames()[i] != kf->get() || (*kf)->key_pts_.size() != 5) return false;
        for (size_t k = 0; k < 5; ++k) {
          const Feature* ftr = (*kf)->key_pts_[k];
          if (ftr == NULL) continue;
          if (ftr->point == NULL) return false;  // (the reference would dereference it)
          key_valid_[5 * i + k] = 1;
          for (int c = 0; c < 3; ++c) key_pos_[3 * (5 * i + k) + c] = ftr->point->pos_[c];
        }
      }
    }
    b_.clear();
    b_.dev = &dev; b_.lane = &lane; b_.mm = &mm;
    b_.T_CAP = mirrorTrialCap();
    mm.ensureDevice(grid_.cell_order, b_.T_CAP);
    lane.arena_chain.reset();
    lane.arena_chain.reserve(b_.arenaBytes(n_mirror + 2) + 4096);
    frames_ = new FrameTable(dev, svo_hip::Device::LANE_TRACKING);
    FrameTable& frames = *frames_;
    for (size_t i = 0; i < n_mirror; ++i)
      if (frames.indexOf(mm.frames()[i]) != (int)i) throw std::logic_error("Reprojector: frame table out of order");
    b_.i_cur = frames.indexOf(cur.get());
    frames.indexOf(ref.get());  // (a keyframe of the mirror, or one more entry)
    n_tab_ = (size_t)frames.size();
    b_.rank_of.assign(n_tab_, -1);  // (written on the device)
    b_.cam = cameraOf(cur->cam_);
    b_.cell_size = grid_.cell_size; b_.n_cols = grid_.grid_n_cols; b_.n_rows = grid_.grid_n_rows;
    b_.n_cells = n_cells; b_.first_cell = 0; b_.max_cells = firstBatchCells();
    b_.align_max_iter = *align_max_iter_;
    b_.predict = svo_hip::Device::speculationEnabled();
    b_.frame_id = cur->id_;
    cur_ = cur;
    rebuilds_at_prepare_ = mm.stats.rebuilds;
    state_ = PREPARED;
    return true;
  }
  void allocInputs(svo_hip::Arena& a, const FramePtr& ref) {
    b_.allocInputs(a, *frames_);
    double* qt = a.alloc<double>(8, &d_qt_);  // the reference frame's pose as the host's product reads it
    hip_dropin::poseToQt(ref->T_f_w_, qt, qt + 4);
    qt[7] = 0.0;
    double* kp = a.alloc<double>(key_pos_.size() ? key_pos_.size() : 1, &d_key_pos_);
    uint8_t* kv = a.alloc<uint8_t>(key_valid_.size() ? key_valid_.size() : 1, &d_key_valid_);
    std::copy(key_pos_.begin(), key_pos_.end(), kp);
    std::copy(key_valid_.begin(), key_valid_.end(), kv);
  }
  void allocOutputs(svo_hip::Arena& a) {
    b_.allocOutputs(a, n_tab_, cur_, b_.lane->spec, true);
    h_Tcomp_ = a.alloc<double>(12, &d_Tcomp_);
    h_rank_ = a.alloc<int32_t>(n_tab_, &d_rank_host_);
    flag_k1_ = a.alloc<int32_t>(1, &d_flag_k1_);
    *flag_k1_ = 0;
  }
  void enqueue(const double* d_T_cur_ref) {
    svo_hip::check(svo_hip_frame_pose_compose(d_T_cur_ref, d_qt_, d_qt_ + 4, const_cast<double*>(b_.ft.d_T_f_w), b_.i_cur,
                                              b_.predict ? b_.d_T : NULL, d_Tcomp_, &b_.cam, (int)n_tab_, (int)n_kf_, d_key_pos_,
                                              d_key_valid_, (int)options_.max_n_kfs, b_.d_rank, d_rank_host_, d_flag_k1_, 1,
                                              b_.lane->stream),
                   "svo_hip_frame_pose_compose");
    b_.launchMatch(b_.lane->arena);
    svo_hip::Speculation& sp = b_.lane->spec;
    if (b_.predict) {
      b_.launchPredictionSameStream(sp);
    } else {
      sp.stream = b_.lane->stream;  // (beginCall() of the lane's next call drains the stream before it refills the arena)
      sp.in_flight = true;
    }
    state_ = IN_FLIGHT;
  }
  const volatile int32_t* k1Signal() const { return flag_k1_; }
  void abandon() { state_ = IDLE; cur_.reset(); }
  ~MirrorChain() { delete frames_; }

  // ---- the other end: Reprojector::reprojectMap asks for the batch of `frame` -----------------------------------------
  // true: the chain's batch is the frame's (verified), its match results have arrived; `ranked` / `rank_of` / n_tab as the
  // ordinary path would have built them.  false: take the ordinary path (whatever was enqueued is drained by its beginCall()).
  bool adopt(const FramePtr& frame, svo_hip::Device& dev, svo_hip::Lane& lane, MirrorBatch& out_batch,
             std::vector<std::pair<FramePtr, int> >& ranked, size_t& n_tab) {
    using namespace hip_dropin;
    if (state_ != IN_FLIGHT) return false;
    state_ = IDLE;
    FramePtr cur;
    cur.swap(cur_);
    std::lock_guard<std::mutex> guard(lane.mut);
    svo_hip::Speculation& sp = lane.spec;
    int why = 0;
    bool ok = b_.lane == &lane && b_.dev == &dev && cur.get() == frame.get() && sp.in_flight && sp.stream == lane.stream &&
              b_.frame_id == frame->id_ && *flag_k1_ == 1;
    if (ok) {  // the pose the device formed is the pose the host formed
      why = 1;
      double T[12];
      poseToRt(frame->T_f_w_, T);
      ok = std::memcmp(T, h_Tcomp_, sizeof(T)) == 0;
    }
    std::vector<std::pair<FramePtr, int> > ranked_now;
    std::vector<int32_t> rank_now(n_tab_, -1);
    if (ok) {  // the device ranked the overlapping keyframes as the host ranks them
      why = 2;
      std::list<KfDist> close_kfs;
      map_.getCloseKeyframes(frame, close_kfs);
      close_kfs.sort(closerKf);
      for (std::list<KfDist>::iterator kf = close_kfs.begin(); kf != close_kfs.end() && ranked_now.size() < options_.max_n_kfs; ++kf) {
        const int idx = b_.mm->frameIndex(kf->first.get());
        if (idx < 0 || (size_t)idx >= n_tab_) { ok = false; break; }
        rank_now[(size_t)idx] = (int32_t)ranked_now.size();
        ranked_now.push_back(std::make_pair(kf->first, idx));
      }
      for (size_t i = 0; ok && i < n_tab_; ++i) ok = h_rank_[i] == rank_now[i];
    }
    if (ok) {  // nothing has touched the map since the chain was built
      why = 3;
      MapMirror& mm = *b_.mm;
      ok = &mm == &mirrorOf(&map_) && mm.sync(map_) && !mm.pending() && mm.stats.rebuilds == rebuilds_at_prepare_;
    }
    if (ok) {
      why = 4;
      {
        svo_hip::StageTimer stage_timer(dev, lane, svo_hip::Device::STAGE_REPROJECT);
        stage_timer.device(0);
        if (b_.predict) svo_hip::spinUntil(b_.flag, 1, lane.stream);
        else dev.finish(lane);
        stage_timer.unmarshal();
      }
      ok = b_.header[0] == 0;
    }
    dev.countChain(ok, why);
    if (!ok) return false;  // (sp.in_flight stays set: the ordinary path's beginCall() waits for the stream)
    if (b_.predict) std::copy(h_Tcomp_, h_Tcomp_ + 12, sp.T_init);
    else sp.in_flight = false;  // (the stream has been waited for)
    out_batch = b_;
    out_batch.rank_of = rank_now;
    ranked = ranked_now;
    n_tab = n_tab_;
    return true;
  }

  static size_t mirrorTrialCap() {
    static const long t_cap_override = [] { const char* v = std::getenv("SVO_HIP_MIRROR_TRIALS"); return v ? std::atol(v) : 0L; }();
    return t_cap_override > 0 ? (size_t)t_cap_override : 4096;
  }
  static size_t firstBatchCells() {
    static const long first_batch_override = [] { const char* v = std::getenv("SVO_HIP_FIRST_BATCH_CELLS"); return v ? std::atol(v) : 0L; }();
    return first_batch_override > 0 ? (size_t)first_batch_override : (size_t)Config::maxFts() + 1 + ((size_t)Config::maxFts() + 1) / 3 + 8;
  }

 private:
  enum State { IDLE = 0, PREPARED = 1, IN_FLIGHT = 2 };
  Map& map_;
  const GridT& grid_;
  const Reprojector::Options& options_;
  const int* align_max_iter_;
  State state_;
  MirrorBatch b_;
  hip_dropin::FrameTable* frames_;
  FramePtr cur_;
  size_t n_kf_;
  std::vector<double> key_pos_;
  std::vector<uint8_t> key_valid_;
  double* d_key_pos_;
  uint8_t* d_key_valid_;
  const int32_t* h_rank_;
  int32_t* d_rank_host_;
  double* d_qt_;
  double *h_Tcomp_, *d_Tcomp_;
  volatile int32_t* flag_k1_;
  int32_t* d_flag_k1_;
  size_t n_tab_;
  uint64_t rebuilds_at_prepare_;
};

// Steps 1-4 of reprojectMap with the walk of the pointer graph replaced by one kernel over the mirror: the host finds
// the overlapping keyframes (the reference's own getCloseKeyframes: a dozen keyframes, five key points each), sends what
// changed in the map since the last frame, and enqueues  reproject_map -> match kernels -> selection -> pose refinement
// back to back; it then replays the candidate bookkeeping (:108-123) and the cell loop (:131-139, 151-200) from the
// result tables.  Returns false -- having changed nothing -- when the frame has to take the list-walking path below.
// `chain`: the Reprojector's MirrorChain, or NULL; when it holds this frame's batch (enqueued behind K1 and verified
// here) the first batch is taken from there.
template <class GridT>
bool reprojectMapMirrored(const FramePtr& frame, std::vector<std::pair<FramePtr, std::size_t> >& overlap_kfs, Map& map_,
                          const GridT& grid_, const Reprojector::Options& options_, int align_max_iter, size_t& n_matches_,
                          size_t& n_trials_, MirrorChain<GridT>* chain) {
  using namespace hip_dropin;
  const size_t n_cells = grid_.cells.size();
  // trials / visits one batch can hold (status 1 beyond: the frame takes the list-walking path; SVO_HIP_MIRROR_TRIALS
  // overrides the trial capacity: the tests use it to force that hand-over)
  const size_t T_CAP = MirrorChain<GridT>::mirrorTrialCap();
  if (n_cells > (size_t)SVO_HIP_REPROJ_MAX_CELLS || options_.max_n_kfs > 16) return false;
  MapMirror& mm = mirrorOf(&map_);
  ++mm.stats.calls;
  svo_hip::Device& dev = ensureDevice(*frame);
  const int L = svo_hip::Device::LANE_TRACKING;
  svo_hip::Lane& lane = dev.lane(L);

  MirrorBatch batch;  // the batch in flight
  std::vector<std::pair<FramePtr, int> > ranked;  // (keyframe, its index in the frame table), closest first
  size_t n_tab = 0;
  std::vector<int32_t> rank_of;
  const bool adopted = chain != NULL && chain->adopt(frame, dev, lane, batch, ranked, n_tab);
  if (adopted) {
    rank_of = batch.rank_of;
  } else {
    std::list<KfDist> close_kfs;
    map_.getCloseKeyframes(frame, close_kfs);
    close_kfs.sort(closerKf);
    if (!mm.sync(map_)) { ++mm.stats.fallbacks; mm.invalidate(); return false; }
    n_tab = mm.frames().size() + 1;  // the mirror's frames, then the current one
    if ((size_t)dev.slots() < n_tab + 2) { ++mm.stats.fallbacks; mm.invalidate(); return false; }  // every keyframe resident at once
    rank_of.assign(n_tab, -1);
    for (std::list<KfDist>::iterator kf = close_kfs.begin(); kf != close_kfs.end() && ranked.size() < options_.max_n_kfs; ++kf) {
      const int idx = mm.frameIndex(kf->first.get());
      if (idx < 0) { ++mm.stats.fallbacks; mm.invalidate(); return false; }  // (a keyframe of the map the mirror does not know: cannot happen after sync)
      rank_of[(size_t)idx] = (int32_t)ranked.size();
      ranked.push_back(std::make_pair(kf->first, idx));
    }
  }
  bool predict = adopted && batch.predict;

  // one batch: cells [first_cell, ...) until max_cells of them hold a trial.  false: capacity exceeded on the device
  auto runBatch = [&](const size_t first_cell, const size_t max_cells) -> bool {
    std::lock_guard<std::mutex> guard(lane.mut);
    dev.beginCall(L);
    svo_hip::StageTimer stage_timer(dev, lane, svo_hip::Device::STAGE_REPROJECT);
    svo_hip::Arena& a = lane.arena;
    a.reset();
    MirrorBatch& b = batch;
    b.clear();
    b.dev = &dev; b.lane = &lane; b.mm = &mm;
    b.T_CAP = T_CAP;
    mm.ensureDevice(grid_.cell_order, T_CAP);
    const size_t n_tab_b = mm.frames().size() + 1;
    a.reserve(b.arenaBytes(n_tab_b));
    FrameTable frames(dev, L);
    for (size_t i = 0; i + 1 < n_tab_b; ++i)
      if (frames.indexOf(mm.frames()[i]) != (int)i) throw std::logic_error("Reprojector: frame table out of order");
    b.i_cur = frames.indexOf(frame.get());
    b.rank_of.assign(rank_of.begin(), rank_of.begin() + std::min(rank_of.size(), n_tab_b));
    b.rank_of.resize(n_tab_b, -1);
    b.cam = cameraOf(frame->cam_);
    b.cell_size = grid_.cell_size; b.n_cols = grid_.grid_n_cols; b.n_rows = grid_.grid_n_rows;
    b.n_cells = n_cells; b.first_cell = first_cell; b.max_cells = max_cells;
    b.align_max_iter = align_max_iter;
    b.frame_id = frame->id_;
    predict = b.predict = first_cell == 0 && svo_hip::Device::speculationEnabled() && frame->fts_.empty();
    svo_hip::Speculation& sp = lane.spec;
    b.allocInputs(a, frames);
    a.endInputs();
    b.allocOutputs(a, n_tab_b, frame, sp, false);
    stage_timer.device(a.used());
    a.uploadAll(lane.stream);
    b.launchMatch(a);
    if (predict && b.flag != NULL) {
      b.launchPredictionSameStream(sp);
      svo_hip::spinUntil(b.flag, 1, lane.stream);
    } else if (predict) {
      // mirrored arena: behind the match results on the lane's second stream, so that the wait below ends with the copy
      // of the match results
      const size_t cap = (size_t)Config::maxFts() + 1;
      int32_t* const d_M = b.out.d_header + 3;
      void* const next = lane.stream_next;
      svo_hip::check(svo_hip_event_record(lane.ev_results, lane.stream), "svo_hip_event_record");
      svo_hip::check(svo_hip_stream_wait_event(next, lane.ev_results), "svo_hip_stream_wait_event");
      svo_hip::check(svo_hip_select_matches_indirect(&b.cam, (int)T_CAP, d_M, b.out.d_trial_cell, b.d_ok, b.out.d_trial_px, b.d_lvl,
                                                     b.out.d_trial_pos, Config::maxFts(), b.d_nsel, b.d_sel, b.d_sf, b.d_slvl, b.d_spos,
                                                     b.d_has, NULL, 0, next),
                     "svo_hip_select_matches_indirect");
      svo_hip::check(svo_hip_pose_optimize_deferred(&b.cam, 1, b.d_nsel, (int)cap, b.d_sf, b.d_slvl, b.d_spos, b.d_has, sp.reproj_thresh,
                                                    sp.n_iter, b.d_T, b.d_Cov, b.d_stats, b.d_ran, next),
                     "svo_hip_pose_optimize_deferred");
      a.downloadRange(b.results_begin, a.used(), next);
      sp.stream = next;
      sp.in_flight = true;
      svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
    } else {
      svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
    }
    stage_timer.unmarshal();
    if (b.header[0] != 0) {  // a capacity was exceeded (or, -1, the kernel never ran): drop what was enqueued behind it
      if (predict) {
        sp.in_flight = false;
        svo_hip::check(svo_hip_stream_sync(sp.stream), "svo_hip_stream_sync");
        predict = false;
      }
      return false;
    }
    return true;
  };

  // The first batch takes the cells a success rate of 3 in 4 would need (see the list-walking path)
  if (!adopted && !runBatch(0, MirrorChain<GridT>::firstBatchCells())) { ++mm.stats.fallbacks; mm.invalidate(); return false; }
  // the tables of the batch in flight (host addresses of arena blocks)
  size_t view_V = (size_t)batch.header[2], view_end_cell = (size_t)batch.header[4];
  const int32_t* point_cell = batch.h_point_cell;
  const int32_t* kf_count = batch.h_kf_count;

  // ---- 1. overlap_kfs: (keyframe, points of it that fell inside the frame), closest first (:82-102)
  overlap_kfs.reserve(options_.max_n_kfs);
  for (size_t k = 0; k < ranked.size(); ++k)
    overlap_kfs.push_back(std::pair<FramePtr, size_t>(ranked[k].first, (size_t)kf_count[ranked[k].second]));

  // ---- 2. candidates that do not reproject lose credit (:108-123), in list order
  {
    boost::unique_lock<boost::mutex> lock(map_.point_candidates_.mut_);
    MapPointCandidates::PointCandidateList& cl = map_.point_candidates_.candidates_;
    const std::vector<int32_t> cands = mm.candidateEntries();  // (a copy: markDead edits the list)
    for (size_t k = 0; k < cands.size(); ++k) {
      const int32_t e = cands[k];
      if (point_cell[e] >= 0) continue;
      const MapMirror::Entry& x = mm.entries()[(size_t)e];
      x.pt->n_failed_reproj_ += 3;
      if (x.pt->n_failed_reproj_ > 30) {
        map_.point_candidates_.deleteCandidate(*x.cand);
        cl.erase(x.cand);
        mm.markDead(e);
      }
    }
  }

  // ---- 4. per cell, in the shuffled order: the best-quality point that matched (:131-139, 151-200)
  svo_hip::Speculation& sp = lane.spec;
  std::vector<int32_t> selected;
  std::vector<const void*> pred_point;  // what the device's selection must have picked, for the optimizer's drop-in to check
  std::vector<double> pred_px;
  std::vector<int32_t> pred_level, pred_trial;
  size_t v = 0;
  for (size_t i = 0; i < n_cells; ++i) {
    if (i == view_end_cell) {
      // the cells of the batch are used up and the loop has not stopped: the rest (the prediction, made on the first
      // batch's trials only, is dropped: beginCall() drains what was enqueued)
      predict = false;
      ++mm.stats.second_batches;
      if (!runBatch(i, n_cells)) throw svo_hip::Error("Reprojector: the second match batch exceeds the mirror's capacity");
      view_V = (size_t)batch.header[2]; view_end_cell = (size_t)batch.header[4];
      v = 0;
    }
    bool matched = false;
    for (; v < view_V && (size_t)batch.h_vc[v] == i; ++v) {
      if (matched) continue;  // reprojectCell returns at the first success: the rest of the cell is not looked at
      ++n_trials_;
      const int32_t e = batch.h_vp[v];
      Point* pt = mm.entries()[(size_t)e].pt;
      const int m = batch.h_vt[v];
      if (!(m >= 0 && batch.h_ok[m] != 0)) {
        pt->n_failed_reproj_++;
        if (pt->type_ == Point::TYPE_UNKNOWN && pt->n_failed_reproj_ > 15) { map_.safeDeletePoint(pt); mm.markDead(e); }
        if (pt->type_ == Point::TYPE_CANDIDATE && pt->n_failed_reproj_ > 30) { map_.point_candidates_.deleteCandidatePoint(pt); mm.markDead(e); }
        continue;
      }
      pt->n_succeeded_reproj_++;
      if (pt->type_ == Point::TYPE_UNKNOWN && pt->n_succeeded_reproj_ > 10) { pt->type_ = Point::TYPE_GOOD; mm.markType(e, 3); }
      const Vector2d px(batch.h_px[2 * m], batch.h_px[2 * m + 1]);
      Feature* new_feature = new Feature(frame.get(), px, batch.h_lvl[m]);
      frame->addFeature(new_feature);
      new_feature->point = pt;
      const Feature* ref_ftr = batch.h_ref[m] >= 0 ? mm.obsFeature(batch.h_ref[m]) : NULL;
      if (ref_ftr != NULL && ref_ftr->type == Feature::EDGELET) {
        new_feature->type = Feature::EDGELET;
        Matrix2d A_cur_ref;
        A_cur_ref(0, 0) = batch.h_A[4 * m]; A_cur_ref(0, 1) = batch.h_A[4 * m + 1];
        A_cur_ref(1, 0) = batch.h_A[4 * m + 2]; A_cur_ref(1, 1) = batch.h_A[4 * m + 3];
        new_feature->grad = A_cur_ref * ref_ftr->grad;
        new_feature->grad.normalize();
      }
      if (predict) {
        pred_point.push_back(pt);
        pred_px.push_back(px[0]); pred_px.push_back(px[1]);
        pred_level.push_back(batch.h_lvl[m]);
        pred_trial.push_back(m);
      }
      selected.push_back(e);
      matched = true;
    }
    if (matched) ++n_matches_;
    if (n_matches_ > (size_t)Config::maxFts()) break;
  }
  if (predict) {  // published under the lane's mutex, in one piece (the lane's next call reads it under the same mutex)
    std::lock_guard<std::mutex> guard(lane.mut);
    sp.point.swap(pred_point); sp.px.swap(pred_px); sp.level.swap(pred_level); sp.trial.swap(pred_trial);
    sp.valid = !sp.point.empty();
  }
  mm.watch(selected);  // FrameHandlerBase::optimizeStructure may move these before the next frame
  return true;
}
}  // namespace

namespace hip_dropin {
// calls, rebuilds, fallbacks to the list-walking path, point records sent, observation records sent, second batches --
// summed over the mirrors of the process (read-outs of the tests and the benchmark; not synchronised with running calls)
void mapMirrorStats(uint64_t out[6]) {
  for (int i = 0; i < 6; ++i) out[i] = 0;
  MirrorRegistry& r = mirrors();
  std::lock_guard<std::mutex> g(r.mut);
  for (std::map<const Map*, MapMirror*>::const_iterator it = r.all.begin(); it != r.all.end(); ++it) {
    const MapMirror::Stats& s = it->second->stats;
    out[0] += s.calls; out[1] += s.rebuilds; out[2] += s.fallbacks; out[3] += s.patched_points; out[4] += s.patched_obs;
    out[5] += s.second_batches;
  }
}
}  // namespace hip_dropin

// The reference's destructor (reprojector.cpp:39-42) plus the release of the map's device-resident shadow and of the
// chain this Reprojector registered on its tracking lane (frame_chain.h).
Reprojector::~Reprojector() {
  std::for_each(grid_.cells.begin(), grid_.cells.end(), [&](Cell* c) { delete c; });
  {
    typedef MirrorChain<Grid> Chain;
    ChainRegistry& r = chains();
    Chain* ch = NULL;
    {
      std::lock_guard<std::mutex> g(r.mut);
      std::map<const void*, hip_dropin::FrameChain*>::iterator it = r.all.find(this);
      if (it != r.all.end()) {
        ch = static_cast<Chain*>(it->second);
        r.all.erase(it);
      }
    }
    if (ch != NULL) {
      if (ch->hook_lane_ != NULL) {
        std::lock_guard<std::mutex> g(ch->hook_lane_->mut);
        if (ch->hook_lane_->chain_hook == static_cast<hip_dropin::FrameChain*>(ch)) ch->hook_lane_->chain_hook = NULL;
      }
      delete ch;
    }
  }
  releaseMirror(&map_);
}

void Reprojector::reprojectMap(FramePtr frame, std::vector<std::pair<FramePtr, std::size_t> >& overlap_kfs) {
  // deferred mapping: the depth filter's update of the previous frame hands its converged seeds to the map before
  // the map is read (no-op otherwise)
  svo_hip::Device::joinDeferredAll();
  resetGrid();
  // the device-resident mirror of the map (row N2): SVO_HIP_MAP_MIRROR=on (default) | verify | off
  if (options_.find_match_direct && hip_dropin::MapMirror::mode() != hip_dropin::MapMirror::OFF) {
    SVO_START_TIMER("feature_align");
    bool done = false;
    // this Reprojector's chain (frame_chain.h): created on first use, registered on the calling thread's tracking lane so
    // that the NEXT frame's SparseImgAlign::run can enqueue this call's device work behind its own
    typedef MirrorChain<Grid> Chain;
    Chain* chain = NULL;
    if (svo_hip::Device::chainEnabled()) {
      ChainRegistry& r = chains();
      std::lock_guard<std::mutex> g(r.mut);
      hip_dropin::FrameChain*& slot = r.all[this];
      if (slot == NULL) slot = new Chain(map_, grid_, options_, &matcher_.options_.align_max_iter);
      chain = static_cast<Chain*>(slot);
    }
    try {
      done = reprojectMapMirrored(frame, overlap_kfs, map_, grid_, options_, matcher_.options_.align_max_iter, n_matches_, n_trials_,
                                  chain);
    } catch (...) {
      // an error between the patch (which marks the shadow's records as sent) and a completed launch leaves the device
      // copy behind the shadow: the next call starts from a fresh walk of the map
      mirrorOf(&map_).invalidate();
      throw;
    }
    SVO_STOP_TIMER("feature_align");
    if (chain != NULL) {  // (the lane of THIS thread: the one the next frame's sparse alignment will run on)
      svo_hip::Lane& lane = hip_dropin::ensureDevice(*frame).lane(svo_hip::Device::LANE_TRACKING);
      std::lock_guard<std::mutex> g(lane.mut);
      if (done) {
        lane.chain_hook = static_cast<hip_dropin::FrameChain*>(chain);
        chain->hook_lane_ = &lane;
      } else if (lane.chain_hook == static_cast<hip_dropin::FrameChain*>(chain)) {
        lane.chain_hook = NULL;  // the map does not fit the mirror (any more): no chain until it does
      }
    }
    if (done) return;
  } else if (hip_dropin::MapMirror::mode() != hip_dropin::MapMirror::OFF) {
    mirrorOf(&map_).invalidate();  // this call changes the map without the mirror looking
  }

  // ---- 1. keyframes sharing the field of view, closest first; bin their points --------------
  SVO_START_TIMER("reproject_kfs");
  std::list<KfDist> close_kfs;
  map_.getCloseKeyframes(frame, close_kfs);
  close_kfs.sort(closerKf);
  overlap_kfs.reserve(options_.max_n_kfs);
  size_t n_kfs = 0;
  for (std::list<KfDist>::iterator kf = close_kfs.begin(); kf != close_kfs.end() && n_kfs < options_.max_n_kfs; ++kf, ++n_kfs) {
    FramePtr ref_frame = kf->first;
    overlap_kfs.push_back(std::pair<FramePtr, size_t>(ref_frame, 0));
    for (Features::iterator f = ref_frame->fts_.begin(); f != ref_frame->fts_.end(); ++f) {
      Point* pt = (*f)->point;
      if (pt == NULL || pt->last_projected_kf_id_ == frame->id_) continue;  // each point once per frame
      pt->last_projected_kf_id_ = frame->id_;
      if (reprojectPoint(frame, pt)) overlap_kfs.back().second++;
    }
  }
  SVO_STOP_TIMER("reproject_kfs");

  // ---- 2. the not yet converged candidates of the depth filter ------------------------------
  SVO_START_TIMER("reproject_candidates");
  {
    boost::unique_lock<boost::mutex> lock(map_.point_candidates_.mut_);
    MapPointCandidates::PointCandidateList& cl = map_.point_candidates_.candidates_;
    for (MapPointCandidates::PointCandidateList::iterator c = cl.begin(); c != cl.end();) {
      if (!reprojectPoint(frame, c->first)) {
        c->first->n_failed_reproj_ += 3;
        if (c->first->n_failed_reproj_ > 30) {
          map_.point_candidates_.deleteCandidate(*c);
          c = cl.erase(c);
          continue;
        }
      }
      ++c;
    }
  }
  SVO_STOP_TIMER("reproject_candidates");

  // ---- 3. device: one findMatchDirect trial per binned candidate -----------------------------
  SVO_START_TIMER("feature_align");
  // What step 4 will find when it walks the cells: per visited candidate (cells in grid_.cell_order, each list in
  // its sorted order, deleted points left out) the index of its trial in the device batch, or -1 when findMatchDirect
  // fails before it reaches the image (no close view, matcher.cpp:137-138); and the trials' results.
  const size_t n_cells = grid_.cells.size();
  std::vector<int32_t> visit;
  std::vector<size_t> visit_begin(n_cells + 1, 0);
  std::vector<double> R_px, R_A;                 // [trial][2], [trial][4]
  std::vector<int32_t> R_ok, R_lvl;              // [trial]
  std::vector<Feature*> R_ref;                   // [trial] Matcher::ref_ftr_
  std::vector<const void*> pred_point;  // the prediction's features as the host selects them (published at the end)
  std::vector<double> pred_px;
  std::vector<int32_t> pred_level, pred_trial;
  bool predict = false;             // pose refinement of this frame has been enqueued behind the match kernels
  svo_hip::Lane* spec_lane = NULL;  // ... on this lane
  size_t enumerated_end = 0;        // cells [0, enumerated_end) of the visiting order have their trials
  // The visiting loop stops once more than maxFts cells have matched (:137-138): of ~300 cells with candidates it
  // typically sees the first ~125, and what lies behind the stop is never looked at (nor are its counters touched).
  // The first batch therefore takes the cells a success rate of 3 in 4 would need; should step 4 run out of them before
  // it stops (it then has matched less than three quarters of the cells), a second batch takes the rest.
  // (SVO_HIP_FIRST_BATCH_CELLS overrides the size of the first batch: the tests use it to force the second one)
  static const long first_batch_override = [] { const char* v = std::getenv("SVO_HIP_FIRST_BATCH_CELLS"); return v ? std::atol(v) : 0L; }();
  const size_t first_batch_cells = first_batch_override > 0 ? (size_t)first_batch_override
                                                            : (size_t)Config::maxFts() + 1 + ((size_t)Config::maxFts() + 1) / 3 + 8;
  // [first_cell, return value): the cells whose candidates were listed and matched
  auto runBatch = [&](const size_t first_cell, const size_t max_cells_with_trials) -> size_t {
    using namespace hip_dropin;
    // The reference observation of a trial (Point::getCloseViewObs, matcher.cpp:137) is chosen HERE,
    // by the reference's own host code: a trial then ships exactly one svo::Feature, and only the
    // keyframes that actually serve as reference need their pyramid on the device (a point seen from
    // 40 keyframes no longer pins 40 pool slots for one 10x10 template).
    std::vector<Candidate*> trials;
    std::vector<Feature*> trial_ref;
    std::vector<int32_t> trial_cell;
    const Vector3d cur_pos(frame->pos());
    FramePositions positions;
    const size_t base = R_ok.size();  // trials of earlier batches keep their indices
    // Trials are listed in the order step 4 visits them: cells in grid_.cell_order, each cell's list sorted first
    // (reprojectCell, :152: good points before unknown ones before candidates; stable, like std::list::sort).  A point
    // lies in one cell only, so sorting a cell before the batch gives the list the visiting loop would produce.
    size_t i = first_cell, cells_with_trials = 0;
    for (; i < n_cells && cells_with_trials < max_cells_with_trials; ++i) {
      Cell& cell = *grid_.cells.at(grid_.cell_order[i]);
      sortCell(cell);
      visit_begin[i] = visit.size();
      const size_t before = trials.size();
      for (Cell::iterator c = cell.begin(); c != cell.end(); ++c) {
        if (c->pt->type_ == Point::TYPE_DELETED) continue;
        Feature* ref_ftr = NULL;
        if (!closeViewObs(*c->pt, cur_pos, positions, ref_ftr)) {  // findMatchDirect returns false at once (:137-138)
          visit.push_back(-1);
          continue;
        }
        visit.push_back((int32_t)(base + trials.size()));
        trials.push_back(&*c);
        trial_ref.push_back(ref_ftr);
        trial_cell.push_back((int32_t)i);
      }
      if (trials.size() > before) ++cells_with_trials;
    }
    const size_t end_cell = i;
    for (size_t k = end_cell; k <= n_cells; ++k) visit_begin[k] = visit.size();
    const size_t M = trials.size();
    const size_t n_obs = M;
    if (M > 0) {
      svo_hip::Device& dev = ensureDevice(*frame);
      const int L = svo_hip::Device::LANE_TRACKING;
      svo_hip::Lane& lane = dev.lane(L);
      std::lock_guard<std::mutex> guard(lane.mut);
      dev.beginCall(L);
      svo_hip::StageTimer stage_timer(dev, lane, svo_hip::Device::STAGE_REPROJECT);
      svo_hip::Arena& a = lane.arena;
      a.reset();
      a.reserve(((size_t)1 << 17) + M * 768 + n_obs * 128 + 4096 * 32);
      FrameTable frames(dev, L);
      const int i_cur = frames.indexOf(frame.get());
      // the frame gets its features here and nowhere else: what the pose optimizer will be handed is known
      predict = first_cell == 0 && svo_hip::Device::speculationEnabled() && frame->fts_.empty();
      const size_t cap = std::min(M, (size_t)Config::maxFts() + 1);

      int32_t *d_cur, *d_ptr, *d_cell; double* d_pos;
      int32_t* cur = a.alloc<int32_t>(M, &d_cur);
      double* pos = a.alloc<double>(3 * M, &d_pos);
      int32_t* ptr = a.alloc<int32_t>(M + 1, &d_ptr);
      int32_t* cellv = a.alloc<int32_t>(M, &d_cell);
      std::vector<double> px_in(2 * M);  // goes into the in/out block below
      FeatureColumns obs;
      obs.alloc(a, n_obs);
      std::vector<Feature*> obs_ftr(n_obs);
      size_t o = 0;
      for (size_t m = 0; m < M; ++m) {
        const Point* pt = trials[m]->pt;
        cur[m] = i_cur;
        cellv[m] = trial_cell[m];
        for (int k = 0; k < 3; ++k) pos[3 * m + k] = pt->pos_[k];
        px_in[2 * m] = trials[m]->px[0]; px_in[2 * m + 1] = trials[m]->px[1];
        ptr[m] = (int32_t)o;
        obs.set(o, frames.indexOf(trial_ref[m]->frame), trial_ref[m]);
        obs_ftr[o] = trial_ref[m];
        ++o;
      }
      ptr[M] = (int32_t)o;
      svo_hip_frames ft;
      frames.emit(a, &ft);
      // the predicted pose refinement's observations: gathered on the device, never read by the host
      double *d_sf = NULL, *d_spos = NULL;
      int32_t* d_slvl = NULL;
      if (predict) {
        a.alloc<double>(3 * cap, &d_sf);
        a.alloc<double>(3 * cap, &d_spos);
        a.alloc<int32_t>(cap, &d_slvl);
      }
      a.endInputs();
      const size_t inputs_end = a.used();

      double *d_px, *d_A; int32_t *d_ok, *d_ref, *d_lvl;
      double* px = a.alloc<double>(2 * M, &d_px);  // in: projection, out: refined pixel (uploadAll + download)
      std::copy(px_in.begin(), px_in.end(), px);
      int32_t* ok = a.alloc<int32_t>(M, &d_ok);
      int32_t* ref = a.alloc<int32_t>(M, &d_ref);
      int32_t* lvl = a.alloc<int32_t>(M, &d_lvl);
      double* A = a.alloc<double>(4 * M, &d_A);
      const size_t match_end = a.used();

      double *d_T = NULL, *d_Cov = NULL, *d_stats = NULL;
      int32_t *d_nsel = NULL, *d_sel = NULL, *d_ran = NULL, *d_flag = NULL;
      volatile int32_t* flag = NULL;
      uint8_t* d_has = NULL;
      size_t results_begin = match_end;
      svo_hip::Speculation& sp = lane.spec;
      if (predict) {
        spec_lane = &lane;
        sp.frame_id = frame->id_;
        sp.point.clear(); sp.px.clear(); sp.level.clear(); sp.trial.clear();
        results_begin = a.used();
        // what comes back: the pose goes in and out like in the optimizer's own call
        double* T = a.alloc<double>(12, &d_T);
        poseToRt(frame->T_f_w_, T);
        std::copy(T, T + 12, sp.T_init);
        sp.T = T;
        sp.n_sel = a.alloc<int32_t>(1, &d_nsel);
        sp.sel = a.alloc<int32_t>(cap, &d_sel);
        sp.has_point = a.alloc<uint8_t>(cap, &d_has);
        sp.Cov = a.alloc<double>(36, &d_Cov);
        sp.stats = a.alloc<double>(4, &d_stats);
        sp.ran = a.alloc<int32_t>(1, &d_ran);
        if (a.mode() != svo_hip::Arena::MIRRORED) flag = a.alloc<int32_t>(1, &d_flag);
        sp.reproj_thresh = Config::poseOptimThresh();
        sp.n_iter = (int)Config::poseOptimNumIter();
      }

      const svo_hip_camera cam = cameraOf(frame->cam_);
      void* ws = dev.workspace(lane, (int)M);
      stage_timer.device(a.used());
      a.uploadAll(lane.stream);
      svo_hip::check(svo_hip_find_match_direct(&dev.layout(), dev.store(), &cam, &ft, (int)M, d_cur, d_pos, d_ptr, &obs.dev,
                                               Config::nPyrLevels(), matcher_.options_.align_max_iter, d_px, d_ok, d_ref, d_lvl,
                                               d_A, NULL, ws, lane.workspace_bytes, lane.stream),
                     "svo_hip_find_match_direct");
      a.downloadRange(inputs_end, match_end, lane.stream);
      if (predict && flag != NULL) {
        // Results land in host memory as the kernels write them (hybrid / mapped arena): the selection kernel stores
        // `flag` when it starts, i.e. when the match kernels are through, and the host polls that instead of waiting
        // for the stream -- pose refinement follows on the same stream, no event, no second queue.
        *flag = 0;
        svo_hip::check(svo_hip_select_matches(&cam, (int)M, d_cell, d_ok, d_px, d_lvl, d_pos, Config::maxFts(), d_nsel, d_sel, d_sf,
                                              d_slvl, d_spos, d_has, d_flag, 1, lane.stream),
                       "svo_hip_select_matches");
        // (the wave kernel alone: a frame it hands over, ran == 2, is finished by the optimizer's drop-in)
        svo_hip::check(svo_hip_pose_optimize_deferred(&cam, 1, d_nsel, (int)cap, d_sf, d_slvl, d_spos, d_has, sp.reproj_thresh,
                                                      sp.n_iter, d_T, d_Cov, d_stats, d_ran, lane.stream),
                       "svo_hip_pose_optimize_deferred");
        sp.stream = lane.stream;
        sp.in_flight = true;  // beginCall() of the lane's next call (or the optimizer's drop-in) waits for it
        svo_hip::spinUntil(flag, 1, lane.stream);
      } else if (predict) {
        // mirrored arena: behind the match results on the lane's second stream, so that the wait below ends with the
        // copy of the match results
        void* const next = lane.stream_next;
        svo_hip::check(svo_hip_event_record(lane.ev_results, lane.stream), "svo_hip_event_record");
        svo_hip::check(svo_hip_stream_wait_event(next, lane.ev_results), "svo_hip_stream_wait_event");
        svo_hip::check(svo_hip_select_matches(&cam, (int)M, d_cell, d_ok, d_px, d_lvl, d_pos, Config::maxFts(), d_nsel, d_sel, d_sf,
                                              d_slvl, d_spos, d_has, NULL, 0, next),
                       "svo_hip_select_matches");
        svo_hip::check(svo_hip_pose_optimize_deferred(&cam, 1, d_nsel, (int)cap, d_sf, d_slvl, d_spos, d_has, sp.reproj_thresh,
                                                      sp.n_iter, d_T, d_Cov, d_stats, d_ran, next),
                       "svo_hip_pose_optimize_deferred");
        a.downloadRange(results_begin, a.used(), next);
        sp.stream = next;
        sp.in_flight = true;
        svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
      } else {
        svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
      }
      stage_timer.unmarshal();

      // out of the arena: a later batch (or anything else on this lane) may reuse it
      R_px.insert(R_px.end(), px, px + 2 * M);
      R_A.insert(R_A.end(), A, A + 4 * M);
      R_ok.insert(R_ok.end(), ok, ok + M);
      R_lvl.insert(R_lvl.end(), lvl, lvl + M);
      for (size_t m = 0; m < M; ++m) R_ref.push_back(ref[m] >= 0 ? obs_ftr[ref[m]] : NULL);
    }
    return end_cell;
  };
  if (options_.find_match_direct) enumerated_end = runBatch(0, first_batch_cells);

  // ---- 4. per cell, in the shuffled order: the best-quality point that matched ---------------
  for (size_t i = 0; i < grid_.cells.size(); ++i) {
    Cell& cell = *grid_.cells.at(grid_.cell_order[i]);
    if (!options_.find_match_direct) sortCell(cell);  // (sorted above otherwise)
    if (options_.find_match_direct && i == enumerated_end) {
      // the cells of the first batch are used up and the loop has not stopped: the rest.  If that brings new trials the
      // prediction goes (the device selected among the first batch's trials only): runBatch() clears `predict` and its
      // beginCall() drains what was enqueued
      enumerated_end = runBatch(i, n_cells);
    }
    bool matched = false;
    size_t v = visit_begin[i];  // the candidates step 3 listed for this cell, in this order
    for (Cell::iterator it = cell.begin(); it != cell.end() && !matched;) {
      ++n_trials_;
      Point* pt = it->pt;
      if (pt->type_ == Point::TYPE_DELETED) { it = cell.erase(it); continue; }
      struct { int trial; bool ok; Vector2d px; int search_level; Feature* ref_ftr; } r;
      if (options_.find_match_direct) {
        if (v >= visit_begin[i + 1]) throw std::logic_error("Reprojector: candidate without a device trial");
        r.trial = visit[v++];
        const int m = r.trial;
        r.ok = m >= 0 && R_ok[m] != 0;
        r.px = m >= 0 ? Vector2d(R_px[2 * m], R_px[2 * m + 1]) : it->px;
        r.search_level = m >= 0 ? R_lvl[m] : 0;
        r.ref_ftr = m >= 0 ? R_ref[m] : NULL;
      } else {  // accept the projection as it is
        r.trial = -1; r.ok = true; r.px = it->px; r.search_level = 0; r.ref_ftr = NULL;
      }
      if (!r.ok) {
        pt->n_failed_reproj_++;
        if (pt->type_ == Point::TYPE_UNKNOWN && pt->n_failed_reproj_ > 15) map_.safeDeletePoint(pt);
        if (pt->type_ == Point::TYPE_CANDIDATE && pt->n_failed_reproj_ > 30) map_.point_candidates_.deleteCandidatePoint(pt);
        it = cell.erase(it);
        continue;
      }
      pt->n_succeeded_reproj_++;
      if (pt->type_ == Point::TYPE_UNKNOWN && pt->n_succeeded_reproj_ > 10) pt->type_ = Point::TYPE_GOOD;

      Feature* new_feature = new Feature(frame.get(), r.px, r.search_level);
      frame->addFeature(new_feature);
      new_feature->point = pt;  // the point learns about this observation only if the frame becomes a keyframe
      if (r.ref_ftr != NULL && r.ref_ftr->type == Feature::EDGELET) {
        new_feature->type = Feature::EDGELET;
        Matrix2d A_cur_ref;
        A_cur_ref(0, 0) = R_A[4 * r.trial]; A_cur_ref(0, 1) = R_A[4 * r.trial + 1];
        A_cur_ref(1, 0) = R_A[4 * r.trial + 2]; A_cur_ref(1, 1) = R_A[4 * r.trial + 3];
        new_feature->grad = A_cur_ref * r.ref_ftr->grad;
        new_feature->grad.normalize();
      }
      if (predict) {  // what the device's selection must have picked, for the optimizer's drop-in to check
        pred_point.push_back(pt);
        pred_px.push_back(r.px[0]); pred_px.push_back(r.px[1]);
        pred_level.push_back(r.search_level);
        pred_trial.push_back(r.trial);
      }
      it = cell.erase(it);
      matched = true;  // at most one feature per cell
    }
    if (matched) ++n_matches_;
    if (n_matches_ > (size_t)Config::maxFts()) break;
  }
  if (predict) {  // published under the lane's mutex, in one piece
    std::lock_guard<std::mutex> guard(spec_lane->mut);
    svo_hip::Speculation& sp = spec_lane->spec;
    sp.point.swap(pred_point); sp.px.swap(pred_px); sp.level.swap(pred_level); sp.trial.swap(pred_trial);
    sp.valid = !sp.point.empty();
  }
  SVO_STOP_TIMER("feature_align");
}

}  // namespace svo
