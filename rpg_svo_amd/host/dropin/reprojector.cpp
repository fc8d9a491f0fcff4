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
#include <svo/reprojector.h>

#include <algorithm>
#include <stdexcept>
#include <unordered_map>

#include <svo/config.h>
#include <svo/feature.h>
#include <svo/frame.h>
#include <svo/map.h>
#include <svo/point.h>

#include "marshal.h"

namespace svo {

namespace {
struct Outcome {  // what one findMatchDirect trial left in the Matcher
  bool ok;
  Vector2d px;
  int search_level;
  Feature* ref_ftr;
  Matrix2d A_cur_ref;
};
typedef std::pair<FramePtr, double> KfDist;
bool closerKf(const KfDist& a, const KfDist& b) { return a.second < b.second; }
}  // namespace

void Reprojector::reprojectMap(FramePtr frame, std::vector<std::pair<FramePtr, std::size_t> >& overlap_kfs) {
  resetGrid();

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
  std::unordered_map<Point*, Outcome> outcome;
  if (options_.find_match_direct) {
    size_t n_binned = 0;
    for (size_t k = 0; k < grid_.cells.size(); ++k) n_binned += grid_.cells[k]->size();
    outcome.reserve(2 * n_binned);
    using namespace hip_dropin;
    // The reference observation of a trial (Point::getCloseViewObs, matcher.cpp:137) is chosen HERE,
    // by the reference's own host code: a trial then ships exactly one svo::Feature, and only the
    // keyframes that actually serve as reference need their pyramid on the device (a point seen from
    // 40 keyframes no longer pins 40 pool slots for one 10x10 template).
    std::vector<Candidate*> trials;
    std::vector<Feature*> trial_ref;
    const Vector3d cur_pos(frame->pos());
    for (size_t k = 0; k < grid_.cells.size(); ++k)
      for (Cell::iterator c = grid_.cells[k]->begin(); c != grid_.cells[k]->end(); ++c) {
        if (c->pt->type_ == Point::TYPE_DELETED) continue;
        Feature* ref_ftr = NULL;
        if (!c->pt->getCloseViewObs(cur_pos, ref_ftr)) {  // findMatchDirect returns false at once (:137-138)
          Outcome r;
          r.ok = false; r.px = c->px; r.search_level = 0; r.ref_ftr = NULL;
          outcome[c->pt] = r;
          continue;
        }
        trials.push_back(&*c);
        trial_ref.push_back(ref_ftr);
      }
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
      a.reserve(((size_t)1 << 16) + M * 512 + n_obs * 128 + 4096 * 32);
      FrameTable frames(dev, L);
      const int i_cur = frames.indexOf(frame.get());

      int32_t *d_cur, *d_ptr; double* d_pos;
      int32_t* cur = a.alloc<int32_t>(M, &d_cur);
      double* pos = a.alloc<double>(3 * M, &d_pos);
      int32_t* ptr = a.alloc<int32_t>(M + 1, &d_ptr);
      std::vector<double> px_in(2 * M);  // goes into the in/out block below
      FeatureColumns obs;
      obs.alloc(a, n_obs);
      std::vector<Feature*> obs_ftr(n_obs);
      size_t o = 0;
      for (size_t m = 0; m < M; ++m) {
        const Point* pt = trials[m]->pt;
        cur[m] = i_cur;
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
      a.endInputs();

      double *d_px, *d_A; int32_t *d_ok, *d_ref, *d_lvl;
      double* px = a.alloc<double>(2 * M, &d_px);  // in: projection, out: refined pixel (uploadAll + download)
      std::copy(px_in.begin(), px_in.end(), px);
      int32_t* ok = a.alloc<int32_t>(M, &d_ok);
      int32_t* ref = a.alloc<int32_t>(M, &d_ref);
      int32_t* lvl = a.alloc<int32_t>(M, &d_lvl);
      double* A = a.alloc<double>(4 * M, &d_A);

      const svo_hip_camera cam = cameraOf(frame->cam_);
      void* ws = dev.workspace(lane, (int)M);
      stage_timer.device(a.used());
      a.uploadAll(lane.stream);
      svo_hip::check(svo_hip_find_match_direct(&dev.layout(), dev.store(), &cam, &ft, (int)M, d_cur, d_pos, d_ptr, &obs.dev,
                                               Config::nPyrLevels(), matcher_.options_.align_max_iter, d_px, d_ok, d_ref, d_lvl,
                                               d_A, NULL, ws, lane.workspace_bytes, lane.stream),
                     "svo_hip_find_match_direct");
      a.download(lane.stream);
      svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");
      stage_timer.unmarshal();

      for (size_t m = 0; m < M; ++m) {
        Outcome r;
        r.ok = ok[m] != 0;
        r.px = Vector2d(px[2 * m], px[2 * m + 1]);
        r.search_level = lvl[m];
        r.ref_ftr = ref[m] >= 0 ? obs_ftr[ref[m]] : NULL;
        r.A_cur_ref(0, 0) = A[4 * m]; r.A_cur_ref(0, 1) = A[4 * m + 1];
        r.A_cur_ref(1, 0) = A[4 * m + 2]; r.A_cur_ref(1, 1) = A[4 * m + 3];
        outcome[trials[m]->pt] = r;
      }
    }
  }

  // ---- 4. per cell, in the shuffled order: the best-quality point that matched ---------------
  for (size_t i = 0; i < grid_.cells.size(); ++i) {
    Cell& cell = *grid_.cells.at(grid_.cell_order[i]);
    // good points before unknown ones before candidates (stable, like std::list::sort)
    cell.sort([](Candidate& l, Candidate& r) { return l.pt->type_ > r.pt->type_; });
    bool matched = false;
    for (Cell::iterator it = cell.begin(); it != cell.end() && !matched;) {
      ++n_trials_;
      Point* pt = it->pt;
      if (pt->type_ == Point::TYPE_DELETED) { it = cell.erase(it); continue; }
      Outcome r;
      if (options_.find_match_direct) {
        std::unordered_map<Point*, Outcome>::iterator f = outcome.find(pt);
        if (f == outcome.end()) throw std::logic_error("Reprojector: candidate without a device trial");
        r = f->second;
      } else {  // accept the projection as it is
        r.ok = true; r.px = it->px; r.search_level = 0; r.ref_ftr = NULL;
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
        new_feature->grad = r.A_cur_ref * r.ref_ftr->grad;
        new_feature->grad.normalize();
      }
      it = cell.erase(it);
      matched = true;  // at most one feature per cell
    }
    if (matched) ++n_matches_;
    if (n_matches_ > (size_t)Config::maxFts()) break;
  }
  SVO_STOP_TIMER("feature_align");
}

}  // namespace svo
