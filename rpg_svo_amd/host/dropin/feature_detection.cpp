// Drop-in body for svo/src/feature_detection.cpp: AbstractDetector / FastDetector of
// svo/include/svo/feature_detection.h with detect() on the MI355X (svo_hip_fast_detect, K7):
// FAST-10 + FAST score + 3x3 non-max + Shi-Tomasi + best corner per free grid cell over the
// pyramid levels, on the device pyramid of the frame.  The occupancy grid stays a host
// vector<bool> (it is written feature by feature by the depth filter).  Optional fifth
// replacement: seed initialisation runs at keyframes only.
#include <svo/feature_detection.h>

#include <cmath>

#include <svo/feature.h>

#include "marshal.h"

namespace svo {
namespace feature_detection {

AbstractDetector::AbstractDetector(const int img_width, const int img_height, const int cell_size, const int n_pyr_levels)
    : cell_size_(cell_size), n_pyr_levels_(n_pyr_levels),
      grid_n_cols_((int)std::ceil(static_cast<double>(img_width) / cell_size_)),
      grid_n_rows_((int)std::ceil(static_cast<double>(img_height) / cell_size_)),
      grid_occupancy_((size_t)grid_n_cols_ * grid_n_rows_, false) {}

void AbstractDetector::resetGrid() { grid_occupancy_.assign(grid_occupancy_.size(), false); }

void AbstractDetector::setGridOccpuancy(const Vector2d& px) {
  grid_occupancy_.at(static_cast<int>(px[1] / cell_size_) * grid_n_cols_ + static_cast<int>(px[0] / cell_size_)) = true;
}

void AbstractDetector::setExistingFeatures(const Features& fts) {
  for (Features::const_iterator it = fts.begin(); it != fts.end(); ++it) setGridOccpuancy((*it)->px);
}

FastDetector::FastDetector(const int img_width, const int img_height, const int cell_size, const int n_pyr_levels)
    : AbstractDetector(img_width, img_height, cell_size, n_pyr_levels) {}

void FastDetector::detect(Frame* frame, const ImgPyr& img_pyr, const double detection_threshold, Features& fts) {
  using namespace hip_dropin;
  (void)img_pyr;  // the device pyramid of `frame` (built by K0 from its level 0) is the one searched
  svo_hip::Device& dev = ensureDevice(*frame);
  const int L = svo_hip::Device::LANE_MAPPING;  // seeds are initialised by the mapping thread
  svo_hip::Lane& lane = dev.lane(L);
  std::lock_guard<std::mutex> guard(lane.mut);
  dev.beginCall(L);
  svo_hip::Arena& a = lane.arena;
  a.reset();
  const size_t n_cells = grid_occupancy_.size();
  a.reserve(((size_t)1 << 16) + n_cells * 32);
  FrameTable frames(dev, L);
  const int idx = frames.indexOf(frame);
  int32_t* d_slot; uint8_t* d_occ;
  int32_t* slot = a.alloc<int32_t>(1, &d_slot);
  uint8_t* occ = a.alloc<uint8_t>(n_cells, &d_occ);
  *slot = frames.slot(idx);
  for (size_t k = 0; k < n_cells; ++k) occ[k] = grid_occupancy_[k] ? 1 : 0;
  a.endInputs();
  int32_t *d_xy, *d_level; float* d_score;
  int32_t* xy = a.alloc<int32_t>(2 * n_cells, &d_xy);
  int32_t* level = a.alloc<int32_t>(n_cells, &d_level);
  float* score = a.alloc<float>(n_cells, &d_score);

  // score maps + per-cell keys: reuse the lane's matcher scratch, grown on demand
  const size_t need = svo_hip_fast_workspace_bytes(&dev.layout(), 1, (int)n_cells);
  if (need > lane.workspace_bytes) {
    if (lane.d_workspace) { svo_hip::check(svo_hip_stream_sync(lane.stream), "sync"); svo_hip_free(lane.d_workspace); }
    svo_hip::check(svo_hip_malloc(&lane.d_workspace, need), "svo_hip_malloc(workspace)");
    lane.workspace_bytes = need;
  }
  a.upload(lane.stream);
  svo_hip::check(svo_hip_fast_detect(&dev.layout(), dev.store(), 1, d_slot, n_pyr_levels_, /*fast threshold*/ 20, cell_size_,
                                     grid_n_cols_, grid_n_rows_, d_occ, detection_threshold, d_xy, d_level, d_score,
                                     lane.d_workspace, lane.workspace_bytes, lane.stream),
                 "svo_hip_fast_detect");
  a.download(lane.stream);
  svo_hip::check(svo_hip_stream_sync(lane.stream), "svo_hip_stream_sync");

  // one feature per cell whose best corner beats the threshold, in cell order (:107-110)
  for (size_t k = 0; k < n_cells; ++k)
    if (score[k] > detection_threshold)
      fts.push_back(new Feature(frame, Vector2d(xy[2 * k], xy[2 * k + 1]), level[k]));
  resetGrid();
}

}  // namespace feature_detection
}  // namespace svo
