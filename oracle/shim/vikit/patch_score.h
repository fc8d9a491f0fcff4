// -*- C++ -*-
// oracle/shim/vikit/patch_score.h -- TEST INFRASTRUCTURE ONLY.  Bodies: ../../orc_vikit.h.
#pragma once
#include <stdint.h>
extern "C" {
#include "orc_vikit.h"
}
namespace vk { namespace patch_score {
template <int HALF_PATCH_SIZE> class ZMSSD {
 public:
  static const int patch_size_ = 2 * HALF_PATCH_SIZE;
  static const int patch_area_ = patch_size_ * patch_size_;
  static const int threshold_ = 2000 * patch_area_;
  uint8_t* ref_patch_;
  int sumA_, sumAA_;
  ZMSSD(uint8_t* ref_patch) : ref_patch_(ref_patch) {
    static_assert(HALF_PATCH_SIZE == 4, "8x8 only");
    orc_zmssd_init(ref_patch_, &sumA_, &sumAA_);
  }
  static int threshold() { return threshold_; }
  int computeScore(uint8_t* cur_patch, int stride) const {
    return orc_zmssd_score(ref_patch_, sumA_, sumAA_, cur_patch, stride);
  }
};
}}  // namespace vk::patch_score
