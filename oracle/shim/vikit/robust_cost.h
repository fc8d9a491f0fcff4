// -*- C++ -*-
// oracle/shim/vikit/robust_cost.h -- TEST INFRASTRUCTURE ONLY.  Bodies: ../../orc_vikit.h.
#pragma once
#include <memory>
#include <vector>
#include <vikit/math_utils.h>
extern "C" {
#include "orc_vikit.h"
}
namespace vk { namespace robust_cost {
class ScaleEstimator { public: virtual ~ScaleEstimator() {} virtual float compute(std::vector<float>& errors) const = 0; };
typedef std::shared_ptr<ScaleEstimator> ScaleEstimatorPtr;
class MADScaleEstimator : public ScaleEstimator {
 public:
  virtual float compute(std::vector<float>& errors) const { return ORC_MAD_NORMALIZER * vk::getMedian(errors); }
};
class WeightFunction { public: virtual ~WeightFunction() {} virtual float value(const float& x) const = 0; };
typedef std::shared_ptr<WeightFunction> WeightFunctionPtr;
class TukeyWeightFunction : public WeightFunction {
 public:
  virtual float value(const float& x) const { return orc_tukey_weight(x); }
};
}}  // namespace vk::robust_cost
