// -*- C++ -*-
// oracle/shim -- TEST INFRASTRUCTURE ONLY.
// boost::math::normal_distribution<float> + pdf(), restated from boost/math/distributions/
// normal.hpp (exponent = x-mean; exponent *= -exponent; exponent /= 2*sd*sd; result =
// exp(exponent); result /= sd*sqrt(2*pi)), all in RealType.  Used by DepthFilter::updateSeed
// (svo/src/depth_filter.cpp:314-317).  The body is orc_normal_pdff (../orc_math.h) so the
// C oracle and _ref share it.
#pragma once
#include <cmath>
extern "C" {
#include "orc_vikit.h"
}
namespace boost { namespace math {
template <typename RealType = double> class normal_distribution {
 public:
  normal_distribution(RealType mean = 0, RealType sd = 1) : m_(mean), s_(sd) {}
  RealType mean() const { return m_; }
  RealType standard_deviation() const { return s_; }
 private:
  RealType m_, s_;
};
inline float pdf(const normal_distribution<float>& d, const float& x) {
  return orc_normal_pdff(x, d.mean(), d.standard_deviation());
}
}}  // namespace boost::math
