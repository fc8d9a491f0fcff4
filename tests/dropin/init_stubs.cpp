// tests/dropin/init_stubs.cpp -- TEST INFRASTRUCTURE ONLY.
// svo/src/initialization.cpp needs OpenCV's KLT tracker and vikit's homography decomposition, neither of
// which exists in this image; the replay starts from setFirstFrame() like the reference's own benchmark
// (svo_ros/src/benchmark_node.cpp:216-235), so the two-view bootstrap is never entered.  A real build
// links the reference's initialization.cpp instead of this file.
#include <svo/initialization.h>

namespace svo {
namespace initialization {
InitResult KltHomographyInit::addFirstFrame(FramePtr) { return FAILURE; }
InitResult KltHomographyInit::addSecondFrame(FramePtr) { return FAILURE; }
void KltHomographyInit::reset() {}
}  // namespace initialization
}  // namespace svo

namespace vk {
int g_halfsample_mode = 2;  // x86 dispatch of vk::halfSample (knob of the shim, oracle/shim/vikit/vision.h)
}
