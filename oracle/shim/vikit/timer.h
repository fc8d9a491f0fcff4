// oracle/shim -- TEST INFRASTRUCTURE ONLY: vk::Timer on std::chrono (seconds)
#pragma once
#include <chrono>
namespace vk {
class Timer {
 public:
  Timer() : t0_(clock::now()), last_(0) {}
  void start() { t0_ = clock::now(); }
  double stop() { last_ = std::chrono::duration<double>(clock::now() - t0_).count(); return last_; }
  double getTime() const { return last_; }
 private:
  typedef std::chrono::steady_clock clock;
  clock::time_point t0_;
  double last_;
};
}  // namespace vk
