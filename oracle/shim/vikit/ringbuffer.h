// oracle/shim -- TEST INFRASTRUCTURE ONLY: vk::RingBuffer as FrameHandlerBase uses it
// (frame_handler_base.h:94-95, frame_handler_base.cpp:113-119): user-feedback statistics only.
#pragma once
#include <cstddef>
#include <vector>
namespace vk {
template <typename T> class RingBuffer {
 public:
  explicit RingBuffer(int size) : arr_((size_t)size), begin_(0), end_(-1), n_(0) {}
  void push_back(const T& v) {
    if (n_ < (int)arr_.size()) { ++end_; ++n_; } else { end_ = (end_ + 1) % (int)arr_.size(); begin_ = (begin_ + 1) % (int)arr_.size(); }
    arr_[(size_t)end_] = v;
  }
  bool empty() const { return n_ == 0; }
  int size() const { return n_; }
  T getMean() const { if (n_ == 0) return T(); T s = T(); for (int i = 0; i < n_; ++i) s += arr_[(size_t)((begin_ + i) % (int)arr_.size())]; return s / (T)n_; }
 private:
  std::vector<T> arr_;
  int begin_, end_, n_;
};
}  // namespace vk
