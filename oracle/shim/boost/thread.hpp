// oracle/shim -- TEST INFRASTRUCTURE ONLY: the boost.thread names the reference uses, on std::.
// Interruption (DepthFilter::stopThread, depth_filter.cpp:69-80, and the interruption_requested()
// test of its loop) is a per-thread flag; condition_variable::wait polls it every 20 ms and
// leaves by throwing thread_interrupted, like boost's interruption points do.
#pragma once
#include <atomic>
#include <chrono>
#include <condition_variable>
#include <memory>
#include <mutex>
#include <thread>
#include <boost/bind.hpp>
namespace boost {
using std::mutex;
using std::unique_lock;
struct thread_interrupted {};
namespace detail {
struct thread_data { std::atomic<bool> interrupt; thread_data() : interrupt(false) {} };
inline thread_data*& current() { static thread_local thread_data* p = nullptr; return p; }
}  // namespace detail
namespace this_thread {
inline bool interruption_requested() { return detail::current() && detail::current()->interrupt.load(); }
}  // namespace this_thread
class condition_variable {
 public:
  void notify_one() { cv_.notify_one(); }
  void notify_all() { cv_.notify_all(); }
  void wait(std::unique_lock<std::mutex>& lk) {  // callers loop on their predicate: early returns are fine
    if (this_thread::interruption_requested()) throw thread_interrupted();
    cv_.wait_for(lk, std::chrono::milliseconds(20));
    if (this_thread::interruption_requested()) throw thread_interrupted();
  }
 private:
  std::condition_variable cv_;
};
class thread {
 public:
  template <typename F, typename... A> explicit thread(F&& f, A&&... a) : d_(new detail::thread_data) {
    std::shared_ptr<detail::thread_data> d = d_;
    auto fn = std::bind(std::forward<F>(f), std::forward<A>(a)...);
    t_ = std::thread([d, fn]() mutable {
      detail::current() = d.get();
      try { fn(); } catch (const thread_interrupted&) {}
      detail::current() = nullptr;
    });
  }
  ~thread() { if (t_.joinable()) { d_->interrupt = true; t_.join(); } }
  void interrupt() { d_->interrupt = true; }
  void join() { if (t_.joinable()) t_.join(); }
 private:
  std::shared_ptr<detail::thread_data> d_;
  std::thread t_;
};
}  // namespace boost
