// oracle/shim -- TEST INFRASTRUCTURE ONLY: the boost.thread names the reference uses, on std::
#pragma once
#include <condition_variable>
#include <mutex>
#include <thread>
#include <boost/bind.hpp>
namespace boost {
using std::mutex;
using std::unique_lock;
using std::condition_variable;
class thread {
 public:
  template <typename F, typename... A> explicit thread(F&& f, A&&... a) : t_(std::forward<F>(f), std::forward<A>(a)...) {}
  void interrupt() {}
  void join() { if (t_.joinable()) t_.join(); }
 private:
  std::thread t_;
};
namespace this_thread { inline bool interruption_requested() { return true; } }
}  // namespace boost
