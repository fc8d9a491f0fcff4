// oracle/shim -- TEST INFRASTRUCTURE ONLY: vk::PerformanceMonitor keeping the last value of
// every timer / log in memory (the reference writes them to a trace CSV; the test harness
// reads them back through get()).
#pragma once
#include <chrono>
#include <map>
#include <string>
namespace vk {
class PerformanceMonitor {
 public:
  void init(const std::string&, const std::string&) {}
  void addTimer(const std::string& n) { val_[n] = 0; }
  void addLog(const std::string& n) { val_[n] = 0; }
  void writeToFile() {}
  void log(const std::string& n, double v) { val_[n] = v; }
  void startTimer(const std::string& n) { t0_[n] = clock::now(); }
  void stopTimer(const std::string& n) { val_[n] = std::chrono::duration<double>(clock::now() - t0_[n]).count(); }
  double get(const std::string& n) const { auto it = val_.find(n); return it == val_.end() ? 0.0 : it->second; }
 private:
  typedef std::chrono::steady_clock clock;
  std::map<std::string, double> val_;
  std::map<std::string, clock::time_point> t0_;
};
}  // namespace vk
