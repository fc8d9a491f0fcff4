// oracle/shim -- TEST INFRASTRUCTURE ONLY: vk::PerformanceMonitor restated from rpg_vikit
// (performance_monitor.{h,cpp}): named timers and logs, one CSV row per writeToFile() -- header =
// timer names then log names (std::map order), values with 15 fixed decimals -- in
// <trace_dir>/<trace_name>.csv.  Additionally keeps the last value of every entry in memory so the
// test harness can read them back through get() (the real class has no such accessor).  A timer /
// log that was not touched since the previous row is written as 0, as the real class resets them.
#pragma once
#include <chrono>
#include <fstream>
#include <map>
#include <set>
#include <string>
namespace vk {
class PerformanceMonitor {
 public:
  void init(const std::string& trace_name, const std::string& trace_dir) {
    if (trace_dir.empty()) return;
    ofs_.open((trace_dir + "/" + trace_name + ".csv").c_str());
    if (!ofs_.is_open()) return;  // (the real class throws; the harness runs without a trace directory)
    bool first = true;
    for (std::map<std::string, double>::const_iterator it = timers_.begin(); it != timers_.end(); ++it, first = false)
      ofs_ << (first ? "" : ",") << it->first;
    for (std::map<std::string, double>::const_iterator it = logs_.begin(); it != logs_.end(); ++it, first = false)
      ofs_ << (first ? "" : ",") << it->first;
    ofs_ << "\n";
  }
  void addTimer(const std::string& n) { timers_[n] = 0; }
  void addLog(const std::string& n) { logs_[n] = 0; }
  void writeToFile() {
    if (ofs_.is_open()) {
      ofs_.precision(15);
      ofs_.setf(std::ios::fixed, std::ios::floatfield);
      bool first = true;
      for (std::map<std::string, double>::const_iterator it = timers_.begin(); it != timers_.end(); ++it, first = false)
        ofs_ << (first ? "" : ",") << (touched_.count(it->first) ? it->second : 0.0);
      for (std::map<std::string, double>::const_iterator it = logs_.begin(); it != logs_.end(); ++it, first = false)
        ofs_ << (first ? "" : ",") << (touched_.count(it->first) ? it->second : 0.0);
      ofs_ << "\n";
      ofs_.flush();
    }
    touched_.clear();
  }
  void log(const std::string& n, double v) { logs_[n] = v; touched_.insert(n); }
  void startTimer(const std::string& n) { t0_[n] = clock::now(); }
  void stopTimer(const std::string& n) {
    timers_[n] = std::chrono::duration<double>(clock::now() - t0_[n]).count();
    touched_.insert(n);
  }
  double get(const std::string& n) const {
    std::map<std::string, double>::const_iterator it = timers_.find(n);
    if (it != timers_.end()) return it->second;
    it = logs_.find(n);
    return it == logs_.end() ? 0.0 : it->second;
  }
 private:
  typedef std::chrono::steady_clock clock;
  std::map<std::string, double> timers_, logs_;
  std::map<std::string, clock::time_point> t0_;
  std::set<std::string> touched_;
  std::ofstream ofs_;
};
}  // namespace vk
