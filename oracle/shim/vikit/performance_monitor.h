// oracle/shim -- TEST INFRASTRUCTURE ONLY: inert vk::PerformanceMonitor
#pragma once
#include <string>
namespace vk { class PerformanceMonitor { public: void log(const std::string&, double) {} void startTimer(const std::string&) {} void stopTimer(const std::string&) {} }; }
