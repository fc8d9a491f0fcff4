// oracle/shim -- TEST INFRASTRUCTURE ONLY: inert vk::Timer
#pragma once
namespace vk { class Timer { public: void start() {} double stop() { return 0; } double getTime() { return 0; } }; }
