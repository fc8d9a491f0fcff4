// oracle/shim -- TEST INFRASTRUCTURE ONLY: boost::function -> std::function
#pragma once
#include <functional>
namespace boost { template <typename S> using function = std::function<S>; }
