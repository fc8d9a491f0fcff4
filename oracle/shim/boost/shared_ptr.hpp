// oracle/shim -- TEST INFRASTRUCTURE ONLY: boost::shared_ptr -> std::shared_ptr
#pragma once
#include <memory>
namespace boost { using std::shared_ptr; }
