// oracle/shim -- TEST INFRASTRUCTURE ONLY
#pragma once
namespace boost { class noncopyable { protected: noncopyable() {} ~noncopyable() {} noncopyable(const noncopyable&) = delete; noncopyable& operator=(const noncopyable&) = delete; }; }
