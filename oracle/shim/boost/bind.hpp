// -*- C++ -*-
// oracle/shim/boost/bind.hpp -- TEST INFRASTRUCTURE ONLY.
// The two boost::bind idioms the reference uses: (1) plain function binding -> std::bind;
// (2) bind(&pair::second, _N) compared with operator< to build a sort predicate
// (svo/src/reprojector.cpp:75-76, svo/src/map.cpp:139-140).
#pragma once
#include <functional>
#include <type_traits>
namespace boost {
struct arg1 {}; struct arg2 {};
template <typename M, typename C, int N> struct bound_member {
  M C::*pm;
  template <typename A, typename B> const M& operator()(const A& a, const B& b) const { return pick(a, b, std::integral_constant<int, N>()); }
 private:
  template <typename A, typename B> const M& pick(const A& a, const B&, std::integral_constant<int, 1>) const { return a.*pm; }
  template <typename A, typename B> const M& pick(const A&, const B& b, std::integral_constant<int, 2>) const { return b.*pm; }
};
template <typename L, typename R> struct less_bound {
  L l; R r;
  template <typename A, typename B> bool operator()(const A& a, const B& b) const { return l(a, b) < r(a, b); }
};
template <typename M, typename C, int N1, int N2>
less_bound<bound_member<M, C, N1>, bound_member<M, C, N2>> operator<(const bound_member<M, C, N1>& l, const bound_member<M, C, N2>& r) {
  return less_bound<bound_member<M, C, N1>, bound_member<M, C, N2>>{l, r};
}
template <typename L, typename R> struct greater_bound {
  L l; R r;
  template <typename A, typename B> bool operator()(const A& a, const B& b) const { return l(a, b) > r(a, b); }
};
template <typename M, typename C, int N1, int N2>
greater_bound<bound_member<M, C, N1>, bound_member<M, C, N2>> operator>(const bound_member<M, C, N1>& l, const bound_member<M, C, N2>& r) {
  return greater_bound<bound_member<M, C, N1>, bound_member<M, C, N2>>{l, r};
}
template <typename M, typename C> bound_member<M, C, 1> bind(M C::*pm, const arg1&) { return bound_member<M, C, 1>{pm}; }
template <typename M, typename C> bound_member<M, C, 2> bind(M C::*pm, const arg2&) { return bound_member<M, C, 2>{pm}; }
// free / static functions of two arguments
template <typename R, typename A, typename B> std::function<R(A, B)> bind(R (*f)(A, B), const arg1&, const arg2&) { return std::function<R(A, B)>(f); }
// member function of two arguments on an object pointer (frame_handler_mono.cpp:47-48)
template <typename R, typename C, typename A, typename B> std::function<R(A, B)> bind(R (C::*f)(A, B), C* obj, const arg1&, const arg2&) {
  return [f, obj](A a, B b) -> R { return (obj->*f)(a, b); };
}
}  // namespace boost
static const boost::arg1 _1 = boost::arg1();
static const boost::arg2 _2 = boost::arg2();
