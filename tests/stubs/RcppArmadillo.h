// STUB -- NOT Rcpp, NOT Armadillo.  Declares only the names spamtree_amd/csrc/rcpp_exports.cpp uses, with plausible
// signatures, so that the compile-guarded R shim can be parsed and type-checked in an image without R
// (tests/test_rcpp_shim.py: g++ -fsyntax-only).  It pins NOTHING numerically and is never linked or run.
#pragma once
#include <cstddef>
#include <initializer_list>
#include <string>
#include <vector>

namespace arma {
typedef unsigned long long uword;
namespace fill { struct fill_zeros {}; static const fill_zeros zeros = fill_zeros(); }
template <typename T>
struct Col {
  uword n_elem = 0, n_rows = 0, n_cols = 1;
  Col() {}
  explicit Col(uword n) : n_elem(n), n_rows(n) {}
  T &operator()(uword i);
  const T &operator()(uword i) const;
  const T *memptr() const;
  T *memptr();
  const T *begin() const;
  const T *end() const;
};
typedef Col<double> vec;
typedef Col<uword> uvec;
struct mat {
  uword n_elem = 0, n_rows = 0, n_cols = 0;
  mat() {}
  mat(uword r, uword c, fill::fill_zeros) : n_elem(r * c), n_rows(r), n_cols(c) {}
  mat(const vec &) {}
  double &operator()(uword i);
  const double &operator()(uword i) const;
  const double *memptr() const;
  double *memptr();
  vec col(uword j) const;
};
struct cube {
  cube(uword r, uword c, uword s, fill::fill_zeros) {}
  double *memptr();
};
template <typename T>
struct field {
  uword n_elem = 0;
  field() {}
  explicit field(uword n) : n_elem(n) {}
  T &operator()(uword i);
  const T &operator()(uword i) const;
};
vec zeros(uword n);
}   // namespace arma

namespace Rcpp {
struct Argument {
  std::string name;
  template <typename T> Argument operator=(const T &) const { return *this; }
};
inline Argument Named(const std::string &n) { return Argument{n}; }
struct List {
  template <typename... A> static List create(const A &...) { return List(); }
};
[[noreturn]] void stop(const std::string &msg);
}   // namespace Rcpp
namespace R { double runif(double a, double b); }
