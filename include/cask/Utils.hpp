// Host utilities of the CASK surface.  API-compatible with the reference's
// src/runtime/Utils.hpp (Timer :15-50, align/size_bytes/ceilDivide :62-86,
// logResult :88-112, Parameter :114-155, ChainedParameterRange :158-202), new
// implementation, no Boost.  The behaviours clients and log scrapers depend on:
//   * logResult prints "Result  <key>=" (two spaces) then every value followed by
//     a comma -- for the variadic form in REVERSE argument order, because the
//     reference's recursion prints the tail before the head;
//   * ChainedParameterRange is an odometer whose FIRST parameter turns fastest
//     (pinned by the reference's test/TestUtils.cpp:12-49 and tests/cpp/test_host.cpp).
#ifndef CASK_UTILS_HPP
#define CASK_UTILS_HPP

#include <chrono>
#include <cstddef>
#include <iostream>
#include <sstream>
#include <stdexcept>
#include <string>
#include <unordered_map>
#include <utility>
#include <vector>

namespace cask {
namespace utils {

// ---------------------------------------------------------------- stopwatches
// tic(name) starts (or restarts) a named stopwatch, toc(name) stops it and returns
// the elapsed seconds; get(name) returns the last completed measurement.
class Timer {
 public:
  using seconds = std::chrono::duration<double>;

  void tic(std::string name) { started_[name] = now(); }

  seconds toc(std::string name) {
    const auto hit = started_.find(name);
    if (hit == started_.end()) throw std::invalid_argument("No previous tic() with " + name);
    const seconds elapsed = now() - hit->second;
    started_.erase(hit);
    done_[name] = elapsed;
    return elapsed;
  }

  seconds get(std::string name) {
    const auto hit = done_.find(name);
    if (hit == done_.end()) throw std::invalid_argument("No previous tic()/toc() with " + name);
    return hit->second;
  }

 private:
  using clock = std::chrono::steady_clock;
  static clock::time_point now() { return clock::now(); }
  std::unordered_map<std::string, clock::time_point> started_;
  std::unordered_map<std::string, seconds> done_;
};

// ---------------------------------------------------------------- small helpers
template <typename Container>
void print(Container c, std::string message = "") {
  std::ostringstream line;
  line << message;
  for (const auto &item : c) line << item << " ";
  std::cout << line.str() << std::endl;
}

// Grow v with default-constructed elements until its size in bytes is a multiple of
// widthInBytes; never adds more than one width's worth of elements.
template <typename T>
void align(std::vector<T> &v, int widthInBytes) {
  const std::size_t width = static_cast<std::size_t>(widthInBytes);
  const std::size_t most = width / sizeof(T);
  std::size_t added = 0;
  while (added < most && (v.size() * sizeof(T)) % width != 0) {
    v.emplace_back();
    ++added;
  }
}

// smallest multiple of `to` that is >= bytes
inline int align(int bytes, int to) {
  const int over = bytes % to;
  return over ? bytes + (to - over) : bytes;
}

template <typename T>
long size_bytes(const std::vector<T> &v) {
  return static_cast<long>(v.size() * sizeof(T));
}

inline int ceilDivide(int a, int b) {
  if (a < 0 || b < 0) throw std::invalid_argument("ceilDivide: arguments must be positive");
  return (a + b - 1) / b;
}

// ---------------------------------------------------------------- "Result  k=v," lines
namespace detail {
inline void emitResult(const std::string &key, const std::vector<std::string> &values, bool newline) {
  std::ostringstream line;
  line << "Result " << " " << key << "=";
  for (const std::string &v : values) line << v << ",";
  std::cout << line.str();
  if (newline) std::cout << std::endl;
}

template <typename V>
std::string shown(const V &v) {
  std::ostringstream s;
  s << v;
  return s.str();
}

inline void collectReversed(std::vector<std::string> &) {}
template <typename Head, typename... Tail>
void collectReversed(std::vector<std::string> &out, const Head &h, const Tail &... t) {
  collectReversed(out, t...);      // tail first: the reference prints its arguments back to front
  out.push_back(shown(h));
}
}  // namespace detail

// header plus values, no line end (the building block the reference exposes as logResultR)
template <typename... Args>
void logResultR(std::string key, Args... values) {
  std::vector<std::string> shownValues;
  detail::collectReversed(shownValues, values...);
  detail::emitResult(key, shownValues, false);
}

template <typename U>
void logResult(std::string key, std::vector<U> values) {
  std::vector<std::string> shownValues;
  for (const U &v : values) shownValues.push_back(detail::shown(v));
  detail::emitResult(key, shownValues, true);
}

template <typename First, typename... Rest>
void logResult(std::string key, First first, Rest... rest) {
  std::vector<std::string> shownValues;
  detail::collectReversed(shownValues, first, rest...);
  detail::emitResult(key, shownValues, true);
}

// ---------------------------------------------------------------- design-parameter sweep
// One integer-like design parameter: `value` walks start, start+step, ... , end.
template <typename T = int>
class Parameter {
 public:
  T start, end, step;
  T value;
  std::string name;

  Parameter(std::string name_, T start_, T end_, T step_) : start(start_), end(end_), step(step_), value(start_), name(name_) {}
  Parameter(std::string name_, T only) : start(only), end(only), step(1), value(only), name(name_) {}

  Parameter first() { return at(start); }
  Parameter last() { return at(end); }
  bool hasNext() { return !(value == end); }
  Parameter next() {
    if (value + step > end) throw std::invalid_argument("Invalid call to next() - no more elements");
    return at(value + step);
  }

 private:
  Parameter at(T v) const {
    Parameter copy(*this);
    copy.value = v;
    return copy;
  }
};

template <typename T>
inline std::ostream &operator<<(std::ostream &s, const Parameter<T> &p) {
  return s << "Parameter{" << p.start << "," << p.end << "," << p.step << "}";
}

// Cross product of several parameters visited like an odometer: the first parameter is the
// least significant digit.
template <typename T = int>
class ChainedParameterRange {
 public:
  ChainedParameterRange(std::vector<Parameter<T>> params) : digits_(std::move(params)) {}
  ChainedParameterRange(std::initializer_list<Parameter<T>> params) : digits_(params) {}

  void start() {
    for (Parameter<T> &d : digits_) d = d.first();
  }

  bool hasNext() {
    for (Parameter<T> &d : digits_)
      if (d.hasNext()) return true;
    return false;
  }

  void next() {
    for (Parameter<T> &d : digits_) {
      if (d.hasNext()) {
        d = d.next();
        return;
      }
      d = d.first();                 // this digit wraps, carry into the next one
    }
    throw std::invalid_argument("No next element available");
  }

  Parameter<T> getParam(std::string name) {
    for (Parameter<T> &d : digits_)
      if (d.name == name) return d;
    throw std::invalid_argument("Param not found " + name);
  }

 private:
  std::vector<Parameter<T>> digits_;
};

}  // namespace utils
}  // namespace cask

#endif  // CASK_UTILS_HPP
