// Host utilities of the CASK surface: Timer, alignment helpers, the "Result  k=v,"
// log lines and the design-parameter sweep (Parameter / ChainedParameterRange).
// Same names and behaviour as the reference's src/runtime/Utils.hpp (:15-202);
// the sweep order -- first parameter fastest -- is pinned by the reference's
// test/TestUtils.cpp:12-49 and by tests/cpp/test_host.cpp here.  No Boost.
#ifndef CASK_UTILS_HPP
#define CASK_UTILS_HPP

#include <chrono>
#include <iostream>
#include <map>
#include <sstream>
#include <stdexcept>
#include <string>
#include <vector>

namespace cask {
namespace utils {

// Named stopwatches: tic(name) ... toc(name) records one duration per name.
class Timer {
  using clock_type = std::chrono::high_resolution_clock;
  using duration_type = std::chrono::duration<double>;
  std::map<std::string, clock_type::time_point> running;
  std::map<std::string, duration_type> finished;

 public:
  void tic(std::string name) { running[name] = clock_type::now(); }

  duration_type toc(std::string name) {
    auto it = running.find(name);
    if (it == running.end()) throw std::invalid_argument("No previous tic() with " + name);
    finished[name] = clock_type::now() - it->second;
    running.erase(it);
    return finished[name];
  }

  duration_type get(std::string name) {
    auto it = finished.find(name);
    if (it == finished.end()) throw std::invalid_argument("No previous tic()/toc() with " + name);
    return it->second;
  }
};

template <typename T>
void print(T v, std::string message = "") {
  std::cout << message;
  for (const auto &e : v) std::cout << e << " ";
  std::cout << std::endl;
}

// Pad v with T{} until its byte size is a multiple of widthInBytes (at most one width of padding).
template <typename T>
void align(std::vector<T> &v, int widthInBytes) {
  const size_t per = static_cast<size_t>(widthInBytes) / sizeof(T);
  if (per == 0) return;
  const size_t rem = (v.size() * sizeof(T)) % static_cast<size_t>(widthInBytes);
  if (rem == 0) return;
  size_t add = (static_cast<size_t>(widthInBytes) - rem + sizeof(T) - 1) / sizeof(T);
  if (add > per) add = per;
  v.resize(v.size() + add, T{});
}

inline int align(int bytes, int to) { return bytes % to == 0 ? bytes : (bytes / to + 1) * to; }

template <typename T>
long size_bytes(const std::vector<T> &v) {
  return static_cast<long>(sizeof(T) * v.size());
}

inline int ceilDivide(int a, int b) {
  if (a < 0 || b < 0) throw std::invalid_argument("ceilDivide: arguments must be positive");
  return a / b + (a % b != 0);
}

// "Result  <key>=<v>,<v>,..." -- two spaces after "Result", a comma after every value, and for the
// variadic form the values in REVERSE argument order followed by the first (the reference's
// recursion, Utils.hpp:88-112, prints the tail before the head).
inline void logResultR(std::string key) { std::cout << "Result " << " " << key << "="; }

template <typename Arg, typename... Args>
void logResultR(std::string key, Arg a, Args... rest) {
  logResultR(key, rest...);
  std::cout << a << ",";
}

template <typename U>
void logResult(std::string key, std::vector<U> vals) {
  logResultR(key);
  for (const auto &v : vals) std::cout << v << ",";
  std::cout << std::endl;
}

template <typename Arg, typename... Args>
void logResult(std::string key, Arg a, Args... rest) {
  logResultR(key, rest...);
  std::cout << a << ",";
  std::cout << std::endl;
}

// One design parameter: a value inside [start, end] stepping by `step`.
template <typename T = int>
class Parameter {
  Parameter(std::string n, T s, T e, T st, T v) : start(s), end(e), step(st), value(v), name(n) {}

 public:
  T start, end, step;
  T value;
  std::string name;

  Parameter(std::string n, T s, T e, T st) : Parameter(n, s, e, st, s) {}
  Parameter(std::string n, T single) : Parameter(n, single, single, 1, single) {}

  Parameter first() { return Parameter(name, start, end, step, start); }
  Parameter last() { return Parameter(name, start, end, step, end); }
  Parameter next() {
    if (value + step > end) throw std::invalid_argument("Invalid call to next() - no more elements");
    return Parameter(name, start, end, step, value + step);
  }
  bool hasNext() { return value != end; }
};

template <typename T>
inline std::ostream &operator<<(std::ostream &s, const Parameter<T> &p) {
  s << "Parameter{" << p.start << "," << p.end << "," << p.step << "}";
  return s;
}

// Odometer over several parameters; the FIRST one turns fastest.
template <typename T = int>
class ChainedParameterRange {
  std::vector<Parameter<T>> range;

 public:
  ChainedParameterRange(std::vector<Parameter<T>> r) : range(r) {}
  ChainedParameterRange(std::initializer_list<Parameter<T>> r) : range(r) {}

  void start() {
    for (auto &p : range) p = p.first();
  }

  bool hasNext() {
    for (auto &p : range)
      if (p.hasNext()) return true;
    return false;
  }

  void next() {
    size_t i = 0;
    while (i < range.size() && !range[i].hasNext()) {
      range[i] = range[i].first();
      i++;
    }
    if (i == range.size()) throw std::invalid_argument("No next element available");
    range[i] = range[i].next();
  }

  Parameter<T> getParam(std::string name) {
    for (auto &p : range)
      if (p.name == name) return p;
    throw std::invalid_argument("Param not found " + name);
  }
};

}  // namespace utils
}  // namespace cask

#endif  // CASK_UTILS_HPP
