// HostCopyPool (cask_amd/csrc/host_copy.hpp) on the CPU: every byte arrives, whatever the size, alignment and number of
// helpers; helpers that went to sleep wake up; several caller threads at once; and -- built with -fsanitize=thread as
// well as with ASan / UBSan (`make asan`) -- no data race between a caller and the helpers.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <thread>
#include <vector>

#include "host_copy.hpp"

static long g_checks = 0, g_failures = 0;
#define CHECK(cond) do { g_checks++; if (!(cond)) { if (g_failures++ < 10) std::printf("FAIL %s:%d %s\n", __FILE__, __LINE__, #cond); } } while (0)

int main() {
  using caskhip::HostCopyPool;
  for (int helpers : {0, 1, 3, 7}) {
    HostCopyPool pool(helpers);
    CHECK(pool.helpers() == helpers);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&] { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return st; };
    for (size_t bytes : {(size_t)0, (size_t)1, (size_t)4095, (size_t)131072, (size_t)131073, (size_t)499608, (size_t)(1 << 20) + 17, (size_t)12683824}) {
      for (size_t mis : {(size_t)0, (size_t)8, (size_t)3}) {
        std::vector<unsigned char> src(bytes + 64), dst(bytes + 128, 0xEE);
        for (auto &b : src) b = (unsigned char)rnd();
        pool.copy(dst.data() + 32 + mis, src.data() + mis, bytes);
        bool same = true;
        for (size_t i = 0; i < bytes; i++) same = same && dst[32 + mis + i] == src[mis + i];
        CHECK(same);
        for (size_t i = 0; i < 32 + mis; i++) CHECK(dst[i] == 0xEE);                       // nothing in front
        for (size_t i = 32 + mis + bytes; i < dst.size(); i++) CHECK(dst[i] == 0xEE);      // nothing behind
      }
    }
    // helpers asleep (the spin budget is 300 us) must wake up
    std::this_thread::sleep_for(std::chrono::milliseconds(5));
    std::vector<double> a(200000), b(200000, -1.0);
    for (size_t i = 0; i < a.size(); i++) a[i] = 0.25 * (double)i;
    for (int rep = 0; rep < 50; rep++) {
      pool.copy(b.data(), a.data(), a.size() * 8);
      CHECK(b[0] == 0.0 && b[199999] == 0.25 * 199999 && b[100000] == 25000.0);
      b[100000] = -1.0;
      if (rep % 10 == 9) std::this_thread::sleep_for(std::chrono::milliseconds(1));
    }
    // several callers at once (serialised inside)
    std::vector<std::thread> callers;
    std::vector<int> ok(4, 0);
    for (int t = 0; t < 4; t++)
      callers.emplace_back([&, t] {
        std::vector<double> s(70000 + 1000 * t, 1.0 + t), d(s.size(), 0.0);
        bool good = true;
        for (int rep = 0; rep < 20; rep++) {
          pool.copy(d.data(), s.data(), s.size() * 8);
          for (size_t i = 0; i < d.size(); i += 997) good = good && d[i] == 1.0 + t;
          d.assign(d.size(), 0.0);
        }
        ok[(size_t)t] = good;
      });
    for (auto &c : callers) c.join();
    for (int t = 0; t < 4; t++) CHECK(ok[(size_t)t]);
  }
  std::printf("%ld checks, %ld failures\n", g_checks, g_failures);
  return g_failures ? 1 : 0;
}
