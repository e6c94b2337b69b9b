// Row f3 of the scope table (MatrixMarket ingest at scale): wall time of cask::io::readMatrix (text, counting-sort
// CSR build) against readMatrixCached (binary cache beside the file) on one matrix.
//   build/ingest_time <file.mtx>
#include <chrono>
#include <cstdio>
#include <string>

#include "cask/IO.hpp"

int main(int argc, char **argv) {
  if (argc < 2) {
    std::fprintf(stderr, "usage: %s <file.mtx>\n", argv[0]);
    return 2;
  }
  const std::string path = argv[1];
  std::remove((path + ".csrbin").c_str());
  auto now = [] { return std::chrono::steady_clock::now(); };
  auto secs = [](std::chrono::steady_clock::time_point a, std::chrono::steady_clock::time_point b) {
    return std::chrono::duration<double>(b - a).count();
  };
  auto t0 = now();
  cask::CsrMatrix a = cask::io::readMatrix(path);
  auto t1 = now();
  cask::CsrMatrix b = cask::io::readMatrixCached(path);      // parses the text again and writes the cache
  auto t2 = now();
  cask::CsrMatrix c = cask::io::readMatrixCached(path);      // served from the cache
  auto t3 = now();
  const bool same = a == b && a == c;
  std::printf("{\"file\": \"%s\", \"rows\": %d, \"cols\": %d, \"nnz\": %d, \"readMatrix_s\": %.3f, "
              "\"readMatrixCached_first_s\": %.3f, \"readMatrixCached_again_s\": %.3f, \"identical\": %s}\n",
              path.c_str(), a.n, a.m, a.nnzs, secs(t0, t1), secs(t1, t2), secs(t2, t3), same ? "true" : "false");
  return same ? 0 : 1;
}
