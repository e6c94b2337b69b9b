// build/main <bench-path> <dse-params.json>  ->  dse_out.json
// The DSE executable of the CASK flow (reference: src/main.cpp:119-207), measuring on the GPU.
// The parameter file keeps the reference's schema (src/frontend/params.json):
//   {"dse_params": {"input_width": {"start","stop","step"}, "cache_size": {...}, ...}}
// input_width = lanes per row, cache_size = x tile width (doubles); num_pipes /
// num_controllers are read and ignored.  No Boost: a small scanner reads the integers.
#include <dirent.h>
#include <sys/stat.h>

#include <chrono>
#include <fstream>
#include <iostream>
#include <regex>
#include <sstream>

#include "cask/Dse.hpp"

namespace {
bool find_range(const std::string &text, const std::string &key, int &start, int &stop, int &step) {
  std::regex block("\"" + key + "\"\\s*:\\s*\\{([^}]*)\\}");
  std::smatch m;
  if (!std::regex_search(text, m, block)) return false;
  const std::string body = m[1];
  auto field = [&](const char *name, int &out) {
    std::regex f(std::string("\"") + name + "\"\\s*:\\s*(-?[0-9]+)");
    std::smatch fm;
    if (std::regex_search(body, fm, f)) out = std::stoi(fm[1]);
  };
  field("start", start);
  field("stop", stop);
  field("step", step);
  return true;
}
}  // namespace

int main(int argc, char **argv) {
  if (argc < 3 || std::string(argv[1]) == "--help") {
    std::cout << "Usage: ./main bench-path dse-params-file" << std::endl;
    return argc < 3 ? 1 : 0;
  }
  const std::string benchPath = argv[1], paramFile = argv[2];
  cask::dse::Benchmark benchmark;
  struct stat st;
  if (stat(benchPath.c_str(), &st) != 0) {
    std::cout << "Error: '" << benchPath << "' not a directory or valid file" << std::endl;
    return 1;
  }
  if (S_ISDIR(st.st_mode)) {
    std::cout << "Using " << benchPath << " as benchmark directory" << std::endl;
    std::vector<std::string> names;
    if (DIR *d = opendir(benchPath.c_str())) {
      while (dirent *e = readdir(d)) {
        std::string n = e->d_name;
        if (n.size() > 4 && n.substr(n.size() - 4) == ".mtx") names.push_back(n);
      }
      closedir(d);
    }
    std::sort(names.begin(), names.end());
    for (auto &n : names) benchmark.add_matrix_path(benchPath + "/" + n);
  } else {
    benchmark.add_matrix_path(benchPath);
  }
  std::cout << benchmark << std::endl;

  std::ifstream pf(paramFile);
  if (!pf) {
    std::cout << "Error: '" << paramFile << "' is not a file" << std::endl;
    return 1;
  }
  std::cout << "Using " << paramFile << " as param file" << std::endl;
  std::stringstream buf;
  buf << pf.rdbuf();
  cask::dse::DseParameters params;
  int a, b, c;
  a = params.inputWidth.start; b = params.inputWidth.end; c = params.inputWidth.step;
  if (find_range(buf.str(), "input_width", a, b, c)) params.inputWidth = cask::utils::Parameter<int>{"inputWidth", a, b, c};
  a = params.cacheSize.start; b = params.cacheSize.end; c = params.cacheSize.step;
  if (find_range(buf.str(), "cache_size", a, b, c)) params.cacheSize = cask::utils::Parameter<int>{"cacheSize", a, b, c};
  a = b = 0; c = 1;
  if (find_range(buf.str(), "wg_size", a, b, c)) {
    params.wgSize.clear();
    for (int v = a; v <= b; v *= 2) params.wgSize.push_back(v);
  }
  if (find_range(buf.str(), "items_per_thread", a, b, c)) {
    params.itemsPerThread.clear();
    for (int v = a; v <= b; v *= 2) params.itemsPerThread.push_back(v);
  }
  params.gflopsOnly = true;
  std::cout << params << std::endl;

  cask::model::Mi355xModel deviceModel;
  std::cout << "Device Model " << deviceModel << std::endl;
  cask::dse::SparkDse dseTool;
  auto start = std::chrono::high_resolution_clock::now();
  try {
    auto results = dseTool.run(benchmark, params, deviceModel);
    double took = std::chrono::duration<double>(std::chrono::high_resolution_clock::now() - start).count();
    cask::dse::write_dse_results(results, took, deviceModel);
    std::cout << "Wrote dse_out.json (" << results.size() << " matrices, " << took << " s)" << std::endl;
  } catch (std::exception &e) {
    std::cout << "DSE failed: " << e.what() << std::endl;
    return 2;
  }
  return 0;
}
