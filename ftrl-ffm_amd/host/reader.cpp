#include "reader.h"

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <future>
#include <iostream>

namespace ftrl {

std::unique_ptr<Parser> make_parser(const std::string &file_type) {
  if (file_type == "libsvm") return std::make_unique<LibsvmParser>();
  if (file_type == "libffm") return std::make_unique<FFMParser>();
  return nullptr;
}

Reader::Reader(const std::string &file_type) : parser(make_parser(file_type)) {}

void Reader::load_from_file(std::string_view file_name, int n_threads) {
  const std::string path(file_name);
  std::ifstream ifs(path, std::ios::binary);
  if (!ifs.good()) {
    std::cerr << "fail to open " << path << std::endl;
    std::exit(EXIT_FAILURE);
  }
  std::printf("Loading data from file: %s\n", path.c_str());
  const auto t0 = std::chrono::steady_clock::now();
  if (n_threads < 1) n_threads = 1;
  ifs.seekg(0, std::ios::end);
  const int64_t len = ifs.tellg();
  std::vector<int64_t> cut(static_cast<size_t>(n_threads) + 1, 0);
  cut[n_threads] = len;
  std::string unused;
  for (int i = 1; i < n_threads; i++) {  // next line start after len/n*i (reader.cpp:39-46)
    ifs.clear();
    ifs.seekg(len / n_threads * i);
    std::getline(ifs, unused);
    const int64_t pos = ifs.tellg();
    cut[i] = pos < 0 ? len : pos;
  }
  std::vector<std::vector<Sample>> parts(n_threads);
  std::vector<std::future<void>> futs;
  for (int i = 0; i < n_threads; i++)
    futs.emplace_back(std::async(std::launch::async, [&, i] {
      std::ifstream f(path, std::ios::binary);
      f.seekg(cut[i]);
      std::string line;
      while (f.tellg() < cut[i + 1] && std::getline(f, line)) {
        Sample s;
        parser->parse(line, s);
        parts[i].emplace_back(std::move(s));
      }
    }));
  for (auto &f : futs) f.get();
  size_t total = 0;
  for (auto &p : parts) total += p.size();
  std::printf("Total number of samples loaded: %zu\n", total);
  data.clear();
  data.reserve(total);
  for (auto &p : parts)
    for (auto &s : p) data.emplace_back(std::move(s));
  data_size = total;
  const double sec = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
  std::printf("parsing data time: %.4lfs\n", sec);
}

}  // namespace ftrl
