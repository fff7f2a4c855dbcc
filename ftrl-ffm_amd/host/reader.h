// reader.h -- whole-file loader (reference src/include/data/reader.h:15-27,
// src/data/reader.cpp:22-91): the file is cut into n_threads byte ranges aligned to line starts,
// each parsed by its own thread, results concatenated in file order.
#pragma once
#include <memory>
#include <string>
#include <string_view>
#include <vector>

#include "parser.h"

namespace ftrl {

class Reader {
 public:
  explicit Reader(const std::string &file_type);
  void load_from_file(std::string_view file_name, int n_threads);
  [[nodiscard]] size_t get_size() const { return data_size; }

  size_t data_size{0};
  std::vector<Sample> data;
  std::shared_ptr<Parser> parser;
};

std::unique_ptr<Parser> make_parser(const std::string &file_type);

}  // namespace ftrl
