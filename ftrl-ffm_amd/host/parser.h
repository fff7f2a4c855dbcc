// parser.h -- libsvm / libffm line parsers (reference src/include/data/parser.h:11-31,
// src/data/parser.cpp:11-103): label > 0 -> y = 1; zero values are dropped; libsvm rows get
// field 0; a malformed token throws std::out_of_range.
#pragma once
#include <string>

#include "types.h"

namespace ftrl {

class Parser {
 public:
  virtual ~Parser() = default;
  virtual void parse(const std::string &line, Sample &sample) = 0;
};

class LibsvmParser : public Parser {
 public:
  void parse(const std::string &line, Sample &sample) override;
};

class FFMParser : public Parser {
 public:
  void parse(const std::string &line, Sample &sample) override;
};

}  // namespace ftrl
