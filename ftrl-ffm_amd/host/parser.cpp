#include "parser.h"

#include <cstdlib>
#include <iostream>
#include <stdexcept>

namespace ftrl {
namespace {

// One "a:b[:c]" token scanner over [p, end).  Mirrors the accept/reject behaviour of the
// reference's find_first_of walk: a missing ':' or an empty trailing number is an error.
[[noreturn]] void bad_line(const std::string &line) {
  std::cout << "wrong input: " << line << std::endl;
  throw std::out_of_range(line);
}

int parse_int(const char *b, const char *e, const std::string &line) {
  if (b >= e) bad_line(line);
  char *stop = nullptr;
  const long v = std::strtol(b, &stop, 10);
  if (stop == b) throw std::invalid_argument("stoi");  // what std::stoi throws
  return static_cast<int>(v);
}

float parse_float(const char *b, const char *e, const std::string &line) {
  if (b >= e) bad_line(line);
  char *stop = nullptr;
  const float v = std::strtof(b, &stop);
  if (stop == b) throw std::invalid_argument("stof");
  return v;
}

void parse_line(const std::string &line, Sample &sample, bool has_field) {
  sample.x.clear();
  const char *p = line.data(), *end = p + line.size();
  while (p < end && *p == ' ') p++;
  const char *tok_end = p;
  while (tok_end < end && *tok_end != ' ') tok_end++;
  sample.y = parse_int(p, tok_end, line) > 0 ? 1 : 0;
  p = tok_end;
  while (true) {
    while (p < end && *p == ' ') p++;
    if (p >= end) break;
    tok_end = p;
    while (tok_end < end && *tok_end != ' ') tok_end++;
    const char *c1 = p;
    while (c1 < tok_end && *c1 != ':') c1++;
    if (c1 >= tok_end) bad_line(line);
    int field = 0, feature;
    const char *vbeg;
    if (has_field) {
      field = parse_int(p, c1, line);
      const char *c2 = c1 + 1;
      if (c2 >= end) bad_line(line);
      while (c2 < tok_end && *c2 != ':') c2++;
      if (c2 >= tok_end) bad_line(line);
      feature = parse_int(c1 + 1, c2, line);
      vbeg = c2 + 1;
    } else {
      feature = parse_int(p, c1, line);
      vbeg = c1 + 1;
    }
    if (vbeg >= end) bad_line(line);
    const float value = parse_float(vbeg, tok_end, line);
    if (value != 0.0f) sample.x.emplace_back(field, feature, value);
    p = tok_end;
  }
}

}  // namespace

void LibsvmParser::parse(const std::string &line, Sample &sample) { parse_line(line, sample, false); }
void FFMParser::parse(const std::string &line, Sample &sample) { parse_line(line, sample, true); }

}  // namespace ftrl
