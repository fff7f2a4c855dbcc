// types.h -- row types of the host interface; same shapes as the reference's
// src/include/utils/types.h:18-25 and src/include/data/sample.h:6-9 so callers port unchanged.
#pragma once
#include <cstdint>
#include <tuple>
#include <vector>

typedef std::tuple<int, int, float> feat;  // (field, feat, value)
typedef std::vector<feat> feat_vec;

enum class ModelType : uint8_t { LR = 0, FM = 1, FFM = 2 };

struct Sample {
  feat_vec x;
  int y;
};

// A block of rows in the engine's CSR wire format (include/ffm_engine.h).
struct CsrBlock {
  std::vector<int32_t> row_ptr{0}, field, feat, label;
  std::vector<float> val;
  int32_t n_rows() const { return static_cast<int32_t>(row_ptr.size()) - 1; }
  void clear() {
    row_ptr.assign(1, 0);
    field.clear(); feat.clear(); label.clear(); val.clear();
  }
  void push(const Sample &s) {
    for (const auto &[f, i, v] : s.x) { field.push_back(f); feat.push_back(i); val.push_back(v); }
    row_ptr.push_back(static_cast<int32_t>(feat.size()));
    label.push_back(s.y);
  }
};
