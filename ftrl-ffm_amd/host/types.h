// types.h -- row types of the host interface; same shapes as the reference's
// src/include/utils/types.h:18-25 and src/include/data/sample.h:6-9 so callers port unchanged.
#pragma once
#include <cstdint>
#include <cstdlib>
#include <new>
#include <tuple>
#include <vector>

typedef int8_t int8;
typedef int16_t int16;
typedef int32_t int32;
typedef int64_t int64;
typedef uint8_t uint8;
typedef uint16_t uint16;
typedef uint32_t uint32;
typedef uint64_t uint64;

typedef std::tuple<int, int, float> feat;  // (field, feat, value)
typedef std::vector<feat> feat_vec;

enum class ModelType : uint8_t { LR = 0, FM = 1, FFM = 2 };

struct Sample {
  feat_vec x;
  int y;
};

// Whole pages per array: a block's arrays can then be page-locked one by one (hipHostRegister
// works on pages: two small arrays sharing a heap page cannot both be registered).
template <typename T>
struct PageAllocator {
  using value_type = T;
  PageAllocator() = default;
  template <typename U> PageAllocator(const PageAllocator<U> &) {}
  T *allocate(size_t n) {
    const size_t bytes = (n * sizeof(T) + 4095) & ~static_cast<size_t>(4095);
    void *p = std::aligned_alloc(4096, bytes ? bytes : 4096);
    if (!p) throw std::bad_alloc();
    return static_cast<T *>(p);
  }
  void deallocate(T *p, size_t) { std::free(p); }
  template <typename U> bool operator==(const PageAllocator<U> &) const { return true; }
  template <typename U> bool operator!=(const PageAllocator<U> &) const { return false; }
};

// A block of rows in the engine's CSR wire format (include/ffm_engine.h).
struct CsrBlock {
  std::vector<int32_t, PageAllocator<int32_t>> row_ptr{0}, field, feat, label;
  std::vector<float, PageAllocator<float>> val;
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
