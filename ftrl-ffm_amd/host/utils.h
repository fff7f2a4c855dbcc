// utils.h -- the helpers of the reference's `struct utils` that callers and tests use
// (src/include/utils/utils.h:15-104): sgn (sgn(0) = -1), sigmoid, Gaussian weight init, the
// exact-zero test, wall-clock timing.  Own implementation; init_weights draws from the engine's
// SEEDED counter-based generator (ffm_engine_init_weights_host: what the device stores, bit for
// bit) instead of a fresh std::random_device per weight (~32 us each in the reference).
#pragma once
#include <algorithm>
#include <chrono>
#include <cmath>
#include <string>
#include <vector>

#include "../../include/ffm_engine.h"
#include "types.h"

struct utils {
  template <typename T>
  static inline T sgn(T x) { return x > 0 ? 1 : -1; }  // utils.h:15-18

  template <typename T>
  static inline T sigmoid(T x) { return 1 / (1 + std::exp(-x)); }  // utils.h:20-23

  // `num` weights ~ N(mean, stddev); stream: which seeded sequence (every call site its own)
  template <typename T>
  static std::vector<T> init_weights(std::size_t num, T mean, T stddev, uint64_t seed = 42) {
    std::vector<float> tmp(num);
    ffm_engine_init_weights_host(seed, static_cast<float>(mean), static_cast<float>(stddev), 0, 0,
                                 static_cast<int64_t>(num), tmp.data());
    return std::vector<T>(tmp.begin(), tmp.end());
  }
  template <typename T>
  static std::vector<std::vector<T>> init_weights(std::size_t num, int n_factors, T mean, T stddev,
                                                  uint64_t seed = 42) {
    std::vector<float> tmp(num * static_cast<std::size_t>(n_factors));
    ffm_engine_init_weights_host(seed, static_cast<float>(mean), static_cast<float>(stddev), 1, 0,
                                 static_cast<int64_t>(tmp.size()), tmp.data());
    std::vector<std::vector<T>> weights(num);
    for (std::size_t i = 0; i < num; i++)
      weights[i].assign(tmp.begin() + i * n_factors, tmp.begin() + (i + 1) * n_factors);
    return weights;
  }
  template <typename T>
  static std::vector<std::vector<T>> init_weights(std::size_t num, int n_fields, int n_factors, T mean,
                                                  T stddev, uint64_t seed = 42) {
    return init_weights(num, n_factors * n_fields, mean, stddev, seed);
  }

  template <typename T>
  static bool has_zero_weights(std::vector<T> &weights) {
    return std::any_of(weights.cbegin(), weights.cend(), [](T w) { return w == 0.0; });
  }
  template <typename T>
  static bool has_zero_weights(std::vector<std::vector<T>> &weights) {
    for (auto &ws : weights)
      if (has_zero_weights(ws)) return true;
    return false;
  }

  using clock_time = std::chrono::time_point<std::chrono::steady_clock>;
  static double compute_time(const clock_time &start_time) {  // seconds since start_time
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - start_time).count();
  }
};
