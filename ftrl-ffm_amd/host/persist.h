// persist.h -- model files in the reference's formats (reference src/model/ffm.cpp:138-200,
// src/model/lr.cpp:26-39, src/compression/compress.cpp:15-51), plus a sidecar for the FTRL
// accumulators so that a checkpoint is actually resumable (the reference saves only w, and its
// first train() after a load recomputes w from zero (n,z)).
//
//   text        : "bias\n", n_feats lines of lin_w, n_feats lines of row_len space-separated floats
//   compressed  : ONE zstd frame holding float32 [bias, lin_w[n_feats], vec_w row-major]
//   state (.nz) : "FTRLNZ1\n" + float32 [bias_n, bias_z, lin_n[], lin_z[], vec_n[], vec_z[]], zstd
//
// zstd is taken from the system's libzstd.so.1 at run time (dlopen); if it is absent the compressed
// calls throw std::runtime_error and the text format still works.
#pragma once
#include <string>
#include <vector>

namespace ftrl {

struct ModelWeights {
  float bias = 0.0f;
  std::vector<float> lin_w;  // [n_feats]
  std::vector<float> vec_w;  // [n_feats * row_len], row-major
};

void write_text_model(const std::string &path, const ModelWeights &w, size_t n_feats, size_t row_len);
ModelWeights read_text_model(const std::string &path, size_t n_feats, size_t row_len);
void write_compressed_model(const std::string &path, const ModelWeights &w, int compress_level);
ModelWeights read_compressed_model(const std::string &path, size_t n_feats, size_t row_len);

// one zstd frame around an arbitrary float array (used by the .nz sidecar)
void write_compressed_floats(const std::string &path, const std::vector<float> &v, int level);
std::vector<float> read_compressed_floats(const std::string &path);
bool zstd_available();

}  // namespace ftrl
