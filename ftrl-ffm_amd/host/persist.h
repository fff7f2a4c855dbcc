// persist.h -- model files in the reference's formats (reference src/model/ffm.cpp:138-200,
// src/model/lr.cpp:26-39, src/compression/compress.cpp:15-51), plus a sidecar for the FTRL
// accumulators so that a checkpoint is actually resumable (the reference saves only w, and its
// first train() after a load recomputes w from zero (n,z)).
//
//   text        : "bias\n", n_feats lines of lin_w, n_feats lines of row_len space-separated floats
//   compressed  : ONE zstd frame holding float32 [bias, lin_w[n_feats], vec_w row-major]
//   state (.nz) : one zstd frame of float32 [bias_n, bias_z, lin_n[], lin_z[], vec_n[], vec_z[]]
//
// Everything streams: a writer takes the floats chunk by chunk (the model hands over a few tens of
// thousands of records at a time, pulled from HBM with ffm_engine_get_rows), a reader hands them
// back chunk by chunk -- the 33 M-feature headline model is 82 GB of w, which no host vector holds.
// The compressed frame is produced with zstd's streaming API and a pledged source size, so its
// header carries the content size the reference's one-shot loader asks for (ZSTD_getFrameContentSize,
// compress.cpp:33-40): files interchange both ways.
//
// zstd is taken from the system's libzstd.so.1 at run time (dlopen); if it is absent the compressed
// calls throw std::runtime_error and the text format still works.
#pragma once
#include <cstdio>
#include <fstream>
#include <memory>
#include <string>
#include <vector>

namespace ftrl {

bool zstd_available();

// One zstd frame of `total_floats` float32 values, fed in any number of pieces.
class FloatFrameWriter {
 public:
  FloatFrameWriter(const std::string &path, size_t total_floats, int level);
  ~FloatFrameWriter();
  void write(const float *p, size_t n);
  void finish();  // flushes the frame; prints the reference's "saving to ..." line

 private:
  struct Impl;
  std::unique_ptr<Impl> d_;
};

// Reads such a frame (streaming or one-shot made) piece by piece.
class FloatFrameReader {
 public:
  explicit FloatFrameReader(const std::string &path);
  ~FloatFrameReader();
  size_t total_floats() const;        // from the frame header
  size_t read(float *p, size_t n);    // up to n floats; 0 at the end of the frame

 private:
  struct Impl;
  std::unique_ptr<Impl> d_;
};

// The text format, line by line (ffm.cpp:163-200).
class TextModelWriter {
 public:
  explicit TextModelWriter(const std::string &path);
  void scalar(float v);                               // "bias\n" / one lin_w line: ostream default precision
  void rows(const float *p, size_t n_rows, size_t row_len);  // shortest round-trip floats, space separated
  void finish();

 private:
  std::ofstream f_;
  std::string path_;
};
class TextModelReader {
 public:
  explicit TextModelReader(const std::string &path);
  float scalar();
  void rows(float *p, size_t n_rows, size_t row_len);

 private:
  std::ifstream f_;
  std::string path_, line_;
};

}  // namespace ftrl
