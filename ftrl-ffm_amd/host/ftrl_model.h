// ftrl_model.h -- host-side mirror of the reference's model interface
// (src/include/model/ftrl_model.h:14-51, ffm.h:11-33, fm.h:11-28, lr.h:10-17) over the C ABI of
// include/ffm_engine.h.  Same class names, same train()/predict() meaning and return values, same
// public members; the arithmetic happens in HBM behind an ffm_engine handle.
//
//   float train(feat_vec&, int)   one reference train(): returns the pre-update logit and erases
//                                 out-of-range entries from the caller's vector, as the reference.
//   float predict(feat_vec&, bool)
//   train_block / predict_block   the same for a block of rows (what the trainers call).
//
// bias / lin_w / vec_w are host mirrors of the device weights: call pull_weights() to refresh them
// after training, push_weights() after editing them (the reference's tests poke them directly).
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "../../include/ffm_engine.h"
#include "cmd_option.h"
#include "types.h"

namespace ftrl {

class FtrlModel {
 public:
  explicit FtrlModel(const config_options &opt, int model_type);
  virtual ~FtrlModel();
  FtrlModel(const FtrlModel &) = delete;
  FtrlModel &operator=(const FtrlModel &) = delete;

  virtual float train(feat_vec &features, int label);
  virtual float predict(feat_vec &features, bool output_prob);
  virtual void remove_out_range(feat_vec &feats);

  // Block entry points used by FtrlOffline / FtrlOnline.  Return sum of loss(y, logit).
  double train_block(const CsrBlock &blk, float *logit_out = nullptr);
  double predict_block(const CsrBlock &blk, bool output_prob, float *out = nullptr);
  // Pipelined training for callers that need only the loss (the trainers): queue blocks, then
  // train_flush() returns the sum of loss(y, logit) over all blocks queued since the last flush.
  // The block's arrays may be reused as soon as train_block_async returns.
  void train_block_async(const CsrBlock &blk);
  double train_flush();
  // The same without the host copy, for blocks gathered in page-locked memory: pin_block() sizes a
  // block's arrays for one engine call and page-locks them (false: could not, use the copying
  // path); train_block_pinned() hands such a block over -- it must then stay untouched until
  // blocks_pulled() has reached the number it returns (blocks handed over so far).  A block that
  // does not fit one engine call goes through the copying path instead.
  bool pin_block(CsrBlock &blk);
  void unpin_block(CsrBlock &blk);
  long long train_block_pinned(const CsrBlock &blk);
  long long blocks_pulled();

  // Model files in the reference's formats (ffm.cpp:138-200, lr.cpp:26-39); available for every
  // model type here (the reference has none for FM).  save_state/load_state add the FTRL
  // accumulators, which make a checkpoint resumable.
  void save_model(const std::string &file_name);
  void load_model(const std::string &file_name);
  void save_compressed_model(const std::string &file_name, int compress_level);
  void load_compressed_model(const std::string &file_name);
  void save_state(const std::string &file_name, int compress_level = 3);
  void load_state(const std::string &file_name);

  void pull_weights();  // device -> bias / lin_w / vec_w
  void push_weights();  // bias / lin_w / vec_w -> device
  bool has_zero_weights();

  ModelType model_type;
  float bias = 0.0f;
  std::vector<float> lin_w;
  std::vector<std::vector<float>> vec_w;  // [n_feats][row_len]; empty for LR

  ffm_engine *engine() { return eng_; }
  int64_t row_len() const { return row_len_; }

 protected:
  int n_feats, n_fields, n_factors;
  int64_t row_len_ = 0;
  ffm_engine *eng_ = nullptr;
  CsrBlock one_;  // scratch for the one-row shims
  // engine capacities chosen at construction; blocks beyond max_nnz_ are split into several
  // engine calls (each still a block in row order), a single row beyond max_row_nnz_ is an error
  int max_rows_ = 0, max_nnz_ = 0, max_row_nnz_ = 0;
  long long handed_over_ = 0;  // blocks passed to the engine's pipelined entry points so far
  CsrBlock part_;
  // calls `fn(sub-block)` for consecutive row ranges of blk that fit the engine
  template <typename Fn> void for_each_fitting(const CsrBlock &blk, Fn fn);
};

class LR : public FtrlModel {
 public:
  explicit LR(const config_options &opt) : FtrlModel(opt, FFM_MODEL_LR) {}
};
class FM : public FtrlModel {
 public:
  explicit FM(const config_options &opt) : FtrlModel(opt, FFM_MODEL_FM) {}
};
class FFM : public FtrlModel {
 public:
  explicit FFM(const config_options &opt) : FtrlModel(opt, FFM_MODEL_FFM) {}
  void remove_out_range(feat_vec &feats) override;  // also filters the field (ffm.cpp:30-36)
};

std::unique_ptr<FtrlModel> make_model(const config_options &opt);  // throws std::invalid_argument

// loss(int y, double logit), src/include/eval/loss.h:8-12
double loss(int y, double logit);

}  // namespace ftrl
