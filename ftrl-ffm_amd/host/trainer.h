// trainer.h -- FtrlOffline / FtrlOnline / Evaluator over the block engine.
//
// Replaces the reference's per-sample thread pool and producer/consumer queue
// (src/task/ftrl_offline.cpp:63-103, src/task/ftrl_online.cpp:42-80, src/eval/evaluate.cpp:23-49,
// src/include/concurrent/*) with a mini-batch scheduler: rows are packed into CSR blocks on the
// host and handed to the engine one block at a time.  Same constructors, train(), evaluate(),
// has_zero_weights(), model_ptr and printed lines as the reference.
//
// Block-size ramp: block t holds min(batch_size, max(1, rows_seen / batch_ramp)) rows, so the
// staleness of the weights a row sees never exceeds 1/batch_ramp of the rows already learned
// from -- that keeps the epoch logloss within 1e-4 of the reference's strictly sequential loop
// (DESIGN.md "Batch semantics"; batch_ramp = 0 turns the ramp off).
#pragma once
#include <fstream>
#include <memory>
#include <string>
#include <vector>

#include "ftrl_model.h"
#include "csr_reader.h"
#include "reader.h"

namespace ftrl {

class BlockScheduler {
 public:
  BlockScheduler(int batch_size, int batch_ramp) : batch_(batch_size), ramp_(batch_ramp) {}
  int next_block_rows() const {
    if (ramp_ <= 0) return batch_;
    const long long r = seen_ / ramp_;
    return static_cast<int>(std::min<long long>(batch_, std::max<long long>(1, r)));
  }
  int max_block_rows() const { return batch_; }  // evaluation has no staleness: full blocks
  void consumed(int rows) { seen_ += rows; }
  long long rows_seen() const { return seen_; }

 private:
  int batch_, ramp_;
  long long seen_ = 0;
};

class FtrlOffline {
 public:
  explicit FtrlOffline(const config_options &opt);
  ~FtrlOffline();
  void train();
  void evaluate(int epoch = 0);
  double one_epoch(std::vector<Sample> &samples, bool train, bool use_pool);
  bool has_zero_weights() { return model_ptr->has_zero_weights(); }

  std::unique_ptr<FtrlModel> model_ptr;

 private:
  int n_epochs, n_threads;
  uint64_t seed_;
  int epoch_no_ = 0;
  BlockScheduler sched_;
  std::unique_ptr<Reader> train_data_loader, eval_data_loader;  // API parity (data stays empty
                                                                // unless load_samples() is called)
  CsrData train_csr_, eval_csr_;                                // what train()/evaluate() walk
  bool has_eval_ = false;
  double csr_epoch(const CsrData &d, bool train);
  // Training blocks are gathered straight into a ring of page-locked blocks which the device
  // pulls from (FtrlModel::train_block_pinned): no host copy, three blocks in flight.
  static constexpr int kRing = 6;
  std::vector<CsrBlock> ring_;
  std::vector<long long> ring_seq_;  // ordinal of the block each ring entry carried last
  bool ring_tried_ = false;
  bool ensure_ring();
};

class FtrlOnline {
 public:
  explicit FtrlOnline(const config_options &opt);
  void train();
  void evaluate(int epoch = 0);
  double get_loss();
  bool has_zero_weights() { return model_ptr->has_zero_weights(); }

  std::shared_ptr<FtrlModel> model_ptr;

 private:
  double run_file(std::ifstream &ifs, bool train);
  int n_epochs;
  bool cmd_;
  BlockScheduler sched_;
  std::unique_ptr<Parser> parser_;
  std::ifstream train_ifs_, eval_ifs_;
  bool has_eval_ = false;
  double loss_sum_ = 0.0;
  unsigned long long loss_rows_ = 0;
};

}  // namespace ftrl
