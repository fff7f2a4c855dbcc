#include "trainer.h"

#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <thread>

namespace ftrl {

using timer = std::chrono::steady_clock;
static double seconds_since(timer::time_point t0) {
  return std::chrono::duration<double>(timer::now() - t0).count();
}

// ---------------- offline: data in memory, seeded shuffle per epoch ----------------

FtrlOffline::FtrlOffline(const config_options &opt)
    : model_ptr(make_model(opt)), n_epochs(opt.epoch), n_threads(opt.thread_num), seed_(opt.seed),
      sched_(opt.batch_size, opt.batch_ramp) {
  // the files go straight into CSR (csr_reader.h); the Sample-based readers exist for callers
  // that use one_epoch(std::vector<Sample>&, ...) as the reference's tests do
  train_data_loader = std::make_unique<Reader>(opt.file_type);
  std::printf("Loading data from file: %s\n", opt.train_path.c_str());
  const auto t0 = timer::now();
  train_csr_ = load_csr(opt.train_path, opt.file_type, n_threads);
  std::printf("Total number of samples loaded: %zu\nparsing data time: %.4lfs\n",
              train_csr_.n_rows(), seconds_since(t0));
  if (!opt.eval_path.empty()) {
    eval_data_loader = std::make_unique<Reader>(opt.file_type);
    eval_csr_ = load_csr(opt.eval_path, opt.file_type, n_threads);
    has_eval_ = true;
  }
}

FtrlOffline::~FtrlOffline() {
  for (auto &b : ring_) model_ptr->unpin_block(b);
}

bool FtrlOffline::ensure_ring() {
  if (ring_tried_) return !ring_.empty();
  ring_tried_ = true;
  if (std::getenv("FTRL_NO_PINNED_RING")) return false;  // (A/B aid: the copying path)
  ring_.resize(kRing);
  ring_seq_.assign(kRing, 0);
  for (int i = 0; i < kRing; i++)
    if (!model_ptr->pin_block(ring_[i])) {  // no page-locked memory to be had: the copying path
      for (int j = 0; j < i; j++) model_ptr->unpin_block(ring_[j]);
      ring_.clear();
      return false;
    }
  return true;
}

// One pass over a CSR file image: training visits the rows in a seeded shuffle, block by block;
// evaluation in file order.  Mean of loss(y, logit) over all rows.
double FtrlOffline::csr_epoch(const CsrData &d, bool train) {
  const size_t total = d.n_rows();
  if (total == 0) return 0.0;
  std::vector<int> indices;
  if (train) {
    indices.resize(total);
    std::iota(indices.begin(), indices.end(), 0);
    std::shuffle(indices.begin(), indices.end(), std::mt19937_64{seed_ + (++epoch_no_)});
  }
  double total_loss = 0.0;
  CsrBlock blk;
  const bool ring = train && ensure_ring();
  int slot = 0;
  size_t pos = 0;
  while (pos < total) {
    const size_t rows = std::min<size_t>(train ? sched_.next_block_rows() : sched_.max_block_rows(), total - pos);
    // training is pipelined: this block is uploaded and grouped while the previous ones train
    // and the next one is gathered here -- straight into page-locked memory when there is a ring
    // entry it fits (the device then pulls it from there; the entry is reused once it has)
    if (train && ring && rows < ring_[slot].row_ptr.capacity() &&
        d.gather_nnz(indices.data() + pos, rows) <= ring_[slot].feat.capacity()) {
      while (model_ptr->blocks_pulled() < ring_seq_[slot]) std::this_thread::yield();
      d.gather(indices.data() + pos, rows, ring_[slot], n_threads);
      ring_seq_[slot] = model_ptr->train_block_pinned(ring_[slot]);
      slot = (slot + 1) % kRing;
    } else if (train) {
      d.gather(indices.data() + pos, rows, blk, n_threads);
      model_ptr->train_block_async(blk);
    } else {
      d.slice(pos, pos + rows, blk);
      total_loss += model_ptr->predict_block(blk, false);
    }
    if (train) sched_.consumed(static_cast<int>(rows));
    pos += rows;
  }
  if (train) total_loss = model_ptr->train_flush();
  return total_loss / static_cast<double>(total);
}

void FtrlOffline::train() {
  for (int i = 1; i <= n_epochs; i++) {
    const auto t0 = timer::now();
    const double train_loss = csr_epoch(train_csr_, true);
    std::printf("epoch %d train time: %.4lfs, train loss: %.4lf\n", i, seconds_since(t0), train_loss);
    if (has_eval_) evaluate(i);
  }
}

void FtrlOffline::evaluate(int epoch) {
  const auto t0 = timer::now();
  const double eval_loss = csr_epoch(eval_csr_, false);
  std::printf("epoch %d eval time: %.4lfs, eval loss: %.4lf\n", epoch, seconds_since(t0), eval_loss);
}

// One pass over `samples`: training visits them in a seeded shuffle (the reference shuffles from
// std::random_device, ftrl_offline.cpp:67-71), block by block; evaluation in order.  Returns the
// mean of loss(y, logit) over all rows, as ftrl_offline.cpp:101-102.
double FtrlOffline::one_epoch(std::vector<Sample> &samples, bool train, bool /*use_pool*/) {
  const size_t total = samples.size();
  if (total == 0) return 0.0;
  std::vector<int> indices(total);
  std::iota(indices.begin(), indices.end(), 0);
  if (train) std::shuffle(indices.begin(), indices.end(), std::mt19937_64{seed_ + (++epoch_no_)});
  double total_loss = 0.0;
  CsrBlock blk;
  size_t pos = 0;
  while (pos < total) {
    const size_t rows = std::min<size_t>(train ? sched_.next_block_rows() : sched_.max_block_rows(), total - pos);
    blk.clear();
    for (size_t r = 0; r < rows; r++) blk.push(samples[indices[pos + r]]);
    if (train) model_ptr->train_block_async(blk); else total_loss += model_ptr->predict_block(blk, false);
    if (train) sched_.consumed(static_cast<int>(rows));
    pos += rows;
  }
  if (train) total_loss = model_ptr->train_flush();
  return total_loss / static_cast<double>(total);
}

// ---------------- online: streaming from the file, file order ----------------

FtrlOnline::FtrlOnline(const config_options &opt)
    : model_ptr(make_model(opt)), n_epochs(opt.epoch), cmd_(opt.cmd),
      sched_(opt.batch_size, opt.batch_ramp), parser_(make_parser(opt.file_type)) {
  if (!cmd_) {
    train_ifs_.open(opt.train_path, std::ios::in | std::ios::binary);
    if (!train_ifs_.good()) {
      std::fprintf(stderr, "open file <%s> error. \n", opt.train_path.c_str());
      std::exit(EXIT_FAILURE);
    }
    if (!opt.eval_path.empty()) {
      eval_ifs_.open(opt.eval_path, std::ios::in | std::ios::binary);
      if (!eval_ifs_.good()) {
        std::fprintf(stderr, "open file <%s> error. \n", opt.eval_path.c_str());
        std::exit(EXIT_FAILURE);
      }
      has_eval_ = true;
    }
  }
}

// Reads the stream to its end, parsing rows into blocks (parse time is inside the timed region,
// as in the reference's online mode) and feeding them to the engine in file order.
double FtrlOnline::run_file(std::ifstream &ifs, bool train) {
  CsrBlock blk;
  std::string line;
  Sample sample;
  double sum = 0.0;
  unsigned long long rows = 0, line_num = 0;
  bool more = true;
  while (more) {
    const int want = train ? sched_.next_block_rows() : sched_.max_block_rows();
    blk.clear();
    while (blk.n_rows() < want) {
      if (!std::getline(ifs, line)) { more = false; break; }
      parser_->parse(line, sample);
      blk.push(sample);
      if (++line_num % 1000000 == 0) std::printf("%llu lines finished...\n", line_num);
    }
    if (blk.n_rows() == 0) break;
    if (train) model_ptr->train_block_async(blk); else sum += model_ptr->predict_block(blk, false);
    if (train) sched_.consumed(blk.n_rows());
    rows += blk.n_rows();
  }
  if (train) sum = model_ptr->train_flush();  // parsing of block t+1 overlapped the training of t
  ifs.clear();
  ifs.seekg(0, std::ios::beg);
  loss_sum_ = sum;
  loss_rows_ = rows;
  return rows ? sum / static_cast<double>(rows) : 0.0;
}

double FtrlOnline::get_loss() {
  const double r = loss_rows_ ? loss_sum_ / static_cast<double>(loss_rows_) : 0.0;
  loss_sum_ = 0.0;
  loss_rows_ = 0;
  return r;
}

void FtrlOnline::train() {
  if (cmd_) return;  // stdin mode is a TODO stub in the reference too (ftrl_online.cpp:55-57)
  for (int i = 1; i <= n_epochs; i++) {
    const auto t0 = timer::now();
    run_file(train_ifs_, true);
    const double train_loss = get_loss();
    std::printf("epoch %d train time: %.4lfs, train loss: %.4lf\n", i, seconds_since(t0), train_loss);
    if (has_eval_) evaluate(i);
  }
}

void FtrlOnline::evaluate(int epoch) {
  const auto t0 = timer::now();
  run_file(eval_ifs_, false);
  const double eval_loss = get_loss();
  std::printf("epoch %d eval time: %.4lfs, eval loss: %.4lf\n", epoch, seconds_since(t0), eval_loss);
}

}  // namespace ftrl
