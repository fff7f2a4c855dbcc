#include "csr_stream.h"

#include <fcntl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <stdexcept>

namespace ftrl {

CsrStream::CsrStream(const std::string &path, const std::string &file_type, int n_threads)
    : has_field_(file_type == "libffm"), n_threads_(std::max(1, n_threads)) {
  fd_ = open(path.c_str(), O_RDONLY);
  if (fd_ < 0) {
    std::fprintf(stderr, "open file <%s> error. \n", path.c_str());  // pc_task.cpp:6-11
    std::exit(EXIT_FAILURE);
  }
  struct stat st {};
  fstat(fd_, &st);
  len_ = static_cast<size_t>(st.st_size);
  if (len_) {
    void *p = mmap(nullptr, len_, PROT_READ, MAP_PRIVATE, fd_, 0);
    if (p == MAP_FAILED) throw std::runtime_error("mmap failed: " + path);
    base_ = static_cast<const char *>(p);
    madvise(const_cast<char *>(base_), len_, MADV_SEQUENTIAL);
  }
  ring_.resize(static_cast<size_t>(2 * n_threads_ + 2));
  start_workers();
}

CsrStream::~CsrStream() {
  stop_workers();
  if (base_) munmap(const_cast<char *>(base_), len_);
  if (fd_ >= 0) close(fd_);
}

void CsrStream::start_workers() {
  stop_ = false;
  for (int i = 0; i < n_threads_; i++) threads_.emplace_back([this] { worker(); });
}
void CsrStream::stop_workers() {
  {
    std::lock_guard<std::mutex> lock(mu_);
    stop_ = true;
  }
  cv_work_.notify_all();
  for (auto &t : threads_) t.join();
  threads_.clear();
}

// Under the lock: the next <= kChunkLines whole lines (a memchr walk: GB/s, a chunk is a few MB).
bool CsrStream::claim(size_t *id, const char **b, const char **e) {
  if (scan_pos_ >= len_) return false;
  const char *p = base_ + scan_pos_, *end = base_ + len_;
  size_t lines = 0;
  while (p < end && lines < kChunkLines) {
    const char *nl = static_cast<const char *>(std::memchr(p, '\n', static_cast<size_t>(end - p)));
    p = nl ? nl + 1 : end;
    lines++;
  }
  *id = next_id_++;
  *b = base_ + scan_pos_;
  *e = p;
  scan_pos_ = static_cast<size_t>(p - base_);
  return true;
}

void CsrStream::worker() {
  for (;;) {
    size_t id = 0;
    const char *b = nullptr, *e = nullptr;
    {
      std::unique_lock<std::mutex> lock(mu_);
      // at most ring_.size() chunks between the consumer and the scanner
      cv_work_.wait(lock, [&] { return stop_ || (scan_pos_ < len_ && next_id_ < consume_id_ + ring_.size()); });
      if (stop_) return;
      if (!claim(&id, &b, &e)) continue;
    }
    Chunk &c = ring_[id % ring_.size()];
    c.part.clear();
    std::string err;
    try {
      parse_csr_range(b, e, has_field_, c.part);
    } catch (const std::out_of_range &ex) {
      err = ex.what();
    }
    c.row_ptr.resize(c.part.nnz.size() + 1);
    c.row_ptr[0] = 0;
    for (size_t r = 0; r < c.part.nnz.size(); r++) c.row_ptr[r + 1] = c.row_ptr[r] + c.part.nnz[r];
    {
      std::lock_guard<std::mutex> lock(mu_);
      c.id = id;
      c.ready = true;
      if (!err.empty() && error_.empty()) error_ = err;
    }
    cv_ready_.notify_all();
  }
}

bool CsrStream::acquire_chunk(const CsrPart **part, const std::vector<int64_t> **row_ptr) {
  if (consume_row_ != 0) throw std::logic_error("acquire_chunk after a partial next()");
  std::unique_lock<std::mutex> lock(mu_);
  if (consume_id_ >= next_id_ && scan_pos_ >= len_) return false;
  cv_ready_.wait(lock, [&] {
    const Chunk &k = ring_[consume_id_ % ring_.size()];
    return !error_.empty() || (k.ready && k.id == consume_id_);
  });
  if (!error_.empty()) throw std::out_of_range(error_);
  const Chunk &c = ring_[consume_id_ % ring_.size()];
  *part = &c.part;
  *row_ptr = &c.row_ptr;
  return true;
}

void CsrStream::release_chunk() {
  {
    std::lock_guard<std::mutex> lock(mu_);
    Chunk &c = ring_[consume_id_ % ring_.size()];
    delivered_ += c.part.nnz.size();
    c.ready = false;
    consume_id_++;
  }
  cv_work_.notify_all();
}

size_t CsrStream::next(size_t want, CsrBlock &out, size_t max_nnz, bool fixed_capacity) {
  out.clear();
  size_t got = 0;
  while (got < want) {
    Chunk *c = nullptr;
    {
      std::unique_lock<std::mutex> lock(mu_);
      if (consume_id_ >= next_id_ && scan_pos_ >= len_) break;  // everything assigned has been consumed
      cv_ready_.wait(lock, [&] {
        const Chunk &k = ring_[consume_id_ % ring_.size()];
        return !error_.empty() || (k.ready && k.id == consume_id_);
      });
      if (!error_.empty()) throw std::out_of_range(error_);
      c = &ring_[consume_id_ % ring_.size()];
    }
    const size_t rows = c->part.nnz.size();
    if (consume_row_ >= rows) {  // (a chunk of blank lines)
      {
        std::lock_guard<std::mutex> lock(mu_);
        c->ready = false;
        consume_id_++;
        consume_row_ = 0;
      }
      cv_work_.notify_all();
      continue;
    }
    const int64_t e0 = c->row_ptr[consume_row_];
    auto entries = [&](size_t t) { return static_cast<size_t>(c->row_ptr[consume_row_ + t] - e0); };
    size_t take = std::min(want - got, rows - consume_row_);
    bool cut = false;  // the entry budget ends the block here
    while (take > 0 && out.feat.size() + entries(take) > max_nnz) { take--; cut = true; }
    if (take == 0) {
      if (got > 0) break;
      // a single row beyond the budget: a page-locked ring entry's vectors are registered at their
      // size -- growing them would free pages the runtime still maps -- so the row is refused here
      // instead of being written (the caller says which kind of block `out` is)
      if (fixed_capacity)
        throw std::length_error("a row has more entries than a block can hold (max_nnz)");
      take = 1;  // (a growable block: the row goes out alone and the model splits or rejects it)
    }
    const int64_t e1 = c->row_ptr[consume_row_ + take];
    const size_t base = out.feat.size();
    out.field.insert(out.field.end(), c->part.field.begin() + e0, c->part.field.begin() + e1);
    out.feat.insert(out.feat.end(), c->part.feat.begin() + e0, c->part.feat.begin() + e1);
    out.val.insert(out.val.end(), c->part.val.begin() + e0, c->part.val.begin() + e1);
    out.label.insert(out.label.end(), c->part.label.begin() + consume_row_, c->part.label.begin() + consume_row_ + take);
    for (size_t r = 0; r < take; r++)
      out.row_ptr.push_back(static_cast<int32_t>(base + static_cast<size_t>(c->row_ptr[consume_row_ + r + 1] - e0)));
    got += take;
    consume_row_ += take;
    if (consume_row_ == rows) {
      {
        std::lock_guard<std::mutex> lock(mu_);
        c->ready = false;
        consume_id_++;
        consume_row_ = 0;
      }
      cv_work_.notify_all();
    }
    if (cut) break;
  }
  delivered_ += got;
  return got;
}

void CsrStream::rewind() {
  stop_workers();
  for (auto &c : ring_) { c.ready = false; c.part.clear(); }
  scan_pos_ = 0;
  next_id_ = consume_id_ = consume_row_ = 0;
  error_.clear();
  start_workers();
}

}  // namespace ftrl
