// csr_reader.h -- the file straight into the engine's CSR wire format: mmap, one byte range per
// thread cut at line starts (the partitioning of reference src/data/reader.cpp:22-48), numbers
// parsed in place with std::from_chars -- no per-row std::vector<std::tuple>.  Same acceptance rules
// as the line parsers (parser.cpp:11-103): label > 0 -> 1, zero values dropped, libsvm field 0,
// malformed token -> std::out_of_range.  This is what feeds the GPU at millions of rows/s; the
// Sample-based Reader stays for API parity.
#pragma once
#include <cstdint>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "types.h"

namespace ftrl {

// std::vector whose resize() leaves new elements uninitialised: the loader sizes the whole file's
// arrays once and its threads then fill disjoint ranges -- a value-initialising resize would first
// zero every page from ONE thread (as slow as the parse itself).
template <class T>
struct default_init_allocator : std::allocator<T> {
  template <class U> struct rebind { using other = default_init_allocator<U>; };
  using std::allocator<T>::allocator;
  template <class U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
  template <class U, class... A> void construct(U *p, A &&...a) { ::new (static_cast<void *>(p)) U(std::forward<A>(a)...); }
};
template <class T> using pod_vector = std::vector<T, default_init_allocator<T>>;

struct CsrData {
  pod_vector<int64_t> row_ptr{0};
  pod_vector<int32_t> field, feat, label;
  pod_vector<float> val;
  size_t n_rows() const { return row_ptr.size() - 1; }
  // rows [r0, r1) as one block
  void slice(size_t r0, size_t r1, CsrBlock &out) const;
  // rows idx[0..n) (any order) as one block -- the offline trainer's shuffled visit
  // (n_threads > 1: the rows are copied by that many OpenMP threads -- a shuffled visit is a
  // random walk over the file image, bound by memory latency, not bandwidth)
  void gather(const int *idx, size_t n, CsrBlock &out, int n_threads = 1) const;
  size_t gather_nnz(const int *idx, size_t n) const;  // entries that block will hold
};

CsrData load_csr(const std::string &path, const std::string &file_type, int n_threads);

// The rows of a byte range of a libffm / libsvm text (whole lines), as parallel arrays: what one
// reader thread produces (load_csr's partitions, csr_stream's chunks).
struct CsrPart {
  std::vector<int32_t> nnz, field, feat, label;  // nnz, label: per row
  std::vector<float> val;
  void clear() { nnz.clear(); field.clear(); feat.clear(); label.clear(); val.clear(); }
  void swap_clear(CsrPart &other) { CsrPart empty; std::swap(other, empty); }  // releases other's memory
};
void parse_csr_range(const char *begin, const char *end, bool has_field, CsrPart &out);

}  // namespace ftrl
