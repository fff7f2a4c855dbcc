#include "persist.h"

#include <dlfcn.h>

#include <cstdlib>
#include <cstring>
#include <sstream>
#include <stdexcept>

namespace ftrl {
namespace {

// libzstd entry points (the reference links the vendored library, compress.cpp:17-40), bound at run time.
struct ZInBuf { const void *src; size_t size; size_t pos; };
struct ZOutBuf { void *dst; size_t size; size_t pos; };
struct Zstd {
  void *(*createCCtx)() = nullptr;
  size_t (*freeCCtx)(void *) = nullptr;
  size_t (*cctxSetParameter)(void *, int, int) = nullptr;
  size_t (*cctxSetPledgedSrcSize)(void *, unsigned long long) = nullptr;
  size_t (*compressStream2)(void *, ZOutBuf *, ZInBuf *, int) = nullptr;
  void *(*createDCtx)() = nullptr;
  size_t (*freeDCtx)(void *) = nullptr;
  size_t (*decompressStream)(void *, ZOutBuf *, ZInBuf *) = nullptr;
  unsigned long long (*getFrameContentSize)(const void *, size_t) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  bool ok = false;
  Zstd() {
    void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
#define BIND(field, sym) field = reinterpret_cast<decltype(field)>(dlsym(h, sym))
    BIND(createCCtx, "ZSTD_createCCtx");
    BIND(freeCCtx, "ZSTD_freeCCtx");
    BIND(cctxSetParameter, "ZSTD_CCtx_setParameter");
    BIND(cctxSetPledgedSrcSize, "ZSTD_CCtx_setPledgedSrcSize");
    BIND(compressStream2, "ZSTD_compressStream2");
    BIND(createDCtx, "ZSTD_createDCtx");
    BIND(freeDCtx, "ZSTD_freeDCtx");
    BIND(decompressStream, "ZSTD_decompressStream");
    BIND(getFrameContentSize, "ZSTD_getFrameContentSize");
    BIND(isError, "ZSTD_isError");
#undef BIND
    ok = createCCtx && freeCCtx && cctxSetParameter && cctxSetPledgedSrcSize && compressStream2 &&
         createDCtx && freeDCtx && decompressStream && getFrameContentSize && isError;
  }
};
Zstd &zstd() {
  static Zstd z;
  return z;
}
constexpr int kParamCompressionLevel = 100;  // ZSTD_c_compressionLevel
constexpr int kEndContinue = 0, kEndEnd = 2;  // ZSTD_e_continue / ZSTD_e_end
constexpr size_t kIoBuf = 1 << 20;

}  // namespace

bool zstd_available() { return zstd().ok; }

struct FloatFrameWriter::Impl {
  std::string path;      // the file as the caller named it
  std::string tmp_path;  // what is written: renamed over `path` by finish(), removed otherwise
  std::FILE *f = nullptr;
  void *cctx = nullptr;
  std::vector<char> out = std::vector<char>(kIoBuf);
  size_t total_bytes = 0, fed = 0, written = 0;
  bool done = false;
  void pump(const void *src, size_t bytes, int end_op) {
    ZInBuf in{src, bytes, 0};
    for (;;) {
      ZOutBuf ob{out.data(), out.size(), 0};
      const size_t r = zstd().compressStream2(cctx, &ob, &in, end_op);
      if (zstd().isError(r)) throw std::runtime_error("ZSTD_compressStream2 failed: " + path);
      if (ob.pos && std::fwrite(out.data(), 1, ob.pos, f) != ob.pos) throw std::runtime_error("cannot write " + path);
      written += ob.pos;
      if (end_op == kEndEnd ? r == 0 : in.pos == in.size) break;
    }
  }
};

FloatFrameWriter::FloatFrameWriter(const std::string &path, size_t total_floats, int level) : d_(new Impl) {
  if (!zstd().ok) throw std::runtime_error("libzstd.so.1 not available");
  d_->path = path;
  d_->total_bytes = total_floats * sizeof(float);
  // written beside the target and renamed when complete: an exception half way (a device error while
  // the model is pulled, a full disk) leaves the previous checkpoint where it was
  d_->tmp_path = path + ".tmp";
  d_->f = std::fopen(d_->tmp_path.c_str(), "wb");
  if (!d_->f) throw std::runtime_error("cannot write " + d_->tmp_path);
  d_->cctx = zstd().createCCtx();
  if (!d_->cctx) throw std::runtime_error("ZSTD_createCCtx failed");
  if (zstd().isError(zstd().cctxSetParameter(d_->cctx, kParamCompressionLevel, level)))
    throw std::runtime_error("ZSTD_CCtx_setParameter(compressionLevel) failed");
  // the frame header then carries the content size, as a one-shot ZSTD_compress frame does -- this
  // reader and the reference's loader both need it there
  if (zstd().isError(zstd().cctxSetPledgedSrcSize(d_->cctx, d_->total_bytes)))
    throw std::runtime_error("ZSTD_CCtx_setPledgedSrcSize failed");
}
FloatFrameWriter::~FloatFrameWriter() {
  if (d_->cctx) zstd().freeCCtx(d_->cctx);
  if (d_->f) std::fclose(d_->f);
  if (!d_->done && !d_->tmp_path.empty()) std::remove(d_->tmp_path.c_str());  // an unfinished file
}
void FloatFrameWriter::write(const float *p, size_t n) {
  d_->fed += n * sizeof(float);
  if (d_->fed > d_->total_bytes) throw std::logic_error("FloatFrameWriter: more floats than pledged");
  d_->pump(p, n * sizeof(float), kEndContinue);
}
void FloatFrameWriter::finish() {
  if (d_->done) return;
  if (d_->fed != d_->total_bytes) throw std::logic_error("FloatFrameWriter: fewer floats than pledged");
  d_->pump(nullptr, 0, kEndEnd);
  const bool closed = std::fclose(d_->f) == 0;
  d_->f = nullptr;
  if (!closed || std::rename(d_->tmp_path.c_str(), d_->path.c_str()) != 0)
    throw std::runtime_error("cannot write " + d_->path);
  d_->done = true;
  std::printf("saving to %s, before: %zu -> after: %zu\n", d_->path.c_str(), d_->total_bytes, d_->written);
}

struct FloatFrameReader::Impl {
  std::string path;
  std::FILE *f = nullptr;
  void *dctx = nullptr;
  std::vector<char> in = std::vector<char>(kIoBuf);
  ZInBuf ib{nullptr, 0, 0};
  unsigned long long content = 0;
  bool frame_done = false;
  char carry[4];
  size_t n_carry = 0;  // bytes of a float split across two reads
};

FloatFrameReader::FloatFrameReader(const std::string &path) : d_(new Impl) {
  if (!zstd().ok) throw std::runtime_error("libzstd.so.1 not available");
  d_->path = path;
  d_->f = std::fopen(path.c_str(), "rb");
  if (!d_->f) throw std::runtime_error("Failed to open loading file " + path);
  const size_t got = std::fread(d_->in.data(), 1, d_->in.size(), d_->f);
  d_->ib = ZInBuf{d_->in.data(), got, 0};
  d_->content = zstd().getFrameContentSize(d_->in.data(), got);
  if (d_->content == 0ULL - 1 || d_->content == 0ULL - 2)  // ZSTD_CONTENTSIZE_UNKNOWN / _ERROR
    throw std::runtime_error(path + ": not compressed by zstd!");
  d_->dctx = zstd().createDCtx();
}
FloatFrameReader::~FloatFrameReader() {
  if (d_->dctx) zstd().freeDCtx(d_->dctx);
  if (d_->f) std::fclose(d_->f);
}
size_t FloatFrameReader::total_floats() const { return static_cast<size_t>(d_->content / sizeof(float)); }
size_t FloatFrameReader::read(float *p, size_t n) {
  char *dst = reinterpret_cast<char *>(p);
  const size_t want = n * sizeof(float);
  size_t have = 0;
  if (d_->n_carry) { std::memcpy(dst, d_->carry, d_->n_carry); have = d_->n_carry; d_->n_carry = 0; }
  while (have < want && !d_->frame_done) {
    if (d_->ib.pos == d_->ib.size) {
      const size_t got = std::fread(d_->in.data(), 1, d_->in.size(), d_->f);
      if (got == 0) throw std::runtime_error(d_->path + ": zstd frame is truncated");
      d_->ib = ZInBuf{d_->in.data(), got, 0};
    }
    ZOutBuf ob{dst, want, have};
    const size_t r = zstd().decompressStream(d_->dctx, &ob, &d_->ib);
    if (zstd().isError(r)) throw std::runtime_error(d_->path + ": zstd frame is corrupt");
    have = ob.pos;
    if (r == 0) d_->frame_done = true;
  }
  const size_t whole = have / sizeof(float);
  d_->n_carry = have - whole * sizeof(float);
  if (d_->n_carry) std::memcpy(d_->carry, dst + whole * sizeof(float), d_->n_carry);
  return whole;
}

// ffm.cpp:163-180: bias and lin_w through an ostream (default precision, 6 significant digits),
// latent rows as the shortest float representation that round-trips, space separated.
TextModelWriter::TextModelWriter(const std::string &path) : f_(path), path_(path) {
  if (!f_.good()) throw std::runtime_error("cannot write " + path);
}
void TextModelWriter::scalar(float v) { f_ << v << "\n"; }
void TextModelWriter::rows(const float *p, size_t n_rows, size_t row_len) {
  std::string out;
  out.reserve(n_rows * row_len * 12);
  char buf[64];
  for (size_t i = 0; i < n_rows; i++) {
    for (size_t j = 0; j < row_len; j++) {
      const int n = std::snprintf(buf, sizeof buf, "%.9g", static_cast<double>(p[i * row_len + j]));
      if (j) out.push_back(' ');
      out.append(buf, static_cast<size_t>(n));
    }
    out.push_back('\n');
  }
  f_ << out;
}
void TextModelWriter::finish() {
  f_.flush();
  if (!f_.good()) throw std::runtime_error("cannot write " + path_);
  f_.close();
}

// ffm.cpp:182-200
TextModelReader::TextModelReader(const std::string &path) : f_(path), path_(path) {
  if (!f_.good()) {
    std::fprintf(stderr, "Failed to open loading file %s\n", path.c_str());
    throw std::runtime_error("Failed to open loading file " + path);
  }
}
float TextModelReader::scalar() {
  std::getline(f_, line_);
  return std::stof(line_);
}
void TextModelReader::rows(float *p, size_t n_rows, size_t row_len) {
  for (size_t i = 0; i < n_rows; i++) {
    std::getline(f_, line_);
    const char *q = line_.c_str();
    for (size_t j = 0; j < row_len; j++) {
      char *end = nullptr;
      p[i * row_len + j] = std::strtof(q, &end);
      if (end == q) throw std::out_of_range("model file row too short: " + path_);
      q = end;
    }
  }
}

}  // namespace ftrl
