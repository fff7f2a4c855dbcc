#include "persist.h"

#include <dlfcn.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>
#include <stdexcept>

namespace ftrl {
namespace {

// The four libzstd entry points the reference uses (compress.cpp:17-40), bound at run time.
struct Zstd {
  size_t (*compressBound)(size_t) = nullptr;
  size_t (*compress)(void *, size_t, const void *, size_t, int) = nullptr;
  size_t (*decompress)(void *, size_t, const void *, size_t) = nullptr;
  unsigned long long (*getFrameContentSize)(const void *, size_t) = nullptr;
  unsigned (*isError)(size_t) = nullptr;
  bool ok = false;
  Zstd() {
    void *h = dlopen("libzstd.so.1", RTLD_NOW | RTLD_LOCAL);
    if (!h) h = dlopen("libzstd.so", RTLD_NOW | RTLD_LOCAL);
    if (!h) return;
    compressBound = reinterpret_cast<decltype(compressBound)>(dlsym(h, "ZSTD_compressBound"));
    compress = reinterpret_cast<decltype(compress)>(dlsym(h, "ZSTD_compress"));
    decompress = reinterpret_cast<decltype(decompress)>(dlsym(h, "ZSTD_decompress"));
    getFrameContentSize =
        reinterpret_cast<decltype(getFrameContentSize)>(dlsym(h, "ZSTD_getFrameContentSize"));
    isError = reinterpret_cast<decltype(isError)>(dlsym(h, "ZSTD_isError"));
    ok = compressBound && compress && decompress && getFrameContentSize && isError;
  }
};
Zstd &zstd() {
  static Zstd z;
  return z;
}

std::vector<char> read_file(const std::string &path) {
  std::ifstream f(path, std::ios::binary);
  if (!f.good()) throw std::runtime_error("Failed to open loading file " + path);
  f.seekg(0, std::ios::end);
  std::vector<char> buf(static_cast<size_t>(f.tellg()));
  f.seekg(0);
  f.read(buf.data(), static_cast<std::streamsize>(buf.size()));
  return buf;
}

}  // namespace

bool zstd_available() { return zstd().ok; }

void write_compressed_floats(const std::string &path, const std::vector<float> &v, int level) {
  if (!zstd().ok) throw std::runtime_error("libzstd.so.1 not available");
  const size_t bytes = v.size() * sizeof(float);
  std::vector<char> out(zstd().compressBound(bytes));
  const size_t n = zstd().compress(out.data(), out.size(), v.data(), bytes, level);
  if (zstd().isError(n)) throw std::runtime_error("ZSTD_compress failed");
  std::ofstream f(path, std::ios::binary);
  if (!f.good()) throw std::runtime_error("cannot write " + path);
  f.write(out.data(), static_cast<std::streamsize>(n));
  std::printf("saving to %s, before: %zu -> after: %zu\n", path.c_str(), bytes, n);
}

std::vector<float> read_compressed_floats(const std::string &path) {
  if (!zstd().ok) throw std::runtime_error("libzstd.so.1 not available");
  const std::vector<char> in = read_file(path);
  const unsigned long long sz = zstd().getFrameContentSize(in.data(), in.size());
  if (sz == 0ULL - 1 || sz == 0ULL - 2)  // ZSTD_CONTENTSIZE_UNKNOWN / _ERROR
    throw std::runtime_error(path + ": not compressed by zstd!");
  std::vector<float> v(static_cast<size_t>(sz) / sizeof(float));
  const size_t n = zstd().decompress(v.data(), static_cast<size_t>(sz), in.data(), in.size());
  if (zstd().isError(n) || n != sz) throw std::runtime_error(path + ": zstd frame is corrupt");
  return v;
}

// ffm.cpp:163-180: bias and lin_w through an ostream (default precision, 6 significant digits),
// latent rows as the shortest float representation that round-trips, space separated.
void write_text_model(const std::string &path, const ModelWeights &w, size_t n_feats, size_t row_len) {
  std::ostringstream ost;
  ost << w.bias << "\n";
  for (size_t i = 0; i < n_feats; i++) ost << w.lin_w[i] << "\n";
  char buf[64];
  for (size_t i = 0; i < n_feats; i++) {
    for (size_t j = 0; j < row_len; j++) {
      std::snprintf(buf, sizeof buf, "%.9g", static_cast<double>(w.vec_w[i * row_len + j]));
      if (j) ost << ' ';
      ost << buf;
    }
    ost << "\n";
  }
  std::ofstream f(path);
  if (!f.good()) throw std::runtime_error("cannot write " + path);
  f << ost.str();
}

// ffm.cpp:182-200
ModelWeights read_text_model(const std::string &path, size_t n_feats, size_t row_len) {
  std::ifstream ifs(path);
  if (!ifs.good()) {
    std::fprintf(stderr, "Failed to open loading file %s\n", path.c_str());
    throw std::runtime_error("Failed to open loading file " + path);
  }
  ModelWeights w;
  w.lin_w.resize(n_feats);
  w.vec_w.resize(n_feats * row_len);
  std::string line;
  std::getline(ifs, line);
  w.bias = std::stof(line);
  for (size_t i = 0; i < n_feats; i++) {
    std::getline(ifs, line);
    w.lin_w[i] = std::stof(line);
  }
  for (size_t i = 0; i < n_feats && row_len; i++) {
    std::getline(ifs, line);
    const char *p = line.c_str();
    for (size_t j = 0; j < row_len; j++) {
      char *end = nullptr;
      w.vec_w[i * row_len + j] = std::strtof(p, &end);
      if (end == p) throw std::out_of_range("model file row too short: " + path);
      p = end;
    }
  }
  return w;
}

// ffm.cpp:138-146 / lr.cpp:26-31
void write_compressed_model(const std::string &path, const ModelWeights &w, int compress_level) {
  std::vector<float> flat;
  flat.reserve(1 + w.lin_w.size() + w.vec_w.size());
  flat.push_back(w.bias);
  flat.insert(flat.end(), w.lin_w.begin(), w.lin_w.end());
  flat.insert(flat.end(), w.vec_w.begin(), w.vec_w.end());
  write_compressed_floats(path, flat, compress_level);
}

// ffm.cpp:148-161 / lr.cpp:33-39
ModelWeights read_compressed_model(const std::string &path, size_t n_feats, size_t row_len) {
  const std::vector<float> flat = read_compressed_floats(path);
  if (flat.size() != 1 + n_feats + n_feats * row_len)
    throw std::runtime_error(path + ": holds a model of a different shape");
  ModelWeights w;
  w.bias = flat[0];
  w.lin_w.assign(flat.begin() + 1, flat.begin() + 1 + static_cast<long>(n_feats));
  w.vec_w.assign(flat.begin() + 1 + static_cast<long>(n_feats), flat.end());
  std::printf("loading from %s, floats: %zu\n", path.c_str(), flat.size());
  return w;
}

}  // namespace ftrl
