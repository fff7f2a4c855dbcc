// cmd_option.h -- flags of the trainer.  Names and defaults are the reference's
// (src/include/utils/cmd_option.h:29-66, src/utils/cmd_option.cpp:61-114); the last block is new.
#pragma once
#include <cstdint>
#include <string>
#include <string_view>

struct config_options {
  std::string model_path, train_path, eval_path;
  std::string model_type = "FFM";
  std::string file_type;
  float init_mean = 0.0f, init_stddev = 0.02f;
  float w_alpha = 1e-4f, w_beta = 1.0f, w_l1 = 0.1f, w_l2 = 5.0f;
  int thread_num = 1;  // host threads for parsing; the model update itself runs on the GPU
  int epoch = 1;
  int n_fields = 8, n_feats = 10000, n_factors = 16;
  bool cmd = false;
  bool online = true;
  // --- new: the mini-batch scheduler that replaces the per-sample thread pool ---
  int batch_size = 4096;   // --batch_size: rows per block handed to the engine
  int batch_ramp = -1;     // --batch_ramp: block size <= rows_seen / ramp (0 = off; -1 = the engine's default
                           //   for w_alpha, ffm_engine_default_batch_ramp: 32 at the reference's rates); DESIGN.md
  uint64_t seed = 42;      // --seed: weight init and the offline shuffle (the reference is unseeded)
  int device = 0;          // --device: HIP device ordinal
  bool learn = false;      // --learn: FFM_FLAG_LEARN, the opt-in variant in which the factors train
  int n_gpus = 1;          // --n_gpus: field-pair shards, one engine per device, RCCL all-reduce per block
  std::string field_ranges = "none";  // --field_ranges uniform: field f owns ids [f*n_feats/F, (f+1)*n_feats/F)
                                      //   (the layout of python/generate_data.py:272-306): shards then store
                                      //   only their slots; none: every shard keeps whole records

  void parse_option(int argc, char *argv[]);  // throws std::invalid_argument like the reference
};

std::string detect_file_type(const std::string &file_path);  // cmd_option.cpp:35-59
extern const std::string_view cmd_help;
