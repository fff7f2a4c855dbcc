// main.cpp -- the trainer CLI (reference src/main.cpp:13-34): parse flags, pick the online or the
// offline task, train.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <string>

#include "cmd_option.h"
#include "trainer.h"

int main(int argc, char *argv[]) {
  config_options opt;
  try {
    opt.parse_option(argc, argv);
  } catch (const std::invalid_argument &e) {
    std::fprintf(stderr, "invalid argument: %s\n%s", e.what(), std::string(cmd_help).c_str());
    return EXIT_FAILURE;
  }
  try {
    // --model_path is parsed but never used by the reference (cmd_option.cpp:67-68); here a
    // trained model is written there: *.zst -> one zstd frame, anything else -> the text format,
    // and the FTRL accumulators next to it (<path>.nz) so training can resume.
    auto save = [&](ftrl::FtrlModel &m) {
      if (opt.model_path.empty()) return;
      const bool zst = opt.model_path.size() > 4 &&
                       opt.model_path.compare(opt.model_path.size() - 4, 4, ".zst") == 0;
      if (zst) m.save_compressed_model(opt.model_path, 3); else m.save_model(opt.model_path);
      m.save_state(opt.model_path + ".nz");
    };
    if (opt.online) {
      ftrl::FtrlOnline task(opt);
      task.train();
      save(*task.model_ptr);
    } else {
      ftrl::FtrlOffline task(opt);
      task.train();
      save(*task.model_ptr);
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return EXIT_FAILURE;
  }
  return 0;
}
