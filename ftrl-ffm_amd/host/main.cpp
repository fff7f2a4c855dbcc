// main.cpp -- the trainer CLI (reference src/main.cpp:13-34): parse flags, pick the online or the
// offline task, train.
#include <cstdio>
#include <cstdlib>
#include <stdexcept>

#include "cmd_option.h"
#include "trainer.h"

int main(int argc, char *argv[]) {
  config_options opt;
  try {
    opt.parse_option(argc, argv);
  } catch (const std::invalid_argument &e) {
    std::fprintf(stderr, "invalid argument: %s\n%s", e.what(), cmd_help);
    return EXIT_FAILURE;
  }
  try {
    if (opt.online) {
      ftrl::FtrlOnline task(opt);
      task.train();
    } else {
      ftrl::FtrlOffline task(opt);
      task.train();
    }
  } catch (const std::exception &e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return EXIT_FAILURE;
  }
  return 0;
}
