#include "cmd_option.h"

#include <algorithm>
#include <cctype>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <sstream>
#include <stdexcept>
#include <vector>

const std::string_view cmd_help =
    "\nUsage: ./ftrl_ffm_main [<options>]\n\noptions:\n"
    "--model_path <model_path>: set the output model path\n"
    "--train_data <data_path>: set the train data path\n"
    "--eval_data <data_path>: set the eval data path\n"
    "--model_type <model_type>: LR, FM or FFM\n"
    "--init_mean <mean>: mean for parameter initialization\tdefault:0.0\n"
    "--init_stddev <stddev>: stddev for parameter initialization\tdefault:0.02\n"
    "--n_fields <n_fields>: number of fields in FFM\tdefault:8\n"
    "--n_feats <n_feats>: number of total features\tdefault:10000\n"
    "--n_factors <n_factors>: number of embed size in FM and FFM\tdefault:16\n"
    "--w_alpha <w_alpha>: alpha is one of the learning rate parameters\tdefault:1e-4\n"
    "--w_beta <w_beta>: beta is one of the learning rate parameters\tdefault:1.0\n"
    "--w_l1 <w_L1_reg>: L1 regularization parameter of w\tdefault:0.1\n"
    "--w_l2 <w_L2_reg>: L2 regularization parameter of w\tdefault:5.0\n"
    "--n_threads <threads_num>: host threads for parsing\tdefault:1\n"
    "--n_epochs <epochs>: how many epochs to train\tdefault:1\n"
    "--online <online>: whether to online training mode\tdefault:true\n"
    "--batch_size <rows>: rows per block sent to the GPU\tdefault:4096\n"
    "--batch_ramp <r>: block size grows as rows_seen/r (0 disables)\tdefault: by w_alpha (32 up to 1e-3)\n"
    "--seed <seed>: seed of the weight init and the offline shuffle\tdefault:42\n"
    "--device <id>: HIP device ordinal\tdefault:0\n"
    "--n_gpus <n>: shard the field pairs over n devices (one engine each, RCCL all-reduce of the\n"
    "              partial logits per block)\tdefault:1\n"
    "--field_ranges <uniform|none>: uniform = field f owns ids [f*n_feats/n_fields, (f+1)*n_feats/n_fields),\n"
    "              shards then store only their own slots\tdefault:none\n"
    "--learn <bool>: keep initial latent weights until their first gradient and use g2*g2 at\n"
    "                ffm.cpp:118, so FM/FFM factors train (NOT the reference's results)\tdefault:false\n";

static bool assign_bool(std::string arg) {
  std::transform(arg.begin(), arg.end(), arg.begin(), [](unsigned char c) { return std::tolower(c); });
  return arg == "true" || arg == "1";
}

std::string detect_file_type(const std::string &file_path) {
  std::ifstream ifs(file_path);
  if (!ifs.good()) {
    std::fprintf(stderr, "fail to open %s\n", file_path.c_str());
    std::exit(EXIT_FAILURE);
  }
  std::string line;
  std::getline(ifs, line);
  std::istringstream is(line);
  std::string label, first;
  is >> label >> first;
  const auto colons = std::count(first.begin(), first.end(), ':');
  if (colons == 1) return "libsvm";
  if (colons == 2) return "libffm";
  std::fprintf(stderr, "unknown file format...\n");
  std::exit(EXIT_FAILURE);
}

void config_options::parse_option(int argc, char *argv[]) {
  std::vector<std::string> args(argv + 1, argv + argc);
  if (args.size() % 2 != 0) throw std::invalid_argument("every option takes exactly one value");
  for (size_t i = 0; i < args.size(); i += 2) {
    const std::string &k = args[i], &v = args[i + 1];
    if (k == "--model_path") model_path = v;
    else if (k == "--model_type") {
      model_type = v;
      std::transform(model_type.begin(), model_type.end(), model_type.begin(),
                     [](unsigned char c) { return std::toupper(c); });
    }
    else if (k == "--online") online = assign_bool(v);
    else if (k == "--n_fields") n_fields = std::stoi(v);
    else if (k == "--n_feats") n_feats = std::stoi(v);
    else if (k == "--n_factors") n_factors = std::stoi(v);
    else if (k == "--train_data") train_path = v;
    else if (k == "--eval_data") eval_path = v;
    else if (k == "--init_mean") init_mean = std::stof(v);
    else if (k == "--init_stddev") init_stddev = std::stof(v);
    else if (k == "--w_alpha") w_alpha = std::stof(v);
    else if (k == "--w_beta") w_beta = std::stof(v);
    else if (k == "--w_l1") w_l1 = std::stof(v);
    else if (k == "--w_l2") w_l2 = std::stof(v);
    else if (k == "--n_threads") thread_num = std::stoi(v);
    else if (k == "--n_epochs") epoch = std::stoi(v);
    else if (k == "--cmd") cmd = assign_bool(v);
    else if (k == "--batch_size") batch_size = std::stoi(v);
    else if (k == "--batch_ramp") batch_ramp = std::stoi(v);
    else if (k == "--seed") seed = std::stoull(v);
    else if (k == "--device") device = std::stoi(v);
    else if (k == "--learn") learn = assign_bool(v);
    else if (k == "--n_gpus") n_gpus = std::stoi(v);
    else if (k == "--field_ranges") field_ranges = v;
    else throw std::invalid_argument("unknown argument: " + k + "\n");
  }
  file_type = detect_file_type(train_path);
  if (model_type == "FFM" && file_type != "libffm") {
    std::fprintf(stderr, "FFM model requires libffm data format...\n");
    std::exit(EXIT_FAILURE);
  }
}
