// tensor-ops-mnist on the HIP backend: app/MNIST.hs against the C++ host mirror.
//
//   loadData (:160-197)   the four IDX files from --data (no downloader here: there is no network;
//                         `--synthetic NTRAIN,NTEST` builds a stand-in data set of the same format)
//   processDat (:199-221) pixel/255, label -> oneHot 1 0 -- built on the device in one pass
//   learn (:236-398)      genNet (layers `zip` repeat (actMap logistic)) actSoftmax; per epoch:
//                         optional white-noise class (:299-306), uniformShuffle (:308), per "batch":
//                         trainAll = foldl' trainNetwork crossEntropy rate (:390-393, per-sample ONLINE
//                         SGD, here one replayed HIP graph per sample), training/validation error,
//                         confusion matrix (:335-389), optional induced digit (:357-365, :399-411)
//
// Differences from the reference, all outside the arithmetic: the RNG is a host splitmix64 /
// the library's counter-based device generator (mwc-random streams are not reproducible anyway);
// validation runs as batched inference + device argMax instead of one runNetwork per image;
// `--epochs` bounds the reference's endless epoch loop; `--minibatch M` (not in the reference)
// switches the update to the batched gradTOp of M samples; `--f64` selects ElemT = Double.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <sstream>

#include "../tensorops/trainer.hpp"

using namespace tensorops;

// ---- IDX files (big-endian header: 0x00000803 images / 0x00000801 labels) -------------------------
struct Idx {
  std::vector<uint8_t> pixels;  // n * rows * cols
  std::vector<uint8_t> labels;  // n
  int64_t n = 0, rows = 0, cols = 0;
};

static uint32_t be32(const uint8_t* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

static std::vector<uint8_t> slurp(const std::string& path) {
  std::ifstream f(path, std::ios::binary);
  if (!f)
    throw std::runtime_error("'" + path + "' not found; this build has no downloader (no network): place the "
                             "uncompressed MNIST IDX files in the --data directory or use --synthetic");
  return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

static Idx load_pair(const std::string& dir, const char* images, const char* labels) {
  Idx d;
  const std::vector<uint8_t> im = slurp(dir + "/" + images), lb = slurp(dir + "/" + labels);
  if (im.size() < 16 || be32(im.data()) != 0x00000803u)
    throw std::runtime_error(std::string("Could not decode image ") + images + ".");
  if (lb.size() < 8 || be32(lb.data()) != 0x00000801u)
    throw std::runtime_error(std::string("Could not decode labels ") + labels + ".");
  d.n = be32(im.data() + 4);
  d.rows = be32(im.data() + 8);
  d.cols = be32(im.data() + 12);
  if ((int64_t)im.size() != 16 + d.n * d.rows * d.cols)
    throw std::runtime_error(std::string("Could not decode image ") + images + ".");
  if ((int64_t)be32(lb.data() + 4) != d.n || (int64_t)lb.size() != 8 + d.n)
    throw std::runtime_error(std::string("Could not combine ") + images + " and " + labels + ".");
  d.pixels.assign(im.begin() + 16, im.end());
  d.labels.assign(lb.begin() + 8, lb.end());
  return d;
}

static uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}

// a stand-in with MNIST's format: class c = a bright 6x6 blob at a class-dependent position plus
// noise (learnable by the same network)
static Idx synthetic(int64_t n, uint64_t seed) {
  Idx d;
  d.n = n; d.rows = 28; d.cols = 28;
  d.pixels.assign((size_t)n * 784, 0);
  d.labels.resize((size_t)n);
  uint64_t s = seed;
  for (int64_t i = 0; i < n; ++i) {
    const int c = (int)(splitmix64(s) % 10);
    d.labels[(size_t)i] = (uint8_t)c;
    const int r0 = 2 + (c / 5) * 12 + (int)(splitmix64(s) % 3), c0 = 1 + (c % 5) * 5 + (int)(splitmix64(s) % 2);
    for (int p = 0; p < 784; ++p) d.pixels[(size_t)i * 784 + p] = (uint8_t)(splitmix64(s) % 40);
    for (int r = r0; r < r0 + 6 && r < 28; ++r)
      for (int q = c0; q < c0 + 6 && q < 28; ++q) d.pixels[(size_t)i * 784 + r * 28 + q] = (uint8_t)(200 + splitmix64(s) % 56);
  }
  return d;
}

// ---- device-side data set --------------------------------------------------------------------------
struct DataSet {
  T x;                          // [n; 784], pixel / 255   (processDat, :207)
  std::vector<int64_t> labels;  // Finite o
  int64_t n() const { return (int64_t)labels.size(); }
};

static int g_dtype = TO_F32;

static T upload_rows(const std::vector<double>& v, int64_t n, int64_t w) {
  Dims d{w};
  to_tensor out = nullptr;
  if (g_dtype == TO_F64) {
    check(to_from_host(TO_F64, 1, d.data(), n, v.data(), &out));
  } else {
    std::vector<float> f(v.begin(), v.end());
    check(to_from_host(TO_F32, 1, d.data(), n, f.data(), &out));
  }
  return T(out);
}

static DataSet process(const Idx& raw, int64_t n_in) {
  if (raw.rows * raw.cols != n_in)
    throw std::runtime_error("Bad input vector (Expected " + std::to_string(n_in) + ", got " +
                             std::to_string(raw.rows * raw.cols) + ")");
  std::vector<double> v((size_t)raw.n * n_in);
  for (size_t i = 0; i < v.size(); ++i) v[i] = raw.pixels[i] / 255.0;
  DataSet d;
  d.x = upload_rows(v, raw.n, n_in);
  d.labels.assign(raw.labels.begin(), raw.labels.end());
  return d;
}

// tr <> fmap (, (noiseClass, noiseFin)) extr  with extr = scaleT alpha (genRand (uniformDistr 0 1))  (:299-306)
static DataSet with_noise(const DataSet& d, int64_t noise_class, uint64_t& seed) {
  const int64_t extra = d.n() / 10;
  if (extra == 0) return d;
  const Dims dims = d.x.dims();
  T v = HipT::genRand(dims, 0, 0.0, 1.0, splitmix64(seed), extra);
  T alpha = HipT::genRand({}, 0, 0.0, 1.0, splitmix64(seed), extra);
  T noise = HipT::gmul(0, 0, 1, alpha, v);  // per-sample scaleT alpha v
  to_tensor all = nullptr;
  check(to_alloc(g_dtype, (int)dims.size(), dims.data(), d.n() + extra, &all));
  T out(all);
  to_tensor a = nullptr, b = nullptr;
  check(to_batch_slice(all, 0, d.n(), &a));
  T va(a);
  check(to_copy_into(a, d.x.h()));
  check(to_batch_slice(all, d.n(), extra, &b));
  T vb(b);
  check(to_copy_into(b, noise.h()));
  DataSet r{out, d.labels};
  r.labels.insert(r.labels.end(), (size_t)extra, noise_class);
  return r;
}

static T gather(const T& x, const std::vector<int64_t>& idx) {
  to_tensor out = nullptr;
  check(to_batch_gather(x.h(), (int64_t)idx.size(), idx.data(), &out));
  return T(out);
}
static T rows(const T& x, int64_t start, int64_t count) {
  to_tensor out = nullptr;
  check(to_batch_slice(x.h(), start, count, &out));
  return T(out);
}

// argMax (runNetwork n x) for every row: ONE batched forward pass + the device argMax
static std::vector<int64_t> classify(const Network& net, const T& x) { return HipT::argMax(runNetwork(net, x)); }

static std::vector<double> to_host(const T& t) {
  int64_t n = std::max<int64_t>(t.batch(), 1);
  for (int64_t d : t.dims()) n *= d;
  std::vector<double> out((size_t)n);
  if (g_dtype == TO_F64) {
    check(to_download(t.h(), out.data(), n * 8));
  } else {
    std::vector<float> f((size_t)n);
    check(to_download(t.h(), f.data(), n * 4));
    out.assign(f.begin(), f.end());
  }
  return out;
}

static std::string render_out(const std::vector<double>& px) {  // renderOut (:421-447)
  std::string s;
  for (int r = 0; r < 28; ++r) {
    for (int c = 0; c < 28; ++c) {
      const double v = px[(size_t)r * 28 + c];
      const char ch = v <= 0.2 ? ' ' : v <= 0.4 ? '.' : v <= 0.8 ? '-' : v <= 1.9 ? '=' : '#';
      s += ch;
      s += ch;
    }
    if (r != 27) s += '\n';
  }
  return s;
}

int main(int argc, char** argv) {
  double rate = 0.02;                  // :89-93
  std::vector<int64_t> layers{300, 100};  // :94-98
  int64_t batch = 1000;                // :99-103
  std::string data_dir = "data/mnist";  // :104-108
  bool noconfusion = false, white = false, check_only = false, f64 = false;
  int induce = -1, epochs = 1, induce_iters = 5000;
  int64_t minibatch = 0, syn_train = 0, syn_test = 0, max_batches = 0;
  uint64_t seed = 0x7e500001ull;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); }
      return argv[++i];
    };
    auto list = [](const char* v) {
      std::vector<int64_t> out;
      std::string s(v), tok;
      for (char& ch : s) if (ch == '[' || ch == ']') ch = ' ';
      std::stringstream ss(s);
      while (std::getline(ss, tok, ',')) {
        const long long x = std::atoll(tok.c_str());
        if (!tok.empty() && tok.find_first_not_of(' ') != std::string::npos) out.push_back(x);
      }
      return out;
    };
    if (a == "--rate" || a == "-r") rate = std::atof(next());
    else if (a == "--layers" || a == "-l") layers = list(next());
    else if (a == "--batch" || a == "-b") batch = std::atoll(next());
    else if (a == "--data" || a == "-d") data_dir = next();
    else if (a == "--noconfusion" || a == "-c") noconfusion = true;
    else if (a == "--white" || a == "-w") white = true;
    else if (a == "--induce" || a == "-i") {
      induce = std::atoi(next());
      if (induce < 0 || induce > 9) { std::fprintf(stderr, "Number %d out of range (9)\n", induce); return 2; }
    }
    else if (a == "--epochs") epochs = std::atoi(next());
    else if (a == "--max-batches") max_batches = std::atoll(next());
    else if (a == "--induce-iters") induce_iters = std::atoi(next());
    else if (a == "--minibatch") minibatch = std::atoll(next());
    else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
    else if (a == "--f64") f64 = true;
    else if (a == "--check-data") check_only = true;
    else if (a == "--synthetic") {
      std::vector<int64_t> v = list(next());
      if (v.size() != 2 || v[0] < 1 || v[1] < 1) { std::fprintf(stderr, "--synthetic NTRAIN,NTEST\n"); return 2; }
      syn_train = v[0]; syn_test = v[1];
    } else {
      std::fprintf(stderr,
                   "tensor-ops-mnist - train neural nets on MNIST data set (HIP backend)\n"
                   "usage: %s [-r STEP] [-l LIST] [-b AMOUNT] [-d PATH] [-c] [-w] [-i DIGIT]\n"
                   "          [--epochs N] [--max-batches N] [--minibatch M] [--f64] [--seed N]\n"
                   "          [--synthetic NTRAIN,NTEST] [--check-data] [--induce-iters N]\n", argv[0]);
      return 2;
    }
  }
  if (batch < 1) { std::fprintf(stderr, "--batch must be positive\n"); return 2; }
  try {
    Idx raw_tr, raw_te;
    if (syn_train > 0) {
      raw_tr = synthetic(syn_train, seed + 11);
      raw_te = synthetic(syn_test, seed + 12);
      std::printf("Synthetic data (%lld training, %lld validation samples).\n", (long long)syn_train, (long long)syn_test);
    } else {
      std::printf("Loading data from %s\n", data_dir.c_str());
      raw_tr = load_pair(data_dir, "train-images-idx3-ubyte", "train-labels-idx1-ubyte");  // mnistFiles (:73-76)
      raw_te = load_pair(data_dir, "t10k-images-idx3-ubyte", "t10k-labels-idx1-ubyte");
    }
    std::printf("Loaded data.\n");
    if (check_only) {  // host-only: what was parsed
      for (const Idx* d : {&raw_tr, &raw_te}) {
        uint64_t sum = 0;
        for (uint8_t p : d->pixels) sum += p;
        int64_t hist[10] = {0};
        for (uint8_t l : d->labels) if (l < 10) ++hist[l];
        std::printf("n=%lld rows=%lld cols=%lld pixel_sum=%llu labels=", (long long)d->n, (long long)d->rows,
                    (long long)d->cols, (unsigned long long)sum);
        for (int c = 0; c < 10; ++c) std::printf("%s%lld", c ? "," : "", (long long)hist[c]);
        std::printf("\n");
      }
      return 0;
    }
    for (uint8_t l : raw_tr.labels)
      if (l > 9) throw std::runtime_error("Label out of range (Got " + std::to_string((int)l) + ", expected [0,10) )");

    check(to_init(0));
    if (f64) {
      g_dtype = TO_F64;
      check(to_set_default_dtype(TO_F64));
    }
    const int64_t n_in = 784, n_out = white ? 11 : 10;  // NOut w (:223-232)
    DataSet tr = process(raw_tr, n_in), vd = process(raw_te, n_in);
    check(to_sync());
    std::printf("Data processed.\n");

    // net0 <- genNet (layers `zip` repeat (actMap logistic)) actSoftmax g   (:262-263)
    std::vector<int64_t> sizes{n_in};
    sizes.insert(sizes.end(), layers.begin(), layers.end());
    sizes.push_back(n_out);
    std::vector<std::pair<T, T>> w;
    for (size_t l = 0; l + 1 < sizes.size(); ++l) {
      Network ff = ffLayerRand(sizes[l], sizes[l + 1], seed + 1000 + 2 * l);
      w.emplace_back(ff.params[0], ff.params[1]);
    }
    Network net = genNet(w, act_of(ACT_MAP_LOGISTIC), act_of(ACT_SOFTMAX));

    std::printf("rate: %f | batch: %lld | layers: [", rate, (long long)batch);
    for (size_t i = 0; i < layers.size(); ++i) std::printf("%s%lld", i ? "," : "", (long long)layers[i]);
    std::printf("]\n");
    if (white) std::printf("white noise class enabled\n");
    if (induce >= 0) std::printf("inducing: %d\n", induce);
    if (minibatch > 0) std::printf("update: batched gradTOp over %lld samples\n", (long long)minibatch);

    const int flags = TRAINER_MEMO | TRAINER_FUSED;
    uint64_t rs = seed + 77;
    int64_t batches_done = 0;
    for (int e = 1; e <= epochs; ++e) {
      std::printf("[Epoch %d]\n", e);
      DataSet tr2 = white ? with_noise(tr, 10, rs) : tr;
      // queue <- uniformShuffle tr' g   (:308)
      std::vector<int64_t> perm((size_t)tr2.n());
      for (size_t i = 0; i < perm.size(); ++i) perm[i] = (int64_t)i;
      for (size_t i = perm.size(); i > 1; --i) std::swap(perm[i - 1], perm[(size_t)(splitmix64(rs) % i)]);
      T qx = gather(tr2.x, perm);
      std::vector<int64_t> ql(perm.size());
      for (size_t i = 0; i < perm.size(); ++i) ql[i] = tr2.labels[(size_t)perm[i]];
      T qy = HipT::oneHot(n_out, 1.0, 0.0, ql, true);  // TT.oneHot 1 0   (:216)
      std::printf("Training on %lld samples in batches of %lld ...\n", (long long)tr2.n(), (long long)batch);

      int64_t b = 1;
      for (int64_t start = 0; start < tr2.n(); start += batch, ++b) {
        const int64_t cnt = std::min(batch, tr2.n() - start);
        std::printf("Batch %lld ...\n", (long long)b);
        T bx = rows(qx, start, cnt), by = rows(qy, start, cnt);
        const auto t0 = std::chrono::steady_clock::now();
        if (minibatch <= 0) {
          net = trainAll(net, LOSS_CROSS_ENTROPY, rate, bx, by, cnt, nullptr, flags);  // trainAll (:390-393)
        } else {
          for (int64_t s = 0; s < cnt; s += minibatch) {
            const int64_t m = std::min(minibatch, cnt - s);
            auto t = Trainer::create(net, LOSS_CROSS_ENTROPY, rate, rows(bx, s, m), rows(by, s, m), flags);
            t->grad();
            t->apply();
            Network nn{t->net.op, {}};
            for (const T& p : t->net.params) nn.params.push_back(HipT::scaleT(1.0, p));  // own copies
            net = nn;
          }
        }
        check(to_sync());  // the `evaluate . force` of `time` (:413-420)
        const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
        std::printf("Trained on %lld samples in %.6fs\n", (long long)cnt, secs);

        DataSet vd2 = white ? with_noise(vd, 10, rs) : vd;
        // tscore = F.fold (validate nt') xs   (:336, :366-375)
        const auto v0 = std::chrono::steady_clock::now();
        std::vector<int64_t> pt = classify(net, bx);
        int64_t ok = 0;
        for (int64_t i = 0; i < cnt; ++i) ok += pt[(size_t)i] == ql[(size_t)(start + i)];
        const double tscore = ok / (double)cnt;
        // vconf = F.fold (confusion nt') vd'   (:337, :376-388): predicted y -> actual r -> count
        std::vector<int64_t> pv = classify(net, vd2.x);
        std::vector<std::vector<int64_t>> conf((size_t)n_out, std::vector<int64_t>((size_t)n_out, 0));
        int64_t diag = 0;
        for (int64_t i = 0; i < vd2.n(); ++i) {
          ++conf[(size_t)pv[(size_t)i]][(size_t)vd2.labels[(size_t)i]];
          diag += pv[(size_t)i] == vd2.labels[(size_t)i];
        }
        const double vscore = diag / (double)vd2.n();
        // (not in the reference, which times `trainAll` only: the two validation folds, for bench.py's app-level leg)
        std::printf("Validated on %lld + %lld samples in %.6fs\n", (long long)cnt, (long long)vd2.n(),
                    std::chrono::duration<double>(std::chrono::steady_clock::now() - v0).count());
        std::printf("Training:   %.2f%% error\n", (1 - tscore) * 100);
        std::printf("Validation: %.2f%% error\n", (1 - vscore) * 100);
        if (!noconfusion) {  // rows = actual class, columns = predicted class (the Box layout of :338-349)
          std::vector<size_t> wcol((size_t)n_out, 1);
          for (int64_t y = 0; y < n_out; ++y)
            for (int64_t r = 0; r < n_out; ++r)
              wcol[(size_t)y] = std::max(wcol[(size_t)y], std::to_string(conf[(size_t)y][(size_t)r]).size());
          const size_t wl = std::to_string(n_out - 1).size() + 2;
          for (int64_t r = 0; r < n_out; ++r) {
            std::string lab = "[" + std::to_string(r) + "]";
            std::printf("%-*s", (int)wl, lab.c_str());
            for (int64_t y = 0; y < n_out; ++y) std::printf(" %*lld", (int)wcol[(size_t)y], (long long)conf[(size_t)y][(size_t)r]);
            std::printf("\n");
          }
        }
        if (induce >= 0) {  // :357-365
          T x0 = HipT::genRand({n_in}, 0, 0.0, 0.05, splitmix64(rs));
          T target = HipT::oneHot(n_out, 1.0, 0.0, {(int64_t)induce}, false);
          T x1 = induceNum(net, crossEntropy(), target, 1.0, induce_iters, x0);
          std::printf("%s\n", render_out(to_host(x1)).c_str());
          std::vector<double> y1 = to_host(runNetwork(net, x1));
          for (size_t k = 0; k < y1.size(); ++k) std::printf("%s%.2f", k ? "/" : "", y1[k]);
          std::printf("\n");
        }
        if (max_batches > 0 && ++batches_done >= max_batches) return 0;
      }
    }
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
