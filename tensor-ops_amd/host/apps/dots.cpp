// tensor-ops-dots on the HIP backend: the body of `netTest` (app/Dots.hs:60-92) against the
// C++ host mirror -- generate 2-D points, label them with the two-circle rule, build
// `genNet (hs `zip` repeat actLogistic) actLogistic`, train with per-sample online SGD
// (`foldl' trainEach`, :74-80: trainNetwork squaredError rate), render the 51x21 ASCII map of
// `join TT.dot . runNetwork` (:83-92).
//
// `ElemT t ~ Double` like the reference (:49) unless `--f32`.  Differences, all outside the
// arithmetic: the point generator is a host splitmix64 (mwc-random streams are not reproducible
// anyway); the samples are uploaded once and `foldl' trainEach` runs as `trainAll` (one replayed
// step per sample, tensorops/trainer.hpp); the 1071 map points are evaluated as ONE batched
// runNetwork instead of 1071 calls.
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <sstream>

#include "../tensorops/trainer.hpp"

using namespace tensorops;

static uint64_t splitmix64(uint64_t& s) {
  uint64_t z = (s += 0x9e3779b97f4a7c15ull);
  z = (z ^ (z >> 30)) * 0xbf58476d1ce4e5b9ull;
  z = (z ^ (z >> 27)) * 0x94d049bb133111ebull;
  return z ^ (z >> 31);
}
static double uniform(uint64_t& s, double lo, double hi) {
  return lo + (hi - lo) * ((splitmix64(s) >> 11) * (1.0 / 9007199254740992.0));
}
static bool in_circle(double x, double y, double cx, double cy, double r) {  // Dots.hs:93-100
  const double dx = x - cx, dy = y - cy;
  return dx * dx + dy * dy <= r * r;
}
static double label(double x, double y) {  // Dots.hs:65-69
  return (in_circle(x, y, 0.33, 0.33, 0.33) || in_circle(x, y, -0.33, -0.33, 0.33)) ? 1.0 : 0.0;
}

static int g_dtype = TO_F64;
static T vec_from(const std::vector<double>& v, int64_t batch = 0) {
  Dims d{(int64_t)(batch > 0 ? v.size() / batch : v.size())};
  to_tensor out = nullptr;
  if (g_dtype == TO_F64) {
    check(to_from_host(TO_F64, 1, d.data(), batch, v.data(), &out));
  } else {
    std::vector<float> f(v.begin(), v.end());
    check(to_from_host(TO_F32, 1, d.data(), batch, f.data(), &out));
  }
  return T(out);
}
static std::vector<double> to_host(const T& t, size_t n) {
  std::vector<double> out(n);
  if (g_dtype == TO_F64) {
    check(to_download(t.h(), out.data(), (int64_t)(n * 8)));
  } else {
    std::vector<float> f(n);
    check(to_download(t.h(), f.data(), (int64_t)(n * 4)));
    out.assign(f.begin(), f.end());
  }
  return out;
}

int main(int argc, char** argv) {
  double rate = 1.0;             // Dots.hs:113
  int samps = 50000;             // :118
  std::vector<int64_t> hs{12, 8};  // :123
  uint64_t seed = 0x7e500001ull;
  for (int i = 1; i < argc; ++i) {
    const std::string a = argv[i];
    auto next = [&]() -> const char* {
      if (i + 1 >= argc) { std::fprintf(stderr, "missing value for %s\n", a.c_str()); std::exit(2); }
      return argv[++i];
    };
    if (a == "--rate" || a == "-r") rate = std::atof(next());
    else if (a == "--samps" || a == "-s") samps = std::atoi(next());
    else if (a == "--seed") seed = std::strtoull(next(), nullptr, 0);
    else if (a == "--f32") g_dtype = TO_F32;
    else if (a == "--f64") g_dtype = TO_F64;
    else if (a == "--layers" || a == "-l") {
      hs.clear();
      std::stringstream ss(next());
      std::string tok;
      while (std::getline(ss, tok, ',')) if (!tok.empty()) hs.push_back(std::atoll(tok.c_str()));
    } else {
      std::fprintf(stderr, "usage: %s [--rate STEP] [--samps COUNT] [--layers 12,8] [--seed N] [--f32|--f64]\n", argv[0]);
      return 2;
    }
  }
  try {
    check(to_init(0));
    check(to_set_default_dtype(g_dtype));
    std::printf("rate: %f | samps: %d | layers: [", rate, samps);
    for (size_t i = 0; i < hs.size(); ++i) std::printf("%s%lld", i ? "," : "", (long long)hs[i]);
    std::printf("]\nTraining BLAS (HIP, MI355X) network ...\n");

    uint64_t rs = seed;
    std::vector<double> xs, ys;
    xs.reserve(2 * (size_t)samps);
    ys.reserve((size_t)samps);
    for (int s = 0; s < samps; ++s) {
      const double x = uniform(rs, -1, 1), y = uniform(rs, -1, 1);
      xs.push_back(x);
      xs.push_back(y);
      ys.push_back(label(x, y));
    }
    T X = vec_from(xs, samps), Y = vec_from(ys, samps);
    std::printf("Generated test points\n");

    // genNet (hs `zip` repeat actLogistic) actLogistic   (Dots.hs:72-73)
    std::vector<int64_t> sizes{2};
    sizes.insert(sizes.end(), hs.begin(), hs.end());
    sizes.push_back(1);
    std::vector<std::pair<T, T>> w;
    for (size_t l = 0; l + 1 < sizes.size(); ++l) {
      Network ff = ffLayerRand(sizes[l], sizes[l + 1], seed + 1000 + 2 * l);
      w.emplace_back(ff.params[0], ff.params[1]);
    }
    Network net = genNet(w, actLogistic(), actLogistic());

    const auto t0 = std::chrono::steady_clock::now();
    // foldl' trainEach (Dots.hs:74-80): trainNetwork squaredError rate, sample after sample
    net = trainAll(net, LOSS_SQUARED_ERROR, rate, X, Y, samps, nullptr, TRAINER_MEMO | TRAINER_FUSED);
    check(to_sync());  // the `deepseq` of the reference
    const double secs = std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
    std::printf("Network trained (%.3fs, %.1f samples/s)\n", secs, samps / secs);

    // the map: one batched runNetwork over the 51 x 21 grid   (Dots.hs:83-86)
    std::vector<double> grid;
    for (int y = 0; y <= 20; ++y)
      for (int x = 0; x <= 50; ++x) {
        grid.push_back(x / 25.0 - 1.0);
        grid.push_back(y / 10.0 - 1.0);
      }
    const int64_t npts = 51 * 21;
    T out = runNetwork(net, vec_from(grid, npts));
    T r2 = HipT::gmul(0, 1, 0, out, out);  // join TT.dot
    std::vector<double> r = to_host(r2, (size_t)npts);
    int correct = 0;
    for (int y = 0; y <= 20; ++y) {
      for (int x = 0; x <= 50; ++x) {
        const double v = r[(size_t)(y * 51 + x)];
        std::putchar(v <= 0.2 ? ' ' : v <= 0.4 ? '.' : v <= 0.6 ? '-' : v <= 0.8 ? '=' : '#');
        correct += ((v > 0.5) == (label(x / 25.0 - 1.0, y / 10.0 - 1.0) > 0.5)) ? 1 : 0;
      }
      std::putchar('\n');
    }
    std::printf("grid accuracy: %.4f\n", correct / (double)npts);
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 1;
  }
  return 0;
}
