// Generates the golden RandInt / RandomSampler streams with the REAL libstdc++
// (std::mt19937 + std::uniform_int_distribution<int>), i.e. what
// theia::RandomNumberGenerator::RandInt (src/theia/util/random.cc:81-84) and
// RandomSampler::Sample (src/theia/solvers/random_sampler.cc:62-72) execute.
// Build + run: see make_golden.py.  Output: JSON on stdout.
#include <cstdio>
#include <numeric>
#include <random>
#include <utility>
#include <vector>

static std::mt19937 gen;
static int RandInt(int lo, int hi) { std::uniform_int_distribution<int> d(lo, hi); return d(gen); }

int main() {
  std::printf("{\n");
  // 1. raw RandInt streams with assorted ranges
  const unsigned seeds[3] = {52u, 65u, 66u};
  const int ranges[6][2] = {{0, 9}, {0, 1999}, {3, 99}, {-5, 5}, {0, 2147483646}, {7, 7}};
  std::printf("\"randint\": [");
  bool first = true;
  for (unsigned s : seeds) {
    gen.seed(s);
    for (int k = 0; k < 600; ++k) {
      const int* r = ranges[k % 6];
      std::printf("%s[%u,%d,%d,%d]", first ? "" : ",", s, r[0], r[1], RandInt(r[0], r[1]));
      first = false;
    }
  }
  std::printf("],\n");
  // 2. RandomSampler streams: (seed, N, m) -> 64 consecutive samples
  const int cfg[4][3] = {{65, 100, 5}, {66, 100, 3}, {1234, 2000, 5}, {7, 6, 5}};
  std::printf("\"sampler\": [");
  for (int c = 0; c < 4; ++c) {
    gen.seed((unsigned)cfg[c][0]);
    std::vector<int> idx(cfg[c][1]);
    std::iota(idx.begin(), idx.end(), 0);
    std::printf("%s{\"seed\":%d,\"N\":%d,\"m\":%d,\"samples\":[", c ? "," : "", cfg[c][0], cfg[c][1], cfg[c][2]);
    for (int it = 0; it < 64; ++it)
      for (int i = 0; i < cfg[c][2]; ++i) {
        std::swap(idx[i], idx[RandInt(i, (int)idx.size() - 1)]);
        std::printf("%s%d", (it || i) ? "," : "", idx[i]);
      }
    std::printf("]}");
  }
  std::printf("]\n}\n");
  return 0;
}
