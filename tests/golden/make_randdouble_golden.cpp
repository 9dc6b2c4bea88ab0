// Generates golden RandDouble streams with the REAL libstdc++ (std::mt19937 + std::uniform_real_distribution<double>), i.e. what
// theia::RandomNumberGenerator::RandDouble (src/theia/util/random.cc:68-72) executes -- the draws the P4Pfr solver takes for its
// "random rotation" (sfm/pose/four_point_focal_length_radial_distortion.cc:134-138) -- alone and interleaved with RandInt on the
// one shared generator (every RandomNumberGenerator object uses the same thread_local std::mt19937, random.cc:46-66).
// Build + run:  g++ -O2 -o /tmp/mk tests/golden/make_randdouble_golden.cpp && /tmp/mk > tests/golden/mt19937_randdouble.json
#include <cstdio>
#include <random>

static std::mt19937 gen;
static double RandDouble(double lo, double hi) { std::uniform_real_distribution<double> d(lo, hi); return d(gen); }
static int RandInt(int lo, int hi) { std::uniform_int_distribution<int> d(lo, hi); return d(gen); }

int main() {
  std::printf("{\n\"randdouble\": [");
  const unsigned seeds[3] = {42u, 64u, 7u};
  const double ranges[3][2] = {{-0.5, 0.5}, {-1.0, 1.0}, {3.0, 10.0}};
  bool first = true;
  for (unsigned s : seeds) {
    gen.seed(s);
    for (int k = 0; k < 300; ++k) {
      const double* r = ranges[k % 3];
      std::printf("%s[%u,%.17g,%.17g,%.17g]", first ? "" : ",", s, r[0], r[1], RandDouble(r[0], r[1]));
      first = false;
    }
  }
  std::printf("],\n\"interleaved\": {\"seed\": 65, \"n\": 200, \"rounds\": [");
  // one RANSAC round of a 4-point estimator over 200 data: four RandInt(i, n - 1) of the partial shuffle, then three RandDouble
  gen.seed(65u);
  for (int it = 0; it < 40; ++it) {
    int a[4];
    for (int i = 0; i < 4; ++i) a[i] = RandInt(i, 199);
    const double x = RandDouble(-0.5, 0.5), y = RandDouble(-0.5, 0.5), z = RandDouble(-0.5, 0.5);
    std::printf("%s[%d,%d,%d,%d,%.17g,%.17g,%.17g]", it ? "," : "", a[0], a[1], a[2], a[3], x, y, z);
  }
  std::printf("]},\n\"interleaved_first_call\": {\"seed\": 65, \"n\": 200, \"rounds\": [");
  // the same in a fresh process: the solver's function-static RandomNumberGenerator(42) re-seeds the shared generator when the first
  // EstimateModel call reaches it -- after the first sample, before the first three RandDouble
  gen.seed(65u);
  for (int it = 0; it < 40; ++it) {
    int a[4];
    for (int i = 0; i < 4; ++i) a[i] = RandInt(i, 199);
    if (it == 0) gen.seed(42u);
    const double x = RandDouble(-0.5, 0.5), y = RandDouble(-0.5, 0.5), z = RandDouble(-0.5, 0.5);
    std::printf("%s[%d,%d,%d,%d,%.17g,%.17g,%.17g]", it ? "," : "", a[0], a[1], a[2], a[3], x, y, z);
  }
  std::printf("]}\n}\n");
  return 0;
}
