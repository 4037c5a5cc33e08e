// Host unit test of mrs_optic_flow_amd/csrc/geom_core.hpp pieces that have a device-only twin: the fully unrolled
// elimination (solve_linear_unrolled, what the device's RANSAC hypotheses run) must agree with the indexed form
// (solve_linear, what the host form and the oracle-facing tests run) BIT FOR BIT -- same pivots, same operations.
// Built with -ffp-contract=off like the library's geometry file. Prints "geom core ok <cases>" or a diagnostic.
#include <cstdio>
#include <cstring>
#include <cstdint>

#include "../../mrs_optic_flow_amd/csrc/geom_core.hpp"

using namespace mof::geom;

static uint64_t rng_state = 0x1234567ull;
static double rnd() {  // [-1, 1)
  rng_state = splitmix64(rng_state);
  return (double)(int64_t)(rng_state >> 11) / (double)(1ull << 52) - 1.0;
}

template <int N>
static int run_cases(int cases) {
  int solved = 0;
  for (int c = 0; c < cases; ++c) {
    double a[N * (N + 1)], b[N * (N + 1)], x[N], y[N];
    for (int i = 0; i < N * (N + 1); ++i) a[i] = rnd() * (c % 7 == 0 ? 1e6 : 1.0);
    if (c % 5 == 1)  // zero leading entries: pivoting has to move rows
      for (int r = 0; r < N - 1; ++r) a[r * (N + 1) + r] = 0.0;
    if (c % 11 == 2)  // a repeated row: singular
      std::memcpy(a + (N + 1), a, sizeof(double) * (N + 1));
    if (c % 13 == 3)  // exact zeros below the diagonal: the `f != 0` shortcut
      for (int r = 1; r < N; ++r) a[r * (N + 1)] = 0.0;
    std::memcpy(b, a, sizeof(a));
    const bool ok1 = solve_linear<N>(a, x), ok2 = solve_linear_unrolled<N>(b, y);
    if (ok1 != ok2) {
      std::printf("case %d: solvable %d vs %d\n", c, (int)ok1, (int)ok2);
      return -1;
    }
    if (ok1) {
      if (std::memcmp(x, y, sizeof(x)) != 0) {
        std::printf("case %d: solutions differ\n", c);
        return -1;
      }
      ++solved;
    }
  }
  return solved;
}

int main() {
  const int s8 = run_cases<8>(4000), s3 = run_cases<3>(500);
  if (s8 < 0 || s3 < 0) return 1;
  if (s8 < 3000) {
    std::printf("too few solvable systems: %d\n", s8);
    return 1;
  }
  std::printf("geom core ok %d %d\n", s8, s3);
  return 0;
}
