// mfma_frag.hpp -- host-side helpers for the f16 operand tables of the matrix-core kernels (pc_passes3.hpp: fwd3_rows_mfma;
// sr_fused_kernel.hip). A DFT matrix entry w (fp32) is split w = hi + lo into two halves: u8 pixels are exact in f16, both
// products are exact in the f32 accumulator, so the matrix pass is as accurate as an fp32 one.
#pragma once

#include <cmath>
#include <cstdint>
#include <cstring>

namespace mof {

// float -> f16 bits, round to nearest even (normal and subnormal halves; |x| < 65504)
inline uint16_t f16_bits(float x) {
  uint32_t u;
  std::memcpy(&u, &x, 4);
  const uint16_t sign = (uint16_t)((u >> 16) & 0x8000u);
  const float ax = std::fabs(x);
  if (ax == 0.f) return sign;
  int ex;
  (void)std::frexp(ax, &ex);                                        // ax = f * 2^ex, f in [0.5, 1)
  const int e = ex - 1 < -14 ? -14 : ex - 1;                        // exponent of the half's leading (or subnormal) bit
  const double q = std::nearbyint(std::ldexp((double)ax, 10 - e));  // significand in units of 2^(e-10); ties to even
  uint32_t sig = (uint32_t)q;
  int be = e + 15;
  if (sig >= 2048u) sig >>= 1, ++be;               // rounded up into the next binade
  if (sig < 1024u) return (uint16_t)(sign | sig);  // subnormal (be == 1 here)
  return (uint16_t)(sign | (be << 10) | (sig - 1024u));
}
inline float f16_value(uint16_t hbits) {
  const int be = (hbits >> 10) & 31, sig = hbits & 1023;
  const float v = be ? std::ldexp((float)(1024 + sig), be - 25) : std::ldexp((float)sig, -24);
  return (hbits & 0x8000u) ? -v : v;
}
// w -> (hi, lo) halves
inline void f16_split(float w, uint16_t* hi, uint16_t* lo) {
  *hi = f16_bits(w);
  *lo = f16_bits(w - f16_value(*hi));
}
// cos and sin of 2 pi idx / n in double, exact on the axes
inline void unit_root(int idx, int n, double* c, double* s) {
  idx %= n;
  if ((4 * idx) % n == 0) {
    const int q = (4 * idx) / n;
    *c = q == 0 ? 1.0 : q == 2 ? -1.0 : 0.0;
    *s = q == 1 ? 1.0 : q == 3 ? -1.0 : 0.0;
    return;
  }
  const double a = 2.0 * 3.14159265358979323846 * (double)idx / (double)n;
  *c = std::cos(a);
  *s = std::sin(a);
}

}  // namespace mof
