// pc_plan_build.hpp -- the plan of the size-generic K1 as constexpr functions: the host builds it per engine (pc_build_plan),
// and the kernel's compile-time instantiations for the 5-smooth transform sizes (pc_kernel_generic.hip, MS > 0) rebuild the very
// same plan as a constant, so that every run-time quantity of the planned passes (radices, strides, divisions, loop counts)
// folds away. Reference citations as in pc_kernel_generic.hip.
#pragma once

#include <stddef.h>
#include <stdint.h>

#include "mof_kernels.h"

namespace mof {

// cv::getOptimalDFTSize: the smallest 2^a 3^b 5^c >= n (published OpenCV algorithm)
constexpr int pc_optimal_dft_size_c(int n) {
  if (n < 1) return -1;
  for (int m = n; m < (1 << 30); ++m) {
    int r = m;
    while (r % 2 == 0) r /= 2;
    while (r % 3 == 0) r /= 3;
    while (r % 5 == 0) r /= 5;
    if (r == 1) return m;
  }
  return -1;
}

// radix chain of a 5-smooth m: the 5s and the 3s first, then 8s and one 4 or 2 for what is left of the power of two (the
// radix set of ocl_getRadixes, FftMethod.cpp:494-520). The LAST radix is even whenever m is: bin m/2 of a Stockham chain is
// then output R/2 of a twiddle-free last butterfly whose inputs are the p = 0 outputs of twiddle-free butterflies all the way
// down -- sums and differences only, like bin 0. On u8 pixels those are exact in f32 (|sum| < 2^24), so the four real-only CCS
// slots (0 | m/2, 0 | m/2) come out of the PACKED transform exactly as the reference's separate transforms produce them. It
// matters: there C = P / (P^2 + eps), which is 0 for P = 0 but up to 1 / (2 sqrt(eps)) = 1448 for a P of rounding
// noise -- a checkerboard higher than the peak. (Found on 30 x 30 patches: alternating pixel sums cancel exactly in about one
// patch per 480 x 480 frame; with the odd radix last the result was off by 0.1 .. 0.9 px there.)
constexpr int pc_radix_chain_c(int m, int* radix, int max_stages) {
  int ns = 0, r = m, p2 = 0, p3 = 0, p5 = 0;
  while (r % 2 == 0) { r /= 2; ++p2; }
  while (r % 3 == 0) { r /= 3; ++p3; }
  while (r % 5 == 0) { r /= 5; ++p5; }
  if (r != 1) return -1;
  for (; p5 > 0; --p5) { if (ns < max_stages) radix[ns] = 5; ++ns; }
  for (; p3 > 0; --p3) { if (ns < max_stages) radix[ns] = 3; ++ns; }
  const int rest = p2 % 3;  // 8s, then the 4 or 2 (an even radix last either way)
  for (int i = 0; i < p2 / 3; ++i) { if (ns < max_stages) radix[ns] = 8; ++ns; }
  if (rest == 2) { if (ns < max_stages) radix[ns] = 4; ++ns; }
  if (rest == 1) { if (ns < max_stages) radix[ns] = 2; ++ns; }
  return ns <= max_stages ? ns : -1;
}

constexpr int pc_pitch_for(int row) {  // = 8 (mod 16): column walks spread over the banks
  int p = row;
  while (p % 16 != 8) ++p;
  return p;
}

// the transform part of a plan (n, m, radix chain); ok = false for n < 2 or lines beyond the stage routine (m > 960)
constexpr bool pc_line_plan_c(int n, PcPlan& pl) {
  if (n < 2) return false;
  pl = PcPlan{};
  pl.n = n;
  pl.m = pc_optimal_dft_size_c(n);
  if (pl.m < 2) return false;
  pl.n_stages = pc_radix_chain_c(pl.m, pl.radix, 8);
  if (pl.n_stages < 1) return false;
  for (int s = 0; s < pl.n_stages; ++s) {
    if (pl.m / pl.radix[s] > 64 * (16 / pl.radix[s])) return false;  // a line's butterflies must fit one group of the stage routine (m <= 960)
    pl.radix_packed |= (uint32_t)pl.radix[s] << (5 * s);
  }
  pl.hermitian = pl.m % 2 == 0 ? 1 : 0;
  return true;
}

// LDS behind the tile: twiddles (m complex), 16 (value, index) slots, 16 flag words, and the box table (m complex: the 1-D spectrum
// of the zero-padded constant line, pc_kernel_generic.hip)
constexpr size_t pc_tile_extra(int m) { return sizeof(float) * 2 * (size_t)m + 16 * 8 + 64 + sizeof(float) * 2 * (size_t)m; }

// the whole plan of the in-LDS kernel; false when the padded tile does not fit one CU's LDS (m > 135)
constexpr bool pc_tile_plan_c(int n, PcPlan& pl) {
  if (!pc_line_plan_c(n, pl)) return false;
  const size_t extra = pc_tile_extra(pl.m), cap = 160u * 1024u;
  const int skew_row = pl.m + ((pl.m - 1) >> 3);
  if ((size_t)pl.m * pc_pitch_for(skew_row) * 8 + extra <= cap) {
    pl.skew_mask = ~0;
    pl.pitch = pc_pitch_for(skew_row);
  } else if ((size_t)pl.m * pc_pitch_for(pl.m) * 8 + extra <= cap) {
    pl.skew_mask = 0;
    pl.pitch = pc_pitch_for(pl.m);
  } else if ((size_t)pl.m * pl.m * 8 + extra <= cap) {
    pl.skew_mask = 0;
    pl.pitch = pl.m;
  } else {
    return false;
  }
  pl.lds_bytes = (int)((size_t)pl.m * pl.pitch * 8 + extra);
  const int t = (pl.m * pl.m / 16 + 63) / 64 * 64;  // ~16 complex elements per lane
  pl.threads = t < 64 ? 64 : (t > 1024 ? 1024 : t);
  if ((pl.m * pl.m + pl.threads - 1) / pl.threads > 18) return false;  // (the kernel's load phase holds at most 18 pixels per lane and image)
  return true;
}

// ---- compile-time plans: composite radices, two stages per 1-D transform ----------------------------------------------------
// register slots a butterfly of radix R occupies in the stage routine (pc_plan.hpp), butterflies per lane and group
constexpr int pc_slots(int R) { return R > 8 ? 16 : (R > 4 ? 8 : 4); }
constexpr bool pc_radix_ok(int R) {
  return R == 2 || R == 3 || R == 4 || R == 5 || R == 6 || R == 8 || R == 9 || R == 10 || R == 12 || R == 15 || R == 16;
}
// lines one group of a two-stage pass (Ra, then Rb) covers: stage 0 has m / Ra = Rb butterflies per line, stage 1 has Ra
constexpr int pc_group_lines(int Ra, int Rb) {
  const int g0 = (16 / pc_slots(Ra)) * (64 / Rb), g1 = (16 / pc_slots(Rb)) * (64 / Ra);
  return g0 < g1 ? g0 : g1;
}
// m = Ra * Rb with both radices available; Rb even when m is (the exactness of bin m / 2, above); the pair whose groups cover the
// most lines (one LDS round trip per stage and wave), ties to the smaller Rb (its Rb - 1 twiddles sit in registers)
constexpr bool pc_two_stage_chain(int m, int& Ra, int& Rb) {
  int best = 0;
  for (int a = 2; a <= 16; ++a) {
    if (m % a != 0 || !pc_radix_ok(a)) continue;
    const int b = m / a;
    if (b > 16 || !pc_radix_ok(b) || ((m & 1) == 0 && (b & 1) != 0)) continue;
    const int g = pc_group_lines(a, b);
    if (g > best || (g == best && b < Rb)) {
      best = g;
      Ra = a;
      Rb = b;
    }
  }
  return best > 0;
}

// Tile pitch of a two-stage plan: the one with the fewest LDS cycles in the bank model of tools/design/planned_banks.py (every
// ds_read_b64 / ds_write_b64 of the four passes with the lane groups and bank widths of MI355X_MICROARCH.md) among the pitches that
// keep the workgroups per CU. The generic rule (pitch = 8 mod 16) is tuned for power-of-two radices; with 4 or 6 butterflies per
// line sixteen lines sit side by side in a wave and their rows land on a quarter of the banks (p60: 42 % of the LDS cycles were
// conflicts, profiles/r04_p60_sq_pmc.csv). 0 = keep the generic rule.
#ifndef MOF_PLANNED_PITCH_TABLE
#define MOF_PLANNED_PITCH_TABLE 1
#endif
constexpr int pc_static_pitch(int m) {
  if (!MOF_PLANNED_PITCH_TABLE) return 0;
  switch (m) {
    case 16: return 20;  case 18: return 20;  case 20: return 28;  case 25: return 45;  case 27: return 39;  case 30: return 39;
    case 36: return 42;  case 40: return 44;  case 45: return 52;  case 48: return 55;  case 50: return 59;  case 54: return 71;
    case 60: return 76;  case 64: return 76;  case 72: return 85;  case 75: return 84;  case 80: return 108; case 81: return 103;
    case 90: return 108; case 96: return 108; case 100: return 134; case 108: return 122; case 120: return 152; case 128: return 148;
    default: return 0;
  }
}
// r05: the same model run WITHOUT the column skew (tile column c at c, not c + (c >> 3)). The skew exists for the stride-8 stage writes
// of power-of-two radices; for most other sizes an unskewed tile with the right pitch has as few or fewer conflict cycles
// (tools/design/planned_banks.py, kernel_cost(m, pitch, skew=False)) -- and every element offset of a stage (x + j bpl, (x - k) R + k + p np)
// is then linear in the loop index: an immediate of the LDS instruction instead of a shift and an add per element. Sizes listed here
// take the unskewed pitch; 0 = keep the skewed plan (16, 32, 64, 96, 128 and what has no two-stage chain).
#ifndef MOF_PLANNED_UNSKEWED
#define MOF_PLANNED_UNSKEWED 1
#endif
constexpr int pc_static_pitch_unskewed(int m) {
  if (!MOF_PLANNED_PITCH_TABLE || !MOF_PLANNED_UNSKEWED) return 0;
  switch (m) {
    case 18: return 20;  case 20: return 44;  case 24: return 44;  case 25: return 44;  case 27: return 46;  case 30: return 52;
    case 36: return 41;  case 40: return 44;  case 45: return 52;  case 48: return 55;  case 50: return 69;  case 54: return 58;
    case 60: return 76;  case 72: return 73;  case 75: return 76;  case 80: return 108; case 81: return 103; case 90: return 116;
    case 100: return 103; case 108: return 108; case 120: return 120;
    default: return 0;
  }
}

constexpr PcPlan pc_static_plan(int m) {  // for a 5-smooth m (n = m); threads == 0 when there is none
  PcPlan pl{};
  if (!pc_tile_plan_c(m, pl) || pl.m != m) return PcPlan{};
  int Ra = 0, Rb = 0;
  if (pc_two_stage_chain(m, Ra, Rb)) {
    const int tu = pc_static_pitch_unskewed(m);
    const int tp = pc_static_pitch(m);
    if (tu >= m && (size_t)m * tu * 8 + pc_tile_extra(m) <= 160u * 1024u) {
      pl.pitch = tu;
      pl.skew_mask = 0;
      pl.lds_bytes = (int)((size_t)m * tu * 8 + pc_tile_extra(m));
    } else if (tp > 0 && pl.skew_mask != 0 && tp >= m + ((m - 1) >> 3)) {
      const size_t extra = pc_tile_extra(m);
      if ((size_t)m * tp * 8 + extra <= 160u * 1024u) {
        pl.pitch = tp;
        pl.lds_bytes = (int)((size_t)m * tp * 8 + extra);
      }
    }
    pl.n_stages = 2;
    pl.radix[0] = Ra;
    pl.radix[1] = Rb;
    for (int s = 2; s < 8; ++s) pl.radix[s] = 0;
    pl.radix_packed = (uint32_t)Ra | ((uint32_t)Rb << 5);
    // as many waves as make one group per stage cover a wave's lines, within 16 waves and 18 pixels per lane
    int lines = pc_group_lines(Ra, Rb);
    lines = lines > m ? m : lines;
    int waves = (m + lines - 1) / lines;
    const int min_waves = (m * m + 18 * 64 - 1) / (18 * 64);
    waves = waves < min_waves ? min_waves : waves;
    waves = waves > 16 ? 16 : waves;
    pl.threads = 64 * waves;
  }
  return pl;
}

// waves per SIMD a kernel of this plan should be compiled for: what the LDS lets sit on a CU, at most 4 (128 VGPRs)
constexpr int pc_plan_waves_per_eu(const PcPlan& pl) {
  if (pl.threads <= 0) return 1;
  int wgs = (160 * 1024) / (pl.lds_bytes > 0 ? pl.lds_bytes : 1);
  const int by_waves = 32 / (pl.threads / 64);
  wgs = wgs < by_waves ? wgs : by_waves;
  const int per_eu = (wgs * (pl.threads / 64) + 3) / 4;
  return per_eu < 1 ? 1 : (per_eu > 4 ? 4 : per_eu);
}

}  // namespace mof
