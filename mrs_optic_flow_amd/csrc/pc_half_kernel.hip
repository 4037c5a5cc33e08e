// pc_half_kernel.hip -- K1h: the fused per-patch phase correlation on a HALF-size LDS tile, from compile-time plans.
//
// Written for the patch sizes whose padded transform size M = cv::getOptimalDFTSize(samplePointSize) is even and lies in (135, 192]
// -- FftMethod takes any samplePointSize (/root/reference/src/FftMethod.cpp:1706-1720, :1829-1866) and every patch goes through
// cv::phaseCorrelate (:1836) -- i.e. the sizes whose M x M COMPLEX tile (pc_kernel_generic.hip's packed z = cur + i prev) no longer
// fits one CU's 160 KB of LDS but whose HALF tile does; until r05 those ran the four-kernel pipeline through HBM scratch
// (pc_large_kernel.hip), 3.5 x slower per pixel than the in-LDS kernels on either side of M = 135. It then turned out to beat the
// full-tile kernels where TWO or more of its workgroups fit a CU: it is the default for samplePointSize 120 (the reference's own
// default, config/default.yaml:31-32: 1.10 -> 1.30 M pairs/s at the reference geometry) and for padded sizes 60 / 96 / 100.
//
// The real formulation (the one pc_seq_half.hip uses for videos and pc_large_kernel.hip streams through HBM) never holds more
// than M/2 x M complex values: each image is transformed on its own,
//   rows    two real rows (2j, 2j+1) per complex line j, staged as raw bytes and converted by the first row stage (HalfRawSrc)
//           -> 1-D transforms along x
//   columns M/2 complex columns of length M; the UNTANGLE of the row pairs (R_2j[u] = (Z[u] + conj Z[M-u]) / 2, R_2j+1[u] = (Z[u] -
//           conj Z[M-u]) / 2i, kept doubled; the real bins u = 0 and u = M/2 of a row share column 0) is the SOURCE of the first column
//           stage (HalfUntangleSrc, a workgroup barrier between that stage's reads and writes); column 0 = the two real columns
//           u = 0 | M/2, separated with the partner bin M - v by its owner
// and the spectra never pass through LDS: the last stage of the previous image's column pass keeps its outputs in REGISTERS
// (HalfSaveSink), the last stage of the current image's pass meets them there and writes the conjugated normalised cross-power
// spectrum instead (HalfXpowSink: real-only-slot rule and all, pc_common.hpp; mulSpectrums :1494, magSpectrums :70-168, divSpectrums
// :1086-1251) -> inverse columns, no workgroup barrier since the forward pass -> Hermitian row PAIRS (2j, 2j+1) as one complex
// transform each, the pairing as the source of their first stage (HalfPairSrc), the first maximum of the fft-shifted surface
// (fftShift :1257-1323, minMaxLoc :1539) riding the last stage's registers (HalfScanSink), and the 5 x 5 fp64 centroid + gate
// (:1337-1383, :1838-1856) by one wave. Every 1-D transform is a planned Stockham chain run by ONE wave on lines it owns
// (pc_plan.hpp: pass_lines_static, stage_rt / stage_rt_ng; compile-time radices, two stages where the size allows).
//
// Tile layout (complex elements, P = pitch of a physical line, even; the skew where the size's plan keeps it):
//   rows layout   line j, element x          at  j P + x + (x >> 3)                      [raw bytes, row passes, result surface]
//   spec layout   logical row r, column u    at  r (P/2) + u + (u >> 3)                  [half spectra, column passes]
// logical rows 2j | 2j+1 are the two halves of physical line j; a column walk is linear in r (stride P/2).
// Zero padding (M > samplePointSize), constant patches (exact-zero spectra: `box_zeros`, the closed-form degenerate answer) as
// pc_large_kernel.hip / pc_kernel_generic.hip handle them. cv::phaseCorrelate's peak model on gray or BGR8 frames; the long-range
// mode and the OpenCL peak model of these sizes stay on the kernels they had.
// MOF_FFT_HALF=0 keeps every size on its r04 kernel, MOF_FFT_HALF=1 also routes 64 and 128 through this one (A/B: the tuned packed
// kernels win there, DESIGN.md section 4 "K1h").

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"
#include <type_traits>
#include "pc_plan.hpp"
#include "pc_plan_build.hpp"

namespace mof {

namespace {

// ---- the compile-time plan ------------------------------------------------------------------------------------------------
struct HalfPlan {
  PcPlan P{};    // m, n = m, n_stages, radix[]
  int lpw = 0;   // lines per wave: row pairs, columns, row pairs again
  int waves = 0;
  int pitch = 0;
  int skew = 0;  // 1: element x of a line sits at x + (x >> shift) (both layouts); 0 where only the unskewed tile fits (M = 192)
  int rskew = 0; // r06 (A/B): row skew -- tile row r starts rskew * (r >> 4) complex elements later (pc_plan.hpp: Walk::lrs / ers)
  int shift = 3; // the skew's shift: 4 for the radix-16-first sizes 96 / 128 / 160 (tools/design/half_lanes.py: stage-0 outputs 16 x + p of the
                 // eight butterflies of a line land in eight different banks mod 16 only with x + (x >> 4))
  int lds_bytes = 0;
  int wgs_per_cu = 0;
  bool ok = false;
};

#ifndef MOF_HALF_RSKEW  // (A/B) the row skew of every instantiation, complex elements per 16 rows; 0 = none (the product)
#define MOF_HALF_RSKEW 0
#endif
constexpr int half_rskew(int m) { return MOF_HALF_RSKEW; }
// complex elements the row skew adds behind the tile: rskew * ((M - 1) >> 4) for the last row, rounded up
constexpr int half_rskew_room(int m) { return half_rskew(m) * ((m >> 4) + 1); }
// LDS behind the tile: the row skew's room, twiddles (m complex), 16 (value, index) slots, 16 flag words
constexpr size_t half_extra(int m) { return sizeof(float) * 2 * ((size_t)m + half_rskew_room(m)) + 16 * 8 + 64; }

constexpr int half_stage_lines(int m, int R) {  // lines one group of a stage covers (pc_plan.hpp: stage_rt)
  const int bpl = m / R, nb = 16 / pc_slots(R);
  return bpl <= 64 ? nb * (64 / bpl) : 1;
}

#ifndef MOF_HALF_PITCH  // (A/B) force the pitch of every instantiation; 0 = the rule below
#define MOF_HALF_PITCH 0
#endif
#ifndef MOF_HALF_SKEW  // (A/B, with MOF_HALF_PITCH) 0 / 1: force the unskewed / skewed layout; -1 = the rule
#define MOF_HALF_SKEW (-1)
#endif
#ifndef MOF_HALF_PITCH_TABLE  // 0: the generic pitch rule for every size (A/B)
#define MOF_HALF_PITCH_TABLE 1
#endif
// (pitch, skew) with the fewest LDS cycles in the bank model of tools/design/half_banks.py (every ds_read_b64 / ds_write_b64 of a
// patch pair with the lane groups and bank widths of MI355X_MICROARCH.md) among the pitches that keep the workgroups per CU; 0 = the
// rule below. Where the unskewed tile is within a few per cent of the best it is preferred: its element offsets are immediates.
#ifndef MOF_HALF_SHIFT  // (A/B) force the skew shift of every skewed instantiation; 0 = the table / 3
#define MOF_HALF_SHIFT 0
#endif
constexpr int half_table_shift(int m) {
  if (MOF_HALF_SHIFT > 0) return MOF_HALF_SHIFT;
  if (!MOF_HALF_PITCH_TABLE) return 3;
  switch (m) {
    case 96: case 128: case 160: case 192: return 4;  // (192: the skewed tile only fits with this shift: 204 x 96 lines)
    default: return 3;
  }
}
constexpr int half_table_pitch(int m, int* skew) {
  if (!MOF_HALF_PITCH_TABLE) return 0;
  switch (m) {
    case 96:  // (with shift 4: x1.09 of the conflict-free LDS cycles in the model against x1.40 at the rule's 120 with shift 3)
      if (half_table_shift(96) != 4) return 0;
      *skew = 1;
      return 104;
    case 60: *skew = 0; return 68;    // (box: 68 / 76 unskewed 1.36 M at p60, the rule's 72 skewed 1.29 M; profiles/r05_half_pitch60_sweep.txt)
    case 120: *skew = 0; return 136;  // (measured too, tools/sweep_half_pitch.sh, profiles/r05_half_pitch120_sweep.txt: 136 unskewed 1.14 M, 120 1.13 M, 152 1.13 M, 128 0.93 M; skewed 136 / 152: 1.05 M)
    case 144: *skew = 1; return 202;
    case 160:  // (box, shift 4: 172 / 170 / 200 729 k at l160, the rule's 184 725 k; shift 3 at 184: 712 k -- profiles/r05_half_pitch160_shift_sweep.txt)
      if (half_table_shift(160) != 4) return 0;
      *skew = 1;
      return 172;
    case 150: *skew = 0; return 180;
    case 162: *skew = 0; return 186;
    case 180: *skew = 0; return 184;
    default: return 0;
  }
}
#ifndef MOF_HALF_SWAP160  // the product's choice for M = 160 (0 until the A/B says otherwise)
#define MOF_HALF_SWAP160 0
#endif
#ifndef MOF_HALF_SWAP  // (A/B) 1: a two-stage chain's radices in the other order (the last one must stay even: pc_plan_build.hpp's exactness rule)
#define MOF_HALF_SWAP 0
#endif
// sizes whose two-stage chain runs the SMALLER radix first (r06, tools/design/half_lanes.py in counter terms + same-box A/B): 160 = 10 x 16 --
// with 16 first the stage-0 outputs 16 x + p of neighbouring butterflies of a column sit 16 rows = 16 P/2 complex = 0 (mod 32 dwords) apart
// for ANY pitch, a 3-way conflict on every first-stage column write
constexpr bool half_swapped(int m) { return MOF_HALF_SWAP != 0 ? true : (m == 160 && MOF_HALF_SWAP160); }
constexpr HalfPlan half_plan(int m) {
  HalfPlan hp{};
  if (m < 16 || m > 192 || (m & 1)) return hp;
  PcPlan pl{};
  if (!pc_line_plan_c(m, pl) || pl.m != m) return hp;  // 5-smooth sizes only (n = m); the generic radix chain as a start
  int Ra = 0, Rb = 0;
  if (pc_two_stage_chain(m, Ra, Rb)) {
    if (half_swapped(m) && (Ra & 1) == 0) {
      const int t = Ra;
      Ra = Rb;
      Rb = t;
    }
    pl.n_stages = 2;
    pl.radix[0] = Ra;
    pl.radix[1] = Rb;
    for (int s = 2; s < 8; ++s) pl.radix[s] = 0;
  } else {
    // three stages a b c, c even (the exactness rule of pc_plan_build.hpp), the first split that exists
    bool found = false;
    for (int a = 16; a >= 2 && !found; --a) {
      if (m % a != 0 || !pc_radix_ok(a)) continue;
      for (int b = 16; b >= 2 && !found; --b) {
        if ((m / a) % b != 0 || !pc_radix_ok(b)) continue;
        const int c = m / a / b;
        if (c < 2 || c > 16 || !pc_radix_ok(c) || (c & 1)) continue;
        pl.n_stages = 3;
        pl.radix[0] = a;
        pl.radix[1] = b;
        pl.radix[2] = c;
        for (int s = 3; s < 8; ++s) pl.radix[s] = 0;
        found = true;
      }
    }
    // (else: the generic chain of pc_line_plan_c stays)
  }
  pl.radix_packed = 0;
  for (int s = 0; s < pl.n_stages; ++s) pl.radix_packed |= (uint32_t)pl.radix[s] << (5 * s);
  const int H = m / 2;
  int g = 1 << 20;
  for (int s = 0; s < pl.n_stages; ++s) {
    const int gl = half_stage_lines(m, pl.radix[s]);
    g = gl < g ? gl : g;
  }
  g = g > H ? H : g;
  const int k = (H + g * 16 - 1) / (g * 16);
  hp.lpw = g * k;
  // M = 96: whole groups (8 lines) make SIX-wave workgroups, and at four waves per SIMD only two of them fit a CU (12 waves: a third
  // would put a fifth wave on two SIMDs). Six lines per wave = eight emptier waves, two workgroups = 16 waves: p96 892 -> 907 k pairs/s
  // same-box (profiles/r05_half_lpw96_ab.txt; four lines = 12 waves: 600 k)
  if (m == 96) hp.lpw = 6;
#ifdef MOF_HALF_LPW  // (A/B, with MOF_HALF_ONLY) lines per wave: fewer than a whole stage group = more, emptier waves
  hp.lpw = MOF_HALF_LPW;
#endif
  hp.waves = (H + hp.lpw - 1) / hp.lpw;
  // pitch: even, room for both skews; P/2 = 4 (mod 8) spreads the rows of a column walk over the banks (8 rows x 4 columns per
  // 32-lane read group) -- the first such pitch that still fits, else the smallest; without the skew where nothing else fits
  const size_t cap = 160u * 1024u;
  const int sh = half_table_shift(m);
  hp.shift = sh;
  int p = 0, skew = 1;
  for (; skew >= 0; --skew) {
    int pmin = m + (skew ? ((m - 1) >> sh) : 0);
    const int hmin = 2 * (H + (skew ? ((H - 1) >> sh) : 0));
    pmin = pmin < hmin ? hmin : pmin;
    pmin += pmin & 1;
    p = pmin;
    while ((p / 2) % 8 != 4) p += 2;
    if ((size_t)H * p * 8 + half_extra(m) > cap) p = pmin;
    if (MOF_HALF_PITCH > 0 && MOF_HALF_PITCH >= pmin && (MOF_HALF_PITCH & 1) == 0) p = MOF_HALF_PITCH;
    if (MOF_HALF_SKEW >= 0 && skew != MOF_HALF_SKEW) continue;
    if ((size_t)H * p * 8 + half_extra(m) <= cap) break;
  }
  if (skew < 0) return hp;
  {
    int tsk = 0;
    const int tp = half_table_pitch(m, &tsk);
    if (tp > 0 && MOF_HALF_PITCH == 0 && (size_t)H * tp * 8 + half_extra(m) <= cap) {
      p = tp;
      skew = tsk;
    }
  }
  hp.skew = skew;
  hp.rskew = half_rskew(m);
  hp.pitch = p;
  hp.lds_bytes = (int)((size_t)H * p * 8 + half_extra(m));
  int wgs = (int)(cap / (size_t)hp.lds_bytes);
  const int by_waves = 32 / hp.waves;
  hp.wgs_per_cu = wgs < by_waves ? wgs : by_waves;
  pl.threads = 64 * hp.waves;
  pl.pitch = p;
  pl.skew_mask = skew ? ~0 : 0;
  pl.hermitian = 1;
  pl.lds_bytes = hp.lds_bytes;
  hp.P = pl;
  hp.ok = true;
  return hp;
}

// most waves per SIMD the launch bounds ask the register allocator for: 4 (128 VGPRs). Measured at 5 (96 VGPRs; profiles/r05_half_wpe5_ab.txt):
// M = 96 -- six-wave workgroups, two per CU at 4, three at 5 -- gains 3.5 % (p96 893 -> 925 k pairs/s) but spills 26 VGPRs, and the scratch
// traffic shows as 2.2 x the algorithmic bytes on the fabric counters (1.0 x without); p60 loses 6 %, 100 loses 3 %, 6 waves at 96 lose 21 %.
// Not adopted: the clean 1.0 x is worth more than 3.5 %. MOF_HALF_WPE_CAP forces a value (A/B).
constexpr int half_wpe_cap(int m) {
#ifdef MOF_HALF_WPE_CAP
  return MOF_HALF_WPE_CAP;
#else
  return 4;
#endif
}
template <int MS>
struct HalfPlanOf {
  static constexpr HalfPlan HP = half_plan(MS);
  static constexpr PcPlan P = HP.P;  // (what pass_lines_static reads)
  static_assert(HP.ok, "no half-tile plan for this size");
  static constexpr int T = 64 * HP.waves;
  // waves per SIMD the registers must allow: what the LDS lets sit on a CU, at most 4 (128 VGPRs)
  static constexpr int WPE_ = (HP.wgs_per_cu * HP.waves + 3) / 4;
  static constexpr int WPE = WPE_ < 1 ? 1 : (WPE_ > half_wpe_cap(MS) ? half_wpe_cap(MS) : WPE_);
};

// arg-max as the sink of the inverse row pass's last stage: line j, element x carries the surface at (2j, x) and (2j + 1, x)
struct HalfScanSink {
  static constexpr bool active = true;
  Best* best;
  int m, H;
  __device__ __forceinline__ void operator()(int j, int x, cf v) const {
    const int xs = x + H >= m ? x + H - m : x + H;
    const int y1 = 2 * j, ys1 = y1 + H >= m ? y1 + H - m : y1 + H, ys2 = y1 + 1 + H >= m ? y1 + 1 + H - m : y1 + 1 + H;
    *best = better(*best, Best{v.x, ys1 * m + xs});
    *best = better(*best, Best{v.y, ys2 * m + xs});
  }
};

// raw pixel staging (as the tuned kernels, pc_passes.hpp raw_store): a chunk's 4 + 4 pixels (rows 2j, 2j + 1) leave as ONE
// ds_write_b64 of interleaved bytes (a0 b0 a1 b1 ..) into the first 2 M bytes of the line's own tile span, and the first row stage
// converts them on its way into the butterfly (u8 -> f32: convertTo, :1805-1806) -- four ds_write_b64 of floats per chunk less
#ifndef MOF_HALF_RAW
#define MOF_HALF_RAW 1
#endif
struct HalfRawSrc {
  static constexpr bool active = true;
  int pitch;  // complex elements per line
  int rs;     // row skew (complex elements per 8 lines)
  __device__ __forceinline__ cf operator()(const cf* z, int l, int e) const {
    typedef const volatile uint16_t __attribute__((address_space(3))) * lds_u16_ptr;
    const uint32_t ab = *(lds_u16_ptr)(reinterpret_cast<const unsigned char*>(z + l * pitch + rs * (l >> 3)) + 2 * e);
    return {(float)(ab & 0xffu), (float)(ab >> 8)};
  }
};

// the pairing of the inverse row pass (spec -> rows layout) as the SOURCE of its first stage: element e of line j is built from the two
// spec rows 2j | 2j + 1 on the way into the butterfly -- the sweep of its own (a read and a write of the half tile) disappears
#ifndef MOF_HALF_PAIR_SRC
#define MOF_HALF_PAIR_SRC 1
#endif
struct HalfPairSrc {
  static constexpr bool active = true;
  int p2, m, skm, sh, rs;
  __device__ __forceinline__ cf operator()(const cf* z, int l, int e) const {
    const int H = m >> 1;
    const int u = e < H ? e : (e == H ? 0 : m - e);
    const int o = u + ((u >> sh) & skm) + rs * (l >> 3);  // (rows 2l and 2l + 1 share (2l) >> 4 = l >> 3)
    const cf d1 = lds_read(&z[(2 * l) * p2 + o]), d2 = lds_read(&z[(2 * l + 1) * p2 + o]);
    if (e == 0) return {d1.x, d2.x};
    if (e == H) return {d1.y, d2.y};
    if (e < H) return {d1.x - d2.y, d1.y + d2.x};
    return {d1.x + d2.y, d2.x - d1.y};
  }
};

// the untangle (rows -> spec layout) as the SOURCE of the forward column pass's first stage, for the sizes whose wave runs ONE group of
// that stage (64, 96, 120, 128, 144): row r = 2j + i of column u is formed from Z_j[u] and Z_j[M - u] of the ROWS layout on its way into
// the butterfly; the stage's outputs land in the spec layout, over rows-layout bins that OTHER waves' columns read -- so a workgroup
// barrier stands between the stage's reads and its writes (WorkgroupSync). The sweep of its own (17 reads + 16 writes per lane and
// image) disappears for 15 more reads and one more barrier.
#ifndef MOF_HALF_UNTANGLE_SRC
#define MOF_HALF_UNTANGLE_SRC 1
#endif
struct HalfUntangleSrc {
  static constexpr bool active = true;
  int p, m, skm, sh, rs;
  __device__ __forceinline__ cf operator()(const cf* z, int l, int e) const {  // column l, row e
    const cf* line = z + (e >> 1) * p + rs * (e >> 4);
    const int H = m >> 1, um = l == 0 ? H : m - l;
    const cf zk = lds_read(&line[l + ((l >> sh) & skm)]), zm = lds_read(&line[um + ((um >> sh) & skm)]);
    cf A, B;
    if (l == 0) {  // the real bins u = 0 and u = M/2 of a row share column 0
      A = {2.f * zk.x, 2.f * zm.x};
      B = {2.f * zk.y, 2.f * zm.y};
    } else {
      untangle2(zk, zm, &A, &B);
    }
    return (e & 1) ? B : A;
  }
};

// The spectra never pass through LDS (sizes whose wave runs ONE group of the LAST column stage too -- the same 64, 96, 120, 128, 144):
// the last stage of the previous image's forward column pass keeps its outputs in registers (HalfSaveSink; only column 0 is written:
// it still has to be taken apart with its partner bins), and the last stage of the current image's pass meets them there -- same
// lane, same register slot -- and writes the CONJUGATED CROSS-POWER SPECTRUM instead of the spectrum (HalfXpowSink). Gone: the copy of
// the previous spectrum into registers, its tile write, and the cross-power sweep's read and write of the wave's columns.
#ifndef MOF_HALF_XPOW_SINK
#define MOF_HALF_XPOW_SINK 1
#endif
#ifndef MOF_HALF_MAX_GROUPS  // groups of a fused stage a wave may hold in registers at once (1: the r05 first form, sizes up to 144 only)
#define MOF_HALF_MAX_GROUPS 2
#endif
struct HalfSaveSink {
  static constexpr bool active = true;
  static constexpr bool transforms = true;
  cf* pv;
  int R;
  __device__ __forceinline__ cf transform(int l, int, cf v, int b, int p, bool* wr) const {
    pv[b * R + p] = v;
    *wr = l == 0;
    return v;
  }
};
template <bool KEEP = false>  // KEEP (the sequence form): this image's spectrum replaces the previous one's in the registers
struct HalfXpowSink {
  static constexpr bool active = true;
  static constexpr bool transforms = true;
  cf* pv;
  int R, H;
  bool box_zeros;
  int q;  // box_zeros: the exact-zero lines of the constant box are the multiples of q (pc_common.hpp, box_zero_period)
  __device__ __forceinline__ cf transform(int l, int o, cf v, int b, int p, bool* wr) const {
    *wr = true;
    cf C = cross_power_ab(v, pv[b * R + p], false);
    if constexpr (KEEP) pv[b * R + p] = v;
    if (box_zeros && (box_zero_line(o, q) || box_zero_line(l, q))) C = {0.f, 0.f};
    return l == 0 ? v : cf{C.x, -C.y};  // (column 0 leaves as it is: taken apart, crossed and put together again by its owner)
  }
};

#ifndef MOF_HALF_FAST_LOAD  // 0: the general pixel loader for every patch (A/B)
#define MOF_HALF_FAST_LOAD 1
#endif
#ifndef MOF_HALF_PREFETCH  // 0: the current image is loaded after the previous image's column pass (A/B)
#define MOF_HALF_PREFETCH 1
#endif
#ifndef MOF_HABL  // diagnostic builds (results wrong by design): 1 no transform passes, 2 no cross-power, 3 no pixel loads, 4 no previous image
                  // (what a sequence form that keeps a frame's spectrum for the next pair could save at most)
#define MOF_HABL 0
#endif

// SEQ (the video form, mof_fft_process_sequence_device): blockIdx.z is a RUN of `run` consecutive pairs of one patch position, frame f at
// a.cur + f * a.cur_stride. The run's first frame is transformed as the previous image of the pair form; after that every frame is the
// CURRENT image once, and its spectrum -- the forward column pass's last-stage outputs, already in the lanes' registers where the
// cross-power meets the previous one -- simply stays there for the next pair (HalfXpowSink<true>): one image transform per pair instead of
// two (a run re-transforms its first frame: 1 / run more).
template <int CH, int MS, bool SEQ = false>
__global__ void __launch_bounds__(HalfPlanOf<MS>::T, HalfPlanOf<MS>::WPE) pc_half_kernel(PcArgs a, int n, int n_pairs, int run) {
  using SP = HalfPlanOf<MS>;
  constexpr HalfPlan HP = SP::HP;
  constexpr int M = MS, H = M / 2, P = HP.pitch, P2 = P / 2, T = SP::T, WAVES = HP.waves, LPW = HP.lpw;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_h[];
  cf* z = reinterpret_cast<cf*>(smem_h);
  constexpr int RSK = HP.rskew;
  cf* tw = z + (size_t)H * P + half_rskew_room(M);
  Best* red = reinterpret_cast<Best*>(tw + M);
  int* flags = reinterpret_cast<int*>(red + 16);  // [0] cur differs from its first pixel, [1] prev does, [2] C_dc bits, [3] / [4] first pixel of cur / prev
  int tid = threadIdx.x, lane = tid & 63;  // (not const: the sequence form hides them from the optimiser once per pair, below)
  const int wave = tid >> 6;

  constexpr int SKM = HP.skew ? ~0 : 0;
  constexpr int SH = HP.shift;
  // lines 0 2 1 3 in the later row stages where two lines of a 32-lane read half overlap in the banks: neighbouring lines P complex = 2 P
  // dwords apart, a line's 16 complex = 32 dwords -- conflict-free iff 2 P = 32 (mod 64); lines two apart: 4 P = 32 (mod 64) iff P = 8 (mod 16)
  constexpr int HALF_LINE_PERM = ((2 * P) % 64 != 32 && (4 * P) % 64 == 32) ? 1 : 0;
  auto rows_at = [&](int j, int x) -> int { return j * P + RSK * (j >> 3) + x + ((x >> SH) & SKM); };
  auto spec_at = [&](int r, int u) -> int { return r * P2 + RSK * (r >> 4) + u + ((u >> SH) & SKM); };

  // ---- patch origin (one workgroup per patch on a 3-D grid: column, row, pair)
  const int px0 = a.origin_x + (int)blockIdx.x * a.stride_x, py0 = a.origin_y + (int)blockIdx.y * a.stride_y;
  const size_t poff = (size_t)py0 * a.pitch + (size_t)(CH * px0);
  const size_t pair0 = SEQ ? (size_t)blockIdx.z * (size_t)run : (size_t)blockIdx.z;  // first (only) pair of this workgroup
  const uint8_t* prev = SEQ ? a.cur + pair0 * a.cur_stride + poff : a.prev + pair0 * a.prev_stride + poff;
  const uint8_t* cur = SEQ ? prev + a.cur_stride : a.cur + pair0 * a.cur_stride + poff;
  size_t p = (pair0 * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;
  int pairs_left = SEQ ? ((size_t)n_pairs - pair0 < (size_t)run ? (int)((size_t)n_pairs - pair0) : run) : 1;

  if (tid < 2) flags[tid] = 0;
#pragma unroll 1
  for (int k = tid; k < M; k += T) tw[k] = {a.twiddles[2 * k], a.twiddles[2 * k + 1]};
  __syncthreads();  // flags zeroed, twiddles in place

  // first stage of a pass (pc_plan.hpp, stage_rt): R0 butterflies ... one group covers GROUP0 lines
  constexpr int R0 = SP::P.radix[0], BPL0 = M / R0, GROUP0 = (16 / pc_slots(R0)) * (64 / BPL0);
  // the wave's columns as NG0 compile-time groups of the first column stage (stage_rt_ng; r05: one group -- 64 .. 144 -- or two -- the
  // sizes of 150 .. 192, 32 complex values per lane in flight)
  constexpr int NG0 = (LPW + GROUP0 - 1) / GROUP0;
  constexpr bool UFUSE = MOF_HALF_UNTANGLE_SRC != 0 && NG0 <= MOF_HALF_MAX_GROUPS;
  constexpr int RL = SP::P.radix[SP::P.n_stages - 1], BPLL = M / RL, NBL = 16 / pc_slots(RL), GROUPL = BPLL <= 64 ? NBL * (64 / BPLL) : 0;
  constexpr int NGL = GROUPL > 0 ? (LPW + GROUPL - 1) / GROUPL : 99;  // ... and of the last one (162's radix-2 stage has long lines: none)
  constexpr bool XSINK = MOF_HALF_XPOW_SINK != 0 && UFUSE && NGL <= MOF_HALF_MAX_GROUPS && SP::P.n_stages >= 2;
  static_assert(!UFUSE || (WAVES - 1) * LPW < H, "every wave owns a line: the barrier inside the fused stage is met by all");
  static_assert(!SEQ || XSINK, "the sequence form keeps the spectrum in the cross-power sink's registers");
  // wave w owns lines [l0, l0 + nl): row pairs in the row passes, columns in the column passes
  const int l0 = wave * LPW;
  const int nl = H - l0 < 0 ? 0 : (H - l0 > LPW ? LPW : H - l0);
  const Walk rows = {P, 1, 0, SKM, 0, HALF_LINE_PERM, SH, RSK, 0}, cols = {1, P2, SKM, 0, 1, 0, SH, 0, RSK};

  auto px_gray = [&](const uint8_t* q) -> uint32_t {  // four pixels -> four gray bytes
    if constexpr (CH == 1) {
      uint32_t w;
      __builtin_memcpy(&w, q, 4);
      return w;
    } else {
      uint32_t w[3];
      __builtin_memcpy(w, q, 12);
      uint32_t g = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) {
        const int i = 3 * b;
        const uint32_t c0 = (w[i >> 2] >> (8 * (i & 3))) & 0xffu, c1 = (w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xffu,
                       c2 = (w[(i + 2) >> 2] >> (8 * ((i + 2) & 3))) & 0xffu;
        g |= rgb2gray_fixed(c0, c1, c2) << (8 * b);
      }
      return g;
    }
  };
  auto px_one = [&](const uint8_t* q) -> uint32_t {
    if constexpr (CH == 1) return q[0];
    else return rgb2gray_fixed(q[0], q[1], q[2]);
  };

  // ---- one image: pixels of the wave's lines -> LDS (u8 -> f32: convertTo, :1805-1806; zeros beyond n x n: copyMakeBorder of
  //      cv::phaseCorrelate), row transforms, untangle into the spec layout. Wave-local throughout. Chunk q = lane + 64 k of the wave's
  //      lines = four pixels of rows 2j and 2j + 1: chunk_load brings them into two registers, chunk_commit stages them in the tile.
  constexpr int CPR = (M + 3) / 4;                     // four-pixel chunks per row of the padded tile
  constexpr int NCH = (LPW * CPR + 63) / 64;           // chunks per lane: both rows of a line ride one chunk
  constexpr int SB = CH == 1 ? NCH : (NCH + 1) / 2;    // loads in flight per lane and sub-batch
  // The patch fills its tile (n = M, M a multiple of four -- the bench's 60 / 96 / 160): no pixel of a chunk lies outside, and the chunk
  // coordinates (line, chunk column) of q = lane + 64 k follow from the lane's own by compile-time steps with one wrap -- the general
  // form spends ~70 instructions per chunk on q / CPR, q % CPR, a 64-bit row offset and four bounds; this one a dozen. A scalar
  // branch: n is a kernel argument. Same-box (profiles/r05_half_fastload_ab.txt): l160 +2 %, p60 +1.7 %, p96 +-0, ref -0.7 % -- the
  // loader's arithmetic mostly hides behind its own loads; M = 120 keeps the general form.
  const bool fills = MOF_HALF_FAST_LOAD != 0 && MOF_HALF_RAW != 0 && (M % 4 == 0) && M != 120 && n == M;
  int li0 = lane / CPR, xc0 = lane - li0 * CPR;  // (not const: SEQ, below)
  // (FILLS rides in as a type: the scalar branch on `fills` stands ONCE around a whole loop of chunks -- a branch per chunk puts every
  // load into a basic block of its own and cost 6 - 8 % at p60 / p96 / l160)
  auto chunk_load = [&](auto fills_tag, int k, const uint8_t* img, uint32_t* pa, uint32_t* pb) {
    constexpr bool FILLS = decltype(fills_tag)::value;
    *pa = *pb = 0u;
    if (k >= NCH || MOF_HABL == 3) return;
    if constexpr (FILLS) {
      const int DL = (64 * k) / CPR, DX = (64 * k) % CPR;
      const bool wrap = xc0 + DX >= CPR;
      const int li = li0 + DL + (wrap ? 1 : 0);
      if (li < nl) {
        const uint32_t pitch = (uint32_t)a.pitch;
        const uint32_t off = (uint32_t)(2 * (l0 + li0)) * pitch + (uint32_t)(4 * CH * xc0) + (uint32_t)(2 * DL) * pitch + (uint32_t)(4 * CH * DX) +
                             (wrap ? 2u * pitch - (uint32_t)(4 * CH * CPR) : 0u);
        const uint8_t* qa = img + off;
        *pa = px_gray(qa);
        *pb = px_gray(qa + pitch);
      }
    } else {
      const int q = lane + 64 * k, li = q / CPR, x0 = 4 * (q % CPR), y = 2 * (l0 + li);
      if (li < nl && x0 < n) {
        const uint8_t* qa = img + (size_t)y * a.pitch + (size_t)CH * x0;
        if (x0 + 3 < n) {
          if (y < n) *pa = px_gray(qa);
          if (y + 1 < n) *pb = px_gray(qa + a.pitch);
        } else {  // the last chunk of a row whose length is not a multiple of four: the pixels inside the patch
          for (int b = 0; x0 + b < n; ++b) {
            if (y < n) *pa |= px_one(qa + CH * b) << (8 * b);
            if (y + 1 < n) *pb |= px_one(qa + a.pitch + CH * b) << (8 * b);
          }
        }
      }
    }
  };
  auto chunk_commit = [&](auto fills_tag, int k, uint32_t va, uint32_t vb, uint32_t pat, uint32_t* diff) {
    constexpr bool FILLS = decltype(fills_tag)::value;
    if (k >= NCH) return;
    typedef uint32_t u2 __attribute__((ext_vector_type(2)));
    typedef u2 __attribute__((address_space(3))) * lds_u2_ptr;
    if constexpr (FILLS) {
      const int DL = (64 * k) / CPR, DX = (64 * k) % CPR;
      const bool wrap = xc0 + DX >= CPR;
      const int li = li0 + DL + (wrap ? 1 : 0);
      if (li < nl) {
        *diff |= (va ^ pat) | (vb ^ pat);
        u2 d;
        d.x = __builtin_amdgcn_perm(vb, va, 0x05010400u);
        d.y = __builtin_amdgcn_perm(vb, va, 0x07030602u);
        const uint32_t lds0 = (uint32_t)((l0 + li0) * (P * 8) + 8 * xc0);
        const uint32_t rsb = RSK != 0 ? (uint32_t)(8 * RSK * ((l0 + li) >> 3)) : 0u;  // (the row skew of line l0 + li)
        *(lds_u2_ptr)(reinterpret_cast<unsigned char*>(z) + (lds0 + rsb + (uint32_t)(DL * (P * 8) + 8 * DX) + (wrap ? (uint32_t)(P * 8 - 8 * CPR) : 0u))) = d;
      }
    } else {
      const int q = lane + 64 * k, li = q / CPR, x0 = 4 * (q % CPR), y = 2 * (l0 + li);
      if (li < nl) {
        if (x0 < n) {
          const uint32_t inside = x0 + 3 < n ? 0xffffffffu : (1u << (8 * (n - x0))) - 1u;
          if (y < n) *diff |= (va ^ pat) & inside;
          if (y + 1 < n) *diff |= (vb ^ pat) & inside;
        }
        cf* line = z + (l0 + li) * P + RSK * ((l0 + li) >> 3);
        if constexpr (MOF_HALF_RAW) {
          u2 d;
          d.x = __builtin_amdgcn_perm(vb, va, 0x05010400u);
          d.y = __builtin_amdgcn_perm(vb, va, 0x07030602u);
          *(lds_u2_ptr)(reinterpret_cast<unsigned char*>(line) + 2 * x0) = d;  // (a partial last chunk spills past 2 M bytes: inside the line, never read)
        } else {
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int x = x0 + b;
            if (x < M) line[x + ((x >> SH) & SKM)] = {(float)((va >> (8 * b)) & 0xffu), (float)((vb >> (8 * b)) & 0xffu)};
          }
        }
      }
    }
  };
  // flags of the image whose pixels were just staged, then its row transforms
  auto rows_of = [&](uint32_t diff, uint32_t first, int which) {
    if (__builtin_amdgcn_ballot_w64(diff != 0u) != 0ull && lane == 0) flags[which] = 1;
    if (tid == 0) flags[3 + which] = (int)first;
    wave_sync();
    if (nl > 0 && MOF_HABL != 1) {
      if constexpr (MOF_HALF_RAW) pass_lines_static<SP, 0, 1, NoSink, HalfRawSrc>(z, tw, rows, l0, nl, lane, false, NoSink{}, HalfRawSrc{P, RSK});
      else pass_lines_static<SP>(z, tw, rows, l0, nl, lane, false);
    }
    if constexpr (UFUSE) return;  // (the untangle rides the forward column pass: HalfUntangleSrc)
    // untangle: line j = rows 2j + i (2j + 1): R_2j[u] = (Z[u] + conj Z[M-u]) / 2, R_2j+1[u] = (Z[u] - conj Z[M-u]) / 2i, kept
    // DOUBLED; the real bins u = 0 and u = M/2 of a row share its column 0. Every read of a line before its first write.
    constexpr int KU = (LPW * H + 63) / 64;
    cf zk[KU], zm[KU], zh[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int q = lane + 64 * k, li = q / H, u = q % H;
      zk[k] = zm[k] = zh[k] = cf{0.f, 0.f};
      if (li < nl) {
        zk[k] = lds_read(&z[rows_at(l0 + li, u)]);
        zm[k] = lds_read(&z[rows_at(l0 + li, u == 0 ? 0 : M - u)]);
        if (u == 0) zh[k] = lds_read(&z[rows_at(l0 + li, H)]);
      }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int q = lane + 64 * k, li = q / H, u = q % H;
      if (li < nl) {
        cf A, B;
        untangle2(zk[k], zm[k], &A, &B);
        if (u == 0) {
          A = {A.x, 2.f * zh[k].x};
          B = {B.x, 2.f * zh[k].y};
        }
        z[spec_at(2 * (l0 + li), u)] = A;
        z[spec_at(2 * (l0 + li) + 1, u)] = B;
      }
    }
  };
  // one image, loads and staging back to back (sub-batches of SB chunks: the loads of a sub-batch in flight together)
  auto load_and_rows = [&](const uint8_t* img, int which) {
    const uint32_t first = px_one(img), pat = first * 0x01010101u;
    uint32_t diff = 0u;
    auto both = [&](auto tag) {
#pragma unroll
      for (int k0 = 0; k0 < NCH; k0 += SB) {
        uint32_t ra[SB], rb[SB];
#pragma unroll
        for (int t = 0; t < SB; ++t) chunk_load(tag, k0 + t, img, &ra[t], &rb[t]);
#pragma unroll
        for (int t = 0; t < SB; ++t) chunk_commit(tag, k0 + t, ra[t], rb[t], pat, &diff);
      }
    };
    if (fills) both(std::true_type{});
    else both(std::false_type{});
    rows_of(diff, first, which);
  };

  // column 0 after a forward column pass = G[v] = F[v][0] + i F[v][M/2] (two real columns): apart with the partner bin M - v, in
  // place: slot v <- F0[v], slot M - v <- FH[v] (0 < v < M/2), slot 0 <- (F0[0], F0[M/2]), slot M/2 <- (FH[0], FH[M/2]) -- every
  // slot then meets its counterpart of the other image element by element. By the wave that owns column 0.
  auto split_col0 = [&]() {
    constexpr int KV = (H + 1 + 63) / 64;
    cf g1[KV], g2[KV];
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int v = lane + 64 * i;
      g1[i] = g2[i] = cf{0.f, 0.f};
      if (v < H) {
        g1[i] = lds_read(&z[spec_at(v, 0)]);
        g2[i] = lds_read(&z[spec_at(v == 0 ? H : M - v, 0)]);  // (the lane of v = 0 takes G[M/2] along)
      }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int v = lane + 64 * i;
      if (v == 0) {
        // G[0] = (F0[0], FH[0]), G[M/2] = (F0[M/2], FH[M/2]) -> (F0[0], F0[M/2]), (FH[0], FH[M/2])
        z[spec_at(0, 0)] = {g1[i].x, g2[i].x};
        z[spec_at(H, 0)] = {g1[i].y, g2[i].y};
      } else if (v < H) {
        cf f0, fh;
        untangle2(g1[i], g2[i], &f0, &fh);
        z[spec_at(v, 0)] = {0.5f * f0.x, 0.5f * f0.y};
        z[spec_at(M - v, 0)] = {0.5f * fh.x, 0.5f * fh.y};
      }
    }
    wave_sync();
  };

  // ---- previous image: rows, barrier, columns; its half spectrum moves into registers
  constexpr int KE = (LPW * M + 63) / 64;  // elements of the wave's columns per lane: element q = lane + 64 k -> (row q / LPW, column q % LPW)
  constexpr int NPV = XSINK ? NGL * NBL * RL : KE;  // XSINK: the last column stage's own outputs: (group, butterfly), output p
  constexpr int KV0 = (H + 1 + 63) / 64;      // XSINK: column 0's slots v and M - v (v <= M/2) of the previous image, lane v
  cf pv[NPV], pv0a[KV0], pv0b[KV0];
  if constexpr (MOF_HABL != 4) load_and_rows(prev, 1);
  __syncthreads();
  // gray frames: the CURRENT image's pixels are requested here, before the previous image's column pass, and wait in 2 NCH registers
  // (8 - 10) until the tile is free -- their memory latency runs under that pass instead of in front of the row pass (MOF_HALF_PREFETCH)
  constexpr bool PREFETCH = MOF_HALF_PREFETCH != 0 && CH == 1 && !SEQ;  // (SEQ: once per run only, and its registers would be live across the run's loop: 39 spills at M = 120)
  uint32_t ca[PREFETCH ? NCH : 1], cb[PREFETCH ? NCH : 1], cfirst = 0u;
  if constexpr (PREFETCH) {
    cfirst = px_one(cur);
    if (fills) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) chunk_load(std::true_type{}, k, cur, &ca[k], &cb[k]);
    } else {
#pragma unroll
      for (int k = 0; k < NCH; ++k) chunk_load(std::false_type{}, k, cur, &ca[k], &cb[k]);
    }
  }
  // forward column pass of the image in the tile (UFUSE: with the untangle as its source and a workgroup barrier inside its first stage)
  auto fwd_cols = [&](auto sink) {
    using Sink = decltype(sink);
    if constexpr (MOF_HABL == 1) return;
    if constexpr (UFUSE)
      pass_lines_static<SP, 0, 1, Sink, HalfUntangleSrc, WorkgroupSync, NG0, (XSINK ? NGL : 0)>(z, tw, cols, l0, nl, lane, false, sink,
                                                                                                HalfUntangleSrc{P, M, SKM, SH, RSK});
    else if (nl > 0)
      pass_lines_static<SP, 0, 1, Sink>(z, tw, cols, l0, nl, lane, false, sink);
  };
  if constexpr (XSINK) {
#pragma unroll
    for (int i = 0; i < NPV; ++i) pv[i] = cf{1.f, 0.f};
    if constexpr (MOF_HABL != 4) fwd_cols(HalfSaveSink{pv, RL});
    if (wave == 0) {
      split_col0();
#pragma unroll
      for (int i = 0; i < KV0; ++i) {
        const int v = lane + 64 * i;
        pv0a[i] = pv0b[i] = cf{1.f, 0.f};
        if (v <= H) {
          pv0a[i] = lds_read(&z[spec_at(v, 0)]);
          pv0b[i] = lds_read(&z[spec_at(v == 0 ? H : M - v, 0)]);
        }
      }
    }
  } else {
    fwd_cols(NoSink{});
    if (wave == 0) split_col0();
#pragma unroll
    for (int k = 0; k < KE; ++k) {
      const int q = lane + 64 * k, r = q / LPW, c = q % LPW;
      pv[k] = (r < M && c < nl) ? lds_read(&z[spec_at(r, l0 + c)]) : cf{1.f, 0.f};
    }
  }
  __syncthreads();  // every wave has its columns: the tile may take the current image

  // ---- current image: rows, barrier, then per wave forward columns -> cross-power -> inverse columns
  if constexpr (PREFETCH) {
    const uint32_t pat = cfirst * 0x01010101u;
    uint32_t diff = 0u;
    if (fills) {
#pragma unroll
      for (int k = 0; k < NCH; ++k) chunk_commit(std::true_type{}, k, ca[k], cb[k], pat, &diff);
    } else {
#pragma unroll
      for (int k = 0; k < NCH; ++k) chunk_commit(std::false_type{}, k, ca[k], cb[k], pat, &diff);
    }
    rows_of(diff, cfirst, 0);
  } else {
    load_and_rows(cur, 0);
  }
  if constexpr (SEQ) goto have_current;  // (every `goto` sits inside `if constexpr`: the pair form has no loop in its control flow)
next_pair:  // SEQ: the run's next frame
  if constexpr (SEQ) {
    // every per-lane index below is invariant across the run's pairs and the optimiser would hoist them all out of the loop (the
    // persistent form of r05: 371 spilled VGPRs); the lane index made opaque once per trip keeps them inside
    asm volatile("" : "+v"(lane), "+v"(tid));
    li0 = lane / CPR;
    xc0 = lane - li0 * CPR;
    load_and_rows(cur, 0);
  }
have_current:
  __syncthreads();
  // A CONSTANT patch that zero padding turned into an n x n box (n, m even): its spectrum is EXACTLY zero on the Nyquist row and
  // column in the reference's transforms (alternating sums of equal numbers), so C = 0 there; here the rows were transformed in
  // pairs and the zeros carry rounding noise that the normalisation would blow up to unit magnitude (pc_large_kernel.hip, L6)
  const bool box_zeros = __builtin_amdgcn_readfirstlane((int)(M > n && (flags[0] == 0 || flags[1] == 0))) != 0;  // (a scalar: the sinks branch on it per bin)
  const int zq = box_zero_period(n, M);  // its exact-zero lines: the multiples of zq (the Nyquist line alone for most sizes)
  // one bin of column 0 (slot rr): the general rule, or -- slots 0 and M/2 -- the two real-only components (C = P / (P^2 + eps),
  // SURVEY F8); box_zeros: slot 0 = (C0[0], C0[M/2]), slots M/2 .. M-1 hold the column u = M/2
  auto col0_bin = [&](cf av, cf bv, int rr, bool on) -> cf {
    cf C = cross_power_ab(av, bv, false);
    if (rr == 0 || rr == H) {
      const float c1 = cross_power_ab(cf{av.x, 0.f}, cf{bv.x, 0.f}, true).x, c2 = cross_power_ab(cf{av.y, 0.f}, cf{bv.y, 0.f}, true).x;
      C = {c1, c2};
      if (rr == 0 && on) flags[2] = __float_as_int(c1);  // C_dc: all that is left of a degenerate pair's spectrum
    }
    if (box_zeros) {  // slot rr < M/2: bin (v = rr, u = 0); slot 0 also (M/2, 0); slot M/2: (0, M/2), (M/2, M/2); slots rr > M/2: (M - rr, M/2)
      const bool zh = box_zero_line(H, zq);
      if (rr == 0) C.y = zh ? 0.f : C.y;
      else if (rr == H) C = zh ? cf{0.f, 0.f} : C;
      else if (rr < H) C = box_zero_line(rr, zq) ? cf{0.f, 0.f} : C;
      else C = (zh || box_zero_line(M - rr, zq)) ? cf{0.f, 0.f} : C;
    }
    return C;
  };
  if constexpr (XSINK) {
    if constexpr (MOF_HABL == 2) fwd_cols(NoSink{});
    else fwd_cols(HalfXpowSink<SEQ>{pv, RL, H, box_zeros, zq});
    if (wave == 0) {
      // column 0: apart, crossed with the previous image's slots (registers, lane v), together again -- G'[v] = conj C0[v] + i conj CH[v],
      // G'[M - v] = C0[v] + i CH[v] (C0, CH Hermitian in v) -- straight from the registers
      split_col0();
      cf c0v[KV0], chv[KV0];
#pragma unroll
      for (int i = 0; i < KV0; ++i) {
        const int v = lane + 64 * i;
        const bool on = v <= H;
        const int ra = on ? v : 1, rb = on ? (v == 0 ? H : M - v) : 1;
        const cf ava = lds_read(&z[spec_at(ra, 0)]), avb = lds_read(&z[spec_at(rb, 0)]);
        c0v[i] = col0_bin(ava, pv0a[i], ra, on);
        chv[i] = col0_bin(avb, pv0b[i], rb, on);
        if constexpr (SEQ) {  // this frame's column 0 is the next pair's previous one
          pv0a[i] = on ? ava : cf{1.f, 0.f};
          pv0b[i] = on ? avb : cf{1.f, 0.f};
        }
      }
      wave_sync();
#pragma unroll
      for (int i = 0; i < KV0; ++i) {
        const int v = lane + 64 * i;
        if (v == 0) {
          z[spec_at(0, 0)] = {c0v[i].x, chv[i].x};  // (C0[0], CH[0])
          z[spec_at(H, 0)] = {c0v[i].y, chv[i].y};  // (C0[M/2], CH[M/2])
        } else if (v < H) {
          z[spec_at(v, 0)] = {c0v[i].x + chv[i].y, chv[i].x - c0v[i].y};
          z[spec_at(M - v, 0)] = {c0v[i].x - chv[i].y, chv[i].x + c0v[i].y};
        }
      }
      wave_sync();
    }
  } else {
  fwd_cols(NoSink{});
  if (wave == 0) split_col0();
  if (MOF_HABL != 2) {
#pragma unroll
    for (int k = 0; k < KE; ++k) {
      const int q = lane + 64 * k, r = q / LPW, c = q % LPW;
      const bool on = r < M && c < nl;
      const int rr = on ? r : 0, u = on ? l0 + c : l0;  // (lanes past the wave's elements repeat a bin: cross_power_ab's
      const cf av = lds_read(&z[spec_at(rr, u)]);       //  wave-uniform branch wants every lane to take part)
      const cf bv = on ? pv[k] : cf{1.f, 0.f};
      cf C = u == 0 ? col0_bin(av, bv, rr, on) : cross_power_ab(av, bv, false);
      if (box_zeros && u != 0 && (box_zero_line(rr, zq) || box_zero_line(u, zq))) C = {0.f, 0.f};
      // conjugated for the inverse (a forward transform of conj C); column 0 keeps C itself until it is put together again below
      if (on) z[spec_at(rr, u)] = u == 0 ? C : cf{C.x, -C.y};
    }
  }
  wave_sync();
  if (wave == 0) {
    // column 0 back together: G'[v] = conj C0[v] + i conj CH[v], G'[M - v] = C0[v] + i CH[v] (C0, CH Hermitian in v)
    constexpr int KV = (H + 1 + 63) / 64;
    cf c0v[KV], chv[KV];
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int v = lane + 64 * i;
      c0v[i] = chv[i] = cf{0.f, 0.f};
      if (v <= H) {
        c0v[i] = lds_read(&z[spec_at(v, 0)]);
        chv[i] = lds_read(&z[spec_at(v == 0 ? H : M - v, 0)]);
      }
    }
    wave_sync();
#pragma unroll
    for (int i = 0; i < KV; ++i) {
      const int v = lane + 64 * i;
      if (v == 0) {
        z[spec_at(0, 0)] = {c0v[i].x, chv[i].x};  // (C0[0], CH[0])
        z[spec_at(H, 0)] = {c0v[i].y, chv[i].y};  // (C0[M/2], CH[M/2])
      } else if (v < H) {
        z[spec_at(v, 0)] = {c0v[i].x + chv[i].y, chv[i].x - c0v[i].y};
        z[spec_at(M - v, 0)] = {c0v[i].x - chv[i].y, chv[i].x + c0v[i].y};
      }
    }
    wave_sync();
  }
  }  // (!XSINK)
  if (nl > 0 && MOF_HABL != 1) pass_lines_static<SP>(z, tw, cols, l0, nl, lane, false);
  __syncthreads();

  // ---- Hermitian row pairs: line j carries rows y1 = 2j, y2 = 2j + 1: E[u] = D[y1][u] + i D[y2][u], D[y][M - u] = conj D[y][u];
  //      column 0 holds (D[y][0], D[y][M/2]), both real. spec -> rows layout in place per line, every read before the first write.
  if constexpr (!MOF_HALF_PAIR_SRC) {
    constexpr int KU = (LPW * H + 63) / 64;
    cf d1[KU], d2[KU];
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int q = lane + 64 * k, li = q / H, u = q % H;
      d1[k] = d2[k] = cf{0.f, 0.f};
      if (li < nl) {
        d1[k] = lds_read(&z[spec_at(2 * (l0 + li), u)]);
        d2[k] = lds_read(&z[spec_at(2 * (l0 + li) + 1, u)]);
      }
    }
    wave_sync();
#pragma unroll
    for (int k = 0; k < KU; ++k) {
      const int q = lane + 64 * k, li = q / H, u = q % H;
      if (li < nl) {
        if (u == 0) {
          z[rows_at(l0 + li, 0)] = {d1[k].x, d2[k].x};
          z[rows_at(l0 + li, H)] = {d1[k].y, d2[k].y};
        } else {
          z[rows_at(l0 + li, u)] = {d1[k].x - d2[k].y, d1[k].y + d2[k].x};
          z[rows_at(l0 + li, M - u)] = {d1[k].x + d2[k].y, d2[k].x - d1[k].y};
        }
      }
    }
    wave_sync();
  }
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  if (nl > 0 && MOF_HABL != 1) {
    if constexpr (MOF_HALF_PAIR_SRC)
      pass_lines_static<SP, 0, 1, HalfScanSink, HalfPairSrc>(z, tw, rows, l0, nl, lane, false, HalfScanSink{&best, M, H}, HalfPairSrc{P2, M, SKM, SH, RSK});
    else
      pass_lines_static<SP, 0, 1, HalfScanSink>(z, tw, rows, l0, nl, lane, false, HalfScanSink{&best, M, H});
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();

  // ---- weighted centroid in double + validity gate (:1337-1383, :1838-1856), wave 0
  if (wave == 0) {
    for (int w = 1; w < WAVES; ++w) best = better(best, red[w]);
    const bool have = best.idx != 0x7fffffff;
    const int py = have ? best.idx / M : 0, pxk = have ? best.idx - py * M : 0;
    const int ys = py - 2 + lane / 5, xs = pxk - 2 + lane % 5;
    double val = 0.0;
    if (have && lane < 25 && ys >= 0 && ys <= M - 1 && xs >= 0 && xs <= M - 1) {  // window clamped to the (padded) patch
      const int y = ys - H < 0 ? ys - H + M : ys - H, x = xs - H < 0 ? xs - H + M : xs - H;  // un-shifted position
      const cf s = z[rows_at(y >> 1, x)];
      val = (double)((y & 1) ? s.y : s.x);
    }
    double cx = (double)xs * val, cy = (double)ys * val, sum = val;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      cx += __shfl_xor(cx, off, 64);
      cy += __shfl_xor(cy, off, 64);
      sum += __shfl_xor(sum, off, 64);
    }
    if (lane == 0) {
      sum += 2.220446049250313e-16;  // DBL_EPSILON :1378
      // shift = -(center - t) = t - M / 2.0 (:1836): cv::phaseCorrelate's centre is that of the PADDED image
      const double half_m = (double)M / 2.0, half_n = (double)n / 2.0;
      double sx = cx / sum - half_m, sy = cy / sum - half_m;
      // a constant patch: its transform is exactly zero off DC, the surface is flat = C_dc (pc_common.hpp). With padding
      // (m > n) only the all-zero patch stays constant on the padded image.
      const bool cconst = flags[0] == 0, pconst = flags[1] == 0;
      const bool degenerate = M == n ? (cconst || pconst) : ((cconst && flags[3] == 0) || (pconst && flags[4] == 0));
      if (degenerate) {
        const double c9 = 9.0 * (double)__int_as_float(flags[2]);
        sx = sy = (c9 > 0.0 ? c9 / (c9 + 2.220446049250313e-16) : 0.0) - half_m;
      }
      // the gate compares with samplePointSize / 2 -- the UNPADDED size (:1841-1842)
      const bool bad = (sx * sx + sy * sy > a.max_px_speed_sq) || (fabs(sx) > half_n) || (fabs(sy) > half_n) || (sx != sx) ||
                       (sy != sy) || (!have && !degenerate);
      if (bad) sx = sy = __builtin_nan("");
      a.out[2 * p] = sx;
      a.out[2 * p + 1] = sy;
      if constexpr (SEQ) {  // the current frame becomes the previous one (this lane was the flags' last reader)
        flags[1] = flags[0];
        flags[4] = flags[3];
        flags[0] = 0;
      }
    }
  }
  if constexpr (SEQ) {
    if (--pairs_left > 0) {
      cur += a.cur_stride;
      p += (size_t)gridDim.x * gridDim.y;
      __syncthreads();  // the centroid window is read, the flags are shifted: the tile may take the next frame
      goto next_pair;
    }
  }
}

// the transform sizes with an instantiation: the even 5-smooth sizes in (135, 192] -- what this kernel was written for --, the sizes
// below that where it beats the full-tile kernels on the box (60, 96, 100 against the planned kernel: +10 .. 25 %; 120 against the tuned
// pair kernel: +18 %; profiles/r05_half_vs_planned_rates.txt, r05_half_vs_planned_bench_ab.txt; 72, 90 since the round's second half:
// +3 %, profiles/r05_half_vs_planned_final.txt -- it loses at 40, 48, 50, 54, 80, 108 and on patches padded to 64), and 128 / 64 for the A/B
// against the tuned pair kernels (MOF_FFT_HALF=1; 128 also serves the video form)
#ifdef MOF_HALF_ONLY  // (A/B sweeps: one instantiation compiles in seconds)
#define MOF_HALF_SIZES(X) X(MOF_HALF_ONLY)
#else
#define MOF_HALF_SIZES(X) X(60) X(64) X(72) X(90) X(96) X(100) X(120) X(128) X(144) X(150) X(160) X(162) X(180) X(192)
#endif
// ... and the sizes with the VIDEO form only: on pairs the full-tile planned kernel is faster there, on a video one image transform per
// pair beats it (108: +28 %, 50: +15 %, 54: +11 % -- profiles/r05_half_vs_planned_video.txt; 48 / 40 tie, 80 loses: not listed)
#ifdef MOF_HALF_ONLY
#define MOF_HALF_SEQ_SIZES(X)
#else
#define MOF_HALF_SEQ_SIZES(X) X(50) X(54) X(108)
#endif

constexpr bool half_seq_size(int m) { return m != 162; }  // (162's last column stage runs long lines: no register-resident spectrum)

template <int CH, int MS>
hipError_t configure_half_one() {
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_half_kernel<CH, MS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                                     HalfPlanOf<MS>::HP.lds_bytes);
  if constexpr (half_seq_size(MS)) {
    if (e == hipSuccess)
      e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_half_kernel<CH, MS, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                              HalfPlanOf<MS>::HP.lds_bytes);
  }
  return e;
}

}  // namespace

bool pc_half_supported(int m) {
  switch (m) {
#define X(M) case M:
    MOF_HALF_SIZES(X)
#undef X
    return true;
    default: return false;
  }
}

int pc_half_workgroups_per_cu(int m) {
  switch (m) {
#define X(M) case M: return HalfPlanOf<M>::HP.wgs_per_cu;
    MOF_HALF_SIZES(X)
    MOF_HALF_SEQ_SIZES(X)
#undef X
    default: return 0;
  }
}

template <int CH, int MS>
hipError_t configure_half_seq_only() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_half_kernel<CH, MS, true>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             HalfPlanOf<MS>::HP.lds_bytes);
}

// raises the dynamic-LDS limit of the ONE padded size `m` an engine will launch (pair form and, where it exists, video form; both
// channel counts) on the calling thread's current device -- not of all ~60 instantiations: that loaded every code object at every
// engine create, on every device of a shard group
hipError_t pc_configure_half(int m) {
  hipError_t e;
  switch (m) {
#define X(M)                                                               \
  case M:                                                                  \
    if ((e = configure_half_one<1, M>()) != hipSuccess) return e;          \
    return configure_half_one<3, M>();
    MOF_HALF_SIZES(X)
#undef X
#define X(M)                                                               \
  case M:                                                                  \
    if ((e = configure_half_seq_only<1, M>()) != hipSuccess) return e;     \
    return configure_half_seq_only<3, M>();
    MOF_HALF_SEQ_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
}

// a.downscale must be 1 and a.peak_model 0 (the caller keeps the other front ends on their pipelines); m = the padded size, n the patch
hipError_t launch_pc_half(const PcArgs& a_in, int m, int n, int n_pairs, hipStream_t stream) {
  if (a_in.downscale != 1 || a_in.peak_model != 0 || (a_in.channels != 1 && a_in.channels != 3) || n > m || n < 2) return hipErrorInvalidValue;
  const int patches = a_in.grid_x * a_in.grid_y;
  for (int k0 = 0; k0 < n_pairs; k0 += 65535) {  // the pair index rides gridDim.z
    const int nk = n_pairs - k0 < 65535 ? n_pairs - k0 : 65535;
    PcArgs c = a_in;
    c.cur = a_in.cur + (size_t)k0 * a_in.cur_stride;
    c.prev = a_in.prev + (size_t)k0 * a_in.prev_stride;
    c.out = a_in.out + (size_t)k0 * patches * 2;
    c.total = nk * patches;
    const dim3 g((unsigned)c.grid_x, (unsigned)c.grid_y, (unsigned)nk);
    switch (m) {
#define X(M)                                                                                                                             \
  case M:                                                                                                                                \
    if (c.channels == 3) hipLaunchKernelGGL((pc_half_kernel<3, M>), g, dim3((unsigned)HalfPlanOf<M>::T), (size_t)HalfPlanOf<M>::HP.lds_bytes, stream, c, n, nk, 1); \
    else hipLaunchKernelGGL((pc_half_kernel<1, M>), g, dim3((unsigned)HalfPlanOf<M>::T), (size_t)HalfPlanOf<M>::HP.lds_bytes, stream, c, n, nk, 1);          \
    break;
      MOF_HALF_SIZES(X)
#undef X
      default: return hipErrorInvalidValue;
    }
  }
  return hipGetLastError();
}

bool pc_half_sequence_supported(int m) {
  switch (m) {
#define X(M) case M:
    MOF_HALF_SEQ_SIZES(X)
#undef X
    return true;
    default: return pc_half_supported(m) && half_seq_size(m);
  }
}

// the video form: a.cur = frame 0 of the launch's first pair, frame f at a.cur + f * a.cur_stride; n_pairs pairs in runs of `run`
// run = 0: chosen here. Workgroups of one launch all take the same time (run + ~0.45 image transforms: a run's first frame is
// transformed without an inverse), so the launch lasts ceil(workgroups / resident slots) rounds of that: the run length in 4 .. 64 with
// the least rounds x (run + 0.45) wins -- 512 pairs of 9 patches at one workgroup per CU: runs of 19 = 243 workgroups in ONE round,
// where runs of 16 make 288 = a full round and a round that uses 32 of 256 CUs (measured: l160seq 605 k -> see DESIGN)
hipError_t launch_pc_half_sequence(const PcArgs& a, int m, int n, int n_pairs, int run, hipStream_t stream) {
  if (a.downscale != 1 || a.peak_model != 0 || (a.channels != 1 && a.channels != 3) || n > m || n < 2 || run < 0 || n_pairs < 1) return hipErrorInvalidValue;
  if (run == 0) {
    int dev = 0, cus = 0;  // per call: the devices of a shard group need not be alike (an attribute query, no synchronisation)
    if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus <= 0) cus = 256;
    const long slots = (long)cus * (pc_half_workgroups_per_cu(m) > 0 ? pc_half_workgroups_per_cu(m) : 1), patches = (long)a.grid_x * a.grid_y;
    double best = 0.0;
    for (int r = 4; r <= 64; ++r) {
      const long wgs = patches * ((n_pairs + r - 1) / r), rounds = (wgs + slots - 1) / slots;
      const double t = (double)rounds * ((double)r + 0.45);
      if (run == 0 || t <= best) best = t, run = r;
    }
  }
  const int n_runs = (n_pairs + run - 1) / run;
  if (n_runs > 65535) return hipErrorInvalidValue;
  PcArgs c = a;
  c.total = n_pairs * a.grid_x * a.grid_y;
  const dim3 g((unsigned)c.grid_x, (unsigned)c.grid_y, (unsigned)n_runs);
  switch (m) {
#define X(M)                                                                                                                                  \
  case M:                                                                                                                                     \
    if constexpr (half_seq_size(M)) {                                                                                                         \
      if (c.channels == 3) hipLaunchKernelGGL((pc_half_kernel<3, M, true>), g, dim3((unsigned)HalfPlanOf<M>::T), (size_t)HalfPlanOf<M>::HP.lds_bytes, stream, c, n, n_pairs, run); \
      else hipLaunchKernelGGL((pc_half_kernel<1, M, true>), g, dim3((unsigned)HalfPlanOf<M>::T), (size_t)HalfPlanOf<M>::HP.lds_bytes, stream, c, n, n_pairs, run);                 \
    } else {                                                                                                                                  \
      return hipErrorInvalidValue;                                                                                                            \
    }                                                                                                                                         \
    break;
    MOF_HALF_SIZES(X)
    MOF_HALF_SEQ_SIZES(X)
#undef X
    default: return hipErrorInvalidValue;
  }
  return hipGetLastError();
}

}  // namespace mof
