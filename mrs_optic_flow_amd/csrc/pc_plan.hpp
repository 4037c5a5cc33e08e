// pc_plan.hpp -- device side of the run-time planned 1-D transforms: a chain of Stockham auto-sort stages with radices from
// {2, 3, 4, 5, 8} (the radix set of the reference's OpenCL plans, ocl_getRadixes, /root/reference/src/FftMethod.cpp:481-539),
// executed by ONE wave on lines it owns, in place, in LDS. Shared by pc_kernel_generic.hip (patch tile in LDS) and
// pc_large_kernel.hip (lines streamed through HBM: whole frames and patches too large for a CU). The host side of the plan
// (pc_build_plan, pc_radix_chain) lives in pc_kernel_generic.hip; PcPlan in mof_kernels.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mof_kernels.h"
#include "pc_common.hpp"

namespace mof {
namespace {

// q / d and q % d for 0 <= q < 2^22, 1 <= d < 2^12, inv = 1.0f / d: the float quotient is within one of the true one,
// one correction step makes it exact
__device__ __forceinline__ int fdiv(int q, int d, float inv, int* rem) {
  if (__builtin_constant_p(d)) {  // compile-time instantiations of a plan: the compiler's multiply-shift
    *rem = q % d;
    return q / d;
  }
  int li = (int)(((float)q + 0.5f) * inv);
  int r = q - li * d;
  if (r < 0) { --li; r += d; }
  else if (r >= d) { ++li; r -= d; }
  *rem = r;
  return li;
}

// Where a pass finds element e of line l in the tile: row passes walk along a tile row, column passes down a tile column;
// the column coordinate c of the tile is stored at c + (c >> 3) when the plan skews (stride-8 stage writes then spread
// over the banks).
struct Walk {
  int ls, es;        // strides (complex elements) of the line index / the element index before the skew
  int lmask, emask;  // ~0 where the skew applies to that coordinate
  int line_fast;     // lane map of a stage: 0 = the butterfly index in the fast lane bits (row walks), 1 = the line (column walks)
  int line_perm = 0; // 1: later stages of a row walk take the lines of a four-line group in the order 0 2 1 3 (stage_rt; the half-tile kernel's pitches)
  int sh = 3;        // the skew's shift: coordinate c sits at c + (c >> sh) (3 everywhere but the half-tile kernel's radix-16-first sizes: 4)
  // r06 (A/B, the half-tile kernel): a ROW skew -- tile row r (a line of the row walks = rows 2j | 2j + 1; an element of the column walks)
  // starts rs * (r >> 4) complex elements later, so that the first-stage column writes of a radix-16-first size, 16 rows apart = 0 (mod 32
  // banks) for any pitch, land in different banks. lrs / ers = rs for the walk whose line / element is the row; 0 everywhere else.
  int lrs = 0, ers = 0;
  __device__ __forceinline__ int loffs(int l) const { return l * ls + ((l >> sh) & lmask) + lrs * (l >> 3); }  // (line j = rows 2j, 2j + 1: (2j) >> 4 = j >> 3)
  __device__ __forceinline__ int eoff(int e) const { return e * es + ((e >> sh) & emask) + ers * (e >> 4); }
  __device__ __forceinline__ int at(int l, int e) const { return loffs(l) + eoff(e); }
};

// Composite radices for the compile-time plans (two stages per 1-D transform for every 5-smooth size up to 135, as the tuned
// kernels have): Cooley-Tukey in registers, n = R2 n1 + n2, k = k1 + R1 k2 -- R1 butterflies over n1, twiddles W_R^{n2 k1}, R2
// butterflies over n2. R1 is the odd factor, R2 the even one: output R/2 = (k1 = 0, k2 = R2/2) then passes no twiddle and is a
// sum of differences, exact on integers (the real-only CCS slots, pc_plan_build.hpp).
template <int R1, int R2>
__device__ __forceinline__ void butterfly_ct(cf* v, const cf* w) {
  cf t[R2][R1];
#pragma unroll
  for (int n2 = 0; n2 < R2; ++n2) {
    cf a[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) a[n1] = v[n2 + R2 * n1];
    if constexpr (R1 == 3) butterfly3(a);
    else if constexpr (R1 == 5) butterfly5(a);
    else butterfly<R1>(a);
#pragma unroll
    for (int k1 = 0; k1 < R1; ++k1) t[n2][k1] = (n2 * k1 == 0) ? a[k1] : cmul(a[k1], w[n2 * k1]);
  }
#pragma unroll
  for (int k1 = 0; k1 < R1; ++k1) {
    cf b[R2];
#pragma unroll
    for (int n2 = 0; n2 < R2; ++n2) b[n2] = t[n2][k1];
    if constexpr (R2 == 3) butterfly3(b);
    else butterfly<R2>(b);
#pragma unroll
    for (int k2 = 0; k2 < R2; ++k2) v[k1 + R1 * k2] = b[k2];
  }
}
__device__ __forceinline__ void butterfly6(cf* v) {  // 3 x 2
  const cf w[3] = {{1.f, 0.f}, {0.5f, -0.86602540378443864676f}, {-0.5f, -0.86602540378443864676f}};
  butterfly_ct<3, 2>(v, w);
}
__device__ __forceinline__ void butterfly9(cf* v) {  // 3 x 3
  const cf w[5] = {{1.f, 0.f},
                   {0.76604444311897803520f, -0.64278760968653932632f},
                   {0.17364817766693034885f, -0.98480775301220805937f},
                   {-0.5f, -0.86602540378443864676f},
                   {-0.93969262078590838405f, -0.34202014332566873304f}};
  butterfly_ct<3, 3>(v, w);
}
__device__ __forceinline__ void butterfly10(cf* v) {  // 5 x 2
  const cf w[5] = {{1.f, 0.f},
                   {0.80901699437494742410f, -0.58778525229247312917f},
                   {0.30901699437494742410f, -0.95105651629515357212f},
                   {-0.30901699437494742410f, -0.95105651629515357212f},
                   {-0.80901699437494742410f, -0.58778525229247312917f}};
  butterfly_ct<5, 2>(v, w);
}
__device__ __forceinline__ void butterfly12(cf* v) {  // 3 x 4
  const cf w[7] = {{1.f, 0.f},  {0.86602540378443864676f, -0.5f}, {0.5f, -0.86602540378443864676f}, {0.f, -1.f},
                   {-0.5f, -0.86602540378443864676f}, {-0.86602540378443864676f, -0.5f}, {-1.f, 0.f}};
  butterfly_ct<3, 4>(v, w);
}

// the butterfly of a run-time radix R <= SLOTS (wave-uniform: a scalar branch)
template <int SLOTS>
__device__ __forceinline__ void bfly_rt(int R, cf* v) {
  if constexpr (SLOTS == 16) {  // compile-time plans only: R is a constant there and one branch survives
    if (R == 16) butterfly<16>(v);
    else if (R == 15) butterfly15(v);
    else if (R == 12) butterfly12(v);
    else if (R == 10) butterfly10(v);
    else butterfly9(v);
  } else if constexpr (SLOTS == 8) {
    if (R == 8) butterfly<8>(v);
    else if (R == 6) butterfly6(v);
    else butterfly5(v);
  } else if constexpr (SLOTS == 5) {
    butterfly5(v);
  } else if constexpr (SLOTS == 4) {
    if (R == 4) butterfly<4>(v);
    else if (R == 3) butterfly3(v);
    else butterfly<2>(v);
  } else if constexpr (SLOTS == 3) {
    butterfly3(v);
  } else {
    butterfly<2>(v);
  }
}

// One Stockham stage (radix R <= SLOTS, `np` = product of the earlier radices, bpl = m / R butterflies per line, tstep =
// m / (np R)) on lines [line0, line0 + nlines) owned by ONE wave, in place. Butterfly x of a line reads elements x + j bpl,
// multiplies them by W_{np R}^{j (x mod np)} and writes the R outputs to (x - x mod np) R + x mod np + p np.
// herm_first: the first stage of the Hermitian inverse column pass -- "line" c is the column PAIR (c, c + m/2), element v is
// E[v] = F1[v][c] + i F1[v][c + m/2] built from rows 0 .. m/2 - 1 of the tile (row 0 = F1[0] + i F1[m/2], both real;
// F1[m - v] = conj F1[v]) -- see col_pass_inv in pc_passes.hpp.
// Lane map. bpl <= 64 (every patch that fits a CU): a lane keeps ONE butterfly index x for the whole call and walks lines --
// lpg = 64 / bpl lines side by side, NB = 16 / SLOTS of those rows per group -- so its R - 1 twiddles and its R + R element
// offsets are formed once per call and live in registers. Row walks put x in the fast lane bits (8 consecutive elements per
// line and 32-lane read group), column walks the line (8 neighbouring columns per row and group): both conflict-free on reads
// with the plan's pitch = 8 (mod 16). bpl > 64 (long lines of the large-patch pipeline): one line at a time, x = lane + 64 t.
// History (r04, p62 = 65,536 patches of 62 x 62 padded to 64): the first form -- lines packed densely into the lanes, one
// division, R address computations and R - 1 twiddle reads from LDS per butterfly, stages as out-of-line functions per radix
// (270 scratch accesses per wave for their call frames) -- issued 5383 VALU + 1561 SALU instructions per wave and patch
// against the tuned kernel's 1412, with 61 % of its LDS cycles bank conflicts (profiles/r04_p62_planned_v1_sq_pmc.csv).
// Sink: what the LAST stage of a pass hands its outputs to besides the tile (line l, element o of the line, value): the planned
// kernel's arg-max rides the final stage of the inverse transform that way instead of a sweep of its own over the tile.
#ifndef MOF_PLANNED_TW8  // 1: radix-8 stages with twiddles use butterfly8_tw (the products ride the first radix-2 layer as FMAs: 8 instructions
                         // fewer per butterfly, the same value up to rounding -- what the tuned kernels do); 0: cmul + butterfly<8> (r04)
#define MOF_PLANNED_TW8 1
#endif
// A sink may declare `transforms = true`: the last stage then hands it every output BEFORE the tile write, with the stage routine's
// compile-time loop indices (butterfly b of the lane's group, output p): val = sink.transform(line, o, val, b, p, &write). The sink may
// keep the value in a per-lane register array laid out in the stage's own lane map, replace it, or suppress the write
// (pc_half_kernel.hip: the previous image's spectrum never passes through LDS, the cross-power spectrum is formed on the way out).
template <class S, class = void>
struct SinkTransforms { static constexpr bool value = false; };
template <class S>
struct SinkTransforms<S, decltype((void)S::transforms)> { static constexpr bool value = S::transforms; };
struct NoSink {
  static constexpr bool active = false;
  __device__ __forceinline__ void operator()(int, int, cf) const {}
};

// Src: where the FIRST stage of a pass takes its inputs from instead of the tile element (line l, element e): the planned kernel's
// inverse row pass forms the conjugated cross-power spectrum on the way in that way instead of in a sweep of its own.
struct NoSrc {
  static constexpr bool active = false;
  __device__ __forceinline__ cf operator()(const cf*, int, int) const { return cf{0.f, 0.f}; }
};

// Sync: what separates a group's reads from its writes. WaveSync (default): the wave's in-order LDS queue -- a stage is in place on lines
// the wave owns. WorkgroupSync: a workgroup barrier -- for a stage whose SOURCE reads lines of other waves that its own writes would
// overwrite (pc_half_kernel.hip: the untangle as the source of the forward column pass); every wave must then run the same number of
// groups (the caller guarantees exactly one).
struct WaveSync {
  __device__ __forceinline__ void operator()() const { wave_sync(); }
};
struct WorkgroupSync {
  __device__ __forceinline__ void operator()() const { __syncthreads(); }
};

// MOF_SITE_ABL (diagnostic builds, results wrong by design; tools/half_site_counters.sh): ONE class of a stage's LDS accesses is removed,
// so that SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE of the build, subtracted from the product's, give that class's share. Site =
// 1 + (column walk ? 4 : 0) + (later stage, np > 1 ? 2 : 0) + (write ? 1 : 0); 0 = the product. A removed read hands the butterfly a
// lane-dependent constant; a removed write keeps its value alive in a register (nothing upstream is dead code).
#ifndef MOF_SITE_ABL
#define MOF_SITE_ABL 0
#endif
#ifndef MOF_PLAN_LINE_PERM  // 0: every stage takes a group's lines in order (A/B)
#define MOF_PLAN_LINE_PERM 1
#endif
__device__ __forceinline__ constexpr bool site_off(int line_fast, int np, int rw) {
  return MOF_SITE_ABL != 0 && (MOF_SITE_ABL - 1) == (line_fast ? 4 : 0) + (np > 1 ? 2 : 0) + rw;
}
__device__ __forceinline__ void site_keep(cf v) { asm volatile("" ::"v"(v.x), "v"(v.y)); }

template <int SLOTS, class Sink = NoSink, class Src = NoSrc, class Sync = WaveSync>
__device__ __forceinline__ void stage_rt(cf* __restrict__ z, const cf* __restrict__ tw, const Walk& w, int m, int R, int np, int bpl,
                                         int tstep, int line0, int nlines, int lane, bool herm_first, Sink sink = Sink{}, Src src = Src{}) {
  constexpr int NB = 16 / SLOTS;  // butterflies per lane and group
  const int H = m >> 1;
  const float inv_np = 1.0f / (float)np;
  if (bpl <= 64) {
    const float inv_bpl = 1.0f / (float)bpl;
    int rem64;
    const int lpg = fdiv(64, bpl, inv_bpl, &rem64);  // lines side by side in the wave
    int x, sub;
    if (w.line_fast) {
      const float inv_lpg = 1.0f / (float)lpg;
      x = fdiv(lane, lpg, inv_lpg, &sub);
    } else {
      sub = fdiv(lane, bpl, inv_bpl, &x);
    }
    const bool lane_on = w.line_fast ? x < bpl : sub < lpg;
    // Line order inside a group (r06; measured per access class with tools/half_site_counters.sh: the later-stage reads of the row walks were
    // 2-way conflicts, 50 % of their LDS cycles, at every K1h size with four lines side by side): a later stage of a row walk reads 16
    // consecutive complex values per line, two lines per 32-lane half; with the pitches in use neighbouring lines start 48 or 16 banks
    // apart (mod 64) and overlap in 16 banks, lines TWO apart start 32 banks apart and do not. Which line a lane group takes is free
    // (a stage is in place per line): sub-groups 0 1 2 3 take lines 0 2 1 3.
    if (MOF_PLAN_LINE_PERM && !w.line_fast && np > 1 && lpg == 4 && w.line_perm) sub = ((sub & 1) << 1) | (sub >> 1);
    int k = 0;
    if (np > 1) (void)fdiv(x, np, inv_np, &k);
    cf t[SLOTS - 1];
    if (np > 1) {
#pragma unroll
      for (int j = 1; j < SLOTS; ++j)
        if (j < R) t[j - 1] = lds_read(&tw[j * k * tstep]);
    }
    // (element offsets are formed where they are used: holding R + R of them across the group loop cost 32 VGPRs in the
    // 16-slot body and spilled)
    const int obase = (x - k) * R + k;
    const int group = NB * lpg;
#pragma unroll 1
    for (int g0 = 0; g0 < nlines; g0 += group) {
      cf v[NB][SLOTS];
      int loff[NB];
      bool on[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int li = g0 + b * lpg + sub;
        const int l = line0 + li;
        on[b] = lane_on && li < nlines;
        loff[b] = w.loffs(l);
        if (on[b]) {
          const int l2off = w.loffs(l + H);  // (herm_first only)
#pragma unroll
          for (int j = 0; j < SLOTS; ++j)
            if (j < R) {
              const int e = x + j * bpl;
              cf a;
              if (site_off(w.line_fast, np, 0)) {
                a = cf{(float)(lane + j), 1.f};
              } else if constexpr (Src::active) {
                a = src(z, l, e);
              } else if (herm_first) {
                const int r = e < H ? e : (e == H ? 0 : m - e);
                const int ro = w.eoff(r);
                const cf pp = lds_read(&z[loff[b] + ro]), c = lds_read(&z[l2off + ro]);  // tile (row r, col l) and (row r, col l + H)
                if (e == 0) a = {pp.x, c.x};
                else if (e == H) a = {pp.y, c.y};
                else if (e < H) a = {pp.x - c.y, pp.y + c.x};
                else a = {pp.x + c.y, c.x - pp.y};
              } else {
                a = lds_read(&z[loff[b] + w.eoff(e)]);
              }
              // (a radix-8 stage behind an earlier one takes its twiddles inside the butterfly: butterfly8_tw, pc_common.hpp)
              if (j > 0 && np > 1 && !(MOF_PLANNED_TW8 && SLOTS == 8 && R == 8)) a = cmul(a, t[j - 1]);
              v[b][j] = a;
            }
          if (MOF_PLANNED_TW8 && SLOTS == 8 && R == 8 && np > 1) butterfly8_tw(v[b], t);
          else bfly_rt<SLOTS>(R, v[b]);
        }
      }
      Sync{}();
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (on[b]) {
#pragma unroll
          for (int p = 0; p < SLOTS; ++p)
            if (p < R) {
              const int o = obase + p * np;
              if constexpr (SinkTransforms<Sink>::value) {
                bool wr = true;
                const cf val = sink.transform(line0 + g0 + b * lpg + sub, o, v[b][p], b, p, &wr);
                if (site_off(w.line_fast, np, 1)) site_keep(val);
                else if (wr) z[loff[b] + w.eoff(o)] = val;
              } else {
                if (site_off(w.line_fast, np, 1)) site_keep(v[b][p]);
                else z[loff[b] + w.eoff(o)] = v[b][p];
                if constexpr (Sink::active) sink(line0 + g0 + b * lpg + sub, o, v[b][p]);
              }
            }
        }
      wave_sync();
    }
    return;
  }
  // long lines: one line at a time, up to NB * 64 butterflies (the plan guarantees bpl <= 64 * floor(16 / R))
  for (int li = 0; li < nlines; ++li) {
    const int l = line0 + li;
    const int loff = w.loffs(l);
    for (int x0 = 0; x0 < bpl; x0 += 64 * NB) {
      cf v[NB][SLOTS];
      int xx[NB], kk[NB];
#pragma unroll
      for (int b = 0; b < NB; ++b) {
        const int x = x0 + lane + 64 * b;
        xx[b] = x;
        kk[b] = 0;
        if (x < bpl) {
          int k = 0;
          if (np > 1) (void)fdiv(x, np, inv_np, &k);
          kk[b] = k;
#pragma unroll
          for (int j = 0; j < SLOTS; ++j)
            if (j < R) {
              const int e = x + j * bpl;
              cf a;
              if constexpr (Src::active) {
                a = src(z, l, e);
              } else if (herm_first) {
                const int r = e < H ? e : (e == H ? 0 : m - e);
                const cf pp = lds_read(&z[w.at(l, r)]), c = lds_read(&z[w.at(l + H, r)]);
                if (e == 0) a = {pp.x, c.x};
                else if (e == H) a = {pp.y, c.y};
                else if (e < H) a = {pp.x - c.y, pp.y + c.x};
                else a = {pp.x + c.y, c.x - pp.y};
              } else {
                a = lds_read(&z[loff + w.eoff(e)]);
              }
              if (j > 0 && np > 1) a = cmul(a, lds_read(&tw[j * k * tstep]));
              v[b][j] = a;
            }
          bfly_rt<SLOTS>(R, v[b]);
        }
      }
      // (in place is safe across the x0 chunks too: butterfly x reads elements x + j bpl and writes (x - k) R + k + p np, and the
      // chunks of one line are separated by the wave's in-order LDS queue only when every chunk's reads precede ITS writes AND no
      // later chunk reads what an earlier one wrote -- which does not hold in general, hence all reads of a LINE happen first:
      // the plan keeps bpl <= 64 NB, so there is exactly one chunk)
      wave_sync();
#pragma unroll
      for (int b = 0; b < NB; ++b)
        if (xx[b] < bpl) {
          const int base = (xx[b] - kk[b]) * R + kk[b];
#pragma unroll
          for (int p = 0; p < SLOTS; ++p)
            if (p < R) {
              const int o = base + p * np;
              if constexpr (SinkTransforms<Sink>::value) {
                bool wr = true;
                const cf val = sink.transform(l, o, v[b][p], b, p, &wr);
                if (wr) z[loff + w.eoff(o)] = val;
              } else {
                z[loff + w.eoff(o)] = v[b][p];
                if constexpr (Sink::active) sink(l, o, v[b][p]);
              }
            }
        }
      wave_sync();
    }
  }
}

// The same stage with the wave's lines as NG compile-time GROUPS, every group's inputs read (and transformed) before the first output
// is written: what a stage needs whose source reads lines that ANOTHER wave's writes of the same stage overwrite (Sync = WorkgroupSync
// between the two phases; each wave runs the stage exactly once, so the barrier is met by all) or whose sink / source addresses a
// per-lane register array by (group, butterfly, element) -- the indices must be compile-time for the array to stay in registers.
// NG * (16 / SLOTS) * SLOTS = 16 NG complex values per lane are in flight. Requires bpl <= 64 and nlines <= NG * group.
template <int SLOTS, int NG, class Sink = NoSink, class Src = NoSrc, class Sync = WaveSync>
__device__ __forceinline__ void stage_rt_ng(cf* __restrict__ z, const cf* __restrict__ tw, const Walk& w, int m, int R, int np, int bpl,
                                            int tstep, int line0, int nlines, int lane, Sink sink = Sink{}, Src src = Src{}) {
  constexpr int NB = 16 / SLOTS;
  const float inv_np = 1.0f / (float)np, inv_bpl = 1.0f / (float)bpl;
  int rem64;
  const int lpg = fdiv(64, bpl, inv_bpl, &rem64);
  int x, sub;
  if (w.line_fast) {
    const float inv_lpg = 1.0f / (float)lpg;
    x = fdiv(lane, lpg, inv_lpg, &sub);
  } else {
    sub = fdiv(lane, bpl, inv_bpl, &x);
  }
  const bool lane_on = w.line_fast ? x < bpl : sub < lpg;
  int k = 0;
  if (np > 1) (void)fdiv(x, np, inv_np, &k);
  cf t[SLOTS - 1];
  if (np > 1) {
#pragma unroll
    for (int j = 1; j < SLOTS; ++j)
      if (j < R) t[j - 1] = lds_read(&tw[j * k * tstep]);
  }
  const int obase = (x - k) * R + k;
  const int group = NB * lpg;
  cf v[NG][NB][SLOTS];
  int loff[NG][NB];
  bool on[NG][NB];
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int li = g * group + b * lpg + sub;
      const int l = line0 + li;
      on[g][b] = lane_on && li < nlines;
      loff[g][b] = w.loffs(l);
      if (on[g][b]) {
#pragma unroll
        for (int j = 0; j < SLOTS; ++j)
          if (j < R) {
            const int e = x + j * bpl;
            cf a;
            if (site_off(w.line_fast, np, 0)) {
              a = cf{(float)(lane + j), 1.f};
            } else if constexpr (Src::active) {
              a = src(z, l, e);
            } else {
              a = lds_read(&z[loff[g][b] + w.eoff(e)]);
            }
            if (j > 0 && np > 1 && !(MOF_PLANNED_TW8 && SLOTS == 8 && R == 8)) a = cmul(a, t[j - 1]);
            v[g][b][j] = a;
          }
        if (MOF_PLANNED_TW8 && SLOTS == 8 && R == 8 && np > 1) butterfly8_tw(v[g][b], t);
        else bfly_rt<SLOTS>(R, v[g][b]);
      }
    }
  Sync{}();
#pragma unroll
  for (int g = 0; g < NG; ++g)
#pragma unroll
    for (int b = 0; b < NB; ++b)
      if (on[g][b]) {
#pragma unroll
        for (int p = 0; p < SLOTS; ++p)
          if (p < R) {
            const int o = obase + p * np;
            const int l = line0 + g * group + b * lpg + sub;
            if constexpr (SinkTransforms<Sink>::value) {
              bool wr = true;
              const cf val = sink.transform(l, o, v[g][b][p], g * NB + b, p, &wr);
              if (site_off(w.line_fast, np, 1)) site_keep(val);
              else if (wr) z[loff[g][b] + w.eoff(o)] = val;
            } else {
              if (site_off(w.line_fast, np, 1)) site_keep(v[g][b][p]);
              else z[loff[g][b] + w.eoff(o)] = v[g][b][p];
              if constexpr (Sink::active) sink(l, o, v[g][b][p]);
            }
          }
      }
  wave_sync();
}

// all stages of one 1-D pass over the wave's lines. EXACT = false: stage bodies with 8 and 4 register slots per butterfly (radix 5
// and 6 ride in the 8-slot body, 3 and 2 in the 4-slot one; WIDE adds the 16-slot body for the composite radices 9 .. 16 of the
// compile-time plans) -- enough for every line of a tile that fits a CU (m / R <= 67) and the smallest code; EXACT = true: one body
// per radix with 16 / R butterflies per lane, which long lines need (m / R up to 64 * 16 / R).
template <bool EXACT, bool WIDE>
__device__ __forceinline__ void pass_stage(cf* z, const cf* tw, const PcPlan& pl, const Walk& w, int line0, int nlines, int lane, bool herm,
                                           int s, int& np, int& rest) {
  const int R = (int)((pl.radix_packed >> (5 * s)) & 31u);  // (a dynamic index into the kernel argument would go through scratch)
  switch (R) {  // rest = m / (np R): divisions by constants
    case 16: rest >>= 4; break;
    case 15: rest /= 15; break;
    case 12: rest /= 12; break;
    case 10: rest /= 10; break;
    case 9: rest /= 9; break;
    case 8: rest >>= 3; break;
    case 6: rest /= 6; break;
    case 5: rest /= 5; break;
    case 4: rest >>= 2; break;
    case 3: rest /= 3; break;
    default: rest >>= 1; break;
  }
  const bool h = herm && s == 0;
  if constexpr (EXACT) {
    switch (R) {
      case 8: stage_rt<8>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h); break;
      case 5: stage_rt<5>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h); break;
      case 4: stage_rt<4>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h); break;
      case 3: stage_rt<3>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h); break;
      default: stage_rt<2>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h); break;
    }
  } else {
    if (WIDE && R > 8) stage_rt<16>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h);
    else if (R > 4) stage_rt<8>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h);
    else stage_rt<4>(z, tw, w, pl.m, R, np, rest * np, rest, line0, nlines, lane, h);
  }
  np *= R;
}

template <bool EXACT = false, bool WIDE = false>
__device__ __forceinline__ void pass_lines(cf* z, const cf* tw, const PcPlan& pl, const Walk& w, int line0, int nlines, int lane,
                                           bool herm) {
  int np = 1, rest = pl.m;
  if constexpr (WIDE) {  // compile-time plan: the stage loop MUST unroll for the radices to be constants
#pragma unroll
    for (int s = 0; s < pl.n_stages; ++s) pass_stage<EXACT, WIDE>(z, tw, pl, w, line0, nlines, lane, herm, s, np, rest);
  } else {
#pragma unroll 1
    for (int s = 0; s < pl.n_stages; ++s) pass_stage<EXACT, WIDE>(z, tw, pl, w, line0, nlines, lane, herm, s, np, rest);
  }
}

// The same for a compile-time plan (SP::P a constexpr PcPlan): the stages are a compile-time recursion, so every radix, stride and
// count reaches the stage routine as a constant (a `#pragma unroll` on the run-time loop is not honoured once the 16-slot bodies make
// it large, and the radix dispatch then stays in the code).
// NG0 / NGL > 0: the first / last stage runs as stage_rt_ng with that many compile-time groups (the caller guarantees
// nlines <= NG * that stage's group and bpl <= 64 there).
template <class SP, int S = 0, int NP = 1, class Sink = NoSink, class Src = NoSrc, class Sync0 = WaveSync, int NG0 = 0, int NGL = 0>
__device__ __forceinline__ void pass_lines_static(cf* z, const cf* tw, const Walk& w, int line0, int nlines, int lane, bool herm,
                                                  Sink sink = Sink{}, Src src = Src{}) {
  if constexpr (S < SP::P.n_stages) {
    constexpr int R = SP::P.radix[S], M = SP::P.m, REST = M / (NP * R);
    constexpr int SL = R > 8 ? 16 : (R > 4 ? 8 : 4);
    static_assert(SP::P.n_stages >= 2 || !(Sink::active && Src::active), "sink and source ride different stages");
    if constexpr (S + 1 == SP::P.n_stages) {
      if constexpr (NGL > 0) stage_rt_ng<SL, NGL, Sink>(z, tw, w, M, R, NP, REST * NP, REST, line0, nlines, lane, sink);
      else stage_rt<SL, Sink>(z, tw, w, M, R, NP, REST * NP, REST, line0, nlines, lane, herm && S == 0, sink);
    } else if constexpr (S == 0) {
      if constexpr (NG0 > 0) stage_rt_ng<SL, NG0, NoSink, Src, Sync0>(z, tw, w, M, R, NP, REST * NP, REST, line0, nlines, lane, NoSink{}, src);
      else stage_rt<SL, NoSink, Src, Sync0>(z, tw, w, M, R, NP, REST * NP, REST, line0, nlines, lane, herm, NoSink{}, src);
    } else {
      stage_rt<SL>(z, tw, w, M, R, NP, REST * NP, REST, line0, nlines, lane, false);
    }
    pass_lines_static<SP, S + 1, NP * R, Sink, Src, Sync0, NG0, NGL>(z, tw, w, line0, nlines, lane, herm, sink, src);
  }
}

}  // namespace
}  // namespace mof
