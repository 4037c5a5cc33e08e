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
  __device__ __forceinline__ int at(int l, int e) const { return l * ls + ((l >> 3) & lmask) + e * es + ((e >> 3) & emask); }
};

template <int R>
__device__ __forceinline__ void bfly_r(cf* v) {
  if constexpr (R == 3) butterfly3(v);
  else if constexpr (R == 5) butterfly5(v);
  else butterfly<R>(v);
}

// One Stockham stage (radix R, `np` = product of the earlier radices) on lines [line0, line0 + nlines) owned by ONE wave,
// in place. Butterfly x of a line reads elements x + j (m / R), multiplies them by W_{np R}^{j (x mod np)} and writes the
// R outputs to (x - x mod np) R + x mod np + p np. HERM: the first stage of the Hermitian inverse column pass -- "line" c is
// the column PAIR (c, c + m/2), element v is E[v] = F1[v][c] + i F1[v][c + m/2] built from rows 0 .. m/2 - 1 of the tile
// (row 0 = F1[0] + i F1[m/2], both real; F1[m - v] = conj F1[v]) -- see col_pass_inv in pc_passes.hpp.
template <int R, bool HERM>
__device__ __attribute__((noinline)) void stage(cf* __restrict__ z, const cf* __restrict__ tw, const Walk w, int m, int np, float inv_np,
                                                int tstep, int line0, int nlines, int lane) {
  constexpr int NB = 16 / R;  // butterflies per lane and group
  const int bpl = m / R;      // butterflies per line
  const float inv_bpl = 1.0f / (float)bpl;  // (one v_rcp per stage; fdiv corrects the last bit)
  int G = (64 * NB) / bpl;    // lines per group
  if (G < 1) G = 1;           // (the plan guarantees bpl <= 64 NB)
  const int H = m >> 1;
  for (int g0 = 0; g0 < nlines; g0 += G) {
    const int gl = nlines - g0 < G ? nlines - g0 : G;
    const int nbf = gl * bpl;
    cf v[NB][R];
    int li[NB], xx[NB], kk[NB];
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int q = lane + 64 * b;
      li[b] = xx[b] = kk[b] = 0;
      if (q < nbf) {
        int x;
        const int l = line0 + g0 + fdiv(q, bpl, inv_bpl, &x);
        int k = 0;
        if (np > 1) (void)fdiv(x, np, inv_np, &k);
        li[b] = l;
        xx[b] = x;
        kk[b] = k;
#pragma unroll
        for (int j = 0; j < R; ++j) {
          const int e = x + j * bpl;
          cf a;
          if constexpr (HERM) {
            const int r = e < H ? e : (e == H ? 0 : m - e);
            const cf p = lds_read(&z[w.at(l, r)]), c = lds_read(&z[w.at(l + H, r)]);  // tile (row r, col l) and (row r, col l + H)
            if (e == 0) a = {p.x, c.x};
            else if (e == H) a = {p.y, c.y};
            else if (e < H) a = {p.x - c.y, p.y + c.x};
            else a = {p.x + c.y, c.x - p.y};
          } else {
            a = lds_read(&z[w.at(l, e)]);
          }
          if (j > 0 && np > 1) a = cmul(a, lds_read(&tw[j * k * tstep]));
          v[b][j] = a;
        }
        bfly_r<R>(v[b]);
      }
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < NB; ++b) {
      const int q = lane + 64 * b;
      if (q < nbf) {
        const int base = (xx[b] - kk[b]) * R + kk[b];
#pragma unroll
        for (int p = 0; p < R; ++p) z[w.at(li[b], base + p * np)] = v[b][p];
      }
    }
    wave_sync();
  }
}

// all stages of one 1-D pass over the wave's lines
template <bool HERM>
__device__ __forceinline__ void pass_lines(cf* z, const cf* tw, const PcPlan& pl, const Walk& w, int line0, int nlines, int lane) {
  int np = 1;
  for (int s = 0; s < pl.n_stages; ++s) {
    const int R = (int)((pl.radix_packed >> (4 * s)) & 15u);  // (a dynamic index into the kernel argument would go through scratch)
    const int tstep = pl.m / (np * R);
    const float inv_np = 1.0f / (float)np;
    if (HERM && s == 0) {
      switch (R) {
        case 8: stage<8, true>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        case 5: stage<5, true>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        case 4: stage<4, true>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        case 3: stage<3, true>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        default: stage<2, true>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
      }
    } else {
      switch (R) {
        case 8: stage<8, false>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        case 5: stage<5, false>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        case 4: stage<4, false>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        case 3: stage<3, false>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
        default: stage<2, false>(z, tw, w, pl.m, np, inv_np, tstep, line0, nlines, lane); break;
      }
    }
    np *= R;
  }
}

}  // namespace
}  // namespace mof
