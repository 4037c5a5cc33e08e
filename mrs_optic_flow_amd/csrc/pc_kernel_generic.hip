// pc_kernel_generic.hip -- K1g: the fused per-patch phase correlation for ANY samplePointSize, from a run-time plan.
//
// The reference takes samplePointSize from a ROS parameter (/root/reference/src/FftMethod.cpp:1680-1720,
// config/default.yaml:31-32) and hands every patch to cv::phaseCorrelate (:1836), which zero-pads it to
// M = cv::getOptimalDFTSize(N) -- the smallest 2^a 3^b 5^c >= N -- before it transforms (published OpenCV algorithm); its
// OpenCL branch plans radix-{2,3,4,5,8} passes per size (`ocl_getRadixes`, :481-539). pc_kernel.hip / pc_kernel_mixed.hip
// hold the hand-tuned instantiations (N = 32, 64, 120, 128); this kernel serves every other size whose padded M x M complex
// tile fits one CU's LDS (M <= 135), with the SAME algorithm:
//   z = cur + i prev packed into ONE complex M x M tile (zero rows / columns beyond N), forward 2-D transform in LDS,
//   untangle + normalised cross-power spectrum with the real-only-slot rule (pc_common.hpp), inverse transform --
//   Hermitian (half the lines) when M is even, a full complex one when M is odd (no Nyquist row to pack) --, first maximum
//   of the fft-shifted surface, 5 x 5 (7 x 7) centroid in fp64 around it, centre M / 2.0, gate against N / 2.
// A 1-D transform is a chain of Stockham auto-sort stages with the plan's radices; wave w owns a contiguous block of lines
// through a whole pass (stages ordered by the wave's in-order LDS queue, no workgroup barrier inside a pass); a lane carries up
// to 16 complex values = floor(16 / R) butterflies per stage, lines are packed into the wave's 64 lanes (division by the
// run-time butterfly count through a host-computed float reciprocal + one correction step, exact for every index that occurs).
// Twiddles: the M-entry table (cos, -sin computed in double on the host) is copied to LDS once per workgroup.
// This is the general path: correctness and a sane mapping first; the tuned sizes above stay the fast path.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"
#include "pc_plan.hpp"
#include "pc_plan_build.hpp"

namespace mof {

namespace {

// one pixel of the correlated image (row y, column x of the patch whose top-left pixel is `base`): as it is, through the
// node's CV_RGB2GRAY on BGR8 data (CH = 3), or as the quarter-resolution pixel of the long-range mode (DS = 4) -- the same
// three front ends as pc_field_kernel (pc_kernel.hip)
template <int DS, int CH>
__device__ __forceinline__ uint32_t fetch_px(const uint8_t* __restrict__ base, size_t pitch, int y, int x) {
  if constexpr (DS == 4) {
    const uint8_t* r1 = base + (size_t)(4 * y + 1) * pitch + 4 * x;
    const uint8_t* r2 = r1 + pitch;
    return ((uint32_t)r1[1] + r1[2] + r2[1] + r2[2] + 2u) >> 2;
  } else if constexpr (CH == 3) {
    const uint8_t* p = base + (size_t)y * pitch + 3 * x;
    return rgb2gray_fixed(p[0], p[1], p[2]);
  } else {
    return base[(size_t)y * pitch + x];
  }
}

// four consecutive pixels x0 .. x0 + 3 of row y, one byte each (x0 + 3 inside the patch): ONE unaligned dword of a gray frame, three dwords
// of a BGR8 frame (12 bytes -> four CV_RGB2GRAY values, as K1h's px_gray), or -- the long-range mode -- the 2 x 2 taps of four quarter-resolution
// pixels from two 16-byte runs of the frame rows 4 y + 1 and 4 y + 2 (cv::resize(1/4, INTER_LINEAR): columns 4 x + 1 and 4 x + 2)
template <int DS, int CH>
__device__ __forceinline__ uint32_t fetch_px4(const uint8_t* __restrict__ base, size_t pitch, int y, int x0) {
  if constexpr (DS == 4) {
    const uint8_t* r1 = base + (size_t)(4 * y + 1) * pitch + 4 * (size_t)x0;
    uint32_t w1[4], w2[4];
    __builtin_memcpy(w1, r1, 16);
    __builtin_memcpy(w2, r1 + pitch, 16);
    uint32_t g = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b)
      g |= ((((w1[b] >> 8) & 0xffu) + ((w1[b] >> 16) & 0xffu) + ((w2[b] >> 8) & 0xffu) + ((w2[b] >> 16) & 0xffu) + 2u) >> 2) << (8 * b);
    return g;
  } else if constexpr (CH == 3) {
    uint32_t w[3];
    __builtin_memcpy(w, base + (size_t)y * pitch + 3 * (size_t)x0, 12);
    uint32_t g = 0;
#pragma unroll
    for (int b = 0; b < 4; ++b) {
      const int i = 3 * b;
      const uint32_t c0 = (w[i >> 2] >> (8 * (i & 3))) & 0xffu, c1 = (w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xffu,
                     c2 = (w[(i + 2) >> 2] >> (8 * ((i + 2) & 3))) & 0xffu;
      g |= rgb2gray_fixed(c0, c1, c2) << (8 * b);
    }
    return g;
  } else {
    uint32_t w;
    __builtin_memcpy(&w, base + (size_t)y * pitch + x0, 4);
    return w;
  }
}

}  // namespace

// the planned kernel's arg-max as a sink of the inverse transform's last stage (pc_plan.hpp: stage_rt): line c, element y of the
// Hermitian column-pair pass carries the surface values at (y, c) and (y, c + H)
struct ScanSink {
  static constexpr bool active = true;
  Best* best;
  int m, H;
  __device__ __forceinline__ void operator()(int c, int y, cf v) const {
    const int ys = y + H >= m ? y + H - m : y + H;
    *best = better(*best, Best{v.x, ys * m + c + H});  // x = c < H -> shifted c + H
    *best = better(*best, Best{v.y, ys * m + c});      // x = c + H -> shifted c
  }
};

// the conjugated normalised cross-power spectrum as the source of the inverse row pass (rows 1 .. H-1; row 0 holds the packed rows
// 0 and H already): bin (v, u) from Z(v, u) and its partner Z(m - v, m - u) in the untouched lower half of the tile
struct XpowSrc {
  static constexpr bool active = true;
  int m, H, pitch, skew_mask;
  bool box_zeros;
  int zq;  // box_zeros: the constant box's exact-zero lines are the multiples of zq (pc_common.hpp, box_zero_period)
  __device__ __forceinline__ cf operator()(const cf* z, int v, int u) const {
    const cf zk = lds_read(&z[v * pitch + u + ((u >> 3) & skew_mask)]);
    if (v == 0) return zk;
    const int um = u == 0 ? 0 : m - u;
    const cf zm = lds_read(&z[(m - v) * pitch + um + ((um >> 3) & skew_mask)]);
    cf C = cross_power<0>(zk, zm, false);
    if (box_zeros && (box_zero_line(u, zq) || box_zero_line(v, zq))) C = {0.f, 0.f};
    return {C.x, -C.y};
  }
};

// MS > 0: the instantiation for ONE transform size, whose plan is a compile-time constant (pc_static_plan(MS) is what the host
// builds for it; only n -- the unpadded size -- and the launch geometry stay run-time values): radices, strides, divisions and
// loop counts fold away. MS = 0: the plan is read from the argument (sizes below 16, and the BGR / long-range / OpenCL-model
// front ends, which keep one general kernel each).
template <int MS>
struct StaticPlanOf {
  static constexpr PcPlan P = pc_static_plan(MS > 0 ? MS : 16);
  static constexpr int T = MS > 0 ? P.threads : 1024;
  static constexpr int WPE = MS > 0 ? pc_plan_waves_per_eu(P) : 1;
  static_assert(MS == 0 || P.threads > 0, "no static plan for this size");
};

#ifndef MOF_PLANNED_XPOW_FUSED  // 1: the cross-power spectrum formed on the way into the inverse row pass (XpowSrc). Measured r04, same box:
                                // SLOWER (p96 753 -> 689 k, p60 1.05 -> 1.03 M pairs/s): it puts the whole cross-power on the waves that own
                                // the H inverse rows, the sweep of its own spreads it over every thread (as K1 found at N = 128)
#define MOF_PLANNED_XPOW_FUSED 0
#endif
#ifndef MOF_PLANNED_SCAN_FUSED  // 0: the arg-max as a sweep of its own (A/B)
#define MOF_PLANNED_SCAN_FUSED 1
#endif
#ifndef MOF_GABL  // diagnostic builds (results wrong by design): 1 no transform passes, 2 no cross-power, 3 no arg-max scan, 4 no pixel loads,
                  // 5 no centroid tail, 6 no load phase at all, 7 none of 1-6 (what is left: launch, twiddles, barriers)
#define MOF_GABL 0
#endif
template <int DS, int CH, int PK, int MS>
__global__ void __launch_bounds__(StaticPlanOf<MS>::T, StaticPlanOf<MS>::WPE) pc_generic_kernel(PcArgs a, PcPlan pl_arg) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem_g[];
  PcPlan pl = pl_arg;
  if constexpr (MS > 0) {
    constexpr PcPlan SP = StaticPlanOf<MS>::P;
    pl.m = SP.m;
    pl.pitch = SP.pitch;
    pl.skew_mask = SP.skew_mask;
    pl.threads = SP.threads;
    pl.n_stages = SP.n_stages;
    pl.radix_packed = SP.radix_packed;
    pl.hermitian = SP.hermitian;
  }
  const int m = pl.m, n = pl.n, H = m >> 1, T = MS > 0 ? pl.threads : (int)blockDim.x, WAVES = T >> 6;
  cf* z = reinterpret_cast<cf*>(smem_g);
  cf* tw = z + (size_t)m * pl.pitch;
  Best* red = reinterpret_cast<Best*>(tw + m);
  int* flags = reinterpret_cast<int*>(red + 16);  // [0] cur differs from its first pixel, [1] prev does, [2] C_dc bits, [3] / [4] first pixel of cur / prev
  cf* dbox = reinterpret_cast<cf*>(flags + 16);   // [m] spectrum of the zero-padded constant line (one_box, below)
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  const float inv_m = 1.0f / (float)m;
  const bool herm = pl.hermitian != 0;
  // the element loops of the phases between the passes (cross-power, arg-max scan): with a compile-time plan their trip counts are
  // constants and a few trips in flight hide the LDS and sqrt / reciprocal latencies (p60, phase ablation: those phases were more
  // than half of the kernel); the run-time-plan form keeps them rolled (registers)
  constexpr int UNR = MS > 0 ? 4 : 1;
  // (i / m, i % m): by a constant under a compile-time plan (multiply-high), by the float reciprocal otherwise
  auto divmod_m = [&](int i, int* rem) -> int {
    if constexpr (MS > 0) {
      constexpr int M_ = StaticPlanOf<MS>::P.m;
      *rem = i % M_;
      return i / M_;
    } else {
      return fdiv(i, m, inv_m, rem);
    }
  };

  // ---- patch origin (one workgroup per patch on a 3-D grid: column, row, pair)
  const int px0 = a.origin_x + (int)blockIdx.x * a.stride_x, py0 = a.origin_y + (int)blockIdx.y * a.stride_y;
  const size_t poff = (size_t)(DS * py0) * a.pitch + (size_t)(CH * DS * px0);
  const uint8_t* cur = a.cur + (size_t)blockIdx.z * a.cur_stride + poff;
  const uint8_t* prev = a.prev + (size_t)blockIdx.z * a.prev_stride + poff;
  const size_t p = ((size_t)blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x;

  if (tid < 2) flags[tid] = 0;
#pragma unroll 1
  for (int k = tid; k < m; k += T) tw[k] = {a.twiddles[2 * k], a.twiddles[2 * k + 1]};

  // ---- load: u8 -> f32 (convertTo, :1805-1806), z = cur + i prev, zero rows / columns beyond N (copyMakeBorder of
  //      cv::phaseCorrelate); constant patches are detected here with integer compares (pc_common.hpp, degenerate pairs)
  {
    const uint32_t c00 = fetch_px<DS, CH>(cur, a.pitch, 0, 0), p00 = fetch_px<DS, CH>(prev, a.pitch, 0, 0);
    uint32_t dc = 0u, dp = 0u;
    // Pixel loads go out in batches of SB per lane and image before the first of a batch is used (a load -> convert -> store
    // loop pays the memory latency once per trip -- 16 trips per patch in the first form of this kernel; one batch of 18 held
    // 36 pixel registers plus their addresses and spilled 646 VGPRs). Element i = tid + t T of the padded tile sits at
    // (y, x) = divmod(i, m); the pair advances by divmod(T, m) per step -- one division per lane instead of one per element.
    // FOUR pixels per load on every front end (r06; until then only gray frames under a compile-time plan -- BGR8, the long-range mode
    // and the OpenCL peak model's run-time-plan kernels issued up to 18 x 2 byte loads per lane, 4 x that at DS = 4): chunk q = tid + k T
    // of the padded tile's rows of ceil(m / 4) chunks (fetch_px4: one dword, three dwords of BGR8, or two 16-byte runs of the tapped
    // quarter-resolution rows). Compile-time plan: every division is by a constant and all of a lane's chunks are in flight together;
    // run-time plan: sub-batches of CMAX chunks (the plan sizes the workgroup so that a lane owns at most 18 elements = 5 chunks).
    {
      constexpr int M_ = MS > 0 ? StaticPlanOf<MS>::P.m : 0, T_ = MS > 0 ? StaticPlanOf<MS>::T : 1;
      constexpr int CMAX = MS > 0 ? (M_ * ((M_ + 3) / 4) + T_ - 1) / T_ : 6;
      const int cpr = (m + 3) >> 2, nchunks = m * cpr;
      const float inv_cpr = 1.0f / (float)cpr;
      const uint32_t c4 = c00 * 0x01010101u, p4 = p00 * 0x01010101u;
      auto coords = [&](int q, int* x0) -> int {  // chunk q -> (row, first column)
        int r;
        int y;
        if constexpr (MS > 0) {
          constexpr int CPR_ = (M_ + 3) / 4;
          y = q / CPR_;
          r = q % CPR_;
        } else {
          y = fdiv(q, cpr, inv_cpr, &r);
        }
        *x0 = 4 * r;
        return y;
      };
#pragma unroll 1
      for (int q0 = tid; q0 < ((MOF_GABL == 6 || MOF_GABL == 7) ? 0 : nchunks); q0 += CMAX * T) {
        uint32_t cw[CMAX], pw[CMAX];
#pragma unroll
        for (int k = 0; k < CMAX; ++k) {
          const int q = q0 + k * T;
          int x0;
          const int y = coords(q, &x0);
          cw[k] = pw[k] = 0u;
          if (MOF_GABL != 4 && q < nchunks && y < n && x0 < n) {
            if (x0 + 3 < n) {
              cw[k] = fetch_px4<DS, CH>(cur, a.pitch, y, x0);
              pw[k] = fetch_px4<DS, CH>(prev, a.pitch, y, x0);
            } else {  // the last chunk of a row whose length is not a multiple of four: the pixels inside the patch
              for (int b = 0; x0 + b < n; ++b) {
                cw[k] |= fetch_px<DS, CH>(cur, a.pitch, y, x0 + b) << (8 * b);
                pw[k] |= fetch_px<DS, CH>(prev, a.pitch, y, x0 + b) << (8 * b);
              }
            }
          }
        }
#pragma unroll
        for (int k = 0; k < CMAX; ++k) {
          const int q = q0 + k * T;
          int x0;
          const int y = coords(q, &x0);
          if (q >= nchunks) continue;
          if (y < n && x0 < n) {
            const uint32_t inside = x0 + 3 < n ? 0xffffffffu : (1u << (8 * (n - x0))) - 1u;
            dc |= (cw[k] ^ c4) & inside;
            dp |= (pw[k] ^ p4) & inside;
          }
          cf* row = z + y * pl.pitch;
#pragma unroll
          for (int b = 0; b < 4; ++b) {
            const int x = x0 + b;
            if (x < m) row[x + ((x >> 3) & pl.skew_mask)] = {(float)((cw[k] >> (8 * b)) & 0xffu), (float)((pw[k] >> (8 * b)) & 0xffu)};
          }
        }
      }
    }
    __syncthreads();  // flags[] zeroed, twiddles in place
    if (__builtin_amdgcn_ballot_w64(dc != 0u) != 0ull && lane == 0) flags[0] = 1;
    if (__builtin_amdgcn_ballot_w64(dp != 0u) != 0ull && lane == 0) flags[1] = 1;
    if (tid == 0) flags[3] = (int)c00, flags[4] = (int)p00;
  }
  __syncthreads();

  // EXACTLY ONE constant patch that zero padding turned into an n x n box: its spectrum is known in closed form,
  // B[v][u] = level D[v] D[u] with D the transform of n ones in a line of m. The packed transform would deliver it with the
  // rounding noise of the TEXTURED patch's spectrum on top (1e-7 of 1e5 against box bins that fall to zero towards the Nyquist
  // lines): 1e-3 px on a 118 x 118 constant-against-texture pair (tools/fft_sr_fuzz.py seed 202). So D goes into LDS here (m sums
  // of n table twiddles) and the cross-power below takes the box from it and the textured spectrum as Z -+ i box: no untangle.
  const bool one_box = m > n && ((flags[0] == 0) != (flags[1] == 0));
  const int zq = box_zero_period(n, m);  // exact-zero lines of a constant n x n box in the m x m tile: the multiples of zq
  if (one_box) {
#pragma unroll 1
    for (int k = tid; k < m; k += T) {
      double ax = 0.0, ay = 0.0;  // (f64 sums: 124 f32 additions lose 4e-4 of D, which is 1e-3 px on a 124 x 124 box in a 125 tile)
      int idx = 0;
#pragma unroll 1
      for (int j = 0; j < n; ++j) {
        const cf t = tw[idx];
        ax += (double)t.x;
        ay += (double)t.y;
        idx += k;
        idx = idx >= m ? idx - m : idx;
      }
      dbox[k] = box_zero_line(k, zq) ? cf{0.f, 0.f} : cf{(float)ax, (float)ay};  // (the box's exact zeros: whole periods of the twiddle)
    }
  }
  const Walk rows = {pl.pitch, 1, 0, pl.skew_mask, 0}, cols = {1, pl.pitch, pl.skew_mask, 0, 1};
  auto zat = [&](int r, int c) -> cf& { return z[r * pl.pitch + c + ((c >> 3) & pl.skew_mask)]; };
  // wave w owns lines [w lpw, min((w + 1) lpw, L))
  auto my_lines = [&](int L, int* l0, int* nl) {
    const int lpw = (L + WAVES - 1) / WAVES;
    *l0 = wave * lpw;
    int cnt = L - *l0;
    cnt = cnt < 0 ? 0 : (cnt > lpw ? lpw : cnt);
    *nl = cnt;
  };

  auto run_pass = [&](const Walk& w, int l0, int nl, bool h) {
    if (MOF_GABL == 1 || MOF_GABL == 7) return;
    if constexpr (MS > 0) pass_lines_static<StaticPlanOf<MS>>(z, tw, w, l0, nl, lane, h);
    else pass_lines<false, false>(z, tw, pl, w, l0, nl, lane, h);
  };
  // ---- forward 2-D transform (dft x2, :1491-1493)
  {
    int l0, nl;
    my_lines(m, &l0, &nl);
    if (nl > 0) run_pass(rows, l0, nl, false);
    __syncthreads();
    if (nl > 0) run_pass(cols, l0, nl, false);
    __syncthreads();
  }

  // ---- untangle, P = A conj(B), C = P |P| / (|P|^2 + eps) with the real-only-slot rule (mulSpectrums :1494, magSpectrums
  //      :70-168, divSpectrums :1086-1251), stored conjugated for the inverse (a forward transform of conj C)
  // A CONSTANT patch that zero padding turned into an n x n box (n, m even): the box's spectrum is EXACTLY zero on the Nyquist
  // row and column in the reference's separate transforms (alternating sums of equal numbers), so P = 0 and C = 0 there -- 127 of
  // 4096 bins at n = 62 -- while the packed transform leaks ~1e-7 of the other patch's spectrum into them, which the
  // normalisation blows up to unit magnitude: 0.5 px off on a constant-against-texture pair (found by the seeded fuzzer classes,
  // r04). The constant patch is known exactly (flags), so are its zero bins.
  const bool box_zeros = m > n && (flags[0] == 0 || flags[1] == 0);
  const bool box_is_cur = flags[0] == 0;
  const float box_level = (float)(box_is_cur ? flags[3] : flags[4]);
  // cross-power of bin (v, u): from the packed pair (untangle), or -- one_box -- from the closed-form box and Z(v, u) alone
  auto xpow = [&](int v, int u, int vm, int um, bool real_only) -> cf {
    if (!one_box) return cross_power<PK>(zat(v, u), zat(vm, um), real_only);
    const cf dv = dbox[v], du = dbox[u], zk = zat(v, u);
    const cf bx = {box_level * (dv.x * du.x - dv.y * du.y), box_level * (dv.x * du.y + dv.y * du.x)};
    cf A, B;
    if (box_is_cur) {
      A = bx;
      B = {zk.y - bx.y, bx.x - zk.x};  // -i (Z - A)
    } else {
      B = bx;
      A = {zk.x + bx.y, zk.y - bx.x};  // Z - i B
    }
    return cross_power_ab<PK>(cf{2.f * A.x, 2.f * A.y}, cf{2.f * B.x, 2.f * B.y}, real_only);
  };
  // Compile-time plan, even M, cv::phaseCorrelate's model, no closed-form box: rows 1 .. H-1 get their cross-power spectrum on the
  // way INTO the inverse row pass (XpowSrc) -- one tile sweep (a write and a read of half the tile, its index arithmetic) less
  constexpr bool XPOW_FUSABLE = MS > 0 && PK == 0 && (StaticPlanOf<MS>::P.m % 2 == 0) && MOF_PLANNED_XPOW_FUSED;
  const bool xpow_fused = XPOW_FUSABLE && !one_box;
  if (MOF_GABL == 2 || MOF_GABL == 7) {
  } else if (herm) {
    // rows 1 .. H-1, every u: the partner (m - v, m - u) lies in the untouched lower half
    // (a counted loop: with a compile-time plan the trip count is a constant and UNR trips are in flight)
    const int xtrips = xpow_fused ? 0 : ((H - 1) * m + T - 1) / T;
    if (!one_box) {  // (the hot loop keeps the packed form alone: the closed-form branch inside it cost 4 % on every pair)
#pragma clang loop unroll_count(UNR)
      for (int k = 0; k < xtrips; ++k) {
        const int i = tid + k * T;
        if (i < (H - 1) * m) {
          int u;
          const int v = 1 + divmod_m(i, &u);
          const int um = u == 0 ? 0 : m - u;
          cf C = cross_power<PK>(zat(v, u), zat(m - v, um), false);
          if (box_zeros && (box_zero_line(u, zq) || box_zero_line(v, zq))) C = {0.f, 0.f};
          zat(v, u) = {C.x, -C.y};
        }
      }
    } else {
#pragma unroll 1
      for (int k = 0; k < xtrips; ++k) {
        const int i = tid + k * T;
        if (i < (H - 1) * m) {
          int u;
          const int v = 1 + divmod_m(i, &u);
          cf C = xpow(v, u, m - v, u == 0 ? 0 : m - u, false);
          if (box_zeros && (box_zero_line(u, zq) || box_zero_line(v, zq))) C = {0.f, 0.f};
          zat(v, u) = {C.x, -C.y};
        }
      }
    }
    // rows 0 and H share row 0: G[u] = conj C[0][u] + i conj C[H][u]; the partner of u is m - u in the same rows
#pragma unroll 1
    for (int u = tid; u <= H; u += T) {
      const int um = u == 0 ? 0 : m - u;
      const bool self = u == um;
      cf C0 = xpow(0, u, 0, um, self);
      cf Ch = xpow(H, u, H, um, self);
      if (box_zeros) {  // C0 = bin (0, u), Ch = bin (M/2, u)
        if (box_zero_line(H, zq) || box_zero_line(u, zq)) Ch = {0.f, 0.f};
        if (box_zero_line(u, zq)) C0 = {0.f, 0.f};
      }
      if (u == 0) flags[2] = __float_as_int(C0.x);  // C_dc: all that is left of a degenerate pair's spectrum
      zat(0, u) = {C0.x + Ch.y, Ch.x - C0.y};
      if (!self) zat(0, um) = {C0.x - Ch.y, Ch.x + C0.y};
    }
  } else {
    // odd M: no Nyquist row to pack, the only real-only slot is DC; every bin pair (k, -k) is formed once
#pragma unroll 1
    for (int i = tid; i < m * m; i += T) {
      int u;
      const int v = divmod_m(i, &u);
      const int vm = v == 0 ? 0 : m - v, um = u == 0 ? 0 : m - u;
      const int ip = vm * m + um;
      if (i > ip) continue;
      cf C = xpow(v, u, vm, um, i == 0);
      if (box_zeros && (box_zero_line(u, zq) || box_zero_line(v, zq))) C = {0.f, 0.f};  // (odd M: e.g. 130 in 135 has zero lines at 27 k)
      if (i == 0) flags[2] = __float_as_int(C.x);
      zat(v, u) = {C.x, -C.y};
      if (i != ip) zat(vm, um) = {C.x, C.y};  // C[-k] = conj C[k]
    }
  }
  __syncthreads();

  // Compile-time plan, even M, cv::phaseCorrelate's peak model: the first maximum of the shifted surface (fftShift :1257-1323,
  // minMaxLoc :1539) is taken from the registers of the inverse transform's last stage -- the separate sweep over the tile was 9 %
  // of the kernel (p60, tools/ab_planned_phases.sh). Ties go to the smaller shifted index either way (better()).
  constexpr bool SCAN_FUSED = MS > 0 && PK == 0 && (StaticPlanOf<MS>::P.m % 2 == 0) && MOF_PLANNED_SCAN_FUSED;
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  // ---- inverse (unscaled, idft :1497) as forward transforms of conj C
  {
    int l0, nl;
    my_lines(herm ? H : m, &l0, &nl);
    if constexpr (XPOW_FUSABLE) {
      if (xpow_fused) {
        if (nl > 0 && MOF_GABL != 1 && MOF_GABL != 7)
          pass_lines_static<StaticPlanOf<MS>, 0, 1, NoSink, XpowSrc>(z, tw, rows, l0, nl, lane, false, NoSink{},
                                                                     XpowSrc{m, H, pl.pitch, pl.skew_mask, box_zeros, zq});
      } else if (nl > 0) {
        run_pass(rows, l0, nl, false);
      }
    } else {
      if (nl > 0) run_pass(rows, l0, nl, false);
    }
    __syncthreads();
    // herm: column pairs (c, c + H), z(y, c) = (S[y][c], S[y][c + H]); else z(y, x).x = S[y][x]
    if constexpr (SCAN_FUSED) {
      // the arg-max rides the last stage: line c, element y carries the surface values at (y, c) and (y, c + H)
      if (nl > 0 && MOF_GABL != 1 && MOF_GABL != 7)
        pass_lines_static<StaticPlanOf<MS>, 0, 1, ScanSink>(z, tw, cols, l0, nl, lane, herm, ScanSink{&best, m, H});
    } else {
      if (nl > 0) run_pass(cols, l0, nl, herm);
    }
    __syncthreads();
  }

  // ---- surface value at UN-shifted (y, x), incl. the OpenCL model's scaling and +-search_radius mask (cl:733, :737-746)
  const float ocl_scale = 1.0f / (float)(m * m);
  auto surf = [&](int y, int x) -> float {
    float s;
    if (herm) {
      const cf t = zat(y, x < H ? x : x - H);
      s = x < H ? t.x : t.y;
    } else {
      s = zat(y, x).x;
    }
    if constexpr (PK == 1) {
      const int sr = a.search_radius;
      const bool masked = (y > sr && y < m - sr) || (x > sr && x < m - sr);
      s = masked ? 0.f : s * ocl_scale;
    }
    return s;
  };

  // ---- first maximum of the fft-shifted surface in row-major order (fftShift :1257-1323: index i -> (i + (m >> 1)) mod m for
  //      even and odd m alike; minMaxLoc :1539)
#pragma clang loop unroll_count(UNR)
  for (int y = wave; y < ((SCAN_FUSED || MOF_GABL == 3 || MOF_GABL == 7) ? 0 : m); y += WAVES) {
    const int ys = y + H >= m ? y + H - m : y + H;
#pragma unroll 1
    for (int x = lane; x < m; x += 64) {
      const int xs = x + H >= m ? x + H - m : x + H;
      best = better(best, Best{surf(y, x), ys * m + xs});
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();

  // ---- weighted centroid in double + validity gate (:1337-1383, :1838-1856), wave 0
  if ((MOF_GABL == 5 || MOF_GABL == 7) && tid == 0) a.out[2 * p] = a.out[2 * p + 1] = (double)best.v;
  if (wave == 0 && MOF_GABL != 5 && MOF_GABL != 7) {
    for (int w = 1; w < WAVES; ++w) best = better(best, red[w]);
    constexpr int RAD = PeakModel<PK>::RAD, W = PeakModel<PK>::W;
    const bool have = best.idx != 0x7fffffff;
    const int py = have ? best.idx / m : 0, pxk = have ? best.idx - py * m : 0;
    const int ys = py - RAD + lane / W, xs = pxk - RAD + lane % W;
    double val = 0.0;
    if (have && lane < W * W && ys >= 0 && ys <= m - 1 && xs >= 0 && xs <= m - 1) {  // window clamped to the (padded) patch
      const int y = ys - H < 0 ? ys - H + m : ys - H, x = xs - H < 0 ? xs - H + m : xs - H;  // un-shifted position
      const float v = surf(y, x);
      val = (double)((PK == 1 && !(v > 0.f)) ? 0.f : v);
    }
    double cx = (double)xs * val, cy = (double)ys * val, sum = val;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      cx += __shfl_xor(cx, off, 64);
      cy += __shfl_xor(cy, off, 64);
      sum += __shfl_xor(sum, off, 64);
    }
    if (lane == 0) {
      sum += PK == 1 ? 1.1920928955078125e-07 : 2.220446049250313e-16;  // FLT_EPSILON cl:1342 / DBL_EPSILON :1378
      // shift = -(center - t) = t - M / 2.0 (:1836): cv::phaseCorrelate's centre is that of the PADDED image
      const double half_m = (double)m / 2.0, half_n = (double)n / 2.0;
      double sx = cx / sum - half_m, sy = cy / sum - half_m;
      // a constant patch: its separate transform is exactly zero off DC, the surface is flat = C_dc (pc_common.hpp). With
      // padding (m > n) only the all-zero patch stays constant on the padded image.
      const bool cconst = flags[0] == 0, pconst = flags[1] == 0;
      const bool degenerate = m == n ? (cconst || pconst)
                                     : ((cconst && fetch_px<DS, CH>(cur, a.pitch, 0, 0) == 0u) || (pconst && fetch_px<DS, CH>(prev, a.pitch, 0, 0) == 0u));
      if (degenerate) {
        if constexpr (PK == 1) {
          sx = sy = __builtin_nan("");
        } else {
          const double c9 = 9.0 * (double)__int_as_float(flags[2]);
          sx = sy = (c9 > 0.0 ? c9 / (c9 + 2.220446049250313e-16) : 0.0) - half_m;
        }
      }
      // the gate compares with samplePointSize / 2 -- the UNPADDED size (:1841-1842)
      const bool bad = (sx * sx + sy * sy > a.max_px_speed_sq) || (fabs(sx) > half_n) || (fabs(sy) > half_n) || (sx != sx) ||
                       (sy != sy) || (!have && !degenerate);
      if (bad) sx = sy = __builtin_nan("");
      a.out[2 * p] = sx;
      a.out[2 * p + 1] = sy;
    }
  }
}

// ---- host: the plan (pc_plan_build.hpp holds it as constexpr functions; these are its run-time entry points) --------------

int pc_optimal_dft_size(int n) { return pc_optimal_dft_size_c(n); }
int pc_radix_chain(int m, int* radix, int max_stages) { return pc_radix_chain_c(m, radix, max_stages); }
bool pc_build_plan(int n, PcPlan* out) {
  PcPlan pl{};
  if (!pc_tile_plan_c(n, pl)) return false;
  *out = pl;
  return true;
}

// the transform sizes with a compile-time instantiation: every 5-smooth m in [16, 135] (what getOptimalDFTSize can return there)
#define MOF_STATIC_SIZES(X) \
  X(16) X(18) X(20) X(24) X(25) X(27) X(30) X(32) X(36) X(40) X(45) X(48) X(50) X(54) X(60) X(64) X(72) X(75) X(80) X(81) \
  X(90) X(96) X(100) X(108) X(120) X(125) X(128) X(135)

template <int DS, int CH, int PK, int MS>
static hipError_t configure_generic_one() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_generic_kernel<DS, CH, PK, MS>), hipFuncAttributeMaxDynamicSharedMemorySize,
                             160 * 1024);
}

hipError_t pc_configure_generic() {
  hipError_t e;
  if ((e = configure_generic_one<1, 1, 0, 0>()) != hipSuccess) return e;
  if ((e = configure_generic_one<1, 3, 0, 0>()) != hipSuccess) return e;
  if ((e = configure_generic_one<4, 1, 0, 0>()) != hipSuccess) return e;
  if ((e = configure_generic_one<1, 1, 1, 0>()) != hipSuccess) return e;
  if ((e = configure_generic_one<1, 3, 1, 0>()) != hipSuccess) return e;
  if ((e = configure_generic_one<4, 1, 1, 0>()) != hipSuccess) return e;
#define X(M)                                                                    \
  if ((e = configure_generic_one<1, 1, 0, M>()) != hipSuccess) return e;          \
  if ((e = configure_generic_one<1, 3, 0, M>()) != hipSuccess) return e;          \
  if ((e = configure_generic_one<4, 1, 0, M>()) != hipSuccess) return e;
  MOF_STATIC_SIZES(X)
#undef X
  return hipSuccess;
}

hipError_t launch_pc_generic(const PcArgs& a_in, const PcPlan& pl, int n_pairs, hipStream_t stream) {
  if (a_in.downscale == 4 && a_in.channels == 3) return hipErrorInvalidValue;
  if (pl.threads < 64 || pl.threads > 1024 || pl.lds_bytes > 160 * 1024) return hipErrorInvalidValue;
  // MOF_PLANNED_STATIC=0: the run-time plan also where a compile-time instantiation exists (A/B and the tests of that form)
  static const bool use_static = [] { const char* v = getenv("MOF_PLANNED_STATIC"); return !v || atoi(v) != 0; }();
  const int patches = a_in.grid_x * a_in.grid_y;
  const dim3 b((unsigned)pl.threads);
  for (int k0 = 0; k0 < n_pairs; k0 += 65535) {  // the pair index rides gridDim.z
    const int nk = n_pairs - k0 < 65535 ? n_pairs - k0 : 65535;
    PcArgs c = a_in;
    c.cur = a_in.cur + (size_t)k0 * a_in.cur_stride;
    c.prev = a_in.prev + (size_t)k0 * a_in.prev_stride;
    c.out = a_in.out + (size_t)k0 * patches * 2;
    c.total = nk * patches;
    const dim3 g((unsigned)c.grid_x, (unsigned)c.grid_y, (unsigned)nk);
    if (c.peak_model == 1) {
      if (c.downscale == 4) hipLaunchKernelGGL((pc_generic_kernel<4, 1, 1, 0>), g, b, (size_t)pl.lds_bytes, stream, c, pl);
      else if (c.channels == 3) hipLaunchKernelGGL((pc_generic_kernel<1, 3, 1, 0>), g, b, (size_t)pl.lds_bytes, stream, c, pl);
      else hipLaunchKernelGGL((pc_generic_kernel<1, 1, 1, 0>), g, b, (size_t)pl.lds_bytes, stream, c, pl);
    } else if (c.downscale == 4) {
      // the long-range mode (FftMethod.cpp:1905-2007) under the transform size's compile-time plan too (r06: lr60 / lr96 ran the run-time-plan
      // kernel at a third of the gray compile-time-plan rate)
      bool done = false;
      if (use_static) {
        switch (pl.m) {
#define X(M)                                                                                                                                                    \
  case M:                                                                                                                                                       \
    hipLaunchKernelGGL((pc_generic_kernel<4, 1, 0, M>), g, dim3((unsigned)StaticPlanOf<M>::T), (size_t)StaticPlanOf<M>::P.lds_bytes, stream, c, pl);           \
    done = true;                                                                                                                                                \
    break;
          MOF_STATIC_SIZES(X)
#undef X
          default: break;
        }
      }
      if (!done) hipLaunchKernelGGL((pc_generic_kernel<4, 1, 0, 0>), g, b, (size_t)pl.lds_bytes, stream, c, pl);
    } else {
      // gray and BGR8 frames (the latter promise the gray path's bits, include/mof.h): the compile-time instantiation of the
      // transform size where there is one
      const bool bgr = c.channels == 3;
      bool done = false;
      if (use_static) {
        switch (pl.m) {
#define X(M)                                                                                                                            \
  case M:                                                                                                                               \
    if (bgr) hipLaunchKernelGGL((pc_generic_kernel<1, 3, 0, M>), g, dim3((unsigned)StaticPlanOf<M>::T), (size_t)StaticPlanOf<M>::P.lds_bytes, stream, c, pl); \
    else hipLaunchKernelGGL((pc_generic_kernel<1, 1, 0, M>), g, dim3((unsigned)StaticPlanOf<M>::T), (size_t)StaticPlanOf<M>::P.lds_bytes, stream, c, pl);     \
    done = true;                                                                                                                        \
    break;
          MOF_STATIC_SIZES(X)
#undef X
          default: break;
        }
      }
      if (!done) {
        if (bgr) hipLaunchKernelGGL((pc_generic_kernel<1, 3, 0, 0>), g, b, (size_t)pl.lds_bytes, stream, c, pl);
        else hipLaunchKernelGGL((pc_generic_kernel<1, 1, 0, 0>), g, b, (size_t)pl.lds_bytes, stream, c, pl);
      }
    }
  }
  return hipGetLastError();
}

}  // namespace mof
