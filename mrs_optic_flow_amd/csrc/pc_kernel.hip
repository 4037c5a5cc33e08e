// pc_kernel.hip -- K1: fused per-patch FFT phase correlation for gfx950 (CDNA4).
//
// One workgroup owns one patch pair and never leaves the CU: the two u8 patches are read once
// from HBM (16 B per lane, any byte alignment), packed as z = cur + i*prev into ONE complex N x N
// tile in LDS ("two-for-one" real transform), transformed, untangled into the two real spectra,
// turned into the normalised cross-power spectrum, inverse-transformed exploiting its Hermitian
// symmetry (half the work of a complex inverse), and reduced to (arg-max, 5x5 centroid): 16 B leave
// the CU per patch. Nothing but the frames and the results touches HBM.
//
// Replaces, per patch, the reference's
//   -cv::phaseCorrelate(cur(roi), prev(roi))              src/FftMethod.cpp:1836
// whose stages the reference spells out at src/FftMethod.cpp:1487-1498, with the helper semantics of
// :70-168 (magSpectrums), :1086-1251 (divSpectrums), :1257-1323 (fftShift), :1329-1385
// (weightedCentroid), followed by the validity gate of :1838-1856.
//
// CDNA4 mapping
//   * N*N/16 threads; each 1-D transform is two Stockham stages N = R1*R2 with 16 points per lane in
//     registers (64 = 8*8, 128 = 16*8, 32 = 8*4), twiddles in registers (host-computed in double);
//   * wave w owns rows (then columns) [w*L, (w+1)*L) through a whole 1-D pass, so the stages of a
//     pass are ordered by the wave's in-order LDS queue and need no workgroup barrier: 5 barriers
//     per patch in total;
//   * LDS tile is skewed (element c of a row sits at c + (c >> SK), row pitch = 8 mod 16 complex) so
//     that the stride-R accesses of the Stockham stages and the column walks are bank-conflict free;
//   * LDS writes cost 3x reads on this chip (MI355X_MICROARCH.md, LDS table): the Hermitian inverse
//     writes half a tile per stage and the arg-max is taken from registers.
//
// Arithmetic: fp32 transforms (as OpenCV for CV_32F), fp64 centroid and gate.
// Algorithmic HBM bytes per patch: 2*N*N in + 16 out.

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <stdint.h>
#include <stdlib.h>

#include "mfma_frag.hpp"
#include "mof_kernels.h"
#include "pc_common.hpp"

#include "pc_passes.hpp"
#include "pc_passes3.hpp"

namespace mof {

// DS = 1: patches are read from the frames as they are. DS = 4: long-range mode -- every patch pixel is the
// quarter-resolution pixel cv::resize(.., 1/4, 1/4, INTER_LINEAR) would produce (FftMethod.cpp:1931-1932), i.e.
// the rounded mean of the 2x2 centre of a 4x4 cell, formed on the fly from the full-resolution frame.
// CH = 3: the frames are interleaved BGR8 and the CV_RGB2GRAY conversion of the node's front end
// (optic_flow.cpp:1622) is fused into the load, so raw camera frames are read from HBM exactly once.
// PK = 1: the peak model of the reference's useOCL=true branch (cl/FftMethod.cl) instead of cv::phaseCorrelate's.
#ifndef MOF_K1_FWD3
#define MOF_K1_FWD3 1
#endif
#ifndef MOF_K1_MFMA_S1  // S1 of the three-stage forward transform on the matrix cores (pc_passes3.hpp)
#define MOF_K1_MFMA_S1 0
#endif
template <int N, int DS, int CH, int PK>
__global__ void __launch_bounds__(PcTraits<N>::T) pc_field_kernel(PcArgs a) {
  static_assert(CH == 1 || (CH == 3 && DS == 1), "BGR front end only for the full-resolution path");
  using P = PcTraits<N>;
  constexpr int T = P::T, H = N / 2, R1 = P::R1, R2 = P::R2, LPW = P::LPW;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  Best* red = reinterpret_cast<Best*>(z + P::TILE);
  int* const_code = reinterpret_cast<int*>(red + 32);  // [16] per-wave constant-patch codes (cur | prev << 16), then C_dc (pc_common.hpp)

  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6;
  const int patches = a.grid_x * a.grid_y;
  constexpr int CPR = N / 16;  // 16-pixel chunks per row
  const int lrow = wave0 * LPW + ord1<N>(lane0 / CPR), lcol = (lane0 % CPR) * 16;  // this lane's 16 pixels of the patch (ord1: pc_passes.hpp)
  const bool ld_on = lane0 < LPW * CPR;  // (all lanes unless the workgroup has more than N*N/16 threads)

  // N >= 128: persistent workgroups (one per CU) walking a linear patch index. Smaller tiles: one workgroup per
  // patch on a 3-D grid (patch column, patch row, pair) -- the hardware dispatcher balances 4 workgroups per CU better
  // than a static stride does, and the patch coordinates come from blockIdx without the integer divisions that
  // otherwise cost ~5 % of the kernel's VALU instructions.
  constexpr bool PERSIST = P::PERSIST;
  constexpr bool FWD3 = MOF_K1_FWD3 && N == 64 && DS == 1 && !P::PERSIST && MOF_RAW_STAGE && T == 256;  // pc_passes3.hpp
  constexpr bool RAW = MOF_RAW_STAGE && (N == 128 || N == 64) && DS == 1 && T == N * N / 16;  // raw pixel staging (pc_passes.hpp)  // raw pixel staging (pc_passes.hpp)
  (void)patches;
  // (x0, y0) of a patch are in the units of the correlated image: full-res pixels, or quarter-res when DS = 4
  auto patch_base = [&](int p, const uint8_t* frames, size_t frame_stride) -> const uint8_t* {
    int pr, bx, by;
    if constexpr (PERSIST) {
      const int pt = p % patches;
      pr = p / patches;
      bx = pt % a.grid_x;
      by = pt / a.grid_x;
    } else {
      pr = blockIdx.z;
      bx = blockIdx.x;
      by = blockIdx.y;
    }
    const int px0 = a.origin_x + bx * a.stride_x, py0 = a.origin_y + by * a.stride_y;
    return frames + (size_t)pr * frame_stride + (size_t)(DS * py0) * a.pitch + (size_t)(CH * DS * px0);
  };

  // twiddles of the second Stockham stage, W_N^{k x}: x = lane % R1 in row passes, lane / (64/R1) in column passes
  cf tw_row[R2 - 1], tw_col[R2 - 1];
  {
    const int xr = lane0 % R1, xc = lane0 / (64 / R1);
#pragma unroll
    for (int k = 1; k < R2; ++k) {
      tw_row[k - 1] = {a.twiddles[2 * (k * xr)], a.twiddles[2 * (k * xr) + 1]};
      tw_col[k - 1] = {a.twiddles[2 * (k * xc)], a.twiddles[2 * (k * xc) + 1]};
    }
  }

  Fwd3Tw tw3;
  if constexpr (FWD3) tw3.load(a.twiddles, wave0, lane0);
  // ---- persistent workgroup: patches p = blockIdx.x, + gridDim.x, ...; the 2 x 16 B of the NEXT patch are
  //      requested from HBM before the current one is transformed, so the ~2 us load latency is off the critical path
  uint32_t cw[4] = {0u, 0u, 0u, 0u}, pw[4] = {0u, 0u, 0u, 0u};
  int p = PERSIST ? (int)blockIdx.x : (int)((blockIdx.z * gridDim.y + blockIdx.y) * gridDim.x + blockIdx.x);
  auto fetch16 = [&](int pp, uint32_t* c, uint32_t* q) {
    const uint8_t* cs = patch_base(pp, a.cur, a.cur_stride) + (size_t)lrow * a.pitch + CH * lcol;
    const uint8_t* ps = patch_base(pp, a.prev, a.prev_stride) + (size_t)lrow * a.pitch + CH * lcol;
    if constexpr (CH == 1) {
      __builtin_memcpy(c, cs, 16);
      __builtin_memcpy(q, ps, 16);
    } else {
      gray16_from_bgr48(cs, c);
      gray16_from_bgr48(ps, q);
    }
  };
  if (DS == 1 && p < a.total && ld_on) fetch16(p, cw, pw);
  // Co-resident workgroups would otherwise run the same phase at the same time (all of them LDS-bound, then all
  // VALU-bound): delay the k-th workgroup of a CU by k quarter-patches so their phases interleave.
  if constexpr (PERSIST) {
    const int slot = (blockIdx.x / a.stagger_div) & 3;
    for (int i = 0; i < slot * a.stagger_units; ++i) __builtin_amdgcn_s_sleep(100);
  }
  for (; p < a.total; p += PERSIST ? (int)gridDim.x : a.total) {
  // The lane / wave indices are laundered once per patch: otherwise LICM hoists every LDS address of the body out
  // of the persistent loop (+70 VGPRs), which costs a whole workgroup of occupancy per CU.
  int lane = lane0, wave = wave0;
  asm volatile("" : "+v"(lane), "+v"(wave));
  lane &= 63;             // give the value ranges back to the optimiser (they fold the skew and the
  wave &= P::WAVES - 1;   // Hermitian row cases at compile time)
  const int tid = wave * 64 + lane;
#ifdef MOF_EXTRA_VALU  // diagnostic build: MOF_EXTRA_VALU independent FMAs per wave and patch (is K1 VALU-issue bound?)
  {
    float d0 = (float)lane, d1 = 1.f, d2 = 2.f, d3 = 3.f;
#pragma unroll
    for (int i = 0; i < MOF_EXTRA_VALU / 4; ++i)
      asm volatile("v_fma_f32 %0, %0, %0, %0\n\tv_fma_f32 %1, %1, %1, %1\n\tv_fma_f32 %2, %2, %2, %2\n\tv_fma_f32 %3, %3, %3, %3"
                   : "+v"(d0), "+v"(d1), "+v"(d2), "+v"(d3));
    if (d0 + d1 + d2 + d3 == 12345.678f) a.out[0] = d0;
  }
#endif
  // ---- the wave's own LPW rows: u8 -> f32 (exact), z = cur + i*prev  (convertTo, :1805-1806)
  {
    const int row = wave * LPW + ord1<N>(lane / CPR), col = (lane % CPR) * 16;
    int cc = 256, cp = 256;  // constant-patch codes of this wave (pc_common.hpp)
    if constexpr (DS == 1) {
      // pre-test: a textured patch has a lane whose first two dwords differ -- two compares and it is out
      const bool maybe_c = __builtin_amdgcn_ballot_w64(ld_on && cw[0] != cw[1]) == 0ull;
      const bool maybe_p = __builtin_amdgcn_ballot_w64(ld_on && pw[0] != pw[1]) == 0ull;
      // (the empty asm keeps the compiler from if-converting the rare branch into unconditional instructions)
      if (__builtin_expect(maybe_c, 0)) {
        asm volatile("");
        uint32_t f, d;
        const_track(cw, 4, true, f, d);
        cc = wave_const_code(ld_on, f, d);
      }
      if (__builtin_expect(maybe_p, 0)) {
        asm volatile("");
        uint32_t f, d;
        const_track(pw, 4, true, f, d);
        cp = wave_const_code(ld_on, f, d);
      }
    }
    uint32_t fc = 0, dc = 0, fp = 0, dp = 0;  // (long-range mode forms its pixels one by one: tracked in its loop)
    if constexpr (RAW) {
      // raw bytes into the wave's staging area; the first row stage converts them (pc_passes.hpp, raw_store)
      raw_store<N>(z, wave * LPW, lane, cw, pw);
      if constexpr (PERSIST) {
        const int pn = p + gridDim.x;
        if (pn < a.total) fetch16(pn, cw, pw);
      }
    } else if constexpr (DS == 1) {
      if (ld_on) {
#pragma unroll
        for (int q = 0; q < 4; ++q)
#pragma unroll
          for (int b = 0; b < 4; ++b) {
#ifdef MOF_ABLATE_NOLOADW  // diagnostic build: one tile store per lane instead of 16 (results wrong by design)
            if (q + b) continue;
#endif
            z[zaddr<N>(row, col + q * 4 + b)] = {(float)((cw[q] >> (8 * b)) & 0xffu), (float)((pw[q] >> (8 * b)) & 0xffu)};
          }
        if constexpr (PERSIST) {
          const int pn = p + gridDim.x;
          if (pn < a.total) fetch16(pn, cw, pw);
        }
      }
    } else if (ld_on) {
      // pixel (row, col+i) <- (s(4r+1,4c+1) + s(4r+1,4c+2) + s(4r+2,4c+1) + s(4r+2,4c+2) + 2) >> 2 of the full-res frame
      const uint8_t* c1 = patch_base(p, a.cur, a.cur_stride) + (size_t)(4 * row + 1) * a.pitch + 4 * col;
      const uint8_t* p1 = patch_base(p, a.prev, a.prev_stride) + (size_t)(4 * row + 1) * a.pitch + 4 * col;
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        uint32_t ca[4], cb[4], pa[4], pb[4];
        __builtin_memcpy(ca, c1 + 16 * q, 16);
        __builtin_memcpy(cb, c1 + a.pitch + 16 * q, 16);
        __builtin_memcpy(pa, p1 + 16 * q, 16);
        __builtin_memcpy(pb, p1 + a.pitch + 16 * q, 16);
#pragma unroll
        for (int b = 0; b < 4; ++b) {
          const uint32_t cs = ((ca[b] >> 8) & 0xffu) + ((ca[b] >> 16) & 0xffu) + ((cb[b] >> 8) & 0xffu) + ((cb[b] >> 16) & 0xffu);
          const uint32_t ps = ((pa[b] >> 8) & 0xffu) + ((pa[b] >> 16) & 0xffu) + ((pb[b] >> 8) & 0xffu) + ((pb[b] >> 16) & 0xffu);
          const uint32_t cv = (cs + 2u) >> 2, pv = (ps + 2u) >> 2;
          if (q == 0 && b == 0) fc = cv, fp = pv, dc = 0u, dp = 0u;
          dc |= cv ^ fc;
          dp |= pv ^ fp;
          z[zaddr<N>(row, col + q * 4 + b)] = {(float)cv, (float)pv};
        }
      }
    }
    if constexpr (DS != 1) {
      cc = wave_const_code(ld_on, fc, dc);
      cp = wave_const_code(ld_on, fp, dp);
    }
    if (lane == 0) const_code[wave] = cc | (cp << 16);
    wave_sync();
  }

  // ---- forward 2-D transform of z: rows (wave-local), barrier, columns (wave-local)  (dft x2, :1491-1493)
#ifndef MOF_ABLATE_NOFWD
  if constexpr (FWD3) {
#if MOF_K1_MFMA_S1
    fwd3_rows_mfma(z, wave, lane, a.twiddles);
#else
    fwd3_rows(z, wave, lane);
#endif
    __syncthreads();
    fwd3_mid(z, wave, lane, tw3);
    wave_sync();
    fwd3_cols(z, wave, lane);
  } else {
    row_pass<N, LPW, RAW>(z, wave * LPW, lane, tw_row);
    __syncthreads();
    col_pass_fwd<N>(z, wave * LPW, lane, tw_col);
  }
#endif
  __syncthreads();

  // ---- untangle A = FFT(cur), B = FFT(prev); P = A conj(B); C = P|P| / (|P|^2 + eps)
  //      (mulSpectrums :1494, magSpectrums :70-168, divSpectrums :1086-1251, incl. the real-only-slot
  //      behaviour of :107-109 / :1127-1129). Only the half spectrum v < N/2 (+ row N/2 packed into the
  //      imaginary part of row 0) is kept, conjugated, for the Hermitian inverse.
  if constexpr (P::FUSE_XPOW) {
  // rows 0 and H share row 0: G'[u] = conj(C[0][u]) + i conj(C[H][u]) (partner of u is N-u in the same rows), packed
  // by wave 0, which owns row 0 in the pass below; rows 1..H-1 get their cross-power inside row_pass_xpow
  if (wave == 0) {
    for (int u = lane; u <= H; u += 64) {
      const int um = (N - u) % N;
      const bool self = (u == um);
      const cf C0 = cross_power<PK>(z[zaddr<N>(0, u)], z[zaddr<N>(0, um)], self);
      const cf Ch = cross_power<PK>(z[zaddr<N>(H, u)], z[zaddr<N>(H, um)], self);
      if (u == 0) *reinterpret_cast<float*>(const_code + 16) = C0.x;  // C_dc: all that is left of a degenerate pair's spectrum
      z[zaddr<N>(0, u)] = {C0.x + Ch.y, Ch.x - C0.y};
      if (!self) z[zaddr<N>(0, um)] = {C0.x - Ch.y, Ch.x + C0.y};
    }
    wave_sync();
  }
  // ---- inverse (unscaled) of the Hermitian spectrum: forward transforms of conj(C); rows 0..H-1 only,
  //      then column pairs  (idft :1497)
#if MOF_XPOW_SPLIT
  if (wave == 0) row_pass_xpow<N, PK, true>(z, 0, lane, tw_row);
  else if (wave < P::WI) row_pass_xpow<N, PK, false>(z, wave * P::LI, lane, tw_row);
#else
  if (wave < P::WI) row_pass_xpow<N, PK, true>(z, wave * P::LI, lane, tw_row);
#endif
  } else {
#ifndef MOF_ABLATE_NOPW
  {
    // rows 1..H-1: every u; partner (N-v, N-u) lies in the untouched lower half. Fixed trip count, so every
    // address is a base plus a compile-time offset.
    {
      constexpr int UPW = (N < 64) ? N : 64;          // columns covered by one wave
      constexpr int RPI = T / UPW;                     // rows covered per iteration
      const int u = tid % UPW, vr = tid / UPW;
      const int um = (N - u) % N;
#pragma unroll
      for (int i = 0; i < (H - 1 + RPI - 1) / RPI; ++i) {
#pragma unroll
        for (int uu = 0; uu < N / UPW; ++uu) {
          const int v = 1 + vr + i * RPI;
          if (v < H) {
            const int uc = u + uu * UPW, umc = (N - uc) % N;
            (void)um;
            const cf zk = z[zaddr<N>(v, uc)], zm = z[zaddr<N>(N - v, umc)];
            const cf C = cross_power<PK>(zk, zm, false);
            z[zaddr<N>(v, uc)] = {C.x, -C.y};  // conj(C[v][u])
          }
        }
      }
    }
    // rows 0 and H share row 0: G'[u] = conj(C[0][u]) + i conj(C[H][u]); partner of u is N-u in the same rows
    for (int u = tid; u <= H; u += T) {
      const int um = (N - u) % N;
      const bool self = (u == um);
      const cf C0 = cross_power<PK>(z[zaddr<N>(0, u)], z[zaddr<N>(0, um)], self);
      const cf Ch = cross_power<PK>(z[zaddr<N>(H, u)], z[zaddr<N>(H, um)], self);
      if (u == 0) *reinterpret_cast<float*>(const_code + 16) = C0.x;  // C_dc (see the fused form above)
      z[zaddr<N>(0, u)] = {C0.x + Ch.y, Ch.x - C0.y};
      if (!self) z[zaddr<N>(0, um)] = {C0.x - Ch.y, Ch.x + C0.y};
    }
  }
#endif
  __syncthreads();

  // ---- inverse (unscaled) of the Hermitian spectrum: forward transforms of conj(C); rows 0..H-1 only,
  //      then column pairs  (idft :1497)
#ifndef MOF_ABLATE_NOINV
  if (wave < P::WI) row_pass<N, P::LI>(z, wave * P::LI, lane, tw_row);
#endif
  }  // FUSE_XPOW
  __syncthreads();
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
#ifndef MOF_ABLATE_NOINV
  if (wave < P::WI) best = col_pass_inv<N, PK>(z, wave * P::LI, lane, tw_col, a.search_radius);
#else
  best = Best{z[zaddr<N>(lane, wave)].x, lane * N + wave};
#endif
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();

  // ---- 5x5 weighted centroid in double + validity gate  (:1337-1383, :1838-1856), wave 0. The window is read
  //      before a second barrier releases the other waves to overwrite the tile with the next patch.
  float wval = 0.f;
  bool degenerate = false;
#ifdef MOF_ABLATE_NOTAIL
  if (wave == 0 && lane == 0) { a.out[2 * (size_t)p] = best.v; a.out[2 * (size_t)p + 1] = (double)best.idx; }
  continue;
#endif
  if (wave == 0) {
    const int my_code = const_code[lane & 15];
    for (int w = 1; w < P::WAVES; ++w) best = better(best, red[w]);
    wval = centroid_window_value<N, PK>(best, lane, [&](int ys, int xs) {
      const int y = (ys + H) % N, x = (xs + H) % N;  // un-shifted position
      const cf s = z[zaddr<N>(y, x % H)];
      return x < H ? s.x : s.y;
    });
    degenerate = const_codes_degenerate<P::WAVES>(my_code, lane);
  }
  __syncthreads();  // (needed by the persistent form only; dropping it for one-workgroup-per-patch sizes measured -1 %)
  if (wave == 0)
    centroid_gate_store<N, PK>(best, wval, lane, a.max_px_speed_sq, a.out + 2 * (size_t)p, degenerate,
                               degenerate ? *reinterpret_cast<const float*>(const_code + 16) : 0.f);
  }  // persistent loop
}

// Diagnostic knob (occupancy experiments only): MOF_PC_EXTRA_LDS=<bytes> pads the dynamic LDS request.
static size_t extra_lds() {
  static const size_t v = [] {
    const char* e = getenv("MOF_PC_EXTRA_LDS");
    return e ? (size_t)atol(e) : (size_t)0;
  }();
  return v;
}

template <int N, int DS, int CH, int PK>
static hipError_t configure_one(int lds) {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_field_kernel<N, DS, CH, PK>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

template <int N>
static hipError_t configure_n() {
  const int lds = (int)(PcTraits<N>::LDS_BYTES + extra_lds());
  hipError_t e;
  if ((e = configure_one<N, 1, 1, 0>(lds)) != hipSuccess) return e;
  if ((e = configure_one<N, 1, 3, 0>(lds)) != hipSuccess) return e;
  if ((e = configure_one<N, 4, 1, 0>(lds)) != hipSuccess) return e;
  if ((e = configure_one<N, 1, 1, 1>(lds)) != hipSuccess) return e;
  if ((e = configure_one<N, 1, 3, 1>(lds)) != hipSuccess) return e;
  return configure_one<N, 4, 1, 1>(lds);
}

static int g_cu_count = 0;  // set by pc_configure()

template <int N>
static hipError_t launch_n(const PcArgs& a_in, int n_pairs, hipStream_t stream) {
  using Tr = PcTraits<N>;
  PcArgs a = a_in;
  a.total = n_pairs * a.grid_x * a.grid_y;
  a.stagger_div = g_cu_count > 0 ? g_cu_count : 256;
  {
    static const int units = [] { const char* e = getenv("MOF_PC_STAGGER"); return e ? atoi(e) : 0; }();
    a.stagger_units = units;
  }
  if (a.downscale == 4 && a.channels == 3) return hipErrorInvalidValue;
  const dim3 b(Tr::T);
  const size_t lds = Tr::LDS_BYTES + extra_lds();
  auto launch = [&](const PcArgs& aa, dim3 g) {
    if (aa.peak_model == 1) {
      if (aa.downscale == 4) hipLaunchKernelGGL((pc_field_kernel<N, 4, 1, 1>), g, b, lds, stream, aa);
      else if (aa.channels == 3) hipLaunchKernelGGL((pc_field_kernel<N, 1, 3, 1>), g, b, lds, stream, aa);
      else hipLaunchKernelGGL((pc_field_kernel<N, 1, 1, 1>), g, b, lds, stream, aa);
    } else {
      if (aa.downscale == 4) hipLaunchKernelGGL((pc_field_kernel<N, 4, 1, 0>), g, b, lds, stream, aa);
      else if (aa.channels == 3) hipLaunchKernelGGL((pc_field_kernel<N, 1, 3, 0>), g, b, lds, stream, aa);
      else hipLaunchKernelGGL((pc_field_kernel<N, 1, 1, 0>), g, b, lds, stream, aa);
    }
  };
  if constexpr (Tr::PERSIST) {
    // as many workgroups as are resident at once (LDS-limited), each loops over patches (c4: +9 % over one per patch)
    const int per_cu = (int)((160u * 1024u) / (Tr::LDS_BYTES + extra_lds()));
    const int resident = (g_cu_count > 0 ? g_cu_count : 256) * (per_cu < 1 ? 1 : (per_cu > 16 ? 16 : per_cu));
    launch(a, dim3((unsigned)(a.total < resident ? a.total : resident)));
  } else {
    // one workgroup per patch; the pair index rides gridDim.z (at most 65535 per launch)
    const int patches = a.grid_x * a.grid_y;
    for (int k0 = 0; k0 < n_pairs; k0 += 65535) {
      const int nk = n_pairs - k0 < 65535 ? n_pairs - k0 : 65535;
      PcArgs c = a;
      c.cur = a.cur + (size_t)k0 * a.cur_stride;
      c.prev = a.prev + (size_t)k0 * a.prev_stride;
      c.out = a.out + (size_t)k0 * patches * 2;
      c.total = nk * patches;
      launch(c, dim3((unsigned)a.grid_x, (unsigned)a.grid_y, (unsigned)nk));
    }
  }
  return hipGetLastError();
}

// The quad-per-line formulation of pc_kernel_quad.hip (an evaluated alternative: a third of the LDS traffic, 35 % more
// VALU instructions, 8 % slower on MI355X -- DESIGN.md section 4, K1) is NOT part of the product library: only the A/B
// build `make quad` (-DMOF_WITH_QUAD -> csrc/ab/libmof_hip_quad.so, loaded through MOF_LIB_PATH) links it, and there
// MOF_PC_QUAD=1 routes N = 64 to it.
#ifdef MOF_WITH_QUAD
static bool classic64() {
  static const bool v = [] { const char* e = getenv("MOF_PC_QUAD"); return !(e && atoi(e) != 0); }();
  return v;
}
#endif

void pc_mfma_s1_fragments(uint32_t* out) {
  // step s = (hi | lo, n1 half ks); lane l = (row r = l % 32 = (k1, re | im), h = l / 32); slot j = 2 q + (cur | prev), n1 = 8 ks + 4 h + q
  for (int s = 0; s < 4; ++s)
    for (int l = 0; l < 64; ++l)
      for (int j = 0; j < 8; ++j) {
        const int ks = s & 1, lo = s >> 1, r = l & 31, h = l >> 5, q = j >> 1, c = j & 1;
        const int k1 = r >> 1, part = r & 1, n1 = 8 * ks + 4 * h + q, idx = (k1 * n1) & 15;
        double cs, sn;
        unit_root(idx, 16, &cs, &sn);
        const float w = (float)(part == 0 ? (c == 0 ? cs : sn) : (c == 0 ? -sn : cs));
        uint16_t hi, lw;
        f16_split(w, &hi, &lw);
        const uint16_t v = lo ? lw : hi;
        uint32_t& d = out[(64 * s + l) * 4 + (j >> 1)];
        d = (j & 1) ? (d & 0xffffu) | ((uint32_t)v << 16) : (d & 0xffff0000u) | v;
      }
}

hipError_t pc_configure(int patch_size) {
  int dev = 0;
  hipDeviceProp_t prop;
  if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) g_cu_count = prop.multiProcessorCount;
  switch (patch_size) {
    case 32: return configure_n<32>();
    case 64: {
      hipError_t e = configure_n<64>();
#ifdef MOF_WITH_QUAD
      if (e == hipSuccess) e = pc_configure_quad64();
#endif
      return e;
    }
    case 128: return configure_n<128>();
    case 120: return pc_configure_120();
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_pc_field(const PcArgs& a, int patch_size, int n_pairs, hipStream_t stream) {
  switch (patch_size) {
    case 32: return launch_n<32>(a, n_pairs, stream);
    case 64:
#ifdef MOF_WITH_QUAD
      if (!classic64()) return launch_pc_field_quad64(a, n_pairs, stream);
#endif
      return launch_n<64>(a, n_pairs, stream);
    case 128: return launch_n<128>(a, n_pairs, stream);
    case 120: return launch_pc_field_120(a, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}

const char* pc_kernel_variant(int patch_size) {
#ifdef MOF_WITH_QUAD
  if (patch_size == 64 && !classic64()) return "quad";
#endif
  return "stockham";
}

bool pc_patch_size_supported(int n) { return n == 32 || n == 64 || n == 128 || n == 120; }

}  // namespace mof
