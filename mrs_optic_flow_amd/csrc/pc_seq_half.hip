// pc_seq_half.hip -- the sequence form of K1 (see pc_seq_kernel.hip) on a HALF-SIZE tile, for 128 x 128 patches.
//
// The real formulation of pc_seq_kernel.hip never needs more than N * N / 2 complex values at a time (N/2 packed lines of N,
// then the N x N/2 half spectrum, then N/2 row pairs of N). At N = 128 that is the difference between one workgroup per CU
// (the pair kernel's 128 x 136 tile = 139 KB) and TWO (64 x 152 = 78 KB each, 512 lanes each): the second workgroup runs
// while the first one waits at a barrier. The tile is laid out so that each phase works in place:
//   physical row j (N/2 rows)  =  complex line j = patch rows (2j, 2j+1), N elements           [load, row transforms]
//                              =  logical spectrum rows 2j | 2j+1, N/2 bins each (u = 0..N/2-1)  [after the wave-local untangle]
//   logical (r, u)             ->  physical (r >> 1, u + (N/2)(r & 1))                            [column passes along r]
//   row pair (y, y + N/2)      ->  result (c[y][x], c[y + N/2][x]), x < N/2 over logical row y, x >= N/2 over row y + N/2
// Columns are skewed as in K1 (c + (c >> SK)) with a gap before the right half; pitch and gap come from the bank model
// (tools/design/lds_conflicts_seq.py: 1.41x the conflict-free cycles, the pair kernel's N = 128 layout has 1.44x).
// Per frame: rows (wave-local) -> barrier -> columns forward, cross-power against the previous frame's spectrum held in
// registers (16 complex bins per lane), columns inverse -> barrier -> row pairs + arg-max -> barrier -> centroid -> barrier.
// R1 = 16 != R2 = 8, so unlike N = 64 the cross-power result passes through LDS once on its way into the inverse transform.
// Template on N: the N = 64 instantiation exists for A/B against pc_seq_kernel.hip's full-tile form (MOF_FFT_SEQ_HALF64=1).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"

namespace mof {

namespace {

template <int N>
struct HalfCfg;
template <>
struct HalfCfg<64> {
  static constexpr int R1 = 8, R2 = 8, SK = 3, GAP = 4, PITCH = 80, WAVES = 4;
};
template <>
struct HalfCfg<128> {
  static constexpr int R1 = 16, R2 = 8, SK = 4, GAP = 8, PITCH = 152, WAVES = 8;
};

template <int N>
struct HalfTile {
  using C = HalfCfg<N>;
  static constexpr int H = N / 2, T = 64 * C::WAVES, R1 = C::R1, R2 = C::R2;
  static constexpr size_t LDS_BYTES = sizeof(cf) * (size_t)H * C::PITCH + 64 * sizeof(Best);
  static_assert(R1 * R2 == N && R2 == 8 && H / 8 == C::WAVES, "eight lines, columns and row pairs per wave");
  static __device__ __forceinline__ int pcol(int c) { return c + (c >> C::SK) + (c >= H ? C::GAP : 0); }
  static __device__ __forceinline__ int line(int j, int x) { return j * C::PITCH + pcol(x); }
  static __device__ __forceinline__ int spec(int r, int u) { return (r >> 1) * C::PITCH + pcol(u + H * (r & 1)); }
  static __device__ __forceinline__ int out(int y1, int c) { return c < H ? spec(y1, c) : spec(y1 + H, c - H); }
};

// Second-stage twiddles W_N^{k x}, k = 1..7. N = 64: held in registers for the whole kernel (as K1). N = 128: fetched from
// the L1-resident table right before each second stage -- the 28 VGPRs of two resident sets do not fit beside the 32 of the
// previous frame's spectrum at four waves per SIMD (72 spilled registers otherwise).
template <int N>
struct HalfTw {
  static constexpr bool IN_REGS = N == 64;
  cf row[IN_REGS ? 7 : 1], col[IN_REGS ? 7 : 1];
  const float* table;
  int xr, xc;
  __device__ __forceinline__ void init(const float* t, int lane) {
    table = t;
    xr = lane % HalfCfg<N>::R1;
    xc = lane / (64 / HalfCfg<N>::R1);
    if constexpr (IN_REGS) {
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        row[k - 1] = {t[2 * (k * xr)], t[2 * (k * xr) + 1]};
        col[k - 1] = {t[2 * (k * xc)], t[2 * (k * xc) + 1]};
      }
    }
  }
  __device__ __forceinline__ void get(bool is_row, cf* out) const {
    if constexpr (IN_REGS) {
#pragma unroll
      for (int k = 0; k < 7; ++k) out[k] = is_row ? row[k] : col[k];
    } else {
      const int x = is_row ? xr : xc;
#pragma unroll
      for (int k = 1; k < 8; ++k) {
        const float2 t = *reinterpret_cast<const float2*>(table + 2 * (k * x));
        out[k - 1] = {t.x, t.y};
      }
    }
  }
};

// ---- raw pixel staging (N = 128; as pc_passes.hpp, raw_store): the lane's 16 + 16 pixels (patch rows 2j, 2j + 1) leave as
// two ds_write_b128 of interleaved bytes into a per-wave area over the wave's own lines; the first row stage converts.
#ifndef MOF_RAW_STAGE
#define MOF_RAW_STAGE 1
#endif
constexpr int HALF_RAW_PITCH = 272;
template <int N>
__device__ __forceinline__ unsigned char* half_raw_area(cf* z, int line0) {
  return reinterpret_cast<unsigned char*>(z + HalfTile<N>::line(line0, 0));
}
template <int N>
__device__ __forceinline__ void half_raw_store(cf* z, int line0, int lane, const uint32_t* ra, const uint32_t* rb) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef u4 __attribute__((address_space(3))) * lds_u4_ptr;
  unsigned char* dst = half_raw_area<N>(z, line0) + (lane >> 3) * HALF_RAW_PITCH + (lane & 7) * 32;
  u4 lo, hi;
  lo.x = __builtin_amdgcn_perm(rb[0], ra[0], 0x05010400u);
  lo.y = __builtin_amdgcn_perm(rb[0], ra[0], 0x07030602u);
  lo.z = __builtin_amdgcn_perm(rb[1], ra[1], 0x05010400u);
  lo.w = __builtin_amdgcn_perm(rb[1], ra[1], 0x07030602u);
  hi.x = __builtin_amdgcn_perm(rb[2], ra[2], 0x05010400u);
  hi.y = __builtin_amdgcn_perm(rb[2], ra[2], 0x07030602u);
  hi.z = __builtin_amdgcn_perm(rb[3], ra[3], 0x05010400u);
  hi.w = __builtin_amdgcn_perm(rb[3], ra[3], 0x07030602u);
  *(lds_u4_ptr)(dst) = lo;
  *(lds_u4_ptr)(dst + 16) = hi;
}

// ---- transforms along x of the wave's 8 lines, then the wave-local untangle into the two half-spectrum rows per line ----
template <int N, bool RAW = false>
__device__ __forceinline__ void half_rows(cf* __restrict__ z, int line0, int lane, const HalfTw<N>& tw) {
  using L = HalfTile<N>;
  constexpr int R1 = L::R1, R2 = L::R2, H = L::H;
  {
    const int ln = line0 + (lane >> 3), x = lane & 7;
    cf v[R1];
    if constexpr (RAW) {
      typedef const volatile uint16_t __attribute__((address_space(3))) * lds_u16_ptr;
      const unsigned char* src = half_raw_area<N>(z, line0) + (lane >> 3) * HALF_RAW_PITCH + 2 * x;
#pragma unroll
      for (int k = 0; k < R1; ++k) {
        const uint32_t ab = *(lds_u16_ptr)(src + 2 * k * R2);
        v[k] = {(float)(ab & 0xffu), (float)(ab >> 8)};
      }
    } else {
#pragma unroll
      for (int k = 0; k < R1; ++k) v[k] = lds_read(&z[L::line(ln, x + k * R2)]);
    }
    butterfly<R1>(v);
    wave_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) z[L::line(ln, x * R1 + k)] = v[k];
    wave_sync();
  }
  {
    constexpr int PER = 8 * R1 / 64;
    cf v[PER][R2], tw_row[7];
    tw.get(true, tw_row);
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b, ln = line0 + q / R1, x = q % R1;
#pragma unroll
      for (int k = 0; k < R2; ++k) v[b][k] = lds_read(&z[L::line(ln, x + k * R1)]);
      butterfly8_tw(v[b], tw_row);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b, ln = line0 + q / R1, x = q % R1;
#pragma unroll
      for (int k = 0; k < R2; ++k) z[L::line(ln, x + k * R1)] = v[b][k];
    }
    wave_sync();
  }
  {
    // line j = rows 2j + i (2j+1): R_2j[u] = (Z[u] + conj Z[N-u]) / 2, R_2j+1[u] = (Z[u] - conj Z[N-u]) / 2i, kept DOUBLED;
    // the real bins u = 0 and u = N/2 of a row share its column 0
    constexpr int M = H / 8;
    const int lr = line0 + (lane >> 3), ug = lane & 7;
    cf zk[M], zm[M], zh = {0.f, 0.f};
#pragma unroll
    for (int m = 0; m < M; ++m) {
      const int u = ug + 8 * m;
      zk[m] = lds_read(&z[L::line(lr, u)]);
      zm[m] = lds_read(&z[L::line(lr, (N - u) & (N - 1))]);
    }
    if (ug == 0) zh = lds_read(&z[L::line(lr, H)]);
    wave_sync();
#pragma unroll
    for (int m = 0; m < M; ++m) {
      cf A, B;
      untangle2(zk[m], zm[m], &A, &B);
      if (m == 0 && ug == 0) {
        z[L::line(lr, 0)] = {A.x, 2.f * zh.x};
        z[L::line(lr, H)] = {B.x, 2.f * zh.y};
      } else {
        z[L::line(lr, ug + 8 * m)] = A;
        z[L::line(lr, ug + 8 * m + H)] = B;
      }
    }
  }
}

// ---- the wave's 8 columns: forward along y, cross-power against `prev`, inverse along y ------------------------------
// Where the previous image's column spectra wait between the two images of a PAIR (pc_pair_half_kernel): in the sequence kernel they
// stay in the registers `prev`; the pair kernel parks them in a per-workgroup slab of global memory (L2 / Infinity Cache resident:
// written and re-read by the same lanes a few microseconds apart), so that the 32 registers are free while the current image's
// rows and columns are transformed -- what lets the 128 x 128 form fit 128 VGPRs = two workgroups per CU.
struct PrevInRegs {
  static constexpr bool parked = false;
  __device__ __forceinline__ void store(int, cf) const {}
  __device__ __forceinline__ cf load(int) const { return cf{0.f, 0.f}; }
};
struct PrevInSlab {
  static constexpr bool parked = true;
  float2* slab;  // [elements per lane][threads], this lane's column
  int threads;
  __device__ __forceinline__ void store(int i, cf v) const { slab[(size_t)i * threads] = make_float2(v.x, v.y); }
  __device__ __forceinline__ cf load(int i) const {
    const float2 t = slab[(size_t)i * threads];
    return cf{t.x, t.y};
  }
};

template <int N, int PK, class Park = PrevInRegs>
__device__ __forceinline__ void half_cols(cf* __restrict__ z, int col0, int lane, const HalfTw<N>& tw, cf (*prev)[8], cf* prev0,
                                          cf* prevH, bool prime, bool has_col0, Park park = Park{}) {
  using L = HalfTile<N>;
  constexpr int R1 = L::R1, R2 = L::R2, H = L::H, PER = 8 * R1 / 64, CW = 64 / R1;
  auto stage1 = [&](bool load) {
    const int col = col0 + (lane & 7), x = lane >> 3;
    cf v[R1];
    (void)load;
#pragma unroll
    for (int k = 0; k < R1; ++k) v[k] = lds_read(&z[L::spec(x + k * R2, col)]);
    butterfly<R1>(v);
    wave_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) z[L::spec(x * R1 + k, col)] = v[k];
    wave_sync();
  };
  stage1(true);
  cf v[PER][R2];
  {
    cf tw_col[7];
    tw.get(false, tw_col);
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = lane / CW;
#pragma unroll
      for (int k = 0; k < R2; ++k) v[b][k] = lds_read(&z[L::spec(x + k * R1, col)]);
      butterfly8_tw(v[b], tw_col);  // v[b][k] = 2 F[x + k R1][col]
    }
  }
  const bool holder = has_col0 && (lane % CW) == 0;  // b = 0 of these lanes is column 0
  if (has_col0) {
    // column 0 = G[v] = F[v][0] + i F[v][H] (two real columns): apart with the partner bin N - v, through LDS
    wave_sync();
    if (holder) {
      const int x = lane / CW;
#pragma unroll
      for (int k = 0; k < R2; ++k) z[L::spec(x + k * R1, 0)] = v[0][k];
    }
    wave_sync();
    cf C0[(H + 64) / 64], Ch[(H + 64) / 64];
#pragma unroll
    for (int i = 0; i < (H + 64) / 64; ++i) {
      const int vv = lane + 64 * i, vm = (N - vv) & (N - 1);
      C0[i] = {0.f, 0.f};
      Ch[i] = {0.f, 0.f};
      if (vv <= H) {
        const bool self = vv == 0 || vv == H;
        cf f0, fh;
        untangle2(lds_read(&z[L::spec(vv, 0)]), lds_read(&z[L::spec(vm, 0)]), &f0, &fh);
        f0 = {0.5f * f0.x, 0.5f * f0.y};
        fh = {0.5f * fh.x, 0.5f * fh.y};
        if (!prime) {
          C0[i] = cross_power_ab<PK>(f0, prev0[i], self);
          Ch[i] = cross_power_ab<PK>(fh, prevH[i], self);
        }
        prev0[i] = f0;
        prevH[i] = fh;
      }
    }
    wave_sync();
    if (!prime) {
#pragma unroll
      for (int i = 0; i < (H + 64) / 64; ++i) {
        const int vv = lane + 64 * i, vm = (N - vv) & (N - 1);
        if (vv <= H) {
          const bool self = vv == 0 || vv == H;
          z[L::spec(vv, 0)] = {C0[i].x + Ch[i].y, Ch[i].x - C0[i].y};
          if (!self) z[L::spec(vm, 0)] = {C0[i].x - Ch[i].y, Ch[i].x + C0[i].y};
        }
      }
    }
    wave_sync();
  }
  if (prime) {
#pragma unroll
    for (int b = 0; b < PER; ++b)
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        if constexpr (Park::parked) park.store(b * R2 + k, v[b][k]);
        else prev[b][k] = v[b][k];
      }
    return;
  }
  if constexpr (Park::parked) {
    cf pv[PER][R2];
#pragma unroll
    for (int b = 0; b < PER; ++b)
#pragma unroll
      for (int k = 0; k < R2; ++k) pv[b][k] = park.load(b * R2 + k);
#pragma unroll
    for (int b = 0; b < PER; ++b)
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        const cf C = cross_power_ab<PK>(v[b][k], pv[b][k], false);
        v[b][k] = {C.x, -C.y};
      }
  } else {
#pragma unroll
    for (int b = 0; b < PER; ++b)
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        const cf C = cross_power_ab<PK>(v[b][k], prev[b][k], false);
        prev[b][k] = v[b][k];
        v[b][k] = {C.x, -C.y};
      }
  }
  if constexpr (R1 == R2) {
    // 64 = 8 x 8: the second-stage output distribution IS the first-stage input distribution: registers go straight on
    const int col = col0 + (lane & 7), x = lane >> 3;
    if (holder) {
#pragma unroll
      for (int k = 0; k < R2; ++k) v[0][k] = lds_read(&z[L::spec(x + k * R1, 0)]);
    }
    butterfly<R1>(v[0]);
    wave_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) z[L::spec(x * R1 + k, col)] = v[0][k];
    wave_sync();
  } else {
    // conj(C) back to its natural place (column 0 is already there), then the first stage reads its own distribution
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = lane / CW;
      if (!(holder && b == 0)) {
#pragma unroll
        for (int k = 0; k < R2; ++k) z[L::spec(x + k * R1, col)] = v[b][k];
      }
    }
    wave_sync();
    stage1(false);
  }
  {
    cf tw_col[7];
    tw.get(false, tw_col);
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = lane / CW;
#pragma unroll
      for (int k = 0; k < R2; ++k) v[b][k] = lds_read(&z[L::spec(x + k * R1, col)]);
      butterfly8_tw(v[b], tw_col);
    }
  }
  wave_sync();
#pragma unroll
  for (int b = 0; b < PER; ++b) {
    const int col = col0 + lane % CW + CW * b, x = lane / CW;
#pragma unroll
    for (int k = 0; k < R2; ++k) z[L::spec(x + k * R1, col)] = v[b][k];
  }
  wave_sync();
}

// ---- the wave's 8 row pairs (y1, y1 + H) along x + arg-max ------------------------------------------------------------
template <int N, int PK>
__device__ __forceinline__ Best half_row_pairs(cf* __restrict__ z, int row0, int lane, const HalfTw<N>& tw, int search_radius) {
  using L = HalfTile<N>;
  constexpr int R1 = L::R1, R2 = L::R2, H = L::H, PER = 8 * R1 / 64;
  {
    const int y1 = row0 + (lane >> 3), y2 = y1 + H, x = lane & 7;
    const bool x0 = x == 0;
    cf v[R1];
#pragma unroll
    for (int k = 0; k < R1; ++k) {
      const int u = x + k * R2;  // u < H exactly for k < R1/2 (x < R2)
      const int uu = (k < R1 / 2) ? u : ((k == R1 / 2 && x0) ? 0 : N - u);
      const cf a = lds_read(&z[L::spec(y1, uu)]), c = lds_read(&z[L::spec(y2, uu)]);
      cf e;
      if (k < R1 / 2) {
        e = {a.x - c.y, a.y + c.x};
        if (k == 0 && x0) e = {a.x, c.x};
      } else {
        e = {a.x + c.y, c.x - a.y};
        if (k == R1 / 2 && x0) e = {a.y, c.y};
      }
      v[k] = e;
    }
    butterfly<R1>(v);
    wave_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) z[L::out(y1, x * R1 + k)] = v[k];
    wave_sync();
  }
  cf v[PER][R2], tw_row[7];
  tw.get(true, tw_row);
#pragma unroll
  for (int b = 0; b < PER; ++b) {
    const int q = lane + 64 * b, y1 = row0 + q / R1, x = q % R1;
#pragma unroll
    for (int k = 0; k < R2; ++k) v[b][k] = lds_read(&z[L::out(y1, x + k * R1)]);
    butterfly8_tw(v[b], tw_row);
    if constexpr (PK == 1) {
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        v[b][k].x = ocl_scale_mask<N>(v[b][k].x, y1, x + k * R1, search_radius);
        v[b][k].y = ocl_scale_mask<N>(v[b][k].y, y1 + H, x + k * R1, search_radius);
      }
    }
  }
  wave_sync();
  float m = -__builtin_huge_valf();
#pragma unroll
  for (int b = 0; b < PER; ++b)
#pragma unroll
    for (int k = 0; k < R2; ++k) m = fmaxf(m, fmaxf(v[b][k].x, v[b][k].y));
  int mi = 0x7fffffff;
#pragma unroll
  for (int b = 0; b < PER; ++b) {
    const int q = lane + 64 * b, y1 = row0 + q / R1, x = q % R1;
#pragma unroll
    for (int k = 0; k < R2; ++k) {
      const int xx = x + k * R1, xs = (xx + H) & (N - 1);
      z[L::out(y1, xx)] = v[b][k];
      mi = min(mi, v[b][k].x == m ? (y1 + H) * N + xs : 0x7fffffff);  // row y1     -> shifted row y1 + H
      mi = min(mi, v[b][k].y == m ? y1 * N + xs : 0x7fffffff);        // row y1 + H -> shifted row y1
    }
  }
  return Best{m, mi};
}

// CH = 3: interleaved BGR8 frames, CV_RGB2GRAY in the load (as pc_seq_kernel.hip)
template <int N, int PK, int CH>
__global__ void __launch_bounds__(HalfTile<N>::T, (N == 64 ? 4 : 2)) pc_seq_half_kernel(PcArgs a, int n_pairs, int run) {
  using L = HalfTile<N>;
  constexpr int H = L::H, R1 = L::R1, W = HalfCfg<N>::WAVES, PPL = N / 8;  // pixels per lane and row
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  Best* red = reinterpret_cast<Best*>(z + H * HalfCfg<N>::PITCH);
  const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
  const int p0 = blockIdx.z * run;
  const int np = n_pairs - p0 < run ? n_pairs - p0 : run;
  const int patches = a.grid_x * a.grid_y, patch = blockIdx.y * a.grid_x + blockIdx.x;
  const int px0 = a.origin_x + blockIdx.x * a.stride_x, py0 = a.origin_y + blockIdx.y * a.stride_y;
  // this lane's 2 x PPL pixels: patch rows 2j, 2j + 1 (j = 8 wave + lane / 8), columns PPL (lane % 8) .. + PPL - 1
  const uint8_t* src = a.cur + (size_t)py0 * a.pitch + CH * px0 + (size_t)(2 * (8 * wave0 + (lane0 >> 3))) * a.pitch + CH * PPL * (lane0 & 7);
  HalfTw<N> tw;
  tw.init(a.twiddles, lane0);
  constexpr int PER = 8 * R1 / 64, NC0 = (H + 64) / 64;
  cf prev[PER][8], prev0[NC0], prevH[NC0];
#pragma unroll
  for (int b = 0; b < PER; ++b)
#pragma unroll
    for (int k = 0; k < 8; ++k) prev[b][k] = {0.f, 0.f};
#pragma unroll
  for (int i = 0; i < NC0; ++i) prev0[i] = prevH[i] = {0.f, 0.f};
  uint32_t ra[PPL / 4], rb[PPL / 4];
  auto fetch = [&](int f) {
    const uint8_t* s = src + (size_t)f * a.cur_stride;
    if constexpr (CH == 1) {
      __builtin_memcpy(ra, s, PPL);
      __builtin_memcpy(rb, s + a.pitch, PPL);
    } else if constexpr (PPL == 16) {
      gray16_from_bgr48(s, ra);
      gray16_from_bgr48(s + a.pitch, rb);
    } else {
      gray8_from_bgr24(s, ra);
      gray8_from_bgr24(s + a.pitch, rb);
    }
  };
  fetch(p0);
  for (int f = 0; f <= np; ++f) {
    int lane = lane0, wave = wave0;
    asm volatile("" : "+v"(lane), "+v"(wave));
    lane &= 63;
    wave &= W - 1;
    {
      constexpr bool RAW = MOF_RAW_STAGE && PPL == 16;
      if constexpr (RAW) {
        half_raw_store<N>(z, 8 * wave, lane, ra, rb);
      } else {
        const int lr = 8 * wave + (lane >> 3), c0 = PPL * (lane & 7);
#pragma unroll
        for (int i = 0; i < PPL; ++i)
          z[L::line(lr, c0 + i)] = {(float)((ra[i >> 2] >> (8 * (i & 3))) & 0xffu), (float)((rb[i >> 2] >> (8 * (i & 3))) & 0xffu)};
      }
      if (f < np) fetch(p0 + f + 1);
      wave_sync();
      half_rows<N, RAW>(z, 8 * wave, lane, tw);
    }
    __syncthreads();
    half_cols<N, PK>(z, 8 * wave, lane, tw, prev, prev0, prevH, f == 0, wave == 0);
    if (f == 0) {
      __syncthreads();
      continue;
    }
    __syncthreads();
    Best best = half_row_pairs<N, PK>(z, 8 * wave, lane, tw, a.search_radius);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
      best = better(best, o);
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    float wval = 0.f;
    if (wave == 0) {
      for (int w = 1; w < W; ++w) best = better(best, red[w]);
      wval = centroid_window_value<N, PK>(best, lane, [&](int ys, int xs) {
        const int y = (ys + H) & (N - 1), x = (xs + H) & (N - 1);
        const cf s = z[L::out(y & (H - 1), x)];
        return y < H ? s.x : s.y;
      });
    }
    __syncthreads();
    if (wave == 0)
      centroid_gate_store<N, PK>(best, wval, lane, a.max_px_speed_sq, a.out + 2 * ((size_t)(p0 + f - 1) * patches + patch));
  }
}

// ---- the PAIR form on the half tile (r05): independent frame pairs (BASELINE c4), a persistent workgroup per slab walks patch pairs
// p = blockIdx.x, + gridDim.x, ...: previous image -> rows, columns, spectrum parked in the workgroup's slab; current image -> rows,
// columns, cross-power against the slab, inverse columns, row pairs + arg-max, centroid. Same passes as the sequence kernel above,
// 1.5 transform units per pair like the packed pair kernel (pc_kernel.hip), but on HALF its LDS: TWO workgroups per CU, so one
// covers the other's barriers and latency chains (the packed N = 128 kernel runs 47 % of its wave-cycles parked, r03 counters).
#ifndef MOF_PAIR_HALF_WPE  // waves per SIMD the pair kernel is compiled for (4 = two workgroups per CU)
#define MOF_PAIR_HALF_WPE 4
#endif
template <int N, int PK, int CH>
__global__ void __launch_bounds__(HalfTile<N>::T, MOF_PAIR_HALF_WPE) pc_pair_half_kernel(PcArgs a, float2* __restrict__ slabs) {
  using L = HalfTile<N>;
  constexpr int H = L::H, R1 = L::R1, W = HalfCfg<N>::WAVES, PPL = N / 8;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  Best* red = reinterpret_cast<Best*>(z + H * HalfCfg<N>::PITCH);
  const int lane0 = threadIdx.x & 63, wave0 = threadIdx.x >> 6;
  const int patches = a.grid_x * a.grid_y;
  HalfTw<N> tw;
  tw.init(a.twiddles, lane0);
  constexpr int PER = 8 * R1 / 64, NC0 = (H + 64) / 64;
  const PrevInSlab park{slabs + (size_t)blockIdx.x * (PER * 8) * L::T + threadIdx.x, L::T};
  // this lane's 2 x PPL pixels of a patch: rows 2j, 2j + 1 (j = 8 wave + lane / 8), columns PPL (lane % 8) .. + PPL - 1
  const size_t lane_off = (size_t)(2 * (8 * wave0 + (lane0 >> 3))) * a.pitch + CH * PPL * (lane0 & 7);
  auto patch_off = [&](int pp, size_t frame_stride) -> size_t {
    const int pair = pp / patches, patch = pp - pair * patches, by = patch / a.grid_x, bx = patch - by * a.grid_x;
    return (size_t)pair * frame_stride + (size_t)(a.origin_y + by * a.stride_y) * a.pitch + (size_t)CH * (a.origin_x + bx * a.stride_x) + lane_off;
  };
  uint32_t ra[PPL / 4], rb[PPL / 4];
  auto fetch = [&](const uint8_t* s) {
    if constexpr (CH == 1) {
      __builtin_memcpy(ra, s, PPL);
      __builtin_memcpy(rb, s + a.pitch, PPL);
    } else if constexpr (PPL == 16) {
      gray16_from_bgr48(s, ra);
      gray16_from_bgr48(s + a.pitch, rb);
    } else {
      gray8_from_bgr24(s, ra);
      gray8_from_bgr24(s + a.pitch, rb);
    }
  };
  int p = blockIdx.x;
  if (p < a.total) fetch(a.prev + patch_off(p, a.prev_stride));
  for (; p < a.total; p += gridDim.x) {
    cf prev[PER][8], prev0[NC0], prevH[NC0];  // (prev itself is never live in this form: the slab holds it)
#pragma unroll
    for (int i = 0; i < NC0; ++i) prev0[i] = prevH[i] = {0.f, 0.f};
#pragma unroll
    for (int f = 0; f < 2; ++f) {  // f = 0: the previous image, f = 1: the current one
      int lane = lane0, wave = wave0;
      asm volatile("" : "+v"(lane), "+v"(wave));
      lane &= 63;
      wave &= W - 1;
      constexpr bool RAW = MOF_RAW_STAGE && PPL == 16;
      if constexpr (RAW) {
        half_raw_store<N>(z, 8 * wave, lane, ra, rb);
      } else {
        const int lr = 8 * wave + (lane >> 3), c0 = PPL * (lane & 7);
#pragma unroll
        for (int i = 0; i < PPL; ++i)
          z[L::line(lr, c0 + i)] = {(float)((ra[i >> 2] >> (8 * (i & 3))) & 0xffu), (float)((rb[i >> 2] >> (8 * (i & 3))) & 0xffu)};
      }
      // the next image's pixels are requested before this one is transformed: the current image of this pair, then the previous
      // image of the workgroup's next pair
      if (f == 0) fetch(a.cur + patch_off(p, a.cur_stride));
      else if (p + (int)gridDim.x < a.total) fetch(a.prev + patch_off(p + (int)gridDim.x, a.prev_stride));
      wave_sync();
      half_rows<N, RAW>(z, 8 * wave, lane, tw);
      __syncthreads();
      half_cols<N, PK, PrevInSlab>(z, 8 * wave, lane, tw, prev, prev0, prevH, f == 0, wave == 0, park);
      __syncthreads();
    }
    int lane = lane0, wave = wave0;
    asm volatile("" : "+v"(lane), "+v"(wave));
    lane &= 63;
    wave &= W - 1;
    // (no constant-patch rule: the images are transformed separately with power-of-two butterflies, so a constant patch has the exactly
    //  zero spectrum the reference's transforms give it and the flat surface produces the degenerate answer by itself, as in the
    //  sequence kernel above)
    Best best = half_row_pairs<N, PK>(z, 8 * wave, lane, tw, a.search_radius);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
      best = better(best, o);
    }
    if (lane == 0) red[wave] = best;
    __syncthreads();
    float wval = 0.f;
    if (wave == 0) {
      for (int w = 1; w < W; ++w) best = better(best, red[w]);
      wval = centroid_window_value<N, PK>(best, lane, [&](int ys, int xs) {
        const int y = (ys + H) & (N - 1), x = (xs + H) & (N - 1);
        const cf s = z[L::out(y & (H - 1), x)];
        return y < H ? s.x : s.y;
      });
    }
    __syncthreads();
    if (wave == 0) centroid_gate_store<N, PK>(best, wval, lane, a.max_px_speed_sq, a.out + 2 * (size_t)p);
  }
}

size_t half_extra_lds() {
  static const size_t v = [] {
    const char* e = getenv("MOF_PC_EXTRA_LDS");
    return e ? (size_t)atol(e) : (size_t)0;
  }();
  return v;
}

template <int N>
hipError_t configure_half() {
  const int lds = (int)(HalfTile<N>::LDS_BYTES + half_extra_lds());
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_half_kernel<N, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_half_kernel<N, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_half_kernel<N, 0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_seq_half_kernel<N, 1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

template <int N>
hipError_t launch_half(const PcArgs& a, int n_pairs, int run, hipStream_t stream) {
  const int runs = (n_pairs + run - 1) / run;
  if (runs > 65535 || (a.channels != 1 && a.channels != 3) || a.downscale != 1) return hipErrorInvalidValue;
  const dim3 g((unsigned)a.grid_x, (unsigned)a.grid_y, (unsigned)runs);
  const size_t lds = HalfTile<N>::LDS_BYTES + half_extra_lds();
  if (a.channels == 3) {
    if (a.peak_model == 1) hipLaunchKernelGGL((pc_seq_half_kernel<N, 1, 3>), g, dim3(HalfTile<N>::T), lds, stream, a, n_pairs, run);
    else hipLaunchKernelGGL((pc_seq_half_kernel<N, 0, 3>), g, dim3(HalfTile<N>::T), lds, stream, a, n_pairs, run);
  } else if (a.peak_model == 1) {
    hipLaunchKernelGGL((pc_seq_half_kernel<N, 1, 1>), g, dim3(HalfTile<N>::T), lds, stream, a, n_pairs, run);
  } else {
    hipLaunchKernelGGL((pc_seq_half_kernel<N, 0, 1>), g, dim3(HalfTile<N>::T), lds, stream, a, n_pairs, run);
  }
  return hipGetLastError();
}

template <int N>
hipError_t launch_pair_half(const PcArgs& a_in, int n_pairs, float* slabs, int n_slabs, hipStream_t stream) {
  if ((a_in.channels != 1 && a_in.channels != 3) || a_in.downscale != 1 || !slabs || n_slabs < 1) return hipErrorInvalidValue;
  PcArgs a = a_in;
  a.total = n_pairs * a.grid_x * a.grid_y;
  const unsigned blocks = (unsigned)(a.total < n_slabs ? a.total : n_slabs);
  const size_t lds = HalfTile<N>::LDS_BYTES + half_extra_lds();
  float2* sl = reinterpret_cast<float2*>(slabs);
  if (a.channels == 3) {
    if (a.peak_model == 1) hipLaunchKernelGGL((pc_pair_half_kernel<N, 1, 3>), dim3(blocks), dim3(HalfTile<N>::T), lds, stream, a, sl);
    else hipLaunchKernelGGL((pc_pair_half_kernel<N, 0, 3>), dim3(blocks), dim3(HalfTile<N>::T), lds, stream, a, sl);
  } else if (a.peak_model == 1) {
    hipLaunchKernelGGL((pc_pair_half_kernel<N, 1, 1>), dim3(blocks), dim3(HalfTile<N>::T), lds, stream, a, sl);
  } else {
    hipLaunchKernelGGL((pc_pair_half_kernel<N, 0, 1>), dim3(blocks), dim3(HalfTile<N>::T), lds, stream, a, sl);
  }
  return hipGetLastError();
}

template <int N>
hipError_t configure_pair_half() {
  const int lds = (int)(HalfTile<N>::LDS_BYTES + half_extra_lds());
  hipError_t e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_pair_half_kernel<N, 0, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_pair_half_kernel<N, 1, 1>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  if ((e = hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_pair_half_kernel<N, 0, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds)) != hipSuccess) return e;
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_pair_half_kernel<N, 1, 3>), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
}

}  // namespace

// The pair form on the half tile (N = 128): slabs = n_slabs * pc_pair_half_slab_floats() floats of device memory owned by the engine
bool pc_pair_half_supported(int patch_size) { return patch_size == 128; }
size_t pc_pair_half_slab_floats(int patch_size) { return patch_size == 128 ? (size_t)2 * (8 * HalfCfg<128>::R1 / 64) * 8 * HalfTile<128>::T : 0; }
hipError_t pc_configure_pair_half(int patch_size) { return patch_size == 128 ? configure_pair_half<128>() : hipErrorInvalidValue; }
hipError_t launch_pc_pair_half(const PcArgs& a, int patch_size, int n_pairs, float* slabs, int n_slabs, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  return patch_size == 128 ? launch_pair_half<128>(a, n_pairs, slabs, n_slabs, stream) : hipErrorInvalidValue;
}

bool pc_sequence_half_supported(int patch_size) { return patch_size == 64 || patch_size == 128; }

hipError_t pc_configure_sequence_half(int patch_size) {
  return patch_size == 64 ? configure_half<64>() : patch_size == 128 ? configure_half<128>() : hipErrorInvalidValue;
}

hipError_t launch_pc_sequence_half(const PcArgs& a, int patch_size, int n_pairs, int run, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  if (run < 1) run = 1;
  return patch_size == 64 ? launch_half<64>(a, n_pairs, run, stream)
                          : patch_size == 128 ? launch_half<128>(a, n_pairs, run, stream) : hipErrorInvalidValue;
}

}  // namespace mof
