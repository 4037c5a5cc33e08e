// pc_passes.hpp -- the in-LDS Stockham passes of K1 (power-of-two patches), shared by pc_kernel.hip (independent pairs)
// and pc_seq_kernel.hip (frame sequences). Device code only; see pc_kernel.hip for the design notes.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mof_kernels.h"
#include "pc_common.hpp"

#ifndef MOF_FUSE_XPOW
#define MOF_FUSE_XPOW 1  // 1: where it pays (PcTraits::FUSE_XPOW), the cross-power spectrum is formed inside the inverse row pass
#endif
#ifndef MOF_FUSE_XPOW_MAXN
#define MOF_FUSE_XPOW_MAXN 64  // (128: measured again under the r03 layout)
#endif
#ifndef MOF_XPOW_SPLIT
#define MOF_XPOW_SPLIT 1  // the inverse row pass of the waves that do not own row 0 carries no select for the packed row
#endif
#ifndef MOF_FUSED_TW
#define MOF_FUSED_TW 1  // twiddles of the radix-8 second stage fused into its first layer (butterfly8_tw)
#endif

namespace mof {

namespace {

template <int N>
struct Cfg;
template <>
struct Cfg<32> {
  static constexpr int R1 = 8, R2 = 4, SK = 3, PITCH = 36;
};
template <>
struct Cfg<64> {
  static constexpr int R1 = 8, R2 = 8, SK = 3, PITCH = 72;
};
#ifndef MOF_SK128
#define MOF_SK128 4
#endif
#ifndef MOF_LAYOUT128
#define MOF_LAYOUT128 1  // 1: pitch 140 + row skew + class-ordered lane maps (below); 0: the r01/r02 layout (pitch 136)
#endif
#ifndef MOF_PITCH128
#define MOF_PITCH128 (MOF_LAYOUT128 ? 140 : 136)
#endif
template <>
struct Cfg<128> {
  static constexpr int R1 = 16, R2 = 8, SK = MOF_SK128, PITCH = MOF_PITCH128;
};

// ---- bank layout of the 128 x 128 tile (r03; model: tools/design/lds_conflicts_128.py, 1.44x -> 1.05x conflict-free) ----
// ds_read_b64 serves 32 lanes per cycle over 64 dword banks (element indices distinct mod 32), ds_write_b64 16 lanes over
// 32 banks (distinct mod 16). With pitch 140 = 12 (mod 32) row r starts at bank-pair 12 r mod 32: the eight rows of an
// aligned block start at the eight multiples of 4 -- even rows at the multiples of 8, odd rows in between -- and
//   * a 32-lane group of 4 lines x 8 consecutive elements is conflict-free when the lines are one parity CLASS
//     ({0,2,4,6} or {1,3,5,7}; both closed under r -> -r, which the Hermitian reads of the inverse column pass need),
//   * a 16-lane group of 2 lines x 8 elements when the lines are 2 apart (280 = 8 mod 16),
//   * a 32-lane group of 2 lines x 16 elements when the lines are 4 apart (560 = 16 mod 32),
//   * 4 columns x 8 (reads) or x 4 (writes) consecutive rows always (12 r mod 32 distinct, 12 r mod 16 distinct).
// Which lines (or butterfly indices of a twiddle-free first stage) share a lane group is free -- the twiddle index of
// every lane stays what it was -- so the passes below only re-order them: ord1 = class order, ord2 = bit reversal.
// Rows 16 x + k of two lane-group neighbours x, x + 2 differ by 4480 = 0 (mod 16): the 8 (r >> 5) row skew separates them.
template <int N>
__device__ __forceinline__ int ord1(int j) {  // j = 0..7 -> 0 2 4 6 1 3 5 7
  if constexpr (N == 128 && MOF_LAYOUT128) return 2 * (j & 3) + (j >> 2);
  else return j;
}
template <int N>
__device__ __forceinline__ int ord2(int j) {  // j = 0..7 -> 0 4 2 6 1 5 3 7
  if constexpr (N == 128 && MOF_LAYOUT128) return ((j & 1) << 2) | (j & 2) | (j >> 2);
  else return j;
}

}  // namespace

#ifndef MOF_SKEW64
#define MOF_SKEW64 1  // 64 x 64 tile: rows 8 x + k of lane-group neighbours x, x + 1 are 576 = 0 (mod 16) apart -> 8 (r >> 3) row skew
#endif
template <int N>
struct PcTraits {
#ifndef MOF_K1_T64
#define MOF_K1_T64 256
#endif
  static constexpr int T = (N == 64) ? MOF_K1_T64 : N * N / 16;
  static constexpr int WAVES = T / 64;
  static constexpr int LPW = N / WAVES;  // lines (rows or columns) owned by a wave
  static constexpr int R1 = Cfg<N>::R1, R2 = Cfg<N>::R2, SK = Cfg<N>::SK, PITCH = Cfg<N>::PITCH;
  static constexpr int BMIN = R1 < R2 ? R1 : R2;
  // inverse (half-size) passes: lines per active wave, active waves
  static constexpr int LI = (LPW / 2 > 64 / BMIN) ? LPW / 2 : 64 / BMIN;
  static constexpr int WI = (N / 2) / LI;
  // complex elements of the tile, incl. the row skew (zaddr): 8 (r >> 5) at N = 128, 8 (r >> 3) at N = 64
  static constexpr int TILE = N * PITCH + ((N == 128 && MOF_LAYOUT128) ? 24 : 0) + ((N == 64 && MOF_SKEW64) ? 56 : 0);
  static constexpr size_t LDS_BYTES = sizeof(float) * 2 * (size_t)TILE + 64 * sizeof(Best);
#ifndef MOF_PERSIST_MIN_N
#define MOF_PERSIST_MIN_N 128
#endif
  static constexpr bool PERSIST = N >= MOF_PERSIST_MIN_N;  // see pc_field_kernel
  // cross-power spectrum inside the inverse row pass (row_pass_xpow): one barrier and 1.5 tile passes fewer. Same-box
  // A/B: +4.4 % at N = 64 (several workgroups per CU, latency-bound), -4.4 % at N = 128 (one workgroup per CU: the
  // separate pass spreads the cross-power over all 16 waves, the fused one over the 8 that run the inverse)
  static constexpr bool FUSE_XPOW = MOF_FUSE_XPOW && N <= MOF_FUSE_XPOW_MAXN;
  static_assert(R1 * R2 == N && T % 64 == 0 && 64 % R1 == 0 && 64 % R2 == 0, "bad plan");
};

// z(r, c): skewed tile address in complex units
template <int N>
__device__ __forceinline__ int zaddr(int r, int c) {
  if constexpr (N == 128 && MOF_LAYOUT128) return r * PcTraits<N>::PITCH + 8 * (r >> 5) + c + (c >> PcTraits<N>::SK);
  else if constexpr (N == 64 && MOF_SKEW64) return r * PcTraits<N>::PITCH + 8 * (r >> 3) + c + (c >> PcTraits<N>::SK);
  else return r * PcTraits<N>::PITCH + c + (c >> PcTraits<N>::SK);
}

// ---- raw pixel staging (MOF_RAW_STAGE, N = 128) -------------------------------------------------------------------
// ds_write_b64 costs 6 cycles of the VGPR -> LDS path per wave-instruction (MI355X_MICROARCH.md, LDS table) and that path,
// not the arithmetic, is what the 128 x 128 kernel waits for. Converting a lane's 16 + 16 pixels to 16 complex floats
// BEFORE the tile store is 16 such stores; instead the lane interleaves the raw bytes (c0 p0 c1 p1 ..: 8 v_perm) and
// stores 32 B with two ds_write_b128 into a per-wave raw area that overlays the wave's own (not yet written) tile rows,
// and the first row stage reads its operands as 16 ds_read_u16 and converts them on the way into the butterfly.
// Slot s = lane / CPR of the wave holds row line0 + ord1(s); a slot pitch of 2 N + 16 B puts the four slots of a 32-lane read
// group 16 B apart (mod 128): conflict-free like the writes (8 lanes x 16 B at a 32-B stride; N = 64: two slots interleave).
// N = 64 (c2): same-box A/B in DESIGN.md.
#ifndef MOF_RAW_STAGE
#define MOF_RAW_STAGE 1
#endif
template <int N>
struct RawCfg {
  static constexpr int PITCH = 2 * N + 16, CPR = N / 16;  // bytes per slot (272 / 144), 16-pixel chunks per row
};
template <int N>
__device__ __forceinline__ unsigned char* raw_area(cf* z, int line0) {
  return reinterpret_cast<unsigned char*>(z + zaddr<N>(line0, 0));
}
__device__ __forceinline__ uint32_t lds_read_u16(const unsigned char* p) {
  typedef const volatile uint16_t __attribute__((address_space(3))) * lds_u16_ptr;
  return *(lds_u16_ptr)(p);
}
// this lane's 16 + 16 pixels (4 + 4 dwords) -> interleaved bytes -> raw area (slot = lane / CPR, chunk = lane % CPR)
template <int N>
__device__ __forceinline__ void raw_store(cf* z, int line0, int lane, const uint32_t* cw, const uint32_t* pw) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef u4 __attribute__((address_space(3))) * lds_u4_ptr;
  unsigned char* dst = raw_area<N>(z, line0) + (lane / RawCfg<N>::CPR) * RawCfg<N>::PITCH + (lane % RawCfg<N>::CPR) * 32;
  u4 lo, hi;
  lo.x = __builtin_amdgcn_perm(pw[0], cw[0], 0x05010400u);
  lo.y = __builtin_amdgcn_perm(pw[0], cw[0], 0x07030602u);
  lo.z = __builtin_amdgcn_perm(pw[1], cw[1], 0x05010400u);
  lo.w = __builtin_amdgcn_perm(pw[1], cw[1], 0x07030602u);
  hi.x = __builtin_amdgcn_perm(pw[2], cw[2], 0x05010400u);
  hi.y = __builtin_amdgcn_perm(pw[2], cw[2], 0x07030602u);
  hi.z = __builtin_amdgcn_perm(pw[3], cw[3], 0x05010400u);
  hi.w = __builtin_amdgcn_perm(pw[3], cw[3], 0x07030602u);
  *(lds_u4_ptr)(dst) = lo;
  *(lds_u4_ptr)(dst + 16) = hi;
}

// 8 lines x 8 chunks of 8 + 8 pixels per wave (the sequence kernel's lines = patch rows 2j | 2j + 1): one ds_write_b128 per lane
template <int N>
__device__ __forceinline__ void raw_store8(cf* z, int line0, int lane, const uint32_t* a, const uint32_t* b) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef u4 __attribute__((address_space(3))) * lds_u4_ptr;
  u4 d;
  d.x = __builtin_amdgcn_perm(b[0], a[0], 0x05010400u);
  d.y = __builtin_amdgcn_perm(b[0], a[0], 0x07030602u);
  d.z = __builtin_amdgcn_perm(b[1], a[1], 0x05010400u);
  d.w = __builtin_amdgcn_perm(b[1], a[1], 0x07030602u);
  *(lds_u4_ptr)(raw_area<N>(z, line0) + (lane >> 3) * RawCfg<N>::PITCH + (lane & 7) * 16) = d;
}

// ---- row pass over LINES lines starting at line0 (wave-local) --------------------------------
// RAW: stage 1 takes z = cur + i prev from the wave's raw area (raw_store above) instead of the tile
template <int N, int LINES, bool RAW = false>
__device__ __forceinline__ void row_pass(cf* __restrict__ z, int line0, int lane, const cf* tw_row) {
  using P = PcTraits<N>;
  constexpr int R1 = P::R1, R2 = P::R2;
  {  // stage 1: radix R1, P = 1; R2 butterflies per line
    constexpr int PER = LINES * R2 / 64;
    static_assert(LINES * R2 % 64 == 0, "row stage 1 does not fill the wave");
    static_assert(!RAW || R2 == 8, "raw staging: one slot per line, 8 butterflies per line");
    cf v[PER][R1];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord1<N>(q / R2), x = q % R2;
      if constexpr (RAW) {
        const unsigned char* src = raw_area<N>(z, line0) + (q / R2) * RawCfg<N>::PITCH + 2 * x;
#pragma unroll
        for (int k = 0; k < R1; ++k) {
          const uint32_t cp = lds_read_u16(src + 2 * k * R2);
          v[b][k] = {(float)(cp & 0xffu), (float)(cp >> 8)};
        }
      } else {
#pragma unroll
        for (int k = 0; k < R1; ++k) v[b][k] = lds_read(&z[zaddr<N>(line, x + k * R2)]);
      }
      butterfly<R1>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord1<N>(q / R2), x = q % R2;
#pragma unroll
      for (int k = 0; k < R1; ++k) z[zaddr<N>(line, x * R1 + k)] = v[b][k];
    }
    wave_sync();
  }
  {  // stage 2: radix R2, P = R1; R1 butterflies per line, twiddle W_N^{k x}
    constexpr int PER = LINES * R1 / 64;
    static_assert(LINES * R1 % 64 == 0, "row stage 2 does not fill the wave");
    cf v[PER][R2];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord2<N>(q / R1), x = q % R1;
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        cf a = lds_read(&z[zaddr<N>(line, x + k * R1)]);
#ifdef MOF_ABLATE_NOBFLY
        v[b][k] = a;
#else
        v[b][k] = (k == 0 || (R2 == 8 && MOF_FUSED_TW)) ? a : cmul(a, tw_row[k - 1]);
#endif
      }
      if constexpr (R2 == 8 && MOF_FUSED_TW) butterfly8_tw(v[b], tw_row);
      else butterfly<R2>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord2<N>(q / R1), x = q % R1;
#pragma unroll
      for (int k = 0; k < R2; ++k) z[zaddr<N>(line, x + k * R1)] = v[b][k];
    }
    wave_sync();
  }
}

// ---- inverse row pass with the cross-power spectrum formed on the fly (MOF_FUSE_XPOW) ------------------------------
// Rows 0..H-1 of the half spectrum D = conj(C): stage 1 reads bin (v, u) AND its Hermitian partner (N-v, N-u) from the
// forward spectrum, forms conj(C[v][u]) in registers and goes straight into the first butterfly. The separate
// cross-power pass (a tile read and half a tile write, one workgroup barrier) disappears. Partner rows are N-1..H+1,
// which nobody writes in this phase; row 0 (packed with row H by the owning wave beforehand) is taken as it is.
// HAS_ROW0: the wave that owns row 0 (wave 0); the others never meet the packed row and carry no per-element select for it.
template <int N, int PK, bool HAS_ROW0>
__device__ __forceinline__ void row_pass_xpow(cf* __restrict__ z, int line0, int lane, const cf* tw_row) {
  using P = PcTraits<N>;
  constexpr int R1 = P::R1, R2 = P::R2, LINES = P::LI;
  {
    constexpr int PER = LINES * R2 / 64;
    static_assert(LINES * R2 % 64 == 0, "row stage 1 does not fill the wave");
    cf v[PER][R1];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord1<N>(q / R2), x = q % R2;
      const int pline = (N - line) % N;
      const bool packed = HAS_ROW0 && line == 0;
#pragma unroll
      for (int k = 0; k < R1; ++k) {
        const int u = x + k * R2, um = (N - u) % N;
        const cf zk = lds_read(&z[zaddr<N>(line, u)]), zm = lds_read(&z[zaddr<N>(pline, um)]);
        const cf C = cross_power<PK>(zk, zm, false);
        v[b][k] = packed ? zk : cf{C.x, -C.y};
      }
      butterfly<R1>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord1<N>(q / R2), x = q % R2;
#pragma unroll
      for (int k = 0; k < R1; ++k) z[zaddr<N>(line, x * R1 + k)] = v[b][k];
    }
    wave_sync();
  }
  {  // stage 2: as row_pass
    constexpr int PER = LINES * R1 / 64;
    cf v[PER][R2];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord2<N>(q / R1), x = q % R1;
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        cf a = lds_read(&z[zaddr<N>(line, x + k * R1)]);
        v[b][k] = (k == 0 || (R2 == 8 && MOF_FUSED_TW)) ? a : cmul(a, tw_row[k - 1]);
      }
      if constexpr (R2 == 8 && MOF_FUSED_TW) butterfly8_tw(v[b], tw_row);
      else butterfly<R2>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int q = lane + 64 * b;
      const int line = line0 + ord2<N>(q / R1), x = q % R1;
#pragma unroll
      for (int k = 0; k < R2; ++k) z[zaddr<N>(line, x + k * R1)] = v[b][k];
    }
    wave_sync();
  }
}

// ---- forward column pass over the LPW columns starting at col0 (wave-local) --------------------
template <int N>
__device__ __forceinline__ void col_pass_fwd(cf* __restrict__ z, int col0, int lane, const cf* tw_col) {
  using P = PcTraits<N>;
  constexpr int R1 = P::R1, R2 = P::R2, LPW = P::LPW;
  {  // stage 1: R2 butterflies per column
    constexpr int PER = LPW * R2 / 64, CW = 64 / R2;
    cf v[PER][R1];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = ord1<N>(lane / CW);
#pragma unroll
      for (int k = 0; k < R1; ++k) v[b][k] = lds_read(&z[zaddr<N>(x + k * R2, col)]);
      butterfly<R1>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = ord1<N>(lane / CW);
#pragma unroll
      for (int k = 0; k < R1; ++k) z[zaddr<N>(x * R1 + k, col)] = v[b][k];
    }
    wave_sync();
  }
  {  // stage 2: R1 butterflies per column
    constexpr int PER = LPW * R1 / 64, CW = 64 / R1;
    cf v[PER][R2];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = lane / CW;
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        cf a = lds_read(&z[zaddr<N>(x + k * R1, col)]);
#ifdef MOF_ABLATE_NOBFLY
        v[b][k] = a;
#else
        v[b][k] = (k == 0 || (R2 == 8 && MOF_FUSED_TW)) ? a : cmul(a, tw_col[k - 1]);
#endif
      }
      if constexpr (R2 == 8 && MOF_FUSED_TW) butterfly8_tw(v[b], tw_col);
      else butterfly<R2>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = lane / CW;
#pragma unroll
      for (int k = 0; k < R2; ++k) z[zaddr<N>(x + k * R1, col)] = v[b][k];
    }
    wave_sync();
  }
}

// ---- inverse column pass on LI column PAIRS (x1, x1 + N/2) starting at col0 (wave-local) -------
// Input: rows 0..N/2-1 hold F1 = FFT_u(conj C) of the packed half spectrum: row 0 = F1[0] + i F1[N/2]
// (both real), rows 1..N/2-1 = F1[v]; F1[N-v] = conj F1[v]. Column x of the result is real, so two
// columns ride one complex transform: E[v] = F1[v][x1] + i F1[v][x2]; Re/Im of its transform are the
// correlation surface at columns x1 / x2. Output: z(y, x1) = (c[y][x1], c[y][x1 + N/2]); returns the
// lane's best (value, shifted index).
template <int N, int PK>
__device__ __forceinline__ Best col_pass_inv(cf* __restrict__ z, int col0, int lane, const cf* tw_col, int search_radius) {
  using P = PcTraits<N>;
  constexpr int R1 = P::R1, R2 = P::R2, LI = P::LI, H = N / 2;
  {  // stage 1
    constexpr int PER = LI * R2 / 64, CW = 64 / R2;
    static_assert(PER >= 1, "inverse column stage 1 does not fill the wave");
    cf v[PER][R1];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = ord1<N>(lane / CW);
      // r = x + k R2 with x < R2: r < H exactly for k < R1/2, so the four Hermitian cases are decided at compile time
      // per k, except for the lanes with x == 0 at k = 0 (r = 0) and k = R1/2 (r = H), which read the packed row 0
      static_assert(R1 % 2 == 0 && 64 / CW == R2, "row classes below assume x < R2 and an even R1");
      const bool x0 = x == 0;
#pragma unroll
      for (int k = 0; k < R1; ++k) {
        const int r = x + k * R2;  // 0..N-1
        const int rr = (k < R1 / 2) ? r : ((k == R1 / 2 && x0) ? 0 : N - r);
        const cf a = lds_read(&z[zaddr<N>(rr, col)]), c = lds_read(&z[zaddr<N>(rr, col + H)]);
        cf e;
        if (k < R1 / 2) {
          e = {a.x - c.y, a.y + c.x};
          if (k == 0 && x0) e = {a.x, c.x};
        } else {
          e = {a.x + c.y, c.x - a.y};
          if (k == R1 / 2 && x0) e = {a.y, c.y};
        }
        v[b][k] = e;
      }
      butterfly<R1>(v[b]);
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = ord1<N>(lane / CW);
#pragma unroll
      for (int k = 0; k < R1; ++k) z[zaddr<N>(x * R1 + k, col)] = v[b][k];
    }
    wave_sync();
  }
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  {  // stage 2 + arg-max from registers (fftShift :1297-1305, minMaxLoc :1539)
    constexpr int PER = LI * R1 / 64, CW = 64 / R1;
    cf v[PER][R2];
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = lane / CW;
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        cf a = lds_read(&z[zaddr<N>(x + k * R1, col)]);
#ifdef MOF_ABLATE_NOBFLY
        v[b][k] = a;
#else
        v[b][k] = (k == 0 || (R2 == 8 && MOF_FUSED_TW)) ? a : cmul(a, tw_col[k - 1]);
#endif
      }
      if constexpr (R2 == 8 && MOF_FUSED_TW) butterfly8_tw(v[b], tw_col);
      else butterfly<R2>(v[b]);
      if constexpr (PK == 1) {  // OpenCL-kernel model: 1/N^2 scaling and the +-search_radius mask (cl:733, :737-746, :823-826)
#pragma unroll
        for (int k = 0; k < R2; ++k) {
          const int y = x + k * R1;
          v[b][k].x = ocl_scale_mask<N>(v[b][k].x, y, col, search_radius);
          v[b][k].y = ocl_scale_mask<N>(v[b][k].y, y, col + H, search_radius);
        }
      }
    }
    wave_sync();
    // lane-local first maximum in two steps (max value, then the smallest shifted index that attains it): half the
    // instructions of a running (value, index) compare per candidate
    float m = -__builtin_huge_valf();
#pragma unroll
    for (int b = 0; b < PER; ++b)
#pragma unroll
      for (int k = 0; k < R2; ++k) m = fmaxf(m, fmaxf(v[b][k].x, v[b][k].y));
    int mi = 0x7fffffff;
#pragma unroll
    for (int b = 0; b < PER; ++b) {
      const int col = col0 + lane % CW + CW * b, x = lane / CW;
#pragma unroll
      for (int k = 0; k < R2; ++k) {
        const int y = x + k * R1;
        z[zaddr<N>(y, col)] = v[b][k];
        const int base = ((y + H) % N) * N + col;
        mi = min(mi, v[b][k].x == m ? base + H : 0x7fffffff);  // column col      -> shifted col + H
        mi = min(mi, v[b][k].y == m ? base : 0x7fffffff);      // column col + H  -> shifted col
      }
    }
    best = Best{m, mi};
  }
  return best;
}

}  // namespace mof
