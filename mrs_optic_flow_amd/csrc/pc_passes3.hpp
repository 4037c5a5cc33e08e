// pc_passes3.hpp -- the forward 2-D transform of a 64 x 64 tile in THREE register stages instead of four (r03).
//
// row_pass + col_pass_fwd (pc_passes.hpp) split each 1-D transform 8 x 8: four stages, four tile stores, three tile reads.
// 16 elements per lane also hold 4 + 4 + 4 of the transform's 12 radix-2 levels:
//   S1  rows, radix 16 over x = 4 n1 + n2 (n1 = 0..15), lane = (row y, n2):            A[y][n2][k1]
//   S2  a 4 x 4 block per lane = (k1, m2): the rows' last radix 4 over n2 (twiddle W64^{n2 k1}) and the columns' first
//       radix 4 over m1, y = m2 + 16 m1, then the column twiddle W64^{m2 l1}:          B[m2][l1][u = k1 + 16 k2]
//   S3  columns, radix 16 over m2, lane = (l1, u):                                      Z[v = l1 + 4 l2][u]
// (x u = 4 n1 k1 + n2 k1 + 16 n2 k2 (mod 64) and y v = m2 l1 + 4 m2 l2 + 16 m1 l1 (mod 64).) One LDS round trip less per
// patch pair; S2 is in place and S2 -> S3 is wave-local (wave w owns k1 = 4w..4w+3, i.e. the columns u = k1 + 16 k2).
// Intermediate layout: element (y, b, k1) at 68 y + 4 (y >> 4) + 17 b + k1 -- S1's stores (4 rows x 4 b per 16 lanes), S2's
// loads (8 m2 x 4 k1 per 32 lanes) and stores are conflict-free, S3's loads 2-way (tools/design: no member of this family
// clears all four). S3 leaves Z in the standard tile layout (zaddr) for the cross-power / inverse passes; the two layouts
// overlap in memory, so every wave reads its S3 operands before any wave stores its results (one more barrier).
#pragma once

#include "pc_passes.hpp"

namespace mof {

struct Fwd3Tw {
  cf wr[3], wc[3];  // W64^{n2 k1}, n2 = 1..3 (k1 = 4 wave + lane % 4);  W64^{m2 l1}, l1 = 1..3 (m2 = lane / 4)
  __device__ __forceinline__ void load(const float* __restrict__ table, int wave, int lane) {
    const int k1 = 4 * wave + (lane & 3), m2 = lane >> 2;
#pragma unroll
    for (int i = 1; i < 4; ++i) {
      wr[i - 1] = {table[2 * (i * k1)], table[2 * (i * k1) + 1]};
      wc[i - 1] = {table[2 * (i * m2)], table[2 * (i * m2) + 1]};
    }
  }
};

__device__ __forceinline__ int t3addr(int y, int b, int k1) { return 68 * y + 4 * (y >> 4) + 17 * b + k1; }

// S1: the wave's 16 rows from its raw area (raw_store, pc_passes.hpp); wave-local
__device__ __forceinline__ void fwd3_rows(cf* __restrict__ z, int wave, int lane) {
  constexpr int N = 64;
  const int slot = lane >> 2, n2 = lane & 3, y = 16 * wave + slot;
  const unsigned char* src = raw_area<N>(z, 16 * wave) + slot * RawCfg<N>::PITCH + 2 * n2;
  cf v[16];
#pragma unroll
  for (int n1 = 0; n1 < 16; ++n1) {
    const uint32_t cp = lds_read_u16(src + 8 * n1);
    v[n1] = {(float)(cp & 0xffu), (float)(cp >> 8)};
  }
#if !defined(MOF_ABLATE_ROWS) && !defined(MOF_ABLATE_S1)  // diagnostic builds (tools/ab_mfma_bound.sh): the row arithmetic an
  butterfly<16>(v);                                       // MFMA row-DFT would take off the VALU (S1 alone, or S1 and S2's row part)
#endif
  wave_sync();  // the raw area lies inside the wave's own part of the intermediate layout: all of it is read by now
#pragma unroll
  for (int k1 = 0; k1 < 16; ++k1) z[t3addr(y, n2, k1)] = v[k1];
}

// S1 on the matrix cores (MOF_K1_MFMA_S1). The radix-16 DFT over n1 of a column (y, n2) of z = cur + i prev is
// a real 32 x 32 product: rows r = (k1, re | im), K slots (n1, cur | prev),
//   re_k1 = sum_n1 cur cos(t) + prev sin(t),  im_k1 = sum_n1 -cur sin(t) + prev cos(t),  t = 2 pi k1 n1 / 16.
// u8 pixels are exact in f16; the matrix is split W = W_hi + W_lo in f16 (both products exact in f32, f32 accumulation), so
// the pass is as accurate as the fp32 butterfly and its k1 = 0, 4, 8, 12 outputs are the same exact integer sums.
// v_mfma_f32_32x32x16_f16: A = W (row r = lane % 32; k = 8 (lane / 32) + j), B = pixels (column = lane % 32 = 8 rows x 4 n2,
// same k), D: column = lane % 32, row (i % 4) + 8 (i / 4) + 4 (lane / 32) -- re and im of a k1 in one lane. A wave owns
// 16 rows x 4 n2 = two column tiles x (2 K steps x (hi, lo)) = 8 MFMA; the 16 dwords of W fragments per lane come from the
// table behind the twiddles (pc_mfma_s1_fragments, built by the host).
typedef _Float16 mof_half8 __attribute__((ext_vector_type(8)));
typedef _Float16 mof_half2 __attribute__((ext_vector_type(2)));
typedef float mof_float16 __attribute__((ext_vector_type(16)));
struct Fwd3Mfma {
  mof_half8 w[4];  // hi n1 0..7, hi n1 8..15, lo, lo
  __device__ __forceinline__ void load(const float* __restrict__ table, int lane) {
    typedef uint32_t u4 __attribute__((ext_vector_type(4)));
    const u4* frag = reinterpret_cast<const u4*>(table + 128);
#pragma unroll
    for (int s = 0; s < 4; ++s) w[s] = __builtin_bit_cast(mof_half8, frag[64 * s + lane]);
  }
};
__device__ __forceinline__ void fwd3_rows_mfma(cf* __restrict__ z, int wave, int lane, const float* __restrict__ table) {
  constexpr int N = 64;
  Fwd3Mfma m;  // (L1 / L2 hits, requested ahead of the LDS reads below; held only for this stage)
  m.load(table, lane);
  const int n = lane & 31, h = lane >> 5, n2 = n & 3;
  // column tile t: rows 16 wave + 8 t + (n >> 2); pixel n1 = 8 ks + 4 h + q of column n2: all 16 reads in flight, then the conversions
  uint32_t cp[2][2][4];
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    const unsigned char* src = raw_area<N>(z, 16 * wave) + (8 * t + (n >> 2)) * RawCfg<N>::PITCH + 2 * n2 + 32 * h;
#pragma unroll
    for (int ks = 0; ks < 2; ++ks)
#pragma unroll
      for (int q = 0; q < 4; ++q) cp[t][ks][q] = lds_read_u16(src + 64 * ks + 8 * q);
  }
  mof_half8 b[2][2];
#pragma unroll
  for (int t = 0; t < 2; ++t)
#pragma unroll
    for (int ks = 0; ks < 2; ++ks) {
      uint32_t d[4];
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // (cur, prev) bytes -> the halves 1024 + cur, 1024 + prev -> - 1024
        const mof_half2 k1024 = {(_Float16)1024.f, (_Float16)1024.f};
        const mof_half2 v = __builtin_bit_cast(mof_half2, __builtin_amdgcn_perm(0x64646464u, cp[t][ks][q], 0x04010400u)) - k1024;
        d[q] = __builtin_bit_cast(uint32_t, v);
      }
      __builtin_memcpy(&b[t][ks], d, 16);
    }
  wave_sync();  // (the raw area lies inside the wave's own part of the intermediate layout)
#pragma unroll
  for (int t = 0; t < 2; ++t) {
    mof_float16 acc;
#pragma unroll
    for (int i = 0; i < 16; ++i) acc[i] = 0.f;
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m.w[0], b[t][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m.w[1], b[t][1], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m.w[2], b[t][0], acc, 0, 0, 0);
    acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(m.w[3], b[t][1], acc, 0, 0, 0);
    const int y = 16 * wave + 8 * t + (n >> 2);
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int e = 0; e < 2; ++e) z[t3addr(y, n2, 4 * q + 2 * h + e)] = {acc[4 * q + 2 * e], acc[4 * q + 2 * e + 1]};
  }
}

// S2: in place
__device__ __forceinline__ void fwd3_mid(cf* __restrict__ z, int wave, int lane, const Fwd3Tw& tw) {
  const int k1 = 4 * wave + (lane & 3), m2 = lane >> 2;
  cf e[4][4];  // [m1][n2] -> [m1][k2] -> [l1][k2]
#pragma unroll
  for (int m1 = 0; m1 < 4; ++m1)
#pragma unroll
    for (int n2 = 0; n2 < 4; ++n2) e[m1][n2] = lds_read(&z[t3addr(m2 + 16 * m1, n2, k1)]);
#pragma unroll
  for (int m1 = 0; m1 < 4; ++m1) {
#ifdef MOF_ABLATE_ROWS
    cf t[4] = {e[m1][0], e[m1][1], e[m1][2], e[m1][3]};
#else
    cf t[4] = {e[m1][0], cmul(e[m1][1], tw.wr[0]), cmul(e[m1][2], tw.wr[1]), cmul(e[m1][3], tw.wr[2])};
    butterfly<4>(t);
#endif
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) e[m1][k2] = t[k2];
  }
#pragma unroll
  for (int k2 = 0; k2 < 4; ++k2) {
    cf c[4] = {e[0][k2], e[1][k2], e[2][k2], e[3][k2]};
    butterfly<4>(c);
    e[0][k2] = c[0];
#pragma unroll
    for (int l1 = 1; l1 < 4; ++l1) e[l1][k2] = cmul(c[l1], tw.wc[l1 - 1]);
  }
#pragma unroll
  for (int l1 = 0; l1 < 4; ++l1)
#pragma unroll
    for (int k2 = 0; k2 < 4; ++k2) z[t3addr(m2 + 16 * l1, k2, k1)] = e[l1][k2];
}

// S3: loads (wave-local after S2), workgroup barrier, stores in the standard layout. Lane map: 16-lane store groups hold
// four j, two k2 of one parity and two neighbouring l1 -- distinct banks under zaddr<64>.
__device__ __forceinline__ void fwd3_cols(cf* __restrict__ z, int wave, int lane) {
  constexpr int N = 64;
  const int j = lane & 3, k2 = 2 * ((lane >> 2) & 1) + ((lane >> 4) & 1), l1 = ((lane >> 3) & 1) + 2 * (lane >> 5);
  const int u = 4 * wave + j + 16 * k2;
  cf v[16];
#pragma unroll
  for (int m2 = 0; m2 < 16; ++m2) v[m2] = lds_read(&z[t3addr(m2 + 16 * l1, k2, 4 * wave + j)]);
  butterfly<16>(v);
  __syncthreads();
#pragma unroll
  for (int l2 = 0; l2 < 16; ++l2) z[zaddr<N>(l1 + 4 * l2, u)] = v[l2];
}

}  // namespace mof
