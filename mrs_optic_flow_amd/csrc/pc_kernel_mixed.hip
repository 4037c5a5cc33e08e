// pc_kernel_mixed.hip -- K1 for the reference's DEFAULT patch size, N = 120 = 15 x 8
// (config/default.yaml:31-32: frame_size 480, sample_point_size 120 -> 4 x 4 patches; the reference's own
// OpenCL plan for it is radix {8, 5, 3}, src/FftMethod.cpp:481-539).
//
// Same algorithm and structure as pc_kernel.hip (two-for-one packed forward transform, normalised
// cross-power spectrum with the real-only-slot rule, Hermitian half-size inverse, arg-max from registers,
// fp64 centroid + gate; wave-local passes, 5 workgroup barriers), with
//   * two Stockham stages per 1-D transform: radix 15 (3 x 5 Cooley-Tukey in registers), then radix 8;
//   * 15 waves (960 threads), wave w owns lines [8w, 8w+8); the radix-8 stage has 15 butterflies per line,
//     which does not divide the wave, so lanes are predicated and the inter-stage twiddles W_120^{kx} are
//     fetched from the (L1-resident, host-computed-in-double) table instead of living in registers;
//   * tile pitch 136 complex (= 8 mod 32) with the lane maps of row_pass / col_pass: every ds_read_b64 / ds_write_b64
//     group of the transform passes hits distinct banks (tools/design/lds_conflicts_120.py; r01: 0.85 -> 0.95 M pairs/s);
//   * persistent: one resident workgroup per CU (the tile is 130 KB of LDS) that prefetches the next patch's pixels.
// The radix-3/5/15 butterflies live in pc_common.hpp (shared with sr_kernel.hip).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mof_kernels.h"
#include "pc_common.hpp"

namespace mof {

namespace {

constexpr int N = 120, H = 60, R1 = 15, R2 = 8, LPW = 8, WAVES = N / LPW, T = WAVES * 64;
// Row pitch 136 = 8 (mod 32) complex elements: with the lane maps below every ds_read_b64 group (32 lanes, 64 banks)
// and every ds_write_b64 group (16 lanes, 32 banks) of the transform passes hits distinct banks
// (tools/design/lds_conflicts_120.py; pitch 121 cost 2.3x the conflict-free LDS cycles, SQ_LDS_BANK_CONFLICT = 53 %).
#ifndef MOF_PITCH120
#define MOF_PITCH120 136
#endif
constexpr int PITCH = MOF_PITCH120;
#ifndef MOF_TW120_LDS  // 1: the 120 inter-stage twiddles live in LDS (one copy per persistent workgroup); 0: fetched from the L1-resident table (r01-r04)
#define MOF_TW120_LDS 1
#endif
constexpr size_t LDS_BYTES_120 = sizeof(float) * 2 * (size_t)N * PITCH + 64 * sizeof(Best) + (MOF_TW120_LDS ? sizeof(float) * 2 * N : 0);

__device__ __forceinline__ int za(int r, int c) { return r * PITCH + c; }

__device__ __forceinline__ cf twiddle(const float* __restrict__ table, int idx) {  // W_120^idx, idx < 120
#if MOF_TW120_LDS
  return lds_read(reinterpret_cast<const cf*>(table) + idx);  // (`table` points into LDS: the kernel copied the 120 entries there)
#else
  const float2 t = *reinterpret_cast<const float2*>(table + 2 * idx);
  return {t.x, t.y};
#endif
}

// ---- raw pixel staging (as pc_passes.hpp, raw_store): a chunk's 8 + 8 pixels leave as ONE ds_write_b128 of interleaved
// bytes (c0 p0 c1 p1 ..) into a per-wave area that overlays the wave's own tile rows, and the first row stage reads its
// operands with ds_read_u16 and converts them on the way into the butterfly -- 8 ds_write_b64 (48 cycles of the
// VGPR -> LDS path) per chunk less. Slot = line within the wave, 272 B apart (four lines of a 32-lane group 16 B apart).
#ifndef MOF_RAW_STAGE
#define MOF_RAW_STAGE 1
#endif
constexpr int RAW_PITCH = 272;
__device__ __forceinline__ unsigned char* raw_area(cf* z, int line0) { return reinterpret_cast<unsigned char*>(z + za(line0, 0)); }
__device__ __forceinline__ void raw_store8(cf* z, int line0, int slot, int chunk, const uint32_t* c, const uint32_t* pv) {
  typedef uint32_t u4 __attribute__((ext_vector_type(4)));
  typedef u4 __attribute__((address_space(3))) * lds_u4_ptr;
  u4 d;
  d.x = __builtin_amdgcn_perm(pv[0], c[0], 0x05010400u);
  d.y = __builtin_amdgcn_perm(pv[0], c[0], 0x07030602u);
  d.z = __builtin_amdgcn_perm(pv[1], c[1], 0x05010400u);
  d.w = __builtin_amdgcn_perm(pv[1], c[1], 0x07030602u);
  *(lds_u4_ptr)(raw_area(z, line0) + slot * RAW_PITCH + chunk * 16) = d;
}

// ---- row pass over lines [line0, line0 + 8) that are < nlines (wave-local) ----------------------------
// RAW: stage 1 takes z = cur + i prev from the wave's raw area
template <bool RAW = false>
__device__ __forceinline__ void row_pass(cf* __restrict__ z, int line0, int nlines, int lane, const float* tw) {
  {  // stage 1: radix 15; 8 butterflies per line -> one per lane
    const int line = line0 + lane / R2, x = lane % R2;
    const bool on = line < nlines;
    cf v[R1];
    if (on) {
      if constexpr (RAW) {
        typedef const volatile uint16_t __attribute__((address_space(3))) * lds_u16_ptr;
        const unsigned char* src = raw_area(z, line0) + (lane / R2) * RAW_PITCH + 2 * x;
#pragma unroll
        for (int k = 0; k < R1; ++k) {
          const uint32_t cp = *(lds_u16_ptr)(src + 2 * k * R2);
          v[k] = {(float)(cp & 0xffu), (float)(cp >> 8)};
        }
      } else {
#pragma unroll
        for (int k = 0; k < R1; ++k) v[k] = lds_read(&z[za(line, x + k * R2)]);
      }
      butterfly15(v);
    }
    wave_sync();
    if (on) {
#pragma unroll
      for (int k = 0; k < R1; ++k) z[za(line, x * R1 + k)] = v[k];
    }
    wave_sync();
  }
  {  // stage 2: radix 8; 15 butterflies per line: x = 0..7 of the wave's 8 lines, then x = 8..14 (lane map as in
     // stage 1: 8 consecutive x of 4 lines per 32-lane group -- conflict-free with the pitch above)
    cf v[2][R2];
    int line[2], x[2];
    bool on[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      line[b] = line0 + lane / 8;
      x[b] = 8 * b + lane % 8;
      on[b] = x[b] < R1 && line[b] < nlines;
      if (on[b]) {
        cf t[R2 - 1];
#pragma unroll
        for (int k = 0; k < R2; ++k) {
          v[b][k] = lds_read(&z[za(line[b], x[b] + k * R1)]);
          if (k > 0) t[k - 1] = twiddle(tw, k * x[b]);
        }
        butterfly8_tw(v[b], t);  // twiddle products fused into the first radix-2 layer
      }
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < 2; ++b)
      if (on[b]) {
#pragma unroll
        for (int k = 0; k < R2; ++k) z[za(line[b], x[b] + k * R1)] = v[b][k];
      }
    wave_sync();
  }
}

// ---- forward column pass over columns [col0, col0 + 8) (wave-local) -----------------------------------
__device__ __forceinline__ void col_pass_fwd(cf* __restrict__ z, int col0, int lane, const float* tw) {
  {  // stage 1: lane -> (column lane % 8, x = lane / 8)
    const int col = col0 + lane % LPW, x = lane / LPW;
    cf v[R1];
#pragma unroll
    for (int k = 0; k < R1; ++k) v[k] = lds_read(&z[za(x + k * R2, col)]);
    butterfly15(v);
    wave_sync();
#pragma unroll
    for (int k = 0; k < R1; ++k) z[za(x * R1 + k, col)] = v[k];
    wave_sync();
  }
  {  // stage 2: x = q / 8 in 0..14
    cf v[2][R2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, col = col0 + q % LPW, x = q / LPW;
      if (x < R1) {
        cf t[R2 - 1];
#pragma unroll
        for (int k = 0; k < R2; ++k) {
          v[b][k] = lds_read(&z[za(x + k * R1, col)]);
          if (k > 0) t[k - 1] = twiddle(tw, k * x);
        }
        butterfly8_tw(v[b], t);
      }
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, col = col0 + q % LPW, x = q / LPW;
      if (x < R1) {
#pragma unroll
        for (int k = 0; k < R2; ++k) z[za(x + k * R1, col)] = v[b][k];
      }
    }
    wave_sync();
  }
}

// ---- inverse column pass on column pairs (col, col + H), col in [col0, col0 + 8) and < H (wave-local);
//      see col_pass_inv in pc_kernel.hip for the data layout --------------------------------------------
template <int PK>
__device__ __forceinline__ Best col_pass_inv(cf* __restrict__ z, int col0, int lane, const float* tw, int search_radius) {
  {  // stage 1
    const int col = col0 + lane % LPW, x = lane / LPW;
    const bool on = col < H;
    cf v[R1];
    if (on) {
#pragma unroll
      for (int k = 0; k < R1; ++k) {
        const int r = x + k * R2;
        const int rr = (r == 0 || r == H) ? 0 : (r < H ? r : N - r);
        const cf a = lds_read(&z[za(rr, col)]), c = lds_read(&z[za(rr, col + H)]);
        cf e;
        if (r == 0) e = {a.x, c.x};
        else if (r == H) e = {a.y, c.y};
        else if (r < H) e = {a.x - c.y, a.y + c.x};
        else e = {a.x + c.y, c.x - a.y};
        v[k] = e;
      }
      butterfly15(v);
    }
    wave_sync();
    if (on) {
#pragma unroll
      for (int k = 0; k < R1; ++k) z[za(x * R1 + k, col)] = v[k];
    }
    wave_sync();
  }
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  {  // stage 2 + arg-max from registers
    cf v[2][R2];
    bool on[2];
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, col = col0 + q % LPW, x = q / LPW;
      on[b] = x < R1 && col < H;
      if (on[b]) {
        cf t[R2 - 1];
#pragma unroll
        for (int k = 0; k < R2; ++k) {
          v[b][k] = lds_read(&z[za(x + k * R1, col)]);
          if (k > 0) t[k - 1] = twiddle(tw, k * x);
        }
        butterfly8_tw(v[b], t);
        if constexpr (PK == 1) {  // OpenCL-kernel model: 1/N^2 scaling, +-search_radius mask (cl:733, :737-746, :823-826)
#pragma unroll
          for (int k = 0; k < R2; ++k) {
            const int y = x + k * R1;
            v[b][k].x = ocl_scale_mask<N>(v[b][k].x, y, col, search_radius);
            v[b][k].y = ocl_scale_mask<N>(v[b][k].y, y, col + H, search_radius);
          }
        }
      }
    }
    wave_sync();
#pragma unroll
    for (int b = 0; b < 2; ++b) {
      const int q = lane + 64 * b, col = col0 + q % LPW, x = q / LPW;
      if (on[b]) {
#pragma unroll
        for (int k = 0; k < R2; ++k) {
          const int y = x + k * R1;
          z[za(y, col)] = v[b][k];
          const int ys = (y + H) % N;
          best = better(best, Best{v[b][k].x, ys * N + col + H});
          best = better(best, Best{v[b][k].y, ys * N + col});
        }
      }
    }
  }
  return best;
}

}  // namespace

template <int DS, int CH, int PK>
__global__ void __launch_bounds__(T) pc_field_kernel_120(PcArgs a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  Best* red = reinterpret_cast<Best*>(z + N * PITCH);
  int* const_code = reinterpret_cast<int*>(red + 32);  // [16] per-wave constant-patch codes (cur | prev << 16), then C_dc

  const int tid0 = threadIdx.x, lane0 = tid0 & 63, wave0 = tid0 >> 6;
  const int patches = a.grid_x * a.grid_y;
#if MOF_TW120_LDS
  // r05: the second-stage twiddles W_120^{kx} (14 per lane and pass) were gathers from the L1-resident global table -- 64 lanes, up
  // to 8 cache lines per instruction through the texture addresser, ~1260 of them per patch and CU; one LDS copy per persistent
  // workgroup turns them into ds_read_b64
  cf* tw_lds = reinterpret_cast<cf*>(reinterpret_cast<int*>(red + 32) + 32);
  for (int k = tid0; k < N; k += T) tw_lds[k] = {a.twiddles[2 * k], a.twiddles[2 * k + 1]};
  __syncthreads();
  const float* tw = reinterpret_cast<const float*>(tw_lds);
#else
  const float* tw = a.twiddles;
#endif
  auto patch_ptr = [&](int pp, const uint8_t* frames, size_t frame_stride) -> const uint8_t* {
    const int pair = pp / patches, patch = pp % patches;
    const int x0 = a.origin_x + (patch % a.grid_x) * a.stride_x, y0 = a.origin_y + (patch / a.grid_x) * a.stride_y;
    return frames + (size_t)pair * frame_stride + (size_t)(DS * y0) * a.pitch + (size_t)(CH * DS * x0);
  };

  // ---- persistent workgroup (one per CU: the tile is 128 KB): patches p = blockIdx.x, + gridDim.x, ... With a single
  //      workgroup on the CU nothing else hides the HBM latency of the patch load, so the raw pixels of the NEXT patch
  //      are requested before the current one is transformed (full-resolution paths; 8 or 24 VGPRs).
  constexpr int PFW = (DS == 1) ? (CH == 1 ? 2 : 6) : 1;  // dwords per image and chunk held in flight
  uint32_t pfc[2][PFW], pfp[2][PFW];
  auto prefetch = [&](int pp) {
    if constexpr (DS == 1) {
      const uint8_t* cb = patch_ptr(pp, a.cur, a.cur_stride);
      const uint8_t* pb = patch_ptr(pp, a.prev, a.prev_stride);
#pragma unroll
      for (int b = 0; b < 2; ++b) {
        const int q = lane0 + 64 * b;
        if (q < LPW * (N / 8)) {
          const size_t off = (size_t)(wave0 * LPW + q / (N / 8)) * a.pitch + (size_t)CH * ((q % (N / 8)) * 8);
          __builtin_memcpy(pfc[b], cb + off, 4 * PFW);
          __builtin_memcpy(pfp[b], pb + off, 4 * PFW);
        }
      }
    }
  };
  int p = blockIdx.x;
  if (p < a.total) prefetch(p);
  for (; p < a.total; p += gridDim.x) {
  // lane / wave laundered once per patch: keeps LICM from hoisting every LDS address of the body out of the loop
  int lane = lane0, wave = wave0;
  asm volatile("" : "+v"(lane), "+v"(wave));
  lane &= 63;
  wave &= 15;
  const int tid = wave * 64 + lane;
  const uint8_t* cur = patch_ptr(p, a.cur, a.cur_stride);
  const uint8_t* prev = patch_ptr(p, a.prev, a.prev_stride);
  (void)cur; (void)prev;

  // ---- load: the wave's own 8 rows in 8-pixel chunks (15 per row), u8 -> f32, z = cur + i*prev (:1805-1806)
  uint32_t fc = 0, dc = 0, fp = 0, dp = 0;  // constant-patch tracking (pc_common.hpp)
  bool maybe_c = true, maybe_p = true;
#pragma unroll
  for (int b = 0; b < 2; ++b) {
    const int q = lane + 64 * b;
    if (q < LPW * (N / 8)) {
      const int row = wave * LPW + q / (N / 8), col = (q % (N / 8)) * 8;
      uint32_t c[2], pv[2];
      if constexpr (DS == 1) {
        if constexpr (CH == 1) {
          c[0] = pfc[b][0]; c[1] = pfc[b][1];
          pv[0] = pfp[b][0]; pv[1] = pfp[b][1];
        } else {  // BGR8 front end (optic_flow.cpp:1622): 8 pixels = 24 bytes = 6 dwords
          auto byte_of = [](const uint32_t* w, int i) -> uint32_t { return (w[i >> 2] >> (8 * (i & 3))) & 0xffu; };
          c[0] = c[1] = pv[0] = pv[1] = 0;
#pragma unroll
          for (int i = 0; i < 8; ++i) {
            c[i >> 2] |= rgb2gray_fixed(byte_of(pfc[b], 3 * i), byte_of(pfc[b], 3 * i + 1), byte_of(pfc[b], 3 * i + 2)) << (8 * (i & 3));
            pv[i >> 2] |= rgb2gray_fixed(byte_of(pfp[b], 3 * i), byte_of(pfp[b], 3 * i + 1), byte_of(pfp[b], 3 * i + 2)) << (8 * (i & 3));
          }
        }
      } else {
        // long-range mode: rounded mean of the 2x2 centre of each 4x4 cell (cv::resize 1/4, FftMethod.cpp:1931-1932)
        const uint8_t* c1 = cur + (size_t)(4 * row + 1) * a.pitch + 4 * col;
        const uint8_t* p1 = prev + (size_t)(4 * row + 1) * a.pitch + 4 * col;
#pragma unroll
        for (int h2 = 0; h2 < 2; ++h2) {
          uint32_t ca[4], cb[4], pa[4], pb[4];
          __builtin_memcpy(ca, c1 + 16 * h2, 16);
          __builtin_memcpy(cb, c1 + a.pitch + 16 * h2, 16);
          __builtin_memcpy(pa, p1 + 16 * h2, 16);
          __builtin_memcpy(pb, p1 + a.pitch + 16 * h2, 16);
          c[h2] = pv[h2] = 0;
#pragma unroll
          for (int i = 0; i < 4; ++i) {
            const uint32_t cs = ((ca[i] >> 8) & 0xffu) + ((ca[i] >> 16) & 0xffu) + ((cb[i] >> 8) & 0xffu) + ((cb[i] >> 16) & 0xffu);
            const uint32_t ps = ((pa[i] >> 8) & 0xffu) + ((pa[i] >> 16) & 0xffu) + ((pb[i] >> 8) & 0xffu) + ((pb[i] >> 16) & 0xffu);
            c[h2] |= ((cs + 2u) >> 2) << (8 * i);
            pv[h2] |= ((ps + 2u) >> 2) << (8 * i);
          }
        }
      }
      if (b == 0) {  // pre-test: a textured patch has a lane whose two dwords differ -- two compares and it is out
        maybe_c = __builtin_amdgcn_ballot_w64(c[0] != c[1]) == 0ull;
        maybe_p = __builtin_amdgcn_ballot_w64(pv[0] != pv[1]) == 0ull;
      }
      if (__builtin_expect(maybe_c, 0)) {  // (the empty asm keeps the rare branch from being if-converted)
        asm volatile("");
        const_track(c, 2, b == 0, fc, dc);
      }
      if (__builtin_expect(maybe_p, 0)) {
        asm volatile("");
        const_track(pv, 2, b == 0, fp, dp);
      }
      if constexpr (MOF_RAW_STAGE) {
        raw_store8(z, wave * LPW, q / (N / 8), q % (N / 8), c, pv);
        continue;
      }
      // The 8 pixels of a chunk are stored in a per-lane rotated order: straight order puts the 16 lanes of a
      // ds_write_b64 group on two banks (chunks are 8 elements apart), an 8-way conflict on every store.
      const int rot = ((q % (N / 8)) >> 1) & 7;
      const uint64_t cr = __builtin_rotateright64(((uint64_t)c[1] << 32) | c[0], 8 * rot);
      const uint64_t pr = __builtin_rotateright64(((uint64_t)pv[1] << 32) | pv[0], 8 * rot);
#pragma unroll
      for (int i = 0; i < 8; ++i)  // byte i of the rotated word is pixel (i + rot) & 7
        z[za(row, col + ((i + rot) & 7))] = {(float)((uint32_t)(cr >> (8 * i)) & 0xffu), (float)((uint32_t)(pr >> (8 * i)) & 0xffu)};
    }
  }
  if (p + (int)gridDim.x < a.total) prefetch(p + (int)gridDim.x);
  {
    int cc = 256, cp = 256;
    if (__builtin_expect(maybe_c, 0)) {  // (every lane holds a chunk b = 0)
      asm volatile("");
      cc = wave_const_code(true, fc, dc);
    }
    if (__builtin_expect(maybe_p, 0)) {
      asm volatile("");
      cp = wave_const_code(true, fp, dp);
    }
    if (lane == 0) const_code[wave] = cc | (cp << 16);
  }
  wave_sync();

  // ---- forward 2-D transform: rows (wave-local), barrier, columns (wave-local)
  row_pass<MOF_RAW_STAGE != 0>(z, wave * LPW, N, lane, tw);
  __syncthreads();
  col_pass_fwd(z, wave * LPW, lane, tw);
  __syncthreads();

  // ---- normalised cross-power spectrum, half spectrum kept conjugated (see pc_kernel.hip)
  for (int g = tid; g < (H - 1) * N; g += T) {
    const int v = 1 + g / N, u = g % N;
    const cf zk = z[za(v, u)], zm = z[za(N - v, (N - u) % N)];
    const cf C = cross_power<PK>(zk, zm, false);
    z[za(v, u)] = {C.x, -C.y};
  }
  for (int u = tid; u <= H; u += T) {
    const int um = (N - u) % N;
    const bool self = (u == um);
    const cf C0 = cross_power<PK>(z[za(0, u)], z[za(0, um)], self);
    const cf Ch = cross_power<PK>(z[za(H, u)], z[za(H, um)], self);
    if (u == 0) *reinterpret_cast<float*>(const_code + 16) = C0.x;  // C_dc: all that is left of a degenerate pair's spectrum
    z[za(0, u)] = {C0.x + Ch.y, Ch.x - C0.y};
    if (!self) z[za(0, um)] = {C0.x - Ch.y, Ch.x + C0.y};
  }
  __syncthreads();

  // ---- Hermitian inverse: rows 0..H-1, then column pairs (idft :1497)
  if (wave * LPW < H) row_pass<false>(z, wave * LPW, H, lane, tw);
  __syncthreads();
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  if (wave * LPW < H) best = col_pass_inv<PK>(z, wave * LPW, lane, tw, a.search_radius);
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();

  // wave 0 reads the centroid window, then a barrier releases the others to overwrite the tile with the next patch
  float wval = 0.f;
  bool degenerate = false;
  if (wave == 0) {
    const int my_code = const_code[lane & 15];
    for (int w = 1; w < WAVES; ++w) best = better(best, red[w]);
    wval = centroid_window_value<N, PK>(best, lane, [&](int ys, int xs) {
      const int y = (ys + H) % N, x = (xs + H) % N;
      const cf s = z[za(y, x % H)];
      return x < H ? s.x : s.y;
    });
    degenerate = const_codes_degenerate<WAVES>(my_code, lane);
  }
  __syncthreads();
  if (wave == 0)
    centroid_gate_store<N, PK>(best, wval, lane, a.max_px_speed_sq, a.out + 2 * (size_t)p, degenerate,
                               degenerate ? *reinterpret_cast<const float*>(const_code + 16) : 0.f);
  }  // persistent loop
}

template <int DS, int CH, int PK>
static hipError_t configure_one_120() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_field_kernel_120<DS, CH, PK>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)LDS_BYTES_120);
}

hipError_t pc_configure_120() {
  hipError_t e;
  if ((e = configure_one_120<1, 1, 0>()) != hipSuccess) return e;
  if ((e = configure_one_120<1, 3, 0>()) != hipSuccess) return e;
  if ((e = configure_one_120<4, 1, 0>()) != hipSuccess) return e;
  if ((e = configure_one_120<1, 1, 1>()) != hipSuccess) return e;
  if ((e = configure_one_120<1, 3, 1>()) != hipSuccess) return e;
  return configure_one_120<4, 1, 1>();
}

static int g_cu_count_120 = 0;

hipError_t launch_pc_field_120(const PcArgs& a_in, int n_pairs, hipStream_t stream) {
  PcArgs a = a_in;
  a.total = n_pairs * a.grid_x * a.grid_y;
  if (g_cu_count_120 == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    g_cu_count_120 = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess)
                         ? prop.multiProcessorCount : 256;
  }
  const unsigned blocks = (unsigned)(a.total < g_cu_count_120 ? a.total : g_cu_count_120);  // one resident workgroup per CU
  if (a.downscale == 4 && a.channels == 3) return hipErrorInvalidValue;
  const dim3 g(blocks), b(T);
  if (a.peak_model == 1) {
    if (a.downscale == 4) hipLaunchKernelGGL((pc_field_kernel_120<4, 1, 1>), g, b, LDS_BYTES_120, stream, a);
    else if (a.channels == 3) hipLaunchKernelGGL((pc_field_kernel_120<1, 3, 1>), g, b, LDS_BYTES_120, stream, a);
    else hipLaunchKernelGGL((pc_field_kernel_120<1, 1, 1>), g, b, LDS_BYTES_120, stream, a);
  } else {
    if (a.downscale == 4) hipLaunchKernelGGL((pc_field_kernel_120<4, 1, 0>), g, b, LDS_BYTES_120, stream, a);
    else if (a.channels == 3) hipLaunchKernelGGL((pc_field_kernel_120<1, 3, 0>), g, b, LDS_BYTES_120, stream, a);
    else hipLaunchKernelGGL((pc_field_kernel_120<1, 1, 0>), g, b, LDS_BYTES_120, stream, a);
  }
  return hipGetLastError();
}

}  // namespace mof
