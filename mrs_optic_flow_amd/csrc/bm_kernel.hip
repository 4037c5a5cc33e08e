// bm_kernel.hip -- K2 (SAD block scan) and K3 (per-axis histogram mode) for gfx950 (CDNA4).
//
// K2 comes in two forms:
//  * bm_scan16_kernel<R> (c3's geometry class: 16 x 16 blocks, scan radius 8 or 16): one WAVE per workgroup, the
//    block's pixels and all live y-shift accumulators in registers, one sweep over an LDS strip of the previous frame;
//  * bm_scan_kernel (every other geometry, up to the reference's default 120 x 120 blocks / radius 21): a workgroup
//    stages the (sps+2r)^2 windows and sps^2 blocks of 1..8 neighbouring blocks in LDS; each lane owns one block, one
//    y-shift and FOUR consecutive x-shifts and walks the block with v_qsad_pk_u16_u8, which produces the four
//    byte-SADs of one 4-pixel group against the 8-byte sliding window in a single VALU instruction (16 absolute
//    differences per lane-op, exact integer arithmetic). Packed u16 partial sums are widened to u32 every <=256
//    pixels so no block size can overflow. The arg-min (first occurrence in row-major order) is a 64-bit LDS
//    atomic min over (sad, index) keys.
//
// Replaces
//   BlockMethod::processImage's scan          /root/reference/src/BlockMethod.cpp:43-66
//   OptFlow_C1_D0                             /root/reference/src/FastSpacedBMMethod.cl:4-84
// K3 replaces
//   the histogram mode of BlockMethod.cpp:65-76 and Histogram_C1_D0 (FastSpacedBMMethod.cl:86-169).
//
// Geometry (both classes): current block at (bx*S + r, by*S + r), previous-frame window
// origin (bx*S, by*S), S = sps + step; candidate (xs, ys) in [0, 2r]^2, result (xs, ys) - r.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>

#include <type_traits>

#include "mof_kernels.h"
#include "pc_common.hpp"  // rgb2gray_fixed, gray16_from_bgr48

namespace mof {

namespace {

constexpr int BM_THREADS_MAX = 1024;  // generic scan: one lane per item, whole waves

struct Cand {
  uint32_t sad;
  uint32_t idx;  // ys*D + xs
};
__device__ __forceinline__ Cand first_min(Cand a, Cand b) {
  return (b.sad < a.sad || (b.sad == a.sad && b.idx < a.idx)) ? b : a;
}

__device__ __forceinline__ int div_up(int a, int b) { return (a + b - 1) / b; }

// compile-time loop: guarantees full unrolling whatever the unroller's size thresholds say
template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
  if constexpr (I < N) {
    f(std::integral_constant<int, I>{});
    static_for<I + 1, N>(f);
  }
}

}  // namespace

// LDS layout per block slot (dwords): window rows [WW][WPD], current block [sps][sps/4]; then per slot one 64-bit
// arg-min key and the centre SAD (for the low-contrast rule). A workgroup holds `bpw` neighbouring blocks of one block
// row and has one lane per (block, y-shift, lane group) item, rounded up to whole waves (launch_bm_scan picks XB, bpw
// and the padded window pitch WPD from a small cost model).
// A lane owns one (block, y-shift) and XB consecutive groups of four x-shifts. The XB chains share a sliding register
// window over the previous-frame row -- chain k at step g needs window dwords (g + k, g + k + 1) -- so one step costs
// ONE new window dword and ONE block dword from LDS for XB v_qsad_pk_u16_u8. The row loop runs in chunks of G
// steps with every LDS offset an immediate and the window rotation resolved by the unrolling: the round-1 form spent
// five address / move VALU instructions per v_qsad (24 of every 40 issue cycles) and sat at 38-41 % of the ceiling.
// Exact i / d for i * d < 2^32 with m = ceil(2^32 / d), m = 0 standing for d = 1 (the staging loops index at most
// 40 K dwords).
__device__ __forceinline__ uint32_t fast_div(uint32_t i, uint32_t m) { return m ? __umulhi(i, m) : i; }
__device__ __forceinline__ uint32_t div_magic(uint32_t d) { return d == 1 ? 0u : (uint32_t)((0x100000000ull + d - 1) / d); }

// four gray pixels (packed u8x4) from 4 gray bytes or 12 interleaved BGR bytes, any alignment
__device__ __forceinline__ uint32_t load_gray4(const uint8_t* p, int channels) {
  if (channels == 1) {
    uint32_t v;
    __builtin_memcpy(&v, p, 4);
    return v;
  }
  uint32_t w[3];
  __builtin_memcpy(w, p, 12);
  uint32_t packed = 0;
#pragma unroll
  for (int b = 0; b < 4; ++b) {
    const int i = 3 * b;
    const uint32_t c0 = (w[i >> 2] >> (8 * (i & 3))) & 0xffu;
    const uint32_t c1 = (w[(i + 1) >> 2] >> (8 * ((i + 1) & 3))) & 0xffu;
    const uint32_t c2 = (w[(i + 2) >> 2] >> (8 * ((i + 2) & 3))) & 0xffu;
    packed |= rgb2gray_fixed(c0, c1, c2) << (8 * b);
  }
  return packed;
}

template <int XB, int G>
__global__ void __launch_bounds__(BM_THREADS_MAX) bm_scan_kernel(BmArgs a, int bpw, int groups_per_row, int WPD, int slot_dwords) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int tid = threadIdx.x, nthreads = blockDim.x;
  const int r = a.radius, sps = a.block, S = a.block + a.step, D = 2 * r + 1;
  const int XG = div_up(D, 4);          // x-shift groups of 4
  const int XGL = div_up(XG, XB);       // lanes per (block, y-shift)
  const int WW = sps + 2 * r;           // window width == height in pixels
  const int CPD = sps / 4;              // current-block row pitch in dwords
  unsigned long long* keys = reinterpret_cast<unsigned long long*>(lds + (size_t)bpw * slot_dwords);  // slot_dwords is even
  uint32_t* centre = reinterpret_cast<uint32_t*>(keys + bpw);

  const int blocks = a.grid_x * a.grid_y;
  const int grp = blockIdx.x % groups_per_row;
  const int by = (blockIdx.x / groups_per_row) % a.grid_y;
  const int pair = blockIdx.x / (groups_per_row * a.grid_y);
  const int bx0 = grp * bpw;
  const int nb = (a.grid_x - bx0 < bpw) ? a.grid_x - bx0 : bpw;  // blocks of this workgroup (>= 1)

  // ---- stage windows + blocks in LDS. Lanes run along a row (a power-of-two count >= the row's data dwords, so the
  // index arithmetic is shifts and masks) and step down the rows of all slots; every load is independent, several are
  // in flight per lane. A dword that straddles the window's right edge is loaded ending AT the edge and shifted down.
  // The padding dwords of a window row (look-ahead of the sliding register window) are NOT written: they only feed
  // x-shifts >= D, whose sums are discarded (the four sums of a v_qsad are separate 16-bit fields).
  // (The first r02 form walked a flat index with two divisions per dword: 25 VALU instructions per staged dword, as
  // many instructions as the whole scan at c1's 32 x 32 blocks.)
  {
    const int CH = a.channels;  // 3: BGR8 frames, gray conversion on the way into LDS
    const uint8_t* prev0 = a.prev + (size_t)pair * a.prev_stride + (size_t)(by * S) * a.pitch + (size_t)(bx0 * S) * CH;
    const uint8_t* cur0 = a.cur + (size_t)pair * a.cur_stride + (size_t)(by * S + r) * a.pitch + (size_t)(bx0 * S + r) * CH;
    const int win_dwords = WW * WPD;
    auto stage = [&](const uint8_t* base, int rows, int row_px, int row_pitch_dw, int lds_off) {
      const int dw = (row_px + 3) / 4;  // data dwords of a row
      int xsh = 0;
      while ((1 << xsh) < dw) ++xsh;
      const int xd = tid & ((1 << xsh) - 1), rstep = nthreads >> xsh;
      if (xd >= dw) return;
      const int x = 4 * xd, off = x + 4 <= row_px ? x : row_px - 4;
      const uint32_t sh = 8u * (uint32_t)(x - off);
      int y = tid >> xsh, sl = 0;  // row within the slot, slot
      while (y >= rows) y -= rows, ++sl;
#pragma unroll 2
      for (; sl < nb;) {
        uint32_t v = load_gray4(base + (size_t)y * a.pitch + (size_t)(sl * S + off) * CH, CH);  // any alignment
        v >>= sh;
        lds[sl * slot_dwords + lds_off + y * row_pitch_dw + xd] = v;
        y += rstep;
        while (y >= rows) y -= rows, ++sl;
      }
    };
    stage(prev0, WW, WW, WPD, 0);
    stage(cur0, sps, sps, CPD, win_dwords);
  }
  if (tid < nb) keys[tid] = ~0ull;
  __syncthreads();

  // ---- SAD scan: item = (slot, ys, lane group) -> x-shifts 4 (xl XB) .. 4 (xl XB + XB) - 1 of block bx0 + slot at ys
  const int rows_per_flush = (256 / sps) > 0 ? (256 / sps) : 1;  // rows*sps <= 256 px -> packed u16 sums cannot overflow
  const int per_block = D * XGL;
  const bool c64 = (CPD & 1) == 0 && ((WW * WPD) & 1) == 0;  // block rows 8-byte aligned in LDS
  for (int item = tid; item < nb * per_block; item += nthreads) {
    const int s = item / per_block, rem = item % per_block;
    const int ys = rem / XGL, xg0 = (rem % XGL) * XB;
    const uint32_t* win = lds + (size_t)s * slot_dwords;
    const uint32_t* wrow = win + ys * WPD + xg0;
    const uint32_t* crow = win + WW * WPD;
    uint32_t acc[XB][4];
#pragma unroll
    for (int k = 0; k < XB; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) acc[k][q] = 0u;
    for (int j0 = 0; j0 < sps; j0 += rows_per_flush) {
      uint64_t pk[XB];
#pragma unroll
      for (int k = 0; k < XB; ++k) pk[k] = 0;
      const int j1 = (j0 + rows_per_flush < sps) ? j0 + rows_per_flush : sps;
      for (int j = j0; j < j1; ++j, wrow += WPD, crow += CPD) {
        uint32_t w[G + XB];
#pragma unroll
        for (int k = 0; k < XB; ++k) w[k] = wrow[k];
        int g0 = 0;
        for (; g0 + G <= CPD; g0 += G) {
          uint32_t c[G];
          if constexpr ((G & 1) == 0) {
            if (c64) {  // block row as 8-byte LDS reads (256 B/clk against ds_read_b32's 128)
#pragma unroll
              for (int i = 0; i < G; i += 2) {
                const uint2 t = *reinterpret_cast<const uint2*>(crow + g0 + i);
                c[i] = t.x, c[i + 1] = t.y;
              }
            } else {
#pragma unroll
              for (int i = 0; i < G; ++i) c[i] = crow[g0 + i];
            }
          } else {
#pragma unroll
            for (int i = 0; i < G; ++i) c[i] = crow[g0 + i];
          }
#pragma unroll
          for (int i = 0; i < G; ++i) w[XB + i] = wrow[g0 + XB + i];
#pragma unroll
          for (int i = 0; i < G; ++i)
#pragma unroll
            for (int k = 0; k < XB; ++k)
              pk[k] = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w[i + k + 1] << 32) | w[i + k], c[i], pk[k]);
#pragma unroll
          for (int k = 0; k < XB; ++k) w[k] = w[G + k];
        }
        if (g0 < CPD) {  // ragged tail of the row (CPD % G steps; launch_bm_scan prefers a G that divides CPD)
#pragma unroll
          for (int i = 0; i < G - 1; ++i) {
            if (g0 + i < CPD) {
              const uint32_t c = crow[g0 + i];
              w[XB + i] = wrow[g0 + XB + i];
#pragma unroll
              for (int k = 0; k < XB; ++k)
                pk[k] = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w[i + k + 1] << 32) | w[i + k], c, pk[k]);
            }
          }
        }
      }
#pragma unroll
      for (int k = 0; k < XB; ++k) {
        acc[k][0] += (uint32_t)(pk[k] & 0xffffu);
        acc[k][1] += (uint32_t)((pk[k] >> 16) & 0xffffu);
        acc[k][2] += (uint32_t)((pk[k] >> 32) & 0xffffu);
        acc[k][3] += (uint32_t)(pk[k] >> 48);
      }
    }
    // arg-min, first occurrence in row-major order (BlockMethod.cpp:63; .cl:50-56, :66-73): min over (sad, index) keys
    unsigned long long best = ~0ull;
#pragma unroll
    for (int k = 0; k < XB; ++k)
#pragma unroll
      for (int q = 0; q < 4; ++q) {
        const int xs = 4 * (xg0 + k) + q;
        if (xs < D) {
          const unsigned long long key = ((unsigned long long)acc[k][q] << 32) | (uint32_t)(ys * D + xs);
          best = key < best ? key : best;
          if (ys == r && xs == r) centre[s] = acc[k][q];  // SAD(0, 0), FastSpacedBMMethod.cl:77
        }
      }
    atomicMin(&keys[s], best);
  }
  __syncthreads();
  if (tid < nb) {
    const unsigned long long k = keys[tid];
    const uint32_t idx = (uint32_t)k, best_sad = (uint32_t)(k >> 32);
    int mx = (int)(idx % (uint32_t)D), my = (int)(idx / (uint32_t)D);
    // low-contrast rule (FastSpacedBMMethod.cl:2, :77-82): int difference vs double threshold
    if (a.low_contrast_rule) {
      const int diff = (int)centre[tid] - (int)best_sad;
      if ((double)diff <= (double)(r * r) * 0.2) {
        mx = r;
        my = r;
      }
    }
    const size_t o = (size_t)pair * blocks + (size_t)by * a.grid_x + bx0 + tid;
    a.dx[o] = (int8_t)(mx - r);
    a.dy[o] = (int8_t)(my - r);
  }
}

// K3: one workgroup per pair: per-axis histogram of the block shifts, first-maximum mode and
// the next two entries of the stable descending order (TestDepth = 3, FastSpacedBMMethod_OCL.cpp:97).
__global__ void __launch_bounds__(256) bm_mode_kernel(BmArgs a) {
  __shared__ int hist[2][128];
  const int tid = threadIdx.x;
  const int pair = blockIdx.x;
  const int blocks = a.grid_x * a.grid_y;
  const int r = a.radius, D = 2 * r + 1;
  if (tid < 128) {
    hist[0][tid] = 0;
    hist[1][tid] = 0;
  }
  __syncthreads();
  const int8_t* dx = a.dx + (size_t)pair * blocks;
  const int8_t* dy = a.dy + (size_t)pair * blocks;
  for (int i = tid; i < blocks; i += 256) {
    atomicAdd(&hist[0][(int)dx[i] + r], 1);
    atomicAdd(&hist[1][(int)dy[i] + r], 1);
  }
  __syncthreads();
  if (tid < 2) {
    int* h = hist[tid];
    int8_t* out = a.mode + (size_t)pair * 8;
    for (int rank = 0; rank < 3; ++rank) {
      int bi = -1;
      for (int i = 0; i < D; ++i)
        if (h[i] >= 0 && (bi < 0 || h[i] > h[bi])) bi = i;
      out[2 * rank + tid] = (bi >= 0) ? (int8_t)(bi - r) : (int8_t)0;
      if (bi >= 0) h[bi] = -1;  // taken
    }
    out[6 + tid] = 0;
  }
}


// ------------------------------------------------------------------------------------------------
// K2 fast path: 16x16 blocks, one workgroup per ROW of blocks, one lane per (block, group of 4
// x-shifts). The lane keeps its block's 16x16 current pixels (64 VGPRs) and ALL its 2r+1 y-shift
// accumulators (packed 4 x u16, exact: 255*256 < 65536) in registers and sweeps the previous-frame
// window once, row by row: each window row (5 dwords from the LDS strip) feeds up to 16 (y-shift,
// block-row) pairs x 4 v_qsad_pk_u16_u8. LDS traffic drops to 5 dwords per 64 SAD instructions and the
// VALU stream is >80 % v_qsad. The strip of the previous frame shared by the whole block row is staged
// in LDS once (windows of neighbouring blocks overlap by 50 %).
// ------------------------------------------------------------------------------------------------
template <int R>
__global__ void __launch_bounds__(64) bm_scan16_kernel(BmArgs a, int strip_dwords, int bpw, int waves_per_row) {
  // ONE wave per workgroup: `bpw` consecutive blocks of one block row; no workgroup barrier anywhere.
  // 2R+1 = 4*(R/2) + 1 x-shifts: R/2 lanes per block take four each through v_qsad, the last x-shift (2R) is
  // shared out over the same lanes by y-shift and done with v_sad_u8 (same 1 byte-difference / lane / cycle rate)
  static_assert(R % 2 == 0, "fast path needs an even scan radius");
  constexpr int SPS = 16, D = 2 * R + 1, XG = R / 2, WW = SPS + 2 * R;
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int lane = threadIdx.x;
  const int S = SPS + a.step;
  const int gx = a.grid_x;
  const int wr = blockIdx.x % waves_per_row;
  const int by = (blockIdx.x / waves_per_row) % a.grid_y;
  const int pair = blockIdx.x / (waves_per_row * a.grid_y);
  const int b0 = wr * bpw;                               // first block of this wave
  const int nb = (gx - b0 < bpw) ? gx - b0 : bpw;        // blocks of this wave (>= 1)
  uint32_t* strip = lds;                                 // [WW][strip_dwords]
  uint32_t* keys = strip + WW * strip_dwords;            // [64]
  const int CH = a.channels;  // 3: BGR8 frames, gray conversion inside the loads
  const uint8_t* prev = a.prev + (size_t)pair * a.prev_stride + (size_t)(by * S) * a.pitch + (size_t)(b0 * S) * CH;
  const uint8_t* cur = a.cur + (size_t)pair * a.cur_stride + (size_t)(by * S + R) * a.pitch + (size_t)(R + b0 * S) * CH;
  const int strip_w = (nb - 1) * S + WW;  // bytes of the frame rows this wave needs

  const int b = lane / XG, xg = lane % XG;      // block within the wave, group of four x-shifts
  const bool active = b < nb;
  const int bc = active ? b : nb - 1;           // idle lanes shadow the last block (never stored)

  // the block's current pixels: 16 rows x 16 B straight from global memory into registers (issued first)
  uint32_t cb[SPS][4];
#pragma unroll
  for (int j = 0; j < SPS; ++j) {
    if (CH == 1) {
      typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
      u32x4 t;
      __builtin_memcpy(&t, cur + (size_t)j * a.pitch + bc * S, 16);  // one global_load_dwordx4, any alignment
      cb[j][0] = t.x;
      cb[j][1] = t.y;
      cb[j][2] = t.z;
      cb[j][3] = t.w;
    } else {
      gray16_from_bgr48(cur + (size_t)j * a.pitch + (size_t)(bc * S) * 3, cb[j]);
    }
  }

  // ---- stage the strip: 16-byte loads at any byte alignment (global_load_dwordx4 -> ds_write_b128);
  //      strip_dwords is a multiple of 4; bytes beyond the strip are zero and never loaded: the one chunk per row that
  //      straddles the strip's right edge is loaded ENDING at the edge and shifted down by whole bytes (strip_w >= 48).
  //      (The first form walked that chunk byte by byte -- every wave iteration holds a few such lanes, so all of them
  //      paid 16 predicated byte loads: 1300 of the wave's 3800 VALU instructions.)
  {
    const int chunks = strip_dwords / 4;
    const uint32_t m_chunks = div_magic((uint32_t)chunks);
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
#pragma unroll 2
    for (int i = lane; i < WW * chunks; i += 64) {
      const int y = (int)fast_div((uint32_t)i, m_chunks), cx = i - y * chunks, x = 16 * cx;
      u32x4 v = {0u, 0u, 0u, 0u};
      if (x < strip_w) {
        const int off = x + 16 <= strip_w ? x : strip_w - 16;
        if (CH == 1) {
          __builtin_memcpy(&v, prev + (size_t)y * a.pitch + off, 16);
        } else {
          uint32_t g4[4];
          gray16_from_bgr48(prev + (size_t)y * a.pitch + (size_t)off * 3, g4);
          v = u32x4{g4[0], g4[1], g4[2], g4[3]};
        }
        const int d = x - off;  // 0, or 1..15 bytes for the edge chunk
        if (d) {
          const int q = d >> 2;
          const uint32_t b = 8u * (uint32_t)(d & 3);
          uint32_t t[8] = {v.x, v.y, v.z, v.w, 0u, 0u, 0u, 0u};
          uint32_t u[5];
#pragma unroll
          for (int k = 0; k < 5; ++k) u[k] = q == 0 ? t[k] : q == 1 ? t[k + 1] : q == 2 ? t[k + 2] : t[k + 3];
          v.x = (uint32_t)((((uint64_t)u[1] << 32) | u[0]) >> b);
          v.y = (uint32_t)((((uint64_t)u[2] << 32) | u[1]) >> b);
          v.z = (uint32_t)((((uint64_t)u[3] << 32) | u[2]) >> b);
          v.w = (uint32_t)((((uint64_t)u[4] << 32) | u[3]) >> b);
        }
      }
      *reinterpret_cast<u32x4*>(strip + (size_t)y * strip_dwords + 4 * cx) = v;
    }
  }
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");

  uint64_t acc[D];
#pragma unroll
  for (int ys = 0; ys < D; ++ys) acc[ys] = 0;
  const uint32_t* wbase = strip + (bc * S) / 4 + xg;
  // straight-line sweep: wy, j, g are compile-time, so acc[] and cb[][] stay in registers. A y-shift is
  // complete once window row ys + 15 has been consumed: it is folded into the running first-minimum key
  // ((sad << 16) | row-major index) right away, so at most 16 accumulators are live.
  uint32_t kmin = 0xffffffffu, centre = 0;
  static_for<0, WW>([&](auto wy_c) {
    constexpr int wy = decltype(wy_c)::value;
    uint32_t w[5];
#pragma unroll
    for (int g = 0; g < 5; ++g) w[g] = wbase[wy * strip_dwords + g];
    static_for<0, SPS>([&](auto j_c) {
      constexpr int j = decltype(j_c)::value;
      constexpr int ys = wy - j;
      if constexpr (ys >= 0 && ys < D) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
          acc[ys] = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)w[g + 1] << 32) | w[g], cb[j][g], acc[ys]);
      }
    });
    constexpr int done = wy - (SPS - 1);
    if constexpr (done >= 0 && done < D) {
      // key = (sad << 16) | index: v_lshl_or_b32 for the low halves, v_and_or_b32 for the high halves
      const uint32_t lo = (uint32_t)acc[done], hi = (uint32_t)(acc[done] >> 32);
      const uint32_t idx0 = (uint32_t)(done * D) + 4u * (uint32_t)xg;
      uint32_t k0 = (lo << 16) | idx0;
      uint32_t k1 = (lo & 0xffff0000u) | (idx0 + 1u);
      uint32_t k2 = (hi << 16) | (idx0 + 2u);
      uint32_t k3 = (hi & 0xffff0000u) | (idx0 + 3u);
      kmin = min(kmin, min(min(k0, k1), min(k2, k3)));
      if constexpr (done == R) centre = (uint32_t)(acc[R] >> (16 * (R % 4))) & 0xffffu;
    }
    // keep the scheduler from hoisting every row's LDS reads to the top (it would spill ~130 VGPRs)
    __builtin_amdgcn_sched_barrier(0);
  });

  // ---- the last x-shift (xs = 2R): lane xg of the block takes y-shifts xg, xg + XG, ...
  for (int ys = xg; ys < D; ys += XG) {
    uint32_t acc1 = 0;
    const uint32_t* wcol = strip + (bc * S) / 4 + (2 * R) / 4 + ys * strip_dwords;
#pragma unroll
    for (int j = 0; j < SPS; ++j) {
#pragma unroll
      for (int g = 0; g < 4; ++g) acc1 = __builtin_amdgcn_sad_u8(wcol[j * strip_dwords + g], cb[j][g], acc1);
    }
    const uint32_t key = (acc1 << 16) | (uint32_t)(ys * D + 2 * R);
    kmin = min(kmin, key);
  }

  // ---- per block: first minimum over its XG lanes (wave-local through LDS), low-contrast rule, store
  keys[lane] = kmin;
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
  // the lane holding the centre SAD (x-shift group R/4, y-shift R) finishes its block
  if (active && xg == R / 4) {
    uint32_t km = 0xffffffffu;
#pragma unroll
    for (int i = 0; i < XG; ++i) km = min(km, keys[b * XG + i]);
    const int idx = (int)(km & 0xffffu), best = (int)(km >> 16);
    int mx = idx % D, my = idx / D;
    if (a.low_contrast_rule && (double)((int)centre - best) <= (double)(R * R) * 0.2) {
      mx = R;
      my = R;
    }
    const size_t o = (size_t)pair * (gx * a.grid_y) + (size_t)by * gx + b0 + b;
    a.dx[o] = (int8_t)(mx - R);
    a.dy[o] = (int8_t)(my - R);
  }
}

template <int R>
static void scan16_plan(const BmArgs& a, int* bpw, int* waves_per_row, int* strip_dwords, size_t* lds) {
  constexpr int XG = R / 2, WW = 16 + 2 * R;
  const int S = 16 + a.step;
  const int cap = 64 / XG;                                 // blocks a wave can hold
  *waves_per_row = (a.grid_x + cap - 1) / cap;
  *bpw = (a.grid_x + *waves_per_row - 1) / *waves_per_row; // balanced
  const int strip_w = (*bpw - 1) * S + WW;
  *strip_dwords = (((strip_w + 3) / 4 + 1) + 3) & ~3;      // + over-read dword, rounded to 16 B
  *lds = sizeof(uint32_t) * ((size_t)WW * *strip_dwords + 64);
}

template <int R>
static hipError_t launch_scan16(const BmArgs& a, int n_pairs, hipStream_t stream) {
  int bpw, wpr, sd;
  size_t lds;
  scan16_plan<R>(a, &bpw, &wpr, &sd, &lds);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  hipLaunchKernelGGL(bm_scan16_kernel<R>, dim3((unsigned)n_pairs * a.grid_y * wpr), dim3(64), lds, stream, a, sd, bpw, wpr);
  return hipGetLastError();
}

// r06: every even radius up to 16 (tools/bm_size_probe.py: radius 12 ran 3 x slower than radius 16 on the generic kernel -- for less work)
template <class F>
static bool scan16_dispatch(int radius, F&& f) {
  switch (radius) {
    case 2: f(std::integral_constant<int, 2>{}); return true;
    case 4: f(std::integral_constant<int, 4>{}); return true;
    case 6: f(std::integral_constant<int, 6>{}); return true;
    case 8: f(std::integral_constant<int, 8>{}); return true;
    case 10: f(std::integral_constant<int, 10>{}); return true;
    case 12: f(std::integral_constant<int, 12>{}); return true;
    case 14: f(std::integral_constant<int, 14>{}); return true;
    case 16: f(std::integral_constant<int, 16>{}); return true;
    default: return false;
  }
}
static bool fast16_ok(const BmArgs& a) {
  if (a.block != 16 || (a.step % 4) != 0) return false;
  int bpw, wpr, sd;
  size_t lds = 0;
  if (!scan16_dispatch(a.radius, [&](auto r) { scan16_plan<decltype(r)::value>(a, &bpw, &wpr, &sd, &lds); })) return false;
  return lds <= 64 * 1024;
}

// ------------------------------------------------------------------------------------------------
// K9/K10: BlockMethod::Refine (/root/reference/src/BlockMethod.cpp:96-147)
// K9: cv::resize(src, dst, 2x) for CV_8UC1, INTER_LINEAR, in OpenCV's fixed point (coefficients 512/1536 of 2048,
//     vertical pass ((b*(S>>4))>>16 ... + 2) >> 2; weights reset at the left/right border, row indices clipped
//     at the top/bottom) -- one thread per destination pixel, integer, bit-exact against the oracle.
// K10: the nine SADs of one refinement pass between the cut-out of A at (1,1) and B at (spx+n, spy+m), n,m in
//     {-1,0,1}: per-thread partial sums, wave shuffle + LDS reduction, one 64-bit atomic per workgroup and shift.
// ------------------------------------------------------------------------------------------------
__global__ void __launch_bounds__(256) bm_resize2x_kernel(const uint8_t* __restrict__ src, size_t pitch, int w, int h,
                                                          uint8_t* __restrict__ dst) {
  const int dw = 2 * w, dh = 2 * h;
  const int dx = blockIdx.x * 256 + threadIdx.x, dy = blockIdx.y;
  if (dx >= dw || dy >= dh) return;
  int sx = (dx >> 1) - ((dx & 1) ? 0 : 1);
  int a0 = (dx & 1) ? 1536 : 512, a1 = 2048 - a0;
  if (sx < 0) { sx = 0; a0 = 2048; a1 = 0; }
  int sx1 = sx + 1;
  if (sx1 >= w) { sx = w - 1; sx1 = w - 1; a0 = 2048; a1 = 0; }
  int sy = (dy >> 1) - ((dy & 1) ? 0 : 1);
  const int b0 = (dy & 1) ? 1536 : 512, b1 = 2048 - b0;
  int sy1 = sy + 1;
  if (sy < 0) sy = 0;
  if (sy1 > h - 1) sy1 = h - 1;
  const uint8_t* r0 = src + (size_t)sy * pitch;
  const uint8_t* r1 = src + (size_t)sy1 * pitch;
  const int S0 = (int)r0[sx] * a0 + (int)r0[sx1] * a1;
  const int S1 = (int)r1[sx] * a0 + (int)r1[sx1] * a1;
  dst[(size_t)dy * dw + dx] = (uint8_t)((((b0 * (S0 >> 4)) >> 16) + ((b1 * (S1 >> 4)) >> 16) + 2) >> 2);
}

__global__ void __launch_bounds__(256) bm_refine_sad_kernel(const uint8_t* __restrict__ A, const uint8_t* __restrict__ B,
                                                            int W2, int spx, int spy, int cw, int ch,
                                                            unsigned long long* __restrict__ out9) {
  __shared__ unsigned int red[4][9];
  const int tid = threadIdx.x;
  unsigned int acc[9];
#pragma unroll
  for (int k = 0; k < 9; ++k) acc[k] = 0;
  // 16 cut-out rows per workgroup; a thread's partial sum stays below 2^32 (16 * ceil(cw/256) * 255)
  const int y0 = blockIdx.x * 16;
  for (int y = y0; y < y0 + 16 && y < ch; ++y) {
    const uint8_t* a = A + (size_t)(1 + y) * W2 + 1;
    for (int x = tid; x < cw; x += 256) {
      const int av = a[x];
#pragma unroll
      for (int m = 0; m < 3; ++m) {
        const uint8_t* b = B + (size_t)(spy + m - 1 + y) * W2 + spx - 1 + x;
#pragma unroll
        for (int n = 0; n < 3; ++n) {
          const int d = av - (int)b[n];
          acc[m * 3 + n] += (unsigned int)(d < 0 ? -d : d);
        }
      }
    }
  }
#pragma unroll
  for (int k = 0; k < 9; ++k) {
    unsigned int v = acc[k];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += (unsigned int)__shfl_xor((int)v, off, 64);
    if ((tid & 63) == 0) red[tid >> 6][k] = v;
  }
  __syncthreads();
  if (tid < 9) atomicAdd(&out9[tid], (unsigned long long)red[0][tid] + red[1][tid] + red[2][tid] + red[3][tid]);
}

hipError_t launch_bm_resize2x(const uint8_t* src, size_t pitch, int w, int h, uint8_t* dst, hipStream_t stream) {
  hipLaunchKernelGGL(bm_resize2x_kernel, dim3((unsigned)((2 * w + 255) / 256), (unsigned)(2 * h)), dim3(256), 0, stream, src,
                     pitch, w, h, dst);
  return hipGetLastError();
}

hipError_t launch_bm_refine_sad(const uint8_t* A, const uint8_t* B, int W2, int spx, int spy, int cw, int ch,
                                unsigned long long* out9, hipStream_t stream) {
  hipLaunchKernelGGL(bm_refine_sad_kernel, dim3((unsigned)((ch + 15) / 16)), dim3(256), 0, stream, A, B, W2, spx, spy, cw, ch, out9);
  return hipGetLastError();
}

constexpr size_t BM_LDS_MAX = 160 * 1024;  // LDS of one CU on gfx950
constexpr int BM_XB_MAX = 3;

// Launch shape of the generic scan for one geometry.
struct BmPlan {
  int xb = 0, g = 8, bpw = 0, wpd = 0, slot_dwords = 0, threads = 0;
  size_t lds = 0;
  double cost = 1e30;  // modelled issue cycles per useful v_qsad (16 = the instruction's own rate)
};

static int bm_env_int(const char* name) {
  const char* e = getenv(name);
  return e ? atoi(e) : 0;
}

// LDS cycles of one window ds_read_b32 of a wave (two 32-lane groups, 32 banks; (ds_read_b32: 2 cycles conflict-free)
static double bm_window_read_cycles(int D, int XGL, int xb, int wpd, int slot, int items) {
  int total = 0, groups = 0;
  for (int g0 = 0; g0 < items && g0 < 1024; g0 += 32, ++groups) {
    int cnt[32] = {0}, worst = 0;
    for (int l = 0; l < 32 && g0 + l < items; ++l) {
      const int rem = (g0 + l) % (D * XGL);
      const int bank = (((g0 + l) / (D * XGL)) * slot + (rem / XGL) * wpd + (rem % XGL) * xb) & 31;
      worst = ++cnt[bank] > worst ? cnt[bank] : worst;
    }
    total += worst;
  }
  return groups ? 2.0 * total / groups : 2.0;
}

static BmPlan bm_plan_search(int block, int radius, int grid_x) {
  const int D = 2 * radius + 1, XG = (D + 3) / 4, WW = block + 2 * radius, CPD = block / 4;
  const int forced_xb = bm_env_int("MOF_BM_XB"), forced_bpw = bm_env_int("MOF_BM_BPW");
  BmPlan best;
  for (int xb = 1; xb <= BM_XB_MAX; ++xb) {
    if (forced_xb && xb != forced_xb) continue;
    const int XGL = (XG + xb - 1) / xb, per_block = D * XGL;
    for (int pad = 0; pad < 8; ++pad) {
      const int wpd = XGL * xb + CPD + 1 + pad;
      const int slot = (WW * wpd + block * CPD + 1) & ~1;  // even: 8-byte LDS reads of the block rows, 64-bit keys behind
      for (int bpw = 1; bpw <= 16 && bpw <= (grid_x > 0 ? grid_x : 1); ++bpw) {
        if (forced_bpw && grid_x > 0 && bpw != forced_bpw && !(forced_bpw > grid_x && bpw == grid_x)) continue;
        const int items = bpw * per_block;
        const int threads = ((items + 63) / 64) * 64;
        if (threads > BM_THREADS_MAX) break;
        const size_t lds = sizeof(uint32_t) * (size_t)bpw * slot + 12 * (size_t)bpw;
        if (lds > BM_LDS_MAX) break;
        const int waves = threads / 64;
        const int wgs_per_cu = (int)(BM_LDS_MAX / lds) < (32 / waves > 0 ? 32 / waves : 1) ? (int)(BM_LDS_MAX / lds) : (32 / waves > 0 ? 32 / waves : 1);
        const int waves_per_cu = waves * (wgs_per_cu < 1 ? 1 : wgs_per_cu);
        // per row step and wave: xb v_qsad (16 cycles each on its SIMD); one window + one block dword from the CU's LDS
        const double valu = 16.0 * xb;
        // four SIMDs share the LDS; the block row is read 8 bytes at a time when its pitch allows (half the cycles per dword)
        const double ldsc = 4.0 * (bm_window_read_cycles(D, XGL, xb, wpd, slot, items) + ((CPD & 1) == 0 && ((WW * wpd) & 1) == 0 ? 1.0 : 2.0));
        double c = (valu > ldsc ? valu : ldsc) / xb;
        c *= (double)(XGL * xb) / XG;                       // padded x-shift groups
        c *= (double)(4 * XG) / D;                          // padded x-shifts of the last group
        c *= (double)threads / items;                       // idle lanes of the last wave
        const int groups = grid_x > 0 ? (grid_x + bpw - 1) / bpw : 1;
        if (grid_x > 0) c *= (double)(groups * bpw) / grid_x;  // ragged last workgroup of a block row
        if (waves_per_cu < 8) c *= 1.0 + 0.08 * (8 - waves_per_cu);  // little left to hide the staging phase behind
        if (wgs_per_cu <= 1) c *= 1.3;                               // ... and nothing at all with one workgroup per CU
        else if (wgs_per_cu == 2) c *= 1.08;
        c *= 1.0 + 0.015 * waves;  // a workgroup's waves wait for one another around the staging phase: smaller ones overlap better
        c *= 1.0 + 0.002 * pad;
        if (c < best.cost) {
          best.cost = c;
          best.xb = xb, best.bpw = bpw, best.wpd = wpd, best.slot_dwords = slot, best.threads = threads, best.lds = lds;
        }
      }
    }
  }
  return best;
}

// The search walks a few hundred candidates with a bank simulation each (~0.1-1 ms of host time): the last plan of each
// thread is kept, an engine asks for the same geometry on every launch.
static BmPlan bm_plan(int block, int radius, int grid_x) {
  struct Key {
    int block, radius, grid_x;
  };
  static thread_local Key key{-1, -1, -1};
  static thread_local BmPlan plan;
  if (key.block != block || key.radius != radius || key.grid_x != grid_x) {
    plan = bm_plan_search(block, radius, grid_x);
    key = Key{block, radius, grid_x};
  }
  return plan;
}

// Block sizes up to 128 (the reference's default sample_point_size is 120, config/default.yaml:32) and radii up to 48
// (its own limit is 2r + 1 <= 50, FastSpacedBMMethod.cl:1), as long as one block's window fits the LDS.
bool bm_config_supported(int block, int radius) {
  return block >= 4 && block <= 128 && (block % 4) == 0 && radius >= 1 && radius <= 48 && bm_plan(block, radius, 0).xb > 0;
}

template <int XB, int G>
static hipError_t launch_bm_generic(const BmArgs& a, int n_pairs, const BmPlan& p, hipStream_t stream) {
  if (p.lds > 64 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&bm_scan_kernel<XB, G>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)p.lds);
    if (e != hipSuccess) return e;
  }
  const int groups_per_row = (a.grid_x + p.bpw - 1) / p.bpw;
  const unsigned wgs = (unsigned)n_pairs * (unsigned)(groups_per_row * a.grid_y);
  hipLaunchKernelGGL((bm_scan_kernel<XB, G>), dim3(wgs), dim3(p.threads), p.lds, stream, a, p.bpw, groups_per_row, p.wpd, p.slot_dwords);
  return hipGetLastError();
}

template <int XB>
static hipError_t launch_bm_generic_g(const BmArgs& a, int n_pairs, const BmPlan& p, hipStream_t stream) {
  switch (p.g) {
    case 6: return launch_bm_generic<XB, 6>(a, n_pairs, p, stream);
    case 7: return launch_bm_generic<XB, 7>(a, n_pairs, p, stream);
    case 10: return launch_bm_generic<XB, 10>(a, n_pairs, p, stream);
    default: return launch_bm_generic<XB, 8>(a, n_pairs, p, stream);
  }
}

hipError_t launch_bm_scan(const BmArgs& a, int n_pairs, hipStream_t stream) {
  if (fast16_ok(a) && !getenv("MOF_BM_GENERIC")) {
    hipError_t err = hipErrorInvalidValue;
    scan16_dispatch(a.radius, [&](auto r) { err = launch_scan16<decltype(r)::value>(a, n_pairs, stream); });
    return err;
  }
  BmPlan p = bm_plan(a.block, a.radius, a.grid_x);
  if (p.xb == 0) return hipErrorInvalidValue;
  // row chunk: the length in {10, 8, 6, 7} that leaves the shortest ragged tail, longer on ties
  const int CPD = a.block / 4, cand[4] = {10, 8, 6, 7};
  int best_tail = 1 << 30;
  for (int g : cand) {
    if (g > CPD && CPD >= 6) continue;
    const int tail = CPD % g;
    if (tail < best_tail) best_tail = tail, p.g = g;
  }
  if (const int fg = bm_env_int("MOF_BM_G")) p.g = fg;
  if (getenv("MOF_BM_VERBOSE"))
    fprintf(stderr, "mof: block scan plan xb %d g %d bpw %d wpd %d threads %d lds %zu cost %.1f\n", p.xb, p.g, p.bpw, p.wpd, p.threads, p.lds, p.cost);
  switch (p.xb) {
    case 1: return launch_bm_generic_g<1>(a, n_pairs, p, stream);
    case 2: return launch_bm_generic_g<2>(a, n_pairs, p, stream);
    default: return launch_bm_generic_g<3>(a, n_pairs, p, stream);
  }
}

hipError_t launch_bm_mode(const BmArgs& a, int n_pairs, hipStream_t stream) {
  hipLaunchKernelGGL(bm_mode_kernel, dim3((unsigned)n_pairs), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mof
