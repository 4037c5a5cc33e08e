// bm_kernel.hip -- K2 (SAD block scan) and K3 (per-axis histogram mode) for gfx950 (CDNA4).
//
// K2: one workgroup per block. The (sps+2r)^2 search window of the previous frame and the
// sps^2 current block are staged once in LDS; each lane then owns one y-shift and FOUR
// consecutive x-shifts and walks the block with v_qsad_pk_u16_u8, which produces the four
// byte-SADs of one 4-pixel group against the 8-byte sliding window in a single VALU
// instruction (16 absolute differences per lane-op, exact integer arithmetic). Packed u16
// partial sums are widened to u32 every <=256 pixels so no block size can overflow.
// The arg-min (first occurrence in row-major order) is a wave-shuffle + LDS reduction.
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

#include "mof_kernels.h"

namespace mof {

namespace {

constexpr int BM_THREADS = 256;

struct Cand {
  uint32_t sad;
  uint32_t idx;  // ys*D + xs
};
__device__ __forceinline__ Cand first_min(Cand a, Cand b) {
  return (b.sad < a.sad || (b.sad == a.sad && b.idx < a.idx)) ? b : a;
}

__device__ __forceinline__ int div_up(int a, int b) { return (a + b - 1) / b; }

}  // namespace

// LDS layout (dwords): window rows [WH][WPD], current block [sps][sps/4], SAD table [D][4*XG],
// reduction scratch.
__global__ void __launch_bounds__(BM_THREADS) bm_scan_kernel(BmArgs a) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lds[];
  const int tid = threadIdx.x;
  const int r = a.radius, sps = a.block, S = a.block + a.step, D = 2 * r + 1;
  const int XG = div_up(D, 4);          // x-shift groups of 4
  const int WW = sps + 2 * r;           // window width == height in pixels
  const int WPD = XG + sps / 4 + 1;     // window row pitch in dwords (covers the 8-byte over-read)
  const int CPD = sps / 4;              // current-block row pitch in dwords
  uint32_t* win = lds;
  uint32_t* blk = win + WW * WPD;
  uint32_t* sad = blk + sps * CPD;
  Cand* red = reinterpret_cast<Cand*>(sad + D * 4 * XG);

  const int blocks = a.grid_x * a.grid_y;
  const int pair = blockIdx.x / blocks;
  const int b = blockIdx.x % blocks;
  const int bx = b % a.grid_x, by = b / a.grid_x;
  const uint8_t* cur = a.cur + (size_t)pair * a.cur_stride + (size_t)(by * S + r) * a.pitch + (bx * S + r);
  const uint8_t* prev = a.prev + (size_t)pair * a.prev_stride + (size_t)(by * S) * a.pitch + bx * S;

  // ---- stage window + block in LDS (bytes beyond the window width are zero, never loaded)
  {
    uint8_t* wb = reinterpret_cast<uint8_t*>(win);
    for (int i = tid; i < WW * WPD * 4; i += BM_THREADS) {
      const int y = i / (WPD * 4), x = i % (WPD * 4);
      wb[i] = (x < WW) ? prev[(size_t)y * a.pitch + x] : (uint8_t)0;
    }
    uint8_t* cb = reinterpret_cast<uint8_t*>(blk);
    for (int i = tid; i < sps * sps; i += BM_THREADS) cb[i] = cur[(size_t)(i / sps) * a.pitch + (i % sps)];
  }
  __syncthreads();

  // ---- SAD scan: item = (ys, xg) -> shifts (4xg..4xg+3, ys)
  const int rows_per_flush = (256 / sps) > 0 ? (256 / sps) : 1;  // rows*sps <= 256 px -> u16 safe
  Cand best = {0xffffffffu, 0xffffffffu};
  for (int item = tid; item < D * XG; item += BM_THREADS) {
    const int ys = item / XG, xg = item % XG;
    uint32_t acc[4] = {0u, 0u, 0u, 0u};
    for (int j0 = 0; j0 < sps; j0 += rows_per_flush) {
      uint64_t pk = 0;
      const int j1 = (j0 + rows_per_flush < sps) ? j0 + rows_per_flush : sps;
      for (int j = j0; j < j1; ++j) {
        const uint32_t* wrow = win + (ys + j) * WPD + xg;
        const uint32_t* crow = blk + j * CPD;
        uint32_t lo = wrow[0];
        for (int g = 0; g < CPD; ++g) {
          const uint32_t hi = wrow[g + 1];
          pk = __builtin_amdgcn_qsad_pk_u16_u8(((uint64_t)hi << 32) | lo, crow[g], pk);
          lo = hi;
        }
      }
      acc[0] += (uint32_t)(pk & 0xffffu);
      acc[1] += (uint32_t)((pk >> 16) & 0xffffu);
      acc[2] += (uint32_t)((pk >> 32) & 0xffffu);
      acc[3] += (uint32_t)(pk >> 48);
    }
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      const int xs = 4 * xg + q;
      sad[ys * 4 * XG + xs] = acc[q];
      if (xs < D) best = first_min(best, Cand{acc[q], (uint32_t)(ys * D + xs)});
    }
  }
  // ---- arg-min, first occurrence in row-major order (BlockMethod.cpp:63; .cl:50-56, :66-73)
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Cand o = {(uint32_t)__shfl_xor((int)best.sad, off, 64), (uint32_t)__shfl_xor((int)best.idx, off, 64)};
    best = first_min(best, o);
  }
  if ((tid & 63) == 0) red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < BM_THREADS / 64; ++w) best = first_min(best, red[w]);
    int mx = (int)(best.idx % (uint32_t)D), my = (int)(best.idx / (uint32_t)D);
    // low-contrast rule (FastSpacedBMMethod.cl:2, :77-82): int difference vs double threshold
    if (a.low_contrast_rule) {
      const int diff = (int)sad[r * 4 * XG + r] - (int)best.sad;
      if ((double)diff <= (double)(r * r) * 0.2) {
        mx = r;
        my = r;
      }
    }
    a.dx[(size_t)pair * blocks + b] = (int8_t)(mx - r);
    a.dy[(size_t)pair * blocks + b] = (int8_t)(my - r);
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

bool bm_config_supported(int block, int radius) {
  return block >= 4 && block <= 64 && (block % 4) == 0 && radius >= 1 && radius <= 48;
}

static size_t bm_lds_bytes(const BmArgs& a) {
  const int D = 2 * a.radius + 1, XG = (D + 3) / 4, WW = a.block + 2 * a.radius;
  const int WPD = XG + a.block / 4 + 1;
  return sizeof(uint32_t) * ((size_t)WW * WPD + (size_t)a.block * (a.block / 4) + (size_t)D * 4 * XG) + 8 * 8;
}

hipError_t launch_bm_scan(const BmArgs& a, int n_pairs, hipStream_t stream) {
  const size_t lds = bm_lds_bytes(a);
  if (lds > 64 * 1024) return hipErrorInvalidValue;
  const unsigned blocks = (unsigned)n_pairs * (unsigned)(a.grid_x * a.grid_y);
  hipLaunchKernelGGL(bm_scan_kernel, dim3(blocks), dim3(BM_THREADS), lds, stream, a);
  return hipGetLastError();
}

hipError_t launch_bm_mode(const BmArgs& a, int n_pairs, hipStream_t stream) {
  hipLaunchKernelGGL(bm_mode_kernel, dim3((unsigned)n_pairs), dim3(256), 0, stream, a);
  return hipGetLastError();
}

}  // namespace mof
