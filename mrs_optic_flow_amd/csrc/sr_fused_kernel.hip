// sr_fused_kernel.hip -- the scale/rotation estimator's row AND column transforms in one kernel (K56), the row part on the
// matrix cores: no row spectra (Zh) in HBM.
//
// cv::phaseCorrelate on the res x res log-polar images (/root/reference/src/scaleRotationEstimator.cpp:117) needs, per pair,
// the 2-D spectra of two real images. K5s (sr_seq_kernel.hip) transforms the rows and writes the half spectra Zh -- 925 KB
// per 480^2 image -- and K6s reads them back line by line for the column pass: 3.7 GB of the 8 GB a 1024-pair pass of BASELINE
// c5 moves, in kernels that sit at 80 % of the achievable HBM rate already. A column group only needs the row spectra of ITS
// bins, for all rows, and that is a dense product:
//   Zh[u][v] = 2 sum_n x[v][n] e^{-2 pi i u n / N}  =  X (N x N, u8)  .  W (N x 2 B: 2 cos | -2 sin of B bins)
// u8 pixels are exact in f16; W is split W = W_hi + W_lo into two f16 matrices (mfma_frag.hpp), both products are exact in the
// f32 accumulator: v_mfma_f32_32x32x16_f16 with A = 32 image rows x 16 pixels, B = 16 pixels x (16 bins x (re | im)).
// A WORKGROUP (4 waves) owns 16 bins of one pair: every wave forms the product for a quarter of the rows (tiles of 32 rows;
// the last tile of a wave overlaps its neighbour when N / 4 is not a multiple of 32), the accumulators go into 16 LDS lines,
// and from there on each wave owns four lines exactly as K6s does: column transforms in LDS (wave_fft), normalised
// cross-power spectrum against the previous frame's spectra held in registers, inverse column transforms, Dt for K7 / K8.
// The image is read once per bin group (16 of them for N = 480) -- from L2, where the remap kernel has just left it.
// The bins u = 0 and u = N/2 get exact integers (weights +-2, 0), so the real-only slots behave as in K5s / K6s.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "mfma_frag.hpp"
#include "mof_kernels.h"
#include "pc_common.hpp"
#include "sr_common.hpp"

namespace mof {

namespace {

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef uint32_t u4_t __attribute__((ext_vector_type(4)));
typedef uint32_t u2_t __attribute__((ext_vector_type(2)));

#ifndef MOF_FUSED_ABLATE  // diagnostic builds: 1 = no product (the lines get zeros), 2 = no column transforms / cross-power
#define MOF_FUSED_ABLATE 0
#endif

#ifndef MOF_FUSED_UNROLL  // double steps per trip of the product loop (behind a back edge the compiler waits with vmcnt(0))
#define MOF_FUSED_UNROLL 1
#endif
#define MOF_PRAGMA_(x) _Pragma(#x)
#define MOF_PRAGMA(x) MOF_PRAGMA_(x)
#define MOF_FUSED_UNROLL_PRAGMA MOF_PRAGMA(unroll MOF_FUSED_UNROLL)

template <int N>
struct Fused {
  static constexpr int H = N / 2, BINS = 16, T = 256;
  static constexpr int RW = N / 4;                 // image rows per wave
  static constexpr int NT = (RW + 31) / 32;        // row tiles of 32 per wave (the last one may overlap the one before)
  static constexpr int KS2 = N / 32;               // double steps: 32 pixels = one 16-byte load per lane = two MFMA K steps
  static constexpr int TAIL = (N % 32) / 16;       // one more single step (8-byte load) when N = 16 (mod 32)
  static constexpr int STEPS = 2 * KS2 + TAIL;     // MFMA K steps of 16 pixels
  static constexpr int GROUPS = (H + 1 + BINS - 1) / BINS;
  static_assert(N % 16 == 0 && RW >= 32, "row tiles of 32, K steps of 16");
  __host__ __device__ static constexpr int tile_row(int t) { return 32 * t < RW - 32 ? 32 * t : RW - 32; }
  // pixel of K slot (step s, lane half h, j): a lane's 16-byte load covers the slots of two consecutive steps
  __host__ __device__ static constexpr int pixel(int s, int h, int j) {
    return s < 2 * KS2 ? 32 * (s >> 1) + 16 * h + 8 * (s & 1) + j : 32 * KS2 + 8 * h + j;
  }
};

// two dwords of pixels -> eight halves: 0x6400 | b is 1024 + b exactly, minus 1024
__device__ __forceinline__ half8_t px_to_half8(uint32_t d0, uint32_t d1) {
  const half2_t k1024 = {(_Float16)1024.f, (_Float16)1024.f};
  uint32_t r[4];
  r[0] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, d0, 0x04010400u)) - k1024);
  r[1] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, d0, 0x04030402u)) - k1024);
  r[2] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, d1, 0x04010400u)) - k1024);
  r[3] = __builtin_bit_cast(uint32_t, __builtin_bit_cast(half2_t, __builtin_amdgcn_perm(0x64646464u, d1, 0x04030402u)) - k1024);
  half8_t v;
  __builtin_memcpy(&v, r, 16);
  return v;
}

template <int N>
__global__ void __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
sr_cols_fused_kernel(const uint8_t* __restrict__ lp_prev, const uint8_t* __restrict__ lp_cur, size_t lp_stride,
                     const u4_t* __restrict__ frags, const float* __restrict__ twiddles, float* __restrict__ Dt, int n_pairs, int run) {
  using P = SrPlan<N>;
  using F = Fused<N>;
  constexpr int H = N / 2, CW = 4, NT = F::NT;
  constexpr int MV = (N + 63) / 64;      // bins per lane and line (v = lane + 64 m)
  constexpr int MQ = (N / 2 + 63) / 64;  // 16-byte pieces per lane and line
  extern __shared__ __attribute__((aligned(16))) unsigned char fused_lds[];
  cf* zall = reinterpret_cast<cf*>(fused_lds);  // [16][LINE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  // XCD-aware order: workgroups b, b + 8, .. share an XCD and its L2, and the 16 bin groups of a pair all read the same two images:
  // pair-run r goes to XCD r % 8, its groups to consecutive workgroups of that XCD
  const int xcd = blockIdx.x & 7, k = blockIdx.x >> 3;
  const int g = k % F::GROUPS, p0 = (8 * (k / F::GROUPS) + xcd) * run;
  if (p0 >= n_pairs) return;  // (the grid is padded to whole rounds of eight pair-runs)
  const int np = n_pairs - p0 < run ? n_pairs - p0 : run;
  const int u0 = F::BINS * g + CW * wave;  // this wave's four bins
  const bool wave_on = u0 <= H;            // (the last group holds the bin N/2 and clamped copies of it)
  cf* z = zall + CW * wave * P::LINE;
  const u4_t* frag = frags + (size_t)g * F::STEPS * 2 * 64 + lane;
  const int m = lane & 31, h = lane >> 5;

  // ---- the row spectra of this workgroup's 16 bins of one image -> the 16 LDS lines -> this wave's column transforms
  auto spectra = [&](const uint8_t* __restrict__ img) {
    float16_t acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
      for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    // Pixels and matrix fragments of double step s2 + 1 are requested while the products of s2 are formed, into the registers
    // step s2 has just released (the pixels once they are converted, a fragment once its four products are issued): the
    // previous frame's column spectra (64 VGPRs) are alive across this loop, separate prefetch registers do not fit beside them.
    // (Measured alternative, r04: the rows staged through the wave's idle LDS lines by LDS-DMA loads -- 64-byte pieces of 16 rows
    //  per instruction instead of one cache line per lane, two buffers, XOR-swizzled slots -- was slower, 2.13 against 1.82 ms:
    //  the compiler's wait insertion puts vmcnt(0) behind the loop's back edge, which also waits for the DMA just issued.)
    const uint8_t* rowp[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) rowp[t] = img + (size_t)(F::RW * wave + F::tile_row(t) + m) * N + 16 * h;
    u4_t px[NT], b[4];
#pragma unroll
    for (int t = 0; t < NT; ++t) px[t] = *reinterpret_cast<const u4_t*>(rowp[t]);
#pragma unroll
    for (int q = 0; q < 4; ++q) b[q] = frag[(size_t)q * 64];  // (step 2 s2: hi, lo), (step 2 s2 + 1: hi, lo)
MOF_FUSED_UNROLL_PRAGMA
    for (int s2 = 0; s2 < (MOF_FUSED_ABLATE == 1 ? 1 : F::KS2); ++s2) {
      const int sn = s2 + 1 < F::KS2 ? s2 + 1 : s2;  // (the last step re-reads itself: no branch inside the pipeline)
      half8_t a[NT][2];
#pragma unroll
      for (int t = 0; t < NT; ++t) a[t][0] = px_to_half8(px[t].x, px[t].y), a[t][1] = px_to_half8(px[t].z, px[t].w);
#pragma unroll
      for (int t = 0; t < NT; ++t) px[t] = *reinterpret_cast<const u4_t*>(rowp[t] + 32 * sn);
#pragma unroll
      for (int q = 0; q < 4; ++q) {  // (the tiles' accumulators take turns: no product waits for the one before it)
#pragma unroll
        for (int t = 0; t < NT; ++t)
          acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[t][q >> 1], __builtin_bit_cast(half8_t, b[q]), acc[t], 0, 0, 0);
        b[q] = frag[(size_t)(4 * sn + q) * 64];
      }
    }
    if constexpr (F::TAIL) {  // N = 16 (mod 32): one more step of 16 pixels, straight from global memory
      const u4_t bh = frag[(size_t)(4 * F::KS2) * 64], bl = frag[(size_t)(4 * F::KS2 + 1) * 64];
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const u2_t d = *reinterpret_cast<const u2_t*>(img + (size_t)(F::RW * wave + F::tile_row(t) + m) * N + 32 * F::KS2 + 8 * h);
        const half8_t a = px_to_half8(d.x, d.y);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8_t, bh), acc[t], 0, 0, 0);
        acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, __builtin_bit_cast(half8_t, bl), acc[t], 0, 0, 0);
      }
    }
    __syncthreads();  // every wave is done with the lines of the frame before
    // D: column = lane % 32 = (re | im, bin), row (i % 4) + 8 (i / 4) + 4 (lane / 32) of the tile
    {
      float* line = reinterpret_cast<float*>(zall + (m & 15) * P::LINE) + (m >> 4);
#pragma unroll
      for (int t = 0; t < NT; ++t) {
        const int r0 = F::RW * wave + F::tile_row(t) + 4 * h;
#pragma unroll
        for (int i = 0; i < 16; ++i) line[2 * (r0 + (i & 3) + 8 * (i >> 2))] = acc[t][i];
      }
    }
    __syncthreads();
    if (wave_on && MOF_FUSED_ABLATE != 2) {
      SrTw<N> tw;
      tw.load(twiddles, lane);
      wave_fft<N>(z, CW, lane, tw, StoreNatural<N>{});
    }
  };

  cf ap[CW][MV];  // column spectra of the previous frame (doubled)
  spectra(lp_prev + (size_t)p0 * lp_stride);
  if (wave_on) {
#pragma unroll
    for (int s = 0; s < CW; ++s)
#pragma unroll
      for (int q = 0; q < MV; ++q) {
        const int v = lane + 64 * q;
        ap[s][q] = v < N ? lds_read(&z[s * P::LINE + v]) : cf{0.f, 0.f};
      }
    wave_sync();
  }
  for (int j = 0; j < np; ++j) {
    spectra(lp_cur + (size_t)(p0 + j) * lp_stride);
    if (!wave_on) continue;  // (wave-uniform; the barriers all sit inside spectra())
#if MOF_FUSED_ABLATE != 2
    // normalised cross-power spectrum of bins (v, u), conjugated in place; the current spectra move into the registers
#pragma unroll
    for (int s = 0; s < CW; ++s) {
      const int u = u0 + s > H ? H : u0 + s;
      const bool u_edge = u == 0 || u == H;
#pragma unroll
      for (int q = 0; q < MV; ++q) {
        const int v = lane + 64 * q;
        const int vv = v < N ? v : N - 1;
        const cf a = lds_read(&z[s * P::LINE + vv]);
        const cf C = cross_power_ab(a, ap[s][q], u_edge && (vv == 0 || vv == H));
        ap[s][q] = a;
        if (v < N) z[s * P::LINE + v] = {C.x, -C.y};
      }
    }
    wave_sync();
    {
      SrTw<N> tw;
      tw.load(twiddles, lane);
      wave_fft<N>(z, CW, lane, tw, StoreNatural<N>{});
    }
#endif
    cf* D = reinterpret_cast<cf*>(Dt) + (size_t)(p0 + j) * (H + 1) * N;
#pragma unroll
    for (int s = 0; s < CW; ++s) {
      const int u = u0 + s;
#pragma unroll
      for (int q = 0; q < MQ; ++q) {
        const int c = lane + 64 * q;
        if (c < N / 2 && u <= H) {
          const cf a0 = z[s * P::LINE + 2 * c], a1 = z[s * P::LINE + 2 * c + 1];
          stream_store(reinterpret_cast<float4*>(D + (size_t)u * N + 2 * c), make_float4(a0.x, a0.y, a1.x, a1.y));
        }
      }
    }
    wave_sync();
  }
}

template <int N>
hipError_t launch_cols_fused_n(const uint8_t* lp_prev, const uint8_t* lp_cur, size_t lp_stride, const uint32_t* frags, const float* tw,
                               float* Dt, int n_pairs, int run, hipStream_t stream) {
  using F = Fused<N>;
  constexpr size_t lds = sizeof(cf) * 16 * SrPlan<N>::LINE;
  const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&sr_cols_fused_kernel<N>),
                                              hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
  if (attr != hipSuccess) return attr;
  const unsigned runs = (unsigned)((n_pairs + run - 1) / run);
  hipLaunchKernelGGL(sr_cols_fused_kernel<N>, dim3(F::GROUPS * ((runs + 7) / 8) * 8), dim3(F::T), lds, stream, lp_prev, lp_cur, lp_stride,
                     reinterpret_cast<const u4_t*>(frags), tw, Dt, n_pairs, run);
  return hipGetLastError();
}

template <int N>
void build_fragments(std::vector<uint32_t>& out) {
  using F = Fused<N>;
  out.assign((size_t)F::GROUPS * F::STEPS * 2 * 64 * 4, 0u);
  for (int g = 0; g < F::GROUPS; ++g)
    for (int s = 0; s < F::STEPS; ++s)
      for (int l = 0; l < 64; ++l)
        for (int j = 0; j < 8; ++j) {
          const int col = l & 31, h = l >> 5, part = col >> 4;
          int bin = F::BINS * g + (col & 15);
          bin = bin > F::H ? F::H : bin;
          const int n = F::pixel(s, h, j);
          double c, sn;
          unit_root((int)(((long long)bin * n) % N), N, &c, &sn);
          const float w = (float)(part == 0 ? 2.0 * c : -2.0 * sn);  // Zh = 2 * rowDFT (the 1/2 of the untangle lives in cross_power_ab's eps)
          uint16_t hl[2];
          f16_split(w, &hl[0], &hl[1]);
          for (int split = 0; split < 2; ++split) {
            uint32_t& d = out[(((size_t)g * F::STEPS + s) * 2 + split) * 256 + (size_t)l * 4 + (j >> 1)];
            d = (j & 1) ? (d & 0xffffu) | ((uint32_t)hl[split] << 16) : (d & 0xffff0000u) | hl[split];
          }
        }
}

}  // namespace

bool sr_fused_supported(int res) { return res == 480 || res == 240 || res == 256; }

std::vector<uint32_t> sr_fused_fragments(int res) {
  std::vector<uint32_t> v;
  switch (res) {
    case 240: build_fragments<240>(v); break;
    case 256: build_fragments<256>(v); break;
    case 480: build_fragments<480>(v); break;
    default: break;
  }
  return v;
}

hipError_t launch_sr_cols_fused(const uint8_t* lp_prev, const uint8_t* lp_cur, size_t lp_stride, const uint32_t* frags,
                                const float* twiddles, float* Dt, int res, int n_pairs, int run, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  if (run < 1) run = 1;
  if (run > 1 && lp_cur != lp_prev + lp_stride) return hipErrorInvalidValue;  // a run walks cur(p) as prev(p + 1)
  switch (res) {
    case 240: return launch_cols_fused_n<240>(lp_prev, lp_cur, lp_stride, frags, twiddles, Dt, n_pairs, run, stream);
    case 256: return launch_cols_fused_n<256>(lp_prev, lp_cur, lp_stride, frags, twiddles, Dt, n_pairs, run, stream);
    case 480: return launch_cols_fused_n<480>(lp_prev, lp_cur, lp_stride, frags, twiddles, Dt, n_pairs, run, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mof
