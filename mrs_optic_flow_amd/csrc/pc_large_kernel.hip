// pc_large_kernel.hip -- phase correlation of images that do not fit a CU (padded side M > 135), from a run-time plan.
//
// Serves, with one set of kernels,
//   * FftMethod with a large samplePointSize -- the reference accepts any divisor of frameSize and falls back to ONE patch =
//     the whole frame otherwise (/root/reference/src/FftMethod.cpp:1706-1720), each patch through cv::phaseCorrelate (:1836);
//   * scaleRotationEstimator at any resolution (/root/reference/src/scaleRotationEstimator.cpp:3-32, :117) other than the three
//     with hand-tuned transforms (240 / 256 / 480, sr_kernel.hip / sr_seq_kernel.hip).
// cv::phaseCorrelate zero-pads to M = getOptimalDFTSize(n) (published OpenCV algorithm), possibly an odd M; everything below
// works on the padded M x M image. The pipeline is the FRAME form of sr_seq_kernel.hip (each image transformed on its own, so a
// black or constant image has the exactly-zero spectrum the reference's separate transforms give it), with every 1-D transform
// a planned Stockham chain run by one wave on lines in LDS (pc_plan.hpp); NU = M/2 + 1 half-spectrum lines:
//   L5 pcl_rows     : one image -> row half-spectra, two real rows per complex line, untangled, DOUBLED, stored transposed:
//                     Zh[u][v] = 2 rowDFT(v)[u], u < NU (16 image rows per workgroup = 128 contiguous bytes per u)
//   L6 pcl_cols     : one wave per (pair, column u): column transforms of cur and prev, normalised cross-power spectrum with the
//                     real-only-slot rule (pc_common.hpp; the slots are (0 | M/2, 0 | M/2) for even M, DC alone for odd M),
//                     conjugate, column transform back -> Dt[u][y]
//   L7 pcl_rows_inv : Hermitian rows, two per complex transform (16 rows per workgroup), fft-shifted first maximum -> candidates
//   L8 pcl_final    : first-maximum reduction, the 5 x 5 window re-evaluated from Dt in fp64 (as K8, sr_kernel.hip), centroid;
//                     then EITHER pt -> (scale, rot) with the estimator's gate (scaleRotationEstimator.cpp:119-124) OR
//                     shift = -pt with FftMethod's gate (FftMethod.cpp:1838-1856, against samplePointSize / 2 -- unpadded).
// Scratch per image: Zh = NU M complex; per pair: Dt = NU M complex (HBM; the caller owns and sizes it).

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mof_kernels.h"
#include "pc_common.hpp"
#include "pc_plan.hpp"
#include "pc_plan_build.hpp"

namespace mof {

namespace {

constexpr int PCL_T = 256;      // four waves
constexpr int PCL_LINES = 8;    // lines per workgroup in L5 / L7 (16 image rows), two per wave

__device__ __forceinline__ int sk(int x) { return x + (x >> 3); }  // line skew: stride-8 stage writes spread over the banks

template <int DS, int CH>
__device__ __forceinline__ uint32_t fetch_px_l(const uint8_t* __restrict__ base, size_t pitch, int y, int x) {
  if constexpr (DS == 4) {  // long-range mode: the quarter-resolution pixel of cv::resize(.., 1/4, 1/4) (FftMethod.cpp:1931-1932)
    const uint8_t* r1 = base + (size_t)(4 * y + 1) * pitch + 4 * x;
    const uint8_t* r2 = r1 + pitch;
    return ((uint32_t)r1[1] + r1[2] + r2[1] + r2[2] + 2u) >> 2;
  } else if constexpr (CH == 3) {  // CV_RGB2GRAY on BGR8 data, as the node applies it (optic_flow.cpp:1622)
    const uint8_t* p = base + (size_t)y * pitch + 3 * x;
    return rgb2gray_fixed(p[0], p[1], p[2]);
  } else {
    return base[(size_t)y * pitch + x];
  }
}

// four consecutive pixels x0 .. x0 + 3 of row y, one byte each (x0 + 3 inside the patch): as pc_kernel_generic.hip's fetch_px4 (r06)
template <int DS, int CH>
__device__ __forceinline__ uint32_t fetch_px4_l(const uint8_t* __restrict__ base, size_t pitch, int y, int x0) {
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

// ---- L5 ------------------------------------------------------------------------------------------------------------------
template <int DS, int CH, bool EXACT>
__global__ void __launch_bounds__(PCL_T) pcl_rows_kernel(PclSrc src, PcPlan pl, const float* __restrict__ twiddles,
                                                         float* __restrict__ zh, size_t zh_stride, int* __restrict__ flags, int line) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pcl_lds[];
  cf* z = reinterpret_cast<cf*>(pcl_lds);  // [PCL_LINES][line]
  cf* tw = z + PCL_LINES * line;           // [m]
  const int m = pl.m, n = pl.n, NU = (m >> 1) + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, img = blockIdx.y, row0 = blockIdx.x * 2 * PCL_LINES;
  const uint8_t* base;
  if (src.paired) {  // image = 2 (pair * patches + patch) + which (0 cur, 1 prev)
    const int which = img & 1, q = img >> 1, patches = src.grid_x * src.grid_y;
    const int pair = q / patches, pt = q - pair * patches, by = pt / src.grid_x, bx = pt - by * src.grid_x;
    base = src.base[which] + (size_t)pair * src.stride[which] + (size_t)(DS * (src.origin_y + by * src.stride_y)) * src.pitch +
           (size_t)(CH * DS * (src.origin_x + bx * src.stride_x));
  } else {
    base = src.base[0] + (size_t)img * src.stride[0];
  }
  for (int k = tid; k < m; k += PCL_T) tw[k] = {twiddles[2 * k], twiddles[2 * k + 1]};
  // the wave's two lines: line l = image rows (row0 + 2l, row0 + 2l + 1) as real and imaginary part; zeros beyond n x n
  // (copyMakeBorder of cv::phaseCorrelate), u8 -> f32 (convertTo, FftMethod.cpp:1805-1806 / scaleRotationEstimator.cpp:115)
  const uint32_t p00 = fetch_px_l<DS, CH>(base, src.pitch, 0, 0);
  uint32_t diff = 0u;
  {
    // all of the lane's pixel loads go out before the first is used (a load-use loop pays the memory latency once per trip); r06: FOUR pixels
    // per load (chunk c = lane + 64 t of a row: a dword, three dwords of BGR8, or two 16-byte runs of the tapped quarter-resolution rows)
    constexpr int NC = 4;  // ceil(960 / 4 / 64)
    uint32_t px[2][2][NC];
    const uint32_t pat = p00 * 0x01010101u;
    const bool tiny = n < 4;  // (no whole chunk in a row: the byte loop; wave-uniform)
    auto load4 = [&](int y, int x0) -> uint32_t {
      if (tiny) {
        if (y >= n || x0 >= n) return 0u;
        uint32_t v = 0u;
        for (int b = 0; x0 + b < n; ++b) v |= fetch_px_l<DS, CH>(base, src.pitch, y, x0 + b) << (8 * b);
        return v;
      }
      // branch-free (r06, as the tuned row kernel's px4): an unconditional load of row min(y, n - 1), of the four pixels that end no later
      // than the row does; what lies outside the patch is shifted / masked away -- so that all of a lane's loads are in flight together
      const int yc = y < n ? y : n - 1, xc = x0 < n - 4 ? x0 : n - 4, sh = x0 - xc;
      uint32_t v = fetch_px4_l<DS, CH>(base, src.pitch, yc, xc);
      v = sh >= 4 ? 0u : v >> (8 * (sh & 3));
      return y < n ? v : 0u;
    };
    auto inside = [&](int y, int x0) -> uint32_t {
      if (y >= n || x0 >= n) return 0u;
      return x0 + 3 < n ? 0xffffffffu : (1u << (8 * (n - x0))) - 1u;
    };
#pragma unroll
    for (int ll = 0; ll < 2; ++ll) {
      const int y0 = row0 + 2 * (2 * wave + ll), y1 = y0 + 1;
#pragma unroll
      for (int t = 0; t < NC; ++t) {
        const int x0 = 4 * (lane + 64 * t);
        px[ll][0][t] = px[ll][1][t] = 0u;
        if (x0 < m) {
          px[ll][0][t] = load4(y0, x0);
          px[ll][1][t] = load4(y1, x0);
        }
      }
    }
#pragma unroll
    for (int ll = 0; ll < 2; ++ll) {
      const int l = 2 * wave + ll, y0 = row0 + 2 * l, y1 = y0 + 1;
#pragma unroll
      for (int t = 0; t < NC; ++t) {
        const int x0 = 4 * (lane + 64 * t);
        if (x0 < m) {
          const uint32_t a = px[ll][0][t], b = px[ll][1][t];
          diff |= ((a ^ pat) & inside(y0, x0)) | ((b ^ pat) & inside(y1, x0));
#pragma unroll
          for (int k = 0; k < 4; ++k)
            if (x0 + k < m) z[l * line + sk(x0 + k)] = {(float)((a >> (8 * k)) & 0xffu), (float)((b >> (8 * k)) & 0xffu)};
        }
      }
    }
  }
  if (flags) {  // bit 0: some pixel differs from pixel (0, 0); bit 1: pixel (0, 0) is not zero (zeroed by the caller before the launch)
    if (__builtin_amdgcn_ballot_w64(diff != 0u) != 0ull && lane == 0) atomicOr(&flags[img], 1);
    if (blockIdx.x == 0 && tid == 0 && p00 != 0u) atomicOr(&flags[img], 2);
  }
  __syncthreads();
  const Walk rows = {line, 1, 0, ~0, 0};
  if (row0 + 4 * wave < m) pass_lines<EXACT>(z, tw, pl, rows, 2 * wave, 2, lane, false);
  __syncthreads();
  // untangle the two rows of every line (doubled: the 1/2 is folded into cross_power_ab's eps) and store transposed
  // (row pitch of Zh[u][row]: a multiple of 8 complex, so that a workgroup's 16 rows = 128 bytes per bin start on a 64-byte sector; with
  //  pitch m the sizes with m % 8 != 0 paid 30 - 50 % on these stores -- r06, as sr_zh_pitch of the tuned transforms)
  const int zp = (m + 7) & ~7;
  cf* out = reinterpret_cast<cf*>(zh + (size_t)img * zh_stride) + row0;
  for (int i = tid; i < PCL_LINES * NU; i += PCL_T) {
    const int u = i >> 3, j = i & 7, r = row0 + 2 * j;
    if (r >= m) continue;
    const cf zk = z[j * line + sk(u)], zm = z[j * line + sk(u == 0 ? 0 : m - u)];
    cf a2, b2;
    untangle2(zk, zm, &a2, &b2);
    out[(size_t)u * zp + 2 * j] = a2;
    if (r + 1 < m) out[(size_t)u * zp + 2 * j + 1] = b2;
  }
}

// ---- L6 ------------------------------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(PCL_T) pcl_cols_kernel(const float* __restrict__ zh_prev, const float* __restrict__ zh_cur,
                                                         size_t zh_stride, PcPlan pl, const float* __restrict__ twiddles,
                                                         float* __restrict__ Dt, float* __restrict__ cdc, const int* __restrict__ flags,
                                                         int line) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pcl_lds[];
  cf* zall = reinterpret_cast<cf*>(pcl_lds);  // [4 waves][2][line]
  cf* tw = zall + 8 * line;
  const int m = pl.m, hu = m >> 1, NU = hu + 1;
  const bool even = (m & 1) == 0;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pair = blockIdx.y, u = blockIdx.x * 4 + wave;
  cf* z = zall + 2 * wave * line;
  for (int k = tid; k < m; k += PCL_T) tw[k] = {twiddles[2 * k], twiddles[2 * k + 1]};
  const bool active = u < NU;
  if (active) {
    const int zp = (m + 7) & ~7;  // (Zh's row pitch, pcl_rows_kernel)
    const cf* c = reinterpret_cast<const cf*>(zh_cur + (size_t)pair * zh_stride) + (size_t)u * zp;
    const cf* p = reinterpret_cast<const cf*>(zh_prev + (size_t)pair * zh_stride) + (size_t)u * zp;
    constexpr int NX = 15;
    cf cv[NX], pv[NX];  // both lines in flight before the first LDS write
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      const int v = lane + 64 * t;
      if (v < m) {
        cv[t] = c[v];
        pv[t] = p[v];
      }
    }
#pragma unroll
    for (int t = 0; t < NX; ++t) {
      const int v = lane + 64 * t;
      if (v < m) {
        z[sk(v)] = cv[t];
        z[line + sk(v)] = pv[t];
      }
    }
  }
  __syncthreads();
  if (!active) return;
  const Walk w = {line, 1, 0, ~0, 0};
  pass_lines<EXACT>(z, tw, pl, w, 0, 2, lane, false);
  // normalised cross-power spectrum of bins (v, u), conjugated in place (rules: pc_common.hpp)
  const bool u_edge = u == 0 || (even && u == hu);
  // A CONSTANT patch that zero padding turned into an n x n box (n, m even): its spectrum is EXACTLY zero on the Nyquist row and
  // column in the reference (alternating sums of equal numbers), so C = 0 there. Here the rows were transformed in pairs (L5): the
  // spectra of rows 2j and 2j + 1 of a constant image differ by rounding, and their alternating column sum is 79 x that
  // difference instead of 0 -- which the normalisation blows up to unit magnitude (0.04 px on a 158 x 158 constant-against-texture
  // pair: tools/fft_sr_fuzz.py seed 101). The constant patch is known exactly (L5's flags), so are its zero bins -- the rule of
  // the in-LDS planned kernel (pc_kernel_generic.hip, box_zeros).
  // r05: the box's exact zeros are ALL the lines k != 0 with k n = 0 (mod m) -- the multiples of zq (pc_common.hpp, box_zero_period) --, the
  // Nyquist line alone for most sizes but e.g. every multiple of 50 for 196 in 200
  const bool box_zeros = flags && m > pl.n && (((flags[2 * pair] & 1) == 0) || ((flags[2 * pair + 1] & 1) == 0));
  const int zq = box_zero_period(pl.n, m);
  for (int v0 = 0; v0 < m; v0 += 64) {
    const int v = v0 + lane, vv = v < m ? v : m - 1;  // (lanes past the line repeat its last bin: cross_power_ab's wave-uniform
    const cf a = lds_read(&z[sk(vv)]), b = lds_read(&z[line + sk(vv)]);  //  branch wants every lane to take part)
    cf C = cross_power_ab(a, b, u_edge && (vv == 0 || (even && vv == hu)));
    if (box_zeros && (box_zero_line(u, zq) || box_zero_line(vv, zq))) C = {0.f, 0.f};
    if (v < m) z[sk(v)] = {C.x, -C.y};
    if (cdc && u == 0 && v == 0) cdc[pair] = C.x;  // C_dc: all that is left of a degenerate pair's spectrum (pc_common.hpp)
  }
  wave_sync();
  pass_lines<EXACT>(z, tw, pl, w, 0, 1, lane, false);
  cf* D = reinterpret_cast<cf*>(Dt) + ((size_t)pair * NU + u) * m;
  for (int v = lane; v < m; v += 64) D[v] = z[sk(v)];
}

// C_dc of a pair from the row spectra alone (the tuned K6s does not hand it out): the 2-D DC bins are the sums of line u = 0
// (exact integers 2 x row sum each; summed in f64), and the DC slot is real-only (pc_common.hpp)
__global__ void __launch_bounds__(64) pcl_cdc_kernel(const float* __restrict__ zh_prev, const float* __restrict__ zh_cur, size_t zh_stride, int m,
                                                     float* __restrict__ cdc) {
  const int pair = blockIdx.x, lane = threadIdx.x;
  const cf* c = reinterpret_cast<const cf*>(zh_cur + (size_t)pair * zh_stride);
  const cf* p = reinterpret_cast<const cf*>(zh_prev + (size_t)pair * zh_stride);
  double sc = 0.0, sp = 0.0;
  for (int v = lane; v < m; v += 64) {
    sc += (double)c[v].x;
    sp += (double)p[v].x;
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    sc += __shfl_xor(sc, off, 64);
    sp += __shfl_xor(sp, off, 64);
  }
  if (lane == 0) cdc[pair] = cross_power_ab(cf{(float)sc, 0.f}, cf{(float)sp, 0.f}, true).x;
}

// ---- L7 ------------------------------------------------------------------------------------------------------------------
template <bool EXACT>
__global__ void __launch_bounds__(PCL_T) pcl_rows_inv_kernel(const float* __restrict__ Dt, PcPlan pl, const float* __restrict__ twiddles,
                                                             float2* __restrict__ cand, int n_cand, int line) {
  extern __shared__ __attribute__((aligned(16))) unsigned char pcl_lds[];
  cf* z = reinterpret_cast<cf*>(pcl_lds);  // [PCL_LINES][line]
  cf* tw = z + PCL_LINES * line;
  Best* red = reinterpret_cast<Best*>(tw + pl.m);
  const int m = pl.m, H = m >> 1, NU = H + 1;
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pair = blockIdx.y, p0 = blockIdx.x * PCL_LINES;
  for (int k = tid; k < m; k += PCL_T) tw[k] = {twiddles[2 * k], twiddles[2 * k + 1]};
  const cf* D = reinterpret_cast<const cf*>(Dt) + (size_t)pair * NU * m;
  // line l carries rows y1 = 2 (p0 + l), y2 = y1 + 1: E[u] = G[y1][u] + i G[y2][u], G[y][M - u] = conj G[y][u]
  {
    constexpr int NI = 16;  // ceil(8 * 481 / 256)
    cf g1[NI], g2[NI];
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int i = tid + PCL_T * t, u = i >> 3, l = i & 7, y1 = 2 * (p0 + l);
      g1[t] = g2[t] = cf{0.f, 0.f};
      if (i < PCL_LINES * NU && y1 < m) {
        g1[t] = D[(size_t)u * m + y1];
        if (y1 + 1 < m) g2[t] = D[(size_t)u * m + y1 + 1];
      }
    }
#pragma unroll
    for (int t = 0; t < NI; ++t) {
      const int i = tid + PCL_T * t, u = i >> 3, l = i & 7, y1 = 2 * (p0 + l);
      if (i < PCL_LINES * NU && y1 < m) {
        z[l * line + sk(u)] = {g1[t].x - g2[t].y, g1[t].y + g2[t].x};
        const int um = m - u;
        if (u > 0 && um != u) z[l * line + sk(um)] = {g1[t].x + g2[t].y, g2[t].x - g1[t].y};
      }
    }
  }
  __syncthreads();
  int nl = 0;
#pragma unroll
  for (int ll = 0; ll < 2; ++ll)
    if (2 * (p0 + 2 * wave + ll) < m) ++nl;
  const Walk w = {line, 1, 0, ~0, 0};
  if (nl > 0) pass_lines<EXACT>(z, tw, pl, w, 2 * wave, nl, lane, false);
  // first maximum of the fft-shifted surface (fftShift: index i -> (i + (m >> 1)) mod m for even and odd m; minMaxLoc)
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  for (int ll = 0; ll < nl; ++ll) {
    const int l = 2 * wave + ll, y1 = 2 * (p0 + l), y2 = y1 + 1;
    const int r1 = (y1 + H >= m ? y1 + H - m : y1 + H) * m, r2 = (y2 + H >= m ? y2 + H - m : y2 + H) * m;
    for (int x = lane; x < m; x += 64) {
      const int xs = x + H >= m ? x + H - m : x + H;
      const cf v = z[l * line + sk(x)];
      best = better(best, Best{v.x, r1 + xs});
      if (y2 < m) best = better(best, Best{v.y, r2 + xs});
    }
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (tid == 0) {
    for (int k = 1; k < PCL_T / 64; ++k) best = better(best, red[k]);
    cand[(size_t)pair * n_cand + blockIdx.x] = make_float2(best.v, __int_as_float(best.idx));
  }
}

// ---- L8 ------------------------------------------------------------------------------------------------------------------
// The 25 window values are re-evaluated from the half spectrum of their rows (Dt), in double, as K8 (sr_kernel.hip) does:
//   S[y][x] = Re G[y][0] + [M even: (-1)^x Re G[y][M/2]] + 2 sum_{u=1}^{(M-1)/2} Re(G[y][u] W^{ux})
__global__ void __launch_bounds__(64) pcl_final_kernel(PclFinal a) {
  __shared__ double part[25][65];
  const int m = a.m, H = m >> 1;
  const bool even = (m & 1) == 0;
  const int umax = even ? H - 1 : H;
  const int lane = threadIdx.x, pair = blockIdx.x;
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  for (int i = lane; i < a.n_cand; i += 64) {
    const float2 c = a.cand[(size_t)pair * a.n_cand + i];
    best = better(best, Best{c.x, __float_as_int(c.y)});
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  const cf* Dt = reinterpret_cast<const cf*>(a.Dt) + (size_t)pair * (H + 1) * m;
  const bool have = best.idx != 0x7fffffff;
  const int px = have ? best.idx % m : 0, py = have ? best.idx / m : 0;
  int wy[5], wx[5];  // window rows / columns in un-shifted coordinates
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    wy[k] = ((((py - 2 + k) % m) + m) % m - H + m) % m;
    wx[k] = ((((px - 2 + k) % m) + m) % m - H + m) % m;
  }
  double acc[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) acc[k] = 0.0;
  for (int u = 1 + lane; u <= umax; u += 64) {
    cf f[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) f[r] = Dt[(size_t)u * m + wy[r]];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float2 w = *reinterpret_cast<const float2*>(a.twiddles + 2 * (int)(((long)u * wx[c]) % m));  // (cos, -sin)
#pragma unroll
      for (int r = 0; r < 5; ++r) acc[r * 5 + c] += (double)f[r].x * (double)w.x - (double)f[r].y * (double)w.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 25; ++k) part[k][lane] = acc[k];
  __syncthreads();
  const int ys = py - 2 + lane / 5, xs = px - 2 + lane % 5;
  double cx = 0.0, cy = 0.0, sum = 0.0;
  if (have && lane < 25 && ys >= 0 && ys <= m - 1 && xs >= 0 && xs <= m - 1) {  // window clamped to the (padded) image
    const int y = wy[lane / 5], x = wx[lane % 5];
    double s2 = 0.0;
    for (int l = 0; l < 64; ++l) s2 += part[lane][l];
    double s0 = (double)Dt[y].x;
    if (even) s0 += ((x & 1) ? -1.0 : 1.0) * (double)Dt[(size_t)H * m + y].x;
    const double val = (double)(float)(s0 + 2.0 * s2);  // the surface is CV_32F
    cx = (double)xs * val;
    cy = (double)ys * val;
    sum = val;
  }
#pragma unroll
  for (int off = 16; off > 0; off >>= 1) {
    cx += __shfl_xor(cx, off, 64);
    cy += __shfl_xor(cy, off, 64);
    sum += __shfl_xor(sum, off, 64);
  }
  if (lane != 0) return;
  sum += 2.220446049250313e-16;  // DBL_EPSILON, FftMethod.cpp:1378
  const double half_m = (double)m / 2.0;  // cv::phaseCorrelate's centre: that of the PADDED image
  if (a.mode == 0) {
    // scaleRotationEstimator: pt = center - t, NOT negated (:117); |pt.x| > resolution / 2 (int division) -> (1, 0) (:119-121)
    const double ptx = half_m - cx / sum, pty = half_m - cy / sum;
    double scale = 1.0, rot = 0.0;
    if (!(fabs(ptx) > (double)(a.n / 2))) {
      scale = exp(ptx / a.M_log);
      rot = (pty / ((double)a.n / 360.0)) * (3.14159265358979323846 / 180.0);
    }
    double* o = a.out + 4 * (size_t)pair;
    o[0] = scale;
    o[1] = rot;
    o[2] = ptx;
    o[3] = pty;
  } else {
    // FftMethod: shift = -cv::phaseCorrelate(cur, prev) = t - M/2 (:1836), gate against samplePointSize / 2 (:1838-1856)
    double sx = cx / sum - half_m, sy = cy / sum - half_m;
    bool degenerate = false;
    if (a.flags) {  // a constant patch (pc_common.hpp, degenerate pairs); padded, only the all-zero patch stays constant
      const int fc = a.flags[2 * pair], fp = a.flags[2 * pair + 1];
      const bool cc = (fc & 1) == 0, pc = (fp & 1) == 0;
      degenerate = m == a.n ? (cc || pc) : ((cc && (fc & 2) == 0) || (pc && (fp & 2) == 0));
    }
    if (degenerate) {
      const double c9 = 9.0 * (double)a.cdc[pair];
      sx = sy = (c9 > 0.0 ? c9 / (c9 + 2.220446049250313e-16) : 0.0) - half_m;
    }
    const double half_n = (double)a.n / 2.0;
    const bool bad = (sx * sx + sy * sy > a.max_px_speed_sq) || (fabs(sx) > half_n) || (fabs(sy) > half_n) || (sx != sx) ||
                     (sy != sy) || (!have && !degenerate);
    if (bad) sx = sy = __builtin_nan("");
    a.out[2 * (size_t)pair] = sx;
    a.out[2 * (size_t)pair + 1] = sy;
  }
}

int pcl_line(int m) { return (m + ((m - 1) >> 3) + 1) | 1; }  // skewed line length, odd: the lines of a workgroup start on different banks

size_t pcl_lds_bytes(int m) { return sizeof(float) * 2 * ((size_t)PCL_LINES * pcl_line(m) + m) + 64; }

// The two-body stage routine (8 and 4 register slots per butterfly, pc_plan.hpp) holds 64 * 16 / slots butterflies of a line:
// enough unless a radix-5 stage meets m > 640, a radix-3 one m > 768 or a radix-2 one m > 512 -- those plans take the kernels
// built with one body per radix (more registers: three waves per SIMD instead of four).
bool needs_exact(const PcPlan& pl) {
  for (int s = 0; s < pl.n_stages; ++s) {
    const int R = pl.radix[s], slots = R > 4 ? 8 : 4;
    if (pl.m / R > 64 * (16 / slots)) return true;
  }
  return false;
}

template <class K>
hipError_t allow_lds(K kernel, size_t bytes) {
  return bytes > 48 * 1024 ? hipFuncSetAttribute(reinterpret_cast<const void*>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)bytes)
                           : hipSuccess;
}

}  // namespace

bool pc_build_line_plan(int n, PcPlan* out) {
  PcPlan pl{};
  if (!pc_line_plan_c(n, pl)) return false;
  *out = pl;
  return true;
}

namespace {
__global__ void __launch_bounds__(256) pcl_seq_flags_kernel(const int* __restrict__ fs, int* __restrict__ f2, int patches, int n_pairs) {
  const int q = blockIdx.x * 256 + threadIdx.x;
  if (q < n_pairs) {
    f2[2 * q] = fs[q + patches];
    f2[2 * q + 1] = fs[q];
  }
}
}  // namespace
hipError_t launch_pcl_seq_flags(const int* fs, int* f2, int patches, int n_pairs, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  hipLaunchKernelGGL(pcl_seq_flags_kernel, dim3((unsigned)((n_pairs + 255) / 256)), dim3(256), 0, stream, fs, f2, patches, n_pairs);
  return hipGetLastError();
}

size_t pcl_zh_floats(const PcPlan& pl) { return (size_t)((pl.m >> 1) + 1) * (((pl.m + 7) & ~7) + 8) * 2; }  // (room for the tuned transforms' row pitch, sr_zh_pitch; the planned kernels use pitch m)
int pcl_candidates(const PcPlan& pl) { return (((pl.m + 1) >> 1) + PCL_LINES - 1) / PCL_LINES; }

hipError_t launch_pcl_rows(const PclSrc& src, const PcPlan& pl, const float* twiddles, float* zh, size_t zh_stride, int* flags,
                           int n_images, int channels, int downscale, hipStream_t stream) {
  if (n_images <= 0) return hipSuccess;
  if ((channels != 1 && channels != 3) || (downscale != 1 && downscale != 4) || (channels == 3 && downscale == 4)) return hipErrorInvalidValue;
  const int line = pcl_line(pl.m);
  const size_t lds = pcl_lds_bytes(pl.m);
  const bool ex = needs_exact(pl);
  hipError_t e;
  if ((e = allow_lds(&pcl_rows_kernel<1, 1, false>, lds)) != hipSuccess || (e = allow_lds(&pcl_rows_kernel<1, 3, false>, lds)) != hipSuccess ||
      (e = allow_lds(&pcl_rows_kernel<4, 1, false>, lds)) != hipSuccess || (e = allow_lds(&pcl_rows_kernel<1, 1, true>, lds)) != hipSuccess ||
      (e = allow_lds(&pcl_rows_kernel<1, 3, true>, lds)) != hipSuccess || (e = allow_lds(&pcl_rows_kernel<4, 1, true>, lds)) != hipSuccess)
    return e;
  const unsigned gx = (unsigned)((pl.m + 2 * PCL_LINES - 1) / (2 * PCL_LINES));
  for (int f0 = 0; f0 < n_images; f0 += 65534) {  // the image index rides gridDim.y (an even count keeps cur / prev pairs together)
    const int nf = n_images - f0 < 65534 ? n_images - f0 : 65534;
    PclSrc s = src;
    float* z0 = zh + (size_t)f0 * zh_stride;
    int* fl = flags ? flags + f0 : nullptr;
    if (s.paired) {
      // pcl_rows_kernel derives (pair, patch) from the image index: shift the bases by whole pairs only
      const int per_pair = 2 * s.grid_x * s.grid_y;
      if (f0 % per_pair != 0) return hipErrorInvalidValue;  // (callers launch at most 65534 images at a time on pair boundaries)
      s.base[0] += (size_t)(f0 / per_pair) * s.stride[0];
      s.base[1] += (size_t)(f0 / per_pair) * s.stride[1];
    } else {
      s.base[0] += (size_t)f0 * s.stride[0];
    }
    const dim3 g(gx, (unsigned)nf);
#define PCL_ROWS(DS_, CH_)                                                                                                              \
  do {                                                                                                                                  \
    if (ex) hipLaunchKernelGGL((pcl_rows_kernel<DS_, CH_, true>), g, dim3(PCL_T), lds, stream, s, pl, twiddles, z0, zh_stride, fl, line); \
    else hipLaunchKernelGGL((pcl_rows_kernel<DS_, CH_, false>), g, dim3(PCL_T), lds, stream, s, pl, twiddles, z0, zh_stride, fl, line);   \
  } while (0)
    if (downscale == 4) PCL_ROWS(4, 1);
    else if (channels == 3) PCL_ROWS(1, 3);
    else PCL_ROWS(1, 1);
#undef PCL_ROWS
  }
  return hipGetLastError();
}

hipError_t launch_pcl_cols(const float* zh_prev, const float* zh_cur, size_t zh_stride, const PcPlan& pl, const float* twiddles,
                           float* Dt, float* cdc, const int* flags, int n_pairs, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  const int line = pcl_line(pl.m), NU = (pl.m >> 1) + 1;
  const size_t lds = pcl_lds_bytes(pl.m);
  const bool ex = needs_exact(pl);
  hipError_t e = ex ? allow_lds(&pcl_cols_kernel<true>, lds) : allow_lds(&pcl_cols_kernel<false>, lds);
  if (e != hipSuccess) return e;
  for (int p0 = 0; p0 < n_pairs; p0 += 65535) {
    const int np = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
    if (ex)
      hipLaunchKernelGGL(pcl_cols_kernel<true>, dim3((unsigned)((NU + 3) / 4), (unsigned)np), dim3(PCL_T), lds, stream,
                         zh_prev + (size_t)p0 * zh_stride, zh_cur + (size_t)p0 * zh_stride, zh_stride, pl, twiddles,
                         Dt + (size_t)p0 * NU * pl.m * 2, cdc ? cdc + p0 : nullptr, flags ? flags + 2 * (size_t)p0 : nullptr, line);
    else
      hipLaunchKernelGGL(pcl_cols_kernel<false>, dim3((unsigned)((NU + 3) / 4), (unsigned)np), dim3(PCL_T), lds, stream,
                         zh_prev + (size_t)p0 * zh_stride, zh_cur + (size_t)p0 * zh_stride, zh_stride, pl, twiddles,
                         Dt + (size_t)p0 * NU * pl.m * 2, cdc ? cdc + p0 : nullptr, flags ? flags + 2 * (size_t)p0 : nullptr, line);
  }
  return hipGetLastError();
}

hipError_t launch_pcl_cdc(const float* zh_prev, const float* zh_cur, size_t zh_stride, int m, float* cdc, int n_pairs, hipStream_t stream) {
  for (int p0 = 0; p0 < n_pairs; p0 += 65535) {
    const int np = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
    hipLaunchKernelGGL(pcl_cdc_kernel, dim3((unsigned)np), dim3(64), 0, stream, zh_prev + (size_t)p0 * zh_stride, zh_cur + (size_t)p0 * zh_stride,
                       zh_stride, m, cdc + p0);
  }
  return hipGetLastError();
}

hipError_t launch_pcl_peak(const PclFinal& a_in, const PcPlan& pl, int n_pairs, hipStream_t stream, bool candidates_done) {
  if (n_pairs <= 0) return hipSuccess;
  const int line = pcl_line(pl.m), NU = (pl.m >> 1) + 1, n_cand = pcl_candidates(pl);
  const size_t lds = pcl_lds_bytes(pl.m);
  const bool ex = needs_exact(pl);
  hipError_t e = ex ? allow_lds(&pcl_rows_inv_kernel<true>, lds) : allow_lds(&pcl_rows_inv_kernel<false>, lds);
  if (e != hipSuccess) return e;
  for (int p0 = 0; p0 < n_pairs; p0 += 65535) {
    const int np = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
    PclFinal a = a_in;
    a.m = pl.m;
    a.n = pl.n;
    a.n_cand = n_cand;
    a.Dt = a_in.Dt + (size_t)p0 * NU * pl.m * 2;
    a.cand = a_in.cand + (size_t)p0 * n_cand;
    a.out = a_in.out + (size_t)p0 * (a.mode == 0 ? 4 : 2);
    if (a.flags) a.flags = a_in.flags + 2 * (size_t)p0;
    if (a.cdc) a.cdc = a_in.cdc + p0;
    if (candidates_done) {
      // (L7 was run by somebody else: the tuned K7 for patches of 240 / 256 / 480 pixels)
    } else if (ex)
      hipLaunchKernelGGL(pcl_rows_inv_kernel<true>, dim3((unsigned)n_cand, (unsigned)np), dim3(PCL_T), lds, stream, a.Dt, pl, a.twiddles,
                         const_cast<float2*>(a.cand), n_cand, line);
    else
      hipLaunchKernelGGL(pcl_rows_inv_kernel<false>, dim3((unsigned)n_cand, (unsigned)np), dim3(PCL_T), lds, stream, a.Dt, pl, a.twiddles,
                         const_cast<float2*>(a.cand), n_cand, line);
    hipLaunchKernelGGL(pcl_final_kernel, dim3((unsigned)np), dim3(64), 0, stream, a);
  }
  return hipGetLastError();
}

}  // namespace mof
