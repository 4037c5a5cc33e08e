// sr_kernel.hip -- K4..K8: the scale/rotation estimator of BASELINE config c5 on gfx950.
//
// Replaces scaleRotationEstimator::processImage (/root/reference/src/scaleRotationEstimator.cpp:34-148):
//   cv::logPolar(imCurr, tempIm, center, M, INTER_CUBIC | INTER_LANCZOS4)     :45, :112     -> K4 sr_logpolar_kernel
//   cv::phaseCorrelate(tempIm_F32, prevIm_F32) on the whole res x res image   :117          -> K5..K8
// The res x res complex tile (480^2 x 8 B = 1.8 MB) does not fit in LDS, so unlike K1 the whole-frame
// correlation is a short pipeline through L2/Infinity-Cache-resident scratch (a few MB per frame pair):
//   K5 sr_rows_fwd   : u8 log-polar rows of cur/prev packed as cur + i*prev, row FFTs in LDS  -> Z
//   K6 sr_cols       : column FFTs of a column group AND its mirror, untangle + normalised cross-power
//                      spectrum (same rules as K1, incl. the real-only slots), inverse column FFTs of the
//                      half spectrum                                                          -> D (N x (N/2+1))
//   K7 sr_rows_inv   : Hermitian rows, two per complex transform -> real surface + per-workgroup arg-max
//   K8 sr_final      : first-maximum reduction, 5x5 fp64 centroid, pt -> (scale, rot) with the reference's gate
// 1-D transforms are Stockham stages in LDS over mixed radices (480 = 15 x 8 x 4, 240 = 15 x 16, 256 = 16 x 16).

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"

namespace mof {

namespace {

constexpr int SR_T = 256;  // threads per workgroup in K5..K7

__device__ __forceinline__ void butterfly3(cf* a) {
  const float s3 = 0.86602540378443864676f;
  const cf s = cadd(a[1], a[2]), d = csub(a[1], a[2]);
  const cf m = {a[0].x - 0.5f * s.x, a[0].y - 0.5f * s.y};
  a[0] = cadd(a[0], s);
  a[1] = {m.x + s3 * d.y, m.y - s3 * d.x};
  a[2] = {m.x - s3 * d.y, m.y + s3 * d.x};
}

__device__ __forceinline__ void butterfly5(cf* a) {
  const float c1 = 0.30901699437494742410f, c2 = -0.80901699437494742410f;
  const float s1 = 0.95105651629515357212f, s2 = 0.58778525229247312917f;
  const cf s14 = cadd(a[1], a[4]), d14 = csub(a[1], a[4]), s23 = cadd(a[2], a[3]), d23 = csub(a[2], a[3]);
  const cf p1 = {a[0].x + c1 * s14.x + c2 * s23.x, a[0].y + c1 * s14.y + c2 * s23.y};
  const cf p2 = {a[0].x + c2 * s14.x + c1 * s23.x, a[0].y + c2 * s14.y + c1 * s23.y};
  const cf q1 = {s1 * d14.x + s2 * d23.x, s1 * d14.y + s2 * d23.y};
  const cf q2 = {s2 * d14.x - s1 * d23.x, s2 * d14.y - s1 * d23.y};
  a[0] = {a[0].x + s14.x + s23.x, a[0].y + s14.y + s23.y};
  a[1] = {p1.x + q1.y, p1.y - q1.x};
  a[4] = {p1.x - q1.y, p1.y + q1.x};
  a[2] = {p2.x + q2.y, p2.y - q2.x};
  a[3] = {p2.x - q2.y, p2.y + q2.x};
}

template <int R>
__device__ __forceinline__ void bfly(cf* v) {
  if constexpr (R == 15) {
    const cf w15[9] = {{1.f, 0.f},
                       {0.91354545764260089550f, -0.40673664307580020775f},
                       {0.66913060635885821383f, -0.74314482547739423501f},
                       {0.30901699437494742410f, -0.95105651629515357212f},
                       {-0.10452846326765347140f, -0.99452189536827333692f},
                       {-0.5f, -0.86602540378443864676f},
                       {-0.80901699437494742410f, -0.58778525229247312917f},
                       {-0.97814760073380563793f, -0.20791169081775933710f},
                       {-0.97814760073380563793f, 0.20791169081775933710f}};
    cf t[5][3];
#pragma unroll
    for (int n2 = 0; n2 < 5; ++n2) {
      cf a[3] = {v[n2], v[5 + n2], v[10 + n2]};
      butterfly3(a);
#pragma unroll
      for (int k1 = 0; k1 < 3; ++k1) t[n2][k1] = (n2 * k1 == 0) ? a[k1] : cmul(a[k1], w15[n2 * k1]);
    }
#pragma unroll
    for (int k1 = 0; k1 < 3; ++k1) {
      cf b[5] = {t[0][k1], t[1][k1], t[2][k1], t[3][k1], t[4][k1]};
      butterfly5(b);
#pragma unroll
      for (int k2 = 0; k2 < 5; ++k2) v[k1 + 3 * k2] = b[k2];
    }
  } else {
    butterfly<R>(v);
  }
}

// One in-place Stockham stage (radix R, P = product of the radices already applied) over `nlines` lines of
// length N stored line-major in LDS. Every thread of the workgroup must call it (two barriers inside).
template <int N, int R, int LINES>
__device__ __forceinline__ void lds_stage(cf* __restrict__ z, int nlines, int P, const float* __restrict__ tw, int tid) {
  constexpr int BPL = N / R, PER = (LINES * BPL + SR_T - 1) / SR_T;
  const int total = nlines * BPL;
  cf v[PER][R];
  int dst[PER];
#pragma unroll
  for (int b = 0; b < PER; ++b) {
    const int g = tid + b * SR_T;
    dst[b] = -1;
    if (g < total) {
      const int line = g / BPL, x = g % BPL, j = x % P;
      const cf* src = z + line * N + x;
      const int tstep = N / (P * R);
#pragma unroll
      for (int k = 0; k < R; ++k) {
        cf a = src[k * BPL];
        if (k > 0 && P > 1) {
          const float2 w = *reinterpret_cast<const float2*>(tw + 2 * ((k * j) * tstep));
          a = cmul(a, cf{w.x, w.y});
        }
        v[b][k] = a;
      }
      bfly<R>(v[b]);
      dst[b] = line * N + (x - j) * R + j;
    }
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < PER; ++b)
    if (dst[b] >= 0) {
#pragma unroll
      for (int k = 0; k < R; ++k) z[dst[b] + k * P] = v[b][k];
    }
  __syncthreads();
}

template <int N, int LINES>
__device__ __forceinline__ void lds_fft(cf* z, int nlines, const float* tw, int tid) {
  if constexpr (N == 480) {
    lds_stage<N, 15, LINES>(z, nlines, 1, tw, tid);
    lds_stage<N, 8, LINES>(z, nlines, 15, tw, tid);
    lds_stage<N, 4, LINES>(z, nlines, 120, tw, tid);
  } else if constexpr (N == 240) {
    lds_stage<N, 15, LINES>(z, nlines, 1, tw, tid);
    lds_stage<N, 16, LINES>(z, nlines, 15, tw, tid);
  } else {
    static_assert(N == 256, "scale/rotation resolutions: 240, 256, 480");
    lds_stage<N, 16, LINES>(z, nlines, 1, tw, tid);
    lds_stage<N, 16, LINES>(z, nlines, 16, tw, tid);
  }
}

#ifndef MOF_SR_ROWS
#define MOF_SR_ROWS 4
#endif
constexpr int ROWS_L = MOF_SR_ROWS;   // rows per workgroup in K5
#ifndef MOF_SR_CW
#define MOF_SR_CW 8
#endif
constexpr int COLS_CW = MOF_SR_CW;  // columns (plus their mirrors) per workgroup in K6
constexpr int INV_L = 8;    // row PAIRS per workgroup in K7

}  // namespace

// ---- K4: cv::logPolar as one gather per destination pixel -----------------------------------------------
// map[pixel] = {anchor x, anchor y, table index, valid}; valid = anchor inside the source (BORDER_TRANSPARENT:
// other pixels keep their content). weights: [32*32][K*K] shorts summing to 2^15. Footprints crossing the
// border take BORDER_REFLECT_101 taps. dst = (sum + 2^14) >> 15 saturated: integer, bit-exact.
template <int K>
__global__ void __launch_bounds__(256) sr_logpolar_kernel(SrLpArgs a) {
  // a wave covers an 8 (phi) x 8 (rho) tile of the destination: its 64 footprints then share a compact patch of the
  // source instead of lying along a ray (the kernel is bound by L1 line look-ups: 26 -> ~10 per wave-load)
  const int res = a.res;
  const int tiles_rho = (res + 31) / 32;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  const int rho = (blockIdx.x % tiles_rho) * 32 + wave * 8 + (lane & 7);
  const int phi = (blockIdx.x / tiles_rho) * 8 + (lane >> 3);
  if (rho >= res || phi >= res) return;
  const int pix = phi * res + rho;
  const int img = blockIdx.y;
  const SrMapEntry m = a.map[pix];
  if (!m.valid) return;
  const uint8_t* src = a.src + (size_t)img * a.src_stride;
#ifdef MOF_SR_ABL_W  // diagnostic build: every lane reads weight row 0 (results wrong by design)
  const int16_t* w = a.weights;
#else
  const int16_t* w = a.weights + (size_t)m.widx * (K * K);
#endif
  constexpr int HALF = K / 2 - 1;
  const int sx = m.ax - HALF, sy = m.ay - HALF;
  int sum = 0;
  if (sx >= 0 && sy >= 0 && sx + K <= res && sy + K <= res) {
    // interior footprint: one K-byte pixel load (any alignment) and one 2K-byte weight load (aligned) per tap row
#pragma unroll
    for (int k1 = 0; k1 < K; ++k1) {
      uint32_t px[K / 4], wq[K / 2];
      __builtin_memcpy(px, src + (size_t)(sy + k1) * a.pitch + sx, K);
      __builtin_memcpy(wq, __builtin_assume_aligned(w + k1 * K, 2 * K), 2 * K);
#pragma unroll
      for (int k2 = 0; k2 < K; ++k2) {
        const int pv = (int)((px[k2 >> 2] >> (8 * (k2 & 3))) & 0xffu);
        const int wv = (int)(int16_t)(wq[k2 >> 1] >> (16 * (k2 & 1)));
        sum += pv * wv;
      }
    }
  } else {
    for (int k1 = 0; k1 < K; ++k1) {
      int yy = sy + k1;
      while (yy < 0 || yy >= res) yy = yy < 0 ? -yy : 2 * res - 2 - yy;
      for (int k2 = 0; k2 < K; ++k2) {
        int xx = sx + k2;
        while (xx < 0 || xx >= res) xx = xx < 0 ? -xx : 2 * res - 2 - xx;
        sum += (int)src[(size_t)yy * a.pitch + xx] * (int)w[k1 * K + k2];
      }
    }
  }
  int v = (sum + (1 << 14)) >> 15;
  v = v < 0 ? 0 : (v > 255 ? 255 : v);
  a.dst[(size_t)img * a.dst_stride + pix] = (uint8_t)v;
}


// Same remap with the whole 2-D weight table resident in LDS. In sr_logpolar_kernel every lane fetches its own
// K*K-short weight row from the 128-KB (Lanczos4) table: 128 B per destination pixel out of L2, 8 TB/s chip-wide at the
// measured rate -- the L2, not the arithmetic, set that kernel's time (an all-lanes-one-row build ran 2x faster).
// Here one persistent 16-wave workgroup per CU copies the table once (rows padded to 144 / 40 B so that random rows
// spread over the banks) and then walks 8 x 8 destination tiles; only the source pixels still come through L1.
template <int K>
struct LpLds {
  static constexpr int ROW_B = (K == 8) ? 144 : 40;  // padded row, bytes (K*K*2 = 128 / 32 payload)
  static constexpr size_t BYTES = (size_t)1024 * ROW_B;
};

template <int K>
__global__ void __launch_bounds__(1024) sr_logpolar_lds_kernel(SrLpArgs a, int n_images) {
  extern __shared__ __attribute__((aligned(16))) unsigned char wl[];
  constexpr int ROW_B = LpLds<K>::ROW_B, PARTS = (K * K * 2) / (2 * K);  // one part = one tap row = 2K bytes
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
  for (int c = tid; c < 1024 * PARTS; c += 1024) {
    uint32_t v[K / 2];
    __builtin_memcpy(v, reinterpret_cast<const unsigned char*>(a.weights) + (size_t)c * 2 * K, 2 * K);
    __builtin_memcpy(wl + (c / PARTS) * ROW_B + (c % PARTS) * 2 * K, v, 2 * K);
  }
  __syncthreads();
  const int res = a.res, tiles = (res + 7) / 8, per_img = tiles * tiles, total = n_images * per_img;
  constexpr int HALF = K / 2 - 1;
  for (int t = blockIdx.x * 16 + wave; t < total; t += gridDim.x * 16) {
    const int img = t / per_img, r = t % per_img;
    const int rho = (r % tiles) * 8 + (lane & 7), phi = (r / tiles) * 8 + (lane >> 3);
    if (rho >= res || phi >= res) continue;
    const int pix = phi * res + rho;
    const SrMapEntry m = a.map[pix];
    if (!m.valid) continue;
    const uint8_t* src = a.src + (size_t)img * a.src_stride;
    const unsigned char* w = wl + (int)m.widx * ROW_B;
    const int sx = m.ax - HALF, sy = m.ay - HALF;
    int sum = 0;
    if (sx >= 0 && sy >= 0 && sx + K <= res && sy + K <= res) {
#pragma unroll
      for (int k1 = 0; k1 < K; ++k1) {
        uint32_t px[K / 4], wq[K / 2];
        __builtin_memcpy(px, src + (size_t)(sy + k1) * a.pitch + sx, K);
        __builtin_memcpy(wq, __builtin_assume_aligned(w + k1 * 2 * K, 8), 2 * K);
#pragma unroll
        for (int k2 = 0; k2 < K; ++k2) {
          const int pv = (int)((px[k2 >> 2] >> (8 * (k2 & 3))) & 0xffu);
          const int wv = (int)(int16_t)(wq[k2 >> 1] >> (16 * (k2 & 1)));
          sum += pv * wv;
        }
      }
    } else {
      const int16_t* ws = reinterpret_cast<const int16_t*>(w);
      for (int k1 = 0; k1 < K; ++k1) {
        int yy = sy + k1;
        while (yy < 0 || yy >= res) yy = yy < 0 ? -yy : 2 * res - 2 - yy;
        for (int k2 = 0; k2 < K; ++k2) {
          int xx = sx + k2;
          while (xx < 0 || xx >= res) xx = xx < 0 ? -xx : 2 * res - 2 - xx;
          sum += (int)src[(size_t)yy * a.pitch + xx] * (int)ws[k1 * K + k2];
        }
      }
    }
    int v = (sum + (1 << 14)) >> 15;
    v = v < 0 ? 0 : (v > 255 ? 255 : v);
    a.dst[(size_t)img * a.dst_stride + pix] = (uint8_t)v;
  }
}

// ---- K5: forward row transforms of z = cur_lp + i prev_lp ------------------------------------------------
template <int N>
__global__ void __launch_bounds__(SR_T) sr_rows_fwd_kernel(SrPcArgs a) {
  __shared__ cf z[ROWS_L * N];
  const int tid = threadIdx.x, pair = blockIdx.y, row0 = blockIdx.x * ROWS_L;
  const uint8_t* cur = a.lp_cur + (size_t)pair * a.lp_stride + (size_t)row0 * N;
  const uint8_t* prev = a.lp_prev + (size_t)pair * a.lp_stride + (size_t)row0 * N;
  for (int i = tid; i < ROWS_L * N; i += SR_T) z[i] = {(float)cur[i], (float)prev[i]};  // convertTo CV_32FC1, :115
  __syncthreads();
  lds_fft<N, ROWS_L>(z, ROWS_L, a.twiddles, tid);
  cf* Z = reinterpret_cast<cf*>(a.Z) + (size_t)pair * N * N + (size_t)row0 * N;
  for (int i = tid; i < ROWS_L * N; i += SR_T) Z[i] = z[i];
}

// ---- K6: column transforms, cross-power spectrum, inverse column transforms of the half spectrum -----------
template <int N>
__global__ void __launch_bounds__(SR_T) sr_cols_kernel(SrPcArgs a) {
  constexpr int H = N / 2, CW = COLS_CW;
  __shared__ cf z[2 * CW * N];  // slots 0..CW-1: columns u0+s ; slots CW..2CW-1: their mirrors (N - u) % N
  const int tid = threadIdx.x, pair = blockIdx.y, u0 = blockIdx.x * CW;
  const cf* Z = reinterpret_cast<const cf*>(a.Z) + (size_t)pair * N * N;
  for (int i = tid; i < 2 * CW * N; i += SR_T) {
    const int v = i / (2 * CW), s = i % (2 * CW);
    int u = u0 + (s % CW);
    if (u > H) u = H;  // tail group: clamp (results of clamped slots are never stored)
    const int col = s < CW ? u : (N - u) % N;
    z[s * N + v] = Z[(size_t)v * N + col];
  }
  __syncthreads();
  lds_fft<N, 2 * CW>(z, 2 * CW, a.twiddles, tid);
  // normalised cross-power spectrum of bins (v, u), conjugated in place (see K1 for the rules)
  for (int i = tid; i < CW * N; i += SR_T) {
    const int s = i / N, v = i % N;
    int u = u0 + s;
    if (u > H) u = H;
    const cf zk = z[s * N + v], zm = z[(CW + s) * N + (N - v) % N];
    const bool real_only = (v == 0 || v == H) && (u == 0 || u == H);
    const cf C = cross_power(zk, zm, real_only);
    z[s * N + v] = {C.x, -C.y};
  }
  __syncthreads();
  lds_fft<N, 2 * CW>(z, CW, a.twiddles, tid);  // only the CW lines of the half spectrum
  cf* D = reinterpret_cast<cf*>(a.D) + (size_t)pair * N * (H + 1);
  for (int i = tid; i < CW * N; i += SR_T) {
    const int y = i / CW, s = i % CW, u = u0 + s;
    if (u <= H) D[(size_t)y * (H + 1) + u] = z[s * N + y];
  }
}

// ---- K7: Hermitian rows back to the real surface, two rows per complex transform ---------------------------
template <int N>
__global__ void __launch_bounds__(SR_T) sr_rows_inv_kernel(SrPcArgs a) {
  constexpr int H = N / 2;
  __shared__ cf z[INV_L * N];
  __shared__ Best red[SR_T / 64];
  const int tid = threadIdx.x, pair = blockIdx.y, p0 = blockIdx.x * INV_L;  // row pairs p0 .. p0+INV_L-1
  const cf* D = reinterpret_cast<const cf*>(a.D) + (size_t)pair * N * (H + 1);
  for (int i = tid; i < INV_L * N; i += SR_T) {
    const int l = i / N, u = i % N, y1 = 2 * (p0 + l), y2 = y1 + 1;
    const int uu = u <= H ? u : N - u;
    const cf f1 = D[(size_t)y1 * (H + 1) + uu], f2 = D[(size_t)y2 * (H + 1) + uu];
    // E[u] = F1[y1][u] + i F1[y2][u], F1[y][N-u] = conj F1[y][u]
    z[i] = u <= H ? cf{f1.x - f2.y, f1.y + f2.x} : cf{f1.x + f2.y, f2.x - f1.y};
  }
  __syncthreads();
  lds_fft<N, INV_L>(z, INV_L, a.twiddles, tid);
  float* S = a.S + (size_t)pair * N * N;
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  for (int i = tid; i < INV_L * N; i += SR_T) {
    const int l = i / N, x = i % N, y1 = 2 * (p0 + l), y2 = y1 + 1;
    const cf e = z[i];
    S[(size_t)y1 * N + x] = e.x;
    S[(size_t)y2 * N + x] = e.y;
    const int xs = (x + H) % N;
    best = better(best, Best{e.x, ((y1 + H) % N) * N + xs});  // fftShift + first maximum (minMaxLoc)
    best = better(best, Best{e.y, ((y2 + H) % N) * N + xs});
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if ((tid & 63) == 0) red[tid >> 6] = best;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < SR_T / 64; ++w) best = better(best, red[w]);
    a.cand[(size_t)pair * a.n_cand + blockIdx.x] = make_float2(best.v, __int_as_float(best.idx));
  }
}

// ---- K8: peak, centroid, (scale, rot) ----------------------------------------------------------------------
// out[pair] = {scale, rot, pt.x, pt.y}; pt = cv::phaseCorrelate(cur_lp, prev_lp) = center - t (NOT negated, :117);
// |pt.x| > res/2 -> (1, 0) (:119-121); scale = exp(pt.x / M), rot = (pt.y / Ky) pi/180, Ky = res/360 (:123-124).
template <int N>
__global__ void __launch_bounds__(64) sr_final_kernel(SrPcArgs a) {
  constexpr int H = N / 2;
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
  const float* S = a.S + (size_t)pair * N * N;
  const int px = best.idx % N, py = best.idx / N;
  const int ys = py - 2 + lane / 5, xs = px - 2 + lane % 5;
  double cx = 0.0, cy = 0.0, sum = 0.0;
  if (lane < 25 && ys >= 0 && ys <= N - 1 && xs >= 0 && xs <= N - 1) {
    const double val = (double)S[(size_t)((ys + H) % N) * N + (xs + H) % N];
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
  if (lane == 0) {
    sum += 2.220446049250313e-16;
    const double ptx = (double)N / 2.0 - cx / sum, pty = (double)N / 2.0 - cy / sum;
    double scale = 1.0, rot = 0.0;
    if (!(fabs(ptx) > (double)(N / 2))) {
      scale = exp(ptx / a.M);
      rot = (pty / ((double)N / 360.0)) * (3.14159265358979323846 / 180.0);
    }
    double* o = a.out + 4 * (size_t)pair;
    o[0] = scale;
    o[1] = rot;
    o[2] = ptx;
    o[3] = pty;
  }
}

bool sr_resolution_supported(int res) { return res == 240 || res == 256 || res == 480; }

template <int K>
static hipError_t launch_lp_lds(const SrLpArgs& a, int n_images, hipStream_t stream) {
  static int cus = 0;
  if (cus == 0) {
    int dev = 0;
    hipDeviceProp_t prop;
    cus = (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) ? prop.multiProcessorCount : 256;
  }
  hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sr_logpolar_lds_kernel<K>),
                                     hipFuncAttributeMaxDynamicSharedMemorySize, (int)LpLds<K>::BYTES);
  if (e != hipSuccess) return e;
  const int tiles = (a.res + 7) / 8;
  const long total = (long)n_images * tiles * tiles;
  const unsigned blocks = (unsigned)((total + 15) / 16 < cus ? (total + 15) / 16 : cus);
  hipLaunchKernelGGL(sr_logpolar_lds_kernel<K>, dim3(blocks), dim3(1024), LpLds<K>::BYTES, stream, a, n_images);
  return hipGetLastError();
}

hipError_t launch_sr_logpolar(const SrLpArgs& a, int interp, int n_images, hipStream_t stream) {
  // MOF_SR_LP_GLOBAL=1: the first formulation (weights fetched from L2 per pixel), kept for A/B measurements; it also
  // serves single images, where staging the table would cost more than it saves
  static const bool global_w = [] { const char* e = getenv("MOF_SR_LP_GLOBAL"); return e && atoi(e) != 0; }();
  if (!global_w && n_images >= 4) return interp == 2 ? launch_lp_lds<4>(a, n_images, stream) : launch_lp_lds<8>(a, n_images, stream);
  const dim3 grid((unsigned)(((a.res + 31) / 32) * ((a.res + 7) / 8)), (unsigned)n_images);
  if (interp == 2)
    hipLaunchKernelGGL(sr_logpolar_kernel<4>, grid, dim3(256), 0, stream, a);
  else
    hipLaunchKernelGGL(sr_logpolar_kernel<8>, grid, dim3(256), 0, stream, a);
  return hipGetLastError();
}

template <int N>
static hipError_t launch_sr_pc_n(const SrPcArgs& a, int n_pairs, hipStream_t stream) {
  constexpr int H = N / 2;
  hipLaunchKernelGGL(sr_rows_fwd_kernel<N>, dim3(N / ROWS_L, (unsigned)n_pairs), dim3(SR_T), 0, stream, a);
  hipLaunchKernelGGL(sr_cols_kernel<N>, dim3((H + 1 + COLS_CW - 1) / COLS_CW, (unsigned)n_pairs), dim3(SR_T), 0, stream, a);
  hipLaunchKernelGGL(sr_rows_inv_kernel<N>, dim3(H / INV_L, (unsigned)n_pairs), dim3(SR_T), 0, stream, a);
  hipLaunchKernelGGL(sr_final_kernel<N>, dim3((unsigned)n_pairs), dim3(64), 0, stream, a);
  return hipGetLastError();
}

int sr_candidates(int res) { return (res / 2) / INV_L; }

hipError_t launch_sr_phase_correlate(const SrPcArgs& a, int res, int n_pairs, hipStream_t stream) {
  switch (res) {
    case 240: return launch_sr_pc_n<240>(a, n_pairs, stream);
    case 256: return launch_sr_pc_n<256>(a, n_pairs, stream);
    case 480: return launch_sr_pc_n<480>(a, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mof
