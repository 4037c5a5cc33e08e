// pc_kernel.hip -- K1: fused per-patch FFT phase correlation for gfx950 (CDNA4).
//
// One workgroup owns one patch pair and never leaves the CU: the two u8 patches are read
// once from HBM (16 B per lane, any byte alignment), packed as z = cur + i*prev into one
// complex N x N tile in LDS, transformed with ONE complex 2-D FFT ("two-for-one" real
// transform), untangled into the two real spectra, turned into the normalised cross-power
// spectrum, inverse-transformed in place, and reduced to (arg-max, 5x5 centroid) -- 16 B
// leave the CU per patch. Nothing but the frames and the results touches HBM.
//
// Replaces, per patch, the reference's
//   -cv::phaseCorrelate(cur(roi), prev(roi))              src/FftMethod.cpp:1836
// whose stages the reference spells out at src/FftMethod.cpp:1487-1498, with the helper
// semantics of :70-168 (magSpectrums), :1086-1251 (divSpectrums), :1257-1323 (fftShift),
// :1329-1385 (weightedCentroid), followed by the validity gate of :1838-1856.
//
// Arithmetic: fp32 transforms with twiddles computed in double on the host (as OpenCV does
// for CV_32F), fp64 centroid and gate. Algorithmic HBM bytes per patch: 2*N*N in + 16 out.

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "mof_kernels.h"

namespace mof {

namespace {

struct cf {
  float x, y;
};

__device__ __forceinline__ cf cmul(cf a, cf b) { return {a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x}; }
__device__ __forceinline__ cf cadd(cf a, cf b) { return {a.x + b.x, a.y + b.y}; }
__device__ __forceinline__ cf csub(cf a, cf b) { return {a.x - b.x, a.y - b.y}; }
// multiply by -i (a quarter turn of the forward kernel e^{-2 pi i k/N})
__device__ __forceinline__ cf mul_mi(cf a) { return {a.y, -a.x}; }

template <int R>
__device__ __forceinline__ void butterfly(cf* v);

template <>
__device__ __forceinline__ void butterfly<2>(cf* v) {
  cf a = v[0], b = v[1];
  v[0] = cadd(a, b);
  v[1] = csub(a, b);
}

template <>
__device__ __forceinline__ void butterfly<4>(cf* v) {
  cf a = cadd(v[0], v[2]), b = csub(v[0], v[2]);
  cf c = cadd(v[1], v[3]), d = mul_mi(csub(v[1], v[3]));
  v[0] = cadd(a, c);
  v[1] = cadd(b, d);
  v[2] = csub(a, c);
  v[3] = csub(b, d);
}

template <>
__device__ __forceinline__ void butterfly<8>(cf* v) {
  const float h = 0.70710678118654752440f;
  // three radix-2 layers, decimation in time on the 8 inputs
  cf e[4] = {v[0], v[2], v[4], v[6]};
  cf o[4] = {v[1], v[3], v[5], v[7]};
  butterfly<4>(e);
  butterfly<4>(o);
  cf w1 = {h * (o[1].x + o[1].y), h * (o[1].y - o[1].x)};   // o1 * e^{-i pi/4}
  cf w2 = mul_mi(o[2]);                                     // o2 * e^{-i pi/2}
  cf w3 = {h * (o[3].y - o[3].x), -h * (o[3].x + o[3].y)};  // o3 * e^{-3 i pi/4}
  v[0] = cadd(e[0], o[0]);
  v[4] = csub(e[0], o[0]);
  v[1] = cadd(e[1], w1);
  v[5] = csub(e[1], w1);
  v[2] = cadd(e[2], w2);
  v[6] = csub(e[2], w2);
  v[3] = cadd(e[3], w3);
  v[7] = csub(e[3], w3);
}

// One Stockham (auto-sort, decimation-in-time) stage of radix R over `LINES` independent
// length-N lines held in LDS; element e of line l lives at z[l*LS + e*ES]. P = product of
// the radices already applied. All T threads take part; the caller provides the barriers'
// surroundings (data ready on entry, data ready on exit).
template <int N, int R, int P, int T, int LS, int ES>
__device__ __forceinline__ void stockham_stage(cf* __restrict__ z, const cf* __restrict__ tw, int tid) {
  constexpr int BPL = N / R;        // butterflies per line
  constexpr int TOTAL = N * BPL;    // butterflies in the tile
  constexpr int PER = TOTAL / T;    // per thread
  static_assert(TOTAL % T == 0, "tile must divide evenly");
  cf v[PER][R];
  int dst[PER];
#pragma unroll
  for (int b = 0; b < PER; ++b) {
    const int g = tid + b * T;
    const int line = g / BPL, x = g % BPL;
    const int j = x % P;
    const cf* src = z + line * LS + x * ES;
#pragma unroll
    for (int k = 0; k < R; ++k) {
      cf a = src[k * BPL * ES];
      if (P > 1 && k > 0) a = cmul(a, tw[(k * j) * (N / (P * R))]);
      v[b][k] = a;
    }
    butterfly<R>(v[b]);
    dst[b] = line * LS + ((x - j) * R + j) * ES;
  }
  __syncthreads();
#pragma unroll
  for (int b = 0; b < PER; ++b)
#pragma unroll
    for (int k = 0; k < R; ++k) z[dst[b] + k * P * ES] = v[b][k];
  __syncthreads();
}

template <int N, int T, int LS, int ES>
__device__ __forceinline__ void fft_lines(cf* z, const cf* tw, int tid) {
  if constexpr (N == 32) {
    stockham_stage<N, 8, 1, T, LS, ES>(z, tw, tid);
    stockham_stage<N, 4, 8, T, LS, ES>(z, tw, tid);
  } else if constexpr (N == 64) {
    stockham_stage<N, 8, 1, T, LS, ES>(z, tw, tid);
    stockham_stage<N, 8, 8, T, LS, ES>(z, tw, tid);
  } else {
    static_assert(N == 128, "supported patch sizes: 32, 64, 128");
    stockham_stage<N, 8, 1, T, LS, ES>(z, tw, tid);
    stockham_stage<N, 8, 8, T, LS, ES>(z, tw, tid);
    stockham_stage<N, 2, 64, T, LS, ES>(z, tw, tid);
  }
}

struct Best {
  float v;
  int idx;
};
__device__ __forceinline__ Best better(Best a, Best b) {
  // first maximum in row-major order of the fft-shifted surface (cv::minMaxLoc)
  return (b.v > a.v || (b.v == a.v && b.idx < a.idx)) ? b : a;
}

}  // namespace

template <int N>
struct PcTraits {
  static constexpr int T = (N * N / 16 > 1024) ? 1024 : (N * N / 16 < 64 ? 64 : N * N / 16);
  static constexpr int PITCH = N + 1;  // complex elements per LDS row (odd: column walks hit all banks)
  static constexpr size_t LDS_BYTES = sizeof(float) * 2 * (size_t)(N * PITCH + N) + 64 * 8;
};

template <int N>
__global__ void __launch_bounds__(PcTraits<N>::T) pc_field_kernel(PcArgs a) {
  constexpr int T = PcTraits<N>::T;
  constexpr int PITCH = PcTraits<N>::PITCH;
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  cf* z = reinterpret_cast<cf*>(smem);
  cf* tw = z + N * PITCH;
  Best* red = reinterpret_cast<Best*>(tw + N);

  const int tid = threadIdx.x;
  const int patches = a.grid_x * a.grid_y;
  const int pair = blockIdx.x / patches;
  const int patch = blockIdx.x % patches;
  const int pi = patch % a.grid_x, pj = patch / a.grid_x;
  const int x0 = a.origin_x + pi * a.stride_x;
  const int y0 = a.origin_y + pj * a.stride_y;
  const uint8_t* cur = a.cur + (size_t)pair * a.cur_stride + (size_t)y0 * a.pitch + x0;
  const uint8_t* prev = a.prev + (size_t)pair * a.prev_stride + (size_t)y0 * a.pitch + x0;

  // twiddles W_N^k (host-computed in double)
  for (int k = tid; k < N; k += T) tw[k] = {a.twiddles[2 * k], a.twiddles[2 * k + 1]};

  // ---- load: 16 B per lane per image, u8 -> f32 (exact), z = cur + i*prev  (convertTo, :1805-1806)
  constexpr int CHUNKS = N * N / 16;
  for (int c = tid; c < CHUNKS; c += T) {
    const int row = c / (N / 16), col = (c % (N / 16)) * 16;
    uint32_t cw[4], pw[4];
    __builtin_memcpy(cw, cur + (size_t)row * a.pitch + col, 16);
    __builtin_memcpy(pw, prev + (size_t)row * a.pitch + col, 16);
    cf* dst = z + row * PITCH + col;
#pragma unroll
    for (int q = 0; q < 4; ++q)
#pragma unroll
      for (int b = 0; b < 4; ++b)
        dst[q * 4 + b] = {(float)((cw[q] >> (8 * b)) & 0xffu), (float)((pw[q] >> (8 * b)) & 0xffu)};
  }
  __syncthreads();

  // ---- forward 2-D transform of z: rows, then columns  (dft x2, :1491-1493)
  fft_lines<N, T, PITCH, 1>(z, tw, tid);
  fft_lines<N, T, 1, PITCH>(z, tw, tid);

  // ---- untangle A = FFT(cur), B = FFT(prev); P = A conj(B); C = P|P| / (|P|^2 + eps)
  //      (mulSpectrums :1494, magSpectrums :70-168, divSpectrums :1086-1251, incl. the
  //      real-only-slot behaviour of :107-109 / :1127-1129). conj(C) is stored so that the
  //      same forward transform yields the unscaled inverse's real part.
  {
    constexpr int H = N / 2;
    const float eps = 1.1920928955078125e-07f;  // FLT_EPSILON, :1117
    for (int g = tid; g < (H + 1) * N; g += T) {
      const int v = g / N, u = g % N;
      if ((v == 0 || v == H) && u > H) continue;  // partner lies in the same row; done by u <= H
      const int vm = (N - v) % N, um = (N - u) % N;
      const cf zk = z[v * PITCH + u], zm = z[vm * PITCH + um];
      // A[k] = (Z[k] + conj(Z[-k]))/2 ; B[k] = (Z[k] - conj(Z[-k]))/(2i)
      const cf A = {0.5f * (zk.x + zm.x), 0.5f * (zk.y - zm.y)};
      const cf B = {0.5f * (zk.y + zm.y), 0.5f * (zm.x - zk.x)};
      cf C;
      if (v == vm && u == um) {
        const float p = A.x * B.x;
        C = {p / (p * p + eps), 0.f};
      } else {
        const float pr = A.x * B.x + A.y * B.y;
        const float pim = A.y * B.x - A.x * B.y;
        const float mag = sqrtf(pr * pr + pim * pim);
        const float den = mag * mag + eps;
        C = {(pr * mag) / den, (pim * mag) / den};
      }
      z[v * PITCH + u] = {C.x, -C.y};      // conj(C[k])
      z[vm * PITCH + um] = {C.x, C.y};     // conj(C[-k]) = conj(conj(C[k]))
    }
  }
  __syncthreads();

  // ---- inverse (unscaled) via forward transform of conj(C): Re(result) = idft(C)  (:1497)
  fft_lines<N, T, PITCH, 1>(z, tw, tid);
  fft_lines<N, T, 1, PITCH>(z, tw, tid);

  // ---- arg-max of the fft-shifted surface, first occurrence (fftShift :1297-1305, minMaxLoc :1539)
  constexpr int H = N / 2;
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  for (int g = tid; g < N * N; g += T) {
    const int y = g / N, x = g % N;                 // un-shifted position
    const int ys = (y + H) % N, xs = (x + H) % N;   // position after fftShift
    best = better(best, Best{z[y * PITCH + x].x, ys * N + xs});
  }
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if ((tid & 63) == 0) red[tid >> 6] = best;
  __syncthreads();

  // ---- 5x5 weighted centroid in double + validity gate, one lane  (:1337-1383, :1838-1856)
  if (tid == 0) {
    for (int w = 1; w < T / 64; ++w) best = better(best, red[w]);
    const int px = best.idx % N, py = best.idx / N;
    int minr = py - 2, maxr = py + 2, minc = px - 2, maxc = px + 2;
    if (minr < 0) minr = 0;
    if (minc < 0) minc = 0;
    if (maxr > N - 1) maxr = N - 1;
    if (maxc > N - 1) maxc = N - 1;
    double cx = 0.0, cy = 0.0, sum = 0.0;
    for (int ys = minr; ys <= maxr; ++ys)
      for (int xs = minc; xs <= maxc; ++xs) {
        const double val = (double)z[((ys + H) % N) * PITCH + ((xs + H) % N)].x;
        cx += (double)xs * val;
        cy += (double)ys * val;
        sum += val;
      }
    sum += 2.220446049250313e-16;  // DBL_EPSILON, :1378
    // shift = -(center - t) = t - N/2   (:1836)
    double sx = cx / sum - (double)N / 2.0;
    double sy = cy / sum - (double)N / 2.0;
    const bool bad = (sx * sx + sy * sy > a.max_px_speed_sq) || (fabs(sx) > (double)N / 2.0) ||
                     (fabs(sy) > (double)N / 2.0) || (sx != sx) || (sy != sy);
    if (bad) sx = sy = __builtin_nan("");
    double* o = a.out + 2 * ((size_t)pair * patches + patch);
    o[0] = sx;
    o[1] = sy;
  }
}

template <int N>
static hipError_t configure_n() {
  return hipFuncSetAttribute(reinterpret_cast<const void*>(&pc_field_kernel<N>),
                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)PcTraits<N>::LDS_BYTES);
}

template <int N>
static hipError_t launch_n(const PcArgs& a, int n_pairs, hipStream_t stream) {
  using Tr = PcTraits<N>;
  const unsigned blocks = (unsigned)n_pairs * (unsigned)(a.grid_x * a.grid_y);
  hipLaunchKernelGGL(pc_field_kernel<N>, dim3(blocks), dim3(Tr::T), Tr::LDS_BYTES, stream, a);
  return hipGetLastError();
}

hipError_t pc_configure(int patch_size) {
  switch (patch_size) {
    case 32: return configure_n<32>();
    case 64: return configure_n<64>();
    case 128: return configure_n<128>();
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_pc_field(const PcArgs& a, int patch_size, int n_pairs, hipStream_t stream) {
  switch (patch_size) {
    case 32: return launch_n<32>(a, n_pairs, stream);
    case 64: return launch_n<64>(a, n_pairs, stream);
    case 128: return launch_n<128>(a, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}

bool pc_patch_size_supported(int n) { return n == 32 || n == 64 || n == 128; }

}  // namespace mof
