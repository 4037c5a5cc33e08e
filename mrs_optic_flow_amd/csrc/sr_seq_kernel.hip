// sr_seq_kernel.hip -- the scale/rotation estimator's transforms for FRAME SEQUENCES (K5s, K6s) on gfx950.
//
// scaleRotationEstimator::processImage (/root/reference/src/scaleRotationEstimator.cpp:34-148) is a stream processor: from
// the second call on, every frame's log-polar image is `cur` once (:112-117) and -- unless the gate of :119-121 fires --
// `prev` once (:128). The pair kernels of sr_kernel.hip pack (cur, prev) into one complex transform, which is the cheapest
// form for INDEPENDENT pairs but transforms every frame of a video twice. Here the unit is the frame:
//   K5s sr_rows_real : one log-polar image -> its row half-spectra, two real rows per complex transform, untangled and
//                      written transposed and DOUBLED: Zh[u][v] = 2 * rowDFT(v)[u], u = 0..N/2 (925 KB per 480^2 frame
//                      instead of 1.84 MB of packed Zt per pair)
//   K6s sr_cols_seq  : ONE WAVE owns four columns u and walks a run of consecutive pairs in time: the column spectra of
//                      the previous frame stay in its registers (60 VGPRs), per new frame it reads four lines of Zh,
//                      transforms them in LDS, forms the normalised cross-power spectrum against the registers (same
//                      rules as K6 / K1, pc_common.hpp), transforms back and writes the same Dt that K7 / K8 of
//                      sr_kernel.hip consume. No workgroup barrier; per pair and column group 8 line transforms instead of
//                      K6's 12, and each Zh line is read (1 + 1/run) times instead of twice.
// The stateful single-frame entry (mof_sr_process) runs the SAME kernels with one pair per launch, so a sequence processed
// in one call and the same frames fed one at a time produce identical bits.

#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"
#include "sr_common.hpp"

namespace mof {

namespace {

// ---- K5s: row half-spectra of one real image -------------------------------------------------------------------------
// ROWS image rows per workgroup = ROWS / 2 packed lines, four lines per wave. The transposed store writes ROWS * 8
// contiguous bytes per u.
#ifndef MOF_K5S_ROTATE
#define MOF_K5S_ROTATE 1
#endif
#ifndef MOF_K5S_ROWS32  // 32 image rows per workgroup where N allows (256-byte segments of the transposed store); 0: 16 rows (more workgroups per CU)
#define MOF_K5S_ROWS32 1
#endif
template <int N>
struct RowsReal {
  static constexpr int ROWS = N > 512 ? 8 : ((N % 32 == 0 && MOF_K5S_ROWS32) ? 32 : (N % 16 == 0 ? 16 : 8));  // (200: 25 one-wave workgroups of 8 rows per image; beyond 512: one wave = four lines = up to 31 KB of LDS per workgroup)
  static constexpr bool TAIL = N % ROWS != 0;  // (270, 300, 450: the last one-wave workgroup holds 3, 2 or 1 row pairs; the rest of its lines are zeros and are not stored)
  static_assert((N % ROWS == 0 || ROWS == 8) && ROWS % 8 == 0, "whole workgroups of four-line waves, a tail only behind one-wave workgroups");  // (odd N: the last row shares its line with zeros)
  static constexpr int GROUPS = (N + ROWS - 1) / ROWS;
  static constexpr int LINES = ROWS / 2;
  // lines per wave: four. (r06 A/B, -DMOF_K5S_LPW2_FROM=540: two lines per wave at the long lines -- twice the waves for the same LDS, as K7 took
  // it -- LOSES here: 576 +9 %, 640 +17 % time, profiles/r06_lpw2_ab.txt; off)
#ifndef MOF_K5S_LPW2_FROM
#define MOF_K5S_LPW2_FROM 1000000
#endif
  static constexpr int LPW = N >= MOF_K5S_LPW2_FROM ? 2 : 4;
  static constexpr int T = LINES / LPW * 64;
};

template <int N>
__global__ void __launch_bounds__(RowsReal<N>::T) sr_rows_real_kernel(const uint8_t* __restrict__ lp, size_t lp_stride,
                                                                       const float* __restrict__ twiddles,
                                                                       float* __restrict__ zh, size_t zh_stride) {
  using P = SrPlan<N>;
  using R = RowsReal<N>;
  constexpr int H = N / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char rr_lds[];
  cf* z = reinterpret_cast<cf*>(rr_lds);  // [LINES][LINE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, frame = blockIdx.y, row0 = blockIdx.x * R::ROWS;
  SrTw<N> tw;
  tw.load(twiddles, lane);
  // wave w owns rows row0 + 8w .. +7 = lines 4w .. 4w+3: line l = rows (2l, 2l+1) as real and imaginary part
  const uint8_t* img = lp + (size_t)frame * lp_stride + (size_t)(row0 + 8 * wave) * N;
  cf* mine = z + 4 * wave * P::LINE;
  {
    constexpr int ND = N / 4, NL = (4 * ND + 63) / 64;  // dwords per row; (even, odd) dword pairs per lane
    uint32_t c[NL], p[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {  // all loads first, then the conversion (convertTo CV_32FC1, :115)
      const int i = lane + 64 * k;
      if (i < 4 * ND) {
        const int l = i / ND, d = i % ND;
        c[k] = stream_load(reinterpret_cast<const uint32_t*>(img + (size_t)(2 * l) * N + 4 * d));
        p[k] = stream_load(reinterpret_cast<const uint32_t*>(img + (size_t)(2 * l + 1) * N + 4 * d));
      }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = lane + 64 * k;
      if (i < 4 * ND) {
        const int l = i / ND, d = i % ND;
        // (a lane's four pixels go out in the order rotated by d / 4: lanes 4 apart -- 32 B x 4 = one bank wrap -- would otherwise
        //  hit the same bank with every one of their four stores: 38 % of this kernel's LDS cycles were conflicts, r04 counters)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
          const int r = MOF_K5S_ROTATE ? (q + (d >> 2)) & 3 : q;
          mine[l * P::LINE + 4 * d + r] = {(float)((c[k] >> (8 * r)) & 0xffu), (float)((p[k] >> (8 * r)) & 0xffu)};
        }
      }
    }
  }
  wave_sync();
  wave_fft<N>(mine, 4, lane, tw, StoreNatural<N>{});
  __syncthreads();
  // untangle the two rows of every line and store transposed: Zh[u][row0 + 2j], Zh[u][row0 + 2j + 1] are neighbours
  cf* out = reinterpret_cast<cf*>(zh + (size_t)(MOF_SR_L2_ABLATE ? 0 : frame) * zh_stride) + row0;
  for (int i = tid; i < R::LINES * (H + 1); i += R::T) {
    const int u = i / R::LINES, j = i % R::LINES;
    const cf zk = z[j * P::LINE + u], zm = z[j * P::LINE + (N - u) % N];
    cf a2, b2;
    untangle2(zk, zm, &a2, &b2);  // 2 * DFT(row 2j)[u], 2 * DFT(row 2j + 1)[u]
    stream_store(reinterpret_cast<float4*>(out + (size_t)u * sr_zh_pitch<N>() + 2 * j), make_float4(a2.x, a2.y, b2.x, b2.y));
  }
}

// ---- K5s on PATCHES OF FRAMES (r04): FftMethod patches of 240 / 256 / 480 pixels -- the reference's whole-frame fallback
// (FftMethod.cpp:1709-1716) among them -- are exactly this estimator's transform sizes, so the large-patch pipeline of the FFT
// engine (pc_large_kernel.hip) hands them to the tuned K5s / K6s / K7 instead of its planned L5 / L6 / L7. The only difference
// to sr_rows_real_kernel is where the pixels come from (image f = 2 (pair * patches + patch) + (0 cur | 1 prev), any row pitch,
// gray or BGR8 through the node's CV_RGB2GRAY) and the constant-image flags the FFT tail wants (as pcl_rows_kernel sets them).
template <int N, int CH, bool PAD>  // PAD: n < N, the patch is zero-padded to the transform size (the unpadded form pays nothing for it)
__global__ void __launch_bounds__(RowsReal<N>::T) sr_rows_real_src_kernel(PclSrc src, const float* __restrict__ twiddles,
                                                                           float* __restrict__ zh, size_t zh_stride, int* __restrict__ flags, int n,
                                                                           int* __restrict__ sums) {
  using P = SrPlan<N>;
  using R = RowsReal<N>;
  constexpr int H = N / 2;
  extern __shared__ __attribute__((aligned(16))) unsigned char rr_lds[];
  cf* z = reinterpret_cast<cf*>(rr_lds);  // [LINES][LINE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, img = blockIdx.y, row0 = blockIdx.x * R::ROWS;
  const int patches = src.grid_x * src.grid_y;
  const int which = src.paired == 2 ? 0 : (img & 1), q = src.paired == 2 ? img : (img >> 1);  // (a video: image = frame * patches + patch, `pair` = the frame)
  const int pair = q / patches, pt = q - pair * patches, by = pt / src.grid_x, bx = pt - by * src.grid_x;
  const uint8_t* base = src.base[which] + (size_t)pair * src.stride[which] + (size_t)(src.origin_y + by * src.stride_y) * src.pitch +
                        (size_t)(CH * (src.origin_x + bx * src.stride_x));
  SrTw<N> tw;
  tw.load(twiddles, lane);
  // r06: n < N -- a patch that cv::phaseCorrelate zero-pads to this transform size (copyMakeBorder): rows and columns beyond n are zeros
  // and take no part in the constant-image test (`inside`)
  // EDGE: some chunk of some line may lie (partly) outside the pixels to read -- a padded patch, a row length that is not a multiple of
  // four (270, 450: the last chunk of a row holds two pixels), or the zero lines behind the image in the last workgroup (RowsReal::TAIL)
  constexpr bool EDGE = PAD || (N % 4 != 0) || R::TAIL;
  const int ne = PAD ? n : N;
  auto px4 = [&](int y, int d) -> uint32_t {  // pixels 4d .. 4d+3 of patch row y, one byte each
    // EDGE, branch-free (r06): the load itself is unconditional -- row min(y, ne - 1), the four pixels that END no later than the row does --
    // and what lies outside the patch is shifted / masked away afterwards. With a branch per chunk (rows past the patch, the half-full last
    // chunk in a byte loop) every chunk's load waited for the previous one's: the sizes with a tail (250, 270, 300, 450) ran 1.7 x slower per
    // pixel than their neighbours (tools/size_probe.py).
    const int yc = EDGE ? (y < ne ? y : ne - 1) : y;
    const int xc = EDGE ? (4 * d < ne - 4 ? 4 * d : ne - 4) : 4 * d;
    const uint8_t* r = base + (size_t)yc * src.pitch + (size_t)CH * xc;
    uint32_t v;
    if constexpr (CH == 1) {
      __builtin_memcpy(&v, r, 4);  // (any alignment: the patch origin and the pitch are the caller's)
    } else {
      v = 0;
#pragma unroll
      for (int b = 0; b < 4; ++b) v |= rgb2gray_fixed(r[3 * b], r[3 * b + 1], r[3 * b + 2]) << (8 * b);
    }
    if constexpr (EDGE) {
      const int sh = 4 * d - xc;  // 0: a whole chunk; 1 .. 3: the last chunk of a row whose length is not a multiple of four; >= 4: past the row
      v = sh >= 4 ? 0u : v >> (8 * (sh & 3));
      v = y < ne ? v : 0u;
    }
    return v;
  };
  auto inside = [&](int y, int d) -> uint32_t {  // byte mask of the chunk's pixels that lie inside the n x n patch
    if (!EDGE) return 0xffffffffu;
    if (y >= ne || 4 * d >= ne) return 0u;
    return 4 * d + 3 < ne ? 0xffffffffu : (1u << (8 * (ne - 4 * d))) - 1u;
  };
  const uint32_t p00 = px4(0, 0) & 0xffu, pat = p00 * 0x01010101u;
  uint32_t diff = 0u;
  int s00 = 0, s01 = 0, s10 = 0, s11 = 0;
  constexpr int LPW = R::LPW;  // lines of this wave
  cf* mine = z + LPW * wave * P::LINE;
  {
    constexpr int ND = (N + 3) / 4, NL = (LPW * ND + 63) / 64;  // (N = 270, 450: the last chunk of a row is half full)
    uint32_t c[NL], p[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = lane + 64 * k;
      if (i < LPW * ND) {
        const int l = i / ND, d = i % ND, y = row0 + 2 * LPW * wave + 2 * l;
        c[k] = px4(y, d);
        p[k] = px4(y + 1, d);
      }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = lane + 64 * k;
      if (i < LPW * ND) {
        const int l = i / ND, d = i % ND, y = row0 + 2 * LPW * wave + 2 * l;
        diff |= ((c[k] ^ pat) & inside(y, d)) | ((p[k] ^ pat) & inside(y + 1, d));
        if constexpr (!SrNyqExact<P>::value) {
          // the four exact integer sums of the image (pixels outside the patch were loaded as zeros): row y is even, y + 1 odd; byte b of a chunk
          // is column 4 d + b, so the byte parity is the column parity
          const int ce = (int)((c[k] & 0xffu) + ((c[k] >> 16) & 0xffu)), co = (int)(((c[k] >> 8) & 0xffu) + (c[k] >> 24));
          const int pe = (int)((p[k] & 0xffu) + ((p[k] >> 16) & 0xffu)), po = (int)(((p[k] >> 8) & 0xffu) + (p[k] >> 24));
          s00 += ce + co + pe + po;
          s01 += ce - co + pe - po;  // (-1)^x
          s10 += ce + co - pe - po;  // (-1)^y
          s11 += ce - co - pe + po;  // (-1)^(x + y)
        }
#pragma unroll
        for (int b = 0; b < 4; ++b)
          mine[l * P::LINE + 4 * d + b] = {(float)((c[k] >> (8 * b)) & 0xffu), (float)((p[k] >> (8 * b)) & 0xffu)};
      }
    }
  }
  if constexpr (!SrNyqExact<P>::value) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
      s00 += __shfl_xor(s00, off, 64);
      s01 += __shfl_xor(s01, off, 64);
      s10 += __shfl_xor(s10, off, 64);
      s11 += __shfl_xor(s11, off, 64);
    }
#ifndef MOF_SR_SUMS_ABLATE  // (diagnostic build: no atomics -- results wrong by design)
    if (lane == 0 && sums) {  // (zeroed by the caller; 255 * 432^2 < 2^31)
      int* q = sums + (size_t)(src.sums_stride ? src.sums_stride : 4) * img;
      atomicAdd(&q[0], s00);
      atomicAdd(&q[1], s01);
      atomicAdd(&q[2], s10);
      atomicAdd(&q[3], s11);
    }
#endif
  }
  if (flags) {  // bit 0: some pixel differs from pixel (0, 0); bit 1: pixel (0, 0) is not zero (zeroed by the caller)
    if (__builtin_amdgcn_ballot_w64(diff != 0u) != 0ull && lane == 0) atomicOr(&flags[img], 1);
    if (blockIdx.x == 0 && tid == 0 && p00 != 0u) atomicOr(&flags[img], 2);
  }
  wave_sync();
  wave_fft<N>(mine, LPW, lane, tw, StoreNatural<N>{});
  __syncthreads();
  cf* out = reinterpret_cast<cf*>(zh + (size_t)img * zh_stride) + row0;
  for (int i = tid; i < R::LINES * (H + 1); i += R::T) {
    const int u = i / R::LINES, j = i % R::LINES;
    if (R::TAIL && row0 + 2 * j >= N) continue;
    const cf zk = z[j * P::LINE + u], zm = z[j * P::LINE + (N - u) % N];
    cf a2, b2;
    untangle2(zk, zm, &a2, &b2);
#ifdef MOF_SR_ABL_ROWSTORE  // (diagnostic build, results wrong by design: what the row kernel's stores cost)
    if (a2.x == 123456.f)
#endif
    stream_store(reinterpret_cast<float4*>(out + (size_t)u * sr_zh_pitch<N>() + 2 * j), make_float4(a2.x, a2.y, b2.x, b2.y));
  }
}

// ---- K6s: column transforms + cross-power + inverse columns, one wave walking a run of pairs -------------------------
#ifndef MOF_SEQ_CW
#define MOF_SEQ_CW 4
#endif
constexpr int SEQ_CW = MOF_SEQ_CW;  // columns per wave
// (r06) the long lines: two columns per wave -- half the LDS, half of the previous-image spectra in registers (from 640 on four columns put
// those in AGPRs at one wave per SIMD: VALU 22 % / LDS 23 % at 720, profiles/r06_l720_sq_pmc.csv), and with a first radix above 16 the second
// stage takes two lines per pass anyway. MOF_SEQ_CW_BIG_FROM: first N of the two-column form (A/B)
#ifndef MOF_SEQ_CW_BIG_FROM
#define MOF_SEQ_CW_BIG_FROM 540
#endif
template <int N>
constexpr int seq_cw() {
  // (two columns wherever the first radix is above 16 -- every N >= 540, and 324 / 486 / 500: 324 +10 %, 500 +5 %, 486 -1 %)
  return (SrPlan<N>::R1 > 16 || N >= MOF_SEQ_CW_BIG_FROM) ? 2 : SEQ_CW;
}

template <int N, bool BOX = false>  // BOX: patches zero-padded to N -- the box-zero rule of padded CONSTANT patches (the plain form pays nothing for it)
__global__ void __launch_bounds__(64) sr_cols_seq_kernel(const float* __restrict__ zh_prev, const float* __restrict__ zh_cur,
                                                         size_t zh_stride, const float* __restrict__ twiddles,
                                                         float* __restrict__ Dt, int n_pairs, int run, const int* __restrict__ flags, int n,
                                                         const int* __restrict__ sums_prev, const int* __restrict__ sums_cur, int sums_stride) {
  using P = SrPlan<N>;
  constexpr int H = N / 2, CW = seq_cw<N>();
  constexpr int MV = (N + 63) / 64;       // bins per lane and line (v = lane + 64 m)
  constexpr bool ODD = (N & 1) != 0;      // (r06) no Nyquist bin: only bin (0, 0) is real-only; Dt rows are not 16-byte aligned
  constexpr int NQ = (N + 1) / 2;         // 16-byte pieces of a line (odd N: the last one carries an element past the line)
  constexpr int MQ = (NQ + 63) / 64;      // 16-byte pieces per lane and line
  __shared__ cf z[CW * P::LINE];
  const int lane = threadIdx.x, u0 = blockIdx.x * CW, p0 = blockIdx.y * run;
  const int np = n_pairs - p0 < run ? n_pairs - p0 : run;
  SrTw<N> tw;
  tw.load(twiddles, lane);

  // four lines of one frame's Zh (columns u0 .. u0+3, clamped in the tail group) -> LDS -> column transforms in place
  auto load_cols = [&](const float* frame) {
    const cf* Zf = reinterpret_cast<const cf*>(frame);
    float4 t[CW][MQ];
#pragma unroll
    for (int s = 0; s < CW; ++s) {
      const int u = u0 + s > H ? H : u0 + s;
#pragma unroll
      for (int m = 0; m < MQ; ++m) {
        const int q = lane + 64 * m;
        if (q < NQ) t[s][m] = stream_load(reinterpret_cast<const float4*>(Zf + (size_t)u * sr_zh_pitch<N>() + 2 * q));
      }
    }
#pragma unroll
    for (int s = 0; s < CW; ++s)
#pragma unroll
      for (int m = 0; m < MQ; ++m) {
        const int q = lane + 64 * m;
        if (q < NQ) {
          z[s * P::LINE + 2 * q] = {t[s][m].x, t[s][m].y};
          z[s * P::LINE + 2 * q + 1] = {t[s][m].z, t[s][m].w};
        }
      }
    wave_sync();
    wave_fft<N>(z, CW, lane, tw, StoreNatural<N>{});
  };

#ifndef MOF_K6S_ABLATE  // diagnostic build (results wrong by design): 1 = every wave reads pair 0's lines (L2 hits instead of HBM reads)
#define MOF_K6S_ABLATE 0
#endif
  cf ap[CW][MV];  // column spectra of the previous frame (doubled): 2 B[v][u]
  load_cols(zh_prev + (size_t)((MOF_K6S_ABLATE != 0 || MOF_SR_L2_ABLATE != 0) ? 0 : p0) * zh_stride);
#pragma unroll
  for (int s = 0; s < CW; ++s)
#pragma unroll
    for (int m = 0; m < MV; ++m) {
      const int v = lane + 64 * m;
      ap[s][m] = v < N ? lds_read(&z[s * P::LINE + v]) : cf{0.f, 0.f};
    }
  wave_sync();
  // r06, FftMethod patches zero-padded to this transform size (n < N; `flags` as pcl_rows_kernel sets them, image 2 p = cur, 2 p + 1 = prev): a
  // CONSTANT patch became an n x n box whose spectrum is exactly zero on the multiples of box_zero_period(n, N) -- the rows were transformed
  // in pairs, their alternating sums carry rounding, and the normalisation would blow those bins up to unit magnitude: C = 0 there, the
  // rule of pcl_cols_kernel (pc_large_kernel.hip) and of the in-LDS kernels (pc_common.hpp)
  const int zq = BOX ? box_zero_period(n, N) : N + 1;
  for (int j = 0; j < np; ++j) {
    load_cols(zh_cur + (size_t)((MOF_K6S_ABLATE != 0 || MOF_SR_L2_ABLATE != 0) ? 0 : p0 + j) * zh_stride);
    bool box_zeros = false;
    if constexpr (BOX) box_zeros = __builtin_amdgcn_readfirstlane((int)(((flags[2 * (p0 + j)] & 1) == 0) || ((flags[2 * (p0 + j) + 1] & 1) == 0))) != 0;
    // normalised cross-power spectrum of bins (v, u), conjugated in place; the current spectra move into the registers
#pragma unroll
    for (int s = 0; s < CW; ++s) {
      const int u = u0 + s > H ? H : u0 + s;
      const bool u_edge = u == 0 || (!ODD && u == H);
#pragma unroll
      for (int m = 0; m < MV; ++m) {
        const int v = lane + 64 * m;
        const int vv = v < N ? v : N - 1;  // (lanes past the line repeat its last bin: the wave-uniform branch inside
        const cf a = lds_read(&z[s * P::LINE + vv]);  //  cross_power_ab wants every lane to take part)
        cf av = a, bv = ap[s][m];
        if constexpr (!SrNyqExact<P>::value) {
          // the four real-only CCS slots from the images' exact integer sums (doubled as the spectra are: Zh = 2 x the row transform): bins
          // (v, u) in {0, N/2}^2 -> sums[(v ? 2 : 0) + (u ? 1 : 0)] = sum (+-1)^y (+-1)^x p  (run = 1: one pair per wave walk)
          if (u_edge && (vv == 0 || (!ODD && vv == H))) {
            const int slot = (vv == H ? 2 : 0) + (u == H ? 1 : 0);
            av = {2.f * (float)sums_cur[(size_t)(p0 + j) * sums_stride + slot], 0.f};
            bv = {2.f * (float)sums_prev[(size_t)(p0 + j) * sums_stride + slot], 0.f};
          }
        }
        cf C = cross_power_ab(av, bv, u_edge && (vv == 0 || (!ODD && vv == H)));
        if constexpr (BOX) {
          if (box_zeros && (box_zero_line(u, zq) || box_zero_line(vv, zq))) C = {0.f, 0.f};
        }
        ap[s][m] = a;
        if (v < N) z[s * P::LINE + v] = {C.x, -C.y};
      }
    }
    wave_sync();
    wave_fft<N>(z, CW, lane, tw, StoreNatural<N>{});
    cf* D = reinterpret_cast<cf*>(Dt) + (size_t)(MOF_SR_L2_ABLATE ? 0 : p0 + j) * (H + 1) * N;
    if constexpr (ODD) {  // rows of N complex start on odd multiples of 8 bytes: element stores
#pragma unroll
      for (int s = 0; s < CW; ++s) {
        const int u = u0 + s;
#pragma unroll
        for (int m = 0; m < MV; ++m) {
          const int v = lane + 64 * m;
          if (v < N && u <= H) D[(size_t)u * N + v] = z[s * P::LINE + v];
        }
      }
    } else {
#pragma unroll
      for (int s = 0; s < CW; ++s) {
        const int u = u0 + s;
  #pragma unroll
        for (int m = 0; m < MQ; ++m) {
          const int q = lane + 64 * m;
          if (q < N / 2 && u <= H) {
            const cf a0 = z[s * P::LINE + 2 * q], a1 = z[s * P::LINE + 2 * q + 1];
            // (MOF_SR_L2_ABLATE = 2: Dt is not stored at all -- thousands of waves storing to the SAME aliased lines serialise in the L2 and
            //  made the aliased build's K6s 44 % slower; dropping the store bounds the write side from above instead)
            if (MOF_SR_L2_ABLATE == 2) asm volatile("" ::"v"(a0.x), "v"(a0.y), "v"(a1.x), "v"(a1.y));
            else stream_store(reinterpret_cast<float4*>(D + (size_t)u * N + 2 * q), make_float4(a0.x, a0.y, a1.x, a1.y));
          }
        }
      }
    }
    wave_sync();
  }
}

// ---- K6p: K6s with the radix-32 stage split over lane PAIRS (r04; N = 480 = 15 x 32) ----------------------------------------
// K6s is bound by its own work at two waves per SIMD, not by HBM (every wave reading pair 0's lines from L2 instead: 595 -> 582 us;
// VALU 33 %, LDS 38 %, 7.8 waves per CU -- profiles/r04_c5_sq_pmc.csv): four lines in LDS (16 KB) and 213 VGPRs per wave hold the
// occupancy down, and each transform makes three LDS round trips (staging, exchange, natural-order output). Here a wave owns TWO
// columns and a transform makes ONE round trip:
//   forward:  lane (line, n2) loads x[32 n1 + n2] straight from HBM (256 contiguous bytes per line and load), radix 15 over n1,
//             twiddle W480^{n2 k1}, -> LDS [line][k1][n2];  lane (line, k1, h) reads the 16 values n2 = 2 j + h, radix 16 over j,
//             the odd half times W32^q, halves swapped with the partner lane (DPP quad_perm xor 1):
//             X[k1 + 15 q] = E + O' in lane h = 0, X[k1 + 15 (q + 16)] = E - O' in lane h = 1      (16 bins = 32 VGPRs per lane)
//   cross-power elementwise against the previous frame's spectra in the same layout (32 VGPRs);
//   inverse:  halves swapped, s = c[q] + c[q + 16] (h = 0) | d = (c[q] - c[q + 16]) W32^q (h = 1), radix 16 over q -> t[m2 = 2 r + h],
//             -> LDS [line][k1][m2];  lane (line, m2) reads k1 = 0..14, twiddle W480^{k1 m2} (the SAME per-lane table), radix 15:
//             S[32 m1 + m2] -> Dt (256 contiguous bytes per line and store).
// 8 KB of LDS and ~half the registers per wave: four waves per SIMD.
#ifndef MOF_K6P_WPE
#define MOF_K6P_WPE 3
#endif
template <int N>
__global__ void __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MOF_K6P_WPE, MOF_K6P_WPE))) sr_cols_split_kernel(const float* __restrict__ zh_prev, const float* __restrict__ zh_cur,
                                                           size_t zh_stride, const float* __restrict__ twiddles,
                                                           float* __restrict__ Dt, int n_pairs, int run) {
  static_assert(N == 480, "15 x 32 only");
  constexpr int H = N / 2, R1 = 15, Y2 = 34, LSZ = R1 * Y2;  // Y2 = 34: the pair lanes' 16-byte-apart reads of 15 rows hit distinct banks
  typedef float v2f_t __attribute__((ext_vector_type(2)));
  __shared__ cf x[2 * LSZ];
  const int lane = threadIdx.x, u0 = blockIdx.x * 2, p0 = blockIdx.y * run;
  const int np = n_pairs - p0 < run ? n_pairs - p0 : run;
  const int l1 = lane >> 5, n2 = lane & 31;                                   // stage 1 / stage B: (line, n2 | m2)
  const int k1 = ((lane & 31) >> 1) < R1 ? ((lane & 31) >> 1) : R1 - 1, h = lane & 1;  // stage 2 / stage A: (line l1, k1, half); the two
  const bool on2 = ((lane & 31) >> 1) < R1;                                   // spare lanes of a line repeat k1 = 14 (finite numbers)
  const int u1 = u0 + l1 > H ? H : u0 + l1;                                   // this lane's column (the tail group repeats u = H)
  // (the 14 inter-stage twiddles W480^{n2 k} of a lane are re-read from the 3.8 KB table -- L1 hits -- at each of their three uses per
  //  pair instead of living in 28 VGPRs: what separates this kernel from four waves per SIMD)
  const float* twl = twiddles + 2 * n2;
  auto twk = [&](int k) -> cf {
    const float2 t = *reinterpret_cast<const float2*>(twl + 2 * (size_t)(n2 * (k - 1)));  // W_N^{n2 k} sits at index n2 k
    return {t.x, t.y};
  };
  // W32^q = (cos(pi q / 16), -sin(pi q / 16))
  const cf w32[16] = {{1.f, 0.f}, {0.98078528040323044913f, -0.19509032201612826785f}, {0.92387953251128675613f, -0.38268343236508977173f},
                      {0.83146961230254523708f, -0.55557023301960222474f}, {0.70710678118654752440f, -0.70710678118654752440f},
                      {0.55557023301960222474f, -0.83146961230254523708f}, {0.38268343236508977173f, -0.92387953251128675613f},
                      {0.19509032201612826785f, -0.98078528040323044913f}, {0.f, -1.f}, {-0.19509032201612826785f, -0.98078528040323044913f},
                      {-0.38268343236508977173f, -0.92387953251128675613f}, {-0.55557023301960222474f, -0.83146961230254523708f},
                      {-0.70710678118654752440f, -0.70710678118654752440f}, {-0.83146961230254523708f, -0.55557023301960222474f},
                      {-0.92387953251128675613f, -0.38268343236508977173f}, {-0.98078528040323044913f, -0.19509032201612826785f}};
  auto swap1 = [](cf v) -> cf {  // the partner lane's value (lane ^ 1)
    return {__builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.x), 0xB1, 0xf, 0xf, true)),
            __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v.y), 0xB1, 0xf, 0xf, true))};
  };
  // column spectra of one frame's two lines: out[q] = X[k1 + 15 (q + 16 h)]. ONE instance of this code serves every frame (the loop
  // below starts at the run's previous frame): a run walked in one launch and the same frames fed one at a time round identically.
  auto forward = [&](const float* __restrict__ frame, cf* out) {
    const cf* line = reinterpret_cast<const cf*>(frame) + (size_t)u1 * sr_zh_pitch<N>() + n2;
    cf a[R1];
#pragma unroll
    for (int n1 = 0; n1 < R1; ++n1) {
      const v2f_t t = __builtin_nontemporal_load(reinterpret_cast<const v2f_t*>(line + 32 * n1));
      a[n1] = {t.x, t.y};
    }
    butterfly15(a);
    cf* row = x + l1 * LSZ + n2;
    row[0] = a[0];
#pragma unroll
    for (int k = 1; k < R1; ++k) row[k * Y2] = cmul(a[k], twk(k));
    wave_sync();
    const cf* src = x + l1 * LSZ + k1 * Y2 + h;
    cf e[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) e[j] = lds_read(&src[2 * j]);
    butterfly<16>(e);
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const cf o = h ? cmul(e[q], w32[q]) : e[q];  // the odd half carries W32^q
      const cf r = swap1(o);
      out[q] = h ? csub(r, o) : cadd(o, r);        // h = 0: E + O';  h = 1: E - O'
    }
    wave_sync();  // (the exchange buffer is reused)
  };

  cf ap[16], ac[16];
#pragma unroll 1
  for (int j = -1; j < np; ++j) {
    forward(j < 0 ? zh_prev + (size_t)p0 * zh_stride : zh_cur + (size_t)(p0 + j) * zh_stride, ac);
    if (j < 0) {
#pragma unroll
      for (int q = 0; q < 16; ++q) ap[q] = ac[q];
      continue;
    }
    // normalised cross-power spectrum of bins (v = k1 + 15 (q + 16 h), u), conjugated; the current spectra become the previous ones
    const bool u_edge = u1 == 0 || u1 == H;
    cf c[16];
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const int v = k1 + 15 * (q + 16 * h);
      const cf C = cross_power_ab(ac[q], ap[q], u_edge && (v == 0 || v == H));
      ap[q] = ac[q];
      c[q] = {C.x, -C.y};
    }
    // inverse (a forward transform of conj C): stage A on the lane pair, exchange, stage B
#pragma unroll
    for (int q = 0; q < 16; ++q) {
      const cf r = swap1(c[q]);
      c[q] = h ? cmul(csub(r, c[q]), w32[q]) : cadd(c[q], r);  // h = 1 holds c[q + 16], its partner c[q]
    }
    butterfly<16>(c);  // c[r] = t[m2 = 2 r + h]
    if (on2) {
      cf* dst = x + l1 * LSZ + k1 * Y2 + h;
#pragma unroll
      for (int r = 0; r < 16; ++r) dst[2 * r] = c[r];
    }
    wave_sync();
    const cf* col = x + l1 * LSZ + n2;
    cf b[R1];
    b[0] = lds_read(&col[0]);
#pragma unroll
    for (int k = 1; k < R1; ++k) b[k] = cmul(lds_read(&col[k * Y2]), twk(k));
    butterfly15(b);
    if (u0 + l1 <= H) {
      cf* D = reinterpret_cast<cf*>(Dt) + ((size_t)(p0 + j) * (H + 1) + (u0 + l1)) * N + n2;
#pragma unroll
      for (int m1 = 0; m1 < R1; ++m1) {
        const v2f_t t = {b[m1].x, b[m1].y};
        __builtin_nontemporal_store(t, reinterpret_cast<v2f_t*>(D + 32 * m1));
      }
    }
    wave_sync();
  }
}

// what processImage returns for the very first frame of a sequence (scaleRotationEstimator.cpp:74): (1, 0), no pt
__global__ void sr_identity_kernel(double* __restrict__ out) {
  if (threadIdx.x < 4) out[threadIdx.x] = threadIdx.x == 0 ? 1.0 : 0.0;
}

template <int N>
hipError_t launch_rows_real_n(const uint8_t* lp, size_t lp_stride, const float* tw, float* zh, size_t zh_stride, int n_frames,
                              hipStream_t stream) {
  using R = RowsReal<N>;
  static_assert(N % R::ROWS == 0, "rows divide evenly over the workgroups");
  constexpr size_t lds = sizeof(cf) * R::LINES * SrPlan<N>::LINE;
  if (lds > 48 * 1024) {
    const hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sr_rows_real_kernel<N>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(sr_rows_real_kernel<N>, dim3(N / R::ROWS, (unsigned)n_frames), dim3(R::T), lds, stream, lp, lp_stride, tw, zh,
                     zh_stride);
  return hipGetLastError();
}

template <int N>
hipError_t launch_rows_real_src_n(const PclSrc& src, const float* tw, float* zh, size_t zh_stride, int* flags, int n_images, int channels,
                                  int n, int* sums, hipStream_t stream) {
  using R = RowsReal<N>;
  constexpr size_t lds = sizeof(cf) * R::LINES * SrPlan<N>::LINE;
  const bool pad = n < N;
  if (lds > 48 * 1024) {
    const void* f = channels == 3 ? (pad ? reinterpret_cast<const void*>(&sr_rows_real_src_kernel<N, 3, true>) : reinterpret_cast<const void*>(&sr_rows_real_src_kernel<N, 3, false>))
                                  : (pad ? reinterpret_cast<const void*>(&sr_rows_real_src_kernel<N, 1, true>) : reinterpret_cast<const void*>(&sr_rows_real_src_kernel<N, 1, false>));
    const hipError_t e = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    if (e != hipSuccess) return e;
  }
  for (int f0 = 0; f0 < n_images; f0 += 65534) {  // (the image index rides gridDim.y; even chunks keep cur | prev pairs together)
    const int nf = n_images - f0 < 65534 ? n_images - f0 : 65534;
    PclSrc s = src;
    const int patches = src.grid_x * src.grid_y;
    const int per_unit = src.paired == 2 ? patches : 2 * patches;  // images per frame (a video) / per frame pair
    if (f0 % per_unit != 0) return hipErrorInvalidValue;  // (the caller splits at whole frames / frame pairs: mof_capi.hip)
    s.base[0] += (size_t)(f0 / per_unit) * src.stride[0];
    if (src.paired != 2) s.base[1] += (size_t)(f0 / per_unit) * src.stride[1];
    const dim3 g(R::GROUPS, (unsigned)nf), b(R::T);
    float* zo = zh + (size_t)f0 * zh_stride;
    int* fl = flags ? flags + f0 : nullptr;
    int* su = sums ? sums + (size_t)(src.sums_stride ? src.sums_stride : 4) * f0 : nullptr;
    if (channels == 3) {
      if (pad) hipLaunchKernelGGL((sr_rows_real_src_kernel<N, 3, true>), g, b, lds, stream, s, tw, zo, zh_stride, fl, n, su);
      else hipLaunchKernelGGL((sr_rows_real_src_kernel<N, 3, false>), g, b, lds, stream, s, tw, zo, zh_stride, fl, n, su);
    } else {
      if (pad) hipLaunchKernelGGL((sr_rows_real_src_kernel<N, 1, true>), g, b, lds, stream, s, tw, zo, zh_stride, fl, n, su);
      else hipLaunchKernelGGL((sr_rows_real_src_kernel<N, 1, false>), g, b, lds, stream, s, tw, zo, zh_stride, fl, n, su);
    }
  }
  return hipGetLastError();
}

template <int N>
hipError_t launch_cols_seq_n(const float* zh_prev, const float* zh_cur, size_t zh_stride, const float* tw, float* Dt, int n_pairs,
                             int run, const int* flags, int n, const int* sums_prev, const int* sums_cur, int sums_stride, hipStream_t stream) {
  if (!SrNyqExact<SrPlan<N>>::value && (!sums_prev || !sums_cur)) return hipErrorInvalidValue;  // (this plan's real-only slots come from the exact sums; a run
  // of pairs in time finds pair j's quadruples at (p0 + j) * sums_stride of either pointer -- sums that live inside the Zh slots walk with them)
  constexpr int H = N / 2;
  const unsigned groups = (H + 1 + seq_cw<N>() - 1) / seq_cw<N>(), runs = (unsigned)((n_pairs + run - 1) / run);
  if constexpr (N == 480) {
    // MOF_SR_COLS_SPLIT=1: K6p, two columns per wave and the radix-32 stage on lane pairs
    static const bool split = [] { const char* v = getenv("MOF_SR_COLS_SPLIT"); return v && atoi(v) != 0; }();
    if (split && !(flags && n < N)) {
      hipLaunchKernelGGL(sr_cols_split_kernel<N>, dim3((H + 2) / 2, runs), dim3(64), 0, stream, zh_prev, zh_cur, zh_stride, tw, Dt, n_pairs, run);
      return hipGetLastError();
    }
  }
  if (flags && n < N) hipLaunchKernelGGL((sr_cols_seq_kernel<N, true>), dim3(groups, runs), dim3(64), 0, stream, zh_prev, zh_cur, zh_stride, tw, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride);
  else hipLaunchKernelGGL((sr_cols_seq_kernel<N, false>), dim3(groups, runs), dim3(64), 0, stream, zh_prev, zh_cur, zh_stride, tw, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride);
  return hipGetLastError();
}

}  // namespace

size_t sr_zh_floats(int res) { return (size_t)(res / 2 + 1) * (((res + 7) & ~7) + 8) * 2; }  // (pitch: sr_zh_pitch)

hipError_t launch_sr_identity(double* out, hipStream_t stream) {
  hipLaunchKernelGGL(sr_identity_kernel, dim3(1), dim3(64), 0, stream, out);
  return hipGetLastError();
}

hipError_t launch_sr_rows_real(const uint8_t* lp, size_t lp_stride, const float* twiddles, float* zh, size_t zh_stride, int res,
                               int n_frames, hipStream_t stream) {
  if (n_frames <= 0) return hipSuccess;
  switch (res) {
    case 240: return launch_rows_real_n<240>(lp, lp_stride, twiddles, zh, zh_stride, n_frames, stream);
    case 256: return launch_rows_real_n<256>(lp, lp_stride, twiddles, zh, zh_stride, n_frames, stream);
    case 480: return launch_rows_real_n<480>(lp, lp_stride, twiddles, zh, zh_stride, n_frames, stream);
    default: {
      // (r06) the other tuned sizes: the frame form of the row kernel that the FFT engine's large patches use -- tightly packed log-polar
      // images are ONE patch per "frame" of a video whose patch is the whole image
      PclSrc src{};
      src.base[0] = lp;
      src.stride[0] = lp_stride;
      src.pitch = (size_t)res;
      src.paired = 2;
      src.grid_x = src.grid_y = 1;
      src.stride_x = src.stride_y = res;
      return launch_sr_rows_real_src(src, twiddles, zh, zh_stride, nullptr, res, n_frames, 1, res, stream, nullptr);
    }
  }
}

hipError_t launch_sr_rows_real_src(const PclSrc& src, const float* twiddles, float* zh, size_t zh_stride, int* flags, int res, int n_images,
                                   int channels, int n, hipStream_t stream, int* sums) {
  if (n_images <= 0) return hipSuccess;
  if ((src.paired != 1 && src.paired != 2) || (channels != 1 && channels != 3) || n < 2 || n > res) return hipErrorInvalidValue;
  switch (res) {
    case 200: return launch_rows_real_src_n<200>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 216: return launch_rows_real_src_n<216>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 240: return launch_rows_real_src_n<240>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 256: return launch_rows_real_src_n<256>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 250: return launch_rows_real_src_n<250>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 400: return launch_rows_real_src_n<400>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 432: return launch_rows_real_src_n<432>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 270: return launch_rows_real_src_n<270>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 300: return launch_rows_real_src_n<300>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 450: return launch_rows_real_src_n<450>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 288: return launch_rows_real_src_n<288>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 320: return launch_rows_real_src_n<320>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 360: return launch_rows_real_src_n<360>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 384: return launch_rows_real_src_n<384>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 480: return launch_rows_real_src_n<480>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 512: return launch_rows_real_src_n<512>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 225: return launch_rows_real_src_n<225>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 243: return launch_rows_real_src_n<243>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 375: return launch_rows_real_src_n<375>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 405: return launch_rows_real_src_n<405>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 625: return launch_rows_real_src_n<625>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 675: return launch_rows_real_src_n<675>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 729: return launch_rows_real_src_n<729>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 128: return launch_rows_real_src_n<128>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 96: return launch_rows_real_src_n<96>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 100: return launch_rows_real_src_n<100>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 108: return launch_rows_real_src_n<108>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 120: return launch_rows_real_src_n<120>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 150: return launch_rows_real_src_n<150>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 162: return launch_rows_real_src_n<162>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 144: return launch_rows_real_src_n<144>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 160: return launch_rows_real_src_n<160>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 180: return launch_rows_real_src_n<180>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 192: return launch_rows_real_src_n<192>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 324: return launch_rows_real_src_n<324>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 486: return launch_rows_real_src_n<486>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 500: return launch_rows_real_src_n<500>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 540: return launch_rows_real_src_n<540>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 576: return launch_rows_real_src_n<576>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 600: return launch_rows_real_src_n<600>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 640: return launch_rows_real_src_n<640>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 648: return launch_rows_real_src_n<648>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 720: return launch_rows_real_src_n<720>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 750: return launch_rows_real_src_n<750>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 768: return launch_rows_real_src_n<768>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 800: return launch_rows_real_src_n<800>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 810: return launch_rows_real_src_n<810>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 864: return launch_rows_real_src_n<864>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 900: return launch_rows_real_src_n<900>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    case 960: return launch_rows_real_src_n<960>(src, twiddles, zh, zh_stride, flags, n_images, channels, n, sums, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_sr_cols_seq(const float* zh_prev, const float* zh_cur, size_t zh_stride, const float* twiddles, float* Dt, int res,
                              int n_pairs, int run, hipStream_t stream, const int* flags, int n, const int* sums_prev, const int* sums_cur, int sums_stride) {
  if (n_pairs <= 0) return hipSuccess;
  if (run < 1) run = 1;
  if (flags && run != 1) return hipErrorInvalidValue;  // (the box-zero flags are per independent pair: image 2 p = cur, 2 p + 1 = prev)
  if (n <= 0) n = res;
  // a run longer than one pair walks cur(p) as prev(p + 1): only valid for a contiguous sequence
  if (run > 1 && zh_cur != zh_prev + zh_stride) return hipErrorInvalidValue;
  switch (res) {
    case 200: return launch_cols_seq_n<200>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 216: return launch_cols_seq_n<216>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 240: return launch_cols_seq_n<240>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 256: return launch_cols_seq_n<256>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 250: return launch_cols_seq_n<250>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 400: return launch_cols_seq_n<400>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 432: return launch_cols_seq_n<432>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 270: return launch_cols_seq_n<270>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 300: return launch_cols_seq_n<300>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 450: return launch_cols_seq_n<450>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 288: return launch_cols_seq_n<288>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 320: return launch_cols_seq_n<320>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 360: return launch_cols_seq_n<360>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 384: return launch_cols_seq_n<384>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 480: return launch_cols_seq_n<480>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 512: return launch_cols_seq_n<512>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 225: return launch_cols_seq_n<225>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 243: return launch_cols_seq_n<243>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 375: return launch_cols_seq_n<375>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 405: return launch_cols_seq_n<405>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 625: return launch_cols_seq_n<625>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 675: return launch_cols_seq_n<675>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 729: return launch_cols_seq_n<729>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 128: return launch_cols_seq_n<128>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 96: return launch_cols_seq_n<96>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 100: return launch_cols_seq_n<100>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 108: return launch_cols_seq_n<108>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 120: return launch_cols_seq_n<120>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 150: return launch_cols_seq_n<150>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 162: return launch_cols_seq_n<162>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 144: return launch_cols_seq_n<144>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 160: return launch_cols_seq_n<160>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 180: return launch_cols_seq_n<180>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 192: return launch_cols_seq_n<192>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 324: return launch_cols_seq_n<324>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 486: return launch_cols_seq_n<486>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 500: return launch_cols_seq_n<500>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 540: return launch_cols_seq_n<540>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 576: return launch_cols_seq_n<576>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 600: return launch_cols_seq_n<600>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 640: return launch_cols_seq_n<640>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 648: return launch_cols_seq_n<648>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 720: return launch_cols_seq_n<720>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 750: return launch_cols_seq_n<750>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 768: return launch_cols_seq_n<768>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 800: return launch_cols_seq_n<800>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 810: return launch_cols_seq_n<810>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 864: return launch_cols_seq_n<864>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 900: return launch_cols_seq_n<900>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    case 960: return launch_cols_seq_n<960>(zh_prev, zh_cur, zh_stride, twiddles, Dt, n_pairs, run, flags, n, sums_prev, sums_cur, sums_stride, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mof
