// sr_kernel.hip -- K4..K8: the scale/rotation estimator of BASELINE config c5 on gfx950.
//
// Replaces scaleRotationEstimator::processImage (/root/reference/src/scaleRotationEstimator.cpp:34-148):
//   cv::logPolar(imCurr, tempIm, center, M, INTER_CUBIC | INTER_LANCZOS4)     :45, :112     -> K4 sr_logpolar_kernel
//   cv::phaseCorrelate(tempIm_F32, prevIm_F32) on the whole res x res image   :117          -> K5..K8
// The res x res complex tile (480^2 x 8 B = 1.8 MB) fits neither the LDS nor the registers of a CU, so unlike K1 the
// whole-frame correlation has to cross CUs twice (after the row pass and after the column pass); everything between
// those two hand-overs stays on chip:
//   K5 sr_rows_fwd   : u8 log-polar rows of cur/prev packed as cur + i*prev, row FFTs in LDS, written TRANSPOSED
//                      (Zt[u][v], 64-byte segments) so that the column pass reads whole lines                  -> Zt
//   K6 sr_cols       : column FFTs of a column group AND its mirror, untangle + normalised cross-power
//                      spectrum (same rules as K1, incl. the real-only slots), inverse column FFTs of the
//                      half spectrum, written line by line                                                    -> Dt (N/2+1 lines of N)
//   K7 sr_rows_inv   : Hermitian rows, two per complex transform; the real surface is NEVER written: the arg-max is
//                      taken from the registers of the last stage                                             -> candidates
//   K8 sr_final      : first-maximum reduction; the 5x5 window around the peak is re-evaluated from Dt (25 dot products
//                      of 240 terms in fp64), centroid, pt -> (scale, rot) with the reference's gate
// A 1-D transform is two in-register stages N = R1 x R2 (480 = 15 x 32, 240 = 15 x 16, 256 = 16 x 16) with ONE LDS
// round trip between them, done by a single wave on the lines it owns (no workgroup barrier inside a transform).

#include <hip/hip_runtime.h>
#include <cstdlib>
#include <stdint.h>
#include <stdlib.h>

#include "mof_kernels.h"
#include "pc_common.hpp"
#include "sr_common.hpp"

namespace mof {

// Eight (four) taps of one footprint row: the u8 pixels are spread to u16 pairs with v_perm_b32 and meet the int16
// weight pairs -- stored exactly like that in the table -- in v_dot2c_i32_i16: one instruction per tap instead of the
// three (two bit-field extracts + multiply-add) of the scalar form; exact integer arithmetic either way. (Measured r02:
// no change in kernel time -- the remap is bound by the L1's line look-ups, one per lane and tap row, not by the VALU.)
template <int K>
__device__ __forceinline__ int dot_row(const uint32_t* px, const uint32_t* wq, int sum) {
  typedef short short2_t __attribute__((ext_vector_type(2)));
#pragma unroll
  for (int d = 0; d < K / 4; ++d) {
    const uint32_t p01 = __builtin_amdgcn_perm(0u, px[d], 0x0c010c00u);  // (p0, p1) as two u16
    const uint32_t p23 = __builtin_amdgcn_perm(0u, px[d], 0x0c030c02u);  // (p2, p3)
    sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, p01), __builtin_bit_cast(short2_t, wq[2 * d]), sum, false);
    sum = __builtin_amdgcn_sdot2(__builtin_bit_cast(short2_t, p23), __builtin_bit_cast(short2_t, wq[2 * d + 1]), sum, false);
  }
  return sum;
}

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
  if (!m.valid) {
    if (a.zero_invalid) a.dst[(size_t)img * a.dst_stride + pix] = 0;
    return;
  }
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
      sum = dot_row<K>(px, wq, sum);
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
    if (!m.valid) {
      // BORDER_TRANSPARENT keeps the destination; a batch whose destination is the estimator's zero-initialised
      // tempIm (scaleRotationEstimator.cpp:27) gets that zero written here instead of a separate memset pass
      if (a.zero_invalid) a.dst[(size_t)img * a.dst_stride + pix] = 0;
      continue;
    }
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
        sum = dot_row<K>(px, wq, sum);
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

// ---- K4, batched fast path: tile-stationary, source boxes staged in LDS ------------------------------------
// The map and the weights depend on the destination pixel only, never on the image. A WAVE owns one 8 (phi) x 8 (rho)
// destination tile, reads its map entries and its K x K weights ONCE into registers and then walks `img_per_wave`
// images. Gathering the taps straight from global memory costs one L1 line look-up per (lane, tap row) -- at the outer
// radii neighbouring destination pixels are 5-7 source pixels apart, so every look-up is a different line, and the L1's
// one line per cycle sets the time (measured r01/r02: 125 us per 64 Lanczos4 images whether the arithmetic is 192 or
// 64 instructions per pixel, twice the cubic kernel's time). Here the wave instead copies the bounding box of its
// tile's footprints (SrTileBox, precomputed on the host from the map, reflected border taps included; a few KB) into LDS with aligned dword loads
// -- a few dozen line look-ups per image -- and gathers from there. The box of image i+1 is fetched into registers
// BEFORE the taps of image i are gathered (U images form a group whose boxes travel together), so the memory latency
// hides behind the arithmetic.
// With pitch % 4 == 0 and src_stride % 4 == 0 a row's misalignment is the same for every row and image and the box is fetched with aligned
// dwords; other layouts fetch every box row from its exact first byte (unaligned dword loads, r06).
// LDS dwords of a wave's box: whole 64-lane slot rows of the ring class (4, 8 or 16 dwords per lane) that holds the
// largest box of the map
__host__ __device__ inline int lp_box_capacity(int box_dwords_max) {
  const int t = (box_dwords_max + 63) / 64;
  return 64 * (t <= 4 ? 4 : (t <= 8 ? 8 : (t <= 12 ? 12 : 16)));
}

#ifdef MOF_LP_WPE
#define MOF_LP_ATTR __attribute__((amdgpu_waves_per_eu(MOF_LP_WPE, MOF_LP_WPE)))
#else
#define MOF_LP_ATTR
#endif
// SUPER: the four waves of a workgroup own the four 8 x 8 tiles of one 16 x 16 SUPER-TILE and share ONE staged box (the
// bounding box of all their footprints, a.sboxes): the texture addresser was what bound the per-wave form (TA_BUSY 84-89 %:
// every box row is a cache line of its own), and one shared box has 2.2-2.5x fewer rows than four separate ones. The
// lanes of all four waves stage it together (slot = thread + 256 t) and meet at a raw s_barrier (a __syncthreads would
// drain the ring's loads in flight).
// SUPER runs four workgroups per CU (r03): 16 waves instead of 12 shorten nothing in a wave's own chain (≈60 VALU instructions
// at one issue per ≈4 clocks + 24 LDS reads and their round trip per image) but put a third more of them side by side:
// c5seq +3.4 % same-box even with 7 spilled dwords; with a 12-deep staging ring where the map's boxes allow it (NR = 12:
// 8 VGPRs less) the Lanczos4 form fits 128 VGPRs.
#ifndef MOF_LP_SUPER_WPE
#define MOF_LP_SUPER_WPE 4
#endif
#ifndef MOF_LP_SUPER_WPE4
#define MOF_LP_SUPER_WPE4 4  // the cubic form
#endif
template <int K, int NR, bool SUPER>
__global__ void __launch_bounds__(256, (SUPER ? (K == 4 ? MOF_LP_SUPER_WPE4 : MOF_LP_SUPER_WPE) : 1)) MOF_LP_ATTR sr_logpolar_staged_kernel(SrLpArgs a, int n_images, int img_per_wave, int xcd_groups) {
  extern __shared__ __attribute__((aligned(16))) uint32_t lp_lds[];
  const int res = a.res, tiles = (res + 7) / 8, n_tiles = SUPER ? (tiles / 2) * (tiles / 2) : tiles * tiles;
  const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
  constexpr int NT = SUPER ? 256 : 64;                 // threads that stage one box
  const int st_tid = SUPER ? (int)threadIdx.x : lane;  // this thread's staging index
  int tile, img0;
  if constexpr (SUPER) {
    // one workgroup per (super-tile, image group); same XCD-aware order
    if (xcd_groups > 0) {
      const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
      const int g = xcd + 8 * (j / n_tiles);
      tile = j % n_tiles;
      img0 = g * img_per_wave;
      if (g >= xcd_groups) return;
    } else {
      tile = blockIdx.x % n_tiles;
      img0 = (blockIdx.x / n_tiles) * img_per_wave;
    }
  } else if (xcd_groups > 0) {
    // XCD-aware order (speed only): workgroups b, b + 8, b + 16, .. share an XCD and its L2, so image group g goes to
    // the workgroups with b % 8 == g % 8 -- every image is then fetched into ONE L2 instead of up to eight
    // (r02: 3.3x the image bytes in FETCH_SIZE with the plain order)
    const int nq = (n_tiles + 3) / 4, xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
    const int g = xcd + 8 * (j / nq);
    tile = 4 * (j % nq) + wave;
    img0 = g * img_per_wave;
    if (g >= xcd_groups || tile >= n_tiles) return;
  } else {
    const int wid = blockIdx.x * 4 + wave;
    tile = wid % n_tiles;
    img0 = (wid / n_tiles) * img_per_wave;
  }
  if (img0 >= n_images) return;
  const int img1 = img0 + img_per_wave < n_images ? img0 + img_per_wave : n_images;
  const int rho = SUPER ? (tile % (tiles / 2)) * 16 + (wave & 1) * 8 + (lane & 7) : (tile % tiles) * 8 + (lane & 7);
  const int phi = SUPER ? (tile / (tiles / 2)) * 16 + (wave >> 1) * 8 + (lane >> 3) : (tile / tiles) * 8 + (lane >> 3);
  const bool inside = rho < res && phi < res;
  const int pix = phi * res + rho;
  SrMapEntry m{0, 0, 0, 0};
  if (inside) m = a.map[pix];
  const bool valid = inside && m.valid;
  constexpr int HALF = K / 2 - 1;
  const int sx = m.ax - HALF, sy = m.ay - HALF;
  uint8_t* dst = a.dst + (size_t)img0 * a.dst_stride + pix;
  const bool wave_valid = __ballot(valid) != 0ull;
  // SUPER: the 16 x 16 output tile leaves through LDS -- every lane drops its byte, sixteen lanes of wave 0 pick up a row
  // each and store 16 bytes: one store instruction and 16 cache lines per image instead of four and 32 (the cubic
  // kernel's texture addresser was 84 % busy, half of it stores). BORDER_TRANSPARENT tiles that must keep destination
  // pixels (no zero_invalid, some lane invalid) store per lane.
  bool row_store = false;
  if constexpr (SUPER) row_store = a.zero_invalid || __syncthreads_and((int)valid) != 0;
  if (SUPER ? __syncthreads_or((int)valid) == 0 : !wave_valid) {  // the whole (super-)tile maps outside the source
    if (a.zero_invalid && inside)
      for (int img = img0; img < img1; ++img, dst += a.dst_stride) *dst = 0;
    return;
  }
  // This pixel's K x K int16 weights as two planes of SIGNED bytes, w = 256 wh + wl, four taps per dword: the staged
  // pixels are kept as signed bytes too (p - 128, one xor when a box is committed), so a tap row is K/4 pairs of
  // v_dot4c_i32_i8 on the raw window bytes -- the u8 -> u16 spreading of the 16-bit form (two v_perm per dword) is gone:
  //   sum w p = 256 sum wh (p - 128) + sum wl (p - 128) + 128 sum w.
  // wh would be 128 for a weight >= 32640 (the one tap of a footprint that sits on a source pixel: 32767); it is kept
  // at 127 and the missing 256 (p - 128) of that tap is added after the rows (`rem_*`, at most one tap per pixel).
  // The planes are built on the host (sr_weight_planes): splitting 64 int16 in the kernel cost 70 VGPRs of set-up.
  uint32_t wh[K * K / 4], wl[K * K / 4];
  int wconst = 0, rem_k = -1;
  {
    uint32_t t[K * K / 2 + 2];
#pragma unroll
    for (int i = 0; i < K * K / 2 + 2; ++i) t[i] = 0;
    if (valid) __builtin_memcpy(t, __builtin_assume_aligned(a.wplanes + (size_t)m.widx * (K * K / 2 + 2), 8), sizeof(t));
#pragma unroll
    for (int i = 0; i < K * K / 4; ++i) wh[i] = t[i], wl[i] = t[K * K / 4 + i];
    wconst = (int)t[K * K / 2];
    rem_k = valid ? (int)t[K * K / 2 + 1] : -1;
  }
  // the box: bw x bh source pixels at (bx, by); LDS rows of lpd dwords; every row starts `mis` bytes into its first dword
  const SrTileBox box = SUPER ? a.sboxes[tile] : a.boxes[tile];
  const int bx = box.x0, by = box.y0, bw = box.w, bh = box.h;
  const int lpd = (bw + 3 + 3) / 4 + 1, cnt = bh * lpd;
  const uint8_t* src = a.src + (size_t)img0 * a.src_stride;
  // (r06: a layout whose rows or frames are NOT a multiple of four bytes apart -- resolution 250, 270, 350, 450 packed tightly -- has a different
  //  misalignment in every row: there the box rows are fetched from their exact first byte with unaligned dword loads, mis = 0)
  const bool dword_layout = ((a.pitch | a.src_stride) & 3) == 0;
  const uint32_t mis = dword_layout ? (uint32_t)(uintptr_t)(src + (size_t)by * a.pitch + bx) & 3u : 0u;
  const uint8_t* b0 = src + (size_t)by * a.pitch + bx - mis;  // dword-aligned; the same offset in every image
  // Box dword i = lane + 64 t comes from byte offset goff[t]. Slots past the box, or past the last needed byte of a
  // row (they may lie outside the buffer), re-read the box's first dword and park it in a dump slot behind the box, so
  // that the T = ceil(cnt / 64) loads and LDS writes of a box are unconditional for the whole wave.
  const int T = __builtin_amdgcn_readfirstlane((cnt + NT - 1) / NT);
  uint32_t goff[NR];
  {
    // (row, dword) of slot st_tid + NT t, advanced by NT slots per step: two integer divisions per lane instead of 2 NR
    int r = st_tid / lpd, j = st_tid % lpd;
    const int dr = NT / lpd, dj = NT % lpd;
#pragma unroll
    for (int t = 0; t < NR; ++t) {
      const bool ok = st_tid + NT * t < cnt && 4 * j < (int)mis + bw;
      goff[t] = ok ? (uint32_t)r * (uint32_t)a.pitch + 4u * (uint32_t)j : 0u;
      j += dj;
      r += dr;
      if (j >= lpd) {
        j -= lpd;
        ++r;
      }
    }
  }
  const int box_dwords = SUPER ? 4 * lp_box_capacity((a.sbox_dwords_max + 3) / 4) : lp_box_capacity(a.box_dwords_max);
  uint32_t* L = SUPER ? lp_lds : lp_lds + (size_t)wave * 2 * box_dwords;  // two boxes: the image being gathered and the next one
  // Footprints that cross the border take BORDER_REFLECT_101 taps. They go through the SAME code as interior ones:
  // tap row k1 is source row reflect(sy + k1); the K taps of a row are read as the K-byte window starting at
  // wx = clamp(sx, 0, res - K) -- it holds every reflected column -- and put in tap order by one byte permute
  // whose selector depends on the pixel only (identity for interior pixels).
  const int wx = sx < 0 ? 0 : (sx > res - K ? res - K : sx);
  uint32_t sel[K / 4];
#pragma unroll
  for (int d = 0; d < K / 4; ++d) {
    sel[d] = 0;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
      int xx = sx + 4 * d + q;
      xx = xx < 0 ? -xx : xx;
      xx = xx >= res ? 2 * res - 2 - xx : xx;
      sel[d] |= (uint32_t)(xx - wx) << (8 * q);
    }
  }
  int rowoff[K];
#pragma unroll
  for (int k1 = 0; k1 < K; ++k1) {
    int yy = sy + k1;
    yy = yy < 0 ? -yy : yy;
    yy = yy >= res ? 2 * res - 2 - yy : yy;
    rowoff[k1] = valid ? (yy - by) * lpd : 0;
  }
  const uint32_t o = valid ? (uint32_t)(wx - bx) + mis : 0u;
  const int lcol = (int)(o >> 2);
  const uint32_t sh = o & 3u;
  // the tap that carries `rem`: LDS dword (relative to the box row start of its tap row) and byte within it
  int rem_off = 0;
  uint32_t rem_sh = 0;
  if (rem_k >= 0) {
    const int k1 = rem_k / K, k2 = rem_k % K;
#pragma unroll
    for (int i = 0; i < K; ++i) rem_off = i == k1 ? rowoff[i] : rem_off;
    const uint32_t byte = sh + ((sel[k2 >> 2] >> (8 * (k2 & 3))) & 0xffu);  // position of tap k2 in the row's window
    rem_off += lcol + (int)(byte >> 2);
    rem_sh = 8u * (byte & 3u);
  }
  const bool wave_rem = __ballot(valid && rem_k >= 0) != 0ull;
  // columns reflected at the border need the byte permute; interior waves (nearly all) skip it
  const bool wave_border = __ballot(valid && (sx < 0 || sx + K > res)) != 0ull;

  auto gather = [&](auto border_c, const uint32_t* Lu) -> int {
    constexpr bool BORDER = decltype(border_c)::value;
    int s_hi = 0, s_lo = 0;
    // (issuing every tap row's LDS reads before the first dot product instead of the compiler's row-by-row order:
    // same-box A/B, no change)
#pragma unroll
    for (int k1 = 0; k1 < K; ++k1) {
      const uint32_t* p = Lu + rowoff[k1] + lcol;
      uint32_t px[K / 4];
      if constexpr (K == 8) {
        const uint32_t d0 = p[0], d1 = p[1], d2 = p[2];
        px[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        px[1] = __builtin_amdgcn_alignbyte(d2, d1, sh);
        if constexpr (BORDER) {
          const uint32_t a0 = px[0], a1 = px[1];
          px[0] = __builtin_amdgcn_perm(a1, a0, sel[0]);
          px[1] = __builtin_amdgcn_perm(a1, a0, sel[1]);
        }
      } else {
        const uint32_t d0 = p[0], d1 = p[1];
        px[0] = __builtin_amdgcn_alignbyte(d1, d0, sh);
        if constexpr (BORDER) px[0] = __builtin_amdgcn_perm(0u, px[0], sel[0]);
      }
#pragma unroll
      for (int d = 0; d < K / 4; ++d) {
        s_hi = __builtin_amdgcn_sdot4((int)px[d], (int)wh[k1 * (K / 4) + d], s_hi, false);
        s_lo = __builtin_amdgcn_sdot4((int)px[d], (int)wl[k1 * (K / 4) + d], s_lo, false);
      }
    }
    int sum = 256 * s_hi + s_lo + wconst;
    if (wave_rem && rem_k >= 0) sum += 256 * (int)(int8_t)(Lu[rem_off] >> rem_sh);
    int v = (sum + (1 << 14)) >> 15;
    return v < 0 ? 0 : (v > 255 ? 255 : v);
  };

  // ---- the image loop. NR staging registers per lane form a ring of D = min(4, NR / TC) boxes of up to TC dwords per
  // lane (TC = the smallest of 1, 2, 3, 4, 6, 8, 12, 16 that holds this tile's T): the boxes of the next D images are in flight while one
  // image is gathered from the wave's single LDS box. One image's gather is ~60 VALU instructions, a fraction of a
  // memory round trip, so a depth of one (r02 first form: 52 % VALU-busy at 14 waves per CU) left the waves waiting.
  // Loads, LDS writes and the ring are unconditional (slots past the box re-read its first dword and land in LDS
  // dwords nobody reads) so that the compiler's counted s_waitcnt vmcnt keeps the younger boxes in flight; images past
  // the wave's last one re-fetch that last image and are not gathered.
  // wave-uniform base in SGPRs + 32-bit lane offsets: global_load_dword v, v_off, s[base] -- no per-load address
  // arithmetic (the explicit global address space keeps the rebuilt pointer from degrading to flat loads, whose
  // waits cannot be counted per box)
  typedef const __attribute__((address_space(1))) uint8_t* gbytes_t;
  typedef const __attribute__((address_space(1))) uint32_t* gwords_t;
  gbytes_t base_u;
  {
    const uint64_t b = (uint64_t)(uintptr_t)b0;
    base_u = (gbytes_t)(uintptr_t)(((uint64_t)(uint32_t)__builtin_amdgcn_readfirstlane((int)(b >> 32)) << 32) |
                                   (uint32_t)__builtin_amdgcn_readfirstlane((int)(uint32_t)b));
  }
  const size_t img_stride = a.src_stride;
  const int n_img = __builtin_amdgcn_readfirstlane(img1 - img0);  // wave-uniform (the analysis cannot see it: img0 depends on the wave index)
  // top-left pixel of the super-tile in image img0 of the destination (row stores)
  uint8_t* dst_tile = a.dst + (size_t)img0 * a.dst_stride + (size_t)(SUPER ? (tile / (tiles / 2)) * 16 : 0) * res + (SUPER ? (tile % (tiles / 2)) * 16 : 0);
  uint32_t ring[NR];
  // the lanes that share a box meet here: one wave (its DS instructions execute in order), or the workgroup at a raw
  // s_barrier behind its own LDS traffic -- NOT __syncthreads(), whose fence would also drain the ring's loads in flight
  auto box_sync = [&]() {
    if constexpr (SUPER) asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
    else wave_sync();
  };
  auto pipeline = [&](auto tc_c) {
    constexpr int TC = decltype(tc_c)::value, D = NR / TC < 4 ? NR / TC : 4;
    auto fetch = [&](auto slot_c, int i) {  // image i (clamped to the wave's last) into ring slot `slot`
      constexpr int slot = decltype(slot_c)::value;
      gbytes_t src_i = base_u + (size_t)(i < n_img ? i : n_img - 1) * img_stride;
#pragma unroll
      for (int t = 0; t < TC; ++t) ring[slot * TC + t] = *(gwords_t)(src_i + goff[t]);
    };
    // the scheduler must not reorder the boxes' loads: the counted waits below rest on slot order == issue order, on
    // the entry path as on the loop's back edge
    static_for<0, D>([&](auto q) {
      fetch(q, (int)decltype(q)::value);
      __builtin_amdgcn_sched_barrier(0);
    });
    auto commit = [&](auto slot_c, uint32_t* Lb) {
      constexpr int slot = decltype(slot_c)::value;
#pragma unroll
      for (int t = 0; t < TC; ++t) Lb[st_tid + NT * t] = ring[slot * TC + t] ^ 0x80808080u;
    };
    // Two LDS boxes: the box of image i + 1 is committed while image i is gathered, so the lanes that share a box meet
    // ONCE per image (after both), and the freed ring slot is refilled right away.
    commit(std::integral_constant<int, 0>{}, L);
    __builtin_amdgcn_sched_barrier(0);
    fetch(std::integral_constant<int, 0>{}, D);
    __builtin_amdgcn_sched_barrier(0);
    box_sync();
    for (int i = 0; i < n_img; i += D) {
      static_for<0, D>([&](auto q) {
        constexpr int slot = decltype(q)::value, next = (slot + 1) % D;
        const int cur = i + slot;
        uint32_t* Lc = L + (cur & 1) * box_dwords;
        commit(std::integral_constant<int, next>{}, L + ((cur + 1) & 1) * box_dwords);  // image cur + 1 (ring slot `next`)
        __builtin_amdgcn_sched_barrier(0);
        fetch(std::integral_constant<int, next>{}, cur + 1 + D);
        __builtin_amdgcn_sched_barrier(0);
        uint8_t* Lout = reinterpret_cast<uint8_t*>(L + 2 * box_dwords) + (cur & 1) * 256;
        if (cur < n_img) {
          int v = 0;
          if (valid) v = wave_border ? gather(std::true_type{}, Lc) : gather(std::false_type{}, Lc);
          if (SUPER && row_store) Lout[((wave >> 1) * 8 + (lane >> 3)) * 16 + (wave & 1) * 8 + (lane & 7)] = (uint8_t)v;
          else if (valid || (a.zero_invalid && inside)) dst[(size_t)cur * a.dst_stride] = (uint8_t)v;
        }
        box_sync();  // image cur + 1 is staged, everyone is done with image cur (and its output tile is complete)
        if (SUPER && row_store && cur < n_img && threadIdx.x < 16) {
          typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
          const u32x4 rowv = *reinterpret_cast<const u32x4*>(Lout + 16 * threadIdx.x);
          __builtin_memcpy(dst_tile + (size_t)cur * a.dst_stride + (size_t)threadIdx.x * res, &rowv, 16);
        }
      });
    }
  };
  // ring classes: 60 % of the 480^2 map's tiles have a box of at most 64 dwords (T = 1), the mean is 2.2 -- rounding T
  // up to {4, 8, 16} issued 2.2x the loads and made the texture addresser the limit (TA_BUSY 85 %)
  static_assert(NR == 16 || NR == 12, "ring classes are written for 12 or 16 staging registers");
  if (T <= 1) pipeline(std::integral_constant<int, 1>{});
  else if (T <= 2) pipeline(std::integral_constant<int, 2>{});
  else if (T <= 3) pipeline(std::integral_constant<int, 3>{});
  else if (T <= 4) pipeline(std::integral_constant<int, 4>{});
  else if (T <= 6) pipeline(std::integral_constant<int, 6>{});
  else if (T <= 8) pipeline(std::integral_constant<int, 8>{});
  else if (NR < 16 || T <= 12) pipeline(std::integral_constant<int, 12>{});
  else if constexpr (NR >= 16) pipeline(std::integral_constant<int, 16>{});
}

// ---- K5: forward row transforms of z = cur_lp + i prev_lp, written transposed -----------------------------
#ifndef MOF_SR_FWD_ROWS
#define MOF_SR_FWD_ROWS 16
#endif
constexpr int FWD_ROWS = MOF_SR_FWD_ROWS;  // rows per workgroup: the transposed store writes FWD_ROWS * 8 bytes per segment
constexpr int FWD_T = FWD_ROWS * 16;       // four rows per wave

template <int N>
__global__ void __launch_bounds__(FWD_T) sr_rows_fwd_kernel(SrPcArgs a) {
  using P = SrPlan<N>;
  extern __shared__ __attribute__((aligned(16))) unsigned char fwd_lds[];
  cf* z = reinterpret_cast<cf*>(fwd_lds);  // [FWD_ROWS][LINE]
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pair = blockIdx.y, row0 = blockIdx.x * FWD_ROWS;
  SrTw<N> tw;
  tw.load(a.twiddles, lane);
  // wave w owns rows row0 + 4w .. +3: it loads them (4 px of cur and prev per lane and step), transforms them ...
  const uint8_t* cur = a.lp_cur + (size_t)pair * a.lp_stride + (size_t)(row0 + 4 * wave) * N;
  const uint8_t* prev = a.lp_prev + (size_t)pair * a.lp_stride + (size_t)(row0 + 4 * wave) * N;
  cf* mine = z + 4 * wave * P::LINE;
  {
    // all loads first (a load-use loop would pay the memory latency once per trip), then the conversion
    constexpr int NL = (N + 63) / 64;  // dwords per lane: four rows of N bytes
    uint32_t c[NL], p[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = lane + 64 * k;
      if (i < N) {
        c[k] = stream_load(reinterpret_cast<const uint32_t*>(cur + 4 * i));
        p[k] = stream_load(reinterpret_cast<const uint32_t*>(prev + 4 * i));
      }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = lane + 64 * k;
      if (i < N) {
        const int l = (4 * i) / N, x = (4 * i) % N;
#pragma unroll
        for (int q = 0; q < 4; ++q)  // convertTo CV_32FC1, :115
          mine[l * P::LINE + x + q] = {(float)((c[k] >> (8 * q)) & 0xffu), (float)((p[k] >> (8 * q)) & 0xffu)};
      }
    }
  }
  wave_sync();
  wave_fft<N>(mine, 4, lane, tw, StoreNatural<N>{});
  __syncthreads();
  // ... and the workgroup stores its rows transposed: Zt[u][row0 .. row0 + FWD_ROWS) is one contiguous segment per u
  cf* Zt = reinterpret_cast<cf*>(a.Zt) + (size_t)pair * N * N + row0;
  // (two neighbouring rows per lane: 16-byte non-temporal stores, as K5s -- the 8-byte form cost 16 % more on this kernel)
  static_assert(FWD_ROWS % 2 == 0, "row pairs");
  for (int i = tid; i < FWD_ROWS / 2 * N; i += FWD_T) {
    const int u = i / (FWD_ROWS / 2), dv = 2 * (i % (FWD_ROWS / 2));
    const cf a0 = z[dv * P::LINE + u], a1 = z[(dv + 1) * P::LINE + u];
    stream_store(reinterpret_cast<float4*>(&Zt[(size_t)u * N + dv]), make_float4(a0.x, a0.y, a1.x, a1.y));
  }
}

// ---- K6: column transforms, cross-power spectrum, inverse column transforms of the half spectrum -----------
template <int N>
__global__ void __launch_bounds__(SR_T) sr_cols_kernel(SrPcArgs a) {
  using P = SrPlan<N>;
  constexpr int H = N / 2, CW = COLS_CW;
  __shared__ cf z[2 * CW * P::LINE];  // lines 0..CW-1: columns u0+s ; lines CW..2CW-1: their mirrors (N - u) % N
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pair = blockIdx.y, u0 = blockIdx.x * CW;
  SrTw<N> tw;
  tw.load(a.twiddles, lane);
  const cf* Zt = reinterpret_cast<const cf*>(a.Zt) + (size_t)pair * N * N;
  // wave 0 loads and transforms the columns, wave 1 their mirrors: whole lines of N complex, 16 bytes per lane
  {
    cf* mine = z + wave * CW * P::LINE;
    constexpr int NL = (CW * N / 2 + 63) / 64;  // 16-byte loads per lane, all in flight before the first LDS write
    float4 t[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = lane + 64 * k;
      if (i < CW * N / 2) {
        const int s = (2 * i) / N, v = (2 * i) % N;
        int u = u0 + s;
        if (u > H) u = H;  // tail group: clamp (results of clamped lines are never stored)
        const int col = wave == 0 ? u : (N - u) % N;
        t[k] = stream_load(reinterpret_cast<const float4*>(Zt + (size_t)col * N + v));
      }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = lane + 64 * k;
      if (i < CW * N / 2) {
        const int s = (2 * i) / N, v = (2 * i) % N;
        mine[s * P::LINE + v] = {t[k].x, t[k].y};
        mine[s * P::LINE + v + 1] = {t[k].z, t[k].w};
      }
    }
    wave_sync();
    wave_fft<N>(mine, CW, lane, tw, StoreNatural<N>{});
  }
  __syncthreads();
  // normalised cross-power spectrum of bins (v, u), conjugated in place (see K1 for the rules)
  for (int i = tid; i < CW * N; i += SR_T) {
    const int s = i / N, v = i % N;
    int u = u0 + s;
    if (u > H) u = H;
    const cf zk = z[s * P::LINE + v], zm = z[(CW + s) * P::LINE + (N - v) % N];
    const bool real_only = (v == 0 || v == H) && (u == 0 || u == H);
    // The DC bin is exact in the packed transform: (sum cur, sum prev). A sum of zero = an all-zero (black) image, whose
    // spectrum is exactly zero in the reference's separate transforms: P = 0 everywhere, flat zero surface, pt = (N/2, N/2).
    // The packed form would leak rounding noise of the other image into it (see pc_common.hpp, degenerate pairs): tell K8.
    if (u0 == 0 && i == 0 && a.degen) a.degen[pair] = (zk.x == 0.f || zk.y == 0.f) ? 1 : 0;
    const cf C = cross_power(zk, zm, real_only);
    z[s * P::LINE + v] = {C.x, -C.y};
  }
  __syncthreads();
  // the CW lines of the half spectrum: two per wave
  wave_fft<N>(z + 2 * wave * P::LINE, 2, lane, tw, StoreNatural<N>{});
  __syncthreads();
  cf* Dt = reinterpret_cast<cf*>(a.Dt) + (size_t)pair * (H + 1) * N;
  for (int i = tid; i < CW * N / 2; i += SR_T) {
    const int s = (2 * i) / N, y = (2 * i) % N, u = u0 + s;
    if (u <= H) {
      const cf a0 = z[s * P::LINE + y], a1 = z[s * P::LINE + y + 1];
      stream_store(reinterpret_cast<float4*>(Dt + (size_t)u * N + y), make_float4(a0.x, a0.y, a1.x, a1.y));
    }
  }
}

// ---- K7: Hermitian rows back to the real surface, two rows per complex transform; arg-max from registers -----
// K7's waves: two of four lines each; at the long lines (r06, N >= 540: first radix above 16, both stages of wave_fft take two lines per pass
// anyway) four of two lines each -- the eight lines in LDS set the workgroups per CU, so that is twice the waves for the same LDS
#ifndef MOF_K7_LPW2_FROM
#define MOF_K7_LPW2_FROM 540
#endif
template <int N>
struct K7Cfg {
  static constexpr int LPW = (SrPlan<N>::R1 > 16 || N >= MOF_K7_LPW2_FROM) ? 2 : 4;  // (also 324 / 486 / 500: +1 .. 4 %)
  static constexpr int T = SR_LINES / LPW * 64;
};
template <int N>
__global__ void __launch_bounds__(K7Cfg<N>::T) sr_rows_inv_kernel(SrPcArgs a) {
  using P = SrPlan<N>;
  constexpr int H = N / 2, K7T = K7Cfg<N>::T, LPW = K7Cfg<N>::LPW;
  __shared__ cf z[SR_LINES * P::LINE];
  __shared__ Best red[K7T / 64];
  const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, pair = blockIdx.y, p0 = blockIdx.x * SR_LINES;
  SrTw<N> tw;
  tw.load(a.twiddles, lane);
  const cf* Dt = reinterpret_cast<const cf*>(a.Dt) + (size_t)(MOF_SR_L2_ABLATE ? 0 : pair) * (H + 1) * N;
  // line l carries rows y1 = 2 (p0 + l), y2 = y1 + 1: E[u] = F[y1][u] + i F[y2][u], F[y][N-u] = conj F[y][u].
  // Dt[u][y1], Dt[u][y2] are neighbours: one 16-byte load; 8 lanes fetch the 16 rows of the workgroup (128 bytes).
  constexpr bool ODD = (N & 1) != 0;   // (r06) the last row of an odd image shares its line with zeros; every u > 0 has a partner N - u != u
  constexpr int NLN = (N + 1) / 2;     // lines = row pairs
  {
    constexpr int NL = (SR_LINES * (H + 1) + K7T - 1) / K7T;  // all loads in flight before the first LDS write
    float4 t[NL];
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + K7T * k;
      constexpr bool TAIL = NLN % SR_LINES != 0;  // (200, 216: a line past the last row pair transforms zeros and is left out of the arg-max)
      if constexpr (TAIL) t[k] = make_float4(0.f, 0.f, 0.f, 0.f);
      if (i < SR_LINES * (H + 1) && (!TAIL || p0 + i % SR_LINES < NLN)) {
        const cf* d = Dt + (size_t)(i / SR_LINES) * N + 2 * (p0 + i % SR_LINES);
        if constexpr (ODD) {  // (rows of N complex are not 16-byte aligned; the row behind the last one is zero)
          const cf g1 = d[0], g2 = 2 * (p0 + i % SR_LINES) + 1 < N ? d[1] : cf{0.f, 0.f};
          t[k] = make_float4(g1.x, g1.y, g2.x, g2.y);
        } else {
          t[k] = stream_load(reinterpret_cast<const float4*>(d));
        }
      }
    }
#pragma unroll
    for (int k = 0; k < NL; ++k) {
      const int i = tid + K7T * k;
      if (i < SR_LINES * (H + 1)) {
        const int u = i / SR_LINES, l = i % SR_LINES;
        z[l * P::LINE + u] = {t[k].x - t[k].w, t[k].y + t[k].z};
        if (u > 0 && (ODD || u < H)) z[l * P::LINE + N - u] = {t[k].x + t[k].w, t[k].z - t[k].y};
      }
    }
  }
  __syncthreads();
  Best best = {-__builtin_huge_valf(), 0x7fffffff};
  wave_fft<N>(z + LPW * wave * P::LINE, LPW, lane, tw, [&](cf*, int l, int k1, const cf* v) {
    if (NLN % SR_LINES != 0 && p0 + LPW * wave + l >= NLN) return;
    const int y1 = 2 * (p0 + LPW * wave + l), y2 = y1 + 1;
    const int r1 = ((y1 + H) % N) * N, r2 = ((y2 + H) % N) * N;  // fftShift + first maximum (minMaxLoc)
#pragma unroll
    for (int k2 = 0; k2 < P::R2; ++k2) {
      const int xs = (k1 + P::R1 * k2 + H) % N;
      best = better(best, Best{v[k2].x, r1 + xs});
      if (!ODD || y2 < N) best = better(best, Best{v[k2].y, r2 + xs});
    }
  });
#pragma unroll
  for (int off = 32; off > 0; off >>= 1) {
    Best o = {__shfl_xor(best.v, off, 64), __shfl_xor(best.idx, off, 64)};
    best = better(best, o);
  }
  if (lane == 0) red[wave] = best;
  __syncthreads();
  if (tid == 0) {
    for (int w = 1; w < K7T / 64; ++w) best = better(best, red[w]);
    a.cand[(size_t)pair * a.n_cand + blockIdx.x] = make_float2(best.v, __int_as_float(best.idx));
  }
}

// ---- K8: peak, centroid, (scale, rot) ----------------------------------------------------------------------
// out[pair] = {scale, rot, pt.x, pt.y}; pt = cv::phaseCorrelate(cur_lp, prev_lp) = center - t (NOT negated, :117);
// |pt.x| > res/2 -> (1, 0) (:119-121); scale = exp(pt.x / M), rot = (pt.y / Ky) pi/180, Ky = res/360 (:123-124).
// The 25 surface values of the window are re-evaluated from the half spectrum of their rows (Dt), in double:
//   S[y][x] = Re F[y][0] + (-1)^x Re F[y][N/2] + 2 sum_{u=1}^{N/2-1} (Re F[y][u] cos(2 pi u x / N) + Im F[y][u] sin(..))
// which is what K7's transform computes for that point (K7 keeps no surface).
template <int N>
__global__ void __launch_bounds__(64) sr_final_kernel(SrPcArgs a) {
  constexpr int H = N / 2;
  __shared__ double part[25][65];
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
  const cf* Dt = reinterpret_cast<const cf*>(a.Dt) + (size_t)pair * (H + 1) * N;
  const bool have = best.idx != 0x7fffffff;
  const int px = have ? best.idx % N : 0, py = have ? best.idx / N : 0;
  // window rows / columns in un-shifted coordinates (entries outside the clamped window are skipped at the end)
  int wy[5], wx[5];
#pragma unroll
  for (int k = 0; k < 5; ++k) {
    wy[k] = (((py - 2 + k) % N + N) + H) % N;
    wx[k] = (((px - 2 + k) % N + N) + H) % N;
  }
  // the 240-term sums are split over the lanes by u (lane, lane + 64, ..), 25 partial sums each, ...
  double acc[25];
#pragma unroll
  for (int k = 0; k < 25; ++k) acc[k] = 0.0;
  for (int u = 1 + lane; u < H; u += 64) {
    cf f[5];
#pragma unroll
    for (int r = 0; r < 5; ++r) f[r] = Dt[(size_t)u * N + wy[r]];
#pragma unroll
    for (int c = 0; c < 5; ++c) {
      const float2 w = *reinterpret_cast<const float2*>(a.twiddles + 2 * (int)(((long)u * wx[c]) % N));  // (cos, -sin)
#pragma unroll
      for (int r = 0; r < 5; ++r) acc[r * 5 + c] += (double)f[r].x * (double)w.x - (double)f[r].y * (double)w.y;
    }
  }
#pragma unroll
  for (int k = 0; k < 25; ++k) part[k][lane] = acc[k];
  __syncthreads();
  // ... and lane k < 25 adds the 64 partial sums of its window point in a fixed order
  const int ys = py - 2 + lane / 5, xs = px - 2 + lane % 5;
  double cx = 0.0, cy = 0.0, sum = 0.0;
  if (have && lane < 25 && ys >= 0 && ys <= N - 1 && xs >= 0 && xs <= N - 1) {
    const int y = wy[lane / 5], x = wx[lane % 5];
    double s2 = 0.0;
    for (int l = 0; l < 64; ++l) s2 += part[lane][l];
    const double s0 = (double)Dt[y].x + ((x & 1) ? -1.0 : 1.0) * (double)Dt[(size_t)H * N + y].x;
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
  if (lane == 0) {
    sum += 2.220446049250313e-16;
    double ptx = (double)N / 2.0 - cx / sum, pty = (double)N / 2.0 - cy / sum;
    if (a.degen && a.degen[pair]) ptx = pty = (double)N / 2.0;  // flat zero surface: first index, centroid 0 / (0 + eps)
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

// Resolutions whose transforms are the tuned in-register ones (K5s / K6s / K7): the estimator's own three, and (r06) every transform size the
// FFT engine's large patches brought along whose Nyquist bin is exact (250 / 400 / 432 would need the exact-sums form of the row kernel: they stay
// on the planned pipeline, like every resolution that is not itself one of these sizes). MOF_SR_TUNED_ALL=0: the three only (A/B, tests).
bool sr_transform_size_tuned(int m, bool* exact_nyquist) {
  static const int exact[] = {225, 243, 375, 405, 625, 675, 729,  // (odd: no Nyquist bin to keep exact)
                              96, 100, 108, 120, 150, 162, 128, 144, 160, 180, 192, 200, 216, 240, 256, 270, 288, 300, 320, 324, 360, 384, 450, 480, 486, 500, 512, 540, 576, 600, 640, 648, 720, 750, 768, 800, 810, 864, 900, 960};
  for (int t : exact)
    if (m == t) {
      if (exact_nyquist) *exact_nyquist = true;
      return true;
    }
  if (m == 250 || m == 400 || m == 432) {  // odd last radix: the real-only slots come from the images' exact integer sums
    if (exact_nyquist) *exact_nyquist = false;
    return true;
  }
  return false;
}
bool sr_pair_kernels_supported(int res) { return res == 240 || res == 256 || res == 480; }  // K5 / K6 (packed pairs), K56, K6p
bool sr_resolution_supported(int res) {
  static const bool all = [] { const char* v = getenv("MOF_SR_TUNED_ALL"); return !v || atoi(v) != 0; }();
  if (sr_pair_kernels_supported(res)) return true;
  if (!all) return false;
  static const int sizes[] = {96, 100, 108, 120, 150, 162, 128, 144, 160, 180, 192, 200, 216, 270, 288, 300, 320, 324, 360, 384, 450, 486, 500, 512, 540, 576, 600, 640, 648, 720, 750, 768, 800, 810, 864, 900, 960};
  for (int t : sizes)
    if (res == t) return true;
  return false;
}

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
  // batches whose rows and frames keep one dword alignment: tile-stationary kernel with LDS-staged source boxes
  // (MOF_SR_LP_STAGED=0 falls back to the table-in-LDS formulation, kept for A/B runs and unaligned layouts)
  static const bool staged_on = [] { const char* e = getenv("MOF_SR_LP_STAGED"); return !e || atoi(e) != 0; }();
  constexpr int NR = 16;
  // (r06) layouts whose rows / frames are not whole dwords apart are staged from unaligned dwords: the last dword of the LAST row of an image may
  // then reach up to three bytes past that image -- into the next one, harmless, except behind the last image of the caller's buffer. So the
  // last image of such a batch takes the per-pixel kernel (`finish`), the others the staged one.
  const bool dword_layout = ((a.pitch | a.src_stride) & 3) == 0;
  const int n_st = dword_layout ? n_images : n_images - 1;  // images of the staged launch
  auto finish = [&]() -> hipError_t {
    if (n_st < n_images) {
      SrLpArgs last = a;
      last.src = a.src + (size_t)n_st * a.src_stride;
      last.dst = a.dst + (size_t)n_st * a.dst_stride;
      const dim3 grid1((unsigned)(((a.res + 31) / 32) * ((a.res + 7) / 8)), 1u);
      if (interp == 2) hipLaunchKernelGGL(sr_logpolar_kernel<4>, grid1, dim3(256), 0, stream, last);
      else hipLaunchKernelGGL(sr_logpolar_kernel<8>, grid1, dim3(256), 0, stream, last);
    }
    return hipGetLastError();
  };
  if (staged_on && !global_w && n_st >= 4 && a.boxes && a.wplanes && a.box_dwords_max <= 64 * NR) {
    const int tiles = (a.res + 7) / 8, n_tiles = tiles * tiles;
    // images per wave: long runs amortise the per-tile set-up; short ones keep the images that the resident waves
    // share within the L2 (MOF_SR_LP_IPW: diagnostic override)
    static const int forced_ipw = [] { const char* e = getenv("MOF_SR_LP_IPW"); return e ? atoi(e) : 0; }();
    const int ipw = forced_ipw > 0 ? forced_ipw : (n_st >= 128 ? 32 : (n_st >= 64 ? 16 : (n_st >= 16 ? 8 : 4)));
    const int groups = (n_st + ipw - 1) / ipw;
    static const bool xcd_off = [] { const char* e = getenv("MOF_SR_LP_XCD"); return e && atoi(e) == 0; }();
    const int nq = (n_tiles + 3) / 4;
    const int xcd_groups = (groups >= 8 && !xcd_off) ? groups : 0;  // fewer than 8 groups would leave XCDs idle
    const unsigned blocks = xcd_groups ? (unsigned)(8 * ((groups + 7) / 8) * nq) : (unsigned)(((long)n_tiles * groups + 3) / 4);
    // shared box per 16 x 16 super-tile (four waves) where the map's largest one fits the staging ring (256 x NR dwords)
    static const bool super_off = [] { const char* e = getenv("MOF_SR_LP_SUPER"); return e && atoi(e) == 0; }();
    if (!super_off && a.sboxes && a.res % 16 == 0 && a.sbox_dwords_max <= 256 * NR) {
      const int n_super = (tiles / 2) * (tiles / 2);
      const unsigned sblocks = xcd_groups ? (unsigned)(8 * ((groups + 7) / 8) * n_super) : (unsigned)((long)n_super * groups);
      const size_t slds = sizeof(uint32_t) * (size_t)2 * 4 * lp_box_capacity((a.sbox_dwords_max + 3) / 4) + 512;  // two boxes + two output tiles
      constexpr int NRS = 12;  // the shorter ring where every box of the map fits it (480^2, M = 49.9: 2472 dwords of 3072)
      // (MOF_SR_LP_RING=16 keeps the 16-deep form for maps that fit the short one: A/B runs and the tests of that form)
      static const bool long_ring = [] { const char* e = getenv("MOF_SR_LP_RING"); return e && atoi(e) == 16; }();
      const bool short_ring = !long_ring && a.sbox_dwords_max <= 256 * NRS;
      if (interp == 2 && short_ring)
        hipLaunchKernelGGL((sr_logpolar_staged_kernel<4, NRS, true>), dim3(sblocks), dim3(256), slds, stream, a, n_st, ipw, xcd_groups);
      else if (interp == 2)
        hipLaunchKernelGGL((sr_logpolar_staged_kernel<4, NR, true>), dim3(sblocks), dim3(256), slds, stream, a, n_st, ipw, xcd_groups);
      else if (short_ring)
        hipLaunchKernelGGL((sr_logpolar_staged_kernel<8, NRS, true>), dim3(sblocks), dim3(256), slds, stream, a, n_st, ipw, xcd_groups);
      else
        hipLaunchKernelGGL((sr_logpolar_staged_kernel<8, NR, true>), dim3(sblocks), dim3(256), slds, stream, a, n_st, ipw, xcd_groups);
      return finish();
    }
    const size_t lds = (size_t)4 * 2 * sizeof(uint32_t) * lp_box_capacity(a.box_dwords_max);
    if (interp == 2)
      hipLaunchKernelGGL((sr_logpolar_staged_kernel<4, NR, false>), dim3(blocks), dim3(256), lds, stream, a, n_st, ipw, xcd_groups);
    else
      hipLaunchKernelGGL((sr_logpolar_staged_kernel<8, NR, false>), dim3(blocks), dim3(256), lds, stream, a, n_st, ipw, xcd_groups);
    return finish();
  }
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
  static_assert(N % FWD_ROWS == 0 && H % SR_LINES == 0, "rows and row pairs divide evenly over the workgroups");
  constexpr size_t fwd_lds = sizeof(cf) * FWD_ROWS * SrPlan<N>::LINE;
  if (fwd_lds > 48 * 1024) {
    hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&sr_rows_fwd_kernel<N>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)fwd_lds);
    if (e != hipSuccess) return e;
  }
  hipLaunchKernelGGL(sr_rows_fwd_kernel<N>, dim3(N / FWD_ROWS, (unsigned)n_pairs), dim3(FWD_T), fwd_lds, stream, a);
  hipLaunchKernelGGL(sr_cols_kernel<N>, dim3((H + 1 + COLS_CW - 1) / COLS_CW, (unsigned)n_pairs), dim3(SR_T), 0, stream, a);
  hipLaunchKernelGGL(sr_rows_inv_kernel<N>, dim3(H / SR_LINES, (unsigned)n_pairs), dim3(K7Cfg<N>::T), 0, stream, a);
  hipLaunchKernelGGL(sr_final_kernel<N>, dim3((unsigned)n_pairs), dim3(64), 0, stream, a);
  return hipGetLastError();
}

int sr_candidates(int res) { return ((res + 1) / 2 + SR_LINES - 1) / SR_LINES; }  // (one per workgroup of K7: eight row pairs)

// K7 + K8 alone: from a Dt that somebody else produced (the sequence pipeline, sr_seq_kernel.hip) to (scale, rot, pt)
template <int N>
static hipError_t launch_sr_peak_n(const SrPcArgs& a, int n_pairs, hipStream_t stream) {
  hipLaunchKernelGGL(sr_rows_inv_kernel<N>, dim3(((N + 1) / 2 + SR_LINES - 1) / SR_LINES, (unsigned)n_pairs), dim3(K7Cfg<N>::T), 0, stream, a);
  hipLaunchKernelGGL(sr_final_kernel<N>, dim3((unsigned)n_pairs), dim3(64), 0, stream, a);
  return hipGetLastError();
}

hipError_t launch_sr_peak(const SrPcArgs& a, int res, int n_pairs, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  switch (res) {
    case 240: return launch_sr_peak_n<240>(a, n_pairs, stream);
    case 256: return launch_sr_peak_n<256>(a, n_pairs, stream);
    case 480: return launch_sr_peak_n<480>(a, n_pairs, stream);
    case 128: return launch_sr_peak_n<128>(a, n_pairs, stream);
    case 96: return launch_sr_peak_n<96>(a, n_pairs, stream);
    case 100: return launch_sr_peak_n<100>(a, n_pairs, stream);
    case 108: return launch_sr_peak_n<108>(a, n_pairs, stream);
    case 120: return launch_sr_peak_n<120>(a, n_pairs, stream);
    case 150: return launch_sr_peak_n<150>(a, n_pairs, stream);
    case 162: return launch_sr_peak_n<162>(a, n_pairs, stream);
    case 144: return launch_sr_peak_n<144>(a, n_pairs, stream);
    case 160: return launch_sr_peak_n<160>(a, n_pairs, stream);
    case 180: return launch_sr_peak_n<180>(a, n_pairs, stream);
    case 192: return launch_sr_peak_n<192>(a, n_pairs, stream);
    case 200: return launch_sr_peak_n<200>(a, n_pairs, stream);
    case 216: return launch_sr_peak_n<216>(a, n_pairs, stream);
    case 270: return launch_sr_peak_n<270>(a, n_pairs, stream);
    case 288: return launch_sr_peak_n<288>(a, n_pairs, stream);
    case 300: return launch_sr_peak_n<300>(a, n_pairs, stream);
    case 320: return launch_sr_peak_n<320>(a, n_pairs, stream);
    case 324: return launch_sr_peak_n<324>(a, n_pairs, stream);
    case 360: return launch_sr_peak_n<360>(a, n_pairs, stream);
    case 384: return launch_sr_peak_n<384>(a, n_pairs, stream);
    case 450: return launch_sr_peak_n<450>(a, n_pairs, stream);
    case 486: return launch_sr_peak_n<486>(a, n_pairs, stream);
    case 500: return launch_sr_peak_n<500>(a, n_pairs, stream);
    case 512: return launch_sr_peak_n<512>(a, n_pairs, stream);
    case 540: return launch_sr_peak_n<540>(a, n_pairs, stream);
    case 576: return launch_sr_peak_n<576>(a, n_pairs, stream);
    case 600: return launch_sr_peak_n<600>(a, n_pairs, stream);
    case 640: return launch_sr_peak_n<640>(a, n_pairs, stream);
    case 648: return launch_sr_peak_n<648>(a, n_pairs, stream);
    case 720: return launch_sr_peak_n<720>(a, n_pairs, stream);
    case 750: return launch_sr_peak_n<750>(a, n_pairs, stream);
    case 768: return launch_sr_peak_n<768>(a, n_pairs, stream);
    case 800: return launch_sr_peak_n<800>(a, n_pairs, stream);
    case 810: return launch_sr_peak_n<810>(a, n_pairs, stream);
    case 864: return launch_sr_peak_n<864>(a, n_pairs, stream);
    case 900: return launch_sr_peak_n<900>(a, n_pairs, stream);
    case 960: return launch_sr_peak_n<960>(a, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}

// K7 alone (r04): Dt -> peak candidates, for the FFT engine's large patches of 240 / 256 / 480 pixels (pc_large_kernel.hip's L8 follows)
template <int N>
static hipError_t launch_sr_rows_inv_n(const SrPcArgs& a, int n_pairs, hipStream_t stream) {
  for (int p0 = 0; p0 < n_pairs; p0 += 65535) {
    const int np = n_pairs - p0 < 65535 ? n_pairs - p0 : 65535;
    SrPcArgs b = a;
    b.Dt = a.Dt + (size_t)p0 * (N / 2 + 1) * N * 2;
    b.cand = a.cand + (size_t)p0 * a.n_cand;
    hipLaunchKernelGGL(sr_rows_inv_kernel<N>, dim3(((N + 1) / 2 + SR_LINES - 1) / SR_LINES, (unsigned)np), dim3(K7Cfg<N>::T), 0, stream, b);
  }
  return hipGetLastError();
}
hipError_t launch_sr_rows_inv(const float* Dt, const float* twiddles, float2* cand, int res, int n_pairs, hipStream_t stream) {
  if (n_pairs <= 0) return hipSuccess;
  SrPcArgs a{};
  a.Dt = const_cast<float*>(Dt);
  a.twiddles = twiddles;
  a.cand = cand;
  a.n_cand = sr_candidates(res);
  switch (res) {
    case 200: return launch_sr_rows_inv_n<200>(a, n_pairs, stream);
    case 216: return launch_sr_rows_inv_n<216>(a, n_pairs, stream);
    case 240: return launch_sr_rows_inv_n<240>(a, n_pairs, stream);
    case 256: return launch_sr_rows_inv_n<256>(a, n_pairs, stream);
    case 250: return launch_sr_rows_inv_n<250>(a, n_pairs, stream);
    case 400: return launch_sr_rows_inv_n<400>(a, n_pairs, stream);
    case 432: return launch_sr_rows_inv_n<432>(a, n_pairs, stream);
    case 270: return launch_sr_rows_inv_n<270>(a, n_pairs, stream);
    case 300: return launch_sr_rows_inv_n<300>(a, n_pairs, stream);
    case 450: return launch_sr_rows_inv_n<450>(a, n_pairs, stream);
    case 288: return launch_sr_rows_inv_n<288>(a, n_pairs, stream);
    case 320: return launch_sr_rows_inv_n<320>(a, n_pairs, stream);
    case 360: return launch_sr_rows_inv_n<360>(a, n_pairs, stream);
    case 384: return launch_sr_rows_inv_n<384>(a, n_pairs, stream);
    case 480: return launch_sr_rows_inv_n<480>(a, n_pairs, stream);
    case 512: return launch_sr_rows_inv_n<512>(a, n_pairs, stream);
    case 225: return launch_sr_rows_inv_n<225>(a, n_pairs, stream);
    case 243: return launch_sr_rows_inv_n<243>(a, n_pairs, stream);
    case 375: return launch_sr_rows_inv_n<375>(a, n_pairs, stream);
    case 405: return launch_sr_rows_inv_n<405>(a, n_pairs, stream);
    case 625: return launch_sr_rows_inv_n<625>(a, n_pairs, stream);
    case 675: return launch_sr_rows_inv_n<675>(a, n_pairs, stream);
    case 729: return launch_sr_rows_inv_n<729>(a, n_pairs, stream);
    case 128: return launch_sr_rows_inv_n<128>(a, n_pairs, stream);
    case 96: return launch_sr_rows_inv_n<96>(a, n_pairs, stream);
    case 100: return launch_sr_rows_inv_n<100>(a, n_pairs, stream);
    case 108: return launch_sr_rows_inv_n<108>(a, n_pairs, stream);
    case 120: return launch_sr_rows_inv_n<120>(a, n_pairs, stream);
    case 150: return launch_sr_rows_inv_n<150>(a, n_pairs, stream);
    case 162: return launch_sr_rows_inv_n<162>(a, n_pairs, stream);
    case 144: return launch_sr_rows_inv_n<144>(a, n_pairs, stream);
    case 160: return launch_sr_rows_inv_n<160>(a, n_pairs, stream);
    case 180: return launch_sr_rows_inv_n<180>(a, n_pairs, stream);
    case 192: return launch_sr_rows_inv_n<192>(a, n_pairs, stream);
    case 324: return launch_sr_rows_inv_n<324>(a, n_pairs, stream);
    case 486: return launch_sr_rows_inv_n<486>(a, n_pairs, stream);
    case 500: return launch_sr_rows_inv_n<500>(a, n_pairs, stream);
    case 540: return launch_sr_rows_inv_n<540>(a, n_pairs, stream);
    case 576: return launch_sr_rows_inv_n<576>(a, n_pairs, stream);
    case 600: return launch_sr_rows_inv_n<600>(a, n_pairs, stream);
    case 640: return launch_sr_rows_inv_n<640>(a, n_pairs, stream);
    case 648: return launch_sr_rows_inv_n<648>(a, n_pairs, stream);
    case 720: return launch_sr_rows_inv_n<720>(a, n_pairs, stream);
    case 750: return launch_sr_rows_inv_n<750>(a, n_pairs, stream);
    case 768: return launch_sr_rows_inv_n<768>(a, n_pairs, stream);
    case 800: return launch_sr_rows_inv_n<800>(a, n_pairs, stream);
    case 810: return launch_sr_rows_inv_n<810>(a, n_pairs, stream);
    case 864: return launch_sr_rows_inv_n<864>(a, n_pairs, stream);
    case 900: return launch_sr_rows_inv_n<900>(a, n_pairs, stream);
    case 960: return launch_sr_rows_inv_n<960>(a, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}

hipError_t launch_sr_phase_correlate(const SrPcArgs& a, int res, int n_pairs, hipStream_t stream) {
  switch (res) {
    case 240: return launch_sr_pc_n<240>(a, n_pairs, stream);
    case 256: return launch_sr_pc_n<256>(a, n_pairs, stream);
    case 480: return launch_sr_pc_n<480>(a, n_pairs, stream);
    default: return hipErrorInvalidValue;
  }
}

}  // namespace mof
