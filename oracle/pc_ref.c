/*
 * pc_ref.c -- CPU restatement of the reference's useOCL=false FFT path.
 * TEST INFRASTRUCTURE ONLY; PARITY UNPINNED (see oracle.h for why).
 * Citations are into /root/reference/src/FftMethod.cpp unless noted.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

static int largest_prime_factor(int n) {
  int best = 1;
  for (int p = 2; p * p <= n; ++p)
    while (n % p == 0) {
      best = p;
      n /= p;
    }
  if (n > 1) best = n;
  return best;
}

/* cv::getOptimalDFTSize(n): the smallest m >= n of the form 2^a 3^b 5^c (OpenCV looks it up in a table of exactly those
 * numbers, optimalDFTSizeTab) [published OpenCV algorithm, unpinned]. */
int oracle_optimal_dft_size(int n) {
  if (n < 1) return -1;
  for (int m = n;; ++m) {
    int r = m;
    while (r % 2 == 0) r /= 2;
    while (r % 3 == 0) r /= 3;
    while (r % 5 == 0) r /= 5;
    if (r == 1) return m;
    if (m == 2147483647) return -1;
  }
}

#define R float
#define SUFFIX _f32
#define FMA_R fmaf
#define SQRT_R sqrtf
#include "pc_ref_impl.h"
#undef R
#undef SUFFIX
#undef FMA_R
#undef SQRT_R

#define R double
#define SUFFIX _f64
#define FMA_R fma
#define SQRT_R sqrt
#include "pc_ref_impl.h"
#undef R
#undef SUFFIX
#undef FMA_R
#undef SQRT_R

int oracle_fft_process_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_fft_layout* L,
                          int precision, double* out_xy, int* n_invalid, oracle_pc_diag* diag) {
  if (!cur || !prev || !L || !out_xy) return -1;
  const int n = L->patch;
  if (n < 2) return -1; /* any patch size: cv::phaseCorrelate pads to getOptimalDFTSize(n) itself */
  if (L->grid_x < 1 || L->grid_y < 1 || L->origin_x < 0 || L->origin_y < 0) return -1;
  if (L->origin_x + (L->grid_x - 1) * L->stride_x + n > L->width) return -1;
  if (L->origin_y + (L->grid_y - 1) * L->stride_y + n > L->height) return -1;
  if (precision != 32 && precision != 64) return -1;

  const size_t nn = (size_t)n * n;
  float* af = (float*)malloc(sizeof(float) * nn * 2);
  double* ad = (double*)malloc(sizeof(double) * nn * 2);
  if (!af || !ad) { free(af); free(ad); return -2; }
  int invalid = 0;
  const double max_sq = L->max_px_speed * L->max_px_speed; /* ref :1686 pow(max_px_speed_t, 2) */

  for (int j = 0; j < L->grid_y; ++j)
    for (int i = 0; i < L->grid_x; ++i) {
      const int xi = L->origin_x + i * L->stride_x; /* ref :1831-1832 (origin 0, stride N) */
      const int yi = L->origin_y + j * L->stride_y;
      /* convertTo(CV_32FC1): exact integer values 0..255 (ref :1805-1806) */
      for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
          uint8_t c = cur[(size_t)(yi + y) * pitch + xi + x];
          uint8_t p = prev[(size_t)(yi + y) * pitch + xi + x];
          af[(size_t)y * n + x] = (float)c;
          af[nn + (size_t)y * n + x] = (float)p;
          ad[(size_t)y * n + x] = (double)c;
          ad[nn + (size_t)y * n + x] = (double)p;
        }
      double pc[2];
      oracle_pc_diag* d = diag ? &diag[i + j * L->grid_x] : NULL;
      int rc = (precision == 32) ? oracle_phase_correlate_f32(af, (size_t)n, af + nn, (size_t)n, n, pc, d, NULL)
                                 : oracle_phase_correlate_f64(ad, (size_t)n, ad + nn, (size_t)n, n, pc, d, NULL);
      if (rc) { free(af); free(ad); return rc; }
      /* shift = -cv::phaseCorrelate(cur, prev)  (ref :1836) */
      double sx = -pc[0], sy = -pc[1];
      /* gate (ref :1840-1856): against samplePointSize / 2 -- the UNPADDED size */
      int valid = 1;
      if (sx * sx + sy * sy > max_sq || fabs(sx) > (double)n / 2 || fabs(sy) > (double)n / 2) valid = 0;
      if (isnan(sx) || isnan(sy)) valid = 0;
      if (!valid) {
        sx = NAN;
        sy = NAN;
        ++invalid;
      }
      out_xy[2 * (i + j * L->grid_x) + 0] = sx; /* ref :1855 index i + j*sqNum */
      out_xy[2 * (i + j * L->grid_x) + 1] = sy;
    }
  if (n_invalid) *n_invalid = invalid;
  free(af);
  free(ad);
  return 0;
}

int oracle_fft_process_ocl_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_fft_layout* L,
                              int search_radius, int precision, double* out_xy, int* n_invalid, oracle_pc_diag* diag) {
  if (!cur || !prev || !L || !out_xy) return -1;
  const int n = L->patch;
  if (n < 8 || (n & 1) || largest_prime_factor(n) > 61) return -1;
  if (L->grid_x < 1 || L->grid_y < 1 || L->origin_x < 0 || L->origin_y < 0) return -1;
  if (L->origin_x + (L->grid_x - 1) * L->stride_x + n > L->width) return -1;
  if (L->origin_y + (L->grid_y - 1) * L->stride_y + n > L->height) return -1;
  if (precision != 32 && precision != 64) return -1;
  const size_t nn = (size_t)n * n;
  float* af = (float*)malloc(sizeof(float) * nn * 2);
  double* ad = (double*)malloc(sizeof(double) * nn * 2);
  if (!af || !ad) { free(af); free(ad); return -2; }
  int invalid = 0;
  const double max_sq = L->max_px_speed * L->max_px_speed;
  for (int j = 0; j < L->grid_y; ++j)
    for (int i = 0; i < L->grid_x; ++i) {
      const int xi = L->origin_x + i * L->stride_x, yi = L->origin_y + j * L->stride_y;
      for (int y = 0; y < n; ++y)
        for (int x = 0; x < n; ++x) {
          uint8_t c = cur[(size_t)(yi + y) * pitch + xi + x];
          uint8_t p = prev[(size_t)(yi + y) * pitch + xi + x];
          af[(size_t)y * n + x] = (float)c;
          af[nn + (size_t)y * n + x] = (float)p;
          ad[(size_t)y * n + x] = (double)c;
          ad[nn + (size_t)y * n + x] = (double)p;
        }
      double s[2];
      oracle_pc_diag* d = diag ? &diag[i + j * L->grid_x] : NULL;
      int rc = (precision == 32)
                   ? oracle_phase_correlate_ocl_f32(af, (size_t)n, af + nn, (size_t)n, n, xi, yi, search_radius, s, d, NULL)
                   : oracle_phase_correlate_ocl_f64(ad, (size_t)n, ad + nn, (size_t)n, n, xi, yi, search_radius, s, d, NULL);
      if (rc) { free(af); free(ad); return rc; }
      /* shift = speeds[i + sqNum*j], not negated (ref :1833); same gate (ref :1840-1856) */
      double sx = s[0], sy = s[1];
      int valid = 1;
      if (sx * sx + sy * sy > max_sq || fabs(sx) > (double)n / 2 || fabs(sy) > (double)n / 2) valid = 0;
      if (isnan(sx) || isnan(sy)) valid = 0;
      if (!valid) {
        sx = NAN;
        sy = NAN;
        ++invalid;
      }
      out_xy[2 * (i + j * L->grid_x) + 0] = sx;
      out_xy[2 * (i + j * L->grid_x) + 1] = sy;
    }
  if (n_invalid) *n_invalid = invalid;
  free(af);
  free(ad);
  return 0;
}

int oracle_resize_quarter_u8(const uint8_t* src, size_t pitch, int w, int h, uint8_t* dst) {
  if (!src || !dst || w < 4 || h < 4 || (w & 3) || (h & 3)) return -1;
  const int dw = w / 4;
  for (int y = 0; y < h / 4; ++y)
    for (int x = 0; x < dw; ++x) {
      const uint8_t* r1 = src + (size_t)(4 * y + 1) * pitch + 4 * x;
      const uint8_t* r2 = r1 + pitch;
      dst[(size_t)y * dw + x] = (uint8_t)((r1[1] + r1[2] + r2[1] + r2[2] + 2) >> 2);
    }
  return 0;
}

int oracle_fft_process_long_range_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_fft_layout* L,
                                     int precision, double* out_xy, int* n_invalid) {
  if (!cur || !prev || !L || !out_xy) return -1;
  if (L->origin_x || L->origin_y || L->stride_x != L->patch || L->stride_y != L->patch) return -1;
  if ((L->width & 3) || (L->height & 3) || L->grid_x < 4 || L->grid_y < 4) return -1;
  const int w = L->width / 4, h = L->height / 4;
  uint8_t* c = (uint8_t*)malloc((size_t)w * h);
  uint8_t* p = (uint8_t*)malloc((size_t)w * h);
  if (!c || !p) { free(c); free(p); return -2; }
  oracle_resize_quarter_u8(cur, pitch, L->width, L->height, c);   /* ref :1931 */
  oracle_resize_quarter_u8(prev, pitch, L->width, L->height, p);  /* ref :1932 */
  oracle_fft_layout lr = *L;
  lr.width = w;
  lr.height = h;
  lr.grid_x = L->grid_x / 4; /* sqNum_lr = sqNum / LONG_RANGE_RATIO, ref :1720 */
  lr.grid_y = L->grid_y / 4;
  /* ref include/FftMethod.h:393 `int max_px_speed_lr, max_px_speed_sq_lr`; src/FftMethod.cpp:1687-1688: the speed is
   * truncated to int, its square is what the gate compares with (:1963). oracle_fft_process_u8 squares
   * max_px_speed itself, and the square of the truncated integer is exact in double. */
  lr.max_px_speed = (double)(int)L->max_px_speed;
  int rc = oracle_fft_process_u8(c, p, (size_t)w, &lr, precision, out_xy, n_invalid, NULL);
  free(c);
  free(p);
  return rc;
}

int oracle_rgb2gray_u8(const uint8_t* src, size_t pitch_bytes, int w, int h, uint8_t* dst) {
  if (!src || !dst || w < 1 || h < 1) return -1;
  for (int y = 0; y < h; ++y)
    for (int x = 0; x < w; ++x) {
      const uint8_t* p = src + (size_t)y * pitch_bytes + 3 * (size_t)x;
      dst[(size_t)y * w + x] = (uint8_t)((p[0] * 4899 + p[1] * 9617 + p[2] * 1868 + (1 << 13)) >> 14);
    }
  return 0;
}

const char* oracle_version(void) { return "mof-oracle 0.1 (parity unpinned: no reference fixtures, OpenCV absent)"; }
