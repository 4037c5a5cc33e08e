/*
 * lp_ref.c -- CPU restatement of cv::logPolar on CV_8UC1 as scaleRotationEstimator uses it
 * (/root/reference/src/scaleRotationEstimator.cpp:45 INTER_CUBIC on the first frame, :112 INTER_LANCZOS4
 * afterwards; destination `tempIm` is a persistent member, :27, so pixels the map sends outside the source keep
 * their previous content -- cv::logPolar calls remap with BORDER_TRANSPARENT).
 *
 * TEST INFRASTRUCTURE ONLY. PARITY UNPINNED, twice over: cv::logPolar / cv::remap live in OpenCV (absent here,
 * version unpinned), and the reference ships no fixtures. What is restated is OpenCV's published algorithm as
 * recalled, in the TWO forms the reference compiles against (scaleRotationEstimator.cpp:41-46, :107-113):
 *   variant 0, ROS Noetic / OpenCV 4.2 `cv::logPolar`: a wrapper of cv::warpPolar(src, dst, src.size(), center,
 *     maxRadius = exp(width / M), flags | WARP_POLAR_LOG): Kmag = log(maxRadius) / width, a FLOAT table
 *     rhos[rho] = (float)(exp(rho * Kmag) - 1.0), Kangle = 2 pi / height, x = rhos[rho] * cos(Kangle * phi) + cx
 *     evaluated in double and stored as float;
 *   variant 1, ROS Melodic / OpenCV 3.2 `cvLogPolar` (C API): a DOUBLE table exp_tab[rho] = exp(rho / M) -- no "- 1" --
 *     and x = exp_tab[rho] * cos(phi * 2 pi / height) + cx, stored as float.
 * Both hand the maps to remap with BORDER_TRANSPARENT (neither call passes WARP_FILL_OUTLIERS). tools/opencv_ab/ dumps
 * the real maps and images on a machine that has OpenCV so that these recollections can be pinned.
 * Rows = phi over 2 pi, cols = rho; remap's fixed-point path for 8-bit images: map coordinates rounded to 1/32 px
 * (INTER_BITS = 5), separable kernel tables (cubic A = -0.75; Lanczos4) expanded to 2-D, scaled to 2^15
 * (INTER_REMAP_COEF_BITS) and corrected to sum exactly 2^15, pixel = (sum + 2^14) >> 15 saturated; footprints
 * crossing the border use BORDER_REFLECT_101 taps, anchors outside the image are skipped.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define TAB 32            /* INTER_TAB_SIZE */
#define COEF_BITS 15      /* INTER_REMAP_COEF_BITS */
#define COEF_SCALE (1 << COEF_BITS)

static void cubic_coeffs(float x, float* c) {
  const float A = -0.75f;
  c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
  c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
  c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
  c[3] = 1.f - c[0] - c[1] - c[2];
}

static void lanczos4_coeffs(float x, float* c) {
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; ++i) c[i] = 0;
    c[3] = 1;
    return;
  }
  float sum = 0;
  const double y0 = -(x + 3) * 3.14159265358979323846 * 0.25, s0 = sin(y0), c0 = cos(y0);
  for (int i = 0; i < 8; ++i) {
    const double y = -(x + 3 - i) * 3.14159265358979323846 * 0.25;
    c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += c[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; ++i) c[i] *= sum;
}

static short sat_short_round(float v) {
  long r = lrintf(v); /* cvRound: nearest, ties to even */
  if (r > 32767) r = 32767;
  if (r < -32768) r = -32768;
  return (short)r;
}

/* 2-D fixed-point table [fy][fx][ksize*ksize], sums corrected to COEF_SCALE */
static short* build_table(int ksize) {
  float tab1[TAB * 8];
  for (int i = 0; i < TAB; ++i) {
    if (ksize == 4) cubic_coeffs((float)i * (1.f / TAB), tab1 + i * 4);
    else lanczos4_coeffs((float)i * (1.f / TAB), tab1 + i * 8);
  }
  short* itab = (short*)malloc(sizeof(short) * TAB * TAB * ksize * ksize);
  if (!itab) return NULL;
  short* t = itab;
  for (int i = 0; i < TAB; ++i)
    for (int j = 0; j < TAB; ++j, t += ksize * ksize) {
      int isum = 0;
      for (int k1 = 0; k1 < ksize; ++k1) {
        const float vy = tab1[i * ksize + k1];
        for (int k2 = 0; k2 < ksize; ++k2) {
          const float v = vy * tab1[j * ksize + k2];
          isum += t[k1 * ksize + k2] = sat_short_round(v * COEF_SCALE);
        }
      }
      if (isum != COEF_SCALE) {
        const int diff = isum - COEF_SCALE, k2h = ksize / 2;
        int Mk1 = k2h, Mk2 = k2h, mk1 = k2h, mk2 = k2h;
        for (int k1 = k2h; k1 < k2h + 2; ++k1)
          for (int k2 = k2h; k2 < k2h + 2; ++k2) {
            if (t[k1 * ksize + k2] < t[mk1 * ksize + mk2]) mk1 = k1, mk2 = k2;
            else if (t[k1 * ksize + k2] > t[Mk1 * ksize + Mk2]) Mk1 = k1, Mk2 = k2;
          }
        if (diff < 0) t[Mk1 * ksize + Mk2] = (short)(t[Mk1 * ksize + Mk2] - diff);
        else t[mk1 * ksize + mk2] = (short)(t[mk1 * ksize + mk2] - diff);
      }
    }
  return itab;
}

static int reflect101(int p, int len) {
  if (len == 1) return 0;
  while (p < 0 || p >= len) {
    if (p < 0) p = -p;
    else p = 2 * len - 2 - p;
  }
  return p;
}

/* The float maps of cv::logPolar for a res x res image centred at (res/2, res/2); mapx/mapy: res*res floats, [phi][rho]. */
int oracle_logpolar_maps(int res, double M, int variant, float* mapx, float* mapy) {
  if (!mapx || !mapy || res < 8 || !(M > 0) || (variant != 0 && variant != 1)) return -1;
  const float cx = (float)(res / 2), cy = (float)(res / 2); /* cv::Point2f(resolution / 2, resolution / 2), :25 */
  const double PI = 3.14159265358979323846;
  if (variant == 0) {
    /* cv::logPolar -> cv::warpPolar (OpenCV >= 3.4.2, 4.x) */
    float* rhos = (float*)malloc(sizeof(float) * (size_t)res);
    if (!rhos) return -2;
    const double maxRadius = exp((double)res / M);
    const double Kmag = log(maxRadius) / (double)res;
    const double Kangle = 2.0 * PI / (double)res;
    for (int rho = 0; rho < res; ++rho) rhos[rho] = (float)(exp(rho * Kmag) - 1.0);
    for (int phi = 0; phi < res; ++phi) {
      const double KKy = Kangle * phi, cp = cos(KKy), sp = sin(KKy);
      for (int rho = 0; rho < res; ++rho) {
        mapx[(size_t)phi * res + rho] = (float)(rhos[rho] * cp + cx);
        mapy[(size_t)phi * res + rho] = (float)(rhos[rho] * sp + cy);
      }
    }
    free(rhos);
  } else {
    /* cvLogPolar (OpenCV 2.4 ... 3.2 C API) */
    double* exp_tab = (double*)malloc(sizeof(double) * (size_t)res);
    if (!exp_tab) return -2;
    for (int rho = 0; rho < res; ++rho) exp_tab[rho] = exp(rho / M);
    for (int phi = 0; phi < res; ++phi) {
      const double cp = cos(phi * 2 * PI / res), sp = sin(phi * 2 * PI / res);
      for (int rho = 0; rho < res; ++rho) {
        const double r = exp_tab[rho];
        mapx[(size_t)phi * res + rho] = (float)(r * cp + cx);
        mapy[(size_t)phi * res + rho] = (float)(r * sp + cy);
      }
    }
    free(exp_tab);
  }
  return 0;
}

int oracle_logpolar_u8(const uint8_t* src, size_t pitch, int res, double M, int interp, uint8_t* dst) {
  return oracle_logpolar_variant_u8(src, pitch, res, M, interp, 0, dst);
}

int oracle_logpolar_variant_u8(const uint8_t* src, size_t pitch, int res, double M, int interp, int variant, uint8_t* dst) {
  if (!src || !dst || res < 8 || !(M > 0) || (interp != 2 && interp != 4) || (variant != 0 && variant != 1)) return -1;
  const int ksize = interp == 2 ? 4 : 8, half = ksize / 2 - 1; /* taps start at anchor - half */
  short* itab = build_table(ksize);
  float* mapx = (float*)malloc(sizeof(float) * (size_t)res * res);
  float* mapy = (float*)malloc(sizeof(float) * (size_t)res * res);
  if (!itab || !mapx || !mapy || oracle_logpolar_maps(res, M, variant, mapx, mapy)) { free(itab); free(mapx); free(mapy); return -2; }
  for (int phi = 0; phi < res; ++phi) {
    for (int rho = 0; rho < res; ++rho) {
      const float mx = mapx[(size_t)phi * res + rho], my = mapy[(size_t)phi * res + rho];
      /* remap: fixed-point coordinates, 1/32 px. Far-away coordinates saturate like saturate_cast<short>. */
      const float fx = mx * (float)TAB, fy = my * (float)TAB;
      if (!(fabsf(fx) < 1.0e9f) || !(fabsf(fy) < 1.0e9f)) continue; /* certainly outside */
      const long isx = lrintf(fx), isy = lrintf(fy);
      long ax = isx >> 5, ay = isy >> 5;
      if (ax > 32767) ax = 32767;
      if (ax < -32768) ax = -32768;
      if (ay > 32767) ay = 32767;
      if (ay < -32768) ay = -32768;
      const int widx = (int)(isy & (TAB - 1)) * TAB + (int)(isx & (TAB - 1));
      /* BORDER_TRANSPARENT: anchor outside -> destination untouched */
      if ((unsigned long)ax >= (unsigned long)res || (unsigned long)ay >= (unsigned long)res) continue;
      const short* w = itab + (size_t)widx * ksize * ksize;
      const int sx = (int)ax - half, sy = (int)ay - half;
      int sum = 0;
      for (int k1 = 0; k1 < ksize; ++k1) {
        const int yy = reflect101(sy + k1, res); /* BORDER_REFLECT_101 for footprints crossing the border */
        for (int k2 = 0; k2 < ksize; ++k2) {
          const int xx = reflect101(sx + k2, res);
          sum += (int)src[(size_t)yy * pitch + xx] * (int)w[k1 * ksize + k2];
        }
      }
      int v = (sum + (1 << (COEF_BITS - 1))) >> COEF_BITS;
      if (v < 0) v = 0;
      if (v > 255) v = 255;
      dst[(size_t)phi * res + rho] = (uint8_t)v;
    }
  }
  free(itab);
  free(mapx);
  free(mapy);
  return 0;
}

int oracle_scale_rotation_step(const uint8_t* frame, size_t pitch, int res, double M, int first, uint8_t* temp_im,
                               float* prev_lp, int precision, double* out, double* pt_xy) {
  return oracle_scale_rotation_step_variant(frame, pitch, res, M, first, temp_im, prev_lp, precision, 0, out, pt_xy);
}

int oracle_scale_rotation_step_variant(const uint8_t* frame, size_t pitch, int res, double M, int first, uint8_t* temp_im,
                                       float* prev_lp, int precision, int variant, double* out, double* pt_xy) {
  if (!frame || !temp_im || !prev_lp || !out || res < 8 || (res & 1)) return -1;
  const size_t nn = (size_t)res * res;
  if (first) {
    int rc = oracle_logpolar_variant_u8(frame, pitch, res, M, 2, variant, temp_im); /* INTER_CUBIC, :45 (:44 on Melodic) */
    if (rc) return rc;
    for (size_t i = 0; i < nn; ++i) prev_lp[i] = (float)temp_im[i]; /* convertTo CV_32FC1, :48 */
    out[0] = 1.0;
    out[1] = 0.0; /* :74 */
    if (pt_xy) pt_xy[0] = pt_xy[1] = 0.0;
    return 0;
  }
  int rc = oracle_logpolar_variant_u8(frame, pitch, res, M, 4, variant, temp_im); /* INTER_LANCZOS4, :112 (:110 on Melodic) */
  if (rc) return rc;
  float* cur = (float*)malloc(sizeof(float) * nn);
  if (!cur) return -2;
  for (size_t i = 0; i < nn; ++i) cur[i] = (float)temp_im[i]; /* :115 */
  double pt[2];
  if (precision == 64) {
    double* a = (double*)malloc(sizeof(double) * nn * 2);
    if (!a) { free(cur); return -2; }
    for (size_t i = 0; i < nn; ++i) { a[i] = cur[i]; a[nn + i] = prev_lp[i]; }
    rc = oracle_phase_correlate_f64(a, (size_t)res, a + nn, (size_t)res, res, pt, NULL, NULL);
    free(a);
  } else {
    rc = oracle_phase_correlate_f32(cur, (size_t)res, prev_lp, (size_t)res, res, pt, NULL, NULL); /* :117 */
  }
  if (rc) { free(cur); return rc; }
  if (pt_xy) { pt_xy[0] = pt[0]; pt_xy[1] = pt[1]; }
  if (fabs(pt[0]) > res / 2 || fabs(pt[0]) > res / 2) { /* :119 (the second clause repeats pt.x) */
    out[0] = 1.0;
    out[1] = 0.0;
    free(cur);
    return 0; /* early return: prevIm_F32 is NOT updated (:120) */
  }
  const double Ky = (double)res / 360.0; /* :26 */
  out[0] = exp(pt[0] / M);                               /* :123 */
  out[1] = (pt[1] / Ky) * (3.14159265358979323846 / 180); /* :124 */
  memcpy(prev_lp, cur, sizeof(float) * nn);               /* :128 */
  free(cur);
  return 0;
}
