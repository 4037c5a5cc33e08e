/* Drives the whole C surface of the CPU oracle under AddressSanitizer + UndefinedBehaviorSanitizer (tests/san/Makefile,
 * tests/test_sanitizers.py). Inputs are small and chosen to touch the edges: patches flush with the frame border,
 * block windows that end on the last row/column, log-polar footprints that cross the border, empty/invalid geometry.
 * Exit code 0 and no sanitizer report = pass. Values are checked elsewhere (tests/test_oracle_*.py). */
#include <math.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

int pcfast_fft_process_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_fft_layout* L, double* out_xy);

static uint32_t rng_state = 12345u;
static uint32_t rnd(void) { rng_state = rng_state * 1664525u + 1013904223u; return rng_state >> 8; }

#define CHECK(cond) do { if (!(cond)) { fprintf(stderr, "check failed: %s (line %d)\n", #cond, __LINE__); return 1; } } while (0)

int main(void) {
  /* ---- FFT path: every supported patch size, patches flush with the right/bottom border, odd pitch ---- */
  const int sizes[4] = {32, 64, 120, 128};
  for (int s = 0; s < 4; ++s) {
    const int n = sizes[s], w = 2 * n + 5, h = n + 3, pitch = w + 7;
    uint8_t* cur = (uint8_t*)malloc((size_t)pitch * h);
    uint8_t* prev = (uint8_t*)malloc((size_t)pitch * h);
    for (int i = 0; i < pitch * h; ++i) { cur[i] = (uint8_t)rnd(); prev[i] = (uint8_t)rnd(); }
    oracle_fft_layout L = {w, h, n, 2, 1, 5, 3, n, 1, 80.0}; /* second patch ends exactly at x = w, y = h */
    double out[4];
    int ninv = 0;
    oracle_pc_diag diag[2];
    for (int prec = 32; prec <= 64; prec += 32) {
      CHECK(oracle_fft_process_u8(cur, prev, (size_t)pitch, &L, prec, out, &ninv, diag) == 0);
      CHECK(oracle_fft_process_ocl_u8(cur, prev, (size_t)pitch, &L, 55, prec, out, &ninv, diag) == 0);
    }
    { /* the tuned bench-only path (oracle/pc_fast.c): power-of-two patches, must decline the rest */
      double fast[4];
      const int rc = pcfast_fft_process_u8(cur, prev, (size_t)pitch, &L, fast);
      CHECK((n & (n - 1)) ? rc != 0 : rc == 0);
    }
    L.grid_x = 3; /* leaves the frame: must be refused, not read */
    CHECK(oracle_fft_process_u8(cur, prev, (size_t)pitch, &L, 32, out, &ninv, NULL) != 0);
    CHECK(pcfast_fft_process_u8(cur, prev, (size_t)pitch, &L, out) != 0);
    free(cur);
    free(prev);
  }
  { /* long-range mode + quarter resize + BGR front end */
    const int fs = 512;
    uint8_t* cur = (uint8_t*)malloc((size_t)fs * fs * 3);
    uint8_t* prev = (uint8_t*)malloc((size_t)fs * fs);
    uint8_t* q = (uint8_t*)malloc((size_t)(fs / 4) * (fs / 4));
    uint8_t* g = (uint8_t*)malloc((size_t)fs * fs);
    for (int i = 0; i < fs * fs * 3; ++i) cur[i] = (uint8_t)rnd();
    for (int i = 0; i < fs * fs; ++i) prev[i] = (uint8_t)rnd();
    oracle_fft_layout L = {fs, fs, 64, 8, 8, 0, 0, 64, 64, 10.9};
    double out[8];
    int ninv;
    CHECK(oracle_fft_process_long_range_u8(cur, prev, (size_t)fs, &L, 32, out, &ninv) == 0);
    CHECK(oracle_resize_quarter_u8(prev, (size_t)fs, fs, fs, q) == 0);
    CHECK(oracle_rgb2gray_u8(cur, (size_t)fs * 3, fs, fs, g) == 0);
    CHECK(oracle_resize_quarter_u8(prev, (size_t)fs, 6, 8, q) != 0); /* not a multiple of 4 */
    free(cur); free(prev); free(q); free(g);
  }
  /* ---- block matching: both geometries, windows ending on the last row/column, refine ---- */
  {
    const int w = 2 * (16 + 8) + 2 * 5, h = 16 + 8 + 2 * 5; /* two spaced blocks wide, one high (grid = (w - 2r) / S) */
    uint8_t* cur = (uint8_t*)malloc((size_t)w * h);
    uint8_t* prev = (uint8_t*)malloc((size_t)w * h);
    for (int i = 0; i < w * h; ++i) { cur[i] = (uint8_t)rnd(); prev[i] = (uint8_t)rnd(); }
    oracle_bm_config c;
    oracle_bm_config_fast_spaced(&c, w, h, 16, 8, 5);
    CHECK(c.grid_x == 2 && c.grid_y == 1);
    int8_t dx[2], dy[2], mode[2], top[3];
    int32_t smin[2], sall[2 * 11 * 11];
    CHECK(oracle_bm_process_u8(cur, prev, (size_t)w, &c, dx, dy, mode, smin, sall) == 0);
    CHECK(oracle_bm_histogram_top(dx, 2, 5, 3, top) == 0);
    free(cur); free(prev);
    const int fs = 40;
    cur = (uint8_t*)malloc((size_t)fs * fs);
    prev = (uint8_t*)malloc((size_t)fs * fs);
    for (int i = 0; i < fs * fs; ++i) { cur[i] = (uint8_t)rnd(); prev[i] = (uint8_t)rnd(); }
    oracle_bm_config_block_method(&c, fs, 8, 4);
    int8_t bx[16], by[16];
    CHECK(c.grid_x == 4);
    CHECK(oracle_bm_process_u8(cur, prev, (size_t)fs, &c, bx, by, mode, NULL, NULL) == 0);
    uint8_t* up = (uint8_t*)malloc((size_t)4 * fs * fs);
    CHECK(oracle_resize_2x_u8(cur, (size_t)fs, fs, fs, up) == 0);
    double r[2];
    int32_t sads[18];
    for (int f = 0; f < 2; ++f)
      for (int ox = -3; ox <= 3; ox += 3) CHECK(oracle_bm_refine_u8(cur, prev, (size_t)fs, fs, fs, ox, -ox, 2, f, r, sads) == 0);
    CHECK(oracle_bm_refine_u8(cur, prev, (size_t)fs, fs, fs, 40, 0, 2, 1, r, sads) != 0); /* empty cut-out */
    free(cur); free(prev); free(up);
  }
  /* ---- log-polar (both OpenCV generations, both interpolations) + one estimator step pair ---- */
  {
    const int res = 64;
    uint8_t* img = (uint8_t*)malloc((size_t)res * res);
    uint8_t* dst = (uint8_t*)calloc((size_t)res * res, 1);
    float* prev_lp = (float*)calloc((size_t)res * res, sizeof(float));
    float* mx = (float*)malloc(sizeof(float) * res * res);
    float* my = (float*)malloc(sizeof(float) * res * res);
    for (int i = 0; i < res * res; ++i) img[i] = (uint8_t)rnd();
    for (int variant = 0; variant < 2; ++variant) {
      CHECK(oracle_logpolar_maps(res, 12.0, variant, mx, my) == 0);
      for (int interp = 2; interp <= 4; interp += 2) CHECK(oracle_logpolar_variant_u8(img, (size_t)res, res, 12.0, interp, variant, dst) == 0);
      double out[2], pt[2];
      CHECK(oracle_scale_rotation_step_variant(img, (size_t)res, res, 12.0, 1, dst, prev_lp, 32, variant, out, pt) == 0);
      CHECK(oracle_scale_rotation_step_variant(img, (size_t)res, res, 12.0, 0, dst, prev_lp, 64, variant, out, pt) == 0);
    }
    CHECK(oracle_logpolar_u8(img, (size_t)res, res, 12.0, 3, dst) != 0);
    free(img); free(dst); free(prev_lp); free(mx); free(my);
  }
  /* ---- geometry tail ---- */
  {
    const oracle_camera cam = {340, 338.5, 376, 240, -0.28, 0.07, 0.0004, -0.0003, -0.006};
    const oracle_geom_layout L = {4, 4, 0, 0, 120, 120, 120};
    double shifts[32], out[7], H[9], o6[6];
    uint8_t mask[16];
    for (int i = 0; i < 16; ++i) { shifts[2 * i] = 3.0 + 0.01 * (i % 4); shifts[2 * i + 1] = -2.0 + 0.02 * (i / 4); }
    shifts[10] = NAN;
    shifts[14] = 55.0; /* an outlier */
    oracle_rt_params p = {2.5, 0.02, 136.0, {0, 0, 0, 1}, {0, 0, 0, 1}, {0, 0, 0}};
    oracle_quat_from_rpy(0.01, -0.02, 0.3, p.ang_rate_q);
    for (int thr = -1; thr <= 17; thr += 3) (void)oracle_get_rt(shifts, &L, &cam, &p, thr, out, mask, H);
    (void)oracle_get_rt(shifts, &L, &cam, &p, 8, out, NULL, NULL);
    p.dt = 0;
    CHECK(oracle_get_rt(shifts, &L, &cam, &p, 8, out, mask, H) == 1);
    for (int i = 0; i < 32; ++i) shifts[i] = NAN;
    p.dt = 0.02;
    CHECK(oracle_get_rt(shifts, &L, &cam, &p, 0, out, mask, H) != 0);
    const oracle_2dt_params p2 = {2.0, 0.02, 0.1, -0.2, 1.0};
    CHECK(oracle_get_2dt(shifts, &L, &cam, &p2, o6) == 2);
    shifts[30] = 1.0; shifts[31] = 2.0;
    CHECK(oracle_get_2dt(shifts, &L, &cam, &p2, o6) == 0);
    double R[36], t[12], nn[12];
    const double Hid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1}, Hz[9] = {0};
    CHECK(oracle_decompose_homography(Hid, R, t, nn) == 1);
    CHECK(oracle_decompose_homography(Hz, R, t, nn) == 0);
    double a[8] = {0, 0, 1, 0, 1, 1, 0, 1}, b[8] = {0.1, 0, 1.1, 0.05, 1.2, 1.1, 0, 0.9};
    CHECK(oracle_find_homography(a, b, 4, H, mask) == 1);
    CHECK(oracle_find_homography(a, b, 3, H, mask) == 0);
  }
  printf("oracle sanitizer driver: ok (%s)\n", oracle_version());
  return 0;
}
