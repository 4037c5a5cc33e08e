/*
 * pc_fast.c -- a TUNED CPU implementation of the useOCL=false FFT path, for bench.py's `cpu_baseline.tuned` leg only.
 *
 * TEST / BENCH INFRASTRUCTURE ONLY (see oracle.h): not the parity oracle and not part of the product. The oracle
 * (pc_ref.c) is written for fidelity -- a recursive generic-radix DFT with modulo twiddle indexing, every
 * cv::phaseCorrelate stage as its own pass over full CCS arrays -- and is ~10x slower than a real cv::phaseCorrelate,
 * so timing it beside the GPU says little. This file computes the SAME estimator
 * (/root/reference/src/FftMethod.cpp:1836: -cv::phaseCorrelate(cur(roi), prev(roi)); stages :1487-1498; the real-only
 * slot rule of magSpectrums :107-109 / divSpectrums :1127-1129; fftShift :1257-1323; first maximum; 5x5 weighted
 * centroid in double :1329-1385; gate :1838-1856) the way a fast CPU library would:
 *   - power-of-two patches only (32, 64, 128); iterative Stockham radix-4 (+ one radix-2) passes, twiddles precomputed
 *     in double and stored as float; every pass runs over a BATCH of independent lines laid out element-major
 *     ([element][line]), so its inner loops are unit-stride and auto-vectorise (AVX2 / AVX-512 with -march=native);
 *   - real-input transforms: two image rows ride one complex line, only the half spectrum u = 0..N/2 is carried
 *     through the column pass, the cross-power and the inverse; the inverse's last pass packs two real rows again;
 *   - no per-call allocation (thread-local work buffers), no CCS arrays.
 * It must agree with the f32 oracle within 1e-4 px on well-conditioned patches (tests/test_pc_fast.py, and bench.py
 * checks the sample it times). Build: gcc -O3 (-march=native when built on the box that runs it).
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define PF_MAXN 128
#define PF_PI 3.14159265358979323846

/* Twiddles W_n^{p}, W_n^{2p}, W_n^{3p} (p < n/4) of every radix-4 level n = 128, 64, .. 4, computed once in double and
 * stored as float (cos, +sin): tab[level][p][6]. Filled on first use; the fill is idempotent, so a race between threads
 * only repeats identical stores. */
static float pf_tw[8][PF_MAXN / 4][6];
static volatile int pf_tw_ready = 0;
static void pf_tw_init(void) {
  if (pf_tw_ready) return;
  for (int lv = 2; lv < 8; ++lv) {
    const int n = 1 << lv;
    for (int p = 0; p < n / 4; ++p) {
      const double a1 = 2.0 * PF_PI * (double)p / (double)n;
      pf_tw[lv][p][0] = (float)cos(a1);
      pf_tw[lv][p][1] = (float)sin(a1);
      pf_tw[lv][p][2] = (float)cos(2 * a1);
      pf_tw[lv][p][3] = (float)sin(2 * a1);
      pf_tw[lv][p][4] = (float)cos(3 * a1);
      pf_tw[lv][p][5] = (float)sin(3 * a1);
    }
  }
  __sync_synchronize();
  pf_tw_ready = 1;
}

/* One complex transform of length n over `lines` lines, element-major: x[(e) * lines + b]. sign = -1 forward,
 * +1 inverse (unscaled). Stockham autosort: ping-pongs between (xr, xi) and (yr, yi); returns 0 if the result is in x,
 * 1 if it is in y. */
static int pf_fft(int n, int lines, int sign, float* restrict xr, float* restrict xi, float* restrict yr, float* restrict yi) {
  int s = 1, flip = 0;
  const float sg = (float)sign;
  while (n >= 4) {
    const int n1 = n / 4;
    const size_t blk = (size_t)s * lines; /* contiguous run that shares (p) */
    const int lv = __builtin_ctz((unsigned)n);
    for (int p = 0; p < n1; ++p) {
      const float* t6 = pf_tw[lv][p];
      const float w1r = t6[0], w1i = sg * t6[1], w2r = t6[2], w2i = sg * t6[3], w3r = t6[4], w3i = sg * t6[5];
      const float *restrict ar = xr + (size_t)p * blk, *restrict ai = xi + (size_t)p * blk;
      const float *restrict br = ar + (size_t)n1 * blk, *restrict bi = ai + (size_t)n1 * blk;
      const float *restrict cr = br + (size_t)n1 * blk, *restrict ci = bi + (size_t)n1 * blk;
      const float *restrict dr = cr + (size_t)n1 * blk, *restrict di = ci + (size_t)n1 * blk;
      float *restrict o0r = yr + (size_t)(4 * p) * blk, *restrict o0i = yi + (size_t)(4 * p) * blk;
      float *restrict o1r = o0r + blk, *restrict o1i = o0i + blk, *restrict o2r = o1r + blk, *restrict o2i = o1i + blk;
      float *restrict o3r = o2r + blk, *restrict o3i = o2i + blk;
#pragma GCC ivdep /* input and output buffers never overlap (ping-pong) */
      for (size_t q = 0; q < blk; ++q) {
        const float apcr = ar[q] + cr[q], apci = ai[q] + ci[q], amcr = ar[q] - cr[q], amci = ai[q] - ci[q];
        const float bpdr = br[q] + dr[q], bpdi = bi[q] + di[q];
        /* j (b - d) with j = sign * i */
        const float jr = -sg * (bi[q] - di[q]), ji = sg * (br[q] - dr[q]);
        o0r[q] = apcr + bpdr;
        o0i[q] = apci + bpdi;
        const float t1r = amcr + jr, t1i = amci + ji;
        o1r[q] = t1r * w1r - t1i * w1i;
        o1i[q] = t1r * w1i + t1i * w1r;
        const float t2r = apcr - bpdr, t2i = apci - bpdi;
        o2r[q] = t2r * w2r - t2i * w2i;
        o2i[q] = t2r * w2i + t2i * w2r;
        const float t3r = amcr - jr, t3i = amci - ji;
        o3r[q] = t3r * w3r - t3i * w3i;
        o3i[q] = t3r * w3i + t3i * w3r;
      }
    }
    float* t;
    t = xr, xr = yr, yr = t;
    t = xi, xi = yi, yi = t;
    flip ^= 1;
    n /= 4;
    s *= 4;
  }
  if (n == 2) {
    const size_t blk = (size_t)s * lines;
#pragma GCC ivdep
    for (size_t q = 0; q < blk; ++q) {
      const float ar = xr[q], ai = xi[q], br = xr[q + blk], bi = xi[q + blk];
      yr[q] = ar + br;
      yi[q] = ai + bi;
      yr[q + blk] = ar - br;
      yi[q + blk] = ai - bi;
    }
    flip ^= 1;
  }
  return flip;
}

/* per-thread work space: every array is N x (N/2 + 1) floats at most, padded */
#define PF_H1 (PF_MAXN / 2 + 1)
typedef struct {
  float zr[PF_MAXN * PF_MAXN], zi[PF_MAXN * PF_MAXN], wr[PF_MAXN * PF_MAXN], wi[PF_MAXN * PF_MAXN];
  float Ar[PF_MAXN * PF_H1], Ai[PF_MAXN * PF_H1], Br[PF_MAXN * PF_H1], Bi[PF_MAXN * PF_H1];
  float surf[PF_MAXN * PF_MAXN];
} pf_work;
static __thread pf_work* pf_tls = NULL;

/* Half spectrum F[v][u], u = 0..H, of one n x n u8 patch -> (Fr, Fi) laid out [v][u] with row pitch H1 = H + 1. */
static void pf_forward(pf_work* w, const uint8_t* img, size_t pitch, int n, float* Fr, float* Fi) {
  const int H = n / 2, H1 = H + 1;
  /* rows 2j, 2j+1 as real / imaginary part of line j; element-major [x][j] */
  for (int j = 0; j < H; ++j) {
    const uint8_t *r0 = img + (size_t)(2 * j) * pitch, *r1 = r0 + pitch;
    for (int x = 0; x < n; ++x) {
      w->zr[(size_t)x * H + j] = (float)r0[x];
      w->zi[(size_t)x * H + j] = (float)r1[x];
    }
  }
  float *xr = w->zr, *xi = w->zi;
  if (pf_fft(n, H, -1, w->zr, w->zi, w->wr, w->wi)) xr = w->wr, xi = w->wi;
  float *cr = (xr == w->zr) ? w->wr : w->zr, *ci = (xi == w->zi) ? w->wi : w->zi;
  /* untangle into the rows' half spectra, stored for the column pass element-major over y with u as the line: [y][u] */
  for (int u = 0; u <= H; ++u) {
    const int um = (n - u) % n;
    const float *ar = xr + (size_t)u * H, *ai = xi + (size_t)u * H, *mr = xr + (size_t)um * H, *mi = xi + (size_t)um * H;
    for (int j = 0; j < H; ++j) {
      const float e_r = 0.5f * (ar[j] + mr[j]), e_i = 0.5f * (ai[j] - mi[j]);  /* row 2j   */
      const float o_r = 0.5f * (ai[j] + mi[j]), o_i = 0.5f * (mr[j] - ar[j]);  /* row 2j+1 */
      cr[(size_t)(2 * j) * H1 + u] = e_r;
      ci[(size_t)(2 * j) * H1 + u] = e_i;
      cr[(size_t)(2 * j + 1) * H1 + u] = o_r;
      ci[(size_t)(2 * j + 1) * H1 + u] = o_i;
    }
  }
  float *sr = (cr == w->zr) ? w->wr : w->zr, *si = (ci == w->zi) ? w->wi : w->zi;
  const int fl = pf_fft(n, H1, -1, cr, ci, sr, si);
  const float *rr = fl ? sr : cr, *ri = fl ? si : ci;
  memcpy(Fr, rr, sizeof(float) * (size_t)n * H1);
  memcpy(Fi, ri, sizeof(float) * (size_t)n * H1);
}

/* -cv::phaseCorrelate(cur, prev) on one patch pair + the gate; out_xy = (x, y) or (NaN, NaN). */
static void pf_patch(pf_work* w, const uint8_t* cur, const uint8_t* prev, size_t pitch, int n, double max_sq, double* out_xy) {
  const int H = n / 2, H1 = H + 1;
  pf_forward(w, cur, pitch, n, w->Ar, w->Ai);
  pf_forward(w, prev, pitch, n, w->Br, w->Bi);
  const float eps = FLT_EPSILON; /* :1117 */
  /* the four real-only slots (v, u) in {0, H}^2: magSpectrums stores the SQUARE there (:107-109) and divSpectrums divides
   * plainly (:1127-1129): C = P / (P^2 + eps), P = A B (both real) */
  float ro[4];
  {
    const int vs[2] = {0, H};
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const size_t i = (size_t)vs[a] * H1 + (size_t)vs[b];
        const float p = w->Ar[i] * w->Br[i];
        ro[2 * a + b] = p / (p * p + eps);
      }
  }
  /* C = P |P| / (|P|^2 + eps), P = A conj(B) (mulSpectrums :1494, magSpectrums :70-168, divSpectrums :1086-1251);
   * kept as conj(C): the unscaled inverse is then one more FORWARD transform */
  for (int v = 0; v < n; ++v) {
    float *ar = w->Ar + (size_t)v * H1, *ai = w->Ai + (size_t)v * H1;
    const float *br = w->Br + (size_t)v * H1, *bi = w->Bi + (size_t)v * H1;
    for (int u = 0; u <= H; ++u) {
      const float pr = ar[u] * br[u] + ai[u] * bi[u], pi = ai[u] * br[u] - ar[u] * bi[u];
      const float q = pr * pr + pi * pi;
      const float sc = sqrtf(q) / (q + eps);
      ar[u] = pr * sc;
      ai[u] = -(pi * sc);
    }
  }
  {
    const int vs[2] = {0, H};
    for (int a = 0; a < 2; ++a)
      for (int b = 0; b < 2; ++b) {
        const size_t i = (size_t)vs[a] * H1 + (size_t)vs[b];
        w->Ar[i] = ro[2 * a + b];
        w->Ai[i] = 0.f;
      }
  }
  /* columns (idft :1497, first half): lines = u, elements = v */
  const int fl = pf_fft(n, H1, -1, w->Ar, w->Ai, w->Br, w->Bi);
  const float *gr = fl ? w->Br : w->Ar, *gi = fl ? w->Bi : w->Ai; /* G[y][u] */
  /* rows: the surface is real, so rows y1 and y1 + H ride one complex line; elements = u, lines = y1 */
  for (int u = 0; u < n; ++u) {
    float *er = w->zr + (size_t)u * H, *ei = w->zi + (size_t)u * H;
    if (u <= H) {
      for (int p = 0; p < H; ++p) {
        const size_t i1 = (size_t)p * H1 + u, i2 = (size_t)(p + H) * H1 + u;
        er[p] = gr[i1] - gi[i2];
        ei[p] = gi[i1] + gr[i2];
      }
    } else {
      const int um = n - u;
      for (int p = 0; p < H; ++p) {
        const size_t i1 = (size_t)p * H1 + um, i2 = (size_t)(p + H) * H1 + um;
        er[p] = gr[i1] + gi[i2];
        ei[p] = gr[i2] - gi[i1];
      }
    }
  }
  const int fl2 = pf_fft(n, H, -1, w->zr, w->zi, w->wr, w->wi);
  const float *sr = fl2 ? w->wr : w->zr, *si = fl2 ? w->wi : w->zi; /* [x][y1]: re = s[y1][x], im = s[y1 + H][x] */
  /* fftShift (:1257-1323) + first maximum in row-major order (minMaxLoc) */
  float best = -INFINITY;
  int bx = 0, by = 0;
  for (int ys = 0; ys < n; ++ys) {
    const int y = (ys + H) % n;
    const float* src = y < H ? sr + y : si + (y - H);
    float* row = w->surf + (size_t)ys * n;
    for (int xs = 0; xs < n; ++xs) row[xs] = src[(size_t)((xs + H) % n) * H];
    for (int xs = 0; xs < n; ++xs)
      if (row[xs] > best) best = row[xs], bx = xs, by = ys;
  }
  /* weightedCentroid over the clamped 5 x 5 window, every value, in double (:1337-1383) */
  int x0 = bx - 2, x1 = bx + 2, y0 = by - 2, y1 = by + 2;
  x0 = x0 < 0 ? 0 : x0;
  y0 = y0 < 0 ? 0 : y0;
  x1 = x1 > n - 1 ? n - 1 : x1;
  y1 = y1 > n - 1 ? n - 1 : y1;
  double cx = 0, cy = 0, sum = 0;
  for (int y = y0; y <= y1; ++y)
    for (int x = x0; x <= x1; ++x) {
      const double c = (double)w->surf[(size_t)y * n + x];
      cx += (double)x * c;
      cy += (double)y * c;
      sum += c;
    }
  sum += DBL_EPSILON; /* :1378 */
  double sx = cx / sum - (double)n / 2.0, sy = cy / sum - (double)n / 2.0; /* -(center - t), :1836 */
  if (sx * sx + sy * sy > max_sq || fabs(sx) > (double)n / 2.0 || fabs(sy) > (double)n / 2.0 || sx != sx || sy != sy)
    sx = sy = NAN; /* :1838-1856 */
  out_xy[0] = sx;
  out_xy[1] = sy;
}

/* FftMethod::processImage, useOCL=false, on one frame pair: same layout / outputs as oracle_fft_process_u8. Power-of-two
 * patches up to 128 only (returns -1 otherwise). Thread-safe (thread-local work space). */
int pcfast_fft_process_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_fft_layout* L, double* out_xy) {
  if (!cur || !prev || !L || !out_xy) return -1;
  const int n = L->patch;
  if (n < 8 || n > PF_MAXN || (n & (n - 1))) return -1;
  if (L->grid_x < 1 || L->grid_y < 1 || L->origin_x < 0 || L->origin_y < 0) return -1;
  if (L->origin_x + (L->grid_x - 1) * L->stride_x + n > L->width) return -1;
  if (L->origin_y + (L->grid_y - 1) * L->stride_y + n > L->height) return -1;
  pf_tw_init();
  if (!pf_tls) {
    pf_tls = (pf_work*)malloc(sizeof(pf_work));
    if (!pf_tls) return -2;
  }
  const double max_sq = L->max_px_speed * L->max_px_speed;
  for (int j = 0; j < L->grid_y; ++j)
    for (int i = 0; i < L->grid_x; ++i) {
      const size_t off = (size_t)(L->origin_y + j * L->stride_y) * pitch + (size_t)(L->origin_x + i * L->stride_x);
      pf_patch(pf_tls, cur + off, prev + off, pitch, n, max_sq, out_xy + 2 * ((size_t)j * L->grid_x + i));
    }
  return 0;
}
