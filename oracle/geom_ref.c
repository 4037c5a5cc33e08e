/*
 * geom_ref.c -- CPU restatement of the geometry tail of the hot path: OpticFlow::get2DT
 * (/root/reference/src/optic_flow.cpp:388-510) and OpticFlow::getRT (:515-774).
 *
 * TEST INFRASTRUCTURE ONLY (see oracle.h). PARITY UNPINNED: the arithmetic getRT delegates to OpenCV calib3d
 * (cv::undistortPoints :549-550, cv::findHomography :559, cv::decomposeHomographyMat :595) and tf2 (:599-748) is not in
 * the reference tree and neither library is installed; what follows restates their published algorithms
 * (OpenCV 4.2 undistort.dispatch.cpp / fundam.cpp / homography_decomp.cpp; tf2 LinearMath Quaternion.h, Matrix3x3.h,
 * Transform.h) from memory. get2DT itself is closed-form text of the reference and needs none of them except tf2's
 * `Vector3 / s == Vector3 * (1 / s)`.
 *
 * cv::findHomography's RANSAC draws its minimal sets from cv::RNG(-1); that sequence is not restated. The sampler
 * below is this project's own (documented in mrs_optic_flow_amd/csrc/geom_core.hpp and repeated here): hypothesis k
 * draws from splitmix64 seeded with (SEED ^ k * 0xD1342543DE82EF95), indices = (state >> 11) % n with duplicates
 * redrawn, up to 10 attempts per hypothesis to find a non-degenerate, orientation-consistent set. Everything else
 * follows OpenCV's structure: forward reprojection error <= threshold^2, best = strictly more inliers,
 * RANSACUpdateNumIters at confidence 0.995 from 2000, normalised DLT on the consensus set, then 10 LM steps.
 *
 * Written independently of geom_core.hpp (plain C, own helpers) but to the same specification and operation order,
 * so that on one host the two agree to the last bit wherever only +, -, *, /, sqrt are involved.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define G_PI 3.14159265358979323846
#define RANSAC_THR 0.01      /* optic_flow.cpp:559 */
#define RANSAC_CONF 0.995    /* cv::findHomography default */
#define RANSAC_ITERS 2000    /* cv::findHomography default */
#define RANSAC_SEED 0x5EED0F10C0FFEEull

static int is_fin(double v) { return isfinite(v); }

/* ---------------------------------------------------------------------------------------------------------------- */
/* cv::undistortPoints (no R, no P): pixel -> normalised, 5 fixed-point iterations                                    */
/* ---------------------------------------------------------------------------------------------------------------- */
void oracle_undistort_point(const oracle_camera* c, double ul_corner_x, double u, double v, double* ox, double* oy) {
  const double cx = c->cx - ul_corner_x; /* camMatrixLocal(0,2) -= ulCorner.x, ref :522 */
  const double ifx = 1. / c->fx, ify = 1. / c->fy;
  double x = (u - cx) * ifx, y = (v - c->cy) * ify;
  const double x0 = x, y0 = y;
  const double k0 = c->k1, k1 = c->k2, k2 = c->p1, k3 = c->p2, k4 = c->k3; /* OpenCV's k[] order */
  for (int j = 0; j < 5; j++) { /* TermCriteria(MAX_ITER, 5, 0.01) */
    double r2 = x * x + y * y;
    double icdist = 1. / (1 + ((k4 * r2 + k1) * r2 + k0) * r2); /* numerator (1 + ((k7 r2 + k6) r2 + k5) r2) is exactly 1 */
    if (icdist < 0) { x = x0; y = y0; break; }
    double deltaX = 2 * k2 * x * y + k3 * (r2 + 2 * x * x);
    double deltaY = k2 * (r2 + 2 * y * y) + 2 * k3 * x * y;
    x = (x0 - deltaX) * icdist;
    y = (y0 - deltaY) * icdist;
  }
  *ox = x;
  *oy = y;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* linear algebra                                                                                                     */
/* ---------------------------------------------------------------------------------------------------------------- */
/* n x n system, augmented rows of n+1; partial pivoting; returns 0 when singular */
static int gauss_solve(double* m, int n, double* x) {
  const int w = n + 1;
  for (int c = 0; c < n; c++) {
    int piv = c;
    double big = fabs(m[c * w + c]);
    for (int r = c + 1; r < n; r++)
      if (fabs(m[r * w + c]) > big) { big = fabs(m[r * w + c]); piv = r; }
    if (!(big > 1e-300)) return 0;
    if (piv != c)
      for (int k = c; k <= n; k++) { double t = m[c * w + k]; m[c * w + k] = m[piv * w + k]; m[piv * w + k] = t; }
    const double inv = 1.0 / m[c * w + c];
    for (int r = c + 1; r < n; r++) {
      const double f = m[r * w + c] * inv;
      if (f != 0.0)
        for (int k = c; k <= n; k++) m[r * w + k] -= f * m[c * w + k];
    }
  }
  for (int r = n - 1; r >= 0; r--) {
    double s = m[r * w + n];
    for (int k = r + 1; k < n; k++) s -= m[r * w + k] * x[k];
    x[r] = s / m[r * w + r];
  }
  return 1;
}

/* cyclic Jacobi, symmetric n x n; eigenvalues end on the diagonal of a, eigenvectors are the columns of v */
static void jacobi(double* a, double* v, int n) {
  for (int i = 0; i < n; i++)
    for (int j = 0; j < n; j++) v[i * n + j] = (i == j);
  for (int sweep = 0; sweep < 60; sweep++) {
    double off = 0, diag = 0;
    for (int i = 0; i < n; i++) {
      diag += a[i * n + i] * a[i * n + i];
      for (int j = i + 1; j < n; j++) off += a[i * n + j] * a[i * n + j];
    }
    if (!(off > 1e-30 * diag) || off == 0.0) break;
    for (int p = 0; p < n - 1; p++)
      for (int q = p + 1; q < n; q++) {
        const double apq = a[p * n + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * n + q] - a[p * n + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < n; k++) {
          const double x = a[k * n + p], y = a[k * n + q];
          a[k * n + p] = c * x - s * y;
          a[k * n + q] = s * x + c * y;
        }
        for (int k = 0; k < n; k++) {
          const double x = a[p * n + k], y = a[q * n + k];
          a[p * n + k] = c * x - s * y;
          a[q * n + k] = s * x + c * y;
        }
        for (int k = 0; k < n; k++) {
          const double x = v[k * n + p], y = v[k * n + q];
          v[k * n + p] = c * x - s * y;
          v[k * n + q] = s * x + c * y;
        }
      }
  }
}

static void mul33(const double* a, const double* b, double* c) {
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) c[i * 3 + j] = a[i * 3] * b[j] + a[i * 3 + 1] * b[3 + j] + a[i * 3 + 2] * b[6 + j];
}

static double det33(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* cv::findHomography(RANSAC)                                                                                         */
/* ---------------------------------------------------------------------------------------------------------------- */
static uint64_t smix(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

static double tri_area2(const double* p, int i, int j, int k) {
  return (p[2 * j] - p[2 * i]) * (p[2 * k + 1] - p[2 * i + 1]) - (p[2 * j + 1] - p[2 * i + 1]) * (p[2 * k] - p[2 * i]);
}

static int fit4(const double* a, const double* b, const int* id, double* H) {
  double m[72], h[8];
  for (int k = 0; k < 4; k++) {
    const double x = a[2 * id[k]], y = a[2 * id[k] + 1], u = b[2 * id[k]], v = b[2 * id[k] + 1];
    const double r0[9] = {x, y, 1, 0, 0, 0, -u * x, -u * y, u};
    const double r1[9] = {0, 0, 0, x, y, 1, -v * x, -v * y, v};
    memcpy(m + 18 * k, r0, sizeof r0);
    memcpy(m + 18 * k + 9, r1, sizeof r1);
  }
  if (!gauss_solve(m, 8, h)) return 0;
  for (int k = 0; k < 8; k++) {
    if (!is_fin(h[k])) return 0;
    H[k] = h[k];
  }
  H[8] = 1.0;
  return 1;
}

static int inlier(const double* H, const double* a, const double* b, int i, double thr2) {
  const double x = a[2 * i], y = a[2 * i + 1];
  const double w = H[6] * x + H[7] * y + H[8];
  if (!(fabs(w) > DBL_EPSILON)) return 0;
  const double iw = 1.0 / w;
  const double dx = (H[0] * x + H[1] * y + H[2]) * iw - b[2 * i], dy = (H[3] * x + H[4] * y + H[5]) * iw - b[2 * i + 1];
  return dx * dx + dy * dy <= thr2;
}

static int hypothesis(const double* a, const double* b, int n, int iter, double thr2, double* H) {
  static const int tri[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
  uint64_t st = smix(RANSAC_SEED ^ ((uint64_t)(uint32_t)iter * 0xD1342543DE82EF95ull));
  for (int attempt = 0; attempt < 10; attempt++) {
    int id[4];
    for (int k = 0; k < 4; k++) {
      for (;;) {
        st = smix(st);
        const int cand = (int)((st >> 11) % (uint64_t)n);
        int dup = 0;
        for (int j = 0; j < k; j++) dup |= id[j] == cand;
        if (!dup) { id[k] = cand; break; }
      }
    }
    int ok = 1;
    for (int t = 0; t < 4 && ok; t++) {
      const double ca = tri_area2(a, id[tri[t][0]], id[tri[t][1]], id[tri[t][2]]);
      const double cb = tri_area2(b, id[tri[t][0]], id[tri[t][1]], id[tri[t][2]]);
      if (!(fabs(ca) > 1e-12) || !(fabs(cb) > 1e-12) || (ca > 0) != (cb > 0)) ok = 0;
    }
    if (!ok) continue;
    if (!fit4(a, b, id, H)) continue;
    int cnt = 0;
    for (int i = 0; i < n; i++) cnt += inlier(H, a, b, i, thr2);
    return cnt;
  }
  return 0;
}

static int update_iters(double p, double ep, int max_iters) { /* cv::RANSACUpdateNumIters, 4 model points */
  if (p < 0) p = 0;
  if (p > 1) p = 1;
  if (ep < 0) ep = 0;
  if (ep > 1) ep = 1;
  double num = 1 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  const double q = 1 - ep;
  double denom = 1 - q * q * q * q;
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  return denom >= 0 || -num >= max_iters * (-denom) ? max_iters : (int)nearbyint(num / denom);
}

static int refine_fit(const double* a, const double* b, const uint8_t* mask, int n, double* H) {
  int cnt = 0;
  double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
  for (int i = 0; i < n; i++)
    if (mask[i]) { cMx += a[2 * i]; cMy += a[2 * i + 1]; cmx += b[2 * i]; cmy += b[2 * i + 1]; cnt++; }
  if (cnt < 4) return 0;
  cMx /= cnt; cMy /= cnt; cmx /= cnt; cmy /= cnt;
  double sMx = 0, sMy = 0, smx = 0, smy = 0;
  for (int i = 0; i < n; i++)
    if (mask[i]) {
      sMx += fabs(a[2 * i] - cMx); sMy += fabs(a[2 * i + 1] - cMy);
      smx += fabs(b[2 * i] - cmx); smy += fabs(b[2 * i + 1] - cmy);
    }
  if (fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON || fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON) return 0;
  sMx = cnt / sMx; sMy = cnt / sMy; smx = cnt / smx; smy = cnt / smy;
  double LtL[81] = {0}, V[81];
  for (int i = 0; i < n; i++)
    if (mask[i]) {
      const double x = (b[2 * i] - cmx) * smx, y = (b[2 * i + 1] - cmy) * smy;
      const double X = (a[2 * i] - cMx) * sMx, Y = (a[2 * i + 1] - cMy) * sMy;
      const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
      const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
      for (int j = 0; j < 9; j++)
        for (int k = j; k < 9; k++) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
  for (int j = 0; j < 9; j++)
    for (int k = 0; k < j; k++) LtL[j * 9 + k] = LtL[k * 9 + j];
  jacobi(LtL, V, 9);
  int lo = 0;
  for (int k = 1; k < 9; k++)
    if (LtL[k * 9 + k] < LtL[lo * 9 + lo]) lo = k;
  double H0[9], T[9];
  for (int k = 0; k < 9; k++) H0[k] = V[k * 9 + lo];
  const double invHnorm[9] = {1. / smx, 0, cmx, 0, 1. / smy, cmy, 0, 0, 1};
  const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
  mul33(invHnorm, H0, T);
  mul33(T, Hnorm2, H0);
  if (!(fabs(H0[8]) > DBL_EPSILON)) return 0;
  const double sc = 1.0 / H0[8];
  double h[8];
  for (int k = 0; k < 8; k++) h[k] = H0[k] * sc;

  double lambda = 1e-3, S = 0;
  for (int i = 0; i < n; i++)
    if (mask[i]) {
      const double X = a[2 * i], Y = a[2 * i + 1], ww = 1.0 / (h[6] * X + h[7] * Y + 1.0);
      const double ex = (h[0] * X + h[1] * Y + h[2]) * ww - b[2 * i], ey = (h[3] * X + h[4] * Y + h[5]) * ww - b[2 * i + 1];
      S += ex * ex + ey * ey;
    }
  for (int it = 0; it < 10; it++) {
    double A[64] = {0}, g[8] = {0};
    for (int i = 0; i < n; i++)
      if (mask[i]) {
        const double X = a[2 * i], Y = a[2 * i + 1], ww = 1.0 / (h[6] * X + h[7] * Y + 1.0);
        const double xi = (h[0] * X + h[1] * Y + h[2]) * ww, yi = (h[3] * X + h[4] * Y + h[5]) * ww;
        const double ex = xi - b[2 * i], ey = yi - b[2 * i + 1];
        const double Jx[8] = {X * ww, Y * ww, ww, 0, 0, 0, -X * ww * xi, -Y * ww * xi};
        const double Jy[8] = {0, 0, 0, X * ww, Y * ww, ww, -X * ww * yi, -Y * ww * yi};
        for (int j = 0; j < 8; j++) {
          g[j] += Jx[j] * ex + Jy[j] * ey;
          for (int k = j; k < 8; k++) A[j * 8 + k] += Jx[j] * Jx[k] + Jy[j] * Jy[k];
        }
      }
    for (int j = 0; j < 8; j++)
      for (int k = 0; k < j; k++) A[j * 8 + k] = A[k * 8 + j];
    int accepted = 0;
    for (int tries = 0; tries < 6 && !accepted; tries++) {
      double m[72], d[8], hn[8];
      for (int j = 0; j < 8; j++) {
        for (int k = 0; k < 8; k++) m[j * 9 + k] = A[j * 8 + k];
        m[j * 9 + j] += lambda * A[j * 8 + j];
        m[j * 9 + 8] = -g[j];
      }
      if (gauss_solve(m, 8, d)) {
        double Sn = 0;
        for (int k = 0; k < 8; k++) hn[k] = h[k] + d[k];
        for (int i = 0; i < n; i++)
          if (mask[i]) {
            const double X = a[2 * i], Y = a[2 * i + 1], ww = 1.0 / (hn[6] * X + hn[7] * Y + 1.0);
            const double ex = (hn[0] * X + hn[1] * Y + hn[2]) * ww - b[2 * i], ey = (hn[3] * X + hn[4] * Y + hn[5]) * ww - b[2 * i + 1];
            Sn += ex * ex + ey * ey;
          }
        if (Sn < S) {
          memcpy(h, hn, sizeof h);
          S = Sn;
          lambda *= 0.1;
          accepted = 1;
          break;
        }
      }
      lambda *= 10.0;
    }
    if (!accepted) break;
  }
  memcpy(H, h, sizeof h);
  H[8] = 1.0;
  return 1;
}

int oracle_find_homography(const double* a, const double* b, int n, double* H, uint8_t* mask) {
  const double thr2 = RANSAC_THR * RANSAC_THR;
  double best[9] = {0};
  for (int i = 0; i < n; i++) mask[i] = 0;
  for (int k = 0; k < 9; k++) H[k] = 0;
  if (n < 4) return 0;
  if (n == 4) {
    const int id[4] = {0, 1, 2, 3};
    if (!fit4(a, b, id, best)) return 0;
    for (int i = 0; i < 4; i++) mask[i] = 1;
  } else {
    int best_count = 0, best_iter = -1, niters = RANSAC_ITERS;
    for (int iter = 0; iter < niters; iter++) { /* cv::RANSACPointSetRegistrator::run */
      double Hk[9];
      const int cnt = hypothesis(a, b, n, iter, thr2, Hk);
      if (cnt > (best_count > 3 ? best_count : 3)) { /* goodCount > max(maxGoodCount, modelPoints - 1) */
        best_count = cnt;
        best_iter = iter;
        memcpy(best, Hk, sizeof best);
        niters = update_iters(RANSAC_CONF, (double)(n - cnt) / n, niters);
      }
    }
    if (best_iter < 0) return 0;
    for (int i = 0; i < n; i++) mask[i] = (uint8_t)inlier(best, a, b, i, thr2);
  }
  if (!refine_fit(a, b, mask, n, H)) memcpy(H, best, sizeof best);
  return 1;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* ATTRIBUTION: restated from memory after OpenCV's modules/calib3d/src/homography_decomp.cpp (HomographyDecompInria:
 * Malis & Vargas, INRIA RR-6303, 2007; third-party, Apache-2.0 / BSD, (C) 2014 Samson Yilma and the OpenCV contributors),
 * not from /root/reference, which only calls it (optic_flow.cpp:595). Unpinned against a real OpenCV.                  */
/* cv::decomposeHomographyMat(H, I): HomographyDecompInria                                                            */
/* ---------------------------------------------------------------------------------------------------------------- */
static int sgn(double x) { return x >= 0 ? 1 : -1; }

static double opp_minor(const double* M, int row, int col) {
  const int x1 = col == 0 ? 1 : 0, x2 = col == 2 ? 1 : 2, y1 = row == 0 ? 1 : 0, y2 = row == 2 ? 1 : 2;
  return M[y1 * 3 + x2] * M[y2 * 3 + x1] - M[y1 * 3 + x1] * M[y2 * 3 + x2];
}

static void r_from_tstar_n(const double* Hn, const double* ts, const double* nv, double v, double* R) {
  double T[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) T[i * 3 + j] = (i == j ? 1.0 : 0.0) - (2 / v) * ts[i] * nv[j];
  mul33(Hn, T, R);
  if (det33(R) < 0)
    for (int k = 0; k < 9; k++) R[k] *= -1;
}

int oracle_decompose_homography(const double* H, double* R, double* t, double* nrm) {
  double G[9], V[9], Hn[9], S[9];
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) G[i * 3 + j] = H[i] * H[j] + H[3 + i] * H[3 + j] + H[6 + i] * H[6 + j];
  jacobi(G, V, 3); /* squares of the singular values (SVD::compute in removeScale) */
  const double e0 = G[0], e1 = G[4], e2 = G[8];
  const double mid = e0 > e1 ? (e1 > e2 ? e1 : (e0 > e2 ? e2 : e0)) : (e0 > e2 ? e0 : (e1 > e2 ? e2 : e1));
  if (!(mid > 0.0)) return 0;
  const double sc = 1.0 / sqrt(mid);
  for (int k = 0; k < 9; k++) Hn[k] = H[k] * sc;
  for (int i = 0; i < 3; i++)
    for (int j = 0; j < 3; j++) S[i * 3 + j] = Hn[i] * Hn[j] + Hn[3 + i] * Hn[3 + j] + Hn[6 + i] * Hn[6 + j];
  S[0] -= 1.0; S[4] -= 1.0; S[8] -= 1.0;
  double ninf = 0;
  for (int k = 0; k < 9; k++)
    if (fabs(S[k]) > ninf) ninf = fabs(S[k]);
  if (ninf < 0.001) { /* H is a rotation */
    memcpy(R, Hn, sizeof Hn);
    for (int k = 0; k < 3; k++) t[k] = nrm[k] = 0.0;
    return 1;
  }
  const double M00 = opp_minor(S, 0, 0), M11 = opp_minor(S, 1, 1), M22 = opp_minor(S, 2, 2);
  const double rtM00 = sqrt(M00), rtM11 = sqrt(M11), rtM22 = sqrt(M22);
  const double M01 = opp_minor(S, 0, 1), M12 = opp_minor(S, 1, 2), M02 = opp_minor(S, 0, 2);
  const int e12 = sgn(M12), e02 = sgn(M02), e01 = sgn(M01);
  const double nS00 = fabs(S[0]), nS11 = fabs(S[4]), nS22 = fabs(S[8]);
  int indx = 0;
  if (nS00 < nS11) { indx = 1; if (nS11 < nS22) indx = 2; }
  else if (nS00 < nS22) indx = 2;
  double npa[3], npb[3];
  switch (indx) {
    case 0:
      npa[0] = S[0];               npb[0] = S[0];
      npa[1] = S[1] + rtM22;       npb[1] = S[1] - rtM22;
      npa[2] = S[2] + e12 * rtM11; npb[2] = S[2] - e12 * rtM11;
      break;
    case 1:
      npa[0] = S[1] + rtM22;       npb[0] = S[1] - rtM22;
      npa[1] = S[4];               npb[1] = S[4];
      npa[2] = S[5] - e02 * rtM00; npb[2] = S[5] + e02 * rtM00;
      break;
    default:
      npa[0] = S[2] + e01 * rtM11; npb[0] = S[2] - e01 * rtM11;
      npa[1] = S[5] + rtM00;       npb[1] = S[5] - rtM00;
      npa[2] = S[8];               npb[2] = S[8];
      break;
  }
  const double traceS = S[0] + S[4] + S[8];
  const double v = 2.0 * sqrt(1 + traceS - M00 - M11 - M22);
  const double ESii = sgn(S[indx * 3 + indx]);
  const double r_2 = 2 + traceS + v, nt_2 = 2 + traceS - v;
  const double r = sqrt(r_2), n_t = sqrt(nt_2);
  const double la = sqrt(npa[0] * npa[0] + npa[1] * npa[1] + npa[2] * npa[2]);
  const double lb = sqrt(npb[0] * npb[0] + npb[1] * npb[1] + npb[2] * npb[2]);
  double na[3], nb[3], tas[3], tbs[3], Ra[9], Rb[9], ta[3], tb[3];
  for (int k = 0; k < 3; k++) { na[k] = npa[k] / la; nb[k] = npb[k] / lb; }
  const double half_nt = 0.5 * n_t, esii_t_r = ESii * r;
  for (int k = 0; k < 3; k++) {
    tas[k] = half_nt * (esii_t_r * nb[k] - n_t * na[k]);
    tbs[k] = half_nt * (esii_t_r * na[k] - n_t * nb[k]);
  }
  r_from_tstar_n(Hn, tas, na, v, Ra);
  r_from_tstar_n(Hn, tbs, nb, v, Rb);
  for (int i = 0; i < 3; i++) {
    ta[i] = Ra[i * 3] * tas[0] + Ra[i * 3 + 1] * tas[1] + Ra[i * 3 + 2] * tas[2];
    tb[i] = Rb[i * 3] * tbs[0] + Rb[i * 3 + 1] * tbs[1] + Rb[i * 3 + 2] * tbs[2];
  }
  memcpy(R, Ra, sizeof Ra); memcpy(R + 9, Ra, sizeof Ra); memcpy(R + 18, Rb, sizeof Rb); memcpy(R + 27, Rb, sizeof Rb);
  for (int k = 0; k < 3; k++) {
    t[k] = ta[k];      nrm[k] = na[k];
    t[3 + k] = -ta[k]; nrm[3 + k] = -na[k];
    t[6 + k] = tb[k];  nrm[6 + k] = nb[k];
    t[9 + k] = -tb[k]; nrm[9 + k] = -nb[k];
  }
  return 4;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* tf2 members                                                                                                        */
/* ---------------------------------------------------------------------------------------------------------------- */
typedef struct { double x, y, z, w; } quat;

static double acos_clamped(double v) { return acos(v < -1 ? -1 : (v > 1 ? 1 : v)); } /* tf2Acos */

static quat q_from_matrix(const double* m) { /* Matrix3x3::getRotation */
  const double trace = m[0] + m[4] + m[8];
  double temp[4];
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    temp[3] = s * 0.5;
    s = 0.5 / s;
    temp[0] = (m[7] - m[5]) * s;
    temp[1] = (m[2] - m[6]) * s;
    temp[2] = (m[3] - m[1]) * s;
  } else {
    const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    temp[i] = s * 0.5;
    s = 0.5 / s;
    temp[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
    temp[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
    temp[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
  }
  quat q = {temp[0], temp[1], temp[2], temp[3]};
  return q;
}

static double q_angle(quat q) { return 2. * acos_clamped(q.w); } /* getAngle */

static void q_axis(quat q, double* a) { /* getAxis */
  const double s_squared = 1. - q.w * q.w;
  if (s_squared < 10. * DBL_EPSILON) { a[0] = 1; a[1] = 0; a[2] = 0; return; }
  const double s = sqrt(s_squared);
  a[0] = q.x / s; a[1] = q.y / s; a[2] = q.z / s;
}

static quat q_axis_angle(const double* axis, double angle) { /* setRotation(axis, angle) */
  const double d = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  const double s = sin(angle * 0.5) / d;
  quat q = {axis[0] * s, axis[1] * s, axis[2] * s, cos(angle * 0.5)};
  return q;
}

static double q_angle_to(quat a, quat b) { /* Quaternion::angle */
  const double s = sqrt((a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) * (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w));
  return acos_clamped((a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w) / s);
}

static void q_basis(quat q, double* m) { /* Matrix3x3::setRotation */
  const double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const double s = 2.0 / d;
  const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  m[0] = 1.0 - (yy + zz); m[1] = xy - wz; m[2] = xz + wy;
  m[3] = xy + wz; m[4] = 1.0 - (xx + zz); m[5] = yz - wx;
  m[6] = xz - wy; m[7] = yz + wx; m[8] = 1.0 - (xx + yy);
}

static void xform(quat q, const double* origin, const double* v, double* out) { /* Transform * Vector3 */
  double m[9];
  q_basis(q, m);
  const double r0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + origin[0];
  const double r1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2] + origin[1];
  const double r2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2] + origin[2];
  out[0] = r0; out[1] = r1; out[2] = r2;
}

void oracle_quat_from_rpy(double roll, double pitch, double yaw, double* q) { /* Quaternion::setRPY, ref :1314 */
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
  q[0] = sr * cp * cy - cr * sp * sy;
  q[1] = cr * sp * cy + sr * cp * sy;
  q[2] = cr * cp * sy - sr * sp * cy;
  q[3] = cr * cp * cy + sr * sp * sy;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* OpticFlow::getRT                                                                                                   */
/* ---------------------------------------------------------------------------------------------------------------- */
int oracle_get_rt(const double* shifts, const oracle_geom_layout* L, const oracle_camera* cam, const oracle_rt_params* p,
                  int shifted_pts_thr, double* out, uint8_t* mask_out, double* H_out) {
  const int total = L->grid_x * L->grid_y;
  for (int k = 0; k < 7; k++) out[k] = k == 3 ? 1.0 : 0.0;
  if (mask_out) memset(mask_out, 0, (size_t)total);
  if (H_out) memset(H_out, 0, 9 * sizeof(double));
  if (!is_fin(1.0 / p->dt)) return 1; /* ref :516-519 */
  double* a = (double*)calloc(2 * (size_t)total, sizeof(double));
  double* b = (double*)calloc(2 * (size_t)total, sizeof(double));
  int* src = (int*)malloc(sizeof(int) * (size_t)total);
  uint8_t* mask = (uint8_t*)calloc((size_t)total, 1);
  int status = 7, n = 0;
  if (!a || !b || !src || !mask) { status = -2; goto done; }
  for (int j = 0; j < L->grid_y; j++) /* ref :527-542 */
    for (int i = 0; i < L->grid_x; i++) {
      const double sx = shifts[2 * (i + L->grid_x * j)], sy = shifts[2 * (i + L->grid_x * j) + 1];
      if (!is_fin(sx) || !is_fin(sy)) continue;
      const int xi = L->origin_x + i * L->stride_x + L->patch / 2, yi = L->origin_y + j * L->stride_y + L->patch / 2;
      oracle_undistort_point(cam, p->ul_corner_x, (double)xi, (double)yi, &a[2 * n], &a[2 * n + 1]);           /* :549 */
      oracle_undistort_point(cam, p->ul_corner_x, (double)xi + sx, (double)yi + sy, &b[2 * n], &b[2 * n + 1]); /* :550 */
      src[n++] = i + L->grid_x * j;
    }
  if (shifted_pts_thr < 0 || n < shifted_pts_thr) { status = 2; goto done; } /* :544-547, uint() of a negative never passes */
  {
    double H[9];
    const int found = oracle_find_homography(a, b, n, H, mask); /* :559 */
    int remaining = 0;
    for (int i = 0; i < n; i++) remaining += mask[i] == 1; /* :563-571 */
    if (mask_out)
      for (int i = 0; i < n; i++) mask_out[src[i]] = mask[i];
    if (remaining < shifted_pts_thr) { status = 3; goto done; } /* :575-578 */
    if (!found) { status = 8; goto done; }
    if (H_out) memcpy(H_out, H, sizeof H);

    double R[36], t[12], nrm[12];
    const int solutions = oracle_decompose_homography(H, R, t, nrm); /* :595 */
    const quat ang = {p->ang_rate_q[0], p->ang_rate_q[1], p->ang_rate_q[2], p->ang_rate_q[3]};
    const quat ang_inv = {-ang.x, -ang.y, -ang.z, ang.w};
    const quat c2b = {p->c2b_q[0], p->c2b_q[1], p->c2b_q[2], p->c2b_q[3]};
    int bestIndex = -1, bestInverse = 0;
    double bestAngDiff = G_PI;
    quat bestQ = {0, 0, 0, 1};
    for (int i = 0; i < solutions; i++) { /* :629-671 */
      double m[9], axis[3], axisB[3];
      for (int j = 0; j < 3; j++)
        for (int k = 0; k < 3; k++) m[k * 3 + j] = R[i * 9 + j * 3 + k]; /* cvMat33ToTf2Mat33, :76-85 */
      const quat q = q_from_matrix(m);                                  /* :639-640 */
      q_axis(q, axis);
      xform(c2b, p->c2b_t, axis, axisB);
      const quat qB = q_axis_angle(axisB, q_angle(q) / p->dt);          /* :643 */
      const double plus = q_angle_to(qB, ang), minus = q_angle_to(qB, ang_inv);
      const double angDiff = plus < minus ? plus : minus;               /* :649-655 */
      const int inverseSolution = nrm[i * 3 + 2] < 0 ? 0 : 1;           /* :657-660 */
      if (bestAngDiff > angDiff) { bestAngDiff = angDiff; bestIndex = i; bestInverse = inverseSolution; bestQ = q; }
    }
    const double zero[3] = {0, 0, 0};
    if (bestIndex != -1 && solutions > 1) { /* :674 */
      if (bestAngDiff > G_PI / 4) { status = 4; goto done; } /* :682-685 */
      double axis[3], tv[3], r[3];
      q_axis(bestQ, axis);
      const quat o = q_axis_angle(axis, q_angle(bestQ) / p->dt); /* :703 */
      const double invUnit = bestInverse ? -1.0 : 1.0;          /* :719 */
      for (int k = 0; k < 3; k++) tv[k] = invUnit * t[bestIndex * 3 + k];
      xform(bestQ, zero, tv, r);
      const double idt = 1.0 / p->dt; /* tf2::Vector3::operator/(v, s) = v * (1 / s) */
      out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
      for (int k = 0; k < 3; k++) out[4 + k] = r[k] * p->height * idt; /* :720-722 */
      status = 0;
    } else if (solutions == 1) { /* :756 */
      if (bestIndex == -1) { status = 5; goto done; }
      double axis[3], r[3];
      q_axis(bestQ, axis);
      const quat o = q_axis_angle(axis, q_angle(bestQ) / p->dt); /* :737 */
      xform(bestQ, zero, t, r);
      const double idt = 1.0 / p->dt;
      out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
      for (int k = 0; k < 3; k++) out[4 + k] = r[k] * p->height * idt; /* :741 */
      status = 0;
      for (int k = 0; k < 7; k++)
        if (!is_fin(out[k])) status = 6; /* :744-751 */
    } else {
      status = solutions == 0 ? 8 : 7; /* :769-771 */
    }
  }
done:
  if (status != 0)
    for (int k = 0; k < 7; k++) out[k] = k == 3 ? 1.0 : 0.0;
  free(a); free(b); free(src); free(mask);
  return status;
}

/* ---------------------------------------------------------------------------------------------------------------- */
/* OpticFlow::get2DT (LONG_RANGE_RATIO == 4 branches only: FftMethod.cpp:3)                                           */
/* ---------------------------------------------------------------------------------------------------------------- */
int oracle_get_2dt(const double* shifts, const oracle_geom_layout* L, const oracle_camera* cam, const oracle_2dt_params* p,
                   double* out) {
  for (int k = 0; k < 6; k++) out[k] = 0.0;
  if (L->grid_x * L->grid_y < 1) return 9; /* ref :389-392 */
  if (!is_fin(1.0 / p->dt)) return 1;      /* :393-396 */
  /* :402-420 -- initialPts / shiftedPts in patch order; only element 0 of undistShifts is ever used (:471) */
  int have = 0;
  double avgx = 0, avgy = 0;
  for (int j = 0; j < L->grid_y && !have; j++)
    for (int i = 0; i < L->grid_x && !have; i++) {
      const double sx = shifts[2 * (i + L->grid_x * j)], sy = shifts[2 * (i + L->grid_x * j) + 1];
      if (!is_fin(sx) || !is_fin(sy)) continue;
      const int xi = L->origin_x + i * L->stride_x + L->patch / 2, yi = L->origin_y + j * L->stride_y + L->patch / 2;
      const double shx = (double)xi + sx, shy = (double)yi + sy; /* shiftedPts */
      avgx = shx - (double)xi;                                   /* undistShifts = shiftedPts - initialPts, :451-454 */
      avgy = shy - (double)yi;
      have = 1;
    }
  if (!have) return 2; /* :425-429 */
  const double multiplier = 4;
  const double fx = cam->fx, fy = cam->fy; /* camMatrixLocal(0,0), (1,1): the -= ulCorner.x touches (0,2) only */
  const double x_corr = -tan(p->roll_rate * p->dt) * fx / multiplier; /* :481 */
  const double y_corr = tan(p->pitch_rate * p->dt) * fy / multiplier; /* :482 */
  const double t_corr = sqrt(y_corr * y_corr + x_corr * x_corr);
  const double yaw_corr = atan2(y_corr, x_corr) + p->cam_yaw;
  const double x_corr_cam = cos(yaw_corr) * t_corr, y_corr_cam = sin(yaw_corr) * t_corr;
  const double idt = 1.0 / p->dt;
  avgx += x_corr_cam;
  avgy += y_corr_cam;
  double tran[3] = {avgx * (p->height / fx * multiplier), avgy * (p->height / fy * multiplier), 0.0};
  for (int k = 0; k < 3; k++) tran[k] = -tran[k] * idt; /* o_tran = -o_tran / dur_.toSec(), :495 */
  avgx += x_corr_cam;
  avgy += y_corr_cam;
  double corr[3] = {avgx * (p->height / fx * multiplier), avgy * (p->height / fy * multiplier), 0.0};
  for (int k = 0; k < 3; k++) corr[k] = -corr[k] * idt; /* :505 */
  for (int k = 0; k < 3; k++) { out[k] = tran[k]; out[3 + k] = corr[k] - tran[k]; } /* :507 */
  return 0;
}
