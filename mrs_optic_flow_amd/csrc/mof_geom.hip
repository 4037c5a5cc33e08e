// mof_geom.hip -- the geometry tail behind the C ABI (mof_geom_* in include/mof.h): OpticFlow::getRT
// (/root/reference/src/optic_flow.cpp:515-774) and OpticFlow::get2DT (:388-510), SURVEY.md section 8(f) N1 / N3.
//
// Two forms of each: a host call for the node's one-frame-at-a-time use (a few hundred points: the GPU has nothing to
// add and a launch + read-back would cost more than the arithmetic), and a batched device form that keeps a whole
// batch of flow fields in HBM -- one wavefront per frame pair, the 64 lanes evaluating 64 RANSAC hypotheses at a time
// -- so that "frames -> camera-frame velocity" needs no host round trip. Both run geom_core.hpp.
// Built with -ffp-contract=off: the same source then rounds the same way on host and device (libm calls aside).

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstring>
#include <vector>

#include "geom_core.hpp"
#include "mof.h"

namespace mof {
int capi_fail(int code, const char* fmt, ...);  // mof_capi.hip
}

namespace {

using namespace mof::geom;

static_assert(sizeof(mof_geom_camera) == sizeof(Camera), "mof_geom_camera mirrors geom::Camera");
static_assert(sizeof(mof_geom_layout) == sizeof(Layout), "mof_geom_layout mirrors geom::Layout");
static_assert(sizeof(mof_geom_rt_params) == sizeof(RtParams), "mof_geom_rt_params mirrors geom::RtParams");
static_assert(sizeof(mof_geom_2dt_params) == sizeof(T2dParams), "mof_geom_2dt_params mirrors geom::T2dParams");

int check_layout(const mof_geom_layout* L) {
  if (!L || L->grid_x < 1 || L->grid_y < 1 || L->patch_size < 1 || L->stride_x < 0 || L->stride_y < 0 ||
      (long)L->grid_x * L->grid_y > kMaxPoints)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "bad geometry layout (1 <= grid_x * grid_y <= %d)", kMaxPoints);
  return MOF_OK;
}

int check_camera(const mof_geom_camera* c) {
  if (!c || !(c->fx != 0.0) || !(c->fy != 0.0) || !finite_d(c->fx) || !finite_d(c->fy))
    return mof::capi_fail(MOF_ERR_BAD_ARG, "bad camera (fx, fy must be finite and non-zero)");
  return MOF_OK;
}

// Valid shifts -> normalised point pairs, in patch order (optic_flow.cpp:527-550). Returns the count.
int gather_points_host(const double* shifts, const Layout& L, const Camera& cam, double ulx, double* a, double* b, int* src_index) {
  int n = 0;
  const double cxl = cam.cx - ulx;  // camMatrixLocal(0, 2) -= ulCorner.x, :522
  for (int j = 0; j < L.grid_y; ++j)
    for (int i = 0; i < L.grid_x; ++i) {
      const double sx = shifts[2 * (i + L.grid_x * j)], sy = shifts[2 * (i + L.grid_x * j) + 1];
      if (!finite_d(sx) || !finite_d(sy)) continue;  // :531-535
      const int xi = L.origin_x + i * L.stride_x + L.patch / 2, yi = L.origin_y + j * L.stride_y + L.patch / 2;  // :537-538
      undistort_point(cam, cxl, (double)xi, (double)yi, &a[2 * n], &a[2 * n + 1]);                                // :549
      undistort_point(cam, cxl, (double)xi + sx, (double)yi + sy, &b[2 * n], &b[2 * n + 1]);                      // :550
      src_index[n] = i + L.grid_x * j;
      ++n;
    }
  return n;
}

// cv::findHomography(a, b, RANSAC, 0.01, mask) as specified in geom_core.hpp. Returns false when no model was found.
bool find_homography_host(const double* a, const double* b, int n, double* H, unsigned char* mask) {
  const double thr2 = kRansacThreshold * kRansacThreshold;
  for (int i = 0; i < n; ++i) mask[i] = 0;
  if (n < 4) return false;
  double best[9];
  if (n == 4) {  // cv::findHomography solves 4 points directly, all inliers
    const int idx[4] = {0, 1, 2, 3};
    if (!homography_4pt(a, b, idx, best)) return false;
    for (int i = 0; i < 4; ++i) mask[i] = 1;
  } else {
    RansacScan scan{0, -1, kRansacMaxIters};
    bool running = true;
    for (int base = 0; base < kRansacMaxIters && running; base += kRansacBatch) {
      for (int l = 0; l < kRansacBatch && running; ++l) {
        double Hk[9];
        const int iter = base + l;
        if (iter >= scan.niters) {
          running = false;
          break;
        }
        const int cnt = ransac_hypothesis(a, b, n, kRansacSeed, iter, thr2, Hk);
        const int before = scan.best_iter;
        running = ransac_accept(&scan, iter, cnt, n);
        if (scan.best_iter != before) std::memcpy(best, Hk, sizeof(best));
      }
    }
    if (scan.best_iter < 0) return false;
    for (int i = 0; i < n; ++i) mask[i] = is_inlier(best, a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1], thr2) ? 1 : 0;
  }
  double fit[9];
  if (homography_fit(a, b, mask, n, fit)) std::memcpy(H, fit, sizeof(fit));
  else std::memcpy(H, best, sizeof(best));
  return true;
}

int get_rt_host(const double* shifts, const Layout& L, const Camera& cam, const RtParams& p, int thr, double* out,
                unsigned char* mask_out, double* H_out) {
  const int total = L.grid_x * L.grid_y;
  if (mask_out) std::memset(mask_out, 0, (size_t)total);
  if (!finite_d(1.0 / p.dt)) return kBadDuration;  // :516-519
  std::vector<double> a(2 * (size_t)total), b(2 * (size_t)total);
  std::vector<int> src((size_t)total);
  std::vector<unsigned char> mask((size_t)total);
  const int n = gather_points_host(shifts, L, cam, p.ul_corner_x, a.data(), b.data(), src.data());
  if (thr < 0 || n < thr) return kTooFewPoints;    // `shiftedPts.size() < uint(_shifted_pts_thr_)`, :544-547
  double H[9];
  const bool found = find_homography_host(a.data(), b.data(), n, H, mask.data());  // :559
  int remaining = 0;
  for (int i = 0; i < n; ++i) remaining += mask[i] == 1;  // :563-571 (`allSmall` can never become true there)
  if (mask_out)
    for (int i = 0; i < n; ++i) mask_out[src[i]] = mask[i];
  if (remaining < thr) return kTooFewInliers;      // :575-578
  if (!found) return kNoHomography;
  if (H_out) std::memcpy(H_out, H, sizeof(H));
  return pick_motion(H, p, out);                   // :594-771
}

// ---- device: one wavefront per frame pair --------------------------------------------------------------------

__device__ __forceinline__ void geom_wave_sync() {
  __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
  __builtin_amdgcn_wave_barrier();
  __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// homography_fit (geom_core.hpp) executed by a whole wavefront. The one-lane form kept its 9 x 9 / 8 x 9 work arrays in
// scratch memory (dynamic indices) and took 2.5 M of the kernel's 2.7 M cycles per pair; here the matrices live in LDS
// and the lanes share the independent pieces: per-point rows (lane = point), normal-matrix entries (lane = entry, the sum
// over the points in point order), the three element loops of a Jacobi rotation and the row operations of the
// elimination (lane = element). Everything that is a sequential recurrence in the host form (rotation angles, pivots,
// back substitution, the ordered error sums) is evaluated redundantly by every lane from the same LDS values. Each
// accumulator therefore sees exactly the host form's operations in the host form's order: the results are bit-identical
// (-ffp-contract=off on both sides). All lanes return the same H.
// ws: tab[n][18], A[81], V[81], M[72], g[8] doubles in LDS.
constexpr int kFitTab = 18;
constexpr int kWaveFitPoints = 256;  // 36 KB of per-point rows: every BASELINE grid (c4: 16 x 16)
__device__ bool homography_fit_wave(const double* a, const double* b, const unsigned char* mask, int n, double* H, double* ws,
                                    int lane) {
  double* tab = ws;
  double* A = ws + (size_t)n * kFitTab;
  double* V = A + 81;
  double* M = V + 81;
  double* g = M + 72;
  int cnt = 0;
  double cMx = 0, cMy = 0, cmx = 0, cmy = 0;
  for (int i = 0; i < n; ++i)
    if (mask[i]) {
      cMx += a[2 * i]; cMy += a[2 * i + 1]; cmx += b[2 * i]; cmy += b[2 * i + 1];
      ++cnt;
    }
  if (cnt < 4) return false;
  cMx /= cnt; cMy /= cnt; cmx /= cnt; cmy /= cnt;
  double sMx = 0, sMy = 0, smx = 0, smy = 0;
  for (int i = 0; i < n; ++i)
    if (mask[i]) {
      sMx += fabs(a[2 * i] - cMx); sMy += fabs(a[2 * i + 1] - cMy);
      smx += fabs(b[2 * i] - cmx); smy += fabs(b[2 * i + 1] - cmy);
    }
  if (fabs(sMx) < DBL_EPSILON || fabs(sMy) < DBL_EPSILON || fabs(smx) < DBL_EPSILON || fabs(smy) < DBL_EPSILON) return false;
  sMx = cnt / sMx; sMy = cnt / sMy; smx = cnt / smx; smy = cnt / smy;
  // ---- normalised DLT: rows of L per point, then LtL entry (j, k), j <= k, per lane
  for (int i = lane; i < n; i += 64)
    if (mask[i]) {
      const double x = (b[2 * i] - cmx) * smx, y = (b[2 * i + 1] - cmy) * smy;
      const double X = (a[2 * i] - cMx) * sMx, Y = (a[2 * i + 1] - cMy) * sMy;
      double* t = tab + (size_t)i * kFitTab;
      t[0] = X, t[1] = Y, t[2] = 1, t[3] = 0, t[4] = 0, t[5] = 0, t[6] = -x * X, t[7] = -x * Y, t[8] = -x;
      t[9] = 0, t[10] = 0, t[11] = 0, t[12] = X, t[13] = Y, t[14] = 1, t[15] = -y * X, t[16] = -y * Y, t[17] = -y;
    }
  geom_wave_sync();
  if (lane < 45) {
    int j = 0, k = lane;
    while (k >= 9 - j) k -= 9 - j, ++j;  // lane -> (j, k = j + offset): row-major upper triangle
    k += j;
    double acc = 0.0;
    for (int i = 0; i < n; ++i)
      if (mask[i]) {
        const double* t = tab + (size_t)i * kFitTab;
        acc += t[j] * t[k] + t[9 + j] * t[9 + k];
      }
    A[j * 9 + k] = acc;
    A[k * 9 + j] = acc;
  }
  for (int e = lane; e < 81; e += 64) V[e] = (e / 9 == e % 9) ? 1.0 : 0.0;
  geom_wave_sync();
  // ---- jacobi_eigen<9>: same sweep order, same arithmetic per element
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < 9; ++i) {
      diag += A[i * 9 + i] * A[i * 9 + i];
      for (int j = i + 1; j < 9; ++j) off += A[i * 9 + j] * A[i * 9 + j];
    }
    if (!(off > 1e-30 * diag) || off == 0.0) break;
    for (int p = 0; p < 8; ++p)
      for (int q = p + 1; q < 9; ++q) {
        const double apq = A[p * 9 + q];
        if (apq == 0.0) continue;
        const double theta = (A[q * 9 + q] - A[p * 9 + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), sn = t * c;
        // columns p, q of A (lanes 0..8) beside columns p, q of V (lanes 16..24)
        if (lane < 9 || (lane >= 16 && lane < 25)) {
          double* m = lane < 9 ? A : V;
          const int k = lane < 9 ? lane : lane - 16;
          const double mkp = m[k * 9 + p], mkq = m[k * 9 + q];
          m[k * 9 + p] = c * mkp - sn * mkq;
          m[k * 9 + q] = sn * mkp + c * mkq;
        }
        geom_wave_sync();
        if (lane < 9) {  // rows p, q of A
          const double apk = A[p * 9 + lane], aqk = A[q * 9 + lane];
          A[p * 9 + lane] = c * apk - sn * aqk;
          A[q * 9 + lane] = sn * apk + c * aqk;
        }
        geom_wave_sync();
      }
  }
  int lo = 0;
  for (int k = 1; k < 9; ++k)
    if (A[k * 9 + k] < A[lo * 9 + lo]) lo = k;
  double H0[9];
  for (int k = 0; k < 9; ++k) H0[k] = V[k * 9 + lo];
  const double invHnorm[9] = {1.0 / smx, 0, cmx, 0, 1.0 / smy, cmy, 0, 0, 1};
  const double Hnorm2[9] = {sMx, 0, -cMx * sMx, 0, sMy, -cMy * sMy, 0, 0, 1};
  double T[9];
  mat3_mul(invHnorm, H0, T);
  mat3_mul(T, Hnorm2, H0);
  if (!(fabs(H0[8]) > DBL_EPSILON)) return false;
  const double sc = 1.0 / H0[8];
  double h[8];
  for (int k = 0; k < 8; ++k) h[k] = H0[k] * sc;
  geom_wave_sync();  // every lane has read A / V before the table is rewritten

  // ---- Levenberg-Marquardt on (h0..h7), h8 = 1
  auto error_sum = [&](const double* hh) {  // the host form's `S += ex * ex + ey * ey` over the points in order
    for (int i = lane; i < n; i += 64)
      if (mask[i]) {
        const double X = a[2 * i], Y = a[2 * i + 1];
        const double ww = 1.0 / (hh[6] * X + hh[7] * Y + 1.0);
        const double ex = (hh[0] * X + hh[1] * Y + hh[2]) * ww - b[2 * i], ey = (hh[3] * X + hh[4] * Y + hh[5]) * ww - b[2 * i + 1];
        tab[(size_t)i * kFitTab + 16] = ex * ex + ey * ey;
      }
    geom_wave_sync();
    double S = 0.0;
    for (int i = 0; i < n; ++i)
      if (mask[i]) S += tab[(size_t)i * kFitTab + 16];
    geom_wave_sync();  // the slot may be rewritten
    return S;
  };
  double lambda = 1e-3;
  double S = error_sum(h);
  for (int it = 0; it < 10; ++it) {
    for (int i = lane; i < n; i += 64)
      if (mask[i]) {
        const double X = a[2 * i], Y = a[2 * i + 1];
        const double ww = 1.0 / (h[6] * X + h[7] * Y + 1.0);
        const double xi = (h[0] * X + h[1] * Y + h[2]) * ww, yi = (h[3] * X + h[4] * Y + h[5]) * ww;
        double* t = tab + (size_t)i * kFitTab;
        t[0] = X * ww, t[1] = Y * ww, t[2] = ww, t[3] = 0, t[4] = 0, t[5] = 0, t[6] = -X * ww * xi, t[7] = -Y * ww * xi;
        t[8] = 0, t[9] = 0, t[10] = 0, t[11] = X * ww, t[12] = Y * ww, t[13] = ww, t[14] = -X * ww * yi, t[15] = -Y * ww * yi;
        t[16] = xi - b[2 * i];
        t[17] = yi - b[2 * i + 1];
      }
    geom_wave_sync();
    if (lane < 36) {  // A8 entry (j, k), j <= k
      int j = 0, k = lane;
      while (k >= 8 - j) k -= 8 - j, ++j;
      k += j;
      double acc = 0.0;
      for (int i = 0; i < n; ++i)
        if (mask[i]) {
          const double* t = tab + (size_t)i * kFitTab;
          acc += t[j] * t[k] + t[8 + j] * t[8 + k];
        }
      A[j * 8 + k] = acc;
      A[k * 8 + j] = acc;
    } else if (lane < 44) {
      const int j = lane - 36;
      double acc = 0.0;
      for (int i = 0; i < n; ++i)
        if (mask[i]) {
          const double* t = tab + (size_t)i * kFitTab;
          acc += t[j] * t[16] + t[8 + j] * t[17];
        }
      g[j] = acc;
    }
    geom_wave_sync();
    bool accepted = false;
    for (int tries = 0; tries < 6 && !accepted; ++tries) {
      for (int e = lane; e < 72; e += 64) {
        const int j = e / 9, k = e % 9;
        double v = k < 8 ? A[j * 8 + k] : -g[j];
        if (k == j) v += lambda * A[j * 8 + j];
        M[e] = v;
      }
      geom_wave_sync();
      // solve_linear<8> on M (8 x 9)
      bool ok = true;
      for (int c = 0; c < 8 && ok; ++c) {
        int piv = c;
        double best = fabs(M[c * 9 + c]);
        for (int r = c + 1; r < 8; ++r) {
          const double v = fabs(M[r * 9 + c]);
          if (v > best) best = v, piv = r;
        }
        if (!(best > 1e-300)) {
          ok = false;
          break;
        }
        if (piv != c) {
          if (lane >= c && lane <= 8) {
            const double t = M[c * 9 + lane];
            M[c * 9 + lane] = M[piv * 9 + lane];
            M[piv * 9 + lane] = t;
          }
          geom_wave_sync();
        }
        const double inv = 1.0 / M[c * 9 + c];
        {
          const int r = c + 1 + lane / 9, k = lane % 9;
          if (r < 8 && k >= c) {
            const double f = M[r * 9 + c] * inv;
            const double mck = M[c * 9 + k], mrk = M[r * 9 + k];
            if (f != 0.0) M[r * 9 + k] = mrk - f * mck;
          }
        }
        geom_wave_sync();
      }
      double d[8];
      if (ok) {
        for (int r = 7; r >= 0; --r) {
          double sacc = M[r * 9 + 8];
          for (int k = r + 1; k < 8; ++k) sacc -= M[r * 9 + k] * d[k];
          d[r] = sacc / M[r * 9 + r];
        }
      }
      geom_wave_sync();  // M is rebuilt by the next try
      if (ok) {
        double hn[8];
        for (int k = 0; k < 8; ++k) hn[k] = h[k] + d[k];
        const double Sn = error_sum(hn);
        if (Sn < S) {
          for (int k = 0; k < 8; ++k) h[k] = hn[k];
          S = Sn;
          lambda *= 0.1;
          accepted = true;
          break;
        }
      }
      lambda *= 10.0;
    }
    if (!accepted) break;
  }
  for (int k = 0; k < 8; ++k) H[k] = h[k];
  H[8] = 1.0;
  return true;
}


__global__ void __launch_bounds__(64) geom_get_rt_kernel(const double* __restrict__ shifts, Layout L, Camera cam,
                                                         const RtParams* __restrict__ params, int thr, int wave_fit,
                                                         double* __restrict__ out) {
  extern __shared__ double lds[];
  const int total = L.grid_x * L.grid_y;
  double* a = lds;
  double* b = lds + 2 * total;
  unsigned char* mask = reinterpret_cast<unsigned char*>(lds + 4 * total);
  const int lane = threadIdx.x, pair = blockIdx.x;
  const RtParams p = params[pair];
  const double* sh = shifts + (size_t)pair * total * 2;
  double* o = out + (size_t)pair * 8;
  // every exit path writes the 8 outputs exactly once, from lane 0: identity rotation and zero translation unless
  // the call succeeds, then the status
  auto leave = [&](int status) {
    if (lane == 0) {
      for (int k = 0; k < 7; ++k) o[k] = k == 3 ? 1.0 : 0.0;
      o[7] = (double)status;
    }
  };
  if (!finite_d(1.0 / p.dt)) {
    leave(kBadDuration);
    return;
  }
  // ordered compaction of the valid patches, undistorted on the way
  const double cxl = cam.cx - p.ul_corner_x;
  int n = 0;
  for (int base = 0; base < total; base += 64) {
    const int idx = base + lane;
    double sx = 0, sy = 0;
    bool valid = false;
    if (idx < total) {
      sx = sh[2 * idx];
      sy = sh[2 * idx + 1];
      valid = finite_d(sx) && finite_d(sy);
    }
    const unsigned long long bal = __ballot(valid);
    if (valid) {
      const int pos = n + __popcll(bal & ((1ull << lane) - 1ull));
      const int i = idx % L.grid_x, j = idx / L.grid_x;
      const int xi = L.origin_x + i * L.stride_x + L.patch / 2, yi = L.origin_y + j * L.stride_y + L.patch / 2;
      undistort_point(cam, cxl, (double)xi, (double)yi, &a[2 * pos], &a[2 * pos + 1]);
      undistort_point(cam, cxl, (double)xi + sx, (double)yi + sy, &b[2 * pos], &b[2 * pos + 1]);
    }
    n += __popcll(bal);
  }
  __syncthreads();
  if (thr < 0 || n < thr) {
    leave(kTooFewPoints);
    return;
  }
#ifdef MOF_GEOM_PROF  // diagnostic build (tools/geom_stage_cycles.py): stage cycle counts instead of results
  const unsigned long long t_a = __builtin_readcyclecounter();
#endif
  const double thr2 = kRansacThreshold * kRansacThreshold;
  double best[9];
  bool found = false;
  if (n == 4) {
    const int idx[4] = {0, 1, 2, 3};
    found = homography_4pt(a, b, idx, best);
    if (lane < 4) mask[lane] = found ? 1 : 0;
  } else if (n > 4) {
    RansacScan scan{0, -1, kRansacMaxIters};
    bool running = true;
    for (int base = 0; base < kRansacMaxIters && running && base < scan.niters; base += kRansacBatch) {
      double Hk[9];
      const int cnt = ransac_hypothesis(a, b, n, kRansacSeed, base + lane, thr2, Hk);
      const int before = scan.best_iter;
      for (int l = 0; l < kRansacBatch; ++l) {  // the host's in-order acceptance, replayed identically by every lane
        const int c = __shfl(cnt, l, 64);
        if (!ransac_accept(&scan, base + l, c, n)) {
          running = false;
          break;
        }
      }
      if (scan.best_iter != before) {
        const int owner = scan.best_iter - base;
        for (int k = 0; k < 9; ++k) best[k] = __shfl(Hk[k], owner, 64);
      }
    }
    found = scan.best_iter >= 0;
    for (int i = lane; i < n; i += 64)
      mask[i] = (found && is_inlier(best, a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1], thr2)) ? 1 : 0;
  } else {
    for (int i = lane; i < n; i += 64) mask[i] = 0;
  }
  __syncthreads();
  int remaining = 0;
  for (int i = 0; i < n; ++i) remaining += mask[i];
  if (remaining < thr || !found) {
    leave(remaining < thr ? kTooFewInliers : kNoHomography);
    return;
  }
#ifdef MOF_GEOM_PROF
  const unsigned long long t_b = __builtin_readcyclecounter();
#endif
  // refit on the inliers: by the whole wave when its LDS work space was granted (up to kWaveFitPoints patches), else by
  // lane 0 through the host form
  double H[9];
  bool fitted;
  if (wave_fit) {
    fitted = homography_fit_wave(a, b, mask, n, H, reinterpret_cast<double*>(mask + ((total + 7) & ~7)), lane);
  } else {
    fitted = lane == 0 && homography_fit(a, b, mask, n, H);
  }
#ifdef MOF_GEOM_PROF
  const unsigned long long t_c = __builtin_readcyclecounter();
#endif
  if (lane == 0) {  // decomposition + IMU-consistent pick: sequential
    double res[7] = {0, 0, 0, 1, 0, 0, 0};
    if (!fitted)
      for (int k = 0; k < 9; ++k) H[k] = best[k];
    const int status = pick_motion(H, p, res);
#ifdef MOF_GEOM_PROF
    res[0] = (double)(t_b - t_a), res[1] = (double)(t_c - t_b), res[2] = (double)(__builtin_readcyclecounter() - t_c);
#endif
#ifdef MOF_GEOM_PROF
    const bool keep = true;
#else
    const bool keep = status == kOk;
#endif
    for (int k = 0; k < 7; ++k) o[k] = keep ? res[k] : (k == 3 ? 1.0 : 0.0);
    o[7] = (double)status;
  }
}

__global__ void __launch_bounds__(64) geom_get_2dt_kernel(const double* __restrict__ shifts, Layout L, Camera cam,
                                                          const T2dParams* __restrict__ params, int n_pairs,
                                                          double* __restrict__ out) {
  const int pair = blockIdx.x * 64 + threadIdx.x;
  if (pair >= n_pairs) return;
  double res[6] = {0, 0, 0, 0, 0, 0};
  const int status = get_2dt(shifts + (size_t)pair * L.grid_x * L.grid_y * 2, L, cam, params[pair], res);
  double* o = out + (size_t)pair * 8;
  for (int k = 0; k < 6; ++k) o[k] = status == kOk ? res[k] : 0.0;
  o[6] = (double)status;
  o[7] = 0.0;
}

}  // namespace

namespace {
// The geometry ABI carries no device ordinal: launch on the device that owns the output buffer (the stream and the
// other pointers must belong to it too), leaving the calling thread's current device as it was.
struct DeviceOf {
  int prev = -1;
  hipError_t err = hipSuccess;
  explicit DeviceOf(const void* p) {
    hipPointerAttribute_t at{};
    err = hipPointerGetAttributes(&at, p);
    if (err != hipSuccess) return;
    int cur = 0;
    if ((err = hipGetDevice(&cur)) != hipSuccess) return;
    if (cur != at.device) {
      if ((err = hipSetDevice(at.device)) == hipSuccess) prev = cur;
    }
  }
  ~DeviceOf() {
    if (prev >= 0) (void)hipSetDevice(prev);
  }
};
}  // namespace

extern "C" {

int mof_geom_layout_reference(mof_geom_layout* L, int frame_size, int sample_point_size) {
  if (!L || frame_size < 1 || sample_point_size < 1 || frame_size / sample_point_size < 1)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "bad reference geometry");
  const int sq = frame_size / sample_point_size;  // sqNum = _frame_size_ / _sample_point_size_, optic_flow.cpp:525
  *L = mof_geom_layout{sq, sq, 0, 0, sample_point_size, sample_point_size, sample_point_size};
  return MOF_OK;
}

int mof_geom_undistort_points(const mof_geom_camera* cam, double ul_corner_x, const double* pts_xy, int n, double* out_xy) {
  int rc = check_camera(cam);
  if (rc) return rc;
  if (n == 0) return MOF_OK;
  if (!pts_xy || !out_xy || n < 0) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad point arguments");
  Camera c;
  std::memcpy(&c, cam, sizeof(c));
  for (int i = 0; i < n; ++i) undistort_point(c, c.cx - ul_corner_x, pts_xy[2 * i], pts_xy[2 * i + 1], &out_xy[2 * i], &out_xy[2 * i + 1]);
  return MOF_OK;
}

int mof_geom_find_homography(const double* a_xy, const double* b_xy, int n, double* H9, uint8_t* mask, int* found) {
  if (!a_xy || !b_xy || !H9 || !mask || !found || n < 0 || n > kMaxPoints)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "bad homography arguments (0 <= n <= %d)", kMaxPoints);
  for (int k = 0; k < 9; ++k) H9[k] = 0.0;
  *found = find_homography_host(a_xy, b_xy, n, H9, mask) ? 1 : 0;
  return MOF_OK;
}

int mof_geom_decompose_homography(const double* H9, double* R, double* t, double* normals, int* n_solutions) {
  if (!H9 || !R || !t || !normals || !n_solutions) return mof::capi_fail(MOF_ERR_BAD_ARG, "null argument");
  *n_solutions = decompose_homography(H9, R, t, normals);
  return MOF_OK;
}

int mof_geom_get_rt(const double* shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                    const mof_geom_rt_params* params, int shifted_pts_thr, double* out_rot_tran, int* status,
                    uint8_t* inlier_mask, double* homography) try {
  int rc = check_layout(layout);
  if (rc) return rc;
  rc = check_camera(cam);
  if (rc) return rc;
  if (!shifts_xy || !params || !out_rot_tran || !status) return mof::capi_fail(MOF_ERR_BAD_ARG, "null argument");
  Layout L;
  Camera c;
  RtParams p;
  std::memcpy(&L, layout, sizeof(L));
  std::memcpy(&c, cam, sizeof(c));
  std::memcpy(&p, params, sizeof(p));
  const double ident[7] = {0, 0, 0, 1, 0, 0, 0};
  double res[7];
  std::memcpy(res, ident, sizeof(res));
  if (homography)
    for (int k = 0; k < 9; ++k) homography[k] = 0.0;
  *status = get_rt_host(shifts_xy, L, c, p, shifted_pts_thr, res, inlier_mask, homography);
  std::memcpy(out_rot_tran, *status == kOk ? res : ident, sizeof(res));
  return MOF_OK;
} catch (const std::bad_alloc&) {
  return mof::capi_fail(MOF_ERR_NO_MEMORY, "mof_geom_get_rt: out of host memory");
}

int mof_geom_get_2dt(const double* shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                     const mof_geom_2dt_params* params, double* out_tran_diff, int* status) {
  int rc = check_layout(layout);
  if (rc) return rc;
  rc = check_camera(cam);
  if (rc) return rc;
  if (!shifts_xy || !params || !out_tran_diff || !status) return mof::capi_fail(MOF_ERR_BAD_ARG, "null argument");
  Layout L;
  Camera c;
  T2dParams p;
  std::memcpy(&L, layout, sizeof(L));
  std::memcpy(&c, cam, sizeof(c));
  std::memcpy(&p, params, sizeof(p));
  double res[6] = {0, 0, 0, 0, 0, 0};
  *status = get_2dt(shifts_xy, L, c, p, res);
  for (int k = 0; k < 6; ++k) out_tran_diff[k] = *status == kOk ? res[k] : 0.0;
  return MOF_OK;
}

int mof_geom_get_rt_batch_device(const double* d_shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                                 const mof_geom_rt_params* d_params, int n_pairs, int shifted_pts_thr, double* d_out,
                                 void* stream) {
  int rc = check_layout(layout);
  if (rc) return rc;
  rc = check_camera(cam);
  if (rc) return rc;
  if (n_pairs == 0) return MOF_OK;
  if (!d_shifts_xy || !d_params || !d_out || n_pairs < 0) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  Layout L;
  Camera c;
  std::memcpy(&L, layout, sizeof(L));
  std::memcpy(&c, cam, sizeof(c));
  DeviceOf dev(d_out);
  if (dev.err != hipSuccess) return mof::capi_fail(MOF_ERR_BAD_ARG, "d_out is not a device pointer: %s", hipGetErrorString(dev.err));
  const int total = L.grid_x * L.grid_y;
  size_t lds = (size_t)(4 * total + (total + 7) / 8) * sizeof(double);  // a, b (2 doubles per point each) + mask bytes
  const int wave_fit = total <= kWaveFitPoints;
  if (wave_fit) lds += ((size_t)total * kFitTab + 81 + 81 + 72 + 8) * sizeof(double);  // homography_fit_wave's work space
  hipLaunchKernelGGL(geom_get_rt_kernel, dim3((unsigned)n_pairs), dim3(64), lds, (hipStream_t)stream, d_shifts_xy, L, c,
                     reinterpret_cast<const RtParams*>(d_params), shifted_pts_thr, wave_fit, d_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "geom_get_rt_kernel: %s", hipGetErrorString(e));
  return MOF_OK;
}

int mof_geom_get_2dt_batch_device(const double* d_shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                                  const mof_geom_2dt_params* d_params, int n_pairs, double* d_out, void* stream) {
  int rc = check_layout(layout);
  if (rc) return rc;
  rc = check_camera(cam);
  if (rc) return rc;
  if (n_pairs == 0) return MOF_OK;
  if (!d_shifts_xy || !d_params || !d_out || n_pairs < 0) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  Layout L;
  Camera c;
  std::memcpy(&L, layout, sizeof(L));
  std::memcpy(&c, cam, sizeof(c));
  DeviceOf dev(d_out);
  if (dev.err != hipSuccess) return mof::capi_fail(MOF_ERR_BAD_ARG, "d_out is not a device pointer: %s", hipGetErrorString(dev.err));
  hipLaunchKernelGGL(geom_get_2dt_kernel, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     d_shifts_xy, L, c, reinterpret_cast<const T2dParams*>(d_params), n_pairs, d_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "geom_get_2dt_kernel: %s", hipGetErrorString(e));
  return MOF_OK;
}

}  // extern "C"
