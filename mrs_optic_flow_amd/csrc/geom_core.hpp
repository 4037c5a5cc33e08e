// geom_core.hpp -- the geometry tail behind the hot path, one implementation for host and device.
//
// N3 tail `OpticFlow::get2DT` (/root/reference/src/optic_flow.cpp:388-510) and N1
// `OpticFlow::getRT` (:515-774): per-patch pixel shifts -> camera-frame velocity. In the reference these run on the
// host per frame on <= 256 points, calling OpenCV calib3d (cv::undistortPoints :549-550, cv::findHomography(RANSAC,
// 0.01) :559, cv::decomposeHomographyMat :595) and tf2 (quaternion algebra :639-748). Neither library exists here,
// so their behaviour is restated from the published algorithms (PARITY UNPINNED, like the rest of the oracle):
//   * undistort_point      cv::undistortPoints, 5-coefficient Brown model, 5 fixed-point iterations (its default
//                          TermCriteria(MAX_ITER, 5, 0.01)), no R / P  -> normalised coordinates;
//   * decompose_homography cv::decomposeHomographyMat with K = I: HomographyDecompInria (Malis & Vargas), scale
//                          removed by the middle singular value, pure-rotation shortcut |H'H - I|_inf < 1e-3;
//   * Quat / tf2_*         tf2::Quaternion / Matrix3x3 / Transform members as used by getRT;
//   * RANSAC               cv::findHomography's structure (4-point minimal sets, forward reprojection error against
//                          threshold^2, adaptive iteration count at confidence 0.995 / 2000 iterations, final
//                          normalised-DLT fit on the consensus set + 10 Levenberg-Marquardt steps) with an OWN
//                          documented counter-based sampler -- cv::RNG's sequence is not reproduced, so the sampled
//                          sets differ from OpenCV's while the consensus set on well-posed data does not.
// All arithmetic is fp64 and sequential per frame pair except the RANSAC hypotheses, which are independent by
// construction (hypothesis k depends only on (seed, k)): the host evaluates them in order, the device kernel 64 at a
// time, one per lane, and both replay the same in-order acceptance scan, so they pick the same model.
#pragma once

#include <float.h>
#include <math.h>
#include <stdint.h>

#if defined(__HIPCC__)
#define MOF_HD __host__ __device__ inline
#else
#define MOF_HD inline
#endif

namespace mof {
namespace geom {

constexpr int kMaxPoints = 1024;             // patches per frame the tail accepts (c4 has 256)
constexpr double kRansacThreshold = 0.01;    // optic_flow.cpp:559
constexpr double kRansacConfidence = 0.995;  // cv::findHomography default
constexpr int kRansacMaxIters = 2000;        // cv::findHomography default
constexpr int kRansacBatch = 64;             // hypotheses per round (one per lane on the device)
constexpr uint64_t kRansacSeed = 0x5EED0F10C0FFEEull;
constexpr double kPi = 3.14159265358979323846;

// getRT / get2DT status (0 = the reference returns true)
enum Status {
  kOk = 0,
  kBadDuration = 1,        // !isfinite(1 / dt)                                   :516-519, :393-396
  kTooFewPoints = 2,       // valid points < shifted_pts_thr (getRT :544-547) / < 1 (get2DT :425-429)
  kTooFewInliers = 3,      // remaining < shifted_pts_thr after RANSAC            :575-578
  kAngleTooLarge = 4,      // best solution differs from the IMU by > pi/4        :682-685
  kSingleNoMatch = 5,      // one solution, none accepted                         :725-728
  kSingleNonFinite = 6,    // one solution with NaN/Inf                           :745-751
  kUnclassified = 7,       // :769-771
  kNoHomography = 8,       // fewer than 4 points or no valid model (cv::findHomography would return an empty Mat
                           // and cv::decomposeHomographyMat would throw)
  kNoPoints = 9            // get2DT: shifts.size() < 1                           :389-392
};

struct Camera {  // camMatrix_ / distCoeffs_ as the node fills them (optic_flow.cpp:1511-1522)
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
};

struct Layout {  // patch centres: (origin + i * stride + patch / 2); the node's tiling is origin 0, stride = patch (:538-539)
  int grid_x, grid_y, origin_x, origin_y, stride_x, stride_y, patch;
};

struct RtParams {        // per frame pair
  double height;         // uav_height_curr                                          (optic_flow.cpp:1719)
  double dt;             // dur_.toSec()
  double ul_corner_x;    // ulCorner.x: camMatrixLocal(0, 2) -= ulCorner.x           (:521-522)
  double ang_rate_q[4];  // angular_rate_tf_ as a quaternion (x, y, z, w)            (:1314: setRPY of the gyro rates)
  double c2b_q[4];       // transformCam2Base_ rotation (x, y, z, w)                 (:600-601)
  double c2b_t[3];       // transformCam2Base_ translation: `tempTfC2B * axis` adds it (:643)
};

struct T2dParams {       // per frame pair, get2DT
  double height;         // uav_height_curr / (cos(pitch) cos(roll)), formed by the caller (optic_flow.cpp:1780)
  double dt;
  double roll_rate, pitch_rate;  // imu_roll_rate_, imu_pitch_rate_ (:481-482)
  double cam_yaw;        // cam_yaw_ (:484)
};

// ---- small dense helpers ---------------------------------------------------------------------------------------

MOF_HD bool finite_d(double v) { return v == v && v - v == 0.0; }

// Gaussian elimination with partial pivoting on an N x N system, augmented matrix a[N][N+1] (row-major). false if singular.
template <int N>
MOF_HD bool solve_linear(double* a, double* x) {
  for (int c = 0; c < N; ++c) {
    int piv = c;
    double best = fabs(a[c * (N + 1) + c]);
    for (int r = c + 1; r < N; ++r) {
      const double v = fabs(a[r * (N + 1) + c]);
      if (v > best) best = v, piv = r;
    }
    if (!(best > 1e-300)) return false;
    if (piv != c)
      for (int k = c; k <= N; ++k) {
        const double t = a[c * (N + 1) + k];
        a[c * (N + 1) + k] = a[piv * (N + 1) + k];
        a[piv * (N + 1) + k] = t;
      }
    const double inv = 1.0 / a[c * (N + 1) + c];
    for (int r = c + 1; r < N; ++r) {
      const double f = a[r * (N + 1) + c] * inv;
      if (f != 0.0)
        for (int k = c; k <= N; ++k) a[r * (N + 1) + k] -= f * a[c * (N + 1) + k];
    }
  }
  for (int r = N - 1; r >= 0; --r) {
    double s = a[r * (N + 1) + N];
    for (int k = r + 1; k < N; ++k) s -= a[r * (N + 1) + k] * x[k];
    x[r] = s / a[r * (N + 1) + r];
  }
  return true;
}

// solve_linear with every index a compile-time constant (loops fully unrolled, the pivot row brought up by conditional
// element swaps): the same pivots, the same operations in the same order -- bit-identical results -- but the matrix
// stays in registers. On the device the indexed form lives in scratch memory (one round of 64 RANSAC hypotheses: 200 k
// cycles, nearly all of them scratch round trips).
template <int N>
MOF_HD bool solve_linear_unrolled(double* a, double* x) {
#pragma unroll
  for (int c = 0; c < N; ++c) {
    int piv = c;
    double best = fabs(a[c * (N + 1) + c]);
#pragma unroll
    for (int r = c + 1; r < N; ++r) {
      const double v = fabs(a[r * (N + 1) + c]);
      if (v > best) best = v, piv = r;
    }
    if (!(best > 1e-300)) return false;
#pragma unroll
    for (int r = c + 1; r < N; ++r) {
      const bool sw = piv == r;
#pragma unroll
      for (int k = c; k <= N; ++k) {
        const double u = a[c * (N + 1) + k], w = a[r * (N + 1) + k];
        a[c * (N + 1) + k] = sw ? w : u;
        a[r * (N + 1) + k] = sw ? u : w;
      }
    }
    const double inv = 1.0 / a[c * (N + 1) + c];
#pragma unroll
    for (int r = c + 1; r < N; ++r) {
      const double f = a[r * (N + 1) + c] * inv;
      if (f != 0.0) {
#pragma unroll
        for (int k = c; k <= N; ++k) a[r * (N + 1) + k] -= f * a[c * (N + 1) + k];
      }
    }
  }
#pragma unroll
  for (int r = N - 1; r >= 0; --r) {
    double s = a[r * (N + 1) + N];
#pragma unroll
    for (int k = r + 1; k < N; ++k) s -= a[r * (N + 1) + k] * x[k];
    x[r] = s / a[r * (N + 1) + r];
  }
  return true;
}

// Cyclic Jacobi eigen-decomposition of a symmetric N x N matrix (a is destroyed; its diagonal ends as the
// eigenvalues, v holds the eigenvectors as COLUMNS). Fixed sweep order: deterministic on host and device.
template <int N>
MOF_HD void jacobi_eigen(double* a, double* v) {
  for (int i = 0; i < N; ++i)
    for (int j = 0; j < N; ++j) v[i * N + j] = i == j ? 1.0 : 0.0;
  for (int sweep = 0; sweep < 60; ++sweep) {
    double off = 0.0, diag = 0.0;
    for (int i = 0; i < N; ++i) {
      diag += a[i * N + i] * a[i * N + i];
      for (int j = i + 1; j < N; ++j) off += a[i * N + j] * a[i * N + j];
    }
    if (!(off > 1e-30 * diag) || off == 0.0) break;
    for (int p = 0; p < N - 1; ++p)
      for (int q = p + 1; q < N; ++q) {
        const double apq = a[p * N + q];
        if (apq == 0.0) continue;
        const double theta = (a[q * N + q] - a[p * N + p]) / (2.0 * apq);
        const double t = (theta >= 0.0 ? 1.0 : -1.0) / (fabs(theta) + sqrt(theta * theta + 1.0));
        const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
        for (int k = 0; k < N; ++k) {
          const double akp = a[k * N + p], akq = a[k * N + q];
          a[k * N + p] = c * akp - s * akq;
          a[k * N + q] = s * akp + c * akq;
        }
        for (int k = 0; k < N; ++k) {
          const double apk = a[p * N + k], aqk = a[q * N + k];
          a[p * N + k] = c * apk - s * aqk;
          a[q * N + k] = s * apk + c * aqk;
        }
        for (int k = 0; k < N; ++k) {
          const double vkp = v[k * N + p], vkq = v[k * N + q];
          v[k * N + p] = c * vkp - s * vkq;
          v[k * N + q] = s * vkp + c * vkq;
        }
      }
  }
}

MOF_HD void mat3_mul(const double* a, const double* b, double* c) {
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) c[i * 3 + j] = a[i * 3 + 0] * b[0 * 3 + j] + a[i * 3 + 1] * b[1 * 3 + j] + a[i * 3 + 2] * b[2 * 3 + j];
}

MOF_HD double mat3_det(const double* m) {
  return m[0] * (m[4] * m[8] - m[5] * m[7]) - m[1] * (m[3] * m[8] - m[5] * m[6]) + m[2] * (m[3] * m[7] - m[4] * m[6]);
}

// ---- cv::undistortPoints -----------------------------------------------------------------------------------------

// One point, pixel (u, v) -> normalised (x, y). cx_local = cx - ulCorner.x (optic_flow.cpp:522). Distortion vector
// (k1, k2, p1, p2, k3); OpenCV's k[5..11] are zero here, so its rational numerator is exactly 1.
MOF_HD void undistort_point(const Camera& c, double cx_local, double u, double v, double* ox, double* oy) {
  const double ifx = 1.0 / c.fx, ify = 1.0 / c.fy;
  double x = (u - cx_local) * ifx, y = (v - c.cy) * ify;
  const double x0 = x, y0 = y;
  for (int j = 0; j < 5; ++j) {
    const double r2 = x * x + y * y;
    const double icdist = 1.0 / (1 + ((c.k3 * r2 + c.k2) * r2 + c.k1) * r2);
    if (icdist < 0) {  // OpenCV >= 4.1 gives up on a sign flip of the radial factor
      x = x0;
      y = y0;
      break;
    }
    const double dx = 2 * c.p1 * x * y + c.p2 * (r2 + 2 * x * x);
    const double dy = c.p1 * (r2 + 2 * y * y) + 2 * c.p2 * x * y;
    x = (x0 - dx) * icdist;
    y = (y0 - dy) * icdist;
  }
  *ox = x;
  *oy = y;
}

// ---- tf2 (Bullet LinearMath) members used by getRT ---------------------------------------------------------------

struct Quat {
  double x, y, z, w;
};

MOF_HD double tf2_acos(double v) { return acos(v < -1.0 ? -1.0 : (v > 1.0 ? 1.0 : v)); }  // tf2Acos clamps

// tf2::Matrix3x3::getRotation (m row-major)
MOF_HD Quat tf2_matrix_to_quat(const double* m) {
  const double trace = m[0] + m[4] + m[8];
  double t[4];
  if (trace > 0.0) {
    double s = sqrt(trace + 1.0);
    t[3] = s * 0.5;
    s = 0.5 / s;
    t[0] = (m[7] - m[5]) * s;
    t[1] = (m[2] - m[6]) * s;
    t[2] = (m[3] - m[1]) * s;
  } else {
    const int i = m[0] < m[4] ? (m[4] < m[8] ? 2 : 1) : (m[0] < m[8] ? 2 : 0);
    const int j = (i + 1) % 3, k = (i + 2) % 3;
    double s = sqrt(m[i * 3 + i] - m[j * 3 + j] - m[k * 3 + k] + 1.0);
    t[i] = s * 0.5;
    s = 0.5 / s;
    t[3] = (m[k * 3 + j] - m[j * 3 + k]) * s;
    t[j] = (m[j * 3 + i] + m[i * 3 + j]) * s;
    t[k] = (m[k * 3 + i] + m[i * 3 + k]) * s;
  }
  return Quat{t[0], t[1], t[2], t[3]};
}

MOF_HD double tf2_quat_angle(const Quat& q) { return 2.0 * tf2_acos(q.w); }  // Quaternion::getAngle

MOF_HD void tf2_quat_axis(const Quat& q, double* a) {  // Quaternion::getAxis
  const double s2 = 1.0 - q.w * q.w;
  if (s2 < 10.0 * DBL_EPSILON) {
    a[0] = 1.0;
    a[1] = 0.0;
    a[2] = 0.0;
    return;
  }
  const double s = sqrt(s2);
  a[0] = q.x / s;
  a[1] = q.y / s;
  a[2] = q.z / s;
}

MOF_HD Quat tf2_quat_from_axis_angle(const double* axis, double angle) {  // Quaternion(axis, angle) -> setRotation
  const double d = sqrt(axis[0] * axis[0] + axis[1] * axis[1] + axis[2] * axis[2]);
  const double s = sin(angle * 0.5) / d;
  return Quat{axis[0] * s, axis[1] * s, axis[2] * s, cos(angle * 0.5)};
}

MOF_HD double tf2_quat_angle_between(const Quat& a, const Quat& b) {  // Quaternion::angle(q)
  const double s = sqrt((a.x * a.x + a.y * a.y + a.z * a.z + a.w * a.w) * (b.x * b.x + b.y * b.y + b.z * b.z + b.w * b.w));
  return tf2_acos((a.x * b.x + a.y * b.y + a.z * b.z + a.w * b.w) / s);
}

MOF_HD void tf2_quat_to_matrix(const Quat& q, double* m) {  // Matrix3x3::setRotation
  const double d = q.x * q.x + q.y * q.y + q.z * q.z + q.w * q.w;
  const double s = 2.0 / d;
  const double xs = q.x * s, ys = q.y * s, zs = q.z * s;
  const double wx = q.w * xs, wy = q.w * ys, wz = q.w * zs;
  const double xx = q.x * xs, xy = q.x * ys, xz = q.x * zs;
  const double yy = q.y * ys, yz = q.y * zs, zz = q.z * zs;
  m[0] = 1.0 - (yy + zz);
  m[1] = xy - wz;
  m[2] = xz + wy;
  m[3] = xy + wz;
  m[4] = 1.0 - (xx + zz);
  m[5] = yz - wx;
  m[6] = xz - wy;
  m[7] = yz + wx;
  m[8] = 1.0 - (xx + yy);
}

MOF_HD Quat tf2_quat_from_rpy(double roll, double pitch, double yaw) {  // Quaternion::setRPY
  const double hy = yaw * 0.5, hp = pitch * 0.5, hr = roll * 0.5;
  const double cy = cos(hy), sy = sin(hy), cp = cos(hp), sp = sin(hp), cr = cos(hr), sr = sin(hr);
  return Quat{sr * cp * cy - cr * sp * sy, cr * sp * cy + sr * cp * sy, cr * cp * sy - sr * sp * cy, cr * cp * cy + sr * sp * sy};
}

// tf2::Transform(rotation q, origin t) * v
MOF_HD void tf2_transform_apply(const Quat& q, const double* t, const double* v, double* out) {
  double m[9];
  tf2_quat_to_matrix(q, m);
  const double r0 = m[0] * v[0] + m[1] * v[1] + m[2] * v[2] + t[0];
  const double r1 = m[3] * v[0] + m[4] * v[1] + m[5] * v[2] + t[1];
  const double r2 = m[6] * v[0] + m[7] * v[1] + m[8] * v[2] + t[2];
  out[0] = r0;
  out[1] = r1;
  out[2] = r2;
}

// ---- homography --------------------------------------------------------------------------------------------------

// counter-based generator: hypothesis k of seed s draws from its own stream
MOF_HD uint64_t splitmix64(uint64_t x) {
  x += 0x9E3779B97F4A7C15ull;
  x = (x ^ (x >> 30)) * 0xBF58476D1CE4E5B9ull;
  x = (x ^ (x >> 27)) * 0x94D049BB133111EBull;
  return x ^ (x >> 31);
}

MOF_HD double cross2(const double* p, int i, int j, int k) {
  return (p[2 * j] - p[2 * i]) * (p[2 * k + 1] - p[2 * i + 1]) - (p[2 * j + 1] - p[2 * i + 1]) * (p[2 * k] - p[2 * i]);
}

// Minimal-set homography b ~ H a through 4 correspondences (h33 = 1). idx: the 4 point indices.
MOF_HD bool homography_4pt(const double* a, const double* b, const int* idx, double* H) {
  double m[8 * 9];
  for (int k = 0; k < 4; ++k) {
    const double x = a[2 * idx[k]], y = a[2 * idx[k] + 1], u = b[2 * idx[k]], v = b[2 * idx[k] + 1];
    double* r0 = m + (2 * k) * 9;
    double* r1 = r0 + 9;
    r0[0] = x; r0[1] = y; r0[2] = 1; r0[3] = 0; r0[4] = 0; r0[5] = 0; r0[6] = -u * x; r0[7] = -u * y; r0[8] = u;
    r1[0] = 0; r1[1] = 0; r1[2] = 0; r1[3] = x; r1[4] = y; r1[5] = 1; r1[6] = -v * x; r1[7] = -v * y; r1[8] = v;
  }
  double h[8];
#ifdef __HIP_DEVICE_COMPILE__
  if (!solve_linear_unrolled<8>(m, h)) return false;
#else
  if (!solve_linear<8>(m, h)) return false;
#endif
  for (int k = 0; k < 8; ++k) {
    if (!finite_d(h[k])) return false;
    H[k] = h[k];
  }
  H[8] = 1.0;
  return true;
}

MOF_HD bool is_inlier(const double* H, double x, double y, double u, double v, double thr2) {
  const double w = H[6] * x + H[7] * y + H[8];
  if (!(fabs(w) > DBL_EPSILON)) return false;
  const double iw = 1.0 / w;
  const double dx = (H[0] * x + H[1] * y + H[2]) * iw - u, dy = (H[3] * x + H[4] * y + H[5]) * iw - v;
  return dx * dx + dy * dy <= thr2;
}

// Hypothesis `iter`: up to 10 attempts to draw 4 distinct, non-degenerate, orientation-consistent correspondences,
// fit the minimal model, count its inliers. Returns the count (0 = no valid model) and H.
MOF_HD int ransac_hypothesis(const double* a, const double* b, int n, uint64_t seed, int iter, double thr2, double* H) {
  uint64_t st = splitmix64(seed ^ ((uint64_t)(uint32_t)iter * 0xD1342543DE82EF95ull));
  for (int attempt = 0; attempt < 10; ++attempt) {
    int idx[4];
    for (int k = 0; k < 4; ++k) {
      for (;;) {
        st = splitmix64(st);
        const int cand = (int)((st >> 11) % (uint64_t)n);
        bool dup = false;
        for (int j = 0; j < k; ++j) dup = dup || idx[j] == cand;
        if (!dup) {
          idx[k] = cand;
          break;
        }
      }
    }
    // degenerate or mirrored minimal sets (cf. HomographyEstimatorCallback::checkSubset)
    bool ok = true;
    const int tri[4][3] = {{0, 1, 2}, {0, 1, 3}, {0, 2, 3}, {1, 2, 3}};
    for (int t = 0; t < 4 && ok; ++t) {
      const double ca = cross2(a, idx[tri[t][0]], idx[tri[t][1]], idx[tri[t][2]]);
      const double cb = cross2(b, idx[tri[t][0]], idx[tri[t][1]], idx[tri[t][2]]);
      if (!(fabs(ca) > 1e-12) || !(fabs(cb) > 1e-12) || (ca > 0) != (cb > 0)) ok = false;
    }
    if (!ok) continue;
    if (!homography_4pt(a, b, idx, H)) continue;
    int cnt = 0;
    for (int i = 0; i < n; ++i) cnt += is_inlier(H, a[2 * i], a[2 * i + 1], b[2 * i], b[2 * i + 1], thr2) ? 1 : 0;
    return cnt;
  }
  return 0;
}

// cv::RANSACUpdateNumIters(confidence, outlier ratio, 4 model points, current maximum)
MOF_HD int ransac_update_iters(double p, double ep, int max_iters) {
  p = p < 0 ? 0 : (p > 1 ? 1 : p);
  ep = ep < 0 ? 0 : (ep > 1 ? 1 : ep);
  double num = 1 - p;
  if (num < DBL_MIN) num = DBL_MIN;
  const double q = 1 - ep;
  double denom = 1 - q * q * q * q;
  if (denom < DBL_MIN) return 0;
  num = log(num);
  denom = log(denom);
  if (denom >= 0 || -num >= max_iters * (-denom)) return max_iters;
  return (int)nearbyint(num / denom);  // cvRound
}

// State of the in-order acceptance scan shared by host and device.
struct RansacScan {
  int best_count, best_iter, niters;
};

// Feed hypothesis `iter` (count) in order; returns false once iter >= niters (the loop would have ended before it).
MOF_HD bool ransac_accept(RansacScan* s, int iter, int count, int n) {
  if (iter >= s->niters) return false;
  const int best = s->best_count > 3 ? s->best_count : 3;  // a model must explain more than its own 4... at least 4 points
  if (count > best) {
    s->best_count = count;
    s->best_iter = iter;
    s->niters = ransac_update_iters(kRansacConfidence, (double)(n - count) / n, s->niters);
  }
  return true;
}

// Least-squares homography through the points with mask[i] != 0: cv::findHomography's normalised DLT
// (HomographyEstimatorCallback::runKernel) followed by 10 Levenberg-Marquardt steps on the forward reprojection error
// (HomographyRefineCallback's residual and Jacobian; own damping schedule). false when the fit is degenerate.
MOF_HD bool homography_fit(const double* a, const double* b, const unsigned char* mask, int n, double* H) {
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
  double LtL[81];
  for (int k = 0; k < 81; ++k) LtL[k] = 0.0;
  for (int i = 0; i < n; ++i)
    if (mask[i]) {
      const double x = (b[2 * i] - cmx) * smx, y = (b[2 * i + 1] - cmy) * smy;
      const double X = (a[2 * i] - cMx) * sMx, Y = (a[2 * i + 1] - cMy) * sMy;
      const double Lx[9] = {X, Y, 1, 0, 0, 0, -x * X, -x * Y, -x};
      const double Ly[9] = {0, 0, 0, X, Y, 1, -y * X, -y * Y, -y};
      for (int j = 0; j < 9; ++j)
        for (int k = j; k < 9; ++k) LtL[j * 9 + k] += Lx[j] * Lx[k] + Ly[j] * Ly[k];
    }
  for (int j = 0; j < 9; ++j)
    for (int k = 0; k < j; ++k) LtL[j * 9 + k] = LtL[k * 9 + j];
  double V[81];
  jacobi_eigen<9>(LtL, V);
  int lo = 0;
  for (int k = 1; k < 9; ++k)
    if (LtL[k * 9 + k] < LtL[lo * 9 + lo]) lo = k;
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

  // Levenberg-Marquardt on (h0..h7), h8 = 1
  double lambda = 1e-3;
  double S = 0.0;
  for (int i = 0; i < n; ++i)
    if (mask[i]) {
      const double X = a[2 * i], Y = a[2 * i + 1];
      const double ww = 1.0 / (h[6] * X + h[7] * Y + 1.0);
      const double ex = (h[0] * X + h[1] * Y + h[2]) * ww - b[2 * i], ey = (h[3] * X + h[4] * Y + h[5]) * ww - b[2 * i + 1];
      S += ex * ex + ey * ey;
    }
  for (int it = 0; it < 10; ++it) {
    double A[64], g[8];
    for (int k = 0; k < 64; ++k) A[k] = 0.0;
    for (int k = 0; k < 8; ++k) g[k] = 0.0;
    for (int i = 0; i < n; ++i)
      if (mask[i]) {
        const double X = a[2 * i], Y = a[2 * i + 1];
        const double ww = 1.0 / (h[6] * X + h[7] * Y + 1.0);
        const double xi = (h[0] * X + h[1] * Y + h[2]) * ww, yi = (h[3] * X + h[4] * Y + h[5]) * ww;
        const double ex = xi - b[2 * i], ey = yi - b[2 * i + 1];
        const double Jx[8] = {X * ww, Y * ww, ww, 0, 0, 0, -X * ww * xi, -Y * ww * xi};
        const double Jy[8] = {0, 0, 0, X * ww, Y * ww, ww, -X * ww * yi, -Y * ww * yi};
        for (int j = 0; j < 8; ++j) {
          g[j] += Jx[j] * ex + Jy[j] * ey;
          for (int k = j; k < 8; ++k) A[j * 8 + k] += Jx[j] * Jx[k] + Jy[j] * Jy[k];
        }
      }
    for (int j = 0; j < 8; ++j)
      for (int k = 0; k < j; ++k) A[j * 8 + k] = A[k * 8 + j];
    bool accepted = false;
    for (int tries = 0; tries < 6 && !accepted; ++tries) {
      double m[8 * 9], d[8], hn[8];
      for (int j = 0; j < 8; ++j) {
        for (int k = 0; k < 8; ++k) m[j * 9 + k] = A[j * 8 + k];
        m[j * 9 + j] += lambda * A[j * 8 + j];
        m[j * 9 + 8] = -g[j];
      }
      if (solve_linear<8>(m, d)) {
        for (int k = 0; k < 8; ++k) hn[k] = h[k] + d[k];
        double Sn = 0.0;
        for (int i = 0; i < n; ++i)
          if (mask[i]) {
            const double X = a[2 * i], Y = a[2 * i + 1];
            const double ww = 1.0 / (hn[6] * X + hn[7] * Y + 1.0);
            const double ex = (hn[0] * X + hn[1] * Y + hn[2]) * ww - b[2 * i], ey = (hn[3] * X + hn[4] * Y + hn[5]) * ww - b[2 * i + 1];
            Sn += ex * ex + ey * ey;
          }
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

// ---- cv::decomposeHomographyMat(H, I) ------------------------------------------------------------------------------
// ATTRIBUTION. This block (signd, opposite_of_minor, rmat_from_tstar_n, decompose_homography) is a restatement, written from
// memory, of OpenCV's modules/calib3d/src/homography_decomp.cpp -- class HomographyDecompInria, the analytical method of
// E. Malis and M. Vargas, "Deeper understanding of the homography decomposition for vision-based control", INRIA RR-6303
// (2007) -- including that file's variable naming (M00.., rtM00.., e12.., npa / npb, ESii, r_2, nt_2). OpenCV is
// third-party code under the Apache-2.0 (4.5+) / 3-clause BSD (3.x, 4.0-4.4) licence, Copyright (C) 2014 Samson Yilma and
// the OpenCV contributors; it is NOT part of /root/reference, which only calls it (optic_flow.cpp:595). Nothing here was
// checked against a real OpenCV build (parity unpinned).

MOF_HD int signd(double x) { return x >= 0 ? 1 : -1; }

MOF_HD double opposite_of_minor(const double* M, int row, int col) {
  const int x1 = col == 0 ? 1 : 0, x2 = col == 2 ? 1 : 2, y1 = row == 0 ? 1 : 0, y2 = row == 2 ? 1 : 2;
  return M[y1 * 3 + x2] * M[y2 * 3 + x1] - M[y1 * 3 + x1] * M[y2 * 3 + x2];
}

// R = Hn (I - (2/v) t* n'), negated when det < 0
MOF_HD void rmat_from_tstar_n(const double* Hn, const double* ts, const double* nn, double v, double* R) {
  double T[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) T[i * 3 + j] = (i == j ? 1.0 : 0.0) - (2 / v) * ts[i] * nn[j];
  mat3_mul(Hn, T, R);
  if (mat3_det(R) < 0)
    for (int k = 0; k < 9; ++k) R[k] *= -1;
}

// Returns the number of solutions (1 or 4; 0 on a degenerate H). R: [4][9] row-major, t: [4][3], nrm: [4][3].
MOF_HD int decompose_homography(const double* H, double* R, double* t, double* nrm) {
  // removeScale(): divide by the middle singular value
  double HtH[9], V[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) HtH[i * 3 + j] = H[0 * 3 + i] * H[0 * 3 + j] + H[1 * 3 + i] * H[1 * 3 + j] + H[2 * 3 + i] * H[2 * 3 + j];
  jacobi_eigen<3>(HtH, V);
  double e0 = HtH[0], e1 = HtH[4], e2 = HtH[8];
  // middle of three
  const double mid = e0 > e1 ? (e1 > e2 ? e1 : (e0 > e2 ? e2 : e0)) : (e0 > e2 ? e0 : (e1 > e2 ? e2 : e1));
  if (!(mid > 0.0)) return 0;
  const double sc = 1.0 / sqrt(mid);
  double Hn[9];
  for (int k = 0; k < 9; ++k) Hn[k] = H[k] * sc;
  // S = Hn' Hn - I
  double S[9];
  for (int i = 0; i < 3; ++i)
    for (int j = 0; j < 3; ++j) S[i * 3 + j] = Hn[0 * 3 + i] * Hn[0 * 3 + j] + Hn[1 * 3 + i] * Hn[1 * 3 + j] + Hn[2 * 3 + i] * Hn[2 * 3 + j];
  S[0] -= 1.0;
  S[4] -= 1.0;
  S[8] -= 1.0;
  double ninf = 0.0;  // cv::norm(S, NORM_INF) on a Matx: the largest absolute element
  for (int k = 0; k < 9; ++k) ninf = fabs(S[k]) > ninf ? fabs(S[k]) : ninf;
  if (ninf < 0.001) {
    for (int k = 0; k < 9; ++k) R[k] = Hn[k];
    for (int k = 0; k < 3; ++k) t[k] = 0.0, nrm[k] = 0.0;
    return 1;
  }
  const double M00 = opposite_of_minor(S, 0, 0), M11 = opposite_of_minor(S, 1, 1), M22 = opposite_of_minor(S, 2, 2);
  const double rtM00 = sqrt(M00), rtM11 = sqrt(M11), rtM22 = sqrt(M22);
  const double M01 = opposite_of_minor(S, 0, 1), M12 = opposite_of_minor(S, 1, 2), M02 = opposite_of_minor(S, 0, 2);
  const int e12 = signd(M12), e02 = signd(M02), e01 = signd(M01);
  const double nS00 = fabs(S[0]), nS11 = fabs(S[4]), nS22 = fabs(S[8]);
  int indx = 0;
  if (nS00 < nS11) {
    indx = 1;
    if (nS11 < nS22) indx = 2;
  } else if (nS00 < nS22) {
    indx = 2;
  }
  double npa[3], npb[3];
  if (indx == 0) {
    npa[0] = S[0];               npb[0] = S[0];
    npa[1] = S[1] + rtM22;       npb[1] = S[1] - rtM22;
    npa[2] = S[2] + e12 * rtM11; npb[2] = S[2] - e12 * rtM11;
  } else if (indx == 1) {
    npa[0] = S[1] + rtM22;       npb[0] = S[1] - rtM22;
    npa[1] = S[4];               npb[1] = S[4];
    npa[2] = S[5] - e02 * rtM00; npb[2] = S[5] + e02 * rtM00;
  } else {
    npa[0] = S[2] + e01 * rtM11; npb[0] = S[2] - e01 * rtM11;
    npa[1] = S[5] + rtM00;       npb[1] = S[5] - rtM00;
    npa[2] = S[8];               npb[2] = S[8];
  }
  const double traceS = S[0] + S[4] + S[8];
  const double v = 2.0 * sqrt(1 + traceS - M00 - M11 - M22);
  const double ESii = signd(S[indx * 3 + indx]);
  const double r_2 = 2 + traceS + v, nt_2 = 2 + traceS - v;
  const double r = sqrt(r_2), n_t = sqrt(nt_2);
  const double la = sqrt(npa[0] * npa[0] + npa[1] * npa[1] + npa[2] * npa[2]);
  const double lb = sqrt(npb[0] * npb[0] + npb[1] * npb[1] + npb[2] * npb[2]);
  double na[3], nb[3], ta_s[3], tb_s[3];
  for (int k = 0; k < 3; ++k) na[k] = npa[k] / la, nb[k] = npb[k] / lb;
  const double half_nt = 0.5 * n_t, esii_t_r = ESii * r;
  for (int k = 0; k < 3; ++k) {
    ta_s[k] = half_nt * (esii_t_r * nb[k] - n_t * na[k]);
    tb_s[k] = half_nt * (esii_t_r * na[k] - n_t * nb[k]);
  }
  double Ra[9], Rb[9], ta[3], tb[3];
  rmat_from_tstar_n(Hn, ta_s, na, v, Ra);
  rmat_from_tstar_n(Hn, tb_s, nb, v, Rb);
  for (int i = 0; i < 3; ++i) {
    ta[i] = Ra[i * 3] * ta_s[0] + Ra[i * 3 + 1] * ta_s[1] + Ra[i * 3 + 2] * ta_s[2];
    tb[i] = Rb[i * 3] * tb_s[0] + Rb[i * 3 + 1] * tb_s[1] + Rb[i * 3 + 2] * tb_s[2];
  }
  for (int k = 0; k < 9; ++k) R[k] = Ra[k], R[9 + k] = Ra[k], R[18 + k] = Rb[k], R[27 + k] = Rb[k];
  for (int k = 0; k < 3; ++k) {
    t[k] = ta[k];       nrm[k] = na[k];
    t[3 + k] = -ta[k];  nrm[3 + k] = -na[k];
    t[6 + k] = tb[k];   nrm[6 + k] = nb[k];
    t[9 + k] = -tb[k];  nrm[9 + k] = -nb[k];
  }
  return 4;
}

// ---- getRT after the homography (optic_flow.cpp:594-771) -----------------------------------------------------------
// out: o_rot (x, y, z, w), o_tran (x, y, z). Returns a Status.
MOF_HD int pick_motion(const double* H, const RtParams& p, double* out) {
  double R[36], t[12], nrm[12];
  const int solutions = decompose_homography(H, R, t, nrm);
  const Quat ang{p.ang_rate_q[0], p.ang_rate_q[1], p.ang_rate_q[2], p.ang_rate_q[3]};
  const Quat ang_inv{-ang.x, -ang.y, -ang.z, ang.w};
  const Quat c2b{p.c2b_q[0], p.c2b_q[1], p.c2b_q[2], p.c2b_q[3]};
  int best = -1;
  bool best_inverse = false;
  double best_diff = kPi;
  Quat best_q{0, 0, 0, 1};
  for (int i = 0; i < solutions; ++i) {
    // cvMat33ToTf2Mat33 transposes: output[k][j] = input(j, k)  (:76-85)
    double m[9];
    for (int j = 0; j < 3; ++j)
      for (int k = 0; k < 3; ++k) m[k * 3 + j] = R[i * 9 + j * 3 + k];
    const Quat q = tf2_matrix_to_quat(m);
    double axis[3], axis_b[3];
    tf2_quat_axis(q, axis);
    tf2_transform_apply(c2b, p.c2b_t, axis, axis_b);                         // tempTfC2B * axis  (:643)
    const Quat qb = tf2_quat_from_axis_angle(axis_b, tf2_quat_angle(q) / p.dt);
    const double plus = tf2_quat_angle_between(qb, ang), minus = tf2_quat_angle_between(qb, ang_inv);
    const double diff = plus < minus ? plus : minus;                          // :651-655
    const bool inverse = !(nrm[i * 3 + 2] < 0);                               // :657-660
    if (best_diff > diff) {                                                   // :664-669
      best_diff = diff;
      best = i;
      best_inverse = inverse;
      best_q = q;
    }
  }
  const double zero3[3] = {0, 0, 0};
  if (best != -1 && solutions > 1) {
    if (best_diff > kPi / 4) return kAngleTooLarge;                           // :682-685
    double axis[3];
    tf2_quat_axis(best_q, axis);
    const Quat o = tf2_quat_from_axis_angle(axis, tf2_quat_angle(best_q) / p.dt);  // :703
    const double inv_unit = best_inverse ? -1.0 : 1.0;                        // :719
    const double tv[3] = {inv_unit * t[best * 3], inv_unit * t[best * 3 + 1], inv_unit * t[best * 3 + 2]};
    double r[3];
    tf2_transform_apply(best_q, zero3, tv, r);
    const double idt = 1.0 / p.dt;                                            // tf2::Vector3::operator/ multiplies by 1/s
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
    out[4] = r[0] * p.height * idt; out[5] = r[1] * p.height * idt; out[6] = r[2] * p.height * idt;  // :720-722
    return kOk;
  }
  if (solutions == 1) {
    if (best == -1) return kSingleNoMatch;                                    // :725-728
    double axis[3];
    tf2_quat_axis(best_q, axis);
    const Quat o = tf2_quat_from_axis_angle(axis, tf2_quat_angle(best_q) / p.dt);  // :737
    double r[3];
    tf2_transform_apply(best_q, zero3, t, r);
    const double idt = 1.0 / p.dt;
    out[0] = o.x; out[1] = o.y; out[2] = o.z; out[3] = o.w;
    out[4] = r[0] * p.height * idt; out[5] = r[1] * p.height * idt; out[6] = r[2] * p.height * idt;  // :741
    for (int k = 0; k < 7; ++k)
      if (!finite_d(out[k])) return kSingleNonFinite;                         // :744-751
    return kOk;
  }
  return solutions == 0 ? kNoHomography : kUnclassified;                      // :769-771
}

// ---- get2DT (optic_flow.cpp:388-510, LONG_RANGE_RATIO 4) -----------------------------------------------------------
// shifts: [grid_y * grid_x][2], long-range vectors in quarter-resolution pixels; out: o_tran (3), o_tran_diff (3).
// The two cv::undistortPoints calls of :442-444 compute values the function never uses (undistShifts is formed from
// the RAW points, :451-454) and are not reproduced.
MOF_HD int get_2dt(const double* shifts, const Layout& L, const Camera& cam, const T2dParams& p, double* out) {
  const int total = L.grid_x * L.grid_y;
  if (total < 1) return kNoPoints;
  if (!finite_d(1.0 / p.dt)) return kBadDuration;
  bool have = false;
  double first[2] = {0, 0};
  for (int j = 0; j < L.grid_y && !have; ++j)
    for (int i = 0; i < L.grid_x && !have; ++i) {
      const double sx = shifts[2 * (i + L.grid_x * j)], sy = shifts[2 * (i + L.grid_x * j) + 1];
      if (!finite_d(sx) || !finite_d(sy)) continue;                           // :406-409
      const int xi = L.origin_x + i * L.stride_x + L.patch / 2, yi = L.origin_y + j * L.stride_y + L.patch / 2;  // :411-412
      // undistShifts[0] = shiftedPts[0] - initialPts[0] = ((xi + s) - xi): the rounding of the sum is part of the result
      first[0] = ((double)xi + sx) - (double)xi;
      first[1] = ((double)yi + sy) - (double)yi;
      have = true;
    }
  if (!have) return kTooFewPoints;                                            // :425-429
  const double multiplier = 4;                                                // :473-477
  double ax = first[0], ay = first[1];                                        // avgShift = undistShifts[0], :471
  const double x_corr = -tan(p.roll_rate * p.dt) * cam.fx / multiplier;       // :481
  const double y_corr = tan(p.pitch_rate * p.dt) * cam.fy / multiplier;       // :482
  const double t_corr = sqrt(y_corr * y_corr + x_corr * x_corr);              // :483
  const double yaw_corr = atan2(y_corr, x_corr) + p.cam_yaw;                  // :484
  const double x_corr_cam = cos(yaw_corr) * t_corr, y_corr_cam = sin(yaw_corr) * t_corr;  // :485-486
  const double idt = 1.0 / p.dt;                                              // tf2::Vector3 / s == * (1 / s)
  ax += x_corr_cam;
  ay += y_corr_cam;                                                           // :489-490
  const double tx = ax * (p.height / cam.fx * multiplier), ty = ay * (p.height / cam.fy * multiplier);  // :491-492
  out[0] = -tx * idt;
  out[1] = -ty * idt;
  out[2] = -0.0 * idt;                                                        // :493-495
  ax += x_corr_cam;
  ay += y_corr_cam;                                                           // :499-500
  const double cx2 = ax * (p.height / cam.fx * multiplier), cy2 = ay * (p.height / cam.fy * multiplier);
  out[3] = -cx2 * idt - out[0];
  out[4] = -cy2 * idt - out[1];
  out[5] = -0.0 * idt - out[2];                                               // :505-507
  return kOk;
}

}  // namespace geom
}  // namespace mof
