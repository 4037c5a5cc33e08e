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

__global__ void __launch_bounds__(64) geom_get_rt_kernel(const double* __restrict__ shifts, Layout L, Camera cam,
                                                         const RtParams* __restrict__ params, int thr,
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
  if (lane < 8) o[lane] = lane == 3 ? 1.0 : 0.0;  // identity rotation, zero translation unless the call succeeds
  if (!finite_d(1.0 / p.dt)) {
    if (lane == 0) o[7] = (double)kBadDuration;
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
    if (lane == 0) o[7] = (double)kTooFewPoints;
    return;
  }
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
  if (lane == 0) {  // the sequential rest: a few hundred flops, not worth spreading over lanes
    int remaining = 0;
    for (int i = 0; i < n; ++i) remaining += mask[i];
    int status;
    if (remaining < thr) status = kTooFewInliers;
    else if (!found) status = kNoHomography;
    else {
      double H[9], res[7] = {0, 0, 0, 1, 0, 0, 0};
      if (!homography_fit(a, b, mask, n, H))
        for (int k = 0; k < 9; ++k) H[k] = best[k];
      status = pick_motion(H, p, res);
      if (status == kOk)
        for (int k = 0; k < 7; ++k) o[k] = res[k];
    }
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
  const int total = L.grid_x * L.grid_y;
  const size_t lds = (size_t)(4 * total + (total + 7) / 8) * sizeof(double);  // a, b (2 doubles per point each) + mask bytes
  hipLaunchKernelGGL(geom_get_rt_kernel, dim3((unsigned)n_pairs), dim3(64), lds, (hipStream_t)stream, d_shifts_xy, L, c,
                     reinterpret_cast<const RtParams*>(d_params), shifted_pts_thr, d_out);
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
  hipLaunchKernelGGL(geom_get_2dt_kernel, dim3((unsigned)((n_pairs + 63) / 64)), dim3(64), 0, (hipStream_t)stream,
                     d_shifts_xy, L, c, reinterpret_cast<const T2dParams*>(d_params), n_pairs, d_out);
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "geom_get_2dt_kernel: %s", hipGetErrorString(e));
  return MOF_OK;
}

}  // extern "C"
