// Drives the HOST side of the product library under AddressSanitizer + UndefinedBehaviorSanitizer without a device
// (tests/san/Makefile builds the library's objects with -Xarch_host -fsanitize=..., device code untouched):
//   * every geometry / configuration / argument-validation path of the C ABI that runs before a device is touched,
//   * every entry point on a null engine,
//   * the scale/rotation estimator's host-built tables (cv::logPolar maps in remap's fixed point, cubic / Lanczos4
//     weight tables), compared entry by entry with the oracle's tables,
//   * the geometry tail's host forms (getRT / get2DT and their stages).
// Where a device IS present the create() calls succeed and the engines are destroyed again.
#include <cmath>
#include <cstdio>
#include <cstring>
#include <vector>

#include "mof.h"
#include "mof_kernels.h"
extern "C" {
#include "oracle.h"
}

#define CHECK(cond) do { if (!(cond)) { std::fprintf(stderr, "check failed: %s (line %d): %s\n", #cond, __LINE__, mof_last_error()); return 1; } } while (0)

int main() {
  std::printf("%s, devices: %d\n", mof_version(), mof_device_count());
  // ---- FftMethod geometry (FftMethod.cpp:1706-1720) ----
  mof_fft_config fc;
  for (int fs = 2; fs < 700; fs += 37)
    for (int sps : {1, 32, 64, 120, 128, 481}) {
      CHECK(mof_fft_config_reference(&fc, fs, sps, 80.0) == MOF_OK);
      const int even = fs - (fs & 1);
      CHECK(fc.frame_width == even && fc.grid_x == fc.grid_y && fc.grid_x * fc.patch_size == even);
    }
  CHECK(mof_fft_config_reference(nullptr, 480, 120, 80) == MOF_ERR_BAD_ARG);
  CHECK(mof_fft_config_reference(&fc, 1, 120, 80) == MOF_ERR_BAD_ARG);
  mof_fft_engine* fe = nullptr;
  CHECK(mof_fft_config_reference(&fc, 480, 120, 80.0) == MOF_OK);
  mof_fft_config bad = fc;
  bad.patch_size = 1000;  // pads to 1000: beyond the planned transforms
  bad.frame_width = bad.frame_height = 4000;
  CHECK(mof_fft_create(&bad, &fe) == MOF_ERR_UNSUPPORTED && fe == nullptr && std::strlen(mof_last_error()) > 0);
  bad = fc; bad.patch_size = 62; bad.peak_model = MOF_PEAK_OCL;  // 62 = 2 * 31: no OpenCL plan in the reference either
  CHECK(mof_fft_create(&bad, &fe) == MOF_ERR_UNSUPPORTED && fe == nullptr);
  bad = fc; bad.patch_size = 1;
  CHECK(mof_fft_create(&bad, &fe) == MOF_ERR_BAD_ARG && fe == nullptr);
  bad = fc; bad.grid_x = 5;
  CHECK(mof_fft_create(&bad, &fe) == MOF_ERR_BAD_ARG);
  bad = fc; bad.max_px_speed = NAN;
  CHECK(mof_fft_create(&bad, &fe) == MOF_ERR_BAD_ARG);
  bad = fc; bad.peak_model = 9;
  CHECK(mof_fft_create(&bad, &fe) == MOF_ERR_BAD_ARG);
  bad = fc; bad.device = 1 << 20;
  { const int rc = mof_fft_create(&bad, &fe); CHECK(rc == MOF_ERR_BAD_ARG || rc == MOF_ERR_NO_DEVICE); }
  CHECK(mof_fft_create(nullptr, &fe) == MOF_ERR_BAD_ARG && mof_fft_create(&fc, nullptr) == MOF_ERR_BAD_ARG);
  { const int rc = mof_fft_create(&fc, &fe); CHECK(rc == MOF_OK || rc == MOF_ERR_NO_DEVICE); if (rc == MOF_OK) mof_fft_destroy(fe); }
  uint8_t px[16] = {0};
  double d2[2];
  int ninv;
  CHECK(mof_fft_set_prev(nullptr, px, 4) == MOF_ERR_NOT_INIT && mof_fft_reset(nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process(nullptr, px, 4, d2, &ninv) == MOF_ERR_NOT_INIT && mof_fft_sync(nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process_long_range(nullptr, px, 4, d2, &ninv) == MOF_ERR_NOT_INIT && mof_fft_long_range_patches(nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process_batch_device(nullptr, px, 0, px, 0, 4, 1, d2, nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process_batch_device_bgr(nullptr, px, 0, px, 0, 4, 1, d2, nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process_long_range_batch_device(nullptr, px, 0, px, 0, 4, 1, d2, nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process_batch_host(nullptr, px, 0, px, 0, 4, 1, d2) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process_sequence_device(nullptr, px, 0, 4, 3, d2, nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_process_sequence_device_bgr(nullptr, px, 0, 12, 3, d2, nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_fft_release_graphs(nullptr) == MOF_ERR_NOT_INIT && mof_fft_graph_pinned(nullptr) == 0);
  CHECK(mof_purge_deferred() == 0 && mof_deferred_count() == 0);
  CHECK(std::strcmp(mof_fft_kernel_variant(nullptr), "") == 0);
  mof_fft_destroy(nullptr);
  // ---- block matching geometry (BlockMethod.cpp:11; FastSpacedBMMethod_OCL.cpp:82-90) ----
  mof_bm_config bc;
  CHECK(mof_bm_config_block_method(&bc, 480, 120, 21) == MOF_OK && bc.grid_x == 3 && bc.low_contrast_rule == 0);
  CHECK(mof_bm_config_fast_spaced(&bc, 752, 480, 16, 8, 16) == MOF_OK && bc.grid_x == 30 && bc.grid_y == 18);
  CHECK(mof_bm_config_fast_spaced(&bc, 752, 480, 120, 24, 21) == MOF_OK && bc.grid_x == 4 && bc.grid_y == 3);
  CHECK(mof_bm_config_block_method(nullptr, 1, 1, 1) == MOF_ERR_BAD_ARG && mof_bm_config_fast_spaced(&bc, 0, 1, 1, 1, 1) == MOF_ERR_BAD_ARG);
  mof_bm_engine* be = nullptr;
  mof_bm_config bbad;
  CHECK(mof_bm_config_fast_spaced(&bbad, 752, 480, 16, 8, 16) == MOF_OK);
  bbad.grid_x = 31;
  CHECK(mof_bm_create(&bbad, &be) == MOF_ERR_BAD_ARG);
  bbad.grid_x = 30; bbad.scan_radius = 0; bbad.block_size = 3;
  CHECK(mof_bm_create(&bbad, &be) == MOF_ERR_UNSUPPORTED);
  int8_t i8[8];
  CHECK(mof_bm_set_prev(nullptr, px, 4) == MOF_ERR_NOT_INIT && mof_bm_reset(nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_bm_process(nullptr, px, 4, i8, i8, i8) == MOF_ERR_NOT_INIT && mof_bm_refine(nullptr, 0, 0, 2, 1, d2) == MOF_ERR_NOT_INIT);
  CHECK(mof_bm_process_batch_device(nullptr, px, 0, px, 0, 4, 1, i8, i8, i8, nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_bm_process_batch_host(nullptr, px, 0, px, 0, 4, 1, i8, i8, i8) == MOF_ERR_NOT_INIT && mof_bm_sync(nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_bm_release_graphs(nullptr) == MOF_ERR_NOT_INIT && mof_bm_graph_pinned(nullptr) == 0);
  mof_bm_destroy(nullptr);
  // ---- scale/rotation estimator: argument paths and the host-built tables against the oracle ----
  mof_sr_engine* se = nullptr;
  mof_sr_config sc{480, 49.9, 0, MOF_LOGPOLAR_CV4, 0, 0};
  mof_sr_config sbad = sc;
  sbad.resolution = 2000;  // pads to 2000: beyond the planned transforms
  CHECK(mof_sr_create(&sbad, &se) == MOF_ERR_UNSUPPORTED);
  sbad.resolution = 101;   // odd
  CHECK(mof_sr_create(&sbad, &se) == MOF_ERR_BAD_ARG);
  sbad = sc; sbad.magnitude = 0;
  CHECK(mof_sr_create(&sbad, &se) == MOF_ERR_BAD_ARG);
  sbad = sc; sbad.logpolar_variant = 2;
  CHECK(mof_sr_create(&sbad, &se) == MOF_ERR_BAD_ARG);
  CHECK(mof_sr_process(nullptr, px, 4, d2) == MOF_ERR_NOT_INIT && mof_sr_reset(nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_sr_process_batch_device(nullptr, px, 0, px, 0, 4, 1, d2, nullptr) == MOF_ERR_NOT_INIT);
  CHECK(mof_sr_logpolar_batch_device(nullptr, px, 0, 4, 1, 2, px, nullptr) == MOF_ERR_NOT_INIT);
  {
    int gated = 7;
    CHECK(mof_sr_process_sequence_device(nullptr, px, 0, 4, 3, d2, nullptr, &gated) == MOF_ERR_NOT_INIT);
  }
  CHECK(mof_sr_release_graphs(nullptr) == MOF_ERR_NOT_INIT && mof_sr_graph_pinned(nullptr) == 0 && mof_sr_reserve(nullptr, 1) == MOF_ERR_NOT_INIT);
  mof_sr_destroy(nullptr);
  for (int variant = 0; variant < 2; ++variant)
    for (int res : {240, 256, 480}) {
      const double M = res == 480 ? 49.9 : 40.0;
      const std::vector<mof::SrMapEntry> map = mof::sr_logpolar_map(res, M, variant);
      std::vector<float> mx((size_t)res * res), my((size_t)res * res);
      CHECK(oracle_logpolar_maps(res, M, variant, mx.data(), my.data()) == 0);
      long valid = 0;
      for (size_t i = 0; i < map.size(); ++i) {
        const long ix = std::lrintf(mx[i] * 32.f), iy = std::lrintf(my[i] * 32.f);
        const long ax = ix >> 5, ay = iy >> 5;
        const bool inside = ax >= 0 && ax < res && ay >= 0 && ay < res;
        CHECK((map[i].valid != 0) == inside);
        if (inside) {
          CHECK(map[i].ax == ax && map[i].ay == ay && map[i].widx == (unsigned)((iy & 31) * 32 + (ix & 31)));
          ++valid;
        }
      }
      CHECK(valid > (long)map.size() / 2);
    }
  {
    // weight tables: rows sum to 2^15; remapping a constant image returns the constant (what the oracle does too)
    for (int ks : {4, 8}) {
      const std::vector<int16_t> tab = mof::sr_weight_table(ks);
      CHECK(tab.size() == (size_t)1024 * ks * ks);
      for (int r = 0; r < 1024; ++r) {
        int sum = 0;
        for (int k = 0; k < ks * ks; ++k) sum += tab[(size_t)r * ks * ks + k];
        CHECK(sum == 1 << 15);
      }
    }
  }
  // ---- geometry tail, host forms ----
  {
    const mof_geom_camera cam{340, 338.5, 376, 240, -0.28, 0.07, 0.0004, -0.0003, -0.006};
    const oracle_camera ocam{340, 338.5, 376, 240, -0.28, 0.07, 0.0004, -0.0003, -0.006};
    mof_geom_layout L;
    CHECK(mof_geom_layout_reference(&L, 480, 120) == MOF_OK && L.grid_x == 4);
    CHECK(mof_geom_layout_reference(&L, 100, 120) == MOF_ERR_BAD_ARG);
    CHECK(mof_geom_layout_reference(&L, 480, 120) == MOF_OK);
    const oracle_geom_layout OL{4, 4, 0, 0, 120, 120, 120};
    double shifts[32], out[7], wout[7], H[9], o6[6], w6[6];
    uint8_t mask[16];
    for (int i = 0; i < 16; ++i) { shifts[2 * i] = 3.0 + 0.01 * (i % 4); shifts[2 * i + 1] = -2.0 + 0.02 * (i / 4); }
    shifts[10] = NAN;
    shifts[14] = 55.0;
    mof_geom_rt_params p{2.5, 0.02, 136.0, {0, 0, 0, 1}, {0, 0, 0, 1}, {0, 0, 0}};
    oracle_rt_params op{2.5, 0.02, 136.0, {0, 0, 0, 1}, {0, 0, 0, 1}, {0, 0, 0}};
    oracle_quat_from_rpy(0.01, -0.02, 0.3, p.ang_rate_q);
    std::memcpy(op.ang_rate_q, p.ang_rate_q, sizeof(op.ang_rate_q));
    int status = -1;
    for (int thr = -1; thr <= 17; thr += 3) {
      CHECK(mof_geom_get_rt(shifts, &L, &cam, &p, thr, out, &status, mask, H) == MOF_OK);
      CHECK(status == oracle_get_rt(shifts, &OL, &ocam, &op, thr, wout, nullptr, nullptr));
      for (int k = 0; k < 7; ++k) CHECK(std::fabs(out[k] - wout[k]) <= 1e-9);
    }
    CHECK(mof_geom_get_rt(shifts, &L, &cam, &p, 8, out, &status, nullptr, nullptr) == MOF_OK);
    CHECK(mof_geom_get_rt(nullptr, &L, &cam, &p, 8, out, &status, nullptr, nullptr) == MOF_ERR_BAD_ARG);
    mof_geom_layout big = L;
    big.grid_x = 64; big.grid_y = 32;
    CHECK(mof_geom_get_rt(shifts, &big, &cam, &p, 8, out, &status, nullptr, nullptr) == MOF_ERR_BAD_ARG);
    const mof_geom_2dt_params p2{2.0, 0.02, 0.1, -0.2, 1.0};
    const oracle_2dt_params op2{2.0, 0.02, 0.1, -0.2, 1.0};
    CHECK(mof_geom_get_2dt(shifts, &L, &cam, &p2, o6, &status) == MOF_OK && status == oracle_get_2dt(shifts, &OL, &ocam, &op2, w6));
    CHECK(std::memcmp(o6, w6, sizeof(o6)) == 0);
    double pts[6] = {0, 0, 240, 240, 479, 479}, und[6];
    CHECK(mof_geom_undistort_points(&cam, 136.0, pts, 3, und) == MOF_OK && mof_geom_undistort_points(&cam, 0, nullptr, 0, nullptr) == MOF_OK);
    double R[36], t[12], nn[12];
    int nsol = -1;
    const double Hid[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    CHECK(mof_geom_decompose_homography(Hid, R, t, nn, &nsol) == MOF_OK && nsol == 1);
    int found = -1;
    double a[8] = {0, 0, 1, 0, 1, 1, 0, 1}, b[8] = {0.1, 0, 1.1, 0.05, 1.2, 1.1, 0, 0.9};
    CHECK(mof_geom_find_homography(a, b, 4, H, mask, &found) == MOF_OK && found == 1);
    CHECK(mof_geom_find_homography(a, b, 3, H, mask, &found) == MOF_OK && found == 0);
    CHECK(mof_geom_find_homography(a, b, -1, H, mask, &found) == MOF_ERR_BAD_ARG);
    // the batched forms refuse null device pointers before any launch
    CHECK(mof_geom_get_rt_batch_device(nullptr, &L, &cam, nullptr, 3, 8, nullptr, nullptr) == MOF_ERR_BAD_ARG);
    CHECK(mof_geom_get_2dt_batch_device(nullptr, &L, &cam, nullptr, 3, nullptr, nullptr) == MOF_ERR_BAD_ARG);
    CHECK(mof_geom_get_rt_batch_device(nullptr, &L, &cam, nullptr, 0, 8, nullptr, nullptr) == MOF_OK);
  }
  std::printf("C-ABI sanitizer driver: ok\n");
  return 0;
}
