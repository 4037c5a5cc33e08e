/*
 * oracle.h -- CPU restatement of the mrs_optic_flow hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This directory is the parity oracle: a plain-C restatement of the reference's
 * useOCL=false FFT phase-correlation path and of its two block-matching paths.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may call it.
 * The shipped product (mrs_optic_flow_amd/, include/mof.h) never links or loads it.
 *
 * PARITY UNPINNED: the reference ships no tests, fixtures or golden vectors
 * (SURVEY.md F11, §8c), its CPU arithmetic lives in OpenCV (cv::phaseCorrelate,
 * unpinned: CMakeLists.txt:23; 4.2.0 implied by ROS Noetic) which is absent here,
 * and no reference translation unit compiles without OpenCV + ROS headers.
 * The restatement therefore follows the in-repo helper copies and call sites
 * cited per function, plus OpenCV's published phaseCorrelate algorithm for the
 * external calls (dft/idft CCS packing, mulSpectrums, minMaxLoc, weightedCentroid).
 * It is cross-checked against an independent numpy twin (tests/twin.py) and
 * analytic known-answer cases, not against reference output.
 */
#ifndef MOF_ORACLE_H
#define MOF_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Patch layout of the FFT path. The reference tiles a square crop contiguously
 * (src/FftMethod.cpp:1719, :1831-1832: xi=i*sps, yi=j*sps); that is the special
 * case origin=0, stride=patch, grid=sqNum. */
typedef struct oracle_fft_layout {
  int width, height;      /* frame size in pixels                       */
  int patch;              /* samplePointSize N (any >= 2; padded inside)*/
  int grid_x, grid_y;     /* patches per row / column                   */
  int origin_x, origin_y; /* top-left of patch (0,0)                    */
  int stride_x, stride_y; /* distance between patch origins             */
  double max_px_speed;    /* gate, src/FftMethod.cpp:1686, :1841        */
} oracle_fft_layout;

/* Diagnostics of one correlation (for conditioning classes in tests). */
typedef struct oracle_pc_diag {
  int peak_x, peak_y;     /* arg-max in fft-shifted coordinates         */
  double peak_value;      /* surface value at the peak                  */
  double second_value;    /* largest value outside the 5x5 window       */
  double response;        /* sum over the 5x5 window / (N*N)            */
} oracle_pc_diag;

/* cv::getOptimalDFTSize: smallest 2^a 3^b 5^c >= n [published OpenCV algorithm, unpinned]. */
int oracle_optimal_dft_size(int n);

/* cv::phaseCorrelate(a, b) restated (src/FftMethod.cpp:1487-1498 stage order;
 * :70-168 magSpectrums; :1086-1251 divSpectrums; :1257-1323 fftShift;
 * :1329-1385 weightedCentroid). a,b: n x n (any n >= 2), row strides in ELEMENTS.
 * As cv::phaseCorrelate does, both are zero-padded (bottom / right) to m x m,
 * m = oracle_optimal_dft_size(n), which may be odd; peak, centroid and centre
 * (m / 2.0) live on the padded image. Returns (center - t) in out_xy[0..1].
 * surface (optional, m*m) receives the fft-shifted correlation surface; diag's
 * peak is in its coordinates. Return 0 ok, <0 bad argument. */
int oracle_phase_correlate_f32(const float* a, size_t a_stride, const float* b, size_t b_stride, int n,
                               double* out_xy, oracle_pc_diag* diag, float* surface);
/* Same arithmetic carried in double everywhere ("truth" variant; eps stays FLT_EPSILON). */
int oracle_phase_correlate_f64(const double* a, size_t a_stride, const double* b, size_t b_stride, int n,
                               double* out_xy, oracle_pc_diag* diag, double* surface);

/* FftMethod::processImage, useOCL=false branch (src/FftMethod.cpp:1805-1806 u8->f32,
 * :1829-1856 patch loop + gating). out_xy: 2*grid_x*grid_y doubles, index
 * (i + j*grid_x), (x,y); invalid -> (NaN,NaN). diag optional (grid_x*grid_y).
 * precision: 32 or 64. */
int oracle_fft_process_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_fft_layout* layout,
                          int precision, double* out_xy, int* n_invalid, oracle_pc_diag* diag);

/* The useOCL=true peak model (cl/FftMethod.cl:971-982 rsqrt normalisation, :737-746/:823-826
 * +-SEARCH_RADIUS mask, :1164-1313 arg-max, :1315-1379 7x7 positive-only float centroid over absolute coordinates),
 * restated as the race-free maths the kernel chain intends -- see pc_ref_impl.h. (x0, y0) is the patch origin in the
 * frame (the centroid is accumulated in absolute coordinates, which costs float precision exactly as in the kernel).
 * out_xy is the SHIFT itself (the host does not negate this branch, src/FftMethod.cpp:1833). */
int oracle_phase_correlate_ocl_f32(const float* a, size_t a_stride, const float* b, size_t b_stride, int n, int x0,
                                   int y0, int search_radius, double* out_xy, oracle_pc_diag* diag, float* surface);
int oracle_phase_correlate_ocl_f64(const double* a, size_t a_stride, const double* b, size_t b_stride, int n, int x0,
                                   int y0, int search_radius, double* out_xy, oracle_pc_diag* diag, double* surface);
/* FftMethod::processImage with useOCL=true under that model (src/FftMethod.cpp:1824-1825, :1833, :1840-1856;
 * SEARCH_RADIUS 55, :820). Same layout/outputs as oracle_fft_process_u8. */
int oracle_fft_process_ocl_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_fft_layout* layout,
                              int search_radius, int precision, double* out_xy, int* n_invalid, oracle_pc_diag* diag);

/* cv::resize(src, dst, Size(), 1/4, 1/4) with the default INTER_LINEAR on CV_8UC1, as called by
 * FftMethod::processImageLongRange (src/FftMethod.cpp:1931-1932, LONG_RANGE_RATIO :3). With scale 4 the
 * source coordinate of destination x is 4x + 1.5, i.e. both taps weigh 1/2 in each axis, and OpenCV's
 * fixed-point path (coefficients 1024, vertical pass ((b*(S>>4))>>16 ... + 2) >> 2) reduces exactly to
 *   dst(y,x) = (src(4y+1,4x+1) + src(4y+1,4x+2) + src(4y+2,4x+1) + src(4y+2,4x+2) + 2) >> 2.
 * [published OpenCV algorithm, unpinned]. w, h must be multiples of 4. dst is (h/4) x (w/4), tightly packed. */
int oracle_resize_quarter_u8(const uint8_t* src, size_t pitch, int w, int h, uint8_t* dst);

/* FftMethod::processImageLongRange, useOCL=false (src/FftMethod.cpp:1905-2007): both frames are reduced
 * to a quarter, then the same per-patch correlation and gate run on the sqNum/4 x sqNum/4 grid of
 * N x N patches (samplePointSize_lr = samplePointSize, :1685; sqNum_lr = sqNum/4, :1720).
 * `layout` describes the FULL-resolution reference tiling (origin 0, stride = patch); out_xy receives
 * 2*(grid_x/4)*(grid_y/4) doubles. */
int oracle_fft_process_long_range_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch,
                                     const oracle_fft_layout* layout, int precision, double* out_xy, int* n_invalid);

/* cv::cvtColor(src, dst, CV_RGB2GRAY) on interleaved 8-bit 3-channel data, as the node's front end applies it to BGR8
 * frames (/root/reference/src/optic_flow.cpp:1465, :1622): dst = (c0*4899 + c1*9617 + c2*1868 + 8192) >> 14
 * (OpenCV's fixed-point RGB2Gray, yuv_shift 14; published algorithm, unpinned). dst is w*h, tightly packed. */
int oracle_rgb2gray_u8(const uint8_t* src, size_t pitch_bytes, int w, int h, uint8_t* dst);

/* cv::logPolar(src, dst, Point2f(res/2, res/2), M, interp) on a res x res CV_8UC1 image, dst pre-existing
 * (pixels mapped outside the source keep their content: BORDER_TRANSPARENT). interp: 2 = INTER_CUBIC,
 * 4 = INTER_LANCZOS4. See lp_ref.c for the restated OpenCV semantics (unpinned). */
int oracle_logpolar_u8(const uint8_t* src, size_t pitch, int res, double M, int interp, uint8_t* dst);
/* Same with the OpenCV generation chosen explicitly: variant 0 = cv::logPolar of OpenCV 4.x (ROS Noetic; the warpPolar
 * form, `exp(rho * Kmag) - 1` in a float table), variant 1 = cvLogPolar of OpenCV 3.2 (ROS Melodic; `exp(rho / M)`,
 * no "- 1", double table). scaleRotationEstimator.cpp:41-46, :107-113 compile one or the other. */
int oracle_logpolar_variant_u8(const uint8_t* src, size_t pitch, int res, double M, int interp, int variant, uint8_t* dst);
/* The float maps either variant hands to cv::remap ([phi][rho], res*res each). */
int oracle_logpolar_maps(int res, double M, int variant, float* mapx, float* mapy);

/* scaleRotationEstimator::processImage (src/scaleRotationEstimator.cpp:34-148), one call:
 * first != 0: temp_im <- logPolar(frame, INTER_CUBIC), prev_lp <- float(temp_im), out = (1, 0)      (:36-74)
 * else: temp_im <- logPolar(frame, INTER_LANCZOS4) (:112); pt = cv::phaseCorrelate(cur_lp, prev_lp) (:117);
 *       |pt.x| > res/2 -> (1, 0) (:119-121, pt.y is never checked); scale = exp(pt.x / M),
 *       rot = (pt.y / Ky) * pi / 180 with Ky = res / 360 (:123-124, :26); prev_lp <- cur_lp (:128).
 * temp_im (res*res u8) and prev_lp (res*res float) are the estimator's state; pt_xy (optional) gets pt. */
int oracle_scale_rotation_step(const uint8_t* frame, size_t pitch, int res, double M, int first, uint8_t* temp_im,
                               float* prev_lp, int precision, double* out_scale_rot, double* pt_xy);

/* Same with the cv::logPolar generation chosen (see oracle_logpolar_variant_u8). */
int oracle_scale_rotation_step_variant(const uint8_t* frame, size_t pitch, int res, double M, int first, uint8_t* temp_im,
                                       float* prev_lp, int precision, int variant, double* out_scale_rot, double* pt_xy);

/* Block geometry shared by both block-matching paths.
 * BlockMethod (src/BlockMethod.cpp:11, :45): step=0, threshold off,
 *   grid = (fs-2r)/sps squared.
 * FastSpacedBM (src/FastSpacedBMMethod_OCL.cpp:82-90): S=sps+step,
 *   grid = ((W-2r)/S, (H-2r)/S), low-contrast threshold on. */
typedef struct oracle_bm_config {
  int width, height;
  int block;      /* samplePointSize */
  int step;       /* stepSize (0 for BlockMethod) */
  int radius;     /* scanRadius */
  int grid_x, grid_y;
  int low_contrast_rule; /* 1: FastSpacedBMMethod.cl:77-82 */
} oracle_bm_config;

/* Fill grid_x/grid_y the way each reference class does. */
void oracle_bm_config_block_method(oracle_bm_config* c, int frame_size, int block, int radius);
void oracle_bm_config_fast_spaced(oracle_bm_config* c, int width, int height, int block, int step, int radius);

/* Exhaustive SAD scan. dx,dy: grid_x*grid_y entries (by*grid_x+bx). mode_xy[2]:
 * per-axis histogram mode (first max). sad_min optional (per block minimum SAD),
 * sad_all optional (grid * (2r+1)^2 ints, [block][ys][xs]).
 * src/BlockMethod.cpp:43-76; src/FastSpacedBMMethod.cl:4-84, :86-169. */
int oracle_bm_process_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, const oracle_bm_config* cfg,
                         int8_t* dx, int8_t* dy, int8_t* mode_xy, int32_t* sad_min, int32_t* sad_all);

/* cv::resize(src, dst, Size(2w, 2h)) with the default INTER_LINEAR on CV_8UC1 (BlockMethod::Refine,
 * src/BlockMethod.cpp:110-111): source coordinate of destination x is x/2 - 0.25, i.e. taps (k-1, k) weighted
 * (1/4, 3/4) for x = 2k and (k, k+1) weighted (3/4, 1/4) for x = 2k+1, clamped at the borders; OpenCV's fixed point:
 * horizontal sums with coefficients 512/1536 (of 2048), vertical ((b0*(S0>>4))>>16) + ((b1*(S1>>4))>>16) + 2) >> 2.
 * [published OpenCV algorithm, unpinned]. dst is 2h x 2w, tightly packed. */
int oracle_resize_2x_u8(const uint8_t* src, size_t pitch, int w, int h, uint8_t* dst);

/* BlockMethod::Refine(imCurr, imPrev, fullpixFlow, passes) -- src/BlockMethod.cpp:96-147 -- restated literally,
 * including (faithful != 0) its two defects: the "previous" image of every pass is resized from the CURRENT one
 * (:110) and non-negative offsets never move the previous-image cut-out (:116-123). With faithful == 0
 * the previous image is resized from the previous image (the evident intent); everything else is unchanged.
 * Per pass: both images are brought to twice the ORIGINAL size (:110-111; the second pass therefore copies), the
 * offset is doubled, nine SADs between the cut-out of the current image at (1,1) and the previous image at
 * startpoint + (n,m) are taken, and the first minimum moves the offset by (n,m). Returns offset / 2^passes in
 * out_xy. Fails (-3) when a cut-out would be empty. */
int oracle_bm_refine_u8(const uint8_t* cur, const uint8_t* prev, size_t pitch, int w, int h, int fullpix_x, int fullpix_y,
                        int passes, int faithful, double* out_xy, int32_t* sads /* optional, passes*9 */);

/* Histogram_C1_D0 top-TestDepth output (src/FastSpacedBMMethod.cl:155-167):
 * sorted shift indices per axis (stable descending by count), first `depth`. */
int oracle_bm_histogram_top(const int8_t* d, int count, int radius, int depth, int8_t* top);

/* ---- geometry tail (geom_ref.c): OpticFlow::get2DT (src/optic_flow.cpp:388-510), OpticFlow::getRT (:515-774) ---- */
typedef struct oracle_camera { double fx, fy, cx, cy, k1, k2, p1, p2, k3; } oracle_camera; /* :1511-1522 */
typedef struct oracle_geom_layout { int grid_x, grid_y, origin_x, origin_y, stride_x, stride_y, patch; } oracle_geom_layout;
typedef struct oracle_rt_params {
  double height, dt, ul_corner_x;
  double ang_rate_q[4]; /* angular_rate_tf_ (x, y, z, w) */
  double c2b_q[4];      /* transformCam2Base_ rotation   */
  double c2b_t[3];      /* and translation               */
} oracle_rt_params;
typedef struct oracle_2dt_params { double height, dt, roll_rate, pitch_rate, cam_yaw; } oracle_2dt_params;

/* cv::undistortPoints on one pixel (camMatrixLocal = camMatrix with cx - ul_corner_x, 5 distortion coefficients). */
void oracle_undistort_point(const oracle_camera* c, double ul_corner_x, double u, double v, double* ox, double* oy);
/* cv::findHomography(a, b, RANSAC, 0.01, mask) with the project's documented sampler (see geom_ref.c). 1 = found. */
int oracle_find_homography(const double* a_xy, const double* b_xy, int n, double* H9, uint8_t* mask);
/* cv::decomposeHomographyMat(H, I, ...): returns the number of solutions (1, 4; 0 degenerate). R[4][9], t[4][3], n[4][3]. */
int oracle_decompose_homography(const double* H9, double* R, double* t, double* normals);
/* tf2::Quaternion::setRPY -> (x, y, z, w) */
void oracle_quat_from_rpy(double roll, double pitch, double yaw, double* q_xyzw);
/* getRT: returns the status (0 = true; codes as MOF_GEOM_* in include/mof.h). out[7] = rot (x,y,z,w), tran (x,y,z). */
int oracle_get_rt(const double* shifts_xy, const oracle_geom_layout* L, const oracle_camera* cam, const oracle_rt_params* p,
                  int shifted_pts_thr, double* out_rot_tran, uint8_t* mask_out /*optional*/, double* H_out /*optional*/);
/* get2DT: returns the status. out[6] = o_tran (3), o_tran_diff (3). */
int oracle_get_2dt(const double* shifts_xy, const oracle_geom_layout* L, const oracle_camera* cam,
                   const oracle_2dt_params* p, double* out_tran_diff);

const char* oracle_version(void);

#ifdef __cplusplus
}
#endif
#endif
