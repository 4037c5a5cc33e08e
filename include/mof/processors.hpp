// processors.hpp -- header-only C++ host side above the C ABI (include/mof.h).
//
// Mirrors the reference's processor classes for the hot path, same method names, argument
// order/meaning and error behaviour, so that a maintainer can swap them in:
//   class OpticFlowCalc                      /root/reference/include/OpticFlowCalc.h:6-22
//   class FftMethod : OpticFlowCalc          /root/reference/include/FftMethod.h:434-441
//   class BlockMethod : OpticFlowCalc        /root/reference/include/BlockMethod.h:40-42
//   class FastSpacedBMMethod : OpticFlowCalc /root/reference/include/FastSpacedBMMethod_OCL.h:38-42
//
// Two layers:
//   * mof::FftMethod / mof::BlockMethod / mof::FastSpacedBMMethod work on mof::ImageView
//     (pointer + rows/cols/step, i.e. what a CV_8UC1 cv::Mat header carries) and need nothing but
//     this header and libmof_hip.so;
//   * when OpenCV and the reference's OpticFlowCalc.h are on the include path,
//     MofFftMethod : public OpticFlowCalc is the literal drop-in for `FftMethod* fftProcessor_`
//     (/root/reference/src/optic_flow.cpp:251, :1001-1002, :1685-1690).
//
// Error behaviour follows the reference: a re-entrant call returns an EMPTY vector
// (FftMethod.cpp:1775-1776); invalid patches are (NaN, NaN) (:1853); constructor failures and
// HIP errors throw std::runtime_error (the reference lets cv::Exception escape, :1393-1395).
#pragma once

#include <cstdint>
#include <stdexcept>
#include <string>
#include <vector>

#include "../mof.h"

namespace mof {

struct Point2d {
  double x, y;
};
struct Point2i {
  int x, y;
};
struct Point2f {
  float x, y;
};

// What the engine needs of a CV_8UC1 cv::Mat: data pointer, size, row step in bytes.
struct ImageView {
  const uint8_t* data;
  int rows, cols;
  size_t step;
};

namespace detail {
inline void check(int rc, const char* what) {
  if (rc != MOF_OK) throw std::runtime_error(std::string(what) + ": " + mof_last_error());
}
}  // namespace detail

class FftMethod {
 public:
  // Argument list of the reference constructor (FftMethod.h:434-435). storeVideo / videoPath /
  // videoFPS / cl_file_name / useOCL and the three *_enable flags are accepted and ignored: the
  // engine is headless and always runs the HIP path (SURVEY.md §8(b) "side effects not to reproduce").
  // useOCL in particular does NOT pick the arithmetic: the default is cv::phaseCorrelate's peak model, the
  // reference's live CPU path; pass peak_model = MOF_PEAK_OCL to get what its OpenCL kernel computes instead
  // (include/mof.h).
  FftMethod(int i_frameSize, int i_samplePointSize, double max_px_speed_t, bool /*i_storeVideo*/ = false,
            bool /*i_raw_enable*/ = false, bool /*i_rot_corr_enable*/ = false, bool /*i_tilt_corr_enable*/ = false,
            std::string* /*videoPath*/ = nullptr, int /*videoFPS*/ = 0, std::string /*i_cl_file_name*/ = "",
            bool /*i_useOCL*/ = true, int device = 0, int peak_model = MOF_PEAK_OPENCV) {
    detail::check(mof_fft_config_reference(&cfg_, i_frameSize, i_samplePointSize, max_px_speed_t), "FftMethod geometry");
    cfg_.device = device;
    cfg_.peak_model = peak_model;
    detail::check(mof_fft_create(&cfg_, &engine_), "mof_fft_create");
  }
  // Generalised patch layout (origin, stride, grid) -- BASELINE c2/c4 grids do not fit the square tiling.
  explicit FftMethod(const mof_fft_config& cfg) : cfg_(cfg) { detail::check(mof_fft_create(&cfg_, &engine_), "mof_fft_create"); }
  ~FftMethod() { mof_fft_destroy(engine_); }
  FftMethod(const FftMethod&) = delete;
  FftMethod& operator=(const FftMethod&) = delete;

  // OpticFlowCalc::setImPrev (OpticFlowCalc.h:14-16)
  // The C ABI sees a pointer and a pitch only: the frame size is checked here.
  void setImPrev(ImageView imPrev_t) {
    if (imPrev_t.rows != cfg_.frame_height || imPrev_t.cols != cfg_.frame_width)
      throw std::runtime_error("setImPrev: frame size does not match the engine geometry");
    detail::check(mof_fft_set_prev(engine_, imPrev_t.data, imPrev_t.step), "setImPrev");
  }

  // FftMethod::processImage (FftMethod.cpp:1772-1903). midPoint_t, yaw_angle, rot_center and raw_output are
  // ignored exactly as the reference ignores them; fx, fy are stored only (:1781-1782).
  std::vector<Point2d> processImage(ImageView imCurr, bool /*gui*/, bool /*debug*/, Point2i /*midPoint_t*/,
                                    double /*yaw_angle*/, Point2d /*rot_center*/, std::vector<Point2d>& /*raw_output*/,
                                    double i_fx = 300, double i_fy = 300) {
    fx_ = i_fx;
    fy_ = i_fy;
    if (imCurr.rows != cfg_.frame_height || imCurr.cols != cfg_.frame_width)
      throw std::runtime_error("processImage: frame size does not match the engine geometry");
    std::vector<Point2d> speeds((size_t)cfg_.grid_x * cfg_.grid_y);
    const int rc = mof_fft_process(engine_, imCurr.data, imCurr.step, reinterpret_cast<double*>(speeds.data()), &last_invalid_);
    if (rc == MOF_ERR_BUSY) return {};  // `if (running) return std::vector<cv::Point2d>();`
    detail::check(rc, "mof_fft_process");
    return speeds;
  }

  // FftMethod::processImageLongRange (FftMethod.cpp:1905-2007): sqNum/4 x sqNum/4 vectors in quarter-resolution
  // pixels; throws when the geometry has no long-range form (sqNum < 4, sides not divisible by 4).
  std::vector<Point2d> processImageLongRange(ImageView imCurr, bool /*gui*/, bool /*debug*/, Point2i /*midPoint_t*/,
                                             double /*yaw_angle*/, Point2d /*rot_center*/,
                                             std::vector<Point2d>& /*raw_output*/, double i_fx = 300, double i_fy = 300) {
    fx_ = i_fx;
    fy_ = i_fy;
    if (imCurr.rows != cfg_.frame_height || imCurr.cols != cfg_.frame_width)
      throw std::runtime_error("processImageLongRange: frame size does not match the engine geometry");
    const int n = mof_fft_long_range_patches(engine_);
    detail::check(n < 0 ? n : MOF_OK, "mof_fft_long_range_patches");
    std::vector<Point2d> speeds((size_t)n);
    const int rc = mof_fft_process_long_range(engine_, imCurr.data, imCurr.step, reinterpret_cast<double*>(speeds.data()), &last_invalid_);
    if (rc == MOF_ERR_BUSY) return {};
    detail::check(rc, "mof_fft_process_long_range");
    return speeds;
  }

  // A video on the DEVICE: pair k = (frame k + 1, frame k), i.e. the vectors the processImage calls of frames 1..n-1 return
  // (`imPrev = imCurr.clone()`, FftMethod.cpp:1872). d_out_xy: (n_frames - 1) * sqNum^2 * 2 doubles. Asynchronous on `stream`.
  void processSequenceDevice(const uint8_t* d_frames, size_t frame_stride, size_t pitch, int n_frames, double* d_out_xy,
                             void* stream = nullptr) {
    detail::check(mof_fft_process_sequence_device(engine_, d_frames, frame_stride, pitch, n_frames, d_out_xy, stream),
                  "mof_fft_process_sequence_device");
  }

  // A video in HOST memory (a replayed camera log): frame i at frames + i * frame_stride, rows `pitch` bytes apart. Returns the vectors the
  // processImage calls of frames 1 .. n - 1 would return, pair-major ((n_frames - 1) * sqNum^2 entries). The frames go up once each on a
  // copy stream while the previous chunk is computed (mof_fft_process_batch_host; frames in pinned memory -- mof_host_alloc /
  // mof_host_register -- are DMA'd from where they lie). The stateful previous frame of processImage is not touched.
  std::vector<Point2d> processVideo(const uint8_t* frames, size_t frame_stride, size_t pitch, int n_frames) {
    if (n_frames < 2) return {};
    std::vector<Point2d> speeds((size_t)(n_frames - 1) * cfg_.grid_x * cfg_.grid_y);
    detail::check(mof_fft_process_batch_host(engine_, frames + frame_stride, frame_stride, frames, frame_stride, pitch, n_frames - 1,
                                             reinterpret_cast<double*>(speeds.data())), "mof_fft_process_batch_host");
    return speeds;
  }

  int sqNum() const { return cfg_.grid_x; }
  int invalidPatches() const { return last_invalid_; }
  const mof_fft_config& config() const { return cfg_; }
  mof_fft_engine* handle() { return engine_; }

 private:
  mof_fft_config cfg_{};
  mof_fft_engine* engine_ = nullptr;
  double fx_ = 300, fy_ = 300;
  int last_invalid_ = 0;
};

// The batched-frames mode across the GPUs of one node (include/mof.h, mof_shard_*): ceil(B / G) contiguous shards, one engine
// and one stream per device in THIS process, one in-place RCCL all-gather of the result slabs. No reference counterpart (the
// reference is one synchronous call per frame on one device); the C++ host that wants it -- a bag replayer, an offline mapper --
// uploads shard g of its frames to device g and calls process().
class ShardedFftMethod {
 public:
  // devices: the HIP ordinals of the shards (empty: 0 .. n_devices - 1)
  ShardedFftMethod(const mof_fft_config& cfg, int n_devices, const std::vector<int>& devices = {}) : cfg_(cfg) {
    if (!devices.empty() && (int)devices.size() != n_devices) throw std::runtime_error("ShardedFftMethod: one device per shard");
    detail::check(mof_shard_fft_create(&cfg_, devices.empty() ? nullptr : devices.data(), n_devices, &group_), "mof_shard_fft_create");
  }
  ~ShardedFftMethod() { mof_shard_fft_destroy(group_); }
  ShardedFftMethod(const ShardedFftMethod&) = delete;
  ShardedFftMethod& operator=(const ShardedFftMethod&) = delete;

  int devices() const { return mof_shard_fft_devices(group_); }
  // pairs [first, first + count) of a batch of n_pairs belong to `shard`
  void partition(int n_pairs, int shard, int* first, int* count) const {
    detail::check(mof_shard_partition(n_pairs, devices(), shard, first, count), "mof_shard_partition");
  }
  // doubles every device's result buffer must hold: devices * ceil(n_pairs / devices) * patches * 2
  size_t resultDoubles(int n_pairs) const {
    return (size_t)devices() * (size_t)mof_shard_slab_pairs(n_pairs, devices()) * cfg_.grid_x * cfg_.grid_y * 2;
  }
  // d_cur[g] / d_prev[g]: shard g's frames ON device g; d_out[g]: resultDoubles() doubles on device g. gather: every device ends
  // up with all n_pairs results (pair k at pair index k). Asynchronous; sync() waits for every device.
  void process(const std::vector<const uint8_t*>& d_cur, size_t cur_stride, const std::vector<const uint8_t*>& d_prev, size_t prev_stride,
               size_t pitch, int n_pairs, const std::vector<double*>& d_out, bool gather = true) {
    if ((int)d_cur.size() != devices() || (int)d_prev.size() != devices() || (int)d_out.size() != devices())
      throw std::runtime_error("ShardedFftMethod::process: one pointer per shard");
    detail::check(mof_shard_fft_process_batch_device(group_, d_cur.data(), cur_stride, d_prev.data(), prev_stride, pitch, n_pairs,
                                                     d_out.data(), gather ? 1 : 0), "mof_shard_fft_process_batch_device");
  }
  void sync() { detail::check(mof_shard_fft_sync(group_), "mof_shard_fft_sync"); }
  // The gather's set-up (RCCL bound, ncclCommInitAll): explicit and blocking, once, BEFORE the first process(.., gather = true) --
  // the asynchronous batch call never builds communicators itself (it throws with MOF_ERR_NOT_INIT's text instead).
  void initGather() { detail::check(mof_shard_fft_init_gather(group_), "mof_shard_fft_init_gather"); }
  bool gatherReady() const { return mof_shard_fft_gather_ready(group_) != 0; }

 private:
  mof_fft_config cfg_{};
  mof_shard_fft* group_ = nullptr;
};

// The same for the block matchers (mof_shard_bm_*): per-block shifts and the per-pair histogram modes -- what
// FastSpacedBMMethod::processImage returns, FastSpacedBMMethod_OCL.cpp:172-175 -- ride in one slab per device
// (dx | dy | mode planes, include/mof.h), so one all-gather moves everything.
class ShardedBlockMatcher {
 public:
  struct Where {  // byte offsets of one pair's results inside a device's result buffer
    size_t dx, dy, mode;
  };
  ShardedBlockMatcher(const mof_bm_config& cfg, int n_devices, const std::vector<int>& devices = {}) : cfg_(cfg) {
    if (!devices.empty() && (int)devices.size() != n_devices) throw std::runtime_error("ShardedBlockMatcher: one device per shard");
    detail::check(mof_shard_bm_create(&cfg_, devices.empty() ? nullptr : devices.data(), n_devices, &group_), "mof_shard_bm_create");
  }
  ~ShardedBlockMatcher() { mof_shard_bm_destroy(group_); }
  ShardedBlockMatcher(const ShardedBlockMatcher&) = delete;
  ShardedBlockMatcher& operator=(const ShardedBlockMatcher&) = delete;

  int devices() const { return mof_shard_bm_devices(group_); }
  void partition(int n_pairs, int shard, int* first, int* count) const {
    detail::check(mof_shard_partition(n_pairs, devices(), shard, first, count), "mof_shard_partition");
  }
  // bytes every device's result buffer must hold: devices * (one rank's slab)
  size_t resultBytes(int n_pairs) const { return (size_t)devices() * mof_shard_bm_slab_bytes(group_, n_pairs); }
  Where locate(int n_pairs, int pair) const {
    Where w{};
    detail::check(mof_shard_bm_locate(group_, n_pairs, pair, &w.dx, &w.dy, &w.mode), "mof_shard_bm_locate");
    return w;
  }
  void process(const std::vector<const uint8_t*>& d_cur, size_t cur_stride, const std::vector<const uint8_t*>& d_prev, size_t prev_stride,
               size_t pitch, int n_pairs, const std::vector<int8_t*>& d_out, bool gather = true) {
    if ((int)d_cur.size() != devices() || (int)d_prev.size() != devices() || (int)d_out.size() != devices())
      throw std::runtime_error("ShardedBlockMatcher::process: one pointer per shard");
    detail::check(mof_shard_bm_process_batch_device(group_, d_cur.data(), cur_stride, d_prev.data(), prev_stride, pitch, n_pairs,
                                                    d_out.data(), gather ? 1 : 0), "mof_shard_bm_process_batch_device");
  }
  void sync() { detail::check(mof_shard_bm_sync(group_), "mof_shard_bm_sync"); }
  void initGather() { detail::check(mof_shard_bm_init_gather(group_), "mof_shard_bm_init_gather"); }
  bool gatherReady() const { return mof_shard_bm_gather_ready(group_) != 0; }

 private:
  mof_bm_config cfg_{};
  mof_shard_bm* group_ = nullptr;
};

// Integer stage of both block matchers behind one engine.
class BlockMatcherBase {
 public:
  ~BlockMatcherBase() { mof_bm_destroy(engine_); }
  BlockMatcherBase(const BlockMatcherBase&) = delete;
  BlockMatcherBase& operator=(const BlockMatcherBase&) = delete;
  void setImPrev(ImageView imPrev_t) {
    if (imPrev_t.rows != cfg_.frame_height || imPrev_t.cols != cfg_.frame_width)
      throw std::runtime_error("setImPrev: frame size does not match the engine geometry");
    detail::check(mof_bm_set_prev(engine_, imPrev_t.data, imPrev_t.step), "setImPrev");
  }
  // per-block integer shifts of the last processImage call, index by*grid_x + bx
  const std::vector<int8_t>& flowX() const { return dx_; }
  const std::vector<int8_t>& flowY() const { return dy_; }
  const mof_bm_config& config() const { return cfg_; }

 protected:
  explicit BlockMatcherBase(const mof_bm_config& cfg) : cfg_(cfg) {
    detail::check(mof_bm_create(&cfg_, &engine_), "mof_bm_create");
    dx_.resize((size_t)cfg_.grid_x * cfg_.grid_y);
    dy_.resize(dx_.size());
  }
  // returns false when the engine was busy
  bool run(ImageView im, int8_t mode[2]) {
    if (im.rows != cfg_.frame_height || im.cols != cfg_.frame_width)
      throw std::runtime_error("processImage: frame size does not match the engine geometry");
    const int rc = mof_bm_process(engine_, im.data, im.step, dx_.data(), dy_.data(), mode);
    if (rc == MOF_ERR_BUSY) return false;
    detail::check(rc, "mof_bm_process");
    return true;
  }
  mof_bm_config cfg_{};
  mof_bm_engine* engine_ = nullptr;
  std::vector<int8_t> dx_, dy_;
};

class BlockMethod : public BlockMatcherBase {
 public:
  // BlockMethod(frameSize, samplePointSize, scanRadius, scanDiameter, scanCount, stepSize) -- BlockMethod.cpp:3
  BlockMethod(int i_frameSize, int i_samplePointSize, int i_scanRadius, int /*i_scanDiameter*/ = 0, int /*i_scanCount*/ = 0,
              int /*i_stepSize*/ = 0, int device = 0)
      : BlockMatcherBase(make(i_frameSize, i_samplePointSize, i_scanRadius, device)) {}
  enum RefineMode { kNoRefine = 0, kFaithful = 1, kRepaired = 2 };
  // BlockMethod::processImage (BlockMethod.cpp:25-94): ONE vector = Refine(histogram mode, 2) (:79). kFaithful
  // (default) reproduces Refine literally, including that its "previous" image is resized from the current one
  // (SURVEY F9); kRepaired resizes it from the previous frame; kNoRefine returns the integer mode.
  void setRefine(RefineMode m) { refine_ = m; }
  std::vector<Point2d> processImage(ImageView imCurr, bool /*gui*/, bool /*debug*/, Point2i /*midPoint_t*/,
                                    double /*yaw_angle*/, Point2d /*tiltCorr*/) {
    int8_t mode[2] = {0, 0};
    if (!run(imCurr, mode)) return {};
    mode_ = Point2i{mode[0], mode[1]};
    if (refine_ == kNoRefine) return {Point2d{(double)mode[0], (double)mode[1]}};
    double r[2] = {0, 0};
    detail::check(mof_bm_refine(engine_, mode[0], mode[1], 2, refine_ == kFaithful ? 1 : 0, r), "mof_bm_refine");
    return {Point2d{r[0], r[1]}};
  }
  Point2i mode() const { return mode_; }

 private:
  RefineMode refine_ = kFaithful;
  Point2i mode_{0, 0};
  static mof_bm_config make(int fs, int sps, int r, int device) {
    mof_bm_config c{};
    detail::check(mof_bm_config_block_method(&c, fs, sps, r), "BlockMethod geometry");
    c.device = device;
    return c;
  }
};

class FastSpacedBMMethod : public BlockMatcherBase {
 public:
  // FastSpacedBMMethod(samplePointSize, scanRadius, stepSize, cx, cy, fx, fy, k1, k2, k3, p1, p2, storeVideo, videoPath)
  // -- FastSpacedBMMethod_OCL.cpp:5-6; the camera parameters are stored-only there and dropped here. The frame
  // size, which the reference takes from the first image, is fixed at construction.
  FastSpacedBMMethod(int i_samplePointSize, int i_scanRadius, int i_stepSize, int frame_width, int frame_height, int device = 0)
      : BlockMatcherBase(make(frame_width, frame_height, i_samplePointSize, i_stepSize, i_scanRadius, device)) {}
  // FastSpacedBMMethod::processImage (FastSpacedBMMethod_OCL.cpp:71-184): one Point2f = (modeX, modeY) (:172-175)
  std::vector<Point2f> processImage(ImageView imCurr, bool /*gui*/, bool /*debug*/, Point2i /*midPoint_t*/,
                                    double /*yaw_angle*/, Point2d /*tiltCorr*/) {
    int8_t mode[2] = {0, 0};
    if (!run(imCurr, mode)) return {};
    return {Point2f{(float)mode[0], (float)mode[1]}};
  }

 private:
  static mof_bm_config make(int w, int h, int sps, int step, int r, int device) {
    mof_bm_config c{};
    detail::check(mof_bm_config_fast_spaced(&c, w, h, sps, step, r), "FastSpacedBM geometry");
    c.device = device;
    return c;
  }
};

// scaleRotationEstimator(resolution, m, storeVideo, videoPath, videoFPS) -- scaleRotationEstimator.cpp:3-32;
// processImage returns (scale, rotation [rad]) as the reference's cv::Point2d(scale, rotat) (:126).
class scaleRotationEstimator {
 public:
  scaleRotationEstimator(int res, double m, bool /*i_storeVideo*/ = false, std::string* /*videoPath*/ = nullptr,
                         int /*videoFPS*/ = 0, int device = 0, int logpolar_variant = MOF_LOGPOLAR_CV4) {
    mof_sr_config c{};  // batched-mode fields (batch_chunk, pipeline_lanes) at their defaults
    c.resolution = res;
    c.magnitude = m;
    c.device = device;
    c.logpolar_variant = logpolar_variant;
    detail::check(mof_sr_create(&c, &engine_), "mof_sr_create");
    res_ = res;
  }
  ~scaleRotationEstimator() { mof_sr_destroy(engine_); }
  scaleRotationEstimator(const scaleRotationEstimator&) = delete;
  scaleRotationEstimator& operator=(const scaleRotationEstimator&) = delete;
  Point2d processImage(ImageView imCurr, bool /*gui*/, bool /*debug*/) {
    if (imCurr.rows != res_ || imCurr.cols != res_) throw std::runtime_error("scaleRotationEstimator: accepts only res x res images");
    double out[2] = {1.0, 0.0};
    detail::check(mof_sr_process(engine_, imCurr.data, imCurr.step, out), "mof_sr_process");
    return Point2d{out[0], out[1]};
  }
  // A video on the DEVICE: what n_frames consecutive processImage calls return (scale, rot, pt.x, pt.y per frame), continuing
  // and updating this estimator's state (first frame INTER_CUBIC, the gate of :119-121 resolved). d_frames: top-left pixel of
  // the res x res crop of frame 0, frame i at + i * frame_stride, `pitch` bytes per row; d_out4: n_frames * 4 doubles.
  // Synchronous; returns the number of gated frames. mof_sr_process_sequence_device (include/mof.h).
  int processSequenceDevice(const uint8_t* d_frames, size_t frame_stride, size_t pitch, int n_frames, double* d_out4,
                            void* stream = nullptr) {
    int gated = 0;
    detail::check(mof_sr_process_sequence_device(engine_, d_frames, frame_stride, pitch, n_frames, d_out4, stream, &gated),
                  "mof_sr_process_sequence_device");
    return gated;
  }
  // The same video in HOST memory (mof_sr_process_sequence_host): out4 = n_frames * 4 doubles; the frames are uploaded in chunks beside the
  // previous chunk's kernels (pinned frames -- mof_host_alloc / mof_host_register -- from where they lie). Returns the number of gated frames.
  int processVideo(const uint8_t* frames, size_t frame_stride, size_t pitch, int n_frames, double* out4) {
    int gated = 0;
    detail::check(mof_sr_process_sequence_host(engine_, frames, frame_stride, pitch, n_frames, out4, &gated), "mof_sr_process_sequence_host");
    return gated;
  }
  mof_sr_engine* handle() { return engine_; }

 private:
  mof_sr_engine* engine_ = nullptr;
  int res_ = 0;
};

}  // namespace mof

// ------------------------------------------------------------------------------------------------
// Literal drop-in for the ROS node, compiled only where the reference's headers and OpenCV exist.
// ------------------------------------------------------------------------------------------------
#if defined(__has_include)
#if __has_include(<opencv2/core.hpp>) && __has_include(<OpticFlowCalc.h>)
#include <OpticFlowCalc.h>

#include <opencv2/core.hpp>

// `final`: OpticFlowCalc has no virtual destructor (OpticFlowCalc.h:6-22), so the object must be destroyed through its own type,
// as the node does with its concrete `FftMethod* fftProcessor_` (optic_flow.cpp:251).
class MofFftMethod final : public OpticFlowCalc {
 public:
  MofFftMethod(int i_frameSize, int i_samplePointSize, double max_px_speed_t, bool i_storeVideo, bool i_raw_enable,
               bool i_rot_corr_enable, bool i_tilt_corr_enable, std::string* videoPath, int videoFPS,
               std::string i_cl_file_name, bool i_useOCL)
      : impl_(i_frameSize, i_samplePointSize, max_px_speed_t, i_storeVideo, i_raw_enable, i_rot_corr_enable,
              i_tilt_corr_enable, videoPath, videoFPS, i_cl_file_name, i_useOCL) {
    max_px_speed_sq = max_px_speed_t * max_px_speed_t;
  }

  std::vector<cv::Point2d> processImage(cv::Mat imCurr, bool gui, bool debug, cv::Point midPoint_t, double yaw_angle,
                                        cv::Point2d rot_center, std::vector<cv::Point2d>& output_vectors_raw,
                                        double i_fx = 300, double i_fy = 300) override {
    CV_Assert(imCurr.type() == CV_8UC1);
    std::vector<mof::Point2d> raw;
    std::vector<mof::Point2d> r =
        impl_.processImage(mof::ImageView{imCurr.data, imCurr.rows, imCurr.cols, imCurr.step}, gui, debug,
                           mof::Point2i{midPoint_t.x, midPoint_t.y}, yaw_angle, mof::Point2d{rot_center.x, rot_center.y},
                           raw, i_fx, i_fy);
    (void)output_vectors_raw;  // untouched by the reference as well
    std::vector<cv::Point2d> out(r.size());
    for (size_t i = 0; i < r.size(); ++i) out[i] = cv::Point2d(r[i].x, r[i].y);
    return out;
  }
  std::vector<cv::Point2d> processImageLongRange(cv::Mat imCurr, bool gui, bool debug, cv::Point midPoint_t,
                                                 double yaw_angle, cv::Point2d rot_center,
                                                 std::vector<cv::Point2d>& output_vectors_raw, double i_fx = 300,
                                                 double i_fy = 300) {
    CV_Assert(imCurr.type() == CV_8UC1);
    std::vector<mof::Point2d> raw;
    std::vector<mof::Point2d> r = impl_.processImageLongRange(
        mof::ImageView{imCurr.data, imCurr.rows, imCurr.cols, imCurr.step}, gui, debug,
        mof::Point2i{midPoint_t.x, midPoint_t.y}, yaw_angle, mof::Point2d{rot_center.x, rot_center.y}, raw, i_fx, i_fy);
    (void)output_vectors_raw;
    std::vector<cv::Point2d> out(r.size());
    for (size_t i = 0; i < r.size(); ++i) out[i] = cv::Point2d(r[i].x, r[i].y);
    return out;
  }
  // hides OpticFlowCalc::setImPrev (non-virtual there): forwards the pixels to the device-resident previous frame
  void setImPrev(cv::Mat imPrev_t) {
    imPrev = imPrev_t;
    impl_.setImPrev(mof::ImageView{imPrev_t.data, imPrev_t.rows, imPrev_t.cols, imPrev_t.step});
  }

 private:
  mof::FftMethod impl_;
};
#endif
#endif
