// mof_sr.hip -- host side of the scale/rotation estimator behind the C ABI (mof_sr_* in include/mof.h).
//
// Mirrors scaleRotationEstimator (/root/reference/src/scaleRotationEstimator.cpp:3-32 ctor, :34-148
// processImage): state = the persistent log-polar image `tempIm` (:27), the previous log-polar image
// `prevIm_F32` (:48, :128) and `first` (:31). What cv::logPolar / cv::remap would compute on the host for
// every frame -- the float maps, their 1/32-px fixed-point form and the 2^15-scaled kernel tables -- depends
// only on (resolution, M), so it is built ONCE here and kept on the device; per frame the GPU does one gather
// (K4) and the whole-frame phase correlation (K5..K8). No CPU compute path.

#include <hip/hip_runtime.h>

#include <atomic>
#include <cfloat>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <new>
#include <vector>

#include "capi_graph.hpp"
#include "host_pipe.hpp"
#include "mof.h"
#include "mof_kernels.h"

namespace mof {
int capi_fail(int code, const char* fmt, ...);  // mof_capi.hip: records the thread's last error
}

namespace {

constexpr int kTab = 32;            // INTER_TAB_SIZE
constexpr int kCoefScale = 1 << 15; // INTER_REMAP_COEF_SCALE
#ifndef MOF_SR_CHUNK
#define MOF_SR_CHUNK 512
#endif
// frame pairs per pipeline pass: 512 pairs of 480^2 = 1.9 GB of scratch (log-polar images, Zt, Dt). Same-box sweeps at c5
// (tools/sweep_c5.sh, profiles/r02_c5_sweep.txt): 64 / 128 / 256 pairs -> 240 / 277 / 308 k pairs/s on one lane, 512 ->
// 326 k, 1024 -> 324 k; with the second lane 254 / 283 / 312 / 318 / 313 k -- once a pass is long enough to fill the
// chip the two-lane overlap has nothing left to hide, so one lane is the default. r03 (frame kernels, non-temporal streams;
// tools/ab_sr_chunk.sh): 256 / 512 / 1024 pairs -> 341 / 360 / 366 k (c5) and 424 / 518 / 535 k (c5seq); passes that fit the
// 256 MB Infinity Cache (64 pairs: Zt 118 MB + Dt 59 MB) are no faster per byte and pay the launch gaps (248 k). 1024-pair
// passes are worth +1.7 % at c5 and cost 3.8 GB of scratch that stays allocated (and pinned once a graph captured the engine),
// so the DEFAULT stays 512 pairs = 1.9 GB (r04, advisor); a caller that has the memory opts in through
// mof_sr_config.batch_chunk (bench.py does for c5 / c5seq).
constexpr int kChunkDefault = MOF_SR_CHUNK;
// mof_sr_config.batch_chunk / .pipeline_lanes; MOF_SR_CHUNK / MOF_SR_OVERLAP in the environment override both at
// create() (sweeps)
int chunk_pairs(int cfg_chunk) {
  const char* e = getenv("MOF_SR_CHUNK");
  const int n = e ? atoi(e) : cfg_chunk;
  return n >= 1 && n <= 4096 ? n : kChunkDefault;
}
bool two_lane_default(int cfg_lanes) {
  const char* v = getenv("MOF_SR_OVERLAP");
  return v ? atoi(v) != 0 : cfg_lanes == 2;
}

#define SR_TRY(expr)                                                                                  \
  do {                                                                                                \
    hipError_t _e = (expr);                                                                           \
    if (_e != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e)); \
  } while (0)

void kernel_1d(int ksize, float x, float* c) {
  if (ksize == 4) {  // bicubic, A = -0.75
    const float A = -0.75f;
    c[0] = ((A * (x + 1) - 5 * A) * (x + 1) + 8 * A) * (x + 1) - 4 * A;
    c[1] = ((A + 2) * x - (A + 3)) * x * x + 1;
    c[2] = ((A + 2) * (1 - x) - (A + 3)) * (1 - x) * (1 - x) + 1;
    c[3] = 1.f - c[0] - c[1] - c[2];
    return;
  }
  // Lanczos4: sin(pi x) sin(pi x / 4) / (pi^2 x^2 / 4) evaluated through the angle-sum table OpenCV uses
  static const double s45 = 0.70710678118654752440084436210485;
  static const double cs[8][2] = {{1, 0}, {-s45, -s45}, {0, 1}, {s45, -s45}, {-1, 0}, {s45, s45}, {0, -1}, {-s45, s45}};
  if (x < FLT_EPSILON) {
    for (int i = 0; i < 8; ++i) c[i] = 0.f;
    c[3] = 1.f;
    return;
  }
  float sum = 0.f;
  const double y0 = -(x + 3) * 3.14159265358979323846 * 0.25, s0 = std::sin(y0), c0 = std::cos(y0);
  for (int i = 0; i < 8; ++i) {
    const double y = -(x + 3 - i) * 3.14159265358979323846 * 0.25;
    c[i] = (float)((cs[i][0] * s0 + cs[i][1] * c0) / (y * y));
    sum += c[i];
  }
  sum = 1.f / sum;
  for (int i = 0; i < 8; ++i) c[i] *= sum;
}

// [fy][fx][ksize^2] fixed-point weights, each set corrected to sum to 2^15 on one of the four centre taps
std::vector<int16_t> weight_table(int ksize) {
  std::vector<float> t1((size_t)kTab * ksize);
  for (int i = 0; i < kTab; ++i) kernel_1d(ksize, (float)i * (1.f / kTab), &t1[(size_t)i * ksize]);
  std::vector<int16_t> tab((size_t)kTab * kTab * ksize * ksize);
  for (int fy = 0; fy < kTab; ++fy)
    for (int fx = 0; fx < kTab; ++fx) {
      int16_t* w = &tab[((size_t)fy * kTab + fx) * ksize * ksize];
      int isum = 0;
      for (int r = 0; r < ksize; ++r)
        for (int c = 0; c < ksize; ++c) {
          long q = std::lrintf(t1[(size_t)fy * ksize + r] * t1[(size_t)fx * ksize + c] * (float)kCoefScale);
          q = q > 32767 ? 32767 : (q < -32768 ? -32768 : q);
          w[r * ksize + c] = (int16_t)q;
          isum += (int)q;
        }
      if (isum != kCoefScale) {
        const int diff = isum - kCoefScale, h = ksize / 2;
        int big = h * ksize + h, small = h * ksize + h;
        for (int r = h; r < h + 2; ++r)
          for (int c = h; c < h + 2; ++c) {
            const int i = r * ksize + c;
            if (w[i] < w[small]) small = i;
            else if (w[i] > w[big]) big = i;
          }
        if (diff < 0) w[big] = (int16_t)(w[big] - diff);
        else w[small] = (int16_t)(w[small] - diff);
      }
    }
  return tab;
}

// cv::logPolar's maps in remap's fixed-point form; rows = phi, columns = rho, centre (res/2, res/2)
// (cv::Point2f(resolution / 2, resolution / 2), scaleRotationEstimator.cpp:25). The reference compiles against one of
// two OpenCV generations (scaleRotationEstimator.cpp:41-46, :107-113) whose maps differ:
//   MOF_LOGPOLAR_CV4 (ROS Noetic, OpenCV 4.2) cv::logPolar = cv::warpPolar(.., maxRadius = exp(width / M), WARP_POLAR_LOG):
//     Kmag = log(maxRadius) / width, float table rhos[rho] = (float)(exp(rho * Kmag) - 1.0),
//     x = rhos[rho] * cos((2 pi / height) * phi) + cx, in double, stored as float;
//   MOF_LOGPOLAR_CV3 (ROS Melodic, OpenCV 3.2) cvLogPolar: double table exp(rho / M) -- no "- 1" --,
//     x = exp_tab[rho] * cos(phi * 2 pi / height) + cx.
// remap then rounds the float coordinates to 1/32 px: (anchor, fractional index) per destination pixel.
}  // namespace

namespace mof {

std::vector<SrMapEntry> sr_logpolar_map(int res, double M, int variant) {
  std::vector<SrMapEntry> map((size_t)res * res);
  const float cx = (float)(res / 2), cy = (float)(res / 2);
  const double PI = 3.14159265358979323846;
  std::vector<float> rhos((size_t)res);
  std::vector<double> exp_tab((size_t)res);
  const double Kmag = std::log(std::exp((double)res / M)) / (double)res, Kangle = 2.0 * PI / (double)res;
  for (int rho = 0; rho < res; ++rho) {
    rhos[(size_t)rho] = (float)(std::exp(rho * Kmag) - 1.0);
    exp_tab[(size_t)rho] = std::exp(rho / M);
  }
  for (int phi = 0; phi < res; ++phi) {
    const double ang = variant == MOF_LOGPOLAR_CV3 ? phi * 2 * PI / res : Kangle * phi;
    const double cp = std::cos(ang), sp = std::sin(ang);
    for (int rho = 0; rho < res; ++rho) {
      const double r = variant == MOF_LOGPOLAR_CV3 ? exp_tab[(size_t)rho] : (double)rhos[(size_t)rho];
      const float mx = (float)(r * cp + (double)cx), my = (float)(r * sp + (double)cy);
      const float fx = mx * (float)kTab, fy = my * (float)kTab;
      SrMapEntry e{0, 0, 0, 0};
      if (std::fabs(fx) < 1.0e9f && std::fabs(fy) < 1.0e9f) {
        const long ix = std::lrintf(fx), iy = std::lrintf(fy);
        long ax = ix >> 5, ay = iy >> 5;
        ax = ax > 32767 ? 32767 : (ax < -32768 ? -32768 : ax);
        ay = ay > 32767 ? 32767 : (ay < -32768 ? -32768 : ay);
        e.ax = (int16_t)ax;
        e.ay = (int16_t)ay;
        e.widx = (uint16_t)((iy & (kTab - 1)) * kTab + (ix & (kTab - 1)));
        e.valid = (ax >= 0 && ax < res && ay >= 0 && ay < res) ? 1 : 0;
      }
      map[(size_t)phi * res + rho] = e;
    }
  }
  return map;
}

std::vector<int16_t> sr_weight_table(int ksize) { return weight_table(ksize); }

std::vector<uint32_t> sr_weight_planes(const std::vector<int16_t>& weights, int ksize) {
  const int kk = ksize * ksize, stride = kk / 2 + 2;
  const size_t rows = weights.size() / kk;
  std::vector<uint32_t> out(rows * stride, 0u);
  for (size_t r = 0; r < rows; ++r) {
    uint32_t* o = out.data() + r * stride;
    int sum = 0, rem = -1;
    for (int t = 0; t < kk; ++t) {
      const int w = weights[r * kk + t];
      const int lo = (int)(int8_t)(w & 0xff);
      int hi = (w - lo) >> 8;
      if (hi > 127) {
        hi = 127;  // w >= 32640: at most one tap of a footprint (the weights sum to 2^15)
        rem = t;
      }
      sum += w;
      o[t / 4] |= (uint32_t)(hi & 0xff) << (8 * (t % 4));
      o[kk / 4 + t / 4] |= (uint32_t)(lo & 0xff) << (8 * (t % 4));
    }
    o[kk / 2] = (uint32_t)(128 * sum);
    o[kk / 2 + 1] = (uint32_t)rem;
  }
  return out;
}

std::vector<SrTileBox> sr_tile_boxes(const std::vector<SrMapEntry>& map, int res, int ksize, int* lds_per_wave, int tile_px) {
  const int tiles = (res + tile_px - 1) / tile_px, half = ksize / 2 - 1;
  std::vector<SrTileBox> boxes((size_t)tiles * tiles, SrTileBox{0, 0, 0, 0});
  int worst = 16;
  for (int ty = 0; ty < tiles; ++ty)
    for (int tx = 0; tx < tiles; ++tx) {
      int x0 = 1 << 20, y0 = 1 << 20, x1 = -1, y1 = -1;
      for (int l = 0; l < tile_px * tile_px; ++l) {
        const int rho = tx * tile_px + l % tile_px, phi = ty * tile_px + l / tile_px;
        if (rho >= res || phi >= res) continue;
        const SrMapEntry& m = map[(size_t)phi * res + rho];
        const int sx = m.ax - half, sy = m.ay - half;
        if (!m.valid) continue;
        // rows: every tap row, rows beyond the border reflected back in (BORDER_REFLECT_101; a valid anchor lies inside
        // the image and the kernel reaches at most ksize pixels from it, so one reflection is enough); columns: the
        // ksize-wide window the kernel reads, clamped into the image -- it contains every reflected tap column
        const int wx = sx < 0 ? 0 : (sx > res - ksize ? res - ksize : sx);
        x0 = wx < x0 ? wx : x0;
        x1 = wx + ksize > x1 ? wx + ksize : x1;
        for (int k = 0; k < ksize; ++k) {
          int yy = sy + k;
          yy = yy < 0 ? -yy : yy;
          yy = yy >= res ? 2 * res - 2 - yy : yy;
          y0 = yy < y0 ? yy : y0;
          y1 = yy + 1 > y1 ? yy + 1 : y1;
        }
      }
      if (x1 < 0) continue;
      boxes[(size_t)ty * tiles + tx] = SrTileBox{(int16_t)x0, (int16_t)y0, (int16_t)(x1 - x0), (int16_t)(y1 - y0)};
      const int lpd = ((x1 - x0) + 3 + 3) / 4 + 1;  // LDS row pitch in dwords, as the kernel computes it
      const int bytes = (y1 - y0) * lpd * 4;
      worst = bytes > worst ? bytes : worst;
    }
  *lds_per_wave = (worst + 15) & ~15;
  return boxes;
}

}  // namespace mof

namespace {

struct BusyGuard {
  std::atomic<bool>& flag;
  bool owned;
  explicit BusyGuard(std::atomic<bool>& f) : flag(f), owned(!f.exchange(true)) {}
  ~BusyGuard() {
    if (owned) flag.store(false);
  }
};

}  // namespace

struct mof_sr_engine {
  mof_sr_config cfg{};
  hipStream_t stream = nullptr;
  mof::SrMapEntry* d_map = nullptr;
  mof::SrTileBox* d_boxes[2] = {nullptr, nullptr};  // cubic, Lanczos4
  int lds_per_wave[2] = {0, 0};
  int16_t* d_w_cubic = nullptr;
  int16_t* d_w_lanczos = nullptr;
  uint32_t* d_wp[2] = {nullptr, nullptr};  // byte planes of the two tables (cubic, Lanczos4)
  mof::SrTileBox* d_sboxes[2] = {nullptr, nullptr};  // super-tile boxes (cubic, Lanczos4); null when res % 16 != 0
  int sbox_dwords[2] = {0, 0};
  float* d_twiddles = nullptr;
  uint8_t* d_frame = nullptr;    // staging for the stateful path (res*res)
  uint8_t* d_temp_im = nullptr;  // tempIm, :27
  uint32_t* d_wfrag = nullptr;   // K56 (sr_fused_kernel.hip): f16 hi / lo fragments of the row-DFT matrix, tuned resolutions only
  float* d_zh_prev = nullptr;    // prevIm_F32 (:48, :128), kept as what the correlation needs of it: its row half-spectra
                                 // (sr_seq_kernel.hip, K5s) -- each frame is remapped and row-transformed ONCE
  uint8_t* d_lp = nullptr;       // batch: [kChunk][2][res*res] log-polar images (cur, prev)
  float *d_Zt = nullptr, *d_Dt = nullptr;
  float2* d_cand = nullptr;
  double* d_out = nullptr;       // [kChunk][4]
  int* d_degen = nullptr;        // [kChunk]: K6 -> K8 flag of the pair pipeline (an all-zero log-polar image)
  uint8_t* h_stage = nullptr;
  double* h_out = nullptr;
  double* h_seq = nullptr;       // pinned [chunk][4]: a pass's results, read back when a sequence call resolves the gate
  int seq_run = 0;               // pairs one wave of K6s walks in time: 0 = chosen per pass (seq_run_for), MOF_SR_SEQ_RUN fixes it (r05: 16)
  int chunk = 0;                 // frame pairs per pipeline pass
  int scratch_pairs = 0;         // pairs per pass the scratch holds now (1 after create, `chunk` after the first batch)
  bool two_lanes = false;        // remap of pass k+1 beside the transforms of pass k (mof_sr_config.pipeline_lanes == 2)
  bool first = true;             // :31
  // resolutions without hand-tuned transforms (anything but 240 / 256 / 480) run the planned pipeline of
  // pc_large_kernel.hip on the size cv::phaseCorrelate pads to, plan.m = getOptimalDFTSize(resolution)
  bool generic = false;
  mof::PcPlan plan{};
  // r06: a generic resolution whose PADDED size plan.m has tuned transforms runs K5s (the zero-padding frame form of the row kernel) / K6s /
  // K7 and only the final kernel of the planned pipeline. pad_sums: plan.m is 250 / 400 / 432 (odd last radix) -- the four exact pixel sums of
  // a frame sit `sums_off` floats into its Zh slot (the slack behind the padded rows), so they travel wherever the slot is copied
  bool tuned_pad = false, pad_sums = false;
  size_t sums_off = 0;
  std::atomic<bool> busy{false};
  std::mutex host_mu;  // mof_sr_process_sequence_host: the upload pipeline (host_pipe.hpp), made by its first call
  mof::HostPipe* host_pipe = nullptr;
  // a batch call was captured into a HIP graph: the graph's kernel nodes hold raw pointers into the scratch below, so
  // from then on the scratch neither grows nor is freed until mof_sr_release_graphs (capi_graph.hpp)
  std::atomic<bool> graph_pinned{false};
  // Ordering of the engine-owned scratch (d_lp, d_Zt, d_Dt, d_cand, d_out) across streams: every call that
  // touches it records `scratch_ev` behind its last kernel; a later call on a DIFFERENT stream first makes its
  // stream wait for that event (same-stream calls are ordered by the stream itself).
  hipEvent_t scratch_ev = nullptr;
  hipStream_t scratch_stream = nullptr;  // stream of the last user
  bool scratch_used = false;
  // Two-lane batch pipeline: the log-polar remaps of chunk k+1 (bound by LDS / L1 latency, little HBM traffic) run on
  // `remap_stream` while the transforms of chunk k (bound by HBM) run on the caller's stream; the log-polar images are
  // double-buffered and handed over with events.
  hipStream_t remap_stream = nullptr;
  hipEvent_t ev_fork = nullptr;
  hipEvent_t ev_lp[2] = {nullptr, nullptr};   // remaps of the chunk using buffer b are done
  hipEvent_t ev_fft[2] = {nullptr, nullptr};  // transforms reading buffer b are done (it may be overwritten)
};

namespace {

// sizes of the transform scratch and the three transform steps, tuned or planned
size_t zh_floats(const mof_sr_engine* e) { return e->generic ? mof::pcl_zh_floats(e->plan) : mof::sr_zh_floats(e->cfg.resolution); }
int peak_candidates(const mof_sr_engine* e) { return e->generic ? mof::pcl_candidates(e->plan) : mof::sr_candidates(e->cfg.resolution); }
hipError_t rows_real(const mof_sr_engine* e, const uint8_t* lp, size_t lp_stride, float* zh, size_t zh_stride, int n_frames, hipStream_t s) {
  const int res = e->cfg.resolution;
  if (!e->generic) return mof::launch_sr_rows_real(lp, lp_stride, e->d_twiddles, zh, zh_stride, res, n_frames, s);
  mof::PclSrc src{};
  src.base[0] = lp;
  src.stride[0] = lp_stride;
  src.pitch = (size_t)res;  // log-polar images are tightly packed
  if (e->tuned_pad) {
    src.paired = 2;  // "a video whose one patch is the whole image": image f = base[0] + f * stride[0]
    src.grid_x = src.grid_y = 1;
    src.stride_x = src.stride_y = res;
    int* sums = nullptr;
    if (e->pad_sums) {
      sums = reinterpret_cast<int*>(zh + e->sums_off);
      src.sums_stride = (int)zh_stride;
      const hipError_t err = n_frames == 1 ? hipMemsetAsync(sums, 0, 4 * sizeof(int), s)  // (the row kernel adds into them; the stateful call passes no stride)
                                           : hipMemset2DAsync(sums, zh_stride * sizeof(float), 0, 4 * sizeof(int), (size_t)n_frames, s);
      if (err != hipSuccess) return err;
    }
    return mof::launch_sr_rows_real_src(src, e->d_twiddles, zh, zh_stride, nullptr, e->plan.m, n_frames, 1, res, s, sums);
  }
  return mof::launch_pcl_rows(src, e->plan, e->d_twiddles, zh, zh_stride, nullptr, n_frames, 1, 1, s);
}
// K56: K5s + K6s in one kernel on the u8 log-polar images (MOF_SR_FUSED, tuned resolutions)
bool fused_requested() {
  static const bool on = [] { const char* v = getenv("MOF_SR_FUSED"); return v && atoi(v) != 0; }();
  return on;
}
bool use_fused(const mof_sr_engine* e) { return e->d_wfrag != nullptr; }  // (the fragments exist only when the knob was set at create)
hipError_t cols_fused(const mof_sr_engine* e, const uint8_t* lp_prev, const uint8_t* lp_cur, size_t lp_stride, int n_pairs, int run, hipStream_t s) {
  return mof::launch_sr_cols_fused(lp_prev, lp_cur, lp_stride, e->d_wfrag, e->d_twiddles, e->d_Dt, e->cfg.resolution, n_pairs, run, s);
}
// Run length of K6s for a pass of m consecutive pairs (r06; tools/video_probe.py showed the same shape on the FFT engine's video kernel): a
// wave walks `run` pairs one after the other, so a short video in runs of 16 leaves most of the chip idle -- 64 frames of 480^2 are 61 column
// groups x 4 runs = 244 one-wave workgroups on 2048 slots. Workgroups last run + ~0.6 column transforms (the first pair's previous spectra
// are transformed too), the launch lasts ceil(workgroups / slots) rounds: the shortest run within 3 % of the least product.
int seq_run_for(const mof_sr_engine* e, int m) {
  if (e->seq_run > 0) return e->seq_run;
  int cus = 0;
  if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, e->cfg.device) != hipSuccess || cus <= 0) cus = 256;
  const int tn = e->generic ? e->plan.m : e->cfg.resolution;
  const int cw = (tn >= 540 || tn == 324 || tn == 486 || tn == 500) ? 2 : 4;  // columns per wave (sr_seq_kernel.hip: seq_cw -- first radix above 16)
  const long groups = (tn / 2 + 1 + cw - 1) / cw, slots = (long)cus * 8;  // one-wave workgroups, two per SIMD
  auto cost = [&](int r) {
    const long wgs = groups * ((m + r - 1) / r), rounds = (wgs + slots - 1) / slots;
    return (double)rounds * ((double)r + 0.6);
  };
  double best = cost(1);
  for (int r = 2; r <= 32; ++r) best = cost(r) < best ? cost(r) : best;
  for (int r = 1; r <= 32; ++r)
    if (cost(r) <= 1.03 * best) return r;
  return 16;
}
hipError_t cols_seq(const mof_sr_engine* e, const float* zh_prev, const float* zh_cur, size_t zh_stride, int n_pairs, int run, hipStream_t s) {
  if (!e->generic) return mof::launch_sr_cols_seq(zh_prev, zh_cur, zh_stride, e->d_twiddles, e->d_Dt, e->cfg.resolution, n_pairs, run, s);
  if (e->tuned_pad)
    return mof::launch_sr_cols_seq(zh_prev, zh_cur, zh_stride, e->d_twiddles, e->d_Dt, e->plan.m, n_pairs, run, s, nullptr, e->cfg.resolution,
                                   e->pad_sums ? reinterpret_cast<const int*>(zh_prev + e->sums_off) : nullptr,
                                   e->pad_sums ? reinterpret_cast<const int*>(zh_cur + e->sums_off) : nullptr, (int)zh_stride);
  return mof::launch_pcl_cols(zh_prev, zh_cur, zh_stride, e->plan, e->d_twiddles, e->d_Dt, nullptr, nullptr, n_pairs, s);
}
hipError_t peak(const mof_sr_engine* e, const mof::SrPcArgs& a, int n_pairs, hipStream_t s) {
  if (!e->generic) return mof::launch_sr_peak(a, e->cfg.resolution, n_pairs, s);
  mof::PclFinal f{};
  f.Dt = a.Dt;
  f.cand = a.cand;
  f.twiddles = a.twiddles;
  f.mode = 0;
  f.M_log = a.M;
  f.out = a.out;
  if (e->tuned_pad) {  // K7 forms the candidates; L8 (the planned pipeline's final kernel: it knows the padded geometry) reads them
    const hipError_t err = mof::launch_sr_rows_inv(a.Dt, a.twiddles, a.cand, e->plan.m, n_pairs, s);
    if (err != hipSuccess) return err;
    return mof::launch_pcl_peak(f, e->plan, n_pairs, s, true);
  }
  return mof::launch_pcl_peak(f, e->plan, n_pairs, s);
}

// Called with the busy flag held, before the first launch that reads or writes the scratch.
hipError_t scratch_acquire(mof_sr_engine* e, hipStream_t s) {
  if (e->scratch_used && e->scratch_stream != s) {
    hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
    hipError_t err = hipStreamIsCapturing(s, &cap);
    if (err != hipSuccess) return err;
    // a capturing stream must not take a dependency on work outside its graph: the caller orders graph replays
    if (cap == hipStreamCaptureStatusNone) {
      err = hipStreamWaitEvent(s, e->scratch_ev, 0);
      if (err != hipSuccess) return err;
    }
  }
  return hipSuccess;
}

// Called behind the last launch of the call.
hipError_t scratch_release(mof_sr_engine* e, hipStream_t s) {
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  hipError_t err = hipStreamIsCapturing(s, &cap);
  if (err != hipSuccess) return err;
  if (cap != hipStreamCaptureStatusNone) return hipSuccess;  // events recorded while capturing belong to the graph
  err = hipEventRecord(e->scratch_ev, s);
  if (err != hipSuccess) return err;
  e->scratch_stream = s;
  e->scratch_used = true;
  return hipSuccess;
}

// The pipeline scratch (log-polar images of two passes, Zt, Dt, peak candidates, results) for `pairs` pairs per pass.
hipError_t scratch_alloc(mof_sr_engine* e, int pairs) {
  mof::RelaxedCapture relaxed;  // allocation / release must not invalidate a capture on another thread
  const int res = e->cfg.resolution;
  const size_t nn = (size_t)res * res;
  void** bufs[] = {(void**)&e->d_lp, (void**)&e->d_Zt, (void**)&e->d_Dt, (void**)&e->d_cand, (void**)&e->d_out, (void**)&e->d_degen};
  for (void** b : bufs) {
    if (*b) (void)hipFree(*b);
    *b = nullptr;
  }
  e->scratch_pairs = 0;
  hipError_t err;
  if ((err = hipMalloc(&e->d_lp, (size_t)2 * pairs * 2 * nn)) != hipSuccess) return err;  // two passes: remap of pass k+1 beside the transforms of pass k
  {
    // pair pipeline: packed row spectra Zt, nn complex per pair; sequence pipeline: (pairs + 1) frames of half spectra
    // (or, pairs through the frame kernels: 2 * pairs frames of half spectra)
    const size_t zt = e->generic ? 0 : (size_t)pairs * nn * 2 * sizeof(float), zh1 = (size_t)(pairs + 1) * zh_floats(e) * sizeof(float),
                 zh2 = (size_t)2 * pairs * zh_floats(e) * sizeof(float), zh = zh1 > zh2 ? zh1 : zh2;
    if ((err = hipMalloc(&e->d_Zt, zt > zh ? zt : zh)) != hipSuccess) return err;
  }
  if ((err = hipMalloc(&e->d_Dt, (size_t)pairs * zh_floats(e) * sizeof(float))) != hipSuccess) return err;  // Dt has Zh's shape
  if ((err = hipMalloc(&e->d_cand, (size_t)pairs * peak_candidates(e) * sizeof(float2))) != hipSuccess) return err;
  if ((err = hipMalloc(&e->d_out, (size_t)pairs * 4 * sizeof(double))) != hipSuccess) return err;
  if ((err = hipMalloc(&e->d_degen, (size_t)pairs * sizeof(int))) != hipSuccess) return err;
  e->scratch_pairs = pairs;
  return hipSuccess;
}

// Pairs per pass a batch of n pairs needs: n rounded up to a power of two (growing batches re-allocate O(log) times),
// a whole pass at most. mof_sr_reserve and the batch entry use the same rule.
int scratch_want(const mof_sr_engine* e, int n_pairs) {
  int want = 1;
  while (want < n_pairs && want < e->chunk) want <<= 1;
  return want < e->chunk ? want : e->chunk;
}

// Makes sure the scratch holds a pass of `pairs` pairs. Growing frees and re-allocates: not possible while `s` is
// being captured into a graph (run one batch, or mof_sr_reserve, before the capture), and only after every earlier
// user of the scratch has finished. Returns a MOF status.
int scratch_reserve(mof_sr_engine* e, int pairs, hipStream_t s) {
  if (pairs <= e->scratch_pairs) return MOF_OK;
  if (e->graph_pinned.load())
    return mof::capi_fail(MOF_ERR_BUSY, "the estimator's scratch would have to grow from %d to %d pairs per pass, but a captured HIP graph "
                                         "still points into it: reserve the largest batch before capturing, or call "
                                         "mof_sr_release_graphs once the graphs are gone", e->scratch_pairs, pairs);
  hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
  if (hipStreamIsCapturing(s, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "the estimator's scratch must grow to %d pairs per pass, which cannot happen inside a graph capture: "
                                            "call mof_sr_reserve (or run one batch) before capturing", pairs);
  if (e->scratch_used) (void)hipEventSynchronize(e->scratch_ev);
  (void)hipStreamSynchronize(e->stream);
  (void)hipStreamSynchronize(e->remap_stream);
  const hipError_t err = scratch_alloc(e, pairs);
  if (err != hipSuccess) {
    (void)scratch_alloc(e, 1);  // keep the stateful entry usable
    return mof::capi_fail(err == hipErrorOutOfMemory ? MOF_ERR_NO_MEMORY : MOF_ERR_HIP, "scale/rotation scratch for %d pairs: %s", pairs,
                          hipGetErrorString(err));
  }
  return MOF_OK;
}

}  // namespace

extern "C" {

int mof_sr_reserve(mof_sr_engine* e, int n_pairs) {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs < 0) return mof::capi_fail(MOF_ERR_BAD_ARG, "n_pairs must be >= 0");
  BusyGuard g(e->busy);
  if (!g.owned) return mof::capi_fail(MOF_ERR_BUSY, "engine busy");
  if (hipSetDevice(e->cfg.device) != hipSuccess) return mof::capi_fail(MOF_ERR_HIP, "hipSetDevice failed");
  return scratch_reserve(e, scratch_want(e, n_pairs), e->stream);
}

static void sr_destroy_now(void* p) {
  mof_sr_engine* e = static_cast<mof_sr_engine*>(p);
  mof::RelaxedCapture relaxed;
  (void)hipSetDevice(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->scratch_ev && e->scratch_used) (void)hipEventSynchronize(e->scratch_ev);  // a batch on a caller's stream may still use the scratch
  delete e->host_pipe;
  void* dev[] = {e->d_boxes[0], e->d_boxes[1], e->d_sboxes[0], e->d_sboxes[1], e->d_map, e->d_w_cubic, e->d_w_lanczos, e->d_wp[0], e->d_wp[1], e->d_twiddles, e->d_frame, e->d_temp_im, e->d_zh_prev, e->d_wfrag,
                 e->d_lp,  e->d_Zt,      e->d_Dt,        e->d_cand,     e->d_out, e->d_degen};
  for (void* p : dev)
    if (p) (void)hipFree(p);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->h_out) (void)hipHostFree(e->h_out);
  if (e->h_seq) (void)hipHostFree(e->h_seq);
  if (e->scratch_ev) (void)hipEventDestroy(e->scratch_ev);
  if (e->remap_stream) (void)hipStreamSynchronize(e->remap_stream);
  for (hipEvent_t ev : {e->ev_fork, e->ev_lp[0], e->ev_lp[1], e->ev_fft[0], e->ev_fft[1]})
    if (ev) (void)hipEventDestroy(ev);
  if (e->remap_stream) (void)hipStreamDestroy(e->remap_stream);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

void mof_sr_destroy(mof_sr_engine* e) {
  if (!e) return;
  if (e->graph_pinned.load()) {  // replays of a captured graph would run through freed scratch: park until released
    mof::park_engine(&sr_destroy_now, e);
    return;
  }
  sr_destroy_now(e);
}

int mof_sr_release_graphs(mof_sr_engine* e) {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  e->graph_pinned.store(false);
  return MOF_OK;
}

int mof_sr_graph_pinned(const mof_sr_engine* e) { return e && e->graph_pinned.load() ? 1 : 0; }

int mof_sr_create(const mof_sr_config* cfg, mof_sr_engine** out) try {
  if (!out) return mof::capi_fail(MOF_ERR_BAD_ARG, "null out");
  *out = nullptr;
  if (!cfg || !(cfg->magnitude > 0.0)) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad scale/rotation config");
  if (cfg->logpolar_variant != MOF_LOGPOLAR_CV4 && cfg->logpolar_variant != MOF_LOGPOLAR_CV3)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "logpolar_variant must be MOF_LOGPOLAR_CV4 (0) or MOF_LOGPOLAR_CV3 (1)");
  if (cfg->batch_chunk < 0 || cfg->batch_chunk > 4096 || cfg->pipeline_lanes < 0 || cfg->pipeline_lanes > 2)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "batch_chunk must be 0..4096 and pipeline_lanes 0..2");
  // any even resolution: scaleRotationEstimator takes `res` from a parameter (scaleRotationEstimator.cpp:3-5); 240 / 256 / 480
  // have hand-tuned transforms, the rest run the planned pipeline on the size cv::phaseCorrelate pads to
  if (cfg->resolution < 16 || (cfg->resolution & 1))
    return mof::capi_fail(MOF_ERR_BAD_ARG, "resolution %d: an even resolution >= 16 is required", cfg->resolution);
  mof::PcPlan plan{};
  const bool generic = !mof::sr_resolution_supported(cfg->resolution);
  if (generic && !mof::pc_build_line_plan(cfg->resolution, &plan))
    return mof::capi_fail(MOF_ERR_UNSUPPORTED, "resolution %d pads to %d: beyond the planned transforms (<= 960)", cfg->resolution,
                          mof::pc_optimal_dft_size(cfg->resolution));
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
    (void)hipGetLastError();
    return mof::capi_fail(MOF_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  }
  if (cfg->device < 0 || cfg->device >= ndev) return mof::capi_fail(MOF_ERR_BAD_ARG, "device %d out of range", cfg->device);
  SR_TRY(hipSetDevice(cfg->device));
  mof::RelaxedCapture relaxed;  // allocating an engine must not invalidate a capture on another thread
  const int res = cfg->resolution;
  const size_t nn = (size_t)res * res;
  const std::vector<mof::SrMapEntry> map = mof::sr_logpolar_map(res, cfg->magnitude, cfg->logpolar_variant);
  const std::vector<int16_t> wc = weight_table(4), wl = weight_table(8);
  int lds_c = 0, lds_l = 0;
  const std::vector<mof::SrTileBox> bc = mof::sr_tile_boxes(map, res, 4, &lds_c, 8), bl = mof::sr_tile_boxes(map, res, 8, &lds_l, 8);
  // boxes of the 16 x 16 super-tiles (one workgroup = four tiles sharing a staged box)
  int slds_c = 0, slds_l = 0;
  std::vector<mof::SrTileBox> sbc, sbl;
  if (res % 16 == 0) {
    sbc = mof::sr_tile_boxes(map, res, 4, &slds_c, 16);
    sbl = mof::sr_tile_boxes(map, res, 8, &slds_l, 16);
  }
  const int tn = generic ? plan.m : res;  // transform size: the planned pipeline works on the padded image
  std::vector<float> tw(2 * (size_t)tn);
  mof_sr_engine* e = new (std::nothrow) mof_sr_engine();
  if (!e) return mof::capi_fail(MOF_ERR_NO_MEMORY, "out of host memory");
  e->cfg = *cfg;
  e->generic = generic;
  e->plan = plan;
  if (generic) {
    static const bool all = [] { const char* v = getenv("MOF_SR_TUNED_ALL"); return !v || atoi(v) != 0; }();
    bool exact = true;
    if (all && mof::sr_transform_size_tuned(plan.m, &exact)) {
      e->tuned_pad = true;
      e->pad_sums = !exact;
      e->sums_off = (size_t)((plan.m >> 1) + 1) * ((plan.m + 7) & ~7) * 2;  // behind the padded rows; pcl_zh_floats leaves 16 floats per row of slack
    }
  }
  e->chunk = chunk_pairs(cfg->batch_chunk);
  e->two_lanes = two_lane_default(cfg->pipeline_lanes);
  for (int k = 0; k < tn; ++k) {
    double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)tn;
    double c = std::cos(ang), s = std::sin(ang);
    if ((4 * k) % tn == 0) {
      const int q = (4 * k) / tn;
      c = (q == 0) ? 1.0 : (q == 2) ? -1.0 : 0.0;
      s = (q == 1) ? -1.0 : (q == 3) ? 1.0 : 0.0;
    }
    tw[2 * (size_t)k] = (float)c;
    tw[2 * (size_t)k + 1] = (float)s;
  }
#define CREATE_TRY(expr)                                                                  \
  do {                                                                                    \
    hipError_t _e = (expr);                                                               \
    if (_e != hipSuccess) {                                                               \
      mof::capi_fail(MOF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));                \
      mof_sr_destroy(e);                                                                  \
      return MOF_ERR_HIP;                                                                 \
    }                                                                                     \
  } while (0)
  CREATE_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  CREATE_TRY(hipEventCreateWithFlags(&e->scratch_ev, hipEventDisableTiming));
  CREATE_TRY(hipStreamCreateWithFlags(&e->remap_stream, hipStreamNonBlocking));
  CREATE_TRY(hipEventCreateWithFlags(&e->ev_fork, hipEventDisableTiming));
  for (int b = 0; b < 2; ++b) {
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_lp[b], hipEventDisableTiming));
    CREATE_TRY(hipEventCreateWithFlags(&e->ev_fft[b], hipEventDisableTiming));
  }
  CREATE_TRY(hipMalloc(&e->d_map, map.size() * sizeof(mof::SrMapEntry)));
  CREATE_TRY(mof::copy_on(e->stream, e->d_map, map.data(), map.size() * sizeof(mof::SrMapEntry), hipMemcpyHostToDevice));
  e->lds_per_wave[0] = lds_c;
  e->lds_per_wave[1] = lds_l;
  CREATE_TRY(hipMalloc(&e->d_boxes[0], bc.size() * sizeof(mof::SrTileBox)));
  CREATE_TRY(mof::copy_on(e->stream, e->d_boxes[0], bc.data(), bc.size() * sizeof(mof::SrTileBox), hipMemcpyHostToDevice));
  CREATE_TRY(hipMalloc(&e->d_boxes[1], bl.size() * sizeof(mof::SrTileBox)));
  CREATE_TRY(mof::copy_on(e->stream, e->d_boxes[1], bl.data(), bl.size() * sizeof(mof::SrTileBox), hipMemcpyHostToDevice));
  if (!sbc.empty()) {
    e->sbox_dwords[0] = slds_c / 4;
    e->sbox_dwords[1] = slds_l / 4;
    if (const char* v = getenv("MOF_SR_VERBOSE"); v && atoi(v) != 0)  // diagnostics: the staged remap's largest boxes (dwords)
      fprintf(stderr, "mof_sr: res %d: largest super-tile box cubic %d, lanczos4 %d dwords; per-wave boxes %d / %d bytes\n", res, e->sbox_dwords[0],
              e->sbox_dwords[1], lds_c, lds_l);
    CREATE_TRY(hipMalloc(&e->d_sboxes[0], sbc.size() * sizeof(mof::SrTileBox)));
    CREATE_TRY(mof::copy_on(e->stream, e->d_sboxes[0], sbc.data(), sbc.size() * sizeof(mof::SrTileBox), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&e->d_sboxes[1], sbl.size() * sizeof(mof::SrTileBox)));
    CREATE_TRY(mof::copy_on(e->stream, e->d_sboxes[1], sbl.data(), sbl.size() * sizeof(mof::SrTileBox), hipMemcpyHostToDevice));
  }
  CREATE_TRY(hipMalloc(&e->d_w_cubic, wc.size() * sizeof(int16_t)));
  CREATE_TRY(mof::copy_on(e->stream, e->d_w_cubic, wc.data(), wc.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  CREATE_TRY(hipMalloc(&e->d_w_lanczos, wl.size() * sizeof(int16_t)));
  CREATE_TRY(mof::copy_on(e->stream, e->d_w_lanczos, wl.data(), wl.size() * sizeof(int16_t), hipMemcpyHostToDevice));
  {
    const std::vector<uint32_t> pc = mof::sr_weight_planes(wc, 4), pl = mof::sr_weight_planes(wl, 8);
    CREATE_TRY(hipMalloc(&e->d_wp[0], pc.size() * sizeof(uint32_t)));
    CREATE_TRY(mof::copy_on(e->stream, e->d_wp[0], pc.data(), pc.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
    CREATE_TRY(hipMalloc(&e->d_wp[1], pl.size() * sizeof(uint32_t)));
    CREATE_TRY(mof::copy_on(e->stream, e->d_wp[1], pl.data(), pl.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  CREATE_TRY(hipMalloc(&e->d_twiddles, tw.size() * sizeof(float)));
  CREATE_TRY(mof::copy_on(e->stream, e->d_twiddles, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
  if (!e->generic && fused_requested() && mof::sr_fused_supported(res)) {  // (1 MB of f16 matrix fragments at 480: only for the opt-in path)
    const std::vector<uint32_t> fr = mof::sr_fused_fragments(res);
    CREATE_TRY(hipMalloc(&e->d_wfrag, fr.size() * sizeof(uint32_t)));
    CREATE_TRY(mof::copy_on(e->stream, e->d_wfrag, fr.data(), fr.size() * sizeof(uint32_t), hipMemcpyHostToDevice));
  }
  CREATE_TRY(hipMalloc(&e->d_frame, nn));
  CREATE_TRY(hipMalloc(&e->d_temp_im, nn));
  CREATE_TRY(hipMalloc(&e->d_zh_prev, zh_floats(e) * sizeof(float)));
  CREATE_TRY(mof::fill_on(e->stream, e->d_temp_im, 0, nn));  // tempIm = cv::Mat::zeros, :27
  CREATE_TRY(mof::fill_on(e->stream, e->d_zh_prev, 0, zh_floats(e) * sizeof(float)));
  CREATE_TRY(scratch_alloc(e, 1));  // the stateful call needs one pair; a batch grows it to a whole pass (scratch_reserve)
  CREATE_TRY(hipHostMalloc(&e->h_stage, nn, hipHostMallocDefault));
  CREATE_TRY(hipHostMalloc(&e->h_out, 4 * sizeof(double), hipHostMallocDefault));
  CREATE_TRY(hipHostMalloc(&e->h_seq, (size_t)e->chunk * 4 * sizeof(double), hipHostMallocDefault));
  {
    const char* v = getenv("MOF_SR_SEQ_RUN");
    if (v && atoi(v) >= 1 && atoi(v) <= 4096) e->seq_run = atoi(v);
  }
#undef CREATE_TRY
  *out = e;
  return MOF_OK;
} catch (const std::bad_alloc&) {
  return mof::capi_fail(MOF_ERR_NO_MEMORY, "mof_sr_create: out of host memory");
}

int mof_sr_reset(mof_sr_engine* e) {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  BusyGuard g(e->busy);
  if (!g.owned) return mof::capi_fail(MOF_ERR_BUSY, "engine busy");
  SR_TRY(hipSetDevice(e->cfg.device));
  SR_TRY(hipMemsetAsync(e->d_temp_im, 0, (size_t)e->cfg.resolution * e->cfg.resolution, e->stream));
  SR_TRY(hipStreamSynchronize(e->stream));
  e->first = true;
  return MOF_OK;
}

static void lp_tables(const mof_sr_engine* e, int interp, mof::SrLpArgs* lp) {
  const int k = interp == 2 ? 0 : 1;
  lp->weights = k == 0 ? e->d_w_cubic : e->d_w_lanczos;
  lp->wplanes = e->d_wp[k];
  lp->sboxes = e->d_sboxes[k];
  lp->sbox_dwords_max = e->sbox_dwords[k];
  lp->boxes = e->d_boxes[k];
  lp->lds_per_wave = e->lds_per_wave[k];
  lp->box_dwords_max = e->lds_per_wave[k] / 4;
}

static mof::SrPcArgs pc_args(const mof_sr_engine* e, const uint8_t* lp_cur, const uint8_t* lp_prev, size_t lp_stride,
                             double* out) {
  mof::SrPcArgs a{};
  a.lp_cur = lp_cur;
  a.lp_prev = lp_prev;
  a.lp_stride = lp_stride;
  a.twiddles = e->d_twiddles;
  a.Zt = e->d_Zt;
  a.Dt = e->d_Dt;
  a.cand = e->d_cand;
  a.n_cand = peak_candidates(e);
  a.M = e->cfg.magnitude;
  a.out = out;
  return a;
}

// One pair through the sequence kernels: (cur spectra, prev spectra) -> Dt slot 0 -> (scale, rot, pt) at `out`
static hipError_t seq_one_pair(mof_sr_engine* e, const float* zh_prev, const float* zh_cur, double* out, hipStream_t s) {
  hipError_t err = cols_seq(e, zh_prev, zh_cur, 0, 1, 1, s);
  if (err != hipSuccess) return err;
  mof::SrPcArgs a = pc_args(e, nullptr, nullptr, 0, out);
  return peak(e, a, 1, s);
}

int mof_sr_process(mof_sr_engine* e, const uint8_t* frame, size_t pitch, double* out_scale_rot) {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  const int res = e->cfg.resolution;
  if (!frame || !out_scale_rot || pitch < (size_t)res) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad frame/pitch/out");
  BusyGuard g(e->busy);
  if (!g.owned) return mof::capi_fail(MOF_ERR_BUSY, "engine busy");
  SR_TRY(hipSetDevice(e->cfg.device));
  const size_t nn = (size_t)res * res, zh_bytes = zh_floats(e) * sizeof(float);
  for (int y = 0; y < res; ++y) std::memcpy(e->h_stage + (size_t)y * res, frame + (size_t)y * pitch, (size_t)res);
  SR_TRY(hipMemcpyAsync(e->d_frame, e->h_stage, nn, hipMemcpyHostToDevice, e->stream));
  mof::SrLpArgs lp{};
  lp.src = e->d_frame;
  lp.src_stride = 0;
  lp.pitch = (size_t)res;
  lp.dst = e->d_temp_im;
  lp.dst_stride = 0;
  lp.map = e->d_map;
  lp.res = res;
  const int interp = e->first ? 2 : 4;  // INTER_CUBIC for the very first frame (:45), INTER_LANCZOS4 from then on (:112)
  lp_tables(e, interp, &lp);
  SR_TRY(mof::launch_sr_logpolar(lp, interp, 1, e->stream));
  SR_TRY(scratch_acquire(e, e->stream));
  // tempIm.convertTo(CV_32FC1) (:47, :115) + the row half of the forward DFT of cv::phaseCorrelate (:117): K5s
  SR_TRY(rows_real(e, e->d_temp_im, 0, e->d_Zt, 0, 1, e->stream));
  if (e->first) {
    SR_TRY(hipMemcpyAsync(e->d_zh_prev, e->d_Zt, zh_bytes, hipMemcpyDeviceToDevice, e->stream));  // prevIm_F32 = .., :48
    SR_TRY(scratch_release(e, e->stream));
    SR_TRY(hipStreamSynchronize(e->stream));
    e->first = false;  // :73
    out_scale_rot[0] = 1.0;
    out_scale_rot[1] = 0.0;  // :74
    return MOF_OK;
  }
  SR_TRY(seq_one_pair(e, e->d_zh_prev, e->d_Zt, e->d_out, e->stream));  // :117
  SR_TRY(hipMemcpyAsync(e->h_out, e->d_out, 4 * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  SR_TRY(hipStreamSynchronize(e->stream));
  out_scale_rot[0] = e->h_out[0];
  out_scale_rot[1] = e->h_out[1];
  // the reference returns early on the gate, BEFORE prevIm_F32 = tempIm_F32.clone() (:119-121 vs :128)
  if (!(std::fabs(e->h_out[2]) > (double)(res / 2)))
    SR_TRY(hipMemcpyAsync(e->d_zh_prev, e->d_Zt, zh_bytes, hipMemcpyDeviceToDevice, e->stream));
  SR_TRY(scratch_release(e, e->stream));
  SR_TRY(hipStreamSynchronize(e->stream));
  return MOF_OK;
}

// A video on the device = n_frames consecutive mof_sr_process calls (see mof.h). Per pass of up to `chunk` new frames the
// scratch holds slot 0 = the spectra the first new frame is correlated with, slots 1..m = the new frames.
int mof_sr_process_sequence_device(mof_sr_engine* e, const uint8_t* d_frames, size_t frame_stride, size_t pitch, int n_frames,
                                   double* d_out, void* stream, int* n_gated) {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  const int res = e->cfg.resolution;
  if (n_gated) *n_gated = 0;
  if (n_frames == 0) return MOF_OK;
  if (!d_frames || !d_out || n_frames < 0 || pitch < (size_t)res) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad sequence arguments");
  BusyGuard g(e->busy);
  if (!g.owned) return mof::capi_fail(MOF_ERR_BUSY, "engine busy");
  SR_TRY(hipSetDevice(e->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const bool capturing = mof::stream_capturing(s);
  if (capturing && n_gated)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "resolving the gate reads results back on the host: pass n_gated = NULL while capturing");
  const size_t nn = (size_t)res * res, zhf = zh_floats(e), zh_bytes = zhf * sizeof(float);
  {
    const int rc = scratch_reserve(e, scratch_want(e, n_frames), s);
    if (rc != MOF_OK) return rc;
  }
  // A fresh estimator's first frame takes the INTER_CUBIC branch (:45) ONCE; captured, that branch would be baked into the
  // graph and every replay would redo it while the engine believes it is in its steady state. Capture on an armed engine.
  if (capturing && e->first)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "capturing a sequence on a fresh estimator would bake its one-off first-frame branch "
                                            "(INTER_CUBIC, scaleRotationEstimator.cpp:45) into the graph: process one frame first");
  SR_TRY(scratch_acquire(e, s));
  if (capturing) e->graph_pinned.store(true);
  const int C = e->scratch_pairs < e->chunk ? e->scratch_pairs : e->chunk;  // new frames per pass
  float* zh = e->d_Zt;
  // The host-side state (`first`, and the device-side prevIm_F32 = d_zh_prev, whose update is the LAST thing enqueued) is
  // committed only when every launch of the call went out: a call that fails half-way leaves the estimator exactly as it was,
  // and the scratch is released on that path too.
  bool first = e->first;
  int gated_total = 0;
  const int rc = [&]() -> int {
    mof::SrLpArgs lp{};
    lp.zero_invalid = 1;  // tempIm starts as zeros (:27) and transparent pixels never change: write the zeros here
    lp.pitch = pitch;
    lp.map = e->d_map;
    lp.res = res;
    lp.src_stride = frame_stride;
    lp.dst_stride = nn;
    int done = 0;
    int carry = -1;  // slot of the previous pass whose spectra are `prev` for the next frame (-1: the engine's state)
    while (done < n_frames) {
      if (first) {  // the very first frame: INTER_CUBIC (:45), becomes prev (:48), returns (1, 0) (:74)
        lp.src = d_frames + (size_t)done * frame_stride;
        lp.dst = e->d_lp;
        lp_tables(e, 2, &lp);
        SR_TRY(mof::launch_sr_logpolar(lp, 2, 1, s));
        SR_TRY(rows_real(e, e->d_lp, nn, zh, zhf, 1, s));
        SR_TRY(mof::launch_sr_identity(d_out + 4 * (size_t)done, s));
        first = false;  // :73
        ++done;
      } else if (carry < 0) {
        SR_TRY(hipMemcpyAsync(zh, e->d_zh_prev, zh_bytes, hipMemcpyDeviceToDevice, s));
      } else if (carry > 0) {
        SR_TRY(hipMemcpyAsync(zh, zh + (size_t)carry * zhf, zh_bytes, hipMemcpyDeviceToDevice, s));
      }
      const int m = n_frames - done < C ? n_frames - done : C;
      carry = 0;
      if (m > 0) {
        lp.src = d_frames + (size_t)done * frame_stride;
        lp.dst = e->d_lp + nn;
        lp_tables(e, 4, &lp);
        SR_TRY(mof::launch_sr_logpolar(lp, 4, m, s));  // INTER_LANCZOS4, :112 -- every frame once
        SR_TRY(rows_real(e, e->d_lp + nn, nn, zh + zhf, zhf, m, s));
        SR_TRY(cols_seq(e, zh, zh + zhf, zhf, m, seq_run_for(e, m), s));
        mof::SrPcArgs a = pc_args(e, nullptr, nullptr, 0, d_out + 4 * (size_t)done);
        SR_TRY(peak(e, a, m, s));
        carry = m;
        if (n_gated) {
          // The gate (:119-121): a frame whose |pt.x| > res/2 returns (1, 0) and does NOT become prev. The pass above
          // correlated every frame with its immediate predecessor; behind a gated frame that is the wrong partner, so
          // walk the results in order and redo the (rare) pairs whose reference partner is an older frame.
          SR_TRY(hipMemcpyAsync(e->h_seq, d_out + 4 * (size_t)done, (size_t)m * 4 * sizeof(double), hipMemcpyDeviceToHost, s));
          SR_TRY(hipStreamSynchronize(s));
          int prev_slot = 0;
          for (int i = 1; i <= m; ++i) {
            double ptx = e->h_seq[4 * (size_t)(i - 1) + 2];
            if (prev_slot != i - 1) {
              double* o = d_out + 4 * (size_t)(done + i - 1);
              SR_TRY(seq_one_pair(e, zh + (size_t)prev_slot * zhf, zh + (size_t)i * zhf, o, s));
              SR_TRY(hipMemcpyAsync(e->h_out, o, 4 * sizeof(double), hipMemcpyDeviceToHost, s));
              SR_TRY(hipStreamSynchronize(s));
              ptx = e->h_out[2];
            }
            if (std::fabs(ptx) > (double)(res / 2)) ++gated_total;
            else prev_slot = i;
          }
          carry = prev_slot;
        }
        done += m;
      }
    }
    // prevIm_F32 <- the last frame that passed the gate (:128)
    if (carry >= 0) SR_TRY(hipMemcpyAsync(e->d_zh_prev, zh + (size_t)carry * zhf, zh_bytes, hipMemcpyDeviceToDevice, s));
    return MOF_OK;
  }();
  if (rc != MOF_OK) {
    (void)scratch_release(e, s);  // (the error text of the failing launch stays the thread's last error)
    return rc;
  }
  e->first = first;
  SR_TRY(scratch_release(e, s));
  if (n_gated) {
    SR_TRY(hipStreamSynchronize(s));
    *n_gated = gated_total;
  }
  return MOF_OK;
}

int mof_sr_process_sequence_host(mof_sr_engine* e, const uint8_t* frames, size_t frame_stride, size_t pitch, int n_frames,
                                 double* out, int* n_gated) try {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_gated) *n_gated = 0;
  if (n_frames == 0) return MOF_OK;
  const int res = e->cfg.resolution;
  if (!frames || !out || n_frames < 0 || pitch < (size_t)res) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad video arguments");
  SR_TRY(hipSetDevice(e->cfg.device));
  mof::RelaxedCapture relaxed;
  const size_t bpp = 4 * sizeof(double);
  {
    std::lock_guard<std::mutex> lock(e->host_mu);
    if (!e->host_pipe) e->host_pipe = new mof::HostPipe((size_t)res * res, &bpp, 1);
  }
  const mof::HostPipe::Out o{out, bpp};
  hipError_t he = hipSuccess;
  int gated_total = 0;
  // the frames go up in chunks on the pipe's copy stream, each chunk runs through the DEVICE video entry -- stateful, so chunk k + 1
  // continues where chunk k stopped, and with the gate resolved (n_gated != NULL there), i.e. exactly the frame-by-frame calls
  const int rc = e->host_pipe->process_frames(
      frames, frame_stride, pitch, res, res, n_frames, &o, e->stream,
      [e, &gated_total](const mof::HostPipe::Chunk& c, hipStream_t s) {
        int g = 0;
        const int r = mof_sr_process_sequence_device(e, c.d_cur, c.stride, (size_t)e->cfg.resolution, c.count, static_cast<double*>(c.d_out[0]), s, &g);
        gated_total += g;
        return r;
      },
      &he);
  if (rc == -1) return mof::capi_fail(MOF_ERR_HIP, "host video pipeline: %s", hipGetErrorString(he));
  if (n_gated) *n_gated = gated_total;
  return rc;
} catch (const std::bad_alloc&) {
  return mof::capi_fail(MOF_ERR_NO_MEMORY, "mof_sr_process_sequence_host: out of host memory");
}

int mof_sr_process_batch_device(mof_sr_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                size_t prev_stride, size_t pitch, int n_pairs, double* d_out, void* stream) {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  const int res = e->cfg.resolution;
  if (n_pairs == 0) return MOF_OK;  // an empty batch carries no pointers to check
  if (!d_cur || !d_prev || !d_out || n_pairs < 0 || pitch < (size_t)res) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  BusyGuard g(e->busy);
  if (!g.owned) return mof::capi_fail(MOF_ERR_BUSY, "engine busy");
  SR_TRY(hipSetDevice(e->cfg.device));
  hipStream_t s = (hipStream_t)stream;
  const size_t nn = (size_t)res * res;
  {
    const int rc = scratch_reserve(e, scratch_want(e, n_pairs), s);
    if (rc != MOF_OK) return rc;
  }
  SR_TRY(scratch_acquire(e, s));
  if (mof::stream_capturing(s)) e->graph_pinned.store(true);
  const int kChunk = e->chunk;
  // Under graph capture the fork / join below pulls the engine's stream into the caller's capture (event record on the
  // capturing stream, wait on the other), so a captured batch replays with the same two lanes.
  const bool two_lanes = e->two_lanes && n_pairs > kChunk;
  hipStream_t sr = two_lanes ? e->remap_stream : s;
  if (two_lanes) {  // fork: the remap lane starts behind whatever the caller's stream holds (also under graph capture)
    SR_TRY(hipEventRecord(e->ev_fork, s));
    SR_TRY(hipStreamWaitEvent(sr, e->ev_fork, 0));
  }
  int chunk_no = 0;
  for (int k0 = 0; k0 < n_pairs; k0 += kChunk, ++chunk_no) {
    const int n = (n_pairs - k0 < kChunk) ? n_pairs - k0 : kChunk;
    const int b = two_lanes ? (chunk_no & 1) : 0;
    uint8_t* lp_buf = e->d_lp + (size_t)b * kChunk * 2 * nn;
    // every pair is the two-call sequence of a fresh estimator: prev -> INTER_CUBIC (:45), cur -> INTER_LANCZOS4
    // (:112) onto the same zero-initialised tempIm; the transparent pixels are the same for both maps, so both
    // remaps simply write zeros there (zero_invalid) and the scratch needs no clearing pass
    if (two_lanes && chunk_no >= 2) SR_TRY(hipStreamWaitEvent(sr, e->ev_fft[b], 0));  // buffer b is free again
    mof::SrLpArgs lp{};
    lp.zero_invalid = 1;
    lp.pitch = pitch;
    lp.map = e->d_map;
    lp.res = res;
    lp.dst_stride = 2 * nn;
    lp.src = d_prev + (size_t)k0 * prev_stride;
    lp.src_stride = prev_stride;
    lp.dst = lp_buf + nn;
    lp_tables(e, 2, &lp);
    SR_TRY(mof::launch_sr_logpolar(lp, 2, n, sr));
    lp.src = d_cur + (size_t)k0 * cur_stride;
    lp.src_stride = cur_stride;
    lp.dst = lp_buf;
    lp_tables(e, 4, &lp);
    SR_TRY(mof::launch_sr_logpolar(lp, 4, n, sr));
    if (two_lanes) {
      SR_TRY(hipEventRecord(e->ev_lp[b], sr));
      SR_TRY(hipStreamWaitEvent(s, e->ev_lp[b], 0));  // also the join: every remap precedes a wait on the caller's stream
    }
    mof::SrPcArgs a = pc_args(e, lp_buf, lp_buf + nn, 2 * nn, d_out + 4 * (size_t)k0);
    // Independent pairs go through the FRAME kernels too (sr_seq_kernel.hip): each of the 2n log-polar images gets its own real
    // row transform (image 2p = cur of pair p, 2p + 1 = prev), K6s correlates slot 2p against 2p + 1, one pair per wave-run.
    // Same-box c5: 360 k pairs/s against 354 k for the packed pair kernels (K5 / K6 of sr_kernel.hip, MOF_SR_PAIR_SEQ=0), and
    // the batch entry now computes exactly what the stateful entry computes for a fresh estimator fed (prev, cur): same bits.
    static const bool via_frames = [] { const char* v = getenv("MOF_SR_PAIR_SEQ"); return !v || atoi(v) != 0; }();
    if (via_frames || e->generic || !mof::sr_pair_kernels_supported(res)) {  // (the packed pair kernels exist for 240 / 256 / 480 only)
      const size_t zhf = zh_floats(e);
      if (use_fused(e)) {
        SR_TRY(cols_fused(e, lp_buf + nn, lp_buf, 2 * nn, n, 1, s));
      } else {
        SR_TRY(rows_real(e, lp_buf, nn, e->d_Zt, zhf, 2 * n, s));
        SR_TRY(cols_seq(e, e->d_Zt + zhf, e->d_Zt, 2 * zhf, n, 1, s));
      }
      SR_TRY(peak(e, a, n, s));
    } else {
      a.degen = e->d_degen;
      SR_TRY(mof::launch_sr_phase_correlate(a, res, n, s));
    }
    if (two_lanes) SR_TRY(hipEventRecord(e->ev_fft[b], s));
  }
  SR_TRY(scratch_release(e, s));
  return MOF_OK;
}


int mof_sr_logpolar_batch_device(mof_sr_engine* e, const uint8_t* d_src, size_t src_stride, size_t pitch, int n_images,
                                 int interpolation, uint8_t* d_dst, void* stream) {
  if (!e) return mof::capi_fail(MOF_ERR_NOT_INIT, "null engine");
  const int res = e->cfg.resolution;
  if (n_images == 0) return MOF_OK;
  if (!d_src || !d_dst || n_images < 0 || pitch < (size_t)res) return mof::capi_fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  if (interpolation != MOF_INTER_CUBIC && interpolation != MOF_INTER_LANCZOS4)
    return mof::capi_fail(MOF_ERR_BAD_ARG, "interpolation must be MOF_INTER_CUBIC (2) or MOF_INTER_LANCZOS4 (4)");
  BusyGuard g(e->busy);
  if (!g.owned) return mof::capi_fail(MOF_ERR_BUSY, "engine busy");
  SR_TRY(hipSetDevice(e->cfg.device));
  mof::SrLpArgs lp{};
  lp.src = d_src;
  lp.src_stride = src_stride;
  lp.pitch = pitch;
  lp.dst = d_dst;
  lp.dst_stride = (size_t)res * res;
  lp.map = e->d_map;
  lp.res = res;
  lp_tables(e, interpolation, &lp);
  if (mof::stream_capturing((hipStream_t)stream)) e->graph_pinned.store(true);  // the map and the tables are the engine's
  SR_TRY(mof::launch_sr_logpolar(lp, interpolation, n_images, (hipStream_t)stream));  // touches no engine scratch
  return MOF_OK;
}

}  // extern "C"
