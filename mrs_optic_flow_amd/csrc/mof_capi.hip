// mof_capi.hip -- implementation of the C ABI declared in include/mof.h.
//
// Host-side engine objects (device buffers, one HIP stream each, the stateful previous
// frame of the reference's processors) around the gfx950 kernels. There is no CPU compute
// path in this library: without a HIP device every create() fails with MOF_ERR_NO_DEVICE.

#include "mof.h"

#include <hip/hip_runtime.h>

#include <atomic>
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
#include "mof_kernels.h"

namespace {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

}  // namespace

namespace mof {
// shared with mof_sr.hip: records the calling thread's last error text, returns `code`
int capi_fail(int code, const char* fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
  return code;
}

namespace {
struct Parked {
  void (*destroy_now)(void*);
  void* engine;
};
std::mutex g_parked_mutex;
std::vector<Parked> g_parked;
}  // namespace

void park_engine(void (*destroy_now)(void*), void* engine) {
  std::lock_guard<std::mutex> lock(g_parked_mutex);
  g_parked.push_back(Parked{destroy_now, engine});
}

int purge_parked() {
  std::vector<Parked> todo;
  {
    std::lock_guard<std::mutex> lock(g_parked_mutex);
    todo.swap(g_parked);
  }
  for (const Parked& p : todo) p.destroy_now(p.engine);
  return (int)todo.size();
}

int parked_count() {
  std::lock_guard<std::mutex> lock(g_parked_mutex);
  return (int)g_parked.size();
}
}  // namespace mof

namespace {

#define HIP_TRY(expr)                                                                          \
  do {                                                                                         \
    hipError_t _e = (expr);                                                                    \
    if (_e != hipSuccess) return fail(MOF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));    \
  } while (0)

// Non-re-entrancy flag of the reference (`running`, FftMethod.cpp:1775-1777), made atomic.
struct BusyGuard {
  std::atomic<bool>& flag;
  bool owned;
  explicit BusyGuard(std::atomic<bool>& f) : flag(f), owned(!f.exchange(true)) {}
  ~BusyGuard() {
    if (owned) flag.store(false);
  }
};

int select_device(int device) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
    (void)hipGetLastError();
    return fail(MOF_ERR_NO_DEVICE, "no HIP device available (this library has no CPU fallback)");
  }
  if (device < 0 || device >= n) return fail(MOF_ERR_BAD_ARG, "device %d out of range (0..%d)", device, n - 1);
  HIP_TRY(hipSetDevice(device));
  return MOF_OK;
}

}  // namespace

struct mof_fft_engine {
  mof_fft_config cfg{};
  hipStream_t stream = nullptr;
  float* d_twiddles = nullptr;
  uint8_t* d_frames[2] = {nullptr, nullptr};  // [cur_slot], [1-cur_slot] = previous
  int prev_slot = 0;
  size_t frame_bytes = 0;
  double* d_out = nullptr;       // one frame's results
  double* h_out = nullptr;       // pinned
  uint8_t* h_stage = nullptr;    // pinned upload staging (tightly packed frame)
  bool first = true;             // FftMethod.cpp:1761
  bool generic = false;          // patch sizes without a hand-tuned instantiation run the planned kernel (pc_kernel_generic.hip)
  bool large = false;            // ... or, when the padded patch does not fit a CU's LDS, the planned pipeline through HBM
  mof::PcPlan plan{};            //     scratch (pc_large_kernel.hip)
  float* d_pair_slabs = nullptr; // MOF_FFT_PAIR_HALF=1 (N = 128): slabs of the pair kernel on the half tile (pc_seq_half.hip), two per CU
  int n_pair_slabs = 0;
  int half_m = 0;                // > 0: cv::phaseCorrelate-model batches on full-resolution frames run the fused half-tile kernel of that
                                 //      transform size instead (pc_half_kernel.hip: even padded sizes in (135, 192]; MOF_FFT_HALF=1: tuned sizes too)
  int seq_half_m = 0;            // > 0: a planned size whose VIDEO entry runs the half-tile kernel's sequence form (its pair entries stay on the planned kernel)
  // scratch of the large-patch pipeline, for `cap` patch pairs per pass: row half-spectra of 2 cap patches, Dt, peak
  // candidates, constant-patch flags, C_dc. Grown by a batch that needs more (never under a graph capture, never while pinned).
  float *d_zh = nullptr, *d_dt = nullptr, *d_cdc = nullptr;
  float2* d_cand = nullptr;
  int* d_flags = nullptr;
  int cap = 0;
  hipEvent_t scratch_ev = nullptr;       // behind the last kernel that used the scratch (cross-stream ordering, as mof_sr)
  hipStream_t scratch_stream = nullptr;
  bool scratch_used = false;
  std::atomic<bool> busy{false};
  std::atomic<bool> graph_pinned{false};  // a batch call was captured into a HIP graph (capi_graph.hpp)
  std::mutex host_mu;                     // mof_fft_process_batch_host: upload / run / download pipeline (host_pipe.hpp), made by its first call
  mof::HostPipe* host_pipe = nullptr;
};

static void large_free(mof_fft_engine* e) {
  void* bufs[] = {e->d_zh, e->d_dt, e->d_cdc, e->d_cand, e->d_flags};
  for (void* b : bufs)
    if (b) (void)hipFree(b);
  e->d_zh = e->d_dt = e->d_cdc = nullptr;
  e->d_cand = nullptr;
  e->d_flags = nullptr;
  e->cap = 0;
}

static hipError_t large_alloc(mof_fft_engine* e, int cap) {
  mof::RelaxedCapture relaxed;
  large_free(e);
  const size_t zhf = mof::pcl_zh_floats(e->plan);
  hipError_t err;
  if ((err = hipMalloc(&e->d_zh, (size_t)2 * cap * zhf * sizeof(float))) != hipSuccess) return err;
  if ((err = hipMalloc(&e->d_dt, (size_t)cap * zhf * sizeof(float))) != hipSuccess) return err;
  if ((err = hipMalloc(&e->d_cand, (size_t)cap * mof::pcl_candidates(e->plan) * sizeof(float2))) != hipSuccess) return err;
  // [0, 2 cap): flags per pair (cur | prev); [2 cap, 4 cap): flags per image of a video pass; [4 cap, 12 cap): four exact pixel sums per image
  // (the tuned transform sizes whose Nyquist bin is not exact: 250, 400, 432)
  if ((err = hipMalloc(&e->d_flags, (size_t)12 * cap * sizeof(int))) != hipSuccess) return err;
  if ((err = hipMalloc(&e->d_cdc, (size_t)cap * sizeof(float))) != hipSuccess) return err;
  e->cap = cap;
  return hipSuccess;
}

// Frame pairs per pass of the large-patch pipeline: as many as keep the scratch (three Zh-sized planes per patch pair) under
// ~1.5 GB, at least one (MOF_FFT_LARGE_PASS overrides, sweeps)
static int large_pass_pairs(const mof_fft_engine* e, int patches) {
  static const int forced = [] { const char* v = getenv("MOF_FFT_LARGE_PASS"); return v ? atoi(v) : 0; }();
  if (forced > 0) return forced;
  const size_t per_pair = (size_t)3 * patches * mof::pcl_zh_floats(e->plan) * sizeof(float);
  const size_t n = ((size_t)3 << 29) / (per_pair ? per_pair : 1);
  return n < 1 ? 1 : (n > 4096 ? 4096 : (int)n);
}

// The large-patch pipeline on n_pairs frame pairs (a.grid_* patches each; a.downscale / a.channels as K1): passes of whole frame
// pairs through the engine's scratch. Returns a MOF status.
static int launch_large(mof_fft_engine* e, const mof::PcArgs& a, int n_pairs, hipStream_t s) {
  const int patches = a.grid_x * a.grid_y;
  const int pass_max = large_pass_pairs(e, patches);
  const int want_pairs = n_pairs < pass_max ? n_pairs : pass_max;
  const bool capturing = mof::stream_capturing(s);
  if ((long)want_pairs * patches > e->cap) {
    if (e->graph_pinned.load())
      return fail(MOF_ERR_BUSY, "the large-patch scratch would have to grow, but a captured HIP graph still points into it: run the "
                                "largest batch once before capturing, or call mof_fft_release_graphs once the graphs are gone");
    if (capturing)
      return fail(MOF_ERR_BAD_ARG, "the large-patch scratch must grow to %d patch pairs, which cannot happen inside a graph capture: "
                                   "run one batch of this size before capturing", want_pairs * patches);
    if (e->scratch_used) (void)hipEventSynchronize(e->scratch_ev);
    (void)hipStreamSynchronize(e->stream);
    const hipError_t err = large_alloc(e, want_pairs * patches);
    if (err != hipSuccess) {
      (void)large_alloc(e, e->cfg.grid_x * e->cfg.grid_y);  // keep the stateful entry usable
      return fail(err == hipErrorOutOfMemory ? MOF_ERR_NO_MEMORY : MOF_ERR_HIP, "large-patch scratch for %d patch pairs: %s",
                  want_pairs * patches, hipGetErrorString(err));
    }
  }
  if (e->scratch_used && e->scratch_stream != s && !capturing) HIP_TRY(hipStreamWaitEvent(s, e->scratch_ev, 0));
  const size_t zhf = mof::pcl_zh_floats(e->plan);
  // Unpadded patches of 240 / 256 / 480 pixels (the reference's whole-frame fallback among them, FftMethod.cpp:1709-1716) are the
  // scale / rotation estimator's transform sizes: its tuned K5s / K6s / K7 (sr_seq_kernel.hip, sr_kernel.hip) run instead of the planned
  // L5 / L6 / L7 -- same Zh / Dt / candidate formats, 2.5 x faster; L8 (the FftMethod tail) stays. Gray and BGR8 frames alike (the
  // latter promise the gray path's bits); the long-range mode keeps the planned kernels. MOF_FFT_LARGE_TUNED=0: planned kernels (A/B).
  static const bool tuned_on = [] { const char* v = getenv("MOF_FFT_LARGE_TUNED"); return !v || atoi(v) != 0; }();
  // r06: 200, 216, 270, 288, 300, 320, 360, 384, 450 too, and patches that PAD to one of these sizes (193 .. 216, 226 .. 240, 251 .. 256, 271 .. 288, 301 .. 320, 325 .. 360, 376 .. 384,
  // 451 .. 480): the row kernel
  // zero-pads, the column kernel applies the box-zero rule of padded constant patches from the row kernel's flags
  static const int tuned_sizes[] = {225, 243, 375, 405, 625, 675, 729,  // (r06: the odd sizes too)
                                    200, 216, 240, 250, 256, 270, 288, 300, 320, 324, 360, 384, 400, 432, 450, 480, 486, 500, 512,
                                    540, 576, 600, 640, 648, 720, 750, 768, 800, 810, 864, 900, 960};  // (r06: 324, 486, 500 and every even size above 512 -- first radix up to 32)
  bool tuned = false;
  for (int t : tuned_sizes) tuned = tuned || e->plan.m == t;
  tuned = tuned && tuned_on && a.downscale == 1;
  // (250 = 10 x 25, 400 = 16 x 25, 432 = 16 x 27: no plan of theirs ends in an even radix, so the Nyquist bins of their transforms are not exact
  //  -- the row kernel accumulates each image's four exact integer sums and the column kernel takes the real-only slots from those)
  const bool odd_tail = e->plan.m == 250 || e->plan.m == 400 || e->plan.m == 432;
  // r06, a VIDEO on the tuned transforms (pair k = (frame k + 1, frame k): mof_fft_process_sequence_device, or any caller whose cur = prev + one frame):
  // every frame's row spectra are formed ONCE per pass -- Zh slot = frame * patches + patch, so pair q = k * patches + patch finds its previous
  // image at slot q and its current one at slot q + patches, which is exactly what the column kernel's (zh_prev, zh_cur, stride) takes; the
  // kernels and their arithmetic are the pair form's, so are the bits. MOF_FFT_LARGE_VIDEO=0 keeps the pair form (A/B and its tests).
  static const bool video_on = [] { const char* v = getenv("MOF_FFT_LARGE_VIDEO"); return !v || atoi(v) != 0; }();
  const bool video = video_on && tuned && n_pairs >= 2 && a.cur == a.prev + a.prev_stride && a.cur_stride == a.prev_stride;
  const int per_pass = e->cap / patches;
  for (int k0 = 0; k0 < n_pairs; k0 += per_pass) {
    const int np = n_pairs - k0 < per_pass ? n_pairs - k0 : per_pass, nq = np * patches;
    if (video) {
      mof::PclSrc src{};
      src.base[0] = a.prev + (size_t)k0 * a.prev_stride;  // frame k0 of the video
      src.stride[0] = a.prev_stride;
      src.pitch = a.pitch;
      src.paired = 2;
      src.grid_x = a.grid_x;
      src.grid_y = a.grid_y;
      src.origin_x = a.origin_x;
      src.origin_y = a.origin_y;
      src.stride_x = a.stride_x;
      src.stride_y = a.stride_y;
      int* fs = e->d_flags + (size_t)2 * e->cap;  // per image of this pass: (np + 1) * patches <= 2 cap
      int* sums = e->d_flags + (size_t)4 * e->cap;  // four ints per image slot
      HIP_TRY(hipMemsetAsync(fs, 0, (size_t)(nq + patches) * sizeof(int), s));
      if (odd_tail) HIP_TRY(hipMemsetAsync(sums, 0, (size_t)4 * (nq + patches) * sizeof(int), s));
      const int frames_per_launch = 65534 / patches > 0 ? 65534 / patches : 1;
      for (int j0 = 0; j0 < np + 1; j0 += frames_per_launch) {
        const int nj = np + 1 - j0 < frames_per_launch ? np + 1 - j0 : frames_per_launch;
        mof::PclSrc sj = src;
        sj.base[0] += (size_t)j0 * a.prev_stride;
        HIP_TRY(mof::launch_sr_rows_real_src(sj, e->d_twiddles, e->d_zh + (size_t)j0 * patches * zhf, zhf, fs + (size_t)j0 * patches, e->plan.m,
                                             nj * patches, a.channels, e->plan.n, s, sums + (size_t)4 * j0 * patches));
      }
      HIP_TRY(mof::launch_pcl_seq_flags(fs, e->d_flags, patches, nq, s));
      const float* zp = e->d_zh;
      const float* zc = e->d_zh + (size_t)patches * zhf;
      HIP_TRY(mof::launch_sr_cols_seq(zp, zc, zhf, e->d_twiddles, e->d_dt, e->plan.m, nq, 1, s, e->d_flags, e->plan.n, sums, sums + (size_t)4 * patches, 4));
      HIP_TRY(mof::launch_pcl_cdc(zp, zc, zhf, e->plan.m, e->d_cdc, nq, s));
      HIP_TRY(mof::launch_sr_rows_inv(e->d_dt, e->d_twiddles, e->d_cand, e->plan.m, nq, s));
      mof::PclFinal f{};
      f.Dt = e->d_dt;
      f.cand = e->d_cand;
      f.twiddles = e->d_twiddles;
      f.mode = 1;
      f.max_px_speed_sq = a.max_px_speed_sq;
      f.out = a.out + (size_t)k0 * patches * 2;
      f.flags = e->d_flags;
      f.cdc = e->d_cdc;
      HIP_TRY(mof::launch_pcl_peak(f, e->plan, nq, s, true));
      continue;
    }
    mof::PclSrc src{};
    src.base[0] = a.cur + (size_t)k0 * a.cur_stride;
    src.base[1] = a.prev + (size_t)k0 * a.prev_stride;
    src.stride[0] = a.cur_stride;
    src.stride[1] = a.prev_stride;
    src.pitch = a.pitch;
    src.paired = 1;
    src.grid_x = a.grid_x;
    src.grid_y = a.grid_y;
    src.origin_x = a.origin_x;
    src.origin_y = a.origin_y;
    src.stride_x = a.stride_x;
    src.stride_y = a.stride_y;
    HIP_TRY(hipMemsetAsync(e->d_flags, 0, (size_t)2 * nq * sizeof(int), s));
    int* sums = e->d_flags + (size_t)4 * e->cap;  // four ints per image 2 q + which
    if (tuned && odd_tail) HIP_TRY(hipMemsetAsync(sums, 0, (size_t)8 * nq * sizeof(int), s));
    // (launch_pcl_rows splits at 65534 images on pair boundaries: keep a pass's image count a multiple of 2 * patches below that)
    const int pairs_per_launch = 65534 / (2 * patches) > 0 ? 65534 / (2 * patches) : 1;
    for (int j0 = 0; j0 < np; j0 += pairs_per_launch) {
      const int nj = np - j0 < pairs_per_launch ? np - j0 : pairs_per_launch;
      mof::PclSrc sj = src;
      sj.base[0] += (size_t)j0 * a.cur_stride;
      sj.base[1] += (size_t)j0 * a.prev_stride;
      if (tuned)
        HIP_TRY(mof::launch_sr_rows_real_src(sj, e->d_twiddles, e->d_zh + (size_t)2 * j0 * patches * zhf, zhf,
                                             e->d_flags + (size_t)2 * j0 * patches, e->plan.m, 2 * nj * patches, a.channels, e->plan.n, s,
                                             sums + (size_t)8 * j0 * patches));
      else
        HIP_TRY(mof::launch_pcl_rows(sj, e->plan, e->d_twiddles, e->d_zh + (size_t)2 * j0 * patches * zhf, zhf,
                                     e->d_flags + (size_t)2 * j0 * patches, 2 * nj * patches, a.channels, a.downscale, s));
    }
    if (tuned) {
      HIP_TRY(mof::launch_sr_cols_seq(e->d_zh + zhf, e->d_zh, 2 * zhf, e->d_twiddles, e->d_dt, e->plan.m, nq, 1, s, e->d_flags, e->plan.n, sums + 4, sums, 8));
      HIP_TRY(mof::launch_pcl_cdc(e->d_zh + zhf, e->d_zh, 2 * zhf, e->plan.m, e->d_cdc, nq, s));
      HIP_TRY(mof::launch_sr_rows_inv(e->d_dt, e->d_twiddles, e->d_cand, e->plan.m, nq, s));
    } else {
      HIP_TRY(mof::launch_pcl_cols(e->d_zh + zhf, e->d_zh, 2 * zhf, e->plan, e->d_twiddles, e->d_dt, e->d_cdc, e->d_flags, nq, s));
    }
    mof::PclFinal f{};
    f.Dt = e->d_dt;
    f.cand = e->d_cand;
    f.twiddles = e->d_twiddles;
    f.mode = 1;
    f.max_px_speed_sq = a.max_px_speed_sq;
    f.out = a.out + (size_t)k0 * patches * 2;
    f.flags = e->d_flags;
    f.cdc = e->d_cdc;
    HIP_TRY(mof::launch_pcl_peak(f, e->plan, nq, s, tuned));
  }
  if (!capturing) {
    HIP_TRY(hipEventRecord(e->scratch_ev, s));
    e->scratch_stream = s;
    e->scratch_used = true;
  }
  return MOF_OK;
}

// every K1 launch of an engine: the hand-tuned instantiation of its patch size, the planned general kernel, or -- for patches
// too large for a CU -- the planned pipeline through HBM scratch. Returns a MOF status.
static int launch_field(mof_fft_engine* e, const mof::PcArgs& a, int n_pairs, hipStream_t stream) {
  if (e->half_m > 0 && a.downscale == 1 && a.peak_model == 0) {  // (no scratch, nothing engine-owned but the twiddles)
    HIP_TRY(mof::launch_pc_half(a, e->half_m, e->cfg.patch_size, n_pairs, stream));
    return MOF_OK;
  }
  if (e->large) return launch_large(e, a, n_pairs, stream);
  if (e->d_pair_slabs && a.downscale == 1) {
    HIP_TRY(mof::launch_pc_pair_half(a, e->cfg.patch_size, n_pairs, e->d_pair_slabs, e->n_pair_slabs, stream));
    return MOF_OK;
  }
  HIP_TRY(e->generic ? mof::launch_pc_generic(a, e->plan, n_pairs, stream) : mof::launch_pc_field(a, e->cfg.patch_size, n_pairs, stream));
  return MOF_OK;
}
#define FIELD_TRY(expr)       \
  do {                        \
    const int _rc = (expr);   \
    if (_rc != MOF_OK) return _rc; \
  } while (0)

struct mof_bm_engine {
  mof_bm_config cfg{};
  hipStream_t stream = nullptr;
  uint8_t* d_frames[2] = {nullptr, nullptr};
  int prev_slot = 0;
  size_t frame_bytes = 0;
  int8_t* d_dx = nullptr;
  int8_t* d_dy = nullptr;
  int8_t* d_mode = nullptr;
  int8_t* h_res = nullptr;     // pinned: dx | dy | mode
  uint8_t* h_stage = nullptr;
  // BlockMethod::Refine scratch (allocated on first use): the two 2x images, nine SADs
  uint8_t* d_up[2] = {nullptr, nullptr};
  unsigned long long* d_sad9 = nullptr;
  unsigned long long* h_sad9 = nullptr;
  bool have_pair = false;      // a processImage call has been made (both frame slots are meaningful)
  std::atomic<bool> busy{false};
  std::atomic<bool> graph_pinned{false};
  std::mutex host_mu;          // mof_bm_process_batch_host's pipeline (host_pipe.hpp)
  mof::HostPipe* host_pipe = nullptr;
};

extern "C" {

const char* mof_version(void) { return "mof-hip 0.1.0 (gfx950)"; }
const char* mof_last_error(void) { return g_err; }

int mof_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) {
    (void)hipGetLastError();
    return 0;
  }
  return n < 0 ? 0 : n;
}

/* ------------------------------------------------------------------------------------------ */
/* FFT                                                                                        */
/* ------------------------------------------------------------------------------------------ */

int mof_fft_config_reference(mof_fft_config* cfg, int frame_size, int sample_point_size, double max_px_speed) {
  if (!cfg || frame_size < 2 || sample_point_size < 1) return fail(MOF_ERR_BAD_ARG, "bad reference geometry");
  std::memset(cfg, 0, sizeof(*cfg));
  if (frame_size % 2 == 1) frame_size--;                                           // FftMethod.cpp:1706-1708
  if (frame_size % sample_point_size != 0) sample_point_size = frame_size;         // :1709-1716
  const int sq = frame_size / sample_point_size;                                   // :1719
  cfg->frame_width = cfg->frame_height = frame_size;
  cfg->patch_size = sample_point_size;
  cfg->grid_x = cfg->grid_y = sq;
  cfg->origin_x = cfg->origin_y = 0;
  cfg->stride_x = cfg->stride_y = sample_point_size;
  cfg->max_px_speed = max_px_speed;
  cfg->device = 0;
  cfg->peak_model = MOF_PEAK_OPENCV;  // useOCL = false, the live path
  cfg->search_radius = 55;            // SEARCH_RADIUS, FftMethod.cpp:820 (only read by MOF_PEAK_OCL)
  return MOF_OK;
}

static int validate_fft(const mof_fft_config* c) {
  if (!c) return fail(MOF_ERR_BAD_ARG, "null config");
  if (c->frame_width < 1 || c->frame_height < 1 || c->grid_x < 1 || c->grid_y < 1 || c->origin_x < 0 ||
      c->origin_y < 0 || c->stride_x < 0 || c->stride_y < 0)
    return fail(MOF_ERR_BAD_ARG, "bad FFT geometry");
  if (c->patch_size < 2) return fail(MOF_ERR_BAD_ARG, "patch_size %d: a patch needs at least 2 x 2 pixels", c->patch_size);
  if (!mof::pc_patch_size_supported(c->patch_size)) {
    // any other samplePointSize (FftMethod.cpp:1680-1720 takes it from a ROS parameter): the planned kernel on the size
    // cv::phaseCorrelate pads to, M = getOptimalDFTSize(N)
    mof::PcPlan plan;
    if (!mof::pc_build_plan(c->patch_size, &plan)) {
      // the padded patch does not fit one CU's LDS (M > 135): the planned pipeline through HBM scratch (pc_large_kernel.hip)
      if (!mof::pc_build_line_plan(c->patch_size, &plan))
        return fail(MOF_ERR_UNSUPPORTED, "patch_size %d pads to %d: beyond the planned transforms (<= 960)", c->patch_size,
                    mof::pc_optimal_dft_size(c->patch_size));
      if (c->peak_model == MOF_PEAK_OCL)
        return fail(MOF_ERR_UNSUPPORTED, "peak_model MOF_PEAK_OCL is available for patches that fit one CU (padded size <= 135); "
                    "patch_size %d runs the cv::phaseCorrelate model only", c->patch_size);
    }
    // useOCL=true plans radix-{2,3,4,5,8} passes for the patch size itself and never pads (FftMethod.cpp:481-539, :787-816):
    // sizes with another prime factor have no OpenCL plan in the reference either; its CCS packing assumes an even size
    if (c->peak_model == MOF_PEAK_OCL && (plan.m != plan.n || (plan.n & 1)))
      return fail(MOF_ERR_UNSUPPORTED, "peak_model MOF_PEAK_OCL needs an even patch_size of the form 2^a 3^b 5^c (the reference's "
                  "OpenCL branch cannot plan %d either)", c->patch_size);
  }
  if (c->origin_x + (long)(c->grid_x - 1) * c->stride_x + c->patch_size > c->frame_width ||
      c->origin_y + (long)(c->grid_y - 1) * c->stride_y + c->patch_size > c->frame_height)
    return fail(MOF_ERR_BAD_ARG, "patch grid leaves the frame");
  if ((long)c->grid_x * c->grid_y > (1 << 20)) return fail(MOF_ERR_BAD_ARG, "too many patches");
  if (!(c->max_px_speed >= 0.0)) return fail(MOF_ERR_BAD_ARG, "max_px_speed must be >= 0");
  if (c->peak_model != MOF_PEAK_OPENCV && c->peak_model != MOF_PEAK_OCL)
    return fail(MOF_ERR_BAD_ARG, "peak_model must be MOF_PEAK_OPENCV or MOF_PEAK_OCL");
  if (c->peak_model == MOF_PEAK_OCL && c->search_radius < 0) return fail(MOF_ERR_BAD_ARG, "search_radius must be >= 0");
  return MOF_OK;
}

int mof_fft_create(const mof_fft_config* cfg, mof_fft_engine** out) try {
  if (!out) return fail(MOF_ERR_BAD_ARG, "null out");
  *out = nullptr;
  int rc = validate_fft(cfg);
  if (rc) return rc;
  rc = select_device(cfg->device);
  if (rc) return rc;
  mof::RelaxedCapture relaxed;  // allocating an engine must not invalidate a capture on another thread
  const size_t res = (size_t)cfg->grid_x * cfg->grid_y * 2;
  mof_fft_engine* e = new (std::nothrow) mof_fft_engine();
  if (!e) return fail(MOF_ERR_NO_MEMORY, "out of host memory");
  e->cfg = *cfg;
  e->frame_bytes = (size_t)cfg->frame_width * cfg->frame_height;
  // diagnostics (A/B of the kernel families on one size, and their parity tests against each other):
  // MOF_FFT_FORCE_PLANNED=1 runs the planned LDS kernel also where a tuned instantiation exists, MOF_FFT_FORCE_LARGE=1 the
  // planned pipeline through HBM scratch at any size (cv::phaseCorrelate model only)
  static const bool force_planned = getenv("MOF_FFT_FORCE_PLANNED") != nullptr, force_large = getenv("MOF_FFT_FORCE_LARGE") != nullptr;
  e->generic = !mof::pc_patch_size_supported(cfg->patch_size) || force_planned || force_large;
  if (e->generic && (force_large && cfg->peak_model == MOF_PEAK_OPENCV ? true : !mof::pc_build_plan(cfg->patch_size, &e->plan))) {
    e->generic = false;
    e->large = true;
    if (!mof::pc_build_line_plan(cfg->patch_size, &e->plan)) {  // (validate_fft has checked it)
      delete e;
      return fail(MOF_ERR_UNSUPPORTED, "no plan for patch_size %d", cfg->patch_size);
    }
  }
  const int n = (e->generic || e->large) ? e->plan.m : cfg->patch_size;  // transform size: the planned kernels work on the padded patch
  {
    // the fused half-tile kernel: the default for large patches whose half tile fits a CU (even padded size <= 192); MOF_FFT_HALF=0
    // keeps them on the pipeline through HBM scratch, MOF_FFT_HALF=1 also routes the tuned / planned sizes it is instantiated for
    // (64, 96, 120, 128) through it -- A/B and the parity tests of the formulation
    static const int half_knob = [] { const char* v = getenv("MOF_FFT_HALF"); return v ? atoi(v) : -1; }();
    // r05: N = 120 -- the reference's default samplePointSize -- takes it by default too: two workgroups per CU and every phase on all
    // waves beat the tuned one-workgroup kernel there (1.14 M against 1.10 M pairs/s same-box, profiles/r05_half_raw_pairsrc_ab.txt);
    // its long-range mode and OpenCL peak model stay on the tuned kernel (launch_field)
    const bool tuned_size_default = n == 120 && !e->generic && !e->large;
    // ... and the planned sizes where it beats the full-tile planned kernel on the box (transform sizes 60, 96, 100: p60 1.10 -> 1.38 M,
    // p96 757 -> 855 k pairs/s, profiles/r05_half_vs_planned_bench_ab.txt, r05_half_vs_planned_rates.txt; on patches padded to 64 it
    // loses 4 %, p62); the tuned N = 64 / 128 pair kernels stay faster than it and keep their sizes. Patches of 109 .. 119 pixels pad to
    // 120 and follow N = 120 itself (the planned kernel at 120: 869 k against 1.30 M pairs/s)
    const bool planned_size_default = e->generic && !force_planned && (n == 60 || n == 72 || n == 90 || n == 96 || n == 100 || n == 120);  // (72, 90: +3 % on pairs, and the video form: profiles/r05_half_vs_planned_final.txt)
    if (cfg->peak_model == MOF_PEAK_OPENCV && mof::pc_half_supported(n) && half_knob != 0 && !force_large && !force_planned &&
        (e->large || half_knob == 1 || tuned_size_default || planned_size_default))
      e->half_m = n;
  }
  // twiddles W_n^k = exp(-2 pi i k / n), double -> float, axis values exact
  std::vector<float> tw(2 * (size_t)n);
  for (int k = 0; k < n; ++k) {
    double ang = -2.0 * 3.14159265358979323846 * (double)k / (double)n;
    double c = std::cos(ang), s = std::sin(ang);
    if ((4 * k) % n == 0) {
      const int q = (4 * k) / n;
      c = (q == 0) ? 1.0 : (q == 2) ? -1.0 : 0.0;
      s = (q == 1) ? -1.0 : (q == 3) ? 1.0 : 0.0;
    }
    tw[2 * k] = (float)c;
    tw[2 * k + 1] = (float)s;
  }
  if (!e->generic && !e->large && n == 64) {  // the tuned N = 64 kernel's matrix-core stage reads its DFT-16 fragments from behind the twiddles
    std::vector<uint32_t> frag(1024, 0u);
    mof::pc_mfma_s1_fragments(frag.data());
    tw.resize(128 + 1024);
    std::memcpy(tw.data() + 128, frag.data(), 4096);
  }
#define CREATE_TRY(expr)                                                                        \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      fail(MOF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));                                \
      mof_fft_destroy(e);                                                                       \
      return MOF_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)
  CREATE_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  CREATE_TRY(hipMalloc(&e->d_twiddles, tw.size() * sizeof(float)));
  CREATE_TRY(mof::copy_on(e->stream, e->d_twiddles, tw.data(), tw.size() * sizeof(float), hipMemcpyHostToDevice));
  CREATE_TRY(hipMalloc(&e->d_frames[0], e->frame_bytes));
  CREATE_TRY(hipMalloc(&e->d_frames[1], e->frame_bytes));
  CREATE_TRY(mof::fill_on(e->stream, e->d_frames[0], 0, e->frame_bytes));
  CREATE_TRY(mof::fill_on(e->stream, e->d_frames[1], 0, e->frame_bytes));
  CREATE_TRY(hipMalloc(&e->d_out, res * sizeof(double)));
  CREATE_TRY(hipHostMalloc(&e->h_out, res * sizeof(double), hipHostMallocDefault));
  CREATE_TRY(hipHostMalloc(&e->h_stage, e->frame_bytes, hipHostMallocDefault));
  if (e->half_m > 0) CREATE_TRY(mof::pc_configure_half(e->half_m));
  {
    // the planned sizes whose VIDEO form is the half-tile kernel's although their pair form is not (fft_sequence)
    static const int half_knob2 = [] { const char* v = getenv("MOF_FFT_HALF"); return v ? atoi(v) : -1; }();
    static const bool force_planned2 = getenv("MOF_FFT_FORCE_PLANNED") != nullptr, force_large2 = getenv("MOF_FFT_FORCE_LARGE") != nullptr;
    if (e->generic && e->half_m == 0 && half_knob2 != 0 && !force_planned2 && !force_large2 && cfg->peak_model == MOF_PEAK_OPENCV &&
        !mof::pc_half_supported(n) && mof::pc_half_sequence_supported(n)) {
      e->seq_half_m = n;
      CREATE_TRY(mof::pc_configure_half(n));
    }
  }
  if (e->large) {
    CREATE_TRY(hipEventCreateWithFlags(&e->scratch_ev, hipEventDisableTiming));
    CREATE_TRY(large_alloc(e, cfg->grid_x * cfg->grid_y));  // one frame pair; a batch grows it to a whole pass
  } else if (e->generic) {
    CREATE_TRY(mof::pc_configure_generic());
  } else {
    CREATE_TRY(mof::pc_configure(n));
    if (mof::pc_sequence_supported(n)) CREATE_TRY(mof::pc_configure_sequence());
    if (mof::pc_sequence_half_supported(n)) CREATE_TRY(mof::pc_configure_sequence_half(n));
    if (n == 128 && cfg->peak_model == MOF_PEAK_OPENCV && mof::pc_half_sequence_supported(n)) CREATE_TRY(mof::pc_configure_half(n));  // (its video form serves 128 x 128, fft_sequence)
    // A/B knob (r05): independent pairs of 128 x 128 patches through the pair kernel on the HALF tile, two workgroups per CU
    static const bool pair_half = [] { const char* v = getenv("MOF_FFT_PAIR_HALF"); return v && atoi(v) != 0; }();
    if (pair_half && mof::pc_pair_half_supported(n)) {
      int dev = 0, cus = 256;
      hipDeviceProp_t prop;
      if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cus = prop.multiProcessorCount;
      static const int per_cu = [] { const char* v = getenv("MOF_FFT_PAIR_HALF_WGS"); return v ? atoi(v) : 2; }();
      e->n_pair_slabs = per_cu * cus;
      CREATE_TRY(mof::pc_configure_pair_half(n));
      CREATE_TRY(hipMalloc(&e->d_pair_slabs, (size_t)e->n_pair_slabs * mof::pc_pair_half_slab_floats(n) * sizeof(float)));
    }
  }
#undef CREATE_TRY
  *out = e;
  return MOF_OK;
} catch (const std::bad_alloc&) {
  return fail(MOF_ERR_NO_MEMORY, "mof_fft_create: out of host memory");
}

const char* mof_fft_kernel_variant(const mof_fft_engine* e) {
  return !e ? "" : (e->half_m > 0 ? "planned-half" : (e->large ? "planned-large" : (e->generic ? "planned" : mof::pc_kernel_variant(e->cfg.patch_size))));
}

static void fft_destroy_now(void* p) {
  mof_fft_engine* e = static_cast<mof_fft_engine*>(p);
  mof::RelaxedCapture relaxed;  // frees must not invalidate a capture running on another thread
  (void)hipSetDevice(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  if (e->scratch_ev && e->scratch_used) (void)hipEventSynchronize(e->scratch_ev);  // a batch on a caller's stream may still use the scratch
  delete e->host_pipe;
  large_free(e);
  if (e->scratch_ev) (void)hipEventDestroy(e->scratch_ev);
  if (e->d_twiddles) (void)hipFree(e->d_twiddles);
  if (e->d_pair_slabs) (void)hipFree(e->d_pair_slabs);
  if (e->d_frames[0]) (void)hipFree(e->d_frames[0]);
  if (e->d_frames[1]) (void)hipFree(e->d_frames[1]);
  if (e->d_out) (void)hipFree(e->d_out);
  if (e->h_out) (void)hipHostFree(e->h_out);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

void mof_fft_destroy(mof_fft_engine* e) {
  if (!e) return;
  if (e->graph_pinned.load()) {  // a captured graph may still read the twiddles: keep them until the owner releases
    mof::park_engine(&fft_destroy_now, e);
    return;
  }
  fft_destroy_now(e);
}

int mof_fft_release_graphs(mof_fft_engine* e) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  e->graph_pinned.store(false);
  return MOF_OK;
}

int mof_fft_graph_pinned(const mof_fft_engine* e) { return e && e->graph_pinned.load() ? 1 : 0; }

int mof_purge_deferred(void) { return mof::purge_parked(); }
int mof_deferred_count(void) { return mof::parked_count(); }

static void pack_frame(uint8_t* dst, const uint8_t* src, size_t pitch, int w, int h) {
  for (int y = 0; y < h; ++y) std::memcpy(dst + (size_t)y * w, src + (size_t)y * pitch, (size_t)w);
}

static mof::PcArgs fft_args(const mof_fft_engine* e, const uint8_t* cur, size_t cs, const uint8_t* prev, size_t ps,
                            size_t pitch, double* out) {
  mof::PcArgs a{};
  a.cur = cur;
  a.prev = prev;
  a.cur_stride = cs;
  a.prev_stride = ps;
  a.pitch = pitch;
  a.grid_x = e->cfg.grid_x;
  a.grid_y = e->cfg.grid_y;
  a.origin_x = e->cfg.origin_x;
  a.origin_y = e->cfg.origin_y;
  a.stride_x = e->cfg.stride_x;
  a.stride_y = e->cfg.stride_y;
  a.downscale = 1;
  a.channels = 1;
  a.peak_model = e->cfg.peak_model;
  a.search_radius = e->cfg.search_radius;
  a.max_px_speed_sq = e->cfg.max_px_speed * e->cfg.max_px_speed;  // pow(max_px_speed_t, 2), FftMethod.cpp:1686
  a.twiddles = e->d_twiddles;
  a.out = out;
  return a;
}

int mof_fft_set_prev(mof_fft_engine* e, const uint8_t* frame, size_t pitch) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (!frame || pitch < (size_t)e->cfg.frame_width) return fail(MOF_ERR_BAD_ARG, "bad frame/pitch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  pack_frame(e->h_stage, frame, pitch, e->cfg.frame_width, e->cfg.frame_height);
  HIP_TRY(hipMemcpyAsync(e->d_frames[e->prev_slot], e->h_stage, e->frame_bytes, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MOF_OK;
}

int mof_fft_reset(mof_fft_engine* e) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  e->first = true;
  return MOF_OK;
}

int mof_fft_process(mof_fft_engine* e, const uint8_t* frame, size_t pitch, double* out_xy, int* n_invalid) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (!frame || !out_xy || pitch < (size_t)e->cfg.frame_width) return fail(MOF_ERR_BAD_ARG, "bad frame/pitch/out");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");  // reference: returns an empty vector
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int cur_slot = 1 - e->prev_slot;
  pack_frame(e->h_stage, frame, pitch, e->cfg.frame_width, e->cfg.frame_height);
  HIP_TRY(hipMemcpyAsync(e->d_frames[cur_slot], e->h_stage, e->frame_bytes, hipMemcpyHostToDevice, e->stream));
  // `first`: the frame is correlated with itself (FftMethod.cpp:1791-1793)
  const uint8_t* prev = e->first ? e->d_frames[cur_slot] : e->d_frames[e->prev_slot];
  mof::PcArgs a = fft_args(e, e->d_frames[cur_slot], 0, prev, 0, (size_t)e->cfg.frame_width, e->d_out);
  FIELD_TRY(launch_field(e, a, 1, e->stream));
  const size_t res = (size_t)e->cfg.grid_x * e->cfg.grid_y * 2;
  HIP_TRY(hipMemcpyAsync(e->h_out, e->d_out, res * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  int bad = 0;
  for (size_t i = 0; i < res; i += 2) {
    out_xy[i] = e->h_out[i];
    out_xy[i + 1] = e->h_out[i + 1];
    if (std::isnan(e->h_out[i])) ++bad;
  }
  if (n_invalid) *n_invalid = bad;
  e->prev_slot = cur_slot;  // imPrev = imCurr.clone(), FftMethod.cpp:1872
  e->first = false;         // :1900
  return MOF_OK;
}

// Long-range geometry (FftMethod.cpp:1685, :1720): same patch size on the quarter-resolution frame,
// sqNum_lr = sqNum / 4 patches per side. Only defined for the reference's own tiling.
static int long_range_args(const mof_fft_engine* e, mof::PcArgs* a) {
  const mof_fft_config& c = e->cfg;
  if (c.origin_x || c.origin_y || c.stride_x != c.patch_size || c.stride_y != c.patch_size)
    return fail(MOF_ERR_UNSUPPORTED, "long-range mode needs the reference tiling (origin 0, stride = patch size)");
  if ((c.frame_width & 3) || (c.frame_height & 3) || c.grid_x < 4 || c.grid_y < 4)
    return fail(MOF_ERR_UNSUPPORTED, "long-range mode needs frame sides divisible by 4 and sqNum >= 4");
  a->grid_x = c.grid_x / 4;
  a->grid_y = c.grid_y / 4;
  a->downscale = 4;
  // the long-range gate is held in ints (`int max_px_speed_lr, max_px_speed_sq_lr`, FftMethod.h:393):
  // max_px_speed_lr = 1 * max_px_speed_t truncates, max_px_speed_sq_lr = pow(max_px_speed_lr, 2) (FftMethod.cpp:1687-1688)
  const int speed_lr = (int)c.max_px_speed;
  a->max_px_speed_sq = (double)(int)std::pow((double)speed_lr, 2);
  return MOF_OK;
}

int mof_fft_long_range_patches(const mof_fft_engine* e) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  mof::PcArgs a{};
  int rc = long_range_args(e, &a);
  return rc ? rc : a.grid_x * a.grid_y;
}

int mof_fft_process_long_range(mof_fft_engine* e, const uint8_t* frame, size_t pitch, double* out_xy, int* n_invalid) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (!frame || !out_xy || pitch < (size_t)e->cfg.frame_width) return fail(MOF_ERR_BAD_ARG, "bad frame/pitch/out");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int cur_slot = 1 - e->prev_slot;
  pack_frame(e->h_stage, frame, pitch, e->cfg.frame_width, e->cfg.frame_height);
  HIP_TRY(hipMemcpyAsync(e->d_frames[cur_slot], e->h_stage, e->frame_bytes, hipMemcpyHostToDevice, e->stream));
  const uint8_t* prev = e->first ? e->d_frames[cur_slot] : e->d_frames[e->prev_slot];  // FftMethod.cpp:1920-1922
  mof::PcArgs a = fft_args(e, e->d_frames[cur_slot], 0, prev, 0, (size_t)e->cfg.frame_width, e->d_out);
  int rc = long_range_args(e, &a);
  if (rc) return rc;
  FIELD_TRY(launch_field(e, a, 1, e->stream));
  const size_t res = (size_t)a.grid_x * a.grid_y * 2;
  HIP_TRY(hipMemcpyAsync(e->h_out, e->d_out, res * sizeof(double), hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  int bad = 0;
  for (size_t i = 0; i < res; i += 2) {
    out_xy[i] = e->h_out[i];
    out_xy[i + 1] = e->h_out[i + 1];
    if (std::isnan(e->h_out[i])) ++bad;
  }
  if (n_invalid) *n_invalid = bad;
  e->prev_slot = cur_slot;  // imPrev = imCurr.clone(), FftMethod.cpp:1992
  e->first = false;         // :2004
  return MOF_OK;
}

int mof_fft_process_long_range_batch_device(mof_fft_engine* e, const uint8_t* d_cur, size_t cur_stride,
                                            const uint8_t* d_prev, size_t prev_stride, size_t pitch, int n_pairs,
                                            double* d_out_xy, void* stream) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs == 0) return MOF_OK;  // an empty batch carries no pointers to check
  if (!d_cur || !d_prev || !d_out_xy || n_pairs < 0 || pitch < (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  if ((unsigned long long)n_pairs * (unsigned long long)(e->cfg.grid_x * e->cfg.grid_y) > 0x7fffffffull)
    return fail(MOF_ERR_BAD_ARG, "batch too large for one launch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  mof::PcArgs a = fft_args(e, d_cur, cur_stride, d_prev, prev_stride, pitch, d_out_xy);
  int rc = long_range_args(e, &a);
  if (rc) return rc;
  if (mof::stream_capturing((hipStream_t)stream)) e->graph_pinned.store(true);
  FIELD_TRY(launch_field(e, a, n_pairs, (hipStream_t)stream));
  return MOF_OK;
}

int mof_fft_process_batch_device(mof_fft_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                 size_t prev_stride, size_t pitch, int n_pairs, double* d_out_xy, void* stream) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs == 0) return MOF_OK;  // an empty batch carries no pointers to check
  if (!d_cur || !d_prev || !d_out_xy || n_pairs < 0 || pitch < (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  if ((unsigned long long)n_pairs * (unsigned long long)(e->cfg.grid_x * e->cfg.grid_y) > 0x7fffffffull)
    return fail(MOF_ERR_BAD_ARG, "batch too large for one launch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  mof::PcArgs a = fft_args(e, d_cur, cur_stride, d_prev, prev_stride, pitch, d_out_xy);
  if (mof::stream_capturing((hipStream_t)stream)) e->graph_pinned.store(true);
  FIELD_TRY(launch_field(e, a, n_pairs, (hipStream_t)stream));
  return MOF_OK;
}

// A video: pair k = (frame k + 1, frame k). 64 x 64 patches run the sequence kernel (one real transform per frame and
// patch, pc_seq_kernel.hip); the other sizes run the pair kernel on cur = frames + 1, prev = frames (no copy either).
static int fft_sequence(mof_fft_engine* e, const uint8_t* d_frames, size_t frame_stride, size_t pitch, int n_frames,
                        double* d_out_xy, void* stream, int channels) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_frames == 0 || n_frames == 1) return MOF_OK;  // no pair
  if (!d_frames || !d_out_xy || n_frames < 0 || pitch < (size_t)channels * (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad sequence arguments");
  const int n_pairs = n_frames - 1;
  if ((unsigned long long)n_pairs * (unsigned long long)(e->cfg.grid_x * e->cfg.grid_y) > 0x7fffffffull)
    return fail(MOF_ERR_BAD_ARG, "sequence too long for one launch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  mof::PcArgs a = fft_args(e, d_frames + frame_stride, frame_stride, d_frames, frame_stride, pitch, d_out_xy);
  a.channels = channels;
  if (mof::stream_capturing((hipStream_t)stream)) e->graph_pinned.store(true);
  static const int run = [] { const char* v = getenv("MOF_FFT_SEQ_RUN"); const int r = v ? atoi(v) : 0; return r >= 1 ? r : 16; }();
  static const bool run_set = [] { const char* v = getenv("MOF_FFT_SEQ_RUN"); return v && atoi(v) >= 1; }();
  static const bool pairs_only = getenv("MOF_FFT_SEQ_PAIRS") != nullptr, half64 = getenv("MOF_FFT_SEQ_HALF64") != nullptr;
  const int n = e->cfg.patch_size;
  const bool half = !e->generic && !e->large && !pairs_only && mof::pc_sequence_half_supported(n) && (n != 64 || half64);
  const bool full = !e->generic && !e->large && !pairs_only && !half && mof::pc_sequence_supported(n);
  // the half-tile kernel's video form (r05): every size it serves by default but 162; MOF_FFT_HALF_SEQ=0 keeps the pair form on consecutive frames
  static const bool half_seq_off = [] { const char* v = getenv("MOF_FFT_HALF_SEQ"); return v && atoi(v) == 0; }();
  // ... and 128 x 128 patches, whose pair form stays on the tuned kernel: on a video the half-tile kernel's sequence form beats the older
  // half-tile sequence kernel (pc_seq_half.hip: one 8-wave workgroup per CU, 192 VGPRs) -- c4seq 93.3 k -> 112 k pairs/s; MOF_FFT_SEQ_HALF128=1 keeps that one
  static const bool old128 = getenv("MOF_FFT_SEQ_HALF128") != nullptr;
  // ... and the planned sizes where only the video form wins (50, 54, 108: pc_half_kernel.hip, MOF_HALF_SEQ_SIZES)
  const int kh_m = e->half_m > 0 ? e->half_m
                                 : ((!e->generic && !e->large && n == 128 && !old128) ? 128 : ((e->generic && e->seq_half_m > 0) ? e->seq_half_m : 0));
  const bool khalf = kh_m > 0 && !pairs_only && !half_seq_off && e->cfg.peak_model == MOF_PEAK_OPENCV && mof::pc_half_sequence_supported(kh_m);
  if (!half && !full && !khalf) {
    FIELD_TRY(launch_field(e, a, n_pairs, (hipStream_t)stream));
    return MOF_OK;
  }
  // the run index rides gridDim.z (at most 65535 per launch): a very long video goes out in several launches
  const size_t per_pair = (size_t)e->cfg.grid_x * e->cfg.grid_y * 2;
  const int max_pairs = 65535 * (run_set ? run : (khalf ? 4 : (full ? 2 : run)));  // (the launchers' own run lengths are at least 4 / 2)
  for (int k0 = 0; k0 < n_pairs; k0 += max_pairs) {
    const int nk = n_pairs - k0 < max_pairs ? n_pairs - k0 : max_pairs;
    mof::PcArgs c = a;
    c.cur = d_frames + (size_t)k0 * frame_stride;  // the sequence kernels index frames, not pairs
    c.out = d_out_xy + (size_t)k0 * per_pair;
    if (khalf) HIP_TRY(mof::launch_pc_half_sequence(c, kh_m, n, nk, run_set ? run : 0, (hipStream_t)stream));  // (0: the launcher picks the run length)
    else if (half) HIP_TRY(mof::launch_pc_sequence_half(c, n, nk, run, (hipStream_t)stream));
    else HIP_TRY(mof::launch_pc_sequence(c, nk, run_set ? run : 0, (hipStream_t)stream));  // (0: the launcher picks the run length)
  }
  return MOF_OK;
}

int mof_fft_process_sequence_device(mof_fft_engine* e, const uint8_t* d_frames, size_t frame_stride, size_t pitch, int n_frames,
                                    double* d_out_xy, void* stream) {
  return fft_sequence(e, d_frames, frame_stride, pitch, n_frames, d_out_xy, stream, 1);
}

int mof_fft_process_sequence_device_bgr(mof_fft_engine* e, const uint8_t* d_frames, size_t frame_stride, size_t pitch, int n_frames,
                                        double* d_out_xy, void* stream) {
  return fft_sequence(e, d_frames, frame_stride, pitch, n_frames, d_out_xy, stream, 3);
}

int mof_fft_process_batch_device_bgr(mof_fft_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                     size_t prev_stride, size_t pitch, int n_pairs, double* d_out_xy, void* stream) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs == 0) return MOF_OK;  // an empty batch carries no pointers to check
  if (!d_cur || !d_prev || !d_out_xy || n_pairs < 0 || pitch < 3 * (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  if ((unsigned long long)n_pairs * (unsigned long long)(e->cfg.grid_x * e->cfg.grid_y) > 0x7fffffffull)
    return fail(MOF_ERR_BAD_ARG, "batch too large for one launch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  mof::PcArgs a = fft_args(e, d_cur, cur_stride, d_prev, prev_stride, pitch, d_out_xy);
  a.channels = 3;
  if (mof::stream_capturing((hipStream_t)stream)) e->graph_pinned.store(true);
  FIELD_TRY(launch_field(e, a, n_pairs, (hipStream_t)stream));
  return MOF_OK;
}

int mof_fft_process_batch_host(mof_fft_engine* e, const uint8_t* cur, size_t cur_stride, const uint8_t* prev,
                               size_t prev_stride, size_t pitch, int n_pairs, double* out_xy) try {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs == 0) return MOF_OK;  // an empty batch carries no pointers to check
  if (!cur || !prev || !out_xy || n_pairs < 0 || pitch < (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  HIP_TRY(hipSetDevice(e->cfg.device));
  mof::RelaxedCapture relaxed;  // the pipeline's first-call allocations must not disturb a capture on another thread
  const size_t res = (size_t)e->cfg.grid_x * e->cfg.grid_y * 2 * sizeof(double);
  {
    std::lock_guard<std::mutex> lock(e->host_mu);
    if (!e->host_pipe) e->host_pipe = new mof::HostPipe(e->frame_bytes, &res, 1);
  }
  const mof::HostPipe::Out out{out_xy, res};
  hipError_t he = hipSuccess;
  // chunks of frames go up on the pipe's copy stream while the engine's stream runs the previous chunk through the DEVICE batch entry
  // (its kernels, its bits); a video -- cur = prev + one frame -- arrives as the two views of ONE uploaded run
  const int rc = e->host_pipe->process(
      cur, cur_stride, prev, prev_stride, pitch, e->cfg.frame_width, e->cfg.frame_height, n_pairs, &out, e->stream,
      [e](const mof::HostPipe::Chunk& c, hipStream_t s) {
        return mof_fft_process_batch_device(e, c.d_cur, c.stride, c.d_prev, c.stride, (size_t)e->cfg.frame_width, c.count,
                                            static_cast<double*>(c.d_out[0]), s);
      },
      &he);
  if (rc == -1) return fail(MOF_ERR_HIP, "host batch pipeline: %s", hipGetErrorString(he));
  return rc;
} catch (const std::bad_alloc&) {
  return fail(MOF_ERR_NO_MEMORY, "mof_fft_process_batch_host: out of host memory");
}

/* Pinned host memory for callers that do not link HIP themselves: frames handed to the *_batch_host entries from such memory are DMA'd
 * from where they lie (host_pipe.hpp). */
int mof_host_alloc(size_t bytes, void** out) {
  if (!out || bytes == 0) return fail(MOF_ERR_BAD_ARG, "mof_host_alloc: null result pointer or zero bytes");
  *out = nullptr;
  HIP_TRY(hipHostMalloc(out, bytes, hipHostMallocDefault));
  return MOF_OK;
}
int mof_host_free(void* p) {
  if (!p) return MOF_OK;
  HIP_TRY(hipHostFree(p));
  return MOF_OK;
}
int mof_host_register(void* p, size_t bytes) {
  if (!p || bytes == 0) return fail(MOF_ERR_BAD_ARG, "mof_host_register: null pointer or zero bytes");
  HIP_TRY(hipHostRegister(p, bytes, hipHostRegisterDefault));
  return MOF_OK;
}
int mof_host_unregister(void* p) {
  if (!p) return fail(MOF_ERR_BAD_ARG, "mof_host_unregister: null pointer");
  HIP_TRY(hipHostUnregister(p));
  return MOF_OK;
}

int mof_fft_sync(mof_fft_engine* e) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MOF_OK;
}

/* ------------------------------------------------------------------------------------------ */
/* Block matching                                                                             */
/* ------------------------------------------------------------------------------------------ */

int mof_bm_config_block_method(mof_bm_config* cfg, int frame_size, int sample_point_size, int scan_radius) {
  if (!cfg || frame_size < 1 || sample_point_size < 1 || scan_radius < 0)
    return fail(MOF_ERR_BAD_ARG, "bad BlockMethod geometry");
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->frame_width = cfg->frame_height = frame_size;
  cfg->block_size = sample_point_size;
  cfg->step_size = 0;
  cfg->scan_radius = scan_radius;
  cfg->grid_x = cfg->grid_y = (frame_size - scan_radius * 2) / sample_point_size;  // BlockMethod.cpp:11
  cfg->low_contrast_rule = 0;
  return MOF_OK;
}

int mof_bm_config_fast_spaced(mof_bm_config* cfg, int width, int height, int sample_point_size, int step_size,
                              int scan_radius) {
  if (!cfg || width < 1 || height < 1 || sample_point_size < 1 || step_size < 0 || scan_radius < 0)
    return fail(MOF_ERR_BAD_ARG, "bad FastSpacedBM geometry");
  std::memset(cfg, 0, sizeof(*cfg));
  cfg->frame_width = width;
  cfg->frame_height = height;
  cfg->block_size = sample_point_size;
  cfg->step_size = step_size;
  cfg->scan_radius = scan_radius;
  const int S = sample_point_size + step_size;            // FastSpacedBMMethod_OCL.cpp:82-83
  cfg->grid_x = (width - scan_radius * 2) / S;            // :90
  cfg->grid_y = (height - scan_radius * 2) / S;
  cfg->low_contrast_rule = 1;
  return MOF_OK;
}

static int validate_bm(const mof_bm_config* c) {
  if (!c) return fail(MOF_ERR_BAD_ARG, "null config");
  if (c->frame_width < 1 || c->frame_height < 1 || c->grid_x < 1 || c->grid_y < 1 || c->step_size < 0)
    return fail(MOF_ERR_BAD_ARG, "bad block-matching geometry");
  if (!mof::bm_config_supported(c->block_size, c->scan_radius))
    return fail(MOF_ERR_UNSUPPORTED, "block_size %d / scan_radius %d not supported by the HIP kernel "
                "(block multiple of 4 in 4..128, radius 1..48, window within the LDS)", c->block_size, c->scan_radius);
  const long S = c->block_size + c->step_size;
  if ((c->grid_x - 1) * S + c->block_size + 2 * c->scan_radius > c->frame_width ||
      (c->grid_y - 1) * S + c->block_size + 2 * c->scan_radius > c->frame_height)
    return fail(MOF_ERR_BAD_ARG, "block grid leaves the frame");
  return MOF_OK;
}

static mof::BmArgs bm_args(const mof_bm_engine* e, const uint8_t* cur, size_t cs, const uint8_t* prev, size_t ps,
                           size_t pitch, int8_t* dx, int8_t* dy, int8_t* mode, int channels = 1) {
  mof::BmArgs a{};
  a.channels = channels;
  a.cur = cur;
  a.prev = prev;
  a.cur_stride = cs;
  a.prev_stride = ps;
  a.pitch = pitch;
  a.grid_x = e->cfg.grid_x;
  a.grid_y = e->cfg.grid_y;
  a.block = e->cfg.block_size;
  a.step = e->cfg.step_size;
  a.radius = e->cfg.scan_radius;
  a.low_contrast_rule = e->cfg.low_contrast_rule;
  a.dx = dx;
  a.dy = dy;
  a.mode = mode;
  return a;
}

int mof_bm_create(const mof_bm_config* cfg, mof_bm_engine** out) {
  if (!out) return fail(MOF_ERR_BAD_ARG, "null out");
  *out = nullptr;
  int rc = validate_bm(cfg);
  if (rc) return rc;
  rc = select_device(cfg->device);
  if (rc) return rc;
  mof::RelaxedCapture relaxed;
  mof_bm_engine* e = new (std::nothrow) mof_bm_engine();
  if (!e) return fail(MOF_ERR_NO_MEMORY, "out of host memory");
  e->cfg = *cfg;
  e->frame_bytes = (size_t)cfg->frame_width * cfg->frame_height;
  const size_t nb = (size_t)cfg->grid_x * cfg->grid_y;
#define CREATE_TRY(expr)                                                                        \
  do {                                                                                          \
    hipError_t _e = (expr);                                                                     \
    if (_e != hipSuccess) {                                                                     \
      fail(MOF_ERR_HIP, "%s: %s", #expr, hipGetErrorString(_e));                                \
      mof_bm_destroy(e);                                                                        \
      return MOF_ERR_HIP;                                                                       \
    }                                                                                           \
  } while (0)
  CREATE_TRY(hipStreamCreateWithFlags(&e->stream, hipStreamNonBlocking));
  CREATE_TRY(hipMalloc(&e->d_frames[0], e->frame_bytes));
  CREATE_TRY(hipMalloc(&e->d_frames[1], e->frame_bytes));
  CREATE_TRY(mof::fill_on(e->stream, e->d_frames[0], 0, e->frame_bytes));  // imPrev = Scalar(0), BlockMethod.cpp:17-18
  CREATE_TRY(mof::fill_on(e->stream, e->d_frames[1], 0, e->frame_bytes));
  CREATE_TRY(hipMalloc(&e->d_dx, nb));
  CREATE_TRY(hipMalloc(&e->d_dy, nb));
  CREATE_TRY(hipMalloc(&e->d_mode, 8));
  CREATE_TRY(hipHostMalloc(&e->h_res, 2 * nb + 8, hipHostMallocDefault));
  CREATE_TRY(hipHostMalloc(&e->h_stage, e->frame_bytes, hipHostMallocDefault));
#undef CREATE_TRY
  *out = e;
  return MOF_OK;
}

static void bm_destroy_now(void* p) {
  mof_bm_engine* e = static_cast<mof_bm_engine*>(p);
  mof::RelaxedCapture relaxed;
  (void)hipSetDevice(e->cfg.device);
  if (e->stream) (void)hipStreamSynchronize(e->stream);
  delete e->host_pipe;
  if (e->d_frames[0]) (void)hipFree(e->d_frames[0]);
  if (e->d_frames[1]) (void)hipFree(e->d_frames[1]);
  if (e->d_dx) (void)hipFree(e->d_dx);
  if (e->d_dy) (void)hipFree(e->d_dy);
  if (e->d_mode) (void)hipFree(e->d_mode);
  if (e->h_res) (void)hipHostFree(e->h_res);
  if (e->h_stage) (void)hipHostFree(e->h_stage);
  if (e->d_up[0]) (void)hipFree(e->d_up[0]);
  if (e->d_up[1]) (void)hipFree(e->d_up[1]);
  if (e->d_sad9) (void)hipFree(e->d_sad9);
  if (e->h_sad9) (void)hipHostFree(e->h_sad9);
  if (e->stream) (void)hipStreamDestroy(e->stream);
  delete e;
}

void mof_bm_destroy(mof_bm_engine* e) {
  if (!e) return;
  if (e->graph_pinned.load()) {
    mof::park_engine(&bm_destroy_now, e);
    return;
  }
  bm_destroy_now(e);
}

int mof_bm_release_graphs(mof_bm_engine* e) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  e->graph_pinned.store(false);
  return MOF_OK;
}

int mof_bm_graph_pinned(const mof_bm_engine* e) { return e && e->graph_pinned.load() ? 1 : 0; }

int mof_bm_set_prev(mof_bm_engine* e, const uint8_t* frame, size_t pitch) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (!frame || pitch < (size_t)e->cfg.frame_width) return fail(MOF_ERR_BAD_ARG, "bad frame/pitch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  pack_frame(e->h_stage, frame, pitch, e->cfg.frame_width, e->cfg.frame_height);
  HIP_TRY(hipMemcpyAsync(e->d_frames[e->prev_slot], e->h_stage, e->frame_bytes, hipMemcpyHostToDevice, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MOF_OK;
}

int mof_bm_reset(mof_bm_engine* e) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipMemsetAsync(e->d_frames[e->prev_slot], 0, e->frame_bytes, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MOF_OK;
}

int mof_bm_process(mof_bm_engine* e, const uint8_t* frame, size_t pitch, int8_t* dx, int8_t* dy, int8_t* mode_xy) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (!frame || !dx || !dy || pitch < (size_t)e->cfg.frame_width) return fail(MOF_ERR_BAD_ARG, "bad arguments");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int cur_slot = 1 - e->prev_slot;
  const size_t nb = (size_t)e->cfg.grid_x * e->cfg.grid_y;
  pack_frame(e->h_stage, frame, pitch, e->cfg.frame_width, e->cfg.frame_height);
  HIP_TRY(hipMemcpyAsync(e->d_frames[cur_slot], e->h_stage, e->frame_bytes, hipMemcpyHostToDevice, e->stream));
  mof::BmArgs a = bm_args(e, e->d_frames[cur_slot], 0, e->d_frames[e->prev_slot], 0, (size_t)e->cfg.frame_width,
                          e->d_dx, e->d_dy, e->d_mode);
  HIP_TRY(mof::launch_bm_scan(a, 1, e->stream));
  HIP_TRY(mof::launch_bm_mode(a, 1, e->stream));
  HIP_TRY(hipMemcpyAsync(e->h_res, e->d_dx, nb, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(e->h_res + nb, e->d_dy, nb, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipMemcpyAsync(e->h_res + 2 * nb, e->d_mode, 8, hipMemcpyDeviceToHost, e->stream));
  HIP_TRY(hipStreamSynchronize(e->stream));
  std::memcpy(dx, e->h_res, nb);
  std::memcpy(dy, e->h_res + nb, nb);
  if (mode_xy) {
    mode_xy[0] = e->h_res[2 * nb];
    mode_xy[1] = e->h_res[2 * nb + 1];
  }
  e->prev_slot = cur_slot;  // imPrev = imCurr.clone(), BlockMethod.cpp:89
  e->have_pair = true;
  return MOF_OK;
}

int mof_bm_refine(mof_bm_engine* e, int fullpix_x, int fullpix_y, int passes, int faithful, double* out_xy) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (!out_xy || passes < 1 || passes > 4) return fail(MOF_ERR_BAD_ARG, "bad refine arguments");
  if (!e->have_pair) return fail(MOF_ERR_NOT_INIT, "mof_bm_refine needs a preceding mof_bm_process call");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  const int w = e->cfg.frame_width, h = e->cfg.frame_height, W2 = 2 * w, H2 = 2 * h;
  if (!e->d_up[0]) {
    mof::RelaxedCapture relaxed;
    HIP_TRY(hipMalloc(&e->d_up[0], (size_t)W2 * H2));
    HIP_TRY(hipMalloc(&e->d_up[1], (size_t)W2 * H2));
    HIP_TRY(hipMalloc(&e->d_sad9, 9 * sizeof(unsigned long long)));
    HIP_TRY(hipHostMalloc(&e->h_sad9, 9 * sizeof(unsigned long long), hipHostMallocDefault));
  }
  // after mof_bm_process the frame just processed sits in the "previous" slot, its predecessor in the other one
  const uint8_t* cur = e->d_frames[e->prev_slot];
  const uint8_t* prev = e->d_frames[1 - e->prev_slot];
  int tx = fullpix_x, ty = fullpix_y, scale = 1;
  for (int i = 1; i <= passes; ++i) {
    scale *= 2;
    tx *= 2;
    ty *= 2;  // BlockMethod.cpp:106-107
    if (i == 1) {
      // :110-111 -- both images go to twice the ORIGINAL size; the reference resizes the "previous" one from the
      // CURRENT image. Later passes resize to the same 2x size, i.e. copy.
      HIP_TRY(mof::launch_bm_resize2x(cur, (size_t)w, w, h, e->d_up[0], e->stream));
      HIP_TRY(mof::launch_bm_resize2x(faithful ? cur : prev, (size_t)w, w, h, e->d_up[1], e->stream));
    }
    int spx, spy;  // :113-121
    if (tx < 0 && ty < 0) { spx = -tx + 1; spy = -ty + 1; }
    else if (tx < 0 && ty >= 0) { spx = -tx + 1; spy = 1; }
    else if (tx >= 0 && ty < 0) { spx = 1; spy = -ty + 1; }
    else { spx = 1; spy = 1; }
    const int cw = W2 - ((tx < 0 ? -tx : tx) + 2), ch = H2 - ((ty < 0 ? -ty : ty) + 2);  // :123
    if (cw <= 0 || ch <= 0) return fail(MOF_ERR_BAD_ARG, "refine: offset (%d, %d) leaves no cut-out", tx, ty);
    HIP_TRY(hipMemsetAsync(e->d_sad9, 0, 9 * sizeof(unsigned long long), e->stream));
    HIP_TRY(mof::launch_bm_refine_sad(e->d_up[0], e->d_up[1], W2, spx, spy, cw, ch, e->d_sad9, e->stream));
    HIP_TRY(hipMemcpyAsync(e->h_sad9, e->d_sad9, 9 * sizeof(unsigned long long), hipMemcpyDeviceToHost, e->stream));
    HIP_TRY(hipStreamSynchronize(e->stream));
    int best = 0;  // cv::minMaxLoc: first minimum, row-major over (m, n)  (:140)
    for (int k = 1; k < 9; ++k)
      if ((int)e->h_sad9[k] < (int)e->h_sad9[best]) best = k;  // absDiffsMatSubpix is CV_32S
    tx += best % 3 - 1;
    ty += best / 3 - 1;  // :142
  }
  out_xy[0] = (double)((float)tx / (float)scale);  // :144
  out_xy[1] = (double)((float)ty / (float)scale);
  return MOF_OK;
}

int mof_bm_process_batch_device(mof_bm_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                size_t prev_stride, size_t pitch, int n_pairs, int8_t* d_dx, int8_t* d_dy,
                                int8_t* d_mode, void* stream) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs == 0) return MOF_OK;  // an empty batch carries no pointers to check
  if (!d_cur || !d_prev || !d_dx || !d_dy || !d_mode || n_pairs < 0 || pitch < (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  if ((unsigned long long)n_pairs * (unsigned long long)(e->cfg.grid_x * e->cfg.grid_y) > 0x7fffffffull)
    return fail(MOF_ERR_BAD_ARG, "batch too large for one launch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t s = (hipStream_t)stream;  // (a captured block-matching batch reads no engine-owned memory: no pin)
  mof::BmArgs a = bm_args(e, d_cur, cur_stride, d_prev, prev_stride, pitch, d_dx, d_dy, d_mode);
  HIP_TRY(mof::launch_bm_scan(a, n_pairs, s));
  HIP_TRY(mof::launch_bm_mode(a, n_pairs, s));
  return MOF_OK;
}

int mof_bm_process_batch_device_bgr(mof_bm_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                    size_t prev_stride, size_t pitch, int n_pairs, int8_t* d_dx, int8_t* d_dy,
                                    int8_t* d_mode, void* stream) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs == 0) return MOF_OK;
  if (!d_cur || !d_prev || !d_dx || !d_dy || !d_mode || n_pairs < 0 || pitch < 3 * (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  if ((unsigned long long)n_pairs * (unsigned long long)(e->cfg.grid_x * e->cfg.grid_y) > 0x7fffffffull)
    return fail(MOF_ERR_BAD_ARG, "batch too large for one launch");
  BusyGuard g(e->busy);
  if (!g.owned) return fail(MOF_ERR_BUSY, "engine busy");
  HIP_TRY(hipSetDevice(e->cfg.device));
  hipStream_t s = (hipStream_t)stream;  // (a captured block-matching batch reads no engine-owned memory: no pin)
  mof::BmArgs a = bm_args(e, d_cur, cur_stride, d_prev, prev_stride, pitch, d_dx, d_dy, d_mode, 3);
  HIP_TRY(mof::launch_bm_scan(a, n_pairs, s));
  HIP_TRY(mof::launch_bm_mode(a, n_pairs, s));
  return MOF_OK;
}

int mof_bm_process_batch_host(mof_bm_engine* e, const uint8_t* cur, size_t cur_stride, const uint8_t* prev,
                              size_t prev_stride, size_t pitch, int n_pairs, int8_t* dx, int8_t* dy, int8_t* mode) try {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  if (n_pairs == 0) return MOF_OK;  // an empty batch carries no pointers to check
  if (!cur || !prev || !dx || !dy || !mode || n_pairs < 0 || pitch < (size_t)e->cfg.frame_width)
    return fail(MOF_ERR_BAD_ARG, "bad batch arguments");
  HIP_TRY(hipSetDevice(e->cfg.device));
  mof::RelaxedCapture relaxed;
  const size_t nb = (size_t)e->cfg.grid_x * e->cfg.grid_y;
  const size_t bpp[3] = {nb, nb, 8};
  {
    std::lock_guard<std::mutex> lock(e->host_mu);
    if (!e->host_pipe) e->host_pipe = new mof::HostPipe(e->frame_bytes, bpp, 3);
  }
  const mof::HostPipe::Out outs[3] = {{dx, nb}, {dy, nb}, {mode, 8}};
  hipError_t he = hipSuccess;
  const int rc = e->host_pipe->process(
      cur, cur_stride, prev, prev_stride, pitch, e->cfg.frame_width, e->cfg.frame_height, n_pairs, outs, e->stream,
      [e](const mof::HostPipe::Chunk& c, hipStream_t s) {
        return mof_bm_process_batch_device(e, c.d_cur, c.stride, c.d_prev, c.stride, (size_t)e->cfg.frame_width, c.count,
                                           static_cast<int8_t*>(c.d_out[0]), static_cast<int8_t*>(c.d_out[1]), static_cast<int8_t*>(c.d_out[2]), s);
      },
      &he);
  if (rc == -1) return fail(MOF_ERR_HIP, "host batch pipeline: %s", hipGetErrorString(he));
  return rc;
} catch (const std::bad_alloc&) {
  return fail(MOF_ERR_NO_MEMORY, "mof_bm_process_batch_host: out of host memory");
}

int mof_bm_sync(mof_bm_engine* e) {
  if (!e) return fail(MOF_ERR_NOT_INIT, "null engine");
  HIP_TRY(hipSetDevice(e->cfg.device));
  HIP_TRY(hipStreamSynchronize(e->stream));
  return MOF_OK;
}

}  // extern "C"
