// mof_kernels.h -- internal launch interface between the C ABI (mof_capi.hip) and the
// gfx950 kernels (pc_kernel.hip, bm_kernel.hip). Not installed; the public surface is include/mof.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

namespace mof {

// ---- K1: per-patch FFT phase correlation ----------------------------------------------------
struct PcArgs {
  const uint8_t* cur;     // device, pair k at cur + k*cur_stride
  const uint8_t* prev;
  size_t cur_stride;      // bytes between consecutive pairs' frames
  size_t prev_stride;
  size_t pitch;           // bytes per frame row
  int grid_x, grid_y;     // patches per frame
  int origin_x, origin_y;
  int stride_x, stride_y;
  int total;              // n_pairs * grid_x * grid_y (set by the launcher)
  int stagger_div, stagger_units;  // start-up stagger of co-resident workgroups (set by the launcher)
  int channels;           // 1 = gray frames; 3 = interleaved BGR8, CV_RGB2GRAY fused into the load
  int downscale;          // 1, or 4 = long-range mode (quarter-resolution patches formed on the fly)
  int peak_model;         // 0 = cv::phaseCorrelate (useOCL=false), 1 = the OpenCL kernel's model (useOCL=true)
  int search_radius;      // peak_model 1: SEARCH_RADIUS (FftMethod.cpp:820)
  double max_px_speed_sq; // FftMethod.cpp:1686
  const float* twiddles;  // device, N (cos, -sin) pairs, computed in double on the host
  double* out;            // device, [pair][patch] (x, y)
};

// 1024 dwords behind the 64 twiddles of an N = 64 engine: the f16 hi / lo fragments of the radix-16 DFT matrix, in the lane order
// of v_mfma_f32_32x32x16_f16's A operand (pc_passes3.hpp, fwd3_rows_mfma)
void pc_mfma_s1_fragments(uint32_t* out);

// Run-time plan of the general K1 (pc_kernel_generic.hip): any samplePointSize n whose padded transform size
// m = cv::getOptimalDFTSize(n) gives an m x m complex tile that fits one CU's LDS (m <= 135)
struct PcPlan {
  int n, m;          // patch size; padded transform size (smallest 2^a 3^b 5^c >= n)
  int pitch;         // complex elements per tile row
  int skew_mask;     // ~0: tile column c sits at c + (c >> 3); 0: no skew
  int threads;       // workgroup size (multiple of 64)
  int n_stages;      // Stockham stages of a 1-D transform
  int radix[8];      // their radices, each of {2, 3, 4, 5, 8}, product m
  uint32_t radix_packed;  // the same, four bits per stage (what the kernel reads)
  int hermitian;     // 1: m even, the inverse runs on the half spectrum; 0: m odd, full complex inverse
  int lds_bytes;     // dynamic LDS of a workgroup
};
int pc_optimal_dft_size(int n);                          // cv::getOptimalDFTSize
int pc_radix_chain(int m, int* radix, int max_stages);   // number of stages, -1 if m is not 5-smooth
bool pc_build_plan(int n, PcPlan* out);                  // false: n needs the large-patch pipeline (or is < 2)
hipError_t pc_configure_generic();
hipError_t launch_pc_generic(const PcArgs& a, const PcPlan& plan, int n_pairs, hipStream_t stream);

// ---- images too large for a CU (padded side m > 135): the planned pipeline through HBM scratch (pc_large_kernel.hip) ----
bool pc_build_line_plan(int n, PcPlan* out);  // n, m, radix chain only (no LDS tile); false for n < 2 or m > 960
// where image f of a launch comes from
struct PclSrc {
  const uint8_t* base[2];  // paired: [0] = cur frames, [1] = prev frames; else base[0] alone
  size_t stride[2];        // bytes between consecutive frame pairs (paired) / images (not paired)
  size_t pitch;            // bytes per frame row
  int paired;              // 1: image f = 2 (pair * patches + patch) + which (0 cur, 1 prev); 0: image f = base[0] + f * stride[0];
                           // 2 (r06, a VIDEO): image f = frame * patches + patch of base[0] + frame * stride[0] -- every frame transformed once
  int grid_x, grid_y, origin_x, origin_y, stride_x, stride_y;  // patch grid inside a frame (paired)
  int sums_stride;         // ints between the exact-sum quadruples of consecutive images (sr_rows_real_src_kernel; 0 = 4: a dense array)
};
struct PclFinal {
  const float* Dt;         // [pairs][m/2 + 1][m] complex
  const float2* cand;      // [pairs][n_cand]
  int n_cand;              // set by the launcher
  const float* twiddles;   // m (cos, -sin) pairs
  int m, n;                // set by the launcher from the plan
  int mode;                // 0: scaleRotationEstimator (out[pair][4] = scale, rot, pt.x, pt.y); 1: FftMethod (out[pair][2] = shift or NaN)
  double M_log;            // mode 0: log-polar magnitude
  double max_px_speed_sq;  // mode 1
  double* out;
  const int* flags;        // mode 1, nullable: [2 pairs] constant-patch flags written by pcl_rows (bit 0: not constant, bit 1: pixel 0 != 0)
  const float* cdc;        // mode 1, with flags: [pairs] DC bin of the cross-power spectrum
};
size_t pcl_zh_floats(const PcPlan& pl);  // floats of one image's row half-spectra Zh: (m/2 + 1) * m complex
int pcl_candidates(const PcPlan& pl);    // peak candidates per pair
// L5: images -> Zh (image f at zh + f * zh_stride floats); flags (nullable, ZEROED by the caller): 1 int per image
hipError_t launch_pcl_rows(const PclSrc& src, const PcPlan& pl, const float* twiddles, float* zh, size_t zh_stride, int* flags,
                           int n_images, int channels, int downscale, hipStream_t stream);
// L6: pair p = (cur: zh_cur + p * zh_stride, prev: zh_prev + p * zh_stride) -> Dt[p]; cdc nullable
hipError_t launch_pcl_cols(const float* zh_prev, const float* zh_cur, size_t zh_stride, const PcPlan& pl, const float* twiddles,
                           float* Dt, float* cdc, const int* flags, int n_pairs, hipStream_t stream);
// L7 + L8 (a.Dt, a.cand, a.twiddles, a.mode, a.out [, a.M_log | a.max_px_speed_sq, a.flags, a.cdc])
hipError_t launch_pcl_peak(const PclFinal& a, const PcPlan& pl, int n_pairs, hipStream_t stream, bool candidates_done = false);
// C_dc of every pair from its row spectra (for pipelines whose column kernel does not hand it out)
hipError_t launch_pcl_cdc(const float* zh_prev, const float* zh_cur, size_t zh_stride, int m, float* cdc, int n_pairs, hipStream_t stream);

// ---- the fused kernel on a HALF-size tile (pc_half_kernel.hip): even padded sizes m in (135, 192] -- and 64 / 96 / 120 / 128 for A/B --,
// cv::phaseCorrelate's model on full-resolution gray or BGR8 frames. m = padded transform size (a.twiddles: m entries), n = patch size
bool pc_half_supported(int m);
int pc_half_workgroups_per_cu(int m);
hipError_t pc_configure_half(int m);
hipError_t launch_pc_half(const PcArgs& a, int m, int n, int n_pairs, hipStream_t stream);
// ... on a video (r05): runs of `run` consecutive pairs per workgroup, a frame's spectrum kept in registers for the next pair; a.cur = the
// launch's first frame, frame f at a.cur + f * a.cur_stride (as launch_pc_sequence)
bool pc_half_sequence_supported(int m);
hipError_t launch_pc_half_sequence(const PcArgs& a, int m, int n, int n_pairs, int run, hipStream_t stream);

bool pc_patch_size_supported(int n);  // the hand-tuned instantiations: 32, 64, 120, 128
const char* pc_kernel_variant(int patch_size);
hipError_t pc_configure(int patch_size);  // once per device before the first launch
hipError_t launch_pc_field(const PcArgs& a, int patch_size, int n_pairs, hipStream_t stream);
// Frame sequences (pc_seq_kernel.hip; 64 x 64 patches): a.cur = frame 0, a.cur_stride = bytes between frames, pair k =
// (frame k + 1, frame k) for k < n_pairs; one workgroup per patch position walks `run` consecutive pairs
bool pc_sequence_supported(int patch_size);
hipError_t pc_configure_sequence();
hipError_t launch_pc_sequence(const PcArgs& a, int n_pairs, int run, hipStream_t stream);
// The same on a half-size tile (pc_seq_half.hip): 128 x 128 patches (two workgroups per CU); 64 x 64 for A/B only
bool pc_sequence_half_supported(int patch_size);
hipError_t pc_configure_sequence_half(int patch_size);
hipError_t launch_pc_sequence_half(const PcArgs& a, int patch_size, int n_pairs, int run, hipStream_t stream);
// The PAIR form on the half tile (pc_seq_half.hip, r05; 128 x 128 patches): persistent workgroups, two per CU, the previous image's
// spectrum parked in a per-workgroup slab of device memory (n_slabs slabs of pc_pair_half_slab_floats floats, owned by the engine)
bool pc_pair_half_supported(int patch_size);
size_t pc_pair_half_slab_floats(int patch_size);
hipError_t pc_configure_pair_half(int patch_size);
hipError_t launch_pc_pair_half(const PcArgs& a, int patch_size, int n_pairs, float* slabs, int n_slabs, hipStream_t stream);
// N = 64, quad-per-line formulation (pc_kernel_quad.hip)
hipError_t pc_configure_quad64();
hipError_t launch_pc_field_quad64(const PcArgs& a, int n_pairs, hipStream_t stream);
// N = 120 (15 x 8) lives in pc_kernel_mixed.hip
hipError_t pc_configure_120();
hipError_t launch_pc_field_120(const PcArgs& a, int n_pairs, hipStream_t stream);

// ---- K2/K3: SAD block scan + histogram mode -------------------------------------------------
struct BmArgs {
  const uint8_t* cur;
  const uint8_t* prev;
  size_t cur_stride, prev_stride, pitch;
  int grid_x, grid_y;
  int block, step, radius;
  int low_contrast_rule;
  int channels;  // 1: gray frames; 3: interleaved BGR8, CV_RGB2GRAY fused into the staging loads
  int8_t* dx;    // [pair][by*grid_x+bx]
  int8_t* dy;
  int8_t* mode;  // [pair][8]
};

bool bm_config_supported(int block, int radius);
hipError_t launch_bm_scan(const BmArgs& a, int n_pairs, hipStream_t stream);
hipError_t launch_bm_mode(const BmArgs& a, int n_pairs, hipStream_t stream);
// BlockMethod::Refine building blocks (K9 2x resize, K10 nine cut-out SADs)
hipError_t launch_bm_resize2x(const uint8_t* src, size_t pitch, int w, int h, uint8_t* dst, hipStream_t stream);
hipError_t launch_bm_refine_sad(const uint8_t* A, const uint8_t* B, int W2, int spx, int spy, int cw, int ch,
                                unsigned long long* out9, hipStream_t stream);

// ---- K4..K8: scale / rotation estimator (log-polar remap + whole-frame phase correlation) ----
struct SrMapEntry {
  int16_t ax, ay;   // anchor pixel (map coordinate >> 5), saturated to int16
  uint16_t widx;    // (fy & 31) * 32 + (fx & 31): row of the 2-D weight table
  uint16_t valid;   // anchor inside the source (BORDER_TRANSPARENT otherwise)
};

// Bounding box (source pixels) of the K x K footprints of the valid pixels of one 8 x 8 destination tile, taps that
// leave the image reflected back in (BORDER_REFLECT_101); w = h = 0 when the tile has no valid pixel.
struct SrTileBox {
  int16_t x0, y0, w, h;
};

struct SrLpArgs {
  const uint8_t* src;   // image i at src + i*src_stride, `pitch` bytes per row, res x res pixels
  size_t src_stride, pitch;
  uint8_t* dst;         // image i at dst + i*dst_stride, tightly packed res*res
  size_t dst_stride;
  const SrMapEntry* map;   // res*res
  const int16_t* weights;  // [1024][K*K]
  const uint32_t* wplanes; // [1024][K*K/2 + 2]: signed-byte planes of the same weights (sr_weight_planes), staged kernel
  const SrTileBox* sboxes; // [res/16 * res/16]: boxes of the 16 x 16 super-tiles (four 8 x 8 tiles of one workgroup), or null
  int sbox_dwords_max;     // dwords of the largest super-tile box as the kernel lays it out
  int res;
  int zero_invalid;        // 1: pixels mapped outside the source are written as 0 (destination known to start as zeros)
  const SrTileBox* boxes;  // [tiles*tiles] for THIS interpolation's footprint size
  int lds_per_wave;        // bytes of LDS one wave needs for the largest box (multiple of 16)
  int box_dwords_max;      // dwords of the largest box
};

struct SrPcArgs {
  const uint8_t* lp_cur;   // pair k at + k*lp_stride, res*res u8 each
  const uint8_t* lp_prev;
  size_t lp_stride;
  const float* twiddles;   // res (cos, -sin) pairs
  float* Zt;               // scratch [pairs][res u][res v] complex: row transforms, stored transposed
  float* Dt;               // scratch [pairs][res/2+1 u][res y] complex: half spectrum after the column passes
  float2* cand;            // scratch [pairs][n_cand] (value, shifted index)
  int n_cand;
  double M;                // log-polar magnitude
  double* out;             // [pairs][4] = scale, rot, pt.x, pt.y
  int* degen;              // pair pipeline only (else null): [pairs], K6 -> K8: one of the two log-polar images is all zero
};

// host-side tables of cv::logPolar / cv::remap (mof_sr.hip); exposed so that the CPU suite can compare them with the oracle
std::vector<SrMapEntry> sr_logpolar_map(int res, double M, int variant);
std::vector<int16_t> sr_weight_table(int ksize /* 4 cubic, 8 Lanczos4 */);  // [32*32][ksize*ksize], each summing to 2^15
// The same weights as two planes of signed bytes, w = 256 hi + lo, four taps per dword, for v_dot4c_i32_i8 against
// (pixel - 128): per table row K*K/4 dwords hi, K*K/4 dwords lo, then 128 * sum(w) and the index of the one tap whose
// hi would be 128 (kept at 127; -1 if none): sum w p = 256 sum hi p' + sum lo p' + 128 sum w [+ 256 p'(tap)].
std::vector<uint32_t> sr_weight_planes(const std::vector<int16_t>& weights, int ksize);
// per-tile footprint boxes for a ksize x ksize kernel; *lds_per_wave receives the LDS bytes of the largest one
// tile = 8 (per-wave boxes) or 16 (super-tile boxes shared by a workgroup)
std::vector<SrTileBox> sr_tile_boxes(const std::vector<SrMapEntry>& map, int res, int ksize, int* lds_per_wave, int tile_px);

bool sr_resolution_supported(int res);      // tuned transforms (K5s / K6s / K7) exist for this resolution
bool sr_pair_kernels_supported(int res);
bool sr_transform_size_tuned(int m, bool* exact_nyquist);  // the tuned transforms exist for transform size m (the FFT engine's list)   // ... and the packed pair kernels K5 / K6 (240, 256, 480)
int sr_candidates(int res);
hipError_t launch_sr_logpolar(const SrLpArgs& a, int interp /*2 cubic, 4 lanczos4*/, int n_images, hipStream_t stream);
hipError_t launch_sr_phase_correlate(const SrPcArgs& a, int res, int n_pairs, hipStream_t stream);
// K7 + K8 alone (uses a.Dt, a.cand, a.twiddles, a.M, a.out)
hipError_t launch_sr_peak(const SrPcArgs& a, int res, int n_pairs, hipStream_t stream);
// Sequence pipeline (sr_seq_kernel.hip). Zh = the row half-spectra of ONE log-polar image, doubled and transposed:
// [(res/2 + 1) u][res v] complex floats = sr_zh_floats(res) floats.
size_t sr_zh_floats(int res);
hipError_t launch_sr_identity(double* out4, hipStream_t stream);  // (1, 0, 0, 0): the first frame's result (:74)
// K5s: images lp + f * lp_stride (res * res u8, tightly packed rows) -> zh + f * zh_stride (strides: bytes / floats)
hipError_t launch_sr_rows_real(const uint8_t* lp, size_t lp_stride, const float* twiddles, float* zh, size_t zh_stride, int res,
                               int n_frames, hipStream_t stream);
// K6s: pair p = (cur: zh_cur + p * zh_stride, prev: zh_prev + p * zh_stride) -> Dt[p]; `run` > 1 lets one wave walk that
// many consecutive pairs re-using cur(p) as prev(p + 1) -- only valid when zh_cur == zh_prev + zh_stride (a sequence)
// K5s on patches of frames (FftMethod patches of 240 / 256 / 480 pixels) and K7 alone: the tuned transforms under the FFT engine's
// large-patch pipeline; flags as launch_pcl_rows
hipError_t launch_sr_rows_real_src(const PclSrc& src, const float* twiddles, float* zh, size_t zh_stride, int* flags, int res, int n_images,
                                   int channels, int n, hipStream_t stream, int* sums = nullptr);  // n <= res: the unpadded patch size (zeros beyond n x n);
                                   // sums: 4 ints per image (zeroed), the exact sums sum (+-1)^y (+-1)^x p -- filled by the plans whose Nyquist bin is not exact (250, 400, 432)
hipError_t launch_sr_rows_inv(const float* Dt, const float* twiddles, float2* cand, int res, int n_pairs, hipStream_t stream);
// a video's per-image flags fs[frame * patches + patch] -> the per-pair layout the column kernel and the tail read: f2[2 q] = cur = fs[q + patches],
// f2[2 q + 1] = prev = fs[q] (pair q = k * patches + patch of frames k + 1, k)
hipError_t launch_pcl_seq_flags(const int* fs, int* f2, int patches, int n_pairs, hipStream_t stream);
hipError_t launch_sr_cols_seq(const float* zh_prev, const float* zh_cur, size_t zh_stride, const float* twiddles, float* Dt, int res,
                              int n_pairs, int run, hipStream_t stream, const int* flags = nullptr, int n = 0,  // flags + n < res: the box-zero rule of padded constant patches (run = 1)
                              const int* sums_prev = nullptr, const int* sums_cur = nullptr, int sums_stride = 0);  // the rows kernel's exact sums, pair p at p * sums_stride ints
// K56 (sr_fused_kernel.hip): K5s + K6s in one kernel, the row transforms as a dense product on the matrix cores -- reads the u8
// log-polar images instead of Zh. `frags` = sr_fused_fragments(res) on the device. Same pair / run semantics as K6s.
bool sr_fused_supported(int res);
std::vector<uint32_t> sr_fused_fragments(int res);
hipError_t launch_sr_cols_fused(const uint8_t* lp_prev, const uint8_t* lp_cur, size_t lp_stride, const uint32_t* frags,
                                const float* twiddles, float* Dt, int res, int n_pairs, int run, hipStream_t stream);

}  // namespace mof
