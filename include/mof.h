/*
 * mof.h -- C ABI of the MI355X optic-flow core (libmof_hip.so).
 *
 * Drop-in boundary for the processor surface of ctu-mrs/mrs_optic_flow:
 *   OpticFlowCalc::processImage / setImPrev      /root/reference/include/OpticFlowCalc.h:9-16
 *   FftMethod::FftMethod / processImage          /root/reference/include/FftMethod.h:434-439,
 *                                                /root/reference/src/FftMethod.cpp:1680-1766, :1772-1903
 *   BlockMethod::BlockMethod / processImage      /root/reference/src/BlockMethod.cpp:3-22, :25-94
 *   FastSpacedBMMethod::processImage             /root/reference/src/FastSpacedBMMethod_OCL.cpp:71-184
 *   scaleRotationEstimator::processImage         /root/reference/src/scaleRotationEstimator.cpp:34-148
 *
 * Plain pointers and sizes only; no C++/torch/OpenCV types; no exceptions cross
 * this boundary. Every entry point returns a status (0 ok, <0 error) and the text
 * of the last error of the calling thread is available from mof_last_error().
 * There is NO CPU fallback: when no HIP device is usable every create() fails
 * with MOF_ERR_NO_DEVICE.
 *
 * Conventions
 *   - frames are 8-bit single-channel, row-major, `pitch` bytes per row;
 *   - FFT results are (x, y) pixel shifts, +x right, +y down, positive = image
 *     content moved that way from the previous to the current frame (the sign of
 *     `-cv::phaseCorrelate(cur, prev)`, FftMethod.cpp:1836); invalid patches are
 *     (NaN, NaN) (FftMethod.cpp:1851-1853); patch (i, j) is stored at index
 *     i + j*grid_x (FftMethod.cpp:1855);
 *   - block-matching results are integer (dx, dy) = arg-min position minus the
 *     scan radius (BlockMethod.cpp:63-66; FastSpacedBMMethod.cl:74-75), block
 *     (bx, by) at index by*grid_x + bx, plus the per-axis histogram mode;
 *   - one engine = one HIP stream, not re-entrant: a call made while another is
 *     in flight on the same engine returns MOF_ERR_BUSY (the reference returns an
 *     empty vector, FftMethod.cpp:1775-1776). Different engines are independent.
 */
#ifndef MOF_H
#define MOF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MOF_OK 0
#define MOF_ERR_BAD_ARG (-1)
#define MOF_ERR_BUSY (-2)
#define MOF_ERR_HIP (-3)
#define MOF_ERR_NOT_INIT (-4)
#define MOF_ERR_UNSUPPORTED (-5)
#define MOF_ERR_NO_DEVICE (-6)
#define MOF_ERR_NO_MEMORY (-7)

const char* mof_version(void);
/* Message of the last failing call made by this thread ("" if none). */
const char* mof_last_error(void);
/* Number of usable HIP devices (0 when there is none; never negative). */
int mof_device_count(void);

/* HIP graphs. Every *_batch_device entry point can be captured into a HIP graph on the caller's stream (no allocation,
 * synchronisation or host read-back inside). A captured kernel node holds raw pointers to engine-owned device memory
 * (FftMethod: the twiddle table; the scale/rotation estimator: its whole pipeline scratch), so a call made while `stream`
 * is capturing PINS its engine:
 *   - while pinned, the estimator's scratch never moves: a later batch that would need more pairs per pass than the
 *     scratch holds fails with MOF_ERR_BUSY instead of re-allocating under the graph (mof_sr_reserve the largest batch
 *     BEFORE capturing);
 *   - mof_*_destroy of a pinned engine does not free anything: the engine is parked on a process-wide list, and replays of
 *     the graph stay valid. mof_purge_deferred() frees the parked engines (call it when the graphs are destroyed);
 *   - mof_*_release_graphs(e) un-pins a live engine: the caller states that no graph that captured it will be replayed.
 * The library's own allocations and frees (create, destroy, reserve, the *_batch_host temporaries) run under
 * hipStreamCaptureModeRelaxed on the calling thread, so creating or destroying an engine on one thread does not
 * invalidate a capture that is running on another (or on the same) thread. Block-matching batches read no engine memory
 * and do not pin. There is no reference counterpart: the reference is one synchronous call per frame. */
int mof_purge_deferred(void);  /* frees every parked engine; returns how many */
int mof_deferred_count(void);  /* engines parked now */

/* ------------------------------------------------------------------------------------------ */
/* FFT phase correlation (FftMethod)                                                          */
/* ------------------------------------------------------------------------------------------ */

typedef struct mof_fft_config {
  int frame_width, frame_height; /* pixels                                                    */
  int patch_size;                /* samplePointSize N, any >= 2 up to 960 after padding (see below) */
  int grid_x, grid_y;            /* patches per row / per column                              */
  int origin_x, origin_y;        /* top-left pixel of patch (0,0)                             */
  int stride_x, stride_y;        /* distance between patch origins                            */
  double max_px_speed;           /* gate on |shift| (FftMethod.cpp:1686, :1841)               */
  int device;                    /* HIP device ordinal                                        */
  int peak_model;                /* MOF_PEAK_OPENCV (0, useOCL=false) or MOF_PEAK_OCL (useOCL=true) */
  int search_radius;             /* MOF_PEAK_OCL only: SEARCH_RADIUS, 55 (FftMethod.cpp:820)  */
} mof_fft_config;

/* Which of the reference's two peak models turns a patch pair into a shift:
 *   MOF_PEAK_OPENCV  the live path, shift = -cv::phaseCorrelate(cur, prev) (FftMethod.cpp:1836): P|P|/(|P|^2+eps)
 *                    normalisation, 5x5 centroid over every value, in double.
 *   MOF_PEAK_OCL     what FftMethod computes with useOCL=true (FftMethod.cpp:1824-1825, :1833 -> cl/FftMethod.cl):
 *                    P rsqrt(|P|^2 + eps) normalisation (cl:971-982; 1/(a b) in the four real-only slots, :1029),
 *                    inverse scaled by 1/N^2, rows and columns with search_radius < index < N - search_radius of the
 *                    un-shifted surface zeroed (cl:737-746, :823-826), first maximum, 7x7 centroid over the values > 0
 *                    with the sum seeded by FLT_EPSILON (cl:1315-1379). The OpenCL kernel orders its work-groups with
 *                    barrier(), which OpenCL does not guarantee, so its literal output is not defined; this is its
 *                    race-free reading. Its centroid sums floats over absolute frame coordinates (~1e-4 px of rounding
 *                    noise); the engine sums patch-local doubles.
 * The gate (FftMethod.cpp:1838-1856) is the same for both. */
enum { MOF_PEAK_OPENCV = 0, MOF_PEAK_OCL = 1 };

/* Geometry exactly as FftMethod's constructor derives it (FftMethod.cpp:1706-1720):
 * frameSize forced even; if it is not a multiple of samplePointSize the patch becomes
 * the whole frame; sqNum = frameSize / samplePointSize; contiguous tiling from (0,0). */
int mof_fft_config_reference(mof_fft_config* cfg, int frame_size, int sample_point_size, double max_px_speed);

typedef struct mof_fft_engine mof_fft_engine;

/* Patch sizes (FftMethod.cpp:1680-1720 takes frameSize / samplePointSize from ROS parameters, config/default.yaml:31-32, and
 * falls back to ONE patch = the whole frame when they do not divide): every patch goes through cv::phaseCorrelate (:1836),
 * which zero-pads it to M = cv::getOptimalDFTSize(N), the smallest 2^a 3^b 5^c >= N (possibly odd: 74 -> 75); peak, centroid and
 * centre (M / 2.0) live on the padded image, the gate compares with N / 2 (:1841-1842). Four kernel families serve this
 * (mof_fft_kernel_variant): hand-tuned packed-pair instantiations for N = 32, 64, 128 -- and 120 for the long-range mode and the OpenCL
 * peak model -- ("stockham"); the fused kernel on a HALF-size tile ("planned-half", csrc/pc_half_kernel.hip, r05: each image
 * transformed on its own, the previous spectrum waiting in registers) for N = 120 -- the reference's default, two workgroups per CU --,
 * for every N whose padded size is an even 136 .. 192, and for padded sizes 60 / 72 / 90 / 96 / 100 / 120; a planned kernel on the full M x M tile for
 * every other N with M <= 135 ("planned", csrc/pc_kernel_generic.hip); a planned pipeline through HBM scratch for larger patches up to
 * M = 960 ("planned-large", csrc/pc_large_kernel.hip; r06: at every M from 200 to 960, even or odd, the transforms inside it are the
 * scale / rotation estimator's tuned in-register ones (csrc/sr_seq_kernel.hip: the row kernel zero-pads, 250 / 400 / 432 take their
 * real-only spectrum slots from exact integer pixel sums); the planned transforms remain for the long-range mode and as the A/B form). mof_fft_create fails with MOF_ERR_UNSUPPORTED
 * only beyond that, and for MOF_PEAK_OCL on sizes the reference's OpenCL branch cannot plan either (odd, not 5-smooth) or M > 135.
 * The large-patch pipeline owns scratch (three half-spectrum planes per patch pair of a pass); it grows with the first batch
 * that needs more -- never inside a HIP graph capture: run the largest batch once before capturing. */
int mof_fft_create(const mof_fft_config* cfg, mof_fft_engine** out);
/* Diagnostics: name of the kernel formulation the engine launches for cv::phaseCorrelate-model batches on full-resolution frames:
 * "stockham" (pc_kernel.hip / pc_kernel_mixed.hip), "planned-half", "planned" or "planned-large" (above) in the product library;
 * "quad" only in the A/B build csrc/ab/libmof_hip_quad.so with MOF_PC_QUAD=1 (pc_kernel_quad.hip, a measured-slower alternative kept
 * for comparison, not shipped). MOF_FFT_HALF=0 / 1 (environment, read once) keeps every size off / forces the instantiated sizes
 * onto the half-tile kernel (A/B and the tests of either family). */
const char* mof_fft_kernel_variant(const mof_fft_engine* e);
void mof_fft_destroy(mof_fft_engine* e);
int mof_fft_release_graphs(mof_fft_engine* e);      /* see "HIP graphs" above */
int mof_fft_graph_pinned(const mof_fft_engine* e);  /* 1 while pinned */

/* setImPrev (OpticFlowCalc.h:14-16): host frame copied to the device-resident previous frame.
 * As in the reference, the first process() call after create()/reset() still correlates the
 * frame with itself (`first`, FftMethod.cpp:1791-1793). */
int mof_fft_set_prev(mof_fft_engine* e, const uint8_t* frame, size_t pitch);
/* Re-arms `first` (FftMethod.cpp:1761). */
int mof_fft_reset(mof_fft_engine* e);

/* processImage (FftMethod.cpp:1772-1903), synchronous: uploads `frame`, correlates it with the
 * previous frame, writes 2*grid_x*grid_y doubles to out_xy, then previous <- current
 * (FftMethod.cpp:1872). n_invalid (optional) counts NaN patches. */
int mof_fft_process(mof_fft_engine* e, const uint8_t* frame, size_t pitch, double* out_xy, int* n_invalid);

/* processImageLongRange (FftMethod.cpp:1905-2007; LONG_RANGE_RATIO 4, :3): both frames are reduced to a quarter
 * exactly as cv::resize(.., 1/4, 1/4) does for CV_8UC1 (rounded mean of the 2x2 centre of each 4x4 cell) -- on the
 * fly inside the kernel -- and the same per-patch correlation runs on the (grid_x/4) x (grid_y/4) grid of patches of
 * the SAME size (samplePointSize_lr, :1685; sqNum_lr, :1720). Shares `first` and the previous frame with
 * mof_fft_process. Only for the reference tiling with frame sides divisible by 4 and sqNum >= 4
 * (else MOF_ERR_UNSUPPORTED). out_xy: 2 * mof_fft_long_range_patches(e) doubles, in quarter-resolution pixels (the
 * node multiplies by 4, optic_flow.cpp:472-476). */
int mof_fft_long_range_patches(const mof_fft_engine* e);
int mof_fft_process_long_range(mof_fft_engine* e, const uint8_t* frame, size_t pitch, double* out_xy, int* n_invalid);
int mof_fft_process_long_range_batch_device(mof_fft_engine* e, const uint8_t* d_cur, size_t cur_stride,
                                            const uint8_t* d_prev, size_t prev_stride, size_t pitch, int n_pairs,
                                            double* d_out_xy, void* stream);

/* Batched mode on DEVICE pointers: pair k correlates d_cur + k*cur_stride with
 * d_prev + k*prev_stride (strides in bytes; a video sequence is cur = frames + frame_bytes,
 * prev = frames, both strides frame_bytes). d_out_xy receives n_pairs*grid_x*grid_y*2 doubles.
 * Asynchronous on `stream` (a hipStream_t; NULL is HIP's null stream, as everywhere in HIP);
 * the caller synchronises. The engine's stateful previous frame is not touched. */
int mof_fft_process_batch_device(mof_fft_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                 size_t prev_stride, size_t pitch, int n_pairs, double* d_out_xy, void* stream);
/* A VIDEO on the device: frame i at d_frames + i*frame_stride; pair k = (cur: frame k + 1, prev: frame k), i.e. what
 * n_frames - 1 consecutive processImage calls return after the first one (`imPrev = imCurr.clone()`, FftMethod.cpp:1872).
 * d_out_xy receives (n_frames - 1) * grid_x * grid_y * 2 doubles. Same estimator as mof_fft_process_batch_device on
 * (d_frames + frame_stride, d_frames): same arg-max, sub-pixel shifts equal within rounding. For 64 x 64 and 128 x 128
 * patches a workgroup owns one patch position and walks consecutive frames, keeping the previous frame's half spectrum
 * in registers: one real 2-D transform forward and one Hermitian inverse per frame and patch -- 1.0 instead of the
 * pair kernel's 1.5 complex-transform units (csrc/pc_seq_kernel.hip; csrc/pc_seq_half.hip for 128 x 128, on a half-size
 * LDS tile); r05: the sizes the half-tile kernel serves (120 x 120 -- the reference's default --, patches padded to 60 / 72 / 90 / 96 /
 * 100 / 120, and 136 .. 192 but 162; on a video also 50 / 54 / 108, whose pair form stays the planned kernel) run its video form the same way (csrc/pc_half_kernel.hip, SEQ: the spectrum stays in the registers of
 * the forward column pass; identical bits to the pair entry on the two views of the video; MOF_FFT_HALF_SEQ=0 keeps the pair form);
 * every other size runs the pair kernel on the two views. The engine's stateful previous frame is not touched. Asynchronous on
 * `stream`. */
int mof_fft_process_sequence_device(mof_fft_engine* e, const uint8_t* d_frames, size_t frame_stride, size_t pitch,
                                    int n_frames, double* d_out_xy, void* stream);
/* The same on interleaved BGR8 frames (the node's camera frames, optic_flow.cpp:1465): CV_RGB2GRAY as the node applies it
 * (:1622) happens inside the kernels' loads, `pitch` in BYTES (3 per pixel), d_frames at the first byte of the crop.
 * Identical bits to the gray entry on the converted frames. */
int mof_fft_process_sequence_device_bgr(mof_fft_engine* e, const uint8_t* d_frames, size_t frame_stride, size_t pitch,
                                        int n_frames, double* d_out_xy, void* stream);
/* Front-end fusion: the frames are interleaved BGR8 as the node receives them
 * (cv_bridge::toCvCopy(msg, BGR8), optic_flow.cpp:1465) and cv::cvtColor(crop, gray, CV_RGB2GRAY) (:1622) -- applied to
 * BGR data, i.e. gray = (B*4899 + G*9617 + R*1868 + 8192) >> 14 -- happens inside the kernel's load. d_cur / d_prev
 * point at the first byte of the frame_width x frame_height crop (the cropping rectangle of :1609-1616), pitch is
 * in BYTES (3 per pixel). Otherwise as mof_fft_process_batch_device. */
int mof_fft_process_batch_device_bgr(mof_fft_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                     size_t prev_stride, size_t pitch, int n_pairs, double* d_out_xy, void* stream);
/* Same on HOST pointers, synchronous: what a caller that holds its frames in host memory (a replayed camera log, the node's cv::Mat
 * frames of optic_flow.cpp:1465) calls. r06: a three-slot pipeline (csrc/host_pipe.hpp) -- chunks of about 16 MB of frames are uploaded on
 * a copy stream of the engine's own while the previous chunk runs through mof_fft_process_batch_device on the engine's stream (so the
 * results are that entry's bits) and the calling thread packs the next one:
 *   - frames in PINNED memory (mof_host_alloc / mof_host_register below, or the caller's own hipHostMalloc) are DMA'd from where they lie;
 *     pageable frames are packed into pinned staging first (MOF_HOST_THREADS helpers, default 4);
 *   - a VIDEO -- cur == prev + prev_stride and cur_stride == prev_stride, i.e. pair k = (frame k + 1, frame k) of one run of frames, what
 *     consecutive processImage calls see (FftMethod.cpp:1872) -- is uploaded once per frame (MOF_HOST_VIDEO=0: as two batches).
 * PCIe-bound either way (profiles/r06_host_entries.txt); never what bench.py's `value` reports. The pipeline's buffers (three slots of two
 * chunks of frames on the device, the same again in pinned host memory once a pageable caller shows up: about 100 MB each) are made by the
 * first call and live until mof_fft_destroy. Calls on one engine are serialised. */
int mof_fft_process_batch_host(mof_fft_engine* e, const uint8_t* cur, size_t cur_stride, const uint8_t* prev,
                               size_t prev_stride, size_t pitch, int n_pairs, double* out_xy);

/* Pinned (page-locked) host memory for callers that do not link HIP themselves: hipHostMalloc / hipHostFree / hipHostRegister /
 * hipHostUnregister behind the C ABI. Frames handed to the *_batch_host entries from such memory skip the staging copy. */
int mof_host_alloc(size_t bytes, void** out);
int mof_host_free(void* p);
int mof_host_register(void* p, size_t bytes);
int mof_host_unregister(void* p);

/* Waits for the engine's own stream. */
int mof_fft_sync(mof_fft_engine* e);

/* ------------------------------------------------------------------------------------------ */
/* Batched-frames mode across the GPUs of one node (BASELINE north_star)  */
/* ------------------------------------------------------------------------------------------ */
/* Frame pairs are independent: a batch of B pairs is cut into contiguous shards of ceil(B / G) pairs, shard g is owned and
 * processed by device g (no data-path collective), and ONE all-gather of the per-rank result slabs (RCCL over xGMI) hands every
 * device the whole result. Native and single-process: one engine and one HIP stream per device inside the group,
 * ncclCommInitAll (mof_shard_*_init_gather, explicit) + one in-place ncclAllGather per batch; RCCL is bound at run time by
 * init_gather (dlopen of librccl.so.1), so single-GPU hosts never load it. No reference counterpart (the reference is one
 * synchronous call per frame on one device); per device the work is exactly mof_fft_process_batch_device /
 * mof_bm_process_batch_device. */

/* The partition: shard `shard` of `n_shards` owns pairs [*first, *first + *count), slabs of ceil(n_pairs / n_shards) pairs,
 * the last one ragged (or empty). Pure host arithmetic, no device needed. */
int mof_shard_slab_pairs(int n_pairs, int n_shards); /* ceil(n_pairs / n_shards) */
int mof_shard_partition(int n_pairs, int n_shards, int shard, int* first, int* count);

typedef struct mof_shard_fft mof_shard_fft;
/* One engine of `cfg` per listed device (cfg->device is ignored; devices == NULL: 0 .. n_devices - 1). A device may be listed
 * once -- except under the rehearsal knob MOF_SHARD_SHARE_DEVICE=1 (environment; tests only), which admits several shards on one
 * device so that the G > 1 slab arithmetic can run on a one-GPU box; such a group cannot gather (MOF_ERR_UNSUPPORTED).
 * No entry point of the group changes the calling thread's current HIP device. */
int mof_shard_fft_create(const mof_fft_config* cfg, const int* devices, int n_devices, mof_shard_fft** out);
void mof_shard_fft_destroy(mof_shard_fft* g);
int mof_shard_fft_devices(const mof_shard_fft* g);
void* mof_shard_fft_stream(const mof_shard_fft* g, int shard); /* the hipStream_t shard's work is enqueued on */
/* The gather's set-up, explicit and BLOCKING (~0.1 s): binds RCCL (dlopen of librccl.so.1) and builds one communicator per device
 * (ncclCommInitAll). Call it once before the first batch with gather != 0; a batch that asks for the gather without it fails
 * with MOF_ERR_NOT_INIT -- the asynchronous batch call never builds communicators itself. Idempotent. */
int mof_shard_fft_init_gather(mof_shard_fft* g);
int mof_shard_fft_gather_ready(const mof_shard_fft* g); /* 1 once init_gather succeeded */
int mof_shard_fft_gather_ranks(const mof_shard_fft* g); /* ranks RCCL formed: ncclCommCount of communicator 0; 0 before init_gather */
/* d_cur[g] / d_prev[g]: device g's OWN shard of the batch (its first pair at the pointer; strides and pitch as in
 * mof_fft_process_batch_device) -- frames are generated or loaded directly on the owning GPU. d_out[g]: on device g,
 * n_shards * mof_shard_slab_pairs(n_pairs, n_shards) * grid_x * grid_y * 2 doubles; pair k's vectors land at pair index k
 * (slabs are contiguous; the padding behind a ragged last shard is not written). gather != 0: after the all-gather every
 * d_out[g] holds all n_pairs results; gather == 0: each device holds only its own slab (at its place). Asynchronous: enqueued on
 * the group's per-device streams; mof_shard_fft_sync waits for all of them. Arguments are checked for every shard before
 * anything is launched; if a later shard's launch fails, the shards already launched are waited for before the error returns. */
int mof_shard_fft_process_batch_device(mof_shard_fft* g, const uint8_t* const* d_cur, size_t cur_stride,
                                       const uint8_t* const* d_prev, size_t prev_stride, size_t pitch, int n_pairs,
                                       double* const* d_out, int gather);
int mof_shard_fft_sync(mof_shard_fft* g);

/* ------------------------------------------------------------------------------------------ */
/* SAD block matching (BlockMethod / FastSpacedBMMethod)                                      */
/* ------------------------------------------------------------------------------------------ */

typedef struct mof_bm_config {
  int frame_width, frame_height;
  int block_size;        /* samplePointSize                                                   */
  int step_size;         /* stepSize: gap between blocks (0 = BlockMethod's contiguous blocks) */
  int scan_radius;       /* scanRadius r; candidates are [-r, r]^2                            */
  int grid_x, grid_y;    /* blocks per row / column                                           */
  int low_contrast_rule; /* 1: (SAD(0,0) - min) <= 0.2 r^2 -> (0,0)  (FastSpacedBMMethod.cl:77-82) */
  int device;
} mof_bm_config;

/* BlockMethod geometry (BlockMethod.cpp:11, :45): grid = (frameSize - 2r)/sps squared, no rule. */
int mof_bm_config_block_method(mof_bm_config* cfg, int frame_size, int sample_point_size, int scan_radius);
/* FastSpacedBM geometry (FastSpacedBMMethod_OCL.cpp:82-90): S = sps + step,
 * grid = ((W - 2r)/S, (H - 2r)/S), rule on. */
int mof_bm_config_fast_spaced(mof_bm_config* cfg, int width, int height, int sample_point_size, int step_size,
                              int scan_radius);

typedef struct mof_bm_engine mof_bm_engine;

int mof_bm_create(const mof_bm_config* cfg, mof_bm_engine** out);
void mof_bm_destroy(mof_bm_engine* e);
int mof_bm_release_graphs(mof_bm_engine* e);      /* block-matching batches never pin; kept for symmetry */
int mof_bm_graph_pinned(const mof_bm_engine* e);
int mof_bm_set_prev(mof_bm_engine* e, const uint8_t* frame, size_t pitch);
int mof_bm_reset(mof_bm_engine* e);

/* processImage of either class, synchronous. dx, dy: grid_x*grid_y int8 each. mode_xy[2]:
 * per-axis histogram mode (BlockMethod.cpp:75-76; Histogram_C1_D0 element 0). Unlike
 * FftMethod the block matchers do NOT self-correlate the first frame: the previous frame
 * starts as zeros (BlockMethod.cpp:17-18) unless set_prev was called. */
int mof_bm_process(mof_bm_engine* e, const uint8_t* frame, size_t pitch, int8_t* dx, int8_t* dy, int8_t* mode_xy);

/* BlockMethod::Refine(imCurr, imPrev, fullpixFlow, passes) (BlockMethod.cpp:96-147) on the two frames of the LAST
 * mof_bm_process call (the reference calls it with passes = 2 on the histogram mode, :79). faithful != 0 reproduces
 * the reference literally -- the "previous" 2x image is resized from the CURRENT one (:110) -- while
 * faithful == 0 resizes it from the previous frame; everything else (cv::resize x2 in OpenCV's 8-bit fixed point, the
 * cut-out rectangles of :113-123, nine SADs, first minimum, offset / 2^passes) is the same. Synchronous. */
int mof_bm_refine(mof_bm_engine* e, int fullpix_x, int fullpix_y, int passes, int faithful, double* out_xy);

/* Batched mode on DEVICE pointers. d_dx, d_dy: n_pairs*grid_x*grid_y int8; d_mode: n_pairs*8 int8
 * = {modeX, modeY, 2nd X, 2nd Y, 3rd X, 3rd Y, 0, 0} (the TestDepth=3 list of
 * FastSpacedBMMethod_OCL.cpp:97 per axis). Asynchronous on `stream`. */
int mof_bm_process_batch_device(mof_bm_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                size_t prev_stride, size_t pitch, int n_pairs, int8_t* d_dx, int8_t* d_dy,
                                int8_t* d_mode, void* stream);
/* Front-end fusion for the block matchers, as mof_fft_process_batch_device_bgr: the frames are
 * interleaved BGR8 (3 bytes per pixel, `pitch` bytes per row >= 3 * frame_width), cv::cvtColor(.., CV_RGB2GRAY) as the
 * node applies it (optic_flow.cpp:1622, on BGR data) happens inside the kernels' staging loads. Same results, to the
 * bit, as the gray entry on the converted frames. */
int mof_bm_process_batch_device_bgr(mof_bm_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                    size_t prev_stride, size_t pitch, int n_pairs, int8_t* d_dx, int8_t* d_dy,
                                    int8_t* d_mode, void* stream);
/* On HOST pointers, synchronous: the same upload / run / download pipeline as mof_fft_process_batch_host (pinned frames DMA'd in place,
 * a video uploaded once per frame), chunks through mof_bm_process_batch_device. */
int mof_bm_process_batch_host(mof_bm_engine* e, const uint8_t* cur, size_t cur_stride, const uint8_t* prev,
                              size_t prev_stride, size_t pitch, int n_pairs, int8_t* dx, int8_t* dy, int8_t* mode);
int mof_bm_sync(mof_bm_engine* e);

/* The same group for the block matchers (BlockMethod / FastSpacedBMMethod): shard g runs mof_bm_process_batch_device on its
 * pairs, and the per-block shifts AND the per-pair histogram modes -- the pair FastSpacedBMMethod returns,
 * /root/reference/src/FastSpacedBMMethod_OCL.cpp:172-175 -- ride in ONE slab per rank ("BM mode vectors ride in the same slab",
 * SURVEY section 8(e)), so one all-gather moves everything. A rank's slab, with sp = mof_shard_slab_pairs(n_pairs, n_shards) and
 * blocks = grid_x * grid_y, is three planes of int8:
 *     dx[sp][blocks] | dy[sp][blocks] | mode[sp][8]          (mode as in mof_bm_process_batch_device)
 * padded to a multiple of 16 bytes = mof_shard_bm_slab_bytes(); d_out[g] holds n_shards such slabs, rank i's at byte i * slab.
 * mof_shard_bm_locate gives the byte offsets of pair k's dx / dy / mode inside such a buffer. Everything else as the FFT group. */
typedef struct mof_shard_bm mof_shard_bm;
int mof_shard_bm_create(const mof_bm_config* cfg, const int* devices, int n_devices, mof_shard_bm** out);
void mof_shard_bm_destroy(mof_shard_bm* g);
int mof_shard_bm_devices(const mof_shard_bm* g);
void* mof_shard_bm_stream(const mof_shard_bm* g, int shard);
int mof_shard_bm_init_gather(mof_shard_bm* g);
int mof_shard_bm_gather_ready(const mof_shard_bm* g);
int mof_shard_bm_gather_ranks(const mof_shard_bm* g);
size_t mof_shard_bm_slab_bytes(const mof_shard_bm* g, int n_pairs); /* bytes of one rank's slab (0 for a null group) */
int mof_shard_bm_locate(const mof_shard_bm* g, int n_pairs, int pair, size_t* dx_off, size_t* dy_off, size_t* mode_off);
int mof_shard_bm_process_batch_device(mof_shard_bm* g, const uint8_t* const* d_cur, size_t cur_stride,
                                      const uint8_t* const* d_prev, size_t prev_stride, size_t pitch, int n_pairs,
                                      int8_t* const* d_out, int gather);
int mof_shard_bm_sync(mof_shard_bm* g);

/* ------------------------------------------------------------------------------------------ */
/* Scale / rotation estimator (scaleRotationEstimator, BASELINE config c5)                    */
/* ------------------------------------------------------------------------------------------ */

typedef struct mof_sr_config {
  int resolution;   /* side of the square image (scaleRotationEstimator.cpp:5): any even value >= 16 that pads to <= 960. 240, 256
                       and 480 have the full set of hand-tuned kernels; r06: every resolution whose padded size is an even
                       91 .. 960 but a few below 200 runs the tuned transforms too (MOF_SR_TUNED_ALL=0: those three
                       only); the rest run the planned pipeline on the padded size                           */
  double magnitude; /* log-polar magnitude M (scale_rot_magnitude, config/default.yaml:13: 49.9)      */
  int device;
  int logpolar_variant; /* MOF_LOGPOLAR_CV4 (0, default) or MOF_LOGPOLAR_CV3: see below              */
  /* Batched mode only (no reference counterpart; zero-initialise for the defaults):                 */
  int batch_chunk;      /* frame pairs per pipeline pass, 0 = default (512; 1..4096). From its first       */
                        /* batch on the engine owns scratch for min(batch size rounded up to a power of   */
                        /* two, batch_chunk) pairs at 4 res^2 + 8 res^2 + 8 res (res/2+1) bytes each      */
                        /* (1.9 GB at 480^2 and 512 pairs; 1024-pair passes measure +1.7 % at c5 for 3.8   */
                        /* GB); until then one pair's worth                                                */
  int pipeline_lanes;   /* 0 = default (1), 1 = every pass on the caller's stream, 2 = the remap of pass  */
                        /* k+1 runs beside the transforms of pass k on a second stream of the engine      */
} mof_sr_config;

/* The reference calls cv::logPolar under ROS Noetic (OpenCV 4.2) and the C API cvLogPolar under ROS Melodic
 * (OpenCV 3.2) (scaleRotationEstimator.cpp:41-46, :107-113). Their maps differ: 4.x goes through cv::warpPolar,
 * radius exp(rho * Kmag) - 1; 3.2 uses radius exp(rho / M) without the "- 1". Both are available; everything after
 * the map (remap's fixed point, the phase correlation) is the same. */
enum { MOF_LOGPOLAR_CV4 = 0, MOF_LOGPOLAR_CV3 = 1 };

typedef struct mof_sr_engine mof_sr_engine;

/* scaleRotationEstimator::scaleRotationEstimator (scaleRotationEstimator.cpp:3-32). */
int mof_sr_create(const mof_sr_config* cfg, mof_sr_engine** out);
void mof_sr_destroy(mof_sr_engine* e);
/* Re-arms `first` and clears tempIm (scaleRotationEstimator.cpp:27, :31). */
int mof_sr_reset(mof_sr_engine* e);
/* Batched mode: grow the engine's scratch to min(n_pairs, batch_chunk) pairs per pass now instead of inside the first
 * batch. Needed before a batch is CAPTURED into a HIP graph on an engine that has not run a batch yet (allocation is
 * not capturable: the batch call then fails with MOF_ERR_BAD_ARG). No reference counterpart. */
int mof_sr_reserve(mof_sr_engine* e, int n_pairs);
int mof_sr_release_graphs(mof_sr_engine* e);      /* see "HIP graphs" at the top of this header */
int mof_sr_graph_pinned(const mof_sr_engine* e);

/* scaleRotationEstimator::processImage (scaleRotationEstimator.cpp:34-148), synchronous. frame: resolution^2
 * CV_8UC1. out_scale_rot[2] = (scale, rotation in rad): first call -> log-polar (INTER_CUBIC) kept as the previous
 * image, returns (1, 0) (:36-74); later calls -> log-polar (INTER_LANCZOS4, :112), pt = cv::phaseCorrelate(current,
 * previous) (:117), |pt.x| > resolution/2 -> (1, 0) without updating the previous image (:119-121), else
 * scale = exp(pt.x / M), rot = (pt.y / (resolution/360)) * pi/180 (:123-124) and previous <- current (:128). */
int mof_sr_process(mof_sr_engine* e, const uint8_t* frame, size_t pitch, double* out_scale_rot);

/* Batched mode on DEVICE pointers; each pair (prev, cur) is processed as the two-call sequence of a fresh estimator --
 * by the same kernels as mof_sr_process, so pair k's (scale, rot) are bit-identical to a fresh engine's stateful calls.
 * d_cur / d_prev point at the top-left pixel of the resolution^2 crop inside each frame (pitch bytes per row).
 * d_out receives n_pairs * 4 doubles: scale, rot, pt.x, pt.y. Asynchronous on `stream`.
 * The pipeline runs through scratch owned by the engine. Calls on the same stream are ordered by the stream; a call
 * on a different stream than the previous one first makes its stream wait (hipStreamWaitEvent) for the previous
 * call's last kernel, so back-to-back batches on different streams are safe but do not overlap. While `stream` is
 * being captured into a HIP graph no cross-stream dependency is taken or left: replays of graphs that contain calls
 * on one engine must be ordered by the caller.
 * A batch runs in pipeline passes of batch_chunk pairs (default 512) on `stream`. With pipeline_lanes = 2 (not the
 * default) the log-polar remaps of pass k+1 run on a stream of the engine's own beside the transforms of pass k, handed
 * over with events; everything is complete when `stream` is. Under graph capture the engine's stream joins the capture
 * (fork / join by events) and the graph replays with the same lanes. A captured call pins the engine ("HIP graphs" at
 * the top of this header): its scratch stays put and its destroy is deferred while a graph may replay through it. */
int mof_sr_process_batch_device(mof_sr_engine* e, const uint8_t* d_cur, size_t cur_stride, const uint8_t* d_prev,
                                size_t prev_stride, size_t pitch, int n_pairs, double* d_out, void* stream);

/* A VIDEO on the device: exactly what n_frames consecutive mof_sr_process calls would return, continuing the engine's
 * stateful sequence (`first`, the previous image) and leaving it as those calls would -- the reference's steady state
 * (scaleRotationEstimator.cpp:34-148): the very first frame after create / reset goes through INTER_CUBIC (:45) and
 * returns (1, 0) (:74); every later frame goes through INTER_LANCZOS4 (:112) ONCE and is correlated with the previous
 * frame's log-polar image (:117), which is what it then becomes itself (:128). Frame i at d_frames + i*frame_stride
 * (the top-left pixel of the resolution^2 crop, `pitch` bytes per row). d_out receives n_frames * 4 doubles:
 * scale, rot, pt.x, pt.y (first frame: 1, 0, 0, 0). Same bits as the frame-by-frame calls (the same kernels run).
 * The gate (:119-121): a frame whose |pt.x| > resolution/2 returns (1, 0) and does NOT replace the previous image, so
 * the next frame is correlated with an older one -- a serial dependency through a result.
 *   n_gated != NULL: the call resolves it (per pipeline pass it reads the pass's results back, re-runs the pairs that
 *     sit behind a gated frame against the right partner) and is SYNCHRONOUS; *n_gated = number of gated frames.
 *   n_gated == NULL: asynchronous on `stream` and capturable into a HIP graph (on an ARMED engine only: the first frame
 *     of a fresh estimator takes a one-off branch that a replay must not repeat -- MOF_ERR_BAD_ARG); every frame is
 *     correlated with its immediate predecessor. Identical unless some frame is gated, which the caller can see in d_out (|pt.x| >
 *     resolution/2 with scale == 1, rot == 0) -- a degenerate case (constant or wrapped images).
 * A call that fails half-way (a HIP error from a launch) leaves the estimator's state -- `first`, the previous image --
 * exactly as it was before the call.
 * Each frame is remapped and row-transformed once (K5s), the column pass walks consecutive pairs in time (K6s,
 * csrc/sr_seq_kernel.hip): about half the remap work and a third less spectrum traffic than n_frames - 1 independent
 * pairs through mof_sr_process_batch_device. Uses the engine's scratch like the batch entry. */
int mof_sr_process_sequence_device(mof_sr_engine* e, const uint8_t* d_frames, size_t frame_stride, size_t pitch,
                                   int n_frames, double* d_out, void* stream, int* n_gated);

/* The same video in HOST memory, synchronous: exactly the n_frames consecutive mof_sr_process calls (gate resolved, the estimator's state
 * continued and left as they would leave it), with the frames uploaded in chunks on a copy stream beside the previous chunk's kernels
 * (csrc/host_pipe.hpp; frames in pinned memory -- mof_host_alloc / mof_host_register -- are DMA'd from where they lie). out receives
 * n_frames * 4 doubles; *n_gated (may be NULL) the number of gated frames. A failure half-way leaves the state as the last completed
 * CHUNK left it (the frames before it have been consumed). */
int mof_sr_process_sequence_host(mof_sr_engine* e, const uint8_t* frames, size_t frame_stride, size_t pitch, int n_frames,
                                 double* out, int* n_gated);

/* The remap stage alone: cv::logPolar(src, dst, Point2f(res/2, res/2), M, interpolation) (scaleRotationEstimator.cpp:45
 * INTER_CUBIC, :112 INTER_LANCZOS4) on n_images res x res CV_8UC1 crops (image i at d_src + i*src_stride, `pitch` bytes
 * per row) into tightly packed res*res outputs at d_dst + i*res*res. As with cv::remap's BORDER_TRANSPARENT,
 * destination pixels whose source falls outside the image KEEP their content, so the caller initialises d_dst (the
 * estimator starts from cv::Mat::zeros, :27). Integer arithmetic; exposed so that the remap can be compared byte for
 * byte. Asynchronous on `stream`; uses no engine scratch. */
enum { MOF_INTER_CUBIC = 2, MOF_INTER_LANCZOS4 = 4 }; /* cv::INTER_CUBIC, cv::INTER_LANCZOS4 */
int mof_sr_logpolar_batch_device(mof_sr_engine* e, const uint8_t* d_src, size_t src_stride, size_t pitch, int n_images,
                                 int interpolation, uint8_t* d_dst, void* stream);

/* ------------------------------------------------------------------------------------------ */
/* Geometry tail: per-patch shifts -> camera-frame velocity (OpticFlow::getRT / get2DT)       */
/* ------------------------------------------------------------------------------------------ */
/* The consumers of processImage / processImageLongRange in the node (optic_flow.cpp:1719 -> :515-774 getRT,
 * :1780 -> :388-510 get2DT), SURVEY section 8(f) N1 / N3. getRT leans on OpenCV calib3d and tf2, which are restated
 * from their published algorithms (csrc/geom_core.hpp lists what exactly): cv::undistortPoints (5 coefficients, 5
 * iterations), cv::decomposeHomographyMat (Malis-Vargas, K = I), the tf2 quaternion members, and -- the one piece
 * that CANNOT reproduce OpenCV's numbers, because cv::RNG's sequence is not restated -- cv::findHomography(RANSAC,
 * 0.01) with the library's own documented counter-based sampler (same threshold, confidence 0.995, 2000 iterations,
 * normalised-DLT + 10 LM steps on the consensus set). On data with a clear consensus set the mask and the refined
 * homography do not depend on the sampler. No HIP device is needed for the host forms. */

typedef struct mof_geom_camera { /* camMatrix_, distCoeffs_ (optic_flow.cpp:1511-1522) */
  double fx, fy, cx, cy;
  double k1, k2, p1, p2, k3;
} mof_geom_camera;

typedef struct mof_geom_layout { /* patch (i, j) centre = origin + (i, j) * stride + patch_size / 2 (:537-538) */
  int grid_x, grid_y, origin_x, origin_y, stride_x, stride_y, patch_size;
} mof_geom_layout;

typedef struct mof_geom_rt_params { /* per frame pair */
  double height;        /* uav_height_curr (:1719)                                                          */
  double dt;            /* dur_.toSec()                                                                     */
  double ul_corner_x;   /* ulCorner.x: camMatrixLocal(0,2) -= ulCorner.x (:521-522); ulCorner.y is unused    */
  double ang_rate_q[4]; /* angular_rate_tf_ (x, y, z, w): Quaternion::setRPY of the gyro rates (:1314)        */
  double c2b_q[4];      /* transformCam2Base_ rotation (x, y, z, w) (:600-601)                               */
  double c2b_t[3];      /* its translation -- `tempTfC2B * axis` applies the WHOLE transform (:643)          */
} mof_geom_rt_params;

typedef struct mof_geom_2dt_params { /* per frame pair */
  double height;        /* uav_height_curr / (cos(imu_pitch_) cos(imu_roll_)), formed by the caller (:1780)  */
  double dt;
  double roll_rate, pitch_rate; /* imu_roll_rate_, imu_pitch_rate_ (:481-482)                               */
  double cam_yaw;       /* cam_yaw_ (:484)                                                                   */
} mof_geom_2dt_params;

/* `status` of getRT / get2DT: 0 <=> the reference returns true; the others name the early return taken. */
enum {
  MOF_GEOM_OK = 0,
  MOF_GEOM_BAD_DURATION = 1,      /* !isfinite(1/dt)                       :516-519, :393-396 */
  MOF_GEOM_TOO_FEW_POINTS = 2,    /* valid points < shifted_pts_thr        :544-547 (get2DT: none valid, :425-429) */
  MOF_GEOM_TOO_FEW_INLIERS = 3,   /* after RANSAC                          :575-578 */
  MOF_GEOM_ANGLE_TOO_LARGE = 4,   /* best solution > pi/4 from the IMU     :682-685 */
  MOF_GEOM_SINGLE_NO_MATCH = 5,   /* :725-728 */
  MOF_GEOM_SINGLE_NON_FINITE = 6, /* :744-751 */
  MOF_GEOM_UNCLASSIFIED = 7,      /* :769-771 */
  MOF_GEOM_NO_HOMOGRAPHY = 8,     /* < 4 points or no model: OpenCV would throw in decomposeHomographyMat */
  MOF_GEOM_NO_POINTS = 9          /* get2DT on an empty vector, :389-392 */
};

/* The node's tiling: sqNum = frame_size / sample_point_size patches per side, origin 0, stride = patch (:525). */
int mof_geom_layout_reference(mof_geom_layout* layout, int frame_size, int sample_point_size);

/* Stages, exposed for parity tests. cv::undistortPoints(pts, out, camMatrixLocal, distCoeffs): pixels -> normalised. */
int mof_geom_undistort_points(const mof_geom_camera* cam, double ul_corner_x, const double* pts_xy, int n, double* out_xy);
/* cv::findHomography(a, b, cv::RANSAC, 0.01, mask): H9 row-major with H[8] = 1, mask[n] of 0/1, *found = 0 when no model. */
int mof_geom_find_homography(const double* a_xy, const double* b_xy, int n, double* H9, uint8_t* mask, int* found);
/* cv::decomposeHomographyMat(H, I, R, t, n): *n_solutions = 1 or 4 (0: degenerate); R[4][9] row-major, t[4][3], normals[4][3]. */
int mof_geom_decompose_homography(const double* H9, double* R, double* t, double* normals, int* n_solutions);

/* OpticFlow::getRT (:515-774). shifts_xy: the 2*grid_x*grid_y doubles processImage returned (NaN = invalid).
 * out_rot_tran[7] = o_rot (x, y, z, w), o_tran (x, y, z); identity/zero unless *status == MOF_GEOM_OK.
 * inlier_mask (optional, grid_x*grid_y bytes, 1 = RANSAC inlier) and homography (optional, 9) are diagnostics. */
int mof_geom_get_rt(const double* shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                    const mof_geom_rt_params* params, int shifted_pts_thr, double* out_rot_tran, int* status,
                    uint8_t* inlier_mask, double* homography);
/* OpticFlow::get2DT (:388-510, LONG_RANGE_RATIO 4). shifts_xy: what processImageLongRange returned (layout = the
 * long-range grid, patch_size = sample_point_size_lr). out_tran_diff[6] = o_tran (3), o_tran_diff (3). Closed form. */
int mof_geom_get_2dt(const double* shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                     const mof_geom_2dt_params* params, double* out_tran_diff, int* status);

/* Batched forms on DEVICE pointers, asynchronous on `stream`: d_shifts_xy is the [n_pairs][grid_y*grid_x][2] output of
 * mof_fft_process_batch_device / ..._long_range_batch_device, d_params one struct per pair. d_out: [n_pairs][8] doubles
 * = getRT: rot (4), tran (3), status; get2DT: tran (3), diff (3), status, 0. One wavefront per frame pair; the 64 lanes
 * evaluate 64 RANSAC hypotheses at a time and replay the host's in-order acceptance, so the model chosen is the host's. */
int mof_geom_get_rt_batch_device(const double* d_shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                                 const mof_geom_rt_params* d_params, int n_pairs, int shifted_pts_thr, double* d_out,
                                 void* stream);
int mof_geom_get_2dt_batch_device(const double* d_shifts_xy, const mof_geom_layout* layout, const mof_geom_camera* cam,
                                  const mof_geom_2dt_params* d_params, int n_pairs, double* d_out, void* stream);

#ifdef __cplusplus
}
#endif
#endif /* MOF_H */
