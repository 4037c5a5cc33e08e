// capi_graph.hpp -- what the C ABI's engines need to be safe next to HIP graphs (host side only).
//
// Two hazards, both seen on the GPU box in round 2 (gpurun_out/r02r_c5graph.err, r02s_c5graph.err):
//  1. A batch call captured into a graph bakes raw pointers to engine-owned device memory (twiddles, the estimator's
//     scratch) into the graph's kernel nodes. Freeing or re-allocating that memory under the graph makes every later
//     replay read and write freed memory. An engine therefore gets PINNED by a captured call: while pinned its scratch
//     never moves (growth fails with MOF_ERR_BUSY) and mof_*_destroy does not free -- the engine is parked on a
//     process-wide list until the owner says the graphs are gone (mof_*_release_graphs / mof_purge_deferred).
//  2. hipFree / hipMalloc on ANY thread while some stream captures in HIP's default global mode is answered by
//     invalidating that capture. The library's own allocation and release paths run under the relaxed mode of the
//     calling thread (RelaxedCapture), the documented way for a library to stay out of other people's captures.
#pragma once

#include <hip/hip_runtime.h>

namespace mof {

struct RelaxedCapture {
  hipStreamCaptureMode mode = hipStreamCaptureModeRelaxed;
  RelaxedCapture() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
  ~RelaxedCapture() { (void)hipThreadExchangeStreamCaptureMode(&mode); }
  RelaxedCapture(const RelaxedCapture&) = delete;
  RelaxedCapture& operator=(const RelaxedCapture&) = delete;
};

inline bool stream_capturing(hipStream_t s) {
  hipStreamCaptureStatus st = hipStreamCaptureStatusNone;
  return hipStreamIsCapturing(s, &st) == hipSuccess && st != hipStreamCaptureStatusNone;
}

// Synchronous copy / fill on a stream of the engine's own (non-blocking): hipMemcpy / hipMemset run on the legacy stream,
// which implicitly joins every blocking stream -- a capturing one included ("operation would make the legacy stream depend
// on a capturing blocking stream"), so they cannot be used by a library that may be called beside a capture.
inline hipError_t copy_on(hipStream_t s, void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  const hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, s);
  return e != hipSuccess ? e : hipStreamSynchronize(s);
}
inline hipError_t fill_on(hipStream_t s, void* dst, int value, size_t bytes) {
  const hipError_t e = hipMemsetAsync(dst, value, bytes, s);
  return e != hipSuccess ? e : hipStreamSynchronize(s);
}

// engines whose destroy was deferred because a captured graph may still use their device memory
void park_engine(void (*destroy_now)(void*), void* engine);  // mof_capi.hip
int purge_parked();                                          // frees every parked engine, returns how many
int parked_count();

}  // namespace mof
