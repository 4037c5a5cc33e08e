// The HOST-pointer batch entries (mof_fft_process_batch_host, mof_bm_process_batch_host) as a three-slot pipeline: while the device runs
// chunk k, the copy engine uploads chunk k + 1 on a stream of its own and the calling thread (plus a few helpers) packs chunk k + 2 into
// pinned staging. Nothing here computes: a chunk is handed to the engine's device batch entry, so the results are that entry's bits.
//   * frames in PINNED memory (hipHostMalloc / hipHostRegister, or mof_host_alloc / mof_host_register of the C ABI) are DMA'd straight from
//     where they lie (one copy per chunk when the rows are dense, 2-D copies otherwise); pageable frames go through the staging slots;
//   * a VIDEO (cur = prev + one frame, same stride -- what a replayed camera stream is) is uploaded ONCE per frame: the device entry gets
//     the two overlapping views of the uploaded run, i.e. half the PCIe bytes for the same bits.
// r05 form of these entries: pack the whole batch into std::vectors, hipMalloc, two synchronous pageable copies, run, hipFree: 4.2 k
// pairs/s at c2 (profiles/r06_host_entries.txt has both).
#pragma once
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>
#include <mutex>
#include <thread>
#include <vector>

namespace mof {

// routing only: hipMemcpyAsync is correct on any host pointer (the runtime stages what it does not know to be pinned) -- it is just slow there
inline bool host_pointer_is_pinned(const void* p) {
  hipPointerAttribute_t at{};
  if (hipPointerGetAttributes(&at, p) != hipSuccess) {
    (void)hipGetLastError();  // (an unregistered pointer is reported as an error by some runtimes: not ours to keep)
    return false;
  }
  return at.type == hipMemoryTypeHost;
}

class HostPipe {
 public:
  static constexpr int SLOTS = 3, MAX_OUTS = 3;
  struct Out {
    void* user = nullptr;       // host destination of the whole batch
    size_t bytes_per_pair = 0;
  };
  struct Chunk {  // what the engine's device entry is called with
    const uint8_t *d_cur, *d_prev;
    size_t stride;  // both views: packed frames
    int count;
    void* d_out[MAX_OUTS];
  };

  HostPipe(size_t frame_bytes, const size_t* out_bytes_per_pair, int n_outs) : fb_(frame_bytes), n_outs_(n_outs) {
    for (int i = 0; i < n_outs; ++i) out_bpp_[i] = out_bytes_per_pair[i];
    const char* c = getenv("MOF_HOST_CHUNK");  // pairs per chunk (tests: ragged chunks on small batches)
    const size_t target = (size_t)16 << 20;    // bytes of frames per upload: long enough for the DMA engine's rate, short enough to overlap
    chunk_ = c && atoi(c) > 0 ? atoi(c) : (int)std::max<size_t>(1, target / std::max<size_t>(1, fb_));
    const char* t = getenv("MOF_HOST_THREADS");
    threads_ = t && atoi(t) > 0 ? atoi(t) : 4;
    const char* v = getenv("MOF_HOST_VIDEO");
    video_on_ = !v || atoi(v) != 0;
  }
  ~HostPipe() { release(); }
  HostPipe(const HostPipe&) = delete;
  HostPipe& operator=(const HostPipe&) = delete;

  int chunk_pairs() const { return chunk_; }

  // run(chunk, stream) -> 0 or the engine's error code (already reported); returns that code, or -1 with `err` set for a HIP failure of the pipe's own
  template <class Run>
  int process(const uint8_t* cur, size_t cs, const uint8_t* prev, size_t ps, size_t pitch, int row_bytes, int rows, int n_pairs,
              const Out* outs, hipStream_t compute, Run&& run, hipError_t* err) {
    std::lock_guard<std::mutex> lock(mu_);
    *err = hipSuccess;
    if ((*err = ensure_device()) != hipSuccess) return -1;
    const bool video = video_on_ && cs == ps && cur == prev + ps;
    const size_t span_c = cs * (size_t)(n_pairs - 1) + pitch * (size_t)(rows - 1) + row_bytes;
    const size_t span_p = ps * (size_t)(n_pairs - 1) + pitch * (size_t)(rows - 1) + row_bytes;
    const bool pinned = host_pointer_is_pinned(cur) && host_pointer_is_pinned(cur + span_c - 1) && host_pointer_is_pinned(prev) &&
                        host_pointer_is_pinned(prev + span_p - 1);
    if (!pinned && (*err = ensure_staging()) != hipSuccess) return -1;
    int rc = 0, issued = 0, retired = 0;
    for (int k0 = 0; k0 < n_pairs && rc == 0; k0 += chunk_, ++issued) {
      Slot& s = slot_[issued % SLOTS];
      if (issued >= SLOTS) {
        if ((*err = retire(slot_[retired % SLOTS], outs)) != hipSuccess) { rc = -1; break; }
        ++retired;
      }
      const int c = std::min(chunk_, n_pairs - k0);
      Chunk ch{};
      if (video) {  // frames k0 .. k0 + c of the stream, once
        *err = upload(s, 0, prev + ps * (size_t)k0, ps, pitch, row_bytes, rows, c + 1, pinned);
        ch.d_prev = s.d_frames;
        ch.d_cur = s.d_frames + fb_;
      } else {
        *err = upload(s, 0, cur + cs * (size_t)k0, cs, pitch, row_bytes, rows, c, pinned);
        if (*err == hipSuccess) *err = upload(s, (size_t)chunk_ * fb_, prev + ps * (size_t)k0, ps, pitch, row_bytes, rows, c, pinned);
        ch.d_cur = s.d_frames;
        ch.d_prev = s.d_frames + (size_t)chunk_ * fb_;
      }
      if (*err == hipSuccess) *err = hipEventRecord(s.up, copy_);
      if (*err == hipSuccess) *err = hipStreamWaitEvent(compute, s.up, 0);
      if (*err != hipSuccess) { rc = -1; break; }
      ch.stride = fb_;
      ch.count = c;
      for (int i = 0; i < n_outs_; ++i) ch.d_out[i] = s.d_out[i];
      rc = run(ch, compute);
      if (rc != 0) break;
      for (int i = 0; i < n_outs_ && *err == hipSuccess; ++i)
        *err = hipMemcpyAsync(s.h_out[i], s.d_out[i], out_bpp_[i] * (size_t)c, hipMemcpyDeviceToHost, compute);
      if (*err == hipSuccess) *err = hipEventRecord(s.done, compute);
      if (*err != hipSuccess) { rc = -1; break; }
      s.first = k0;
      s.count = c;
      s.busy = true;
    }
    for (; retired < issued; ++retired) {  // in order; after a failure: drain, keep the first error
      Slot& s = slot_[retired % SLOTS];
      if (!s.busy) continue;
      if (rc == 0) {
        if ((*err = retire(s, outs)) != hipSuccess) rc = -1;
      } else {
        (void)hipEventSynchronize(s.done);
        s.busy = false;
      }
    }
    if (rc != 0) {  // nothing of this call may still be in flight when the caller gets its buffers back
      (void)hipStreamSynchronize(copy_);
      (void)hipStreamSynchronize(compute);
      for (Slot& s : slot_) s.busy = false;
    }
    return rc;
  }

  // A run of FRAMES (no pairing: the estimator's video, whose device entry is stateful and -- resolving its gate -- synchronous): chunk k + 1 is
  // packed and its upload queued BEFORE chunk k is handed to `run`, so the DMA rides beside the kernels and the read-back of chunk k.
  // run(chunk with d_cur = the frames, d_prev = nullptr, count = frames, stream) must have finished its work on `compute` when it returns an
  // error; on success the pipe itself waits for `compute` before it reads the results.
  template <class Run>
  int process_frames(const uint8_t* frames, size_t stride, size_t pitch, int row_bytes, int rows, int n_frames, const Out* outs,
                     hipStream_t compute, Run&& run, hipError_t* err) {
    std::lock_guard<std::mutex> lock(mu_);
    *err = hipSuccess;
    if ((*err = ensure_device()) != hipSuccess) return -1;
    const size_t span = stride * (size_t)(n_frames - 1) + pitch * (size_t)(rows - 1) + row_bytes;
    const bool pinned = host_pointer_is_pinned(frames) && host_pointer_is_pinned(frames + span - 1);
    if (!pinned && (*err = ensure_staging()) != hipSuccess) return -1;
    const int per = 2 * chunk_;  // a slot holds two chunks of frames
    auto queue = [&](int i) -> hipError_t {  // upload of chunk i into slot i % SLOTS
      Slot& s = slot_[i % SLOTS];
      const int f0 = i * per, c = std::min(per, n_frames - f0);
      hipError_t e = upload(s, 0, frames + stride * (size_t)f0, stride, pitch, row_bytes, rows, c, pinned);
      if (e == hipSuccess) e = hipEventRecord(s.up, copy_);
      return e;
    };
    const int n_chunks = (n_frames + per - 1) / per;
    int rc = 0;
    if ((*err = queue(0)) != hipSuccess) rc = -1;
    for (int i = 0; i < n_chunks && rc == 0; ++i) {
      Slot& s = slot_[i % SLOTS];
      const int f0 = i * per, c = std::min(per, n_frames - f0);
      if (i + 1 < n_chunks && (*err = queue(i + 1)) != hipSuccess) { rc = -1; break; }  // (slot i + 1 was drained two chunks ago: everything below is synchronous)
      if ((*err = hipStreamWaitEvent(compute, s.up, 0)) != hipSuccess) { rc = -1; break; }
      Chunk ch{};
      ch.d_cur = s.d_frames;
      ch.d_prev = nullptr;
      ch.stride = fb_;
      ch.count = c;
      for (int k = 0; k < n_outs_; ++k) ch.d_out[k] = s.d_out[k];
      rc = run(ch, compute);
      if (rc != 0) break;
      for (int k = 0; k < n_outs_ && *err == hipSuccess; ++k)
        *err = hipMemcpyAsync(s.h_out[k], s.d_out[k], out_bpp_[k] * (size_t)c, hipMemcpyDeviceToHost, compute);
      if (*err == hipSuccess) *err = hipStreamSynchronize(compute);
      if (*err != hipSuccess) { rc = -1; break; }
      for (int k = 0; k < n_outs_; ++k)
        std::memcpy(static_cast<uint8_t*>(outs[k].user) + outs[k].bytes_per_pair * (size_t)f0, s.h_out[k], outs[k].bytes_per_pair * (size_t)c);
    }
    if (rc != 0) {
      (void)hipStreamSynchronize(copy_);
      (void)hipStreamSynchronize(compute);
    } else {
      (void)hipStreamSynchronize(copy_);  // (the last queued upload has been consumed; nothing may outlive the call)
    }
    return rc;
  }

  void release() {
    for (Slot& s : slot_) {
      if (s.d_frames) (void)hipFree(s.d_frames);
      if (s.h_frames) (void)hipHostFree(s.h_frames);
      for (int i = 0; i < MAX_OUTS; ++i) {
        if (s.d_out[i]) (void)hipFree(s.d_out[i]);
        if (s.h_out[i]) (void)hipHostFree(s.h_out[i]);
      }
      if (s.up) (void)hipEventDestroy(s.up);
      if (s.done) (void)hipEventDestroy(s.done);
      s = Slot{};
    }
    if (copy_) (void)hipStreamDestroy(copy_);
    copy_ = nullptr;
    ready_ = staged_ = false;
  }

 private:
  struct Slot {
    uint8_t* d_frames = nullptr;  // 2 * chunk frames: cur block | prev block, or the chunk + 1 frames of a video run
    uint8_t* h_frames = nullptr;  // pinned staging of the same shape (pageable callers only)
    void* d_out[MAX_OUTS] = {nullptr, nullptr, nullptr};
    void* h_out[MAX_OUTS] = {nullptr, nullptr, nullptr};  // pinned
    hipEvent_t up = nullptr, done = nullptr;
    int first = 0, count = 0;
    bool busy = false;
  };

  hipError_t ensure_device() {
    if (ready_) return hipSuccess;
    hipError_t e;
    if ((e = hipStreamCreateWithFlags(&copy_, hipStreamNonBlocking)) != hipSuccess) return e;
    for (Slot& s : slot_) {
      if ((e = hipMalloc(&s.d_frames, (size_t)2 * chunk_ * fb_)) != hipSuccess) return e;
      for (int i = 0; i < n_outs_; ++i) {
        if ((e = hipMalloc(&s.d_out[i], out_bpp_[i] * (size_t)2 * chunk_)) != hipSuccess) return e;  // (2 x: a slot of the FRAMES form holds two chunks)
        if ((e = hipHostMalloc(&s.h_out[i], out_bpp_[i] * (size_t)2 * chunk_, hipHostMallocDefault)) != hipSuccess) return e;
      }
      if ((e = hipEventCreateWithFlags(&s.up, hipEventDisableTiming)) != hipSuccess) return e;
      if ((e = hipEventCreateWithFlags(&s.done, hipEventDisableTiming)) != hipSuccess) return e;
    }
    ready_ = true;
    return hipSuccess;
  }
  hipError_t ensure_staging() {
    if (staged_) return hipSuccess;
    for (Slot& s : slot_) {
      const hipError_t e = hipHostMalloc(&s.h_frames, (size_t)2 * chunk_ * fb_, hipHostMallocDefault);
      if (e != hipSuccess) return e;
    }
    staged_ = true;
    return hipSuccess;
  }

  // `count` frames from host memory (frame i at src + i * stride, rows `pitch` apart) to s.d_frames + off, packed
  hipError_t upload(Slot& s, size_t off, const uint8_t* src, size_t stride, size_t pitch, int row_bytes, int rows, int count, bool pinned) {
    uint8_t* d = s.d_frames + off;
    const bool dense = pitch == (size_t)row_bytes;
    if (pinned) {
      if (dense && stride == fb_) return hipMemcpyAsync(d, src, fb_ * (size_t)count, hipMemcpyHostToDevice, copy_);
      if (dense) return hipMemcpy2DAsync(d, fb_, src, stride, fb_, (size_t)count, hipMemcpyHostToDevice, copy_);
      for (int i = 0; i < count; ++i) {
        const hipError_t e = hipMemcpy2DAsync(d + fb_ * (size_t)i, (size_t)row_bytes, src + stride * (size_t)i, pitch, (size_t)row_bytes, (size_t)rows,
                                              hipMemcpyHostToDevice, copy_);
        if (e != hipSuccess) return e;
      }
      return hipSuccess;
    }
    uint8_t* h = s.h_frames + off;
    const long total_rows = (long)count * rows;
    const int nt = (int)std::min<long>(threads_, std::max<long>(1, (long)(fb_ * (size_t)count >> 20)));  // one helper per MiB at most
    auto work = [&](int t) {
      const long r0 = total_rows * t / nt, r1 = total_rows * (t + 1) / nt;
      if (dense && stride == fb_) {
        std::memcpy(h + (size_t)r0 * row_bytes, src + (size_t)r0 * row_bytes, (size_t)(r1 - r0) * row_bytes);
        return;
      }
      for (long r = r0; r < r1; ++r) {
        const long f = r / rows, y = r % rows;
        std::memcpy(h + (size_t)r * row_bytes, src + stride * (size_t)f + pitch * (size_t)y, (size_t)row_bytes);
      }
    };
    std::vector<std::thread> helpers;
    helpers.reserve(nt > 1 ? nt - 1 : 0);
    for (int t = 1; t < nt; ++t) helpers.emplace_back(work, t);
    work(0);
    for (std::thread& th : helpers) th.join();
    return hipMemcpyAsync(d, h, fb_ * (size_t)count, hipMemcpyHostToDevice, copy_);
  }

  hipError_t retire(Slot& s, const Out* outs) {
    if (!s.busy) return hipSuccess;
    const hipError_t e = hipEventSynchronize(s.done);
    s.busy = false;
    if (e != hipSuccess) return e;
    for (int i = 0; i < n_outs_; ++i)
      std::memcpy(static_cast<uint8_t*>(outs[i].user) + outs[i].bytes_per_pair * (size_t)s.first, s.h_out[i], outs[i].bytes_per_pair * (size_t)s.count);
    return hipSuccess;
  }

  std::mutex mu_;
  size_t fb_;
  size_t out_bpp_[MAX_OUTS] = {0, 0, 0};
  int n_outs_, chunk_ = 1, threads_ = 4;
  bool video_on_ = true, ready_ = false, staged_ = false;
  hipStream_t copy_ = nullptr;
  Slot slot_[SLOTS];
};

}  // namespace mof
