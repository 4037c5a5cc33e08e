// Host-side C++ test driver for include/mof/processors.hpp (the C++ mirror of the reference's
// processor classes). Reads a raw u8 frame sequence, runs the stateful processImage() calls the ROS
// node would make, prints the results; tests/test_gpu_cpp_host.py compares them with the oracle.
//   usage: test_processors fft <frameSize> <sps> <max_px_speed> <nframes> <file>
//          test_processors fftocl ...same...   (peak_model = MOF_PEAK_OCL)
//          test_processors bm  <frameSize> <sps> <radius> <nframes> <file>
//          test_processors fsbm <w> <h> <sps> <step> <radius> <nframes> <file>
//          test_processors fftseq <frameSize> <sps> <max_px_speed> <nframes> <file>   (video on the device, sequence entry)
//          test_processors srseq <res> <M> <nframes> <file>                          (estimator: sequence entry + stateful loop)
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

#include <hip/hip_runtime_api.h>  // only for the sequence modes below: device buffers for a video (the mirror itself needs no HIP)

#include "mof/processors.hpp"

static std::vector<uint8_t> read_all(const char* path, size_t bytes) {
  std::vector<uint8_t> buf(bytes);
  FILE* f = std::fopen(path, "rb");
  if (!f || std::fread(buf.data(), 1, bytes, f) != bytes) {
    std::fprintf(stderr, "cannot read %zu bytes from %s\n", bytes, path);
    std::exit(2);
  }
  std::fclose(f);
  return buf;
}

int main(int argc, char** argv) {
  try {
    if (argc >= 7 && (!std::strcmp(argv[1], "fft") || !std::strcmp(argv[1], "fftocl"))) {
      const int fs = std::atoi(argv[2]), sps = std::atoi(argv[3]), n = std::atoi(argv[5]);
      const double mps = std::atof(argv[4]);
      auto frames = read_all(argv[6], (size_t)fs * fs * n);
      const int peak = !std::strcmp(argv[1], "fftocl") ? MOF_PEAK_OCL : MOF_PEAK_OPENCV;
      mof::FftMethod proc(fs, sps, mps, false, false, false, false, nullptr, 30, "unused.cl", true, 0, peak);
      std::vector<uint8_t> zeros((size_t)fs * fs, 0);
      proc.setImPrev(mof::ImageView{zeros.data(), fs, fs, (size_t)fs});  // optic_flow.cpp:1016-1018
      std::vector<mof::Point2d> raw;
      for (int t = 0; t < n; ++t) {
        auto v = proc.processImage(mof::ImageView{frames.data() + (size_t)t * fs * fs, fs, fs, (size_t)fs}, false, false,
                                   mof::Point2i{fs / 2, fs / 2}, 0.0, mof::Point2d{0, 0}, raw, 300, 300);
        std::printf("frame %d n %zu", t, v.size());
        for (auto& p : v) std::printf(" %.17g %.17g", p.x, p.y);
        std::printf("\n");
      }
      return 0;
    }
    if (argc >= 7 && !std::strcmp(argv[1], "bm")) {
      const int fs = std::atoi(argv[2]), sps = std::atoi(argv[3]), r = std::atoi(argv[4]), n = std::atoi(argv[5]);
      auto frames = read_all(argv[6], (size_t)fs * fs * n);
      mof::BlockMethod proc(fs, sps, r, 2 * r + 1, (2 * r + 1) * (2 * r + 1), 0);
      for (int t = 0; t < n; ++t) {
        auto v = proc.processImage(mof::ImageView{frames.data() + (size_t)t * fs * fs, fs, fs, (size_t)fs}, false, false,
                                   mof::Point2i{fs / 2, fs / 2}, 0.0, mof::Point2d{0, 0});
        std::printf("frame %d mode %d %d refined %.9g %.9g blocks", t, proc.mode().x, proc.mode().y, v.at(0).x, v.at(0).y);
        for (size_t b = 0; b < proc.flowX().size(); ++b) std::printf(" %d %d", proc.flowX()[b], proc.flowY()[b]);
        std::printf("\n");
      }
      return 0;
    }
    if (argc >= 9 && !std::strcmp(argv[1], "fsbm")) {
      const int w = std::atoi(argv[2]), h = std::atoi(argv[3]), sps = std::atoi(argv[4]), step = std::atoi(argv[5]),
                r = std::atoi(argv[6]), n = std::atoi(argv[7]);
      auto frames = read_all(argv[8], (size_t)w * h * n);
      mof::FastSpacedBMMethod proc(sps, r, step, w, h);
      for (int t = 0; t < n; ++t) {
        auto v = proc.processImage(mof::ImageView{frames.data() + (size_t)t * w * h, h, w, (size_t)w}, false, false,
                                   mof::Point2i{w / 2, h / 2}, 0.0, mof::Point2d{0, 0});
        std::printf("frame %d mode %g %g blocks", t, (double)v.at(0).x, (double)v.at(0).y);
        for (size_t b = 0; b < proc.flowX().size(); ++b) std::printf(" %d %d", proc.flowX()[b], proc.flowY()[b]);
        std::printf("\n");
      }
      return 0;
    }
    if (argc >= 7 && !std::strcmp(argv[1], "fftvideo")) {  // a video in HOST memory: FftMethod::processVideo (pageable, then pinned through mof_host_register)
      const int fs = std::atoi(argv[2]), sps = std::atoi(argv[3]), n = std::atoi(argv[5]);
      const double mps = std::atof(argv[4]);
      auto frames = read_all(argv[6], (size_t)fs * fs * n);
      mof::FftMethod proc(fs, sps, mps);
      const size_t per = (size_t)proc.sqNum() * proc.sqNum();
      auto pageable = proc.processVideo(frames.data(), (size_t)fs * fs, (size_t)fs, n);
      if (mof_host_register(frames.data(), frames.size()) != MOF_OK) throw std::runtime_error(mof_last_error());
      auto pinned = proc.processVideo(frames.data(), (size_t)fs * fs, (size_t)fs, n);
      if (mof_host_unregister(frames.data()) != MOF_OK) throw std::runtime_error(mof_last_error());
      if (pageable.size() != per * (n - 1) || pinned.size() != pageable.size() ||
          std::memcmp(pageable.data(), pinned.data(), pageable.size() * sizeof(mof::Point2d)) != 0)
        throw std::runtime_error("pinned and pageable videos disagree");
      for (int t = 0; t + 1 < n; ++t) {
        std::printf("pair %d n %zu", t, per);
        for (size_t i = 0; i < per; ++i) std::printf(" %.17g %.17g", pageable[(size_t)t * per + i].x, pageable[(size_t)t * per + i].y);
        std::printf("\n");
      }
      return 0;
    }
    if (argc >= 7 && !std::strcmp(argv[1], "fftseq")) {
      const int fs = std::atoi(argv[2]), sps = std::atoi(argv[3]), n = std::atoi(argv[5]);
      const double mps = std::atof(argv[4]);
      auto frames = read_all(argv[6], (size_t)fs * fs * n);
      mof::FftMethod proc(fs, sps, mps);
      const size_t per = (size_t)proc.sqNum() * proc.sqNum() * 2;
      uint8_t* d_frames = nullptr;
      double* d_out = nullptr;
      if (hipMalloc((void**)&d_frames, frames.size()) != hipSuccess || hipMalloc((void**)&d_out, per * (n - 1) * sizeof(double)) != hipSuccess ||
          hipMemcpy(d_frames, frames.data(), frames.size(), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("device buffers");
      proc.processSequenceDevice(d_frames, (size_t)fs * fs, (size_t)fs, n, d_out, nullptr);
      std::vector<double> out(per * (n - 1));
      if (hipDeviceSynchronize() != hipSuccess || hipMemcpy(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess)
        throw std::runtime_error("read-back");
      for (int t = 0; t + 1 < n; ++t) {
        std::printf("pair %d n %zu", t, per / 2);
        for (size_t i = 0; i < per; ++i) std::printf(" %.17g", out[(size_t)t * per + i]);
        std::printf("\n");
      }
      (void)hipFree(d_frames);
      (void)hipFree(d_out);
      return 0;
    }
    if (argc >= 6 && !std::strcmp(argv[1], "srseq")) {
      const int res = std::atoi(argv[2]), n = std::atoi(argv[4]);
      const double M = std::atof(argv[3]);
      auto frames = read_all(argv[5], (size_t)res * res * n);
      mof::scaleRotationEstimator seq(res, M), one(res, M);
      uint8_t* d_frames = nullptr;
      double* d_out = nullptr;
      if (hipMalloc((void**)&d_frames, frames.size()) != hipSuccess || hipMalloc((void**)&d_out, (size_t)n * 4 * sizeof(double)) != hipSuccess ||
          hipMemcpy(d_frames, frames.data(), frames.size(), hipMemcpyHostToDevice) != hipSuccess)
        throw std::runtime_error("device buffers");
      const int gated = seq.processSequenceDevice(d_frames, (size_t)res * res, (size_t)res, n, d_out, nullptr);
      std::vector<double> out((size_t)n * 4);
      if (hipMemcpy(out.data(), d_out, out.size() * sizeof(double), hipMemcpyDeviceToHost) != hipSuccess) throw std::runtime_error("read-back");
      std::printf("gated %d\n", gated);
      for (int t = 0; t < n; ++t) {
        const mof::Point2d p = one.processImage(mof::ImageView{frames.data() + (size_t)t * res * res, res, res, (size_t)res}, false, false);
        std::printf("frame %d seq %.17g %.17g %.17g %.17g stateful %.17g %.17g\n", t, out[4 * (size_t)t], out[4 * (size_t)t + 1],
                    out[4 * (size_t)t + 2], out[4 * (size_t)t + 3], p.x, p.y);
      }
      (void)hipFree(d_frames);
      (void)hipFree(d_out);
      return 0;
    }
    std::fprintf(stderr, "bad usage\n");
    return 2;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
}
