// Compile-and-run test of the literal drop-in adapter `MofFftMethod : public OpticFlowCalc`
// (include/mof/processors.hpp) against the reference's own interface header
// /root/reference/include/OpticFlowCalc.h:6-22 (on the include path via -I, read-only) and the call sites of the
// node: construction /root/reference/src/optic_flow.cpp:1001-1002, priming :1016-1018, per-frame calls :1685-1690.
// OpenCV is absent here: tests/stubs/ supplies the handful of cv:: types for type-checking only.
//   usage: test_adapter <frameSize> <sps> <max_px_speed> <nframes> <file>
// Prints, per frame, the vectors of processImage and (where the geometry has one) processImageLongRange.
#include <cstdio>
#include <cstdlib>
#include <string>
#include <type_traits>
#include <vector>

#include "mof/processors.hpp"

#if !(defined(__has_include) && __has_include(<OpticFlowCalc.h>))
#error "test_adapter needs -I/root/reference/include (OpticFlowCalc.h) and -Itests/stubs"
#endif

static_assert(std::is_base_of<OpticFlowCalc, MofFftMethod>::value, "the adapter must derive from the reference's OpticFlowCalc");
static_assert(!std::is_abstract<MofFftMethod>::value, "the adapter must override OpticFlowCalc::processImage (the "
              "reference's own BlockMethod does not: SURVEY F4)");

int main(int argc, char** argv) {
  if (argc < 6) return 2;
  const int _frame_size_ = std::atoi(argv[1]), _sample_point_size_ = std::atoi(argv[2]), n = std::atoi(argv[4]);
  const double _max_pixel_speed_ = std::atof(argv[3]);
  std::vector<unsigned char> buf((size_t)_frame_size_ * _frame_size_ * n);
  FILE* f = std::fopen(argv[5], "rb");
  if (!f || std::fread(buf.data(), 1, buf.size(), f) != buf.size()) return 2;
  std::fclose(f);
  try {
    // -- optic_flow.cpp:1001-1002 (same argument list; the member there is `FftMethod* fftProcessor_`, :251) --------
    bool store_video_ = false, _raw_enabled_ = false, _rotation_correction_ = false, _useOCL_ = false;
    std::string video_path_ = "", _fft_cl_file_ = "unused.cl";
    int videoFPS = 30;
    MofFftMethod* fftProcessor_ = new MofFftMethod(_frame_size_, _sample_point_size_, _max_pixel_speed_, store_video_, _raw_enabled_,
                                                   _rotation_correction_, false, &video_path_, videoFPS, _fft_cl_file_, _useOCL_);
    // -- :1016-1018 ----------------------------------------------------------------------------------------------------
    cv::Mat imPrev_ = cv::Mat(_frame_size_, _frame_size_, CV_8UC1);
    imPrev_ = cv::Scalar(0);
    fftProcessor_->setImPrev(imPrev_);
    // -- :1685-1690, through the abstract interface for processImage (virtual) ------------------------------------------
    OpticFlowCalc* processClass = fftProcessor_;
    std::vector<cv::Point2d> mrs_optic_flow_vectors, mrs_optic_flow_vectors_raw;
    cv::Point2i mid_point(_frame_size_ / 2, _frame_size_ / 2);
    double temp_angle_diff = 0.0, fx_ = 300, fy_ = 300;
    bool _gui_ = false, _debug_ = false;
    for (int t = 0; t < n; ++t) {
      // a cv::Mat header over the caller's pixels, wider than the frame would be legal too (step is forwarded)
      cv::Mat imCurr_(_frame_size_, _frame_size_, CV_8UC1, buf.data() + (size_t)t * _frame_size_ * _frame_size_);
      const bool long_range_mode = (t % 2 == 1) && (_frame_size_ / _sample_point_size_ >= 4) && (_frame_size_ % 4 == 0);
      if (!long_range_mode)
        mrs_optic_flow_vectors =
            processClass->processImage(imCurr_, _gui_, _debug_, mid_point, temp_angle_diff, cv::Point(0, 0), mrs_optic_flow_vectors_raw, fx_, fy_);
      else
        mrs_optic_flow_vectors =
            fftProcessor_->processImageLongRange(imCurr_, _gui_, _debug_, mid_point, temp_angle_diff, cv::Point(0, 0), mrs_optic_flow_vectors_raw, fx_, fy_);
      std::printf("frame %d %s n %zu", t, long_range_mode ? "lr" : "std", mrs_optic_flow_vectors.size());
      for (auto& p : mrs_optic_flow_vectors) std::printf(" %.17g %.17g", p.x, p.y);
      std::printf("\n");
    }
    // a wrong-sized frame must be refused, not read out of bounds
    cv::Mat small(_frame_size_ / 2, _frame_size_ / 2, CV_8UC1);
    small = cv::Scalar(1);
    bool threw = false;
    try {
      processClass->processImage(small, false, false, mid_point, 0.0, cv::Point(0, 0), mrs_optic_flow_vectors_raw);
    } catch (const std::exception&) {
      threw = true;
    }
    std::printf("wrong-size %s\n", threw ? "refused" : "ACCEPTED");
    delete fftProcessor_;
    return threw ? 0 : 4;
  } catch (const std::exception& e) {
    std::fprintf(stderr, "error: %s\n", e.what());
    return 3;
  }
}
