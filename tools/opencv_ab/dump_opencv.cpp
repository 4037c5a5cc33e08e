// dump_opencv.cpp -- runs the REAL OpenCV calls of the reference's hot path on the inputs written by export_inputs.py
// and stores their outputs as <name>_cv*.bin (raw little-endian) for compare_with_oracle.py. Needs OpenCV 3.x or 4.x;
// never built or run by this project's tests (see README.md).
//   g++ -O2 -std=c++14 dump_opencv.cpp -o dump_opencv $(pkg-config --cflags --libs opencv4)
//   ./dump_opencv ab_data/
// Calls mirrored (reference file:line):
//   cv::phaseCorrelate                      src/FftMethod.cpp:1836, src/scaleRotationEstimator.cpp:117
//   cv::logPolar / cvLogPolar               src/scaleRotationEstimator.cpp:45, :112 (Noetic) / :44, :110 (Melodic)
//   cv::resize 1/4, x2                      src/FftMethod.cpp:1931-1932, src/BlockMethod.cpp:110-111
//   cv::cvtColor(CV_RGB2GRAY)               src/optic_flow.cpp:1622
//   cv::undistortPoints, cv::findHomography(RANSAC, 0.01), cv::decomposeHomographyMat   src/optic_flow.cpp:549-550, :559, :595
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <sstream>
#include <string>
#include <vector>

#include <opencv2/calib3d.hpp>
#include <opencv2/core.hpp>
#include <opencv2/imgproc.hpp>
#if CV_MAJOR_VERSION < 4
#include <opencv2/imgproc/imgproc_c.h>
#endif

static std::string dir;

template <class T>
static std::vector<T> load(const std::string& name, size_t count) {
  std::vector<T> v(count);
  std::ifstream f(dir + "/" + name + ".bin", std::ios::binary);
  if (!f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(count * sizeof(T)))) {
    std::cerr << "cannot read " << name << "\n";
    std::exit(2);
  }
  return v;
}

template <class T>
static void save(const std::string& name, const T* p, size_t count) {
  std::ofstream f(dir + "/" + name + ".bin", std::ios::binary);
  f.write(reinterpret_cast<const char*>(p), (std::streamsize)(count * sizeof(T)));
}

static void save_mat_u8(const std::string& name, const cv::Mat& m) {
  cv::Mat c = m.isContinuous() ? m : m.clone();
  save(name, c.ptr<unsigned char>(), (size_t)c.total() * c.channels());
}

int main(int argc, char** argv) {
  if (argc < 2) return 2;
  dir = argv[1];
  std::printf("OpenCV %s\n", CV_VERSION);
  std::ifstream man(dir + "/manifest.txt");
  std::string line;
  while (std::getline(man, line)) {
    std::istringstream is(line);
    std::string kind, name;
    is >> kind >> name;
    if (kind == "pc") {
      int n;
      is >> n;
      auto a = load<float>(name + "_a", (size_t)n * n), b = load<float>(name + "_b", (size_t)n * n);
      cv::Mat A(n, n, CV_32FC1, a.data()), B(n, n, CV_32FC1, b.data());
      const cv::Point2d p = cv::phaseCorrelate(A, B);
      const double out[2] = {p.x, p.y};
      save(name + "_cv", out, 2);
    } else if (kind == "optdft") {
      int count;
      is >> count;
      auto sizes = load<int>(name + "_sizes", (size_t)count);
      std::vector<int> out((size_t)count);
      for (int i = 0; i < count; ++i) out[(size_t)i] = cv::getOptimalDFTSize(sizes[(size_t)i]);  // what cv::phaseCorrelate pads to
      save(name + "_cv", out.data(), (size_t)count);
    } else if (kind == "lp") {
      int res;
      double M;
      is >> res >> M;
      auto src = load<unsigned char>(name + "_src", (size_t)res * res);
      cv::Mat S(res, res, CV_8UC1, src.data());
      const cv::Point2f center((float)(res / 2), (float)(res / 2));  // cv::Point2f(resolution / 2, resolution / 2), :25
      for (int interp : {cv::INTER_CUBIC, cv::INTER_LANCZOS4}) {
        cv::Mat D(res, res, CV_8UC1, cv::Scalar(37));
        cv::logPolar(S, D, center, M, interp);
        save_mat_u8(name + "_cv_logPolar_i" + std::to_string(interp), D);
#if CV_MAJOR_VERSION < 4
        cv::Mat D3(res, res, CV_8UC1, cv::Scalar(37));
        IplImage ia = S, ib = D3;  // as the reference does under ROS Melodic
        cvLogPolar(&ia, &ib, cvPoint2D32f(center.x, center.y), M, interp);
        save_mat_u8(name + "_cv_cvLogPolar_i" + std::to_string(interp), D3);
#endif
        // cv::remap on the ORACLE's maps: tells a map difference from a remap difference
        for (int variant = 0; variant < 2; ++variant) {
          auto mx = load<float>(name + "_mapx_v" + std::to_string(variant), (size_t)res * res);
          auto my = load<float>(name + "_mapy_v" + std::to_string(variant), (size_t)res * res);
          cv::Mat MX(res, res, CV_32FC1, mx.data()), MY(res, res, CV_32FC1, my.data()), DR(res, res, CV_8UC1, cv::Scalar(37));
          cv::remap(S, DR, MX, MY, interp, cv::BORDER_TRANSPARENT);
          save_mat_u8(name + "_cv_remap_i" + std::to_string(interp) + "_v" + std::to_string(variant), DR);
        }
      }
    } else if (kind == "srseq") {
      // scaleRotationEstimator::processImage as a stream (src/scaleRotationEstimator.cpp:34-148), call for call
      int res, nf;
      double M;
      is >> res >> M >> nf;
      auto video = load<unsigned char>(name + "_video", (size_t)nf * res * res);
      const cv::Point2f center((float)(res / 2), (float)(res / 2));  // :25
      cv::Mat tempIm = cv::Mat::zeros(res, res, CV_8UC1), tempF, prevF;  // :27
      std::vector<double> out((size_t)nf * 4);
      bool first = true;  // :31
      for (int t = 0; t < nf; ++t) {
        cv::Mat im(res, res, CV_8UC1, video.data() + (size_t)t * res * res);
        double sc = 1.0, ro = 0.0, px = 0.0, py = 0.0;
        if (first) {
          cv::logPolar(im, tempIm, center, M, cv::INTER_CUBIC);  // :45
          tempIm.convertTo(prevF, CV_32FC1);                      // :47-48
          first = false;                                          // :73
        } else {
          cv::logPolar(im, tempIm, center, M, cv::INTER_LANCZOS4);  // :112
          tempIm.convertTo(tempF, CV_32FC1);                        // :115
          const cv::Point2d pt = cv::phaseCorrelate(tempF, prevF);  // :117
          px = pt.x, py = pt.y;
          if (!(std::fabs(pt.x) > res / 2)) {                       // :119-121
            sc = std::exp(pt.x / M);                                 // :123
            ro = (pt.y / ((double)res / 360.0)) * (3.14159265358979323846 / 180.0);  // :124
            prevF = tempF.clone();                                   // :128
          }
        }
        out[4 * (size_t)t] = sc, out[4 * (size_t)t + 1] = ro, out[4 * (size_t)t + 2] = px, out[4 * (size_t)t + 3] = py;
      }
      save(name + "_cv", out.data(), out.size());
    } else if (kind == "resize_quarter" || kind == "resize_2x" || kind == "gray") {
      int h, w;
      is >> h >> w;
      if (kind == "gray") {
        auto src = load<unsigned char>(name + "_src", (size_t)h * w * 3);
        cv::Mat S(h, w, CV_8UC3, src.data()), D;
        cv::cvtColor(S, D, cv::COLOR_RGB2GRAY);
        save_mat_u8(name + "_cv", D);
      } else {
        auto src = load<unsigned char>(name + "_src", (size_t)h * w);
        cv::Mat S(h, w, CV_8UC1, src.data()), D;
        if (kind == "resize_quarter") cv::resize(S, D, cv::Size(), 0.25, 0.25);  // 1.0 / LONG_RANGE_RATIO, FftMethod.cpp:1931
        else cv::resize(S, D, cv::Size(2 * w, 2 * h));                             // BlockMethod.cpp:110
        save_mat_u8(name + "_cv", D);
      }
    } else if (kind == "undistort") {
      int n;
      double c[9], ulx;
      is >> n;
      for (double& v : c) is >> v;
      is >> ulx;
      auto pts = load<double>(name + "_pts", (size_t)2 * n);
      std::vector<cv::Point2d> in(n), out;
      for (int i = 0; i < n; ++i) in[i] = cv::Point2d(pts[2 * i], pts[2 * i + 1]);
      cv::Matx33d K(c[0], 0, c[2] - ulx, 0, c[1], c[3], 0, 0, 1);  // camMatrixLocal(0, 2) -= ulCorner.x, optic_flow.cpp:522
      cv::Mat dist = (cv::Mat_<double>(1, 5) << c[4], c[5], c[6], c[7], c[8]);
      cv::undistortPoints(in, out, K, dist);
      std::vector<double> o(2 * n);
      for (int i = 0; i < n; ++i) o[2 * i] = out[i].x, o[2 * i + 1] = out[i].y;
      save(name + "_cv", o.data(), o.size());
    } else if (kind == "homography") {
      int n;
      is >> n;
      auto a = load<double>(name + "_a", (size_t)2 * n), b = load<double>(name + "_b", (size_t)2 * n);
      std::vector<cv::Point2d> A(n), B(n);
      for (int i = 0; i < n; ++i) A[i] = cv::Point2d(a[2 * i], a[2 * i + 1]), B[i] = cv::Point2d(b[2 * i], b[2 * i + 1]);
      cv::Mat mask;
      cv::Mat H = cv::findHomography(A, B, cv::RANSAC, 0.01, mask);
      std::vector<double> h(9, 0.0);
      if (!H.empty()) for (int i = 0; i < 9; ++i) h[i] = H.at<double>(i / 3, i % 3);
      save(name + "_cv_H", h.data(), 9);
      std::vector<unsigned char> m(n, 0);
      for (int i = 0; i < n && !mask.empty(); ++i) m[i] = mask.at<unsigned char>(i);
      save(name + "_cv_mask", m.data(), m.size());
      if (!H.empty()) {
        std::vector<cv::Mat> R, t, nn;
        const int k = cv::decomposeHomographyMat(H, cv::Matx33d::eye(), R, t, nn);
        std::vector<double> o;
        for (int s = 0; s < k; ++s) {
          for (int i = 0; i < 9; ++i) o.push_back(R[s].at<double>(i / 3, i % 3));
          for (int i = 0; i < 3; ++i) o.push_back(t[s].at<double>(i));
          for (int i = 0; i < 3; ++i) o.push_back(nn[s].at<double>(i));
        }
        save(name + "_cv_decomp", o.data(), o.size());  // k x (9 + 3 + 3) doubles
      }
    }
    std::printf("%s %s done\n", kind.c_str(), name.c_str());
  }
  return 0;
}
