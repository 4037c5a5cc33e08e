// tests/stubs/cv_bridge/cv_bridge.h -- COMPILE-TEST STAND-IN (see ../opencv2/core.hpp).
// /root/reference/include/OpticFlowCalc.h:4 includes <cv_bridge/cv_bridge.h> only to reach cv::Mat / cv::Point.
#pragma once
#include <opencv2/core.hpp>

#include <string>
#include <vector>
