// compat/opencv2/core.hpp -- forwards the OpenCV header name /root/reference/src/main_vi_slamGPU.cpp:8-10 includes to the value types the class
// surface needs (host/cv_compat.hpp: Mat, Point3_, KeyPoint, DMatch, Matx33f, Ptr, String, CommandLineParser); with VISLAM_USE_OPENCV the real header
#include "../../cv_compat.hpp"
