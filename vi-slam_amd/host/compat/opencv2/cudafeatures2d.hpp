// compat/opencv2/cudafeatures2d.hpp -- stands in for the OpenCV-CUDA header that
// /root/reference/src/main_vi_slamGPU.cpp:6 includes.  Declares only what that file and the class surface use
// (cv::cuda::DeviceInfo :23, getCudaEnabledDeviceCount :41, setDevice :43; the GpuMat / DescriptorMatcher handle
// members of include/CameraGPU.hpp:29-31 and include/MatcherGPU.hpp:23-24), backed by the C ABI.
#ifndef VISLAM_COMPAT_CUDAFEATURES2D_HPP_
#define VISLAM_COMPAT_CUDAFEATURES2D_HPP_
#include "../../cv_compat.hpp"
#include "../../../../include/vislam_hip.h"
namespace cv { namespace cuda {
struct DeviceInfo { int id = 0; };
inline int getCudaEnabledDeviceCount() { return vis_device_count(); }
void setDevice(int device);          // defined in vislam_host.cpp: remembers the device for the next context
int currentDevice();
// opaque device-buffer handle that replaces the cuda::GpuMat members of CameraGPU / MatcherGPU: main never touches them.
struct GpuMat { int slot = -1; void release() { slot = -1; } bool empty() const { return slot < 0; } };
enum { NORM_L2 = 4, NORM_HAMMING = 6 };
// handle of the device matcher (cuda::DescriptorMatcher::createBFMatcher, src/MatcherGPU.cpp:23,30): the matcher itself is
// the library's knn kernel, the handle only records the norm
struct DescriptorMatcher {
    int norm = NORM_HAMMING;
    static Ptr<DescriptorMatcher> createBFMatcher(int normType = NORM_L2) { Ptr<DescriptorMatcher> p(new DescriptorMatcher()); p->norm = normType; return p; }
};
}}  // namespace cv::cuda
#endif
