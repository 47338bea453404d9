// stands in for opencv2/xfeatures2d/cuda.hpp (src/main_vi_slamGPU.cpp:7); SURF_CUDA is outside the hot path.
#ifndef VISLAM_COMPAT_XFEATURES2D_CUDA_HPP_
#define VISLAM_COMPAT_XFEATURES2D_CUDA_HPP_
#include "../cudafeatures2d.hpp"
#endif
