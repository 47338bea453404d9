// forwards the reference header name include/Camera.hpp to the adapter surface (vi-slam_amd/host/vislam_host.hpp)
#include "../vislam_host.hpp"
