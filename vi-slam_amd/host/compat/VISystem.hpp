// forwards the reference header name include/VISystem.hpp to the adapter surface (vi-slam_amd/host/vislam_host.hpp)
#include "../vislam_host.hpp"
