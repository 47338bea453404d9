// compat/Visualizer.hpp -- stand-in for /root/reference/include/Visualizer.hpp:20-47 WITHOUT ROS: the shape enum and a VisualizerMarker with
// the constructor and UpdateMessages signatures main_vi_slamGPU.cpp:73,75,132,134 uses; publishing markers to rviz is out of scope.
#ifndef VISLAM_COMPAT_VISUALIZER_HPP_
#define VISLAM_COMPAT_VISUALIZER_HPP_
#include <cstdint>
#include <string>
#include "../vislam_host.hpp"
enum { ARROW = 0u, CUBE = 1u, SPHERE = 2u, CYLINDER = 3u };
class VisualizerMarker {
public:
    VisualizerMarker(std::string marker, std::string headerID, double rate, uint32_t, int32_t, cv::Point3f, cv::Point3f) : markerName(marker), headerFrameID(headerID), rateHZ(rate) {}
    void UpdateMessages(cv::Point3d, Quaterniond) { updates++; }
    std::string getMarkerName() { return markerName; }
    std::string getHeaderFrameID() { return headerFrameID; }
    double getRateHZ() { return rateHZ; }
    long updates = 0;
private:
    std::string markerName, headerFrameID;
    double rateHZ;
};
#endif
