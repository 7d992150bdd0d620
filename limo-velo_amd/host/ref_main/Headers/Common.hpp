// Headers/Common.hpp as the reference's src/main.cpp finds it when it is compiled against the SHIM (host/Makefile `refmain`):
// the include path puts this directory where include/ of the reference was, so main.cpp's own `#include "Headers/*.hpp"` lines
// (:3-11) resolve to the shim's classes — Accumulator, Compensator, Mapper, Localizator, State, Points, Params — and nothing of the
// reference's headers is read.  TEST INFRASTRUCTURE (the drop-in proof north_star asks for: "keeping the existing
// Localizator / Mapper C++ API so it drops into the ROS pipeline unchanged").
#pragma once
#include <cfloat>
#include <algorithm>

#include <ros/ros.h>

#include "limovelo_shim.hpp"

// Publishers (reference include/Headers/Publishers.hpp): rviz / tf output, no part of the data path — every method a no-op here.
class Publishers {
  public:
    explicit Publishers(ros::NodeHandle&) {}
    void state(const State&, bool) {}
    void tf(const State&) {}
    void pointcloud(Points&, bool) {}
    void extrinsics(const State&) {}
};
