#ifndef LVREF_GM_QUATERNION_STUB
#define LVREF_GM_QUATERNION_STUB
#include <ros/ros.h>
namespace geometry_msgs { struct Quaternion { double x = 0, y = 0, z = 0, w = 1; }; }
#endif
