#ifndef LVREF_GM_VECTOR3_STUB
#define LVREF_GM_VECTOR3_STUB
#include <ros/ros.h>
namespace geometry_msgs {
struct Vector3 { double x = 0, y = 0, z = 0; };
struct Point { double x = 0, y = 0, z = 0; };
}
#endif
