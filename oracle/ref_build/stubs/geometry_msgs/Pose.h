#ifndef LVREF_GM_POSE_STUB
#define LVREF_GM_POSE_STUB
#include <geometry_msgs/Vector3.h>
#include <geometry_msgs/Quaternion.h>
namespace geometry_msgs {
struct Pose { Point position; Quaternion orientation; };
struct PoseWithCovariance { Pose pose; };
struct Twist { Vector3 linear, angular; };
struct TwistWithCovariance { Twist twist; };
}
#endif
