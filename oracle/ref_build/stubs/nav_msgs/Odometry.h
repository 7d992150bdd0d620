#ifndef LVREF_NAV_ODOMETRY_STUB
#define LVREF_NAV_ODOMETRY_STUB
#include <geometry_msgs/Pose.h>
namespace nav_msgs { struct Odometry { std_msgs::Header header; std::string child_frame_id; geometry_msgs::PoseWithCovariance pose; geometry_msgs::TwistWithCovariance twist; }; }
#endif
