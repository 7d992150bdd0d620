#ifndef LVREF_GM_POSEARRAY_STUB
#define LVREF_GM_POSEARRAY_STUB
#include <geometry_msgs/Pose.h>
namespace geometry_msgs { struct PoseArray { std_msgs::Header header; std::vector<Pose> poses; }; }
#endif
