#ifndef LVREF_GM_TRANSFORMSTAMPED_STUB
#define LVREF_GM_TRANSFORMSTAMPED_STUB
#include <geometry_msgs/Transform.h>
namespace geometry_msgs { struct TransformStamped { std_msgs::Header header; std::string child_frame_id; Transform transform; }; }
#endif
