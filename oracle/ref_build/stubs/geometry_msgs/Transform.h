#ifndef LVREF_GM_TRANSFORM_STUB
#define LVREF_GM_TRANSFORM_STUB
#include <geometry_msgs/Vector3.h>
#include <geometry_msgs/Quaternion.h>
namespace geometry_msgs { struct Transform { Vector3 translation; Quaternion rotation; }; }
#endif
