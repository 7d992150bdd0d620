// tf stub (oracle/ref_build; TEST INFRASTRUCTURE): the types include/Headers/Publishers.hpp names
#ifndef LVREF_TF_STUB
#define LVREF_TF_STUB
#include <string>
#include <ros/ros.h>
namespace tf {
struct Vector3 { double x, y, z; Vector3(double a = 0, double b = 0, double c = 0) : x(a), y(b), z(c) {} };
struct Quaternion { double x_ = 0, y_ = 0, z_ = 0, w_ = 1; void setW(double v) { w_ = v; } void setX(double v) { x_ = v; } void setY(double v) { y_ = v; } void setZ(double v) { z_ = v; } };
struct Transform { Vector3 o; Quaternion q; void setOrigin(const Vector3& v) { o = v; } void setRotation(const Quaternion& r) { q = r; } };
struct StampedTransform { StampedTransform(const Transform&, const ros::Time&, const std::string&, const std::string&) {} };
struct TransformBroadcaster { void sendTransform(const StampedTransform&) {} };
struct TransformListener {};
}  // namespace tf
#endif
