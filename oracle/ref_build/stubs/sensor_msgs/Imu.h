#ifndef LVREF_SM_IMU_STUB
#define LVREF_SM_IMU_STUB
#include <geometry_msgs/Vector3.h>
#include <geometry_msgs/Quaternion.h>
namespace sensor_msgs {
struct Imu {
    std_msgs::Header header;
    geometry_msgs::Quaternion orientation;
    geometry_msgs::Vector3 angular_velocity, linear_acceleration;
    typedef boost::shared_ptr<Imu const> ConstPtr;
};
typedef boost::shared_ptr<Imu const> ImuConstPtr;
}
#endif
