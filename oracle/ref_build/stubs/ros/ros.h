// ros/ros.h — stub (oracle/ref_build: compiling the reference's sources without ROS; TEST INFRASTRUCTURE).  Only what
// include/Headers/*.hpp and src/**.cpp of the reference name; nothing publishes anywhere.
#ifndef LVREF_ROS_STUB
#define LVREF_ROS_STUB
#include <cassert>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <iomanip>
#include <iostream>
#include <memory>
#include <string>
#include <vector>
#include <boost/shared_ptr.hpp>
#include <Eigen/Dense>

#define ROS_ERROR(...) do { std::fprintf(stderr, "[ROS_ERROR] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) ROS_ERROR(__VA_ARGS__)
#define ROS_INFO(...) do { } while (0)

namespace ros {
struct Time {   // sec / nsec like the real one, so that a stamp survives the trip through a message exactly
    uint32_t sec = 0, nsec = 0;
    Time() {}
    Time(double t) {   // TimeBase::fromSec: floor + round
        const int64_t s = (int64_t)std::floor(t);
        int64_t n = (int64_t)std::llround((t - (double)s) * 1e9);
        sec = (uint32_t)(s + n / 1000000000ll);
        nsec = (uint32_t)(n % 1000000000ll);
    }
    double toSec() const { return (double)sec + 1e-9 * (double)nsec; }
    uint64_t toNSec() const { return (uint64_t)sec * 1000000000ull + (uint64_t)nsec; }
    static Time now() { return Time(); }
};
struct Publisher {
    template <typename M> void publish(const M&) const {}
    int getNumSubscribers() const { return 0; }
};
struct Subscriber {};
struct NodeHandle {
    template <typename M> Publisher advertise(const std::string&, int) { return Publisher(); }
    template <typename T> bool param(const std::string&, T& v, const T& d) { v = d; return false; }
};
struct Rate { Rate(double) {} void sleep() {} };
inline bool ok() { return false; }
inline void spinOnce() {}
inline void init(int&, char**, const std::string&) {}
}  // namespace ros

namespace std_msgs {
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
}
#endif
