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
#include <functional>
#include <iostream>
#include <map>
#include <memory>
#include <string>
#include <vector>
#include <boost/shared_ptr.hpp>
#include <Eigen/Dense>

#define ROS_ERROR(...) do { std::fprintf(stderr, "[ROS_ERROR] "); std::fprintf(stderr, __VA_ARGS__); std::fprintf(stderr, "\n"); } while (0)
#define ROS_WARN(...) ROS_ERROR(__VA_ARGS__)
#define ROS_INFO(...) do { } while (0)

// ---- hooks the whole-loop replay uses (oracle/ref_build/ref_stream_main.cpp runs the reference's src/main.cpp on a recorded
// stream): parameter values for fill_config (src/main.cpp:133-175), the subscribers' callbacks by topic, the message feed behind
// ros::ok() / ros::spinOnce()
namespace lvref {
struct Overrides {
    std::map<std::string, double> num;
    std::map<std::string, std::string> str;
    std::map<std::string, std::vector<double>> vec;
};
inline Overrides& overrides() { static Overrides o; return o; }
inline bool lookup(const std::string& n, bool& v) { auto it = overrides().num.find(n); if (it == overrides().num.end()) return false; v = it->second != 0.0; return true; }
inline bool lookup(const std::string& n, int& v) { auto it = overrides().num.find(n); if (it == overrides().num.end()) return false; v = (int)it->second; return true; }
inline bool lookup(const std::string& n, float& v) { auto it = overrides().num.find(n); if (it == overrides().num.end()) return false; v = (float)it->second; return true; }
inline bool lookup(const std::string& n, double& v) { auto it = overrides().num.find(n); if (it == overrides().num.end()) return false; v = it->second; return true; }
inline bool lookup(const std::string& n, std::string& v) { auto it = overrides().str.find(n); if (it == overrides().str.end()) return false; v = it->second; return true; }
template <typename E>
inline bool lookup(const std::string& n, std::vector<E>& v) {
    auto it = overrides().vec.find(n);
    if (it == overrides().vec.end()) return false;
    v.clear();
    for (double x : it->second) v.push_back((E)x);
    return true;
}
struct Feed {
    std::function<bool()> ok;
    std::function<void()> spin;
    std::map<std::string, std::function<void(const void*)>> subscribers;   // topic -> callback(pointer to the message's ConstPtr)
};
inline Feed& feed() { static Feed f; return f; }
}  // namespace lvref

namespace ros {
struct Time {   // sec / nsec like the real one, so that a stamp survives the trip through a message exactly
    uint32_t sec = 0, nsec = 0;
    double exact = 0.0;     // (a time constructed from a double hands the same double back)
    bool has_exact = false;
    Time() {}
    Time(double t) : exact(t), has_exact(true) {   // TimeBase::fromSec: floor + round
        const int64_t s = (int64_t)std::floor(t);
        int64_t n = (int64_t)std::llround((t - (double)s) * 1e9);
        sec = (uint32_t)(s + n / 1000000000ll);
        nsec = (uint32_t)(n % 1000000000ll);
    }
    double toSec() const { return has_exact ? exact : (double)sec + 1e-9 * (double)nsec; }
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
    template <typename T> bool param(const std::string& name, T& v, const T& d) {
        if (lvref::lookup(name, v)) return true;
        v = d;
        return false;
    }
    template <typename M, typename T>
    Subscriber subscribe(const std::string& topic, uint32_t, void (T::*fp)(const boost::shared_ptr<M const>&), T* obj) {
        lvref::feed().subscribers[topic] = [fp, obj](const void* p) { (obj->*fp)(*static_cast<const boost::shared_ptr<M const>*>(p)); };
        return Subscriber();
    }
};
struct Rate { Rate(double) {} void sleep() {} };
inline bool ok() { return lvref::feed().ok ? lvref::feed().ok() : false; }
inline void spinOnce() { if (lvref::feed().spin) lvref::feed().spin(); }
inline void init(int&, char**, const std::string&) {}
}  // namespace ros

namespace std_msgs {
struct Header { uint32_t seq = 0; ros::Time stamp; std::string frame_id; };
}
#endif
