// ros/ros.h — what the reference's src/main.cpp names of ROS, for compiling THAT FILE, unchanged, against the shim
// (host/Makefile target `refmain`; tests/test_gpu_ref.py).  TEST INFRASTRUCTURE: it stands where the ROS master stood — answers
// fill_config's parameter queries (main.cpp:133-175), keeps the two callbacks main() subscribes (:32-41) by topic, and turns
// ros::ok() / ros::spinOnce() (:52, :126) into the recorded stream's feed (ref_main_feed.cpp).  Nothing publishes anywhere.
#pragma once
#include <cfloat>
#include <cstdint>
#include <functional>
#include <map>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

namespace lvmain {
struct Overrides {   // what the parameter server would answer
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
    // topic -> callback(pointer to a std::tuple of the subscribed member function's arguments, by value)
    std::map<std::string, std::function<void(const void*)>> subscribers;
};
inline Feed& feed() { static Feed f; return f; }
}  // namespace lvmain

namespace ros {
struct Subscriber {};
struct NodeHandle {
    template <typename T> bool param(const std::string& name, T& v, const T& d) {
        if (lvmain::lookup(name, v)) return true;
        v = d;
        return false;
    }
    // main.cpp:32-41 hands over `&Accumulator::receive_lidar` / `&Accumulator::receive_imu`: whatever their argument lists are
    // (the reference's take ROS message pointers, the shim's the message's payload: INTEGRATION.md), the feed calls them with a
    // tuple of exactly those arguments
    template <typename R, typename T, typename... A>
    Subscriber subscribe(const std::string& topic, uint32_t, R (T::*fp)(A...), T* obj) {
        lvmain::feed().subscribers[topic] = [fp, obj](const void* p) {
            std::apply([&](const auto&... a) { (obj->*fp)(a...); }, *static_cast<const std::tuple<std::decay_t<A>...>*>(p));
        };
        return Subscriber();
    }
};
struct Rate { Rate(double) {} void sleep() {} };
inline bool ok() { return lvmain::feed().ok ? lvmain::feed().ok() : false; }
inline void spinOnce() { if (lvmain::feed().spin) lvmain::feed().spin(); }
inline void init(int&, char**, const std::string&) {}
}  // namespace ros
