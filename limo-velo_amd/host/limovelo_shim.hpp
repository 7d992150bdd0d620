// limovelo_shim.hpp — C++ host side above the C-ABI: the reference's `Mapper` / `Localizator`
// interface for the iterated-KF-update path, ROS/PCL/Eigen-free.
//
// Same class names, method names, argument meaning and "no map yet => silently do nothing" behaviour
// as the reference (Huguet57/LIMO-Velo):
//   Mapper        include/Headers/Mapper.hpp:8-48,       src/Modules/Mapper.cpp:18-90
//   Localizator   include/Headers/Localizator.hpp:8-54,  src/Modules/Localizator.cpp:18-179
//   Point / Normal / Plane / Match / State   include/Headers/Objects.hpp:20-190
//   Params (hot keys)                         include/Headers/Common.hpp:56-107
// so that src/main.cpp:76-102 (`loc.correct(...)`, `loc.latest_state()`, `map.add(...)`) compiles
// against it unchanged.  Differences forced by the environment (documented in INTEGRATION.md):
//   * Eigen is not available here: the two Eigen-typed signatures use the tiny row-major `MatrixXd` /
//     `VectorXd` stand-ins below (operator()(i,j), rows(), cols()); with Eigen present they are
//     drop-in replaceable by Eigen::Map views.
//   * state_ikfom is the plain lv_state record (26 doubles, quaternions x,y,z,w) instead of the MTK
//     compound manifold; IKFoM's esekf lives on the GPU (lv_update).
//   * State(const state_ikfom&, double) no longer reaches into the Accumulator singleton
//     (reference src/Objects/State.cpp:41-51): IMU-derived members are not on this path.
// Every bulk computation of the path (matching, plane fits, Jacobians, the filter algebra, de-skew, voxel grid, map
// maintenance) runs in liblimovelo_hip.so.  What stays on the host is the reference's own bookkeeping restated:
// the IMU / state buffers of the Accumulator, the single-state f32 motion model State::operator+= that builds the
// handful of states surrounding a window (Compensator::upsample), and RotTransl algebra on single transforms.
#pragma once

#include <cmath>
#include <cstddef>
#include <cstdint>
#include <cstring>
#include <deque>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/limovelo_hip.h"

typedef double TimeType;

struct InitializationParams {  // reference Common.hpp:51-54
    std::vector<double> times;
    std::vector<double> deltas = {0.1};
};

struct Params {  // reference struct Params (Common.hpp:56-107): every key src/main.cpp's fill_config reads, the reference's types
    bool estimate_extrinsics = false;
    bool print_extrinsics = false;                           // (publishers only)
    double degeneracy_threshold = 5.0;
    bool print_degeneracy_values = false;
    int MAX_NUM_ITERS = 3;
    int MAX_POINTS2MATCH = 10;
    std::vector<double> LIMITS = std::vector<double>(23, 0.001);
    int NUM_MATCH_POINTS = 5;
    double MAX_DIST_PLANE = 2.0;
    float PLANES_THRESHOLD = 5.e-2f;
    float PLANES_CHOOSE_CONSTANT = 9.0f;                     // (read by fill_config, used nowhere in the reference either)
    double LiDAR_noise = 0.001;
    double cov_acc = 1.e-2, cov_gyro = 1.e-4, cov_bias_acc = 1.e-4, cov_bias_gyro = 1.e-5;  // config/params.yaml:39-42
    double wx_MULTIPLIER = 1, wy_MULTIPLIER = 1, wz_MULTIPLIER = 1;   // (read by fill_config, used nowhere in the reference)
    double full_rotation_time = 0.1;
    bool stamp_beginning = false, offset_beginning = false;   // config/params.yaml:30-31
    int downsample_rate = 4;                                 // :35
    double min_dist = 4.;                                    // :34
    std::string LiDAR_type = "hesai";                        // (the wire format: lv_cloud_format_preset; the ROS side's switch)
    float downsample_prec = 0.5f;
    bool high_quality_publish = false;                       // (publishers only)
    std::vector<float> initial_gravity = {0.f, 0.f, -9.807f};
    std::vector<float> I_Rotation_L = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::vector<float> I_Translation_L = {0, 0, 0};
    bool real_time = false, mapping_online = true;           // config/params.yaml:2-3
    double real_time_delay = 0.1;                            // :24
    double imu_rate = 400;                                   // :33
    double empty_lidar_time = 0.2;                           // :23
    std::string points_topic = "/velodyne_points", imus_topic = "/vectornav/IMU";
    InitializationParams Initialization;                     // :59-66
};
extern struct Params Config;  // the reference's global (src/main.cpp:14)

class Point {  // reference Objects.hpp:20-28 — 32 bytes, xyz at offset 0
  public:
    float x, y, z;
    TimeType time;
    float intensity;
    float range;
    Point() : x(0), y(0), z(0), time(0), intensity(0), range(0) {}
    Point(float x_, float y_, float z_, TimeType t = 0) : x(x_), y(y_), z(z_), time(t), intensity(0), range(0) {}
};
static_assert(sizeof(Point) == 32, "Point must keep the reference's 32-byte layout");

class IMU {  // reference Objects.hpp IMU (a, w, q, time); the ROS constructors are ingest-only
  public:
    float a[3] = {0, 0, 0};
    float w[3] = {0, 0, 0};
    float q[4] = {0, 0, 0, 1};   // orientation of the message (x, y, z, w): initial attitude, Localizator.cpp:137
    TimeType time = 0;
    IMU() = default;
    IMU(const float a_[3], const float w_[3], TimeType t) : time(t) { for (int i = 0; i < 3; ++i) { a[i] = a_[i]; w[i] = w_[i]; } }
};

template <typename ContentType>
class Buffer {  // reference Objects.hpp:4-18, src/Objects/Buffer.cpp: newest content at the front
  public:
    std::deque<ContentType> content;
    void push(const ContentType& cnt) { content.push_front(cnt); }
    void pop_front() { content.pop_front(); }
    void pop_back() { content.pop_back(); }
    ContentType front() { return content.front(); }
    ContentType back() { return content.back(); }
    bool empty() { return content.empty(); }
    int size() { return (int)content.size(); }
    void clear() { content.clear(); }
    void clear(TimeType t) { while (!content.empty() && t >= content.back().time) content.pop_back(); }
};
typedef std::deque<IMU> IMUs;

typedef std::deque<Point> Points;
typedef std::vector<Point> PointVector;
typedef lv_state state_ikfom;

struct MatrixXd {  // minimal row-major stand-in for Eigen::MatrixXd on the calculate_H signature
    int r = 0, c = 0;
    std::vector<double> d;
    void resize(int rows_, int cols_) { r = rows_; c = cols_; d.assign((size_t)rows_ * cols_, 0.0); }
    int rows() const { return r; }
    int cols() const { return c; }
    double& operator()(int i, int j) { return d[(size_t)i * c + j]; }
    double operator()(int i, int j) const { return d[(size_t)i * c + j]; }
};
struct VectorXd {
    std::vector<double> d;
    void resize(int n) { d.assign((size_t)n, 0.0); }
    int size() const { return (int)d.size(); }
    double& operator()(int i) { return d[(size_t)i]; }
    double operator()(int i) const { return d[(size_t)i]; }
};

class Normal {  // reference Objects.hpp:153-162
  public:
    float A = 0, B = 0, C = 0, D = 0;
};
class Plane {  // reference Objects.hpp:164-179
  public:
    bool is_plane = false;
    Point centroid;  // visualisation only in the reference (buggy accumulator): left at 0
    Normal n;
    float dist_to_plane(const Point& p) const { return n.A * p.x + n.B * p.y + n.C * p.z + n.D; }
};
class Match {  // reference Objects.hpp:181-190
  public:
    Point point;  // world frame
    Plane plane;
    float distance = 0;
    bool is_chosen() { return plane.is_plane; }
};
typedef std::vector<Match> Matches;

class State;
class RotTransl {  // reference Objects.hpp:139-151, src/Objects/RotTransl.cpp; row-major R
  public:
    float R[9], t[3];
    RotTransl() : R{1, 0, 0, 0, 1, 0, 0, 0, 1}, t{0, 0, 0} {}
    explicit RotTransl(const State& S);
    RotTransl(const float dR[9], const float dt[3]) { std::memcpy(R, dR, sizeof(R)); std::memcpy(t, dt, sizeof(t)); }
    RotTransl inv() const;
    friend RotTransl operator*(const RotTransl&, const RotTransl&);
    friend Point operator*(const RotTransl&, const Point&);
    friend Points operator*(const RotTransl&, const Points&);
};

class State {  // f32 mirror of the filter state (reference Objects.hpp:97-137), row-major matrices
  public:
    float R[9], pos[3], vel[3], bw[3], ba[3], g[3];
    float RLI[9], tLI[3];
    TimeType time = 0;
    float a[3], w[3];  // last controls
    state_ikfom x;     // the f64 source it was built from
    State();
    State(const state_ikfom& s, double t);
    void operator+=(const IMU& imu);  // State::update -> propagate_f (State.cpp:94-121); host math for single states
    lv_motion_state motion() const;   // the record lv_scan_deskew consumes
    RotTransl I_Rt_L() const { return RotTransl(RLI, tLI); }           // State.cpp:64-69
    RotTransl inv() const { return RotTransl(*this).inv(); }           // :71-73
    friend Point operator*(const State& X, const Point& p) { return RotTransl(X) * p; }            // :79-81
    friend RotTransl operator*(const State& X, const RotTransl& RT) { return RotTransl(X) * RT; } // :83-85
    friend Points operator*(const State& X, const Points& pts) { return RotTransl(X) * pts; }      // :87-89
};
typedef std::deque<State> States;

// The reference's Accumulator (include/Headers/Accumulator.hpp, src/Modules/Accumulator.cpp) without ROS: the IMU
// and state buffers live here on the host (a few hundred small records), the LiDAR buffer on the device
// (LidarBuffer below).  Same method names and time-interval semantics (content sorted new -> old, closed intervals).
class Accumulator {
  public:
    Buffer<IMU> BUFFER_I;
    Buffer<State> BUFFER_X;
    double initial_time = 0;

    void add(State cnt, double time = -1);
    void add(IMU cnt, double time = -1);
    void receive_imu(const IMU& imu) { add(imu); }                      // Accumulator.cpp:50-55 (the IMU_msg -> IMU step is ROS)
    // Accumulator::receive_lidar (:38-48) for the payload of one sensor_msgs/PointCloud2: processed on the device
    size_t receive_lidar(const void* data, size_t n_points, const lv_cloud_format& format, uint64_t header_stamp_usec);
    void clear_buffers();
    void clear_buffers(TimeType t);
    void clear_lidar(TimeType t);
    State get_prev_state(double t);
    IMU get_next_imu(double t);
    States get_states(double t1, double t2);
    Points get_points(double t1, double t2);
    IMUs get_imus(double t1, double t2);
    bool ready();
    double update_delta(const InitializationParams&, double t);
    double latest_time();
    static Accumulator& getInstance() {
        static Accumulator* a = new Accumulator();
        return *a;
    }

  private:
    bool is_ready = false;
    bool enough_imus();
    void set_initial_time();
};

class Compensator {  // reference include/Headers/Compensator.hpp
  public:
    // Compensator::compensate(t1, t2) (Compensator.cpp:18-35): points of [t1, t2] from the (device) LiDAR buffer, states
    // from the Accumulator (path -> upsample -> get_t2), de-skewed on the device; NOT down-sampled, as in the reference
    Points compensate(double t1, double t2);
    States path(double t1, double t2);                                  // :37-50
    Points downsample(const Points& points);                            // :104-107 -> voxel grid with Config.downsample_prec, on the device
    // compensate(t1, t2) + downsample in ONE device pass, the result staying on the device as the current scan
    // (no fetch): what a maintainer calls instead of the two lines above to keep the cycle off the host; returns
    // the number of down-sampled points
    size_t compensate_downsample_on_device(double t1, double t2);
    // Compensator::compensate(states, Xt2, points) (Compensator.cpp:123-146) followed by
    // Compensator::downsample (:104-107,148-163) with leaf = downsample_prec; <= 0 skips the voxel grid
    Points compensate(const States& states, const State& Xt2, const Points& points, float downsample_prec);
    // Compensator::compensate(t1, t2) (Compensator.cpp:18-35) with the points taken from the device LiDAR buffer
    Points compensate(const States& states, const State& Xt2, double t1, double t2, float downsample_prec);
};

// LiDAR side of the reference's Accumulator (include/Headers/Accumulator.hpp): the buffer of processed, time-stamped
// points lives on the device.  process() = Accumulator::process + push of every point (Accumulator.cpp:141-153) for
// the payload of one sensor_msgs/PointCloud2 (msg->data; `format` from msg->fields);
// get_points / clear_lidar as in Accumulator.cpp:64-70,93-95.
class LidarBuffer {
  public:
    size_t process(const void* data, size_t n_points, const lv_cloud_format& format, uint64_t header_stamp_usec);
    Points get_points(double t1, double t2);
    void clear_lidar(double t);
    size_t size();
    static LidarBuffer& getInstance() {
        static LidarBuffer* b = new LidarBuffer();
        return *b;
    }
};

// One GPU context shared by the two singletons (the reference's singletons share the process).
class HipRuntime {
  public:
    static lv_ctx* ctx();
    static void configure(int device, float voxel_size = 0.5f, int lanes_per_query = 8);  // before first use
    static void shutdown();
};

class Mapper {
  public:
    double last_map_time = -1;

    bool exists();
    int size();
    void add(Points&, double time, bool downsample = false);
    // map.add(Xt2 * Xt2.I_Rt_L() * ds_compensated, t2, true) (src/main.cpp:92,102) with the scan and the posterior the
    // device already holds (lv_map_add_scan): the mapping step without a host round trip
    void add_current_scan(double time, bool downsample = true);
    Matches match(const State&, const Points&);
    bool hasToMap(double t);

    static Mapper& getInstance() {
        static Mapper* mapper = new Mapper();
        return *mapper;
    }

  private:
    Mapper() = default;
    Mapper(const Mapper&) = delete;
    Mapper& operator=(const Mapper&) = delete;
};

class Localizator {
  public:
    Points points2match;
    double last_time_integrated = -1;
    double last_time_updated = -1;
    bool initialized = false;

    // filter state: x_ and P_ of esekf<state_ikfom, 12, input_ikfom> (reference Localizator.hpp:19) live on
    // the GPU (lv_filter_set / lv_predict / lv_correct); the host copies below are refreshed on demand
    void init_state(const state_ikfom& x0);  // init_IKFoM_state's x0 / P0 (Localizator.cpp:135-153)
    const state_ikfom& get_x();
    const double* get_P();
    void change_x(const state_ikfom& x);
    void change_P(const double* P);

    void initialize(double t);               // Localizator.cpp:119-127: initial IMU from the Accumulator -> init_IKFoM_state
    void correct(const Points&, double time);
    void correct_current_scan(double time);  // the same update on the scan the device already holds (compensate_downsample_on_device)
    void propagate_to(double t);             // Localizator.cpp:59-75: the IMUs of (last_time_integrated, t] from the Accumulator
    // the same with the IMU interval handed in by the caller
    void propagate_to(const IMUs& imus, double t);
    void propagate(const IMU& imu);          // Localizator.cpp:159-173
    void calculate_H(const state_ikfom&, const Matches&, MatrixXd& H, VectorXd& h);
    State latest_state();
    int last_passes = 0;  // measurement passes of the last correct()

    static Localizator& getInstance() {
        static Localizator* localizator = new Localizator();
        return *localizator;
    }

  private:
    Localizator();
    Localizator(const Localizator&) = delete;
    Localizator& operator=(const Localizator&) = delete;
    void IKFoM_update(const Points&);
    void pull();   // device -> host copies if stale
    void push();   // host copies -> device
    state_ikfom x_;
    double P_[23 * 23];
    bool host_stale_ = false;
    bool passes_stale_ = false;   // last_passes is refreshed by the next pull() (correct_current_scan does not wait for the device)
};
