// ref_glue.cpp — C interface over the REFERENCE's own classes, compiled from /root/reference/src in place (oracle/ref_build/
// Makefile -> oracle/_ref/liblvref.so).  TEST INFRASTRUCTURE: only tests/ may load the library; it cannot travel to the GPU box
// as source (the reference is not there) — the built .so does.
//
// Every entry point here CALLS reference code: State(const state_ikfom&, double) (src/Objects/State.cpp:51-62), the operators of
// RotTransl / State (RotTransl.cpp:36-48, State.cpp:79-89), Mapper::add / match (src/Modules/Mapper.cpp:22-56), Plane
// (src/Objects/Plane.cpp), R3Math (src/Utils/Utils.cpp), Match, Localizator::calculate_H / correct / initialize / propagate_to
// (src/Modules/Localizator.cpp), State::operator+= (State.cpp:94-121), Compensator::compensate (src/Modules/Compensator.cpp:
// 123-146), PointCloudProcessor (src/Utils/PointCloudProcessor.cpp), Accumulator (src/Modules/Accumulator.cpp).  What is NOT the
// reference's: Eigen (stubs/Eigen/Dense), the two absent submodules (slam/), ROS / PCL types (stubs/), and this file's
// IKFoM::h_share_model — the glue between esekf and the reference's match + calculate_H, which lives in the absent
// use-ikfom.cpp [UPSTREAM-RECALL].
#ifndef __OBJECTS_H__
#define __OBJECTS_H__
#include "Headers/Common.hpp"
#include "Headers/Utils.hpp"
#include "Headers/Objects.hpp"
#include "Headers/Publishers.hpp"
#include "Headers/PointClouds.hpp"
#include "Headers/Accumulator.hpp"
#include "Headers/Compensator.hpp"
#include "Headers/Localizator.hpp"
#include "Headers/Mapper.hpp"
#endif
#include "lvref.h"

#ifndef LVREF_WITH_MAIN
Params Config;   // (src/main.cpp:15 defines it in the reference's executable; main.cpp is not part of the library build)
#endif
namespace lvref { uint32_t last_points2match = 0; }   // Localizator::points2match.size() of every update (replay output)

// ---- use-ikfom.cpp stand-in ------------------------------------------------------------------------------------------------
namespace IKFoM {
Eigen::Matrix<double, 24, 1> get_f(state_ikfom&, const input_ikfom&) { return Eigen::Matrix<double, 24, 1>::Zero(); }
Eigen::Matrix<double, 24, 23> df_dx(state_ikfom&, const input_ikfom&) { return Eigen::Matrix<double, 24, 23>::Zero(); }
Eigen::Matrix<double, 24, 12> df_dw(state_ikfom&, const input_ikfom&) { return Eigen::Matrix<double, 24, 12>::Zero(); }
// [UPSTREAM-RECALL LIMO-Velo use-ikfom.cpp]: match the scan kept by Localizator::IKFoM_update (Localizator.cpp:131) against
// the map at the current iterate, fill h_x / h through Localizator::calculate_H; no matches -> valid = false
void h_share_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& ekfom_data) {
    Localizator& KF = Localizator::getInstance();
    Mapper& MAP = Mapper::getInstance();
    Matches matches = MAP.match(State(s, 0.), KF.points2match);
    lvref::last_points2match = (uint32_t)KF.points2match.size();
    if (const char* pre = getenv("LV_DEMO_DUMP_PREFIX")) {   // (diagnostic: the scans of the first updates, first pass of each)
        static int k = 0;
        static const void* last_scan = nullptr;
        static double last_t = -1;
        const double tnow = KF.points2match.empty() ? -2.0 : (double)KF.points2match.front().x * 1e3 + (double)KF.points2match.back().y + (double)KF.points2match.size();
        if (k < 6 && tnow != last_t) {
            last_t = tnow;
            FILE* fd = fopen((std::string(pre) + "_" + std::to_string(k++) + ".bin").c_str(), "wb");
            if (fd) { for (const Point& q : KF.points2match) { const float v[3] = {q.x, q.y, q.z}; fwrite(v, 4, 3, fd); } fclose(fd); }
        }
        (void)last_scan;
    }
    if (matches.empty()) { ekfom_data.valid = false; return; }
    KF.calculate_H(s, matches, ekfom_data.h_x, ekfom_data.h);
    ekfom_data.valid = true;
}
}  // namespace IKFoM

namespace {
typedef esekfom::esekf<state_ikfom, 12, input_ikfom> Kf;
state_ikfom make_state(const double x[26]) {
    lvo_state o;
    std::memcpy(&o, x, sizeof(o));
    state_ikfom s;
    lvref::from_oracle(o, s);
    return s;
}
void put_state(const state_ikfom& s, double x[26]) {
    lvo_state o;
    lvref::to_oracle(s, o);
    std::memcpy(x, &o, sizeof(o));
}
Points make_points(const float* xyz, size_t n, const double* times = nullptr) {
    Points pts;
    for (size_t i = 0; i < n; ++i) {
        Point p(Eigen::Matrix<float, 3, 1>(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2]));
        p.time = times ? times[i] : (double)i;   // (the index rides in `time`: Point(p, attributes) carries it through every transform)
        p.intensity = 0.f;
        p.range = 0.f;
        pts.push_back(p);
    }
    return pts;
}
KD_TREE<Point>* the_tree() {
    Mapper::getInstance();
    auto& v = KD_TREE<Point>::instances();
    return v.empty() ? nullptr : v.front();
}
State from_motion(const lvo_motion_state& m) {
    State S;   // (State() reads Config.initial_gravity / I_Rotation_L / I_Translation_L: overwritten below)
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { S.R(i, j) = m.R[i * 3 + j]; S.RLI(i, j) = m.RLI[i * 3 + j]; }
        S.pos(i) = m.pos[i]; S.vel(i) = m.vel[i]; S.bw(i) = m.bw[i]; S.ba(i) = m.ba[i]; S.g(i) = m.g[i];
        S.tLI(i) = m.tLI[i]; S.a(i) = m.a[i]; S.w(i) = m.w[i];
    }
    S.time = m.time;
    return S;
}
void to_motion(const State& S, lvo_motion_state& m) {
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { m.R[i * 3 + j] = S.R(i, j); m.RLI[i * 3 + j] = S.RLI(i, j); }
        m.pos[i] = S.pos(i); m.vel[i] = S.vel(i); m.bw[i] = S.bw(i); m.ba[i] = S.ba(i); m.g[i] = S.g(i);
        m.tLI[i] = S.tLI(i); m.a[i] = S.a(i); m.w[i] = S.w(i);
    }
    m.time = S.time;
}
}  // namespace

extern "C" {

void lvr_set_config(const lvr_config* c) {
    Config.mapping_online = true;
    Config.real_time = false;
    Config.estimate_extrinsics = c->estimate_extrinsics != 0;
    Config.print_extrinsics = false;
    Config.initial_gravity.assign(c->initial_gravity, c->initial_gravity + 3);
    Config.I_Rotation_L.assign(c->I_Rotation_L, c->I_Rotation_L + 9);
    Config.I_Translation_L.assign(c->I_Translation_L, c->I_Translation_L + 3);
    Config.empty_lidar_time = 20.;
    Config.real_time_delay = c->real_time_delay;
    Config.full_rotation_time = c->full_rotation_time;
    Config.imu_rate = c->imu_rate;
    Config.downsample_rate = c->downsample_rate;
    Config.downsample_prec = c->downsample_prec;
    Config.high_quality_publish = false;
    Config.min_dist = c->min_dist;
    Config.LiDAR_type = c->lidar_type == 0 ? LIDAR_TYPE::Velodyne : c->lidar_type == 1 ? LIDAR_TYPE::Hesai : c->lidar_type == 2 ? LIDAR_TYPE::Ouster : LIDAR_TYPE::Custom;
    Config.offset_beginning = c->offset_beginning != 0;
    Config.stamp_beginning = c->stamp_beginning != 0;
    Config.degeneracy_threshold = c->degeneracy_threshold;
    Config.print_degeneracy_values = false;
    Config.MAX_NUM_ITERS = c->max_num_iters;
    Config.MAX_POINTS2MATCH = c->max_points2match;
    Config.LIMITS.assign(c->limits, c->limits + 23);
    Config.NUM_MATCH_POINTS = c->num_match_points;
    Config.MAX_DIST_PLANE = c->max_dist_plane;
    Config.PLANES_THRESHOLD = c->planes_threshold;
    Config.PLANES_CHOOSE_CONSTANT = 9.0f;
    Config.wx_MULTIPLIER = Config.wy_MULTIPLIER = Config.wz_MULTIPLIER = 1.;
    Config.cov_acc = c->cov_acc;
    Config.cov_gyro = c->cov_gyro;
    Config.cov_bias_acc = c->cov_bias_acc;
    Config.cov_bias_gyro = c->cov_bias_gyro;
    Config.LiDAR_noise = c->lidar_noise;
    Config.Initialization.times = {};
    Config.Initialization.deltas = {Config.full_rotation_time};
    // the Localizator singleton registered MAX_NUM_ITERS / LIMITS when it was constructed (Localizator::init_IKFoM,
    // Localizator.cpp:105-117): register them again, the same call with the same arguments
    Localizator::getInstance();
    if (Kf::last())
        Kf::last()->init_dyn_share(IKFoM::get_f, IKFoM::df_dx, IKFoM::df_dw, IKFoM::h_share_model, Config.MAX_NUM_ITERS, Config.LIMITS);
}

void lvr_reset(void) {
    if (KD_TREE<Point>* t = the_tree()) t->clear();
    Mapper::getInstance().last_map_time = -1;
    Accumulator& A = Accumulator::getInstance();
    A.BUFFER_L.clear(); A.BUFFER_I.clear(); A.BUFFER_X.clear();
    Localizator& L = Localizator::getInstance();
    L.points2match.clear();
    L.last_time_integrated = -1;
    L.last_time_updated = -1;
    L.initialized = false;
}

// State(const state_ikfom&, double): the f32 mirror (row-major R, pos, RLI, tLI = lvo_pose_f32)
void lvr_state_to_pose(const double x[26], float out[24]) {
    state_ikfom s = make_state(x);
    State S(s, 0.);
    for (int i = 0; i < 3; ++i) {
        for (int j = 0; j < 3; ++j) { out[i * 3 + j] = S.R(i, j); out[12 + i * 3 + j] = S.RLI(i, j); }
        out[9 + i] = S.pos(i);
        out[21 + i] = S.tLI(i);
    }
}

// Mapper.cpp:51: X * X.I_Rt_L() * p for every point
void lvr_transform(const double x[26], const float* scan_xyz, size_t n, float* out_xyz) {
    state_ikfom s = make_state(x);
    State X(s, 0.);
    for (size_t i = 0; i < n; ++i) {
        Point p(Eigen::Matrix<float, 3, 1>(scan_xyz[3 * i], scan_xyz[3 * i + 1], scan_xyz[3 * i + 2]));
        Point w = X * X.I_Rt_L() * p;
        out_xyz[3 * i] = w.x; out_xyz[3 * i + 1] = w.y; out_xyz[3 * i + 2] = w.z;
    }
}

void lvr_map_add(const float* xyz, size_t n, double time, int downsample) {
    Points pts = make_points(xyz, n);
    Mapper::getInstance().add(pts, time, downsample != 0);
}
size_t lvr_map_size(void) { return Mapper::getInstance().exists() ? (size_t)Mapper::getInstance().size() : 0; }
void lvr_map_fetch(float* out_xyz) {
    KD_TREE<Point>* t = the_tree();
    if (!t) return;
    size_t i = 0;
    for (const Point& p : t->points()) { out_xyz[3 * i] = p.x; out_xyz[3 * i + 1] = p.y; out_xyz[3 * i + 2] = p.z; ++i; }
}

// Mapper::match(State(s, 0.), points): the chosen matches in scan order (this build compiles the loop without OpenMP)
size_t lvr_match(const double x[26], const float* scan_xyz, size_t n, uint32_t* src_index, float* p_world, float* abcd, float* dist) {
    state_ikfom s = make_state(x);
    Points pts = make_points(scan_xyz, n);
    Matches m = Mapper::getInstance().match(State(s, 0.), pts);
    for (size_t i = 0; i < m.size(); ++i) {
        src_index[i] = (uint32_t)m[i].point.time;
        p_world[3 * i] = m[i].point.x; p_world[3 * i + 1] = m[i].point.y; p_world[3 * i + 2] = m[i].point.z;
        abcd[4 * i] = m[i].plane.n.A; abcd[4 * i + 1] = m[i].plane.n.B; abcd[4 * i + 2] = m[i].plane.n.C; abcd[4 * i + 3] = m[i].plane.n.D;
        dist[i] = m[i].distance;
    }
    return m.size();
}

// Plane(near, sq_dists): returns is_plane; abcd only when it is one
int lvr_plane(const float* near_xyz, const float* sq_dists, int found, float abcd[4]) {
    PointVector near;
    std::vector<float> sq;
    for (int j = 0; j < found; ++j) {
        near.push_back(Point(Eigen::Matrix<float, 3, 1>(near_xyz[3 * j], near_xyz[3 * j + 1], near_xyz[3 * j + 2])));
        sq.push_back(sq_dists[j]);
    }
    Plane pl(near, sq);
    if (pl.is_plane) { abcd[0] = pl.n.A; abcd[1] = pl.n.B; abcd[2] = pl.n.C; abcd[3] = pl.n.D; }
    else abcd[0] = abcd[1] = abcd[2] = abcd[3] = 0.f;
    return pl.is_plane ? 1 : 0;
}

// R3Math::estimate_plane alone (no gates): the raw QR solution, for the noise-floor comparison
void lvr_estimate_plane(const float* near_xyz, int npts, float abcd[4]) {
    PointVector near;
    for (int j = 0; j < npts; ++j) near.push_back(Point(Eigen::Matrix<float, 3, 1>(near_xyz[3 * j], near_xyz[3 * j + 1], near_xyz[3 * j + 2])));
    Eigen::Matrix<float, 4, 1> r = R3Math::estimate_plane(near);
    for (int i = 0; i < 4; ++i) abcd[i] = r(i);
}

// Localizator::calculate_H on matches rebuilt from (world point, plane): H is n x 12 row-major on return
void lvr_calculate_H(const double x[26], size_t n, const float* p_world, const float* abcd, double* H, double* h, float* dist_out) {
    state_ikfom s = make_state(x);
    Matches ms;
    for (size_t i = 0; i < n; ++i) {
        Plane pl;
        pl.is_plane = true;
        Eigen::Matrix<float, 4, 1> v;
        v << abcd[4 * i], abcd[4 * i + 1], abcd[4 * i + 2], abcd[4 * i + 3];
        pl.n = Normal(v);
        Point p(Eigen::Matrix<float, 3, 1>(p_world[3 * i], p_world[3 * i + 1], p_world[3 * i + 2]));
        ms.push_back(Match(p, pl));   // (Match::Match computes the signed distance: Match.cpp:18-22)
        if (dist_out) dist_out[i] = ms.back().distance;
    }
    Eigen::MatrixXd Hm;
    Eigen::VectorXd hv;
    Localizator::getInstance().calculate_H(s, ms, Hm, hv);
    for (size_t i = 0; i < n; ++i) {
        for (int c = 0; c < 12; ++c) H[i * 12 + c] = Hm((Eigen::Index)i, c);
        h[i] = hv((Eigen::Index)i);
    }
}

// Localizator::correct(points, time) from a given prior: x (26) and P (23 x 23 row-major) in / out
int lvr_update(double x[26], double* P, const float* scan_xyz, size_t n, lvo_iter_out* sums_log, double* state_log) {
    Localizator& L = Localizator::getInstance();
    Kf* kf = Kf::last();
    state_ikfom s = make_state(x);
    kf->change_x(s);
    Kf::cov Pm;
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) Pm(i, j) = P[i * 23 + j];
    kf->change_P(Pm);
    Points pts = make_points(scan_xyz, n);
    kf->passes = 0;   // (Localizator::correct returns before the update when there is no map: Localizator.cpp:24)
    kf->sums_log.clear();
    kf->trace_log.clear();
    L.correct(pts, 1.0);
    put_state(kf->get_x(), x);
    Pm = kf->get_P();
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) P[i * 23 + j] = Pm(i, j);
    for (int p = 0; p < kf->passes; ++p) {
        if (sums_log) sums_log[p] = kf->sums_log[(size_t)p];
        if (state_log) std::memcpy(state_log + 26 * p, &kf->trace_log[(size_t)p], sizeof(lvo_state));
    }
    return kf->passes;
}

// Localizator::initialize(t) (Localizator.cpp:119-153) after one IMU (a, w, orientation q = x, y, z, w) has been received
void lvr_initialize(const float a[3], const float w[3], const float q_xyzw[4], double t, double x[26], double* P) {
    Accumulator& A = Accumulator::getInstance();
    A.add(IMU(Eigen::Vector3f(a[0], a[1], a[2]), Eigen::Vector3f(w[0], w[1], w[2]), Eigen::Quaternionf(q_xyzw[3], q_xyzw[0], q_xyzw[1], q_xyzw[2]), t));
    Localizator& L = Localizator::getInstance();
    L.initialize(t);
    Kf* kf = Kf::last();
    put_state(kf->get_x(), x);
    Kf::cov Pm = kf->get_P();
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) P[i * 23 + j] = Pm(i, j);
}

// Localizator::propagate_to(t) (Localizator.cpp:59-75, :159-173) over IMUs pushed into the Accumulator; x / P in and out
void lvr_propagate(double x[26], double* P, double last_time_integrated, const float* imu_a, const float* imu_w, const double* imu_t, size_t n_imu, double t) {
    Accumulator& A = Accumulator::getInstance();
    A.BUFFER_I.clear();
    for (size_t i = 0; i < n_imu; ++i)
        A.add(IMU(Eigen::Vector3f(imu_a[3 * i], imu_a[3 * i + 1], imu_a[3 * i + 2]), Eigen::Vector3f(imu_w[3 * i], imu_w[3 * i + 1], imu_w[3 * i + 2]), imu_t[i]));
    Localizator& L = Localizator::getInstance();
    Kf* kf = Kf::last();
    state_ikfom s = make_state(x);
    kf->change_x(s);
    Kf::cov Pm;
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) Pm(i, j) = P[i * 23 + j];
    kf->change_P(Pm);
    L.last_time_integrated = last_time_integrated;
    L.propagate_to(t);
    put_state(kf->get_x(), x);
    Pm = kf->get_P();
    for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) P[i * 23 + j] = Pm(i, j);
}

// State::operator+=(IMU(a, w, t))
void lvr_state_integrate(lvo_motion_state* m, const float a[3], const float w[3], double t) {
    State S = from_motion(*m);
    S += IMU(Eigen::Vector3f(a[0], a[1], a[2]), Eigen::Vector3f(w[0], w[1], w[2]), t);
    to_motion(S, *m);
}

// Compensator::compensate(states, Xt2, points); returns the number of points written (the reference's walk may stop early)
size_t lvr_deskew(const float* xyz, const double* times, size_t n, const lvo_motion_state* states, size_t n_states, const lvo_motion_state* Xt2,
                  float* out_xyz) {
    States st;
    for (size_t i = 0; i < n_states; ++i) st.push_back(from_motion(states[i]));
    Points pts = make_points(xyz, n, times);
    Compensator comp;
    Points out = comp.compensate(st, from_motion(*Xt2), pts);
    for (size_t i = 0; i < out.size(); ++i) { out_xyz[3 * i] = out[i].x; out_xyz[3 * i + 1] = out[i].y; out_xyz[3 * i + 2] = out[i].z; }
    return out.size();
}

// Compensator::path(t1, t2) over states / IMUs pushed into the Accumulator (newest first inside, any order here: sorted by time)
size_t lvr_path(const lvo_motion_state* states, size_t n_states, const float* imu_a, const float* imu_w, const double* imu_t, size_t n_imu,
                double t1, double t2, lvo_motion_state* out, size_t cap) {
    Accumulator& A = Accumulator::getInstance();
    A.BUFFER_I.clear(); A.BUFFER_X.clear();
    for (size_t i = 0; i < n_imu; ++i)
        A.add(IMU(Eigen::Vector3f(imu_a[3 * i], imu_a[3 * i + 1], imu_a[3 * i + 2]), Eigen::Vector3f(imu_w[3 * i], imu_w[3 * i + 1], imu_w[3 * i + 2]), imu_t[i]));
    for (size_t i = 0; i < n_states; ++i) A.add(from_motion(states[i]));
    Compensator comp;
    States p = comp.path(t1, t2);
    for (size_t i = 0; i < p.size() && i < cap; ++i) to_motion(p[i], out[i]);
    return p.size();
}

// PointCloudProcessor: msg2points -> downsample -> sort_points (what Accumulator::process runs, Accumulator.cpp:143-153) on a
// PointCloud2 assembled from raw records and a field table
size_t lvr_cloud_ingest(const uint8_t* data, size_t n, uint32_t point_step, int nfields, const char* const* names, const uint32_t* offsets,
                        const uint8_t* datatypes, uint64_t stamp_usec, lvo_point* out) {
    boost::shared_ptr<sensor_msgs::PointCloud2> msg(new sensor_msgs::PointCloud2());
    msg->header.stamp.sec = (uint32_t)(stamp_usec / 1000000ull);
    msg->header.stamp.nsec = (uint32_t)((stamp_usec % 1000000ull) * 1000ull);
    msg->width = (uint32_t)n;
    msg->height = 1;
    msg->point_step = point_step;
    msg->row_step = point_step * (uint32_t)n;
    for (int i = 0; i < nfields; ++i) {
        sensor_msgs::PointField f;
        f.name = names[i]; f.offset = offsets[i]; f.datatype = datatypes[i]; f.count = 1;
        msg->fields.push_back(f);
    }
    msg->data.assign(data, data + (size_t)point_step * n);
    PointCloudProcessor proc;
    PointCloud_msg cmsg = msg;
    Points pts = proc.msg2points(cmsg);
    Points ds = proc.downsample(pts);
    Points sorted = proc.sort_points(ds);
    for (size_t i = 0; i < sorted.size(); ++i) {
        out[i].x = sorted[i].x; out[i].y = sorted[i].y; out[i].z = sorted[i].z; out[i].pad_ = 0.f;
        out[i].time = sorted[i].time; out[i].intensity = sorted[i].intensity; out[i].range = sorted[i].range;
    }
    return sorted.size();
}

// Accumulator::get_points / get_imus windows and Buffer::clear (Accumulator.hpp:73-127, Buffer.cpp:61-66): the times that survive
size_t lvr_buffer_window(const double* times, size_t n, double t1, double t2, double clear_t, double* out) {
    Accumulator& A = Accumulator::getInstance();
    A.BUFFER_L.clear();
    for (size_t i = 0; i < n; ++i) { Point p(Eigen::Matrix<float, 3, 1>(0.f, 0.f, 0.f)); p.time = times[i]; A.add(p); }
    if (clear_t > -1e300) A.clear_lidar(clear_t);
    Points w = A.get_points(t1, t2);
    for (size_t i = 0; i < w.size(); ++i) out[i] = w[i].time;
    return w.size();
}

}  // extern "C"
