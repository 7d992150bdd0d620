// limovelo_shim.cpp — see limovelo_shim.hpp.  Data movement only; every number of the path is
// produced by liblimovelo_hip.so.
#include "limovelo_shim.hpp"

struct Params Config;

namespace {
lv_ctx* g_ctx = nullptr;
int g_device = 0;
float g_voxel = 0.5f;
int g_lanes = 8;

void check(int rc, const char* what) {
    if (rc != LV_OK) throw std::runtime_error(std::string(what) + ": " + lv_last_error());
}

// deque<Point> -> contiguous 32-byte records (the reference does the same copy: Mapper.cpp:69,74)
PointVector as_vector(const Points& points) { return PointVector(points.begin(), points.end()); }
}  // namespace

void HipRuntime::configure(int device, float voxel_size, int lanes_per_query) {
    g_device = device;
    g_voxel = voxel_size;
    g_lanes = lanes_per_query;
}

lv_ctx* HipRuntime::ctx() {
    if (!g_ctx) {
        lv_params p;
        lv_default_params(&p);
        p.MAX_NUM_ITERS = Config.MAX_NUM_ITERS;
        p.NUM_MATCH_POINTS = Config.NUM_MATCH_POINTS;
        p.MAX_DIST_PLANE = Config.MAX_DIST_PLANE;
        p.PLANES_THRESHOLD = Config.PLANES_THRESHOLD;
        p.estimate_extrinsics = Config.estimate_extrinsics ? 1 : 0;
        p.LiDAR_noise = Config.LiDAR_noise;
        for (int i = 0; i < 23; ++i) p.LIMITS[i] = i < (int)Config.LIMITS.size() ? Config.LIMITS[i] : 0.001;
        p.degeneracy_threshold = Config.degeneracy_threshold;
        p.voxel_size = g_voxel;
        p.lanes_per_query = g_lanes;
        check(lv_create(&p, g_device, &g_ctx), "lv_create");
    }
    return g_ctx;
}

void HipRuntime::shutdown() {
    if (g_ctx) lv_destroy(g_ctx);
    g_ctx = nullptr;
}

// State::State(const state_ikfom&, double) — reference src/Objects/State.cpp:51-62 (pose members only)
State::State(const state_ikfom& s, double t) {
    std::memset(this, 0, sizeof(*this));
    time = t;
    x = s;
    auto q2r = [](const double q[4], float R[9]) {  // Eigen Quaternion::toRotationMatrix, then cast<float>
        const double qx = q[0], qy = q[1], qz = q[2], qw = q[3];
        const double tx = 2 * qx, ty = 2 * qy, tz = 2 * qz;
        const double twx = tx * qw, twy = ty * qw, twz = tz * qw, txx = tx * qx, txy = ty * qx, txz = tz * qx;
        const double tyy = ty * qy, tyz = tz * qy, tzz = tz * qz;
        const double M[9] = {1 - (tyy + tzz), txy - twz, txz + twy, txy + twz, 1 - (txx + tzz), tyz - twx,
                             txz - twy,       tyz + twx, 1 - (txx + tyy)};
        for (int i = 0; i < 9; ++i) R[i] = (float)M[i];
    };
    q2r(s.rot, R);
    q2r(s.offset_R_L_I, RLI);
    for (int i = 0; i < 3; ++i) { pos[i] = (float)s.pos[i]; tLI[i] = (float)s.offset_T_L_I[i]; }
}

// ---- Mapper (reference src/Modules/Mapper.cpp) ---------------------------------------------------
bool Mapper::exists() { return size() > 0; }                             // :36-38,78-80
int Mapper::size() { return (int)lv_map_size(HipRuntime::ctx()); }       // :32-34

void Mapper::add(Points& points, double time, bool downsample) {         // :22-30
    if (points.empty()) return;
    PointVector v = as_vector(points);
    if (!exists()) check(lv_map_build(HipRuntime::ctx(), v.data(), sizeof(Point), v.size()), "lv_map_build");      // :68-71
    else check(lv_map_add(HipRuntime::ctx(), v.data(), sizeof(Point), v.size(), downsample ? 1 : 0), "lv_map_add");  // :73-76
    last_map_time = time;
}

Matches Mapper::match(const State& X, const Points& points) {            // :40-56
    Matches matches;
    if (!exists()) return matches;
    lv_ctx* c = HipRuntime::ctx();
    PointVector v = as_vector(points);
    check(lv_scan_set(c, v.data(), sizeof(Point), v.size()), "lv_scan_set");
    lv_sums sums;
    check(lv_iterate(c, &X.x, &sums), "lv_iterate");
    const size_t n = v.size();
    std::vector<uint8_t> valid(n);
    std::vector<float> pw(3 * n), abcd(4 * n), dist(n);
    check(lv_fetch_matches(c, valid.data(), pw.data(), abcd.data(), dist.data()), "lv_fetch_matches");
    matches.reserve((size_t)sums.n_valid);
    for (size_t i = 0; i < n; ++i) {  // deterministic scan order (the reference's push_back order is racy)
        if (!valid[i]) continue;
        Match m;
        m.point = v[i];
        m.point.x = pw[3 * i]; m.point.y = pw[3 * i + 1]; m.point.z = pw[3 * i + 2];
        m.plane.is_plane = true;
        m.plane.n.A = abcd[4 * i]; m.plane.n.B = abcd[4 * i + 1]; m.plane.n.C = abcd[4 * i + 2]; m.plane.n.D = abcd[4 * i + 3];
        m.distance = dist[i];
        matches.push_back(m);
    }
    return matches;
}

bool Mapper::hasToMap(double t) {                                        // :58-61
    if (last_map_time < 0) last_map_time = t;
    return t - last_map_time >= Config.full_rotation_time;
}

// ---- Localizator (reference src/Modules/Localizator.cpp) ------------------------------------------
Localizator::Localizator() {
    std::memset(&x_, 0, sizeof(x_));
    x_.rot[3] = 1.0;
    x_.offset_R_L_I[3] = 1.0;
    x_.grav[2] = -9.809;
    for (int i = 0; i < 23 * 23; ++i) P_[i] = (i / 23 == i % 23) ? 1.0 : 0.0;
}

void Localizator::pull() {
    if (!host_stale_) return;
    check(lv_filter_get(HipRuntime::ctx(), &x_, P_), "lv_filter_get");
    host_stale_ = false;
}
void Localizator::push() { check(lv_filter_set(HipRuntime::ctx(), &x_, P_), "lv_filter_set"); host_stale_ = false; }
const state_ikfom& Localizator::get_x() { pull(); return x_; }
const double* Localizator::get_P() { pull(); return P_; }
void Localizator::change_x(const state_ikfom& x) { pull(); x_ = x; push(); }
void Localizator::change_P(const double* P) { pull(); std::memcpy(P_, P, sizeof(P_)); push(); }

void Localizator::propagate(const IMU& imu) {                            // :159-173
    double Q[144] = {0};
    for (int i = 0; i < 3; ++i) {
        Q[(0 + i) * 12 + 0 + i] = Config.cov_gyro;
        Q[(3 + i) * 12 + 3 + i] = Config.cov_acc;
        Q[(6 + i) * 12 + 6 + i] = Config.cov_bias_gyro;
        Q[(9 + i) * 12 + 9 + i] = Config.cov_bias_acc;
    }
    const double acc[3] = {imu.a[0], imu.a[1], imu.a[2]}, gyro[3] = {imu.w[0], imu.w[1], imu.w[2]};
    const double dt = imu.time - last_time_integrated;
    check(lv_predict(HipRuntime::ctx(), dt, Q, acc, gyro), "lv_predict");
    host_stale_ = true;
}

void Localizator::propagate_to(const IMUs& imus, double t) {             // :59-75
    if (last_time_integrated < 0) last_time_integrated = t;
    for (const IMU& imu : imus) {
        propagate(imu);
        last_time_integrated = imu.time;
    }
    if (!imus.empty()) {
        IMU last(imus.back().a, imus.back().w, t);
        propagate(last);
        last_time_integrated = t;
    }
}

void Localizator::init_state(const state_ikfom& x0) {                    // :135-153
    x_ = x0;
    for (int i = 0; i < 23 * 23; ++i) P_[i] = (i / 23 == i % 23) ? 1.0 : 0.0;
    for (int i : {6, 7, 8, 9, 10, 11}) P_[i * 23 + i] = 0.00001;
    for (int i : {15, 16, 17}) P_[i * 23 + i] = 0.0001;
    for (int i : {18, 19, 20}) P_[i * 23 + i] = 0.001;
    for (int i : {21, 22}) P_[i * 23 + i] = 0.00001;
    push();
    initialized = true;
}

void Localizator::correct(const Points& points, double time) {           // :23-27
    if (!Mapper::getInstance().exists()) return;
    if (!initialized) { push(); initialized = true; }
    IKFoM_update(points);
    last_time_updated = time;
}

void Localizator::IKFoM_update(const Points& points) {                   // :129-133
    points2match = points;
    lv_ctx* c = HipRuntime::ctx();
    PointVector v = as_vector(points);
    check(lv_scan_set(c, v.data(), sizeof(Point), v.size()), "lv_scan_set");
    check(lv_correct(c, &last_passes), "lv_correct");  // update_iterated_dyn_share_modified :132, on the resident state
    host_stale_ = true;
}

void Localizator::calculate_H(const state_ikfom& s, const Matches& matches, MatrixXd& H, VectorXd& h) {  // :29-57
    const size_t n = matches.size();
    H.resize((int)n, 12);
    h.resize((int)n);
    if (n == 0) return;
    std::vector<float> pw(3 * n), abcd(4 * n), dist(n);
    for (size_t i = 0; i < n; ++i) {
        pw[3 * i] = matches[i].point.x; pw[3 * i + 1] = matches[i].point.y; pw[3 * i + 2] = matches[i].point.z;
        abcd[4 * i] = matches[i].plane.n.A; abcd[4 * i + 1] = matches[i].plane.n.B;
        abcd[4 * i + 2] = matches[i].plane.n.C; abcd[4 * i + 3] = matches[i].plane.n.D;
        dist[i] = matches[i].distance;
    }
    check(lv_calculate_H(HipRuntime::ctx(), &s, pw.data(), abcd.data(), dist.data(), n, H.d.data(), h.d.data()), "lv_calculate_H");
}

State Localizator::latest_state() {                                      // :77-97
    pull();
    if (last_time_updated < 0) return State(x_, last_time_integrated);
    return State(x_, last_time_updated);
}
