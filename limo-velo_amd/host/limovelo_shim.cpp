// limovelo_shim.cpp — see limovelo_shim.hpp.  Bulk computation is in liblimovelo_hip.so; the host keeps the
// reference's bookkeeping (buffers, single-state motion model, single transforms).
#include "../csrc/lv_sincos.hpp"
#include "limovelo_shim.hpp"

// (`Params Config` — the reference's global, src/main.cpp:14 — is NOT defined in this library: the program defines it, as the
// reference's main.cpp does; programs that do not name it link liblimovelo_shim_config.a, which holds a default-constructed one
// (limovelo_config.cpp).  A definition in here, however weak, would be constructed and destroyed a second time at the address
// of the program's own.)

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

// State::State() — reference src/Objects/State.cpp:19-39
State::State() {
    std::memset(this, 0, sizeof(*this));
    for (int i = 0; i < 3; ++i) {
        g[i] = i < (int)Config.initial_gravity.size() ? Config.initial_gravity[i] : 0.f;
        tLI[i] = i < (int)Config.I_Translation_L.size() ? Config.I_Translation_L[i] : 0.f;
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) RLI[i * 3 + j] = Config.I_Rotation_L.size() == 9 ? Config.I_Rotation_L[i * 3 + j] : (i == j);
    R[0] = R[4] = R[8] = 1.f;
    x.rot[3] = 1.0;
    x.offset_R_L_I[3] = 1.0;
}

// State::State(const state_ikfom&, double) — reference src/Objects/State.cpp:41-62 (the controls come from the IMU that
// follows t in the Accumulator, :46-49)
State::State(const state_ikfom& s, double t) : State() {
    time = t;
    x = s;
    {
        const IMU imu = Accumulator::getInstance().get_next_imu(t);
        for (int i = 0; i < 3; ++i) { a[i] = imu.a[i]; w[i] = imu.w[i]; }
    }
    for (int i = 0; i < 3; ++i) { vel[i] = (float)s.vel[i]; bw[i] = (float)s.bg[i]; ba[i] = (float)s.ba[i]; }
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

namespace {
inline float dot3(float a0, float b0, float a1, float b1, float a2, float b2) { return a0 * b0 + (a1 * b1 + a2 * b2); }
using lv::sincos_f32;   // ../csrc/lv_sincos.hpp: the one definition the device kernels use too
}  // namespace

// State::update / propagate_f — reference src/Objects/State.cpp:94-121 (SO3Math::Exp: Utils.hpp:30-53)
void State::operator+=(const IMU& imu) {
    const float dt = (float)(imu.time - time);
    const float wm[3] = {imu.w[0] - bw[0], imu.w[1] - bw[1], imu.w[2] - bw[2]};
    float E[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    const float nrm = std::sqrt(dot3(wm[0], wm[0], wm[1], wm[1], wm[2], wm[2]));
    if ((double)nrm > 0.0000001) {
        const float r[3] = {wm[0] / nrm, wm[1] / nrm, wm[2] / nrm};
        const float K[9] = {0.f, -r[2], r[1], r[2], 0.f, -r[0], -r[1], r[0], 0.f};
        float sn, cs;
        sincos_f32(nrm * dt, sn, cs);
        const float c = (float)(1.0 - (double)cs);
        float cK[9], cKK[9];
        for (int i = 0; i < 9; ++i) cK[i] = c * K[i];
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) cKK[i * 3 + j] = dot3(cK[i * 3], K[j], cK[i * 3 + 1], K[3 + j], cK[i * 3 + 2], K[6 + j]);
        for (int i = 0; i < 9; ++i) E[i] = (E[i] + sn * K[i]) + cKK[i];
    }
    float Rn[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = dot3(R[i * 3], E[j], R[i * 3 + 1], E[3 + j], R[i * 3 + 2], E[6 + j]);
    const float am[3] = {imu.a[0] - ba[0], imu.a[1] - ba[1], imu.a[2] - ba[2]};
    float veln[3], posn[3];
    for (int i = 0; i < 3; ++i) {
        const float v = dot3(R[i * 3], am[0], R[i * 3 + 1], am[1], R[i * 3 + 2], am[2]) - g[i];
        veln[i] = vel[i] + v * dt;
        posn[i] = pos[i] + (vel[i] * dt + ((0.5f * v) * dt) * dt);
    }
    for (int i = 0; i < 9; ++i) R[i] = Rn[i];
    for (int i = 0; i < 3; ++i) { vel[i] = veln[i]; pos[i] = posn[i]; }
    time = imu.time;
    for (int i = 0; i < 3; ++i) { a[i] = 0.5f * a[i] + 0.5f * imu.a[i]; w[i] = 0.5f * w[i] + 0.5f * imu.w[i]; }
}

lv_motion_state State::motion() const {
    lv_motion_state m;
    std::memset(&m, 0, sizeof(m));
    std::memcpy(m.R, R, sizeof(m.R)); std::memcpy(m.pos, pos, sizeof(m.pos)); std::memcpy(m.vel, vel, sizeof(m.vel));
    std::memcpy(m.bw, bw, sizeof(m.bw)); std::memcpy(m.ba, ba, sizeof(m.ba)); std::memcpy(m.g, g, sizeof(m.g));
    std::memcpy(m.RLI, RLI, sizeof(m.RLI)); std::memcpy(m.tLI, tLI, sizeof(m.tLI));
    std::memcpy(m.a, a, sizeof(m.a)); std::memcpy(m.w, w, sizeof(m.w));
    m.time = time;
    return m;
}

// ---- RotTransl (reference src/Objects/RotTransl.cpp:19-54), f32, Eigen's 3-term order x0 + (x1 + x2) ----------
RotTransl::RotTransl(const State& S) { std::memcpy(R, S.R, sizeof(R)); std::memcpy(t, S.pos, sizeof(t)); }
RotTransl RotTransl::inv() const {
    RotTransl o;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = R[j * 3 + i];
    for (int i = 0; i < 3; ++i) o.t[i] = dot3(-o.R[i * 3], t[0], -o.R[i * 3 + 1], t[1], -o.R[i * 3 + 2], t[2]);
    return o;
}
RotTransl operator*(const RotTransl& a, const RotTransl& b) {
    RotTransl o;
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = dot3(a.R[i * 3], b.R[j], a.R[i * 3 + 1], b.R[3 + j], a.R[i * 3 + 2], b.R[6 + j]);
    for (int i = 0; i < 3; ++i) o.t[i] = dot3(a.R[i * 3], b.t[0], a.R[i * 3 + 1], b.t[1], a.R[i * 3 + 2], b.t[2]) + a.t[i];
    return o;
}
Point operator*(const RotTransl& a, const Point& p) {
    Point o = p;   // attributes copied (Point.cpp:32-35)
    o.x = dot3(a.R[0], p.x, a.R[1], p.y, a.R[2], p.z) + a.t[0];
    o.y = dot3(a.R[3], p.x, a.R[4], p.y, a.R[5], p.z) + a.t[1];
    o.z = dot3(a.R[6], p.x, a.R[7], p.y, a.R[8], p.z) + a.t[2];
    return o;
}
Points operator*(const RotTransl& a, const Points& pts) {
    Points moved = pts;
    for (Point& p : moved) p = a * p;
    return moved;
}

// ---- Accumulator (reference src/Modules/Accumulator.cpp; include/Headers/Accumulator.hpp:62-118) ----------------
namespace {
// index of the newest content with time <= t in a new -> old deque (Algorithms::binary_search(content, t, true))
template <typename T>
int before_t(const std::deque<T>& c, double t) {
    int lo = 0, hi = (int)c.size();
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (c[(size_t)mid].time > t) lo = mid + 1; else hi = mid;
    }
    return lo;   // == size() when every content is newer than t
}
// number of contents with time >= t at the front of a new -> old deque (= index of the first one strictly older than t)
template <typename T>
int before_t_inclusive(const std::deque<T>& c, double t) {
    int lo = 0, hi = (int)c.size();
    while (lo < hi) {
        const int mid = (lo + hi) / 2;
        if (c[(size_t)mid].time >= t) lo = mid + 1; else hi = mid;
    }
    return lo;
}
template <typename T>
std::deque<T> get_between(Buffer<T>& src, double t1, double t2) {   // Accumulator.hpp:62-74: t1 <= time <= t2, old -> new
    std::deque<T> result;
    for (int k = before_t(src.content, t2); k < (int)src.content.size(); ++k) {
        const T& cnt = src.content[(size_t)k];
        if (t1 > cnt.time) break;
        if (t2 >= cnt.time) result.push_front(cnt);
    }
    return result;
}
}  // namespace

void Accumulator::add(State cnt, double time) { if (time > 0) cnt.time = time; BUFFER_X.push(cnt); }
void Accumulator::add(IMU cnt, double time) { if (time > 0) cnt.time = time; BUFFER_I.push(cnt); }
size_t Accumulator::receive_lidar(const void* data, size_t n_points, const lv_cloud_format& format, uint64_t header_stamp_usec) {
    return LidarBuffer::getInstance().process(data, n_points, format, header_stamp_usec);
}
void Accumulator::clear_buffers() { LidarBuffer::getInstance().clear_lidar(1e300); BUFFER_I.clear(); }
void Accumulator::clear_buffers(TimeType t) { LidarBuffer::getInstance().clear_lidar(t); BUFFER_I.clear(t); }
void Accumulator::clear_lidar(TimeType t) { LidarBuffer::getInstance().clear_lidar(t); }
Points Accumulator::get_points(double t1, double t2) { return LidarBuffer::getInstance().get_points(t1, t2); }
IMUs Accumulator::get_imus(double t1, double t2) { return get_between(BUFFER_I, t1, t2); }
States Accumulator::get_states(double t1, double t2) { return get_between(BUFFER_X, t1, t2); }

IMU Accumulator::get_next_imu(double t) {                               // Accumulator.hpp:76-92
    const std::deque<IMU>& c = BUFFER_I.content;
    if (c.empty()) return IMU();
    if (c.back().time > t) return IMU();
    if (t > c.front().time) return c.front();
    const int k = before_t(c, t);                 // newest content with time <= t; the one before it in the deque follows t
    return k > 0 ? c[(size_t)k - 1] : c.front();
}
State Accumulator::get_prev_state(double t) {                           // Accumulator.cpp:77-86, Accumulator.hpp:94-107
    if (BUFFER_X.empty()) {
        State X = Localizator::getInstance().latest_state();
        add(X, t);
        X.time = t;
        return X;
    }
    // The reference's index arithmetic, quirks included (Accumulator.hpp:94-107 over Algorithms::binary_search, Utils.hpp:9-23):
    // with F = number of buffered states at or after t (new -> old deque), binary_search returns max(0, F - 2), get_prev starts at
    // that index + 1 (clamped to the oldest) and walks towards NEWER states until one lies strictly before t.  Hence: one state
    // at or after t -> the newest state before t; NO state at or after t (every state is older: a cycle skipped for too few
    // points, or real time ahead of the last update) -> the SECOND newest; two or more at or after t -> a default State().
    // Compensator::path starts from this state, so the de-skewed points follow the reference bit for bit only with the same choice.
    const std::deque<State>& c = BUFFER_X.content;
    const int F = before_t_inclusive(c, t);
    int k_t = (F - 2 > 0 ? F - 2 : 0) + 1;
    if (k_t >= (int)c.size()) k_t = (int)c.size() - 1;
    for (int k = k_t; k >= 0; --k)
        if (t > c[(size_t)k].time) return c[(size_t)k];
    return State();
}
bool Accumulator::enough_imus() { return BUFFER_I.size() > 2 * Config.real_time_delay * Config.imu_rate + 10; }   // :156-158
void Accumulator::set_initial_time() {                                  // :160-165
    if (BUFFER_I.size() < 1) return;
    initial_time = BUFFER_I.front().time - Config.real_time_delay;
}
bool Accumulator::ready() {                                             // :102-114
    if (is_ready) return true;
    if (enough_imus()) {
        set_initial_time();
        Localizator::getInstance().initialize(initial_time);
        return is_ready = true;
    }
    return is_ready = false;
}
double Accumulator::update_delta(const InitializationParams& ini, double t) {   // :123-126, 167-178
    if (ini.times.empty()) return ini.deltas.back();
    if (t - initial_time >= ini.times.back()) return ini.deltas.back();
    for (size_t k = 0; k < ini.times.size(); ++k)
        if (t - initial_time < ini.times[k]) return ini.deltas[k];
    return ini.deltas.back();
}
double Accumulator::latest_time() { return BUFFER_I.front().time - Config.real_time_delay; }   // :128-134

// ---- Compensator (reference src/Modules/Compensator.cpp) -----------------------------------------------------
namespace {
// Compensator::upsample (Compensator.cpp:63-95): every state, then the states integrated IMU by IMU up to the next one
States upsample(const States& states, const IMUs& imus) {
    size_t s = 0, u = 0;
    States up;
    State cur = states[0];
    while (s + 1 < states.size()) {
        up.push_back(states[s]);
        while (u < imus.size() && imus[u].time < states[s + 1].time) {
            cur += imus[u++];
            up.push_back(cur);
        }
        cur = states[s++];
    }
    if (u >= imus.size()) u = imus.size() - 1;
    up.push_back(states.back());
    cur = states.back();
    while (cur.time < imus.back().time && u < imus.size()) {
        cur += imus[u++];
        up.push_back(cur);
    }
    return up;
}
// Compensator::get_t2 (Compensator.cpp:52-61)
State get_t2(const States& states, double t2) {
    int s = (int)states.size() - 1;
    while (s > 0 && t2 < states[(size_t)s].time) --s;
    State Xt2 = states[(size_t)s];
    Xt2 += IMU(Xt2.a, Xt2.w, t2);
    return Xt2;
}
std::vector<lv_motion_state> motions(const States& states) {
    std::vector<lv_motion_state> ms;
    ms.reserve(states.size());
    for (const State& st : states) ms.push_back(st.motion());
    return ms;
}
Points fetch_scan(lv_ctx* c, double time) {
    Points out;
    const size_t n = lv_scan_size(c);
    std::vector<float> xyz(3 * n);
    check(lv_scan_fetch(c, xyz.data(), n), "lv_scan_fetch");
    for (size_t i = 0; i < n; ++i) out.push_back(Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], time));
    return out;
}
}  // namespace

States Compensator::path(double t1, double t2) {                        // Compensator.cpp:37-50
    Accumulator& accum = Accumulator::getInstance();
    States states = accum.get_states(t1, t2);
    states.push_front(accum.get_prev_state(t1));
    IMUs imus = accum.get_imus(states.front().time, t2);
    imus.push_back(accum.get_next_imu(t2));
    return upsample(states, imus);
}

Points Compensator::compensate(double t1, double t2) {                  // Compensator.cpp:18-35
    lv_ctx* c = HipRuntime::ctx();
    States path_taken = path(t1, t2);
    if (path_taken.size() < 2) return Points();
    const State Xt2 = get_t2(path_taken, t2);
    const std::vector<lv_motion_state> ms = motions(path_taken);
    const lv_motion_state x2 = Xt2.motion();
    size_t nw = 0;
    check(lv_scan_deskew_window(c, t1, t2, ms.data(), ms.size(), &x2, 0.f, &nw), "lv_scan_deskew_window");
    if (nw == 0) return Points();                                        // :24
    return fetch_scan(c, t2);
}

size_t Compensator::compensate_downsample_on_device(double t1, double t2) {
    lv_ctx* c = HipRuntime::ctx();
    States path_taken = path(t1, t2);
    if (path_taken.size() < 2) return 0;
    const State Xt2 = get_t2(path_taken, t2);
    const std::vector<lv_motion_state> ms = motions(path_taken);
    const lv_motion_state x2 = Xt2.motion();
    size_t nw = 0;
    check(lv_scan_deskew_window(c, t1, t2, ms.data(), ms.size(), &x2, Config.downsample_prec, &nw), "lv_scan_deskew_window");
    return nw == 0 ? 0 : lv_scan_size(c);
}

Points Compensator::downsample(const Points& points) {                  // Compensator.cpp:104-107, 148-163
    if (points.empty()) return Points();
    lv_ctx* c = HipRuntime::ctx();
    PointVector v = as_vector(points);
    check(lv_scan_downsample(c, v.data(), sizeof(Point), v.size(), Config.downsample_prec), "lv_scan_downsample");
    return fetch_scan(c, points.back().time);
}

// ---- Compensator, explicit-path overloads (reference src/Modules/Compensator.cpp:123-163) ---------------------------
Points Compensator::compensate(const States& states, const State& Xt2, const Points& points, float downsample_prec) {
    Points out;
    if (points.empty() || states.size() < 2) return out;
    lv_ctx* c = HipRuntime::ctx();
    PointVector v = as_vector(points);
    std::vector<lv_motion_state> ms;
    ms.reserve(states.size());
    for (const State& s : states) ms.push_back(s.motion());
    const lv_motion_state x2 = Xt2.motion();
    check(lv_scan_deskew(c, v.data(), sizeof(Point), offsetof(Point, time), v.size(), ms.data(), ms.size(), &x2, downsample_prec),
          "lv_scan_deskew");
    const size_t n = lv_scan_size(c);
    std::vector<float> xyz(3 * n);
    check(lv_scan_fetch(c, xyz.data(), n), "lv_scan_fetch");
    for (size_t i = 0; i < n; ++i) out.push_back(Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], Xt2.time));
    return out;
}

Points Compensator::compensate(const States& states, const State& Xt2, double t1, double t2, float downsample_prec) {
    Points out;
    if (states.size() < 2) return out;
    lv_ctx* c = HipRuntime::ctx();
    std::vector<lv_motion_state> ms;
    ms.reserve(states.size());
    for (const State& s : states) ms.push_back(s.motion());
    const lv_motion_state x2 = Xt2.motion();
    size_t nw = 0;
    check(lv_scan_deskew_window(c, t1, t2, ms.data(), ms.size(), &x2, downsample_prec, &nw), "lv_scan_deskew_window");
    if (nw == 0) return out;   // Compensator.cpp:24
    const size_t n = lv_scan_size(c);
    std::vector<float> xyz(3 * n);
    check(lv_scan_fetch(c, xyz.data(), n), "lv_scan_fetch");
    for (size_t i = 0; i < n; ++i) out.push_back(Point(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], Xt2.time));
    return out;
}

// ---- LiDAR buffer (reference src/Modules/Accumulator.cpp, src/Utils/PointCloudProcessor.cpp) ----------------
size_t LidarBuffer::process(const void* data, size_t n_points, const lv_cloud_format& format, uint64_t header_stamp_usec) {
    lv_ingest_params p;
    std::memset(&p, 0, sizeof(p));
    p.header_stamp_usec = header_stamp_usec;
    p.stamp_beginning = Config.stamp_beginning ? 1 : 0;
    p.offset_beginning = Config.offset_beginning ? 1 : 0;
    p.full_rotation_time = Config.full_rotation_time;
    p.downsample_rate = Config.downsample_rate;
    p.min_dist = Config.min_dist;
    size_t kept = 0;
    check(lv_cloud_ingest(HipRuntime::ctx(), data, n_points, &format, &p, &kept), "lv_cloud_ingest");
    return kept;
}
Points LidarBuffer::get_points(double t1, double t2) {
    lv_ctx* c = HipRuntime::ctx();
    std::vector<Point> v(lv_cloud_size(c) + 1);
    size_t n = 0;
    check(lv_cloud_fetch(c, t1, t2, v.data(), v.size(), &n), "lv_cloud_fetch");
    return Points(v.begin(), v.begin() + (std::ptrdiff_t)n);
}
void LidarBuffer::clear_lidar(double t) { check(lv_cloud_clear(HipRuntime::ctx(), t), "lv_cloud_clear"); }
size_t LidarBuffer::size() { return lv_cloud_size(HipRuntime::ctx()); }

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

void Mapper::add_current_scan(double time, bool downsample) {
    check(lv_map_add_scan(HipRuntime::ctx(), downsample ? 1 : 0), "lv_map_add_scan");   // builds the map if there is none (Mapper.cpp:26)
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
    if (passes_stale_) { last_passes = lv_last_passes(HipRuntime::ctx()); passes_stale_ = false; }
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

void Localizator::propagate_to(double t) {                               // :59-75
    propagate_to(Accumulator::getInstance().get_imus(last_time_integrated, t), t);
}

void Localizator::initialize(double t) {                                 // :119-127 -> init_IKFoM_state :135-153
    IMUs imus = Accumulator::getInstance().get_imus(-1, t);
    if (imus.empty()) return;
    const IMU& imu = imus.back();
    state_ikfom x0;
    std::memset(&x0, 0, sizeof(x0));
    for (int i = 0; i < 4; ++i) x0.rot[i] = imu.q[i];                    // init_state.rot = imu.q
    for (int i = 0; i < 3; ++i) x0.grav[i] = -(double)(i < (int)Config.initial_gravity.size() ? Config.initial_gravity[i] : 0.f);
    // offset_R_L_I = SO3(Map<Matrix3f>(I_Rotation_L)) — column-major Map, no transpose: matrix -> quaternion
    float M[9];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) M[i * 3 + j] = Config.I_Rotation_L.size() == 9 ? Config.I_Rotation_L[(size_t)(j * 3 + i)] : (i == j);
    {
        const double tr = (double)M[0] + M[4] + M[8];
        double q[4];
        if (tr > 0) {
            const double sq = std::sqrt(tr + 1.0) * 2.0;
            q[3] = 0.25 * sq; q[0] = (M[7] - M[5]) / sq; q[1] = (M[2] - M[6]) / sq; q[2] = (M[3] - M[1]) / sq;
        } else if (M[0] > M[4] && M[0] > M[8]) {
            const double sq = std::sqrt(1.0 + M[0] - M[4] - M[8]) * 2.0;
            q[3] = (M[7] - M[5]) / sq; q[0] = 0.25 * sq; q[1] = (M[1] + M[3]) / sq; q[2] = (M[2] + M[6]) / sq;
        } else if (M[4] > M[8]) {
            const double sq = std::sqrt(1.0 + M[4] - M[0] - M[8]) * 2.0;
            q[3] = (M[2] - M[6]) / sq; q[0] = (M[1] + M[3]) / sq; q[1] = 0.25 * sq; q[2] = (M[5] + M[7]) / sq;
        } else {
            const double sq = std::sqrt(1.0 + M[8] - M[0] - M[4]) * 2.0;
            q[3] = (M[3] - M[1]) / sq; q[0] = (M[2] + M[6]) / sq; q[1] = (M[5] + M[7]) / sq; q[2] = 0.25 * sq;
        }
        for (int i = 0; i < 4; ++i) x0.offset_R_L_I[i] = q[i];
    }
    for (int i = 0; i < 3; ++i) x0.offset_T_L_I[i] = i < (int)Config.I_Translation_L.size() ? Config.I_Translation_L[(size_t)i] : 0.f;
    init_state(x0);
}

void Localizator::correct_current_scan(double time) {                    // :23-27 on the device-resident scan
    if (!Mapper::getInstance().exists()) return;
    if (!initialized) { push(); initialized = true; }
    check(lv_correct(HipRuntime::ctx(), nullptr), "lv_correct");   // (no wait here: pull() fetches the state AND the pass count)
    passes_stale_ = true;
    host_stale_ = true;
    last_time_updated = time;
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
