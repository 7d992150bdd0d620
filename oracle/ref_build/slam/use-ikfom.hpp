// use-ikfom.hpp — STAND-IN for the IKFoM submodule (github.com/Huguet57/IKFoM, absent from the reference mount: SURVEY F1).
// oracle/ref_build, TEST INFRASTRUCTURE.  It provides the NAMES the reference's in-tree sources use (include/Headers/
// Localizator.hpp:19-40, src/Modules/Localizator.cpp:105-178, src/Objects/State.cpp:51-61) so that those sources compile
// unmodified; the filter algebra behind them is NOT the fork's code — it is the oracle's restatement
// (oracle/lv_oracle.cpp::lvo_kf_step / lvo_predict, [UPSTREAM-RECALL IKFoM esekfom.hpp as vendored by FAST-LIO2]) reached
// through the oracle's C interface.  What a comparison against this build pins is therefore the reference's OWN code on both
// sides of the filter (State / Mapper::match / Plane / R3Math / Match / Localizator::calculate_H / the update call and the
// x0, P0, Q set-up), never the esekf internals.
#ifndef LVREF_USE_IKFOM_STUB
#define LVREF_USE_IKFOM_STUB
#include <functional>
#include <vector>
#include <Eigen/Dense>
#include "lv_oracle.h"

typedef Eigen::Vector3d vect3;   // MTK::vect<3, double>
// MTK::SO3<double>: a unit quaternion with manifold operations (only construction / conversion are needed here)
struct SO3 : public Eigen::Quaterniond {
    SO3() : Eigen::Quaterniond(1.0, 0.0, 0.0, 0.0) {}
    SO3(const Eigen::Quaterniond& q) : Eigen::Quaterniond(q) {}
    SO3(const Eigen::Matrix3d& R) : Eigen::Quaterniond(R) {}
    SO3& operator=(const Eigen::Quaterniond& q) { Eigen::Quaterniond::operator=(q); return *this; }
};
// MTK::S2<double, 98090, 10000, 1>: a vector of fixed length 9.809 [UPSTREAM-RECALL S2.hpp: normalise, scale by den / num]
struct S2 {
    Eigen::Vector3d vec;
    static double length() { return 98090.0 / 10000.0; }
    S2() : vec(length() * 1.0, 0.0, 0.0) {}
    S2(const Eigen::Vector3d& v) : vec(v) { vec.normalize(); const double l = length(); vec = vec * l; }
};

struct state_ikfom {   // field order: State.cpp:53-61, Localizator.cpp:137-150
    vect3 pos = vect3::Zero();
    SO3 rot;
    SO3 offset_R_L_I;
    vect3 offset_T_L_I = vect3::Zero();
    vect3 vel = vect3::Zero();
    vect3 bg = vect3::Zero();
    vect3 ba = vect3::Zero();
    S2 grav;
};
struct input_ikfom {
    vect3 acc = vect3::Zero();
    vect3 gyro = vect3::Zero();
};

namespace esekfom {
template <typename T>
struct dyn_share_datastruct {   // [UPSTREAM-RECALL esekfom.hpp]
    bool valid = true;
    bool converge = false;
    Eigen::Matrix<T, Eigen::Dynamic, 1> z;
    Eigen::Matrix<T, Eigen::Dynamic, 1> h;
    Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> h_v;
    Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> h_x;
    Eigen::Matrix<T, Eigen::Dynamic, Eigen::Dynamic> R;
};
}  // namespace esekfom

namespace IKFoM {
// process model: never evaluated by this stand-in (esekf::predict below goes through lvo_predict, which restates them)
Eigen::Matrix<double, 24, 1> get_f(state_ikfom& s, const input_ikfom& in);
Eigen::Matrix<double, 24, 23> df_dx(state_ikfom& s, const input_ikfom& in);
Eigen::Matrix<double, 24, 12> df_dw(state_ikfom& s, const input_ikfom& in);
// the measurement model: defined in ref_glue.cpp from the reference's own Mapper::match + Localizator::calculate_H
void h_share_model(state_ikfom& s, esekfom::dyn_share_datastruct<double>& ekfom_data);
}  // namespace IKFoM

namespace lvref {
inline void to_oracle(const state_ikfom& s, lvo_state& o) {
    for (int i = 0; i < 3; ++i) { o.pos[i] = s.pos(i); o.offset_T_L_I[i] = s.offset_T_L_I(i); o.vel[i] = s.vel(i); o.bg[i] = s.bg(i); o.ba[i] = s.ba(i); o.grav[i] = s.grav.vec(i); }
    o.rot[0] = s.rot.x(); o.rot[1] = s.rot.y(); o.rot[2] = s.rot.z(); o.rot[3] = s.rot.w();
    o.offset_R_L_I[0] = s.offset_R_L_I.x(); o.offset_R_L_I[1] = s.offset_R_L_I.y(); o.offset_R_L_I[2] = s.offset_R_L_I.z(); o.offset_R_L_I[3] = s.offset_R_L_I.w();
}
inline void from_oracle(const lvo_state& o, state_ikfom& s) {
    for (int i = 0; i < 3; ++i) { s.pos(i) = o.pos[i]; s.offset_T_L_I(i) = o.offset_T_L_I[i]; s.vel(i) = o.vel[i]; s.bg(i) = o.bg[i]; s.ba(i) = o.ba[i]; s.grav.vec(i) = o.grav[i]; }
    s.rot = Eigen::Quaterniond(o.rot[3], o.rot[0], o.rot[1], o.rot[2]);
    s.offset_R_L_I = Eigen::Quaterniond(o.offset_R_L_I[3], o.offset_R_L_I[0], o.offset_R_L_I[1], o.offset_R_L_I[2]);
}
}  // namespace lvref

namespace esekfom {
template <typename state, int process_noise_dof, typename input>
class esekf {
public:
    typedef Eigen::Matrix<double, 23, 23> cov;
    typedef std::function<void(state&, dyn_share_datastruct<double>&)> measurementModel_dyn_share;

    esekf() { P_.setIdentity(); last() = this; }
    // the instance the reference's Localizator singleton owns (the glue reads / writes x and P in f64 through it)
    static esekf*& last() { static esekf* p = nullptr; return p; }

    template <typename F, typename FX, typename FW>
    void init_dyn_share(F, FX, FW, measurementModel_dyn_share h, int maximum_iteration, const std::vector<double>& limit_vector) {
        h_dyn_share = h;
        maximum_iter = maximum_iteration;
        limit = limit_vector;
    }
    const state& get_x() const { return x_; }
    void change_x(state& s) { x_ = s; }
    cov get_P() const { return P_; }
    void change_P(cov& P) {
        P_ = P;
        if (on_change_P()) { auto f = on_change_P(); on_change_P() = nullptr; f(); on_change_P() = f; }   // (replay hook: the filter has just been initialised)
    }
    static std::function<void()>& on_change_P() { static std::function<void()> f; return f; }
    std::vector<lvo_state> update_log;    // the state after every update_iterated_dyn_share_modified (replay output)

    // esekf::predict(dt, Q, i_in) -> the oracle's restatement; row-major buffers across the C interface
    void predict(double& dt, Eigen::Matrix<double, process_noise_dof, process_noise_dof>& Q, const input& i_in) {
        lvo_state xo;
        lvref::to_oracle(x_, xo);
        double P[23 * 23], Qr[12 * 12];
        for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) P[i * 23 + j] = P_(i, j);
        for (int i = 0; i < 12; ++i) for (int j = 0; j < 12; ++j) Qr[i * 12 + j] = Q(i, j);
        const double a[3] = {i_in.acc(0), i_in.acc(1), i_in.acc(2)}, g[3] = {i_in.gyro(0), i_in.gyro(1), i_in.gyro(2)};
        lvo_predict(&xo, P, dt, Qr, a, g);
        lvref::from_oracle(xo, x_);
        for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) P_(i, j) = P[i * 23 + j];
    }

    // the fork's 4-argument iterated update (call site Localizator.cpp:132) [UPSTREAM-RECALL; degeneracy stage OFF]: loop from
    // i = -1, the registered measurement model evaluated at the current iterate, the oracle's kf_step on H^T H / H^T h, the
    // posterior committed when two passes have converged or the iterations are used up.  Same control flow as lvo_update; the
    // measurement comes from the REFERENCE's code through h_dyn_share.
    void update_iterated_dyn_share_modified(double R, double /*degeneracy_threshold*/, double& /*solve_time*/, bool /*print*/) {
        lvo_params prm;
        lvo_default_params(&prm);
        prm.max_num_iters = maximum_iter;
        prm.lidar_noise = R;
        for (int i = 0; i < 23; ++i) prm.limits[i] = i < (int)limit.size() ? limit[(size_t)i] : 0.001;
        lvo_state x, x_prop;
        lvref::to_oracle(x_, x);
        x_prop = x;
        double P_prop[23 * 23], P_post[23 * 23];
        for (int i = 0; i < 23; ++i) for (int j = 0; j < 23; ++j) P_prop[i * 23 + j] = P_(i, j);
        int t = 0;
        passes = 0;
        sums_log.clear();
        trace_log.clear();
        for (int i = -1; i < maximum_iter; ++i) {
            dyn_share_datastruct<double> dyn_share;
            dyn_share.valid = true;
            state xs;
            lvref::from_oracle(x, xs);
            h_dyn_share(xs, dyn_share);
            lvo_iter_out sums;
            std::memset(&sums, 0, sizeof(sums));
            if (dyn_share.valid) {
                const Eigen::Index n = dyn_share.h_x.rows();
                for (Eigen::Index r = 0; r < n; ++r) {   // H^T H, H^T h in f64, rows in match order
                    for (int a = 0; a < 12; ++a) {
                        for (int b = 0; b < 12; ++b) sums.HTH[a * 12 + b] += dyn_share.h_x(r, a) * dyn_share.h_x(r, b);
                        sums.HTh[a] += dyn_share.h_x(r, a) * dyn_share.h(r);
                    }
                    sums.sum_h2 += dyn_share.h(r) * dyn_share.h(r);
                }
                sums.n_valid = (int64_t)n;
            }
            sums_log.push_back(sums);
            ++passes;
            if (!dyn_share.valid) { trace_log.push_back(x); continue; }
            double dxo[23];
            const int converge = lvo_kf_step(&x, &x_prop, P_prop, &prm, &sums, dxo, 1, P_post);
            trace_log.push_back(x);
            if (converge) t++;
            if (t > 1 || i == maximum_iter - 1) {
                for (int a = 0; a < 23; ++a) for (int b = 0; b < 23; ++b) P_(a, b) = P_post[a * 23 + b];
                break;
            }
        }
        lvref::from_oracle(x, x_);
        update_log.push_back(x);
        pass_log.push_back(passes);
    }

    int passes = 0;
    std::vector<int> pass_log;   // measurement passes of every update so far (ref_stream_main dumps it: LV_DEMO_PASSES_DUMP)
    std::vector<lvo_iter_out> sums_log;
    std::vector<lvo_state> trace_log;
private:
    state x_;
    cov P_;
    measurementModel_dyn_share h_dyn_share;
    int maximum_iter = 0;
    std::vector<double> limit;
};
}  // namespace esekfom
#endif
