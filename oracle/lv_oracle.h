/*
 * lv_oracle.h — C interface of the CPU ORACLE for the LIMO-Velo iterated-KF-update hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.  Only tests/, __graft_entry__.smoke() and
 * bench.py's cpu_baseline leg may load this library, and there only as the checker / the timed
 * CPU baseline.  The product path (limo-velo_amd/) never links or calls it.
 *
 * PARITY, what it is pinned to (round 5): the reference (Huguet57/LIMO-Velo) ships no tests and no golden vectors, and its build
 * needs ROS / PCL / Eigen and two submodules that are absent here.  Its IN-TREE sources, however, compile
 * in place against stand-in headers (oracle/ref_build/ -> oracle/_ref/liblvref.so), and tests/test_oracle_ref.py holds this
 * oracle to what that code computes, bit for bit: the world transform and State mirror, the Plane gates, estimate_plane's
 * call structure / normalisation, is_plane, Match, the chosen set, calculate_H's rows, the x0 / P0 / Q / propagate_to
 * schedule, State::propagate_f + Compensator::compensate, the Accumulator windows and the PointCloud2 time rules.
 * STILL UNPINNED — restated from the published algorithms of absent dependencies, tagged [UPSTREAM-RECALL] in lv_oracle.cpp,
 * and stand-ins on the reference side of that comparison as well: Eigen's QR internals and reduction order, ikd-Tree's search
 * order / tie rule / Add_Points box rule, esekf's update algebra (incl. the fork's degeneracy stage) and pcl::VoxelGrid.
 * Those are cross-checked against scipy / numpy / algebraic properties (tests/test_oracle*.py), not against reference outputs.
 */
#ifndef LV_ORACLE_H
#define LV_ORACLE_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Hot-path keys of struct Params (reference include/Headers/Common.hpp:56-107; defaults
 * config/params.yaml:32,46-53). */
typedef struct lvo_params {
    int    max_num_iters;        /* MAX_NUM_ITERS (3) -> maximum_iter; passes = max_num_iters+1 */
    int    num_match_points;     /* NUM_MATCH_POINTS (5); this oracle supports 1..8 */
    double max_dist_plane;       /* MAX_DIST_PLANE (2.0) */
    float  planes_threshold;     /* PLANES_THRESHOLD (0.05) */
    int    estimate_extrinsics;  /* estimate_extrinsics (false) */
    double lidar_noise;          /* LiDAR_noise (1e-3) -> R */
    double limits[23];           /* LIMITS (23 x 1e-3) */
    /* degeneracy stage of the fork's 4-argument update_iterated_dyn_share_modified (Localizator.cpp:132;
     * config/params.yaml:51-53).  The fork's source is absent: [UNKNOWN-FORK].  mode 0 = off (every
     * parity / benchmark run), 1 = eigenvalues of the pose block of H^T H reported only ("print_degeneracy_values"),
     * 2 = a plausible restatement (solution remapping, Zhang/Kaess/Singh 2016, in information form): measurement
     * information along eigen-directions of the 6x6 pose block whose eigenvalue is below the threshold is removed. */
    int    degeneracy_mode;
    double degeneracy_threshold;
} lvo_params;

/* state_ikfom (IKFoM fork; field order confirmed by reference src/Objects/State.cpp:53-61 and
 * Localizator.cpp:137-150).  Quaternions are stored in Eigen coefficient order x,y,z,w. */
typedef struct lvo_state {
    double pos[3];
    double rot[4];
    double offset_R_L_I[4];
    double offset_T_L_I[3];
    double vel[3];
    double bg[3];
    double ba[3];
    double grav[3];
} lvo_state;

/* f32 mirror built by State(const state_ikfom&, double) (State.cpp:51-62): row-major 3x3. */
typedef struct lvo_pose_f32 {
    float R[9];
    float pos[3];
    float RLI[9];
    float tLI[3];
} lvo_pose_f32;

typedef struct lvo_iter_out {
    double HTH[144];   /* row-major 12x12 */
    double HTh[12];
    double sum_h2;     /* sum of h(i)^2 over matches */
    int64_t n_valid;   /* number of chosen matches */
} lvo_iter_out;

void lvo_default_params(lvo_params* p);
void lvo_state_to_pose(const lvo_state* s, lvo_pose_f32* out);

/* World transform of Mapper.cpp:51:  p_w = X * X.I_Rt_L() * p  (f32, op order of
 * State.cpp:83-85, RotTransl.cpp:36-48).  xyz arrays are N x 3 packed floats. */
void lvo_transform_scan(const lvo_pose_f32* pose, const float* scan_xyz, size_t n, float* out_xyz);

/* Exact k-NN, brute force, distance (ax-bx)^2+(ay-by)^2+(az-bz)^2 in f32 unfused
 * [UPSTREAM-RECALL ikd-Tree calc_dist]; ties -> lowest map index.  idx/d2 are N x k, ascending;
 * entries past found[i] are 0xFFFFFFFF / +inf.  n_ties (may be NULL) counts queries whose k-th
 * and (k+1)-th distances are bit-equal. */
void lvo_knn_brute(const float* map_xyz, size_t m, const float* q_xyz, size_t n, int k,
                   uint32_t* idx, float* d2, int32_t* found, int64_t* n_ties, int nthreads);

/* Pointer kd-tree, one point per node, box-pruned descent, size-k max-heap
 * [UPSTREAM-RECALL ikd-Tree Build/Search].  Same result contract as lvo_knn_brute. */
void* lvo_kdtree_build(const float* map_xyz, size_t m);
void  lvo_kdtree_free(void* tree);
size_t lvo_kdtree_size(const void* tree);
void  lvo_kdtree_knn(const void* tree, const float* q_xyz, size_t n, int k,
                     uint32_t* idx, float* d2, int32_t* found, int nthreads);

/* Plane(near, sq_dists) (Plane.cpp:19-55) + R3Math::estimate_plane / is_plane (Utils.cpp:32-66).
 * near_xyz: found x 3 neighbours in ascending distance order.  Returns is_plane (0/1). */
int lvo_plane_fit(const float* near_xyz, const float* sq_dists, int found, const lvo_params* prm,
                  float abcd[4]);

/* f64 exact least-squares solution of the same 5x3 system (noise-floor reference for the f32 QR). */
void lvo_plane_fit_f64(const float* near_xyz, int npts, double abcd[4]);

/* One Jacobian row of Localizator::calculate_H (Localizator.cpp:29-57) for a world point p_w
 * matched to plane abcd with signed distance dist. */
void lvo_calculate_H_row(const lvo_state* s, const float p_w[3], const float abcd[4], float dist,
                         int estimate_extrinsics, double Hrow[12], double* h);

/* One measurement-model evaluation = IKFoM::h_share_model [UPSTREAM-RECALL glue] =
 * Mapper::match (Mapper.cpp:40-56) + Localizator::calculate_H + the H^T H / H^T h products of
 * esekf (a-8).  tree == NULL -> brute-force kNN over map_xyz.  Optional per-point outputs (any
 * may be NULL): knn_idx N x k, knn_d2 N x k, valid N, abcd N x 4, dist N, Hrows N x 12, h N. */
void lvo_iterate(const lvo_state* s, const lvo_params* prm, const void* tree,
                 const float* map_xyz, size_t m, const float* scan_xyz, size_t n,
                 lvo_iter_out* out, uint32_t* knn_idx, float* knn_d2, uint8_t* valid,
                 float* abcd, float* dist, double* Hrows, double* h, int nthreads);

/* esekf::update_iterated_dyn_share_modified [UPSTREAM-RECALL, degeneracy stage off].
 * x, P (row-major 23x23) updated in place.  trace (may be NULL): per pass 23 doubles dx_ followed
 * by the 26 state doubles after boxplus (49 per pass, up to max_num_iters+1 passes).
 * Returns number of measurement passes executed. */
int lvo_update(lvo_state* x, double* P, const lvo_params* prm, const void* tree,
               const float* map_xyz, size_t m, const float* scan_xyz, size_t n,
               double* trace, lvo_iter_out* per_pass_out, int nthreads);

/* The algebra of one esekf pass given the reduced sums (used to check the device solve alone):
 * consumes HTH/HTh/n_valid, x (current), x_prop, P_prop; produces dx_ and updates x; when
 * `finalize` it also writes the posterior P.  Returns 1 if |dx_| <= limits for all 23 dof. */
int lvo_kf_step(lvo_state* x, const lvo_state* x_prop, const double* P_prop, const lvo_params* prm,
                const lvo_iter_out* sums, double dx_out[23], int finalize, double* P_out);
/* The degeneracy stage alone: eigenvalues (cyclic Jacobi, 8 sweeps, unsorted) of the 6x6 pose block of sums->HTH into
 * eig[6]; with prm->degeneracy_mode == 2 the sums are modified in place (see lvo_params). */
void lvo_degeneracy(lvo_iter_out* sums, const lvo_params* prm, double eig[6]);

/* Manifold helpers exposed for tests. */
void lvo_boxplus(lvo_state* x, const double dx[23]);
void lvo_boxminus(const lvo_state* x, const lvo_state* other, double dx[23]);

/* IMU predict  esekf::predict(dt, Q, in) with LIMO-Velo's process model [UPSTREAM-RECALL
 * use-ikfom get_f/df_dx/df_dw]; Q is row-major 12x12.  Used to produce realistic P for tests. */
void lvo_predict(lvo_state* x, double* P, double dt, const double* Q, const double acc[3],
                 const double gyro[3]);

/* KD_TREE::Add_Points(PointToAdd, downsample_on) [UPSTREAM-RECALL ikd-Tree], call site reference
 * src/Modules/Mapper.cpp:73-76 with box_length = 0.2 m (Mapper.cpp:65), processed SEQUENTIALLY in input order:
 * for every new point p the 0.2 m box holding it is searched; the point nearest to the box centre among
 * {p} U (points currently in the box) is kept when the box held more than one point or p itself is that
 * nearest point (ties: p wins against the current occupants, which must be STRICTLY closer); otherwise the
 * box is left untouched.  downsample == 0 appends.  The resulting map is written to out_xyz as
 * [surviving old points in their old order] + [surviving new points in input order]; returns its size
 * (capacity needed <= m + k). */
size_t lvo_map_add(const float* map_xyz, size_t m, const float* new_xyz, size_t k, int downsample, float box_length,
                   float* out_xyz);

/* ---- row f-2: Compensator (de-skew + voxel-grid down-sampling) ---------------------------------------- */
/* f32 members of the reference's State that State::propagate_f reads (src/Objects/State.cpp:94-110;
 * include/Headers/Objects.hpp:97-137).  Row-major matrices. */
typedef struct lvo_motion_state {
    float R[9], pos[3], vel[3], bw[3], ba[3], g[3], RLI[9], tLI[3], a[3], w[3];
    float pad_[2];
    double time;
} lvo_motion_state;   /* 184 bytes */

/* State::operator+=(IMU(a, w, t)) = State::update -> propagate_f (State.cpp:94-121), f32. */
void lvo_state_integrate(lvo_motion_state* s, const float a[3], const float w[3], double t);
/* on != 0: SO3Math::Exp inside lvo_state_integrate / lvo_deskew evaluates sin / cos with this platform's sinf / cosf (what the
 * reference calls) instead of the pinned polynomial below — used only to compare the oracle with oracle/_ref bit for bit. */
void lvo_set_sincos_libm(int on);
/* Experiment switch (NOT the default): the plane fit's back substitution in Eigen 3.3's column-oriented order (tests/test_oracle_pins.py, DESIGN "if the recall is wrong here"). */
void lvo_set_qr_backsub_columns(int on);
/* sin / cos of f32 arguments as rows f-2 / f-3 evaluate them (one fixed f64 polynomial, rounded to f32) */
void lvo_sincos_f32(const float* x, size_t n, float* sn, float* cs);
/* The pinned sin / cos polynomial of row f-2 against this platform's sinf / cosf (what the reference calls,
 * include/Headers/Utils.hpp:46): how many of the n arguments differ, and by how many ulps at most. */
void lvo_sincos_vs_libm(const float* x, size_t n, int64_t* n_sin_diff, int64_t* n_cos_diff, int32_t* max_ulp);

/* Compensator::compensate(states, Xt2, points) (src/Modules/Compensator.cpp:123-146): for every point (xyz +
 * time, time-sorted) take the surrounding state, integrate it to the point's time with its own last IMU, move
 * the point to the world and back into the LiDAR frame at t2.  out_xyz: n x 3. */
void lvo_deskew(const float* xyz, const double* times, size_t n, const lvo_motion_state* states, size_t n_states,
                const lvo_motion_state* Xt2, float* out_xyz);

/* Compensator::voxelgrid_downsample (Compensator.cpp:148-163) = pcl::VoxelGrid with leaf size `leaf`
 * [UPSTREAM-RECALL PCL 1.8 VoxelGrid::applyFilter]: centroid (f32 sums in input order / count) of every
 * occupied leaf, leaves in ascending index order i + j*dx + k*dx*dy.  Returns the number of output points. */
size_t lvo_voxelgrid(const float* xyz, size_t n, float leaf, float* out_xyz);

/* ---- row f-4: LiDAR wire formats (PointCloud2 -> time-stamped Points) -------------------------------------- */
/* Field layout of one point record (msg.fields / msg.point_step).  time_type: 0 = F32 seconds (velodyne `time`),
 * 1 = F64 seconds (hesai / custom `timestamp`), 2 = U32 nanoseconds (ouster `t`); intensity_type / range_type:
 * 0 none, 1 F32, 2 U8, 3 U16, 4 U32. */
typedef struct lvo_cloud_format {
    uint32_t point_step, off_x, off_y, off_z, off_time;
    int time_type;
    uint32_t off_intensity;
    int intensity_type;
    uint32_t off_range;
    int range_type;
    int relative_time;
} lvo_cloud_format;
typedef struct lvo_ingest_params {
    uint64_t header_stamp_usec;
    int stamp_beginning, offset_beginning;
    double full_rotation_time;
    int downsample_rate;
    float min_dist;
} lvo_ingest_params;
typedef struct lvo_point {   /* reference Point, include/Headers/Objects.hpp:20-28 */
    float x, y, z, pad_;
    double time;
    float intensity, range;
} lvo_point;
/* Accumulator::process (src/Modules/Accumulator.cpp:143-153): PointCloudProcessor::msg2points (per-sensor time
 * rules of src/Objects/Point.cpp:37-111 and get_begin_time, src/Utils/PointCloudProcessor.cpp:43-91), ::downsample
 * (:99-110) and ::sort_points (:112-121; stable here — std::sort leaves the order of equal stamps unspecified).
 * out: capacity n records; returns the number of points kept. */
size_t lvo_cloud_ingest(const void* data, size_t n, const lvo_cloud_format* f, const lvo_ingest_params* prm, lvo_point* out);

#ifdef __cplusplus
}
#endif
#endif
