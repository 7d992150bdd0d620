/*
 * limovelo_hip.h — C-ABI of the MI355X-native LIMO-Velo iterated-KF-update hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  Each
 * entry point names the reference interface it replaces (paths relative to the reference repo
 * Huguet57/LIMO-Velo).  The C++ shim classes `Mapper` / `Localizator` in
 * limo-velo_amd/host/ keep the reference's method names on top of these calls; INTEGRATION.md shows
 * the binding a maintainer adds to the ROS node.
 *
 * Conventions
 *   - every function returns an int status: LV_OK (0) or a negative LV_E* code; lv_last_error()
 *     returns a thread-local message for the last failure.  "No map yet" and "fewer than k map
 *     points" are NOT errors: they yield n_valid = 0, mirroring Mapper::match returning an empty
 *     vector (src/Modules/Mapper.cpp:42) and Localizator::correct returning early
 *     (src/Modules/Localizator.cpp:24).
 *   - point arrays are passed as (base pointer, stride in bytes, count); x,y,z are three consecutive
 *     floats at the start of each record.  stride = 32 accepts the reference's `Point` records
 *     (include/Headers/Objects.hpp:20-28: float x,y,z; double time; float intensity, range) as they
 *     lie in a PointVector; stride = 12 accepts packed xyz.
 *   - a context is single-caller and non-reentrant (the reference modules are singletons driven from
 *     one thread: src/main.cpp:52,128).  All device work of a context is issued on one HIP stream.
 *   - the HIP extension is mandatory: there is no CPU fallback behind this ABI.
 */
#ifndef LIMOVELO_HIP_H
#define LIMOVELO_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LV_OK 0
#define LV_EINVAL (-1)   /* bad argument */
#define LV_EHIP (-2)     /* HIP runtime error (message in lv_last_error) */
#define LV_ENODEV (-3)   /* no usable GPU */
#define LV_ESTATE (-4)   /* call out of order (e.g. iterate before scan_set) */
#define LV_ERANGE (-5)   /* coordinates outside the supported voxel range */

#define LV_STATE_DOF 23
#define LV_SUMS_LEN 96   /* doubles in the per-pass reduction record (see lv_sums_layout below) */

typedef struct lv_ctx lv_ctx;

/* Hot-path keys of `struct Params` (include/Headers/Common.hpp:56-107; defaults from
 * config/params.yaml:32,46-53 and src/main.cpp:145), same names. */
typedef struct lv_params {
    int    MAX_NUM_ITERS;        /* 3  -> esekf maximum_iter; up to MAX_NUM_ITERS+1 measurement passes */
    int    NUM_MATCH_POINTS;     /* 5  (k of Nearest_Search and rows of the plane fit, Mapper.cpp:85-86, Utils.cpp:33; 3..8: every tuned
                                  *     path — one launch per pass, the benchmark — is built for 5, other values run a general build of
                                  *     the three-kernel pass with lanes_per_query 8; LV_EINVAL outside 3..8) */
    double MAX_DIST_PLANE;       /* 2.0 */
    float  PLANES_THRESHOLD;     /* 0.05 */
    int    estimate_extrinsics;  /* 0 */
    double LiDAR_noise;          /* 0.001 */
    double LIMITS[LV_STATE_DOF]; /* 23 x 0.001 */
    double degeneracy_threshold; /* 5.0 (config/params.yaml:52; kitti.yaml:46 = 400): used by degeneracy_mode 2 */
    /* ---- structure tuning (no reference counterpart) ---- */
    float  voxel_size;           /* level-0 cell edge of the voxel hash in metres (default 0.5) */
    int    lanes_per_query;      /* 1,2,4,8,16: lanes of a wavefront cooperating on one scan point (default 8) */
    /* ---- degeneracy stage of the fork's update_iterated_dyn_share_modified(R, degeneracy_threshold, solve_time,
     * print_degeneracy_values) (src/Modules/Localizator.cpp:132).  The fork's IKFoM source is absent, so
     * the stage is a HOOK with an opt-in restatement, off by default:
     *   0  off: esekf's plain iterated update (every parity and benchmark run); lv_create warns once on stderr if
     *      degeneracy_threshold was changed from its default, because it then has no effect;
     *   1  the eigenvalues of the 6x6 pose block of H^T H are computed every pass and kept for
     *      lv_get_degeneracy_values ("print the degeneracy eigenvalues to guess what the threshold must be",
     *      config/params.yaml:53); the update itself is unchanged;
     *   2  [UNKNOWN-FORK, plausible restatement] solution remapping (Zhang, Kaess, Singh, ICRA 2016) in information
     *      form: measurement information along pose eigen-directions whose eigenvalue is below degeneracy_threshold is
     *      removed before the gain is formed; those directions keep their propagated value.
     * print_degeneracy_values != 0 (with mode >= 1) also prints the eigenvalues of every pass to stderr. */
    int    degeneracy_mode;
    int    print_degeneracy_values;
} lv_params;

/* state_ikfom of the IKFoM fork (field order evidenced by src/Objects/State.cpp:53-61 and
 * src/Modules/Localizator.cpp:137-150).  Quaternions in Eigen coefficient order x,y,z,w.
 * 26 doubles, no padding. */
typedef struct lv_state {
    double pos[3];
    double rot[4];
    double offset_R_L_I[4];
    double offset_T_L_I[3];
    double vel[3];
    double bg[3];
    double ba[3];
    double grav[3];
} lv_state;

/* Result of one measurement-model evaluation: what esekf needs from the N-sized data
 * (SURVEY §8 a-8): H^T H (12x12 row-major), H^T h, sum h^2, number of chosen matches. */
typedef struct lv_sums {
    double  HTH[144];
    double  HTh[12];
    double  sum_h2;
    int64_t n_valid;
} lv_sums;

/* Device-side record layout of the LV_SUMS_LEN doubles that lv_pass_reduce() produces and that is
 * all-reduced (sum) across GPUs: [0..77] upper triangle of H^T H row by row, [78..89] H^T h,
 * [90] n_valid (as double), [91] sum h^2, [92..95] zero padding. */

void        lv_default_params(lv_params* p);
const char* lv_last_error(void);
const char* lv_version(void);

/* ---- context -------------------------------------------------------------------------------- */
/* device = HIP device ordinal.  Replaces the construction of the Mapper / Localizator singletons
 * (include/Headers/Mapper.hpp:35-38, src/Modules/Localizator.cpp:100-117 init_IKFoM). */
int  lv_create(const lv_params* params, int device, lv_ctx** out);
void lv_destroy(lv_ctx* ctx);
/* Issue all work on an existing hipStream_t (e.g. torch's current stream); NULL = context's own. */
int  lv_set_stream(lv_ctx* ctx, void* hip_stream);
void* lv_get_stream(lv_ctx* ctx);
int  lv_synchronize(lv_ctx* ctx);

/* ---- Mapper side ---------------------------------------------------------------------------- */
/* KD_TREE<Point>::Build(PointVector)              — call site src/Modules/Mapper.cpp:68-71 */
int    lv_map_build(lv_ctx* ctx, const void* points, size_t stride, size_t n);
/* Mapper::add(points, time, downsample) — src/Modules/Mapper.cpp:19-30: on an empty map the points BUILD it
 * (KD_TREE::Build, no down-sampling), otherwise KD_TREE<Point>::Add_Points(PointVector&, bool) — call site
 * src/Modules/Mapper.cpp:73-76.  downsample != 0 applies ikd-Tree's box rule with box_length 0.2 m (Mapper.cpp:65),
 * evaluated exactly as upstream's sequential loop would: in every 0.2 m box touched by new points only the point
 * nearest to the box centre survives (occupants must be strictly closer to beat a new point).  The map afterwards is
 * [surviving old points, old order] + [surviving new points, input order] — the index space of lv_fetch_knn.
 * INCREMENTAL: only the neighbourhood buckets / voxel lists that contain a new or a deleted point are touched
 * (appends into slack, tombstones); the structure is re-linearised (ids compacted, everything rebuilt) when a pool,
 * a table or the share of dead ids runs high — see lv_map_get_stats. */
int    lv_map_add(lv_ctx* ctx, const void* points, size_t stride, size_t n, int downsample);
/* The mapping step of the main loop without leaving the device (src/main.cpp:92,102):
 *     Points global_ds_compensated = Xt2 * Xt2.I_Rt_L() * ds_compensated;   map.add(global_ds_compensated, t2, true);
 * the current scan (lv_scan_set / lv_scan_deskew*) is moved to the world with the state the device holds — the
 * resident filter's (lv_filter_set / lv_predict / lv_correct) if there is one, otherwise the result of the last
 * lv_update — in f32 exactly as State::State(const state_ikfom&) + RotTransl do (rows a-1), and inserted in scan
 * order.  Non-finite points are skipped.  The insert is enqueued and runs BESIDE whatever the caller enqueues next (on a
 * second stream of the context, when the context owns its stream): every call that touches the map waits for it first, so
 * the only visible effect is that lv_map_add / lv_map_add_scan return before the map has changed. */
int    lv_map_add_scan(lv_ctx* ctx, int downsample);
/* Rolling window (BASELINE configs[4]; ikd-Tree's Delete_Point_Boxes, which the reference never calls —
 * README.md:127 — but a bounded map needs): keep_inside != 0 removes every point OUTSIDE the axis-aligned box
 * [lo, hi], keep_inside == 0 every point INSIDE it.  lv_map_evict_oldest removes the n_oldest oldest living points
 * (map order = age).  Indices of the survivors shift down as in any deletion (lv_fetch_knn reports ranks among
 * the living). */
int    lv_map_evict_box(lv_ctx* ctx, const float lo[3], const float hi[3], int keep_inside, size_t* n_evicted);
int    lv_map_evict_oldest(lv_ctx* ctx, size_t n_oldest, size_t* n_evicted);
/* Force the periodic re-linearisation now (compaction of the ids + rebuild of every bucket). */
int    lv_map_relinearise(lv_ctx* ctx);
/* The same compaction + rebuild WITHOUT stopping the world (round 5).  ikd-Tree rebuilds unbalanced sub-trees on a second thread
 * while searches go on (the tree the reference constructs with delete / balance criteria 0.3 / 0.6, src/Modules/Mapper.cpp:65);
 * here a compacted copy of the living points is rebuilt by a worker thread on a stream of its own while searches and inserts
 * keep using the active structure, the inserts / evictions of the meantime are replayed on the copy, and the two are swapped
 * at the first map call after the worker has caught up (a cycle boundary of src/main.cpp:75-103).  Point set, id order and
 * every search result are those of lv_map_relinearise.  lv_map_add / lv_map_add_scan start one by themselves when a third of
 * the id space is dead and the map holds >= 200 000 points (lv_set_option "async_relinearise" 0: always the stop-the-world
 * form; "async_relinearise_min": the size threshold).  Returns at once. */
int    lv_map_relinearise_async(lv_ctx* ctx);
/* Set-up time (round 6): allocate — and touch — everything a background rebuild of the map as it stands needs (the second store's
 * id buffers, pools and tables at the sizes a rebuild takes, the journal arena), by one synchronous rebuild of a snapshot that is
 * then discarded.  Without it the FIRST background rebuild of a context spends its opening ~80 cycles of a 100 Hz stream in
 * hipMalloc on the worker thread (a 10 M-point map: ~20 GB); later ones find the previous active store waiting either way.  The
 * price is the second store's memory from the start instead of from the first rebuild.  Blocks; LV_ESTATE while a rebuild is in
 * flight.  (The reference's counterpart: ikd-Tree's rebuild thread allocates its Rebuild_PCL_Storage on demand.) */
int    lv_map_reserve_rebuild(lv_ctx* ctx);
/* out = {state, rebuilds started, rebuilds adopted, journaled operations not yet replayed}; wait != 0: block until a rebuild in
 * flight has been adopted.  States: 0 idle; 4 the worker is allocating the second store; 5 allocated, waiting for the snapshot the
 * next map call enqueues; 1 rebuilding / replaying the journal; 2 rebuilt and waiting to be adopted; 3 failed (the next map call
 * reports it on stderr, keeps the map as it is and switches this context to stop-the-world re-linearisations).  "In flight" is any
 * state other than 0 and 3. */
int    lv_map_rebuild_status(lv_ctx* ctx, int wait, uint64_t out[4]);
/* KD_TREE<Point>::size()                          — src/Modules/Mapper.cpp:33,79 */
size_t lv_map_size(lv_ctx* ctx);
/* Copy the current map points (xyz packed, map order = the index space of lv_fetch_knn). */
int    lv_map_fetch(lv_ctx* ctx, float* xyz_out, size_t capacity);
typedef struct lv_map_stats {
    uint64_t living, ids, capacity;          /* points alive / ids handed out since the last re-linearisation / id slots allocated */
    uint64_t pool_used[4], pool_cap[4];      /* entries of [0] the level-0 bucket pool, [1] the voxel-list pool ([2], [3] unused since the
                                                single-replicated-level map of round 6; rounds 1-5: three bucket pools + the lists) */
    uint64_t slots_used[4], slots_cap[4];    /* occupied / total slots of [0] the bucket table, [1] the voxel-list table */
    uint64_t tombstones, dropped;            /* dead bucket entries since the last rebuild; points refused (non-finite / out of range) */
    uint64_t relinearisations, incremental_adds;
    uint64_t bytes;                          /* device memory held by the map */
} lv_map_stats;
int    lv_map_get_stats(lv_ctx* ctx, lv_map_stats* out);

/* ---- Localizator side ----------------------------------------------------------------------- */
/* `this->points2match = points`                   — src/Modules/Localizator.cpp:131.
 * Uploads the scan (LiDAR frame) once per correct(); it is invariant across IKFoM passes. */
int lv_scan_set(lv_ctx* ctx, const void* points, size_t stride, size_t n);

/* ---- row f-2: Compensator on the device ------------------------------------------------------------------
 * f32 members of the reference's `State` that State::propagate_f reads (include/Headers/Objects.hpp:97-137,
 * src/Objects/State.cpp:94-110); matrices row-major.  184 bytes. */
typedef struct lv_motion_state {
    float R[9], pos[3], vel[3], bw[3], ba[3], g[3], RLI[9], tLI[3], a[3], w[3];
    float pad_[2];
    double time;
} lv_motion_state;

/* Compensator::compensate(states, Xt2, points) + Compensator::downsample (src/Modules/Compensator.cpp:123-163):
 * de-skews a time-stamped raw scan (records: x,y,z floats at offset 0, a double time stamp at `time_offset`
 * bytes — 16 for the reference's Point) with the piecewise-constant-IMU motion model of State::propagate_f
 * over `states` (time-ordered, surrounding the points), moves every point into the LiDAR frame of Xt2 and,
 * if downsample_prec > 0, voxel-grid down-samples it (pcl::VoxelGrid semantics: centroid per leaf).  The
 * result becomes the current scan exactly as if lv_scan_set had been called with it. */
int lv_scan_deskew(lv_ctx* ctx, const void* points, size_t stride, size_t time_offset, size_t n,
                   const lv_motion_state* states, size_t n_states, const lv_motion_state* Xt2, float downsample_prec);
/* Compensator::downsample(points) on its own (src/Modules/Compensator.cpp:104-107,148-163): the same voxel grid for
 * points that are already compensated (downsample_prec <= 0: the points become the scan as they are, = lv_scan_set). */
int lv_scan_downsample(lv_ctx* ctx, const void* points, size_t stride, size_t n, float downsample_prec);
/* number of points of the current scan / copy them out (xyz packed, de-skew output order) */
/* ---- row f-4: LiDAR wire formats (sensor_msgs/PointCloud2 -> the reference's time-stamped Points) -------------
 * lv_cloud_ingest = Accumulator::process (src/Modules/Accumulator.cpp:143-153): PointCloudProcessor::msg2points
 * for the velodyne / hesai / ouster / custom point types (src/Utils/PointCloudProcessor.cpp:24-97 with the time
 * rules of src/Objects/Point.cpp:37-111), ::downsample (every downsample_rate-th point whose |p| > min_dist,
 * :99-110) and ::sort_points (by time, :112-121), followed by Accumulator::push of every point (:141) — into a
 * device-resident LiDAR buffer (BUFFER_L).  `data` = msg.data (n_points records of format->point_step bytes; the
 * field offsets come from msg.fields, lv_cloud_format_preset gives the PCL in-memory layouts of
 * include/Headers/Common.hpp:109-221).  lv_cloud_fetch = Accumulator::get_points(t1, t2) (t1 <= time <= t2, oldest
 * first; records are the reference's 32-byte Point: x,y,z floats, double time at 16, intensity, range);
 * lv_cloud_clear = Accumulator::clear_lidar(t) (drops time <= t from the old end; applied by the next window kernel, or by
 * whatever needs the buffer's true extent first);
 * lv_scan_deskew_window = the point half of Compensator::compensate(t1, t2) (src/Modules/Compensator.cpp:18-35):
 * the buffered points of [t1, t2] are de-skewed exactly as lv_scan_deskew does, without leaving the device. */
enum { LV_LIDAR_VELODYNE = 0, LV_LIDAR_HESAI = 1, LV_LIDAR_OUSTER = 2, LV_LIDAR_CUSTOM = 3 };
enum { LV_TIME_F32_SEC = 0, LV_TIME_F64_SEC = 1, LV_TIME_U32_NSEC = 2 };
enum { LV_ATTR_NONE = 0, LV_ATTR_F32 = 1, LV_ATTR_U8 = 2, LV_ATTR_U16 = 3, LV_ATTR_U32 = 4 };
typedef struct lv_cloud_format {
    uint32_t point_step;            /* msg.point_step */
    uint32_t off_x, off_y, off_z;   /* FLOAT32 fields */
    uint32_t off_time;              /* velodyne `time` (F32 s), hesai / custom `timestamp` (F64 s), ouster `t` (U32 ns) */
    int time_type;                  /* LV_TIME_* */
    uint32_t off_intensity;         /* `intensity` (F32; hesai U8) or ouster `reflectivity` (U16) */
    int intensity_type;             /* LV_ATTR_* */
    uint32_t off_range;             /* ouster `range` (U32); LV_ATTR_NONE: range = |p| */
    int range_type;
    int relative_time;              /* 1: stamps relative to the header stamp (velodyne, ouster); 0: absolute */
} lv_cloud_format;
typedef struct lv_ingest_params {   /* config/params.yaml:29-35 */
    uint64_t header_stamp_usec;     /* pcl header stamp (microseconds) */
    int stamp_beginning;
    int offset_beginning;
    double full_rotation_time;
    int downsample_rate;
    float min_dist;
} lv_ingest_params;
int    lv_cloud_format_preset(int lidar_type, lv_cloud_format* out);
int    lv_cloud_ingest(lv_ctx* ctx, const void* data, size_t n_points, const lv_cloud_format* format, const lv_ingest_params* params,
                       size_t* n_kept);
size_t lv_cloud_size(lv_ctx* ctx);
int    lv_cloud_fetch(lv_ctx* ctx, double t1, double t2, void* points_out, size_t capacity, size_t* n);
int    lv_cloud_clear(lv_ctx* ctx, double t);
/* Optional: allocate the ingest staging (two pinned buffers + the device work arrays) for messages of up to
 * max_points_per_message records of point_step bytes and a LiDAR buffer of buffer_points points now, instead of on the first
 * message (pinned allocations take milliseconds: the first sweep of a stream otherwise pays them). */
int    lv_cloud_reserve(lv_ctx* ctx, size_t max_points_per_message, size_t point_step, size_t buffer_points);
/* Optional: size the work buffers of the 100 Hz cycle now — the de-skew / voxel-grid buffers for windows of up to
 * max_window_points raw points, the scan / hand-over / insert buffers for scans of up to max_scan_points points — instead of
 * growing them (allocate, synchronise, free) during the first cycles of a stream. */
int    lv_reserve_stream(lv_ctx* ctx, size_t max_window_points, size_t max_scan_points);
int    lv_scan_deskew_window(lv_ctx* ctx, double t1, double t2, const lv_motion_state* states, size_t n_states,
                             const lv_motion_state* Xt2, float downsample_prec, size_t* n_window);

size_t lv_scan_size(lv_ctx* ctx);
int    lv_scan_fetch(lv_ctx* ctx, float* xyz_out, size_t capacity);

/* One IKFoM::h_share_model evaluation (registered at src/Modules/Localizator.cpp:112) =
 * Mapper::match (Mapper.cpp:40-56) + Localizator::calculate_H (Localizator.cpp:29-57), reduced to
 * H^T H / H^T h on the GPU.  Synchronous. */
int lv_iterate(lv_ctx* ctx, const lv_state* x, lv_sums* out);

/* The Eigen-free half of IKFoM::h_share_model for an UNMODIFIED esekf loop (src/Modules/Localizator.cpp:105-117: the callback
 * update_iterated_dyn_share_modified calls; :132 the update): esekf depends on the N x 12 Jacobian and the residuals only
 * through H^T H and H^T h, so the callback hands it a pseudo measurement of *rows <= 12 rows with
 *     h_x^T h_x = sums->HTH      h_x^T h = sums->HTh
 * (h_x row-major, 12 columns = the first 12 tangent coordinates, rows >= *rows zero).  Without estimate_extrinsics: the upper
 * Cholesky factor of the leading 6 x 6 block padded with zero columns (6 rows; Localizator.cpp:52 zeroes columns 6..11);
 * with it, or if that block is not positive definite: the rank-revealing factor sqrt(lam) V^T of the symmetric eigen-
 * decomposition (12 / 6 rows, the rows of vanished eigenvalues zero).  sums->n_valid == 0 (dyn_share.valid = false) gives
 * *rows = 0.  Host arithmetic only: needs no context and no device.  The maintainer's h_share_model is then
 *     lv_iterate(ctx, &x, &sums); lv_pseudo_measurement(&sums, ext, hx, h, &rows); copy hx / h into dyn_share.h_x / .h
 * (both esekf gain branches checked in tests/test_hshare.py). */
int lv_pseudo_measurement(const lv_sums* sums, int estimate_extrinsics, double h_x[144], double h[12], int* rows);

/* esekf::update_iterated_dyn_share_modified(R, degeneracy_threshold, solve_time, print)
 *                                                 — call site src/Modules/Localizator.cpp:132.
 * Runs the whole iterated update on the device.  x and P (23x23 row-major) are updated in place.
 * passes (may be NULL) receives the number of measurement passes executed; per_pass (may be NULL)
 * receives the sums of each pass (capacity MAX_NUM_ITERS+1); trace (may be NULL) receives per pass
 * 23 doubles dx_ followed by the 26 state doubles after boxplus (49 x (MAX_NUM_ITERS+1)). */
int lv_update(lv_ctx* ctx, lv_state* x, double* P, int* passes, lv_sums* per_pass, double* trace);

/* ---- resident filter (row f-3): x and P stay on the device between prediction and correction -----------
 * lv_filter_set / lv_filter_get     = esekf::change_x + change_P / get_x + get_P (src/Modules/Localizator.cpp:136-152)
 * lv_predict(dt, Q, acc, gyro)      = esekf::predict(dt, Q, in) as called by Localizator::propagate
 *                                     (Localizator.cpp:159-173; Q row-major 12x12, acc = imu.a, gyro = imu.w)
 * lv_correct(passes)                = lv_update on the resident state; asynchronous when passes == NULL.
 * lv_filter_set itself neither uploads nor waits (round 4): the filter stays in pinned host memory until something needs it on the
 * device; an lv_correct that follows takes it along in its first launch's kernel arguments (as lv_update takes its x / P), so "set
 * the prior, correct" is two enqueue-only calls — bench.py's timed step.
 * Nothing here waits for the device except lv_filter_get and lv_correct with passes != NULL: lv_predict calls are queued (up to
 * eight steps with the same Q go out as one launch when something needs the filter: lv_correct, lv_filter_get, lv_map_add_scan,
 * lv_synchronize, an update by value); after lv_correct the posterior stays in the update's working copy until something needs
 * it elsewhere, and lv_filter_get reads it — and the pass count, lv_last_passes — from the host-mapped mailbox the update's
 * finishing pass writes (a poll, no copy).  The results are bit-identical to one launch per call with eager copies
 * (tests/test_gpu_filter.py). */
int lv_filter_set(lv_ctx* ctx, const lv_state* x, const double* P);
int lv_filter_get(lv_ctx* ctx, lv_state* x, double* P);
int lv_predict(lv_ctx* ctx, double dt, const double* Q, const double acc[3], const double gyro[3]);
int lv_correct(lv_ctx* ctx, int* passes);
/* Eigenvalues of the pose block (pos, rot) of H^T H of every pass of the last update run with degeneracy_mode >= 1:
 * eig receives n_passes x 6 doubles (capacity_passes rows available; Jacobi order, unsorted).  A REPORT, not a filter input:
 * updates that ran one launch per pass derive the values on the host from the logged sums, the three-kernel path derives them on
 * the device (FMA-contracted), both by the same fixed 8 cyclic Jacobi sweeps; the two agree to rounding, not bit for bit —
 * stated and tested tolerance 1e-9 relative to the largest eigenvalue (tests/test_gpu_configs.py, tests/test_gpu_parity.py). */
int lv_get_degeneracy_values(lv_ctx* ctx, double* eig, int capacity_passes, int* n_passes);

/* Split form of lv_update for multi-GPU runs (scan points sharded across ranks, map replicated):
 *   lv_update_begin(x, P)
 *   repeat MAX_NUM_ITERS+1 times:
 *       lv_pass_reduce()                      -> fills the device record lv_sums_device_ptr()
 *       <all-reduce(sum) LV_SUMS_LEN doubles at lv_sums_device_ptr() across ranks, same stream>
 *       lv_pass_solve()                       -> 23-dof solve, boxplus, convergence bookkeeping
 *   lv_update_end(x, P, passes)
 * All calls are asynchronous on the context stream except lv_update_end. */
int   lv_update_begin(lv_ctx* ctx, const lv_state* x, const double* P);
int   lv_pass_reduce(lv_ctx* ctx);
void* lv_sums_device_ptr(lv_ctx* ctx);
/* Use caller-owned device memory (>= LV_SUMS_LEN doubles, e.g. a torch tensor handed to RCCL) as
 * the sums record; NULL restores the context's own buffer. */
int   lv_set_sums_buffer(lv_ctx* ctx, void* device_ptr);
int   lv_pass_solve(lv_ctx* ctx);
int   lv_update_end(lv_ctx* ctx, lv_state* x, double* P, int* passes);

/* ---- multi-GPU, collective inside the library (SURVEY §8 row e) --------------------------------
 * One process per GPU; the scan is sharded (each rank calls lv_scan_set with its range), the map is
 * replicated.  After lv_comm_init every measurement pass of lv_update / lv_correct all-reduces the
 * 96-double record with RCCL (over xGMI) on the context stream: the whole iterated update is enqueued
 * without a host round trip per pass, and every rank ends with the bitwise identical state.  The reference
 * has no counterpart (single process); this replaces the same esekf call as lv_update.
 *   rccl_library: path of librccl to bind at run time (NULL: "librccl.so.1" from the loader path).  In a
 *                 process that also runs torch pass torch's bundled copy (<torch>/lib/librccl.so) so that
 *                 one RCCL serves both.
 *   lv_comm_unique_id: rank 0 creates the 128-byte id (ncclUniqueId) and hands it to the other ranks by any
 *                 means (torch.distributed broadcast, MPI, a file); lv_comm_init is collective.
 * lv_iterate stays per-rank (its sums describe this rank's points). */
#define LV_COMM_ID_BYTES 128
int lv_comm_unique_id(const char* rccl_library, void* id128);
int lv_comm_init(lv_ctx* ctx, const char* rccl_library, const void* id128, int rank, int world);
int lv_comm_destroy(lv_ctx* ctx);
int lv_comm_world(lv_ctx* ctx);
/* With a communicator lv_update / lv_correct run a pass as search / fit / reduce -> ncclAllReduce -> solve (three kernels
 * and a collective).  If the caller tells the LARGEST shard of the current scan over the ranks — n_max, the same value on
 * every rank, after every lv_scan_set* (which forgets it) — they take the one-launch-per-pass form instead: every rank
 * launches the same grid (sized for n_max), leaves its workgroup partials in its slot of a gather buffer, ncclAllGather
 * (in place, on the context stream) hands every rank all partials, and the next launch's prologue folds them in the same
 * fixed order on every rank: identical states without a broadcast, one launch + one collective per pass.
 * lv_set_comm_fused(ctx, 0) (or LV_COMM_FUSED=0) keeps the three-kernel form. */
int lv_comm_set_shard_max(lv_ctx* ctx, size_t n_max);
int lv_set_comm_fused(lv_ctx* ctx, int enabled);
/* The same one-launch-per-pass multi-rank form with the CALLER's transport instead of librccl (bring-up on fabrics RCCL
 * does not serve, and the two-ranks-on-one-GPU test of exactly the kernels, buffers and fold the RCCL route uses): after
 * every searching launch the library copies this rank's slot of the gather buffer to host memory and calls
 *   fn(user, slots, bytes_per_rank, rank, world)
 * with `slots` = world x bytes_per_rank bytes of host memory holding this rank's partials at slots + rank * bytes_per_rank;
 * fn fills in every other rank's slot (an all-gather by whatever means; it returns 0 when they are all there) and the
 * library copies the whole buffer back.  Tell the largest shard with lv_comm_set_shard_max as above; without it, or for
 * scans the one-launch form does not take, lv_update / lv_correct fail with LV_ESTATE (there is no all-reduce transport
 * here).  fn = NULL removes it.  Not combinable with lv_comm_init. */
typedef int (*lv_gather_fn)(void* user, void* slots, size_t bytes_per_rank, int rank, int world);
int lv_comm_set_host_gather(lv_ctx* ctx, int rank, int world, lv_gather_fn fn, void* user);

/* The same one-launch-per-pass multi-rank form over PEER-MAPPED memory (HIP IPC over xGMI), no collective library: every
 * rank exports the handles of its gather buffers and of its flag word (lv_comm_peer_export: LV_PEER_HANDLE_BYTES = 128 bytes,
 * two HIP IPC handles), the caller carries the blobs of all ranks to every rank by whatever means it has (they are plain
 * bytes), and lv_comm_peer_init maps them.  After each pass one small kernel publishes "my partials of this launch are in
 * memory" and pulls the other ranks' slots straight out of their buffers — a one-shot peer read instead of a ring collective
 * (SURVEY 8e).  Tell the largest shard with lv_comm_set_shard_max as above; ranks end bitwise equal.
 * Failure: a rank that does not publish within the give-up time (LV_PEER_TIMEOUT_MS in the environment, default 2000 ms —
 * far above ordinary host skew between processes) ends the others' wait; the rank that gave up poisons its own flag, so
 * every rank of the node fails the SAME update.  lv_update then returns LV_ESTATE with x and P untouched; the resident
 * filter (lv_correct) is declared unset and every later call fails with LV_ESTATE until lv_comm_destroy + lv_filter_set.
 * At most 8 ranks (one node); HSA_ENABLE_IPC_MODE_LEGACY=0 must be set in the environment of every rank.
 * EXPERIMENTAL and opt-in: verified with two processes on one GPU (shared L2), not yet across GPUs; a second
 * lv_comm_peer_init on the same context returns LV_ESTATE.  Not combinable with lv_comm_init / lv_comm_set_host_gather. */
#define LV_PEER_HANDLE_BYTES 128
int lv_comm_peer_export(lv_ctx* ctx, void* handles /* LV_PEER_HANDLE_BYTES */);
int lv_comm_peer_init(lv_ctx* ctx, int rank, int world, const void* handles /* world x LV_PEER_HANDLE_BYTES, in rank order */);

/* ---- API-parity / debug fetches (results of the most recent CAPTURED pass; original scan order) --
 * lv_iterate always captures; lv_update captures only after lv_set_capture(ctx, 1) (the last pass
 * executed wins).  Capturing writes ~200 B per scan point and is off on the fast path. */
int lv_set_capture(lv_ctx* ctx, int enabled);
/* Nearest_Search outputs: idx N x k (index into the map in insertion order, 0xFFFFFFFF = none),
 * d2 N x k squared distances ascending (+inf = none); k = NUM_MATCH_POINTS. */
int lv_fetch_knn(lv_ctx* ctx, uint32_t* idx, float* d2);
/* The same Nearest_Search outputs as the hand-over records of the most recent pass hold them, whatever build
 * ran it — in particular the NON-capturing kernels of lv_update / lv_correct / lv_pass_reduce (the timed path),
 * which carry no map indices: nbr_xyz N x k x 3 neighbour coordinates (zeros = none), d2 N x k (+inf = none; k = NUM_MATCH_POINTS),
 * p_world N x 3 (the transformed scan point, Mapper.cpp:51), found N.  Original scan order; any pointer may be
 * NULL.  Tests compare these with the oracle's neighbours to pin the fast build (the index-carrying lv_fetch_knn
 * needs a capturing launch). */
int lv_fetch_neighbors(lv_ctx* ctx, float* nbr_xyz, float* d2, float* p_world, int32_t* found);
/* The single-GPU lv_update / lv_correct run ONE launch per pass (pass_kernel: the solve of the previous pass in every
 * workgroup, the search, the plane fits; no capture, degeneracy_mode 0, lanes_per_query 8, up to 16 rounds per workgroup = 1 M scan points on a 256-CU part) and keep the
 * hand-over records in LDS (also with estimate_extrinsics since round 3).  lv_set_record_dump(ctx, 1) makes that same kernel also store them to memory so that
 * lv_fetch_neighbors can pin it (one uniform branch; off by default).  lv_last_update_fused: 1 if the most recent
 * lv_update / lv_correct took the one-launch-per-pass route, 0 if the three-kernel pass (search / fit / solve). */
int lv_set_record_dump(lv_ctx* ctx, int enabled);
int lv_last_update_fused(lv_ctx* ctx);
/* Measurement passes of the most recent lv_update / lv_correct.  lv_correct(ctx, NULL) does not wait for the device: the figure
 * is then valid after the next call that does (lv_filter_get, lv_synchronize) — one host/device round trip per cycle instead
 * of two when the caller fetches the state anyway (Localizator::latest_state after ::correct, src/main.cpp:88-89). */
int lv_last_passes(lv_ctx* ctx);
/* Geometry of the one-launch-per-pass kernel for an n_scan-point scan on a part with n_cus compute units (pure host
 * logic, no GPU needed): out = {searching workgroups, search steps per round (1 or 2), rounds per workgroup,
 * 1 if one more workgroup only keeps the books (a CU is left over) else 0}.  A workgroup searches 4 tiles of 32 points
 * per step; lv_update takes this route up to 16 rounds (1 M points on a 256-CU part), with and without estimate_extrinsics
 * (round 5: its multi-round form stages a fit wavefront's rows in two halves). */
int lv_pass_geometry(size_t n_scan, int n_cus, int out[4]);
/* A/B knob: 0 = always the three-kernel pass (environment LV_FUSED_PASS sets the default at lv_create). */
int lv_set_fused_pass(lv_ctx* ctx, int enabled);
/* Tuning / test knobs by name (the environment variables LV_<NAME> set the defaults at lv_create): "fused_pass",
 * "fused_ext" (one launch per pass also with estimate_extrinsics), "fused_multi_round" (1: ... whatever the rounds per workgroup, 0: up to three — the rule of
 * round 3; default: up to 16), "keeper_by_cost", "tile_lpt", "spin_wait", "comm_fused", "small_window" / "small_insert" (windows /
 * insert batches of up to 2048 points take their one-launch forms), "multi_overlap" (0: multi-round scans fit every round
 * between two barriers), "async_relinearise" / "async_relinearise_min" (the background map rebuild, lv_map_relinearise_async),
 * "async_relinearise_pause_us" (round 6, default 100: behind every slice the worker leaves the chip empty for that many microseconds, so
 * that what the calling cycle launches meanwhile starts at once; 0: slices back to back — the rebuild is ~8 x quicker, the cycles beside
 * it 15 % slower with a p99 of 0.65 instead of 0.55 ms),
 * "async_relinearise_slice_wgs" (the worker's large grids go out in slices of that many workgroups; default 256, 0: whole grids;
 * round 5's opt-in "async_relinearise_paced_*" form was removed in round 6: LV_EINVAL like any unknown name),
 * "async_relinearise_journal_max" (default 4096: the number of map operations that may wait for the worker; a rebuild that falls
 * further behind is cancelled, the map stays as it is, and this context re-linearises stop-the-world from then on).
 * None of those changes a result beyond the summation order of the workgroup partials.  ONE option does: "fast_fit" (default
 * 0) switches pass_kernel's plane fit to hardware reciprocal / square root + one Newton step — within a few f32 ulps of the
 * exact path, NOT bit-exact against the reference (tests/test_gpu_fast_fit.py states the flips and the state difference);
 * it exists for the default 6-column configuration only and is ignored elsewhere.  LV_EINVAL for an unknown name. */
int lv_set_option(lv_ctx* ctx, const char* name, int value);
/* instrumentation (contexts created with LV_PASS_CLK=1 in the environment): for every launch of the last update
 * (MAX_NUM_ITERS + 2 of them) and every workgroup slot (capacity_wg >= CUs + 1 of them per launch; slot *n_wg is the
 * launch's designated workgroup), 16 shader-clock stamps followed by 16 wall-clock stamps (100 MHz) at its phase
 * boundaries: out holds (MAX_NUM_ITERS + 2) x capacity_wg... see scripts/pass_clocks.py for the layout. */
int lv_get_pass_clocks(lv_ctx* ctx, long long* out, int capacity_wg, int* n_wg);   /* out = NULL: *n_wg = slots per launch */
/* Mapper::match outputs: valid N (Match::is_chosen), p_world N x 3, abcd N x 4 (Normal A,B,C,D),
 * dist N (Match::distance).  Any pointer may be NULL. */
int lv_fetch_matches(lv_ctx* ctx, uint8_t* valid, float* p_world, float* abcd, float* dist);
/* calculate_H outputs as N x 12 rows / N residuals with zero rows for rejected points. */
int lv_fetch_rows(lv_ctx* ctx, double* H, double* h);

/* Localizator::calculate_H(const state_ikfom&, const Matches&, MatrixXd& H, VectorXd& h)
 *                                                 — src/Modules/Localizator.cpp:29-57
 * for caller-supplied matches: p_world n x 3 (Match::point), abcd n x 4 (Match::plane.n), dist n
 * (Match::distance).  H receives n x 12 row-major rows, h receives n residuals.  Synchronous. */
int lv_calculate_H(lv_ctx* ctx, const lv_state* x, const float* p_world, const float* abcd, const float* dist, size_t n,
                   double* H, double* h);

/* ---- instrumentation ------------------------------------------------------------------------- */
typedef struct lv_timing {
    float last_update_ms;      /* device time of the last lv_update (HIP events on the ctx stream) */
    float last_reduce_ms;      /* average device time of the dominant kernel in the last lv_update: pass_kernel (one launch
                                  per pass) or search_kernel (three-kernel pass) */
    float last_solve_ms;       /* average device time of fit_reduce_kernel + solve_kernel in the last lv_update (three-kernel pass; else 0) */
    int   last_passes;
    int   fallback_queries;    /* scan points that left the bucketed voxel levels (generic search) in the last update */
    float pass_match_ms[8];    /* device time of search_kernel per pass of the last profiled lv_update */
    float pass_solve_ms[8];    /* device time of fit_reduce_kernel + solve_kernel per pass */
    int   mailbox_resyncs;     /* updates (since lv_create) whose result mailbox failed its checksum at first sight: the host
                                  then waited with hipStreamSynchronize instead (expected: 0; an update that ends on a pass
                                  without matches carries no checksum and is not counted) */
    float pass_collective_ms[8]; /* multi-GPU forms, profiled updates: device time of the pass' collective (ncclAllGather of the
                                  workgroup partials / ncclAllReduce of the record), HIP events around it on the ctx stream */
} lv_timing;
int lv_get_timing(lv_ctx* ctx, lv_timing* out);
/* 0 = off; 1 = per-kernel HIP-event timing inside lv_update (adds event records to the stream);
 * 2 = search_kernel / fit_reduce_kernel stamp 8 shader-clock values per workgroup (phase boundaries) */
int lv_set_profiling(lv_ctx* ctx, int enabled);
/* after a pass run with lv_set_profiling(ctx, 2): out receives n_blocks x 8 clock64() stamps */
int lv_get_phase_clocks(lv_ctx* ctx, long long* out, int capacity_blocks, int* n_blocks);
/* captured passes (lv_iterate / lv_set_capture): how many scan points were decided at voxel-bucket level
 * 0, 1, 2 ([0..2]), by the generic multi-level search ([3]) or by brute force ([4]) since the last begin */
int lv_get_level_histogram(lv_ctx* ctx, int out[8]);
/* shader-clock stamps of the solve kernel's phases of the last update: 16 passes x 16 stamps (capacity >= 256) */
int lv_get_solve_clocks(lv_ctx* ctx, long long* out, int capacity);

#ifdef __cplusplus
}
#endif
#endif /* LIMOVELO_HIP_H */
