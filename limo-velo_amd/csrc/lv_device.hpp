// lv_device.hpp — shared device-side types and arithmetic of the MI355X KF-update hot path.
//
// Everything numeric here must produce the same bits as the reference's x86-64 -O3 build without
// FMA (reference CMakeLists.txt:8,16): this translation unit family is compiled with
// -ffp-contract=off, f32 sqrt/div are correctly rounded (hipcc default), and every sum is written
// in the evaluation order of the reference expression it mirrors (cited per function).
#pragma once

#include "lv_sincos.hpp"
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace lv {

constexpr int KNN = 5;              // NUM_MATCH_POINTS (config/params.yaml:48): the value every tuned path is built for
constexpr int KNN_MIN = 3, KNN_MAX = 8;   // other values take the general-K build of the three-kernel pass (lv_match.hip)
// search -> fit hand-over record of a scan point, in float4 slots: K neighbours, the world point, K distances + `found`
constexpr int qrec_slots(int k) { return k + 1 + (k + 1 + 3) / 4; }
constexpr int NS = 23;              // state dof
constexpr int NX = 26;              // state doubles (lv_state)
constexpr int SUMS_LEN = 96;        // per-pass reduction record (include/limovelo_hip.h)
constexpr int PARTIAL_GROUP = 32;  // block partials folded per group record (stage 1 of the reduction)
constexpr int N_OUT = 92;           // used entries of the record
constexpr int CELL_OFFSET = 1 << 20;
constexpr int CELL_FAR = 1 << 19;   // |cell - offset| beyond this -> brute-force path
constexpr int MAX_LEVELS = 3;             // voxel levels of the search: edge voxel_size * 2^l (the Morton key resolves three bits per level)
// Round 6: ONE level of 27-fold replicated neighbourhood buckets (level 0).  The 27-voxel block of a LEVEL-1 voxel v — level-0 voxels
// [2v - 2, 2v + 4) per axis — is tiled exactly by the EIGHT level-0 buckets centred on voxels 2v - 1 and 2v + 2 per axis (each covers
// three voxels per axis, disjoint), and every level-0 bucket belongs to exactly one such GROUP (an odd centre c is tile 0 of
// v = (c + 1) / 2, an even one tile 1 of v = (c - 2) / 2, per axis).  A (re)build lays the eight runs of a group out side by side
// (slack filled with +inf), so level 1 is ONE probe of the group table + ONE contiguous stream over level-0 storage (group_attempt,
// lv_match.hip) and stores no points of its own; a group that an insert breaks up (a run moved, a tile appeared) is marked and its
// points go on to the lists until the next re-linearisation.  Levels 2 and 3 are searched as the 27 / 216 un-replicated level-2
// voxel lists.  Rounds 1-5 replicated three levels: 4.1 KB per map point, 81 runs touched per inserted point, 266 DRAM lines per
// deleted one.
constexpr int REPL_LEVELS = 1;            // level 0: 27-fold replicated neighbourhood buckets
constexpr int SORTED_LEVELS = 1;          // ... in ascending id, 12-byte points (== REPL_LEVELS: there is no unordered replicated level any more)
constexpr int CELL_LEVEL = 2;             // level-2 voxels keep one plain point list each (level-2 block = 27 lists, level-3 block = 216)
constexpr int N_OCC = 2;                  // occupancy tables of a (re)build: [0] level-0 voxels, [1] level-2 voxels (lives on as the voxel-list table)
constexpr int OCC_CELL = 1;
__host__ __device__ constexpr int occ_level(int t) { return t == 0 ? 0 : CELL_LEVEL; }
constexpr uint64_t EMPTY_KEY = ~0ull;
constexpr int MAX_PASSES = 16;

struct RT32 {  // RotTransl (reference include/Headers/Objects.hpp:139-151), row-major R
    float R[9];
    float t[3];
};

// Constants of one measurement pass, derived on the device from the f64 state.
struct PoseConsts {
    RT32 Tc;              // X * X.I_Rt_L()                       (Mapper.cpp:51)
    RT32 back;            // S.I_Rt_L().inv() * S.inv()            (Localizator.cpp:38)
    RT32 LI;              // S.I_Rt_L()                            (Localizator.cpp:39)
    double R_inv[9];      // s.rot.conjugate().toRotationMatrix()  (Localizator.cpp:43)
    double I_R_L_inv[9];  // s.offset_R_L_I.conjugate()...         (Localizator.cpp:44)
};

// Device-resident filter state of one lv_update (esekf x_, P_, loop bookkeeping).
// Layout: [x, P_prop] is the upload region, [x .. fallback_queries] the download region.
struct KfDev {
    double x[NX];
    double P_prop[NS * NS];
    double P_post[NS * NS];
    int t;        // converge counter
    int iter;     // upstream loop index i (starts at -1)
    int done;
    int passes;
    int fallback_queries;
    int pad_[3];
    double x_prop[NX];
    double trace[MAX_PASSES * 49];
    double sums_log[MAX_PASSES * SUMS_LEN];
    PoseConsts pose;
    long long solve_clk[MAX_PASSES * 16];  // instrumentation: shader-clock stamps of solve_kernel phases
    // record-independent half of the next solve, left here by fit_reduce_kernel's extra workgroup (solve_prep)
    double prep_dxnew[NS];
    double prep_P[NS * NS];
    double prep_A1[12 * 12];
    int level_hist[8];  // captured passes only: scan points decided at bucket level 0,1,2 / level-3 lists / every id / bounded stop
    double degen_eig[MAX_PASSES * 6];  // degeneracy_mode >= 1: eigenvalues of the pose block of H^T H, per pass
    // pass_kernel (one launch per pass): what launch i hands to launch i + 1, double-buffered by launch parity — the
    // bookkeeping workgroup of launch i writes ps[(i + 1) & 1] while workgroups of the same launch may still be reading
    // ps[i & 1] (dispatch order and timing are not a contract)
    struct PassState {
        double x[NX];
        double prep_dxnew[NS];
        double prep_P[NS * NS];
        double prep_A1[12 * 12];
        int t, iter, passes, pad_;
    } ps[2];
};

// What an update starts from, handed to the FIRST search launch of the update as kernel arguments (lv_update /
// lv_iterate: the state comes from the host anyway, so no begin kernel runs): the state, its covariance and the
// pass constants the host derived from the state (compute_pose_consts is the same code on both sides, f64 -> f32
// casts and f32 sums in the same order => same bits).
struct BeginArg {
    double x[NX];
    double P[NS * NS];
    PoseConsts pose;
};

// Pinned, host-mapped mailbox of one update: x_in / P_in stage the inputs (handed to kf_begin_kernel as kernel
// arguments), the results are stored by solve_kernel straight across PCIe, so an update needs no copy kernels.
struct KfHostIO {
    double x_in[NX];
    double P_in[NS * NS];
    double x[NX];
    double P_post[NS * NS];
    int passes;
    int fallback_queries;
    // {sequence number of the last FINISHED update (low word), checksum of x / P_post / passes as stored (high word)},
    // written with ONE 64-bit store after all results.  The results are write-through stores that have retired
    // (s_waitcnt) before this word is stored, but nothing in the memory model orders them across PCIe: the host
    // verifies the checksum and falls back to hipStreamSynchronize on a mismatch (MAILBOX_UNCHECKED: always).
    unsigned long long seqcheck;
};
constexpr uint32_t MAILBOX_UNCHECKED = 0xFFFFFFFFu;
// KfDev::fallback_queries / KfHostIO::fallback_queries: bit 30 = a bounded wait inside a kernel expired (the update's results are
// not to be trusted; sticky) — the counter proper stays far below it
constexpr unsigned int KF_FAULT_BIT = 0x40000000u;
__host__ __device__ __forceinline__ uint32_t mailbox_mix(double v, uint32_t idx) {
    unsigned long long b;
#if defined(__HIP_DEVICE_COMPILE__)
    b = (unsigned long long)__double_as_longlong(v);
#else
    __builtin_memcpy(&b, &v, 8);
#endif
    return ((uint32_t)b * 31u + (uint32_t)(b >> 32)) ^ (idx * 0x9E3779B9u + 0x7F4A7C15u);
}

// Filter state resident on the device between lv_predict / lv_correct calls (row f-3).
struct FilterDev {
    double x[NX];
    double P[NS * NS];
};

struct GridLevel {
    const uint4* table;  // {key lo, key hi, start, count}
    uint32_t mask;
    uint32_t shift;      // 64 - log2(size)
};

// Per-slot bookkeeping of a bucket / cell table (parallel array): capacity of the slot's run in its pool and the
// counters of an incremental insert in flight (lv_mapinc.hpp).
struct SlotAux {
    uint32_t cap;      // entries the run can hold before it has to move
    uint32_t pending;  // entries the current insert batch will append
    uint32_t dead;     // deleted entries (tombstones) among the run's entries: what an in-place compaction would give back (bucket runs)
    uint32_t tail0;    // count before the batch
};

// The map as the search sees it.  Points are addressed by ID = insertion order; an id is never reused, a deleted
// point keeps its slot with x = +inf (distance +inf: it can never win).  The order of ids equals the order of the
// reference's map ([surviving old points] + [surviving new points]), so (distance, id) is the oracle's
// (distance, index) order; lv_fetch_knn translates ids to ranks among the living.
struct MapView {
    const float4* orig;     // xyz by id
    uint32_t m;             // living points
    uint32_t n_ids;         // ids handed out so far (range of the brute-force scan)
    float origin[3];
    float cell;             // level-0 cell edge
    float inv_cell;
    // level 0: for every voxel whose 3x3x3 block holds at least one point, the points of that block in one
    // contiguous run ("bucket") with slack behind it for appends.  bt[0].table entries are {key lo, key hi, bucket
    // start, bucket count}; ascending id (deleted entries keep their place with x = +inf), 12-byte points (what the
    // search streams) + a parallel id array (capturing launches / deletions).  The level-1 block is eight of these
    // buckets (see REPL_LEVELS): gt maps a level-1 voxel to the region {start, extent} its group's eight runs occupy in the
    // pool (extent 0: the group is no longer in one piece).
    GridLevel bt[REPL_LEVELS];
    GridLevel gt;
    const float* bxyz[SORTED_LEVELS];
    const uint32_t* bidx[SORTED_LEVELS];
    // one plain list of {x, y, z, id} records per level-2 voxel: the level-2 block is searched as the 27 lists around the
    // query's voxel, the level-3 block as the 216 lists that tile it (whole wavefronts, (distance, id) keys)
    GridLevel ct;
    const float4* cell4;
};

struct MatchParams {
    double R_inv;               // 1 / LiDAR_noise (solve_prep)
    double max_dist_plane_sq;  // MAX_DIST_PLANE * MAX_DIST_PLANE (f64, Plane.cpp:42)
    float planes_threshold;
    int estimate_extrinsics;
    int fast_fit;               // opt-in (lv_set_option "fast_fit"): approximate division / square root in the plane fit (pass_kernel, 6 columns)
};

struct DebugOut {  // all optional (nullptr = skip); indexed by ORIGINAL scan index
    uint32_t* knn_idx;  // N x 5
    float* knn_d2;      // N x 5
    uint8_t* valid;     // N
    float* p_world;     // N x 3
    float* abcd;        // N x 4
    float* dist;        // N
    double* rows;       // N x 12
    double* h;          // N
    long long* clk;     // instrumentation: 8 shader-clock stamps per workgroup (first block iteration)
    int clk_blocks;     // workgroups that have a stamp slot (wall-clock records follow at clk + clk_blocks * 8)
};

// ---------------------------------------------------------------------------------------------
// f32 / f64 3-vector algebra in Eigen 3.3 fixed-size evaluation order: x0 + (x1 + x2).
// ---------------------------------------------------------------------------------------------
__host__ __device__ __forceinline__ float dot3f(float a0, float b0, float a1, float b1, float a2, float b2) {
    float p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    float t = p1 + p2;
    return p0 + t;
}
__host__ __device__ __forceinline__ double dot3d(double a0, double b0, double a1, double b1, double a2, double b2) {
    double p0 = a0 * b0, p1 = a1 * b1, p2 = a2 * b2;
    double t = p1 + p2;
    return p0 + t;
}

// RotTransl operator*(RT1, RT2) — reference src/Objects/RotTransl.cpp:36-41
__host__ __device__ inline RT32 rt_compose(const RT32& a, const RT32& b) {
    RT32 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j)
            o.R[i * 3 + j] = dot3f(a.R[i * 3 + 0], b.R[0 * 3 + j], a.R[i * 3 + 1], b.R[1 * 3 + j], a.R[i * 3 + 2], b.R[2 * 3 + j]);
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o.t[i] = dot3f(a.R[i * 3 + 0], b.t[0], a.R[i * 3 + 1], b.t[1], a.R[i * 3 + 2], b.t[2]) + a.t[i];
    return o;
}
// Point operator*(RT, p) — reference src/Objects/RotTransl.cpp:43-48
__device__ __forceinline__ void rt_apply(const RT32& a, float px, float py, float pz, float& ox, float& oy, float& oz) {
    ox = dot3f(a.R[0], px, a.R[1], py, a.R[2], pz) + a.t[0];
    oy = dot3f(a.R[3], px, a.R[4], py, a.R[5], pz) + a.t[1];
    oz = dot3f(a.R[6], px, a.R[7], py, a.R[8], pz) + a.t[2];
}
// RotTransl::inv() — reference src/Objects/RotTransl.cpp:29-34
__host__ __device__ inline RT32 rt_inv(const RT32& a) {
    RT32 o;
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) o.R[i * 3 + j] = a.R[j * 3 + i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
        o.t[i] = dot3f(-o.R[i * 3 + 0], a.t[0], -o.R[i * 3 + 1], a.t[1], -o.R[i * 3 + 2], a.t[2]);
    return o;
}

// [UPSTREAM-RECALL Eigen Quaternion::toRotationMatrix]; q = (x,y,z,w)
__host__ __device__ inline void quat_to_rot(const double q[4], double R[9]) {
    const double x = q[0], y = q[1], z = q[2], w = q[3];
    const double tx = 2.0 * x, ty = 2.0 * y, tz = 2.0 * z;
    const double twx = tx * w, twy = ty * w, twz = tz * w;
    const double txx = tx * x, txy = ty * x, txz = tz * x;
    const double tyy = ty * y, tyz = tz * y, tzz = tz * z;
    R[0] = 1.0 - (tyy + tzz); R[1] = txy - twz;         R[2] = txz + twy;
    R[3] = txy + twz;         R[4] = 1.0 - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy;         R[7] = tyz + twx;         R[8] = 1.0 - (txx + tyy);
}

// State(const state_ikfom&, double) (reference src/Objects/State.cpp:51-62) followed by the
// per-pass transforms.  x = lv_state as 26 doubles.
__host__ __device__ inline void compute_pose_consts(const double* x, PoseConsts* out) {
    double R[9], RLI[9];
    quat_to_rot(x + 3, R);
    quat_to_rot(x + 7, RLI);
    RT32 X, LI;
#pragma unroll
    for (int i = 0; i < 9; ++i) { X.R[i] = (float)R[i]; LI.R[i] = (float)RLI[i]; }
#pragma unroll
    for (int i = 0; i < 3; ++i) { X.t[i] = (float)x[i]; LI.t[i] = (float)x[11 + i]; }
    out->Tc = rt_compose(X, LI);
    out->back = rt_compose(rt_inv(LI), rt_inv(X));
    out->LI = LI;
    double qc[4];
    qc[0] = -x[3]; qc[1] = -x[4]; qc[2] = -x[5]; qc[3] = x[6];
    quat_to_rot(qc, out->R_inv);
    qc[0] = -x[7]; qc[1] = -x[8]; qc[2] = -x[9]; qc[3] = x[10];
    quat_to_rot(qc, out->I_R_L_inv);
}

// ---- row f-2: the reference's f32 motion model (State::propagate_f, src/Objects/State.cpp:94-110) ------
struct MotionState {  // == lv_motion_state (include/limovelo_hip.h): f32 members of the reference's State
    float R[9], pos[3], vel[3], bw[3], ba[3], g[3], RLI[9], tLI[3], a[3], w[3];
    float pad_[2];
    double time;
};

// sin / cos of an f32 argument: lv_sincos.hpp (the single definition shared with the host shim)

// SO3Math::Exp<float,float>(ang_vel, dt) — reference include/Headers/Utils.hpp:30-53
__device__ inline void so3_exp_f32(const float w[3], float dt, float E[9]) {
    const float nrm = sqrtf(dot3f(w[0], w[0], w[1], w[1], w[2], w[2]));
#pragma unroll
    for (int i = 0; i < 9; ++i) E[i] = (i % 4 == 0) ? 1.f : 0.f;
    if (!((double)nrm > 0.0000001)) return;
    const float r[3] = {w[0] / nrm, w[1] / nrm, w[2] / nrm};
    const float K[9] = {0.f, -r[2], r[1], r[2], 0.f, -r[0], -r[1], r[0], 0.f};
    const float r_ang = nrm * dt;
    float sn, cs;
    sincos_f32(r_ang, sn, cs);
    const float c = (float)(1.0 - (double)cs);
    float cK[9], cKK[9];
#pragma unroll
    for (int i = 0; i < 9; ++i) cK[i] = c * K[i];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) cKK[i * 3 + j] = dot3f(cK[i * 3], K[j], cK[i * 3 + 1], K[3 + j], cK[i * 3 + 2], K[6 + j]);
#pragma unroll
    for (int i = 0; i < 9; ++i) E[i] = (E[i] + sn * K[i]) + cKK[i];
}

// State::propagate_f (State.cpp:94-110): pose part only (what de-skewing needs): R, pos after dt
__device__ inline void motion_integrate_pose(const MotionState& s, float dt, float Rn[9], float posn[3]) {
    const float wm[3] = {s.w[0] - s.bw[0], s.w[1] - s.bw[1], s.w[2] - s.bw[2]};
    float E[9];
    so3_exp_f32(wm, dt, E);
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) Rn[i * 3 + j] = dot3f(s.R[i * 3], E[j], s.R[i * 3 + 1], E[3 + j], s.R[i * 3 + 2], E[6 + j]);
    const float am[3] = {s.a[0] - s.ba[0], s.a[1] - s.ba[1], s.a[2] - s.ba[2]};
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const float v = dot3f(s.R[i * 3], am[0], s.R[i * 3 + 1], am[1], s.R[i * 3 + 2], am[2]) - s.g[i];
        posn[i] = s.pos[i] + (s.vel[i] * dt + ((0.5f * v) * dt) * dt);
    }
}

// voxel coordinates -----------------------------------------------------------------------------
__device__ __forceinline__ int cell_coord(float p, float origin, float inv_cell) {
    float f = floorf((p - origin) * inv_cell);
    // clamp in float first: out-of-range / NaN map to the clamp limits (NaN -> lower limit)
    f = fminf(fmaxf(f, -1048000.0f), 1048000.0f);
    return (int)f + CELL_OFFSET;
}
__device__ __forceinline__ uint64_t pack_cell(uint32_t ix, uint32_t iy, uint32_t iz) {
    return (uint64_t)ix | ((uint64_t)iy << 21) | ((uint64_t)iz << 42);
}
// the tile group of the level-0 bucket around voxel (cx, cy, cz): the level-1 voxel whose block it tiles and its place (0..7) there
__device__ __forceinline__ void tile_group_of(uint32_t cx, uint32_t cy, uint32_t cz, uint32_t& vx, uint32_t& vy, uint32_t& vz, int& r) {
    const uint32_t ax = (cx & 1u) ^ 1u, ay = (cy & 1u) ^ 1u, az = (cz & 1u) ^ 1u;   // odd centre: tile 0, even centre: tile 1
    vx = ax ? (cx - 2u) >> 1 : (cx + 1u) >> 1;
    vy = ay ? (cy - 2u) >> 1 : (cy + 1u) >> 1;
    vz = az ? (cz - 2u) >> 1 : (cz + 1u) >> 1;
    r = (int)(ax | (ay << 1) | (az << 2));
}
__device__ __forceinline__ uint32_t hash_cell(uint64_t key, uint32_t shift) {
    return (uint32_t)((key * 0x9E3779B97F4A7C15ull) >> shift);
}
__device__ __forceinline__ uint64_t spread21(uint32_t v) {  // 21 bits -> every third bit
    uint64_t x = v & 0x1fffff;
    x = (x | x << 32) & 0x1f00000000ffffull;
    x = (x | x << 16) & 0x1f0000ff0000ffull;
    x = (x | x << 8) & 0x100f00f00f00f00full;
    x = (x | x << 4) & 0x10c30c30c30c30c3ull;
    x = (x | x << 2) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ uint32_t compact21(uint64_t x) {
    x &= 0x1249249249249249ull;
    x = (x ^ (x >> 2)) & 0x10c30c30c30c30c3ull;
    x = (x ^ (x >> 4)) & 0x100f00f00f00f00full;
    x = (x ^ (x >> 8)) & 0x1f0000ff0000ffull;
    x = (x ^ (x >> 16)) & 0x1f00000000ffffull;
    x = (x ^ (x >> 32)) & 0x1fffff;
    return (uint32_t)x;
}
__device__ __forceinline__ uint64_t morton3(uint32_t ix, uint32_t iy, uint32_t iz) {
    return spread21(ix) | (spread21(iy) << 1) | (spread21(iz) << 2);
}

}  // namespace lv

// host-side launch declarations (defined in the .hip files) ----------------------------------------
namespace lv {
struct MapBuffers;  // lv_map.hip
}
