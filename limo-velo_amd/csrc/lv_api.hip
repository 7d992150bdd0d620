// lv_api.hip — the C-ABI of include/limovelo_hip.h on top of the HIP kernels.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <new>

#include "lv_host.hpp"

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>

namespace lv {

static thread_local char g_err[512] = "";

void set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

}  // namespace lv

using namespace lv;

struct lv_ctx {
    lv_params prm;
    int device = 0;
    hipStream_t own_stream = nullptr;
    hipStream_t side_stream = nullptr;   // lv_map_add_scan's insert runs here, beside the next cycle's prediction / window (see there)
    hipEvent_t ev_staged = nullptr;
    bool overlap_insert = true;          // lv_set_option "overlap_insert" / LV_OVERLAP_INSERT=0: the insert stays on the context's stream
    hipStream_t stream = nullptr;

    MapStore map;
    // ---- background re-linearisation of the map (row f-1; ikd-Tree rebuilds beside its searches too: the tree is constructed with
    // delete / balance criteria 0.3 / 0.6, src/Modules/Mapper.cpp:65, and rebuilds sub-trees on a second thread).  When the
    // active map wants a compaction (MapStore::wants_relinearise: a third of its id space is dead) and holds at least
    // relin_async_min points, a compacted copy of its living points is taken (one launch chain on the context's stream) and its
    // search structure is rebuilt by a WORKER THREAD on a stream of its own, while searches and inserts go on against the active
    // map; every mutation of the active map in the meantime is journaled (a device copy of the staged batch) and replayed on the
    // copy by the worker; the stores are swapped at the next map call after the worker has caught up.  The map's point set, id
    // order and hence every search result are the same as with the stop-the-world rebuild (tests/test_gpu_map_async.py).
    struct RelinEntry {
        int kind = 0;                     // 0 add (Add_Points rule), 1 add, building if empty (Mapper::add), 2 evict box, 3 evict oldest
        float4* d_pts = nullptr;          // device copy of the staged batch (owned)
        uint32_t n = 0;
        int downsample = 0;
        float box = 0.2f;
        float lo[3] = {0, 0, 0}, hi[3] = {0, 0, 0};
        int keep_inside = 0;
        uint32_t n_oldest = 0;
        hipEvent_t ready = nullptr;       // the copy has landed (recorded on the stream that made it)
        bool from_arena = false;
    };
    std::condition_variable relin_cv;
    bool relin_cancelled = false;
    size_t relin_want = 0;                // id slots the copy is allocated for (living points + slack)
    void* relin_arena = nullptr;          // device memory the journal's batch copies are carved from (allocated once, by the worker)
    size_t relin_arena_bytes = 64u << 20, relin_arena_used = 0;
    MapStore relin_shadow;
    std::thread relin_worker;
    std::mutex relin_mu;
    std::deque<RelinEntry> relin_journal;
    std::atomic<int> relin_state{0};      // 0 idle, 4 worker allocating, 5 allocated (waiting for the snapshot), 1 rebuilding / replaying, 2 ready
                                          // (worker caught up), 3 failed — written under relin_mu
    std::string relin_error;
    hipStream_t relin_stream = nullptr;
    hipEvent_t relin_snapshot = nullptr;
    bool relin_snap_side = false;         // the snapshot was enqueued on the side stream and the context's stream has not been ordered behind it yet (relin_order)
    bool relin_async = true;              // lv_set_option "async_relinearise" / LV_ASYNC_RELINEARISE=0: always stop-the-world
    size_t relin_async_min = 200000;      // smaller maps rebuild in ~2 ms: not worth a thread
    uint64_t relin_started = 0, relin_swapped = 0;
    std::atomic<uint64_t> relin_replayed{0};   // (written by the worker, read by lv_map_rebuild_status)
    size_t relin_journal_max = 4096;       // journal entries beyond which a rebuild that cannot keep up is given up: the
                                           // copy is cancelled and the active map takes the stop-the-world path when it needs one
    uint32_t relin_pause_us = 100;        // LV_RELIN_PAUSE_US / "async_relinearise_pause_us": the worker's slices are spaced by that long (lv_map.hip launch_sliced):
                                          // a cycle beside the rebuild 0.25 instead of 0.29 ms, p99 0.53-0.57 instead of 0.62-0.68, the rebuild ~8 x longer in wall time
                                          // (profiles/experiments_r06/rebuild_spaced_slices_ab.txt); 0: slices back to back
    uint32_t relin_slice_wgs = 256;       // LV_RELIN_SLICE_WGS: slice size of the worker's large launches (0: whole grids).  (Round 5's opt-in PACED form — the
                                          // grids as at most 32 looping 1024-thread workgroups — is gone: p99 0.5 instead of 0.6 ms, but two 4.6 ms cycles in 5 of
                                          // 17 replays that plain slices never showed, cause not found: profiles/experiments_r05/async_rebuild.txt §10-11)
    bool relin_test_race = false;         // lv_set_option "async_relinearise_test_race": see relin_journal_add
    int relin_test_delay_ms = 0;          // lv_set_option "async_relinearise_test_delay_ms": the worker pauses between rebuild and replay (tests)

    ScanStore scan;
    CloudStore cloud;   // row f-4: device-resident LiDAR buffer
    float4* h_stage = nullptr;  // pinned upload staging
    size_t h_stage_cap = 0;
    MotionState* h_states_ring = nullptr;   // pinned: the motion states of lv_scan_deskew_window, 8 slots of 64 (a copy out of the
    int states_slot = 0;                    // caller's pageable memory blocks the host; every cycle synchronises at least once, so
                                            // a slot is free again long before its turn comes round)

    KfDev* d_kf = nullptr;
    FilterDev* d_filter = nullptr;  // x, P resident between lv_predict / lv_correct (row f-3)
    FilterDev* h_filter = nullptr;  // pinned staging
    bool filter_set = false;
    int state_src = 0;          // who produced the latest state: 0 nobody yet, 1 the resident filter (lv_filter_set / lv_predict /
                                // lv_correct), 2 lv_update (d_kf->x) — lv_map_add_scan transforms the scan with that state
    KfDev* h_kf = nullptr;  // pinned mirror (logs / trace downloads)
    KfHostIO* h_io = nullptr;  // pinned, host-mapped mailbox: update inputs and results (no copy kernels)
    KfHostIO* d_io = nullptr;  // its device address
    int update_seq = 0;        // number of the update in flight (the finishing pass echoes it into h_io->seq)
    // lv_predict calls are queued (same Q, up to PREDICT_BATCH steps) and launched together by whatever needs the filter next
    double pred_Q[144] = {};
    double pred_steps[PREDICT_BATCH][7] = {};
    int pred_n = 0;
    bool pred_src_kf = false;         // the first queued step reads the posterior from kf (filter_in_kf at the time it was queued)
    bool batch_predict = true;        // lv_set_option "batch_predict" / LV_BATCH_PREDICT=0: one launch per lv_predict
    bool filter_in_kf = false;        // the resident filter has not been copied out of kf->x / kf->P_post yet (see materialise_filter)
    bool filter_in_mailbox = false;   // the resident filter == the results in the mailbox (set by lv_correct, cleared by whatever changes the filter)
    bool mail_filter = true;          // lv_filter_get reads them from there (lv_set_option "mail_filter" 0: always copy)
    bool spin_wait = true;     // lv_update_end polls the mailbox before falling back to hipStreamSynchronize (LV_SPIN_WAIT=0: off)
    double* d_partials = nullptr;
    double* d_groups = nullptr;    // group records (reduce stage 1)
    int ngroups = 0;
    double* d_sums = nullptr;      // record in use (own or caller-provided)
    double* d_sums_own = nullptr;
    double* h_sums = nullptr;  // pinned
    int max_blocks = 1024;
    int grid = 1;
    void* comm = nullptr;          // RCCL communicator (lv_comm_init): every pass all-reduces the record
    int comm_rank = 0, comm_world = 1;
    bool fold_direct = false;      // this pass: solve_kernel reads the block partials directly
    bool tile_lpt = true;          // dispatch the search tiles farthest-first (LV_TILE_LPT=0: plain order, A/B knob)
    float4* d_qrec = nullptr;      // search -> fit hand-over: 8 float4 planes of qstride entries (one record per scan point)
    uint32_t qstride = 0;
    // one launch per pass (pass_kernel, lv_pass_dev.hpp): compact workgroup partials, ping-pong by pass parity
    double* d_cpart[2] = {nullptr, nullptr};
    // multi-GPU form of the one-launch-per-pass update: every rank's partials gathered on every rank (ncclAllGather, in place)
    double* d_gather[2] = {nullptr, nullptr};
    size_t gather_cap = 0;            // doubles per buffer
    size_t comm_shard_max = 0;        // largest shard of the CURRENT scan over the ranks (lv_comm_set_shard_max); 0: unknown
    bool comm_fused = true;           // lv_set_comm_fused / LV_COMM_FUSED=0: always the three-kernel pass + all-reduce with a communicator
    PeerSet peer;                      // lv_comm_peer_export / _init: the partials pulled out of the other ranks' peer-mapped buffers (lv_peer.hip)
    lv_gather_fn gather_cb = nullptr;  // lv_comm_set_host_gather: the partials of the ranks exchanged by the caller through host memory
    void* gather_user = nullptr;       //   (test / bring-up transport of the one-launch-per-pass multi-rank form; no librccl involved)
    double* h_gather = nullptr;        //   pinned staging, world x slot doubles
    size_t h_gather_cap = 0;
    uint32_t* d_wgcost[2] = {nullptr, nullptr};   // per searching workgroup: how long its search + fits took (picks the next bookkeeper)
    int pass_max_wg = 256;         // search workgroups of pass_kernel: all resident at once (one 1024-thread workgroup per CU)
    bool fused_pass = true;        // LV_FUSED_PASS=0: the three-kernel pass (search / fit / solve) also where pass_kernel applies
    bool record_dump = false;      // lv_set_record_dump: pass_kernel also writes its hand-over records to d_qrec (lv_fetch_neighbors)
    bool last_update_fused = false;
    long long* d_pclk = nullptr;   // LV_PASS_CLK=1: phase stamps of pass_kernel's workgroups (lv_get_pass_clocks)
    int pclk_wg = 0;
    bool keeper_by_cost = true;       // LV_KEEPER_BY_COST=0: the last searching workgroup always keeps the books (A/B knob)
    int fused_multi_round = -1;       // LV_FUSED_MULTI: 1 = pass_kernel whatever the rounds per workgroup, 0 = up to three (the rule of
                                      // round 3), -1 (default) = up to PK_DEFAULT_MAX_ROUNDS (three with estimate_extrinsics)
    bool fused_ext = true;         // LV_FUSED_EXT=0: the three-kernel pass with estimate_extrinsics (one launch per pass measured 22.1 k vs 21.4 k it/s, r03)

    // capture (debug / API-parity) buffers, sized for the current scan
    bool capture = false;
    size_t cap_n = 0;
    DebugOut dbg{};
    bool dbg_valid = false;
    bool qrec_valid = false;       // d_qrec holds the records of a pass over the CURRENT scan (lv_fetch_neighbors)

    bool begin_pending = false;    // the update's state waits in h_begin for the first search launch (no begin kernel)
    bool filter_host = false;      // lv_filter_set just wrote the filter: it lives in h_filter (pinned) until something needs it on the
                                   // device — the next lv_correct does not: the prior rides in its first launch's arguments
    bool filter_up_pending = false;   // an upload out of h_filter may still be in flight (ev_filter_up)
    hipEvent_t ev_filter_up = nullptr;
    BeginArg h_begin;
    int fallback_base = 0;         // device counter value before the update in flight (the device never resets it)
    long mailbox_resyncs = 0;      // updates whose mailbox checksum did not match at first sight (stream synchronised instead)

    int pass_index = -1;           // index of the pass being enqueued by lv_update's three-kernel loop (events of its collective)
    bool coll_timed = false;       // the last profiled update recorded events around its collectives
    bool fast_fit = false;       // lv_set_option "fast_fit": the opt-in approximate plane fit of pass_kernel (v_rcp / v_sqrt + Newton; NOT bit-exact)
    bool multi_overlap = true;   // multi-round scans: plane fits beside the next round's search (LV_MULTI_OVERLAP=0: round 3's barrier form)
    bool in_update = false;
    bool want_log = false;         // download trace / per-pass sums at lv_update_end
    int passes_issued = 0;

    bool profiling = false;
    bool phase_clocks = false;     // lv_set_profiling(ctx, 2): per-workgroup phase stamps of the match kernel
    long long* d_clk = nullptr;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    std::vector<hipEvent_t> ev_pass;  // 3 per pass: before match kernel, after it, after reduce_partials + solve
    hipEvent_t ev_mid = nullptr;      // set while a profiled pass is in flight: recorded right after the match kernel
    std::vector<hipEvent_t> ev_coll;  // 2 per pass: around the pass' collective (multi-GPU forms, profiled updates)
    lv_timing timing{};
};

namespace {

// LV_SLOW_CALL_MS=<ms> in the environment: every entry point that lasted longer than that reports itself on stderr (a
// diagnostic for latency hiccups: which call of a cycle stalled, profiles/experiments_r05/async_rebuild.txt)
struct SlowCall {
    const char* name;
    std::chrono::steady_clock::time_point t0;
    static double limit_ms() { static const double v = [] { const char* e = getenv("LV_SLOW_CALL_MS"); return e ? atof(e) : 0.0; }(); return v; }
    explicit SlowCall(const char* n) : name(n) { if (limit_ms() > 0.0) t0 = std::chrono::steady_clock::now(); }
    ~SlowCall() {
        if (limit_ms() > 0.0) {
            const double ms = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
            if (ms > limit_ms()) fprintf(stderr, "[limovelo_hip] slow call: %s took %.3f ms\n", name, ms);
        }
    }
};
#define LV_CHECK_CTX(ctx)                        \
    SlowCall _slow_call(__func__);               \
    do {                                         \
        if (!(ctx)) {                            \
            set_error("null context");           \
            return LV_EINVAL;                    \
        }                                        \
        hipError_t _e = hipSetDevice((ctx)->device); \
        if (_e != hipSuccess) {                  \
            set_error("hipSetDevice(%d): %s", (ctx)->device, hipGetErrorString(_e)); \
            return LV_EHIP;                      \
        }                                        \
    } while (0)

// The peer-mapped exchange (lv_peer.hip) failed in an earlier launch of this context (a rank of the node gave up waiting for
// another, or met a rank that had): whatever those launches computed is built on stale partials.  Nothing of it is adopted
// — the resident filter is declared unset — and every later call of the data path fails until the exchange is torn down
// (lv_comm_destroy) and the filter re-seeded.
static int peer_poisoned(lv_ctx* c) {
    if (!c->peer.active || !peer_failed(c->peer)) return LV_OK;
    c->filter_set = false;
    c->filter_in_kf = false;
    c->filter_in_mailbox = false;
    c->pred_n = 0;
    c->in_update = false;
    set_error("peer-mapped gather: a rank of the node did not publish its partials in time (or reported a failed exchange): the update "
              "was not adopted; lv_comm_destroy, then lv_filter_set / a new exchange");
    return LV_ESTATE;
}
#define LV_CHECK_PEER(c)                 \
    do {                                 \
        int _rq = peer_poisoned(c);      \
        if (_rq) return _rq;             \
    } while (0)

// an incremental insert leaves its outcome in flight (MapStore::settle): pick it up before the map's bookkeeping is used
#define LV_SETTLE_MAP(c)                               \
    do {                                               \
        int _rs = (c)->map.settle((c)->stream);        \
        if (_rs) return _rs;                           \
    } while (0)

int ensure_stage(lv_ctx* c, size_t n) {
    if (n <= c->h_stage_cap) return LV_OK;
    size_t cap = c->h_stage_cap ? c->h_stage_cap : 4096;
    while (cap < n) cap *= 2;
    // a copy out of the old buffer may still be pending on the (possibly caller-supplied, non-blocking) stream
    LV_HIP(hipStreamSynchronize(c->stream));
    if (c->h_stage) hipHostFree(c->h_stage);
    c->h_stage = nullptr;
    c->h_stage_cap = 0;
    LV_HIP(hipHostMalloc((void**)&c->h_stage, cap * sizeof(float4), hipHostMallocDefault));
    c->h_stage_cap = cap;
    return LV_OK;
}

inline void read_xyz(const void* base, size_t stride, size_t i, float& x, float& y, float& z) {
    const float* p = reinterpret_cast<const float*>(reinterpret_cast<const char*>(base) + i * stride);
    x = p[0]; y = p[1]; z = p[2];
}

void free_capture(lv_ctx* c) {
    hipFree(c->dbg.knn_idx); hipFree(c->dbg.knn_d2); hipFree(c->dbg.valid); hipFree(c->dbg.p_world);
    hipFree(c->dbg.abcd); hipFree(c->dbg.dist); hipFree(c->dbg.rows); hipFree(c->dbg.h);
    c->dbg = DebugOut{};
    c->cap_n = 0;
    c->dbg_valid = false;
    c->qrec_valid = false;
}

int ensure_capture(lv_ctx* c, size_t n) {
    if (n <= c->cap_n && c->dbg.knn_idx) return LV_OK;
    free_capture(c);
    size_t cap = n ? n : 1;
    LV_HIP(hipMalloc(&c->dbg.knn_idx, cap * c->prm.NUM_MATCH_POINTS * sizeof(uint32_t)));
    LV_HIP(hipMalloc(&c->dbg.knn_d2, cap * c->prm.NUM_MATCH_POINTS * sizeof(float)));
    LV_HIP(hipMalloc(&c->dbg.valid, cap));
    LV_HIP(hipMalloc(&c->dbg.p_world, cap * 3 * sizeof(float)));
    LV_HIP(hipMalloc(&c->dbg.abcd, cap * 4 * sizeof(float)));
    LV_HIP(hipMalloc(&c->dbg.dist, cap * sizeof(float)));
    LV_HIP(hipMalloc(&c->dbg.rows, cap * 12 * sizeof(double)));
    LV_HIP(hipMalloc(&c->dbg.h, cap * sizeof(double)));
    c->cap_n = cap;
    return LV_OK;
}

// eigenvalues (cyclic Jacobi, 8 sweeps, unsorted: the order degeneracy_stage of lv_solve_dev.hpp leaves them in) of the 6 x 6
// pose block of the H^T H packed in a 96-double sums record
static void host_pose_eigenvalues(const double* rec, double eig[6]) {
    double A[6][6];
    int idx = 0;
    for (int i = 0; i < 12; ++i)
        for (int j = i; j < 12; ++j) {
            if (j < 6) { A[i][j] = rec[idx]; A[j][i] = rec[idx]; }
            ++idx;
        }
    for (int sweep = 0; sweep < 8; ++sweep)
        for (int p = 0; p < 5; ++p)
            for (int q = p + 1; q < 6; ++q) {
                const double apq = A[p][q];
                if (std::fabs(apq) < 1e-300) continue;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * apq);
                const double t = (theta >= 0.0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 6; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq;
                    A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 6; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk;
                    A[q][k] = s * apk + c * aqk;
                }
            }
    for (int i = 0; i < 6; ++i) eig[i] = A[i][i];
}

void unpack_sums(const double* rec, lv_sums* out) {
    int idx = 0;
    for (int i = 0; i < 12; ++i)
        for (int j = i; j < 12; ++j) {
            out->HTH[i * 12 + j] = rec[idx];
            out->HTH[j * 12 + i] = rec[idx];
            ++idx;
        }
    for (int i = 0; i < 12; ++i) out->HTh[i] = rec[78 + i];
    out->n_valid = (int64_t)rec[90];
    out->sum_h2 = rec[91];
}

// from_host: x / P_prop wait in the pinned mailbox (lv_update_begin), otherwise they are in d_kf already (copied
// from the resident filter): take them over, derive the pass constants
// The resident filter after lv_correct IS kf->x / kf->P_post (filter_in_kf): the next lv_predict reads it from there and the
// next lv_correct starts from there; only something that needs it in d_filter, or that is about to overwrite kf (an update
// by value, lv_iterate), copies it out first — in the 100 Hz cycle that launch never happens.
// The filter as lv_filter_set left it (pinned host memory) goes to d_filter: needed by whatever reads the resident filter on
// the device other than an lv_correct that starts right from it.
static int filter_to_device(lv_ctx* c) {
    if (!c->filter_host) return LV_OK;
    c->filter_host = false;
    LV_HIP(hipMemcpyAsync(c->d_filter, c->h_filter, sizeof(FilterDev), hipMemcpyHostToDevice, c->stream));
    if (!c->ev_filter_up) LV_HIP(hipEventCreateWithFlags(&c->ev_filter_up, hipEventDisableTiming));
    LV_HIP(hipEventRecord(c->ev_filter_up, c->stream));
    c->filter_up_pending = true;
    return LV_OK;
}
static int flush_predicts(lv_ctx* c) {
    if (c->pred_n == 0) return LV_OK;
    { int ru = filter_to_device(c); if (ru) return ru; }
    const int n = c->pred_n;
    c->pred_n = 0;
    const KfDev* src = c->pred_src_kf ? c->d_kf : nullptr;
    c->pred_src_kf = false;
    return launch_predict(c->stream, c->d_filter, src, c->pred_Q, n, c->pred_steps);
}
#define LV_FLUSH_PREDICTS(c)             \
    do {                                 \
        int _rp = flush_predicts(c);     \
        if (_rp) return _rp;             \
    } while (0)

static int materialise_filter(lv_ctx* c) {
    if (!c->filter_in_kf) return LV_OK;
    c->filter_in_kf = false;
    return launch_kf_to_filter(c->stream, c->d_kf, c->d_filter);
}

int begin_device(lv_ctx* c, const double* x_host, bool defer, bool from_filter) {
    LV_FLUSH_PREDICTS(c);   // (a queued prediction may still have to read the posterior from kf, which this update is about to overwrite)
    // a filter that still waits on the host goes to d_filter now: kf_begin_kernel reads it there (from_filter), and an update by
    // value / lv_iterate / lv_calculate_H uses kf as its working copy but leaves the resident filter alone (the lv_correct that
    // CONSUMES the host copy through its launch arguments has cleared filter_host before it calls)
    { int ru = filter_to_device(c); if (ru) return ru; }
    if (!from_filter) { int rm = materialise_filter(c); if (rm) return rm; }
    c->begin_pending = false;
    c->filter_in_mailbox = false;   // (the mailbox is about to receive this update's results)
    if (c->capture) LV_HIP(hipMemsetAsync(c->d_kf->level_hist, 0, sizeof(int) * 8, c->stream));   // instrumentation of capturing passes
    if (defer && x_host && c->scan.n > 0 && c->map.view.m > 0) {
        // the first search launch installs everything (x_host: x followed by P_prop, KfHostIO layout)
        std::memcpy(c->h_begin.x, x_host, sizeof(c->h_begin.x));
        std::memcpy(c->h_begin.P, x_host + NX, sizeof(c->h_begin.P));
        compute_pose_consts(c->h_begin.x, &c->h_begin.pose);
        c->begin_pending = true;
    } else {
        int rc = launch_kf_begin(c->stream, c->d_kf, c->d_io, x_host, from_filter ? c->d_filter : nullptr, (from_filter && c->filter_in_kf) ? 1 : 0);
        if (rc) return rc;
        if (from_filter) c->filter_in_kf = false;   // (kf->P_post is about to become this update's working copy: the filter proper is d_filter again ... after kf_to_filter / the next predict)
    }
    c->grid = fit_grid_size(c->scan.n, c->max_blocks);
    if ((uint32_t)c->scan.n > c->qstride) {
        uint32_t cap = c->qstride ? c->qstride : 4096;
        while (cap < (uint32_t)c->scan.n) cap *= 2;
        LV_HIP(hipStreamSynchronize(c->stream));
        hipFree(c->d_qrec);
        c->d_qrec = nullptr;
        c->qstride = 0;
        LV_HIP(hipMalloc(&c->d_qrec, (size_t)cap * qrec_slots(c->prm.NUM_MATCH_POINTS) * sizeof(float4)));
        c->qstride = cap;
    }
    return LV_OK;
}

int begin_common(lv_ctx* c, const lv_state* x, const double* P, bool defer = true) {
    KfHostIO* io = c->h_io;
    std::memcpy(io->x_in, x, sizeof(double) * NX);
    if (P) {
        std::memcpy(io->P_in, P, sizeof(double) * NS * NS);
    } else {
        for (int i = 0; i < NS * NS; ++i) io->P_in[i] = (i / NS == i % NS) ? 1.0 : 0.0;
    }
    c->update_seq = (c->update_seq + 1) & 0x3fffffff;
    return begin_device(c, io->x_in, defer, false);   // x_in and P_in (contiguous) ride in the kernel arguments
}

int pass_solve(lv_ctx* c, bool from_groups);

int pass_reduce(lv_ctx* c, bool finalize) {
    MatchParams mp;
    mp.R_inv = 1.0 / c->prm.LiDAR_noise;
    mp.max_dist_plane_sq = c->prm.MAX_DIST_PLANE * c->prm.MAX_DIST_PLANE;
    mp.planes_threshold = c->prm.PLANES_THRESHOLD;
    mp.estimate_extrinsics = c->prm.estimate_extrinsics;
    DebugOut dbg{};
    if (c->capture) {
        int rc = ensure_capture(c, c->scan.n);
        if (rc) return rc;
        dbg = c->dbg;
        c->dbg_valid = true;
    }
    if (c->phase_clocks) {
        if (!c->d_clk) LV_HIP(hipMalloc(&c->d_clk, (size_t)(c->max_blocks + 8) * 16 * sizeof(long long)));
        dbg.clk = c->d_clk;
        dbg.clk_blocks = c->grid;
    }
    int rc = LV_OK;
    c->qrec_valid = c->scan.n > 0;
    if (c->scan.n > 0) {
        rc = launch_search(c->stream, c->prm.lanes_per_query, c->map.view, c->scan.d_sorted, c->scan.n, c->d_kf, c->d_qrec,
                           c->qstride, c->tile_lpt ? c->scan.d_tile_order : nullptr, c->scan.n_tiles, mp.max_dist_plane_sq, dbg,
                           c->begin_pending ? &c->h_begin : nullptr, c->d_io, c->prm.NUM_MATCH_POINTS);
        c->begin_pending = false;
    }
    if (rc) return rc;
    if (c->ev_mid) LV_HIP(hipEventRecord(c->ev_mid, c->stream));   // profiled pass: brackets the search kernel
    rc = launch_fit_reduce(c->stream, c->d_qrec, c->qstride, c->scan.n, c->d_kf, mp, c->d_partials, c->grid, dbg, c->prm.NUM_MATCH_POINTS);
    if (rc) return rc;
    // few enough block partials (a 64k-point scan leaves 256): solve_kernel folds them itself in one memory
    // round trip; otherwise stage 1 of the reduction runs as its own kernel
    c->fold_direct = !finalize && c->grid <= solve_direct_records();
    if (c->fold_direct) return LV_OK;
    if (finalize && c->grid <= solve_direct_records())   // the rank's record straight from its block partials
        return launch_reduce_final(c->stream, c->d_partials, c->grid, c->d_sums, c->d_kf);
    rc = launch_reduce_groups(c->stream, c->d_partials, c->grid, c->d_groups, &c->ngroups, c->d_kf);
    if (rc) return rc;
    if (finalize) return launch_reduce_final(c->stream, c->d_groups, c->ngroups, c->d_sums, c->d_kf);
    return LV_OK;
}

// one measurement pass + solve as lv_update / lv_correct run it.  With a communicator (multi-GPU): the rank's
// record is all-reduced in place on the stream and every rank solves from the identical record.
int pass_full(lv_ctx* c) {
    // (a rank without points still runs the pass: its partials are zeros and solve_prep must run)
    int rc = pass_reduce(c, c->comm != nullptr);
    if (rc) return rc;
    if (c->comm) {
        const int i = c->pass_index;
        const bool timed = c->profiling && i >= 0 && (size_t)(2 * i + 1) < c->ev_coll.size();
        if (timed) LV_HIP(hipEventRecord(c->ev_coll[2 * i], c->stream));
        rc = comm_allreduce_record(c->comm, c->d_sums, c->stream);
        if (rc) return rc;
        if (timed) { LV_HIP(hipEventRecord(c->ev_coll[2 * i + 1], c->stream)); c->coll_timed = true; }
    }
    return pass_solve(c, c->comm == nullptr);
}

int pass_solve(lv_ctx* c, bool from_groups) {
    SolveParams sp;
    sp.R = c->prm.LiDAR_noise;
    sp.R_inv = 1.0 / c->prm.LiDAR_noise;
    for (int i = 0; i < NS; ++i) sp.limits[i] = c->prm.LIMITS[i];
    sp.maximum_iter = c->prm.MAX_NUM_ITERS;
    sp.estimate_extrinsics = c->prm.estimate_extrinsics;
    sp.seq = c->update_seq;
    sp.degeneracy_mode = c->prm.degeneracy_mode;
    sp.degeneracy_threshold = c->prm.degeneracy_threshold;
    if (from_groups && c->fold_direct) return launch_solve(c->stream, c->d_kf, c->d_io, c->d_partials, c->grid, c->d_sums, sp);
    if (from_groups) return launch_solve(c->stream, c->d_kf, c->d_io, c->d_groups, c->ngroups, c->d_sums, sp);
    return launch_solve(c->stream, c->d_kf, c->d_io, c->d_sums, 1, nullptr, sp);
}

// One launch per pass (pass_kernel) applies to the plain single-GPU update: no capture / phase clocks, no
// communicator (the all-reduce sits between fit and solve), no degeneracy stage, 8 lanes per scan point.
// the scan size that fixes pass_kernel's geometry: the local scan, or with a communicator the largest shard over the ranks
// (every rank launches the same grid; workgroups without a tile contribute zero partials)
// KF_FAULT_BIT (a bounded wait inside pass_kernel expired) rides in the device's fallback counter.  It fails the update that
// raised it — and only that one: the host clears the bit on the device once it has reported it (the stream is idle at that
// point), so the context stays usable, and everything that uses the word as a counter masks the bit out.
static int report_kf_fault(lv_ctx* c, int raw) {
    const int cnt = raw & (int)~KF_FAULT_BIT;
    LV_HIP(hipStreamSynchronize(c->stream));
    LV_HIP(hipMemcpy(&c->d_kf->fallback_queries, &cnt, sizeof(int), hipMemcpyHostToDevice));
    c->h_io->fallback_queries = cnt;
    c->fallback_base = cnt;
    set_error("a bounded wait inside pass_kernel expired: this update's results are invalid (the flag has been cleared; the context stays usable)");
    return LV_EHIP;
}
static inline int fallback_count(int raw) { return raw & (int)~KF_FAULT_BIT; }

static inline bool multi_rank(const lv_ctx* c) { return c->comm != nullptr || c->gather_cb != nullptr || c->peer.active; }
// transports that only carry the one-launch form's partials (no 96-double all-reduce behind them)
static inline bool gather_only(const lv_ctx* c) { return c->gather_cb != nullptr || c->peer.active; }
// doubles per compact workgroup partial (PassDims<W>::OW, lv_pass_dev.hpp): 32 for the 6-column rows, 96 with extrinsics
static inline size_t partial_width(const lv_ctx* c) { return c->prm.estimate_extrinsics ? 96u : 32u; }
uint32_t pass_geometry_points(const lv_ctx* c) { return multi_rank(c) ? (uint32_t)c->comm_shard_max : c->scan.n; }

bool pass_fused_applies(const lv_ctx* c) {
    // (multi-round scans run pass_kernel<.., MULTI>: a round's plane fits run on four wavefronts beside the next round's search
    // on the other twelve; measured r04 per update: 131 072 points 191.8 us, 196 608 249.1, 262 144 304.9 against 329.3 with
    // three kernels.  The estimate_extrinsics build keeps round 3's form — a barrier either side of every round's fits — and
    // its limit of three rounds: 262 144 points 356 vs 352 us)
    if (multi_rank(c)) {
        // with a communicator: the caller has told the largest shard of this scan (lv_comm_set_shard_max), librccl has
        // ncclAllGather (or the caller exchanges the partials itself: lv_comm_set_host_gather), the gather buffers are in
        // place; a rank without points still runs every launch
        if (!c->comm_fused || c->comm_shard_max == 0 || c->scan.n > c->comm_shard_max || (c->comm && !comm_has_allgather()) ||
            !c->d_gather[0])
            return false;
    } else if (c->scan.n == 0) {
        return false;
    }
    int nwg = 0, rounds = 0, steps = 0, dedicated = 0;
    pass_grid_size(pass_geometry_points(c), c->pass_max_wg, &nwg, &steps, &rounds, &dedicated);
    // (round 5: estimate_extrinsics takes the overlapped multi-round form as well — its rows staged in two halves — so the
    // limit of three rounds it had is gone: xaloc.yaml's configuration stays on one launch per pass up to the same sixteen)
    const int max_rounds = c->fused_multi_round > 0 ? INT32_MAX : c->fused_multi_round == 0 ? 3 : PK_DEFAULT_MAX_ROUNDS;
    if (rounds > max_rounds) return false;
    if (multi_rank(c) && (size_t)nwg * (size_t)c->comm_world * partial_width(c) > c->gather_cap) return false;
    // (degeneracy_mode 1 only REPORTS eigenvalues: derived on the host from the logged sums, lv_get_degeneracy_values)
    return c->fused_pass && !c->capture && !c->phase_clocks && c->prm.degeneracy_mode <= 1 && c->prm.NUM_MATCH_POINTS == KNN &&
           c->prm.lanes_per_query == 8 && (c->prm.estimate_extrinsics == 0 || c->fused_ext) && c->map.view.m > 0;
}

// The whole iterated update as npass + 1 launches: launch i = [solve of pass i-1 in every workgroup] + pass i, the
// closing launch (one workgroup) = the solve of the last pass.  begin_device() ran before: either the state waits in
// h_begin (begin_pending: it rides in the first launch's kernel arguments) or a begin kernel installed it in d_kf.
int update_fused(lv_ctx* c) {
    const int npass = c->prm.MAX_NUM_ITERS + 1;
    PassLaunch pl{};
    pl.map = &c->map.view;
    pl.scan = c->scan.d_sorted;
    pl.tile_order = (c->tile_lpt && c->scan.tile_points == 32u && c->scan.n_tiles == (c->scan.n + 31u) / 32u) ? c->scan.d_tile_order : nullptr;
    pl.n = c->scan.n;
    pl.kf = c->d_kf;
    pl.io = c->d_io;
    pl.sums_out = c->d_sums;
    pl.mp.R_inv = 1.0 / c->prm.LiDAR_noise;
    pl.mp.max_dist_plane_sq = c->prm.MAX_DIST_PLANE * c->prm.MAX_DIST_PLANE;
    pl.mp.planes_threshold = c->prm.PLANES_THRESHOLD;
    pl.mp.estimate_extrinsics = c->prm.estimate_extrinsics;
    pl.mp.fast_fit = c->fast_fit && !c->prm.estimate_extrinsics;
    pl.sp.R = c->prm.LiDAR_noise;
    pl.sp.R_inv = 1.0 / c->prm.LiDAR_noise;
    for (int i = 0; i < NS; ++i) pl.sp.limits[i] = c->prm.LIMITS[i];
    pl.sp.maximum_iter = c->prm.MAX_NUM_ITERS;
    pl.sp.estimate_extrinsics = c->prm.estimate_extrinsics;
    pl.sp.seq = c->update_seq;
    pl.sp.degeneracy_mode = 0;
    pl.sp.degeneracy_threshold = c->prm.degeneracy_threshold;
    int nwg = 0, rounds = 0, steps = 0, dedicated = 0;
    pass_grid_size(pass_geometry_points(c), c->pass_max_wg, &nwg, &steps, &rounds, &dedicated);
    const bool gathered = multi_rank(c);                      // multi-GPU: the partials of all ranks, gathered after every launch
    const size_t slot = (size_t)nwg * partial_width(c);       // doubles per rank in the gather buffers (compact records)
    pl.multi_overlap = c->multi_overlap;
    pl.qrec = c->record_dump ? c->d_qrec : nullptr;
    c->pclk_wg = nwg + dedicated;   // (the last slot is the bookkeeping workgroup either way)
    pl.qstride = c->qstride;
    c->qrec_valid = c->record_dump;
    c->last_update_fused = true;
    for (int i = 0; i <= npass; ++i) {
        const bool closing = i == npass;
        pl.mode = i == 0 ? (c->begin_pending ? 0 : 2) : 1;
        pl.recs_in = gathered ? c->d_gather[(i + 1) & 1] : c->d_cpart[(i + 1) & 1];
        pl.part_out = gathered ? c->d_gather[i & 1] + (size_t)c->comm_rank * slot : c->d_cpart[i & 1];
        pl.nrec = gathered ? nwg * c->comm_world : nwg;
        pl.cost_in = (i > 0 && c->keeper_by_cost) ? c->d_wgcost[(i + 1) & 1] : nullptr;
        pl.cost_out = c->d_wgcost[i & 1];
        pl.nwg = nwg;
        pl.rounds = closing ? 0 : rounds;
        pl.steps = steps;
        pl.dedicated = dedicated;
        pl.launch = i;
        pl.clk = c->d_pclk ? c->d_pclk + (size_t)i * (c->pass_max_wg + 1) * pass_clock_words() : nullptr;
        if (c->profiling && !closing) LV_HIP(hipEventRecord(c->ev_pass[3 * i + 0], c->stream));
        int rc = launch_pass(c->stream, pl, (i == 0 && c->begin_pending) ? &c->h_begin : nullptr);
        c->begin_pending = false;
        if (rc) return rc;
        if (c->profiling && !closing) LV_HIP(hipEventRecord(c->ev_pass[3 * i + 1], c->stream));   // (the pass kernel alone)
        if (gathered && !closing) {
            if (c->profiling) LV_HIP(hipEventRecord(c->ev_coll[2 * i], c->stream));
            if (c->gather_cb) {
                // the caller's transport: this rank's slot to pinned host memory, the callback fills in the other ranks' slots
                // (it blocks until they are there), everything back — the launches of an update are no longer back to back
                double* own = c->h_gather + (size_t)c->comm_rank * slot;
                LV_HIP(hipMemcpyAsync(own, c->d_gather[i & 1] + (size_t)c->comm_rank * slot, slot * sizeof(double), hipMemcpyDeviceToHost, c->stream));
                LV_HIP(hipStreamSynchronize(c->stream));
                if (c->gather_cb(c->gather_user, c->h_gather, slot * sizeof(double), c->comm_rank, c->comm_world) != 0) {
                    set_error("host gather callback failed (launch %d)", i);
                    return LV_ESTATE;
                }
                LV_HIP(hipMemcpyAsync(c->d_gather[i & 1], c->h_gather, slot * sizeof(double) * (size_t)c->comm_world, hipMemcpyHostToDevice, c->stream));
                rc = LV_OK;
            } else if (c->peer.active) {
                rc = peer_gather(c->peer, i & 1, slot, c->stream);   // publish this rank's slot, pull the others' (one small kernel)
            } else {
                rc = comm_allgather_inplace(c->comm, c->d_gather[i & 1], slot, c->comm_rank, c->stream);
            }
            if (rc) return rc;
            if (c->profiling) LV_HIP(hipEventRecord(c->ev_coll[2 * i + 1], c->stream));
        }
        if (c->profiling && !closing) LV_HIP(hipEventRecord(c->ev_pass[3 * i + 2], c->stream));
    }
    c->coll_timed = gathered && c->profiling;
    return LV_OK;
}

}  // namespace

extern "C" {

void lv_default_params(lv_params* p) {
    if (!p) return;
    std::memset(p, 0, sizeof(*p));
    p->MAX_NUM_ITERS = 3;           // config/params.yaml:46
    p->NUM_MATCH_POINTS = 5;        // :48
    p->MAX_DIST_PLANE = 2.0;        // :49
    p->PLANES_THRESHOLD = 5.e-2f;   // :50
    p->estimate_extrinsics = 0;     // :13
    p->LiDAR_noise = 0.001;         // :32
    for (int i = 0; i < LV_STATE_DOF; ++i) p->LIMITS[i] = 0.001;  // src/main.cpp:145
    p->degeneracy_threshold = 5.0;  // :52 (applied by degeneracy_mode 2 only)
    p->degeneracy_mode = 0;
    p->print_degeneracy_values = 0; // :53
    p->voxel_size = 0.5f;
    p->lanes_per_query = 8;
}

const char* lv_last_error(void) { return lv::g_err; }
const char* lv_version(void) { return "limovelo_hip 0.1 (gfx950)"; }

int lv_create(const lv_params* params, int device, lv_ctx** out) {
    if (!params || !out) { set_error("null argument"); return LV_EINVAL; }
    *out = nullptr;
    if (params->NUM_MATCH_POINTS < KNN_MIN || params->NUM_MATCH_POINTS > KNN_MAX) {
        set_error("NUM_MATCH_POINTS=%d unsupported (%d..%d)", params->NUM_MATCH_POINTS, KNN_MIN, KNN_MAX);
        return LV_EINVAL;
    }
    if (params->NUM_MATCH_POINTS != KNN && params->lanes_per_query != 8) {
        set_error("NUM_MATCH_POINTS=%d runs the general build: lanes_per_query must be 8", params->NUM_MATCH_POINTS);
        return LV_EINVAL;
    }
    if (params->MAX_NUM_ITERS < 0 || params->MAX_NUM_ITERS + 1 > MAX_PASSES) { set_error("MAX_NUM_ITERS out of range"); return LV_EINVAL; }
    if (!(params->voxel_size > 0.f)) { set_error("voxel_size must be > 0"); return LV_EINVAL; }
    const int S = params->lanes_per_query;
    if (!(S == 1 || S == 2 || S == 4 || S == 8 || S == 16)) { set_error("lanes_per_query must be 1,2,4,8,16"); return LV_EINVAL; }
    if (params->degeneracy_mode < 0 || params->degeneracy_mode > 2) { set_error("degeneracy_mode must be 0, 1 or 2"); return LV_EINVAL; }
    if (params->degeneracy_mode == 0 && params->degeneracy_threshold != 5.0) {
        static bool warned = false;
        if (!warned) {
            warned = true;
            fprintf(stderr, "limovelo_hip: degeneracy_threshold = %g is set but has no effect: the degeneracy stage of the fork's "
                            "update_iterated_dyn_share_modified is not part of the reference mount; degeneracy_mode = 2 enables a "
                            "documented restatement, degeneracy_mode = 1 reports the eigenvalues (include/limovelo_hip.h)\n",
                    params->degeneracy_threshold);
        }
    }
    int ndev = 0;
    hipError_t e = hipGetDeviceCount(&ndev);
    if (e != hipSuccess || ndev <= 0) { set_error("no HIP device available (%s)", hipGetErrorString(e)); return LV_ENODEV; }
    if (device < 0 || device >= ndev) { set_error("device %d out of range (%d devices)", device, ndev); return LV_EINVAL; }
    LV_HIP(hipSetDevice(device));
    lv_ctx* c = new (std::nothrow) lv_ctx();
    if (!c) { set_error("out of host memory"); return LV_EINVAL; }
    c->prm = *params;
    c->device = device;
    c->scan.tile_points = 256u / (uint32_t)S;
    c->map.cell = params->voxel_size;
    hipDeviceProp_t prop;
    LV_HIP(hipGetDeviceProperties(&prop, device));
    int per_cu = 4;
    if (const char* e = getenv("LV_BLOCKS_PER_CU")) per_cu = atoi(e) > 0 ? atoi(e) : 4;  // tuning knob
    if (const char* e = getenv("LV_TILE_LPT")) c->tile_lpt = atoi(e) != 0;
    if (const char* e = getenv("LV_SPIN_WAIT")) c->spin_wait = atoi(e) != 0;
    if (const char* e = getenv("LV_FUSED_PASS")) c->fused_pass = atoi(e) != 0;
    if (const char* e = getenv("LV_FUSED_EXT")) c->fused_ext = atoi(e) != 0;
    if (const char* e = getenv("LV_FUSED_MULTI")) c->fused_multi_round = atoi(e) != 0 ? 1 : 0;
    if (const char* e = getenv("LV_ASYNC_RELINEARISE")) c->relin_async = atoi(e) != 0;
    if (const char* e = getenv("LV_RELIN_SLICE_WGS")) c->relin_slice_wgs = (uint32_t)atol(e);
    if (const char* e = getenv("LV_RELIN_PAUSE_US")) c->relin_pause_us = (uint32_t)atol(e);
    if (const char* e = getenv("LV_MULTI_OVERLAP")) c->multi_overlap = atoi(e) != 0;   // A/B: 0 = every round's fits between two barriers
    if (const char* e = getenv("LV_KEEPER_BY_COST")) c->keeper_by_cost = atoi(e) != 0;
    if (const char* e = getenv("LV_COMM_FUSED")) c->comm_fused = atoi(e) != 0;
    if (const char* e = getenv("LV_SMALL_WINDOW")) c->scan.small_enabled = atoi(e) != 0;
    if (const char* e = getenv("LV_LARGE_WINDOW")) c->scan.large_enabled = atoi(e) != 0;
    if (const char* e = getenv("LV_MAIL_FILTER")) c->mail_filter = atoi(e) != 0;
    if (const char* e = getenv("LV_SMALL_INSERT")) c->map.small_front = atoi(e) != 0;
    if (const char* e = getenv("LV_SURV_LIST")) c->map.surv_list = atoi(e) != 0;
    // (the stamp buffer below is strided by pass_max_wg + 1 workgroup slots: fix the grid limit first)
    c->pass_max_wg = prop.multiProcessorCount;
    if (const char* e = getenv("LV_PASS_WG")) c->pass_max_wg = atoi(e) > 0 ? atoi(e) : c->pass_max_wg;
    if (const char* e = getenv("LV_PASS_CLK")) {
        if (atoi(e) != 0) {
            const size_t words = (size_t)(MAX_PASSES + 1) * (c->pass_max_wg + 1) * pass_clock_words();   // per launch of an update
            LV_HIP(hipMalloc(&c->d_pclk, words * sizeof(long long)));
            LV_HIP(hipMemset(c->d_pclk, 0, words * sizeof(long long)));
        }
    }
    c->max_blocks = prop.multiProcessorCount * per_cu;
    if (c->max_blocks < 64) c->max_blocks = 64;
    LV_HIP(hipStreamCreateWithFlags(&c->own_stream, hipStreamNonBlocking));
    c->stream = c->own_stream;
    LV_HIP(hipStreamCreateWithFlags(&c->side_stream, hipStreamNonBlocking));
    {   // the background map rebuild's stream (lowest priority) and snapshot event: created here, creating a stream costs ~15 ms
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; (void)hipGetLastError(); }
        if (hipStreamCreateWithPriority(&c->relin_stream, hipStreamNonBlocking, lo) != hipSuccess) { c->relin_stream = nullptr; (void)hipGetLastError(); }
        if (hipEventCreateWithFlags(&c->relin_snapshot, hipEventDisableTiming) != hipSuccess) { c->relin_snapshot = nullptr; (void)hipGetLastError(); }
    }
    LV_HIP(hipEventCreateWithFlags(&c->ev_staged, hipEventDisableTiming));
    if (const char* e = getenv("LV_OVERLAP_INSERT")) c->overlap_insert = atoi(e) != 0;
    if (const char* e = getenv("LV_BATCH_PREDICT")) c->batch_predict = atoi(e) != 0;
    if (const char* e = getenv("LV_MERGED_INSERT")) c->map.merged_back = atoi(e) != 0;
    if (const char* e = getenv("LV_SWEEP_EVICT")) c->map.sweep_evict = atoi(e) != 0;
    LV_HIP(hipMalloc(&c->d_kf, sizeof(KfDev)));
    LV_HIP(hipMemset(c->d_kf, 0, sizeof(KfDev)));
    LV_HIP(hipMalloc(&c->d_filter, sizeof(FilterDev)));
    LV_HIP(hipMemset(c->d_filter, 0, sizeof(FilterDev)));
    LV_HIP(hipHostMalloc((void**)&c->h_filter, sizeof(FilterDev), hipHostMallocDefault));
    LV_HIP(hipHostMalloc((void**)&c->h_kf, sizeof(KfDev), hipHostMallocDefault));
    std::memset(c->h_kf, 0, sizeof(KfDev));
    LV_HIP(hipHostMalloc((void**)&c->h_io, sizeof(KfHostIO), hipHostMallocMapped));
    std::memset(c->h_io, 0, sizeof(KfHostIO));
    LV_HIP(hipHostGetDevicePointer((void**)&c->d_io, c->h_io, 0));
    LV_HIP(hipMalloc(&c->d_partials, (size_t)(c->max_blocks + 8) * SUMS_LEN * sizeof(double)));
    LV_HIP(hipMalloc(&c->d_groups, (size_t)(c->max_blocks / 32 + 2) * SUMS_LEN * sizeof(double)));
    for (int i = 0; i < 2; ++i) {
        LV_HIP(hipMalloc(&c->d_cpart[i], (size_t)(c->pass_max_wg + 8) * SUMS_LEN * sizeof(double)));
        LV_HIP(hipMemset(c->d_cpart[i], 0, (size_t)(c->pass_max_wg + 8) * SUMS_LEN * sizeof(double)));
        LV_HIP(hipMalloc(&c->d_wgcost[i], (size_t)(c->pass_max_wg + 8) * sizeof(uint32_t)));
        LV_HIP(hipMemset(c->d_wgcost[i], 0, (size_t)(c->pass_max_wg + 8) * sizeof(uint32_t)));
    }
    LV_HIP(hipMalloc(&c->d_sums_own, SUMS_LEN * sizeof(double)));
    LV_HIP(hipMemset(c->d_sums_own, 0, SUMS_LEN * sizeof(double)));
    c->d_sums = c->d_sums_own;
    LV_HIP(hipHostMalloc((void**)&c->h_sums, SUMS_LEN * sizeof(double), hipHostMallocDefault));
    LV_HIP(hipEventCreate(&c->ev_begin));
    LV_HIP(hipEventCreate(&c->ev_end));
    c->ev_pass.resize((size_t)(params->MAX_NUM_ITERS + 1) * 3);
    for (auto& ev : c->ev_pass) LV_HIP(hipEventCreate(&ev));
    c->ev_coll.resize((size_t)(params->MAX_NUM_ITERS + 1) * 2);
    for (auto& ev : c->ev_coll) LV_HIP(hipEventCreate(&ev));
    *out = c;
    return LV_OK;
}

namespace {
void relin_cancel(lv_ctx* c);   // (background re-linearisation: defined with the map entry points below)
// the worker's stream and the snapshot event, created when the context is (creating a stream costs ~15 ms: not inside a cycle)
int relin_streams(lv_ctx* c) {
    if (!c->relin_stream) {
        // LOWEST priority: the rebuild's kernels sort and scatter millions of points; the cycle's small launches on the context's
        // streams must get the compute units as they free up, not queue behind them
        int lo = 0, hi = 0;
        if (hipDeviceGetStreamPriorityRange(&lo, &hi) != hipSuccess) { lo = 0; (void)hipGetLastError(); }
        if (hipStreamCreateWithPriority(&c->relin_stream, hipStreamNonBlocking, lo) != hipSuccess) {
            (void)hipGetLastError();
            LV_HIP(hipStreamCreateWithFlags(&c->relin_stream, hipStreamNonBlocking));
        }
    }
    if (!c->relin_snapshot) LV_HIP(hipEventCreateWithFlags(&c->relin_snapshot, hipEventDisableTiming));
    return LV_OK;
}
}  // namespace

void lv_destroy(lv_ctx* c) {
    if (!c) return;
    hipSetDevice(c->device);
    c->cloud.release();
    if (c->comm) { hipStreamSynchronize(c->stream); comm_destroy(c->comm); c->comm = nullptr; }
    relin_cancel(c);
    hipDeviceSynchronize();
    c->map.release();
    c->relin_shadow.release();
    if (c->relin_arena) hipFree(c->relin_arena);
    if (c->relin_stream) hipStreamDestroy(c->relin_stream);
    if (c->relin_snapshot) hipEventDestroy(c->relin_snapshot);
    c->scan.release();
    free_capture(c);
    if (c->h_stage) hipHostFree(c->h_stage);
    if (c->h_kf) hipHostFree(c->h_kf);
    if (c->h_io) hipHostFree(c->h_io);
    if (c->ev_filter_up) hipEventDestroy(c->ev_filter_up);
    if (c->h_filter) hipHostFree(c->h_filter);
    if (c->h_states_ring) hipHostFree(c->h_states_ring);
    hipFree(c->d_filter);
    if (c->h_sums) hipHostFree(c->h_sums);
    hipFree(c->d_cpart[0]); hipFree(c->d_cpart[1]); hipFree(c->d_pclk); hipFree(c->d_wgcost[0]); hipFree(c->d_wgcost[1]);
    if (c->h_gather) hipHostFree(c->h_gather);
    if (c->peer.local_alloc) {   // (once lv_comm_peer_init succeeded the gather buffers ARE the peer allocation: freed by peer_close alone)
        if (c->peer.active) c->d_gather[0] = c->d_gather[1] = nullptr;
        peer_close(c->peer);
    }
    hipFree(c->d_gather[0]); hipFree(c->d_gather[1]);
    hipFree(c->d_qrec); hipFree(c->d_clk); hipFree(c->d_kf); hipFree(c->d_partials); hipFree(c->d_groups); hipFree(c->d_sums_own);
    if (c->ev_begin) hipEventDestroy(c->ev_begin);
    if (c->ev_end) hipEventDestroy(c->ev_end);
    for (auto ev : c->ev_pass) hipEventDestroy(ev);
    for (auto ev : c->ev_coll) hipEventDestroy(ev);
    if (c->side_stream) { hipStreamSynchronize(c->side_stream); hipStreamDestroy(c->side_stream); }
    if (c->ev_staged) hipEventDestroy(c->ev_staged);
    if (c->own_stream) hipStreamDestroy(c->own_stream);
    delete c;
}

int lv_set_stream(lv_ctx* c, void* hip_stream) {
    LV_CHECK_CTX(c);
    LV_FLUSH_PREDICTS(c);
    // whatever is still tied to the OLD stream is settled on it: a remembered Buffer::clear (CloudStore keeps the stream it was
    // issued on: the caller may destroy that stream after this call) and the outcome of an incremental map insert
    { int rs = c->cloud.settle(); if (rs) return rs; }
    LV_SETTLE_MAP(c);
    LV_HIP(hipStreamSynchronize(c->stream));
    if (c->side_stream) LV_HIP(hipStreamSynchronize(c->side_stream));
    c->stream = hip_stream ? (hipStream_t)hip_stream : c->own_stream;
    return LV_OK;
}
void* lv_get_stream(lv_ctx* c) { return c ? (void*)c->stream : nullptr; }
int lv_synchronize(lv_ctx* c) {
    LV_CHECK_CTX(c);
    LV_FLUSH_PREDICTS(c);
    LV_HIP(hipStreamSynchronize(c->stream));
    if (c->side_stream) LV_HIP(hipStreamSynchronize(c->side_stream));
    return LV_OK;
}

// repack caller points into the pinned staging buffer (float4) and upload them to `dst` (device)
static int stage_map_points(lv_ctx* c, const void* points, size_t stride, size_t n, float4* dst) {
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    LV_HIP(hipStreamSynchronize(c->stream));  // staging buffer reuse
    for (size_t i = 0; i < n; ++i) {
        float x, y, z;
        read_xyz(points, stride, i, x, y, z);
        c->h_stage[i] = make_float4(x, y, z, 0.f);
    }
    if (n) LV_HIP(hipMemcpyAsync(dst, c->h_stage, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    return LV_OK;
}

static int check_map_points(const void* points, size_t stride, size_t n) {
    if (n && (!points || stride < 12)) { set_error("bad point array (stride %zu)", stride); return LV_EINVAL; }
    if (n > 0xFFFFFFF0ull) { set_error("map too large"); return LV_EINVAL; }
    for (size_t i = 0; i < n; ++i) {
        float x, y, z;
        read_xyz(points, stride, i, x, y, z);
        if (!(std::isfinite(x) && std::isfinite(y) && std::isfinite(z))) { set_error("non-finite map point at %zu", i); return LV_EINVAL; }
    }
    return LV_OK;
}

// the stream an incremental insert runs on: the context's side stream, ordered behind what the context's stream holds so far
// (the staged points), whenever the overlap applies (see lv_map_add_scan); the context's stream otherwise
static hipStream_t insert_stream(lv_ctx* c) {
    if (!(c->overlap_insert && c->side_stream && c->stream == c->own_stream && c->map.built && c->map.m > 0)) return c->stream;
    if (hipEventRecord(c->ev_staged, c->stream) != hipSuccess || hipStreamWaitEvent(c->side_stream, c->ev_staged, 0) != hipSuccess) return c->stream;
    return c->side_stream;
}

// ---- background re-linearisation (see lv_ctx::relin_*) -------------------------------------------------------------------------
namespace {
void relin_free_entry(lv_ctx::RelinEntry& e) {
    if (e.d_pts && !e.from_arena) hipFree(e.d_pts);
    if (e.ready) hipEventDestroy(e.ready);
    e.d_pts = nullptr;
    e.ready = nullptr;
}

// the worker.  Phase 1 (state 4): every allocation the copy needs — the id-indexed buffers for `relin_want` points, the journal
// arena — happens HERE, not on the caller's thread (hipMalloc of ~1 GB takes ~20 ms for a 10 M-point map); then it reports
// "allocated" (state 5) and sleeps until the caller has enqueued the snapshot (state 1).  Phase 2: rebuild the copy's search
// structure, then replay what the active map went through since the snapshot, until the journal is empty at a moment the lock is
// held — from then on the two stores hold the same point set (state 2).
void relin_worker_main(lv_ctx* c) {
    auto fail = [&](const char* what) {
        std::lock_guard<std::mutex> g(c->relin_mu);
        c->relin_error = std::string(what) + ": " + lv_last_error();
        c->relin_state = 3;
    };
    if (hipSetDevice(c->device) != hipSuccess) { fail("hipSetDevice"); return; }
    MapStore& S = c->relin_shadow;
    S.n_ids = 0;
    S.m = 0;
    S.built = false;
    // the worker's big kernels leave the compute units every ~0.1 ms (launch_sliced, lv_map.hip)
    S.slice_wgs = c->relin_slice_wgs;
    set_slice_pause_us(c->relin_pause_us);   // (this thread's launches only)
    if (S.reserve(c->relin_want) != LV_OK) { fail("reserve"); return; }
    if (!c->relin_arena) {
        if (hipMalloc(&c->relin_arena, c->relin_arena_bytes) != hipSuccess) { c->relin_arena = nullptr; (void)hipGetLastError(); }
    }
    {
        std::unique_lock<std::mutex> g(c->relin_mu);
        c->relin_state = 5;
        c->relin_cv.wait(g, [&] { return c->relin_state != 5; });
        if (c->relin_state != 1) return;   // cancelled
    }
    hipStream_t st = c->relin_stream;
    if (hipStreamWaitEvent(st, c->relin_snapshot, 0) != hipSuccess) { fail("wait for the snapshot"); return; }
    if (S.rebuild(st) != LV_OK) { fail("rebuild"); return; }
    if (c->relin_test_delay_ms > 0) std::this_thread::sleep_for(std::chrono::milliseconds(c->relin_test_delay_ms));   // (test hook: lets the journal fill)
    for (;;) {
        lv_ctx::RelinEntry e;
        {
            std::lock_guard<std::mutex> g(c->relin_mu);
            if (c->relin_journal.empty()) { c->relin_state = 2; return; }
            e = c->relin_journal.front();
            c->relin_journal.pop_front();
        }
        int rc = LV_OK;
        if (e.kind <= 1) {
            if (hipStreamWaitEvent(st, e.ready, 0) != hipSuccess) rc = LV_EHIP;
            if (!rc) rc = S.reserve_batch(e.n);
            if (!rc && hipMemcpyAsync(S.d_new, e.d_pts, (size_t)e.n * sizeof(float4), hipMemcpyDeviceToDevice, st) != hipSuccess) rc = LV_EHIP;
            if (!rc) rc = S.add_staged(st, e.n, e.downsample, e.box, e.kind == 1);
            if (!rc) rc = S.settle(st);
        } else if (e.kind == 2) {
            rc = S.evict_box(st, e.lo, e.hi, e.keep_inside, nullptr);
        } else {
            rc = S.evict_oldest(st, e.n_oldest, nullptr);
        }
        if (hipStreamSynchronize(st) != hipSuccess) rc = rc ? rc : LV_EHIP;
        relin_free_entry(e);
        if (rc) { fail("replay of a journaled map operation"); return; }
        ++c->relin_replayed;
    }
}

// join the worker and throw its work away (lv_map_build / lv_map_relinearise / lv_destroy while a rebuild is in flight)
void relin_cancel(lv_ctx* c) {
    {
        std::lock_guard<std::mutex> g(c->relin_mu);
        if (c->relin_state == 4 || c->relin_state == 5) c->relin_cancelled = true;
    }
    if (c->relin_worker.joinable()) {
        for (;;) {   // a worker that is still allocating reaches its wait first
            {
                std::lock_guard<std::mutex> g(c->relin_mu);
                if (c->relin_state == 5) { c->relin_state = 0; c->relin_cv.notify_all(); }
                if (c->relin_state != 4) break;
            }
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
        c->relin_worker.join();
    }
    if (c->relin_snap_side) {   // (a snapshot may still be reading the active store on the side stream: build / relinearise rewrite it)
        if (c->side_stream) hipStreamSynchronize(c->side_stream);
        c->relin_snap_side = false;
    }
    for (auto& e : c->relin_journal) relin_free_entry(e);
    c->relin_journal.clear();
    c->relin_state = 0;
    c->relin_cancelled = false;
    c->relin_arena_used = 0;
    c->map.defer_relinearise = false;
}

// Called at the start of every map call (the map is settled): the cycle boundary the reference's loop gives us
// (src/main.cpp:102).  An allocated copy gets its snapshot here (a launch chain on the context's stream: no allocation, no wait);
// a finished rebuild is adopted here; a failed one is dropped (the stop-the-world path remains).
int relin_poll(lv_ctx* c) {
    if (c->relin_state == 0) return LV_OK;
    int st;
    { std::lock_guard<std::mutex> g(c->relin_mu); st = c->relin_state; }
    if (st == 1 || st == 4) return LV_OK;
    if (st == 5) {
        if ((size_t)c->map.m + 1 > c->relin_shadow.capacity) {   // (the map outgrew the slack while the worker was allocating: rare)
            int rr = c->relin_shadow.reserve((size_t)c->map.m + 1);
            if (rr) return rr;
        }
        // The snapshot only READS the active store: with the context's own stream it goes to the side stream (behind everything
        // the context's stream holds so far), where the inserts run as well — the cycle's next prediction / window / update do not
        // queue up behind the compaction of every id (0.3-0.4 ms at 10 M points: it was the slowest cycle of a rebuild).  What
        // MUTATES the map afterwards is ordered behind it: inserts by running on the side stream, anything on the context's
        // stream by relin_order().
        hipStream_t ss = c->stream;
        if (c->overlap_insert && c->side_stream && c->stream == c->own_stream && hipEventRecord(c->ev_staged, c->stream) == hipSuccess &&
            hipStreamWaitEvent(c->side_stream, c->ev_staged, 0) == hipSuccess)
            ss = c->side_stream;
        int rc = c->map.snapshot_into(c->relin_shadow, ss);
        if (rc) return rc;
        LV_HIP(hipEventRecord(c->relin_snapshot, ss));
        c->relin_snap_side = ss != c->stream;
        std::lock_guard<std::mutex> g(c->relin_mu);
        c->relin_state = 1;
        c->relin_cv.notify_all();
        return LV_OK;
    }
    if (c->relin_worker.joinable()) c->relin_worker.join();
    if (st == 3) {
        //  a failed worker — out of memory for the second store is the likely, and persistent, cause — must not be
        // started again by the very call that found it failed: relin_cancel lifts the deferral, and with the background form
        // switched off for this context the insert that follows takes the stop-the-world path (MapStore::needs_relinearise),
        // which rebuilds in place and needs no second store.  lv_set_option "async_relinearise" 1 turns it back on.
        fprintf(stderr, "[limovelo_hip] background map rebuild failed (%s); the map stays as it is, re-linearisations of this context stop the world from here on\n",
                c->relin_error.c_str());
        relin_cancel(c);
        c->relin_async = false;
        return LV_OK;
    }
    // ready: the worker's stream is drained (it synchronised after its last operation).  Nothing the caller enqueued against
    // the old store is disturbed — its buffers stay allocated (it becomes the next rebuild's target, written by launches that
    // are ordered behind everything enqueued on the context's stream so far) — so the swap needs no wait at all.
    std::swap(c->map, c->relin_shadow);           // (relin_shadow is the OLD active store from here on: it carries the history)
    c->map.slice_wgs = 0;
    c->relin_snap_side = false;   // (the snapshot was complete before the worker's first launch)
    c->map.defer_relinearise = false;
    c->relin_shadow.defer_relinearise = false;
    c->map.relinearisations = c->relin_shadow.relinearisations + 1;
    c->map.incremental_adds = c->relin_shadow.incremental_adds;
    c->map.dropped_total = c->relin_shadow.dropped_total;
    c->map.refresh_view();
    c->relin_state = 0;
    c->relin_arena_used = 0;
    ++c->relin_swapped;
    return LV_OK;
}

// start the worker (its first phase allocates; the snapshot follows at the next map call: relin_poll)
int relin_start(lv_ctx* c) {
    if (c->relin_state != 0) return LV_OK;
    { int rs = relin_streams(c); if (rs) return rs; }
    c->relin_want = (size_t)c->map.m + (size_t)c->map.m / 8 + 262144;
    c->map.defer_relinearise = true;
    c->relin_error.clear();
    c->relin_state = 4;
    ++c->relin_started;
    c->relin_worker = std::thread(relin_worker_main, c);
    return LV_OK;
}

// an insert of n points is about to go to the active map: start a background rebuild if the map wants one
int relin_maybe_start(lv_ctx* c, size_t incoming) {
    if (c->relin_state != 0 || !c->relin_async || !c->map.built || c->map.m < c->relin_async_min) return LV_OK;
    if ((uint64_t)c->map.n_ids + incoming > 0xFFFFFFF0ull) return LV_OK;     // id space exhausted: only the stop-the-world path helps
    if (!c->map.wants_relinearise(incoming)) return LV_OK;
    return relin_start(c);
}

// a mutation of the active map is about to be enqueued on `s`: it must not overtake a snapshot that is being taken on the side stream
int relin_order(lv_ctx* c, hipStream_t s) {
    if (!c->relin_snap_side || s == c->side_stream) return LV_OK;
    LV_HIP(hipStreamWaitEvent(s, c->relin_snapshot, 0));
    if (s == c->stream) c->relin_snap_side = false;   // (everything later on the context's stream is behind it as well)
    return LV_OK;
}

// journal a batch staged in c->map.d_new (copied on the context's stream, which staged it) for the worker to replay
int relin_journal_add(lv_ctx* c, uint32_t n, int downsample, float box, bool build_if_empty) {
    if (n == 0) return LV_OK;
    if (c->relin_test_race && c->relin_state == 1) {   // (test hook: let the worker report "ready" right here, between this call's poll and its journal entry)
        const auto t0 = std::chrono::steady_clock::now();
        while (c->relin_state == 1 && std::chrono::steady_clock::now() - t0 < std::chrono::seconds(5)) std::this_thread::sleep_for(std::chrono::microseconds(100));
    }
    int st = c->relin_state;
    if (st != 1 && st != 2) return LV_OK;    // (0 idle; 4 / 5: the snapshot is still to come and will contain this batch; 3 failed)
    lv_ctx::RelinEntry e;
    if (st == 1) {
        e.kind = build_if_empty ? 1 : 0;
        e.n = n;
        e.downsample = downsample;
        e.box = box;
        const size_t bytes = (((size_t)n * sizeof(float4)) + 255) & ~(size_t)255;
        if (c->relin_arena && c->relin_arena_used + bytes <= c->relin_arena_bytes) {   // a bump arena: no hipMalloc on the caller's thread
            e.d_pts = reinterpret_cast<float4*>(static_cast<char*>(c->relin_arena) + c->relin_arena_used);
            e.from_arena = true;
            c->relin_arena_used += bytes;
        } else {
            LV_HIP(hipMalloc(&e.d_pts, (size_t)n * sizeof(float4)));
        }
        // (an error below must not leak the entry: ADVICE r05)
        hipError_t je = hipMemcpyAsync(e.d_pts, c->map.d_new, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream);
        if (je == hipSuccess) je = hipEventCreateWithFlags(&e.ready, hipEventDisableTiming);
        if (je == hipSuccess) je = hipEventRecord(e.ready, c->stream);
        if (je != hipSuccess) {
            relin_free_entry(e);
            set_error("map rebuild journal: %s", hipGetErrorString(je));
            return LV_EHIP;
        }
        {
            std::lock_guard<std::mutex> g(c->relin_mu);
            st = c->relin_state;
            if (st == 1 && c->relin_journal.size() < c->relin_journal_max) { c->relin_journal.push_back(e); return LV_OK; }
        }
        relin_free_entry(e);
        if (st == 1) {   // the worker does not drain the journal as fast as the caller fills it: give the copy up (bounded memory, bounded deferral)
            fprintf(stderr, "[limovelo_hip] background map rebuild cannot keep up (%zu journaled operations): cancelled; re-linearisations of this context stop the world from here on\n",
                    c->relin_journal_max);
            relin_cancel(c);
            c->relin_async = false;
            return LV_OK;
        }
    }
    if (st == 2) {
        // The worker caught up and reported "ready" after this call's relin_poll: the copy no longer takes journal entries, so
        // this batch must go to the copy AS THE ACTIVE MAP — adopt it now and move the staged batch over (it sits in the old
        // store's staging buffer); the caller's add_staged then acts on the adopted store.
        float4* staged_old = c->map.d_new;
        int rc = relin_poll(c);
        if (rc) return rc;
        rc = c->map.reserve_batch(n);
        if (rc) return rc;
        LV_HIP(hipMemcpyAsync(c->map.d_new, staged_old, (size_t)n * sizeof(float4), hipMemcpyDeviceToDevice, c->stream));
    }
    return LV_OK;   // (3: the worker failed meanwhile — the active map simply stays)
}
int relin_journal_evict(lv_ctx* c, int kind, const float* lo, const float* hi, int keep_inside, uint32_t n_oldest) {
    if (c->relin_state != 1 && c->relin_state != 2) return LV_OK;
    lv_ctx::RelinEntry e;
    e.kind = kind;
    if (lo) for (int a = 0; a < 3; ++a) { e.lo[a] = lo[a]; e.hi[a] = hi[a]; }
    e.keep_inside = keep_inside;
    e.n_oldest = n_oldest;
    int st;
    {
        std::lock_guard<std::mutex> g(c->relin_mu);
        st = c->relin_state;
        if (st == 1 && c->relin_journal.size() < c->relin_journal_max) { c->relin_journal.push_back(e); return LV_OK; }
    }
    if (st == 1) { relin_cancel(c); c->relin_async = false; return LV_OK; }   // (journal full: see relin_journal_add)
    if (st == 2) return relin_poll(c);   // (ready since this call's poll: adopt the copy first, the caller's eviction then acts on it)
    return LV_OK;
}
}  // namespace

#define LV_RELIN_POLL(c)            \
    do {                            \
        int _rp = relin_poll(c);    \
        if (_rp) return _rp;        \
    } while (0)

int lv_map_build(lv_ctx* c, const void* points, size_t stride, size_t n) {
    LV_CHECK_CTX(c);
    relin_cancel(c);
    LV_SETTLE_MAP(c);
    int rc = check_map_points(points, stride, n);   // bad input leaves the previous map untouched
    if (rc) return rc;
    LV_HIP(hipStreamSynchronize(c->stream));
    c->map.n_ids = 0;
    c->map.m = 0;
    c->map.built = false;
    c->map.refresh_view();
    rc = c->map.reserve(n);
    if (rc) return rc;
    rc = stage_map_points(c, points, stride, n, c->map.d_orig);
    if (rc) return rc;
    c->map.n_ids = (uint32_t)n;
    c->map.origin_set = false;
    c->map.cell = c->prm.voxel_size;
    return c->map.rebuild(c->stream);
}

int lv_map_add(lv_ctx* c, const void* points, size_t stride, size_t n, int downsample) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    LV_RELIN_POLL(c);
    if (n == 0) return LV_OK;
    int rc = check_map_points(points, stride, n);
    if (rc) return rc;
    if ((uint64_t)c->map.n_ids + n > 0xFFFFFFF0ull) { set_error("map too large"); return LV_EINVAL; }
    rc = relin_maybe_start(c, n);
    if (rc) return rc;
    rc = c->map.reserve_batch(n);
    if (rc) return rc;
    rc = stage_map_points(c, points, stride, n, c->map.d_new);
    if (rc) return rc;
    rc = relin_journal_add(c, (uint32_t)n, downsample, 0.2f, false);
    if (rc) return rc;
    hipStream_t is = insert_stream(c);
    rc = relin_order(c, is);
    if (rc) return rc;
    return c->map.add_staged(is, (uint32_t)n, downsample, 0.2f, false);  // box_length of KD_TREE(0.3, 0.6, 0.2), Mapper.cpp:65
}

namespace {
// `Xt2 * Xt2.I_Rt_L() * p` (src/main.cpp:92; State.cpp:51-62,83-85; RotTransl.cpp:36-48) for every point of the
// current scan, in scan order, with the state taken from the device (the resident filter or the last update)
__global__ void scan_to_world_kernel(const double* __restrict__ x, const float4* __restrict__ scan, uint32_t n, float4* __restrict__ out) {
    __shared__ PoseConsts s_pc;
    if (threadIdx.x == 0) compute_pose_consts(x, &s_pc);
    __syncthreads();
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const float4 p = scan[i];
    float4 o;
    rt_apply(s_pc.Tc, p.x, p.y, p.z, o.x, o.y, o.z);
    o.w = 0.f;
    out[i] = o;
}
}  // namespace

int lv_map_add_scan(lv_ctx* c, int downsample) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    LV_RELIN_POLL(c);
    const uint32_t n = c->scan.n;
    if (n == 0) return LV_OK;   // Mapper::add returns on an empty cloud (Mapper.cpp:20)
    LV_FLUSH_PREDICTS(c);
    int rc = relin_maybe_start(c, n);
    if (rc) return rc;
    rc = c->map.reserve_batch(n);
    if (rc) return rc;
    // the state of whichever path ran last (main.cpp:92,102: Xt2 = the state the update just produced — or, before the first
    // map exists, the propagated state the caller handed to lv_update)
    { int ru = filter_to_device(c); if (ru) return ru; }
    const double* x = (c->state_src == 2 || !c->filter_set || c->filter_in_kf) ? c->d_kf->x : c->d_filter->x;
    hipLaunchKernelGGL(scan_to_world_kernel, dim3((n + 255) / 256), dim3(256), 0, c->stream, x, c->scan.d_raw, n, c->map.d_new);
    LV_HIP(hipGetLastError());
    // The insert depends on nothing but the world points just staged, and nothing depends on it until the next search: it runs
    // on the context's SIDE stream, beside whatever the caller enqueues next (in the reference's loop, src/main.cpp:52-128: the
    // next cycle's IMU propagation, LiDAR window, de-skew and voxel grid — ~100 us of small launches while the insert's chain
    // of ~150 us occupies a handful of CUs).  Everything that touches the map settles the insert first (LV_SETTLE_MAP: the
    // host waits for the note the chain's last kernel posts), so no other ordering is needed.  Only with the context's own
    // stream: a caller-provided stream keeps everything on that stream.
    rc = relin_journal_add(c, n, downsample, 0.2f, true);
    if (rc) return rc;
    hipStream_t is = insert_stream(c);
    rc = relin_order(c, is);
    if (rc) return rc;
    return c->map.add_staged(is, n, downsample, 0.2f, true);   // Mapper::add: an empty map is built from the cloud
}

int lv_map_evict_box(lv_ctx* c, const float lo[3], const float hi[3], int keep_inside, size_t* n_evicted) {
    LV_CHECK_CTX(c);
    if (!lo || !hi) { set_error("null argument"); return LV_EINVAL; }
    uint32_t ne = 0;
    LV_SETTLE_MAP(c);
    LV_RELIN_POLL(c);
    int rc = relin_journal_evict(c, 2, lo, hi, keep_inside, 0);
    if (!rc) rc = relin_order(c, c->stream);
    if (!rc) rc = c->map.evict_box(c->stream, lo, hi, keep_inside, &ne);
    if (n_evicted) *n_evicted = ne;
    return rc;
}

int lv_map_evict_oldest(lv_ctx* c, size_t n_oldest, size_t* n_evicted) {
    LV_CHECK_CTX(c);
    uint32_t ne = 0;
    LV_SETTLE_MAP(c);
    LV_RELIN_POLL(c);
    int rc = relin_journal_evict(c, 3, nullptr, nullptr, 0, (uint32_t)(n_oldest > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : n_oldest));
    if (!rc) rc = relin_order(c, c->stream);
    if (!rc) rc = c->map.evict_oldest(c->stream, (uint32_t)(n_oldest > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : n_oldest), &ne);
    if (n_evicted) *n_evicted = ne;
    return rc;
}

int lv_map_relinearise(lv_ctx* c) {
    LV_CHECK_CTX(c);
    relin_cancel(c);     // (a background rebuild in flight is superseded by this synchronous one)
    LV_SETTLE_MAP(c);
    if (!c->map.built) return LV_OK;
    return c->map.relinearise(c->stream);
}

int lv_map_relinearise_async(lv_ctx* c) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    LV_RELIN_POLL(c);
    if (!c->map.built || c->map.m == 0) return LV_OK;
    return relin_start(c);
}

// Every allocation a background rebuild needs, made NOW (set-up time) instead of by the first rebuild's worker: the second store is
// allocated for the map as it stands, filled from a snapshot and rebuilt once — synchronously, in whole grids — so that its id
// buffers, pools and tables exist at the sizes a rebuild of this map takes, have been touched, and the journal arena is there.
// The first background rebuild of the context then behaves like every later one (which find the previous active store waiting).
int lv_map_reserve_rebuild(lv_ctx* c) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    LV_RELIN_POLL(c);
    if (c->relin_state != 0) { set_error("lv_map_reserve_rebuild: a background rebuild is in flight"); return LV_ESTATE; }
    if (!c->map.built || c->map.m == 0) return LV_OK;
    { int rs = relin_streams(c); if (rs) return rs; }
    MapStore& S = c->relin_shadow;
    S.n_ids = 0;
    S.m = 0;
    S.built = false;
    S.slice_wgs = 0;
    int rc = S.reserve((size_t)c->map.m + (size_t)c->map.m / 8 + 262144);
    if (rc) return rc;
    if (!c->relin_arena && hipMalloc(&c->relin_arena, c->relin_arena_bytes) != hipSuccess) { c->relin_arena = nullptr; (void)hipGetLastError(); }
    LV_HIP(hipStreamSynchronize(c->stream));
    if (c->side_stream) LV_HIP(hipStreamSynchronize(c->side_stream));
    rc = c->map.snapshot_into(S, c->stream);
    if (!rc) rc = S.rebuild(c->stream);
    if (!rc && c->map.batch_cap) rc = S.reserve_batch(c->map.batch_cap);
    if (!rc) rc = S.ensure_boxes(c->stream, 0.2f);   // (the 0.2 m box table a down-sampling replay builds: Mapper.cpp:65)
    LV_HIP(hipStreamSynchronize(c->stream));
    S.have_boxes = false;
    // (the copy's contents are not kept: the next rebuild takes its own snapshot)
    S.n_ids = 0;
    S.m = 0;
    S.built = false;
    return rc;
}

int lv_map_rebuild_status(lv_ctx* c, int wait, uint64_t out[4]) {
    LV_CHECK_CTX(c);
    if (wait && c->relin_state != 0) {
        LV_SETTLE_MAP(c);
        for (;;) {   // allocating -> (snapshot) -> rebuilding -> ready -> adopted
            LV_RELIN_POLL(c);
            int st;
            { std::lock_guard<std::mutex> g(c->relin_mu); st = c->relin_state; }
            if (st == 0) break;
            std::this_thread::sleep_for(std::chrono::microseconds(200));
        }
    }
    if (out) {
        int st;
        { std::lock_guard<std::mutex> g(c->relin_mu); st = c->relin_state; out[3] = (uint64_t)c->relin_journal.size(); }
        out[0] = (uint64_t)st;
        out[1] = c->relin_started;
        out[2] = c->relin_swapped;
    }
    return LV_OK;
}

int lv_map_get_stats(lv_ctx* c, lv_map_stats* out) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    LV_RELIN_POLL(c);
    if (!out) { set_error("null argument"); return LV_EINVAL; }
    static_assert(sizeof(lv_map_stats) == sizeof(MapStats), "lv_map_stats layout");
    MapStats st;
    if (c->map.h_cnt && c->map.d_cnt && c->map.built) {   // (the pools' cursors live on the device: fetch them for the statistics)
        LV_HIP(hipMemcpyAsync(c->map.h_cnt, c->map.d_cnt, sizeof(MapCounters), hipMemcpyDeviceToHost, c->stream));
        LV_HIP(hipStreamSynchronize(c->stream));
    }
    c->map.stats(&st);
    std::memcpy(out, &st, sizeof(st));
    return LV_OK;
}

size_t lv_map_size(lv_ctx* c) {
    if (!c) return 0;
    c->map.settle(c->stream);
    relin_poll(c);
    return c->map.m;
}

// the living points in map order (ids ascending): the index space of lv_fetch_knn
int lv_map_fetch(lv_ctx* c, float* xyz_out, size_t capacity) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    const size_t m = c->map.m, ids = c->map.n_ids;
    if (capacity < m || (!xyz_out && capacity)) { set_error("capacity too small"); return LV_EINVAL; }
    if (m == 0) return LV_OK;
    std::vector<float4> tmp(ids);
    LV_HIP(hipStreamSynchronize(c->stream));
    LV_HIP(hipMemcpy(tmp.data(), c->map.d_orig, ids * sizeof(float4), hipMemcpyDeviceToHost));
    size_t o = 0;
    for (size_t i = 0; i < ids && o < m; ++i) {
        if (!std::isfinite(tmp[i].x)) continue;
        xyz_out[3 * o] = tmp[i].x; xyz_out[3 * o + 1] = tmp[i].y; xyz_out[3 * o + 2] = tmp[i].z;
        ++o;
    }
    if (o != m) { set_error("map bookkeeping: %zu living points found, %zu expected", o, m); return LV_ESTATE; }
    return LV_OK;
}

int lv_scan_set(lv_ctx* c, const void* points, size_t stride, size_t n) {
    LV_CHECK_CTX(c);
    if (n && (!points || stride < 12)) { set_error("bad point array (stride %zu)", stride); return LV_EINVAL; }
    if (n > 0xFFFFFFF0ull) { set_error("scan too large"); return LV_EINVAL; }
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    rc = c->scan.reserve(n);
    if (rc) return rc;
    LV_HIP(hipStreamSynchronize(c->stream));  // staging buffer reuse
    float bmin[3] = {INFINITY, INFINITY, INFINITY};
    for (size_t i = 0; i < n; ++i) {
        float x, y, z;
        read_xyz(points, stride, i, x, y, z);
        uint32_t ui = (uint32_t)i;
        float w;
        std::memcpy(&w, &ui, 4);
        c->h_stage[i] = make_float4(x, y, z, w);
        if (x < bmin[0]) bmin[0] = x;
        if (y < bmin[1]) bmin[1] = y;
        if (z < bmin[2]) bmin[2] = z;
    }
    c->scan.n = (uint32_t)n;
    c->comm_shard_max = 0;   // (a new scan: the caller tells its largest shard again)
    c->dbg_valid = false;
    c->qrec_valid = false;
    if (n == 0) return LV_OK;
    LV_HIP(hipMemcpyAsync(c->scan.d_raw, c->h_stage, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    return c->scan.sort(c->stream, bmin, c->prm.voxel_size);
}

int lv_scan_deskew(lv_ctx* c, const void* points, size_t stride, size_t time_offset, size_t n, const lv_motion_state* states,
                   size_t n_states, const lv_motion_state* Xt2, float downsample_prec) {
    LV_CHECK_CTX(c);
    static_assert(sizeof(lv_motion_state) == sizeof(MotionState), "lv_motion_state layout");
    if (n && (!points || stride < 12 || time_offset + 8 > stride)) { set_error("bad point array (stride %zu, time offset %zu)", stride, time_offset); return LV_EINVAL; }
    if (!states || n_states < 2 || !Xt2) { set_error("need >= 2 surrounding states and Xt2"); return LV_EINVAL; }
    if (n > 0xFFFFFFF0ull) { set_error("scan too large"); return LV_EINVAL; }
    c->dbg_valid = false;
    c->qrec_valid = false;
    c->scan.n = 0;
    c->comm_shard_max = 0;
    if (n == 0) return LV_OK;
    int rc = ensure_stage(c, n + (n + 1) / 2);  // float4 xyz + packed doubles behind them
    if (rc) return rc;
    rc = c->scan.reserve_raw(n, n_states);
    if (rc) return rc;
    LV_HIP(hipStreamSynchronize(c->stream));  // staging buffer reuse
    double* h_times = reinterpret_cast<double*>(c->h_stage + n);
    for (size_t i = 0; i < n; ++i) {
        float x, y, z;
        read_xyz(points, stride, i, x, y, z);
        c->h_stage[i] = make_float4(x, y, z, 0.f);
        std::memcpy(&h_times[i], reinterpret_cast<const char*>(points) + i * stride + time_offset, sizeof(double));
    }
    LV_HIP(hipMemcpyAsync(c->scan.d_in, c->h_stage, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    LV_HIP(hipMemcpyAsync(c->scan.d_times, h_times, n * sizeof(double), hipMemcpyHostToDevice, c->stream));
    LV_HIP(hipMemcpyAsync(c->scan.d_states, states, n_states * sizeof(MotionState), hipMemcpyHostToDevice, c->stream));
    MotionState xt2;
    std::memcpy(&xt2, Xt2, sizeof(xt2));
    return c->scan.deskew_downsample(c->stream, (uint32_t)n, (uint32_t)n_states, xt2, downsample_prec, c->prm.voxel_size);
}

// Compensator::downsample(points) on its own (Compensator.cpp:104-107, 148-163): the voxel grid of lv_scan_deskew
// applied to points that are already compensated; the result becomes the current scan
int lv_scan_downsample(lv_ctx* c, const void* points, size_t stride, size_t n, float downsample_prec) {
    LV_CHECK_CTX(c);
    if (n && (!points || stride < 12)) { set_error("bad point array (stride %zu)", stride); return LV_EINVAL; }
    if (n > 0xFFFFFFF0ull) { set_error("scan too large"); return LV_EINVAL; }
    c->dbg_valid = false;
    c->qrec_valid = false;
    c->scan.n = 0;
    c->comm_shard_max = 0;
    if (n == 0) return LV_OK;
    int rc = ensure_stage(c, n);
    if (rc) return rc;
    rc = c->scan.reserve_raw(n, 2);
    if (rc) return rc;
    rc = c->scan.reserve(n);
    if (rc) return rc;
    LV_HIP(hipStreamSynchronize(c->stream));  // staging buffer reuse
    for (size_t i = 0; i < n; ++i) {
        float x, y, z;
        read_xyz(points, stride, i, x, y, z);
        uint32_t ui = (uint32_t)i;
        float w;
        std::memcpy(&w, &ui, 4);
        c->h_stage[i] = make_float4(x, y, z, w);
    }
    float4* dst = downsample_prec > 0.f ? c->scan.d_desk : c->scan.d_raw;
    LV_HIP(hipMemcpyAsync(dst, c->h_stage, n * sizeof(float4), hipMemcpyHostToDevice, c->stream));
    return c->scan.voxel_and_sort(c->stream, (uint32_t)n, downsample_prec, c->prm.voxel_size, true);
}

// ---- row f-4: LiDAR wire formats ----------------------------------------------------------------------------
int lv_cloud_format_preset(int lidar_type, lv_cloud_format* out) {
    if (!out) { set_error("null argument"); return LV_EINVAL; }
    std::memset(out, 0, sizeof(*out));
    out->off_x = 0; out->off_y = 4; out->off_z = 8;   // PCL_ADD_POINT4D
    switch (lidar_type) {
        case LV_LIDAR_VELODYNE:   // velodyne_ros::Point, Common.hpp:109-117
            out->point_step = 32; out->off_intensity = 16; out->intensity_type = LV_ATTR_F32;
            out->off_time = 20; out->time_type = LV_TIME_F32_SEC; out->relative_time = 1;
            break;
        case LV_LIDAR_HESAI:      // hesai_ros::Point, :119-127
            out->point_step = 48; out->off_intensity = 16; out->intensity_type = LV_ATTR_U8;
            out->off_time = 24; out->time_type = LV_TIME_F64_SEC; out->relative_time = 0;
            break;
        case LV_LIDAR_OUSTER:     // ouster_ros::Point, :155-165 (intensity <- reflectivity, range <- range: Point.cpp:173-176)
            out->point_step = 32; out->off_intensity = 24; out->intensity_type = LV_ATTR_U16;
            out->off_time = 20; out->time_type = LV_TIME_U32_NSEC; out->relative_time = 1;
            out->off_range = 28; out->range_type = LV_ATTR_U32;
            break;
        case LV_LIDAR_CUSTOM:     // custom::Point == full_info::Point, :129-153
            out->point_step = 48; out->off_intensity = 20; out->intensity_type = LV_ATTR_F32;
            out->off_time = 32; out->time_type = LV_TIME_F64_SEC; out->relative_time = 0;
            break;
        default: set_error("unknown LiDAR type %d", lidar_type); return LV_EINVAL;
    }
    return LV_OK;
}

namespace {
double microsec_to_sec(uint64_t t) {   // Conversions::microsec2Sec (src/Utils/Utils.cpp:18-23): int arithmetic as there
    const int order = 1000000;
    const int secs = (int)(t / (uint64_t)order);
    const int musecs = (int)(t % (uint64_t)order);
    return secs + musecs * 1e-6;
}
double nanosec_to_sec_host(uint32_t t) {   // Conversions::nanosec2Sec (:25-30)
    const int order = 1000000000;
    const int secs = (int)(t / (uint32_t)order);
    const int nsecs = (int)(t % (uint32_t)order);
    return secs + nsecs * 1e-9;
}
double raw_time(const unsigned char* rec, const lv_cloud_format& f) {
    if (f.time_type == LV_TIME_F32_SEC) { float v; std::memcpy(&v, rec + f.off_time, 4); return (double)v; }
    if (f.time_type == LV_TIME_U32_NSEC) { uint32_t v; std::memcpy(&v, rec + f.off_time, 4); return nanosec_to_sec_host(v); }
    double v; std::memcpy(&v, rec + f.off_time, 8); return v;
}
}  // namespace

int lv_cloud_ingest(lv_ctx* c, const void* data, size_t n, const lv_cloud_format* f, const lv_ingest_params* prm, size_t* n_kept) {
    LV_CHECK_CTX(c);
    static_assert(sizeof(lv_cloud_format) == sizeof(CloudFormat) && sizeof(lv_ingest_params) == sizeof(IngestParams), "f-4 struct layouts");
    if (n_kept) *n_kept = 0;
    if (!f || !prm || (n && !data)) { set_error("null argument"); return LV_EINVAL; }
    const uint32_t tsz = f->time_type == LV_TIME_F64_SEC ? 8u : 4u;
    if (f->point_step < 12 || f->off_x + 4 > f->point_step || f->off_y + 4 > f->point_step || f->off_z + 4 > f->point_step ||
        f->off_time + tsz > f->point_step || f->time_type < 0 || f->time_type > 2 ||
        (f->intensity_type != LV_ATTR_NONE && f->off_intensity + 4 > f->point_step + 3) || f->off_range + 4 > f->point_step + 4) {
        set_error("bad cloud format (point_step %u)", f->point_step);
        return LV_EINVAL;
    }
    if (n > 0x7FFFFFF0ull) { set_error("message too large"); return LV_EINVAL; }
    if (n == 0) return LV_OK;
    // PointCloudProcessor::get_begin_time (PointCloudProcessor.cpp:43-47,57-60,70-74,84-91)
    double begin = 0.0;
    if (f->relative_time) {
        const unsigned char* raw = static_cast<const unsigned char*>(data);
        const double front = raw_time(raw, *f), back = raw_time(raw + (n - 1) * (size_t)f->point_step, *f);
        begin = microsec_to_sec(prm->header_stamp_usec) + front;
        if (!prm->stamp_beginning) begin = begin - back;
    }
    CloudFormat cf;
    IngestParams ip;
    std::memcpy(&cf, f, sizeof(cf));
    std::memcpy(&ip, prm, sizeof(ip));
    return c->cloud.ingest(c->stream, data, n, cf, ip, begin, n_kept);
}

int lv_cloud_reserve(lv_ctx* c, size_t max_points_per_message, size_t point_step, size_t buffer_points) {
    LV_CHECK_CTX(c);
    if (point_step < 12 || max_points_per_message > 0x7FFFFFF0ull || buffer_points > 0x7FFFFFF0ull) { set_error("bad sizes"); return LV_EINVAL; }
    int rc = c->cloud.init();
    if (rc) return rc;
    if (max_points_per_message) rc = c->cloud.reserve_msg(max_points_per_message, max_points_per_message * point_step);
    if (rc) return rc;
    rc = c->cloud.settle();
    if (rc) return rc;
    return buffer_points ? c->cloud.reserve_buffer(c->stream, (size_t)c->cloud.size + buffer_points) : LV_OK;
}

int lv_reserve_stream(lv_ctx* c, size_t max_window_points, size_t max_scan_points) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    if (max_window_points > 0x7FFFFFF0ull || max_scan_points > 0x7FFFFFF0ull) { set_error("bad sizes"); return LV_EINVAL; }
    LV_HIP(hipStreamSynchronize(c->stream));
    int rc = LV_OK;
    if (max_window_points) {
        rc = c->scan.reserve_raw(max_window_points, 64);
        if (!rc) rc = c->scan.reserve(max_window_points);
        if (!rc && c->scan.tile_points) rc = c->scan.reserve_tiles((uint32_t)((max_window_points + c->scan.tile_points - 1) / c->scan.tile_points));
        if (rc) return rc;
        LV_HIP(note_alloc(c->scan.notes));
        if (!c->h_states_ring) LV_HIP(hipHostMalloc((void**)&c->h_states_ring, 8 * 64 * sizeof(MotionState), hipHostMallocDefault));
    }
    if (max_scan_points) {
        rc = c->scan.reserve(max_scan_points);
        if (!rc) rc = c->map.reserve_batch(max_scan_points);
        if (rc) return rc;
        LV_HIP(note_alloc(c->map.notes));
        if ((uint32_t)max_scan_points > c->qstride) {
            uint32_t cap = c->qstride ? c->qstride : 4096;
            while (cap < (uint32_t)max_scan_points) cap *= 2;
            hipFree(c->d_qrec);
            c->d_qrec = nullptr;
            c->qstride = 0;
            LV_HIP(hipMalloc(&c->d_qrec, (size_t)cap * qrec_slots(c->prm.NUM_MATCH_POINTS) * sizeof(float4)));
            c->qstride = cap;
        }
    }
    return LV_OK;
}

size_t lv_cloud_size(lv_ctx* c) {
    if (!c) return 0;
    c->cloud.settle();
    return (size_t)(c->cloud.size - c->cloud.head);
}

int lv_cloud_fetch(lv_ctx* c, double t1, double t2, void* out, size_t capacity, size_t* n) {
    LV_CHECK_CTX(c);
    if (n) *n = 0;
    uint32_t lo = 0, hi = 0;
    int rc = c->cloud.window(c->stream, t1, t2, &lo, &hi);
    if (rc) return rc;
    const size_t cnt = hi - lo;
    if (n) *n = cnt;
    if (cnt == 0) return LV_OK;
    if (!out || capacity < cnt) { set_error("capacity %zu too small for %zu points", capacity, cnt); return LV_EINVAL; }
    LV_HIP(hipMemcpy(out, c->cloud.d_buf + lo, cnt * sizeof(CloudPoint), hipMemcpyDeviceToHost));
    return LV_OK;
}

int lv_cloud_clear(lv_ctx* c, double t) {
    LV_CHECK_CTX(c);
    return c->cloud.clear_before(c->stream, t);
}

int lv_scan_deskew_window(lv_ctx* c, double t1, double t2, const lv_motion_state* states, size_t n_states, const lv_motion_state* Xt2,
                          float downsample_prec, size_t* n_window) {
    LV_CHECK_CTX(c);
    if (n_window) *n_window = 0;
    if (!states || n_states < 2 || !Xt2) { set_error("need >= 2 surrounding states and Xt2"); return LV_EINVAL; }
    c->dbg_valid = false;
    c->qrec_valid = false;
    c->scan.n = 0;
    c->comm_shard_max = 0;
    uint32_t lo = 0, hi = 0;
    int rc = c->scan.reserve_raw(1, n_states);   // (the bounds words exist: the window kernel resets them on the way)
    if (rc) return rc;
    rc = c->cloud.window(c->stream, t1, t2, &lo, &hi, c->scan.d_bounds);
    if (rc) return rc;
    const uint32_t n = hi - lo;
    if (n_window) *n_window = n;
    if (n == 0) return LV_OK;   // Compensator::compensate returns no points (Compensator.cpp:24)
    rc = c->scan.reserve_raw(n, n_states);
    if (rc) return rc;
    if (n_states <= 64) {
        if (!c->h_states_ring) LV_HIP(hipHostMalloc((void**)&c->h_states_ring, 8 * 64 * sizeof(MotionState), hipHostMallocDefault));
        MotionState* slot = c->h_states_ring + (size_t)c->states_slot * 64;
        c->states_slot = (c->states_slot + 1) & 7;
        std::memcpy(slot, states, n_states * sizeof(MotionState));
        LV_HIP(hipMemcpyAsync(c->scan.d_states, slot, n_states * sizeof(MotionState), hipMemcpyHostToDevice, c->stream));
    } else {
        LV_HIP(hipMemcpyAsync(c->scan.d_states, states, n_states * sizeof(MotionState), hipMemcpyHostToDevice, c->stream));
    }
    MotionState xt2;
    std::memcpy(&xt2, Xt2, sizeof(xt2));
    if (!c->scan.small_window_applies(n) && c->scan.large_window_applies(n, (uint32_t)n_states, downsample_prec)) {
        bool fell_back = false;   // (more points out than the one-workgroup tail takes: the general chain, from the raw points)
        rc = c->scan.window_large(c->stream, c->cloud.d_buf + lo, n, (uint32_t)n_states, xt2, downsample_prec, c->prm.voxel_size, &fell_back);
        if (rc || !fell_back) return rc;
    }
    rc = c->cloud.unpack(c->stream, lo, n, c->scan.d_in, c->scan.d_times);
    if (rc) return rc;
    return c->scan.deskew_downsample(c->stream, n, (uint32_t)n_states, xt2, downsample_prec, c->prm.voxel_size);
}

size_t lv_scan_size(lv_ctx* c) { return c ? c->scan.n : 0; }

int lv_scan_fetch(lv_ctx* c, float* xyz_out, size_t capacity) {
    LV_CHECK_CTX(c);
    const size_t n = c->scan.n;
    if (capacity < n || (!xyz_out && capacity)) { set_error("capacity too small"); return LV_EINVAL; }
    if (n == 0) return LV_OK;
    std::vector<float4> tmp(n);
    LV_HIP(hipStreamSynchronize(c->stream));
    LV_HIP(hipMemcpy(tmp.data(), c->scan.d_raw, n * sizeof(float4), hipMemcpyDeviceToHost));
    for (size_t i = 0; i < n; ++i) { xyz_out[3 * i] = tmp[i].x; xyz_out[3 * i + 1] = tmp[i].y; xyz_out[3 * i + 2] = tmp[i].z; }
    return LV_OK;
}

int lv_set_capture(lv_ctx* c, int enabled) {
    LV_CHECK_CTX(c);
    c->capture = enabled != 0;
    return LV_OK;
}

int lv_set_record_dump(lv_ctx* c, int enabled) {
    LV_CHECK_CTX(c);
    c->record_dump = enabled != 0;
    return LV_OK;
}

int lv_last_update_fused(lv_ctx* c) { return (c && c->last_update_fused) ? 1 : 0; }
int lv_last_passes(lv_ctx* c) { return c ? c->h_io->passes : 0; }

int lv_pass_geometry(size_t n_scan, int n_cus, int out[4]) {
    if (!out || n_cus < 1 || n_scan > 0xFFFFFFF0ull) { set_error("lv_pass_geometry: bad arguments"); return LV_EINVAL; }
    pass_grid_size((uint32_t)n_scan, n_cus, &out[0], &out[1], &out[2], &out[3]);
    return LV_OK;
}

int lv_set_fused_pass(lv_ctx* c, int enabled) {
    LV_CHECK_CTX(c);
    c->fused_pass = enabled != 0;
    return LV_OK;
}

int lv_set_option(lv_ctx* c, const char* name, int value) {
    LV_CHECK_CTX(c);
    if (!name) { set_error("null argument"); return LV_EINVAL; }
    if (c->in_update) { set_error("lv_set_option inside an update"); return LV_ESTATE; }
    const bool on = value != 0;
    if (!std::strcmp(name, "fused_pass")) c->fused_pass = on;
    else if (!std::strcmp(name, "fused_ext")) c->fused_ext = on;
    else if (!std::strcmp(name, "fast_fit")) c->fast_fit = on;
    else if (!std::strcmp(name, "async_relinearise")) c->relin_async = on;
    else if (!std::strcmp(name, "async_relinearise_min")) c->relin_async_min = value > 0 ? (size_t)value : 0;
    else if (!std::strcmp(name, "async_relinearise_pause_us")) c->relin_pause_us = value > 0 ? (uint32_t)value : 0u;
    else if (!std::strcmp(name, "async_relinearise_slice_wgs")) c->relin_slice_wgs = value > 0 ? (uint32_t)value : 0u;      // 0: whole grids
    else if (!std::strcmp(name, "async_relinearise_journal_max")) c->relin_journal_max = value > 0 ? (size_t)value : 1;
    else if (!std::strcmp(name, "async_relinearise_test_delay_ms")) c->relin_test_delay_ms = value;
    else if (!std::strcmp(name, "async_relinearise_test_race")) c->relin_test_race = on;
    else if (!std::strcmp(name, "multi_overlap")) c->multi_overlap = on;
    else if (!std::strcmp(name, "fused_multi_round")) c->fused_multi_round = on ? 1 : 0;
    else if (!std::strcmp(name, "keeper_by_cost")) c->keeper_by_cost = on;
    else if (!std::strcmp(name, "tile_lpt")) c->tile_lpt = on;
    else if (!std::strcmp(name, "spin_wait")) c->spin_wait = on;
    else if (!std::strcmp(name, "comm_fused")) c->comm_fused = on;
    else if (!std::strcmp(name, "mail_filter")) c->mail_filter = on;
    else if (!std::strcmp(name, "overlap_insert")) c->overlap_insert = on;
    else if (!std::strcmp(name, "batch_predict")) { int rp = flush_predicts(c); if (rp) return rp; c->batch_predict = on; }
    else if (!std::strcmp(name, "merged_insert")) c->map.merged_back = on;
    else if (!std::strcmp(name, "sweep_evict")) c->map.sweep_evict = on;
    else if (!std::strcmp(name, "small_window")) c->scan.small_enabled = on;
    else if (!std::strcmp(name, "large_window")) c->scan.large_enabled = on;
    else if (!std::strcmp(name, "small_insert")) c->map.small_front = on;
    else if (!std::strcmp(name, "survivor_list")) c->map.surv_list = on;
    else { set_error("lv_set_option: unknown option '%s'", name); return LV_EINVAL; }
    return LV_OK;
}

int lv_get_pass_clocks(lv_ctx* c, long long* out, int capacity_wg, int* n_wg) {
    LV_CHECK_CTX(c);
    if (!c->d_pclk) { set_error("no pass clocks: create the context with LV_PASS_CLK=1 in the environment"); return LV_ESTATE; }
    // layout: [launch 0 .. MAX_NUM_ITERS + 1][pass_max_wg + 1 workgroup slots][pass_clock_words()]; slot n_wg - 1 of a launch is
    // its bookkeeping workgroup (stamp 10 = books done); the closing launch has one workgroup (slot 0)
    const int slots = c->pass_max_wg + 1;
    if (!out) {   // size query: *n_wg = workgroup slots per launch (the stride of the layout)
        if (n_wg) *n_wg = slots;
        return LV_OK;
    }
    if (capacity_wg < slots) { set_error("lv_get_pass_clocks: capacity %d < %d workgroup slots per launch", capacity_wg, slots); return LV_EINVAL; }
    LV_HIP(hipStreamSynchronize(c->stream));
    LV_HIP(hipMemcpy(out, c->d_pclk, (size_t)(c->prm.MAX_NUM_ITERS + 2) * slots * pass_clock_words() * sizeof(long long), hipMemcpyDeviceToHost));
    if (n_wg) *n_wg = c->pclk_wg;
    return LV_OK;
}

int lv_iterate(lv_ctx* c, const lv_state* x, lv_sums* out) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    if (!x || !out) { set_error("null argument"); return LV_EINVAL; }
    std::memset(out, 0, sizeof(*out));
    if (c->map.view.m == 0 || c->scan.n == 0) { c->dbg_valid = false; return LV_OK; }  // Mapper.cpp:42 — empty Matches
    const bool cap = c->capture;
    c->capture = true;  // lv_iterate is the API-parity path: always captures per-point outputs
    int rc = begin_common(c, x, nullptr);
    if (!rc) rc = pass_reduce(c, true);
    c->capture = cap;
    if (rc) return rc;
    LV_HIP(hipMemcpyAsync(c->h_sums, c->d_sums, SUMS_LEN * sizeof(double), hipMemcpyDeviceToHost, c->stream));
    LV_HIP(hipMemcpyAsync(&c->h_kf->fallback_queries, &c->d_kf->fallback_queries, sizeof(int), hipMemcpyDeviceToHost, c->stream));
    LV_HIP(hipStreamSynchronize(c->stream));
    unpack_sums(c->h_sums, out);
    c->timing.fallback_queries = fallback_count(c->h_kf->fallback_queries) - c->fallback_base;
    c->fallback_base = fallback_count(c->h_kf->fallback_queries);
    return LV_OK;
}

int lv_update_begin(lv_ctx* c, const lv_state* x, const double* P) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    if (!x || !P) { set_error("null argument"); return LV_EINVAL; }
    c->in_update = true;
    c->passes_issued = 0;
    return begin_common(c, x, P);
}

int lv_pass_reduce(lv_ctx* c) {
    LV_CHECK_CTX(c);
    if (!c->in_update) { set_error("lv_pass_reduce outside lv_update_begin/end"); return LV_ESTATE; }
    return pass_reduce(c, true);
}

// ---- multi-GPU (row e): RCCL issued from the library on the context stream ------------------------------
int lv_comm_unique_id(const char* rccl_library, void* id128) {
    if (!id128) { set_error("null argument"); return LV_EINVAL; }
    return comm_unique_id(rccl_library, id128);
}

int lv_comm_init(lv_ctx* c, const char* rccl_library, const void* id128, int rank, int world) {
    LV_CHECK_CTX(c);
    if (!id128 || world < 1 || rank < 0 || rank >= world) { set_error("lv_comm_init: bad arguments (rank %d, world %d)", rank, world); return LV_EINVAL; }
    if (c->in_update) { set_error("lv_comm_init inside an update"); return LV_ESTATE; }
    if (c->comm) { set_error("communicator already initialised"); return LV_ESTATE; }
    if (c->gather_cb) { set_error("a host gather transport is in place (lv_comm_set_host_gather)"); return LV_ESTATE; }
    if (c->peer.local_alloc) { set_error("a peer-mapped gather is in place (lv_comm_peer_export)"); return LV_ESTATE; }
    void* comm = nullptr;
    int rc = comm_init(rccl_library, id128, rank, world, &comm);   // collective: every rank calls it
    if (rc) return rc;
    c->comm = comm;
    c->comm_rank = rank;
    c->comm_world = world;
    return LV_OK;
}

int lv_comm_destroy(lv_ctx* c) {
    LV_CHECK_CTX(c);
    if (c->in_update) { set_error("lv_comm_destroy inside an update"); return LV_ESTATE; }
    if (c->peer.local_alloc) {   // the peer-mapped exchange: unmap the other ranks' buffers, free this rank's (the gather buffers WERE that allocation)
        LV_HIP(hipStreamSynchronize(c->stream));
        if (c->peer.active) { c->d_gather[0] = c->d_gather[1] = nullptr; c->gather_cap = 0; }
        peer_close(c->peer);
        c->comm_world = 1;
        c->comm_rank = 0;
        c->comm_shard_max = 0;
    }
    if (!c->comm) return LV_OK;
    LV_HIP(hipStreamSynchronize(c->stream));
    int rc = comm_destroy(c->comm);
    c->comm = nullptr;
    c->comm_world = 1;
    c->comm_rank = 0;
    return rc;
}

int lv_comm_set_host_gather(lv_ctx* c, int rank, int world, lv_gather_fn fn, void* user) {
    LV_CHECK_CTX(c);
    if (c->in_update) { set_error("lv_comm_set_host_gather inside an update"); return LV_ESTATE; }
    if (c->comm) { set_error("a library communicator is in place"); return LV_ESTATE; }
    if (c->peer.local_alloc) { set_error("a peer-mapped gather is in place (lv_comm_peer_export)"); return LV_ESTATE; }
    if (fn && (world < 1 || rank < 0 || rank >= world)) { set_error("lv_comm_set_host_gather: bad arguments (rank %d, world %d)", rank, world); return LV_EINVAL; }
    LV_HIP(hipStreamSynchronize(c->stream));
    c->gather_cb = fn;
    c->gather_user = fn ? user : nullptr;
    c->comm_rank = fn ? rank : 0;
    c->comm_world = fn ? world : 1;
    c->comm_shard_max = 0;
    return LV_OK;
}

int lv_comm_peer_export(lv_ctx* c, void* handle_blob) {
    LV_CHECK_CTX(c);
    if (!handle_blob) { set_error("null argument"); return LV_EINVAL; }
    if (c->in_update) { set_error("lv_comm_peer_export inside an update"); return LV_ESTATE; }
    if (c->comm || c->gather_cb) { set_error("another multi-rank transport is in place"); return LV_ESTATE; }
    LV_HIP(hipStreamSynchronize(c->stream));
    // sized once for the largest case (every CU a workgroup, 96-double partials, LV_PEER_MAX ranks): the other ranks map this
    // allocation, so it never moves
    const size_t cap = (size_t)(c->pass_max_wg + 8) * 96u * (size_t)LV_PEER_MAX;
    return peer_export(c->peer, cap, handle_blob);
}

int lv_comm_peer_init(lv_ctx* c, int rank, int world, const void* handles) {
    LV_CHECK_CTX(c);
    if (!handles) { set_error("null argument"); return LV_EINVAL; }
    if (c->in_update) { set_error("lv_comm_peer_init inside an update"); return LV_ESTATE; }
    int rc = peer_init(c->peer, rank, world, handles);
    if (rc) return rc;
    LV_HIP(hipStreamSynchronize(c->stream));
    hipFree(c->d_gather[0]); hipFree(c->d_gather[1]);
    c->d_gather[0] = c->peer.buf[0];
    c->d_gather[1] = c->peer.buf[1];
    c->gather_cap = c->peer.cap;
    c->comm_rank = rank;
    c->comm_world = world;
    c->comm_shard_max = 0;
    return LV_OK;
}

int lv_comm_world(lv_ctx* c) { return c ? c->comm_world : 0; }

int lv_comm_set_shard_max(lv_ctx* c, size_t n_max) {
    LV_CHECK_CTX(c);
    if (c->in_update) { set_error("lv_comm_set_shard_max inside an update"); return LV_ESTATE; }
    if (n_max > 0xFFFFFFF0ull) { set_error("shard too large"); return LV_EINVAL; }
    c->comm_shard_max = n_max;
    if (!multi_rank(c) || n_max == 0) return LV_OK;
    int nwg = 0, rounds = 0, steps = 0, dedicated = 0;
    pass_grid_size((uint32_t)n_max, c->pass_max_wg, &nwg, &steps, &rounds, &dedicated);
    const size_t need = (size_t)nwg * partial_width(c) * (size_t)c->comm_world;
    if (c->gather_cb && need > c->h_gather_cap) {
        LV_HIP(hipStreamSynchronize(c->stream));
        if (c->h_gather) hipHostFree(c->h_gather);
        c->h_gather = nullptr;
        c->h_gather_cap = 0;
        LV_HIP(hipHostMalloc((void**)&c->h_gather, need * sizeof(double), hipHostMallocDefault));
        std::memset(c->h_gather, 0, need * sizeof(double));
        c->h_gather_cap = need;
    }
    if (need > c->gather_cap && c->peer.active) {
        set_error("peer-mapped gather: %zu doubles exceed the exported buffers (%zu)", need, c->gather_cap);
        return LV_EINVAL;
    }
    if (need > c->gather_cap) {
        LV_HIP(hipStreamSynchronize(c->stream));
        for (int i = 0; i < 2; ++i) {
            hipFree(c->d_gather[i]);
            c->d_gather[i] = nullptr;
        }
        c->gather_cap = 0;
        for (int i = 0; i < 2; ++i) {
            LV_HIP(hipMalloc(&c->d_gather[i], need * sizeof(double)));
            LV_HIP(hipMemset(c->d_gather[i], 0, need * sizeof(double)));
        }
        c->gather_cap = need;
    }
    return LV_OK;
}

int lv_set_comm_fused(lv_ctx* c, int enabled) {
    LV_CHECK_CTX(c);
    c->comm_fused = enabled != 0;
    return LV_OK;
}

void* lv_sums_device_ptr(lv_ctx* c) { return c ? (void*)c->d_sums : nullptr; }

int lv_set_sums_buffer(lv_ctx* c, void* device_ptr) {
    LV_CHECK_CTX(c);
    if (c->in_update) { set_error("lv_set_sums_buffer inside an update"); return LV_ESTATE; }
    LV_HIP(hipStreamSynchronize(c->stream));
    c->d_sums = device_ptr ? static_cast<double*>(device_ptr) : c->d_sums_own;
    return LV_OK;
}

int lv_pass_solve(lv_ctx* c) {
    LV_CHECK_CTX(c);
    if (!c->in_update) { set_error("lv_pass_solve outside lv_update_begin/end"); return LV_ESTATE; }
    c->passes_issued++;
    return pass_solve(c, false);
}

// The pass that finishes an update stores the sequence number after all results (system-scope stores): poll it for a bounded
// time (an update takes ~0.2 ms) instead of paying the stream-synchronise wake-up; anything unusual (errors, very long updates)
// still ends in hipStreamSynchronize (by the caller, when this returns false).
static bool mailbox_wait(lv_ctx* c) {
    bool seen = false;
    if (c->spin_wait && !c->profiling && !c->phase_clocks && !c->capture) {
        volatile const unsigned long long* sc = &c->h_io->seqcheck;
        const auto t0 = std::chrono::steady_clock::now();
        unsigned long long word = 0;
        for (int it = 0;; ++it) {
            word = *sc;
            if ((uint32_t)word == (uint32_t)c->update_seq) { seen = true; break; }
            if ((it & 1023) == 1023 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(20)) break;
        }
        std::atomic_thread_fence(std::memory_order_acquire);
        if (seen) {
            // the results were stored before the word, but only the checksum proves that they have all ARRIVED
            const uint32_t want = (uint32_t)(word >> 32);
            const KfHostIO* io = c->h_io;
            uint32_t chk = 0;
            for (int i = 0; i < NS * NS; ++i) chk ^= mailbox_mix(io->P_post[i], (uint32_t)i);
            for (int i = 0; i < NX; ++i) chk ^= mailbox_mix(io->x[i], 1000u + (uint32_t)i);
            chk ^= mailbox_mix((double)io->passes, 2000u);
            if (chk == MAILBOX_UNCHECKED) chk = 0u;
            // (MAILBOX_UNCHECKED: the update ended on a pass without matches — a legitimate path that carries no
            // checksum: synchronise the stream without counting it)
            if (want == MAILBOX_UNCHECKED) seen = false;
            else if (chk != want) { seen = false; ++c->mailbox_resyncs; }
        }
    }
    return seen;
}

int lv_update_end(lv_ctx* c, lv_state* x, double* P, int* passes) {
    LV_CHECK_CTX(c);
    if (!c->in_update) { set_error("lv_update_end without lv_update_begin"); return LV_ESTATE; }
    c->in_update = false;
    // results were stored into the pinned mailbox by kf_begin_kernel / solve_kernel; the device copy of the logs
    // is only downloaded when the caller asked for them
    if (c->want_log) {
        LV_HIP(hipMemcpyAsync(c->h_kf, c->d_kf, offsetof(KfDev, pose), hipMemcpyDeviceToHost, c->stream));
        LV_HIP(hipStreamSynchronize(c->stream));
    } else {
        const bool seen = mailbox_wait(c);
        if (!seen) LV_HIP(hipStreamSynchronize(c->stream));
    }
    const KfHostIO* io = c->h_io;
    if ((unsigned)io->fallback_queries & KF_FAULT_BIT) { c->in_update = false; return report_kf_fault(c, io->fallback_queries); }
    c->timing.fallback_queries = fallback_count(io->fallback_queries) - c->fallback_base;
    c->fallback_base = fallback_count(io->fallback_queries);
    if (x) std::memcpy(x, io->x, sizeof(double) * NX);
    if (P) std::memcpy(P, io->P_post, sizeof(double) * NS * NS);
    if (passes) *passes = io->passes;
    c->h_kf->passes = io->passes;
    c->timing.last_passes = c->h_kf->passes;
    if (c->prm.degeneracy_mode && c->prm.print_degeneracy_values) {   // print_degeneracy_values (config/params.yaml:53)
        double eig[MAX_PASSES * 6];
        int np = 0;
        if (lv_get_degeneracy_values(c, eig, MAX_PASSES, &np) == LV_OK)
            for (int i = 0; i < np; ++i)
                fprintf(stderr, "degeneracy eigenvalues (pass %d): %.6g %.6g %.6g %.6g %.6g %.6g\n", i, eig[6 * i], eig[6 * i + 1],
                        eig[6 * i + 2], eig[6 * i + 3], eig[6 * i + 4], eig[6 * i + 5]);
    }
    return LV_OK;
}

int lv_update(lv_ctx* c, lv_state* x, double* P, int* passes, lv_sums* per_pass, double* trace) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    if (!x || !P) { set_error("null argument"); return LV_EINVAL; }
    if (passes) *passes = 0;
    c->state_src = 2;
    if (c->map.view.m == 0) {  // Localizator::correct returns without a map (Localizator.cpp:24)
        // ... but the state the caller propagated is the one the first map is built with (main.cpp:92,102 -> lv_map_add_scan)
        LV_FLUSH_PREDICTS(c);                        // (a resident filter that still lives in kf moves out before kf->x is overwritten)
        { int rm = materialise_filter(c); if (rm) return rm; }
        LV_HIP(hipStreamSynchronize(c->stream));   // (x_in of an earlier update may still be in flight)
        std::memcpy(c->h_io->x_in, x, sizeof(double) * NX);
        LV_HIP(hipMemcpyAsync(c->d_kf->x, c->h_io->x_in, sizeof(double) * NX, hipMemcpyHostToDevice, c->stream));
        return LV_OK;
    }
    const int npass = c->prm.MAX_NUM_ITERS + 1;
    LV_CHECK_PEER(c);
    if (!x || !P) { set_error("lv_update: null argument"); return LV_EINVAL; }   // (before the snapshot below dereferences them)
    lv_state x_prior;
    std::vector<double> P_prior;
    if (c->peer.active) {   // (a failed exchange must hand the caller's buffers back untouched: lv_update_end writes into them)
        x_prior = *x;
        P_prior.assign(P, P + NS * NS);
    }
    int rc = lv_update_begin(c, x, P);
    if (rc) return rc;
    if (c->profiling) LV_HIP(hipEventRecord(c->ev_begin, c->stream));
    c->last_update_fused = false;
    c->coll_timed = false;
    if (gather_only(c) && !pass_fused_applies(c)) {
        c->in_update = false;
        set_error("host-staged / peer-mapped gather: this scan does not take the one-launch-per-pass form (largest shard told? size? options?)");
        return LV_ESTATE;
    }
    if (pass_fused_applies(c)) {
        rc = update_fused(c);
        if (rc) { c->in_update = false; return rc; }
    } else {
        for (int i = 0; i < npass; ++i) {
            if (c->profiling) LV_HIP(hipEventRecord(c->ev_pass[3 * i + 0], c->stream));
            c->ev_mid = c->profiling ? c->ev_pass[3 * i + 1] : nullptr;
            c->pass_index = i;
            rc = pass_full(c);
            c->pass_index = -1;
            c->ev_mid = nullptr;
            if (rc) { c->in_update = false; return rc; }
            if (c->profiling) LV_HIP(hipEventRecord(c->ev_pass[3 * i + 2], c->stream));
        }
    }
    if (c->profiling) LV_HIP(hipEventRecord(c->ev_end, c->stream));
    int np = 0;
    c->want_log = (per_pass != nullptr) || (trace != nullptr);
    rc = lv_update_end(c, x, P, &np);
    c->want_log = false;
    if (rc) return rc;
    if (c->peer.active) {
        // lv_update_end has waited for the update's last launch, so every pull of this update has run: a pull that gave up on a
        // peer (or met a poisoned flag) has set the sticky status word.  The caller gets its prior back, nothing is adopted.
        LV_HIP(hipStreamSynchronize(c->stream));
        if (peer_failed(c->peer)) {
            *x = x_prior;
            std::memcpy(P, P_prior.data(), sizeof(double) * NS * NS);
            return peer_poisoned(c);
        }
    }
    if (passes) *passes = np;
    if (per_pass)
        for (int i = 0; i < np && i < MAX_PASSES; ++i) unpack_sums(c->h_kf->sums_log + (size_t)i * SUMS_LEN, &per_pass[i]);
    if (trace) std::memcpy(trace, c->h_kf->trace, sizeof(double) * 49 * (size_t)(np < MAX_PASSES ? np : MAX_PASSES));
    if (c->profiling) {
        float ms = 0.f, r = 0.f, s = 0.f;
        hipEventElapsedTime(&ms, c->ev_begin, c->ev_end);
        c->timing.last_update_ms = ms;
        int cnt = np > 0 ? np : 1;
        for (int i = 0; i < np && i < npass; ++i) {
            float a = 0.f, b = 0.f;
            hipEventElapsedTime(&a, c->ev_pass[3 * i + 0], c->ev_pass[3 * i + 1]);
            hipEventElapsedTime(&b, c->ev_pass[3 * i + 1], c->ev_pass[3 * i + 2]);
            r += a;
            s += b;
            if (i < 8) {
                c->timing.pass_match_ms[i] = a;
                c->timing.pass_solve_ms[i] = b;
                float cm = 0.f;
                if (c->coll_timed) hipEventElapsedTime(&cm, c->ev_coll[2 * i], c->ev_coll[2 * i + 1]);
                c->timing.pass_collective_ms[i] = cm;
            }
        }
        c->timing.last_reduce_ms = r / cnt;
        c->timing.last_solve_ms = s / cnt;
    }
    return LV_OK;
}

// ---- resident filter (row f-3) ---------------------------------------------------------------------
int lv_filter_set(lv_ctx* c, const lv_state* x, const double* P) {
    LV_CHECK_CTX(c);
    if (!x || !P) { set_error("null argument"); return LV_EINVAL; }
    c->pred_n = 0;   // (queued predictions of a filter that is being replaced)
    c->pred_src_kf = false;
    // Nothing goes to the device here (round 4; rounds 1-3 synchronised the stream and uploaded): the filter waits in pinned host
    // memory.  An lv_correct that follows takes it along in its first launch's kernel arguments — exactly as lv_update takes its
    // x / P — so "set the prior, correct" enqueues without a copy, a begin kernel or a wait; anything else that needs the
    // filter on the device (lv_predict, lv_map_add_scan, a correct on a route without the argument hand-over) uploads it first.
    if (c->filter_up_pending) {   // (h_filter is about to be overwritten: an upload out of it must have completed)
        LV_HIP(hipEventSynchronize(c->ev_filter_up));
        c->filter_up_pending = false;
    }
    std::memcpy(c->h_filter->x, x, sizeof(double) * NX);
    std::memcpy(c->h_filter->P, P, sizeof(double) * NS * NS);
    c->filter_host = true;
    c->filter_set = true;
    c->filter_in_mailbox = false;
    c->filter_in_kf = false;
    c->state_src = 1;
    return LV_OK;
}

int lv_filter_get(lv_ctx* c, lv_state* x, double* P) {
    LV_CHECK_CTX(c);
    if (!c->filter_set) { set_error("lv_filter_get before lv_filter_set"); return LV_ESTATE; }
    LV_FLUSH_PREDICTS(c);
    if (c->filter_host) {   // (set and never touched since: it is still where lv_filter_set put it)
        if (x) std::memcpy(x, c->h_filter->x, sizeof(double) * NX);
        if (P) std::memcpy(P, c->h_filter->P, sizeof(double) * NS * NS);
        return LV_OK;
    }
    if (c->filter_in_mailbox && c->mail_filter) {
        // the resident filter is the posterior of the lv_correct just enqueued: its finishing pass stores x, P (and the pass count)
        // into the host-mapped mailbox as well — wait for THAT (a poll) instead of a copy + stream synchronise (~30 us of wake-up,
        // once per 100 Hz cycle: the reference's main loop reads the state after every correct, src/main.cpp:96-102)
        if (!mailbox_wait(c)) LV_HIP(hipStreamSynchronize(c->stream));
        const KfHostIO* io = c->h_io;
        if ((unsigned)io->fallback_queries & KF_FAULT_BIT) return report_kf_fault(c, io->fallback_queries);
        if (x) std::memcpy(x, io->x, sizeof(double) * NX);
        if (P) std::memcpy(P, io->P_post, sizeof(double) * NS * NS);
        return LV_OK;
    }
    { int rm = materialise_filter(c); if (rm) return rm; }
    LV_HIP(hipMemcpyAsync(c->h_filter, c->d_filter, sizeof(FilterDev), hipMemcpyDeviceToHost, c->stream));
    LV_HIP(hipStreamSynchronize(c->stream));
    if (x) std::memcpy(x, c->h_filter->x, sizeof(double) * NX);
    if (P) std::memcpy(P, c->h_filter->P, sizeof(double) * NS * NS);
    return LV_OK;
}

int lv_predict(lv_ctx* c, double dt, const double* Q, const double acc[3], const double gyro[3]) {
    LV_CHECK_CTX(c);
    if (!Q || !acc || !gyro) { set_error("null argument"); return LV_EINVAL; }
    if (!c->filter_set) { set_error("lv_predict before lv_filter_set"); return LV_ESTATE; }
    c->state_src = 1;
    c->filter_in_mailbox = false;
    if (c->pred_n > 0 && (c->pred_n >= PREDICT_BATCH || std::memcmp(c->pred_Q, Q, sizeof(c->pred_Q)) != 0)) LV_FLUSH_PREDICTS(c);
    if (c->pred_n == 0) {
        std::memcpy(c->pred_Q, Q, sizeof(c->pred_Q));
        c->pred_src_kf = c->filter_in_kf;
        c->filter_in_kf = false;
    }
    double* st = c->pred_steps[c->pred_n++];
    st[0] = dt;
    for (int i = 0; i < 3; ++i) { st[1 + i] = acc[i]; st[4 + i] = gyro[i]; }
    if (!c->batch_predict) LV_FLUSH_PREDICTS(c);
    return LV_OK;
}

int lv_correct(lv_ctx* c, int* passes) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    if (!c->filter_set) { set_error("lv_correct before lv_filter_set"); return LV_ESTATE; }
    LV_FLUSH_PREDICTS(c);
    c->state_src = 1;
    if (passes) *passes = 0;
    if (c->map.view.m == 0) return LV_OK;  // Localizator::correct returns without a map (Localizator.cpp:24)
    LV_CHECK_PEER(c);   // (a failed exchange of an EARLIER, asynchronous lv_correct surfaces here at the latest)
    if (gather_only(c) && !pass_fused_applies(c)) {   // (before anything is enqueued: the filter stays exactly where it was)
        set_error("host-staged / peer-mapped gather: this scan does not take the one-launch-per-pass form (largest shard told? size? options?)");
        return LV_ESTATE;
    }
    c->update_seq = (c->update_seq + 1) & 0x3fffffff;   // (the finishing pass echoes it into the mailbox: lv_filter_get polls for it)
    int rc;
    if (c->filter_host) {
        // the prior was just set by the host: it rides in the first launch's arguments (x followed by P: FilterDev = the layout
        // begin_device expects), like an update by value — no upload, no begin kernel
        static_assert(offsetof(FilterDev, P) == sizeof(double) * NX, "x followed by P");
        c->filter_host = false;
        rc = begin_device(c, reinterpret_cast<const double*>(c->h_filter), true, false);
    } else {
        rc = begin_device(c, nullptr, false, true);       // (kf_begin_kernel installs the filter in kf: no launch of its own)
    }
    if (rc) return rc;
    c->in_update = true;
    const int npass = c->prm.MAX_NUM_ITERS + 1;
    c->last_update_fused = false;
    if (pass_fused_applies(c)) {
        rc = update_fused(c);
    } else {
        for (int i = 0; i < npass && !rc; ++i) rc = pass_full(c);
    }
    c->in_update = false;
    if (rc) {
        // an enqueue failed part of the way: kf_begin_kernel has installed the prior in kf->x / kf->P_post (P_post is only
        // rewritten by the pass that ends an update), but passes that did run may have moved kf->x.  Neither "the filter is
        // d_filter" (possibly older than the prior) nor "the filter is kf" (possibly a half-iterated state) is right:
        // the filter is declared unset and the caller re-seeds it.
        c->filter_set = false;
        c->filter_in_kf = false;
        c->filter_in_mailbox = false;
        return rc;
    }
    c->filter_in_kf = true;        // (the posterior stays where the update left it: materialise_filter)
    c->filter_in_mailbox = true;
    if (passes) {  // optional: the only synchronisation point
        LV_HIP(hipStreamSynchronize(c->stream));
        LV_CHECK_PEER(c);
        *passes = c->h_io->passes;   // stored by solve_kernel into the pinned mailbox
    }
    return LV_OK;
}

static int fetch_check(lv_ctx* c) {
    if (!c->dbg_valid || !c->dbg.knn_idx) { set_error("no captured pass: call lv_iterate (or lv_set_capture(1) before lv_update)"); return LV_ESTATE; }
    LV_HIP(hipStreamSynchronize(c->stream));
    return LV_OK;
}

int lv_fetch_knn(lv_ctx* c, uint32_t* idx, float* d2) {
    LV_CHECK_CTX(c);
    LV_SETTLE_MAP(c);
    int rc = fetch_check(c);
    if (rc) return rc;
    const size_t n = c->scan.n, K = (size_t)c->prm.NUM_MATCH_POINTS;
    if (idx) {
        LV_HIP(hipMemcpy(idx, c->dbg.knn_idx, n * K * sizeof(uint32_t), hipMemcpyDeviceToHost));
        if (c->map.n_ids != c->map.m) {   // the kernels report point ids; the API's index space is the rank among the living
            const size_t ids = c->map.n_ids;
            std::vector<float4> pts(ids);
            LV_HIP(hipMemcpy(pts.data(), c->map.d_orig, ids * sizeof(float4), hipMemcpyDeviceToHost));
            std::vector<uint32_t> rank(ids);
            uint32_t r = 0;
            for (size_t i = 0; i < ids; ++i) { rank[i] = r; r += std::isfinite(pts[i].x) ? 1u : 0u; }
            for (size_t i = 0; i < n * K; ++i)
                if (idx[i] != 0xFFFFFFFFu && idx[i] < ids) idx[i] = rank[idx[i]];
        }
    }
    if (d2) LV_HIP(hipMemcpy(d2, c->dbg.knn_d2, n * K * sizeof(float), hipMemcpyDeviceToHost));
    return LV_OK;
}

int lv_fetch_matches(lv_ctx* c, uint8_t* valid, float* p_world, float* abcd, float* dist) {
    LV_CHECK_CTX(c);
    int rc = fetch_check(c);
    if (rc) return rc;
    const size_t n = c->scan.n;
    if (valid) LV_HIP(hipMemcpy(valid, c->dbg.valid, n, hipMemcpyDeviceToHost));
    if (p_world) LV_HIP(hipMemcpy(p_world, c->dbg.p_world, n * 3 * sizeof(float), hipMemcpyDeviceToHost));
    if (abcd) LV_HIP(hipMemcpy(abcd, c->dbg.abcd, n * 4 * sizeof(float), hipMemcpyDeviceToHost));
    if (dist) LV_HIP(hipMemcpy(dist, c->dbg.dist, n * sizeof(float), hipMemcpyDeviceToHost));
    return LV_OK;
}

int lv_fetch_neighbors(lv_ctx* c, float* nbr_xyz, float* d2, float* p_world, int32_t* found) {
    LV_CHECK_CTX(c);
    const size_t n = c->scan.n;
    if (n == 0) return LV_OK;
    if (!c->qrec_valid || !c->d_qrec || c->qstride < n) { set_error("no pass has run on the current scan"); return LV_ESTATE; }
    LV_HIP(hipStreamSynchronize(c->stream));
    const int K = c->prm.NUM_MATCH_POINTS, NSL = qrec_slots(K);
    std::vector<float4> rec((size_t)NSL * n);
    for (int sl = 0; sl < NSL; ++sl)
        LV_HIP(hipMemcpy(rec.data() + (size_t)sl * n, c->d_qrec + (size_t)sl * c->qstride, n * sizeof(float4), hipMemcpyDeviceToHost));
    const float inf = std::numeric_limits<float>::infinity();
    const auto dword = [&](size_t q, int w) {   // word w of the distance slots: distance of neighbour w, then `found`
        const float4& v = rec[(size_t)(K + 1 + w / 4) * n + q];
        return (w & 3) == 0 ? v.x : (w & 3) == 1 ? v.y : (w & 3) == 2 ? v.z : v.w;
    };
    for (size_t q = 0; q < n; ++q) {   // records are in the scan's Morton order: slot K carries the original index
        uint32_t oq;
        std::memcpy(&oq, &rec[(size_t)K * n + q].w, 4);
        if (oq >= n) { set_error("corrupt hand-over record %zu", q); return LV_ESTATE; }
        int32_t fnd;
        const float fw = dword(q, K);
        std::memcpy(&fnd, &fw, 4);
        for (int j = 0; j < K; ++j) {
            const bool have = j < fnd;
            if (nbr_xyz) {
                nbr_xyz[((size_t)oq * K + j) * 3 + 0] = have ? rec[(size_t)j * n + q].x : 0.f;
                nbr_xyz[((size_t)oq * K + j) * 3 + 1] = have ? rec[(size_t)j * n + q].y : 0.f;
                nbr_xyz[((size_t)oq * K + j) * 3 + 2] = have ? rec[(size_t)j * n + q].z : 0.f;
            }
            if (d2) d2[(size_t)oq * K + j] = have ? dword(q, j) : inf;
        }
        if (p_world) { const float4& w = rec[(size_t)K * n + q]; p_world[(size_t)oq * 3] = w.x; p_world[(size_t)oq * 3 + 1] = w.y; p_world[(size_t)oq * 3 + 2] = w.z; }
        if (found) found[oq] = fnd;
    }
    return LV_OK;
}

int lv_fetch_rows(lv_ctx* c, double* H, double* h) {
    LV_CHECK_CTX(c);
    int rc = fetch_check(c);
    if (rc) return rc;
    const size_t n = c->scan.n;
    if (H) LV_HIP(hipMemcpy(H, c->dbg.rows, n * 12 * sizeof(double), hipMemcpyDeviceToHost));
    if (h) LV_HIP(hipMemcpy(h, c->dbg.h, n * sizeof(double), hipMemcpyDeviceToHost));
    return LV_OK;
}

int lv_calculate_H(lv_ctx* c, const lv_state* x, const float* p_world, const float* abcd, const float* dist, size_t n,
                   double* H, double* h) {
    LV_CHECK_CTX(c);
    if (!x || (n && (!p_world || !abcd || !dist || !H || !h))) { set_error("null argument"); return LV_EINVAL; }
    if (n == 0) return LV_OK;
    if (c->in_update) { set_error("lv_calculate_H inside an update"); return LV_ESTATE; }
    int rc = begin_common(c, x, nullptr, false);  // derives the pass constants from x on the device (begin kernel)
    if (rc) return rc;
    float *d_in = nullptr;
    double* d_out = nullptr;
    LV_HIP(hipMalloc(&d_in, n * 8 * sizeof(float)));
    LV_HIP(hipMalloc(&d_out, n * 13 * sizeof(double)));
    float* d_pw = d_in;
    float* d_abcd = d_in + 3 * n;
    float* d_dist = d_in + 7 * n;
    hipError_t e1 = hipMemcpyAsync(d_pw, p_world, n * 3 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    hipError_t e2 = hipMemcpyAsync(d_abcd, abcd, n * 4 * sizeof(float), hipMemcpyHostToDevice, c->stream);
    hipError_t e3 = hipMemcpyAsync(d_dist, dist, n * sizeof(float), hipMemcpyHostToDevice, c->stream);
    rc = (e1 != hipSuccess || e2 != hipSuccess || e3 != hipSuccess) ? LV_EHIP : LV_OK;
    if (!rc) rc = launch_rows_from_matches(c->stream, c->d_kf, d_pw, d_abcd, d_dist, (uint32_t)n, c->prm.estimate_extrinsics, d_out, d_out + n * 12);
    if (!rc && hipMemcpyAsync(H, d_out, n * 12 * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = LV_EHIP;
    if (!rc && hipMemcpyAsync(h, d_out + n * 12, n * sizeof(double), hipMemcpyDeviceToHost, c->stream) != hipSuccess) rc = LV_EHIP;
    if (hipStreamSynchronize(c->stream) != hipSuccess) rc = LV_EHIP;
    hipFree(d_in);
    hipFree(d_out);
    if (rc == LV_EHIP) set_error("lv_calculate_H: HIP error");
    return rc;
}

int lv_get_degeneracy_values(lv_ctx* c, double* eig, int capacity_passes, int* n_passes) {
    LV_CHECK_CTX(c);
    if (n_passes) *n_passes = 0;
    if (!eig || capacity_passes < 0) { set_error("null argument"); return LV_EINVAL; }
    if (c->prm.degeneracy_mode == 0) { set_error("degeneracy_mode is 0: no eigenvalues are computed"); return LV_ESTATE; }
    LV_HIP(hipStreamSynchronize(c->stream));
    int np = 0;
    LV_HIP(hipMemcpy(&np, &c->d_kf->passes, sizeof(int), hipMemcpyDeviceToHost));
    if (np > MAX_PASSES) np = MAX_PASSES;
    if (np > capacity_passes) np = capacity_passes;
    if (np > 0 && c->last_update_fused) {
        // degeneracy_mode 1 is a REPORT (print_degeneracy_values, config/params.yaml:53): nothing in the update depends on it, so
        // the one-launch-per-pass form does not compute it on the device at all (a cyclic Jacobi is a serial f64 chain of ~75 us
        // in one lane) — the eigenvalues of a pass' pose block are derived here, on demand, from the sums record that pass'
        // bookkeeping logged (KfDev::sums_log), by the same fixed 8 sweeps as degeneracy_stage (lv_solve_dev.hpp)
        std::vector<double> log((size_t)np * SUMS_LEN);
        LV_HIP(hipMemcpy(log.data(), c->d_kf->sums_log, log.size() * sizeof(double), hipMemcpyDeviceToHost));
        for (int p = 0; p < np; ++p) host_pose_eigenvalues(&log[(size_t)p * SUMS_LEN], eig + 6 * p);
    } else if (np > 0) {
        LV_HIP(hipMemcpy(eig, c->d_kf->degen_eig, (size_t)np * 6 * sizeof(double), hipMemcpyDeviceToHost));
    }
    if (n_passes) *n_passes = np;
    return LV_OK;
}

int lv_get_level_histogram(lv_ctx* c, int out[8]) {
    LV_CHECK_CTX(c);
    if (!out) { set_error("null argument"); return LV_EINVAL; }
    LV_HIP(hipStreamSynchronize(c->stream));
    LV_HIP(hipMemcpy(out, c->d_kf->level_hist, 8 * sizeof(int), hipMemcpyDeviceToHost));
    return LV_OK;
}

int lv_get_solve_clocks(lv_ctx* c, long long* out, int capacity) {
    LV_CHECK_CTX(c);
    if (!out || capacity < MAX_PASSES * 16) { set_error("capacity must be >= %d", MAX_PASSES * 16); return LV_EINVAL; }
    LV_HIP(hipStreamSynchronize(c->stream));
    LV_HIP(hipMemcpy(out, c->d_kf->solve_clk, sizeof(long long) * MAX_PASSES * 16, hipMemcpyDeviceToHost));
    return LV_OK;
}

int lv_get_timing(lv_ctx* c, lv_timing* out) {
    if (!c || !out) { set_error("null argument"); return LV_EINVAL; }
    *out = c->timing;
    out->mailbox_resyncs = (int)c->mailbox_resyncs;
    return LV_OK;
}

int lv_set_profiling(lv_ctx* c, int enabled) {
    if (!c) { set_error("null context"); return LV_EINVAL; }
    c->profiling = enabled == 1;
    c->phase_clocks = enabled == 2;
    return LV_OK;
}

int lv_get_phase_clocks(lv_ctx* c, long long* out, int capacity_blocks, int* n_blocks) {
    LV_CHECK_CTX(c);
    if (!c->d_clk) { set_error("no phase clocks captured: lv_set_profiling(ctx, 2) first"); return LV_ESTATE; }
    // layout: grid x 8 shader-clock stamps, then grid x 8 records whose first two entries are wall_clock64()
    // (100 MHz, chip-global) at workgroup start / end
    const int nb = 2 * c->grid < capacity_blocks ? 2 * c->grid : capacity_blocks;
    LV_HIP(hipStreamSynchronize(c->stream));
    LV_HIP(hipMemcpy(out, c->d_clk, (size_t)nb * 8 * sizeof(long long), hipMemcpyDeviceToHost));
    if (n_blocks) *n_blocks = nb;
    return LV_OK;
}

}  // extern "C"
