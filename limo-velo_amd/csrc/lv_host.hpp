// lv_host.hpp — host-side state behind the C-ABI (include/limovelo_hip.h).
#pragma once
#include "lv_note.hpp"

#include <cstdarg>
#include <cstdio>
#include <vector>

#include "../../include/limovelo_hip.h"
#include "lv_device.hpp"
#include "lv_mapinc.hpp"

namespace lv {

void set_error(const char* fmt, ...);

#define LV_HIP(expr)                                                                              \
    do {                                                                                          \
        hipError_t _e = (expr);                                                                   \
        if (_e != hipSuccess) {                                                                   \
            ::lv::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(_e)); \
            return LV_EHIP;                                                                       \
        }                                                                                         \
    } while (0)

struct MapStats {   // == lv_map_stats (include/limovelo_hip.h)
    uint64_t living, ids, capacity;
    uint64_t pool_used[4], pool_cap[4];
    uint64_t slots_used[4], slots_cap[4];
    uint64_t tombstones, dropped;
    uint64_t relinearisations, incremental_adds;
    uint64_t bytes;
};

void set_slice_pause_us(uint32_t us);   // lv_map.hip: the calling THREAD's sliced launches are spaced by that many microseconds (0: back to back)
struct MapStore {
    // ---- points by id (insertion order; deleted ids keep their slot with x = +inf)
    float4* d_orig = nullptr;
    float4* d_orig2 = nullptr;     // compaction target (swapped with d_orig by relinearise)
    size_t capacity = 0;           // id slots allocated
    uint32_t n_ids = 0;            // ids handed out
    uint32_t m = 0;                // living points
    // ---- build scaffolding: Morton-sorted copy + per-level occupancy tables over it
    float4* d_sorted = nullptr;
    uint64_t* d_keys = nullptr;
    uint64_t* d_keys_sorted = nullptr;
    uint32_t* d_idx = nullptr;
    uint32_t* d_idx_sorted = nullptr;
    void* d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    uint32_t* d_counts = nullptr;
    uint4* d_tables[N_OCC] = {};           // occupied voxels -> run in d_sorted: [0] level 0, [OCC_CELL] level 2 (lives on as the voxel-list table)
    uint32_t table_size[N_OCC] = {};
    uint32_t n_cells[N_OCC] = {};
    // ---- level 0: neighbourhood buckets with slack (built straight into their pool: no float4 staging copy since round 6)
    uint4* d_btable[REPL_LEVELS] = {};
    SlotAux* d_baux[REPL_LEVELS] = {};
    uint32_t btable_size[REPL_LEVELS] = {};
    float* d_bxyz[SORTED_LEVELS] = {};    // 12-byte points: what the search kernel streams
    uint32_t* d_bidx[SORTED_LEVELS] = {}; // ids (ascending inside a bucket)
    uint16_t* d_backpos = nullptr;        // [id * 27 + c]: position of a point inside the bucket of its neighbour c (BACKPOS_FAR: beyond 16 bits,
                                          // found by binary search): a deletion is 27 probes + 27 direct writes
    uint32_t* d_cellpos = nullptr;        // [id]: position of a point in its voxel's list (allocated with d_backpos)
    size_t backptr_cap = 0;               // ids they are allocated for
    uint32_t* d_biglist = nullptr;        // build scratch: buckets of more than 64 points (a workgroup each)
    size_t biglist_cap = 0;
    // tile groups (lv_device.hpp REPL_LEVELS): level-1 voxel -> {start, extent} of the region its eight level-0 runs were laid out in
    uint4* d_gtable = nullptr;
    uint32_t gtable_size = 0;
    uint32_t* d_gext = nullptr;           // build scratch: a group's extent while its runs take their places, then the groups' offsets
    uint32_t* d_goff = nullptr;
    size_t gscratch_cap = 0;
    uint32_t* d_broken = nullptr;         // groups an insert batch broke up (table slots) + their re-layout plans (lv_mapinc.hpp inc_regroup_*)
    RegroupPlan* d_regroup = nullptr;
    uint32_t broken_cap = 0;
    uint32_t n_groups = 0;
    uint4* d_comp = nullptr;              // runs an insert batch compacts in place (lv_mapinc.hpp inc_compact_*), their staging area
    float4* d_cstage = nullptr;
    uint32_t* d_cnew = nullptr;
    uint32_t comp_cap = 0, cstage_cap = 0;
    size_t pool_cap[INC_LEVELS] = {};     // entries per pool ([CELL_SLOT]: d_cell4)
    uint32_t pool_base[INC_LEVELS] = {};  // entries laid out by the last (re)build; the rest is split into arenas
    uint32_t n_bcells[REPL_LEVELS] = {};
    uint32_t* d_cell_slots = nullptr;  // scratch: table slots of the bucket voxels of the level being built
    uint32_t* d_bcount = nullptr;
    uint32_t* d_bcap = nullptr;
    uint32_t* d_boff = nullptr;
    size_t cells_cap = 0;
    uint32_t* d_flags = nullptr;       // [0] = append cursor, [1] = overflow flag
    void* d_scan_tmp = nullptr;
    size_t scan_tmp_bytes = 0;
    // ---- level 2: one list per voxel
    SlotAux* d_caux = nullptr;
    uint32_t caux_size = 0;
    float4* d_cell4 = nullptr;
    // ---- incremental maintenance (lv_mapinc.hpp)
    MapCounters* d_cnt = nullptr;
    MapCounters* h_cnt = nullptr;      // pinned mirror
    float4* d_new = nullptr;           // staged batch
    uint64_t* d_nkeys = nullptr;
    uint64_t* d_nkeys_sorted = nullptr;
    uint32_t* d_nidx = nullptr;
    uint32_t* d_nidx_sorted = nullptr;
    uint32_t* d_nalive = nullptr;
    uint32_t* d_napos = nullptr;
    uint32_t* d_nsurv = nullptr;       // the batch's survivors in sorted-batch (Morton box) order, + the flags / positions that build it
    uint32_t* d_nsflag = nullptr;
    uint32_t* d_nspos = nullptr;
    uint32_t* d_rank = nullptr;
    void* d_ntmp = nullptr;
    size_t ntmp_bytes = 0, batch_cap = 0;
    // voxel groups of a batch (lv_mapinc.hpp GroupRW)
    uint4* d_gtab[REPL_LEVELS] = {};
    uint32_t gtab_size = 0;
    uint32_t* d_prank = nullptr;
    uint32_t* d_pslot = nullptr;
    uint32_t* d_gbase[REPL_LEVELS] = {};
    uint32_t* d_gslot[REPL_LEVELS] = {};
    uint4* d_gdst[REPL_LEVELS] = {};
    uint32_t* d_gcnt = nullptr;        // [0] relocations of the batch
    uint4* d_reloc = nullptr;
    uint32_t reloc_cap = 0;
    float4* d_dead = nullptr;
    size_t dead_cap = 0;
    uint32_t* d_alive = nullptr;       // flags / ranks over all ids (eviction of the oldest, compaction)
    uint32_t* d_apos = nullptr;
    void* d_ascan_tmp = nullptr;
    size_t ascan_tmp_bytes = 0, alive_cap = 0;
    // 0.2 m boxes of ikd-Tree's down-sampling insert
    uint4* d_box = nullptr;
    uint32_t* d_box_next = nullptr;
    uint32_t box_size = 0;
    size_t box_next_cap = 0;
    bool have_boxes = false;

    // the tail of an incremental insert (new ids / living points / overflow, read from the device counters) is settled by the
    // next call that needs the map's bookkeeping, not by a wait at the end of the insert
    bool sweep_evict = true;     // lv_map_evict_box tests runs, not points (inc_evict_sweep_kernel; LV_SWEEP_EVICT=0 / lv_set_option: the per-point search)
    bool merged_back = true;     // independent stages of the insert's back half share launches (LV_MERGED_INSERT=0 / lv_set_option: off)
    bool surv_list = true;       // large down-sampling batches: the append passes walk a survivor list in Morton (box sort) order (LV_SURV_LIST=0 / "survivor_list": off)
    bool small_front = true;     // batches of up to 2048 points: the insert's front half in one workgroup launch (LV_SMALL_INSERT=0: off)
    NoteBoard notes;             // the insert's counters come back as a note (lv_note.hpp): n_new, n_dead, dropped, overflow
    uint32_t counters_seq = 0;
    hipStream_t counters_stream = nullptr;
    bool counters_pending = false;
    bool pending_counted_kill = false;
    uint32_t pending_n_dead = 0;
    int settle(hipStream_t stream);
    bool origin_set = false;
    float origin[3] = {0, 0, 0};
    float cell = 0.5f;
    float bbox_min[3], bbox_max[3];    // of everything ever inserted since the origin was chosen
    bool built = false;                // the search structure describes d_orig[0 .. n_ids)
    uint64_t relinearisations = 0, incremental_adds = 0, dropped_total = 0, tombstones = 0;
    MapView view{};

    int build_buckets(hipStream_t stream, int level, uint32_t n_occupied);
    int build_cells(hipStream_t stream, uint32_t n_occupied);
    int reserve(size_t cap);           // room for `cap` ids (keeps d_orig[0 .. n_ids))
    int rebuild(hipStream_t stream);   // search structure over d_orig[0 .. n_ids) (all of them living)
    // compaction of the living (ids become ranks) + rebuild
    int relinearise(hipStream_t stream);
    // k points staged in d_new[0 .. k): ikd-Tree Add_Points(points, downsample) without a rebuild; falls back to
    // relinearise when a pool or table runs full.  Synchronises the stream.
    // build_if_empty: Mapper::add's rule (an empty map is BUILT from the points, Mapper.cpp:22-27) instead of
    // Add_Points's (the box rule applies among the new points themselves)
    int add_staged(hipStream_t stream, uint32_t k, int downsample, float box_length, bool build_if_empty);
    int ensure_counters();
    int reserve_batch(size_t k);
    int evict_box(hipStream_t stream, const float lo[3], const float hi[3], int keep_inside, uint32_t* n_evicted);
    int evict_oldest(hipStream_t stream, uint32_t n_oldest, uint32_t* n_evicted);
    int kill_dead_list(hipStream_t stream, uint32_t n_dead);
    int ensure_boxes(hipStream_t stream, float box_length);
    int ensure_alive_scratch();
    bool needs_relinearise(size_t incoming) const;
    // background re-linearisation (lv_api.hip, round 5): while a compacted copy of this map is being rebuilt on another stream /
    // thread, the stop-the-world relinearise is deferred (only id-space exhaustion still forces it)
    bool defer_relinearise = false;
    uint32_t last_new = 0;    // survivors of the previous insert batch: sizes the next batch's per-survivor launches (an estimate: add_staged)
    bool have_last_new = false;
    bool pool_low = false;    // the last insert left less than a quarter of the bucket pool's free part (settle): a re-linearisation is wanted
    uint32_t slice_wgs = 0;   // != 0: the large grids of a (re)build go out in slices of that many workgroups (a store rebuilt in the background)
    bool wants_relinearise(size_t incoming) const;   // the trigger, whatever defer_relinearise says
    // the living points of this map, compacted in id order, into dst.d_orig (dst: an idle store whose search structure is not
    // built yet): enqueued on `stream`, no host wait; m must be settled
    int snapshot_into(MapStore& dst, hipStream_t stream);
    MapRW rw() const;
    void refresh_view();
    void stats(MapStats* out) const;
    void release();
};

// lv_comm.hip — RCCL bound at run time (row e)
struct UniqueId128 { char internal[128]; };  // ncclUniqueId
int comm_unique_id(const char* library, void* id128);
int comm_init(const char* library, const void* id128, int rank, int world, void** comm_out);
int comm_destroy(void* comm);
int comm_allreduce_record(void* comm, double* record, hipStream_t stream);
bool comm_has_allgather();
int comm_allgather_inplace(void* comm, double* buf, size_t count_per_rank, int rank, hipStream_t stream);
// lv_peer.hip — the same exchange by peer-mapped memory (HIP IPC), no collective library
constexpr int LV_PEER_MAX = 8;
struct PeerSet {
    void* local_alloc = nullptr;                 // [gather buffer 0 | gather buffer 1]: coarse-grained; written by pass kernels, read by
                                                 // the peers only AFTER the kernel that wrote them has ended (a kernel boundary publishes it)
    void* flag_alloc = nullptr;                  // [flag word]: its own FINE-GRAINED allocation — polled across devices in the middle of a
                                                 // kernel, which coarse-grained memory does not guarantee to be coherent for
    bool buf_fine = false;                       // the gather slots are in fine-grained memory (false: plain allocation)
    bool flag_fine = false;                      // (false: the runtime refused a fine-grained IPC allocation; the flag lives in a plain one)
    double* buf[2] = {nullptr, nullptr};
    unsigned long long* flag = nullptr;
    uint32_t* h_status = nullptr;                // pinned, host-mapped: set by a pull that gave up (or met a poisoned flag); sticky
    uint32_t* d_status = nullptr;                // ... its device address
    void* mapped[LV_PEER_MAX] = {};              // the other ranks' gather allocations as mapped here
    void* mapped_flag[LV_PEER_MAX] = {};         // ... and their flag allocations
    double* peer_buf[2][LV_PEER_MAX] = {};
    unsigned long long* peer_flag[LV_PEER_MAX] = {};
    size_t cap = 0;                              // doubles per gather buffer
    int rank = 0, world = 1;
    unsigned long long seq = 0;                  // launches published so far (every rank counts alike)
    long long timeout_ticks = 0;                 // give-up time of a pull's wait, 100 MHz ticks (LV_PEER_TIMEOUT_MS, default 2000 ms)
    bool active = false;
};
constexpr int LV_PEER_BLOB = 128;               // two HIP IPC handles: gather buffers, flag word (= LV_PEER_HANDLE_BYTES of the ABI)
int peer_export(PeerSet& P, size_t cap_doubles, void* handle_blob);
int peer_init(PeerSet& P, int rank, int world, const void* handles);
int peer_gather(PeerSet& P, int parity, size_t slot_doubles, hipStream_t stream);
// true once a pull of this context gave up on a peer or met a poisoned flag (a plain read of the host-mapped word: meaningful
// for the launches that have completed)
bool peer_failed(const PeerSet& P);
void peer_close(PeerSet& P);

// lv_match.hip
// search_kernel writes one 128-byte record per scan point (8 float4 planes of qstride entries; qrec_slots(K) planes with K neighbours);
// fit_reduce_kernel turns them into `grid` block partials (and one extra workgroup runs solve_prep)
// begin != nullptr: the first launch of an update (no begin kernel ran): the state / covariance / pass constants
// travel as kernel arguments and one extra workgroup installs them in kf and the mailbox io
int launch_search(hipStream_t stream, int lanes_per_query, const MapView& map, const float4* scan_sorted, uint32_t n, KfDev* kf,
                  float4* qrec, uint32_t qstride, const uint32_t* tile_order, uint32_t n_tiles, double max_dist_sq, const DebugOut& dbg,
                  const BeginArg* begin, KfHostIO* io, int num_match = KNN);
int launch_fit_reduce(hipStream_t stream, const float4* qrec, uint32_t qstride, uint32_t n, KfDev* kf, const MatchParams& prm,
                      double* partials, int grid, const DebugOut& dbg, int num_match = KNN);   // num_match: NUM_MATCH_POINTS (3..8; 5 = the tuned build)
int fit_grid_size(uint32_t n, int max_blocks);
// lv_solve.hip
int launch_kf_begin(hipStream_t stream, KfDev* kf, KfHostIO* io, const double* x_host, const FilterDev* filt = nullptr, int filt_in_kf = 0);  // io: device pointer of the pinned mailbox;
// x_host != nullptr: x (NX doubles) followed by P_prop (NS*NS doubles) on the host, passed as kernel arguments
int launch_reduce_groups(hipStream_t stream, const double* partials, int nblocks, double* groups, int* ngroups_out, KfDev* kf);
int solve_direct_records();  // most records solve_kernel folds in one round trip
int launch_reduce_final(hipStream_t stream, const double* groups, int ngroups, double* sums, KfDev* kf);
struct SolveParams {
    double R;
    double R_inv;
    double limits[NS];
    int maximum_iter;
    int estimate_extrinsics;
    int seq;   // written to the host mailbox by the pass that finishes the update
    int degeneracy_mode;           // lv_params
    double degeneracy_threshold;
};
int launch_solve(hipStream_t stream, KfDev* kf, KfHostIO* io, const double* recs, int nrec, double* sums_out, const SolveParams& prm);
// lv_match.hip — one launch per pass (pass_kernel): prologue solve of the previous pass in every workgroup + search +
// plane fits + one compact partial per workgroup (lv_pass_dev.hpp)
struct PassLaunch {
    const MapView* map;
    const float4* scan;
    const uint32_t* tile_order;   // 32-point tiles (ScanStore::order_tiles(32)) or nullptr
    uint32_t n;
    KfDev* kf;
    KfHostIO* io;
    const double* recs_in;        // compact partials of the previous launch (mode 1), nrec of them
    double* part_out;
    double* sums_out;
    float4* qrec;                 // optional (debug): the hand-over records also go to memory, 8 planes of qstride entries
    long long* clk;               // optional (instrumentation): 16 stamps per search workgroup
    uint32_t qstride;
    int nrec;
    int mode;                     // 0: the update starts here (state in the BeginArg); 1: solve the previous pass first; 2: state already in kf
    int rounds, launch, nwg;      // nwg searching workgroups; rounds = 0: closing launch (one workgroup, no search);
                                  // launch = index of the launch in the update
    const uint32_t* cost_in;      // per searching workgroup: time of its search + fits in the previous launch (nullptr: none) ...
    uint32_t* cost_out;           // ... and where this launch leaves its own
    bool multi_overlap = true;    // rounds > 1: a round's plane fits beside the next round's search (pass_kernel<.., MULTI>); false: round 3's barrier form
    int steps, dedicated;         // search steps per round (1 or 2); dedicated != 0: one more workgroup only keeps the books
                                  // (otherwise the last searching workgroup does, after its own fits)
    MatchParams mp;
    SolveParams sp;
};
void pass_grid_size(uint32_t n, int max_wg, int* nsearch, int* steps, int* rounds, int* dedicated);
constexpr int PK_DEFAULT_MAX_ROUNDS = 16;   // rounds per workgroup up to which lv_update takes pass_kernel by default (1 M points on 256 CUs)
int pass_clock_words();   // stamp words per workgroup (PassLaunch::clk)
int launch_pass(hipStream_t stream, const PassLaunch& pl, const BeginArg* begin);
// lv_predict.hip
// src != nullptr: the state and covariance are read from kf->x / kf->P_post (the posterior of the update just run) instead of f
constexpr int PREDICT_BATCH = 8;   // (== PREDICT_BATCH_MAX of lv_predict.hip)
// n <= PREDICT_BATCH steps {dt, acc[3], gyro[3]} with one Q, in one launch
int launch_predict(hipStream_t stream, FilterDev* f, const KfDev* src, const double* Q, int n, const double (*steps)[7]);
int launch_filter_to_kf(hipStream_t stream, const FilterDev* f, KfDev* kf);
int launch_kf_to_filter(hipStream_t stream, const KfDev* kf, FilterDev* f);
// lv_rows.hip
int launch_rows_from_matches(hipStream_t stream, const KfDev* kf, const float* p_world, const float* abcd, const float* dist,
                             uint32_t n, int estimate_extrinsics, double* H, double* h);
// lv_scan.hip
struct CloudPoint;
struct ScanStore {
    float4* d_raw = nullptr;     // upload order; w = original index
    float4* d_sorted = nullptr;  // Morton order (LiDAR frame)
    uint32_t* d_keys = nullptr;
    uint32_t* d_keys_sorted = nullptr;
    uint32_t* d_idx = nullptr;
    uint32_t* d_idx_sorted = nullptr;
    void* d_sort_tmp = nullptr;
    size_t sort_tmp_bytes = 0;
    size_t capacity = 0;
    uint32_t n = 0;
    uint32_t* d_tile_order = nullptr;  // search-kernel tiles, farthest-from-sensor first (order_tiles)
    uint32_t n_tiles = 0, tile_cap = 0;
    uint32_t tile_points = 0;          // scan points per search-kernel workgroup (256 / lanes_per_query); 0: no ordering
    // row f-2 (de-skew + voxel grid): raw time-stamped input and work buffers
    float4* d_in = nullptr;
    double* d_times = nullptr;
    float4* d_desk = nullptr;
    MotionState* d_states = nullptr;
    uint64_t* d_vkeys = nullptr;
    uint64_t* d_vkeys_sorted = nullptr;
    uint32_t* d_vidx = nullptr;
    uint32_t* d_vidx_sorted = nullptr;
    uint32_t* d_heads = nullptr;
    uint32_t* d_hpos = nullptr;
    unsigned* d_bounds = nullptr;
    void* d_vsort_tmp = nullptr;
    void* d_vscan_tmp = nullptr;
    size_t vsort_tmp_bytes = 0, vscan_tmp_bytes = 0, raw_cap = 0, states_cap = 0;
    int reserve(size_t cap);
    int reserve_raw(size_t cap, size_t n_states);
    int deskew_downsample(hipStream_t stream, uint32_t n_in, uint32_t n_states, const MotionState& xt2, float leaf, float sort_cell);
    bool small_enabled = true;   // windows of up to 2048 points take the one-launch chain (LV_SMALL_WINDOW=0 / lv_set_option: off)
    int voxel_and_sort(hipStream_t stream, uint32_t n_in, float leaf, float sort_cell, bool try_small = false);
    bool small_window_applies(uint32_t n_in) const;
    int window_small(hipStream_t stream, const float4* src, uint32_t n_in, uint32_t n_states, const MotionState* xt2, float leaf,
                     float sort_cell, bool* fell_back);
    bool large_enabled = true;   // larger windows: de-skew + bounds, keys, sort, one tail workgroup (LV_LARGE_WINDOW=0 / lv_set_option: off)
    bool large_window_applies(uint32_t n_in, uint32_t n_states, float leaf) const;
    int window_large(hipStream_t stream, const struct CloudPoint* cloud, uint32_t n_in, uint32_t n_states, const MotionState& xt2, float leaf,
                     float sort_cell, bool* fell_back);
    NoteBoard notes;             // (points out, status) of the window kernels: lv_note.hpp
    long long* d_tail_clk = nullptr;   // LV_TAIL_CLK=1: phase stamps of window_tail_kernel
    int sort(hipStream_t stream, const float bbox_min[3], float cell);
    int order_tiles(hipStream_t stream, uint32_t tile_points);
    int reserve_tiles(uint32_t nt);
    void release();
};

// ---- row f-4: LiDAR wire formats -> device-resident LiDAR buffer (lv_cloud.hip) ----------------------------
struct CloudFormat {   // == lv_cloud_format
    uint32_t point_step, off_x, off_y, off_z, off_time;
    int time_type;
    uint32_t off_intensity;
    int intensity_type;
    uint32_t off_range;
    int range_type;
    int relative_time;
};
struct IngestParams {  // == lv_ingest_params
    uint64_t header_stamp_usec;
    int stamp_beginning, offset_beginning;
    double full_rotation_time;
    int downsample_rate;
    float min_dist;
};
struct CloudPoint {    // the reference's Point (include/Headers/Objects.hpp:20-28), 32 bytes
    float x, y, z, pad_;
    double time;
    float intensity, range;
};
static_assert(sizeof(CloudPoint) == 32, "reference Point layout");

struct CloudStore {
    CloudPoint* d_buf = nullptr;   // BUFFER_L: time ordered, live range [head, size)
    uint32_t head = 0, size = 0;
    size_t buf_cap = 0;
    unsigned char* d_rawmsg = nullptr;
    unsigned char* h_rawmsg = nullptr;   // pinned staging, two buffers alternating between messages
    unsigned char* h_rawmsg2 = nullptr;
    hipEvent_t ev_stage[2] = {nullptr, nullptr};
    bool stage_busy[2] = {false, false};
    int stage_next = 0;
    size_t raw_cap = 0;
    CloudPoint* d_decoded = nullptr;
    CloudPoint* d_kept = nullptr;
    unsigned char* d_keep = nullptr;
    uint64_t* d_keys = nullptr;
    uint64_t* d_keys_sorted = nullptr;
    uint32_t* d_ids = nullptr;
    uint32_t* d_ids_sorted = nullptr;
    void* d_tmp = nullptr;
    size_t tmp_bytes = 0, msg_cap = 0;
    uint32_t* d_count = nullptr;
    uint32_t* h_count = nullptr;   // pinned
    // clear_before does not wait for its result (the new head): the kernel posts it as a note (lv_note.hpp) that is picked up
    // by whoever touches the buffer next (settle), by which time it has long arrived — the wait was one of six host/device
    // round trips of a 100 Hz cycle; the window's index range comes back as a note too (words 0, 1; the clear's: word 3)
    NoteBoard notes;
    double clear_t = 0.0;        // the pending Buffer::clear(t)
    hipStream_t clear_stream = nullptr;
    bool clear_pending = false;
    int settle();
    int init();
    int reserve_msg(size_t n, size_t bytes);
    int reserve_buffer(hipStream_t stream, size_t total);
    int ingest(hipStream_t stream, const void* data, size_t n, const CloudFormat& fmt, const IngestParams& prm, double begin_time,
               size_t* n_kept);
    // reset8 != nullptr: the kernel also sets those eight words to the voxel grid's "no bounds seen" (ScanStore::d_bounds)
    int window(hipStream_t stream, double t1, double t2, uint32_t* lo, uint32_t* hi, unsigned* reset8 = nullptr);
    int clear_before(hipStream_t stream, double t);
    int unpack(hipStream_t stream, uint32_t lo, uint32_t n, float4* xyz, double* times);
    void release();
};

}  // namespace lv
