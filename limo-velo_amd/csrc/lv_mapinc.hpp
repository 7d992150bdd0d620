// lv_mapinc.hpp — incremental maintenance of the map's search structure (row f-1 of SURVEY.md §8).
//
// Replaces, per mapping cycle, KD_TREE<Point>::Add_Points(points, downsample) (call site reference
// src/Modules/Mapper.cpp:73-76, every cycle from src/main.cpp:102) without rebuilding anything: only the buckets /
// voxel lists whose 27-block contains a new or deleted point are touched.
//
//   points      ids are handed out in insertion order and never reused; a deleted point keeps its slot in `orig`
//               with x = +inf.  Relative order of the living == the reference's map order.
//   level 0     neighbourhood buckets with slack: a new point is appended to the 27 buckets of its voxel's
//               neighbours (ids only grow, so "ascending id" survives an append once the batch's own tail is
//               ordered); a bucket that runs out of room moves to fresh space at the end of its pool.  Every point
//               remembers its position in each of its 27 buckets (backpos, 16 bits: positions are relative to the
//               run's start, so a run that moves keeps them), so a deletion is 27 table probes + 27 direct writes
//               (x set to +inf: tombstone); rounds 1-5 searched the bucket's ids (266 DRAM lines per deleted point
//               over three replicated levels).  Level 1 has no storage (eight level-0 buckets tile its block).
//   voxel lists one list per level-2 voxel, unordered: append / tombstone in the point's own voxel only.
//   0.2 m boxes ikd-Tree's down-sampling rule needs "the points currently in this box": a hash table
//               box -> chain of ids (box_next), built lazily by the first down-sampling insert.
//
// Every kernel here is one-thread-per-item with plain atomics (no wave intrinsics, no LDS, no barriers), so the
// same source also compiles for the host in tests/emu (a sequential emulation that checks the bookkeeping against
// the CPU restatement of the reference's rule); the product only ever runs them on the GPU.
//
// An insert batch of k points (already staged on the device):
//   box_keys -> stable sort by box -> box_rule (sequential replay per box, [UPSTREAM-RECALL ikd-Tree]) -> scan
//   -> commit_points (ids, orig, box chains) -> group (new points by voxel, per level) -> kill (tombstones for the
//   occupants that lost) -> register (per voxel group: find / create its 27 target slots, reserve tail shares) ->
//   reserve + relocate (room) -> fill (ids into the tails) -> rank (order of each tail) -> place -> commit (counts).  Any pool or table running full raises
//   `overflow`: the mutating kernels then do nothing and the host re-linearises (compaction + full rebuild).
#pragma once

#include "lv_device.hpp"

namespace lv {

constexpr uint32_t ID_NONE = 0xFFFFFFFFu;
constexpr int INC_SLOTS_PER_POINT = 27 * REPL_LEVELS + 1;   // its 27 level-0 buckets + its level-2 voxel's list
constexpr int INC_LEVELS = REPL_LEVELS + 1;                 // tables: bt[0], voxel lists
constexpr uint16_t BACKPOS_FAR = 0xFFFFu;                   // back-position of an entry beyond 16 bits: found by binary search over the run's ids
constexpr int CELL_SLOT = REPL_LEVELS;                      // index of the voxel-list table in the per-table arrays

constexpr int N_ARENAS = 64;   // the free part of every pool is split into arenas with their own cursors: a run that
                               // needs room bumps the cursor of arena (slot mod 64) — one shared cursor costs ~10 ns per
                               // allocation (same-address atomics whose result is needed), 200 us when 20k buckets are new
// Work lists of an insert batch (runs that move, runs compacted in place, tile groups broken up): 64 sub-lists with their own
// cursors — shard s owns the indices [s * cap / 64, (s + 1) * cap / 64) — because an append needs the atomic's RESULT, and such
// an atomic costs ~10 ns when a whole launch hits one address: 30 000 appends to one cursor were most of inc_reserve_kernel's
// 120 us (round 6; the arenas above are the same cure for the pool cursors).
constexpr uint32_t LIST_SHARDS = 64;
struct MapCounters {
    uint32_t arena_cur[INC_LEVELS][N_ARENAS];   // next free entry of each arena
    uint32_t arena_end[INC_LEVELS][N_ARENAS];
    uint32_t slots_used[INC_LEVELS];   // occupied table slots
    uint32_t n_new;                    // surviving new points of the batch
    uint32_t n_dead;                   // length of the dead list
    uint32_t overflow;                 // a pool, a table or a work list ran full: re-linearise
    uint32_t dropped;                  // new points that were not finite or outside the voxel range
    uint32_t box_slots_used;
    uint32_t tombstones;               // (unused on the device: the host counts INC_SLOTS_PER_POINT per deleted point)
    uint32_t gslots_used;              // occupied slots of the tile-group table
};

struct LevelRW {
    uint4* table;
    SlotAux* aux;
    uint32_t mask;
    uint32_t shift;
    uint32_t slot_limit;   // occupied slots beyond this raise `overflow` (load factor guard)
    uint32_t pool_cap;     // entries in the level's pool
};

struct MapRW {
    float4* orig;
    LevelRW lv[INC_LEVELS];          // [0]: bucket table; [CELL_SLOT]: level-2 voxel-list table
    float* bxyz[SORTED_LEVELS];
    uint32_t* bidx[SORTED_LEVELS];
    uint16_t* backpos;               // [id * 27 + c]: position of point id inside the bucket of its neighbour c (BACKPOS_FAR: see above)
    uint4* gtable;                   // tile groups: level-1 voxel -> {start, extent} of its eight runs' region (extent 0: not in one piece)
    uint32_t gmask, gshift, gslot_limit;
    uint32_t* broken;                // optional: the groups broken up by the batch in flight (table slots), re-laid out by inc_regroup_*
    uint32_t broken_cap;
    uint32_t* n_broken;
    // optional (an insert batch): runs that outgrew their room only because of the deleted entries they carry are compacted IN PLACE
    // (inc_compact_*) instead of moving: {table slot, living entries, entries before, first staging entry} per run, the staging area
    uint4* comp;
    uint32_t comp_cap;
    uint32_t* n_comp;                // [0 .. 64): the list's cursors, [64 .. 128): staging entries handed out per shard
    float4* cstage;                  // {x, y, z, id} of a listed run's entries as they were
    uint32_t* cnew;                  // their new positions (ID_NONE: a deleted entry)
    uint32_t cstage_cap;
    uint32_t* cellpos;               // [id]: position of point id inside its voxel's list
    float4* cell4;
    float origin[3];
    float inv_cell;
    MapCounters* cnt;
};

struct BoxRW {
    uint4* table;        // {key lo, key hi, head id, -}; memset 0xFF: EMPTY key, head ID_NONE
    uint32_t* next;      // chain link by id
    uint32_t mask;
    uint32_t shift;
    uint32_t slot_limit;
    float len;           // box_length (0.2 m, Mapper.cpp:65)
};

struct RegroupPlan {   // per listed group: the table slot, old start, new start, count and capacity of its tiles (slot ID_NONE: no such bucket)
    uint32_t slot[8], from[8], to[8], count[8], cap[8];
};

// The kernels below are compiled by ONE translation unit (lv_map.hip, or the host emulation in tests/emu): it
// defines LV_MAPINC_KERNELS before including this header; everybody else only sees the types above.
#ifdef LV_MAPINC_KERNELS

// XCD-aware work order.  Consecutive workgroup indices are dealt round-robin to the eight XCDs, each with its own L2; the work
// items of the append / tombstone passes are spatially ordered (inc_box_key), so workgroup b takes the block of items that
// keeps every XCD on ONE contiguous stretch of the list — neighbours that share table slots, run tails and id arrays then
// share an L2 as well.  A bijection of [0, n) for every n.
__device__ __forceinline__ uint32_t inc_block_of(uint32_t b, uint32_t n) {
    const uint32_t q = n / 8u, r = n % 8u, x = b % 8u;
    return x * q + (x < r ? x : r) + b / 8u;
}
__device__ __forceinline__ uint32_t inc_thread_id() { return inc_block_of(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x; }
// The per-survivor passes of an insert are launched for an ESTIMATE of the batch's survivors (what the previous batch left: the
// host no longer waits for the count in the middle of the insert) and walk any items beyond their grid in strides: `per` items
// per surviving point, `first` / `stride` from the workgroups [b0, b0 + nb) that serve the stage.
#define LV_INC_ITEMS(per, b0, nb, ...)                                                                                          \
    {                                                                                                                          \
        const uint64_t n_items_ = (uint64_t)(G.surv ? *G.n_live : k) * (uint64_t)(per);                                        \
        const uint64_t stride_ = (uint64_t)(nb) * blockDim.x;                                                                  \
        for (uint64_t t_ = (uint64_t)inc_block_of(blockIdx.x - (b0), (nb)) * blockDim.x + threadIdx.x; t_ < n_items_; t_ += stride_) { \
            const uint32_t t = (uint32_t)t_;                                                                                   \
            __VA_ARGS__;                                                                                                       \
        }                                                                                                                      \
    }

__device__ __forceinline__ bool pt_alive(const float4& p) { return p.x < __uint_as_float(0x7F800000u) && p.x > -__uint_as_float(0x7F800000u); }
__device__ __forceinline__ float pos_inf() { return __uint_as_float(0x7F800000u); }

__device__ __forceinline__ uint64_t entry_key(const uint4& e) { return (uint64_t)e.x | ((uint64_t)e.y << 32); }

// slot of `key`, or ID_NONE
__device__ __forceinline__ uint32_t table_find(const uint4* table, uint32_t mask, uint32_t shift, uint64_t key) {
    uint32_t slot = hash_cell(key, shift) & mask;
    for (uint32_t probes = 0; probes <= mask; ++probes) {
        const uint64_t ek = entry_key(table[slot]);
        if (ek == key) return slot;
        if (ek == EMPTY_KEY) return ID_NONE;
        slot = (slot + 1) & mask;
    }
    return ID_NONE;
}

// append to a sharded list: the entry's index, or ID_NONE if the shard is full
__device__ __forceinline__ uint32_t list_push(uint32_t* cursors, uint32_t cap, uint32_t shard) {
    const uint32_t seg = cap / LIST_SHARDS, s = shard & (LIST_SHARDS - 1u);
    const uint32_t i = atomicAdd(&cursors[s], 1u);
    return i < seg ? s * seg + i : ID_NONE;
}
// worker w of W visits every listed index once across the workers (W a multiple of 64: worker w serves shard w % 64; any other W —
// the host emulation's small launches — walks every shard)
template <class F>
__device__ __forceinline__ void list_for_each(const uint32_t* cursors, uint32_t cap, uint32_t w, uint32_t W, F f) {
    const uint32_t seg = cap / LIST_SHARDS;
    if (W % LIST_SHARDS == 0u) {
        const uint32_t s = w % LIST_SHARDS, n = cursors[s] < seg ? cursors[s] : seg;
        for (uint32_t i = w / LIST_SHARDS; i < n; i += W / LIST_SHARDS) f(s * seg + i);
    } else {
        for (uint32_t s = 0; s < LIST_SHARDS; ++s) {
            const uint32_t n = cursors[s] < seg ? cursors[s] : seg;
            for (uint32_t i = w; i < n; i += W) f(s * seg + i);
        }
    }
}
__device__ __forceinline__ uint32_t list_count(const uint32_t* cursors, uint32_t cap) {
    const uint32_t seg = cap / LIST_SHARDS;
    uint32_t n = 0;
    for (uint32_t s = 0; s < LIST_SHARDS; ++s) n += cursors[s] < seg ? cursors[s] : seg;
    return n;
}

// The tile group of the level-0 bucket `key` is no longer one contiguous region of up-to-date runs (one of its runs moved, a new
// tile appeared outside the region, a run was dropped with its entries left as they were): level 1 stops streaming it
// (bucket_attempt, lv_match.hip: extent 0 -> the point goes on to the lists).  An insert batch lists the groups it breaks
// (M.broken) and lays them out again, in fresh space, behind its last pass (inc_regroup_*): a group stays broken only until
// then — or, broken by an eviction sweep (no list), until an insert touches it or the map is re-linearised.
// Entry states: extent > 0 intact; extent 0 broken; start == ID_NONE: broken and listed by the batch in flight.
__device__ __forceinline__ void inc_break_group(const MapRW& M, uint64_t key) {
    if (!M.gtable) return;
    uint32_t vx, vy, vz;
    int r;
    tile_group_of((uint32_t)(key & 0x1fffff), (uint32_t)((key >> 21) & 0x1fffff), (uint32_t)((key >> 42) & 0x1fffff), vx, vy, vz, r);
    if (vx >= (1u << 21) || vy >= (1u << 21) || vz >= (1u << 21)) return;
    const uint64_t gkey = pack_cell(vx, vy, vz);
    uint32_t slot = hash_cell(gkey, M.gshift) & M.gmask;
    bool found = false, created = false;
    for (uint32_t probes = 0; probes <= M.gmask && probes < 256u; ++probes) {
        uint64_t ek = entry_key(M.gtable[slot]);
        if (ek == EMPTY_KEY) {
            if (!M.broken) return;   // (a sweep does not create groups)
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&M.gtable[slot]), (unsigned long long)EMPTY_KEY,
                                                     (unsigned long long)gkey);
            if (old == (unsigned long long)EMPTY_KEY) {   // a group of newly mapped space (start == ID_NONE from the table's 0xFF fill: "listed"
                created = true;                            // — by this thread, below)
                const uint32_t n = atomicAdd(&M.cnt->gslots_used, 1u);
                if (n + 1u > M.gslot_limit) atomicExch(&M.cnt->overflow, 1u);
                ek = gkey;
            } else {
                ek = (uint64_t)old;
            }
        }
        if (ek == gkey) { found = true; break; }
        slot = (slot + 1) & M.gmask;
    }
    if (!found) { if (M.broken) atomicExch(&M.cnt->overflow, 1u); return; }
    M.gtable[slot].w = 0u;
    if (!M.broken) return;
    if (created || atomicExch(&M.gtable[slot].z, ID_NONE) != ID_NONE) {   // first breaker of this batch
        const uint32_t i = list_push(M.n_broken, M.broken_cap, slot);
        if (i != ID_NONE) M.broken[i] = slot;
        else atomicExch(&M.cnt->overflow, 1u);
    }
}

// (level, neighbour) of work item w in [0, INC_SLOTS_PER_POINT): the key of the slot that must hold point p.
// Returns false if the neighbour voxel is outside the coordinate range.
__device__ __forceinline__ bool inc_slot_key(const MapRW& M, const float4& p, int w, int& level, uint64_t& key) {
    const int cx = cell_coord(p.x, M.origin[0], M.inv_cell), cy = cell_coord(p.y, M.origin[1], M.inv_cell),
              cz = cell_coord(p.z, M.origin[2], M.inv_cell);
    if (w < 27 * REPL_LEVELS) {
        level = w / 27;
        const int c = w % 27;
        const int dz = c / 9 - 1, dy = (c / 3) % 3 - 1, dx = c % 3 - 1;
        const uint32_t nx = (uint32_t)((cx >> level) + dx), ny = (uint32_t)((cy >> level) + dy), nz = (uint32_t)((cz >> level) + dz);
        if (nx >= (1u << 21) || ny >= (1u << 21) || nz >= (1u << 21)) return false;
        key = pack_cell(nx, ny, nz);
    } else {
        level = CELL_SLOT;
        key = pack_cell((uint32_t)(cx >> CELL_LEVEL), (uint32_t)(cy >> CELL_LEVEL), (uint32_t)(cz >> CELL_LEVEL));
    }
    return true;
}

__device__ __forceinline__ bool inc_point_ok(const MapRW& M, const float4& p) {
    if (!(pt_alive(p) && p.y == p.y && p.z == p.z && fabsf(p.y) < pos_inf() && fabsf(p.z) < pos_inf())) return false;
    const int cx = cell_coord(p.x, M.origin[0], M.inv_cell), cy = cell_coord(p.y, M.origin[1], M.inv_cell),
              cz = cell_coord(p.z, M.origin[2], M.inv_cell);
    const int amax = max(abs(cx - CELL_OFFSET), max(abs(cy - CELL_OFFSET), abs(cz - CELL_OFFSET)));
    return amax < CELL_FAR;
}

// where a point of the bucket around voxel (bx, by, bz) sits SEEN FROM THE POINT: the bucket's voxel is neighbour c of the point's
// own voxel (target c of the insert passes) — the slot of the point's back-position
__device__ __forceinline__ uint32_t backpos_slot(const float4& p, const float* origin, float inv_cell, uint32_t bx, uint32_t by, uint32_t bz) {
    const int dx = (int)bx - cell_coord(p.x, origin[0], inv_cell), dy = (int)by - cell_coord(p.y, origin[1], inv_cell),
              dz = (int)bz - cell_coord(p.z, origin[2], inv_cell);
    return (uint32_t)((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1));
}

// ---- ikd-Tree box rule --------------------------------------------------------------------------------------
__device__ __forceinline__ int inc_box_coord(float v, float len) {
    float f = floorf(v / len);
    f = fminf(fmaxf(f, -1048000.0f), 1048000.0f);
    return (int)f + CELL_OFFSET;
}
// The key of a box interleaves the bits of its three 21-bit coordinates (Morton order).  Any injective key serves the box
// table and the grouping sort; this one makes the SORTED batch spatially coherent, and with it everything that is issued in
// that order: the box rule's dead list and the survivor list the append passes walk (round 4: neighbours in the list share
// their 27-blocks, so the table slots, run tails and id arrays a thread touches are the ones its neighbours just touched —
// the append / tombstone kernels were bound by random 64-byte transactions, not by arithmetic).
__device__ __forceinline__ uint64_t inc_spread21(uint32_t v) {
    uint64_t x = v & 0x1fffffu;
    x = (x | (x << 32)) & 0x001f00000000ffffull;
    x = (x | (x << 16)) & 0x001f0000ff0000ffull;
    x = (x | (x << 8)) & 0x100f00f00f00f00full;
    x = (x | (x << 4)) & 0x10c30c30c30c30c3ull;
    x = (x | (x << 2)) & 0x1249249249249249ull;
    return x;
}
__device__ __forceinline__ uint64_t inc_box_key(const float4& p, float len) {
    return inc_spread21((uint32_t)inc_box_coord(p.x, len)) | (inc_spread21((uint32_t)inc_box_coord(p.y, len)) << 1) |
           (inc_spread21((uint32_t)inc_box_coord(p.z, len)) << 2);
}
// calc_dist(point, centre of its box) [UPSTREAM-RECALL ikd-Tree Add_Points]: f32, centre = min + (max - min) / 2 with the
// division in double as upstream writes it
__device__ __forceinline__ float inc_box_center_dist(const float4& p, float len) {
    const float c[3] = {p.x, p.y, p.z};
    float mid[3];
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const float vmin = floorf(c[a] / len) * len;
        const float vmax = vmin + len;
        mid[a] = (float)((double)vmin + (double)(vmax - vmin) / 2.0);
    }
    const float dx = p.x - mid[0], dy = p.y - mid[1], dz = p.z - mid[2];
    const float sx = dx * dx, sy = dy * dy, sz = dz * dz;
    const float s = sx + sy;
    return s + sz;
}

// the slot of a box, inserting the key if absent (head stays ID_NONE from the table's 0xFF fill)
__device__ __forceinline__ uint32_t box_find_or_create(const BoxRW& B, uint64_t key, MapCounters* cnt) {
    uint32_t slot = hash_cell(key, B.shift) & B.mask;
    for (uint32_t probes = 0; probes <= B.mask; ++probes) {
        unsigned long long* kp = reinterpret_cast<unsigned long long*>(&B.table[slot]);
        const unsigned long long old = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
        if (old == (unsigned long long)EMPTY_KEY) {
            const uint32_t n = atomicAdd(&cnt->box_slots_used, 1u);
            if (n + 1u > B.slot_limit) atomicExch(&cnt->overflow, 1u);
            return slot;
        }
        if (old == (unsigned long long)key) return slot;
        slot = (slot + 1) & B.mask;
    }
    atomicExch(&cnt->overflow, 1u);
    return ID_NONE;
}

// every living id -> its box chain (first down-sampling insert after a (re)build)
// (block_base: the launch may be one slice of the grid — MapStore::launch_sliced)
__device__ __forceinline__ void box_build_item(const BoxRW& B, const float4* __restrict__ orig, uint32_t n_ids, MapCounters* cnt, uint32_t id) {
    if (id >= n_ids) return;
    const float4 p = orig[id];
    if (!pt_alive(p)) { B.next[id] = ID_NONE; return; }
    const uint32_t slot = box_find_or_create(B, inc_box_key(p, B.len), cnt);
    if (slot == ID_NONE) return;
    B.next[id] = atomicExch(&B.table[slot].z, id);
}
__global__ void box_build_kernel(BoxRW B, const float4* __restrict__ orig, uint32_t n_ids, MapCounters* cnt, uint32_t block_base = 0) {
    box_build_item(B, orig, n_ids, cnt, (blockIdx.x + block_base) * blockDim.x + threadIdx.x);
}

// keys of the new points for the stable sort by box; points that cannot be inserted sort last
// (check_range == 0: the map is empty and has no origin yet — only finiteness is checked, the build that follows
// validates the extent)
// (every one-thread-per-item kernel below is a thin wrapper around its *_item function: small batches run several of them
// back to back inside ONE workgroup — inc_small_front_kernel, lv_map.hip — instead of one launch each)
__device__ __forceinline__ void inc_box_keys_item(const MapRW& M, const float4* __restrict__ newp, uint32_t k, float len,
                                                  uint64_t* __restrict__ keys, uint32_t* __restrict__ idx, uint32_t* __restrict__ alive,
                                                  int downsample, int check_range, uint32_t j) {
    if (j >= k) return;
    const float4 p = newp[j];
    idx[j] = j;
    const bool finite = pt_alive(p) && fabsf(p.y) < pos_inf() && fabsf(p.z) < pos_inf();
    if (!(check_range ? inc_point_ok(M, p) : finite)) {
        keys[j] = ~0ull >> 1;
        alive[j] = 0u;
        atomicAdd(&M.cnt->dropped, 1u);
        return;
    }
    keys[j] = inc_box_key(p, len);
    alive[j] = downsample ? 0u : 1u;   // without down-sampling every insertable point lives
}
__global__ void inc_box_keys_kernel(MapRW M, const float4* __restrict__ newp, uint32_t k, float len, uint64_t* __restrict__ keys,
                                    uint32_t* __restrict__ idx, uint32_t* __restrict__ alive, int downsample, int check_range) {
    inc_box_keys_item(M, newp, k, len, keys, idx, alive, downsample, check_range, blockIdx.x * blockDim.x + threadIdx.x);
}

// One thread per box touched by the batch replays upstream's sequential rule: for every new point p of the box, in
// input order, the point nearest to the box centre among {p} U (points currently in the box) survives if the box
// held more than one point or p itself is that nearest point (occupants must be STRICTLY closer to beat p; among
// equally near occupants the oldest wins); otherwise the box is left alone.  Old occupants that lose are put on
// the dead list (coordinates + id) and their slot in `orig` is marked; `alive` flags the new points that live.
__device__ __forceinline__ void inc_box_rule_item(const BoxRW& B, float4* __restrict__ orig, const float4* __restrict__ newp,
                                                  const uint64_t* __restrict__ keys_sorted, const uint32_t* __restrict__ idx_sorted,
                                                  uint32_t k, uint32_t* __restrict__ alive, float4* __restrict__ dead, uint32_t dead_cap,
                                                  MapCounters* cnt, uint32_t i) {
    if (i >= k) return;
    const uint64_t key = keys_sorted[i];
    if (key == (~0ull >> 1)) return;                       // dropped points
    if (i > 0 && keys_sorted[i - 1] == key) return;        // not the head of its group
    uint32_t end = i + 1;
    while (end < k && keys_sorted[end] == key) ++end;
    const uint32_t bslot = table_find(B.table, B.mask, B.shift, key);
    const uint32_t head = bslot == ID_NONE ? ID_NONE : B.table[bslot].z;
    // living occupants of the box
    uint32_t nE = 0, only = ID_NONE;
    for (uint32_t e = head; e != ID_NONE; e = B.next[e])
        if (pt_alive(orig[e])) { ++nE; only = e; }
    bool multi = nE > 1;
    bool cur_new = false;                                   // survivor so far: a new point (index) or an old id
    uint32_t cur = nE == 1 ? only : ID_NONE;
    float dcur = nE == 1 ? inc_box_center_dist(orig[only], B.len) : 0.f;
    bool changed = false;
    for (uint32_t g = i; g < end; ++g) {
        const uint32_t j = idx_sorted[g];
        const float4 p = newp[j];
        float md = inc_box_center_dist(p, B.len);
        bool best_new = true;
        uint32_t best = j;
        if (multi) {
            for (uint32_t e = head; e != ID_NONE; e = B.next[e]) {
                const float4 q = orig[e];
                if (!pt_alive(q)) continue;
                const float d = inc_box_center_dist(q, B.len);
                if (d < md || (d == md && !best_new && e < best)) { md = d; best = e; best_new = false; }
            }
        } else if (cur != ID_NONE) {
            if (dcur < md) { md = dcur; best = cur; best_new = cur_new; }
        }
        if (multi || (best_new && best == j)) {
            // everything in the box but `best` goes
            if (multi) {
                for (uint32_t e = head; e != ID_NONE; e = B.next[e]) {
                    const float4 q = orig[e];
                    if (!pt_alive(q) || (!best_new && e == best)) continue;
                    const uint32_t di = atomicAdd(&cnt->n_dead, 1u);
                    if (di < dead_cap) dead[di] = make_float4(q.x, q.y, q.z, __uint_as_float(e));
                    else atomicExch(&cnt->overflow, 1u);
                    orig[e].x = pos_inf();
                }
            } else if (cur != ID_NONE) {
                if (cur_new) {
                    alive[cur] = 0u;
                } else {
                    const float4 q = orig[cur];
                    const uint32_t di = atomicAdd(&cnt->n_dead, 1u);
                    if (di < dead_cap) dead[di] = make_float4(q.x, q.y, q.z, __uint_as_float(cur));
                    else atomicExch(&cnt->overflow, 1u);
                    orig[cur].x = pos_inf();
                }
            }
            if (best_new) alive[best] = 1u;
            cur = best;
            cur_new = best_new;
            dcur = md;
            multi = false;
            changed = true;
        }
    }
    if (changed && bslot != ID_NONE) {
        // the box now holds exactly one point: an old id (its chain shrinks to itself) or a new point (linked by
        // inc_commit_points_kernel once its id is known)
        if (cur_new) {
            B.table[bslot].z = ID_NONE;
        } else {
            B.table[bslot].z = cur;
            B.next[cur] = ID_NONE;
        }
    }
}
__global__ void inc_box_rule_kernel(BoxRW B, float4* __restrict__ orig, const float4* __restrict__ newp,
                                    const uint64_t* __restrict__ keys_sorted, const uint32_t* __restrict__ idx_sorted, uint32_t k,
                                    uint32_t* __restrict__ alive, float4* __restrict__ dead, uint32_t dead_cap, MapCounters* cnt) {
    inc_box_rule_item(B, orig, newp, keys_sorted, idx_sorted, k, alive, dead, dead_cap, cnt, blockIdx.x * blockDim.x + threadIdx.x);
}

// surviving new points get their ids (id_base + rank among the survivors, input order), their slot in `orig` and,
// if the box table exists, their place at the head of their box chain
__device__ __forceinline__ void inc_commit_points_item(const MapRW& M, const BoxRW& B, int have_boxes, const float4* __restrict__ newp,
                                                       const uint32_t* __restrict__ alive, const uint32_t* __restrict__ apos, uint32_t k,
                                                       uint32_t id_base, uint32_t j) {
    if (j >= k) return;
    if (j == k - 1) M.cnt->n_new = apos[j] + alive[j];
    if (!alive[j]) return;
    const uint32_t id = id_base + apos[j];
    const float4 p = newp[j];
    M.orig[id] = make_float4(p.x, p.y, p.z, 0.f);
    if (have_boxes) {
        const uint32_t slot = box_find_or_create(B, inc_box_key(p, B.len), M.cnt);
        if (slot == ID_NONE) return;
        B.next[id] = atomicExch(&B.table[slot].z, id);
    }
}
__global__ void inc_commit_points_kernel(MapRW M, BoxRW B, int have_boxes, const float4* __restrict__ newp,
                                         const uint32_t* __restrict__ alive, const uint32_t* __restrict__ apos, uint32_t k,
                                         uint32_t id_base) {
    inc_commit_points_item(M, B, have_boxes, newp, alive, apos, k, id_base, blockIdx.x * blockDim.x + threadIdx.x);
}

// ---- tombstones ------------------------------------------------------------------------------------------------
// dead[j] = {x, y, z, id} of a point that left the map: one thread per (point, slot that holds it)
__device__ __forceinline__ void inc_kill_slot(const MapRW& M, const float4* __restrict__ dead, uint32_t j, int w);
__global__ void inc_kill_kernel(MapRW M, const float4* __restrict__ dead, uint32_t n_dead) {
    const uint32_t t = inc_thread_id();
    const uint32_t j = t / (uint32_t)INC_SLOTS_PER_POINT;
    const int w = (int)(t % (uint32_t)INC_SLOTS_PER_POINT);
    if (j >= n_dead) return;
    inc_kill_slot(M, dead, j, w);
}
// the same with the length of the list read on the device (small batches: no host round trip for it), grid-stride
__device__ __forceinline__ void inc_kill_counted_item(const MapRW& M, const float4* __restrict__ dead, uint32_t dead_cap, uint32_t t0, uint32_t n_threads) {
    const uint32_t n_dead = M.cnt->n_dead < dead_cap ? M.cnt->n_dead : dead_cap;
    const uint64_t total = (uint64_t)n_dead * (uint32_t)INC_SLOTS_PER_POINT;
    for (uint64_t t = t0; t < total; t += (uint64_t)n_threads)
        inc_kill_slot(M, dead, (uint32_t)(t / (uint32_t)INC_SLOTS_PER_POINT), (int)(t % (uint32_t)INC_SLOTS_PER_POINT));
}
__global__ void inc_kill_counted_kernel(MapRW M, const float4* __restrict__ dead, uint32_t dead_cap) {
    inc_kill_counted_item(M, dead, dead_cap, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
__device__ __forceinline__ void inc_kill_slot(const MapRW& M, const float4* __restrict__ dead, uint32_t j, int w) {
    const float4 p = dead[j];
    const uint32_t id = __float_as_uint(p.w);
    int level;
    uint64_t key;
    if (!inc_slot_key(M, p, w, level, key)) return;
    const LevelRW& L = M.lv[level];
    const uint32_t slot = table_find(L.table, L.mask, L.shift, key);
    if (slot == ID_NONE) return;
    const uint4 e = L.table[slot];
    if (level < SORTED_LEVELS) {
        // the point knows where it sits (the 27 back-positions of an id are 54 contiguous bytes: one line per deleted point);
        // only a position beyond 16 bits is searched for
        uint32_t pos = M.backpos[(size_t)id * 27 + (uint32_t)(w % 27)];
        if (pos == (uint32_t)BACKPOS_FAR) {
            const uint32_t* ids = M.bidx[level] + e.z;
            uint32_t lo = (uint32_t)BACKPOS_FAR < e.w ? (uint32_t)BACKPOS_FAR : e.w, hi = e.w;   // first position with ids[pos] >= id (ids ascend, tombstones keep theirs)
            while (lo < hi) {
                const uint32_t mid = lo + ((hi - lo) >> 1);
                if (ids[mid] < id) lo = mid + 1; else hi = mid;
            }
            pos = (lo < e.w && ids[lo] == id) ? lo : e.w;
        }
        if (pos < e.w) {
            M.bxyz[level][((size_t)e.z + pos) * 3] = pos_inf();
            atomicAdd(&L.aux[slot].dead, 1u);   // (a point dies once: every tombstone is counted once)
        }
    } else {
        const uint32_t pos = M.cellpos[id];
        if (pos < e.w && __float_as_uint(M.cell4[(size_t)e.z + pos].w) == id) M.cell4[(size_t)e.z + pos].x = pos_inf();
    }
}

// ---- appends ---------------------------------------------------------------------------------------------------
// The new points of a batch are grouped by their voxel on every level (a scratch hash table per level, no sort):
// all points of a voxel go to the same 27 buckets, so ONE thread per (voxel, neighbour) finds / creates the bucket
// slot and reserves the voxel's share of the bucket's tail with a single atomicAdd (point-by-point registration
// cost 82 contended atomics per point: 4.3 ms of a 5.9 ms insert).  A point then knows its place without any
// atomic: tail start + its voxel's offset in that bucket + its rank inside its voxel group.
struct GroupRW {
    uint4* table[REPL_LEVELS];     // scratch tables (0xFF-filled): {key lo, key hi, count - 1, -}; a group IS its slot
    uint32_t mask, shift, size;    // same geometry on all levels
    uint32_t* prank;               // [j * REPL_LEVELS + l]: rank of point j inside its level-l voxel group
    uint32_t* pslot;               // [j * REPL_LEVELS + l]: scratch-table slot of that group
    uint32_t* gbase[REPL_LEVELS];  // [slot * GROUP_TARGETS + c]: offset of the group's points in the batch tail of target c
    uint32_t* gslot[REPL_LEVELS];  // [slot * GROUP_TARGETS + c]: table slot of target c (ID_NONE: outside the range)
    uint4* gdst[REPL_LEVELS];      // [slot * GROUP_TARGETS + c]: {absolute pool index of target c's batch tail, entries the batch appends
                                   // to it, entries the target held before the batch, -} (x = ID_NONE: no such target), written once per
                                   // (group, target) after the room is made (inc_resolve): the per-point passes then need neither the
                                   // table nor the aux record of the target
    const uint32_t* surv;          // optional: the batch's survivors (indices into the batch) in the order the passes walk them —
                                   // the box sort's, i.e. Morton — and their number n_live (device word); nullptr: every point of
                                   // the batch is visited in input order and the dead leave at once
    const uint32_t* n_live;
};
// work item s of a per-point pass -> the new point j it stands for (false: nothing to do)
__device__ __forceinline__ bool inc_item_point(const GroupRW& G, const uint32_t* __restrict__ alive, uint32_t k, uint32_t s, uint32_t& j) {
    if (G.surv) {
        if (s >= *G.n_live) return false;
        j = G.surv[s];
        return true;
    }
    if (s >= k || !alive[s]) return false;
    j = s;
    return true;
}
// the survivors in sorted-batch order: flag of sorted position i ...
__device__ __forceinline__ void inc_surv_flag_item(const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ alive, uint32_t k,
                                                   uint32_t* __restrict__ flag, uint32_t i) {
    if (i < k) flag[i] = alive[idx_sorted[i]];
}
__global__ void inc_surv_flag_kernel(const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ alive, uint32_t k, uint32_t* __restrict__ flag) {
    inc_surv_flag_item(idx_sorted, alive, k, flag, blockIdx.x * blockDim.x + threadIdx.x);
}
// ... and, after an exclusive scan of the flags, the list itself
__device__ __forceinline__ void inc_surv_list_item(const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ flag,
                                                   const uint32_t* __restrict__ fpos, uint32_t k, uint32_t* __restrict__ surv, uint32_t i) {
    if (i < k && flag[i]) surv[fpos[i]] = idx_sorted[i];
}
__global__ void inc_surv_list_kernel(const uint32_t* __restrict__ idx_sorted, const uint32_t* __restrict__ flag, const uint32_t* __restrict__ fpos,
                                     uint32_t k, uint32_t* __restrict__ surv) {
    inc_surv_list_item(idx_sorted, flag, fpos, k, surv, blockIdx.x * blockDim.x + threadIdx.x);
}
constexpr int GROUP_TARGETS = 28;   // 27 neighbour buckets + the list of the level-2 voxel the group's voxel lies in
// (No global work lists or group counters: an atomic whose result is needed costs ~10 ns when every thread of a
// launch hits the same address — 400 000 list appends were 4 ms of a 6 ms insert.  The group that reserves the
// FIRST share of a target's tail (offset 0) owns that target for the rest of the batch: it makes room and commits.)

// pass 1: every surviving new point joins its voxel group on each level
__device__ __forceinline__ void inc_group_item(const MapRW& M, const GroupRW& G, const float4* __restrict__ newp,
                                               const uint32_t* __restrict__ alive, uint32_t k, uint32_t t) {
    uint32_t j;
    const int l = (int)(t % (uint32_t)REPL_LEVELS);
    if (!inc_item_point(G, alive, k, t / (uint32_t)REPL_LEVELS, j)) return;
    const float4 p = newp[j];
    const int cx = cell_coord(p.x, M.origin[0], M.inv_cell) >> l, cy = cell_coord(p.y, M.origin[1], M.inv_cell) >> l,
              cz = cell_coord(p.z, M.origin[2], M.inv_cell) >> l;
    const uint64_t key = pack_cell((uint32_t)cx, (uint32_t)cy, (uint32_t)cz);
    uint4* tab = G.table[l];
    uint32_t slot = hash_cell(key, G.shift) & G.mask;
    for (uint32_t probes = 0; probes <= G.mask; ++probes) {
        uint64_t ek = entry_key(tab[slot]);
        if (ek == EMPTY_KEY) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&tab[slot]), (unsigned long long)EMPTY_KEY,
                                                     (unsigned long long)key);
            ek = old == (unsigned long long)EMPTY_KEY ? key : (uint64_t)old;
        }
        if (ek == key) {
            G.prank[(size_t)j * REPL_LEVELS + l] = atomicAdd(&tab[slot].z, 1u) + 1u;   // z starts at 0xFFFFFFFF: first rank 0
            G.pslot[(size_t)j * REPL_LEVELS + l] = slot;
            return;
        }
        slot = (slot + 1) & G.mask;
    }
    atomicExch(&M.cnt->overflow, 1u);
}
__global__ void inc_group_kernel(MapRW M, GroupRW G, const float4* __restrict__ newp, const uint32_t* __restrict__ alive, uint32_t k) {
    inc_group_item(M, G, newp, alive, k, inc_thread_id());
}

// slot of `key` in a bucket / list table, inserting it if absent (plain probe first: most slots exist)
constexpr uint32_t MAX_PROBES = 256;   // a longer probe sequence means the table is too full: re-linearise
__device__ __forceinline__ uint32_t table_get_slot(const LevelRW& L, uint64_t key, uint32_t* slots_used, uint32_t* overflow, bool* created = nullptr) {
    uint32_t slot = hash_cell(key, L.shift) & L.mask;
    for (uint32_t probes = 0; probes <= L.mask && probes < MAX_PROBES; ++probes) {
        uint64_t ek = entry_key(L.table[slot]);
        if (ek == EMPTY_KEY) {
            const unsigned long long old = atomicCAS(reinterpret_cast<unsigned long long*>(&L.table[slot]), (unsigned long long)EMPTY_KEY,
                                                     (unsigned long long)key);
            if (old == (unsigned long long)EMPTY_KEY) {
                L.table[slot].z = 0u;
                L.table[slot].w = 0u;
                atomicAdd(slots_used, 1u);   // statistics only (result unused)
                if (created) *created = true;
                return slot;
            }
            ek = (uint64_t)old;
        }
        if (ek == key) return slot;
        slot = (slot + 1) & L.mask;
    }
    atomicExch(overflow, 1u);
    return ID_NONE;
}

// The passes that work per (voxel group, target) run one thread per (new point, level, target); only the group's
// first point (rank 0) acts, the others leave at once.
__device__ __forceinline__ bool inc_group_leader(const GroupRW& G, const uint32_t* __restrict__ alive, uint32_t k, uint32_t t, int& l,
                                                 int& c, uint32_t& gs) {
    uint32_t j;
    const uint32_t r = t % (uint32_t)(REPL_LEVELS * GROUP_TARGETS);
    l = (int)(r / (uint32_t)GROUP_TARGETS);
    c = (int)(r % (uint32_t)GROUP_TARGETS);
    if (!inc_item_point(G, alive, k, t / (uint32_t)(REPL_LEVELS * GROUP_TARGETS), j)) return false;
    if (G.prank[(size_t)j * REPL_LEVELS + l] != 0u) return false;
    gs = G.pslot[(size_t)j * REPL_LEVELS + l];
    return true;
}

// pass 2: per (voxel group, target): find / create the target's slot and take the group's share of its batch tail
__device__ __forceinline__ void inc_register_item(const MapRW& M, const GroupRW& G, const uint32_t* __restrict__ alive, uint32_t k, uint32_t t) {
    int l, c;
    uint32_t gs;
    if (!inc_group_leader(G, alive, k, t, l, c, gs)) return;
    const size_t r = (size_t)gs * GROUP_TARGETS + (size_t)c;
    G.gslot[l][r] = ID_NONE;
    G.gbase[l][r] = 0u;
    const uint4 ge = G.table[l][gs];
    const uint64_t vkey = entry_key(ge);
    const uint32_t n_v = ge.z + 1u;
    const uint32_t vx = (uint32_t)(vkey & 0x1fffff), vy = (uint32_t)((vkey >> 21) & 0x1fffff), vz = (uint32_t)((vkey >> 42) & 0x1fffff);
    int tl;          // index of the target table
    uint64_t key;
    if (c < 27) {
        const int dz = c / 9 - 1, dy = (c / 3) % 3 - 1, dx = c % 3 - 1;
        const uint32_t nx = vx + (uint32_t)dx, ny = vy + (uint32_t)dy, nz = vz + (uint32_t)dz;
        if (nx >= (1u << 21) || ny >= (1u << 21) || nz >= (1u << 21)) return;
        tl = l;
        key = pack_cell(nx, ny, nz);
    } else {
        if (l != REPL_LEVELS - 1) return;
        tl = CELL_SLOT;   // (several voxel groups may share a list: they take their shares of its tail like the groups of a bucket)
        key = pack_cell(vx >> (CELL_LEVEL - l), vy >> (CELL_LEVEL - l), vz >> (CELL_LEVEL - l));
    }
    const LevelRW& L = M.lv[tl];
    bool created = false;
    const uint32_t slot = table_get_slot(L, key, &M.cnt->slots_used[tl], &M.cnt->overflow, &created);
    if (slot == ID_NONE) return;
    if (created && tl < REPL_LEVELS) inc_break_group(M, key);   // a new bucket lies outside its group's region
    G.gbase[l][r] = atomicAdd(&L.aux[slot].pending, n_v);
    G.gslot[l][r] = slot;
}
__global__ void inc_register_kernel(MapRW M, GroupRW G, const uint32_t* __restrict__ alive, uint32_t k) {
    LV_INC_ITEMS(REPL_LEVELS * GROUP_TARGETS, 0u, gridDim.x, inc_register_item(M, G, alive, k, t));
}

// pass 3: the group that took offset 0 of a target's tail makes room for the whole batch: a run that cannot take
// its pending entries gets fresh space at the end of the pool (1.5x the new size); the move itself is listed for
// inc_relocate_kernel
__device__ __forceinline__ void inc_reserve_item(const MapRW& M, const GroupRW& G, const uint32_t* __restrict__ alive, uint32_t k, uint4* __restrict__ reloc,
                                   uint32_t reloc_cap, uint32_t* __restrict__ n_reloc, uint32_t t) {
    int l, c;
    uint32_t gs;
    if (!inc_group_leader(G, alive, k, t, l, c, gs)) return;
    const size_t r = (size_t)gs * GROUP_TARGETS + (size_t)c;
    const uint32_t slot = G.gslot[l][r];
    if (slot == ID_NONE || G.gbase[l][r] != 0u) return;
    const int level = c < 27 ? l : CELL_SLOT;
    const LevelRW& L = M.lv[level];
    const uint4 e = L.table[slot];
    const SlotAux a = L.aux[slot];
    const uint32_t need = e.w + a.pending;
    uint32_t tail0 = e.w;
    bool compacted = false;
    if (need > a.cap && level < SORTED_LEVELS && M.comp && e.w >= 8u) {
        // A bucket over re-observed ground grows by its DELETED entries: every scan's points compete with the occupants of their
        // 0.2 m boxes, the loser stays behind as a tombstone.  If the living entries and the batch fit the run's room, the run is
        // compacted where it lies (inc_compact_gather / _scatter: order kept, back-positions rewritten) instead of moving — a
        // move takes the run's whole tile group along to fresh space (inc_regroup_*) and the pool with it: five 64k-point scans
        // over the same ground used up the pool of a 1 M-point map, a re-linearisation every sixth scan.
        const uint32_t living = e.w - (a.dead < e.w ? a.dead : e.w);   // (the run's tombstones are counted as they are written: inc_kill_slot, the eviction sweep)
        if (living + a.pending <= a.cap && living + 4u <= e.w) {
            const uint32_t sseg = M.cstage_cap / LIST_SHARDS, sh = slot & (LIST_SHARDS - 1u);
            const uint32_t sb = atomicAdd(&M.n_comp[LIST_SHARDS + sh], e.w);   // (the staging area is sharded like the list)
            if ((uint64_t)sb + e.w <= (uint64_t)sseg) {
                const uint32_t ci = list_push(M.n_comp, M.comp_cap, slot);
                if (ci != ID_NONE) {
                    M.comp[ci] = uint4{slot, living, e.w, sh * sseg + sb};
                    tail0 = living;
                    compacted = true;
                }
            }
        }
    }
    if (need > a.cap && !compacted) {
        const uint32_t extra = need / 2u > 8u ? need / 2u : 8u;
        const uint32_t ncap = need + extra;
        uint32_t ns = 0;
        bool got = false;
        for (int tr = 0; tr < 4 && !got; ++tr) {   // (a failed try leaves its arena exhausted: the others still serve)
            const uint32_t ar = (slot + (uint32_t)tr * 17u) & (uint32_t)(N_ARENAS - 1);
            ns = atomicAdd(&M.cnt->arena_cur[level][ar], ncap);
            got = (uint64_t)ns + ncap <= (uint64_t)M.cnt->arena_end[level][ar];
        }
        if (!got) {
            atomicExch(&M.cnt->overflow, 1u);
        } else {
            if (e.w) {
                const uint32_t ri = list_push(n_reloc, reloc_cap, slot);
                if (ri != ID_NONE) reloc[ri] = uint4{(uint32_t)level, e.z, ns, e.w};
                else atomicExch(&M.cnt->overflow, 1u);
            }
            L.table[slot].z = ns;
            L.aux[slot].cap = ncap;
            if (level < REPL_LEVELS) inc_break_group(M, entry_key(e));   // the run leaves its group's region
        }
    }
    L.aux[slot].tail0 = tail0;
}
__global__ void inc_reserve_kernel(MapRW M, GroupRW G, const uint32_t* __restrict__ alive, uint32_t k, uint4* __restrict__ reloc,
                                   uint32_t reloc_cap, uint32_t* __restrict__ n_reloc) {
    LV_INC_ITEMS(REPL_LEVELS * GROUP_TARGETS, 0u, gridDim.x, inc_reserve_item(M, G, alive, k, reloc, reloc_cap, n_reloc, t));
}

// pass 3a: every (voxel group, target) notes where the target's batch tail starts in the pool and how long it is — table slot and
// aux record are read ONCE per (group, target) here instead of once per (point, target) in each of the three passes below
// (round 4: those passes are bound by the requests they issue)
__device__ __forceinline__ void inc_resolve_item(const MapRW& M, const GroupRW& G, const uint32_t* __restrict__ alive, uint32_t k, uint32_t t) {
    int l, c;
    uint32_t gs;
    if (!inc_group_leader(G, alive, k, t, l, c, gs)) return;
    const size_t r = (size_t)gs * GROUP_TARGETS + (size_t)c;
    const uint32_t slot = G.gslot[l][r];
    if (slot == ID_NONE) { G.gdst[l][r] = uint4{ID_NONE, 0u, 0u, 0u}; return; }
    const LevelRW& L = M.lv[c < 27 ? l : CELL_SLOT];
    const SlotAux a = L.aux[slot];
    G.gdst[l][r] = uint4{L.table[slot].z + a.tail0, a.pending, a.tail0, 0u};
}
__global__ void inc_resolve_kernel(MapRW M, GroupRW G, const uint32_t* __restrict__ alive, uint32_t k) {
    LV_INC_ITEMS(REPL_LEVELS * GROUP_TARGETS, 0u, gridDim.x, inc_resolve_item(M, G, alive, k, t));
}

// pass 3b: the listed runs move, one wavefront per run — a whole workgroup per run for small batches (LANES = 256: few runs, and
// the longest one is the launch's duration).  (Round 4: four independent copies in flight per lane instead of 32 lanes with one —
// a level-2 run of ~1000 entries was 31 dependent trips to memory, ~30 us of a small batch's insert; source and destination never
// overlap: the new place is a fresh allocation at the end of the pool.)
constexpr int RELOC_LANES = 64, RELOC_LANES_SMALL = 256, RELOC_UNROLL = 4;
template <int LANES, typename T>
__device__ __forceinline__ void inc_copy_run(T* __restrict__ dst, const T* __restrict__ src, uint32_t n, uint32_t lane) {
    for (uint32_t base = 0; base < n; base += (uint32_t)(LANES * RELOC_UNROLL)) {
        T v[RELOC_UNROLL];
#pragma unroll
        for (int u = 0; u < RELOC_UNROLL; ++u) {
            const uint32_t i = base + (uint32_t)(u * LANES) + lane;
            if (i < n) v[u] = src[i];
        }
#pragma unroll
        for (int u = 0; u < RELOC_UNROLL; ++u) {
            const uint32_t i = base + (uint32_t)(u * LANES) + lane;
            if (i < n) dst[i] = v[u];
        }
    }
}
template <int LANES = RELOC_LANES>
__device__ __forceinline__ void inc_relocate_item(const MapRW& M, const uint4* __restrict__ reloc, uint32_t reloc_cap, const uint32_t* __restrict__ n_reloc,
                                                  uint32_t t, uint32_t n_threads) {
    if (M.cnt->overflow) return;
    const uint32_t lane = t % (uint32_t)LANES;
    // (grid-stride over the listed runs: how many there are is only known here, and a launch sized for the list's capacity
    // spent 23 us of a small batch on workgroups that had nothing to move)
    list_for_each(n_reloc, reloc_cap, t / (uint32_t)LANES, n_threads / (uint32_t)LANES, [&](uint32_t r) {
        const uint4 m = reloc[r];   // {level, old start, new start, count}
        if ((int)m.x < SORTED_LEVELS) {
            float* xs = M.bxyz[m.x];
            uint32_t* is = M.bidx[m.x];
            inc_copy_run<LANES>(xs + (size_t)m.z * 3, xs + (size_t)m.y * 3, m.w * 3u, lane);   // (12-byte points as plain words: coalesced)
            inc_copy_run<LANES>(is + (size_t)m.z, is + (size_t)m.y, m.w, lane);
        } else {
            inc_copy_run<LANES>(M.cell4 + (size_t)m.z, M.cell4 + (size_t)m.y, m.w, lane);
        }
    });
}
__global__ void inc_relocate_kernel(MapRW M, const uint4* __restrict__ reloc, uint32_t reloc_cap, const uint32_t* __restrict__ n_reloc) {
    inc_relocate_item(M, reloc, reloc_cap, n_reloc, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}

// where new point j goes in target w: the level's table index, the absolute pool index of the target's batch tail, the tail's
// length and the point's offset inside it (arbitrary order: its group's share + its rank in the group); c = target of its level
__device__ __forceinline__ bool inc_dst_of(const GroupRW& G, uint32_t j, int w, int& tl, int& l, int& c, size_t& r, uint4& dst, uint32_t& off) {
    if (w < 27 * REPL_LEVELS) { l = w / 27; c = w % 27; tl = l; }
    else { l = REPL_LEVELS - 1; c = 27; tl = CELL_SLOT; }
    r = (size_t)G.pslot[(size_t)j * REPL_LEVELS + l] * GROUP_TARGETS + (size_t)c;
    dst = G.gdst[l][r];
    if (dst.x == ID_NONE) return false;
    off = G.gbase[l][r] + G.prank[(size_t)j * REPL_LEVELS + l];
    return true;
}

// pass 4: ids into the tails of the buckets (arbitrary order inside a tail); the unordered runs (voxel lists) take the whole
// record at once
__device__ __forceinline__ void inc_fill_item(const MapRW& M, const GroupRW& G, const float4* __restrict__ newp, const uint32_t* __restrict__ alive,
                                const uint32_t* __restrict__ apos, uint32_t k, uint32_t id_base, uint32_t t) {
    if (M.cnt->overflow) return;
    uint32_t j;
    const int w = (int)(t % (uint32_t)INC_SLOTS_PER_POINT);
    if (!inc_item_point(G, alive, k, t / (uint32_t)INC_SLOTS_PER_POINT, j)) return;
    int tl, l, c;
    size_t r;
    uint4 dst;
    uint32_t off;
    if (!inc_dst_of(G, j, w, tl, l, c, r, dst, off)) return;
    const uint32_t id = id_base + apos[j];
    const size_t at = (size_t)dst.x + off;
    if (tl < SORTED_LEVELS) {
        M.bidx[tl][at] = id;
    } else {
        // the lists remember positions RELATIVE to the run's start (a run may move): count before the batch + place in its tail
        const float4 p = newp[j];
        M.cell4[at] = make_float4(p.x, p.y, p.z, __uint_as_float(id));
        M.cellpos[id] = dst.z + off;
    }
}
__global__ void inc_fill_kernel(MapRW M, GroupRW G, const float4* __restrict__ newp, const uint32_t* __restrict__ alive,
                                const uint32_t* __restrict__ apos, uint32_t k, uint32_t id_base) {
    LV_INC_ITEMS(INC_SLOTS_PER_POINT, 0u, gridDim.x, inc_fill_item(M, G, newp, alive, apos, k, id_base, t));
}

// pass 5: rank of every new bucket entry among the ids of its tail (levels 0, 1)
__device__ __forceinline__ void inc_rank_item(const MapRW& M, const GroupRW& G, const uint32_t* __restrict__ alive, const uint32_t* __restrict__ apos, uint32_t k,
                                uint32_t id_base, uint32_t* __restrict__ rank, uint32_t t) {
    if (M.cnt->overflow) return;
    uint32_t j;
    const int w = (int)(t % (uint32_t)(27 * SORTED_LEVELS));
    if (!inc_item_point(G, alive, k, t / (uint32_t)(27 * SORTED_LEVELS), j)) return;
    int tl, l, c;
    size_t rr;
    uint4 dst;
    uint32_t off;
    if (!inc_dst_of(G, j, w, tl, l, c, rr, dst, off)) return;
    const uint32_t* ids = M.bidx[tl] + (size_t)dst.x;
    const uint32_t id = id_base + apos[j];
    uint32_t r = 0;
    for (uint32_t i = 0; i < dst.y; ++i) r += ids[i] < id ? 1u : 0u;
    rank[t] = r;
}
__global__ void inc_rank_kernel(MapRW M, GroupRW G, const uint32_t* __restrict__ alive, const uint32_t* __restrict__ apos, uint32_t k,
                                uint32_t id_base, uint32_t* __restrict__ rank) {
    LV_INC_ITEMS(27 * SORTED_LEVELS, 0u, gridDim.x, inc_rank_item(M, G, alive, apos, k, id_base, rank, t));
}

// pass 6: every new bucket entry goes to its ranked place (all reads of pass 5 are done: kernel boundary)
__device__ __forceinline__ void inc_place_item(const MapRW& M, const GroupRW& G, const float4* __restrict__ newp, const uint32_t* __restrict__ alive,
                                 const uint32_t* __restrict__ apos, uint32_t k, uint32_t id_base, const uint32_t* __restrict__ rank, uint32_t t) {
    if (M.cnt->overflow) return;
    uint32_t j;
    const int w = (int)(t % (uint32_t)(27 * SORTED_LEVELS));
    if (!inc_item_point(G, alive, k, t / (uint32_t)(27 * SORTED_LEVELS), j)) return;
    int tl, l, c;
    size_t rr;
    uint4 dst;
    uint32_t off;
    if (!inc_dst_of(G, j, w, tl, l, c, rr, dst, off)) return;
    const float4 p = newp[j];
    const size_t at = (size_t)dst.x + rank[t];
    const uint32_t id = id_base + apos[j];
    M.bxyz[tl][at * 3 + 0] = p.x;
    M.bxyz[tl][at * 3 + 1] = p.y;
    M.bxyz[tl][at * 3 + 2] = p.z;
    M.bidx[tl][at] = id;
    const uint32_t pos = dst.z + rank[t];   // relative to the run's start (the 27 threads of a point write 54 contiguous bytes)
    M.backpos[(size_t)id * 27 + (uint32_t)c] = (uint16_t)(pos < (uint32_t)BACKPOS_FAR ? pos : (uint32_t)BACKPOS_FAR);
}
__global__ void inc_place_kernel(MapRW M, GroupRW G, const float4* __restrict__ newp, const uint32_t* __restrict__ alive,
                                 const uint32_t* __restrict__ apos, uint32_t k, uint32_t id_base, const uint32_t* __restrict__ rank) {
    LV_INC_ITEMS(27 * SORTED_LEVELS, 0u, gridDim.x, inc_place_item(M, G, newp, alive, apos, k, id_base, rank, t));
}

// pass 7: the owner of every touched target takes the batch tail in
__device__ __forceinline__ void inc_commit_item(const MapRW& M, const GroupRW& G, const uint32_t* __restrict__ alive, uint32_t k, uint32_t t) {
    int l, c;
    uint32_t gs;
    if (!inc_group_leader(G, alive, k, t, l, c, gs)) return;
    const size_t r = (size_t)gs * GROUP_TARGETS + (size_t)c;
    const uint32_t slot = G.gslot[l][r];
    if (slot == ID_NONE || G.gbase[l][r] != 0u) return;
    const LevelRW& L = M.lv[c < 27 ? l : CELL_SLOT];
    if (M.cnt->overflow == 0u) L.table[slot].w = L.aux[slot].tail0 + L.aux[slot].pending;   // (tail0: the count before the batch — after an in-place compaction, the living entries)
    L.aux[slot].pending = 0u;
}
__global__ void inc_commit_kernel(MapRW M, GroupRW G, const uint32_t* __restrict__ alive, uint32_t k) {
    LV_INC_ITEMS(REPL_LEVELS * GROUP_TARGETS, 0u, gridDim.x, inc_commit_item(M, G, alive, k, t));
}

// ---- runs compacted in place (see inc_reserve_item) -------------------------------------------------------------------------------
// COMPACT_LANES threads per listed run.  gather (with the launch that moves runs, before the batch's entries arrive): every entry
// is staged as it was, a living one with its new position = the number of living entries before it; scatter (with the launch that
// writes the batch's ids into the tails [living, living + pending), which it does not touch): the living entries go to their new
// positions — ascending id survives — with their back-positions, the positions behind the new tail read +inf.
constexpr int COMPACT_LANES = 64;
__device__ __forceinline__ void inc_compact_gather_item(const MapRW& M, uint32_t t, uint32_t n_threads) {
    if (M.cnt->overflow) return;
    const uint32_t lane = t % (uint32_t)COMPACT_LANES;
    list_for_each(M.n_comp, M.comp_cap, t / (uint32_t)COMPACT_LANES, n_threads / (uint32_t)COMPACT_LANES, [&](uint32_t r) {
        const uint4 c = M.comp[r];   // {slot, living, count before, staging base}
        const uint4 e = M.lv[0].table[c.x];
        const float* xs = M.bxyz[0] + (size_t)e.z * 3;
        const uint32_t* is = M.bidx[0] + e.z;
#if defined(__HIP_DEVICE_COMPILE__)
        // on the device the COMPACT_LANES threads of a run ARE one wavefront (consecutive thread ids, workgroups of a multiple of
        // 64 threads): the living entries before entry i = a ballot + a population count, 64 entries per step.  (The host
        // emulation below counts them one by one — the same positions; the one exception to "no wave intrinsics in this header":
        // the counting loop was 50 us of every batch.)
        static_assert(COMPACT_LANES == 64, "one wavefront per run");
        uint32_t before = 0;
        for (uint32_t i0 = 0; i0 < c.z; i0 += 64u) {
            const uint32_t i = i0 + lane;
            const bool in = i < c.z;
            const float x = in ? xs[(size_t)i * 3] : pos_inf();
            const bool alive = in && x < pos_inf();
            const unsigned long long mask = __ballot(alive);
            if (in) {
                M.cstage[c.w + i] = make_float4(x, xs[(size_t)i * 3 + 1], xs[(size_t)i * 3 + 2], __uint_as_float(is[i]));
                M.cnew[c.w + i] = alive ? before + (uint32_t)__popcll(mask & ((1ull << lane) - 1ull)) : ID_NONE;
            }
            before += (uint32_t)__popcll(mask);
        }
        if (lane == 0u && before != c.y) atomicExch(&M.cnt->overflow, 1u);   // (the count the decision was taken with: a mismatch would misplace the batch's tail)
#else
        for (uint32_t i = lane; i < c.z; i += (uint32_t)COMPACT_LANES) {
            const float x = xs[(size_t)i * 3];
            M.cstage[c.w + i] = make_float4(x, xs[(size_t)i * 3 + 1], xs[(size_t)i * 3 + 2], __uint_as_float(is[i]));
            uint32_t np = ID_NONE;
            if (x < pos_inf()) {
                np = 0;
                for (uint32_t j = 0; j < i; ++j) np += xs[(size_t)j * 3] < pos_inf() ? 1u : 0u;
            }
            M.cnew[c.w + i] = np;
            if (i + 1u == c.z) {   // (the run's last entry checks the count the decision was taken with: a mismatch would misplace the batch's tail)
                const uint32_t total = (np == ID_NONE ? 0u : np + 1u);
                uint32_t before = 0;
                if (np == ID_NONE) for (uint32_t j = 0; j < i; ++j) before += xs[(size_t)j * 3] < pos_inf() ? 1u : 0u;
                if ((np == ID_NONE ? before : total) != c.y) atomicExch(&M.cnt->overflow, 1u);
            }
        }
#endif
    });
}
__global__ void inc_compact_gather_kernel(MapRW M) { inc_compact_gather_item(M, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); }
__device__ __forceinline__ void inc_compact_scatter_item(const MapRW& M, uint32_t t, uint32_t n_threads) {
    if (M.cnt->overflow) return;
    const uint32_t lane = t % (uint32_t)COMPACT_LANES;
    list_for_each(M.n_comp, M.comp_cap, t / (uint32_t)COMPACT_LANES, n_threads / (uint32_t)COMPACT_LANES, [&](uint32_t r) {
        const uint4 c = M.comp[r];
        const uint4 e = M.lv[0].table[c.x];
        const uint64_t key = entry_key(e);
        const uint32_t bx = (uint32_t)(key & 0x1fffff), by = (uint32_t)((key >> 21) & 0x1fffff), bz = (uint32_t)((key >> 42) & 0x1fffff);
        const uint32_t tail_end = c.y + M.lv[0].aux[c.x].pending;   // the batch's entries take [living, living + pending)
        float* xs = M.bxyz[0] + (size_t)e.z * 3;
        uint32_t* is = M.bidx[0] + e.z;
        if (lane == 0u) M.lv[0].aux[c.x].dead = 0u;
        for (uint32_t i = lane; i < c.z; i += (uint32_t)COMPACT_LANES) {
            const uint32_t np = M.cnew[c.w + i];
            if (np != ID_NONE) {
                const float4 p = M.cstage[c.w + i];
                xs[(size_t)np * 3] = p.x; xs[(size_t)np * 3 + 1] = p.y; xs[(size_t)np * 3 + 2] = p.z;
                const uint32_t id = __float_as_uint(p.w);
                is[np] = id;
                M.backpos[(size_t)id * 27 + backpos_slot(p, M.origin, M.inv_cell, bx, by, bz)] = (uint16_t)(np < (uint32_t)BACKPOS_FAR ? np : (uint32_t)BACKPOS_FAR);
            }
            if (i >= tail_end) { xs[(size_t)i * 3] = pos_inf(); xs[(size_t)i * 3 + 1] = pos_inf(); xs[(size_t)i * 3 + 2] = pos_inf(); }
        }
    });
}
__global__ void inc_compact_scatter_kernel(MapRW M) { inc_compact_scatter_item(M, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x); }

// ---- tile groups broken up by the batch: laid out again ------------------------------------------------------------------------
// (behind the batch's last pass: counts and capacities are final).  A group's region = its up to eight runs side by side, every
// run with its capacity, the slack reading +inf; the old runs are abandoned where they lie (space comes back with the next
// re-linearisation, like the old place of a run that moved).  Positions inside a run do not change: back-positions stay valid.
// pass R1, one thread per listed group: find the group's buckets, take room for the region, publish the group's entry
__device__ __forceinline__ void inc_regroup_plan_item(const MapRW& M, RegroupPlan* __restrict__ plan, uint32_t g) {
    if (M.cnt->overflow) return;
    const uint32_t gs = M.broken[g];
    const uint64_t gkey = entry_key(M.gtable[gs]);
    const uint32_t vx = (uint32_t)(gkey & 0x1fffff), vy = (uint32_t)((gkey >> 21) & 0x1fffff), vz = (uint32_t)((gkey >> 42) & 0x1fffff);
    const LevelRW& L = M.lv[0];
    RegroupPlan& P = plan[g];
    uint32_t extent = 0;
    for (int r = 0; r < 8; ++r) {
        const uint32_t cx = 2u * vx - 1u + 3u * (uint32_t)(r & 1), cy = 2u * vy - 1u + 3u * (uint32_t)((r >> 1) & 1), cz = 2u * vz - 1u + 3u * (uint32_t)((r >> 2) & 1);
        P.slot[r] = ID_NONE;
        P.count[r] = P.cap[r] = P.from[r] = P.to[r] = 0u;
        if (cx >= (1u << 21) || cy >= (1u << 21) || cz >= (1u << 21)) continue;
        const uint32_t slot = table_find(L.table, L.mask, L.shift, pack_cell(cx, cy, cz));
        if (slot == ID_NONE) continue;
        const uint4 e = L.table[slot];
        P.slot[r] = slot;
        P.from[r] = e.z;
        P.count[r] = e.w;
        // every run of the group gets room again (half its count, like a run that moves): a group that is laid out afresh should
        // not come back with the next batch because a sibling run was one entry short
        const uint32_t fresh = e.w + (e.w / 2u > 8u ? e.w / 2u : 8u), have = L.aux[slot].cap;
        P.cap[r] = have > fresh ? have : fresh;
        P.to[r] = extent;        // (relative: the region's start is added below)
        extent += P.cap[r];
    }
    uint32_t ns = 0;
    bool got = extent == 0u;
    for (int tr = 0; tr < 4 && !got; ++tr) {
        const uint32_t ar = (gs + (uint32_t)tr * 17u) & (uint32_t)(N_ARENAS - 1);
        ns = atomicAdd(&M.cnt->arena_cur[0][ar], extent);
        got = (uint64_t)ns + extent <= (uint64_t)M.cnt->arena_end[0][ar];
    }
    if (!got) {
        atomicExch(&M.cnt->overflow, 1u);
        for (int r = 0; r < 8; ++r) P.slot[r] = ID_NONE;   // (nothing moves; the group stays broken until the re-linearisation that follows)
        M.gtable[gs].z = 0u;
        return;
    }
    for (int r = 0; r < 8; ++r) P.to[r] += ns;
    M.gtable[gs].z = ns;
    M.gtable[gs].w = extent;
}
__global__ void inc_regroup_plan_kernel(MapRW M, RegroupPlan* __restrict__ plan) {
    list_for_each(M.n_broken, M.broken_cap, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, [&](uint32_t g) { inc_regroup_plan_item(M, plan, g); });
}
// pass R2, REGROUP_LANES threads per listed group: the runs move (entries [0, count)), the slack behind them reads +inf
constexpr int REGROUP_LANES = 64;
__device__ __forceinline__ void inc_regroup_move_item(const MapRW& M, const RegroupPlan* __restrict__ plan, uint32_t t, uint32_t n_threads) {
    if (M.cnt->overflow) return;
    const uint32_t lane = t % (uint32_t)REGROUP_LANES;
    list_for_each(M.n_broken, M.broken_cap, t / (uint32_t)REGROUP_LANES, n_threads / (uint32_t)REGROUP_LANES, [&](uint32_t g) {
        const RegroupPlan& P = plan[g];
        for (int r = 0; r < 8; ++r) {
            if (P.slot[r] == ID_NONE) continue;
            float* xs = M.bxyz[0];
            uint32_t* is = M.bidx[0];
            for (uint32_t i = lane; i < P.cap[r]; i += (uint32_t)REGROUP_LANES) {
                const size_t d = (size_t)P.to[r] + i, sidx = (size_t)P.from[r] + i;
                if (i < P.count[r]) {
                    xs[d * 3] = xs[sidx * 3]; xs[d * 3 + 1] = xs[sidx * 3 + 1]; xs[d * 3 + 2] = xs[sidx * 3 + 2];
                    is[d] = is[sidx];
                } else {
                    xs[d * 3] = pos_inf(); xs[d * 3 + 1] = pos_inf(); xs[d * 3 + 2] = pos_inf();
                }
            }
        }
    });
}
__global__ void inc_regroup_move_kernel(MapRW M, const RegroupPlan* __restrict__ plan) {
    inc_regroup_move_item(M, plan, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x);
}
// pass R3, one thread per listed group: its buckets point at their new runs
__device__ __forceinline__ void inc_regroup_commit_item(const MapRW& M, const RegroupPlan* __restrict__ plan, uint32_t g) {
    if (M.cnt->overflow) return;
    const RegroupPlan& P = plan[g];
    for (int r = 0; r < 8; ++r) {
        if (P.slot[r] == ID_NONE) continue;
        M.lv[0].table[P.slot[r]].z = P.to[r];
        M.lv[0].aux[P.slot[r]].cap = P.cap[r];
    }
}
__global__ void inc_regroup_commit_kernel(MapRW M, const RegroupPlan* __restrict__ plan) {
    list_for_each(M.n_broken, M.broken_cap, blockIdx.x * blockDim.x + threadIdx.x, gridDim.x * blockDim.x, [&](uint32_t g) { inc_regroup_commit_item(M, plan, g); });
}

// ---- eviction --------------------------------------------------------------------------------------------------
// living points outside (keep_inside != 0) / inside (keep_inside == 0) the axis-aligned box [lo, hi] leave the map
__global__ void inc_evict_box_kernel(float4* __restrict__ orig, uint32_t n_ids, float lx, float ly, float lz, float hx, float hy,
                                     float hz, int keep_inside, float4* __restrict__ dead, uint32_t dead_cap, MapCounters* cnt) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_ids) return;
    const float4 p = orig[id];
    if (!pt_alive(p)) return;
    const bool inside = p.x >= lx && p.x <= hx && p.y >= ly && p.y <= hy && p.z >= lz && p.z <= hz;
    if (inside == (keep_inside != 0)) return;
    const uint32_t di = atomicAdd(&cnt->n_dead, 1u);
    if (di < dead_cap) dead[di] = make_float4(p.x, p.y, p.z, __uint_as_float(id));
    else atomicExch(&cnt->overflow, 1u);
    orig[id].x = pos_inf();
}

// the n_oldest oldest living points leave the map (rank[id] = number of living ids below id)
__global__ void inc_evict_oldest_kernel(float4* __restrict__ orig, uint32_t n_ids, const uint32_t* __restrict__ rank,
                                        uint32_t n_oldest, float4* __restrict__ dead, uint32_t dead_cap, MapCounters* cnt) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_ids) return;
    const float4 p = orig[id];
    if (!pt_alive(p) || rank[id] >= n_oldest) return;
    const uint32_t di = atomicAdd(&cnt->n_dead, 1u);
    if (di < dead_cap) dead[di] = make_float4(p.x, p.y, p.z, __uint_as_float(id));
    else atomicExch(&cnt->overflow, 1u);
    orig[id].x = pos_inf();
}

__global__ void inc_alive_flags_kernel(const float4* __restrict__ orig, uint32_t n_ids, uint32_t* __restrict__ flags) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_ids) return;
    flags[id] = pt_alive(orig[id]) ? 1u : 0u;
}

// compaction of the living into a fresh id space (re-linearisation)
__global__ void inc_compact_kernel(const float4* __restrict__ orig, const uint32_t* __restrict__ flags, const uint32_t* __restrict__ pos,
                                   uint32_t n_ids, float4* __restrict__ out) {
    const uint32_t id = blockIdx.x * blockDim.x + threadIdx.x;
    if (id >= n_ids) return;
    if (flags[id]) out[pos[id]] = orig[id];
}

#endif  // LV_MAPINC_KERNELS

}  // namespace lv
