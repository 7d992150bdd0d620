// lv_map.hip — map residency: voxel-hashed neighbourhood buckets in HBM, built once and then maintained in place.
//
// Replaces ikd-Tree's Build / Add_Points (call sites reference src/Modules/Mapper.cpp:68-76): instead of a pointer
// kd-tree (one point per node, ~100 B/node, log2(M) dependent loads per query) the map is held as
//   * `orig`: float4 by point id (insertion order = the index space of the kNN results);
//   * levels 0, 1 (voxel edge voxel_size * 2^l): for every voxel whose 3x3x3 block holds a point, the points of
//     that block in one contiguous run with slack behind it (bucket table {voxel -> run}); one probe + one
//     coalesced stream answer a query;
//   * level 2: one point list per voxel (27 / 216 lists searched by a whole wavefront for the few sparse queries).
// (Re)build: Morton sort of the points + per-level occupancy tables (scaffolding: `sorted`, `tables`) -> buckets.
// Afterwards inserts, down-sampling deletions and evictions only touch the buckets concerned (lv_mapinc.hpp); a
// re-linearisation (compaction of the ids + rebuild) happens when a pool, a table or the tombstones run high.
// Exactness of the search that uses these structures is argued in lv_match.hip.
#define LV_MAPINC_KERNELS
#include "lv_host.hpp"
#include "lv_ldssort.hpp"

#include <cstddef>
#include <cstring>

#include <hipcub/hipcub.hpp>

namespace lv {

__global__ void map_keys_kernel(const float4* __restrict__ pts, uint32_t m, float ox, float oy, float oz,
                                float inv_cell, uint64_t* __restrict__ keys, uint32_t* __restrict__ idx) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    float4 p = pts[i];
    uint32_t cx = (uint32_t)cell_coord(p.x, ox, inv_cell);
    uint32_t cy = (uint32_t)cell_coord(p.y, oy, inv_cell);
    uint32_t cz = (uint32_t)cell_coord(p.z, oz, inv_cell);
    keys[i] = morton3(cx, cy, cz);
    idx[i] = i;
}

__global__ void map_gather_kernel(const float4* __restrict__ pts, const uint32_t* __restrict__ idx_sorted, uint32_t m,
                                  float4* __restrict__ sorted) {
    uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    uint32_t j = idx_sorted[i];
    float4 p = pts[j];
    p.w = __uint_as_float(j);
    sorted[i] = p;
}

// number of Morton levels (from 0) at which element i starts a new voxel
__device__ __forceinline__ int head_levels(const uint64_t* keys, uint32_t i, int n_levels) {
    if (i == 0) return n_levels;
    uint64_t x = keys[i] ^ keys[i - 1];
    if (x == 0) return 0;
    int hb = 63 - __clzll((long long)x);
    int nl = hb / 3 + 1;
    return nl < n_levels ? nl : n_levels;
}

// occupied voxels per occupancy table (level 0, level 2)
__global__ void map_count_heads_kernel(const uint64_t* __restrict__ keys, uint32_t m, uint32_t* __restrict__ counts, uint32_t block_base) {
    __shared__ uint32_t s_cnt[N_OCC];
    if (threadIdx.x < N_OCC) s_cnt[threadIdx.x] = 0;
    __syncthreads();
    uint32_t i = (blockIdx.x + block_base) * blockDim.x + threadIdx.x;
    if (i < m) {
        const int nl = head_levels(keys, i, MAX_LEVELS);
#pragma unroll
        for (int t = 0; t < N_OCC; ++t)
            if (nl > occ_level(t)) atomicAdd(&s_cnt[t], 1u);
    }
    __syncthreads();
    if (threadIdx.x < N_OCC && s_cnt[threadIdx.x]) atomicAdd(&counts[threadIdx.x], s_cnt[threadIdx.x]);
}

struct GridLevelW {
    uint4* table;
    uint32_t mask;
    uint32_t shift;
};

struct TablePtrs {
    uint4* table[N_OCC];
    uint32_t mask[N_OCC];
    uint32_t shift[N_OCC];
};

__device__ __forceinline__ void map_insert_item(const uint64_t* __restrict__ keys, uint32_t m, const TablePtrs& tp, uint32_t i) {
    if (i >= m) return;
    const int nl = head_levels(keys, i, MAX_LEVELS);
    uint64_t k0 = keys[i];
#pragma unroll
    for (int t = 0; t < N_OCC; ++t) {
        const int l = occ_level(t);
        if (nl <= l) continue;
        uint64_t prefix = k0 >> (3 * l);
        // end = first j > i whose level-l prefix differs (keys are sorted)
        uint32_t lo = i + 1, hi = m;
        while (lo < hi) {
            uint32_t mid = lo + ((hi - lo) >> 1);
            if ((keys[mid] >> (3 * l)) == prefix) lo = mid + 1;
            else hi = mid;
        }
        uint32_t cx = compact21(k0) >> l, cy = compact21(k0 >> 1) >> l, cz = compact21(k0 >> 2) >> l;
        uint64_t key = pack_cell(cx, cy, cz);
        uint32_t slot = hash_cell(key, tp.shift[t]) & tp.mask[t];
        uint4* tbl = tp.table[t];
        for (;;) {
            unsigned long long* kp = reinterpret_cast<unsigned long long*>(&tbl[slot]);
            unsigned long long old = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)key);
            if (old == (unsigned long long)EMPTY_KEY) {
                tbl[slot].z = i;
                tbl[slot].w = lo - i;
                break;
            }
            slot = (slot + 1) & tp.mask[t];
        }
    }
}
__global__ void map_insert_kernel(const uint64_t* __restrict__ keys, uint32_t m, TablePtrs tp, uint32_t block_base) {
    map_insert_item(keys, m, tp, (blockIdx.x + block_base) * blockDim.x + threadIdx.x);
}


// ---- neighbourhood buckets -----------------------------------------------------------------------
// For every level-0 voxel c whose 3x3x3 block holds at least one point (the occupied voxels dilated by
// one), the points of that block are copied into one contiguous run ("bucket").  A query then needs ONE
// hash probe and ONE coalesced stream instead of 27 probes + 27 short dependent gathers; the
// search region, and therefore the exactness argument of lv_match.hip, is unchanged.  Every point lands
// in 27 buckets: HBM capacity (288 GB) is traded for latency — on ONE level since round 6 (the level-1
// block is eight of these buckets, lv_device.hpp REPL_LEVELS).
__device__ __forceinline__ bool probe_cell(const GridLevelW& g, uint64_t key, uint32_t& start, uint32_t& count) {
    uint32_t slot = hash_cell(key, g.shift) & g.mask;
    for (;;) {
        const uint4 e = g.table[slot];
        const uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
        if (ek == key) { start = e.z; count = e.w; return true; }
        if (ek == EMPTY_KEY) { start = 0; count = 0; return false; }
        slot = (slot + 1) & g.mask;
    }
}

// pass 1: every occupied voxel registers its 27 neighbours (incl. itself) in the bucket table
__device__ __forceinline__ void bucket_register_item(const GridLevelW& occ, uint32_t occ_slots, const GridLevelW& bt, uint32_t* __restrict__ cell_slots,
                                                     uint32_t cell_cap, uint32_t* __restrict__ flags, uint32_t t) {
    const uint32_t slot = t / 27, nb = t % 27;
    if (slot >= occ_slots) return;
    const uint4 e = occ.table[slot];
    const uint64_t key = (uint64_t)e.x | ((uint64_t)e.y << 32);
    if (key == EMPTY_KEY) return;
    const uint32_t cx = (uint32_t)(key & 0x1fffff), cy = (uint32_t)((key >> 21) & 0x1fffff), cz = (uint32_t)((key >> 42) & 0x1fffff);
    const int dz = (int)nb / 9 - 1, dy = ((int)nb / 3) % 3 - 1, dx = (int)nb % 3 - 1;
    const uint32_t nx = cx + dx, ny = cy + dy, nz = cz + dz;
    if (nx >= (1u << 21) || ny >= (1u << 21) || nz >= (1u << 21)) return;
    const uint64_t nkey = pack_cell(nx, ny, nz);
    uint32_t bs = hash_cell(nkey, bt.shift) & bt.mask;
    for (uint32_t probes = 0; probes <= bt.mask; ++probes) {
        unsigned long long* kp = reinterpret_cast<unsigned long long*>(&bt.table[bs]);
        const unsigned long long old = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)nkey);
        if (old == (unsigned long long)EMPTY_KEY) {
            const uint32_t idx = atomicAdd(&flags[0], 1u);
            if (idx < cell_cap) cell_slots[idx] = bs;
            else flags[1] = 1;
            return;
        }
        if (old == (unsigned long long)nkey) return;
        bs = (bs + 1) & bt.mask;
    }
    flags[1] = 1;
}
__global__ void bucket_register_kernel(GridLevelW occ, uint32_t occ_slots, GridLevelW bt, uint32_t* __restrict__ cell_slots,
                                       uint32_t cell_cap, uint32_t* __restrict__ flags, uint32_t block_base) {
    bucket_register_item(occ, occ_slots, bt, cell_slots, cell_cap, flags, (blockIdx.x + block_base) * blockDim.x + threadIdx.x);
}

// room a run of `count` entries is given when it is laid out: half its count, at least 8 entries (slack for appends; lv_mapinc.hpp
// relocates a run that outgrows it, to 1.5 x its new size, and lays its tile group out again).  With one replicated level the
// slack IS the map's memory (27 x 16 bytes x slack per point) and level 1 streams it with the runs (a group's region: runs +
// slack), so less was tried (profiles/experiments_r06): a quarter / 8 saves 0.13 KB per point and 1 us of the headline update's
// first launch — and makes every insert of new points pay for thousands of runs that outgrow their room at once, each taking its
// whole group along (64k new points 0.9 ms instead of 0.55, the pool gone after five scans); an eighth / 4 re-linearises every
// other scan.
#ifndef LV_RUN_SLACK_DIV
#define LV_RUN_SLACK_DIV 2u
#endif
#ifndef LV_RUN_SLACK_MIN
#define LV_RUN_SLACK_MIN 8u
#endif
__host__ __device__ __forceinline__ uint32_t run_capacity(uint32_t count) { return count + (count / LV_RUN_SLACK_DIV > LV_RUN_SLACK_MIN ? count / LV_RUN_SLACK_DIV : LV_RUN_SLACK_MIN); }

// The 27 source runs (in `sorted`) of bucket voxel `cell`: lane c < 27 of a wavefront probes neighbour c of the bucket's voxel in the
// occupancy table; returns this lane's (start, count) and the inclusive wave scan of the counts.
__device__ __forceinline__ void bucket_sources(const GridLevelW& occ, const GridLevelW& bt, uint32_t slot, int lane, uint32_t& start, uint32_t& count,
                                               uint32_t& incl, uint32_t& total) {
    const uint4 e = bt.table[slot];
    const uint64_t key = (uint64_t)e.x | ((uint64_t)e.y << 32);
    const uint32_t cx = (uint32_t)(key & 0x1fffff), cy = (uint32_t)((key >> 21) & 0x1fffff), cz = (uint32_t)((key >> 42) & 0x1fffff);
    start = 0; count = 0;
    if (lane < 27) {
        const int dz = lane / 9 - 1, dy = (lane / 3) % 3 - 1, dx = lane % 3 - 1;
        const uint32_t nx = cx + dx, ny = cy + dy, nz = cz + dz;
        if (nx < (1u << 21) && ny < (1u << 21) && nz < (1u << 21)) probe_cell(occ, pack_cell(nx, ny, nz), start, count);
    }
    incl = count;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        uint32_t v = __shfl_up(incl, off);
        if (lane >= off) incl += v;
    }
    total = __shfl(incl, 63);
}

// pass 2 (count): one wavefront per bucket voxel; buckets of more than 64 points are listed for bucket_build_big_kernel
__global__ __launch_bounds__(256) void bucket_count_kernel(GridLevelW occ, GridLevelW bt, const uint32_t* __restrict__ cell_slots, uint32_t n_cells,
                                                           uint32_t* __restrict__ bcount, uint32_t* __restrict__ bcap, uint32_t* __restrict__ biglist,
                                                           uint32_t* __restrict__ flags, uint32_t block_base) {
    const uint32_t cell = (blockIdx.x + block_base) * 4u + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63u);
    if (cell >= n_cells) return;
    uint32_t start, count, incl, total;
    bucket_sources(occ, bt, cell_slots[cell], lane, start, count, incl, total);
    if (lane == 0) {
        bcount[cell] = total;
        bcap[cell] = run_capacity(total);
        if (total > 64u) biglist[atomicAdd(&flags[2], 1u)] = cell;
    }
}

// ---- tile groups: the eight level-0 runs that tile a level-1 block side by side (lv_device.hpp REPL_LEVELS) ----------------------
// pass 2b: every bucket joins its group (a slot of the group table) and takes its place in the group's region: the running sum
// of the capacities of the runs that came before it (the order of the runs inside a region is of no consequence: level 1 streams
// the region as ONE candidate array).  boff[cell] = the group's slot, bcount[cell] = the run's offset inside the region.
__global__ void group_join_kernel(GridLevelW bt, const uint32_t* __restrict__ cell_slots, uint32_t n_cells, const uint32_t* __restrict__ bcap,
                                  GridLevelW gt, uint32_t* __restrict__ gext, uint32_t* __restrict__ boff, uint32_t* __restrict__ bcount,
                                  uint32_t* __restrict__ flags, uint32_t block_base) {
    const uint32_t cell = (blockIdx.x + block_base) * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    const uint4 e = bt.table[cell_slots[cell]];
    const uint64_t key = (uint64_t)e.x | ((uint64_t)e.y << 32);
    uint32_t vx, vy, vz;
    int r;
    tile_group_of((uint32_t)(key & 0x1fffff), (uint32_t)((key >> 21) & 0x1fffff), (uint32_t)((key >> 42) & 0x1fffff), vx, vy, vz, r);
    const uint64_t gkey = pack_cell(vx & 0x1fffffu, vy & 0x1fffffu, vz & 0x1fffffu);
    uint32_t gs = hash_cell(gkey, gt.shift) & gt.mask;
    for (uint32_t probes = 0; probes <= gt.mask; ++probes) {
        unsigned long long* kp = reinterpret_cast<unsigned long long*>(&gt.table[gs]);
        const unsigned long long old = atomicCAS(kp, (unsigned long long)EMPTY_KEY, (unsigned long long)gkey);
        if (old == (unsigned long long)EMPTY_KEY) atomicAdd(&flags[3], 1u);   // groups (the host checks the table's load)
        if (old == (unsigned long long)EMPTY_KEY || old == (unsigned long long)gkey) {
            boff[cell] = gs;
            bcount[cell] = atomicAdd(&gext[gs], bcap[cell]);
            return;
        }
        gs = (gs + 1) & gt.mask;
    }
    flags[1] = 1;
}
// pass 2c (after the exclusive scan of the extents): a run's place in the pool = its group's offset + its offset in the region;
// the group table's entries become {key, region start, region extent}
__global__ void group_place_kernel(uint32_t n_cells, const uint32_t* __restrict__ goff, uint32_t* __restrict__ boff, const uint32_t* __restrict__ bcount,
                                   uint32_t block_base) {
    const uint32_t cell = (blockIdx.x + block_base) * blockDim.x + threadIdx.x;
    if (cell >= n_cells) return;
    boff[cell] = goff[boff[cell]] + bcount[cell];
}
__global__ void group_commit_kernel(GridLevelW gt, const uint32_t* __restrict__ goff, const uint32_t* __restrict__ gext, uint32_t block_base) {
    const uint32_t gs = (blockIdx.x + block_base) * blockDim.x + threadIdx.x;
    if (gs > gt.mask) return;
    const uint4 e = gt.table[gs];
    if (((uint64_t)e.x | ((uint64_t)e.y << 32)) == EMPTY_KEY) return;
    gt.table[gs].z = goff[gs];
    gt.table[gs].w = gext[gs];
}

struct BuildOut {
    float* bxyz;
    uint32_t* bidx;
    uint16_t* backpos;
    SlotAux* aux;
    float origin[3];
    float inv_cell;
};
__device__ __forceinline__ void bucket_emit(const BuildOut& o, size_t at, uint32_t pos, const float4& p, uint32_t bx, uint32_t by, uint32_t bz) {
    o.bxyz[at * 3] = p.x;
    o.bxyz[at * 3 + 1] = p.y;
    o.bxyz[at * 3 + 2] = p.z;
    const uint32_t id = __float_as_uint(p.w);
    o.bidx[at] = id;
    o.backpos[(size_t)id * 27 + backpos_slot(p, o.origin, o.inv_cell, bx, by, bz)] = (uint16_t)(pos < (uint32_t)BACKPOS_FAR ? pos : (uint32_t)BACKPOS_FAR);
}

// pass 3 (build), buckets of up to 64 points (the bulk): one wavefront per bucket gathers the block's points from `sorted`, orders
// them by ORIGINAL index — the match kernel orders candidates by (distance, position-in-bucket); with this layout that equals
// the reference's (distance, index) order, ties included — with a bitonic network through cross-lane shuffles (no LDS for the
// keys, no barriers) and writes the 12-byte points, the ids and every point's back-position straight into the pool.  Rounds 1-5
// went through a float4 staging copy of the whole level (fill -> sort -> pack: three passes, 730 bytes per map point of scratch).
__global__ __launch_bounds__(256) void bucket_build_kernel(GridLevelW occ, GridLevelW bt, const uint32_t* __restrict__ cell_slots, uint32_t n_cells,
                                                           const float4* __restrict__ sorted, const uint32_t* __restrict__ bcap,
                                                           const uint32_t* __restrict__ boff, BuildOut o, uint32_t block_base) {
    __shared__ uint32_t s_pref[4][32], s_start[4][32];
    const uint32_t cell = (blockIdx.x + block_base) * 4u + (threadIdx.x >> 6);
    const int lane = (int)(threadIdx.x & 63u), w = (int)(threadIdx.x >> 6);
    if (cell >= n_cells) return;   // (whole wavefronts leave: the LDS rows are per wavefront)
    const uint32_t slot = cell_slots[cell];
    uint32_t start, count, incl, total;
    bucket_sources(occ, bt, slot, lane, start, count, incl, total);
    const uint32_t off = boff[cell];
    if (lane == 0) {
        bt.table[slot].z = off;
        bt.table[slot].w = total;
        o.aux[slot] = SlotAux{bcap[cell], 0u, 0u, 0u};
    }
    if (total == 0u || total > 64u) return;   // (more than 64: bucket_build_big_kernel)
    if (lane < 32) { s_pref[w][lane] = lane < 27 ? incl - count : total; s_start[w][lane] = start; }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    float4 v = make_float4(0.f, 0.f, 0.f, __uint_as_float(0xFFFFFFFFu));
    if ((uint32_t)lane < total) {
        int L = 0;   // the last source run whose first position is <= lane (empty runs share their successor's)
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1)
            if (s_pref[w][L + step] <= (uint32_t)lane) L += step;
        v = sorted[s_start[w][L] + ((uint32_t)lane - s_pref[w][L])];
    }
    for (int k = 2; k <= 64; k <<= 1) {
        for (int j = k >> 1; j > 0; j >>= 1) {
            const int partner = lane ^ j;
            float4 q;
            q.x = __shfl(v.x, partner); q.y = __shfl(v.y, partner); q.z = __shfl(v.z, partner); q.w = __shfl(v.w, partner);
            const bool take_min = ((lane & k) == 0) == ((lane & j) == 0);
            const uint32_t a = __float_as_uint(v.w), b = __float_as_uint(q.w);
            const bool use_other = take_min ? (b < a) : (b > a);
            if (use_other) v = q;
        }
    }
    if ((uint32_t)lane < total) {
        const uint4 e = bt.table[slot];
        bucket_emit(o, (size_t)off + (uint32_t)lane, (uint32_t)lane, v, e.x & 0x1fffffu, ((e.x >> 21) | (e.y << 11)) & 0x1fffffu, (e.y >> 10) & 0x1fffffu);
    }
}

// ... and the buckets of more than 64 points, a workgroup each: rank by counting — original indices are unique, so the rank of
// a point is the number of points of the bucket with a smaller index; up to BIG_LDS ids out of LDS (broadcast reads, no barriers
// between steps), beyond that (a 1.5-voxel cube with thousands of points: a map far denser than its voxel size was chosen for) by
// re-gathering them from memory: slow, exact.  The source is `sorted`, the destination the pool: nothing is permuted in place.
constexpr int BIG_T = 256;
constexpr uint32_t BIG_LDS = 4096;
__global__ __launch_bounds__(BIG_T) void bucket_build_big_kernel(GridLevelW occ, GridLevelW bt, const uint32_t* __restrict__ cell_slots,
                                                                 const uint32_t* __restrict__ biglist, uint32_t n_big,
                                                                 const float4* __restrict__ sorted, const uint32_t* __restrict__ boff, BuildOut o,
                                                                 uint32_t block_base) {
    __shared__ uint32_t s_pref[32], s_start[32];
    __shared__ uint32_t s_id[BIG_LDS];
    const uint32_t b = blockIdx.x + block_base;
    if (b >= n_big) return;
    const uint32_t cell = biglist[b];
    const uint32_t slot = cell_slots[cell];
    const uint32_t tid = threadIdx.x;
    uint32_t total = 0;
    if (tid < 64u) {
        uint32_t start, count, incl;
        bucket_sources(occ, bt, slot, (int)tid, start, count, incl, total);
        if (tid < 32u) { s_pref[tid] = tid < 27u ? incl - count : total; s_start[tid] = start; }
    }
    __syncthreads();
    total = s_pref[31];
    const uint4 e = bt.table[slot];
    const uint32_t bx = e.x & 0x1fffffu, by = ((e.x >> 21) | (e.y << 11)) & 0x1fffffu, bz = (e.y >> 10) & 0x1fffffu;
    const uint32_t off = boff[cell];
    auto src_of = [&](uint32_t i) {
        int L = 0;
#pragma unroll
        for (int step = 16; step >= 1; step >>= 1)
            if (s_pref[L + step] <= i) L += step;
        return s_start[L] + (i - s_pref[L]);
    };
    const bool in_lds = total <= BIG_LDS;
    if (in_lds) {
        for (uint32_t i = tid; i < total; i += BIG_T) s_id[i] = __float_as_uint(sorted[src_of(i)].w);
        __syncthreads();
    }
    for (uint32_t i = tid; i < total; i += BIG_T) {
        const float4 p = sorted[src_of(i)];
        const uint32_t mine = __float_as_uint(p.w);
        uint32_t rank = 0;
        if (in_lds) {
            for (uint32_t j = 0; j < total; ++j) rank += s_id[j] < mine ? 1u : 0u;
        } else {
            for (uint32_t j = 0; j < total; ++j) rank += __float_as_uint(sorted[src_of(j)].w) < mine ? 1u : 0u;
        }
        bucket_emit(o, (size_t)off + rank, rank, p, bx, by, bz);
    }
}

// ---- level 2: voxel lists -----------------------------------------------------------------------------------
__global__ void cell_caps_kernel(const uint4* __restrict__ table, uint32_t size, uint32_t* __restrict__ ccap) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= size) return;
    const uint4 e = table[slot];
    const uint64_t key = (uint64_t)e.x | ((uint64_t)e.y << 32);
    ccap[slot] = key == EMPTY_KEY ? 0u : run_capacity(e.w) + 8u;
}
// sorted[i] goes to its level-2 voxel's list, at the rank it has inside the voxel's run of `sorted`
__global__ void cell_fill_kernel(const uint64_t* __restrict__ keys_sorted, const float4* __restrict__ sorted, uint32_t m,
                                 GridLevelW t2, const uint32_t* __restrict__ coff, float4* __restrict__ cell4,
                                 uint32_t* __restrict__ cellpos) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= m) return;
    const uint64_t k0 = keys_sorted[i];
    const uint32_t cx = compact21(k0) >> CELL_LEVEL, cy = compact21(k0 >> 1) >> CELL_LEVEL, cz = compact21(k0 >> 2) >> CELL_LEVEL;
    const uint64_t key = pack_cell(cx, cy, cz);
    uint32_t slot = hash_cell(key, t2.shift) & t2.mask;
    for (;;) {
        const uint4 e = t2.table[slot];
        const uint64_t ek = (uint64_t)e.x | ((uint64_t)e.y << 32);
        if (ek == key) {
            const float4 p = sorted[i];
            cell4[(size_t)coff[slot] + (i - e.z)] = p;
            cellpos[__float_as_uint(p.w)] = i - e.z;
            return;
        }
        if (ek == EMPTY_KEY) return;   // cannot happen: every point's voxel was inserted
        slot = (slot + 1) & t2.mask;
    }
}
__global__ void cell_commit_kernel(uint4* __restrict__ table, uint32_t size, const uint32_t* __restrict__ ccap,
                                   const uint32_t* __restrict__ coff, SlotAux* __restrict__ aux) {
    const uint32_t slot = blockIdx.x * blockDim.x + threadIdx.x;
    if (slot >= size) return;
    aux[slot] = SlotAux{ccap[slot], 0u, 0u, 0u};
    if (ccap[slot]) table[slot].z = coff[slot];
}

// bounds of the living points (order-preserving float -> uint, atomicMin / atomicMax)
__device__ __forceinline__ unsigned flipf(float f) {
    const unsigned u = __float_as_uint(f);
    return (u & 0x80000000u) ? ~u : (u | 0x80000000u);
}
static inline float unflipf(unsigned u) {
    const unsigned v = (u & 0x80000000u) ? (u & 0x7fffffffu) : ~u;
    float f;
    memcpy(&f, &v, 4);
    return f;
}
__global__ __launch_bounds__(256) void map_bounds_kernel(const float4* __restrict__ pts, uint32_t n, unsigned* __restrict__ bounds) {
    // per-workgroup bounds in LDS first: six same-address atomics per POINT cost 1.1 ms per million points
    __shared__ unsigned s_b[6];
    if (threadIdx.x < 3) s_b[threadIdx.x] = 0xFFFFFFFFu;
    else if (threadIdx.x < 6) s_b[threadIdx.x] = 0u;
    __syncthreads();
    const uint32_t stride = gridDim.x * blockDim.x;
    unsigned lo[3] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu}, hi[3] = {0u, 0u, 0u};
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const float4 p = pts[i];
        if (!pt_alive(p)) continue;
        const unsigned f[3] = {flipf(p.x), flipf(p.y), flipf(p.z)};
#pragma unroll
        for (int a = 0; a < 3; ++a) { lo[a] = f[a] < lo[a] ? f[a] : lo[a]; hi[a] = f[a] > hi[a] ? f[a] : hi[a]; }
    }
#pragma unroll
    for (int a = 0; a < 3; ++a) { atomicMin(&s_b[a], lo[a]); atomicMax(&s_b[3 + a], hi[a]); }
    __syncthreads();
    if (threadIdx.x < 3) atomicMin(&bounds[threadIdx.x], s_b[threadIdx.x]);
    else if (threadIdx.x < 6) atomicMax(&bounds[threadIdx.x], s_b[threadIdx.x]);
}

// staged points that can be inserted (finite) — used when a batch BUILDS the map (Mapper::add on an empty map)
__global__ void staged_ok_kernel(const float4* __restrict__ pts, uint32_t k, uint32_t* __restrict__ flags) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= k) return;
    const float4 p = pts[i];
    flags[i] = (pt_alive(p) && fabsf(p.y) < pos_inf() && fabsf(p.z) < pos_inf()) ? 1u : 0u;
}

static inline uint32_t next_pow2(uint64_t v) {
    uint32_t p = 64;
    while (p < v) p <<= 1;
    return p;
}
static inline int log2u(uint32_t size) {
    int lg = 0;
    while ((1u << lg) < size) ++lg;
    return lg;
}

#define LV_REALLOC(ptr, type, count)                                       \
    do {                                                                   \
        if (ptr) hipFree(ptr);                                             \
        ptr = nullptr;                                                     \
        LV_HIP(hipMalloc(&ptr, (size_t)(count) * sizeof(type)));           \
    } while (0)

// A (re)build's large grids — millions of small workgroups, 2..10 ms per kernel at 10 M points — go out in SLICES when the store
// is being rebuilt in the background (MapStore::slice_wgs != 0): a workgroup that needs a whole compute unit (pass_kernel: 1024
// threads, 128 VGPRs, 156 KB of LDS) cannot be placed while another stream keeps every CU topped up with small workgroups — it
// waits for that stream's kernel to run out of workgroups, however the streams' priorities are set, and a CU mask on the
// other stream is not honoured here (scripts/ubench/cu_mask_starve.hip: 52 ms behind a 52 ms flood).  At the end of a slice the
// CUs drain and the waiting workgroup gets its unit: the stall is bounded by one slice (~0.1 ms: slice_wgs workgroups of the
// heaviest kernel, bucket_build_kernel; more of the lighter ones), not by one kernel.
// Every sliced kernel takes the first block index of its slice as its LAST argument.
// (Round 5 also carried a PACED twin of every such kernel — at most 32 looping 1024-thread workgroups, so that half of the CUs
// stay empty — which removed the waiting but showed two 4.6 ms cycles in 5 of 17 replays that plain slices never did; the cause
// was not found and the form is gone: profiles/experiments_r05/async_rebuild.txt sections 7-11 keep its measurements.)
// (round 6) ... and the slices can be SPACED: behind every slice one wavefront that sleeps for `pause` microseconds, so that the chip is
// empty for that long and whatever the calling cycle launches meanwhile starts at once (set by the rebuild's worker thread for its own
// launches only: slice_pause_us is thread-local; 0 = slices back to back)
static thread_local uint32_t slice_pause_us = 0;
void set_slice_pause_us(uint32_t us) { slice_pause_us = us; }
__global__ void slice_pause_kernel(uint32_t us) {
    const long long t0 = wall_clock64();
    while (wall_clock64() - t0 < (long long)us * 100) __builtin_amdgcn_s_sleep(64);   // (100 MHz)
}
template <typename K, typename... A>
static void launch_sliced(uint32_t slice, K kernel, uint32_t grid, uint32_t block, hipStream_t stream, A... args) {
    if (slice == 0 || grid <= slice) { hipLaunchKernelGGL(kernel, dim3(grid), dim3(block), 0, stream, args..., 0u); return; }
    for (uint32_t b = 0; b < grid; b += slice) {
        hipLaunchKernelGGL(kernel, dim3(grid - b < slice ? grid - b : slice), dim3(block), 0, stream, args..., b);
        if (slice_pause_us) hipLaunchKernelGGL(slice_pause_kernel, dim3(1), dim3(64), 0, stream, slice_pause_us);
    }
}

int MapStore::reserve(size_t cap) {
    if (cap <= capacity) return LV_OK;
    size_t ncap = capacity ? capacity : 4096;
    while (ncap < cap) ncap *= 2;
    float4* n_orig = nullptr;
    LV_HIP(hipMalloc(&n_orig, ncap * sizeof(float4)));
    LV_HIP(hipDeviceSynchronize());
    if (d_orig && n_ids) LV_HIP(hipMemcpy(n_orig, d_orig, (size_t)n_ids * sizeof(float4), hipMemcpyDeviceToDevice));
    if (d_orig) hipFree(d_orig);
    d_orig = n_orig;
    LV_REALLOC(d_orig2, float4, ncap);
    LV_REALLOC(d_sorted, float4, ncap);
    LV_REALLOC(d_keys, uint64_t, ncap);
    LV_REALLOC(d_keys_sorted, uint64_t, ncap);
    LV_REALLOC(d_idx, uint32_t, ncap);
    LV_REALLOC(d_idx_sorted, uint32_t, ncap);
    if (d_sort_tmp) hipFree(d_sort_tmp);
    d_sort_tmp = nullptr;
    sort_tmp_bytes = 0;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(nullptr, sort_tmp_bytes, d_keys, d_keys_sorted, d_idx, d_idx_sorted,
                                                           (int)ncap, 0, 63, (hipStream_t)0));
    LV_HIP(hipMalloc(&d_sort_tmp, sort_tmp_bytes));
    LV_REALLOC(d_dead, float4, ncap);
    dead_cap = ncap;
    if (d_box_next) {   // chains are by id: keep them
        uint32_t* nn = nullptr;
        LV_HIP(hipMalloc(&nn, ncap * sizeof(uint32_t)));
        if (n_ids) LV_HIP(hipMemcpy(nn, d_box_next, (size_t)n_ids * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        hipFree(d_box_next);
        d_box_next = nn;
        box_next_cap = ncap;
    }
    if (d_backpos) {   // positions inside the buckets are by id: keep them
        uint16_t* nb = nullptr;
        LV_HIP(hipMalloc(&nb, ncap * 27 * sizeof(uint16_t)));
        if (n_ids) LV_HIP(hipMemcpy(nb, d_backpos, (size_t)n_ids * 27 * sizeof(uint16_t), hipMemcpyDeviceToDevice));
        hipFree(d_backpos);
        d_backpos = nb;
        uint32_t* nc = nullptr;
        LV_HIP(hipMalloc(&nc, ncap * sizeof(uint32_t)));
        if (n_ids) LV_HIP(hipMemcpy(nc, d_cellpos, (size_t)n_ids * sizeof(uint32_t), hipMemcpyDeviceToDevice));
        hipFree(d_cellpos);
        d_cellpos = nc;
        backptr_cap = ncap;
    }
    capacity = ncap;
    refresh_view();
    return LV_OK;
}

int MapStore::ensure_alive_scratch() {
    if (alive_cap >= capacity && d_alive) return LV_OK;
    LV_REALLOC(d_alive, uint32_t, capacity);
    LV_REALLOC(d_apos, uint32_t, capacity);
    if (d_ascan_tmp) hipFree(d_ascan_tmp);
    d_ascan_tmp = nullptr;
    ascan_tmp_bytes = 0;
    LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, ascan_tmp_bytes, d_alive, d_apos, (int)capacity, (hipStream_t)0));
    LV_HIP(hipMalloc(&d_ascan_tmp, ascan_tmp_bytes));
    alive_cap = capacity;
    return LV_OK;
}

void MapStore::release() {
    hipFree(d_orig); hipFree(d_orig2); hipFree(d_sorted); hipFree(d_keys); hipFree(d_keys_sorted); hipFree(d_idx); hipFree(d_idx_sorted);
    hipFree(d_sort_tmp); hipFree(d_counts);
    hipFree(d_cell_slots); hipFree(d_flags); hipFree(d_bcount); hipFree(d_bcap); hipFree(d_boff); hipFree(d_scan_tmp);
    for (int l = 0; l < REPL_LEVELS; ++l) { hipFree(d_btable[l]); hipFree(d_baux[l]); }
    for (int l = 0; l < SORTED_LEVELS; ++l) { hipFree(d_bxyz[l]); hipFree(d_bidx[l]); }
    hipFree(d_backpos); hipFree(d_cellpos); hipFree(d_biglist); hipFree(d_gtable); hipFree(d_gext); hipFree(d_goff);
    hipFree(d_broken); hipFree(d_regroup); hipFree(d_comp); hipFree(d_cstage); hipFree(d_cnew);
    hipFree(d_caux); hipFree(d_cell4);
    for (int l = 0; l < N_OCC; ++l) hipFree(d_tables[l]);
    hipFree(d_cnt);
    if (h_cnt) hipHostFree(h_cnt);
    hipFree(d_new); hipFree(d_nkeys); hipFree(d_nkeys_sorted); hipFree(d_nidx); hipFree(d_nidx_sorted); hipFree(d_nalive);
    hipFree(d_napos); hipFree(d_nsurv); hipFree(d_nsflag); hipFree(d_nspos); hipFree(d_rank); hipFree(d_ntmp); hipFree(d_dead); hipFree(d_alive); hipFree(d_apos); hipFree(d_ascan_tmp);
    hipFree(d_box); hipFree(d_box_next);
    for (int l = 0; l < REPL_LEVELS; ++l) { hipFree(d_gtab[l]); hipFree(d_gbase[l]); hipFree(d_gslot[l]); hipFree(d_gdst[l]); }
    note_free(notes);
    hipFree(d_prank); hipFree(d_pslot); hipFree(d_gcnt); hipFree(d_reloc);
    *this = MapStore();
}

void MapStore::refresh_view() {
    view.orig = d_orig;
    view.m = built ? m : 0u;
    view.n_ids = n_ids;
    view.cell = cell;
    view.inv_cell = 1.0f / cell;
    for (int a = 0; a < 3; ++a) view.origin[a] = origin[a];
    for (int l = 0; l < REPL_LEVELS; ++l) {
        view.bt[l].table = d_btable[l];
        view.bt[l].mask = btable_size[l] ? btable_size[l] - 1 : 0;
        view.bt[l].shift = (uint32_t)(64 - log2u(btable_size[l] ? btable_size[l] : 1));
    }
    for (int l = 0; l < SORTED_LEVELS; ++l) {
        view.bxyz[l] = d_bxyz[l];
        view.bidx[l] = d_bidx[l];
    }
    view.gt.table = d_gtable;
    view.gt.mask = gtable_size ? gtable_size - 1 : 0;
    view.gt.shift = (uint32_t)(64 - log2u(gtable_size ? gtable_size : 1));
    view.ct.table = d_tables[OCC_CELL];
    view.ct.mask = table_size[OCC_CELL] ? table_size[OCC_CELL] - 1 : 0;
    view.ct.shift = (uint32_t)(64 - log2u(table_size[OCC_CELL] ? table_size[OCC_CELL] : 1));
    view.cell4 = d_cell4;
}

MapRW MapStore::rw() const {
    MapRW M{};
    M.orig = d_orig;
    for (int l = 0; l < REPL_LEVELS; ++l) {
        M.lv[l].table = d_btable[l];
        M.lv[l].aux = d_baux[l];
        M.lv[l].mask = btable_size[l] - 1;
        M.lv[l].shift = (uint32_t)(64 - log2u(btable_size[l]));
        M.lv[l].slot_limit = (uint32_t)((uint64_t)btable_size[l] * 7 / 10);
        M.lv[l].pool_cap = (uint32_t)(pool_cap[l] > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : pool_cap[l]);
    }
    for (int l = 0; l < SORTED_LEVELS; ++l) {
        M.bxyz[l] = d_bxyz[l];
        M.bidx[l] = d_bidx[l];
    }
    M.backpos = d_backpos;
    M.gtable = d_gtable;
    M.gmask = gtable_size ? gtable_size - 1 : 0;
    M.gshift = (uint32_t)(64 - log2u(gtable_size ? gtable_size : 1));
    M.gslot_limit = (uint32_t)((uint64_t)gtable_size * 7 / 10);
    M.broken = nullptr;   // (add_staged hands the batch's lists over; sweeps break groups without listing them)
    M.broken_cap = 0;
    M.n_broken = nullptr;
    M.comp = nullptr;
    M.comp_cap = 0;
    M.n_comp = nullptr;
    M.cstage = nullptr;
    M.cnew = nullptr;
    M.cstage_cap = 0;
    M.cellpos = d_cellpos;
    M.lv[CELL_SLOT].table = d_tables[OCC_CELL];
    M.lv[CELL_SLOT].aux = d_caux;
    M.lv[CELL_SLOT].mask = table_size[OCC_CELL] - 1;
    M.lv[CELL_SLOT].shift = (uint32_t)(64 - log2u(table_size[OCC_CELL]));
    M.lv[CELL_SLOT].slot_limit = (uint32_t)((uint64_t)table_size[OCC_CELL] * 6 / 10);
    M.lv[CELL_SLOT].pool_cap = (uint32_t)(pool_cap[CELL_SLOT] > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : pool_cap[CELL_SLOT]);
    M.cell4 = d_cell4;
    for (int a = 0; a < 3; ++a) M.origin[a] = origin[a];
    M.inv_cell = 1.0f / cell;
    M.cnt = d_cnt;
    return M;
}

void MapStore::stats(MapStats* out) const {
    *out = MapStats{};
    out->living = m;
    out->ids = n_ids;
    out->capacity = capacity;
    if (h_cnt && built) {
        for (int l = 0; l < INC_LEVELS; ++l) {
            uint64_t used = pool_base[l];
            for (int a = 0; a < N_ARENAS; ++a) {
                const uint32_t start = pool_base[l] + (uint32_t)(((uint64_t)(pool_cap[l] - pool_base[l]) * a) / N_ARENAS);
                const uint32_t cur = h_cnt->arena_cur[l][a] < h_cnt->arena_end[l][a] ? h_cnt->arena_cur[l][a] : h_cnt->arena_end[l][a];
                used += cur - start;
            }
            out->pool_used[l] = used;
            out->slots_used[l] = h_cnt->slots_used[l];
        }
        out->tombstones = tombstones;
    }
    for (int l = 0; l < INC_LEVELS; ++l) out->pool_cap[l] = pool_cap[l];
    for (int l = 0; l < REPL_LEVELS; ++l) out->slots_cap[l] = btable_size[l];
    out->slots_cap[CELL_SLOT] = table_size[OCC_CELL];
    out->dropped = dropped_total;
    out->relinearisations = relinearisations;
    out->incremental_adds = incremental_adds;
    // every device allocation that grows with the map (the per-batch scratch of an insert does not: reserve_batch)
    uint64_t b = (uint64_t)capacity * (16 + 16 + 16 + 8 + 8 + 4 + 4 + 16);                                   // orig, orig2, sorted, keys x 2, idx x 2, dead
    for (int l = 0; l < REPL_LEVELS; ++l) b += (uint64_t)pool_cap[l] * 16 + (uint64_t)btable_size[l] * 32;   // 12-byte points + ids; table + aux
    b += (uint64_t)pool_cap[CELL_SLOT] * 16 + (uint64_t)caux_size * 16 + (uint64_t)backptr_cap * (27 * 2 + 4);   // lists, their aux; back-positions + cellpos
    for (int l = 0; l < N_OCC; ++l) b += (uint64_t)table_size[l] * 16;
    b += (uint64_t)cells_cap * 16 + (uint64_t)biglist_cap * 4;                                                // build scratch per bucket voxel
    b += (uint64_t)gtable_size * 16 + (uint64_t)gscratch_cap * 8;                                             // tile groups: table + build scratch
    b += (uint64_t)alive_cap * 8;                                                                             // compaction / eviction flags + ranks
    b += (uint64_t)box_size * 16 + (uint64_t)box_next_cap * 4;
    out->bytes = b;
}

// Search structure over d_orig[0 .. n_ids), all living (m == n_ids).
int MapStore::rebuild(hipStream_t stream) {
    built = false;
    have_boxes = false;
    pool_low = false;
    tombstones = 0;
    view = MapView();
    refresh_view();
    {
        const int rc0 = ensure_counters();
        if (rc0) return rc0;
    }
    std::memset(h_cnt, 0, sizeof(MapCounters));
    m = n_ids;
    if (m == 0) {
        LV_HIP(hipMemcpyAsync(d_cnt, h_cnt, sizeof(MapCounters), hipMemcpyHostToDevice, stream));
        LV_HIP(hipStreamSynchronize(stream));
        return LV_OK;   // no map: view.m stays 0
    }
    const int B = 256;
    const uint32_t grid = (m + B - 1) / B;
    if (!d_flags) LV_HIP(hipMalloc(&d_flags, 8 * sizeof(uint32_t)));
    const unsigned init[8] = {0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u, 0u, 0u, 0u};
    LV_HIP(hipMemcpyAsync(d_flags, init, sizeof(init), hipMemcpyHostToDevice, stream));
    hipLaunchKernelGGL(map_bounds_kernel, dim3(grid < 2048u ? grid : 2048u), dim3(B), 0, stream, d_orig, m, d_flags);
    unsigned hb[8];
    LV_HIP(hipMemcpyAsync(hb, d_flags, sizeof(hb), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));
    for (int a = 0; a < 3; ++a) { bbox_min[a] = unflipf(hb[a]); bbox_max[a] = unflipf(hb[3 + a]); }
    if (!origin_set) {  // origin = bbox centre snapped to the level-0 lattice; kept for the map's lifetime
        for (int a = 0; a < 3; ++a) origin[a] = floorf(0.5f * (bbox_min[a] + bbox_max[a]) / cell) * cell;
        origin_set = true;
    }
    for (int a = 0; a < 3; ++a) {
        const float lo = floorf((bbox_min[a] - origin[a]) / cell), hi = floorf((bbox_max[a] - origin[a]) / cell);
        if (!(fabsf(lo) < (float)CELL_FAR) || !(fabsf(hi) < (float)CELL_FAR)) {
            set_error("map extent exceeds +-%d voxels of %.3f m around the map origin", CELL_FAR, cell);
            return LV_ERANGE;
        }
    }
    const float inv_cell = 1.0f / cell;
    hipLaunchKernelGGL(map_keys_kernel, dim3(grid), dim3(B), 0, stream, d_orig, m, origin[0], origin[1], origin[2], inv_cell, d_keys,
                       d_idx);
    size_t tmp = sort_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_sort_tmp, tmp, d_keys, d_keys_sorted, d_idx, d_idx_sorted, (int)m,
                                                           0, 63, stream));
    hipLaunchKernelGGL(map_gather_kernel, dim3(grid), dim3(B), 0, stream, d_orig, d_idx_sorted, m, d_sorted);
    if (!d_counts) LV_HIP(hipMalloc(&d_counts, 16 * sizeof(uint32_t)));
    LV_HIP(hipMemsetAsync(d_counts, 0, 16 * sizeof(uint32_t), stream));
    launch_sliced(slice_wgs * 2, map_count_heads_kernel, grid, (uint32_t)B, stream, (const uint64_t*)d_keys_sorted, m, d_counts);
    uint32_t counts[16];
    LV_HIP(hipMemcpyAsync(counts, d_counts, sizeof(counts), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));

    TablePtrs tp{};
    for (int l = 0; l < N_OCC; ++l) {
        // the level-2 table lives on as the voxel-list table of incremental inserts: give it room to grow
        uint32_t size = next_pow2((uint64_t)counts[l] * (l == OCC_CELL ? 8 : 4));
        // (re)builds also give memory BACK: a table 8x larger than this map wants — a rolling window that shrank from its
        // initial extent — is re-allocated (hysteresis: growth doubles, so 8x cannot oscillate)
        if (size > table_size[l] || ((uint64_t)size * 8 <= table_size[l] && table_size[l] > (1u << 20))) {
            if (d_tables[l]) hipFree(d_tables[l]);
            d_tables[l] = nullptr;
            LV_HIP(hipMalloc(&d_tables[l], (size_t)size * sizeof(uint4)));
            table_size[l] = size;
        }
        size = table_size[l];
        LV_HIP(hipMemsetAsync(d_tables[l], 0xFF, (size_t)size * sizeof(uint4), stream));
        tp.table[l] = d_tables[l];
        tp.mask[l] = size - 1;
        tp.shift[l] = (uint32_t)(64 - log2u(size));
        n_cells[l] = counts[l];
    }
    launch_sliced(slice_wgs * 2, map_insert_kernel, grid, (uint32_t)B, stream, (const uint64_t*)d_keys_sorted, m, tp);
    LV_HIP(hipGetLastError());
    for (int l = 0; l < REPL_LEVELS; ++l) {
        int rc = build_buckets(stream, l, counts[l]);
        if (rc) return rc;
    }
    int rc = build_cells(stream, counts[OCC_CELL]);
    if (rc) return rc;
    for (int l = 0; l < INC_LEVELS; ++l) {   // the free part of every pool, split into arenas
        const uint64_t cap = pool_cap[l] > 0xFFFFFFF0ull ? 0xFFFFFFF0ull : pool_cap[l];
        const uint64_t free_entries = cap - pool_base[l];
        for (int a = 0; a < N_ARENAS; ++a) {
            h_cnt->arena_cur[l][a] = pool_base[l] + (uint32_t)((free_entries * a) / N_ARENAS);
            h_cnt->arena_end[l][a] = pool_base[l] + (uint32_t)((free_entries * (a + 1)) / N_ARENAS);
        }
    }
    for (int l = 0; l < REPL_LEVELS; ++l) h_cnt->slots_used[l] = n_bcells[l];
    h_cnt->gslots_used = n_groups;
    h_cnt->slots_used[CELL_SLOT] = counts[OCC_CELL];
    LV_HIP(hipMemcpyAsync(d_cnt, h_cnt, sizeof(MapCounters), hipMemcpyHostToDevice, stream));
    LV_HIP(hipStreamSynchronize(stream));
    built = true;
    refresh_view();
    return LV_OK;
}

// level 0 (the one replicated level; `level` stays a parameter of the pool / table arrays)
int MapStore::build_buckets(hipStream_t stream, int level, uint32_t n_occupied) {
    GridLevelW occ{d_tables[level], table_size[level] - 1, (uint32_t)(64 - log2u(table_size[level]))};
    const uint32_t occ_slots = table_size[level];
    uint32_t size = next_pow2((uint64_t)n_occupied * 16);  // dilation factor <= 8 keeps the load <= 0.5
    const bool give_back = (uint64_t)size * 8 <= btable_size[level] && btable_size[level] > (1u << 20);   // (see the occupancy tables)
    if (size < btable_size[level] && !give_back) size = btable_size[level];
    if (backptr_cap < capacity) {   // back-positions and list positions are by id
        LV_REALLOC(d_backpos, uint16_t, capacity * 27);
        LV_REALLOC(d_cellpos, uint32_t, capacity);
        backptr_cap = capacity;
    }
    for (;;) {
        if (size != btable_size[level] || !d_btable[level]) {
            LV_REALLOC(d_btable[level], uint4, size);
            LV_REALLOC(d_baux[level], SlotAux, size);
            btable_size[level] = size;
        }
        const size_t need_cells = (size_t)size / 2;
        if (need_cells > cells_cap) {
            LV_REALLOC(d_cell_slots, uint32_t, need_cells);
            LV_REALLOC(d_bcount, uint32_t, need_cells);
            LV_REALLOC(d_bcap, uint32_t, need_cells);
            LV_REALLOC(d_boff, uint32_t, need_cells);
            if (d_scan_tmp) hipFree(d_scan_tmp);
            d_scan_tmp = nullptr;
            scan_tmp_bytes = 0;
            LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp_bytes, d_bcap, d_boff, (int)need_cells, (hipStream_t)0));
            LV_HIP(hipMalloc(&d_scan_tmp, scan_tmp_bytes));
            cells_cap = need_cells;
        }
        if (need_cells > biglist_cap) {
            LV_REALLOC(d_biglist, uint32_t, need_cells);
            biglist_cap = need_cells;
        }
        LV_HIP(hipMemsetAsync(d_btable[level], 0xFF, (size_t)size * sizeof(uint4), stream));
        LV_HIP(hipMemsetAsync(d_baux[level], 0, (size_t)size * sizeof(SlotAux), stream));
        LV_HIP(hipMemsetAsync(d_flags, 0, 4 * sizeof(uint32_t), stream));
        GridLevelW bt{d_btable[level], size - 1, (uint32_t)(64 - log2u(size))};
        const uint64_t threads = (uint64_t)occ_slots * 27;
        launch_sliced(slice_wgs * 2, bucket_register_kernel, (uint32_t)((threads + 255) / 256), 256u, stream, occ, occ_slots, bt,
                      d_cell_slots, (uint32_t)(size / 2), d_flags);
        uint32_t flags[2];
        LV_HIP(hipMemcpyAsync(flags, d_flags, sizeof(flags), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        if (flags[1] || flags[0] > size / 2) {  // table too small for this map's dilation factor: double and retry
            size *= 2;
            continue;
        }
        const uint32_t nb = flags[0];
        n_bcells[level] = nb;
        launch_sliced(slice_wgs * 2, bucket_count_kernel, (nb + 3) / 4, 256u, stream, occ, bt, (const uint32_t*)d_cell_slots, nb, d_bcount, d_bcap,
                      d_biglist, d_flags);
        // the runs are laid out GROUP BY GROUP (the eight buckets that tile a level-1 block side by side): every bucket joins its
        // group and takes its place in the group's region, the regions follow each other in table-slot order
        uint32_t gsize = next_pow2(nb);   // (a group holds up to eight buckets; on surfaces ~4: load ~0.25; checked below)
        uint64_t total = 0;
        uint32_t n_big = 0;
        for (;;) {
            if (gsize > gtable_size || ((uint64_t)gsize * 8 <= gtable_size && gtable_size > (1u << 20))) {
                LV_REALLOC(d_gtable, uint4, gsize);
                gtable_size = gsize;
            }
            gsize = gtable_size;
            if (gsize > gscratch_cap) {
                LV_REALLOC(d_gext, uint32_t, gsize);
                LV_REALLOC(d_goff, uint32_t, gsize);
                gscratch_cap = gsize;
            }
            {   // (the scan's scratch was sized for the bucket voxels; a group table can be larger for tiny maps and after a retry)
                size_t need = 0;
                LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, need, d_gext, d_goff, (int)gsize, (hipStream_t)0));
                if (need > scan_tmp_bytes) {
                    if (d_scan_tmp) hipFree(d_scan_tmp);
                    d_scan_tmp = nullptr;
                    LV_HIP(hipMalloc(&d_scan_tmp, need));
                    scan_tmp_bytes = need;
                }
            }
            LV_HIP(hipMemsetAsync(d_gtable, 0xFF, (size_t)gsize * sizeof(uint4), stream));
            LV_HIP(hipMemsetAsync(d_gext, 0, (size_t)gsize * sizeof(uint32_t), stream));
            LV_HIP(hipMemsetAsync(d_flags + 3, 0, sizeof(uint32_t), stream));
            GridLevelW gt{d_gtable, gsize - 1, (uint32_t)(64 - log2u(gsize))};
            launch_sliced(slice_wgs * 8, group_join_kernel, (nb + 255) / 256, 256u, stream, bt, (const uint32_t*)d_cell_slots, nb, (const uint32_t*)d_bcap, gt,
                          d_gext, d_boff, d_bcount, d_flags);
            size_t stmp = scan_tmp_bytes;
            LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_scan_tmp, stmp, d_gext, d_goff, (int)gsize, stream));
            uint32_t last_off = 0, last_ext = 0, hf[4] = {0, 0, 0, 0};
            LV_HIP(hipMemcpyAsync(&last_off, d_goff + (gsize - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            LV_HIP(hipMemcpyAsync(&last_ext, d_gext + (gsize - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
            LV_HIP(hipMemcpyAsync(hf, d_flags, sizeof(hf), hipMemcpyDeviceToHost, stream));
            LV_HIP(hipStreamSynchronize(stream));
            if (hf[1] || (uint64_t)hf[3] * 10 > (uint64_t)gsize * 7) {   // too many groups for this table (isolated buckets): double and retry
                gsize *= 2;
                LV_HIP(hipMemsetAsync(d_flags + 1, 0, sizeof(uint32_t), stream));
                continue;
            }
            total = (uint64_t)last_off + last_ext;   // entries laid out, slack included
            n_big = hf[2];
            n_groups = hf[3];
            launch_sliced(slice_wgs * 8, group_place_kernel, (nb + 255) / 256, 256u, stream, nb, (const uint32_t*)d_goff, d_boff, (const uint32_t*)d_bcount);
            launch_sliced(slice_wgs * 8, group_commit_kernel, (gsize + 255) / 256, 256u, stream, gt, (const uint32_t*)d_goff, (const uint32_t*)d_gext);
            break;
        }
        // the pool keeps room for runs that move and for the buckets of newly mapped space
        // (the floor follows the map: 1 Mi entries for the small maps of tests and multi-context processes, up to the 8 Mi a
        // streaming map wants for a few scans' worth of newly mapped space between two re-linearisations)
        const uint64_t floor_entries = total < (1ull << 20) ? (1ull << 20) : (total > (8ull << 20) ? (8ull << 20) : total);
        const uint64_t want = total + total / 4 + floor_entries;
        if (want > 0xFFFFFFF0ull) { set_error("bucket pool exceeds 2^32 entries at level %d", level); return LV_ERANGE; }
        if (want > pool_cap[level] || (want * 3 <= pool_cap[level] && pool_cap[level] > (32u << 20))) {   // (grow, or give back: see the occupancy tables)
            pool_cap[level] = 0;
            LV_REALLOC(d_bxyz[level], float, want * 3 + 4);
            LV_REALLOC(d_bidx[level], uint32_t, want);
            pool_cap[level] = (size_t)want;
        }
        // the slack behind every run reads as +inf (a candidate at distance +inf, like a deleted entry): level 1 streams a group's
        // region runs AND slack
        if (total) LV_HIP(hipMemsetD32Async((hipDeviceptr_t)d_bxyz[level], 0x7F800000, (size_t)total * 3, stream));
        BuildOut bo{d_bxyz[level], d_bidx[level], d_backpos, d_baux[level], {origin[0], origin[1], origin[2]}, 1.0f / cell};
        launch_sliced(slice_wgs, bucket_build_kernel, (nb + 3) / 4, 256u, stream, occ, bt, (const uint32_t*)d_cell_slots, nb, (const float4*)d_sorted,
                      (const uint32_t*)d_bcap, (const uint32_t*)d_boff, bo);
        if (n_big)
            launch_sliced(slice_wgs, bucket_build_big_kernel, n_big, (uint32_t)BIG_T, stream, occ, bt, (const uint32_t*)d_cell_slots,
                          (const uint32_t*)d_biglist, n_big, (const float4*)d_sorted, (const uint32_t*)d_boff, bo);
        LV_HIP(hipGetLastError());
        pool_base[level] = (uint32_t)total;
        return LV_OK;
    }
}

// level 2: the occupancy table of the level (runs of `sorted`) turns into the voxel-list table (runs of cell4)
int MapStore::build_cells(hipStream_t stream, uint32_t n_occupied) {
    (void)n_occupied;
    const uint32_t size = table_size[OCC_CELL];
    if (size > caux_size) {
        LV_REALLOC(d_caux, SlotAux, size);
        caux_size = size;
    }
    if ((size_t)size > cells_cap) {
        LV_REALLOC(d_cell_slots, uint32_t, size);
        LV_REALLOC(d_bcount, uint32_t, size);
        LV_REALLOC(d_bcap, uint32_t, size);
        LV_REALLOC(d_boff, uint32_t, size);
        if (d_scan_tmp) hipFree(d_scan_tmp);
        d_scan_tmp = nullptr;
        scan_tmp_bytes = 0;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, scan_tmp_bytes, d_bcap, d_boff, (int)size, (hipStream_t)0));
        LV_HIP(hipMalloc(&d_scan_tmp, scan_tmp_bytes));
        cells_cap = size;
    }
    const int B = 256;
    hipLaunchKernelGGL(cell_caps_kernel, dim3((size + B - 1) / B), dim3(B), 0, stream, d_tables[OCC_CELL], size, d_bcap);
    size_t stmp = scan_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_scan_tmp, stmp, d_bcap, d_boff, (int)size, stream));
    uint32_t last_off = 0, last_cap = 0;
    LV_HIP(hipMemcpyAsync(&last_off, d_boff + (size - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipMemcpyAsync(&last_cap, d_bcap + (size - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));
    const uint64_t total = (uint64_t)last_off + last_cap;
    const uint64_t want = total + total / 4 + (1ull << 20);
    if (want > 0xFFFFFFF0ull) { set_error("voxel-list pool exceeds 2^32 entries"); return LV_ERANGE; }
    if (want > pool_cap[CELL_SLOT]) {
        pool_cap[CELL_SLOT] = 0;
        LV_REALLOC(d_cell4, float4, want);
        pool_cap[CELL_SLOT] = (size_t)want;
    }
    GridLevelW t2{d_tables[OCC_CELL], size - 1, (uint32_t)(64 - log2u(size))};
    hipLaunchKernelGGL(cell_fill_kernel, dim3((m + B - 1) / B), dim3(B), 0, stream, d_keys_sorted, d_sorted, m, t2, d_boff, d_cell4, d_cellpos);
    hipLaunchKernelGGL(cell_commit_kernel, dim3((size + B - 1) / B), dim3(B), 0, stream, d_tables[OCC_CELL], size, d_bcap, d_boff, d_caux);
    LV_HIP(hipGetLastError());
    pool_base[CELL_SLOT] = (uint32_t)total;
    return LV_OK;
}

// ---- incremental maintenance: host side ---------------------------------------------------------------------
int MapStore::reserve_batch(size_t k) {
    if (k <= batch_cap) return LV_OK;
    size_t ncap = batch_cap ? batch_cap : 4096;
    while (ncap < k) ncap *= 2;
    LV_REALLOC(d_new, float4, ncap);
    LV_REALLOC(d_nkeys, uint64_t, ncap);
    LV_REALLOC(d_nkeys_sorted, uint64_t, ncap);
    LV_REALLOC(d_nidx, uint32_t, ncap);
    LV_REALLOC(d_nidx_sorted, uint32_t, ncap);
    LV_REALLOC(d_nalive, uint32_t, ncap);
    LV_REALLOC(d_napos, uint32_t, ncap);
    LV_REALLOC(d_nsurv, uint32_t, ncap);
    LV_REALLOC(d_nsflag, uint32_t, ncap);
    LV_REALLOC(d_nspos, uint32_t, ncap);
    LV_REALLOC(d_rank, uint32_t, ncap * 27 * SORTED_LEVELS);
    gtab_size = next_pow2((uint64_t)ncap * 4);
    for (int l = 0; l < REPL_LEVELS; ++l) {
        LV_REALLOC(d_gtab[l], uint4, gtab_size);
        LV_REALLOC(d_gbase[l], uint32_t, (size_t)gtab_size * GROUP_TARGETS);
        LV_REALLOC(d_gslot[l], uint32_t, (size_t)gtab_size * GROUP_TARGETS);
        LV_REALLOC(d_gdst[l], uint4, (size_t)gtab_size * GROUP_TARGETS);
    }
    LV_REALLOC(d_prank, uint32_t, ncap * REPL_LEVELS);
    LV_REALLOC(d_pslot, uint32_t, ncap * REPL_LEVELS);
    if (!d_gcnt) LV_HIP(hipMalloc(&d_gcnt, 4 * LIST_SHARDS * sizeof(uint32_t)));
    reloc_cap = (uint32_t)(ncap * 27 > 0x0FFFFFF0ull ? 0x0FFFFFF0ull : ncap * 27);
    LV_REALLOC(d_reloc, uint4, reloc_cap);
    // a batch of k points breaks up at most 27 k groups (one per target bucket), in practice a few per cent of k: more than this
    // raises `overflow` and the map is re-linearised
    broken_cap = (uint32_t)(ncap * 2 < 16384 ? 16384 : ncap * 2);   // (sharded 64 ways by table slot: room for an uneven spread)
    LV_REALLOC(d_broken, uint32_t, broken_cap);
    LV_REALLOC(d_regroup, RegroupPlan, broken_cap);
    // runs compacted in place: the list (one entry per run) and the staging area (one entry per bucket entry of a listed run; a
    // run that finds it full moves instead)
    comp_cap = reloc_cap < (1u << 18) ? reloc_cap : (1u << 18);
    cstage_cap = (uint32_t)(ncap * 32 < (1u << 20) ? (1u << 20) : ncap * 32);
    LV_REALLOC(d_comp, uint4, comp_cap);
    LV_REALLOC(d_cstage, float4, cstage_cap);
    LV_REALLOC(d_cnew, uint32_t, cstage_cap);
    if (d_ntmp) hipFree(d_ntmp);
    d_ntmp = nullptr;
    size_t a = 0, b = 0;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(nullptr, a, d_nkeys, d_nkeys_sorted, d_nidx, d_nidx_sorted, (int)ncap, 0, 63,
                                                           (hipStream_t)0));
    LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(nullptr, b, d_nalive, d_napos, (int)ncap, (hipStream_t)0));
    ntmp_bytes = a > b ? a : b;
    LV_HIP(hipMalloc(&d_ntmp, ntmp_bytes));
    batch_cap = ncap;
    return LV_OK;
}

int MapStore::ensure_boxes(hipStream_t stream, float box_length) {
    if (have_boxes) return LV_OK;
    const uint32_t size = next_pow2((uint64_t)capacity * 4);
    if (size > box_size) {
        LV_REALLOC(d_box, uint4, size);
        box_size = size;
    }
    if (box_next_cap < capacity) {
        LV_REALLOC(d_box_next, uint32_t, capacity);
        box_next_cap = capacity;
    }
    LV_HIP(hipMemsetAsync(d_box, 0xFF, (size_t)box_size * sizeof(uint4), stream));
    LV_HIP(hipMemsetAsync(&d_cnt->box_slots_used, 0, sizeof(uint32_t), stream));
    BoxRW Bx{d_box, d_box_next, box_size - 1, (uint32_t)(64 - log2u(box_size)), (uint32_t)((uint64_t)box_size * 6 / 10), box_length};
    // (ALWAYS sliced: the boxes are built by the first down-sampling insert after a (re)build, on the insert's side stream,
    // beside the cycle's whole-CU launches — 4 ms in one piece at 10 M ids, ~0.1 ms per slice of 1024 workgroups)
    if (n_ids) launch_sliced(1024u, box_build_kernel, (n_ids + 255) / 256, 256u, stream, Bx, (const float4*)d_orig, n_ids, d_cnt);
    LV_HIP(hipGetLastError());
    have_boxes = true;
    return LV_OK;
}

bool MapStore::wants_relinearise(size_t incoming) const {
    if (!built) return false;
    if ((uint64_t)n_ids + incoming > 0xFFFFFFF0ull) return true;
    const uint64_t dead = (uint64_t)n_ids - m;
    if (dead > 65536 && dead > (uint64_t)n_ids / 3) return true;                      // a third of the id space is dead
    if (pool_low) return true;                                                         // the bucket pool is running out of room (settle)
    return false;
}
bool MapStore::needs_relinearise(size_t incoming) const {
    if (!built) return false;
    if ((uint64_t)n_ids + incoming > 0xFFFFFFF0ull) return true;
    return defer_relinearise ? false : wants_relinearise(incoming);
}

int MapStore::snapshot_into(MapStore& dst, hipStream_t stream) {
    int rc = ensure_alive_scratch();
    if (rc) return rc;
    dst.n_ids = 0;
    dst.m = 0;
    dst.built = false;
    dst.have_boxes = false;
    dst.counters_pending = false;
    rc = dst.reserve((size_t)m + 1);
    if (rc) return rc;
    dst.cell = cell;
    dst.origin_set = origin_set;          // the same voxel lattice
    for (int a = 0; a < 3; ++a) dst.origin[a] = origin[a];
    dst.sweep_evict = sweep_evict; dst.merged_back = merged_back; dst.surv_list = surv_list; dst.small_front = small_front;
    if (n_ids) {
        const uint32_t grid = (n_ids + 255) / 256;
        hipLaunchKernelGGL(inc_alive_flags_kernel, dim3(grid), dim3(256), 0, stream, d_orig, n_ids, d_alive);
        size_t tmp = ascan_tmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ascan_tmp, tmp, d_alive, d_apos, (int)n_ids, stream));
        hipLaunchKernelGGL(inc_compact_kernel, dim3(grid), dim3(256), 0, stream, d_orig, d_alive, d_apos, n_ids, dst.d_orig);
        LV_HIP(hipGetLastError());
    }
    dst.n_ids = m;   // (the living count is host-known once the last insert has been settled)
    return LV_OK;
}

int MapStore::relinearise(hipStream_t stream) {
    int rc = ensure_alive_scratch();
    if (rc) return rc;
    if (n_ids) {
        const uint32_t grid = (n_ids + 255) / 256;
        hipLaunchKernelGGL(inc_alive_flags_kernel, dim3(grid), dim3(256), 0, stream, d_orig, n_ids, d_alive);
        size_t tmp = ascan_tmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ascan_tmp, tmp, d_alive, d_apos, (int)n_ids, stream));
        hipLaunchKernelGGL(inc_compact_kernel, dim3(grid), dim3(256), 0, stream, d_orig, d_alive, d_apos, n_ids, d_orig2);
        uint32_t last_pos = 0, last_alive = 0;
        LV_HIP(hipMemcpyAsync(&last_pos, d_apos + (n_ids - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipMemcpyAsync(&last_alive, d_alive + (n_ids - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        float4* t = d_orig;
        d_orig = d_orig2;
        d_orig2 = t;
        n_ids = last_pos + last_alive;
    }
    m = n_ids;
    ++relinearisations;
    return rebuild(stream);
}

int MapStore::kill_dead_list(hipStream_t stream, uint32_t n_dead) {
    if (n_dead == 0) return LV_OK;
    const uint64_t threads = (uint64_t)n_dead * INC_SLOTS_PER_POINT;
    hipLaunchKernelGGL(inc_kill_kernel, dim3((uint32_t)((threads + 255) / 256)), dim3(256), 0, stream, rw(), d_dead, n_dead);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

// (the kernels of the incremental insert that are not one-thread-per-item live here, not in lv_mapinc.hpp: that header is
// also compiled for the host by tests/emu)
// Stable sort of a small batch's (box key, input index) pairs by ONE workgroup (bitonic network in LDS over the composite
// (key, index): the order a stable radix sort of the keys gives) — the library sort costs ~20 us in two launches at any size.
constexpr int SMALL_BATCH = 2048;
__global__ __launch_bounds__(1024) void inc_sort_small_kernel(const uint64_t* __restrict__ keys, const uint32_t* __restrict__ idx, uint32_t k,
                                                              uint64_t* __restrict__ keys_sorted, uint32_t* __restrict__ idx_sorted) {
    __shared__ uint64_t s_key[SMALL_BATCH];
    __shared__ uint32_t s_idx[SMALL_BATCH];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < (uint32_t)SMALL_BATCH; i += 1024) {
        s_key[i] = i < k ? keys[i] : ~0ull;
        s_idx[i] = i < k ? idx[i] : 0xFFFFFFFFu;
    }
    __syncthreads();
    lds_bitonic_sort_u64_u32<1024>(s_key, s_idx, lds_sort_len(k), (int)tid);
    for (uint32_t i = tid; i < k; i += 1024) { keys_sorted[i] = s_key[i]; idx_sorted[i] = s_idx[i]; }
}

// The front half of a small batch's insert by ONE workgroup: box keys -> stable sort -> the sequential box rule -> exclusive
// scan of the survivors -> ids / orig / box chains -> voxel groups.  Seven launches (two of them library calls) of a few
// microseconds of work each otherwise; the stages are the *_item functions of lv_mapinc.hpp, separated by workgroup barriers
// (one workgroup: a barrier also orders its global-memory traffic).
__global__ __launch_bounds__(1024) void inc_small_front_kernel(MapRW M, BoxRW Bx, int have_boxes, GroupRW G, const float4* __restrict__ newp,
                                                               uint32_t k, float len, int downsample, uint32_t* __restrict__ alive,
                                                               uint32_t* __restrict__ apos, float4* __restrict__ dead, uint32_t dead_cap,
                                                               uint32_t id_base) {
    // (round 4: the stages hand their keys, flags and positions to each other in LDS — the *_item functions take generic pointers —
    // instead of through global memory: every such hand-over was a write, a barrier and a read of ~1.5 us; `alive` and `apos`, which
    // the back half reads, leave for global memory once, at the end)
    __shared__ uint64_t s_key[SMALL_BATCH];
    __shared__ uint32_t s_idx[SMALL_BATCH];
    __shared__ uint32_t s_alive[SMALL_BATCH], s_apos[SMALL_BATCH];
    __shared__ uint32_t s_wsum[1024 / 64 + 1];
    const uint32_t tid = threadIdx.x;
    for (uint32_t i = tid; i < (uint32_t)SMALL_BATCH; i += 1024) {
        if (i < k) {
            inc_box_keys_item(M, newp, k, len, s_key, s_idx, s_alive, downsample, 1, i);
        } else {
            s_key[i] = ~0ull;
            s_idx[i] = 0xFFFFFFFFu;
            s_alive[i] = 0u;
        }
    }
    __syncthreads();
    if (downsample) {
        lds_bitonic_sort_u64_u32<1024>(s_key, s_idx, lds_sort_len(k), (int)tid);   // (ends with a barrier of its own)
        for (uint32_t i = tid; i < k; i += 1024) inc_box_rule_item(Bx, M.orig, newp, s_key, s_idx, k, s_alive, dead, dead_cap, M.cnt, i);
        __syncthreads();
    }
    {   // exclusive scan of alive[0 .. k) (k <= 2048: two elements per thread)
        const uint32_t i0 = 2u * tid, i1 = i0 + 1u;
        const uint32_t a0 = s_alive[i0], a1 = s_alive[i1];
        uint32_t incl = a0 + a1;
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const uint32_t v = __shfl_up(incl, o);
            if ((tid & 63u) >= (uint32_t)o) incl += v;
        }
        if ((tid & 63u) == 63u) s_wsum[tid >> 6] = incl;
        __syncthreads();
        if (tid == 0) {
            uint32_t run = 0;
            for (int w = 0; w < 1024 / 64; ++w) { const uint32_t v = s_wsum[w]; s_wsum[w] = run; run += v; }
        }
        __syncthreads();
        const uint32_t excl = s_wsum[tid >> 6] + incl - (a0 + a1);
        s_apos[i0] = excl;
        s_apos[i1] = excl + a0;
        if (i0 < k) { alive[i0] = a0; apos[i0] = excl; }
        if (i1 < k) { alive[i1] = a1; apos[i1] = excl + a0; }
    }
    __syncthreads();
    for (uint32_t j = tid; j < k; j += 1024) inc_commit_points_item(M, Bx, have_boxes, newp, s_alive, s_apos, k, id_base, j);
    for (uint32_t t = tid; t < k * (uint32_t)REPL_LEVELS; t += 1024) inc_group_item(M, G, newp, s_alive, k, t);
}

// ---- the back half of a small batch's insert in four launches instead of eight ---------------------------------------------------
// Stages that do not depend on each other share a launch (the workgroup index picks the stage; the *_item functions are the
// stand-alone kernels' bodies): the occupants that lost are tombstoned (kill) while the voxel groups find / create their
// target slots (register: the probes of kill look for keys that exist, concurrent inserts of other keys do not disturb them);
// the listed runs move (relocate: entries [0, count) of a run) while the new entries go to the tails behind them (fill); the
// owners close their targets (commit: table counts, pending) while the new bucket entries go to their ranked places (place:
// coordinates and ids inside the tail).  reserve and rank need everything before them: own launches.
__global__ __launch_bounds__(256) void inc_kill_register_kernel(MapRW M, GroupRW G, const uint32_t* __restrict__ alive, uint32_t k,
                                                                const float4* __restrict__ dead, uint32_t dead_cap, uint32_t g_kill) {
    if (blockIdx.x < g_kill) inc_kill_counted_item(M, dead, dead_cap, blockIdx.x * blockDim.x + threadIdx.x, g_kill * blockDim.x);
    else LV_INC_ITEMS(REPL_LEVELS * GROUP_TARGETS, g_kill, gridDim.x - g_kill, inc_register_item(M, G, alive, k, t));
}
// (round 4: the listed runs move while every (group, target) notes where its target's batch tail lies — inc_resolve — and the new
// entries go to the tails in a launch of their own behind it)
template <int LANES>
__global__ __launch_bounds__(256) void inc_relocate_resolve_kernel(MapRW M, GroupRW G, const uint32_t* __restrict__ alive, uint32_t k,
                                                                   const uint4* __restrict__ reloc, uint32_t reloc_cap,
                                                                   const uint32_t* __restrict__ n_reloc, uint32_t g_rel) {
    if (blockIdx.x < g_rel) inc_relocate_item<LANES>(M, reloc, reloc_cap, n_reloc, blockIdx.x * blockDim.x + threadIdx.x, g_rel * blockDim.x);
    else LV_INC_ITEMS(REPL_LEVELS * GROUP_TARGETS, g_rel, gridDim.x - g_rel, inc_resolve_item(M, G, alive, k, t));
}
__global__ __launch_bounds__(256) void inc_place_commit_kernel(MapRW M, GroupRW G, const float4* __restrict__ newp,
                                                               const uint32_t* __restrict__ alive, const uint32_t* __restrict__ apos, uint32_t k,
                                                               uint32_t id_base, const uint32_t* __restrict__ rank, uint32_t g_place) {
    if (blockIdx.x < g_place) LV_INC_ITEMS(27 * SORTED_LEVELS, 0u, g_place, inc_place_item(M, G, newp, alive, apos, k, id_base, rank, t))
    else LV_INC_ITEMS(REPL_LEVELS * GROUP_TARGETS, g_place, gridDim.x - g_place, inc_commit_item(M, G, alive, k, t))
}

// ---- lv_map_evict_box without a per-point search ---------------------------------------------------------------------------------
// Every evicted point lives in 81 runs (27 neighbourhood buckets on three levels + its voxel list).  Finding them point by point
// (inc_kill_kernel: a table probe and, on the sorted levels, a binary search over the run's ids per (point, run)) cost 8.3 ms per
// million evicted points.  The box test can be applied to the RUNS instead: a run of table `ti` holds only points of the voxel
// block around its voxel (27 voxels of its level; the voxel itself for the lists), so its bounding box — widened by one cell
// against rounding — is either entirely in the region that stays (nothing to do), entirely in the region that goes (the run's
// count drops to zero: its space is reused by later inserts, and the next sweep skips it), or cut by the box: only then are its
// entries tested one by one, with the predicate of inc_evict_mark_kernel on the very same coordinates.  One pass over the four
// tables, 64 slots per wavefront step; a cut run is walked by the whole wavefront.
__global__ __launch_bounds__(256) void inc_evict_mark_kernel(float4* __restrict__ orig, uint32_t n_ids, float lx, float ly, float lz, float hx, float hy,
                                                             float hz, int keep_inside, MapCounters* cnt) {
    // (grid-stride, ONE atomic per workgroup: a count per point — or per wavefront — on one address serialises in the L2's
    // atomic unit: 0.55 ms for 3.4 M evicted points, more than the sweep of every table)
    __shared__ uint32_t s_n;
    if (threadIdx.x == 0) s_n = 0u;
    __syncthreads();
    uint32_t mine = 0;
    for (uint32_t id = blockIdx.x * blockDim.x + threadIdx.x; id < n_ids; id += gridDim.x * blockDim.x) {
        const float4 p = orig[id];
        if (!pt_alive(p)) continue;
        const bool inside = p.x >= lx && p.x <= hx && p.y >= ly && p.y <= hy && p.z >= lz && p.z <= hz;
        if (inside != (keep_inside != 0)) { orig[id].x = pos_inf(); ++mine; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o);
    if ((threadIdx.x & 63u) == 0u && mine) atomicAdd(&s_n, mine);
    __syncthreads();
    if (threadIdx.x == 0 && s_n) atomicAdd(&cnt->n_dead, s_n);
}
__global__ __launch_bounds__(256) void inc_evict_sweep_kernel(MapRW M, int ti, float lx, float ly, float lz, float hx, float hy, float hz,
                                                              int keep_inside) {
    const LevelRW& L = M.lv[ti];
    const uint32_t n_slots = L.mask + 1u;
    const uint32_t lane = threadIdx.x & 63u;
    const uint32_t wave = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, n_waves = (gridDim.x * blockDim.x) >> 6;
    const int lvl = ti == CELL_SLOT ? CELL_LEVEL : ti;
    const int reach = ti == CELL_SLOT ? 0 : 1;
    const float cell = 1.0f / M.inv_cell;
    const float blo[3] = {lx, ly, lz}, bhi[3] = {hx, hy, hz};
    for (uint32_t s0 = wave * 64u; s0 < n_slots; s0 += n_waves * 64u) {
        const uint32_t slot = s0 + lane;
        uint4 e = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0u, 0u);
        if (slot < n_slots) e = L.table[slot];
        const uint64_t key = entry_key(e);
        int action = 0;   // 0: untouched, 1: everything goes, 2: cut by the box
        if (slot < n_slots && key != EMPTY_KEY && e.w > 0u) {
            const int v[3] = {(int)(key & 0x1fffff), (int)((key >> 21) & 0x1fffff), (int)((key >> 42) & 0x1fffff)};
            bool all_in = true, disjoint = false, clamped = false;
#pragma unroll
            for (int a = 0; a < 3; ++a) {
                const int c0 = ((v[a] - reach) << lvl) - CELL_OFFSET - 1, c1 = ((v[a] + 1 + reach) << lvl) - CELL_OFFSET + 1;
                clamped = clamped || c0 <= -1047990 || c1 >= 1047990;   // (cell_coord clamps: the outermost cells hold points from anywhere beyond)
                const float w0 = M.origin[a] + (float)c0 * cell, w1 = M.origin[a] + (float)c1 * cell;
                all_in = all_in && w0 >= blo[a] && w1 <= bhi[a];
                disjoint = disjoint || w1 < blo[a] || w0 > bhi[a];
            }
            const bool none_goes = keep_inside ? all_in : disjoint, all_go = keep_inside ? disjoint : all_in;
            action = clamped ? 2 : (none_goes ? 0 : (all_go ? 1 : 2));
        }
        if (action == 1) {
            L.table[slot].w = 0u;
            L.aux[slot].dead = 0u;   // (an empty run carries no tombstones)
            if (ti < REPL_LEVELS) inc_break_group(M, key);   // (the entries stay as they are: the group's region is no longer a valid candidate array)
        }
        unsigned long long cut = __ballot(action == 2);
        while (cut) {
            const int src = __ffsll((long long)cut) - 1;
            cut &= cut - 1;
            const uint32_t start = (uint32_t)__shfl((int)e.z, src), count = (uint32_t)__shfl((int)e.w, src);
            bool survivor = false;
            for (uint32_t i = lane; i < count; i += 64u) {
                const size_t at = (size_t)start + i;
                float x, y, z;
                if (ti < SORTED_LEVELS) { x = M.bxyz[ti][at * 3 + 0]; y = M.bxyz[ti][at * 3 + 1]; z = M.bxyz[ti][at * 3 + 2]; }
                else { const float4 p = M.cell4[at]; x = p.x; y = p.y; z = p.z; }
                const bool inside = x >= lx && x <= hx && y >= ly && y <= hy && z >= lz && z <= hz;
                const bool alive = x < pos_inf() && x > -pos_inf();
                if (inside != (keep_inside != 0)) {   // (an entry that is dead already stays dead either way)
                    if (alive) {
                        if (ti < SORTED_LEVELS) { M.bxyz[ti][at * 3 + 0] = pos_inf(); atomicAdd(&L.aux[s0 + (uint32_t)src].dead, 1u); }
                        else M.cell4[at].x = pos_inf();
                    }
                } else {
                    survivor = survivor || alive;
                }
            }
            // a run without a living entry left is dropped like one that lies outside altogether: later sweeps (a rolling window
            // cuts the same neighbourhood again and again) and searches skip it, later inserts reuse its space
            if (!__any(survivor) && lane == 0u) { L.table[s0 + (uint32_t)src].w = 0u; L.aux[s0 + (uint32_t)src].dead = 0u; }
        }
    }
}

// the outcome of an insert for the host (MapStore::settle), as a note
__global__ void inc_post_counters_kernel(const MapCounters* __restrict__ cnt, unsigned long long* __restrict__ note, uint32_t seq) {
    if (threadIdx.x != 0) return;
    note_post(note + 0, seq, cnt->n_new);
    note_post(note + 1, seq, cnt->n_dead);
    note_post(note + 2, seq, cnt->dropped);
    note_post(note + 3, seq, cnt->overflow);
    // what is left of the bucket pool's free part (entries): the host starts a re-linearisation BEFORE a batch runs out of room
    uint32_t left = 0;
    for (int a = 0; a < N_ARENAS; ++a) {
        const uint32_t cur = cnt->arena_cur[0][a], end = cnt->arena_end[0][a];
        left += cur < end ? end - cur : 0u;
    }
    note_post(note + 4, seq, left);
}

// the scratch tables of a batch back to empty (0xFF) and its group counters to zero: one launch instead of four fills
__global__ void inc_clear_groups_kernel(GroupRW G, uint32_t* __restrict__ gcnt, MapCounters* __restrict__ cnt) {
    const uint32_t t = blockIdx.x * blockDim.x + threadIdx.x;
    if (t < 4u * LIST_SHARDS) gcnt[t] = 0u;   // (the sharded cursors of the batch's work lists: runs that move, groups broken up, runs compacted, their staging area)
    if (t == 0u && cnt) { cnt->n_new = 0u; cnt->n_dead = 0u; cnt->overflow = 0u; cnt->dropped = 0u; }   // (reset_batch_counters)
    const uint32_t l = t / G.size, e = t % G.size;
    if (l < (uint32_t)REPL_LEVELS) G.table[l][e] = make_uint4(0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu);
}

static int reset_batch_counters(MapStore& S, hipStream_t stream) {
    // n_new .. dropped are contiguous (MapCounters)
    LV_HIP(hipMemsetAsync(&S.d_cnt->n_new, 0, offsetof(MapCounters, box_slots_used) - offsetof(MapCounters, n_new), stream));
    return LV_OK;
}

int MapStore::ensure_counters() {
    if (d_cnt) return LV_OK;
    LV_HIP(hipMalloc(&d_cnt, sizeof(MapCounters)));
    LV_HIP(hipHostMalloc((void**)&h_cnt, sizeof(MapCounters), hipHostMallocDefault));
    std::memset(h_cnt, 0, sizeof(MapCounters));
    LV_HIP(hipMemset(d_cnt, 0, sizeof(MapCounters)));
    return LV_OK;
}

int MapStore::add_staged(hipStream_t stream, uint32_t k, int downsample, float box_length, bool build_if_empty) {
    if (k == 0) return LV_OK;
    int rc = settle(stream);
    if (rc) return rc;
    if ((!built || m == 0) && downsample && !build_if_empty) {
        // KD_TREE::Add_Points(points, true) into an EMPTY map: the box rule among the new points themselves (every box
        // keeps, of its points in input order, the one the sequential rule leaves), then a build from the survivors
        n_ids = 0;
        m = 0;
        built = false;
        rc = reserve(k);
        if (rc) return rc;
        rc = ensure_counters();
        if (rc) return rc;
        rc = reset_batch_counters(*this, stream);
        if (rc) return rc;
        have_boxes = false;
        rc = ensure_boxes(stream, box_length);
        if (rc) return rc;
        MapRW M{};
        M.orig = d_orig;
        M.cnt = d_cnt;
        const BoxRW Bx{d_box, d_box_next, box_size - 1, (uint32_t)(64 - log2u(box_size)), (uint32_t)((uint64_t)box_size * 6 / 10), box_length};
        const uint32_t gk = (k + 255) / 256;
        hipLaunchKernelGGL(inc_box_keys_kernel, dim3(gk), dim3(256), 0, stream, M, d_new, k, box_length, d_nkeys, d_nidx, d_nalive, 1, 0);
        size_t tmp = ntmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_ntmp, tmp, d_nkeys, d_nkeys_sorted, d_nidx, d_nidx_sorted, (int)k, 0, 63,
                                                               stream));
        hipLaunchKernelGGL(inc_box_rule_kernel, dim3(gk), dim3(256), 0, stream, Bx, d_orig, d_new, d_nkeys_sorted, d_nidx_sorted, k,
                           d_nalive, d_dead, (uint32_t)dead_cap, d_cnt);
        tmp = ntmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ntmp, tmp, d_nalive, d_napos, (int)k, stream));
        hipLaunchKernelGGL(inc_commit_points_kernel, dim3(gk), dim3(256), 0, stream, M, Bx, 0, d_new, d_nalive, d_napos, k, 0u);
        LV_HIP(hipGetLastError());
        LV_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(MapCounters), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        n_ids = h_cnt->n_new;
        dropped_total += h_cnt->dropped;
        origin_set = false;
        return rebuild(stream);
    }
    if (!built || m == 0) {
        // Mapper::add on an empty map builds it from the points as they are (Mapper.cpp:22-27): keep the finite ones
        n_ids = 0;
        m = 0;
        rc = reserve(k);
        if (rc) return rc;
        rc = ensure_alive_scratch();
        if (rc) return rc;
        const uint32_t grid = (k + 255) / 256;
        hipLaunchKernelGGL(staged_ok_kernel, dim3(grid), dim3(256), 0, stream, d_new, k, d_alive);
        size_t tmp = ascan_tmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ascan_tmp, tmp, d_alive, d_apos, (int)k, stream));
        hipLaunchKernelGGL(inc_compact_kernel, dim3(grid), dim3(256), 0, stream, d_new, d_alive, d_apos, k, d_orig);
        uint32_t last_pos = 0, last_ok = 0;
        LV_HIP(hipMemcpyAsync(&last_pos, d_apos + (k - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipMemcpyAsync(&last_ok, d_alive + (k - 1), sizeof(uint32_t), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        n_ids = last_pos + last_ok;
        dropped_total += k - n_ids;
        origin_set = false;
        return rebuild(stream);
    }
    if (needs_relinearise(k)) {
        rc = relinearise(stream);
        if (rc) return rc;
    }
    rc = reserve((size_t)n_ids + k);
    if (rc) return rc;
    if (have_boxes && box_next_cap < capacity) have_boxes = false;
    // the batch counters go back to zero in the launch that clears the group tables — unless the boxes are about to be built
    // (the first down-sampling insert): their build may raise `overflow`, so the reset has to come before it
    const bool reset_early = downsample && !have_boxes;
    if (reset_early) {
        rc = reset_batch_counters(*this, stream);
        if (rc) return rc;
    }
    MapRW M = rw();
    M.broken = d_broken;
    M.broken_cap = broken_cap;
    M.n_broken = d_gcnt + LIST_SHARDS;   // (zeroed with the batch's other counters: inc_clear_groups_kernel)
    M.comp = d_comp;
    M.comp_cap = comp_cap;
    M.n_comp = d_gcnt + 2 * LIST_SHARDS;   // (+ the staging cursors behind them)
    M.cstage = d_cstage;
    M.cnew = d_cnew;
    M.cstage_cap = cstage_cap;
    BoxRW Bx{};
    if (downsample) {
        rc = ensure_boxes(stream, box_length);
        if (rc) return rc;
    }
    if (have_boxes)
        Bx = BoxRW{d_box, d_box_next, box_size - 1, (uint32_t)(64 - log2u(box_size)), (uint32_t)((uint64_t)box_size * 6 / 10), box_length};
    const int B = 256;
    const uint32_t gk = (k + B - 1) / B;
    // voxel-group tables of the batch (cleared first: the small batch's front kernel fills them)
    GroupRW G{};
    for (int l = 0; l < REPL_LEVELS; ++l) {
        G.table[l] = d_gtab[l];
        G.gbase[l] = d_gbase[l];
        G.gslot[l] = d_gslot[l];
        G.gdst[l] = d_gdst[l];
    }
    G.mask = gtab_size - 1;
    G.shift = (uint32_t)(64 - log2u(gtab_size));
    G.size = gtab_size;
    G.prank = d_prank;
    G.pslot = d_pslot;
    // large down-sampling batches walk their survivors in the order of the box sort (Morton): see inc_box_key
    const bool listed = downsample && !(small_front && k <= (uint32_t)SMALL_BATCH) && k > (uint32_t)SMALL_BATCH && surv_list;
    G.surv = listed ? d_nsurv : nullptr;
    G.n_live = &d_cnt->n_new;
    hipLaunchKernelGGL(inc_clear_groups_kernel, dim3((uint32_t)(((uint64_t)gtab_size * REPL_LEVELS + B - 1) / B)), dim3(B), 0, stream, G,
                       d_gcnt, reset_early ? nullptr : d_cnt);
    const bool fused_front = small_front && k <= (uint32_t)SMALL_BATCH;
    uint32_t n_dead = 0;
    if (fused_front) {
        hipLaunchKernelGGL(inc_small_front_kernel, dim3(1), dim3(1024), 0, stream, M, Bx, have_boxes ? 1 : 0, G, d_new, k, box_length,
                           downsample, d_nalive, d_napos, d_dead, (uint32_t)dead_cap, n_ids);
    } else {
    hipLaunchKernelGGL(inc_box_keys_kernel, dim3(gk), dim3(B), 0, stream, M, d_new, k, box_length, d_nkeys, d_nidx, d_nalive, downsample, 1);
    if (downsample) {
        if (k <= (uint32_t)SMALL_BATCH) {
            hipLaunchKernelGGL(inc_sort_small_kernel, dim3(1), dim3(1024), 0, stream, d_nkeys, d_nidx, k, d_nkeys_sorted, d_nidx_sorted);
        } else {
            size_t tmp = ntmp_bytes;
            LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_ntmp, tmp, d_nkeys, d_nkeys_sorted, d_nidx, d_nidx_sorted, (int)k, 0,
                                                                   63, stream));
        }
        hipLaunchKernelGGL(inc_box_rule_kernel, dim3(gk), dim3(B), 0, stream, Bx, d_orig, d_new, d_nkeys_sorted, d_nidx_sorted, k,
                           d_nalive, d_dead, (uint32_t)dead_cap, d_cnt);
    }
    {
        size_t tmp = ntmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ntmp, tmp, d_nalive, d_napos, (int)k, stream));
    }
    hipLaunchKernelGGL(inc_commit_points_kernel, dim3(gk), dim3(B), 0, stream, M, Bx, have_boxes ? 1 : 0, d_new, d_nalive, d_napos, k,
                       n_ids);
    if (listed) {   // the survivor list in sorted-batch order (flags -> scan -> scatter: three small launches)
        hipLaunchKernelGGL(inc_surv_flag_kernel, dim3(gk), dim3(B), 0, stream, d_nidx_sorted, d_nalive, k, d_nsflag);
        size_t tmp = ntmp_bytes;
        LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ntmp, tmp, d_nsflag, d_nspos, (int)k, stream));
        hipLaunchKernelGGL(inc_surv_list_kernel, dim3(gk), dim3(B), 0, stream, d_nidx_sorted, d_nsflag, d_nspos, k, d_nsurv);
    }
    // voxel groups of the survivors on every level
    hipLaunchKernelGGL(inc_group_kernel, dim3((uint32_t)(((uint64_t)k * REPL_LEVELS + B - 1) / B)), dim3(B), 0, stream, M, G, d_new,
                       d_nalive, k);
    }   // !fused_front
    // (round 6: every batch leaves the length of its dead list on the device — the tombstone pass walks it in strides beside the
    // registration pass — instead of fetching it in the middle of the insert: one host round trip and one launch less per insert;
    // the append passes then cover every point of the batch, the ones that did not survive leaving at once)
    const bool counted_kill = downsample != 0;
    if (!counted_kill && downsample) {
        // the counters of the front half (survivors, occupants that lost) size the launches that follow: fetched as a NOTE (a
        // one-thread kernel posts them into pinned memory, the host polls: lv_note.hpp) instead of a copy + stream synchronise,
        // whose wake-up leaves the device idle for ~30 us in the middle of the insert (round 4)
        LV_HIP(note_alloc(notes));
        const uint32_t mid_seq = notes.next();
        hipLaunchKernelGGL(inc_post_counters_kernel, dim3(1), dim3(64), 0, stream, d_cnt, notes.d, mid_seq);
        LV_HIP(hipGetLastError());
        uint32_t mv[4] = {0, 0, 0, 0};   // n_new, n_dead, dropped, overflow
        if (!note_wait(notes, 0, 4, mid_seq, mv, stream)) { set_error("map insert: the front half's counters never arrived"); return LV_EHIP; }
        h_cnt->n_new = mv[0];
        h_cnt->n_dead = mv[1];
        n_dead = h_cnt->n_dead < dead_cap ? h_cnt->n_dead : (uint32_t)dead_cap;
        rc = kill_dead_list(stream, n_dead);
        if (rc) return rc;
    }
    // work items of the append passes: with a survivor list (and the counters just read) one per SURVIVOR and slot, else one per
    // point of the batch and slot (the dead leave at once)
    // grids of the per-survivor passes: an ESTIMATE of the survivors — half again what the previous batch left — where the batch
    // walks a survivor list (the kernels cover whatever lies beyond their grid in strides: LV_INC_ITEMS), else every point
    uint64_t kk = (uint64_t)k;
    if (listed && counted_kill && have_last_new) {
        const uint64_t est = (uint64_t)last_new + (uint64_t)last_new / 2u + 2048u;
        kk = est < (uint64_t)k ? est : (uint64_t)k;
    }
    const uint64_t t_grp = kk * REPL_LEVELS * GROUP_TARGETS;
    const uint64_t t_all = kk * INC_SLOTS_PER_POINT, t_rep = kk * 27 * SORTED_LEVELS;
    const uint32_t g_grp = (uint32_t)((t_grp + B - 1) / B), g_all = (uint32_t)((t_all + B - 1) / B), g_rep = (uint32_t)((t_rep + B - 1) / B);
    const uint64_t t_rel = (uint64_t)(reloc_cap < (1u << 18) ? reloc_cap : (1u << 18)) * RELOC_LANES;   // runs moved per batch (more: re-linearise)
    // at most one run per (group, target) of this batch can be listed; 2048 workgroups walk longer lists in strides
    const uint64_t t_need = t_grp * RELOC_LANES < t_rel ? t_grp * RELOC_LANES : t_rel;
    const uint64_t g_need = (t_need + B - 1) / B;
    const uint32_t g_rel = ((uint32_t)(g_need < 2048 ? g_need : 2048) + 15u) & ~15u;   // (a multiple of 16 workgroups: 64 x n runs at a time, one per list shard)
    const uint32_t g_cmp = k <= (uint32_t)SMALL_BATCH ? 64u : 1024u;   // in-place compactions: how many is only known on the device (grid-stride)
    if (merged_back) {   // (see inc_kill_register_kernel)
        const uint32_t g_kill = counted_kill ? (k <= (uint32_t)SMALL_BATCH ? 256u : 2048u) : 0u;   // the occupants that lost: how many is only known on the device
        hipLaunchKernelGGL(inc_kill_register_kernel, dim3(g_kill + g_grp), dim3(B), 0, stream, M, G, d_nalive, k, d_dead, (uint32_t)dead_cap, g_kill);
        hipLaunchKernelGGL(inc_reserve_kernel, dim3(g_grp), dim3(B), 0, stream, M, G, d_nalive, k, d_reloc, (uint32_t)(t_rel / RELOC_LANES), d_gcnt);
        // runs that only their deleted entries made too long are compacted where they lie: staged here, written back below
        hipLaunchKernelGGL(inc_compact_gather_kernel, dim3(g_cmp), dim3(B), 0, stream, M);
        if (k <= (uint32_t)SMALL_BATCH)   // few runs move: a workgroup each
            hipLaunchKernelGGL(inc_relocate_resolve_kernel<RELOC_LANES_SMALL>, dim3(g_rel + g_grp), dim3(B), 0, stream, M, G, d_nalive, k, d_reloc,
                               (uint32_t)(t_rel / RELOC_LANES), d_gcnt, g_rel);
        else
            hipLaunchKernelGGL(inc_relocate_resolve_kernel<RELOC_LANES>, dim3(g_rel + g_grp), dim3(B), 0, stream, M, G, d_nalive, k, d_reloc,
                               (uint32_t)(t_rel / RELOC_LANES), d_gcnt, g_rel);
        hipLaunchKernelGGL(inc_compact_scatter_kernel, dim3(g_cmp), dim3(B), 0, stream, M);
        hipLaunchKernelGGL(inc_fill_kernel, dim3(g_all), dim3(B), 0, stream, M, G, d_new, d_nalive, d_napos, k, n_ids);
        hipLaunchKernelGGL(inc_rank_kernel, dim3(g_rep), dim3(B), 0, stream, M, G, d_nalive, d_napos, k, n_ids, d_rank);
        hipLaunchKernelGGL(inc_place_commit_kernel, dim3(g_rep + g_grp), dim3(B), 0, stream, M, G, d_new, d_nalive, d_napos, k, n_ids, d_rank, g_rep);
    } else {
    if (counted_kill) hipLaunchKernelGGL(inc_kill_counted_kernel, dim3(k <= (uint32_t)SMALL_BATCH ? 256 : 2048), dim3(B), 0, stream, M, d_dead, (uint32_t)dead_cap);
    hipLaunchKernelGGL(inc_register_kernel, dim3(g_grp), dim3(B), 0, stream, M, G, d_nalive, k);
    hipLaunchKernelGGL(inc_reserve_kernel, dim3(g_grp), dim3(B), 0, stream, M, G, d_nalive, k, d_reloc, (uint32_t)(t_rel / RELOC_LANES), d_gcnt);
    hipLaunchKernelGGL(inc_compact_gather_kernel, dim3(g_cmp), dim3(B), 0, stream, M);
    hipLaunchKernelGGL(inc_relocate_kernel, dim3(g_rel), dim3(B), 0, stream, M, d_reloc, (uint32_t)(t_rel / RELOC_LANES), d_gcnt);
    hipLaunchKernelGGL(inc_resolve_kernel, dim3(g_grp), dim3(B), 0, stream, M, G, d_nalive, k);
    hipLaunchKernelGGL(inc_compact_scatter_kernel, dim3(g_cmp), dim3(B), 0, stream, M);
    hipLaunchKernelGGL(inc_fill_kernel, dim3(g_all), dim3(B), 0, stream, M, G, d_new, d_nalive, d_napos, k, n_ids);
    hipLaunchKernelGGL(inc_rank_kernel, dim3(g_rep), dim3(B), 0, stream, M, G, d_nalive, d_napos, k, n_ids, d_rank);
    hipLaunchKernelGGL(inc_place_kernel, dim3(g_rep), dim3(B), 0, stream, M, G, d_new, d_nalive, d_napos, k, n_ids, d_rank);
    hipLaunchKernelGGL(inc_commit_kernel, dim3(g_grp), dim3(B), 0, stream, M, G, d_nalive, k);
    }
    // the tile groups this batch broke up (a run outgrew its room, a bucket appeared) are laid out again: level 1 streams a group's
    // region as one candidate array (bucket_attempt, lv_match.hip); three small launches that find nothing to do in most batches
    hipLaunchKernelGGL(inc_regroup_plan_kernel, dim3(64), dim3(B), 0, stream, M, d_regroup);
    hipLaunchKernelGGL(inc_regroup_move_kernel, dim3(512), dim3(B), 0, stream, M, (const RegroupPlan*)d_regroup);
    hipLaunchKernelGGL(inc_regroup_commit_kernel, dim3(64), dim3(B), 0, stream, M, (const RegroupPlan*)d_regroup);
    LV_HIP(hipGetLastError());
    // the insert's outcome (ids handed out, occupants that lost, overflow) is read back WITHOUT waiting for it: settle() picks
    // it up when the map's bookkeeping is needed next (the following search or insert), by which time it has long arrived
    LV_HIP(note_alloc(notes));
    counters_seq = notes.next();
    counters_stream = stream;
    hipLaunchKernelGGL(inc_post_counters_kernel, dim3(1), dim3(64), 0, stream, d_cnt, notes.d, counters_seq);
    LV_HIP(hipGetLastError());
    counters_pending = true;
    pending_counted_kill = counted_kill;
    pending_n_dead = n_dead;
    return LV_OK;
}

int MapStore::settle(hipStream_t stream) {
    if (!counters_pending) return LV_OK;
    counters_pending = false;
    // (the note is posted by the last kernel of the insert's chain, on the stream the chain ran on — the context's side stream
    // when the insert overlaps the next cycle's prediction and window: once it has arrived the whole insert has completed, so
    // whatever the caller enqueues next, on any stream, sees the finished map)
    uint32_t v[5] = {0, 0, 0, 0, 0};   // n_new, n_dead, dropped, overflow, free entries left in the bucket pool
    if (!note_wait(notes, 0, 5, counters_seq, v, counters_stream)) { set_error("map insert: the counters never arrived"); return LV_EHIP; }
    // three quarters of the pool's free part are gone (runs that moved, groups laid out again, newly mapped space): ask for a
    // re-linearisation now — in the background for a large map (lv_api.hip relin_maybe_start) — instead of meeting `overflow`
    // in the middle of a later batch, which costs that batch's work and a stop-the-world rebuild
    pool_low = (uint64_t)v[4] * 4 < (uint64_t)(pool_cap[0] > pool_base[0] ? pool_cap[0] - pool_base[0] : 0);
    uint32_t n_dead = pending_n_dead;
    if (pending_counted_kill) n_dead = v[1] < dead_cap ? v[1] : (uint32_t)dead_cap;
    n_ids += v[0];
    m += v[0];
    last_new = v[0];
    have_last_new = true;
    m -= n_dead;
    tombstones += (uint64_t)n_dead * INC_SLOTS_PER_POINT;
    dropped_total += v[2];
    ++incremental_adds;
    refresh_view();
    if (v[3]) return relinearise(stream);   // a pool / table ran full: the new points are in `orig`, rebuild around them
    return LV_OK;
}

int MapStore::evict_box(hipStream_t stream, const float lo[3], const float hi[3], int keep_inside, uint32_t* n_evicted) {
    if (n_evicted) *n_evicted = 0;
    { int rcs = settle(stream); if (rcs) return rcs; }
    if (!built || m == 0) return LV_OK;
    int rc = reset_batch_counters(*this, stream);
    if (rc) return rc;
    uint32_t n_dead = 0;
    if (sweep_evict) {   // (see inc_evict_sweep_kernel)
        const uint32_t g_mark = (n_ids + 255) / 256;
        hipLaunchKernelGGL(inc_evict_mark_kernel, dim3(g_mark < 2048u ? g_mark : 2048u), dim3(256), 0, stream, d_orig, n_ids, lo[0], lo[1], lo[2], hi[0],
                           hi[1], hi[2], keep_inside, d_cnt);
        const MapRW M = rw();
        for (int ti = 0; ti < INC_LEVELS; ++ti) {
            const uint64_t slots = (uint64_t)M.lv[ti].mask + 1u;
            const uint64_t wgs = (slots + 255) / 256;   // 64 slots per wavefront step, four wavefronts per workgroup
            hipLaunchKernelGGL(inc_evict_sweep_kernel, dim3((uint32_t)(wgs < 4096 ? wgs : 4096)), dim3(256), 0, stream, M, ti, lo[0], lo[1], lo[2],
                               hi[0], hi[1], hi[2], keep_inside);
        }
        LV_HIP(hipGetLastError());
        LV_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(MapCounters), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        n_dead = h_cnt->n_dead;
    } else {
        hipLaunchKernelGGL(inc_evict_box_kernel, dim3((n_ids + 255) / 256), dim3(256), 0, stream, d_orig, n_ids, lo[0], lo[1], lo[2], hi[0],
                           hi[1], hi[2], keep_inside, d_dead, (uint32_t)dead_cap, d_cnt);
        LV_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(MapCounters), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
        n_dead = h_cnt->n_dead;
        rc = kill_dead_list(stream, n_dead);
        if (rc) return rc;
        LV_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(MapCounters), hipMemcpyDeviceToHost, stream));
        LV_HIP(hipStreamSynchronize(stream));
    }
    m -= n_dead;
    tombstones += (uint64_t)n_dead * INC_SLOTS_PER_POINT;
    if (n_evicted) *n_evicted = n_dead;
    refresh_view();
    if (m == 0) { n_ids = 0; return rebuild(stream); }
    return LV_OK;
}

int MapStore::evict_oldest(hipStream_t stream, uint32_t n_oldest, uint32_t* n_evicted) {
    if (n_evicted) *n_evicted = 0;
    { int rcs = settle(stream); if (rcs) return rcs; }
    if (!built || m == 0 || n_oldest == 0) return LV_OK;
    if (n_oldest > m) n_oldest = m;
    int rc = ensure_alive_scratch();
    if (rc) return rc;
    rc = reset_batch_counters(*this, stream);
    if (rc) return rc;
    const uint32_t grid = (n_ids + 255) / 256;
    hipLaunchKernelGGL(inc_alive_flags_kernel, dim3(grid), dim3(256), 0, stream, d_orig, n_ids, d_alive);
    size_t tmp = ascan_tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceScan::ExclusiveSum(d_ascan_tmp, tmp, d_alive, d_apos, (int)n_ids, stream));
    hipLaunchKernelGGL(inc_evict_oldest_kernel, dim3(grid), dim3(256), 0, stream, d_orig, n_ids, d_apos, n_oldest, d_dead,
                       (uint32_t)dead_cap, d_cnt);
    LV_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(MapCounters), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));
    const uint32_t n_dead = h_cnt->n_dead;
    rc = kill_dead_list(stream, n_dead);
    if (rc) return rc;
    LV_HIP(hipMemcpyAsync(h_cnt, d_cnt, sizeof(MapCounters), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));
    m -= n_dead;
    tombstones += (uint64_t)n_dead * INC_SLOTS_PER_POINT;
    if (n_evicted) *n_evicted = n_dead;
    refresh_view();
    if (m == 0) { n_ids = 0; return rebuild(stream); }
    return LV_OK;
}

}  // namespace lv
