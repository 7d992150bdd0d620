// tests/emu/mapinc_emu.cpp — HOST EMULATION of the incremental-map kernels (limo-velo_amd/csrc/lv_mapinc.hpp).
// TEST INFRASTRUCTURE ONLY: built by tests/test_mapinc_emulation.py into tests/emu/_build/, never shipped, never
// loaded by the product.  The kernels are the product's own source, compiled for the host through the stand-in
// hip_runtime.h next to this file and executed as sequential loops; the orchestration below mirrors
// MapStore::add_staged / evict_* (lv_map.hip) step by step.  A scalar builder lays the structure out exactly as
// the GPU build does (buckets in ascending id, slack, tables), and emu_check() verifies after every operation:
//   * every bucket run: ids ascending, living entries carry the coordinates of their id, tombstones belong to dead
//     ids, the living ids are EXACTLY the living points of the bucket voxel's 3x3x3 block, count <= capacity,
//     runs disjoint and inside the pool;
//   * every voxel whose block holds a living point has a bucket; level-2 lists hold every living id exactly once;
//   * box chains reach every living id exactly once.
// The Python test compares the living points (order included) with the oracle's lvo_map_add.
#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstdio>
#include <map>
#include <set>
#include <string>
#include <vector>

emu_dim3 threadIdx, blockIdx, blockDim, gridDim;

#define LV_MAPINC_KERNELS
#include "../../limo-velo_amd/csrc/lv_mapinc.hpp"

using namespace lv;

namespace {

int g_order = 0;          // 0 forward, 1 reverse, 2 shuffled
uint64_t g_rng = 0x9E3779B97F4A7C15ull;
uint32_t rnd() { g_rng ^= g_rng << 13; g_rng ^= g_rng >> 7; g_rng ^= g_rng << 17; return (uint32_t)(g_rng >> 32); }

template <typename K, typename... A>
void launch(K kernel, uint64_t threads, A... args) {
    const uint32_t B = 256;
    const uint32_t G = (uint32_t)((threads + B - 1) / B);
    gridDim.x = G ? G : 1;
    blockDim.x = B;
    const uint64_t total = (uint64_t)gridDim.x * B;
    std::vector<uint32_t> perm;
    if (g_order == 2) {
        perm.resize(total);
        for (uint64_t i = 0; i < total; ++i) perm[i] = (uint32_t)i;
        for (uint64_t i = total; i > 1; --i) std::swap(perm[i - 1], perm[rnd() % i]);
    }
    for (uint64_t s = 0; s < total; ++s) {
        const uint64_t t = g_order == 0 ? s : g_order == 1 ? total - 1 - s : perm[s];
        blockIdx.x = (uint32_t)(t / B);
        threadIdx.x = (uint32_t)(t % B);
        kernel(args...);
    }
}

uint32_t next_pow2(uint64_t v) { uint32_t p = 64; while (p < v) p <<= 1; return p; }
int log2u(uint32_t s) { int lg = 0; while ((1u << lg) < s) ++lg; return lg; }
uint32_t run_capacity(uint32_t c) { return c + (c / 2u > 8u ? c / 2u : 8u); }

struct Emu {
    float cell = 0.5f, box_len = 0.2f;
    float origin[3] = {0, 0, 0};
    std::vector<float4> orig;           // capacity-sized
    uint32_t n_ids = 0, m = 0;
    std::vector<uint4> table[INC_LEVELS];
    std::vector<SlotAux> aux[INC_LEVELS];
    std::vector<float> bxyz[SORTED_LEVELS];
    std::vector<uint32_t> bidx[SORTED_LEVELS];
    std::vector<uint16_t> backpos;      // [id * 27 + c]: position of id inside the bucket of its neighbour c
    std::vector<uint4> gtable;          // tile groups: level-1 voxel -> {start, extent} of its runs' region
    std::vector<uint32_t> broken;       // groups the batch in flight broke up
    std::vector<RegroupPlan> plans;
    uint32_t n_broken[LIST_SHARDS] = {};
    uint64_t regrouped = 0, compacted = 0;
    std::vector<uint4> comp;            // runs compacted in place by the batch in flight + their staging area
    std::vector<float4> cstage;
    std::vector<uint32_t> cnew;
    uint32_t n_comp[2 * LIST_SHARDS] = {};
    std::vector<uint32_t> cellpos;      // [id]
    std::vector<float4> cell4;
    uint32_t pool_cap[INC_LEVELS] = {};
    MapCounters cnt{};
    std::vector<uint4> box;
    std::vector<uint32_t> box_next;
    bool have_boxes = false, built = false;
    uint32_t pool_reserve = 4096;       // small on purpose: relocation / overflow paths get exercised
    uint64_t relinearisations = 0, relocations = 0, n_killed = 0;
    uint32_t batch_no = 0;
    std::string err;

    void cell_of(const float4& p, int c[3]) const {
        c[0] = cell_coord(p.x, origin[0], 1.0f / cell);
        c[1] = cell_coord(p.y, origin[1], 1.0f / cell);
        c[2] = cell_coord(p.z, origin[2], 1.0f / cell);
    }
    static void table_put(std::vector<uint4>& t, uint64_t key, uint32_t start, uint32_t count, uint32_t& slot_out) {
        const uint32_t mask = (uint32_t)t.size() - 1, shift = (uint32_t)(64 - log2u((uint32_t)t.size()));
        uint32_t slot = hash_cell(key, shift) & mask;
        while (entry_key(t[slot]) != EMPTY_KEY) slot = (slot + 1) & mask;
        t[slot] = uint4{(uint32_t)key, (uint32_t)(key >> 32), start, count};
        slot_out = slot;
    }
    void set_arenas(int l, uint32_t used) {
        const uint64_t free_entries = (uint64_t)pool_cap[l] - used;
        for (int a = 0; a < N_ARENAS; ++a) {
            cnt.arena_cur[l][a] = used + (uint32_t)((free_entries * a) / N_ARENAS);
            cnt.arena_end[l][a] = used + (uint32_t)((free_entries * (a + 1)) / N_ARENAS);
        }
    }
    MapRW rw() {
        MapRW M{};
        M.orig = orig.data();
        for (int l = 0; l < INC_LEVELS; ++l) {
            M.lv[l].table = table[l].data();
            M.lv[l].aux = aux[l].data();
            M.lv[l].mask = (uint32_t)table[l].size() - 1;
            M.lv[l].shift = (uint32_t)(64 - log2u((uint32_t)table[l].size()));
            M.lv[l].slot_limit = (uint32_t)(table[l].size() * (l < REPL_LEVELS ? 7 : 6) / 10);
            M.lv[l].pool_cap = pool_cap[l];
        }
        for (int l = 0; l < SORTED_LEVELS; ++l) { M.bxyz[l] = bxyz[l].data(); M.bidx[l] = bidx[l].data(); }
        M.backpos = backpos.data();
        M.gtable = gtable.data();
        M.gmask = (uint32_t)gtable.size() - 1;
        M.gshift = (uint32_t)(64 - log2u((uint32_t)gtable.size()));
        M.gslot_limit = (uint32_t)(gtable.size() * 7 / 10);
        M.cellpos = cellpos.data();
        M.cell4 = cell4.data();
        for (int a = 0; a < 3; ++a) M.origin[a] = origin[a];
        M.inv_cell = 1.0f / cell;
        M.cnt = &cnt;
        return M;
    }
    BoxRW bx() {
        return BoxRW{box.data(), box_next.data(), (uint32_t)box.size() - 1, (uint32_t)(64 - log2u((uint32_t)box.size())),
                     (uint32_t)(box.size() * 6 / 10), box_len};
    }

    // scalar twin of MapStore::rebuild: ids 0 .. n_ids are all living
    void rebuild(bool keep_origin) {
        cnt = MapCounters{};
        have_boxes = false;
        built = false;
        m = n_ids;
        if (m == 0) return;
        if (!keep_origin) {
            float lo[3] = {INFINITY, INFINITY, INFINITY}, hi[3] = {-INFINITY, -INFINITY, -INFINITY};
            for (uint32_t i = 0; i < n_ids; ++i) {
                const float c[3] = {orig[i].x, orig[i].y, orig[i].z};
                for (int a = 0; a < 3; ++a) { lo[a] = fminf(lo[a], c[a]); hi[a] = fmaxf(hi[a], c[a]); }
            }
            for (int a = 0; a < 3; ++a) origin[a] = floorf(0.5f * (lo[a] + hi[a]) / cell) * cell;
        }
        for (int l = 0; l < REPL_LEVELS; ++l) {
            std::map<uint64_t, std::vector<uint32_t>> buckets;   // bucket voxel -> ids of its block, ascending
            std::set<uint64_t> occ;
            for (uint32_t id = 0; id < n_ids; ++id) {
                int c[3];
                cell_of(orig[id], c);
                occ.insert(pack_cell((uint32_t)(c[0] >> l), (uint32_t)(c[1] >> l), (uint32_t)(c[2] >> l)));
                for (int n = 0; n < 27; ++n) {
                    const int dz = n / 9 - 1, dy = (n / 3) % 3 - 1, dx = n % 3 - 1;
                    const uint32_t nx = (uint32_t)((c[0] >> l) + dx), ny = (uint32_t)((c[1] >> l) + dy), nz = (uint32_t)((c[2] >> l) + dz);
                    if (nx >= (1u << 21) || ny >= (1u << 21) || nz >= (1u << 21)) continue;
                    buckets[pack_cell(nx, ny, nz)].push_back(id);
                }
            }
            uint32_t size = next_pow2((uint64_t)occ.size() * 16);
            while (buckets.size() > size / 2) size *= 2;
            table[l].assign(size, uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
            aux[l].assign(size, SlotAux{0, 0, 0, 0});
            uint64_t total = 0;
            for (auto& kv : buckets) total += run_capacity((uint32_t)kv.second.size());
            pool_cap[l] = (uint32_t)(total + total / 4 + pool_reserve);
            bxyz[l].assign((size_t)pool_cap[l] * 3 + 4, 0.f);
            bidx[l].assign(pool_cap[l], 0u);
            backpos.assign(orig.size() * 27, (uint16_t)0xFFFEu);
            cellpos.assign(orig.size(), 0xFFFFFFFFu);
            // the runs are laid out group by group (the eight buckets that tile a level-1 block side by side), the slack reads +inf
            std::map<uint64_t, std::vector<uint64_t>> groups;
            for (auto& kv : buckets) {
                uint32_t vx, vy, vz; int r;
                tile_group_of((uint32_t)(kv.first & 0x1fffff), (uint32_t)((kv.first >> 21) & 0x1fffff), (uint32_t)((kv.first >> 42) & 0x1fffff), vx, vy, vz, r);
                groups[pack_cell(vx, vy, vz)].push_back(kv.first);
            }
            gtable.assign(next_pow2((uint64_t)buckets.size()), uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
            cnt.gslots_used = (uint32_t)groups.size();
            for (size_t i = 0; i < (size_t)pool_cap[l] * 3; ++i) bxyz[l][i] = INFINITY;
            uint32_t off = 0;
            for (auto& gk : groups) {
              const uint32_t gstart = off;
              for (uint64_t bkey : gk.second) {
                auto kvit = buckets.find(bkey);
                auto& kv = *kvit;
                uint32_t slot;
                table_put(table[l], kv.first, off, (uint32_t)kv.second.size(), slot);
                aux[l][slot].cap = run_capacity((uint32_t)kv.second.size());
                const int bcx = (int)(kv.first & 0x1fffff), bcy = (int)((kv.first >> 21) & 0x1fffff), bcz = (int)((kv.first >> 42) & 0x1fffff);
                for (size_t i = 0; i < kv.second.size(); ++i) {
                    const uint32_t id = kv.second[i];
                    const float4 p = orig[id];
                    bxyz[l][(off + i) * 3 + 0] = p.x; bxyz[l][(off + i) * 3 + 1] = p.y; bxyz[l][(off + i) * 3 + 2] = p.z;
                    bidx[l][off + i] = id;
                    int c[3];
                    cell_of(p, c);
                    const int dx = bcx - (c[0] >> l), dy = bcy - (c[1] >> l), dz = bcz - (c[2] >> l);   // bucket voxel seen from the point
                    backpos[(size_t)id * 27 + (size_t)((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1))] = (uint16_t)(i < 0xFFFFu ? i : 0xFFFFu);
                }
                off += aux[l][slot].cap;
              }
              uint32_t gslot;
              table_put(gtable, gk.first, gstart, off - gstart, gslot);
            }
            set_arenas(l, off);
            cnt.slots_used[l] = (uint32_t)buckets.size();
        }
        {
            std::map<uint64_t, std::vector<uint32_t>> cells;
            for (uint32_t id = 0; id < n_ids; ++id) {
                int c[3];
                cell_of(orig[id], c);
                cells[pack_cell((uint32_t)(c[0] >> 2), (uint32_t)(c[1] >> 2), (uint32_t)(c[2] >> 2))].push_back(id);
            }
            const uint32_t size = next_pow2((uint64_t)cells.size() * 8);
            table[CELL_SLOT].assign(size, uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
            aux[CELL_SLOT].assign(size, SlotAux{0, 0, 0, 0});
            uint64_t total = 0;
            for (auto& kv : cells) total += run_capacity((uint32_t)kv.second.size()) + 8;
            pool_cap[CELL_SLOT] = (uint32_t)(total + total / 4 + pool_reserve);
            cell4.assign(pool_cap[CELL_SLOT], float4{0, 0, 0, 0});
            uint32_t off = 0;
            for (auto& kv : cells) {
                uint32_t slot;
                table_put(table[CELL_SLOT], kv.first, off, (uint32_t)kv.second.size(), slot);
                aux[CELL_SLOT][slot].cap = run_capacity((uint32_t)kv.second.size()) + 8;
                // the GPU build lays a list out in Morton order of the points; any order is valid: use descending id here
                for (size_t i = 0; i < kv.second.size(); ++i) {
                    const uint32_t id = kv.second[kv.second.size() - 1 - i];
                    const float4 p = orig[id];
                    cell4[off + i] = make_float4(p.x, p.y, p.z, __uint_as_float(id));
                    cellpos[id] = (uint32_t)i;
                }
                off += aux[CELL_SLOT][slot].cap;
            }
            set_arenas(CELL_SLOT, off);
            cnt.slots_used[CELL_SLOT] = (uint32_t)cells.size();
        }
        built = true;
    }

    void relinearise() {
        std::vector<float4> keep;
        for (uint32_t i = 0; i < n_ids; ++i)
            if (pt_alive(orig[i])) keep.push_back(orig[i]);
        std::copy(keep.begin(), keep.end(), orig.begin());
        n_ids = (uint32_t)keep.size();
        ++relinearisations;
        rebuild(true);
    }

    void reserve(size_t cap) {
        if (cap <= orig.size()) return;
        size_t n = orig.size() ? orig.size() : 1024;
        while (n < cap) n *= 2;
        orig.resize(n, float4{0, 0, 0, 0});
        backpos.resize(n * 27, (uint16_t)0xFFFEu);
        cellpos.resize(n, 0xFFFFFFFFu);
        if (have_boxes) box_next.resize(n, ID_NONE);
    }

    void ensure_boxes() {
        if (have_boxes) return;
        box.assign(next_pow2((uint64_t)orig.size() * 4), uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
        box_next.assign(orig.size(), ID_NONE);
        cnt.box_slots_used = 0;
        if (n_ids) launch(box_build_kernel, n_ids, bx(), (const float4*)orig.data(), n_ids, &cnt, 0u);
        have_boxes = true;
    }

    void reset_batch() {
        cnt.n_new = cnt.n_dead = cnt.overflow = cnt.dropped = 0;
    }

    // twin of MapStore::add_staged
    void add(const float* xyz, uint32_t k, int downsample) {
        if (k == 0) return;
        std::vector<float4> newp(k);
        for (uint32_t j = 0; j < k; ++j) newp[j] = make_float4(xyz[3 * j], xyz[3 * j + 1], xyz[3 * j + 2], 0.f);
        if ((!built || m == 0) && downsample) {   // Add_Points(points, true) into an empty map: the rule among the new points
            n_ids = 0;
            m = 0;
            built = false;
            reserve(k);
            reset_batch();
            have_boxes = false;
            ensure_boxes();
            MapRW M{};
            M.orig = orig.data();
            M.cnt = &cnt;
            BoxRW B = bx();
            std::vector<uint64_t> keys(k), keys_sorted(k);
            std::vector<uint32_t> idx(k), idx_sorted(k), alive(k), apos(k), order(k);
            std::vector<float4> dead(orig.size());
            launch(inc_box_keys_kernel, k, M, (const float4*)newp.data(), k, box_len, keys.data(), idx.data(), alive.data(), 1, 0);
            for (uint32_t j = 0; j < k; ++j) order[j] = j;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
            for (uint32_t i = 0; i < k; ++i) { keys_sorted[i] = keys[order[i]]; idx_sorted[i] = idx[order[i]]; }
            launch(inc_box_rule_kernel, k, B, orig.data(), (const float4*)newp.data(), (const uint64_t*)keys_sorted.data(),
                   (const uint32_t*)idx_sorted.data(), k, alive.data(), dead.data(), (uint32_t)dead.size(), &cnt);
            uint32_t run = 0;
            for (uint32_t j = 0; j < k; ++j) { apos[j] = run; run += alive[j]; }
            launch(inc_commit_points_kernel, k, M, B, 0, (const float4*)newp.data(), (const uint32_t*)alive.data(),
                   (const uint32_t*)apos.data(), k, 0u);
            n_ids = cnt.n_new;
            rebuild(false);
            return;
        }
        if (!built || m == 0) {
            n_ids = 0;
            reserve(k);
            for (uint32_t j = 0; j < k; ++j)
                if (std::isfinite(newp[j].x) && std::isfinite(newp[j].y) && std::isfinite(newp[j].z)) orig[n_ids++] = newp[j];
            rebuild(false);
            return;
        }
        const uint64_t dead_ids = (uint64_t)n_ids - m;
        if (dead_ids > 64 && dead_ids > n_ids / 3) relinearise();
        reserve((size_t)n_ids + k);
        reset_batch();
        if (downsample) ensure_boxes();
        MapRW M = rw();
        broken.assign(((size_t)k * 27 + 64) * LIST_SHARDS, 0xDEADBEEFu);   // (every shard can take the whole batch: the spread over the shards is the hash's)
        plans.assign(broken.size(), RegroupPlan{});
        for (auto& v : n_broken) v = 0;
        M.broken = broken.data();
        M.broken_cap = (uint32_t)broken.size();
        M.n_broken = n_broken;
        comp.assign(((size_t)k * 27 + 64) * LIST_SHARDS, uint4{0, 0, 0, 0});
        cstage.assign((size_t)(1u << 14) * LIST_SHARDS, float4{0, 0, 0, 0});
        cnew.assign(cstage.size(), 0xDEADBEEFu);
        for (auto& v : n_comp) v = 0;
        M.comp = comp.data();
        M.comp_cap = (uint32_t)comp.size();
        M.n_comp = n_comp;
        M.cstage = cstage.data();
        M.cnew = cnew.data();
        M.cstage_cap = (uint32_t)cstage.size();
        BoxRW B = have_boxes ? bx() : BoxRW{};
        std::vector<uint64_t> keys(k), keys_sorted(k);
        std::vector<uint32_t> idx(k), idx_sorted(k), alive(k), apos(k);
        std::vector<float4> dead(orig.size());
        launch(inc_box_keys_kernel, k, M, (const float4*)newp.data(), k, box_len, keys.data(), idx.data(), alive.data(), downsample, 1);
        if (downsample) {
            std::vector<uint32_t> order(k);
            for (uint32_t j = 0; j < k; ++j) order[j] = j;
            std::stable_sort(order.begin(), order.end(), [&](uint32_t a, uint32_t b) { return keys[a] < keys[b]; });
            for (uint32_t i = 0; i < k; ++i) { keys_sorted[i] = keys[order[i]]; idx_sorted[i] = idx[order[i]]; }
            launch(inc_box_rule_kernel, k, B, orig.data(), (const float4*)newp.data(), (const uint64_t*)keys_sorted.data(),
                   (const uint32_t*)idx_sorted.data(), k, alive.data(), dead.data(), (uint32_t)dead.size(), &cnt);
        }
        uint32_t run = 0;
        for (uint32_t j = 0; j < k; ++j) { apos[j] = run; run += alive[j]; }
        launch(inc_commit_points_kernel, k, M, B, have_boxes ? 1 : 0, (const float4*)newp.data(), (const uint32_t*)alive.data(),
               (const uint32_t*)apos.data(), k, n_ids);
        const uint32_t n_dead = cnt.n_dead;
        // small batches leave the list's length on the device (inc_kill_counted_kernel, a fixed grid walking it in strides);
        // alternate between the two forms so that both run under every thread order
        if (n_dead && (k & 1u)) launch(inc_kill_kernel, (uint64_t)n_dead * INC_SLOTS_PER_POINT, M, (const float4*)dead.data(), n_dead);
        else if (n_dead) launch(inc_kill_counted_kernel, 512, M, (const float4*)dead.data(), (uint32_t)dead.size());
        // voxel groups (twin of the GroupRW set-up in MapStore::add_staged)
        const uint32_t gsize = next_pow2((uint64_t)k * 4);
        std::vector<uint4> gtab[REPL_LEVELS];
        std::vector<uint32_t> gbase[REPL_LEVELS], gslot[REPL_LEVELS];
        std::vector<uint4> gdst[REPL_LEVELS];
        std::vector<uint32_t> prank((size_t)k * REPL_LEVELS, 0u), pslot((size_t)k * REPL_LEVELS, 0u), gcnt(LIST_SHARDS, 0u);
        GroupRW G{};
        for (int l = 0; l < REPL_LEVELS; ++l) {
            gtab[l].assign(gsize, uint4{0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu, 0xFFFFFFFFu});
            gbase[l].assign((size_t)gsize * GROUP_TARGETS, 0xDEADBEEFu);
            gslot[l].assign((size_t)gsize * GROUP_TARGETS, 0xDEADBEEFu);
            gdst[l].assign((size_t)gsize * GROUP_TARGETS, uint4{0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu, 0xDEADBEEFu});
            G.table[l] = gtab[l].data(); G.gbase[l] = gbase[l].data(); G.gslot[l] = gslot[l].data(); G.gdst[l] = gdst[l].data();
        }
        G.mask = gsize - 1;
        G.shift = (uint32_t)(64 - log2u(gsize));
        G.size = gsize;
        G.prank = prank.data();
        G.pslot = pslot.data();
        // every other down-sampling batch walks its survivors through the list in sorted-batch order, as the product's large
        // batches do (twin of the listed path of MapStore::add_staged); the others visit every point in input order
        std::vector<uint32_t> sflag(k), spos(k), surv(k, 0xDEADBEEFu);
        const bool listed = downsample && (batch_no++ & 1u);
        if (listed) {
            launch(inc_surv_flag_kernel, k, (const uint32_t*)idx_sorted.data(), (const uint32_t*)alive.data(), k, sflag.data());
            uint32_t r2 = 0;
            for (uint32_t i = 0; i < k; ++i) { spos[i] = r2; r2 += sflag[i]; }
            launch(inc_surv_list_kernel, k, (const uint32_t*)idx_sorted.data(), (const uint32_t*)sflag.data(), (const uint32_t*)spos.data(), k, surv.data());
            G.surv = surv.data();
            G.n_live = &cnt.n_new;
        }
        const uint64_t kk = listed ? (uint64_t)(cnt.n_new ? cnt.n_new : 1u) : (uint64_t)k;
        launch(inc_group_kernel, (uint64_t)k * REPL_LEVELS, M, G, (const float4*)newp.data(), (const uint32_t*)alive.data(), k);
        std::vector<uint32_t> rank((size_t)k * 27 * SORTED_LEVELS, 0u);
        std::vector<uint4> reloc(((size_t)k * 27 + 64) * LIST_SHARDS);
        const uint64_t t_grp = kk * REPL_LEVELS * GROUP_TARGETS;
        const uint64_t t_all = kk * INC_SLOTS_PER_POINT, t_rep = kk * 27 * SORTED_LEVELS;
        launch(inc_register_kernel, t_grp, M, G, (const uint32_t*)alive.data(), k);
        launch(inc_reserve_kernel, t_grp, M, G, (const uint32_t*)alive.data(), k, reloc.data(), (uint32_t)reloc.size(), gcnt.data());
        // (the product launches a grid for the runs a batch can list, at most 2048 workgroups, and walks longer lists in strides:
        // 16 runs per sweep here, so that the stride loop is exercised)
        launch(inc_compact_gather_kernel, (uint64_t)8 * COMPACT_LANES, M);   // (8 runs per sweep: the stride loop is exercised)
        launch(inc_relocate_kernel, (uint64_t)16 * RELOC_LANES, M, (const uint4*)reloc.data(), (uint32_t)reloc.size(), (const uint32_t*)gcnt.data());
        launch(inc_resolve_kernel, t_grp, M, G, (const uint32_t*)alive.data(), k);
        launch(inc_compact_scatter_kernel, (uint64_t)8 * COMPACT_LANES, M);
        compacted += list_count(n_comp, M.comp_cap);
        launch(inc_fill_kernel, t_all, M, G, (const float4*)newp.data(), (const uint32_t*)alive.data(), (const uint32_t*)apos.data(), k, n_ids);
        launch(inc_rank_kernel, t_rep, M, G, (const uint32_t*)alive.data(), (const uint32_t*)apos.data(), k, n_ids, rank.data());
        launch(inc_place_kernel, t_rep, M, G, (const float4*)newp.data(), (const uint32_t*)alive.data(), (const uint32_t*)apos.data(), k, n_ids,
               (const uint32_t*)rank.data());
        launch(inc_commit_kernel, t_grp, M, G, (const uint32_t*)alive.data(), k);
        // the groups the batch broke up are laid out again (twin of the three launches at the end of MapStore::add_staged)
        launch(inc_regroup_plan_kernel, 512, M, plans.data());
        launch(inc_regroup_move_kernel, 1024, M, (const RegroupPlan*)plans.data());
        launch(inc_regroup_commit_kernel, 512, M, (const RegroupPlan*)plans.data());
        regrouped += list_count(n_broken, M.broken_cap);
        relocations += list_count(gcnt.data(), (uint32_t)reloc.size());
        n_ids += cnt.n_new;
        m += cnt.n_new;
        m -= n_dead;
        n_killed += n_dead;
        if (cnt.overflow) relinearise();
    }

    uint32_t evict_box(const float lo[3], const float hi[3], int keep_inside) {
        if (!built || m == 0) return 0;
        reset_batch();
        std::vector<float4> dead(orig.size());
        launch(inc_evict_box_kernel, n_ids, orig.data(), n_ids, lo[0], lo[1], lo[2], hi[0], hi[1], hi[2], keep_inside, dead.data(),
               (uint32_t)dead.size(), &cnt);
        const uint32_t n_dead = cnt.n_dead;
        if (n_dead) launch(inc_kill_kernel, (uint64_t)n_dead * INC_SLOTS_PER_POINT, rw(), (const float4*)dead.data(), n_dead);
        m -= n_dead;
        if (m == 0) { n_ids = 0; rebuild(true); }
        return n_dead;
    }

    uint32_t evict_oldest(uint32_t n_oldest) {
        if (!built || m == 0 || n_oldest == 0) return 0;
        if (n_oldest > m) n_oldest = m;
        reset_batch();
        std::vector<uint32_t> flags(n_ids), rank(n_ids);
        launch(inc_alive_flags_kernel, n_ids, (const float4*)orig.data(), n_ids, flags.data());
        uint32_t run = 0;
        for (uint32_t i = 0; i < n_ids; ++i) { rank[i] = run; run += flags[i]; }
        std::vector<float4> dead(orig.size());
        launch(inc_evict_oldest_kernel, n_ids, orig.data(), n_ids, (const uint32_t*)rank.data(), n_oldest, dead.data(), (uint32_t)dead.size(),
               &cnt);
        const uint32_t n_dead = cnt.n_dead;
        if (n_dead) launch(inc_kill_kernel, (uint64_t)n_dead * INC_SLOTS_PER_POINT, rw(), (const float4*)dead.data(), n_dead);
        m -= n_dead;
        if (m == 0) { n_ids = 0; rebuild(true); }
        return n_dead;
    }

    // ---- invariants ----------------------------------------------------------------------------------------
    bool fail(const char* fmt, ...) {
        char buf[512];
        va_list ap;
        va_start(ap, fmt);
        vsnprintf(buf, sizeof(buf), fmt, ap);
        va_end(ap);
        err = buf;
        return false;
    }
    bool check() {
        err.clear();
        uint32_t living = 0;
        for (uint32_t i = 0; i < n_ids; ++i) living += pt_alive(orig[i]) ? 1u : 0u;
        if (living != m) return fail("m = %u but %u living ids", m, living);
        if (!built) return m == 0 ? true : fail("points without a structure");
        for (int l = 0; l < REPL_LEVELS; ++l) {
            std::map<uint64_t, std::vector<uint32_t>> want;
            for (uint32_t id = 0; id < n_ids; ++id) {
                if (!pt_alive(orig[id])) continue;
                int c[3];
                cell_of(orig[id], c);
                for (int n = 0; n < 27; ++n) {
                    const int dz = n / 9 - 1, dy = (n / 3) % 3 - 1, dx = n % 3 - 1;
                    const uint32_t nx = (uint32_t)((c[0] >> l) + dx), ny = (uint32_t)((c[1] >> l) + dy), nz = (uint32_t)((c[2] >> l) + dz);
                    if (nx >= (1u << 21) || ny >= (1u << 21) || nz >= (1u << 21)) continue;
                    want[pack_cell(nx, ny, nz)].push_back(id);
                }
            }
            std::vector<std::pair<uint32_t, uint32_t>> runs;
            uint32_t used_slots = 0;
            for (size_t s = 0; s < table[l].size(); ++s) {
                const uint4 e = table[l][s];
                const uint64_t key = entry_key(e);
                if (key == EMPTY_KEY) continue;
                ++used_slots;
                const SlotAux a = aux[l][s];
                if (a.pending) return fail("level %d slot %zu: batch counters not cleared", l, s);
                uint32_t n_tomb = 0;
                if (e.w > a.cap) return fail("level %d slot %zu: count %u > cap %u", l, s, e.w, a.cap);
                if (a.cap) runs.push_back({e.z, a.cap});
                std::vector<uint32_t> got;
                uint32_t prev = 0;
                for (uint32_t i = 0; i < e.w; ++i) {
                    const uint32_t id = bidx[l][e.z + i];
                    if (i && id <= prev) return fail("level %d bucket %llx: ids not ascending at %u (%u after %u)", l, (unsigned long long)key, i, id, prev);
                    prev = id;
                    if (id >= n_ids) return fail("level %d bucket %llx: id %u out of range", l, (unsigned long long)key, id);
                    const float* rec = &bxyz[l][(size_t)(e.z + i) * 3];
                    if (std::isinf(rec[0])) {
                        if (pt_alive(orig[id])) return fail("level %d bucket %llx: tombstone for living id %u", l, (unsigned long long)key, id);
                        ++n_tomb;
                        continue;
                    }
                    if (!pt_alive(orig[id])) return fail("level %d bucket %llx: dead id %u still listed", l, (unsigned long long)key, id);
                    if (std::memcmp(rec, &orig[id], 12) != 0)
                        return fail("level %d bucket %llx: coordinates of id %u differ", l, (unsigned long long)key, id);
                    {   // the point must know this position (16 bits; beyond: the FAR mark)
                        int c[3];
                        cell_of(orig[id], c);
                        const int dx = (int)(key & 0x1fffff) - (c[0] >> l), dy = (int)((key >> 21) & 0x1fffff) - (c[1] >> l),
                                  dz = (int)((key >> 42) & 0x1fffff) - (c[2] >> l);
                        if (backpos[(size_t)id * 27 + (size_t)((dz + 1) * 9 + (dy + 1) * 3 + (dx + 1))] != (uint16_t)(i < 0xFFFFu ? i : 0xFFFFu))
                            return fail("level %d bucket %llx: back-position of id %u is stale", l, (unsigned long long)key, id);
                    }
                    got.push_back(id);
                }
                auto it = want.find(key);
                const std::vector<uint32_t> none;
                const std::vector<uint32_t>& w = it == want.end() ? none : it->second;
                if (got != w) return fail("level %d bucket %llx: %zu living entries, %zu expected", l, (unsigned long long)key, got.size(), w.size());
                if (n_tomb != a.dead) return fail("level %d bucket %llx: %u tombstones, the run's count says %u", l, (unsigned long long)key, n_tomb, a.dead);
                if (it != want.end()) want.erase(it);
            }
            if (!want.empty()) return fail("level %d: %zu voxels with living neighbours have no bucket", l, want.size());
            if (used_slots != cnt.slots_used[l]) return fail("level %d: slots_used %u, table holds %u", l, cnt.slots_used[l], used_slots);
            std::sort(runs.begin(), runs.end());
            for (size_t i = 0; i < runs.size(); ++i) {
                if ((uint64_t)runs[i].first + runs[i].second > pool_cap[l]) return fail("level %d: run beyond the pool", l);
                if (i && runs[i - 1].first + runs[i - 1].second > runs[i].first) return fail("level %d: runs overlap", l);
            }
        }
        {   // tile groups: an intact group's region is exactly its buckets' runs side by side, everything outside the runs' entries
            // reads +inf, and every bucket that belongs to the group lies inside; no group is left "listed"
            std::map<uint64_t, std::vector<size_t>> members;
            for (size_t s = 0; s < table[0].size(); ++s) {
                const uint64_t key = entry_key(table[0][s]);
                if (key == EMPTY_KEY) continue;
                uint32_t vx, vy, vz; int r;
                tile_group_of((uint32_t)(key & 0x1fffff), (uint32_t)((key >> 21) & 0x1fffff), (uint32_t)((key >> 42) & 0x1fffff), vx, vy, vz, r);
                members[pack_cell(vx, vy, vz)].push_back(s);
            }
            uint32_t gused = 0;
            for (size_t g = 0; g < gtable.size(); ++g) {
                const uint4 ge = gtable[g];
                const uint64_t gkey = entry_key(ge);
                if (gkey == EMPTY_KEY) continue;
                ++gused;
                if (ge.w == 0u) { if (ge.z == ID_NONE) return fail("group %llx: still listed after the batch", (unsigned long long)gkey); continue; }   // broken (by a sweep): nothing is claimed
                if ((uint64_t)ge.z + ge.w > pool_cap[0]) return fail("group %llx: region beyond the pool", (unsigned long long)gkey);
                std::vector<uint8_t> covered(ge.w, 0);
                auto it = members.find(gkey);
                if (it == members.end()) return fail("group %llx: intact but without buckets", (unsigned long long)gkey);
                for (size_t s : it->second) {
                    const uint4 e = table[0][s];
                    const uint32_t cap = aux[0][s].cap;
                    if (e.z < ge.z || (uint64_t)e.z + cap > (uint64_t)ge.z + ge.w) return fail("group %llx: a bucket's run lies outside the region", (unsigned long long)gkey);
                    for (uint32_t i = 0; i < e.w; ++i) covered[e.z - ge.z + i] = 1;
                }
                for (uint32_t i = 0; i < ge.w; ++i)
                    if (!covered[i] && !std::isinf(bxyz[0][(size_t)(ge.z + i) * 3])) return fail("group %llx: slack entry %u does not read +inf", (unsigned long long)gkey, i);
            }
            if (gused != cnt.gslots_used) return fail("groups: gslots_used %u, table holds %u", cnt.gslots_used, gused);
            // (a bucket whose group has no entry, or a broken one, is searched through the lists: allowed)
        }
        {
            std::vector<uint8_t> seen(n_ids, 0);
            std::vector<std::pair<uint32_t, uint32_t>> runs;
            for (size_t s = 0; s < table[CELL_SLOT].size(); ++s) {
                const uint4 e = table[CELL_SLOT][s];
                const uint64_t key = entry_key(e);
                if (key == EMPTY_KEY) continue;
                const SlotAux a = aux[CELL_SLOT][s];
                if (a.pending) return fail("voxel list %zu: batch counters not cleared", s);
                if (e.w > a.cap) return fail("voxel list %zu: count %u > cap %u", s, e.w, a.cap);
                if (a.cap) runs.push_back({e.z, a.cap});
                for (uint32_t i = 0; i < e.w; ++i) {
                    const float4 r = cell4[e.z + i];
                    const uint32_t id = __float_as_uint(r.w);
                    if (id >= n_ids) return fail("voxel list %llx: id %u out of range", (unsigned long long)key, id);
                    if (std::isinf(r.x)) { if (pt_alive(orig[id])) return fail("voxel list: tombstone for living id %u", id); continue; }
                    if (!pt_alive(orig[id])) return fail("voxel list: dead id %u still listed", id);
                    if (std::memcmp(&r, &orig[id], 12) != 0) return fail("voxel list: coordinates of id %u differ", id);
                    int c[3];
                    cell_of(orig[id], c);
                    if (pack_cell((uint32_t)(c[0] >> 2), (uint32_t)(c[1] >> 2), (uint32_t)(c[2] >> 2)) != key) return fail("voxel list: id %u in the wrong voxel", id);
                    if (seen[id]++) return fail("voxel list: id %u listed twice", id);
                    if (cellpos[id] != i) return fail("voxel list: position of id %u is stale", id);
                }
            }
            for (uint32_t id = 0; id < n_ids; ++id)
                if (pt_alive(orig[id]) && !seen[id]) return fail("voxel lists: living id %u missing", id);
            std::sort(runs.begin(), runs.end());
            for (size_t i = 0; i < runs.size(); ++i) {
                if ((uint64_t)runs[i].first + runs[i].second > pool_cap[CELL_SLOT]) return fail("voxel lists: run beyond the pool");
                if (i && runs[i - 1].first + runs[i - 1].second > runs[i].first) return fail("voxel lists: runs overlap");
            }
        }
        if (have_boxes) {
            std::vector<uint8_t> seen(n_ids, 0);
            for (size_t s = 0; s < box.size(); ++s) {
                if (entry_key(box[s]) == EMPTY_KEY) continue;
                uint32_t steps = 0;
                for (uint32_t e = box[s].z; e != ID_NONE; e = box_next[e]) {
                    if (e >= n_ids || ++steps > n_ids) return fail("box chain broken at slot %zu", s);
                    if (!pt_alive(orig[e])) continue;
                    if (inc_box_key(orig[e], box_len) != entry_key(box[s])) return fail("box chain: id %u in the wrong box", e);
                    if (seen[e]++) return fail("box chain: id %u reached twice", e);
                }
            }
            for (uint32_t id = 0; id < n_ids; ++id)
                if (pt_alive(orig[id]) && !seen[id]) return fail("box chains: living id %u unreachable", id);
        }
        return true;
    }
};

}  // namespace

extern "C" {
void* emu_create(float cell, uint32_t pool_reserve) { Emu* e = new Emu(); e->cell = cell; e->pool_reserve = pool_reserve; return e; }
void emu_destroy(void* h) { delete static_cast<Emu*>(h); }
void emu_set_order(int order, uint64_t seed) { g_order = order; g_rng = seed | 1ull; }
void emu_add(void* h, const float* xyz, uint32_t k, int downsample) { static_cast<Emu*>(h)->add(xyz, k, downsample); }
uint32_t emu_evict_box(void* h, const float* lo, const float* hi, int keep_inside) { return static_cast<Emu*>(h)->evict_box(lo, hi, keep_inside); }
uint32_t emu_evict_oldest(void* h, uint32_t n) { return static_cast<Emu*>(h)->evict_oldest(n); }
void emu_relinearise(void* h) { static_cast<Emu*>(h)->relinearise(); }
uint32_t emu_size(void* h) { return static_cast<Emu*>(h)->m; }
uint32_t emu_ids(void* h) { return static_cast<Emu*>(h)->n_ids; }
uint64_t emu_relinearisations(void* h) { return static_cast<Emu*>(h)->relinearisations; }
uint64_t emu_relocations(void* h) { return static_cast<Emu*>(h)->relocations; }
uint64_t emu_regrouped(void* h) { return static_cast<Emu*>(h)->regrouped; }
uint64_t emu_compacted(void* h) { return static_cast<Emu*>(h)->compacted; }
// the pure helpers of the insert's work order (lv_mapinc.hpp), exposed for tests/test_mapinc_emulation.py
uint32_t emu_block_of(uint32_t b, uint32_t n) { return inc_block_of(b, n); }
uint64_t emu_box_key(float x, float y, float z, float len) { return inc_box_key(float4{x, y, z, 0.f}, len); }
uint32_t emu_tombstones(void* h) { return (uint32_t)static_cast<Emu*>(h)->n_killed; }
uint32_t emu_fetch(void* h, float* out) {
    Emu* e = static_cast<Emu*>(h);
    uint32_t o = 0;
    for (uint32_t i = 0; i < e->n_ids; ++i)
        if (pt_alive(e->orig[i])) { out[3 * o] = e->orig[i].x; out[3 * o + 1] = e->orig[i].y; out[3 * o + 2] = e->orig[i].z; ++o; }
    return o;
}
int emu_check(void* h, char* msg, int cap) {
    Emu* e = static_cast<Emu*>(h);
    const bool ok = e->check();
    if (msg && cap > 0) { std::strncpy(msg, e->err.c_str(), (size_t)cap - 1); msg[cap - 1] = 0; }
    return ok ? 1 : 0;
}
}
