// lv_pass_dev.hpp — the filter algebra of one pass as device functions over caller-provided LDS, for
// pass_kernel (lv_match.hip): ONE launch per measurement pass instead of search / fit / solve.
//
// pass_kernel(p) = [prologue: the solve of pass p-1, computed REDUNDANTLY by every workgroup from the
// workgroup partials that pass_kernel(p-1) left in memory] -> [search + plane fits of pass p] -> [one
// partial per workgroup].  The redundant solve removes two dependent kernel boundaries per pass (fit ->
// solve -> search) and the single-workgroup solve kernel's launch ramp; it costs every workgroup the
// solve's ~5 us latency chain once, at a time when the GPU would otherwise idle behind that chain anyway.
// All workgroups execute the same instructions on the same inputs, so they derive bit-identical states and
// pass constants (no broadcast, no inter-workgroup wait).  One workgroup also keeps the books — a dedicated
// one when the scan leaves a CU free, otherwise the searching workgroup that was quickest in the previous
// launch, AFTER its own search and fits: state / trace / sums log in KfDev, the record-independent half of
// the NEXT solve (prepare_next), and on the pass that ends the update the posterior covariance and the host
// mailbox.  What a launch hands to the next one lives in KfDev::ps[launch parity].
//
// Same algebra as solve_kernel / solve_prep (lv_solve.hip, lv_solve_dev.hpp):
// esekf::update_iterated_dyn_share_modified [IKFoM absent from the reference mount; UPSTREAM-RECALL of
// hku-mars/IKFoM esekfom.hpp; call site reference src/Modules/Localizator.cpp:132].
#pragma once

#include "lv_host.hpp"
#include "lv_solve_dev.hpp"

#pragma clang fp contract(fast)

namespace lv {

// reduction outputs: output t owns the product column pair (a, b) of a staged Jacobian row and its slot
// `rec` in the 96-double record ([0..77] upper triangle of the 12x12 H^T H, [78..89] H^T h, 90 n_valid,
// 91 sum h^2).  W = number of Jacobian columns that can be non-zero (6 without extrinsics, else 12).
template <int W>
__device__ __forceinline__ void out_pair(int t, int& a, int& b, int& rec) {
    constexpr int NTRI = W * (W + 1) / 2;
    if (t < NTRI) {
        int i = 0, rem = t;
        while (rem >= W - i) { rem -= W - i; ++i; }
        a = i;
        b = i + rem;
        rec = a * 12 - a * (a - 1) / 2 + (b - a);
    } else if (t < NTRI + W) {
        a = t - NTRI;
        b = W;          // h
        rec = 78 + a;
    } else if (t == NTRI + W) {
        a = W + 1;      // valid * valid
        b = W + 1;
        rec = 90;
    } else {
        a = W;          // h * h
        b = W;
        rec = 91;
    }
}

// Workgroup partials of pass_kernel are COMPACT: OW doubles per workgroup, entry t = output t of out_pair<W>
// (t < NOUT), so that the prologue's fold reads whole cache lines of live data.
template <int NW> struct PassDims {
    static constexpr int NOUT = NW * (NW + 1) / 2 + NW + 2;   // 29 / 92
    static constexpr int OW = NW == 6 ? 32 : 96;              // record stride in doubles
};

#ifndef LV_PK_THREADS
#define LV_PK_THREADS 1024
#endif
constexpr int PK_THREADS = LV_PK_THREADS;   // one workgroup per CU: 16 wavefronts (-DLV_PK_THREADS=512: the eight-wavefront / 256-VGPR experiment, profiles/experiments_r05/)
static_assert(PK_THREADS == 1024 || PK_THREADS == 512, "pass_kernel is written for 16 or 8 wavefronts");

// LDS of the prologue solve (every workgroup)
struct SolveLds {
    double part[PK_THREADS];   // [parts][OW]
    double rec[SUMS_LEN];      // the folded record in the 96-double layout of the C-ABI (lv_sums)
    double HTH[12][12];
    double HTh[12];
    double G[12][12];          // A1 = (P_/R)_ww^-1   (prepare)
    double A[NS][12];          // (P_/R)[:, 0:NW]
    double W[2][12][13];
    double T[12][12];
    double X[NS][12];
    double v[12];
    double dxnew[NS];
    double dxo[NS];
    double x[NX];
    double Rot[4][9];
    float ptmp[8];
    int last, conv, n_valid0, t_new;
    int kf_t, kf_iter, pass, done;
    int cheapest, pad_[3];     // the workgroup whose search + fits took least time in the previous launch (keeps the books now)
};
// What the bookkeeping workgroup must remember of its prologue solve while region 0 of its LDS serves the search
struct KeepLds {
    double rec[SUMS_LEN];
    double HTH[12][12];
    double X[NS][12];
    double dxo[NS];
    double x[NX];
    double Pprop[NS * NS];   // propagated covariance and state: constant during an update, fetched while the prologue's loads fly
    double xp[NX];
    int last, n_valid0, t_new, kf_iter, pass, pad_;
};
// LDS of the bookkeeping workgroup's extra work (prepare / terminal pass); follows SolveLds
struct BookLds {
    double P[NS][LD], A[NS][LD], B[NS][LD], J[NS][LD];
    double Kx[NS][12], XA[NS][12];
    double W[2][12][13];
    double xp[NX], dx[NS];
    uint32_t chk, pad_;
};

// The solve of one pass from `nrec` compact workgroup partials: x <- x [+] dx_, convergence bookkeeping, the f32
// constants of the next pass.  T = PK_THREADS threads, all must call.  On return (after its final barrier):
// L.x (new state), L.dxo, L.X, L.HTH, L.rec, L.last, L.t_new, L.n_valid0, *pose.
// Returns false (to every thread) if the update had already ended in an earlier launch (kf->done): nothing was computed.
// clk: optional stamp slot of this workgroup (instrumentation).
template <int NW, bool WITH_POSE = true>
__device__ inline bool solve_core(SolveLds& L, const KfDev* __restrict__ kf, const KfDev::PassState* __restrict__ in,
                                  const double* __restrict__ recs, int nrec, const SolveParams& prm, PoseConsts* pose, int tid,
                                  long long* clk, const uint32_t* __restrict__ cost_in = nullptr, int ncost = 0, BookLds* term = nullptr) {
    // term (closing launch only): the terminal books' scratch, with J = identity and xp = the propagated state already in it:
    // the three projection blocks of the terminal pass (A-matrix of dx_ on the two SO3 blocks, Nx / Mx on S2) depend on dx_ —
    // and the S2 one on the new gravity direction — only, so they are computed HERE, beside the boxplus, by lanes that would
    // idle, instead of behind two more barriers in the books (round 4: the closing launch's serial chain is shorter by them)
    // cost_in (optional): how long each of the ncost searching workgroups of the PREVIOUS launch took; L.cheapest = the
    // quickest one (ties: the highest index), -1 without a history
    constexpr int T = PK_THREADS;
    constexpr int OW = PassDims<NW>::OW, NOUT = PassDims<NW>::NOUT;
    constexpr int PARTS = T / OW;                 // 32 / 10
    constexpr int DEPTH = NW == 6 ? 16 : 28;      // records per thread issued in one memory round trip: x PARTS = 512 records (one
                                                  // per CU on one GPU; all ranks' workgroups in the multi-GPU form), more in further trips
    const int wave = tid >> 6, lane = tid & 63;
    const int fo = tid % OW, fpart = tid / OW;
    // ---- every global read up front (one memory round trip)
    double fv[DEPTH];
    const bool folder = fpart < PARTS && fo < NOUT;
#pragma unroll
    for (int i = 0; i < DEPTH; ++i) {
        const int r = fpart + PARTS * i;
        fv[i] = (folder && r < nrec) ? recs[(size_t)r * OW + fo] : 0.0;
    }
    if (tid < NX) L.x[tid] = in->x[tid];
    for (int e = tid; e < NS * NW; e += T) L.A[e / NW][e % NW] = in->prep_P[(e / NW) * NS + (e % NW)] * prm.R_inv;
    if (tid >= 64 && tid < 64 + NS) L.dxnew[tid - 64] = in->prep_dxnew[tid - 64];
    if (tid >= 128 && tid < 128 + NW * NW) L.G[(tid - 128) / NW][(tid - 128) % NW] = in->prep_A1[tid - 128];
    if (tid < SUMS_LEN) L.rec[tid] = 0.0;
    if (tid == 448) { L.kf_t = in->t; L.kf_iter = in->iter; L.pass = in->passes; L.done = kf->done; L.conv = 1; L.last = 0; L.n_valid0 = 0; }
    // the previous launch's workgroup times: loaded with everything else (one round trip for up to 512 workgroups; more are
    // simply never chosen), reduced by the last wavefront AFTER the fold's barrier, beside the solve chain
    uint32_t cv[8];
    if (tid >= PK_THREADS - 64) {
#pragma unroll
        for (int u = 0; u < 8; ++u) cv[u] = (tid - (PK_THREADS - 64) + 64 * u < ncost) ? cost_in[tid - (PK_THREADS - 64) + 64 * u] : 0xFFFFFFFFu;
    }
    // ---- fold, fixed order: thread (fo, fpart) sums records fpart, fpart + PARTS, ... (four interleaved running sums),
    // the PARTS part sums are then added left to right
    {
        double a0 = fv[0], a1 = fv[1], a2 = fv[2], a3 = fv[3];
#pragma unroll
        for (int i = 4; i + 3 < DEPTH; i += 4) { a0 += fv[i]; a1 += fv[i + 1]; a2 += fv[i + 2]; a3 += fv[i + 3]; }
        static_assert(DEPTH % 4 == 0, "fold assumes a multiple of four");
        double s = (a0 + a1) + (a2 + a3);
        for (int r = fpart + PARTS * DEPTH; folder && r < nrec; r += PARTS) s += recs[(size_t)r * OW + fo];   // (larger grids)
        if (fpart < PARTS) L.part[fpart * OW + fo] = s;
    }
    __syncthreads();
    if (L.done) return false;   // (the loads above were issued before this word was known: one round trip, not two)
    if (clk && tid == 0) { clk[1] = clock64(); clk[17] = wall_clock64(); }
    if (tid >= PK_THREADS - 64) {   // argmin of the workgroup times (key = time << 16 | 65535 - index: ties go to the highest index)
        const int ln = tid - (PK_THREADS - 64);
        unsigned long long best = ~0ull;
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int w = ln + 64 * u;
            const unsigned long long key = w < ncost ? (((unsigned long long)cv[u] << 16) | (unsigned long long)(65535 - w)) : ~0ull;
            best = key < best ? key : best;
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const unsigned long long o = (unsigned long long)__shfl_xor((long long)best, m);
            best = o < best ? o : best;
        }
        if (ln == 0) L.cheapest = (cost_in && ncost > 0) ? 65535 - (int)(best & 0xFFFFull) : -1;
    }
    if (tid < NOUT) {
        double s = L.part[tid];
#pragma unroll
        for (int p = 1; p < PARTS; ++p) s += L.part[p * OW + tid];
        int a, b, rec;
        out_pair<NW>(tid, a, b, rec);
        L.rec[rec] = s;
        if (b < NW) { L.HTH[a][b] = s; L.HTH[b][a] = s; }
        else if (b == NW && a < NW) L.HTh[a] = s;
    }
    __syncthreads();
    const double n_valid = L.rec[90];
    const int kf_t = L.kf_t, kf_iter = L.kf_iter;
    if (n_valid == 0.0) {   // h_share_model: dyn_share.valid = false -> `continue`: the state does not move
        if (tid == 0) {
            L.n_valid0 = 1;
            L.t_new = kf_t;
            L.last = (kf_iter + 1 >= prm.maximum_iter) ? 1 : 0;
        }
        if (tid < NS) L.dxo[tid] = 0.0;
    } else {
        // X = P_inv[:, 0:NW]:  X_top = (Pr_ww^-1 + HTH_ww)^-1,  X_bot = Pr[NW:, 0:NW] Pr_ww^-1 X_top   (lv_solve.hip)
        int cur = 0;
        if (tid < NW * NW) {
            const int i = tid / NW, j = tid % NW;
            L.W[cur][i][j] = L.G[i][j] + L.HTH[i][j];
        }
        if (tid >= 64 && tid < 64 + NW) {   // v = HTh + HTH dx_new[:NW]
            const int i = tid - 64;
            double s = L.HTh[i];
            for (int j = 0; j < NW; ++j) s += L.HTH[i][j] * L.dxnew[j];
            L.v[i] = s;
        }
        __syncthreads();
        gj_spd<NW>(L.W, cur, tid);               // L.W[cur] = X_top
        if (clk && tid == 0) { clk[2] = clock64(); clk[18] = wall_clock64(); }
        if (tid < NW * NW) {                     // T = A1 X_top
            const int i = tid / NW, c = tid % NW;
            double s = 0.0;
            for (int j = 0; j < NW; ++j) s += L.G[i][j] * L.W[cur][j][c];
            L.T[i][c] = s;
        }
        __syncthreads();
        if (tid < NS * NW) {                     // X = [X_top ; Pr[NW:, :NW] T]
            const int i = tid / NW, c = tid % NW;
            double v;
            if (i < NW) {
                v = L.W[cur][i][c];
            } else {
                double s = 0.0;
                for (int j = 0; j < NW; ++j) s += L.A[i][j] * L.T[j][c];
                v = s;
            }
            L.X[i][c] = v;
        }
        __syncthreads();
        if (tid < NS) {  // dx_ = X v - dx_new
            double s = 0.0;
            for (int j = 0; j < NW; ++j) s += L.X[tid][j] * L.v[j];
            const double d = s - L.dxnew[tid];
            L.dxo[tid] = d;
            if (fabs(d) > prm.limits[tid]) L.conv = 0;  // dyn_share.converge
        }
        __syncthreads();
        // x_.boxplus(dx_)
        // (round 4, measured and dropped: the two rotation lanes going straight on to their matrices and the conjugates' instead
        // of a barrier and four other lanes — two more serial f64 chains per lane cost more than the barrier they replace:
        // prologue 6.16 -> 6.25 us)
        if (wave < 3 && lane == 0) {
            boxplus_block(wave, L.x, L.dxo);
            if (term && wave == 2) manifold_block(2, 1, L.x, term->xp, L.dxo, nullptr, term->J);   // (after its boxplus: needs the new gravity)
        }
        if (term && (wave == 4 || wave == 5) && lane == 0) manifold_block(wave - 4, 1, L.x, term->xp, L.dxo, nullptr, term->J);
        if (wave == 3 && lane < 15) {
            const int dof = lane < 3 ? lane : lane + 6;
            L.x[vect_state_index(dof)] += L.dxo[dof];
        }
        if (tid == 256) {
            int t = kf_t;
            if (L.conv) t++;
            L.t_new = t;
            L.last = (t > 1 || kf_iter == prm.maximum_iter - 1) ? 1 : 0;
        }
    }
    if (!WITH_POSE) { __syncthreads(); return true; }   // (closing launch: no pass follows)
    __syncthreads();
    // the constants of the coming pass by ONE wavefront: four rotation matrices, one lane each, then the composed transforms
    // spread over its lanes in two stages — wavefront-local fences between the three steps instead of the two workgroup
    // barriers of rounds 2-3 (same operations, same order as compute_pose_consts => same bits)
    if (tid >= 64 && tid < 128) {
        if (tid < 68) {
            const int w = tid - 64;                // 0: rot, 1: offset_R_L_I, 2: conj(rot), 3: conj(offset_R_L_I)
            const int q = (w & 1) ? 7 : 3;
            const double sg = (w & 2) ? -1.0 : 1.0;
            const double qq[4] = {sg * L.x[q], sg * L.x[q + 1], sg * L.x[q + 2], L.x[q + 3]};
            quat_to_rot(qq, &L.Rot[w][0]);
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pose_consts_stage_a(tid - 64, L.x, L.Rot, pose, L.ptmp);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        pose_consts_stage_b(tid - 64, L.Rot, pose, L.ptmp);
    }
    __syncthreads();
    return true;
}

// ---- the books --------------------------------------------------------------------------------------------------
// Run by T threads (tid = 0 .. T - 1) that synchronise through `bar` (today always the whole workgroup, WgBar; the
// functions are written against a barrier object because a sub-group of wavefronts running them beside the plane fits
// was measured — DESIGN §3 — and may return once the books fit a smaller register budget).
struct WgBar {
    __device__ __forceinline__ void operator()() { __syncthreads(); }
};
// Barrier among a SUBSET of a workgroup's wavefronts (nw of them, all must call) over an LDS counter: the bookkeeping
// workgroup's twelve idle wavefronts keep the books beside the four that fit planes.  The counter only grows (the k-th
// barrier waits for k * nw arrivals), starts at zero at kernel start and is private to one use per launch.
struct SubBar {
    int* ctr;
    int nw, target;
    __device__ __forceinline__ SubBar(int* c, int n) : ctr(c), nw(n), target(0) {}
    __device__ __forceinline__ void operator()() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");   // this wavefront's LDS / memory writes are out
        target += nw;
        if ((threadIdx.x & 63u) == 0) atomicAdd(ctr, 1);
        while (__hip_atomic_load(ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < target) __builtin_amdgcn_s_sleep(1);
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    }
};
// in-place (ping-pong) Gauss-Jordan inverse of an SPD NW x NW matrix as gj_spd (lv_solve_dev.hpp), synchronising through bar
template <int NW, class Bar>
__device__ inline void gj_spd_bar(double (*W)[12][13], int& cur, int tid, Bar& bar) {
    if (NW == 6) {
        if (tid < 64) {   // the whole matrix lives in the registers of one wavefront (gj6_in_lanes, lv_solve_dev.hpp)
            const int l = tid < 36 ? tid : 35;
            const double w = gj6_in_lanes(W[cur][l / 6][l % 6], tid);
            if (tid < 36) W[cur][tid / 6][tid % 6] = w;
        }
        bar();
        return;
    }
    for (int k = 0; k < NW; ++k) {
        if (tid < NW * NW) {
            const int i = tid / NW, j = tid % NW;
            const double rp = ddiv(1.0, W[cur][k][k]);
            double v;
            if (i == k) {
                v = (j == k) ? rp : W[cur][k][j] * rp;
            } else {
                const double f = W[cur][i][k];
                v = (j == k) ? -(f * rp) : W[cur][i][j] - f * (W[cur][k][j] * rp);
            }
            W[cur ^ 1][i][j] = v;
        }
        bar();
        cur ^= 1;
    }
}

// The record-independent half of the NEXT solve (solve_prep of lv_solve_dev.hpp over caller-provided LDS): dx = x [-] x_prop
// with its projection J, dx_new = J dx, P_ = J P_prop J^T and A1 = (P_/R)_ww^-1 -> out->prep_*.  x: the state the coming
// pass is evaluated at (LDS).  The caller's threads have stored the propagated covariance in Bk.B and the propagated
// state in Bk.xp (no barrier needed in between).
template <int NW, int T, class Bar>
__device__ inline void prepare_next(BookLds& Bk, KfDev::PassState* __restrict__ out, const double* x, const double* xprop, double R_inv,
                                    int tid, Bar& bar, long long* clk) {
    // x / xprop: LDS, published before the call (a barrier lies between their last write and this call); Bk.B = P_prop:
    // written by the caller's threads right before the call (the first barrier below publishes it)
    const int wave = tid >> 6, lane = tid & 63;
    // J = identity outside the three manifold blocks; those are written by manifold_block below: disjoint, no barrier needed
    for (int e = tid; e < NS * NS; e += T) {
        const int i = e / NS, j = e % NS;
        int bi, ni, bj, nj;
        blk_range(i, bi, ni);
        blk_range(j, bj, nj);
        if (!(ni > 1 && bi == bj)) Bk.J[i][j] = (i == j) ? 1.0 : 0.0;
    }
    if (clk && tid == 0) { clk[11] = clock64(); clk[27] = wall_clock64(); }
    if (wave < 3 && lane == 0) manifold_block(wave, 0, x, xprop, nullptr, Bk.dx, Bk.J);
    if (wave == 3 && lane < 15) {
        const int dof = lane < 3 ? lane : lane + 6;  // 0..2, 9..20
        const int si = vect_state_index(dof);
        Bk.dx[dof] = x[si] - xprop[si];
    }
    bar();
    if (clk && tid == 0) { clk[12] = clock64(); clk[28] = wall_clock64(); }
    if (tid < NS) {  // dx_new = J dx (identity outside the blocks)
        double s = 0.0;
        const int b = (tid >= 3 && tid < 6) ? 3 : (tid >= 6 && tid < 9) ? 6 : (tid >= 21) ? 21 : -1;
        if (b < 0) s = Bk.dx[tid];
        else if (b == 21) s = Bk.J[tid][21] * Bk.dx[21] + Bk.J[tid][22] * Bk.dx[22];
        else s = dot3d(Bk.J[tid][b], Bk.dx[b], Bk.J[tid][b + 1], Bk.dx[b + 1], Bk.J[tid][b + 2], Bk.dx[b + 2]);
        out->prep_dxnew[tid] = s;
    }
    congruence<T>(Bk.P, Bk.J, Bk.B, tid);  // P_ = J P_prop J^T
    bar();
    if (clk && tid == 0) { clk[13] = clock64(); clk[29] = wall_clock64(); }
    for (int e = tid; e < NS * NS; e += T) out->prep_P[e] = Bk.P[e / NS][e % NS];
    if (NW == 6) {   // A1 = (P_/R)_ww^-1 straight from P_ in the registers of one wavefront
        if (tid < 64) {
            const int l = tid < 36 ? tid : 35;
            const double w = gj6_in_lanes(Bk.P[l / 6][l % 6] * R_inv, tid);
            if (tid < 36) out->prep_A1[tid] = w;
        }
        return;
    }
    if (tid < NW * NW) Bk.W[0][tid / NW][tid % NW] = Bk.P[tid / NW][tid % NW] * R_inv;
    bar();
    int cur = 0;
    gj_spd_bar<NW>(Bk.W, cur, tid, bar);
    if (tid < NW * NW) out->prep_A1[tid] = Bk.W[cur][tid / NW][tid % NW];
}

#define LV_IO_STORE(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)

// The books of the pass just solved (what solve_core left, saved in K): state / trace / sums log, the hand-over to the next
// launch (out); on the pass that ends the update the posterior covariance and the host mailbox (as solve_kernel's terminal
// part), otherwise prepare_next.  Pprop / xprop: the propagated covariance / state (K's early-fetched copies, or kf's);
// prep_preloaded: Bk.P already holds in->prep_P and Bk.xp the propagated state (closing launch).
template <int NW, int T, class Bar>
__device__ __forceinline__ void bookkeeping(const KeepLds& K, BookLds& Bk, KfDev* __restrict__ kf, const KfDev::PassState* __restrict__ in,
                                   KfDev::PassState* __restrict__ out, KfHostIO* io, double* __restrict__ sums_out,
                                   double prm_R_inv, int prm_seq, const PoseConsts* pose, const double* Pprop, const double* xprop,
                                   bool prep_preloaded, int tid, Bar& bar, long long* clk, bool blocks_ready = false) {
    // blocks_ready: Bk.J already holds the terminal pass' projection (solve_core computed its three blocks: closing launch)
    const int wave = tid >> 6, lane = tid & 63;
    const int pass = K.pass, last = K.last, kf_iter = K.kf_iter;
    if (tid < SUMS_LEN) {
        if (sums_out) sums_out[tid] = K.rec[tid];
        if (pass < MAX_PASSES) kf->sums_log[pass * SUMS_LEN + tid] = K.rec[tid];
    }
    if (tid < NX) { kf->x[tid] = K.x[tid]; out->x[tid] = K.x[tid]; }
    if (tid >= 64 && tid < 64 + 49 && pass < MAX_PASSES) {
        const int e = tid - 64;
        kf->trace[pass * 49 + e] = e < NS ? K.dxo[e] : K.x[e - NS];
    }
    {
        constexpr int NW32 = (int)(sizeof(PoseConsts) / 4);
        // (as solve_kernel: the constants of the next pass; none follows the pass that ends the update)
        if (!last && tid >= 128 && tid < 128 + NW32) reinterpret_cast<uint32_t*>(&kf->pose)[tid - 128] = reinterpret_cast<const uint32_t*>(pose)[tid - 128];
    }
    if (tid == 0) {
        kf->t = K.t_new;
        kf->passes = pass + 1;
        kf->iter = kf_iter + 1;
        out->t = K.t_new;
        out->passes = pass + 1;
        out->iter = kf_iter + 1;
        if (last) {
            kf->done = 1;
            LV_IO_STORE(&io->passes, pass + 1);
            LV_IO_STORE(&io->fallback_queries, kf->fallback_queries);
        }
    }
    if (!last) {
        for (int e = tid; e < NS * NS; e += T) Bk.B[e / NS][e % NS] = Pprop[e];
        prepare_next<NW, T>(Bk, out, K.x, xprop, prm_R_inv, tid, bar, clk);
        return;
    }
    if (tid < NX) LV_IO_STORE(&io->x[tid], K.x[tid]);
    if (K.n_valid0) {
        // the update ends on a pass without matches: the mailbox keeps the covariance the install stored (propagated);
        // no checksum on this rare path: the host synchronises the stream
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        bar();
        if (tid == 0)
            __hip_atomic_store(&io->seqcheck, ((unsigned long long)MAILBOX_UNCHECKED << 32) | (unsigned long long)(uint32_t)prm_seq,
                               __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    // terminal pass: L_ = J2 P_ J2^T, K_x rows projected, P_ <- P_ J2^T, P = L_ - K_x[:, :NW] P_[0:NW, :]
    if (!prep_preloaded) {
        for (int e = tid; e < NS * NS; e += T) Bk.P[e / NS][e % NS] = in->prep_P[e];
        if (tid < NX) Bk.xp[tid] = xprop[tid];
    }
    if (tid == 0) Bk.chk = 0u;
    if (!blocks_ready) {
        set_identity<T>(Bk.J, tid);
        bar();
        if (wave < 3 && lane == 0) manifold_block(wave, 1, K.x, Bk.xp, K.dxo, nullptr, Bk.J);
    }
    bar();
    congruence<T>(Bk.B, Bk.J, Bk.P, tid);  // B = L_ = J2 P_ J2^T
    for (int e = tid; e < NS * NS; e += T) {   // A = P_ J2^T (J2 = identity outside its three blocks: at most 3 terms each)
        const int i = e / NS, j = e % NS;
        int jb, nb;
        blk_range(j, jb, nb);
        double t = 0.0;
        for (int b = 0; b < nb; ++b) t += Bk.P[i][jb + b] * Bk.J[j][jb + b];
        Bk.A[i][j] = t;
    }
    if (tid < NS * NW) {                    // K_x[:, :NW] = X HTH (columns >= NW are zero)
        const int i = tid / NW, c = tid % NW;
        double t = 0.0;
        for (int j = 0; j < NW; ++j) t += K.X[i][j] * K.HTH[j][c];
        Bk.Kx[i][c] = t;
    }
    bar();
    if (tid < NS * NW) {                    // K_x <- J2 K_x (rows; block structure again)
        const int i = tid / NW, c = tid % NW;
        int ib, nb;
        blk_range(i, ib, nb);
        double s = 0;
        for (int r = 0; r < nb; ++r) s += Bk.J[i][ib + r] * Bk.Kx[ib + r][c];
        Bk.XA[i][c] = s;
    }
    bar();
    uint32_t mine = 0u;   // this thread's share of the mailbox checksum (xor: order-free), reduced per wavefront before LDS
    for (int e = tid; e < NS * NS; e += T) {
        const int i = e / NS, j = e % NS;
        double s = 0;
        for (int c = 0; c < NW; ++c) s += Bk.XA[i][c] * Bk.A[c][j];
        const double pv = Bk.B[i][j] - s;
        kf->P_post[e] = pv;
        LV_IO_STORE(&io->P_post[e], pv);
        mine ^= mailbox_mix(pv, (uint32_t)e);
    }
    if (tid < NX) mine ^= mailbox_mix(K.x[tid], 1000u + (uint32_t)tid);
    if (tid == 0) mine ^= mailbox_mix((double)(pass + 1), 2000u);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mine ^= (uint32_t)__shfl_xor((int)mine, m);
    if (lane == 0 && mine) atomicXor(&Bk.chk, mine);
    // every mailbox store is a system-scope write-through store: once a lane's stores have retired they are visible to
    // the host; the host verifies the checksum (lv_update_end)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    bar();
    if (tid == 0) {
        uint32_t chk = Bk.chk;
        if (chk == MAILBOX_UNCHECKED) chk = 0u;
        __hip_atomic_store(&io->seqcheck, ((unsigned long long)chk << 32) | (unsigned long long)(uint32_t)prm_seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

// What the books need of the prologue solve, copied out of the solve scratch (all threads of the workgroup call; the caller
// provides the barrier that follows)
__device__ inline void keep_solve(KeepLds& K, const SolveLds& L, int tid) {
    for (int e = tid; e < SUMS_LEN; e += PK_THREADS) K.rec[e] = L.rec[e];
    for (int e = tid; e < 144; e += PK_THREADS) K.HTH[e / 12][e % 12] = L.HTH[e / 12][e % 12];
    for (int e = tid; e < NS * 12; e += PK_THREADS) K.X[e / 12][e % 12] = L.X[e / 12][e % 12];
    if (tid < NS) K.dxo[tid] = L.dxo[tid];
    if (tid < NX) K.x[tid] = L.x[tid];
    if (tid == 0) { K.last = L.last; K.n_valid0 = L.n_valid0; K.t_new = L.t_new; K.kf_iter = L.kf_iter; K.pass = L.pass; }
}

}  // namespace lv

// back to the build default for the includer's own code (the f32 path is bit-exact against FMA-free arithmetic)
#pragma clang fp contract(off)
