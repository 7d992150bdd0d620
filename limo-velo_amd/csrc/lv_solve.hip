// lv_solve.hip — the N-independent tail of each IKFoM pass, kept on the device so that the
// iterated update never leaves the GPU:
//   reduce_groups_kernel : 1024 block partials -> 32 group records   (fixed summation order;
//                          a single CU only pulls ~25-60 GB/s, so the 786 KB of partials are first
//                          folded by 32 workgroups in parallel)
//   reduce_final_kernel  : group records -> the one 96-double record (multi-GPU path: this is what
//                          RCCL all-reduces)
//   solve_kernel         : esekf::update_iterated_dyn_share_modified's per-pass algebra
//                          [IKFoM absent from the reference mount; UPSTREAM-RECALL of hku-mars/IKFoM
//                          esekfom.hpp; call site reference src/Modules/Localizator.cpp:132] —
//                          manifold projections, gain, boxplus, LIMITS test (src/main.cpp:145),
//                          posterior covariance on the terminal pass — and the f32 pose constants of
//                          the next pass (State(const state_ikfom&, double), src/Objects/State.cpp:51-62).
//
// Gain algebra.  Upstream computes  P_inv = ((P_/R)^-1 + E HTH E^T)^-1  (two 23x23 inverses) and uses
// only X = P_inv[:, 0:12].  Multiplying the defining equation by Pr = P_/R gives the block system
//     (I + Pr11 HTH) X_top = Pr11,   X_bot = Pr21 (I - HTH X_top)
// whose numerically safe form is  X_top = (Pr11^-1 + HTH)^-1,  X_bot = Pr21 Pr11^-1 X_top :
// two 12x12 SPD inversions (unpivoted Gauss-Jordan) instead of two 23x23 ones, no cancellation
// (the naive S = I + Pr11 HTH solve loses ~6 digits in the posterior covariance; measured in
// tests/test_oracle.py::test_gain_form_numerics).  Algebraically identical to upstream, rounding-level different.
#include <cstring>

#include "lv_host.hpp"
#include "lv_solve_dev.hpp"

// the 23-dof algebra of this file is compared with the oracle at 1e-9, not bitwise (see lv_manifold.hpp)
#pragma clang fp contract(fast)

namespace lv {

// from_host: the state and its covariance travel as kernel arguments (4.4 KB in the dispatch's kernarg buffer: no
// upload on the stream, no read across PCIe inside the kernel — lv_update); otherwise both are already in kf
// (resident filter).
struct StateArg {
    double v[NX];
    double P[NS * NS];
};
// filt != nullptr (lv_correct): the state and covariance come from the resident filter — or, when that filter is still the
// posterior of the previous update and has not been copied out of kf yet (filt_in_kf), from kf->x / kf->P_post — and are
// installed in kf->x / kf->P_prop as filter_to_kf_kernel would have done in a launch of its own.
__global__ __launch_bounds__(576) void kf_begin_kernel(KfDev* kf, KfHostIO* io, int from_host, StateArg xin, const FilterDev* filt, int filt_in_kf) {
    __shared__ double s_x[NX];
    __shared__ double s_rot[4][9];
    __shared__ float s_tmp[8];
    const int tid = threadIdx.x;
    double p = 0.0;
    const bool install = from_host || filt != nullptr;
    if (tid < NS * NS) p = from_host ? xin.P[tid] : (filt ? (filt_in_kf ? kf->P_post[tid] : filt->P[tid]) : kf->P_prop[tid]);
    if (tid < NX) {
        const double v = from_host ? xin.v[tid] : ((filt && !filt_in_kf) ? filt->x[tid] : kf->x[tid]);
        if (install) kf->x[tid] = v;
        kf->x_prop[tid] = v;
        io->x[tid] = v;
        s_x[tid] = v;
    }
    if (tid == 0) {
        io->passes = 0;
        kf->t = 0;
        kf->iter = -1;  // upstream loop starts at i = -1
        kf->done = 0;
        kf->passes = 0;
        // (fallback_queries only ever grows on the device: the host reports differences — an update that starts in its
        // first search launch has no kernel before it that could reset the counter)
    }
    __syncthreads();
    // compute_pose_consts (lv_device.hpp) spread over lanes: four quaternion -> matrix conversions, then the
    // composed transforms (same operations, same order => same bits as the serial form)
    if (tid >= 64 && tid < 68) {
        const int w = tid - 64;                // 0: rot, 1: offset_R_L_I, 2: conj(rot), 3: conj(offset_R_L_I)
        const int q = (w & 1) ? 7 : 3;
        const double sg = (w & 2) ? -1.0 : 1.0;
        const double qq[4] = {sg * s_x[q], sg * s_x[q + 1], sg * s_x[q + 2], s_x[q + 3]};
        quat_to_rot(qq, &s_rot[w][0]);
    }
    __syncthreads();
    if (tid >= 64 && tid < 128) pose_consts_stage_a(tid - 64, s_x, s_rot, &kf->pose, s_tmp);
    __syncthreads();
    if (tid >= 64 && tid < 128) pose_consts_stage_b(tid - 64, s_rot, &kf->pose, s_tmp);
    if (tid < NS * NS) {
        if (install) kf->P_prop[tid] = p;
        kf->P_post[tid] = p;
        io->P_post[tid] = p;   // an update without a terminal pass returns the propagated covariance
    }
}

// ---- staged, order-fixed reduction of the block partials ------------------------------------------
constexpr int RG_THREADS = 384;  // 96 outputs x 4 segments
__global__ __launch_bounds__(RG_THREADS) void reduce_groups_kernel(const double* __restrict__ partials, int nblocks, int group,
                                                                    double* __restrict__ groups, const KfDev* kf) {
    __shared__ double s_seg[4][SUMS_LEN];
    if (kf->done) return;
    const int o = threadIdx.x % SUMS_LEN, seg = threadIdx.x / SUMS_LEN;
    const int b0 = blockIdx.x * group;
    int b1 = b0 + group;
    if (b1 > nblocks) b1 = nblocks;
    double s = 0.0;
    for (int b = b0 + seg; b < b1; b += 4) s += partials[(size_t)b * SUMS_LEN + o];
    s_seg[seg][o] = s;
    __syncthreads();
    if (seg == 0) groups[(size_t)blockIdx.x * SUMS_LEN + o] = ((s_seg[0][o] + s_seg[1][o]) + s_seg[2][o]) + s_seg[3][o];
}

constexpr int FOLD_THREADS = 576;
constexpr int FOLD_PARTS = FOLD_THREADS / SUMS_LEN;  // 6
constexpr int FOLD_DEPTH = 44;                       // x 6 parts: up to 264 records folded in one memory round trip
// Fold of nrec <= FOLD_PARTS * FOLD_DEPTH records by a 576-thread workgroup, fixed order: thread (o, part) owns
// records part, part + 6, ... and sums them sequentially; the 6 part sums are combined pairwise.  fold_issue
// only issues the loads (so a caller can overlap them with its other reads), fold_finish needs a
// __syncthreads() between its two halves and returns the total to threads tid < SUMS_LEN.
__device__ __forceinline__ void fold_issue(double (&fv)[FOLD_DEPTH], const double* __restrict__ recs, int nrec, int tid) {
    const int fo = tid % SUMS_LEN, fpart = tid / SUMS_LEN;
#pragma unroll
    for (int i = 0; i < FOLD_DEPTH; ++i) {
        const int g = fpart + FOLD_PARTS * i;
        fv[i] = g < nrec ? recs[(size_t)g * SUMS_LEN + fo] : 0.0;
    }
}
__device__ __forceinline__ void fold_stage(const double (&fv)[FOLD_DEPTH], double (*s_part)[SUMS_LEN], int tid) {
    // four interleaved running sums, combined pairwise: a quarter of the dependent-add chain, still a fixed order
    double a0 = fv[0], a1 = fv[1], a2 = fv[2], a3 = fv[3];
#pragma unroll
    for (int i = 4; i + 3 < FOLD_DEPTH; i += 4) { a0 += fv[i]; a1 += fv[i + 1]; a2 += fv[i + 2]; a3 += fv[i + 3]; }
    static_assert(FOLD_DEPTH % 4 == 0, "fold_stage assumes a multiple of four");
    s_part[tid / SUMS_LEN][tid % SUMS_LEN] = (a0 + a1) + (a2 + a3);
}
__device__ __forceinline__ double fold_total(const double (*s_part)[SUMS_LEN], int o) {
    return ((s_part[0][o] + s_part[1][o]) + (s_part[2][o] + s_part[3][o])) + (s_part[4][o] + s_part[5][o]);
}

// multi-GPU path: the block partials (or group records) of this rank -> its one 96-double record, which RCCL
// then all-reduces in place
__global__ __launch_bounds__(FOLD_THREADS) void reduce_final_kernel(const double* __restrict__ recs, int nrec,
                                                                   double* __restrict__ sums, const KfDev* kf) {
    __shared__ double s_part[FOLD_PARTS][SUMS_LEN];
    const int tid = threadIdx.x;
    double fv[FOLD_DEPTH];
    if (nrec <= FOLD_PARTS * FOLD_DEPTH) fold_issue(fv, recs, nrec, tid);
    if (kf->done) return;
    if (nrec <= FOLD_PARTS * FOLD_DEPTH) {
        fold_stage(fv, s_part, tid);
        __syncthreads();
        if (tid < SUMS_LEN) sums[tid] = fold_total(s_part, tid);
    } else if (tid < SUMS_LEN) {
        double s = 0.0;
        for (int g = 0; g < nrec; ++g) s += recs[(size_t)g * SUMS_LEN + tid];
        sums[tid] = s;
    }
}

// ---- solve ------------------------------------------------------------------------------------------
constexpr int SOLVE_THREADS = FOLD_THREADS;
// store to the pinned host mailbox that bypasses the device caches (system-scope, relaxed)
#define IO_STORE(ptr, val) __hip_atomic_store((ptr), (val), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM)

// NW = number of Jacobian columns that can be non-zero: 6 without extrinsic estimation (H^T H lives in
// the leading 6x6 block, only P_inv[:, 0:6] is needed), 12 with it.
template <int NW>
__global__ __launch_bounds__(SOLVE_THREADS) void solve_kernel(KfDev* kf, KfHostIO* io, const double* __restrict__ recs, int nrec,
                                                              double* __restrict__ sums_out, SolveParams prm) {
    __shared__ double sP[NS][LD], sA[NS][LD], sB[NS][LD], sJ[NS][LD];
    __shared__ double sW[2][12][13], sT[12][12];
    __shared__ double sX[NS][12], sG[NS][12], sKx[NS][12], sHTH[12][12], sHTh[12];
    __shared__ double sv[12];
    __shared__ double sdxnew[NS], sdxo[NS], sx[NX], sxp[NX], srec[SUMS_LEN];
    __shared__ double s_part[FOLD_PARTS][SUMS_LEN];
    __shared__ double sRot[4][9];
    __shared__ PoseConsts s_pose;
    __shared__ float s_ptmp[8];
    __shared__ int s_last, s_conv;
    __shared__ uint32_t s_chk;
    const int tid = threadIdx.x;
    const int wave = tid >> 6, lane = tid & 63;
    // every global read of this kernel is issued up front (one memory round trip): the records, x, x_prop and
    // the prepared half — independent of the done / passes words read next
    double fv[FOLD_DEPTH];
    if (nrec <= FOLD_PARTS * FOLD_DEPTH) fold_issue(fv, recs, nrec, tid);
    if (tid >= 128 && tid < 128 + NX) { sx[tid - 128] = kf->x[tid - 128]; sxp[tid - 128] = kf->x_prop[tid - 128]; }
    // the record-independent half of this pass was computed by fit_reduce_kernel's extra workgroup (solve_prep)
    if (tid < NS * NS) sP[tid / NS][tid % NS] = kf->prep_P[tid];
    if (tid < NS) sdxnew[tid] = kf->prep_dxnew[tid];
    if (tid >= 192 && tid < 192 + NW * NW) sG[(tid - 192) / NW][(tid - 192) % NW] = kf->prep_A1[tid - 192];
    if (kf->done) return;
    const int pass = kf->passes;
    // loop bookkeeping words, read once up front (a late global read would sit on the critical path of one lane)
    const int kf_t = kf->t, kf_iter = kf->iter, kf_fallback = kf->fallback_queries;
#define SV_STAMP(i) do { if (tid == 0 && pass < MAX_PASSES) kf->solve_clk[pass * 16 + (i)] = clock64(); } while (0)
    SV_STAMP(0);
    __syncthreads();
    if (nrec <= FOLD_PARTS * FOLD_DEPTH) fold_stage(fv, s_part, tid);
    __syncthreads();
    if (tid < SUMS_LEN) {
        double s = 0.0;
        if (nrec <= FOLD_PARTS * FOLD_DEPTH) {
            s = fold_total(s_part, tid);
        } else {
            for (int g = 0; g < nrec; ++g) s += recs[(size_t)g * SUMS_LEN + tid];
        }
        srec[tid] = s;
        if (sums_out) sums_out[tid] = s;
        if (pass < MAX_PASSES) kf->sums_log[pass * SUMS_LEN + tid] = s;
    }
    if (tid == 200) { s_conv = 1; s_chk = 0u; }
    __syncthreads();
    if (tid < 144) {
        const int a = tid / 12, b = tid % 12;
        const int lo = a < b ? a : b, hi = a < b ? b : a;
        sHTH[a][b] = srec[lo * 12 - lo * (lo - 1) / 2 + (hi - lo)];
    }
    if (tid >= 192 && tid < 204) sHTh[tid - 192] = srec[78 + tid - 192];
    const double n_valid = srec[90];
    if (n_valid == 0.0) {  // h_share_model: dyn_share.valid = false -> `continue`
        if (tid == 0) {
            if (pass < MAX_PASSES) {
                for (int i = 0; i < NS; ++i) kf->trace[pass * 49 + i] = 0.0;
                for (int i = 0; i < NX; ++i) kf->trace[pass * 49 + NS + i] = kf->x[i];
            }
            kf->passes = pass + 1;
            kf->iter = kf_iter + 1;
            if (kf_iter + 1 >= prm.maximum_iter) {
                kf->done = 1;
                // the update ends on a pass without matches: the mailbox gets the state the earlier passes left
                // (the passes in between store nothing across PCIe)
                IO_STORE(&io->passes, pass + 1);
                IO_STORE(&io->fallback_queries, kf_fallback);
                for (int i = 0; i < NX; ++i) IO_STORE(&io->x[i], kf->x[i]);
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // the write-through mailbox stores above have retired
                // final: the host may stop waiting (no checksum on this rare path: the host synchronises the stream)
                __hip_atomic_store(&io->seqcheck, ((unsigned long long)MAILBOX_UNCHECKED << 32) | (unsigned long long)(uint32_t)prm.seq,
                                   __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
        }
        return;
    }

    SV_STAMP(1);
    __syncthreads();   // sHTH / sHTh complete
    if (prm.degeneracy_mode) {   // the fork's degeneracy stage (hook; off by default)
        if (tid == 0) degeneracy_stage(sHTH, sHTh, prm.degeneracy_mode, prm.degeneracy_threshold, &kf->degen_eig[(pass < MAX_PASSES ? pass : 0) * 6]);
        __syncthreads();
    }
    // P_ (sP), dx_new and A1 = (P_/R)_ww^-1 (sG) come from solve_prep.  Pr = P_/R -> sA
    if (tid < NS * NS) sA[tid / NS][tid % NS] = sP[tid / NS][tid % NS] * prm.R_inv;
    // X = P_inv[:, 0:NW]:  X_top = (Pr_ww^-1 + HTH_ww)^-1 (unpivoted Gauss-Jordan inversion of an SPD NW x NW
    // matrix),  X_bot = Pr[NW:, 0:NW] Pr_ww^-1 X_top
    int cur = 0;
    if (tid < NW * NW) {
        const int i = tid / NW, j = tid % NW;
        sW[cur][i][j] = sG[i][j] + sHTH[i][j];
    }
    // dx_ = K_h + (K_x - I) dx_new with K_h = X HTh, K_x[:, :NW] = X HTH  ==  X (HTh + HTH dx_new[:NW]) - dx_new:
    // the NW-vector v is formed here, next to the Gauss-Jordan steps; K_x itself is only needed by the terminal pass
    if (tid >= 64 && tid < 64 + NW) {
        const int i = tid - 64;
        double s = sHTh[i];
        for (int j = 0; j < NW; ++j) s += sHTH[i][j] * sdxnew[j];
        sv[i] = s;
    }
    __syncthreads();
    SV_STAMP(4);
    gj_spd<NW>(sW, cur, tid);               // sW[cur] = X_top
    SV_STAMP(5);
    if (tid < NW * NW) {                    // T = A1 X_top
        const int i = tid / NW, c = tid % NW;
        double s = 0.0;
        for (int j = 0; j < NW; ++j) s += sG[i][j] * sW[cur][j][c];
        sT[i][c] = s;
    }
    __syncthreads();
    if (tid < NS * NW) {                    // X = [X_top ; Pr[NW:, :NW] T]
        const int i = tid / NW, c = tid % NW;
        double v;
        if (i < NW) {
            v = sW[cur][i][c];
        } else {
            double s = 0.0;
            for (int j = 0; j < NW; ++j) s += sA[i][j] * sT[j][c];
            v = s;
        }
        sX[i][c] = v;
    }
    __syncthreads();
    if (tid < NS) {  // dx_ = X v - dx_new
        double s = 0.0;
        for (int j = 0; j < NW; ++j) s += sX[tid][j] * sv[j];
        const double d = s - sdxnew[tid];
        sdxo[tid] = d;
        if (fabs(d) > prm.limits[tid]) s_conv = 0;  // dyn_share.converge
    }
    __syncthreads();
    SV_STAMP(6);
    // x_.boxplus(dx_)
    if (wave < 3 && lane == 0) boxplus_block(wave, sx, sdxo);
    if (wave == 3 && lane < 15) {
        const int dof = lane < 3 ? lane : lane + 6;
        sx[vect_state_index(dof)] += sdxo[dof];
    }
    if (tid == 256) {
        int t = kf_t;
        if (s_conv) t++;
        kf->t = t;
        s_last = (t > 1 || kf_iter == prm.maximum_iter - 1) ? 1 : 0;
    }
    __syncthreads();
    SV_STAMP(7);
    const int last = s_last;
    if (tid < NX) {
        kf->x[tid] = sx[tid];
        if (last) IO_STORE(&io->x[tid], sx[tid]);   // the mailbox (write-through across PCIe) only hears from the pass that ends the update
    }
    if (tid >= 64 && tid < 64 + 49 && pass < MAX_PASSES) {
        const int e = tid - 64;
        kf->trace[pass * 49 + e] = e < NS ? sdxo[e] : sx[e - NS];
    }
    if (!last && tid >= 320 && tid < 324) {  // the four rotation matrices of the next pass, one lane each
        const int w = tid - 320;               // 0: rot, 1: offset_R_L_I, 2: conj(rot), 3: conj(offset_R_L_I)
        const int q = (w & 1) ? 7 : 3;
        const double sg = (w & 2) ? -1.0 : 1.0;
        const double qq[4] = {sg * sx[q], sg * sx[q + 1], sg * sx[q + 2], sx[q + 3]};
        quat_to_rot(qq, &sRot[w][0]);
    }
    __syncthreads();
    if (tid == 0) {
        kf->passes = pass + 1;
        kf->iter = kf_iter + 1;
        if (last) {
            kf->done = 1;
            IO_STORE(&io->passes, pass + 1);
            IO_STORE(&io->fallback_queries, kf_fallback);
        }
    }
    if (!last) {  // the next pass' constants: finish_pose_consts spread over one wavefront, then coalesced stores
        if (tid >= 64 && tid < 128) pose_consts_stage_a(tid - 64, sx, sRot, &s_pose, s_ptmp);
        __syncthreads();
        if (tid >= 64 && tid < 128) pose_consts_stage_b(tid - 64, sRot, &s_pose, s_ptmp);
        __syncthreads();
        constexpr int NW32 = (int)(sizeof(PoseConsts) / 4);
        const uint32_t* src = reinterpret_cast<const uint32_t*>(&s_pose);
        uint32_t* dst = reinterpret_cast<uint32_t*>(&kf->pose);
        if (tid < NW32) dst[tid] = src[tid];
    }
    SV_STAMP(8);
    if (!last) return;

    // terminal pass: L_ = J2 P_ J2^T, K_x rows projected, P_ <- P_ J2^T, P = L_ - K_x[:, :12] P_[0:12, :]
    __syncthreads();
    set_identity<SOLVE_THREADS>(sJ, tid);
    __syncthreads();
    if (wave < 3 && lane == 0) manifold_block(wave, 1, sx, sxp, sdxo, nullptr, sJ);
    __syncthreads();
    congruence<SOLVE_THREADS>(sB, sJ, sP, tid);  // sB = L_ = J2 P_ J2^T
    mm<SOLVE_THREADS>(sA, sP, sJ, true, tid);    // sA = P_ J2^T
    __syncthreads();
    if (tid < NS * NW) {         // K_x[:, :NW] = X HTH (columns >= NW are zero)
        const int i = tid / NW, c = tid % NW;
        double t = 0.0;
        for (int j = 0; j < NW; ++j) t += sX[i][j] * sHTH[j][c];
        sKx[i][c] = t;
    }
    __syncthreads();
    if (tid < NS * NW) {         // K_x <- J2 K_x (rows)
        const int i = tid / NW, c = tid % NW;
        double s = 0;
        for (int r = 0; r < NS; ++r) s += sJ[i][r] * sKx[r][c];
        sX[i][c] = s;
    }
    __syncthreads();
    uint32_t mine = 0u;   // this thread's share of the mailbox checksum (xor: order-free), reduced per wavefront before LDS:
                          // 556 atomics on one LDS word were 2.7 us of the terminal pass
    if (tid < NS * NS) {
        const int i = tid / NS, j = tid % NS;
        double s = 0;
        for (int c = 0; c < NW; ++c) s += sX[i][c] * sA[c][j];
        const double pv = sB[i][j] - s;
        kf->P_post[tid] = pv;
        IO_STORE(&io->P_post[tid], pv);
        mine ^= mailbox_mix(pv, (uint32_t)tid);
    }
    if (tid < NX) mine ^= mailbox_mix(sx[tid], 1000u + (uint32_t)tid);
    if (tid == 0) mine ^= mailbox_mix((double)(pass + 1), 2000u);
#pragma unroll
    for (int m = 32; m >= 1; m >>= 1) mine ^= (uint32_t)__shfl_xor((int)mine, m);
    if (lane == 0 && mine) atomicXor(&s_chk, mine);
    SV_STAMP(9);
    // the update is final (kf->done was set above): every result store is ordered before the sequence number
    // the host waits on
    // (every mailbox store of this kernel is a system-scope write-through store, IO_STORE: once a lane's stores
    // have retired they are visible to the host; a system-scope release fence would also write back the whole
    // L2, ~4 us — and plain stores without it were observed to arrive after the sequence number)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        uint32_t chk = s_chk;
        if (chk == MAILBOX_UNCHECKED) chk = 0u;   // (the value that means "unchecked" is not used as a checksum; the host maps it likewise)
        __hip_atomic_store(&io->seqcheck, ((unsigned long long)chk << 32) | (unsigned long long)(uint32_t)prm.seq, __ATOMIC_RELAXED,
                           __HIP_MEMORY_SCOPE_SYSTEM);
    }
#undef SV_STAMP
}

int launch_kf_begin(hipStream_t stream, KfDev* kf, KfHostIO* io, const double* x_host, const FilterDev* filt, int filt_in_kf) {
    StateArg xin;   // 4.4 KB of kernel arguments (copied into the kernarg buffer by the launch)
    if (x_host) { std::memcpy(xin.v, x_host, sizeof(xin.v)); std::memcpy(xin.P, x_host + NX, sizeof(xin.P)); }   // KfHostIO: x_in, P_in contiguous
    else std::memset(&xin, 0, sizeof(xin));
    hipLaunchKernelGGL(kf_begin_kernel, dim3(1), dim3(576), 0, stream, kf, io, x_host ? 1 : 0, xin, filt, filt_in_kf);
    LV_HIP(hipGetLastError());
    return LV_OK;
}
int launch_reduce_groups(hipStream_t stream, const double* partials, int nblocks, double* groups, int* ngroups_out, KfDev* kf) {
    const int group = PARTIAL_GROUP;
    const int ngroups = (nblocks + group - 1) / group;
    hipLaunchKernelGGL(reduce_groups_kernel, dim3(ngroups), dim3(RG_THREADS), 0, stream, partials, nblocks, group, groups, kf);
    LV_HIP(hipGetLastError());
    *ngroups_out = ngroups;
    return LV_OK;
}
int solve_direct_records() { return FOLD_PARTS * FOLD_DEPTH; }
int launch_reduce_final(hipStream_t stream, const double* groups, int ngroups, double* sums, KfDev* kf) {
    hipLaunchKernelGGL(reduce_final_kernel, dim3(1), dim3(FOLD_THREADS), 0, stream, groups, ngroups, sums, kf);
    LV_HIP(hipGetLastError());
    return LV_OK;
}
int launch_solve(hipStream_t stream, KfDev* kf, KfHostIO* io, const double* recs, int nrec, double* sums_out, const SolveParams& prm) {
    if (prm.estimate_extrinsics)
        hipLaunchKernelGGL((solve_kernel<12>), dim3(1), dim3(SOLVE_THREADS), 0, stream, kf, io, recs, nrec, sums_out, prm);
    else
        hipLaunchKernelGGL((solve_kernel<6>), dim3(1), dim3(SOLVE_THREADS), 0, stream, kf, io, recs, nrec, sums_out, prm);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

}  // namespace lv
