// lv_peer.hip — the exchange step of the one-launch-per-pass multi-GPU form WITHOUT a collective library: every rank keeps
// its gather buffers in ONE device allocation that the other ranks of the node map through HIP IPC (xGMI peer access); after
// a pass a single small kernel publishes "my partials of launch #seq are in memory" in the rank's flag word and PULLS every
// other rank's slot straight out of that rank's buffer as soon as its flag says so — a one-shot peer read per rank (SURVEY
// §5 / §8e "one-shot p2p, not ring") instead of ncclAllGather's launch + ring steps.  The next launch's prologue then folds
// the local, now complete, gather buffer exactly as in the RCCL form: same slots, same fixed order, bit-identical ranks.
//
// Why a rank may overwrite its slot: its slot of buffer (p & 1) is rewritten by ITS launch p + 2, which is enqueued behind
// its pull #(p + 1), which waits for every peer's flag #(p + 1), which a peer publishes in its own pull #(p + 1) — a kernel
// that runs after that peer's pull #p, the last reader of the slot.  Flags only grow (one counter per context).
// A pull that waits longer than the give-up time (LV_PEER_TIMEOUT_MS in the environment, default 2000 ms: ordinary host skew
// between processes — a first-touch allocation, Python's collector, a 20 ms note fallback — must never trip it) sets the
// context's sticky status word (pinned host memory: the host reads it without a copy) AND poisons this rank's own flag (top
// bit), so that every rank waiting for this one stops as well: all ranks fail the same update instead of diverging.  The host
// side (lv_api.hip) then refuses to adopt that update's posterior and fails every later call with LV_ESTATE.
// Memory: the gather slots are written by a pass kernel and read by peers only after that kernel has ended and a LATER kernel
// of the same stream has published the flag, i.e. across a kernel boundary — correct in plain (coarse-grained) memory by the
// argument written out in peer_gather_kernel; since round 5 they are nevertheless allocated fine-grained when the runtime can
// export such memory (one assumption fewer on the first real multi-GPU run).  The flag word is polled across devices in the
// middle of a kernel and lives in its own fine-grained allocation (hipExtMallocWithFlags; a plain one, with a warning on
// stderr, only if the runtime refuses to export a fine-grained one; LV_PEER_REQUIRE_FINE=1 turns the warning into an error).
// EXPERIMENTAL, opt-in (lv_comm_peer_export / lv_comm_peer_init): proven with two and four processes on ONE GPU
// (tests/test_gpu_distributed.py), where all ranks share one L2; NOT yet run across GPUs — no multi-GPU node was available to
// rounds 1-5, and no scaling curve exists.
#include <cstdio>
#include <cstdlib>
#include <cstring>

#include "lv_host.hpp"

namespace lv {

namespace {
constexpr unsigned long long PEER_POISON = 1ull << 63;
struct PeerArgs {
    double* local;                          // this rank's gather buffer (parity of the launch)
    const double* peer[LV_PEER_MAX];        // the same buffer of every rank (self: local)
    const unsigned long long* pflag[LV_PEER_MAX];
    unsigned long long* my_flag;
    unsigned long long seq;
    size_t slot;                            // doubles per rank
    int rank, world;
    uint32_t* status;                       // host-mapped, sticky
    long long timeout_ticks;
};

__global__ __launch_bounds__(256) void peer_gather_kernel(PeerArgs a) {
    const int b = blockIdx.x;   // one workgroup per rank of the node
    if (b == a.rank) {          // publish: this rank's partials of launch #seq are complete (the pass kernel ended before this one began)
        if (threadIdx.x == 0) {
            // (a context that has already failed keeps telling its peers so)
            const unsigned long long bad = __hip_atomic_load(a.status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) ? PEER_POISON : 0ull;
            // fetch_max, not a store: a pulling workgroup of this or of the previous exchange kernel may have OR-ed PEER_POISON into
            // this word already; a plain store landing after that would erase the poison until the next launch re-read the status
            // word, and peers could adopt an update this rank has failed.  The poison is the top bit, so max() keeps it;
            // flags only grow, so max() == store otherwise.  RELEASE at system scope: the write-back of this XCD's L2 precedes the
            // flag (the pass kernel's own stores were written back from every XCD's L2 when THAT kernel ended — see below).
            __hip_atomic_fetch_max(a.my_flag, a.seq | bad, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        return;
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int ok = 1;
        for (;;) {
            const unsigned long long v = __hip_atomic_load(a.pflag[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM);
            if (v & PEER_POISON) { ok = 0; break; }                     // that rank failed: so does this one, in the same update
            if (v >= a.seq) break;
            if (wall_clock64() - t0 > a.timeout_ticks) { ok = 0; break; }   // the peer is gone
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) {
            __hip_atomic_store(a.status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            __hip_atomic_fetch_or(a.my_flag, PEER_POISON, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    // Why the slot reads below see the peer's partials (the argument the flag protocol rests on):
    //  writer side — the slot was written by the peer's pass kernel with ordinary stores; that kernel ENDED before the peer's
    //    exchange kernel (same stream) began, and the end-of-kernel release writes every XCD's dirty L2 lines back to memory;
    //    the peer's publish is a system-scope RELEASE after that, so flag >= seq implies the slot is in the peer's HBM / MALL
    //    (memory-side, coherent for every requester);
    //  reader side — thread 0's ACQUIRE load of the flag at system scope orders (and the barrier above extends that to the
    //    workgroup) the loads below, which are themselves system-scope atomic loads: they bypass this GPU's L2 / TCP (sc0 sc1),
    //    so a line of the peer buffer cached from an earlier exchange can never be served.  Nothing here depends on the two ranks
    //    sharing an L2 — that is merely the only configuration rounds 1-5 could run (one GPU per lease).
    //  The slots are allocated fine-grained when the runtime exports such memory (peer_export), which removes the dependence on
    //  the end-of-kernel write-back as well; coarse-grained is the fallback and is reported (PeerSet::buf_fine, stderr).
    const double* src = a.peer[b] + (size_t)b * a.slot;
    double* dst = a.local + (size_t)b * a.slot;
    for (size_t i = threadIdx.x; i < a.slot; i += blockDim.x)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (never from a stale cache line)
}
}  // namespace

int peer_export(PeerSet& P, size_t cap_doubles, void* handle_blob) {
    if (P.local_alloc) { set_error("peer buffers already exported"); return LV_ESTATE; }
    const size_t bytes = 2 * cap_doubles * sizeof(double);
    hipIpcMemHandle_t h[2];
    static_assert(sizeof(h) == LV_PEER_BLOB, "two 64-byte HIP IPC handles");
    // the gather slots: fine-grained if the runtime can allocate AND export such memory (LV_PEER_COARSE_SLOTS=1 keeps round 4's
    // plain allocation), else coarse-grained (correct across a kernel boundary: see peer_gather_kernel)
    P.buf_fine = false;
    const char* coarse_env = getenv("LV_PEER_COARSE_SLOTS");
    if (!(coarse_env && atoi(coarse_env) != 0) && hipExtMallocWithFlags(&P.local_alloc, bytes, hipDeviceMallocFinegrained) == hipSuccess && P.local_alloc) {
        hipIpcMemHandle_t probe;
        if (hipIpcGetMemHandle(&probe, P.local_alloc) == hipSuccess) P.buf_fine = true;
        else { hipFree(P.local_alloc); P.local_alloc = nullptr; }
    }
    (void)hipGetLastError();
    if (!P.buf_fine) LV_HIP(hipMalloc(&P.local_alloc, bytes));
    LV_HIP(hipMemset(P.local_alloc, 0, bytes));
    // the flag word: fine-grained if the runtime can allocate AND export such memory
    P.flag_fine = false;
    if (hipExtMallocWithFlags(&P.flag_alloc, 4096, hipDeviceMallocFinegrained) == hipSuccess && P.flag_alloc) {
        if (hipIpcGetMemHandle(&h[1], P.flag_alloc) == hipSuccess) P.flag_fine = true;
        else { hipFree(P.flag_alloc); P.flag_alloc = nullptr; }
    }
    (void)hipGetLastError();
    if (!P.flag_fine) {
        // the flag is polled across devices in the MIDDLE of a kernel: in coarse-grained memory that is only coherent when the
        // poller bypasses its caches (it does: system-scope atomic loads) and the writer's store reaches memory at once (a
        // system-scope atomic does).  It has worked wherever it was tried, but it is outside what the memory model promises: say so.
        fprintf(stderr, "[limovelo_hip] warning: the runtime exports no fine-grained memory; the peer exchange's flag word lives in "
                        "coarse-grained memory (set LV_PEER_REQUIRE_FINE=1 to refuse instead)\n");
        if (const char* e = getenv("LV_PEER_REQUIRE_FINE")) if (atoi(e) != 0) {
            hipFree(P.local_alloc); P.local_alloc = nullptr;
            set_error("lv_comm_peer_export: no exportable fine-grained allocation for the flag word (LV_PEER_REQUIRE_FINE)");
            return LV_EHIP;
        }
        LV_HIP(hipMalloc(&P.flag_alloc, 4096));
        LV_HIP(hipIpcGetMemHandle(&h[1], P.flag_alloc));
    }
    LV_HIP(hipMemset(P.flag_alloc, 0, 4096));
    LV_HIP(hipHostMalloc((void**)&P.h_status, 64, hipHostMallocMapped));
    *P.h_status = 0u;
    LV_HIP(hipHostGetDevicePointer((void**)&P.d_status, P.h_status, 0));
    LV_HIP(hipDeviceSynchronize());
    P.cap = cap_doubles;
    P.buf[0] = static_cast<double*>(P.local_alloc);
    P.buf[1] = P.buf[0] + cap_doubles;
    P.flag = static_cast<unsigned long long*>(P.flag_alloc);
    LV_HIP(hipIpcGetMemHandle(&h[0], P.local_alloc));
    std::memcpy(handle_blob, h, sizeof(h));
    double ms = 2000.0;
    if (const char* e = getenv("LV_PEER_TIMEOUT_MS")) { const double v = atof(e); if (v > 0.0) ms = v; }
    P.timeout_ticks = (long long)(ms * 1e5);   // wall_clock64: 100 MHz
    return LV_OK;
}

int peer_init(PeerSet& P, int rank, int world, const void* handles) {
    if (!P.local_alloc) { set_error("lv_comm_peer_export first"); return LV_ESTATE; }
    if (P.active) { set_error("the peer exchange of this context is already set up (lv_comm_destroy first)"); return LV_ESTATE; }
    if (world < 1 || world > LV_PEER_MAX || rank < 0 || rank >= world) { set_error("peer exchange: rank %d of %d (at most %d ranks)", rank, world, LV_PEER_MAX); return LV_EINVAL; }
    P.rank = rank;
    P.world = world;
    for (int r = 0; r < world; ++r) {
        void* base = P.local_alloc;
        void* fbase = P.flag_alloc;
        if (r != rank) {
            hipIpcMemHandle_t h[2];
            std::memcpy(h, static_cast<const char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
            base = nullptr;
            LV_HIP(hipIpcOpenMemHandle(&base, h[0], hipIpcMemLazyEnablePeerAccess));
            P.mapped[r] = base;
            fbase = nullptr;
            LV_HIP(hipIpcOpenMemHandle(&fbase, h[1], hipIpcMemLazyEnablePeerAccess));
            P.mapped_flag[r] = fbase;
        }
        double* b0 = static_cast<double*>(base);
        P.peer_buf[0][r] = b0;
        P.peer_buf[1][r] = b0 + P.cap;
        P.peer_flag[r] = static_cast<unsigned long long*>(fbase);
    }
    P.active = true;
    return LV_OK;
}

// after the pass kernel of a launch: publish + pull (ordered on `stream`)
int peer_gather(PeerSet& P, int parity, size_t slot_doubles, hipStream_t stream) {
    if (slot_doubles * (size_t)P.world > P.cap) { set_error("peer exchange: %zu doubles per rank x %d ranks exceed the exported buffers", slot_doubles, P.world); return LV_EINVAL; }
    PeerArgs a{};
    a.local = P.buf[parity];
    for (int r = 0; r < P.world; ++r) { a.peer[r] = P.peer_buf[parity][r]; a.pflag[r] = P.peer_flag[r]; }
    a.my_flag = P.flag;
    a.seq = ++P.seq;
    a.slot = slot_doubles;
    a.rank = P.rank;
    a.world = P.world;
    a.status = P.d_status;
    a.timeout_ticks = P.timeout_ticks;
    hipLaunchKernelGGL(peer_gather_kernel, dim3(P.world), dim3(256), 0, stream, a);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

bool peer_failed(const PeerSet& P) {
    return P.h_status && __atomic_load_n(P.h_status, __ATOMIC_ACQUIRE) != 0u;
}

void peer_close(PeerSet& P) {
    for (int r = 0; r < LV_PEER_MAX; ++r) {
        if (P.mapped[r]) { hipIpcCloseMemHandle(P.mapped[r]); P.mapped[r] = nullptr; }
        if (P.mapped_flag[r]) { hipIpcCloseMemHandle(P.mapped_flag[r]); P.mapped_flag[r] = nullptr; }
    }
    if (P.local_alloc) hipFree(P.local_alloc);
    if (P.flag_alloc) hipFree(P.flag_alloc);
    if (P.h_status) hipHostFree(P.h_status);
    P = PeerSet();
}

}  // namespace lv
