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
// A peer that never shows up (crashed rank) ends the wait after 50 ms of wall clock with the status word set: the update
// then fails with LV_ESTATE instead of hanging the GPU.
// Opt-in (lv_comm_peer_export / lv_comm_peer_init): proven with two processes on ONE GPU (tests/test_gpu_distributed.py);
// not yet run across GPUs — no multi-GPU node was available to rounds 1-3.
#include <cstring>

#include "lv_host.hpp"

namespace lv {

namespace {
struct PeerArgs {
    double* local;                          // this rank's gather buffer (parity of the launch)
    const double* peer[LV_PEER_MAX];        // the same buffer of every rank (self: local)
    const unsigned long long* pflag[LV_PEER_MAX];
    unsigned long long* my_flag;
    unsigned long long seq;
    size_t slot;                            // doubles per rank
    int rank, world;
    uint32_t* status;
};

__global__ __launch_bounds__(256) void peer_gather_kernel(PeerArgs a) {
    const int b = blockIdx.x;   // one workgroup per rank of the node
    if (b == a.rank) {          // publish: this rank's partials of launch #seq are complete (the pass kernel ended before this one began)
        if (threadIdx.x == 0) __hip_atomic_store(a.my_flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        return;
    }
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        const long long t0 = wall_clock64();
        int ok = 1;
        while (__hip_atomic_load(a.pflag[b], __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) < a.seq) {
            if (wall_clock64() - t0 > 5000000ll) { ok = 0; break; }   // 50 ms at 100 MHz: the peer is gone
            __builtin_amdgcn_s_sleep(8);
        }
        if (!ok) atomicExch(a.status, 1u);
        s_ok = ok;
    }
    __syncthreads();
    if (!s_ok) return;
    const double* src = a.peer[b] + (size_t)b * a.slot;
    double* dst = a.local + (size_t)b * a.slot;
    for (size_t i = threadIdx.x; i < a.slot; i += blockDim.x)
        dst[i] = __hip_atomic_load(src + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);   // (never from a stale cache line)
}
}  // namespace

int peer_export(PeerSet& P, size_t cap_doubles, void* handle64) {
    if (P.local_alloc) { set_error("peer buffers already exported"); return LV_ESTATE; }
    const size_t bytes = (2 * cap_doubles + 16) * sizeof(double);
    LV_HIP(hipMalloc(&P.local_alloc, bytes));
    LV_HIP(hipMemset(P.local_alloc, 0, bytes));
    LV_HIP(hipDeviceSynchronize());
    P.cap = cap_doubles;
    P.buf[0] = static_cast<double*>(P.local_alloc);
    P.buf[1] = P.buf[0] + cap_doubles;
    P.flag = reinterpret_cast<unsigned long long*>(P.buf[1] + cap_doubles);
    P.status = reinterpret_cast<uint32_t*>(P.flag + 8);
    hipIpcMemHandle_t h;
    LV_HIP(hipIpcGetMemHandle(&h, P.local_alloc));
    static_assert(sizeof(h) == 64, "HIP IPC handles are 64 bytes");
    std::memcpy(handle64, &h, sizeof(h));
    return LV_OK;
}

int peer_init(PeerSet& P, int rank, int world, const void* handles) {
    if (!P.local_alloc) { set_error("lv_comm_peer_export first"); return LV_ESTATE; }
    if (world < 1 || world > LV_PEER_MAX || rank < 0 || rank >= world) { set_error("peer exchange: rank %d of %d (at most %d ranks)", rank, world, LV_PEER_MAX); return LV_EINVAL; }
    P.rank = rank;
    P.world = world;
    for (int r = 0; r < world; ++r) {
        void* base = P.local_alloc;
        if (r != rank) {
            hipIpcMemHandle_t h;
            std::memcpy(&h, static_cast<const char*>(handles) + (size_t)r * sizeof(h), sizeof(h));
            base = nullptr;
            LV_HIP(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
            P.mapped[r] = base;
        }
        double* b0 = static_cast<double*>(base);
        P.peer_buf[0][r] = b0;
        P.peer_buf[1][r] = b0 + P.cap;
        P.peer_flag[r] = reinterpret_cast<unsigned long long*>(b0 + 2 * P.cap);
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
    a.status = P.status;
    hipLaunchKernelGGL(peer_gather_kernel, dim3(P.world), dim3(256), 0, stream, a);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

// 0: every pull so far found its peers; 1: a wait timed out (cleared by the read)
int peer_status(PeerSet& P, hipStream_t stream, int* timed_out) {
    uint32_t st = 0;
    LV_HIP(hipMemcpyAsync(&st, P.status, sizeof(st), hipMemcpyDeviceToHost, stream));
    LV_HIP(hipStreamSynchronize(stream));
    if (st) LV_HIP(hipMemsetAsync(P.status, 0, sizeof(st), stream));
    *timed_out = st ? 1 : 0;
    return LV_OK;
}

void peer_close(PeerSet& P) {
    for (int r = 0; r < LV_PEER_MAX; ++r)
        if (P.mapped[r]) { hipIpcCloseMemHandle(P.mapped[r]); P.mapped[r] = nullptr; }
    if (P.local_alloc) hipFree(P.local_alloc);
    P = PeerSet();
}

}  // namespace lv
