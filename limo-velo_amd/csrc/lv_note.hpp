// lv_note.hpp — a kernel's small result record for the host WITHOUT a copy + stream synchronise.
//
// The 100 Hz cycle (reference src/main.cpp:52-128) asks the device for a handful of words several times per update: the index
// range of the LiDAR window, the number of points the voxel grid left, the counters of the map insert.  Each of them used to be
// hipMemcpyAsync + hipStreamSynchronize (or an event wait), whose wake-up costs ~30 us with the device idle meanwhile: three to
// six times per update, a fifth of the cycle.  A note is a few 64-bit words in pinned, host-mapped memory; the kernel stores
// (sequence number << 32 | value) with ONE system-scope store per word (no tearing, no checksum needed), the host polls the
// words until they carry the sequence number it handed to the launch, for a bounded time, and falls back to
// hipStreamSynchronize (after which the words must be there).
#pragma once
#include <hip/hip_runtime.h>

#include <chrono>
#include <cstdint>
#include <cstring>

namespace lv {

constexpr int NOTE_WORDS = 16;

struct NoteBoard {
    unsigned long long* h = nullptr;   // pinned, host-mapped
    unsigned long long* d = nullptr;   // its device address
    uint32_t seq = 0;                  // last sequence number handed out (0 = never: the words start as 0)
    uint32_t next() { if (++seq == 0) ++seq; return seq; }
};

__device__ __forceinline__ void note_post(unsigned long long* w, uint32_t seq, uint32_t value) {
    __hip_atomic_store(w, ((unsigned long long)seq << 32) | (unsigned long long)value, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

inline hipError_t note_alloc(NoteBoard& b) {
    if (b.h) return hipSuccess;
    hipError_t e = hipHostMalloc((void**)&b.h, NOTE_WORDS * sizeof(unsigned long long), hipHostMallocMapped);
    if (e != hipSuccess) return e;
    std::memset(b.h, 0, NOTE_WORDS * sizeof(unsigned long long));
    return hipHostGetDevicePointer((void**)&b.d, b.h, 0);
}
inline void note_free(NoteBoard& b) {
    if (b.h) hipHostFree(b.h);
    b = NoteBoard();
}
// true: words [first, first + k) carry `seq` (their values in vals).  Polls for up to `budget_ms`, then synchronises the stream
// once and looks again; false only if the kernel never posted (an error on the stream).
inline bool note_wait(const NoteBoard& b, int first, int k, uint32_t seq, uint32_t* vals, hipStream_t stream, int budget_ms = 20) {
    volatile const unsigned long long* w = b.h + first;
    const auto t0 = std::chrono::steady_clock::now();
    bool synced = false;
    for (int it = 0;; ++it) {
        int ok = 0;
        for (int i = 0; i < k; ++i) {
            const unsigned long long v = w[i];
            if ((uint32_t)(v >> 32) != seq) break;
            vals[i] = (uint32_t)v;
            ++ok;
        }
        if (ok == k) return true;
        if (synced) return false;
        if ((it & 255) == 255 && std::chrono::steady_clock::now() - t0 > std::chrono::milliseconds(budget_ms)) {
            hipStreamSynchronize(stream);
            synced = true;
        }
    }
}
// non-blocking: have the words arrived?
inline bool note_ready(const NoteBoard& b, int first, int k, uint32_t seq, uint32_t* vals) {
    volatile const unsigned long long* w = b.h + first;
    for (int i = 0; i < k; ++i) {
        const unsigned long long v = w[i];
        if ((uint32_t)(v >> 32) != seq) return false;
        vals[i] = (uint32_t)v;
    }
    return true;
}

}  // namespace lv
