// lv_cloud.hip — row f-4 of SURVEY §8: LiDAR wire formats.  One sensor_msgs/PointCloud2 message of any of the
// reference's drivers -> the reference's time-stamped `Point`s, on the device, into a device-resident LiDAR
// buffer from which de-skew windows are cut without a host round trip.
//
//   PointCloudProcessor::msg2points   src/Utils/PointCloudProcessor.cpp:24-34,37-97   (velodyne / hesai / ouster / custom)
//   Point::Point(sensor point[, offset]) src/Objects/Point.cpp:37-111  (time rules), :152-178 (xyz, intensity, range)
//   PointCloudProcessor::downsample   :18-21,99-110   (every downsample_rate-th point, min_dist < |p|)
//   PointCloudProcessor::sort_points  :112-121        (by time; std::sort there — stable here, see below)
//   Accumulator::process / push / get_points / clear_lidar   src/Modules/Accumulator.cpp:64-70,93-95,141-153,
//                                     include/Headers/Accumulator.hpp:62-74, src/Objects/Buffer.cpp:57-62
//   point layouts                     include/Headers/Common.hpp:109-221 (what pcl::fromROSMsg maps BY FIELD NAME; on
//                                     the wire the offsets come from msg.fields, hence lv_cloud_format)
//
// Order of equal time stamps: the reference sorts with std::sort (unstable: unspecified among equal keys), so any
// fixed rule restates it; here the sort is stable (equal stamps keep message order), as in the oracle.
#include <hipcub/hipcub.hpp>

#include "lv_host.hpp"

#include <chrono>
#include <cstdlib>

namespace lv {

namespace {

template <typename T>
__device__ __forceinline__ T load_unaligned(const unsigned char* p) {   // wire fields need not be aligned
    T v;
    unsigned char* d = reinterpret_cast<unsigned char*>(&v);
#pragma unroll
    for (int i = 0; i < (int)sizeof(T); ++i) d[i] = p[i];
    return v;
}

// Conversions::nanosec2Sec (src/Utils/Utils.cpp:25-30): int seconds + int nanoseconds * 1e-9
__device__ __forceinline__ double nanosec_to_sec(uint32_t t) {
    const int order = 1000000000;
    const int secs = (int)(t / (uint32_t)order);
    const int nsecs = (int)(t % (uint32_t)order);
    return (double)secs + (double)nsecs * 1e-9;
}

__global__ void cloud_decode_kernel(const unsigned char* __restrict__ raw, uint32_t n, CloudFormat fmt, IngestParams prm,
                                    double begin_time, CloudPoint* __restrict__ out, unsigned char* __restrict__ keep) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned char* p = raw + (size_t)i * fmt.point_step;
    CloudPoint o;
    o.x = load_unaligned<float>(p + fmt.off_x);                             // Point::set_XYZ, Point.cpp:152-157
    o.y = load_unaligned<float>(p + fmt.off_y);
    o.z = load_unaligned<float>(p + fmt.off_z);
    o.pad_ = 0.f;
    // |p| as Eigen evaluates Vector3f::norm(): sqrt(x0^2 + (x1^2 + x2^2)), f32
    const float nrm = sqrtf(dot3f(o.x, o.x, o.y, o.y, o.z, o.z));
    switch (fmt.intensity_type) {                                           // Point::set_attributes, Point.cpp:166-178
        case LV_ATTR_F32: o.intensity = load_unaligned<float>(p + fmt.off_intensity); break;
        case LV_ATTR_U8: o.intensity = (float)p[fmt.off_intensity]; break;
        case LV_ATTR_U16: o.intensity = (float)load_unaligned<uint16_t>(p + fmt.off_intensity); break;
        default: o.intensity = 0.f; break;
    }
    o.range = fmt.range_type == LV_ATTR_U32 ? (float)load_unaligned<uint32_t>(p + fmt.off_range) : nrm;
    double t;
    switch (fmt.time_type) {
        case LV_TIME_F32_SEC: t = (double)load_unaligned<float>(p + fmt.off_time); break;      // velodyne `time`
        case LV_TIME_U32_NSEC: t = nanosec_to_sec(load_unaligned<uint32_t>(p + fmt.off_time)); break;  // ouster `t`
        default: t = load_unaligned<double>(p + fmt.off_time); break;                              // hesai / custom `timestamp`
    }
    if (fmt.relative_time && !prm.offset_beginning) t = prm.full_rotation_time + t;          // Point.cpp:57-60,76-79
    o.time = t + begin_time;                                                // Point(p, time_offset): time += offset
    out[i] = o;
    // temporal_downsample (PointCloudProcessor.cpp:99-110): the counter advances on every point
    const bool every = prm.downsample_rate <= 1 || ((i + 1u) % (uint32_t)prm.downsample_rate) == 0u;
    keep[i] = (every && prm.min_dist < nrm) ? 1 : 0;
}

// f64 time stamp -> u64 whose unsigned order is the numeric order
__global__ void cloud_time_keys_kernel(const CloudPoint* __restrict__ pts, uint32_t n, uint64_t* __restrict__ keys, uint32_t* __restrict__ ids) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const uint64_t b = (uint64_t)__double_as_longlong(pts[i].time);
    keys[i] = (b & 0x8000000000000000ull) ? ~b : (b | 0x8000000000000000ull);
    ids[i] = i;
}
__global__ void cloud_gather_kernel(const CloudPoint* __restrict__ in, const uint32_t* __restrict__ ids, uint32_t n, CloudPoint* __restrict__ out) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = in[ids[i]];
}

// Accumulator::get_points(t1, t2): the buffered points with t1 <= time <= t2, oldest first.  The buffer is time
// ordered (messages arrive in order and each is sorted), so the window is one contiguous range [lo, hi).
// first index in [lo, hi) whose time does NOT satisfy `t < bound` (strict) / `t <= bound`: the buffer is time ordered, so
// a wavefront narrows the range 64-fold per round trip (a scalar bisection was 17 dependent loads, 7 us)
__device__ __forceinline__ uint32_t wave_time_bound(const CloudPoint* __restrict__ pts, uint32_t lo, uint32_t hi, double bound, bool inclusive) {
    const uint32_t lane = threadIdx.x & 63u;
    while (lo < hi) {
        const uint32_t span = hi - lo;
        const uint32_t step = (span + 63u) / 64u;
        const uint32_t idx = lo + lane * step;
        bool right = false;
        if (idx < hi) {
            const double t = pts[idx].time;
            right = inclusive ? (t <= bound) : (t < bound);
        }
        const uint32_t c = (uint32_t)__popcll(__ballot(right));   // the predicate is monotone: the first c samples go right
        if (step == 1u) return lo + c;
        if (c == 0u) return lo;
        const uint32_t nlo = lo + (c - 1u) * step + 1u;
        const uint32_t nhi = lo + c * step < hi ? lo + c * step : hi;
        lo = nlo;
        hi = nhi;
    }
    return lo;
}
// (the answers go to the host as notes: lv_note.hpp)
// do_clear: a Buffer::clear(t_clear) that has not been applied yet rides along — wavefront 2 finds the new head (first index
// with time > t_clear), and the window is taken over what is left: lo = max(lo, head).
__global__ void cloud_window_kernel(const CloudPoint* __restrict__ pts, uint32_t head, uint32_t n, double t1, double t2, uint32_t* __restrict__ out,
                                    unsigned long long* __restrict__ note, uint32_t seq, unsigned* __restrict__ reset8, int do_clear, double t_clear) {
    __shared__ uint32_t s_r[3];
    const uint32_t wave = threadIdx.x >> 6;   // wavefront 0: first index with time >= t1; wavefront 1: first with time > t2
    if (reset8 && threadIdx.x < 8) reset8[threadIdx.x] = threadIdx.x < 3 ? 0xFFFFFFFFu : 0u;   // (vg_bounds_init_kernel)
    uint32_t r = head;
    if (wave < 2) r = wave_time_bound(pts, head, n, wave ? t2 : t1, wave != 0);
    else if (wave == 2 && do_clear) r = wave_time_bound(pts, head, n, t_clear, true);
    if ((threadIdx.x & 63u) == 0 && wave < 3) s_r[wave] = r;
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t nh = s_r[2];
        const uint32_t lo = s_r[0] > nh ? s_r[0] : nh, hi = s_r[1] > nh ? s_r[1] : nh;
        out[0] = lo; out[1] = hi; out[3] = nh;
        note_post(note, seq, lo);
        note_post(note + 1, seq, hi);
        note_post(note + 2, seq, nh);
    }
}
// Buffer::clear(t) (Buffer.cpp:57-62): drop from the old end while t >= time
__global__ void cloud_clear_kernel(const CloudPoint* __restrict__ pts, uint32_t head, uint32_t n, double t, uint32_t* __restrict__ out,
                                   unsigned long long* __restrict__ note, uint32_t seq) {
    if (threadIdx.x >= 64) return;
    const uint32_t r = wave_time_bound(pts, head, n, t, true);   // first index with time > t
    if (threadIdx.x == 0) { out[0] = r; note_post(note, seq, r); }
}
__global__ void cloud_post_count_kernel(const uint32_t* __restrict__ count, unsigned long long* __restrict__ note, uint32_t seq) {
    if (threadIdx.x == 0) note_post(note, seq, count[0]);
}
__global__ void cloud_unpack_kernel(const CloudPoint* __restrict__ pts, uint32_t n, float4* __restrict__ xyz, double* __restrict__ times) {
    const uint32_t i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const CloudPoint p = pts[i];
    xyz[i] = make_float4(p.x, p.y, p.z, 0.f);
    times[i] = p.time;
}

}  // namespace

int CloudStore::reserve_msg(size_t n, size_t bytes) {
    if (bytes > raw_cap) {
        size_t cap = raw_cap ? raw_cap : (1u << 20);
        while (cap < bytes) cap *= 2;
        LV_HIP(hipDeviceSynchronize());   // (a staging buffer may still be in flight)
        hipFree(d_rawmsg);
        if (h_rawmsg) hipHostFree(h_rawmsg);
        if (h_rawmsg2) hipHostFree(h_rawmsg2);
        d_rawmsg = nullptr; h_rawmsg = h_rawmsg2 = nullptr; raw_cap = 0;
        LV_HIP(hipMalloc(&d_rawmsg, cap));
        LV_HIP(hipHostMalloc((void**)&h_rawmsg, cap, hipHostMallocDefault));
        LV_HIP(hipHostMalloc((void**)&h_rawmsg2, cap, hipHostMallocDefault));
        for (int b = 0; b < 2; ++b) {
            if (!ev_stage[b]) LV_HIP(hipEventCreateWithFlags(&ev_stage[b], hipEventDisableTiming));
            stage_busy[b] = false;
        }
        raw_cap = cap;
    }
    if (n > msg_cap) {
        size_t cap = msg_cap ? msg_cap : 65536;
        while (cap < n) cap *= 2;
        hipFree(d_decoded); hipFree(d_kept); hipFree(d_keep); hipFree(d_keys); hipFree(d_keys_sorted); hipFree(d_ids);
        hipFree(d_ids_sorted); hipFree(d_tmp);
        d_decoded = d_kept = nullptr; d_keep = nullptr; d_keys = d_keys_sorted = nullptr; d_ids = d_ids_sorted = nullptr; d_tmp = nullptr;
        msg_cap = 0;
        LV_HIP(hipMalloc(&d_decoded, cap * sizeof(CloudPoint)));
        LV_HIP(hipMalloc(&d_kept, cap * sizeof(CloudPoint)));
        LV_HIP(hipMalloc(&d_keep, cap));
        LV_HIP(hipMalloc(&d_keys, cap * sizeof(uint64_t)));
        LV_HIP(hipMalloc(&d_keys_sorted, cap * sizeof(uint64_t)));
        LV_HIP(hipMalloc(&d_ids, cap * sizeof(uint32_t)));
        LV_HIP(hipMalloc(&d_ids_sorted, cap * sizeof(uint32_t)));
        size_t a = 0, b = 0;
        LV_HIP((hipError_t)hipcub::DeviceSelect::Flagged(nullptr, a, d_decoded, d_keep, d_kept, d_count, (int)cap, (hipStream_t)0));
        LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(nullptr, b, d_keys, d_keys_sorted, d_ids, d_ids_sorted, (int)cap, 0, 64,
                                                               (hipStream_t)0));
        tmp_bytes = a > b ? a : b;
        LV_HIP(hipMalloc(&d_tmp, tmp_bytes));
        msg_cap = cap;
    }
    return LV_OK;
}

int CloudStore::reserve_buffer(hipStream_t stream, size_t total) {
    if (total <= buf_cap) return LV_OK;
    // Buffer::clear only advances `head`: before the buffer grows, the living range [head, size) moves to the front when it
    // fits there without overlapping itself (it does as soon as half of what was ever appended has been cleared) — the buffer of
    // a stream then stays at a few sweeps instead of doubling for ever (and reallocating, 0.3 ms, every time)
    const size_t live = size - head;
    if (head >= live && total - head <= buf_cap) {
        if (live) LV_HIP(hipMemcpyAsync(d_buf, d_buf + head, live * sizeof(CloudPoint), hipMemcpyDeviceToDevice, stream));
        size = (uint32_t)live;
        head = 0;
        return LV_OK;
    }
    size_t cap = buf_cap ? buf_cap : (1u << 18);
    while (cap < total) cap *= 2;
    CloudPoint* nb = nullptr;
    LV_HIP(hipMalloc(&nb, cap * sizeof(CloudPoint)));
    if (size > head) LV_HIP(hipMemcpyAsync(nb, d_buf + head, (size_t)(size - head) * sizeof(CloudPoint), hipMemcpyDeviceToDevice, stream));
    LV_HIP(hipStreamSynchronize(stream));
    hipFree(d_buf);
    d_buf = nb;
    size -= head;
    head = 0;
    buf_cap = cap;
    return LV_OK;
}

int CloudStore::init() {
    if (d_count) return LV_OK;
    LV_HIP(hipMalloc(&d_count, 4 * sizeof(uint32_t)));
    LV_HIP(hipHostMalloc((void**)&h_count, 4 * sizeof(uint32_t), hipHostMallocDefault));
    LV_HIP(note_alloc(notes));
    return LV_OK;
}

int CloudStore::ingest(hipStream_t stream, const void* data, size_t n, const CloudFormat& fmt, const IngestParams& prm,
                       double begin_time, size_t* n_kept) {
    if (n_kept) *n_kept = 0;
    if (n == 0) return LV_OK;
    static const bool timing = getenv("LV_INGEST_TIMING") != nullptr;
    auto now = [] { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count() * 1e6; };
    const double tt0 = timing ? now() : 0.0;
    double tt1 = 0, tt2 = 0, tt3 = 0, tt4 = 0;
    int rc = init();
    if (rc) return rc;
    rc = settle();
    if (rc) return rc;
    const size_t bytes = n * (size_t)fmt.point_step;
    rc = reserve_msg(n, bytes);
    if (rc) return rc;
    // The message goes through a pinned staging buffer in four pieces: the host copies piece i + 1 while piece i is on its way
    // to the device (a 4 MB sweep: 0.3 ms of host memcpy + 0.1 ms of DMA back to back before), and the two staging buffers
    // alternate between messages, so that this call does not wait for whatever the stream is still busy with (the previous
    // cycle's map insert, typically) before it may touch the buffer.
    if (timing) tt1 = now();
    const int sb = stage_next;
    stage_next ^= 1;
    if (stage_busy[sb]) { LV_HIP(hipEventSynchronize(ev_stage[sb])); stage_busy[sb] = false; }   // (two messages ago: long done)
    unsigned char* hs = sb ? h_rawmsg2 : h_rawmsg;
    // (the device copy of the previous message is read by kernels of this stream that were enqueued before this copy: in order)
    const size_t piece = ((bytes + 3) / 4 + 255) & ~(size_t)255;
    for (size_t off = 0; off < bytes; off += piece) {
        const size_t len = bytes - off < piece ? bytes - off : piece;
        std::memcpy(hs + off, static_cast<const unsigned char*>(data) + off, len);
        LV_HIP(hipMemcpyAsync(d_rawmsg + off, hs + off, len, hipMemcpyHostToDevice, stream));
    }
    LV_HIP(hipEventRecord(ev_stage[sb], stream));
    stage_busy[sb] = true;
    if (timing) tt2 = now();
    const int B = 256;
    const uint32_t grid = (uint32_t)((n + B - 1) / B);
    hipLaunchKernelGGL(cloud_decode_kernel, dim3(grid), dim3(B), 0, stream, d_rawmsg, (uint32_t)n, fmt, prm, begin_time, d_decoded, d_keep);
    size_t tmp = tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceSelect::Flagged(d_tmp, tmp, d_decoded, d_keep, d_kept, d_count, (int)n, stream));
    const uint32_t seq = notes.next();
    hipLaunchKernelGGL(cloud_post_count_kernel, dim3(1), dim3(64), 0, stream, d_count, notes.d + 4, seq);
    LV_HIP(hipGetLastError());
    uint32_t m = 0;   // the kept count sizes the sort and the append
    if (!note_wait(notes, 4, 1, seq, &m, stream)) { set_error("LiDAR buffer: the ingest did not report"); return LV_EHIP; }
    if (timing) tt3 = now();
    if (n_kept) *n_kept = m;
    if (m == 0) return LV_OK;
    rc = reserve_buffer(stream, (size_t)size + m);
    if (rc) return rc;
    const uint32_t g2 = (m + B - 1) / B;
    hipLaunchKernelGGL(cloud_time_keys_kernel, dim3(g2), dim3(B), 0, stream, d_kept, m, d_keys, d_ids);
    tmp = tmp_bytes;
    LV_HIP((hipError_t)hipcub::DeviceRadixSort::SortPairs(d_tmp, tmp, d_keys, d_keys_sorted, d_ids, d_ids_sorted, (int)m, 0, 64, stream));
    hipLaunchKernelGGL(cloud_gather_kernel, dim3(g2), dim3(B), 0, stream, d_kept, d_ids_sorted, m, d_buf + size);   // Accumulator::push
    LV_HIP(hipGetLastError());
    size += m;
    if (timing) { tt4 = now(); fprintf(stderr, "ingest us: settle+reserve %.0f  staging+H2D %.0f  decode+select+note %.0f  append launches %.0f\n", tt1 - tt0, tt2 - tt1, tt3 - tt2, tt4 - tt3); }
    return LV_OK;
}

// A Buffer::clear(t) is only remembered (clears are monotone: the latest, largest t stands for all of them); it is applied by
// the window kernel that follows it in the 100 Hz cycle, in the same launch, or — if something else needs the buffer's true
// head first (an ingest, a size query) — by a launch of its own here.
int CloudStore::settle() {
    if (!clear_pending) return LV_OK;
    clear_pending = false;
    if (size <= head) { head = size = 0; return LV_OK; }
    const uint32_t seq = notes.next();
    hipLaunchKernelGGL(cloud_clear_kernel, dim3(1), dim3(64), 0, clear_stream, d_buf, head, size, clear_t, d_count + 3, notes.d + 3, seq);
    LV_HIP(hipGetLastError());
    uint32_t v = 0;
    if (!note_wait(notes, 3, 1, seq, &v, clear_stream)) { set_error("LiDAR buffer: the clear kernel did not report"); return LV_EHIP; }
    head = v;
    if (head >= size) head = size = 0;
    return LV_OK;
}

int CloudStore::window(hipStream_t stream, double t1, double t2, uint32_t* lo, uint32_t* hi, unsigned* reset8) {
    *lo = *hi = head;
    if (size <= head) { clear_pending = false; head = size = 0; *lo = *hi = 0; return LV_OK; }
    const uint32_t seq = notes.next();
    const int do_clear = clear_pending ? 1 : 0;
    clear_pending = false;
    hipLaunchKernelGGL(cloud_window_kernel, dim3(1), dim3(192), 0, stream, d_buf, head, size, t1, t2, d_count, notes.d, seq, reset8, do_clear,
                       clear_t);
    LV_HIP(hipGetLastError());
    uint32_t v[3] = {0, 0, 0};
    if (!note_wait(notes, 0, 3, seq, v, stream)) { set_error("LiDAR buffer: the window kernel did not report"); return LV_EHIP; }
    head = v[2];
    if (head >= size) { head = size = 0; *lo = *hi = 0; return LV_OK; }
    *lo = v[0];
    *hi = v[1] > v[0] ? v[1] : v[0];
    return LV_OK;
}

int CloudStore::clear_before(hipStream_t stream, double t) {
    if (size <= head && !clear_pending) return LV_OK;
    clear_t = clear_pending ? (t > clear_t ? t : clear_t) : t;
    clear_stream = stream;
    clear_pending = true;   // (applied by the next window, or by whatever needs the true head first: settle)
    return LV_OK;
}

int CloudStore::unpack(hipStream_t stream, uint32_t lo, uint32_t n, float4* xyz, double* times) {
    if (n == 0) return LV_OK;
    hipLaunchKernelGGL(cloud_unpack_kernel, dim3((n + 255) / 256), dim3(256), 0, stream, d_buf + lo, n, xyz, times);
    LV_HIP(hipGetLastError());
    return LV_OK;
}

void CloudStore::release() {
    hipFree(d_rawmsg); hipFree(d_decoded); hipFree(d_kept); hipFree(d_keep); hipFree(d_keys); hipFree(d_keys_sorted);
    hipFree(d_ids); hipFree(d_ids_sorted); hipFree(d_tmp); hipFree(d_buf); hipFree(d_count);
    if (h_rawmsg) hipHostFree(h_rawmsg);
    if (h_rawmsg2) hipHostFree(h_rawmsg2);
    for (int b = 0; b < 2; ++b) if (ev_stage[b]) hipEventDestroy(ev_stage[b]);
    if (h_count) hipHostFree(h_count);
    note_free(notes);
    *this = CloudStore();
}

}  // namespace lv
