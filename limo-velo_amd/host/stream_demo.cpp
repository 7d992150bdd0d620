// stream_demo.cpp — drives main_loop.hpp (the reference's loop, src/main.cpp:52-128) through the C++ shim on a
// recorded stream, so that tests/test_gpu_shim.py can check the trajectory against the ground truth and against the
// Python-driven pipeline.  Input: a binary file written by the test (little endian):
//   u32 magic 'LVST', u32 on_device, f64 delta, u32 n_map, f32 map[n_map*3],
//   u32 n_imu, { f64 t, f32 a[3], f32 w[3], f32 q[4] } * n_imu,
//   u32 n_msgs, { f64 arrival, u64 stamp_usec, u32 n_points, u8 payload[n_points * 48] } * n_msgs   (hesai layout)
// IMU samples and LiDAR messages are fed in time order; after every IMU sample the loop body runs as often as it can.
// Output: u32 n_updates, { f64 t2, f64 x[26], u32 n_points } * n_updates.
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "main_loop.hpp"

template <typename T>
static T rd(std::ifstream& f) {
    T v;
    f.read(reinterpret_cast<char*>(&v), sizeof(T));
    return v;
}

int main(int argc, char** argv) {
    if (argc != 3) { std::cerr << "usage: stream_demo in.bin out.bin\n"; return 2; }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        if (!f || rd<uint32_t>(f) != 0x5453564Cu) throw std::runtime_error("bad input file");
        const bool on_device = rd<uint32_t>(f) != 0;
        const double delta = rd<double>(f);
        const uint32_t n_map = rd<uint32_t>(f);
        std::vector<float> mapv((size_t)n_map * 3);
        f.read(reinterpret_cast<char*>(mapv.data()), (std::streamsize)(mapv.size() * 4));
        struct ImuRec { double t; float a[3], w[3], q[4]; };
        std::vector<ImuRec> imus(rd<uint32_t>(f));
        for (auto& r : imus) { r.t = rd<double>(f); f.read(reinterpret_cast<char*>(r.a), 12); f.read(reinterpret_cast<char*>(r.w), 12); f.read(reinterpret_cast<char*>(r.q), 16); }
        struct Msg { double arrival; uint64_t stamp; uint32_t n; std::vector<unsigned char> data; };
        std::vector<Msg> msgs(rd<uint32_t>(f));
        for (auto& m : msgs) {
            m.arrival = rd<double>(f); m.stamp = rd<uint64_t>(f); m.n = rd<uint32_t>(f);
            m.data.resize((size_t)m.n * 48);
            f.read(reinterpret_cast<char*>(m.data.data()), (std::streamsize)m.data.size());
        }
        Config.Initialization.times = {};
        Config.Initialization.deltas = {delta};
        Config.real_time = false;
        Config.real_time_delay = 0.1;
        Config.imu_rate = 100;
        Config.empty_lidar_time = 1.0;
        Config.mapping_online = getenv("LV_DEMO_MAPPING_OFFLINE") == nullptr;   // (offline: main.cpp:105-116, once per full rotation)
        Config.initial_gravity = {0.f, 0.f, -9.809f};

        Accumulator& accum = Accumulator::getInstance();
        Compensator comp;
        Localizator& loc = Localizator::getInstance();
        Mapper& map = Mapper::getInstance();
        Points map_pts;
        for (uint32_t i = 0; i < n_map; ++i) map_pts.push_back(Point(mapv[3 * i], mapv[3 * i + 1], mapv[3 * i + 2], 0.0));
        map.add(map_pts, 0.0, false);                       // a prior map (the reference starts empty; either is Mapper::add)

        lv_cloud_format fmt;
        lv_cloud_format_preset(LV_LIDAR_HESAI, &fmt);
        LoopClock clk;
        clk.delta = delta;
        std::vector<double> out_t;
        std::vector<state_ikfom> out_x;
        std::vector<uint32_t> out_n;
        size_t mi = 0;
        bool positioned = false;
        {   // a node sizes its LiDAR buffers when it starts, not when the first sweep arrives
            size_t max_n = 0;
            for (const auto& m : msgs) max_n = std::max(max_n, (size_t)m.n);
            if (max_n && !getenv("LV_DEMO_NO_RESERVE")) {
                lv_cloud_reserve(HipRuntime::ctx(), max_n, fmt.point_step, 4 * max_n);
                // a window of delta seconds holds delta / FULL_ROTATION_TIME of a sweep (twice that for slack); scans after the voxel grid are smaller
                const size_t win = (size_t)(2.0 * (double)max_n * std::max(delta, 0.01) / std::max(Config.full_rotation_time, 0.01)) + 4096;
                lv_reserve_stream(HipRuntime::ctx(), on_device ? win : 0, 8192);
                // ... and the second store of the background map rebuild (LV_DEMO_NO_REBUILD_RESERVE: left to the first rebuild's worker)
                if (!getenv("LV_DEMO_NO_REBUILD_RESERVE") && lv_map_reserve_rebuild(HipRuntime::ctx()))
                    throw std::runtime_error(std::string("lv_map_reserve_rebuild: ") + lv_last_error());
            }
        }
        // LV_DEMO_FORCE_REBUILD=K[:sync] — re-linearise the map after the K-th update (in the background; ":sync": stop-the-world)
        std::vector<double> cycle_s;
        int force_after = 0;
        bool force_sync = false;
        double forced_call_s = 0.0;
        if (const char* e = getenv("LV_DEMO_FORCE_REBUILD")) { force_after = atoi(e); force_sync = std::string(e).find(":sync") != std::string::npos; }
        int force_after2 = 0;   // LV_DEMO_FORCE_REBUILD2=K2: adopt the first rebuild (blocking, untimed), then force another one after update K2
        if (const char* e = getenv("LV_DEMO_FORCE_REBUILD2")) force_after2 = atoi(e);
        loop_times().on = getenv("LV_DEMO_TIMING") != nullptr;
        FILE* passes_dump = getenv("LV_DEMO_PASSES_DUMP") ? fopen(getenv("LV_DEMO_PASSES_DUMP"), "w") : nullptr;   // one line per update: its measurement passes
        FILE* cycle_dump = getenv("LV_DEMO_CYCLE_DUMP") ? fopen(getenv("LV_DEMO_CYCLE_DUMP"), "w") : nullptr;   // "cycle ms rebuild-state adopted journal" per line
        double t_ingest = 0.0, t_imu = 0.0;
        constexpr size_t STEADY_AFTER = 30;   // the first three sweeps' worth of updates: first-touch allocations, buffers growing to size
        auto wall_steady0 = std::chrono::steady_clock::now();
        const auto wall0 = std::chrono::steady_clock::now();   // the whole replay: message ingest, IMU handling, every cycle
        for (const ImuRec& r : imus) {
            const double ti0 = loop_times().on ? LoopTimes::now() : 0.0;
            while (mi < msgs.size() && msgs[mi].arrival <= r.t) {
                accum.receive_lidar(msgs[mi].data.data(), msgs[mi].n, fmt, msgs[mi].stamp);
                ++mi;
            }
            const double ti1 = loop_times().on ? LoopTimes::now() : 0.0;
            IMU imu(r.a, r.w, r.t);
            std::memcpy(imu.q, r.q, sizeof(imu.q));
            accum.receive_imu(imu);
            if (loop_times().on) { t_ingest += ti1 - ti0; t_imu += LoopTimes::now() - ti1; }
            if (accum.ready() && !positioned) {
                // the test's trajectory does not start at the origin: place the filter (the reference starts at pos = 0
                // in its own map frame; with a prior map the start pose has to be given)
                state_ikfom x0 = loc.get_x();
                std::ifstream pf(std::string(argv[1]) + ".x0", std::ios::binary);
                if (pf) pf.read(reinterpret_cast<char*>(&x0), sizeof(x0));
                loc.change_x(x0);
                // the given state IS the state at the initial time: without this the first propagate_to would integrate
                // every buffered IMU sample of [-1, t] (Localizator.cpp:61-62) — harmless for a sensor at rest, as the
                // reference assumes at start-up, a 0.2 m kick for one that is already moving at 9 m/s
                loc.last_time_integrated = accum.initial_time;
                // ... and the first state of the path: the reference's start-up stamps its latest state with t1 when the
                // state buffer is empty (Accumulator.cpp:77-83), which is the same thing for a sensor at rest only
                loc.last_time_updated = accum.initial_time;
                accum.add(loc.latest_state(), accum.initial_time);
                clk.t2 = accum.initial_time;               // main.cpp:45 starts from DBL_MAX and lets min() pick latest_time()
                positioned = true;
            }
            for (int guard = 0; guard < 64; ++guard) {
                State Xt2;
                size_t np = 0;
                const auto cyc0 = std::chrono::steady_clock::now();
                if (!run_cycle(accum, comp, loc, map, clk, on_device, &Xt2, &np)) break;
                cycle_s.push_back(std::chrono::duration<double>(std::chrono::steady_clock::now() - cyc0).count());
                if (cycle_dump) {   // (diagnostic: every cycle's time beside the background rebuild's state, read outside the timed part)
                    uint64_t rs[4] = {0, 0, 0, 0};
                    lv_map_rebuild_status(HipRuntime::ctx(), 0, rs);
                    fprintf(cycle_dump, "%zu %.4f %llu %llu %llu\n", cycle_s.size(), 1e3 * cycle_s.back(), (unsigned long long)rs[0], (unsigned long long)rs[2],
                            (unsigned long long)rs[3]);
                }
                if (force_after2 > 0 && out_t.size() + 1 == (size_t)force_after2) {
                    // a SECOND forced background rebuild: the first one is adopted first (a blocking wait, outside the cycle
                    // timing) so that this one runs into a store whose buffers are all allocated — the steady state of a node
                    uint64_t rbw[4];
                    lv_map_rebuild_status(HipRuntime::ctx(), 1, rbw);
                    const auto r0 = std::chrono::steady_clock::now();
                    if (lv_map_relinearise_async(HipRuntime::ctx())) throw std::runtime_error(std::string("forced rebuild 2: ") + lv_last_error());
                    cycle_s.back() += std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
                }
                if (force_after > 0 && out_t.size() + 1 == (size_t)force_after) {
                    // a forced re-linearisation of the (10 M-point) map in the middle of the stream: stop-the-world
                    // (lv_map_relinearise: the caller waits for compaction + rebuild) or in the background
                    // (lv_map_relinearise_async: a worker thread rebuilds a copy, the cycles go on)
                    const auto r0 = std::chrono::steady_clock::now();
                    const int rcr = force_sync ? lv_map_relinearise(HipRuntime::ctx()) : lv_map_relinearise_async(HipRuntime::ctx());
                    if (rcr) throw std::runtime_error(std::string("forced rebuild: ") + lv_last_error());
                    forced_call_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - r0).count();
                    cycle_s.back() += forced_call_s;   // (the call is part of the cycle that made it)
                }
                if (getenv("LV_DEMO_VERBOSE") && out_t.size() < 6)
                    fprintf(stderr, "update %zu: t1 %.4f t2 %.4f points %zu pos %.4f %.4f %.4f vel %.3f %.3f passes %d states %d\n", out_t.size(),
                            clk.t1, clk.t2, np, loc.get_x().pos[0], loc.get_x().pos[1], loc.get_x().pos[2], loc.get_x().vel[0], loc.get_x().vel[1],
                            loc.last_passes, accum.BUFFER_X.size());
                out_t.push_back(clk.t2);
                out_x.push_back(loc.get_x());
                if (passes_dump) fprintf(passes_dump, "%d\n", loc.last_passes);   // (get_x() has pulled the update's results)
                out_n.push_back((uint32_t)np);
                if (out_t.size() == STEADY_AFTER) wall_steady0 = std::chrono::steady_clock::now();
            }
        }
        const double wall_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall0).count();
        const double steady_s = std::chrono::duration<double>(std::chrono::steady_clock::now() - wall_steady0).count();
        std::ofstream o(argv[2], std::ios::binary);
        const uint32_t n = (uint32_t)out_t.size();
        o.write(reinterpret_cast<const char*>(&n), 4);
        for (uint32_t i = 0; i < n; ++i) {
            o.write(reinterpret_cast<const char*>(&out_t[i]), 8);
            o.write(reinterpret_cast<const char*>(&out_x[i]), sizeof(state_ikfom));
            o.write(reinterpret_cast<const char*>(&out_n[i]), 4);
        }
        std::cout << "stream_demo: " << n << " updates, map " << map.size() << " points\n";
        if (loop_times().on && n) {
            const LoopTimes& lt = loop_times();
            fprintf(stderr, "host wall clock per update [us]: propagate %.1f  window %.1f  correct+state %.1f  map add %.1f  clear %.1f  | LiDAR ingest %.1f  "
                    "IMU %.1f  | total %.1f\n", 1e6 * lt.propagate / n, 1e6 * lt.window / n, 1e6 * lt.correct / n, 1e6 * lt.map_add / n, 1e6 * lt.clear / n,
                    1e6 * t_ingest / n, 1e6 * t_imu / n, 1e6 * wall_s / n);
        }
        // one JSON line for scripts/stream_bench_cpp.py: the reference's loop as a C++ host program runs it (no per-stage
        // synchronisation beyond what the calls themselves need)
        double mean_pts = 0;
        for (uint32_t v : out_n) mean_pts += v;
        const double steady = n > STEADY_AFTER ? (double)(n - STEADY_AFTER) / steady_s : 0.0;
        // cycle times (one run_cycle that produced an update) after the warm-up: median / p99 / max — what a forced rebuild does to them
        double c_med = 0, c_p99 = 0, c_max = 0, c_max_after = 0, c_max_after2 = 0, c_p99_after2 = 0, c_med_after2 = 0;
        uint64_t rb[4] = {0, 0, 0, 0};
        if (cycle_s.size() > STEADY_AFTER + 2) {
            std::vector<double> v(cycle_s.begin() + STEADY_AFTER, cycle_s.end());
            std::sort(v.begin(), v.end());
            c_med = v[v.size() / 2]; c_p99 = v[(size_t)((v.size() - 1) * 0.99)]; c_max = v.back();
            if (force_after > 0 && (size_t)force_after <= cycle_s.size()) {
                const size_t end1 = (force_after2 > force_after && (size_t)force_after2 <= cycle_s.size()) ? (size_t)(force_after2 - 1) : cycle_s.size();
                c_max_after = *std::max_element(cycle_s.begin() + (force_after - 1), cycle_s.begin() + end1);
                if (end1 < cycle_s.size()) {
                    std::vector<double> w(cycle_s.begin() + end1, cycle_s.end());
                    std::sort(w.begin(), w.end());
                    c_max_after2 = w.back();
                    c_p99_after2 = w[(size_t)((w.size() - 1) * 0.99)];
                    c_med_after2 = w[w.size() / 2];
                }
            }
        }
        if (force_after > 0) lv_map_rebuild_status(HipRuntime::ctx(), 1, rb);
        printf("{\"updates\": %u, \"wall_s\": %.6f, \"updates_per_s\": %.1f, \"updates_per_s_after_30\": %.1f, \"on_device\": %d, \"scan_points_mean\": %.1f, "
               "\"map_points\": %zu, \"cycle_ms\": {\"median\": %.4f, \"p99\": %.4f, \"max\": %.4f}, \"forced_rebuild\": {\"after_update\": %d, \"sync\": %d, "
               "\"call_ms\": %.3f, \"max_cycle_ms_from_there\": %.4f, \"second_after_update\": %d, \"second_cycle_ms\": {\"median\": %.4f, \"p99\": %.4f, \"max\": %.4f}, "
               "\"rebuilds_started\": %llu, \"rebuilds_adopted\": %llu, \"note\": \"%s\"}}\n",
               n, wall_s, n / wall_s, steady, (int)on_device, n ? mean_pts / n : 0.0, (size_t)map.size(), 1e3 * c_med, 1e3 * c_p99, 1e3 * c_max,
               force_after, (int)force_sync, 1e3 * forced_call_s, 1e3 * c_max_after, force_after2, 1e3 * c_med_after2, 1e3 * c_p99_after2, 1e3 * c_max_after2,
               (unsigned long long)rb[1], (unsigned long long)rb[2],
               force_after2 > 0 ? "wall_s / updates_per_s include the BLOCKING adoption of the first forced rebuild before the second is forced (a wait outside "
                                  "the cycle clock: cycle_ms does not contain it); read the cycle times, not the rate, for what a background rebuild costs"
                                : "");
        if (cycle_dump) fclose(cycle_dump);
        if (passes_dump) fclose(passes_dump);
        HipRuntime::shutdown();
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "stream_demo: " << e.what() << "\n";
        return 1;
    }
}
