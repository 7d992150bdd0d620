// main_loop.hpp — the body of the reference's processing loop (src/main.cpp:52-128) written against the shim, to show
// (and compile-check) that the lines a maintainer keeps are the reference's own: only the ROS publishers are gone.
//   ref:  while (accum.ready()) { ... loc.propagate_to(t2); comp.compensate(t1, t2); comp.downsample(...);
//          loc.correct(ds_compensated, t2); State Xt2 = loc.latest_state(); accum.add(Xt2, t2);
//          Points global_ds_compensated = Xt2 * Xt2.I_Rt_L() * ds_compensated; map.add(global_ds_compensated, t2, true);
//          (or, mapping offline, :105-116: if (map.hasToMap(t2)) { comp.compensate(t2 - full_rotation_time, t2) -> world ->
//          comp.downsample -> map.add })   accum.clear_lidar(t2 - Config.empty_lidar_time); break; }
// run_cycle() is one turn of that inner loop; `on_device` selects the three calls that keep the scan on the GPU
// between the stages (same results, no host round trips) instead of the reference's by-value hand-overs.
#pragma once

#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <string>

#include "limovelo_shim.hpp"

struct LoopClock {      // the time variables of main.cpp:44-49
    double t1 = 0, t2 = 1e300, delta = 0;
};

// optional host wall-clock per stage of run_cycle (stream_demo: LV_DEMO_TIMING=1); no effect on the loop itself
struct LoopTimes {
    bool on = false;
    double propagate = 0, window = 0, correct = 0, map_add = 0, clear = 0;
    unsigned long cycles = 0;
    static double now() { return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
};
inline LoopTimes& loop_times() { static LoopTimes t; return t; }

// returns true if a localisation happened in this turn
inline bool run_cycle(Accumulator& accum, Compensator& comp, Localizator& loc, Mapper& map, LoopClock& clk, bool on_device,
                      State* Xt2_out = nullptr, size_t* n_points = nullptr) {
    if (!accum.ready()) return false;
    // Step 0. TIME MANAGEMENT (main.cpp:58-73)
    if (Config.real_time) clk.t2 = accum.latest_time();
    else clk.t2 = std::min(clk.t2 + clk.delta, accum.latest_time());
    clk.delta = accum.update_delta(Config.Initialization, clk.t2);
    clk.t1 = std::max(clk.t2 - clk.delta, loc.last_time_updated);
    if (clk.t2 - clk.t1 < clk.delta - 1e-6) return false;
    // Step 1. LOCALIZATION (:75-93)
    LoopTimes& lt = loop_times();
    double tq = lt.on ? LoopTimes::now() : 0.0;
    auto lap = [&](double& acc) { if (lt.on) { const double t = LoopTimes::now(); acc += t - tq; tq = t; } };
    loc.propagate_to(clk.t2);
    lap(lt.propagate);
    State Xt2;
    if (!on_device) {
        Points compensated = comp.compensate(clk.t1, clk.t2);
        Points ds_compensated = comp.downsample(compensated);
        if ((int)ds_compensated.size() < Config.MAX_POINTS2MATCH) return false;
        if (const char* pre = getenv("LV_DEMO_DUMP_PREFIX")) {   // (diagnostic: the scans of the first updates, tests/test_gpu_ref.py)
            static int k = 0;
            if (k < 6) {
                FILE* fd = fopen((std::string(pre) + "_" + std::to_string(k++) + ".bin").c_str(), "wb");
                if (fd) { for (const Point& q : ds_compensated) { const float v[3] = {q.x, q.y, q.z}; fwrite(v, 4, 3, fd); } fclose(fd); }
            }
        }
        loc.correct(ds_compensated, clk.t2);
        Xt2 = loc.latest_state();
        accum.add(Xt2, clk.t2);
        Points global_ds_compensated = Xt2 * Xt2.I_Rt_L() * ds_compensated;
        // Step 2. MAPPING (:98-103)
        if (Config.mapping_online) map.add(global_ds_compensated, clk.t2, true);
        if (n_points) *n_points = ds_compensated.size();
    } else {
        const size_t n_ds = comp.compensate_downsample_on_device(clk.t1, clk.t2);
        lap(lt.window);
        if ((int)n_ds < Config.MAX_POINTS2MATCH) return false;
        loc.correct_current_scan(clk.t2);
        Xt2 = loc.latest_state();
        accum.add(Xt2, clk.t2);
        lap(lt.correct);
        if (Config.mapping_online) map.add_current_scan(clk.t2, true);
        lap(lt.map_add);
        if (n_points) *n_points = n_ds;
    }
    // Step 2, mapping offline (:105-116): once per full rotation the whole sweep [t2 - FULL_ROTATION_TIME, t2] is de-skewed
    // to t2, taken to the world frame, down-sampled THERE (the voxel grid of the reference acts on the global points) and
    // added; the reference's own by-value hand-overs in both modes (it runs every tenth cycle at delta = 0.01)
    if (!Config.mapping_online && map.hasToMap(clk.t2)) {
        State X = loc.latest_state();
        Points full_compensated = comp.compensate(clk.t2 - Config.full_rotation_time, clk.t2);
        Points global_full_compensated = X * X.I_Rt_L() * full_compensated;
        Points global_full_ds_compensated = comp.downsample(global_full_compensated);
        map.add(global_full_ds_compensated, clk.t2, true);
    }
    // Step 3. ERASE OLD DATA (:116-118)
    accum.clear_lidar(clk.t2 - Config.empty_lidar_time);
    lap(lt.clear);
    ++lt.cycles;
    if (Xt2_out) *Xt2_out = Xt2;
    return true;
}
