// ref_main_feed.cpp — feeds the REFERENCE'S OWN src/main.cpp, compiled in place against the shim (its `main` renamed lvmain_main by
// -Dmain=lvmain_main; host/Makefile `refmain`), the recorded stream that host/stream_demo.cpp and oracle/ref_build/
// ref_stream_main.cpp read: same input file, same output file.  It stands where the ROS master stood: fill_config's parameter
// values (the ones stream_demo.cpp sets), the two subscribed callbacks, one IMU sample — preceded by the LiDAR sweeps that have
// arrived by then — per ros::spinOnce().  TEST INFRASTRUCTURE (tests/test_gpu_ref.py: this program's trajectory must equal
// stream_demo's bit for bit — the loop a maintainer keeps IS the reference's file).
#include <cstdio>
#include <cstdlib>
#include <fstream>
#include <iostream>

#include "Headers/Common.hpp"

int lvmain_main(int argc, char** argv);

template <typename T>
static T rd(std::ifstream& f) {
    T v;
    f.read(reinterpret_cast<char*>(&v), sizeof(T));
    return v;
}

int main(int argc, char** argv) {
    if (argc != 3) { std::cerr << "usage: ref_main_over_shim in.bin out.bin\n"; return 2; }
    try {
        std::ifstream f(argv[1], std::ios::binary);
        if (!f || rd<uint32_t>(f) != 0x5453564Cu) throw std::runtime_error("bad input file");
        rd<uint32_t>(f);   // on_device: the reference's loop hands scans over by value
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
        // ---- what the parameter server would answer: stream_demo.cpp's settings, everything else fill_config's own defaults ...
        lvmain::Overrides& ov = lvmain::overrides();
        ov.num = {{"mapping_online", getenv("LV_DEMO_MAPPING_OFFLINE") ? 0 : 1}, {"real_time", 0}, {"real_time_delay", 0.1}, {"imu_rate", 100},
                  {"empty_lidar_time", 1.0},
                  // ... except where the shim's defaults (= config/params.yaml, what stream_demo runs with) differ from fill_config's
                  {"downsample_prec", 0.5}, {"PLANES_THRESHOLD", 0.05}, {"min_dist", 4.0}};
        ov.str = {{"LiDAR_type", "hesai"}, {"points_topic", "/points"}, {"imus_topic", "/imu"}};
        ov.vec = {{"/Initialization/deltas", {delta}}, {"initial_gravity", {0.0, 0.0, -9.809}}, {"I_Rotation_L", {1, 0, 0, 0, 1, 0, 0, 0, 1}},
                  {"I_Translation_L", {0, 0, 0}}};
        // ---- a prior map (the reference starts empty; either is Mapper::add — stream_demo does the same)
        {
            Points map_pts;
            for (uint32_t i = 0; i < n_map; ++i) map_pts.push_back(Point(mapv[3 * i], mapv[3 * i + 1], mapv[3 * i + 2], 0.0));
            Mapper::getInstance().add(map_pts, 0.0, false);
        }
        lv_cloud_format fmt;
        lv_cloud_format_preset(LV_LIDAR_HESAI, &fmt);
        Accumulator& accum = Accumulator::getInstance();
        Localizator& loc = Localizator::getInstance();
        std::vector<double> out_t;
        std::vector<state_ikfom> out_x;
        std::vector<uint32_t> out_n;
        size_t ii = 0, mi = 0;
        int tail = 4;
        bool positioned = false;
        lvmain::Feed& feed = lvmain::feed();
        feed.ok = [&] { return ii < imus.size() || tail-- > 0; };
        double last_t = -1e300;
        auto collect = [&] {   // every accum.add(Xt2, t2) of main.cpp:85 leaves a state at the front of BUFFER_X (newest first)
            if (!positioned) return;
            const auto& bx = accum.BUFFER_X.content;
            for (auto it = bx.rbegin(); it != bx.rend(); ++it) {
                if (!(it->time > last_t)) continue;
                last_t = it->time;
                out_t.push_back(it->time);
                out_x.push_back(it->x);
                out_n.push_back((uint32_t)loc.points2match.size());
            }
        };
        feed.spin = [&] {
            collect();
            if (ii >= imus.size()) return;
            const ImuRec& r = imus[ii++];
            while (mi < msgs.size() && msgs[mi].arrival <= r.t) {
                const std::tuple<const void*, size_t, lv_cloud_format, uint64_t> args((const void*)msgs[mi].data.data(), (size_t)msgs[mi].n, fmt, msgs[mi].stamp);
                feed.subscribers.at("/points")(&args);
                ++mi;
            }
            IMU imu(r.a, r.w, r.t);
            std::memcpy(imu.q, r.q, sizeof(imu.q));
            const std::tuple<IMU> iarg(imu);
            feed.subscribers.at("/imu")(&iarg);
            if (accum.ready() && !positioned) {   // the start pose: as stream_demo.cpp places it (the stream does not start at the origin at rest)
                state_ikfom x0 = loc.get_x();
                std::ifstream pf(std::string(argv[1]) + ".x0", std::ios::binary);
                if (pf) pf.read(reinterpret_cast<char*>(&x0), sizeof(x0));
                loc.change_x(x0);
                loc.last_time_integrated = accum.initial_time;
                loc.last_time_updated = accum.initial_time;
                accum.add(loc.latest_state(), accum.initial_time);
                last_t = accum.initial_time;
                positioned = true;
            }
        };
        char* av[] = {argv[0], nullptr};
        lvmain_main(1, av);
        collect();
        std::ofstream o(argv[2], std::ios::binary);
        const uint32_t n = (uint32_t)out_t.size();
        o.write(reinterpret_cast<const char*>(&n), 4);
        for (uint32_t i = 0; i < n; ++i) {
            o.write(reinterpret_cast<const char*>(&out_t[i]), 8);
            o.write(reinterpret_cast<const char*>(&out_x[i]), sizeof(state_ikfom));
            o.write(reinterpret_cast<const char*>(&out_n[i]), 4);
        }
        printf("{\"updates\": %u, \"map_points\": %zu, \"program\": \"the reference's src/main.cpp over the shim\"}\n", n, (size_t)Mapper::getInstance().size());
        HipRuntime::shutdown();
        return 0;
    } catch (const std::exception& e) {
        std::cerr << "ref_main_over_shim: " << e.what() << "\n";
        return 1;
    }
}
