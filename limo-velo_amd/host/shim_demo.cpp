// shim_demo.cpp — drives the reference-style C++ API exactly like src/main.cpp:84-102 does
// (map.add -> loc.correct -> loc.latest_state), plus Mapper::match / Localizator::calculate_H.
// Input/Output are raw little-endian files so tests/test_gpu_shim.py can compare with the oracle.
//   shim_demo <map.f32> <scan.f32> <state26.f64> <P529.f64> <out.bin>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <fstream>
#include <iostream>

#include "limovelo_shim.hpp"

template <typename T>
static std::vector<T> slurp(const char* path) {
    std::ifstream f(path, std::ios::binary);
    if (!f) throw std::runtime_error(std::string("cannot open ") + path);
    f.seekg(0, std::ios::end);
    size_t n = (size_t)f.tellg() / sizeof(T);
    f.seekg(0);
    std::vector<T> v(n);
    f.read(reinterpret_cast<char*>(v.data()), (std::streamsize)(n * sizeof(T)));
    return v;
}

int main(int argc, char** argv) {
    if (argc != 6) { std::cerr << "usage: shim_demo map.f32 scan.f32 state.f64 P.f64 out.bin\n"; return 2; }
    try {
        auto mapv = slurp<float>(argv[1]);
        auto scanv = slurp<float>(argv[2]);
        auto xs = slurp<double>(argv[3]);
        auto Pv = slurp<double>(argv[4]);
        Points map_pts, scan_pts;
        for (size_t i = 0; i + 2 < mapv.size(); i += 3) map_pts.push_back(Point(mapv[i], mapv[i + 1], mapv[i + 2], 0.0));
        for (size_t i = 0; i + 2 < scanv.size(); i += 3) scan_pts.push_back(Point(scanv[i], scanv[i + 1], scanv[i + 2], 0.1));
        state_ikfom x0;
        std::memcpy(&x0, xs.data(), sizeof(x0));

        Mapper& map = Mapper::getInstance();
        Localizator& loc = Localizator::getInstance();
        loc.correct(scan_pts, 0.05);  // no map yet: must be a silent no-op (Localizator.cpp:24)
        if (loc.last_time_updated >= 0) { std::cerr << "correct() without a map must not update\n"; return 1; }
        map.add(map_pts, 0.0, false);
        loc.init_state(x0);
        loc.change_P(Pv.data());

        Matches matches = map.match(State(loc.get_x(), 0.1), scan_pts);
        MatrixXd H;
        VectorXd h;
        loc.calculate_H(loc.get_x(), matches, H, h);

        // two zero-length IMU steps exercise propagate_to (Localizator.cpp:59-75) without moving the state
        IMUs imus;
        const float a0[3] = {0.f, 0.f, 9.809f}, w0[3] = {0.f, 0.f, 0.f};
        loc.last_time_integrated = 0.1;
        imus.push_back(IMU(a0, w0, 0.1));
        loc.propagate_to(imus, 0.1);
        loc.correct(scan_pts, 0.1);
        State Xt2 = loc.latest_state();

        std::ofstream out(argv[5], std::ios::binary);
        double header[4] = {(double)map.size(), (double)matches.size(), (double)loc.last_passes, Xt2.time};
        out.write(reinterpret_cast<const char*>(header), sizeof(header));
        out.write(reinterpret_cast<const char*>(&loc.get_x()), sizeof(state_ikfom));
        out.write(reinterpret_cast<const char*>(loc.get_P()), sizeof(double) * 529);
        out.write(reinterpret_cast<const char*>(H.d.data()), (std::streamsize)(sizeof(double) * H.d.size()));
        out.write(reinterpret_cast<const char*>(h.d.data()), (std::streamsize)(sizeof(double) * h.d.size()));
        for (auto& m : matches) {
            float rec[8] = {m.point.x, m.point.y, m.point.z, m.plane.n.A, m.plane.n.B, m.plane.n.C, m.plane.n.D, m.distance};
            out.write(reinterpret_cast<const char*>(rec), sizeof(rec));
        }
        std::printf("shim_demo: map %d pts, %zu matches, %d passes, pos %.6f %.6f %.6f\n", map.size(), matches.size(),
                    loc.last_passes, loc.get_x().pos[0], loc.get_x().pos[1], loc.get_x().pos[2]);
        // ---- LiDAR wire format -> device buffer -> windowed de-skew (row f-4), self-checked ------------------------
        {
            struct VelodynePoint { float x, y, z, pad, intensity, time; uint16_t ring; uint8_t fill[6]; };   // velodyne_ros::Point
            static_assert(sizeof(VelodynePoint) == 32, "velodyne_ros::Point layout");
            const size_t n = 20000;
            std::vector<VelodynePoint> msg(n);
            for (size_t i = 0; i < n; ++i) {
                const float az = 6.2831853f * (float)i / (float)n;
                const float r = 5.f + (float)(i % 37);
                msg[i] = VelodynePoint{r * std::cos(az), r * std::sin(az), 0.1f * (float)(i % 11), 0.f, (float)(i % 255),
                                       -0.1f + 0.1f * (float)i / (float)n, (uint16_t)(i % 16), {0, 0, 0, 0, 0, 0}};
            }
            lv_cloud_format fmt;
            if (lv_cloud_format_preset(LV_LIDAR_VELODYNE, &fmt) != LV_OK) throw std::runtime_error("preset");
            LidarBuffer& buf = LidarBuffer::getInstance();
            const size_t kept = buf.process(msg.data(), n, fmt, 1000000000ull /* 1000 s */);
            Points all = buf.get_points(-1e300, 1e300);
            if (kept == 0 || all.size() != kept || buf.size() != kept) throw std::runtime_error("LidarBuffer: count mismatch");
            for (size_t i = 1; i < all.size(); ++i)
                if (all[i].time < all[i - 1].time) throw std::runtime_error("LidarBuffer: not time ordered");
            const double t1 = all[all.size() / 4].time, t2 = all[3 * all.size() / 4].time;
            Points win = buf.get_points(t1, t2);
            States path;
            State s0;
            s0.time = t1 - 0.01;
            s0.vel[0] = 3.f; s0.w[2] = 0.3f; s0.a[2] = 9.807f;
            path.push_back(s0);
            for (int k = 1; k <= 12; ++k) {
                State sk = path.back();
                float a[3] = {0.1f, 0.f, 9.807f}, w[3] = {0.f, 0.f, 0.3f};
                sk += IMU(a, w, s0.time + 0.01 * k);
                path.push_back(sk);
            }
            Compensator comp;
            Points a = comp.compensate(path, path.back(), win, 0.5f);
            Points b = comp.compensate(path, path.back(), t1, t2, 0.5f);
            if (a.size() != b.size() || a.empty()) throw std::runtime_error("windowed de-skew: size mismatch");
            for (size_t i = 0; i < a.size(); ++i)
                if (std::memcmp(&a[i].x, &b[i].x, 12) != 0) throw std::runtime_error("windowed de-skew differs from de-skew of the fetched points");
            buf.clear_lidar(t2);
            if (buf.size() != kept - buf.get_points(-1e300, t2).size() && buf.size() >= kept) throw std::runtime_error("clear_lidar");
            std::printf("shim_demo: lidar buffer %zu of %zu points kept, window %zu -> %zu de-skewed points\n", kept, n, win.size(), a.size());
        }
        HipRuntime::shutdown();
    } catch (const std::exception& e) {
        std::cerr << "shim_demo failed: " << e.what() << "\n";
        return 1;
    }
    return 0;
}
