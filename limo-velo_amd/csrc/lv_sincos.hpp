// lv_sincos.hpp — sin / cos of an f32 argument through ONE fixed f64 polynomial (Cody-Waite reduction + Taylor), rounded
// to f32.  SO3Math::Exp (reference include/Headers/Utils.hpp:45-46) calls std::sin / std::cos on floats, whose last ulp
// depends on the libm and the platform; the product pins the operation sequence instead, and this header is its single
// definition: the device kernels (lv_device.hpp: de-skew, predict) and the host shim's State::operator+=
// (host/limovelo_shim.cpp) include it.  The CPU checker under oracle/ keeps an independent restatement of the same sequence
// on purpose — the tests compare the two bit for bit.
#pragma once

#include <cmath>

#if defined(__HIPCC__)
#define LV_SINCOS_HD __host__ __device__
#else
#define LV_SINCOS_HD
#endif

namespace lv {

LV_SINCOS_HD inline void sincos_f32(float xf, float& sn, float& cs) {
    const double x = (double)xf;
    const double k = ::rint(x * 0.63661977236758134308);
    double r = x - k * 1.57079632673412561417e+00;
    r = r - k * 6.07710050650619224932e-11;
    r = r - k * 2.02226624879595063154e-21;
    const double z = r * r;
    double ps = 1.58969099521155010221e-10;
    ps = ps * z - 2.50507602534068634195e-08;
    ps = ps * z + 2.75573137070700676789e-06;
    ps = ps * z - 1.98412698298579493134e-04;
    ps = ps * z + 8.33333333332248946124e-03;
    ps = ps * z - 1.66666666666666324348e-01;
    const double s0 = r + r * z * ps;
    double pc = -1.13596475577881948265e-11;
    pc = pc * z + 2.08757232129817482790e-09;
    pc = pc * z - 2.75573143513906633035e-07;
    pc = pc * z + 2.48015872894767294178e-05;
    pc = pc * z - 1.38888888888741095749e-03;
    pc = pc * z + 4.16666666666666019037e-02;
    const double c0 = 1.0 - 0.5 * z + z * z * pc;
    const int q = (int)k & 3;
    const double sd = (q == 0) ? s0 : (q == 1) ? c0 : (q == 2) ? -s0 : -c0;
    const double cd = (q == 0) ? c0 : (q == 1) ? -s0 : (q == 2) ? -c0 : s0;
    sn = (float)sd;
    cs = (float)cd;
}

}  // namespace lv
