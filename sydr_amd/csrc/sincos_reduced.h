// fp64 sin/cos of a phase of up to ~1e6 rad, shared by the correlators (correlator.h) and the acquisition's carrier
// wipe-off (pcps.hip).
#pragma once

#include <hip/hip_runtime.h>

#include <cmath>

namespace sdr {

constexpr double kHalfPiHi = 1.57079632679489655800e+00;  // fl(pi/2)
constexpr double kHalfPiLo = 6.12323399573676603587e-17;  // pi/2 - fl(pi/2)
constexpr double kTwoOverPi = 6.36619772367581382433e-01;

// fp64 sin/cos of a phase of up to ~1e6 rad: Cody-Waite reduction to |t| <= pi/4 by whole
// quarter turns (the k*hi product is exact inside the FMA), then the classic degree-13 / 14
// minimax kernels (fdlibm k_sin / k_cos coefficients, < 1 ulp on the reduced range).  This
// replaces the general libm sincos, whose Payne-Hanek path is dead weight here.
__host__ __device__ __forceinline__ void sincos_reduced(double ph, double* s, double* c) {
    const double k = rint(ph * kTwoOverPi);
    double t = __builtin_fma(-k, kHalfPiHi, ph);
    t = __builtin_fma(-k, kHalfPiLo, t);
    const double z = t * t;
    double ps = __builtin_fma(z, 1.58969099521155010221e-10, -2.50507602534068634195e-08);
    ps = __builtin_fma(z, ps, 2.75573137070700676789e-06);
    ps = __builtin_fma(z, ps, -1.98412698298579493134e-04);
    ps = __builtin_fma(z, ps, 8.33333333332248946124e-03);
    ps = __builtin_fma(z, ps, -1.66666666666666324348e-01);
    const double sn = __builtin_fma(t * z, ps, t);
    double pc = __builtin_fma(z, -1.13596475577881948265e-11, 2.08757232129817482790e-09);
    pc = __builtin_fma(z, pc, -2.75573143513906633035e-07);
    pc = __builtin_fma(z, pc, 2.48015872894767294178e-05);
    pc = __builtin_fma(z, pc, -1.38888888888741095749e-03);
    pc = __builtin_fma(z, pc, 4.16666666666666019037e-02);
    const double cs = __builtin_fma(z * z, pc, __builtin_fma(z, -0.5, 1.0));
    // the quadrant k mod 4 without converting k itself (a finite but absurd phase -- |ph| > 3.4e9 rad -- would overflow the
    // conversion: undefined on the host, found by `make check-sanitize`; same bits as (int)k & 3 wherever that is defined)
    const int q = (int)(k - 4.0 * floor(k * 0.25));
    const double a = (q & 1) ? cs : sn;
    const double b = (q & 1) ? sn : cs;
    *s = (q & 2) ? -a : a;
    *c = ((q + 1) & 2) ? -b : b;
}

}  // namespace sdr
