// Pieces shared by the PCPS translation units (pcps.hip and pcps_fused.hip): the (value, index) record of the map-free
// search, complex helpers, the wave-wide maximum and the radix-5 codelet of the register-resident transforms.
#pragma once

#include <hip/hip_runtime.h>

#include <type_traits>

namespace {

struct Best {
    double v;
    long long i;
};

__device__ __forceinline__ double2 cadd(double2 a, double2 b) { return make_double2(a.x + b.x, a.y + b.y); }
__device__ __forceinline__ double2 csub(double2 a, double2 b) { return make_double2(a.x - b.x, a.y - b.y); }

// (value, index) maximum over the 64 lanes of a wave, left in lane 63: larger value, smaller index on ties.  DPP
// row shifts inside each row of 16 lanes, then row broadcasts -- no LDS traffic.
__device__ __forceinline__ void wave_best(double& v, int& i) {
    auto step = [&](auto ctrl, auto rows) {
        constexpr int C = decltype(ctrl)::value, RM = decltype(rows)::value;
        const int lo = __builtin_amdgcn_update_dpp(__double2loint(v), __double2loint(v), C, RM, 0xf, false);
        const int hi = __builtin_amdgcn_update_dpp(__double2hiint(v), __double2hiint(v), C, RM, 0xf, false);
        const int oi = __builtin_amdgcn_update_dpp(i, i, C, RM, 0xf, false);
        const double ov = __hiloint2double(hi, lo);
        const bool take = ov > v || (ov == v && oi < i);
        v = take ? ov : v;
        i = take ? oi : i;
    };
    step(std::integral_constant<int, 0x111>{}, std::integral_constant<int, 0xf>{});  // row_shr:1
    step(std::integral_constant<int, 0x112>{}, std::integral_constant<int, 0xf>{});  // row_shr:2
    step(std::integral_constant<int, 0x114>{}, std::integral_constant<int, 0xf>{});  // row_shr:4
    step(std::integral_constant<int, 0x118>{}, std::integral_constant<int, 0xf>{});  // row_shr:8  -> lane 15 of each row holds the row's best
    step(std::integral_constant<int, 0x142>{}, std::integral_constant<int, 0xa>{});  // row_bcast:15 into rows 1 and 3
    step(std::integral_constant<int, 0x143>{}, std::integral_constant<int, 0xc>{});  // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave's best
}

}  // namespace

namespace {
namespace fast25k {

// Transform arithmetic with fused multiply-adds spelled out (pcps.hip is compiled with -ffp-contract=off for the
// stages that must round like NumPy; a transform is free arithmetic and an FMA only makes it more accurate).
__device__ __forceinline__ double2 cmulf(double2 a, double2 b) {       // a * b
    return make_double2(__builtin_fma(-a.y, b.y, a.x * b.x), __builtin_fma(a.y, b.x, a.x * b.y));
}
__device__ __forceinline__ double2 cmul_conj(double2 a, double2 w) {   // a * conj(w)
    return make_double2(__builtin_fma(a.y, w.y, a.x * w.x), __builtin_fma(-a.x, w.y, a.y * w.x));
}
// Inverse radix-5 butterfly: out[k] = sum_t v[t] exp(+2 pi i t k / 5); 36 instructions (the contracted-off general one: 48).
__device__ __forceinline__ void ibf5(double2* v) {
    const double c1 = 0.30901699437494742410, c2 = -0.80901699437494742410;  // cos(2pi/5), cos(4pi/5)
    const double s1 = 0.95105651629515357212, s2 = 0.58778525229247312917;   // sin(2pi/5), sin(4pi/5)
    const double2 a1 = cadd(v[1], v[4]), b1 = csub(v[1], v[4]);
    const double2 a2 = cadd(v[2], v[3]), b2 = csub(v[2], v[3]);
    const double2 m1 = make_double2(__builtin_fma(c2, a2.x, __builtin_fma(c1, a1.x, v[0].x)), __builtin_fma(c2, a2.y, __builtin_fma(c1, a1.y, v[0].y)));
    const double2 m2 = make_double2(__builtin_fma(c1, a2.x, __builtin_fma(c2, a1.x, v[0].x)), __builtin_fma(c1, a2.y, __builtin_fma(c2, a1.y, v[0].y)));
    const double2 r1 = make_double2(__builtin_fma(s2, b2.x, s1 * b1.x), __builtin_fma(s2, b2.y, s1 * b1.y));
    const double2 r2 = make_double2(__builtin_fma(-s1, b2.x, s2 * b1.x), __builtin_fma(-s1, b2.y, s2 * b1.y));
    v[0] = cadd(v[0], cadd(a1, a2));
    v[1] = make_double2(m1.x - r1.y, m1.y + r1.x);       // m1 + i r1
    v[4] = make_double2(m1.x + r1.y, m1.y - r1.x);
    v[2] = make_double2(m2.x - r2.y, m2.y + r2.x);
    v[3] = make_double2(m2.x + r2.y, m2.y - r2.x);
}

}  // namespace fast25k
}  // namespace
