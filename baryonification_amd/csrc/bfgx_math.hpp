// fp64 device math tuned for the per-pair loop: straight-line (branch-free) sequences built on the
// gfx950 hardware seeds v_rcp_f64 / v_rsq_f64 plus Newton steps, a frexp + atanh-series log and a
// Taylor sincos for the small azimuth differences inside a disc.  All are accurate to a few ulp
// (|rel err| <~ 4e-16), well inside the 1e-10 parity budget of the fp64-accumulator tests.
#pragma once
#include <hip/hip_runtime.h>

namespace bfgx {

// 1/x for normal x (2 Newton steps on the hardware seed)
__device__ inline double fast_rcp(double x)
{
    double y = __builtin_amdgcn_rcp(x);
    double e = __builtin_fma(-x, y, 1.0);
    y = __builtin_fma(y, e, y);
    e = __builtin_fma(-x, y, 1.0);
    return __builtin_fma(y, e, y);
}

// 1/sqrt(x) for normal x > 0
__device__ inline double fast_rsq(double x)
{
    double y = __builtin_amdgcn_rsq(x);
    double h = 0.5 * y;
    double e = __builtin_fma(-x * y, h, 0.5);     // 0.5 - 0.5 x y^2
    y = __builtin_fma(y, e, y);
    h = 0.5 * y;
    e = __builtin_fma(-x * y, h, 0.5);
    return __builtin_fma(y, e, y);
}

// ln(x) for finite normal x > 0:  x = m 2^e, m in [sqrt(1/2), sqrt(2)),  ln m = 2 atanh((m-1)/(m+1))
__device__ inline double fast_log(double x)
{
    double m = __builtin_amdgcn_frexp_mant(x);            // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool lowm = m < 0.70710678118654752440;
    m = lowm ? 2.0 * m : m;
    e = lowm ? e - 1 : e;
    const double s = (m - 1.0) * fast_rcp(m + 1.0);
    const double u = s * s;
    double p = 1.0 / 21.0;
    p = __builtin_fma(p, u, 1.0 / 19.0);
    p = __builtin_fma(p, u, 1.0 / 17.0);
    p = __builtin_fma(p, u, 1.0 / 15.0);
    p = __builtin_fma(p, u, 1.0 / 13.0);
    p = __builtin_fma(p, u, 1.0 / 11.0);
    p = __builtin_fma(p, u, 1.0 / 9.0);
    p = __builtin_fma(p, u, 1.0 / 7.0);
    p = __builtin_fma(p, u, 1.0 / 5.0);
    p = __builtin_fma(p, u, 1.0 / 3.0);
    const double lm = __builtin_fma(2.0 * s * u, p, 2.0 * s);
    const double fe = (double)e;
    // ln 2 split so that fe * hi is exact for |e| < 2^11
    return __builtin_fma(fe, 0.693147180369123816490, __builtin_fma(fe, 1.90821492927058770002e-10, lm));
}

// sin/cos for |x| <= 0.5 (Taylor to x^15 / x^16: truncation < 3e-20)
__device__ inline void sincos_small(double x, double &s, double &c)
{
    const double u = x * x;
    double ps = -1.0 / 1307674368000.0;
    ps = __builtin_fma(ps, u, 1.0 / 6227020800.0);
    ps = __builtin_fma(ps, u, -1.0 / 39916800.0);
    ps = __builtin_fma(ps, u, 1.0 / 362880.0);
    ps = __builtin_fma(ps, u, -1.0 / 5040.0);
    ps = __builtin_fma(ps, u, 1.0 / 120.0);
    ps = __builtin_fma(ps, u, -1.0 / 6.0);
    s = __builtin_fma(x * u, ps, x);
    double pc = 1.0 / 20922789888000.0;
    pc = __builtin_fma(pc, u, -1.0 / 87178291200.0);
    pc = __builtin_fma(pc, u, 1.0 / 479001600.0);
    pc = __builtin_fma(pc, u, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, u, 1.0 / 40320.0);
    pc = __builtin_fma(pc, u, -1.0 / 720.0);
    pc = __builtin_fma(pc, u, 1.0 / 24.0);
    pc = __builtin_fma(pc, u, -0.5);
    c = __builtin_fma(u, pc, 1.0);
}

// sin/cos of an azimuth difference in (-2 pi, 2 pi): fold to (-pi, pi], Taylor when small, libm otherwise
__device__ inline void sincos_dphi(double x, double &s, double &c)
{
    constexpr double kPi_ = 3.141592653589793238462643383279502884197;
    x = (x > kPi_) ? x - 2.0 * kPi_ : x;
    x = (x < -kPi_) ? x + 2.0 * kPi_ : x;
    if (__builtin_expect(fabs(x) <= 0.5, 1)) sincos_small(x, s, c);
    else sincos(x, &s, &c);
}

}  // namespace bfgx
