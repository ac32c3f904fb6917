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

// exp(x) for any double: x = k ln2 + r, |r| <= ln2 / 2 (two-term ln 2, FMA), Taylor to r^13 (truncation < 3e-18 relative),
// scaling by ldexp.  Branch-free; +inf above 709.78, 0 below -745.2, NaN stays NaN.  ~1 ulp.
__device__ inline double fast_exp(double x)
{
    const double xc = fmin(fmax(x, -746.0), 710.0);                         // keeps k in int range; NaN propagates through fmin/fmax? no:
    const double xx = (x != x) ? x : xc;                                    // fmin/fmax drop NaN, so put it back
    const double k = __builtin_rint(xx * 1.44269504088896338700);           // 1 / ln 2
    double r = __builtin_fma(-k, 0.693147180369123816490, xx);
    r = __builtin_fma(-k, 1.90821492927058770002e-10, r);
    double p = 1.0 / 6227020800.0;                                          // r^13 / 13!
    p = __builtin_fma(p, r, 1.0 / 479001600.0);
    p = __builtin_fma(p, r, 1.0 / 39916800.0);
    p = __builtin_fma(p, r, 1.0 / 3628800.0);
    p = __builtin_fma(p, r, 1.0 / 362880.0);
    p = __builtin_fma(p, r, 1.0 / 40320.0);
    p = __builtin_fma(p, r, 1.0 / 5040.0);
    p = __builtin_fma(p, r, 1.0 / 720.0);
    p = __builtin_fma(p, r, 1.0 / 120.0);
    p = __builtin_fma(p, r, 1.0 / 24.0);
    p = __builtin_fma(p, r, 1.0 / 6.0);
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    // two-step scaling so that results in the subnormal range round once
    const int ki = (int)k, k1 = ki / 2, k2 = ki - k1;
    return __builtin_amdgcn_ldexp(__builtin_amdgcn_ldexp(p, k1), k2);
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

// sin/cos for |x| <~ 1e3: Cody-Waite reduction by pi/2 (two-term, FMA) + Taylor on [-pi/4, pi/4]; branch-free
__device__ inline void sincos_bounded(double x, double &s, double &c)
{
    const double k = __builtin_rint(x * 0.63661977236758134308);           // 2/pi
    double r = __builtin_fma(-k, 1.57079632679489655800e+00, x);
    r = __builtin_fma(-k, 6.12323399573676603587e-17, r);
    const double u = r * r;
    double ps = -1.0 / 121645100408832000.0;                               // r^19
    ps = __builtin_fma(ps, u, 1.0 / 355687428096000.0);
    ps = __builtin_fma(ps, u, -1.0 / 1307674368000.0);
    ps = __builtin_fma(ps, u, 1.0 / 6227020800.0);
    ps = __builtin_fma(ps, u, -1.0 / 39916800.0);
    ps = __builtin_fma(ps, u, 1.0 / 362880.0);
    ps = __builtin_fma(ps, u, -1.0 / 5040.0);
    ps = __builtin_fma(ps, u, 1.0 / 120.0);
    ps = __builtin_fma(ps, u, -1.0 / 6.0);
    const double sr = __builtin_fma(r * u, ps, r);
    double pc = 1.0 / 2432902008176640000.0;                               // r^20
    pc = __builtin_fma(pc, u, -1.0 / 6402373705728000.0);
    pc = __builtin_fma(pc, u, 1.0 / 20922789888000.0);
    pc = __builtin_fma(pc, u, -1.0 / 87178291200.0);
    pc = __builtin_fma(pc, u, 1.0 / 479001600.0);
    pc = __builtin_fma(pc, u, -1.0 / 3628800.0);
    pc = __builtin_fma(pc, u, 1.0 / 40320.0);
    pc = __builtin_fma(pc, u, -1.0 / 720.0);
    pc = __builtin_fma(pc, u, 1.0 / 24.0);
    pc = __builtin_fma(pc, u, -0.5);
    const double cr = __builtin_fma(u, pc, 1.0);
    const int q = (int)k & 3;
    const double s0 = (q & 1) ? cr : sr, c0 = (q & 1) ? sr : cr;
    s = (q & 2) ? -s0 : s0;
    c = ((q + 1) & 2) ? -c0 : c0;
}

// sin/cos of an azimuth difference in (-2 pi, 2 pi): fold to (-pi, pi], Taylor when small, Cody-Waite otherwise
__device__ inline void sincos_dphi(double x, double &s, double &c)
{
    constexpr double kPi_ = 3.141592653589793238462643383279502884197;
    x = (x > kPi_) ? x - 2.0 * kPi_ : x;
    x = (x < -kPi_) ? x + 2.0 * kPi_ : x;
    if (__builtin_expect(fabs(x) <= 0.5, 1)) sincos_small(x, s, c);
    else sincos_bounded(x, s, c);
}

// atan(t) for |t| <= 0.1 (series to t^15: truncation < 6e-18)
__device__ inline double atan_small(double t)
{
    const double u = t * t;
    double p = -1.0 / 15.0;
    p = __builtin_fma(p, u, 1.0 / 13.0);
    p = __builtin_fma(p, u, -1.0 / 11.0);
    p = __builtin_fma(p, u, 1.0 / 9.0);
    p = __builtin_fma(p, u, -1.0 / 7.0);
    p = __builtin_fma(p, u, 1.0 / 5.0);
    p = __builtin_fma(p, u, -1.0 / 3.0);
    return __builtin_fma(t * u, p, t);
}

// asin(q) for |q| <= 0.05 (series to q^11: truncation < 2e-18)
__device__ inline double asin_small(double q)
{
    const double u = q * q;
    double p = 63.0 / 2816.0;
    p = __builtin_fma(p, u, 35.0 / 1152.0);
    p = __builtin_fma(p, u, 5.0 / 112.0);
    p = __builtin_fma(p, u, 3.0 / 40.0);
    p = __builtin_fma(p, u, 1.0 / 6.0);
    return __builtin_fma(q * u, p, q);
}

// atan2(y, x) for finite arguments, not both zero: octant reduction, tangent half-angle steps down to
// |t| <= 0.1, then the series.  Register-light (no libm); used on rare paths only.
__device__ inline double atan2_generic(double y, double x)
{
    const double ay = fabs(y), ax = fabs(x);
    const bool swap = ay > ax;
    const double num = swap ? ax : ay, den = swap ? ay : ax;
    double t = (den > 0.0) ? num * fast_rcp(den) : 0.0;                     // in [0, 1]
    double scale = 1.0;
    for (int it = 0; it < 4 && t > 0.1; ++it) {                             // atan t = 2 atan(t / (1 + sqrt(1 + t^2)))
        const double hyp = 1.0 + t * t;
        t = t * fast_rcp(1.0 + hyp * fast_rsq(hyp));
        scale *= 2.0;
    }
    double a = scale * atan_small(t);
    a = swap ? 1.570796326794896619231321691639751442099 - a : a;
    a = (x < 0.0) ? 3.141592653589793238462643383279502884197 - a : a;
    return (y < 0.0) ? -a : a;
}

// acos(c) for |c| <= 1, branch-free: fdlibm's rational approximation of asin on [0, 0.5] with the
// half-angle identity acos(c) = 2 asin(sqrt((1 - c) / 2)) beyond (a few ulp)
__device__ inline double acos_fast(double c)
{
    const double ac = fabs(c);
    const bool big = ac > 0.5;
    const double z = big ? 0.5 * (1.0 - ac) : c * c;
    double p = 3.47933107596021167570e-05;
    p = __builtin_fma(p, z, 7.91534994289814532176e-04);
    p = __builtin_fma(p, z, -4.00555345006794114027e-02);
    p = __builtin_fma(p, z, 2.01212532134862925881e-01);
    p = __builtin_fma(p, z, -3.25565818622400915405e-01);
    p = __builtin_fma(p, z, 1.66666666666666657415e-01);
    double q = 7.70381505559019352791e-02;
    q = __builtin_fma(q, z, -6.88283971605453293030e-01);
    q = __builtin_fma(q, z, 2.02094576023350569471e+00);
    q = __builtin_fma(q, z, -2.40339491173441421878e+00);
    q = __builtin_fma(q, z, 1.0);
    const double R = z * p * fast_rcp(q);
    const double zs = z > 0.0 ? z : 1.0;
    const double s = big ? zs * fast_rsq(zs) : c;                           // sqrt(z) or c
    const double as = __builtin_fma(s, R, s);                               // asin(s)
    const double kHalfPi_ = 1.570796326794896619231321691639751442099, kPi_ = 3.141592653589793238462643383279502884197;
    const double r_small = kHalfPi_ - as;
    const double r_big = (c > 0.0) ? 2.0 * as : kPi_ - 2.0 * as;
    return big ? ((z > 0.0) ? r_big : (c > 0.0 ? 0.0 : kPi_)) : r_small;
}

// sqrt(x) for normal x > 0 via the rsq seed; returns 0 for x <= 0
__device__ inline double fast_sqrt(double x)
{
    const double xs = x > 0.0 ? x : 1.0;
    const double r = xs * fast_rsq(xs);
    return x > 0.0 ? r : 0.0;
}

// a * b and a + b that may not be contracted into an FMA (numpy rounds every operation separately)
__device__ inline double mul_nc(double a, double b)
{
#pragma clang fp contract(off)
    return a * b;
}
__device__ inline double add_nc(double a, double b)
{
#pragma clang fp contract(off)
    return a + b;
}

}  // namespace bfgx
