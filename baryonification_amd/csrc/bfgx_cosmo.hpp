// Host-side background cosmology for the hot path: what the reference gets from pyccl at
// HealpixRunner.py:268-280 (ccl.Cosmology + angular_diameter_distance + CubicSpline),
// HealpixRunner.py:296 and BaryonCorrection.py:370 (MassDef.get_radius).
// Flat wCDM + photons + massless neutrinos with pyccl-2.x defaults; CCL physical constants.
// The closed forms (E^2, R_Delta) are mirrored on the device in bfgx_kernels (halo_prep).
#pragma once
#include <cmath>
#include <vector>
#include "../../include/bfgx.h"

namespace bfgx {

constexpr double kClight      = 299792458.0;
constexpr double kGnewt       = 6.67408e-11;
constexpr double kSolarMass   = 1.9884754153381438e30;
constexpr double kMpcToMeter  = 3.085677581491367399198952281e22;
constexpr double kStBoltz     = 5.670367e-8;
constexpr double kPi          = 3.141592653589793238462643383279502884197;
constexpr double kRhoCritical = ((3.0 * 100.0 * 100.0) / (8.0 * kPi * kGnewt)) * (1000.0 * 1000.0 * kMpcToMeter / kSolarMass);
constexpr int    kDaKnots     = 1000;     // HealpixRunner.py:279 linspace(0, 30, 1000)
constexpr double kDaZmax      = 30.0;

// closed-form background, plain-old-data so it can be passed to kernels by value
struct Background {
    double Omega_m, Omega_l, Omega_r, w0, h;
    double rho_crit0;            // RHO_CRITICAL * h^2   [Msun / Mpc^3]
    double hubble_dist;          // c / H0               [Mpc]
};

inline Background make_background(const bfgx_cosmo &c)
{
    Background b;
    const double T = (c.T_CMB > 0) ? c.T_CMB : 2.725;
    const double Neff = (c.Neff >= 0) ? c.Neff : 3.046;
    const double H0 = c.h * 1e5 / kMpcToMeter;                      // 1/s
    const double rho_crit_si = 3.0 * H0 * H0 / (8.0 * kPi * kGnewt);
    const double Og = 4.0 * kStBoltz / (kClight * kClight * kClight) * T * T * T * T / rho_crit_si;
    const double Tnu = T * std::pow(4.0 / 11.0, 1.0 / 3.0);
    const double Onu = Neff * 7.0 / 8.0 * 4.0 * kStBoltz / (kClight * kClight * kClight) * Tnu * Tnu * Tnu * Tnu / rho_crit_si;
    b.Omega_m = c.Omega_m;
    b.Omega_r = Og + Onu;
    b.Omega_l = 1.0 - c.Omega_m - b.Omega_r;
    b.w0 = c.w0;
    b.h = c.h;
    b.rho_crit0 = kRhoCritical * c.h * c.h;
    b.hubble_dist = kClight / 1e5 / c.h;
    return b;
}

inline double E2(const Background &b, double a)
{
    const double a3 = a * a * a;
    return b.Omega_m / a3 + b.Omega_l * std::pow(a, -3.0 * (1.0 + b.w0)) + b.Omega_r / (a3 * a);
}

// ccl MassDef.get_radius: physical Mpc
inline double radius_delta(const Background &b, const bfgx_massdef &md, double M, double a)
{
    double rho = b.rho_crit0 * E2(b, a);
    if (md.rho_type == 1) rho = b.rho_crit0 * b.Omega_m / (a * a * a);
    return std::cbrt(M / (4.18879020479 * md.Delta * rho));
}

// 16-point Gauss-Legendre on [-1,1]
static const double kGLx[8] = {0.0950125098376374401853193, 0.2816035507792589132304605, 0.4580167776572273863424194,
                               0.6178762444026437484466718, 0.7554044083550030338951012, 0.8656312023878317438804679,
                               0.9445750230732325760779884, 0.9894009349916499325961542};
static const double kGLw[8] = {0.1894506104550684962853967, 0.1826034150449235888667637, 0.1691565193950025381893121,
                               0.1495959888165767320815017, 0.1246289712555338720524763, 0.0951585116824927848099251,
                               0.0622535239386478928628438, 0.0271524594117540948517806};

inline double inv_E_of_z(const Background &b, double z) { return 1.0 / std::sqrt(E2(b, 1.0 / (1.0 + z))); }

inline double chi_panel(const Background &b, double lo, double hi)
{
    const double hw = 0.5 * (hi - lo), mid = 0.5 * (hi + lo);
    double s = 0.0;
    for (int i = 0; i < 8; ++i)
        s += kGLw[i] * (inv_E_of_z(b, mid - hw * kGLx[i]) + inv_E_of_z(b, mid + hw * kGLx[i]));
    return s * hw * b.hubble_dist;
}

// comoving distance [Mpc]; panels no wider than dz_max keep 16-pt GL at ~1e-15
inline double comoving_distance(const Background &b, double z)
{
    const double dz_max = 0.05;
    const int np = (int)std::ceil(z / dz_max);
    double chi = 0.0;
    for (int i = 0; i < np; ++i) chi += chi_panel(b, z * i / np, z * (i + 1) / np);
    return chi;
}

inline double angular_diameter_distance(const Background &b, double z) { return comoving_distance(b, z) / (1.0 + z); }

// Not-a-knot cubic spline through (x_i, y_i), as scipy.interpolate.CubicSpline (default bc_type).
// coef[i] = {c3, c2, c1, c0} on [x_i, x_{i+1}], value = ((c3 t + c2) t + c1) t + c0, t = x - x_i.
inline void notaknot_spline(const std::vector<double> &x, const std::vector<double> &y, std::vector<double> &coef)
{
    const int n = (int)x.size();
    std::vector<double> dx(n - 1), m(n - 1), lo(n, 0.0), di(n, 0.0), up(n, 0.0), rhs(n, 0.0), s(n, 0.0);
    for (int i = 0; i < n - 1; ++i) { dx[i] = x[i + 1] - x[i]; m[i] = (y[i + 1] - y[i]) / dx[i]; }
    for (int i = 1; i < n - 1; ++i) {
        lo[i] = dx[i]; di[i] = 2.0 * (dx[i - 1] + dx[i]); up[i] = dx[i - 1];
        rhs[i] = 3.0 * (dx[i] * m[i - 1] + dx[i - 1] * m[i]);
    }
    {   // not-a-knot at both ends
        double d = x[2] - x[0];
        di[0] = dx[1]; up[0] = d;
        rhs[0] = ((dx[0] + 2.0 * d) * dx[1] * m[0] + dx[0] * dx[0] * m[1]) / d;
        d = x[n - 1] - x[n - 3];
        di[n - 1] = dx[n - 3]; lo[n - 1] = d;
        rhs[n - 1] = (dx[n - 2] * dx[n - 2] * m[n - 3] + (2.0 * d + dx[n - 2]) * dx[n - 3] * m[n - 2]) / d;
    }
    // Thomas with partial pivoting restricted to the first/last row pattern is not needed for the
    // uniform knot grid used here; plain forward elimination is stable (checked against scipy).
    for (int i = 1; i < n; ++i) {
        const double w = lo[i] / di[i - 1];
        di[i] -= w * up[i - 1];
        rhs[i] -= w * rhs[i - 1];
    }
    s[n - 1] = rhs[n - 1] / di[n - 1];
    for (int i = n - 2; i >= 0; --i) s[i] = (rhs[i] - up[i] * s[i + 1]) / di[i];
    coef.resize((size_t)(n - 1) * 4);
    for (int i = 0; i < n - 1; ++i) {
        const double t = (s[i] + s[i + 1] - 2.0 * m[i]) / dx[i];
        coef[4 * i + 0] = t / dx[i];
        coef[4 * i + 1] = (m[i] - s[i]) / dx[i] - t;
        coef[4 * i + 2] = s[i];
        coef[4 * i + 3] = y[i];
    }
}

inline void da_spline(const Background &b, std::vector<double> &knots, std::vector<double> &coef)
{
    knots.resize(kDaKnots);
    std::vector<double> y(kDaKnots);
    double chi = 0.0;
    const double step = kDaZmax / (kDaKnots - 1);
    for (int i = 0; i < kDaKnots; ++i) {
        const double z = (i == kDaKnots - 1) ? kDaZmax : i * step;      // np.linspace end-point exact
        if (i > 0) chi += chi_panel(b, knots[i - 1], z);
        knots[i] = z;
        y[i] = chi / (1.0 + z);
    }
    notaknot_spline(knots, y, coef);
}

inline double da_eval(const std::vector<double> &knots, const std::vector<double> &coef, double z)
{
    const int n = (int)knots.size();
    int i = (int)std::floor(z / (kDaZmax / (n - 1)));
    if (i < 0) i = 0;
    if (i > n - 2) i = n - 2;
    while (i > 0 && z < knots[i]) --i;
    while (i < n - 2 && z >= knots[i + 1]) ++i;
    const double t = z - knots[i];
    const double *c = &coef[4 * (size_t)i];
    return ((c[0] * t + c[1]) * t + c[2]) * t + c[3];
}

}  // namespace bfgx
