// HIP kernels for gfx950 (MI355X): the per-halo HEALPix-shell hot path of BaryonForge.
//
//   halo_prep_kernel      K0  per-halo scalars: HealpixRunner.py:293-305 (+ BaryonCorrection.py:364-370)
//   halo_scatter_kernel   K1  BaryonifyShell loop  HealpixRunner.py:306-331   (MODE_OFFSETS)
//                         K3  PaintProfilesShell   HealpixRunner.py:432-445   (MODE_PAINT)
//                             pair census                                      (MODE_COUNT)
//   regrid_kernel         K2  HealpixRunner.py:333-341 + regrid_pixels_hpix :60-64
//   sum2_kernel               HealpixRunner.py:344-345 (the two sums of the mass-conservation check)
//
// Design (one wavefront = one halo): the 64 lanes first act as 64 HEALPix rings of the halo's disc
// (ring -> first pixel, phi-range, pixel count), a wave-wide prefix sum turns the ragged rows into one
// flat pair index space, and the lanes then sweep that space 64 pairs at a time (row found by a 6-step
// binary search in LDS).  All geometry is fp64 (the chord D*(v_p - v_j) and nw_vec - vec are differences
// of nearly equal unit vectors); only the final accumulate is fp32 (or fp64) global atomics.
// Geometry follows the published HEALPix RING algorithms (healpix_cxx: ring_above, get_ring_info2,
// query_disc with fact=0, get_interpol, pix2loc).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bfgx_cosmo.hpp"

namespace bfgx {

constexpr double kTwoPi    = 6.283185307179586476925286766559005768394;
constexpr double kHalfPi   = 1.570796326794896619231321691639751442099;
constexpr double kInvTwoPi = 1.0 / kTwoPi;
constexpr double kTwoThird = 2.0 / 3.0;
constexpr double kDeg2Rad  = kPi / 180.0;
constexpr double kRad2Deg  = 180.0 / kPi;

constexpr int kWave = 64;
constexpr int kWavesPerBlock = 4;
constexpr int kNC = 4;                 // (z, M) corner rows per halo for a 3-axis table

enum { MODE_OFFSETS = 0, MODE_PAINT = 1, MODE_COUNT = 2 };

// ---------------------------------------------------------------------------------- HEALPix (RING)
struct Hpx {
    int64_t nside, npix, ncap;
    double fact1, fact2;
};

__host__ __device__ inline Hpx make_hpx(int64_t nside)
{
    Hpx h;
    h.nside = nside;
    h.npix = 12 * nside * nside;
    h.ncap = 2 * nside * (nside - 1);
    h.fact2 = 4.0 / (double)h.npix;
    h.fact1 = (double)(nside << 1) * h.fact2;
    return h;
}

__device__ inline int64_t ring_above(const Hpx &h, double z)
{
    const double az = fabs(z);
    if (az <= kTwoThird) return (int64_t)((double)h.nside * (2.0 - 1.5 * z));
    const int64_t iring = (int64_t)((double)h.nside * sqrt(3.0 * (1.0 - az)));
    return (z > 0) ? iring : 4 * h.nside - iring - 1;
}

__device__ inline void ring_info_small(const Hpx &h, int64_t ring, int64_t &start, int64_t &nr, bool &shifted)
{
    if (ring < h.nside) {
        shifted = true; nr = 4 * ring; start = 2 * ring * (ring - 1);
    } else if (ring < 3 * h.nside) {
        shifted = ((ring - h.nside) & 1) == 0; nr = 4 * h.nside; start = h.ncap + (ring - h.nside) * nr;
    } else {
        const int64_t q = 4 * h.nside - ring;
        shifted = true; nr = 4 * q; start = h.npix - 2 * q * (q + 1);
    }
}

// colatitude of a ring the way get_interpol needs it (healpix_cxx get_ring_info2)
__device__ inline void ring_info2(const Hpx &h, int64_t ring, int64_t &start, int64_t &nr, double &theta, bool &shifted)
{
    const int64_t northring = (ring > 2 * h.nside) ? 4 * h.nside - ring : ring;
    if (northring < h.nside) {
        const double tmp = (double)(northring * northring) * h.fact2;
        theta = atan2(sqrt(tmp * (2.0 - tmp)), 1.0 - tmp);
        nr = 4 * northring; shifted = true; start = 2 * northring * (northring - 1);
    } else {
        theta = acos((double)(2 * h.nside - northring) * h.fact1);
        nr = 4 * h.nside; shifted = ((northring - h.nside) & 1) == 0;
        start = h.ncap + (northring - h.nside) * nr;
    }
    if (northring != ring) { theta = kPi - theta; start = h.npix - start - nr; }
}

// z and sin(theta) of a ring (healpix_cxx ring2z + the pix2loc small-angle form near the poles)
__device__ inline void ring_z_sth(const Hpx &h, int64_t ring, double &z, double &sth)
{
    if (ring < h.nside) {
        const double tmp = (double)(ring * ring) * h.fact2;
        z = 1.0 - tmp;
        sth = (z > 0.99) ? sqrt(tmp * (2.0 - tmp)) : sqrt((1.0 - z) * (1.0 + z));
    } else if (ring <= 3 * h.nside) {
        z = (double)(2 * h.nside - ring) * h.fact1;
        sth = sqrt((1.0 - z) * (1.0 + z));
    } else {
        const int64_t q = 4 * h.nside - ring;
        const double tmp = (double)(q * q) * h.fact2;
        z = tmp - 1.0;
        sth = (z < -0.99) ? sqrt(tmp * (2.0 - tmp)) : sqrt((1.0 - z) * (1.0 + z));
    }
}

__device__ inline int64_t isqrt64(int64_t v) { return (int64_t)sqrt((double)v + 0.5); }

// healpix_cxx pix2loc (RING): z, sin(theta), phi of a pixel centre
__device__ inline void pix2loc(const Hpx &h, int64_t pix, double &z, double &sth, double &phi)
{
    if (pix < h.ncap) {
        const int64_t iring = (1 + isqrt64(1 + 2 * pix)) >> 1;
        const int64_t iphi = (pix + 1) - 2 * iring * (iring - 1);
        const double tmp = (double)(iring * iring) * h.fact2;
        z = 1.0 - tmp;
        sth = (z > 0.99) ? sqrt(tmp * (2.0 - tmp)) : sqrt((1.0 - z) * (1.0 + z));
        phi = ((double)iphi - 0.5) * kHalfPi / (double)iring;
    } else if (pix < h.npix - h.ncap) {
        const int64_t nl4 = 4 * h.nside;
        const int64_t ip = pix - h.ncap;
        const int64_t tmp = ip / nl4;
        const int64_t iring = tmp + h.nside, iphi = ip - nl4 * tmp + 1;
        const double fodd = ((iring + h.nside) & 1) ? 1.0 : 0.5;
        z = (double)(2 * h.nside - iring) * h.fact1;
        sth = sqrt((1.0 - z) * (1.0 + z));
        phi = ((double)iphi - fodd) * kPi * 0.75 * h.fact1;
    } else {
        const int64_t ip = h.npix - pix;
        const int64_t iring = (1 + isqrt64(2 * ip - 1)) >> 1;
        const int64_t iphi = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
        const double tmp = (double)(iring * iring) * h.fact2;
        z = tmp - 1.0;
        sth = (z < -0.99) ? sqrt(tmp * (2.0 - tmp)) : sqrt((1.0 - z) * (1.0 + z));
        phi = ((double)iphi - 0.5) * kHalfPi / (double)iring;
    }
}

// healpix_cxx get_interpol (RING). want_w = false skips the ring colatitudes (pixels only).
template <bool WANT_W>
__device__ inline void get_interpol(const Hpx &h, double theta, double phi, int64_t pix[4], double wgt[4])
{
    const double z = cos(theta);
    const int64_t ir1 = ring_above(h, z);
    const int64_t ir2 = ir1 + 1;
    double theta1 = 0.0, theta2 = 0.0;
    pix[0] = pix[1] = pix[2] = pix[3] = 0;
    wgt[0] = wgt[1] = wgt[2] = wgt[3] = 0.0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int64_t ir = half ? ir2 : ir1;
        if (half ? (ir < 4 * h.nside) : (ir > 0)) {
            int64_t sp, nr; bool shifted; double th = 0.0;
            if (WANT_W) ring_info2(h, ir, sp, nr, th, shifted);
            else ring_info_small(h, ir, sp, nr, shifted);
            const double dphi = kTwoPi / (double)nr;
            const double sh = shifted ? 0.5 : 0.0;
            const double tmp = phi / dphi - sh;
            int64_t i1 = (tmp < 0) ? (int64_t)tmp - 1 : (int64_t)tmp;
            const double w1 = (phi - ((double)i1 + sh) * dphi) / dphi;
            int64_t i2 = i1 + 1;
            if (i1 < 0) i1 += nr;
            if (i2 >= nr) i2 -= nr;
            pix[2 * half] = sp + i1; pix[2 * half + 1] = sp + i2;
            wgt[2 * half] = 1.0 - w1; wgt[2 * half + 1] = w1;
            if (half) theta2 = th; else theta1 = th;
        }
    }
    if (ir1 == 0) {
        const double wtheta = theta / theta2;
        wgt[2] *= wtheta; wgt[3] *= wtheta;
        const double fac = (1.0 - wtheta) * 0.25;
        wgt[0] = fac; wgt[1] = fac; wgt[2] += fac; wgt[3] += fac;
        pix[0] = (pix[2] + 2) & 3;
        pix[1] = (pix[3] + 2) & 3;
    } else if (ir2 == 4 * h.nside) {
        const double wtheta = (theta - theta1) / (kPi - theta1);
        wgt[0] *= (1.0 - wtheta); wgt[1] *= (1.0 - wtheta);
        const double fac = wtheta * 0.25;
        wgt[0] += fac; wgt[1] += fac; wgt[2] = fac; wgt[3] = fac;
        pix[2] = ((pix[0] + 2) & 3) + h.npix - 4;
        pix[3] = ((pix[1] + 2) & 3) + h.npix - 4;
    } else {
        const double wtheta = (theta - theta1) / (theta2 - theta1);
        wgt[0] *= (1.0 - wtheta); wgt[1] *= (1.0 - wtheta);
        wgt[2] *= wtheta; wgt[3] *= wtheta;
    }
}

// ---------------------------------------------------------------------------------- device-side model
struct DevTable {
    int32_t ndim;
    int32_t n[BFGX_MAX_DIM];
    const double *axis[BFGX_MAX_DIM];     // device pointers
    const double *values;                 // device pointer
    int32_t rdelta, logv;
    double eps_model;
    int32_t r_uniform;                    // ln r axis is uniform: index guess = (x - r0) * inv_dr
    double r0, inv_dr;
};

struct DevModel {
    DevTable tab;
    Background bg_runner, bg_model;
    bfgx_massdef md_runner, md_model;
    double eps_runner;
    const double *da_coef;                // device, [kDaKnots-1][4]
    double da_step;
};

// per-halo record written by K0, read (wave-uniformly) by K1/K3
struct alignas(16) HaloRec {
    double z0, xa, s0, phi0;              // query_disc pointing: cos/sin colatitude, azimuth in [0, 2pi)
    double cph0, sph0;                    // cos/sin(phi0)
    double cosr;                          // cos(disc radius)
    double theta, phi;                    // lonlat2thetaphi(ra, dec), for the <4-pixel fallback
    double D, a, rcut, lnRmod;
    double w[kNC];                        // (z,M) corner weights in scipy corner order
    int32_t rowoff[kNC];                  // element offset of each corner's radial row in values[]
    int32_t irmin, irmax, rfirst, rlast;  // phi-tested ring range, full row range (incl. polar caps)
    int32_t oob;                          // 1: (z, M[, params]) outside the table -> NaN read-out
    int32_t _pad[3];
};

__device__ inline double dev_E2(const Background &b, double a)
{
    const double a3 = a * a * a;
    return b.Omega_m / a3 + b.Omega_l * pow(a, -3.0 * (1.0 + b.w0)) + b.Omega_r / (a3 * a);
}

__device__ inline double dev_radius(const Background &b, const bfgx_massdef &md, double M, double a)
{
    double rho = b.rho_crit0 * dev_E2(b, a);
    if (md.rho_type == 1) rho = b.rho_crit0 * b.Omega_m / (a * a * a);
    return cbrt(M / (4.18879020479 * md.Delta * rho));
}

// scipy find_indices: largest i with g[i] <= x clipped to [0, n-2]; returns -1 when x is outside
// [g[0], g[n-1]] or NaN (RegularGridInterpolator bounds_error=False, fill_value=nan)
__device__ inline int axis_find(const double *g, int n, double x)
{
    if (!(x >= g[0]) || !(x <= g[n - 1])) return -1;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) {
        const int mid = (lo + hi) >> 1;
        if (x >= g[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

// ---------------------------------------------------------------------------------- K0
__global__ void __launch_bounds__(256)
halo_prep_kernel(DevModel m, Hpx h, int64_t nhalo,
                 const double *__restrict__ M, const double *__restrict__ z,
                 const double *__restrict__ ra, const double *__restrict__ dec,
                 HaloRec *__restrict__ rec)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nhalo) return;
    HaloRec r;
    const double M_j = M[j], z_j = z[j];
    const double a = 1.0 / (1.0 + z_j);                                   // HealpixRunner.py:295
    const double R = dev_radius(m.bg_runner, m.md_runner, M_j, a);        // :296 physical Mpc
    double D;                                                             // :297 CubicSpline D_a(z)
    {
        int i = (int)floor(z_j / m.da_step);
        i = max(0, min(i, kDaKnots - 2));
        const double t = z_j - (double)i * m.da_step;
        const double *c = m.da_coef + 4 * i;
        D = ((c[0] * t + c[1]) * t + c[2]) * t + c[3];
    }
    const double radius = R * m.eps_runner / D;                           // :305

    // hp.ang2vec(ra, dec, lonlat=True)  (:303)
    const double theta = kHalfPi - dec[j] * kDeg2Rad;
    const double phi = ra[j] * kDeg2Rad;
    const double st = sin(theta);
    const double vx = st * cos(phi), vy = st * sin(phi), vz = cos(theta);
    // query_disc's pointing(vec) (:306)
    const double thq = atan2(sqrt(vx * vx + vy * vy), vz);
    double phq = (vx == 0.0 && vy == 0.0) ? 0.0 : atan2(vy, vx);
    if (phq < 0) phq += kTwoPi;
    r.z0 = cos(thq);
    r.s0 = sin(thq);
    r.xa = 1.0 / sqrt((1.0 - r.z0) * (1.0 + r.z0));
    r.phi0 = phq;
    r.cph0 = cos(phq); r.sph0 = sin(phq);
    r.theta = theta; r.phi = phi;
    r.D = D; r.a = a;

    const int64_t nl4 = 4 * h.nside;
    if (radius >= kPi) {
        r.cosr = -1.0; r.irmin = (int32_t)nl4; r.irmax = 0; r.rfirst = 1; r.rlast = (int32_t)(nl4 - 1);
    } else {
        r.cosr = cos(radius);
        const double rlat1 = thq - radius;
        int64_t irmin = ring_above(h, cos(rlat1)) + 1;
        const double rlat2 = thq + radius;
        int64_t irmax = ring_above(h, cos(rlat2));
        if (irmax > nl4 - 1) irmax = nl4 - 1;
        r.irmin = (int32_t)irmin; r.irmax = (int32_t)irmax;
        r.rfirst = (int32_t)(((rlat1 <= 0) && (irmin > 1)) ? 1 : irmin);
        r.rlast = (int32_t)(((rlat2 >= kPi) && (irmax + 1 < nl4)) ? nl4 - 1 : irmax);
    }

    // model-side radius and table coordinates (BaryonCorrection.py:364-370, Tabulate.py:279-283)
    const double Rmod = dev_radius(m.bg_model, m.md_model, M_j, a) / a;
    r.rcut = m.tab.eps_model * Rmod;
    r.lnRmod = log(Rmod);
    const double x0 = log(1.0 / a), x1 = log(M_j);
    const int iz = axis_find(m.tab.axis[0], m.tab.n[0], x0);
    const int im = axis_find(m.tab.axis[1], m.tab.n[1], x1);
    r.oob = (iz < 0 || im < 0) ? 1 : 0;
    if (!r.oob) {
        const double *gz = m.tab.axis[0], *gm = m.tab.axis[1];
        const double tz = (x0 - gz[iz]) / (gz[iz + 1] - gz[iz]);
        const double tm = (x1 - gm[im]) / (gm[im + 1] - gm[im]);
        const int nr = m.tab.n[2];
        r.w[0] = (1.0 * (1.0 - tz)) * (1.0 - tm);
        r.w[1] = (1.0 * (1.0 - tz)) * tm;
        r.w[2] = (1.0 * tz) * (1.0 - tm);
        r.w[3] = (1.0 * tz) * tm;
        r.rowoff[0] = (iz * m.tab.n[1] + im) * nr;
        r.rowoff[1] = (iz * m.tab.n[1] + im + 1) * nr;
        r.rowoff[2] = ((iz + 1) * m.tab.n[1] + im) * nr;
        r.rowoff[3] = ((iz + 1) * m.tab.n[1] + im + 1) * nr;
    } else {
        for (int c = 0; c < kNC; ++c) { r.w[c] = 0.0; r.rowoff[c] = 0; }
    }
    r._pad[0] = r._pad[1] = r._pad[2] = 0;
    rec[j] = r;
}

// ---------------------------------------------------------------------------------- K1 / K3
template <typename ACC> __device__ inline void atomic_accumulate(ACC *p, double v) { atomicAdd(p, (ACC)v); }

// linear read-out along ln r of the 4 (z,M)-corner rows; NaN outside the axis (scipy RGI semantics)
__device__ inline double radial_readout(const DevTable &t, const HaloRec &r, double lx)
{
    const double *g = t.axis[2];
    const int n = t.n[2];
    if (!(lx >= g[0]) || !(lx <= g[n - 1])) return __builtin_nan("");
    int i;
    if (t.r_uniform) {
        i = (int)((lx - t.r0) * t.inv_dr);
        i = max(0, min(i, n - 2));
        while (i > 0 && lx < g[i]) --i;
        while (i < n - 2 && lx >= g[i + 1]) ++i;
    } else {
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lx >= g[mid]) lo = mid; else hi = mid; }
        i = lo;
    }
    const double tr = (lx - g[i]) / (g[i + 1] - g[i]);
    double val = 0.0;
#pragma unroll
    for (int c = 0; c < kNC; ++c) {
        const double *row = t.values + r.rowoff[c] + i;
        val = val + row[0] * (r.w[c] * (1.0 - tr));
        val = val + row[1] * (r.w[c] * tr);
    }
    return val;
}

template <int MODE, typename ACC>
__device__ inline void process_pair(const DevModel &m, const HaloRec &r, int64_t pix,
                                    double z, double sth, double phi_pix, ACC *__restrict__ out)
{
    // pixel unit vector in the frame rotated by -phi0 about the polar axis: halo at (s0, 0, z0)
    double sd, cd;
    sincos(phi_pix - r.phi0, &sd, &cd);
    const double vx = sth * cd, vy = sth * sd, vz = z;
    const double dx = r.D * (vx - r.s0), dy = r.D * vy, dz = r.D * (vz - r.z0);   // :314-316
    const double r_sep = sqrt(dx * dx + dy * dy + dz * dz);                         // :317
    const double r_com = r_sep / r.a;                                               // :321
    const double lx = m.tab.rdelta ? (log(r_com) - r.lnRmod) : log(r_com);
    double d = r.oob ? __builtin_nan("") : radial_readout(m.tab, r, lx);
    if (MODE == MODE_PAINT) {
        const double paint = exp(d);                                                // Tabulate.py:286
        if (isfinite(paint) && paint != 0.0) atomic_accumulate(out + pix, paint);   // :442, :445
        return;
    }
    if (!(r_com < r.rcut)) d = 0.0;                                                 // BaryonCorrection.py:381-382
    d *= r.a;                                                                       // :321
    const double inv_r = 1.0 / r_sep;
    double ox = d * (dx * inv_r), oy = d * (dy * inv_r), oz = d * (dz * inv_r);     // :322
    if (!isfinite(ox)) ox = 0.0;                                                    // :323
    if (!isfinite(oy)) oy = 0.0;
    if (!isfinite(oz)) oz = 0.0;
    if (ox == 0.0 && oy == 0.0 && oz == 0.0) return;       // reference adds ~1e-17 rounding noise here
    const double nx = r.D * vx + ox, ny = r.D * vy + oy, nz = r.D * vz + oz;        // :326
    const double inv_n = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);                   // :327
    const double ex = nx * inv_n - vx, ey = ny * inv_n - vy, ez = nz * inv_n - vz;  // :328
    // rotate back by +phi0
    const double gx = ex * r.cph0 - ey * r.sph0;
    const double gy = ex * r.sph0 + ey * r.cph0;
    ACC *o = out + 3 * pix;
    atomic_accumulate(o + 0, gx);                                                   // :331
    atomic_accumulate(o + 1, gy);
    atomic_accumulate(o + 2, ez);
}

struct RowLds {                      // one wave's 64 ring rows
    int32_t prefix[kWave];           // exclusive prefix of pixel counts
    int32_t nr[kWave];               // pixels in ring
    int32_t lo[kWave];               // first in-disc pixel index within the ring, in [0, nr)
    int64_t start[kWave];            // first pixel of ring
    double z[kWave], sth[kWave], shift[kWave];
};

template <int MODE, typename ACC>
__global__ void __launch_bounds__(kWave * kWavesPerBlock)
halo_scatter_kernel(DevModel m, Hpx h, int64_t nhalo, const HaloRec *__restrict__ recs,
                    ACC *__restrict__ out, int64_t *__restrict__ counts, int fallback4)
{
    __shared__ RowLds lds[kWavesPerBlock];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // wave-uniform -> scalar loads of the record
    const int64_t j = (int64_t)blockIdx.x * kWavesPerBlock + wid;
    if (j >= nhalo) return;                      // whole wave exits together
    RowLds &L = lds[wid];
    const HaloRec r = recs[j];

    int64_t npairs = 0;
    for (int rbase = r.rfirst; rbase <= r.rlast; rbase += kWave) {
        const int ring = rbase + lane;
        int cnt = 0, lo = 0;
        int64_t start = 0, nr = 1; bool shifted = false;
        double z = 0.0, sth = 0.0;
        if (ring <= r.rlast) {
            ring_info_small(h, ring, start, nr, shifted);
            ring_z_sth(h, ring, z, sth);
            if (ring < r.irmin || ring > r.irmax) {
                cnt = (int)nr;                   // polar cap rows: whole ring inside the disc
            } else {
                const double x = (r.cosr - z * r.z0) * r.xa;
                const double ysq = 1.0 - z * z - x * x;
                if (ysq > 0.0) {
                    const double dphi = atan2(sqrt(ysq), x);
                    if (dphi > 0.0) {
                        const double sh = shifted ? 0.5 : 0.0;
                        const int64_t ip_lo = (int64_t)floor((double)nr * kInvTwoPi * (r.phi0 - dphi) - sh) + 1;
                        const int64_t ip_hi = (int64_t)floor((double)nr * kInvTwoPi * (r.phi0 + dphi) - sh);
                        int64_t c = ip_hi - ip_lo + 1;
                        c = c < 0 ? 0 : (c > nr ? nr : c);
                        cnt = (int)c;
                        int64_t l = ip_lo % nr; if (l < 0) l += nr;
                        lo = (int)l;
                    }
                }
            }
        }
        // wave-wide inclusive scan of cnt
        int incl = cnt;
#pragma unroll
        for (int s = 1; s < kWave; s <<= 1) {
            const int v = __shfl_up(incl, s, kWave);
            if (lane >= s) incl += v;
        }
        const int total = __shfl(incl, kWave - 1, kWave);
        L.prefix[lane] = incl - cnt;
        L.nr[lane] = (int)nr; L.lo[lane] = lo; L.start[lane] = start;
        L.z[lane] = z; L.sth[lane] = sth; L.shift[lane] = shifted ? 0.5 : 0.0;
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_s_waitcnt(0xc07f);      // lgkmcnt(0): LDS writes visible to the wave

        const bool single_chunk = (rbase == r.rfirst) && (r.rfirst + kWave > r.rlast);
        if (fallback4 && single_chunk && total < 4) break;      // -> 4-neighbour fallback below
        npairs += total;

        if (MODE != MODE_COUNT) {
            for (int t = lane; t < total; t += kWave) {
                int row = 0;                      // largest row with prefix[row] <= t
#pragma unroll
                for (int s = kWave >> 1; s > 0; s >>= 1)
                    if (L.prefix[row + s] <= t) row += s;
                const int nrr = L.nr[row];
                int k = L.lo[row] + (t - L.prefix[row]);
                if (k >= nrr) k -= nrr;
                const double phi_pix = ((double)k + L.shift[row]) * (kTwoPi / (double)nrr);
                process_pair<MODE, ACC>(m, r, L.start[row] + k, L.z[row], L.sth[row], phi_pix, out);
            }
        }
        __builtin_amdgcn_wave_barrier();
    }

    if (fallback4 && npairs < 4 && (r.rlast - r.rfirst < kWave)) {      // HealpixRunner.py:309-310
        npairs = 4;
        if (MODE != MODE_COUNT && lane < 4) {
            int64_t pix[4]; double wdummy[4];
            get_interpol<false>(h, r.theta, r.phi, pix, wdummy);
            const int64_t p = pix[lane];
            double z, sth, phi_pix;
            pix2loc(h, p, z, sth, phi_pix);
            process_pair<MODE, ACC>(m, r, p, z, sth, phi_pix, out);
        }
    }
    if (counts && lane == 0) counts[j] = npairs;
}

// ---------------------------------------------------------------------------------- K2
template <typename ACC>
__global__ void __launch_bounds__(256)
regrid_kernel(Hpx h, const double *__restrict__ map_in, const ACC *__restrict__ offsets,
              double *__restrict__ map_out)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= h.npix) return;
    const double val = map_in[p];
    if (!(val > 0.0)) return;                                            // :335
    double z, sth, phi;
    pix2loc(h, p, z, sth, phi);
    double s, c;
    sincos(phi, &s, &c);
    const double nx = sth * c + (double)offsets[3 * p + 0];              // :333
    const double ny = sth * s + (double)offsets[3 * p + 1];
    const double nz = z + (double)offsets[3 * p + 2];
    // hp.vec2ang(lonlat=True) (:334) then get_interp_weights(lonlat=True) (:337)
    const double dnorm = sqrt(nx * nx + ny * ny + nz * nz);
    const double theta = acos(nz / dnorm);
    double ph = atan2(ny, nx);
    if (ph < 0) ph += kTwoPi;
    const double lon = ph * kRad2Deg, lat = 90.0 - theta * kRad2Deg;
    const double th2 = kHalfPi - lat * kDeg2Rad, ph2 = lon * kDeg2Rad;
    int64_t cp[4]; double w[4];
    get_interpol<true>(h, th2, ph2, cp, w);
#pragma unroll
    for (int k = 0; k < 4; ++k) atomicAdd(map_out + cp[k], w[k] * val);  // :64
}

// sums[0] += sum(a), sums[1] += sum(b)
__global__ void __launch_bounds__(256)
sum2_kernel(int64_t n, const double *__restrict__ a, const double *__restrict__ b, double *__restrict__ sums)
{
    __shared__ double sa[256 / kWave], sb[256 / kWave];
    double xa = 0.0, xb = 0.0;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        xa += a[i]; xb += b[i];
    }
#pragma unroll
    for (int s = kWave >> 1; s > 0; s >>= 1) { xa += __shfl_down(xa, s, kWave); xb += __shfl_down(xb, s, kWave); }
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) { sa[wid] = xa; sb[wid] = xb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) { ta += sa[w]; tb += sb[w]; }
        atomicAdd(sums + 0, ta);
        atomicAdd(sums + 1, tb);
    }
}

}  // namespace bfgx
