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
#ifndef BFGX_PAIR32
#define BFGX_PAIR32 1      // 1: fp32 pair tail when pix_offsets are stored as fp32 (pair_value_fast32)
#endif
#ifndef BFGX_ABLATE
#define BFGX_ABLATE 0      // >0: timing-only ablation builds (scripts/ablate.sh); never shipped
#endif
#ifndef BFGX_ABL0
#define BFGX_ABL0 0        // K0 ablations (timing only): 1 no record stores, 2 no tile binning, 3 neither, 16 no list stores, 32 no census of small discs
#endif
#include "bfgx_cosmo.hpp"
#include "bfgx_math.hpp"

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


// One ring row of a disc (healpix_cxx query_disc_internal, fact = 0): pixels (lo + t) mod nr, t in [0, cnt)
struct RowSpan {
    int64_t start, nr;
    int32_t lo, cnt;
    bool shifted;
    double z, sth;
};

// the pixel span [lo, lo + cnt) (mod nr) of a disc in one ring whose geometry is already known (tiled kernels: per-tile
// ring table); same arithmetic as disc_row
__device__ inline void disc_row_span(int nr, bool shifted, double z, double fnr, int ring, double z0, double xa, double cosr,
                                     double phi0, int irmin, int irmax, int &lo, int &cnt)
{
    lo = 0; cnt = 0;
    if (ring < irmin || ring > irmax) { cnt = nr; return; }              // polar-cap row: whole ring inside
    const double x = (cosr - z * z0) * xa;
    const double ysq = 1.0 - z * z - x * x;
    if (!(ysq > 0.0)) return;
    const double dphi = atan2(sqrt(ysq), x);
    if (!(dphi > 0.0)) return;
    const double sh = shifted ? 0.5 : 0.0;
    const int64_t ip_lo = (int64_t)floor(fnr * (phi0 - dphi) - sh) + 1;
    const int64_t ip_hi = (int64_t)floor(fnr * (phi0 + dphi) - sh);
    int64_t c = ip_hi - ip_lo + 1;
    c = c < 0 ? 0 : (c > nr ? nr : c);
    cnt = (int)c;
    int64_t l = ip_lo;
    if (l < 0) l += nr;
    if (l >= nr) l -= nr;
    if (l < 0) l += nr;
    lo = (int)l;
}

__device__ inline void disc_row(const Hpx &h, int ring, double z0, double xa, double cosr, double phi0,
                                int irmin, int irmax, RowSpan &s)
{
    ring_info_small(h, ring, s.start, s.nr, s.shifted);
    ring_z_sth(h, ring, s.z, s.sth);
    s.lo = 0; s.cnt = 0;
    if (ring < irmin || ring > irmax) { s.cnt = (int)s.nr; return; }     // polar-cap row: whole ring inside
    const double x = (cosr - s.z * z0) * xa;
    const double ysq = 1.0 - s.z * s.z - x * x;
    if (!(ysq > 0.0)) return;
    const double dphi = atan2(sqrt(ysq), x);
    if (!(dphi > 0.0)) return;
    const double sh = s.shifted ? 0.5 : 0.0;
    const double fnr = (double)s.nr * kInvTwoPi;
    const int64_t ip_lo = (int64_t)floor(fnr * (phi0 - dphi) - sh) + 1;
    const int64_t ip_hi = (int64_t)floor(fnr * (phi0 + dphi) - sh);
    int64_t c = ip_hi - ip_lo + 1;
    c = c < 0 ? 0 : (c > s.nr ? s.nr : c);
    s.cnt = (int)c;
    int64_t l = ip_lo;                       // ip_lo is in (-nr, 2 nr): fold without a 64-bit modulo
    if (l < 0) l += s.nr;
    if (l >= s.nr) l -= s.nr;
    if (l < 0) l += s.nr;
    s.lo = (int)l;
}

// pixels-only variant of get_interpol: the 4 neighbours as (ring, index in ring)
__device__ inline void interp_neighbours(const Hpx &h, double theta, double phi, int32_t ring[4], int32_t k[4])
{
    int64_t pix[4]; double w[4];
    get_interpol<false>(h, theta, phi, pix, w);
    const double z = cos(theta);
    const int64_t ir1 = ring_above(h, z), ir2 = ir1 + 1;
    for (int q = 0; q < 4; ++q) {
        int64_t rg = (q < 2) ? ir1 : ir2;
        if (rg < 1) rg = 1;                                  // north-pole case: polar pixels 0..3 (ring 1)
        if (rg > 4 * h.nside - 1) rg = 4 * h.nside - 1;      // south-pole case: last ring
        int64_t st, nr; bool sh;
        ring_info_small(h, rg, st, nr, sh);
        ring[q] = (int32_t)rg;
        k[q] = (int32_t)(pix[q] - st);
    }
}

// ---------------------------------------------------------------------------------- device-side model
struct DevTable {
    int32_t ndim;
    int32_t n[BFGX_MAX_DIM];
    const double *axis[BFGX_MAX_DIM];     // device pointers
    const double *values;                 // device pointer
    const float *values32;                // fp32 copy of the values (same layout) for the mixed-precision pair path
    int32_t rdelta, logv;
    double eps_model;
    int32_t r_uniform;                    // ln r axis is uniform: index guess = (x - r0) * inv_dr
    double r0, inv_dr, r1;                // first / last node of the ln r axis
};

// what the per-pair read-out needs of the table (kept small: it travels in scalar registers)
struct PairTable {
    const double *values, *raxis;         // table values (r innermost) and the ln r axis
    const float *values32;
    double r0, inv_dr, r1;
    int32_t nr, r_uniform, rdelta, _pad;
};

__host__ __device__ inline PairTable make_pair_table(const DevTable &t)
{
    PairTable p;
    p.values = t.values; p.raxis = t.axis[2]; p.values32 = t.values32;
    p.r0 = t.r0; p.inv_dr = t.inv_dr; p.r1 = t.r1;
    p.nr = t.n[2]; p.r_uniform = t.r_uniform; p.rdelta = t.rdelta; p._pad = 0;
    return p;
}

struct DevModel {
    DevTable tab;
    Background bg_runner, bg_model;
    bfgx_massdef md_runner, md_model;
    double eps_runner;
    int32_t same_model;                   // model cosmology / mass definition identical to the runner's
    const double *da_coef;                // device, [kDaKnots-1][4]
    double da_step;
};

// corner rows of a table with extra parameter axes (p_keys): 4 * 2^K rows per halo.  The shell path takes K <= BFGX_MAX_EXTRA = 4 (64 rows: the
// reference's RegularGridInterpolator takes any number, Tabulate.py:524-561); the regular-grid and snapshot records keep K <= 2 (kNCmax)
constexpr int kNCmax = 16;
constexpr int kNCmaxShell = 4 << BFGX_MAX_EXTRA;
struct RowSetX {
    double w[kNCmaxShell];
    int32_t rowoff[kNCmaxShell];
};
// the per-halo values of the table's extra axes (the catalog's property columns, HealpixRunner.py:298)
struct ExtraCols { const double *p[BFGX_MAX_EXTRA]; };

// per-halo record written by K0, read (wave-uniformly) by K1/K3
struct alignas(16) HaloRec {
    double z0, xa, s0, phi0;              // query_disc pointing: cos/sin colatitude, azimuth in [0, 2pi)
    double cph0, sph0;                    // cos/sin(phi0)
    double cosr;                          // cos(disc radius)
    double theta, phi;                    // lonlat2thetaphi(ra, dec), for the <4-pixel fallback
    double D, a, inv_a, rcut, lnoff;      // lnoff = ln(1/a) [- ln R_model when Rdelta_sampling]
    double w[kNC];                        // (z,M) corner weights in scipy corner order
    int32_t rowoff[kNC];                  // element offset of each corner's radial row in values[]
    int32_t irmin, irmax, rfirst, rlast;  // phi-tested ring range, full row range (incl. polar caps)
    int32_t oob;                          // 1: (z, M[, params]) outside the table -> NaN read-out
    int32_t allphi;                       // disc spans all azimuths (pole inside / touching)
    int32_t fb;                           // 1: fewer than 4 pixels in the disc -> 4-neighbour fallback (:309-310)
    int32_t _pad;
    double flo, fhi;                      // azimuth range of the disc as fractions of 2 pi (unwrapped)
    int32_t fb_ring[4], fb_k[4];          // the 4 fallback pixels as (ring, index in ring)
};

__device__ inline double dev_E2(const Background &b, double a)
{
    const double a3 = a * a * a;
    const double de = (b.w0 == -1.0) ? b.Omega_l : b.Omega_l * pow(a, -3.0 * (1.0 + b.w0));   // LCDM: no pow()
    return b.Omega_m / a3 + de + b.Omega_r / (a3 * a);
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

// (z, M[, p0, p1]) corner rows of the table for one halo: weights in scipy's itertools.product order and the offsets
// of the corners' radial rows in the device layout [z][M][p0][p1][r] (r innermost).  Returns true when a coordinate is
// outside its axis (RegularGridInterpolator fill_value = nan).
template <int NC>
__device__ inline bool table_corners(const DevTable &tab, const double *gz, const double *gm, double x0, double x1, const double *xe,
                                     double *wv, int32_t *ro)
{
    // gz / gm: the ln(1 + z) and ln M axes (K0 stages them in LDS: the binary searches are chains of dependent loads); xe[K]: the halo's
    // values on the K extra axes
    const int iz = axis_find(gz, tab.n[0], x0);
    const int im = axis_find(gm, tab.n[1], x1);
    constexpr int K = (NC == 4) ? 0 : (NC == 8 ? 1 : (NC == 16 ? 2 : (NC == 32 ? 3 : 4)));
    static_assert(NC == (4 << K), "NC = 4 * 2^K");
    int ip[K > 0 ? K : 1];
    double tp[K > 0 ? K : 1];
    bool oob = (iz < 0 || im < 0);
#pragma unroll
    for (int k = 0; k < K; ++k) {
        const double *g = tab.axis[3 + k];
        ip[k] = axis_find(g, tab.n[3 + k], xe[k]);
        tp[k] = 0.0;
        oob = oob || ip[k] < 0;
        if (ip[k] >= 0) tp[k] = (xe[k] - g[ip[k]]) / (g[ip[k] + 1] - g[ip[k]]);
    }
    if (!oob) {
        const double tz = (x0 - gz[iz]) / (gz[iz + 1] - gz[iz]);
        const double tm = (x1 - gm[im]) / (gm[im + 1] - gm[im]);
        const int nr = tab.n[2];
        for (int c = 0; c < NC; ++c) {
            // corner c in scipy's itertools.product order over (z, M, p0, .., p_{K-1}): bit K + 1 = z, bit K = M, bit K - 1 - k = axis k
            const int bz = (c >> (K + 1)) & 1, bm = (c >> K) & 1;
            double w = (1.0 * (bz ? tz : 1.0 - tz)) * (bm ? tm : 1.0 - tm);
            int idx = (iz + bz) * tab.n[1] + im + bm;
#pragma unroll
            for (int k = 0; k < K; ++k) {
                const int b = (c >> (K - 1 - k)) & 1;
                w *= (b ? tp[k] : 1.0 - tp[k]);
                idx = idx * tab.n[3 + k] + ip[k] + b;
            }
            wv[c] = w;
            ro[c] = idx * nr;
        }
    } else {
        for (int c = 0; c < NC; ++c) { wv[c] = 0.0; ro[c] = 0; }
    }
    return oob;
}

// ---------------------------------------------------------------------------------- tiling
// The sphere is cut into tiles = (band of BR consecutive rings) x (1/nphi of the azimuth).  In ring i
// with nr pixels, tile j of the ring's band owns k in [ceil(j nr/nphi), ceil((j+1) nr/nphi)), i.e. the
// pixels with floor(k nphi / nr) == j; at most W pixels wide.  One workgroup owns one tile's accumulators
// in LDS, so the per-halo scatter needs no global atomics and every output pixel is stored exactly once.
struct Tiling {
    int32_t BR, W, nbands, ntiles;
    const int32_t *band_tile0;            // [nbands + 1] first tile of each band
    const int32_t *band_nphi;             // [nbands]
    const int32_t *band_nrmin;            // [nbands] shortest ring of the band
    const int32_t *tile_band;             // [ntiles]
    const int32_t *tile_order;            // [ntiles] launch order of the scatter kernel: largest tiles (in pixels) first
};

// first ring-local pixel of azimuth slice j: ceil(j nr / nphi); j nr < 2^31 for nside <= 8192
__device__ inline int tile_ks(int j, int nr, int nphi) { return (int)(((unsigned)j * (unsigned)nr + (unsigned)nphi - 1u) / (unsigned)nphi); }

__device__ inline int ring_len(const Hpx &h, int ring)
{
    const int64_t n = h.nside;
    return (int)(4 * (ring < n ? ring : (ring > 3 * n ? 4 * n - ring : n)));
}

// where K0 already reserved this halo's slots in its tiles' entry lists (n <= kRefMax), or n > kRefMax: the
// placement pass enumerates the tiles again (from the span kept in the same bytes) and takes slots by cursor
constexpr int kRefMax = 4;
struct TileRef {
    union {
        struct { int32_t tile[kRefMax], slot[kRefMax]; } few;                       // n <= kRefMax
        struct { int32_t rfirst, rlast, allphi, _p; double flo, fhi; } many;        // n >  kRefMax (never a fallback halo)
    };
    int32_t n;
    int32_t cls;                           // kClsNone / kClsNarrow / kClsWide
    int32_t _pad[2];
};

// halo classes of the tiled scatter: narrow discs (|azimuth difference| <= 0.45 for every pixel, no pole inside) go to the
// fast kernel (bfgx_scatter2.hpp), everything else (polar caps, very low z) to the generic tile kernel below
enum { kClsNone = 0, kClsNarrow = 1, kClsWide = 2 };

// what the tile enumeration needs of a disc
struct DiscSpan {
    int32_t fb, rfirst, rlast, allphi;
    int32_t fb_ring[4], fb_k[4];
    double flo, fhi;
};

// calls f(tile) once for every tile that may hold pixels of the halo's disc (conservative superset)
template <typename F>
__device__ inline void for_each_tile(const Hpx &h, const Tiling &T, const DiscSpan &r, F &&f)
{
    if (r.fb) {
        int seen[4], ns = 0;
        for (int q = 0; q < 4; ++q) {
            const int b = (r.fb_ring[q] - 1) / T.BR;
            const int nphi = T.band_nphi[b];
            const int t = T.band_tile0[b] + (int)(((int64_t)r.fb_k[q] * nphi) / ring_len(h, r.fb_ring[q]));
            bool dup = false;
            for (int i = 0; i < ns; ++i) dup |= (seen[i] == t);
            if (!dup) { seen[ns++] = t; f(t); }
        }
        return;
    }
    if (r.rlast < r.rfirst) return;
    const int b0 = (r.rfirst - 1) / T.BR, b1 = (r.rlast - 1) / T.BR;
    for (int b = b0; b <= b1; ++b) {
        const int nphi = T.band_nphi[b], t0 = T.band_tile0[b];
        bool all = r.allphi || nphi == 1;
        int jlo = 0, jhi = nphi - 1;
        if (!all) {
            const double margin = 2.0 * (double)nphi / (double)T.band_nrmin[b] + 1e-6;
            jlo = (int)floor((double)nphi * r.flo - margin);
            jhi = (int)floor((double)nphi * r.fhi + margin);
            if (jhi - jlo + 1 >= nphi) { all = true; jlo = 0; jhi = nphi - 1; }
        }
        for (int jj = jlo; jj <= jhi; ++jj) {
            int j = jj % nphi;
            if (j < 0) j += nphi;
            f(t0 + j);
        }
    }
}

// ---------------------------------------------------------------------------------- records of the fast tiled scatter
// (bfgx_scatter2.hpp).  K0 writes them for narrow halos; the 272-byte HaloRec is only written for wide halos.
struct alignas(16) RowRec {                // what the ring-row phase needs of a halo
    double z0, s0, xa, cosr, phi0;         // query_disc pointing: cos / sin colatitude, 1 / sin, cos(radius), azimuth
    int32_t rfirst, rlast;                 // ring range of the disc (no pole inside: every row is phi-tested)
    int32_t fb;                            // bit 0: the rows are the 4 fallback pixels of FbRec (HealpixRunner.py:309-310); bit 1: a WIDE disc (a pole
                                           // inside, or pixels further than 0.40 rad from the halo's azimuth): general row spans, FbRec.ring[0..1] = irmin, irmax
    int32_t _pad;
    double cut2;                           // (rcut a / D)^2 in fp64, for the rare ambiguous fp32 cut decisions
};
struct FbRec { int32_t ring[4], k[4]; };   // the 4 get_interp_weights neighbours as (ring, index in ring); a wide disc without them: ring[0..1] = irmin, irmax

template <typename real>
struct alignas(16) PairRecT {              // what the pair phase needs of a halo, in the precision of the pair math
    real scale2;                           // (D / a [/ R_model])^2: ln r axis coordinate = ln(|u| D / a) = ln(|u|^2 scale2) / 2,  u = diff / D
    real cut2;                             // 1 / (rcut a / D)^2, or 0 when the disc itself implies r < eps R
    real aD;                               // a / D: offset / D = d aD u / |u|
    real cph0, sph0;                       // rotation back by +phi0
    real w[4];                             // (z, M) corner weights
    int32_t cell;                          // element offset of the (z, M) cell's block in the interleaved table
    int32_t oob;                           // (z, M) outside the table -> NaN read-out
    int32_t hidx, _pad;
};

// interleaved copy of a 3-axis table for the fast kernel: for every (z, M) cell and radial interval i the 8 numbers
// {A_c, B_c} (c = 2 bz + bm),  A_c = T[iz + bz][im + bm][i],  B_c = T[..][i + 1] - A_c, so that one pair reads 2 x 16 B (f32)
template <typename real>
struct Tab8T {
    const real *v;                         // [(nz - 1)(nm - 1)][nr - 1][8]
    real r0, r1, inv_dr;                   // uniform ln r axis
    int32_t nr, _pad;
};

// One slot per active lane in the list of its tile, with ONE returning atomic per run of consecutive lanes that name the same tile (the run's
// first lane adds the run's length, the others take base + their place in the run).  A catalog in random order has runs of one lane, and the
// cost is a dozen instructions; a catalog written patch by patch -- lightcone catalogs usually are -- has runs of tens of lanes, and without
// this its halos add to ONE counter back to back: same-address atomics serialise at ~5 ns each (K0 0.13 -> 0.62 ms measured on a catalog in
// (band, azimuth) order).  The counters of region A are cnt_pad words apart (PrepOut) for the same reason: atomics to one 64-byte LINE serialise too, and
// a patch-ordered catalog keeps the ~1000 waves that run together on the ~25 lines of neighbouring tiles (0.25 ms with runs alone).  Every lane of the wave must call (shuffles); lanes with active == false get 0.
// (in two halves, so that the atomic's latency overlaps whatever the caller computes in between: wave_run_issue returns what the run's first
// lane got from the atomic -- not yet valid in the other lanes --, wave_run_resolve, called by every lane again, hands it round)
constexpr int kCntPadMax = 32;        // words between the region-A counters of two tiles in the padded array: a 128-byte line each while the
                                      // array stays below 1 MB (6208 tiles at NSIDE 1024), fewer for finer shells (bfgx_plan.cnt_pad)
struct RunSlot { int base, hr; };       // hr: first lane of the run | (place in the run + 1) << 8, 0 in an inactive lane
__device__ inline RunSlot wave_run_issue(int32_t *cnt, int tile, bool active)
{
    RunSlot r;
    r.base = 0; r.hr = 0;
    if (__ballot(active) == 0ull) return r;                            // (wave-uniform)
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int key = active ? tile : -1 - lane;                         // (an inactive lane is a run of its own, and does nothing)
    const int prev = __shfl_up(key, 1, kWave);
    const bool head = (lane == 0) || (key != prev);
    const unsigned long long hm = __ballot(head);
    const int hl = 63 - __clzll((long long)(hm & ((2ull << lane) - 1ull)));          // first lane of this lane's run
    const unsigned long long up = (lane == kWave - 1) ? 0ull : (hm >> (lane + 1));
    const int next = up ? lane + __ffsll((long long)up) : kWave;                    // first lane of the next run
    r.hr = active ? (hl | ((lane - hl + 1) << 8)) : 0;
    if (head && active) r.base = atomicAdd(cnt + tile, next - lane);
    return r;
}
__device__ inline int wave_run_resolve(const RunSlot &r)
{
    const int b = __shfl(r.base, r.hr & 63, kWave);
    return r.hr ? b + (r.hr >> 8) - 1 : 0;
}

// ---------------------------------------------------------------------------------- K0
// where K0 writes (device pointers; nullptr = not wanted)
struct PrepOut {
    HaloRec *rec;                 // wide halos (or every halo when rec_all)
    RowSetX *rowsx;               // NC > 4
    RowRec *rowrec;               // fast path records (fast != 0)
    void *pairrec;                // PairRecT<real> *
    FbRec *fbrec;
    TileRef *tref;                // tile binning (nullptr: no binning)
    int32_t *cnt_a, *cnt_b, *cnt_w;
    // direct placement of the narrow halos that touch <= kRefMax tiles (almost all of them): their slot in a tile's FIXED-CAPACITY list
    // entries_a[tile][cap_a] is the value the counting atomic returns, so K0 stores the halo index there itself -- no TileRef, no placement
    // pass.  Only the others (wide discs, discs over more than kRefMax tiles, slots beyond cap_a) are listed per K0 workgroup
    // (slow_list[block][256], slow_cnt[block]) for tile_place_kernel, which draws their slots from cursors after the scan.
    int32_t *entries_a, *slow_list, *slow_cnt;
    float *work_est;              // per K0 workgroup: the sum of its narrow halos' estimated pixels (pi radius^2 / pixel area): what the fast
                                  // kernel's form is chosen by when the halo count alone does not say (nullptr: not wanted)
    int32_t cap_a;
    int32_t cnt_pad;              // words between two tiles' counters in cnt_a when entries_a is set (direct placement)
    int32_t fast;                 // 1: narrow halos are class kClsNarrow (fast kernel), 0: every halo is kClsWide (generic kernel), 2: the fast kernel
                                  // takes the wide discs too (RowRec.fb bit 1): no halo is left to the generic kernel's wide pass
    int32_t rec_all;              // 1: HaloRec for every halo (halo-centric algo 0)
    int32_t ncell_m, nrm1;        // (nm - 1), (nr - 1) of the interleaved table
    // a BLOCKED catalog (the rows an all_to_all has just delivered, [block][column][blk_rows]): halo j reads its columns at
    // (j / blk_rows) * blk_stride + (j % blk_rows) from the column pointers of block 0 (blk_rows = 0: plain columns)
    int64_t blk_rows, blk_stride;
};

// thread per halo.  NC = 4 * 2^K corner rows (K extra parameter axes): for NC > 4 the rows go to rowsx[j] instead of the
// record.  `real` = precision of the fast kernel's pair records.  lnz1 / lnM (optional): ln(1 + z), ln M computed by the
// caller (numpy on the host), so that halos on a table edge are classified exactly as the reference does (README.md:78-80).
template <int NC, typename real>
__device__ __forceinline__ void halo_prep_one(const DevModel &m, const Hpx &h, int64_t j, bool live, const double *gz, const double *gm,
                                              const double *__restrict__ M, const double *__restrict__ z,
                                              const double *__restrict__ ra, const double *__restrict__ dec,
                                              ExtraCols ex,
                                              const double *__restrict__ lnz1, const double *__restrict__ lnM,
                                              int fallback4, const Tiling &T, const PrepOut &o, int *s_nslow, float *s_work);

// workgroups of 256 per CU the kernel is compiled for (register budget 512 / this per lane).  Measured, config 2 on the S19 table (K0 in ms):
// 3: 0.133, 4: 0.136, 5: 0.164, 6: 0.219, 8: 0.269 -- more waves only buy spills: the kernel is bound by the instructions it issues (2560 per
// wave), not by the latency of its atomics
#ifndef BFGX_K0_OCC
#define BFGX_K0_OCC 4
#endif
template <int NC, typename real>
__global__ void __launch_bounds__(256, BFGX_K0_OCC)
halo_prep_kernel(DevModel m, Hpx h, int64_t nhalo,
                 const double *__restrict__ M, const double *__restrict__ z,
                 const double *__restrict__ ra, const double *__restrict__ dec,
                 ExtraCols ex,
                 const double *__restrict__ lnz1, const double *__restrict__ lnM,
                 int fallback4, Tiling T, PrepOut o)
{
    __shared__ int s_nslow;                               // halos of this workgroup left to the placement pass
    __shared__ float s_work;                              // estimated pixels of its narrow halos
    if (threadIdx.x == 0) { s_nslow = 0; s_work = 0.0f; }
    // the (z, M) axes of the table go to LDS when they are short (the usual 10 - 30 nodes)
    constexpr int kAxisLds = 128;
    __shared__ double ax_lds[2 * kAxisLds];
    const bool ax_in_lds = m.tab.n[0] <= kAxisLds && m.tab.n[1] <= kAxisLds;
    if (ax_in_lds) {
        for (int i = threadIdx.x; i < m.tab.n[0]; i += blockDim.x) ax_lds[i] = m.tab.axis[0][i];
        for (int i = threadIdx.x; i < m.tab.n[1]; i += blockDim.x) ax_lds[kAxisLds + i] = m.tab.axis[1][i];
        __syncthreads();
    }
    const double *gz = ax_in_lds ? ax_lds : m.tab.axis[0], *gm = ax_in_lds ? ax_lds + kAxisLds : m.tab.axis[1];
    // the per-band tables of the tiling (first tile, azimuth slices, shortest ring) go to LDS too: the tile enumeration of
    // every halo reads them through chains of dependent loads
    constexpr int kBandLds = 1024;
    __shared__ int32_t band_lds[3 * kBandLds + 1];
    if (o.tref && T.nbands <= kBandLds) {
        for (int i = threadIdx.x; i <= T.nbands; i += blockDim.x) band_lds[i] = T.band_tile0[i];
        for (int i = threadIdx.x; i < T.nbands; i += blockDim.x) { band_lds[kBandLds + 1 + i] = T.band_nphi[i]; band_lds[2 * kBandLds + 1 + i] = T.band_nrmin[i]; }
        __syncthreads();
        T.band_tile0 = band_lds; T.band_nphi = band_lds + kBandLds + 1; T.band_nrmin = band_lds + 2 * kBandLds + 1;
    }
    __syncthreads();                                       // (s_nslow)
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    // (every lane runs it -- the binning shuffles across the wave --, the lanes beyond the catalog on its last halo and without any store)
    halo_prep_one<NC, real>(m, h, j < nhalo ? j : nhalo - 1, j < nhalo, gz, gm, M, z, ra, dec, ex, lnz1, lnM, fallback4, T, o, &s_nslow, &s_work);
    if (o.slow_cnt) {
        __syncthreads();
        if (threadIdx.x == 0) o.slow_cnt[blockIdx.x] = s_nslow;
        if (threadIdx.x == 0 && o.work_est) o.work_est[blockIdx.x] = s_work;
    }
}

template <int NC, typename real>
__device__ __forceinline__ void halo_prep_one(const DevModel &m, const Hpx &h, int64_t j, bool live, const double *gz, const double *gm,
                                              const double *__restrict__ M, const double *__restrict__ z,
                                              const double *__restrict__ ra, const double *__restrict__ dec,
                                              ExtraCols ex,
                                              const double *__restrict__ lnz1, const double *__restrict__ lnM,
                                              int fallback4, const Tiling &T, const PrepOut &o, int *s_nslow, float *s_work)
{
    HaloRec r;
    const int64_t jin = o.blk_rows > 0 ? (j / o.blk_rows) * o.blk_stride + (j % o.blk_rows) : j;      // where this halo's input columns are
    const double M_j = M[jin], z_j = z[jin];
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

    // hp.ang2vec(ra, dec, lonlat=True) (:303) followed by query_disc's pointing(vec) (:306): the pointing of the unit
    // vector (sin t cos p, sin t sin p, cos t) is (t, p mod 2 pi) again, so the two atan2 of the round trip are skipped
    // (they return theta and phi to within an ulp)
    const double theta = kHalfPi - dec[jin] * kDeg2Rad;
    const double phi = ra[jin] * kDeg2Rad;
    double st, ct, sp, cp;
    sincos_bounded(theta, st, ct);                                        // libm-free (bfgx_math.hpp): K0 is latency-bound
    sincos_bounded(phi, sp, cp);
    const double thq = theta;
    double phq = phi - kTwoPi * floor(phi * kInvTwoPi);
    if (phq >= kTwoPi) phq -= kTwoPi;
    if (st == 0.0) phq = 0.0;
    r.z0 = ct;
    r.s0 = st;
    r.xa = 1.0 / sqrt((1.0 - r.z0) * (1.0 + r.z0));
    r.phi0 = phq;
    r.cph0 = cp; r.sph0 = sp;
    r.theta = theta; r.phi = phi;
    r.D = D; r.a = a;

    const int64_t nl4 = 4 * h.nside;
    bool pole = false;
    double sr_ = 0.0;
    // a halo with a non-finite / non-positive disc radius or position can touch no pixel: give it an empty row range
    // (this also keeps every ring index below in range whatever the input columns hold)
    const bool bad = !live || !(radius > 0.0) || !isfinite(radius) || !(theta >= 0.0) || !(theta <= kPi) || !isfinite(phi);
    if (bad) {
        r.cosr = 1.0; r.irmin = 1; r.irmax = 0; r.rfirst = 1; r.rlast = 0;
        r.z0 = 1.0; r.s0 = 0.0; r.xa = 1.0; r.phi0 = 0.0; r.cph0 = 1.0; r.sph0 = 0.0;
    } else if (radius >= kPi) {
        r.cosr = -1.0; r.irmin = (int32_t)nl4; r.irmax = 0; r.rfirst = 1; r.rlast = (int32_t)(nl4 - 1);
        pole = true;
    } else {
        double cr_;
        sincos_bounded(radius, sr_, cr_);
        r.cosr = cr_;
        const double rlat1 = thq - radius;
        const double rlat2 = thq + radius;
        // cos(theta -+ radius) by the addition theorem (no further libm calls)
        int64_t irmin = ring_above(h, ct * cr_ + st * sr_) + 1;
        int64_t irmax = ring_above(h, ct * cr_ - st * sr_);
        if (irmax > nl4 - 1) irmax = nl4 - 1;
        r.irmin = (int32_t)irmin; r.irmax = (int32_t)irmax;
        r.rfirst = (int32_t)(((rlat1 <= 0) && (irmin > 1)) ? 1 : irmin);
        r.rlast = (int32_t)(((rlat2 >= kPi) && (irmax + 1 < nl4)) ? nl4 - 1 : irmax);
        pole = (rlat1 <= 0) || (rlat2 >= kPi);
    }
    // azimuthal extent of the disc (for tile binning): half-width asin(sin r / sin theta0) when no pole inside
    double dmax;
    {
        const double sr = (radius >= kHalfPi) ? 1.0 : sr_;
        r.allphi = (pole || radius >= kHalfPi || !(sr < 0.999 * r.s0)) ? 1 : 0;
        // (needed to ~1e-6 only: tile binning adds margins, the class threshold has slack) series below 0.35, libm above
        const double xs = sr / r.s0, xs2 = xs * xs;
        const double as_small = xs * (1.0 + xs2 * (1.0 / 6.0 + xs2 * (3.0 / 40.0 + xs2 * (15.0 / 336.0 + xs2 * (105.0 / 3456.0)))));
        dmax = r.allphi ? kPi : (xs < 0.35 ? as_small * (1.0 + 2e-6) : asin(xs));
        r.flo = (phq - dmax) * kInvTwoPi;
        r.fhi = (phq + dmax) * kInvTwoPi;
    }

    // <4-pixel fallback (HealpixRunner.py:309-310): only discs of a few pixels can qualify -> exact census
    r.fb = 0; r._pad = 0;
    for (int q = 0; q < 4; ++q) { r.fb_ring[q] = 0; r.fb_k[q] = 0; }
    if (fallback4 && !bad && (r.rlast - r.rfirst) < 8 && !(BFGX_ABL0 & 32)) {
        // (from the middle ring outwards -- mid, mid + 1, mid - 1, ... --: the longest rows come first, and a disc of 4 or more pixels is
        // recognised after two rows instead of three or four from the edge; the slowest lane of a wave sets the trip count)
        int total = 0;
        const int span = r.rlast - r.rfirst, mid = (r.rfirst + r.rlast) >> 1;
        for (int t = 0; t <= span && total < 4; ++t) {
            const int ring = mid + ((t & 1) ? ((t + 1) >> 1) : -(t >> 1));
            RowSpan s;
            disc_row(h, ring, r.z0, r.xa, r.cosr, r.phi0, r.irmin, r.irmax, s);
            total += s.cnt;
        }
        if (total < 4) {
            r.fb = 1;
            interp_neighbours(h, theta, phi, r.fb_ring, r.fb_k);
        }
    }

    // class: the fast kernel takes discs without a pole inside whose pixels all lie within 0.40 rad of the halo's azimuth
    // (fallback pixels: within one pixel of it, so rings of at least 64 pixels are narrow enough)
    int cls = kClsWide;
    bool geo_wide = false;                 // (o.fast == 2) a disc the fast kernel takes through its general row spans and full-range sin / cos
    if (bad) cls = kClsNone;
    else if (o.fast && NC == 4 && !r.allphi && dmax <= 0.40) {
        cls = kClsNarrow;
        if (r.fb) for (int q = 0; q < 4; ++q) if (r.fb_ring[q] < 16 || r.fb_ring[q] > (int)nl4 - 16) cls = kClsWide;
    }
    if (o.fast == 2 && NC == 4 && cls == kClsWide) { cls = kClsNarrow; geo_wide = true; }
    // tile binning, pass 1: reserve one slot per touched tile.  Issued HERE, before the model-side arithmetic and the record
    // stores, so that the returning atomics (the longest latency of this kernel) are in flight while the rest is computed;
    // the TileRef that holds the slots is stored last.
    if (o.work_est) {
        // pixels of the disc: pi radius^2 / (4 pi / npix); one add per wave (the waves of a workgroup share one LDS word)
        float w = (cls == kClsNarrow) ? (r.fb ? 4.0f : (float)(radius * radius) * 0.25f * (float)h.npix) : 0.0f;
#pragma unroll
        for (int sft = kWave >> 1; sft > 0; sft >>= 1) w += __shfl_xor(w, sft, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) atomicAdd(s_work, w);
    }
    TileRef ref;
    RunSlot run[kRefMax];
    for (int i = 0; i < kRefMax; ++i) { run[i].base = 0; run[i].hr = 0; }
    ref.n = 0; ref.cls = cls; ref._pad[0] = ref._pad[1] = 0;
    for (int i = 0; i < kRefMax; ++i) { ref.few.tile[i] = 0; ref.few.slot[i] = 0; }
    if (o.tref && !(BFGX_ABL0 & 2)) {
        DiscSpan ds;
        ds.fb = r.fb; ds.rfirst = r.rfirst; ds.rlast = r.rlast; ds.allphi = r.allphi; ds.flo = r.flo; ds.fhi = r.fhi;
        for (int q = 0; q < 4; ++q) { ds.fb_ring[q] = r.fb_ring[q]; ds.fb_k[q] = r.fb_k[q]; }
        int tl[kRefMax] = {0, 0, 0, 0};
        int nt = 0;
        if (cls != kClsNone) for_each_tile(h, T, ds, [&](int t) { if (nt < kRefMax) tl[nt] = t; ++nt; });
        ref.n = nt; ref.cls = cls; ref._pad[0] = ref._pad[1] = 0;
        // narrow halos over at most kRefMax tiles take their slots now (returning atomics, one per run of lanes that name the same tile; the
        // halo index is stored into the slot at the end of this function, when the atomic has long returned) -- every lane of the wave goes
        // through wave_run_slot; wide ones are placed by cursor after the scan
#pragma unroll
        for (int i = 0; i < kRefMax; ++i) {
            run[i] = wave_run_issue(o.cnt_a, o.entries_a ? tl[i] * o.cnt_pad : tl[i], nt <= kRefMax && i < nt && cls == kClsNarrow);
            if (nt <= kRefMax) { ref.few.tile[i] = (i < nt) ? tl[i] : 0; ref.few.slot[i] = 0; }
        }
        if (nt <= kRefMax) {
            for (int i = 0; i < kRefMax; ++i)
                if (i < nt && cls != kClsNarrow) atomicAdd(o.cnt_w + tl[i], 1);
        } else {
            ref.many.rfirst = r.rfirst; ref.many.rlast = r.rlast; ref.many.allphi = r.allphi; ref.many._p = 0;
            ref.many.flo = r.flo; ref.many.fhi = r.fhi;
            int32_t *cnt = (cls == kClsNarrow) ? o.cnt_b : o.cnt_w;
            for_each_tile(h, T, ds, [&](int t) { atomicAdd(cnt + t, 1); });
        }
    }
    // model-side radius and table coordinates (BaryonCorrection.py:364-370, Tabulate.py:279-283)
    const double Rmod = (m.same_model ? R : dev_radius(m.bg_model, m.md_model, M_j, a)) / a;
    r.rcut = m.tab.eps_model * Rmod;
    r.inv_a = 1.0 / a;
    const double x0 = lnz1 ? lnz1[jin] : fast_log(1.0 / a), x1 = lnM ? lnM[jin] : fast_log(M_j);
    r.lnoff = m.tab.rdelta ? (x0 - fast_log(Rmod)) : x0;
    double wv[NC];
    int32_t ro[NC];
    double xe[BFGX_MAX_EXTRA];
#pragma unroll
    for (int k = 0; k < BFGX_MAX_EXTRA; ++k) xe[k] = ((4 << k) < NC) ? ex.p[k][jin] : 0.0;          // (axis k exists iff NC >= 8 << k)
    const bool oob = table_corners<NC>(m.tab, gz, gm, x0, x1, xe, wv, ro);
    r.oob = oob ? 1 : 0;
    if (NC == 4) {
        for (int c = 0; c < 4; ++c) { r.w[c] = wv[c]; r.rowoff[c] = ro[c]; }
    } else {
        for (int c = 0; c < 4; ++c) { r.w[c] = 0.0; r.rowoff[c] = 0; }
        RowSetX rx;
        for (int c = 0; c < kNCmaxShell; ++c) { rx.w[c] = c < NC ? wv[c] : 0.0; rx.rowoff[c] = c < NC ? ro[c] : 0; }
        if (live) o.rowsx[j] = rx;
    }

    if (live && o.rec && (o.rec_all || cls == kClsWide)) o.rec[j] = r;
    if (cls == kClsNarrow && !(BFGX_ABL0 & 1)) {
        RowRec rr;
        rr.z0 = r.z0; rr.s0 = r.s0; rr.xa = r.xa; rr.cosr = r.cosr; rr.phi0 = r.phi0;
        rr.rfirst = r.rfirst; rr.rlast = r.rlast; rr.fb = (r.fb ? 1 : 0) | (geo_wide ? 2 : 0); rr._pad = 0;
        // r_sep / a < rcut  <=>  |u|^2 < (rcut a / D)^2,  u = diff / D.  Every pixel of the disc has a chord
        // D |u| <= 2 D sin(radius / 2); when that is below rcut a the cut can never fire (eps_model >= eps_runner)
        const double cut = r.rcut * a / D;
        rr.cut2 = cut * cut;
        double sh, chh;
        sincos_bounded(0.5 * fmin(radius, kPi), sh, chh);      // (the longest chord of a disc: 2 sin(radius / 2), 2 from radius = pi)
        const bool implied = (m.tab.logv != 0) || (!r.fb && cut >= 2.0 * sh * (1.0 + 1e-9));     // paint: no model-side cut at all
        o.rowrec[j] = rr;
        PairRecT<real> pr;
        // a halo outside the (z, M) table reads NaN (RGI fill_value): its ln r offset is pushed beyond any table, so that the pair
        // phase's range test drops it without a flag of its own
        // the scale goes INTO the logarithm's argument: ln|u| ~ -7 and ln(D / a) ~ +7 added afterwards cancel to an O(1) number that carries
        // the absolute rounding of both (7e-7 in fp32); ln(|u|^2 scale2) / 2 carries half an ulp of the O(1) result
        double scl = D * r.inv_a;
        if (m.tab.rdelta) scl /= Rmod;
        pr.scale2 = oob ? (real)1.0e30 : (real)(scl * scl);
        pr.cut2 = implied ? (real)0 : (real)(1.0 / rr.cut2);       // 1 / cut^2 (see pair_eval); 0: the disc itself implies r < eps R
        pr.aD = (real)(a / D);
        pr.cph0 = (real)r.cph0; pr.sph0 = (real)r.sph0;
        for (int c = 0; c < 4; ++c) pr.w[c] = (real)wv[c < NC ? c : 0];
        // cell (iz, im) from the first corner's row offset: ro[0] = (iz nm + im) nr
        const int nr = o.nrm1 + 1, nm = o.ncell_m + 1;
        const int row0 = oob ? 0 : ro[0] / nr;
        const int iz = row0 / nm, im = row0 - iz * nm;
        pr.cell = (iz * o.ncell_m + im) * o.nrm1 * 8;
        pr.oob = oob ? 1 : 0;
        pr.hidx = (int32_t)j; pr._pad = 0;
        reinterpret_cast<PairRecT<real> *>(o.pairrec)[j] = pr;
        if (r.fb || geo_wide) {
            FbRec f;
            for (int q = 0; q < 4; ++q) { f.ring[q] = r.fb_ring[q]; f.k[q] = r.fb_k[q]; }
            if (!r.fb) { f.ring[0] = r.irmin; f.ring[1] = r.irmax; }
            o.fbrec[j] = f;
        }
    }
    if (BFGX_ABL0 & 1) { if (r.cosr == 1.2345 && wv[0] == 0.5) o.rowrec[j].z0 = r.lnoff + r.rcut; }      // keep the values alive
    if (o.tref && !(BFGX_ABL0 & 2)) {                          // (the slots were reserved above, see there)
#pragma unroll
        for (int i = 0; i < kRefMax; ++i) {
            const int sl = wave_run_resolve(run[i]);
            if (ref.n <= kRefMax && ref.cls == kClsNarrow && i < ref.n) ref.few.slot[i] = sl;
        }
        bool slow = (ref.cls != kClsNone) && (ref.n > kRefMax || ref.cls != kClsNarrow);
        if (ref.cls == kClsNarrow && ref.n <= kRefMax) {
            for (int i = 0; i < kRefMax; ++i) if (i < ref.n) {
                const int sl = ref.few.slot[i];
                if (o.entries_a && sl < o.cap_a) { if (!(BFGX_ABL0 & 16)) o.entries_a[(int64_t)ref.few.tile[i] * o.cap_a + sl] = (int32_t)j; }
                else if (o.entries_a) { ref.few.slot[i] = -1; atomicAdd(o.cnt_b + ref.few.tile[i], 1); slow = true; }      // the tile's fixed list is full: region B
                else slow = true;                                                                                         // (no direct placement: every halo is listed)
            }
        }
        if (live && (slow || !o.slow_list)) o.tref[j] = ref;
        if (slow && o.slow_list) o.slow_list[(int64_t)blockIdx.x * 256 + atomicAdd(s_nslow, 1)] = (int32_t)j;
    }
}

// Ring range [first, last] (1-based, inclusive; first > last: nothing) that a halo's disc -- or its 4 fallback pixels -- can touch,
// widened by 2 rings: what a rank that owns a range of bands needs to know to take the halos that matter to it (spatially
// sharded multi-GPU runs).  Same arithmetic as halo_prep_kernel for a, R, D_A and the colatitude band of the disc.
__global__ void __launch_bounds__(256)
disc_rings_kernel(DevModel m, Hpx h, int64_t nhalo, const double *__restrict__ M, const double *__restrict__ z,
                  const double *__restrict__ dec, int32_t *__restrict__ rings, int margin)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nhalo) return;
    const double M_j = M[j], z_j = z[j];
    const double a = 1.0 / (1.0 + z_j);
    const double R = dev_radius(m.bg_runner, m.md_runner, M_j, a);
    double D;
    {
        int i = (int)floor(z_j / m.da_step);
        i = max(0, min(i, kDaKnots - 2));
        const double t = z_j - (double)i * m.da_step;
        const double *c = m.da_coef + 4 * i;
        D = ((c[0] * t + c[1]) * t + c[2]) * t + c[3];
    }
    const double radius = R * m.eps_runner / D;
    const double theta = kHalfPi - dec[j] * kDeg2Rad;
    const int nl4 = (int)(4 * h.nside);
    int first = 1, last = 0;
    const bool bad = !(radius > 0.0) || !isfinite(radius) || !(theta >= 0.0) || !(theta <= kPi) || !(M_j > 0.0) || !isfinite(M_j) || !(z_j > -1.0);
    if (!bad) {
        if (radius >= kPi) { first = 1; last = nl4 - 1; }
        else {
            const double lo = fmax(theta - radius, 0.0), hi = fmin(theta + radius, kPi);
            first = (int)ring_above(h, cos(lo)) - 1;              // ring_above: last ring with colatitude <= the argument's
            last = (int)ring_above(h, cos(hi)) + 2;
            first = max(1, first - 1); last = min(nl4 - 1, last + 1);
        }
        // (a rank that also computes the bands next to its own -- the aprons of its regrid -- takes the halos of those too)
        first = max(1, first - margin); last = min(nl4 - 1, last + margin);
    }
    rings[2 * j] = first; rings[2 * j + 1] = last;
}

// Routing of scattered halos to the ranks whose ring bands their discs can touch (spatially sharded multi-GPU runs): a halo with
// ring range [first, last] goes to the contiguous run of ranks j with bounds[j] <= last and bounds[j + 1] > first.
constexpr int kRouteMaxRanks = 64, kRouteMaxCols = 8;
struct RouteArgs {
    int32_t world, ncols;
    int32_t bounds[kRouteMaxRanks + 1];          // first ring of every rank's run of bands, + one past the last ring
    int64_t start[kRouteMaxRanks];               // fill pass: first row of every destination in the send buffer
    const double *col[kRouteMaxCols];            // the catalog columns to pack into rows
    int64_t blockcap;                            // > 0: fixed-capacity blocks [world][ncols][blockcap] (column-major inside a block)
    int32_t *overflow;                           // blockcap > 0: set to 1 when a destination receives more than blockcap halos
};

__device__ inline void route_range(const RouteArgs &a, int first, int last, int &jlo, int &jhi)
{
    jlo = 0; jhi = -1;
    if (first > last) return;
    while (jlo < a.world - 1 && a.bounds[jlo + 1] <= first) ++jlo;
    jhi = jlo;
    while (jhi < a.world - 1 && a.bounds[jhi + 1] <= last) ++jhi;
}

// FILL = false: counts[j] += halos that go to rank j.  FILL = true: rows[start[j] + ...] = the packed rows (cursor zeroed by the
// caller); the order of the rows inside a destination is arbitrary.  A few hundred workgroups, each owning a contiguous run of
// halos: a workgroup adds to / reserves from the per-destination counters ONCE (with one workgroup per 256 halos the 3 900
// same-address atomics per counter serialised into 20 us).
constexpr int kRouteGrid = 512;
template <bool FILL>
__global__ void __launch_bounds__(256)
route_halos_kernel(RouteArgs a, int64_t n, const int32_t *__restrict__ rings, int32_t *__restrict__ counts, int32_t *__restrict__ cursor,
                   double *__restrict__ rows)
{
    __shared__ int hist[kRouteMaxRanks], base[kRouteMaxRanks], taken[kRouteMaxRanks];
    const int tid = threadIdx.x;
    if (tid < kRouteMaxRanks) { hist[tid] = 0; taken[tid] = 0; }
    __syncthreads();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = min(n, j0 + per);
    for (int64_t j = j0 + tid; j < j1; j += 256) {
        int jlo, jhi;
        route_range(a, rings[2 * j], rings[2 * j + 1], jlo, jhi);
        for (int d = jlo; d <= jhi; ++d) atomicAdd(&hist[d], 1);
    }
    __syncthreads();
    if (!FILL) {
        if (tid < a.world && hist[tid]) atomicAdd(counts + tid, hist[tid]);
        return;
    }
    if (tid < a.world) base[tid] = hist[tid] ? atomicAdd(cursor + tid, hist[tid]) : 0;       // one range per (workgroup, destination)
    __syncthreads();
    for (int64_t j = j0 + tid; j < j1; j += 256) {
        int jlo, jhi;
        route_range(a, rings[2 * j], rings[2 * j + 1], jlo, jhi);
        for (int d = jlo; d <= jhi; ++d) {
            const int64_t slot = base[d] + atomicAdd(&taken[d], 1);
            if (a.blockcap > 0) {
                // fixed-capacity blocks: the splits of the all_to_all are known without reading any count back
                if (slot >= a.blockcap) { *a.overflow = 1; continue; }
                double *blk = rows + (int64_t)d * a.ncols * a.blockcap + slot;
                for (int c = 0; c < a.ncols; ++c) blk[c * a.blockcap] = a.col[c][j];
            } else {
                double *row = rows + (a.start[d] + slot) * a.ncols;
                for (int c = 0; c < a.ncols; ++c) row[c] = a.col[c][j];
            }
        }
    }
}

// column 0 (M) of every fixed-capacity block := NaN, so that the rows a destination does not receive are dropped by K0 as invalid halos
__global__ void __launch_bounds__(256)
route_blank_kernel(int32_t world, int32_t ncols, int64_t blockcap, double *__restrict__ rows)
{
    const int64_t n = (int64_t)world * blockcap;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        rows[(i / blockcap) * ncols * blockcap + (i % blockcap)] = __builtin_nan("");
}

// ONE routing pass of a resident multi-GPU step (bfgx_route_step_device): per halo the ring range its disc can touch (the arithmetic of
// disc_rings_kernel, + margin rings) and its rows packed into fixed-capacity blocks [position][column][blockcap], one per destination.  The rows
// bound for this rank ITSELF never enter the collective: they are written straight into the last block of the receive buffer; the blocks of
// the other destinations are packed in rank order without a gap (position q(d) = d below this rank, d - 1 above), which is the layout
// all_to_all_single gives the receiver too when the split towards oneself is empty.  Column 0 (M) of every block was set to NaN by
// route_prepare_kernel: rows nobody fills are dropped by K0 as invalid halos.
struct RouteStepArgs {
    RouteArgs r;
    int32_t rank, margin;
    const double *M, *z, *dec;                   // the local halos' columns the ring range needs (also among r.col)
    double *send, *recv;                         // [world - 1] blocks to send, [world] blocks received (self last)
};
__global__ void __launch_bounds__(256)
route_prepare_kernel(int32_t world, int32_t ncols, int64_t blockcap, double *__restrict__ send, double *__restrict__ recv, int32_t *__restrict__ cursor)
{
    if (blockIdx.x == 0 && (int)threadIdx.x < world) cursor[threadIdx.x] = 0;
    const int64_t nb = (int64_t)(world - 1) * blockcap, nr = (int64_t)world * blockcap;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < nb + nr; i += (int64_t)gridDim.x * 256) {
        double *base = i < nb ? send : recv;
        const int64_t k = i < nb ? i : i - nb;
        base[(k / blockcap) * ncols * blockcap + (k % blockcap)] = __builtin_nan("");
    }
}
__global__ void __launch_bounds__(256)
route_step_kernel(RouteStepArgs s, DevModel m, Hpx h, int64_t n, int32_t *__restrict__ cursor)
{
    const RouteArgs &a = s.r;
    __shared__ int hist[kRouteMaxRanks], base[kRouteMaxRanks], taken[kRouteMaxRanks];
    const int tid = threadIdx.x;
    if (tid < kRouteMaxRanks) { hist[tid] = 0; taken[tid] = 0; }
    __syncthreads();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = min(n, j0 + per);
    const int nl4 = (int)(4 * h.nside);
    auto ring_range = [&](int64_t j, int &first, int &last) {           // disc_rings_kernel's arithmetic
        const double M_j = s.M[j], z_j = s.z[j];
        const double aa = 1.0 / (1.0 + z_j);
        const double R = dev_radius(m.bg_runner, m.md_runner, M_j, aa);
        double D;
        {
            int i = (int)floor(z_j / m.da_step);
            i = max(0, min(i, kDaKnots - 2));
            const double t = z_j - (double)i * m.da_step;
            const double *c = m.da_coef + 4 * i;
            D = ((c[0] * t + c[1]) * t + c[2]) * t + c[3];
        }
        const double radius = R * m.eps_runner / D;
        const double theta = kHalfPi - s.dec[j] * kDeg2Rad;
        first = 1; last = 0;
        const bool bad = !(radius > 0.0) || !isfinite(radius) || !(theta >= 0.0) || !(theta <= kPi) || !(M_j > 0.0) || !isfinite(M_j) || !(z_j > -1.0);
        if (bad) return;
        if (radius >= kPi) { first = 1; last = nl4 - 1; }
        else {
            const double lo = fmax(theta - radius, 0.0), hi = fmin(theta + radius, kPi);
            first = (int)ring_above(h, cos(lo)) - 1;
            last = (int)ring_above(h, cos(hi)) + 2;
            first = max(1, first - 1); last = min(nl4 - 1, last + 1);
        }
        first = max(1, first - s.margin); last = min(nl4 - 1, last + s.margin);
    };
    // (pass 1: how many rows this workgroup has for every destination; the ring ranges of the workgroup's run of halos wait in LDS for pass 2)
    constexpr int kStash = 4096;
    __shared__ int2 stash[kStash];
    for (int64_t j = j0 + tid; j < j1; j += 256) {
        int first, last, jlo, jhi;
        ring_range(j, first, last);
        if (j - j0 < kStash) stash[j - j0] = make_int2(first, last);
        route_range(a, first, last, jlo, jhi);
        for (int d = jlo; d <= jhi; ++d) atomicAdd(&hist[d], 1);
    }
    __syncthreads();
    if (tid < a.world) base[tid] = hist[tid] ? atomicAdd(cursor + tid, hist[tid]) : 0;       // one range per (workgroup, destination)
    __syncthreads();
    for (int64_t j = j0 + tid; j < j1; j += 256) {
        int first, last, jlo, jhi;
        if (j - j0 < kStash) { const int2 q = stash[j - j0]; first = q.x; last = q.y; } else ring_range(j, first, last);
        route_range(a, first, last, jlo, jhi);
        for (int d = jlo; d <= jhi; ++d) {
            const int64_t slot = base[d] + atomicAdd(&taken[d], 1);
            if (slot >= a.blockcap) { *a.overflow = 1; continue; }
            double *blk = (d == s.rank) ? s.recv + (int64_t)(a.world - 1) * a.ncols * a.blockcap
                                        : s.send + (int64_t)(d < s.rank ? d : d - 1) * a.ncols * a.blockcap;
            blk += slot;
            for (int c = 0; c < a.ncols; ++c) blk[c * a.blockcap] = a.col[c][j];
        }
    }
}

// The fast kernel's form when the halo count does not decide it: form[0] = 1 (fluid) if the catalog's estimated pairs per tile reach
// `thr`, else 0 (barrier per tile).  One workgroup sums K0's per-workgroup estimates.
__global__ void __launch_bounds__(1024)
k1_form_kernel(int nblk, const float *__restrict__ work_est, float thr_total, int32_t *__restrict__ form)
{
    __shared__ float red[1024 / kWave];
    float s = 0.0f;
    for (int i = threadIdx.x; i < nblk; i += 1024) s += work_est[i];
#pragma unroll
    for (int sft = kWave >> 1; sft > 0; sft >>= 1) s += __shfl_xor(s, sft, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) red[threadIdx.x / kWave] = s;
    __syncthreads();
    if (threadIdx.x == 0) {
        float t = 0.0f;
        for (int w = 0; w < 1024 / kWave; ++w) t += red[w];
        form[0] = t >= thr_total ? 1 : 0;
    }
}

// exclusive scan of the per-tile entry counts (one workgroup); start[ntiles] = total.  A tile's list is laid out as
// [narrow, slots reserved by K0 | narrow, many-tile halos | wide]
__global__ void __launch_bounds__(1024)
tile_scan_kernel(int ntiles, const int32_t *__restrict__ cnt_a, const int32_t *__restrict__ cnt_b, const int32_t *__restrict__ cnt_w,
                 int32_t *__restrict__ start, int32_t *__restrict__ wide_tiles)
{
    // both exclusive scans (entries per tile; tiles that have wide entries) in one pass: per-thread run of consecutive tiles,
    // wave scan by shuffles, one barrier, the 16 wave totals scanned by every thread from LDS
    constexpr int kPer = 8;                              // tiles per thread and round (8192 tiles per round)
    __shared__ int32_t wtot[2][1024 / kWave];
    __shared__ int32_t carry[2];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    if (tid < 2) carry[tid] = 0;
    __syncthreads();
    for (int base = 0; base < ntiles; base += 1024 * kPer) {
        const int lo = min(base + tid * kPer, ntiles), hi = min(lo + kPer, ntiles);
        int32_t c[kPer], wv[kPer], s = 0, nw = 0;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int i = lo + q;
            c[q] = 0; wv[q] = 0;
            if (i < hi) { const int32_t w = cnt_w[i]; c[q] = cnt_a[i] + cnt_b[i] + w; wv[q] = w > 0 ? 1 : 0; }
            s += c[q]; nw += wv[q];
        }
        int32_t is = s, iw = nw;                         // inclusive scans inside the wave
#pragma unroll
        for (int off = 1; off < kWave; off <<= 1) {
            const int32_t us = __shfl_up(is, off, kWave), uw = __shfl_up(iw, off, kWave);
            if (lane >= off) { is += us; iw += uw; }
        }
        if (lane == kWave - 1) { wtot[0][wid] = is; wtot[1][wid] = iw; }
        __syncthreads();
        int32_t ps = carry[0], pw = carry[1];
        for (int w = 0; w < wid; ++w) { ps += wtot[0][w]; pw += wtot[1][w]; }
        int32_t run = ps + is - s, pos = pw + iw - nw;
#pragma unroll
        for (int q = 0; q < kPer; ++q) {
            const int i = lo + q;
            if (i < hi) {
                start[i] = run; run += c[q];
                if (wv[q]) wide_tiles[1 + pos++] = i;    // the tiles that have wide entries: [0] = their number, [1..] = the tiles
            }
        }
        __syncthreads();
        if (tid == 1023) { carry[0] = run; carry[1] = pos; }
        __syncthreads();
    }
    if (tid == 0) { start[ntiles] = carry[0]; wide_tiles[0] = carry[1]; }
}

// The same scan for shells with more tiles than one workgroup covers in a round (NSIDE >= 2048: 2.4e4 .. 3.9e5 tiles, where the
// single workgroup above spends up to 1 ms in 48 serial rounds): PHASE 0 = every workgroup sums its 8192 tiles into
// block_tot[2 b .. 2 b + 1]; tile_scan_blocks_kernel scans those (<= 1024 blocks) into block_off and writes the two totals;
// PHASE 1 = every workgroup scans its tiles starting from its offsets.
constexpr int kScanTilesPerWg = 1024 * 8;
template <int PHASE>
__global__ void __launch_bounds__(1024)
tile_scan_part_kernel(int ntiles, const int32_t *__restrict__ cnt_a, const int32_t *__restrict__ cnt_b, const int32_t *__restrict__ cnt_w,
                      int32_t *__restrict__ start, int32_t *__restrict__ wide_tiles, int32_t *__restrict__ block_tot, const int32_t *__restrict__ block_off)
{
    constexpr int kPer = 8;
    __shared__ int32_t wtot[2][1024 / kWave];
    const int tid = threadIdx.x, lane = tid & (kWave - 1), wid = tid / kWave;
    const int base = blockIdx.x * kScanTilesPerWg;
    const int lo = min(base + tid * kPer, ntiles), hi = min(lo + kPer, ntiles);
    int32_t c[kPer], wv[kPer], s = 0, nw = 0;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int i = lo + q;
        c[q] = 0; wv[q] = 0;
        if (i < hi) { const int32_t w = cnt_w[i]; c[q] = cnt_a[i] + cnt_b[i] + w; wv[q] = w > 0 ? 1 : 0; }
        s += c[q]; nw += wv[q];
    }
    int32_t is = s, iw = nw;
#pragma unroll
    for (int off = 1; off < kWave; off <<= 1) {
        const int32_t us = __shfl_up(is, off, kWave), uw = __shfl_up(iw, off, kWave);
        if (lane >= off) { is += us; iw += uw; }
    }
    if (lane == kWave - 1) { wtot[0][wid] = is; wtot[1][wid] = iw; }
    __syncthreads();
    int32_t ps = 0, pw = 0;
    for (int w = 0; w < wid; ++w) { ps += wtot[0][w]; pw += wtot[1][w]; }
    if (PHASE == 0) {
        if (tid == 1023) { block_tot[2 * blockIdx.x] = ps + is; block_tot[2 * blockIdx.x + 1] = pw + iw; }
        return;
    }
    int32_t run = block_off[2 * blockIdx.x] + ps + is - s, pos = block_off[2 * blockIdx.x + 1] + pw + iw - nw;
#pragma unroll
    for (int q = 0; q < kPer; ++q) {
        const int i = lo + q;
        if (i < hi) {
            start[i] = run; run += c[q];
            if (wv[q]) wide_tiles[1 + pos++] = i;
        }
    }
}

// exclusive scan of the per-block totals (nblocks <= 1024); start[ntiles] = all entries, wide_tiles[0] = tiles with wide entries
__global__ void __launch_bounds__(1024)
tile_scan_blocks_kernel(int nblocks, int ntiles, const int32_t *__restrict__ block_tot, int32_t *__restrict__ block_off,
                        int32_t *__restrict__ start, int32_t *__restrict__ wide_tiles)
{
    __shared__ int32_t sa[1024], sb[1024];
    const int tid = threadIdx.x;
    sa[tid] = tid < nblocks ? block_tot[2 * tid] : 0;
    sb[tid] = tid < nblocks ? block_tot[2 * tid + 1] : 0;
    __syncthreads();
    if (tid == 0) {
        int32_t ra = 0, rb = 0;
        for (int b = 0; b < nblocks; ++b) { const int32_t ta = sa[b], tb = sb[b]; block_off[2 * b] = ra; block_off[2 * b + 1] = rb; ra += ta; rb += tb; }
        start[ntiles] = ra; wide_tiles[0] = rb;
    }
}

// tile binning, pass 2 (thread per halo): narrow halos with reserved slots are a plain scatter of halo indices; the others
// draw slots from their region's cursor (many-tile halos enumerate their tiles again from the span kept in the TileRef)
__global__ void __launch_bounds__(256)
tile_place_kernel(Hpx h, Tiling T, int64_t nhalo, const TileRef *__restrict__ tref,
                  const int32_t *__restrict__ start, const int32_t *__restrict__ cnt_a, const int32_t *__restrict__ cnt_b,
                  int32_t *__restrict__ cur_b, int32_t *__restrict__ cur_w,
                  int32_t *__restrict__ entries, int64_t capacity, int32_t *__restrict__ overflow,
                  const int32_t *__restrict__ slow_list, const int32_t *__restrict__ slow_cnt)
{
    // workgroup b = the halos K0's workgroup b left to this pass (slow_list[b][256], slow_cnt[b]); cnt_a is an array of zeros when region
    // A lives in the fixed-capacity lists K0 fills itself (the shared list then holds [B | wide] per tile)
    int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (slow_list) {
        if ((int)threadIdx.x >= slow_cnt[blockIdx.x]) return;
        j = slow_list[j];
    }
    if (j >= nhalo) return;
    const TileRef ref = tref[j];
    if (ref.cls == kClsNone) return;
    auto put = [&](int64_t pos) { if (pos < capacity) entries[pos] = (int32_t)j; else *overflow = 1; };
    if (ref.n <= kRefMax) {
        for (int i = 0; i < kRefMax; ++i) if (i < ref.n) {
            const int t = ref.few.tile[i];
            if (ref.cls == kClsNarrow) {
                if (!slow_list) put((int64_t)start[t] + ref.few.slot[i]);
                else if (ref.few.slot[i] < 0) put((int64_t)start[t] + cnt_a[t] + atomicAdd(cur_b + t, 1));      // its tile's fixed list was full
            }
            else put((int64_t)start[t] + cnt_a[t] + cnt_b[t] + atomicAdd(cur_w + t, 1));
        }
    } else {
        DiscSpan ds;
        ds.fb = 0; ds.rfirst = ref.many.rfirst; ds.rlast = ref.many.rlast; ds.allphi = ref.many.allphi;
        ds.flo = ref.many.flo; ds.fhi = ref.many.fhi;
        for (int q = 0; q < 4; ++q) { ds.fb_ring[q] = 0; ds.fb_k[q] = 0; }
        for_each_tile(h, T, ds, [&](int t) {
            if (ref.cls == kClsNarrow) put((int64_t)start[t] + cnt_a[t] + atomicAdd(cur_b + t, 1));
            else put((int64_t)start[t] + cnt_a[t] + cnt_b[t] + atomicAdd(cur_w + t, 1));
        });
    }
}

// ---------------------------------------------------------------------------------- per-pair math
template <typename ACC> __device__ inline void atomic_accumulate(ACC *p, double v) { atomicAdd(p, (ACC)v); }

// linear read-out along ln r of the 4 (z,M)-corner rows; NaN outside the axis (scipy RGI semantics)
template <int NC>
__device__ inline double radial_readout(const PairTable &t, const int32_t *rowoff, const double *w, double lx)
{
    const int n = t.nr;
    if (!(lx >= t.r0) || !(lx <= t.r1)) return __builtin_nan("");
    int i;
    double tr;
    if (t.r_uniform) {        // uniform ln r axis: O(1) index, fraction from the same affine map
        const double u = (lx - t.r0) * t.inv_dr;
        i = min((int)u, n - 2);
        tr = u - (double)i;
    } else {
        const double *g = t.raxis;
        int lo = 0, hi = n - 1;
        while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (lx >= g[mid]) lo = mid; else hi = mid; }
        i = lo;
        tr = (lx - g[i]) / (g[i + 1] - g[i]);
    }
    const double t0 = 1.0 - tr;
    double val = 0.0;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double *row = t.values + rowoff[c] + i;
        val = val + row[0] * (w[c] * t0);
        val = val + row[1] * (w[c] * tr);
    }
    return val;
}

// halo fields the pair phase needs (a view: scalar registers in the halo-centric kernel, LDS in the tiled one)
template <int NC>
struct PairHaloT {
    double z0, s0, phi0, cph0, sph0, D, a, inv_a, rcut;
    double lnoff;                         // ln(1/a) [- ln R_model when Rdelta_sampling]: ln r axis coordinate offset
    double w[NC];
    int32_t rowoff[NC];
    int32_t oob, hidx;
};
using PairHalo = PairHaloT<kNC>;

template <int NC>
__device__ inline void load_pair_halo(PairHaloT<NC> &p, const HaloRec &r, int hidx, const RowSetX *rowsx = nullptr)
{
    p.z0 = r.z0; p.s0 = r.s0; p.phi0 = r.phi0; p.cph0 = r.cph0; p.sph0 = r.sph0;
    p.D = r.D; p.a = r.a; p.inv_a = r.inv_a; p.rcut = r.rcut; p.lnoff = r.lnoff;
    if (NC == kNC) {
        for (int c = 0; c < kNC; ++c) { p.w[c] = r.w[c]; p.rowoff[c] = r.rowoff[c]; }
    } else {
        const RowSetX &x = rowsx[hidx];
        for (int c = 0; c < NC; ++c) { p.w[c] = x.w[c]; p.rowoff[c] = x.rowoff[c]; }
    }
    p.oob = r.oob; p.hidx = hidx;
}

// One (halo, pixel) pair.  Returns false when the pair contributes nothing.
//   MODE_OFFSETS: v[0..2] = nw_vec - vec (HealpixRunner.py:314-328);  MODE_PAINT: v[0] = Paint (:441-442)
template <int MODE, int NC>
__device__ inline bool pair_value(const PairTable &tab, const PairHaloT<NC> &r, double z, double sth, double phi_pix, double v[3])
{
    // pixel unit vector in the frame rotated by -phi0 about the polar axis: halo at (s0, 0, z0)
    double sd, cd;
    sincos_dphi(phi_pix - r.phi0, sd, cd);
    const double vx = sth * cd, vy = sth * sd, vz = z;
    const double dx = r.D * (vx - r.s0), dy = r.D * vy, dz = r.D * (vz - r.z0);   // :314-316
    const double r2 = dx * dx + dy * dy + dz * dz;
    if (!(r2 > 0.0)) return false;                 // r_sep = 0: diff/r_sep is NaN -> 0 (:322-323); ln 0 -> NaN paint
    const double inv_r = fast_rsq(r2);
    const double r_sep = r2 * inv_r;                                                // :317
    const double r_com = r_sep * r.inv_a;                                           // :321 (r_sep / a)
    const double lx = __builtin_fma(0.5, fast_log(r2), r.lnoff);                    // ln(r_sep/a) [- ln R when Rdelta]
    double d = r.oob ? __builtin_nan("") : radial_readout<NC>(tab, r.rowoff, r.w, lx);
    if (MODE == MODE_PAINT) {
        const double paint = fast_exp(d);                                                // Tabulate.py:286
        v[0] = paint;
        return isfinite(paint) && paint != 0.0;                                     // :442
    }
    if (!(r_com < r.rcut)) d = 0.0;                                                 // BaryonCorrection.py:381-382
    if (!isfinite(d) || d == 0.0) return false;    // non-finite -> 0 (:323); the reference then adds ~1e-17 noise
    d *= r.a * inv_r;                                                               // :321-322
    const double ox = d * dx, oy = d * dy, oz = d * dz;
    const double nx = __builtin_fma(r.D, vx, ox), ny = __builtin_fma(r.D, vy, oy), nz = __builtin_fma(r.D, vz, oz);   // :326
    const double inv_n = fast_rsq(nx * nx + ny * ny + nz * nz);                     // :327
    const double ex = __builtin_fma(nx, inv_n, -vx), ey = __builtin_fma(ny, inv_n, -vy);   // :328
    v[2] = __builtin_fma(nz, inv_n, -vz);
    v[0] = ex * r.cph0 - ey * r.sph0;                                               // rotate back by +phi0
    v[1] = ex * r.sph0 + ey * r.cph0;
    return true;
}

// Branch-free (fully predicated) variant for the common case: uniform ln r axis and |dphi| <= 0.5, so that two
// pairs per lane can be interleaved by the scheduler.  x = azimuth difference already folded to (-pi, pi].
template <int MODE, int NC>
__device__ inline bool pair_value_fast(const PairTable &tab, const PairHaloT<NC> &r, double z, double sth, double x, double v[3])
{
    double sd, cd;
    sincos_small(x, sd, cd);
    const double vx = sth * cd, vy = sth * sd, vz = z;
    const double dx = r.D * (vx - r.s0), dy = r.D * vy, dz = r.D * (vz - r.z0);   // :314-316
    const double r2 = dx * dx + dy * dy + dz * dz;
    bool ok = (r2 > 0.0) && !r.oob;
    const double r2s = (r2 > 0.0) ? r2 : 1.0;
    const double inv_r = fast_rsq(r2s);
    const double r_com = r2s * inv_r * r.inv_a;                                     // :317, :321
    const double lx = __builtin_fma(0.5, fast_log(r2s), r.lnoff);
    ok = ok && (lx >= tab.r0) && (lx <= tab.r1);                                    // RGI fill_value = nan
    const double u = (lx - tab.r0) * tab.inv_dr;
    const int i = max(0, min((int)u, tab.nr - 2));
    const double tr = u - (double)i, t0 = 1.0 - tr;
    double d = 0.0;
#if BFGX_ABLATE == 1      // timing-only build: no table loads
    d = tr * r.w[0] + t0 * r.w[1] + 1e-3;
#else
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const double *row = tab.values + r.rowoff[c] + i;
        d = d + row[0] * (r.w[c] * t0);
        d = d + row[1] * (r.w[c] * tr);
    }
#endif
    if (MODE == MODE_PAINT) {
        const double paint = fast_exp(d);                                                // Tabulate.py:286
        v[0] = paint;
        return ok && isfinite(paint) && paint != 0.0;                               // :442
    }
    ok = ok && (r_com < r.rcut) && isfinite(d) && d != 0.0;                         // BaryonCorrection.py:381-382, :323
    d *= r.a * inv_r;                                                               // :321-322
    const double ox = d * dx, oy = d * dy, oz = d * dz;
    const double nx = __builtin_fma(r.D, vx, ox), ny = __builtin_fma(r.D, vy, oy), nz = __builtin_fma(r.D, vz, oz);   // :326
    const double inv_n = fast_rsq(nx * nx + ny * ny + nz * nz);                     // :327
    const double ex = __builtin_fma(nx, inv_n, -vx), ey = __builtin_fma(ny, inv_n, -vy);   // :328
    v[2] = __builtin_fma(nz, inv_n, -vz);
    v[0] = ex * r.cph0 - ey * r.sph0;
    v[1] = ex * r.sph0 + ey * r.cph0;
    return ok;
}

__device__ inline double fold_dphi(double x)
{
    x = (x > kPi) ? x - kTwoPi : x;
    return (x < -kPi) ? x + kTwoPi : x;
}

// ---------------------------------------------------------------------------------- K1 / K3, halo-centric
// (algo 0: one wave per halo, global float atomics; kept as the simple reference implementation)
struct RowLds {                      // one wave's 64 ring rows
    int32_t prefix[kWave];           // exclusive prefix of pixel counts
    int32_t nr[kWave];               // pixels in ring
    int32_t lo[kWave];               // first in-disc pixel index within the ring, in [0, nr)
    int64_t start[kWave];            // first pixel of ring
    double z[kWave], sth[kWave], shift[kWave];
};

__device__ inline int wave_scan_incl(int v, int lane)
{
#pragma unroll
    for (int s = 1; s < kWave; s <<= 1) {
        const int u = __shfl_up(v, s, kWave);
        if (lane >= s) v += u;
    }
    return v;
}

// Mixed-precision form of pair_value_fast for MODE_OFFSETS with fp32 pix_offsets output.  The chord -- a difference of
// nearly equal unit vectors -- and the r < rcut decision stay fp64; everything downstream of r^2 runs in fp32: 1/r and
// ln r from the hardware v_rsq_f32 / v_log_f32, the table read-out from an fp32 copy of the table, and the renormalised
// offset as a series in eps = offset / D,
//     (v + eps) / |v + eps| - v = eps + g (v + eps),   g = -t/2 + 3 t^2/8 - 5 t^3/16,   t = 2 v.eps + eps.eps   (|v| = 1),
// which has no cancellation left.  Relative error of an offset ~3e-7 (|offset| ~ 1e-5 rad, so ~1e-12 rad absolute): far
// inside the fp32 storage of pix_offsets and the stated 1e-6 mean(map) bound of the default BaryonifyShell path.
template <int NC>
__device__ inline bool pair_value_fast32(const PairTable &tab, const PairHaloT<NC> &r, double z, double sth, double x, double v[3])
{
    double sd, cd;
    sincos_small(x, sd, cd);
    const double vx = sth * cd, vy = sth * sd;
    const double dx = r.D * (vx - r.s0), dy = r.D * vy, dz = r.D * (z - r.z0);      // :314-316
    const double r2 = dx * dx + dy * dy + dz * dz;
    const double rc = r.rcut * r.a;                                                 // r_sep / a < rcut  (:321, BaryonCorrection.py:381)
    bool ok = (r2 > 0.0) && !r.oob && (r2 < rc * rc);
    const float r2f = (float)((r2 > 0.0) ? r2 : 1.0);
    float inv_r = __builtin_amdgcn_rsqf(r2f);
    inv_r = inv_r * __builtin_fmaf(-0.5f * r2f * inv_r, inv_r, 1.5f);               // one Newton step
    const float lx = __builtin_fmaf(0.34657359027997264f, __builtin_amdgcn_logf(r2f), (float)r.lnoff);   // 0.5 ln 2 * log2(r^2)
    const float r0f = (float)tab.r0;
    ok = ok && (lx >= r0f) && (lx <= (float)tab.r1);                                // RGI fill_value = nan
    const float u = (lx - r0f) * (float)tab.inv_dr;
    const int i = max(0, min((int)u, tab.nr - 2));
    const float tr = u - (float)i, t0 = 1.0f - tr;
    float d = 0.0f;
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const float *row = tab.values32 + r.rowoff[c] + i;
        const float w = (float)r.w[c];
        d = __builtin_fmaf(row[0], w * t0, d);
        d = __builtin_fmaf(row[1], w * tr, d);
    }
    ok = ok && isfinite(d) && d != 0.0f;                                            // :323
    const float sc = d * (float)r.a * inv_r * __builtin_amdgcn_rcpf((float)r.D);    // offset / D = d a diff / (r D)
    const float ex0 = sc * (float)dx, ey0 = sc * (float)dy, ez0 = sc * (float)dz;
    const float fx = (float)vx, fy = (float)vy, fz = (float)z;
    const float t = 2.0f * (fx * ex0 + fy * ey0 + fz * ez0) + (ex0 * ex0 + ey0 * ey0 + ez0 * ez0);
    const float g = t * __builtin_fmaf(t, __builtin_fmaf(t, -0.3125f, 0.375f), -0.5f);
    const float ex = __builtin_fmaf(g, fx + ex0, ex0), ey = __builtin_fmaf(g, fy + ey0, ey0), ez = __builtin_fmaf(g, fz + ez0, ez0);   // :326-328
    const float cp = (float)r.cph0, sp = (float)r.sph0;
    v[0] = (double)(ex * cp - ey * sp);                                             // rotate back by +phi0
    v[1] = (double)(ex * sp + ey * cp);
    v[2] = (double)ez;
    return ok;
}

template <int MODE, typename ACC>
__global__ void __launch_bounds__(kWave * kWavesPerBlock)
halo_scatter_kernel(PairTable pt, Hpx h, int64_t nhalo, const HaloRec *__restrict__ recs,
                    ACC *__restrict__ out, int64_t *__restrict__ counts)
{
    __shared__ RowLds lds[kWavesPerBlock];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);   // wave-uniform -> scalar loads of the record
    const int64_t j = (int64_t)blockIdx.x * kWavesPerBlock + wid;
    if (j >= nhalo) return;                      // whole wave exits together
    RowLds &L = lds[wid];
    const HaloRec &r = recs[j];
    PairHalo ph;
    load_pair_halo(ph, r, (int)j);
    constexpr int NCOMP = (MODE == MODE_OFFSETS) ? 3 : 1;

    int64_t npairs = 0;
    if (r.fb) {                                                           // HealpixRunner.py:309-310
        npairs = 4;
        if (MODE != MODE_COUNT && lane < 4) {
            int64_t st, nr; bool sh; double z, sth, v[3];
            ring_info_small(h, r.fb_ring[lane], st, nr, sh);
            ring_z_sth(h, r.fb_ring[lane], z, sth);
            const double phi_pix = ((double)r.fb_k[lane] + (sh ? 0.5 : 0.0)) * (kTwoPi / (double)nr);
            if (pair_value<MODE, kNC>(pt, ph, z, sth, phi_pix, v)) {
                ACC *o = out + NCOMP * (st + r.fb_k[lane]);
                for (int c = 0; c < NCOMP; ++c) atomic_accumulate(o + c, v[c]);
            }
        }
    } else {
        for (int rbase = r.rfirst; rbase <= r.rlast; rbase += kWave) {
            const int ring = rbase + lane;
            RowSpan s;
            s.cnt = 0; s.lo = 0; s.start = 0; s.nr = 1; s.shifted = false; s.z = 0.0; s.sth = 0.0;
            if (ring <= r.rlast) disc_row(h, ring, r.z0, r.xa, r.cosr, r.phi0, r.irmin, r.irmax, s);
            const int incl = wave_scan_incl(s.cnt, lane);
            const int total = __shfl(incl, kWave - 1, kWave);
            L.prefix[lane] = incl - s.cnt;
            L.nr[lane] = (int)s.nr; L.lo[lane] = s.lo; L.start[lane] = s.start;
            L.z[lane] = s.z; L.sth[lane] = s.sth; L.shift[lane] = s.shifted ? 0.5 : 0.0;
            __builtin_amdgcn_wave_barrier();
            npairs += total;
            if (MODE != MODE_COUNT) {
                for (int t = lane; t < total; t += kWave) {
                    int row = 0;                      // largest row with prefix[row] <= t
#pragma unroll
                    for (int st = kWave >> 1; st > 0; st >>= 1)
                        if (L.prefix[row + st] <= t) row += st;
                    const int nrr = L.nr[row];
                    int k = L.lo[row] + (t - L.prefix[row]);
                    if (k >= nrr) k -= nrr;
                    const double phi_pix = ((double)k + L.shift[row]) * (kTwoPi / (double)nrr);
                    double v[3];
                    if (pair_value<MODE, kNC>(pt, ph, L.z[row], L.sth[row], phi_pix, v)) {
                        ACC *o = out + NCOMP * (L.start[row] + k);
                        for (int c = 0; c < NCOMP; ++c) atomic_accumulate(o + c, v[c]);
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    if (counts && lane == 0) counts[j] = npairs;
}

// ---------------------------------------------------------------------------------- models that are Python callables
// The reference calls model.displacement(r_sep / a, M, a) / model.projected(cosmo, r_sep / a, M, a) ONCE PER HALO on any object, with the
// separations of that halo's own pixels (HealpixRunner.py:314-321, :436-441).  A model that carries no table can be served exactly that way:
// WHAT 0 writes r_sep / a of every (halo, pixel) pair -- halo j's pairs at off[j] .., in the order of halo_scatter_kernel's enumeration --,
// the host calls the model per halo on its slice, and WHAT 1 / 2 add what it returned:
//   WHAT 1 (BaryonifyShell):      offset = value * a * diff / r_sep, non-finite components -> 0 (:321-323), new unit vector - old (:326-328)
//   WHAT 2 (PaintProfilesShell):  Paint = value, non-finite -> 0 (:441-442)
// No table, no model-side cut (the callable makes its own, BaryonCorrection.py:381-382).  Global fp64 atomics: the per-halo Python calls
// around this kernel take seconds, the kernel microseconds.
template <int WHAT>
__global__ void __launch_bounds__(kWave * kWavesPerBlock)
halo_pairs_kernel(Hpx h, int64_t nhalo, const HaloRec *__restrict__ recs, const int64_t *__restrict__ off, const double *__restrict__ vals,
                  double *__restrict__ r_out, double *__restrict__ out)
{
    __shared__ RowLds lds[kWavesPerBlock];
    const int lane = threadIdx.x & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(threadIdx.x / kWave);
    const int64_t j = (int64_t)blockIdx.x * kWavesPerBlock + wid;
    if (j >= nhalo) return;
    RowLds &L = lds[wid];
    const HaloRec &r = recs[j];
    const int64_t base = off[j];
    // one pair: pixel (start + k) of a ring with (z, sth); q = its index in the pair arrays
    auto pair = [&](int64_t q, int64_t pix, double z, double sth, double phi_pix) {
        double sd, cd;
        sincos_dphi(phi_pix - r.phi0, sd, cd);
        const double vx = sth * cd, vy = sth * sd, vz = z;                              // (frame rotated by -phi0: the halo at (s0, 0, z0))
        const double dx = r.D * (vx - r.s0), dy = r.D * vy, dz = r.D * (vz - r.z0);   // :314-316
        const double r_sep = sqrt(dx * dx + dy * dy + dz * dz);                         // :317
        if (WHAT == 0) { r_out[q] = r_sep * r.inv_a; return; }                          // :321 r_sep / a_j
        const double val = vals[q];
        if (WHAT == 2) {
            if (isfinite(val) && val != 0.0) atomic_accumulate(out + pix, val);         // :441-445
            return;
        }
        const double d = val * r.a;                                                     // :321
        double ox = d * (dx / r_sep), oy = d * (dy / r_sep), oz = d * (dz / r_sep);     // :322
        ox = isfinite(ox) ? ox : 0.0; oy = isfinite(oy) ? oy : 0.0; oz = isfinite(oz) ? oz : 0.0;   // :323, element by element
        if (ox == 0.0 && oy == 0.0 && oz == 0.0) return;
        const double nx = __builtin_fma(r.D, vx, ox), ny = __builtin_fma(r.D, vy, oy), nz = __builtin_fma(r.D, vz, oz);   // :326
        const double inv_n = 1.0 / sqrt(nx * nx + ny * ny + nz * nz);                   // :327
        const double ex = __builtin_fma(nx, inv_n, -vx), ey = __builtin_fma(ny, inv_n, -vy), ez = __builtin_fma(nz, inv_n, -vz);   // :328
        double *o = out + 3 * pix;
        atomic_accumulate(o, ex * r.cph0 - ey * r.sph0);                                // rotate back by +phi0
        atomic_accumulate(o + 1, ex * r.sph0 + ey * r.cph0);
        atomic_accumulate(o + 2, ez);
    };
    if (r.fb) {                                                                         // HealpixRunner.py:309-310
        if (lane < 4) {
            int64_t st, nr; bool sh; double z, sth;
            ring_info_small(h, r.fb_ring[lane], st, nr, sh);
            ring_z_sth(h, r.fb_ring[lane], z, sth);
            pair(base + lane, st + r.fb_k[lane], z, sth, ((double)r.fb_k[lane] + (sh ? 0.5 : 0.0)) * (kTwoPi / (double)nr));
        }
        return;
    }
    int64_t done = 0;
    for (int rbase = r.rfirst; rbase <= r.rlast; rbase += kWave) {
        const int ring = rbase + lane;
        RowSpan s;
        s.cnt = 0; s.lo = 0; s.start = 0; s.nr = 1; s.shifted = false; s.z = 0.0; s.sth = 0.0;
        if (ring <= r.rlast) disc_row(h, ring, r.z0, r.xa, r.cosr, r.phi0, r.irmin, r.irmax, s);
        const int incl = wave_scan_incl(s.cnt, lane);
        const int total = __shfl(incl, kWave - 1, kWave);
        L.prefix[lane] = incl - s.cnt;
        L.nr[lane] = (int)s.nr; L.lo[lane] = s.lo; L.start[lane] = s.start;
        L.z[lane] = s.z; L.sth[lane] = s.sth; L.shift[lane] = s.shifted ? 0.5 : 0.0;
        __builtin_amdgcn_wave_barrier();
        for (int t = lane; t < total; t += kWave) {
            int row = 0;
#pragma unroll
            for (int st = kWave >> 1; st > 0; st >>= 1)
                if (L.prefix[row + st] <= t) row += st;
            const int nrr = L.nr[row];
            int k = L.lo[row] + (t - L.prefix[row]);
            if (k >= nrr) k -= nrr;
            pair(base + done + t, L.start[row] + k, L.z[row], L.sth[row], ((double)k + L.shift[row]) * (kTwoPi / (double)nrr));
        }
        done += total;
        __builtin_amdgcn_wave_barrier();
    }
}

// ---------------------------------------------------------------------------------- K1 / K3, tiled
// (algo 1, default) one workgroup per tile: accumulators live in LDS, entries (halos touching the tile) are
// taken 16 at a time by each wave: lanes = entries -> lanes = ring rows (clipped to the tile) -> lanes = pairs.
constexpr int kChunk = 16;
constexpr int kMaskWords = 66;           // 64 rows x up to 64 pixels (W <= 64) = 4096 pairs -> 64 words (+2 spare)

struct RingSlot {                    // ring-phase view of one entry (the rare <4-pixel fallback pixels stay in the HaloRec)
    double z0, xa, cosr, phi0;
    int32_t irmin, irmax, ring_lo, prefix, fb, hidx;
};

// one ring of a tile: what the ring-row phase needs of it, computed once per workgroup instead of once per (halo, ring)
struct TileRow {
    double z, sth, dphi, fnr;         // cos / sin of the colatitude, 2 pi / nr, nr / (2 pi)
    int32_t nr, ks, ke, shifted;      // ring length, the tile's pixel span [ks, ke) in the ring, half-pixel shift flag
};

template <int NC>
struct TileWaveLdsT {
    PairHaloT<NC> pair[kChunk];
    RingSlot ring[kChunk];
    // rows of the current block, compacted to the non-empty ones
    int32_t prefix[kWave], firstA[kWave], cntA[kWave], firstB[kWave], ldsbase[kWave], eslot[kWave];
    double z[kWave], sth[kWave];
    double dphi[kWave], xoff[kWave];      // azimuth of pixel k relative to the halo: k * dphi + xoff  (xoff = shift * dphi - phi0)
    unsigned long long mask[kMaskWords];  // bit t set <=> pair t is the first pair of a row
};

template <int NC>
__host__ __device__ inline size_t tile_lds_bytes(int BR, int W, int ncomp)
{
    size_t a = (size_t)BR * W * ncomp * sizeof(double);
    a = (a + 15) & ~(size_t)15;
    return a + sizeof(TileWaveLdsT<NC>) * kWavesPerBlock + 16 + sizeof(TileRow) * (size_t)BR;
}

template <int MODE, typename ACC, int NC>
__global__ void __launch_bounds__(kWave * kWavesPerBlock)
tile_scatter_kernel(PairTable pt, Hpx h, Tiling T, const HaloRec *__restrict__ recs, const RowSetX *__restrict__ rowsx,
                    const int32_t *__restrict__ tile_start, const int32_t *__restrict__ entries, int64_t capacity,
                    ACC *__restrict__ out, unsigned long long *__restrict__ pair_total,
                    const int32_t *__restrict__ cnt_a, const int32_t *__restrict__ cnt_b, int addmode,
                    const int32_t *__restrict__ wide_tiles, unsigned int *__restrict__ omax2, int tile_lo, int tile_n)
{
    // wide_tiles != nullptr ("wide pass"): only the wide-halo region of every tile's entry list is processed (the narrow
    // halos went through the fast kernel, bfgx_scatter2.hpp, which has stored the tile: addmode) and only the tiles that
    // have wide entries are visited: wide_tiles[0] = their number, wide_tiles[1..] = the tiles (tile_scan_kernel); the grid
    // is small and fixed, so a catalog without wide halos costs a few microseconds
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NCOMP = (MODE == MODE_OFFSETS) ? 3 : 1;
    // tile_n >= 0: only the tiles [tile_lo, tile_lo + tile_n) (a rank that owns a range of bands)
    const int nvisit = wide_tiles ? wide_tiles[0] : (tile_n < 0 ? T.ntiles : tile_n);
    for (int bi = blockIdx.x; bi < nvisit; bi += gridDim.x) {
    // full pass: heavy (equatorial) tiles are dispatched first, the light polar ones fill the tail
    const int tile = wide_tiles ? wide_tiles[1 + bi] : (tile_n < 0 ? T.tile_order[bi] : tile_lo + bi);
    if (tile_n >= 0 && (tile < tile_lo || tile >= tile_lo + tile_n)) continue;       // (wide pass: the list covers the sphere)
    const int band = T.tile_band[tile];
    const int nphi = T.band_nphi[band];
    const int tj = tile - T.band_tile0[band];
    const int i0 = 1 + band * T.BR;
    const int i1 = min(i0 + T.BR, (int)(4 * h.nside));            // exclusive
    const int acc_n = T.BR * T.W * NCOMP;
    // LDS accumulators are always fp64: on gfx950 ds_add_f64 runs at ~7 lanes/clk/CU while ds_add_f32 manages
    // only ~0.3 (scripts/ubench/lds_atomics.hip); ACC is only the type of the global output.
    double *acc = reinterpret_cast<double *>(smem);
    size_t acc_bytes = ((size_t)acc_n * sizeof(double) + 15) & ~(size_t)15;
    using TileWaveLds = TileWaveLdsT<NC>;
    using PairH = PairHaloT<NC>;
    TileWaveLds *wl = reinterpret_cast<TileWaveLds *>(smem + acc_bytes);
    int *next_chunk = reinterpret_cast<int *>(wl + kWavesPerBlock);
    TileRow *rowtab = reinterpret_cast<TileRow *>(reinterpret_cast<unsigned char *>(next_chunk) + 16);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (MODE != MODE_COUNT)
        for (int i = tid; i < acc_n; i += kWave * kWavesPerBlock) acc[i] = 0.0;
    if (tid == 0) *next_chunk = 0;
    if (tid < T.BR && i0 + tid < i1) {
        const int ring = i0 + tid;
        int64_t st, n64; bool shf;
        TileRow tr;
        ring_info_small(h, ring, st, n64, shf);
        ring_z_sth(h, ring, tr.z, tr.sth);
        tr.nr = (int)n64; tr.shifted = shf ? 1 : 0;
        tr.dphi = kTwoPi / (double)tr.nr;
        tr.fnr = (double)n64 * kInvTwoPi;
        tr.ks = tile_ks(tj, tr.nr, nphi); tr.ke = tile_ks(tj + 1, tr.nr, nphi);
        rowtab[tid] = tr;
    }
    __syncthreads();

    const int64_t e0 = (int64_t)tile_start[tile] + (cnt_a ? (int64_t)cnt_a[tile] + cnt_b[tile] : 0);
    int64_t e1 = tile_start[tile + 1];
    if (e1 > capacity) e1 = capacity;
    const int ne = (int)(e1 > e0 ? e1 - e0 : 0);
    // entries a wave takes at a time: the wide pass visits a few polar tiles with a handful of entries each (and thousands of
    // pixels per entry), so small chunks keep all four waves of the tile busy
    // (full pass: one chunk per wave when the tile's list is short, as in the fast kernel)
    const int chunk = wide_tiles ? 2 : max(1, min(kChunk, (ne + kWavesPerBlock - 1) / kWavesPerBlock));
    const int nchunks = (ne + chunk - 1) / chunk;
    TileWaveLds &L = wl[wid];
    unsigned long long npairs = 0;

    while (true) {
        int c = 0;
        if (lane == 0) c = atomicAdd(next_chunk, 1);
        c = __shfl(c, 0, kWave);
        if (c >= nchunks) break;

        // ---- lanes = entries of this chunk
        int nrows = 0;
        if (lane < chunk && c * chunk + lane < ne) {
            const int hidx = entries[e0 + c * chunk + lane];
            const HaloRec &r = recs[hidx];
            RingSlot rs;
            rs.z0 = r.z0; rs.xa = r.xa; rs.cosr = r.cosr; rs.phi0 = r.phi0;
            rs.irmin = r.irmin; rs.irmax = r.irmax; rs.fb = r.fb; rs.hidx = hidx;
            if (r.fb) { nrows = 4; rs.ring_lo = 0; }
            else {
                const int lo = max(r.rfirst, i0), hi = min(r.rlast, i1 - 1);
                nrows = max(0, hi - lo + 1);
                rs.ring_lo = lo;
            }
            rs.prefix = 0;
            L.ring[lane] = rs;
            load_pair_halo<NC>(L.pair[lane], r, hidx, rowsx);
        }
        const int incl_e = wave_scan_incl(nrows, lane);
        const int total_rows = __shfl(incl_e, kWave - 1, kWave);
        if (lane < kChunk) L.ring[lane].prefix = incl_e - nrows;
        __builtin_amdgcn_wave_barrier();

        for (int rb = 0; rb < (BFGX_ABLATE == 8 ? 0 : total_rows); rb += kWave) {
            // ---- lanes = ring rows (clipped to this tile)
            const int R = rb + lane;
            int firstA = 0, cntA = 0, firstB = 0, cntB = 0, nr = 1, ldsbase = 0, es = 0;
            double z = 0.0, sth = 0.0, shift = 0.0, dphv = 0.0;
            if (R < total_rows) {
                int e = 0;                                        // largest entry with prefix <= R
#pragma unroll
                for (int st = kChunk >> 1; st > 0; st >>= 1)
                    if (L.ring[e + st].prefix <= R) e += st;
                es = e;
                const RingSlot &rs = L.ring[e];
                const int q = R - rs.prefix;
                if (rs.fb) {
                    const HaloRec &hr = recs[rs.hidx];            // rare: the 4 fallback pixels live in the halo record
                    const int ring = hr.fb_ring[q], fk = hr.fb_k[q];
                    if (ring >= i0 && ring < i1) {
                        const TileRow &tr = rowtab[ring - i0];
                        z = tr.z; sth = tr.sth; nr = tr.nr; shift = tr.shifted ? 0.5 : 0.0; dphv = tr.dphi;
                        if (fk >= tr.ks && fk < tr.ke) { firstA = fk; cntA = 1; }
                        ldsbase = (ring - i0) * T.W - tr.ks;
                    }
                } else {
                    const int ring = rs.ring_lo + q;
                    const TileRow &tr = rowtab[ring - i0];
                    int slo, scnt;
                    disc_row_span(tr.nr, tr.shifted != 0, tr.z, tr.fnr, ring, rs.z0, rs.xa, rs.cosr, rs.phi0, rs.irmin, rs.irmax, slo, scnt);
                    nr = tr.nr; z = tr.z; sth = tr.sth; shift = tr.shifted ? 0.5 : 0.0; dphv = tr.dphi;
                    const int ks = tr.ks, ke = tr.ke;
                    const int endA = min(slo + scnt, nr);
                    firstA = max(slo, ks);
                    cntA = max(0, min(endA, ke) - firstA);
                    const int endB = slo + scnt - nr;             // > 0 when the row wraps past phi = 2 pi
                    firstB = ks;                                  // max(0, ks)
                    cntB = max(0, min(endB, ke) - firstB);
                    ldsbase = (ring - i0) * T.W - ks;
                }
            }
            const int cnt = cntA + cntB;
            const int incl = wave_scan_incl(cnt, lane);
            const int total = __shfl(incl, kWave - 1, kWave);
            npairs += (unsigned long long)total;
            if (MODE != MODE_COUNT && total > 0) {
                // compact the non-empty rows and mark each row's first pair in a bit mask
                const int nwords = (total + kWave - 1) / kWave + 2;
                for (int wI = lane; wI < nwords; wI += kWave) L.mask[wI] = 0ull;
                __builtin_amdgcn_wave_barrier();
                const unsigned long long ne = __ballot(cnt > 0);
                const unsigned long long lt = (1ull << lane) - 1ull;
                if (cnt > 0) {
                    const int slot = __popcll(ne & lt);
                    const int pre = incl - cnt;
                    L.prefix[slot] = pre;
                    L.firstA[slot] = firstA; L.cntA[slot] = cntA; L.firstB[slot] = firstB;
                    L.ldsbase[slot] = ldsbase; L.eslot[slot] = es;
                    L.z[slot] = z; L.sth[slot] = sth;
                    L.dphi[slot] = dphv; L.xoff[slot] = shift * dphv - L.pair[es].phi0;
                    atomicOr(&L.mask[pre >> 6], 1ull << (pre & 63));
                }
                __builtin_amdgcn_wave_barrier();

                // ---- lanes = (halo, pixel) pairs, two per lane per trip
                const unsigned long long le = lt | (1ull << lane);
                const bool fastok = pt.r_uniform != 0;
                int base = 0;                                      // rows started before the current 64 pairs
                for (int T0 = 0; T0 < (BFGX_ABLATE == 4 ? 0 : total); T0 += 2 * kWave) {
                    const unsigned long long mA = L.mask[T0 >> 6], mB = L.mask[(T0 >> 6) + 1];
                    const int tA = T0 + lane, tB = T0 + kWave + lane;
                    const bool actA = tA < total, actB = tB < total;
                    const int rowA = actA ? base + __popcll(mA & le) - 1 : 0;
                    base += __popcll(mA);
                    const int rowB = actB ? base + __popcll(mB & le) - 1 : 0;
                    base += __popcll(mB);
                    int kA, kB; double xA, xB;
                    {
                        const int jj = tA - L.prefix[rowA], ca = L.cntA[rowA];
                        kA = (jj < ca) ? (L.firstA[rowA] + jj) : (L.firstB[rowA] + (jj - ca));
                        xA = fold_dphi(__builtin_fma((double)kA, L.dphi[rowA], L.xoff[rowA]));
                    }
                    {
                        const int jj = tB - L.prefix[rowB], ca = L.cntA[rowB];
                        kB = (jj < ca) ? (L.firstA[rowB] + jj) : (L.firstB[rowB] + (jj - ca));
                        xB = fold_dphi(__builtin_fma((double)kB, L.dphi[rowB], L.xoff[rowB]));
                    }
                    const PairH &hA = L.pair[L.eslot[rowA]], &hB = L.pair[L.eslot[rowB]];
                    double vA[3], vB[3];
                    bool okA, okB;
                    const bool small = (!actA || fabs(xA) <= 0.5) && (!actB || fabs(xB) <= 0.5);
#if BFGX_ABLATE == 3 || BFGX_ABLATE == 6 || BFGX_ABLATE == 7     // timing-only build: no per-pair math at all
                    if (true) { okA = okB = true; vA[0] = vA[1] = vA[2] = xA; vB[0] = vB[1] = vB[2] = xB; } else
#endif
                    if (fastok && __all(small)) {
                        if (MODE == MODE_OFFSETS && sizeof(ACC) == 4 && BFGX_PAIR32) {
                            okA = pair_value_fast32<NC>(pt, hA, L.z[rowA], L.sth[rowA], xA, vA);
                            okB = pair_value_fast32<NC>(pt, hB, L.z[rowB], L.sth[rowB], xB, vB);
                        } else {
                            okA = pair_value_fast<MODE, NC>(pt, hA, L.z[rowA], L.sth[rowA], xA, vA);
                            okB = pair_value_fast<MODE, NC>(pt, hB, L.z[rowB], L.sth[rowB], xB, vB);
                        }
                    } else {
                        // generic path (non-uniform ln r axis or a wide azimuth span): one pair per pass of a
                        // deliberately rolled loop so that its code and registers exist only once
                        okA = okB = false;
#pragma unroll 1
                        for (int pass = 0; pass < 2; ++pass) {
                            const PairH &hh = pass ? hB : hA;
                            const int rw = pass ? rowB : rowA;
                            double vv[3] = {0.0, 0.0, 0.0};
                            const bool ok = pair_value<MODE, NC>(pt, hh, L.z[rw], L.sth[rw], (pass ? xB : xA) + hh.phi0, vv);
                            if (pass) { okB = ok; vB[0] = vv[0]; vB[1] = vv[1]; vB[2] = vv[2]; }
                            else { okA = ok; vA[0] = vv[0]; vA[1] = vv[1]; vA[2] = vv[2]; }
                        }
                    }
#if BFGX_ABLATE == 7      // timing-only: plain LDS stores instead of atomics
                    if (actA && okA) { double *o = acc + NCOMP * (L.ldsbase[rowA] + kA); for (int cc = 0; cc < NCOMP; ++cc) o[cc] = vA[cc]; }
                    if (actB && okB) { double *o = acc + NCOMP * (L.ldsbase[rowB] + kB); for (int cc = 0; cc < NCOMP; ++cc) o[cc] = vB[cc]; }
#elif BFGX_ABLATE == 2 || BFGX_ABLATE == 6     // timing-only build: no LDS accumulation (keep the values alive)
                    if (actA && okA && vA[0] == 1.2345e300) acc[0] = vA[1] + vA[2];
                    if (actB && okB && vB[0] == 1.2345e300) acc[1] = vB[1] + vB[2];
#else
                    if (actA && okA) {
                        double *o = acc + NCOMP * (L.ldsbase[rowA] + kA);
                        for (int cc = 0; cc < NCOMP; ++cc) atomicAdd(o + cc, vA[cc]);           // ds_add_f64
                    }
                    if (actB && okB) {
                        double *o = acc + NCOMP * (L.ldsbase[rowB] + kB);
                        for (int cc = 0; cc < NCOMP; ++cc) atomicAdd(o + cc, vB[cc]);
                    }
#endif
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    __syncthreads();

    if (MODE == MODE_COUNT) {
        if (lane == 0 && npairs) atomicAdd(pair_total, npairs);
        continue;
    }
    // ---- flush: every pixel of the tile is stored exactly once (plain, row-contiguous stores)
    float mcomp = 0.0f;                                // largest |component| stored (by-product for the regrid, K2)
    for (int rr = wid; rr < i1 - i0; rr += kWavesPerBlock) {
        const int ring = i0 + rr;
        int64_t st, n64; bool shf;
        ring_info_small(h, ring, st, n64, shf);
        const int nr = (int)n64;
        const int ks = tile_ks(tj, nr, nphi), ke = tile_ks(tj + 1, nr, nphi);
        const int n = (ke - ks) * NCOMP;
        ACC *dst = out + NCOMP * (st + ks);
        const double *src = acc + NCOMP * rr * T.W;
        if (addmode) { for (int x = lane; x < n; x += kWave) { const ACC v = (ACC)((double)dst[x] + src[x]); dst[x] = v; mcomp = fmaxf(mcomp, fabsf((float)v)); } }
        else { for (int x = lane; x < n; x += kWave) { const ACC v = (ACC)src[x]; dst[x] = v; mcomp = fmaxf(mcomp, fabsf((float)v)); } }
    }
    if (MODE == MODE_OFFSETS && omax2 != nullptr) {
        // largest |offset|^2 of the tile as float bits, bounded by 3 max|component|^2 (zeroed with the binning counters)
#pragma unroll
        for (int sft = kWave >> 1; sft > 0; sft >>= 1) mcomp = fmaxf(mcomp, __shfl_down(mcomp, sft, kWave));
        if (lane == 0 && mcomp > 0.0f) atomicMax(omax2 + tile, __float_as_uint(3.0f * mcomp * mcomp * 1.0001f));
    }
    __syncthreads();                                   // the LDS tile is reused by the next visit
    }
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

// colatitude of a ring centre exactly as get_interpol uses it (healpix_cxx get_ring_info2)
__device__ inline double ring_theta(const Hpx &h, int ring)
{
    int64_t st, nr; double th; bool sh;
    ring_info2(h, ring, st, nr, th, sh);
    return th;
}

// ---------------------------------------------------------------------------------- K2, tiled (bfgx_regrid2.hpp)
// one ring of the tiled regrid's ring table: everything the per-pixel code
// needs of a ring, so that no pixel re-derives ring geometry (integer divisions, square roots, 2 pi / n)
struct RegRow {
    double theta, z, sth;                 // colatitude (-inf / +inf beyond the poles), cos, sin
    double dphi, inv_dphi;                // 2 pi / nr and nr / (2 pi)
    double c0, s0;                        // cos / sin of the azimuth of the tile's first pixel (ks) in this ring
    int64_t start;
    int32_t nr, ks, ke, shf;              // nr == 0: no such ring
    double lim2;                          // gathering regrid: a pixel of this ring is gathered iff |offset|^2 < lim2
};

// colatitude of a ring centre without libm (atan2 of the ring's sin/cos), for the tiled regrid
__device__ inline double ring_theta_nolibm(const Hpx &h, int ring)
{
    double z, sth;
    ring_z_sth(h, ring, z, sth);
    return atan2_generic(sth, z);
}

// sums[0] = sum of tile_sums[2 t], sums[1] = sum of tile_sums[2 t + 1]   (one workgroup)
__global__ void __launch_bounds__(256)
sum_tiles_kernel(int ntiles, const double *__restrict__ tile_sums, double *__restrict__ sums)
{
    __shared__ double sa[256 / kWave], sb[256 / kWave];
    double xa = 0.0, xb = 0.0;
    for (int t = threadIdx.x; t < ntiles; t += 256) { xa += tile_sums[2 * t]; xb += tile_sums[2 * t + 1]; }
#pragma unroll
    for (int s = kWave >> 1; s > 0; s >>= 1) { xa += __shfl_down(xa, s, kWave); xb += __shfl_down(xb, s, kWave); }
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) { sa[wid] = xa; sb[wid] = xb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) { ta += sa[w]; tb += sb[w]; }
        sums[0] = ta; sums[1] = tb;
    }
}

// the same over many tiles (NSIDE >= 4096: 1e5 .. 4e5 tiles, 0.6 ms for one workgroup): sums[0 .. 1] += the workgroup's part (sums zeroed
// by the caller; one atomic pair per workgroup)
__global__ void __launch_bounds__(256)
sum_tiles_multi_kernel(int ntiles, const double *__restrict__ tile_sums, double *__restrict__ sums)
{
    __shared__ double sa[256 / kWave], sb[256 / kWave];
    double xa = 0.0, xb = 0.0;
    const double2 *ts = reinterpret_cast<const double2 *>(tile_sums);
    for (int t = blockIdx.x * 256 + threadIdx.x; t < ntiles; t += gridDim.x * 256) { const double2 v = ts[t]; xa += v.x; xb += v.y; }
#pragma unroll
    for (int s = kWave >> 1; s > 0; s >>= 1) { xa += __shfl_down(xa, s, kWave); xb += __shfl_down(xb, s, kWave); }
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) { sa[wid] = xa; sb[wid] = xb; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) { ta += sa[w]; tb += sb[w]; }
        atomicAdd(sums + 0, ta); atomicAdd(sums + 1, tb);
    }
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
