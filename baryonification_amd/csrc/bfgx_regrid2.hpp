// K2 for gfx950 (MI355X): the post-loop regrid of BaryonifyShell (HealpixRunner.py:333-341 + regrid_pixels_hpix :60-64).
//
// Per-pixel arithmetic.  The reference turns pix2vec + offset into (theta, phi) with vec2ang and then asks
// get_interp_weights; both steps move O(1) numbers although the displacement is ~1e-5 rad.  Here the displaced position is
// carried as the pixel's OWN (ring, k) plus two small angles, formed from the offset's components in the pixel's local frame
//     a = o.e_rho,  b = o.e_phi,  o_z      (e_rho, e_phi: in-plane radial / azimuthal unit vectors at the pixel)
//     dphi   = atan(b / (sth + a))
//     dtheta = asin( [ (a z - o_z sth) + (sth + a) z (sqrt(1 + t^2) - 1) ] / |v + o| ),   t = b / (sth + a)
// in which nothing cancels, so the pair runs in the precision of pix_offsets (fp32 by default; fp64 keeps 1e-10 parity):
// the ring above / below and the colatitude weight follow from dtheta and the ring spacing, the azimuth weights on the
// pixel's own ring from dphi / (2 pi / nr), and on the neighbouring ring from the exact rational (k + shift) nr' / nr (the
// only fp64 arithmetic left per pixel).  Pixels next to a pole and displacements beyond one ring / 3 columns / 0.02 rad take
// the generic fp64 evaluation (healpix_cxx get_interpol on the displaced vector).
//
// Ownership.  Every OUTPUT pixel belongs to one workgroup (tile_regrid3_kernel below): no global atomics, no zero-fill.
#pragma once
#include "bfgx_kernels.hpp"
#include "bfgx_scatter2.hpp"

namespace bfgx {

// fp32 / fp64 view of one ring of the regrid window: what the fast per-pixel code needs
template <typename real>
struct alignas(16) RegRowC {
    real z, sth, dphi, inv_dphi;          // cos / sin of the colatitude, 2 pi / nr, nr / (2 pi)
    real c0, s0;                          // cos / sin of the azimuth of the tile's first pixel (ks) in this ring
    real inv_dth_up, inv_dth_dn;          // 1 / (theta_r - theta_{r-1}), 1 / (theta_{r+1} - theta_r); 0 where there is no such ring
};
// fp64 geometry reads the first six from the fp64 ring table itself (RegRow): only the two spacings are kept beside it -- 3 KB of LDS less per
// workgroup, which is what lets FOUR workgroups of the walking kernel share a CU with split fp32 pix_offsets (42.9 -> 39.7 KB each; three: K2 0.62 ms)
template <>
struct alignas(16) RegRowC<double> { double inv_dth_up, inv_dth_dn; };
// what the per-pixel code needs of one ring, in the precision of the geometry
template <typename real> struct RowGeom { real z, sth, dphi, inv_dphi, c0, s0, inv_dth_up, inv_dth_dn; };
__device__ inline RowGeom<float> row_geom(const RegRow *, const RegRowC<float> *rowc, int t)
{
    const RegRowC<float> c = rowc[t];
    return RowGeom<float>{c.z, c.sth, c.dphi, c.inv_dphi, c.c0, c.s0, c.inv_dth_up, c.inv_dth_dn};
}
__device__ inline RowGeom<double> row_geom(const RegRow *rows, const RegRowC<double> *rowc, int t)
{
    const RegRow &r = rows[t];
    const RegRowC<double> c = rowc[t];
    return RowGeom<double>{r.z, r.sth, r.dphi, r.inv_dphi, r.c0, r.s0, c.inv_dth_up, c.inv_dth_dn};
}
__device__ inline float row_inv_dphi(const RegRow *, const RegRowC<float> *rowc, int t) { return rowc[t].inv_dphi; }
__device__ inline double row_inv_dphi(const RegRow *rows, const RegRowC<double> *, int t) { return rows[t].inv_dphi; }
__host__ __device__ inline size_t regrowc_bytes(size_t real_size) { return real_size == 4 ? sizeof(RegRowC<float>) : sizeof(RegRowC<double>); }

// the 4 targets of one displaced pixel by the generic route: get_interpol (healpix_cxx) on (theta, phi) of v + o, fp64
__device__ inline void regrid_targets_generic(const Hpx &h, const RegRow *rows, int LR, int rth0, int ti, int x,
                                                    double o0, double o1, double o2, int tr[4], int tk[4], double w[4])
{
    const int nl4 = (int)(4 * h.nside);
    const RegRow &rw = rows[ti];
    const double z = rw.z, sth = rw.sth;
    const double phi = ((double)(rw.ks + x) + (rw.shf ? 0.5 : 0.0)) * rw.dphi;
    double s, c;
    sincos_bounded(phi, s, c);
    const double nx = sth * c + o0, ny = sth * s + o1, nz = z + o2;                  // HealpixRunner.py:333
    const double xr = nx * c + ny * s, yr = ny * c - nx * s;
    const double inv = fast_rsq(nx * nx + ny * ny + nz * nz);
    const double zc = nz * inv;
    const double tq = yr * fast_rcp(xr);
    const double sn = fast_sqrt(xr * xr + yr * yr) * inv;
    const double q = sn * z - zc * sth;
    double theta, ph;
    if (xr > 0.0 && fabs(tq) <= 0.1 && fabs(q) <= 0.05) {
        ph = phi + atan_small(tq);
        theta = rw.theta + asin_small(q);
    } else {
        theta = atan2_generic(sn, zc);
        ph = atan2_generic(ny, nx);
    }
    if (ph < 0) ph += kTwoPi;
    if (ph >= kTwoPi) ph -= kTwoPi;
    int t1 = ti;
    while (t1 > 0 && theta < rows[t1].theta) --t1;
    while (t1 < LR + 1 && theta >= rows[t1 + 1].theta) ++t1;
    const bool in_window = (theta >= rows[t1].theta) && (t1 < LR + 1) && (theta < rows[t1 + 1].theta);
    const int ir1 = in_window ? rth0 + t1 : (int)ring_above(h, zc);
    const int ir2 = ir1 + 1;
    double theta1 = 0.0, theta2 = 0.0;
    for (int q4 = 0; q4 < 4; ++q4) { tr[q4] = 0; tk[q4] = 0; w[q4] = 0.0; }
#pragma unroll
    for (int half = 0; half < 2; ++half) {
        const int ir = half ? ir2 : ir1;
        if (half ? (ir < nl4) : (ir > 0)) {
            const int t2 = ir - rth0;
            int n2; double shd, invd, th;
            if (t2 >= 0 && t2 < LR + 2) {
                const RegRow &r2 = rows[t2];
                n2 = r2.nr; shd = r2.shf ? 0.5 : 0.0; invd = r2.inv_dphi; th = r2.theta;
            } else {
                n2 = ring_len(h, ir);
                const bool sh2 = (ir < h.nside) || (ir >= 3 * h.nside) || (((ir - (int)h.nside) & 1) == 0);
                shd = sh2 ? 0.5 : 0.0; invd = (double)n2 * kInvTwoPi; th = ring_theta_nolibm(h, ir);
            }
            const double tmp = ph * invd - shd;
            int j1 = (int)floor(tmp);
            const double w1 = tmp - (double)j1;
            int j2 = j1 + 1;
            if (j1 < 0) j1 += n2;
            if (j2 >= n2) j2 -= n2;
            tr[2 * half] = ir; tr[2 * half + 1] = ir;
            tk[2 * half] = j1; tk[2 * half + 1] = j2;
            w[2 * half] = 1.0 - w1; w[2 * half + 1] = w1;
            if (half) theta2 = th; else theta1 = th;
        }
    }
    if (ir1 == 0) {                                    // north pole: missing ring -> the 4 polar pixels
        const double wtheta = theta / theta2;
        w[2] *= wtheta; w[3] *= wtheta;
        const double fac = (1.0 - wtheta) * 0.25;
        w[0] = fac; w[1] = fac; w[2] += fac; w[3] += fac;
        tr[0] = 1; tr[1] = 1; tk[0] = (tk[2] + 2) & 3; tk[1] = (tk[3] + 2) & 3;
    } else if (ir2 == nl4) {                           // south pole
        const double wtheta = (theta - theta1) / (kPi - theta1);
        w[0] *= (1.0 - wtheta); w[1] *= (1.0 - wtheta);
        const double fac = wtheta * 0.25;
        w[0] += fac; w[1] += fac; w[2] = fac; w[3] = fac;
        tr[2] = nl4 - 1; tr[3] = nl4 - 1; tk[2] = (tk[0] + 2) & 3; tk[3] = (tk[1] + 2) & 3;
    } else {
        const double wtheta = (theta - theta1) * fast_rcp(theta2 - theta1);
        w[0] *= (1.0 - wtheta); w[1] *= (1.0 - wtheta);
        w[2] *= wtheta; w[3] *= wtheta;
    }
}

__device__ inline float abs_(float v) { return __builtin_fabsf(v); }
__device__ inline double abs_(double v) { return __builtin_fabs(v); }
// small-angle pieces of the displaced position, valid for |o| <= 0.02, |t| <= 0.1: fp32 by series (truncation below the
// fp32 rounding), fp64 through the fp64 helpers (1e-16)
template <typename real> struct RMath;
template <> struct RMath<float> {
    static __device__ inline float rcp(float x) { const float y = __builtin_amdgcn_rcpf(x); return y * __builtin_fmaf(-x, y, 2.0f); }
    // 1 / sqrt(1 + e) = 1 - e/2 + 3 e^2/8 - 5 e^3/16 + 35 e^4/128
    static __device__ inline float inv_norm(float e) { return fma_(e, fma_(e, fma_(e, fma_(e, 0.2734375f, -0.3125f), 0.375f), -0.5f), 1.0f); }
    // sqrt(1 + t^2) - 1 = t^2/2 - t^4/8 + t^6/16
    static __device__ inline float sqrt1pm1(float t2) { return t2 * fma_(t2, fma_(t2, 0.0625f, -0.125f), 0.5f); }
    static __device__ inline float asin_(float q) { const float q2 = q * q; return q * fma_(q2, fma_(q2, fma_(q2, (float)(15.0 / 336.0), 0.075f), (float)(1.0 / 6.0)), 1.0f); }
    static __device__ inline float atan_(float t) { const float t2 = t * t; return t * fma_(t2, fma_(t2, fma_(t2, fma_(t2, (float)(1.0 / 9.0), (float)(-1.0 / 7.0)), 0.2f), (float)(-1.0 / 3.0)), 1.0f); }
};
template <> struct RMath<double> {
    static __device__ inline double rcp(double x) { return fast_rcp(x); }
    static __device__ inline double inv_norm(double e) { return fast_rsq(1.0 + e); }
    static __device__ inline double sqrt1pm1(double t2) { const double h = 1.0 + t2; return t2 * fast_rcp(1.0 + h * fast_rsq(h)); }
    static __device__ inline double asin_(double q) { return asin_small(q); }
    static __device__ inline double atan_(double t) { return atan_small(t); }
};

// the parity-grade mode (SPLIT pix_offsets): the same small-angle pieces to ~1e-11 -- fp32 hardware seeds + one Newton step, series cut at the
// bounds of a gathered pixel (|e| <= 0.045, t^2 <= 0.01, |q| <= 0.021, |t| <= 0.1).  A displaced position is needed to ~1e-7 of a pixel side at
// displacements of up to 20: 5e-9 of the offset.
struct RMathE {
    static __device__ inline double rcp(double x) { const double y = (double)__builtin_amdgcn_rcpf((float)x); return __builtin_fma(y, __builtin_fma(-x, y, 1.0), y); }
    static __device__ inline double inv_norm(double e)
    {
        const double x = 1.0 + e, y = (double)__builtin_amdgcn_rsqf((float)x);
        const double h = 0.5 * y, r = __builtin_fma(-x * y, h, 0.5);
        return __builtin_fma(y, r, y);
    }
    // sqrt(1 + t^2) - 1 = t^2/2 - t^4/8 + t^6/16 - 5 t^8/128 + 7 t^10/256
    static __device__ inline double sqrt1pm1(double t2) { return t2 * __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(t2, 7.0 / 256.0, -5.0 / 128.0), 0.0625), -0.125), 0.5); }
    static __device__ inline double asin_(double q) { const double q2 = q * q; return q * __builtin_fma(q2, __builtin_fma(q2, __builtin_fma(q2, 15.0 / 336.0, 0.075), 1.0 / 6.0), 1.0); }
    static __device__ inline double atan_(double t)
    {
        const double t2 = t * t;
        return t * __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(t2, __builtin_fma(t2, -1.0 / 11.0, 1.0 / 9.0), -1.0 / 7.0), 0.2), -1.0 / 3.0), 1.0);
    }
};

// ---------------------------------------------------------------------------------- K2, gathering form (full-map regrid)
// Every OUTPUT pixel is the property of one workgroup: a tile evaluates the displaced position of its own pixels AND of the
// pixels in an apron around it, keeps only the deposits that land inside the tile (LDS), and stores the tile once with plain
// stores -- no global atomics, no zero-fill (the scatter form ends in ~1.3 global fp64 atomics per map pixel, which is what
// bounds it: the memory-side atomic rate).
//
// Which pixels are gathered is a property of the SOURCE pixel alone, decided by arithmetic every tile repeats bit for bit:
//     gathered  <=>  |o|^2 < lim(ring)^2,   lim = min(cap, 0.09 sin(theta_ring), 0.999 x the distance to the first / last ring)
// (cap = regrid_cap(nside), at most kReachMax rings); the move is then small enough for the small-angle formulas (|t| < 0.1,
// |q| < 0.021) and never crosses a pole.  Everything else (huge displacements, the pixels next to a pole) is appended by the
// tile that OWNS the source pixel to a global list which a small fix-up kernel adds to the stored map afterwards; if that
// list overflows, a second pass over the owners (regrid_far_pass_kernel) applies those deposits with atomics instead.
//
// The apron of a tile follows the data: tile_reach_kernel leaves the largest |o| of every tile, a tile takes the maximum m
// over the tiles around it and evaluates R = regrid_reach_rings(m) rings and K columns beyond its own pixels -- 1 ring and
// 4 columns (1.2x the evaluations) for sub-pixel displacements, more only where the map really moves that far.
#ifndef BFGX_ABLK2
#define BFGX_ABLK2 0                // timing-only ablation builds of the walking kernel (scripts/k2_variants.sh)
#endif
#ifndef BFGX_K2U
#define BFGX_K2U 2
#endif
#ifndef BFGX_K2LEAN_U
#define BFGX_K2LEAN_U 2          // window pixels per lane and trip of the lean / repair kernels
#endif
constexpr int kReachMax = 16;       // most rings a gathered deposit travels; the ring tables hold BR + 2 kReachMax + 2 rings
constexpr int kReachColsMax = 256;  // most apron columns per side

__host__ __device__ inline double regrid_cap(int64_t nside)
{
    const double c = 9.9 / (double)nside;                // 9.9 / nside radians = 14.85 ring spacings at most (below)
    return c < 0.02 ? c : 0.02;
}
// rings a displacement of m radians can cross: neighbouring rings are at least 2 / (3 nside) apart everywhere
__host__ __device__ inline int regrid_reach_rings(int64_t nside, double m)
{
    const double r = m * 1.5 * (double)nside;
    const int R = (r < (double)kReachMax ? (int)r : kReachMax) + 1;
    return R > kReachMax ? kReachMax : R;
}
// the largest displacement a fixed apron of R rings serves (banded regrid: every rank uses the same R)
__host__ __device__ inline double regrid_cap_for_rings(int64_t nside, int R)
{
    const double c = 0.999 * (double)R / (1.5 * (double)nside), cap = regrid_cap(nside);
    return (R >= kReachMax || c > cap) ? cap : c;
}

// what the scan of the walking kernel needs of one ring of the window (one 32-byte LDS read)
struct alignas(16) RowScan {
    int32_t start, nr, ks, span;          // first pixel of the ring, its length, the tile's first column and width in it (nr == 0: no such ring)
    float need2, cf2, lim2;               // thresholds on |o|^2: (colatitude to cross)^2, (columns per radian moved)^2, gathered iff |o|^2 < lim2
    int32_t klr;                          // columns scanned left | right << 10 of the span, | 1 << 20 for the tile's own rings
};

struct FarList {
    unsigned long long *count;    // entries appended (may exceed cap: overflow)
    int64_t *pix;
    double *val;
    int64_t cap;
    int32_t *overflow;
};

struct ReachArgs {
    const int32_t *apron;         // [ntiles][2]: rings / columns of apron of every tile (tile_apron_kernel)
    double cap;                   // no pixel with |o| >= cap is gathered
    double theta_first, theta_last;   // colatitude of the first / last ring
};

// the walking kernel (PASS 2) compacts its source pixels: per wave a queue of kWalkQ survivors {position, offset, value}, per
// wave a second queue for its own pixels that take the generic route, per ring of the window two thresholds of the scan
constexpr int kWalkQ = 128;         // entries per wave: a wave pushes at most 64 onto fewer than 64
constexpr int kWalkFar = (256 / kWave) * kWalkQ;      // per wave a second queue of kWalkQ positions: its own pixels that take the generic route
// real_size: the type of the per-pixel geometry (ring tables); acc_size: the type pix_offsets are stored in (the walking kernel's queue holds
// the offsets as they were read: with split fp32 pix_offsets + fp64 geometry the low halves are fetched when a survivor is evaluated)
__host__ __device__ inline size_t regrid3_lds_bytes(int BR, int W, size_t real_size, bool walk = false, size_t acc_size = 0)
{
    if (acc_size == 0) acc_size = real_size;
    const size_t base = (size_t)BR * W * sizeof(double) + (size_t)(BR + 2 * kReachMax + 2) * (sizeof(RegRow) + regrowc_bytes(real_size)) + 16;
    if (!walk) return base;
    return base + (size_t)(BR + 2 * kReachMax + 2) * sizeof(RowScan) + (size_t)(256 / kWave) * kWalkQ * (sizeof(double) + sizeof(int32_t) + 3 * acc_size)
           + (size_t)(kWalkFar + 4) * sizeof(int32_t);
}

// largest |o|^2 of the pixels of every tile (offsets that did not come from this plan's K1, which leaves it as a by-product)
template <typename ACC>
__global__ void __launch_bounds__(256)
tile_reach_kernel(Hpx h, Tiling T, const ACC *__restrict__ offsets, float *__restrict__ tile_omax)
{
    __shared__ float red[256 / kWave];
    const int tile = blockIdx.x;
    const int band = T.tile_band[tile];
    const int nphi = T.band_nphi[band];
    const int tj = tile - T.band_tile0[band];
    const int nl4 = (int)(4 * h.nside);
    const int i0 = 1 + band * T.BR, i1 = min(i0 + T.BR, nl4);
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    float m2 = 0.0f;
    for (int ring = i0 + wid; ring < i1; ring += 256 / kWave) {
        int64_t st, nr64; bool shf;
        ring_info_small(h, ring, st, nr64, shf);
        const int nr = (int)nr64, ks = tile_ks(tj, nr, nphi), ke = tile_ks(tj + 1, nr, nphi);
        const ACC *o = offsets + 3 * (st + ks);
        for (int x = lane; x < ke - ks; x += kWave) {
            const float a = (float)o[3 * x], b = (float)o[3 * x + 1], c = (float)o[3 * x + 2];
            m2 = fmaxf(m2, fma_(a, a, fma_(b, b, c * c)));
        }
    }
#pragma unroll
    for (int sft = kWave >> 1; sft > 0; sft >>= 1) m2 = fmaxf(m2, __shfl_down(m2, sft, kWave));
    if (lane == 0) red[wid] = m2;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 256 / kWave; ++w) m2 = fmaxf(m2, red[w]);
        tile_omax[tile] = m2;
    }
}

// The apron of every tile: R rings and K columns beyond its own pixels, from the largest |o| among the tiles a gathered
// deposit could come from (tile_omax; nullptr: the fixed reach `rings` of the banded regrid, with m = cap).  Tiles whose
// reach exceeds one ring are listed in todo ([0] = count) for the kernel with the ring walk.  One thread per tile.
__global__ void __launch_bounds__(256)
tile_apron_kernel(Hpx h, Tiling T, const float *__restrict__ tile_omax, int rings, double cap, int tile0, int ntiles,
                  int32_t *__restrict__ apron, int32_t *__restrict__ todo, int32_t *__restrict__ lean = nullptr)
{
    // (lean != nullptr: the tiles whose reach is at most one ring are listed too -- [0] = count, zeroed by the regrid's last launch --, so that
    // the lean kernel is a small persistent grid over THAT list and costs nothing where every tile walks: 6208 workgroups of 40 KB that
    // return at once were 16 us per step on the S19 table)
    const int i = blockIdx.x * 256 + threadIdx.x;
    bool is_lean = false;
    if (i < ntiles) {
    const int tile = tile0 + i;
    const int band = T.tile_band[tile];
    const int nphi = T.band_nphi[band];
    const int tj = tile - T.band_tile0[band];
    const int nl4 = (int)(4 * h.nside);
    float m = (float)cap;
    int R = rings;
    if (tile_omax != nullptr) {
        m = 0.0f;
        const int nb = (kReachMax + T.BR - 1) / T.BR;
        for (int db = -nb; db <= nb; ++db) {
            const int b2 = band + db;
            if (b2 < 0 || b2 >= T.nbands) continue;
            const int n2 = T.band_nphi[b2], t20 = T.band_tile0[b2];
            // tiles of band b2 that overlap this tile's azimuth range, one more either side (all of them for narrow tiles)
            int lo = (int)(((int64_t)tj * n2) / nphi) - 1, hi = (int)(((int64_t)(tj + 1) * n2 + nphi - 1) / nphi) + 1;
            if (T.W < 32 || hi - lo >= n2) { lo = 0; hi = n2; }
            for (int t2 = lo; t2 < hi; ++t2) {
                const int u = t2 < 0 ? t2 + n2 : (t2 >= n2 ? t2 - n2 : t2);
                m = fmaxf(m, tile_omax[t20 + u]);
            }
        }
        const bool still = !(m > 0.0f);                                  // no offset in the tile nor within reach of it
        m = fminf(__builtin_sqrtf(m) * 1.001f, (float)cap);             // (tile_omax holds |o|^2)
        R = still ? 0 : regrid_reach_rings(h.nside, (double)m);          // R = 0: the regrid of this tile is a copy
    }
    // apron columns: a source pixel c columns outside the tile on its ring reaches it only if
    // c <= 1.5 + (its move in columns of its own ring) + nr_source / nr_target.  The move in columns, lim / (sth - lim) nr / 2 pi,
    // grows towards the ring where the polar cap meets the equatorial belt (ring nside / 3 nside) on either side of it, and
    // the ring length grows towards the belt: the first and last ring of the window and those two rings cover the extremes.
    const int i0 = 1 + band * T.BR, i1 = min(i0 + T.BR, nl4);
    const int rlo = max(i0 - R, 1), rhi = min(i1 + R, nl4) - 1, ns = (int)h.nside;
    int nmin = 0x7fffffff, nmax = 0;
    float mc = 0.0f;
    const int probe[4] = {rlo, rhi, ns, 3 * ns};
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int ring = probe[q];
        if (ring < rlo || ring > rhi) continue;
        double z, sth;
        ring_z_sth(h, ring, z, sth);
        const int nr = ring_len(h, ring);
        nmin = min(nmin, nr); nmax = max(nmax, nr);
        const float limf = fminf(m, (float)(0.09 * sth)), den = (float)sth - limf;
        if (den > 0.0f) mc = fmaxf(mc, limf / den * 1.02f * (float)nr * (float)kInvTwoPi);
    }
    const float kk = mc + (float)nmax / (float)nmin + 2.5f;
    apron[2 * tile] = R;
    apron[2 * tile + 1] = (nphi == 1) ? 0 : (kk < (float)kReachColsMax ? (int)kk + 1 : kReachColsMax);
    if (todo != nullptr && R > 1) todo[1 + atomicAdd(todo, 1)] = tile;
    is_lean = R <= 1;
    }
    if (lean != nullptr) {                               // one atomic per wave
        const unsigned long long m = __ballot(is_lean);
        if (m) {
            const int lane = threadIdx.x & (kWave - 1), leader = __ffsll((long long)m) - 1;
            int base = 0;
            if (lane == leader) base = atomicAdd(lean, __popcll(m));
            base = __shfl(base, leader, kWave);
            const int pos = base + __popcll(m & ((1ull << lane) - 1ull));
            if (is_lean && pos < ntiles) lean[1 + pos] = tile0 + i;           // (pos < ntiles always; the bound keeps a stale count inside the list)
        }
    }
}

// largest |o|^2 of n pixels as the bits of a float (non-negative floats order like unsigned integers); *out zeroed by the caller
template <typename ACC>
__global__ void __launch_bounds__(256)
max_offset_kernel(int64_t n, const ACC *__restrict__ offsets, unsigned *__restrict__ out)
{
    float m2 = 0.0f;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const float a = (float)offsets[3 * i], b = (float)offsets[3 * i + 1], c = (float)offsets[3 * i + 2];
        m2 = fmaxf(m2, fma_(a, a, fma_(b, b, c * c)));
    }
#pragma unroll
    for (int sft = kWave >> 1; sft > 0; sft >>= 1) m2 = fmaxf(m2, __shfl_down(m2, sft, kWave));
    // one atomic per workgroup (the counter is ONE address: atomics on it serialise)
    __shared__ float wm[256 / kWave];
    if ((threadIdx.x & (kWave - 1)) == 0) wm[threadIdx.x / kWave] = m2;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = 0.0f;
        for (int w = 0; w < 256 / kWave; ++w) m = fmaxf(m, wm[w]);
        if (m > 0.0f) atomicMax(out, __float_as_uint(m));
    }
}

// largest of n non-negative floats given as their bit patterns (the per-tile maxima K1 leaves): one workgroup
__global__ void __launch_bounds__(1024)
max_bits_kernel(int n, const unsigned *__restrict__ v, unsigned *__restrict__ out)
{
    unsigned m = 0u;
    for (int i = threadIdx.x; i < n; i += 1024) m = max(m, v[i]);
#pragma unroll
    for (int sft = kWave >> 1; sft > 0; sft >>= 1) m = max(m, (unsigned)__shfl_down((int)m, sft, kWave));
    if ((threadIdx.x & (kWave - 1)) == 0 && m) atomicMax(out, m);
}

// One source pixel of the gathering regrid: the 4 targets of its displaced position, (ring index in the tile's ring tables,
// column) + weight.  The pixel must be a gathered one (see above).  Returns false
// when the move leaves the ring tables (then no target lies in the tile).  x = column of the pixel relative to the tile's
// first pixel of its ring (negative / beyond the tile for apron pixels).
template <typename real, bool WALK, bool ECON = false>
__device__ inline bool regrid_gather_targets(const RegRow *rows, const RegRowC<real> *rowc, int NT, int ti, int x,
                                             real o0, real o1, real o2, int tt[4], int tk[4], real w[4])
{
    using RM = typename std::conditional<ECON, RMathE, RMath<real>>::type;
    using PMx = typename std::conditional<ECON, PMathE, PMath<real>>::type;
    const RowGeom<real> rc = row_geom(rows, rowc, ti);
    const RegRow &rw = rows[ti];
    // cos / sin of the pixel's azimuth: rotation of the tile's first pixel by x dphi (series), or Cody-Waite in fp64
    // where the tile spans more than 0.45 rad (polar bands)
    real c, s;
    const real alpha = (real)x * rc.dphi;
    if (abs_(alpha) <= (real)0.45) {
        real sa, oma;
        PMx::sin_omc(alpha, sa, oma);
        c = rc.c0 - (rc.c0 * oma + rc.s0 * sa);
        s = rc.s0 - (rc.s0 * oma - rc.c0 * sa);
    } else {
        double s64, c64;
        sincos_bounded(((double)(rw.ks + x) + (rw.shf ? 0.5 : 0.0)) * rw.dphi, s64, c64);
        c = (real)c64; s = (real)s64;
    }
    const real a = o0 * c + o1 * s, b = o1 * c - o0 * s;               // in-plane radial / azimuthal components of the offset
    const real xr = rc.sth + a;
    const real t = b * RM::rcp(xr);
    const real t2 = t * t;
    const real e = (real)2 * (rc.sth * a + rc.z * o2) + (a * a + b * b + o2 * o2);       // |v + o|^2 = 1 + e
    const real invn = RM::inv_norm(e);
    const real q = (fma_(a, rc.z, -(o2 * rc.sth)) + xr * rc.z * RM::sqrt1pm1(t2)) * invn;      // sin(theta_new - theta)
    const real dth = RM::asin_(q);
    const real dph = RM::atan_(t);
    // ring above (t1) / below (t1 + 1) the new colatitude and the weight of the lower one
    const bool down = dth >= (real)0;
    const real wq = down ? dth * rc.inv_dth_dn : -dth * rc.inv_dth_up;      // fraction of the ring spacing moved
    int t1 = ti;
    real wtheta;
    const bool near = wq < (real)1 && (down ? rc.inv_dth_dn : rc.inv_dth_up) > (real)0;
    if (!WALK || near) {                                                    // (!WALK: the tile's reach is one ring, the move is shorter)
        wtheta = down ? wq : (real)1 - wq;
    } else {                                                                // more than one ring: walk the table
        // (the ring spacing changes by a few per cent over the reach: start from the move in units of the local spacing, then correct)
        const double d = (double)dth, th0 = rw.theta;
        const int stp = (int)(wq < (real)(2 * kReachMax) ? wq : (real)(2 * kReachMax));
        t1 = down ? ti + stp : ti - 1 - stp;
        t1 = t1 < 0 ? 0 : (t1 > NT - 2 ? NT - 2 : t1);
        while (t1 + 2 < NT && d >= rows[t1 + 1].theta - th0) ++t1;
        while (t1 > 0 && d < rows[t1].theta - th0) --t1;
        const double lo = rows[t1].theta - th0, hi = rows[t1 + 1].theta - th0;
        if (!(d >= lo && d < hi) || rows[t1].nr == 0 || rows[t1 + 1].nr == 0) return false;
        wtheta = (real)((d - lo) * fast_rcp(hi - lo));
    }
    const int nro = rw.nr;
    int kown = rw.ks + x;
    if (kown < 0) kown += nro;
    if (kown >= nro) kown -= nro;
    // columns on two rings X and Y.  Usual case (the move stays within one ring spacing): X = the pixel's own ring, where
    // u = k + dphi / (2 pi / nr), Y = the ring above or below.  After a walk: X = t1, Y = t1 + 1.  On a ring other than
    // the pixel's own  u' = (k + sh) nr' / nr - sh' + dphi / (2 pi / nr'), the first two terms as an exact rational
    // evaluated in fp64 (never an integer for rings of different length or shift).
    auto other_ring = [&](int tr, int &j, real &wj) {
        const RegRow &rn = rows[tr];
        const double B = ((double)kown + (rw.shf ? 0.5 : 0.0)) * ((double)rn.nr * rw.dphi * kInvTwoPi) - (rn.shf ? 0.5 : 0.0);
        const double Bf = floor(B);
        const real wall = (real)(B - Bf) + dph * row_inv_dphi(rows, rowc, tr), f = __builtin_floor(wall);
        wj = wall - f; j = (int)Bf + (int)f;
    };
    const bool nr_ = !WALK || near;
    const bool x_upper = !nr_ || down;                   // X is the upper ring of the pair
    const int tx = nr_ ? ti : t1, ty = nr_ ? (down ? ti + 1 : ti - 1) : t1 + 1;
    int jx, jy; real wx, wy;
    if (nr_) {
        const real wall = dph * rc.inv_dphi, f = __builtin_floor(wall);
        wx = wall - f; jx = kown + (int)f;
    } else {
        other_ring(tx, jx, wx);
    }
    other_ring(ty, jy, wy);
    const int nrx = rows[tx].nr, nry = rows[ty].nr;
    int jx2 = jx + 1, jy2 = jy + 1;
    jx = jx < 0 ? jx + nrx : (jx >= nrx ? jx - nrx : jx);
    jx2 = jx2 < 0 ? jx2 + nrx : (jx2 >= nrx ? jx2 - nrx : jx2);
    jy = jy < 0 ? jy + nry : (jy >= nry ? jy - nry : jy);
    jy2 = jy2 < 0 ? jy2 + nry : (jy2 >= nry ? jy2 - nry : jy2);
    // wtheta = weight of the lower ring of the pair (the order of the 4 deposits does not matter)
    const real wtx = x_upper ? (real)1 - wtheta : wtheta, wty = (real)1 - wtx;
    tt[0] = tx; tt[1] = tx; tk[0] = jx; tk[1] = jx2;
    w[0] = ((real)1 - wx) * wtx; w[1] = wx * wtx;
    tt[2] = ty; tt[3] = ty; tk[2] = jy; tk[3] = jy2;
    w[2] = ((real)1 - wy) * wty; w[3] = wy * wty;
    return true;
}

// one source pixel by the generic route, its four deposits appended to the far list; returns the sum of the deposits.  (Out of line it costs
// the walking kernel 23 spilled registers around the call and 3 % of its time, the lean kernel 20 %: inlined by default.)
#ifndef BFGX_FAR_INLINE
#define BFGX_FAR_INLINE 1
#endif
#if BFGX_FAR_INLINE
__device__ inline double regrid_far_pixel(
#else
__device__ __noinline__ double regrid_far_pixel(
#endif
    const Hpx &h, const RegRow *rows, int LR, int rth0, int ti, int x, double o0, double o1, double o2,
                                                double val, FarList far)
{
    int tr[4], tk[4];
    double w[4], tot = 0.0;
    regrid_targets_generic(h, rows, LR, rth0, ti, x, o0, o1, o2, tr, tk, w);
    // ONE returning atomic per wave reserves the four entries of every active lane (one per deposit would be tens of millions of atomics on
    // a single address where the field moves far: they serialise at ~5 ns each)
    const unsigned long long act = __ballot(1);
    const int rank = (int)__builtin_amdgcn_mbcnt_hi((unsigned)(act >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)act, 0u));
    const int leader = __ffsll((long long)act) - 1;
    unsigned long long base = 0;
    if (rank == 0) base = atomicAdd(far.count, 4ull * (unsigned long long)__popcll(act));
    base = ((unsigned long long)(unsigned)__shfl((int)(base >> 32), leader, kWave) << 32) | (unsigned)__shfl((int)base, leader, kWave);
    for (int q4 = 0; q4 < 4; ++q4) {
        int64_t st_t, nr64; bool sh_t;
        ring_info_small(h, tr[q4], st_t, nr64, sh_t);
        const double v = w[q4] * val;
        const unsigned long long i = base + 4ull * (unsigned long long)rank + (unsigned long long)q4;
        if ((int64_t)i < far.cap) { far.pix[i] = st_t + tk[q4]; far.val[i] = v; } else atomicOr(far.overflow, 1);
        tot += v;
    }
    return tot;
}

// PASS 0: the gathering regrid of the tiles whose reach is ONE ring (no deposit travels further than the next ring: the
// lean code, 4 waves / SIMD); tiles with a longer reach are appended to `todo` and left to PASS 2, the same kernel with the
// ring walk compiled in, which a small persistent grid runs over that list (or over all tiles: banded regrid with a fixed
// reach > 1).  PASS 1: the fall-back for an overflowing far list (every owner applies the deposits of its far pixels with
// global atomics; runs after the map has been stored, does nothing unless the list overflowed).
// SPLIT (the parity-grade mode): pix_offsets are two fp32 arrays, offsets = hi and offsets_lo = (float)(o - hi), indexed alike; the scan decides on
// hi alone (its thresholds are fp32 comparisons anyway), a survivor's low halves are fetched when it is evaluated -- in fp64 (real = double)
template <typename ACC, typename real, int PASS, bool SPLIT = false>
#ifndef BFGX_K2_OCC
#define BFGX_K2_OCC 4             // waves per SIMD the regrid kernels are compiled for.  Measured, config 2 / S19 table (K2 in ms): 3: 0.240 / 0.514, 4: 0.201 / 0.354, 5: 0.228 / 0.412, 6: 0.261
#endif
__global__ void __launch_bounds__(256, BFGX_K2_OCC)
tile_regrid3_kernel(Hpx h, Tiling T, const double *__restrict__ map_in, const ACC *__restrict__ offsets, const ACC *__restrict__ offsets_lo,
                    double *__restrict__ map_out, FarList far, ReachArgs reach, double *__restrict__ tile_sums, int tile_off, int ntiles,
                    int *__restrict__ todo, double *__restrict__ sums_out, int *__restrict__ lean_reset = nullptr)
{
    // map_in, offsets and map_out are indexed by GLOBAL pixel number.  tile_off < 0: all tiles, heavy ones first; tile_off >= 0
    // (a rank that owns a range of bands): tiles tile_off + blockIdx.x, and the caller passes offsets / map_out pointers
    // shifted so that only the pixels this rank holds (its bands + reach.rings rings either side / its bands) are touched.
    static_assert(!SPLIT || (sizeof(ACC) == 4 && sizeof(real) == 8), "split pix_offsets: fp32 halves, fp64 geometry");
    extern __shared__ __align__(16) unsigned char smem[];
    const int NTmax = T.BR + 2 * kReachMax + 2;
    double *acc = reinterpret_cast<double *>(smem);            // [BR][W]: the tile's own pixels
    RegRow *rows = reinterpret_cast<RegRow *>(acc + T.BR * T.W);
    RegRowC<real> *rowc = reinterpret_cast<RegRowC<real> *>(rows + NTmax);
    const int tid = threadIdx.x;
    const int nl4 = (int)(4 * h.nside);
    // (PASS 1 is the regrid's last launch: the lean kernel's list is free again -- its count starts the next regrid at zero; FIRST, before the
    // early return below: a count that is never reset grows past the list)
    if (PASS == 1 && lean_reset != nullptr && blockIdx.x == 0 && tid == 0) *lean_reset = 0;
    if (PASS == 1 && sums_out != nullptr && blockIdx.x == 0) {
        // the two sums of the mass check from the per-tile totals the gather kernels left (the last launch of the regrid: all
        // tiles are done); sums_out[0] = sum of the source values, [1] = sum of the deposits
        double sa = 0.0, sb = 0.0;
        for (int i = tid; i < ntiles; i += 256) { sa += tile_sums[2 * (int64_t)i]; sb += tile_sums[2 * (int64_t)i + 1]; }
#pragma unroll
        for (int sft = kWave >> 1; sft > 0; sft >>= 1) { sa += __shfl_down(sa, sft, kWave); sb += __shfl_down(sb, sft, kWave); }
        double *red = reinterpret_cast<double *>(smem);
        if ((tid & (kWave - 1)) == 0) { red[2 * (tid / kWave)] = sa; red[2 * (tid / kWave) + 1] = sb; }
        __syncthreads();
        if (tid == 0) {
            double ta = 0.0, tb = 0.0;
            for (int w = 0; w < 256 / kWave; ++w) { ta += red[2 * w]; tb += red[2 * w + 1]; }
            sums_out[0] = ta; sums_out[1] = tb;
        }
        __syncthreads();
    }
    if (PASS == 1 && *far.overflow == 0) {                    // the usual case: add the listed far deposits to the stored map
        const unsigned long long n = *far.count;
        for (unsigned long long i = (unsigned long long)blockIdx.x * 256 + tid; i < n; i += (unsigned long long)gridDim.x * 256)
            atomicAdd(map_out + far.pix[i], far.val[i]);
        return;
    }
    // PASS 2 walks the tiles tile_apron_kernel listed in `todo`; PASS 0 (full-map regrid) the tiles of ITS list, a small persistent grid -- or,
    // without a list (banded regrid), one tile per workgroup
    const bool listed = (PASS == 2 || PASS == 0) && todo != nullptr;
    if (listed) ntiles = min(todo[0], ntiles);
    for (int blk = blockIdx.x; blk < ntiles; blk += ((PASS == 0 && !listed) ? ntiles : (int)gridDim.x)) {
    const int tile = listed ? todo[1 + blk] : ((tile_off < 0) ? T.tile_order[blk] : tile_off + blk);
    const int band = T.tile_band[tile];
    const int nphi = T.band_nphi[band];
    const int tj = tile - T.band_tile0[band];
    const int i0 = 1 + band * T.BR;
    const int i1 = min(i0 + T.BR, nl4);                        // exclusive
    double sum_in = 0.0, sum_out = 0.0;                        // mass-conservation sums (HealpixRunner.py:344-345)
    __syncthreads();                                           // the previous tile of this workgroup is done with the tables

    // ---- reach of this tile (tile_apron_kernel): R rings, K columns
    const int R = (PASS == 2) ? reach.apron[2 * tile] : 1;     // (a compile-time constant in the lean kernel)
    const int kap = (PASS == 1 || nphi == 1) ? 0 : reach.apron[2 * tile + 1];
    if (PASS == 0 && reach.apron[2 * tile] > 1) continue;      // (block-uniform) left to the kernel with the ring walk
    if (PASS == 0 && reach.apron[2 * tile] == 0) {
        // a still tile (a sparse catalog leaves most of the sphere alone): nothing moves in it nor into it from within the reach,
        // so its pixels deposit into themselves -- a copy, with the reference's rule that only positive pixels are regridded
        const int lane = tid & (kWave - 1), wid = tid / kWave;
        for (int rr = wid; rr < i1 - i0; rr += 256 / kWave) {
            int64_t st; int64_t nr64; bool shf_;
            ring_info_small(h, i0 + rr, st, nr64, shf_);
            const int ks = tile_ks(tj, (int)nr64, nphi), ke = tile_ks(tj + 1, (int)nr64, nphi);
            for (int k = ks + lane; k < ke; k += kWave) {
                const double v = map_in[st + k];
                const double dep = (v > 0.0) ? v : 0.0;                       // HealpixRunner.py:335
                map_out[st + k] = dep;
                sum_in += v; sum_out += dep;
            }
        }
        if (tile_sums) {
#pragma unroll
            for (int sft = kWave >> 1; sft > 0; sft >>= 1) { sum_in += __shfl_down(sum_in, sft, kWave); sum_out += __shfl_down(sum_out, sft, kWave); }
            if (lane == 0) { acc[2 * wid] = sum_in; acc[2 * wid + 1] = sum_out; }
            __syncthreads();
            if (tid == 0) {
                double sa_ = 0.0, sb_ = 0.0;
                for (int wv = 0; wv < 256 / kWave; ++wv) { sa_ += acc[2 * wv]; sb_ += acc[2 * wv + 1]; }
                tile_sums[2 * (int64_t)tile] = sa_; tile_sums[2 * (int64_t)tile + 1] = sb_;
            }
        }
        continue;
    }
    for (int i = tid; i < T.BR * T.W; i += 256) acc[i] = 0.0;
    const int NT = T.BR + 2 * R + 2;                           // rings rth0 .. rth0 + NT - 1 in the ring tables
    const int rth0 = i0 - R - 1;
    if (tid < NT) {
        const int ring = rth0 + tid;
        RegRow rw;
        rw.theta = (ring < 1) ? -1.0e300 : 1.0e300;
        rw.z = rw.sth = rw.dphi = rw.inv_dphi = rw.c0 = rw.s0 = 0.0;
        rw.start = 0; rw.nr = 0; rw.ks = 0; rw.ke = 0; rw.shf = 0; rw.lim2 = 0.0;
        if (ring >= 1 && ring <= nl4 - 1) {
            int64_t st, nr64; bool shf_;
            ring_info_small(h, ring, st, nr64, shf_);
            ring_z_sth(h, ring, rw.z, rw.sth);
            rw.theta = (BFGX_ABLK2 == 5) ? kHalfPi - rw.z : atan2_generic(rw.sth, rw.z);        // (5: timing only -- what the tables' trigonometry costs)
            rw.start = st; rw.nr = (int)nr64; rw.shf = shf_ ? 1 : 0;
            rw.dphi = kTwoPi / (double)rw.nr;
            rw.inv_dphi = (double)rw.nr * kInvTwoPi;
            rw.ks = tile_ks(tj, rw.nr, nphi);
            rw.ke = tile_ks(tj + 1, rw.nr, nphi);
            if (BFGX_ABLK2 == 5) { rw.s0 = 0.0; rw.c0 = 1.0; } else
            sincos_bounded(((double)rw.ks + (shf_ ? 0.5 : 0.0)) * rw.dphi, rw.s0, rw.c0);
            // gathered <=> |o|^2 < lim2
            const double lim = fmin(fmin(reach.cap, 0.09 * rw.sth), 0.999 * fmin(rw.theta - reach.theta_first, reach.theta_last - rw.theta));
            rw.lim2 = (lim > 0.0 && ring > 1 && ring < nl4 - 1) ? (double)(float)(lim * lim) : 0.0;      // (a float: the walking kernel's scan compares in fp32)
        }
        rows[tid] = rw;
    }
    __syncthreads();
    if (tid < NT) {
        const RegRow &rw = rows[tid];
        RegRowC<real> rc;
        if constexpr (sizeof(real) == 4) {
            rc.z = (real)rw.z; rc.sth = (real)rw.sth; rc.dphi = (real)rw.dphi; rc.inv_dphi = (real)rw.inv_dphi;
            rc.c0 = (real)rw.c0; rc.s0 = (real)rw.s0;
        }
        rc.inv_dth_up = (real)0; rc.inv_dth_dn = (real)0;
        if (rw.nr > 0) {
            if (tid > 0 && rows[tid - 1].nr > 0) rc.inv_dth_up = (real)(1.0 / (rw.theta - rows[tid - 1].theta));
            if (tid < NT - 1 && rows[tid + 1].nr > 0) rc.inv_dth_dn = (real)(1.0 / (rows[tid + 1].theta - rw.theta));
        }
        rowc[tid] = rc;
        if (PASS == 2) {
            // thresholds of the scan (below): a pixel of this ring reaches the tile only if it moves (a) beyond the ring next to the
            // tile's first / last own ring -- |o| >= the colatitude it has to cross (the angle moved is at most asin |o|) -- and
            // (b), from c columns outside the tile's span on its own ring, by more than c - 2.5 - nmax / nmin columns: a move of
            // |o| is at most |o| / (sth - lim) nr / 2 pi columns (the arithmetic of tile_apron_kernel, per pixel instead of per tile)
            RowScan *scan = reinterpret_cast<RowScan *>(rowc + NTmax);
            const int T0 = R + 1, T1 = R + (i1 - i0);                          // first / last own ring of the tile in the ring tables
            double need = 0.0;
            if (tid < T0 - 1 && rows[T0 - 1].nr > 0) need = rows[T0 - 1].theta - rw.theta;
            else if (tid > T1 + 1 && rows[T1 + 1].nr > 0) need = rw.theta - rows[T1 + 1].theta;
            const float nf = (rw.nr > 0 && need > 0.0) ? (float)(0.999 * need) : 0.0f;
            const double den = rw.sth - fast_sqrt(rw.lim2);
            const float cf = (rw.nr > 0 && den > 0.0) ? (float)(1.02 * rw.inv_dphi / den) : 3.0e18f;
            const int span = rw.ke - rw.ks, rest = rw.nr - span;
            const int kl = min(kap, rest >> 1), kr = min(kap, rest - kl);     // (a short ring: what is left of it, split between the two sides)
            RowScan rs;
            rs.start = (int32_t)rw.start; rs.nr = rw.nr; rs.ks = rw.ks; rs.span = span;
            rs.need2 = nf * nf; rs.cf2 = cf * cf; rs.lim2 = (float)rw.lim2;
            rs.klr = kl | (kr << 10) | ((tid >= T0 && tid <= T1) ? (1 << 20) : 0);
            scan[tid] = rs;
        }
    }
    if (PASS == 2 && tid == 0) reinterpret_cast<int32_t *>(reinterpret_cast<RowScan *>(rowc + NTmax) + NTmax)[0] = 0;       // far pixels listed by this tile
    __syncthreads();
    auto far_add = [&](int64_t p, double v) {
        if (PASS != 1) {
            const unsigned long long i = atomicAdd(far.count, 1ull);
            if ((int64_t)i < far.cap) { far.pix[i] = p; far.val[i] = v; } else atomicOr(far.overflow, 1);
            sum_out += v;
        } else {
            atomicAdd(map_out + p, v);
        }
    };

    // source pixels: the tile's rings +- R, its columns +- K (a tile that spans whole rings has no column apron)
    // (in a short ring the apron is what is left of the ring, split between the two sides, so that no pixel is visited twice)
    const int NR = T.BR + 2 * R;
    int maxspan = 0, nrmin = 0x7fffffff, nrmax = 1;
    for (int i = 1; i <= NR; ++i) {
        maxspan = max(maxspan, rows[i].ke - rows[i].ks);
        if (PASS == 2 && rows[i].nr > 0) { nrmin = min(nrmin, rows[i].nr); nrmax = max(nrmax, rows[i].nr); }
    }
    const int LWs = maxspan + 2 * kap;
    const unsigned inv_lws = (unsigned)((0x100000000ull + (unsigned)LWs - 1u) / (unsigned)LWs);      // idx / LWs for idx < 2^16
    using OT = typename std::conditional<SPLIT, double, ACC>::type;      // a source pixel's offset as the lean / repair passes hold it
    struct Src { int ti, x; bool ok, own; double val; OT o0, o1, o2; float hsq; };      // (hsq: |hi|^2 in fp32, what SPLIT classes a pixel by -- as the scan does)
    auto fetch = [&](int idx) {
        Src sx;
        sx.ok = false; sx.own = false; sx.ti = 0; sx.x = 0; sx.val = 0.0; sx.o0 = sx.o1 = sx.o2 = (OT)0; sx.hsq = 0.0f;
        if (idx < NR * LWs) {
            const int r = (int)__umulhi((unsigned)idx, inv_lws), x = idx - r * LWs - kap;
            const RegRow &rw = rows[r + 1];
            const int span = rw.ke - rw.ks, rest = rw.nr - span;
            const int kl = min(kap, rest >> 1), kr = min(kap, rest - kl);
            if (rw.nr > 0 && x >= -kl && x < span + kr) {
                int k = rw.ks + x;
                if (k < 0) k += rw.nr;
                if (k >= rw.nr) k -= rw.nr;
                const int64_t p = (BFGX_ABLK2 == 4) ? (int64_t)((rw.start + k) & 4095) : rw.start + k;      // (4: every tile reads the same 4096 pixels -- what the loads' latency costs)
                sx.own = (r >= R) && (r < R + (i1 - i0)) && (x >= 0) && (x < span);
                if (PASS != 1 || sx.own) {
                    sx.ok = true; sx.ti = r + 1; sx.x = x;
                    sx.val = map_in[p];
                    if (SPLIT) {
                        const ACC h0 = offsets[3 * p + 0], h1 = offsets[3 * p + 1], h2 = offsets[3 * p + 2];
                        const ACC l0 = offsets_lo[3 * p + 0], l1 = offsets_lo[3 * p + 1], l2 = offsets_lo[3 * p + 2];
                        sx.o0 = (OT)((double)h0 + (double)l0); sx.o1 = (OT)((double)h1 + (double)l1); sx.o2 = (OT)((double)h2 + (double)l2);
                        sx.hsq = (float)fma_(h0, h0, fma_(h1, h1, h2 * h2));
                    } else {
                    sx.o0 = offsets[3 * p + 0]; sx.o1 = offsets[3 * p + 1]; sx.o2 = offsets[3 * p + 2];
                    }
                }
            }
        }
        return sx;
    };
    if constexpr (PASS == 2) {
        // ---- the tile's window with a reach of several rings holds 2 - 3 source pixels per stored one, most of which cannot reach the
        // tile.  SCAN: every wave tests 64 window pixels at a time with two comparisons of |o|^2 (thresholds above) and pushes the
        // survivors {position, offset, value} onto its own LDS queue; whenever 64 have come together they are taken off the top and
        // EVALUATED one per lane (the displaced position, the ring walk, four LDS adds): the costly half runs on full waves and on
        // ~1.3 pixels per stored one.  Own pixels that take the generic route are listed per tile and evaluated together at the end.
        const int lane = tid & (kWave - 1), wid = __builtin_amdgcn_readfirstlane(tid / kWave);
        const RowScan *scan = reinterpret_cast<const RowScan *>(rowc + NTmax);
        int32_t *nfar = reinterpret_cast<int32_t *>(const_cast<RowScan *>(scan) + NTmax), *farq = nfar + 4;
        double *qval = reinterpret_cast<double *>(farq + kWalkFar) + wid * kWalkQ;
        using QT = typename std::conditional<SPLIT, ACC, real>::type;          // the queue holds the offsets as read (SPLIT: the high halves)
        QT *qo = reinterpret_cast<QT *>(reinterpret_cast<double *>(farq + kWalkFar) + (256 / kWave) * kWalkQ) + wid * 3 * kWalkQ;
        int32_t *qpos = reinterpret_cast<int32_t *>(reinterpret_cast<QT *>(reinterpret_cast<double *>(farq + kWalkFar) + (256 / kWave) * kWalkQ)
                                                    + (256 / kWave) * 3 * kWalkQ) + wid * kWalkQ;
        const float colslack = 2.5f + (float)nrmax / (float)nrmin;
        const int nown = i1 - i0, total = NR * LWs;
        auto deposit = [&](int ti, int x, QT q0, QT q1, QT q2, double val) {
            int tt[4], tk[4];
            real w[4];
            if (BFGX_ABLK2 == 1) return;
            real o0 = (real)q0, o1 = (real)q1, o2 = (real)q2;
            if (SPLIT) {                                                       // the low halves of this survivor (its pixel from the ring table)
                const RegRow &rs_ = rows[ti];
                int k = rs_.ks + x;
                k += (k < 0) ? rs_.nr : 0;
                k -= (k >= rs_.nr) ? rs_.nr : 0;
                const int64_t p = rs_.start + k;
                o0 += (real)offsets_lo[3 * p + 0]; o1 += (real)offsets_lo[3 * p + 1]; o2 += (real)offsets_lo[3 * p + 2];
            }
            if (!regrid_gather_targets<real, true, SPLIT>(rows, rowc, NT, ti, x, o0, o1, o2, tt, tk, w)) return;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {                                 // regrid_pixels_hpix :64, deposits into this tile only
                const int rr = tt[q4] - (R + 1);
                if (rr >= 0 && rr < nown) {
                    const RegRow &rt = rows[tt[q4]];
                    if (tk[q4] >= rt.ks && tk[q4] < rt.ke) {
                        const double v = (double)w[q4] * val;
                        atomicAdd(acc + rr * T.W + (tk[q4] - rt.ks), v);
                    }
                }
            }
        };
        // a listed position (ring of the tables << 16 | column + kap) of one of the tile's own pixels -> its deposits on the far list
        auto far_from = [&](int pk) {
            const int ti = pk >> 16, x = (pk & 0xffff) - kap;
            const RegRow &rw = rows[ti];
            int k = rw.ks + x;
            if (k < 0) k += rw.nr;
            if (k >= rw.nr) k -= rw.nr;
            const int64_t p = rw.start + k;
            double f0 = (double)offsets[3 * p + 0], f1 = (double)offsets[3 * p + 1], f2 = (double)offsets[3 * p + 2];
            if (SPLIT) { f0 += (double)offsets_lo[3 * p + 0]; f1 += (double)offsets_lo[3 * p + 1]; f2 += (double)offsets_lo[3 * p + 2]; }
            sum_out += regrid_far_pixel(h, rows, NT - 2, rth0, ti, x, f0, f1, f2, map_in[p], far);
        };
        int32_t *fq = farq + wid * kWalkQ;
        int fn = 0;                                                            // entries in this wave's queue of generic-route pixels
        int qn = 0;                                                            // entries in this wave's queue (wave-uniform)
        constexpr int U = BFGX_K2U;                                                   // window pixels per lane and trip: 2 U loads in flight
        const int64_t pdummy = rows[R + 1].start + rows[R + 1].ks;             // (a pixel this tile may read, for the lanes without one)
        for (int base = wid * kWave; base < (BFGX_ABLK2 == 2 ? 0 : total); base += 256 * U) {
            int cpos[U], cx[U], cspan[U];
            bool cin[U], cown[U];
            float cneed2[U], ccf2[U], clim2[U];
            ACC ca0[U], ca1[U], ca2[U];
            double cval[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                                      // positions, all loads issued
                const int idx = base + u * 256 + lane, idc = min(idx, total - 1);
                const int r = (int)__umulhi((unsigned)idc, inv_lws), x = idc - r * LWs - kap;
                const RowScan rs = scan[r + 1];
                cx[u] = x; cpos[u] = ((r + 1) << 16) | (x + kap); cspan[u] = rs.span;
                cneed2[u] = rs.need2; ccf2[u] = rs.cf2; clim2[u] = rs.lim2;
                cin[u] = idx < total && rs.nr > 0 && x >= -(rs.klr & 1023) && x < rs.span + ((rs.klr >> 10) & 1023);
                cown[u] = cin[u] && (rs.klr >> 20) && x >= 0 && x < rs.span;
                int k = rs.ks + x;
                k += (k < 0) ? rs.nr : 0;
                k -= (k >= rs.nr) ? rs.nr : 0;
                const int64_t p = cin[u] ? (int64_t)(rs.start + k) : pdummy;
                ca0[u] = offsets[3 * p + 0]; ca1[u] = offsets[3 * p + 1]; ca2[u] = offsets[3 * p + 2];
                cval[u] = map_in[p];
            }
            bool cpush[U], cfar[U];
#pragma unroll
            for (int u = 0; u < U; ++u) {                                      // the two thresholds
                const int x = cx[u];
                if (cown[u]) sum_in += cval[u];
                const QT o0 = (QT)ca0[u], o1 = (QT)ca1[u], o2 = (QT)ca2[u];
                const QT osq = fma_(o0, o0, fma_(o1, o1, o2 * o2));
                const bool live = cin[u] && cval[u] > 0.0;                     // HealpixRunner.py:335
                const bool gathered = osq < (QT)clim2[u];
                const int dc = x < 0 ? -x : (x >= cspan[u] ? x - cspan[u] + 1 : 0);
                const float g = (float)dc - colslack, of = (float)osq;
                const bool reaches = cown[u] || (of >= cneed2[u] && (g <= 0.0f || of * ccf2[u] >= g * g));
                cpush[u] = live && gathered && reaches;
                cfar[u] = live && !gathered && cown[u];                        // the generic route (|o| beyond the cap, next to a pole)
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {                                      // survivors onto the queue; a full wave of them evaluated
                const unsigned long long m = __ballot(cpush[u]);
                if (m == 0ull) continue;                                       // (wave-uniform)
                if (cpush[u]) {
                    const int sl = qn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                    qval[sl] = cval[u]; qo[sl] = (QT)ca0[u]; qo[kWalkQ + sl] = (QT)ca1[u]; qo[2 * kWalkQ + sl] = (QT)ca2[u]; qpos[sl] = cpos[u];
                }
                qn += __popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (qn >= kWave) {
                    qn -= kWave;
                    const int sl = qn + lane, pk = qpos[sl];
                    deposit(pk >> 16, (pk & 0xffff) - kap, qo[sl], qo[kWalkQ + sl], qo[2 * kWalkQ + sl], qval[sl]);
                    __builtin_amdgcn_wave_barrier();
                }
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {                                      // the same for the pixels of the generic route (usually none)
                const unsigned long long m = __ballot(cfar[u]);
                if (m == 0ull) continue;
                if (cfar[u]) fq[fn + (int)__builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u))] = cpos[u];
                fn += __popcll(m);
                __builtin_amdgcn_wave_barrier();
                if (fn >= kWave) {
                    fn -= kWave;
                    far_from(fq[fn + lane]);
                    __builtin_amdgcn_wave_barrier();
                }
            }
        }
        if (lane < qn) {
            const int pk = qpos[lane];
            deposit(pk >> 16, (pk & 0xffff) - kap, qo[lane], qo[kWalkQ + lane], qo[2 * kWalkQ + lane], qval[lane]);
        }
        if (lane < fn) far_from(fq[lane]);
    } else {
    auto handle = [&](const Src &cur) {
        if (cur.ok && cur.own) sum_in += cur.val;
        if (!cur.ok || !(cur.val > 0.0)) return;                             // HealpixRunner.py:335
        const double val = cur.val;
        const real o0 = (real)cur.o0, o1 = (real)cur.o1, o2 = (real)cur.o2;
        const real osq = fma_(o0, o0, fma_(o1, o1, o2 * o2));
        // (whether a pixel is gathered is decided by the same arithmetic in every pass: SPLIT -- fp32 on the high halves, as the scan)
        const bool gathered = SPLIT ? (cur.hsq < (float)rows[cur.ti].lim2) : ((double)osq < rows[cur.ti].lim2);
        if (gathered) {
            if (PASS == 1) return;
            int tt[4], tk[4];
            real w[4];
            if (!regrid_gather_targets<real, false, SPLIT>(rows, rowc, NT, cur.ti, cur.x, o0, o1, o2, tt, tk, w)) return;
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {                                 // regrid_pixels_hpix :64, deposits into this tile only
                const int rr = tt[q4] - (R + 1);
                if (rr >= 0 && rr < i1 - i0) {
                    const RegRow &rt = rows[tt[q4]];
                    if (tk[q4] >= rt.ks && tk[q4] < rt.ke) {
                        const double v = (double)w[q4] * val;
                        atomicAdd(acc + rr * T.W + (tk[q4] - rt.ks), v);
                    }
                }
            }
        } else if (cur.own) {                                                // rare: the owner of the source pixel lists its deposits
            int tr[4], tk[4];
            double w[4];
            regrid_targets_generic(h, rows, NT - 2, rth0, cur.ti, cur.x, (double)cur.o0, (double)cur.o1, (double)cur.o2, tr, tk, w);
            for (int q4 = 0; q4 < 4; ++q4) {
                int64_t st_t, nr64; bool sh_t;
                ring_info_small(h, tr[q4], st_t, nr64, sh_t);
                far_add(st_t + tk[q4], w[q4] * val);
            }
        }
    };
    if (BFGX_K2LEAN_U == 2) {
        // two window pixels per lane and trip: the four loads are in flight before either pixel is evaluated
        for (int idx = tid; idx < (BFGX_ABLK2 == 3 ? 0 : NR * LWs); idx += 512) {
            const Src a = fetch(idx), b = fetch(idx + 256);
            handle(a);
            handle(b);
        }
    } else
        for (int idx = tid; idx < NR * LWs; idx += 256) handle(fetch(idx));
    }
    if (PASS == 1) continue;
    __syncthreads();

    // flush: every pixel of the tile is stored exactly once
    const int lane = tid & (kWave - 1), wid = tid / kWave;
    for (int rr = wid; rr < i1 - i0; rr += 256 / kWave) {
        const RegRow &rt = rows[rr + R + 1];
        double *dst = map_out + rt.start + rt.ks;
        const int n = rt.ke - rt.ks;
        for (int xx = lane; xx < n; xx += kWave) {
            const double v = acc[rr * T.W + xx];
            dst[xx] = v;
            sum_out += v;                                   // (the deposits that landed in this tile: summed here, once per pixel)
        }
    }
    if (tile_sums) {
        __syncthreads();                                   // acc is free again: reuse its first words for the block reduction
#pragma unroll
        for (int sft = kWave >> 1; sft > 0; sft >>= 1) { sum_in += __shfl_down(sum_in, sft, kWave); sum_out += __shfl_down(sum_out, sft, kWave); }
        if (lane == 0) { acc[2 * wid] = sum_in; acc[2 * wid + 1] = sum_out; }
        __syncthreads();
        if (tid == 0) {
            double sa_ = 0.0, sb_ = 0.0;
            for (int wv = 0; wv < 256 / kWave; ++wv) { sa_ += acc[2 * wv]; sb_ += acc[2 * wv + 1]; }
            tile_sums[2 * (int64_t)tile] = sa_; tile_sums[2 * (int64_t)tile + 1] = sb_;
        }
    }
    __syncthreads();
    }
}

// banded regrid: adds the listed deposits whose pixel lies in [p0, p1) to the slice that starts at pixel p0, counts the others
// (tile_sums != NULL: the last workgroup also adds up the per-tile sums of the banded regrid into sums[2] -- one launch less per step)
__global__ void __launch_bounds__(256)
regrid_far_local_kernel(FarList far, double *__restrict__ out_slice, int64_t p0, int64_t p1, unsigned long long *__restrict__ foreign,
                        int ntiles = 0, const double *__restrict__ tile_sums = nullptr, double *__restrict__ sums = nullptr)
{
    if (tile_sums != nullptr && blockIdx.x == gridDim.x - 1) {
        __shared__ double sa[256 / kWave], sb[256 / kWave];
        double xa = 0.0, xb = 0.0;
        for (int t = threadIdx.x; t < ntiles; t += 256) { xa += tile_sums[2 * t]; xb += tile_sums[2 * t + 1]; }
#pragma unroll
        for (int sft = kWave >> 1; sft > 0; sft >>= 1) { xa += __shfl_down(xa, sft, kWave); xb += __shfl_down(xb, sft, kWave); }
        const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
        if (lane == 0) { sa[wid] = xa; sb[wid] = xb; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ta = 0.0, tb = 0.0;
            for (int w = 0; w < 256 / kWave; ++w) { ta += sa[w]; tb += sb[w]; }
            sums[0] = ta; sums[1] = tb;
        }
    }
    unsigned long long n = *far.count;
    if ((int64_t)n > far.cap) n = (unsigned long long)far.cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const int64_t p = far.pix[i];
        if (p >= p0 && p < p1) atomicAdd(out_slice + (p - p0), far.val[i]);
        else if (foreign) atomicAdd(foreign, 1ull);
    }
}

}  // namespace bfgx
