// bfgx_grid.hpp -- regular-grid (2D / 3D, periodic) baryonification and painting kernels for gfx950.
//
//   grid_prep_kernel        per-halo scalars of BaryonifyGrid / PaintProfilesGrid   Map2DRunner.py:478-523, :710-747
//   grid_scatter_kernel     cutout loops: MODE 0 pixel offsets :519-575, MODE 1 painting :749-812
//   grid_regrid_kernel      post-loop regrid :577-599 + regrid_pixels_2D/3D :14-163
//   pixel_deposit_kernel    regrid_pixels_2D/3D on caller-given positions
//   particle_deposit_kernel ParticleSnapshot.make_map (np.histogramdd, io.py:622-670)
//
// Work decomposition: a halo's cutout is a cube of Nsize^d pixels, of which only the ball r < rcut contributes.
// K_prep clips the cube to the ball's bounding box and cuts it into work items of kGridChunk pixels, reserving a
// contiguous range of the item table with one returning atomic; K_scatter runs a fixed number of workgroups that
// stride over the item table, so one 256^3-pixel cluster cutout and 10^5 8^3-pixel group cutouts load the chip
// evenly.  Accumulation is fp64 global atomics (hardware global_atomic_add_f64); NaN contributions are added as
// they are, because the reference zeroes a pixel's WHOLE accumulated offset when any contribution was non-finite
// (np.where(np.isfinite(pix_offsets), ..., 0) after the loop, :580/:591).
//
// Reference conventions kept as they are (see oracle/bfg_oracle.c): meshgrid(indexing='xy') pairs the first array
// axis of the cutout with the "y" coordinate x[i] + dy and the second with x[j] + dx while the first axis is
// indexed around x_cen; linspace(-N/2, N/2, N) * res spaces the cutout samples by N/(N-1) pixels.
#pragma once
#include <cstddef>
#include "bfgx_kernels.hpp"

namespace bfgx {

constexpr int kGridChunk = 4096;      // bounding-box pixels per work item
constexpr int kGridBlock = 256;

struct GridGeom {
    int32_t ndim, npix;
    const double *bins;               // device, [npix] pixel-centre coordinates, strictly ascending
    double res;                       // bins[1] - bins[0]
    double half_box;                  // max(bins) / 2
    double a;                         // 1 / (1 + redshift)
    int64_t ntot;                     // npix^ndim
    int32_t slab_lo, slab_n;          // planes of the FIRST array axis this plan owns (the whole grid: 0, npix); outputs of the
                                      // halo loop and inputs of the regrid hold slab_n x npix (x npix) cells
};

struct GridCatalog {
    int64_t n;
    const double *M, *x, *y, *z;      // device columns; z unused for 2D maps
    const double *lnM;                // optional: the ln M table coordinate as the caller evaluated it (float32 log)
    const double *rmat;               // optional [n][4]: row-major 2x2 shear matrices (use_ellipticity, 2D only)
    const double *extra[BFGX_MAX_EXTRA];
};

struct GridHaloRec {
    double dax[3];                    // pixel-centre minus halo position, by CUTOUT axis: {dy, dx, dz}
    double step, start, top;          // np.linspace(-N/2, N/2, N): x[i] = (i * step + start) * res, x[N-1] = top * res
    double rcut;                      // MODE 0: eps_model * R_model (comoving); MODE 1: eps_runner * R_j (comoving)
    double lnoff;                     // added to ln r: -ln R_model when the table is Rdelta-sampled, else 0
    double rmat[4];
    int32_t cen[3];                   // x_cen, y_cen, z_cen = centre pixel along cutout axes i, j, k
    int32_t nsize;                    // 0: halo skipped
    int32_t lo[3], n[3];              // bounding box of the contributing cutout indices
    int32_t oob, ell;
    int32_t chunk0, nchunks;
    // the (z, M[, property]) corner rows LAST: a kernel that stages records in LDS copies the head and the NC corners a table has
    // (4 for a 3-axis table), not all kNCmax
    double w[kNCmax];
    int32_t rowoff[kNCmax];
};
static_assert(offsetof(GridHaloRec, w) % 8 == 0 && offsetof(GridHaloRec, rowoff) == offsetof(GridHaloRec, w) + 8 * kNCmax, "GridHaloRec layout");

// np.argmin(np.abs(bins - x)) for strictly ascending bins (first index attaining the minimum)
__device__ inline int nearest_bin(const double *__restrict__ b, int n, double x)
{
    if (!(x > b[0])) return 0;
    if (x >= b[n - 1]) return n - 1;
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (b[mid] <= x) lo = mid; else hi = mid; }
    return (fabs(b[hi] - x) < fabs(b[lo] - x)) ? hi : lo;
}

__device__ inline double cutout_coord(int idx, int nsize, double step, double start, double top, double res)
{
    // (i * step + start) * res with the roundings numpy makes (no contraction); the last sample is set to stop
    const double y = (idx == nsize - 1) ? top : add_nc(mul_nc((double)idx, step), start);
    return mul_nc(y, res);
}

// counters: [0] = work items reserved, [1] = flags (bit 0: the reference's "Halo offsets ... larger than res" assert)
template <int NC>
__global__ void __launch_bounds__(kGridBlock)
grid_prep_kernel(DevModel m, GridGeom g, GridCatalog c, int mode, GridHaloRec *__restrict__ recs,
                 int32_t *__restrict__ chunk_halo, int64_t capacity, int32_t *__restrict__ counters)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= c.n) return;
    GridHaloRec r;
    r.nsize = 0; r.nchunks = 0; r.chunk0 = 0; r.oob = 1; r.ell = 0;
    for (int q = 0; q < 3; ++q) { r.dax[q] = 0.0; r.cen[q] = 0; r.lo[q] = 0; r.n[q] = 0; }
    for (int q = 0; q < 4; ++q) r.rmat[q] = 0.0;
    for (int q = 0; q < kNCmax; ++q) { r.w[q] = 0.0; r.rowoff[q] = 0; }
    r.step = r.start = r.top = r.rcut = r.lnoff = 0.0;

    const double M_j = c.M[j], x_j = c.x[j], y_j = c.y[j], z_j = (g.ndim == 3) ? c.z[j] : 0.0;
    const double a = g.a;
    const bool valid = (M_j > 0.0) && isfinite(M_j) && isfinite(x_j) && isfinite(y_j) && isfinite(z_j);
    double Ns = 0.0, R_com = 0.0, R_phys = 0.0;
    if (valid) {
        R_phys = dev_radius(m.bg_runner, m.md_runner, M_j, a);            // physical Mpc (:486, :718)
        R_com = R_phys / a;
        if (mode == 0) {
            double R_q = m.eps_runner * R_phys / a;                       // :487
            R_q = fmin(fmax(R_q, 0.0), g.half_box);                       // :488 np.clip(R_q, 0, max(bins)/2)
            Ns = 2.0 * R_q / g.res;                                       // :496
        } else {
            Ns = 2.0 * m.eps_runner * R_com / g.res;                      // :726
        }
    }
    int nsize = 0;
    if (valid && isfinite(Ns)) {
        const double half = floor(Ns * 0.5);                              // int(Nsize // 2) * 2
        nsize = (half > 1.0e6) ? 2000000 : 2 * (int)half;
        if (mode == 0) { if (nsize < 2) nsize = 0; }                      // :498 skip
        else nsize = max(2, min(nsize, g.npix / 2));                      // :728 np.clip(Nsize, 2, bins.size // 2)
        if (nsize > g.npix) nsize = g.npix - (g.npix & 1);                // a cutout never exceeds the box (R_q <= box/2)
    }
    if (nsize >= 2) {
        r.nsize = nsize;
        r.start = -(double)nsize / 2.0;
        r.top = (double)nsize / 2.0;
        r.step = (r.top - r.start) / (double)(nsize - 1);
        r.cen[0] = nearest_bin(g.bins, g.npix, x_j);
        r.cen[1] = nearest_bin(g.bins, g.npix, y_j);
        r.cen[2] = (g.ndim == 3) ? nearest_bin(g.bins, g.npix, z_j) : 0;
        const double dx = g.bins[r.cen[0]] - x_j, dy = g.bins[r.cen[1]] - y_j;
        const double dz = (g.ndim == 3) ? g.bins[r.cen[2]] - z_j : 0.0;
        r.dax[0] = dy; r.dax[1] = dx; r.dax[2] = dz;
        if (g.ndim == 2 && !(dx <= g.res && dy <= g.res)) atomicOr(counters + 1, 1);   // :516, :747

        const double Rmod = (m.same_model ? R_phys : dev_radius(m.bg_model, m.md_model, M_j, a)) / a;
        r.rcut = (mode == 0) ? m.tab.eps_model * Rmod : R_com * m.eps_runner;
        r.lnoff = (mode == 0 && m.tab.rdelta) ? -log(Rmod) : 0.0;
        const double x0 = log(1.0 / a);
        const double x1 = c.lnM ? c.lnM[j] : (double)logf((float)M_j);    // float32 log of the float32 catalog mass
        double wv[NC];
        int32_t ro[NC];
        const double xe[2] = {(NC >= 8) ? c.extra[0][j] : 0.0, (NC >= 16) ? c.extra[1][j] : 0.0};
        const bool oob = table_corners<NC>(m.tab, m.tab.axis[0], m.tab.axis[1], x0, x1, xe, wv, ro);
        r.oob = oob ? 1 : 0;
        for (int q = 0; q < NC; ++q) { r.w[q] = wv[q]; r.rowoff[q] = ro[q]; }
        if (c.rmat) { r.ell = 1; for (int q = 0; q < 4; ++q) r.rmat[q] = c.rmat[4 * j + q]; }

        // bounding box of the ball r < rcut inside the cutout (|coordinate| >= rcut implies r >= rcut, also after
        // rounding); sheared radii are not bounded by the coordinates, so an elliptical halo keeps its whole cutout
        int64_t vol = 1;
        for (int ax = 0; ax < g.ndim; ++ax) {
            int lo = 0, hi = nsize;
            if (!r.ell) {
                while (lo < hi && !(cutout_coord(lo, nsize, r.step, r.start, r.top, g.res) + r.dax[ax] > -r.rcut)) ++lo;
                while (hi > lo && !(cutout_coord(hi - 1, nsize, r.step, r.start, r.top, g.res) + r.dax[ax] < r.rcut)) --hi;
            }
            if (ax == 0 && g.slab_n < g.npix) {
                // slab decomposition: keep the cutout rows whose (periodic) plane can belong to this slab -- the hull of the
                // intersections with the slab and its two periodic images; the scatter kernel checks ownership per pixel
                const int off = r.cen[0] - (nsize >> 1), P0 = off + lo, P1 = off + hi;
                int a = 0x7fffffff, b = -0x7fffffff;
                for (int mm = -1; mm <= 1; ++mm) {
                    const int lo_m = max(P0, g.slab_lo + mm * g.npix), hi_m = min(P1, g.slab_lo + g.slab_n + mm * g.npix);
                    if (lo_m < hi_m) { a = min(a, lo_m); b = max(b, hi_m); }
                }
                if (a < b) { lo = a - off; hi = b - off; } else hi = lo;
            }
            r.lo[ax] = lo; r.n[ax] = hi - lo;
            vol *= (hi - lo);
        }
        if (g.ndim == 2) { r.lo[2] = 0; r.n[2] = 1; }
        // out-of-table halos contribute NaN to every pixel of the ball in MODE 0 (poisoning, see header) and nothing in MODE 1
        if (oob && mode == 1) vol = 0;
        const int nchunks = (int)((vol + kGridChunk - 1) / kGridChunk);
        r.nchunks = nchunks;
        if (nchunks > 0) {
            const int c0 = atomicAdd(counters + 0, nchunks);
            r.chunk0 = c0;
            if ((int64_t)c0 + nchunks <= capacity)
                for (int q = 0; q < nchunks; ++q) chunk_halo[c0 + q] = (int32_t)j;
        }
    }
    recs[j] = r;
}

template <int DIM, int MODE, int NC>
__global__ void __launch_bounds__(kGridBlock)
grid_scatter_kernel(PairTable pt, GridGeom g, const GridHaloRec *__restrict__ recs, const int32_t *__restrict__ chunk_halo,
                    const int32_t *__restrict__ counters, double *__restrict__ out, unsigned long long *__restrict__ pair_total)
{
    __shared__ GridHaloRec R;
    const int nitems = counters[0];
    unsigned long long npairs = 0;
    for (int item = blockIdx.x; item < nitems; item += gridDim.x) {
        __syncthreads();
        {
            const int32_t *src = reinterpret_cast<const int32_t *>(recs + chunk_halo[item]);
            int32_t *dst = reinterpret_cast<int32_t *>(&R);
            for (int t = threadIdx.x; t < (int)(sizeof(GridHaloRec) / 4); t += kGridBlock) dst[t] = src[t];
        }
        __syncthreads();
        const int n1 = R.n[1], n2 = (DIM == 3) ? R.n[2] : 1;
        const int V = R.n[0] * n1 * n2;
        const int t0 = (item - R.chunk0) * kGridChunk, t1 = min(V, t0 + kGridChunk);
        const int wdt = R.nsize >> 1, N = g.npix;
        for (int t = t0 + (int)threadIdx.x; t < t1; t += kGridBlock) {
            const int k2 = (DIM == 3) ? t % n2 : 0;
            const int q = (DIM == 3) ? t / n2 : t;
            const int i = R.lo[0] + q / n1, jj = R.lo[1] + q % n1, k = R.lo[2] + k2;
            // meshgrid(x, x[, x], indexing='xy'): x_grid[i, j, k] = x[j], y_grid = x[i], z_grid = x[k]
            const double Y = cutout_coord(i, R.nsize, R.step, R.start, R.top, g.res) + R.dax[0];
            const double X = cutout_coord(jj, R.nsize, R.step, R.start, R.top, g.res) + R.dax[1];
            const double Z = (DIM == 3) ? cutout_coord(k, R.nsize, R.step, R.start, R.top, g.res) + R.dax[2] : 0.0;
            double r2 = add_nc(mul_nc(X, X), mul_nc(Y, Y));
            if (DIM == 3) r2 = add_nc(r2, mul_nc(Z, Z));
            const double rr = __dsqrt_rn(r2);                                     // :519, :556
            double r_eval = rr;
            if (DIM == 2 && R.ell) {                                              // :525-530
                const double Xe = X * R.rmat[0] + Y * R.rmat[2], Ye = X * R.rmat[1] + Y * R.rmat[3];
                r_eval = __dsqrt_rn(add_nc(mul_nc(Xe, Xe), mul_nc(Ye, Ye)));
            }
            int pi = R.cen[0] - wdt + i, pj = R.cen[1] - wdt + jj, pk = (DIM == 3) ? R.cen[2] - wdt + k : 0;   // pick_indices
            pi += (pi < 0) ? N : 0; pi -= (pi >= N) ? N : 0;
            pj += (pj < 0) ? N : 0; pj -= (pj >= N) ? N : 0;
            if (DIM == 3) { pk += (pk < 0) ? N : 0; pk -= (pk >= N) ? N : 0; }
            pi -= g.slab_lo;                                                      // plane inside the slab this plan owns
            pi += (pi < 0) ? N : 0;
            if (pi >= g.slab_n) continue;
            const int64_t flat = (DIM == 3) ? ((int64_t)pi * N + pj) * N + pk : (int64_t)pi * N + pj;
            const double lx = log(r_eval) + R.lnoff;
            double d = R.oob ? __builtin_nan("") : radial_readout<NC>(pt, R.rowoff, R.w, lx);
            if (MODE == 0) {
                if (!(r_eval < R.rcut)) d = 0.0;                                  // BaryonCorrection.py:381-382
                if (d == 0.0) continue;
                ++npairs;
                const double off = d / g.res;                                     // :534, :569
                atomicAdd(out + DIM * flat + 0, off * (X / rr));
                atomicAdd(out + DIM * flat + 1, off * (Y / rr));
                if (DIM == 3) atomicAdd(out + DIM * flat + 2, off * (Z / rr));
            } else {
                const double P = fast_exp(d);                                          // Tabulate.py:285-286
                if (!(isfinite(P) && r_eval < R.rcut) || P == 0.0) continue;      // :800-801
                ++npairs;
                atomicAdd(out + flat, P);
            }
        }
    }
    if (pair_total) {                     // one atomic per workgroup (same-address atomics serialise)
        __shared__ unsigned long long wsum[kGridBlock / kWave];
#pragma unroll
        for (int s = kWave >> 1; s > 0; s >>= 1) npairs += __shfl_down(npairs, s, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = npairs;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < kGridBlock / kWave; ++w) t += wsum[w];
            if (t) atomicAdd(pair_total, t);
        }
    }
}

// Python's float % for a positive modulus (CPython float_rem)
__device__ inline double py_mod_pos(double x, double n)
{
    // a displaced pixel lies within one box length of the grid: there fmod(x, n) is x itself (|x| < n) or x - n (exact), and the
    // ~100-instruction fmod is left to the positions further out
    double m;
    if (x > -n && x < 2.0 * n) m = (x >= n) ? x - n : x;
    else m = fmod(x, n);
    if (m != 0.0) { if (m < 0.0) m += n; }
    else m = 0.0;
    return m;
}

// one axis of regrid_pixels_*: the displaced unit cell [s, s + 1] overlaps cell A = int(s) by (A + 1) - s and the next
// cell by (s + 1) - (A + 1); of the reference's 5 candidate cells only these two can have a positive overlap, and the
// periodic-image branches (:72-78) evaluate to the same two expressions.  Needs N >= 5 (below that the reference's
// 5-cell loop visits a cell twice).
struct AxisSplit { int cell[2]; double w[2]; };
__device__ inline AxisSplit split_axis(double pos, int N)
{
    AxisSplit s;
    const double xs = py_mod_pos(pos, (double)N);
    const double xe = xs + 1.0;
    const int i0 = (int)xs;
    s.cell[0] = (i0 >= N) ? i0 - N : i0;
    s.cell[1] = (i0 + 1 >= N) ? i0 + 1 - N : i0 + 1;
    s.w[0] = (double)(i0 + 1) - xs;
    s.w[1] = xe - (double)(i0 + 1);
    return s;
}

// plane of the first array axis -> plane of a slab buffer that holds [slab_lo - apron, slab_lo + slab_n + apron) (periodic);
// -1: outside the buffer.  The whole grid: slab_lo = 0, slab_n = N, apron = 0 (identity).
struct SlabWin { int lo, n, apron; };
__device__ inline int slab_plane(const SlabWin &s, int plane, int N)
{
    int l = plane - s.lo + s.apron;
    l += (l < 0) ? N : 0;
    l -= (l >= N) ? N : 0;
    return (l < s.n + 2 * s.apron) ? l : -1;
}

template <int DIM>
__device__ inline double deposit_cell(const double pos[3], double v, int N, double *__restrict__ grid, const SlabWin sw = SlabWin{0, 0x40000000, 0},
                                      int32_t *__restrict__ missed = nullptr)
{
    double dep = 0.0;                     // what was actually added to the grid
    // pos[0] moves along the SECOND array axis (j), pos[1] along the first (i), pos[2] along the third (k)
    const AxisSplit sx = split_axis(pos[0], N), sy = split_axis(pos[1], N);
    if (DIM == 3) {
        const AxisSplit sz = split_axis(pos[2], N);
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int c = 0; c < 2; ++c) {
                    const double dy = sy.w[a], dx = sx.w[b], dz = sz.w[c];
                    if (dx > 0.0 && dy > 0.0 && dz > 0.0) {
                        const double w = mul_nc(mul_nc(mul_nc(dx, dy), dz), v);
                        const int pl = slab_plane(sw, sy.cell[a], N);
                        if (pl < 0) { if (missed) atomicOr(missed, 1); continue; }
                        atomicAdd(grid + ((int64_t)pl * N + sx.cell[b]) * N + sz.cell[c], w);
                        dep += w;
                    }
                }
    } else {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                const double dy = sy.w[a], dx = sx.w[b];
                if (dx > 0.0 && dy > 0.0) {
                    const double w = mul_nc(mul_nc(dx, dy), v);
                    const int pl = slab_plane(sw, sy.cell[a], N);
                    if (pl < 0) { if (missed) atomicOr(missed, 1); continue; }
                    atomicAdd(grid + (int64_t)pl * N + sx.cell[b], w);
                    dep += w;
                }
            }
    }
    return dep;
}

// Map2DRunner.py:577-599: offsets -> finite or 0, plus the pixel's own (x, y[, z]) = (second, first[, third]) index.
// A pixel whose offsets are all exactly zero (most of a map: everything outside the halos' balls) overlaps only its
// own cell with weight 1, so it skips the modulo / split arithmetic.  block_sums (optional, [gridDim.x][2]) receives
// the workgroup's {sum of source values, sum of deposited values} for the mass-conservation check (:601-605).
template <int DIM>
__global__ void __launch_bounds__(256)
grid_regrid_kernel(int N, int64_t ntot, const double *__restrict__ offsets, const double *__restrict__ map_in,
                   double *__restrict__ map_out, double *__restrict__ block_sums, SlabWin sw, int32_t *__restrict__ missed)
{
    // ntot = cells of the slab (sw.n planes); map_in / offsets hold the slab, map_out the slab + sw.apron planes either side
    const int64_t plane_cells = (DIM == 3) ? (int64_t)N * N : (int64_t)N;
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    double sum_in = 0.0, sum_out = 0.0;
    if (p < ntot) {
        const double v = map_in[p];
        sum_in = v;
        if (v != 0.0) {                           // an empty cell adds exactly nothing
            double o[3] = {0.0, 0.0, 0.0};
#pragma unroll
            for (int c = 0; c < DIM; ++c) { o[c] = offsets[DIM * p + c]; if (!isfinite(o[c])) o[c] = 0.0; }
            if (o[0] == 0.0 && o[1] == 0.0 && o[2] == 0.0) {
                atomicAdd(map_out + p + (int64_t)sw.apron * plane_cells, v);
                sum_out = v;
            } else {
                // (32-bit arithmetic: npix^ndim < 2^31 for every grid the plan accepts, and 64-bit divisions are software loops)
                int p0, p1, p2 = 0;
                const unsigned pu = (unsigned)p, Nu = (unsigned)N;
                if (DIM == 3) { p2 = (int)(pu % Nu); const unsigned q = pu / Nu; p1 = (int)(q % Nu); p0 = (int)(q / Nu); }
                else { p1 = (int)(pu % Nu); p0 = (int)(pu / Nu); }
                const double pos[3] = {o[0] + (double)p1, o[1] + (double)(p0 + sw.lo), o[2] + (double)p2};
                sum_out = deposit_cell<DIM>(pos, v, N, map_out, sw, missed);
            }
        }
    }
    if (block_sums) {
        __shared__ double sa[256 / kWave], sb[256 / kWave];
#pragma unroll
        for (int s = kWave >> 1; s > 0; s >>= 1) { sum_in += __shfl_down(sum_in, s, kWave); sum_out += __shfl_down(sum_out, s, kWave); }
        const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
        if (lane == 0) { sa[wid] = sum_in; sb[wid] = sum_out; }
        __syncthreads();
        if (threadIdx.x == 0) {
            double ta = 0.0, tb = 0.0;
            for (int w = 0; w < 256 / kWave; ++w) { ta += sa[w]; tb += sb[w]; }
            block_sums[2 * (int64_t)blockIdx.x] = ta; block_sums[2 * (int64_t)blockIdx.x + 1] = tb;
        }
    }
}

// sums[0] += sum of block_sums[2 b], sums[1] += sum of block_sums[2 b + 1]   (grid-stride, one atomic pair per workgroup)
__global__ void __launch_bounds__(256)
sum_blocks_kernel(int64_t nblocks, const double *__restrict__ block_sums, double *__restrict__ sums)
{
    __shared__ double sa[256 / kWave], sb[256 / kWave];
    double xa = 0.0, xb = 0.0;
    const double2 *bs = reinterpret_cast<const double2 *>(block_sums);
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * 256) { const double2 v = bs[b]; xa += v.x; xb += v.y; }
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

// regrid_pixels_2D / regrid_pixels_3D on caller-given positions [n][DIM] and values [n]
template <int DIM>
__global__ void __launch_bounds__(256)
pixel_deposit_kernel(int N, int64_t n, const double *__restrict__ positions, const double *__restrict__ values,
                     double *__restrict__ grid)
{
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    double pos[3] = {0.0, 0.0, 0.0};
#pragma unroll
    for (int c = 0; c < DIM; ++c) pos[c] = positions[DIM * p + c];
    (void)deposit_cell<DIM>(pos, values[p], N, grid);
}

// np.histogramdd(coords, bins = (edges,) * DIM, weights = mass): bin b holds edges[b] <= v < edges[b + 1], the last
// edge inclusive; anything else is dropped
// `scale` = nb / (e[nb] - e[0]) only seeds the search (computed once per thread: an fp64 division per particle and axis is what
// the deposit kernels would otherwise spend most of their arithmetic on); the two loops make the result exact whatever the seed
__device__ inline int histogram_bin(const double *__restrict__ e, int nb, double v, double scale)
{
    const double e0 = e[0];
    if (!(v >= e0) || !(v <= e[nb])) return -1;
    int b = (int)((v - e0) * scale);
    b = max(0, min(b, nb - 1));
    while (b > 0 && e[b] > v) --b;
    while (b < nb - 1 && e[b + 1] <= v) ++b;
    return b;
}
__device__ inline int histogram_bin(const double *__restrict__ e, int nb, double v) { return histogram_bin(e, nb, v, (double)nb / (e[nb] - e[0])); }

template <int DIM>
__global__ void __launch_bounds__(256)
particle_deposit_kernel(int64_t n, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                        const double *__restrict__ mass, int nb, const double *__restrict__ edges, double *__restrict__ out,
                        int plane_lo, int plane_n, int64_t stride)
{
    // out holds the planes [plane_lo, plane_lo + plane_n) of the first axis (the whole grid: 0, nb)
    const int64_t p = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= n) return;
    const double scale = (double)nb / (edges[nb] - edges[0]);
    int bx = histogram_bin(edges, nb, x[p * stride], scale);
    const int by = histogram_bin(edges, nb, y[p * stride], scale);
    const int bz = (DIM == 3) ? histogram_bin(edges, nb, z[p * stride], scale) : 0;
    bx = (bx >= plane_lo && bx < plane_lo + plane_n) ? bx - plane_lo : -1;
    if (bx < 0 || by < 0 || bz < 0) return;
    const int64_t flat = (DIM == 3) ? ((int64_t)bx * nb + by) * nb + bz : (int64_t)bx * nb + by;
    atomicAdd(out + flat, mass ? mass[p * stride] : 1.0);
}

// *flag |= 1 when any of n doubles at ptr[i * stride] is NaN (ParticleSnapshot.make_map's assert on the masses, io.py:636)
__global__ void __launch_bounds__(256)
nan_scan_strided_kernel(int64_t n, const double *__restrict__ ptr, int64_t stride, int32_t *__restrict__ flag)
{
    bool any = false;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) any |= (ptr[i * stride] != ptr[i * stride]);
    if (__any(any) && (threadIdx.x & (kWave - 1)) == 0) atomicOr(flag, 1);
}

}  // namespace bfgx
