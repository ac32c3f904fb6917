// bfgx_snapshot.hpp -- particle-snapshot baryonification for gfx950 (BaryonifySnapshot.process, SnapshotRunner.py:173-262).
//
// The reference loops over halos, asks a periodic KD-tree for the particles within R_q and adds a radial offset to
// each.  Here the roles are swapped so that the scatter becomes a gather: halos are binned into a uniform periodic
// cell grid (a halo is listed in every cell its query ball's bounding cube touches: count -> scan -> fill), then one
// thread per particle walks the halo list of ITS cell, keeps the halos whose ball contains it, sums their offsets in
// registers and writes the displaced, re-wrapped position once.  No atomics on the particle side, no neighbour search
// structure over the 10^8 particles, every particle read and written exactly once.
//
//   snap_halo_prep_kernel    per-halo scalars (:217-222) + cell range + per-cell counts
//   scan_* kernels           exclusive scan of the per-cell counts
//   snap_halo_fill_kernel    halo index into each touched cell's list
//   snap_displace_kernel     per particle: min-image separation (:67-92), displacement read-out, offset (:225-252),
//                            periodic re-wrap (:254-262).  KEYS = true (BaryonifySnapshot.process() followed by
//                            ParticleSnapshot.make_map(N), io.py:622-670, when only the map is wanted): the displaced
//                            position goes straight into its deposit key (bfgx_deposit.hpp) instead of three columns --
//                            24 B per particle are neither written nor read again, and deposit_keys_kernel does not run
#pragma once
#include "bfgx_kernels.hpp"
#include "bfgx_tables.hpp"
#include "bfgx_deposit.hpp"

namespace bfgx {

struct SnapGeom {
    int32_t ndim, nc;                 // dimensions; cells per side
    double L, inv_cell;               // box size; nc / L
    double a;                         // 1 / (1 + redshift)
    int64_t ncell;                    // nc^ndim
};

struct SnapHaloRec {
    double pos[3];
    double Rq2;                       // squared query radius (scipy's ball query compares squared distances)
    double rcut;                      // eps_model * R_model (comoving)
    double lnoff;                     // added to ln d: -ln R_model for Rdelta-sampled tables
    double w[kNC];
    int32_t rowoff[kNC];
    int32_t clo[3], cn[3];            // first cell and number of cells per axis (periodic)
    int32_t oob, valid;
};

// one halo of a cell's list, 16 bytes -- four to a 64-byte line (the displace kernel is bound by random 64-byte requests, not by
// bytes; with the fp64 position and radius in the list an entry took 40 bytes and a cell's list two to three lines).  What sits here is
// a conservative PRE-FILTER of the containment test: the halo's wrapped position in 20-bit fixed point per axis and an upper bound of
// the query radius that covers the quantisation (L / 2^21 per axis) and the fp32 arithmetic of the test.  A particle that passes is
// queued, and the exact fp64 test of the reference (scipy's squared-distance comparison, SnapshotRunner.py:225) is made on the full
// record, which is fetched for queued candidates only.
struct SnapEntry {
    uint64_t xyz;                     // floor(p / L * 2^20) for x | y << 20 | z << 40
    float rq_ub;                      // >= R_q + the slack above
    int32_t idx;
};
constexpr double kSnapFix = 1048576.0;            // 2^20

__device__ inline SnapEntry snap_make_entry(const SnapGeom &g, const SnapHaloRec &r, int32_t idx)
{
    SnapEntry en;
    en.xyz = 0;
    for (int ax = 0; ax < g.ndim; ++ax) {
        double p = r.pos[ax];
        p -= floor(p / g.L) * g.L;
        unsigned long long q = (unsigned long long)(p / g.L * kSnapFix);
        if (q > 1048575ull) q = 1048575ull;
        en.xyz |= q << (20 * ax);
    }
    const double rq = sqrt(r.Rq2) + 3.0e-6 * g.L;             // quantisation sqrt(3) L 2^-21 + fp32 rounding of |L| values, with room
    float f = (float)rq;
    if ((double)f < rq) f = __int_as_float(__float_as_int(f) + 1);
    en.rq_ub = f * 1.000001f;
    en.idx = idx;
    return en;
}

__device__ inline int snap_cell(double v, const SnapGeom &g)
{
    const int c = (int)floor(v * g.inv_cell);
    return max(0, min(c, g.nc - 1));
}

__device__ inline int64_t snap_cell_index(const SnapGeom &g, int cx, int cy, int cz)
{
    return (g.ndim == 3) ? ((int64_t)cx * g.nc + cy) * g.nc + cz : (int64_t)cx * g.nc + cy;
}

__global__ void __launch_bounds__(256)
snap_halo_prep_kernel(DevModel m, SnapGeom g, int64_t nh, const double *__restrict__ M, const double *__restrict__ hx,
                      const double *__restrict__ hy, const double *__restrict__ hz, const double *__restrict__ lnM,
                      SnapHaloRec *__restrict__ recs, int32_t *__restrict__ flags)
{
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nh) return;
    SnapHaloRec r;
    const double M_j = M[j];
    r.pos[0] = hx[j]; r.pos[1] = hy[j]; r.pos[2] = (g.ndim == 3) ? hz[j] : 0.0;
    r.valid = (M_j > 0.0) && isfinite(M_j) && isfinite(r.pos[0]) && isfinite(r.pos[1]) && isfinite(r.pos[2]);
    r.Rq2 = 0.0; r.rcut = 0.0; r.lnoff = 0.0; r.oob = 1;
    for (int q = 0; q < kNC; ++q) { r.w[q] = 0.0; r.rowoff[q] = 0; }
    for (int q = 0; q < 3; ++q) { r.clo[q] = 0; r.cn[q] = 0; }
    if (r.valid) {
        const double a = g.a;
        const double R_phys = dev_radius(m.bg_runner, m.md_runner, M_j, a);             // :220 physical Mpc
        double R_q = m.eps_runner * R_phys / a;                                         // :221
        R_q = fmin(fmax(R_q, 0.0), g.L / 2);                                            // :222
        r.Rq2 = R_q * R_q;
        const double Rmod = (m.same_model ? R_phys : dev_radius(m.bg_model, m.md_model, M_j, a)) / a;
        r.rcut = m.tab.eps_model * Rmod;
        r.lnoff = m.tab.rdelta ? -log(Rmod) : 0.0;
        const double x1 = lnM ? lnM[j] : (double)logf((float)M_j);                      // float32 log, see bfgx_grid.hpp
        double wv[kNC];
        int32_t ro[kNC];
        r.oob = table_corners<kNC>(m.tab, m.tab.axis[0], m.tab.axis[1], log(1.0 / a), x1, (const double *)nullptr, wv, ro) ? 1 : 0;
        for (int q = 0; q < kNC; ++q) { r.w[q] = wv[q]; r.rowoff[q] = ro[q]; }
        if (!(R_q > 0.0) || !isfinite(R_q)) r.valid = 0;
        for (int ax = 0; ax < g.ndim && r.valid; ++ax) {
            // scipy's periodic ball query wraps the query point into the box (BoxDist1D::wrap_position); the separations
            // below keep the halo position as given, as the reference does (:226-228)
            double p = r.pos[ax];
            p -= floor(p / g.L) * g.L;
            // cells of the unwrapped interval [p - R_q, p + R_q]; one extra cell on a side that crosses the box face
            // (floor((x +- L) * inv) and floor(x * inv) +- nc may differ by one after rounding)
            const double lo_v = p - R_q, hi_v = p + R_q;
            int cl = (int)floor(lo_v * g.inv_cell), ch = (int)floor(hi_v * g.inv_cell);
            if (lo_v <= 0.0) cl -= 1;
            if (hi_v >= g.L) ch += 1;
            int n = ch - cl + 1;
            if (n >= g.nc) { cl = 0; n = g.nc; }
            cl %= g.nc; if (cl < 0) cl += g.nc;
            r.clo[ax] = cl; r.cn[ax] = n;
        }
        // an out-of-table halo displaces nothing (its read-out is NaN -> offset 0, :242): keep it out of the lists
        if (r.oob) r.valid = 0;
    }
    recs[j] = r;
}

// the c-th cell (c < cn[0] cn[1] cn[2]) of a halo's cube of cells, periodic
__device__ inline int64_t snap_cube_cell(const SnapGeom &g, const SnapHaloRec &r, int c)
{
    const int nz = (g.ndim == 3) ? r.cn[2] : 1;
    const int iz = c % nz, q = c / nz, iy = q % r.cn[1], ix = q / r.cn[1];
    int cx = r.clo[0] + ix; cx -= (cx >= g.nc) ? g.nc : 0;
    int cy = r.clo[1] + iy; cy -= (cy >= g.nc) ? g.nc : 0;
    int cz = (g.ndim == 3) ? r.clo[2] + iz : 0; cz -= (cz >= g.nc) ? g.nc : 0;
    return snap_cell_index(g, cx, cy, cz);
}

// does the query ball of halo r reach into the c-th cell of its cube?  Per axis the distance from the halo to the cell's slab
// [i h, (i + 1) h] (periodic, halo position wrapped into the box), widened by 1e-9 L against rounding; the ball touches the cell iff the
// squared distances add up to no more than R_q^2.  A ball fills pi / 6 of its bounding cube: listing the halo only where the ball
// reaches keeps the corner cells of the cube (and their particles) out of the lists -- the displace kernel is bound by the list
// look-ups of the particles in listed cells.  Conservative: a cell that holds a point of the ball is never dropped.
__device__ inline bool snap_cube_cell_touched(const SnapGeom &g, const SnapHaloRec &r, int c)
{
    const int nz = (g.ndim == 3) ? r.cn[2] : 1;
    const int iz = c % nz, q = c / nz, iy = q % r.cn[1], ix = q / r.cn[1];
    const int idx[3] = {ix, iy, iz};
    const double h = g.L / (double)g.nc, slack = 1e-9 * g.L;
    double d2 = 0.0;
    for (int ax = 0; ax < g.ndim; ++ax) {
        if (r.cn[ax] >= g.nc) continue;                      // the cube spans the whole axis: no restriction from it
        int cc = r.clo[ax] + idx[ax]; cc -= (cc >= g.nc) ? g.nc : 0;
        double p = r.pos[ax];
        p -= floor(p / g.L) * g.L;
        double dc = ((double)cc + 0.5) * h - p;              // cell centre - halo, minimum image
        if (dc > 0.5 * g.L) dc -= g.L;
        if (dc < -0.5 * g.L) dc += g.L;
        const double d = fabs(dc) - 0.5 * h - slack;
        if (d > 0.0) d2 += d * d;
    }
    return d2 <= r.Rq2;
}

// one WAVE per halo, lanes = the cells of its cube (27 - 125 of them): the atomics of a halo are in flight together instead of
// one after the other in a single thread (list fill 0.63 -> 0.07 ms at 1e5 halos)
__global__ void __launch_bounds__(256)
snap_halo_count_kernel(SnapGeom g, int64_t nh, const SnapHaloRec *__restrict__ recs, int32_t *__restrict__ cell_count)
{
    const int64_t j = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (j >= nh) return;
    const SnapHaloRec &r = recs[j];
    if (!r.valid) return;
    const int ncell = r.cn[0] * r.cn[1] * ((g.ndim == 3) ? r.cn[2] : 1);
    for (int c = lane; c < ncell; c += kWave)
        if (snap_cube_cell_touched(g, r, c)) atomicAdd(cell_count + snap_cube_cell(g, r, c), 1);
}

__global__ void __launch_bounds__(256)
snap_halo_fill_kernel(SnapGeom g, int64_t nh, const SnapHaloRec *__restrict__ recs, const int32_t *__restrict__ cell_start,
                      int32_t *__restrict__ cell_cursor, SnapEntry *__restrict__ entries)
{
    const int64_t j = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (j >= nh) return;
    const SnapHaloRec &r = recs[j];
    if (!r.valid) return;
    const SnapEntry en = snap_make_entry(g, r, (int32_t)j);
    const int ncell = r.cn[0] * r.cn[1] * ((g.ndim == 3) ? r.cn[2] : 1);
    for (int c = lane; c < ncell; c += kWave) {
        if (!snap_cube_cell_touched(g, r, c)) continue;
        const int64_t cell = snap_cube_cell(g, r, c);
        entries[cell_start[cell] + atomicAdd(cell_cursor + cell, 1)] = en;
    }
}

// ---- exclusive scan of int32 counts (n up to 2^31): 4096 elements per block, block sums scanned by one block
constexpr int kScanPerBlock = 4 * kTabThreads;

__global__ void __launch_bounds__(kTabThreads)
scan_blocks_kernel(int64_t n, const int32_t *__restrict__ in, int32_t *__restrict__ out, int32_t *__restrict__ block_sums, int in_stride)
{
    // in_stride: words between two counts (counters that many workgroups add to sit a cache line apart)
    __shared__ int sh[kTabThreads];
    const int64_t base = (int64_t)blockIdx.x * kScanPerBlock + 4 * (int64_t)threadIdx.x;
    int v[4], s = 0;
    for (int q = 0; q < 4; ++q) { v[q] = (base + q < n) ? in[(base + q) * in_stride] : 0; s += v[q]; }
    int tot;
    int pre = block_excl_scan_int(s, sh, tot);
    for (int q = 0; q < 4; ++q) { if (base + q < n) out[base + q] = pre; pre += v[q]; }
    if (threadIdx.x == 0) block_sums[blockIdx.x] = tot;
}

// one block: exclusive scan of up to kScanPerBlock block sums, in place; total -> *total_out
__global__ void __launch_bounds__(kTabThreads)
scan_sums_kernel(int nb, int32_t *__restrict__ block_sums, int64_t *__restrict__ total_out)
{
    __shared__ int sh[kTabThreads];
    __shared__ long long carry;
    if (threadIdx.x == 0) carry = 0;
    __syncthreads();
    for (int c0 = 0; c0 < nb; c0 += kScanPerBlock) {
        const int base = c0 + 4 * threadIdx.x;
        int v[4], s = 0;
        for (int q = 0; q < 4; ++q) { v[q] = (base + q < nb) ? block_sums[base + q] : 0; s += v[q]; }
        int tot;
        int pre = block_excl_scan_int(s, sh, tot) + (int)carry;
        for (int q = 0; q < 4; ++q) { if (base + q < nb) block_sums[base + q] = pre; pre += v[q]; }
        __syncthreads();
        if (threadIdx.x == 0) carry += tot;
        __syncthreads();
    }
    if (threadIdx.x == 0) *total_out = carry;
}

__global__ void __launch_bounds__(256)
scan_add_kernel(int64_t n, int32_t *__restrict__ out, const int32_t *__restrict__ block_sums)
{
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] += block_sums[i / kScanPerBlock];
}

// one bit per cell: does any halo list the cell?  nc^3 bits (256 KB at 128^3) stay cache-resident, so the ~60 % of the
// particles that sit in empty cells never touch the 8 MB cell_start table (random 64-byte requests are what bounds the
// displace kernel, not bytes)
__global__ void __launch_bounds__(256)
cell_bitmap_kernel(int64_t ncell, const int32_t *__restrict__ cell_count, uint32_t *__restrict__ bitmap)
{
    const int64_t w = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (w * 32 >= ncell) return;
    uint32_t bits = 0;
    for (int b = 0; b < 32; ++b) {
        const int64_t c = w * 32 + b;
        if (c < ncell && cell_count[c] > 0) bits |= (1u << b);
    }
    bitmap[w] = bits;
}

// min-image of a coordinate difference (SnapshotRunner.py:87-91)
__device__ inline double min_image(double dx, double L)
{
    if (dx > L / 2) dx -= L;
    if (dx < -L / 2) dx += L;
    return dx;
}

// flags: bit 1 = a particle lies outside [0, L] (scipy's periodic KDTree refuses such data)
//
// A workgroup takes 256 particles at a time.  Phase 1 (lane = particle): cell, occupancy bit, the cell's halo list, containment
// test; a hit is only QUEUED in LDS.  Phase 2 (lane = queued hit): the fp64 read-out and the offset, added to the particle's
// LDS accumulator -- with one thread per particle doing this inline a wave ran the read-out whenever ANY of its 64 unrelated
// particles had a hit (~15 % lane utilisation, 0.9 of the kernel's 2.9 ms).  Phase 3 (lane = particle): displaced, re-wrapped
// position, written once.
#ifndef BFGX_SNAP_PPT
#define BFGX_SNAP_PPT 1             // particles per thread and round.  2 / 4 keep that many look-up chains (occupancy bit -> cell start -> list
#endif                              // entries) of a thread in flight together; measured (round 4): 2.15 / 2.99 / 2.34 ms for 1 / 2 / 4 -- the entry
                                    // records of the extra chains cost registers (126 -> 239 VGPRs at 4: two waves per SIMD), and waves hide this
                                    // kernel's latency better than chains do.  Capping the registers instead (launch bounds for 6 / 8 waves per
                                    // SIMD: 92 / 160 bytes of scratch per lane in the fp64 read-out) gave 3.1 / 3.7 ms.
constexpr int kSnapPPT = BFGX_SNAP_PPT;
constexpr int kSnapBlock = 256 * kSnapPPT;           // particles per workgroup and round
constexpr int kSnapQueue = 768 * kSnapPPT;           // queued hits per round (3 per particle; more are done inline)

#ifndef BFGX_SNAP_OCC
#define BFGX_SNAP_OCC 4             // workgroups of 256 threads per SIMD group the kernel is compiled for (register budget 512 / (4 x this))
#endif
// KEYS: where the deposit keys of the displaced particles go.  The workgroup then owns CHUNKS of kDepChunk consecutive particles -- the unit
// deposit_split_kernel<1> works on -- and leaves each chunk's histogram of level-1 buckets in wg_hist[bucket][chunk], as deposit_keys_kernel does.
struct SnapDepArgs {
    DepGeom dg;
    const double *edges;              // [dg.nb + 1]
    uint32_t *keys;                   // [np]
    int32_t *wg_hist;                 // [dg.B1][nchunks]
    int32_t nchunks, _pad;
};

template <int DIM, bool KEYS>
__global__ void __launch_bounds__(256, BFGX_SNAP_OCC)
snap_displace_kernel(PairTable pt, SnapGeom g, int64_t np, const double *px, const double *py,
                     const double *pz, const SnapHaloRec *__restrict__ recs, const uint32_t *__restrict__ bitmap,
                     const int32_t *__restrict__ cell_start, const SnapEntry *__restrict__ entries, double *ox, double *oy, double *oz,
                     int32_t *__restrict__ flags, unsigned long long *__restrict__ pair_total, int64_t sin, int64_t sout, SnapDepArgs da)
{
    static_assert(!KEYS || kDepChunk % kSnapBlock == 0, "a chunk of the deposit's sort is a whole number of rounds");
    __shared__ int dhist[KEYS ? 1024 : 1];
    __shared__ double sedge[KEYS ? kDepEdgesLds : 1];
    const bool lds_edges = KEYS && da.dg.nb + 1 <= kDepEdgesLds;
    if (KEYS && lds_edges) for (int i = threadIdx.x; i <= da.dg.nb; i += 256) sedge[i] = da.edges[i];      // (the first barrier of the loop publishes them)
    const double dscale = KEYS ? (double)da.dg.nb / (da.edges[da.dg.nb] - da.edges[0]) : 0.0;
    // sin / sout: distance in doubles between consecutive particles of the input / output columns (1: plain columns; the records entry
    // passes the record size / 8 and column pointers into the record buffer -- possibly the SAME buffer: a particle's own coordinates are
    // read before they are written, and no other thread reads them)
    __shared__ double spos[3][kSnapBlock], sacc[3][kSnapBlock];
    __shared__ int qslot[kSnapQueue], qent[kSnapQueue];
    __shared__ int qn;
    const int tid = threadIdx.x;
    unsigned long long npairs = 0;

    // exact containment test of particle `slot` against halo `idx` (:225 / :237 query_ball_point, fp64 as the reference), then the
    // read-out of the hit and its contribution to the particle (SnapshotRunner.py:228-244)
    auto hit = [&](int slot, int idx) {
        const SnapHaloRec &r = recs[idx];
        const double dx = min_image(spos[0][slot] - r.pos[0], g.L), dy = min_image(spos[1][slot] - r.pos[1], g.L);
        const double dz = (DIM == 3) ? min_image(spos[2][slot] - r.pos[2], g.L) : 0.0;
        double d2 = add_nc(mul_nc(dx, dx), mul_nc(dy, dy));
        if (DIM == 3) d2 = add_nc(d2, mul_nc(dz, dz));
        if (!(d2 <= r.Rq2)) return;
        // libm-free (bfgx_math.hpp)
        const double inv_d = (d2 > 0.0) ? fast_rsq(d2) : 0.0;
        const double d = d2 * inv_d;                                             // :228 compute_distance
        const double lnd = (d2 > 0.0) ? 0.5 * fast_log(d2) : -1.0e300;           // ln 0 -> below any table: NaN read-out
        double disp = radial_readout<kNC>(pt, r.rowoff, r.w, lnd + r.lnoff);     // BaryonCorrection.py:356-379
        if (!(d < r.rcut)) disp = 0.0;                                           // :381-382
        double off = disp * g.a;                                                 // :240 displacement * a
        if (!isfinite(off)) off = 0.0;                                           // :241
        if (off == 0.0 && d > 0.0) return;
        ++npairs;
        if (d2 > 0.0) {
            const double oi = off * inv_d;                                       // :242-244 offset * (dx / d)
            atomicAdd(&sacc[0][slot], oi * dx); atomicAdd(&sacc[1][slot], oi * dy);
            if (DIM == 3) atomicAdd(&sacc[2][slot], oi * dz);
        } else {                                                                 // d = 0: 0 * (0 / 0) = NaN, as the reference
            const double nan = __builtin_nan("");
            atomicAdd(&sacc[0][slot], nan); atomicAdd(&sacc[1][slot], nan); atomicAdd(&sacc[2][slot], nan);
        }
    };

    // pre-filter of particle `slot` (fp32 position pf) against list entry e: a candidate is only QUEUED for phase 2 (queuing every
    // entry, so that this test runs one per lane, was measured: no gain -- the loop waits for the list entries, not for diverged lanes)
    const float Lf = (float)g.L, fixf = (float)(g.L / kSnapFix);
    auto test = [&](int slot, const float pf[3], const SnapEntry en) {
        float d2 = 0.0f;
#pragma unroll
        for (int ax = 0; ax < DIM; ++ax) {
            const float hp = ((float)((en.xyz >> (20 * ax)) & 1048575ull) + 0.5f) * fixf;
            float d = pf[ax] - hp;
            d -= (d > 0.5f * Lf) ? Lf : 0.0f;
            d += (d < -0.5f * Lf) ? Lf : 0.0f;
            d2 = __builtin_fmaf(d, d, d2);
        }
        if (!(d2 <= en.rq_ub * en.rq_ub)) return;
        const int qi = atomicAdd(&qn, 1);
        if (qi < kSnapQueue) { qslot[qi] = slot; qent[qi] = en.idx; }
        else hit(slot, en.idx);                                                  // queue full (a cluster core): done here
    };

    // grid-stride over blocks of 256 particles (KEYS: over chunks of kDepChunk, a chunk in rounds of 256): a bounded number of workgroups,
    // so that the pair census ends in a few thousand atomics on one address instead of one per wave (10^6 same-address atomics cost ~10 ms)
    constexpr int kRounds = KEYS ? kDepChunk / kSnapBlock : 1;
    const int64_t nunits = KEYS ? (int64_t)da.nchunks : (np + kSnapBlock - 1) / kSnapBlock;
    for (int64_t unit = blockIdx.x; unit < nunits; unit += gridDim.x) {
    if (KEYS) for (int i = tid; i < da.dg.B1; i += 256) dhist[i] = 0;          // (used in phase 3, two barriers from here)
    for (int rnd = 0; rnd < kRounds; ++rnd) {
        const int64_t base = (unit * kRounds + rnd) * kSnapBlock;
        if (base >= np) break;                                  // (the whole workgroup)
        // particle k of this thread: slot = k * 256 + tid (consecutive lanes = consecutive particles: coalesced)
        bool on[kSnapPPT];
        double x[kSnapPPT], y[kSnapPPT], z[kSnapPPT];
#pragma unroll
        for (int k = 0; k < kSnapPPT; ++k) {
            const int64_t p = base + k * 256 + tid;
            on[k] = p < np;
            x[k] = on[k] ? px[p * sin] : 0.0; y[k] = on[k] ? py[p * sin] : 0.0; z[k] = (DIM == 3 && on[k]) ? pz[p * sin] : 0.0;
        }
#pragma unroll
        for (int k = 0; k < kSnapPPT; ++k) {
            const int sl = k * 256 + tid;
            spos[0][sl] = x[k]; spos[1][sl] = y[k]; spos[2][sl] = z[k];
            sacc[0][sl] = 0.0; sacc[1][sl] = 0.0; sacc[2][sl] = 0.0;
        }
        if (tid == 0) qn = 0;
        __syncthreads();
        // ---- phase 1: the chains of the thread's particles advance together (all occupancy bits, then all cell starts, then a line of
        // entries for every particle that still has some)
        int e0[kSnapPPT], e1[kSnapPPT];
        int64_t cell[kSnapPPT];
        uint32_t bits[kSnapPPT];
#pragma unroll
        for (int k = 0; k < kSnapPPT; ++k) {
            e0[k] = 0; e1[k] = 0; cell[k] = 0; bits[k] = 0u;
            if (on[k]) {
                const bool inside = (x[k] >= 0.0 && x[k] <= g.L) && (y[k] >= 0.0 && y[k] <= g.L) && (DIM == 2 || (z[k] >= 0.0 && z[k] <= g.L));
                if (!inside) { atomicOr(flags, 2); on[k] = false; }
                else {
                    cell[k] = snap_cell_index(g, snap_cell(x[k], g), snap_cell(y[k], g), (DIM == 3) ? snap_cell(z[k], g) : 0);
                    bits[k] = bitmap[cell[k] >> 5];
                }
            }
        }
        bool outside[kSnapPPT];
#pragma unroll
        for (int k = 0; k < kSnapPPT; ++k) {
            outside[k] = !on[k] && (base + k * 256 + tid < np);
            if (on[k] && ((bits[k] >> (cell[k] & 31)) & 1u)) { e0[k] = cell_start[cell[k]]; e1[k] = cell_start[cell[k] + 1]; }
        }
        bool more = false;
#pragma unroll
        for (int k = 0; k < kSnapPPT; ++k) more = more || (e0[k] < e1[k]);
        while (more) {
            // four entries (one 64-byte line when aligned) per particle and trip: the loop waits for list entries, so its length in round
            // trips is what counts
            SnapEntry en[kSnapPPT][4];
#pragma unroll
            for (int k = 0; k < kSnapPPT; ++k)
                if (e0[k] < e1[k]) {
#pragma unroll
                    for (int q = 0; q < 4; ++q) en[k][q] = entries[min(e0[k] + q, e1[k] - 1)];
                }
            more = false;
#pragma unroll
            for (int k = 0; k < kSnapPPT; ++k)
                if (e0[k] < e1[k]) {
                    const float pf[3] = {(float)x[k], (float)y[k], (float)z[k]};
#pragma unroll
                    for (int q = 0; q < 4; ++q) if (e0[k] + q < e1[k]) test(k * 256 + tid, pf, en[k][q]);
                    e0[k] += 4;
                    more = more || (e0[k] < e1[k]);
                }
        }
        __syncthreads();
        // ---- phase 2: the queued hits, one per lane
        const int nq = min(qn, kSnapQueue);
        for (int i = tid; i < nq; i += 256) hit(qslot[i], qent[i]);
        __syncthreads();
        // ---- phase 3
#pragma unroll
        for (int k = 0; k < kSnapPPT; ++k) {
            const int64_t p = base + k * 256 + tid;
            const int sl = k * 256 + tid;
            double nx = 0.0, ny = 0.0, nz = 0.0;
            if (on[k] || outside[k]) {
                nx = x[k] + sacc[0][sl]; ny = y[k] + sacc[1][sl]; nz = z[k] + sacc[2][sl];         // :254-257
                if (nx > g.L) nx -= g.L;                                                     // :259-262
                if (nx < 0.0) nx += g.L;
                if (ny > g.L) ny -= g.L;
                if (ny < 0.0) ny += g.L;
                if (DIM == 3) {
                    if (nz > g.L) nz -= g.L;
                    if (nz < 0.0) nz += g.L;
                }
                if (!KEYS) {
                    ox[p * sout] = nx; oy[p * sout] = ny;
                    if (DIM == 3) oz[p * sout] = nz;
                }
            }
            if constexpr (KEYS) {
                // ParticleSnapshot.make_map's histogram bin of the displaced position (io.py:640-668; deposit_keys_kernel), every lane
                // of the wave (lds_hist_rank shuffles).  The search runs on `e` once per address space (see deposit_keys_kernel).
                const bool have = p < np;
                auto bins = [&](const auto *e) {
                    const int bx = histogram_bin(e, da.dg.nb, nx, dscale), by = histogram_bin(e, da.dg.nb, ny, dscale);
                    const int bz = (DIM == 3) ? histogram_bin(e, da.dg.nb, nz, dscale) : 0;
                    return (have && bx >= 0 && by >= 0 && bz >= 0) ? dep_key(da.dg, bx, by, bz) : kDepNoKey;
                };
                const uint32_t key = lds_edges ? bins(sedge) : bins(da.edges);
                lds_hist_rank(dhist, (int)((key >> kDepLocalBits) >> da.dg.lB2), key != kDepNoKey);
                if (have) da.keys[p] = key;
            }
        }
        __syncthreads();                                   // the LDS buffers are reused by the next block of particles
    }
    if (KEYS) {
        for (int i = tid; i < da.dg.B1; i += 256) da.wg_hist[(int64_t)i * da.nchunks + unit] = dhist[i];
        __syncthreads();                                   // (before the next chunk's zeroing)
    }
    }
    if (pair_total) {
        __shared__ unsigned long long wsum[256 / kWave];
#pragma unroll
        for (int s = kWave >> 1; s > 0; s >>= 1) npairs += __shfl_down(npairs, s, kWave);
        if ((threadIdx.x & (kWave - 1)) == 0) wsum[threadIdx.x / kWave] = npairs;
        __syncthreads();
        if (threadIdx.x == 0) {
            unsigned long long t = 0;
            for (int w = 0; w < 256 / kWave; ++w) t += wsum[w];
            if (t) atomicAdd(pair_total, t);
        }
    }
}

}  // namespace bfgx
