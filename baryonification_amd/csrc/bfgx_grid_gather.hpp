// bfgx_grid_gather.hpp -- cell-owned BaryonifyGrid for gfx950: halo loop + regrid without a pix_offsets array.
//
//   grid_block_lists_kernel   halo -> the blocks of cells its ball's bounding box touches (count, then fill)
//   grid_gather_regrid_kernel one workgroup per block of cells: every cell sums the offsets of the halos listed for its block
//                             in registers (Map2DRunner.py:519-575, same arithmetic per (halo, pixel) pair as grid_scatter_kernel)
//                             and is regridded at once (:577-599 + regrid_pixels_2D/3D :14-163)
//
// Why: on a 512^3 grid pix_offsets is 3.2 GB.  The scatter formulation zeroes it (0.65 ms), adds into it with 3 global fp64
// atomics per pair (1.04 ms) and reads it back in the regrid (3.2 of the 4.3 GB that kernel moves, 1.83 ms): 3.5 ms.  A cell that
// owns its offset needs none of that traffic: the step reads map_in once and adds into map_out, which both formulations must do
// (2.0 ms + 0.05 ms of list building on the same workload, BASELINE config 5).
//
// The reference adds the halos' contributions to a pixel in catalog order; fp64 sums in a different order differ by a few ulp
// (the atomics of the scatter kernels have no fixed order either).  Here a block's list is filled through an atomic cursor, so
// the order of the sum inside a cell can vary from run to run at that level.
#pragma once
#include "bfgx_grid.hpp"

namespace bfgx {

// blocks of cells: 8 x 8 x 8 (3D, two cells per thread) or 16 x 16 (2D)
template <int DIM> struct GatherBlk {
    static constexpr int S = (DIM == 3) ? 3 : 4;         // log2(cells per axis)
    static constexpr int B = 1 << S;
    static constexpr int cells = (DIM == 3) ? B * B * B : B * B;
    static constexpr int per_thread = cells / 256;
};
constexpr int kGatherBatch = 8;                          // halo records staged in LDS per round

__host__ __device__ inline int gather_blocks_per_axis(int N, int S) { return (N + (1 << S) - 1) >> S; }

// the blocks one axis of a (periodic) pixel range touches: block(k) for k in [0, count)
struct AxisBlocks {
    int first, count, split;
    __device__ int block(int k) const { return k < split ? k : first + (k - split); }
};
__device__ inline AxisBlocks axis_blocks(int p0, int len, int N, int S)
{
    // p0: first pixel, unwrapped, in [-N, 2N); len >= 1 pixels
    const int nbk = gather_blocks_per_axis(N, S);
    if (len >= N) return AxisBlocks{0, nbk, nbk};
    int a = p0;
    a += (a < 0) ? N : 0; a -= (a >= N) ? N : 0;
    int b = a + len - 1;
    if (b < N) return AxisBlocks{a >> S, (b >> S) - (a >> S) + 1, 0};
    b -= N;                                              // wrapped: pixels [a, N) and [0, b]
    if ((a >> S) <= (b >> S)) return AxisBlocks{0, nbk, nbk};
    const int nlo = (b >> S) + 1;
    return AxisBlocks{a >> S, nlo + nbk - (a >> S), nlo};
}

// one WAVE per halo, lanes over the blocks of its bounding box.  FILL = 0: count[block] += 1; FILL = 1: the halo is appended to
// every block's list (list[start[block] + cursor[block]++])
template <int DIM, int FILL>
__global__ void __launch_bounds__(256)
grid_block_lists_kernel(GridGeom g, int64_t nh, const GridHaloRec *__restrict__ recs, int32_t *__restrict__ count,
                        const int32_t *__restrict__ start, int32_t *__restrict__ list)
{
    const int64_t j = (int64_t)blockIdx.x * (256 / kWave) + threadIdx.x / kWave;
    const int lane = threadIdx.x & (kWave - 1);
    if (j >= nh) return;
    const GridHaloRec &r = recs[j];
    if (r.nsize < 2 || r.nchunks == 0) return;           // skipped halo / empty bounding box
    constexpr int S = GatherBlk<DIM>::S;
    const int N = g.npix, nbk = gather_blocks_per_axis(N, S), wdt = r.nsize >> 1;
    AxisBlocks ab[3] = {{0, 1, 0}, {0, 1, 0}, {0, 1, 0}};
#pragma unroll
    for (int ax = 0; ax < DIM; ++ax) ab[ax] = axis_blocks(r.cen[ax] - wdt + r.lo[ax], r.n[ax], N, S);
    const int total = ab[0].count * ab[1].count * ab[2].count;
    for (int t = lane; t < total; t += kWave) {
        const int k2 = t % ab[2].count, q = t / ab[2].count, k1 = q % ab[1].count, k0 = q / ab[1].count;
        const int64_t b = (DIM == 3) ? ((int64_t)ab[0].block(k0) * nbk + ab[1].block(k1)) * nbk + ab[2].block(k2)
                                     : (int64_t)ab[0].block(k0) * nbk + ab[1].block(k1);
        const int32_t at = atomicAdd(count + b, 1);
        if (FILL) list[start[b] + at] = (int32_t)j;
    }
}

template <int DIM, int NC>
__global__ void __launch_bounds__(256, 6)
grid_gather_regrid_kernel(PairTable pt, GridGeom g, const GridHaloRec *__restrict__ recs, const int32_t *__restrict__ blk_start,
                          const int32_t *__restrict__ blk_list, const int32_t *__restrict__ blk_nz, const double *__restrict__ map_in,
                          double *__restrict__ map_out, double *__restrict__ block_sums, float axis0_limit, int32_t *__restrict__ axis0_flag)
{
    // (axis0_flag: set when a cell moves further than axis0_limit cells along the FIRST array axis -- the streamed host entry gathers the map in
    // ranges of planes and must know when a deposit could leave the planes it has in hand; nullptr: not wanted)
    // Two phases per batch of kGatherBatch halos.  (1) every thread tests its cells against the halos' bounding boxes and a
    // conservative r^2 bound (integer + a few fp64 operations per (halo, cell)) and queues the survivors in LDS -- a ball fills a
    // few per cent of the blocks it touches, so running the table readout under that test would leave most lanes idle.  (2) the
    // queue is drained with all lanes busy: exact cut, readout, and the three offset components added into the block's LDS
    // accumulators (ds_add_f64).
    using Blk = GatherBlk<DIM>;
    constexpr int S = Blk::S, CPT = Blk::per_thread;
    __shared__ GridHaloRec R[kGatherBatch];
    __shared__ double oacc[Blk::cells * DIM];
    __shared__ uint16_t queue[kGatherBatch * Blk::cells];                // (halo of the batch << 12) | cell: cannot overflow
    __shared__ int qn;
    const int N = g.npix, nbk = gather_blocks_per_axis(N, S), tid = threadIdx.x, lane = tid & (kWave - 1);
    const unsigned blk = (unsigned)blk_nz[blockIdx.x];    // launched over the blocks that list a halo only
    int c0, c1, c2 = 0;                                  // first cell of the block along the array axes (32-bit divisions)
    if (DIM == 3) { c2 = (int)(blk % (unsigned)nbk) << S; const unsigned q = blk / (unsigned)nbk; c1 = (int)(q % (unsigned)nbk) << S; c0 = (int)(q / (unsigned)nbk) << S; }
    else { c1 = (int)(blk % (unsigned)nbk) << S; c0 = (int)(blk / (unsigned)nbk) << S; }
    auto cell_pixel = [&](int cell, int pc[3]) {
        if (DIM == 3) { pc[2] = c2 + (cell & (Blk::B - 1)); pc[1] = c1 + ((cell >> S) & (Blk::B - 1)); pc[0] = c0 + (cell >> (2 * S)); }
        else { pc[2] = 0; pc[1] = c1 + (cell & (Blk::B - 1)); pc[0] = c0 + (cell >> S); }
    };
    // cutout index of a pixel along every axis (pick_indices inverted: pixel = (cen - wdt + index) mod N); false: outside the
    // bounding box of the ball
    auto cutout_index = [&](const GridHaloRec &r, const int pc[3], int idx[3]) -> bool {
        const int wdt = r.nsize >> 1;
#pragma unroll
        for (int ax = 0; ax < DIM; ++ax) {
            int i = pc[ax] - (r.cen[ax] - wdt);
            i += (i < 0) ? N : 0; i -= (i >= N) ? N : 0;
            if ((unsigned)(i - r.lo[ax]) >= (unsigned)r.n[ax]) return false;
            idx[ax] = i;
        }
        if (DIM == 2) idx[2] = 0;
        return true;
    };
    int pc[CPT][3];
    bool live[CPT];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        cell_pixel(tid + 256 * c, pc[c]);
        live[c] = pc[c][0] < N && pc[c][1] < N && pc[c][2] < N;
    }
    unsigned long long npairs = 0;
    const int s0 = blk_start[blk], s1 = blk_start[blk + 1];
    if (s1 > s0) {
        for (int t = tid; t < Blk::cells * DIM; t += 256) oacc[t] = 0.0;
        if (tid == 0) qn = 0;
    }
    for (int base = s0; base < s1; base += kGatherBatch) {
        const int nb = min(kGatherBatch, s1 - base);
        __syncthreads();
        // head of the record + the NC corner weights and row offsets this table has (not all kNCmax: 50 of 86 words for NC = 4)
        constexpr int kHead = (int)(offsetof(GridHaloRec, w) / 4), kUsed = kHead + 3 * NC;
        for (int t = tid; t < nb * kUsed; t += 256) {
            const int h = t / kUsed, k = t - h * kUsed;
            const int w = (k < kHead + 2 * NC) ? k : k + 2 * (kNCmax - NC);
            reinterpret_cast<int32_t *>(&R[h])[w] = reinterpret_cast<const int32_t *>(recs + blk_list[base + h])[w];
        }
        __syncthreads();
        for (int h = 0; h < nb; ++h) {
            const GridHaloRec &r = R[h];
            const double lim2 = r.ell ? __builtin_inf() : r.rcut * r.rcut * (1.0 + 1e-12);    // r < rcut implies r^2 < lim2
#pragma unroll
            for (int c = 0; c < CPT; ++c) {
                bool hit = false;
                int idx[3];
                if (live[c] && cutout_index(r, pc[c], idx)) {
                    const double Y = cutout_coord(idx[0], r.nsize, r.step, r.start, r.top, g.res) + r.dax[0];
                    const double X = cutout_coord(idx[1], r.nsize, r.step, r.start, r.top, g.res) + r.dax[1];
                    const double Z = (DIM == 3) ? cutout_coord(idx[2], r.nsize, r.step, r.start, r.top, g.res) + r.dax[2] : 0.0;
                    hit = !(X * X + Y * Y + Z * Z >= lim2);
                }
                const unsigned long long mask = __ballot(hit);
                if (mask) {                                  // one LDS atomic per wave
                    int at = 0;
                    if (lane == __ffsll((long long)mask) - 1) at = atomicAdd(&qn, __popcll(mask));
                    at = __shfl(at, __ffsll((long long)mask) - 1, kWave);
                    if (hit) queue[at + __popcll(mask & ((1ull << lane) - 1ull))] = (uint16_t)((h << 12) | (tid + 256 * c));
                }
            }
        }
        __syncthreads();
        const int nq = qn;
        for (int e = tid; e < nq; e += 256) {
            const uint32_t ent = queue[e];
            const int cell = (int)(ent & 0xfffu);
            const GridHaloRec &r = R[ent >> 12];
            int pq[3], idx[3];
            cell_pixel(cell, pq);
            (void)cutout_index(r, pq, idx);
            // from here: the arithmetic of grid_scatter_kernel<DIM, 0, NC> for the pair (halo, pixel)
            const double Y = cutout_coord(idx[0], r.nsize, r.step, r.start, r.top, g.res) + r.dax[0];
            const double X = cutout_coord(idx[1], r.nsize, r.step, r.start, r.top, g.res) + r.dax[1];
            const double Z = (DIM == 3) ? cutout_coord(idx[2], r.nsize, r.step, r.start, r.top, g.res) + r.dax[2] : 0.0;
            double r2 = add_nc(mul_nc(X, X), mul_nc(Y, Y));
            if (DIM == 3) r2 = add_nc(r2, mul_nc(Z, Z));
            const double rr = __dsqrt_rn(r2);
            double r_eval = rr;
            if (DIM == 2 && r.ell) {
                const double Xe = X * r.rmat[0] + Y * r.rmat[2], Ye = X * r.rmat[1] + Y * r.rmat[3];
                r_eval = __dsqrt_rn(add_nc(mul_nc(Xe, Xe), mul_nc(Ye, Ye)));
            }
            if (!(r_eval < r.rcut)) continue;                                      // BaryonCorrection.py:381-382
            const double lx = log(r_eval) + r.lnoff;
            const double d = r.oob ? __builtin_nan("") : radial_readout<NC>(pt, r.rowoff, r.w, lx);
            if (d == 0.0) continue;
            ++npairs;
            const double off = d / g.res;                                          // :534, :569
            atomicAdd(&oacc[cell * DIM + 0], mul_nc(off, X / rr));                 // pix_offsets[inds] += ...
            atomicAdd(&oacc[cell * DIM + 1], mul_nc(off, Y / rr));
            if (DIM == 3) atomicAdd(&oacc[cell * DIM + 2], mul_nc(off, Z / rr));
        }
        __syncthreads();
        if (tid == 0) qn = 0;
    }
    double o[CPT][3];
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        o[c][0] = o[c][1] = o[c][2] = 0.0;
        if (s1 > s0) {
#pragma unroll
            for (int q = 0; q < DIM; ++q) o[c][q] = oacc[(tid + 256 * c) * DIM + q];
        }
    }
    // the regrid of the block's cells (grid_regrid_kernel, whole-grid case).  map_out already holds a copy of map_in
    // (grid_copy_sum_kernel): a cell that is not displaced deposits into itself with weight 1, i.e. it is done -- only the ~10 % of
    // cells inside a ball take their value out again and spread it over their 2^d overlaps.  Deposits that land INSIDE the block (most
    // of them: the displacements are fractions of a cell) are summed in LDS, together with the -v of the moved cell, and reach map_out
    // as ONE global atomic per touched cell; only the deposits that leave the block are global atomics of their own (1.2e8 -> ~4.5e7
    // fp64 atomics on a 512^3 grid; the atomics are what bounds this half of the kernel).  sum_in here = the sum of (deposited -
    // source) over the moved cells.
    double *lacc = oacc;                                 // the offsets are in registers now: the LDS is free for the block's own cells
    __syncthreads();
    for (int t = tid; t < Blk::cells; t += 256) lacc[t] = 0.0;
    __syncthreads();
    // cell (i, j, k) of the grid -> index in the block, or -1 when outside (i along the first array axis)
    auto in_block = [&](int i, int j, int k) -> int {
        const unsigned di = (unsigned)(i - c0), dj = (unsigned)(j - c1), dk = (DIM == 3) ? (unsigned)(k - c2) : 0u;
        if (di >= (unsigned)Blk::B || dj >= (unsigned)Blk::B || dk >= (unsigned)Blk::B) return -1;
        return (DIM == 3) ? (int)((di << (2 * S)) | (dj << S) | dk) : (int)((di << S) | dj);
    };
    double sum_in = 0.0, sum_out = 0.0;
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        if (!live[c]) continue;
#pragma unroll
        for (int q = 0; q < DIM; ++q) if (!isfinite(o[c][q])) o[c][q] = 0.0;       // :580 / :591
        if (o[c][0] == 0.0 && o[c][1] == 0.0 && o[c][2] == 0.0) continue;
        if (axis0_flag != nullptr && fabs(o[c][1]) > (double)axis0_limit) atomicOr(axis0_flag, 1);      // (pos[1] moves along the first array axis; rare)
        const int64_t p = (DIM == 3) ? ((int64_t)pc[c][0] * N + pc[c][1]) * N + pc[c][2] : (int64_t)pc[c][0] * N + pc[c][1];
        const double v = map_in[p];
        if (v == 0.0) continue;                          // an empty cell adds exactly nothing
        // pos[0] moves along the SECOND array axis (j), pos[1] along the first (i), pos[2] along the third (k): deposit_cell's arithmetic
        const AxisSplit sx = split_axis(o[c][0] + (double)pc[c][1], N), sy = split_axis(o[c][1] + (double)pc[c][0], N);
        const AxisSplit sz = (DIM == 3) ? split_axis(o[c][2] + (double)pc[c][2], N) : AxisSplit{{0, 0}, {1.0, 0.0}};
        double dep = 0.0;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int cz = 0; cz < (DIM == 3 ? 2 : 1); ++cz) {
                    const double dy = sy.w[a], dx = sx.w[b], dz = sz.w[cz];
                    if (!(dx > 0.0 && dy > 0.0 && dz > 0.0)) continue;
                    const double w = (DIM == 3) ? mul_nc(mul_nc(mul_nc(dx, dy), dz), v) : mul_nc(mul_nc(dx, dy), v);
                    const int lb = in_block(sy.cell[a], sx.cell[b], sz.cell[cz]);
                    if (lb >= 0) atomicAdd(lacc + lb, w);                          // ds_add_f64
                    else atomicAdd(map_out + ((DIM == 3) ? ((int64_t)sy.cell[a] * N + sx.cell[b]) * N + sz.cell[cz]
                                                          : (int64_t)sy.cell[a] * N + sx.cell[b]), w);
                    dep += w;
                }
        atomicAdd(lacc + (tid + 256 * c), -v);
        sum_out += dep - v;
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CPT; ++c) {
        const double d = lacc[tid + 256 * c];
        if (live[c] && d != 0.0) {
            const int64_t p = (DIM == 3) ? ((int64_t)pc[c][0] * N + pc[c][1]) * N + pc[c][2] : (int64_t)pc[c][0] * N + pc[c][1];
            atomicAdd(map_out + p, d);                   // (other blocks may be depositing into this cell: atomic)
        }
    }
    __shared__ double sa[256 / kWave], sb[256 / kWave];
    __shared__ unsigned long long sn[256 / kWave];
#pragma unroll
    for (int s = kWave >> 1; s > 0; s >>= 1) {
        sum_in += __shfl_down(sum_in, s, kWave); sum_out += __shfl_down(sum_out, s, kWave); npairs += __shfl_down(npairs, s, kWave);
    }
    const int wid = tid / kWave;
    if (lane == 0) { sa[wid] = sum_in; sb[wid] = sum_out; sn[wid] = npairs; }
    __syncthreads();
    if (tid == 0) {
        double ta = 0.0, tb = 0.0;
        unsigned long long tn = 0;
        for (int w = 0; w < 256 / kWave; ++w) { ta += sa[w]; tb += sb[w]; tn += sn[w]; }
        // (no atomic on one pair counter here: 2.6e5 workgroups adding to the same address serialise -- 1.2 ms on a 512^3 grid)
        block_sums[3 * (size_t)blk] = ta; block_sums[3 * (size_t)blk + 1] = tb; block_sums[3 * (size_t)blk + 2] = (double)tn;
    }
}

// nz[0 .. *nnz) = the blocks whose halo list is not empty (any order): one atomic per workgroup on the counter
__global__ void __launch_bounds__(256)
grid_nonempty_blocks_kernel(int64_t blk_lo, int64_t nblk, const int32_t *__restrict__ start, int32_t *__restrict__ nz, int32_t *__restrict__ nnz)
{
    // (the blocks [blk_lo, nblk) that list a halo -> nz[0 ..), their number -> *nnz)
    __shared__ int s_n, s_base;
    __shared__ int32_t s_list[256];
    for (int64_t b0 = blk_lo + (int64_t)blockIdx.x * 256; b0 < nblk; b0 += (int64_t)gridDim.x * 256) {
        if (threadIdx.x == 0) s_n = 0;
        __syncthreads();
        const int64_t b = b0 + threadIdx.x;
        if (b < nblk && start[b + 1] > start[b]) s_list[atomicAdd(&s_n, 1)] = (int32_t)b;
        __syncthreads();
        if (threadIdx.x == 0 && s_n) s_base = atomicAdd(nnz, s_n);
        __syncthreads();
        if ((int)threadIdx.x < s_n) nz[s_base + threadIdx.x] = s_list[threadIdx.x];
        __syncthreads();
    }
}

// map_out = map_in (every cell deposits into itself unless the halo loop moves it); wg_sums[workgroup] = its part of sum(map_in)
__global__ void __launch_bounds__(256)
grid_copy_sum_kernel(int64_t n, const double *__restrict__ in, double *__restrict__ out, double *__restrict__ wg_sums)
{
    __shared__ double sw[256 / kWave];
    double acc = 0.0;
    if (((reinterpret_cast<uintptr_t>(in) | reinterpret_cast<uintptr_t>(out)) & 15) == 0) {      // 16-byte accesses
        const int64_t n2 = n >> 1;
        const double2 *in2 = reinterpret_cast<const double2 *>(in);
        double2 *out2 = reinterpret_cast<double2 *>(out);
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n2; i += (int64_t)gridDim.x * 256) {
            const double2 v = in2[i];
            out2[i] = v;
            acc += v.x + v.y;
        }
        if ((n & 1) && blockIdx.x == 0 && threadIdx.x == 0) { out[n - 1] = in[n - 1]; acc += in[n - 1]; }
    } else {                                                 // a caller's map at an odd multiple of 8 bytes
        for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) { const double v = in[i]; out[i] = v; acc += v; }
    }
#pragma unroll
    for (int s = kWave >> 1; s > 0; s >>= 1) acc += __shfl_down(acc, s, kWave);
    if ((threadIdx.x & (kWave - 1)) == 0) sw[threadIdx.x / kWave] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) t += sw[w];
        wg_sums[blockIdx.x] = t;
    }
}

// sums[0] += sum(map_in) (the copy's per-workgroup sums), sums[1] += sum(map_in) + the moved cells' (deposited - source) (sums
// optional); *pair_total += contributing pairs (exact: < 2^53)
__global__ void __launch_bounds__(256)
gather_sums_kernel(int64_t nblocks, const double *__restrict__ block_sums, int ncopy, const double *__restrict__ copy_sums,
                   double *__restrict__ sums, unsigned long long *__restrict__ pair_total)
{
    __shared__ double sa[256 / kWave], sb[256 / kWave], sc[256 / kWave];
    double xa = 0.0, xb = 0.0, xc = 0.0;
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < nblocks; b += (int64_t)gridDim.x * 256) {
        xb += block_sums[3 * b + 1]; xc += block_sums[3 * b + 2];
    }
    for (int64_t b = (int64_t)blockIdx.x * 256 + threadIdx.x; b < ncopy; b += (int64_t)gridDim.x * 256) { xa += copy_sums[b]; xb += copy_sums[b]; }
#pragma unroll
    for (int s = kWave >> 1; s > 0; s >>= 1) { xa += __shfl_down(xa, s, kWave); xb += __shfl_down(xb, s, kWave); xc += __shfl_down(xc, s, kWave); }
    const int lane = threadIdx.x & (kWave - 1), wid = threadIdx.x / kWave;
    if (lane == 0) { sa[wid] = xa; sb[wid] = xb; sc[wid] = xc; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double ta = 0.0, tb = 0.0, tc = 0.0;
        for (int w = 0; w < 256 / kWave; ++w) { ta += sa[w]; tb += sb[w]; tc += sc[w]; }
        if (sums) { atomicAdd(sums + 0, ta); atomicAdd(sums + 1, tb); }
        if (tc > 0.0) atomicAdd(pair_total, (unsigned long long)tc);
    }
}

}  // namespace bfgx
