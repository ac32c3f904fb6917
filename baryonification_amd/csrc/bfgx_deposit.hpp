// ParticleSnapshot.make_map (np.histogramdd with weights, BaryonForge/utils/io.py:622-670) on gfx950, tile-owned form.
//
// One thread per particle + one fp64 global atomic per particle (particle_deposit_kernel, bfgx_grid.hpp) sits on the
// memory-side atomic rate: 6.7e7 particles in random order on a 512^3 grid = 3.75 ms, 1.8e10 atomics/s.  Here the grid is
// cut into tiles of 8192 cells (16 x 16 x 32, the last axis contiguous in memory; 64 x 128 in 2-D) that fit the LDS of
// one workgroup; the particles are brought into tile order by a two-level counting sort of 4-byte keys
// (tile << 13 | cell in tile), and every tile is then accumulated in LDS by ONE workgroup and stored once with plain
// row-contiguous stores: no global fp64 atomics, no zero-fill of the map.
//
//   deposit_keys_kernel    x, y, z -> key (histogram_bin per axis, exactly np.histogramdd's edge rules); per-workgroup LDS
//                          histogram of the level-1 buckets (groups of B2 consecutive tiles)
//   deposit_split_kernel   one level of the sort: a workgroup takes 4096 consecutive keys, orders them by bucket in LDS and
//                          copies the runs to its range of every bucket (level 1: ranges from the scanned per-workgroup
//                          histograms, no atomics; level 2: one returning atomic per (workgroup, tile))
//   deposit_count_kernel   level-2 counts (particles per tile) from the level-1-sorted keys
//   deposit_tiles_kernel   workgroup per tile: LDS accumulate (ds_add_f64), store the tile
#pragma once
#include "bfgx_grid.hpp"
#include <type_traits>

namespace bfgx {

constexpr int kDepLocalBits = 13;                 // 8192 cells per tile = 64 KB of fp64 accumulators
constexpr int kDepTileCells = 1 << kDepLocalBits;
constexpr int kDepChunk = 4096;                   // keys per workgroup in the sort passes (256 threads x 16)
constexpr int kDepPer = kDepChunk / 256;
constexpr int kDepTileThreads = 1024;             // 64 KB of LDS per tile: two workgroups per CU, so make them large
constexpr int kDepEdgesLds = 2049;                // edges staged in LDS up to 2048 cells per axis
#ifndef BFGX_DEP_PAD
#define BFGX_DEP_PAD 16
#endif
constexpr int kDepPad = BFGX_DEP_PAD;             // words between two tiles' level-2 cursors (a workgroup adds to the ~128 cursors of its bucket, the ~128 workgroups of
                                                  // that bucket run together: atomics to one cache line serialise)
constexpr int kDepHist = 2048;                    // LDS histogram entries of a sort pass (buckets a workgroup ranks locally)

struct DepGeom {
    int32_t dim, nb;                  // grid: nb^dim cells
    int32_t plane_lo, plane_n;        // slab of the FIRST axis that is deposited ([plane_lo, plane_lo + plane_n); the whole grid: 0, nb):
                                      // the output holds plane_n x nb (x nb) cells, particles of other planes are dropped
    int32_t sx, sy, sz;               // tile shape in cells
    int32_t ntx, nty, ntz;            // tiles per axis
    int32_t T, B1, B2;                // tiles; level-1 buckets; tiles per level-1 bucket (B1 * B2 >= T)
    int32_t lsx, lsy, lsz, lB2;       // log2 of the tile shape and of B2 (all powers of two: the per-particle index arithmetic is shifts,
                                      // seven integer divisions per particle were most of deposit_keys' instructions)
};

__host__ inline DepGeom dep_geom(int dim, int nb, int plane_lo, int plane_n)
{
    DepGeom g;
    g.dim = dim; g.nb = nb; g.plane_lo = plane_lo; g.plane_n = plane_n;
    if (dim == 3) { g.sx = 16; g.sy = 16; g.sz = 32; g.lsx = 4; g.lsy = 4; g.lsz = 5; }
    else { g.sx = 64; g.sy = 128; g.sz = 1; g.lsx = 6; g.lsy = 7; g.lsz = 0; }
    g.ntx = (plane_n + g.sx - 1) / g.sx;
    g.nty = (nb + g.sy - 1) / g.sy;
    g.ntz = (dim == 3) ? (nb + g.sz - 1) / g.sz : 1;
    g.T = g.ntx * g.nty * g.ntz;
    int b = 1, lb = 0;
    while ((int64_t)b * b < g.T) { b <<= 1; ++lb; }
    g.B2 = b; g.lB2 = lb; g.B1 = (g.T + b - 1) / b;
    return g;
}

// key of a cell: tile << 13 | index in tile (the last grid axis runs fastest inside the tile, as in memory)
__device__ inline uint32_t dep_key(const DepGeom &g, int bx, int by, int bz)
{
    const int tx = bx >> g.lsx, ty = by >> g.lsy, tz = (g.dim == 3) ? bz >> g.lsz : 0;
    const int lx = bx & (g.sx - 1), ly = by & (g.sy - 1), lz = (g.dim == 3) ? bz & (g.sz - 1) : 0;
    const uint32_t tile = (uint32_t)((tx * g.nty + ty) * g.ntz + tz);
    const uint32_t local = (uint32_t)((((lx << g.lsy) + ly) << g.lsz) + lz);
    return (tile << kDepLocalBits) | local;
}

constexpr uint32_t kDepNoKey = 0xffffffffu;        // particle outside the edges: dropped (np.histogramdd)

// hist[idx] += 1 for the active lanes of the wave, with ONE LDS atomic per run of consecutive lanes that name the same bin; returns the
// lane's rank in the bin (the count before its own add; -1 for an inactive lane).  Particles in random order give runs of one lane (and
// a dozen instructions more than a plain atomic); a snapshot as simulations write it -- ordered along a space-filling curve -- gives
// whole waves one bin, which as 64 same-address LDS atomics cost the sort passes 70 - 130 us each.  Every lane must call.
__device__ inline int lds_hist_rank(int *hist, int idx, bool active)
{
    const int lane = (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u));
    const int key = active ? idx : -1 - lane;
    const int prev = __shfl_up(key, 1, kWave);
    const bool head = (lane == 0) || (key != prev);
    const unsigned long long hm = __ballot(head);
    if (__popcll(hm) > 40) return active ? atomicAdd(hist + idx, 1) : -1;          // (wave-uniform) hardly any runs: one atomic per lane as before
    const int hl = 63 - __clzll((long long)(hm & ((2ull << lane) - 1ull)));
    const unsigned long long up = (lane == kWave - 1) ? 0ull : (hm >> (lane + 1));
    const int next = up ? lane + __ffsll((long long)up) : kWave;
    int base = 0;
    if (head && active) base = atomicAdd(hist + idx, next - lane);
    base = __shfl(base, hl, kWave);
    return active ? base + (lane - hl) : -1;
}

template <int DIM>
__global__ void __launch_bounds__(256)
deposit_keys_kernel(DepGeom g, int64_t n, const double *__restrict__ x, const double *__restrict__ y, const double *__restrict__ z,
                    const double *__restrict__ edges, uint32_t *__restrict__ keys, int32_t *__restrict__ wg_hist, int64_t stride)
{
    // stride: doubles between consecutive particles of a column (1: plain columns; > 1: fields of a record buffer)
    __shared__ int hist[1024];
    __shared__ double sedge[kDepEdgesLds];
    const int tid = threadIdx.x;
    for (int i = tid; i < g.B1; i += 256) hist[i] = 0;
    const bool lds_edges = g.nb + 1 <= kDepEdgesLds;                 // the bin search then never leaves the CU
    if (lds_edges) for (int i = tid; i <= g.nb; i += 256) sedge[i] = edges[i];
    __syncthreads();
    const double scale = (double)g.nb / (edges[g.nb] - edges[0]);
    const int64_t base = (int64_t)blockIdx.x * kDepChunk;
    // 8 particles per thread in flight: the loads of a batch are issued before any of the bin searches
    // the search runs on `e`: called once per address space so that the LDS copy is read with ds_read (a pointer selected at run time
    // between LDS and global memory is a flat pointer: every edge read then goes through the flat path)
    auto run = [&](const auto *e) {
        constexpr int kBatch = 8;
        for (int q0 = 0; q0 < kDepPer; q0 += kBatch) {
            double vx[kBatch], vy[kBatch], vz[kBatch];
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
                const int64_t p = base + (q0 + q) * 256 + tid;
                const bool on = p < n;
                vx[q] = on ? x[p * stride] : 0.0; vy[q] = on ? y[p * stride] : 0.0; vz[q] = (DIM == 3 && on) ? z[p * stride] : 0.0;
            }
#pragma unroll
            for (int q = 0; q < kBatch; ++q) {
                const int64_t p = base + (q0 + q) * 256 + tid;
                const bool on = p < n;
                int bx = histogram_bin(e, g.nb, vx[q], scale);
                const int by = histogram_bin(e, g.nb, vy[q], scale);
                const int bz = (DIM == 3) ? histogram_bin(e, g.nb, vz[q], scale) : 0;
                uint32_t key = kDepNoKey;
                bx = (bx >= g.plane_lo && bx < g.plane_lo + g.plane_n) ? bx - g.plane_lo : -1;
                const bool inside = on && bx >= 0 && by >= 0 && bz >= 0;
                if (inside) key = dep_key(g, bx, by, bz);
                lds_hist_rank(hist, (int)((key >> kDepLocalBits) >> g.lB2), inside);
                if (on) keys[p] = key;
            }
        }
    };
    if (lds_edges) run(sedge); else run(edges);
    __syncthreads();
    // per-workgroup histogram, bucket-major: an exclusive scan of this array hands every workgroup its range in every bucket
    // (global cursors would be 128 addresses hammered by 16 384 workgroups)
    for (int i = tid; i < g.B1; i += 256) wg_hist[(int64_t)i * gridDim.x + blockIdx.x] = hist[i];
}

// bucket of a key at LEVEL 1 (group of B2 tiles) / LEVEL 2 (tile)
template <int LEVEL>
__device__ inline int dep_bucket(const DepGeom &g, uint32_t key)
{
    const uint32_t tile = key >> kDepLocalBits;
    return LEVEL == 1 ? (int)(tile >> g.lB2) : (int)tile;
}

// level-2 counts: keys sorted by level-1 bucket -> particles per tile.  A chunk of consecutive keys spans few level-1
// buckets; tiles within kDepHist of the chunk's first bucket are counted in LDS, the others (only when buckets hold
// almost nothing) with global atomics.
__global__ void __launch_bounds__(256)
deposit_count_kernel(DepGeom g, const int32_t *__restrict__ nvalid, const uint32_t *__restrict__ keys, int32_t *__restrict__ count2)
{
    __shared__ int hist[kDepHist];
    __shared__ int bmin_s;
    const int tid = threadIdx.x;
    const int64_t n = *nvalid, base = (int64_t)blockIdx.x * kDepChunk;
    if (base >= n) return;
    for (int i = tid; i < kDepHist; i += 256) hist[i] = 0;
    if (tid == 0) bmin_s = (int)(((keys[base] >> kDepLocalBits) >> g.lB2) << g.lB2);
    __syncthreads();
    const int bmin = bmin_s;
#pragma unroll 4
    for (int q = 0; q < kDepPer; ++q) {
        const int64_t p = base + q * 256 + tid;
        const bool on = p < n;
        const int t = on ? (int)(keys[p] >> kDepLocalBits) : 0, l = t - bmin;
        const bool local = on && l >= 0 && l < kDepHist;
        lds_hist_rank(hist, l, local);
        if (on && !local) atomicAdd(count2 + t, 1);
    }
    __syncthreads();
    for (int i = tid; i < kDepHist; i += 256) if (hist[i]) atomicAdd(count2 + bmin + i, hist[i]);        // (padding these like the cursors below: no gain)
}

// one level of the counting sort (see the header).  n_dev: number of input keys as a device scalar (LEVEL 2: the valid ones).
// The workgroup's 4096 keys are first ordered by bucket in LDS, so that the copy to the buckets' ranges is made of
// contiguous runs written by neighbouring lanes.
template <int LEVEL, bool MASS>
__global__ void __launch_bounds__(256)
deposit_split_kernel(DepGeom g, int64_t n_host, const int32_t *__restrict__ n_dev, const uint32_t *__restrict__ keys_in,
                     const double *__restrict__ mass_in, const int32_t *__restrict__ start, int32_t *__restrict__ cursor,
                     uint32_t *__restrict__ keys_out, double *__restrict__ mass_out, int64_t mstride)
{
    // mstride: doubles between consecutive masses of mass_in (1 unless level 1 reads a field of a record buffer)
    __shared__ int hist[kDepHist];                  // per bucket: count -> local prefix -> (global base - local prefix)
    __shared__ uint32_t skey[kDepChunk];
    // masses travel in rounds of kMassRound through 8 KB of LDS: staging all 4096 of a workgroup (32 KB) left room for two workgroups per
    // CU, and the two passes took 0.93 ms each against 0.22 without masses
    constexpr int kMassRound = 1024;
    __shared__ double smass[MASS ? kMassRound : 1];
    __shared__ int part[256];
    __shared__ int bmin_s;
    const int tid = threadIdx.x;
    const int64_t n = n_dev ? (int64_t)*n_dev : n_host, base = (int64_t)blockIdx.x * kDepChunk;
    if (base >= n) return;
    for (int i = tid; i < kDepHist; i += 256) hist[i] = 0;
    if (tid == 0) bmin_s = (LEVEL == 1) ? 0 : (int)(((keys_in[base] >> kDepLocalBits) >> g.lB2) << g.lB2);
    __syncthreads();
    const int bmin = bmin_s;
    uint32_t key[kDepPer];
    int rank[kDepPer];                              // rank inside the workgroup's share of the bucket; -1: no key; -2: direct
#pragma unroll
    for (int q = 0; q < kDepPer; ++q) {
        const int64_t p = base + q * 256 + tid;
        key[q] = (p < n) ? keys_in[p] : kDepNoKey;
        const int l = (key[q] != kDepNoKey) ? dep_bucket<LEVEL>(g, key[q]) - bmin : -1;
        const bool local = l >= 0 && l < kDepHist;
        rank[q] = lds_hist_rank(hist, l, local);                       // (-1: no key)
        if (key[q] != kDepNoKey && !local) rank[q] = -2;
    }
    __syncthreads();
    // exclusive scan of the counts: thread t owns entries [8 t, 8 t + 8)
    constexpr int kOwn = kDepHist / 256;
    int cnt[kOwn], mine = 0;
#pragma unroll
    for (int e = 0; e < kOwn; ++e) { cnt[e] = hist[tid * kOwn + e]; mine += cnt[e]; }
    // (scan by shuffles inside the wave + the four wave totals through LDS: one barrier instead of the sixteen of a 256-entry Hillis-Steele scan)
    const int lane_ = tid & (kWave - 1), wid_ = tid / kWave;
    const int incl_ = wave_scan_incl(mine, lane_);
    if (lane_ == kWave - 1) part[wid_] = incl_;
    __syncthreads();
    int wpre_ = 0, total = 0;
#pragma unroll
    for (int w = 0; w < 256 / kWave; ++w) { const int pw = part[w]; total += pw; if (w < wid_) wpre_ += pw; }
    int pre = wpre_ + incl_ - mine;
    int lpre[kOwn];
#pragma unroll
    for (int e = 0; e < kOwn; ++e) { lpre[e] = pre; hist[tid * kOwn + e] = pre; pre += cnt[e]; }
    __syncthreads();
    // order the keys by bucket in LDS
    int slotq[MASS ? kDepPer : 1];                  // (masses: where each of the thread's keys went)
#pragma unroll
    for (int q = 0; q < kDepPer; ++q) {
        if (MASS) slotq[q] = -1;
        if (rank[q] >= 0) {
            const int slot = hist[dep_bucket<LEVEL>(g, key[q]) - bmin] + rank[q];
            skey[slot] = key[q];
            if (MASS) slotq[q] = slot;
        } else if (rank[q] == -2) {                 // bucket beyond the local table (almost empty buckets only)
            const int b = dep_bucket<LEVEL>(g, key[q]);
            const int64_t dst = (int64_t)start[b] + atomicAdd(cursor + (int64_t)b * (LEVEL == 2 ? kDepPad : 1), 1);
            keys_out[dst] = key[q];
            if (MASS) mass_out[dst] = mass_in[(base + q * 256 + tid) * mstride];
        }
    }
    __syncthreads();
    // one range per (workgroup, bucket); hist <- global base - local prefix
#pragma unroll
    for (int e = 0; e < kOwn; ++e) {
        const int i = tid * kOwn + e;
        if (cnt[e]) hist[i] = (LEVEL == 1 ? start[(int64_t)i * gridDim.x + blockIdx.x]          // scanned per-workgroup histogram
                                           : start[bmin + i] + atomicAdd(cursor + (int64_t)(bmin + i) * kDepPad, cnt[e])) - lpre[e];
    }
    __syncthreads();
    for (int i = tid; i < total; i += 256) {
        const uint32_t k = skey[i];
        const int64_t dst = (int64_t)hist[dep_bucket<LEVEL>(g, k) - bmin] + i;
        keys_out[dst] = k;
    }
    if (MASS) {
        // (the thread's sixteen masses are requested together, once; the rounds only move them through LDS)
        double mq[kDepPer];
#pragma unroll
        for (int q = 0; q < kDepPer; ++q) mq[q] = (slotq[q] >= 0) ? mass_in[(base + q * 256 + tid) * mstride] : 0.0;
        for (int r0 = 0; r0 < total; r0 += kMassRound) {
            __syncthreads();                                   // (the round before has been copied out)
#pragma unroll
            for (int q = 0; q < kDepPer; ++q)
                if (slotq[q] >= r0 && slotq[q] < r0 + kMassRound) smass[slotq[q] - r0] = mq[q];
            __syncthreads();
            const int nr = min(kMassRound, total - r0);
            for (int i = tid; i < nr; i += 256) {
                const uint32_t k = skey[r0 + i];
                mass_out[(int64_t)hist[dep_bucket<LEVEL>(g, k) - bmin] + r0 + i] = smass[i];
            }
        }
    }
}

// workgroup per tile: accumulate the tile's particles in LDS, store every cell of the tile once.  Unit masses (MASS = false) are COUNTED:
// 32-bit integer cells (ds_add_u32: 32 KB per tile instead of 64 KB of fp64 -- four workgroups of NT = 512 threads per CU instead of two of
// 1024, half the zeroing and read-out), converted on the way out -- exact, like the fp64 sum of ones it replaces.
#ifndef BFGX_DEP_TILE_NT
#define BFGX_DEP_TILE_NT 512
#endif
template <bool MASS> struct DepTileThreads { static constexpr int n = MASS ? kDepTileThreads : BFGX_DEP_TILE_NT; };
template <bool MASS>
__global__ void __launch_bounds__(DepTileThreads<MASS>::n)
deposit_tiles_kernel(DepGeom g, const int32_t *__restrict__ start2, const uint32_t *__restrict__ keys, const double *__restrict__ mass,
                     double *__restrict__ out, double *__restrict__ out2, double *__restrict__ tile_sums)
{
    // out2 / tile_sums (optional): a second copy of every cell and the tile's total -- what BaryonifyGrid's cell-owned pass starts from
    // (map_out = map_in, sum(map_in)), written here while the cell is in a register instead of by a pass over the finished map
    constexpr int NT = DepTileThreads<MASS>::n;
    using cell_t = typename std::conditional<MASS, double, uint32_t>::type;
    __shared__ cell_t acc[kDepTileCells];
    __shared__ double red[NT / kWave];
    const int tile = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < kDepTileCells; i += NT) acc[i] = (cell_t)0;
    __syncthreads();
    const int s0 = start2[tile], s1 = start2[tile + 1];
    for (int i = s0 + tid; i < s1; i += NT) {
        if constexpr (MASS) atomicAdd(acc + (keys[i] & (kDepTileCells - 1)), mass[i]);
        else atomicAdd(acc + (keys[i] & (kDepTileCells - 1)), 1u);
    }
    __syncthreads();
    const int tz = tile % g.ntz, ty = (tile / g.ntz) % g.nty, tx = tile / (g.ntz * g.nty);
    const int x0 = tx * g.sx, y0 = ty * g.sy, z0 = tz * g.sz;
    double tsum = 0.0;
    for (int l = tid; l < g.sx * g.sy * g.sz; l += NT) {
        const int lz = l & (g.sz - 1), ly = (l >> g.lsz) & (g.sy - 1), lx = l >> (g.lsz + g.lsy);
        const int bx = x0 + lx, by = y0 + ly, bz = z0 + lz;
        if (bx >= g.plane_n || by >= g.nb || (g.dim == 3 && bz >= g.nb)) continue;
        const int64_t flat = (g.dim == 3) ? ((int64_t)bx * g.nb + by) * g.nb + bz : (int64_t)bx * g.nb + by;
        const double v = (double)acc[l];
        out[flat] = v;                                     // (non-temporal stores measured: this kernel 448 -> 487 us, the FFT's first pass 431 -> 417)
        if (out2) out2[flat] = v;
        tsum += v;
    }
    if (tile_sums) {
#pragma unroll
        for (int sft = kWave >> 1; sft > 0; sft >>= 1) tsum += __shfl_down(tsum, sft, kWave);
        if ((tid & (kWave - 1)) == 0) red[tid / kWave] = tsum;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int w = 0; w < NT / kWave; ++w) t += red[w];
            tile_sums[tile] = t;
        }
    }
}

// ---- routing of particles to the ranks that own their planes (slab decomposition over GPUs, utils/GridSlabs.py) -----------------
// owner[j] = rank whose planes [r cnt, (r + 1) cnt) hold the particle's first-axis bin (np.histogramdd's rule), kRouteDropped for a
// particle outside the edges.  A few hundred workgroups, each owning a contiguous run of particles and touching the `world`
// counters once (same-address atomics serialise).
constexpr int kPartRouteGrid = 1024;
constexpr int kPartRouteMaxRanks = 255;
constexpr uint8_t kRouteDropped = 255;

__global__ void __launch_bounds__(256)
route_particles_count_kernel(int64_t n, const double *__restrict__ x, const double *__restrict__ edges, int nb, int cnt, int world,
                             uint8_t *__restrict__ owner, int32_t *__restrict__ counts)
{
    __shared__ int hist[256];
    __shared__ double sedge[kDepEdgesLds];
    const int tid = threadIdx.x;
    hist[tid] = 0;
    const bool lds_edges = nb + 1 <= kDepEdgesLds;
    if (lds_edges) for (int i = tid; i <= nb; i += 256) sedge[i] = edges[i];
    __syncthreads();
    const double scale = (double)nb / (edges[nb] - edges[0]);
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = min(n, j0 + per);
    auto run = [&](const auto *e) {                     // (once per address space: see deposit_keys_kernel)
        for (int64_t j = j0 + tid; j < j1; j += 256) {
            const int b = histogram_bin(e, nb, x[j], scale);
            const int d = (b >= 0) ? min(b / cnt, world - 1) : (int)kRouteDropped;
            owner[j] = (uint8_t)d;
            if (b >= 0) atomicAdd(&hist[d], 1);
        }
    };
    if (lds_edges) run(sedge); else run(edges);
    __syncthreads();
    if (tid < world && hist[tid]) atomicAdd(counts + tid, hist[tid]);
}

// cols_out[c][start[d] + ...] = column c of the particles bound for rank d (any order inside a destination); `total` = particles kept
struct PartRouteArgs {
    int32_t world, ncols;
    int64_t start[kPartRouteMaxRanks];
    const double *col[4];
};

__global__ void __launch_bounds__(256)
route_particles_fill_kernel(PartRouteArgs a, int64_t n, int64_t total, const uint8_t *__restrict__ owner, int32_t *__restrict__ cursor,
                            double *__restrict__ cols_out)
{
    __shared__ int hist[256], base[256], taken[256];
    const int tid = threadIdx.x;
    hist[tid] = 0; taken[tid] = 0;
    __syncthreads();
    const int64_t per = (n + gridDim.x - 1) / gridDim.x;
    const int64_t j0 = (int64_t)blockIdx.x * per, j1 = min(n, j0 + per);
    for (int64_t j = j0 + tid; j < j1; j += 256) {
        const int d = owner[j];
        if (d != kRouteDropped) atomicAdd(&hist[d], 1);
    }
    __syncthreads();
    if (tid < a.world) base[tid] = hist[tid] ? atomicAdd(cursor + tid, hist[tid]) : 0;       // one range per (workgroup, destination)
    __syncthreads();
    for (int64_t j = j0 + tid; j < j1; j += 256) {
        const int d = owner[j];
        if (d == kRouteDropped) continue;
        const int64_t pos = a.start[d] + base[d] + atomicAdd(&taken[d], 1);
        for (int c = 0; c < a.ncols; ++c) cols_out[(int64_t)c * total + pos] = a.col[c][j];
    }
}

}  // namespace bfgx
