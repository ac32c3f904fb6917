// C ABI of libbfgx.so (see include/bfgx.h for the reference lines each entry point replaces).
// Host code only orchestrates: uploads the model once per plan, then enqueues
//   K0 halo_prep -> K1 halo_scatter<OFFSETS> -> K2 regrid (+ sum2)      (BaryonifyShell.process)
//   K0 halo_prep -> K3 halo_scatter<PAINT>                               (PaintProfilesShell.process)
// on one HIP stream.  There is no CPU fallback: without a visible GPU every compute entry point
// returns BFGX_ERR_NO_DEVICE.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <algorithm>
#include <climits>
#include <cstdint>
#include <cstdlib>
#include <chrono>
#include <cstring>
#include <functional>
#include <map>
#include <mutex>
#include <string>
#include <tuple>
#include <type_traits>
#include <vector>

#include "../../include/bfgx.h"

// every device allocation the library makes is counted (bfgx_debug_alloc_count): a warm one-shot call must make none
#include <atomic>
static std::atomic<long long> g_bfgx_allocs{0};
static std::atomic<long long> g_host_pinned_in_place{0}, g_host_staged{0}, g_host_pin_min_bytes{-1};      // HostSpan (bfgx_debug_host_spans)
template <typename T> static inline hipError_t bfgx_counted_malloc(T **p, size_t bytes) { ++g_bfgx_allocs; return hipMalloc((void **)p, bytes); }
#define hipMalloc(ptr, bytes) bfgx_counted_malloc(ptr, bytes)
#include "bfgx_cosmo.hpp"
#include "bfgx_kernels.hpp"
#include "bfgx_scatter2.hpp"
#include "bfgx_regrid2.hpp"
#ifndef BFGX_NP
#define BFGX_NP 1          // pairs per lane per trip of the fast kernel's pair loop
#endif
#include "bfgx_tables.hpp"
#include "bfgx_grid.hpp"
#include "bfgx_fft.hpp"
#include "bfgx_deposit.hpp"
#include "bfgx_snapshot.hpp"
#include "bfgx_grid_gather.hpp"
#include "bfgx_fftlog.hpp"

using namespace bfgx;

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof(buf), fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                          \
    do {                                                                                       \
        hipError_t e_ = (expr);                                                                \
        if (e_ != hipSuccess)                                                                  \
            return fail(BFGX_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), __FILE__, __LINE__); \
    } while (0)

int validate_cosmo(const bfgx_cosmo &c)
{
    if (!(c.Omega_m > 0) || !(c.h > 0) || !(c.Omega_b >= 0) || c.w0 != c.w0)
        return fail(BFGX_ERR_INVALID, "cosmology needs Omega_m > 0, h > 0, Omega_b >= 0, finite w0");
    return BFGX_OK;
}

int validate_table(const bfgx_table &t)
{
    if (t.ndim < 3 || t.ndim > BFGX_MAX_DIM) return fail(BFGX_ERR_INVALID, "table ndim must be in [3, %d]", BFGX_MAX_DIM);
    if (!t.values) return fail(BFGX_ERR_INVALID, "table values pointer is NULL");
    for (int d = 0; d < t.ndim; ++d) {
        if (t.n[d] < 2) return fail(BFGX_ERR_INVALID, "table axis %d needs >= 2 points", d);
        if (!t.axis[d]) return fail(BFGX_ERR_INVALID, "table axis %d pointer is NULL", d);
        for (int i = 1; i < t.n[d]; ++i)
            if (!(t.axis[d][i] > t.axis[d][i - 1]))
                return fail(BFGX_ERR_INVALID, "table axis %d must be strictly ascending", d);
    }
    double tot = 1.0;
    for (int d = 0; d < t.ndim; ++d) tot *= t.n[d];
    if (tot > 2.0e9) return fail(BFGX_ERR_INVALID, "table too large for 32-bit row offsets");
    return BFGX_OK;
}

}  // namespace

struct bfgx_plan {
    int device = 0, num_cus = 256;
    hipStream_t stream = nullptr;
    int64_t nside = 0, max_halos = 0;
    Hpx hpx;
    DevModel model;
    std::vector<void *> owned;     // device allocations freed with the plan
    HaloRec *recs = nullptr;
    RowSetX *rowsx = nullptr;      // corner rows when the table has extra parameter axes (NC > 4)
    int NC = 4;
    // tile-owned accumulation (algo 1): tiling tables + halo -> tile binning workspace
    int algo = 1;
    bool blocking_growth = false;   // one-shot host API: grow the entry list on overflow (needs a sync)
    float *k0_work_est = nullptr;   // per K0 workgroup: estimated pixels of its narrow halos (k1_form_kernel)
    int32_t *k1_form = nullptr;     // device word: the fast kernel's form when the halo count does not decide it (1 fluid, 0 barrier per tile)
    int64_t k1_nhalos = 0;          // halos of the catalog the binning step has just listed (the fast kernel's form follows their density)
    int k1_wide = 1;                // the fast kernel takes the WIDE discs too (a pole inside, pixels beyond 0.40 rad of the halo's azimuth: RowRec.fb bit 1) and the generic kernel's wide pass is not launched; BFGX_K1_WIDE=0 at plan creation: they go through the wide pass (tests: the two must agree)
    int k1_fluid = 1;               // the fast kernel's fluid form (bfgx_scatter2.hpp): 1 = from 2 tiles per CU; BFGX_K1_FLUID at plan creation: 0 = never (the barrier-per-tile form), 2 = always (tests)
    Tiling tiling;
    int32_t *tile_count = nullptr, *tile_count_b = nullptr, *tile_count_w = nullptr, *tile_start = nullptr, *tile_cursor = nullptr,
            *tile_cursor_w = nullptr, *entries = nullptr, *overflow = nullptr;
    // narrow halos over <= 4 tiles are placed by K0 itself into fixed-capacity lists [ntiles][entries_a_cap]; the others are listed per K0
    // workgroup for the placement pass; tile_zeros: ntiles zeros (the shared entry list holds only the regions [B | wide] of a tile)
    int32_t *entries_a = nullptr, *slow_list = nullptr, *slow_cnt = nullptr, *tile_zeros = nullptr;
    int32_t entries_a_cap = 0;
    int32_t *tile_count_pad = nullptr;      // region A's counters, cnt_pad words apart (a 128-byte line each up to 8192 tiles): see wave_run_issue
    int32_t cnt_pad = 1;
    TileRef *tref = nullptr;
    FarList far;                     // deposits of the gathering regrid that need the generic route (bfgx_regrid2.hpp)
    int32_t *far_overflow_full = nullptr;   // full-map regrid: overflow of the list is repaired in-stream (pass 1), not an error
    int32_t *regrid_todo = nullptr;  // [0] = count, then the tiles the lean gather kernel leaves to the one with the ring walk
    int32_t *regrid_lean = nullptr;  // [0] = count (zeroed by the regrid's last launch), then the tiles of the lean kernel (reach <= 1 ring)
    float *tile_omax = nullptr;      // largest |offset|^2 of every tile (K1's flush or tile_reach_kernel): the reach of the gathering regrid
    bool omax_from_k1 = false;       // set while a fused offsets + regrid call is in flight
    bool paint_pair_f32 = false;     // set while a paint call with acc_f64 = 2 is in flight: f32 pair math into the f64 map
    // precision of the displacement path when the caller leaves it to the plan (BFGX_ACC_AUTO): chosen at plan creation from how far the
    // TABLE can move a pixel (plan_pick_precision): fp32 pair math holds SURVEY 8(d)'s 1e-6 mean(map) while one halo moves a pixel by less
    // than kAutoDispPixels; beyond that the parity-grade mode (BFGX_ACC_PARITY), or fp64 throughout where the fast kernel cannot take the table
    int auto_acc = 0;
    double table_disp_pixels = 0.0;  // largest |d| a / D_A of the table within its model-side cut, in pixel sides of this plan's NSIDE
    float *offsets_lo = nullptr;     // set while a BFGX_ACC_PARITY call is in flight: the low halves of the split pix_offsets
    int k1_tile_lo = 0, k1_tile_n = -1;   // tiles K1 / K3 process (-1: the whole sphere); set by the *_bands_device entries
    int k1_spread_tiles = -1;             // tiles the binned catalog is spread over when that is NOT the tile range of the launch (the one-shot paint entry bins
                                          // the whole catalog once and launches K3 range by range: the density that picks the kernel's form is per sphere); -1: the launch's tiles
    int32_t *tile_apron = nullptr;   // [ntiles][2] rings / columns of apron (tile_apron_kernel)
    int band_reach = 1;              // banded regrid: rings of apron every rank uses (bfgx_plan_set_band_reach)
    int route_margin = 0;            // rings by which bfgx_disc_rings_device widens every halo's range (bfgx_plan_set_route_margin)
    int64_t cat_blk_rows = 0, cat_blk_stride = 0;      // K0 reads a blocked catalog (bfgx_plan_set_catalog_blocks)
    int32_t *wide_tiles = nullptr;   // [1 + ntiles]: number of tiles with wide entries, then those tiles (tile_scan_kernel)
    // fast tiled scatter (bfgx_scatter2.hpp): slim per-halo records + interleaved copies of the table
    bool fast_ok = false;            // 3-axis table with a uniform ln r axis, small enough to interleave
    RowRec *rowrec = nullptr;
    void *pairrec = nullptr;         // PairRecT<float> or PairRecT<double>, whichever the last K0 wrote
    FbRec *fbrec = nullptr;
    const float *tab8f = nullptr;
    const double *tab8d = nullptr;
    unsigned long long *pair_total = nullptr;
    double *tile_sums = nullptr;     // [ntiles][2] per-tile {sum of source values, sum of deposits} of the tiled regrid
    std::vector<int32_t> band_tile0_host;   // first tile of every band (+ total)
    int64_t capacity = 0;
    // optional per-kernel HIP-event timing (bfgx_plan_timing_*)
    bool timing = false;
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev[BFGX_NUM_KERNELS];
    std::vector<std::pair<hipEvent_t, hipEvent_t>> ev_free;
};

namespace {

// brackets one kernel launch with a pair of events on the plan's stream when timing is enabled
template <class Plan>
struct KernelTimerT {
    Plan *p; int kind; std::pair<hipEvent_t, hipEvent_t> e{nullptr, nullptr};
    KernelTimerT(Plan *p_, int kind_) : p(p_), kind(kind_)
    {
        if (!p->timing) return;
        if (!p->ev_free.empty()) { e = p->ev_free.back(); p->ev_free.pop_back(); }
        else { (void)hipEventCreate(&e.first); (void)hipEventCreate(&e.second); }
        (void)hipEventRecord(e.first, p->stream);
    }
    ~KernelTimerT()
    {
        if (!p->timing) return;
        (void)hipEventRecord(e.second, p->stream);
        p->ev[kind].push_back(e);
    }
};
using KernelTimer = KernelTimerT<bfgx_plan>;

static int owned_upload(std::vector<void *> &owned, hipStream_t stream, const void *host, size_t bytes, const void **dev_out)
{
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, bytes));
    owned.push_back(d);
    HIP_TRY(hipMemcpyAsync(d, host, bytes, hipMemcpyHostToDevice, stream));
    *dev_out = d;
    return BFGX_OK;
}

static int plan_upload(bfgx_plan *p, const void *host, size_t bytes, const void **dev_out)
{
    return owned_upload(p->owned, p->stream, host, bytes, dev_out);
}

static int validate_model(const bfgx_model *model)
{
    if (int rc = validate_cosmo(model->cosmo_runner)) return rc;
    if (int rc = validate_cosmo(model->cosmo_model)) return rc;
    if (int rc = validate_table(model->table)) return rc;
    if (!(model->eps_runner > 0)) return fail(BFGX_ERR_INVALID, "epsilon_max must be > 0");
    if (!(model->massdef_runner.Delta > 0) || !(model->massdef_model.Delta > 0))
        return fail(BFGX_ERR_INVALID, "mass definition needs numeric Delta > 0");
    return BFGX_OK;
}

// device copy of the model: table axes + values (r innermost), backgrounds, mass definitions; `with_da` adds the
// D_A(z) spline the lightcone runners need.  Synchronises the stream before returning (staging buffers are local).
static int upload_model(std::vector<void *> &owned, hipStream_t stream, const bfgx_model *model, bool with_da, DevModel &m, int &NC)
{
    const bfgx_table &t = model->table;
    std::memset(&m, 0, sizeof(m));
    m.tab.ndim = t.ndim;
    size_t nvals = 1;
    for (int d = 0; d < t.ndim; ++d) {
        m.tab.n[d] = t.n[d];
        nvals *= (size_t)t.n[d];
        const void *dv = nullptr;
        if (int rc = owned_upload(owned, stream, t.axis[d], sizeof(double) * t.n[d], &dv)) return rc;
        m.tab.axis[d] = (const double *)dv;
    }
    {
        // device layout [z][M][p0][p1][r]: the radial row of every (z, M, params) corner is contiguous
        std::vector<double> tr;
        const double *src = t.values;
        if (t.ndim > 3) {
            // host order [z][M][r][p...] (np C order, the extra axes last) -> device order [z][M][p...][r]
            const size_t nz = t.n[0], nm = t.n[1], nr = t.n[2];
            size_t nx = 1;
            for (int d = 3; d < t.ndim; ++d) nx *= (size_t)t.n[d];
            tr.resize(nvals);
            for (size_t zm = 0; zm < nz * nm; ++zm)
                for (size_t ir = 0; ir < nr; ++ir)
                    for (size_t q = 0; q < nx; ++q)
                        tr[(zm * nx + q) * nr + ir] = t.values[(zm * nr + ir) * nx + q];
            src = tr.data();
        }
        const void *dv = nullptr;
        if (int rc = owned_upload(owned, stream, src, sizeof(double) * nvals, &dv)) return rc;
        m.tab.values = (const double *)dv;
        std::vector<float> v32(nvals);                 // fp32 copy for the mixed-precision pair path (pair_value_fast32)
        for (size_t q = 0; q < nvals; ++q) v32[q] = (float)src[q];
        if (int rc = owned_upload(owned, stream, v32.data(), sizeof(float) * nvals, &dv)) return rc;
        m.tab.values32 = (const float *)dv;
        HIP_TRY(hipStreamSynchronize(stream));
        NC = 4 << (t.ndim - 3);
    }
    m.tab.rdelta = t.rdelta_sampling;
    m.tab.logv = t.log_values;
    m.tab.eps_model = t.eps_model;
    {   // uniform ln r axis -> O(1) index guess
        const double *g = t.axis[2];
        const int n = t.n[2];
        const double step = (g[n - 1] - g[0]) / (n - 1);
        bool uni = true;
        for (int i = 0; i < n; ++i)
            if (std::fabs(g[i] - (g[0] + i * step)) > 1e-9 * std::fabs(step)) { uni = false; break; }
        m.tab.r_uniform = uni ? 1 : 0;
        m.tab.r0 = g[0];
        m.tab.r1 = g[n - 1];
        m.tab.inv_dr = 1.0 / step;
    }
    m.bg_runner = make_background(model->cosmo_runner);
    m.bg_model = make_background(model->cosmo_model);
    m.md_runner = model->massdef_runner;
    m.md_model = model->massdef_model;
    m.eps_runner = model->eps_runner;
    m.same_model = (std::memcmp(&m.bg_runner, &m.bg_model, sizeof(Background)) == 0 && m.md_runner.Delta == m.md_model.Delta &&
                    m.md_runner.rho_type == m.md_model.rho_type) ? 1 : 0;
    if (with_da) {
        std::vector<double> knots, coef;
        da_spline(m.bg_runner, knots, coef);
        const void *dv = nullptr;
        if (int rc = owned_upload(owned, stream, coef.data(), sizeof(double) * coef.size(), &dv)) return rc;
        m.da_coef = (const double *)dv;
        m.da_step = kDaZmax / (kDaKnots - 1);
        HIP_TRY(hipStreamSynchronize(stream));
    }
    return BFGX_OK;
}

static int check_catalog(const bfgx_plan *p, const bfgx_catalog *c)
{
    if (!p || !c) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (c->n < 0 || c->n > p->max_halos) return fail(BFGX_ERR_INVALID, "catalog size %lld exceeds plan max_halos %lld",
                                                     (long long)c->n, (long long)p->max_halos);
    if (c->n > 0 && (!c->M || !c->z || !c->ra || !c->dec)) return fail(BFGX_ERR_INVALID, "catalog column pointer is NULL");
    for (int k = 0; k < p->model.tab.ndim - 3; ++k)
        if (c->n > 0 && !c->extra[k]) return fail(BFGX_ERR_INVALID, "catalog is missing the column of table parameter axis %d", k);
    return BFGX_OK;
}

static bool use_fast(const bfgx_plan *p) { return p->fast_ok && p->algo == 1; }

// the precision a displacement call runs in: BFGX_ACC_AUTO -> what the plan chose from its table; BFGX_ACC_PARITY needs the fast tile kernel
// (3-axis table, uniform ln r) with the wide discs in it and falls back to fp64 throughout (a superset of its accuracy) elsewhere
static int resolve_acc(const bfgx_plan *p, int acc)
{
    if (acc == BFGX_ACC_AUTO) acc = p->auto_acc;
    if (acc == BFGX_ACC_PARITY && !(use_fast(p) && p->k1_wide)) acc = BFGX_ACC_F64;
    return acc;
}
static bool acc_valid(int acc) { return acc == BFGX_ACC_AUTO || acc == BFGX_ACC_F32 || acc == BFGX_ACC_F64 || acc == BFGX_ACC_PARITY; }

// K0.  bin: also reserve the halo's slots in the tile entry lists; f64: precision of the fast kernel's pair records;
// rec_all: write the full HaloRec of every halo (halo-centric kernels), otherwise only of the wide ones
static int launch_prep(bfgx_plan *p, const bfgx_catalog *c, int fallback4, bool bin, bool f64, bool rec_all)
{
    if (c->n == 0) return BFGX_OK;
    const unsigned grid = (unsigned)((c->n + 255) / 256);
    KernelTimer kt(p, BFGX_K_PREP);
    PrepOut o;
    std::memset(&o, 0, sizeof(o));
    o.rec = p->recs; o.rowsx = p->rowsx;
    o.fast = (bin && use_fast(p)) ? (p->k1_wide ? 2 : 1) : 0;
    o.rec_all = (rec_all || !o.fast) ? 1 : 0;
    o.rowrec = p->rowrec; o.pairrec = p->pairrec; o.fbrec = p->fbrec;
    if (bin) {
        o.tref = p->tref; o.cnt_a = p->tile_count; o.cnt_b = p->tile_count_b; o.cnt_w = p->tile_count_w;
        if (o.fast && p->entries_a) o.cnt_a = p->tile_count_pad;          // (direct placement: the padded counters, cnt_pad words apart)
        o.cnt_pad = p->cnt_pad;
        o.work_est = o.fast ? p->k0_work_est : nullptr;
        o.entries_a = o.fast ? p->entries_a : nullptr; o.cap_a = p->entries_a_cap; o.slow_list = p->slow_list; o.slow_cnt = p->slow_cnt;
    }
    o.ncell_m = p->model.tab.n[1] - 1; o.nrm1 = p->model.tab.n[2] - 1;
    o.blk_rows = p->cat_blk_rows; o.blk_stride = p->cat_blk_stride;
    ExtraCols ex;
    for (int k = 0; k < BFGX_MAX_EXTRA; ++k) ex.p[k] = c->extra[k];
#define BFGX_PREP(NCV, REAL)                                                                                        \
    hipLaunchKernelGGL((halo_prep_kernel<NCV, REAL>), dim3(grid), dim3(256), 0, p->stream, p->model, p->hpx, c->n, c->M, c->z, \
                       c->ra, c->dec, ex, c->ln1pz, c->lnM, fallback4, p->tiling, o)
    if (p->NC == 4) { if (f64) BFGX_PREP(4, double); else BFGX_PREP(4, float); }
    else if (p->NC == 8) BFGX_PREP(8, float);
    else if (p->NC == 16) BFGX_PREP(16, float);
    else if (p->NC == 32) BFGX_PREP(32, float);
    else BFGX_PREP(64, float);
#undef BFGX_PREP
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

static int launch_place(bfgx_plan *p, const bfgx_catalog *c)
{
    const unsigned grid = (unsigned)((c->n + 255) / 256);
    hipLaunchKernelGGL(tile_place_kernel, dim3(grid), dim3(256), 0, p->stream, p->hpx, p->tiling, c->n,
                       (const TileRef *)p->tref, (const int32_t *)p->tile_start, (const int32_t *)p->tile_zeros,
                       (const int32_t *)p->tile_count_b, p->tile_cursor, p->tile_cursor_w, p->entries, p->capacity, p->overflow,
                       (const int32_t *)p->slow_list, (const int32_t *)p->slow_cnt);
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

// halo -> tile entry lists: zero counters, K0 (counts), scan, fill
static int launch_prep_and_bin(bfgx_plan *p, const bfgx_catalog *c, int fallback4, bool f64)
{
    p->k1_nhalos = c->n;
    const size_t nt = (size_t)p->tiling.ntiles;
    // cnt_a, cnt_b, cnt_w, cur_b, cur_w, tile counter of the fast kernel, largest |offset|^2 per tile, and the
    // four control words of the regrid's far list behind them (entries, overflow, tiles left to the walking kernel)
    // (in front of them, in the same allocation: region A's counters padded to a line each)
    HIP_TRY(hipMemsetAsync(p->tile_count_pad, 0, sizeof(int32_t) * (nt * (size_t)p->cnt_pad + (((size_t)7 * (nt + 1) + 3) & ~(size_t)3) + 4), p->stream));
    if (int rc = launch_prep(p, c, fallback4, true, f64, false)) return rc;
    KernelTimer kt(p, BFGX_K_BIN);
    const int nt_i = p->tiling.ntiles, nsb = (nt_i + kScanTilesPerWg - 1) / kScanTilesPerWg;
    if (nsb <= 1 || nsb > 1024) {
        hipLaunchKernelGGL(tile_scan_kernel, dim3(1), dim3(1024), 0, p->stream, nt_i, (const int32_t *)p->tile_zeros,
                           (const int32_t *)p->tile_count_b, (const int32_t *)p->tile_count_w, p->tile_start, p->wide_tiles);
    } else {        // NSIDE >= 2048: the scan spread over the tiles' blocks (three short launches instead of up to 48 serial rounds)
        int32_t *tot = p->wide_tiles + nt_i + 1, *off = tot + 2048;
        hipLaunchKernelGGL(tile_scan_part_kernel<0>, dim3(nsb), dim3(1024), 0, p->stream, nt_i, (const int32_t *)p->tile_zeros, (const int32_t *)p->tile_count_b,
                           (const int32_t *)p->tile_count_w, p->tile_start, p->wide_tiles, tot, (const int32_t *)off);
        hipLaunchKernelGGL(tile_scan_blocks_kernel, dim3(1), dim3(1024), 0, p->stream, nsb, nt_i, (const int32_t *)tot, off, p->tile_start, p->wide_tiles);
        hipLaunchKernelGGL(tile_scan_part_kernel<1>, dim3(nsb), dim3(1024), 0, p->stream, nt_i, (const int32_t *)p->tile_zeros, (const int32_t *)p->tile_count_b,
                           (const int32_t *)p->tile_count_w, p->tile_start, p->wide_tiles, tot, (const int32_t *)off);
    }
    HIP_TRY(hipGetLastError());
    if (c->n > 0) if (int rc = launch_place(p, c)) return rc;
    return BFGX_OK;
}

template <int MODE, typename ACC>
static int launch_scatter(bfgx_plan *p, int64_t n, ACC *out, int64_t *counts)
{
    if (n == 0) return BFGX_OK;
    const unsigned grid = (unsigned)((n + kWavesPerBlock - 1) / kWavesPerBlock);
    KernelTimer kt(p, MODE == MODE_OFFSETS ? BFGX_K_OFFSETS : (MODE == MODE_PAINT ? BFGX_K_PAINT : BFGX_K_COUNT));
    hipLaunchKernelGGL((halo_scatter_kernel<MODE, ACC>), dim3(grid), dim3(kWave * kWavesPerBlock), 0, p->stream,
                       make_pair_table(p->model.tab), p->hpx, n, (const HaloRec *)p->recs, out, counts);
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

template <int MODE, typename ACC, int NC>
static int launch_tile_scatter_nc(bfgx_plan *p, ACC *out, bool wide_only)
{
    constexpr int NCOMP = (MODE == MODE_OFFSETS) ? 3 : 1;
    const size_t lds = tile_lds_bytes<NC>(p->tiling.BR, p->tiling.W, NCOMP);
    auto kern = tile_scatter_kernel<MODE, ACC, NC>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    KernelTimer kt(p, wide_only ? BFGX_K_WIDE : (MODE == MODE_OFFSETS ? BFGX_K_OFFSETS : (MODE == MODE_PAINT ? BFGX_K_PAINT : BFGX_K_COUNT)));
    // wide pass: a small fixed grid walks the list of tiles that have wide entries (usually empty or a few polar tiles)
    const int grid = wide_only ? std::min(p->tiling.ntiles, 512) : (p->k1_tile_n < 0 ? p->tiling.ntiles : std::max(p->k1_tile_n, 1));
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWave * kWavesPerBlock), lds, p->stream,
                       make_pair_table(p->model.tab), p->hpx, p->tiling, (const HaloRec *)p->recs, (const RowSetX *)p->rowsx,
                       (const int32_t *)p->tile_start, (const int32_t *)p->entries, p->capacity, out, p->pair_total,
                       wide_only ? (const int32_t *)p->tile_zeros : (const int32_t *)nullptr,
                       wide_only ? (const int32_t *)p->tile_count_b : (const int32_t *)nullptr, wide_only ? 1 : 0,
                       wide_only ? (const int32_t *)p->wide_tiles : (const int32_t *)nullptr,
                       (MODE == MODE_OFFSETS && p->omax_from_k1) ? (unsigned int *)p->tile_omax : (unsigned int *)nullptr,
                       p->k1_tile_lo, p->k1_tile_n);
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

// fast kernel over the narrow-halo region of every tile (stores the tile)
template <int MODE, typename ACC, typename real, int PM = 0>
static int launch_tile_scatter2(bfgx_plan *p, ACC *out, ACC *out_lo = nullptr)
{
    constexpr int NCOMP = (MODE == MODE_OFFSETS) ? 3 : 1;
    const size_t lds = tile2_lds_bytes<real>(p->tiling.BR, p->tiling.W, NCOMP);
    auto kern = tile_scatter2_kernel<MODE, ACC, real, BFGX_NP, PM>;
    HIP_TRY(hipFuncSetAttribute((const void *)kern, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    Tab8T<real> tb;
    tb.v = (sizeof(real) == 4) ? (const real *)p->tab8f : (const real *)p->tab8d;
    tb.r0 = (real)p->model.tab.r0; tb.r1 = (real)p->model.tab.r1; tb.inv_dr = (real)p->model.tab.inv_dr;
    tb.nr = p->model.tab.n[2]; tb._pad = 0;
    // persistent grid: two 512-thread workgroups per CU (LDS and registers allow exactly that) draw tiles from a counter that the
    // binning step has reset (the sixth block of the counter array)
    unsigned int *tile_counter = (unsigned int *)(p->tile_count + 5 * ((size_t)p->tiling.ntiles + 1));
    const int ntodo = p->k1_tile_n < 0 ? p->tiling.ntiles : std::max(p->k1_tile_n, 1);
    KernelTimer kt(p, MODE == MODE_OFFSETS ? BFGX_K_OFFSETS : (MODE == MODE_PAINT ? BFGX_K_PAINT : BFGX_K_COUNT));
    int32_t *form = nullptr;               // non-NULL: both forms are launched and this device word says which one works
    if constexpr (MODE != MODE_COUNT) {
        // the fluid form (one 1024-thread workgroup per CU, two tile slots, no barrier between tiles): BFGX_K1_FLUID=0 at plan creation keeps the other
        const size_t ldsf = tile2f_lds_bytes<real>(p->tiling.BR, p->tiling.W, NCOMP);
        // (fewer than two tiles per CU -- NSIDE <= 256 -- keep the other form: half as many workgroups cannot balance so few tiles; a rank that owns an eighth of the NSIDE-1024 sphere, 776 tiles, is 4 % faster in the fluid form)
        // ... and a sparse catalog (a patch of the sky, a few thousand halos: most tiles list nothing): sixteen waves walking through an
        // empty slot and one wave's chain of dependent loads per drawn tile cost more than the barrier form's zero-and-store (1e4 halos at
        // NSIDE 4096: 2.34 against 1.37 ms; 1e5 at NSIDE 2048: 0.68 against 0.58).  Work per tile follows the halo DENSITY (a halo's
        // pixels grow with NSIDE^2 as the tiles do): 6e5 halos per sphere of the benchmark's catalog are the crossover (5e5 at NSIDE 1024: 0.278
        // against 0.269 ms; 7e5: 0.319 against 0.342) -- 4800 pairs per tile.  A count of halos does not say how large they are, though (1e5
        // halos at z < 0.05 hold 2300 pixels each: fluid 0.82 / 0.68 ms against 1.03 / 1.09 for displacement / painting): from 6e5 halos
        // per sphere the fluid form is taken outright; below, K0's per-workgroup sums of the halos' estimated pixels decide ON THE DEVICE
        // (k1_form_kernel) and both forms are launched -- the one not chosen returns at once (~5 us).
        const bool eligible = p->k1_fluid && ldsf <= (size_t)160 * 1024 && p->tiling.BR <= kWave && ntodo >= 2 * p->num_cus;
        const int nspread = p->k1_spread_tiles > 0 ? p->k1_spread_tiles : ntodo;      // (the tiles the binned halos cover)
        const bool dense = (double)p->k1_nhalos * (double)p->tiling.ntiles >= 6.0e5 * (double)nspread;
        const bool forced = p->k1_fluid == 2 && ldsf <= (size_t)160 * 1024 && p->tiling.BR <= kWave;
        if (eligible && !dense && !forced && p->k0_work_est && p->k1_nhalos > 0) {
            const int nblk = (int)((p->k1_nhalos + 255) / 256);
            hipLaunchKernelGGL(k1_form_kernel, dim3(1), dim3(1024), 0, p->stream, nblk, (const float *)p->k0_work_est, 4800.0f * (float)nspread, p->k1_form);
            form = p->k1_form;
        }
        if (forced || (eligible && (dense || form))) {
            auto kf = tile_scatter2f_kernel<MODE, ACC, real, PM>;
            HIP_TRY(hipFuncSetAttribute((const void *)kf, hipFuncAttributeMaxDynamicSharedMemorySize, (int)ldsf));
            const int gridf = std::min(ntodo, p->num_cus);
            hipLaunchKernelGGL(kf, dim3(std::max(gridf, 1)), dim3(kWave * FluidWaves<real>::n), ldsf, p->stream, tb, p->hpx, p->tiling,
                               (const RowRec *)p->rowrec, (const PairRecT<real> *)p->pairrec, (const FbRec *)p->fbrec,
                               (const int32_t *)p->tile_start, (const int32_t *)(p->entries_a ? p->tile_count_pad : p->tile_count), (const int32_t *)p->tile_count_b,
                               (const int32_t *)p->entries, p->capacity, (const int32_t *)p->entries_a, (int)p->entries_a_cap, (int)p->cnt_pad, out, out_lo, tile_counter,
                               (MODE == MODE_OFFSETS && p->omax_from_k1) ? (unsigned int *)p->tile_omax : (unsigned int *)nullptr,
                               p->k1_tile_lo, p->k1_tile_n, p->overflow, (const int32_t *)form, 1);
            HIP_TRY(hipGetLastError());
            if (!form) return BFGX_OK;
        }
    }
    const int grid = std::min(ntodo, 2 * p->num_cus);
    hipLaunchKernelGGL(kern, dim3(grid), dim3(kWave * kW2), lds, p->stream, tb, p->hpx, p->tiling,
                       (const RowRec *)p->rowrec, (const PairRecT<real> *)p->pairrec, (const FbRec *)p->fbrec,
                       (const int32_t *)p->tile_start, (const int32_t *)(p->entries_a ? p->tile_count_pad : p->tile_count), (const int32_t *)p->tile_count_b,
                       (const int32_t *)p->entries, p->capacity, (const int32_t *)p->entries_a, (int)p->entries_a_cap, (int)p->cnt_pad, out, out_lo, p->pair_total, tile_counter,
                       (MODE == MODE_OFFSETS && p->omax_from_k1) ? (unsigned int *)p->tile_omax : (unsigned int *)nullptr,
                       p->k1_tile_lo, p->k1_tile_n, (const int32_t *)form, 0);
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

// the tiled scatter: fast kernel for the narrow halos (if the table allows) + generic kernel for the rest
template <int MODE, typename ACC>
static int launch_tile_scatter(bfgx_plan *p, ACC *out)
{
    if (use_fast(p)) {
        using real = typename std::conditional<sizeof(ACC) == 8, double, float>::type;
        if constexpr (MODE == MODE_OFFSETS && sizeof(ACC) == 4) {
            // BFGX_ACC_PARITY: fp64 pair math (K0 wrote fp64 pair records) with the parity-grade functions, pix_offsets as two fp32 arrays
            if (p->offsets_lo) return launch_tile_scatter2<MODE, ACC, double, 1>(p, out, (ACC *)p->offsets_lo);
        }
        if constexpr (MODE == MODE_PAINT && sizeof(ACC) == 8) {
            // acc_f64 = 2: the pair phase in fp32 (K0 wrote fp32 pair records), LDS accumulation and the stored map in fp64
            if (p->paint_pair_f32) {
                if (int rc = launch_tile_scatter2<MODE, ACC, float>(p, out)) return rc;
                return p->k1_wide ? BFGX_OK : launch_tile_scatter_nc<MODE, ACC, 4>(p, out, true);
            }
        }
        if (int rc = launch_tile_scatter2<MODE, ACC, real>(p, out)) return rc;
        return p->k1_wide ? BFGX_OK : launch_tile_scatter_nc<MODE, ACC, 4>(p, out, true);      // (k1_wide: K0 classed no halo as wide)
    }
    if (p->NC == 4) return launch_tile_scatter_nc<MODE, ACC, 4>(p, out, false);
    if (MODE == MODE_COUNT) return launch_tile_scatter_nc<MODE, ACC, 4>(p, out, false);      // census never reads the rows
    if (p->NC == 8) return launch_tile_scatter_nc<MODE, ACC, 8>(p, out, false);
    if (p->NC == 16) return launch_tile_scatter_nc<MODE, ACC, 16>(p, out, false);
    if (p->NC == 32) return launch_tile_scatter_nc<MODE, ACC, 32>(p, out, false);       // three / four property axes (Tabulate.py:524-561)
    return launch_tile_scatter_nc<MODE, ACC, 64>(p, out, false);
}

// blocking: if the entry list overflowed, grow it to the exact size and redo the fill pass
static int ensure_entry_capacity(bfgx_plan *p, const bfgx_catalog *c)
{
    int32_t ov = 0, total = 0;
    HIP_TRY(hipMemcpyAsync(&ov, p->overflow, sizeof(ov), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(&total, p->tile_start + p->tiling.ntiles, sizeof(total), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (!(ov & 1)) return BFGX_OK;             // (bit 2: the fluid kernel gave up -- bfgx_plan_status reports it)
    void *d = nullptr;
    HIP_TRY(hipMalloc(&d, sizeof(int32_t) * ((size_t)total + 16)));
    p->owned.push_back(d);                       // the old list is released with the plan
    p->entries = (int32_t *)d;
    p->capacity = (int64_t)total + 16;
    HIP_TRY(hipMemsetAsync(p->overflow, 0, sizeof(int32_t), p->stream));
    HIP_TRY(hipMemsetAsync(p->tile_cursor, 0, sizeof(int32_t) * 2 * ((size_t)p->tiling.ntiles + 1), p->stream));   // cur_b, cur_w
    return launch_place(p, c);
}

// interleaved copy of a 3-axis table for the fast kernel (Tab8T): [(nz-1)(nm-1)][nr-1][8] = {A_c, B_c}, c = 2 bz + bm,
// A_c = T[iz+bz][im+bm][i], B_c = T[iz+bz][im+bm][i+1] - A_c  (differences formed in fp64)
template <typename real>
static void build_tab8(const bfgx_table &t, std::vector<real> &out)
{
    const size_t nz = t.n[0], nm = t.n[1], nr = t.n[2];
    out.resize((nz - 1) * (nm - 1) * (nr - 1) * 8);
    for (size_t iz = 0; iz + 1 < nz; ++iz)
        for (size_t im = 0; im + 1 < nm; ++im)
            for (size_t i = 0; i + 1 < nr; ++i) {
                real *o = out.data() + ((iz * (nm - 1) + im) * (nr - 1) + i) * 8;
                for (int c = 0; c < 4; ++c) {
                    const double *row = t.values + ((iz + (c >> 1)) * nm + im + (c & 1)) * nr;
                    o[2 * c] = (real)row[i];
                    o[2 * c + 1] = (real)(row[i + 1] - row[i]);
                }
            }
}

// host-side construction of the tiling tables for one nside
static void build_tiling(int64_t nside, bool paint, int &BR, int &W, std::vector<int32_t> &tile0, std::vector<int32_t> &nphi,
                         std::vector<int32_t> &nrmin, std::vector<int32_t> &tband)
{
    auto pow2_floor = [](int64_t v) { int p = 1; while (2 * p <= v) p *= 2; return p; };
    W = std::max(16, std::min(64, pow2_floor(std::max<int64_t>(1, nside / 2))));
    // the painted map needs 8 B of LDS per pixel, pix_offsets 24 B: taller tiles for painting (fewer duplicated ring phases)
    BR = std::max(4, std::min(paint ? 64 : 32, pow2_floor(std::max<int64_t>(1, nside / 8))));
    if (const char *e = std::getenv("BFGX_TILE_W")) W = pow2_floor(std::max(4, std::min(64, std::atoi(e))));   // tuning knobs (W <= 64, a power of two)
    if (const char *e = std::getenv("BFGX_TILE_BR")) BR = std::max(1, std::min(64, std::atoi(e)));
    const int64_t nrings = 4 * nside - 1;
    const int nbands = (int)((nrings + BR - 1) / BR);
    auto rlen = [&](int64_t ring) { return 4 * (ring < nside ? ring : (ring > 3 * nside ? 4 * nside - ring : nside)); };
    tile0.assign(nbands + 1, 0); nphi.assign(nbands, 1); nrmin.assign(nbands, 4);
    for (int b = 0; b < nbands; ++b) {
        const int64_t i0 = 1 + (int64_t)b * BR, i1 = std::min<int64_t>(i0 + BR, 4 * nside);
        int64_t lmax = 0, lmin = INT64_MAX;
        for (int64_t i = i0; i < i1; ++i) { lmax = std::max(lmax, rlen(i)); lmin = std::min(lmin, rlen(i)); }
        nphi[b] = (int)((lmax + W - 1) / W);
        nrmin[b] = (int)lmin;
        tile0[b + 1] = tile0[b] + nphi[b];
    }
    tband.resize(tile0[nbands]);
    for (int b = 0; b < nbands; ++b)
        for (int t = tile0[b]; t < tile0[b + 1]; ++t) tband[t] = b;
}


// How far can this table move a pixel?  The largest |d(r)| a / D_A(z) over the table's (z, M) nodes within the model-side cut r < eps_model R
// and the runner's disc, in pixel sides of the plan's NSIDE.  fp32 pair math carries ~4e-7 of every contribution, and the bilinear deposit turns
// an error of e pixel sides in a source pixel's position into ~e of its value: 1e-6 mean(map) -- SURVEY 8(d)'s tolerance -- holds while a
// halo moves a pixel by a small fraction of a pixel (measured at config 2, closed-form table scaled: 2.6e-7 / 3.9e-7 / 1.3e-6 / 8.0e-6 mean(map) at 0.03 /
// 0.09 / 0.31 / 0.93 pixel sides per halo; 2.1e-5 on the S19 table, which moves 8.6).  Beyond kAutoDispPixels the plan picks the
// parity-grade mode.
constexpr double kAutoDispPixels = 0.1;
static void plan_pick_precision(bfgx_plan *p, const bfgx_model *model)
{
    const bfgx_table &t = model->table;
    p->auto_acc = BFGX_ACC_F32;
    p->table_disp_pixels = 0.0;
    if (t.log_values) return;                                // a profile table: painting has its own modes
    const Background bgr = make_background(model->cosmo_runner), bgm = make_background(model->cosmo_model);
    const double pix = std::sqrt(4.0 * 3.14159265358979323846 / (double)p->hpx.npix);
    const size_t nz = t.n[0], nm = t.n[1], nr = t.n[2];
    size_t nx = 1;
    for (int d = 3; d < t.ndim; ++d) nx *= (size_t)t.n[d];
    double worst = 0.0;
    for (size_t iz = 0; iz < nz; ++iz) {
        const double z = std::exp(t.axis[0][iz]) - 1.0, a = 1.0 / (1.0 + z);
        const double D = z > 0.0 ? angular_diameter_distance(bgr, z) : 0.0;
        for (size_t im = 0; im < nm; ++im) {
            const double M = std::exp(t.axis[1][im]);
            if (!(M > 0.0) || !(a > 0.0)) continue;
            const double Rm = radius_delta(bgm, model->massdef_model, M, a) / a, Rr = radius_delta(bgr, model->massdef_runner, M, a) / a;
            const double rcut = std::min(t.eps_model * Rm, model->eps_runner * Rr);          // comoving Mpc
            for (size_t ir = 0; ir < nr; ++ir) {
                const double r = std::exp(t.axis[2][ir]) * (t.rdelta_sampling ? Rm : 1.0);
                double m = 0.0;
                const double *v = t.values + ((iz * nm + im) * nr + ir) * nx;
                for (size_t q = 0; q < nx; ++q) if (std::isfinite(v[q])) m = std::max(m, std::fabs(v[q]));
                const double px = (D > 0.0) ? m * a / D / pix : (m > 0.0 ? 1.0e30 : 0.0);
                worst = std::max(worst, px);
                if (r >= rcut) break;                        // (the node that closes the interval holding the cut is the last that counts)
            }
        }
    }
    p->table_disp_pixels = worst;
    if (worst > kAutoDispPixels) p->auto_acc = BFGX_ACC_PARITY;
    if (const char *e = std::getenv("BFGX_AUTO_ACC")) { const int v = std::atoi(e); if (v == 0 || v == 1 || v == 3) p->auto_acc = v; }      // tests / A-B runs
}

}  // namespace

extern "C" {

int bfgx_abi_version(void) { return BFGX_ABI_VERSION; }

#if BFGX_K1F_PROF
// variant builds only (scripts/k1f_prof.py): the fluid kernel's shader-clock accounting, summed over waves and launches since the last reset
int bfgx_debug_k1f_prof(unsigned long long *out8, int reset)
{
    HIP_TRY(hipDeviceSynchronize());
    if (out8) HIP_TRY(hipMemcpyFromSymbol(out8, HIP_SYMBOL(bfgx::g_k1f_prof), 8 * sizeof(unsigned long long)));
    if (reset) { unsigned long long z[8] = {0}; HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(bfgx::g_k1f_prof), z, sizeof(z))); }
    return BFGX_OK;
}
#endif


const char *bfgx_last_error(void) { return g_err.c_str(); }

int bfgx_device_count(void)
{
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

// ------------------------------------------------------------------------------ host cosmology
int bfgx_cosmo_E2(const bfgx_cosmo *c, int64_t n, const double *a, double *out)
{
    if (!c || !a || !out) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (int rc = validate_cosmo(*c)) return rc;
    const Background b = make_background(*c);
    for (int64_t i = 0; i < n; ++i) out[i] = E2(b, a[i]);
    return BFGX_OK;
}

int bfgx_cosmo_radius(const bfgx_cosmo *c, const bfgx_massdef *md, int64_t n, const double *M, const double *a, double *out)
{
    if (!c || !md || !M || !a || !out) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (int rc = validate_cosmo(*c)) return rc;
    const Background b = make_background(*c);
    for (int64_t i = 0; i < n; ++i) out[i] = radius_delta(b, *md, M[i], a[i]);
    return BFGX_OK;
}

int bfgx_cosmo_angular_diameter_distance(const bfgx_cosmo *c, int64_t n, const double *z, double *out)
{
    if (!c || !z || !out) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (int rc = validate_cosmo(*c)) return rc;
    const Background b = make_background(*c);
    for (int64_t i = 0; i < n; ++i) out[i] = angular_diameter_distance(b, z[i]);
    return BFGX_OK;
}

int bfgx_cosmo_da_spline(const bfgx_cosmo *c, double *knots, double *coef)
{
    if (!c || !knots || !coef) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (int rc = validate_cosmo(*c)) return rc;
    std::vector<double> k, cf;
    da_spline(make_background(*c), k, cf);
    std::memcpy(knots, k.data(), k.size() * sizeof(double));
    std::memcpy(coef, cf.data(), cf.size() * sizeof(double));
    return BFGX_OK;
}

int bfgx_cosmo_da_eval(const bfgx_cosmo *c, int64_t n, const double *z, double *out)
{
    if (!c || !z || !out) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (int rc = validate_cosmo(*c)) return rc;
    std::vector<double> k, cf;
    da_spline(make_background(*c), k, cf);
    for (int64_t i = 0; i < n; ++i) out[i] = da_eval(k, cf, z[i]);
    return BFGX_OK;
}

// ------------------------------------------------------------------------------ plan
int bfgx_plan_create(int device, void *hip_stream, int64_t nside, int64_t max_halos,
                     const bfgx_model *model, bfgx_plan **out)
{
    if (!model || !out) return fail(BFGX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    // (NSIDE 16384 has 3.2e9 pixels: the tile kernels keep first-pixel-of-ring offsets and per-tile counters in 32 bits -- RowScan.start, the
    // walking regrid; 8192, 8.05e8 pixels / 6.4 GB of fp64 map, is the largest shell that is tested)
    if (nside < 1 || nside > 8192) return fail(BFGX_ERR_INVALID, "nside must be 1 .. 8192 (the tile kernels index pixels with 32 bits inside a ring table)");
    if (max_halos < 0) return fail(BFGX_ERR_INVALID, "max_halos < 0");
    if (int rc = validate_model(model)) return rc;
    if (bfgx_device_count() <= 0)
        return fail(BFGX_ERR_NO_DEVICE, "no HIP device visible: libbfgx has no CPU fallback");
    HIP_TRY(hipSetDevice(device));

    bfgx_plan *p = new bfgx_plan();
    p->device = device;
    {
        int cus = 0;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device) == hipSuccess && cus > 0) p->num_cus = cus;
    }
    p->nside = nside;
    p->max_halos = max_halos;
    p->hpx = make_hpx(nside);
    // hip_stream == NULL means the legacy default stream (what torch's default stream is), so that the
    // plan's kernels stay ordered with the caller's own work; the plan never creates a private stream.
    p->stream = (hipStream_t)hip_stream;
    auto bail = [&](int rc) { bfgx_plan_destroy(p); return rc; };

    const bfgx_table &t = model->table;
    if (int rc = upload_model(p->owned, p->stream, model, true, p->model, p->NC)) return bail(rc);
    plan_pick_precision(p, model);
    {
        void *d = nullptr;
        if (hipMalloc(&d, sizeof(HaloRec) * (size_t)(max_halos > 0 ? max_halos : 1)) != hipSuccess)
            return bail(fail(BFGX_ERR_HIP, "hipMalloc(halo records) failed"));
        p->owned.push_back(d);
        p->recs = (HaloRec *)d;
        if (p->NC > 4) {
            if (hipMalloc(&d, sizeof(RowSetX) * (size_t)(max_halos > 0 ? max_halos : 1)) != hipSuccess)
                return bail(fail(BFGX_ERR_HIP, "hipMalloc(corner rows) failed"));
            p->owned.push_back(d);
            p->rowsx = (RowSetX *)d;
        }
    }
    {   // fast tiled scatter: possible for a 3-axis table with a uniform ln r axis whose interleaved copy stays small
        const size_t n8 = (size_t)(t.n[0] - 1) * (size_t)(t.n[1] - 1) * (size_t)(t.n[2] - 1) * 8;
        p->fast_ok = (t.ndim == 3) && p->model.tab.r_uniform && n8 * sizeof(double) <= ((size_t)256 << 20) && !std::getenv("BFGX_NO_FAST");
        if (p->fast_ok) {
            std::vector<float> v32;
            std::vector<double> v64;
            build_tab8<float>(t, v32);
            build_tab8<double>(t, v64);
            const void *dv = nullptr;
            if (int rc = plan_upload(p, v32.data(), sizeof(float) * v32.size(), &dv)) return bail(rc);
            p->tab8f = (const float *)dv;
            if (int rc = plan_upload(p, v64.data(), sizeof(double) * v64.size(), &dv)) return bail(rc);
            p->tab8d = (const double *)dv;
            if (hipStreamSynchronize(p->stream) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "stream sync failed"));
            const size_t nh = (size_t)(max_halos > 0 ? max_halos : 1);
            void *d = nullptr;
            if (hipMalloc(&d, sizeof(RowRec) * nh) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "hipMalloc(row records) failed"));
            p->owned.push_back(d); p->rowrec = (RowRec *)d;
            if (hipMalloc(&d, sizeof(PairRecT<double>) * nh) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "hipMalloc(pair records) failed"));
            p->owned.push_back(d); p->pairrec = d;
            if (hipMalloc(&d, sizeof(FbRec) * nh) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "hipMalloc(fallback records) failed"));
            p->owned.push_back(d); p->fbrec = (FbRec *)d;
        }
    }
    {   // tiling tables and halo -> tile binning workspace
        int BR, W;
        std::vector<int32_t> tile0, nphi, nrmin, tband;
        build_tiling(nside, t.log_values != 0, BR, W, tile0, nphi, nrmin, tband);
        Tiling &T = p->tiling;
        T.BR = BR; T.W = W; T.nbands = (int)nphi.size(); T.ntiles = (int)tband.size();
        const void *dv = nullptr;
        if (int rc = plan_upload(p, tile0.data(), sizeof(int32_t) * tile0.size(), &dv)) return bail(rc);
        T.band_tile0 = (const int32_t *)dv;
        p->band_tile0_host = tile0;
        if (int rc = plan_upload(p, nphi.data(), sizeof(int32_t) * nphi.size(), &dv)) return bail(rc);
        T.band_nphi = (const int32_t *)dv;
        if (int rc = plan_upload(p, nrmin.data(), sizeof(int32_t) * nrmin.size(), &dv)) return bail(rc);
        T.band_nrmin = (const int32_t *)dv;
        if (int rc = plan_upload(p, tband.data(), sizeof(int32_t) * tband.size(), &dv)) return bail(rc);
        T.tile_band = (const int32_t *)dv;
        std::vector<int32_t> order(tband.size());
        {   // launch order: tiles by their shortest ring / azimuth slices (~ pixels per tile), descending; stable
            std::vector<double> weight(tband.size());
            for (size_t q = 0; q < tband.size(); ++q) { order[q] = (int32_t)q; weight[q] = (double)nrmin[tband[q]] / (double)nphi[tband[q]]; }
            std::stable_sort(order.begin(), order.end(), [&](int32_t x, int32_t y) { return weight[x] > weight[y]; });
        }
        if (int rc = plan_upload(p, order.data(), sizeof(int32_t) * order.size(), &dv)) return bail(rc);
        T.tile_order = (const int32_t *)dv;
        if (hipStreamSynchronize(p->stream) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "stream sync failed"));
        p->capacity = 8 * max_halos + 4096;
        if (const char *e = std::getenv("BFGX_K1_FLUID")) p->k1_fluid = std::max(0, std::min(2, std::atoi(e)));
        if (const char *e = std::getenv("BFGX_K1_WIDE")) p->k1_wide = std::atoi(e) != 0;
        if (const char *e = std::getenv("BFGX_ENTRY_CAP")) p->capacity = std::max<int64_t>(16, std::atoll(e));   // tests: force regrowth
        auto dalloc = [&](size_t bytes, void **ptr) {
            if (hipMalloc(ptr, bytes) != hipSuccess) return 1;
            p->owned.push_back(*ptr);
            return 0;
        };
        void *d0 = nullptr, *d1 = nullptr, *d3 = nullptr, *d4 = nullptr, *d5 = nullptr, *d6 = nullptr;
        // the binning counters and, right behind them (16-byte aligned), the control words of the regrid's far list: ONE memset per step
        // zeroes both (a length that is not a multiple of 16 bytes, or two buffers, cost a fill kernel each: 4.5 us)
        const size_t n7p = ((size_t)7 * (T.ntiles + 1) + 3) & ~(size_t)3;
        // region A's counters: a line each while that array stays below 4 MB (atomics to one line serialise; a larger array costs misses)
        p->cnt_pad = kCntPadMax;
        while (p->cnt_pad > 1 && (size_t)T.ntiles * p->cnt_pad * sizeof(int32_t) > ((size_t)4 << 20)) p->cnt_pad >>= 1;
        const size_t npad = (size_t)T.ntiles * p->cnt_pad;
        if (dalloc(sizeof(int32_t) * (npad + n7p + (size_t)T.ntiles + 8), &d0) || dalloc(sizeof(int32_t) * (T.ntiles + 1), &d1) ||
            dalloc(sizeof(int32_t) * (size_t)p->capacity, &d3) || dalloc(sizeof(int32_t), &d4) ||
            dalloc(sizeof(unsigned long long), &d5) || dalloc(sizeof(TileRef) * (size_t)(max_halos > 0 ? max_halos : 1), &d6))
            return bail(fail(BFGX_ERR_HIP, "hipMalloc(binning workspace) failed"));
        p->tile_count_pad = (int32_t *)d0;
        p->tile_count = (int32_t *)d0 + npad; p->tile_count_b = p->tile_count + (T.ntiles + 1);
        p->tile_count_w = p->tile_count + 2 * (T.ntiles + 1);
        p->tile_cursor = p->tile_count + 3 * (T.ntiles + 1); p->tile_cursor_w = p->tile_count + 4 * (T.ntiles + 1);
        p->tile_start = (int32_t *)d1; p->tref = (TileRef *)d6;
        p->entries = (int32_t *)d3; p->overflow = (int32_t *)d4; p->pair_total = (unsigned long long *)d5;
        {
            // capacity of a tile's fixed list: 16 x the mean entries per tile (a rank of an N-GPU run bins its share of the halos into 1 / N
            // of the tiles; a sky patch does the same), a power of two in [64, 32768] (4 x the mean above 512 entries per tile: a catalog of 1e7 small halos lists 4200 per tile, and what does not fit takes the slow route -- K3 3.7 against 1.4 ms), the whole array at most 1 GB
            const double mean = 1.3 * (double)std::max<int64_t>(max_halos, 1) / (double)T.ntiles;
            int cap_a = 64;
            while (cap_a < 32768 && (double)cap_a < (mean > 512.0 ? 4.0 : 16.0) * mean + 64.0) cap_a <<= 1;
            while (cap_a > 64 && (size_t)T.ntiles * cap_a * sizeof(int32_t) > ((size_t)1 << 30)) cap_a >>= 1;
            if (const char *e = std::getenv("BFGX_TILE_LIST_CAP")) cap_a = std::max(1, std::atoi(e));        // tests: force the overflow into region B
            void *da = nullptr, *ds = nullptr, *dc = nullptr;
            const size_t nblk = ((size_t)std::max<int64_t>(max_halos, 1) + 255) / 256;
            void *dw = nullptr, *df = nullptr;
            if (dalloc(sizeof(float) * nblk, &dw) || dalloc(sizeof(int32_t) * 4, &df)) return bail(fail(BFGX_ERR_HIP, "hipMalloc(work estimate) failed"));
            p->k0_work_est = (float *)dw; p->k1_form = (int32_t *)df;
            if (dalloc(sizeof(int32_t) * (size_t)T.ntiles * cap_a, &da) || dalloc(sizeof(int32_t) * nblk * 256, &ds) || dalloc(sizeof(int32_t) * nblk, &dc))
                return bail(fail(BFGX_ERR_HIP, "hipMalloc(tile lists) failed"));
            p->entries_a = (int32_t *)da; p->slow_list = (int32_t *)ds; p->slow_cnt = (int32_t *)dc; p->entries_a_cap = cap_a;
            p->tile_zeros = p->tile_count + 5 * ((size_t)T.ntiles + 1) + 1;      // the words behind K1's tile counter: zeroed with the counters, never written
        }
        {
            void *f0 = nullptr, *f1 = nullptr, *f2 = nullptr, *f3 = nullptr;
            // deposits listed for the generic route: four per pixel that moves beyond the gathering reach (15 rings = 9.9 / NSIDE rad).  At NSIDE
            // 1024 on the S19 table that is 0.2 % of the pixels; at NSIDE 2048 the same field in radians sends 8 % there (16 M entries), and a
            // list that overflows costs a second pass with the generic evaluation under divergence (25 ms): room for half the pixels, 1 GB at most
            // (round 4, second half: at config 4's REAL density -- 1e7 halos on one NSIDE-2048 shell, 43 discs over every pixel -- the S19
            // displacements add up to 15 pixels on average and 58 % of the pixels move beyond the 15-ring reach: 1.4e8 listed deposits,
            // 2.8 per pixel.  A list for half the pixels overflowed into the repair pass: K2 35 ms; with room for four deposits of EVERY
            // pixel, 4.3 GB at most, 10 ms.)
            // Sized from the table (ADVICE, round 4): a listed deposit needs a pixel that moves beyond 9.7 pixel sides in total.  Tables that move a
            // pixel by less than a quarter of a pixel per halo list the pole caps only (a few dozen entries): room for 1 / 16 of the pixels
            // (16 B each: 13 MB at NSIDE 1024 instead of 0.8 GB); up to two pixels per halo: half the pixels; beyond: four deposits of every pixel.
            // A list that overflows all the same (many overlapping discs) is repaired in-stream by the regrid's last pass.
            const int64_t npx = p->hpx.npix;
            const int64_t want = p->table_disp_pixels < 0.25 ? npx / 16 : (p->table_disp_pixels < 2.0 ? npx / 2 : 4 * npx);
            p->far.cap = std::min<int64_t>(std::max<int64_t>((int64_t)1 << 20, want), (int64_t)1 << 28);
            if (const char *e = std::getenv("BFGX_FAR_CAP")) p->far.cap = std::max<int64_t>(1024, std::atoll(e));          // tests: force the overflow
            // control words in one allocation: [0..7] entries listed, [8..11] overflow (full-map regrid), [12..15] tiles left to
            // the walking kernel (followed by their numbers), ... ; the banded regrid's overflow flag lives after the tile list
            void *f4 = nullptr;
            f0 = p->tile_count + n7p;
            if (dalloc(sizeof(int64_t) * (size_t)p->far.cap, &f1) ||
                dalloc(sizeof(double) * (size_t)p->far.cap, &f2) ||
                dalloc(sizeof(int32_t) * 2 * (size_t)(T.ntiles + 1), &f3))
                return bail(fail(BFGX_ERR_HIP, "hipMalloc(far list) failed"));
            p->tile_apron = (int32_t *)f3;
            {
                void *fl = nullptr;
                if (dalloc(sizeof(int32_t) * (size_t)(T.ntiles + 4), &fl)) return bail(fail(BFGX_ERR_HIP, "hipMalloc(lean tile list) failed"));
                p->regrid_lean = (int32_t *)fl;
                if (hipMemsetAsync(fl, 0, sizeof(int32_t) * 4, p->stream) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "hipMemset failed"));
            }
            int32_t *ctrl = (int32_t *)f0;
            p->far.count = (unsigned long long *)ctrl; p->far.pix = (int64_t *)f1; p->far.val = (double *)f2;
            p->far_overflow_full = ctrl + 2;
            p->regrid_todo = ctrl + 3;                       // [0] count, [1 ..] tiles
            p->far.overflow = ctrl + 4 + T.ntiles;
            p->tile_omax = (float *)(p->tile_count + 6 * ((size_t)T.ntiles + 1));
            (void)f4;
            if (hipMemsetAsync(ctrl, 0, sizeof(int32_t) * (size_t)(T.ntiles + 8), p->stream) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "hipMemset failed"));
        }
        void *d9 = nullptr;
        // (+ 4096 ints behind the list: per-block totals and offsets of the multi-workgroup tile scan)
        if (dalloc(sizeof(int32_t) * (size_t)(T.ntiles + 1 + 4096), &d9)) return bail(fail(BFGX_ERR_HIP, "hipMalloc(wide tile list) failed"));
        p->wide_tiles = (int32_t *)d9;
        void *d8 = nullptr;
        if (dalloc(sizeof(double) * 2 * (size_t)(T.ntiles + 1), &d8)) return bail(fail(BFGX_ERR_HIP, "hipMalloc(tile sums) failed"));
        p->tile_sums = (double *)d8;
        if (hipMemsetAsync(p->overflow, 0, sizeof(int32_t), p->stream) != hipSuccess)
            return bail(fail(BFGX_ERR_HIP, "hipMemset failed"));
    }
    if (hipStreamSynchronize(p->stream) != hipSuccess) return bail(fail(BFGX_ERR_HIP, "stream sync failed"));
    *out = p;
    return BFGX_OK;
}

int bfgx_plan_timing_enable(bfgx_plan *p, int on)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    p->timing = on != 0;
    return BFGX_OK;
}

int bfgx_plan_timing_read(bfgx_plan *p, double *ms_sum, int64_t *launches)
{
    if (!p || !ms_sum || !launches) return fail(BFGX_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipStreamSynchronize(p->stream));
    for (int k = 0; k < BFGX_NUM_KERNELS; ++k) {
        double tot = 0.0;
        for (auto &e : p->ev[k]) {
            float ms = 0.f;
            HIP_TRY(hipEventElapsedTime(&ms, e.first, e.second));
            tot += ms;
            p->ev_free.push_back(e);
        }
        ms_sum[k] = tot;
        launches[k] = (int64_t)p->ev[k].size();
        p->ev[k].clear();
    }
    return BFGX_OK;
}

void bfgx_plan_destroy(bfgx_plan *p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    (void)hipStreamSynchronize(p->stream);
    for (int k = 0; k < BFGX_NUM_KERNELS; ++k)
        for (auto &e : p->ev[k]) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (auto &e : p->ev_free) { (void)hipEventDestroy(e.first); (void)hipEventDestroy(e.second); }
    for (void *d : p->owned) (void)hipFree(d);
    delete p;
}

int bfgx_plan_set_algo(bfgx_plan *p, int algo)
{
    if (!p || (algo != 0 && algo != 1)) return fail(BFGX_ERR_INVALID, "algo must be 0 (per-halo global atomics) or 1 (LDS tiles)");
    p->algo = algo;
    return BFGX_OK;
}

int bfgx_plan_status(bfgx_plan *p)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    HIP_TRY(hipSetDevice(p->device));
    int32_t ov = 0;
    HIP_TRY(hipMemcpyAsync(&ov, p->overflow, sizeof(ov), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (ov & 4) {
        // the fluid form of the fast kernel: a wave waited 2^22 sleeps for a tile slot / a flush / a staged tile of its own workgroup and
        // left (the grid drained); never seen outside fault injection
        HIP_TRY(hipMemsetAsync(p->overflow, 0, sizeof(int32_t), p->stream));
        return fail(BFGX_ERR_HIP, "tile_scatter2f_kernel: a wave gave up waiting inside its workgroup; results are incomplete (BFGX_K1_FLUID=0 selects the other form)");
    }
    if (ov) return fail(BFGX_ERR_INVALID, "halo->tile entry list overflowed its capacity (%lld); results are incomplete",
                        (long long)p->capacity);
    int32_t fov = 0;
    HIP_TRY(hipMemcpyAsync(&fov, p->far.overflow, sizeof(fov), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if (fov) {
        HIP_TRY(hipMemsetAsync(p->far.overflow, 0, sizeof(int32_t), p->stream));
        return fail(BFGX_ERR_INVALID, "the regrid's list of far deposits overflowed (%lld entries): displacements of many pixels over a large "
                                      "part of the map", (long long)p->far.cap);
    }
    return BFGX_OK;
}

int bfgx_plan_precision(bfgx_plan *p, int acc_requested, int32_t *acc_resolved, double *table_disp_pixels)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    if (!acc_valid(acc_requested)) return fail(BFGX_ERR_INVALID, "acc must be BFGX_ACC_AUTO (-1), 0 (f32), 1 (f64) or BFGX_ACC_PARITY (3)");
    if (acc_resolved) *acc_resolved = resolve_acc(p, acc_requested);
    if (table_disp_pixels) *table_disp_pixels = p->table_disp_pixels;
    return BFGX_OK;
}

int bfgx_plan_bands(bfgx_plan *p, int32_t *nbands, int64_t *band_first_pixel)
{
    if (!p || !nbands) return fail(BFGX_ERR_INVALID, "NULL argument");
    *nbands = p->tiling.nbands;
    if (band_first_pixel) {                       // band b = rings [1 + BR b, 1 + BR (b + 1)): a contiguous RING pixel range
        const int64_t nl4 = 4 * p->hpx.nside;
        for (int b = 0; b <= p->tiling.nbands; ++b) {
            const int64_t ring = std::min<int64_t>(1 + (int64_t)p->tiling.BR * b, nl4);
            if (ring >= nl4) { band_first_pixel[b] = p->hpx.npix; continue; }
            const int64_t ns = p->hpx.nside, q = nl4 - ring;
            band_first_pixel[b] = (ring < ns) ? 2 * ring * (ring - 1)
                                : (ring < 3 * ns) ? p->hpx.ncap + (ring - ns) * 4 * ns
                                : p->hpx.npix - 2 * q * (q + 1);
        }
    }
    return BFGX_OK;
}

// arguments of the gathering regrid that do not depend on the data
static ReachArgs reach_args(const bfgx_plan *p, double cap)
{
    const double ns = (double)p->hpx.nside, om = 1.0 / (3.0 * ns * ns);        // ring 1: z = 1 - 1 / (3 nside^2)
    const double th1 = std::atan2(std::sqrt(om * (2.0 - om)), 1.0 - om);
    return ReachArgs{p->tile_apron, cap, th1, 3.14159265358979323846 - th1};
}

// first pixel of ring `ring` (1 .. 4 nside - 1), npix beyond
static int64_t ring_first_pixel(const Hpx &h, int64_t ring)
{
    const int64_t ns = h.nside, nl4 = 4 * ns;
    if (ring < 1) return 0;
    if (ring >= nl4) return h.npix;
    const int64_t q = nl4 - ring;
    return (ring < ns) ? 2 * ring * (ring - 1) : (ring < 3 * ns) ? h.ncap + (ring - ns) * 4 * ns : h.npix - 2 * q * (q + 1);
}

int bfgx_plan_band_apron(bfgx_plan *p, int32_t band0, int32_t band1, int64_t *olo, int64_t *ohi)
{
    if (!p || !olo || !ohi) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (band0 < 0 || band1 > p->tiling.nbands || band0 > band1) return fail(BFGX_ERR_INVALID, "band range out of bounds");
    const int64_t i0 = 1 + (int64_t)p->tiling.BR * band0, i1 = 1 + (int64_t)p->tiling.BR * band1;      // rings [i0, i1)
    *olo = ring_first_pixel(p->hpx, i0 - p->band_reach);
    *ohi = ring_first_pixel(p->hpx, i1 + p->band_reach);
    return BFGX_OK;
}

int bfgx_plan_reach_rings(bfgx_plan *p, double max_offset, int32_t *rings)
{
    if (!p || !rings) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (!(max_offset >= 0.0)) return fail(BFGX_ERR_INVALID, "max_offset must be >= 0");
    *rings = regrid_reach_rings(p->hpx.nside, std::min(max_offset * 1.001, regrid_cap(p->hpx.nside)));
    return BFGX_OK;
}

int bfgx_plan_set_band_reach(bfgx_plan *p, int32_t rings)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    if (rings < 1 || rings > kReachMax) return fail(BFGX_ERR_INVALID, "band reach must be 1 .. %d rings", kReachMax);
    p->band_reach = rings;
    return BFGX_OK;
}

// reset_far = false: the far-deposit list keeps what earlier calls have listed (the one-shot host entry regrids the sphere in several
// band ranges while the map is still arriving, and applies one list at the end)
// offsets_lo_dev != NULL: split pix_offsets (the parity-grade mode; the one-shot host entry), the low halves of the same pixels [olo, ohi);
// otherwise acc_f64 says whether the one array is fp32 or fp64 (BFGX_ACC_PARITY / AUTO: as bfgx_offsets_bands_device resolves them)
static int regrid_bands_impl(bfgx_plan *p, int32_t band0, int32_t band1, const double *map_in_dev, const void *offsets_dev,
                             int64_t olo, int64_t ohi, int acc_f64, double *out_slice_dev, double *sums_dev, bool reset_far,
                             const float *offsets_lo_dev = nullptr, bool sums_later = false)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (!acc_valid(acc_f64)) return fail(BFGX_ERR_INVALID, "acc_f64 out of range");
    acc_f64 = offsets_lo_dev ? 0 : (resolve_acc(p, acc_f64) != BFGX_ACC_F32 ? 1 : 0);
    if (p->algo != 1) return fail(BFGX_ERR_UNSUPPORTED, "banded regrid needs the tiled algorithm (algo 1)");
    int64_t need_lo = 0, need_hi = 0;
    if (int rc = bfgx_plan_band_apron(p, band0, band1, &need_lo, &need_hi)) return rc;
    if (band0 == band1) {
        // a rank that owns no band (more ranks than bands): empty buffers, whose data pointers are NULL, are fine
        HIP_TRY(hipSetDevice(p->device));
        if (reset_far) HIP_TRY(hipMemsetAsync(p->far.count, 0, sizeof(unsigned long long), p->stream));
        if (sums_dev) HIP_TRY(hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), p->stream));
        return BFGX_OK;
    }
    if (!map_in_dev || !offsets_dev || !out_slice_dev) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (olo > need_lo || ohi < need_hi || olo < 0 || ohi > p->hpx.npix)
        return fail(BFGX_ERR_INVALID, "pix_offsets range [%lld, %lld) does not cover the bands and %d ring(s) either side [%lld, %lld)",
                    (long long)olo, (long long)ohi, p->band_reach, (long long)need_lo, (long long)need_hi);
    HIP_TRY(hipSetDevice(p->device));
    if (reset_far) HIP_TRY(hipMemsetAsync(p->far.count, 0, sizeof(unsigned long long), p->stream));
    const int64_t p0 = ring_first_pixel(p->hpx, 1 + (int64_t)p->tiling.BR * band0);
    const int t0 = p->band_tile0_host[band0], t1 = p->band_tile0_host[band1];
    {
        KernelTimer kt(p, BFGX_K_REGRID);
        const size_t lds = regrid3_lds_bytes(p->tiling.BR, p->tiling.W, (acc_f64 || offsets_lo_dev) ? sizeof(double) : sizeof(float), p->band_reach > 1,
                                             offsets_lo_dev ? sizeof(float) : 0);
        // virtual bases: the kernel indexes every array by global pixel number
        double *out_base = out_slice_dev - p0;
        // every rank gathers with the same fixed reach (so that all of them classify a source pixel the same way)
        const ReachArgs reach = reach_args(p, regrid_cap_for_rings(p->hpx.nside, p->band_reach));
        hipLaunchKernelGGL(tile_apron_kernel, dim3((t1 - t0 + 255) / 256), dim3(256), 0, p->stream, p->hpx, p->tiling, (const float *)nullptr,
                           p->band_reach, reach.cap, t0, t1 - t0, p->tile_apron, (int32_t *)nullptr);
        double *ts = sums_dev ? p->tile_sums : nullptr;
        const dim3 grid(t1 - t0), blk(256);
        int *none = nullptr;
        if (offsets_lo_dev) {
            const float *o = (const float *)offsets_dev - 3 * olo, *ol = offsets_lo_dev - 3 * olo;
            if (p->band_reach == 1)
                hipLaunchKernelGGL((tile_regrid3_kernel<float, double, 0, true>), grid, blk, lds, p->stream, p->hpx, p->tiling, map_in_dev, o, ol, out_base, p->far, reach, ts, t0, t1 - t0, none, (double *)nullptr);
            else
                hipLaunchKernelGGL((tile_regrid3_kernel<float, double, 2, true>), grid, blk, lds, p->stream, p->hpx, p->tiling, map_in_dev, o, ol, out_base, p->far, reach, ts, t0, t1 - t0, none, (double *)nullptr);
        } else if (acc_f64) {
            const double *o = (const double *)offsets_dev - 3 * olo;
            if (p->band_reach == 1)
                hipLaunchKernelGGL((tile_regrid3_kernel<double, double, 0>), grid, blk, lds, p->stream, p->hpx, p->tiling, map_in_dev, o, (decltype(o))nullptr, out_base, p->far, reach, ts, t0, t1 - t0, none, (double *)nullptr);
            else
                hipLaunchKernelGGL((tile_regrid3_kernel<double, double, 2>), grid, blk, lds, p->stream, p->hpx, p->tiling, map_in_dev, o, (decltype(o))nullptr, out_base, p->far, reach, ts, t0, t1 - t0, none, (double *)nullptr);
        } else {
            const float *o = (const float *)offsets_dev - 3 * olo;
            if (p->band_reach == 1)
                hipLaunchKernelGGL((tile_regrid3_kernel<float, float, 0>), grid, blk, lds, p->stream, p->hpx, p->tiling, map_in_dev, o, (decltype(o))nullptr, out_base, p->far, reach, ts, t0, t1 - t0, none, (double *)nullptr);
            else
                hipLaunchKernelGGL((tile_regrid3_kernel<float, float, 2>), grid, blk, lds, p->stream, p->hpx, p->tiling, map_in_dev, o, (decltype(o))nullptr, out_base, p->far, reach, ts, t0, t1 - t0, none, (double *)nullptr);
        }
        HIP_TRY(hipGetLastError());
    }
    if (sums_dev && !sums_later) {             // (sums_later: the caller's next kernel adds the per-tile sums up)
        KernelTimer kt(p, BFGX_K_SUM);
        if (t1 - t0 > 16384) {
            HIP_TRY(hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), p->stream));
            hipLaunchKernelGGL(sum_tiles_multi_kernel, dim3(64), dim3(256), 0, p->stream, t1 - t0, (const double *)(p->tile_sums + 2 * (size_t)t0), sums_dev);
        } else
            hipLaunchKernelGGL(sum_tiles_kernel, dim3(1), dim3(256), 0, p->stream, t1 - t0, (const double *)(p->tile_sums + 2 * (size_t)t0), sums_dev);
        HIP_TRY(hipGetLastError());
    }
    return BFGX_OK;
}

int bfgx_plan_set_route_margin(bfgx_plan *p, int32_t rings)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    if (rings < 0 || rings > 4 * p->hpx.nside) return fail(BFGX_ERR_INVALID, "route margin must be 0 .. 4 nside rings");
    p->route_margin = rings;
    return BFGX_OK;
}

int bfgx_regrid_bands_device(bfgx_plan *p, int32_t band0, int32_t band1, const double *map_in_dev, const void *offsets_dev,
                             int64_t olo, int64_t ohi, int acc_f64, double *out_slice_dev, double *sums_dev)
{
    return regrid_bands_impl(p, band0, band1, map_in_dev, offsets_dev, olo, ohi, acc_f64, out_slice_dev, sums_dev, true);
}

int bfgx_plan_far_apply_device(bfgx_plan *p, double *out_slice_dev, int64_t p0, int64_t p1, unsigned long long *foreign_dev)
{
    if (!p || !out_slice_dev) return fail(BFGX_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    hipLaunchKernelGGL(regrid_far_local_kernel, dim3(64), dim3(256), 0, p->stream, p->far, out_slice_dev, p0, p1, foreign_dev);
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

int bfgx_plan_far_fetch(bfgx_plan *p, int64_t cap, int64_t *pix_host, double *val_host, int64_t *n_host)
{
    if (!p || !n_host) return fail(BFGX_ERR_INVALID, "NULL argument");
    HIP_TRY(hipSetDevice(p->device));
    unsigned long long n = 0;
    HIP_TRY(hipMemcpyAsync(&n, p->far.count, sizeof(n), hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    if ((int64_t)n > p->far.cap) return fail(BFGX_ERR_INVALID, "the regrid's list of far deposits overflowed (%lld entries)", (long long)p->far.cap);
    *n_host = (int64_t)n;
    if (n == 0 || (!pix_host && !val_host)) return BFGX_OK;          // count only
    if ((int64_t)n > cap || !pix_host || !val_host) return fail(BFGX_ERR_INVALID, "far-deposit buffers hold %lld entries, %lld are listed", (long long)cap, (long long)n);
    HIP_TRY(hipMemcpyAsync(pix_host, p->far.pix, sizeof(int64_t) * n, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipMemcpyAsync(val_host, p->far.val, sizeof(double) * n, hipMemcpyDeviceToHost, p->stream));
    HIP_TRY(hipStreamSynchronize(p->stream));
    return BFGX_OK;
}

int bfgx_plan_regrid_stats(bfgx_plan *p, int64_t *far_listed, int32_t *far_overflowed, int32_t *tiles_walked, int32_t *max_reach_rings)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    if (p->algo != 1) return fail(BFGX_ERR_UNSUPPORTED, "regrid statistics exist for the tiled algorithm (algo 1) only");
    HIP_TRY(hipSetDevice(p->device));
    int32_t ctrl[4] = {0, 0, 0, 0};                          // far count (64 bits), overflow of the full-map list, tiles left to the walking kernel
    HIP_TRY(hipMemcpyAsync(ctrl, p->far.count, sizeof(ctrl), hipMemcpyDeviceToHost, p->stream));
    std::vector<int32_t> apron;
    if (max_reach_rings) {
        apron.resize(2 * (size_t)p->tiling.ntiles);
        HIP_TRY(hipMemcpyAsync(apron.data(), p->tile_apron, sizeof(int32_t) * apron.size(), hipMemcpyDeviceToHost, p->stream));
    }
    HIP_TRY(hipStreamSynchronize(p->stream));
    unsigned long long n = 0;
    std::memcpy(&n, ctrl, sizeof(n));
    if (far_listed) *far_listed = (int64_t)n;
    if (far_overflowed) *far_overflowed = ctrl[2];
    if (tiles_walked) *tiles_walked = ctrl[3];
    if (max_reach_rings) {
        int32_t m = 0;
        for (int t = 0; t < p->tiling.ntiles; ++t) m = std::max(m, apron[2 * (size_t)t]);
        *max_reach_rings = m;
    }
    return BFGX_OK;
}

int bfgx_offsets_device(bfgx_plan *p, const bfgx_catalog *cat, void *offsets_dev, int acc_f64)
{
    if (int rc = check_catalog(p, cat)) return rc;
    if (!offsets_dev) return fail(BFGX_ERR_INVALID, "offsets pointer is NULL");
    if (p->model.tab.logv) return fail(BFGX_ERR_INVALID, "displacement read-out needs a table with log_values = 0");
    if (!acc_valid(acc_f64)) return fail(BFGX_ERR_INVALID, "acc_f64 must be BFGX_ACC_AUTO (-1), 0 (f32), 1 (f64) or BFGX_ACC_PARITY (3)");
    acc_f64 = resolve_acc(p, acc_f64);
    HIP_TRY(hipSetDevice(p->device));
    if (p->algo == 1) {      // tile-owned: every element of offsets is overwritten, no zero-fill needed
        if (int rc = launch_prep_and_bin(p, cat, 1, acc_f64 != 0)) return rc;
        if (p->blocking_growth) if (int rc = ensure_entry_capacity(p, cat)) return rc;
        if (acc_f64 == BFGX_ACC_PARITY) {      // (resolve_acc: the fast kernel takes this table) hi [npix][3], then lo [npix][3]
            p->offsets_lo = (float *)offsets_dev + 3 * (size_t)p->hpx.npix;
            const int rc = launch_tile_scatter<MODE_OFFSETS, float>(p, (float *)offsets_dev);
            p->offsets_lo = nullptr;
            return rc;
        }
        if (acc_f64) return launch_tile_scatter<MODE_OFFSETS, double>(p, (double *)offsets_dev);
        return launch_tile_scatter<MODE_OFFSETS, float>(p, (float *)offsets_dev);
    }
    if (p->NC != 4) return fail(BFGX_ERR_UNSUPPORTED, "tables with extra parameter axes need algo 1 (LDS tiles)");
    if (int rc = launch_prep(p, cat, 1, false, false, true)) return rc;
    if (acc_f64) return launch_scatter<MODE_OFFSETS, double>(p, cat->n, (double *)offsets_dev, nullptr);
    return launch_scatter<MODE_OFFSETS, float>(p, cat->n, (float *)offsets_dev, nullptr);
}

// K0 + K1 / K3 for the tiles of bands [band0, band1) only: out_slice_dev points at the first pixel of band0
static int bands_scatter(bfgx_plan *p, const bfgx_catalog *cat, int32_t band0, int32_t band1, void *out_slice_dev, int acc_f64, bool paint,
                         float *lo_slice_dev = nullptr)
{
    // lo_slice_dev (displacements only): the parity-grade mode -- out_slice_dev takes the fp32 high halves of pix_offsets, lo_slice_dev the low halves of the
    // same pixels (a caller that keeps both on ONE device: nothing of them crosses a link)
    if (int rc = check_catalog(p, cat)) return rc;
    if (p->algo != 1) return fail(BFGX_ERR_UNSUPPORTED, "band-restricted passes need the tiled algorithm (algo 1)");
    if (band0 < 0 || band1 > p->tiling.nbands || band0 > band1) return fail(BFGX_ERR_INVALID, "band range out of bounds");
    if ((p->model.tab.logv != 0) != paint) return fail(BFGX_ERR_INVALID, paint ? "profile painting needs a table with log_values = 1"
                                                                               : "displacement read-out needs a table with log_values = 0");
    // a rank that owns no band (more ranks than bands) passes the data pointer of an empty buffer, which is NULL: nothing to do
    if (band0 == band1) return BFGX_OK;
    if (!out_slice_dev) return fail(BFGX_ERR_INVALID, "output pointer is NULL");
    HIP_TRY(hipSetDevice(p->device));
    const int64_t p0 = ring_first_pixel(p->hpx, 1 + (int64_t)p->tiling.BR * band0);
    p->k1_tile_lo = p->band_tile0_host[band0];
    p->k1_tile_n = p->band_tile0_host[band1] - p->k1_tile_lo;
    struct Reset { bfgx_plan *p; ~Reset() { p->k1_tile_lo = 0; p->k1_tile_n = -1; } } reset{p};
    const bool split = !paint && lo_slice_dev != nullptr && use_fast(p);
    if (!paint) {        // ONE array of pix_offsets (what travels between ranks): the parity-grade mode is served by fp64 throughout there
        if (!acc_valid(acc_f64)) return fail(BFGX_ERR_INVALID, "acc_f64 out of range");
        acc_f64 = split ? 0 : (resolve_acc(p, acc_f64) != BFGX_ACC_F32 ? 1 : 0);
    }
    if (acc_f64 < 0 || acc_f64 > (paint ? 2 : 1)) return fail(BFGX_ERR_INVALID, "acc_f64 out of range");
    const bool mixed = paint && acc_f64 == 2 && use_fast(p);
    if (int rc = launch_prep_and_bin(p, cat, paint ? 0 : 1, (acc_f64 != 0 && !mixed) || split)) return rc;
    if (p->blocking_growth) if (int rc = ensure_entry_capacity(p, cat)) return rc;
    p->paint_pair_f32 = mixed;
    struct ResetP { bfgx_plan *p; ~ResetP() { p->paint_pair_f32 = false; } } resetp{p};
    // virtual base: the kernels index the output by global pixel number
    if (paint) {
        if (acc_f64) return launch_tile_scatter<MODE_PAINT, double>(p, (double *)out_slice_dev - p0);
        return launch_tile_scatter<MODE_PAINT, float>(p, (float *)out_slice_dev - p0);
    }
    // K1's flush leaves the largest |offset|^2 of every tile it stores (bfgx_bands_max_offset2_device reduces them: the reach of the
    // regrid then needs no second pass over the slice)
    p->omax_from_k1 = true;
    struct ResetO { bfgx_plan *p; ~ResetO() { p->omax_from_k1 = false; } } reseto{p};
    if (split) {         // fp64 pair records + the parity-grade pair functions, high and low halves stored (launch_tile_scatter)
        p->offsets_lo = lo_slice_dev - 3 * p0;
        struct ResetL { bfgx_plan *p; ~ResetL() { p->offsets_lo = nullptr; } } resetl{p};
        return launch_tile_scatter<MODE_OFFSETS, float>(p, (float *)out_slice_dev - 3 * p0);
    }
    if (acc_f64) return launch_tile_scatter<MODE_OFFSETS, double>(p, (double *)out_slice_dev - 3 * p0);
    return launch_tile_scatter<MODE_OFFSETS, float>(p, (float *)out_slice_dev - 3 * p0);
}

int bfgx_offsets_bands_device(bfgx_plan *p, const bfgx_catalog *cat, int32_t band0, int32_t band1, void *offsets_slice_dev, int acc_f64)
{
    return bands_scatter(p, cat, band0, band1, offsets_slice_dev, acc_f64, false);
}

int bfgx_paint_bands_device(bfgx_plan *p, const bfgx_catalog *cat, int32_t band0, int32_t band1, void *map_slice_dev, int acc_f64)
{
    return bands_scatter(p, cat, band0, band1, map_slice_dev, acc_f64, true);
}

// One rank's share of a resident multi-GPU BaryonifyShell step after the routing, as ONE enqueue-only call (Parallelize.py:250-318 has one call
// drive all workers): K0 + binning + K1 for the bands [B0, B1) (the rank's own and, with a route margin, the band either side: the aprons of
// its regrid) into offsets_dev (pixels from the first of band B0), the banded regrid of ITS bands [b0, b1) into out_slice_dev, the listed far
// deposits that fall into its pixels added, the others counted into *foreign_dev, the two sums of the mass check.  One memset (the binning's)
// zeroes every control word; eight launches.
int bfgx_offsets_regrid_bands_device(bfgx_plan *p, const bfgx_catalog *cat, int32_t B0, int32_t B1, void *offsets_dev, int acc_f64,
                                     int32_t b0, int32_t b1, const double *map_in_dev, double *out_slice_dev, double *sums_dev,
                                     unsigned long long *foreign_dev)
{
    if (!p || !cat) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (B0 < 0 || B1 > p->tiling.nbands || B0 > b0 || b1 > B1 || b0 > b1) return fail(BFGX_ERR_INVALID, "band ranges must nest: [B0, B1) around [b0, b1)");
    if (!acc_valid(acc_f64)) return fail(BFGX_ERR_INVALID, "acc_f64 out of range");
    const int64_t wlo = ring_first_pixel(p->hpx, 1 + (int64_t)p->tiling.BR * B0), whi = ring_first_pixel(p->hpx, 1 + (int64_t)p->tiling.BR * B1);
    // the parity-grade mode: both halves of pix_offsets stay on this device (24 bytes per pixel, as fp64 would take): high halves
    // [whi - wlo][3] fp32, then the low halves
    float *lo = (resolve_acc(p, acc_f64) == BFGX_ACC_PARITY && use_fast(p) && offsets_dev) ? (float *)offsets_dev + 3 * (size_t)(whi - wlo) : nullptr;
    if (int rc = bands_scatter(p, cat, B0, B1, offsets_dev, acc_f64, false, lo)) return rc;       // (its memset also zeroes the far list's counter)
    if (b0 == b1) {
        if (sums_dev) HIP_TRY(hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), p->stream));
        return BFGX_OK;
    }
    if (int rc = regrid_bands_impl(p, b0, b1, map_in_dev, offsets_dev, wlo, whi, acc_f64, out_slice_dev, sums_dev, false, lo, true)) return rc;
    const int64_t p0 = ring_first_pixel(p->hpx, 1 + (int64_t)p->tiling.BR * b0), p1 = ring_first_pixel(p->hpx, 1 + (int64_t)p->tiling.BR * b1);
    const int t0 = p->band_tile0_host[b0], t1 = p->band_tile0_host[b1];
    hipLaunchKernelGGL(regrid_far_local_kernel, dim3(64), dim3(256), 0, p->stream, p->far, out_slice_dev, p0, p1, foreign_dev, t1 - t0,
                       sums_dev ? (const double *)(p->tile_sums + 2 * (size_t)t0) : (const double *)nullptr, sums_dev);
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

int bfgx_bands_max_offset2_device(bfgx_plan *p, int32_t band0, int32_t band1, float *out_dev)
{
    if (!p || !out_dev) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (p->algo != 1) return fail(BFGX_ERR_UNSUPPORTED, "band-restricted passes need the tiled algorithm (algo 1)");
    if (band0 < 0 || band1 > p->tiling.nbands || band0 > band1) return fail(BFGX_ERR_INVALID, "band range out of bounds");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float), p->stream));
    const int t0 = p->band_tile0_host[band0], nt = p->band_tile0_host[band1] - t0;
    if (nt > 0) {
        hipLaunchKernelGGL(max_bits_kernel, dim3(1), dim3(1024), 0, p->stream, nt, (const unsigned *)p->tile_omax + t0, (unsigned *)out_dev);
        HIP_TRY(hipGetLastError());
    }
    return BFGX_OK;
}

int bfgx_max_offset2_device(bfgx_plan *p, const void *offsets_dev, int64_t npixels, int acc_f64, float *out_dev)
{
    if (!p || !out_dev || (npixels > 0 && !offsets_dev)) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (npixels < 0) return fail(BFGX_ERR_INVALID, "npixels < 0");
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemsetAsync(out_dev, 0, sizeof(float), p->stream));
    if (npixels > 0) {
        const unsigned grid = (unsigned)std::min<int64_t>((npixels + 255) / 256, 1024);
        if (acc_valid(acc_f64) && resolve_acc(p, acc_f64) != BFGX_ACC_F32) hipLaunchKernelGGL(max_offset_kernel<double>, dim3(grid), dim3(256), 0, p->stream, npixels, (const double *)offsets_dev, (unsigned *)out_dev);
        else hipLaunchKernelGGL(max_offset_kernel<float>, dim3(grid), dim3(256), 0, p->stream, npixels, (const float *)offsets_dev, (unsigned *)out_dev);
        HIP_TRY(hipGetLastError());
    }
    return BFGX_OK;
}

// shared argument checks / set-up of the two routing passes
static int route_args(bfgx_plan *p, const bfgx_catalog *cat, int32_t world, const int32_t *ring_bounds, int32_t ncols, const double *const *cols,
                      RouteArgs &a)
{
    if (!p || !cat || !ring_bounds || !cols) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (world < 1 || world > kRouteMaxRanks) return fail(BFGX_ERR_INVALID, "routing supports 1 .. %d ranks", kRouteMaxRanks);
    if (ncols < 1 || ncols > kRouteMaxCols) return fail(BFGX_ERR_INVALID, "routing packs 1 .. %d columns", kRouteMaxCols);
    if (cat->n < 0) return fail(BFGX_ERR_INVALID, "catalog size < 0");
    std::memset(&a, 0, sizeof(a));
    a.world = world; a.ncols = ncols;
    for (int j = 0; j <= world; ++j) {
        if (j > 0 && ring_bounds[j] < ring_bounds[j - 1]) return fail(BFGX_ERR_INVALID, "ring bounds must ascend");
        a.bounds[j] = ring_bounds[j];
    }
    for (int c = 0; c < ncols; ++c) { if (cat->n > 0 && !cols[c]) return fail(BFGX_ERR_INVALID, "column pointer is NULL"); a.col[c] = cols[c]; }
    return BFGX_OK;
}

int bfgx_route_count_device(bfgx_plan *p, int64_t n, const int32_t *rings_dev, int32_t world, const int32_t *ring_bounds, int32_t *counts_dev)
{
    bfgx_catalog c;
    std::memset(&c, 0, sizeof(c));
    c.n = n;
    RouteArgs a;
    const double *none[1] = {nullptr};
    if (n < 0 || !counts_dev || (n > 0 && !rings_dev)) return fail(BFGX_ERR_INVALID, "NULL argument");
    c.n = 0;                                                    // (no columns are read in the counting pass)
    if (int rc = route_args(p, &c, world, ring_bounds, 1, none, a)) return rc;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemsetAsync(counts_dev, 0, sizeof(int32_t) * world, p->stream));
    if (n > 0) {
        hipLaunchKernelGGL(route_halos_kernel<false>, dim3((unsigned)std::min<int64_t>((n + 255) / 256, kRouteGrid)), dim3(256), 0, p->stream, a, n, rings_dev, counts_dev,
                           (int32_t *)nullptr, (double *)nullptr);
        HIP_TRY(hipGetLastError());
    }
    return BFGX_OK;
}

int bfgx_route_fill_device(bfgx_plan *p, int64_t n, const int32_t *rings_dev, int32_t world, const int32_t *ring_bounds, const int64_t *start,
                           int32_t ncols, const double *const *cols_dev, int32_t *cursor_dev, double *rows_dev)
{
    bfgx_catalog c;
    std::memset(&c, 0, sizeof(c));
    c.n = n;
    RouteArgs a;
    if (n < 0 || !start || !cursor_dev || (n > 0 && (!rings_dev || !rows_dev))) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (int rc = route_args(p, &c, world, ring_bounds, ncols, cols_dev, a)) return rc;
    for (int j = 0; j < world; ++j) a.start[j] = start[j];
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemsetAsync(cursor_dev, 0, sizeof(int32_t) * world, p->stream));
    if (n > 0) {
        hipLaunchKernelGGL(route_halos_kernel<true>, dim3((unsigned)std::min<int64_t>((n + 255) / 256, kRouteGrid)), dim3(256), 0, p->stream, a, n, rings_dev, (int32_t *)nullptr,
                           cursor_dev, rows_dev);
        HIP_TRY(hipGetLastError());
    }
    return BFGX_OK;
}

int bfgx_route_pack_device(bfgx_plan *p, int64_t n, const int32_t *rings_dev, int32_t world, const int32_t *ring_bounds, int64_t blockcap,
                           int32_t ncols, const double *const *cols_dev, int32_t *cursor_dev, double *blocks_dev, int32_t *overflow_dev)
{
    bfgx_catalog c;
    std::memset(&c, 0, sizeof(c));
    c.n = n;
    RouteArgs a;
    if (n < 0 || !cursor_dev || !blocks_dev || !overflow_dev || (n > 0 && !rings_dev)) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (blockcap < 1) return fail(BFGX_ERR_INVALID, "block capacity must be >= 1");
    if (int rc = route_args(p, &c, world, ring_bounds, ncols, cols_dev, a)) return rc;
    a.blockcap = blockcap; a.overflow = overflow_dev;
    HIP_TRY(hipSetDevice(p->device));
    HIP_TRY(hipMemsetAsync(cursor_dev, 0, sizeof(int32_t) * world, p->stream));
    hipLaunchKernelGGL(route_blank_kernel, dim3((unsigned)std::min<int64_t>(((int64_t)world * blockcap + 255) / 256, 1024)), dim3(256), 0, p->stream,
                       world, ncols, blockcap, blocks_dev);
    HIP_TRY(hipGetLastError());
    if (n > 0) {
        hipLaunchKernelGGL(route_halos_kernel<true>, dim3((unsigned)std::min<int64_t>((n + 255) / 256, kRouteGrid)), dim3(256), 0, p->stream, a, n, rings_dev, (int32_t *)nullptr,
                           cursor_dev, blocks_dev);
        HIP_TRY(hipGetLastError());
    }
    return BFGX_OK;
}

// The routing of one resident multi-GPU step as ONE call and two launches (bfgx_disc_rings_device + bfgx_route_pack_device were four, with a
// fill): route_prepare_kernel (NaN into column 0 of every block, the cursors zeroed) and route_step_kernel (ring range per halo, rows
// packed by destination).  send_blocks_dev holds (world - 1) blocks [ncols][blockcap] in rank order WITHOUT this rank, recv_blocks_dev
// world blocks: the other ranks' in rank order (what all_to_all_single delivers when the split towards oneself is empty), this rank's
// own rows LAST -- they never enter the collective.  K0 reads the received blocks as they are (bfgx_plan_set_catalog_blocks).
int bfgx_route_step_device(bfgx_plan *p, const bfgx_catalog *cat, int32_t world, int32_t rank, const int32_t *ring_bounds, int64_t blockcap,
                           int32_t ncols, const double *const *cols_dev, int32_t *cursor_dev, double *send_blocks_dev, double *recv_blocks_dev,
                           int32_t *overflow_dev)
{
    if (!p || !cat || !cursor_dev || !recv_blocks_dev || !overflow_dev || (world > 1 && !send_blocks_dev)) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (rank < 0 || rank >= world) return fail(BFGX_ERR_INVALID, "rank out of range");
    if (blockcap < 1) return fail(BFGX_ERR_INVALID, "block capacity must be >= 1");
    if (cat->n > 0 && (!cat->M || !cat->z || !cat->dec)) return fail(BFGX_ERR_INVALID, "catalog column pointer is NULL");
    RouteStepArgs s;
    std::memset(&s, 0, sizeof(s));
    if (int rc = route_args(p, cat, world, ring_bounds, ncols, cols_dev, s.r)) return rc;
    s.r.blockcap = blockcap; s.r.overflow = overflow_dev;
    s.rank = rank; s.margin = p->route_margin;
    s.M = cat->M; s.z = cat->z; s.dec = cat->dec;
    s.send = send_blocks_dev; s.recv = recv_blocks_dev;
    HIP_TRY(hipSetDevice(p->device));
    const int64_t nblank = (int64_t)(2 * world - 1) * blockcap;
    hipLaunchKernelGGL(route_prepare_kernel, dim3((unsigned)std::min<int64_t>((nblank + 255) / 256, 1024)), dim3(256), 0, p->stream, world, ncols, blockcap,
                       send_blocks_dev, recv_blocks_dev, cursor_dev);
    if (cat->n > 0)
        // (2048 workgroups: eight waves per SIMD hide the latency of the strided column stores; each adds to the per-destination cursors once)
        hipLaunchKernelGGL(route_step_kernel, dim3((unsigned)std::min<int64_t>((cat->n + 255) / 256, 4 * kRouteGrid)), dim3(256), 0, p->stream, s, p->model, p->hpx,
                           cat->n, cursor_dev);
    HIP_TRY(hipGetLastError());
    return BFGX_OK;
}

// the catalogs of the following K0 launches are BLOCKED: rows halo j's columns at (j / rows) * stride + (j % rows) from the column pointers (what
// an all_to_all of fixed-capacity blocks [block][column][rows] delivers: stride = ncols * rows); rows = 0: plain columns again
int bfgx_plan_set_catalog_blocks(bfgx_plan *p, int64_t rows, int64_t stride)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    if (rows < 0 || (rows > 0 && stride < rows)) return fail(BFGX_ERR_INVALID, "blocked catalog: rows >= 0, stride >= rows");
    p->cat_blk_rows = rows; p->cat_blk_stride = rows > 0 ? stride : 0;
    return BFGX_OK;
}

int bfgx_plan_tile_shape(bfgx_plan *p, int32_t *rings_per_band, int32_t *max_columns)
{
    if (!p) return fail(BFGX_ERR_INVALID, "NULL plan");
    if (rings_per_band) *rings_per_band = p->tiling.BR;
    if (max_columns) *max_columns = p->tiling.W;
    return BFGX_OK;
}

int bfgx_disc_rings_device(bfgx_plan *p, const bfgx_catalog *cat, int32_t *rings_dev)
{
    // (uses none of the plan's per-halo workspace: the catalog may be larger than the plan's max_halos)
    if (!p || !cat) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (cat->n < 0) return fail(BFGX_ERR_INVALID, "catalog size < 0");
    if (cat->n > 0 && (!cat->M || !cat->z || !cat->dec)) return fail(BFGX_ERR_INVALID, "catalog column pointer is NULL");
    if (cat->n > 0 && !rings_dev) return fail(BFGX_ERR_INVALID, "rings pointer is NULL");
    HIP_TRY(hipSetDevice(p->device));
    if (cat->n > 0) {
        hipLaunchKernelGGL(disc_rings_kernel, dim3((unsigned)((cat->n + 255) / 256)), dim3(256), 0, p->stream, p->model, p->hpx, cat->n, cat->M, cat->z,
                           cat->dec, rings_dev, p->route_margin);
        HIP_TRY(hipGetLastError());
    }
    return BFGX_OK;
}

int bfgx_paint_device(bfgx_plan *p, const bfgx_catalog *cat, void *map_out_dev, int acc_f64)
{
    if (int rc = check_catalog(p, cat)) return rc;
    if (!map_out_dev) return fail(BFGX_ERR_INVALID, "map pointer is NULL");
    if (!p->model.tab.logv) return fail(BFGX_ERR_INVALID, "profile painting needs a table with log_values = 1");
    HIP_TRY(hipSetDevice(p->device));
    if (acc_f64 < 0 || acc_f64 > 2) return fail(BFGX_ERR_INVALID, "acc_f64 must be 0, 1 or 2");
    if (p->algo == 1) {
        const bool mixed = acc_f64 == 2 && use_fast(p);          // (tables the fast kernel cannot take: fp64 throughout)
        if (int rc = launch_prep_and_bin(p, cat, 0, acc_f64 != 0 && !mixed)) return rc;
        if (p->blocking_growth) if (int rc = ensure_entry_capacity(p, cat)) return rc;
        p->paint_pair_f32 = mixed;
        struct Reset { bfgx_plan *p; ~Reset() { p->paint_pair_f32 = false; } } reset{p};
        if (acc_f64) return launch_tile_scatter<MODE_PAINT, double>(p, (double *)map_out_dev);
        return launch_tile_scatter<MODE_PAINT, float>(p, (float *)map_out_dev);
    }
    if (p->NC != 4) return fail(BFGX_ERR_UNSUPPORTED, "tables with extra parameter axes need algo 1 (LDS tiles)");
    if (int rc = launch_prep(p, cat, 0, false, false, true)) return rc;
    if (acc_f64) return launch_scatter<MODE_PAINT, double>(p, cat->n, (double *)map_out_dev, nullptr);
    return launch_scatter<MODE_PAINT, float>(p, cat->n, (float *)map_out_dev, nullptr);
}

}  // extern "C"

// the gathering regrid (algo 1): [largest |offset|^2 per tile unless K1 left it], aprons, lean gather, gather with the ring
// walk over the tiles that need it, far list / repair
template <typename ACC, typename real, bool SPLIT = false>
static void launch_regrid_gather(bfgx_plan *p, const double *map_in_dev, const ACC *o, const ACC *o_lo, double *map_out_dev, double *ts, bool from_k1, double *sums_dev)
{
    const size_t lds = regrid3_lds_bytes(p->tiling.BR, p->tiling.W, sizeof(real)),
                 lds_walk = regrid3_lds_bytes(p->tiling.BR, p->tiling.W, sizeof(real), true, SPLIT ? sizeof(ACC) : sizeof(real));
    FarList far = p->far;
    far.overflow = p->far_overflow_full;
    const int nt = p->tiling.ntiles, nfix = std::min(nt, 2 * p->num_cus), nwalk = std::min(nt, 4 * p->num_cus);
    const ReachArgs reach = reach_args(p, regrid_cap(p->hpx.nside));
    if (!from_k1)
        hipLaunchKernelGGL(tile_reach_kernel<ACC>, dim3(nt), dim3(256), 0, p->stream, p->hpx, p->tiling, o, p->tile_omax);
    hipLaunchKernelGGL(tile_apron_kernel, dim3((nt + 255) / 256), dim3(256), 0, p->stream, p->hpx, p->tiling, (const float *)p->tile_omax, 0,
                       reach.cap, 0, nt, p->tile_apron, p->regrid_todo, p->regrid_lean);
    // (the lean kernel: a persistent grid over the tiles whose reach is one ring -- all of them on a table of sub-pixel moves, none on the S19 table)
    hipLaunchKernelGGL((tile_regrid3_kernel<ACC, real, 0, SPLIT>), dim3(std::min(nt, 8 * p->num_cus)), dim3(256), lds, p->stream, p->hpx, p->tiling, map_in_dev, o, o_lo,
                       map_out_dev, far, reach, ts, -1, nt, p->regrid_lean, (double *)nullptr);
    hipLaunchKernelGGL((tile_regrid3_kernel<ACC, real, 2, SPLIT>), dim3(nwalk), dim3(256), lds_walk, p->stream, p->hpx, p->tiling, map_in_dev, o, o_lo,
                       map_out_dev, far, reach, ts, -1, nt, p->regrid_todo, (double *)nullptr);
    // the two sums of the mass check: added up by the last launch, or -- above 16384 tiles, where that single workgroup's loop
    // takes 0.2 - 0.6 ms -- by 64 workgroups afterwards
    const bool many = nt > 16384 && sums_dev != nullptr && ts != nullptr;
    hipLaunchKernelGGL((tile_regrid3_kernel<ACC, real, 1, SPLIT>), dim3(nfix), dim3(256), lds, p->stream, p->hpx, p->tiling, map_in_dev, o, o_lo,
                       map_out_dev, far, reach, ts, -1, nt, p->regrid_todo, many ? (double *)nullptr : sums_dev, p->regrid_lean);
    if (many) {
        (void)hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), p->stream);
        hipLaunchKernelGGL(sum_tiles_multi_kernel, dim3(64), dim3(256), 0, p->stream, nt, (const double *)ts, sums_dev);
    }
}

namespace { void dep_release_all(); void fft_release_all(); void gcache_release_all(); void scache_release_all(); void deprec_release_all(); }

extern "C" {

static int regrid_impl(bfgx_plan *p, const double *map_in_dev, const void *offsets_dev, int acc_f64, double *map_out_dev, double *sums_dev,
                       bool from_k1)
{
    if (!p || !map_in_dev || !offsets_dev || !map_out_dev) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (!acc_valid(acc_f64)) return fail(BFGX_ERR_INVALID, "acc_f64 must be BFGX_ACC_AUTO (-1), 0 (f32), 1 (f64) or BFGX_ACC_PARITY (3)");
    acc_f64 = resolve_acc(p, acc_f64);
    HIP_TRY(hipSetDevice(p->device));
    {
    KernelTimer kt(p, BFGX_K_REGRID);
    if (p->algo == 1) {
        // gathering form: every output pixel is stored once by the tile that owns it (no atomics, no zero-fill needed);
        // the few deposits that need the generic route are listed and added afterwards
        // entries, overflow, tiles left to the walking kernel (the fused call's binning step has zeroed them with its counters)
        if (!from_k1) HIP_TRY(hipMemsetAsync(p->far.count, 0, 4 * sizeof(int32_t), p->stream));
        double *ts = sums_dev ? p->tile_sums : nullptr;
        if (acc_f64 == BFGX_ACC_PARITY)      // pix_offsets as two fp32 arrays (hi, then lo behind the npix high triples), fp64 geometry
            launch_regrid_gather<float, double, true>(p, map_in_dev, (const float *)offsets_dev, (const float *)offsets_dev + 3 * (size_t)p->hpx.npix,
                                                      map_out_dev, ts, from_k1, sums_dev);
        else if (acc_f64) launch_regrid_gather<double, double>(p, map_in_dev, (const double *)offsets_dev, (const double *)nullptr, map_out_dev, ts, from_k1, sums_dev);
        else launch_regrid_gather<float, float>(p, map_in_dev, (const float *)offsets_dev, (const float *)nullptr, map_out_dev, ts, from_k1, sums_dev);
    } else {
        const unsigned grid = (unsigned)((p->hpx.npix + 255) / 256);
        if (acc_f64)
            hipLaunchKernelGGL(regrid_kernel<double>, dim3(grid), dim3(256), 0, p->stream, p->hpx, map_in_dev,
                               (const double *)offsets_dev, map_out_dev);
        else
            hipLaunchKernelGGL(regrid_kernel<float>, dim3(grid), dim3(256), 0, p->stream, p->hpx, map_in_dev,
                               (const float *)offsets_dev, map_out_dev);
    }
    }
    HIP_TRY(hipGetLastError());
    if (sums_dev) {
        // algo 1: the gather kernels left per-tile {sum of source values, sum of deposits} and the regrid's last launch added them up
        if (p->algo != 1) {
            KernelTimer kt(p, BFGX_K_SUM);
            HIP_TRY(hipMemsetAsync(sums_dev, 0, 2 * sizeof(double), p->stream));
            hipLaunchKernelGGL(sum2_kernel, dim3(1024), dim3(256), 0, p->stream, p->hpx.npix, map_in_dev,
                               (const double *)map_out_dev, sums_dev);
            HIP_TRY(hipGetLastError());
        }
    }
    return BFGX_OK;
}

int bfgx_regrid_device(bfgx_plan *p, const double *map_in_dev, const void *offsets_dev, int acc_f64,
                       double *map_out_dev, double *sums_dev)
{
    return regrid_impl(p, map_in_dev, offsets_dev, acc_f64, map_out_dev, sums_dev, false);
}

int bfgx_baryonify_device(bfgx_plan *p, const bfgx_catalog *cat, const double *map_in_dev, void *offsets_work_dev, int acc_f64,
                          double *map_out_dev, double *sums_dev)
{
    if (!p || !map_in_dev || !offsets_work_dev || !map_out_dev) return fail(BFGX_ERR_INVALID, "NULL argument");
    const bool fused = p->algo == 1;         // K1's flush leaves the largest |offset|^2 of every tile for the regrid's aprons
    p->omax_from_k1 = fused;
    const int rc = bfgx_offsets_device(p, cat, offsets_work_dev, acc_f64);
    p->omax_from_k1 = false;
    if (rc) return rc;
    return regrid_impl(p, map_in_dev, offsets_work_dev, acc_f64, map_out_dev, sums_dev, fused);
}

int bfgx_count_pairs_device(bfgx_plan *p, const bfgx_catalog *cat, int fallback4, int64_t *counts_dev, int64_t *total_host)
{
    if (int rc = check_catalog(p, cat)) return rc;
    HIP_TRY(hipSetDevice(p->device));
    if (p->algo == 1 && !counts_dev) {       // census through the tile path (checks the binning is complete)
        HIP_TRY(hipMemsetAsync(p->pair_total, 0, sizeof(unsigned long long), p->stream));
        if (int rc = launch_prep_and_bin(p, cat, fallback4, false)) return rc;
        if (int rc = launch_tile_scatter<MODE_COUNT, float>(p, (float *)nullptr)) return rc;
        unsigned long long tot = 0;
        HIP_TRY(hipMemcpyAsync(&tot, p->pair_total, sizeof(tot), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        if (total_host) *total_host = (int64_t)tot;
        return bfgx_plan_status(p);
    }
    int64_t *counts = counts_dev;
    void *tmp = nullptr;
    if (!counts) {
        HIP_TRY(hipMalloc(&tmp, sizeof(int64_t) * (size_t)(cat->n > 0 ? cat->n : 1)));
        counts = (int64_t *)tmp;
    }
    int rc = launch_prep(p, cat, fallback4, false, false, true);
    if (!rc) rc = launch_scatter<MODE_COUNT, float>(p, cat->n, (float *)nullptr, counts);
    if (!rc && total_host) {
        std::vector<int64_t> h((size_t)cat->n);
        if (hipMemcpyAsync(h.data(), counts, sizeof(int64_t) * (size_t)cat->n, hipMemcpyDeviceToHost, p->stream) != hipSuccess ||
            hipStreamSynchronize(p->stream) != hipSuccess)
            rc = fail(BFGX_ERR_HIP, "count copy-back failed");
        else {
            int64_t tot = 0;
            for (int64_t v : h) tot += v;
            *total_host = tot;
        }
    } else if (!rc) {
        if (hipStreamSynchronize(p->stream) != hipSuccess) rc = fail(BFGX_ERR_HIP, "stream sync failed");
    }
    if (tmp) (void)hipFree(tmp);
    return rc;
}

// ------------------------------------------------------------------------------ one-shot host API
namespace {

struct DevBuf {
    void *p = nullptr;
    ~DevBuf() { if (p) (void)hipFree(p); }
    int alloc(size_t bytes) { return hipMalloc(&p, bytes ? bytes : 1) == hipSuccess ? 0 : 1; }
};

// Every one-shot host entry leaves NOTHING in flight when it returns -- on success and on every error path: asynchronous copies read the
// caller's arrays (page-locked for the call, or pinned on the fly by the runtime) and write into the caller's result; a copy that is
// still running when the caller frees or reuses those arrays is a GPU memory fault at a host address.  Declared first in an entry, it
// is destroyed last (after the Pin objects have unregistered): its streams are drained at scope exit whatever the return path.
struct DrainOnExit {
    hipStream_t *s[4] = {nullptr, nullptr, nullptr, nullptr};
    bool null_stream = false;
    ~DrainOnExit()
    {
        for (hipStream_t *q : s) if (q && *q) (void)hipStreamSynchronize(*q);
        if (null_stream) (void)hipStreamSynchronize(nullptr);
    }
};

// A caller's host array as the source / destination of the ASYNCHRONOUS copies of a one-shot entry:
//  * already page-locked by the caller (bfgx_host_alloc): used as it is;
//  * >= 32 MiB -- beyond glibc's largest mmap threshold, i.e. a mapping of its own that shares no page with another object: page-locked in
//    place for the call (hipHostRegister);
//  * smaller: NEVER page-locked in place.  A small numpy array lives in the process heap next to other live objects; page-locking and
//    unlocking those pages call after call left a later, ordinary pageable copy from the same heap region reading through a mapping that
//    was gone (seen twice in round 4 as "Memory access fault by GPU ... on address <heap address>").  Small arrays take the entry's
//    synchronous route; `stage` (tests force the streamed route on small maps: BFGX_PIPE_CHUNKS) goes through a page-locked buffer of ours.
constexpr size_t kPinInPlaceMin = (size_t)32 << 20;
struct HostSpan {
    void *user = nullptr, *use = nullptr;
    size_t bytes = 0;
    bool registered = false, staged = false, is_out = false;
    hipStream_t *streams[3] = {nullptr, nullptr, nullptr};       // copies of the span may be in flight on these when an error returns early
    bool open(const void *q, size_t nb, bool out, bool stage)
    {
        user = (void *)q; bytes = nb; is_out = out;
        hipPointerAttribute_t at;
        if (hipPointerGetAttributes(&at, q) == hipSuccess && at.type == hipMemoryTypeHost) { use = user; return true; }
        (void)hipGetLastError();
        if (nb >= kPinInPlaceMin) {
            if (hipHostRegister(user, nb, hipHostRegisterDefault) == hipSuccess) {
                registered = true; use = user;
                ++g_host_pinned_in_place;
                long long m = g_host_pin_min_bytes.load();
                while ((m < 0 || (long long)nb < m) && !g_host_pin_min_bytes.compare_exchange_weak(m, (long long)nb)) {}
                return true;
            }
            (void)hipGetLastError();
            return false;
        }
        if (!stage) return false;
        if (hipHostMalloc(&use, nb ? nb : 1, hipHostMallocDefault) != hipSuccess) { (void)hipGetLastError(); use = nullptr; return false; }
        staged = true;
        ++g_host_staged;
        if (!out) std::memcpy(use, q, nb);
        return true;
    }
    void drain() const { for (hipStream_t *s : streams) if (s && *s) (void)hipStreamSynchronize(*s); }
    // success path, after the entry has drained its streams: a staged result reaches the caller's array
    void commit() { if (staged && is_out && use) { drain(); std::memcpy(user, use, bytes); } }
    ~HostSpan()
    {
        if (!registered && !staged) return;
        drain();                                             // never unlock / free a buffer a copy still uses
        if (registered) (void)hipHostUnregister(user);
        if (staged && use) (void)hipHostFree(use);
    }
};

struct Timer {
    hipEvent_t a = nullptr, b = nullptr;
    Timer() { (void)hipEventCreate(&a); (void)hipEventCreate(&b); }
    ~Timer() { if (a) (void)hipEventDestroy(a); if (b) (void)hipEventDestroy(b); }
    void start(hipStream_t s) { (void)hipEventRecord(a, s); }
    double stop(hipStream_t s) { (void)hipEventRecord(b, s); (void)hipEventSynchronize(b); float ms = 0; (void)hipEventElapsedTime(&ms, a, b); return ms; }
};

constexpr int kCatCols = 4 + BFGX_MAX_EXTRA + 2;       // M, z, ra, dec, extra[2], ln1pz, lnM

int upload_catalog(bfgx_plan *p, const bfgx_catalog *h, DevBuf cols[kCatCols], bfgx_catalog *d, std::vector<double> &hostlog)
{
    const int nex = p->model.tab.ndim - 3;
    // the (z, M) table coordinates travel as columns: the caller's numpy values when given, else libm on the host here
    // (never the device's log: README.md:78-80 puts table edges exactly on the catalog's min/max)
    const double *lnz = h->ln1pz, *lnm = h->lnM;
    if (h->n > 0 && (!lnz || !lnm)) {
        if (!h->M || !h->z) return fail(BFGX_ERR_INVALID, "catalog column pointer is NULL");
        hostlog.resize(2 * (size_t)h->n);
        for (int64_t i = 0; i < h->n; ++i) {
            const double a = 1.0 / (1.0 + h->z[i]);                          // HealpixRunner.py:295
            hostlog[i] = std::log(1.0 / a);                                  // BaryonCorrection.py:364
            hostlog[(size_t)h->n + i] = std::log(h->M[i]);                   // :369
        }
        if (!lnz) lnz = hostlog.data();
        if (!lnm) lnm = hostlog.data() + h->n;
    }
    const double *src[kCatCols];
    src[0] = h->M; src[1] = h->z; src[2] = h->ra; src[3] = h->dec;
    for (int k = 0; k < BFGX_MAX_EXTRA; ++k) src[4 + k] = k < nex ? h->extra[k] : nullptr;
    src[4 + BFGX_MAX_EXTRA] = lnz; src[5 + BFGX_MAX_EXTRA] = lnm;
    for (int i = 0; i < kCatCols; ++i) {
        const bool wanted = i < 4 + nex || i >= 4 + BFGX_MAX_EXTRA;
        if (!wanted) continue;
        if (h->n > 0 && !src[i]) return fail(BFGX_ERR_INVALID, "catalog column pointer is NULL");
        if (cols[i].alloc(sizeof(double) * (size_t)h->n)) return fail(BFGX_ERR_HIP, "hipMalloc(catalog) failed");
        if (h->n > 0) HIP_TRY(hipMemcpyAsync(cols[i].p, src[i], sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, p->stream));
    }
    std::memset(d, 0, sizeof(*d));
    d->n = h->n;
    d->M = (const double *)cols[0].p; d->z = (const double *)cols[1].p;
    d->ra = (const double *)cols[2].p; d->dec = (const double *)cols[3].p;
    for (int k = 0; k < nex; ++k) d->extra[k] = (const double *)cols[4 + k].p;
    d->ln1pz = (const double *)cols[4 + BFGX_MAX_EXTRA].p; d->lnM = (const double *)cols[5 + BFGX_MAX_EXTRA].p;
    return BFGX_OK;
}

}  // namespace

// ---- plan cache of the one-shot API: a process() call re-uses the plan (model on the device, tiling, binning workspace) and
// the device buffers of the previous call with the same model / nside / device, so that a warm call performs no hipMalloc
namespace {

struct PoolBuf {
    void *p = nullptr; size_t cap = 0;
    int need(size_t bytes)
    {
        if (bytes <= cap) return 0;
        if (p) (void)hipFree(p);
        p = nullptr; cap = 0;
        const size_t want = bytes + bytes / 4 + 256;
        if (hipMalloc(&p, want) != hipSuccess) { p = nullptr; return 1; }
        cap = want;
        return 0;
    }
    void release() { if (p) (void)hipFree(p); p = nullptr; cap = 0; }
};

struct CacheEntry {
    uint64_t key = 0, key2 = 0, stamp = 0;     // two independent 64-bit hashes of (device, nside, model contents) + table size
    int64_t table_values = 0;
    bfgx_plan *plan = nullptr;
    PoolBuf cols[kCatCols], in, out, off, sums;
    hipStream_t copy_stream = nullptr;       // the map travels to the device while K0 / K1 run on the plan's stream
    hipStream_t out_stream = nullptr;        // the regridded band ranges travel back while the rest of the map is still arriving
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    std::vector<hipEvent_t> ev_in, ev_k2;    // per band range: its part of the map has arrived / has been regridded
    // the catalog columns of the last call stay on the device under the caller's token (bfgx_opts.catalog_token)
    uint64_t cat_token = 0;
    int64_t cat_n = -1;
    bfgx_catalog cat_dev;
};

std::mutex g_cache_mu;
std::vector<CacheEntry *> g_cache;
uint64_t g_cache_stamp = 0;
constexpr size_t kCacheMax = 4;

uint64_t g_hash_mul = 0x100000001b3ull;       // multiplier of hash_bytes (set by model_key: two passes with different constants)

uint64_t hash_bytes(uint64_t h, const void *data, size_t bytes)
{
    const unsigned char *c = (const unsigned char *)data;
    const uint64_t mul = g_hash_mul;
    size_t i = 0;
    for (; i + 8 <= bytes; i += 8) { uint64_t w; std::memcpy(&w, c + i, 8); h = (h ^ w) * mul; h ^= h >> 29; }
    for (; i < bytes; ++i) h = (h ^ c[i]) * mul;
    return h;
}

// which = 0: FNV-style offset basis and prime; which = 1: a second, independent pass (other basis, other odd multiplier) -- a cache hit
// needs both to agree (and the table size), so that one 64-bit collision cannot hand a model another model's device table
uint64_t model_key(int device, int64_t nside, const bfgx_model *m, int which = 0)
{
    g_hash_mul = which ? 0x9e3779b97f4a7c15ull : 0x100000001b3ull;
    struct Restore { ~Restore() { g_hash_mul = 0x100000001b3ull; } } restore;
    uint64_t h = which ? 0x2545f4914f6cdd1dull : 0xcbf29ce484222325ull;
    h = hash_bytes(h, &device, sizeof(device));
    h = hash_bytes(h, &nside, sizeof(nside));
    const bfgx_table &t = m->table;
    h = hash_bytes(h, &t.ndim, sizeof(t.ndim));
    h = hash_bytes(h, t.n, sizeof(t.n));
    size_t nv = 1;
    for (int d = 0; d < t.ndim; ++d) { h = hash_bytes(h, t.axis[d], sizeof(double) * (size_t)t.n[d]); nv *= (size_t)t.n[d]; }
    h = hash_bytes(h, t.values, sizeof(double) * nv);
    h = hash_bytes(h, &t.rdelta_sampling, sizeof(int32_t) * 2);
    h = hash_bytes(h, &t.eps_model, sizeof(double));
    h = hash_bytes(h, &m->cosmo_runner, sizeof(bfgx_cosmo));
    h = hash_bytes(h, &m->cosmo_model, sizeof(bfgx_cosmo));
    h = hash_bytes(h, &m->massdef_runner.Delta, sizeof(double)); h = hash_bytes(h, &m->massdef_runner.rho_type, sizeof(int32_t));
    h = hash_bytes(h, &m->massdef_model.Delta, sizeof(double)); h = hash_bytes(h, &m->massdef_model.rho_type, sizeof(int32_t));
    h = hash_bytes(h, &m->eps_runner, sizeof(double));
    return h;
}

void cache_drop(CacheEntry *e)
{
    if (e->plan) { (void)hipSetDevice(e->plan->device); bfgx_plan_destroy(e->plan); }
    for (auto &c : e->cols) c.release();
    e->in.release(); e->out.release(); e->off.release(); e->sums.release();
    if (e->copy_stream) (void)hipStreamDestroy(e->copy_stream);
    if (e->out_stream) (void)hipStreamDestroy(e->out_stream);
    for (auto &v : e->ev) if (v) (void)hipEventDestroy(v);
    for (auto &v : e->ev_in) (void)hipEventDestroy(v);
    for (auto &v : e->ev_k2) (void)hipEventDestroy(v);
    delete e;
}

// the cached entry for (device, nside, model), its plan large enough for n halos; g_cache_mu is held by the caller
int cache_acquire(int device, int64_t nside, const bfgx_model *model, int64_t n, CacheEntry **out)
{
    if (int rc = validate_model(model)) return rc;           // before hashing: the table pointers must be readable
    const uint64_t key = model_key(device, nside, model), key2 = model_key(device, nside, model, 1);
    int64_t nvals = 1;
    for (int d = 0; d < model->table.ndim; ++d) nvals *= model->table.n[d];
    CacheEntry *e = nullptr;
    for (CacheEntry *c : g_cache) if (c->key == key && c->key2 == key2 && c->table_values == nvals) e = c;
    if (e && e->plan->max_halos < n) {                       // grew: rebuild the plan (its workspace scales with max_halos)
        (void)hipSetDevice(device);
        bfgx_plan_destroy(e->plan);
        e->plan = nullptr;
    }
    if (!e) {
        if (g_cache.size() >= kCacheMax) {                   // evict the least recently used entry
            size_t lru = 0;
            for (size_t i = 1; i < g_cache.size(); ++i) if (g_cache[i]->stamp < g_cache[lru]->stamp) lru = i;
            cache_drop(g_cache[lru]);
            g_cache.erase(g_cache.begin() + (long)lru);
        }
        e = new CacheEntry();
        e->key = key; e->key2 = key2; e->table_values = nvals;
        g_cache.push_back(e);
    }
    if (!e->plan) {
        const int64_t cap = std::max<int64_t>(n + n / 4, 1024);
        if (int rc = bfgx_plan_create(device, nullptr, nside, cap, model, &e->plan)) {
            e->plan = nullptr;
            g_cache.erase(std::find(g_cache.begin(), g_cache.end(), e));
            cache_drop(e);
            return rc;
        }
        e->plan->blocking_growth = true;
    }
    e->stamp = ++g_cache_stamp;
    *out = e;
    return BFGX_OK;
}

std::atomic<long long> g_catalog_uploads{0};

int upload_catalog_pooled(CacheEntry *e, const bfgx_catalog *h, bfgx_catalog *d, std::vector<double> &hostlog, uint64_t token)
{
    bfgx_plan *p = e->plan;
    if (token != 0 && token == e->cat_token && h->n == e->cat_n) {      // the same catalog as last time: its columns are still there
        *d = e->cat_dev;
        return BFGX_OK;
    }
    e->cat_token = 0;
    g_catalog_uploads.fetch_add(1);
    const int nex = p->model.tab.ndim - 3;
    const double *lnz = h->ln1pz, *lnm = h->lnM;
    if (h->n > 0 && (!lnz || !lnm)) {                        // see upload_catalog
        if (!h->M || !h->z) return fail(BFGX_ERR_INVALID, "catalog column pointer is NULL");
        hostlog.resize(2 * (size_t)h->n);
        for (int64_t i = 0; i < h->n; ++i) {
            const double a = 1.0 / (1.0 + h->z[i]);
            hostlog[i] = std::log(1.0 / a);
            hostlog[(size_t)h->n + i] = std::log(h->M[i]);
        }
        if (!lnz) lnz = hostlog.data();
        if (!lnm) lnm = hostlog.data() + h->n;
    }
    const double *src[kCatCols];
    src[0] = h->M; src[1] = h->z; src[2] = h->ra; src[3] = h->dec;
    for (int k = 0; k < BFGX_MAX_EXTRA; ++k) src[4 + k] = k < nex ? h->extra[k] : nullptr;
    src[4 + BFGX_MAX_EXTRA] = lnz; src[5 + BFGX_MAX_EXTRA] = lnm;
    const double *dp[kCatCols] = {};
    for (int i = 0; i < kCatCols; ++i) {
        const bool wanted = i < 4 + nex || i >= 4 + BFGX_MAX_EXTRA;
        if (!wanted) continue;
        if (h->n > 0 && !src[i]) return fail(BFGX_ERR_INVALID, "catalog column pointer is NULL");
        if (e->cols[i].need(sizeof(double) * (size_t)std::max<int64_t>(h->n, 1))) return fail(BFGX_ERR_HIP, "hipMalloc(catalog) failed");
        if (h->n > 0) HIP_TRY(hipMemcpyAsync(e->cols[i].p, src[i], sizeof(double) * (size_t)h->n, hipMemcpyHostToDevice, p->stream));
        dp[i] = (const double *)e->cols[i].p;
    }
    std::memset(d, 0, sizeof(*d));
    d->n = h->n;
    d->M = dp[0]; d->z = dp[1]; d->ra = dp[2]; d->dec = dp[3];
    for (int k = 0; k < nex; ++k) d->extra[k] = dp[4 + k];
    d->ln1pz = dp[4 + BFGX_MAX_EXTRA]; d->lnM = dp[5 + BFGX_MAX_EXTRA];
    // (the copies above are from pageable memory: they have left the host buffers when hipMemcpyAsync returns)
    e->cat_token = token; e->cat_n = h->n; e->cat_dev = *d;
    return BFGX_OK;
}

}  // namespace

long long bfgx_debug_catalog_uploads(void) { return g_catalog_uploads.load(); }

void bfgx_debug_host_spans(long long *pinned_in_place, long long *staged, long long *smallest_pinned_bytes)
{
    if (pinned_in_place) *pinned_in_place = g_host_pinned_in_place.load();
    if (staged) *staged = g_host_staged.load();
    if (smallest_pinned_bytes) *smallest_pinned_bytes = g_host_pin_min_bytes.load();
}

void bfgx_cache_clear(void)
{
    {
        std::lock_guard<std::mutex> lk(g_cache_mu);
        for (CacheEntry *e : g_cache) cache_drop(e);
        g_cache.clear();
    }
    dep_release_all();           // workspace of the tiled particle deposit (bfgx_grid_api.inc)
    fft_release_all();           // twiddle / wavenumber tables of the power spectrum
    gcache_release_all();        // plans + device maps of the one-shot grid entries
    scache_release_all();        // plan + device record buffer of the snapshot records entry
    deprec_release_all();        // device buffers of the deposit's records entry
}

long long bfgx_debug_alloc_count(void) { return (long long)g_bfgx_allocs.load(); }

int bfgx_host_alloc(size_t bytes, void **out)
{
    if (!out) return fail(BFGX_ERR_INVALID, "NULL argument");
    *out = nullptr;
    if (bfgx_device_count() <= 0) return fail(BFGX_ERR_NO_DEVICE, "no HIP device visible");
    HIP_TRY(hipHostMalloc(out, bytes ? bytes : 1, hipHostMallocDefault));
    return BFGX_OK;
}

void bfgx_host_free(void *p) { if (p) (void)hipHostFree(p); }

int bfgx_baryonify_shell(const bfgx_catalog *cat, const bfgx_model *model, int64_t nside,
                         const double *map_in, double *map_out, const bfgx_opts *opts, bfgx_stats *stats)
{
    if (!cat || !model || !map_in || !map_out) return fail(BFGX_ERR_INVALID, "NULL argument");
    bfgx_opts o;
    std::memset(&o, 0, sizeof(o));
    o.check_mass = 1;
    o.algo = 1;
    o.acc_offsets_f64 = BFGX_ACC_AUTO;
    if (opts) o = *opts;
    if (o.algo != 0 && o.algo != 1) return fail(BFGX_ERR_INVALID, "opts.algo must be 0 or 1");
    if (cat->n < 0) return fail(BFGX_ERR_INVALID, "catalog size < 0");
    std::lock_guard<std::mutex> lk(g_cache_mu);              // one one-shot call at a time (they share the cached plans)
    CacheEntry *e = nullptr;
    if (int rc = cache_acquire(o.device, nside, model, cat->n, &e)) return rc;
    bfgx_plan *p = e->plan;
    p->algo = o.algo;
    HIP_TRY(hipSetDevice(p->device));
    DrainOnExit drain;
    drain.s[0] = &p->stream; drain.s[1] = &e->copy_stream; drain.s[2] = &e->out_stream; drain.null_stream = (p->stream == nullptr);

    const size_t npix = (size_t)p->hpx.npix;
    if (!acc_valid(o.acc_offsets_f64)) return fail(BFGX_ERR_INVALID, "opts.acc_offsets_f64 must be BFGX_ACC_AUTO (-1), 0 (f32), 1 (f64) or BFGX_ACC_PARITY (3)");
    const int acc = resolve_acc(p, o.acc_offsets_f64);       // (BFGX_ACC_AUTO: what the plan chose from the model's table)
    const bool off_split = (acc == BFGX_ACC_PARITY);         // pix_offsets as two fp32 arrays: hi [npix][3], then lo [npix][3]
    const size_t acc_bytes = npix * 3 * (acc != BFGX_ACC_F32 ? sizeof(double) : sizeof(float));
    std::vector<double> hostlog;
    bfgx_catalog dcat;
    if (!e->copy_stream) {
        HIP_TRY(hipStreamCreateWithFlags(&e->copy_stream, hipStreamNonBlocking));
        for (auto &v : e->ev) HIP_TRY(hipEventCreate(&v));
    }
    // phases: catalog -> device, K0 + K1 launched (enqueue-only), THEN the map -> device on a second stream (a copy from
    // pageable memory occupies the host, the kernels run meanwhile), K2 after both, map -> host
    HIP_TRY(hipEventRecord(e->ev[0], p->stream));
    if (int rc = upload_catalog_pooled(e, cat, &dcat, hostlog, o.catalog_token)) return rc;
    if (e->in.need(npix * sizeof(double)) || e->out.need(npix * sizeof(double)) || e->off.need(acc_bytes) || e->sums.need(40 * sizeof(double)))
        return fail(BFGX_ERR_HIP, "hipMalloc(map buffers) failed");
    if (o.algo == 0) {                                       // global-atomic kernels accumulate; the tiled ones store every element once
        HIP_TRY(hipMemsetAsync(e->off.p, 0, acc_bytes, p->stream));
        HIP_TRY(hipMemsetAsync(e->out.p, 0, npix * sizeof(double), p->stream));
    }
    p->omax_from_k1 = (o.algo == 1);                         // K1's flush leaves the largest |offset|^2 of every tile for K2's aprons
    const int rc_off = bfgx_offsets_device(p, &dcat, e->off.p, acc);
    p->omax_from_k1 = false;
    if (rc_off) return rc_off;
    double sums[2] = {0, 0};
    // Large maps travel in band ranges (contiguous RING pixel ranges): a range is regridded as soon as it and its apron have arrived
    // and goes back to the host while the next ones are still coming in, so that the two directions of the link work at the same
    // time.  The gathering regrid stores every pixel of a range exactly once; the few deposits it cannot gather (next to a pole) are
    // listed and added on the host.
    // Range size: copies below ~16 MB take a slower path here (8 ranges of 12.6 MB: 4.0 ms per call, 6 of 16.8 MB: 3.3 ms), so a map is cut
    // into ranges of at least that, at most 16 of them; both host buffers are page-locked for the duration of the call (hipHostRegister:
    // 2 us here) -- from pageable memory the copies in and out take turns on the host thread and the one-pass route below is faster
    // (4.3 against 4.5 ms), which is also where a map goes that cannot be registered.
    constexpr int kChunksMax = 16;
    int kChunks = (int)std::min<size_t>(kChunksMax, npix * sizeof(double) / ((size_t)16 << 20));
    if (const char *ce = std::getenv("BFGX_PIPE_CHUNKS")) kChunks = std::max(2, std::min(kChunksMax, std::atoi(ce)));       // (tests: small maps)
    bool piped = o.algo == 1 && kChunks >= 2 && p->tiling.nbands >= 2 * kChunks && !std::getenv("BFGX_NO_PIPELINE");
    bool whole_map_sent = false;
    HostSpan hin, hout;
    hin.streams[0] = hout.streams[0] = &e->copy_stream; hin.streams[1] = hout.streams[1] = &p->stream; hin.streams[2] = hout.streams[2] = &e->out_stream;
    const bool stage_small = std::getenv("BFGX_PIPE_CHUNKS") != nullptr;
    if (piped) piped = hin.open(map_in, npix * sizeof(double), false, stage_small) && hout.open(map_out, npix * sizeof(double), true, stage_small);
    const double *const user_map_in = map_in;
    double *const user_map_out = map_out;
    (void)user_map_in; (void)user_map_out;
    if (piped) { map_in = (const double *)hin.use; map_out = (double *)hout.use; }       // (from here on: the page-locked views)
    if (piped) {
        int nb = 0;
        std::vector<int64_t> bfp((size_t)p->tiling.nbands + 1);
        if (int rc = bfgx_plan_bands(p, &nb, bfp.data())) return rc;
        int cb[kChunksMax + 1];
        cb[0] = 0; cb[kChunks] = nb;
        for (int c = 1; c < kChunks; ++c) {
            const int64_t target = (int64_t)(npix * (size_t)c / kChunks);
            int b = (int)(std::lower_bound(bfp.begin(), bfp.end(), target) - bfp.begin());
            cb[c] = std::min(std::max(b, cb[c - 1] + 1), nb - (kChunks - c));
        }
        if (!e->out_stream) HIP_TRY(hipStreamCreateWithFlags(&e->out_stream, hipStreamNonBlocking));
        while ((int)e->ev_in.size() < kChunks) { hipEvent_t v; HIP_TRY(hipEventCreateWithFlags(&v, hipEventDisableTiming)); e->ev_in.push_back(v); }
        while ((int)e->ev_k2.size() < kChunks) { hipEvent_t v; HIP_TRY(hipEventCreateWithFlags(&v, hipEventDisableTiming)); e->ev_k2.push_back(v); }
        if (e->sums.need((2 * kChunks + 2) * sizeof(double))) return fail(BFGX_ERR_HIP, "hipMalloc(sums) failed");
        double *dsums = (double *)e->sums.p;
        float *domax = (float *)(dsums + 2 * kChunks);
        auto send = [&](int c) -> int {
            const int64_t lo = bfp[cb[c]], n = bfp[cb[c + 1]] - lo;
            HIP_TRY(hipMemcpyAsync((double *)e->in.p + lo, map_in + lo, (size_t)n * sizeof(double), hipMemcpyHostToDevice, e->copy_stream));
            HIP_TRY(hipEventRecord(e->ev_in[c], e->copy_stream));
            return BFGX_OK;
        };
        if (int rc = send(0)) return rc;
        if (int rc = send(1)) return rc;
        // how many rings a deposit travels: from the largest displacement K1 has seen (by now K1 has finished underneath the two copies)
        float m2 = 0.0f;
        if (int rc = bfgx_bands_max_offset2_device(p, 0, nb, domax)) return rc;
        HIP_TRY(hipMemcpyAsync(&m2, domax, sizeof(float), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        int32_t rings = kReachMax;
        if (std::isfinite(m2)) { if (int rc = bfgx_plan_reach_rings(p, std::sqrt((double)m2), &rings)) return rc; }
        const int32_t reach_before = p->band_reach;
        if (rings > 4) piped = false;                        // large displacements: the data-sized aprons and the in-stream repair of regrid_impl
        int sent = 2;
        if (piped) {
            p->band_reach = rings;
            int next = 0;
            int rc_run = BFGX_OK;
            auto run_ready = [&]() -> int {                  // regrid every range whose apron lies within the ranges sent so far
                while (next < kChunks) {
                    int64_t olo = 0, ohi = 0;
                    if (int rc = bfgx_plan_band_apron(p, cb[next], cb[next + 1], &olo, &ohi)) return rc;
                    int need = next;
                    while (need < kChunks - 1 && bfp[cb[need + 1]] < ohi) ++need;
                    if (need >= sent) break;
                    HIP_TRY(hipStreamWaitEvent(p->stream, e->ev_in[need], 0));
                    const int64_t lo = bfp[cb[next]], n = bfp[cb[next + 1]] - lo;
                    if (int rc = regrid_bands_impl(p, cb[next], cb[next + 1], (const double *)e->in.p, e->off.p, 0, (int64_t)npix, acc,
                                                   (double *)e->out.p + lo, dsums + 2 * next, next == 0,
                                                   off_split ? (const float *)e->off.p + 3 * npix : nullptr)) return rc;
                    HIP_TRY(hipEventRecord(e->ev_k2[next], p->stream));
                    HIP_TRY(hipStreamWaitEvent(e->out_stream, e->ev_k2[next], 0));
                    HIP_TRY(hipMemcpyAsync(map_out + lo, (double *)e->out.p + lo, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->out_stream));
                    ++next;
                }
                return BFGX_OK;
            };
            rc_run = run_ready();
            for (; rc_run == BFGX_OK && sent < kChunks; ) {
                rc_run = send(sent);
                ++sent;
                if (rc_run == BFGX_OK) rc_run = run_ready();
            }
            p->band_reach = reach_before;
            if (rc_run) { (void)hipStreamSynchronize(e->copy_stream); (void)hipStreamSynchronize(p->stream); (void)hipStreamSynchronize(e->out_stream); return rc_run; }
            HIP_TRY(hipEventRecord(e->ev[1], e->copy_stream));
            HIP_TRY(hipEventRecord(e->ev[2], p->stream));
            double hs[2 * kChunksMax];
            HIP_TRY(hipMemcpyAsync(hs, dsums, sizeof(double) * 2 * (size_t)kChunks, hipMemcpyDeviceToHost, p->stream));
            HIP_TRY(hipStreamSynchronize(p->stream));
            HIP_TRY(hipStreamSynchronize(e->out_stream));
            HIP_TRY(hipEventRecord(e->ev[3], p->stream));
            for (int c = 0; c < kChunks; ++c) { sums[0] += hs[2 * c]; sums[1] += hs[2 * c + 1]; }
            // the listed deposits (pixels next to a pole): added on the host; an overflowing list sends the call down the one-pass route
            int64_t nfar = 0;
            if (bfgx_plan_far_fetch(p, 0, nullptr, nullptr, &nfar) != BFGX_OK) piped = false;
            else if (nfar > 0) {
                std::vector<int64_t> fp((size_t)nfar);
                std::vector<double> fv((size_t)nfar);
                if (int rc = bfgx_plan_far_fetch(p, nfar, fp.data(), fv.data(), &nfar)) return rc;
                for (int64_t i = 0; i < nfar; ++i) if (fp[(size_t)i] >= 0 && fp[(size_t)i] < (int64_t)npix) map_out[fp[(size_t)i]] += fv[(size_t)i];
            }
            whole_map_sent = true;
        }
        if (!piped) {                                        // the rest of the map in one piece, then the one-pass route below
            if (!whole_map_sent && sent < kChunks) {
                const int64_t lo = bfp[cb[sent]];
                HIP_TRY(hipMemcpyAsync((double *)e->in.p + lo, map_in + lo, (npix - (size_t)lo) * sizeof(double), hipMemcpyHostToDevice, e->copy_stream));
            }
            whole_map_sent = true;
        }
    }
    if (!piped) {
        if (!whole_map_sent) HIP_TRY(hipMemcpyAsync(e->in.p, map_in, npix * sizeof(double), hipMemcpyHostToDevice, e->copy_stream));
        HIP_TRY(hipEventRecord(e->ev[1], e->copy_stream));
        HIP_TRY(hipStreamWaitEvent(p->stream, e->ev[1], 0));
        if (int rc = regrid_impl(p, (const double *)e->in.p, e->off.p, acc, (double *)e->out.p, (double *)e->sums.p, o.algo == 1)) return rc;
        HIP_TRY(hipEventRecord(e->ev[2], p->stream));
        HIP_TRY(hipMemcpyAsync(map_out, e->out.p, npix * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipMemcpyAsync(sums, e->sums.p, sizeof(sums), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipEventRecord(e->ev[3], p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    }
    float f_h2d = 0, f_k = 0, f_d2h = 0;
    (void)hipEventElapsedTime(&f_h2d, e->ev[0], e->ev[1]);   // both uploads (K0 + K1 run underneath the second one)
    (void)hipEventElapsedTime(&f_k, e->ev[1], e->ev[2]);     // what the kernels add after the last byte has arrived
    (void)hipEventElapsedTime(&f_d2h, e->ev[2], e->ev[3]);
    const double ms_h2d = f_h2d, ms_k = f_k, ms_d2h = f_d2h;
    if (int rc = bfgx_plan_status(p)) return rc;             // far-deposit list / entry list
    hout.commit();                                           // (a staged result reaches the caller's array)

    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->sum_in = sums[0]; stats->sum_out = sums[1];
        stats->ms_h2d = ms_h2d; stats->ms_kernels = ms_k; stats->ms_d2h = ms_d2h;
        stats->n_pairs = -1;
    }
    if (o.check_mass) {      // np.isclose(new_sum, old_sum): rtol 1e-5, atol 1e-8  (HealpixRunner.py:344-346)
        if (!(std::fabs(sums[1] - sums[0]) <= 1e-8 + 1e-5 * std::fabs(sums[0])))
            return fail(BFGX_ERR_MASS, "ERROR in pixel regridding, sum(new_map) [%0.14e] != sum(oldmap) [%0.14e]", sums[1], sums[0]);
    }
    return BFGX_OK;
}

int bfgx_paint_shell(const bfgx_catalog *cat, const bfgx_model *model, int64_t nside,
                     double *map_out, const bfgx_opts *opts, bfgx_stats *stats)
{
    if (!cat || !model || !map_out) return fail(BFGX_ERR_INVALID, "NULL argument");
    bfgx_opts o;
    std::memset(&o, 0, sizeof(o));
    o.acc_paint_f64 = 1;
    o.algo = 1;
    if (opts) o = *opts;
    if (o.algo != 0 && o.algo != 1) return fail(BFGX_ERR_INVALID, "opts.algo must be 0 or 1");
    if (cat->n < 0) return fail(BFGX_ERR_INVALID, "catalog size < 0");
    std::lock_guard<std::mutex> lk(g_cache_mu);
    CacheEntry *e = nullptr;
    if (int rc = cache_acquire(o.device, nside, model, cat->n, &e)) return rc;
    bfgx_plan *p = e->plan;
    p->algo = o.algo;
    HIP_TRY(hipSetDevice(p->device));
    DrainOnExit drain;
    drain.s[0] = &p->stream; drain.s[1] = &e->copy_stream; drain.s[2] = &e->out_stream; drain.null_stream = (p->stream == nullptr);

    const size_t npix = (size_t)p->hpx.npix;
    std::vector<double> hostlog;
    bfgx_catalog dcat;
    Timer t;
    t.start(p->stream);
    if (int rc = upload_catalog_pooled(e, cat, &dcat, hostlog, o.catalog_token)) return rc;
    if (e->out.need(npix * sizeof(double))) return fail(BFGX_ERR_HIP, "hipMalloc(map) failed");
    const double ms_h2d = t.stop(p->stream);
    t.start(p->stream);
    // Large fp64 maps are painted in band ranges, each copied back while the next ones are painted (as in bfgx_baryonify_shell: ranges of
    // at least 16 MB, at most 16, map_out page-locked for the call); K0 runs once, K3 on the tiles of each range
    constexpr int kChunksMax = 16;
    int kChunks = (int)std::min<size_t>(kChunksMax, npix * sizeof(double) / ((size_t)16 << 20));
    if (const char *ce = std::getenv("BFGX_PIPE_CHUNKS")) kChunks = std::max(2, std::min(kChunksMax, std::atoi(ce)));       // (tests: small maps)
    bool piped = o.algo == 1 && o.acc_paint_f64 != 0 && kChunks >= 2 && p->tiling.nbands >= 2 * kChunks && !std::getenv("BFGX_NO_PIPELINE");
    HostSpan hout;
    hout.streams[0] = &p->stream; hout.streams[1] = &e->out_stream;
    if (piped) piped = hout.open(map_out, npix * sizeof(double), true, std::getenv("BFGX_PIPE_CHUNKS") != nullptr);
    if (piped) map_out = (double *)hout.use;                 // (from here on: the page-locked view)
    double ms_k = 0.0;
    if (piped) {
        if (int rc = check_catalog(p, &dcat)) return rc;
        if (!p->model.tab.logv) return fail(BFGX_ERR_INVALID, "profile painting needs a table with log_values = 1");
        if (o.acc_paint_f64 < 0 || o.acc_paint_f64 > 2) return fail(BFGX_ERR_INVALID, "acc_f64 must be 0, 1 or 2");
        int nb = 0;
        std::vector<int64_t> bfp((size_t)p->tiling.nbands + 1);
        if (int rc = bfgx_plan_bands(p, &nb, bfp.data())) return rc;
        int cb[kChunksMax + 1];
        cb[0] = 0; cb[kChunks] = nb;
        for (int c = 1; c < kChunks; ++c) {
            const int64_t target = (int64_t)(npix * (size_t)c / kChunks);
            const int b = (int)(std::lower_bound(bfp.begin(), bfp.end(), target) - bfp.begin());
            cb[c] = std::min(std::max(b, cb[c - 1] + 1), nb - (kChunks - c));
        }
        if (!e->out_stream) HIP_TRY(hipStreamCreateWithFlags(&e->out_stream, hipStreamNonBlocking));
        while ((int)e->ev_k2.size() < kChunks) { hipEvent_t v; HIP_TRY(hipEventCreateWithFlags(&v, hipEventDisableTiming)); e->ev_k2.push_back(v); }
        const bool mixed = o.acc_paint_f64 == 2 && use_fast(p);      // (tables the fast kernel cannot take: fp64 throughout)
        if (int rc = launch_prep_and_bin(p, &dcat, 0, !mixed)) return rc;
        if (p->blocking_growth) if (int rc = ensure_entry_capacity(p, &dcat)) return rc;
        p->paint_pair_f32 = mixed;
        struct Reset { bfgx_plan *p; ~Reset() { p->paint_pair_f32 = false; p->k1_tile_lo = 0; p->k1_tile_n = -1; p->k1_spread_tiles = -1; } } reset{p};
        p->k1_spread_tiles = p->tiling.ntiles;           // K0 has binned the WHOLE catalog; every range launch sees the same density (ADVICE, round 4)
        for (int c = 0; c < kChunks; ++c) {
            p->k1_tile_lo = p->band_tile0_host[cb[c]];
            p->k1_tile_n = p->band_tile0_host[cb[c + 1]] - p->k1_tile_lo;
            // (the persistent kernel draws its tiles from a counter that the binning step zeroes: once more for every further launch)
            if (c > 0) HIP_TRY(hipMemsetAsync(p->tile_count + 5 * ((size_t)p->tiling.ntiles + 1), 0, sizeof(unsigned int), p->stream));
            if (int rc = launch_tile_scatter<MODE_PAINT, double>(p, (double *)e->out.p)) return rc;      // (indexed by global pixel number)
            HIP_TRY(hipEventRecord(e->ev_k2[c], p->stream));
            HIP_TRY(hipStreamWaitEvent(e->out_stream, e->ev_k2[c], 0));
            const int64_t lo = bfp[cb[c]], n = bfp[cb[c + 1]] - lo;
            HIP_TRY(hipMemcpyAsync(map_out + lo, (double *)e->out.p + lo, (size_t)n * sizeof(double), hipMemcpyDeviceToHost, e->out_stream));
        }
        ms_k = t.stop(p->stream);
        t.start(p->stream);
        HIP_TRY(hipStreamSynchronize(e->out_stream));
    } else {
    if (o.algo == 0) HIP_TRY(hipMemsetAsync(e->out.p, 0, npix * sizeof(double), p->stream));
    if (int rc = bfgx_paint_device(p, &dcat, e->out.p, o.acc_paint_f64)) return rc;
    ms_k = t.stop(p->stream);
    t.start(p->stream);
    if (o.acc_paint_f64) {
        HIP_TRY(hipMemcpyAsync(map_out, e->out.p, npix * sizeof(double), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
    } else {
        std::vector<float> tmp(npix);
        HIP_TRY(hipMemcpyAsync(tmp.data(), e->out.p, npix * sizeof(float), hipMemcpyDeviceToHost, p->stream));
        HIP_TRY(hipStreamSynchronize(p->stream));
        for (size_t i = 0; i < npix; ++i) map_out[i] = (double)tmp[i];
    }
    }
    const double ms_d2h = t.stop(p->stream);
    if (int rc = bfgx_plan_status(p)) return rc;
    hout.commit();
    if (stats) {
        std::memset(stats, 0, sizeof(*stats));
        stats->ms_h2d = ms_h2d; stats->ms_kernels = ms_k; stats->ms_d2h = ms_d2h;
        stats->n_pairs = -1;
    }
    return BFGX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------ table builders (a6-a8)
namespace {

struct DevArr {
    void *p = nullptr;
    ~DevArr() { if (p) (void)hipFree(p); }
    int up(const void *host, size_t bytes)
    {
        if (hipMalloc(&p, bytes ? bytes : 8) != hipSuccess) return 1;
        if (host && bytes && hipMemcpy(p, host, bytes, hipMemcpyHostToDevice) != hipSuccess) return 1;
        return 0;
    }
    template <typename T> T *as() { return (T *)p; }
};

int tables_begin(int device)
{
    if (bfgx_device_count() <= 0) return fail(BFGX_ERR_NO_DEVICE, "no HIP device visible: libbfgx has no CPU fallback");
    HIP_TRY(hipSetDevice(device));
    return BFGX_OK;
}

}  // namespace

extern "C" {

int bfgx_project_profile(int device, int64_t nrows, int32_t nl, const double *l, const double *rho,
                         int64_t nr, const double *r, double scale, double *sigma_out)
{
    if (!l || !rho || !r || !sigma_out) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (nrows < 1 || nrows > 65535 || nl < 2 || nl > 2048 || nr < 1) return fail(BFGX_ERR_INVALID, "bad sizes (1 <= nrows <= 65535, 2 <= nl <= 2048)");
    if (int rc = tables_begin(device)) return rc;
    DevArr dl, drho, dr, dout;
    if (dl.up(l, sizeof(double) * nl) || drho.up(rho, sizeof(double) * nrows * nl) || dr.up(r, sizeof(double) * nr) ||
        dout.up(nullptr, sizeof(double) * nrows * nr))
        return fail(BFGX_ERR_HIP, "device allocation/copy failed");
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)((nr + 255) / 256), (unsigned)nrows), dim3(256), 3 * sizeof(double) * nl, 0,
                       nl, dl.as<double>(), drho.as<double>(), nr, dr.as<double>(), scale, dout.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(sigma_out, dout.p, sizeof(double) * nrows * nr, hipMemcpyDeviceToHost));
    return BFGX_OK;
}

static int enclosed_mass_rows(int dim, int device, int64_t nrows, int64_t n_int, const double *r_int, const double *Sigma,
                              int32_t nr, const double *r, double *M_f)
{
    if (!r_int || !Sigma || !r || !M_f) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (nrows < 1 || n_int < 3 || nr < 1) return fail(BFGX_ERR_INVALID, "bad sizes");
    if (int rc = tables_begin(device)) return rc;
    DevArr dri, dS, dr, dcx, dcy, dM;
    if (dri.up(r_int, sizeof(double) * n_int) || dS.up(Sigma, sizeof(double) * nrows * n_int) || dr.up(r, sizeof(double) * nr) ||
        dcx.up(nullptr, sizeof(double) * nrows * n_int) || dcy.up(nullptr, sizeof(double) * nrows * n_int) ||
        dM.up(nullptr, sizeof(double) * nrows * nr))
        return fail(BFGX_ERR_HIP, "device allocation/copy failed");
    hipLaunchKernelGGL(enclosed_mass_kernel, dim3((unsigned)nrows), dim3(kTabThreads), 0, 0, n_int, dri.as<double>(),
                       dS.as<double>(), nr, dr.as<double>(), dcx.as<double>(), dcy.as<double>(), dM.as<double>(), dim);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(M_f, dM.p, sizeof(double) * nrows * nr, hipMemcpyDeviceToHost));
    return BFGX_OK;
}

int bfgx_enclosed_mass_from_sigma(int device, int64_t nrows, int64_t n_int, const double *r_int, const double *Sigma,
                                  int32_t nr, const double *r, double *M_f)
{
    return enclosed_mass_rows(2, device, nrows, n_int, r_int, Sigma, nr, r, M_f);
}

int bfgx_enclosed_mass_3d(int device, int64_t nrows, int64_t n_int, const double *r_int, const double *rho,
                          int32_t nr, const double *r, double *M_f)
{
    return enclosed_mass_rows(3, device, nrows, n_int, r_int, rho, nr, r, M_f);
}

int bfgx_enclosed_mass_2d(int device, int64_t nrows, int32_t nl, const double *l, const double *rho, double a,
                          int64_t n_int, const double *r_int, int32_t nr, const double *r, double *M_f)
{
    if (!l || !rho || !r_int || !r || !M_f) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (nrows < 1 || nrows > 65535 || nl < 2 || nl > 2048 || n_int < 3 || nr < 1) return fail(BFGX_ERR_INVALID, "bad sizes");
    if (int rc = tables_begin(device)) return rc;
    DevArr dl, drho, dri, dS, dr, dcx, dcy, dM;
    if (dl.up(l, sizeof(double) * nl) || drho.up(rho, sizeof(double) * nrows * nl) || dri.up(r_int, sizeof(double) * n_int) ||
        dS.up(nullptr, sizeof(double) * nrows * n_int) || dr.up(r, sizeof(double) * nr) ||
        dcx.up(nullptr, sizeof(double) * nrows * n_int) || dcy.up(nullptr, sizeof(double) * nrows * n_int) ||
        dM.up(nullptr, sizeof(double) * nrows * nr))
        return fail(BFGX_ERR_HIP, "device allocation/copy failed");
    // Sigma = model.projected(r_int) * a   (BaryonCorrection.py:646), then the prefix sum + log-log PCHIP
    hipLaunchKernelGGL(project_kernel, dim3((unsigned)((n_int + 255) / 256), (unsigned)nrows), dim3(256), 3 * sizeof(double) * nl, 0,
                       nl, dl.as<double>(), drho.as<double>(), n_int, dri.as<double>(), a, dS.as<double>());
    HIP_TRY(hipGetLastError());
    hipLaunchKernelGGL(enclosed_mass_kernel, dim3((unsigned)nrows), dim3(kTabThreads), 0, 0, n_int, dri.as<double>(),
                       dS.as<double>(), nr, dr.as<double>(), dcx.as<double>(), dcy.as<double>(), dM.as<double>(), 2);
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(M_f, dM.p, sizeof(double) * nrows * nr, hipMemcpyDeviceToHost));
    return BFGX_OK;
}

int bfgx_displacement_rows(int device, int64_t nrows, int32_t nr, const double *r, const double *M_dmo, const double *M_dmb,
                           double *d_out, int32_t *status)
{
    if (!r || !M_dmo || !M_dmb || !d_out || !status) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (nrows < 1 || nr < 3 || nr > kMaxNR) return fail(BFGX_ERR_INVALID, "N_samples_R must be in [3, %d]", kMaxNR);
    if (int rc = tables_begin(device)) return rc;
    DevArr dr, da, db, dd, ds;
    if (dr.up(r, sizeof(double) * nr) || da.up(M_dmo, sizeof(double) * nrows * nr) || db.up(M_dmb, sizeof(double) * nrows * nr) ||
        dd.up(nullptr, sizeof(double) * nrows * nr) || ds.up(nullptr, sizeof(int32_t) * nrows))
        return fail(BFGX_ERR_HIP, "device allocation/copy failed");
    const size_t lds = sizeof(double) * 7 * nr + sizeof(int) * 2 * nr;
    HIP_TRY(hipFuncSetAttribute((const void *)displacement_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipLaunchKernelGGL(displacement_kernel, dim3((unsigned)nrows), dim3(256), lds, 0, nr, dr.as<double>(), da.as<double>(),
                       db.as<double>(), dd.as<double>(), ds.as<int32_t>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(d_out, dd.p, sizeof(double) * nrows * nr, hipMemcpyDeviceToHost));
    HIP_TRY(hipMemcpy(status, ds.p, sizeof(int32_t) * nrows, hipMemcpyDeviceToHost));
    return BFGX_OK;
}

int bfgx_pressure_profile(int device, int64_t nrows, const double *r500, const double *rho_tot, const double *rho_gas,
                          int32_t nr_out, const double *r_out, double cutoff, double *P_out)
{
    if (!r500 || !rho_tot || !rho_gas || !r_out || !P_out) return fail(BFGX_ERR_INVALID, "NULL argument");
    if (nrows < 1 || nr_out < 1) return fail(BFGX_ERR_INVALID, "bad sizes");
    if (int rc = tables_begin(device)) return rc;
    DevArr dg, dt, dgas, dr, dP;
    if (dg.up(r500, sizeof(double) * kPressureN) || dt.up(rho_tot, sizeof(double) * nrows * kPressureN) ||
        dgas.up(rho_gas, sizeof(double) * nrows * kPressureN) || dr.up(r_out, sizeof(double) * nr_out) ||
        dP.up(nullptr, sizeof(double) * nrows * nr_out))
        return fail(BFGX_ERR_HIP, "device allocation/copy failed");
    const double G = kGnewt / (kMpcToMeter * kMpcToMeter * kMpcToMeter) * kSolarMass;       // Thermodynamic.py:11
    const double unit = (kSolarMass * 1e3) / (kMpcToMeter * 1e2);                           // :265
    hipLaunchKernelGGL(pressure_kernel, dim3((unsigned)nrows), dim3(256), 0, 0, dg.as<double>(), dt.as<double>(),
                       dgas.as<double>(), nr_out, dr.as<double>(), G, unit, cutoff, dP.as<double>());
    HIP_TRY(hipGetLastError());
    HIP_TRY(hipMemcpy(P_out, dP.p, sizeof(double) * nrows * nr_out, hipMemcpyDeviceToHost));
    return BFGX_OK;
}

}  // extern "C"

// ------------------------------------------------------------------------------ all GPUs of a node from one call
#include "bfgx_multi_api.inc"

// ------------------------------------------------------------------------------ pixel-window convolution (8f-3)
#include "bfgx_fftlog_api.inc"

// ------------------------------------------------------------------------------ regular-grid path (8f-1)
#include "bfgx_grid_api.inc"

// ------------------------------------------------------------------------------ particle snapshots (8f-2)
#include "bfgx_snapshot_api.inc"

// ------------------------------------------------------------------------------ models that are Python callables
#include "bfgx_callable_api.inc"
