/*
 * bfgx.h -- C ABI of libbfgx.so, the MI355X (gfx950) engine behind
 * baryonification_amd.Runners.{BaryonifyShell,PaintProfilesShell}.process().
 *
 * The reference (BaryonForge, pure Python) has no FFI layer; its de-facto operator
 * interface for this path is the runner/model protocol.  Each entry point below
 * names the reference code it replaces (paths relative to the reference root):
 *
 *   bfgx_baryonify_shell   <- BaryonForge/Runners/HealpixRunner.py:240-349  BaryonifyShell.process
 *   bfgx_paint_shell       <- BaryonForge/Runners/HealpixRunner.py:366-447  PaintProfilesShell.process
 *   bfgx_offsets_device    <- HealpixRunner.py:291-331 (halo loop) + BaryonCorrection.py:324-390 (_readout)
 *   bfgx_regrid_device     <- HealpixRunner.py:333-346 + regrid_pixels_hpix :13-67
 *   bfgx_paint_device      <- HealpixRunner.py:418-445 + utils/Tabulate.py:246-294, 569-621 (_readout)
 *   bfgx_cosmo_*           <- the pyccl calls on the path: HealpixRunner.py:268-280, :296 and
 *                             BaryonCorrection.py:370 (ccl.Cosmology background, angular_diameter_distance,
 *                             MassDef.get_radius)
 *
 * Conventions: plain pointers and sizes only; every function returns 0 on success or a
 * negative bfgx_status, with a thread-local message in bfgx_last_error().  "_host"
 * pointers are caller-owned host memory (numpy); "_dev" pointers are device memory on the
 * plan's device (e.g. a torch tensor's data_ptr()).  Nothing is retained after return
 * except inside an explicit bfgx_plan.  All calls are blocking w.r.t. the host unless noted
 * ("_device" entry points only enqueue on the plan's stream).
 */
#ifndef BFGX_H
#define BFGX_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define BFGX_ABI_VERSION 4

/* Precision of the displacement path (the `acc_f64` argument of the *_device entries, bfgx_opts.acc_offsets_f64).  The reference computes in
 * float64 throughout (HealpixRunner.py:289-341); SURVEY 8(d) states the tolerance a faster mode must hold: |d| <= 1e-6 mean(map) per pixel.
 *   BFGX_ACC_F32    fp32 pair math and fp32 pix_offsets [npix][3] (fp64 ring-row geometry and accumulation).  Holds 1e-6 mean(map) while a halo moves
 *                   a pixel by less than ~0.1 pixel sides (measured: 2.6e-7 / 3.9e-7 / 1.3e-6 / 8e-6 mean(map) at 0.03 / 0.09 / 0.3 / 0.9 pixels per halo, 2e-5 on a table that moves 9).
 *   BFGX_ACC_F64    fp64 throughout, pix_offsets double [npix][3]: the reference's own arithmetic, 1e-10 of the map scale.
 *   BFGX_ACC_PARITY the parity-grade mode: fp64 pair math whose elementary functions carry 1e-11 (fp32 hardware seeds + one Newton step), pix_offsets
 *                   as TWO fp32 arrays -- hi [npix][3], then lo [npix][3] = (float)(o - hi) behind it (24 bytes per pixel, as fp64) --, the regrid scans
 *                   the high halves and evaluates survivors in fp64 on hi + lo.  Within 1e-9 mean(map) of BFGX_ACC_F64 at any displacement.  Needs the fast
 *                   tile kernel (3-axis table, uniform ln r axis); elsewhere, and in the band-restricted entries, it runs as BFGX_ACC_F64.
 *   BFGX_ACC_AUTO   the plan chooses from its table at creation: BFGX_ACC_F32 while the table cannot move a pixel by more than 0.1 pixel sides of the
 *                   plan's NSIDE (largest |d| a / D_A over the table's (z, M) nodes inside the model-side cut), BFGX_ACC_PARITY beyond.  The default of
 *                   the one-shot host entries and of the Python runners.  Scratch for pix_offsets must then hold 24 bytes per pixel.
 * bfgx_plan_precision tells what a request resolves to and how far the table moves a pixel. */
#define BFGX_ACC_AUTO   (-1)
#define BFGX_ACC_F32    0
#define BFGX_ACC_F64    1
#define BFGX_ACC_PARITY 3
#define BFGX_MAX_EXTRA 4          /* extra (per-halo parameter) table axes, model.p_keys (Tabulate.py:524-561 takes any number; four = 64 corner rows
                                     per halo is what the generic tile kernel is instantiated for; the regular-grid runners take two) */
#define BFGX_MAX_DIM (3 + BFGX_MAX_EXTRA)

typedef enum bfgx_status {
    BFGX_OK = 0,
    BFGX_ERR_INVALID = -1,        /* bad argument (ValueError on the Python side) */
    BFGX_ERR_HIP = -2,            /* HIP runtime error (RuntimeError) */
    BFGX_ERR_NO_DEVICE = -3,      /* no GPU visible (RuntimeError; there is NO CPU fallback) */
    BFGX_ERR_UNSUPPORTED = -4,    /* NotImplementedError */
    BFGX_ERR_MASS = -5,           /* mass-conservation check failed (AssertionError, HealpixRunner.py:344-346) */
    BFGX_ERR_ASSERT = -6          /* another assert of the reference fired (AssertionError), e.g. Map2DRunner.py:516 */
} bfgx_status;

/* cosmology dict of io.py:79-85 plus the pyccl-2.x defaults the reference inherits */
typedef struct bfgx_cosmo {
    double Omega_m, Omega_b, h, sigma8, n_s, w0;
    double T_CMB;                 /* <= 0 -> 2.725 K  */
    double Neff;                  /* <  0 -> 3.046    */
} bfgx_cosmo;

/* ccl.halos.massdef.MassDef(Delta, rho_type) */
typedef struct bfgx_massdef {
    double  Delta;                /* e.g. 200 */
    int32_t rho_type;             /* 0 = 'critical', 1 = 'matter' */
    int32_t _pad;
} bfgx_massdef;

/* a tabulated model: raw_input_* of Baryonification2D/3D (BaryonCorrection.py:309-313) or of
 * TabulatedProfile / ParamTabulatedProfile (utils/Tabulate.py:231-235, 553-558) */
typedef struct bfgx_table {
    int32_t ndim;                         /* 3 + number of extra axes */
    int32_t n[BFGX_MAX_DIM];              /* points per axis */
    const double *axis[BFGX_MAX_DIM];     /* host: ln(1+z), ln M, ln r | ln(r/R_Delta), extra params */
    const double *values;                 /* host, C-order [n0][n1][n2]...; displacement [comoving Mpc],
                                             or ln(projected profile * a) when log_values = 1 */
    int32_t rdelta_sampling;              /* BaryonCorrection.py:374-379 */
    int32_t log_values;                   /* 1: read-out returns exp(interp) (Tabulate.py:285-286) */
    double  eps_model;                    /* model.epsilon_max (BaryonCorrection.py:381); unused for paint */
} bfgx_table;

/* HaloLightConeCatalog.cat columns (io.py:58-60): M [Msun], z, ra/dec [deg] */
typedef struct bfgx_catalog {
    int64_t n;
    const double *M, *z, *ra, *dec;
    const double *extra[BFGX_MAX_EXTRA];  /* cat[p_keys[k]], in table-axis order; NULL if unused */
    /* optional (NULL = derive on the device): the halo's table coordinates as the CALLER's numpy computes them, in exactly the
     * reference's form: `a = 1 / (1 + z); ln1pz = np.log(1 / a)` (NOT np.log(1 + z): the two differ in the last bit) and
     * `lnM = np.log(M)` (BaryonCorrection.py:364-366, Tabulate.py:279-281; the Python binding: _lib.table_coords).  README.md:78-80
     * builds tables whose edges are exactly the catalog's min/max, so whether an edge halo is inside the table hangs on the last bit
     * of these logs.  The one-shot host API fills them itself, in the same form (libm), when they are NULL. */
    const double *ln1pz, *lnM;
} bfgx_catalog;

typedef struct bfgx_model {
    bfgx_table   table;
    bfgx_cosmo   cosmo_runner;            /* catalog.cosmology -> R_j, D_j (HealpixRunner.py:268-297) */
    bfgx_massdef massdef_runner;          /* runner mass_def (HealpixRunner.py:150) */
    bfgx_cosmo   cosmo_model;             /* model.cosmo -> R in _readout (BaryonCorrection.py:370) */
    bfgx_massdef massdef_model;           /* model.mass_def */
    double       eps_runner;              /* runner epsilon_max (HealpixRunner.py:305) */
} bfgx_model;

typedef struct bfgx_opts {
    int32_t device;                       /* HIP device ordinal */
    int32_t acc_offsets_f64;              /* BaryonifyShell precision: BFGX_ACC_AUTO (-1; what a NULL opts selects), BFGX_ACC_F32 (0), BFGX_ACC_F64 (1) or
                                             BFGX_ACC_PARITY (3), see above */
    int32_t acc_paint_f64;                /* painted map: 0 = f32 throughout, 1 = f64 throughout (default for host API),
                                             2 = f32 pair math, f64 accumulation and f64 map (2.3x faster, ~1e-5 relative) */
    int32_t check_mass;                   /* 1: enforce np.isclose(sum(new), sum(old)) like the reference */
    int32_t algo;                         /* 1 = LDS tiles (default), 0 = per-halo global atomics */
    int32_t _pad;
    uint64_t catalog_token;               /* 0: the catalog columns are copied to the device on every call.  Non-zero: the caller's name for
                                             the CONTENT of the catalog (e.g. a hash of its bytes); the one-shot entries keep the columns of
                                             the last call on the device and skip the copy when the same token and size come again.  The
                                             caller vouches that equal tokens mean equal columns (HealpixRunner.py re-reads `cat` on every
                                             call: the Python runner hashes every byte of the catalog) */
} bfgx_opts;

typedef struct bfgx_stats {
    int64_t n_pairs;                      /* (halo, pixel) pairs visited */
    double  sum_in, sum_out;              /* map sums (baryonify only) */
    double  ms_h2d, ms_kernels, ms_d2h;   /* host-API phase timings */
} bfgx_stats;

/* kernel kinds reported by bfgx_plan_timing_read */
enum { BFGX_K_PREP = 0, BFGX_K_OFFSETS = 1, BFGX_K_REGRID = 2, BFGX_K_PAINT = 3, BFGX_K_SUM = 4, BFGX_K_COUNT = 5,
       BFGX_K_BIN = 6, BFGX_K_WIDE = 7, BFGX_NUM_KERNELS = 8 };
/* OFFSETS / PAINT: the fast tile kernel (narrow discs); WIDE: the generic tile kernel over the wide discs (polar caps, very low z) */
/* the grid plan reports its kernels under the same kinds: PREP, OFFSETS (halo loop), PAINT, REGRID, SUM */

typedef struct bfgx_plan bfgx_plan;       /* opaque: device, stream, resident model + workspace */

/* ---- library ---------------------------------------------------------------------------- */
int         bfgx_abi_version(void);
const char *bfgx_last_error(void);
int         bfgx_device_count(void);      /* 0 when no GPU is visible */

/* ---- host-side background cosmology (pure CPU, double) ---------------------------------- */
int bfgx_cosmo_E2(const bfgx_cosmo *c, int64_t n, const double *a, double *out);
int bfgx_cosmo_radius(const bfgx_cosmo *c, const bfgx_massdef *md, int64_t n,
                      const double *M, const double *a, double *out_phys_mpc);
int bfgx_cosmo_angular_diameter_distance(const bfgx_cosmo *c, int64_t n, const double *z, double *out_mpc);
/* D_a = CubicSpline(linspace(0,30,1000), D_A) of HealpixRunner.py:279-280 (not-a-knot):
 * knots[1000], coef[999][4] = {c3,c2,c1,c0} of ((c3 t + c2) t + c1) t + c0, t = z - knots[i] */
int bfgx_cosmo_da_spline(const bfgx_cosmo *c, double *knots, double *coef);
int bfgx_cosmo_da_eval(const bfgx_cosmo *c, int64_t n, const double *z, double *out_mpc);

/* ---- one-shot host API (numpy in, numpy out; copies H2D/D2H itself) --------------------- */
int bfgx_baryonify_shell(const bfgx_catalog *cat_host, const bfgx_model *model, int64_t nside,
                         const double *map_in_host, double *map_out_host,
                         const bfgx_opts *opts, bfgx_stats *stats);
int bfgx_paint_shell(const bfgx_catalog *cat_host, const bfgx_model *model, int64_t nside,
                     double *map_out_host, const bfgx_opts *opts, bfgx_stats *stats);

/* ---- all GPUs of one node from ONE call (SURVEY 8b "multi-GPU variant taking a device list") -----------------------
 * The counterpart of SplitJoinParallel (utils/Parallelize.py:191-320) for a binder that has nothing but this C ABI: the
 * catalog (already shuffled by the caller if wanted, Parallelize.py:255) is cut into ndev contiguous shards
 * (ceil(n / ndev) halos, :250-266) that go up to device devices[d]; the ring bands of the sphere are dealt out to the devices and
 * every halo's row (48 B) is pulled, peer to peer over xGMI, by the device(s) whose bands its disc touches; each device computes
 * the pixels it owns (no accumulator crosses a link), pulls the apron rings of pix_offsets the gathering regrid needs from its
 * neighbours, regrids the OUTPUT pixels of its bands and copies its slice straight into map_out_host.  Same arguments, results and
 * errors as the single-device calls; opts->device is ignored, opts->algo must be 1.  `devices` may repeat a device. */
int bfgx_baryonify_shell_multi(const bfgx_catalog *cat_host, const bfgx_model *model, int64_t nside, const double *map_in_host,
                               double *map_out_host, int32_t ndev, const int32_t *devices, const bfgx_opts *opts, bfgx_stats *stats);
int bfgx_paint_shell_multi(const bfgx_catalog *cat_host, const bfgx_model *model, int64_t nside, double *map_out_host,
                           int32_t ndev, const int32_t *devices, const bfgx_opts *opts, bfgx_stats *stats);

/* The one-shot calls keep the plan (model on the device, tiling, binning workspace) and their device buffers in a small
 * process-wide cache keyed by (device, nside, model contents): a repeated call with the same model performs no device
 * allocation.  bfgx_cache_clear frees the cache; bfgx_debug_alloc_count = device allocations made so far (tests).
 * bfgx_host_alloc / bfgx_host_free: page-locked host memory for map_out (D2H at full PCIe rate; plain memory works too). */
void      bfgx_cache_clear(void);
long long bfgx_debug_alloc_count(void);
/* catalog copies host -> device made by the one-shot entries so far (tests: a call that repeats bfgx_opts.catalog_token makes none) */
long long bfgx_debug_catalog_uploads(void);
/* how the one-shot host entries have treated the caller's arrays so far (tests): arrays page-locked IN PLACE for a call (hipHostRegister; only
 * arrays of >= 32 MiB, which own their pages -- DESIGN.md section 9 "the two GPU memory faults of round 4"), arrays staged through a page-locked
 * buffer of the library's, and the smallest array ever page-locked in place (-1: none).  Any pointer may be NULL. */
void bfgx_debug_host_spans(long long *pinned_in_place, long long *staged, long long *smallest_pinned_bytes);
/* streamed bfgx_baryonify_grid calls so far that found a cell moving further along the first array axis than their plane ranges allow (2^S - 1 cells:
 * 7 in 3-D, 15 in 2-D) and repeated the regrid in one pass on the uploaded map (tests) */
long long bfgx_debug_grid_pipe_fallbacks(void);
int       bfgx_host_alloc(size_t bytes, void **out);
void      bfgx_host_free(void *p);

/* ---- resident API (inputs already in HBM; enqueue-only on `hip_stream`) ------------------
 * hip_stream is a hipStream_t; NULL = the legacy default stream (torch's default stream). */
int  bfgx_plan_create(int device, void *hip_stream, int64_t nside, int64_t max_halos,
                      const bfgx_model *model, bfgx_plan **out);
void bfgx_plan_destroy(bfgx_plan *p);
/* accumulation algorithm of K1/K3: 1 (default) = tile-owned LDS accumulators, plain stores, every output
 * element written exactly once (no zero-fill needed); 0 = one wave per halo with global float atomics
 * (output must be zeroed by the caller) */
int  bfgx_plan_set_algo(bfgx_plan *p, int algo);
/* blocking health check of the plan's workspace (entry-list capacity); call outside timed regions */
int  bfgx_plan_status(bfgx_plan *p);
/* what `acc_requested` (BFGX_ACC_*) resolves to on this plan, and the largest displacement of the plan's table in pixel sides of its NSIDE
 * (either pointer may be NULL); HealpixRunner.py has no counterpart: the reference knows one precision */
int  bfgx_plan_precision(bfgx_plan *p, int acc_requested, int32_t *acc_resolved, double *table_disp_pixels);
/* K0 + K1: pix_offsets[npix][3] = sum of the per-halo unit-vector offsets (HealpixRunner.py:291-331); acc_f64 = BFGX_ACC_* selects precision and
 * layout (f32 / f64 [npix][3], or the split hi / lo arrays of BFGX_ACC_PARITY: 24 bytes per pixel) */
int  bfgx_offsets_device(bfgx_plan *p, const bfgx_catalog *cat_dev, void *offsets_dev, int acc_f64);
/* K2: map_out[npix] (f64) = bilinear regrid of the displaced pixels.  algo 1: every pixel of map_out is stored exactly
 * once by the tile that owns it (no zero-fill needed); algo 0: map_out += (must be zeroed by the caller).
 * sums_dev (optional, double[2]) receives {sum(map_in), sum(map_out)} */
int  bfgx_regrid_device(bfgx_plan *p, const double *map_in_dev, const void *offsets_dev, int acc_f64,
                        double *map_out_dev, double *sums_dev);
/* K0 + K1 + K2 in one enqueue-only call (BaryonifyShell.process() on device buffers): offsets_work_dev[npix][3] is the
 * pix_offsets scratch (12 bytes per pixel for BFGX_ACC_F32, 24 otherwise; every element overwritten), the other arguments as above.  Same results as
 * bfgx_offsets_device followed by bfgx_regrid_device; K1 hands the regrid the largest displacement of every tile, which
 * the separate calls have to find with one more pass over pix_offsets. */
int  bfgx_baryonify_device(bfgx_plan *p, const bfgx_catalog *cat_dev, const double *map_in_dev, void *offsets_work_dev,
                           int acc_f64, double *map_out_dev, double *sums_dev);
/* Multi-GPU form of K2: the sphere is cut into bands of consecutive rings (contiguous RING pixel ranges); a rank that
 * owns bands [band0, band1) produces exactly THEIR output pixels.  bfgx_plan_bands returns the number of bands and, in
 * band_first_pixel[nbands + 1] (may be NULL), the first pixel of every band (+ npix).  The gathering regrid evaluates
 * the displaced position of the rank's own pixels and of `rings` rings either side, where `rings` follows the largest
 * displacement of the SUMMED pix_offsets over all ranks: bfgx_plan_reach_rings(max |offset| in radians) -> rings (1 for
 * sub-pixel displacements, at most 16), bfgx_plan_set_band_reach(rings) -- every rank must set the same value (default
 * 1).  bfgx_plan_band_apron then returns the pixel range [olo, ohi) of summed pix_offsets the rank must hold
 * (offsets_dev points at pixel olo, [ohi - olo][3]; a wider range is accepted).
 * out_slice_dev points at the rank's first own pixel ([p1 - p0] values, every one stored exactly once, no zero-fill);
 * map_in_dev is the full map; sums_dev (optional, double[2]) receives {sum of the rank's source pixels, sum of its
 * deposits, the listed far ones included}.  Deposits that need the generic route (pixels next to a pole, displacements
 * beyond what `rings` rings of apron serve) are NOT applied by this call: they are listed with global pixel numbers;
 * bfgx_plan_far_fetch copies the list of the last regrid to the host (blocking; n_host = 0 almost always; NULL
 * buffers: count only) so that the caller can add them to whichever rank's slice holds the pixel.  (bfgx_regrid_device
 * sizes its aprons tile by tile from the data, applies its own list, and repairs an overflowing list in-stream.) */
int  bfgx_plan_bands(bfgx_plan *p, int32_t *nbands, int64_t *band_first_pixel);
/* the same maximum for the slice bfgx_offsets_bands_device has JUST written for the bands [band0, band1): reduced from the per-tile
 * maxima K1's flush leaves behind (a few thousand values) instead of a second pass over the slice.  Valid only while nothing
 * else has been added to that slice (spatial sharding: every rank computes its own pixels). */
int  bfgx_bands_max_offset2_device(bfgx_plan *p, int32_t band0, int32_t band1, float *out_dev);
/* *out_dev (float, device) = largest |offset|^2 of npixels pixels of pix_offsets (enqueue-only): what the ranks all-reduce (MAX) */
int  bfgx_max_offset2_device(bfgx_plan *p, const void *offsets_dev, int64_t npixels, int acc_f64, float *out_dev);
int  bfgx_plan_reach_rings(bfgx_plan *p, double max_offset, int32_t *rings);
int  bfgx_plan_set_band_reach(bfgx_plan *p, int32_t rings);
int  bfgx_plan_band_apron(bfgx_plan *p, int32_t band0, int32_t band1, int64_t *olo, int64_t *ohi);
int  bfgx_regrid_bands_device(bfgx_plan *p, int32_t band0, int32_t band1, const double *map_in_dev,
                              const void *offsets_dev, int64_t olo, int64_t ohi, int acc_f64, double *out_slice_dev, double *sums_dev);
/* A rank's share of one resident multi-GPU step in TWO calls (Parallelize.py:250-318: one call drives all workers).
 * bfgx_route_step_device: the routing -- per local halo the ring range its disc can touch (+ the plan's route margin) and its rows packed by
 * destination into fixed-capacity blocks [ncols][blockcap] (column 0 = M; rows nobody fills keep M = NaN, which K0 drops).  send_blocks_dev:
 * (world - 1) blocks in rank order WITHOUT this rank -- the input of ONE all_to_all_single whose split towards oneself is empty;
 * recv_blocks_dev: world blocks, the other ranks' in rank order first (where that all_to_all puts them), this rank's own rows LAST: they are
 * written there directly and never enter the collective.  *overflow_dev is set when a block was too small.  Two launches, nothing read back.
 * bfgx_plan_set_catalog_blocks(rows, stride): the catalogs of the following K0 launches are such blocks -- halo j's columns at
 * (j / rows) * stride + (j % rows) from the column pointers of block 0 (stride = ncols * blockcap); rows = 0: plain columns again.
 * bfgx_offsets_regrid_bands_device: K0 + binning + K1 for the bands [B0, B1) (the rank's own [b0, b1) and, with a route margin of one band, the
 * band either side: the aprons of its regrid, computed locally) into offsets_dev (pixels from the first pixel of band B0: [pixels][3] f32 or f64;
 * BFGX_ACC_PARITY -- or BFGX_ACC_AUTO on a table that asks for it -- keeps the parity-grade mode here, since nothing of pix_offsets leaves the
 * device: [pixels][3] f32 high halves followed by [pixels][3] f32 low halves, 24 bytes per pixel like f64), the banded regrid of [b0, b1) into out_slice_dev, the listed far deposits that fall into those
 * pixels added (the others counted into *foreign_dev, which the caller zeroes once), the two sums of the mass check.  One memset, eight launches. */
int  bfgx_route_step_device(bfgx_plan *p, const bfgx_catalog *cat_dev, int32_t world, int32_t rank, const int32_t *ring_bounds, int64_t blockcap,
                            int32_t ncols, const double *const *cols_dev, int32_t *cursor_dev, double *send_blocks_dev, double *recv_blocks_dev,
                            int32_t *overflow_dev);
int  bfgx_plan_set_catalog_blocks(bfgx_plan *p, int64_t rows, int64_t stride);
int  bfgx_offsets_regrid_bands_device(bfgx_plan *p, const bfgx_catalog *cat_dev, int32_t B0, int32_t B1, void *offsets_dev, int acc_f64,
                                      int32_t b0, int32_t b1, const double *map_in_dev, double *out_slice_dev, double *sums_dev,
                                      unsigned long long *foreign_dev);
int  bfgx_plan_far_fetch(bfgx_plan *p, int64_t cap, int64_t *pix_host, double *val_host, int64_t *n_host);
/* What the LAST full-map regrid (bfgx_regrid_device / bfgx_baryonify_device) did, read back from its control words (blocking;
 * diagnostics for tests and bench lines -- regrid_pixels_hpix, HealpixRunner.py:13-67, has no counterpart): deposits listed for the
 * generic route, whether that list overflowed (then repaired in-stream), tiles whose reach exceeded one ring (run by the walking
 * kernel), the largest reach of any tile in rings.  Any pointer may be NULL. */
int  bfgx_plan_regrid_stats(bfgx_plan *p, int64_t *far_listed, int32_t *far_overflowed, int32_t *tiles_walked, int32_t *max_reach_rings);
/* enqueue-only alternative: adds the listed deposits whose pixel lies in [p0, p1) to out_slice_dev (which starts at pixel
 * p0) and adds the number of the others to *foreign_dev (optional device counter the caller inspects later) */
int  bfgx_plan_far_apply_device(bfgx_plan *p, double *out_slice_dev, int64_t p0, int64_t p1, unsigned long long *foreign_dev);
/* Spatially sharded multi-GPU runs: a rank that owns the bands [band0, band1) (bfgx_plan_bands) takes every halo whose disc can
 * touch them -- bfgx_disc_rings_device gives, per halo, the ring range [first, last] (1-based, inclusive, widened by 2 rings;
 * first > last: nothing) -- and runs K0 + K1 (or K3) for ITS tiles only: offsets_slice_dev / map_slice_dev point at the first
 * pixel of band0 ([p1 - p0][3] resp. [p1 - p0] elements, every one stored exactly once); halos that reach into other bands are
 * clipped.  No accumulator travels between the ranks; the regrid then needs the neighbours' apron rings as above. */
int  bfgx_plan_tile_shape(bfgx_plan *p, int32_t *rings_per_band, int32_t *max_columns);     /* band b = rings [1 + b R, 1 + (b + 1) R) */
int  bfgx_disc_rings_device(bfgx_plan *p, const bfgx_catalog *cat_dev, int32_t *rings_dev /* [n][2]; n may exceed the plan's max_halos */);
/* rings by which bfgx_disc_rings_device widens every (non-empty) range from now on (default 0).  With a margin of one band
 * (bfgx_plan_tile_shape) a rank receives the halos of the bands next to its own as well and can compute the apron rows of its regrid
 * itself (bfgx_offsets_bands_device over [band0 - 1, band1 + 1)) instead of exchanging them with its neighbours: one collective less
 * per pass for ~2 bands of extra K1 work (valid while the reach, at most 16 rings, fits into a band). */
int  bfgx_plan_set_route_margin(bfgx_plan *p, int32_t rings);
/* Routing of halos that start out scattered over the ranks: ring_bounds[j] (host, world + 1 ascending entries) = first ring of rank
 * j's run of bands; a halo goes to every rank whose rings its range [first, last] (rings_dev, bfgx_disc_rings_device) touches.
 * Pass 1 counts the halos per destination (counts_dev[world]); the caller forms the exclusive prefix `start` (host) and exchanges the
 * counts; pass 2 packs the rows (ncols doubles per halo, from the ncols device columns cols_dev[]) into rows_dev, destination by
 * destination, ready for ONE all_to_all.  cursor_dev: int32[world] scratch.  At most 64 ranks, 8 columns. */
int  bfgx_route_count_device(bfgx_plan *p, int64_t n, const int32_t *rings_dev, int32_t world, const int32_t *ring_bounds, int32_t *counts_dev);
int  bfgx_route_fill_device(bfgx_plan *p, int64_t n, const int32_t *rings_dev, int32_t world, const int32_t *ring_bounds, const int64_t *start,
                            int32_t ncols, const double *const *cols_dev, int32_t *cursor_dev, double *rows_dev);
/* The same in ONE pass and with nothing read back by the host (a resident step: Parallelize.py:255-273 shuffles the catalog, so the counts
 * per destination are close to n / world): fixed-capacity blocks blocks_dev[world][ncols][blockcap], column-major inside a block, the
 * equal splits of ONE all_to_all.  Rows a destination does not receive keep M = NaN (column 0), which K0 drops as invalid halos;
 * *overflow_dev (int32, zeroed by the caller) is set when a destination would receive more than blockcap halos. */
int  bfgx_route_pack_device(bfgx_plan *p, int64_t n, const int32_t *rings_dev, int32_t world, const int32_t *ring_bounds, int64_t blockcap,
                            int32_t ncols, const double *const *cols_dev, int32_t *cursor_dev, double *blocks_dev, int32_t *overflow_dev);
int  bfgx_offsets_bands_device(bfgx_plan *p, const bfgx_catalog *cat_dev, int32_t band0, int32_t band1, void *offsets_slice_dev, int acc_f64);
int  bfgx_paint_bands_device(bfgx_plan *p, const bfgx_catalog *cat_dev, int32_t band0, int32_t band1, void *map_slice_dev, int acc_f64);
/* K0 + K3: map_out[npix] += painted profile; accumulator f32 or f64 */
/* acc_f64: 0 = float map, fp32 pair math; 1 = double map, fp64 throughout; 2 = double map and fp64 LDS accumulation with the pair
 * phase (chord, ln r, table read-out, exp) in fp32 -- the stated fp64 -> fp32 tolerance of this mode: 5e-5 of the pixel value */
int  bfgx_paint_device(bfgx_plan *p, const bfgx_catalog *cat_dev, void *map_out_dev, int acc_f64);
/* optional per-kernel timing with HIP events recorded on the plan's stream around every launch.
 * bfgx_plan_timing_read synchronises the stream, returns summed milliseconds and launch counts per
 * kernel kind (arrays of BFGX_NUM_KERNELS) since the previous read, and resets the counters. */
int  bfgx_plan_timing_enable(bfgx_plan *p, int on);
int  bfgx_plan_timing_read(bfgx_plan *p, double *ms_sum, int64_t *launches);

/* pair census (same enumeration as K1/K3, no scatter): counts_dev int64[n] or NULL; fallback4 = 1
 * counts the <4-pixel fallback of BaryonifyShell (HealpixRunner.py:309-310). Blocking. */
int  bfgx_count_pairs_device(bfgx_plan *p, const bfgx_catalog *cat_dev, int fallback4,
                             int64_t *counts_dev, int64_t *total_host);

/* ---- table builders: host arrays in, host arrays out, run once per model ----
 * The profile physics stays on the host: these take 3-D densities SAMPLED on the radial grids the
 * reference uses.  All double.  Replaces:
 *   bfgx_project_profile          SchneiderProfiles._projected_realspace, Profiles/Schneider19.py:245-252
 *                                 sigma[row][j] = scale * 2 trapz(interp(sqrt(l^2 + r_j^2), l, rho_row), l)
 *   bfgx_enclosed_mass_2d         Baryonification2D.get_masses, Profiles/BaryonCorrection.py:639-661
 *                                 (projection onto r_int times a, Sigma<0 -> 0, prefix sum, log-log PCHIP at r)
 *   bfgx_enclosed_mass_from_sigma the same from a projected profile the caller computed some other way
 *   bfgx_enclosed_mass_3d         Baryonification3D.get_masses, Profiles/BaryonCorrection.py:519-546: rho sampled on
 *                                 r_int = geomspace(min(r,1e-6)/1.2, max(r,1000)*1.2, 50 000), rho<0 -> 0,
 *                                 cumsum(4 pi r^3 rho dlnr), log-log PCHIP at r over the points with rho > 0
 *   bfgx_displacement_rows        setup_interpolator's per-mass loop body, BaryonCorrection.py:226-301;
 *                                 status[row]: 0 ok, 1 mass profile nearly constant (iterate > 30),
 *                                 2 fewer than 5 usable points -- both give d = 0 and warrant the reference's warning
 *   bfgx_pressure_profile         Pressure._real, Profiles/Thermodynamic.py:240-271 (cgs), r500 = geomspace(1e-6,1e3,500) */
int bfgx_project_profile(int device, int64_t nrows, int32_t nl, const double *l, const double *rho,
                         int64_t nr, const double *r, double scale, double *sigma_out);
int bfgx_enclosed_mass_2d(int device, int64_t nrows, int32_t nl, const double *l, const double *rho, double a,
                          int64_t n_int, const double *r_int, int32_t nr, const double *r, double *M_f);
int bfgx_enclosed_mass_from_sigma(int device, int64_t nrows, int64_t n_int, const double *r_int, const double *Sigma,
                                  int32_t nr, const double *r, double *M_f);
int bfgx_enclosed_mass_3d(int device, int64_t nrows, int64_t n_int, const double *r_int, const double *rho,
                          int32_t nr, const double *r, double *M_f);
int bfgx_displacement_rows(int device, int64_t nrows, int32_t nr, const double *r, const double *M_dmo,
                           const double *M_dmb, double *d_out, int32_t *status);
int bfgx_pressure_profile(int device, int64_t nrows, const double *r500, const double *rho_tot, const double *rho_gas,
                          int32_t nr_out, const double *r_out, double cutoff, double *P_out);

/* ---- pixel-window convolution ---------------------------------------------------
 * BaryonForge/utils/Pixel.py: ConvolvedProfile.real (:106-157) / .projected (:160-224) = FFTLog forward, x pixel window,
 * FFTLog back, PCHIP in ln r.  The transform is pyccl.pyutils._fftlog_transform(rs, frs, dim, mu, power_law_index)
 * (pyccl 2.8.0, a C port of Hamilton's FFTLog); pyccl is absent here, so what is implemented is the published algorithm
 * (Hamilton 2000, App. B) with the bias exponent q = dim/2 + power_law_index and the low-ringing k r nearest to 1:
 *   dim 3: F(k) = (2 pi)^-3 int d^3r f(r) j_mu(k r)-kernel;   dim 2: F(k) = (2 pi)^-2 int d^2r f(r) J_mu(k r).
 * Host arrays in / out; f and prof are [nrows][n] on the log grid r (positive, ascending, log-uniform). */
/* the output grid of one transform (pure CPU): k_j = (k r)_lowring / r[n-1-j] */
int bfgx_fftlog_kgrid(int32_t n, const double *r, int32_t dim, double mu, double plaw, double *k_out);
/* ks, fks = _fftlog_transform(rs, frs, dim, mu, plaw) */
int bfgx_fftlog_transform(int device, int64_t nrows, int32_t n, const double *r, const double *f, int32_t dim, double mu,
                          double plaw, double *k_out, double *F_out);
/* Pixel.py:146-155 / :208-222 in one call: transform(prof; plaw_fwd) * window -> transform(.; plaw_back) ->
 * PchipInterpolator(ln(r_out * r_scale), ., extrapolate=False)(ln r_eval), NaN -> 0, x (2 pi)^dim.
 * window[n] = Pixel.real / Pixel.projected evaluated by the caller at bfgx_fftlog_kgrid(r_fft, plaw_fwd); r_eval[nq] is
 * already clipped at pixel_size / 5 (:153, :217-219); r_scale = D_A for harmonic pixels (r_fft given in radians), else 1 */
int bfgx_fftlog_convolve(int device, int64_t nrows, int32_t n, const double *r_fft, const double *prof, int32_t dim, double mu,
                         double plaw_fwd, double plaw_back, const double *window, int32_t nq, const double *r_eval, double r_scale,
                         double *out);

/* ---- regular-grid path: periodic square / cubic maps ---------------------------------
 * Replaces BaryonForge/Runners/Map2DRunner.py: BaryonifyGrid.process :431-607, PaintProfilesGrid.process :676-817,
 * regrid_pixels_2D :14-83, regrid_pixels_3D :86-163; utils/io.py ParticleSnapshot.make_map :622-670; and the FFT
 * P(k) summary of examples/10_Reproduce_Schneider_deltaPk.ipynb cells 12, 15.
 * Maps are C-order [npix]^ndim doubles.  The grid runners build ccl.Cosmology WITHOUT w0 (Map2DRunner.py:456-459):
 * callers pass cosmo_runner with w0 = -1.  HaloNDCatalog stores float32 columns (io.py:205): the caller passes those
 * values widened to double, plus (optionally) lnM = the float32 logarithm the reference's read-out takes of the mass
 * (BaryonCorrection.py:369, Tabulate.py:283), evaluated with the caller's numpy so that its last bit agrees. */
typedef struct bfgx_grid {
    int32_t ndim;                         /* 2 or 3 */
    int32_t npix;                         /* pixels per side, >= 5 */
    const double *bins;                   /* [npix] pixel-centre coordinates, comoving Mpc, strictly ascending */
    double redshift;
} bfgx_grid;

typedef struct bfgx_grid_catalog {
    int64_t n;
    const double *M, *x, *y, *z;          /* z (a Cartesian coordinate) may be NULL for 2D maps */
    const double *lnM;                    /* optional, see above; NULL -> (double)logf((float)M) on the device */
    const double *rmat;                   /* optional [n][4]: row-major 2x2 matrices of DefaultRunnerGrid.build_Rmat
                                             (use_ellipticity = True, 2D maps only) */
    const double *extra[BFGX_MAX_EXTRA];
} bfgx_grid_catalog;

typedef struct bfgx_grid_plan bfgx_grid_plan;

/* one-shot host API.  bfgx_paint_grid reads the LOG of raw_input_2D (2D maps) / raw_input_3D (3D maps) as table. */
int bfgx_baryonify_grid(const bfgx_grid_catalog *cat_host, const bfgx_model *model, const bfgx_grid *grid,
                        const double *map_in_host, double *map_out_host, const bfgx_opts *opts, bfgx_stats *stats);
int bfgx_paint_grid(const bfgx_grid_catalog *cat_host, const bfgx_model *model, const bfgx_grid *grid,
                    double *map_out_host, const bfgx_opts *opts, bfgx_stats *stats);
/* regrid_pixels_2D / regrid_pixels_3D: grid[npix]^ndim += overlap-weighted values; positions [n][ndim] */
int bfgx_regrid_pixels(int device, int32_t ndim, int32_t npix, int64_t n, const double *positions_host,
                       const double *values_host, double *grid_inout_host);
/* ParticleSnapshot.make_map: histogramdd of n particles on edges = linspace(0, L, n_grid + 1) (edges [n_grid + 1]
 * given by the caller), weights = mass (NULL -> 1).  x, y, z host arrays; z NULL for 2D. */
int bfgx_deposit_particles(int device, int32_t ndim, int64_t n, const double *x, const double *y, const double *z,
                           const double *mass, int32_t n_grid, const double *edges, double *map_out_host);
/* The same on the snapshot's RECORDS (io.py:378-470: one structured array with float64 fields M, x, y, z): records [n][itemsize bytes] go
 * to the device as they are and the kernels read the fields at byte offsets off_x / off_y / off_z (ignored in 2-D) / off_mass (< 0: unit
 * masses) -- the four strided column gathers of make_map's host side (io.py:640-650) do not happen.  itemsize and the offsets must be
 * multiples of 8.  Device buffers are kept between calls (bfgx_cache_clear). */
int bfgx_deposit_particles_records(int device, int32_t ndim, int64_t n, const void *records, int32_t itemsize, int32_t off_x, int32_t off_y,
                                   int32_t off_z, int32_t off_mass, int32_t n_grid, const double *edges, double *map_out);
/* |FFT(map)|^2 averaged in nk linear k-bins between 2 pi / L and the Nyquist frequency (3D, n_grid a power of two):
 * pk[nk], kcen[nk] (mean |k| per bin), counts[nk] (modes per bin).  Empty bins give NaN as in the notebook. */
int bfgx_power_spectrum(int device, int32_t n_grid, const double *map_host, double L, int32_t nk,
                        double *pk, double *kcen, int64_t *counts);

/* resident API: inputs already in HBM; enqueue-only except for one small read-back of the work-item count */
int  bfgx_grid_plan_create(int device, void *hip_stream, const bfgx_grid *grid_host_bins, int64_t max_halos,
                           const bfgx_model *model, bfgx_grid_plan **out);
void bfgx_grid_plan_destroy(bfgx_grid_plan *p);
/* halo loop of BaryonifyGrid: offsets_dev [npix^ndim][ndim] f64 is zeroed and filled; n_pairs_host optional */
int  bfgx_grid_offsets_device(bfgx_grid_plan *p, const bfgx_grid_catalog *cat_dev, double *offsets_dev, int64_t *n_pairs_host);
/* halo loop of PaintProfilesGrid: map_out_dev [npix^ndim] f64 is zeroed and filled */
int  bfgx_grid_paint_device(bfgx_grid_plan *p, const bfgx_grid_catalog *cat_dev, double *map_out_dev, int64_t *n_pairs_host);
/* post-loop regrid; map_out_dev is zeroed by the call; sums_dev (optional, double[2], zeroed by the caller)
 * receives {sum(map_in), sum(map_out)} */
int  bfgx_grid_regrid_device(bfgx_grid_plan *p, const double *map_in_dev, const double *offsets_dev,
                             double *map_out_dev, double *sums_dev);
/* Slab decomposition over GPUs (BASELINE config 5): a plan may own the planes [plane_lo, plane_lo + plane_n) of the FIRST
 * array axis.  bfgx_grid_offsets_device / _paint_device then fill plane_n x npix (x npix) cells (every rank passes the
 * whole catalog; cutouts are clipped to the slab), and bfgx_grid_regrid_slab_device regrids the slab's source cells
 * (map_in, offsets: the slab) into map_out = the slab + `apron` planes either side ([plane_n + 2 apron] planes, zeroed by
 * the call, periodic): the caller adds the apron planes to the neighbours' slabs.  *missed_dev (optional, zeroed by the
 * caller) is set when a deposit fell outside the buffer (apron too small). */
int  bfgx_grid_plan_set_slab(bfgx_grid_plan *p, int32_t plane_lo, int32_t plane_n);
int  bfgx_grid_regrid_slab_device(bfgx_grid_plan *p, const double *map_in_dev, const double *offsets_dev, int32_t apron,
                                  double *map_out_dev, double *sums_dev, int32_t *missed_dev);
/* BaryonifyGrid.process() on device arrays, single GPU: the halo loop (Map2DRunner.py:476-575) and the regrid (:577-605) as ONE
 * cell-owned pass -- halos are listed per block of cells, every cell sums its offsets in registers and is regridded at once, so no
 * pix_offsets array exists.  map_out [npix^ndim] is zeroed by the call; sums_dev [2] (optional, zeroed by the caller) receives
 * {sum(map_in), sum(map_out)}.  Same result as bfgx_grid_offsets_device + bfgx_grid_regrid_device up to the order of fp64 sums. */
int  bfgx_grid_baryonify_device(bfgx_grid_plan *p, const bfgx_grid_catalog *cat_dev, const double *map_in_dev, double *map_out_dev,
                                double *sums_dev, int64_t *n_pairs_host);
/* ParticleSnapshot.make_map (io.py:622-670) + BaryonifyGrid.process (Map2DRunner.py:431-607) in one call, all pointers device, the
 * whole grid on one GPU: the particles (x, y[, z], optional mass; unit mass when NULL) are histogrammed on the plan's grid with
 * np.histogramdd's bin rule against edges_dev [npix + 1] -> map_in_dev, and the cell-owned pass of bfgx_grid_baryonify_device turns it
 * into map_out_dev.  Same results as bfgx_deposit_particles_device followed by bfgx_grid_baryonify_device; the deposit stores map_out's
 * start value and the map's sum along with map_in, so the map is not read again to copy it.  sums_dev as above. */
int  bfgx_grid_deposit_baryonify_device(bfgx_grid_plan *p, const bfgx_grid_catalog *cat_dev, int64_t n_particles, const double *x,
                                        const double *y, const double *z, const double *mass, const double *edges_dev, double *map_in_dev,
                                        double *map_out_dev, double *sums_dev, int64_t *n_pairs_host);
int  bfgx_grid_plan_timing_enable(bfgx_grid_plan *p, int on);
int  bfgx_grid_plan_timing_read(bfgx_grid_plan *p, double *ms_sum, int64_t *launches);
/* device-resident variants of the two deposit kernels and the P(k) summary (all pointers device) */
int  bfgx_deposit_particles_device(int device, void *hip_stream, int32_t ndim, int64_t n, const double *x, const double *y,
                                   const double *z, const double *mass, int32_t n_grid, const double *edges_dev,
                                   double *map_out_dev);
/* the same for the planes [plane_lo, plane_lo + plane_n) of the FIRST axis only (slab decomposition over GPUs): map_out_dev
 * holds plane_n x n_grid (x n_grid) cells, particles of other planes are dropped */
int  bfgx_deposit_particles_slab_device(int device, void *hip_stream, int32_t ndim, int64_t n, const double *x, const double *y,
                                        const double *z, const double *mass, int32_t n_grid, const double *edges_dev,
                                        int32_t plane_lo, int32_t plane_n, double *map_out_dev);
/* work_dev: scratch of bfgx_power_spectrum_work_doubles(n_grid) doubles (the half spectrum, complex [n_grid][n_grid][pitch] with
 * rows padded to whole 128-byte lines; its content is undefined afterwards: the last FFT pass bins |F|^2 straight from LDS) */
int64_t bfgx_power_spectrum_work_doubles(int32_t n_grid);
/* complex values per row of the half spectrum in every work array below: n_grid/2 + 1 rounded up to a multiple of 8 */
int32_t bfgx_fft_pitch(int32_t n_grid);
/* routing of particles to the ranks that own their planes (slab decomposition: rank r owns the planes [r n/world, (r + 1) n/world) of
 * the first axis; bin rule of np.histogramdd on x, particles outside the edges are dropped).  Pass 1 (enqueue-only): owner_dev[n]
 * (uint8 scratch) and counts_dev[world] (int32) = particles bound for every rank.  Pass 2, after the caller has read the counts:
 * cols_out_dev[c][start[d] + ...] = column c of the particles bound for rank d (ncols <= 4 columns of `total` = sum(counts) kept
 * particles each, any order inside a destination); cursor_dev[world] int32 scratch. */
int  bfgx_route_particles_count_device(int device, void *hip_stream, int64_t n, const double *x_dev, int32_t n_grid, const double *edges_dev,
                                       int32_t world, uint8_t *owner_dev, int32_t *counts_dev);
int  bfgx_route_particles_fill_device(int device, void *hip_stream, int64_t n, int32_t ncols, const double *const *cols_dev,
                                      const uint8_t *owner_dev, int32_t world, const int64_t *start_host, int64_t total,
                                      int32_t *cursor_dev, double *cols_out_dev);
int  bfgx_power_spectrum_device(int device, void *hip_stream, int32_t n_grid, const double *map_dev, double L, int32_t nk,
                                double *work_dev, double *pk_sum_dev, double *k_sum_dev, unsigned long long *counts_dev);

/* The same P(k) summary for a grid that is slab-decomposed over GPUs: (1) every rank transforms its `planes` planes of the
 * first axis along the last two axes (map [planes][n][n] -> complex work [planes][n][pitch], pitch = bfgx_fft_pitch(n): the pad
 * columns hold nothing and are never looked at); (2) the caller transposes between the ranks (all_to_all) so that every rank holds
 * all n planes of `ncols` columns of the MIDDLE axis, work [n][ncols][pitch]; (3) every rank transforms along the first axis and bins its modes (col0 = its first column; work is
 * scratch afterwards): the three sums are partial and are added over the ranks by the caller. */
int  bfgx_fft_slab_planes_device(int device, void *hip_stream, int32_t n_grid, int32_t planes, const double *map_dev, double *work_dev);
int  bfgx_fft_slab_axis0_pk_device(int device, void *hip_stream, int32_t n_grid, int32_t ncols, int32_t col0, double *work_dev, double L,
                                   int32_t nk, double *pk_sum_dev, double *k_sum_dev, unsigned long long *counts_dev);

/* ---- particle-snapshot path ----------------------------------------------------------
 * Replaces BaryonifySnapshot.process, BaryonForge/Runners/SnapshotRunner.py:173-262: every particle within
 * R_q = clip(epsilon_max R200c / a, 0, L/2) of a halo (periodic box) moves radially by displacement(d, M, a) * a;
 * positions are re-wrapped into [0, L] once.  Halos are given as bfgx_grid_catalog (float32-valued columns, lnM as
 * for the grid runners; rmat / extra unused).  cosmo_runner.w0 = -1 (SnapshotRunner.py:204-207 does not pass w0). */
typedef struct bfgx_snapshot {
    int32_t ndim;                         /* 2 or 3 */
    int32_t _pad;
    int64_t n;                            /* particles */
    const double *x, *y, *z;              /* coordinates in [0, L]; z may be NULL for 2D */
    double L;                             /* box size, comoving Mpc */
    double redshift;
} bfgx_snapshot;

int bfgx_baryonify_snapshot(const bfgx_grid_catalog *halos_host, const bfgx_model *model, const bfgx_snapshot *snap_host,
                            double *x_out_host, double *y_out_host, double *z_out_host, const bfgx_opts *opts, bfgx_stats *stats);
/* The same on the snapshot's RECORDS as ParticleSnapshot keeps them (io.py:378-470: one structured array `cat` with float64 fields M, x, y,
 * z): records_in [n][itemsize bytes] is copied to the device in chunks, the coordinate fields at byte offsets off_x / off_y / off_z (off_z
 * ignored in 2-D) are displaced in place, and the records -- every other field untouched -- arrive in records_out (may not alias
 * records_in): what `new_cat = cat.copy(); new_cat['x'] = ...` (SnapshotRunner.py:254-262) leaves, without a strided gather or scatter
 * of the columns on the host.  Upload, displacement and download of successive chunks overlap; plan and device buffer are cached
 * (bfgx_cache_clear).  itemsize and the offsets must be multiples of 8. */
int bfgx_baryonify_snapshot_records(const bfgx_grid_catalog *halos_host, const bfgx_model *model, int32_t ndim, double L, double redshift,
                                    int64_t n, const void *records_in, void *records_out, int32_t itemsize, int32_t off_x, int32_t off_y,
                                    int32_t off_z, const bfgx_opts *opts, bfgx_stats *stats);
/* ParticleSnapshot(cat = BaryonifySnapshot(...).process()).make_map(n_grid) in one call (SnapshotRunner.py:173-262 then io.py:622-670), for a
 * caller who wants the map of the displaced particles only: the records go up once, the displaced coordinates are never stored or sent back,
 * map_out [n_grid^ndim] float64 (host) = np.histogramdd of the displaced positions on edges [n_grid + 1] (host) with the weights of the float64
 * field at off_mass (off_mass < 0: unit masses; NaN masses -> BFGX_ERR_ASSERT, io.py:636).  Same record rules as above. */
int bfgx_baryonify_snapshot_records_map(const bfgx_grid_catalog *halos_host, const bfgx_model *model, int32_t ndim, double L, double redshift,
                                        int64_t n, const void *records_in, int32_t itemsize, int32_t off_x, int32_t off_y, int32_t off_z,
                                        int32_t off_mass, int32_t n_grid, const double *edges, double *map_out, const bfgx_opts *opts,
                                        bfgx_stats *stats);
/* resident form: the model and the halo-cell workspace live in a plan (one per box / redshift / model); all columns and
 * outputs are device pointers (out may not alias in); blocking (two small read-backs) */
typedef struct bfgx_snapshot_plan bfgx_snapshot_plan;
int  bfgx_snapshot_plan_create(int device, void *hip_stream, const bfgx_model *model, int32_t ndim, double L, double redshift,
                               int64_t max_halos, bfgx_snapshot_plan **out);
void bfgx_snapshot_plan_destroy(bfgx_snapshot_plan *p);
int  bfgx_snapshot_displace_device(bfgx_snapshot_plan *p, const bfgx_grid_catalog *halos_dev, int64_t n_part, const double *x_dev,
                                   const double *y_dev, const double *z_dev, double *x_out_dev, double *y_out_dev, double *z_out_dev,
                                   int64_t *n_pairs_host);
/* BaryonifySnapshot.process() (SnapshotRunner.py:173-262) followed by ParticleSnapshot.make_map(n_grid) (io.py:622-670) of the result,
 * for a caller who wants the MAP of the displaced particles and not the particles: map_out_dev [n_grid^ndim] float64 = np.histogramdd of the
 * displaced, re-wrapped positions on edges_dev [n_grid + 1] with weights mass_dev (NULL: unit masses).  The displaced coordinates are never
 * stored: the displacement kernel writes the deposit's sort keys.  BFGX_ERR_UNSUPPORTED for grids the tile-owned deposit does not take
 * (call bfgx_snapshot_displace_device + bfgx_deposit_particles_device then).  Blocking like bfgx_snapshot_displace_device. */
int  bfgx_snapshot_displace_deposit_device(bfgx_snapshot_plan *p, const bfgx_grid_catalog *halos_dev, int64_t n_part, const double *x_dev,
                                           const double *y_dev, const double *z_dev, const double *mass_dev, int32_t n_grid,
                                           const double *edges_dev, double *map_out_dev, int64_t *n_pairs_host);
/* one call = plan create + displace + destroy */
int bfgx_baryonify_snapshot_device(int device, void *hip_stream, const bfgx_grid_catalog *halos_dev, const bfgx_model *model,
                                   const bfgx_snapshot *snap_dev, double *x_out_dev, double *y_out_dev, double *z_out_dev,
                                   int64_t *n_pairs_host);

/* ---- models that are Python callables --------------------------------------------------------------
 * The reference calls model.displacement(r_sep / a_j, M_j, a_j) (HealpixRunner.py:321) / model.projected(cosmo, r_sep / a_j, M_j, a_j)
 * (:441) once per halo on ANY object.  For a model that carries no table the binding serves it exactly so:
 *   begin   the halos' discs on the device (`model` carries the runner's cosmology / mass definition / epsilon_max; its 3-axis table must be
 *           valid and is ignored); counts_host[n] = pixels of every halo's disc (4 for BaryonifyShell's < 4-pixel fallback, :309-310)
 *   radii   r_host[sum counts] = r_sep / a_j of every (halo, pixel) pair, halo j's at the exclusive prefix sum of the counts
 *   (the caller evaluates its model per halo on that halo's slice of r_host)
 *   apply   vals_host[sum counts] = what the model returned.  paint = 0: offset = value a diff / r_sep, non-finite components -> 0, accumulated
 *           in the pixels' unit vectors (:321-331), then the regrid of map_in into map_out (:333-341) and the mass check (:344-346: BFGX_ERR_MASS);
 *           paint = 1: map_out[pixel] += value, non-finite -> 0 (:441-445; map_in ignored).  fp64 throughout.
 *   end     frees the handle (also after an error). */
typedef struct bfgx_pairs bfgx_pairs;
int  bfgx_shell_pairs_begin(const bfgx_catalog *cat_host, const bfgx_model *model, int64_t nside, int32_t paint, int32_t device, bfgx_pairs **out,
                            int64_t *counts_host);
int  bfgx_shell_pairs_radii(bfgx_pairs *h, double *r_host);
int  bfgx_shell_pairs_apply(bfgx_pairs *h, const double *vals_host, const double *map_in, double *map_out, int32_t check_mass, bfgx_stats *stats);
void bfgx_shell_pairs_end(bfgx_pairs *h);

#ifdef __cplusplus
}
#endif
#endif /* BFGX_H */
