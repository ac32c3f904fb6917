// Fast tiled scatter for gfx950 (MI355X): K1 (BaryonifyShell halo loop, HealpixRunner.py:306-331) and K3
// (PaintProfilesShell, HealpixRunner.py:432-445) for a 3-axis table with a uniform ln r axis.  The code every chunk runs is
// written for "narrow" discs (no pole inside, every pixel within 0.40 rad of the halo's azimuth); a chunk that lists a WIDE disc
// (a pole inside, very low redshift: a handful of halos in a full-sky catalog) takes a second copy of the row and pair phases
// with the generic kernel's row spans and full-range sin / cos (k1_chunk, rows_and_pairs).  Property-axis tables and non-uniform
// radial axes stay on the generic tile kernel of bfgx_kernels.hpp (which, with BFGX_K1_WIDE=0, also takes the wide discs in a
// "wide pass" that adds into the stored tiles: the arrangement up to round 4, kept for the tests).
//
// One 512-thread workgroup owns one tile (BR rings x <= W pixels) of the output: its accumulators are fp64 planes in LDS
// (one plane per component, so that consecutive pixels hit consecutive banks), every pixel is stored exactly once.
// Each wave takes 16 entries (halos touching the tile) at a time:
//   lanes = entries   -> ring range clipped to the tile, pair records staged in LDS
//   lanes = ring rows -> the exact fp64 query_disc span of the row (healpix_cxx arithmetic), clipped to the tile; per row
//                        the differences that the chord needs are formed ONCE in fp64:  dz = z_ring - z0,
//                        ds = sin(theta_ring) - sin(theta0), x0 = azimuth difference of the row's first pixel
//   lanes = pairs     -> with u = (v_pix - v_halo) in the frame rotated by -phi0,
//                            u = (ds - sth (1 - cos x), sth sin x, dz),   x = x0 + j dphi,
//                        nothing in the pair phase subtracts nearly equal numbers any more, so it runs in the precision of
//                        the output (fp32 for the default fp32 pix_offsets, fp64 otherwise): r = D |u|, ln r from one
//                        v_log_f32, the (z, M)-blended radial read-out from ONE 32-byte record of an interleaved copy of the
//                        table, and the renormalised offset (v + e)/|v + e| - v as a series in e = offset / D.
// 4 waves/SIMD: <= 128 VGPRs, 76 KB of LDS per workgroup (fp32 pix_offsets).
#pragma once
#include "bfgx_kernels.hpp"
#include <type_traits>

namespace bfgx {

#ifndef BFGX_ABL2
#define BFGX_ABL2 0               // >0: timing-only ablation builds (scripts/ablate2.sh); never shipped
#endif
constexpr int kW2 = 8;            // waves per workgroup
#ifndef BFGX_CHUNK2
#define BFGX_CHUNK2 16
#endif
constexpr int kChunk2 = BFGX_CHUNK2;   // entries a wave takes at a time (8 / 12 / 20 measured: 0.486 / 0.466 / see DESIGN section 4)
constexpr int kPlanePad = 11;     // doubles between accumulator planes: plane stride = 22 banks mod 64 (conflict-free flush)

// ---------------------------------------------------------------------------------- pair-phase math per precision
template <typename real> struct PMath;
__device__ inline float fma_(float a, float b, float c) { return __builtin_fmaf(a, b, c); }      // __builtin_fma is the double form
__device__ inline double fma_(double a, double b, double c) { return __builtin_fma(a, b, c); }

template <> struct PMath<float> {
    static constexpr float kHuge = 1.0e37f;
    // v_rsq_f32 is good to 1 ulp, which is what every other fp32 step of the pair phase carries: no Newton step
    static __device__ inline float rsq(float x) { return __builtin_amdgcn_rsqf(x); }
    static __device__ inline float half_ln(float x) { return 0.34657359027997264f * __builtin_amdgcn_logf(x); }   // 0.5 ln 2 log2 x
    static __device__ inline float expv(float d) { return __builtin_amdgcn_exp2f(d * 1.4426950408889634f); }
    // sin x and 1 - cos x for |x| <= 0.5 (truncation < 5e-9 relative)
    static __device__ inline void sin_omc(float x, float &sn, float &omc)
    {
        const float u = x * x;
        float ps = -1.0f / 5040.0f;                                        // (x^9 / 9! < 5e-9 x for |x| <= 0.5)
        ps = __builtin_fmaf(ps, u, 1.0f / 120.0f);
        ps = __builtin_fmaf(ps, u, -1.0f / 6.0f);
        sn = __builtin_fmaf(x * u, ps, x);
        float pc = -1.0f / 40320.0f;
        pc = __builtin_fmaf(pc, u, 1.0f / 720.0f);
        pc = __builtin_fmaf(pc, u, -1.0f / 24.0f);
        pc = __builtin_fmaf(pc, u, 0.5f);
        omc = u * pc;
    }
    static __device__ inline bool finite(float v) { return __builtin_isfinite(v); }
    static __device__ inline float med3(float x, float lo, float hi) { return __builtin_amdgcn_fmed3f(x, lo, hi); }
};

template <> struct PMath<double> {
    static constexpr double kHuge = 1.0e37;
    static __device__ inline double rsq(double x) { return fast_rsq(x); }
    static __device__ inline double half_ln(double x) { return 0.5 * fast_log(x); }
    static __device__ inline void sin_omc(double x, double &sn, double &omc)
    {
        double sh, ch;
        sincos_small(0.5 * x, sh, ch);                                     // half angle: 1 - cos x = 2 sin^2(x/2), no cancellation
        sn = 2.0 * sh * ch;
        omc = 2.0 * sh * sh;
    }
    static __device__ inline double expv(double d) { return fast_exp(d); }
    static __device__ inline bool finite(double v) { return __builtin_isfinite(v); }
    static __device__ inline double med3(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
};

// The PARITY-GRADE pair phase (acc mode 3, bfgx.h): fp64 data path whose elementary functions carry ~1e-11 instead of 4e-16 -- fp32 hardware
// seeds and ONE Newton step, series cut where the truncation falls below 5e-11.  A displaced pixel needs its offset to ~1e-8 of itself for the
// regridded map to hold SURVEY 8(d)'s 1e-6 mean(map) at displacements of 20 pixels (fp32 pair math: 4e-7, i.e. 2e-5 mean(map) there); the
// 1e-10 parity path (acc mode 1) keeps PMath<double>.  Measured on the S19 table at config 2: K1 0.652 -> 0.606 ms, map within 1.7e-10 mean(map)
// of the full-precision fp64 result.
struct PMathE {
    static constexpr double kHuge = 1.0e37;
    static __device__ inline double rsq(double x)                        // x in [1e-37, ~4]: inside the fp32 range
    {
        const double y = (double)__builtin_amdgcn_rsqf((float)x);
        const double h = 0.5 * y, e = __builtin_fma(-x * y, h, 0.5);
        return __builtin_fma(y, e, y);                                   // 1.5 (1.5e-7)^2
    }
    static __device__ inline double half_ln(double x)
    {
        double m = __builtin_amdgcn_frexp_mant(x);
        int e = __builtin_amdgcn_frexp_exp(x);
        const bool lowm = m < 0.70710678118654752440;
        m = lowm ? 2.0 * m : m;
        e = lowm ? e - 1 : e;
        const double den = m + 1.0;
        double y = (double)__builtin_amdgcn_rcpf((float)den);
        y = __builtin_fma(y, __builtin_fma(-den, y, 1.0), y);
        const double s = (m - 1.0) * y, u = s * s;                       // |s| <= 0.1716: the series to s^11 leaves u^6 / 13 = 5e-11
        double p = 1.0 / 11.0;
        p = __builtin_fma(p, u, 1.0 / 9.0);
        p = __builtin_fma(p, u, 1.0 / 7.0);
        p = __builtin_fma(p, u, 1.0 / 5.0);
        p = __builtin_fma(p, u, 1.0 / 3.0);
        const double hl = __builtin_fma(s * u, p, s);                    // ln(m) / 2
        return __builtin_fma((double)e, 0.34657359027997264, hl);
    }
    static __device__ inline double expv(double d) { return fast_exp(d); }
    static __device__ inline void sin_omc(double x, double &sn, double &omc)      // |x| <= 0.5: x^11 / 11! and x^12 / 12! dropped (2e-11, 4e-12)
    {
        const double u = x * x;
        double ps = 1.0 / 362880.0;
        ps = __builtin_fma(ps, u, -1.0 / 5040.0);
        ps = __builtin_fma(ps, u, 1.0 / 120.0);
        ps = __builtin_fma(ps, u, -1.0 / 6.0);
        sn = __builtin_fma(x * u, ps, x);
        double pc = 1.0 / 3628800.0;
        pc = __builtin_fma(pc, u, -1.0 / 40320.0);
        pc = __builtin_fma(pc, u, 1.0 / 720.0);
        pc = __builtin_fma(pc, u, -1.0 / 24.0);
        pc = __builtin_fma(pc, u, 0.5);
        omc = u * pc;
    }
    static __device__ inline bool finite(double v) { return __builtin_isfinite(v); }
    static __device__ inline double med3(double x, double lo, double hi) { return fmin(fmax(x, lo), hi); }
};
// PM = 1: the parity-grade mode (PMathE pair math, pix_offsets stored as two fp32 arrays hi + lo); 0: the plain kernels
template <typename real, int PM> struct PMSel { using type = PMath<real>; };
template <> struct PMSel<double, 1> { using type = PMathE; };

// ---------------------------------------------------------------------------------- LDS records
// one clipped ring row (a contiguous pixel run inside the tile) as the pair phase sees it
template <typename real>
struct alignas(16) RowC2 {
    uint32_t pk;                  // pair prefix (12 bits) | rotated column of the first pixel (6) | ring in tile (6) | entry slot (4)
    real dz, ds, x0;              // z_ring - z0, sin(theta_ring) - sin(theta0), azimuth difference of the first pixel
};
// what the pair phase needs of a ring of the tile
template <typename real>
struct alignas(16) RingC2 { real sth, zf, dphi, _pad; };      // sin / cos of the colatitude, 2 pi / nr

struct EntC2 { int32_t hidx, prefix, ring_lo, fb; };      // one non-empty entry of the chunk (compacted)
struct RowGeo { double z0, s0, xa, cosr, phi0; };          // what the ring-row phase needs of a halo's RowRec, staged per entry (BFGX_K1_GEO)
#ifndef BFGX_K1_GEO
#define BFGX_K1_GEO 0
#endif

// entries per chunk and row slots per pass, by the precision of the pair phase (LDS per wave: 2.9 KB fp32, 4.5 KB fp64 at 16 / 64)
// (measured, fp64 pair math: sixteen waves per workgroup with chunks of 8 entries and 32 row slots -- what LDS then holds -- 0.739 ms against
// 0.606 with twelve waves and 16 / 64: small chunks cost more than the fourth wave per SIMD hides)
template <typename real> struct K1Cfg { static constexpr int chunk = kChunk2, rowl = kWave; };
template <typename real>
struct Wave2Lds {
    PairRecT<real> pair[K1Cfg<real>::chunk];
    RowC2<real> rows[K1Cfg<real>::rowl];
    unsigned long long mask[K1Cfg<real>::rowl + 4];      // bit t set <=> pair t is the first pair of a row (<= 64 rows x 64 pixels)
    EntC2 ent[K1Cfg<real>::chunk];
    unsigned long long emask[K1Cfg<real>::chunk];       // bit R set <=> row R of the chunk is the first row of an entry (<= 16 x 64 rows)
#if BFGX_K1_GEO
    RowGeo geo[K1Cfg<real>::chunk];
#endif
};

template <typename real>
__host__ __device__ inline size_t tile2_lds_bytes(int BR, int W, int ncomp)
{
    size_t a = (size_t)ncomp * ((size_t)BR * W + kPlanePad) * sizeof(double);
    a = (a + 15) & ~(size_t)15;
    return a + sizeof(Wave2Lds<real>) * kW2 + 16 + (sizeof(TileRow) + sizeof(RingC2<real>)) * (size_t)BR;
}

// atan(t) for 0 <= t <= 7/16: fdlibm's kernel polynomial (s_atan.c, no argument reduction needed below 7/16), < 1 ulp
__device__ inline double atan_lt_7_16(double t)
{
    const double z = t * t, w = z * z;
    double s1 = 1.62858201153657823623e-02;
    s1 = __builtin_fma(s1, w, 4.97687799461593236017e-02);
    s1 = __builtin_fma(s1, w, 6.66107313738753120669e-02);
    s1 = __builtin_fma(s1, w, 9.09088713343650656196e-02);
    s1 = __builtin_fma(s1, w, 1.42857142725034663711e-01);
    s1 = __builtin_fma(s1, w, 3.33333333333329318027e-01);
    double s2 = -3.65315727442169155270e-02;
    s2 = __builtin_fma(s2, w, -5.83357013379057348645e-02);
    s2 = __builtin_fma(s2, w, -7.69187620504482999495e-02);
    s2 = __builtin_fma(s2, w, -1.11111104054623557880e-01);
    s2 = __builtin_fma(s2, w, -1.99999999998764832476e-01);
    return __builtin_fma(-t, __builtin_fma(z, s1, w * s2), t);
}

// the pixel span [lo, lo + cnt) (mod nr) of a disc in one ring: healpix_cxx query_disc_internal (fact = 0) for a ring
// inside [irmin, irmax] (narrow discs have no polar-cap rows).  x and ysq are formed exactly as disc_row_span forms them;
// the half-width dphi = atan2(sqrt(ysq), x) of a narrow disc (0 < dphi <= 0.4, so x > 0 and sqrt(ysq) / x <= tan 0.4 < 7/16)
// comes from an rsq seed + Newton and the fdlibm polynomial instead of libm's sqrt and atan2 (a few ulp either way: the
// span changes only where fnr (phi0 +- dphi) lies within ~1e-13 of a pixel boundary).  Ring lengths are below 2^31.
__device__ inline void disc_row_span_narrow(int nr, bool shifted, double z, double fnr, double z0, double xa, double cosr,
                                            double phi0, int &lo, int &cnt)
{
    lo = 0; cnt = 0;
    const double x = (cosr - z * z0) * xa;
    const double ysq = 1.0 - z * z - x * x;
    if (!(ysq > 0.0) || !(x > 0.0)) return;
    const double t = ysq * fast_rsq(ysq) * fast_rcp(x);              // sqrt(ysq) / x
    const double dphi = atan_lt_7_16(t);
    if (!(dphi > 0.0)) return;
    const double sh = shifted ? 0.5 : 0.0;
    const int ip_lo = (int)floor(fnr * (phi0 - dphi) - sh) + 1;
    const int ip_hi = (int)floor(fnr * (phi0 + dphi) - sh);
    cnt = max(0, min(ip_hi - ip_lo + 1, nr));
    int l = ip_lo;
    if (l < 0) l += nr;
    if (l >= nr) l -= nr;
    if (l < 0) l += nr;
    lo = l;
}

// the model-side cut r_sep / a < eps R of ONE pair, chord in fp64 (only where the fp32 chord is within 4e-6 of the cut)
__device__ __noinline__ bool exact_cut_test(const TileRow &tr, const RowRec &rr, int kloc)
{
    const int kk = tr.ks + kloc;
    const double xx = fold_dphi(__builtin_fma((double)kk + (tr.shifted ? 0.5 : 0.0), tr.dphi, -rr.phi0));
    double sh, ch;
    if (fabs(xx) <= 1.0) sincos_small(0.5 * xx, sh, ch);
    else sincos_bounded(0.5 * xx, sh, ch);                              // (a wide disc: RowRec.fb bit 1)
    const double s2 = 2.0 * sh * ch, o2 = 2.0 * sh * sh;
    const double wx = (tr.sth - rr.s0) - tr.sth * o2, wy = tr.sth * s2, wz = tr.z - rr.z0;
    return (wx * wx + wy * wy + wz * wz) < rr.cut2;
}

// A WIDE disc (RowRec.fb bit 1: a pole inside, or pixels further than 0.40 rad from the halo's azimuth; a handful of polar halos in a
// full-sky catalog) goes through the same chunk: its row spans by the general arithmetic of the generic kernel (polar-cap rows are whole
// rings, atan2 over the full range), the sin / cos of its pairs over the full range.  Both are calls, so that the narrow path keeps its registers.
__device__ __noinline__ int2 disc_row_span_wide(const TileRow &tr, int ring, const RowRec &rr, const FbRec &fr)
{
    int slo, scnt;
    disc_row_span(tr.nr, tr.shifted != 0, tr.z, tr.fnr, ring, rr.z0, rr.xa, rr.cosr, rr.phi0, fr.ring[0], fr.ring[1], slo, scnt);
    return make_int2(slo, scnt);
}
// (sin x, 1 - cos x) for any |x| <~ 1e3, through the half angle in fp64
__device__ __noinline__ double2 sin_omc_wide(double x)
{
    double sh, ch;
    sincos_bounded(0.5 * x, sh, ch);
    return make_double2(2.0 * sh * ch, 2.0 * sh * sh);
}

__device__ inline unsigned long long wave_uniform64(unsigned long long v)
{
    const unsigned lo = __builtin_amdgcn_readfirstlane((unsigned)v), hi = __builtin_amdgcn_readfirstlane((unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}

// ---------------------------------------------------------------------------------- one (halo, pixel) pair
template <typename real>
struct PairEval {
    real v0, v1, v2;              // MODE_OFFSETS: nw_vec - vec (HealpixRunner.py:326-328); MODE_PAINT: v0 = Paint (:441)
    int la;                       // LDS pixel offset in the plane (rotated column)
    int hidx;
    bool ok, ok_nocut, amb;       // contributes; the same without the model-side cut; fp32 cannot decide the cut
};

// Branch-free evaluation of pair t of the current row block.
template <int MODE, typename real, bool FAR = false, int PM = 0>
__device__ inline void pair_eval(PairEval<real> &o, const Tab8T<real> &tb, const RowC2<real> rc, const PairRecT<real> *pairs,
                                 const RingC2<real> *ringc, int t, bool act, int wsh, int wmask)
{
    using PMt = typename PMSel<real, PM>::type;
    const int jj = t - (int)(rc.pk & 0xFFFu);
    const int rl = (int)((rc.pk >> 18) & 63u);
    o.la = (rl << wsh) + (((int)((rc.pk >> 12) & 63u) + jj) & wmask);
    const PairRecT<real> ph = pairs[(rc.pk >> 24) & 31u];
    const RingC2<real> rg = ringc[rl];
    o.hidx = ph.hidx;
    const real x = fma_((real)jj, rg.dphi, rc.x0);
    real sn, omc;
    PMt::sin_omc(x, sn, omc);
    if (FAR) {                                                         // (a row pass that holds pairs beyond 0.5 rad of their halo's azimuth: k1_chunk)
        if (act && !(fabs((double)x) <= 0.5)) { const double2 w = sin_omc_wide((double)x); sn = (real)w.x; omc = (real)w.y; }
    }
    const real ux = fma_(-rg.sth, omc, rc.ds), uy = rg.sth * sn, uz = rc.dz;   // (v_pix - v_halo) / D, HealpixRunner.py:314-316
    // |u|^2.  Painting needs nothing else of u, and with sin^2 x + (1 - cos x)^2 = 2 (1 - cos x):
    //   |u|^2 = ds^2 + dz^2 + 2 sth (sth - ds) (1 - cos x),   sth - ds = sin(theta_halo)
    // -- the chord to the ring's point at the halo's azimuth plus the azimuthal part, both positive: sin x, ux and uy drop out
    const real u2 = (MODE == MODE_PAINT) ? fma_((real)2 * rg.sth * (rg.sth - rc.ds), omc, fma_(rc.ds, rc.ds, rc.dz * rc.dz))
                                         : ux * ux + uy * uy + uz * uz;
    // r_sep = 0 (diff / r_sep is NaN -> 0, :322-323): |u|^2 is floored at 1e-37, whose ln lies below any table (and u = 0 adds nothing
    // anyway); a halo outside the (z, M) table carries scale2 = +1e30 (K0), whose logarithm fails the range test: no flags of their own
    bool ok = act;
    const real u2s = (u2 > (real)1e-37) ? u2 : (real)1e-37;
    const real rinv = PMt::rsq(u2s);                                // 1 / |u|
    const real lx = PMt::half_ln(u2s * ph.scale2);                  // ln(r_sep / a) [- ln R when Rdelta]
    ok = ok && (lx >= tb.r0) && (lx <= tb.r1);                     // RGI fill_value = nan
    const real uu = (lx - tb.r0) * tb.inv_dr;
    const real uc = PMt::med3(uu, (real)0, (real)(tb.nr - 2));      // (clamped BEFORE the conversion: uu may be +-huge)
    const int i = (int)uc;
    const real tr_ = uu - (real)i;
    const real *tp = tb.v + (unsigned)(ph.cell + i * 8);           // (cell >= 0: 32-bit offset from the uniform table base)
    real q[8];
    if (BFGX_ABL2 == 6) { for (int k = 0; k < 8; ++k) q[k] = tr_ * (real)(k + 1); }
    else if (sizeof(real) == 4) {
        const float4 a0 = reinterpret_cast<const float4 *>(tp)[0], a1 = reinterpret_cast<const float4 *>(tp)[1];
        q[0] = a0.x; q[1] = a0.y; q[2] = a0.z; q[3] = a0.w; q[4] = a1.x; q[5] = a1.y; q[6] = a1.z; q[7] = a1.w;
    } else {
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const double2 a = reinterpret_cast<const double2 *>(tp)[k];
            q[2 * k] = a.x; q[2 * k + 1] = a.y;
        }
    }
    real d = (real)0;
#pragma unroll
    for (int k = 0; k < 4; ++k) d = fma_(ph.w[k], fma_(tr_, q[2 * k + 1], q[2 * k]), d);
    o.amb = false;
    if (MODE == MODE_PAINT) {
        const real paint = PMt::expv(d);                            // Tabulate.py:286
        o.ok = o.ok_nocut = ok && PMt::finite(paint) && paint != (real)0;       // :442
        o.v0 = paint; o.v1 = o.v2 = (real)0;
        return;
    }
    ok = ok && PMt::finite(d) && d != (real)0;                      // :323
    o.ok_nocut = ok;
    // BaryonCorrection.py:381-382: r < eps R  <=>  t = |u|^2 / cut^2 - 1 < 0.  The pair record carries 1 / cut^2 (0 when the disc itself
    // implies r < eps R: t = -1, always inside, never ambiguous); the fp32 decision is re-made in fp64 where |t| <= 4e-6
    const real tc = fma_(u2, ph.cut2, (real)-1);
    o.ok = ok && (tc < (real)0);
    if (sizeof(real) == 4) o.amb = ok && (fabsf((float)tc) <= 4e-6f);
    // offset / D = d a diff / (r_sep D) = (d a / D) u / |u|; renormalised (v + e)/|v + e| - v (:326-328) as a
    // series in t = 2 v.e + e.e (|v| = 1): e + g (v + e), g = -t/2 + 3 t^2/8 - 5 t^3/16 + 35 t^4/128
    const real sc = d * ph.aD * rinv;                              // e = sc u
    const real fx = fma_(-rg.sth, omc, rg.sth), fy = uy, fz = rg.zf;   // pixel unit vector v, rotated frame
    // v.u = v.(v - v_halo) = 1 - v.v_halo = |u|^2 / 2 for unit vectors, so t = 2 v.e + e.e = sc |u|^2 (1 + sc): no dot products
    const real tt = sc * u2 * ((real)1 + sc);
    real g = (real)(35.0 / 128.0);
    g = fma_(g, tt, (real)-0.3125);
    g = fma_(g, tt, (real)0.375);
    g = fma_(g, tt, (real)-0.5);
    g = g * tt;
    const real c1 = fma_(g, sc, sc);                               // e + g (v + e) = (1 + g) sc u + g v
    const real ex = fma_(c1, ux, g * fx), ey = fma_(c1, uy, g * fy), ez = fma_(c1, uz, g * fz);
    o.v0 = ex * ph.cph0 - ey * ph.sph0;                            // rotate back by +phi0
    o.v1 = ex * ph.sph0 + ey * ph.cph0;
    o.v2 = ez;
}

// ---------------------------------------------------------------------------------- one chunk of a tile's entry list
// What a wave does with one chunk -- the entries [ebeg, ebeg + ecnt), ecnt <= kChunk2 -- of a tile's narrow-halo list: entries -> ring rows -> pairs, accumulated into the tile's
// LDS planes `acc`.  Shared by the barrier-per-tile kernel (tile_scatter2_kernel) and the fluid kernel (tile_scatter2f_kernel).
template <int MODE, typename real, int NP, int PM = 0>
__device__ __forceinline__ void k1_chunk(const Tab8T<real> &tb, const RowRec *__restrict__ rowrecs, const PairRecT<real> *__restrict__ pairrecs,
                                         const FbRec *__restrict__ fbrecs, const int32_t *__restrict__ ea, const int32_t *__restrict__ eb,
                                         int na, int ebeg, int ecnt, int estride, int i0, int i1, int nphi, int wsh, int wmask, int PL,
                                         double *acc, Wave2Lds<real> &L, const TileRow *rowtab, const RingC2<real> *ringc, int lane,
                                         unsigned long long &npairs)
{
    const unsigned long long lt = (1ull << lane) - 1ull;
    // ---- lanes = entries of this chunk; the non-empty ones are compacted into L.ent / L.pair
    int nrows = 0;
    EntC2 en;
    en.hidx = 0; en.prefix = 0; en.ring_lo = 0; en.fb = 0;
    RowGeo gg;
    gg.z0 = gg.s0 = gg.xa = gg.cosr = gg.phi0 = 0.0;
    if (lane < ecnt) {
        const int ei = ebeg + lane * estride;            // (estride > 1: the chunk takes every estride-th entry of region A, see the fluid kernel)
        en.hidx = ei < na ? ea[ei] : eb[ei - na];
        const RowRec &rr = rowrecs[en.hidx];
        if (BFGX_K1_GEO) { gg.z0 = rr.z0; gg.s0 = rr.s0; gg.xa = rr.xa; gg.cosr = rr.cosr; gg.phi0 = rr.phi0; }
        en.fb = rr.fb;
        if (en.fb & 1) nrows = 4;
        else {
            const int lo = max(rr.rfirst, i0), hi = min(rr.rlast, i1 - 1);
            nrows = max(0, hi - lo + 1);
            en.ring_lo = lo;
        }
    }
    const int incl_e = wave_scan_incl(nrows, lane);
    const int total_rows = __builtin_amdgcn_readlane(incl_e, kWave - 1);
    en.prefix = incl_e - nrows;
    if (lane < K1Cfg<real>::chunk) L.emask[lane] = 0ull;
    __builtin_amdgcn_wave_barrier();
    {
        const unsigned long long nzE = __ballot(nrows > 0);
        if (nrows > 0) {
            const int slot = __popcll(nzE & lt);
            L.ent[slot] = en;
#if BFGX_K1_GEO
            L.geo[slot] = gg;
#endif
            if (MODE != MODE_COUNT) L.pair[slot] = pairrecs[en.hidx];
            atomicOr(&L.emask[en.prefix >> 6], 1ull << (en.prefix & 63));
        }
    }
    __builtin_amdgcn_wave_barrier();

    // a row can hold two pixel runs only in a tile that spans whole rings (a disc across phi = 0): such tiles (the
    // innermost polar bands) use 32 row lanes per pass so that the runs of one pass always fit the 64 row slots
    // (and a WIDE disc can hold two runs of a row in any tile -- a span that leaves only a gap inside the tile's slice: a chunk that lists one does the same)
    auto rows_and_pairs = [&](auto wide_tag) __attribute__((always_inline)) {
    constexpr bool WIDE = decltype(wide_tag)::value;
    const int rowlanes = (nphi == 1 || WIDE) ? K1Cfg<real>::rowl / 2 : K1Cfg<real>::rowl;
    for (int rb = 0; rb < (BFGX_ABL2 == 2 ? 0 : total_rows); rb += rowlanes) {
        // ---- lanes = ring rows (clipped to this tile)
        const int R = rb + lane;
        const bool rvalid = lane < rowlanes && R < total_rows;
        int es = 0;                                            // the entry this row belongs to: set bits of emask at or below R, minus 1
        {
            const int wq = min(R, total_rows - 1) >> 6, bq = min(R, total_rows - 1) & 63;
            for (int w = 0; w < wq; ++w) es += __popcll(L.emask[w]);          // at most 15 words; almost always 0 or 1
            es += __popcll(L.emask[wq] & ((2ull << bq) - 1ull)) - 1;
        }
        const EntC2 ee = L.ent[es];
        const int eh = ee.hidx, ep = ee.prefix, erl = ee.ring_lo, efb = ee.fb;
        int kA = 0, cA = 0, cB = 0, rloc = 0;
        double x0A = 0.0, x0B = 0.0, dzv = 0.0, dsv = 0.0;
        if (rvalid) {
            const int q = R - ep;
#if BFGX_K1_GEO
            const RowGeo rr = L.geo[es];                       // (staged by the entry phase: one global read per halo instead of one per ring row)
#else
            const RowRec &rr = rowrecs[eh];
#endif
            if (efb & 1) {
                const int ring = fbrecs[eh].ring[q], fk = fbrecs[eh].k[q];
                if (ring >= i0 && ring < i1) {
                    const TileRow &tr = rowtab[ring - i0];
                    rloc = ring - i0;
                    dzv = tr.z - rr.z0; dsv = tr.sth - rr.s0;
                    if (fk >= tr.ks && fk < tr.ke) {
                        kA = fk - tr.ks; cA = 1;
                        x0A = fold_dphi(__builtin_fma((double)fk + (tr.shifted ? 0.5 : 0.0), tr.dphi, -rr.phi0));
                    }
                }
            } else {
                const int ring = erl + q;
                const TileRow &tr = rowtab[ring - i0];
                int slo, scnt;
                if (WIDE && (efb & 2)) { const int2 sp = disc_row_span_wide(tr, ring, rowrecs[eh], fbrecs[eh]); slo = sp.x; scnt = sp.y; }
                else disc_row_span_narrow(tr.nr, tr.shifted != 0, tr.z, tr.fnr, rr.z0, rr.xa, rr.cosr, rr.phi0, slo, scnt);
                rloc = ring - i0;
                dzv = tr.z - rr.z0; dsv = tr.sth - rr.s0;
                const int nr = tr.nr, ks = tr.ks, ke = tr.ke;
                const double xoff = (tr.shifted ? 0.5 : 0.0) * tr.dphi - rr.phi0;
                const int endA = min(slo + scnt, nr);
                const int firstA = max(slo, ks);
                cA = max(0, min(endA, ke) - firstA);
                kA = firstA - ks;
                x0A = fold_dphi(__builtin_fma((double)firstA, tr.dphi, xoff));
                const int endB = slo + scnt - nr;              // > 0 when the row wraps past phi = 2 pi: second run [ks, endB)
                cB = max(0, min(endB, ke) - ks);
                x0B = fold_dphi(__builtin_fma((double)ks, tr.dphi, xoff));
            }
        }
        // pairs and row slots of this pass: one scan over (pairs | runs << 16)
        const int nrun = (cA > 0 ? 1 : 0) + (cB > 0 ? 1 : 0);
        const int incl = wave_scan_incl((cA + cB) | (nrun << 16), lane);
        const int tot2 = __builtin_amdgcn_readlane(incl, kWave - 1);
        const int total = tot2 & 0xFFFF;
        npairs += (unsigned long long)total;
        if (MODE == MODE_COUNT || total == 0 || BFGX_ABL2 == 3) continue;
        {
            // compact the runs into row slots and mark each run's first pair in a bit mask
            const int excl = incl - ((cA + cB) | (nrun << 16));
            const int pre = excl & 0xFFFF, slot = excl >> 16;
            const int nwords = (total + kWave - 1) / kWave + 2;
            for (int wI = lane; wI < nwords; wI += kWave) L.mask[wI] = 0ull;
            __builtin_amdgcn_wave_barrier();
            const int rot = ((rloc & 7) << wsh) >> 3;                  // ring r is rotated by (r & 7) W / 8 columns
            if (cA > 0) {
                RowC2<real> rc;
                rc.pk = (uint32_t)pre | ((uint32_t)((kA + rot) & wmask) << 12) | ((uint32_t)rloc << 18) | ((uint32_t)es << 24);
                rc.dz = (real)dzv; rc.ds = (real)dsv; rc.x0 = (real)x0A;
                L.rows[slot] = rc;
                atomicOr(&L.mask[pre >> 6], 1ull << (pre & 63));
            }
            if (cB > 0) {
                const int preB = pre + cA;
                RowC2<real> rc;
                rc.pk = (uint32_t)preB | ((uint32_t)(rot & wmask) << 12) | ((uint32_t)rloc << 18) | ((uint32_t)es << 24);
                rc.dz = (real)dzv; rc.ds = (real)dsv; rc.x0 = (real)x0B;
                L.rows[slot + (cA > 0 ? 1 : 0)] = rc;
                atomicOr(&L.mask[preB >> 6], 1ull << (preB & 63));
            }
            __builtin_amdgcn_wave_barrier();

            // ---- lanes = (halo, pixel) pairs, NP per lane per trip: the evaluations are straight-line code so that
            // the LDS reads and table loads of the NP pairs are in flight together
            int base = 0;                                      // rows started before the current 64 pairs
            // the mask word and the row record of the NEXT trip are fetched while this trip computes
            unsigned long long m_nx = wave_uniform64(L.mask[0]);
            RowC2<real> rc_nx[NP];
#pragma unroll
            for (int u = 0; u < NP; ++u) {
                const unsigned long long m = (u == 0) ? m_nx : wave_uniform64(L.mask[u]);
                const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                const int row = base + below + (int)((m >> lane) & 1ull) - 1;
                base += __popcll(m);
                rc_nx[u] = L.rows[(u * kWave + lane < total) ? row : 0];
            }
            for (int T0 = 0; T0 < (BFGX_ABL2 == 4 ? 0 : total); T0 += NP * kWave) {
                PairEval<real> pv[NP];
                RowC2<real> rc_cur[NP];
#pragma unroll
                for (int u = 0; u < NP; ++u) rc_cur[u] = rc_nx[u];
                const int T1 = T0 + NP * kWave;
                if (T1 < total) {
#pragma unroll
                    for (int u = 0; u < NP; ++u) {
                        const unsigned long long m = wave_uniform64(L.mask[(T1 >> 6) + u]);
                        const int below = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
                        const int row = base + below + (int)((m >> lane) & 1ull) - 1;
                        base += __popcll(m);
                        rc_nx[u] = L.rows[(T1 + u * kWave + lane < total) ? row : 0];
                    }
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    const int t = T0 + u * kWave + lane;
                    pair_eval<MODE, real, WIDE, PM>(pv[u], tb, rc_cur[u], L.pair, ringc, t, t < total, wsh, wmask);
                }
                if (MODE == MODE_OFFSETS && sizeof(real) == 4) {
                    // pairs whose fp32 chord is within 4e-6 of the model-side cut (BaryonCorrection.py:381-382): decide in fp64 (rare)
                    bool anyamb = false;
#pragma unroll
                    for (int u = 0; u < NP; ++u) anyamb = anyamb || pv[u].amb;
                    if (__builtin_expect(__any(anyamb), 0)) {
#pragma unroll
                        for (int u = 0; u < NP; ++u)
                            if (pv[u].amb) {
                                const int rl = pv[u].la >> wsh;
                                const bool in = exact_cut_test(rowtab[rl], rowrecs[pv[u].hidx], (pv[u].la - (((rl & 7) << wsh) >> 3)) & wmask);
                                pv[u].ok = pv[u].ok_nocut && in;
                            }
                    }
                }
#pragma unroll
                for (int u = 0; u < NP; ++u) {
                    if (BFGX_ABL2 == 5) { if (pv[u].ok && pv[u].v0 == (real)1.2345e30) acc[pv[u].la] = (double)(pv[u].v1 + pv[u].v2); }
                    else if (pv[u].ok) {
                        atomicAdd(acc + pv[u].la, (double)pv[u].v0);                         // ds_add_f64
                        if (MODE == MODE_OFFSETS) {
                            atomicAdd(acc + PL + pv[u].la, (double)pv[u].v1);
                            atomicAdd(acc + 2 * PL + pv[u].la, (double)pv[u].v2);
                        }
                    }
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
    }
    };
    // A chunk that lists a WIDE disc (RowRec.fb bit 1; a handful of polar halos in a full-sky catalog) takes a second copy of the row and pair
    // phases -- general row spans, sin / cos over the full range where a pair lies beyond 0.5 rad of its halo's azimuth --, so that the
    // copy every other chunk runs is the narrow code alone.
    if (__builtin_expect(__any((en.fb & 2) != 0), 0)) rows_and_pairs(std::true_type{}); else rows_and_pairs(std::false_type{});
}

// ---------------------------------------------------------------------------------- the kernel
template <int MODE, typename ACC, typename real, int NP, int PM = 0>
__global__ void __launch_bounds__(kWave * kW2, (sizeof(real) == 4 ? 4 : 2))
tile_scatter2_kernel(Tab8T<real> tb, Hpx h, Tiling T, const RowRec *__restrict__ rowrecs,
                     const PairRecT<real> *__restrict__ pairrecs, const FbRec *__restrict__ fbrecs,
                     const int32_t *__restrict__ tile_start, const int32_t *__restrict__ cnt_a, const int32_t *__restrict__ cnt_b,
                     const int32_t *__restrict__ entries, int64_t capacity, const int32_t *__restrict__ entries_a, int cap_a, int cnt_pad,
                     ACC *__restrict__ out, ACC *__restrict__ out_lo, unsigned long long *__restrict__ pair_total, unsigned int *__restrict__ tile_counter,
                     unsigned int *__restrict__ omax2, int tile_lo, int tile_n, const int32_t *__restrict__ form, int my_form)
{
    // (PM = 1, the parity-grade mode: pix_offsets leave as two fp32 arrays, out = hi and out_lo = (float)(o - hi), indexed alike)
    static_assert(PM == 0 || (MODE == MODE_OFFSETS && sizeof(ACC) == 4 && sizeof(real) == 8), "PM = 1: fp64 pair math into split fp32 pix_offsets");
    // (form: which of the fast kernel's two forms runs was left to the device -- both are launched, the other one returns here)
    if (form != nullptr && *form != my_form) return;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NCOMP = (MODE == MODE_OFFSETS) ? 3 : 1;
    // persistent workgroups (two per CU): tiles are drawn from a device counter in the order of T.tile_order -- heavy
    // (equatorial) tiles first, the light polar ones fill the tail -- so that no workgroup launch sits between two tiles
    __shared__ int s_tile;
    while (true) {
    if (threadIdx.x == 0) s_tile = (int)atomicAdd(tile_counter, 1u);
    __syncthreads();
    const int tile_idx = s_tile;
    // tile_n < 0: the whole sphere, heaviest tiles first; tile_n >= 0: the tiles [tile_lo, tile_lo + tile_n) only (a rank that owns a
    // range of bands; `out` is then based so that only the pixels of those bands are touched)
    if (tile_idx >= (tile_n < 0 ? T.ntiles : tile_n)) break;
    const int tile = tile_n < 0 ? T.tile_order[tile_idx] : tile_lo + tile_idx;
    const int band = T.tile_band[tile];
    const int nphi = T.band_nphi[band];
    const int tj = tile - T.band_tile0[band];
    const int i0 = 1 + band * T.BR;
    const int i1 = min(i0 + T.BR, (int)(4 * h.nside));            // exclusive
    const int PL = T.BR * T.W + kPlanePad;                        // accumulator plane stride (doubles)
    const int wsh = __ffs(T.W) - 1, wmask = T.W - 1;              // W is a power of two (build_tiling)
    // LDS accumulators are always fp64: on gfx950 ds_add_f64 runs at ~7 lanes/clk/CU while ds_add_f32 manages only
    // ~0.3 (profiles/r01_ubench_lds_atomics.txt); ACC is only the type of the global output.
    double *acc = reinterpret_cast<double *>(smem);
    size_t off = ((size_t)NCOMP * PL * sizeof(double) + 15) & ~(size_t)15;
    using WaveLds = Wave2Lds<real>;
    WaveLds *wl = reinterpret_cast<WaveLds *>(smem + off);
    off += sizeof(WaveLds) * kW2;
    int *next_chunk = reinterpret_cast<int *>(smem + off);
    off += 16;
    TileRow *rowtab = reinterpret_cast<TileRow *>(smem + off);
    off += sizeof(TileRow) * (size_t)T.BR;
    RingC2<real> *ringc = reinterpret_cast<RingC2<real> *>(smem + off);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
    if (MODE != MODE_COUNT)
        for (int i = tid; i < NCOMP * PL; i += kWave * kW2) acc[i] = 0.0;
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
        RingC2<real> rg;
        rg.sth = (real)tr.sth; rg.zf = (real)tr.z; rg.dphi = (real)tr.dphi; rg._pad = (real)0;
        ringc[tid] = rg;
    }
    __syncthreads();

    // the narrow halos of the tile: region A = the tile's fixed-capacity list K0 filled directly (entries_a[tile][cap_a]; without it: the
    // head of the shared list), region B = discs over more than kRefMax tiles and what did not fit A, at tile_start in the shared list
    const int64_t e0 = tile_start[tile];
    const int na = entries_a ? min(cnt_a[(int64_t)tile * cnt_pad], cap_a) : cnt_a[tile];      // (direct placement: padded counters)
    const int32_t *ea = entries_a ? entries_a + (int64_t)tile * cap_a : entries + e0;
    const int64_t eb0 = e0 + (entries_a ? 0 : na);
    int64_t nb64 = cnt_b[tile];
    if (eb0 + nb64 > capacity) nb64 = capacity > eb0 ? capacity - eb0 : 0;
    if (!entries_a && e0 + na > capacity) nb64 = 0;
    const int32_t *eb = entries + eb0;
    const int ne = (!entries_a && e0 + na > capacity) ? (int)(capacity > e0 ? capacity - e0 : 0) : na + (int)nb64;
    // entries per chunk: at most kChunk2, fewer when the tile's list is short, so that every wave of the workgroup gets a chunk (a
    // large-NSIDE tile lists ~20 halos with hundreds of pixels each: in chunks of 16 entries two of the eight waves did all the
    // work).  One chunk per wave measured best (NSIDE 2048: K1 1.91 -> 1.56 ms, 8192: 56.8 -> 24.0 ms; 16 or 32 chunks per tile pack worse)
    const int csz = max(1, min(K1Cfg<real>::chunk, (ne + kW2 - 1) / kW2));
    const int nchunks = (BFGX_ABL2 == 1) ? 0 : (ne + csz - 1) / csz;
    WaveLds &L = wl[wid];
    unsigned long long npairs = 0;

    while (true) {
        int c = 0;
        if (lane == 0) c = atomicAdd(next_chunk, 1);
        c = __builtin_amdgcn_readfirstlane(c);
        if (c >= nchunks) break;

        k1_chunk<MODE, real, NP, PM>(tb, rowrecs, pairrecs, fbrecs, ea, eb, na, c * csz, min(csz, ne - c * csz), 1, i0, i1, nphi, wsh, wmask, PL, acc, L, rowtab, ringc,
                                     lane, npairs);
    }
    __syncthreads();

    if (MODE == MODE_COUNT) {
        if (lane == 0 && npairs) atomicAdd(pair_total, npairs);
        continue;
    }
    // ---- flush: every pixel of the tile is stored exactly once (plain, row-contiguous stores of [pixel][component])
    float om2 = 0.0f;
    for (int rr = wid; rr < i1 - i0; rr += kW2) {
        const int ring = i0 + rr;
        int64_t st, n64; bool shf;
        ring_info_small(h, ring, st, n64, shf);
        const TileRow &tr = rowtab[rr];
        const int n = (tr.ke - tr.ks) * NCOMP;
        ACC *dst = out + NCOMP * (st + tr.ks);
        const double *src = acc + (rr << wsh);
        const int rot = ((rr & 7) << wsh) >> 3;
        if (NCOMP == 3) {
            // one pixel per lane: three conflict-free LDS reads, one 3-component store (contiguous across the wave), and the
            // largest |offset|^2 of the tile as a by-product for the regrid (K2)
            struct alignas(sizeof(ACC)) Px { ACC c[3]; };
            Px *dpx = reinterpret_cast<Px *>(dst);
            for (int px = lane; px < tr.ke - tr.ks; px += kWave) {
                const int ix = (px + rot) & wmask;
                Px v;
                const double d0 = src[ix], d1 = src[PL + ix], d2 = src[2 * PL + ix];
                v.c[0] = (ACC)d0; v.c[1] = (ACC)d1; v.c[2] = (ACC)d2;
                dpx[px] = v;
                if (PM == 1) {
                    Px w;
                    w.c[0] = (ACC)(d0 - (double)v.c[0]); w.c[1] = (ACC)(d1 - (double)v.c[1]); w.c[2] = (ACC)(d2 - (double)v.c[2]);
                    reinterpret_cast<Px *>(out_lo + NCOMP * (st + tr.ks))[px] = w;
                }
                const float a = (float)v.c[0], b = (float)v.c[1], c = (float)v.c[2];
                om2 = fmaxf(om2, fma_(a, a, fma_(b, b, c * c)));
            }
        } else {
            for (int x = lane; x < n; x += kWave) dst[x] = (ACC)src[(x + rot) & wmask];
        }
    }
    if (MODE == MODE_OFFSETS && omax2 != nullptr) {      // (the array was zeroed with the binning counters; the wide pass may raise it)
#pragma unroll
        for (int sft = kWave >> 1; sft > 0; sft >>= 1) om2 = fmaxf(om2, __shfl_down(om2, sft, kWave));
        if (lane == 0 && om2 > 0.0f) atomicMax(omax2 + tile, __float_as_uint(om2 * 1.000001f));
    }
    __syncthreads();                                   // the LDS tile is reused by the next tile
    }
}


// ---------------------------------------------------------------------------------- the fluid kernel (round 4)
// The kernel above separates its tiles with workgroup barriers: 14 chunks over 8 waves leave two waves one chunk short, and every wave
// waits at the end-of-tile barrier for the slowest (19 % of the wave time by the shader clock), then for the flush, the zero-fill and
// the next tile's ring table.  Here ONE 1024-thread workgroup per CU (the same 16 waves and the same LDS as two workgroups of the
// kernel above) owns TWO tile slots and no wave ever waits for another at a tile boundary: a wave that finds no chunk left in tile k
// leaves it (one LDS counter) and goes on to tile k + 1 in the other slot.  The LAST wave to leave tile k flushes it on its own
// (2048 pixels: 32 rows of one trip each, zeroing the planes as it reads them), draws the workgroup's tile k + 2, builds its ring
// table and descriptor in the slot it has just emptied and publishes the slot's sequence number; a wave that reaches tile k + 2
// earlier than that sleeps on the number.  Waits are bounded: a wave that has slept 2^22 times raises *err and leaves, so the grid
// always drains.
// waves per workgroup of the fluid kernel: sixteen (four per SIMD at <= 128 VGPRs) for fp32 pair math; twelve for fp64 pair math, whose waves need
// ~155 VGPRs and 4.5 KB of LDS each: three per SIMD, against two for the barrier form (one 8-wave workgroup per CU: its LDS does not fit twice)
// (measured, config 2: 16 waves x 1 pair per lane 0.387 ms; 16 x 2: 0.410; 12 x 2 -- 132 VGPRs --: 0.431; 12 x 1: 0.444.  More than 16 waves would be
// a second workgroup per CU, which LDS does not hold.)
#ifndef BFGX_K1F_WAVES32
#define BFGX_K1F_WAVES32 16
#endif
#ifndef BFGX_K1F_CSZ
#define BFGX_K1F_CSZ kChunk2      // entries per chunk of a long list (<= kChunk2).  Measured, config 2: 10 / 12 / 14 / 16 entries 0.421 / 0.406 / 0.393 / 0.387 ms;
#endif                            // BFGX_CHUNK2 = 18 / 20 (5 KB more LDS: the last that fits): 0.387 / 0.387 (config 3: 1.002 -> 0.987)
#ifndef BFGX_K1F_NP
#define BFGX_K1F_NP 1
#endif
template <typename real> struct FluidWaves { static constexpr int n = sizeof(real) == 4 ? BFGX_K1F_WAVES32 : 12; };
#ifndef BFGX_CHUNKB
#define BFGX_CHUNKB 2
#endif
constexpr int kChunkB = BFGX_CHUNKB;  // entries per chunk of region B
#ifndef BFGX_K1F_SLEEP
#define BFGX_K1F_SLEEP 2              // s_sleep argument (x 64 clocks) of a wave that polls for its next tile slot
#endif
#ifndef BFGX_K1F_PROF
#define BFGX_K1F_PROF 0              // 1: shader-clock accounting of the fluid kernel's waves (variant builds only: scripts/k1f_prof.py)
#endif
#if BFGX_K1F_PROF
__device__ unsigned long long g_k1f_prof[8];       // wait for a slot, chunks, flush + refill, whole wave, flushes
#endif
struct alignas(16) FluidSlot {
    int32_t seq;                      // sequence number (per workgroup) of the tile this slot holds; -1: none yet
    int32_t tile;                     // < 0: no tile left
    int32_t next_chunk, done;
    int32_t i0, i1, nphi, na;
    int32_t ne, csz, nchunks, _pad;      // (_pad: the chunks of region B)
    const int32_t *ea, *eb;
    int32_t flushing;                 // 1 while the slot's tile is being flushed: waves waiting for the slot take row groups
    int32_t fl_next, fl_done;         // row groups handed out / finished
    uint32_t om2;                     // largest |offset|^2 of the tile (bits of a non-negative float)
    // the slot's NEXT tile, drawn and prepared while the current one is being worked on (stage_next), moved up by promote()
    int32_t staged;                   // sequence number of the staged tile (-1: none)
    int32_t nx_tile, nx_i0, nx_i1;
    int32_t nx_nphi, nx_na, nx_ne, nx_pad;
    const int32_t *nx_ea, *nx_eb;
#if BFGX_K1F_PROF
    uint32_t pf_maxchunk, pf_pad[3];
#endif
};

template <typename real>
__host__ __device__ inline size_t tile2f_lds_bytes(int BR, int W, int ncomp)
{
    size_t a = (size_t)ncomp * ((size_t)BR * W + kPlanePad) * sizeof(double);
    a = (a + 15) & ~(size_t)15;
    return 2 * a + sizeof(Wave2Lds<real>) * FluidWaves<real>::n + 2 * sizeof(FluidSlot) + 4 * (sizeof(TileRow) + sizeof(RingC2<real>)) * (size_t)BR;
}

template <int MODE, typename ACC, typename real, int PM = 0>
__global__ void __launch_bounds__(kWave * FluidWaves<real>::n, 1)
tile_scatter2f_kernel(Tab8T<real> tb, Hpx h, Tiling T, const RowRec *__restrict__ rowrecs,
                      const PairRecT<real> *__restrict__ pairrecs, const FbRec *__restrict__ fbrecs,
                      const int32_t *__restrict__ tile_start, const int32_t *__restrict__ cnt_a, const int32_t *__restrict__ cnt_b,
                      const int32_t *__restrict__ entries, int64_t capacity, const int32_t *__restrict__ entries_a, int cap_a, int cnt_pad,
                      ACC *__restrict__ out, ACC *__restrict__ out_lo, unsigned int *__restrict__ tile_counter, unsigned int *__restrict__ omax2,
                      int tile_lo, int tile_n, int32_t *__restrict__ err, const int32_t *__restrict__ form, int my_form)
{
    if (form != nullptr && *form != my_form) return;
    static_assert(MODE == MODE_OFFSETS || MODE == MODE_PAINT, "the census runs in tile_scatter2_kernel");
    static_assert(PM == 0 || (MODE == MODE_OFFSETS && sizeof(ACC) == 4 && sizeof(real) == 8), "PM = 1: fp64 pair math into split fp32 pix_offsets");
    constexpr int kWF = FluidWaves<real>::n;
    extern __shared__ __align__(16) unsigned char smem[];
    constexpr int NCOMP = (MODE == MODE_OFFSETS) ? 3 : 1;
    const int PL = T.BR * T.W + kPlanePad;
    const int wsh = __ffs(T.W) - 1, wmask = T.W - 1;
    const size_t plane_bytes = ((size_t)NCOMP * PL * sizeof(double) + 15) & ~(size_t)15;
    using WaveLds = Wave2Lds<real>;
    size_t off = 2 * plane_bytes;
    WaveLds *wl = reinterpret_cast<WaveLds *>(smem + off);
    off += sizeof(WaveLds) * kWF;
    FluidSlot *slots = reinterpret_cast<FluidSlot *>(smem + off);
    off += 2 * sizeof(FluidSlot);
    TileRow *rowtab_all = reinterpret_cast<TileRow *>(smem + off);
    off += 4 * sizeof(TileRow) * (size_t)T.BR;        // [slot][parity of the slot's tile count]: the live table and the staged one
    RingC2<real> *ringc_all = reinterpret_cast<RingC2<real> *>(smem + off);

    const int tid = threadIdx.x;
    const int lane = tid & (kWave - 1);
    const int wid = __builtin_amdgcn_readfirstlane(tid / kWave);
    const int ntodo = tile_n < 0 ? T.ntiles : tile_n;

    // draw_tile(): the workgroup's next tile from the device counter and what its descriptor needs from global memory
    struct Drawn { int tile, band, nphi, tj, na, ne; const int32_t *ea, *eb; };
    auto draw_tile = [&]() {
        Drawn d;
        d.tile = -1; d.band = 0; d.nphi = 1; d.tj = 0; d.na = 0; d.ne = 0; d.ea = entries; d.eb = entries;
        int tile_idx = 0;
        if (lane == 0) tile_idx = (int)atomicAdd(tile_counter, 1u);
        tile_idx = __builtin_amdgcn_readfirstlane(tile_idx);
        if (tile_idx < ntodo) {
            const int tile = tile_n < 0 ? T.tile_order[tile_idx] : tile_lo + tile_idx;
            d.tile = tile;
            d.band = T.tile_band[tile];
            d.nphi = T.band_nphi[d.band];
            d.tj = tile - T.band_tile0[d.band];
            // (the same list arithmetic as tile_scatter2_kernel: region A = the tile's fixed-capacity list, region B in the shared list)
            const int64_t e0 = tile_start[tile];
            const int na = entries_a ? min(cnt_a[(int64_t)tile * cnt_pad], cap_a) : cnt_a[tile];      // (direct placement: padded counters)
            d.ea = entries_a ? entries_a + (int64_t)tile * cap_a : entries + e0;
            const int64_t eb0 = e0 + (entries_a ? 0 : na);
            int64_t nb64 = cnt_b[tile];
            if (eb0 + nb64 > capacity) nb64 = capacity > eb0 ? capacity - eb0 : 0;
            if (!entries_a && e0 + na > capacity) nb64 = 0;
            d.na = na;
            d.ne = (!entries_a && e0 + na > capacity) ? (int)(capacity > e0 ? capacity - e0 : 0) : na + (int)nb64;
            d.eb = entries + eb0;
        }
        return d;
    };
    // ONE wave, right after it has published the slot's tile kk - 2: draw the slot's tile kk and prepare it beside the live one (ring table
    // in the other half of the slot's table pair, descriptor in the nx_ fields), so that the chain of dependent global loads (tile counter
    // -> tile order -> band -> list bounds) is over long before the live tile has been flushed
    auto stage_next = [&](int s, int kk) {
        FluidSlot &S = slots[s];
        const Drawn d = draw_tile();
        if (d.tile < 0) {
            if (lane == 0) { S.nx_tile = -1; S.nx_ne = 0; }
        } else {
            const int i0 = 1 + d.band * T.BR;
            const int i1 = min(i0 + T.BR, (int)(4 * h.nside));
            const int par = (kk >> 1) & 1;
            TileRow *rowtab = rowtab_all + (size_t)(2 * s + par) * T.BR;
            RingC2<real> *ringc = ringc_all + (size_t)(2 * s + par) * T.BR;
            if (lane < T.BR && i0 + lane < i1) {
                const int ring = i0 + lane;
                int64_t st, n64; bool shf;
                TileRow tr;
                ring_info_small(h, ring, st, n64, shf);
                ring_z_sth(h, ring, tr.z, tr.sth);
                tr.nr = (int)n64; tr.shifted = shf ? 1 : 0;
                tr.dphi = kTwoPi / (double)tr.nr;
                tr.fnr = (double)n64 * kInvTwoPi;
                tr.ks = tile_ks(d.tj, tr.nr, d.nphi); tr.ke = tile_ks(d.tj + 1, tr.nr, d.nphi);
                rowtab[lane] = tr;
                RingC2<real> rg;
                rg.sth = (real)tr.sth; rg.zf = (real)tr.z; rg.dphi = (real)tr.dphi; rg._pad = (real)0;
                ringc[lane] = rg;
            }
            if (lane == 0) {
                S.nx_tile = d.tile; S.nx_i0 = i0; S.nx_i1 = i1; S.nx_nphi = d.nphi; S.nx_na = d.na; S.nx_ne = d.ne;
                S.nx_ea = d.ea; S.nx_eb = d.eb;
            }
        }
        __builtin_amdgcn_wave_barrier();
        if (lane == 0) __hip_atomic_store(&S.staged, kk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    // the staged tile kk becomes the slot's live tile (the slot is empty: flushed, planes zeroed); false: the staging never arrived
    auto promote = [&](int s, int kk) -> bool {
        FluidSlot &S = slots[s];
        int spins = 0;
        while (__hip_atomic_load(&S.staged, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != kk) {
            __builtin_amdgcn_s_sleep(1);
            if (++spins > (1 << 22)) return false;
        }
        if (lane == 0) {
            S.tile = S.nx_tile; S.next_chunk = 0; S.done = 0;
            S.i0 = S.nx_i0; S.i1 = S.nx_i1; S.nphi = S.nx_nphi; S.na = S.nx_na;
            // entries per chunk: at most kChunk2, fewer when the list is short, so that every wave gets a chunk (a tile of a fine shell lists a
            // few halos of hundreds of pixels each); full chunks from half a workgroup's worth of them.  Smaller chunks for long lists, or
            // half chunks at the end of the list, measured slower.
            // Region B of the list (discs over more than kRefMax tiles: the large ones) goes out FIRST and in chunks of kChunkB entries, so that
            // the long items are not what the tile's waves finish on; then region A in chunks of csz.
            constexpr int kCh = K1Cfg<real>::chunk;
            S.ne = S.nx_ne; S.csz = S.nx_na >= kCh * kWF / 2 ? min(BFGX_K1F_CSZ, kCh) : max(1, min(kCh, (S.nx_na + kWF - 1) / kWF));
            // (painting: pairs are cheap, so a list of 128 - 384 entries is better cut into ~24 chunks than into 8 - 24 of sixteen entries --
            // config 3 lists 210 per tile: K3 1.058 -> 1.01 ms; the displacement kernel loses 2 % with the same rule)
            if (MODE == MODE_PAINT && S.nx_na >= kCh * kWF / 2 && S.nx_na < 24 * kCh) S.csz = max(1, min(kCh, (S.nx_na + 23) / 24));
            S._pad = (S.nx_ne - S.nx_na + kChunkB - 1) / kChunkB;                      // chunks of region B
            S.nchunks = S.nx_tile < 0 ? 0 : S._pad + (S.nx_na + S.csz - 1) / S.csz;
            S.ea = S.nx_ea; S.eb = S.nx_eb;
            __hip_atomic_store(&S.seq, kk, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
        return true;
    };

    {
        double *accz = reinterpret_cast<double *>(smem);
        const int nz = (int)(2 * plane_bytes / sizeof(double));
        for (int i = tid; i < nz; i += kWave * kWF) accz[i] = 0.0;
        if (tid < 2) { slots[tid].seq = -1; slots[tid].staged = -1; slots[tid].flushing = 0; slots[tid].fl_next = 1 << 30; slots[tid].fl_done = 0; slots[tid].om2 = 0u;
#if BFGX_K1F_PROF
            slots[tid].pf_maxchunk = 0u;
#endif
        }
    }
    __syncthreads();
    if (wid < 2) { stage_next(wid, wid); promote(wid, wid); stage_next(wid, wid + 2); }

    WaveLds &L = wl[wid];
    unsigned long long npairs = 0;
    bool ended = false;
#if BFGX_K1F_PROF
    unsigned long long pf_wait = 0, pf_chunk = 0, pf_flush = 0, pf_nfl = 0, pf_nchunk = 0, pf_maxsum = 0;
    const unsigned long long pf_t0 = __builtin_readcyclecounter();
#define PF_NOW() __builtin_readcyclecounter()
#else
#define PF_NOW() 0ull
#endif
    // The flush of a slot, in groups of four rows handed out by an LDS counter: to the last wave out of the tile, and to every wave that
    // reaches the slot's NEXT tile while the flush is still going on (instead of sleeping).  A group: twelve LDS reads in flight (one pixel per
    // lane and row; W <= 64), the planes zeroed as they are read, four stores; the group's largest |offset|^2 goes to the slot.
    auto flush_groups = [&](int s, int par) {
        FluidSlot &S = slots[s];
        const int i0 = __builtin_amdgcn_readfirstlane(S.i0), nrow = __builtin_amdgcn_readfirstlane(S.i1) - i0;
        const int ngroups = (nrow + 3) >> 2;
        double *acc = reinterpret_cast<double *>(smem + (size_t)s * plane_bytes);
        const TileRow *rowtab = rowtab_all + (size_t)(2 * s + par) * T.BR;
        // (at most ngroups + 1 draws per caller: the loop is bounded whatever the counter holds)
        for (int it = 0; it <= ngroups; ++it) {
            int g = 0;
            if (lane == 0) g = __hip_atomic_fetch_add(&S.fl_next, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            g = __builtin_amdgcn_readfirstlane(g);
            if (g >= ngroups) break;
            float om2 = 0.0f;
            struct alignas(sizeof(ACC)) Px3 { ACC c[3]; };
            using VT = typename std::conditional<PM == 1, double, ACC>::type;      // (PM = 0: converted as it is read, as before)
            VT vd[4][NCOMP];
            bool on[4];
            ACC *dp[4], *dl[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int rr = min(4 * g + q, nrow - 1);
                int64_t st, n64; bool shf;
                ring_info_small(h, i0 + rr, st, n64, shf);
                const TileRow &tr = rowtab[rr];
                const int ks = tr.ks, npx = tr.ke - tr.ks;
                double *src = acc + (rr << wsh);
                const int ix = (lane + (((rr & 7) << wsh) >> 3)) & wmask;
                on[q] = (4 * g + q < nrow) && lane < npx;
                dp[q] = out + NCOMP * (st + ks + lane);
                dl[q] = (PM == 1) ? out_lo + NCOMP * (st + ks + lane) : nullptr;
#pragma unroll
                for (int cc = 0; cc < NCOMP; ++cc) vd[q][cc] = (VT)0;
                if (on[q]) {
#pragma unroll
                    for (int cc = 0; cc < NCOMP; ++cc) { vd[q][cc] = (VT)src[cc * PL + ix]; src[cc * PL + ix] = 0.0; }
                }
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (NCOMP == 3) {
                    Px3 w;
                    w.c[0] = (ACC)vd[q][0]; w.c[1] = (ACC)vd[q][1]; w.c[2] = (ACC)vd[q][NCOMP - 1];
                    if (on[q]) *reinterpret_cast<Px3 *>(dp[q]) = w;
                    if (PM == 1) {           // the part of the fp64 sum that its fp32 value drops
                        Px3 wl;
                        wl.c[0] = (ACC)((double)vd[q][0] - (double)w.c[0]); wl.c[1] = (ACC)((double)vd[q][1] - (double)w.c[1]); wl.c[2] = (ACC)((double)vd[q][NCOMP - 1] - (double)w.c[2]);
                        if (on[q]) *reinterpret_cast<Px3 *>(dl[q]) = wl;
                    }
                    const float a = (float)w.c[0], b = (float)w.c[1], c = (float)w.c[2];
                    om2 = fmaxf(om2, fma_(a, a, fma_(b, b, c * c)));
                } else if (on[q]) *dp[q] = (ACC)vd[q][0];
            }
            if (MODE == MODE_OFFSETS) {
#pragma unroll
                for (int sft = kWave >> 1; sft > 0; sft >>= 1) om2 = fmaxf(om2, __shfl_down(om2, sft, kWave));
                if (lane == 0 && om2 > 0.0f) atomicMax(&S.om2, __float_as_uint(om2));
            }
            if (lane == 0) __hip_atomic_fetch_add(&S.fl_done, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    };

    for (int k = 0;; ++k) {
        const int s = k & 1;
        const unsigned long long pf_a = PF_NOW();
        (void)pf_a;
        FluidSlot &S = slots[s];
        // wait for the slot to hold tile k of this workgroup (it held tile k - 2 until the last wave out of that one had flushed it), and
        // help with that flush meanwhile
        {
            int spins = 0;
            while (__hip_atomic_load(&S.seq, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != k) {
                if (__hip_atomic_load(&S.flushing, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) == 1) flush_groups(s, ((k >> 1) & 1) ^ 1);
                __builtin_amdgcn_s_sleep(BFGX_K1F_SLEEP);
                if (++spins > (1 << 22)) {
                    if (lane == 0) atomicOr(err, 4);
                    return;
                }
            }
        }
        // The workgroup's two slots are refilled by different waves, so the draws of tiles k and k + 1 may come in either order: a slot
        // without a tile is passed through like an empty tile (its successor may still hold one); two of them in a row end the
        // sequence, because the tile counter only grows and every later draw follows one of those two.
        const unsigned long long pf_b = PF_NOW();
        (void)pf_b;
#if BFGX_K1F_PROF
        pf_wait += pf_b - pf_a;
#endif
        const int tile = __builtin_amdgcn_readfirstlane(S.tile);
        if (tile < 0 && ended) break;
        ended = tile < 0;
        const int i0 = __builtin_amdgcn_readfirstlane(S.i0), i1 = __builtin_amdgcn_readfirstlane(S.i1);
        const int nphi = __builtin_amdgcn_readfirstlane(S.nphi), na = __builtin_amdgcn_readfirstlane(S.na);
        const int ne = __builtin_amdgcn_readfirstlane(S.ne), nchb = __builtin_amdgcn_readfirstlane(S._pad);
        const int nchunks = (BFGX_ABL2 == 1 || tile < 0) ? 0 : __builtin_amdgcn_readfirstlane(S.nchunks);
        const int32_t *ea = reinterpret_cast<const int32_t *>(wave_uniform64((unsigned long long)S.ea));
        const int32_t *eb = reinterpret_cast<const int32_t *>(wave_uniform64((unsigned long long)S.eb));
        double *acc = reinterpret_cast<double *>(smem + (size_t)s * plane_bytes);
        const TileRow *rowtab = rowtab_all + (size_t)(2 * s + ((k >> 1) & 1)) * T.BR;
        const RingC2<real> *ringc = ringc_all + (size_t)(2 * s + ((k >> 1) & 1)) * T.BR;

        while (true) {
            int c = 0;
            if (lane == 0) c = atomicAdd(&S.next_chunk, 1);
            c = __builtin_amdgcn_readfirstlane(c);
            if (c >= nchunks) break;
#if BFGX_K1F_PROF
            const unsigned long long pf_c0 = PF_NOW();
#endif
            // region A is dealt out INTERLEAVED: chunk j takes the entries j, j + n, j + 2 n, ... (n = the number of chunks), so that whatever
            // order the catalog came in -- heaviest halos first (a chunk of sixteen heavy ones: K1 + 7 %), patch by patch (sixteen
            // neighbours adding to the same pixels: + 2 %) -- every chunk is a sample of the whole list
            const int ncha = nchunks - nchb;
            const int ebeg = c < nchb ? na + c * kChunkB : c - nchb;
            const int ecnt = c < nchb ? min(kChunkB, ne - ebeg) : (na - ebeg + ncha - 1) / ncha;
            const int estride = c < nchb ? 1 : ncha;
            k1_chunk<MODE, real, (sizeof(real) == 4 ? BFGX_K1F_NP : 1), PM>(tb, rowrecs, pairrecs, fbrecs, ea, eb, na, ebeg, ecnt, estride, i0, i1, nphi, wsh, wmask, PL, acc, L, rowtab, ringc,
                                    lane, npairs);
#if BFGX_K1F_PROF
            if (lane == 0) { const unsigned d = (unsigned)(PF_NOW() - pf_c0); atomicMax(&S.pf_maxchunk, d); pf_nchunk += 1; }
#endif
        }
#if BFGX_K1F_PROF
        const unsigned long long pf_c = PF_NOW();
        pf_chunk += pf_c - pf_b;
#endif
        // leave tile k; the last wave out starts the flush, takes part in it, and refills the slot when the last row group is done
        int prev = 0;
        if (lane == 0) prev = __hip_atomic_fetch_add(&S.done, 1, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_WORKGROUP);
        prev = __builtin_amdgcn_readfirstlane(prev);
        if (prev != kWF - 1) continue;
        if (tile < 0) {
            if (!promote(s, k + 2)) { if (lane == 0) atomicOr(err, 4); return; }
            stage_next(s, k + 4);
            continue;
        }
        // (no wave can still be inside flush_groups for this slot's previous tile: all sixteen have since entered and left tile k)
        if (lane == 0) {
            __hip_atomic_store(&S.fl_next, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&S.fl_done, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&S.om2, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&S.flushing, 1, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
#if BFGX_K1F_PROF
        if (lane == 0) { pf_maxsum += S.pf_maxchunk; S.pf_maxchunk = 0u; }
#endif
        flush_groups(s, (k >> 1) & 1);
        {
            const int ngroups = (i1 - i0 + 3) >> 2;
            int spins = 0;
            while (__hip_atomic_load(&S.fl_done, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_WORKGROUP) != ngroups) {
                __builtin_amdgcn_s_sleep(1);
                if (++spins > (1 << 22)) {
                    if (lane == 0) atomicOr(err, 4);
                    return;
                }
            }
        }
        if (lane == 0) {
            __hip_atomic_store(&S.flushing, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            __hip_atomic_store(&S.fl_next, 1 << 30, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);        // a wave that still believes the flush is on draws no group of the NEXT tile (the slot's row count changes below)
            const uint32_t ob = __hip_atomic_load(&S.om2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
            if (MODE == MODE_OFFSETS && omax2 != nullptr && ob != 0u) atomicMax(omax2 + tile, __float_as_uint(__uint_as_float(ob) * 1.000001f));
        }
        if (!promote(s, k + 2)) { if (lane == 0) atomicOr(err, 4); return; }
        stage_next(s, k + 4);
#if BFGX_K1F_PROF
        pf_flush += PF_NOW() - pf_c; pf_nfl += 1;
#endif
    }
#if BFGX_K1F_PROF
    if (lane == 0) {
        atomicAdd(&g_k1f_prof[0], pf_wait); atomicAdd(&g_k1f_prof[1], pf_chunk); atomicAdd(&g_k1f_prof[2], pf_flush);
        atomicAdd(&g_k1f_prof[3], PF_NOW() - pf_t0); atomicAdd(&g_k1f_prof[4], pf_nfl); atomicAdd(&g_k1f_prof[5], 1ull);
        atomicAdd(&g_k1f_prof[6], pf_nchunk); atomicAdd(&g_k1f_prof[7], pf_maxsum);
    }
#endif
}

}  // namespace bfgx
