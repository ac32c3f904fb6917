/*
 * bfg_oracle.c -- CPU restatement (fp64, scalar C99) of BaryonForge's per-halo
 * HEALPix-shell hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under baryonification_amd/ may import, link
 * or call this file.  It exists so that tests/, __graft_entry__.smoke() and the
 * cpu_baseline leg of bench.py can check / time the HIP path against an
 * independent statement of the reference algorithm.
 *
 * Reference lines restated (relative to /root/reference):
 *   BaryonForge/Runners/HealpixRunner.py:291-331   BaryonifyShell halo loop   -> bfgo_baryonify_offsets
 *   BaryonForge/Runners/HealpixRunner.py:333-341   displaced-pixel regrid     -> bfgo_regrid
 *   BaryonForge/Runners/HealpixRunner.py:13-67     regrid_pixels_hpix         -> (inner loop of bfgo_regrid)
 *   BaryonForge/Runners/HealpixRunner.py:418-445   PaintProfilesShell loop    -> bfgo_paint
 *   BaryonForge/Profiles/BaryonCorrection.py:324-390  _readout                -> rgi_eval + eps mask
 *   BaryonForge/utils/Tabulate.py:246-294, 569-621    _readout (exp of RGI)   -> rgi_eval + exp
 *
 * Third-party arithmetic the reference calls and that is NOT under /root/reference
 * (healpy / healpix_cxx, un-pinned in setup.py:21; scipy RegularGridInterpolator):
 * restated here from the published HEALPix RING algorithms (Gorski et al. 2005;
 * healpix_cxx T_Healpix_Base: pix2loc, ring_above, get_ring_info2,
 * query_disc_internal with fact=0, get_interpol) and from scipy's
 * RegularGridInterpolator(method="linear", bounds_error=False, fill_value=nan).
 * The scipy part is pinned by tests against scipy itself; the healpy part is pinned
 * only by definition-level brute force (tests/test_oracle_healpix.py): healpy is not
 * installable here -> "parity unpinned" for healpy.
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef int64_t i64;

static const double PI      = 3.141592653589793238462643383279502884197;
static const double TWOPI   = 6.283185307179586476925286766559005768394;
static const double HALFPI  = 1.570796326794896619231321691639751442099;
static const double INV_TWOPI = 1.0 / 6.283185307179586476925286766559005768394;
static const double TWOTHIRD = 2.0 / 3.0;

/* ------------------------------------------------------------------ HEALPix RING */

typedef struct {
    i64 nside, npix, ncap;
    double fact1, fact2;
} hpx_t;

static hpx_t hpx_make(i64 nside)
{
    hpx_t h;
    h.nside = nside;
    h.npix  = 12 * nside * nside;
    h.ncap  = 2 * nside * (nside - 1);
    h.fact2 = 4.0 / (double)h.npix;
    h.fact1 = (double)(nside << 1) * h.fact2;
    return h;
}

static i64 hpx_isqrt(i64 v) { return (i64)sqrt((double)v + 0.5); }

/* healpix_cxx T_Healpix_Base::ring_above */
static i64 hpx_ring_above(const hpx_t *h, double z)
{
    double az = fabs(z);
    if (az <= TWOTHIRD) return (i64)((double)h->nside * (2.0 - 1.5 * z));
    i64 iring = (i64)((double)h->nside * sqrt(3.0 * (1.0 - az)));
    return (z > 0) ? iring : 4 * h->nside - iring - 1;
}

/* healpix_cxx get_ring_info_small */
static void hpx_ring_info_small(const hpx_t *h, i64 ring, i64 *startpix, i64 *ringpix, int *shifted)
{
    if (ring < h->nside) {
        *shifted = 1; *ringpix = 4 * ring; *startpix = 2 * ring * (ring - 1);
    } else if (ring < 3 * h->nside) {
        *shifted = (((ring - h->nside) & 1) == 0);
        *ringpix = 4 * h->nside;
        *startpix = h->ncap + (ring - h->nside) * (*ringpix);
    } else {
        i64 nr = 4 * h->nside - ring;
        *shifted = 1; *ringpix = 4 * nr; *startpix = h->npix - 2 * nr * (nr + 1);
    }
}

/* healpix_cxx get_ring_info2 (also returns the ring colatitude) */
static void hpx_ring_info2(const hpx_t *h, i64 ring, i64 *startpix, i64 *ringpix, double *theta, int *shifted)
{
    i64 northring = (ring > 2 * h->nside) ? 4 * h->nside - ring : ring;
    if (northring < h->nside) {
        double tmp = (double)(northring * northring) * h->fact2;
        double costheta = 1.0 - tmp;
        double sintheta = sqrt(tmp * (2.0 - tmp));
        *theta = atan2(sintheta, costheta);
        *ringpix = 4 * northring;
        *shifted = 1;
        *startpix = 2 * northring * (northring - 1);
    } else {
        *theta = acos((double)(2 * h->nside - northring) * h->fact1);
        *ringpix = 4 * h->nside;
        *shifted = (((northring - h->nside) & 1) == 0);
        *startpix = h->ncap + (northring - h->nside) * (*ringpix);
    }
    if (northring != ring) {
        *theta = PI - *theta;
        *startpix = h->npix - *startpix - *ringpix;
    }
}

/* healpix_cxx ring2z */
static double hpx_ring2z(const hpx_t *h, i64 ring)
{
    if (ring < h->nside) return 1.0 - (double)(ring * ring) * h->fact2;
    if (ring <= 3 * h->nside) return (double)(2 * h->nside - ring) * h->fact1;
    ring = 4 * h->nside - ring;
    return (double)(ring * ring) * h->fact2 - 1.0;
}

/* healpix_cxx pix2loc (RING) followed by loc2vec */
static void hpx_pix2vec(const hpx_t *h, i64 pix, double v[3])
{
    double z, phi, sth = 0.0;
    int have_sth = 0;
    if (pix < h->ncap) {
        i64 iring = (1 + hpx_isqrt(1 + 2 * pix)) >> 1;
        i64 iphi  = (pix + 1) - 2 * iring * (iring - 1);
        double tmp = (double)(iring * iring) * h->fact2;
        z = 1.0 - tmp;
        if (z > 0.99) { sth = sqrt(tmp * (2.0 - tmp)); have_sth = 1; }
        phi = ((double)iphi - 0.5) * HALFPI / (double)iring;
    } else if (pix < (h->npix - h->ncap)) {
        i64 nl4 = 4 * h->nside;
        i64 ip  = pix - h->ncap;
        i64 tmp = ip / nl4;
        i64 iring = tmp + h->nside, iphi = ip - nl4 * tmp + 1;
        double fodd = ((iring + h->nside) & 1) ? 1.0 : 0.5;
        z = (double)(2 * h->nside - iring) * h->fact1;
        phi = ((double)iphi - fodd) * PI * 0.75 * h->fact1;
    } else {
        i64 ip = h->npix - pix;
        i64 iring = (1 + hpx_isqrt(2 * ip - 1)) >> 1;
        i64 iphi  = 4 * iring + 1 - (ip - 2 * iring * (iring - 1));
        double tmp = (double)(iring * iring) * h->fact2;
        z = tmp - 1.0;
        if (z < -0.99) { sth = sqrt(tmp * (2.0 - tmp)); have_sth = 1; }
        phi = ((double)iphi - 0.5) * HALFPI / (double)iring;
    }
    if (!have_sth) sth = sqrt((1.0 - z) * (1.0 + z));
    v[0] = sth * cos(phi);
    v[1] = sth * sin(phi);
    v[2] = z;
}

/* healpy lonlat2thetaphi + ang2vec */
static void hpx_lonlat2thetaphi(double lon, double lat, double *theta, double *phi)
{
    *theta = HALFPI - lat * (PI / 180.0);
    *phi   = lon * (PI / 180.0);
}

static void hpx_ang2vec(double theta, double phi, double v[3])
{
    double st = sin(theta);
    v[0] = st * cos(phi);
    v[1] = st * sin(phi);
    v[2] = cos(theta);
}

/* healpy vec2ang(lonlat=True): theta = arccos(z/|v|), phi = arctan2(y,x) wrapped to [0,2pi),
 * then thetaphi2lonlat: lon = degrees(phi), lat = 90 - degrees(theta) */
static void hpx_vec2lonlat(const double v[3], double *lon, double *lat)
{
    double dnorm = sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    double theta = acos(v[2] / dnorm);
    double phi   = atan2(v[1], v[0]);
    if (phi < 0) phi += TWOPI;
    *lon = phi * (180.0 / PI);
    *lat = 90.0 - theta * (180.0 / PI);
}

/* healpix_cxx get_interpol (RING) */
static void hpx_get_interpol(const hpx_t *h, double theta, double phi, i64 pix[4], double wgt[4])
{
    double z = cos(theta);
    i64 ir1 = hpx_ring_above(h, z);
    i64 ir2 = ir1 + 1;
    double theta1 = 0, theta2 = 0, w1, tmp, dphi;
    i64 sp, nr, i1, i2;
    int shift;
    if (ir1 > 0) {
        hpx_ring_info2(h, ir1, &sp, &nr, &theta1, &shift);
        dphi = TWOPI / (double)nr;
        tmp = (phi / dphi - 0.5 * shift);
        i1 = (tmp < 0) ? (i64)tmp - 1 : (i64)tmp;
        w1 = (phi - ((double)i1 + 0.5 * shift) * dphi) / dphi;
        i2 = i1 + 1;
        if (i1 < 0) i1 += nr;
        if (i2 >= nr) i2 -= nr;
        pix[0] = sp + i1; pix[1] = sp + i2;
        wgt[0] = 1 - w1; wgt[1] = w1;
    }
    if (ir2 < 4 * h->nside) {
        hpx_ring_info2(h, ir2, &sp, &nr, &theta2, &shift);
        dphi = TWOPI / (double)nr;
        tmp = (phi / dphi - 0.5 * shift);
        i1 = (tmp < 0) ? (i64)tmp - 1 : (i64)tmp;
        w1 = (phi - ((double)i1 + 0.5 * shift) * dphi) / dphi;
        i2 = i1 + 1;
        if (i1 < 0) i1 += nr;
        if (i2 >= nr) i2 -= nr;
        pix[2] = sp + i1; pix[3] = sp + i2;
        wgt[2] = 1 - w1; wgt[3] = w1;
    }
    if (ir1 == 0) {
        double wtheta = theta / theta2;
        wgt[2] *= wtheta; wgt[3] *= wtheta;
        double fac = (1 - wtheta) * 0.25;
        wgt[0] = fac; wgt[1] = fac; wgt[2] += fac; wgt[3] += fac;
        pix[0] = (pix[2] + 2) & 3;
        pix[1] = (pix[3] + 2) & 3;
    } else if (ir2 == 4 * h->nside) {
        double wtheta = (theta - theta1) / (PI - theta1);
        wgt[0] *= (1 - wtheta); wgt[1] *= (1 - wtheta);
        double fac = wtheta * 0.25;
        wgt[0] += fac; wgt[1] += fac; wgt[2] = fac; wgt[3] = fac;
        pix[2] = ((pix[0] + 2) & 3) + h->npix - 4;
        pix[3] = ((pix[1] + 2) & 3) + h->npix - 4;
    } else {
        double wtheta = (theta - theta1) / (theta2 - theta1);
        wgt[0] *= (1 - wtheta); wgt[1] *= (1 - wtheta);
        wgt[2] *= wtheta; wgt[3] *= wtheta;
    }
}

/* growable pixel list */
typedef struct { i64 *p; i64 n, cap; } pixlist_t;

static void pl_append_range(pixlist_t *l, i64 a, i64 b) /* [a,b) */
{
    if (b <= a) return;
    i64 need = l->n + (b - a);
    if (need > l->cap) {
        i64 c = l->cap ? l->cap : 1024;
        while (c < need) c *= 2;
        l->p = (i64 *)realloc(l->p, (size_t)c * sizeof(i64));
        l->cap = c;
    }
    for (i64 i = a; i < b; ++i) l->p[l->n++] = i;
}

/* healpix_cxx query_disc_internal, RING, fact = 0 (inclusive=False):
 * all pixels whose centre lies within `radius` of the direction `vec`. */
static void hpx_query_disc(const hpx_t *h, const double vec[3], double radius, pixlist_t *out)
{
    out->n = 0;
    double theta = atan2(sqrt(vec[0] * vec[0] + vec[1] * vec[1]), vec[2]);
    double phi = (vec[0] == 0.0 && vec[1] == 0.0) ? 0.0 : atan2(vec[1], vec[0]);
    if (phi < 0) phi += TWOPI;

    double rsmall = radius, rbig = radius;
    if (rsmall >= PI) { pl_append_range(out, 0, h->npix); return; }
    if (rbig > PI) rbig = PI;
    double cosrbig = cos(rbig);
    double z0 = cos(theta);
    double xa = 1.0 / sqrt((1.0 - z0) * (1.0 + z0));

    double rlat1 = theta - rsmall;
    double zmax = cos(rlat1);
    i64 irmin = hpx_ring_above(h, zmax) + 1;
    if ((rlat1 <= 0) && (irmin > 1)) { /* north pole in the disc */
        i64 sp, rp; int dummy;
        hpx_ring_info_small(h, irmin - 1, &sp, &rp, &dummy);
        pl_append_range(out, 0, sp + rp);
    }
    double rlat2 = theta + rsmall;
    double zmin = cos(rlat2);
    i64 irmax = hpx_ring_above(h, zmin);

    for (i64 iz = irmin; iz <= irmax; ++iz) {
        double z = hpx_ring2z(h, iz);
        double x = (cosrbig - z * z0) * xa;
        double ysq = 1.0 - z * z - x * x;
        double dphi = (ysq <= 0) ? 0.0 : atan2(sqrt(ysq), x);
        if (dphi > 0) {
            i64 nr, ipix1; int shifted;
            hpx_ring_info_small(h, iz, &ipix1, &nr, &shifted);
            double shift = shifted ? 0.5 : 0.0;
            i64 ipix2 = ipix1 + nr - 1;
            i64 ip_lo = (i64)floor((double)nr * INV_TWOPI * (phi - dphi) - shift) + 1;
            i64 ip_hi = (i64)floor((double)nr * INV_TWOPI * (phi + dphi) - shift);
            if (ip_hi >= nr) { ip_lo -= nr; ip_hi -= nr; }
            if (ip_lo < 0) {
                pl_append_range(out, ipix1, ipix1 + ip_hi + 1);
                pl_append_range(out, ipix1 + ip_lo + nr, ipix2 + 1);
            } else {
                pl_append_range(out, ipix1 + ip_lo, ipix1 + ip_hi + 1);
            }
        }
    }
    if ((rlat2 >= PI) && (irmax + 1 < 4 * h->nside)) { /* south pole in the disc */
        i64 sp, rp; int dummy;
        hpx_ring_info_small(h, irmax + 1, &sp, &rp, &dummy);
        pl_append_range(out, sp, h->npix);
    }
}

/* ------------------------------------------------ scipy RegularGridInterpolator(linear) */

#define BFGO_MAXDIM 8

typedef struct {
    int ndim;                     /* 3 + number of extra parameter axes */
    int n[BFGO_MAXDIM];           /* points per axis */
    const double *axis[BFGO_MAXDIM];
    const double *values;         /* C-order [n0][n1]...[n_{ndim-1}] */
} rgi_t;

/* scipy _rgi_cython.find_indices: largest i with g[i] <= x, clipped to [0, n-2] */
static int rgi_find(const double *g, int n, double x)
{
    int lo = 0, hi = n - 1;           /* invariant: g[lo] <= x < g[hi] when in range */
    if (!(x >= g[0])) return 0;
    if (x >= g[n - 1]) return n - 2;
    while (hi - lo > 1) {
        int mid = (lo + hi) >> 1;
        if (x >= g[mid]) lo = mid; else hi = mid;
    }
    return lo;
}

/* scipy RegularGridInterpolator._evaluate_linear with bounds_error=False, fill_value=nan.
 * Corner order = itertools.product over dims of (i, i+1); weight = prod of per-dim weights
 * multiplied left to right starting from 1.0; value accumulated starting from 0.0. */
static double rgi_eval(const rgi_t *t, const double *x)
{
    int idx[BFGO_MAXDIM];
    double nd[BFGO_MAXDIM];
    for (int d = 0; d < t->ndim; ++d) {
        const double *g = t->axis[d];
        int n = t->n[d];
        if (x[d] != x[d]) return NAN;
        if (x[d] < g[0] || x[d] > g[n - 1]) return NAN;
        int i = rgi_find(g, n, x[d]);
        idx[d] = i;
        nd[d]  = (x[d] - g[i]) / (g[i + 1] - g[i]);
    }
    i64 stride[BFGO_MAXDIM];
    stride[t->ndim - 1] = 1;
    for (int d = t->ndim - 2; d >= 0; --d) stride[d] = stride[d + 1] * t->n[d + 1];
    double value = 0.0;
    int ncorner = 1 << t->ndim;
    for (int c = 0; c < ncorner; ++c) {
        double w = 1.0;
        i64 off = 0;
        for (int d = 0; d < t->ndim; ++d) {
            int bit = (c >> (t->ndim - 1 - d)) & 1;  /* first dim varies slowest, as itertools.product */
            w = w * (bit ? nd[d] : (1.0 - nd[d]));
            off += (i64)(idx[d] + bit) * stride[d];
        }
        value = value + t->values[off] * w;
    }
    return value;
}

static rgi_t rgi_make(int ndim, const int *n, const double *const *axes, const double *values)
{
    rgi_t t;
    t.ndim = ndim;
    for (int d = 0; d < ndim; ++d) { t.n[d] = n[d]; t.axis[d] = axes[d]; }
    t.values = values;
    return t;
}

/* ---------------------------------------------------------------- exported API */

/* geometry probes used by the oracle's own tests and by the refshim cross-checks */
void bfgo_pix2vec(i64 nside, i64 n, const i64 *pix, double *vec /*[n][3]*/)
{
    hpx_t h = hpx_make(nside);
    for (i64 i = 0; i < n; ++i) hpx_pix2vec(&h, pix[i], vec + 3 * i);
}

void bfgo_ang2vec_lonlat(i64 n, const double *lon, const double *lat, double *vec)
{
    for (i64 i = 0; i < n; ++i) {
        double th, ph;
        hpx_lonlat2thetaphi(lon[i], lat[i], &th, &ph);
        hpx_ang2vec(th, ph, vec + 3 * i);
    }
}

void bfgo_vec2ang_lonlat(i64 n, const double *vec, double *lon, double *lat)
{
    for (i64 i = 0; i < n; ++i) hpx_vec2lonlat(vec + 3 * i, lon + i, lat + i);
}

/* returns number of pixels; writes at most cap of them (ascending RING order) */
i64 bfgo_query_disc(i64 nside, const double *vec, double radius, i64 *out, i64 cap)
{
    hpx_t h = hpx_make(nside);
    pixlist_t l = {0, 0, 0};
    hpx_query_disc(&h, vec, radius, &l);
    i64 n = l.n;
    for (i64 i = 0; i < n && i < cap; ++i) out[i] = l.p[i];
    free(l.p);
    return n;
}

void bfgo_get_interp_weights_lonlat(i64 nside, i64 n, const double *lon, const double *lat,
                                    i64 *pix /*[n][4]*/, double *wgt /*[n][4]*/)
{
    hpx_t h = hpx_make(nside);
    for (i64 i = 0; i < n; ++i) {
        double th, ph;
        hpx_lonlat2thetaphi(lon[i], lat[i], &th, &ph);
        hpx_get_interpol(&h, th, ph, pix + 4 * i, wgt + 4 * i);
    }
}

double bfgo_rgi_eval(int ndim, const int *n, const double *const *axes, const double *values, const double *x)
{
    rgi_t t = rgi_make(ndim, n, axes, values);
    return rgi_eval(&t, x);
}

/*
 * BaryonifyShell halo loop, HealpixRunner.py:291-331.
 *   per-halo inputs (computed by the caller exactly as the reference does, lines 293-297
 *   and BaryonCorrection.py:370):
 *     M[j], a[j] = 1/(1+z_j), R[j] = mass_def.get_radius(cosmo, M_j, a_j) [physical Mpc],
 *     D[j] = D_a(z_j) [physical Mpc], Rmod[j] = model.mass_def.get_radius(model.cosmo, M_j, a_j)/a_j,
 *     lnz1[j] = np.log(1/a_j), lnM[j] = np.log(M_j): the table coordinates of BaryonCorrection.py:364, :369 evaluated by
 *     the caller's numpy (README.md:78-80 puts table edges exactly on the catalog's min/max, so the last bit of these logs
 *     decides whether an edge halo reads NaN); NULL -> libm log here,
 *     extra[k][j] = cat[p_keys[k]][j]
 *   table: displacement table (ndim = 3 + nextra), axes ln(1+z), ln M, ln r | ln(r/Rdelta), params
 *   out: pix_offsets [npix][3], ACCUMULATED INTO (caller zero-initialises), float64
 * Returns total number of (halo, pixel) pairs visited; npairs_per_halo (optional) gets counts.
 */
i64 bfgo_baryonify_offsets(i64 nside, i64 nhalo,
                           const double *ra, const double *dec, const double *M,
                           const double *a, const double *R, const double *D, const double *Rmod,
                           const double *lnz1, const double *lnM,
                           int nextra, const double *const *extra,
                           int ndim, const int *tn, const double *const *taxes, const double *tvalues,
                           int rdelta_sampling, double eps_runner, double eps_model,
                           double *pix_offsets, i64 *npairs_per_halo)
{
    hpx_t h = hpx_make(nside);
    rgi_t t = rgi_make(ndim, tn, taxes, tvalues);
    pixlist_t l = {0, 0, 0};
    i64 total = 0;
    (void)nextra;
    for (i64 j = 0; j < nhalo; ++j) {
        double M_j = M[j], a_j = a[j], R_j = R[j], D_j = D[j];
        double th, ph, vec_j[3];
        hpx_lonlat2thetaphi(ra[j], dec[j], &th, &ph);
        hpx_ang2vec(th, ph, vec_j);

        double radius = R_j * eps_runner / D_j;                       /* :305 */
        hpx_query_disc(&h, vec_j, radius, &l);                        /* :306 */
        i64 fb[4]; double fw[4];
        const i64 *pixind = l.p; i64 npx = l.n;
        if (npx < 4) {                                                /* :309-310 */
            hpx_get_interpol(&h, th, ph, fb, fw);
            pixind = fb; npx = 4;
        }
        if (npairs_per_halo) npairs_per_halo[j] = npx;
        total += npx;

        double x[BFGO_MAXDIM];
        x[0] = lnz1 ? lnz1[j] : log(1.0 / a_j);                       /* BaryonCorrection.py:364: np.log(1/a), as the caller's numpy gives it */
        x[1] = lnM ? lnM[j] : log(M_j);                               /* :369 */
        for (int k = 3; k < ndim; ++k) x[k] = extra[k - 3][j];
        double Rc = Rmod[j];                                          /* :370 */

        for (i64 i = 0; i < npx; ++i) {
            i64 p = pixind[i];
            double vec[3], pos[3], diff[3];
            hpx_pix2vec(&h, p, vec);                                  /* :312 */
            for (int c = 0; c < 3; ++c) {
                double pos_j = vec_j[c] * D_j;                        /* :314 */
                pos[c] = vec[c] * D_j;                                /* :315 */
                diff[c] = pos[c] - pos_j;                             /* :316 */
            }
            double r_sep = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]); /* :317 */
            double r_com = r_sep / a_j;                               /* :321 */
            x[2] = rdelta_sampling ? (log(r_com) - log(Rc)) : log(r_com);   /* :365, :378 */
            double d = rgi_eval(&t, x);                               /* :376/379 */
            if (!(r_com < eps_model * Rc)) d = 0.0;                   /* :381-382 */
            d = d * a_j;                                              /* :321 */
            double nw[3], nrm2 = 0.0;
            for (int c = 0; c < 3; ++c) {
                double o = d * (diff[c] / r_sep);                     /* :322 */
                if (!isfinite(o)) o = 0.0;                            /* :323 */
                nw[c] = pos[c] + o;                                   /* :326 */
            }
            nrm2 = nw[0] * nw[0] + nw[1] * nw[1] + nw[2] * nw[2];
            double nrm = sqrt(nrm2);
            for (int c = 0; c < 3; ++c)
                pix_offsets[3 * p + c] += nw[c] / nrm - vec[c];       /* :327-331 */
        }
    }
    free(l.p);
    return total;
}

/*
 * Post-loop regrid, HealpixRunner.py:333-341 + regrid_pixels_hpix :60-64.
 * new_map must be zero-initialised by the caller ([npix], float64).
 */
void bfgo_regrid_range(i64 nside, i64 p0, i64 p1, const double *orig_map, const double *pix_offsets, double *new_map);

void bfgo_regrid(i64 nside, const double *orig_map, const double *pix_offsets, double *new_map)
{
    bfgo_regrid_range(nside, 0, 12 * nside * nside, orig_map, pix_offsets, new_map);
}

/* source pixels [p0, p1) only (lets a caller spread the map over threads, each with its own new_map) */
void bfgo_regrid_range(i64 nside, i64 p0, i64 p1, const double *orig_map, const double *pix_offsets, double *new_map)
{
    hpx_t h = hpx_make(nside);
    for (i64 p = p0; p < p1; ++p) {
        if (!(orig_map[p] > 0)) continue;                             /* :335 */
        double v[3], lon, lat, th, ph, w[4];
        i64 cp[4];
        hpx_pix2vec(&h, p, v);
        for (int c = 0; c < 3; ++c) v[c] += pix_offsets[3 * p + c];   /* :333 */
        hpx_vec2lonlat(v, &lon, &lat);                                /* :334 */
        hpx_lonlat2thetaphi(lon, lat, &th, &ph);                      /* :337 (lonlat=True) */
        hpx_get_interpol(&h, th, ph, cp, w);
        for (int j = 0; j < 4; ++j) new_map[cp[j]] += w[j] * orig_map[p];   /* :64 */
    }
}

/*
 * PaintProfilesShell halo loop, HealpixRunner.py:418-445, for a Tabulated/ParamTabulated
 * profile: Paint = exp(RGI(ln table)(ln(1/a), ln M, ln r, params)), non-finite -> 0.
 * `tvalues` holds the LOG of raw_input_2D (Tabulate.py:238, :561).
 * new_map is accumulated into (caller zero-initialises).
 */
i64 bfgo_paint(i64 nside, i64 nhalo,
               const double *ra, const double *dec, const double *M,
               const double *a, const double *R, const double *D,
               const double *lnz1, const double *lnM,
               int nextra, const double *const *extra,
               int ndim, const int *tn, const double *const *taxes, const double *tvalues,
               double eps_runner, double *new_map, i64 *npairs_per_halo)
{
    hpx_t h = hpx_make(nside);
    rgi_t t = rgi_make(ndim, tn, taxes, tvalues);
    pixlist_t l = {0, 0, 0};
    i64 total = 0;
    (void)nextra;
    for (i64 j = 0; j < nhalo; ++j) {
        double a_j = a[j], D_j = D[j];
        double th, ph, vec_j[3];
        hpx_lonlat2thetaphi(ra[j], dec[j], &th, &ph);
        hpx_ang2vec(th, ph, vec_j);
        double radius = R[j] * eps_runner / D_j;                      /* :431 */
        hpx_query_disc(&h, vec_j, radius, &l);                        /* :432 */
        if (npairs_per_halo) npairs_per_halo[j] = l.n;
        total += l.n;
        double x[BFGO_MAXDIM];
        x[0] = lnz1 ? lnz1[j] : log(1.0 / a_j);                       /* Tabulate.py:279: np.log(1/a), as the caller's numpy gives it */
        x[1] = lnM ? lnM[j] : log(M[j]);                              /* :283 */
        for (int k = 3; k < ndim; ++k) x[k] = extra[k - 3][j];
        for (i64 i = 0; i < l.n; ++i) {
            i64 p = l.p[i];
            double vec[3], diff[3];
            hpx_pix2vec(&h, p, vec);
            for (int c = 0; c < 3; ++c) diff[c] = vec[c] * D_j - vec_j[c] * D_j;   /* :435-437 */
            double r_sep = sqrt(diff[0] * diff[0] + diff[1] * diff[1] + diff[2] * diff[2]);
            x[2] = log(r_sep / a_j);                                  /* :441, Tabulate.py:280 */
            double paint = exp(rgi_eval(&t, x));                      /* Tabulate.py:285-286 */
            if (!isfinite(paint)) paint = 0.0;                        /* :442 */
            new_map[p] += paint;                                      /* :445 */
        }
    }
    free(l.p);
    return total;
}

/* ====================================================================== regular-grid path
 * Map2DRunner.py: regrid_pixels_2D :14-83, regrid_pixels_3D :86-163, BaryonifyGrid.process :431-607,
 * PaintProfilesGrid.process :676-817; io.py ParticleSnapshot.make_map :622-670 (np.histogramdd).
 * Conventions of the reference kept as they are: meshgrid(indexing='xy') pairs the FIRST array axis of a cutout
 * with the "y" coordinate (x[i] + dy) and the SECOND with "x" (x[j] + dx), while the first axis is indexed around
 * x_cen; linspace(-N/2, N/2, N) * res spaces the cutout samples by N/(N-1) pixels.
 */

/* Python's float % (CPython float_rem; numba follows it) */
static double py_fmod(double x, double n)
{
    double m = fmod(x, n);
    if (m != 0.0) { if ((n < 0) != (m < 0)) m += n; }
    else m = copysign(0.0, n);
    return m;
}

static double dmin(double a, double b) { return a < b ? a : b; }
static double dmax(double a, double b) { return a > b ? a : b; }

/* overlap of the unit cell [c, c+1] with [s, e], including the two periodic images (Map2DRunner.py:70-78) */
static double cell_overlap(i64 c, double s, double e, double N)
{
    double d = dmin((double)(c + 1), e) - dmax((double)c, s);
    if (d < 0) d = dmin((double)(c + 1), e + N) - dmax((double)c, s + N);
    if (d < 0) d = dmin((double)(c + 1), e - N) - dmax((double)c, s - N);
    return d;
}

static i64 wrap_cell(i64 c, i64 N)
{
    if (c < 0) c += N;                    /* :61-62 */
    if (c + 1 > N) c = c % N;
    return c;
}

/* regrid_pixels_2D / regrid_pixels_3D: `grid` [N]^ndim is accumulated into; pos [n][ndim]; val [n] */
void bfgo_regrid_pixels(int ndim, i64 N, i64 n, const double *pos, const double *val, double *grid)
{
    const double Nf = (double)N;
    for (i64 p = 0; p < n; ++p) {
        double xs = py_fmod(pos[ndim * p + 0], Nf), ys = py_fmod(pos[ndim * p + 1], Nf);
        double zs = ndim == 3 ? py_fmod(pos[ndim * p + 2], Nf) : 0.0;
        double xe = xs + 1, ye = ys + 1, ze = zs + 1;
        const int bound = 2;
        i64 x_min = (i64)xs - bound, x_max = (i64)xe + bound;
        i64 y_min = (i64)ys - bound, y_max = (i64)ye + bound;
        i64 z_min = ndim == 3 ? (i64)zs - bound : 0, z_max = ndim == 3 ? (i64)ze + bound : 1;
        for (i64 i0 = y_min; i0 < y_max; ++i0)
            for (i64 j0 = x_min; j0 < x_max; ++j0)
                for (i64 k0 = z_min; k0 < z_max; ++k0) {
                    i64 i = wrap_cell(i0, N), j = wrap_cell(j0, N);
                    double dx = cell_overlap(j, xs, xe, Nf), dy = cell_overlap(i, ys, ye, Nf);
                    if (ndim == 3) {
                        i64 k = wrap_cell(k0, N);
                        double dz = cell_overlap(k, zs, ze, Nf);
                        if (dx > 0 && dy > 0 && dz > 0) grid[(i * N + j) * N + k] += dx * dy * dz * val[p];
                    } else if (dx > 0 && dy > 0) {
                        grid[i * N + j] += dx * dy * val[p];
                    }
                }
    }
}

static i64 argmin_abs(const double *bins, i64 n, double x)   /* np.argmin(np.abs(bins - x)): first minimum */
{
    i64 best = 0; double bv = fabs(bins[0] - x);
    for (i64 i = 1; i < n; ++i) { double v = fabs(bins[i] - x); if (v < bv) { bv = v; best = i; } }
    return best;
}

static void np_linspace_sym(i64 Nsize, double res, double *x)  /* np.linspace(-Nsize/2, Nsize/2, Nsize) * res */
{
    double start = -(double)Nsize / 2, stop = (double)Nsize / 2;
    double step = (stop - start) / (double)(Nsize - 1);
    for (i64 i = 0; i < Nsize; ++i) { double y = (double)i * step; y += start; x[i] = y; }
    x[Nsize - 1] = stop;
    for (i64 i = 0; i < Nsize; ++i) x[i] = x[i] * res;
}

static void pick_indices(i64 center, i64 width, i64 Npix, i64 *out)   /* Map2DRunner.py:403-429 */
{
    for (i64 t = 0; t < 2 * width; ++t) {
        i64 v = center - width + t;
        if (v < 0) v += Npix;
        if (v >= Npix) v -= Npix;
        out[t] = v;
    }
}

/*
 * Halo loop of BaryonifyGrid.process (mode 0, Map2DRunner.py:476-575) or PaintProfilesGrid.process (mode 1, :708-812).
 *   bins [npix] pixel centres; a = 1/(1+redshift); R[j] = mass_def.get_radius(cosmo, M_j, a) physical Mpc;
 *   lnM[j] = the ln M table coordinate.  HaloNDCatalog stores float32 columns (io.py:205) and the read-out takes
 *   np.log of that float32 scalar (BaryonCorrection.py:369, Tabulate.py:283), i.e. a float32 logarithm whose last
 *   bit depends on numpy's SIMD logf: the caller evaluates it with numpy and passes the result in;
 *   Rmod[j] = model-side comoving radius (BaryonCorrection.py:370), used by mode 0 only;
 *   rmat [nhalo][4] row-major 2x2 shear matrices (use_ellipticity, 2D only) or NULL;
 *   table: mode 0 displacement table, mode 1 the LOG of raw_input_2D (2D maps) / raw_input_3D (3D maps).
 *   out: mode 0 pix_offsets [npix^ndim][ndim], mode 1 new_map [npix^ndim]; accumulated into.
 * Returns the number of (halo, pixel) pairs of the cutouts; *assert_fail is set when the reference's
 * "Halo offsets ... are larger than res" assert (:516, :747) would fire.
 */
i64 bfgo_grid_loop(int mode, int ndim, i64 npix, const double *bins, i64 nhalo,
                   const double *hx, const double *hy, const double *hz, const double *lnM,
                   double a, const double *R, const double *Rmod, const double *rmat,
                   int nextra, const double *const *extra,
                   int tdim, const int *tn, const double *const *taxes, const double *tvalues,
                   int rdelta_sampling, double eps_runner, double eps_model, double *out, int *assert_fail)
{
    rgi_t t = rgi_make(tdim, tn, taxes, tvalues);
    const double res = bins[1] - bins[0];
    double bmax = bins[0];
    for (i64 i = 1; i < npix; ++i) if (bins[i] > bmax) bmax = bins[i];
    i64 total = 0;
    (void)nextra;
    double *x = (double *)malloc(sizeof(double) * (size_t)(npix + 2));
    i64 *xi = (i64 *)malloc(sizeof(i64) * (size_t)(npix + 2)), *yi = (i64 *)malloc(sizeof(i64) * (size_t)(npix + 2)),
        *zi = (i64 *)malloc(sizeof(i64) * (size_t)(npix + 2));
    if (assert_fail) *assert_fail = 0;
    for (i64 j = 0; j < nhalo; ++j) {
        i64 Nsize;
        double R_j = R[j];
        if (mode == 0) {
            double R_q = eps_runner * R_j / a;                        /* :487 */
            if (R_q < 0) R_q = 0;                                     /* np.clip(R_q, 0, max(bins)/2) :488 */
            if (R_q > bmax / 2) R_q = bmax / 2;
            double Ns = 2 * R_q / res;                                /* :496 */
            Nsize = (i64)floor(Ns / 2) * 2;                           /* :497 */
            if (Nsize < 2) continue;                                  /* :498 */
        } else {
            R_j = R_j / a;                                            /* :718 comoving */
            double Ns = 2 * eps_runner * R_j / res;                   /* :726 */
            Nsize = (i64)floor(Ns / 2) * 2;
            if (Nsize < 2) Nsize = 2;                                 /* np.clip(Nsize, 2, bins.size//2) :728 */
            if (Nsize > npix / 2) Nsize = npix / 2;
        }
        np_linspace_sym(Nsize, res, x);                               /* :500 */
        i64 w = Nsize / 2;
        i64 xc = argmin_abs(bins, npix, hx[j]), yc = argmin_abs(bins, npix, hy[j]);
        i64 zc = ndim == 3 ? argmin_abs(bins, npix, hz[j]) : 0;
        pick_indices(xc, w, npix, xi);
        pick_indices(yc, w, npix, yi);
        if (ndim == 3) pick_indices(zc, w, npix, zi);
        double dx = bins[xc] - hx[j], dy = bins[yc] - hy[j], dz = ndim == 3 ? bins[zc] - hz[j] : 0.0;
        if (ndim == 2 && !(dx <= res && dy <= res)) { if (assert_fail) *assert_fail = 1; }

        double tx[BFGO_MAXDIM];
        tx[0] = log(1.0 / a);
        tx[1] = lnM[j];
        for (int k = 3; k < tdim; ++k) tx[k] = extra[k - 3][j];
        const i64 nk = ndim == 3 ? Nsize : 1;
        for (i64 i = 0; i < Nsize; ++i)
            for (i64 jj = 0; jj < Nsize; ++jj)
                for (i64 k = 0; k < nk; ++k) {
                    /* meshgrid(x, x[, x], indexing='xy'): x_grid[i,j,k] = x[j], y_grid = x[i], z_grid = x[k] */
                    double X = x[jj] + dx, Y = x[i] + dy, Z = ndim == 3 ? x[k] + dz : 0.0;
                    double r = ndim == 3 ? sqrt(X * X + Y * Y + Z * Z) : sqrt(X * X + Y * Y);
                    double xh = X / r, yh = Y / r, zh = Z / r;
                    if (rmat) {                                       /* :525-530 (2D only) */
                        const double *Rm = rmat + 4 * j;
                        double Xe = X * Rm[0] + Y * Rm[2], Ye = X * Rm[1] + Y * Rm[3];
                        r = sqrt(Xe * Xe + Ye * Ye);
                    }
                    i64 flat = ndim == 3 ? (xi[i] * npix + yi[jj]) * npix + zi[k] : xi[i] * npix + yi[jj];
                    ++total;
                    if (mode == 0) {
                        double Rc = Rmod[j];
                        tx[2] = rdelta_sampling ? (log(r) - log(Rc)) : log(r);
                        double d = rgi_eval(&t, tx);
                        if (!(r < eps_model * Rc)) d = 0.0;           /* BaryonCorrection.py:381-382 */
                        double off = d / res;                         /* :534 */
                        out[ndim * flat + 0] += off * xh;
                        out[ndim * flat + 1] += off * yh;
                        if (ndim == 3) out[ndim * flat + 2] += off * zh;
                    } else {
                        tx[2] = log(r);
                        double P = exp(rgi_eval(&t, tx));
                        int ok = isfinite(P) && (r < R_j * eps_runner);   /* :800-801 */
                        if (ok) out[flat] += P;
                    }
                }
    }
    free(x); free(xi); free(yi); free(zi);
    return total;
}

/* np.histogramdd(coords, bins=(edges,)*ndim, weights=w): edges [nedge] ascending; last edge inclusive */
void bfgo_histogramdd(int ndim, i64 n, const double *const *coords, const double *weights, i64 nedge, const double *edges, double *out)
{
    const i64 nb = nedge - 1;
    for (i64 p = 0; p < n; ++p) {
        i64 idx[3] = {0, 0, 0};
        int ok = 1;
        for (int d = 0; d < ndim && ok; ++d) {
            double v = coords[d][p];
            /* searchsorted(edges, v, side='right') */
            i64 lo = 0, hi = nedge;
            while (lo < hi) { i64 mid = (lo + hi) >> 1; if (edges[mid] <= v) lo = mid + 1; else hi = mid; }
            i64 c = lo;
            if (v == edges[nedge - 1]) c -= 1;
            if (c < 1 || c > nb || v != v) ok = 0;
            idx[d] = c - 1;
        }
        if (!ok) continue;
        i64 flat = ndim == 3 ? (idx[0] * nb + idx[1]) * nb + idx[2] : idx[0] * nb + idx[1];
        out[flat] += weights ? weights[p] : 1.0;
    }
}

/* ====================================================================== particle-snapshot path
 * BaryonifySnapshot.process, SnapshotRunner.py:173-262.  The KD-tree ball query (scipy cKDTree, periodic, p = 2:
 * squared min-image separation <= R_q^2) is restated as a brute-force halo x particle loop.
 *   R[j] = mass_def.get_radius(cosmo, M_j, a) physical Mpc; Rmod[j] model-side comoving radius; lnM as in the grid loop.
 *   out: tot_offsets [npart][ndim], accumulated into.  Returns the number of (halo, particle) pairs inside the balls.
 */
static double min_image_d(double dx, double L)
{
    if (dx > L / 2) dx = dx - L;          /* compute_distance / enforce_periodicity :87-91 */
    if (dx < -L / 2) dx = dx + L;
    return dx;
}

i64 bfgo_snapshot_offsets(int ndim, i64 npart, const double *px, const double *py, const double *pz, double L,
                          i64 nhalo, const double *hx, const double *hy, const double *hz, const double *lnM,
                          double a, const double *R, const double *Rmod,
                          int tdim, const int *tn, const double *const *taxes, const double *tvalues,
                          int rdelta_sampling, double eps_runner, double eps_model, double *tot_offsets)
{
    rgi_t t = rgi_make(tdim, tn, taxes, tvalues);
    i64 total = 0;
    for (i64 j = 0; j < nhalo; ++j) {
        double R_q = eps_runner * R[j] / a;                           /* :221 */
        if (R_q < 0) R_q = 0;                                         /* :222 np.clip(R_q, 0, L/2) */
        if (R_q > L / 2) R_q = L / 2;
        double tx[BFGO_MAXDIM];
        tx[0] = log(1.0 / a);
        tx[1] = lnM[j];
        double Rc = Rmod[j];
        for (i64 p = 0; p < npart; ++p) {
            double dx = min_image_d(px[p] - hx[j], L), dy = min_image_d(py[p] - hy[j], L);
            double dz = ndim == 3 ? min_image_d(pz[p] - hz[j], L) : 0.0;
            double d2 = ndim == 3 ? dx * dx + dy * dy + dz * dz : dx * dx + dy * dy;
            if (!(d2 <= R_q * R_q)) continue;                         /* :225 / :237 query_ball_point */
            ++total;
            double d = sqrt(d2);                                      /* :228 / :241 */
            tx[2] = rdelta_sampling ? (log(d) - log(Rc)) : log(d);
            double disp = rgi_eval(&t, tx);
            if (!(d < eps_model * Rc)) disp = 0.0;                    /* BaryonCorrection.py:381-382 */
            double off = disp * a;                                    /* :232 / :247 */
            if (!isfinite(off)) off = 0.0;                            /* :233 / :248 */
            tot_offsets[ndim * p + 0] += off * (dx / d);
            tot_offsets[ndim * p + 1] += off * (dy / d);
            if (ndim == 3) tot_offsets[ndim * p + 2] += off * (dz / d);
        }
    }
    return total;
}
