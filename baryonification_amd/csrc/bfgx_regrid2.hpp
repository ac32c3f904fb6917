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
template <typename real> struct RMath;
template <> struct RMath<float> {
    static __device__ inline float rcp(float x) { const float y = __builtin_amdgcn_rcpf(x); return y * __builtin_fmaf(-x, y, 2.0f); }
};
template <> struct RMath<double> {
    static __device__ inline double rcp(double x) { return fast_rcp(x); }
};

// The 4 targets of one displaced pixel by the fast route (see the header comment); returns false when the pixel needs the
// generic route: next to a pole, a move of more than one ring, more than `maxcols` columns or 0.02 rad.  x = column of the
// pixel relative to the tile's first pixel of its ring (negative / beyond the tile for apron pixels of the gathering kernel).
template <typename real>
__device__ inline bool regrid_near_targets(const RegRow *rows, const RegRowC<real> *rowc, int rth0, int ti, int x,
                                           real o0, real o1, real o2, real maxcols, int tr[4], int tk[4], double w[4])
{
    const RegRowC<real> rc = rowc[ti];
    const RegRow &rw = rows[ti];
    // cos / sin of the pixel's azimuth: rotation of the tile's first pixel by x dphi (series), or Cody-Waite in fp64
    // where the tile spans more than 0.45 rad (polar bands)
    real c, s;
    const real alpha = (real)x * rc.dphi;
    if (abs_(alpha) <= (real)0.45) {
        real sa, oma;
        PMath<real>::sin_omc(alpha, sa, oma);
        c = rc.c0 - (rc.c0 * oma + rc.s0 * sa);
        s = rc.s0 - (rc.s0 * oma - rc.c0 * sa);
    } else {
        double s64, c64;
        sincos_bounded(((double)(rw.ks + x) + (rw.shf ? 0.5 : 0.0)) * rw.dphi, s64, c64);
        c = (real)c64; s = (real)s64;
    }
    const real a = o0 * c + o1 * s, b = o1 * c - o0 * s;               // in-plane radial / azimuthal components of the offset
    const real xr = rc.sth + a;
    const real t = b * RMath<real>::rcp(xr);
    const real t2 = t * t;
    // |v + o|^2 = 1 + e;  1/|v + o| = 1 - e/2 + 3 e^2/8 - 5 e^3/16
    const real e = (real)2 * (rc.sth * a + rc.z * o2) + (a * a + b * b + o2 * o2);
    const real invn = fma_(e, fma_(e, fma_(e, (real)-0.3125, (real)0.375), (real)-0.5), (real)1);
    // sqrt(1 + t^2) - 1 = t^2/2 - t^4/8 + t^6/16
    const real sq1 = t2 * fma_(t2, fma_(t2, (real)0.0625, (real)-0.125), (real)0.5);
    const real q = (fma_(a, rc.z, -(o2 * rc.sth)) + xr * rc.z * sq1) * invn;              // sin(theta_new - theta)
    const real q2 = q * q;
    const real dth = q * fma_(q2, fma_(q2, fma_(q2, (real)(15.0 / 336.0), (real)0.075), (real)(1.0 / 6.0)), (real)1);   // asin q
    const real dph = t * fma_(t2, fma_(t2, fma_(t2, (real)(-1.0 / 7.0), (real)0.2), (real)(-1.0 / 3.0)), (real)1);      // atan t
    // ring above / below and the colatitude weight
    const bool down = dth >= (real)0;                       // theta_new in [theta_r, theta_{r+1})
    const real wq = down ? dth * rc.inv_dth_dn : -dth * rc.inv_dth_up;      // fraction of the ring spacing moved
    const real wo_all = dph * rc.inv_dphi;                  // displacement along the own ring in pixels
    const bool fast = (xr > (real)0) && (abs_(t) <= (real)0.02) && (abs_(q) <= (real)0.02) && (wq < (real)1) && (abs_(wo_all) <= maxcols) &&
                      (down ? rc.inv_dth_dn : rc.inv_dth_up) > (real)0;
    if (!fast) return false;
    const real wtheta = down ? wq : (real)1 - wq;           // weight of the lower ring (ir2)
    // own ring: u = k + dphi / (2 pi / nr)
    const int nro = rw.nr;
    int kown = rw.ks + x;
    if (kown < 0) kown += nro;
    if (kown >= nro) kown -= nro;
    const real fo = __builtin_floor(wo_all);
    const real wo = wo_all - fo;
    int jo = kown + (int)fo;
    // neighbouring ring (r + 1 if down else r - 1): u' = (k + sh) nr' / nr - sh' + dphi / (2 pi / nr'), the first two terms
    // as an exact rational evaluated in fp64 (never an integer for rings of different length or shift)
    const RegRow &rn = rows[down ? ti + 1 : ti - 1];
    const real inv_dphi_n = rowc[down ? ti + 1 : ti - 1].inv_dphi;
    const double B = ((double)kown + (rw.shf ? 0.5 : 0.0)) * ((double)rn.nr * rw.dphi * kInvTwoPi) - (rn.shf ? 0.5 : 0.0);
    const double Bf = floor(B);
    const real wn_all = (real)(B - Bf) + dph * inv_dphi_n;
    const real fn = __builtin_floor(wn_all);
    const real wn = wn_all - fn;
    int jn = (int)Bf + (int)fn;
    const int nrn = rn.nr;
    int jo2 = jo + 1, jn2 = jn + 1;
    jo = jo < 0 ? jo + nro : (jo >= nro ? jo - nro : jo);
    jo2 = jo2 < 0 ? jo2 + nro : (jo2 >= nro ? jo2 - nro : jo2);
    jn = jn < 0 ? jn + nrn : (jn >= nrn ? jn - nrn : jn);
    jn2 = jn2 < 0 ? jn2 + nrn : (jn2 >= nrn ? jn2 - nrn : jn2);
    const int ring_o = rth0 + ti, ring_n = down ? ring_o + 1 : ring_o - 1;
    // get_interpol order: upper ring (ir1) first
    const real w_own = down ? (real)1 - wtheta : wtheta, w_nb = (real)1 - w_own;
    tr[0] = down ? ring_o : ring_n; tr[1] = tr[0]; tr[2] = down ? ring_n : ring_o; tr[3] = tr[2];
    const int ju = down ? jo : jn, ju2 = down ? jo2 : jn2, jl = down ? jn : jo, jl2 = down ? jn2 : jo2;
    const real wu = down ? wo : wn, wl = down ? wn : wo;
    const real wtu = down ? w_own : w_nb, wtl = down ? w_nb : w_own;
    tk[0] = ju; tk[1] = ju2; tk[2] = jl; tk[3] = jl2;
    w[0] = (double)(((real)1 - wu) * wtu); w[1] = (double)(wu * wtu);
    w[2] = (double)(((real)1 - wl) * wtl); w[3] = (double)(wl * wtl);
    return true;
}

// ---------------------------------------------------------------------------------- K2, gathering form (full-map regrid)
// The scatter form above ends in ~1.3 global fp64 atomics per map pixel (tile + apron), which is what bounds it (the
// memory-side atomic rate), and it needs a zeroed output.  The gathering form makes every OUTPUT pixel the property of one
// workgroup: a tile evaluates the displaced position of its own pixels AND of the pixels in a thin apron around it
// (kGatherR rings, kGatherK columns: 1.3x the evaluations), keeps only the deposits that land inside the tile (LDS), and
// stores the tile once with plain stores -- no global atomics, no zero-fill.  A deposit is "near" when the fast route
// applies with a move of at most 3 columns, a property of the source pixel alone, so every tile that sees the pixel
// classifies it the same way; anything else (pole caps, very large displacements) is appended by the tile that OWNS the
// source pixel to a global list that a small fix-up kernel adds to the stored map afterwards.
constexpr int kGatherR = 1;       // apron rings above / below the tile (near deposits go to the own ring or the next one)
constexpr int kGatherK = 7;       // apron columns left / right of the tile (3 columns of move + 1 + 2 of ring-to-ring column mapping + 1)

struct FarList {
    unsigned long long *count;    // entries appended (may exceed cap: overflow)
    int64_t *pix;
    double *val;
    int64_t cap;
    int32_t *overflow;
};

__host__ __device__ inline size_t regrid3_lds_bytes(int BR, int W, size_t real_size)
{
    return (size_t)BR * W * sizeof(double) + (size_t)(BR + 2 * kGatherR + 2) * (sizeof(RegRow) + 8 * real_size);
}

template <typename ACC, typename real>
__global__ void __launch_bounds__(256)
tile_regrid3_kernel(Hpx h, Tiling T, const double *__restrict__ map_in, const ACC *__restrict__ offsets,
                    double *__restrict__ map_out, FarList far, double *__restrict__ tile_sums, int tile_off)
{
    // map_in, offsets and map_out are indexed by GLOBAL pixel number.  tile_off < 0: all tiles, heavy ones first; tile_off >= 0
    // (a rank that owns a range of bands): tiles tile_off + blockIdx.x, and the caller passes offsets / map_out pointers
    // shifted so that only the pixels this rank holds (its bands + one ring either side / its bands) are touched.
    extern __shared__ __align__(16) unsigned char smem[];
    const int NT = T.BR + 2 * kGatherR + 2;                    // rings rth0 .. rth0 + NT - 1 in the ring tables
    double *acc = reinterpret_cast<double *>(smem);            // [BR][W]: the tile's own pixels
    RegRow *rows = reinterpret_cast<RegRow *>(acc + T.BR * T.W);
    RegRowC<real> *rowc = reinterpret_cast<RegRowC<real> *>(rows + NT);
    const int tile = (tile_off < 0) ? T.tile_order[blockIdx.x] : tile_off + (int)blockIdx.x;
    const int band = T.tile_band[tile];
    const int nphi = T.band_nphi[band];
    const int tj = tile - T.band_tile0[band];
    const int nl4 = (int)(4 * h.nside);
    const int i0 = 1 + band * T.BR;
    const int i1 = min(i0 + T.BR, nl4);                        // exclusive
    const int tid = threadIdx.x;
    const int rth0 = i0 - kGatherR - 1;
    double sum_in = 0.0, sum_out = 0.0;                        // mass-conservation sums (HealpixRunner.py:344-345)
    for (int i = tid; i < T.BR * T.W; i += 256) acc[i] = 0.0;
    if (tid < NT) {
        const int ring = rth0 + tid;
        RegRow rw;
        rw.theta = (ring < 1) ? -1.0e300 : 1.0e300;
        rw.z = rw.sth = rw.dphi = rw.inv_dphi = rw.c0 = rw.s0 = 0.0;
        rw.start = 0; rw.nr = 0; rw.ks = 0; rw.ke = 0; rw.shf = 0;
        if (ring >= 1 && ring <= nl4 - 1) {
            int64_t st, nr64; bool shf;
            ring_info_small(h, ring, st, nr64, shf);
            ring_z_sth(h, ring, rw.z, rw.sth);
            rw.theta = atan2_generic(rw.sth, rw.z);
            rw.start = st; rw.nr = (int)nr64; rw.shf = shf ? 1 : 0;
            rw.dphi = kTwoPi / (double)rw.nr;
            rw.inv_dphi = (double)rw.nr * kInvTwoPi;
            rw.ks = tile_ks(tj, rw.nr, nphi);
            rw.ke = tile_ks(tj + 1, rw.nr, nphi);
            sincos_bounded(((double)rw.ks + (shf ? 0.5 : 0.0)) * rw.dphi, rw.s0, rw.c0);
        }
        rows[tid] = rw;
    }
    __syncthreads();
    if (tid < NT) {
        const RegRow &rw = rows[tid];
        RegRowC<real> rc;
        rc.z = (real)rw.z; rc.sth = (real)rw.sth; rc.dphi = (real)rw.dphi; rc.inv_dphi = (real)rw.inv_dphi;
        rc.c0 = (real)rw.c0; rc.s0 = (real)rw.s0;
        rc.inv_dth_up = (real)0; rc.inv_dth_dn = (real)0;
        if (rw.nr > 0) {
            if (tid > 0 && rows[tid - 1].nr > 0) rc.inv_dth_up = (real)(1.0 / (rw.theta - rows[tid - 1].theta));
            if (tid < NT - 1 && rows[tid + 1].nr > 0) rc.inv_dth_dn = (real)(1.0 / (rows[tid + 1].theta - rw.theta));
        }
        rowc[tid] = rc;
    }
    __syncthreads();

    auto far_add = [&](int64_t p, double v) {
        const unsigned long long i = atomicAdd(far.count, 1ull);
        if ((int64_t)i < far.cap) { far.pix[i] = p; far.val[i] = v; } else atomicOr(far.overflow, 1);
        sum_out += v;
    };

    // source pixels: the tile's rings +- kGatherR, its columns +- kGatherK (a tile that spans whole rings has no column apron)
    // (in a short ring the apron is what is left of the ring, split between the two sides, so that no pixel is visited twice)
    const int NR = T.BR + 2 * kGatherR;
    const int kap = (nphi == 1) ? 0 : kGatherK;
    int maxspan = 0;
    for (int i = 1; i <= NR; ++i) maxspan = max(maxspan, rows[i].ke - rows[i].ks);
    const int LWs = maxspan + 2 * kap;
    const unsigned inv_lws = (unsigned)((0x100000000ull + (unsigned)LWs - 1u) / (unsigned)LWs);      // idx / LWs for idx < 2^16
    struct Src { int ti, x; bool ok, own; double val; ACC o0, o1, o2; };
    auto fetch = [&](int idx) {
        Src sx;
        sx.ok = false; sx.own = false; sx.ti = 0; sx.x = 0; sx.val = 0.0; sx.o0 = sx.o1 = sx.o2 = (ACC)0;
        if (idx < NR * LWs) {
            const int r = (int)__umulhi((unsigned)idx, inv_lws), x = idx - r * LWs - kap;
            const RegRow &rw = rows[r + 1];
            const int span = rw.ke - rw.ks, rest = rw.nr - span;
            const int kl = min(kap, rest >> 1), kr = min(kap, rest - kl);
            if (rw.nr > 0 && x >= -kl && x < span + kr) {
                int k = rw.ks + x;
                if (k < 0) k += rw.nr;
                if (k >= rw.nr) k -= rw.nr;
                const int64_t p = rw.start + k;
                sx.ok = true; sx.ti = r + 1; sx.x = x;
                sx.own = (r >= kGatherR) && (r < kGatherR + (i1 - i0)) && (x >= 0) && (x < rw.ke - rw.ks);
                sx.val = map_in[p];
                sx.o0 = offsets[3 * p + 0]; sx.o1 = offsets[3 * p + 1]; sx.o2 = offsets[3 * p + 2];
            }
        }
        return sx;
    };
    Src nxt = fetch(tid);
    for (int idx = tid; idx < NR * LWs; idx += 256) {
        const Src cur = nxt;
        nxt = fetch(idx + 256);
        if (cur.ok && cur.own) sum_in += cur.val;
        if (!cur.ok || !(cur.val > 0.0)) continue;                           // HealpixRunner.py:335
        const double val = cur.val;
        int tr[4], tk[4];
        double w[4];
        if (regrid_near_targets<real>(rows, rowc, rth0, cur.ti, cur.x, (real)cur.o0, (real)cur.o1, (real)cur.o2, (real)3, tr, tk, w)) {
#pragma unroll
            for (int q4 = 0; q4 < 4; ++q4) {                                 // regrid_pixels_hpix :64, deposits into this tile only
                const int tt = tr[q4] - rth0;
                if (tr[q4] >= i0 && tr[q4] < i1) {
                    const RegRow &rt = rows[tt];
                    if (tk[q4] >= rt.ks && tk[q4] < rt.ke) {
                        const double v = w[q4] * val;
                        atomicAdd(acc + (tr[q4] - i0) * T.W + (tk[q4] - rt.ks), v);
                        sum_out += v;
                    }
                }
            }
        } else if (cur.own) {                                                // rare: the owner of the source pixel lists its deposits
            regrid_targets_generic(h, rows, NT - 2, rth0, cur.ti, cur.x, (double)cur.o0, (double)cur.o1, (double)cur.o2, tr, tk, w);
            for (int q4 = 0; q4 < 4; ++q4) {
                int64_t st_t, nr64; bool sh_t;
                ring_info_small(h, tr[q4], st_t, nr64, sh_t);
                far_add(st_t + tk[q4], w[q4] * val);
            }
        }
    }
    __syncthreads();

    // flush: every pixel of the tile is stored exactly once
    const int lane = tid & (kWave - 1), wid = tid / kWave;
    for (int rr = wid; rr < i1 - i0; rr += 256 / kWave) {
        const RegRow &rt = rows[rr + kGatherR + 1];
        double *dst = map_out + rt.start + rt.ks;
        const int n = rt.ke - rt.ks;
        for (int xx = lane; xx < n; xx += kWave) dst[xx] = acc[rr * T.W + xx];
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
}

// adds the listed far deposits to the stored map
__global__ void __launch_bounds__(256)
regrid_far_kernel(FarList far, double *__restrict__ map_out)
{
    unsigned long long n = *far.count;
    if ((int64_t)n > far.cap) n = (unsigned long long)far.cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x)
        atomicAdd(map_out + far.pix[i], far.val[i]);
}

// banded regrid: adds the listed deposits whose pixel lies in [p0, p1) to the slice that starts at pixel p0, counts the others
__global__ void __launch_bounds__(256)
regrid_far_local_kernel(FarList far, double *__restrict__ out_slice, int64_t p0, int64_t p1, unsigned long long *__restrict__ foreign)
{
    unsigned long long n = *far.count;
    if ((int64_t)n > far.cap) n = (unsigned long long)far.cap;
    for (unsigned long long i = (unsigned long long)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (unsigned long long)gridDim.x * blockDim.x) {
        const int64_t p = far.pix[i];
        if (p >= p0 && p < p1) atomicAdd(out_slice + (p - p0), far.val[i]);
        else if (foreign) atomicAdd(foreign, 1ull);
    }
}

}  // namespace bfgx
