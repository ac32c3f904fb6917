// Table-builder kernels, fp64 throughout.
//
//   project_kernel          K5  SchneiderProfiles._projected_realspace    Schneider19.py:245-252
//   enclosed_mass_kernel    K4a Baryonification2D.get_masses              BaryonCorrection.py:645-661
//   displacement_kernel     K4b setup_interpolator per-mass loop body     BaryonCorrection.py:226-301
//   pressure_kernel         K6  Pressure._real                            Thermodynamic.py:240-271
//
// Inputs are 3-D densities SAMPLED by the host on the radial grids the reference uses; the kernels do the
// line-of-sight projection, the radial prefix sums, the log-log PCHIP interpolation (scipy
// PchipInterpolator semantics incl. end-slope rules, extrapolate=False -> NaN) and the monotone-mask /
// inverse-PCHIP composition d(r) = M_DMB^-1(M_DMO(r)) - r.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace bfgx {

constexpr int kTabThreads = 1024;
constexpr double kPressureAtInfinity = 1e-200;      // Thermodynamic.py:38

// ------------------------------------------------------------------ block-wide helpers (blockDim = kTabThreads)
__device__ inline double block_excl_scan_sum(double v, double *sh /*[kTabThreads]*/, double &total)
{
    const int tid = threadIdx.x;
    sh[tid] = v;
    __syncthreads();
    for (int off = 1; off < kTabThreads; off <<= 1) {
        const double u = (tid >= off) ? sh[tid - off] : 0.0;
        __syncthreads();
        sh[tid] += u;
        __syncthreads();
    }
    total = sh[kTabThreads - 1];
    const double r = sh[tid] - v;
    __syncthreads();
    return r;
}

// (integers: shuffles inside the 64-lane wave, the 16 wave totals through LDS -- two barriers instead of the twenty-two of the scan above)
__device__ inline int block_excl_scan_int(int v, int *sh, int &total)
{
    constexpr int kW = 64;
    const int tid = threadIdx.x, lane = tid & (kW - 1), wid = tid / kW;
    int incl = v;
#pragma unroll
    for (int s = 1; s < kW; s <<= 1) {
        const int u = __shfl_up(incl, s, kW);
        if (lane >= s) incl += u;
    }
    if (lane == kW - 1) sh[wid] = incl;
    __syncthreads();
    int wpre = 0, tot = 0;
#pragma unroll
    for (int w = 0; w < kTabThreads / kW; ++w) { const int pw = sh[w]; tot += pw; if (w < wid) wpre += pw; }
    total = tot;
    __syncthreads();                                    // (sh is the caller's to reuse)
    return wpre + incl - v;
}

// ------------------------------------------------------------------ scipy PchipInterpolator
__device__ inline double dsign(double v) { return (v > 0.0) - (v < 0.0); }

__device__ inline double pchip_edge(double h0, double h1, double m0, double m1)
{
    double d = ((2.0 * h0 + h1) * m0 - h0 * m1) / (h0 + h1);
    if (dsign(d) != dsign(m0)) d = 0.0;
    else if (dsign(m0) != dsign(m1) && fabs(d) > 3.0 * fabs(m0)) d = 3.0 * m0;
    return d;
}

// derivative at node k of the PCHIP through (x[0..n), y[0..n)), n >= 2
__device__ inline double pchip_deriv(const double *x, const double *y, int n, int k)
{
    if (n == 2) return (y[1] - y[0]) / (x[1] - x[0]);
    if (k == 0) {
        const double h0 = x[1] - x[0], h1 = x[2] - x[1];
        return pchip_edge(h0, h1, (y[1] - y[0]) / h0, (y[2] - y[1]) / h1);
    }
    if (k == n - 1) {
        const double h0 = x[n - 1] - x[n - 2], h1 = x[n - 2] - x[n - 3];
        return pchip_edge(h0, h1, (y[n - 1] - y[n - 2]) / h0, (y[n - 2] - y[n - 3]) / h1);
    }
    const double hm = x[k] - x[k - 1], hp = x[k + 1] - x[k];
    const double mm = (y[k] - y[k - 1]) / hm, mp = (y[k + 1] - y[k]) / hp;
    if (dsign(mm) != dsign(mp) || mm == 0.0 || mp == 0.0) return 0.0;
    const double w1 = 2.0 * hp + hm, w2 = hp + 2.0 * hm;
    return 1.0 / ((w1 / mm + w2 / mp) / (w1 + w2));
}

// PchipInterpolator(x, y, extrapolate=False)(q)
__device__ inline double pchip_eval(const double *x, const double *y, int n, double q)
{
    if (n < 2 || !(q >= x[0]) || !(q <= x[n - 1])) return __builtin_nan("");
    int lo = 0, hi = n - 1;
    while (hi - lo > 1) { const int mid = (lo + hi) >> 1; if (q >= x[mid]) lo = mid; else hi = mid; }
    const double dx = x[lo + 1] - x[lo];
    const double slope = (y[lo + 1] - y[lo]) / dx;
    const double d0 = pchip_deriv(x, y, n, lo), d1 = pchip_deriv(x, y, n, lo + 1);
    const double t = (d0 + d1 - 2.0 * slope) / dx;
    const double c0 = t / dx, c1 = (slope - d0) / dx - t, c2 = d0, c3 = y[lo];
    const double s = q - x[lo];
    return c3 + s * (c2 + s * (c1 + s * c0));
}

// ------------------------------------------------------------------ K5
// grid (ceil(nr / 256), nrows), block 256; LDS: l[nl], rho[nl], slope[nl]
__global__ void __launch_bounds__(256)
project_kernel(int nl, const double *__restrict__ l, const double *__restrict__ rho /*[nrows][nl]*/,
               int64_t nr, const double *__restrict__ r, double scale, double *__restrict__ out /*[nrows][nr]*/)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *sl = reinterpret_cast<double *>(smem), *sr = sl + nl, *ss = sr + nl;
    const int row = blockIdx.y;
    for (int i = threadIdx.x; i < nl; i += blockDim.x) {
        sl[i] = l[i];
        sr[i] = rho[(int64_t)row * nl + i];
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nl - 1; i += blockDim.x) ss[i] = (sr[i + 1] - sr[i]) / (sl[i + 1] - sl[i]);
    __syncthreads();
    const int64_t j = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= nr) return;
    const double rj = r[j], rj2 = rj * rj;
    int m = 0;
    double acc = 0.0, fprev = 0.0;
    for (int k = 0; k < nl; ++k) {
        const double x = sqrt(sl[k] * sl[k] + rj2);
        double f;
        if (x >= sl[nl - 1]) f = sr[nl - 1];                  // np.interp clamps to fp[-1]
        else {
            while (m < nl - 2 && x >= sl[m + 1]) ++m;
            f = ss[m] * (x - sl[m]) + sr[m];
        }
        if (k > 0) acc += (sl[k] - sl[k - 1]) * (f + fprev) / 2.0;      // np.trapz
        fprev = f;
    }
    out[(int64_t)row * nr + j] = 2.0 * acc * scale;
}

// ------------------------------------------------------------------ K4a
// one block (kTabThreads) per row.  dim 2: Sigma[row][n] (already carrying the factor a), terms 2 pi r^2 Sigma dlnr
// (Baryonification2D.get_masses); dim 3: the 3-D density, terms 4 pi r^3 rho dlnr (Baryonification3D.get_masses,
// BaryonCorrection.py:519-546).  Negative samples count as 0, points with a zero sample or a non-finite running sum are
// dropped before the log-log PCHIP.  Scratch cx/cy: [nrows][n].
__device__ inline double shell_term(int dim, double r, double v, double dlnr)
{
    return dim == 3 ? 4.0 * kPi * (r * r * r) * v * dlnr : 2.0 * kPi * r * r * v * dlnr;
}

__global__ void __launch_bounds__(kTabThreads)
enclosed_mass_kernel(int64_t n, const double *__restrict__ r_int, const double *__restrict__ Sigma,
                     int nr, const double *__restrict__ r, double *__restrict__ cx, double *__restrict__ cy,
                     double *__restrict__ M_f /*[nrows][nr]*/, int dim)
{
    __shared__ double shd[kTabThreads];
    __shared__ int shi[kTabThreads];
    const int row = blockIdx.x, tid = threadIdx.x;
    const double *S = Sigma + (int64_t)row * n;
    double *ox = cx + (int64_t)row * n, *oy = cy + (int64_t)row * n;
    const double dlnr = log(r_int[1] / r_int[0]);
    const int64_t per = (n + kTabThreads - 1) / kTabThreads;
    const int64_t lo = min((int64_t)tid * per, n), hi = min(lo + per, n);
    double s = 0.0;
    int c = 0;
    for (int64_t i = lo; i < hi; ++i) {
        const double sg = S[i] < 0.0 ? 0.0 : S[i];
        s += shell_term(dim, r_int[i], sg, dlnr);
    }
    double tot;
    const double base = block_excl_scan_sum(s, shd, tot);
    double run = base;
    for (int64_t i = lo; i < hi; ++i) {                       // count usable points
        const double sg = S[i] < 0.0 ? 0.0 : S[i];
        run += shell_term(dim, r_int[i], sg, dlnr);
        c += (sg > 0.0 && isfinite(run)) ? 1 : 0;
    }
    int ctot;
    int cbase = block_excl_scan_int(c, shi, ctot);
    run = base;
    for (int64_t i = lo; i < hi; ++i) {                       // compact (ln r, ln M_enc)
        const double sg = S[i] < 0.0 ? 0.0 : S[i];
        run += shell_term(dim, r_int[i], sg, dlnr);
        if (sg > 0.0 && isfinite(run)) { ox[cbase] = log(r_int[i]); oy[cbase] = log(run); ++cbase; }
    }
    __threadfence_block();
    __syncthreads();
    for (int j = tid; j < nr; j += kTabThreads)
        M_f[(int64_t)row * nr + j] = exp(pchip_eval(ox, oy, ctot, log(r[j])));
}

// ------------------------------------------------------------------ K4b
// one block (256) per row; N_R <= kMaxNR.  status: 0 ok, 1 iterate > 30, 2 fewer than 5 usable points.
constexpr int kMaxNR = 4096;

__global__ void __launch_bounds__(256)
displacement_kernel(int nr, const double *__restrict__ r, const double *__restrict__ M_dmo, const double *__restrict__ M_dmb,
                    double *__restrict__ d_out, int32_t *__restrict__ status)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *lb = reinterpret_cast<double *>(smem);     // ln M_DMB
    double *lo_ = lb + nr;                             // ln M_DMO
    double *lr = lo_ + nr;                             // ln r
    double *ax = lr + nr, *ay = ax + nr;               // compacted DMB: (ln M_DMB, ln r)
    double *bx = ay + nr, *by = bx + nr;               // compacted DMO: (ln r, ln M_DMO)
    int *mask = reinterpret_cast<int *>(by + nr);
    int *prev = mask + nr;                             // previous masked index (-1: none)
    __shared__ int s_cnt, s_iter, s_stat, s_na, s_nb;
    __shared__ double s_min;
    __shared__ int red_i[256];
    __shared__ double red_d[256];
    const int row = blockIdx.x, tid = threadIdx.x;
    for (int i = tid; i < nr; i += 256) {
        lb[i] = log(M_dmb[(int64_t)row * nr + i]);
        lo_[i] = log(M_dmo[(int64_t)row * nr + i]);
        lr[i] = log(r[i]);
        mask[i] = 1;
    }
    if (tid == 0) { s_cnt = nr; s_iter = 0; s_stat = 0; s_min = -INFINITY; }
    __syncthreads();

    const int per = (nr + 255) / 256;
    const int c0 = min(tid * per, nr), c1 = min(c0 + per, nr);
    auto build_prev = [&]() {        // prev[i] = last masked index < i  (chunked, with carry through LDS)
        int last = -1;
        for (int i = c0; i < c1; ++i) if (mask[i]) last = i;
        red_i[tid] = last;
        __syncthreads();
        int carry = -1;
        for (int t = tid - 1; t >= 0; --t) if (red_i[t] >= 0) { carry = red_i[t]; break; }
        __syncthreads();
        last = carry;
        for (int i = c0; i < c1; ++i) { prev[i] = last; if (mask[i]) last = i; }
        __syncthreads();
    };
    auto count_and_min = [&](bool want_min) {
        int c = 0;
        double mn = INFINITY;
        bool nan_seen = false;
        for (int i = c0; i < c1; ++i) if (mask[i]) {
            ++c;
            if (want_min && prev[i] >= 0) {              // np.diff(x[mask], prepend=0)[1:]
                const double dv = lb[i] - lb[prev[i]];
                if (dv != dv) nan_seen = true; else mn = fmin(mn, dv);
            }
        }
        red_i[tid] = c;
        red_d[tid] = nan_seen ? __builtin_nan("") : mn;
        __syncthreads();
        if (tid == 0) {
            int ct = 0; double m = INFINITY; bool nn = false;
            for (int t = 0; t < 256; ++t) { ct += red_i[t]; if (red_d[t] != red_d[t]) nn = true; else m = fmin(m, red_d[t]); }
            s_cnt = ct;
            if (want_min) s_min = nn ? __builtin_nan("") : m;       // np.min propagates NaN
        }
        __syncthreads();
    };

    while (true) {
        if (!((s_min < 1e-5) && (s_cnt > 5))) break;            // BaryonCorrection.py:240
        build_prev();
        for (int i = c0; i < c1; ++i) if (mask[i]) {             // :242-245
            const double dv = lb[i] - (prev[i] >= 0 ? lb[prev[i]] : 0.0);
            const bool keep = (dv > 1e-5) && ((fabs(lb[i] - lo_[i]) > 1e-6) || (lo_[i] != lo_[i])) && isfinite(lb[i]);
            mask[i] = keep ? 1 : 0;
        }
        __syncthreads();
        if (tid == 0) { mask[0] = 1; s_iter += 1; }              // :248-250
        __syncthreads();
        if (s_iter > 30) {                                       // :252-259
            for (int i = tid; i < nr; i += 256) mask[i] = 0;
            if (tid == 0) { s_stat = 1; s_cnt = 0; }
            __syncthreads();
            break;
        }
        build_prev();
        count_and_min(true);
        if (s_cnt < 5) { if (tid == 0) s_stat = 2; __syncthreads(); break; }      // :261-265
    }
    __syncthreads();

    double *out = d_out + (int64_t)row * nr;
    if (s_cnt > 5) {                                             // :272
        if (tid == 0) {                                          // serial compaction (N_R is small)
            int na = 0, nb = 0;
            for (int i = 0; i < nr; ++i) {
                if (mask[i]) { ax[na] = lb[i]; ay[na] = lr[i]; ++na; }
                const double dv = lo_[i] - (i > 0 ? lo_[i - 1] : 0.0);                       // :276
                const bool fin = (dv > 1e-5) && ((fabs(lb[i] - lo_[i]) > 1e-6) || (lb[i] != lb[i])) && isfinite(lo_[i]);
                if (fin) { bx[nb] = lr[i]; by[nb] = lo_[i]; ++nb; }
            }
            s_na = na; s_nb = nb;
        }
        __syncthreads();
        for (int i = tid; i < nr; i += 256) {
            const double q = pchip_eval(bx, by, s_nb, lr[i]);                             // interp_DMO(ln r)
            const double off = exp(pchip_eval(ax, ay, s_na, q)) - r[i];                   // :283
            out[i] = isfinite(off) ? off : 0.0;                                           // :284
        }
    } else {
        for (int i = tid; i < nr; i += 256) out[i] = 0.0;                                 // :293
        if (tid == 0 && s_stat == 0) s_stat = 2;
    }
    __syncthreads();
    if (tid == 0) status[row] = s_stat;
}

// ------------------------------------------------------------------ K6
// one block (256) per row; 500-point hard-coded grid geomspace(1e-6, 1e3, 500).  G in Mpc^3 / Msun / s^2.
constexpr int kPressureN = 500;

__global__ void __launch_bounds__(256)
pressure_kernel(const double *__restrict__ r500, const double *__restrict__ rho_tot, const double *__restrict__ rho_gas,
                int nr_out, const double *__restrict__ r_out, double G, double unit, double cutoff,
                double *__restrict__ P_out /*[nrows][nr_out]*/)
{
    __shared__ double lx[kPressureN], ly[kPressureN], work[kPressureN];
    const int row = blockIdx.x, tid = threadIdx.x;
    const double dlnr = log(r500[1]) - log(r500[0]);
    if (tid == 0) {                                     // two 500-term running sums: serial is fine
        double run = 0.0;
        for (int i = 0; i < kPressureN; ++i) {
            const double ri = r500[i];
            run += ri * ri * ri * rho_tot[(int64_t)row * kPressureN + i] * dlnr;
            const double M_total = 4.0 * kPi * run;                                          // :246
            const double dP_dr = -G * M_total * rho_gas[(int64_t)row * kPressureN + i] / (ri * ri);   // :249
            work[i] = dP_dr * ri * dlnr;
        }
        run = 0.0;
        for (int i = kPressureN - 1; i >= 0; --i) {                                          // :258
            run += work[i];
            ly[i] = log(-run + kPressureAtInfinity);                                         // :260
            lx[i] = log(r500[i]);
        }
    }
    __syncthreads();
    for (int j = tid; j < nr_out; j += 256) {
        // scipy's PchipInterpolator validates y: a NaN/inf in ly poisons the neighbouring intervals exactly
        // as it does there because the Hermite coefficients are formed from the same local stencil
        double p = exp(pchip_eval(lx, ly, kPressureN, log(r_out[j]))) - kPressureAtInfinity;   // :261
        if (!isfinite(p)) p = 0.0;                                                           // :262
        p *= unit;                                                                           // :265
        double arg = r_out[j] - cutoff;
        const double kfac = (arg > 30.0) ? 0.0 : 1.0 / (1.0 + exp(2.0 * arg));                // :268-270
        P_out[(int64_t)row * nr_out + j] = p * kfac;
    }
}

}  // namespace bfgx
