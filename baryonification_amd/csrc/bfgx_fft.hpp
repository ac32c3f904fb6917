// bfgx_fft.hpp -- 3-D power-spectrum summary of a gridded map (examples/10_Reproduce_Schneider_deltaPk.ipynb
// cells 12 + 15: np.fft.fftn -> |F|^2 -> np.bincount in linear k-bins) for gfx950.
//
// The map is real, so only the kz <= N/2 half of the spectrum is computed (r2c along the contiguous axis, then two
// complex passes along the strided axes); the mirrored half enters the bin sums through a weight of 2.
// Every 1-D transform runs in LDS: decimation in time, bit-reversed on the way in, two radix-2 stages fused per
// barrier (radix-4 data movement), twiddles from a host-built fp64 table (copied to LDS in the strided passes); the r2c
// pass packs two real lines into one complex transform.  Rows of the half spectrum are padded to whole 128-byte lines
// (fft_pitch); the strided passes move tiles of 4 (or 8) lines that are adjacent in memory, so that global accesses stay
// contiguous and LDS accesses conflict-free (line index fastest).  The LAST pass writes nothing back: the transformed tile is
// binned from LDS (|F|^2, |k|, counts per linear k-bin), so the spectrum crosses HBM 5 times instead of 8.
// N must be a power of two, 8 <= N <= 1024.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bfgx_math.hpp"

namespace bfgx {

constexpr int kFftBlock = 256;
constexpr int kFftTileLog2 = 2;       // 1 << this = lines per workgroup in the strided passes (BFGX_FFT_TILE = 4 | 8 overrides)

__device__ inline int bit_reverse(int i, int log2n) { return (int)(__brev((unsigned)i) >> (32 - log2n)); }

__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-place decimation-in-time butterflies over `nl` interleaved lines held in LDS as buf[i * nl + l] (input already in
// bit-reversed order).  Two radix-2 stages (half sizes m and 2m) are fused per barrier: a thread takes the four elements
// p, p+m, p+2m, p+3m of a 4m-block, so every element crosses LDS once per TWO stages; a last single stage follows when
// log2(N) is odd.  tw[k] = exp(-2 pi i k / N), k < N/2.
__device__ inline void lds_fft_stages(double2 *buf, int N, int nl, const double2 *__restrict__ tw)
{
    int m = 1;
    const int nq = (N >> 2) * nl;
    for (; 4 * m <= N; m <<= 2) {
        __syncthreads();
        const int ts1 = (N >> 1) / m, ts2 = (N >> 2) / m;           // twiddle strides of the stages m and 2m
        for (int t = threadIdx.x; t < nq; t += kFftBlock) {
            const int l = t % nl, b = t / nl;                       // b: butterfly quad index
            const int pos = b & (m - 1);
            const int p = ((b - pos) << 2) + pos;
            const double2 w1 = tw[pos * ts1], w2 = tw[pos * ts2];
            const double2 w3 = make_double2(w2.y, -w2.x);           // W_4m^(pos + m) = -i W_4m^pos
            double2 a0 = buf[p * nl + l], a1 = buf[(p + m) * nl + l], a2 = buf[(p + 2 * m) * nl + l], a3 = buf[(p + 3 * m) * nl + l];
            const double2 t1 = cmul(w1, a1), t3 = cmul(w1, a3);     // stage m: (a0, a1) and (a2, a3)
            const double2 b0 = make_double2(a0.x + t1.x, a0.y + t1.y), b1 = make_double2(a0.x - t1.x, a0.y - t1.y);
            const double2 b2 = make_double2(a2.x + t3.x, a2.y + t3.y), b3 = make_double2(a2.x - t3.x, a2.y - t3.y);
            const double2 u2 = cmul(w2, b2), u3 = cmul(w3, b3);     // stage 2m: (b0, b2) and (b1, b3)
            buf[p * nl + l] = make_double2(b0.x + u2.x, b0.y + u2.y);
            buf[(p + 2 * m) * nl + l] = make_double2(b0.x - u2.x, b0.y - u2.y);
            buf[(p + m) * nl + l] = make_double2(b1.x + u3.x, b1.y + u3.y);
            buf[(p + 3 * m) * nl + l] = make_double2(b1.x - u3.x, b1.y - u3.y);
        }
    }
    if (m < N) {                                                    // remaining single stage (m = N/2)
        __syncthreads();
        const int nb = (N >> 1) * nl;
        const int tstep = (N >> 1) / m;
        for (int t = threadIdx.x; t < nb; t += kFftBlock) {
            const int l = t % nl, b = t / nl;
            const int pos = b & (m - 1);
            const int i = ((b - pos) << 1) + pos, j = i + m;
            const double2 u = buf[i * nl + l], v = cmul(tw[pos * tstep], buf[j * nl + l]);
            buf[i * nl + l] = make_double2(u.x + v.x, u.y + v.y);
            buf[j * nl + l] = make_double2(u.x - v.x, u.y - v.y);
        }
    }
    __syncthreads();
}

// pass 1: real lines along the contiguous axis -> first N/2 + 1 coefficients.  One workgroup transforms TWO adjacent
// lines a, b as the single complex sequence a + i b and separates the spectra afterwards:
//   A_k = (Z_k + conj(Z_{N-k})) / 2,   B_k = (Z_k - conj(Z_{N-k})) / (2i).
__global__ void __launch_bounds__(kFftBlock)
fft_r2c_lines_kernel(const double *__restrict__ map, double2 *__restrict__ out, int N, int log2n, const double2 *__restrict__ tw, int pitch)
{
    extern __shared__ double2 fbuf[];
    const int64_t line = 2 * (int64_t)blockIdx.x;
    const double *sa = map + line * N, *sb = sa + N;
    for (int i = threadIdx.x; i < N; i += kFftBlock) fbuf[bit_reverse(i, log2n)] = make_double2(sa[i], sb[i]);
    lds_fft_stages(fbuf, N, 1, tw);
    const int nz = (N >> 1) + 1;
    double2 *da = out + line * pitch, *db = da + pitch;                  // rows of `pitch` >= nz complex values
    for (int k = threadIdx.x; k < nz; k += kFftBlock) {
        const double2 z = fbuf[k], zc = fbuf[(N - k) & (N - 1)];    // Z_{N-k}, with Z_N = Z_0
        da[k] = make_double2(0.5 * (z.x + zc.x), 0.5 * (z.y - zc.y));
        db[k] = make_double2(0.5 * (z.y + zc.y), 0.5 * (zc.x - z.x));
    }
}

// what the fused last pass needs to bin |F|^2 (pk_bin_kernel's arguments)
struct PkBins {
    const double *klin;
    double kb0, dk, inv_dk;
    int nk, b0;
    double *pk_sum, *k_sum;
    unsigned long long *counts;
    const uint8_t *tab;           // BIN = 2: the k-bin of every mode in the order the last pass reads it, 255 = outside the bins
    int tab_pitch;
};

// the k-bin of mode (a, b, c): k = sqrt(klin[a]^2 + klin[c]^2 + klin[b]^2) (the notebook's summation order), bin = floor((k - kb0) / dk);
// -1 outside [0, nk)
__device__ inline int pk_bin_of(const PkBins &pb, int a, int b, int c, double &k)
{
    const double ka = pb.klin[a], kb = pb.klin[b], kc = pb.klin[c];
    k = sqrt(add_nc(add_nc(mul_nc(ka, ka), mul_nc(kc, kc)), mul_nc(kb, kb)));
    // floor((k - kb0) / dk): the quotient through the reciprocal, the division itself only where that could change the floor
    const double ue = (k - pb.kb0) * pb.inv_dk;
    double u = floor(ue);
    if (fabs(ue - rint(ue)) < 1e-6 * (fabs(ue) + 1.0)) u = floor((k - pb.kb0) / pb.dk);
    if (!(u >= 0.0) || !(u < (double)pb.nk)) return -1;
    return (int)u;
}

// one |F|^2 sample into the LDS histogram: k = sqrt(klin[a]^2 + klin[c]^2 + klin[b]^2) (the notebook's summation order),
// bin = floor((k - kb0) / dk); the c-mirrored half of the spectrum counts through a weight of 2
__device__ inline void pk_bin_sample(const PkBins &pb, int N, int a, int b, int c, double2 f, double *hp, double *hk, unsigned long long *hc)
{
    double k;
    const int bin = pk_bin_of(pb, a, b, c, k);
    if (bin < 0) return;
    const double p = f.x * f.x + f.y * f.y;
    const int mult = (c > 0 && c < (N >> 1)) ? 2 : 1;
    atomicAdd(hp + bin, mult * p);
    atomicAdd(hk + bin, mult * k);
    atomicAdd(hc + bin, (unsigned long long)mult);
}

// Which bin a mode falls into, how many modes a bin holds and the sum of their |k| depend on (N, L, nk) alone, not on the map: they are
// tabulated once per configuration (like the twiddles) -- the bin of every mode as one byte [b][a][c] and, per column b of the middle
// axis, the bins' counts and |k| sums (a slab rank adds up the columns it owns) -- so that the last pass of a power spectrum only
// looks the bin up and adds |F|^2 (no sqrt, no floor, one LDS atomic per mode instead of three).  One workgroup per b.
__global__ void __launch_bounds__(kFftBlock)
pk_table_build_kernel(PkBins pb, int N, int nz, int lt, int nzt, uint8_t *__restrict__ tab, double *__restrict__ ksum_b,
                      unsigned long long *__restrict__ cnt_b)
{
    // layout = the order in which the last pass reads it: [b][kz tile][i][l], i the (permuted) row slot of the tile: a wave's 64 look-ups
    // are 64 consecutive bytes
    extern __shared__ double2 fbuf[];
    double *hk = reinterpret_cast<double *>(fbuf);
    unsigned long long *hc = reinterpret_cast<unsigned long long *>(hk + pb.nk);
    const int b = blockIdx.x;
    for (int i = threadIdx.x; i < pb.nk; i += kFftBlock) { hk[i] = 0.0; hc[i] = 0ull; }
    __syncthreads();
    const int per_tile = N << lt;
    for (int t = threadIdx.x; t < nzt * per_tile; t += kFftBlock) {
        const int kzt = t / per_tile, r = t - kzt * per_tile;
        const int l = r & ((1 << lt) - 1), i = r >> lt;
        const int a = (i * 37) & (N - 1), c = (kzt << lt) + l;
        int bin = -1;
        if (c < nz) {
            double k;
            bin = pk_bin_of(pb, a, b, c, k);
            if (bin >= 0) {
                const int mult = (c > 0 && c < (N >> 1)) ? 2 : 1;
                atomicAdd(hk + bin, mult * k);
                atomicAdd(hc + bin, (unsigned long long)mult);
            }
        }
        tab[((size_t)b * nzt + kzt) * per_tile + r] = (bin < 0) ? (uint8_t)255 : (uint8_t)bin;
    }
    __syncthreads();
    for (int i = threadIdx.x; i < pb.nk; i += kFftBlock) { ksum_b[(size_t)b * pb.nk + i] = hk[i]; cnt_b[(size_t)b * pb.nk + i] = hc[i]; }
}

// k_sum[bin] += sum over the columns [b0, b0 + nb) of ksum_b, counts likewise (the tabulated part of a power spectrum's sums): one
// workgroup per bin, its threads over the columns
__global__ void __launch_bounds__(256)
pk_table_sums_kernel(int nk, int b0, int nb, const double *__restrict__ ksum_b, const unsigned long long *__restrict__ cnt_b,
                     double *__restrict__ k_sum, unsigned long long *__restrict__ counts)
{
    __shared__ double sk[256 / kWave];
    __shared__ unsigned long long sc[256 / kWave];
    const int i = blockIdx.x;
    double ks = 0.0;
    unsigned long long cs = 0ull;
    for (int b = b0 + threadIdx.x; b < b0 + nb; b += 256) { ks += ksum_b[(size_t)b * nk + i]; cs += cnt_b[(size_t)b * nk + i]; }
#pragma unroll
    for (int s = kWave >> 1; s > 0; s >>= 1) { ks += __shfl_down(ks, s, kWave); cs += __shfl_down(cs, s, kWave); }
    if ((threadIdx.x & (kWave - 1)) == 0) { sk[threadIdx.x / kWave] = ks; sc[threadIdx.x / kWave] = cs; }
    __syncthreads();
    if (threadIdx.x == 0) {
        double tk = 0.0;
        unsigned long long tc = 0ull;
        for (int w = 0; w < 256 / kWave; ++w) { tk += sk[w]; tc += sc[w]; }
        k_sum[i] += tk; counts[i] += tc;
    }
}

// passes 2 and 3: complex transforms along a strided axis.  Element i of line l of tile (o, kz0) sits at
// data[o * outer_stride + i * stride + kz0 + l]; the workgroups stride over the nzt * nouter tiles (kz tile fastest).
// BIN = false: in place.  BIN = true (last pass of the power spectrum): nothing is written back -- the transformed tile goes
// straight from LDS into the workgroup's k-bin histogram (the spectrum itself is not an output), which is flushed once at the
// end; the line index i is then the FIRST array axis and o the column b0 + o of the middle axis.
// the butterflies of lds_fft_stages for a compile-time number of interleaved lines (1 << LT) and twiddles held in LDS: the index
// arithmetic is shifts and masks (with a run-time line count it was three quarters of the strided passes' instructions)
template <int LT>
__device__ inline void lds_fft_stages_t(double2 *buf, int N, const double2 *tw)
{
    constexpr int NL = 1 << LT;
    int m = 1;
    const int nq = (N >> 2) << LT;
    for (; 4 * m <= N; m <<= 2) {
        __syncthreads();
        const int ts1 = (N >> 1) / m, ts2 = (N >> 2) / m;
        for (int t = threadIdx.x; t < nq; t += kFftBlock) {
            const int l = t & (NL - 1), b = t >> LT;
            const int pos = b & (m - 1);
            const int p = ((b - pos) << 2) + pos;
            const double2 w1 = tw[pos * ts1], w2 = tw[pos * ts2];
            const double2 w3 = make_double2(w2.y, -w2.x);
            double2 *q = buf + (p << LT) + l;
            const double2 a0 = q[0], a1 = q[m << LT], a2 = q[(2 * m) << LT], a3 = q[(3 * m) << LT];
            const double2 t1 = cmul(w1, a1), t3 = cmul(w1, a3);
            const double2 b0 = make_double2(a0.x + t1.x, a0.y + t1.y), b1 = make_double2(a0.x - t1.x, a0.y - t1.y);
            const double2 b2 = make_double2(a2.x + t3.x, a2.y + t3.y), b3 = make_double2(a2.x - t3.x, a2.y - t3.y);
            const double2 u2 = cmul(w2, b2), u3 = cmul(w3, b3);
            q[0] = make_double2(b0.x + u2.x, b0.y + u2.y);
            q[(2 * m) << LT] = make_double2(b0.x - u2.x, b0.y - u2.y);
            q[m << LT] = make_double2(b1.x + u3.x, b1.y + u3.y);
            q[(3 * m) << LT] = make_double2(b1.x - u3.x, b1.y - u3.y);
        }
    }
    if (m < N) {
        __syncthreads();
        const int nb = (N >> 1) << LT;
        const int tstep = (N >> 1) / m;
        for (int t = threadIdx.x; t < nb; t += kFftBlock) {
            const int l = t & (NL - 1), b = t >> LT;
            const int pos = b & (m - 1);
            double2 *q = buf + ((((b - pos) << 1) + pos) << LT) + l;
            const double2 u = q[0], v = cmul(tw[pos * tstep], q[m << LT]);
            q[0] = make_double2(u.x + v.x, u.y + v.y);
            q[m << LT] = make_double2(u.x - v.x, u.y - v.y);
        }
    }
    __syncthreads();
}

// LDS of one workgroup of the strided passes: the tile, the twiddles, (BIN) the histogram
__host__ __device__ inline size_t fft_c2c_lds_bytes(int N, int lt, int nk) { return sizeof(double2) * (((size_t)N << lt) + (size_t)(N >> 1)) + sizeof(double) * 3 * (size_t)nk; }

// BIN: 0 = in place; 1 = binned from LDS, bins computed per mode; 2 = binned from LDS, bins from the table (PkBins::tab)
template <int BIN, int LT>
__global__ void __launch_bounds__(kFftBlock)
fft_c2c_strided_kernel(double2 *__restrict__ data, int N, int log2n, int nz, int64_t stride, int64_t outer_stride,
                       const double2 *__restrict__ tw, int nzt, int nouter, PkBins pb)
{
    // a tile = the (1 << LT) adjacent lines kz0 .. kz0 + (1 << LT) of one `o`; rows are padded (fft_pitch), so the last tile of
    // a row reads up to 7 pad columns: lines of their own, transformed like the others and never binned or looked at
    constexpr int NL = 1 << LT;
    extern __shared__ double2 fbuf[];
    double2 *twl = fbuf + ((size_t)N << LT);
    double *hp = reinterpret_cast<double *>(twl + (N >> 1)), *hk = hp + pb.nk;
    unsigned long long *hc = reinterpret_cast<unsigned long long *>(hk + pb.nk);
    for (int i = threadIdx.x; i < (N >> 1); i += kFftBlock) twl[i] = tw[i];
    if (BIN) {
        for (int i = threadIdx.x; i < pb.nk; i += kFftBlock) { hp[i] = 0.0; hk[i] = 0.0; hc[i] = 0ull; }
    }
    const int64_t ntiles = (int64_t)nzt * nouter;
    for (int64_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int kz0 = (int)(tl % nzt) << LT, o = (int)(tl / nzt);
        double2 *base = data + (int64_t)o * outer_stride + kz0;
        __syncthreads();                                             // (the previous tile's readers are done with fbuf)
        for (int t = threadIdx.x; t < (N << LT); t += kFftBlock) {
            const int l = t & (NL - 1), i = t >> LT;
            fbuf[(bit_reverse(i, log2n) << LT) + l] = base[(int64_t)i * stride + l];
        }
        lds_fft_stages_t<LT>(fbuf, N, twl);
        for (int t = threadIdx.x; t < (N << LT); t += kFftBlock) {
            const int l = t & (NL - 1), i = t >> LT;
            if (BIN) {
                // neighbouring lanes take rows 37 apart (an odd multiplier permutes the rows): rows next to each other have
                // almost the same |k| and would serialise on one histogram cell
                const int a = (i * 37) & (N - 1);
                if (BIN == 2) {
                    const int bin = pb.tab[(((size_t)(pb.b0 + o) * nzt + (kz0 >> LT)) * N << LT) + t];  // (pad columns carry 255)
                    if (bin != 255) {
                        const double2 f = fbuf[(a << LT) + l];
                        const int c = kz0 + l;
                        atomicAdd(hp + bin, ((c > 0 && c < (N >> 1)) ? 2.0 : 1.0) * (f.x * f.x + f.y * f.y));
                    }
                } else if (kz0 + l < nz) pk_bin_sample(pb, N, a, pb.b0 + o, kz0 + l, fbuf[(a << LT) + l], hp, hk, hc);
            } else base[(int64_t)i * stride + l] = fbuf[t];
        }
    }
    if (BIN) {
        __syncthreads();
        for (int i = threadIdx.x; i < pb.nk; i += kFftBlock) {
            if (BIN == 2) { if (hp[i] != 0.0) atomicAdd(pb.pk_sum + i, hp[i]); }
            else if (hc[i]) { atomicAdd(pb.pk_sum + i, hp[i]); atomicAdd(pb.k_sum + i, hk[i]); atomicAdd(pb.counts + i, hc[i]); }
        }
    }
}

}  // namespace bfgx
