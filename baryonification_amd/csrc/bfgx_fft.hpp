// bfgx_fft.hpp -- 3-D power-spectrum summary of a gridded map (examples/10_Reproduce_Schneider_deltaPk.ipynb
// cells 12 + 15: np.fft.fftn -> |F|^2 -> np.bincount in linear k-bins) for gfx950.
//
// The map is real, so only the kz <= N/2 half of the spectrum is computed (r2c along the contiguous axis, then two
// complex passes along the strided axes); the mirrored half enters the bin sums through a weight of 2.
// Every 1-D transform runs in LDS: decimation in time, bit-reversed on the way in, two radix-2 stages fused per
// barrier (radix-4 data movement), twiddles from a host-built fp64 table (copied to LDS in the strided passes); the r2c
// pass packs two real lines into one complex transform.  Rows of the half spectrum are padded to whole 128-byte lines
// (fft_pitch); the strided passes move tiles of 8 (or 4) lines that are adjacent in memory, so that global accesses are whole
// 128-byte lines, do their first radix-4 pass on the way into LDS and their last pass on the way out (fft_c2c_strided_kernel).
// The LAST strided pass writes nothing back: its results are binned as they leave the butterflies (|F|^2, |k|, counts per linear
// k-bin), so the spectrum crosses HBM 5 times instead of 8.
// N must be a power of two, 8 <= N <= 1024.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bfgx_math.hpp"

#ifndef BFGX_ABLPK
#define BFGX_ABLPK 0                  // 1: timing only -- the last pass keeps its histogram to itself
#endif
namespace bfgx {

constexpr int kFftBlock = 256;
constexpr int kFftTileLog2 = 3;       // 1 << this = lines per workgroup in the strided passes (BFGX_FFT_TILE = 4 | 8 overrides)

__device__ inline int bit_reverse(int i, int log2n) { return (int)(__brev((unsigned)i) >> (32 - log2n)); }

__device__ inline double2 cmul(double2 a, double2 b) { return make_double2(a.x * b.x - a.y * b.y, a.x * b.y + a.y * b.x); }

// in-place decimation-in-time butterflies over `nl` interleaved lines held in LDS as buf[i * nl + l] (input already in
// bit-reversed order).  Two radix-2 stages (half sizes m and 2m) are fused per barrier: a thread takes the four elements
// p, p+m, p+2m, p+3m of a 4m-block, so every element crosses LDS once per TWO stages; a last single stage follows when
// log2(N) is odd.  tw[k] = exp(-2 pi i k / N), k < N/2.
__device__ inline void lds_fft_stages(double2 *buf, int N, int nl, const double2 *__restrict__ tw, int tid, int nthr)
{
    int m = 1;
    const int nq = (N >> 2) * nl;
    for (; 4 * m <= N; m <<= 2) {
        __syncthreads();
        const int ts1 = (N >> 1) / m, ts2 = (N >> 2) / m;           // twiddle strides of the stages m and 2m
        for (int t = tid; t < nq; t += nthr) {
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
        for (int t = tid; t < nb; t += nthr) {
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
template <int PAIRS, int TPP>
__global__ void __launch_bounds__(PAIRS * TPP)
fft_r2c_lines_kernel(const double *__restrict__ map, double2 *__restrict__ out, int N, int log2n, const double2 *__restrict__ tw, int pitch)
{
    extern __shared__ double2 fbuf[];
    // PAIRS line pairs per workgroup, TPP threads each (the barriers are shared): a workgroup per pair is launch-bound at N <= 512
    constexpr int H = TPP;
    const int sub = threadIdx.x / H, tid = threadIdx.x % H;
    const int64_t line = 2 * (PAIRS * (int64_t)blockIdx.x + sub);
    const double *sa = map + line * N, *sb = sa + N;
    double2 *fb = fbuf + (size_t)sub * N;
    for (int i = tid; i < N; i += H) fb[bit_reverse(i, log2n)] = make_double2(sa[i], sb[i]);
    lds_fft_stages(fb, N, 1, tw, tid, H);
    const int nz = (N >> 1) + 1;
    double2 *da = out + line * pitch, *db = da + pitch;                  // rows of `pitch` >= nz complex values
    for (int k = tid; k < nz; k += H) {
        const double2 z = fb[k], zc = fb[(N - k) & (N - 1)];        // Z_{N-k}, with Z_N = Z_0
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
    // layout [b][kz tile][row a][line l]: the slice of a tile is contiguous (the last pass copies it into LDS with the tile)
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
        const int a = i, c = (kzt << lt) + l;
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

// LDS position of row p of a tile: the low two bits are XORed with the next two, so that the four rows p, p+1, p+2, p+3 of every
// aligned group of 4 -- 64 or 128 bytes each, the chunks one wave instruction of the first pass writes at a stride of four rows --
// fall into four different quarters of the banks; groups of 4 and of 16 consecutive rows (the later passes) stay permutations of themselves
__device__ inline int fft_swz(int p) { return p ^ ((p >> 2) & 3); }

// LDS of one workgroup of the strided passes: the tile, the twiddles, then the histogram (BIN 1: |F|^2, |k|, counts; BIN 2: |F|^2 and
// the tile's slice of the bin table)
__host__ __device__ inline size_t fft_c2c_lds_bytes(int N, int lt, int nk, int bin)
{
    size_t a = sizeof(double2) * (((size_t)N << lt) + (size_t)(N >> 1));
    if (bin == 1) a += sizeof(double) * 3 * (size_t)nk;
    if (bin == 2) a += sizeof(double) * (size_t)nk + ((size_t)N << lt);
    return a;
}

// passes 2 and 3: complex transforms along a strided axis.  Element i of line l of tile (o, kz0) sits at
// data[o * outer_stride + i * stride + kz0 + l]; the workgroups stride over the nzt * nouter tiles (kz tile fastest).
// Decimation in time over a tile of (1 << LT) adjacent lines held in LDS as buf[row][line]:
//   * the FIRST radix-4 pass is done on the way in: a thread loads the rows j, j + N/2, j + N/4, j + 3N/4 (the bit-reversed positions
//     4b .. 4b + 3, j = bitrev(b)), combines them in registers (the twiddles are 1 and -i) and writes the four results;
//   * radix-4 passes (two radix-2 stages per barrier) in LDS for the middle bits;
//   * the LAST pass (radix-2 when an odd number of bits is left, radix-4 otherwise) is done on the way out: its results go straight to
//     memory (BIN 0) or into the k-bin histogram (BIN 1, 2) and never back to LDS.
// N = 512: one write, three read+write passes and one read of the tile instead of one write, five read+write passes and one read,
// and four barriers per tile instead of seven.
// BIN: 0 = in place; 1 = binned, bins computed per mode; 2 = binned, bins from the table (PkBins::tab, [b][kz tile][row][line], whose
// slice for the tile is fetched into LDS together with the tile).  Binned passes write nothing back: the spectrum itself is not an
// output; the histogram is flushed once at the end; the line index is then the FIRST array axis and o the column b0 + o of the middle axis.
template <int BIN, int LT, int NT>
__global__ void __launch_bounds__(NT, NT == 1024 ? 8 : 4)       // (1024 threads: 64 VGPRs, so that the two workgroups LDS holds both run)
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
    uint8_t *tb8 = reinterpret_cast<uint8_t *>(hp + pb.nk);           // (BIN 2; BIN 1 has hk there)
    const int tid = threadIdx.x;
    for (int i = tid; i < (N >> 1); i += NT) twl[i] = tw[i];
    if (BIN) {
        for (int i = tid; i < pb.nk; i += NT) { hp[i] = 0.0; if (BIN == 1) { hk[i] = 0.0; hc[i] = 0ull; } }
    }
    const int rem = log2n - 2;                                       // bits left after the first pass (N >= 8)
    const int last_bits = (rem & 1) ? 1 : 2;
    const int nmid = (rem - last_bits) >> 1;
    const int nq = (N >> 2) << LT;
    const int64_t qs = (int64_t)(N >> 2) * stride;
    const int64_t ntiles = (int64_t)nzt * nouter;
    for (int64_t tl = blockIdx.x; tl < ntiles; tl += gridDim.x) {
        const int kz0 = (int)(tl % nzt) << LT, o = (int)(tl / nzt);
        double2 *base = data + (int64_t)o * outer_stride + kz0;
        __syncthreads();                                             // (the previous tile's readers are done with fbuf and tb8)
        if (BIN == 2) {
            const uint32_t *src = reinterpret_cast<const uint32_t *>(pb.tab + ((((size_t)(pb.b0 + o) * nzt + (kz0 >> LT)) * N) << LT));
            for (int t = tid; t < (N << LT) / 4; t += NT) reinterpret_cast<uint32_t *>(tb8)[t] = src[t];
        }
        // ---- in: four rows per thread, first radix-4 pass in registers
        for (int t = tid; t < nq; t += NT) {
            const int l = t & (NL - 1), b = t >> LT;
            const double2 *src = base + (int64_t)bit_reverse(b, rem) * stride + l;
            const double2 a0 = src[0], a2 = src[qs], a1 = src[2 * qs], a3 = src[3 * qs];
            const double2 b0 = make_double2(a0.x + a1.x, a0.y + a1.y), b1 = make_double2(a0.x - a1.x, a0.y - a1.y);
            const double2 b2 = make_double2(a2.x + a3.x, a2.y + a3.y), b3 = make_double2(a2.x - a3.x, a2.y - a3.y);
            double2 *q = fbuf + ((4 * b) << LT) + l;
            const int sw = b & 3;                                    // fft_swz(4 b + k) = 4 b + (k ^ (b & 3))
            q[(0 ^ sw) << LT] = make_double2(b0.x + b2.x, b0.y + b2.y);
            q[(2 ^ sw) << LT] = make_double2(b0.x - b2.x, b0.y - b2.y);
            q[(1 ^ sw) << LT] = make_double2(b1.x + b3.y, b1.y - b3.x);           // b1 + (-i) b3
            q[(3 ^ sw) << LT] = make_double2(b1.x - b3.y, b1.y + b3.x);
        }
        // ---- middle: radix-4 passes in LDS (half sizes m and 2m per barrier)
        int m = 4;
        for (int sgi = 0; sgi < nmid; ++sgi, m <<= 2) {
            __syncthreads();
            const int ts1 = (N >> 1) / m, ts2 = (N >> 2) / m;
            for (int t = tid; t < nq; t += NT) {
                const int l = t & (NL - 1), b = t >> LT;
                const int pos = b & (m - 1);
                const int p = ((b - pos) << 2) + pos;
                const double2 w1 = twl[pos * ts1], w2 = twl[pos * ts2];
                const double2 w3 = make_double2(w2.y, -w2.x);
                double2 *q0 = fbuf + (fft_swz(p) << LT) + l, *q1 = fbuf + (fft_swz(p + m) << LT) + l;
                double2 *q2 = fbuf + (fft_swz(p + 2 * m) << LT) + l, *q3 = fbuf + (fft_swz(p + 3 * m) << LT) + l;
                const double2 a0 = *q0, a1 = *q1, a2 = *q2, a3 = *q3;
                const double2 t1 = cmul(w1, a1), t3 = cmul(w1, a3);
                const double2 b0 = make_double2(a0.x + t1.x, a0.y + t1.y), b1 = make_double2(a0.x - t1.x, a0.y - t1.y);
                const double2 b2 = make_double2(a2.x + t3.x, a2.y + t3.y), b3 = make_double2(a2.x - t3.x, a2.y - t3.y);
                const double2 u2 = cmul(w2, b2), u3 = cmul(w3, b3);
                *q0 = make_double2(b0.x + u2.x, b0.y + u2.y);
                *q2 = make_double2(b0.x - u2.x, b0.y - u2.y);
                *q1 = make_double2(b1.x + u3.x, b1.y + u3.y);
                *q3 = make_double2(b1.x - u3.x, b1.y - u3.y);
            }
        }
        __syncthreads();
        // ---- out: the last pass; row `row` of line l of the transform leaves through `emit`
        auto emit = [&](int row, int l, double2 f) {
            if (BIN == 0) base[(int64_t)row * stride + l] = f;
            else if (BIN == 2) {
                const int bin = tb8[(row << LT) + l];                              // (pad columns carry 255)
                if (bin != 255) {
                    const int cc = kz0 + l;
                    atomicAdd(hp + bin, ((cc > 0 && cc < (N >> 1)) ? 2.0 : 1.0) * (f.x * f.x + f.y * f.y));
                }
            } else if (kz0 + l < nz) pk_bin_sample(pb, N, row, pb.b0 + o, kz0 + l, f, hp, hk, hc);
        };
        if (last_bits == 1) {
            const int h = N >> 1;
            for (int t = tid; t < (h << LT); t += NT) {
                const int l = t & (NL - 1), i0 = t >> LT;
                // binned: neighbouring lanes take rows 37 apart (an odd multiplier permutes the rows): rows next to each other have
                // almost the same |k| and would serialise on one histogram cell
                const int i = BIN ? ((i0 * 37) & (h - 1)) : i0;
                const double2 u = fbuf[(fft_swz(i) << LT) + l], v = cmul(twl[i], fbuf[(fft_swz(i + h) << LT) + l]);
                emit(i, l, make_double2(u.x + v.x, u.y + v.y));
                emit(i + h, l, make_double2(u.x - v.x, u.y - v.y));
            }
        } else {
            const int mq = N >> 2;
            for (int t = tid; t < (mq << LT); t += NT) {
                const int l = t & (NL - 1), b0_ = t >> LT;
                const int p = BIN ? ((b0_ * 37) & (mq - 1)) : b0_;
                const double2 w1 = twl[p * 2], w2 = twl[p];
                const double2 w3 = make_double2(w2.y, -w2.x);
                const double2 a0 = fbuf[(fft_swz(p) << LT) + l], a1 = fbuf[(fft_swz(p + mq) << LT) + l];
                const double2 a2 = fbuf[(fft_swz(p + 2 * mq) << LT) + l], a3 = fbuf[(fft_swz(p + 3 * mq) << LT) + l];
                const double2 t1 = cmul(w1, a1), t3 = cmul(w1, a3);
                const double2 b0 = make_double2(a0.x + t1.x, a0.y + t1.y), b1 = make_double2(a0.x - t1.x, a0.y - t1.y);
                const double2 b2 = make_double2(a2.x + t3.x, a2.y + t3.y), b3 = make_double2(a2.x - t3.x, a2.y - t3.y);
                const double2 u2 = cmul(w2, b2), u3 = cmul(w3, b3);
                emit(p, l, make_double2(b0.x + u2.x, b0.y + u2.y));
                emit(p + 2 * mq, l, make_double2(b0.x - u2.x, b0.y - u2.y));
                emit(p + mq, l, make_double2(b1.x + u3.x, b1.y + u3.y));
                emit(p + 3 * mq, l, make_double2(b1.x - u3.x, b1.y - u3.y));
            }
        }
    }
    if (BIN) {
        __syncthreads();
        for (int i = tid; i < pb.nk; i += NT) {
            if (BIN == 2) { if (hp[i] != 0.0 && !BFGX_ABLPK) atomicAdd(pb.pk_sum + i, hp[i]); }
            else if (hc[i]) { atomicAdd(pb.pk_sum + i, hp[i]); atomicAdd(pb.k_sum + i, hk[i]); atomicAdd(pb.counts + i, hc[i]); }
        }
    }
}

}  // namespace bfgx
