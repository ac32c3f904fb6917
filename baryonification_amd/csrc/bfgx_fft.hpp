// bfgx_fft.hpp -- 3-D power-spectrum summary of a gridded map (examples/10_Reproduce_Schneider_deltaPk.ipynb
// cells 12 + 15: np.fft.fftn -> |F|^2 -> np.bincount in linear k-bins) for gfx950.
//
// The map is real, so only the kz <= N/2 half of the spectrum is computed (r2c along the contiguous axis, then two
// in-place c2c passes along the strided axes); the mirrored half enters the bin sums through a weight of 2.
// Every 1-D transform runs in LDS: decimation in time, bit-reversed on the way in, two radix-2 stages fused per
// barrier (radix-4 data movement), twiddles from a host-built fp64 table; the r2c pass packs two real lines into one
// complex transform.  The strided passes move tiles of kFftTile lines that
// are adjacent in memory, so that global accesses stay contiguous (tile * 16 B) and LDS accesses conflict-free
// (line index fastest).  N must be a power of two, 8 <= N <= 1024.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "bfgx_math.hpp"

namespace bfgx {

constexpr int kFftBlock = 256;
constexpr int kFftTile = 4;           // lines per workgroup in the strided passes (measured best of 1..16 at 512^3; BFGX_FFT_TILE overrides)

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
fft_r2c_lines_kernel(const double *__restrict__ map, double2 *__restrict__ out, int N, int log2n, const double2 *__restrict__ tw)
{
    extern __shared__ double2 fbuf[];
    const int64_t line = 2 * (int64_t)blockIdx.x;
    const double *sa = map + line * N, *sb = sa + N;
    for (int i = threadIdx.x; i < N; i += kFftBlock) fbuf[bit_reverse(i, log2n)] = make_double2(sa[i], sb[i]);
    lds_fft_stages(fbuf, N, 1, tw);
    const int nz = (N >> 1) + 1;
    double2 *da = out + line * nz, *db = da + nz;
    for (int k = threadIdx.x; k < nz; k += kFftBlock) {
        const double2 z = fbuf[k], zc = fbuf[(N - k) & (N - 1)];    // Z_{N-k}, with Z_N = Z_0
        da[k] = make_double2(0.5 * (z.x + zc.x), 0.5 * (z.y - zc.y));
        db[k] = make_double2(0.5 * (z.y + zc.y), 0.5 * (zc.x - z.x));
    }
}

// passes 2 and 3: in-place complex transforms along a strided axis.  Element i of line l of tile (o, kz0) sits at
// data[o * outer_stride + i * stride + kz0 + l]; blockIdx.x = kz tile, blockIdx.y = o.
__global__ void __launch_bounds__(kFftBlock)
fft_c2c_strided_kernel(double2 *__restrict__ data, int N, int log2n, int nz, int64_t stride, int64_t outer_stride,
                       const double2 *__restrict__ tw, int tile)
{
    extern __shared__ double2 fbuf[];
    const int kz0 = blockIdx.x * tile;
    const int nl = min(tile, nz - kz0);
    double2 *base = data + (int64_t)blockIdx.y * outer_stride + kz0;
    for (int t = threadIdx.x; t < N * nl; t += kFftBlock) {
        const int l = t % nl, i = t / nl;
        fbuf[bit_reverse(i, log2n) * nl + l] = base[(int64_t)i * stride + l];
    }
    lds_fft_stages(fbuf, N, nl, tw);
    for (int t = threadIdx.x; t < N * nl; t += kFftBlock) {
        const int l = t % nl, i = t / nl;
        base[(int64_t)i * stride + l] = fbuf[i * nl + l];
    }
}

// |F|^2, |k| and mode counts per linear k-bin.  F[a][b][c], c <= N/2; k = sqrt(klin[a]^2 + klin[c]^2 + klin[b]^2)
// (the notebook's summation order); bin = floor((k - kb0) / dk); the c-mirrored half counts through a weight of 2.
// F may hold only the columns b0 <= b < b0 + nb of the middle axis (slab decomposition after the transpose: [N][nb][nz]).
__global__ void __launch_bounds__(256)
pk_bin_kernel(const double2 *__restrict__ F, int N, const double *__restrict__ klin, double kb0, double dk, int nk,
              double *__restrict__ pk_sum, double *__restrict__ k_sum, unsigned long long *__restrict__ counts, int nb, int b0)
{
    extern __shared__ double hist[];                       // [nk] power, [nk] k, then [nk] counts (u64)
    double *hp = hist, *hk = hist + nk;
    unsigned long long *hc = reinterpret_cast<unsigned long long *>(hist + 2 * nk);
    for (int i = threadIdx.x; i < nk; i += blockDim.x) { hp[i] = 0.0; hk[i] = 0.0; hc[i] = 0ull; }
    __syncthreads();
    const int nz = (N >> 1) + 1;
    const int64_t total = (int64_t)N * nb * nz;
    for (int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int c = (int)(t % nz);
        const int64_t q = t / nz;
        const int b = b0 + (int)(q % nb), a = (int)(q / nb);
        const double ka = klin[a], kb = klin[b], kc = klin[c];
        const double k = sqrt(add_nc(add_nc(mul_nc(ka, ka), mul_nc(kc, kc)), mul_nc(kb, kb)));
        const double u = floor((k - kb0) / dk);
        if (!(u >= 0.0) || !(u < (double)nk)) continue;
        const int bin = (int)u;
        const double2 f = F[t];
        const double p = f.x * f.x + f.y * f.y;
        const int mult = (c > 0 && c < (N >> 1)) ? 2 : 1;
        atomicAdd(hp + bin, mult * p);
        atomicAdd(hk + bin, mult * k);
        atomicAdd(hc + bin, (unsigned long long)mult);
    }
    __syncthreads();
    for (int i = threadIdx.x; i < nk; i += blockDim.x) {
        if (hc[i]) { atomicAdd(pk_sum + i, hp[i]); atomicAdd(k_sum + i, hk[i]); atomicAdd(counts + i, hc[i]); }
    }
}

}  // namespace bfgx
