// FFTLog Hankel transforms on gfx950 for the pixel-window convolution of profiles:
//   BaryonForge/utils/Pixel.py:106-157 (ConvolvedProfile.real), :160-224 (.projected), which call
//   pyccl.pyutils._fftlog_transform twice (forward, x window, back) and then a PCHIP in ln r.
// The transform is Hamilton's FFTLog (MNRAS 312 (2000) 257, App. B): on a log grid
//     b = reverse( IFFT( FFT(a) u ) ),   u_m = (k_c r_c)^(-2 pi i m / L) 2^z Gamma((mu+1+z)/2) / Gamma((mu+1-z)/2),  z = q + 2 pi i m / L.
// The u coefficients (complex log-gamma) are computed on the host (bfgx_api.hip); the device does the two DFTs.
// The grid lengths are n = n_per_decade * decades (~1100, not a power of two) and there are only N_M ~ 30 rows per
// call, so each row is one workgroup doing plain O(n^2) DFT sums from LDS (real input -> half spectrum -> real output;
// 2.4 M complex multiply-adds per row, tens of microseconds) -- exact to a few ulp for any n, no radix restrictions.
#pragma once
#include <hip/hip_runtime.h>
#include "bfgx_math.hpp"
#include "bfgx_tables.hpp"

namespace bfgx {

constexpr int kFhtThreads = 256;
constexpr int kFhtMaxN = 4096;        // 32 n bytes of LDS per row

__host__ __device__ inline size_t fht_lds_bytes(int n) { return sizeof(double) * ((size_t)3 * n + 2 * ((size_t)n / 2 + 1)); }

// out[row][j] = post[j] * b[j],  b = reverse(irfft(rfft(in[row] * pre) * u)),  u[0 .. n/2] complex (re, im)
__global__ void __launch_bounds__(kFhtThreads)
fht_rows_kernel(int n, const double *__restrict__ in, const double *__restrict__ pre, const double2 *__restrict__ u,
                const double *__restrict__ post, double *__restrict__ out)
{
    extern __shared__ __align__(16) unsigned char smem[];
    double *a = reinterpret_cast<double *>(smem);      // [n]
    double *tc = a + n, *ts = tc + n;                  // cos / sin(2 pi j / n)
    double *br = ts + n, *bi = br + (n / 2 + 1);       // half spectrum times u
    const int row = blockIdx.x, tid = threadIdx.x;
    const int nh = n / 2;
    const double w = 6.283185307179586476925286766559005768394 / (double)n;
    for (int j = tid; j < n; j += kFhtThreads) {
        a[j] = in[(size_t)row * n + j] * (pre ? pre[j] : 1.0);
        double s, c;
        sincos_bounded(w * (double)j, s, c);
        tc[j] = c; ts[j] = s;
    }
    __syncthreads();
    // forward: B_m = sum_j a_j exp(-2 pi i j m / n), m = 0 .. n/2; then times u_m
    for (int m = tid; m <= nh; m += kFhtThreads) {
        double re = 0.0, im = 0.0;
        int idx = 0;
        for (int j = 0; j < n; ++j) {
            const double aj = a[j];
            re = __builtin_fma(aj, tc[idx], re);
            im = __builtin_fma(-aj, ts[idx], im);
            idx += m;
            if (idx >= n) idx -= n;
        }
        const double2 um = u[m];
        br[m] = re * um.x - im * um.y;
        bi[m] = re * um.y + im * um.x;
    }
    __syncthreads();
    // inverse (numpy irfft): c_p = (1/n) [ Re B_0 + 2 sum_{0<m<n/2} Re(B_m e^{+2 pi i m p / n}) + (n even) Re B_{n/2} (-1)^p ]
    const int mtop = (n % 2 == 0) ? nh - 1 : nh;
    const double invn = 1.0 / (double)n;
    for (int p = tid; p < n; p += kFhtThreads) {
        double acc = 0.0;
        int idx = 0;
        for (int m = 1; m <= mtop; ++m) {
            idx += p;
            if (idx >= n) idx -= n;
            acc = __builtin_fma(br[m], tc[idx], acc);
            acc = __builtin_fma(-bi[m], ts[idx], acc);
        }
        double c = br[0] + 2.0 * acc;
        if (n % 2 == 0) c += (p & 1) ? -br[nh] : br[nh];
        const int j = n - 1 - p;                       // the transform comes out reversed
        out[(size_t)row * n + j] = c * invn * (post ? post[j] : 1.0);
    }
}

// out[row][i] = PchipInterpolator(x, y[row], extrapolate=False)(q[i]), NaN -> 0, times scale   (Pixel.py:154-155, :221-222)
__global__ void __launch_bounds__(256)
pchip_rows_kernel(int n, const double *__restrict__ x, const double *__restrict__ y, int nq, const double *__restrict__ q,
                  double scale, double *__restrict__ out)
{
    const int row = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= nq) return;
    double v = pchip_eval(x, y + (size_t)row * n, n, q[i]);
    if (v != v) v = 0.0;
    out[(size_t)row * nq + i] = v * scale;
}

}  // namespace bfgx
