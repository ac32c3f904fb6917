// Microbenchmark for gfx950: is a 64-lane ds_add_f64 two LDS passes by construction?  The same atomic-add loop with three address patterns:
//   consecutive   lane l -> double l + c            (K1's flush-friendly layout: 64 lanes x 8 B = 512 contiguous bytes)
//   stride 3      lane l -> double 3 l + c          ([pixel][3] layout)
//   same word     every lane -> one double          (full serialisation)
// Prints lanes per clock per CU; run it under `rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_INSTS_LDS` to see what the
// conflict counter reports for the conflict-free pattern.
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/bin/lds_add_f64_pattern scripts/ubench/lds_add_f64_pattern.hip
#include <hip/hip_runtime.h>
#include <cstdio>

template <typename T, int PATTERN>
__global__ void __launch_bounds__(256) k(T *out, int niter)
{
    __shared__ T acc[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = T(0);
    __syncthreads();
    const unsigned lane = threadIdx.x & 63u, wave = threadIdx.x >> 6;
    unsigned idx = (PATTERN == 0) ? (wave * 1024u + lane) : (PATTERN == 1 ? (wave * 1024u + 3u * lane) : wave * 1024u);
    T v = T(threadIdx.x + 1);
    for (int it = 0; it < niter; ++it) {
        atomicAdd(&acc[idx & 4095u], v);
        idx += (PATTERN == 2) ? 0u : 64u;                     // next row of the wave's own 1024-double region
        if ((idx & 1023u) < 64u * 3u && PATTERN != 2 && (idx >> 10) != wave) idx -= 1024u;
    }
    __syncthreads();
    T s = T(0);
    for (int i = threadIdx.x; i < 4096; i += 256) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename T, int PATTERN>
void run(const char *name)
{
    const int blocks = 256 * 4, niter = 20000;
    T *d; hipMalloc(&d, sizeof(T) * blocks * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, PATTERN>), dim3(blocks), dim3(256), 0, 0, d, niter);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    const double lanes = (double)blocks * 256 * niter;
    printf("%-34s %8.3f ms  %6.2f lanes/clk/CU = %5.1f clk per 64-lane instruction (2.4 GHz, 256 CUs)\n", name, ms,
           lanes / (ms * 1e-3) / 256 / 2.4e9, 64.0 / (lanes / (ms * 1e-3) / 256 / 2.4e9));
    hipFree(d);
}

int main()
{
    run<double, 0>("ds_add_f64 consecutive doubles");
    run<double, 1>("ds_add_f64 stride 3");
    run<double, 2>("ds_add_f64 one address");
    run<unsigned long long, 0>("ds_add_u64 consecutive");
    run<unsigned, 0>("ds_add_u32 consecutive");
    run<float, 0>("ds_add_f32 consecutive");
    return 0;
}
