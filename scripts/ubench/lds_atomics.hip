// Microbenchmark: LDS atomic-add throughput on gfx950 for f32 / u32 / u64 / f64 and plain RMW.
// Each block of 256 threads does NITER wave-instructions of atomics into a 16 KB LDS array with a
// stride-3 pattern (like the [pixel][3] accumulators).  Prints lanes per clock per CU.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <typename T, int KIND>
__global__ void __launch_bounds__(256) k(T *out, int niter)
{
    __shared__ T acc[4096];
    for (int i = threadIdx.x; i < 4096; i += 256) acc[i] = T(0);
    __syncthreads();
    unsigned idx = (threadIdx.x * 3u + blockIdx.x) & 4095u;
    T v = T(threadIdx.x + 1);
    for (int it = 0; it < niter; ++it) {
        if (KIND == 0) atomicAdd(&acc[idx], v);
        else { T o = acc[idx]; acc[idx] = o + v; }          // plain read-modify-write
        idx = (idx + 193u) & 4095u;
    }
    __syncthreads();
    T s = T(0);
    for (int i = threadIdx.x; i < 4096; i += 256) s += acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <typename T, int KIND>
void run(const char *name)
{
    const int blocks = 256 * 4, niter = 20000;
    T *d; hipMalloc(&d, sizeof(T) * blocks * 256);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipLaunchKernelGGL((k<T, KIND>), dim3(blocks), dim3(256), 0, 0, d, 100);
    hipEventRecord(a);
    hipLaunchKernelGGL((k<T, KIND>), dim3(blocks), dim3(256), 0, 0, d, niter);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double lanes = (double)blocks * 256 * niter;
    printf("%-28s %8.3f ms  %7.2f Glane-ops/s  %6.2f lanes/clk/CU (2.4 GHz, 256 CUs)\n", name, ms, lanes / ms / 1e6,
           lanes / (ms * 1e-3) / 256 / 2.4e9);
    hipFree(d);
}

int main()
{
    run<float, 0>("ds_add_f32 (atomicAdd)");
    run<unsigned, 0>("ds_add_u32 (atomicAdd)");
    run<unsigned long long, 0>("ds_add_u64 (atomicAdd)");
    run<double, 0>("ds_add_f64 (atomicAdd)");
    run<float, 1>("f32 plain RMW");
    run<double, 1>("f64 plain RMW");
    return 0;
}
