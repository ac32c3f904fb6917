// VALU issue-rate microbenchmark for gfx950: wave64 instructions per SIMD-cycle for v_fma_f32, v_pk_fma_f32, v_fma_f64,
// v_log_f32 and v_cvt_f64_f32 at 1, 2, 4, 8 waves per SIMD (independent accumulators, no memory traffic).
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/bin/valu_rate scripts/ubench/valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ void __launch_bounds__(256) k(float *out, int iters, float seed)
{
    float a[8]; f2 p[8]; double d[8];
    for (int i = 0; i < 8; ++i) { a[i] = seed + i + threadIdx.x; p[i] = f2{a[i], a[i] + 1.f}; d[i] = a[i]; }
    const float m = 1.0000001f; const f2 pm = {m, m}; const double dm = 1.0000001;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = __builtin_fmaf(a[i], m, 0.5f);
                if (KIND == 1) p[i] = __builtin_elementwise_fma(p[i], pm, pm);
                if (KIND == 2) d[i] = __builtin_fma(d[i], dm, 0.5);
                if (KIND == 3) a[i] = __builtin_amdgcn_logf(a[i]) + 3.f;
                if (KIND == 4) { d[i] = (double)a[i]; a[i] = a[i] * m; }      // cvt + mul
            }
        }
    }
    float s = 0; for (int i = 0; i < 8; ++i) s += a[i] + p[i].x + p[i].y + (float)d[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int KIND> void run(const char *name, int per_inst)
{
    float *out; hipMalloc(&out, sizeof(float) * 256 * 256 * 8 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int iters = 2000;
    for (int wps : {1, 2, 4, 8}) {
        const int blocks = 256 * wps;            // 256-thread blocks = 4 waves = one per SIMD of a CU
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, 10, 1.f);
        hipEventRecord(e0);
        hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, out, iters, 1.f);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double insts = (double)iters * 64 * per_inst * wps;             // wave-instructions per SIMD
        printf("%-14s waves/SIMD %d: %.3f ms  -> %.2f ns per wave-instruction per SIMD (%.2f cycles at 2.4 GHz)\n", name, wps, ms,
               ms * 1e6 / insts, ms * 1e6 / insts * 2.4);
    }
}
int main() { run<0>("v_fma_f32", 1); run<1>("v_pk_fma_f32", 1); run<2>("v_fma_f64", 1); run<3>("v_log_f32+add", 2); run<4>("cvt_f64_f32+mul", 2); return 0; }
