// Returning integer atomics on 6200 scattered global counters (the halo -> tile slot reservation of K0): agent scope
// (memory-side, coherent across the 8 XCDs) vs workgroup scope on per-XCD counter sets (executed in the XCD's own L2).
// Build: hipcc -O3 --offload-arch=gfx950 -o scripts/ubench/bin/atomic_scope scripts/ubench/atomic_scope.hip
#include <hip/hip_runtime.h>
#include <cstdio>

__device__ inline unsigned xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | 20) & 7u; }      // HW_REG_XCC_ID[3:0]

template <int SCOPE>
__global__ void __launch_bounds__(256) k(unsigned *cnt, unsigned *out, int ncnt, int per_thread)
{
    const unsigned gid = blockIdx.x * 256 + threadIdx.x;
    unsigned h = gid * 2654435761u, acc = 0;
    unsigned *base = cnt + (SCOPE == 1 ? xcc_id() * (unsigned)ncnt : 0u);
    for (int i = 0; i < per_thread; ++i) {
        h = h * 1664525u + 1013904223u;
        unsigned *p = base + (h >> 8) % (unsigned)ncnt;
        if (SCOPE == 0) acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        else acc += __hip_atomic_fetch_add(p, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    }
    out[gid] = acc;
}

int main()
{
    const int ncnt = 6200, threads = 1000000, per = 2;
    unsigned *cnt, *out;
    hipMalloc(&cnt, sizeof(unsigned) * ncnt * 8); hipMalloc(&out, sizeof(unsigned) * threads);
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int scope = 0; scope < 2; ++scope) {
        for (int rep = 0; rep < 3; ++rep) {
            hipMemset(cnt, 0, sizeof(unsigned) * ncnt * 8);
            hipEventRecord(a);
            if (scope == 0) hipLaunchKernelGGL(k<0>, dim3((threads + 255) / 256), dim3(256), 0, 0, cnt, out, ncnt, per);
            else hipLaunchKernelGGL(k<1>, dim3((threads + 255) / 256), dim3(256), 0, 0, cnt, out, ncnt, per);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            static unsigned hc[6200 * 8];
            hipMemcpy(hc, cnt, sizeof(unsigned) * ncnt * 8, hipMemcpyDeviceToHost);
            unsigned long long tot = 0; for (int i = 0; i < ncnt * 8; ++i) tot += hc[i];
            printf("%s scope: %d returning atomics on %d counters%s: %.3f ms  (%.1f G atomics/s), sum of counters %llu\n",
                   scope ? "workgroup" : "agent    ", threads * per, ncnt, scope ? " x 8 XCD sets" : "", ms, threads * per / ms / 1e6, tot);
        }
    }
    return 0;
}
