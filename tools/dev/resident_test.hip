// Are two workgroups of a given shape resident on one compute unit?  Every workgroup spins for a fixed time; a grid of
// 512 takes as long as one of 256 if two fit a compute unit, twice as long if not.
// hipcc --offload-arch=gfx950 -O2 tools/dev/resident_test.hip -o tools/dev/resident_test && tools/dev/resident_test
#include <hip/hip_runtime.h>
#include <cstdio>
template <int V>
__global__ void __launch_bounds__(896) spin(unsigned long long ticks, int *out) {
    extern __shared__ float lds[];
    lds[threadIdx.x] = 1.f;
    // (register pressure of the kernel under test: clobbers make the compiler allocate up to that register)
    if constexpr (V == 1) asm volatile("; v61" ::: "v61");
    if constexpr (V == 2) asm volatile("; v61 s100" ::: "v61", "s100");
    if constexpr (V == 3) asm volatile("; v63 s100" ::: "v63", "s100");
    if constexpr (V == 4) asm volatile("; v71" ::: "v71");
    if constexpr (V == 5) asm volatile("; s100" ::: "s100");
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    while (__builtin_amdgcn_s_memrealtime() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
    if (threadIdx.x == 0 && lds[0] < 0.f) out[blockIdx.x] = 1;
}
int main() {
    int *out;
    hipMalloc(&out, 4096 * 4);
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    typedef void (*K)(unsigned long long, int *);
    const K kernels[] = {spin<0>, spin<1>, spin<2>, spin<3>, spin<4>, spin<5>};
    const char *names[] = {"plain", "v61", "v61 s100", "v63 s100", "v71", "s100"};
    const int bs = 896, lds = 55744;
    for (int v = 0; v < 6; ++v) {
            K spin = kernels[v];
            hipFuncSetAttribute(reinterpret_cast<const void *>(spin), hipFuncAttributeMaxDynamicSharedMemorySize, lds);
            int nb = 0;
            hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, spin, bs, lds);
            float ms[2];
            for (int k = 0; k < 2; ++k) {
                const int grid = k ? 512 : 256;
                hipLaunchKernelGGL(spin, dim3(grid), dim3(bs), lds, 0, 20000ull, out);  // warm
                hipEventRecord(a);
                hipLaunchKernelGGL(spin, dim3(grid), dim3(bs), lds, 0, 20000ull, out);  // 0.2 ms at 100 MHz
                hipEventRecord(b);
                hipEventSynchronize(b);
                hipEventElapsedTime(&ms[k], a, b);
            }
            printf("%-9s block %4d lds %5d: occupancy API %d per CU; grid 256 %.3f ms, grid 512 %.3f ms\n", names[v], bs, lds, nb, ms[0], ms[1]);
        }
    return 0;
}
