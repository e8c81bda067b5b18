// micro-benchmark: scalar / branch / waitcnt issue cost per CU with 16 waves
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters, unsigned mask) {
    float a0 = threadIdx.x * 1e-3f, a1 = 1.f;
    unsigned s0 = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), s1 = mask;
    const float w = 1.0001f, c = 1e-7f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (MODE == 0) {  // 4 VALU only
                asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(w), "v"(c));
            } else if (MODE == 1) {  // 4 VALU + 4 SALU
                asm volatile("v_fma_f32 %0, %0, %4, %5\n\ts_add_u32 %2, %2, 1\n\tv_fma_f32 %1, %1, %4, %5\n\ts_xor_b32 %3, %3, %2\n\tv_fma_f32 %0, %0, %4, %5\n\ts_add_u32 %2, %2, 3\n\tv_fma_f32 %1, %1, %4, %5\n\ts_xor_b32 %3, %3, %2"
                             : "+v"(a0), "+v"(a1), "+s"(s0), "+s"(s1) : "v"(w), "v"(c) : "scc");
            } else if (MODE == 2) {  // 4 VALU + bitcmp + (not taken) branch
                asm volatile("v_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\tv_fma_f32 %0, %0, %3, %4\n\tv_fma_f32 %1, %1, %3, %4\n\ts_bitcmp1_b32 %2, 5\n\ts_cbranch_scc1 1f\n\t1:"
                             : "+v"(a0), "+v"(a1), "+s"(s1) : "v"(w), "v"(c) : "scc");
            } else if (MODE == 3) {  // 4 VALU + 2 waitcnt
                asm volatile("s_waitcnt lgkmcnt(5)\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\ts_waitcnt lgkmcnt(4)\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3" : "+v"(a0), "+v"(a1) : "v"(w), "v"(c));
            } else if (MODE == 5) {  // 4 branches (not taken), SCC stale
                asm volatile("s_cbranch_scc1 1f\n\t1:\n\ts_cbranch_scc1 2f\n\t2:\n\ts_cbranch_scc1 3f\n\t3:\n\ts_cbranch_scc1 4f\n\t4:" ::: "memory");
            } else if (MODE == 6) {  // 4 VALU + 1 branch
                asm volatile("v_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\tv_fma_f32 %0, %0, %2, %3\n\tv_fma_f32 %1, %1, %2, %3\n\ts_cbranch_scc1 1f\n\t1:"
                             : "+v"(a0), "+v"(a1) : "v"(w), "v"(c));
            } else if (MODE == 7) {  // 4 taken branches
                asm volatile("s_cbranch_scc0 1f\n\ts_nop 0\n\t1:\n\ts_cbranch_scc0 2f\n\ts_nop 0\n\t2:\n\ts_cbranch_scc0 3f\n\ts_nop 0\n\t3:\n\ts_cbranch_scc0 4f\n\ts_nop 0\n\t4:" ::: "memory");
            } else {  // 8 SALU only
                asm volatile("s_add_u32 %0, %0, 1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, 3\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, 1\n\ts_xor_b32 %1, %1, %0\n\ts_add_u32 %0, %0, 3\n\ts_xor_b32 %1, %1, %0" : "+s"(s0), "+s"(s1) : : "scc");
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + s0 + s1;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *o; unsigned long long *c;
    hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 256 * 8);
    const int iters = 2000;
    const char *names[] = {"4 VALU", "4 VALU + 4 SALU", "4 VALU + bitcmp + branch", "4 VALU + 2 waitcnt", "8 SALU", "4 branches not taken", "4 VALU + 1 branch", "4 branches taken"};
    for (int threads : {64, 256, 512, 1024})
    for (int mode = 0; mode < 8; ++mode) {
        if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        if (mode == 3) hipLaunchKernelGGL(k<3>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        if (mode == 5) hipLaunchKernelGGL(k<5>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        if (mode == 6) hipLaunchKernelGGL(k<6>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        if (mode == 7) hipLaunchKernelGGL(k<7>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        if (mode == 4) hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 0, 0, o, c, iters, 0u);
        unsigned long long h[256];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        printf("%2d waves/CU %-28s: %.2f cycles per group per wave\n", threads / 64, names[mode], double(h[0]) / (iters * 16.0));
    }
    return 0;
}
