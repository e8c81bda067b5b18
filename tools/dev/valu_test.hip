// micro-benchmark: cycles per wave64 VALU instruction on one SIMD with W waves per SIMD (v_fma_f32, v_pk_fma_f32, v_exp_f32)
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters) {
    float a0 = threadIdx.x * 1e-3f, a1 = a0 + 1.f, a2 = a0 + 2.f, a3 = a0 + 3.f, a4 = a0 + 4.f, a5 = a0 + 5.f, a6 = a0 + 6.f, a7 = a0 + 7.f;
    const float w = 1.0001f, c = 1e-7f;
    typedef float f2 __attribute__((ext_vector_type(2)));
    f2 p0 = {a0, a1}, p1 = {a2, a3}, p2 = {a4, a5}, p3 = {a6, a7}, pw = {w, w}, pc = {c, c};
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        if (MODE == 0) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_fma_f32 %0, %0, %8, %9\n\tv_fma_f32 %1, %1, %8, %9\n\tv_fma_f32 %2, %2, %8, %9\n\tv_fma_f32 %3, %3, %8, %9\n\t"
                             "v_fma_f32 %4, %4, %8, %9\n\tv_fma_f32 %5, %5, %8, %9\n\tv_fma_f32 %6, %6, %8, %9\n\tv_fma_f32 %7, %7, %8, %9"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7) : "v"(w), "v"(c));
            }
        } else if (MODE == 1) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5\n\t"
                             "v_pk_fma_f32 %0, %0, %4, %5\n\tv_pk_fma_f32 %1, %1, %4, %5\n\tv_pk_fma_f32 %2, %2, %4, %5\n\tv_pk_fma_f32 %3, %3, %4, %5"
                             : "+v"(p0), "+v"(p1), "+v"(p2), "+v"(p3) : "v"(pw), "v"(pc));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                asm volatile("v_exp_f32 %0, %0\n\tv_exp_f32 %1, %1\n\tv_exp_f32 %2, %2\n\tv_exp_f32 %3, %3\n\t"
                             "v_log_f32 %4, %4\n\tv_log_f32 %5, %5\n\tv_log_f32 %6, %6\n\tv_log_f32 %7, %7"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "+v"(a4), "+v"(a5), "+v"(a6), "+v"(a7));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + a4 + a5 + a6 + a7 + p0.x + p1.y + p2.x + p3.y;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *o; unsigned long long *c;
    hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 256 * 8);
    const int iters = 2000;
    for (int mode = 0; mode < 3; ++mode)
        for (int waves : {4, 8, 16}) {
            if (mode == 0) hipLaunchKernelGGL(k<0>, dim3(256), dim3(64 * waves), 0, 0, o, c, iters);
            if (mode == 1) hipLaunchKernelGGL(k<1>, dim3(256), dim3(64 * waves), 0, 0, o, c, iters);
            if (mode == 2) hipLaunchKernelGGL(k<2>, dim3(256), dim3(64 * waves), 0, 0, o, c, iters);
            unsigned long long h[256];
            hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
            double cy = double(h[0]);
            const double per_simd_instr = double(iters) * 64.0 * (waves / 4.0);
            printf("mode %d (%s) waves/CU %2d: %.2f cycles per wave-instruction per SIMD\n", mode, mode == 0 ? "v_fma_f32" : mode == 1 ? "v_pk_fma_f32" : "v_exp/log", waves, cy / per_simd_instr);
        }
    return 0;
}
