// micro-benchmark: cycles per wave64 instruction per SIMD of the float64 instructions the double-precision exact kernels
// use (v_fma_f64, v_cvt_f64_f32, v_cvt_f64_f32 + v_fma_f64 interleaved, v_ldexp_f64, v_frexp_*), 4 waves per SIMD
#include <hip/hip_runtime.h>
#include <cstdio>
template <int MODE>
__global__ void __launch_bounds__(1024) k(double *out, unsigned long long *cyc, int iters) {
    double a0 = threadIdx.x * 1e-3, a1 = a0 + 1., a2 = a0 + 2., a3 = a0 + 3.;
    float f0 = threadIdx.x * 1e-3f + 1.f, f1 = f0 + 1.f, f2 = f0 + 2.f, f3 = f0 + 3.f;
    const double w = 1.0001, c = 1e-7;
    int e0 = 1, e1 = -1;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            if (MODE == 0) {
                asm volatile("v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5\n\t"
                             "v_fma_f64 %0, %0, %4, %5\n\tv_fma_f64 %1, %1, %4, %5\n\tv_fma_f64 %2, %2, %4, %5\n\tv_fma_f64 %3, %3, %4, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(w), "v"(c));
            } else if (MODE == 1) {
                asm volatile("v_cvt_f64_f32 %0, %4\n\tv_cvt_f64_f32 %1, %5\n\tv_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\t"
                             "v_cvt_f64_f32 %0, %5\n\tv_cvt_f64_f32 %1, %6\n\tv_cvt_f64_f32 %2, %7\n\tv_cvt_f64_f32 %3, %4"
                             : "=&v"(a0), "=&v"(a1), "=&v"(a2), "=&v"(a3) : "v"(f0), "v"(f1), "v"(f2), "v"(f3));
            } else if (MODE == 2) {  // the arc: cvt the weight, fma into the running sum (4 + 4)
                double t0d, t1d, t2d, t3d;
                asm volatile("v_cvt_f64_f32 %4, %8\n\tv_cvt_f64_f32 %5, %9\n\tv_cvt_f64_f32 %6, %10\n\tv_cvt_f64_f32 %7, %11\n\t"
                             "v_fma_f64 %0, %4, %12, %0\n\tv_fma_f64 %1, %5, %12, %1\n\tv_fma_f64 %2, %6, %12, %2\n\tv_fma_f64 %3, %7, %12, %3"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3), "=&v"(t0d), "=&v"(t1d), "=&v"(t2d), "=&v"(t3d)
                             : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(c));
            } else if (MODE == 3) {
                asm volatile("v_ldexp_f64 %0, %0, %4\n\tv_ldexp_f64 %1, %1, %5\n\tv_ldexp_f64 %2, %2, %4\n\tv_ldexp_f64 %3, %3, %5\n\t"
                             "v_ldexp_f64 %0, %0, %5\n\tv_ldexp_f64 %1, %1, %4\n\tv_ldexp_f64 %2, %2, %5\n\tv_ldexp_f64 %3, %3, %4"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(e0), "v"(e1));
            } else if (MODE == 4) {
                int x0, x1, x2, x3;
                double m0, m1, m2, m3;
                asm volatile("v_frexp_exp_i32_f64 %0, %8\n\tv_frexp_exp_i32_f64 %1, %9\n\tv_frexp_exp_i32_f64 %2, %10\n\tv_frexp_exp_i32_f64 %3, %11\n\t"
                             "v_frexp_mant_f64 %4, %8\n\tv_frexp_mant_f64 %5, %9\n\tv_frexp_mant_f64 %6, %10\n\tv_frexp_mant_f64 %7, %11"
                             : "=&v"(x0), "=&v"(x1), "=&v"(x2), "=&v"(x3), "=&v"(m0), "=&v"(m1), "=&v"(m2), "=&v"(m3)
                             : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
                e0 += x0 + x1 + x2 + x3;
                a0 += m0 * 1e-30 + m1 * 1e-30 + m2 * 1e-30 + m3 * 1e-30;
            } else if (MODE == 5) {
                asm volatile("v_cvt_f32_f64 %0, %4\n\tv_cvt_f32_f64 %1, %5\n\tv_cvt_f32_f64 %2, %6\n\tv_cvt_f32_f64 %3, %7\n\t"
                             "v_cvt_f32_f64 %0, %5\n\tv_cvt_f32_f64 %1, %6\n\tv_cvt_f32_f64 %2, %7\n\tv_cvt_f32_f64 %3, %4"
                             : "=&v"(f0), "=&v"(f1), "=&v"(f2), "=&v"(f3) : "v"(a0), "v"(a1), "v"(a2), "v"(a3));
            } else if (MODE == 7) {  // ONE dependent chain: acc = fma(cvt(w), x, acc), 8 arcs
                double t0d;
                asm volatile("v_cvt_f64_f32 %1, %2\n\tv_fma_f64 %0, %1, %6, %0\n\tv_cvt_f64_f32 %1, %3\n\tv_fma_f64 %0, %1, %6, %0\n\t"
                             "v_cvt_f64_f32 %1, %4\n\tv_fma_f64 %0, %1, %6, %0\n\tv_cvt_f64_f32 %1, %5\n\tv_fma_f64 %0, %1, %6, %0"
                             : "+v"(a0), "=&v"(t0d) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(c));
            } else if (MODE == 8) {  // TWO chains
                double t0d, t1d;
                asm volatile("v_cvt_f64_f32 %2, %4\n\tv_cvt_f64_f32 %3, %5\n\tv_fma_f64 %0, %2, %8, %0\n\tv_fma_f64 %1, %3, %8, %1\n\t"
                             "v_cvt_f64_f32 %2, %6\n\tv_cvt_f64_f32 %3, %7\n\tv_fma_f64 %0, %2, %8, %0\n\tv_fma_f64 %1, %3, %8, %1"
                             : "+v"(a0), "+v"(a1), "=&v"(t0d), "=&v"(t1d) : "v"(f0), "v"(f1), "v"(f2), "v"(f3), "v"(c));
            } else if (MODE == 9) {  // ONE chain of v_fma_f32 for comparison
                asm volatile("v_fma_f32 %0, %1, %5, %0\n\tv_fma_f32 %0, %2, %5, %0\n\tv_fma_f32 %0, %3, %5, %0\n\tv_fma_f32 %0, %4, %5, %0\n\t"
                             "v_fma_f32 %0, %1, %5, %0\n\tv_fma_f32 %0, %2, %5, %0\n\tv_fma_f32 %0, %3, %5, %0\n\tv_fma_f32 %0, %4, %5, %0"
                             : "+v"(f0) : "v"(f1), "v"(f2), "v"(f3), "v"(f1), "v"(f2));
            } else if (MODE == 6) {
                asm volatile("v_add_f64 %0, %0, %4\n\tv_add_f64 %1, %1, %4\n\tv_add_f64 %2, %2, %4\n\tv_add_f64 %3, %3, %4\n\t"
                             "v_mul_f64 %0, %0, %5\n\tv_mul_f64 %1, %1, %5\n\tv_mul_f64 %2, %2, %5\n\tv_mul_f64 %3, %3, %5"
                             : "+v"(a0), "+v"(a1), "+v"(a2), "+v"(a3) : "v"(c), "v"(w));
            }
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = a0 + a1 + a2 + a3 + f0 + f1 + f2 + f3 + e0;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
template <int MODE>
static void run(const char *name, double *o, unsigned long long *c) {
    const int iters = 1000;
    for (int waves : {4, 16}) {
        hipLaunchKernelGGL(k<MODE>, dim3(256), dim3(64 * waves), 0, 0, o, c, iters);
        unsigned long long h[256];
        hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
        const double per_simd_instr = double(iters) * 64.0 * (waves / 4.0);
        printf("%-28s waves/CU %2d: %.2f memtime ticks per wave-instruction per SIMD\n", name, waves, double(h[0]) / per_simd_instr);
    }
}
int main() {
    double *o; unsigned long long *c;
    hipMalloc(&o, 1024 * 256 * 8); hipMalloc(&c, 256 * 8);
    run<0>("v_fma_f64", o, c);
    run<1>("v_cvt_f64_f32", o, c);
    run<2>("cvt + fma_f64 (the arc)", o, c);
    run<3>("v_ldexp_f64", o, c);
    run<4>("v_frexp_exp/mant_f64", o, c);
    run<5>("v_cvt_f32_f64", o, c);
    run<6>("v_add_f64 / v_mul_f64", o, c);
    run<7>("cvt+fma_f64, ONE chain", o, c);
    run<8>("cvt+fma_f64, TWO chains", o, c);
    run<9>("v_fma_f32, ONE chain", o, c);
    return 0;
}
