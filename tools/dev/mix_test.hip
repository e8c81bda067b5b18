// micro-benchmark: the pair kernels' inner sequence (2 x ds_read_b64, 2 x v_pk_fma_f32, s_bitcmp, s_cbranch, s_waitcnt)
// per wave at 1, 2 and 4 waves per SIMD: which part of it contends between the waves of a SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
template <int MODE>
__global__ void __launch_bounds__(1024) k(float *out, unsigned long long *cyc, int iters, unsigned mask) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = 1e-3f * i;
    unsigned a0 = (threadIdx.x * 40503u >> 3) % 2040u * 8u, a1 = (threadIdx.x * 9973u >> 2) % 2040u * 8u;  // random-ish 8-byte slots
    if (MODE & 8) { a0 = 8u * (threadIdx.x & 63); a1 = 8u * (threadIdx.x & 63) + 512u; }                    // conflict free
    f2 acc0 = {0.f, 0.f}, acc1 = {0.f, 0.f}, w = {1.0001f, 0.9999f};
    unsigned em = mask;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        f2 x[8];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[2 * j]) : "v"(a0), "i"(j * 16));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[2 * j + 1]) : "v"(a1), "i"(j * 16));
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const int s = (2 * j) % 6;
            asm volatile("s_waitcnt lgkmcnt(4)" ::: "memory");
            if (MODE & 1) {
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel_hi:[0,1,1]" : "+v"(acc0) : "v"(w), "v"(x[s]));
                asm volatile("v_pk_fma_f32 %0, %1, %2, %0 op_sel:[1,0,0] op_sel_hi:[1,1,1]" : "+v"(acc1) : "v"(w), "v"(x[s + 1]));
            }
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[s]) : "v"(a0), "i"(j * 16 + 48));
            asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(x[s + 1]) : "v"(a1), "i"(j * 16 + 48));
            if (MODE & 2) asm volatile("s_bitcmp0_b32 %0, 5\n\ts_cbranch_scc1 1f\n\ts_nop 0\n\t1:" ::"s"(em) : "scc");
            if (MODE & 4) asm volatile("s_bitcmp0_b32 %0, 5\n\ts_and_b32 %0, %0, %0" : "+s"(em) : : "scc");
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0.x + acc0.y + acc1.x + acc1.y + em;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *o; unsigned long long *c;
    hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 256 * 8);
    const int iters = 500;
    const char *names[16] = {"reads only", "reads + pk_fma", "reads + branch", "reads + pk_fma + branch", "reads + 2 SALU", "reads + pk_fma + 2 SALU", "", "",
                             "cf reads only", "cf reads + pk_fma", "cf reads + branch", "cf reads + pk_fma + branch", "cf reads + 2 SALU", "cf reads + pk_fma + 2 SALU", "", ""};
    for (int mode : {0, 1, 2, 3, 4, 5, 8, 9, 11})
        for (int threads : {64, 256, 512, 1024}) {
#define L(M) if (mode == M) hipLaunchKernelGGL(k<M>, dim3(256), dim3(threads), 65536, 0, o, c, iters, 0u);
            L(0) L(1) L(2) L(3) L(4) L(5) L(8) L(9) L(11)
            unsigned long long h[256];
            hipMemcpy(h, c, sizeof(h), hipMemcpyDeviceToHost);
            printf("%-28s %2d waves/CU: %.1f cycles per pair per wave\n", names[mode], threads / 64, double(h[0]) / (iters * 16.0));
        }
    return 0;
}
