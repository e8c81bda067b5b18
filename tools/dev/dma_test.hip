// micro-test: does LDS-DMA (global_load_lds_dwordx4 / dword) reach LDS addresses >= 64 KiB through M0?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__device__ __forceinline__ void dma_b128(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__device__ __forceinline__ void dma_b32(const void *gsrc, unsigned lds_dst) {
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}
__global__ void k(const float *src, float *out, unsigned dst) {
    extern __shared__ float lds[];
    const int lane = threadIdx.x;
    for (unsigned q = lane; q < 160 * 256; q += 64) lds[q] = -1.f;
    __syncthreads();
    dma_b128(src + 4 * lane, dst);
    dma_b32(src + 1000 + lane, dst + 2048);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    for (int j = 0; j < 4; ++j) out[lane * 4 + j] = lds[dst / 4 + lane * 4 + j];
    out[256 + lane] = lds[(dst + 2048) / 4 + lane];
}
int main() {
    std::vector<float> h(2048);
    for (int i = 0; i < 2048; ++i) h[i] = float(i);
    float *d, *o;
    hipMalloc(&d, 8192); hipMalloc(&o, 4096);
    hipMemcpy(d, h.data(), 8192, hipMemcpyHostToDevice);
    hipFuncSetAttribute((const void *)k, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (unsigned dst : {4096u, 60000u & ~15u, 65536u, 100000u & ~15u, 150000u & ~15u}) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 160 * 1024, 0, d, o, dst);
        std::vector<float> r(320);
        hipMemcpy(r.data(), o, 1280, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int i = 0; i < 256; ++i) bad += r[i] != float(i);
        for (int i = 0; i < 64; ++i) bad += r[256 + i] != float(1000 + i);
        printf("dst %u: %s (r[0]=%g r[255]=%g r[256]=%g)\n", dst, bad ? "MISMATCH" : "ok", r[0], r[255], r[256]);
    }
    return 0;
}
