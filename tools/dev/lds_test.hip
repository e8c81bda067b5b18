// micro-benchmark: which lanes of a wave64 ds_read_b64 / ds_read_b32 conflict with each other on gfx950
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int W>
__global__ void __launch_bounds__(1024) k(const unsigned *addr, float *out, unsigned long long *cyc, int iters) {
    extern __shared__ float lds[];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i;
    unsigned a = addr[threadIdx.x & 63];
    float acc = 0.f;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            if (W == 8) {
                float2 v;
                asm volatile("ds_read_b64 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(j * 2048));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc += v.x;
            } else {
                float v;
                asm volatile("ds_read_b32 %0, %1 offset:%2" : "=v"(v) : "v"(a), "i"(j * 2048));
                asm volatile("s_waitcnt lgkmcnt(8)" ::: "memory");
                acc += v;
            }
        }
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}
int main() {
    float *o; unsigned long long *c; unsigned *da;
    hipMalloc(&o, 1024 * 256 * 4); hipMalloc(&c, 256 * 8); hipMalloc(&da, 256);
    const int iters = 500;
    struct Pat { const char *name; int w; unsigned (*f)(int); };
    Pat pats[] = {
        {"b64 linear 8*l", 8, [](int l) { return 8u * l; }},
        {"b64 same bankpair l, l+16 (other addr)", 8, [](int l) { return 8u * (l % 16) + 256u * (l / 16) + 0u; }},
        {"b64 same bankpair l, l+32 only", 8, [](int l) { return 8u * (l % 32) + 256u * (l / 32); }},
        {"b64 2-way inside 16 lanes (l, l+8)", 8, [](int l) { return 8u * (l % 8) + 256u * ((l % 16) / 8) + 64u * (l / 16); }},
        {"b64 2-way l, l+1 (pairs)", 8, [](int l) { return 8u * (l / 2) + 256u * (l % 2); }},
        {"b64 all lanes same bank pair (64-way)", 8, [](int l) { return 256u * l % 16384u; }},
        {"b64 stride 16 B (even bank pairs only)", 8, [](int l) { return 16u * l; }},
        {"b64 broadcast (all same address)", 8, [](int) { return 64u; }},
        {"b64 bank+1 shift 4B-mis... l*8+256*(l/32)", 8, [](int l) { return 8u * l + 256u * (l / 32); }},
        {"b32 linear 4*l", 4, [](int l) { return 4u * l; }},
        {"b32 same bank l, l+32 (other addr)", 4, [](int l) { return 4u * (l % 32) + 128u * (l / 32); }},
        {"b32 same bank l, l+16", 4, [](int l) { return 4u * (l % 16) + 128u * (l / 16); }},
        {"b32 2-way l, l+1", 4, [](int l) { return 4u * (l / 2) + 128u * (l % 2); }},
        {"b32 stride 8 B", 4, [](int l) { return 8u * l; }},
    };
    for (auto &p : pats) {
        unsigned h[64];
        for (int l = 0; l < 64; ++l) h[l] = p.f(l);
        hipMemcpy(da, h, 256, hipMemcpyHostToDevice);
        for (int threads : {64, 1024}) {
            if (p.w == 8) hipLaunchKernelGGL(k<8>, dim3(256), dim3(threads), 65536, 0, da, o, c, iters);
            else hipLaunchKernelGGL(k<4>, dim3(256), dim3(threads), 65536, 0, da, o, c, iters);
            unsigned long long hc[256];
            hipMemcpy(hc, c, sizeof(hc), hipMemcpyDeviceToHost);
            printf("%-44s %2d waves: %.2f ticks per wave-read (per CU: %.2f)\n", p.name, threads / 64, double(hc[0]) / (iters * 16.0),
                   double(hc[0]) / (iters * 16.0 * (threads / 64)));
        }
    }
    return 0;
}
