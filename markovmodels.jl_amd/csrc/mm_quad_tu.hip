// mm_quad_tu.hip -- translation unit of the quad kernels (mm_kernel_quad.hip): their instances and launches.
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_quad.hip"

namespace mm {

template <int KQ, int RPT, int PASS>
static int launch_quad_kq_rpt(const QuadLaunch *h, const RunParams &p, hipStream_t stream) {
    const size_t lds = h->lds;
    auto kernel = mm_fbq_kernel<KQ, RPT, PASS>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize,
                                int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(h->B)), dim3(64 * h->nw), lds, stream, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

template <int KQ, int PASS>
static int launch_quad_kq(const QuadLaunch *h, const RunParams &p, hipStream_t stream) {
    // rows per thread = ceil(max S1 / threads): 2 register-carried rows when that is enough
    const int NT = 64 * h->nw, rows = (h->max_S1p + NT - 1) / NT;
    if (rows <= 2) return launch_quad_kq_rpt<KQ, 2, PASS>(h, p, stream);
    return launch_quad_kq_rpt<KQ, 3, PASS>(h, p, stream);
}

template <int PASS>
static int launch_quad_pass(const QuadLaunch *h, const RunParams &p, hipStream_t stream) {
    switch (h->kq) {
        case 1: return launch_quad_kq<1, PASS>(h, p, stream);
        case 2: return launch_quad_kq<2, PASS>(h, p, stream);
        case 3: return launch_quad_kq<3, PASS>(h, p, stream);
        case 5: return launch_quad_kq<5, PASS>(h, p, stream);
        case 6: return launch_quad_kq<6, PASS>(h, p, stream);
        case 7: return launch_quad_kq<7, PASS>(h, p, stream);
        case 9: return launch_quad_kq<9, PASS>(h, p, stream);
        case 10: return launch_quad_kq<10, PASS>(h, p, stream);
        case 11: return launch_quad_kq<11, PASS>(h, p, stream);
        case 13: return launch_quad_kq<13, PASS>(h, p, stream);
        // 8-wave geometries: twice the quads per lane in twice the registers (more gathers in flight)
        case 15: return launch_quad_kq<15, PASS>(h, p, stream);
        case 17: return launch_quad_kq<17, PASS>(h, p, stream);
        case 19: return launch_quad_kq<19, PASS>(h, p, stream);
        case 21: return launch_quad_kq<21, PASS>(h, p, stream);
        case 23: return launch_quad_kq<23, PASS>(h, p, stream);
        case 25: return launch_quad_kq<25, PASS>(h, p, stream);
        case 27: return launch_quad_kq<27, PASS>(h, p, stream);
        case 29: return launch_quad_kq<29, PASS>(h, p, stream);
        default: return MM_ERR_UNSUPPORTED;
    }
}

int mm_launch_quad_pass(int pass, const QuadLaunch &ql, const RunParams &p, hipStream_t stream) {
    return pass == 0 ? launch_quad_pass<0>(&ql, p, stream) : launch_quad_pass<1>(&ql, p, stream);
}

}  // namespace mm
