// mm_vit_tu.hip -- translation unit of the Viterbi kernels on the row-lane form (mm_kernel_vit.hip).
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_vit.hip"

namespace mm {

template <int N4, int N2, int NJ, int VSZ>
static int launch_vit(const VitLaunch &vl, const RunParams &p, hipStream_t stream) {
    const size_t lds = 2 * size_t(VSZ) + 2 * size_t(MM_VIT_ESZ) + 4 * size_t(NJ) * 256;
    auto kernel = mm_vit_kernel<N4, N2, NJ, VSZ>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(vl.B)), dim3(1024), lds, stream, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
template <int N4, int N2>
static int launch_vit_shape(const VitLaunch &vl, const RunParams &p, hipStream_t stream) {
    // (15 waves x 6 positions x 64 lanes = 5760 rows at most: a state vector never exceeds 24 KB)
    if (vl.max_P1 <= 128) return launch_vit<N4, N2, 2, 24576>(vl, p, stream);
    return launch_vit<N4, N2, 4, 24576>(vl, p, stream);
}
int mm_launch_viterbi(const VitLaunch &vl, const RunParams &p, hipStream_t stream) {
    if (vl.max_P1 > 256 || size_t(vl.max_S1p) * 4 > 24576) return mm_fail(MM_ERR_UNSUPPORTED, "Viterbi kernel: graph too large");
    int rc;
    // (the layouts the forms are built for, mm_engine.hip vit_variant: wide x narrow positions per wave)
    if (vl.n4 == 1 && vl.n2 == 5) rc = launch_vit_shape<1, 5>(vl, p, stream);
    else if (vl.n4 == 2 && vl.n2 == 4) rc = launch_vit_shape<2, 4>(vl, p, stream);
    else if (vl.n4 == 6 && vl.n2 == 0) rc = launch_vit_shape<6, 0>(vl, p, stream);
    else return mm_fail(MM_ERR_UNSUPPORTED, "Viterbi kernel: no instance for this layout");
    if (rc) return rc;
    // back-trace: a double-buffered ring of R byte rows (each padded to 256 bytes) and, when they leave room for at least
    // 2 x 4 rows, the graph's row pointers and sources
    const int RSB = vl.bp_row;  // (the row stride the forward kernel wrote with)
    const size_t csr = (size_t(vl.max_S1p + 1) * 4 + size_t(vl.max_arcs) * 2 + 15) & ~size_t(15);
    const size_t budget = 160 * 1024;
    const bool csrl = csr + 2 * (4 * size_t(RSB) + 1024) <= budget;
    const size_t room = budget - (csrl ? csr : 0) - 2 * 1024;  // (every buffer of the ring has a spare KB: mm_vit_backtrace_kernel)
    int R = int(room / (2 * size_t(RSB)));
    R = R > 64 ? 64 : R;
    if (R < 1) return mm_fail(MM_ERR_UNSUPPORTED, "Viterbi back-trace: a row of back-pointers does not fit the LDS");
    const size_t lds = (csrl ? csr : 0) + 2 * (size_t(R) * size_t(RSB) + 1024);
    if (csrl) {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mm_vit_backtrace_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(mm_vit_backtrace_kernel<true>, dim3(unsigned(vl.B)), dim3(512), lds, stream, p, R, RSB);
    } else {
        HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(mm_vit_backtrace_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        hipLaunchKernelGGL(mm_vit_backtrace_kernel<false>, dim3(unsigned(vl.B)), dim3(512), lds, stream, p, R, RSB);
    }
    HIP_TRY(hipGetLastError());
    return MM_OK;
}

}  // namespace mm
