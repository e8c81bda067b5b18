// mm_wave_tu.hip -- translation unit of the wave kernel (mm_kernel_wave.hip): its instances and launches.
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_wave.hip"

namespace mm {

template <int NSEG, int NJ, bool TWO = false>
static int launch_wave_one(const WaveLaunch &wl, const RunParams &p, hipStream_t stream) {
    static_assert(MM_WAVE_RS == MM_WAVE_VSZ && MM_WAVE_WAVES == MM_WAVE_NWD, "packer and kernel disagree");
    const size_t lds = 2 * size_t(MM_WAVE_SLICE);
    auto kernel = mm_wave_kernel<NSEG, NJ, TWO>;
    HIP_TRY(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
    hipLaunchKernelGGL(kernel, dim3(unsigned(wl.B)), dim3(128 * (MM_WAVE_NWD + 3)), lds, stream, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
int mm_launch_wave(const WaveLaunch &wl, const RunParams &p, hipStream_t stream) {
    // (wl.nseg: the most segments ONE wave of an agent holds)
    if (wl.nseg > 4 || wl.max_P1 > 256) return mm_fail(MM_ERR_UNSUPPORTED, "wave kernel: graph too large");
    // (more utterances than compute units: the instance of which two workgroups fit a compute unit, mm_kernel_wave.hip)
    if (wl.nseg <= 2 && wl.B > wl.n_cus) return wl.max_P1 <= 128 ? launch_wave_one<2, 2, true>(wl, p, stream) : launch_wave_one<2, 4, true>(wl, p, stream);
    if (wl.nseg <= 2) return wl.max_P1 <= 128 ? launch_wave_one<2, 2>(wl, p, stream) : launch_wave_one<2, 4>(wl, p, stream);
    return wl.max_P1 <= 128 ? launch_wave_one<4, 2>(wl, p, stream) : launch_wave_one<4, 4>(wl, p, stream);
}

}  // namespace mm
