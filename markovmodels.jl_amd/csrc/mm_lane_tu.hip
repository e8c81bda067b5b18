// mm_lane_tu.hip -- translation unit of the lane kernel (mm_kernel_lane.hip: graphs of up to 64 states): instances and launch.
#define MM_SECONDARY_TU
#include "mm_internal.h"
#include "mm_kernel_lane.hip"

namespace mm {

size_t mm_lane_dev_bytes() { return sizeof(LaneDev); }
// fill a host image of a LaneDev from device pointers (the struct's layout stays in this translation unit)
void mm_lane_dev_fill(void *dst, const double *w0, const double *w1, const float *init, const float *fin, const int *s2p, const int *pdf_ptr,
                      const int *pdf_states, int S, int P, int ident) {
    LaneDev d;
    d.w[0] = w0;
    d.w[1] = w1;
    d.init = init;
    d.fin = fin;
    d.s2p = s2p;
    d.pdf_ptr = pdf_ptr;
    d.pdf_states = pdf_states;
    d.S = S;
    d.P = P;
    d.ident = ident;
    d.pad = 0;
    *static_cast<LaneDev *>(dst) = d;
}
template <int NS>
static int launch_lane_ns(int64_t B, const RunParams &p, hipStream_t stream) {
    hipLaunchKernelGGL(mm_lane_kernel<NS>, dim3(unsigned(B)), dim3(256), 0, stream, p);
    HIP_TRY(hipGetLastError());
    return MM_OK;
}
// max_S: the most real states of an FSM of the batch (<= 64): the product runs over that many lanes, rounded up
int mm_launch_lane(int64_t B, int max_S, const RunParams &p, hipStream_t stream) {
    if (max_S <= 8) return launch_lane_ns<8>(B, p, stream);
    if (max_S <= 16) return launch_lane_ns<16>(B, p, stream);
    if (max_S <= 32) return launch_lane_ns<32>(B, p, stream);
    if (max_S <= 64) return launch_lane_ns<64>(B, p, stream);
    return mm_fail(MM_ERR_UNSUPPORTED, "lane kernel: more than 64 states");
}

}  // namespace mm
