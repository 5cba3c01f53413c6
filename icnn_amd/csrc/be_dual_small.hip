// Dual step for narrow rows (n <= 16, variant RL): four samples per wave64 (device code: be_dual_small_dev.h).
#include "be_dual_small_dev.h"

namespace icnn_be {

namespace {

template <typename CutT, int KS>
__global__ __launch_bounds__(256) void dual_step_small_kernel(SmallArgs a) {
    const int wave = (blockIdx.x * 256 + threadIdx.x) >> 6;
    dual_step_quad_rl<CutT, KS>(a, 4 * wave, 4, a.round);
}

}  // namespace

bool dual_step_small_fits(const icnn_be_state &st, int budget) {
    return st.variant == ICNN_BE_VARIANT_RL && st.n <= 16 && st.slots <= 15 && (st.iters == 0 || st.iters == st.slots) && budget == 0 &&
           // time-sliced solves park samples mid-Newton (st.phase, st.park): the quad kernel restarts a solve from scratch, so
           // the finishing rounds of such a solve stay with the wave-per-sample kernel that honours the parked state
           !(st.flags & (ICNN_BE_FLAG_WAVE_PER_SAMPLE | ICNN_BE_FLAG_TIME_SLICE)) && dual_profile_buffer() == nullptr;
}

hipError_t launch_dual_step_small(const icnn_be_state &st, int round, const void *f, const void *g, hipStream_t stream) {
    SmallArgs a;
    a.st = st;
    a.f = f;
    a.g = g;
    a.round = round;
    const dim3 grid((st.batch + 15) / 16), block(256);
    const bool f64 = st.cut_dtype == ICNN_BE_CUT_F64;
    if (st.slots <= 5) {
        if (f64) hipLaunchKernelGGL((dual_step_small_kernel<double, 5>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((dual_step_small_kernel<float, 5>), grid, block, 0, stream, a);
    } else if (st.slots <= 7) {
        if (f64) hipLaunchKernelGGL((dual_step_small_kernel<double, 8>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((dual_step_small_kernel<float, 8>), grid, block, 0, stream, a);
    } else {
        if (f64) hipLaunchKernelGGL((dual_step_small_kernel<double, 16>), grid, block, 0, stream, a);
        else hipLaunchKernelGGL((dual_step_small_kernel<float, 16>), grid, block, 0, stream, a);
    }
    return hipGetLastError();
}

}  // namespace icnn_be
