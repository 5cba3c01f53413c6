// FC-PICNN energy/gradient: kernel wrapper, weight packing entry points, launcher (device code: be_picnn_fc_dev.h).
#include "be_picnn_fc_dev.h"

namespace icnn_be {

namespace {

__global__ __launch_bounds__(NTHREADS) void fc_fg_kernel(FcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fc_fg_tile(a, blockIdx.x, lds);
}

}  // namespace

static long long *g_fc_prof = nullptr;
void set_fc_profile_buffer(long long *buf) { g_fc_prof = buf; }
long long *fc_profile_buffer() { return g_fc_prof; }

size_t fc_pack_floats(const icnn_be_fc_model &m) { return pack_offsets(m).total; }

int fc_pack(const icnn_be_fc_model &m, const float *const *w_yu, const float *const *w_zu, float *out) {
    const PackOffsets o = pack_offsets(m);
    const int L = m.n_layers - 1;
    for (size_t i = 0; i < o.total; ++i) out[i] = 0.f;
    for (int i = 0; i <= L; ++i) {
        const int wi = m.width[i];
        if (i < L) {
            pack_operand(w_yu[i], m.n, wi, false, out + o.yu_f[i]);      // forward:  [n] x [wi]
            pack_operand(w_yu[i], wi, m.n, true, out + o.yu_b[i]);       // backward: Wyu^T
            if (i > 0) {
                pack_operand(w_zu[i], m.width[i - 1], wi, false, out + o.zu_f[i]);
                pack_operand(w_zu[i], wi, m.width[i - 1], true, out + o.zu_b[i]);
            }
        } else {
            for (int j = 0; j < m.n; ++j) out[o.yu_f[i] + j] = w_yu[i][j];
            for (int j = 0; j < m.width[i - 1]; ++j) out[o.zu_f[i] + j] = w_zu[i][j];
        }
    }
    return 0;
}

hipError_t launch_fc_fg(const icnn_be_fc_model &m, const float *ctx, const double *y, int batch,
                        float *f, float *g, const int *finished, hipStream_t stream) {
    FcArgs a{};
    int lds = 0;
    if (fill_args(m, a, lds) != 0) return hipErrorInvalidValue;
    a.ctx = ctx; a.y = y; a.f = f; a.g = g; a.finished = finished; a.batch = batch;
    a.prof = g_fc_prof;
    static int configured = 0;
    if (lds > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(fc_fg_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, lds);
        if (e != hipSuccess) return e;
        configured = lds;
    }
    hipLaunchKernelGGL(fc_fg_kernel, dim3((batch + TM - 1) / TM), dim3(NTHREADS), lds, stream, a);
    return hipGetLastError();
}

int fc_check_model(const icnn_be_fc_model &m) {
    FcArgs a{};
    int lds = 0;
    return fill_args(m, a, lds);
}

}  // namespace icnn_be
