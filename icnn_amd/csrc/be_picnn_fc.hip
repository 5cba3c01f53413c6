// FC-PICNN energy/gradient: kernel wrapper, weight packing entry points, launcher (device code: be_picnn_fc_dev.h).
#include "be_picnn_fc_dev.h"
#include "be_picnn_fc_rows_dev.h"

namespace icnn_be {

namespace {

__global__ __launch_bounds__(NTHREADS) void fc_fg_kernel(FcArgs a) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    fc_fg_tile(a, blockIdx.x, lds);
}

// Batches of at most one sample per CU: a workgroup per sample on the VALU path (be_picnn_fc_rows_dev.h) instead of
// 16-row MFMA tiles that would be mostly empty rows on a few CUs.  Same results bit for bit.
struct FgRowsArgs {
    FcArgs fa;
    RowsLayout lay;
    int per_wg;
};
__global__ __launch_bounds__(RTHREADS) void fc_fg_rows_kernel(FgRowsArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FcArgs &fa = a.fa;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = fa.n, RF = a.lay.row_floats;
    const int s_base = blockIdx.x * a.per_wg;
    const int batch = fa.batch - s_base < a.per_wg ? fa.batch - s_base : a.per_wg;
    if (fa.finished) {                   // nothing to do if every sample of the workgroup has left the loop
        int live = 0;
        if (tid < batch) live = fa.finished[s_base + tid] == 0;
        if (!__syncthreads_or(live)) return;
    }
    rows_setup(fa, a.lay, lds, s_base, batch, tid);
    if (wave < batch) {                  // network input: y rounded to float32 like a TensorFlow feed; RL wrapper feeds 2y-1
        float *row = lds + wave * RF;
        for (int j = lane; j < n; j += 64) {
            const double yd = fa.y[(size_t)(s_base + wave) * n + j];
            rows_set_input(fa, a.lay, row, j, fa.action_box ? (float)(2.0 * yd - 1.0) : (float)yd);
        }
    }
    __syncthreads();
    rows_eval(fa, a.lay, lds, batch, tid, [](int) {});
    if (wave < batch) {
        const float *row = lds + wave * RF;
        const float gscale = fa.action_box ? 2.f : 1.f;    // RL/src/icnn.py:152  grad *= 2
        if (lane == 0) fa.f[s_base + wave] = lds[a.lay.f_off + wave];
        for (int j = lane; j < n; j += 64) fa.g[(size_t)(s_base + wave) * n + j] = gscale * row[a.lay.g_off + j];
    }
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
    if (!g_fc_prof) {       // (the phase profiler instruments the tile kernel)
        const int cus = device_cus();
        FgRowsArgs r{};
        const int per_wg = (batch + cus - 1) / cus;
        const int rows_lds = rows_layout(m, per_wg <= 2 ? per_wg : 1, r.lay);
        if (per_wg <= 2 && rows_lds <= 160 * 1024) {        // at most two samples per CU
            r.fa = a;
            r.per_wg = per_wg;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fc_fg_rows_kernel), rows_lds); e != hipSuccess) return e;
            hipLaunchKernelGGL(fc_fg_rows_kernel, dim3((batch + per_wg - 1) / per_wg), dim3(RTHREADS), rows_lds, stream, r);
            return hipGetLastError();
        }
    }
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(fc_fg_kernel), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(fc_fg_kernel, dim3((batch + TM - 1) / TM), dim3(NTHREADS), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_fc_clamp(const icnn_be_fc_model &m, int mode, hipStream_t stream) {
    const PackOffsets o = pack_offsets(m);
    const int L = m.n_layers - 1;
    float *w = const_cast<float *>(m.wpack);
    for (int i = 1; i <= L; ++i) {
        if (i < L) {
            const size_t cnt = packed_floats(m.width[i - 1], m.width[i]);
            hipError_t e = launch_clamp(w + o.zu_f[i], cnt, mode, stream);
            if (e == hipSuccess) e = launch_clamp(w + o.zu_b[i], packed_floats(m.width[i], m.width[i - 1]), mode, stream);
            if (e != hipSuccess) return e;
        } else {
            hipError_t e = launch_clamp(w + o.zu_f[i], (size_t)pad16(m.width[i - 1]), mode, stream);
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

int fc_check_model(const icnn_be_fc_model &m) {
    FcArgs a{};
    int lds = 0;
    return fill_args(m, a, lds);
}

}  // namespace icnn_be
