// Internal launch interfaces between the C ABI (be_api.hip) and the kernels.
#pragma once
#include <hip/hip_runtime.h>

#include "be_common.h"
#include "icnn_be.h"

namespace icnn_be {

struct DualArgs {
    icnn_be_state st;
    const void *f;
    const void *g;
    int round;       // launch round: index into st.pending
    int budget;      // Newton updates a sample may spend in this launch; 0 = unlimited
    int n_pad;   // n rounded up to a multiple of 16 (four f64 MFMA k-steps, unrolled)
    int ldA;     // LDS row pitch of the staged bundle, in elements
    int rows;    // bundle rows the LDS staging area holds in this launch (<= round + 1)
    long long *prof;   // optional [B][DUAL_PROF_PHASES] cycle counters (diagnostic), else nullptr
    PairwisePlan plan;
};
constexpr int DUAL_PROF_PHASES = 16;
void set_dual_profile_buffer(long long *buf);
void set_fc_profile_buffer(long long *buf);
void set_conv_profile_buffer(long long *buf);

// Row pitch with pitch % 32 == 2: the MFMA operand gather (16 rows x 2 adjacent
// columns per 32-lane group) then touches 32 distinct LDS banks.
inline int dual_row_pitch(int n_pad) {
    int p = n_pad;
    while (p % 32 != 2) ++p;
    return p;
}

// Per-device launch configuration (a process may drive several GPUs: the caches are keyed by device ordinal and
// guarded by a mutex).  ensure_dynamic_lds raises a kernel's dynamic-LDS limit on the CURRENT device when needed;
// device_cus is the CU count of the current device.
hipError_t ensure_dynamic_lds(const void *kernel, int bytes);
int device_cus();

int dual_lds_bytes(int n, int slots, int cut_dtype, int variant, int rows = 0);
// most bundle rows (<= slots) whose staging fits the 160 KB of LDS; 0: not even one
int dual_rows_fit(int n, int slots, int cut_dtype, int variant);
size_t scratch_bytes(const icnn_be_state &st);
hipError_t launch_state_init(const icnn_be_state &st, hipStream_t stream);
hipError_t launch_mark_unfinished(const icnn_be_state &st, hipStream_t stream);
hipError_t launch_dual_step(const icnn_be_state &st, int round, int budget, const void *f, const void *g,
                            hipStream_t stream);
// narrow rows (n <= 16), variant RL: four samples per wave, one per 16-lane DPP row (be_dual_small.hip)
bool dual_step_small_fits(const icnn_be_state &st, int budget);
hipError_t launch_dual_step_small(const icnn_be_state &st, int round, const void *f, const void *g, hipStream_t stream);

hipError_t launch_fast_math(int which, const double *x, double *out, int count, hipStream_t stream);
hipError_t launch_export_active(const icnn_be_state &st, const int *row_offset, void *G_rows, double *ys_rows, double *h_rows,
                                double *lam_rows, hipStream_t stream);
hipError_t launch_implicit_feed(const icnn_be_state &st, const double *y_true, int loss, const int *row_offset,
                                double *fd_y, double *fd_v, double *fd_c, int *fd_sample, hipStream_t stream);

// ---- FC-PICNN energy / gradient --------------------------------------------------
int fc_check_model(const icnn_be_fc_model &m);
size_t fc_pack_floats(const icnn_be_fc_model &m);
int fc_pack(const icnn_be_fc_model &m, const float *const *w_yu, const float *const *w_zu, float *out);
hipError_t launch_fc_fg(const icnn_be_fc_model &m, const float *ctx, const double *y, int batch,
                        float *f, float *g, const int *finished, hipStream_t stream);

// x-only context producer and clamps (be_context.hip)
int ctx_check(const icnn_be_fc_ctx &c);
size_t ctx_work_floats(const icnn_be_fc_ctx &c, int batch);
hipError_t launch_fc_context(const icnn_be_fc_ctx &c, const float *x, int batch, float *ctx, int ctx_width, float *work,
                             hipStream_t stream);
hipError_t launch_fc_context_stage(const icnn_be_fc_ctx &c, int i, const float *x, int batch, float *ctx, int ctx_width,
                                   float *work, hipStream_t stream);
int launch_fc_context_sums(const icnn_be_fc_ctx &c, int i, int batch, float *work, double *stats, hipStream_t stream,
                           hipError_t &err);
hipError_t launch_fc_context_norm(const icnn_be_fc_ctx &c, int i, int batch, double batch_total, const double *stats,
                                  float *work, hipStream_t stream);
hipError_t launch_clamp(float *w, size_t count, int mode, hipStream_t stream);
hipError_t launch_fc_clamp(const icnn_be_fc_model &m, int mode, hipStream_t stream);
// conv PICNN: geometry of the u-path / heads and the context row layout (filled by conv_ctx_shape, be_picnn_conv.hip)
struct ConvCtxShape {
    int H, W, F[3], K[3], S[3], pad[3], oh[3], ow[3], P[3];   // P = oh * ow
    int flat, fch, ctx_width;
    int c_yu[3], c_zu[3], c_gate[5], c_zu3, c_zu4;            // column offsets inside a context row
};
int conv_ctx_shape(const icnn_be_conv_model &m, ConvCtxShape &g);
size_t conv_ctx_work_floats(const ConvCtxShape &g, int batch);
hipError_t launch_conv_context(const ConvCtxShape &g, const icnn_be_conv_ctx &c, const float *x, int batch, float *ctx,
                               float *work, hipStream_t stream);
hipError_t launch_conv_clamp(const icnn_be_conv_model &m, int mode, hipStream_t stream);

// Persistent per-tile solve (be_fused.hip); hipErrorNotSupported = shape outside this path, use the two-kernel rounds
// budget: Newton updates a sample may spend per round before it is parked (0 = unlimited, lockstep inside the tile)
hipError_t launch_fused_fc_solve(const icnn_be_fc_model &m, const float *ctx, const icnn_be_state &st, float *f_work,
                                 float *g_work, long long *dual_prof, hipStream_t stream, int tile_rows = 16, int budget = 0);
// persistent workgroup per sample or pair of samples (batches of at most two samples per CU)
hipError_t launch_fused_rows_solve(const icnn_be_fc_model &m, const float *ctx, const icnn_be_state &st, float *f_work,
                                   float *g_work, int per_wg, long long *dual_prof, hipStream_t stream, bool resume = false);
int dual_waves(int n, int cut_dtype, int variant);
long long *dual_profile_buffer();
void set_dual_trace_buffer(long long *buf);
long long *dual_trace_buffer();
constexpr int DUAL_TRACE_WORDS = 4;     // per (sample, round): phase start, dual step start, dual step end, Newton updates so far
long long *fc_profile_buffer();

// Adam inner optimiser of the RL agent, whole loop in one launch (be_adam.hip); hipErrorNotSupported = the batch
// has more tiles than a cooperative launch keeps resident
size_t adam_workspace_bytes(int batch, int n);
hipError_t launch_adam_fc(const icnn_be_fc_model &m, const float *ctx, int batch, int max_iter, double *act_best,
                          float *f_best, int *iters, void *workspace, hipStream_t stream,
                          const icnn_be_fc_ctx *cx = nullptr, const float *obs = nullptr);

// ---- conv PICNN energy / gradient -------------------------------------------------
int conv_check_model(const icnn_be_conv_model &m);
size_t conv_pack_floats(const icnn_be_conv_model &m);
size_t conv_work_floats(const icnn_be_conv_model &m, int batch);
int conv_pack(const icnn_be_conv_model &m, const float *const *w_yu, const float *const *w_yr,
              const float *const *b_yr, const float *const *w_zu, const float *w_fc3, const float *w_fc4,
              float *out);
hipError_t launch_conv_fg(const icnn_be_conv_model &m, const float *ctx, const double *y, int batch, float *f,
                          float *g, const int *skip, hipStream_t stream);

}  // namespace icnn_be
