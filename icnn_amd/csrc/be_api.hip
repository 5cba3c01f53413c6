// C ABI of libicnn_be.so (see include/icnn_be.h for the contract and the reference lines
// each entry point replaces).  Everything here only validates arguments and enqueues work.
#include <hip/hip_runtime.h>
#include <cstdlib>

#include <map>
#include <mutex>
#include <utility>

#include "be_kernels.h"
#include "icnn_be.h"

namespace {
thread_local hipError_t g_last = hipSuccess;

int fail(hipError_t e) {
    g_last = e;
    return e == hipErrorInvalidValue ? ICNN_BE_EINVAL : ICNN_BE_ELAUNCH;
}

int check_state(const icnn_be_state *st) {
    if (!st) return ICNN_BE_EINVAL;
    if (st->batch < 0 || st->n < 1) return ICNN_BE_EINVAL;
    if (st->slots < 1) return ICNN_BE_EINVAL;
    if (st->slots > ICNN_BE_MAX_SLOTS || st->iters > ICNN_BE_MAX_ITERS) return ICNN_BE_ELIMIT;
    if (st->iters != 0 && st->iters < st->slots) return ICNN_BE_EINVAL;
    if (st->cut_dtype != ICNN_BE_CUT_F32 && st->cut_dtype != ICNN_BE_CUT_F64) return ICNN_BE_EINVAL;
    if (st->variant != ICNN_BE_VARIANT_DUAL && st->variant != ICNN_BE_VARIANT_RL && st->variant != ICNN_BE_VARIANT_PDIPM)
        return ICNN_BE_EINVAL;
    if (!st->y || !st->G || !st->h || !st->ys || !st->lam || !st->active || !st->count ||
        !st->n_iters || !st->finished || !st->status || !st->newton_iters || !st->t_next || !st->phase ||
        !st->skip_fg || !st->pending || !st->park)
        return ICNN_BE_EINVAL;
    /* a bundle of at least two cuts (one, for a single iteration) must fit the LDS of a workgroup */
    const int fit = icnn_be::dual_rows_fit(st->n, st->slots, st->cut_dtype, st->variant);
    if (fit < (st->slots < 2 ? st->slots : 2)) return ICNN_BE_ELIMIT;
    return 0;
}
}  // namespace

namespace icnn_be {
namespace {
std::mutex g_cfg_mutex;
std::map<std::pair<int, const void *>, int> g_lds_limit;   // (device, kernel) -> configured dynamic LDS bytes
std::map<int, int> g_cus;                                  // device -> CU count
int current_device() {
    int dev = 0;
    return hipGetDevice(&dev) == hipSuccess ? dev : 0;
}
}  // namespace

hipError_t ensure_dynamic_lds(const void *kernel, int bytes) {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cfg_mutex);
    int &have = g_lds_limit[std::make_pair(dev, kernel)];
    if (bytes <= have) return hipSuccess;
    hipError_t e = hipFuncSetAttribute(kernel, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    if (e == hipSuccess) have = bytes;
    else (void)hipGetLastError();      // do not leave the refusal behind as the "last error" of a later, successful launch
    return e;
}

int device_cus() {
    const int dev = current_device();
    std::lock_guard<std::mutex> lock(g_cfg_mutex);
    int &cus = g_cus[dev];
    if (cus == 0) {
        hipDeviceProp_t prop;
        cus = hipGetDeviceProperties(&prop, dev) == hipSuccess && prop.multiProcessorCount > 0
                  ? prop.multiProcessorCount : 256;
    }
    return cus;
}
}  // namespace icnn_be

namespace {
// rounds of { energy/gradient ; dual step } -- shared by the FC and the conv entry points
struct NoFinish { hipError_t operator()() const { return hipErrorNotSupported; } };
// finish_fn: one launch that brings every sample still behind after the T time-sliced rounds to the end at its own pace
// (hipErrorNotSupported: none available, the stragglers get whole rounds)
template <typename LaunchFg, typename FinishFn = NoFinish>
int solve_rounds(const icnn_be_state *st, float *f_work, float *g_work, hipStream_t s, LaunchFg launch_fg,
                 int lockstep_up_to = 15, FinishFn finish_fn = FinishFn()) {
    const int T = st->iters > 0 ? st->iters : st->slots;          /* outer iterations */
    /* the interior-point solve has a fixed cap of 20 iterations per round: nothing to slice */
    const bool lockstep = (st->flags & ICNN_BE_FLAG_LOCKSTEP) || st->variant == ICNN_BE_VARIANT_PDIPM ? true
                          : (st->flags & ICNN_BE_FLAG_TIME_SLICE) ? false : T <= lockstep_up_to;
    const int slice = 8;   /* Newton updates per round before a sample is parked: covers ~99 % of the solves; measured with the
                              finishing launch below (4096 samples, nIter 30): budget 4 19.1 ms, 6 12.8, 8 12.1, 12 13.8, 16 14.3 */
    int rounds = 0;
    auto one_round = [&](int budget) -> hipError_t {
        hipError_t e = launch_fg();
        if (e != hipSuccess) return e;
        e = icnn_be::launch_dual_step(*st, rounds, budget, f_work, g_work, s);
        ++rounds;
        return e;
    };
    for (int r = 0; r < T; ++r) {
        hipError_t e = one_round(lockstep ? 0 : slice);
        if (e != hipSuccess) return fail(e);
    }
    if (lockstep) return rounds;
    /* stragglers -- samples parked in a Newton loop, or behind by the rounds they were parked in.  Whole rounds for them
       cost a launch pair each and last as long as the longest Newton chain of the round (nIter = 30, 4096 samples: twelve
       rounds, 6.8 ms of a 14.2 ms solve); the persistent per-sample kernel instead lets each of them run its remaining
       rounds back to back on a CU of its own (workgroups of finished samples leave at once), without asking the host */
    if (!(st->flags & ICNN_BE_FLAG_TWO_KERNELS)) {
        hipError_t e = finish_fn();
        if (e == hipSuccess) return rounds + 1;
        if (e != hipErrorNotSupported) return fail(e);
    }
    /* No such kernel for this model (the conv PICNN: its evaluation couples sixteen samples in the 2048 x 512 layer), or
       the caller insists on launch pairs.  Nobody else is waiting any more, so no budget: every further round completes one
       outer iteration of every sample that is behind.  How many are needed -- the largest lag -- is only known on the
       device, and the host does NOT ask (no synchronisation anywhere in this library; the call stays capturable in a HIP
       graph): T more rounds are enqueued -- a sample cannot be behind by more than the T iterations it has, and an unbudgeted
       round completes one of them whatever its Newton solve takes, so T rounds always suffice --, in which the kernels of a
       sample that has nothing left to do leave at their first instruction (an empty round costs its launches, ~10 us).
       The closing launch marks anything still behind with ICNN_BE_ST_UNFINISHED (a safety net: unreachable by the argument
       above). */
    const int extra = T;
    for (int r = 0; r < extra && rounds < ICNN_BE_MAX_ROUNDS; ++r) {
        hipError_t e = one_round(0);
        if (e != hipSuccess) return fail(e);
    }
    hipError_t e = icnn_be::launch_mark_unfinished(*st, s);
    if (e != hipSuccess) return fail(e);
    return rounds;
}
}  // namespace

extern "C" {

int icnn_be_abi_version(void) { return ICNN_BE_ABI_VERSION; }

const char *icnn_be_last_hip_error(void) { return hipGetErrorString(g_last); }

size_t icnn_be_struct_size(int which) {
    return which == 0 ? sizeof(icnn_be_state) : which == 1 ? sizeof(icnn_be_fc_model)
         : which == 2 ? sizeof(icnn_be_fc_ctx) : which == 3 ? sizeof(icnn_be_conv_model)
         : which == 4 ? sizeof(icnn_be_conv_ctx) : 0;
}

/* diagnostic hooks (include/icnn_be.h): per-phase cycle counters */
void icnn_be_debug_profile(long long *device_buf) { icnn_be::set_dual_profile_buffer(device_buf); }
void icnn_be_debug_profile_fc(long long *device_buf) { icnn_be::set_fc_profile_buffer(device_buf); }
void icnn_be_debug_profile_conv(long long *device_buf) { icnn_be::set_conv_profile_buffer(device_buf); }

int icnn_be_debug_fast_math(int which, const double *x, double *out, int count, void *stream) {
    if (which < 0 || which > 3 || !x || !out || count < 0) return ICNN_BE_EINVAL;
    if (count == 0) return 0;
    hipError_t e = icnn_be::launch_fast_math(which, x, out, count, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_debug_profile_phases(void) { return icnn_be::DUAL_PROF_PHASES; }
void icnn_be_debug_trace(long long *device_buf) { icnn_be::set_dual_trace_buffer(device_buf); }

int icnn_be_bundle_capacity(int n, int slots, int cut_dtype, int variant) {
    if (n < 1 || slots < 1 || slots > ICNN_BE_MAX_SLOTS) return ICNN_BE_EINVAL;
    return icnn_be::dual_rows_fit(n, slots, cut_dtype, variant);
}

size_t icnn_be_scratch_bytes(const icnn_be_state *shape) {
    if (!shape || shape->batch < 0 || shape->n < 1 || shape->slots < 1 || shape->slots > ICNN_BE_MAX_SLOTS) return 0;
    return icnn_be::scratch_bytes(*shape);
}

int icnn_be_dual_lds_bytes(int n, int slots, int cut_dtype) {
    if (n < 1 || slots < 1 || slots > ICNN_BE_MAX_SLOTS) return ICNN_BE_EINVAL;
    return icnn_be::dual_lds_bytes(n, slots, cut_dtype, ICNN_BE_VARIANT_PDIPM);   /* the variant with the most column buffers */
}

int icnn_be_state_init(const icnn_be_state *st, void *stream) {
    if (int rc = check_state(st)) return rc;
    if (st->batch == 0) return 0;
    hipError_t e = icnn_be::launch_state_init(*st, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_dual_step(const icnn_be_state *st, int t, const void *f, const void *g, void *stream) {
    if (int rc = check_state(st)) return rc;
    if (t < 0 || t >= (st->iters > 0 ? st->iters : st->slots) || !f || !g) return ICNN_BE_EINVAL;
    if (st->batch == 0) return 0;
    /* lockstep: every unfinished sample is at outer iteration t and completes it in this launch */
    hipError_t e = icnn_be::launch_dual_step(*st, t, 0, f, g, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

size_t icnn_be_fc_pack_floats(const icnn_be_fc_model *shape) {
    if (!shape || icnn_be::fc_check_model(*shape) != 0) return 0;
    return icnn_be::fc_pack_floats(*shape);
}

int icnn_be_fc_pack(const icnn_be_fc_model *shape, const float *const *w_yu_host,
                    const float *const *w_zu_host, float *out_host) {
    if (!shape || !w_yu_host || !w_zu_host || !out_host) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::fc_check_model(*shape)) return rc;
    for (int i = 0; i < shape->n_layers; ++i) {
        if (!w_yu_host[i]) return ICNN_BE_EINVAL;
        if (i > 0 && !w_zu_host[i]) return ICNN_BE_EINVAL;
    }
    return icnn_be::fc_pack(*shape, w_yu_host, w_zu_host, out_host);
}

int icnn_be_fc_fg(const icnn_be_fc_model *model, const float *ctx, const double *y, int batch,
                  float *f, float *g, const int *finished, void *stream) {
    if (!model || !ctx || !y || !f || !g || batch < 0 || !model->wpack) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::fc_check_model(*model)) return rc;
    if (batch == 0) return 0;
    hipError_t e = icnn_be::launch_fc_fg(*model, ctx, y, batch, f, g, finished,
                                         static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_solve_fc(const icnn_be_fc_model *model, const float *ctx, const icnn_be_state *st,
                     float *f_work, float *g_work, void *stream) {
    if (int rc = check_state(st)) return rc;
    if (!model || !ctx || !f_work || !g_work || !model->wpack) return ICNN_BE_EINVAL;
    if (st->cut_dtype != ICNN_BE_CUT_F32 || st->n != model->n) return ICNN_BE_EINVAL;
    if (st->flags & ICNN_BE_FLAG_F64_ENERGY) return ICNN_BE_EINVAL;        /* the fused energies are float32 */
    if (int rc = icnn_be::fc_check_model(*model)) return rc;
    if (st->batch == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    /* lockstep rounds (the default for nIter <= 15): the persistent per-tile kernel where the shape fits it */
    const int iters = st->iters > 0 ? st->iters : st->slots;
    const bool lockstep = (st->flags & ICNN_BE_FLAG_LOCKSTEP) || (!(st->flags & ICNN_BE_FLAG_TIME_SLICE) && iters <= 15);
    bool persistent = lockstep && !(st->flags & ICNN_BE_FLAG_TWO_KERNELS);
    const int cus = icnn_be::device_cus();
    /* at most two samples per CU: a persistent workgroup per sample or pair of samples (be_fused.hip).  Every sample
       runs at its own pace there, so no time slicing is needed however many outer iterations there are. */
    const int forced = ICNN_BE_FLAG_TWO_KERNELS | ICNN_BE_FLAG_PERSISTENT | ICNN_BE_FLAG_TIME_SLICE | ICNN_BE_FLAG_LOCKSTEP;
    const int per_wg = (st->batch + cus - 1) / cus;
    const bool ipm = st->variant == ICNN_BE_VARIANT_PDIPM;   /* (round 4: the interior-point variant runs through the persistent
                                                                kernels too; its solve has a fixed cap of 20 iterations per
                                                                round, so there is nothing to time-slice: lockstep at any nIter) */
    if (ipm) persistent = !(st->flags & ICNN_BE_FLAG_TWO_KERNELS);
    if (!(st->flags & forced) && per_wg <= 4) {
        hipError_t e = icnn_be::launch_fused_rows_solve(*model, ctx, *st, f_work, g_work, per_wg,
                                                        icnn_be::dual_profile_buffer(), s);
        if (e == hipSuccess) return iters;
        if (e != hipErrorNotSupported) return fail(e);
    }
    /* a tile of 16 samples per workgroup: worth it when the CUs are neither mostly idle nor several tiles deep
       (the RL variant runs through the same kernel bit-identically but measured 8-40 % slower at every batch
       size -- its dual step is long and evenly long --, so it only takes this path when forced) */
    const int tiles = (st->batch + 15) / 16;
    const bool tile_shape = (st->variant == ICNN_BE_VARIANT_DUAL || ipm) && 4 * tiles >= cus && tiles <= 2 * cus;
    if (persistent && !(st->flags & ICNN_BE_FLAG_PERSISTENT)) persistent = tile_shape;
    /* more outer iterations than that (nIter > 15, where launch pairs are time-sliced): the SAME persistent tile kernel with the
       dual phase in groups (the bundles of sixteen samples at 15+ cuts each do not fit the LDS together, be_fused.hip).  One
       launch instead of 2 x nIter.  A tile only waits for the slowest of ITS sixteen samples, and since limit cycles at their
       rounding floor are recognised (be_dual_dev.h, NOISE_TOL) a sample's longest Newton solve is a few dozen updates, so the
       tiles run in lockstep WITHOUT the per-round update budget of the launch pairs (measured, 4096 samples, nIter 30:
       unlimited 6.8-7.0 ms, budget 8 7.4-7.6, 12 7.3-7.5; with a budget a parked sample skips phase A, resumes in its tile's
       next dual phase, and ONE finishing launch of the per-sample kernel brings the samples that are behind to the end). */
    /* (no upper limit on the tiles per CU here: 16384 samples at nIter 30 take 28.0 ms as persistent tiles, four per CU one
       after the other, against 35.8 ms as launch pairs -- tools/big_batch_long_experiment.py) */
    const bool long_tile_shape = st->variant == ICNN_BE_VARIANT_DUAL && 4 * tiles >= cus;
    if (!lockstep && !ipm && !(st->flags & ICNN_BE_FLAG_TWO_KERNELS) &&
        ((long_tile_shape && !(st->flags & ICNN_BE_FLAG_TIME_SLICE)) || (st->flags & ICNN_BE_FLAG_PERSISTENT))) {
        int tile_rows = 16;
        if (per_wg <= 8) tile_rows = per_wg <= 4 ? 4 : 8;
        static const int env_budget = [] {         /* tuning knob (tools/tile_budget_sweep.py); default measured there */
            const char *v = std::getenv("ICNN_BE_TILE_BUDGET");
            return v ? std::atoi(v) : 0;
        }();
        /* ICNN_BE_FLAG_PERSISTENT | ICNN_BE_FLAG_TIME_SLICE: the budgeted form (eight updates per round + finishing launch) */
        const int tile_budget = (st->flags & ICNN_BE_FLAG_TIME_SLICE) ? 8 : env_budget;
        hipError_t e = icnn_be::launch_fused_fc_solve(*model, ctx, *st, f_work, g_work, icnn_be::dual_profile_buffer(), s,
                                                      tile_rows, tile_budget);
        if (e == hipSuccess) {
            if (tile_budget <= 0) return iters;     /* nobody was parked: every sample is at its end */
            e = icnn_be::launch_fused_rows_solve(*model, ctx, *st, f_work, g_work, 1, icnn_be::dual_profile_buffer(), s, true);
            if (e == hipSuccess) return iters + 1;
            return fail(e);        /* (the same shapes fit both kernels: nothing to fall back to half-way) */
        }
        if (e != hipErrorNotSupported) return fail(e);
    }
    if (persistent) {
        /* five to eight samples per CU (MI355X: 1025..2048 samples, e.g. the shard of the 4096 batch on two GPUs): partial
           tiles -- 8 samples in the 16-row MFMA tile -- so that every CU has a tile and a tile's dual phase runs two waves
           per SIMD and waits for the slowest of 8 (2048 samples: 1.00 ms against 1.17 with full tiles on half the CUs,
           bit-identical; tools/partial_tiles_experiment.py).  Up to four per CU the per-sample kernel above is faster
           (0.68 / 0.85 ms at 512 / 1024 against 0.91 with 4-sample tiles, which ICNN_BE_FLAG_PERSISTENT still selects). */
        int tile_rows = 16;
        if (per_wg <= 8) tile_rows = per_wg <= 4 ? 4 : 8;
        hipError_t e = icnn_be::launch_fused_fc_solve(*model, ctx, *st, f_work, g_work, icnn_be::dual_profile_buffer(), s,
                                                      tile_rows);
        if (e == hipSuccess) return iters;
        if (e != hipErrorNotSupported) return fail(e);
    }
    return solve_rounds(st, f_work, g_work, s, [&]() {
        return icnn_be::launch_fc_fg(*model, ctx, st->y, st->batch, f_work, g_work, st->skip_fg, s);
    }, 15, [&]() {
        return icnn_be::launch_fused_rows_solve(*model, ctx, *st, f_work, g_work, 1, icnn_be::dual_profile_buffer(), s, true);
    });
}

size_t icnn_be_fc_context_work_floats(const icnn_be_fc_ctx *c, int batch) {
    if (!c || batch < 0 || icnn_be::ctx_check(*c) != 0) return 0;
    return icnn_be::ctx_work_floats(*c, batch);
}

int icnn_be_fc_context(const icnn_be_fc_ctx *c, const float *x, int batch, float *ctx, int ctx_width, float *work,
                       void *stream) {
    if (!c || !x || !ctx || !work || batch < 0) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::ctx_check(*c)) return rc;
    if (batch == 0) return 0;
    hipError_t e = icnn_be::launch_fc_context(*c, x, batch, ctx, ctx_width, work, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_fc_context_stage(const icnn_be_fc_ctx *c, int stage, const float *x, int batch, float *ctx, int ctx_width,
                             float *work, double *stats, void *stream) {
    /* an empty shard (a rank of a data-parallel group whose batch is smaller than the group) has no rows: its zero-element
       tensors have null data pointers, and it must still take part in the all-reduce of the sums */
    if (!c || batch < 0 || (batch > 0 && (!x || !ctx || !work))) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::ctx_check(*c)) return rc;
    if (stage < 0 || stage >= c->n_layers) return ICNN_BE_EINVAL;
    if (batch == 0) return stage < c->n_layers - 2 && c->batchnorm ? 1 : 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = icnn_be::launch_fc_context_stage(*c, stage, x, batch, ctx, ctx_width, work, s);
    if (e != hipSuccess) return fail(e);
    if (stage >= c->n_layers - 2 || !c->batchnorm) return 0;     /* no BatchNorm behind this stage */
    if (!stats) return ICNN_BE_EINVAL;
    icnn_be::launch_fc_context_sums(*c, stage, batch, work, stats, s, e);
    return e == hipSuccess ? 1 : fail(e);
}

int icnn_be_fc_context_norm(const icnn_be_fc_ctx *c, int stage, int batch, double batch_total, const double *stats,
                            float *work, void *stream) {
    if (!c || !stats || !work || batch < 0 || !(batch_total >= 1.0)) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::ctx_check(*c)) return rc;
    if (stage < 0 || stage >= c->n_layers - 2 || !c->batchnorm) return ICNN_BE_EINVAL;
    if (batch == 0) return 0;
    hipError_t e = icnn_be::launch_fc_context_norm(*c, stage, batch, batch_total, stats, work, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_fc_clamp(const icnn_be_fc_model *model, int mode, void *stream) {
    if (!model || !model->wpack || mode < ICNN_BE_CLAMP_ABS || mode > ICNN_BE_CLAMP_ABS_HALF) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::fc_check_model(*model)) return rc;
    hipError_t e = icnn_be::launch_fc_clamp(*model, mode, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

size_t icnn_be_adam_workspace_bytes(int batch, int n) {
    return batch < 0 || n < 1 ? 0 : icnn_be::adam_workspace_bytes(batch, n);
}

int icnn_be_adam_fc(const icnn_be_fc_model *model, const float *ctx, int batch, int max_iter, double *act_best,
                    float *f_best, int *iters, void *workspace, void *stream) {
    if (!model || !ctx || !act_best || !f_best || !iters || !workspace || !model->wpack) return ICNN_BE_EINVAL;
    if (batch < 0 || max_iter < 1 || model->action_box) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::fc_check_model(*model)) return rc;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (batch == 0) {
        hipError_t e = hipMemsetAsync(iters, 0, sizeof(int), s);
        return e == hipSuccess ? 0 : fail(e);
    }
    hipError_t e = icnn_be::launch_adam_fc(*model, ctx, batch, max_iter, act_best, f_best, iters, workspace, s);
    if (e == hipErrorNotSupported) return ICNN_BE_ELIMIT;
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_adam_fc_obs(const icnn_be_fc_model *model, const icnn_be_fc_ctx *cx, const float *obs, int batch, int max_iter,
                        double *act_best, float *f_best, int *iters, void *workspace, void *stream) {
    if (!model || !cx || !obs || !act_best || !f_best || !iters || !workspace || !model->wpack) return ICNN_BE_EINVAL;
    if (batch < 0 || max_iter < 1 || model->action_box) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::fc_check_model(*model)) return rc;
    if (int rc = icnn_be::ctx_check(*cx)) return rc;
    /* the in-kernel context producer sizes its reads from the MODEL's layer widths while the stage matrices were laid out
       for cx's: both structs must describe the same network */
    if (cx->n != model->n || cx->n_layers != model->n_layers) return ICNN_BE_EINVAL;
    for (int i = 0; i < model->n_layers; ++i)
        if (cx->width[i] != model->width[i]) return ICNN_BE_EINVAL;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (batch == 0) {
        hipError_t e = hipMemsetAsync(iters, 0, sizeof(int), s);
        return e == hipSuccess ? 0 : fail(e);
    }
    hipError_t e = icnn_be::launch_adam_fc(*model, nullptr, batch, max_iter, act_best, f_best, iters, workspace, s, cx, obs);
    if (e == hipErrorNotSupported) return ICNN_BE_ELIMIT;
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_implicit_feed(const icnn_be_state *st, const double *y_true, int loss, const int *row_offset,
                          double *fd_y, double *fd_v, double *fd_c, int *fd_sample, void *stream) {
    if (int rc = check_state(st)) return rc;
    if (!y_true || !row_offset || !fd_y || !fd_v || !fd_c || !fd_sample) return ICNN_BE_EINVAL;
    if (loss != ICNN_BE_LOSS_XENT && loss != ICNN_BE_LOSS_MSE) return ICNN_BE_EINVAL;
    if (st->batch == 0) return 0;
    hipError_t e = icnn_be::launch_implicit_feed(*st, y_true, loss, row_offset, fd_y, fd_v, fd_c, fd_sample,
                                                 static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_export_active(const icnn_be_state *st, const int *row_offset, void *G_rows, double *ys_rows, double *h_rows,
                          double *lam_rows, void *stream) {
    if (int rc = check_state(st)) return rc;
    if (!row_offset || !G_rows || !ys_rows || !h_rows || !lam_rows) return ICNN_BE_EINVAL;
    if (st->batch == 0) return 0;
    hipError_t e = icnn_be::launch_export_active(*st, row_offset, G_rows, ys_rows, h_rows, lam_rows, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

size_t icnn_be_conv_pack_floats(const icnn_be_conv_model *shape) {
    return shape ? icnn_be::conv_pack_floats(*shape) : 0;
}

size_t icnn_be_conv_work_floats(const icnn_be_conv_model *shape, int batch) {
    return shape ? icnn_be::conv_work_floats(*shape, batch) : 0;
}

int icnn_be_conv_pack(const icnn_be_conv_model *shape, const float *const *w_yu_host,
                      const float *const *w_yr_host, const float *const *b_yr_host,
                      const float *const *w_zu_host, const float *w_fc3_host, const float *w_fc4_host,
                      float *out_host) {
    if (!shape || !w_yu_host || !w_yr_host || !b_yr_host || !w_zu_host || !w_fc3_host || !w_fc4_host || !out_host)
        return ICNN_BE_EINVAL;
    for (int l = 0; l < 3; ++l) {
        if (!w_yu_host[l] || (l < 2 && (!w_yr_host[l] || !b_yr_host[l])) || (l > 0 && !w_zu_host[l]))
            return ICNN_BE_EINVAL;
    }
    return icnn_be::conv_pack(*shape, w_yu_host, w_yr_host, b_yr_host, w_zu_host, w_fc3_host, w_fc4_host, out_host);
}

int icnn_be_conv_fg(const icnn_be_conv_model *model, const float *ctx, const double *y, int batch,
                    float *f, float *g, const int *finished, void *stream) {
    if (!model || !ctx || !y || !f || !g || batch < 0 || !model->wpack) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::conv_check_model(*model)) return rc;
    if (batch == 0) return 0;
    if (!model->work || model->work_batch < batch) return ICNN_BE_EINVAL;
    hipError_t e = icnn_be::launch_conv_fg(*model, ctx, y, batch, f, g, finished, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

size_t icnn_be_conv_context_work_floats(const icnn_be_conv_model *shape, int batch) {
    icnn_be::ConvCtxShape g{};
    if (!shape || batch < 0 || icnn_be::conv_ctx_shape(*shape, g) != 0) return 0;
    return icnn_be::conv_ctx_work_floats(g, batch);
}

int icnn_be_conv_context(const icnn_be_conv_model *shape, const icnn_be_conv_ctx *c, const float *x, int batch, float *ctx,
                         float *work, void *stream) {
    if (!shape || !c || !x || !ctx || !work || batch < 0) return ICNN_BE_EINVAL;
    for (int s = 0; s < 7; ++s)
        if (!c->w_stage[s] || !c->b_stage[s]) return ICNN_BE_EINVAL;
    for (int i = 0; i < 4; ++i)
        if (!c->bn_gamma[i] || !c->bn_beta[i]) return ICNN_BE_EINVAL;
    icnn_be::ConvCtxShape g{};
    if (int rc = icnn_be::conv_ctx_shape(*shape, g)) return rc;
    if (batch == 0) return 0;
    hipError_t e = icnn_be::launch_conv_context(g, *c, x, batch, ctx, work, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_conv_clamp(const icnn_be_conv_model *model, int mode, void *stream) {
    if (!model || !model->wpack || mode < ICNN_BE_CLAMP_ABS || mode > ICNN_BE_CLAMP_ABS_HALF) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::conv_check_model(*model)) return rc;
    hipError_t e = icnn_be::launch_conv_clamp(*model, mode, static_cast<hipStream_t>(stream));
    return e == hipSuccess ? 0 : fail(e);
}

int icnn_be_solve_conv(const icnn_be_conv_model *model, const float *ctx, const icnn_be_state *st,
                       float *f_work, float *g_work, void *stream) {
    if (int rc = check_state(st)) return rc;
    if (!model || !ctx || !f_work || !g_work || !model->wpack) return ICNN_BE_EINVAL;
    if (st->cut_dtype != ICNN_BE_CUT_F32 || st->n != model->H * model->W) return ICNN_BE_EINVAL;
    if (st->flags & ICNN_BE_FLAG_F64_ENERGY) return ICNN_BE_EINVAL;
    if (st->batch > 0 && (!model->work || model->work_batch < st->batch)) return ICNN_BE_EINVAL;
    if (int rc = icnn_be::conv_check_model(*model)) return rc;
    if (st->batch == 0) return 0;
    hipStream_t s = static_cast<hipStream_t>(stream);
    /* Lockstep rounds at EVERY nIter (round 3; time slicing on request, ICNN_BE_FLAG_TIME_SLICE).  Slicing pays when a round
       is held up by a Newton solve that runs its 100-update cap; since limit cycles at their rounding floor are recognised
       (be_dual_dev.h, NOISE_TOL) those are one solve in a thousand, while every sliced solve pays nIter finishing rounds of
       five launches for its few laggards.  Measured at 256 samples (tools/conv_slice_experiment.py): nIter 30 lockstep
       12.0 / 11.0 ms against 16.4 / 15.0 sliced on two instances, nIter 5 2.24 / 1.73 against 2.22 / 2.04. */
    return solve_rounds(st, f_work, g_work, s, [&]() {
        return icnn_be::launch_conv_fg(*model, ctx, st->y, st->batch, f_work, g_work, st->skip_fg, s);
    }, ICNN_BE_MAX_ITERS);
}

}  // extern "C"
