/*
 * icnn_be.h -- C ABI of the MI355X bundle-entropy inference library (libicnn_be.so).
 *
 * The reference (locuslab/icnn) has no FFI: its solver is a Python function
 *   solveBatch(fg, initXs, nIter, callback)   lib/bundle_entropy_dual.py:129-179
 *                                             RL/src/bundle_entropy.py:85-136
 * called from multi-label-cls/icnn_ebundle.py:225, completion/icnn_ebundle.py:226 and
 * RL/src/icnn.py:155, and the energy it minimises is a TensorFlow graph evaluated with
 *   sess.run([E_, dE_dy_])                    multi-label-cls/icnn_ebundle.py:218-221.
 * The entry points below are what a binding for that path would call; every
 * comment cites the reference lines the entry point replaces.  INTEGRATION.md
 * shows the ctypes stub that turns them back into `bundle_entropy.solveBatch`.
 *
 * Conventions
 *   - every data pointer is DEVICE memory unless the name ends in _host;
 *     row-major; caller-allocated; the library never frees or keeps them.
 *   - `stream` is a hipStream_t passed as void* (0 = default stream).  All
 *     calls only enqueue work; NONE synchronises or copies to the host, so every
 *     call can be captured into a HIP graph (round 3: the time-sliced solves used
 *     to read a device counter after nIter + 4 rounds; they now enqueue a fixed
 *     number of finishing rounds whose kernels leave at once when nothing is left).
 *   - return value: 0 on success, a negative ICNN_BE_E* code for argument /
 *     launch errors.  Per-sample numerical conditions are reported in
 *     icnn_be_state.status[] (device memory), not in the return value.
 */
#ifndef ICNN_BE_H
#define ICNN_BE_H

#include <stddef.h>

#ifdef __cplusplus
extern "C" {
#endif

#if defined(__GNUC__)
#define ICNN_BE_API __attribute__((visibility("default")))
#else
#define ICNN_BE_API
#endif

#define ICNN_BE_ABI_VERSION 10
#define ICNN_BE_MAX_LAYERS 8   /* z-layers of a PICNN including the final scalar one */
#define ICNN_BE_MAX_SLOTS 31   /* bundle slots (= outer iterations) per solve */
#define ICNN_BE_MAX_ITERS 64   /* outer iterations per solve (icnn_be_state.iters; beyond MAX_SLOTS the slots are recycled) */
#define ICNN_BE_MAX_ROUNDS 128 /* launch rounds of one fused solve (scheduling, see icnn_be_solve_fc) */

/* solver variants */
#define ICNN_BE_VARIANT_DUAL 0 /* lib/bundle_entropy_dual.py */
#define ICNN_BE_VARIANT_RL 1   /* RL/src/bundle_entropy.py   */
#define ICNN_BE_VARIANT_PDIPM 2 /* lib/bundle_entropy.py, solver='pc': the per-sample subproblem
                                  min t - H(y) s.t. G y + h <= t by Mehrotra's predictor-corrector
                                  interior-point method (pdipm_pc :5-78); rank test as in DUAL,
                                  multipliers <= 1e-8 pruned (:234-237).  The module the
                                  icnn_ebundle.py scripts import. */

/* dtype of the cuts (f, g) handed to the solver: whatever `fg` returns */
#define ICNN_BE_CUT_F32 0
#define ICNN_BE_CUT_F64 1

/* per-sample status bits */
#define ICNN_BE_ST_OK 0
#define ICNN_BE_ST_SINGULAR 1  /* Newton system exactly singular: the reference raises
                                  numpy.linalg.LinAlgError in variant DUAL (:56-63) and
                                  keeps the current multipliers in variant RL (:55-62);
                                  variant PDIPM: the KKT matrix is not positive definite
                                  (numpy.linalg.cholesky raises, lib/bundle_entropy.py:42) */
#define ICNN_BE_ST_NONFINITE 2 /* a non-finite value reached the bundle */
#define ICNN_BE_ST_OVERFLOW 4  /* the sample's active bundle outgrew what one workgroup can stage in LDS
                                  (icnn_be_bundle_capacity; only wide rows reach it: n = 2048 holds 13 cuts).
                                  The sample stops at its current iterate; the reference has no such limit */

#define ICNN_BE_ST_UNFINISHED 8 /* a fused solve with time-sliced rounds gives the samples that fall behind (parked Newton
                                  solves) nIter finishing rounds instead of asking the device how many are needed -- always
                                  enough: a sample has nIter iterations and a finishing round completes one --; a sample
                                  still behind after them would say so here (safety net of icnn_be_solve_conv) */

/* return codes */
#define ICNN_BE_EINVAL (-1)    /* bad argument */
#define ICNN_BE_ELIMIT (-2)    /* size beyond a compiled-in limit */
#define ICNN_BE_ELAUNCH (-3)   /* HIP launch failed (see icnn_be_last_hip_error) */

/* flags */
#define ICNN_BE_FLAG_NO_CYCLE_SHORTCUT 1 /* always run the full Newton cap (dual :30, rl :29).  Without this flag a Newton
                                          * iteration that has entered a limit cycle of period 1..4 is cut short: exact repeats
                                          * (to 1e-13) return the very iterate the cap would end on; cycles at their rounding
                                          * floor (variant dual, NOISE_TOL in be_dual_dev.h) and extrapolated slow 2-cycles
                                          * return a point of the same cycle that differs from the reference's lam_100 by the
                                          * cycle's own jitter (<= ~2.5e-11 on 10 243 recorded solves).  "The reference's
                                          * sequence of operations, bit for bit" below therefore holds with this flag set;
                                          * the default agrees with it to ~1e-11 in lam (tests/test_gpu_parity.py keeps one
                                          * parity test on each side) */
#define ICNN_BE_FLAG_TIME_SLICE 2        /* fused solve: always park Newton solves that exceed a per-round
                                            budget and resume them in later rounds (icnn_be_solve_fc) */
#define ICNN_BE_FLAG_LOCKSTEP 4          /* fused solve: never do that; exactly nIter rounds, no sync.
                                            Neither flag: time slicing when nIter > 15 (measured) */
#define ICNN_BE_FLAG_TWO_KERNELS 8       /* icnn_be_solve_fc: one launch per phase and round, never a persistent kernel */
#define ICNN_BE_FLAG_PERSISTENT 16       /* icnn_be_solve_fc: the persistent per-tile kernel (a workgroup per 16 samples)
                                          * whenever the shape fits it (float32 cuts, nIter <= 15, narrow rows, lockstep),
                                          * whatever the batch size (with 4- or 8-sample partial tiles below one full tile per CU).
                                          * Default (no flag): a persistent workgroup per 1-4
                                          * samples for batches of at most four samples per CU (MI355X: <= 1024; any nIter,
                                          * no time slicing needed), the per-tile kernel for batches that
                                          * give every CU between a quarter of a tile and two tiles (1024..8192, variants dual and -- round 4 --
                                          * pdipm, which always runs in lockstep rounds: its solve has a fixed iteration cap),
                                          * two kernels otherwise.  Results are bit-identical whichever path runs. */

#define ICNN_BE_FLAG_WAVE_PER_SAMPLE 128  /* narrow rows (n <= 16, variant RL) run four samples per wave by default (one per
                                          * 16-lane DPP row, be_dual_small.hip); this flag keeps the wave-per-sample kernel.
                                          * Same operations in the same order: bit-identical results */
#define ICNN_BE_FLAG_MFMA_CONTRACTION 256  /* variant dual: keep the float64-MFMA sweep for H = A diag(w) A^T, A z instead of the
                                          * fused VALU pass (be_dual_valu_dev.h) that wide rows (several waves per sample)
                                          * take for bundles of up to 20 cuts and -- round 4 -- one-wave samples with float32
                                          * rows of up to 192 columns for bundles of up to 8 cuts.  Same sums in another
                                          * order: results agree to rounding, not bit for bit.  Whichever it is, every
                                          * dispatch path takes the same decision (bit-identical among themselves) */
#define ICNN_BE_FLAG_GLOBAL_BUNDLE 64      /* stage the bundle of EVERY round in st->scratch instead of LDS (diagnostic: the
                                          * rounds whose bundle does not fit LDS do so anyway); same arithmetic, same bits.
                                          * One exception: variant PDIPM on float32 rows of up to 192 columns with 2..12 cuts
                                          * forms its column sums (M = G Hinv G^T, G Hinv ry, G y) by the unrolled VALU
                                          * passes from LDS and by the f64-MFMA sweep / wave reductions from st->scratch
                                          * (another summation order: results agree to ~1e-13, not bit for bit;
                                          * tests/test_gpu_parity.py).  Wider float32 rows take the same column-chunked
                                          * passes from either place (same order, same bits). */
#define ICNN_BE_FLAG_F64_ENERGY 32        /* icnn_be_dual_step: f is float64 [B] whatever the cut dtype (an `fg` that
                                          * returns float64 energies with float32 gradients: the reference's
                                          * bi = fi - sum(gi * x) keeps fi's precision, dual :143) */

/*
 * Bundle state of one solveBatch call, slot-addressed: the cut taken at outer
 * iteration t lives in slot t; `active[u]` lists, in order, the slots still in
 * sample u's bundle and `lam[u]` their multipliers.  This replaces the ragged
 * Python lists A, b, xs, lam of the reference (dual :131-134) one for one:
 *   A[u][i]  = G[u][active[u][i]]      b[u][i] = h[u][active[u][i]]
 *   xs[u][i] = ys[u][active[u][i]]     lam[u]  = lam[u][0:count[u]]
 */
typedef struct icnn_be_state {
    int batch;          /* B */
    int n;              /* dim(y) */
    int slots;          /* T = nIter of this call, 1..ICNN_BE_MAX_SLOTS */
    int cut_dtype;      /* ICNN_BE_CUT_* : dtype of G (and of f, g passed per step) */
    int variant;        /* ICNN_BE_VARIANT_* */
    int flags;          /* ICNN_BE_FLAG_* */
    double *y;          /* [B][n]    in: start point (initXs); out: minimiser, updated in place */
    void *G;            /* [B][T][n] cut gradients, cut dtype */
    double *h;          /* [B][T]    cut offsets  f - <g, y> */
    double *ys;         /* [B][T][n] points the cuts were taken at */
    double *lam;        /* [B][T]    multipliers of the active slots (after pruning: all > 0) */
    int *active;        /* [B][T]    ordered active slots */
    int *count;         /* [B]       len(active[u]) */
    int *n_iters;       /* [B]       reference `nIters` */
    int *finished;      /* [B]       1 once the sample left the loop (rank test / stall / error) */
    int *status;        /* [B]       ICNN_BE_ST_* bits */
    int *newton_iters;  /* [B]       total Newton updates spent on the sample (diagnostic) */
    /* scheduling state, internal to the library (caller only allocates it) */
    int *t_next;        /* [B]       next outer iteration of the sample */
    int *phase;         /* [B]       0 = needs a cut at y, 1 = Newton solve parked mid-way */
    int *skip_fg;       /* [B]       1 = the sample needs no energy/gradient in the next round */
    int *pending;       /* [ICNN_BE_MAX_ROUNDS] per round: non-zero if any sample still has work afterwards */
    double *park;       /* [B][5*T+4] parked Newton state (lam, four previous iterates, counters) */
    void *scratch;      /* icnn_be_scratch_bytes() bytes or NULL: staging area in device memory for the rounds whose
                           bundle exceeds the LDS capacity (wide rows: n = 2048 stages 12 cuts in LDS); NULL: such a
                           sample stops with ICNN_BE_ST_OVERFLOW */
    double *fvals;      /* [B][T] or NULL: energy f of the cut in each slot as fg returned it (the host replays the reference's
                           callback(t, f, y) of a fused solve from fvals and ys, lib/bundle_entropy_dual.py:144-145) */
    int iters;          /* outer iterations of this call, slots..ICNN_BE_MAX_ITERS; 0 = slots.  More iterations than slots
                           (the reference has no cap on nIter, dual :129): a new cut takes the lowest slot that is not in
                           the sample's active list -- pruned cuts give their slots back --, so only the ACTIVE bundle is
                           limited to `slots` cuts (a sample that would exceed it stops with ICNN_BE_ST_OVERFLOW); slot t
                           is then no longer iteration t, and fvals / ys hold no per-iteration history */
} icnn_be_state;

/*
 * Shape + packed weights of the y-dependent part of a fully-connected PICNN,
 *   z_i = act( (z_{i-1} * gate_i) Wzu_i  +  (y * yu_i) Wyu_i  +  zu_i ),  i = 0..L,
 * multi-label-cls/icnn_ebundle.py:349-388, RL/src/icnn.py:356-404.  width[L] = 1.
 * gate_i, yu_i, zu_i are the x-only "context", one row of `ctx_width` floats per
 * sample laid out per layer as  yu_i[n] | zu_i[width[i]] | gate_i[width[i-1]] (i>0).
 */
typedef struct icnn_be_fc_model {
    int n;                              /* dim(y) */
    int n_layers;                       /* L+1 */
    int width[ICNN_BE_MAX_LAYERS];      /* s_0 .. s_L, s_L == 1 */
    float alpha;                        /* leaky-ReLU slope of the hidden z-layers; 0 = ReLU */
    int action_box;                     /* 1: RL wrapper, network sees 2y-1 and dE/dy is doubled
                                           (RL/src/icnn.py:148-158) */
    int ctx_width;                      /* floats per context row (checked against the shape) */
    const float *wpack;                 /* packed y-path weights, icnn_be_fc_pack_floats() floats */
} icnn_be_fc_model;

/*
 * Shape + packed weights of the y-dependent part of the convolutional PICNN of the image
 * completion experiment (completion/icnn_ebundle.py:376-452): three conv z-layers
 * (reference: 32 k8 s4, 64 k4 s2, 64 k3 s1 on a 64x32x1 image) with the learned down-sampling
 * chain y_red, then fc `fc_hidden` and fc 1.  NHWC, 'SAME' padding.  Context row per sample:
 *   yu_0[H*W] | zu_0 | gate_1 | yu_1 | zu_1 | gate_2 | yu_2 | zu_2 | gate_3[flat] | zu_3 | gate_4 | zu_4[1]
 * (gate_l has the shape of z_{l-1}, yu_l of y_red_l, zu_l of z_l).
 */
typedef struct icnn_be_conv_model {
    int H, W;                /* y (and x) are H x W x 1; n = H*W */
    int filters[3], ksize[3], stride[3];
    int fc_hidden;
    int ctx_width;           /* floats per context row (checked against the shape) */
    const float *wpack;      /* packed y-path weights, icnn_be_conv_pack_floats() floats */
    float *work;             /* device scratch of icnn_be_conv_work_floats(model, work_batch) floats: the activations
                                that cross the four launches of one evaluation (ReLU masks, flatten(z_2), z_3, delta_2) */
    int work_batch;          /* batch the scratch was sized for (>= every batch passed with this model).
                                SINGLE-STREAM: `work` is written by every icnn_be_conv_fg / icnn_be_solve_conv call that names
                                this model although the struct is passed as const -- two calls that share one model struct
                                must be ordered on ONE stream (or use two structs with their own `work` and the same
                                `wpack`); the library does not synchronise them */
} icnn_be_conv_model;

ICNN_BE_API int icnn_be_abi_version(void);
ICNN_BE_API const char *icnn_be_last_hip_error(void);

/* sizeof(icnn_be_state) for which = 0, sizeof(icnn_be_fc_model) for 1, sizeof(icnn_be_fc_ctx) for 2,
 * sizeof(icnn_be_conv_model) for 3, sizeof(icnn_be_conv_ctx) for 4: lets a
 * foreign-language binding verify its struct layout at load time. */
ICNN_BE_API size_t icnn_be_struct_size(int which);

/* bytes of dynamic LDS one workgroup of the dual-step kernel needs (diagnostic) */
ICNN_BE_API int icnn_be_dual_lds_bytes(int n, int slots, int cut_dtype);

/* Most cuts (the new one included) a sample's ACTIVE bundle may hold at once: min(slots, what fits 160 KB of LDS).
 * The number of outer iterations (slots) is not limited by it -- the reference keeps only the cuts with a positive
 * multiplier from one iteration to the next (dual :171-174) and so does the state; a sample whose active bundle would
 * exceed the capacity gets ICNN_BE_ST_OVERFLOW unless st->scratch is provided (below).  Negative: error code. */
ICNN_BE_API int icnn_be_bundle_capacity(int n, int slots, int cut_dtype, int variant);

/* Bytes of st->scratch that lift the capacity to `slots` cuts (batch, n, slots, cut_dtype, variant of *shape are read):
 * from the round on in which the bundle could outgrow the LDS, the dual step stages it in device memory instead -- same
 * kernel, same arithmetic, the sweeps then run at L2 latency.  0: not needed (everything fits LDS) or not available
 * (the RL variant, whose action vectors are narrow).  */
ICNN_BE_API size_t icnn_be_scratch_bytes(const icnn_be_state *shape);

/* Reset count/finished/status/n_iters/newton_iters for a new solve (dual :130-139). */
ICNN_BE_API int icnn_be_state_init(const icnn_be_state *st, void *stream);

/*
 * One outer bundle iteration t for the whole batch, given the batch's energies
 * f[B] and gradients g[B][n] (cut dtype) at the current st->y: append the cut,
 * rank test, projected-Newton dual solve, y <- sigmoid(-G^T lam), prune.
 * Replaces the body of the reference's `for u in range(bsize)` loop,
 * lib/bundle_entropy_dual.py:143-174 and RL/src/bundle_entropy.py:102-131
 * (proj_newton_logistic :15-85 / :14-83 and logexp1p :6-12 included).
 */
ICNN_BE_API int icnn_be_dual_step(const icnn_be_state *st, int t, const void *f, const void *g, void *stream);

/* Number of floats of the packed weight buffer for a model shape (wpack may be NULL). */
ICNN_BE_API size_t icnn_be_fc_pack_floats(const icnn_be_fc_model *shape);

/*
 * Pack the y-path weights for the kernels (host -> host; upload the result and
 * store the device pointer in model->wpack).  w_yu_host[i] is 'z{i}_yu/W'
 * [n][width[i]], w_zu_host[i] (i >= 1) is 'z{i}_zu_proj/W' [width[i-1]][width[i]],
 * both row-major float32 as tflearn stores them (icnn_ebundle.py:357-368).
 */
ICNN_BE_API int icnn_be_fc_pack(const icnn_be_fc_model *shape, const float *const *w_yu_host,
                    const float *const *w_zu_host, float *out_host);

/*
 * E[B] and dE/dy[B][n] (float32) of the PICNN at y (float64, rounded to float32
 * on entry like a TensorFlow feed).  Replaces sess.run([E_, dE_dy_]) --
 * multi-label-cls/icnn_ebundle.py:218-221, RL/src/icnn.py:127-131 -- for the
 * y-dependent part; ctx[B][ctx_width] is the x-only part.  Rows whose
 * finished[u] != 0 are skipped (finished may be NULL).
 */
ICNN_BE_API int icnn_be_fc_fg(const icnn_be_fc_model *model, const float *ctx, const double *y, int batch,
                  float *f, float *g, const int *finished, void *stream);

/*
 * The whole solveBatch loop on the device for a PICNN energy: nIter rounds of
 * { icnn_be_fc_fg ; dual step }, enqueued without any host synchronisation.
 *
 * With time slicing (default for nIter > 15, see the flags) the samples do not advance in lockstep: a sample whose Newton
 * solve exceeds a per-round budget (the un-line-searched iteration of the reference falls into
 * limit cycles on ~0.1 % of the solves and then runs its full 100-iteration cap) is parked and
 * resumed in the next round while all other samples move on; every sample still performs exactly
 * the same sequence of operations as in lockstep rounds (bit-identical results between the paths; against the
 * reference see ICNN_BE_FLAG_NO_CYCLE_SHORTCUT).  The samples that are behind after
 * the nIter budgeted rounds are finished without asking the host: by ONE launch of the persistent per-sample
 * kernel (FC models), or by nIter unbudgeted rounds whose kernels leave at once where nothing is left (conv
 * model, ICNN_BE_FLAG_TWO_KERNELS).  For 1024..8192 samples the budgeted rounds themselves are ONE launch of
 * the persistent per-tile kernel (dual phase in LDS groups sized by the cuts the samples hold).  Measured on
 * MI355X at batch 4096, nIter 30: 6.3-6.7 ms (24.7 ms in lockstep launch pairs, 10.3 ms as time-sliced launch
 * pairs); nIter 10: lockstep (the extra rounds cost more than the slicing saves).  Replaces
 * bundle_entropy.solveBatch(fg, y0, nIter) at multi-label-cls/icnn_ebundle.py:225-226
 * with fg = the TensorFlow closure of :218-221.  f_work[B], g_work[B][n] are scratch.
 * The state must have been reset with icnn_be_state_init; st->cut_dtype must be F32.
 * Returns the number of rounds issued (> 0) or a negative error code.
 */
ICNN_BE_API int icnn_be_solve_fc(const icnn_be_fc_model *model, const float *ctx, const icnn_be_state *st,
                     float *f_work, float *g_work, void *stream);

/* ---- x-only context producer and weight clamps ------------------------ */

/*
 * Weights of everything that depends on x alone, one column-wise concatenation per stage i = 0 .. n_layers-1
 * (stage i reads prev_i: x for i = 0, the u-path activation u_{i-1} otherwise):
 *   w_stage[i] = [ 'u{i}/W' (i < n_layers-1) | 'z{i}_yu_u/W' | 'z{i}_u/W' | 'z{i}_zu_u/W' (i > 0) ]   row-major
 *                [K_i][ld_i], K_0 = n_features, K_i = width[i-1], ld_i = the column count rounded up to a multiple of 4
 *                (pad columns zero);  b_stage[i] the biases in the same column order.
 * multi-label-cls/icnn_ebundle.py:339-347 (u-path), :354-374 (heads); RL/src/icnn.py:339-385.  Hidden u layers
 * are ReLU'd, and batch-normalised with the statistics of the batch when `batchnorm` (bn_gamma/bn_beta[i], i <
 * n_layers-2, epsilon bn_eps = 1e-5); the last u layer is linear.  All pointers device memory.
 */
typedef struct icnn_be_fc_ctx {
    int n_features, n, n_layers;
    int width[ICNN_BE_MAX_LAYERS];      /* as icnn_be_fc_model.width */
    int batchnorm;
    float bn_eps;
    const float *w_stage[ICNN_BE_MAX_LAYERS];
    const float *b_stage[ICNN_BE_MAX_LAYERS];
    const float *bn_gamma[ICNN_BE_MAX_LAYERS];
    const float *bn_beta[ICNN_BE_MAX_LAYERS];
} icnn_be_fc_ctx;

/* floats of device scratch icnn_be_fc_context needs for a batch (the u-path activations) */
ICNN_BE_API size_t icnn_be_fc_context_work_floats(const icnn_be_fc_ctx *c, int batch);

/*
 * ctx[batch][ctx_width] from x[batch][n_features] (float32): the part of the reference's graph that does not depend
 * on y -- it sits inside `fg`'s sess.run on every bundle iteration there (multi-label-cls/icnn_ebundle.py:218-221)
 * and is computed once per minibatch here.  Row layout as icnn_be_fc_model expects (yu_i | zu_i | gate_i per layer).
 * BatchNorm uses the statistics of exactly these `batch` rows: shard AFTER this call.
 */
ICNN_BE_API int icnn_be_fc_context(const icnn_be_fc_ctx *c, const float *x, int batch, float *ctx, int ctx_width,
                                   float *work, void *stream);

/*
 * The same for DATA-PARALLEL ranks that each hold a shard of the minibatch: the u-path BatchNorm uses the
 * statistics of the GLOBAL batch (tflearn.batch_normalization in training mode, multi-label-cls/icnn_ebundle.py:345), so
 * the producer is issued stage by stage and the ranks all-reduce 2 x width[stage] doubles behind every normalised stage:
 *     for stage in 0 .. n_layers-1:
 *         rc = icnn_be_fc_context_stage(c, stage, x, batch, ctx, ctx_width, work, stats, stream)   // GEMM of the stage
 *         if rc == 1:      // u_stage is batch-normalised: stats[0..w) = sum u, stats[w..2w) = sum u^2 of THIS rank's rows
 *             all_reduce(stats, SUM)                                       // RCCL; 2 x 600 doubles for the Bibsonomy model
 *             icnn_be_fc_context_norm(c, stage, batch, batch_total, stats, work, stream)
 * `x` is this rank's [batch][n_features] rows (read by stage 0 only), `work` as for icnn_be_fc_context (the stages find their
 * inputs in it), `stats` a device buffer of 2 * max(width) doubles.  With one rank (batch_total = batch) the result equals
 * icnn_be_fc_context's up to the float32 rounding of the variance (E[u^2] - mean^2 in float64 here, two passes in float32 there).
 * icnn_be_fc_context_stage returns 1 when statistics were written, 0 when the stage has no BatchNorm, < 0 on error.
 */
ICNN_BE_API int icnn_be_fc_context_stage(const icnn_be_fc_ctx *c, int stage, const float *x, int batch, float *ctx, int ctx_width,
                                         float *work, double *stats, void *stream);
ICNN_BE_API int icnn_be_fc_context_norm(const icnn_be_fc_ctx *c, int stage, int batch, double batch_total, const double *stats,
                                        float *work, void *stream);

/* makeCvx (ICNN_BE_CLAMP_ABS, icnn_ebundle.py:143,:204) / proj (ICNN_BE_CLAMP_RELU, :144,:244-245) on the
 * 'z{i}_zu_proj/W' operands inside model->wpack (both packed orientations), in place on the device. */
#define ICNN_BE_CLAMP_ABS 0
#define ICNN_BE_CLAMP_RELU 1
#define ICNN_BE_CLAMP_ABS_HALF 2    /* |W| / 2: the completion model's makeCvx, completion/icnn_ebundle.py:145 */
ICNN_BE_API int icnn_be_fc_clamp(const icnn_be_fc_model *model, int mode, void *stream);

/*
 * The same for the convolutional PICNN of the completion experiment (completion/icnn_ebundle.py:346-367 u-path with
 * BatchNorm, :376-452 the x-only halves of every layer).  Operands that read the same input through the same window
 * are concatenated column-wise on the host (icnn_amd/picnn.py: ConvModel.repack_context): stage s is a device matrix
 * [K][ld] row-major float32, K = k*k*Cin in tflearn's [k][k][Cin][F] order read as [K][F], ld = columns rounded up to a
 * multiple of 4 (pad columns zero), b_stage[s] the biases in column order:
 *   0  x,  8x8 / 4 :  u0 (F0)            | zu0 = z0_u (F0)
 *   1  x,  3x3 / 1 :  yu0 = z0_yu_u (1)
 *   2  u0, 4x4 / 2 :  u1 (F1)            | zu1 = z1_u (F1)
 *   3  u0, 3x3 / 1 :  gate1 = z1_zu_u (F0) | yu1 = z1_yu_u (1)
 *   4  u1, 3x3 / 1 :  u2 (F2) | gate2 = z2_zu_u (F1) | yu2 = z2_yu_u (1) | zu2 = z2_u (F2)
 *   5  flat u2     :  u3 (fc_hidden)     | gate3 = z3_zu_u (flat) | zu3 = z3_u (fc_hidden)
 *   6  u3          :  gate4 = z4_zu_u (fc_hidden) | zu4 = z4_u (1)
 * u0..u3 are ReLU'd and batch-normalised with the statistics of the batch (bn_gamma/bn_beta[0..3], bn_eps = 1e-5),
 * gates are ReLU'd.  x is [batch][H][W][1] float32 (already h-flipped by the caller, :215); the context row layout is
 * the one icnn_be_conv_fg reads.  Seven GEMM launches (f32 MFMA, implicit im2col) + four BatchNorm launches.
 */
typedef struct icnn_be_conv_ctx {
    const float *w_stage[7];
    const float *b_stage[7];
    const float *bn_gamma[4];
    const float *bn_beta[4];
    float bn_eps;
} icnn_be_conv_ctx;
ICNN_BE_API size_t icnn_be_conv_context_work_floats(const icnn_be_conv_model *shape, int batch);
ICNN_BE_API int icnn_be_conv_context(const icnn_be_conv_model *shape, const icnn_be_conv_ctx *c, const float *x, int batch,
                                     float *ctx, float *work, void *stream);
/* makeCvx (ICNN_BE_CLAMP_ABS_HALF, completion/icnn_ebundle.py:145,:190) / proj (ICNN_BE_CLAMP_RELU, :146,:248-249)
 * on the 'z{1..4}_zu_proj/W' operands inside model->wpack (every packed orientation), in place on the device. */
ICNN_BE_API int icnn_be_conv_clamp(const icnn_be_conv_model *model, int mode, void *stream);

/* ---- implicit-differentiation feed of a training step ----------------- */
#define ICNN_BE_LOSS_XENT 0   /* crossEntrGrad, multi-label-cls/icnn_ebundle.py:390-417 */
#define ICNN_BE_LOSS_MSE 1    /* mseGrad,       completion/icnn_ebundle.py:493-522     */

/*
 * From a finished solve: one output row per active cut of every sample with a non-empty bundle,
 *   fd_y[r] = ys_{j,i},  fd_v[r] = lam_i c_y + c_lam,i (y*_j - ys_{j,i}),  fd_c[r] = c_lam,i,  fd_sample[r] = j,
 * rows of sample j starting at row_offset[j] (exclusive prefix sum of st->count, device).  y_true is
 * [B][n] float64.  Replaces train_step_fd (multi-label-cls/icnn_ebundle.py:296-314,
 * completion/icnn_ebundle.py:315-335) including the per-sample (k+1)x(k+1) KKT solve.
 */
ICNN_BE_API int icnn_be_implicit_feed(const icnn_be_state *st, const double *y_true, int loss,
                                      const int *row_offset, double *fd_y, double *fd_v, double *fd_c,
                                      int *fd_sample, void *stream);

/* ---- the reference's return value (SURVEY.md 8(b) "Return / ownership") ------------------------ */

/*
 * From a finished solve: the ACTIVE cuts of every sample packed row by row, sample by sample in bundle order --
 *   G_rows[r] = G[u][active[u][i]] (cut dtype, n values),  ys_rows[r] = ys[u][active[u][i]],
 *   h_rows[r] = h[u][active[u][i]],  lam_rows[r] = lam[u][i]        r = row_offset[u] + i,  i < count[u]
 * (row_offset = exclusive prefix sum of st->count, device; the buffers hold sum(count) rows).  These are the ragged
 * Python lists A, xs, b, lam that solveBatch returns (lib/bundle_entropy_dual.py:131-134, :171-179), ready for ONE
 * device-to-host copy: what an unmodified multi-label-cls/icnn_ebundle.py:225-226 / :296-314 consumes.
 */
ICNN_BE_API int icnn_be_export_active(const icnn_be_state *st, const int *row_offset, void *G_rows, double *ys_rows,
                                      double *h_rows, double *lam_rows, void *stream);

/* ---- Adam inner optimiser of the RL agent ---------------------------- */

/* bytes of device scratch icnn_be_adam_fc needs for a batch (iterate, moments, per-iteration f and g, barrier) */
ICNN_BE_API size_t icnn_be_adam_workspace_bytes(int batch, int n);

/*
 * `Agent.adam(func, obs)` of RL/src/icnn.py:160-215 with func = `_fg_entr` (:59-63, :131; what act() and train()
 * pass, :270-272, :306-317): projected Adam on  negQ(obs, act) - H((act+1)/2)  over act in [-1+1e-8, 1-1e-8]^n from
 * act = 0 (b1 0.9, b2 0.999, alpha 0.01, eps 1e-8, step m_hat / (sqrt(v) + eps) with the UNcorrected v exactly as
 * :201 has it), best iterate per state, stop when the smoothed mean displacement of the best iterates is below
 * 1e-3 after more than 5 iterations (:184-192), else after max_iter (reference: 1000) evaluations.
 * model: the negQ PICNN with action_box = 0 (the action is fed as is); ctx[B][ctx_width] its x-only context.
 * Out (device): act_best[B][n] float64, f_best[B] float32 (negQ_entr at act_best), *iters = evaluations before the
 * rule fired (what the reference prints), max_iter if it never did.  One kernel launch, no host synchronisation.
 * The stopping rule couples the whole batch, so all ceil(B/16) workgroups must be resident at once: returns
 * ICNN_BE_ELIMIT beyond that (MI355X: one 1024-thread workgroup per CU, 256 CUs = 4096 states).  Up to 1024 states
 * (dim(action) <= 64) run with 1-4 states per CU on the VALU latency path, larger batches as 16-state MFMA tiles.
 */
ICNN_BE_API int icnn_be_adam_fc(const icnn_be_fc_model *model, const float *ctx, int batch, int max_iter,
                                double *act_best, float *f_best, int *iters, void *workspace, void *stream);

/*
 * The same from the observations themselves: obs[B][n_features] -> the x-only context rows (cx: the stage weights of
 * icnn_be_fc_context) -> the Adam loop, ONE launch: what `act()` does per environment step (RL/src/icnn.py:264-288).
 * Latency path only -- at most four states per workgroup (MI355X: B <= 1024), dim(action) <= 64, a model without
 * BatchNorm (the agent's default icnn_bn=False, RL/src/agent.py:23) -- otherwise ICNN_BE_ELIMIT: call
 * icnn_be_fc_context, then icnn_be_adam_fc.
 */
ICNN_BE_API int icnn_be_adam_fc_obs(const icnn_be_fc_model *model, const icnn_be_fc_ctx *cx, const float *obs, int batch,
                                    int max_iter, double *act_best, float *f_best, int *iters, void *workspace,
                                    void *stream);

/* ---- convolutional PICNN (completion/icnn_ebundle.py) ---------------------------------------- */

/* Number of floats of the packed weight buffer (0 if the shape is rejected). */
ICNN_BE_API size_t icnn_be_conv_pack_floats(const icnn_be_conv_model *shape);

/* floats of device scratch an evaluation of `batch` samples needs (model->work). */
ICNN_BE_API size_t icnn_be_conv_work_floats(const icnn_be_conv_model *shape, int batch);

/*
 * Pack the y-path weights (host -> host), all row-major float32 as tflearn stores them:
 *   w_yu_host[l]  'z{l}_yu/W'      [k][k][1][F_l]            l = 0..2   (:390-392)
 *   w_yr_host[l]  'z{l}_y_red/W'   [k][k][1][1], b_yr_host[l] its bias [1],  l = 0, 1   (:394-396)
 *   w_zu_host[l]  'z{l}_zu_proj/W' [k][k][F_{l-1}][F_l]      l = 1, 2   (:385-387; [0] ignored)
 *   w_fc3_host    'z3_zu_proj/W'   [flat][fc_hidden],  w_fc4_host 'z4_zu_proj/W' [fc_hidden][1]  (:418-420)
 */
ICNN_BE_API int icnn_be_conv_pack(const icnn_be_conv_model *shape, const float *const *w_yu_host,
                                  const float *const *w_yr_host, const float *const *b_yr_host,
                                  const float *const *w_zu_host, const float *w_fc3_host,
                                  const float *w_fc4_host, float *out_host);

/*
 * E[B] and dE/dy[B][H*W] (float32) at y (float64, flat [B][H*W], rounded to float32 on entry).
 * Replaces sess.run([E_, dE_dyFlat_]) of completion/icnn_ebundle.py:217-221 for the y-dependent
 * part.  Rows whose finished[u] != 0 are skipped (finished may be NULL).
 */
ICNN_BE_API int icnn_be_conv_fg(const icnn_be_conv_model *model, const float *ctx, const double *y, int batch,
                                float *f, float *g, const int *finished, void *stream);

/* The whole solveBatch loop for the conv PICNN (completion/icnn_ebundle.py:226-227); see icnn_be_solve_fc. */
ICNN_BE_API int icnn_be_solve_conv(const icnn_be_conv_model *model, const float *ctx, const icnn_be_state *st,
                                   float *f_work, float *g_work, void *stream);

/*
 * Diagnostic hooks (no reference counterpart; used by tools/dual_phase_profile.py, tools/fc_phase_profile.py,
 * tools/conv_dual_phase_profile.py).  While a buffer is set, the kernels add per-phase cycle counts (s_memtime laps of lane 0)
 * to it and the dispatcher picks the instrumented kernels; NULL switches the hooks off again.  Process-wide, not thread-safe,
 * not for production use.  The laps are compiled into the PROFILING variant of the library only (the same sources with
 * -DICNN_BE_PROF=1: `python -m icnn_amd.build --prof` -> icnn_amd/csrc/prof/libicnn_be.so, what the tools load): in the production
 * library they would cost the benchmark solve 1.8 %, and there these calls set a pointer that no kernel reads.
 *   icnn_be_debug_profile       device_buf [max(B, 4096) + 8][16] int64: dual-step phases per sample
 *   icnn_be_debug_profile_fc    device_buf int64, FC-PICNN phases per workgroup and wave: [ceil(B / 16)][16][16] for the tile
 *                               kernels (icnn_be_fc_fg, the persistent tile solve); the per-sample kernels (up to four samples
 *                               per CU: fused_rows_solve_kernel) index it [ceil(B / per_wg)][8][16] with per_wg =
 *                               ceil(B / CUs) -- up to B workgroups: size the buffer max(ceil(B / 16) * 16, B * 8) * 16
 *                               entries to cover both (tools/rows_phase_profile.py)
 *   icnn_be_debug_profile_conv  device_buf: conv-PICNN phases per workgroup and wave
 */
ICNN_BE_API void icnn_be_debug_profile(long long *device_buf);
/* Per-round timeline of the persistent tile kernel's grouped dual phase (nIter > 15; profiling variant only,
 * tools/c4_timeline.py): device_buf [B][ICNN_BE_MAX_ITERS][4] int64, per sample and round the shader-clock stamps of
 * { the tile's dual phase starting, this sample's dual step starting (behind its wait for a staging region), its end } and
 * the sample's Newton updates before the round.  NULL switches it off. */
ICNN_BE_API void icnn_be_debug_trace(long long *device_buf);
/* counters per sample of icnn_be_debug_profile's buffer (16): size the buffer from this, not from a literal */
ICNN_BE_API int icnn_be_debug_profile_phases(void);
/*
 * The exp / log / softplus / sigmoid the INNER iterations use in place of the math library's (be_dual_dev.h: fast_exp,
 * fast_log, softplus_fast, sigmoid_fast -- lean argument reductions + Horner polynomials; values that are final, y =
 * 1 / (1 + exp(A^T lam)) of lib/bundle_entropy_dual.py:165, keep the library routines), evaluated on caller-supplied
 * arguments: out[i] = f(x[i]), which = 0 exp, 1 log, 2 softplus (logexp1p, dual :6-12), 3 sigmoid.  Lets a test bound
 * their error against extended precision (tests/test_gpu_parity.py: relative error <= 5e-16 on the ranges the iterations
 * feed them, exact limits at the clamp +-750 and at +-inf).  NaN: the argument clamp of fast_exp (v_max / v_min) maps NaN
 * to a finite value, so fast_exp(NaN) = exp(-750) = 0 and sigmoid_fast(NaN) is finite: a non-finite A^T lam inside the Newton
 * loop is NOT propagated by these routines; it surfaces at the y update, whose library exp yields NaN and sets
 * ICNN_BE_ST_NONFINITE (non-finite energies / gradients are caught before, when the cut is taken).
 */
ICNN_BE_API int icnn_be_debug_fast_math(int which, const double *x, double *out, int count, void *stream);
ICNN_BE_API void icnn_be_debug_profile_fc(long long *device_buf);
ICNN_BE_API void icnn_be_debug_profile_conv(long long *device_buf);

#ifdef __cplusplus
}
#endif
#endif /* ICNN_BE_H */
