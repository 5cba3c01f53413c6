// Adam inner optimiser of the RL agent (RL/src/icnn.py:160-215 `adam(func, obs)` with func = `_fg_entr`, :59-63,131):
// projected Adam on negQ(obs, act) - H((act+1)/2) over act in [-1+1e-8, 1-1e-8]^n from act = 0, best iterate per
// sample, batch-level stopping rule -- up to 1000 evaluations of the PICNN and its action gradient per call.  The
// reference pays one sess.run per evaluation; here the WHOLE loop is one launch: a persistent workgroup per tile of
// 16 states alternates
//     phase A  fc_fg_tile (be_picnn_fc_dev.h: negQ and d negQ / d act of the tile on the f32 MFMA chain)
//     phase B  one wave per state: entropy term, best-iterate bookkeeping, moment update, clipped step (float64)
// and the only cross-workgroup traffic is the stopping rule's mean displacement of the best iterates: one double
// per tile and iteration through a grid barrier (none at all for batch <= 16, the agent's act() shape).
//
// Arithmetic contract (oracle/adam_oracle.py reproduces it bit for bit): entropy per element from the float32
// action with float64 logs rounded to float32, summed sequentially in float32; the moment products (1-b1) g and
// (1-b2) g g in float32 as NumPy forms them, everything else float64 with IEEE division and square root, no
// contraction; the step divides by sqrt(v) (not vhat) as the reference does.
#include "be_dual_dev.h"
#include "be_picnn_fc_dev.h"
#include "be_picnn_fc_rows_dev.h"

namespace icnn_be {

namespace {

struct AdamArgs {
    FcArgs fa;            // fa.y = act, fa.f / fa.g = per-iteration energies / gradients, fa.finished = nullptr
    double *act, *m, *v;  // [B][n] iterate and moments (workspace; initialised by the kernel)
    double *act_best;     // [B][n] out
    float *f_best;        // [B] out
    double *partial;      // [2][tiles] per-tile sums of ||best_t - best_{t-1}||, double-buffered by iteration parity
    unsigned *arrive;     // [tiles] iteration number each workgroup has published, zero at launch
    int *iters;           // out: iterations run (== max_iter when the rule never fired)
    int max_iter, tiles, red_off;   // red_off: float offset of the reduction scratch behind fc_fg_tile's LDS
};

__device__ __forceinline__ float lane_value(float v, int l) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), l));
}

// Sum of one double per workgroup over all workgroups of the (cooperative) launch, the same value (same order) in
// every workgroup; called by all lanes of ONE wave per workgroup.  Flag-based, no read-modify-write: a counter
// that 256 workgroups increment serialises at the device's coherence point (0.15 us per arrival: 40 us per
// iteration), and release/acquire semantics at agent scope would write back and INVALIDATE the whole L2 of every
// XCD each iteration, so that the read-only weights came from memory again.  Here a workgroup publishes its
// partial sum and then its iteration number (both agent-scope atomic stores, the second after the first has been
// acknowledged), and every workgroup polls all iteration numbers, 64 per round across the lanes, and then reads
// the partial sums: everything the workgroups exchange goes through agent-scope atomics, nothing else needs
// ordering.  `partial` is double-buffered by iteration parity (a slot is rewritten only after everybody has passed
// the next exchange).
__device__ __forceinline__ double grid_sum(double *partial, unsigned *seq, int tiles, int it, int me, double mine,
                                           int lane) {
    double *slot = partial + (size_t)(it & 1) * tiles;
    if (lane == 0) {
        __hip_atomic_store(slot + me, mine, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_store(seq + me, (unsigned)it, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    for (;;) {
        bool ready = true;
        for (int t = lane; t < tiles; t += 64)
            ready &= __hip_atomic_load(seq + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >= (unsigned)it;
        if (__all(ready)) break;
        __builtin_amdgcn_s_sleep(4);
    }
    asm volatile("" ::: "memory");
    double acc = 0.0;
    for (int t = lane; t < tiles; t += 64) acc += __hip_atomic_load(slot + t, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    return wave_sum(acc);
}

// Phase A reads its arguments from the kernel-argument segment and is inlined into the iteration loop (as a function
// of its own it saved and restored 48 callee-saved VGPRs per call; see be_fused.hip for why the inlined form needs
// the opaque thread index and -mllvm -disable-machine-licm to stay at 17 spilled values).
typedef const __attribute__((address_space(4))) AdamArgs KArgs;
__device__ __forceinline__ void phase_fg(KArgs *kp, int tile) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    asm volatile("" : "+s"(kp), "+s"(tile));
    fc_fg_tile(kp->fa, tile, lds);
}

__global__ __launch_bounds__(NTHREADS) void adam_fc_kernel(AdamArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    KArgs *kp = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    double *red = reinterpret_cast<double *>(lds + a.red_off);      // [NWAVE] per-state moves, [NWAVE] the batch sum
    const int tile = blockIdx.x, tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = a.fa.n, batch = a.fa.batch;
    const int u = tile * TM + wave;
    const bool mine = u < batch;
    const size_t row = (size_t)(mine ? u : 0) * n;
    const double b1 = 0.9, b2 = 0.999, box = 1. - 1e-8;
    const float c1 = (float)(1. - b1), c2 = (float)(1. - b2);       // NumPy: weak Python scalar times a float32 array
    if (mine)
        for (int j = lane; j < n; j += 64) a.act[row + j] = a.m[row + j] = a.v[row + j] = 0.0;
    __syncthreads();
    double pow1 = 1.0, pow2 = 1.0, drift = -1.0;                    // drift < 0: not yet defined (:186)
    float best_f = 0.f;                                             // wave-uniform, written out when it improves
    int it = 0;
    long long tick = ICNN_BE_PROF_ON(a.fa.prof) ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int phase) {          // diagnostic only (tools/fc_phase_profile.py adam): slots 14/15 of phase A's table
        if (ICNN_BE_PROF_ON(a.fa.prof)) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(a.fa.prof) +
                              ((size_t)tile * NWAVE + wave) * FC_PROF_PHASES + phase,
                          (unsigned long long)(now - tick));
            tick = now;
        }
    };
    for (; it < a.max_iter; ++it) {
        phase_fg(kp, tile);
        __syncthreads();
        if (ICNN_BE_PROF_ON(a.fa.prof)) tick = (long long)__builtin_readcyclecounter();
        // ---- f = negQ + sum_j pen_j, g += d pen / d act; best iterate (:176-183) ----
        double moved = 0.0;
        if (mine) {
            float tot = 0.f;
            for (int j0 = 0; j0 < n; j0 += 64) {
                const int j = j0 + lane;
                float pen = 0.f;
                if (j < n) {
                    const float af = (float)a.act[row + j];
                    const float half = (af + 1.f) * 0.5f;
                    const float p = fminf(fmaxf(half, 1e-4f), 0.9999f);          // tf.clip_by_value, :456
                    const float q = 1.f - p;
                    const float lp = (float)log((double)p), lq = (float)log((double)q);
                    pen = p * lp + q * lq;
                    const bool inside = half >= 1e-4f && half <= 0.9999f;
                    const float ge = a.fa.g[row + j] + (inside ? 0.5f * (lp - lq) : 0.f);
                    a.fa.g[row + j] = ge;                                        // read back by the same lane below
                }
                const int cnt = n - j0 < 64 ? n - j0 : 64;
                for (int l = 0; l < cnt; ++l) tot = tot + lane_value(pen, l);
            }
            const float fe = a.fa.f[u] + tot;
            const bool better = it == 0 || fe < best_f;
            if (better) {
                double d2 = 0.0;
                for (int j = lane; j < n; j += 64) {
                    const double x = a.act[row + j];
                    const double d = it == 0 ? 0.0 : x - a.act_best[row + j];
                    d2 += d * d;
                    a.act_best[row + j] = x;
                }
                best_f = fe;
                if (lane == 0) a.f_best[u] = fe;
                moved = sqrt(wave_sum(d2));
            }
        }
        // ---- stopping rule over the whole batch (:184-192) ----
        if (it > 0) {
            if (lane == 0) red[wave] = moved;
            __syncthreads();
            if (wave == 0) {
                double s = 0.0;
                for (int w = 0; w < NWAVE; ++w) s += red[w];
                if (a.tiles > 1) s = grid_sum(a.partial, a.arrive, a.tiles, it, tile, s, lane);
                if (lane == 0) red[NWAVE] = s;
            }
            __syncthreads();
            const double step_mean = red[NWAVE] / (double)batch;
            drift = drift < 0.0 ? step_mean : 0.5 * drift + 0.5 * step_mean;
            if (drift < 1e-3 && it > 5) break;
        }
        lap(14);
        // ---- moments and the clipped step (:194-203) ----
        pow1 *= b1;
        pow2 *= b2;
        if (mine)
            for (int j = lane; j < n; j += 64) {
                const float ge = a.fa.g[row + j];
                const double mj = b1 * a.m[row + j] + (double)(c1 * ge);
                const double vj = b2 * a.v[row + j] + (double)(c2 * (ge * ge));
                a.m[row + j] = mj;
                a.v[row + j] = vj;
                const double mhat = mj / (1. - pow1);
                double x = a.act[row + j] - (0.01 * mhat) / (sqrt(vj) + 1e-8);
                x = fmin(fmax(x, -box), box);
                a.act[row + j] = x;
            }
        __syncthreads();                                            // act visible to the tile's next phase A
        lap(15);
    }
    if (tile == 0 && tid == 0) *a.iters = it;
}

// ---------------------------------------------------------------------------------------------------------
// Latency path: 1-4 states per workgroup -- the agent's act() optimises ONE observation per environment step
// (RL/src/icnn.py:264-288), its training step a minibatch of 256 (one state per CU here, the stopping rule through
// the same grid barrier as above).  A 16-row MFMA tile would spend the same 31 us per evaluation on one row as on
// sixteen (the chain of k-blocks through a single matrix pipe per SIMD); here each (state, 64 columns) unit is a
// wave of its own running the k-ordered fma chain of its columns on the VALU -- the very order the MFMA
// applies (kk = 16 kb + 4 q + s, s outer; oracle/picnn_chain.c), so the two paths agree bit for bit -- with
// the packed weights read as 16-byte fragments straight from L2 and EVERYTHING else (context rows, activations,
// iterate, moments, best iterate) resident in LDS and registers for the whole loop: no global store until the end.
// ---------------------------------------------------------------------------------------------------------
struct RowsArgs {
    AdamArgs a;
    RowsLayout lay;
    int per_wg;                        // states per workgroup (<= ROWS_MAX); a.tiles = number of workgroups
    CtxRowsArgs cr;                    // cr.obs != nullptr: the context rows are computed here from the observations
};

__global__ __launch_bounds__(RTHREADS) void adam_rows_kernel(RowsArgs r) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const FcArgs &fa = r.a.fa;
    const RowsLayout &lay = r.lay;
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    const int n = fa.n, total = fa.batch, RF = lay.row_floats;
    const int s_base = blockIdx.x * r.per_wg;                         // this workgroup's states
    const int batch = total - s_base < r.per_wg ? total - s_base : r.per_wg;
    float *fbuf = lds + lay.f_off;                                    // [ROWS_MAX] energies
    double *red = reinterpret_cast<double *>(lds + lay.misc_off);     // [ROWS_MAX] moves, [1] the batch sum
    if (r.cr.obs) rows_context_from_obs(r.cr, fa, lay, lds, s_base, batch, tid);
    rows_setup(fa, lay, lds, s_base, batch, tid, r.cr.obs != nullptr);   // act = 0 (:169): all operands zero
    // Adam state of (state = wave, action component = lane) in registers
    const bool mine = wave < batch && lane < n;
    const double b1 = 0.9, b2 = 0.999, box = 1. - 1e-8;
    const float c1 = (float)(1. - b1), c2 = (float)(1. - b2);
    double x = 0.0, m1 = 0.0, m2 = 0.0, best_x = 0.0, pow1 = 1.0, pow2 = 1.0, drift = -1.0;
    float best_f = 0.f;
    int it = 0;
    long long tick = ICNN_BE_PROF_ON(fa.prof) ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int phase) {          // diagnostic only (tools/fc_phase_profile.py adam): [wave][phase] cycle sums
        if (ICNN_BE_PROF_ON(fa.prof)) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(fa.prof) + (size_t)wave * FC_PROF_PHASES + phase,
                          (unsigned long long)(now - tick));
            tick = now;
        }
    };
    for (; it < r.a.max_iter; ++it) {
        // ================= phase A: negQ and d negQ / d act of every state (be_picnn_fc_rows_dev.h) ===============
        rows_eval(fa, lay, lds, batch, tid, lap);
        // ================= phase B: the same operations as adam_fc_kernel, state in registers =================
        double moved = 0.0;
        float ge = 0.f;
        if (wave < batch) {
            const float *row = lds + wave * RF;
            float pen = 0.f;
            if (lane < n) {
                const float af = (float)x;
                const float half = (af + 1.f) * 0.5f;
                const float p = fminf(fmaxf(half, 1e-4f), 0.9999f);
                const float q = 1.f - p;
                const float lp = (float)log((double)p), lq = (float)log((double)q);
                pen = p * lp + q * lq;
                const bool inside = half >= 1e-4f && half <= 0.9999f;
                ge = row[lay.g_off + lane] + (inside ? 0.5f * (lp - lq) : 0.f);
            }
            float tot = 0.f;
            for (int l = 0; l < n; ++l) tot = tot + lane_value(pen, l);
            const float fe = fbuf[wave] + tot;
            if (it == 0 || fe < best_f) {
                const double d = it == 0 || lane >= n ? 0.0 : x - best_x;
                best_x = x;
                best_f = fe;
                moved = sqrt(wave_sum(d * d));
            }
        }
        if (it > 0) {
            if (lane == 0 && wave < ROWS_MAX) red[wave] = wave < batch ? moved : 0.0;
            __syncthreads();
            double ssum = 0.0;
            for (int w = 0; w < batch; ++w) ssum += red[w];
            if (r.a.tiles > 1) {          // several workgroups: one double each, exchanged as in adam_fc_kernel
                if (wave == 0) {
                    const double tsum = grid_sum(r.a.partial, r.a.arrive, r.a.tiles, it, blockIdx.x, ssum, lane);
                    if (lane == 0) red[ROWS_MAX] = tsum;
                }
                __syncthreads();
                ssum = red[ROWS_MAX];
            }
            const double step_mean = ssum / (double)total;
            drift = drift < 0.0 ? step_mean : 0.5 * drift + 0.5 * step_mean;
            if (drift < 1e-3 && it > 5) break;
        }
        lap(14);
        pow1 *= b1;
        pow2 *= b2;
        if (mine) {
            m1 = b1 * m1 + (double)(c1 * ge);
            m2 = b2 * m2 + (double)(c2 * (ge * ge));
            const double mhat = m1 / (1. - pow1);
            x = x - (0.01 * mhat) / (sqrt(m2) + 1e-8);
            x = fmin(fmax(x, -box), box);
            // network input of the next evaluation, already multiplied into every layer's operand y * yu_i
            rows_set_input(fa, lay, lds + wave * RF, lane, (float)x);
        }
        __syncthreads();
        lap(15);
    }
    if (mine) r.a.act_best[(size_t)(s_base + wave) * n + lane] = best_x;
    if (wave < batch && lane == 0) r.a.f_best[s_base + wave] = best_f;
    if (blockIdx.x == 0 && tid == 0) *r.a.iters = it;
}

struct Workspace {
    size_t act, m, v, g, f, partial, arrive, total;
};
Workspace workspace(int batch, int n) {
    Workspace w;
    const size_t bn = (size_t)(batch > 0 ? batch : 1) * n, tiles = (size_t)batch + 1;   // up to one workgroup per state
    size_t o = 0;
    auto take = [&](size_t bytes) { const size_t at = o; o += (bytes + 255) & ~(size_t)255; return at; };
    w.act = take(bn * 8); w.m = take(bn * 8); w.v = take(bn * 8);
    w.g = take(bn * 4); w.f = take((size_t)(batch > 0 ? batch : 1) * 4);
    w.partial = take(2 * tiles * 8); w.arrive = take(tiles * 4);
    w.total = o;
    return w;
}

}  // namespace

size_t adam_workspace_bytes(int batch, int n) { return workspace(batch, n).total; }

// hipErrorNotSupported: more tiles than a cooperative launch can keep resident (the stopping rule needs them all)
// `cx` + `obs` instead of `ctx`: observation -> action in ONE launch (latency path only: at most ROWS_MAX states per
// workgroup, a model without BatchNorm); hipErrorNotSupported otherwise -- the caller produces the context first.
hipError_t launch_adam_fc(const icnn_be_fc_model &m, const float *ctx, int batch, int max_iter, double *act_best,
                          float *f_best, int *iters, void *ws, hipStream_t stream, const icnn_be_fc_ctx *cx,
                          const float *obs) {
    AdamArgs a{};
    int lds = 0;
    if (fill_args(m, a.fa, lds) != 0) return hipErrorInvalidValue;
    const Workspace w = workspace(batch, m.n);
    unsigned char *base = static_cast<unsigned char *>(ws);
    a.act = reinterpret_cast<double *>(base + w.act);
    a.m = reinterpret_cast<double *>(base + w.m);
    a.v = reinterpret_cast<double *>(base + w.v);
    a.partial = reinterpret_cast<double *>(base + w.partial);
    a.arrive = reinterpret_cast<unsigned *>(base + w.arrive);
    a.fa.ctx = ctx; a.fa.y = a.act; a.fa.batch = batch; a.fa.finished = nullptr; a.fa.prof = fc_profile_buffer();
    a.fa.g = reinterpret_cast<float *>(base + w.g);
    a.fa.f = reinterpret_cast<float *>(base + w.f);
    a.act_best = act_best; a.f_best = f_best; a.iters = iters;
    a.max_iter = max_iter;
    a.tiles = (batch + TM - 1) / TM;
    if (m.n <= 64) {   // latency path (everything in LDS and registers): 1-4 states per workgroup, as many workgroups as fit
        RowsArgs r{};
        const int rows_lds = rows_layout(m, ROWS_MAX, r.lay) + (ROWS_MAX + 1) * 8;
        int resident = 1;
        if (rows_lds <= 160 * 1024 && batch > ROWS_MAX) {      // more than one workgroup: they must all be resident
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(adam_rows_kernel), rows_lds); e != hipSuccess) return e;
            int per_cu = 0;
            hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(adam_rows_kernel),
                                                                        RTHREADS, rows_lds);
            if (e != hipSuccess) return e;
            resident = per_cu * device_cus();
        }
        const int per_wg = batch <= ROWS_MAX ? batch : (batch + resident - 1) / (resident > 0 ? resident : 1);
        if (rows_lds <= 160 * 1024 && per_wg >= 1 && per_wg <= ROWS_MAX) {
            r.a = a;
            r.per_wg = per_wg;
            if (obs) {
                if (!cx || cx->batchnorm || cx->n != m.n || cx->n_layers != m.n_layers) return hipErrorNotSupported;
                for (int i = 0; i < m.n_layers; ++i)           // (icnn_be_adam_fc_obs has refused these already: the stage matrices
                    if (cx->width[i] != m.width[i]) return hipErrorInvalidValue;   //  are laid out for cx's widths, read with m's)
                int wmax = cx->n_features;
                for (int i = 0; i + 1 < m.n_layers; ++i) wmax = m.width[i] > wmax ? m.width[i] : wmax;
                if (2 * wmax > r.lay.ctx_off) return hipErrorNotSupported;      // scratch = the row's operand region
                r.cr.obs = obs;
                r.cr.n_features = cx->n_features;
                for (int i = 0; i < m.n_layers; ++i) { r.cr.w_stage[i] = cx->w_stage[i]; r.cr.b_stage[i] = cx->b_stage[i]; }
            }
            r.a.tiles = (batch + per_wg - 1) / per_wg;
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(adam_rows_kernel), rows_lds); e != hipSuccess) return e;
            if (r.a.tiles == 1) {
                hipLaunchKernelGGL(adam_rows_kernel, dim3(1), dim3(RTHREADS), rows_lds, stream, r);
                return hipGetLastError();
            }
            hipError_t e = hipMemsetAsync(r.a.arrive, 0, sizeof(unsigned) * r.a.tiles, stream);
            if (e != hipSuccess) return e;
            void *params[] = {&r};
            return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(adam_rows_kernel), dim3(r.a.tiles),
                                              dim3(RTHREADS), params, (unsigned)rows_lds, stream);
        }
    }
    if (obs) return hipErrorNotSupported;
    a.red_off = (a.fa.lds_floats + 3) & ~3;
    lds = a.red_off * 4 + (NWAVE + 1) * 8;
    if (lds > 160 * 1024) return hipErrorNotSupported;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(adam_fc_kernel), lds); e != hipSuccess) return e;
    if (a.tiles == 1) {
        hipLaunchKernelGGL(adam_fc_kernel, dim3(1), dim3(NTHREADS), lds, stream, a);
        return hipGetLastError();
    }
    int per_cu = 0;
    hipError_t e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, reinterpret_cast<const void *>(adam_fc_kernel),
                                                                NTHREADS, lds);
    if (e != hipSuccess) return e;
    if (a.tiles > per_cu * device_cus()) return hipErrorNotSupported;
    e = hipMemsetAsync(a.arrive, 0, sizeof(unsigned) * a.tiles, stream);
    if (e != hipSuccess) return e;
    void *params[] = {&a};
    return hipLaunchCooperativeKernel(reinterpret_cast<const void *>(adam_fc_kernel), dim3(a.tiles), dim3(NTHREADS),
                                      params, (unsigned)lds, stream);
}

}  // namespace icnn_be
