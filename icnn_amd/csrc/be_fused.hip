// Persistent per-tile solve of a fully-connected PICNN: one workgroup owns a tile of 16 samples for ALL rounds
// of the bundle-entropy loop and alternates the two phases in place,
//     phase A  fc_fg_tile     (16 waves: the MFMA chain of be_picnn_fc_dev.h, energy + gradient of the tile)
//     phase B  dual_step_body (one wave per sample: cut, rank test, projected Newton, y update)
// Both kernels are latency chains -- fc_fg takes as long for one tile as for 256, a dual step as long as its
// slowest sample -- so with one launch per phase every sample waited for the slowest of the whole batch ten times
// per solve (about 15 Newton updates when the mean is 3.8).  Here a tile only waits for the slowest of ITS
// sixteen samples, and there is a single launch.  The arithmetic is exactly that of the two-kernel path (same
// device functions), so the results are bit-identical (tests/test_gpu_parity.py).
//
// LDS is shared in time: phase A uses its activation buffers, phase B the 16 staged bundles plus one shared
// pair of constant rows (zeros / ones for the MFMA operand masking); each phase re-initialises what it needs.
#include "be_dual_dev.h"
#include "be_dual_small_dev.h"
#include "be_picnn_fc_dev.h"
#include "be_picnn_fc_rows_dev.h"

namespace icnn_be {

namespace {

static_assert(NWAVE == TM, "one wave per sample in the dual phase");

// One kernel parameter, read through the kernel-argument segment (s_load) by both phases.
//
// Both phases are INLINED into the round loop.  As non-inlined functions (round 1) each of them saved and restored
// the 44 callee-saved VGPRs it uses on every call: 239 dwords per lane, round and wave = 2.5 GB of scratch traffic
// per solve of 4096 samples, three times the algorithmic bytes (profiles/r02_a_pmc.md).  Inlined naively they
// spilled instead, for a reason that has nothing to do with their own register need: everything derived from
// threadIdx.x (lane, wave, fragment coordinates, LDS offsets) and every literal (the double-precision polynomial
// coefficients of exp) is invariant in the round loop, gets hoisted in front of it -- IR-level LICM for the
// former, MachineLICM for the latter -- and, each phase needing the whole 128-register budget, is then spilled
// and reloaded at every use.  Two measures make the inlined kernel spill-free (0 scratch loads/stores, 127 VGPRs):
// the thread index is read through an opaque move (thread_id(), be_common.h), and this translation unit is
// compiled with -mllvm -disable-machine-licm (icnn_amd/build.py), so literals are materialised where they are used.
struct FusedArgs {
    DualArgs da;       // first: dual_step_body re-reads it at offset 0 of the kernel-argument segment
    FcArgs fa;
    int rounds, crow_off, samples_off, sample_bytes;
    int grouped;       // 1: the sixteen staged bundles of a round do not fit the LDS together (more than ~10 cuts per
                       // sample at n = 159): every round the live samples are dealt into groups whose bundles do fit --
                       // each sample sized for the cuts it holds NOW, finished samples for nothing -- and the groups run
                       // the dual phase one after the other (group_cap = bytes of the staging region, need_off = a
                       // [TM] int array behind the constant rows)
    int group_cap, need_off;
    long long *trace;  // diagnostic (profiling build, icnn_be_debug_trace): [B][ICNN_BE_MAX_ITERS][DUAL_TRACE_WORDS] or null
};
typedef const __attribute__((address_space(4))) FusedArgs KArgs;

// RL: the variant of RL/src/bundle_entropy.py (clipped y, Armijo search, early stop); IPM: the interior-point variant of
// lib/bundle_entropy.py (round 4: the module the reference's scripts import runs through the persistent kernels as well)
// SLICED: the budgeted form (ICNN_BE_FLAG_PERSISTENT | ICNN_BE_FLAG_TIME_SLICE, variant dual): samples may be parked
template <bool RL, int KT, bool IPM = false, bool SLICED = false>
__global__ __launch_bounds__(NTHREADS) void fused_fc_solve_kernel(FusedArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KArgs *kp0 = (KArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    {
        float *crow = reinterpret_cast<float *>(smem + args.crow_off);  // behind phase A's buffers: written once
        for (int j = thread_id(); j < 2 * args.da.ldA; j += NTHREADS) crow[j] = j < args.da.ldA ? 0.f : 1.f;
        if (args.grouped && thread_id() < TM) reinterpret_cast<int *>(smem + args.need_off)[TM + thread_id()] = 0;   // done flags
    }
    const int rounds = args.rounds;
    for (int r = 0; r < rounds; ++r) {
        KArgs *kp = kp0;
        int tile = blockIdx.x;
        asm volatile("" : "+s"(kp), "+s"(tile));                    // nothing derived from the arguments is carried
        fc_fg_tile(kp->fa, tile, reinterpret_cast<float *>(smem));  // phase A: f, g of the tile -> global work arrays
        __syncthreads();                                            // ... visible to the tile's dual waves
        int wave = uni(thread_id() >> 6), round = r;
        asm volatile("" : "+s"(kp), "+s"(tile), "+s"(wave), "+s"(round));
        KArgs &k = *kp;
        const int u = tile * k.fa.tile_rows + wave;
        const bool mine = wave < k.fa.tile_rows && u < k.da.st.batch;
        if (!k.grouped) {
            if (mine) {                                             // phase B: wave w = sample w of the tile
                const int rows_cap = round + 1 < k.da.st.slots ? round + 1 : k.da.st.slots;
                dual_step_body<float, KT, 1, RL, IPM, false, 0, SLICED>(k.da, u, thread_id() & 63,
                                                                        smem + k.samples_off + wave * k.sample_bytes, round, rows_cap,
                                                                        reinterpret_cast<const float *>(smem + k.crow_off));
            }
            __syncthreads();                                        // y, skip flags visible to the next phase A
            continue;
        }
        // phase B in groups: what a sample needs this round follows from the cuts it holds (count + the new one)
        int *need = reinterpret_cast<int *>(smem + k.need_off);
        int kk = 0;
        if (mine && k.da.st.finished[u] == 0 && k.da.st.t_next[u] < k.rounds) kk = k.da.st.count[u] + 1;
        kk = uni(kk);
        // never more rows than the sample has slots: with every slot active and one more cut to take (iterations beyond the
        // slot count recycle the slots of PRUNED cuts only) the dual step must see k > rows_cap and report ICNN_BE_ST_OVERFLOW,
        // as the launch pairs and the per-sample kernel do, instead of writing a 32nd row behind the sample's arrays
        if (kk > k.da.st.slots) kk = k.da.st.slots;
        if ((thread_id() & 63) == 0)
            need[wave] = kk > 0 ? (carve(KT, kk, k.da.ldA, k.da.n_pad, 4, k.da.plan.n_leaves, RL, 1, false, IPM).total + 15) & ~15 : 0;
        __syncthreads();
        // (Round 6, tried and withdrawn: dealing the regions longest-first by the Newton updates of the sample's previous dual
        //  step.  The sample that ends a round is not predictable from its last round -- rank 10 of 16 on average,
        //  profiles/r06_c4_trace_longest_first.txt --, the deal moved the queueing around without shortening a tile's dual phase: 6.51 -> 6.55 ms.)
        int g = 0, off = 0, my_g = 0, my_off = 0;
        for (int i = 0; i < TM; ++i) {                              // the same greedy deal in every wave
            const int nb = need[i];
            if (off + nb > k.group_cap) { ++g; off = 0; }
            if (i == wave) { my_g = g; my_off = off; }
            off += nb;
        }
        my_g = uni(my_g); my_off = uni(my_off);
        int alive = kk > 0;
        // The groups do not wait for each other as groups: a sample of a later group starts as soon as every sample of an
        // EARLIER group whose staging region overlaps its own has finished (a flag per sample, stamped with the round; only
        // waves of earlier groups are ever waited for, the first group waits for nobody: no cycle, and a finished wave's
        // flag is set on every path out of the dual step).  A tile's dual phase then lasts as long as its longest chain of
        // overlapping samples instead of the sum over groups of each group's slowest sample.
        int *done = need + TM;
        // diagnostic (profiling build): cycles this sample spends queueing for its staging region (phase 14) and, below,
        // waiting at the barrier that ends the tile's dual phase (phase 15); tools/c4_timeline.py
        long long tq = ICNN_BE_PROF_ON(k.da.prof) ? (long long)__builtin_readcyclecounter() : 0;
        auto wait_lap = [&](int phase) {
            if (ICNN_BE_PROF_ON(k.da.prof)) {
                const long long now = (long long)__builtin_readcyclecounter();
                if (mine && (thread_id() & 63) == 0)
                    atomicAdd(reinterpret_cast<unsigned long long *>(k.da.prof) + (size_t)u * DUAL_PROF_PHASES + phase,
                              (unsigned long long)(now - tq));
                tq = now;
            }
        };
        long long *tr = nullptr;
        if (ICNN_BE_PROF_ON(k.trace) && kk > 0 && (thread_id() & 63) == 0) {
            tr = k.trace + ((size_t)u * ICNN_BE_MAX_ITERS + round) * DUAL_TRACE_WORDS;
            tr[0] = tq; tr[3] = k.da.st.newton_iters[u];
        }
        if (kk > 0) {
            if (my_g > 0) {
                const int my_end = my_off + need[wave];
                int gi = 0, oi = 0;
                for (int i = 0; i < TM; ++i) {
                    const int nb = need[i];
                    if (oi + nb > k.group_cap) { ++gi; oi = 0; }
                    if (gi >= my_g) break;
                    if (nb > 0 && oi < my_end && oi + nb > my_off)
                        while (__atomic_load_n(&done[i], __ATOMIC_RELAXED) != round + 1) __builtin_amdgcn_s_sleep(4);
                    oi += nb;
                }
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            }
            wait_lap(14);
            if (ICNN_BE_PROF_ON(k.trace) && tr) tr[1] = (long long)__builtin_readcyclecounter();
            dual_step_body<float, KT, 1, RL, IPM, false, 0, SLICED>(k.da, u, thread_id() & 63, smem + k.samples_off + my_off, round, kk,
                                                                    reinterpret_cast<const float *>(smem + k.crow_off));
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");      // its LDS traffic is complete before the flag is seen
            if ((thread_id() & 63) == 0) __atomic_store_n(&done[wave], round + 1, __ATOMIC_RELAXED);
            if (ICNN_BE_PROF_ON(k.da.prof)) tq = (long long)__builtin_readcyclecounter();
            if (ICNN_BE_PROF_ON(k.trace) && tr) tr[2] = (long long)__builtin_readcyclecounter();
        }
        const int any_alive = __syncthreads_or(alive);
        if (kk > 0) wait_lap(15);
        if (!any_alive) break;                        // (also the barrier that ends the dual phase) no sample
                                                                    // of the tile has work left: none gets any later either
    }
}

// ---------------------------------------------------------------------------------------------------------
// Batches of at most one sample per CU: a persistent workgroup PER SAMPLE (8 waves).  Phase A is the VALU
// evaluation of be_picnn_fc_rows_dev.h (all waves), phase B the dual step of the sample on wave 0; the context
// row and the iteration-invariant products stay in LDS for the whole solve, both phases have their own LDS
// regions.  Besides the single launch, every sample now runs at its own pace: the solve ends with the sample
// whose ten rounds are longest in SUM, not with the sum over rounds of the slowest sample of each round, and a
// sample that leaves the loop (rank test, RL stall) frees its CU at once.  Same device functions, same bits.
// ---------------------------------------------------------------------------------------------------------
struct FusedRowsArgs {
    DualArgs da;       // first: dual_step_body re-reads it at offset 0 of the kernel-argument segment
    FcArgs fa;
    RowsLayout lay;
    int rounds, per_wg;                // samples per workgroup (1 or 2)
    int dual_off, sample_bytes, crow_off;    // byte offsets of the dual steps' regions and of the constant rows
    int iters;                         // outer iterations (icnn_be_state.iters, or its slots)
    int resume;                        // finishing pass after time-sliced rounds: only samples that still have rounds to
                                       // run (parked in a Newton loop or behind by the rounds they were parked in) do
                                       // anything, each from its own outer-iteration counter, with no update budget
};
typedef const __attribute__((address_space(4))) FusedRowsArgs KRArgs;

// phase A of the per-sample kernel: the VALU evaluation of be_picnn_fc_rows_dev.h on all waves
__device__ __forceinline__ void rows_phase_fg(KRArgs *kp, int s_base, int batch, int tid) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    KRArgs &k = *kp;
    float *lds = reinterpret_cast<float *>(smem);
    const int wave = tid >> 6, lane = tid & 63, n = k.fa.n, RF = k.lay.row_floats;
    long long tick = ICNN_BE_PROF_ON(k.fa.prof) ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int phase) {          // diagnostic only (tools/rows_phase_profile.py)
        if (ICNN_BE_PROF_ON(k.fa.prof)) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(k.fa.prof) +
                              ((size_t)blockIdx.x * RWAVES + wave) * FC_PROF_PHASES + phase,
                          (unsigned long long)(now - tick));
            tick = now;
        }
    };
    if (wave < batch)                    // network input: y rounded to float32 like a TensorFlow feed; RL wrapper feeds 2y-1
        for (int j = lane; j < n; j += 64) {
            const double yd = k.da.st.y[(size_t)(s_base + wave) * n + j];
            rows_set_input(k.fa, k.lay, lds + wave * RF, j, k.fa.action_box ? (float)(2.0 * yd - 1.0) : (float)yd);
        }
    __syncthreads();
    lap(13);
    rows_eval(k.fa, k.lay, lds, batch, tid, lap);
    if (wave < batch) {                  // hand-over to the dual step of the sample (same wave) through its work arrays
        const float gscale = k.fa.action_box ? 2.f : 1.f;
        if (lane == 0) k.fa.f[s_base + wave] = lds[k.lay.f_off + wave];
        for (int j = lane; j < n; j += 64) k.fa.g[(size_t)(s_base + wave) * n + j] = gscale * lds[wave * RF + k.lay.g_off + j];
    }
    lap(14);
}

// KS > 0: narrow rows (n <= 16, variant RL): the dual steps of ALL samples of the workgroup (at most four) run on wave 0, one
// sample per 16-lane DPP row with the bundle in registers (be_dual_small_dev.h, KS = its row count) -- the RL agent's act()
// and its replay batches up to four samples per CU.  Bit-identical to the wave-per-sample phase.
template <bool RL, int KT, int KS = 0, bool IPM = false>
__global__ __launch_bounds__(RTHREADS) void fused_rows_solve_kernel(FusedRowsArgs args) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int per_wg = args.per_wg;
    int s_base0 = blockIdx.x * per_wg;
    const int batch0 = args.da.st.batch - s_base0 < per_wg ? args.da.st.batch - s_base0 : per_wg;
    KRArgs *kp0 = (KRArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    if (args.resume) {                                              // nothing to do for a workgroup whose samples are done
        int todo = 0;
        if (thread_id() < batch0) {
            const int u = s_base0 + thread_id();
            todo = args.da.st.finished[u] == 0 && args.da.st.t_next[u] < args.iters;
        }
        if (!__syncthreads_or(todo)) return;
    }
    {
        float *crow = reinterpret_cast<float *>(smem + args.crow_off);
        for (int j = thread_id(); j < 2 * args.da.ldA; j += RTHREADS) crow[j] = j < args.da.ldA ? 0.f : 1.f;
    }
    rows_setup(args.fa, args.lay, reinterpret_cast<float *>(smem), s_base0, batch0, thread_id());
    const int rounds = args.rounds;
    for (int r = 0; r < rounds; ++r) {
        // both phases inlined, arguments and thread index re-read opaquely per round (see fused_fc_solve_kernel)
        KRArgs *kp = kp0;
        int s_base = s_base0, batch = batch0, round = r;
        asm volatile("" : "+s"(kp), "+s"(s_base), "+s"(batch), "+s"(round));
        rows_phase_fg(kp, s_base, batch, thread_id());
        __syncthreads();                                            // f, g visible (written by the dual wave itself)
        const int wave = uni(thread_id() >> 6);
        asm volatile("" : "+s"(kp), "+s"(s_base), "+s"(batch), "+s"(round));
        KRArgs &k = *kp;
        int live = 0;                                               // (each dual wave reads the flags it wrote itself)
        if constexpr (KS > 0) {
            if (wave == 0) {
                dual_step_quad_rl<float, KS>(k.da, s_base, batch, round);
                if ((thread_id() & 63) < batch) live = k.da.st.skip_fg[s_base + (thread_id() & 63)] == 0;
            }
        } else {
            if (wave < batch) {
                const int rows_cap = !k.resume && round + 1 < k.da.st.slots ? round + 1 : k.da.st.slots;
                dual_step_body<float, KT, 1, RL, IPM>(k.da, s_base + wave, thread_id() & 63, smem + k.dual_off + wave * k.sample_bytes,
                                                      round, rows_cap, reinterpret_cast<const float *>(smem + k.crow_off));
            }
            if (wave < batch && (thread_id() & 63) == 0) {
                const int u = s_base + wave;
                live = k.resume ? (k.da.st.finished[u] == 0 && k.da.st.t_next[u] < k.iters) : k.da.st.skip_fg[u] == 0;
            }
        }
        if (!__syncthreads_or(live)) break;                         // every sample of the workgroup has left the loop
    }
}

}  // namespace

// Returns hipErrorNotSupported when the shape does not fit this path (the caller falls back to one launch per phase).
hipError_t launch_fused_rows_solve(const icnn_be_fc_model &m, const float *ctx, const icnn_be_state &st, float *f_work,
                                   float *g_work, int per_wg, long long *dual_prof, hipStream_t stream, bool resume) {
    const bool rl = st.variant == ICNN_BE_VARIANT_RL;
    const bool ipm = st.variant == ICNN_BE_VARIANT_PDIPM;
    if (st.cut_dtype != ICNN_BE_CUT_F32 || per_wg < 1 || per_wg > ROWS_MAX) return hipErrorNotSupported;
    if (dual_waves(st.n, st.cut_dtype, st.variant) != 1 || (ipm && resume)) return hipErrorNotSupported;
    FusedRowsArgs args{};
    int unused = 0;
    if (fill_args(m, args.fa, unused) != 0) return hipErrorInvalidValue;
    args.fa.ctx = ctx; args.fa.y = st.y; args.fa.f = f_work; args.fa.g = g_work; args.fa.finished = nullptr;
    args.fa.batch = st.batch; args.fa.prof = fc_profile_buffer();
    DualArgs &da = args.da;
    da.st = st;
    da.f = f_work;
    da.g = g_work;
    da.round = 0;
    da.budget = 0;
    da.n_pad = (st.n + 15) & ~15;
    da.ldA = dual_row_pitch(da.n_pad);
    da.rows = st.slots;
    da.prof = dual_prof;
    if (!pw_build(da.plan, st.n)) return hipErrorInvalidValue;
    const bool big = st.slots > 15;
    const int rows_bytes = (rows_layout(m, per_wg, args.lay) + 15) & ~15;
    args.sample_bytes = (carve(big ? 32 : 16, st.slots, da.ldA, da.n_pad, 4, da.plan.n_leaves, rl, 1, false, ipm).total + 15) & ~15;
    args.per_wg = per_wg;
    args.dual_off = rows_bytes;
    args.crow_off = rows_bytes + per_wg * args.sample_bytes;
    const int lds = args.crow_off + ((2 * da.ldA * 4 + 15) & ~15);
    if (lds > 160 * 1024) return hipErrorNotSupported;
    args.rounds = args.iters = st.iters > 0 ? st.iters : st.slots;
    args.resume = resume ? 1 : 0;
    const int which = (rl ? 1 : 0) + (big ? 2 : 0);
    auto kern = which == 0 ? fused_rows_solve_kernel<false, 16> : which == 1 ? fused_rows_solve_kernel<true, 16>
              : which == 2 ? fused_rows_solve_kernel<false, 32> : fused_rows_solve_kernel<true, 32>;
    if (ipm) kern = big ? fused_rows_solve_kernel<false, 32, 0, true> : fused_rows_solve_kernel<false, 16, 0, true>;
    if (!resume && dual_step_small_fits(st, 0))         // narrow rows, variant RL: all dual steps of the workgroup on wave 0
        kern = st.slots <= 5 ? fused_rows_solve_kernel<true, 16, 5> : st.slots <= 7 ? fused_rows_solve_kernel<true, 16, 8>
                                                                                  : fused_rows_solve_kernel<true, 16, 16>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3((st.batch + per_wg - 1) / per_wg), dim3(RTHREADS), lds, stream, args);
    return hipGetLastError();
}

// Returns hipErrorNotSupported when the shape does not fit this path (the caller falls back to one launch per phase).
hipError_t launch_fused_fc_solve(const icnn_be_fc_model &m, const float *ctx, const icnn_be_state &st, float *f_work,
                                 float *g_work, long long *dual_prof, hipStream_t stream, int tile_rows, int budget) {
    const bool rl = st.variant == ICNN_BE_VARIANT_RL, ipm = st.variant == ICNN_BE_VARIANT_PDIPM;
    if (st.cut_dtype != ICNN_BE_CUT_F32) return hipErrorNotSupported;
    if (dual_waves(st.n, st.cut_dtype, st.variant) != 1 || ((ipm || rl) && budget > 0)) return hipErrorNotSupported;
    const bool big = st.slots > 15;
    FcArgs fa{};
    int fg_bytes = 0;
    if (fill_args(m, fa, fg_bytes) != 0) return hipErrorInvalidValue;
    fa.ctx = ctx; fa.y = st.y; fa.f = f_work; fa.g = g_work; fa.finished = st.skip_fg; fa.batch = st.batch;
    fa.prof = nullptr;
    if (tile_rows != 4 && tile_rows != 8 && tile_rows != TM) return hipErrorInvalidValue;
    fa.tile_rows = tile_rows;
    DualArgs da;
    da.st = st;
    da.f = f_work;
    da.g = g_work;
    da.round = 0;
    da.budget = budget;
    da.n_pad = (st.n + 15) & ~15;
    da.ldA = dual_row_pitch(da.n_pad);
    da.rows = st.slots;
    da.prof = dual_prof;
    if (!pw_build(da.plan, st.n)) return hipErrorInvalidValue;
    const int KT = big ? 32 : 16;
    const int sample_bytes = (carve(KT, st.slots, da.ldA, da.n_pad, 4, da.plan.n_leaves, rl, 1, false, ipm).total + 15) & ~15;
    // phase B: sixteen bundles from offset 0 (they overlay phase A's buffers); the shared constant rows live behind
    // whichever region is larger, where neither phase overwrites them
    const int samples_off = 0, crow_bytes = (2 * da.ldA * 4 + 15) & ~15;
    const int dual_bytes = samples_off + TM * sample_bytes;
    int crow_off = ((fg_bytes > dual_bytes ? fg_bytes : dual_bytes) + 15) & ~15;
    int lds = crow_off + crow_bytes;
    FusedArgs args;
    args.grouped = 0; args.group_cap = 0; args.need_off = 0;
    if (lds > 160 * 1024 || big) {
        // the sixteen full-size bundles do not fit together: groups sized by what the samples hold (FusedArgs::grouped).
        // The staging region takes everything the workgroup can have; one sample's largest bundle must fit it.
        const int need_bytes = 2 * TM * 4;            // per sample: bytes needed this round, and its done flag
        crow_off = (160 * 1024 - 1024 - crow_bytes - need_bytes) & ~15;      // (1 KB: the kernel's static LDS, 256 B today)
        if (crow_off < fg_bytes || crow_off < sample_bytes) return hipErrorNotSupported;
        args.grouped = 1; args.group_cap = crow_off; args.need_off = crow_off + crow_bytes;
        lds = args.need_off + need_bytes;
    }
    auto kern = big ? (rl ? fused_fc_solve_kernel<true, 32> : fused_fc_solve_kernel<false, 32>)
                    : (rl ? fused_fc_solve_kernel<true, 16> : fused_fc_solve_kernel<false, 16>);
    if (ipm) kern = big ? fused_fc_solve_kernel<false, 32, true> : fused_fc_solve_kernel<false, 16, true>;
    if (budget > 0) kern = big ? fused_fc_solve_kernel<false, 32, false, true> : fused_fc_solve_kernel<false, 16, false, true>;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    args.da = da; args.fa = fa; args.trace = dual_trace_buffer();
    args.rounds = st.iters > 0 ? st.iters : st.slots; args.crow_off = crow_off; args.samples_off = samples_off; args.sample_bytes = sample_bytes;
    hipLaunchKernelGGL(kern, dim3((st.batch + tile_rows - 1) / tile_rows), dim3(NTHREADS), lds, stream, args);
    return hipGetLastError();
}

}  // namespace icnn_be
