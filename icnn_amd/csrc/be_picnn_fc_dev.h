// Device code of the FC-PICNN energy/gradient (included by be_picnn_fc.hip and be_fused.hip).
#pragma once
// Fused forward + y-gradient of the y-dependent part of a fully-connected PICNN.
//
//   z_i = act( (z_{i-1} * gate_i) Wzu_i + (y * yu_i) Wyu_i + zu_i ),  i = 0..L,  E = z_L
//   (multi-label-cls/icnn_ebundle.py:349-388, RL/src/icnn.py:356-404) and
//   dE/dy = sum_i yu_i * (delta_i Wyu_i^T),  delta_{i-1} = gate_i * (delta_i Wzu_i^T) * act'(.)
//   (what tf.gradients(E_, y_) evaluates, icnn_ebundle.py:146).
//
// One workgroup owns a tile of TM = 16 samples (one MFMA M-tile) for the whole chain: the
// activations never leave LDS, the x-only context (yu, zu, gate) is read once per layer with
// coalesced loads, and the non-negative W^(z) / unconstrained W^(y) weights are streamed from
// L2/HBM exactly once per workgroup per pass as pre-packed v_mfma_f32_16x16x4_f32 B-fragments
// (16 B per lane, 1 KiB per wave-instruction).  Each wave owns a strided set of 16-column
// output tiles.  fp32 throughout, like the reference's TensorFlow graph.
#include <hip/hip_runtime.h>

#include "be_common.h"
#include "be_kernels.h"
#include "icnn_be.h"

namespace icnn_be {

namespace {

constexpr int TM = 16;     // samples per workgroup
constexpr int NWAVE = 16;  // waves per workgroup
constexpr int NTHREADS = NWAVE * 64;

__host__ __device__ inline int pad16(int v) { return (v + 15) & ~15; }

struct FcArgs {
    int n, L;                              // L = number of hidden z-layers (n_layers - 1)
    int width[ICNN_BE_MAX_LAYERS];
    float alpha;
    int action_box;
    int ctx_width;
    int yu_off[ICNN_BE_MAX_LAYERS], zu_off[ICNN_BE_MAX_LAYERS], gate_off[ICNN_BE_MAX_LAYERS];
    long long w_yu_f[ICNN_BE_MAX_LAYERS], w_yu_b[ICNN_BE_MAX_LAYERS];   // float offsets into wpack
    long long w_zu_f[ICNN_BE_MAX_LAYERS], w_zu_b[ICNN_BE_MAX_LAYERS];
    int zb_off[ICNN_BE_MAX_LAYERS], zb_ld[ICNN_BE_MAX_LAYERS];          // LDS float offsets / pitches
    int ldY, ybuf_off, aop_off[ICNN_BE_MAX_LAYERS], gbuf_off, dl_off, lds_floats;   // aop_i = y * yu_i
    const float *wpack, *ctx;
    const double *y;
    float *f, *g;
    const int *finished;
    int batch;
    int tile_rows;      // samples a workgroup's 16-row MFMA tile actually holds (16; 4 or 8 when the batch has fewer
                        // than a full tile per CU: be_fused.hip) -- the other rows are zero
    long long *prof;    // diagnostic: [workgroup][wave][FC_PROF_PHASES] cycle counters, else nullptr
};
constexpr int FC_PROF_PHASES = 16;

// PF: the k-blocks of every packed operand are padded (zero fragments) to a multiple of it -- the unroll factor and ring
// size of the per-sample kernel's fma chains (gemv_chain, be_picnn_fc_rows_dev.h), whose body then needs no bounds checks.
// TILE_RD: depth of the B-fragment register ring (= unroll factor) of the MFMA tile loops.  Measured on one box, whole
// evaluation: depth 5 over the padded k-blocks 124.5 k cycles, depth 2 over the padded k-blocks the same, depth 2 over the
// real k-blocks (38 instead of 40 for K = 600) 118.5 k, depth 1 147 k, depth 10 (one-tile loops) 125.6 k.
constexpr int PF = 5;
constexpr int TILE_RD = 2;
// TILE_PAIR: a wave takes its tiles nt, nt + NWAVE as a pair that shares every A fragment read from LDS (two-tile loop of
// gemm_loop) instead of one after the other.  Off: on one box 1.036 against 1.042 ms for the headline solve -- the LDS reads
// were never the bound -- with eight accumulator and ring registers fewer in the 128-register kernels.
constexpr bool TILE_PAIR = false;
constexpr int TILE_STEP = (TILE_PAIR ? 2 : 1) * NWAVE;
__host__ __device__ inline int kblocks(int K) { return (pad16(K) / 16 + PF - 1) / PF * PF; }
// k-blocks the MFMA tile loops of fc_fg_tile walk: the real ones, up to a multiple of their ring depth (the pack's further
// zero fragments would add nothing: acc + 0 * a == acc bit for bit, acc is never -0)
// (an odd number of real k-blocks that IS the pack's -- 5, 15, 25: widths 65-80, 225-240, ... -- has no zero fragment to round
// up into; gemm_tiles runs their last k-block behind the loop)
__host__ __device__ inline int kblocks_tile(int K) {
    const int up = (pad16(K) / 16 + TILE_RD - 1) / TILE_RD * TILE_RD;
    return up <= kblocks(K) ? up : kblocks(K);
}

// LDS row pitch (floats): multiple of 4 and == 8 (mod 64) so that the ds_read_b128 A-fragment gather
// (16 rows x 4 k-quads) is bank-conflict free, and wide enough for every k-block the GEMM
// loops read (the padded ones included).
__host__ __device__ inline int lds_pitch(int width) {
    int p = kblocks(width) * 16;
    while ((p & 63) != 8) p += 4;
    return p;
}

// floats of one packed GEMM operand W[K][N]
inline size_t packed_floats(int K, int N) { return (size_t)kblocks(K) * (pad16(N) / 16) * 256; }

// pack[(kb*NT + nt)*256 + lane*4 + s] = W[kb*16 + 4*(lane>>4) + s][nt*16 + (lane&15)]
// `transpose`: the logical operand is src^T (src stored [N][K] row-major).
// k-block major: at any moment the eight waves of a workgroup (each on its own output tiles, all at about
// the same k-block) read neighbouring 1 KiB fragments, i.e. one contiguous 8-16 KiB window that spreads
// over all L2 channels.  With the tile-major order used at first, the sixteen concurrent streams were
// 10 or 38 KiB apart and marched through the same few channels in lockstep: 13 B/clk per CU instead of
// the ~55 B/clk a workgroup can pull from L2 (tools/probes/l2_stream_probe.hip).
void pack_operand(const float *src, int K, int N, bool transpose, float *dst) {
    const int KB = kblocks(K), NT = pad16(N) / 16;
    for (int nt = 0; nt < NT; ++nt)
        for (int kb = 0; kb < KB; ++kb)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const int kk = kb * 16 + 4 * (lane >> 4) + s, nn = nt * 16 + (lane & 15);
                    float v = 0.f;
                    if (kk < K && nn < N) v = transpose ? src[(size_t)nn * K + kk] : src[(size_t)kk * N + nn];
                    dst[((size_t)(kb * NT + nt) * 64 + lane) * 4 + s] = v;
                }
}

struct PackOffsets {
    long long yu_f[ICNN_BE_MAX_LAYERS], yu_b[ICNN_BE_MAX_LAYERS], zu_f[ICNN_BE_MAX_LAYERS],
        zu_b[ICNN_BE_MAX_LAYERS];
    size_t total;
};
PackOffsets pack_offsets(const icnn_be_fc_model &m) {
    PackOffsets o{};
    size_t at = 0;
    const int L = m.n_layers - 1;
    for (int i = 0; i <= L; ++i) {
        const int wi = m.width[i];
        if (i < L) {
            o.yu_f[i] = (long long)at; at += packed_floats(m.n, wi);
            o.yu_b[i] = (long long)at; at += packed_floats(wi, m.n);
            if (i > 0) {
                o.zu_f[i] = (long long)at; at += packed_floats(m.width[i - 1], wi);
                o.zu_b[i] = (long long)at; at += packed_floats(wi, m.width[i - 1]);
            }
        } else {   // final scalar layer: plain vectors, 16-float aligned
            o.yu_f[i] = o.yu_b[i] = (long long)at; at += (size_t)pad16(m.n);
            o.zu_f[i] = o.zu_b[i] = (long long)at; at += (size_t)pad16(m.width[i - 1]);
        }
    }
    o.total = at;
    return o;
}

__device__ __forceinline__ float act_fn(float p, float alpha) { return p > 0.f ? p : alpha * p; }

// acc{0,1} += A[16][K] (LDS, pitch ld) * packed weight tiles nt0 / nt1 (nt1 < 0: only one tile; with TILE_PAIR off that is
// every call).  Two output tiles would share every A fragment read.  The B fragments (16 B per lane from L2) run an RD-deep
// register ring ahead of the MFMAs and the A fragment of the next k-block is read before the MFMAs of the current one.  KB is a
// multiple of RD (kblocks_tile: the real k-blocks; the LDS pad columns behind them are zero), the one- and two-tile cases are
// separate loops and the tile indices are wave-uniform: the loop body is straight-line code whose accumulators and ring slots
// never change registers (a copy of an MFMA result drains the matrix pipe; a copy of a fragment waits for its load).  What the
// machine code has to look like, and what it took (round 4, read off the ISA): the first RD requests in the order of use --
// otherwise the wait-count pass, merging that order with the loop's at the loop header, drains the ring with vmcnt(0) in every
// turn --, and a slot's refill behind the MFMAs that read it.
// Per output element the accumulation is the k-ordered fma chain oracle/picnn_chain.c reproduces.
template <bool TWO, int RD>              // RD: depth of the fragment ring = unroll factor; KB a multiple of it
__device__ __forceinline__ void gemm_loop(const float *ap, const f4 *bp0, const f4 *bp1, size_t kstride, int KB,
                                          f4 &acc0, f4 &acc1) {
    f4 b0[RD], b1[RD];
#pragma unroll
    for (int d = 0; d < RD; ++d) {       // issued in the order of use, pinned: the wait-count pass merges this order with the
        b0[d] = bp0[(size_t)d * kstride];   // loop's own at the loop header, and a fragment requested out of turn here costs
        __builtin_amdgcn_sched_barrier(0);  // a full drain of the ring (s_waitcnt vmcnt(0)) in EVERY turn of the loop
        if (TWO) {
            b1[d] = bp1[(size_t)d * kstride];
            __builtin_amdgcn_sched_barrier(0);
        }
    }
    f4 an = *reinterpret_cast<const f4 *>(ap);
    for (int kb0 = 0; kb0 < KB; kb0 += RD) {
#pragma unroll
        for (int d = 0; d < RD; ++d) {
            const int kb = kb0 + d;
            const f4 a = an;
            an = *reinterpret_cast<const f4 *>(ap + (kb + 1 < KB ? kb + 1 : kb) * 16);
            const f4 x0 = b0[d], x1 = b1[d];
            const int nk = kb + RD < KB ? kb + RD : kb;          // ring refill (clamped re-read at the tail)
            if (TWO) {
                b0[d] = bp0[(size_t)nk * kstride];
                b1[d] = bp1[(size_t)nk * kstride];
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.y, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x0.z, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x0.w, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x1.w, acc1, 0, 0, 0);
#pragma unroll
            for (int g = 0; g < (TWO ? 8 : 0); ++g) {            // one MFMA, then up to two other instructions
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x126, 2, 0);
            }
            if (!TWO) {
                // one tile per wave: the slot is refilled BEHIND the MFMAs that read it, pinned there -- requested in front
                // of them the new fragment needs a second register and a copy, and the copy waits for the load on the spot
                __builtin_amdgcn_sched_barrier(0);
                b0[d] = bp0[(size_t)nk * kstride];
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
}
__device__ __forceinline__ void gemm_tiles(const float *A, int ld, const float *Wp, int KB, int NT, int nt0, int nt1,
                                           f4 &acc0, f4 &acc1) {
    const int lane = thread_id() & 63, r16 = lane & 15, q = lane >> 4;
    nt0 = __builtin_amdgcn_readfirstlane(nt0);
    nt1 = __builtin_amdgcn_readfirstlane(nt1);
    const float *ap = A + r16 * ld + 4 * q;
    const f4 *bp0 = reinterpret_cast<const f4 *>(Wp) + (size_t)nt0 * 64 + lane;
    const f4 *bp1 = reinterpret_cast<const f4 *>(Wp) + (size_t)(nt1 >= 0 ? nt1 : nt0) * 64 + lane;
    const size_t kstride = (size_t)NT * 64;              // f4 elements between consecutive k-blocks of a tile
    static_assert(TILE_RD == 2, "the odd k-block below");
    const int KBe = KB & ~1;
    if (nt1 >= 0) gemm_loop<true, TILE_RD>(ap, bp0, bp1, kstride, KBe, acc0, acc1);
    else gemm_loop<false, TILE_RD>(ap, bp0, bp1, kstride, KBe, acc0, acc1);
    // (see kblocks_tile: rare widths; one k-block without a ring behind the loop.  A second loop instance with a ring of PF for
    //  them -- never executed on the benchmark's network -- cost it 3 %: 1.069 against 1.038 ms on one box)
    if (KB & 1) {
        const f4 a = *reinterpret_cast<const f4 *>(ap + KBe * 16);
        const f4 x0 = bp0[(size_t)KBe * kstride];
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.y, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x0.z, acc0, 0, 0, 0);
        acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x0.w, acc0, 0, 0, 0);
        if (nt1 >= 0) {
            const f4 x1 = bp1[(size_t)KBe * kstride];
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.x, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x1.y, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x1.z, acc1, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x1.w, acc1, 0, 0, 0);
        }
    }
}

// One tile of TM samples (workgroup-wide: NTHREADS threads, `lds` = the dynamic shared memory of the workgroup).
// Stand-alone kernel: one workgroup per tile.  be_fused.hip calls it once per round from its persistent workgroup.
template <typename ArgsT>          // FcArgs by value, or a reference into the kernel-argument segment (be_fused.hip)
__device__ __forceinline__ void fc_fg_tile(const ArgsT &a, int tile, float *lds) {
    // Every float32 operation below is written out (explicit fmaf, no compiler contraction) so that
    // oracle/picnn_chain.c can reproduce the kernel's result bit for bit.
#pragma clang fp contract(off)
    const int tid = thread_id(), lane = tid & 63, wave = tid >> 6;
    const int r16 = lane & 15, q = lane >> 4;
    const int TR = a.tile_rows;
    const int s0 = tile * TR;
    const int rows = min(TR, a.batch - s0);
    const int n = a.n, L = a.L, C = a.ctx_width, ldY = a.ldY;
    const int npad = pad16(n);
    float *ybuf = lds + a.ybuf_off;      // y (network input)
    long long tick = ICNN_BE_PROF_ON(a.prof) ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int phase) {          // diagnostic only (tools/fc_phase_profile.py)
        if (ICNN_BE_PROF_ON(a.prof)) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(a.prof) +
                              ((size_t)tile * NWAVE + wave) * FC_PROF_PHASES + phase,
                          (unsigned long long)(now - tick));
            tick = now;
        }
    };

    // nothing to do if every sample of the tile has left the loop -- the flags are REQUESTED here and looked at behind the
    // first batch of input loads below (one memory round trip per evaluation instead of two in a row)
    int live = 1;
    if (a.finished) {
        live = 0;
        if (tid < rows) live = a.finished[s0 + tid] == 0;
    }
    const float *ctx = a.ctx + (size_t)s0 * C;
    float *gbuf = lds + a.gbuf_off;      // dE/dy accumulator of the backward pass
    float *dl = lds + a.dl_off;          // delta_{L-1} (the activations z_{L-1} stay intact for the energy)

    // Wave w prepares row w of the tile (TM == NWAVE): no index arithmetic beyond lane strides.
    // The GEMMs read k-blocks up to a multiple of PF, i.e. pad columns [pad16(width), pitch) that no phase writes:
    // zero them (their packed weights are zero, but 0 * stale-NaN would not be).  Everything below pad16(width) is
    // written for all TM rows by the phase that produces the buffer.
    static_assert(TM == NWAVE, "one wave per row in the preparation phase");
    {
        auto zero_pad = [&](float *buf, int ld, int width) {
            const int w16 = pad16(width);
            for (int j = w16 + lane; j < ld; j += 64) buf[wave * ld + j] = 0.f;
        };
        for (int i = 0; i < L; ++i) zero_pad(lds + a.aop_off[i], ldY, n);
        for (int i = 0; i < L; ++i) zero_pad(lds + a.zb_off[i], a.zb_ld[i], a.width[i]);
        zero_pad(dl, a.zb_ld[L - 1], a.width[L - 1]);
    }
    // network input: y rounded to float32 like a TensorFlow feed (RL wrapper feeds 2y-1), and with it the y-operand
    // y * yu_i of EVERY layer; all loads of a lane's elements are issued before the first use
    {
        const int r = wave;
        const bool row_ok = r < rows;
        for (int j0 = 0; j0 < npad; j0 += 4 * 64) {
            double yd[4];
            float cu[4][ICNN_BE_MAX_LAYERS];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 64 * k + lane;
                const bool ok = row_ok && j < n;
                yd[k] = ok ? a.y[(size_t)(s0 + r) * n + j] : 0.0;
#pragma unroll
                for (int i = 0; i < ICNN_BE_MAX_LAYERS; ++i)
                    cu[k][i] = ok && i < L ? ctx[(size_t)r * C + a.yu_off[i] + j] : 0.f;
            }
            if (j0 == 0 && a.finished && !__syncthreads_or(live)) return;     // (uniform: every thread takes j0 = 0)
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 64 * k + lane;
                if (j < npad) {
                    const float v = a.action_box ? (float)(2.0 * yd[k] - 1.0) : (float)yd[k];
                    const bool ok = row_ok && j < n;
                    ybuf[r * ldY + j] = ok ? v : 0.f;
#pragma unroll
                    for (int i = 0; i < ICNN_BE_MAX_LAYERS; ++i)
                        if (i < L) lds[a.aop_off[i] + r * ldY + j] = ok ? v * cu[k][i] : 0.f;
                }
            }
        }
    }
    __syncthreads();
    lap(0);

    // ---------------- forward ------------------------------------------------------------
    const float *wyL = a.wpack + a.w_yu_f[L];       // final scalar layer: plain vectors
    const float *wzL = a.wpack + a.w_zu_f[L];
    for (int i = 0; i < L; ++i) {
        const int wi = a.width[i], wpad = pad16(wi);
        const bool last = i == L - 1;
        float *zout = lds + a.zb_off[i];
        const int ldo = a.zb_ld[i];
        const int NT = wpad / 16, KBy = kblocks_tile(n);
        const float *Wy = a.wpack + a.w_yu_f[i];
        for (int nt = wave; nt < NT; nt += TILE_STEP) {
            const int nt1 = TILE_PAIR && nt + NWAVE < NT ? nt + NWAVE : -1;
            f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            // epilogue operands (x-only context) are requested before the MFMA loops so that their
            // HBM/L2 latency is hidden behind them
            float czu[2][4], cgt[2][4], wz[2];
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tile = h == 0 ? nt : nt1;
                const int col = tile * 16 + r16;
                wz[h] = last && tile >= 0 && col < wi ? wzL[col] : 0.f;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q + r;
                    const bool ok = tile >= 0 && row < rows && col < wi;
                    const float *c = ctx + (size_t)(ok ? row : 0) * C;
                    czu[h][r] = ok ? c[a.zu_off[i] + col] : 0.f;
                    cgt[h][r] = ok ? c[a.gate_off[i + 1] + col] : 0.f;
                }
            }
            gemm_tiles(lds + a.aop_off[i], ldY, Wy, KBy, NT, nt, nt1, acc[0], acc[1]);
            if (i > 0)
                gemm_tiles(lds + a.zb_off[i - 1], a.zb_ld[i - 1], a.wpack + a.w_zu_f[i],
                           kblocks_tile(a.width[i - 1]), NT, nt, nt1, acc[0], acc[1]);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int tile = h == 0 ? nt : nt1;
                if (tile < 0) continue;
                const int col = tile * 16 + r16;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * q + r;
                    float v = 0.f, d = 0.f;
                    if (row < rows && col < wi) {
                        const float z = act_fn(acc[h][r] + czu[h][r], a.alpha);
                        v = z * cgt[h][r];                    // operand of the next layer: z_i * gate_{i+1}
                        // last hidden layer: delta_{L-1} = gate_L * wzu_L * act'(pre); sign(pre) = sign(z * gate), gate > 0
                        const float gw = cgt[h][r] * wz[h];
                        d = gw * (v > 0.f ? 1.f : a.alpha);
                    }
                    zout[row * ldo + col] = v;
                    if (last) dl[row * ldo + col] = d;
                }
            }
        }
        lap(2 + 3 * i);
        __syncthreads();
        lap(3 + 3 * i);
    }

    // ---------------- backward (the energy of the final scalar layer rides along in its last phase) ---------
    for (int i = L - 1; i >= 0; --i) {
        const int wi = a.width[i];
        const bool first = i == L - 1;
        const float *delta = first ? dl : lds + a.zb_off[i];
        const int ldd = a.zb_ld[i], KB = kblocks_tile(wi);
        const int NTy = npad / 16;
        {   // dE/dy (+)= yu_i * (delta_i Wyu_i^T), starting from yu_L * wyu_L
            const float *Wt = a.wpack + a.w_yu_b[i];
            // (with a delta product behind it in the same phase, i > 0, the few dE/dy tiles go to the waves counted from the
            //  top: those have the fewest delta tiles -- Bibsonomy step 1: ten dE/dy tiles + 38 delta tiles = three per wave
            //  instead of four on waves 0..5 and two on waves 10..15)
            const int wy = i > 0 ? NWAVE - 1 - wave : wave;
            for (int nt = wy; nt < NTy; nt += TILE_STEP) {
                const int nt1 = TILE_PAIR && nt + NWAVE < NTy ? nt + NWAVE : -1;
                f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                float cyu[2][4], cyL[2][4], wy[2];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tile = h == 0 ? nt : nt1;
                    const int col = tile * 16 + r16;
                    wy[h] = first && tile >= 0 && col < n ? wyL[col] : 0.f;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * q + r;
                        const bool ok = tile >= 0 && row < rows && col < n;
                        cyu[h][r] = ok ? ctx[(size_t)row * C + a.yu_off[i] + col] : 0.f;
                        cyL[h][r] = ok && first ? ctx[(size_t)row * C + a.yu_off[L] + col] : 0.f;
                    }
                }
                gemm_tiles(delta, ldd, Wt, KB, NTy, nt, nt1, acc[0], acc[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tile = h == 0 ? nt : nt1;
                    if (tile < 0) continue;
                    const int col = tile * 16 + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * q + r;
                        if (row < rows && col < n) {
                            const float g_in = first ? cyL[h][r] * wy[h] : gbuf[row * ldY + col];
                            gbuf[row * ldY + col] = __builtin_fmaf(cyu[h][r], acc[h][r], g_in);
                        }
                    }
                }
            }
        }
        lap(i == 0 ? 11 : 8);
        if (i > 0) {   // delta_{i-1} = gate_i * (delta_i Wzu_i^T) * act'(pre_{i-1})
            const int wp = a.width[i - 1];
            float *zprev = lds + a.zb_off[i - 1];
            const int ldp = a.zb_ld[i - 1];
            const float *Wt = a.wpack + a.w_zu_b[i];
            const int NTp = pad16(wp) / 16;
            for (int nt = wave; nt < NTp; nt += TILE_STEP) {
                const int nt1 = TILE_PAIR && nt + NWAVE < NTp ? nt + NWAVE : -1;
                f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                float cga[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tile = h == 0 ? nt : nt1;
                    const int col = tile * 16 + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * q + r;
                        const bool ok = tile >= 0 && row < rows && col < wp;
                        cga[h][r] = ok ? ctx[(size_t)row * C + a.gate_off[i] + col] : 0.f;
                    }
                }
                gemm_tiles(delta, ldd, Wt, KB, NTp, nt, nt1, acc[0], acc[1]);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int tile = h == 0 ? nt : nt1;
                    if (tile < 0) continue;
                    const int col = tile * 16 + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * q + r;
                        float d = 0.f;
                        if (row < rows && col < wp) {
                            const float gate = cga[h][r];
                            const float ga = gate * acc[h][r];
                            d = ga * (zprev[row * ldp + col] > 0.f ? 1.f : a.alpha);
                        }
                        zprev[row * ldp + col] = d;
                    }
                }
            }
            lap(9);
        } else {
            // E = z_{L-1} . wzu_L + (y * yu_L) . wyu_L + zu_L, one wave per row: on the waves that have no dE/dy tile
            // in this phase where there are any (n < 256), after the tile otherwise
            const float *zl = lds + a.zb_off[L - 1];
            const int ldz = a.zb_ld[L - 1], wl = a.width[L - 1];
            const int busy = NTy < NWAVE ? NTy : NWAVE, idle = NWAVE - busy;
            const int r0 = idle > 0 ? wave - busy : wave, rstep = idle > 0 ? idle : NWAVE;
            if (r0 >= 0)
                for (int r = r0; r < rows; r += rstep) {
                    const float *c = ctx + (size_t)r * C;
                    float part = 0.f;
                    for (int j = lane; j < wl; j += 64) part = __builtin_fmaf(zl[r * ldz + j], wzL[j], part);
                    for (int j = lane; j < n; j += 64) {
                        const float yy = ybuf[r * ldY + j] * c[a.yu_off[L] + j];
                        part = __builtin_fmaf(yy, wyL[j], part);
                    }
                    const float e = wave_sum_f(part) + c[a.zu_off[L]];
                    if (lane == 0) a.f[s0 + r] = e;
                }
            lap(7);
        }
        __syncthreads();
        lap(i == 0 ? 12 : 10);
    }

    const float gscale = a.action_box ? 2.f : 1.f;    // RL/src/icnn.py:152  grad *= 2
    if (wave < rows)
        for (int j = lane; j < n; j += 64) a.g[(size_t)(s0 + wave) * n + j] = gscale * gbuf[wave * ldY + j];
    lap(13);
}


inline int fill_args(const icnn_be_fc_model &m, FcArgs &a, int &lds_bytes) {
    const int L = m.n_layers - 1;
    if (L < 1 || m.n_layers > ICNN_BE_MAX_LAYERS || m.width[L] != 1 || m.n < 1) return ICNN_BE_EINVAL;
    a.n = m.n;
    a.L = L;
    a.alpha = m.alpha;
    a.action_box = m.action_box;
    a.tile_rows = TM;
    int o = 0, lo = 0;
    for (int i = 0; i <= L; ++i) {
        if (m.width[i] < 1) return ICNN_BE_EINVAL;
        a.width[i] = m.width[i];
        a.yu_off[i] = o; o += m.n;
        a.zu_off[i] = o; o += m.width[i];
        a.gate_off[i] = -1;
        if (i > 0) { a.gate_off[i] = o; o += m.width[i - 1]; }
    }
    if (o != m.ctx_width) return ICNN_BE_EINVAL;
    a.ctx_width = o;
    const PackOffsets po = pack_offsets(m);
    for (int i = 0; i <= L; ++i) {
        a.w_yu_f[i] = po.yu_f[i]; a.w_yu_b[i] = po.yu_b[i];
        a.w_zu_f[i] = po.zu_f[i]; a.w_zu_b[i] = po.zu_b[i];
    }
    a.ldY = lds_pitch(m.n);
    a.ybuf_off = lo; lo += TM * a.ldY;
    for (int i = 0; i < L; ++i) { a.aop_off[i] = lo; lo += TM * a.ldY; }
    a.gbuf_off = lo; lo += TM * a.ldY;
    for (int i = 0; i < L; ++i) {
        a.zb_ld[i] = lds_pitch(m.width[i]);
        a.zb_off[i] = lo; lo += TM * a.zb_ld[i];
    }
    a.dl_off = lo; lo += TM * a.zb_ld[L - 1];
    a.lds_floats = lo;
    lds_bytes = lo * 4;
    if (lds_bytes > 160 * 1024) return ICNN_BE_ELIMIT;
    a.wpack = m.wpack;
    return 0;
}

}  // namespace
}  // namespace icnn_be
