// Small-batch evaluation of the FC-PICNN (energy + y-gradient): one workgroup holds 1-4 samples and every
// (sample, 64 columns) unit is a wave running the k-ordered fma chains of its columns on the VALU -- the very
// order v_mfma_f32_16x16x4_f32 applies (kk = 16 kb + 4 q + s, s outer; oracle/picnn_chain.c), so this path and the
// 16-row MFMA tile of be_picnn_fc_dev.h agree bit for bit.  A 16-row tile costs the same whether 1 or 16 rows are
// real (54 us for the Bibsonomy network, 25 us for the RL agent's); with one sample per CU the cost is the weight
// stream through that CU's 64 B/clk L1 fill port.  Packed weights are read as 16-byte fragments straight from L2;
// context row, activations and operands of a sample live in LDS.  Used by fc_fg_rows_kernel (be_picnn_fc.hip:
// batches of at most one sample per CU) and adam_rows_kernel (be_adam.hip).
#pragma once
#include "be_picnn_fc_dev.h"

namespace icnn_be {

namespace {

constexpr int ROWS_MAX = 4;
constexpr int GV_AHEAD = 2;     // k-blocks of weight fragments in flight per lane (16 VGPRs each), <= PF
constexpr int RWAVES = 8, RTHREADS = RWAVES * 64;   // 8 waves: 256 VGPRs per lane for the fragment ring

// LDS floats per sample: yop_0 .. yop_{L-1} (y * yu_i, the GEMV operands) | ysc (y * yu_L) | g (dE/dy) |
// g0 (yu_L * wyu_L) | z_0 .. z_{L-1} | dl (delta_{L-1}) | gw (gate_L * wzu_L) | ctx row; then, shared by the samples
// of the workgroup: the scalar layer's weight vectors, the energies, caller scratch.
struct RowsLayout {
    int row_floats;
    int yop_off[ICNN_BE_MAX_LAYERS], ysc_off, g_off, g0_off, z_off[ICNN_BE_MAX_LAYERS], dl_off, gw_off, ctx_off;
    int wz_off, wy_off, f_off, misc_off;
};

// `rows` samples per workgroup; returns the LDS bytes up to misc_off (the caller appends its own scratch there)
inline int rows_layout(const icnn_be_fc_model &m, int rows, RowsLayout &r) {
    const int npad = pad16(m.n), L = m.n_layers - 1, ypad = kblocks(m.n) * 16;
    int o = 0;
    for (int i = 0; i < L; ++i) { r.yop_off[i] = o; o += ypad; }      // chained operands: zero up to their padded k-blocks
    r.ysc_off = o; o += npad;
    r.g_off = o; o += npad;
    r.g0_off = o; o += npad;
    for (int i = 0; i < L; ++i) { r.z_off[i] = o; o += kblocks(m.width[i]) * 16; }
    r.dl_off = o; o += kblocks(m.width[L - 1]) * 16;
    r.gw_off = o; o += pad16(m.width[L - 1]);
    r.ctx_off = o; o += (m.ctx_width + 3) & ~3;
    r.row_floats = o;
    o *= rows;
    r.wz_off = o; o += pad16(m.width[L - 1]);
    r.wy_off = o; o += npad;
    r.f_off = o; o += ROWS_MAX;
    r.misc_off = o;
    return o * 4;
}

// acc[s] += A[s][0 .. 16 KB) . W[., col] in MFMA order for S samples that share every weight fragment; A[s] in LDS,
// Wp a packed operand.  KB = kblocks(K), a multiple of PF: the pack carries zero fragments and the LDS operands zero
// columns up to there, so the loop body is straight-line code -- a PF-slot fragment ring filled GV_AHEAD k-blocks
// ahead, the A fragments of the next k-block read before the fma chains of the current one.
template <int S>
__device__ __forceinline__ void gemv_chain(float (&acc)[S], const float *const (&A)[S], const float *Wp, int KB, int NT,
                                           int col) {
#pragma clang fp contract(off)
    const f4 *bp = reinterpret_cast<const f4 *>(Wp) + (size_t)(col >> 4) * 64 + (col & 15);
    const size_t ks = (size_t)NT * 64;
    f4 w[PF][4], an[S][4];
#pragma unroll
    for (int d = 0; d < GV_AHEAD; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            w[d][q] = bp[(size_t)d * ks + q * 16];
            __builtin_amdgcn_sched_barrier(0);      // issued in the order of use (see gemm_loop, be_picnn_fc_dev.h)
        }
#pragma unroll
    for (int s = 0; s < S; ++s)
#pragma unroll
        for (int q = 0; q < 4; ++q) an[s][q] = *reinterpret_cast<const f4 *>(A[s] + 4 * q);
    for (int kb0 = 0; kb0 < KB; kb0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kb = kb0 + d;
            f4 av[S][4];
            const int ka = kb + 1 < KB ? kb + 1 : kb;                 // (clamped re-read at the tail)
#pragma unroll
            for (int s = 0; s < S; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    av[s][q] = an[s][q];
                    an[s][q] = *reinterpret_cast<const f4 *>(A[s] + ka * 16 + 4 * q);
                }
            const int nk = kb + GV_AHEAD < KB ? kb + GV_AHEAD : kb;
            f4 x[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) x[q] = w[d][q];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[(d + GV_AHEAD) % PF][q] = bp[(size_t)nk * ks + q * 16];
            __builtin_amdgcn_sched_barrier(0);      // (requests stay GV_AHEAD k-blocks in front of their fma chains)
#pragma unroll
            for (int s = 0; s < S; ++s) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[s][q].x, x[q].x, acc[s]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[s][q].y, x[q].y, acc[s]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[s][q].z, x[q].z, acc[s]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[s][q].w, x[q].w, acc[s]);
            }
            __builtin_amdgcn_sched_barrier(0);
        }
    }
}

// The same chain for a narrow operand (K <= 64, the action itself): only the real k-blocks, all fragments requested
// at once.  Skipping the pack's zero k-blocks changes nothing: fma(0, 0, acc) == acc (acc is never -0: it starts
// at +0 and x + (-x) rounds to +0).
template <int S>
__device__ __forceinline__ void gemv_short(float (&acc)[S], const float *const (&A)[S], const float *Wp, int KBr, int NT,
                                           int col) {
#pragma clang fp contract(off)
    const f4 *bp = reinterpret_cast<const f4 *>(Wp) + (size_t)(col >> 4) * 64 + (col & 15);
    const size_t ks = (size_t)NT * 64;
#pragma unroll
    for (int d = 0; d < 4; ++d)
        if (d < KBr) {
            f4 w[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) w[q] = bp[(size_t)d * ks + q * 16];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                f4 av[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) av[q] = *reinterpret_cast<const f4 *>(A[s] + d * 16 + 4 * q);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[q].x, w[q].x, acc[s]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[q].y, w[q].y, acc[s]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[q].z, w[q].z, acc[s]);
#pragma unroll
                for (int q = 0; q < 4; ++q) acc[s] = __builtin_fmaf(av[q].w, w[q].w, acc[s]);
            }
        }
}

// The x-only context rows of the workgroup's samples computed IN the kernel from their observations (the RL agent's
// act(): observation -> action in one launch; RL/src/icnn.py:339-385 without BatchNorm, the agent's default).  Stage i
// multiplies prev_i (x, then the u-path activation u_{i-1}) with the column-wise concatenation
// [ u{i}/W | z{i}_yu_u/W | z{i}_u/W | z{i}_zu_u/W ] of icnn_be_fc_ctx: a thread per output column runs the chain
// acc = fma(prev[k], W[k][col], acc) for k = 0.., adds the bias, applies ReLU to hidden u layers and gates, and routes
// the value to the next stage's input (the still unused operand region of the row serves as scratch) or to its slot of
// the context row.  oracle/picnn_chain.c (picnn_context_rows_chain) reproduces the order.  Ends with a barrier.
struct CtxRowsArgs {
    const float *obs;                               // [batch][n_features], nullptr = context rows come from memory
    int n_features;
    const float *w_stage[ICNN_BE_MAX_LAYERS], *b_stage[ICNN_BE_MAX_LAYERS];
};
template <typename ArgsT, typename LayT>
__device__ __forceinline__ void rows_context_from_obs(const CtxRowsArgs &cr, const ArgsT &fa, const LayT &r, float *lds,
                                                      int s_base, int batch, int tid) {
#pragma clang fp contract(off)
    const int L = fa.L, n = fa.n, RF = r.row_floats;
    int wmax = cr.n_features;
    for (int i = 0; i < L; ++i) wmax = fa.width[i] > wmax ? fa.width[i] : wmax;
    // scratch inside the row's operand region (zeroed later by rows_setup): two vectors of wmax floats
    for (int s = 0; s < batch; ++s)
        for (int j = tid; j < cr.n_features; j += RTHREADS) lds[s * RF + j] = cr.obs[(size_t)(s_base + s) * cr.n_features + j];
    __syncthreads();
    int K = cr.n_features, cur = 0;
    for (int i = 0; i <= L; ++i) {
        const int wu = i < L ? fa.width[i] : 0, wg = i > 0 ? fa.width[i - 1] : 0;
        const int cols = wu + n + fa.width[i] + wg, ld = (cols + 3) & ~3;
        const float *W = cr.w_stage[i], *b = cr.b_stage[i];
        // a thread per output column, all samples of the workgroup off every weight load, eight loads in flight
        for (int col = tid; col < cols; col += RTHREADS) {
            float acc[ROWS_MAX];
#pragma unroll
            for (int s = 0; s < ROWS_MAX; ++s) acc[s] = 0.f;
            const float *wc = W + col;
            int k = 0;
            // (one workgroup streams the whole stage matrix: what matters is the number of bytes in flight -- 32 loads
            //  per thread; the fma chain per column stays k-ascending)
            for (; k + 32 <= K; k += 32) {
                float w32[32];
#pragma unroll
                for (int d = 0; d < 32; ++d) w32[d] = wc[(size_t)(k + d) * ld];
#pragma unroll
                for (int d = 0; d < 32; ++d)
#pragma unroll
                    for (int s = 0; s < ROWS_MAX; ++s)
                        if (s < batch) acc[s] = __builtin_fmaf(lds[s * RF + cur * wmax + k + d], w32[d], acc[s]);
            }
            for (; k + 8 <= K; k += 8) {
                float w8[8];
#pragma unroll
                for (int d = 0; d < 8; ++d) w8[d] = wc[(size_t)(k + d) * ld];
#pragma unroll
                for (int d = 0; d < 8; ++d)
#pragma unroll
                    for (int s = 0; s < ROWS_MAX; ++s)
                        if (s < batch) acc[s] = __builtin_fmaf(lds[s * RF + cur * wmax + k + d], w8[d], acc[s]);
            }
            for (; k < K; ++k) {
                const float w1 = wc[(size_t)k * ld];
#pragma unroll
                for (int s = 0; s < ROWS_MAX; ++s)
                    if (s < batch) acc[s] = __builtin_fmaf(lds[s * RF + cur * wmax + k], w1, acc[s]);
            }
            const float bias = b[col];
#pragma unroll
            for (int s = 0; s < ROWS_MAX; ++s) {
                if (s >= batch) continue;
                float *row = lds + s * RF;
                float *next = row + (1 - cur) * wmax;
                const float v = acc[s] + bias;
                int c = col;
                if (c < wu) { next[c] = i < L - 1 ? fmaxf(v, 0.f) : v; continue; }       // u_i (hidden layers ReLU'd)
                c -= wu;
                if (c < n) { row[r.ctx_off + fa.yu_off[i] + c] = v; continue; }
                c -= n;
                if (c < fa.width[i]) { row[r.ctx_off + fa.zu_off[i] + c] = v; continue; }
                c -= fa.width[i];
                row[r.ctx_off + fa.gate_off[i] + c] = fmaxf(v, 0.f);                     // gate_i = relu(.)
            }
        }
        __syncthreads();
        K = wu;
        cur = 1 - cur;
    }
}

// Once per workgroup: operands zero (y = 0 and every pad column), context rows of samples s_base .. s_base+batch-1
// (copied from memory unless `ctx_in_lds`: rows_context_from_obs has produced them) and the iteration-invariant
// products into LDS.  Ends with a workgroup barrier.
template <typename ArgsT, typename LayT>      // FcArgs / RowsLayout by value, or references into the kernel-argument segment
__device__ __forceinline__ void rows_setup(const ArgsT &fa, const LayT &r, float *lds, int s_base, int batch, int tid,
                                           bool ctx_in_lds = false) {
#pragma clang fp contract(off)
    const int n = fa.n, L = fa.L, C = fa.ctx_width, npad = pad16(n), RF = r.row_floats;
    const int wl = fa.width[L - 1], wlp = pad16(wl);
    float *wzs = lds + r.wz_off, *wys = lds + r.wy_off;               // scalar layer: 'z{L}_zu_proj/W', 'z{L}_yu/W'
    for (int s = 0; s < batch; ++s) {
        float *row = lds + s * RF;
        for (int j = tid; j < r.ctx_off; j += RTHREADS) row[j] = 0.f;
        if (!ctx_in_lds)
            for (int j = tid; j < C; j += RTHREADS) row[r.ctx_off + j] = fa.ctx[(size_t)(s_base + s) * C + j];
    }
    for (int j = tid; j < wlp; j += RTHREADS) wzs[j] = j < wl ? fa.wpack[fa.w_zu_f[L] + j] : 0.f;
    for (int j = tid; j < npad; j += RTHREADS) wys[j] = j < n ? fa.wpack[fa.w_yu_f[L] + j] : 0.f;
    __syncthreads();
    for (int s = 0; s < batch; ++s) {
        float *row = lds + s * RF;
        for (int j = tid; j < wl; j += RTHREADS) row[r.gw_off + j] = row[r.ctx_off + fa.gate_off[L] + j] * wzs[j];
        for (int j = tid; j < n; j += RTHREADS) row[r.g0_off + j] = row[r.ctx_off + fa.yu_off[L] + j] * wys[j];
    }
    __syncthreads();
}

// Network input of one sample (called by one wave, lanes over j): the operands y * yu_i of every layer
template <typename ArgsT, typename LayT>
__device__ __forceinline__ void rows_set_input(const ArgsT &fa, const LayT &r, float *row, int j, float y32) {
#pragma clang fp contract(off)
    for (int i = 0; i < fa.L; ++i) row[r.yop_off[i] + j] = y32 * row[r.ctx_off + fa.yu_off[i] + j];
    row[r.ysc_off + j] = y32 * row[r.ctx_off + fa.yu_off[fa.L] + j];
}

// E (-> lds[f_off + s]) and dE/dy (-> row[g_off + j]) of the `batch` samples of the workgroup; one barrier per layer
// and direction (the final scalar layer rides along with the first backward phase on an idle wave, delta_{L-1} is
// written by the forward epilogue).  Every value is formed by the same float32 operations as in fc_fg_tile.  With
// more than one sample in the workgroup a wave handles 64 columns of a PAIR of samples (S = 2): one weight fragment
// feeds both fma chains, so the weight stream -- what bounds this path -- is halved.
template <int S, typename ArgsT, typename LayT, typename Lap>
__device__ __forceinline__ void rows_eval_impl(const ArgsT &fa, const LayT &r, float *lds, int batch, int tid, Lap lap) {
#pragma clang fp contract(off)
    const int wave = tid >> 6, lane = tid & 63;
    const int n = fa.n, L = fa.L, npad = pad16(n), RF = r.row_floats, KBy = npad / 16;
    const int wl = fa.width[L - 1];
    const int groups = (batch + S - 1) / S;
    const float *wzs = lds + r.wz_off, *wys = lds + r.wy_off;
    float *fbuf = lds + r.f_off;
    for (int i = 0; i < L; ++i) {
        const int wi = fa.width[i], wpad = pad16(wi), NT = wpad / 16;
        const int cw = (wpad + 63) / 64;
        for (int unit = wave; unit < groups * cw; unit += RWAVES) {
            int gi = 0, cg = unit;
            while (cg >= cw) { cg -= cw; ++gi; }
            const int col = cg * 64 + lane;
            float *row[S];
            bool has[S];
            const float *A[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                has[s] = gi * S + s < batch;
                row[s] = lds + (has[s] ? gi * S + s : gi * S) * RF;
                A[s] = row[s] + r.yop_off[i];
            }
            if (col < wpad) {
                float acc[S];
#pragma unroll
                for (int s = 0; s < S; ++s) acc[s] = 0.f;
                if (KBy <= 4) gemv_short<S>(acc, A, fa.wpack + fa.w_yu_f[i], KBy, NT, col);
                else gemv_chain<S>(acc, A, fa.wpack + fa.w_yu_f[i], kblocks(n), NT, col);
                if (i > 0) {
#pragma unroll
                    for (int s = 0; s < S; ++s) A[s] = row[s] + r.z_off[i - 1];
                    gemv_chain<S>(acc, A, fa.wpack + fa.w_zu_f[i], kblocks(fa.width[i - 1]), NT, col);
                }
#pragma unroll
                for (int s = 0; s < S; ++s)
                    if (has[s]) {
                        float v = 0.f;
                        if (col < wi) {
                            const float z = act_fn(acc[s] + row[s][r.ctx_off + fa.zu_off[i] + col], fa.alpha);
                            v = z * row[s][r.ctx_off + fa.gate_off[i + 1] + col];
                        }
                        row[s][r.z_off[i] + col] = v;
                        if (i == L - 1)        // delta_{L-1} = gate_L * wzu_L * act'(pre): sign(pre) = sign(z * gate), gate > 0
                            row[s][r.dl_off + col] = col < wi ? row[s][r.gw_off + col] * (v > 0.f ? 1.f : fa.alpha) : 0.f;
                    }
            }
        }
        lap(2 * i);
        __syncthreads();
        lap(2 * i + 1);
    }
    for (int i = L - 1; i >= 0; --i) {
        const int wi = fa.width[i], KB = kblocks(wi);
        const int cwn = (npad + 63) / 64;
        const int wp = i > 0 ? fa.width[i - 1] : 0, wppad = pad16(wp), cwp = (wppad + 63) / 64;
        const int per_group = cwn + cwp;
        const int n_units = groups * per_group + (i == L - 1 ? batch : 0);   // + the energy of every sample (final scalar layer)
        for (int unit = wave; unit < n_units; unit += RWAVES) {
            if (unit >= groups * per_group) {                     // E = z_{L-1} . wzu_L + (y * yu_L) . wyu_L + zu_L
                const int s = unit - groups * per_group;
                const float *rw = lds + s * RF;
                const float *zl = rw + r.z_off[L - 1];
                float psum = 0.f;
                for (int j = lane; j < wl; j += 64) psum = __builtin_fmaf(zl[j], wzs[j], psum);
                for (int j = lane; j < n; j += 64) psum = __builtin_fmaf(rw[r.ysc_off + j], wys[j], psum);
                const float e = wave_sum_f(psum) + rw[r.ctx_off + fa.zu_off[L]];
                if (lane == 0) fbuf[s] = e;
                continue;
            }
            int gi = 0, part = unit;
            while (part >= per_group) { part -= per_group; ++gi; }
            float *row[S];
            bool has[S];
            const float *delta[S];
#pragma unroll
            for (int s = 0; s < S; ++s) {
                has[s] = gi * S + s < batch;
                row[s] = lds + (has[s] ? gi * S + s : gi * S) * RF;
                delta[s] = row[s] + (i == L - 1 ? r.dl_off : r.z_off[i]);
            }
            float acc[S];
#pragma unroll
            for (int s = 0; s < S; ++s) acc[s] = 0.f;
            if (part < cwn) {                                     // dE/dy += yu_i * (delta_i Wyu_i^T)
                const int col = part * 64 + lane;
                if (col < npad) {
                    gemv_chain<S>(acc, delta, fa.wpack + fa.w_yu_b[i], KB, npad / 16, col);
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        if (has[s] && col < n) {
                            const float g_in = row[s][(i == L - 1 ? r.g0_off : r.g_off) + col];
                            row[s][r.g_off + col] = __builtin_fmaf(row[s][r.ctx_off + fa.yu_off[i] + col], acc[s], g_in);
                        }
                }
            } else {                                              // delta_{i-1} = gate_i * (delta_i Wzu_i^T) * act'
                const int col = (part - cwn) * 64 + lane;
                if (col < wppad) {
                    gemv_chain<S>(acc, delta, fa.wpack + fa.w_zu_b[i], KB, wppad / 16, col);
#pragma unroll
                    for (int s = 0; s < S; ++s)
                        if (has[s]) {
                            float d = 0.f;
                            if (col < wp) {
                                const float ga = row[s][r.ctx_off + fa.gate_off[i] + col] * acc[s];
                                d = ga * (row[s][r.z_off[i - 1] + col] > 0.f ? 1.f : fa.alpha);
                            }
                            row[s][r.z_off[i - 1] + col] = d;
                        }
                }
            }
        }
        lap(i == 0 ? 11 : 8);
        __syncthreads();
        lap(i == 0 ? 12 : 10);
    }
}
template <typename ArgsT, typename LayT, typename Lap>
__device__ __forceinline__ void rows_eval(const ArgsT &fa, const LayT &r, float *lds, int batch, int tid, Lap lap) {
    if (batch == 1) rows_eval_impl<1>(fa, r, lds, batch, tid, lap);
    else rows_eval_impl<2>(fa, r, lds, batch, tid, lap);
}

}  // namespace
}  // namespace icnn_be
