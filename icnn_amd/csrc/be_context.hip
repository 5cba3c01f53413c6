// x-only context producer of the fully-connected PICNN on the device:
//   u-path   u_i = [BN](relu(u_{i-1} U_i + b_i)), last layer linear        multi-label-cls/icnn_ebundle.py:339-347
//   heads    yu_i = prev_i Wyu_u_i + b,  zu_i = prev_i Wu_i + b,  gate_i = relu(prev_i Wzu_u_i + b)      :354-374
//            (prev_0 = x, prev_i = u_{i-1}); RL: RL/src/icnn.py:339-385
// and the clamp ops of the convex weights, makeCvx / proj (:143-144, :204, :244-245).
//
// Everything that reads the same input is ONE GEMM: stage i multiplies prev_i [B][K_i] with the column-wise
// concatenation  [ U_i | Wyu_u_i | Wu_i | Wzu_u_i ]  (host-side concatenation, icnn_amd/picnn.py) and the epilogue routes
// column ranges to their destinations -- the next stage's input (with ReLU) or the slots of the context row the
// solve kernels read (include/icnn_be.h: yu_i | zu_i | gate_i per layer).  f32 MFMA (v_mfma_f32_16x16x4_f32), 64 x 64
// output tile per workgroup of four waves, both operands staged through LDS (A as is, W transposed so that a lane's
// four k-values are one ds_read_b128), register prefetch of the next k-block.  BatchNorm uses the statistics of the
// batch it is given, like the reference (tflearn.is_training(True), :209,:259): one workgroup per 64 columns, three
// passes over its column block (mean, variance of the centred values, normalise in place).
#include <hip/hip_runtime.h>

#include <initializer_list>

#include "be_common.h"
#include "be_kernels.h"
#include "icnn_be.h"

namespace icnn_be {

namespace {

constexpr int BM = 64, BN = 64, BK = 16, GT = 256, PITCH = BK + 4;

struct CtxSeg {          // output columns [c0, c1) of a stage -> dst[row * ld + off + (col - c0)]
    int c0, c1, ld, off, relu;
    int P;               // > 0: rows are (sample, output position) pairs, P positions per sample, and the destination is the
                         // sample's context row: dst[(row / P) * ld + off + (row % P) * (c1 - c0) + (col - c0)]
    float *dst;
};
// conv = 1: A is an NHWC image batch [B][IH][IW][IC] and the GEMM row (b, oy, ox) gathers its K = KS*KS*IC operand
// (ky, kx, ci) from in[b][oy*ST - PD + ky][ox*ST - PD + kx][ci], zero outside the image -- tflearn's [k][k][Cin][F] weight
// read as [K][F] is the B operand as it is stored.
struct CtxGemmArgs {
    const float *A, *W, *bias;
    int lda, M, K, ldw, N, nseg, a_vec;
    int conv, IH, IW, IC, KS, ST, PD, OW, P;
    CtxSeg seg[4];
};

__global__ __launch_bounds__(GT) void ctx_gemm_kernel(CtxGemmArgs a) {
    __shared__ __attribute__((aligned(16))) float As[2][BM][PITCH];
    __shared__ __attribute__((aligned(16))) float Bt[2][BN][PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const int m0 = blockIdx.x * BM, n0 = blockIdx.y * BN;   // row tiles in grid.x (limit 2^31-1): batch * positions of a conv stage exceeds grid.y's 65535 tiles from batch 2048 on
    // global -> register staging: A tile 64 x 16 (thread: row tid/4, four k), W tile 16 x 64 (thread: k tid/16, four n)
    const int arow = tid >> 2, akq = (tid & 3) * 4, wk = tid >> 4, wn4 = (tid & 15) * 4;
    const bool arow_ok = m0 + arow < a.M;
    const float *ap = a.A + (size_t)(arow_ok && !a.conv ? m0 + arow : 0) * a.lda;
    int iy0 = 0, ix0 = 0;
    if (a.conv) {
        const int m = arow_ok ? m0 + arow : 0, b = m / a.P, pos = m - b * a.P, oy = pos / a.OW, ox = pos - oy * a.OW;
        iy0 = oy * a.ST - a.PD; ix0 = ox * a.ST - a.PD;
        ap = a.A + (size_t)b * a.IH * a.IW * a.IC;
    }
    auto load_a = [&](int k0) -> f4 {
        f4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = k0 + akq;
        if (a.conv) {
            if (!arow_ok || k >= a.K) return v;
            if (a.a_vec) {           // IC % 4 == 0: the four k are four channels of one tap
                const int tap = k / a.IC, ci = k - tap * a.IC, ky = tap / a.KS, kx = tap - ky * a.KS;
                const int iy = iy0 + ky, ix = ix0 + kx;
                if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW)
                    v = *reinterpret_cast<const f4 *>(ap + ((size_t)iy * a.IW + ix) * a.IC + ci);
                return v;
            }
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const int kk = k + j;
                if (kk >= a.K) break;
                const int tap = kk / a.IC, ci = kk - tap * a.IC, ky = tap / a.KS, kx = tap - ky * a.KS;
                const int iy = iy0 + ky, ix = ix0 + kx;
                if (iy >= 0 && iy < a.IH && ix >= 0 && ix < a.IW) v[j] = ap[((size_t)iy * a.IW + ix) * a.IC + ci];
            }
            return v;
        }
        if (arow_ok) {
            if (a.a_vec && k + 3 < a.K) v = *reinterpret_cast<const f4 *>(ap + k);
            else {
                if (k < a.K) v.x = ap[k];
                if (k + 1 < a.K) v.y = ap[k + 1];
                if (k + 2 < a.K) v.z = ap[k + 2];
                if (k + 3 < a.K) v.w = ap[k + 3];
            }
        }
        return v;
    };
    auto load_w = [&](int k0) -> f4 {     // ldw is a multiple of 4 and the columns beyond N are zero (host contract)
        f4 v = {0.f, 0.f, 0.f, 0.f};
        const int k = k0 + wk, n = n0 + wn4;
        if (k < a.K && n < a.ldw) v = *reinterpret_cast<const f4 *>(a.W + (size_t)k * a.ldw + n);
        return v;
    };
    f4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t) acc[t] = f4{0.f, 0.f, 0.f, 0.f};
    f4 ra = load_a(0), rw = load_w(0);
    const int nkb = (a.K + BK - 1) / BK;
    for (int kb = 0; kb < nkb; ++kb) {
        const int buf = kb & 1;
        *reinterpret_cast<f4 *>(&As[buf][arow][akq]) = ra;
        Bt[buf][wn4 + 0][wk] = rw.x;
        Bt[buf][wn4 + 1][wk] = rw.y;
        Bt[buf][wn4 + 2][wk] = rw.z;
        Bt[buf][wn4 + 3][wk] = rw.w;
        __syncthreads();                  // (two buffers: the stores of k-block kb+1 cannot overtake the reads of kb-1)
        if (kb + 1 < nkb) { ra = load_a((kb + 1) * BK); rw = load_w((kb + 1) * BK); }
        const f4 af = *reinterpret_cast<const f4 *>(&As[buf][16 * wave + r16][4 * q]);
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const f4 bf = *reinterpret_cast<const f4 *>(&Bt[buf][16 * t + r16][4 * q]);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.x, bf.x, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.y, bf.y, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.z, bf.z, acc[t], 0, 0, 0);
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(af.w, bf.w, acc[t], 0, 0, 0);
        }
    }
    // epilogue: bias, optional ReLU, routed store.  acc[t][r] = C[m0 + 16 wave + 4 q + r][n0 + 16 t + r16]
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        const int col = n0 + 16 * t + r16;
        if (col >= a.N) continue;
        int s = 0;
        while (s + 1 < a.nseg && col >= a.seg[s].c1) ++s;
        const CtxSeg sg = a.seg[s];
        const float b = a.bias[col];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = m0 + 16 * wave + 4 * q + r;
            if (row >= a.M) continue;
            float v = acc[t][r] + b;
            if (sg.relu) v = fmaxf(v, 0.f);
            if (sg.P > 0) {
                const int smp = row / sg.P, pos = row - smp * sg.P;
                sg.dst[(size_t)smp * sg.ld + sg.off + (size_t)pos * (sg.c1 - sg.c0) + (col - sg.c0)] = v;
            } else {
                sg.dst[(size_t)row * sg.ld + sg.off + (col - sg.c0)] = v;
            }
        }
    }
}

// BatchNorm with batch statistics, in place on u[M][ld], columns [0, N): tflearn.batch_normalization in training
// mode = tf.nn.moments + tf.nn.batch_normalization (epsilon 1e-5), multi-label-cls/icnn_ebundle.py:345.
// One workgroup per 32 columns (ld is a multiple of 4: float4 accesses, 128 contiguous bytes per row), 128 row groups.
constexpr int BNT = 1024, BNC = 32, BNQ = BNC / 4, BNG = BNT / BNQ;
__global__ __launch_bounds__(BNT) void ctx_bn_kernel(float *u, int ld, int M, int N, const float *gamma, const float *beta,
                                                     float eps) {
    __shared__ f4 red[BNG][BNQ];
    __shared__ f4 stat[2][BNQ];
    const int cq = threadIdx.x % BNQ, g = threadIdx.x / BNQ, col = blockIdx.x * BNC + 4 * cq;
    const bool ok = col < N;                       // (columns N .. ld-1 of the last quad are padding: harmless)
    auto column_total = [&](f4 mine, f4 *out) {    // deterministic tree: per thread rows g, g+BNG, ..; then over groups
        red[g][cq] = mine;
        __syncthreads();
        if (g == 0) {
            f4 tot = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < BNG; ++i) tot += red[i][cq];
            *out = tot / (float)M;
        }
        __syncthreads();
    };
    f4 s = {0.f, 0.f, 0.f, 0.f};
    if (ok) for (int r = g; r < M; r += BNG) s += *reinterpret_cast<const f4 *>(u + (size_t)r * ld + col);
    column_total(s, &stat[0][cq]);
    const f4 mean = stat[0][cq];
    s = f4{0.f, 0.f, 0.f, 0.f};
    if (ok) for (int r = g; r < M; r += BNG) {
        const f4 d = *reinterpret_cast<const f4 *>(u + (size_t)r * ld + col) - mean;
        s += d * d;
    }
    column_total(s, &stat[1][cq]);
    if (!ok) return;
    const f4 var = stat[1][cq];
    f4 inv, ga, be;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        inv[i] = 1.f / sqrtf(var[i] + eps);
        ga[i] = col + i < N ? gamma[col + i] : 0.f;
        be[i] = col + i < N ? beta[col + i] : 0.f;
    }
    for (int r = g; r < M; r += BNG) {
        f4 *p = reinterpret_cast<f4 *>(u + (size_t)r * ld + col);
        *p = (*p - mean) * inv * ga + be;
    }
}

// Data-parallel ranks (SURVEY.md 8(e) caveat): the statistics are those of the GLOBAL batch, every rank holds a shard of the
// rows.  ctx_bn_sums_kernel writes this shard's per-column sum and sum of squares (float64, fixed order: thread rows
// g, g+BNG, .. then over the groups); the host all-reduces the 2 x N doubles (RCCL over xGMI; the ONE collective this
// stage needs) and ctx_bn_apply_kernel normalises the shard with mean = S1 / M_total, var = S2 / M_total - mean^2.
__global__ __launch_bounds__(BNT) void ctx_bn_sums_kernel(const float *u, int ld, int M, int N, double *stats) {
    __shared__ double red[BNG][BNC];
    const int cq = threadIdx.x % BNQ, g = threadIdx.x / BNQ, col = blockIdx.x * BNC + 4 * cq;
    double s1[4] = {0, 0, 0, 0}, s2[4] = {0, 0, 0, 0};
    if (col < N)
        for (int r = g; r < M; r += BNG) {
            const f4 v = *reinterpret_cast<const f4 *>(u + (size_t)r * ld + col);
#pragma unroll
            for (int i = 0; i < 4; ++i) { s1[i] += (double)v[i]; s2[i] += (double)v[i] * (double)v[i]; }
        }
    for (int which = 0; which < 2; ++which) {
#pragma unroll
        for (int i = 0; i < 4; ++i) red[g][4 * cq + i] = which ? s2[i] : s1[i];
        __syncthreads();
        if (threadIdx.x < BNC) {
            double tot = 0.0;
            for (int i = 0; i < BNG; ++i) tot += red[i][threadIdx.x];
            if (blockIdx.x * BNC + threadIdx.x < N) stats[(size_t)which * N + blockIdx.x * BNC + threadIdx.x] = tot;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(BNT) void ctx_bn_apply_kernel(float *u, int ld, int M, int N, const double *stats, double m_total,
                                                           const float *gamma, const float *beta, float eps) {
    const int cq = threadIdx.x % BNQ, g = threadIdx.x / BNQ, col = blockIdx.x * BNC + 4 * cq;
    if (col >= N) return;
    f4 mean, inv, ga, be;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const bool ok = col + i < N;
        const double m = ok ? stats[col + i] / m_total : 0.0;
        const double var = ok ? fmax(stats[(size_t)N + col + i] / m_total - m * m, 0.0) : 1.0;
        mean[i] = (float)m;
        inv[i] = 1.f / sqrtf((float)var + eps);
        ga[i] = ok ? gamma[col + i] : 0.f;
        be[i] = ok ? beta[col + i] : 0.f;
    }
    for (int r = g; r < M; r += BNG) {
        f4 *p = reinterpret_cast<f4 *>(u + (size_t)r * ld + col);
        *p = (*p - mean) * inv * ga + be;
    }
}

// The same BatchNorm for TALL matrices -- the conv u-maps, [batch * positions][32 or 64 channels]: a workgroup per column
// block would leave one or two workgroups with all the rows.  Three launches, each over BNB row blocks: pass 0 writes
// the per-block column sums, pass 1 the per-block sums of the squared deviations from the mean (which every workgroup
// forms from the pass-0 partials in the same fixed order), pass 2 normalises.  No atomics: the statistics are the same
// bits whatever the schedule.
constexpr int BNB = 128, TBT = 256;
__global__ __launch_bounds__(TBT) void ctx_bn_tall_kernel(float *u, int ld, int M, int N, float *part, const float *gamma,
                                                          const float *beta, float eps, int pass) {
    __shared__ f4 red[TBT];
    const int nq = N / 4, cq = threadIdx.x % nq, g = threadIdx.x / nq, ng = TBT / nq;    // N a multiple of 4, N <= 256
    const bool live = g < ng;
    const int rows_per = (M + BNB - 1) / BNB, r0 = blockIdx.x * rows_per, r1 = min(M, r0 + rows_per);
    float *psum = part, *psq = part + (size_t)BNB * N;
    auto total = [&](const float *p) {              // column totals from the per-block partials, fixed order
        f4 t = {0.f, 0.f, 0.f, 0.f};
        for (int b = 0; b < BNB; ++b) t += *reinterpret_cast<const f4 *>(p + (size_t)b * N + 4 * cq);
        return t / (float)M;
    };
    auto block_sum = [&](f4 mine, float *dst) {     // over the row groups of this workgroup, fixed order
        red[threadIdx.x] = mine;
        __syncthreads();
        if (live && g == 0) {
            f4 t = {0.f, 0.f, 0.f, 0.f};
            for (int i = 0; i < ng; ++i) t += red[i * nq + cq];
            *reinterpret_cast<f4 *>(dst + (size_t)blockIdx.x * N + 4 * cq) = t;
        }
    };
    f4 mean = {0.f, 0.f, 0.f, 0.f};
    if (pass > 0 && live) mean = total(psum);
    if (pass == 0) {
        f4 s = {0.f, 0.f, 0.f, 0.f};
        if (live) for (int r = r0 + g; r < r1; r += ng) s += *reinterpret_cast<const f4 *>(u + (size_t)r * ld + 4 * cq);
        block_sum(s, psum);
    } else if (pass == 1) {
        f4 s = {0.f, 0.f, 0.f, 0.f};
        if (live) for (int r = r0 + g; r < r1; r += ng) {
            const f4 d = *reinterpret_cast<const f4 *>(u + (size_t)r * ld + 4 * cq) - mean;
            s += d * d;
        }
        block_sum(s, psq);
    } else if (live) {
        const f4 var = total(psq);
        f4 inv, ga, be;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            inv[i] = 1.f / sqrtf(var[i] + eps);
            ga[i] = gamma[4 * cq + i];
            be[i] = beta[4 * cq + i];
        }
        for (int r = r0 + g; r < r1; r += ng) {
            f4 *p = reinterpret_cast<f4 *>(u + (size_t)r * ld + 4 * cq);
            *p = (*p - mean) * inv * ga + be;
        }
    }
}

// makeCvx (|W|) / proj (max(W, 0)) on the packed 'zu_proj' operands of a model, both orientations, in place
__global__ void clamp_kernel(float *w, size_t count, int mode) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < count; i += (size_t)gridDim.x * blockDim.x)
        w[i] = mode == ICNN_BE_CLAMP_ABS ? fabsf(w[i]) : mode == ICNN_BE_CLAMP_ABS_HALF ? 0.5f * fabsf(w[i]) : fmaxf(w[i], 0.f);
}

}  // namespace

int ctx_check(const icnn_be_fc_ctx &c) {
    if (c.n < 1 || c.n_features < 1 || c.n_layers < 2 || c.n_layers > ICNN_BE_MAX_LAYERS) return ICNN_BE_EINVAL;
    if (c.width[c.n_layers - 1] != 1) return ICNN_BE_EINVAL;
    for (int i = 0; i < c.n_layers; ++i) {
        if (c.width[i] < 1 || !c.w_stage[i] || !c.b_stage[i]) return ICNN_BE_EINVAL;
        if (c.batchnorm && i < c.n_layers - 2 && (!c.bn_gamma[i] || !c.bn_beta[i])) return ICNN_BE_EINVAL;
    }
    return 0;
}

// columns of stage i: [ u_i (width[i], i < L) | yu_i (n) | zu_i (width[i]) | gate_i (width[i-1], i > 0) ]
int ctx_stage_cols(const icnn_be_fc_ctx &c, int i) {
    const int L = c.n_layers - 1;
    return (i < L ? c.width[i] : 0) + c.n + c.width[i] + (i > 0 ? c.width[i - 1] : 0);
}
int ctx_stage_ld(const icnn_be_fc_ctx &c, int i) { return (ctx_stage_cols(c, i) + 3) & ~3; }

size_t ctx_work_floats(const icnn_be_fc_ctx &c, int batch) {
    size_t tot = 0;
    for (int i = 0; i + 1 < c.n_layers; ++i) tot += (size_t)batch * ((c.width[i] + 3) & ~3);
    return tot;
}

// Stage i of the context producer: ONE GEMM over everything that reads prev_i (x for i = 0, else u_{i-1} in `work`), epilogue
// routed to u_i (in `work`, input of the next stage) and to the context rows.  u_i of stage j lives at work + sum_{l<j} batch *
// ld_l, so the stages can be issued one by one (icnn_be_fc_context_stage: data-parallel ranks all-reduce the BatchNorm
// statistics between them) or back to back (icnn_be_fc_context).
static int stage_bn(const icnn_be_fc_ctx &c, int i) { return c.batchnorm && i < c.n_layers - 2; }
static float *stage_u(const icnn_be_fc_ctx &c, int i, int batch, float *work, int &u_ld) {
    float *wk = work;
    for (int l = 0; l < i; ++l) wk += (size_t)batch * ((c.width[l] + 3) & ~3);
    u_ld = (c.width[i] + 3) & ~3;
    return wk;
}
hipError_t launch_fc_context_stage(const icnn_be_fc_ctx &c, int i, const float *x, int batch, float *ctx, int ctx_width,
                                   float *work, hipStream_t stream) {
    const int L = c.n_layers - 1;
    int expect = 0, ctx_off = 0;
    for (int l = 0; l <= L; ++l) {
        if (l == i) ctx_off = expect;
        expect += c.n + c.width[l] + (l > 0 ? c.width[l - 1] : 0);
    }
    if (expect != ctx_width || i < 0 || i > L) return hipErrorInvalidValue;       // before anything is written
    const float *prev = x;
    int prev_ld = c.n_features, prev_k = c.n_features;
    if (i > 0) { prev = stage_u(c, i - 1, batch, work, prev_ld); prev_k = c.width[i - 1]; }
    CtxGemmArgs a{};
    a.A = prev; a.lda = prev_ld; a.M = batch; a.K = prev_k;
    a.W = c.w_stage[i]; a.ldw = ctx_stage_ld(c, i); a.N = ctx_stage_cols(c, i); a.bias = c.b_stage[i];
    a.a_vec = (prev_ld % 4 == 0) && (reinterpret_cast<uintptr_t>(prev) % 16 == 0);
    int col = 0, s = 0;
    if (i < L) {            // u_i: input of the next stage; hidden layers are ReLU'd (:343), the last one is linear
        int u_ld = 0;
        float *u_out = stage_u(c, i, batch, work, u_ld);
        a.seg[s++] = CtxSeg{col, col + c.width[i], u_ld, 0, i < L - 1 ? 1 : 0, 0, u_out};
        col += c.width[i];
    }
    a.seg[s++] = CtxSeg{col, col + c.n, ctx_width, ctx_off, 0, 0, ctx};                       // yu_i
    col += c.n; ctx_off += c.n;
    a.seg[s++] = CtxSeg{col, col + c.width[i], ctx_width, ctx_off, 0, 0, ctx};                // zu_i
    col += c.width[i]; ctx_off += c.width[i];
    if (i > 0) {
        a.seg[s++] = CtxSeg{col, col + c.width[i - 1], ctx_width, ctx_off, 1, 0, ctx};        // gate_i = relu(.)
        col += c.width[i - 1]; ctx_off += c.width[i - 1];
    }
    a.nseg = s;
    hipLaunchKernelGGL(ctx_gemm_kernel, dim3((batch + BM - 1) / BM, (a.N + BN - 1) / BN), dim3(GT), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_fc_context(const icnn_be_fc_ctx &c, const float *x, int batch, float *ctx, int ctx_width, float *work,
                             hipStream_t stream) {
    const int L = c.n_layers - 1;
    for (int i = 0; i <= L; ++i) {
        hipError_t e = launch_fc_context_stage(c, i, x, batch, ctx, ctx_width, work, stream);
        if (e != hipSuccess) return e;
        if (i < L && stage_bn(c, i)) {
            int u_ld = 0;
            float *u_out = stage_u(c, i, batch, work, u_ld);
            hipLaunchKernelGGL(ctx_bn_kernel, dim3((c.width[i] + BNC - 1) / BNC), dim3(BNT), 0, stream, u_out, u_ld, batch,
                               c.width[i], c.bn_gamma[i], c.bn_beta[i], c.bn_eps);
            e = hipGetLastError();
            if (e != hipSuccess) return e;
        }
    }
    return hipSuccess;
}

// statistics of stage i's u (this rank's rows) -> stats[2][width_i] float64; 1 = the stage has no BatchNorm (nothing written)
int launch_fc_context_sums(const icnn_be_fc_ctx &c, int i, int batch, float *work, double *stats, hipStream_t stream,
                           hipError_t &err) {
    err = hipSuccess;
    if (i < 0 || i >= c.n_layers - 1 || !stage_bn(c, i)) return 1;
    int u_ld = 0;
    const float *u = stage_u(c, i, batch, work, u_ld);
    hipLaunchKernelGGL(ctx_bn_sums_kernel, dim3((c.width[i] + BNC - 1) / BNC), dim3(BNT), 0, stream, u, u_ld, batch, c.width[i],
                       stats);
    err = hipGetLastError();
    return 0;
}
hipError_t launch_fc_context_norm(const icnn_be_fc_ctx &c, int i, int batch, double batch_total, const double *stats,
                                  float *work, hipStream_t stream) {
    if (i < 0 || i >= c.n_layers - 1 || !stage_bn(c, i)) return hipErrorInvalidValue;
    int u_ld = 0;
    float *u = stage_u(c, i, batch, work, u_ld);
    hipLaunchKernelGGL(ctx_bn_apply_kernel, dim3((c.width[i] + BNC - 1) / BNC), dim3(BNT), 0, stream, u, u_ld, batch, c.width[i],
                       stats, batch_total, c.bn_gamma[i], c.bn_beta[i], c.bn_eps);
    return hipGetLastError();
}

// ---- conv PICNN (completion/icnn_ebundle.py:346-367 u-path, :376-452 heads) --------------------------------------
// Seven GEMM launches (every operand that reads the same input through the same window is one launch) and four
// BatchNorm launches; u-maps stay NHWC in `work`, heads go straight into the context rows.
size_t conv_ctx_work_floats(const ConvCtxShape &g, int batch) {
    return (size_t)batch * ((size_t)g.P[0] * g.F[0] + (size_t)g.P[1] * g.F[1] + (size_t)g.P[2] * g.F[2] + (size_t)((g.fch + 3) & ~3)) +
           2 * (size_t)BNB * 256;                    // + the BatchNorm partials
}

hipError_t launch_conv_context(const ConvCtxShape &g, const icnn_be_conv_ctx &c, const float *x, int batch, float *ctx,
                               float *work, hipStream_t stream) {
    float *u0 = work, *u1 = u0 + (size_t)batch * g.P[0] * g.F[0], *u2 = u1 + (size_t)batch * g.P[1] * g.F[1],
          *u3 = u2 + (size_t)batch * g.P[2] * g.F[2], *part = u3 + (size_t)batch * ((g.fch + 3) & ~3);
    const int C = g.ctx_width, u3_ld = (g.fch + 3) & ~3;
    auto gemm = [&](int stage, const float *A, int conv, int IH, int IW, int IC, int KS, int ST, int PD, int OH, int OW,
                    int lda, int K, std::initializer_list<CtxSeg> segs) -> hipError_t {
        CtxGemmArgs a{};
        a.A = A; a.W = c.w_stage[stage]; a.bias = c.b_stage[stage];
        a.conv = conv; a.IH = IH; a.IW = IW; a.IC = IC; a.KS = KS; a.ST = ST; a.PD = PD; a.OW = OW; a.P = OH * OW;
        a.M = conv ? batch * OH * OW : batch;
        a.lda = lda; a.K = K;
        int n = 0;
        for (const CtxSeg &sg : segs) { a.seg[a.nseg++] = sg; n = sg.c1; }
        a.N = n; a.ldw = (n + 3) & ~3;
        a.a_vec = conv ? (IC % 4 == 0) : (lda % 4 == 0 && reinterpret_cast<uintptr_t>(A) % 16 == 0);
        hipLaunchKernelGGL(ctx_gemm_kernel, dim3((a.M + BM - 1) / BM, (a.N + BN - 1) / BN), dim3(GT), 0, stream, a);
        return hipGetLastError();
    };
    auto bn = [&](float *u, int ld, int rows, int cols, int i) -> hipError_t {
        if (rows >= 16 * BNB && cols % 4 == 0 && cols <= 256 && ld == cols) {          // tall: row-parallel passes
            for (int pass = 0; pass < 3; ++pass)
                hipLaunchKernelGGL(ctx_bn_tall_kernel, dim3(BNB), dim3(TBT), 0, stream, u, ld, rows, cols, part, c.bn_gamma[i],
                                   c.bn_beta[i], c.bn_eps, pass);
            return hipGetLastError();
        }
        hipLaunchKernelGGL(ctx_bn_kernel, dim3((cols + BNC - 1) / BNC), dim3(BNT), 0, stream, u, ld, rows, cols, c.bn_gamma[i],
                           c.bn_beta[i], c.bn_eps);
        return hipGetLastError();
    };
    const int H = g.H, W = g.W, F0 = g.F[0], F1 = g.F[1], F2 = g.F[2];
    const int *oh = g.oh, *ow = g.ow, *K = g.K, *S = g.S, *Pd = g.pad, *P = g.P;
    hipError_t e;
    // input x: 8x8/4 -> u0 (ReLU, then BN) | zu0;   3x3/1 -> yu0
    e = gemm(0, x, 1, H, W, 1, K[0], S[0], Pd[0], oh[0], ow[0], 0, K[0] * K[0],
             {CtxSeg{0, F0, F0, 0, 1, 0, u0}, CtxSeg{F0, 2 * F0, C, g.c_zu[0], 0, P[0], ctx}});
    if (e != hipSuccess) return e;
    e = gemm(1, x, 1, H, W, 1, 3, 1, 1, H, W, 0, 9, {CtxSeg{0, 1, C, g.c_yu[0], 0, H * W, ctx}});
    if (e != hipSuccess) return e;
    if ((e = bn(u0, F0, batch * P[0], F0, 0)) != hipSuccess) return e;
    // input u0: 4x4/2 -> u1 | zu1;   3x3/1 -> gate1 (ReLU) | yu1
    e = gemm(2, u0, 1, oh[0], ow[0], F0, K[1], S[1], Pd[1], oh[1], ow[1], 0, K[1] * K[1] * F0,
             {CtxSeg{0, F1, F1, 0, 1, 0, u1}, CtxSeg{F1, 2 * F1, C, g.c_zu[1], 0, P[1], ctx}});
    if (e != hipSuccess) return e;
    e = gemm(3, u0, 1, oh[0], ow[0], F0, 3, 1, 1, oh[0], ow[0], 0, 9 * F0,
             {CtxSeg{0, F0, C, g.c_gate[1], 1, P[0], ctx}, CtxSeg{F0, F0 + 1, C, g.c_yu[1], 0, P[0], ctx}});
    if (e != hipSuccess) return e;
    if ((e = bn(u1, F1, batch * P[1], F1, 1)) != hipSuccess) return e;
    // input u1: 3x3/1 -> u2 | gate2 (ReLU) | yu2 | zu2
    e = gemm(4, u1, 1, oh[1], ow[1], F1, K[2], S[2], Pd[2], oh[2], ow[2], 0, K[2] * K[2] * F1,
             {CtxSeg{0, F2, F2, 0, 1, 0, u2}, CtxSeg{F2, F2 + F1, C, g.c_gate[2], 1, P[1], ctx},
              CtxSeg{F2 + F1, F2 + F1 + 1, C, g.c_yu[2], 0, P[1], ctx},
              CtxSeg{F2 + F1 + 1, 2 * F2 + F1 + 1, C, g.c_zu[2], 0, P[2], ctx}});
    if (e != hipSuccess) return e;
    if ((e = bn(u2, F2, batch * P[2], F2, 2)) != hipSuccess) return e;
    // flat u2 [B][flat]: -> u3 (ReLU, BN) | gate3 (ReLU) | zu3;   u3 -> gate4 (ReLU) | zu4
    e = gemm(5, u2, 0, 0, 0, 0, 0, 0, 0, 1, 1, g.flat, g.flat,
             {CtxSeg{0, g.fch, u3_ld, 0, 1, 0, u3}, CtxSeg{g.fch, g.fch + g.flat, C, g.c_gate[3], 1, 0, ctx},
              CtxSeg{g.fch + g.flat, 2 * g.fch + g.flat, C, g.c_zu3, 0, 0, ctx}});
    if (e != hipSuccess) return e;
    if ((e = bn(u3, u3_ld, batch, g.fch, 3)) != hipSuccess) return e;
    return gemm(6, u3, 0, 0, 0, 0, 0, 0, 0, 1, 1, u3_ld, g.fch,
                {CtxSeg{0, g.fch, C, g.c_gate[4], 1, 0, ctx}, CtxSeg{g.fch, g.fch + 1, C, g.c_zu4, 0, 0, ctx}});
}

hipError_t launch_clamp(float *w, size_t count, int mode, hipStream_t stream) {
    if (count == 0) return hipSuccess;
    const int blocks = (int)((count + 255) / 256 < 2048 ? (count + 255) / 256 : 2048);
    hipLaunchKernelGGL(clamp_kernel, dim3(blocks), dim3(256), 0, stream, w, count, mode);
    return hipGetLastError();
}

}  // namespace icnn_be
