// Dual step of the bundle-entropy method: kernels and launchers (device code: be_dual_dev.h).
#include "be_dual_dev.h"

namespace icnn_be {

namespace {

// ---------------------------------------------------------------------------------------------
// Implicit-differentiation feed: per sample, from the solver's result,
//   Z^-1 = 1/(1/y + 1/(1-y)),  [[G Z^-1 G^T, 1], [1^T, 0]] [c_lam; c_t] = [G Z^-1 dl; 0],
//   c_y = Z^-1 dl - (G Z^-1)^T c_lam,  rows (y = ys_i, v = lam_i c_y + c_lam,i (y* - ys_i), c = c_lam,i)
// multi-label-cls/icnn_ebundle.py:296-314 + crossEntrGrad :390-417; completion/icnn_ebundle.py:315-335 +
// mseGrad :493-522.  Same layout as the dual step: bundle rows in LDS, G Z^-1 G^T and G Z^-1 dl from one
// f64 MFMA sweep, the bordered system by block elimination on the SPD block (two right-hand sides).
// ---------------------------------------------------------------------------------------------
template <int KS>
__device__ __noinline__ void spd_solve2_ks(const double *Hm, int HP, int k, double &x1, double &x2) {
    const int lane = lane_id();
    double M[KS + 2];
#pragma unroll
    for (int j = 0; j < KS; ++j) {
        double v = j == lane ? 1.0 : 0.0;
        if (lane < k && j < k) v = Hm[lane * HP + j];
        M[j] = v;
    }
    M[KS] = lane < k ? Hm[lane * HP + k] : 0.0;
    M[KS + 1] = lane < k ? 1.0 : 0.0;
#pragma unroll
    for (int p = 0; p < KS; ++p) {
        if (p < k) {
            const double d = bcast(M[p], p);
            const double f = lane > p ? M[p] / d : 0.0;
#pragma unroll
            for (int j = p + 1; j < KS; ++j)
                if (j < k) M[j] -= f * bcast(M[j], p);
            M[KS] -= f * bcast(M[KS], p);
            M[KS + 1] -= f * bcast(M[KS + 1], p);
        }
    }
#pragma unroll
    for (int p = KS - 1; p >= 0; --p) {
        if (p < k) {
            const double d = bcast(M[p], p);
            const double xa = bcast(M[KS], p) / d, xb = bcast(M[KS + 1], p) / d;
            if (lane == p) { M[KS] = xa; M[KS + 1] = xb; }
            else if (lane < p) { M[KS] -= M[p] * xa; M[KS + 1] -= M[p] * xb; }
        }
    }
    x1 = lane < k ? M[KS] : 0.0;
    x2 = lane < k ? M[KS + 1] : 0.0;
}

struct FeedArgs {
    icnn_be_state st;
    const double *y_true;
    const int *row_offset;
    double *fd_y, *fd_v, *fd_c;
    int *fd_sample;
    int loss, n_pad, ldA;
    PairwisePlan plan;
};

template <typename CutT, int KT, bool GLB = false>      // GLB: the bundle is staged in st.scratch (wide rows, be_dual_dev.h)
__global__ __launch_bounds__(64, 3) void implicit_feed_kernel(FeedArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const icnn_be_state &st = a.st;
    const int u = blockIdx.x, lane = threadIdx.x;
    const int k = __builtin_amdgcn_readfirstlane(st.count[u]);
    if (k == 0) return;                                     // completion/icnn_ebundle.py:319-320
    const int n = st.n, T = st.slots, n_pad = a.n_pad, ldA = a.ldA;
    const int HP = (T + 1) | 1;
    const Carve cv = carve(KT, T, ldA, n_pad, (int)sizeof(CutT), a.plan.n_leaves, false, 1, true, false, GLB);
    CutT *As = GLB ? static_cast<CutT *>(st.scratch) + (size_t)u * (T + 2) * ldA : reinterpret_cast<CutT *>(smem + cv.As);
    double *zs = reinterpret_cast<double *>(smem + cv.zs);
    double *ws = reinterpret_cast<double *>(smem + cv.ws);
    double *Hm = reinterpret_cast<double *>(smem + cv.Hm);
    int *slots = reinterpret_cast<int *>(smem + cv.ints);
    const CutT *G_u = static_cast<const CutT *>(st.G) + (size_t)u * T * n;
    const double *ys_u = st.ys + (size_t)u * T * n;
    const double *y_row = st.y + (size_t)u * n, *t_row = a.y_true + (size_t)u * n;
    if (lane < k) slots[lane] = st.active[(size_t)u * T + lane];
    __syncthreads();
    for (int r = 0; r < k; ++r) {
        const CutT *src = G_u + (size_t)slots[r] * n;
        for (int j = lane; j < n_pad; j += 64) As[r * ldA + j] = j < n ? src[j] : (CutT)0;
    }
    for (int j = lane; j < ldA; j += 64) { As[T * ldA + j] = (CutT)0; As[(T + 1) * ldA + j] = (CutT)1; }
    for (int j = lane; j < n_pad; j += 64) {
        double zinv = 0.0, zd = 0.0;
        if (j < n) {
            const double y = y_row[j], t = t_row[j];
            double z, dl;
            if (a.loss == 0) {                                  // cross entropy, :393-411
                const double yc = fmin(fmax(y, 1e-8), 1.0 - 1e-8);
                z = 1.0 / yc + 1.0 / (1.0 - yc);
                dl = t / yc - (1.0 - t) / (1.0 - yc);
            } else {                                            // squared error, completion :508,:515
                z = 1.0 / y + 1.0 / (1.0 - y);
                dl = -(y - t);
            }
            zinv = 1.0 / z;
            zd = zinv * dl;
        }
        ws[j] = zinv;
        zs[j] = zd;
    }
    __syncthreads();
    contract_mfma<CutT, KT, true>(As, ldA, k, As + T * ldA, 0, n_pad, ws, zs, Hm, HP);   // Hm = G Z^-1 G^T | G Z^-1 dl
    __syncthreads();
    double x1, x2;
    if (k <= 4) spd_solve2_ks<4>(Hm, HP, k, x1, x2);
    else if (k <= 8) spd_solve2_ks<8>(Hm, HP, k, x1, x2);
    else if (KT == 16 || k <= 16) spd_solve2_ks<16>(Hm, HP, k, x1, x2);
    else spd_solve2_ks<KT>(Hm, HP, k, x1, x2);
    double s1 = 0.0, s2 = 0.0;
    for (int i = 0; i < k; ++i) { s1 += bcast(x1, i); s2 += bcast(x2, i); }
    const double ct = s1 / s2;
    const double clam = x1 - ct * x2;                           // row layout, lane i < k
    const double lam = lane < k ? st.lam[(size_t)u * T + lane] : 0.0;
    const int row0 = a.row_offset[u];
    if (lane < k) { a.fd_c[row0 + lane] = clam; a.fd_sample[row0 + lane] = u; }
    for (int j = lane; j < n; j += 64) {
        const double y = y_row[j];
        double gz = 0.0;
        for (int i = 0; i < k; ++i) gz += (double)As[i * ldA + j] * ws[j] * bcast(clam, i);
        double cy = zs[j] - gz;                                 // :415
        if (y == 0.0 || y == 1.0) cy = 0.0;                     // :416
        for (int i = 0; i < k; ++i) {
            const double ysi = ys_u[(size_t)slots[i] * n + j];
            a.fd_y[(size_t)(row0 + i) * n + j] = ysi;                                        // :304
            a.fd_v[(size_t)(row0 + i) * n + j] = bcast(lam, i) * cy + bcast(clam, i) * (y - ysi);   // :305
        }
    }
}

// Pack of the ACTIVE bundle rows, sample by sample in bundle order: what the reference hands back as its ragged lists
// A, b, xs, lam (dual :171-179) and the host mirror copies to the host in one piece.  One workgroup per sample.
struct ExportArgs {
    icnn_be_state st;
    const int *row_offset;
    void *G_rows;
    double *ys_rows, *h_rows, *lam_rows;
};
template <typename CutT>
__global__ __launch_bounds__(256) void export_active_kernel(ExportArgs a) {
    const icnn_be_state &st = a.st;
    const int u = blockIdx.x, T = st.slots, n = st.n;
    const int k = st.count[u], row0 = a.row_offset[u];
    const CutT *G_u = static_cast<const CutT *>(st.G) + (size_t)u * T * n;
    const double *ys_u = st.ys + (size_t)u * T * n;
    CutT *G_out = static_cast<CutT *>(a.G_rows) + (size_t)row0 * n;
    double *ys_out = a.ys_rows + (size_t)row0 * n;
    for (int i = 0; i < k; ++i) {
        const int slot = st.active[(size_t)u * T + i];
        for (int j = threadIdx.x; j < n; j += 256) {
            G_out[(size_t)i * n + j] = G_u[(size_t)slot * n + j];
            ys_out[(size_t)i * n + j] = ys_u[(size_t)slot * n + j];
        }
        if (threadIdx.x == 0) {
            a.h_rows[row0 + i] = st.h[(size_t)u * T + slot];
            a.lam_rows[row0 + i] = st.lam[(size_t)u * T + i];
        }
    }
}

// diagnostic (icnn_be_debug_fast_math): the inner loops' exp / log / softplus / sigmoid evaluated on caller-supplied arguments
__global__ void fast_math_kernel(int which, const double *x, double *out, int count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= count) return;
    const double v = x[i];
    out[i] = which == 0 ? fast_exp(v) : which == 1 ? fast_log(v) : which == 2 ? softplus_fast(v) : sigmoid_fast(v);
}

__global__ void state_init_kernel(icnn_be_state st) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < ICNN_BE_MAX_ROUNDS) st.pending[u] = 0;
    if (u >= st.batch) return;
    st.count[u] = 0;
    st.finished[u] = 0;
    st.status[u] = 0;
    st.n_iters[u] = st.iters > 0 ? st.iters : st.slots;     // dual :139
    st.newton_iters[u] = 0;
    st.t_next[u] = 0;
    st.phase[u] = 0;
    st.skip_fg[u] = 0;
}

// closing launch of a solve whose stragglers were given a fixed number of rounds: who is still behind says so
__global__ void mark_unfinished_kernel(icnn_be_state st) {
    const int u = blockIdx.x * blockDim.x + threadIdx.x;
    if (u < st.batch && st.finished[u] == 0 && st.t_next[u] < (st.iters > 0 ? st.iters : st.slots)) st.status[u] |= ICNN_BE_ST_UNFINISHED;
}

}  // namespace

hipError_t launch_export_active(const icnn_be_state &st, const int *row_offset, void *G_rows, double *ys_rows, double *h_rows,
                                double *lam_rows, hipStream_t stream) {
    ExportArgs a{st, row_offset, G_rows, ys_rows, h_rows, lam_rows};
    if (st.cut_dtype == ICNN_BE_CUT_F64) hipLaunchKernelGGL(export_active_kernel<double>, dim3(st.batch), dim3(256), 0, stream, a);
    else hipLaunchKernelGGL(export_active_kernel<float>, dim3(st.batch), dim3(256), 0, stream, a);
    return hipGetLastError();
}

hipError_t launch_fast_math(int which, const double *x, double *out, int count, hipStream_t stream) {
    hipLaunchKernelGGL(fast_math_kernel, dim3((count + 255) / 256), dim3(256), 0, stream, which, x, out, count);
    return hipGetLastError();
}

hipError_t launch_mark_unfinished(const icnn_be_state &st, hipStream_t stream) {
    hipLaunchKernelGGL(mark_unfinished_kernel, dim3((st.batch + 255) / 256), dim3(256), 0, stream, st);
    return hipGetLastError();
}

static long long *g_prof = nullptr;
void set_dual_profile_buffer(long long *buf) { g_prof = buf; }
long long *dual_profile_buffer() { return g_prof; }
static long long *g_trace = nullptr;
void set_dual_trace_buffer(long long *buf) { g_trace = buf; }
long long *dual_trace_buffer() { return g_trace; }

// waves per sample: wide workgroups only where the column work dominates (n >= 1024), and only for the
// configuration they are implemented for (variant dual, float32 cuts)
int dual_waves(int n, int cut_dtype, int variant) {
    // (round 5: the interior-point variant too -- ipm_solve_waves, be_ipm_dev.h)
    return ((variant == ICNN_BE_VARIANT_DUAL || variant == ICNN_BE_VARIANT_PDIPM) && cut_dtype == ICNN_BE_CUT_F32 && n >= 1024) ? 8 : 1;
}

// bytes of st.scratch ([B][slots + 2][pitch] cuts) that lift the bundle capacity of wide rows to `slots`; 0: not needed
// or not available (the staging area in device memory exists for the variants dual and pdipm)
size_t scratch_bytes(const icnn_be_state &st) {
    if (st.variant != ICNN_BE_VARIANT_DUAL && st.variant != ICNN_BE_VARIANT_PDIPM) return 0;
    if (!(st.flags & ICNN_BE_FLAG_GLOBAL_BUNDLE) && dual_rows_fit(st.n, st.slots, st.cut_dtype, st.variant) >= st.slots) return 0;
    return (size_t)st.batch * (st.slots + 2) * dual_row_pitch((st.n + 15) & ~15) * (st.cut_dtype == ICNN_BE_CUT_F64 ? 8 : 4);
}

int dual_lds_bytes(int n, int slots, int cut_dtype, int variant, int rows) {
    const int KT = slots <= 15 ? 16 : 32;
    if (rows <= 0 || rows > slots) rows = slots;
    const int n_pad = (n + 15) & ~15;
    PairwisePlan plan;
    if (!pw_build(plan, n)) return -1;
    return carve(KT, rows, dual_row_pitch(n_pad), n_pad, cut_dtype == ICNN_BE_CUT_F64 ? 8 : 4,
                 plan.n_leaves, variant == ICNN_BE_VARIANT_RL, dual_waves(n, cut_dtype, variant), true,
                 variant == ICNN_BE_VARIANT_PDIPM).total;
}

int dual_rows_fit(int n, int slots, int cut_dtype, int variant) {
    int rows = slots;
    while (rows > 0) {
        const int b = dual_lds_bytes(n, slots, cut_dtype, variant, rows);
        if (b < 0) return 0;
        if (b <= 160 * 1024) break;
        --rows;
    }
    return rows;
}

hipError_t launch_state_init(const icnn_be_state &st, hipStream_t stream) {
    const int threads = st.batch > ICNN_BE_MAX_ROUNDS ? st.batch : ICNN_BE_MAX_ROUNDS;
    hipLaunchKernelGGL(state_init_kernel, dim3((threads + 255) / 256), dim3(256), 0, stream, st);
    return hipGetLastError();
}

template <typename CutT, int KT, int NW, bool RL, bool IPM = false>
static hipError_t launch_rl(const DualArgs &a, int lds, hipStream_t stream) {
    auto kern = dual_step_kernel<CutT, KT, NW, RL, IPM>;
    if (lds > 48 * 1024)
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.st.batch), dim3(64 * NW), lds, stream, a);
    return hipGetLastError();
}
template <typename CutT, int KT, int NW>
static hipError_t launch_one(const DualArgs &a, int lds, hipStream_t stream) {
    if (a.st.variant == ICNN_BE_VARIANT_RL) {
        if (NW != 1) return hipErrorInvalidValue;          // dual_waves() never widens the RL variant
        return launch_rl<CutT, KT, 1, true>(a, lds, stream);
    }
    if (a.st.variant == ICNN_BE_VARIANT_PDIPM) return launch_rl<CutT, KT, NW, false, true>(a, lds, stream);
    return launch_rl<CutT, KT, NW, false>(a, lds, stream);
}

hipError_t launch_dual_step(const icnn_be_state &st, int round, int budget, const void *f, const void *g,
                            hipStream_t stream) {
    if (dual_step_small_fits(st, budget)) return launch_dual_step_small(st, round, f, g, stream);
    DualArgs a;
    a.st = st;
    a.f = f;
    a.g = g;
    a.round = round;
    a.budget = budget;
    a.n_pad = (st.n + 15) & ~15;
    a.ldA = dual_row_pitch(a.n_pad);
    a.prof = g_prof;
    if (!pw_build(a.plan, st.n)) return hipErrorInvalidValue;
    a.rows = round + 1 < st.slots ? round + 1 : st.slots;
    // wide rows: the LDS holds fewer cuts than there are iterations.  From the round on in which a bundle could outgrow
    // it, the bundle is staged in st.scratch instead (same kernel, the sweeps at L2 latency); without a scratch area the
    // capacity stays and a sample that exceeds it stops with ICNN_BE_ST_OVERFLOW
    const int fit = dual_rows_fit(st.n, st.slots, st.cut_dtype, st.variant);
    if ((a.rows > fit || (st.flags & ICNN_BE_FLAG_GLOBAL_BUNDLE)) && st.scratch && scratch_bytes(st) > 0) {
        const bool ipm = st.variant == ICNN_BE_VARIANT_PDIPM, f64 = st.cut_dtype == ICNN_BE_CUT_F64;
        int nw = dual_waves(st.n, st.cut_dtype, st.variant);
        int lds_g = carve(32, a.rows, a.ldA, a.n_pad, f64 ? 8 : 4, a.plan.n_leaves, false, nw, true, ipm, true).total;
        if (lds_g > 160 * 1024 && nw > 1) {
            // the per-wave systems of an eight-wave sample do not fit next to the column buffers (interior point: five of
            // them; n_pad = 3072 from 23 rows on): the one-wave instance of the same body, whose carve-up has none
            nw = 1;
            lds_g = carve(32, a.rows, a.ldA, a.n_pad, f64 ? 8 : 4, a.plan.n_leaves, false, 1, true, ipm, true).total;
        }
        if (lds_g > 160 * 1024) return hipErrorInvalidValue;
        auto go = [&](auto kern, int waves) -> hipError_t {
            if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds_g); e != hipSuccess) return e;
            hipLaunchKernelGGL(kern, dim3(st.batch), dim3(64 * waves), lds_g, stream, a);
            return hipGetLastError();
        };
        if (f64) return ipm ? go(dual_step_kernel<double, 32, 1, false, true, true>, 1)
                            : go(dual_step_kernel<double, 32, 1, false, false, true>, 1);
        if (ipm) return nw > 1 ? go(dual_step_kernel<float, 32, 8, false, true, true>, 8)
                               : go(dual_step_kernel<float, 32, 1, false, true, true>, 1);
        if (nw > 1 && !(st.flags & ICNN_BE_FLAG_GLOBAL_BUNDLE)) {
            // split staging (dual_step_wide_kernel): LDS for the carve-up of a bundle of up to HV_KMAX cuts + the mirror of
            // its WIDE_LR oldest rows, or for the plain device-memory body of a larger one
            const int mid = a.rows < HV_KMAX ? a.rows : HV_KMAX;
            const int lds_b = ((carve(32, mid, a.ldA, a.n_pad, 4, a.plan.n_leaves, false, nw, true, false, true).total + 15) & ~15) +
                              WIDE_LR * a.ldA * 4;
            const int lds_w = lds_b > lds_g ? lds_b : lds_g;
            if (lds_w <= 160 * 1024) {
                if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(dual_step_wide_kernel), lds_w); e != hipSuccess)
                    return e;
                hipLaunchKernelGGL(dual_step_wide_kernel, dim3(st.batch), dim3(512), lds_w, stream, a);
                return hipGetLastError();
            }
        }
        return nw > 1 ? go(dual_step_kernel<float, 32, 8, false, false, true>, 8)
                      : go(dual_step_kernel<float, 32, 1, false, false, true>, 1);
    }
    if (a.rows > fit) a.rows = fit;
    const int lds = dual_lds_bytes(st.n, st.slots, st.cut_dtype, st.variant, a.rows);
    const bool big = st.slots > 15;
    if (st.cut_dtype == ICNN_BE_CUT_F64)
        return big ? launch_one<double, 32, 1>(a, lds, stream) : launch_one<double, 16, 1>(a, lds, stream);
    if (dual_waves(st.n, st.cut_dtype, st.variant) > 1)
        return big ? launch_one<float, 32, 8>(a, lds, stream) : launch_one<float, 16, 8>(a, lds, stream);
    return big ? launch_one<float, 32, 1>(a, lds, stream) : launch_one<float, 16, 1>(a, lds, stream);
}

template <typename CutT, int KT, bool GLB = false>
static hipError_t launch_feed_one(const FeedArgs &a, int lds, hipStream_t stream) {
    auto kern = implicit_feed_kernel<CutT, KT, GLB>;
    if (lds > 48 * 1024)
        if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(kern), lds); e != hipSuccess) return e;
    hipLaunchKernelGGL(kern, dim3(a.st.batch), dim3(64), lds, stream, a);
    return hipGetLastError();
}

hipError_t launch_implicit_feed(const icnn_be_state &st, const double *y_true, int loss, const int *row_offset,
                                double *fd_y, double *fd_v, double *fd_c, int *fd_sample, hipStream_t stream) {
    FeedArgs a;
    a.st = st;
    a.y_true = y_true; a.row_offset = row_offset;
    a.fd_y = fd_y; a.fd_v = fd_v; a.fd_c = fd_c; a.fd_sample = fd_sample;
    a.loss = loss;
    a.n_pad = (st.n + 15) & ~15;
    a.ldA = dual_row_pitch(a.n_pad);
    if (!pw_build(a.plan, st.n)) return hipErrorInvalidValue;
    const bool big = st.slots > 15;
    const int KT = big ? 32 : 16, cb = st.cut_dtype == ICNN_BE_CUT_F64 ? 8 : 4;     // (one wave per sample here)
    const int lds = carve(KT, st.slots, a.ldA, a.n_pad, cb, a.plan.n_leaves, false).total;
    if (lds > 160 * 1024) {              // wide rows with more slots than LDS rows: the staging area in device memory
        if (!st.scratch || scratch_bytes(st) == 0) return hipErrorInvalidValue;
        const int lds_g = carve(KT, st.slots, a.ldA, a.n_pad, cb, a.plan.n_leaves, false, 1, true, false, true).total;
        if (st.cut_dtype == ICNN_BE_CUT_F64)
            return big ? launch_feed_one<double, 32, true>(a, lds_g, stream) : launch_feed_one<double, 16, true>(a, lds_g, stream);
        return big ? launch_feed_one<float, 32, true>(a, lds_g, stream) : launch_feed_one<float, 16, true>(a, lds_g, stream);
    }
    if (st.cut_dtype == ICNN_BE_CUT_F64)
        return big ? launch_feed_one<double, 32>(a, lds, stream) : launch_feed_one<double, 16>(a, lds, stream);
    return big ? launch_feed_one<float, 32>(a, lds, stream) : launch_feed_one<float, 16>(a, lds, stream);
}

}  // namespace icnn_be
