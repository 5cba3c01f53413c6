// Dual step for NARROW rows (n <= 16), variant RL: FOUR samples per wave64, one per 16-lane DPP row.
#pragma once
// The wave-per-sample kernel (be_dual_dev.h) gives a sample with n = 6 columns and at most five cuts -- the RL agent's
// action vector, RL/src/icnn.py:148-158 with RL/src/bundle_entropy.py:85-136 -- a whole wave64: 58 of 64 lanes idle in the
// column phase, 59 in the row algebra, and its bundle is staged in LDS for two MFMA instructions' worth of work.  The row
// algebra of that kernel already lives in ONE 16-lane DPP row (row broadcasts, butterfly reductions), so here every
// 16-lane row of a wave is a sample of its own:
//
//   lane r of a row is BOTH column r of the bundle (a = A^T lam, z, w, softplus terms, y) and bundle row r (lam_r, c_r,
//   gradient, Newton row r); the bundle is register resident in both orientations (Acol[i] = A[i][r], Arow[j] = A[r][j],
//   loaded straight from the state's G -- no LDS at all); H = A diag(w) A^T and A z are formed on the VALU as exactly
//   the fused multiply-add chains v_mfma_f64_4x4x4 / 16x16x4 apply (ascending k from the accumulator, measured:
//   tools/probes/mfma_f64_order_probe.hip), two chains over the column quads where the 8x8 MFMA form is used; the
//   reduced Newton system is eliminated with DPP64 row broadcasts as in newton_step_dpp, four systems at once.
//
// Samples of a wave run their Newton updates and Armijo trials in lockstep under per-row predicates (everything that was
// wave-uniform per sample -- k, pivot, step length, loop exits -- is a per-lane value that is equal within a row).
// EVERY floating-point operation is the one the wave-per-sample kernel performs on the same operands in the same order,
// so the two kernels agree bit for bit (tests/test_gpu_parity.py); that kernel remains the reference implementation.
//
// Cited lines: rl = RL/src/bundle_entropy.py.
#include "be_dual_dev.h"

namespace icnn_be {
namespace {

template <int P> __device__ __forceinline__ double rbc(double v) { return row_bcast<P>(v); }
template <int P> __device__ __forceinline__ float rbc(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x150 + P, 0xf, 0xf, true));
}
template <int P> __device__ __forceinline__ int rbc(int v) { return __builtin_amdgcn_update_dpp(0, v, 0x150 + P, 0xf, 0xf, true); }

// lane `src` (0..15, equal within the row) of the caller's own 16-lane row
__device__ __forceinline__ double row_shfl(double v, int src) { return __shfl(v, (thread_id() & 48) + src); }
__device__ __forceinline__ unsigned row_ballot(bool p) { return (unsigned)(__ballot(p) >> (thread_id() & 48)) & 0xffffu; }
// butterfly over the 16 lanes of every row; all lanes of a row end up with the value lane 0 of row16_reduce returns
template <typename Op> __device__ __forceinline__ double row_all(double v, Op op) {
    v = op(v, dpp_move<0xB1>(v));
    v = op(v, dpp_move<0x4E>(v));
    v = op(v, dpp_move<0x141>(v));
    v = op(v, dpp_move<0x140>(v));
    return v;
}

// NumPy's pairwise sum (be_common.h, np_pairwise_rows) of n <= 16 values: fewer than 8 sequentially from zero; otherwise
// eight strided accumulators combined as ((a0+a1)+(a2+a3))+((a4+a5)+(a6+a7)), then the tail sequentially.  n is wave-uniform.
template <typename T> __device__ __forceinline__ T np_sum16(const T (&v)[16], int n) {
    if (n < 8) {
        T res = (T)0;
        static_for<0, 8>([&](auto J) { constexpr int j = decltype(J)::value; if (j < n) res = res + v[j]; });
        return res;
    }
    const int body = n & ~7;
    T a[8];
    static_for<0, 8>([&](auto C) { constexpr int c = decltype(C)::value; a[c] = v[c]; if (body == 16) a[c] = a[c] + v[8 + c]; });
    T res = ((a[0] + a[1]) + (a[2] + a[3])) + ((a[4] + a[5]) + (a[6] + a[7]));
    static_for<8, 16>([&](auto J) { constexpr int j = decltype(J)::value; if (j >= body && j < n) res = res + v[j]; });
    return res;
}
// ... of the values e held one per lane of the row (element j in lane j)
template <typename T> __device__ __forceinline__ T np_sum_row(T e, int n) {
    T v[16];
    static_for<0, 16>([&](auto J) { constexpr int j = decltype(J)::value; v[j] = rbc<j>(e); });
    return np_sum16(v, n);
}

struct SmallArgs {
    icnn_be_state st;
    const void *f;
    const void *g;
    int round;
};

// Samples u0 .. u0 + 3 (those below st.batch), one per 16-lane row of the calling wave.  KS = rows the register-resident
// bundle holds (every unrolled loop over the bundle is KS long, the elimination KS^2): 5 (slots <= 5 -- the RL default
// nIter), 8 (slots <= 7; up to here k + 1 <= 8: always the 8x8 MFMA form) or 16 (slots <= 15, both forms, per sample).
// nsamp: rows of this wave that hold a sample (4, or what a persistent workgroup of be_fused.hip owns).
template <typename CutT, int KS, typename ArgsT>
__device__ __forceinline__ void dual_step_quad_rl(const ArgsT &a, int u0, int nsamp, int round) {
    static_assert(KS == 5 || KS == 8 || KS == 16, "rows of the register-resident bundle");
    const auto &st = a.st;
    const int lane = thread_id() & 63, r = lane & 15;
    const int n = st.n, T = st.slots;
    const int u_raw = u0 + (lane >> 4);
    const bool valid = (lane >> 4) < nsamp && u_raw < st.batch;
    const int u = valid ? u_raw : st.batch - 1;                  // clamped: loads stay in bounds, stores are predicated
    const int finished_u = st.finished[u], t = st.t_next[u], cnt = st.count[u];
    bool live = valid && finished_u == 0 && t < T;
    if (!__any(live)) return;
    const int k = cnt + 1;                                       // <= T <= KS
    const int my_slot = r < cnt ? st.active[(size_t)u * T + r] : (r == cnt ? t : 0);

    const CutT *g_row = static_cast<const CutT *>(a.g) + (size_t)u * n;
    const double f_u = (sizeof(CutT) == 4 && (st.flags & ICNN_BE_FLAG_F64_ENERGY))
                           ? static_cast<const double *>(a.f)[u] : (double)static_cast<const CutT *>(a.f)[u];
    double *y_row = st.y + (size_t)u * n;
    CutT *G_u = static_cast<CutT *>(st.G) + (size_t)u * T * n;
    double *ys_u = st.ys + (size_t)u * T * n;
    double *h_u = st.h + (size_t)u * T;

    // ---- 1. the new cut (rl :102-110) and the register-resident bundle, all loads in flight together --------------
    const bool col = r < n;
    const CutT g_r = col ? g_row[r] : (CutT)0;
    const double y_r = col ? y_row[r] : 0.0;
    const double h_old = r < cnt ? h_u[my_slot] : 0.0;
    CutT Acol[KS], Arow[16];
    static_for<0, KS>([&](auto I) {
        constexpr int i = decltype(I)::value;
        const int slot_i = rbc<i>(my_slot);
        Acol[i] = (i < cnt && col) ? G_u[(size_t)slot_i * n + r] : (CutT)0;
    });
    static_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value;
        Arow[j] = (r < cnt && j < n) ? G_u[(size_t)my_slot * n + j] : (CutT)0;
    });
    static_for<0, KS>([&](auto I) { constexpr int i = decltype(I)::value; if (i == cnt) Acol[i] = g_r; });
    static_for<0, 16>([&](auto J) {
        constexpr int j = decltype(J)::value;
        const CutT gj = rbc<j>(g_r);
        if (r == cnt) Arow[j] = gj;
    });
    bool bad = !isfinite(f_u) || (col && !isfinite((double)g_r));
    if (live && col) {
        G_u[(size_t)t * n + r] = g_r;
        ys_u[(size_t)t * n + r] = y_r;
    }
    const double prod = (double)g_r * y_r;                       // rl :106  gi * x in float64 (0 for r >= n)
    const double h_new = f_u - np_sum_row(prod, n);              // fi - np.sum(gi * x)
    if (live && r == 0) {
        h_u[t] = h_new;
        if (st.fvals) st.fvals[(size_t)u * T + t] = f_u;
    }
    bad = row_ballot(bad) != 0;
    if (live && bad && r == 0) { st.status[u] |= ICNN_BE_ST_NONFINITE; st.finished[u] = 1; st.skip_fg[u] = 1; }
    live = live && !bad;
    const double h_i = r < cnt ? h_old : h_new;                  // row layout (r < k)

    // ---- 4. multipliers: projected Newton on the simplex (rl :14-83), row layout ----------------------------------
    double lam = (k == 1) ? (r == 0 ? 1.0 : 0.0) : (r < k ? 1.0 / (double)k : 0.0);     // rl :112 / :24
    int updates = 0;
    bool singular = false;
    {
        // c = np.sum(A, axis=1) + b with the row sum in the cut dtype (rl :17)
        const double c_i = r < k ? (double)np_sum16(Arow, n) + h_i : 0.0;
        const int cap = 20, backoff_cap = 10;                    // rl :29, :65
        const bool shortcut = !(st.flags & ICNN_BE_FLAG_NO_CYCLE_SHORTCUT);
        double prev1 = 0.0, prev2 = 0.0, prev3 = 0.0;
        int hist = 0;
        bool run = live && k > 1;                                // this row's sample is still in its Newton loop
        const bool path8 = KS <= 8 || k + 1 <= 8;                // the 8x8 MFMA form: two accumulator chains (be_dual_dev.h)
        while (__any(run)) {
            // a = A^T lam, z = sigmoid(a), w = z (1 - z), softplus terms                     rl :31-34
            // (rows i >= k contribute lam_i A[i][j] = 0 * 0: the chain of the wave-per-sample kernel, which stops at k, and
            //  this one agree bit for bit -- an accumulator that starts at +0 never becomes -0)
            double aj = 0.0;
            static_for<0, KS>([&](auto I) {
                constexpr int i = decltype(I)::value;
                aj = aj + rbc<i>(lam) * (double)Acol[i];
            });
            double z = sigmoid_fast(aj);                         // (be_dual_dev.h; the same routine as the wave-per-sample kernel)
            double w = z * (1.0 - z);
            if (!col) { z = 0.0; w = 0.0; }
            const double sp = col ? softplus_fast(aj) : 0.0;
            // H = A diag(w) A^T and A z, row r of both in lane r: the MFMA's chains over the columns
            double Pcol[KS];
            static_for<0, KS>([&](auto J) { constexpr int j = decltype(J)::value; Pcol[j] = (double)Acol[j] * w; });
            double H0[KS], H1[KS], Az0 = 0.0, Az1 = 0.0;
            static_for<0, KS>([&](auto J) { constexpr int j = decltype(J)::value; H0[j] = 0.0; H1[j] = 0.0; });
            static_for<0, 16>([&](auto C) {
                constexpr int c = decltype(C)::value;
                if (c < n) {                                     // (columns >= n hold zeros: fma(0, 0, acc) = acc)
                    const double ar = (double)Arow[c];
                    constexpr bool odd = ((c >> 2) & 1) != 0;    // second and fourth MFMA of a 16-column stage
                    const double zc = rbc<c>(z);
                    if (!odd) {
                        Az0 = __builtin_fma(ar, zc, Az0);
                    } else if (KS <= 8) {
                        Az1 = __builtin_fma(ar, zc, Az1);
                    } else {
                        const double tz = __builtin_fma(ar, zc, path8 ? Az1 : Az0);
                        Az1 = path8 ? tz : Az1;
                        Az0 = path8 ? Az0 : tz;
                    }
                    static_for<0, KS>([&](auto J) {
                        constexpr int j = decltype(J)::value;
                        const double b = rbc<c>(Pcol[j]);
                        if (!odd) {
                            H0[j] = __builtin_fma(ar, b, H0[j]);
                        } else if (KS <= 8) {
                            H1[j] = __builtin_fma(ar, b, H1[j]);
                        } else {
                            const double th = __builtin_fma(ar, b, path8 ? H1[j] : H0[j]);
                            H1[j] = path8 ? th : H1[j];
                            H0[j] = path8 ? H0[j] : th;
                        }
                    });
                }
            });
            double Hrow[KS];
            static_for<0, KS>([&](auto J) { constexpr int j = decltype(J)::value; Hrow[j] = path8 ? H0[j] + H1[j] : H0[j]; });
            const double Az = path8 ? Az0 + Az1 : Az0;

            const double grad = r < k ? -c_i + Az : 0.0;                             // rl :34
            // first maximum of lam (:38)
            const double mx = row_all(r < k ? lam : -1e300, [](double x, double y) { return fmax(x, y); });
            const unsigned at = row_ballot(r < k && lam == mx);
            const int piv = at ? __builtin_ctz(at) : 0;
            const bool is_piv = r == piv;
            const double red = is_piv ? 1.0 : lam;                                   // :39-40
            const double keep = is_piv ? 0.0 : 1.0;                                  // :41
            const double g0 = grad - keep * row_shfl(grad, piv);                     // :43
            const bool bound = is_piv || (red <= BOUND_EPS && g0 > 0.0);             // :47-48
            const bool is_free = r < k && !bound;
            const unsigned fmask = row_ballot(is_free);
            const double nrm2 = row_all(is_free ? g0 * g0 : 0.0, [](double x, double y) { return x + y; });
            const bool small = sqrt(nrm2) < GRAD_TOL;                                // :49 -> return lam

            // reduced Newton system (newton_step_dpp, four systems side by side)
            double M[KS + 1];
            double h_ip = 0.0;
            static_for<0, KS>([&](auto J) { constexpr int j = decltype(J)::value; h_ip = piv == j ? Hrow[j] : h_ip; });
            const double h_pp = row_shfl(h_ip, piv);
            static_for<0, KS>([&](auto J) {
                constexpr int j = decltype(J)::value;
                double hv = ((Hrow[j] - rbc<j>(h_ip)) - h_ip) + h_pp;
                pin(hv);
                const bool use = j < k && is_free && ((fmask >> j) & 1u);
                M[j] = use ? hv : (j == r ? 1.0 : 0.0);
            });
            M[KS] = is_free ? -g0 : 0.0;
            double rinv = 1.0;
            bool zero_pivot = false;
            static_for<0, KS>([&](auto P) {
                constexpr int p = decltype(P)::value;
                const double d = rbc<p>(M[p]);
                zero_pivot |= !(d != 0.0);
                const double inv = rcp_nr(d);
                rinv = r == p ? inv : rinv;
                const double nf = r > p ? -(M[p] * inv) : 0.0;
                static_for<p + 1, KS + 1>([&](auto J) {
                    constexpr int j = decltype(J)::value;
                    M[j] = __builtin_fma(nf, rbc<p>(M[j]), M[j]);
                });
                __builtin_amdgcn_sched_barrier(0);
            });
            zero_pivot = row_ballot(zero_pivot) != 0;
            static_for<0, KS>([&](auto Q) {
                constexpr int p = KS - 1 - decltype(Q)::value;
                const double x = rbc<p>(M[KS] * rinv);
                M[KS] = r == p ? x : (r < p ? __builtin_fma(-M[p], x, M[KS]) : M[KS]);
            });
            const double step = is_free ? M[KS] : 0.0;
            const bool ls = run && !small && !zero_pivot;        // this row goes through the line search
            if (run && !small && zero_pivot) singular = true;    // rl :55-62: keep lam, leave the loop

            // Armijo line search (rl :64-78)
            const double dmax = row_all(fabs(step), [](double x, double y) { return fmax(x, y); });
            double tt = fmin(1.0 / dmax, 1.0);                                       // :64
            const double psum = np_sum_row(sp, n);
            double cl = 0.0, slope = 0.0;
            {
                const double pc = c_i * lam, ps = step * g0;         // both 0 in lanes >= k
                static_for<0, KS>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    cl = cl + rbc<i>(pc);
                    slope = slope + rbc<i>(ps);
                });
            }
            const double fval = -cl + psum;                                          // :34
            double lam_new = lam;
            bool returned = false, searching = ls;
            for (int bt = 0; bt < backoff_cap; ++bt) {
                if (!__any(searching)) break;
                const double trial = is_piv ? 1.0 : fmax(red + tt * step, 0.0);      // :66-67
                const double s = row_all((r < k && !is_piv) ? trial : 0.0, [](double x, double y) { return x + y; });
                const double lam_p = 1.0 - s;                                        // :69
                const double cand = r < k ? (is_piv ? lam_p : trial) : 0.0;
                double a2 = 0.0;
                static_for<0, KS>([&](auto I) {
                    constexpr int i = decltype(I)::value;
                    a2 = a2 + rbc<i>(cand) * (double)Acol[i];
                });
                const double sp2 = col ? softplus_fast(a2) : 0.0;
                const double psum2 = np_sum_row(sp2, n);
                double cl2 = 0.0;
                {
                    const double pc = c_i * cand;
                    static_for<0, KS>([&](auto I) { constexpr int i = decltype(I)::value; cl2 = cl2 + rbc<i>(pc); });
                }
                const double f_new = -cl2 + psum2;
                const bool accept = lam_p >= 0.0 && f_new < fval + tt * ARMIJO_ALPHA * slope;   // :71-74
                const double mv = row_all(tt * fabs(step), [](double x, double y) { return fmax(x, y); });
                const bool tiny = !accept && mv < TINY;                              // :77
                if (searching) {
                    lam_new = cand;
                    if (accept) searching = false;
                    else if (tiny) { returned = true; searching = false; }
                    else tt *= 0.5;
                }
            }
            // bookkeeping of the update, per row
            bool leave = run && (small || zero_pivot);           // lam unchanged
            if (ls) {
                ++updates;
                double lam_next = lam_new;
                bool stop = returned;
                if (!stop && shortcut && hist >= 1) {
                    if (row_ballot(fabs(lam_new - prev1) > CYCLE_TOL) == 0) {
                        stop = true;
                    } else if (hist >= 2 && row_ballot(fabs(lam_new - prev2) > CYCLE_TOL) == 0) {
                        lam_next = ((cap - updates) & 1) ? prev1 : lam_new;
                        stop = true;
                    } else if (hist >= 3 && row_ballot(fabs(lam_new - prev3) > CYCLE_TOL) == 0) {
                        const int rem = (cap - updates) % 3;
                        lam_next = rem == 0 ? lam_new : (rem == 1 ? prev2 : prev1);
                        stop = true;
                    }
                }
                if (!stop) {
                    prev3 = prev2;
                    prev2 = prev1;
                    prev1 = lam_new;
                    hist = hist < 4 ? hist + 1 : 4;
                }
                lam = lam_next;
                leave = stop || updates >= cap;
            }
            run = run && !leave;
        }
    }
    if (live && singular && r == 0) st.status[u] |= ICNN_BE_ST_SINGULAR;

    // ---- 5. y <- clip(sigmoid(-A^T lam)), stall test, prune (rl :114-131) --------------------------------------
    double ynew;
    {
        double aj = 0.0;
        static_for<0, KS>([&](auto I) { constexpr int i = decltype(I)::value; aj = aj + rbc<i>(lam) * (double)Acol[i]; });
        const double y_many = 1.0 / (1.0 + exp(aj));                                 // rl :116
        const double y_one = (double)Cut<CutT>::sigmoid_neg(Acol[0]);                // rl :114, cut-dtype arithmetic
        ynew = k == 1 ? y_one : y_many;
    }
    ynew = fmin(fmax(ynew, 0.03), 0.97);                                             // rl :118,:123
    const double move = row_all(col ? fabs(y_r - ynew) : 0.0, [](double x, double y) { return fmax(x, y); });
    const bool nonfinite = row_ballot(col && !isfinite(ynew)) != 0;
    if (live && col) y_row[r] = ynew;
    bool fin = move < 1e-6;                                                          // rl :125-126
    if (nonfinite) fin = true;
    const bool pos = r < k && lam > 0.0;                                             // rl :127-131
    const unsigned pmask = row_ballot(pos);
    if (live && pos) {
        const int at = __popc(pmask & ((1u << r) - 1u));
        st.active[(size_t)u * T + at] = my_slot;
        st.lam[(size_t)u * T + at] = lam;
    }
    if (live && r == 0) {
        if (nonfinite) st.status[u] |= ICNN_BE_ST_NONFINITE;
        st.count[u] = __popc(pmask);
        st.newton_iters[u] += updates;
        if (fin) st.finished[u] = 1;
        const bool more = !fin && t + 1 < T;
        st.t_next[u] = t + 1;
        st.phase[u] = 0;
        st.skip_fg[u] = more ? 0 : 1;
        if (more) st.pending[round] = 1;
    }
}

}  // namespace
}  // namespace icnn_be
