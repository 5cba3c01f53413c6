// Interior-point variant of the per-sample subproblem (included by be_dual_dev.h, inside its namespaces):
//     min_{y,t}  t - H(y)   s.t.  G y + h <= t 1        lib/bundle_entropy.py:5-78 (pdipm_pc)
// Mehrotra's predictor-corrector method on the primal-dual system, the solver behind
// `bundle_entropy.solveBatch(..., solver='pc')` -- the module multi-label-cls/icnn_ebundle.py:27-30 and
// completion/icnn_ebundle.py:28-31 import.  One wave64 = one sample, like the dual step:
//   column layout (lane l owns columns l, l+64, ...): y, the residual ry, Hinv = 1/(1/y + 1/(1-y)), dy   (LDS, f64)
//   row layout    (lane i < k owns cut i):           z, s, rd, rc, dz, ds                               (registers)
// M = G Hinv G^T and G Hinv ry come out of the f64 MFMA sweep of the dual step (contract_mfma with w = Hinv and the
// extra column Hinv ry); M + diag(s/z) is symmetric positive definite, and where the reference calls
// np.linalg.cholesky + cho_solve (:42-43,:48) the k x k systems are solved by Gaussian elimination in natural order,
// register resident, lane = matrix row (same solutions up to rounding; oracle/bundle_entropy_oracle.py's restatement
// of the reference and this formulation agree to 2e-10 on every golden problem).
#pragma once

struct Pair {
    double a, b;
    int ok;
};

// Full-wave reductions without the LDS crossbar: __shfl_xor is two ds_bpermute per double and step, six dependent steps --
// about 1.2 k cycles per reduction, and an interior-point iteration does two dozen (round 4: 28 k of its ~45 k cycles).  Here:
// quad permutes, row_half_mirror, row_mirror (DPP, inside a 16-lane row), then v_permlane16_swap / v_permlane32_swap across
// the rows; every lane ends up with the result.  The tree is fixed (lane ^ 1, ^ 2, ^ 7, ^ 15, row pairs, halves).
__device__ __forceinline__ void ipm_swap16(double &x, double &y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void ipm_swap32(double &x, double &y) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
template <typename Op>
__device__ __forceinline__ double wave_reduce_dpp(double v, Op op) {
    v = op(v, dpp_move<0xB1>(v));
    v = op(v, dpp_move<0x4E>(v));
    v = op(v, dpp_move<0x141>(v));
    v = op(v, dpp_move<0x140>(v));
    double a = v, b = v;
    ipm_swap16(a, b);
    v = op(a, b);
    a = v; b = v;
    ipm_swap32(a, b);
    return op(a, b);
}
__device__ __forceinline__ double wave_min(double v) { return wave_reduce_dpp(v, [](double x, double y) { return fmin(x, y); }); }
__device__ __forceinline__ double ipm_sum(double v) { return wave_reduce_dpp(v, [](double x, double y) { return x + y; }); }

// (M + diag) x = ra and (M + diag) x = rb for the k x k matrix in Hm; lane i < k passes its diagonal term and its
// right-hand sides and gets back its components of the two solutions.  ok = 0 on a pivot that is not positive
// (numpy.linalg.cholesky raises LinAlgError there, :42).
template <int KT>
__device__ __noinline__ Pair spd_solve2(const double *Hm_, int HP, int k, double diag, double ra, double rb) {
    const int lane = lane_id();
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k);
    double M[KT];
    const int rl = lane < k ? lane : 0;
#pragma unroll
    for (int j = 0; j < KT; ++j) M[j] = Hm[rl * HP + (j < k ? j : 0)];
#pragma unroll
    for (int j = 0; j < KT; ++j) pin(M[j]);
#pragma unroll
    for (int j = 0; j < KT; ++j)
        M[j] = (lane < k && j < k) ? M[j] + (j == lane ? diag : 0.0) : (j == lane ? 1.0 : 0.0);
    if (!(lane < k)) { ra = 0.0; rb = 0.0; }
    double rinv = 1.0;
    bool bad = false;
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        if (p < k) {
            const double d = bcast(M[p], p);
            bad |= !(d > 0.0);
            const double inv = 1.0 / d;
            rinv = lane == p ? inv : rinv;
            const double f = lane > p ? M[p] * inv : 0.0;
#pragma unroll
            for (int j = p + 1; j < KT; ++j) M[j] -= f * bcast(M[j], p);
            ra -= f * bcast(ra, p);
            rb -= f * bcast(rb, p);
        }
    }
#pragma unroll
    for (int p = KT - 1; p >= 0; --p) {
        if (p < k) {
            const double xa = bcast(ra * rinv, p), xb = bcast(rb * rinv, p);
            ra = lane == p ? xa : (lane < p ? ra - M[p] * xa : ra);
            rb = lane == p ? xb : (lane < p ? rb - M[p] * xb : rb);
        }
    }
    return Pair{ra, rb, __any(bad) ? 0 : 1};        // (results in registers: an output reference would live in scratch)
}

// The same for bundles of up to 16 cuts with DPP row broadcasts (cf. newton_step_dpp): the system sits in lanes 0..15, a pivot
// row reaches the other lanes by one 64-bit DPP move instead of two v_readlane + hazard nops, identity rows (>= k) have pivot
// 1 and multiplier -0 and are processed like any other, so the elimination is straight-line code; 1 / pivot by v_rcp_f64 +
// two Newton-Raphson steps.  Same elimination order as spd_solve2 (natural order, no pivoting).  Round 4: 4384 -> 3466 us
// per 4096 x 10 solve, 1745 -> 1365 us at 128 x 10.
template <int KS>
__device__ __noinline__ Pair spd_solve2_dpp(const double *Hm_, int HP, int k, double diag, double ra, double rb) {
    static_assert(KS <= 16, "row broadcasts stay inside one 16-lane row");
    const int lane = lane_id();
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k);
    double M[KS];
    const int rl = lane < k ? lane : 0;
#pragma unroll
    for (int j = 0; j < KS; ++j) M[j] = Hm[rl * HP + (j < k ? j : 0)];
#pragma unroll
    for (int j = 0; j < KS; ++j) pin(M[j]);
#pragma unroll
    for (int j = 0; j < KS; ++j)
        M[j] = (lane < k && j < k) ? M[j] + (j == lane ? diag : 0.0) : (j == lane ? 1.0 : 0.0);
    if (!(lane < k)) { ra = 0.0; rb = 0.0; }
    double rinv = 1.0;
    bool bad = false;
    static_for<0, KS>([&](auto P) {
        constexpr int p = decltype(P)::value;
        const double d = row_bcast<p>(M[p]);
        bad |= !(d > 0.0);
        const double inv = rcp_nr(d);
        rinv = lane == p ? inv : rinv;
        const double nf = lane > p ? -(M[p] * inv) : 0.0;
        static_for<p + 1, KS>([&](auto J) {
            constexpr int j = decltype(J)::value;
            M[j] = __builtin_fma(nf, row_bcast<p>(M[j]), M[j]);
        });
        ra = __builtin_fma(nf, row_bcast<p>(ra), ra);
        rb = __builtin_fma(nf, row_bcast<p>(rb), rb);
        __builtin_amdgcn_sched_barrier(0);     // keep the broadcasts of later pivots from being hoisted (registers)
    });
    static_for<0, KS>([&](auto Q) {
        constexpr int p = KS - 1 - decltype(Q)::value;
        const double xa = row_bcast<p>(ra * rinv), xb = row_bcast<p>(rb * rinv);
        ra = lane == p ? xa : (lane < p ? __builtin_fma(-M[p], xa, ra) : ra);
        rb = lane == p ? xb : (lane < p ? __builtin_fma(-M[p], xb, rb) : rb);
    });
    return Pair{ra, rb, (__ballot(bad && lane < KS) & 0xffffull) ? 0 : 1};
}
// The corrector's system has the SAME matrix as the predictor's (:41-43 factor once, :48 and :63 solve twice): FACTOR = true
// leaves the elimination's multipliers (below the diagonal), 1 / pivot (diagonal) and upper triangle in F[KS][KS] (LDS, row =
// lane), and spd_resolve_dpp replays them on another right-hand side: KS forward and KS backward steps instead of the
// O(KS^2) elimination.
template <int KS>
__device__ __noinline__ Pair spd_factor2_dpp(const double *Hm_, int HP, int k, double diag, double ra, double rb, double *F_) {
    static_assert(KS <= 16, "row broadcasts stay inside one 16-lane row");
    typedef __attribute__((address_space(3))) double *LdsDbl;
    const int lane = lane_id();
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    LdsDbl F = (LdsDbl)F_;
    HP = uni(HP); k = uni(k);
    double M[KS];
    const int rl = lane < k ? lane : 0;
#pragma unroll
    for (int j = 0; j < KS; ++j) M[j] = Hm[rl * HP + (j < k ? j : 0)];
#pragma unroll
    for (int j = 0; j < KS; ++j) pin(M[j]);
#pragma unroll
    for (int j = 0; j < KS; ++j)
        M[j] = (lane < k && j < k) ? M[j] + (j == lane ? diag : 0.0) : (j == lane ? 1.0 : 0.0);
    if (!(lane < k)) { ra = 0.0; rb = 0.0; }
    bool bad = false;
    static_for<0, KS>([&](auto P) {
        constexpr int p = decltype(P)::value;
        const double d = row_bcast<p>(M[p]);
        bad |= !(d > 0.0);
        const double inv = rcp_nr(d);
        const double nf = lane > p ? -(M[p] * inv) : 0.0;
        static_for<p + 1, KS>([&](auto J) {
            constexpr int j = decltype(J)::value;
            M[j] = __builtin_fma(nf, row_bcast<p>(M[j]), M[j]);
        });
        ra = __builtin_fma(nf, row_bcast<p>(ra), ra);
        rb = __builtin_fma(nf, row_bcast<p>(rb), rb);
        M[p] = lane > p ? nf : (lane == p ? inv : M[p]);       // the factor: multiplier | 1 / pivot | upper triangle
        __builtin_amdgcn_sched_barrier(0);
    });
    if (lane < KS) {
#pragma unroll
        for (int j = 0; j < KS; ++j) F[lane * KS + j] = M[j];
    }
    static_for<0, KS>([&](auto Q) {
        constexpr int p = KS - 1 - decltype(Q)::value;
        const double xa = row_bcast<p>(lane == p ? ra * M[p] : 0.0), xb = row_bcast<p>(lane == p ? rb * M[p] : 0.0);
        ra = lane == p ? xa : (lane < p ? __builtin_fma(-M[p], xa, ra) : ra);
        rb = lane == p ? xb : (lane < p ? __builtin_fma(-M[p], xb, rb) : rb);
    });
    return Pair{ra, rb, (__ballot(bad && lane < KS) & 0xffffull) ? 0 : 1};
}
template <int KS>
__device__ __noinline__ double spd_resolve_dpp(const double *F_, int k, double r) {
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    const int lane = lane_id();
    LdsCDbl F = (LdsCDbl)F_;
    k = uni(k);
    double M[KS];
    const int rl = lane < KS ? lane : 0;
#pragma unroll
    for (int j = 0; j < KS; ++j) M[j] = F[rl * KS + j];
#pragma unroll
    for (int j = 0; j < KS; ++j) pin(M[j]);
    if (!(lane < k)) r = 0.0;
    static_for<0, KS>([&](auto P) {
        constexpr int p = decltype(P)::value;
        r = __builtin_fma(lane > p ? M[p] : 0.0, row_bcast<p>(r), r);
    });
    static_for<0, KS>([&](auto Q) {
        constexpr int p = KS - 1 - decltype(Q)::value;
        const double x = row_bcast<p>(lane == p ? r * M[p] : 0.0);
        r = lane == p ? x : (lane < p ? __builtin_fma(-M[p], x, r) : r);
    });
    return r;
}
// smallest instance that holds k cuts and whose factor fits the scratch (`cap` doubles); 0: none (full solve twice)
__device__ __forceinline__ int spd_factor_size(int k, int cap) {
    const int ks = k <= 4 ? 4 : (k <= 6 ? 6 : (k <= 8 ? 8 : (k <= 10 ? 10 : (k <= 12 ? 12 : (k <= 16 ? 16 : 0)))));
    return ks * ks <= cap ? ks : 0;
}
__device__ __forceinline__ Pair spd_factor2_k(int ks, const double *Hm, int HP, int k, double diag, double ra, double rb, double *F) {
    switch (ks) {
    case 4: return spd_factor2_dpp<4>(Hm, HP, k, diag, ra, rb, F);
    case 6: return spd_factor2_dpp<6>(Hm, HP, k, diag, ra, rb, F);
    case 8: return spd_factor2_dpp<8>(Hm, HP, k, diag, ra, rb, F);
    case 10: return spd_factor2_dpp<10>(Hm, HP, k, diag, ra, rb, F);
    case 12: return spd_factor2_dpp<12>(Hm, HP, k, diag, ra, rb, F);
    default: return spd_factor2_dpp<16>(Hm, HP, k, diag, ra, rb, F);
    }
}
__device__ __forceinline__ double spd_resolve_k(int ks, const double *F, int k, double r) {
    switch (ks) {
    case 4: return spd_resolve_dpp<4>(F, k, r);
    case 6: return spd_resolve_dpp<6>(F, k, r);
    case 8: return spd_resolve_dpp<8>(F, k, r);
    case 10: return spd_resolve_dpp<10>(F, k, r);
    case 12: return spd_resolve_dpp<12>(F, k, r);
    default: return spd_resolve_dpp<16>(F, k, r);
    }
}

template <int KT>
__device__ __forceinline__ Pair spd_solve2_k(const double *Hm, int HP, int k, double diag, double ra, double rb) {
    if (k <= 4) return spd_solve2_dpp<4>(Hm, HP, k, diag, ra, rb);
    if (k <= 6) return spd_solve2_dpp<6>(Hm, HP, k, diag, ra, rb);
    if (k <= 8) return spd_solve2_dpp<8>(Hm, HP, k, diag, ra, rb);
    if (k <= 10) return spd_solve2_dpp<10>(Hm, HP, k, diag, ra, rb);
    if (k <= 12) return spd_solve2_dpp<12>(Hm, HP, k, diag, ra, rb);
    if (KT == 16 || k <= 16) return spd_solve2_dpp<16>(Hm, HP, k, diag, ra, rb);
    return spd_solve2<KT>(Hm, HP, k, diag, ra, rb);
}

// get_step of the reference (:158-163): min over the entries with dv < 0 of -v/dv, and 1 if there is none.  Lanes accumulate
// the per-entry ratios with `ratio_step` (NO_STEP where dv >= 0); the four kinds of entries (z, s, y, 1 - y) share ONE wave
// minimum: the affine step is min(1, all of them) whatever kind is empty, the corrector's 0.99 min(...) needs to know whether
// some kind was empty (its get_step is then 1) -- a ballot per kind.  The quotient is v * (1 / dv) with v_rcp_f64 + two
// Newton-Raphson steps instead of the IEEE division sequence (a step length that differs in the last bit moves nothing: the
// iteration converges to the same point).
constexpr double NO_STEP = 1e300;
__device__ __forceinline__ double ratio_step(double v, double dv) { return dv < 0.0 ? -v * rcp_nr(dv) : NO_STEP; }
// min(ratio_step(y, dy), ratio_step(1 - y, -dy)) with ONE reciprocal: of the two kinds only one has a negative direction in a
// given column (dy < 0: y, dy > 0: 1 - y), and rcp_nr(-d) == -rcp_nr(d) bit for bit, so this is the same number
__device__ __forceinline__ double ratio_step_box(double y, double dy) {
    const double num = dy < 0.0 ? y : 1.0 - y;
    const double q = -num * rcp_nr(dy < 0.0 ? dy : -dy);
    return dy != 0.0 ? q : NO_STEP;
}

// ---- Round 5: the three column passes of an interior-point iteration as K-templated, fully unrolled functions --------------
// One wave per sample on float32 rows of up to 192 columns (three per lane) and bundles of 2 .. IPM_KMAX cuts.  The loops they
// replace walked the columns and, inside, the cuts one LDS read at a time (`cols_dot`: k dependent read - broadcast - fma
// round trips per column and pass, three passes per iteration: ~15 k of the iteration's 26 k cycles were LDS latency).  Here a
// pass reads its 3 K bundle entries and its column values in one batch, keeps the lane's three columns in registers and
//   pass 1  forms the residual ry, Hinv = y (1 - y) AND, in the same registers, every sum over the columns the iteration
//           needs: M = G Hinv G^T (K (K + 1) / 2 values, the fused pass of be_dual_valu_dev.h: same per-entry summation tree,
//           same bits as hv_weighted_pass_fn), G Hinv ry (K), G y (K: fresh every iteration, so no carried G y and no k
//           wave reductions over the columns) and |ry|^2 (1) -- one transposing butterfly for all of them;
//   pass 2  the affine dy and its step ratios;   pass 3  the corrector's dy, its ratios and the sign flags.
// Hinv is re-formed from y where it is needed (two operations) instead of being stored and re-read.
constexpr int IPM_KMAX = 12;       // one wave per sample (the persistent kernels: more instances cost them registers, +12 % measured)
// (IPM_KMAX_WAVES = 20, the most cuts ipm_solve_waves takes: be_dual_dev.h, next to the LDS carve-up that depends on it)
__host__ __device__ constexpr int ipm_nv(int K) { return K * (K + 1) / 2 + 2 * K + 1; }
__host__ __device__ constexpr int ipm_chunk(int K) { return hv_chunk_len(ipm_nv(K), 24); }

template <typename CutT, int K, int E0, int EN, int NV, typename PP>
__device__ __forceinline__ void ipm_chunks(const CutT (&av)[K][3], const double (&w)[3], const double (&zz)[3], const double (&yy)[3],
                                           const double (&ry)[3], int lane, PP Pw) {
    if constexpr (E0 < NV) {
        constexpr int N = E0 + EN <= NV ? EN : NV - E0, T = K * (K + 1) / 2;
        double v[N];
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = 0.0;
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            double ad[K];
#pragma unroll
            for (int i = 0; i < K; ++i) ad[i] = (double)av[i][c];
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const int ge = E0 + e;                                // (compile-time after unrolling)
                if (ge < T) v[e] = __builtin_fma(ad[hv_wrow(ge)], ad[hv_wcol(ge)] * w[c], v[e]);      // as hv_chunks
                else if (ge < T + K) v[e] = __builtin_fma(ad[ge - T < K ? ge - T : 0], zz[c], v[e]);
                else if (ge < T + 2 * K) v[e] = __builtin_fma(ad[ge - T - K >= 0 && ge - T - K < K ? ge - T - K : 0], yy[c], v[e]);
                else v[e] = __builtin_fma(ry[c], ry[c], v[e]);
            }
        }
        hv_transpose_reduce<N>(v, lane);
        const int idx = hv_index(N, lane);
        if (idx >= 0) Pw[E0 + idx] = v[0];
        ipm_chunks<CutT, K, E0 + EN, EN, NV>(av, w, zz, yy, ry, lane, Pw);
    }
}

// the lane's three columns of the staged bundle (rows >= k of a padded instance: zero by a select on an in-bounds read)
template <typename CutT, int K, typename AP>
__device__ __forceinline__ void ipm_load_columns(AP As, int ldA, int k, const int (&jc)[3], CutT (&av)[K][3]) {
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const CutT v = As[(i < k ? i : 0) * ldA + jc[c]];
            av[i][c] = i < k ? v : (CutT)0;
        }
}

template <typename CutT, int K>
__device__ __noinline__ void ipm_pass1_fn(const CutT *As_, int ldA, int k, int n, int n_pad, const double *yv_, double *rys_,
                                          double *Pw_, double z) {
    typedef const __attribute__((address_space(3))) CutT *LdsCut;
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    LdsCut As = (LdsCut)As_;
    LdsCDbl yv = (LdsCDbl)yv_;
    LdsDbl rys = (LdsDbl)rys_, Pw = (LdsDbl)Pw_;
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad);
    const int lane = lane_id();
    int jc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) jc[c] = lane + 64 * c < n_pad ? lane + 64 * c : n_pad - 1;
    CutT av[K][3];
    ipm_load_columns<CutT, K>(As, ldA, k, jc, av);
    double y[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) y[c] = yv[jc[c]];
    double zi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) zi[i] = bcast(z, i);                  // (z is 0 in the lanes beyond the bundle)
    double w[3], zz[3], yy[3], ry[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int j = lane + 64 * c;
        const double grad = fast_log(y[c] * rcp_nr(1.0 - y[c]));           // :17  log y - log(1 - y)
        const double hinv = y[c] * (1.0 - y[c]);                      // :19  1 / (1/y + 1/(1-y))
        double gz = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) gz += zi[i] * (double)av[i][c];   // (G^T z)_j, cuts in order (absent rows: + 0 * 0)
        const double r = j < n ? grad + gz : 0.0;
        if (j < n_pad) rys[j] = r;
        ry[c] = r;
        w[c] = j < n ? hinv : 0.0;
        zz[c] = j < n ? hinv * r : 0.0;
        yy[c] = j < n ? y[c] : 0.0;
    }
    ipm_chunks<CutT, K, 0, ipm_chunk(K), ipm_nv(K)>(av, w, zz, yy, ry, lane, Pw);
}

// dy = -Hinv (ry + G^T dz) (:50) -> dyv; returns the lane's smallest step ratio over y and 1 - y (NO_STEP: none)
template <typename CutT, int K>
__device__ __noinline__ double ipm_pass2_fn(const CutT *As_, int ldA, int k, int n, int n_pad, const double *yv_,
                                            const double *rys_, double *dyv_, double dz) {
    typedef const __attribute__((address_space(3))) CutT *LdsCut;
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    LdsCut As = (LdsCut)As_;
    LdsCDbl yv = (LdsCDbl)yv_, rys = (LdsCDbl)rys_;
    LdsDbl dyv = (LdsDbl)dyv_;
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad);
    const int lane = lane_id();
    int jc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) jc[c] = lane + 64 * c < n_pad ? lane + 64 * c : n_pad - 1;
    CutT av[K][3];
    ipm_load_columns<CutT, K>(As, ldA, k, jc, av);
    double y[3], ry[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { y[c] = yv[jc[c]]; ry[c] = rys[jc[c]]; }
    double di[K];
#pragma unroll
    for (int i = 0; i < K; ++i) di[i] = bcast(dz, i);
    double m = NO_STEP;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int j = lane + 64 * c;
        const double hinv = j < n ? y[c] * (1.0 - y[c]) : 0.0;
        double gd = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) gd += di[i] * (double)av[i][c];
        const double dy = -hinv * (ry[c] + gd);
        if (j < n_pad) dyv[j] = dy;
        if (j < n) m = fmin(m, ratio_step_box(y[c], dy));
    }
    return m;
}

// dy -= Hinv G^T dz_c (:66) -> dyv; the lane's smallest ratio and whether it saw a negative / a positive direction
struct IpmStep { double m; int neg; };
template <typename CutT, int K>
__device__ __noinline__ IpmStep ipm_pass3_fn(const CutT *As_, int ldA, int k, int n, int n_pad, const double *yv_, double *dyv_,
                                             double dz) {
    typedef const __attribute__((address_space(3))) CutT *LdsCut;
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    LdsCut As = (LdsCut)As_;
    LdsCDbl yv = (LdsCDbl)yv_;
    LdsDbl dyv = (LdsDbl)dyv_;
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad);
    const int lane = lane_id();
    int jc[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) jc[c] = lane + 64 * c < n_pad ? lane + 64 * c : n_pad - 1;
    CutT av[K][3];
    ipm_load_columns<CutT, K>(As, ldA, k, jc, av);
    double y[3], d0[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) { y[c] = yv[jc[c]]; d0[c] = dyv[jc[c]]; }
    double di[K];
#pragma unroll
    for (int i = 0; i < K; ++i) di[i] = bcast(dz, i);
    double m = NO_STEP;
    int neg = 0;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
        const int j = lane + 64 * c;
        const double hinv = j < n ? y[c] * (1.0 - y[c]) : 0.0;
        double gd = 0.0;
#pragma unroll
        for (int i = 0; i < K; ++i) gd += di[i] * (double)av[i][c];
        const double dy = d0[c] - hinv * gd;
        if (j < n_pad) dyv[j] = dy;
        if (j < n) {
            m = fmin(m, ratio_step_box(y[c], dy));
            neg |= (dy < 0.0 ? 1 : 0) | (dy > 0.0 ? 2 : 0);
        }
    }
    return IpmStep{m, neg};
}

// ---- the same passes for WIDE rows (n_pad > 192: the completion model's 2048 pixels on one wave): column chunks of 192 ------
// A lane walks its columns three at a time (lane + 64 c + 192 chunk), every chunk one batch of LDS reads.  Pass 1 in two sweeps:
// (a) residual, Hinv, Hinv ry per column -> rys / ws / zs; (b) the sums -- per run of at most 24 values one sweep over the
// chunks with the bundle columns and the column values re-read (the accumulators of ALL sums would not fit the registers), ONE
// transposing butterfly per run -> Pw (which must not alias ws / zs / rys / yv).
// GSRC: the bundle rows are read from device memory (a round whose bundle is staged in st->scratch) instead of LDS
template <typename CutT, int K, bool GSRC = false>
__device__ __noinline__ void ipm_wide1a_fn(const CutT *As_, int ldA, int k, int n, int n_pad, const double *yv_, double *rys_,
                                           double *ws_, double *zs_, double z, int c_first, int c_step) {
    typedef std::conditional_t<GSRC, const __attribute__((address_space(1))) CutT *, const __attribute__((address_space(3))) CutT *> LdsCut;
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    LdsCut As = (LdsCut)As_;
    LdsCDbl yv = (LdsCDbl)yv_;
    LdsDbl rys = (LdsDbl)rys_, ws = (LdsDbl)ws_, zs = (LdsDbl)zs_;
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad); c_first = uni(c_first); c_step = uni(c_step);
    const int lane = lane_id();
    double zi[K];
#pragma unroll
    for (int i = 0; i < K; ++i) zi[i] = bcast(z, i);
    for (int c0 = c_first; c0 < n_pad; c0 += c_step) {
        int jc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) jc[c] = c0 + lane + 64 * c < n_pad ? c0 + lane + 64 * c : n_pad - 1;
        CutT av[K][3];
        ipm_load_columns<CutT, K>(As, ldA, k, jc, av);
        double y[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) y[c] = yv[jc[c]];
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int j = c0 + lane + 64 * c;
            const double grad = fast_log(y[c] * rcp_nr(1.0 - y[c]));
            const double hinv = y[c] * (1.0 - y[c]);
            double gz = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) gz += zi[i] * (double)av[i][c];
            const double r = j < n ? grad + gz : 0.0;
            if (j < n_pad) {
                rys[j] = r;
                ws[j] = j < n ? hinv : 0.0;
                zs[j] = j < n ? hinv * r : 0.0;
            }
        }
    }
}

template <typename CutT, int K, int E0, int EN, int NV, typename AP, typename CP, typename PP>
__device__ __forceinline__ void ipm_wide_sums(AP As, int ldA, int k, int n, int n_pad, CP ws, CP zs, CP yv, CP rys, int lane, PP Pw,
                                              int c_first, int c_step) {
    if constexpr (E0 < NV) {
        constexpr int N = E0 + EN <= NV ? EN : NV - E0, T = K * (K + 1) / 2;
        double v[N];
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = 0.0;
        for (int c0 = c_first; c0 < n_pad; c0 += c_step) {
            int jc[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) jc[c] = c0 + lane + 64 * c < n_pad ? c0 + lane + 64 * c : n_pad - 1;
            CutT av[K][3];
            ipm_load_columns<CutT, K>(As, ldA, k, jc, av);
            double w[3], zz[3], yy[3], ry[3];
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                const bool in = c0 + lane + 64 * c < n;          // (beyond n: the stored values are zeros, y is not)
                w[c] = ws[jc[c]]; zz[c] = zs[jc[c]]; ry[c] = rys[jc[c]];
                const double yr = yv[jc[c]];
                if (!(c0 + lane + 64 * c < n_pad)) { w[c] = 0.0; zz[c] = 0.0; ry[c] = 0.0; }     // clamped re-reads
                yy[c] = in ? yr : 0.0;
            }
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                double ad[K];
#pragma unroll
                for (int i = 0; i < K; ++i) ad[i] = (double)av[i][c];
#pragma unroll
                for (int e = 0; e < N; ++e) {
                    const int ge = E0 + e;
                    if (ge < T) v[e] = __builtin_fma(ad[hv_wrow(ge)], ad[hv_wcol(ge)] * w[c], v[e]);
                    else if (ge < T + K) v[e] = __builtin_fma(ad[ge - T < K ? ge - T : 0], zz[c], v[e]);
                    else if (ge < T + 2 * K) v[e] = __builtin_fma(ad[ge - T - K >= 0 && ge - T - K < K ? ge - T - K : 0], yy[c], v[e]);
                    else v[e] = __builtin_fma(ry[c], ry[c], v[e]);
                }
            }
        }
        hv_transpose_reduce<N>(v, lane);
        const int idx = hv_index(N, lane);
        if (idx >= 0) Pw[E0 + idx] = v[0];
        ipm_wide_sums<CutT, K, E0 + EN, EN, NV>(As, ldA, k, n, n_pad, ws, zs, yv, rys, lane, Pw, c_first, c_step);
    }
}
template <typename CutT, int K, bool GSRC = false>
__device__ __noinline__ void ipm_wide1b_fn(const CutT *As_, int ldA, int k, int n, int n_pad, const double *ws_, const double *zs_,
                                           const double *yv_, const double *rys_, double *Pw_, int c_first, int c_step) {
    typedef std::conditional_t<GSRC, const __attribute__((address_space(1))) CutT *, const __attribute__((address_space(3))) CutT *> LdsCut;
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad); c_first = uni(c_first); c_step = uni(c_step);
    ipm_wide_sums<CutT, K, 0, ipm_chunk(K), ipm_nv(K)>((LdsCut)As_, ldA, k, n, n_pad, (LdsCDbl)ws_, (LdsCDbl)zs_, (LdsCDbl)yv_,
                                                       (LdsCDbl)rys_, lane_id(), (LdsDbl)Pw_, c_first, c_step);
}

// passes 2 and 3 over column chunks (FIRST: the affine dy = -Hinv (ry + G^T dz); else dy -= Hinv G^T dz and the sign flags)
template <typename CutT, int K, bool FIRST, bool GSRC = false>
__device__ __noinline__ IpmStep ipm_wide23_fn(const CutT *As_, int ldA, int k, int n, int n_pad, const double *yv_,
                                              const double *rys_, double *dyv_, double dz, int c_first, int c_step) {
    typedef std::conditional_t<GSRC, const __attribute__((address_space(1))) CutT *, const __attribute__((address_space(3))) CutT *> LdsCut;
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    LdsCut As = (LdsCut)As_;
    LdsCDbl yv = (LdsCDbl)yv_, rys = (LdsCDbl)rys_;
    LdsDbl dyv = (LdsDbl)dyv_;
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad);
    const int lane = lane_id();
    double di[K];
#pragma unroll
    for (int i = 0; i < K; ++i) di[i] = bcast(dz, i);
    double m = NO_STEP;
    int neg = 0;
    c_first = uni(c_first); c_step = uni(c_step);
    for (int c0 = c_first; c0 < n_pad; c0 += c_step) {
        int jc[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) jc[c] = c0 + lane + 64 * c < n_pad ? c0 + lane + 64 * c : n_pad - 1;
        CutT av[K][3];
        ipm_load_columns<CutT, K>(As, ldA, k, jc, av);
        double y[3], b0[3];
#pragma unroll
        for (int c = 0; c < 3; ++c) { y[c] = yv[jc[c]]; b0[c] = FIRST ? rys[jc[c]] : dyv[jc[c]]; }
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const int j = c0 + lane + 64 * c;
            const double hinv = j < n ? y[c] * (1.0 - y[c]) : 0.0;
            double gd = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i) gd += di[i] * (double)av[i][c];
            const double dy = FIRST ? -hinv * (b0[c] + gd) : b0[c] - hinv * gd;
            if (j < n_pad) dyv[j] = dy;
            if (j < n) {
                m = fmin(m, ratio_step_box(y[c], dy));
                neg |= (dy < 0.0 ? 1 : 0) | (dy > 0.0 ? 2 : 0);
            }
        }
    }
    return IpmStep{m, neg};
}

#define IPM_K_SWITCH(kk, CALL)                                                                                     \
    switch (hv_padded(kk)) {                                                                                       \
    case 1: case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; case 5: CALL(5); break;        \
    case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; case 10: CALL(10); break;             \
    default: CALL(12); break;                                                                                      \
    }
#define IPM_K_SWITCH_WAVES(kk, CALL)                                                                               \
    switch (hv_padded(kk)) {                                                                                       \
    case 1: case 2: CALL(2); break; case 3: CALL(3); break; case 4: CALL(4); break; case 5: CALL(5); break;        \
    case 6: CALL(6); break; case 7: CALL(7); break; case 8: CALL(8); break; case 10: CALL(10); break;             \
    case 12: CALL(12); break; case 14: CALL(14); break; case 16: CALL(16); break; case 18: CALL(18); break;       \
    default: CALL(20); break;                                                                                      \
    }

// wide float32 rows on ONE wave (n_pad > 192 where dual_waves() keeps a single wave: fewer than 1024 columns): the NW = 1 instance
// of ipm_solve_waves (below) as a function of its own, so that the kernels of narrow rows only carry a call
struct IpmOut { double z; int status; };
template <typename CutT, int KT, bool GSRC>
__device__ __noinline__ IpmOut ipm_solve_wide_one_fn(const CutT *As, int ldA, int k, int n, int n_pad, double *ws, double *zs, double *rys,
                                                    double *yv, double *dyv, double *Hm, int HP, double h_i);

// Runs pdipm_pc on the k staged cuts (rows of As, offsets h_i in row layout).  On return yv[0..n) holds y (LDS) and the
// result is this lane's multiplier z_i (0 beyond k).  *status: 0 ok, 1 = M not positive definite / non-finite.
//
// Round 4 (tools/dual_phase_profile.py pdipm: 35 k cycles per interior-point iteration, ~6.7 iterations per round): the same
// iteration with fewer instructions where the result does not depend on them to more than rounding --
//   * Hinv = 1 / (1/y + 1/(1-y)) = y (1 - y), grad = log y - log(1 - y) = log(y / (1 - y)): one logarithm, one reciprocal
//     instead of two logarithms and three divisions per column;
//   * G y is carried from iteration to iteration: y moves by alpha dy and G dy = -(G Hinv ry) - M dz is a k x k product of
//     quantities that are already there (M = G Hinv G^T and G Hinv ry come out of the MFMA sweep), instead of k wave
//     reductions over the columns;
//   * sums over the multipliers (row layout, lanes < k) by the DPP row reduction of the dual step, one merged wave minimum
//     for the step lengths.
template <typename CutT, int KT, bool GLB_ROWS = false, typename LapF = NoLap>      // GLB_ROWS: the bundle is staged in device memory
__device__ __forceinline__ double ipm_solve(const CutT *As, int ldA, int k, const CutT *crow, int n, int n_pad,
                                            double *ws, double *zs, double *rys, double *yv, double *dyv, double *Hm,
                                            int HP, double h_i, int lane, int *status, LapF lap = LapF()) {
    const bool row = lane < k;
    // (GLB_ROWS: the weighted pass has no device-memory source instance -- a bundle staged in st->scratch keeps the MFMA sweep,
    //  whose sums agree with the pass to rounding, not bit for bit: include/icnn_be.h, ICNN_BE_FLAG_GLOBAL_BUNDLE)
    const bool hv_ok = sizeof(CutT) == 4 && n_pad <= 192 && !GLB_ROWS;
    // round 5: the unrolled passes above (bundle in LDS, float32 rows of up to 192 columns, 2 .. IPM_KMAX cuts)
    const bool fast = hv_ok && k >= 2 && k <= IPM_KMAX && ipm_nv(hv_padded(k)) <= n_pad;
    const int KP = hv_padded(k), TP = KP * (KP + 1) / 2;
    if (sizeof(CutT) == 4 && n_pad > 192 && k <= IPM_KMAX_WAVES && ((ipm_nv(k < 2 ? 2 : KP) + 3) & ~3) <= n_pad) {
        // wide float32 rows on one wave: the column-chunked passes (bundle in LDS or, GLB_ROWS, in device memory)
        if constexpr (sizeof(CutT) == 4) {
            const IpmOut o = ipm_solve_wide_one_fn<CutT, KT, GLB_ROWS>(As, ldA, k, n, n_pad, ws, zs, rys, yv, dyv, Hm, HP, h_i);
            *status = uni(o.status);
            return o.z;
        }
    }
    double z = row ? 1.0 / (double)k : 0.0;                    // :11
    double s = row ? 1.0 : 0.0;                                // :13
    double t = 1.0;                                            // :14
    for (int j = lane; j < n_pad; j += 64) yv[j] = 0.5;        // :12
    sample_sync<1>();
    *status = 0;
    auto cols_dot = [&](double v, int j) -> double {           // (G^T v)_j, v in row layout
        double acc = 0.0;
        for (int i = 0; i < k; ++i) acc += bcast(v, i) * (double)As[i * ldA + j];
        return acc;
    };
    const auto add = [](double x, double y) { return x + y; };
    auto rsum = [&](double v) -> double { return rows_reduce<KT>(row ? v : 0.0, k, add); };
    auto rows_dot = [&](const double *vec) -> double {        // (G vec)_i for lane i
        double mine = 0.0;
        for (int i = 0; i < k; ++i) {
            double part = 0.0;
            for (int j = lane; j < n_pad; j += 64) part += (double)As[i * ldA + j] * vec[j];
            part = ipm_sum(part);
            if (lane == i) mine = part;
        }
        return mine;
    };
    // G y: from the columns at the start point and again whenever the residuals are small enough that the stopping test
    // (:39, 1e-8) is in sight; the carried value only serves the iterations that are far from it (its drift is ~1e-15 per
    // iteration).  The decision to stop is always made on a freshly computed G y: an iteration that falls from >= 1e-4 to
    // < 1e-8 in one step passes the test on the carried value first and is re-tested on a fresh one (below).
    double gy = 0.0, near = 1.0;
    for (int it = 0; it < 20; ++it) {                          // :16
        const bool fresh = fast || it == 0 || near < 1e-4;
        if (fresh && !fast) gy = rows_dot(yv);
        // residuals (:26-29)
        double pri2 = 0.0;
        if (fast) {
            // ry -> rys; M | G Hinv ry | G y | |ry|^2 -> zs (packed: triangle, then the three runs), then H | G Hinv ry -> Hm
#define IPM_P1(KK) ipm_pass1_fn<CutT, KK>(As, ldA, k, n, n_pad, yv, rys, zs, z)
            IPM_K_SWITCH(k, IPM_P1)
#undef IPM_P1
            sample_sync<1>();
            hv_gather<1, true>(zs, 0, Hm, HP, k, lane, 64, hv_entry<true>(lane, k, HP));
            gy = row ? zs[TP + KP + lane] : 0.0;
            pri2 = zs[TP + 2 * KP];
        } else
        for (int j = lane; j < n_pad; j += 64) {
            const double y = yv[j];
            const double grad = fast_log(y * rcp_nr(1.0 - y));      // :17  log y - log(1 - y)
            const double hinv = y * (1.0 - y);                 // :19  1 / (1/y + 1/(1-y))
            const double ry = j < n ? grad + cols_dot(z, j) : 0.0;
            rys[j] = ry;
            ws[j] = j < n ? hinv : 0.0;
            zs[j] = j < n ? hinv * ry : 0.0;
            pri2 += ry * ry;
        }
        sample_sync<1>();
        lap(4);                                                // (diagnostic laps: tools/dual_phase_profile.py, variant pdipm)
        const double rt = 1.0 - rsum(z);                       // :27
        double rd = row ? gy + h_i - t + s : 0.0;              // :29
        const double pri_res = sqrt((fast ? pri2 : ipm_sum(pri2)) + rt * rt);
        double dual_res = sqrt(rsum(rd * rd));
        lap(8);
        if (pri_res < 1e-8 && dual_res < 1e-8) {               // :39
            if (fresh) break;
            gy = rows_dot(yv);                                 // passed on the carried G y: decide on a fresh one
            rd = row ? gy + h_i - t + s : 0.0;
            dual_res = sqrt(rsum(rd * rd));
            if (dual_res < 1e-8) break;
        }
        near = fmax(pri_res, dual_res);
        // M = G Hinv G^T (+ diag(s/z) below) and G Hinv ry in one MFMA sweep (:41, :46)
        // (round 4: bundles of up to 8 cuts of float32 rows of up to 192 columns by the fused VALU pass -- no operand
        //  gathers --, like the dual variant's Newton update; the sums land in zs, which the pass has read by then)
        if (fast) {
            // (the sums came out of pass 1)
        } else if (hv_ok && k >= 2 && k <= HV_K1MAX && hv_pitch(k) <= n_pad) {
            hv_weighted_pass_k<CutT>(As, ldA, k, n, n_pad, ws, zs, zs);
            sample_sync<1>();
            hv_gather<1, true>(zs, hv_pitch(k), Hm, HP, k, lane, 64, hv_entry<true>(lane, k, HP));
        } else {
            contract_mfma<CutT, KT, true>(As, ldA, k, crow, 0, n_pad, ws, zs, Hm, HP);
        }
        sample_sync<1>();
        lap(5);
        const double soz = row ? s * rcp_nr(z) : 1.0;          // (reciprocal + two Newton steps, like every quotient of this loop: measured in
                                                               //  round 5, IEEE division here moves no digit of lam on any golden -- lse_n33 stays at
                                                               //  5.47e-6 -- and costs ~100 instructions per iteration where the phase is issue-bound)
        const double ghr = row ? Hm[lane * HP + k] : 0.0;
        // affine direction (:53): r = rd - G Hinv ry - (s/z) rc with rc = z
        const double r = rd - ghr - soz * z;
        // (the factor of M + diag(s/z) goes to zs: G Hinv ry has been consumed by the sweep, the buffer is free until the
        //  next iteration's residual pass)
        const int fks = spd_factor_size(k, n_pad);
        const Pair um = fks ? spd_factor2_k(fks, Hm, HP, k, soz, r, 1.0, zs) : spd_solve2_k<KT>(Hm, HP, k, soz, r, 1.0);
        if (!uni(um.ok) || !isfinite(pri_res)) { *status = 1; break; }
        lap(9);
        const double m1 = row ? um.b : 0.0, m1inv = rcp_nr(rsum(m1));
        const double dt_a = (rsum(r * m1) - rt) * m1inv;
        const double dz_a = row ? um.a - dt_a * m1 : 0.0;      // = M^-1 (r - dt), :48
        const double ds_a = -soz * (z + dz_a);                 // :49
        double mall = row ? fmin(ratio_step(z, dz_a), ratio_step(s, ds_a)) : NO_STEP;
        if (fast) {
#define IPM_P2(KK) mall = fmin(mall, ipm_pass2_fn<CutT, KK>(As, ldA, k, n, n_pad, yv, rys, dyv, dz_a))
            IPM_K_SWITCH(k, IPM_P2)
#undef IPM_P2
        } else
        for (int j = lane; j < n_pad; j += 64) {
            const double dy = -ws[j] * (rys[j] + cols_dot(dz_a, j));   // :50
            dyv[j] = dy;
            if (j < n) mall = fmin(mall, fmin(ratio_step(yv[j], dy), ratio_step(1.0 - yv[j], -dy)));
        }
        double alpha = fmin(wave_min(mall), 1.0);              // :55-56
        lap(6);
        const double sz = rsum(s * z);
        const double q = rsum((s + alpha * ds_a) * (z + alpha * dz_a)) * rcp_nr(sz);
        const double sig = q * q * q;                          // :57
        const double mu = sz / (double)k;                      // :59
        // corrector (:61-63): ry = rt = rd = 0, rc = -(mu sig - ds_aff dz_aff) / s
        const double rc2 = row ? -(mu * sig - ds_a * dz_a) * rcp_nr(s) : 0.0;
        const double r2 = -(soz * rc2);
        sample_sync<1>();
        const double u2a = fks ? spd_resolve_k(fks, zs, k, r2) : spd_solve2_k<KT>(Hm, HP, k, soz, r2, 0.0).a;
        lap(10);
        const double dt_c = rsum(r2 * m1) * m1inv;
        const double dz_c = row ? u2a - dt_c * m1 : 0.0;
        const double ds_c = -soz * (rc2 + dz_c);
        const double dz = dz_a + dz_c, ds = ds_a + ds_c, dt = dt_a + dt_c;   // :65-68
        mall = row ? fmin(ratio_step(s, ds), ratio_step(z, dz)) : NO_STEP;
        bool neg_y = false, neg_1y = false;                    // does the kind have an entry with a negative direction?
        if (fast) {
            IpmStep st3{NO_STEP, 0};
#define IPM_P3(KK) st3 = ipm_pass3_fn<CutT, KK>(As, ldA, k, n, n_pad, yv, dyv, dz_c)
            IPM_K_SWITCH(k, IPM_P3)
#undef IPM_P3
            mall = fmin(mall, st3.m);
            neg_y = (st3.neg & 1) != 0;
            neg_1y = (st3.neg & 2) != 0;
        } else
        for (int j = lane; j < n_pad; j += 64) {
            const double dy = dyv[j] - ws[j] * cols_dot(dz_c, j);
            dyv[j] = dy;
            if (j < n) {
                mall = fmin(mall, fmin(ratio_step(yv[j], dy), ratio_step(1.0 - yv[j], -dy)));
                neg_y |= dy < 0.0;
                neg_1y |= dy > 0.0;
            }
        }
        // min over the four get_step values: a kind without a negative direction contributes 1 (:158-163)
        const bool some_empty = !__any(row && ds < 0.0) || !__any(row && dz < 0.0) || !__any(neg_y) || !__any(neg_1y);
        const double mwave = wave_min(mall);
        const double gmin = some_empty ? fmin(mwave, 1.0) : mwave;
        alpha = fmax(0.0, fmin(1.0, 0.99 * gmin));             // :70-71
        for (int j = lane; j < n; j += 64) yv[j] += alpha * dyv[j];       // :73
        // G (y + alpha dy) = G y + alpha G dy,  G dy = -(G Hinv ry) - M dz  (dy = -Hinv (ry + G^T dz), M = G Hinv G^T in Hm)
        if (!fast) {                                           // (pass 1 forms G y from the columns in every iteration)
            double mdz = 0.0;
            for (int j = 0; j < k; ++j) mdz += Hm[(row ? lane : 0) * HP + j] * bcast(dz, j);
            gy += alpha * (-ghr - mdz);
        }
        t += alpha * dt;                                       // :74
        s += alpha * ds;                                       // :75
        z += alpha * dz;                                       // :76
        sample_sync<1>();
        lap(11);
    }
    return row ? z : 0.0;
}

// ---- wide rows on NW waves (round 5): the interior-point solve of a sample whose columns are split over the waves of its
// workgroup, like the dual variant's wide rows (dual_step_body, NW = 8 at n >= 1024).  Column chunks of 192 are dealt round-robin
// to the waves; every wave reduces its own partial sums (one transposing butterfly per run of values) into its row of `part`,
// and adds up the NW rows in wave order into its OWN copy Hq of the k x (k + 1) system -- the row-layout algebra (residuals, the
// SPD factorisation, both solves, step lengths) runs redundantly and identically in every wave, so no multiplier is ever
// exchanged; what IS exchanged per iteration: the partial sums, two step-length minima and the two sign flags (xch: two
// doubles per wave).  k = 1 takes the two-cut instance with an empty second row.  Same iteration as ipm_solve; the sums are
// formed in another order (per wave, then over the waves), so results agree with the one-wave path to rounding.
template <typename CutT, int KT, int NW, bool GSRC = false, typename LapF = NoLap>
__device__ __forceinline__ double ipm_solve_waves(const CutT *As, int ldA, int k, int n, int n_pad, double *ws, double *zs, double *rys,
                                                  double *yv, double *dyv, double *Hq, int HP, double h_i, int tid, int *status,
                                                  LapF lap = LapF()) {
    constexpr int NT = 64 * NW;
    const int lane = tid & 63, wave = uni(tid >> 6);
    const bool row = lane < k;
    const int KP = k < 2 ? 2 : hv_padded(k), TP = KP * (KP + 1) / 2, NVP = (ipm_nv(KP) + 3) & ~3;
    const int c_first = wave * 192, c_step = 192 * NW;
    double *part = dyv;                                        // [NW][NVP] partial sums (dyv is free until pass 2)
    double *xch = ws;                                          // [2 NW]   (ws is free between sweep (b) and the next sweep (a))
    double z = row ? 1.0 / (double)k : 0.0, s = row ? 1.0 : 0.0, t = 1.0;
    for (int j = tid; j < n_pad; j += NT) yv[j] = 0.5;
    sample_sync<NW>();
    *status = 0;
    const auto add = [](double x, double y) { return x + y; };
    auto rsum = [&](double v) -> double { return rows_reduce<KT>(row ? v : 0.0, k, add); };
    for (int it = 0; it < 20; ++it) {
#define IPM_M1A(KK) ipm_wide1a_fn<CutT, KK, GSRC>(As, ldA, k, n, n_pad, yv, rys, ws, zs, z, c_first, c_step)
        IPM_K_SWITCH_WAVES(k, IPM_M1A)
#undef IPM_M1A
        sample_sync<NW>();
#define IPM_M1B(KK) ipm_wide1b_fn<CutT, KK, GSRC>(As, ldA, k, n, n_pad, ws, zs, yv, rys, part + wave * NVP, c_first, c_step)
        IPM_K_SWITCH_WAVES(k, IPM_M1B)
#undef IPM_M1B
        sample_sync<NW>();
        lap(4);
        double gy = 0.0, pri2 = 0.0;
        if (k >= 2) {
            hv_gather<NW, true>(part, NVP, Hq, HP, k, lane, 64, hv_entry<true>(lane, k, HP));
        } else {                                               // k = 1 in the two-cut layout: M at 0, G Hinv ry at TP
            double m = 0.0, g = 0.0;
#pragma unroll
            for (int w = 0; w < NW; ++w) { m += part[w * NVP]; g += part[w * NVP + TP]; }
            if (lane == 0) { Hq[0] = m; Hq[1] = g; }
            sample_sync<1>();
        }
#pragma unroll
        for (int w = 0; w < NW; ++w) {
            gy += row ? part[w * NVP + TP + KP + lane] : 0.0;
            pri2 += part[w * NVP + TP + 2 * KP];
        }
        sample_sync<NW>();                                     // every wave has read the partial sums: dyv, ws may be rewritten
        lap(5);
        const double rt = 1.0 - rsum(z);                       // :27
        const double rd = row ? gy + h_i - t + s : 0.0;        // :29
        const double pri_res = sqrt(pri2 + rt * rt), dual_res = sqrt(rsum(rd * rd));
        lap(8);
        if (pri_res < 1e-8 && dual_res < 1e-8) break;          // :39 (identical in every wave)
        const double soz = row ? s * rcp_nr(z) : 1.0;
        const double ghr = row ? Hq[lane * HP + k] : 0.0;
        const double r = rd - ghr - soz * z;
        // (the factor goes to zs: every wave writes the same numbers to the same places, and reads them back)
        const int fks = spd_factor_size(k, n_pad);
        const Pair um = fks ? spd_factor2_k(fks, Hq, HP, k, soz, r, 1.0, zs) : spd_solve2_k<KT>(Hq, HP, k, soz, r, 1.0);
        if (!uni(um.ok) || !isfinite(pri_res)) { *status = 1; break; }
        lap(9);
        const double m1 = row ? um.b : 0.0, m1inv = rcp_nr(rsum(m1));
        const double dt_a = (rsum(r * m1) - rt) * m1inv;
        const double dz_a = row ? um.a - dt_a * m1 : 0.0;
        const double ds_a = -soz * (z + dz_a);
        double mall = row ? fmin(ratio_step(z, dz_a), ratio_step(s, ds_a)) : NO_STEP;
#define IPM_M2(KK) mall = fmin(mall, ipm_wide23_fn<CutT, KK, true, GSRC>(As, ldA, k, n, n_pad, yv, rys, dyv, dz_a, c_first, c_step).m)
        IPM_K_SWITCH_WAVES(k, IPM_M2)
#undef IPM_M2
        {
            const double wm = wave_min(mall);
            if (lane == 0) xch[wave] = wm;
            sample_sync<NW>();
            mall = xch[0];
#pragma unroll
            for (int w = 1; w < NW; ++w) mall = fmin(mall, xch[w]);
            sample_sync<NW>();
        }
        double alpha = fmin(mall, 1.0);                        // :55-56
        lap(6);
        const double sz = rsum(s * z);
        const double q = rsum((s + alpha * ds_a) * (z + alpha * dz_a)) * rcp_nr(sz);
        const double sig = q * q * q;
        const double mu = sz / (double)k;
        const double rc2 = row ? -(mu * sig - ds_a * dz_a) * rcp_nr(s) : 0.0;
        const double r2 = -(soz * rc2);
        const double u2a = fks ? spd_resolve_k(fks, zs, k, r2) : spd_solve2_k<KT>(Hq, HP, k, soz, r2, 0.0).a;
        lap(10);
        const double dt_c = rsum(r2 * m1) * m1inv;
        const double dz_c = row ? u2a - dt_c * m1 : 0.0;
        const double ds_c = -soz * (rc2 + dz_c);
        const double dz = dz_a + dz_c, ds = ds_a + ds_c, dt = dt_a + dt_c;
        mall = row ? fmin(ratio_step(s, ds), ratio_step(z, dz)) : NO_STEP;
        IpmStep st3{NO_STEP, 0};
#define IPM_M3(KK) st3 = ipm_wide23_fn<CutT, KK, false, GSRC>(As, ldA, k, n, n_pad, yv, rys, dyv, dz_c, c_first, c_step)
        IPM_K_SWITCH_WAVES(k, IPM_M3)
#undef IPM_M3
        mall = fmin(mall, st3.m);
        bool neg_y, neg_1y;
        {
            const double wm = wave_min(mall);
            const int flags = (__any(st3.neg & 1) ? 1 : 0) | (__any(st3.neg & 2) ? 2 : 0);
            if (lane == 0) { xch[wave] = wm; xch[NW + wave] = (double)flags; }
            sample_sync<NW>();
            mall = xch[0];
            int fl = (int)xch[NW];
#pragma unroll
            for (int w = 1; w < NW; ++w) { mall = fmin(mall, xch[w]); fl |= (int)xch[NW + w]; }
            neg_y = (fl & 1) != 0;
            neg_1y = (fl & 2) != 0;
            sample_sync<NW>();
        }
        const bool some_empty = !__any(row && ds < 0.0) || !__any(row && dz < 0.0) || !neg_y || !neg_1y;
        const double gmin = some_empty ? fmin(mall, 1.0) : mall;
        alpha = fmax(0.0, fmin(1.0, 0.99 * gmin));             // :70-71
        for (int j = tid; j < n; j += NT) yv[j] += alpha * dyv[j];      // :73
        t += alpha * dt;
        s += alpha * ds;
        z += alpha * dz;
        sample_sync<NW>();
        lap(11);
    }
    return row ? z : 0.0;
}

template <typename CutT, int KT, bool GSRC>
__device__ __noinline__ IpmOut ipm_solve_wide_one_fn(const CutT *As, int ldA, int k, int n, int n_pad, double *ws, double *zs, double *rys,
                                                    double *yv, double *dyv, double *Hm, int HP, double h_i) {
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad); HP = uni(HP);
    int st = 0;
    const double z = ipm_solve_waves<CutT, KT, 1, GSRC>(As, ldA, k, n, n_pad, ws, zs, rys, yv, dyv, Hm, HP, h_i, lane_id(), &st);
    return IpmOut{z, st};
}
