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

// get_step of the reference (:158-163): min over the entries with dv < 0 of -v/dv, and 1 if there is none.  Lanes
// accumulate the per-entry ratios with `ratio_step` (NO_STEP where dv >= 0) and `get_step` turns the wave minimum into
// the reference's value.
constexpr double NO_STEP = 1e300;
__device__ __forceinline__ double ratio_step(double v, double dv) { return dv < 0.0 ? -v / dv : NO_STEP; }
__device__ __forceinline__ double get_step(double lane_min) {
    const double m = wave_min(lane_min);
    return m == NO_STEP ? 1.0 : m;
}

// Runs pdipm_pc on the k staged cuts (rows of As, offsets h_i in row layout).  On return yv[0..n) holds y (LDS) and the
// result is this lane's multiplier z_i (0 beyond k).  *status: 0 ok, 1 = M not positive definite / non-finite.
template <typename CutT, int KT>
__device__ __forceinline__ double ipm_solve(const CutT *As, int ldA, int k, const CutT *crow, int n, int n_pad,
                                            double *ws, double *zs, double *rys, double *yv, double *dyv, double *Hm,
                                            int HP, double h_i, int lane, int *status) {
    const bool row = lane < k;
    double z = row ? 1.0 / (double)k : 0.0;                    // :11
    double s = row ? 1.0 : 0.0;                                // :13
    double t = 1.0;                                            // :14
    for (int j = lane; j < n_pad; j += 64) yv[j] = 0.5;        // :12
    sample_sync<1>();
    *status = 0;
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
    auto cols_dot = [&](double v, int j) -> double {           // (G^T v)_j, v in row layout
        double acc = 0.0;
        for (int i = 0; i < k; ++i) acc += bcast(v, i) * (double)As[i * ldA + j];
        return acc;
    };
    auto rsum = [&](double v) -> double { return ipm_sum(row ? v : 0.0); };
    for (int it = 0; it < 20; ++it) {                          // :16
        // residuals (:26-29)
        double pri2 = 0.0;
        for (int j = lane; j < n_pad; j += 64) {
            const double y = yv[j];
            const double grad = log(y) - log(1.0 - y);         // :17
            const double hinv = 1.0 / (1.0 / y + 1.0 / (1.0 - y));   // :19
            const double ry = j < n ? grad + cols_dot(z, j) : 0.0;
            rys[j] = ry;
            ws[j] = j < n ? hinv : 0.0;
            zs[j] = j < n ? hinv * ry : 0.0;
            pri2 += ry * ry;
        }
        sample_sync<1>();
        const double rt = 1.0 - rsum(z);                       // :27
        const double gy = rows_dot(yv);
        const double rd = row ? gy + h_i - t + s : 0.0;        // :29
        const double pri_res = sqrt(ipm_sum(pri2) + rt * rt), dual_res = sqrt(rsum(rd * rd));
        if (pri_res < 1e-8 && dual_res < 1e-8) break;          // :39
        // M = G Hinv G^T (+ diag(s/z) below) and G Hinv ry in one MFMA sweep (:41, :46)
        contract_mfma<CutT, KT, true>(As, ldA, k, crow, 0, n_pad, ws, zs, Hm, HP);
        sample_sync<1>();
        const double soz = row ? s / z : 1.0;
        const double ghr = row ? Hm[lane * HP + k] : 0.0;
        // affine direction (:53): r = rd - G Hinv ry - (s/z) rc with rc = z
        const double r = rd - ghr - soz * z;
        const Pair um = spd_solve2<KT>(Hm, HP, k, soz, r, 1.0);
        if (!uni(um.ok) || !isfinite(pri_res)) { *status = 1; break; }
        const double m1 = row ? um.b : 0.0, m1sum = rsum(m1);
        const double dt_a = (rsum(r * m1) - rt) / m1sum;
        const double dz_a = row ? um.a - dt_a * m1 : 0.0;      // = M^-1 (r - dt), :48
        const double ds_a = -soz * (z + dz_a);                 // :49
        double my = NO_STEP, m1y = NO_STEP;
        for (int j = lane; j < n_pad; j += 64) {
            const double dy = -ws[j] * (rys[j] + cols_dot(dz_a, j));   // :50
            dyv[j] = dy;
            if (j < n) { my = fmin(my, ratio_step(yv[j], dy)); m1y = fmin(m1y, ratio_step(1.0 - yv[j], -dy)); }
        }
        double alpha = fmin(fmin(fmin(get_step(row ? ratio_step(z, dz_a) : NO_STEP), get_step(row ? ratio_step(s, ds_a) : NO_STEP)),
                                 fmin(get_step(my), get_step(m1y))), 1.0);   // :55-56
        const double sz = rsum(s * z);
        const double q = rsum((s + alpha * ds_a) * (z + alpha * dz_a)) / sz;
        const double sig = q * q * q;                          // :57
        const double mu = sz / (double)k;                      // :59
        // corrector (:61-63): ry = rt = rd = 0, rc = -(mu sig - ds_aff dz_aff) / s
        const double rc2 = row ? -(mu * sig - ds_a * dz_a) / s : 0.0;
        const double r2 = -(soz * rc2);
        const Pair u2 = spd_solve2<KT>(Hm, HP, k, soz, r2, 0.0);
        const double dt_c = rsum(r2 * m1) / m1sum;
        const double dz_c = row ? u2.a - dt_c * m1 : 0.0;
        const double ds_c = -soz * (rc2 + dz_c);
        const double dz = dz_a + dz_c, ds = ds_a + ds_c, dt = dt_a + dt_c;   // :65-68
        my = NO_STEP; m1y = NO_STEP;
        for (int j = lane; j < n_pad; j += 64) {
            const double dy = dyv[j] - ws[j] * cols_dot(dz_c, j);
            dyv[j] = dy;
            if (j < n) { my = fmin(my, ratio_step(yv[j], dy)); m1y = fmin(m1y, ratio_step(1.0 - yv[j], -dy)); }
        }
        const double gmin = fmin(fmin(get_step(row ? ratio_step(s, ds) : NO_STEP), get_step(row ? ratio_step(z, dz) : NO_STEP)),
                                 fmin(get_step(my), get_step(m1y)));
        alpha = fmax(0.0, fmin(1.0, 0.99 * gmin));             // :70-71
        for (int j = lane; j < n; j += 64) yv[j] += alpha * dyv[j];       // :73
        t += alpha * dt;                                       // :74
        s += alpha * ds;                                       // :75
        z += alpha * dz;                                       // :76
        sample_sync<1>();
    }
    return row ? z : 0.0;
}
