// Device code of the dual step (included by be_dual.hip and be_fused.hip): helpers, the f64 MFMA contractions,
// the register-resident small algebra and dual_step_body.  Everything is in an anonymous namespace.
#pragma once
// Dual step of the bundle-entropy method: one wave64 = one sample.
//
// For outer iteration t and every unfinished sample u (one workgroup of 64 lanes):
//   1. append the cut (g_t, h_t = f_t - <g_t, y>) to slot t            dual :143,:151-153
//   2. stage the active bundle rows A[k][n] in LDS
//   3. rank test on A (variant DUAL)                                   dual :155-161
//   4. projected Newton on the simplex for lam                         dual :15-85 / rl :14-83
//   5. y <- 1/(1+exp(A^T lam)), clip/stall test (RL), prune lam == 0   dual :165-174 / rl :117-131
//
// Layouts inside the wave
//   column layout: lane l owns columns l, l+64, ... of the bundle (a = A^T lam, z, w, y)
//   row layout:    lane i < k owns bundle row i (lam_i, c_i, g_i, step_i)
// The k x k contractions A diag(w) A^T and A z go through v_mfma_f64_16x16x4_f64 with
// both operands gathered from the LDS-resident bundle; the (k-1) x (k-1) Newton system
// is solved in LDS by Gaussian elimination with partial pivoting, lane = matrix row.
#include <hip/hip_runtime.h>
#include <math.h>
#include <type_traits>

#include "be_common.h"
#include "be_kernels.h"
#include "icnn_be.h"

namespace icnn_be {

namespace {

constexpr double BOUND_EPS = 1e-12;    // dual :21
constexpr double ARMIJO_ALPHA = 1e-5;  // dual :22
constexpr double GRAD_TOL = 1e-10;     // dual :50
constexpr double TINY = 1e-10;         // dual :79
constexpr double CYCLE_TOL = 1e-13;    // DESIGN.md "limit-cycle shortcut"
// A limit cycle whose own rounding jitter is ABOVE CYCLE_TOL never passes the repeat test (ill-conditioned Newton systems: the
// iterates repeat to 1e-12, not 1e-13) and ran the full cap of 100 updates -- the same sample, in every outer iteration; a third
// of all cycling solves of the benchmark batch (tools: /DESIGN.md section 6c), and THE straggler of every long solve.  Noise-floor
// rule (variant dual): from NOISE_T0 updates on, period p is also accepted when max|lam_t - lam_{t-p}| is below NOISE_TOL and has
// STOPPED SHRINKING -- not below its value an update cycle earlier: a converging sequence shrinks geometrically, a cycle at its
// rounding floor fluctuates.  What is returned is the iterate of the right phase, as for exact repeats; it differs from the
// reference's lam_100 by the cycle's own jitter (max 2.5e-11 on 10 243 recorded solves, against 4.1e-11 without the rule).
constexpr double NOISE_TOL = 1e-10;
constexpr int NOISE_T0 = 12;
constexpr int ACCEL_T0 = 24;           // extrapolation of slow 2-cycles: not before this many updates,
constexpr int ACCEL_GAP = 6;           // this many updates apart,
constexpr double ACCEL_NEG_RESID = 1e-10;  // oscillating subsequences (negative ratio): see the extrapolation in dual_step_body
constexpr double ACCEL_D2MAX = 1e-3;   // only once lam_t - lam_{t-2} is this small
constexpr double ACCEL_RMAX = 0.98;    // and the contraction ratio is below this

template <typename T> struct Cut;
// NumPy's float32 exp is not correctly rounded (39 % of results are 1-2 ulp off); the
// reference computes the k = 1 update y = 1/(1+exp(g)) with it in float32 (dual :168), and
// that y is the point of the next cut.  To start from bit-identical iterates the device
// evaluates the same operation sequence as numpy/_core/src/umath/
// loops_exponent_log.dispatch.c.src (simd_exp_FLOAT, AVX512F/AVX2 paths): Cody-Waite
// reduction by round(x log2e) with fused multiply-adds, a (5,2) rational minimax, scalef.
// Every operation is a single IEEE float32 rounding, so the result is bit-identical
// (checked against np.exp on the GPU box by tests/test_gpu_parity.py).
__device__ __noinline__ float numpy_expf(float x) {
#pragma clang fp contract(off)
    if (x != x) return x;
    if (x >= 88.72283935546875f) return __builtin_inff();
    if (x <= -103.97208404541015625f) return 0.0f;
    float q = x * 1.44269504088896341f;
    q = q + 12582912.0f;                       // 0x1.8p23: round to nearest integer
    q = q - 12582912.0f;
    float r = __builtin_fmaf(q, -6.93145752e-1f, x);
    r = __builtin_fmaf(q, -1.42860677e-6f, r);
    float num = __builtin_fmaf(5.082762527590693718096e-04f, r, 6.757896990527504603057e-03f);
    num = __builtin_fmaf(num, r, 5.114512081637298353406e-02f);
    num = __builtin_fmaf(num, r, 2.473615434895520810817e-01f);
    num = __builtin_fmaf(num, r, 7.257664613233124478488e-01f);
    num = __builtin_fmaf(num, r, 9.999999999980870924916e-01f);
    float den = __builtin_fmaf(2.159509375685829852307e-02f, r, -2.742335390411667452936e-01f);
    den = __builtin_fmaf(den, r, 1.0f);
    const float poly = num / den;
    return ldexpf(poly, (int)q);
}

template <> struct Cut<float> {
    static constexpr double eps = 1.1920928955078125e-07;
    static __device__ __forceinline__ float sigmoid_neg(float g) {
#pragma clang fp contract(off)
        const float e = numpy_expf(g);
        const float d = 1.0f + e;
        return 1.0f / d;
    }
};
template <> struct Cut<double> {
    static constexpr double eps = 2.220446049250313e-16;
    static __device__ __forceinline__ double sigmoid_neg(double g) { return 1.0 / (1.0 + exp(g)); }
};

__device__ __forceinline__ double softplus_stable(double v) {   // dual :6-12
    return v > 1.0 ? log1p(exp(-v)) + v : log1p(exp(v));
}

// LDS carve-up shared by host (size query) and device.  The pairwise-sum scratch aliases the
// Hm region (they are never live together).
struct Carve {
    int As, zs, ws, sp, yv, dv, Hm, Hp, ints, total;
};
constexpr int IPM_KMAX_WAVES = 20; // most cuts ipm_solve_waves (be_ipm_dev.h, 256-register kernels) takes
// rl: variant RL keeps the softplus terms in a column buffer of its own; ipm: the interior-point variant needs that
// buffer (residual ry) and two more (y, dy)
__host__ __device__ inline Carve carve(int KT, int rows, int ldA, int n_pad, int cut_bytes, int n_leaves,
                                       bool rl, int nw = 1, bool own_const_rows = true, bool ipm = false,
                                       bool glb = false) {
    Carve c;
    int o = 0;
    auto take = [&](int bytes) { int at = o; o += (bytes + 15) & ~15; return at; };
    // + a row of zeros and a row of ones (contract_mfma) unless shared; glb: the rows live in device memory (st.scratch)
    c.As = take(glb ? 0 : (rows + (own_const_rows ? 2 : 0)) * ldA * cut_bytes);
    c.zs = take(n_pad * 8);
    c.ws = take(n_pad * 8);
    c.sp = (rl || ipm) ? take(n_pad * 8) : c.ws;
    c.yv = ipm ? take(n_pad * 8) : c.zs;
    c.dv = ipm ? take(n_pad * 8) : c.zs;
    const int hp = (rows + 1) | 1;
    int hm = rows * hp * 8;
    const int scratch = (KT * n_leaves + 2 * KT) * 8;
    if (scratch > hm) hm = scratch;
    c.Hm = take(hm);
    // per-wave partial contractions (the rank test's Gram matrix of ALL rows in both variants, the Newton system of variant
    // dual) / each wave's own copy of the k x (k + 1) system (interior point on several waves).  Where this does not fit next to
    // the column buffers the launcher takes the one-wave instance (launch_dual_step: interior point at n_pad = 3072 from 23 rows)
    c.Hp = nw > 1 ? take(nw * rows * hp * 8) : c.Hm;
    c.ints = take(KT * 4);
    c.total = o;
    return c;
}

// wave-uniform source lane -> value of that lane in every lane (v_readlane_b32 x2, no LDS)
__device__ __forceinline__ double bcast(double x, int src_lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), src_lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(x), src_lane);
    return __hiloint2double(hi, lo);
}

// Arguments of a non-inlined device function arrive in VGPRs and pointers as generic addresses: the
// compiler then treats every branch on them as divergent (exec-mask juggling) and every LDS access
// as a FLAT access.  These helpers restore what the caller knows: wave-uniform scalars, LDS pointers.
typedef const __attribute__((address_space(3))) double lds_cdouble;
// Opaque use of a value: the compiler must materialise it here (stops it from sinking loads into branches).
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ int uni(int v) { return __builtin_amdgcn_readfirstlane(v); }
__device__ __forceinline__ unsigned long long uni(unsigned long long v) {
    const unsigned lo = __builtin_amdgcn_readfirstlane((int)(unsigned)v);
    const unsigned hi = __builtin_amdgcn_readfirstlane((int)(unsigned)(v >> 32));
    return ((unsigned long long)hi << 32) | lo;
}
__device__ __forceinline__ double uni(double v) {
    return __hiloint2double(__builtin_amdgcn_readfirstlane(__double2hiint(v)),
                            __builtin_amdgcn_readfirstlane(__double2loint(v)));
}

// D = A B^T style contractions over the columns of the LDS bundle with f64 MFMA.
//   HESS = false:  Hm[i][j] = sum_c A[i][c] A[j][c]                     (Gram, rank test)
//   HESS = true :  Hm[i][j] = sum_c A[i][c] w[c] A[j][c]   (j < k)       dual :36
//                  Hm[i][k] = sum_c A[i][c] z[c]                         dual :35
// A operand: lane (r16, q) holds A[ti*16+r16][c0+q]; B operand: lane holds B[c0+q][tj*16+r16].
// Result: lane holds D[ti*16 + q + 4r][tj*16 + r16], r = 0..3 (f64 C/D map).
// Small bundles (at most 8 rows/columns of output): v_mfma_f64_4x4x4_4b_f64, whose four 4x4 blocks
// are used as the 2x2 tiling of an 8x8 result.  Measured lane layout on gfx950
// (tools/probes/mfma_f64_4x4_probe.hip): with kq = lane>>4, block = (lane>>2)&3, r = lane&3
//   A operand lane holds A_block[r][kq], B operand lane holds B_block[kq][r],
//   result lane holds D_block[lane>>4][lane&3].
// Same 4 columns per instruction as the 16x16x4 form, but 4 passes instead of 16.
// Operand masking without arithmetic: the bundle in LDS is followed by a row of zeros (row `zrow`) and
// a row of ones (row `zrow + 1`).  A lane whose output row/column is outside the bundle reads the zero
// row; the lane that produces column k (A z) reads the ones row and z instead of a bundle row and w.
// So per k-step a lane converts two cut values and does one multiply -- same bits as masking with 0/1
// factors (x * 1 = x, fma(x, w, +-0) = x * w).
struct NoLap { __device__ void operator()(int) const {} };
template <typename CutT, bool HESS, typename LapF = NoLap>
__device__ void contract_mfma_8x8(const CutT *As, int ldA, int k, const CutT *crow, int cbeg, int cend,
                                  const double *ws, const double *zs, double *Hm, int HP, LapF lapf = LapF()) {
    const int lane = thread_id() & 63, kq = lane >> 4, blk = (lane >> 2) & 3, r = lane & 3;
    const int ra = 4 * (blk >> 1) + r, cb = 4 * (blk & 1) + r;
    const bool zcol = HESS && cb == k;
    // crow: the constant rows -- zeros at crow[0 .. ldA), ones at crow[ldA .. 2 ldA)
    const CutT *pa = (ra < k ? As + ra * ldA : crow) + kq;
    const CutT *pb = (cb < k ? As + cb * ldA : (zcol ? crow + ldA : crow)) + kq;
    const double *pwz = (zcol ? zs : ws) + kq;
    double acc0 = 0.0, acc1 = 0.0;                       // two chains hide the MFMA latency
    cbeg = uni(cbeg); cend = uni(cend);                  // scalar loop control
    // Software pipeline, distance one: the LDS reads of the next 16 columns are issued before the four
    // MFMAs of the current 16, so the LDS round trip overlaps the arithmetic instead of preceding it.
    // Two register sets alternate (no copies); scheduling barriers and an opaque use after the MFMAs
    // keep the compiler from folding the prefetch back into "load, wait, use".
    CutT xa[4], xb[4], ya[4], yb[4];
    double xw[4], yw[4];
    auto gather = [&](int c0, CutT (&ga)[4], CutT (&gb)[4], double (&gw)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            ga[s] = pa[c0 + 4 * s];
            gb[s] = pb[c0 + 4 * s];
            if (HESS) gw[s] = pwz[c0 + 4 * s];
        }
    };
    auto stage = [&](int cnext, CutT (&ca)[4], CutT (&cb_)[4], double (&cw)[4], CutT (&na)[4], CutT (&nb)[4],
                     double (&nw)[4]) {
        gather(cnext, na, nb, nw);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            const double av = (double)ca[s];
            const double bv = HESS ? (double)cb_[s] * cw[s] : (double)cb_[s];
            if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) { pin(na[s]); pin(nb[s]); if (HESS) pin(nw[s]); }
    };
    if (cbeg < cend) gather(cbeg, xa, xb, xw);
    // Nothing may be outstanding when the loop is entered: otherwise the wait-count pass, merging the
    // preheader state into the loop header, puts an lgkmcnt(0) right behind the prefetch of every stage.
    __builtin_amdgcn_s_waitcnt(0xc07f);                  // lgkmcnt(0), visible to the wait-count pass
    lapf(10);
    const int clast = cend - 16;
    for (int c0 = cbeg; c0 < cend; c0 += 32) {           // column range is a multiple of 16
        stage(c0 + 16 < cend ? c0 + 16 : clast, xa, xb, xw, ya, yb, yw);
        if (c0 + 16 < cend) stage(c0 + 32 < cend ? c0 + 32 : clast, ya, yb, yw, xa, xb, xw);
    }
    lapf(11);
    const int row = 4 * (blk >> 1) + kq, col = 4 * (blk & 1) + r;
    const int ncolsB = HESS ? k + 1 : k;
    if (row < k && col < ncolsB) Hm[row * HP + col] = acc0 + acc1;
}

// 17 .. 20 cuts (round 6): TWO 16 x 16 tiles instead of the three of the (ti, tj) tiling below, whose tiles (0, 1) and (1, 1)
// hold 1 .. 5 useful columns each.  A tile's A-operand rows and B-operand columns need not be contiguous -- every lane picks its
// row pointer --, so the pairs that involve the extra rows e = 16 .. k - 1 fit ONE tile:
//   tile 1   rows 0 .. 15  x  columns 0 .. 14 and A z (HESS) / 0 .. 15 (Gram);  H[i][15] = H[15][i] by symmetry (HESS)
//   tile 2   rows 4 .. 19  x  columns { 16 .. 19,  A z,  0 .. 3,  15 }: (i, e) for i = 4 .. 19, (A z)_e, (e, i) for i = 0 .. 3
//            mirrored into (i, e), and the diagonal entry (15, 15) that tile 1 gave up for A z.
// A float64 16x16x4 MFMA occupies the wave for ~125 cycles with its operand set-up (tools/probes/dual_update_probe.hip: 5.0 k
// cycles per tile sweep of 160 columns), so a Newton update at 17 .. 20 cuts drops from 15 k to 10 k cycles here, the rank
// test's Gram matrix from 13.6 k to 9.4 k.  Same k-ordered fma chains per entry; the entries (i, 15), i < 15, and (i, e), i < 4,
// are now formed with the OTHER factor carrying the weight w (H is symmetric: they differ from the three-tile result in the
// last bit).  Every kernel takes this path for such bundles, so the dispatch paths stay bit-identical among themselves.
template <typename CutT, bool HESS>
__device__ void contract_mfma_cover2(const CutT *As, int ldA, int k, const CutT *crow, int cbeg, int cend, const double *ws,
                                     const double *zs, double *Hm, int HP) {
    const int lane = thread_id() & 63, r16 = lane & 15, q = lane >> 4;
    cbeg = uni(cbeg); cend = uni(cend);
    constexpr int ZC = 64, NONE = 65;                          // B-operand column codes besides a bundle row
#pragma unroll 1
    for (int t = 0; t < 2; ++t) {
        const int ra = t == 0 ? r16 : 4 + r16;                 // A-operand row of this lane
        int cb;                                                // B-operand column of this lane
        if (t == 0) cb = HESS ? (r16 < 15 ? r16 : ZC) : r16;
        else if (r16 < 4) cb = 16 + r16;
        else if (HESS) cb = r16 == 4 ? ZC : (r16 < 9 ? r16 - 5 : (r16 == 9 ? 15 : NONE));
        else cb = r16 < 8 ? r16 - 4 : NONE;
        const bool zcol = cb == ZC;
        const CutT *pa = (ra < k ? As + ra * ldA : crow) + q;
        const CutT *pb = (cb < k ? As + cb * ldA : (zcol ? crow + ldA : crow)) + q;
        const double *pwz = (zcol ? zs : ws) + q;
        d4 acc = {0.0, 0.0, 0.0, 0.0}, acc_odd = {0.0, 0.0, 0.0, 0.0};
        CutT xa[4], xb[4], ya[4], yb[4];                       // software pipeline as in contract_mfma_8x8
        double xw[4], yw[4];
        auto gather = [&](int c0, CutT (&ga)[4], CutT (&gb)[4], double (&gw)[4]) {
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                ga[s] = pa[c0 + 4 * s];
                gb[s] = pb[c0 + 4 * s];
                if (HESS) gw[s] = pwz[c0 + 4 * s];
            }
        };
        auto stage = [&](int cnext, CutT (&ca)[4], CutT (&cb_)[4], double (&cw)[4], CutT (&na)[4], CutT (&nb)[4], double (&nw)[4]) {
            gather(cnext, na, nb, nw);
            __builtin_amdgcn_sched_barrier(0);
            // all four operand chains (cvt, cvt, mul) FIRST, then the four MFMAs back to back: float64 VALU work issued while
            // the wave's own float64 MFMA is in flight stalls on the shared DP pipe -- interleaved, an MFMA with its chain costs
            // a lone wave 204 cycles, batched 115 (the MFMA alone: 70; tools/probes/mfma_overlap_probe.hip).  Same arithmetic.
            double av[4], bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                av[s] = (double)ca[s];
                bv[s] = HESS ? (double)cb_[s] * cw[s] : (double)cb_[s];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s & 1) acc_odd = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc_odd, 0, 0, 0);
                else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) { pin(na[s]); pin(nb[s]); if (HESS) pin(nw[s]); }
        };
        if (cbeg < cend) gather(cbeg, xa, xb, xw);
        __builtin_amdgcn_s_waitcnt(0xc07f);                    // lgkmcnt(0), see contract_mfma_8x8
        const int clast = cend - 16;
        for (int c0 = cbeg; c0 < cend; c0 += 32) {             // column range is a multiple of 16
            stage(c0 + 16 < cend ? c0 + 16 : clast, xa, xb, xw, ya, yb, yw);
            if (c0 + 16 < cend) stage(c0 + 32 < cend ? c0 + 32 : clast, ya, yb, yw, xa, xb, xw);
        }
        acc += acc_odd;
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = (t == 0 ? 0 : 4) + q + 4 * r;      // bundle row of this result
            if (row >= k) continue;
            if (zcol) {
                if (t == 0 || row >= 16) Hm[row * HP + k] = acc[r];
            } else if (t == 0) {
                Hm[row * HP + cb] = acc[r];                                        // (rows 0 .. 15) x (columns 0 .. 14 | 15)
                if (HESS && row == 15) Hm[cb * HP + 15] = acc[r];                  // H[i][15] = H[15][i]
            } else if (cb < k) {
                if (cb >= 16) {                                                    // (i, e), i = 4 .. 19
                    Hm[row * HP + cb] = acc[r];
                    if (row < 16) Hm[cb * HP + row] = acc[r];
                } else if (cb < 4) {                                               // (e, i), i = 0 .. 3, and its mirror
                    if (row >= 16) { Hm[row * HP + cb] = acc[r]; Hm[cb * HP + row] = acc[r]; }
                } else if (row == 15) {                                            // cb == 15: the diagonal entry
                    Hm[15 * HP + 15] = acc[r];
                }
            }
        }
    }
}

template <typename CutT, int KT, bool HESS, typename LapF = NoLap>
__device__ void contract_mfma(const CutT *As, int ldA, int k, const CutT *crow, int cbeg, int cend, const double *ws,
                              const double *zs, double *Hm, int HP, LapF lapf = LapF()) {
    const int lane = thread_id() & 63, r16 = lane & 15, q = lane >> 4;
    const int ncolsB = HESS ? k + 1 : k;
    if (ncolsB <= 8) {
        contract_mfma_8x8<CutT, HESS>(As, ldA, k, crow, cbeg, cend, ws, zs, Hm, HP, lapf);
        return;
    }
    if (KT > 16 && k >= 17 && k <= 20) {
        contract_mfma_cover2<CutT, HESS>(As, ldA, k, crow, cbeg, cend, ws, zs, Hm, HP);
        return;
    }
    cbeg = uni(cbeg); cend = uni(cend);
    for (int ti = 0; ti * 16 < k; ++ti) {
        for (int tj = ti; tj * 16 < ncolsB; ++tj) {
            d4 acc = {0.0, 0.0, 0.0, 0.0}, acc_odd = {0.0, 0.0, 0.0, 0.0};   // two chains: a dependent
            const int ra = ti * 16 + r16, cb = tj * 16 + r16;              // 16x16x4 f64 MFMA costs 65 cycles
            const bool zcol = HESS && cb == k;
            const CutT *pa = (ra < k ? As + ra * ldA : crow) + q;
            const CutT *pb = (cb < k ? As + cb * ldA : (zcol ? crow + ldA : crow)) + q;
            const double *pwz = (zcol ? zs : ws) + q;
            CutT xa[4], xb[4], ya[4], yb[4];                    // software pipeline as in contract_mfma_8x8
            double xw[4], yw[4];
            auto gather = [&](int c0, CutT (&ga)[4], CutT (&gb)[4], double (&gw)[4]) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    ga[s] = pa[c0 + 4 * s];
                    gb[s] = pb[c0 + 4 * s];
                    if (HESS) gw[s] = pwz[c0 + 4 * s];
                }
            };
            auto stage = [&](int cnext, CutT (&ca)[4], CutT (&cb_)[4], double (&cw)[4], CutT (&na)[4],
                             CutT (&nb)[4], double (&nw)[4]) {
                gather(cnext, na, nb, nw);
                __builtin_amdgcn_sched_barrier(0);
                // (round 6: the four operand chains first, then the MFMAs back to back -- see contract_mfma_cover2)
                double av[4], bv[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    av[s] = (double)ca[s];
                    bv[s] = HESS ? (double)cb_[s] * cw[s] : (double)cb_[s];
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    // (only in the 32-slot kernels, whose bundles actually live here: the 16-slot kernel is held
                    //  to 128 VGPRs and the second accumulator would spill in its Newton loop)
                    if (KT > 16 && (s & 1)) acc_odd = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc_odd, 0, 0, 0);
                    else acc = __builtin_amdgcn_mfma_f64_16x16x4f64(av[s], bv[s], acc, 0, 0, 0);
                }
                __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                for (int s = 0; s < 4; ++s) { pin(na[s]); pin(nb[s]); if (HESS) pin(nw[s]); }
            };
            if (cbeg < cend) gather(cbeg, xa, xb, xw);
            __builtin_amdgcn_s_waitcnt(0xc07f);                 // lgkmcnt(0), see contract_mfma_8x8
            const int clast = cend - 16;
            for (int c0 = cbeg; c0 < cend; c0 += 32) {          // column range is a multiple of 16
                stage(c0 + 16 < cend ? c0 + 16 : clast, xa, xb, xw, ya, yb, yw);
                if (c0 + 16 < cend) stage(c0 + 32 < cend ? c0 + 32 : clast, ya, yb, yw, xa, xb, xw);
            }
            if (KT > 16) acc += acc_odd;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = ti * 16 + q + 4 * r, col = tj * 16 + r16;
                if (row < k && col < ncolsB) {
                    Hm[row * HP + col] = acc[r];
                    if (tj != ti && col < k) Hm[col * HP + row] = acc[r];
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------
// Small dense algebra, register resident: lane i holds row i of the (<= KT x KT) system in
// statically indexed registers, pivot rows are broadcast with v_readlane -- no LDS, no barriers.
// ---------------------------------------------------------------------------------------------

// 1/d to within an ulp or two: v_rcp_f64 plus two Newton-Raphson steps (what the IEEE division
// expansion does internally, minus its scaling and fix-up instructions).  Pivots are never denormal
// or infinite here (entries of A w A^T with |A| finite, w <= 1/4), so the shortcuts are safe.
__device__ __forceinline__ double rcp_nr(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = __builtin_fma(-d, r, 1.0);
    r = __builtin_fma(r, e, r);
    e = __builtin_fma(-d, r, 1.0);
    return __builtin_fma(r, e, r);
}

// exp and log for the inner loops (round 5).  The library routines (ocml) cost ~55 (exp) and ~130 (log) instructions per call
// for a correctly rounded-ish result with full special-case handling; a Newton update evaluates three sigmoids per lane, an
// interior-point iteration three logarithms, and at sixteen samples per CU the dual phase is bound by instruction issue
// (DESIGN.md section 6).  These are plain argument reductions + Horner polynomials, ~21 and ~33 instructions, relative error
// below 3e-16 on their domains -- the iterations they feed are self-correcting (a fixed point of the Newton / KKT iteration
// is a fixed point with either routine, to that accuracy).  NOT used where a value is final: y = 1 / (1 + exp(A^T lam))
// (dual :165) keeps the library exp and IEEE division.
//   fast_exp(x): x clamped to [-750, 750] (0 / +inf beyond, as exp itself), k = rint(x log2 e), r = x - k ln2 (two-term ln2,
//   |r| <= 0.3466), exp(r) by its Taylor polynomial of degree 13 (next term 0.3466^14 / 14! = 4e-18), scaled by 2^k.
__device__ __forceinline__ double fast_exp(double x) {
    x = fmin(fmax(x, -750.0), 750.0);
    const double kf = __builtin_rint(x * 1.4426950408889634);
    double r = __builtin_fma(kf, -6.93147180369123816490e-01, x);        // ln2 hi (trailing zeros: k * hi is exact)
    r = __builtin_fma(kf, -1.90821492927058770002e-10, r);               // ln2 lo
    double p = 1.6059043836821613e-10;                                   // 1/13!
    p = __builtin_fma(p, r, 2.08767569878681e-09);                       // 1/12!
    p = __builtin_fma(p, r, 2.505210838544172e-08);                      // 1/11!
    p = __builtin_fma(p, r, 2.755731922398589e-07);                      // 1/10!
    p = __builtin_fma(p, r, 2.7557319223985893e-06);                     // 1/9!
    p = __builtin_fma(p, r, 2.48015873015873e-05);                       // 1/8!
    p = __builtin_fma(p, r, 1.984126984126984e-04);                      // 1/7!
    p = __builtin_fma(p, r, 1.388888888888889e-03);                      // 1/6!
    p = __builtin_fma(p, r, 8.333333333333333e-03);                      // 1/5!
    p = __builtin_fma(p, r, 4.1666666666666664e-02);                     // 1/4!
    p = __builtin_fma(p, r, 1.6666666666666666e-01);                     // 1/3!
    p = __builtin_fma(p, r, 0.5);
    p = __builtin_fma(p, r, 1.0);
    p = __builtin_fma(p, r, 1.0);
    return __builtin_amdgcn_ldexp(p, (int)kf);
}
//   fast_log(x), x > 0 finite: x = m 2^e with m in [sqrt(1/2), sqrt(2)), s = (m - 1) / (m + 1) (|s| <= 0.1716),
//   log m = 2 atanh(s) = 2 s + s^3 (2/3 + 2/5 s^2 + ... + 2/21 s^18) (next term 1.5e-18 relative), + e ln2 in two terms.
__device__ __forceinline__ double fast_log(double x) {
    double m = __builtin_amdgcn_frexp_mant(x);                           // [0.5, 1)
    int e = __builtin_amdgcn_frexp_exp(x);
    const bool low = m < 0.70710678118654752;
    m = low ? m + m : m;
    e = low ? e - 1 : e;
    const double s = (m - 1.0) * rcp_nr(m + 1.0);
    const double s2 = s * s;
    double p = 2.0 / 21.0;
    p = __builtin_fma(p, s2, 2.0 / 19.0);
    p = __builtin_fma(p, s2, 2.0 / 17.0);
    p = __builtin_fma(p, s2, 2.0 / 15.0);
    p = __builtin_fma(p, s2, 2.0 / 13.0);
    p = __builtin_fma(p, s2, 2.0 / 11.0);
    p = __builtin_fma(p, s2, 2.0 / 9.0);
    p = __builtin_fma(p, s2, 2.0 / 7.0);
    p = __builtin_fma(p, s2, 2.0 / 5.0);
    p = __builtin_fma(p, s2, 2.0 / 3.0);
    const double lm = __builtin_fma(s * s2, p, s + s);
    const double ed = (double)e;
    return __builtin_fma(ed, 6.93147180369123816490e-01, __builtin_fma(ed, 1.90821492927058770002e-10, lm));
}

// softplus for the RL variant's objective values (rl :34, :72: one per column and Armijo trial), same branches as
// softplus_stable (dual :6-12): log1p(u) = log(w) + (u - (w - 1)) / w with w = 1 + u rounded -- the second term restores what
// the rounding of 1 + u lost (u itself when u < 1e-16), so the result keeps full relative accuracy for tiny u
__device__ __forceinline__ double softplus_fast(double v) {
    const bool big = v > 1.0;
    const double u = fast_exp(big ? -v : v);
    const double w = 1.0 + u;
    const double l = __builtin_fma(u - (w - 1.0), __builtin_amdgcn_rcp(w), fast_log(w));
    return big ? l + v : l;
}

// sigmoid(a) = 1 / (1 + exp(-a)) for the Newton update's column weights (dual :33): exp(-a) capped at e^700 so that the
// reciprocal's Newton steps stay finite (z = 1e-304 where the exact quotient underflows to 0: w = z (1 - z) is as negligible)
__device__ __forceinline__ double sigmoid_fast(double a) { return rcp_nr(1.0 + fast_exp(fmin(-a, 700.0))); }

// Unpivoted LDL^T inertia: number of eigenvalues of S (k x k, in Hm) that are not above mu.
// Rows and columns >= k are identity, so the elimination needs no per-column bound checks.
template <int KT>
__device__ __noinline__ int inertia_not_above_ks(const double *Hm_, int HP, int k, double mu) {
    const int lane = lane_id();
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k); mu = uni(mu);
    double M[KT];
#pragma unroll
    for (int j = 0; j < KT; ++j) M[j] = Hm[(lane < k ? lane : 0) * HP + (j < k ? j : 0)];
#pragma unroll
    for (int j = 0; j < KT; ++j) pin(M[j]);                       // unconditional loads, all in flight
#pragma unroll
    for (int j = 0; j < KT; ++j)
        M[j] = (lane < k && j < k) ? M[j] - (j == lane ? mu : 0.0) : (j == lane ? 1.0 : 0.0);
    int neg = 0;
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        if (p < k) {
            const double d = bcast(M[p], p);
            if (!(d > 0.0)) {
                ++neg;
                if (d == 0.0) return neg + (k - p - 1);
            }
            const double f = lane > p ? M[p] * rcp_nr(d) : 0.0;
#pragma unroll
            for (int j = p + 1; j < KT; ++j) M[j] -= f * bcast(M[j], p);
        }
    }
    return neg;
}

// Cyclic Jacobi eigenvalues of the symmetric k x k matrix in Hm (destroyed), lane 0 only.
// Rare path of the rank test.  Eigenvalues end up on the diagonal.
template <int KT, int NW>
__device__ void jacobi_lane0(double *Ms, int HP, int k, int tid) {
    if (tid == 0) {
        for (int sweep = 0; sweep < 30; ++sweep) {
            double off = 0.0, tr = 0.0;
            for (int p = 0; p < k; ++p) tr += Ms[p * HP + p];
            for (int p = 0; p < k - 1; ++p)
                for (int q = p + 1; q < k; ++q) {
                    const double apq = Ms[p * HP + q];
                    off += apq * apq;
                    if (apq == 0.0) continue;
                    const double theta = (Ms[q * HP + q] - Ms[p * HP + p]) / (2.0 * apq);
                    const double t = theta != 0.0
                        ? copysign(1.0, theta) / (fabs(theta) + sqrt(theta * theta + 1.0)) : 1.0;
                    const double c = 1.0 / sqrt(t * t + 1.0), s = t * c;
                    for (int j = 0; j < k; ++j) {
                        const double rp = Ms[p * HP + j], rq = Ms[q * HP + j];
                        Ms[p * HP + j] = c * rp - s * rq;
                        Ms[q * HP + j] = s * rp + c * rq;
                    }
                    for (int i = 0; i < k; ++i) {
                        const double cp = Ms[i * HP + p], cq = Ms[i * HP + q];
                        Ms[i * HP + p] = c * cp - s * cq;
                        Ms[i * HP + q] = s * cp + c * cq;
                    }
                }
            const double lim = 1e-22 * tr;
            if (off <= 1e-60 || off <= lim * lim) break;
        }
    }
    sample_sync<NW>();
}

// Solve the reduced Newton system H0[free,free] d = -g0[free] (dual :45,:53-55).  Lane i builds row
// i of the masked full-size system (identity on bound rows, so the free block is untouched),
// Gaussian elimination in natural order: the reduced Hessian is symmetric positive semi-definite,
// no pivoting is needed and the result equals LAPACK's up to rounding.  Returns false on an exactly
// zero pivot -- what LAPACK reports as singular: variant DUAL raises there (dual :56-63), variant RL
// keeps lam and leaves the Newton loop (rl :55-62).  With duplicate cuts (the RL variant has no rank
// test) the MFMA-built Hessian has bit-identical rows, so the exact zero is the normal outcome there;
// see DESIGN.md "RL variant and degenerate bundles" for what the reference's own LAPACK does on them.
// Result in registers: an output reference of a non-inlined function would live in scratch memory
// (a round trip through the vector memory path on every Newton update).
struct StepResult {
    double step;
    int ok;
};
template <int KT>
__device__ __noinline__ StepResult newton_step_ks(const double *Hm_, int HP, int k, int piv, unsigned long long fmask,
                                                  bool is_free, double g0) {
    const int lane = lane_id();
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k); piv = uni(piv); fmask = uni(fmask);
    double M[KT + 1];
    // unconditional (clamped) LDS reads + selects: no exec-mask branches around the loads.  Eight columns at a time:
    // two KT-sized operand arrays next to M made the 32-slot kernels need 241 VGPRs (two waves per SIMD); the
    // arithmetic per entry is unchanged
    const int rl = lane < k ? lane : 0;
    double h_ip = Hm[rl * HP + piv], h_pp = Hm[piv * HP + piv];
    pin(h_ip);
    pin(h_pp);
    static_assert(KT % 4 == 0, "columns are loaded four at a time");
#pragma unroll
    for (int j0 = 0; j0 < KT; j0 += 4) {
        double h_ij[4], h_jp[4];
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j0 + jj, jc = j < k ? j : 0;
            h_ij[jj] = Hm[rl * HP + jc];
            h_jp[jj] = Hm[jc * HP + piv];
        }
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) { pin(h_ij[jj]); pin(h_jp[jj]); }   // the four pairs in flight, none sunk into a branch
#pragma unroll
        for (int jj = 0; jj < 4; ++jj) {
            const int j = j0 + jj;
            // H0[i][j] = ((H[i][j] - keep_i H[j][piv]) - H[i][piv] keep_j) + H[piv][piv] keep_i keep_j
            const double hv = ((h_ij[jj] - h_jp[jj]) - h_ip) + h_pp;
            const bool use = j < k && is_free && ((fmask >> j) & 1ull);
            M[j] = use ? hv : (j == lane ? 1.0 : 0.0);
        }
    }
    M[KT] = is_free ? -g0 : 0.0;
    // Rows/columns that are bound or >= k are identity: pivots there are skipped (scalar branch), and
    // columns need no bound checks.  Lane p keeps 1/pivot_p for the back substitution.
    double rinv = 1.0;
#pragma unroll
    for (int p = 0; p < KT; ++p) {
        if (p < k && ((fmask >> p) & 1ull)) {
            const double d = bcast(M[p], p);
            if (!(d != 0.0)) return StepResult{0.0, 0};           // exact zero (or NaN): singular for LAPACK
            const double inv = rcp_nr(d);
            rinv = lane == p ? inv : rinv;
            const double f = lane > p ? M[p] * inv : 0.0;
#pragma unroll
            for (int j = p + 1; j < KT; ++j) M[j] -= f * bcast(M[j], p);
            M[KT] -= f * bcast(M[KT], p);
        }
    }
#pragma unroll
    for (int p = KT - 1; p >= 0; --p) {
        if (p < k && ((fmask >> p) & 1ull)) {
            const double x = bcast(M[KT] * rinv, p);
            M[KT] = lane == p ? x : (lane < p ? M[KT] - M[p] * x : M[KT]);
        }
    }
    return StepResult{is_free ? M[KT] : 0.0, 1};
}

// ---- KS <= 16: the same eliminations with DPP64 row broadcasts ---------------------------------
// gfx90a+ can broadcast one lane of every 16-lane row to the whole row in a single 64-bit DPP move
// (v_mov_b64_dpp row_newbcast:P).  With the system in lanes 0..15 this replaces the two v_readlane +
// hazard nop of the SGPR route, keeps everything in VGPRs and lets the elimination be straight-line
// code: identity rows (bound, or >= k) have pivot 1 and multiplier -0, so they are processed like any
// other row instead of being branched around.  Lanes 16..63 compute on garbage and are ignored.
template <int I, int N, typename F>
__device__ __forceinline__ void static_for(F &&f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}
template <int P>
__device__ __forceinline__ double row_bcast(double v) {
    return __builtin_amdgcn_update_dpp(0.0, v, 0x150 + P, 0xf, 0xf, true);   // bound_ctrl: no "old" value to set up
}

template <int KS>
__device__ __noinline__ int inertia_not_above_dpp(const double *Hm_, int HP, int k, double mu) {
    static_assert(KS <= 16, "row broadcasts stay inside one 16-lane row");
    const int lane = lane_id();
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k); mu = uni(mu);
    double M[KS];
#pragma unroll
    for (int j = 0; j < KS; ++j) M[j] = Hm[(lane < k ? lane : 0) * HP + (j < k ? j : 0)];
#pragma unroll
    for (int j = 0; j < KS; ++j) pin(M[j]);                       // unconditional loads, all in flight
#pragma unroll
    for (int j = 0; j < KS; ++j)
        M[j] = (lane < k && j < k) ? M[j] - (j == lane ? mu : 0.0) : (j == lane ? 1.0 : 0.0);
    double dp = 1.0;                                              // lane p keeps pivot p
    static_for<0, KS>([&](auto P) {
        constexpr int p = decltype(P)::value;
        const double d = row_bcast<p>(M[p]);
        dp = lane == p ? d : dp;
        const double nf = lane > p ? -(M[p] * rcp_nr(d)) : 0.0;
        static_for<p + 1, KS>([&](auto J) {
            constexpr int j = decltype(J)::value;
            M[j] = __builtin_fma(nf, row_bcast<p>(M[j]), M[j]);
        });
    });
    // pivots up to the first exact zero count individually; behind it everything counts as suspect
    const unsigned long long nonpos = __ballot(lane < k && !(dp > 0.0));
    const unsigned long long zero = __ballot(lane < k && dp == 0.0);
    if (zero) {
        const int p0 = __builtin_ctzll(zero);
        return __popcll(nonpos & ((2ull << p0) - 1ull)) + (k - p0 - 1);
    }
    return __popcll(nonpos);
}

// TRI: the system is read from the packed upper triangle the fused VALU pass leaves for a one-wave sample -- entry (r, c),
// r <= c, at c (c + 1) / 2 + r (be_dual_valu_dev.h) -- instead of from a k x (k + 1) matrix: no gather of the sums into a matrix
// in front of every Newton update, three index operations per entry here.  Same values, same elimination.
template <int KS, bool TRI = false>
__device__ __noinline__ StepResult newton_step_dpp(const double *Hm_, int HP, int k, int piv,
                                                   unsigned long long fmask, bool is_free, double g0) {
    static_assert(KS <= 16, "row broadcasts stay inside one 16-lane row");
    const int lane = lane_id();
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k); piv = uni(piv); fmask = uni(fmask);
    double M[KS + 1];
    const int rl = lane < k ? lane : 0;
    auto at = [&](int i, int j) -> int {                 // where H[i][j] lives
        if (!TRI) return i * HP + j;
        const int c = i > j ? i : j, r = i > j ? j : i;
        return c * (c + 1) / 2 + r;
    };
    double h_ip = Hm[at(rl, piv)], h_pp = Hm[at(piv, piv)];
#pragma unroll
    for (int j = 0; j < KS; ++j) M[j] = Hm[at(rl, j < k ? j : 0)];
    pin(h_ip);
    pin(h_pp);
#pragma unroll
    for (int j = 0; j < KS; ++j) pin(M[j]);                       // all loads in flight, none sunk into a branch
    // H0[i][j] = ((H[i][j] - keep_i H[j][piv]) - H[i][piv] keep_j) + H[piv][piv] keep_i keep_j, where H[j][piv]
    // is what lane j holds as its own H[.][piv]: one row broadcast instead of a second LDS read (and no second
    // register array -- this function's register need is what the caller has to spill around the call)
    static_for<0, KS>([&](auto J) {
        constexpr int j = decltype(J)::value;
        double hv = ((M[j] - row_bcast<j>(h_ip)) - h_ip) + h_pp;
        pin(hv);                                                  // plain select below, no exec-mask branch
        const bool use = j < k && is_free && ((fmask >> j) & 1ull);
        M[j] = use ? hv : (j == lane ? 1.0 : 0.0);
    });
    M[KS] = is_free ? -g0 : 0.0;
    double rinv = 1.0;                                            // lane p keeps 1 / pivot p
    bool bad = false;
    static_for<0, KS>([&](auto P) {
        constexpr int p = decltype(P)::value;
        const double d = row_bcast<p>(M[p]);
        bad |= !(d != 0.0);                                       // exact zero (or NaN): singular for LAPACK
        const double inv = rcp_nr(d);
        rinv = lane == p ? inv : rinv;
        const double nf = lane > p ? -(M[p] * inv) : 0.0;
        static_for<p + 1, KS + 1>([&](auto J) {
            constexpr int j = decltype(J)::value;
            M[j] = __builtin_fma(nf, row_bcast<p>(M[j]), M[j]);
        });
        __builtin_amdgcn_sched_barrier(0);     // keep the broadcasts of later pivots from being hoisted (registers)
    });
    if (__ballot(bad) & 0xffffull) return StepResult{0.0, 0};
    static_for<0, KS>([&](auto Q) {
        constexpr int p = KS - 1 - decltype(Q)::value;
        const double x = row_bcast<p>(M[KS] * rinv);
        M[KS] = lane == p ? x : (lane < p ? __builtin_fma(-M[p], x, M[KS]) : M[KS]);
    });
    return StepResult{is_free ? M[KS] : 0.0, 1};
}

// ---- KS > 16 (round 6): the same eliminations spread over the WHOLE wave ---------------------------------------------------
// Bundles of 17 .. 32 cuts used to fall back to one row per lane with `v_readlane` broadcasts (newton_step_ks: 12.4 k cycles
// at 17 cuts against 6.7 k for the DPP form at 16; inertia 10 k against 3.8 k -- tools/probes/dual_update_probe.hip), with 20 to 32
// of the 64 lanes holding a row of 20 .. 32 entries each.  Here the system is dealt over the four 16-lane DPP rows ("quarters"):
//   lane 16 q + r holds, of matrix rows r and 16 + r, the columns j = 4 s + q (slot s): KS / 4 entries per row instead of KS,
// so a pivot step is one row broadcast + one or two fused multiply-adds per SLOT (not per column), all four quarters working.
// What crosses the quarters is the multiplier column -(M[i][p] / d) -- formed by the quarter that owns column p, handed to the
// others by v_permlane16_swap + v_permlane32_swap (gfx950) -- and, in the back substitution, the four columns of a slot at a time.
// The right-hand side is kept in every quarter.  Every entry sees exactly the operations of newton_step_dpp / newton_step_ks in the
// same order (fma(-(M[i][p] inv), M[p][j], M[i][j]) per pivot, identity rows with multiplier -0), so the result is bit-identical.
__device__ __forceinline__ void swap16(double &x, double &y) {      // x = [x0 y0 x2 y2], y = [x1 y1 x3 y3] (quarters of the wave)
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void swap32(double &x, double &y) {      // x = [x0 x1 y0 y1], y = [x2 x3 y2 y3]
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
// every lane gets what lane 16 Q + (lane & 15) holds
template <int Q>
__device__ __forceinline__ double quarter_bcast(double v) {
    double a = v, b = v;
    swap16(a, b);                                    // a = [v0 v0 v2 v2], b = [v1 v1 v3 v3]
    double c = (Q & 1) ? b : a, d = c;
    swap32(c, d);                                    // c = lower half twice, d = upper half twice
    return Q < 2 ? c : d;
}
// the four quarters' values of v, in every lane: out[q] = what lane 16 q + (lane & 15) holds
__device__ __forceinline__ void quarter_gather(double v, double (&out)[4]) {
    double a = v, b = v;
    swap16(a, b);
    double a2 = a, b2 = b;
    swap32(a, a2);                                   // a = [v0 v0 v0 v0], a2 = [v2 ...]
    swap32(b, b2);                                   // b = [v1 ...],      b2 = [v3 ...]
    out[0] = a; out[1] = b; out[2] = a2; out[3] = b2;
}

template <int KS>
__device__ __noinline__ StepResult newton_step_2d(const double *Hm_, int HP, int k, int piv, unsigned long long fmask,
                                                  bool is_free, double g0) {
    static_assert(KS > 16 && KS <= 32 && KS % 4 == 0, "two matrix rows per lane, four columns per slot");
    constexpr int NS = KS / 4;
    const int lane = lane_id(), q = lane >> 4, r = lane & 15;
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k); piv = uni(piv); fmask = uni(fmask);
    const int i0 = r, i1 = 16 + r;                       // the two matrix rows of this lane
    const int rl0 = i0 < k ? i0 : 0, rl1 = i1 < k ? i1 : 0;
    // -g0 of the lane's two rows: row layout (lane i = row i) -> quarters 0 and 1 hold them
    const double gm = is_free ? -g0 : 0.0;
    double b0 = quarter_bcast<0>(gm), b1 = quarter_bcast<1>(gm);
    const bool free0 = (fmask >> i0) & 1ull, free1 = (fmask >> i1) & 1ull;
    double h_ip0 = Hm[rl0 * HP + piv], h_ip1 = Hm[rl1 * HP + piv], h_pp = Hm[piv * HP + piv];
    double M0[NS], M1[NS], hc[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = 4 * s + q, jc = j < k ? j : 0;
        M0[s] = Hm[rl0 * HP + jc];
        M1[s] = Hm[rl1 * HP + jc];
        hc[s] = Hm[jc * HP + piv];
    }
    pin(h_ip0); pin(h_ip1); pin(h_pp);
#pragma unroll
    for (int s = 0; s < NS; ++s) { pin(M0[s]); pin(M1[s]); pin(hc[s]); }      // all loads in flight, none sunk into a branch
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        // H0[i][j] = ((H[i][j] - keep_i H[j][piv]) - H[i][piv] keep_j) + H[piv][piv] keep_i keep_j
        const int j = 4 * s + q;
        const bool fj = j < k && ((fmask >> j) & 1ull);
        double hv0 = ((M0[s] - hc[s]) - h_ip0) + h_pp, hv1 = ((M1[s] - hc[s]) - h_ip1) + h_pp;
        pin(hv0); pin(hv1);
        M0[s] = (fj && free0) ? hv0 : (j == i0 ? 1.0 : 0.0);
        M1[s] = (fj && free1) ? hv1 : (j == i1 ? 1.0 : 0.0);
    }
    double rinv0 = 1.0, rinv1 = 1.0;                     // lane (p & 3, p & 15) keeps 1 / pivot p
    bool bad = false;
    static_for<0, KS>([&](auto P) {
        constexpr int p = decltype(P)::value, qp = p & 3, sp = p >> 2, rp = p & 15;
        constexpr bool upper = p >= 16;                  // the pivot row is one of the rows 16 + r
        const double d = row_bcast<rp>(upper ? M1[sp] : M0[sp]);          // (meaningful in quarter qp)
        bad |= q == qp && !(d != 0.0);                   // exact zero (or NaN): singular for LAPACK
        const double inv = rcp_nr(d);
        double nf0 = 0.0, nf1;
        if constexpr (!upper) {
            rinv0 = i0 == p ? inv : rinv0;
            nf0 = quarter_bcast<qp>(i0 > p ? -(M0[sp] * inv) : 0.0);
            nf1 = quarter_bcast<qp>(-(M1[sp] * inv));
        } else {
            rinv1 = i1 == p ? inv : rinv1;
            nf1 = quarter_bcast<qp>(i1 > p ? -(M1[sp] * inv) : 0.0);
        }
        static_for<sp, NS>([&](auto S) {                 // (columns <= p of slot sp take part: they are never read again)
            constexpr int s = decltype(S)::value;
            const double pr = row_bcast<rp>(upper ? M1[s] : M0[s]);
            if constexpr (!upper) M0[s] = __builtin_fma(nf0, pr, M0[s]);
            M1[s] = __builtin_fma(nf1, pr, M1[s]);
        });
        const double pb = row_bcast<rp>(upper ? b1 : b0);
        if constexpr (!upper) b0 = __builtin_fma(nf0, pb, b0);
        b1 = __builtin_fma(nf1, pb, b1);
        __builtin_amdgcn_sched_barrier(0);     // keep the broadcasts of later pivots from being hoisted (registers)
    });
    if (__ballot(bad)) return StepResult{0.0, 0};
    // back substitution, a slot (four columns, one per quarter) at a time: every quarter gets the slot's columns of its rows
    static_for<0, NS>([&](auto S) {
        constexpr int s = NS - 1 - decltype(S)::value;
        double u0[4], u1[4];
        quarter_gather(M0[s], u0);
        if constexpr (4 * s + 3 > 16) quarter_gather(M1[s], u1);
        static_for<0, 4>([&](auto C) {
            constexpr int qp = 3 - decltype(C)::value, p = 4 * s + qp, rp = p & 15;
            constexpr bool upper = p >= 16;
            const double x = bcast((upper ? b1 : b0) * (upper ? rinv1 : rinv0), 16 * qp + rp);
            b0 = i0 == p ? x : (i0 < p ? __builtin_fma(-u0[qp], x, b0) : b0);
            if constexpr (p >= 16) b1 = i1 == p ? x : (i1 < p ? __builtin_fma(-u1[qp], x, b1) : b1);
        });
    });
    // back to the row layout: lane i < 16 has its step in b0 (quarter 0), lane 16 + r in b1 -- which quarter 1 holds as well
    const double step = q == 0 ? b0 : (q == 1 ? b1 : 0.0);
    return StepResult{is_free ? step : 0.0, 1};
}

// Unpivoted LDL^T inertia in the same layout (cf. inertia_not_above_dpp): no right-hand side, no back substitution
template <int KS>
__device__ __noinline__ int inertia_not_above_2d(const double *Hm_, int HP, int k, double mu) {
    static_assert(KS > 16 && KS <= 32 && KS % 4 == 0, "two matrix rows per lane, four columns per slot");
    constexpr int NS = KS / 4;
    const int lane = lane_id(), q = lane >> 4, r = lane & 15;
    lds_cdouble *Hm = (lds_cdouble *)Hm_;
    HP = uni(HP); k = uni(k); mu = uni(mu);
    const int i0 = r, i1 = 16 + r;
    double M0[NS], M1[NS];
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = 4 * s + q, jc = j < k ? j : 0;
        M0[s] = Hm[(i0 < k ? i0 : 0) * HP + jc];
        M1[s] = Hm[(i1 < k ? i1 : 0) * HP + jc];
    }
#pragma unroll
    for (int s = 0; s < NS; ++s) { pin(M0[s]); pin(M1[s]); }
#pragma unroll
    for (int s = 0; s < NS; ++s) {
        const int j = 4 * s + q;
        M0[s] = (i0 < k && j < k) ? M0[s] - (j == i0 ? mu : 0.0) : (j == i0 ? 1.0 : 0.0);
        M1[s] = (i1 < k && j < k) ? M1[s] - (j == i1 ? mu : 0.0) : (j == i1 ? 1.0 : 0.0);
    }
    double dp0 = 1.0, dp1 = 1.0;                         // lane (p & 3, p & 15) keeps pivot p
    static_for<0, KS>([&](auto P) {
        constexpr int p = decltype(P)::value, qp = p & 3, sp = p >> 2, rp = p & 15;
        constexpr bool upper = p >= 16;
        const double d = row_bcast<rp>(upper ? M1[sp] : M0[sp]);
        const double inv = rcp_nr(d);
        double nf0 = 0.0, nf1;
        if constexpr (!upper) {
            dp0 = (q == qp && i0 == p) ? d : dp0;
            nf0 = quarter_bcast<qp>(i0 > p ? -(M0[sp] * inv) : 0.0);
            nf1 = quarter_bcast<qp>(-(M1[sp] * inv));
        } else {
            dp1 = (q == qp && i1 == p) ? d : dp1;
            nf1 = quarter_bcast<qp>(i1 > p ? -(M1[sp] * inv) : 0.0);
        }
        static_for<sp, NS>([&](auto S) {
            constexpr int s = decltype(S)::value;
            const double pr = row_bcast<rp>(upper ? M1[s] : M0[s]);
            if constexpr (!upper) M0[s] = __builtin_fma(nf0, pr, M0[s]);
            M1[s] = __builtin_fma(nf1, pr, M1[s]);
        });
        __builtin_amdgcn_sched_barrier(0);
    });
    // pivot p sits in lane 16 (p & 3) + (p & 15) as dp0 (p < 16) or dp1: ballots per pivot class, in pivot order
    // of every group of lanes r, 16 + r, 32 + r, 48 + r exactly one -- quarter r & 3 -- owns pivots r and 16 + r
    const bool own = (r & 3) == q;
    auto fold = [](unsigned long long m) -> unsigned { return (unsigned)((m | (m >> 16) | (m >> 32) | (m >> 48)) & 0xffffull); };
    unsigned nonpos = fold(__ballot(own && !(dp0 > 0.0))) | (fold(__ballot(own && !(dp1 > 0.0))) << 16);
    unsigned zero = fold(__ballot(own && dp0 == 0.0)) | (fold(__ballot(own && dp1 == 0.0)) << 16);
    const unsigned live = k >= 32 ? 0xffffffffu : ((1u << k) - 1u);
    nonpos &= live; zero &= live;
    // pivots up to the first exact zero count individually; behind it everything counts as suspect
    if (zero) {
        const int p0 = __builtin_ctz(zero);
        return __popc(nonpos & ((2u << p0) - 1u)) + (k - p0 - 1);
    }
    return __popc(nonpos);
}

// The statically unrolled routines above cost O(KS^2) predicated steps whatever k is, so they are
// instantiated for several sizes and the smallest one that holds the bundle is used.
template <int KT>
__device__ __forceinline__ int inertia_not_above(const double *Hm, int HP, int k, double mu) {
    if (k <= 4) return inertia_not_above_dpp<4>(Hm, HP, k, mu);
    if (k <= 6) return inertia_not_above_dpp<6>(Hm, HP, k, mu);
    if (k <= 8) return inertia_not_above_dpp<8>(Hm, HP, k, mu);
    if (k <= 10) return inertia_not_above_dpp<10>(Hm, HP, k, mu);
    if (k <= 12) return inertia_not_above_dpp<12>(Hm, HP, k, mu);
    if (KT == 16 || k <= 16) return inertia_not_above_dpp<16>(Hm, HP, k, mu);
    if constexpr (KT > 16) {                             // (round 6: the system dealt over the whole wave, above)
        if (k <= 20) return inertia_not_above_2d<20>(Hm, HP, k, mu);
        if (k <= 24) return inertia_not_above_2d<24>(Hm, HP, k, mu);
        return inertia_not_above_2d<32>(Hm, HP, k, mu);
    }
    return 0;
}
// (one-wave samples on the fused VALU pass: bundles of up to HV_K1MAX = 8 cuts, system in the pass's packed triangle)
__device__ __forceinline__ StepResult newton_step_tri(const double *P, int k, int piv, unsigned long long fmask, bool is_free,
                                                      double g0) {
    if (k <= 4) return newton_step_dpp<4, true>(P, 0, k, piv, fmask, is_free, g0);
    if (k <= 6) return newton_step_dpp<6, true>(P, 0, k, piv, fmask, is_free, g0);
    return newton_step_dpp<8, true>(P, 0, k, piv, fmask, is_free, g0);
}
template <int KT>
__device__ __forceinline__ StepResult newton_step(const double *Hm, int HP, int k, int piv, unsigned long long fmask,
                                                  bool is_free, double g0) {
    if (k <= 4) return newton_step_dpp<4>(Hm, HP, k, piv, fmask, is_free, g0);
    if (k <= 6) return newton_step_dpp<6>(Hm, HP, k, piv, fmask, is_free, g0);
    if (k <= 8) return newton_step_dpp<8>(Hm, HP, k, piv, fmask, is_free, g0);
    if (k <= 10) return newton_step_dpp<10>(Hm, HP, k, piv, fmask, is_free, g0);
    if (k <= 12) return newton_step_dpp<12>(Hm, HP, k, piv, fmask, is_free, g0);
    if (KT == 16 || k <= 16) return newton_step_dpp<16>(Hm, HP, k, piv, fmask, is_free, g0);
    if constexpr (KT > 16) {                             // (round 6: the system dealt over the whole wave, above)
        if (k <= 20) return newton_step_2d<20>(Hm, HP, k, piv, fmask, is_free, g0);
        if (k <= 24) return newton_step_2d<24>(Hm, HP, k, piv, fmask, is_free, g0);
        return newton_step_2d<32>(Hm, HP, k, piv, fmask, is_free, g0);
    }
    return StepResult{0.0, 0};
}

// Butterfly reduction over the first 16 lanes (the row that holds a bundle of up to 16 multipliers):
// quad permutes, row_half_mirror, row_mirror -- 4 DPP steps, every lane of the row ends up with the
// result (lane 0 is read back as the wave-uniform value).  Replaces k-step v_readlane scans.
template <typename Op>
__device__ __forceinline__ double row16_reduce(double v, Op op) {
    v = op(v, dpp_move<0xB1>(v));
    v = op(v, dpp_move<0x4E>(v));
    v = op(v, dpp_move<0x141>(v));
    v = op(v, dpp_move<0x140>(v));
    return uni(v);
}

// The same for bundles of up to 32 cuts (the 32-slot kernels): the rows of lanes 0..15 and 16..31 are reduced
// side by side, then combined.  `v` must be the operation's neutral element outside the bundle; k <= 16 (wave-uniform)
// takes the single-row result, so the 32-slot kernels run the 16-slot kernels' instruction sequence on small bundles.
template <int KT, typename Op>
__device__ __forceinline__ double rows_reduce(double v, int k, Op op) {
    v = op(v, dpp_move<0xB1>(v));
    v = op(v, dpp_move<0x4E>(v));
    v = op(v, dpp_move<0x141>(v));
    v = op(v, dpp_move<0x140>(v));
    if (KT == 16 || k <= 16) return uni(v);
    return op(bcast(v, 0), bcast(v, 16));
}

// a_j = sum_i lam_i A[i][j] for the columns j = tid + c * nt owned by this thread, NC of them in
// statically indexed registers: the LDS reads of several rows and all NC columns are in flight together
// and the NC transcendental chains that follow (exp, divide) interleave instead of running one after the
// other.  `fin(j, valid, a_j)` must do its arithmetic unconditionally and only guard its stores.
// Accumulation order over i is the plain sequential one (same bits as the scalar loop).
template <typename CutT, int NC, typename F>
__device__ __forceinline__ void columns_nc(const CutT *As, int ldA, int k, int zrow, int n_pad, int nt, int tid,
                                           double lam, F &&fin) {
    double acc[NC];
    int jc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = tid + c * nt;
        jc[c] = j < n_pad ? j : n_pad - 1;
        acc[c] = 0.0;
    }
    // rows in groups of four (4 * NC LDS reads in flight), then the remainder one by one -- the order of
    // the accumulation is the plain i = 0..k-1 one
    int i0 = 0;
    for (; i0 + 4 <= k; i0 += 4) {
        CutT av[4][NC];
        double li[4];
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            li[d] = bcast(lam, i0 + d);
#pragma unroll
            for (int c = 0; c < NC; ++c) av[d][c] = As[(i0 + d) * ldA + jc[c]];
        }
#pragma unroll
        for (int d = 0; d < 4; ++d)
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] += li[d] * (double)av[d][c];
    }
    for (; i0 < k; ++i0) {
        const double li = bcast(lam, i0);
#pragma unroll
        for (int c = 0; c < NC; ++c) acc[c] += li * (double)As[i0 * ldA + jc[c]];
    }
#pragma unroll
    for (int c = 0; c < NC; ++c) fin(tid + c * nt, tid + c * nt < n_pad, acc[c]);
}
template <typename CutT, typename F>
__device__ __forceinline__ void for_columns(const CutT *As, int ldA, int k, int zrow, int n_pad, int nt, int tid,
                                            double lam, F &&fin) {
    const int per_thread = (n_pad + nt - 1) / nt;
    if (per_thread == 3) columns_nc<CutT, 3>(As, ldA, k, zrow, n_pad, nt, tid, lam, fin);
    else if (per_thread <= 2) columns_nc<CutT, 2>(As, ldA, k, zrow, n_pad, nt, tid, lam, fin);
    else if (per_thread == 4) columns_nc<CutT, 4>(As, ldA, k, zrow, n_pad, nt, tid, lam, fin);
    else
        for (int j0 = 0; j0 < n_pad; j0 += 4 * nt)       // wide rows: four columns per thread at a time
            columns_nc<CutT, 4>(As + j0, ldA, k, zrow, n_pad - j0, nt, tid, lam,
                                [&](int j, bool valid, double aj) { fin(j0 + j, valid, aj); });
}

#include "be_dual_valu_dev.h"
#include "be_ipm_dev.h"

// NW = waves per sample.  NW = 1: one wave64 owns the sample (n up to a few hundred).  NW > 1 (large n,
// e.g. the 2048-pixel completion model): the columns are split over NW waves -- column phase, MFMA sweep
// (per-wave partial results summed through LDS) and y update scale with NW -- while every wave runs the
// small row-layout algebra redundantly on identical data, so no multiplier ever has to be exchanged.
// The body works on sample `u` with the 64 NW threads whose index is `tid`, in the LDS region `smem`; `round`
// and `rows` (the most bundle rows any sample can hold in this round) come from the caller; `crow` points at
// shared constant rows (zeros, ones) or is null, in which case the sample keeps its own behind its bundle.
// Stand-alone kernel below: one workgroup per sample.  be_fused.hip: one single-wave sample per wave of a
// 16-wave workgroup (NW = 1, so nothing in here synchronises beyond the wave).
// ArgsT: DualArgs as the kernel's by-value parameter, or a reference into the kernel-argument segment
// (address space 4) when the caller is a non-inlined phase function of be_fused.hip.
// IPM: the interior-point variant (lib/bundle_entropy.py): same cut bookkeeping and rank test, then y AND the
// multipliers come from pdipm_pc (be_ipm_dev.h) and multipliers <= 1e-8 are pruned (:234-237).
// GLB: the staged bundle (rows + the two constant rows) lives in the sample's slice of st.scratch instead of LDS -- the
// rounds of a wide-row solve whose bundle no longer fits the workgroup's 160 KB.  Same code, same arithmetic.
// LR > 0 (with GLB; dual_step_wide_kernel): split staging -- the LR oldest rows of the bundle are ALSO copied into LDS behind the
// carve-up, for the fused VALU pass of be_dual_valu_dev.h (everything else reads the device-memory copy as in any GLB round).
// SLICED = false: the caller never parks a sample (no update budget, state fresh from icnn_be_state_init: the persistent
// tile kernel in its default form) -- the park / resume paths and their live values are compiled out (headline solve 1.024 ->
// 1.016 ms on one box; cf. ICNN_BE_PROF, be_common.h).
template <typename CutT, int KT, int NW, bool RL, bool IPM = false, bool GLB = false, int LR = 0, bool SLICED = true, typename ArgsT>
__device__ __forceinline__ void dual_step_body(const ArgsT &a, int u, int tid, unsigned char *smem, int round,
                                               int rows_cap, const CutT *crow_shared) {
    constexpr int NT = 64 * NW;
    const auto &st = a.st;
    const int lane = tid & 63, wave = tid >> 6;
    const bool w0 = wave == 0;                 // the wave that writes per-sample results
    auto wg_any = [&](bool p) -> bool { return NW == 1 ? (bool)__any(p) : (bool)__syncthreads_or(p); };
    // reduce the per-wave partial contractions into Hm (no-op for a single wave)
    auto combine = [&](double *Hm_, const double *Hp_, int HP_, int k_, int ncols) {
        if (NW == 1) return;
        sample_sync<NW>();
        for (int e = tid; e < k_ * ncols; e += NT) {
            const int r = e / ncols, c = e - r * ncols;
            double acc = 0.0;
            for (int w = 0; w < NW; ++w) acc += Hp_[(w * rows_cap + r) * HP_ + c];
            Hm_[r * HP_ + c] = acc;
        }
    };
    long long tick = ICNN_BE_PROF_ON(a.prof) ? (long long)__builtin_readcyclecounter() : 0;
    const int T = st.slots;                               // bundle slots
    const int TI = st.iters > 0 ? st.iters : T;           // outer iterations (more than slots: slots are recycled, below)
    // The per-sample control words, the active-slot list and this thread's part of the new cut are requested
    // before the first branch: read one after the other behind the early exits they cost four dependent
    // memory round trips at the head of every launch.
    const int finished_u = st.finished[u], t_raw = st.t_next[u], phase_u = st.phase[u], cnt_raw = st.count[u];
    const int slot_pre = tid < T ? st.active[(size_t)u * T + tid] : 0;
    const int slot_lane = NW == 1 ? slot_pre : (lane < T ? st.active[(size_t)u * T + lane] : 0);   // per wave
    if (finished_u) return;
    // every sample carries its own outer-iteration counter: samples are independent, so one that
    // was parked mid-Newton simply lags behind the others (icnn_be.h, icnn_be_solve_fc)
    const int t = __builtin_amdgcn_readfirstlane(t_raw);
    if (t >= TI) return;
    const bool resume = SLICED && __builtin_amdgcn_readfirstlane(phase_u) != 0;

    const int n = st.n, n_pad = a.n_pad, ldA = a.ldA;
    const int HP = (rows_cap + 1) | 1;     // odd pitch of the (k x k+1) matrix H | A z in LDS
    // RL (variant of RL/src/bundle_entropy.py) is a template parameter: its Armijo line search, softplus
    // sums and pivot regularisation are compiled out of the dual-variant kernels.
    // a bundle cannot hold more cuts than outer iterations have been started: rows <= round + 1
    static_assert(!IPM || !RL, "interior point is a variant of its own");
    const Carve cv = carve(KT, rows_cap, ldA, n_pad, (int)sizeof(CutT), a.plan.n_leaves, RL, NW, crow_shared == nullptr, IPM, GLB);
    // this wave's share of the columns (multiple of 16)
    const int cchunk = NW == 1 ? n_pad : (((n_pad / 16 + NW - 1) / NW) * 16);
    const int cbeg = wave * cchunk < n_pad ? wave * cchunk : n_pad;
    const int cend = cbeg + cchunk < n_pad ? cbeg + cchunk : n_pad;
    CutT *As = GLB ? static_cast<CutT *>(st.scratch) + (size_t)u * (st.slots + 2) * ldA : reinterpret_cast<CutT *>(smem + cv.As);
    static_assert(LR == 0 || GLB, "split staging belongs to the device-memory rounds");
    CutT *AsL = reinterpret_cast<CutT *>(smem + ((cv.total + 15) & ~15));          // LR > 0: [LR][ldA], the oldest rows
    const bool mirror = LR > 0 && rows_cap <= HV_KMAX;     // split staging: the LR oldest rows also in LDS (there is room up to
                                                           // the carve-up of a HV_KMAX-cut bundle, dual_step_wide_kernel)
    double *zs = reinterpret_cast<double *>(smem + cv.zs);
    double *ws = reinterpret_cast<double *>(smem + cv.ws);
    double *sp = reinterpret_cast<double *>(smem + cv.sp);     // == ws unless RL
    double *Hm = reinterpret_cast<double *>(smem + cv.Hm);
    double *Hp = reinterpret_cast<double *>(smem + cv.Hp) + (NW == 1 ? 0 : wave * rows_cap * HP);   // my partial
    double *Hp0 = reinterpret_cast<double *>(smem + cv.Hp);
    double *leaf = Hm;                                        // pairwise-sum scratch aliases Hm
    double *psum = Hm + KT * a.plan.n_leaves;                 // [2*KT] results of pairwise sums
    int *slots = reinterpret_cast<int *>(smem + cv.ints);

    const CutT *g_row = static_cast<const CutT *>(a.g) + (size_t)u * n;
    // energy of the new cut: cut dtype, or float64 when the caller's fg returns it so (ICNN_BE_FLAG_F64_ENERGY)
    const double f_u = (sizeof(CutT) == 4 && (st.flags & ICNN_BE_FLAG_F64_ENERGY))
                           ? static_cast<const double *>(a.f)[u] : (double)static_cast<const CutT *>(a.f)[u];
    double *y_row = st.y + (size_t)u * n;
    CutT *G_u = static_cast<CutT *>(st.G) + (size_t)u * T * n;
    double *ys_u = st.ys + (size_t)u * T * n;
    double *h_u = st.h + (size_t)u * T;

    const int cnt = __builtin_amdgcn_readfirstlane(cnt_raw);
    const int k = cnt + 1;
    if (k > rows_cap) {                 // the active bundle no longer fits the staging area (wide rows only: the launch
        if (tid == 0) {                 // sizes it for min(round + 1, slots, what 160 KB hold)): stop at this iterate
            st.status[u] |= ICNN_BE_ST_OVERFLOW;
            st.finished[u] = 1;
            st.skip_fg[u] = 1;
        }
        return;
    }
    auto lap = [&](int phase) {                     // diagnostic only: cycles per phase, per sample
        if (ICNN_BE_PROF_ON(a.prof)) {
            const long long now = (long long)__builtin_readcyclecounter();
            // no-return atomic: fire and forget (a read-modify-write would bill its memory round trip
            // to the next phase)
            if (tid == 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(a.prof) + (size_t)u * DUAL_PROF_PHASES + phase,
                          (unsigned long long)(now - tick));
            tick = now;
        }
    };
    lap(12);                                        // the control words have arrived
    // Slot of the new cut: slot t while there are as many slots as iterations (the layout the host's per-iteration views
    // rely on).  With more iterations than slots (nIter > ICNN_BE_MAX_SLOTS: the reference has no cap, dual :129) a cut
    // takes the lowest slot that is not in the active list -- pruned cuts (dual :171-174) give their slots back; the
    // active list is only rewritten when an iteration completes, so a parked solve finds the same slot again.
    int slot_new = t;
    if (TI > T) {
        unsigned used = 0;
        for (int i = 0; i < cnt; ++i) used |= 1u << __builtin_amdgcn_readlane(slot_lane, i);
        slot_new = __builtin_ctz(~used);
    }
    if (tid < cnt) slots[tid] = slot_pre;
    if (tid == cnt) slots[tid] = slot_new;
    const double h_old = lane < cnt ? h_u[slot_lane] : 0.0;   // offsets of the older cuts, requested with the new cut's rows
    if (NW > 1) sample_sync<NW>();                    // other waves read the slot list

    // ---- 1. the new cut: slot t <- (g, h, y); 2. stage the older active rows -----------------
    // The global reads of both steps are issued back to back (new cut into registers, then the older
    // rows) so that their memory round trips overlap; the h reduction follows.
    const int per_row = (n_pad + NT - 1) / NT;                            // workgroup-wide chunks per row
    constexpr int MAXC = 4;
    auto stage_older = [&]() {
        if constexpr (NW == 1 && !GLB && LR == 0 && sizeof(CutT) == 4) {
            // Round 6: the older rows go from device memory STRAIGHT into LDS (global_load_lds_dword: lane l of a 64-column chunk
            // lands at chunk base + 4 l; M0 carries the chunk's LDS address), every chunk of every row requested back to back and
            // awaited once -- through registers, four rows per batch, a bundle of nine rows cost three memory round trips and
            // sixteen registers, seventeen rows five.  Columns n .. n_pad - 1 (no lane loads them) are zeroed by hand: the
            // region held another phase's data, and 0 x garbage must stay 0 in the sweeps.
            if (per_row <= MAXC) {
                for (int r = 0; r < cnt; ++r) {
                    const CutT *src = G_u + (size_t)uni(slots[r]) * n;
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int j = tid + c * 64;
                        if (c < per_row && j < n)
                            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void *)(src + j),
                                                             (__attribute__((address_space(3))) void *)(As + r * ldA + c * 64), 4, 0, 0);
                    }
                }
                for (int j = n + tid; j < n_pad; j += 64)
                    for (int r = 0; r < cnt; ++r) As[r * ldA + j] = (CutT)0;
                __builtin_amdgcn_s_waitcnt(0x0f70);                // vmcnt(0): the rows are in LDS
                return;
            }
        }
        if (per_row <= MAXC) {
            // four rows per batch, all their loads in flight before the first LDS store: a bundle of nine rows costs
            // three memory round trips (one chunk at a time with a 4-deep unroll it cost seven; eight rows per batch measured slower)
            constexpr int RB = 4;
            for (int r0 = 0; r0 < cnt; r0 += RB) {
                CutT v[RB][MAXC];
#pragma unroll
                for (int rr = 0; rr < RB; ++rr) {
                    const bool rok = r0 + rr < cnt;
                    const CutT *src = G_u + (size_t)(rok ? slots[r0 + rr] : 0) * n;
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int j = tid + c * NT;
                        v[rr][c] = rok && c < per_row && j < n ? src[j] : (CutT)0;
                    }
                }
#pragma unroll
                for (int rr = 0; rr < RB; ++rr)
#pragma unroll
                    for (int c = 0; c < MAXC; ++c) {
                        const int j = tid + c * NT;
                        if (r0 + rr < cnt && c < per_row && j < n_pad) {
                            As[(r0 + rr) * ldA + j] = v[rr][c];
                            if (LR > 0 && mirror && r0 + rr < LR) AsL[(r0 + rr) * ldA + j] = v[rr][c];
                        }
                    }
            }
            return;
        }
        const int chunks = cnt * per_row;
#pragma unroll 4
        for (int c = 0; c < chunks; ++c) {
            const int r = c / per_row, j = (c - r * per_row) * NT + tid;
            if (j < n_pad) {
                const CutT v = j < n ? G_u[(size_t)slots[r] * n + j] : (CutT)0;
                As[r * ldA + j] = v;
                if (LR > 0 && mirror && r < LR) AsL[r * ldA + j] = v;
            }
        }
    };
    double h_new;
    if (!resume) {
        bool bad = !isfinite(f_u);
        if (per_row <= MAXC) {
            CutT gr[MAXC];
            double yr[MAXC];
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int j = tid + c * NT;
                gr[c] = j < n ? g_row[j] : (CutT)0;
                yr[c] = j < n ? y_row[j] : 0.0;
            }
            stage_older();
#pragma unroll
            for (int c = 0; c < MAXC; ++c) {
                const int j = tid + c * NT;
                if (j < n) {
                    G_u[(size_t)slot_new * n + j] = gr[c];
                    ys_u[(size_t)slot_new * n + j] = yr[c];
                    bad |= !isfinite((double)gr[c]);
                }
                if (j < n_pad) {
                    As[cnt * ldA + j] = gr[c];
                    if (LR > 0 && mirror && cnt < LR) AsL[cnt * ldA + j] = gr[c];
                    sp[j] = (double)gr[c] * yr[c];                // dual :143  gi * x in float64 (0 for j >= n)
                }
            }
        } else {
            for (int j = tid; j < n_pad; j += NT) {
                double prod = 0.0;
                if (j < n) {
                    const CutT gj = g_row[j];
                    const double yj = y_row[j];
                    G_u[(size_t)slot_new * n + j] = gj;
                    ys_u[(size_t)slot_new * n + j] = yj;
                    prod = (double)gj * yj;                       // dual :143  gi * x in float64
                    bad |= !isfinite((double)gj);
                    As[cnt * ldA + j] = gj;
                    if (LR > 0 && mirror && cnt < LR) AsL[cnt * ldA + j] = gj;
                } else {
                    As[cnt * ldA + j] = (CutT)0;
                    if (LR > 0 && mirror && cnt < LR) AsL[cnt * ldA + j] = (CutT)0;
                }
                sp[j] = prod;
            }
            stage_older();
        }
        sample_sync<NW>();
        lap(13);
        np_pairwise_rows<NW, double>(a.plan, 1, [&](int, int j) { return sp[j]; }, leaf, psum, tid);
        h_new = f_u - psum[0];                        // fi - np.sum(gi * x)
        if (tid == 0) {
            h_u[slot_new] = h_new;
            if (st.fvals) st.fvals[(size_t)u * T + slot_new] = f_u;          // energy of the cut (callback replay, host)
        }
        if (wg_any(bad)) {
            if (tid == 0) { st.status[u] |= ICNN_BE_ST_NONFINITE; st.finished[u] = 1; st.skip_fg[u] = 1; }
            return;
        }
    } else {                                                  // parked solve: the cut is already in slot t
        for (int j = tid; j < n_pad; j += NT) {
            const CutT v = j < n ? G_u[(size_t)slot_new * n + j] : (CutT)0;
            As[cnt * ldA + j] = v;
            if (LR > 0 && mirror && cnt < LR) AsL[cnt * ldA + j] = v;
        }
        h_new = h_u[slot_new];
        stage_older();
    }
    lap(0);
    const CutT *crow = crow_shared ? crow_shared : As + rows_cap * ldA;
    if (!crow_shared)
        for (int j = tid; j < ldA; j += NT) { As[rows_cap * ldA + j] = (CutT)0; As[(rows_cap + 1) * ldA + j] = (CutT)1; }
    const double h_i = lane < cnt ? h_old : h_new;                // row layout (lane < k)
    // Wide rows, small bundle: column phase and contraction fused on the VALU (be_dual_valu_dev.h).  The per-wave partial
    // sums live where z would (two buffers, alternating by update: one barrier per update instead of four).
    const bool hv_on = !(st.flags & ICNN_BE_FLAG_MFMA_CONTRACTION);
    const int hv_p = hv_pitch(k);                          // partial sums per wave (two buffers of NW rows where z and w would live)
    // Round 4: one-wave samples with rows of up to 192 columns (float32 cuts) take the same pass -- three columns per lane,
    // the k (k + 3) / 2 sums through the transposing butterfly, H | A z written by the lanes that end up with them: no z / w
    // round trip through LDS, no operand gathers (the MFMA sweep reads ten times the bundle per update and is what the CU's
    // LDS bandwidth bounds when sixteen samples share it).  Every kernel that runs dual_step_body takes the same decision,
    // so the dispatch paths stay bit-identical; ICNN_BE_FLAG_MFMA_CONTRACTION keeps the sweep everywhere.
    // Up to HV_K1MAX = 8 cuts: the pass costs k (k + 3) / 2 sums per column and their reduction, the sweep a fixed number of
    // MFMAs -- measured crossover on the long solves (4096 x 30: sweep only 6.65 ms, pass up to 8 cuts 6.50, up to 16 7.24;
    // 4096 x 15: 1.97 / 1.76 / 1.78).
    const bool valu = hv_on && (NW > 1 || (n_pad <= 192 && sizeof(CutT) == 4 && k <= HV_K1MAX)) && !RL && !IPM && k >= 2 &&
                      k <= HV_KMAX && (LR == 0 || mirror) && n_pad <= 256 * NW && NW * hv_p <= n_pad;
    double *hv_part = zs;
    const HvEntry hv_first = hv_entry<true>(lane, k, HP);  // this lane's first entry of the per-update gather
    sample_sync<NW>();

    lap(1);
    // ---- 3. rank test (variant DUAL only) -----------------------------------------------
    if (!RL && !resume) {
        bool deficient = false;
        const double cfac = (double)(k > n ? k : n) * Cut<CutT>::eps;   // max(M.shape) * eps
        if (k == 1) {
            bool nz = false;
            for (int j = tid; j < n; j += NT) nz |= As[j] != (CutT)0;
            deficient = !wg_any(nz);
        } else if (sizeof(CutT) == 8 && NW == 1) {
            // float64 cuts: one-sided Jacobi on the rows (in place; restaged afterwards)
            for (int sweep = 0; sweep < 30; ++sweep) {
                bool rotated = false;
                for (int p = 0; p < k - 1; ++p)
                    for (int q = p + 1; q < k; ++q) {
                        double al = 0, be = 0, ga = 0;
                        for (int j = lane; j < n; j += 64) {
                            const double vp = (double)As[p * ldA + j], vq = (double)As[q * ldA + j];
                            al += vp * vp; be += vq * vq; ga += vp * vq;
                        }
                        al = wave_sum(al); be = wave_sum(be); ga = wave_sum(ga);
                        if (ga == 0.0 || fabs(ga) <= 2.3e-16 * sqrt(al * be)) continue;
                        rotated = true;
                        const double zeta = (be - al) / (2.0 * ga);
                        const double tt = zeta != 0.0
                            ? copysign(1.0, zeta) / (fabs(zeta) + sqrt(1.0 + zeta * zeta)) : 1.0;
                        const double c = 1.0 / sqrt(1.0 + tt * tt), s = c * tt;
                        for (int j = lane; j < n; j += 64) {
                            const double vp = (double)As[p * ldA + j], vq = (double)As[q * ldA + j];
                            As[p * ldA + j] = (CutT)(c * vp - s * vq);
                            As[q * ldA + j] = (CutT)(s * vp + c * vq);
                        }
                    }
                if (!rotated) break;
            }
            double sv = 0.0, smax = 0.0;
            for (int r = 0; r < k; ++r) {
                double ss = 0;
                for (int j = lane; j < n; j += 64) { const double v = (double)As[r * ldA + j]; ss += v * v; }
                ss = sqrt(wave_sum(ss));
                if (lane == r) sv = ss;
                smax = fmax(smax, ss);
            }
            deficient = __popcll(__ballot(lane < k && sv > smax * cfac)) < k;
            sample_sync<NW>();
            for (int r = 0; r < k; ++r) {                 // restage
                const CutT *src = r < cnt ? G_u + (size_t)slots[r] * n : g_row;
                for (int j = lane; j < n_pad; j += 64) As[r * ldA + j] = j < n ? src[j] : (CutT)0;
            }
            sample_sync<NW>();
        } else {
            if (valu) {                                    // (be_dual_valu_dev.h)
                hv_column_pass_k<CutT, NW, false, LR, GLB, (NW == 1 && KT > 16)>(As, AsL, ldA, k, rows_cap, n, n_pad, tid, 0.0, hv_part + wave * hv_p);
                sample_sync<NW>();
                hv_gather<NW, false>(hv_part, hv_p, Hm, HP, k, tid, NT, hv_entry<false>(tid, k, HP));   // the whole sample: the shared copy
            } else {
                contract_mfma<CutT, KT, false>(As, ldA, k, crow, cbeg, cend, ws, zs, Hp, HP);
                combine(Hm, Hp0, HP, k, k);
            }
            sample_sync<NW>();
            // brackets lo <= lambda_max <= hi, replicated in every lane
            double trace = 0, total = 0, dmax = 0, rmax = 0;
            {
                double diag = 0, rsum = 0, rabs = 0;
                if (lane < k) {
                    diag = Hm[lane * HP + lane];
                    for (int c = 0; c < k; ++c) { const double v = Hm[lane * HP + c]; rsum += v; rabs += fabs(v); }
                }
                for (int i = 0; i < k; ++i) {
                    const double di = bcast(diag, i);
                    trace += di; dmax = fmax(dmax, di);
                    total += bcast(rsum, i); rmax = fmax(rmax, bcast(rabs, i));
                }
            }
            const double lo = fmax(dmax, total / (double)k);     // Rayleigh quotients <= lambda_max
            const double hi = fmin(trace, rmax);                 // trace, Gershgorin  >= lambda_max
            const double c2 = cfac * cfac;
            if (inertia_not_above<KT>(Hm, HP, k, c2 * hi) == 0) {
                deficient = false;
            } else if (inertia_not_above<KT>(Hm, HP, k, c2 * lo) > 0) {
                deficient = true;
            } else {
                jacobi_lane0<KT, NW>(Hm, HP, k, tid);
                const double ev = lane < k ? fmax(Hm[lane * HP + lane], 0.0) : 0.0;
                const double svv = sqrt(ev), smax = wave_max(svv);
                deficient = __popcll(__ballot(lane < k && svv > smax * cfac)) < k;
            }
            sample_sync<NW>();
        }
        if (deficient) {                                   // dual :156-161
            if (tid == 0) { st.finished[u] = 1; st.n_iters[u] = t - 1; st.skip_fg[u] = 1; }
            return;
        }
    }

    lap(2);
    // ---- 4. multipliers (row layout: lane i < k holds lam_i) -----------------------------------
    double lam = 0.0;
    int updates = 0, updates_before = 0;
    if constexpr (IPM) {
        int ipm_status = 0;
        if constexpr (NW > 1) {
            // wide rows (round 5): the columns over the NW waves (ipm_solve_waves; GLB: bundle rows from device memory); what it
            // does not cover -- more than IPM_KMAX_WAVES cuts -- runs the one-wave solve on wave 0 while the others wait
            const int kp = k < 2 ? 2 : hv_padded(k);
            const bool waves_ok = sizeof(CutT) == 4 && k <= IPM_KMAX_WAVES && NW * ((ipm_nv(kp) + 3) & ~3) <= n_pad;
            if (waves_ok) {
                lam = ipm_solve_waves<CutT, KT, NW, GLB>(As, ldA, k, n, n_pad, ws, zs, sp, reinterpret_cast<double *>(smem + cv.yv),
                                                    reinterpret_cast<double *>(smem + cv.dv), Hp, HP, h_i, tid, &ipm_status, lap);
            } else {
                if (w0) {
                    lam = ipm_solve<CutT, KT, GLB>(As, ldA, k, crow, n, n_pad, ws, zs, sp, reinterpret_cast<double *>(smem + cv.yv),
                                                   reinterpret_cast<double *>(smem + cv.dv), Hm, HP, h_i, lane, &ipm_status, lap);
                    if (lane == 0) slots[KT - 1] = ipm_status;      // (slot list: at most KT - 1 entries)
                }
                sample_sync<NW>();
                ipm_status = slots[KT - 1];
            }
        } else
        lam = ipm_solve<CutT, KT, GLB>(As, ldA, k, crow, n, n_pad, ws, zs, sp, reinterpret_cast<double *>(smem + cv.yv),
                                  reinterpret_cast<double *>(smem + cv.dv), Hm, HP, h_i, lane, &ipm_status, lap);
        if (ipm_status) {                                  // numpy.linalg.cholesky raises (:42): the caller sees LinAlgError
            if (tid == 0) { st.status[u] |= ICNN_BE_ST_SINGULAR; st.finished[u] = 1; st.skip_fg[u] = 1; }
            return;
        }
    } else if (k == 1) {
        lam = lane == 0 ? 1.0 : 0.0;                       // dual :167
    } else {
        // c = np.sum(A, axis=1) + b with the row sum in the cut dtype (dual :18)
        CutT *rowsum = reinterpret_cast<CutT *>(psum);
        np_pairwise_rows<NW, CutT>(a.plan, k, [&](int r, int j) { return As[r * ldA + j]; },
                                   reinterpret_cast<CutT *>(leaf), rowsum, tid);
        const double c_i = lane < k ? (double)rowsum[lane] + h_i : 0.0;
        sample_sync<NW>();
        lap(3);
        const int cap = RL ? 20 : 100;                     // rl :29 / dual :30
        const int backoff_cap = RL ? 10 : 50;              // rl :65 / dual :67
        const bool shortcut = !(st.flags & ICNN_BE_FLAG_NO_CYCLE_SHORTCUT);
        lam = lane < k ? 1.0 / (double)k : 0.0;            // dual :26
        double prev1 = 0.0, prev2 = 0.0, prev3 = 0.0, prev4 = 0.0;
        int hist = 0, last_jump = -1000;                   // valid previous iterates (<= 4); update of the last jump
        double d4_prev = -1.0;                             // max|lam_{t-1} - lam_{t-5}| of the previous update; -1: not available
        bool abort_sample = false, parked = false;
        int upd0 = 0;                                      // updates done in earlier rounds
        double *park = st.park + (size_t)u * (5 * T + 4);
        if (resume) {
            updates = upd0 = updates_before = uni((int)park[5 * T]);
            hist = uni((int)park[5 * T + 1]);
            last_jump = uni((int)park[5 * T + 2]);
            d4_prev = uni(park[5 * T + 3]);
            if (lane < k) {
                lam = park[lane]; prev1 = park[T + lane]; prev2 = park[2 * T + lane]; prev3 = park[3 * T + lane];
                prev4 = park[4 * T + lane];
            }
        }
        int budget = a.budget > 0 ? a.budget : cap;

        while (updates < cap) {
            if (SLICED && budget-- <= 0) { parked = true; break; }
            // a = A^T lam, z = sigmoid(a), w = z (1 - z)                     dual :32-33
            if (valu) {
                double *part = hv_part + (updates & 1) * (NW * hv_p);
                hv_column_pass_k<CutT, NW, true, LR, GLB, (NW == 1 && KT > 16)>(As, AsL, ldA, k, rows_cap, n, n_pad, tid, lam, part + wave * hv_p);
                sample_sync<NW>();
                lap(4);
                if (NW > 1) hv_gather<NW, true>(part, hv_p, Hp, HP, k, lane, 64, hv_first);  // this wave's own copy of H | A z
                lap(5);
            } else {
                for_columns<CutT>(As, ldA, k, rows_cap, n_pad, NT, tid, lam, [&](int j, bool valid, double aj) {
                    double z = sigmoid_fast(aj);
                    double w = z * (1.0 - z);
                    if (j >= n) { z = 0.0; w = 0.0; }
                    if (valid) {
                        zs[j] = z;
                        ws[j] = w;
                        if (RL) sp[j] = j < n ? softplus_fast(aj) : 0.0;
                    }
                });
                sample_sync<NW>();
                lap(4);
                contract_mfma<CutT, KT, true>(As, ldA, k, crow, cbeg, cend, ws, zs, Hp, HP, lap);
                combine(Hm, Hp0, HP, k, k + 1);
                sample_sync<NW>();
                lap(5);
            }
            const double *Hq = valu ? Hp : Hm;                                 // the system this wave reads
            // one wave per sample on the pass: there is no other wave's share to add, the sums are read where the pass left
            // them -- the packed upper triangle (k <= 8: the instance's size is the bundle's), A z behind it
            static_assert(HV_K1MAX <= 8 && hv_padded(HV_K1MAX) == HV_K1MAX, "the packed triangle of a k-cut bundle is the k-cut instance's");
            const bool tri = NW == 1 && valu;
            const int hvT = k * (k + 1) / 2;

            const double grad = lane < k ? -c_i + (tri ? hv_part[(updates & 1) * (NW * hv_p) + hvT + lane] : Hq[lane * HP + k]) : 0.0;     // dual :35
            // first maximum of lam (:39), replicated scan
            int piv_v = 0;
            {
                const double mx = rows_reduce<KT>(lane < k ? lam : -1e300, k, [](double x, double y) { return fmax(x, y); });
                const unsigned long long at = __ballot(lane < k && lam == mx);
                piv_v = at ? __builtin_ctzll(at) : 0;
            }
            const int piv = __builtin_amdgcn_readfirstlane(piv_v);
            const bool is_piv = lane == piv;
            const double red = is_piv ? 1.0 : lam;                               // :40-41
            const double keep = is_piv ? 0.0 : 1.0;                              // :42
            const double g0 = grad - keep * bcast(grad, piv);                    // :44
            const bool bound = is_piv || (red <= BOUND_EPS && g0 > 0.0);         // :48-49
            const bool is_free = lane < k && !bound;
            const unsigned long long fmask = __ballot(is_free);
            const double nrm2 = rows_reduce<KT>(is_free ? g0 * g0 : 0.0, k, [](double x, double y) { return x + y; });
            if (sqrt(nrm2) < GRAD_TOL) break;                                    // :50 -> return lam

            lap(8);
            const StepResult sr = tri ? newton_step_tri(hv_part + (updates & 1) * (NW * hv_p), k, piv, fmask, is_free, g0)
                                      : newton_step<KT>(Hq, HP, k, piv, fmask, is_free, g0);
            const double step = sr.step;
            if (!__builtin_amdgcn_readfirstlane(sr.ok)) {
                if (tid == 0) st.status[u] |= ICNN_BE_ST_SINGULAR;
                if (!RL) abort_sample = true;              // dual :63 raises
                break;                                     // rl :62 keeps lam
            }

            lap(9);
            double tt = 1.0;                                                     // dual :66
            double fval = 0.0, slope = 0.0;
            if (RL) {
                double dmax = 0.0;
                for (int i = 0; i < k; ++i) dmax = fmax(dmax, fabs(bcast(step, i)));
                tt = fmin(1.0 / dmax, 1.0);                                      // rl :64
                sample_sync<NW>();
                np_pairwise_rows<NW, double>(a.plan, 1, [&](int, int j) { return sp[j]; }, leaf, psum, tid);
                double cl = 0.0;
                for (int i = 0; i < k; ++i) { cl += bcast(c_i * lam, i); slope += bcast(step * g0, i); }
                fval = -cl + psum[0];                                            // :34
                sample_sync<NW>();
            }
            double lam_new = lam;
            bool returned = false;
            for (int bt = 0; bt < backoff_cap; ++bt) {
                const double trial = is_piv ? 1.0 : fmax(red + tt * step, 0.0);  // :68-69
                const double s = rows_reduce<KT>((lane < k && !is_piv) ? trial : 0.0, k,              // e.dot(y_n)
                                                 [](double x, double y) { return x + y; });
                const double lam_p = 1.0 - s;                                    // :71
                lam_new = lane < k ? (is_piv ? lam_p : trial) : 0.0;
                bool accept = false;
                if (lam_p >= 0.0) {
                    if (RL) {                                                    // rl :71-74
                        for (int j = tid; j < n_pad; j += NT) {
                            double aj = 0.0;
                            for (int i = 0; i < k; ++i) aj += bcast(lam_new, i) * (double)As[i * ldA + j];
                            sp[j] = j < n ? softplus_fast(aj) : 0.0;
                        }
                        sample_sync<NW>();
                        np_pairwise_rows<NW, double>(a.plan, 1, [&](int, int j) { return sp[j]; }, leaf, psum, tid);
                        double cl = 0.0;
                        for (int i = 0; i < k; ++i) cl += bcast(c_i * lam_new, i);
                        const double f_new = -cl + psum[0];
                        sample_sync<NW>();
                        accept = f_new < fval + tt * ARMIJO_ALPHA * slope;
                    } else {
                        accept = true;
                    }
                }
                if (accept) break;
                if (RL) {
                    double mv = 0.0;
                    for (int i = 0; i < k; ++i) mv = fmax(mv, tt * fabs(bcast(step, i)));
                    if (mv < TINY) { returned = true; break; }                   // rl :77
                } else if (tt < TINY) { returned = true; break; }                // dual :79
                tt *= 0.5;
            }
            ++updates;
            if (returned) { lam = lam_new; break; }
            if (shortcut && hist >= 1) {
                if (!__any(fabs(lam_new - prev1) > CYCLE_TOL)) { lam = lam_new; break; }
                if (hist >= 2 && !__any(fabs(lam_new - prev2) > CYCLE_TOL)) {
                    lam = ((cap - updates) & 1) ? prev1 : lam_new;
                    break;
                }
                // period 3: lam_cap = lam_{updates + r}, r = (cap - updates) mod 3, and lam_{t+1} = lam_{t-2}
                if (hist >= 3 && !__any(fabs(lam_new - prev3) > CYCLE_TOL)) {
                    const int r = (cap - updates) % 3;
                    lam = r == 0 ? lam_new : (r == 1 ? prev2 : prev1);
                    break;
                }
                // period 4: lam_{t+1} = lam_{t-3}
                if (hist >= 4 && !__any(fabs(lam_new - prev4) > CYCLE_TOL)) {
                    const int r = (cap - updates) & 3;
                    lam = r == 0 ? lam_new : (r == 1 ? prev3 : (r == 2 ? prev2 : prev1));
                    break;
                }
                // the same periods at their rounding floor (NOISE_TOL above).  Stateless but for one scalar: "an update cycle
                // earlier" is formed from the four stored iterates (period 3 compares with ONE update earlier, lam_{t-1} -
                // lam_{t-4}; period 4 with its own value of the previous update, which is parked with the iterates).
                if (!RL && updates >= NOISE_T0 && hist >= 2) {
                    const bool in = lane < k;
                    const auto mx = [](double x, double y) { return fmax(x, y); };
                    const double d1 = rows_reduce<KT>(in ? fabs(lam_new - prev1) : 0.0, k, mx);
                    if (d1 <= NOISE_TOL && d1 >= rows_reduce<KT>(in ? fabs(prev1 - prev2) : 0.0, k, mx)) { lam = lam_new; break; }
                    if (hist >= 4) {
                        const double d2 = rows_reduce<KT>(in ? fabs(lam_new - prev2) : 0.0, k, mx);
                        if (d2 <= NOISE_TOL && d2 >= rows_reduce<KT>(in ? fabs(prev2 - prev4) : 0.0, k, mx)) {
                            lam = ((cap - updates) & 1) ? prev1 : lam_new;
                            break;
                        }
                        const double d3 = rows_reduce<KT>(in ? fabs(lam_new - prev3) : 0.0, k, mx);
                        if (d3 <= NOISE_TOL && d3 >= rows_reduce<KT>(in ? fabs(prev1 - prev4) : 0.0, k, mx)) {
                            const int r = (cap - updates) % 3;
                            lam = r == 0 ? lam_new : (r == 1 ? prev2 : prev1);
                            break;
                        }
                        const double d4 = rows_reduce<KT>(in ? fabs(lam_new - prev4) : 0.0, k, mx);
                        if (d4 <= NOISE_TOL && d4_prev >= 0.0 && d4 >= d4_prev) {
                            const int r = (cap - updates) & 3;
                            lam = r == 0 ? lam_new : (r == 1 ? prev3 : (r == 2 ? prev2 : prev1));
                            break;
                        }
                        d4_prev = d4;
                    } else {
                        d4_prev = -1.0;
                    }
                }
            }
            // Slow 2-cycles (contraction 0.6-0.7 per update: 35-85 updates until lam_t - lam_{t-2} <= 1e-13; a
            // handful per 40 000 solves, each of which holds its whole tile or launch): once the even and the
            // odd subsequence converge geometrically, both are extrapolated to their limits (one step of the
            // vector Aitken / Anderson-1 formula with a common ratio) and the iteration goes on from there.  The
            // stopping rule above is unchanged, so what is returned is still an iterate that repeats within
            // 1e-13 after two updates, i.e. a point of the limit cycle the reference's 100 updates end on.
            bool jumped = false;
            if (shortcut && hist >= 4 && updates >= ACCEL_T0 && updates - last_jump >= ACCEL_GAP) {
                const bool in = lane < k;
                const double d_t = in ? lam_new - prev2 : 0.0, d_p = in ? prev2 - prev4 : 0.0;
                const auto mx = [](double x, double y) { return fmax(x, y); };
                const auto ad = [](double x, double y) { return x + y; };
                const double n1 = rows_reduce<KT>(fabs(d_t), k, mx), n0 = rows_reduce<KT>(fabs(d_p), k, mx);
                if (n1 > 0.0 && n1 < ACCEL_D2MAX && n1 < n0) {
                    const double ratio = rows_reduce<KT>(d_t * d_p, k, ad) / rows_reduce<KT>(d_p * d_p, k, ad);
                    // ratio < 0 (each subsequence oscillates around its limit): only where the reference's own remaining
                    // updates would close the gap anyway, |lam_cap - limit| ~ n1 |ratio|^((cap - t) / 2) <= ACCEL_NEG_RESID
                    bool take = ratio > 0.0 && ratio < ACCEL_RMAX;
                    if (ratio < 0.0 && -ratio < ACCEL_RMAX) {
                        double left = n1;
                        for (int m = (cap - updates) / 2; m > 0; --m) left *= -ratio;
                        take = left <= ACCEL_NEG_RESID;
                    }
                    if (take) {
                        const double gain = ratio / (1.0 - ratio);
                        const double xe = lam_new + d_t * gain;                       // limit of lam_t, lam_{t+-2}, ..
                        const double xo = in ? prev1 + (prev1 - prev3) * gain : 0.0;  // limit of lam_{t-1}, ..
                        if (!__any(in && (xe < 0.0 || xo < 0.0))) {
                            lam_new = xe;                // lam_t := xe, lam_{t-1} := xo; older history is void
                            prev1 = xe;
                            prev2 = xo;
                            hist = 2;
                            d4_prev = -1.0;
                            last_jump = updates;
                            jumped = true;
                        }
                    }
                }
            }
            if (!jumped) {
                prev4 = prev3;
                prev3 = prev2;
                prev2 = prev1;
                prev1 = lam_new;
                hist = hist < 4 ? hist + 1 : 4;
            }
            lam = lam_new;                                                       // :84
            if (!valu) sample_sync<NW>();       // Hm / zs / ws are rewritten by the next iteration
            lap(6);
        }
        if (abort_sample) {
            if (tid == 0) { st.finished[u] = 1; st.skip_fg[u] = 1; }
            return;
        }
        if (parked) {                                      // continue in the next round
            if (w0 && lane < k) {
                park[lane] = lam; park[T + lane] = prev1; park[2 * T + lane] = prev2; park[3 * T + lane] = prev3;
                park[4 * T + lane] = prev4;
            }
            if (tid == 0) {
                park[5 * T] = (double)updates;
                park[5 * T + 1] = (double)hist;
                park[5 * T + 2] = (double)last_jump;
                park[5 * T + 3] = d4_prev;
                st.newton_iters[u] += updates - upd0;
                st.phase[u] = 1;
                st.skip_fg[u] = 1;
                st.pending[round] = 1;   // plain store: only "any work left" is needed, and a
                                           // same-address atomic per sample costs ~13 ns each (50 us per launch)
            }
            return;
        }
    }

    lap(6);
    // The bookkeeping below re-reads its pointers from the kernel-argument segment through an opaque
    // pointer: otherwise a dozen 64-bit pointers stay live in SGPRs across the whole Newton loop, whose
    // scalar registers then spill (v_readlane/v_writelane traffic on the critical path).
    typedef const __attribute__((address_space(4))) DualArgs KernArgs;   // the by-value argument, in place
    KernArgs *ea;
    if constexpr (std::is_same<ArgsT, DualArgs>::value) ea = (KernArgs *)__builtin_amdgcn_kernarg_segment_ptr();
    else ea = (KernArgs *)uni((unsigned long long)&a);    // already a reference into the kernel-argument segment
    asm volatile("" : "+s"(ea));
    KernArgs &eargs = *ea;
    const auto &es = eargs.st;
    // ---- 5. y <- sigmoid(-A^T lam), bookkeeping ---------------------------------------
    double *ey_row = es.y + (size_t)u * n;
    double move = 0.0;
    bool nonfinite = false;
    auto commit = [&](int j, double ynew) {
        if (RL) {
            ynew = fmin(fmax(ynew, 0.03), 0.97);                   // rl :118,:123
            move = fmax(move, fabs(ey_row[j] - ynew));
        }
        nonfinite |= !isfinite(ynew);
        ey_row[j] = ynew;
    };
    if constexpr (IPM) {
        const double *yv = reinterpret_cast<const double *>(smem + cv.yv);
        for (int j = tid; j < n; j += NT) commit(j, yv[j]);         // lib/bundle_entropy.py:225: x[u] = y of pdipm_pc
    } else if (k == 1) {
        for (int j = tid; j < n; j += NT)
            commit(j, (double)Cut<CutT>::sigmoid_neg(As[j]));      // dual :168, cut-dtype arithmetic
    } else {
        for_columns<CutT>(As, ldA, k, rows_cap, n_pad, NT, tid, lam, [&](int j, bool, double aj) {
            const double ynew = 1.0 / (1.0 + exp(aj));             // dual :165
            if (j < n) commit(j, ynew);
        });
    }
    bool fin = false;
    if (RL && wave_max(move) < 1e-6) fin = true;                        // rl :125-126 (NW == 1 only)
    if (wg_any(nonfinite)) { fin = true; if (tid == 0) es.status[u] |= ICNN_BE_ST_NONFINITE; }

    const bool pos = lane < k && lam > (IPM ? 1e-8 : 0.0);           // dual :171-174; lib/bundle_entropy.py:234-237
    const unsigned long long pmask = __ballot(pos);
    if (pos && w0) {
        const int at = __popcll(pmask & ((1ull << lane) - 1ull));
        es.active[(size_t)u * T + at] = slots[lane];
        es.lam[(size_t)u * T + at] = lam;
    }
    if (tid == 0) {
        es.count[u] = __popcll(pmask);
        es.newton_iters[u] += updates - updates_before;
        if (fin) es.finished[u] = 1;
        const bool more = !fin && t + 1 < TI;
        es.t_next[u] = t + 1;
        es.phase[u] = 0;
        es.skip_fg[u] = more ? 0 : 1;
        if (more) es.pending[round] = 1;   // plain store: only "any work left" is needed, and a
                                           // same-address atomic per sample costs ~13 ns each (50 us per launch)
    }
    lap(7);
}

// Wide rows past the LDS capacity (n = 2048: 12 cuts), the rounds in which a bundle could outgrow it: the bundle is staged in
// device memory as in any GLB round, and -- split staging -- its WIDE_LR oldest rows are mirrored in LDS for the fused VALU pass,
// which re-reads them there and keeps the few younger rows in registers: per Newton update a sample with 15 cuts streams 3 rows
// from L2 instead of 15 (256 samples x 123 KB do not fit the L2s, the GLB rounds were bound by that traffic).  The mirror fits
// next to the carve-up of a bundle of up to HV_KMAX cuts; a sample beyond that (rows_cap = every row the round allows) runs the
// plain GLB body: no mirror, MFMA sweep.  Same arithmetic whichever way a sample is staged: no bit of its result changes.
constexpr int WIDE_LR = 12;
__global__ __launch_bounds__(512, 1) void dual_step_wide_kernel(DualArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int k = __builtin_amdgcn_readfirstlane(a.st.count[blockIdx.x]) + 1;
    const int mid = a.rows < HV_KMAX ? a.rows : HV_KMAX;
    dual_step_body<float, 32, 8, false, false, true, WIDE_LR>(a, blockIdx.x, threadIdx.x, smem, a.round, k <= mid ? mid : a.rows, nullptr);
}

template <typename CutT, int KT, int NW, bool RL, bool IPM = false, bool GLB = false>
__global__ __launch_bounds__(64 * NW, NW == 1 ? (RL && KT > 16 ? 2 : 4) : 1) void dual_step_kernel(DualArgs a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    dual_step_body<CutT, KT, NW, RL, IPM, GLB>(a, blockIdx.x, threadIdx.x, smem, a.round, a.rows, nullptr);
}


}  // namespace
}  // namespace icnn_be
