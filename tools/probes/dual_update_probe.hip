// Cycles of the pieces of one Newton update of the dual step (be_dual_dev.h) as a function of the bundle size k: the very
// device functions of the kernels, timed in isolation on a lone wave (W = 1) and with eight waves per CU (W = 8), each wave on
// a bundle of its own in LDS.  Round 6: what does an update cost at 13..24 cuts, where BASELINE configs[3] spends its rounds?
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -I include -I icnn_amd/csrc tools/probes/dual_update_probe.hip -o /tmp/dup && /tmp/dup
#include "be_dual_dev.h"
#include <cstdio>
#include <vector>
using namespace icnn_be;

__device__ __forceinline__ long long now() {
    __builtin_amdgcn_sched_barrier(0);
    long long t = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}

constexpr int N = 159, NPAD = 160, LDA = 162, KT = 32, NSLOT = 8;

// (exactly representable values: the host's reference sums see the very operands the device does)
__host__ __device__ inline float a_val(int r, int c, int k) { return c < N ? (float)((r * 7 + c * 13) % 31 - 15 + (r == c % k ? 32 : 0)) * 0.015625f : 0.f; }
__host__ __device__ inline double z_val(int i) { return (double)(i + 3) * 0.00390625; }
__host__ __device__ inline double w_val(int i) { return i < N ? (double)(1 + i % 7) * 0.03125 : 0.0; }

__global__ __launch_bounds__(512) void probe(double *out, double *sink, int k, int per_wave_bytes, double *hdump) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = thread_id() >> 6, lane = thread_id() & 63;
    unsigned char *base = smem + wave * per_wave_bytes;
    const int HP = (k + 1) | 1;
    float *As = reinterpret_cast<float *>(base);                           // k rows + zeros + ones
    double *zs = reinterpret_cast<double *>(base + (((k + 2) * LDA * 4 + 15) & ~15));
    double *ws = zs + NPAD;
    double *Hm = ws + NPAD;
    for (int i = lane; i < k * LDA; i += 64) { const int r = i / LDA, c = i % LDA; As[i] = a_val(r, c, k); }
    for (int i = lane; i < LDA; i += 64) { As[k * LDA + i] = 0.f; As[(k + 1) * LDA + i] = 1.f; }
    for (int i = lane; i < NPAD; i += 64) { zs[i] = z_val(i); ws[i] = w_val(i); }
    sample_sync<1>();
    if (hdump && blockIdx.x == 0 && wave == 0) {               // correctness: H | A z and the Gram matrix against the host's sums
        contract_mfma<float, KT, true>(As, LDA, k, As + k * LDA, 0, NPAD, ws, zs, Hm, HP);
        sample_sync<1>();
        for (int e = lane; e < k * (k + 1); e += 64) hdump[e] = Hm[(e / (k + 1)) * HP + e % (k + 1)];
        sample_sync<1>();
        contract_mfma<float, KT, false>(As, LDA, k, As + k * LDA, 0, NPAD, ws, zs, Hm, HP);
        sample_sync<1>();
        for (int e = lane; e < k * k; e += 64) hdump[1024 + e] = Hm[(e / k) * HP + e % k];
        sample_sync<1>();
        // the whole-wave eliminations against the one-row-per-lane ones: bit for bit
        if (k > 16) {
            contract_mfma<float, KT, true>(As, LDA, k, As + k * LDA, 0, NPAD, ws, zs, Hm, HP);
            sample_sync<1>();
            const double grad = lane < k ? Hm[lane * HP + k] : 0.0;
            int diff = 0;
            for (int trial = 0; trial < 3; ++trial) {
                const int piv = trial == 0 ? 0 : (trial == 1 ? k - 1 : 7);
                unsigned long long fm = ((k >= 64 ? ~0ull : (1ull << k) - 1ull)) & ~(1ull << piv);
                if (trial == 2) fm &= ~((1ull << 3) | (1ull << 17) | (1ull << (k - 2)));       // some bound rows
                const bool fr = (fm >> lane) & 1ull;
                const StepResult a = k <= 20 ? newton_step_ks<20>(Hm, HP, k, piv, fm, fr, grad) : (k <= 24 ? newton_step_ks<24>(Hm, HP, k, piv, fm, fr, grad) : newton_step_ks<32>(Hm, HP, k, piv, fm, fr, grad));
                const StepResult b = newton_step<KT>(Hm, HP, k, piv, fm, fr, grad);
                diff += (__double_as_longlong(a.step) != __double_as_longlong(b.step)) || a.ok != b.ok;
            }
            contract_mfma<float, KT, false>(As, LDA, k, As + k * LDA, 0, NPAD, ws, zs, Hm, HP);
            sample_sync<1>();
            int idiff = 0;
            for (double mu : {1e-9, 0.5, 3.0, 1e3}) {
                const int a = k <= 20 ? inertia_not_above_ks<20>(Hm, HP, k, mu) : (k <= 24 ? inertia_not_above_ks<24>(Hm, HP, k, mu) : inertia_not_above_ks<32>(Hm, HP, k, mu));
                const int b = inertia_not_above<KT>(Hm, HP, k, mu);
                idiff += a != b;
                if (lane == 0) hdump[2040 + (mu > 100 ? 3 : mu > 1 ? 2 : mu > 0.1 ? 1 : 0)] = a * 100 + b;
            }
            const unsigned long long dm = __ballot(diff != 0);
            if (lane == 0) { hdump[2046] = (double)__popcll(dm); hdump[2047] = idiff; }
        }
    }
    const float *crow = As + k * LDA;
    double lam = lane < k ? 1.0 / k : 0.0, acc = 0.0;
    long long t[NSLOT + 1];
    for (int rep = 0; rep < 3; ++rep) {
        t[0] = now();
        for_columns<float>(As, LDA, k, k, NPAD, 64, lane, lam, [&](int j, bool valid, double aj) {
            double z = sigmoid_fast(aj), w = z * (1.0 - z);
            if (j >= N) { z = 0.0; w = 0.0; }
            if (valid) { zs[j] = z + z_val(j); ws[j] = w; }
        });
        sample_sync<1>();
        t[1] = now();
        contract_mfma<float, KT, true>(As, LDA, k, crow, 0, NPAD, ws, zs, Hm, HP);
        sample_sync<1>();
        t[2] = now();
        const double grad = lane < k ? Hm[lane * HP + k] : 0.0;
        const unsigned long long fmask = ((1ull << k) - 1ull) & ~1ull;
        const StepResult sr = newton_step<KT>(Hm, HP, k, 0, fmask, lane > 0 && lane < k, grad);
        acc += sr.step;
        sample_sync<1>();
        t[3] = now();
        contract_mfma<float, KT, false>(As, LDA, k, crow, 0, NPAD, ws, zs, Hm, HP);
        sample_sync<1>();
        t[4] = now();
        acc += inertia_not_above<KT>(Hm, HP, k, 1e-9);
        sample_sync<1>();
        t[5] = now();
        double mx = rows_reduce<KT>(lane < k ? lam + acc * 1e-30 : -1e300, k, [](double x, double y) { return fmax(x, y); });
        mx += rows_reduce<KT>(lane < k ? lam : 0.0, k, [](double x, double y) { return x + y; });
        acc += mx * 1e-30;
        t[6] = now();
        pin(acc);
    }
    if (lane == 0)
        for (int s = 0; s < 6; ++s) out[((size_t)blockIdx.x * (blockDim.x >> 6) + wave) * NSLOT + s] = (double)(t[s + 1] - t[s]);
    sink[(size_t)blockIdx.x * blockDim.x + thread_id()] = acc + Hm[lane % (k * HP)];
}

// ---- Record of a withdrawn attempt (round 6): the Hessian sweep on SEVERAL waves, bit for bit. --------------
// The maps below restate the two tiles of contract_mfma_cover2 (be_dual_dev.h); the probe asserts that the blocked sweep on
// eight waves reproduces the one-wave sweep in every entry for 9 .. 31 cuts.  It did (profiles/r06_blocked_sweep_probe.txt), and
// the cooperative continuation of parked Newton solves built on it was bit-identical on every shape -- and slower
// (profiles/r06_coop_continuation_withdrawn.txt), so neither is in the library.
namespace icnn_be { namespace {
constexpr int MF_ZC = 64, MF_NONE = 65;                    // B-operand column codes besides a bundle row: A z, nothing
// The two tiles as maps (shared with the several-wave form of the sweep, contract_mfma_blocked): A-operand row and B-operand column
// of position r16 of tile t, and where a result (bundle row `row`, column code `cb`) goes.
__device__ __forceinline__ int cover2_row(int t, int r16) { return t == 0 ? r16 : 4 + r16; }
template <bool HESS>
__device__ __forceinline__ int cover2_col(int t, int r16) {
    if (t == 0) return HESS ? (r16 < 15 ? r16 : MF_ZC) : r16;
    if (r16 < 4) return 16 + r16;
    if (HESS) return r16 == 4 ? MF_ZC : (r16 < 9 ? r16 - 5 : (r16 == 9 ? 15 : MF_NONE));
    return r16 < 8 ? r16 - 4 : MF_NONE;
}
template <bool HESS>
__device__ __forceinline__ void cover2_store(int t, int row, int cb, int k, double v, double *Hm, int HP) {
    if (row >= k) return;
    if (cb == MF_ZC) {
        if (t == 0 || row >= 16) Hm[row * HP + k] = v;
    } else if (t == 0) {
        Hm[row * HP + cb] = v;                                                 // (rows 0 .. 15) x (columns 0 .. 14 | 15)
        if (HESS && row == 15) Hm[cb * HP + 15] = v;                           // H[i][15] = H[15][i]
    } else if (cb < k) {
        if (cb >= 16) {                                                        // (i, e), i = 4 .. 19
            Hm[row * HP + cb] = v;
            if (row < 16) Hm[cb * HP + row] = v;
        } else if (cb < 4) {                                                   // (e, i), i = 0 .. 3, and its mirror
            if (row >= 16) { Hm[row * HP + cb] = v; Hm[cb * HP + row] = v; }
        } else if (row == 15) {                                                // cb == 15: the diagonal entry
            Hm[15 * HP + 15] = v;
        }
    }
}


// The same sums on SEVERAL waves, bit for bit (round 6: the cooperative Newton updates of a long solve, be_fused.hip).  A 16 x 16
// tile of the one-wave sweep is sixteen 4 x 4 blocks; v_mfma_f64_4x4x4_4b computes four independent 4 x 4 blocks per instruction
// with the very k-ordered fma chain per entry that v_mfma_f64_16x16x4 applies (tools/probes/mfma_f64_order_probe.hip), so the
// blocks of the SAME tiles -- same operand roles per entry, same pair of accumulator chains over the even and the odd k-steps --
// can be dealt to `nw` waves: instruction j = (tile j / 4, block row j % 4, the four block columns), wave w takes j = w, w + nw, ..
// Every entry is written by exactly one lane of one wave, through the store rules of the one-wave sweep.  Wave-uniform control.
template <typename CutT, int KT, bool HESS>
__device__ void contract_mfma_blocked(const CutT *As, int ldA, int k, const CutT *crow, int n_pad, const double *ws,
                                      const double *zs, double *Hm, int HP, int wave, int nw) {
    static_assert(KT > 16, "two accumulator chains per entry, as in the 32-slot form of contract_mfma");
    const int lane = thread_id() & 63, kq = lane >> 4, blk = (lane >> 2) & 3, r = lane & 3;
    const int ncolsB = HESS ? k + 1 : k;
    const bool cover = k >= 17 && k <= 20;
    const int nt_r = (k + 15) >> 4, nt_c = (ncolsB + 15) >> 4;            // tiles of the (ti, tj) tiling: ti < nt_r, ti <= tj < nt_c
    const int ntile = cover ? 2 : (nt_r == 1 ? nt_c : 2 * nt_c - 1);
    n_pad = uni(n_pad);
    for (int j = uni(wave); j < 4 * ntile; j += nw) {
        const int t = j >> 2, br = j & 3;
        int ti = 0, tj = t;                                               // (ti, tj) of tile t: (0, 0 .. nt_c - 1), then (1, 1 ..)
        if (!cover && t >= nt_c) { ti = 1; tj = t - nt_c + 1; }
        const int r16a = 4 * br + r, r16b = 4 * blk + r;
        int ra, cb;
        if (cover) { ra = cover2_row(t, r16a); cb = cover2_col<HESS>(t, r16b); }
        else { ra = ti * 16 + r16a; cb = tj * 16 + r16b; if (HESS && cb == k) cb = MF_ZC; }
        const bool zcol = cb == MF_ZC;
        const CutT *pa = (ra < k ? As + ra * ldA : crow) + kq;
        const CutT *pb = (cb < k ? As + cb * ldA : (zcol ? crow + ldA : crow)) + kq;
        const double *pwz = (zcol ? zs : ws) + kq;
        double acc0 = 0.0, acc1 = 0.0;
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
        auto stage = [&](int cnext, CutT (&ca)[4], CutT (&cb_)[4], double (&cw)[4], CutT (&na)[4], CutT (&nb)[4], double (&nw_)[4]) {
            gather(cnext, na, nb, nw_);
            __builtin_amdgcn_sched_barrier(0);
            double av[4], bv[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                av[s] = (double)ca[s];
                bv[s] = HESS ? (double)cb_[s] * cw[s] : (double)cb_[s];
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av[s], bv[s], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av[s], bv[s], acc0, 0, 0, 0);
            }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int s = 0; s < 4; ++s) { pin(na[s]); pin(nb[s]); if (HESS) pin(nw_[s]); }
        };
        gather(0, xa, xb, xw);
        __builtin_amdgcn_s_waitcnt(0xc07f);
        const int clast = n_pad - 16;
        for (int c0 = 0; c0 < n_pad; c0 += 32) {
            stage(c0 + 16 < n_pad ? c0 + 16 : clast, xa, xb, xw, ya, yb, yw);
            if (c0 + 16 < n_pad) stage(c0 + 32 < n_pad ? c0 + 32 : clast, ya, yb, yw, xa, xb, xw);
        }
        const double v = acc0 + acc1;                                    // (even chain + odd chain, as `acc += acc_odd`)
        // result lane: D_block[lane >> 4][lane & 3] of block `blk`
        const int row16 = 4 * br + kq, col16 = 4 * blk + r;
        if (cover) {
            cover2_store<HESS>(t, cover2_row(t, row16), cover2_col<HESS>(t, col16), k, v, Hm, HP);
        } else {
            const int row = ti * 16 + row16, col = tj * 16 + col16;
            if (row < k && col < ncolsB) {
                Hm[row * HP + col] = v;
                if (tj != ti && col < k) Hm[col * HP + row] = v;
            }
        }
    }
}

}}
// the several-wave sweep (contract_mfma_blocked) against the one-wave sweep, bit for bit: one bundle, eight waves
__global__ __launch_bounds__(512) void probe_blocked(int k, double *res) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
    const int wave = thread_id() >> 6, lane = thread_id() & 63, tid = thread_id();
    const int HP = (k + 1) | 1;
    float *As = reinterpret_cast<float *>(smem);
    double *zs = reinterpret_cast<double *>(smem + (((k + 2) * LDA * 4 + 15) & ~15));
    double *ws = zs + NPAD, *H1 = ws + NPAD, *H2 = H1 + 33 * 35;
    // (irrational-looking values: every product and sum rounds)
    for (int i = tid; i < k * LDA; i += 512) { const int rr = i / LDA, c = i % LDA; As[i] = c < N ? 0.37f * __sinf(0.7f * rr + 1.3f * c) + 0.011f * c : 0.f; }
    for (int i = tid; i < LDA; i += 512) { As[k * LDA + i] = 0.f; As[(k + 1) * LDA + i] = 1.f; }
    for (int i = tid; i < NPAD; i += 512) { zs[i] = 0.3 + 0.0137 * i; ws[i] = i < N ? 0.05 + 0.19 / (1.0 + 0.1 * i) : 0.0; }
    for (int i = tid; i < 33 * 35; i += 512) { H1[i] = -1.0; H2[i] = -2.0; }
    __syncthreads();
    long long t0 = 0, t1 = 0, t2 = 0;
    if (wave == 0) { t0 = now(); contract_mfma<float, KT, true>(As, LDA, k, As + k * LDA, 0, NPAD, ws, zs, H1, HP); t1 = now(); }
    __syncthreads();
    const long long t3 = now();
    contract_mfma_blocked<float, KT, true>(As, LDA, k, As + k * LDA, NPAD, ws, zs, H2, HP, wave, 8);
    __syncthreads();
    t2 = now();
    int diff = 0;
    for (int e = tid; e < k * (k + 1); e += 512) {
        const int i = e / (k + 1), j = e % (k + 1);
        diff += __double_as_longlong(H1[i * HP + j]) != __double_as_longlong(H2[i * HP + j]);
    }
    diff = __syncthreads_count(diff);
    if (tid == 0) { res[0] = diff; res[1] = (double)(t1 - t0); res[2] = (double)(t2 - t3); res[3] = H1[1 * HP + 0]; res[4] = H2[1 * HP + 0]; }
}

int main() {
    {
        double *res, h[5];
        (void)hipMalloc(&res, 64);
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(probe_blocked), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        for (int k : {9, 12, 15, 16, 17, 19, 20, 21, 24, 28, 31}) {
            probe_blocked<<<1, 512, ((k + 2) * LDA * 4 + 16) + 2 * NPAD * 8 + 2 * 33 * 35 * 8>>>(k, res);
            (void)hipMemcpy(h, res, 40, hipMemcpyDeviceToHost);
            printf("k = %2d: blocked sweep on 8 waves differs from the one-wave sweep in %d of %d entries (H[1][0] %.17g / %.17g); one wave %.0f cycles, eight waves %.0f\n",
                   k, (int)h[0], k * (k + 1), h[3], h[4], h[1], h[2]);
        }
    }
    const char *names[6] = {"column phase (a, z, w)", "mfma H | A z", "Newton solve", "mfma Gram (rank test)", "inertia (one call)", "2 row reductions"};
    double *out, *sink, *hdump;
    hipMalloc(&hdump, 2048 * sizeof(double));
    hipMalloc(&out, 256 * 8 * NSLOT * sizeof(double));
    hipMalloc(&sink, 256 * 512 * sizeof(double));
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    for (int W : {1, 8}) {
        printf("---- %d wave(s) per CU, 256 workgroups ----\n%-26s", W, "k");
        const int ks[] = {9, 12, 15, 16, 17, 20, 21, 24};
        for (int k : ks) printf("%8d", k);
        printf("\n");
        std::vector<std::vector<double>> res(6);
        std::vector<double> checks, asyms;
        for (int k : ks) {
            const int hp = (k + 1) | 1;
            const int per_wave = ((((k + 2) * LDA * 4 + 15) & ~15) + 2 * NPAD * 8 + k * hp * 8 + 255) & ~255;
            if (W * per_wave > 160 * 1024) { for (auto &r : res) r.push_back(-1); continue; }
            hipMemset(out, 0, 256 * 8 * NSLOT * sizeof(double));
            probe<<<256, 64 * W, W * per_wave>>>(out, sink, k, per_wave, W == 1 ? hdump : nullptr);
            if (W == 1) {
                std::vector<double> hd(2048);
                hipMemcpy(hd.data(), hdump, 2048 * sizeof(double), hipMemcpyDeviceToHost);
                double worst = 0, asym = 0;
                for (int i = 0; i < k; ++i)
                    for (int j = 0; j <= k; ++j) {
                        long double ref = 0, gram = 0;
                        for (int c = 0; c < N; ++c) {
                            const long double a = a_val(i, c, k), b = j < k ? (long double)a_val(j, c, k) : 1.0L;
                            ref += a * b * (j < k ? (long double)w_val(c) : (long double)z_val(c));
                            gram += a * b;
                        }
                        worst = fmax(worst, fabs((double)(hd[i * (k + 1) + j] - ref)) / fmax(1e-30, fabs((double)ref)));
                        if (j < k) {
                            worst = fmax(worst, fabs((double)(hd[1024 + i * k + j] - gram)) / fmax(1e-30, fabs((double)gram)));
                            asym = fmax(asym, fabs(hd[i * (k + 1) + j] - hd[j * (k + 1) + i]));
                        }
                    }
                checks.push_back(worst); asyms.push_back(asym);
                if (k > 16) printf("k = %d: whole-wave Newton step differs from the one-row-per-lane one in %d lanes, inertia in %d of 4 shifts (ks*100+2d: %g %g %g %g)\n", k, (int)hd[2046], (int)hd[2047], hd[2040], hd[2041], hd[2042], hd[2043]);
            }
            std::vector<double> h(256 * W * NSLOT);
            hipMemcpy(h.data(), out, h.size() * sizeof(double), hipMemcpyDeviceToHost);
            for (int s = 0; s < 6; ++s) {
                double m = 0;
                for (int i = 0; i < 256 * W; ++i) m += h[(size_t)i * NSLOT + s];
                res[s].push_back(m / (256 * W));
            }
        }
        for (int s = 0; s < 6; ++s) {
            printf("%-26s", names[s]);
            for (double v : res[s]) printf("%8.0f", v);
            printf("\n");
        }
        if (!checks.empty()) {
            printf("%-26s", "H, Gram: max rel err");
            for (double v : checks) printf(" %7.0e", v);
            printf("\n%-26s", "H: max |H - H^T|");
            for (double v : asyms) printf(" %7.0e", v);
            printf("\n");
        }
    }
    return 0;
}
