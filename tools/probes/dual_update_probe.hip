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

int main() {
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
