// In which order does v_mfma_f64_4x4x4_4b_f64 (and v_mfma_f64_16x16x4_f64) add its four k-products to C?  (GPU box only.)
// The small-n dual step (be_dual_small.hip) recomputes on the VALU what the wave-per-sample kernel gets from the MFMA
// sweep, and the two are compared bit for bit; that needs the MFMA's internal association.  Candidates evaluated on the host
// with fma(): ascending chain from C, descending chain, pairwise, unfused.  Prints how many of the 64 x TRIALS results
// each candidate reproduces exactly.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe4(const double *a, const double *b, const double *c, double *d, int trials) {
    const int l = threadIdx.x;
    for (int t = 0; t < trials; ++t)
        d[t * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a[t * 64 + l], b[t * 64 + l], c[t * 64 + l], 0, 0, 0);
}
__global__ void probe16(const double *a, const double *b, const double *c, double *d, int trials) {
    const int l = threadIdx.x;
    for (int t = 0; t < trials; ++t) {
        d4 acc = {c[(t * 64 + l) * 4 + 0], c[(t * 64 + l) * 4 + 1], c[(t * 64 + l) * 4 + 2], c[(t * 64 + l) * 4 + 3]};
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a[t * 64 + l], b[t * 64 + l], acc, 0, 0, 0);
        for (int r = 0; r < 4; ++r) d[(t * 64 + l) * 4 + r] = acc[r];
    }
}
static double rnd() {
    const double m = (double)rand() / RAND_MAX * 2.0 - 1.0;
    return std::ldexp(m, rand() % 24 - 12);
}
int main() {
    const int T = 256;
    std::vector<double> a(T * 64), b(T * 64), c(T * 64 * 4), d(T * 64 * 4);
    for (auto &v : a) v = rnd();
    for (auto &v : b) v = rnd();
    for (auto &v : c) v = rnd();
    double *da, *db, *dc, *dd;
    hipMalloc(&da, a.size() * 8); hipMalloc(&db, b.size() * 8); hipMalloc(&dc, c.size() * 8); hipMalloc(&dd, d.size() * 8);
    hipMemcpy(da, a.data(), a.size() * 8, hipMemcpyHostToDevice); hipMemcpy(db, b.data(), b.size() * 8, hipMemcpyHostToDevice);
    // ---- 4x4x4, four blocks: A lane (kq = l>>4, blk = (l>>2)&3, r = l&3) holds A_blk[r][kq], B lane holds B_blk[kq][r],
    //      result lane holds D_blk[l>>4][l&3]
    std::vector<double> c1(T * 64);
    for (int i = 0; i < T * 64; ++i) c1[i] = c[i];
    hipMemcpy(dc, c1.data(), c1.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe4, dim3(1), dim3(64), 0, 0, da, db, dc, dd, T);
    hipMemcpy(d.data(), dd, T * 64 * 8, hipMemcpyDeviceToHost);
    int hit[4] = {0, 0, 0, 0};
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < 64; ++l) {
            const int i = l >> 4, blk = (l >> 2) & 3, j = l & 3;
            double p[4], q[4];
            for (int k = 0; k < 4; ++k) { p[k] = a[t * 64 + k * 16 + blk * 4 + i]; q[k] = b[t * 64 + k * 16 + blk * 4 + j]; }
            const double c0 = c1[t * 64 + l], got = d[t * 64 + l];
            double asc = c0, desc = c0;
            for (int k = 0; k < 4; ++k) asc = std::fma(p[k], q[k], asc);
            for (int k = 3; k >= 0; --k) desc = std::fma(p[k], q[k], desc);
            const double pair = (std::fma(p[0], q[0], p[1] * q[1]) + std::fma(p[2], q[2], p[3] * q[3])) + c0;
            const double unf = (((c0 + p[0] * q[0]) + p[1] * q[1]) + p[2] * q[2]) + p[3] * q[3];
            hit[0] += asc == got; hit[1] += desc == got; hit[2] += pair == got; hit[3] += unf == got;
        }
    printf("mfma_f64_4x4x4: of %d results  ascending fma chain %d  descending %d  pairwise %d  unfused %d\n", T * 64, hit[0], hit[1],
           hit[2], hit[3]);
    // ---- 16x16x4: A lane (r16 = l&15, q = l>>4) holds A[r16][q], B lane holds B[q][r16], result lane holds D[q + 4 r][r16]
    hipMemcpy(dc, c.data(), c.size() * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe16, dim3(1), dim3(64), 0, 0, da, db, dc, dd, T);
    hipMemcpy(d.data(), dd, d.size() * 8, hipMemcpyDeviceToHost);
    int h16[2] = {0, 0};
    for (int t = 0; t < T; ++t)
        for (int l = 0; l < 64; ++l)
            for (int r = 0; r < 4; ++r) {
                const int row = (l >> 4) + 4 * r, col = l & 15;
                double asc = c[(t * 64 + l) * 4 + r], desc = asc;
                for (int k = 0; k < 4; ++k) asc = std::fma(a[t * 64 + k * 16 + row], b[t * 64 + k * 16 + col], asc);
                for (int k = 3; k >= 0; --k) desc = std::fma(a[t * 64 + k * 16 + row], b[t * 64 + k * 16 + col], desc);
                h16[0] += asc == d[(t * 64 + l) * 4 + r]; h16[1] += desc == d[(t * 64 + l) * 4 + r];
            }
    printf("mfma_f64_16x16x4: of %d results  ascending fma chain %d  descending %d\n", T * 256, h16[0], h16[1]);
    return 0;
}
