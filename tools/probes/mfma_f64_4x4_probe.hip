// Probe of the v_mfma_f64_4x4x4_4b_f64 operand/result lane layout (GPU box only).
// run r < 16: A is one-hot at in-block lane r (same in all 4 blocks), B[l] = 1 + l.
// run 16: random-ish A, B for a full check.  Output: out[17][64] doubles, then A and B of run 16.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ void probe(double *out, const double *a16, const double *b16) {
    const int l = threadIdx.x;
    for (int r = 0; r < 16; ++r) {
        double a = ((l & 15) == r) ? 1.0 : 0.0;
        double b = 1.0 + l;
        double d = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, 0.0, 0, 0, 0);
        out[r * 64 + l] = d;
    }
    out[16 * 64 + l] = __builtin_amdgcn_mfma_f64_4x4x4f64(a16[l], b16[l], 0.0, 0, 0, 0);
}
int main() {
    double *out, *a, *b;
    hipMalloc(&out, 17 * 64 * 8); hipMalloc(&a, 512); hipMalloc(&b, 512);
    std::vector<double> ha(64), hb(64), ho(17 * 64);
    for (int i = 0; i < 64; ++i) { ha[i] = 1 + (i * 7) % 11; hb[i] = 2 + (i * 5) % 13; }
    hipMemcpy(a, ha.data(), 512, hipMemcpyHostToDevice); hipMemcpy(b, hb.data(), 512, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, a, b);
    hipMemcpy(ho.data(), out, 17 * 64 * 8, hipMemcpyDeviceToHost);
    for (int r = 0; r < 17; ++r) { printf("run %d:", r); for (int l = 0; l < 64; ++l) printf(" %g", ho[r * 64 + l]); printf("\n"); }
    return 0;
}
