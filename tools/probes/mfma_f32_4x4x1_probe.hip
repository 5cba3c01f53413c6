// Probe of v_mfma_f32_4x4x1_16b_f32 on gfx950 (GPU box only): operand / result lane layout and issue rate.
// run r (0..63): A is one-hot at lane r, B[l] = 1 + l -> which (lane, register) of D receives which B value.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
__global__ void probe(float *out, long long *cycles) {
    const int l = threadIdx.x;
    for (int r = 0; r < 64; ++r) {
        const float a = l == r ? 1.f : 0.f, b = 1.f + l;
        f4 d = {0.f, 0.f, 0.f, 0.f};
        d = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, d, 0, 0, 0);
        for (int v = 0; v < 4; ++v) out[(r * 64 + l) * 4 + v] = d[v];
    }
    // timing: 256 instructions on one accumulator (dependent) and on four (independent)
    f4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
    float a = 1.f + l, b = 2.f;
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < 256; ++i) c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
    long long t1 = __builtin_readcyclecounter();
    for (int i = 0; i < 64; ++i) {
        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c0, 0, 0, 0);
        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c1, 0, 0, 0);
        c2 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c2, 0, 0, 0);
        c3 = __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c3, 0, 0, 0);
    }
    long long t2 = __builtin_readcyclecounter();
    f4 e0 = {0, 0, 0, 0}, e1 = e0;
    for (int i = 0; i < 128; ++i) {
        e0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, e0, 0, 0, 0);
        e1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b, e1, 0, 0, 0);
    }
    long long t3 = __builtin_readcyclecounter();
    if (l == 0) { cycles[0] = t1 - t0; cycles[1] = t2 - t1; cycles[2] = t3 - t2; }
    out[64 * 64 * 4 + l] = c0[0] + c1[1] + c2[2] + c3[3] + e0[0] + e1[1];
}
int main() {
    float *out; long long *cyc;
    hipMalloc(&out, (64 * 64 * 4 + 64) * 4); hipMalloc(&cyc, 64);
    hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, out, cyc);
    std::vector<float> ho(64 * 64 * 4 + 64); long long hc[3];
    hipMemcpy(ho.data(), out, ho.size() * 4, hipMemcpyDeviceToHost); hipMemcpy(hc, cyc, 24, hipMemcpyDeviceToHost);
    for (int r = 0; r < 64; r += 1) {
        if (r > 9 && r % 16 > 1) continue;
        printf("A one-hot at lane %2d:", r);
        for (int l = 0; l < 64; ++l) for (int v = 0; v < 4; ++v) { float x = ho[(r * 64 + l) * 4 + v]; if (x != 0.f) printf(" D[lane %d][reg %d]=B[lane %g]", l, v, x - 1); }
        printf("\n");
    }
    printf("256 dependent 4x4x1: %lld cycles (%.1f each); 256 over 4 accumulators: %lld (%.1f each); 256 16x16x4 over 2 accumulators: %lld (%.1f each)\n",
           hc[0], hc[0] / 256.0, hc[1], hc[1] / 256.0, hc[2], hc[2] / 256.0);
    return 0;
}
