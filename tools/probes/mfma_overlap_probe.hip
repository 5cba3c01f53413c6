// Does a lone wave's VALU work overlap with its own float64 16x16x4 MFMA in flight?  (round 6: the Hessian sweep of the dual
// step costs 125-165 cycles per MFMA on a lone wave, the instruction itself 65.)
//   hipcc --offload-arch=gfx950 -O3 tools/probes/mfma_overlap_probe.hip -o /tmp/mop && /tmp/mop
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double d4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ long long now() {
    __builtin_amdgcn_sched_barrier(0);
    long long t = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
template <int MODE, int NV>   // MODE 0: MFMA only (two chains), 1: NV independent f64 FMAs only, 2: both interleaved, 3: cvt+cvt+mul feeding each MFMA
__device__ double body(double a, double b, float fa, float fb, int reps) {
    d4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    double v[8] = {a, b, a + 1, b + 1, a + 2, b + 2, a + 3, b + 3};
    for (int r = 0; r < reps; ++r) {
        if (MODE == 0 || MODE == 2) {
            if (r & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc0, 0, 0, 0);
        }
        if (MODE == 1 || MODE == 2) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = __builtin_fma(v[i & 7], 1.0000001, 0.5);
        }
        if (MODE == 3) {
            pin(a);
            const double av = (double)fa, bv = (double)fb * v[r & 7];
            fa += 1.f; fb += 1.f;
            if (r & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av, bv, acc0, 0, 0, 0);
        }
        if (MODE == 4 && (r & 3) == 0) {          // a stage of four: all operand chains first, then the four MFMAs back to back
            pin(a);
            double av[4], bv[4];
#pragma unroll
            for (int i = 0; i < 4; ++i) { av[i] = (double)(fa + (float)i); bv[i] = (double)(fb + (float)i) * v[(r + i) & 7]; }
            fa += 1.f; fb += 1.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                if (i & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc0, 0, 0, 0);
            }
        }
        if (MODE == 5 && (r & 7) == 0) {          // the same in stages of eight
            pin(a);
            double av[8], bv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) { av[i] = (double)(fa + (float)i); bv[i] = (double)(fb + (float)i) * v[(r + i) & 7]; }
            fa += 1.f; fb += 1.f;
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (i & 1) acc1 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc1, 0, 0, 0);
                else acc0 = __builtin_amdgcn_mfma_f64_16x16x4f64(av[i], bv[i], acc0, 0, 0, 0);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    double s = acc0[0] + acc1[1];
    for (int i = 0; i < 8; ++i) s += v[i];
    return s;
}
__global__ void probe(double *out, double *sink, double a, double b) {
    double tot = 0;
    const int reps = 200;
#define RUN(M, NV, slot) { long long t0 = now(); double s = body<M, NV>(a, b, (float)a, (float)b, reps); pin(s); long long t1 = now(); tot += s; if (threadIdx.x == 0) out[slot] = (double)(t1 - t0) / reps; }
    RUN(0, 0, 0) RUN(1, 4, 1) RUN(1, 8, 2) RUN(2, 4, 3) RUN(2, 8, 4) RUN(1, 12, 5) RUN(2, 12, 6) RUN(3, 0, 7) RUN(4, 0, 8) RUN(5, 0, 9)
    sink[threadIdx.x] = tot;
}
int main() {
    double *out, *sink, h[10];
    (void)hipMalloc(&out, 80); (void)hipMalloc(&sink, 512);
    probe<<<1, 64>>>(out, sink, 1.25, 0.75);
    (void)hipMemcpy(h, out, 80, hipMemcpyDeviceToHost);
    const char *n[10] = {"MFMA f64 16x16x4 only (two chains)", "4 f64 FMAs only", "8 f64 FMAs only", "MFMA + 4 FMAs", "MFMA + 8 FMAs", "12 f64 FMAs only", "MFMA + 12 FMAs", "cvt, cvt, mul -> MFMA (the sweep's operand chain)", "4 x (cvt, cvt, mul), then 4 MFMAs: per MFMA", "8 x (cvt, cvt, mul), then 8 MFMAs: per MFMA"};
    for (int i = 0; i < 10; ++i) printf("%-52s %6.1f cycles per iteration\n", n[i], h[i]);
    return 0;
}
