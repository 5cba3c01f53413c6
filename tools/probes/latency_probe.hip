// Single-wave instruction latency/throughput probe for gfx950 (cycles via s_memtime).
#include <hip/hip_runtime.h>
#include <cstdio>
#define REP 64
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(int &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ long long now() {
    __builtin_amdgcn_sched_barrier(0);
    long long t = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
// the pins before/after keep the measured chain between the two counter reads
#define TIME(name, setup, body)                                                     \
    {                                                                               \
        setup;                                                                      \
        pin(a); pin(e); pin(f); pin(g); pin(h); pin(c); pin(d); pin(fa); pin(aux);  \
        long long t0 = now();                                                       \
        _Pragma("unroll") for (int r = 0; r < REP; ++r) { body; }                    \
        pin(a); pin(e); pin(f); pin(g); pin(h); pin(c); pin(d); pin(fa); pin(aux);  \
        long long t1 = now();                                                       \
        if (threadIdx.x == 0) out[idx] = (double)(t1 - t0) / REP;                    \
        ++idx;                                                                      \
    }
typedef double d4 __attribute__((ext_vector_type(4)));
__global__ void probe(double *out, double *sink, const double *in) {
    __shared__ double lds[1024];
    int idx = 0;
    double aux = 0;
    double a = in[threadIdx.x], b = in[64 + threadIdx.x], c = in[128 + threadIdx.x], d = in[192 + threadIdx.x];
    double e = a + 1, f = b + 1, g = c + 1, h = d + 1;
    float fa = (float)a, fb = (float)b;
    lds[threadIdx.x] = a; lds[threadIdx.x + 64] = b;
    __syncthreads();
    // 0: dependent f64 fma chain
    TIME("dep fma64", , a = __builtin_fma(a, b, c));
    // 1: 4 independent f64 fma chains (per-instruction cost)
    TIME("4x indep fma64", , a = __builtin_fma(a, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, d); g = __builtin_fma(g, b, d));
    // 2: dependent f32 fma chain
    TIME("dep fma32", , fa = __builtin_fmaf(fa, fb, fb));
    // 3: dependent f64 add
    TIME("dep add64", , a = a + b);
    // 4: dpp row bcast + dependent fma
    TIME("dpp64+fma", , a = __builtin_fma(b, __builtin_amdgcn_update_dpp(0.0, a, 0x153, 0xf, 0xf, true), a));
    // 5: readlane pair + fma
    TIME("readlane+fma", , { int lo = __builtin_amdgcn_readlane(__double2loint(a), 3); int hi = __builtin_amdgcn_readlane(__double2hiint(a), 3); a = __builtin_fma(b, __hiloint2double(hi, lo), a); });
    // 6: dependent LDS read chain (address depends on loaded value)
    { int p = threadIdx.x & 63; lds[512 + p] = 0.0; __syncthreads();
      TIME("dep lds read", , { double v = lds[512 + p]; p = (p + (int)v) & 63; aux = p; });
      a += p; }
    // 7: rcp64
    TIME("dep rcp64", , a = __builtin_amdgcn_rcp(a));
    // 8: exp (ocml) dependent
    TIME("dep exp64", , a = exp(a * 1e-3));
    // 9: f64 division dependent
    TIME("dep div64", , a = 1.0 / (1.0 + a));
    // 10: mfma 4x4x4 f64 dependent chain
    { double acc = 0;
      TIME("dep mfma4x4x4", , acc = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc, 0, 0, 0); if (r == REP - 1) aux = acc);
      a += acc; }
    // 11: mfma 4x4x4 two chains
    { double acc0 = 0, acc1 = 0;
      TIME("2x mfma4x4x4", , acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(a, b, acc0, 0, 0, 0); acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(c, d, acc1, 0, 0, 0); if (r == REP - 1) aux = acc0 + acc1);
      a += acc0 + acc1; }
    // 12: mfma 16x16x4 f64 dependent
    { d4 acc = {0, 0, 0, 0};
      TIME("dep mfma16x16x4", , acc = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc, 0, 0, 0); if (r == REP - 1) aux = acc[0] + acc[3]);
      a += acc[0] + acc[1] + acc[2] + acc[3]; }
    // 13: cvt f32->f64 + mul dependent
    TIME("cvt+mul", , a = (double)(float)a * b);
    // 14: salu chain
    { int s = __builtin_amdgcn_readfirstlane((int)a);
      TIME("dep salu add", , s = s * 3 + 1; if (r == REP - 1) aux = s);
      a += s; }
    // 15: independent 8 f64 fma
    TIME("8x indep fma64", , a = __builtin_fma(a, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, d); g = __builtin_fma(g, b, d);
         h = __builtin_fma(h, b, c); c = __builtin_fma(c, b, b); d = __builtin_fma(d, b, b); a = __builtin_fma(a, e, f));
    // 16..19: f32 MFMA 16x16x4, 1 / 2 / 4 independent accumulators; 32x32x2
    typedef float f4v __attribute__((ext_vector_type(4)));
    typedef float f16v __attribute__((ext_vector_type(16)));
    { f4v m0 = {0, 0, 0, 0};
      TIME("dep mfma f32 16x16x4", , m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m0, 0, 0, 0); if (r == REP - 1) aux = m0[0]);
      a += m0[1]; }
    { f4v m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0};
      TIME("2x mfma f32 16x16x4", , m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, m1, 0, 0, 0); if (r == REP - 1) aux = m0[0] + m1[0]);
      a += m0[1] + m1[1]; }
    { f4v m0 = {0, 0, 0, 0}, m1 = {0, 0, 0, 0}, m2 = {0, 0, 0, 0}, m3 = {0, 0, 0, 0};
      TIME("4x mfma f32 16x16x4", , m0 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fb, m0, 0, 0, 0); m1 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fa, m1, 0, 0, 0);
           m2 = __builtin_amdgcn_mfma_f32_16x16x4f32(fa, fa, m2, 0, 0, 0); m3 = __builtin_amdgcn_mfma_f32_16x16x4f32(fb, fb, m3, 0, 0, 0); if (r == REP - 1) aux = m0[0] + m1[0] + m2[0] + m3[0]);
      a += m0[1] + m1[1] + m2[1] + m3[1]; }
    { f16v m0 = {0};
      TIME("dep mfma f32 32x32x2", , m0 = __builtin_amdgcn_mfma_f32_32x32x2f32(fa, fb, m0, 0, 0, 0); if (r == REP - 1) aux = m0[0]);
      a += m0[1]; }
    sink[threadIdx.x] = a + e + f + g + h + fa + c + d + aux;
}
int main() {
    double *out, *sink, *in, h_in[256], h_out[32];
    for (int i = 0; i < 256; ++i) h_in[i] = 1.0 + 1e-3 * (i % 7);
    hipMalloc(&out, 32 * 8); hipMalloc(&sink, 64 * 8); hipMalloc(&in, 256 * 8);
    hipMemcpy(in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    for (int rep = 0; rep < 2; ++rep) probe<<<1, 64>>>(out, sink, in);
    hipMemcpy(h_out, out, sizeof(h_out), hipMemcpyDeviceToHost);
    const char *names[] = {"dep fma64", "4x indep fma64 (per group)", "dep fma32", "dep add64", "dpp64 bcast + fma", "2 readlane + fma",
                           "dep lds read", "dep rcp64", "dep exp64(ocml)+mul", "dep div64 (+add)", "dep mfma f64 4x4x4", "2 chains mfma 4x4x4 (per pair)",
                           "dep mfma f64 16x16x4", "cvt f32->f64->f32 + mul", "dep salu mul+add", "8 fma64 mostly independent (per group)", "dep mfma f32 16x16x4", "2 chains f32 16x16x4 (per pair)", "4 chains f32 16x16x4 (per four)", "dep mfma f32 32x32x2"};
    for (int i = 0; i < 20; ++i) printf("%-44s %8.1f cycles\n", names[i], h_out[i]);
    return 0;
}
