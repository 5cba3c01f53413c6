// Where do ~340 cycles per 16-column stage of the Hessian sweep go?  Single wave, same LDS layout and
// operand pattern as contract_mfma_8x8 (be_dual.hip); variants drop one ingredient at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
__device__ __forceinline__ void pin(double &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ void pin(float &v) { asm volatile("" : "+v"(v)); }
__device__ __forceinline__ long long now() {
    __builtin_amdgcn_sched_barrier(0);
    long long t = __builtin_readcyclecounter();
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_sched_barrier(0);
    return t;
}
template <int MODE>   // 0 full, 1 no mfma (fma instead), 2 no LDS (registers), 3 mfma only, 4 LDS only
__device__ double sweep(const float *As, const double *ws, int ldA, int k, int zrow, int ncol) {
    const int lane = threadIdx.x & 63, kq = lane >> 4, blk = (lane >> 2) & 3, r = lane & 3;
    const int ra = 4 * (blk >> 1) + r, cb = 4 * (blk & 1) + r;
    const float *pa = As + (ra < k ? ra : zrow) * ldA + kq;
    const float *pb = As + (cb < k ? cb : zrow) * ldA + kq;
    const double *pw = ws + kq;
    double acc0 = 0, acc1 = 0;
    float ra_[4] = {1.f, 2.f, 3.f, 4.f}, rb_[4] = {1.f, 2.f, 3.f, 4.f};
    double rw_[4] = {1., 2., 3., 4.};
    for (int c0 = 0; c0 < ncol; c0 += 16) {
        float xa[4], xb[4];
        double xw[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE == 2 || MODE == 3) { xa[s] = ra_[s]; xb[s] = rb_[s]; xw[s] = rw_[s]; pin(xa[s]); pin(xb[s]); pin(xw[s]); }
            else { xa[s] = pa[c0 + 4 * s]; xb[s] = pb[c0 + 4 * s]; xw[s] = pw[c0 + 4 * s]; }
        }
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (MODE == 4) { acc0 += xw[s]; acc1 += (double)(xa[s] + xb[s]); continue; }
            const double av = MODE == 3 ? xw[s] : (double)xa[s];
            const double bv = MODE == 3 ? xw[s] : (double)xb[s] * xw[s];
            if (MODE == 1) { if (s & 1) acc1 = __builtin_fma(av, bv, acc1); else acc0 = __builtin_fma(av, bv, acc0); }
            else if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc0, 0, 0, 0);
        }
    }
    return acc0 + acc1;
}
// Software-pipelined variants: loads of stage i+1 are issued before the arithmetic of stage i.
// PM 0: as in the kernel (f32 cuts: 2 cvt + 1 mul per MFMA); 1: cuts already f64 in LDS (no cvt);
// 2: B operand pre-multiplied f64 in LDS (1 cvt, no mul); 3: both operands ready-made f64 (loads + MFMA only)
template <int PM>
__device__ double sweep_pipe(const float *As, const double *A64, const double *ws, int ldA, int k, int zrow, int ncol) {
    const int lane = threadIdx.x & 63, kq = lane >> 4, blk = (lane >> 2) & 3, r = lane & 3;
    const int ra = 4 * (blk >> 1) + r, cb = 4 * (blk & 1) + r;
    const int rowa = (ra < k ? ra : zrow), rowb = (cb < k ? cb : zrow);
    const float *pa = As + rowa * ldA + kq, *pb = As + rowb * ldA + kq;
    const double *qa = A64 + rowa * ldA + kq, *qb = A64 + rowb * ldA + kq;
    const double *pw = ws + kq;
    double acc0 = 0, acc1 = 0;
    float xa[4], xb[4], ya[4], yb[4];
    double xw[4], yw[4], xda[4], xdb[4], yda[4], ydb[4];
    auto gather = [&](int c0, float (&ga)[4], float (&gb)[4], double (&gw)[4], double (&gda)[4], double (&gdb)[4]) {
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (PM == 0) { ga[s] = pa[c0 + 4 * s]; gb[s] = pb[c0 + 4 * s]; gw[s] = pw[c0 + 4 * s]; }
            if (PM == 1) { gda[s] = qa[c0 + 4 * s]; gdb[s] = qb[c0 + 4 * s]; gw[s] = pw[c0 + 4 * s]; }
            if (PM == 2) { ga[s] = pa[c0 + 4 * s]; gdb[s] = qb[c0 + 4 * s]; }
            if (PM == 3) { gda[s] = qa[c0 + 4 * s]; gdb[s] = qb[c0 + 4 * s]; }
        }
    };
    auto stage = [&](int cn, float (&ca)[4], float (&cb_)[4], double (&cw)[4], double (&cda)[4], double (&cdb)[4],
                     float (&na)[4], float (&nb)[4], double (&nw)[4], double (&nda)[4], double (&ndb)[4]) {
        gather(cn, na, nb, nw, nda, ndb);
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            double av, bv;
            if (PM == 0) { av = (double)ca[s]; bv = (double)cb_[s] * cw[s]; }
            if (PM == 1) { av = cda[s]; bv = cdb[s] * cw[s]; }
            if (PM == 2) { av = (double)ca[s]; bv = cdb[s]; }
            if (PM == 3) { av = cda[s]; bv = cdb[s]; }
            if (s & 1) acc1 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc1, 0, 0, 0);
            else acc0 = __builtin_amdgcn_mfma_f64_4x4x4f64(av, bv, acc0, 0, 0, 0);
        }
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            if (PM == 0) { pin(na[s]); pin(nb[s]); pin(nw[s]); }
            if (PM == 1) { pin(nda[s]); pin(ndb[s]); pin(nw[s]); }
            if (PM == 2) { pin(na[s]); pin(ndb[s]); }
            if (PM == 3) { pin(nda[s]); pin(ndb[s]); }
        }
    };
    gather(0, xa, xb, xw, xda, xdb);
    __builtin_amdgcn_s_waitcnt(0xc07f);
    const int clast = ncol - 16;
    for (int c0 = 0; c0 < ncol; c0 += 32) {
        stage(c0 + 16 < ncol ? c0 + 16 : clast, xa, xb, xw, xda, xdb, ya, yb, yw, yda, ydb);
        if (c0 + 16 < ncol) stage(c0 + 32 < ncol ? c0 + 32 : clast, ya, yb, yw, yda, ydb, xa, xb, xw, xda, xdb);
    }
    return acc0 + acc1;
}
__global__ void probe(double *out, double *sink, int ldA, int k, int zrow, int ncol) {
    extern __shared__ unsigned char smem[];
    float *As = reinterpret_cast<float *>(smem);
    double *ws = reinterpret_cast<double *>(smem + 12 * ldA * 4);
    double *A64 = reinterpret_cast<double *>(smem + 12 * ldA * 4 + 256 * 8);
    for (int i = threadIdx.x; i < 12 * ldA; i += 64) { As[i] = 1.0f + (i % 13) * 0.01f; A64[i] = As[i]; }
    for (int i = threadIdx.x; i < 256; i += 64) ws[i] = 0.5 + i * 1e-3;
    __syncthreads();
    double total = 0;
#define RUN(M, slot)                                                   \
    {                                                                  \
        double v = 0;                                                  \
        for (int rep = 0; rep < 3; ++rep) {                            \
            pin(total);                                                \
            long long t0 = now();                                      \
            v = sweep<M>(As, ws, ldA, k, zrow, ncol);                  \
            pin(v);                                                    \
            long long t1 = now();                                      \
            if (threadIdx.x == 0) out[slot] = (double)(t1 - t0);        \
        }                                                              \
        total += v;                                                    \
    }
    RUN(0, 0) RUN(1, 1) RUN(2, 2) RUN(3, 3) RUN(4, 4)
#define RUNP(M, slot)                                                  \
    {                                                                  \
        double v = 0;                                                  \
        for (int rep = 0; rep < 3; ++rep) {                            \
            pin(total);                                                \
            long long t0 = now();                                      \
            v = sweep_pipe<M>(As, A64, ws, ldA, k, zrow, ncol);        \
            pin(v);                                                    \
            long long t1 = now();                                      \
            if (threadIdx.x == 0) out[slot] = (double)(t1 - t0);        \
        }                                                              \
        total += v;                                                    \
    }
    RUNP(0, 5) RUNP(1, 6) RUNP(2, 7) RUNP(3, 8)
    sink[threadIdx.x] = total;
}
int main() {
    double *out, *sink, h[16];
    hipMalloc(&out, 128); hipMalloc(&sink, 512);
    const char *names[] = {"full stage loop", "fma instead of mfma", "operands from registers (cvt+mul+mfma)", "mfma only", "LDS reads only", "pipelined, f32 cuts (kernel)", "pipelined, f64 cuts (no cvt)", "pipelined, B pre-multiplied (1 cvt)", "pipelined, both operands ready f64"};
    for (int k : {5, 7}) {
        probe<<<1, 64, 12 * 162 * 4 + 256 * 8 + 12 * 162 * 8>>>(out, sink, 162, k, 10, 160);
        hipMemcpy(h, out, 72, hipMemcpyDeviceToHost);
        for (int i = 0; i < 9; ++i) printf("k=%d  %-44s %7.0f cycles per sweep (10 stages) = %5.1f per stage\n", k, names[i], h[i], h[i] / 10);
    }
    return 0;
}
