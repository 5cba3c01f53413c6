// What bounds the k-loop of the 4x4x1 half-tile GEMM (be_picnn_fc_half_dev.h)?  Cycles per k-block and wave for variants of
// the loop body, 8 waves per workgroup, one workgroup per CU or a single one.
//   hipcc --offload-arch=gfx950 -O3 -o /tmp/half_gemm_probe tools/probes/half_gemm_probe.hip && /tmp/half_gemm_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int KB = 100, NT = 10, LD = 1608;   // K = 1600 from LDS rows of pitch LD (= 8 mod 64)

template <int MODE, int PF = 5, bool XLDS = true>   // 0: swap + half loads; 1: duplicate loads, no swap; 2: no global loads (B reused), no swap; 3: as 0, swaps only (no mfma)
                      // 4: as 1 with two independent accumulator chains (even / odd k-blocks; NOT the product's order)
__global__ __launch_bounds__(512, 4) void probe(const float *W, float *out, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 8 * LD; e += 512) lds[e] = 1.f + (e % 7);
    __syncthreads();
    const int j = lane & 3, h = lane >> 5, c32 = 4 * ((lane >> 2) & 7) + j;
    const int u = wave % 5, col = 32 * u + c32, nt = col >> 4;
    const f4 *bp = reinterpret_cast<const f4 *>(W) + (size_t)nt * 64 + (MODE == 0 || MODE == 3 ? 32 * h : 0) + (col & 15);
    const size_t ks = (size_t)NT * 64;
    const float *ap = lds + (4 * h + j) * LD;
    constexpr int NQ = (MODE == 0 || MODE == 3) ? 2 : 4;
    f4 ring[PF][NQ], x[4], acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
#pragma unroll
    for (int d = 0; d < PF; ++d)
#pragma unroll
        for (int t = 0; t < NQ; ++t) ring[d][t] = bp[d * ks + 16 * t];
#pragma unroll
    for (int q = 0; q < 4; ++q) x[q] = *reinterpret_cast<const f4 *>(ap + 4 * q);
    const long long t0 = __builtin_readcyclecounter();
    for (int kb0 = 0; kb0 < KB; kb0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kb = kb0 + d;
            f4 b[4];
            const int nk = kb + PF < KB ? kb + PF : kb;
            if (MODE == 0 || MODE == 3) {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f4 own = ring[d][t];
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const auto sw = __builtin_amdgcn_permlane32_swap(__float_as_uint(own[c]), __float_as_uint(own[c]), false, false);
                        b[t][c] = __uint_as_float(sw[0]);
                        b[2 + t][c] = __uint_as_float(sw[1]);
                    }
                    ring[d][t] = bp[(size_t)nk * ks + 16 * t];
                }
            } else {
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    b[t] = ring[d][t];
                    if (MODE != 2) ring[d][t] = bp[(size_t)nk * ks + 16 * t];
                }
            }
            if (MODE == 3) {
#pragma unroll
                for (int q = 0; q < 4; ++q) acc += b[q] * x[q];
            } else {
                f4 &a = (MODE == 4 && (d & 1)) ? acc2 : acc;
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q) a = __builtin_amdgcn_mfma_f32_4x4x1f32(x[q][s], b[q][s], a, 0, 0, 0);
                const float *an = ap + 16 * (kb + 1 < KB ? kb + 1 : kb);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    a = __builtin_amdgcn_mfma_f32_4x4x1f32(x[q][3], b[q][3], a, 0, 0, 0);
                    if (XLDS) x[q] = *reinterpret_cast<const f4 *>(an + 4 * q);
                }
            }
        }
    }
    const long long t1 = __builtin_readcyclecounter();
    acc += acc2;
    out[(size_t)blockIdx.x * 512 + tid] = acc[0] + acc[1] + acc[2] + acc[3];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = t1 - t0;
}

// 64 columns x 8 samples per wave: lane = column, two accumulators (samples 0-3 / 4-7), every B fragment loaded once.
// NU: distinct units among the 8 waves (wave % NU): 3 = the narrow layers (three busy waves would be the real case; here all
// eight run, waves with equal units hit in L1).  BUSY: only waves < BUSY work.
template <int PF, int NU, int BUSY>
__global__ __launch_bounds__(512, 4) void probe64(const float *W, float *out, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 8 * LD; e += 512) lds[e] = 1.f + (e % 7);
    __syncthreads();
    long long dt = 0;
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (wave < BUSY) {
        const int i = lane & 3, u = wave % NU, nt = (4 * u + (lane >> 4)) % NT;
        const f4 *bp = reinterpret_cast<const f4 *>(W) + (size_t)nt * 64 + (lane & 15);
        const size_t ks = (size_t)NT * 64;
        const float *a0 = lds + i * LD, *a1 = lds + (4 + i) * LD;
        f4 ring[PF][4], x0[4], x1[4];
#pragma unroll
        for (int d = 0; d < PF; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) ring[d][q] = bp[d * ks + 16 * q];
#pragma unroll
        for (int q = 0; q < 4; ++q) { x0[q] = *reinterpret_cast<const f4 *>(a0 + 4 * q); x1[q] = *reinterpret_cast<const f4 *>(a1 + 4 * q); }
        const long long t0 = __builtin_readcyclecounter();
        for (int kb0 = 0; kb0 < KB; kb0 += PF) {
#pragma unroll
            for (int d = 0; d < PF; ++d) {
                const int kb = kb0 + d;
                f4 b[4];
                const int nk = kb + PF < KB ? kb + PF : kb;
#pragma unroll
                for (int q = 0; q < 4; ++q) { b[q] = ring[d][q]; ring[d][q] = bp[(size_t)nk * ks + 16 * q]; }
#pragma unroll
                for (int s = 0; s < 3; ++s)
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0[q][s], b[q][s], acc0, 0, 0, 0);
                        acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[q][s], b[q][s], acc1, 0, 0, 0);
                    }
                const int nb = 16 * (kb + 1 < KB ? kb + 1 : kb);
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0[q][3], b[q][3], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[q][3], b[q][3], acc1, 0, 0, 0);
                    x0[q] = *reinterpret_cast<const f4 *>(a0 + nb + 4 * q);
                    x1[q] = *reinterpret_cast<const f4 *>(a1 + nb + 4 * q);
                }
            }
        }
        dt = __builtin_readcyclecounter() - t0;
    }
    acc0 += acc1;
    out[(size_t)blockIdx.x * 512 + tid] = acc0[0] + acc0[1] + acc0[2] + acc0[3];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = dt;
}
template <int PF, int NU, int BUSY>
void run64(const char *name, const float *W, float *out, long long *cyc, int wgs) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe64<PF, NU, BUSY>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * LD * 4);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe64<PF, NU, BUSY>), dim3(wgs), dim3(512), 8 * LD * 4, 0, W, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> c(wgs * 8);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; long long mx = 0; int cnt = 0;
    for (auto v : c) if (v) { mean += v; ++cnt; if (v > mx) mx = v; }
    mean /= cnt;
    printf("%-44s %4d workgroups: %.0f cycles / k-block (mean), %.0f (slowest wave)\n", name, wgs, mean / KB, (double)mx / KB);
}

// As probe64 with ring 2, but the refill loads are inline asm issued at the top of a k-block, with hand-placed waits:
// the compiler cannot sink them behind the MFMAs.
__device__ __forceinline__ f4 gload(const f4 *p) {
    f4 r;
    asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(r) : "v"(p) : "memory");
    return r;
}
template <int N>
__device__ __forceinline__ void wait_vm(f4 &a, f4 &b, f4 &c, f4 &d) {
    asm volatile("s_waitcnt vmcnt(%4)" : "+v"(a), "+v"(b), "+v"(c), "+v"(d) : "n"(N) : "memory");
}
template <int NU, int BUSY>
__global__ __launch_bounds__(512, 4) void probe64asm(const float *W, float *out, long long *cyc) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    for (int e = tid; e < 8 * LD; e += 512) lds[e] = 1.f + (e % 7);
    __syncthreads();
    long long dt = 0;
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (wave < BUSY) {
        const int i = lane & 3, u = wave % NU, nt = (4 * u + (lane >> 4)) % NT;
        const f4 *bp = reinterpret_cast<const f4 *>(W) + (size_t)nt * 64 + (lane & 15);
        const size_t ks = (size_t)NT * 64;
        const float *a0 = lds + i * LD, *a1 = lds + (4 + i) * LD;
        f4 s0[4], s1[4], x0[4], x1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) s0[q] = gload(bp + 16 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) s1[q] = gload(bp + ks + 16 * q);
#pragma unroll
        for (int q = 0; q < 4; ++q) { x0[q] = *reinterpret_cast<const f4 *>(a0 + 4 * q); x1[q] = *reinterpret_cast<const f4 *>(a1 + 4 * q); }
        const long long t0 = __builtin_readcyclecounter();
        auto block = [&](f4 (&b)[4], int kb) {
#pragma unroll
            for (int s = 0; s < 3; ++s)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0[q][s], b[q][s], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[q][s], b[q][s], acc1, 0, 0, 0);
                }
            const int nb = 16 * (kb + 1 < KB ? kb + 1 : kb);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = __builtin_amdgcn_mfma_f32_4x4x1f32(x0[q][3], b[q][3], acc0, 0, 0, 0);
                acc1 = __builtin_amdgcn_mfma_f32_4x4x1f32(x1[q][3], b[q][3], acc1, 0, 0, 0);
                x0[q] = *reinterpret_cast<const f4 *>(a0 + nb + 4 * q);
                x1[q] = *reinterpret_cast<const f4 *>(a1 + nb + 4 * q);
            }
        };
        for (int kb = 0; kb < KB; kb += 2) {
            f4 t0v[4], t1v[4];
            const f4 *n0 = bp + (size_t)(kb + 2 < KB ? kb + 2 : kb) * ks, *n1 = bp + (size_t)(kb + 3 < KB ? kb + 3 : kb) * ks;
#pragma unroll
            for (int q = 0; q < 4; ++q) t0v[q] = gload(n0 + 16 * q);
            wait_vm<8>(s0[0], s0[1], s0[2], s0[3]);
            block(s0, kb);
#pragma unroll
            for (int q = 0; q < 4; ++q) t1v[q] = gload(n1 + 16 * q);
            wait_vm<8>(s1[0], s1[1], s1[2], s1[3]);
            block(s1, kb + 1);
#pragma unroll
            for (int q = 0; q < 4; ++q) { s0[q] = t0v[q]; s1[q] = t1v[q]; }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        dt = __builtin_readcyclecounter() - t0;
    }
    acc0 += acc1;
    out[(size_t)blockIdx.x * 512 + tid] = acc0[0] + acc0[1] + acc0[2] + acc0[3];
    if (lane == 0) cyc[blockIdx.x * 8 + wave] = dt;
}
template <int NU, int BUSY>
void run64asm(const char *name, const float *W, float *out, long long *cyc, int wgs) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe64asm<NU, BUSY>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * LD * 4);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe64asm<NU, BUSY>), dim3(wgs), dim3(512), 8 * LD * 4, 0, W, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> c(wgs * 8);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; long long mx = 0; int cnt = 0;
    for (auto v : c) if (v) { mean += v; ++cnt; if (v > mx) mx = v; }
    mean /= cnt;
    printf("%-44s %4d workgroups: %.0f cycles / k-block (mean), %.0f (slowest wave)\n", name, wgs, mean / KB, (double)mx / KB);
}

template <int MODE, int PF = 5, bool XLDS = true>
void run(const char *name, const float *W, float *out, long long *cyc, int wgs, int active_waves_note) {
    hipFuncSetAttribute(reinterpret_cast<const void *>(probe<MODE, PF, XLDS>), hipFuncAttributeMaxDynamicSharedMemorySize, 8 * LD * 4);
    for (int r = 0; r < 3; ++r) hipLaunchKernelGGL((probe<MODE, PF, XLDS>), dim3(wgs), dim3(512), 8 * LD * 4, 0, W, out, cyc);
    hipDeviceSynchronize();
    std::vector<long long> c(wgs * 8);
    hipMemcpy(c.data(), cyc, c.size() * 8, hipMemcpyDeviceToHost);
    double mean = 0; long long mx = 0;
    for (auto v : c) { mean += v; if (v > mx) mx = v; }
    mean /= c.size();
    printf("%-44s %4d workgroups: %.0f cycles / k-block (mean), %.0f (slowest wave)\n", name, wgs, mean / KB, (double)mx / KB);
}

int main() {
    float *W, *out; long long *cyc;
    const size_t wf = (size_t)(KB + 16) * NT * 256;
    hipMalloc(&W, wf * 4); hipMalloc(&out, 512 * 512 * 4); hipMalloc(&cyc, 512 * 8 * 8);
    std::vector<float> hw(wf);
    for (size_t i = 0; i < wf; ++i) hw[i] = (float)((i * 2654435761u) % 1000) * 1e-3f;
    hipMemcpy(W, hw.data(), wf * 4, hipMemcpyHostToDevice);
    for (int wgs : {1, 256}) {
        run<0>("half loads + permlane32_swap", W, out, cyc, wgs, 8);
        run<1>("duplicate loads, no swap", W, out, cyc, wgs, 8);
        run<2>("no global loads in the loop", W, out, cyc, wgs, 8);
        run<3>("half loads + swap, VALU instead of MFMA", W, out, cyc, wgs, 8);
        run<4>("duplicate loads, two accumulator chains", W, out, cyc, wgs, 8);
        run<0, 5, false>("half loads + swap, A operand not re-read", W, out, cyc, wgs, 8);
        run<0, 10>("half loads + swap, ring of 10 k-blocks", W, out, cyc, wgs, 8);
        run<0, 2>("half loads + swap, ring of 2 k-blocks", W, out, cyc, wgs, 8);
        run<1, 2>("duplicate loads, ring of 2 k-blocks", W, out, cyc, wgs, 8);
        run64<2, 8, 8>("64-col units: 8 waves, 8 units, ring 2", W, out, cyc, wgs);
        run64<3, 8, 8>("64-col units: 8 waves, 8 units, ring 3", W, out, cyc, wgs);
        run64<2, 3, 3>("64-col units: 3 waves busy, ring 2", W, out, cyc, wgs);
        run64<3, 3, 3>("64-col units: 3 waves busy, ring 3", W, out, cyc, wgs);
        run64asm<8, 8>("64-col, asm loads: 8 waves, 8 units", W, out, cyc, wgs);
        run64asm<3, 3>("64-col, asm loads: 3 waves busy", W, out, cyc, wgs);
        run64asm<8, 4>("64-col, asm loads: 4 waves busy", W, out, cyc, wgs);
    }
    return 0;
}
