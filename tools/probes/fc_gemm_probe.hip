// gemm_tiles (be_picnn_fc.hip) in isolation: one workgroup, 8 waves, z0 -> 159 shape (KB = 38, NT = 10)
// and y -> 600 shape (KB = 10, NT = 38).  Variants drop one ingredient at a time.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
constexpr int PF = 4;
template <int MODE>   // 0 full, 1 no mfma, 2 no LDS read (A from registers), 3 no global loads (B from registers)
__device__ __forceinline__ void gemm_tiles(const float *A, int ld, const float *Wp, int KB, int NT, int nt0, int nt1,
                                           f4 &acc0, f4 &acc1) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    const float *ap = A + r16 * ld + 4 * q;
    const f4 *bp0 = reinterpret_cast<const f4 *>(Wp) + (size_t)nt0 * 64 + lane;
    const bool two = nt1 >= 0;
    const f4 *bp1 = reinterpret_cast<const f4 *>(Wp) + (size_t)(two ? nt1 : nt0) * 64 + lane;
    const size_t kstride = (size_t)NT * 64;
    f4 b0[PF], b1[PF];
    f4 areg = {1.f, 2.f, 3.f, 4.f};
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kb = d < KB ? d : KB - 1;
        b0[d] = bp0[(size_t)kb * kstride];
        b1[d] = bp1[(size_t)kb * kstride];
    }
    for (int kb0 = 0; kb0 < KB; kb0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kb = kb0 + d;
            if (kb < KB) {
                f4 a;
                if (MODE == 2) { a = areg; asm volatile("" : "+v"(a)); }
                else a = *reinterpret_cast<const f4 *>(ap + kb * 16);
                const f4 x0 = b0[d], x1 = b1[d];
                if (MODE != 3) {
                    const int nk = kb + PF < KB ? kb + PF : KB - 1;
                    b0[d] = bp0[(size_t)nk * kstride];
                    b1[d] = bp1[(size_t)nk * kstride];
                }
                if (MODE == 1) {
                    acc0 += a * x0;
                    if (two) acc1 += a * x1;
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.y, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x0.z, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x0.w, acc0, 0, 0, 0);
                    if (two) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x1.w, acc1, 0, 0, 0);
                }
            }
        }
    }
}
// MODE 4/5: wave-uniform tile indices (scalar branch on `two`, no exec-masked MFMAs) and the A fragment of
// the next k-block read before the MFMAs of the current one.  MODE 5: additionally B from registers.
template <int MODE>
__device__ __forceinline__ void gemm_tiles2(const float *A, int ld, const float *Wp, int KB, int NT, int nt0, int nt1,
                                            f4 &acc0, f4 &acc1) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    nt0 = __builtin_amdgcn_readfirstlane(nt0);
    nt1 = __builtin_amdgcn_readfirstlane(nt1);
    const float *ap = A + r16 * ld + 4 * q;
    const f4 *bp0 = reinterpret_cast<const f4 *>(Wp) + (size_t)nt0 * 64 + lane;
    const bool two = nt1 >= 0;
    const f4 *bp1 = reinterpret_cast<const f4 *>(Wp) + (size_t)(two ? nt1 : nt0) * 64 + lane;
    const size_t kstride = (size_t)NT * 64;
    f4 b0[PF], b1[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        const int kb = d < KB ? d : KB - 1;
        b0[d] = bp0[(size_t)kb * kstride];
        if (two) b1[d] = bp1[(size_t)kb * kstride];
    }
    f4 an = *reinterpret_cast<const f4 *>(ap);
    for (int kb0 = 0; kb0 < KB; kb0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kb = kb0 + d;
            if (kb < KB) {
                const f4 a = an;
                an = *reinterpret_cast<const f4 *>(ap + (kb + 1 < KB ? kb + 1 : kb) * 16);
                const f4 x0 = b0[d], x1 = b1[d];
                if (MODE != 5) {
                    const int nk = kb + PF < KB ? kb + PF : KB - 1;
                    b0[d] = bp0[(size_t)nk * kstride];
                    if (two) b1[d] = bp1[(size_t)nk * kstride];
                }
                if (two) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.x, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.y, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x1.y, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x0.z, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x1.z, acc1, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x0.w, acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x1.w, acc1, 0, 0, 0);
                } else {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.y, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x0.z, acc0, 0, 0, 0);
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x0.w, acc0, 0, 0, 0);
                }
            }
        }
    }
}
// MODE 6/7: k-blocks padded to a multiple of PF (zero fragments), so the ring body has no conditionals;
// one- and two-tile cases are separate loops; accumulators never change registers.  MODE 7: B from registers.
template <int MODE, bool TWO>
__device__ __forceinline__ void gemm_loop3(const float *ap, const f4 *bp0, const f4 *bp1, size_t kstride, int KBp,
                                           f4 &acc0, f4 &acc1) {
    f4 b0[PF], b1[PF];
#pragma unroll
    for (int d = 0; d < PF; ++d) {
        b0[d] = bp0[(size_t)d * kstride];
        if (TWO) b1[d] = bp1[(size_t)d * kstride];
    }
    f4 an = *reinterpret_cast<const f4 *>(ap);
    for (int kb0 = 0; kb0 < KBp; kb0 += PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const int kb = kb0 + d;
            const f4 a = an;
            an = *reinterpret_cast<const f4 *>(ap + (kb + 1 < KBp ? kb + 1 : kb) * 16);
            const f4 x0 = b0[d], x1 = b1[d];
            if (MODE != 7) {
                const int nk = kb + PF < KBp ? kb + PF : kb;
                b0[d] = bp0[(size_t)nk * kstride];
                if (TWO) b1[d] = bp1[(size_t)nk * kstride];
            }
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x0.x, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x1.x, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x0.y, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x1.y, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x0.z, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x1.z, acc1, 0, 0, 0);
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x0.w, acc0, 0, 0, 0);
            if (TWO) acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x1.w, acc1, 0, 0, 0);
            if (MODE >= 8) {
                // interleave: one MFMA, then up to two non-MFMA instructions (they issue while the MFMA runs)
#pragma unroll
                for (int g = 0; g < (TWO ? 8 : 4); ++g) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x126, 2, 0);   // VALU | SALU | VMEM read | DS read
                }
            }
        }
    }
}
template <int MODE>
__device__ __forceinline__ void gemm_tiles3(const float *A, int ld, const float *Wp, int KB, int NT, int nt0, int nt1,
                                            f4 &acc0, f4 &acc1) {
    const int lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    nt0 = __builtin_amdgcn_readfirstlane(nt0);
    nt1 = __builtin_amdgcn_readfirstlane(nt1);
    const float *ap = A + r16 * ld + 4 * q;
    const f4 *bp0 = reinterpret_cast<const f4 *>(Wp) + (size_t)nt0 * 64 + lane;
    const f4 *bp1 = reinterpret_cast<const f4 *>(Wp) + (size_t)(nt1 >= 0 ? nt1 : nt0) * 64 + lane;
    const size_t kstride = (size_t)NT * 64;
    const int KBp = (KB + PF - 1) / PF * PF;
    if (nt1 >= 0) gemm_loop3<MODE, true>(ap, bp0, bp1, kstride, KBp, acc0, acc1);
    else gemm_loop3<MODE, false>(ap, bp0, bp1, kstride, KBp, acc0, acc1);
}
template <int MODE>
__global__ __launch_bounds__(512) void probe(const float *Wp, int KB, int NT, int ld, float *sink, long long *cyc) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 16 * ld; i += 512) lds[i] = 0.001f * (i % 97);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    f4 tot = {0, 0, 0, 0};
    for (int nt = wave; nt < NT; nt += 16) {
        const int nt1 = nt + 8 < NT ? nt + 8 : -1;
        f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
        if (MODE >= 6) gemm_tiles3<MODE>(lds, ld, Wp, KB, NT, nt, nt1, acc0, acc1);
        else if (MODE >= 4) gemm_tiles2<MODE>(lds, ld, Wp, KB, NT, nt, nt1, acc0, acc1);
        else gemm_tiles<MODE>(lds, ld, Wp, KB, NT, nt, nt1, acc0, acc1);
        tot += acc0 + acc1;
    }
    const long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    const long long t2 = __builtin_readcyclecounter();
    if ((threadIdx.x & 63) == 0) { cyc[blockIdx.x * 16 + wave] = t1 - t0; cyc[blockIdx.x * 16 + 8 + wave] = t2 - t0; }
    sink[blockIdx.x * 512 + threadIdx.x] = tot.x + tot.y + tot.z + tot.w;
}
// 16 waves, one tile each (NT = 16), KB k-blocks: how close to the pipe-bound time 4 waves * KB * 4 * 33?
template <int MODE>
__global__ __launch_bounds__(1024) void probe16(const float *Wp, int KB, int NT, int ld, float *sink, long long *cyc, int active) {
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63, r16 = lane & 15, q = lane >> 4;
    for (int i = threadIdx.x; i < 16 * ld; i += 1024) lds[i] = 0.001f * (i % 97);
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    f4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    if (wave < active) {
        const float *ap = lds + r16 * ld + 4 * q;
        const f4 *bp0 = reinterpret_cast<const f4 *>(Wp) + (size_t)wave * 64 + lane;
        gemm_loop3<MODE, false>(ap, bp0, bp0, (size_t)NT * 64, KB, acc0, acc1);
    }
    const long long t1 = __builtin_readcyclecounter();
    if (lane == 0) cyc[wave] = t1 - t0;
    sink[blockIdx.x * 1024 + threadIdx.x] = acc0.x + acc0.y + acc0.z + acc0.w;
}
int main() {
    float *w, *sink; long long *cyc, h[16];
    hipMalloc(&w, 44 * 40 * 1024); hipMemset(w, 0, 44 * 40 * 1024);
    hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 256 * 16 * 8);
    const char *names[] = {"full", "no mfma", "A from registers", "B from registers", "v2 full", "v2 B from registers", "v3 full", "v3 B from registers", "v3 + interleave"};
    struct { int KB, NT, ld; const char *what; } shapes[] = {{38, 10, 648, "z0(600)->159"}, {10, 38, 200, "y(159)->600"}};
    for (auto sh : shapes)
        for (int mode = 4; mode < 9; ++mode) {
            for (int rep = 0; rep < 3; ++rep) {
                const size_t lds = 16 * sh.ld * 4;
                if (mode == 0) probe<0><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 1) probe<1><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 2) probe<2><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 3) probe<3><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 4) probe<4><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 5) probe<5><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 6) probe<6><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 7) probe<7><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
                if (mode == 8) probe<8><<<1, 512, lds>>>(w, sh.KB, sh.NT, sh.ld, sink, cyc);
            }
            hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
            long long mx = 0, mn = 1LL << 60; for (int i = 0; i < 8; ++i) { mx = h[i] > mx ? h[i] : mx; mn = h[i] < mn ? h[i] : mn; }
            printf("%-14s %-18s per-wave loop cycles min %6lld max %6lld   phase (to barrier) %6lld\n", sh.what, names[mode], mn, mx, h[8]);
        }
    for (int active : {4, 8, 12, 16})
        for (int mode : {6, 7, 8}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (mode == 6) probe16<6><<<1, 1024, 16 * 648 * 4>>>(w, 50, 16, 648, sink, cyc, active);
                if (mode == 7) probe16<7><<<1, 1024, 16 * 648 * 4>>>(w, 50, 16, 648, sink, cyc, active);
                if (mode == 8) probe16<8><<<1, 1024, 16 * 648 * 4>>>(w, 50, 16, 648, sink, cyc, active);
            }
            hipMemcpy(h, cyc, 128, hipMemcpyDeviceToHost);
            long long mx = 0; for (int i = 0; i < active; ++i) mx = h[i] > mx ? h[i] : mx;
            printf("16-wave WG, %2d waves active, single tile, KB=50, %-22s: %6lld cycles (pipe bound %d)\n", active,
                   mode == 6 ? "v3" : mode == 7 ? "v3 B from registers" : "v3 + interleave", mx, (active + 3) / 4 * 50 * 4 * 33);
        }
    return 0;
}
