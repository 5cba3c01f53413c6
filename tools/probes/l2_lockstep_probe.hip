// Does it matter that every CU streams the SAME weight pack in the SAME order at the same time?  
//
// Models phase A of fused_fc_solve_kernel: 256 workgroups x 16 waves, one per CU; a wave owns the output tiles nt = w, w + 16, ..
// of a packed operand [KB][NT] of 1 KiB fragments (be_picnn_fc_dev.h, pack_operand: fragment (kb, nt) at (kb NT + nt) KiB) and
// walks the k-blocks of a tile with a register ring two deep, four v_mfma_f32_16x16x4_f32 per fragment -- gemm_loop<false, 2>.
// Shapes: the Bibsonomy network's wide GEMMs (KB = 10, NT = 38: y -> 600 and d1 Wzu1^T) and narrow ones (KB = 38, NT = 10:
// z0 -> 159 and d0 Wyu0^T), 1.73 MB per pass over the four of them like one evaluation.
// Variants of the order in which a workgroup takes its tiles (the k order inside a tile never changes, so the kernel's bits
// would not either):
//   lockstep   tile nt first on every CU                                   (what the kernel does)
//   rot_cu     tile (nt + 7 b) mod NT first on workgroup b                 (every CU somewhere else in the pack)
//   rot_xcd    tile (nt + (b mod 8) NT / 8) mod NT                         (the CUs of an XCD in step, the XCDs apart)
//   rot_k      k-blocks of a tile start at (b mod KB) and wrap             (NOT bit-preserving: shown for the size of the effect)
//   flat       lockstep order, but the ring runs on across the tile boundaries of a wave inside a phase (no restart per tile;
//              odd fragment counts read one clamped fragment more)
// and, for each, MFMA = 1 (paced like the kernel) or 0 (pure stream, the delivery limit).  grid 1 = one workgroup alone.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f4 __attribute__((ext_vector_type(4)));

struct Shape { int KB, NT; long long off; };      // off: first fragment of the operand (KiB units = 64 f4)
struct Args { Shape s[4]; int ns; };

template <int MODE, bool MFMA>
__global__ __launch_bounds__(1024) void stream(const f4 *w, Args a, float *sink, long long *cycles) {
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int b = blockIdx.x;
    f4 acc = {0.f, 0.f, 0.f, 0.f};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int si = 0; si < a.ns; ++si) {
        const int KB = a.s[si].KB, NT = a.s[si].NT;
        const f4 *base = w + (size_t)a.s[si].off * 64 + lane;
        int rot = 0, krot = 0;
        if (MODE == 1) rot = (7 * b) % NT;
        if (MODE == 2) rot = ((b & 7) * NT) / 8;
        if (MODE == 3) krot = b % KB;
        if (MODE == 4) {
            // one stream per wave and phase: fragment i of the wave = (tile wave + 16 (i / KB), k-block i % KB); the ring never
            // restarts at a tile boundary (the accumulator would be written out there: modelled by an add into `sink`-bound acc2)
            const int ntiles = (NT - wave + 15) / 16, total = ntiles * KB;
            const size_t ks = (size_t)NT * 64;
            auto frag = [&](int i) -> const f4 * {
                const int ii = i < total ? i : total - 1, t = ii / KB, kb = ii - t * KB;
                return base + (size_t)(wave + 16 * t) * 64 + (size_t)kb * ks;
            };
            if (total > 0) {
                f4 r0 = *frag(0);
                __builtin_amdgcn_sched_barrier(0);
                f4 r1 = *frag(1);
                __builtin_amdgcn_sched_barrier(0);
                for (int i = 0; i < total; i += 2) {
                    {
                        const f4 x = r0;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.w, acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        r0 = *frag(i + 2);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                    {
                        const f4 x = r1;
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.w, acc, 0, 0, 0);
                        __builtin_amdgcn_sched_barrier(0);
                        r1 = *frag(i + 3);
                        __builtin_amdgcn_sched_barrier(0);
                    }
                }
            }
        } else
        for (int nt0 = wave; nt0 < NT; nt0 += 16) {
            int nt = nt0 + rot;
            if (nt >= NT) nt -= NT;
            const f4 *p = base + (size_t)nt * 64;
            const size_t ks = (size_t)NT * 64;
            auto kb_of = [&](int i) { int k = i + krot; return k >= KB ? k - KB : k; };
            f4 r0 = p[(size_t)kb_of(0) * ks];
            __builtin_amdgcn_sched_barrier(0);
            f4 r1 = p[(size_t)kb_of(1 < KB ? 1 : 0) * ks];
            __builtin_amdgcn_sched_barrier(0);
            for (int kb = 0; kb < KB; kb += 2) {
                {
                    const f4 x = r0;
                    if (MFMA) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.w, acc, 0, 0, 0);
                    } else acc += x;
                    __builtin_amdgcn_sched_barrier(0);
                    r0 = p[(size_t)kb_of(kb + 2 < KB ? kb + 2 : kb) * ks];
                    __builtin_amdgcn_sched_barrier(0);
                }
                {
                    const f4 x = r1;
                    if (MFMA) {
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.x, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.y, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.z, acc, 0, 0, 0);
                        acc = __builtin_amdgcn_mfma_f32_16x16x4f32(1.f, x.w, acc, 0, 0, 0);
                    } else acc += x;
                    __builtin_amdgcn_sched_barrier(0);
                    r1 = p[(size_t)kb_of(kb + 3 < KB ? kb + 3 : kb + 1 < KB ? kb + 1 : kb) * ks];
                    __builtin_amdgcn_sched_barrier(0);
                }
            }
        }
        __syncthreads();                      // the phase barrier of the evaluation
    }
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[b] = t1 - t0;
    sink[(size_t)b * 1024 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}

template <int MODE, bool MFMA>
static void run(const char *name, int grid, const f4 *w, const Args &a, float *sink, long long *cyc, double kib) {
    long long h[256];
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 3; ++rep) stream<MODE, MFMA><<<grid, 1024>>>(w, a, sink, cyc);
    hipEventRecord(e0);
    const int reps = 10;
    for (int rep = 0; rep < reps; ++rep) stream<MODE, MFMA><<<grid, 1024>>>(w, a, sink, cyc);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
    double mean = 0, mx = 0;
    for (int i = 0; i < grid; ++i) { mean += h[i]; if (h[i] > mx) mx = h[i]; }
    mean /= grid;
    printf("| %-9s | %d | %3d | %8.0f | %8.0f | %5.1f | %6.2f | %6.1f |\n", name, MFMA ? 1 : 0, grid, mean, mx, kib * 1024.0 / mean,
           grid * kib / mean, 1e3 * ms / reps);
}

int main() {
    Args a{};
    // one evaluation of the Bibsonomy network: y->600 (10 x 38), {z0, y}->159 (48 x 10), d1 Wzu1^T | d1 Wyu1^T (10 x 48),
    // d0 Wyu0^T (38 x 10)
    const int kb[4] = {10, 48, 10, 38}, nt[4] = {38, 10, 48, 10};
    long long off = 0;
    for (int i = 0; i < 4; ++i) { a.s[i] = Shape{kb[i], nt[i], off}; off += (long long)kb[i] * nt[i]; }
    a.ns = 4;
    const double kib = (double)off;
    f4 *w; float *sink; long long *cyc;
    hipMalloc(&w, (size_t)off * 1024); hipMemset(w, 0, (size_t)off * 1024);
    hipMalloc(&sink, 256 * 1024 * 4); hipMalloc(&cyc, 256 * 8);
    printf("pack: %.0f KiB per evaluation; cycles = s_memtime of a workgroup (mean, max), launch = HIP events\n", kib);
    printf("| order | mfma | grid | cycles mean | cycles max | B/clk/CU | KB/clk chip | us/launch |\n|---|---|---|---|---|---|---|---|\n");
    for (int grid : {1, 32, 256}) {
        run<0, true>("lockstep", grid, w, a, sink, cyc, kib);
        run<1, true>("rot_cu", grid, w, a, sink, cyc, kib);
        run<2, true>("rot_xcd", grid, w, a, sink, cyc, kib);
        run<3, true>("rot_k", grid, w, a, sink, cyc, kib);
        run<4, true>("flat", grid, w, a, sink, cyc, kib);
        run<0, false>("lockstep", grid, w, a, sink, cyc, kib);
        run<1, false>("rot_cu", grid, w, a, sink, cyc, kib);
        run<2, false>("rot_xcd", grid, w, a, sink, cyc, kib);
        run<3, false>("rot_k", grid, w, a, sink, cyc, kib);
    }
    return 0;
}
