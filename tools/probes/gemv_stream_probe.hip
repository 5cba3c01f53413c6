// What limits the weight stream of the latency-path GEMV (be_adam.hip gemv_chain)?  One workgroup on an idle GPU,
// W waves, each reading 4 KiB per k-block (4 x global_load_dwordx4) out of an NT KiB wide k-block row, AHEAD k-blocks
// in flight.  Pattern 0: the packed-fragment addressing of the kernel (per instruction four 256-byte pieces 1 KiB
// apart); pattern 1: 1 KiB contiguous per instruction.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PATTERN, int AHEAD>
__global__ __launch_bounds__(1024) void stream(const f4 *w, int KB, int NT, float *sink, long long *cycles, int reps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const size_t ks = (size_t)NT * 64;                                  // f4 per k-block
    const f4 *bp = PATTERN == 0 ? w + (size_t)(wave * 4 + (lane >> 4)) * 64 + (lane & 15)     // column tile = 4 wave + lane/16
                                : w + (size_t)wave * 256 + lane;
    const int qs = PATTERN == 0 ? 16 : 64;
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    for (int r = 0; r < reps; ++r) {
        f4 ring[AHEAD][4];
#pragma unroll
        for (int d = 0; d < AHEAD; ++d)
#pragma unroll
            for (int q = 0; q < 4; ++q) ring[d][q] = bp[(size_t)d * ks + q * qs];
        for (int kb0 = 0; kb0 < KB; kb0 += AHEAD) {
#pragma unroll
            for (int d = 0; d < AHEAD; ++d) {
                const int nk = kb0 + d + AHEAD < KB ? kb0 + d + AHEAD : kb0 + d;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    acc += ring[d][q];
                    ring[d][q] = bp[(size_t)nk * ks + q * qs];
                }
            }
        }
        asm volatile("buffer_inv sc0 sc1" ::: "memory");
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[0] = (t1 - t0) / reps;
    sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    const int KB = 15, NT = 13;
    f4 *w; float *sink; long long *cyc, h;
    hipMalloc(&w, 4 << 20); hipMemset(w, 0, 4 << 20);
    hipMalloc(&sink, 4096); hipMalloc(&cyc, 64);
#define RUN(P, A, W)                                                                                     \
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((stream<P, A>), dim3(1), dim3(64 * W), 0, 0, w, KB, W * 4, sink, cyc, 50); \
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);                                                            \
    printf("pattern %d ahead %d waves %2d: %6lld cycles for %3d KiB -> %5.1f B/clk (%4lld cycles per k-block)\n", P, A, W, h, \
           KB * W * 4, KB * W * 4096.0 / h, h / KB);
    RUN(0, 3, 4) RUN(1, 3, 4) RUN(0, 5, 4) RUN(1, 5, 4) RUN(0, 1, 4) RUN(1, 1, 4)
    RUN(0, 3, 1) RUN(1, 3, 1) RUN(0, 3, 2) RUN(0, 3, 8) RUN(1, 3, 8) RUN(0, 3, 16) RUN(1, 3, 16)
    (void)NT;
    return 0;
}
