// How fast can one workgroup (8 waves, one per CU) stream a 1.7 MB weight pack from L2?
// Each wave reads 1 KiB per instruction (global_load_dwordx4), PF instructions in flight, strided like
// gemm_tiles in be_picnn_fc.hip (wave w takes chunks w, w+8, ...).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f4 __attribute__((ext_vector_type(4)));
template <int PF>
__global__ __launch_bounds__(512) void stream(const f4 *w, int chunks, float *sink, long long *cycles) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const f4 *p = w + lane;
    f4 ring[PF];
    f4 acc = {0, 0, 0, 0};
    __syncthreads();
    const long long t0 = __builtin_readcyclecounter();
    int c = wave;
#pragma unroll
    for (int d = 0; d < PF; ++d) ring[d] = p[(size_t)(c + 8 * d < chunks ? c + 8 * d : wave) * 64];
    for (; c < chunks; c += 8 * PF) {
#pragma unroll
        for (int d = 0; d < PF; ++d) {
            const f4 x = ring[d];
            const int nc = c + 8 * (d + PF);
            ring[d] = p[(size_t)(nc < chunks ? nc : wave) * 64];
            acc += x;
        }
    }
    __syncthreads();
    const long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) cycles[blockIdx.x] = t1 - t0;
    sink[blockIdx.x * 512 + threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    const int chunks = 1728;                 // 1.73 MB in 1 KiB chunks
    f4 *w; float *sink; long long *cyc, h[256];
    hipMalloc(&w, (size_t)chunks * 1024); hipMemset(w, 0, (size_t)chunks * 1024);
    hipMalloc(&sink, 256 * 512 * 4); hipMalloc(&cyc, 256 * 8);
    for (int grid : {1, 256}) {
        for (int pf : {2, 4, 8, 16}) {
            for (int rep = 0; rep < 3; ++rep) {
                if (pf == 2) stream<2><<<grid, 512>>>(w, chunks, sink, cyc);
                if (pf == 4) stream<4><<<grid, 512>>>(w, chunks, sink, cyc);
                if (pf == 8) stream<8><<<grid, 512>>>(w, chunks, sink, cyc);
                if (pf == 16) stream<16><<<grid, 512>>>(w, chunks, sink, cyc);
            }
            hipMemcpy(h, cyc, grid * 8, hipMemcpyDeviceToHost);
            double mean = 0; for (int i = 0; i < grid; ++i) mean += h[i]; mean /= grid;
            printf("grid %3d  PF %2d: %8.0f cycles per workgroup for %d KiB -> %5.1f B/clk per CU\n", grid, pf, mean, chunks,
                   chunks * 1024.0 / mean);
        }
    }
    return 0;
}
