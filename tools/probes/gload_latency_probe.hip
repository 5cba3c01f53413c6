// Global-load latency seen by ONE wave on an otherwise idle MI355X (cycles via s_memtime, 100 MHz-independent
// shader clock count): dependent pointer chase over footprints that fit L1 (16 KB), L2 (1 MB), the Infinity
// Cache (64 MB) and HBM (1 GB), plus the cost of a burst of N independent 16-byte loads (what the fragment
// rings of the GEMV / GEMM loops issue).  Build: hipcc --offload-arch=gfx950 -O3 gload_latency_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void chase(const int *next, int start, int steps, long long *out, int *sink) {
    int p = start + threadIdx.x;           // every lane chases its own chain (lanes on consecutive ints: one line)
    for (int i = 0; i < 64; ++i) p = next[p];          // warm
    long long t0 = __builtin_readcyclecounter();
    for (int i = 0; i < steps; ++i) p = next[p];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) out[0] = (t1 - t0) / steps;
    sink[threadIdx.x] = p;
}
typedef float f4 __attribute__((ext_vector_type(4)));
template <int N> __global__ void burst(const f4 *src, int stride_f4, long long *out, float *sink, int reps) {
    f4 acc = {0, 0, 0, 0};
    const f4 *p = src + threadIdx.x;
    for (int i = 0; i < N; ++i) acc += p[(size_t)i * stride_f4];      // warm L2
    long long total = 0;
    for (int r = 0; r < reps; ++r) {
        f4 v[N];
        long long t0 = __builtin_readcyclecounter();
#pragma unroll
        for (int i = 0; i < N; ++i) v[i] = p[(size_t)i * stride_f4];
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        long long t1 = __builtin_readcyclecounter();
        total += t1 - t0;
#pragma unroll
        for (int i = 0; i < N; ++i) acc += v[i];
        asm volatile("buffer_inv sc0 sc1" ::: "memory");                 // drop L1 so that the next round goes to L2
    }
    if (threadIdx.x == 0) out[0] = total / reps;
    sink[threadIdx.x] = acc.x + acc.y + acc.z + acc.w;
}
int main() {
    long long *out; int *sink; float *fsink;
    hipMalloc(&out, 64); hipMalloc(&sink, 4096); hipMalloc(&fsink, 4096);
    const size_t sizes[] = {16u << 10, 1u << 20, 64u << 20, 1u << 30};
    const char *names[] = {"16 KB (L1)", "1 MB (L2)", "64 MB (Infinity Cache)", "1 GB (HBM)"};
    for (int s = 0; s < 4; ++s) {
        const size_t n = sizes[s] / 4, lines = n / 64;     // 256-byte rows of 64 ints; chain hops between rows
        std::vector<int> h(n);
        std::vector<size_t> perm(lines);
        for (size_t i = 0; i < lines; ++i) perm[i] = i;
        srand(1);
        for (size_t i = lines - 1; i > 0; --i) { size_t j = ((size_t)rand() * 32768 + rand()) % (i + 1); std::swap(perm[i], perm[j]); }
        for (size_t i = 0; i < lines; ++i) {
            const size_t from = perm[i], to = perm[(i + 1) % lines];
            for (int l = 0; l < 64; ++l) h[from * 64 + l] = (int)(to * 64 + l);
        }
        int *d; hipMalloc(&d, n * 4); hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
        const int steps = s == 0 ? 2000 : 4000;
        for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL(chase, dim3(1), dim3(64), 0, 0, d, 0, steps, out, sink);
        long long c; hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost);
        printf("dependent load, footprint %-24s %lld cycles\n", names[s], c);
        hipFree(d);
    }
    f4 *src; hipMalloc(&src, 1 << 20); hipMemset(src, 0, 1 << 20);
    long long c;
#define BURST(N) hipLaunchKernelGGL(burst<N>, dim3(1), dim3(64), 0, 0, src, 64 * 13, out, fsink, 200); \
    hipMemcpy(&c, out, 8, hipMemcpyDeviceToHost); printf("burst of %2d independent 1 KiB wave loads from L2: %lld cycles\n", N, c);
    BURST(1) BURST(4) BURST(8) BURST(16) BURST(32)
    return 0;
}
