// The C ABI of libicnn_be.so used from plain C++ -- no Python, no torch: the generic solveBatch loop of the reference
// (lib/bundle_entropy_dual.py:129-179) with a host-side fg, i.e. BASELINE.json configs[0] (a convex quadratic over
// [0,1]^n, float64 cuts).  The caller owns every buffer (hipMalloc), the library only launches kernels on the stream
// it is given.
//
//   hipcc -O2 -I include examples/c_abi_solve.cpp -o c_abi_solve -L icnn_amd/csrc -licnn_be -Wl,-rpath,$PWD/icnn_amd/csrc
//   ./c_abi_solve problem.bin result.bin
//
// problem.bin: int32 B, n, nIter; float64 Q[n][n] (symmetric positive definite), P[B][n], y0[B][n]
// result.bin:  float64 y[B][n]; int32 count[B], n_iters[B], status[B]
// f_u(y) = 1/2 y^T Q y + P_u . y,  grad = Q y + P_u.  tests/test_c_abi_example.py compares the result with
// icnn_amd.bundle_entropy.solveBatch on the same problem (bit-identical: the same kernels on the same inputs).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#include "icnn_be.h"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { std::fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define BE_OK(x) do { int r_ = (x); if (r_ != 0) { std::fprintf(stderr, "%s -> %d (%s)\n", #x, r_, icnn_be_last_hip_error()); return 3; } } while (0)

template <typename T>
static T *dev_alloc(size_t count) {
    void *p = nullptr;
    if (hipMalloc(&p, (count ? count : 1) * sizeof(T)) != hipSuccess) return nullptr;
    (void)hipMemset(p, 0, (count ? count : 1) * sizeof(T));
    return static_cast<T *>(p);
}

int main(int argc, char **argv) {
    if (argc != 3) { std::fprintf(stderr, "usage: %s problem.bin result.bin\n", argv[0]); return 1; }
    std::FILE *in = std::fopen(argv[1], "rb");
    if (!in) { std::perror(argv[1]); return 1; }
    int hdr[3];
    if (std::fread(hdr, sizeof(int), 3, in) != 3) return 1;
    const int B = hdr[0], n = hdr[1], T = hdr[2];
    std::vector<double> Q((size_t)n * n), P((size_t)B * n), y((size_t)B * n);
    if (std::fread(Q.data(), 8, Q.size(), in) != Q.size() || std::fread(P.data(), 8, P.size(), in) != P.size() ||
        std::fread(y.data(), 8, y.size(), in) != y.size())
        return 1;
    std::fclose(in);
    if (icnn_be_abi_version() != ICNN_BE_ABI_VERSION) { std::fprintf(stderr, "ABI mismatch\n"); return 1; }

    hipStream_t stream;
    HIP_OK(hipStreamCreate(&stream));
    icnn_be_state st{};
    st.batch = B; st.n = n; st.slots = T;
    st.cut_dtype = ICNN_BE_CUT_F64; st.variant = ICNN_BE_VARIANT_DUAL; st.flags = 0;
    st.y = dev_alloc<double>((size_t)B * n);
    st.G = dev_alloc<double>((size_t)B * T * n);
    st.h = dev_alloc<double>((size_t)B * T);
    st.ys = dev_alloc<double>((size_t)B * T * n);
    st.lam = dev_alloc<double>((size_t)B * T);
    st.active = dev_alloc<int>((size_t)B * T);
    st.count = dev_alloc<int>(B); st.n_iters = dev_alloc<int>(B); st.finished = dev_alloc<int>(B);
    st.status = dev_alloc<int>(B); st.newton_iters = dev_alloc<int>(B);
    st.t_next = dev_alloc<int>(B); st.phase = dev_alloc<int>(B); st.skip_fg = dev_alloc<int>(B);
    st.pending = dev_alloc<int>(ICNN_BE_MAX_ROUNDS);
    st.park = dev_alloc<double>((size_t)B * (5 * T + 4));
    const size_t scratch = icnn_be_scratch_bytes(&st);          // 0 unless the rows are wider than the LDS holds T of
    st.scratch = scratch ? dev_alloc<char>(scratch) : nullptr;
    double *f_dev = dev_alloc<double>(B), *g_dev = dev_alloc<double>((size_t)B * n);
    HIP_OK(hipMemcpy(st.y, y.data(), y.size() * 8, hipMemcpyHostToDevice));
    BE_OK(icnn_be_state_init(&st, stream));

    std::vector<double> f(B), g((size_t)B * n);
    std::vector<int> finished(B);
    for (int t = 0; t < T; ++t) {                               // dual :141
        HIP_OK(hipMemcpyAsync(y.data(), st.y, y.size() * 8, hipMemcpyDeviceToHost, stream));
        HIP_OK(hipMemcpyAsync(finished.data(), st.finished, B * sizeof(int), hipMemcpyDeviceToHost, stream));
        HIP_OK(hipStreamSynchronize(stream));
        bool all_done = true;
        for (int u = 0; u < B; ++u) all_done = all_done && finished[u];
        if (all_done) break;
        for (int u = 0; u < B; ++u) {                           // fi, gi = fg(x), dual :142 -- the caller's model
            double fu = 0.0;
            for (int i = 0; i < n; ++i) {
                double qy = 0.0;
                for (int j = 0; j < n; ++j) qy += Q[(size_t)i * n + j] * y[(size_t)u * n + j];
                g[(size_t)u * n + i] = qy + P[(size_t)u * n + i];
                fu += y[(size_t)u * n + i] * (0.5 * qy + P[(size_t)u * n + i]);
            }
            f[u] = fu;
        }
        HIP_OK(hipMemcpyAsync(f_dev, f.data(), B * 8, hipMemcpyHostToDevice, stream));
        HIP_OK(hipMemcpyAsync(g_dev, g.data(), g.size() * 8, hipMemcpyHostToDevice, stream));
        BE_OK(icnn_be_dual_step(&st, t, f_dev, g_dev, stream));  // dual :143-174 for the whole batch
    }
    std::vector<int> count(B), n_iters(B), status(B);
    HIP_OK(hipMemcpyAsync(y.data(), st.y, y.size() * 8, hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(count.data(), st.count, B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(n_iters.data(), st.n_iters, B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipMemcpyAsync(status.data(), st.status, B * sizeof(int), hipMemcpyDeviceToHost, stream));
    HIP_OK(hipStreamSynchronize(stream));
    std::FILE *out = std::fopen(argv[2], "wb");
    if (!out) { std::perror(argv[2]); return 1; }
    std::fwrite(y.data(), 8, y.size(), out);
    std::fwrite(count.data(), sizeof(int), B, out);
    std::fwrite(n_iters.data(), sizeof(int), B, out);
    std::fwrite(status.data(), sizeof(int), B, out);
    std::fclose(out);
    double sum = 0.0;
    for (double v : y) sum += v;
    std::printf("solved %d samples of dimension %d in at most %d bundle iterations: sum(y*) = %.12f\n", B, n, T, sum);
    return 0;
}
