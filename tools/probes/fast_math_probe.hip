// Accuracy of fast_exp / fast_log (be_dual_dev.h) against the library routines, on the GPU: max relative error over dense
// samples of the ranges the dual step feeds them.
#include "be_dual_dev.h"
#include <cstdio>
#include <cmath>
using namespace icnn_be;
__global__ void k(double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double u = (i + 0.5) / n;
    const double x = -60.0 + 120.0 * u;                 // exp argument
    const double e0 = exp(x), e1 = fast_exp(x);
    out[i] = fabs(e1 - e0) / e0;
    const double t = exp(-40.0 + 80.0 * u);             // log argument: y / (1 - y) over 35 decades
    const double l0 = log(t), l1 = fast_log(t);
    out[n + i] = fabs(l1 - l0) / fmax(fabs(l0), 1e-300);
    const double y = u;                                 // near t = 1: absolute error matters
    const double t2 = y / (1.0 - y);
    out[2 * n + i] = fabs(fast_log(t2) - log(t2));
    out[3 * n + i] = fabs(fast_exp(x * 12.0) - exp(x * 12.0)) / fmax(exp(x * 12.0), 1e-300);   // +-720: under/overflow ends
}
int main() {
    const int n = 1 << 22;
    double *d, *h = new double[4 * n];
    hipMalloc(&d, 4 * n * sizeof(double));
    k<<<n / 256, 256>>>(d, n);
    hipMemcpy(h, d, 4 * n * sizeof(double), hipMemcpyDeviceToHost);
    const char *names[4] = {"exp  rel, x in [-60, 60]", "log  rel, t in [e^-40, e^40]", "log  abs, t = y/(1-y), y in (0,1)", "exp  rel, x in [-720, 720]"};
    for (int q = 0; q < 4; ++q) {
        double mx = 0; int bad = 0;
        for (int i = 0; i < n; ++i) { const double v = h[q * n + i]; if (!(v == v)) ++bad; else if (v > mx && v < 1e300) mx = v; }
        printf("%-36s max %.3e  (nan %d)\n", names[q], mx, bad);
    }
    return 0;
}
