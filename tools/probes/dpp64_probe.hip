#include <hip/hip_runtime.h>
template <int P> __device__ __forceinline__ void fmac_bcast(double &acc, double src, double mul) {
    asm volatile("v_fmac_f64_dpp %0, %1, %2 row_newbcast:%3 row_mask:0xf bank_mask:0xf" : "+v"(acc) : "v"(src), "v"(mul), "n"(P));
}
template <int P> __device__ __forceinline__ double mov_bcast(double src) {
    double r;
    asm volatile("v_mov_b64_dpp %0, %1 row_newbcast:%2 row_mask:0xf bank_mask:0xf" : "=v"(r) : "v"(src), "n"(P));
    return r;
}
__global__ void k(double *out, const double *in) {
    double m = in[threadIdx.x], acc = in[64 + threadIdx.x], f = in[128 + threadIdx.x];
    fmac_bcast<3>(acc, m, f);          // acc += m[lane 3 of my row] * f
    out[threadIdx.x] = acc;
    out[64 + threadIdx.x] = mov_bcast<5>(m);
}
int main() {
    double *d_in, *d_out, h_in[192], h_out[128];
    for (int i = 0; i < 192; ++i) h_in[i] = i * 0.5 + 1;
    hipMalloc(&d_in, sizeof(h_in)); hipMalloc(&d_out, sizeof(h_out));
    hipMemcpy(d_in, h_in, sizeof(h_in), hipMemcpyHostToDevice);
    k<<<1, 64>>>(d_out, d_in);
    hipMemcpy(h_out, d_out, sizeof(h_out), hipMemcpyDeviceToHost);
    int bad = 0;
    for (int l = 0; l < 64; ++l) {
        double want = h_in[64 + l] + h_in[(l & ~15) + 3] * h_in[128 + l];
        if (h_out[l] != want) ++bad;
        if (h_out[64 + l] != h_in[(l & ~15) + 5]) ++bad;
    }
    printf("dpp64 probe: %s (%d bad) out[0]=%g out[17]=%g mov[20]=%g\n", bad ? "FAIL" : "OK", bad, h_out[0], h_out[17], h_out[84]);
    return bad != 0;
}
