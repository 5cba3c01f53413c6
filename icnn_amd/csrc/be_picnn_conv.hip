// Energy + y-gradient of the y-dependent part of the convolutional PICNN used for image
// completion (completion/icnn_ebundle.py:376-452; gradient = tf.gradients(E_, y_), :120-121).
//
//   y_red_0 = y,  y_red_{l+1} = conv(y_red_l; k_l, s_l) + b                               (:394-396)
//   z_l = relu( conv(z_{l-1} * gate_l; Wzu_l >= 0)[l>0] + conv(y_red_l * yu_l; Wyu_l) + zu_l ),  l = 0..2
//   z_3 = relu( (flatten(z_2) * gate_3) W_3 + zu_3 ),   E = (z_3 * gate_4) . w_4 + zu_4    (:411-445)
//
// One workgroup = one sample: every activation of the chain (13 k floats) lives in LDS for the
// forward and the backward sweep; the x-only context (gate, yu, zu; 19 617 floats per sample)
// and the weights are read with coalesced loads (output-channel fastest forward, a transposed
// copy input-channel fastest backward).  The contractions here are small strided convolutions
// (k8/s4, k4/s2, k3/s1 on <= 16x8 maps) plus one 2048x512 GEMV per sample; they run on the VALU.
// NHWC, 'SAME' padding (symmetric for these kernel/stride pairs), float32 like the reference.
#include <hip/hip_runtime.h>

#include "be_common.h"
#include "be_kernels.h"
#include "icnn_be.h"

namespace icnn_be {

namespace {

constexpr int CT = 512;   // threads per workgroup

struct ConvArgs {
    int H, W, F[3], K[3], S[3], P[3];      // image, filters / kernel / stride / pad per conv layer
    int oh[3], ow[3];                      // output map of each conv layer
    int fch, flat, n, C;                   // fc width, flattened conv output, H*W, ctx width
    // context offsets (floats)
    int c_yu[3], c_zu[3], c_gate[5], c_zu3, c_zu4;
    // weight offsets (floats)
    long long w_yu[3], w_yr[2], b_yr[2], w_zu[3], w_zut[3], w_fc3, w_fc3t, w_fc4;
    const float *wpack, *ctx;
    const double *y;
    float *f, *g;
    const int *skip;
    int batch;
};

// one output element of a forward convolution: in [IH][IW][Cin], W [K][K][Cin][F]
__device__ __forceinline__ float conv_out(const float *in, int IH, int IW, int Cin, const float *W, int K, int S,
                                          int P, int F, int oy, int ox, int f) {
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky) {
        const int iy = oy * S + ky - P;
        if (iy < 0 || iy >= IH) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int ix = ox * S + kx - P;
            if (ix < 0 || ix >= IW) continue;
            const float *ip = in + (iy * IW + ix) * Cin;
            const float *wp = W + ((ky * K + kx) * Cin) * F + f;
#pragma unroll 8
            for (int c = 0; c < Cin; ++c) acc = __builtin_fmaf(ip[c], wp[c * F], acc);
        }
    }
    return acc;
}

// gradient w.r.t. one input element: dout [OH][OW][F], Wt [K][K][F][Cin] (transposed copy)
__device__ __forceinline__ float conv_din(const float *dout, int OH, int OW, int F, const float *Wt, int K, int S,
                                          int P, int Cin, int iy, int ix, int c) {
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky) {
        const int ty = iy + P - ky;
        if (ty < 0 || ty % S) continue;
        const int oy = ty / S;
        if (oy >= OH) continue;
        for (int kx = 0; kx < K; ++kx) {
            const int tx = ix + P - kx;
            if (tx < 0 || tx % S) continue;
            const int ox = tx / S;
            if (ox >= OW) continue;
            const float *dp = dout + (oy * OW + ox) * F;
            const float *wp = Wt + ((ky * K + kx) * F) * Cin + c;
#pragma unroll 8
            for (int f = 0; f < F; ++f) acc = __builtin_fmaf(dp[f], wp[f * Cin], acc);
        }
    }
    return acc;
}

__global__ __launch_bounds__(CT) void conv_fg_kernel(ConvArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.skip && a.skip[u]) return;
    const int n = a.n, H = a.H, W = a.W;
    const int p1 = a.oh[0] * a.ow[0], p2 = a.oh[1] * a.ow[1], p3 = a.oh[2] * a.ow[2];
    const int n1 = p1 * a.F[0], n2 = p2 * a.F[1], n3 = p3 * a.F[2];
    float *ybuf = lds, *a0 = ybuf + n, *yr1 = a0 + n, *yr2 = yr1 + p1, *ay1 = yr2 + p2, *ay2 = ay1 + p1;
    float *dyr1 = ay2 + p2, *dyr2 = dyr1 + p1;
    float *A1 = dyr2 + p2, *A2 = A1 + n1, *A3 = A2 + n2, *A4 = A3 + n3, *red = A4 + a.fch;
    const float *ctx = a.ctx + (size_t)u * a.C;
    const float *wp = a.wpack;

    // P0: y (rounded to float32 like a TensorFlow feed), y * yu_0
    for (int j = tid; j < n; j += CT) {
        const float v = (float)a.y[(size_t)u * n + j];
        ybuf[j] = v;
        a0[j] = v * ctx[a.c_yu[0] + j];
    }
    __syncthreads();
    // P1: y_red_1 and z_0 -> A1 = z_0 * gate_1
    for (int j = tid; j < p1; j += CT)
        yr1[j] = conv_out(ybuf, H, W, 1, wp + a.w_yr[0], a.K[0], a.S[0], a.P[0], 1, j / a.ow[0], j % a.ow[0], 0)
                 + wp[a.b_yr[0]];
    for (int e = tid; e < n1; e += CT) {
        const int f = e % a.F[0], pos = e / a.F[0];
        const float pre = conv_out(a0, H, W, 1, wp + a.w_yu[0], a.K[0], a.S[0], a.P[0], a.F[0], pos / a.ow[0],
                                   pos % a.ow[0], f) + ctx[a.c_zu[0] + e];
        A1[e] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[1] + e];
    }
    __syncthreads();
    // P2: y_red_1 * yu_1, y_red_2
    for (int j = tid; j < p1; j += CT) ay1[j] = yr1[j] * ctx[a.c_yu[1] + j];
    for (int j = tid; j < p2; j += CT)
        yr2[j] = conv_out(yr1, a.oh[0], a.ow[0], 1, wp + a.w_yr[1], a.K[1], a.S[1], a.P[1], 1, j / a.ow[1],
                          j % a.ow[1], 0) + wp[a.b_yr[1]];
    __syncthreads();
    // P3: z_1 -> A2 = z_1 * gate_2 ; y_red_2 * yu_2
    for (int e = tid; e < n2; e += CT) {
        const int f = e % a.F[1], pos = e / a.F[1], oy = pos / a.ow[1], ox = pos % a.ow[1];
        float pre = conv_out(A1, a.oh[0], a.ow[0], a.F[0], wp + a.w_zu[1], a.K[1], a.S[1], a.P[1], a.F[1], oy, ox, f);
        pre = pre + conv_out(ay1, a.oh[0], a.ow[0], 1, wp + a.w_yu[1], a.K[1], a.S[1], a.P[1], a.F[1], oy, ox, f);
        pre = pre + ctx[a.c_zu[1] + e];
        A2[e] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[2] + e];
    }
    for (int j = tid; j < p2; j += CT) ay2[j] = yr2[j] * ctx[a.c_yu[2] + j];
    __syncthreads();
    // P4: z_2 -> A3 = flatten(z_2) * gate_3
    for (int e = tid; e < n3; e += CT) {
        const int f = e % a.F[2], pos = e / a.F[2], oy = pos / a.ow[2], ox = pos % a.ow[2];
        float pre = conv_out(A2, a.oh[1], a.ow[1], a.F[1], wp + a.w_zu[2], a.K[2], a.S[2], a.P[2], a.F[2], oy, ox, f);
        pre = pre + conv_out(ay2, a.oh[1], a.ow[1], 1, wp + a.w_yu[2], a.K[2], a.S[2], a.P[2], a.F[2], oy, ox, f);
        pre = pre + ctx[a.c_zu[2] + e];
        A3[e] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[3] + e];
    }
    __syncthreads();
    // P5: z_3 -> A4 = z_3 * gate_4   (one output per thread, weights read output-fastest)
    for (int j = tid; j < a.fch; j += CT) {
        const float *w3 = wp + a.w_fc3 + j;
        float acc = 0.f;
#pragma unroll 16
        for (int k = 0; k < a.flat; ++k) acc = __builtin_fmaf(A3[k], w3[(size_t)k * a.fch], acc);
        const float pre = acc + ctx[a.c_zu3 + j];
        A4[j] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[4] + j];
    }
    __syncthreads();
    // P6: energy
    {
        float part = 0.f;
        for (int j = tid; j < a.fch; j += CT) part = __builtin_fmaf(A4[j], wp[a.w_fc4 + j], part);
        part = wave_sum_f(part);
        if (lane == 0) red[wave] = part;
        __syncthreads();
        if (tid == 0) {
            float e = 0.f;
            for (int w = 0; w < CT / 64; ++w) e += red[w];
            a.f[u] = e + ctx[a.c_zu4];
        }
    }
    // P7: delta_3 = gate_4 * w_4 * [z_3 > 0]
    for (int j = tid; j < a.fch; j += CT) {
        const float gw = ctx[a.c_gate[4] + j] * wp[a.w_fc4 + j];
        A4[j] = A4[j] > 0.f ? gw : 0.f;
    }
    __syncthreads();
    // P8: delta_2 = gate_3 * (W_3 delta_3) * [z_2 > 0]   (one output per thread from the transposed copy:
    //     coalesced weight reads, no cross-lane reduction)
    for (int k = tid; k < a.flat; k += CT) {
        const float *w3t = wp + a.w_fc3t + k;
        float acc = 0.f;
#pragma unroll 16
        for (int j = 0; j < a.fch; ++j) acc = __builtin_fmaf(w3t[(size_t)j * a.flat], A4[j], acc);
        const float dz = ctx[a.c_gate[3] + k] * acc;
        A3[k] = A3[k] > 0.f ? dz : 0.f;
    }
    __syncthreads();
    // P9: delta_1 = gate_2 * convT(delta_2; Wzu_2) * [z_1 > 0] ; d y_red_2 = yu_2 * convT(delta_2; Wyu_2)
    for (int e = tid; e < n2; e += CT) {
        const int c = e % a.F[1], pos = e / a.F[1];
        const float d = conv_din(A3, a.oh[2], a.ow[2], a.F[2], wp + a.w_zut[2], a.K[2], a.S[2], a.P[2], a.F[1],
                                 pos / a.ow[1], pos % a.ow[1], c);
        const float dz = ctx[a.c_gate[2] + e] * d;
        A2[e] = A2[e] > 0.f ? dz : 0.f;
    }
    for (int j = tid; j < p2; j += CT)
        dyr2[j] = ctx[a.c_yu[2] + j] * conv_din(A3, a.oh[2], a.ow[2], a.F[2], wp + a.w_yu[2], a.K[2], a.S[2], a.P[2],
                                                 1, j / a.ow[1], j % a.ow[1], 0);
    __syncthreads();
    // P10: delta_0 ; d y_red_1 = yu_1 * convT(delta_1; Wyu_1) + convT(d y_red_2; Wyr_1)
    for (int e = tid; e < n1; e += CT) {
        const int c = e % a.F[0], pos = e / a.F[0];
        const float d = conv_din(A2, a.oh[1], a.ow[1], a.F[1], wp + a.w_zut[1], a.K[1], a.S[1], a.P[1], a.F[0],
                                 pos / a.ow[0], pos % a.ow[0], c);
        const float dz = ctx[a.c_gate[1] + e] * d;
        A1[e] = A1[e] > 0.f ? dz : 0.f;
    }
    for (int j = tid; j < p1; j += CT) {
        const int iy = j / a.ow[0], ix = j % a.ow[0];
        const float t1 = ctx[a.c_yu[1] + j] * conv_din(A2, a.oh[1], a.ow[1], a.F[1], wp + a.w_yu[1], a.K[1], a.S[1],
                                                       a.P[1], 1, iy, ix, 0);
        dyr1[j] = t1 + conv_din(dyr2, a.oh[1], a.ow[1], 1, wp + a.w_yr[1], a.K[1], a.S[1], a.P[1], 1, iy, ix, 0);
    }
    __syncthreads();
    // P11: dE/dy = yu_0 * convT(delta_0; Wyu_0) + convT(d y_red_1; Wyr_0)
    for (int j = tid; j < n; j += CT) {
        const int iy = j / W, ix = j % W;
        const float t0 = ctx[a.c_yu[0] + j] * conv_din(A1, a.oh[0], a.ow[0], a.F[0], wp + a.w_yu[0], a.K[0], a.S[0],
                                                       a.P[0], 1, iy, ix, 0);
        a.g[(size_t)u * n + j] = t0 + conv_din(dyr1, a.oh[0], a.ow[0], 1, wp + a.w_yr[0], a.K[0], a.S[0], a.P[0], 1,
                                               iy, ix, 0);
    }
}

struct ConvLayout {
    ConvArgs a;
    size_t pack_floats;
    int lds_bytes;
};

int conv_layout(const icnn_be_conv_model &m, ConvLayout &L) {
    ConvArgs &a = L.a;
    if (m.H < 1 || m.W < 1 || m.fc_hidden < 1) return ICNN_BE_EINVAL;
    a.H = m.H; a.W = m.W; a.n = m.H * m.W; a.fch = m.fc_hidden;
    int h = m.H, w = m.W, cin = 1, o = 0;
    long long wo = 0;
    for (int l = 0; l < 3; ++l) {
        const int k = m.ksize[l], s = m.stride[l], f = m.filters[l];
        if (k < 1 || s < 1 || f < 1) return ICNN_BE_EINVAL;
        const int oh = (h + s - 1) / s, ow = (w + s - 1) / s;
        const int ph = (oh - 1) * s + k - h, pw = (ow - 1) * s + k - w;
        // 'SAME' padding must be symmetric for this kernel (true for 8/4, 4/2, 3/1 on the reference's maps)
        if (ph < 0 || pw < 0 || ph % 2 || pw % 2 || ph != pw) return ICNN_BE_EINVAL;
        a.F[l] = f; a.K[l] = k; a.S[l] = s; a.P[l] = ph / 2; a.oh[l] = oh; a.ow[l] = ow;
        if (l > 0) { a.c_gate[l] = o; o += h * w * cin; }
        a.c_yu[l] = o; o += h * w;
        a.c_zu[l] = o; o += oh * ow * f;
        a.w_yu[l] = wo; wo += (long long)k * k * f;
        if (l < 2) { a.w_yr[l] = wo; wo += k * k; a.b_yr[l] = wo; wo += 1; }
        if (l > 0) {
            a.w_zu[l] = wo; wo += (long long)k * k * cin * f;
            a.w_zut[l] = wo; wo += (long long)k * k * cin * f;
        }
        h = oh; w = ow; cin = f;
    }
    a.flat = h * w * cin;
    a.c_gate[3] = o; o += a.flat;
    a.c_zu3 = o; o += a.fch;
    a.c_gate[4] = o; o += a.fch;
    a.c_zu4 = o; o += 1;
    if (o != m.ctx_width) return ICNN_BE_EINVAL;
    a.C = o;
    a.w_fc3 = wo; wo += (long long)a.flat * a.fch;
    a.w_fc3t = wo; wo += (long long)a.flat * a.fch;
    a.w_fc4 = wo; wo += a.fch;
    L.pack_floats = (size_t)wo;
    const int p1 = a.oh[0] * a.ow[0], p2 = a.oh[1] * a.ow[1], p3 = a.oh[2] * a.ow[2];
    const int floats = 2 * a.n + 3 * p1 + 3 * p2 + p1 * a.F[0] + p2 * a.F[1] + p3 * a.F[2] + a.fch + 16;
    L.lds_bytes = floats * 4;
    if (L.lds_bytes > 160 * 1024) return ICNN_BE_ELIMIT;
    a.wpack = m.wpack;
    return 0;
}

}  // namespace

size_t conv_pack_floats(const icnn_be_conv_model &m) {
    ConvLayout L{};
    return conv_layout(m, L) == 0 ? L.pack_floats : 0;
}

// w_yu[l]: 'z{l}_yu/W' [k][k][1][F]; w_yr[l], b_yr[l] (l = 0, 1): 'z{l}_y_red/W' [k][k][1][1], '/b';
// w_zu[l] (l = 1, 2): 'z{l}_zu_proj/W' [k][k][Cin][F]; w_fc3 [flat][fch]; w_fc4 [fch][1]
int conv_pack(const icnn_be_conv_model &m, const float *const *w_yu, const float *const *w_yr,
              const float *const *b_yr, const float *const *w_zu, const float *w_fc3, const float *w_fc4,
              float *out) {
    ConvLayout L{};
    if (int rc = conv_layout(m, L)) return rc;
    const ConvArgs &a = L.a;
    int cin = 1;
    for (int l = 0; l < 3; ++l) {
        const int k = a.K[l], f = a.F[l];
        for (int i = 0; i < k * k * f; ++i) out[a.w_yu[l] + i] = w_yu[l][i];
        if (l < 2) {
            for (int i = 0; i < k * k; ++i) out[a.w_yr[l] + i] = w_yr[l][i];
            out[a.b_yr[l]] = b_yr[l][0];
        }
        if (l > 0) {
            for (int t = 0; t < k * k; ++t)
                for (int c = 0; c < cin; ++c)
                    for (int ff = 0; ff < f; ++ff) {
                        const float v = w_zu[l][((size_t)t * cin + c) * f + ff];
                        out[a.w_zu[l] + ((size_t)t * cin + c) * f + ff] = v;
                        out[a.w_zut[l] + ((size_t)t * f + ff) * cin + c] = v;
                    }
        }
        cin = f;
    }
    for (size_t i = 0; i < (size_t)a.flat * a.fch; ++i) out[a.w_fc3 + i] = w_fc3[i];
    for (int k = 0; k < a.flat; ++k)
        for (int j = 0; j < a.fch; ++j) out[a.w_fc3t + (size_t)j * a.flat + k] = w_fc3[(size_t)k * a.fch + j];
    for (int i = 0; i < a.fch; ++i) out[a.w_fc4 + i] = w_fc4[i];
    return 0;
}

int conv_check_model(const icnn_be_conv_model &m) {
    ConvLayout L{};
    return conv_layout(m, L);
}

hipError_t launch_conv_fg(const icnn_be_conv_model &m, const float *ctx, const double *y, int batch, float *f,
                          float *g, const int *skip, hipStream_t stream) {
    ConvLayout L{};
    if (conv_layout(m, L) != 0) return hipErrorInvalidValue;
    ConvArgs a = L.a;
    a.ctx = ctx; a.y = y; a.f = f; a.g = g; a.skip = skip; a.batch = batch;
    static int configured = 0;
    if (L.lds_bytes > configured) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void *>(conv_fg_kernel),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, L.lds_bytes);
        if (e != hipSuccess) return e;
        configured = L.lds_bytes;
    }
    hipLaunchKernelGGL(conv_fg_kernel, dim3(batch), dim3(CT), L.lds_bytes, stream, a);
    return hipGetLastError();
}

}  // namespace icnn_be
