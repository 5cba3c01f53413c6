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
    long long *prof;     // diagnostic: [sample][CONV_PROF_PHASES] cycle counters (wave 0), else nullptr
};
constexpr int CONV_PROF_PHASES = 16;

// Every map lives in LDS with a zero border (PADI pixels for the image-sized buffers, 1 pixel for the
// feature maps), so that neither the forward convolutions ('SAME' padding <= border) nor the transposed ones
// (an output position just outside the map contributes zero) need bounds checks.
constexpr int PADI = 2;    // image border: the first convolution (k8/s4) pads by 2
constexpr int PADM = 1;    // feature-map border: k4/s2 and k3/s1 pad by 1; transposed taps reach 1 outside
constexpr int NBMAX = 8;   // output positions per thread that share every weight load

struct Map {               // a padded [h][w][c] buffer in LDS
    float *p;
    int w, c, pad;
    __device__ __forceinline__ int at(int y, int x) const { return ((y + pad) * (w + 2 * pad) + x + pad) * c; }
};

// NB output positions of one output channel `ch`, all computed from the same stream of weights (one weight
// load feeds NB fused multiply-adds; the first version re-read every weight once per output position).
//   forward    (TR = false): out[oy][ox][ch] += sum_{ky,kx,r} in[oy*S+ky-P][ox*S+kx-P][r] * W[ky][kx][r][ch]
//   transposed (TR = true):  din[iy][ix][ch] += sum_{ky = (iy+P) mod S, +S, ..; kx likewise} sum_r
//                                               dout[(iy+P-ky)/S][(ix+P-kx)/S][r] * Wt[ky][kx][r][ch]
// (W is [K][K][R][Cout] in both cases: the forward pack for TR = false, the transposed copy for TR = true.)
// For TR the positions of one thread must share (y+P) mod S -- guaranteed by the thread maps below.
template <bool TR, int NB>
__device__ __forceinline__ void conv_block(const Map in, int R, const float *W, int Cout, int ch, int K, int S, int P,
                                           const int (&py)[NBMAX], const int (&px)[NBMAX], float (&acc)[NBMAX]) {
    const int ry = TR ? (py[0] + P) % S : 0, rx = TR ? (px[0] + P) % S : 0;
    const int step = TR ? S : 1;
    for (int ky = ry; ky < K; ky += step)
        for (int kx = rx; kx < K; kx += step) {
            int off[NBMAX];
#pragma unroll
            for (int i = 0; i < NB; ++i)
                off[i] = TR ? in.at((py[i] + P - ky) / S, (px[i] + P - kx) / S) : in.at(py[i] * S + ky - P, px[i] * S + kx - P);
            const float *wp = W + (size_t)((ky * K + kx) * R) * Cout + ch;
#pragma unroll(NB >= 8 ? 4 : 8)
            for (int r = 0; r < R; ++r) {
                const float w = wp[(size_t)r * Cout];
#pragma unroll
                for (int i = 0; i < NB; ++i) acc[i] = __builtin_fmaf(in.p[off[i] + r], w, acc[i]);
            }
        }
}

// positions slot, slot + nslots, ... of a map with `npos` positions and width `mw`; unused entries repeat the first
__device__ __forceinline__ int block_positions(int slot, int nslots, int npos, int mw, int (&py)[NBMAX], int (&px)[NBMAX]) {
    int nb = 0;
#pragma unroll
    for (int i = 0; i < NBMAX; ++i) {
        const int p = slot + nslots * i;
        const bool ok = p < npos;
        const int q = ok ? p : slot;
        py[i] = q / mw;
        px[i] = q % mw;
        nb += ok;
    }
    return nb;
}

__global__ __launch_bounds__(CT) void conv_fg_kernel(ConvArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (a.skip && a.skip[u]) return;
    const int n = a.n, H = a.H, W = a.W;
    const int oh0 = a.oh[0], ow0 = a.ow[0], oh1 = a.oh[1], ow1 = a.ow[1], oh2 = a.oh[2], ow2 = a.ow[2];
    const int p1 = oh0 * ow0, p2 = oh1 * ow1, p3 = oh2 * ow2;
    const int F0 = a.F[0], F1 = a.F[1], F2 = a.F[2];
    // padded LDS buffers
    const int simg = (H + 2 * PADI) * (W + 2 * PADI), sm1 = (oh0 + 2) * (ow0 + 2), sm2 = (oh1 + 2) * (ow1 + 2),
              sm3 = (oh2 + 2) * (ow2 + 2);
    float *base = lds;
    auto take = [&](int floats) { float *q = base; base += floats; return q; };
    const Map ybuf{take(simg), W, 1, PADI}, a0{take(simg), W, 1, PADI};
    const Map yr1{take(sm1), ow0, 1, PADM}, ay1{take(sm1), ow0, 1, PADM}, dyr1{take(sm1), ow0, 1, PADM};
    const Map yr2{take(sm2), ow1, 1, PADM}, ay2{take(sm2), ow1, 1, PADM}, dyr2{take(sm2), ow1, 1, PADM};
    const Map A1{take(sm1 * F0), ow0, F0, PADM}, A2{take(sm2 * F1), ow1, F1, PADM}, A3{take(sm3 * F2), ow2, F2, PADM};
    float *A4 = take(a.fch), *red = take(16);
    const int lds_floats = (int)(base - lds);
    const float *ctx = a.ctx + (size_t)u * a.C;
    const float *wp = a.wpack;
    long long tick = a.prof ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int phase) {          // diagnostic only (tools/conv_phase_profile.py)
        if (a.prof) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (tid == 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(a.prof) + (size_t)u * CONV_PROF_PHASES + phase,
                          (unsigned long long)(now - tick));
            tick = now;
        }
    };
    int py[NBMAX], px[NBMAX];
    float acc[NBMAX], acc2[NBMAX];
    auto clear = [&](float (&v)[NBMAX]) {
#pragma unroll
        for (int i = 0; i < NBMAX; ++i) v[i] = 0.f;
    };

    // borders (and everything else) start at zero
    for (int e = tid; e < lds_floats; e += CT) lds[e] = 0.f;
    __syncthreads();
    // P0: y (rounded to float32 like a TensorFlow feed), y * yu_0
    for (int j = tid; j < n; j += CT) {
        const float v = (float)a.y[(size_t)u * n + j];
        ybuf.p[ybuf.at(j / W, j % W)] = v;
        a0.p[a0.at(j / W, j % W)] = v * ctx[a.c_yu[0] + j];
    }
    __syncthreads();
    lap(0);
    // P1: y_red_1 and z_0 -> A1 = z_0 * gate_1
    if (tid < p1) {
        block_positions(tid, p1, p1, ow0, py, px);
        clear(acc);
        conv_block<false, 1>(ybuf, 1, wp + a.w_yr[0], 1, 0, a.K[0], a.S[0], a.P[0], py, px, acc);
        yr1.p[yr1.at(py[0], px[0])] = acc[0] + wp[a.b_yr[0]];
    }
    {
        const int ch = tid % F0, slot = tid / F0, nslots = CT / F0;
        const int nb = block_positions(slot, nslots, p1, ow0, py, px);
        clear(acc);
        conv_block<false, 8>(a0, 1, wp + a.w_yu[0], F0, ch, a.K[0], a.S[0], a.P[0], py, px, acc);
#pragma unroll
        for (int i = 0; i < NBMAX; ++i)
            if (i < nb) {
                const int e = (py[i] * ow0 + px[i]) * F0 + ch;
                const float pre = acc[i] + ctx[a.c_zu[0] + e];
                A1.p[A1.at(py[i], px[i]) + ch] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[1] + e];
            }
    }
    __syncthreads();
    lap(1);
    // P2: y_red_1 * yu_1, y_red_2
    if (tid < p1) ay1.p[ay1.at(tid / ow0, tid % ow0)] = yr1.p[yr1.at(tid / ow0, tid % ow0)] * ctx[a.c_yu[1] + tid];
    if (tid < p2) {
        block_positions(tid, p2, p2, ow1, py, px);
        clear(acc);
        conv_block<false, 1>(yr1, 1, wp + a.w_yr[1], 1, 0, a.K[1], a.S[1], a.P[1], py, px, acc);
        yr2.p[yr2.at(py[0], px[0])] = acc[0] + wp[a.b_yr[1]];
    }
    __syncthreads();
    lap(2);
    // P3: z_1 -> A2 = z_1 * gate_2 ; y_red_2 * yu_2
    {
        const int ch = tid % F1, slot = tid / F1, nslots = CT / F1;
        const int nb = block_positions(slot, nslots, p2, ow1, py, px);
        clear(acc);
        clear(acc2);
        conv_block<false, 4>(A1, F0, wp + a.w_zu[1], F1, ch, a.K[1], a.S[1], a.P[1], py, px, acc);
        conv_block<false, 4>(ay1, 1, wp + a.w_yu[1], F1, ch, a.K[1], a.S[1], a.P[1], py, px, acc2);
#pragma unroll
        for (int i = 0; i < NBMAX; ++i)
            if (i < nb) {
                const int e = (py[i] * ow1 + px[i]) * F1 + ch;
                float pre = acc[i] + acc2[i];
                pre = pre + ctx[a.c_zu[1] + e];
                A2.p[A2.at(py[i], px[i]) + ch] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[2] + e];
            }
    }
    if (tid < p2) ay2.p[ay2.at(tid / ow1, tid % ow1)] = yr2.p[yr2.at(tid / ow1, tid % ow1)] * ctx[a.c_yu[2] + tid];
    __syncthreads();
    lap(3);
    // P4: z_2 -> A3 = flatten(z_2) * gate_3
    {
        const int ch = tid % F2, slot = tid / F2, nslots = CT / F2;
        const int nb = block_positions(slot, nslots, p3, ow2, py, px);
        clear(acc);
        clear(acc2);
        conv_block<false, 4>(A2, F1, wp + a.w_zu[2], F2, ch, a.K[2], a.S[2], a.P[2], py, px, acc);
        conv_block<false, 4>(ay2, 1, wp + a.w_yu[2], F2, ch, a.K[2], a.S[2], a.P[2], py, px, acc2);
#pragma unroll
        for (int i = 0; i < NBMAX; ++i)
            if (i < nb) {
                const int e = (py[i] * ow2 + px[i]) * F2 + ch;
                float pre = acc[i] + acc2[i];
                pre = pre + ctx[a.c_zu[2] + e];
                A3.p[A3.at(py[i], px[i]) + ch] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[3] + e];
            }
    }
    __syncthreads();
    lap(4);
    // P5: z_3 -> A4 = z_3 * gate_4.  Thread (jq, part): four neighbouring outputs (one 16-B weight load feeds
    //     four fmas, 1 KiB per wave-instruction) over a quarter of the positions; the four partial sums of an
    //     output meet in LDS and are added in order.
    {
        const int nq = a.fch / 4, parts = CT / nq;            // 128 output quads x 4 position ranges
        const int jq = tid % nq, part = tid / nq;
        const int pos_per = (p3 + parts - 1) / parts;
        f4 acc4 = {0.f, 0.f, 0.f, 0.f};
        for (int pos = part * pos_per; pos < (part + 1) * pos_per && pos < p3; ++pos) {
            const float *zin = A3.p + A3.at(pos / ow2, pos % ow2);
            const f4 *w3 = reinterpret_cast<const f4 *>(wp + a.w_fc3 + (size_t)pos * F2 * a.fch) + jq;
#pragma unroll 8
            for (int f = 0; f < F2; ++f) {
                const f4 w = w3[(size_t)f * nq];
                const float z = zin[f];
                acc4.x = __builtin_fmaf(z, w.x, acc4.x);
                acc4.y = __builtin_fmaf(z, w.y, acc4.y);
                acc4.z = __builtin_fmaf(z, w.z, acc4.z);
                acc4.w = __builtin_fmaf(z, w.w, acc4.w);
            }
        }
        f4 *part_sums = reinterpret_cast<f4 *>(ybuf.p);       // the image buffer is free until the next launch
        part_sums[part * nq + jq] = acc4;
        __syncthreads();
        for (int j = tid; j < a.fch; j += CT) {
            const float *ps = reinterpret_cast<const float *>(part_sums);
            float acc3 = ps[j];
            for (int q = 1; q < parts; ++q) acc3 = acc3 + ps[q * a.fch + j];
            const float pre = acc3 + ctx[a.c_zu3 + j];
            A4[j] = (pre > 0.f ? pre : 0.f) * ctx[a.c_gate[4] + j];
        }
    }
    __syncthreads();
    lap(5);
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
    lap(6);
    // P7: delta_3 = gate_4 * w_4 * [z_3 > 0]
    for (int j = tid; j < a.fch; j += CT) {
        const float gw = ctx[a.c_gate[4] + j] * wp[a.w_fc4 + j];
        A4[j] = A4[j] > 0.f ? gw : 0.f;
    }
    __syncthreads();
    lap(7);
    // P8: delta_2 = gate_3 * (W_3 delta_3) * [z_2 > 0]   (four neighbouring outputs per thread from the
    //     transposed copy: 16-B coalesced weight reads, no cross-lane reduction)
    for (int k4 = tid; k4 < a.flat / 4; k4 += CT) {
        const f4 *w3t = reinterpret_cast<const f4 *>(wp + a.w_fc3t) + k4;
        f4 acc4 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll 8
        for (int j = 0; j < a.fch; ++j) {
            const f4 w = w3t[(size_t)j * (a.flat / 4)];
            const float d = A4[j];
            acc4.x = __builtin_fmaf(w.x, d, acc4.x);
            acc4.y = __builtin_fmaf(w.y, d, acc4.y);
            acc4.z = __builtin_fmaf(w.z, d, acc4.z);
            acc4.w = __builtin_fmaf(w.w, d, acc4.w);
        }
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int k = 4 * k4 + s4;
            const float dz = ctx[a.c_gate[3] + k] * acc4[s4];
            const int pos = k / F2, f = k - pos * F2;
            float *zp = A3.p + A3.at(pos / ow2, pos % ow2) + f;
            *zp = *zp > 0.f ? dz : 0.f;
        }
    }
    __syncthreads();
    lap(8);
    // P9: delta_1 = gate_2 * convT(delta_2; Wzu_2) * [z_1 > 0] ; d y_red_2 = yu_2 * convT(delta_2; Wyu_2)
    {
        const int ch = tid % F1, slot = tid / F1, nslots = CT / F1;
        const int nb = block_positions(slot, nslots, p2, ow1, py, px);
        clear(acc);
        conv_block<true, 4>(A3, F2, wp + a.w_zut[2], F1, ch, a.K[2], a.S[2], a.P[2], py, px, acc);
        float dz[NBMAX];
#pragma unroll
        for (int i = 0; i < NBMAX; ++i) dz[i] = ctx[a.c_gate[2] + (i < nb ? (py[i] * ow1 + px[i]) * F1 + ch : 0)] * acc[i];
#pragma unroll
        for (int i = 0; i < NBMAX; ++i)
            if (i < nb) {
                float *zp = A2.p + A2.at(py[i], px[i]) + ch;
                *zp = *zp > 0.f ? dz[i] : 0.f;
            }
    }
    if (tid < p2) {
        block_positions(tid, p2, p2, ow1, py, px);
        clear(acc);
        conv_block<true, 1>(A3, F2, wp + a.w_yu[2], 1, 0, a.K[2], a.S[2], a.P[2], py, px, acc);
        dyr2.p[dyr2.at(py[0], px[0])] = ctx[a.c_yu[2] + tid] * acc[0];
    }
    __syncthreads();
    lap(9);
    // P10: delta_0 ; d y_red_1 = yu_1 * convT(delta_1; Wyu_1) + convT(d y_red_2; Wyr_1)
    {
        const int ch = tid % F0, slot = tid / F0, nslots = CT / F0;
        const int nb = block_positions(slot, nslots, p1, ow0, py, px);
        clear(acc);
        conv_block<true, 8>(A2, F1, wp + a.w_zut[1], F0, ch, a.K[1], a.S[1], a.P[1], py, px, acc);
#pragma unroll
        for (int i = 0; i < NBMAX; ++i)
            if (i < nb) {
                const float dz = ctx[a.c_gate[1] + (py[i] * ow0 + px[i]) * F0 + ch] * acc[i];
                float *zp = A1.p + A1.at(py[i], px[i]) + ch;
                *zp = *zp > 0.f ? dz : 0.f;
            }
    }
    if (tid < p1) {
        block_positions(tid, p1, p1, ow0, py, px);
        clear(acc);
        clear(acc2);
        conv_block<true, 1>(A2, F1, wp + a.w_yu[1], 1, 0, a.K[1], a.S[1], a.P[1], py, px, acc);
        conv_block<true, 1>(dyr2, 1, wp + a.w_yr[1], 1, 0, a.K[1], a.S[1], a.P[1], py, px, acc2);
        const float t1 = ctx[a.c_yu[1] + tid] * acc[0];
        dyr1.p[dyr1.at(py[0], px[0])] = t1 + acc2[0];
    }
    __syncthreads();
    lap(10);
    // P11: dE/dy = yu_0 * convT(delta_0; Wyu_0) + convT(d y_red_1; Wyr_0)
    {
        const int nb = block_positions(tid, CT, n, W, py, px);
        clear(acc);
        clear(acc2);
        conv_block<true, 4>(A1, F0, wp + a.w_yu[0], 1, 0, a.K[0], a.S[0], a.P[0], py, px, acc);
        conv_block<true, 4>(dyr1, 1, wp + a.w_yr[0], 1, 0, a.K[0], a.S[0], a.P[0], py, px, acc2);
#pragma unroll
        for (int i = 0; i < NBMAX; ++i)
            if (i < nb) {
                const int j = py[i] * W + px[i];
                const float t0 = ctx[a.c_yu[0] + j] * acc[i];
                a.g[(size_t)u * n + j] = t0 + acc2[i];
            }
    }
    lap(11);
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
    // what the blocked kernel is specialised for (true for the reference's completion network): output channel
    // counts that divide the workgroup, at most NBMAX positions per thread, kernel sizes that are multiples of
    // the stride, borders that cover the padding, and thread maps whose positions share (y+P) mod S
    for (int l = 0; l < 3; ++l)
        if (CT % a.F[l] || a.K[l] % a.S[l] || a.P[l] > (l == 0 ? PADI : PADM)) return ICNN_BE_ELIMIT;
    if (p1 > 8 * (CT / a.F[0]) || p2 > 4 * (CT / a.F[1]) || p3 > 4 * (CT / a.F[2]) || a.n > 4 * CT || p1 > CT)
        return ICNN_BE_ELIMIT;                            // blocking factors of the phases of conv_fg_kernel
    if (a.fch % 4 || CT % (a.fch / 4) || a.flat % 4 || (CT / (a.fch / 4)) * a.fch > (a.H + 2 * PADI) * (a.W + 2 * PADI))
        return ICNN_BE_ELIMIT;                            // P5 / P8 quads and the P5 partial-sum scratch
    if ((CT / a.F[0]) % a.ow[0] || ((CT / a.F[0]) / a.ow[0]) % a.S[1] || (CT / a.F[1]) % a.ow[1] ||
        CT % a.W || (CT / a.W) % a.S[0])
        return ICNN_BE_ELIMIT;
    const int simg = (a.H + 2 * PADI) * (a.W + 2 * PADI), sm1 = (a.oh[0] + 2) * (a.ow[0] + 2),
              sm2 = (a.oh[1] + 2) * (a.ow[1] + 2), sm3 = (a.oh[2] + 2) * (a.ow[2] + 2);
    const int floats = 2 * simg + 3 * sm1 + 3 * sm2 + sm1 * a.F[0] + sm2 * a.F[1] + sm3 * a.F[2] + a.fch + 16;
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

static long long *g_conv_prof = nullptr;
void set_conv_profile_buffer(long long *buf) { g_conv_prof = buf; }

hipError_t launch_conv_fg(const icnn_be_conv_model &m, const float *ctx, const double *y, int batch, float *f,
                          float *g, const int *skip, hipStream_t stream) {
    ConvLayout L{};
    if (conv_layout(m, L) != 0) return hipErrorInvalidValue;
    ConvArgs a = L.a;
    a.ctx = ctx; a.y = y; a.f = f; a.g = g; a.skip = skip; a.batch = batch;
    a.prof = g_conv_prof;
    if (hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(conv_fg_kernel), L.lds_bytes); e != hipSuccess) return e;
    hipLaunchKernelGGL(conv_fg_kernel, dim3(batch), dim3(CT), L.lds_bytes, stream, a);
    return hipGetLastError();
}

}  // namespace icnn_be
