/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY): float32 energy and y-gradient of the convolutional PICNN of the image
 * completion experiment, evaluated in the accumulation order of the MI355X kernel (conv_fg_kernel,
 * icnn_amd/csrc/be_picnn_conv.hip).
 *
 * Restates the layer algebra of completion/icnn_ebundle.py:376-452 (Model.f, z-path: three conv layers with the
 * learned down-sampling chain y_red, fc 512, fc 1) and :118-121 (tf.gradients(E_, y_)) exactly like
 * oracle/picnn_conv_oracle.py does with torch autograd, but with every float32 sum in one fixed order:
 *   convolutions        per output a chain of fused multiply-adds over (ky, kx, input channel), in that nesting;
 *                       the zu_proj and the yu contribution of a layer are two chains added afterwards, then zu
 *   transposed convs    per input position the taps ky = (y + P) mod S, +S, ...; kx likewise; channel innermost
 *   fc 2048 -> 512      four partial chains over eight positions each (position-major, channel inner), added in
 *                       order, then zu;  its transpose: one chain over the 512 outputs
 *   energy              one product per thread of the 512-thread workgroup, xor-butterfly sum inside each wave of 64
 *                       (offsets 32 .. 1), the eight wave sums added in order, then zu_4
 * so that the HIP kernel can be compared with this file BIT FOR BIT.  float32 sums are order dependent and
 * TensorFlow's own order is unknowable (third-party, absent): parity status of the network itself is unpinned,
 * see oracle/picnn_conv_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2 -mfma -ffp-contract=off; fmaf() is the exact fused op)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define CT 512          /* threads of the kernel's workgroup: fixes the partial-sum structure of fc3 and the energy */

typedef struct {
    int h, w, c;        /* dense [h][w][c] map, reads outside are zero */
    const float *p;
} Map;

static float at(const Map *m, int y, int x, int r) {
    if (y < 0 || y >= m->h || x < 0 || x >= m->w) return 0.f;
    return m->p[((size_t)y * m->w + x) * m->c + r];
}

/* forward: sum_{ky,kx,r} in[oy*S+ky-P][ox*S+kx-P][r] * W[((ky*K+kx)*R + r)*Cout + ch] */
static float conv_at(const Map *in, const float *W, int Cout, int ch, int K, int S, int P, int oy, int ox) {
    float acc = 0.f;
    for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx)
            for (int r = 0; r < in->c; ++r)
                acc = fmaf(at(in, oy * S + ky - P, ox * S + kx - P, r), W[((size_t)(ky * K + kx) * in->c + r) * Cout + ch], acc);
    return acc;
}

/* transposed: sum over ky = (iy+P) mod S, +S, ..; kx likewise; r:  dout[(iy+P-ky)/S][(ix+P-kx)/S][r] * Wt[((ky*K+kx)*R + r)*Cout + ch] */
static float convt_at(const Map *dout, const float *Wt, int Cout, int ch, int K, int S, int P, int iy, int ix) {
    float acc = 0.f;
    for (int ky = (iy + P) % S; ky < K; ky += S)
        for (int kx = (ix + P) % S; kx < K; kx += S)
            for (int r = 0; r < dout->c; ++r)
                acc = fmaf(at(dout, (iy + P - ky) / S, (ix + P - kx) / S, r), Wt[((size_t)(ky * K + kx) * dout->c + r) * Cout + ch], acc);
    return acc;
}

static float wave_sum64(float *p) {   /* xor butterfly 32,16,8,4,2,1 as wave_sum_f in be_common.h */
    float t[64];
    for (int o = 32; o > 0; o >>= 1) {
        for (int l = 0; l < 64; ++l) t[l] = p[l] + p[l ^ o];
        memcpy(p, t, sizeof(t));
    }
    return p[0];
}

/*
 * ctx row: yu0[n] zu0 | gate1 yu1 zu1 | gate2 yu2 zu2 | gate3[flat] zu3[fch] | gate4[fch] zu4[1]   (include/icnn_be.h)
 * weights in tflearn layout: w_yu[l] [k][k][1][F_l]; w_yr[l], b_yr[l] (l = 0, 1) [k][k][1][1], [1];
 * w_zu[l] (l = 1, 2) [k][k][F_{l-1}][F_l]; w_fc3 [flat][fch]; w_fc4 [fch][1]
 */
void picnn_conv_chain_fg(int B, int H, int W, const int *F, const int *K, const int *S, int fch, const float *ctx, int C,
                         const float *const *w_yu, const float *const *w_yr, const float *const *b_yr,
                         const float *const *w_zu, const float *w_fc3, const float *w_fc4, const double *y, float *E,
                         float *g) {
    int oh[3], ow[3], P[3], h = H, w = W;
    for (int l = 0; l < 3; ++l) {
        oh[l] = (h + S[l] - 1) / S[l]; ow[l] = (w + S[l] - 1) / S[l];
        P[l] = ((oh[l] - 1) * S[l] + K[l] - h) / 2;
        h = oh[l]; w = ow[l];
    }
    const int n = H * W, p1 = oh[0] * ow[0], p2 = oh[1] * ow[1], p3 = oh[2] * ow[2], flat = p3 * F[2];
    int o = 0, c_yu[3], c_zu[3], c_gate[5], cin = 1;
    h = H; w = W;
    for (int l = 0; l < 3; ++l) {
        if (l > 0) { c_gate[l] = o; o += h * w * cin; }
        c_yu[l] = o; o += h * w;
        c_zu[l] = o; o += oh[l] * ow[l] * F[l];
        h = oh[l]; w = ow[l]; cin = F[l];
    }
    c_gate[3] = o; o += flat;
    const int c_zu3 = o; o += fch;
    c_gate[4] = o; o += fch;
    const int c_zu4 = o; o += 1;
    if (o != C) abort();
    /* transposed copies of the zu weights, [tap][f_out][c_in], as the kernel's pack holds them */
    float *w_zut[3] = {0, 0, 0};
    cin = F[0];
    for (int l = 1; l < 3; ++l) {
        w_zut[l] = (float *)malloc(sizeof(float) * K[l] * K[l] * cin * F[l]);
        for (int t = 0; t < K[l] * K[l]; ++t)
            for (int c = 0; c < cin; ++c)
                for (int f = 0; f < F[l]; ++f) w_zut[l][((size_t)t * F[l] + f) * cin + c] = w_zu[l][((size_t)t * cin + c) * F[l] + f];
        cin = F[l];
    }
#pragma omp parallel for schedule(dynamic, 1)
    for (int u = 0; u < B; ++u) {
        const float *cx = ctx + (size_t)u * C;
        float *yv = (float *)calloc(n, 4), *a0 = (float *)calloc(n, 4);
        float *yr1 = (float *)calloc(p1, 4), *ay1 = (float *)calloc(p1, 4), *dyr1 = (float *)calloc(p1, 4);
        float *yr2 = (float *)calloc(p2, 4), *ay2 = (float *)calloc(p2, 4), *dyr2 = (float *)calloc(p2, 4);
        float *A1 = (float *)calloc((size_t)p1 * F[0], 4), *A2 = (float *)calloc((size_t)p2 * F[1], 4);
        float *A3 = (float *)calloc((size_t)p3 * F[2], 4), *A4 = (float *)calloc(fch, 4);
        float *T1 = (float *)calloc((size_t)p1 * F[0], 4), *T2 = (float *)calloc((size_t)p2 * F[1], 4);
        const Map Myv = {H, W, 1, yv}, Ma0 = {H, W, 1, a0}, Myr1 = {oh[0], ow[0], 1, yr1}, May1 = {oh[0], ow[0], 1, ay1};
        const Map Mdyr1 = {oh[0], ow[0], 1, dyr1}, May2 = {oh[1], ow[1], 1, ay2}, Mdyr2 = {oh[1], ow[1], 1, dyr2};
        const Map MA1 = {oh[0], ow[0], F[0], A1}, MA2 = {oh[1], ow[1], F[1], A2}, MA3 = {oh[2], ow[2], F[2], A3};
        /* P0 */
        for (int j = 0; j < n; ++j) { yv[j] = (float)y[(size_t)u * n + j]; a0[j] = yv[j] * cx[c_yu[0] + j]; }
        /* P1: y_red_1, z_0 -> A1 = z_0 * gate_1 */
        for (int p = 0; p < p1; ++p) yr1[p] = conv_at(&Myv, w_yr[0], 1, 0, K[0], S[0], P[0], p / ow[0], p % ow[0]) + b_yr[0][0];
        for (int p = 0; p < p1; ++p)
            for (int ch = 0; ch < F[0]; ++ch) {
                const int e = p * F[0] + ch;
                const float pre = conv_at(&Ma0, w_yu[0], F[0], ch, K[0], S[0], P[0], p / ow[0], p % ow[0]) + cx[c_zu[0] + e];
                A1[e] = (pre > 0.f ? pre : 0.f) * cx[c_gate[1] + e];
            }
        /* P2 */
        for (int p = 0; p < p1; ++p) ay1[p] = yr1[p] * cx[c_yu[1] + p];
        for (int p = 0; p < p2; ++p) yr2[p] = conv_at(&Myr1, w_yr[1], 1, 0, K[1], S[1], P[1], p / ow[1], p % ow[1]) + b_yr[1][0];
        /* P3: z_1 -> A2 */
        for (int p = 0; p < p2; ++p)
            for (int ch = 0; ch < F[1]; ++ch) {
                const int e = p * F[1] + ch;
                const float acc = conv_at(&MA1, w_zu[1], F[1], ch, K[1], S[1], P[1], p / ow[1], p % ow[1]);
                const float acc2 = conv_at(&May1, w_yu[1], F[1], ch, K[1], S[1], P[1], p / ow[1], p % ow[1]);
                float pre = acc + acc2;
                pre = pre + cx[c_zu[1] + e];
                A2[e] = (pre > 0.f ? pre : 0.f) * cx[c_gate[2] + e];
            }
        for (int p = 0; p < p2; ++p) ay2[p] = yr2[p] * cx[c_yu[2] + p];
        /* P4: z_2 -> A3 */
        for (int p = 0; p < p3; ++p)
            for (int ch = 0; ch < F[2]; ++ch) {
                const int e = p * F[2] + ch;
                const float acc = conv_at(&MA2, w_zu[2], F[2], ch, K[2], S[2], P[2], p / ow[2], p % ow[2]);
                const float acc2 = conv_at(&May2, w_yu[2], F[2], ch, K[2], S[2], P[2], p / ow[2], p % ow[2]);
                float pre = acc + acc2;
                pre = pre + cx[c_zu[2] + e];
                A3[e] = (pre > 0.f ? pre : 0.f) * cx[c_gate[3] + e];
            }
        /* P5: z_3 -> A4 = z_3 * gate_4 (partial chains over position ranges, added in order) */
        {
            const int nq = fch / 4, parts = CT / nq, pos_per = (p3 + parts - 1) / parts;
            for (int j = 0; j < fch; ++j) {
                float tot = 0.f;
                for (int part = 0; part < parts; ++part) {
                    float acc = 0.f;
                    for (int pos = part * pos_per; pos < (part + 1) * pos_per && pos < p3; ++pos)
                        for (int f = 0; f < F[2]; ++f) acc = fmaf(A3[pos * F[2] + f], w_fc3[((size_t)pos * F[2] + f) * fch + j], acc);
                    tot = part == 0 ? acc : tot + acc;
                }
                const float pre = tot + cx[c_zu3 + j];
                A4[j] = (pre > 0.f ? pre : 0.f) * cx[c_gate[4] + j];
            }
        }
        /* P6: energy */
        {
            float red[CT / 64], lanes[64], e = 0.f;
            for (int wv = 0; wv < CT / 64; ++wv) {
                for (int l = 0; l < 64; ++l) {
                    float part = 0.f;
                    for (int j = wv * 64 + l; j < fch; j += CT) part = fmaf(A4[j], w_fc4[j], part);
                    lanes[l] = part;
                }
                red[wv] = wave_sum64(lanes);
            }
            for (int wv = 0; wv < CT / 64; ++wv) e += red[wv];
            E[u] = e + cx[c_zu4];
        }
        /* P7: delta_3 */
        for (int j = 0; j < fch; ++j) A4[j] = A4[j] > 0.f ? cx[c_gate[4] + j] * w_fc4[j] : 0.f;
        /* P8: delta_2 (in place of A3) */
        for (int k = 0; k < flat; ++k) {
            float acc = 0.f;
            for (int j = 0; j < fch; ++j) acc = fmaf(w_fc3[(size_t)k * fch + j], A4[j], acc);
            const float dz = cx[c_gate[3] + k] * acc;
            A3[k] = A3[k] > 0.f ? dz : 0.f;
        }
        /* P9: delta_1 ; d y_red_2 */
        for (int p = 0; p < p2; ++p) {
            for (int ch = 0; ch < F[1]; ++ch) {
                const int e = p * F[1] + ch;
                const float dz = cx[c_gate[2] + e] * convt_at(&MA3, w_zut[2], F[1], ch, K[2], S[2], P[2], p / ow[1], p % ow[1]);
                T2[e] = A2[e] > 0.f ? dz : 0.f;
            }
            dyr2[p] = cx[c_yu[2] + p] * convt_at(&MA3, w_yu[2], 1, 0, K[2], S[2], P[2], p / ow[1], p % ow[1]);
        }
        memcpy(A2, T2, sizeof(float) * p2 * F[1]);
        /* P10: delta_0 ; d y_red_1 */
        for (int p = 0; p < p1; ++p) {
            for (int ch = 0; ch < F[0]; ++ch) {
                const int e = p * F[0] + ch;
                const float dz = cx[c_gate[1] + e] * convt_at(&MA2, w_zut[1], F[0], ch, K[1], S[1], P[1], p / ow[0], p % ow[0]);
                T1[e] = A1[e] > 0.f ? dz : 0.f;
            }
            const float t1 = cx[c_yu[1] + p] * convt_at(&MA2, w_yu[1], 1, 0, K[1], S[1], P[1], p / ow[0], p % ow[0]);
            dyr1[p] = t1 + convt_at(&Mdyr2, w_yr[1], 1, 0, K[1], S[1], P[1], p / ow[0], p % ow[0]);
        }
        memcpy(A1, T1, sizeof(float) * p1 * F[0]);
        /* P11: dE/dy */
        for (int j = 0; j < n; ++j) {
            const float t0 = cx[c_yu[0] + j] * convt_at(&MA1, w_yu[0], 1, 0, K[0], S[0], P[0], j / W, j % W);
            g[(size_t)u * n + j] = t0 + convt_at(&Mdyr1, w_yr[0], 1, 0, K[0], S[0], P[0], j / W, j % W);
        }
        free(yv); free(a0); free(yr1); free(ay1); free(dyr1); free(yr2); free(ay2); free(dyr2);
        free(A1); free(A2); free(A3); free(A4); free(T1); free(T2);
    }
    free(w_zut[1]); free(w_zut[2]);
}
