/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY): float32 energy and y-gradient of the convolutional PICNN of the image
 * completion experiment, evaluated in the accumulation order of the MI355X kernel (conv_fg_kernel,
 * icnn_amd/csrc/be_picnn_conv.hip).
 *
 * Restates the layer algebra of completion/icnn_ebundle.py:376-452 (Model.f, z-path: three conv layers with the
 * learned down-sampling chain y_red, fc 512, fc 1) and :118-121 (tf.gradients(E_, y_)) exactly like
 * oracle/picnn_conv_oracle.py does with torch autograd, but with every float32 sum in one fixed order -- the one the
 * kernels apply.  Every contraction with more than one output channel is an implicit GEMM on
 * v_mfma_f32_16x16x4_f32 there, i.e. per output a chain of fused multiply-adds over the K index in the order
 * kk = 16 kb + 4 q + s  for kb = 0.., s = 0..3 (instruction), q = 0..3 (inside the instruction), with
 *   convolutions          K index = (ky K + kx) Cin + r; a layer's single-channel yu term is a separate (ky, kx) chain
 *                         added afterwards, then zu
 *   fc 2048 -> 512        K index = flattened NHWC position, four chains over a quarter of K each added in order;
 *                         its transpose: one chain, K index = the 512 outputs
 *   stride-1 transpose    K index = (ky K + kx) F_out + f
 *   stride-2 transpose    per parity class of the position: K index = (valid tap, f), taps ky = ry, ry + S; kx likewise
 *   last transpose        "pixel shuffle": K index = (3 x 3 neighbouring cell, r) with zero weight where no tap reaches
 *   energy                a chain per lane over j = lane, lane + 64, .., the xor butterfly of a wave of 64, then zu_4
 * and the single-channel pieces (y_red chain, its transposes) are (ky, kx[, r]) chains on the VALU,
 * so that the HIP path can be compared with this file BIT FOR BIT.  float32 sums are order dependent and
 * TensorFlow's own order is unknowable (third-party, absent): parity status of the network itself is unpinned,
 * see oracle/picnn_conv_oracle.py.
 *
 * Build: make -C oracle   (gcc -O2 -mfma -ffp-contract=off; fmaf() is the exact fused op)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* acc = sum_k a[k] w[k] as the MFMA accumulates it: k-blocks of 16, kk = 16 kb + 4 q + s, s outer, q inner */
static float chain16(const float *a, const float *w, int K) {
    float acc = 0.f;
    const int KB = (K + 15) / 16;
    for (int kb = 0; kb < KB; ++kb)
        for (int s = 0; s < 4; ++s)
            for (int q = 0; q < 4; ++q) {
                const int kk = 16 * kb + 4 * q + s;
                if (kk < K) acc = fmaf(a[kk], w[kk], acc);
            }
    return acc;
}

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

/* im2col row of a forward convolution at (oy, ox): a[(ky K + kx) C + r] */
static void gather_fwd(const Map *in, int K, int S, int P, int oy, int ox, float *a) {
    for (int ky = 0; ky < K; ++ky)
        for (int kx = 0; kx < K; ++kx)
            for (int r = 0; r < in->c; ++r) a[(size_t)(ky * K + kx) * in->c + r] = at(in, oy * S + ky - P, ox * S + kx - P, r);
}

/* transposed convolution onto one channel as convt_group8 of the kernel: eight partial chains (channels c, c+8, ..,
 * taps outer), combined ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) */
static float convt_group8(const Map *dout, const float *Wt, int K, int S, int P, int iy, int ix) {
    float part[8];
    const int R = dout->c;
    for (int c = 0; c < 8; ++c) {
        float acc = 0.f;
        for (int ky = (iy + P) % S; ky < K; ky += S)
            for (int kx = (ix + P) % S; kx < K; kx += S)
                for (int m = 0; m < R / 8; ++m)
                    acc = fmaf(at(dout, (iy + P - ky) / S, (ix + P - kx) / S, c + 8 * m), Wt[(size_t)(ky * K + kx) * R + c + 8 * m], acc);
        part[c] = acc;
    }
    return ((part[0] + part[1]) + (part[2] + part[3])) + ((part[4] + part[5]) + (part[6] + part[7]));
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
        int kmax = flat > 9 * F[0] ? flat : 9 * F[0];
        for (int l = 1; l < 3; ++l) if (K[l] * K[l] * F[l - 1] > kmax) kmax = K[l] * K[l] * F[l - 1];
        for (int l = 1; l < 3; ++l) if (K[l] * K[l] * F[l] > kmax) kmax = K[l] * K[l] * F[l];
        float *av = (float *)calloc(kmax, 4), *wv = (float *)calloc(kmax, 4), *wm = (float *)calloc((size_t)kmax * F[0], 4);
        const Map Myv = {H, W, 1, yv}, Ma0 = {H, W, 1, a0}, Myr1 = {oh[0], ow[0], 1, yr1}, May1 = {oh[0], ow[0], 1, ay1};
        const Map Mdyr1 = {oh[0], ow[0], 1, dyr1}, May2 = {oh[1], ow[1], 1, ay2}, Mdyr2 = {oh[1], ow[1], 1, dyr2};
        const Map MA1 = {oh[0], ow[0], F[0], A1}, MA2 = {oh[1], ow[1], F[1], A2}, MA3 = {oh[2], ow[2], F[2], A3};
        /* P0 */
        for (int j = 0; j < n; ++j) { yv[j] = (float)y[(size_t)u * n + j]; a0[j] = yv[j] * cx[c_yu[0] + j]; }
        /* P1: y_red_1, z_0 -> A1 = z_0 * gate_1 */
        for (int p = 0; p < p1; ++p) yr1[p] = conv_at(&Myv, w_yr[0], 1, 0, K[0], S[0], P[0], p / ow[0], p % ow[0]) + b_yr[0][0];
        for (int p = 0; p < p1; ++p) {
            gather_fwd(&Ma0, K[0], S[0], P[0], p / ow[0], p % ow[0], av);
            for (int ch = 0; ch < F[0]; ++ch) {
                const int e = p * F[0] + ch, KK = K[0] * K[0];
                for (int k = 0; k < KK; ++k) wv[k] = w_yu[0][(size_t)k * F[0] + ch];
                const float pre = chain16(av, wv, KK) + cx[c_zu[0] + e];
                A1[e] = (pre > 0.f ? pre : 0.f) * cx[c_gate[1] + e];
            }
        }
        /* P2 */
        for (int p = 0; p < p1; ++p) ay1[p] = yr1[p] * cx[c_yu[1] + p];
        for (int p = 0; p < p2; ++p) yr2[p] = conv_at(&Myr1, w_yr[1], 1, 0, K[1], S[1], P[1], p / ow[1], p % ow[1]) + b_yr[1][0];
        /* P3: z_1 -> A2 */
        for (int p = 0; p < p2; ++p) {
            gather_fwd(&MA1, K[1], S[1], P[1], p / ow[1], p % ow[1], av);
            for (int ch = 0; ch < F[1]; ++ch) {
                const int e = p * F[1] + ch, KK = K[1] * K[1] * F[0];
                for (int k = 0; k < KK; ++k) wv[k] = w_zu[1][(size_t)k * F[1] + ch];
                const float acc = chain16(av, wv, KK);
                const float acc2 = conv_at(&May1, w_yu[1], F[1], ch, K[1], S[1], P[1], p / ow[1], p % ow[1]);
                float pre = acc + acc2;
                pre = pre + cx[c_zu[1] + e];
                A2[e] = (pre > 0.f ? pre : 0.f) * cx[c_gate[2] + e];
            }
        }
        for (int p = 0; p < p2; ++p) ay2[p] = yr2[p] * cx[c_yu[2] + p];
        /* P4: z_2 -> A3 */
        for (int p = 0; p < p3; ++p) {
            gather_fwd(&MA2, K[2], S[2], P[2], p / ow[2], p % ow[2], av);
            for (int ch = 0; ch < F[2]; ++ch) {
                const int e = p * F[2] + ch, KK = K[2] * K[2] * F[1];
                for (int k = 0; k < KK; ++k) wv[k] = w_zu[2][(size_t)k * F[2] + ch];
                const float acc = chain16(av, wv, KK);
                const float acc2 = conv_at(&May2, w_yu[2], F[2], ch, K[2], S[2], P[2], p / ow[2], p % ow[2]);
                float pre = acc + acc2;
                pre = pre + cx[c_zu[2] + e];
                A3[e] = (pre > 0.f ? pre : 0.f) * cx[c_gate[3] + e];
            }
        }
        /* P5: z_3 -> A4 = z_3 * gate_4: four chains over a quarter of the flattened positions each, ((p0+p1)+p2)+p3 */
        for (int j = 0; j < fch; ++j) {
            for (int k = 0; k < flat; ++k) wv[k] = w_fc3[(size_t)k * fch + j];
            const int qk = flat / 4;
            float tot = chain16(A3, wv, qk) + chain16(A3 + qk, wv + qk, qk);
            tot = tot + chain16(A3 + 2 * qk, wv + 2 * qk, qk);
            tot = tot + chain16(A3 + 3 * qk, wv + 3 * qk, qk);
            const float pre = tot + cx[c_zu3 + j];
            A4[j] = (pre > 0.f ? pre : 0.f) * cx[c_gate[4] + j];
        }
        /* P6: energy: lane chains over j = lane, lane + 64, .., butterfly of the wave */
        {
            float lanes[64];
            for (int l = 0; l < 64; ++l) {
                float part = 0.f;
                for (int j = l; j < fch; j += 64) part = fmaf(A4[j], w_fc4[j], part);
                lanes[l] = part;
            }
            E[u] = wave_sum64(lanes) + cx[c_zu4];
        }
        /* P7: delta_3 */
        for (int j = 0; j < fch; ++j) A4[j] = A4[j] > 0.f ? cx[c_gate[4] + j] * w_fc4[j] : 0.f;
        /* P8: delta_2 (in place of A3) */
        for (int k = 0; k < flat; ++k) {
            const float dz = cx[c_gate[3] + k] * chain16(A4, w_fc3 + (size_t)k * fch, fch);
            A3[k] = A3[k] > 0.f ? dz : 0.f;
        }
        /* P9: delta_1 ; d y_red_2 */
        for (int p = 0; p < p2; ++p) {
            {   /* a[(ky K + kx) F2 + f] = delta_2[iy + P - ky][ix + P - kx][f] */
                const int iy = p / ow[1], ix = p % ow[1], KK = K[2] * K[2] * F[2];
                for (int ky = 0; ky < K[2]; ++ky)
                    for (int kx = 0; kx < K[2]; ++kx)
                        for (int f = 0; f < F[2]; ++f) av[(size_t)(ky * K[2] + kx) * F[2] + f] = at(&MA3, iy + P[2] - ky, ix + P[2] - kx, f);
                for (int ch = 0; ch < F[1]; ++ch) {
                    const int e = p * F[1] + ch;
                    for (int k = 0; k < KK; ++k) wv[k] = w_zut[2][(size_t)k * F[1] + ch];
                    const float dz = cx[c_gate[2] + e] * chain16(av, wv, KK);
                    T2[e] = A2[e] > 0.f ? dz : 0.f;
                }
            }
            dyr2[p] = cx[c_yu[2] + p] * convt_group8(&MA3, w_yu[2], K[2], S[2], P[2], p / ow[1], p % ow[1]);
        }
        memcpy(A2, T2, sizeof(float) * p2 * F[1]);
        /* P10: delta_0 ; d y_red_1 */
        for (int p = 0; p < p1; ++p) {
            {   /* valid taps ky = ry + S ti, kx = rx + S tj in (ti, tj) order: a[(ti TS + tj) F1 + f] */
                const int iy = p / ow[0], ix = p % ow[0], TS = K[1] / S[1], KK = TS * TS * F[1];
                const int ry = (iy + P[1]) % S[1], rx = (ix + P[1]) % S[1];
                for (int ti = 0; ti < TS; ++ti)
                    for (int tj = 0; tj < TS; ++tj) {
                        const int ky = ry + S[1] * ti, kx = rx + S[1] * tj;
                        for (int f = 0; f < F[1]; ++f) {
                            av[(size_t)(ti * TS + tj) * F[1] + f] = at(&MA2, (iy + P[1] - ky) / S[1], (ix + P[1] - kx) / S[1], f);
                            for (int ch = 0; ch < F[0]; ++ch)
                                wm[((size_t)(ti * TS + tj) * F[1] + f) * F[0] + ch] = w_zut[1][((size_t)(ky * K[1] + kx) * F[1] + f) * F[0] + ch];
                        }
                    }
                for (int ch = 0; ch < F[0]; ++ch) {
                    const int e = p * F[0] + ch;
                    for (int k = 0; k < KK; ++k) wv[k] = wm[(size_t)k * F[0] + ch];
                    const float dz = cx[c_gate[1] + e] * chain16(av, wv, KK);
                    T1[e] = A1[e] > 0.f ? dz : 0.f;
                }
            }
            const float t1 = cx[c_yu[1] + p] * convt_group8(&MA2, w_yu[1], K[1], S[1], P[1], p / ow[0], p % ow[0]);
            dyr1[p] = t1 + convt_at(&Mdyr2, w_yr[1], 1, 0, K[1], S[1], P[1], p / ow[0], p % ow[0]);
        }
        memcpy(A1, T1, sizeof(float) * p1 * F[0]);
        /* P11: dE/dy */
        for (int j = 0; j < n; ++j) {
            /* K index = ((ny, nx) neighbour cell of the pixel's cell, r); weight zero where no tap reaches */
            const int iy = j / W, ix = j % W, cy = iy / S[0], cx_ = ix / S[0], pa = iy % S[0], pb = ix % S[0], KK = 9 * F[0];
            for (int ny = 0; ny < 3; ++ny)
                for (int nx = 0; nx < 3; ++nx) {
                    const int ky = pa + P[0] - S[0] * (ny - 1), kx = pb + P[0] - S[0] * (nx - 1);
                    const int ok = ky >= 0 && ky < K[0] && kx >= 0 && kx < K[0];
                    for (int r = 0; r < F[0]; ++r) {
                        av[(size_t)(ny * 3 + nx) * F[0] + r] = at(&MA1, cy + ny - 1, cx_ + nx - 1, r);
                        wv[(size_t)(ny * 3 + nx) * F[0] + r] = ok ? w_yu[0][(size_t)(ky * K[0] + kx) * F[0] + r] : 0.f;
                    }
                }
            const float t0 = cx[c_yu[0] + j] * chain16(av, wv, KK);
            g[(size_t)u * n + j] = t0 + convt_at(&Mdyr1, w_yr[0], 1, 0, K[0], S[0], P[0], j / W, j % W);
        }
        free(yv); free(a0); free(yr1); free(ay1); free(dyr1); free(yr2); free(ay2); free(dyr2);
        free(A1); free(A2); free(A3); free(A4); free(T1); free(T2); free(av); free(wv); free(wm);
    }
    free(w_zut[1]); free(w_zut[2]);
}
