/*
 * CPU oracle (TEST INFRASTRUCTURE ONLY): float32 energy and y-gradient of the fully-connected
 * PICNN, evaluated in the accumulation order of the MI355X kernel.
 *
 * Restates the layer algebra of the reference model
 *   multi-label-cls/icnn_ebundle.py:349-388 (Model.f, z-path) and :146 (tf.gradients(E_, y_)),
 *   RL/src/icnn.py:356-404 (negQ) with the action-box wrapper :148-158,
 * exactly like oracle/picnn_oracle.py does, but float32 dot products are order dependent and
 * TensorFlow's own summation order is unknowable (third-party, absent).  This file fixes the order
 * to the one v_mfma_f32_16x16x4_f32 applies -- a k-ordered chain of fused multiply-adds, with k
 * running over 16-blocks as kk = 16*kb + 4*q + s for s = 0..3 (instruction), q = 0..3 (k inside
 * the instruction) -- so that the HIP path can be compared with the oracle BIT FOR BIT and any
 * remaining difference in y* is attributable to the solver, not to float32 rounding noise.
 * Parity status of the network itself: unpinned (see oracle/picnn_oracle.py).
 *
 * Build: make -C oracle   (gcc -O2 -mfma -ffp-contract=off; fmaf() is the exact fused op)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#define MAXL 8
static int pad16(int v) { return (v + 15) & ~15; }

/* acc += a[0..K) . W[., col] in MFMA order; W is [K][N] row-major (transpose = 0) or [N][K] (1) */
static float chain(float acc, const float *a, int K, const float *W, int N, int col, int transpose) {
    const int KB = pad16(K) / 16;
    for (int kb = 0; kb < KB; ++kb)
        for (int s = 0; s < 4; ++s)
            for (int q = 0; q < 4; ++q) {
                const int kk = kb * 16 + 4 * q + s;
                if (kk < K) {
                    const float w = transpose ? W[(size_t)col * K + kk] : W[(size_t)kk * N + col];
                    acc = fmaf(a[kk], w, acc);
                }
                /* padded k: fmaf(0, 0, acc) == acc exactly */
            }
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

/* width[0..L]: s_0..s_L with s_L == 1; ctx row: per layer yu_i[n] | zu_i[s_i] | gate_i[s_{i-1}] (i>0) */
void picnn_chain_fg(int B, int n, int n_layers, const int *width, float alpha, int action_box,
                    const float *ctx, int C, const float *const *w_yu, const float *const *w_zu,
                    const double *y, float *E, float *g) {
    const int L = n_layers - 1;
    int yu_off[MAXL], zu_off[MAXL], gate_off[MAXL], o = 0, wmax = n;
    for (int i = 0; i <= L; ++i) {
        yu_off[i] = o; o += n;
        zu_off[i] = o; o += width[i];
        gate_off[i] = -1;
        if (i > 0) { gate_off[i] = o; o += width[i - 1]; }
        if (width[i] > wmax) wmax = width[i];
    }
    (void)C;
#pragma omp parallel for schedule(static)
    for (int u = 0; u < B; ++u) {
        const float *c = ctx + (size_t)u * C;
        float *y32 = (float *)malloc(sizeof(float) * (n + 2 * wmax * (L + 1) + 4 * wmax));
        float *abuf = y32 + n;                 /* y * yu_i, later the dE/dy accumulator */
        float *zb[MAXL];                       /* z_i * gate_{i+1}, later delta_i */
        for (int i = 0; i < L; ++i) zb[i] = abuf + wmax + (size_t)i * wmax;
        for (int j = 0; j < n; ++j) {
            const double yd = y[(size_t)u * n + j];
            y32[j] = action_box ? (float)(2.0 * yd - 1.0) : (float)yd;
        }
        /* forward */
        for (int i = 0; i < L; ++i) {
            const int wi = width[i];
            for (int j = 0; j < n; ++j) abuf[j] = y32[j] * c[yu_off[i] + j];
            for (int col = 0; col < wi; ++col) {
                float acc = chain(0.0f, abuf, n, w_yu[i], wi, col, 0);
                if (i > 0) acc = chain(acc, zb[i - 1], width[i - 1], w_zu[i], wi, col, 0);
                const float p = acc + c[zu_off[i] + col];
                const float z = p > 0.0f ? p : alpha * p;
                zb[i][col] = z * c[gate_off[i + 1] + col];
            }
        }
        /* final scalar layer: lane-strided partial sums, then the wave butterfly */
        {
            const float *wy = w_yu[L], *wz = w_zu[L];
            const int wl = width[L - 1];
            float part[64];
            for (int l = 0; l < 64; ++l) {
                float p = 0.0f;
                for (int j = l; j < wl; j += 64) p = fmaf(zb[L - 1][j], wz[j], p);
                for (int j = l; j < n; j += 64) { const float t = y32[j] * c[yu_off[L] + j]; p = fmaf(t, wy[j], p); }
                part[l] = p;
            }
            E[u] = wave_sum64(part) + c[zu_off[L]];
            for (int j = 0; j < wl; ++j) {
                const float gw = c[gate_off[L] + j] * wz[j];
                zb[L - 1][j] = gw * (zb[L - 1][j] > 0.0f ? 1.0f : alpha);
            }
            for (int j = 0; j < n; ++j) abuf[j] = c[yu_off[L] + j] * wy[j];
        }
        /* backward */
        for (int i = L - 1; i >= 0; --i) {
            const int wi = width[i];
            for (int col = 0; col < n; ++col) {
                const float acc = chain(0.0f, zb[i], wi, w_yu[i], wi, col, 1);    /* delta_i . Wyu_i[col][:] */
                abuf[col] = fmaf(c[yu_off[i] + col], acc, abuf[col]);
            }
            if (i > 0) {
                const int wp = width[i - 1];
                float *nd = (float *)malloc(sizeof(float) * wp);
                for (int col = 0; col < wp; ++col) {
                    const float acc = chain(0.0f, zb[i], wi, w_zu[i], wi, col, 1);   /* delta_i . Wzu_i[col][:] */
                    const float ga = c[gate_off[i] + col] * acc;
                    nd[col] = ga * (zb[i - 1][col] > 0.0f ? 1.0f : alpha);
                }
                memcpy(zb[i - 1], nd, sizeof(float) * wp);
                free(nd);
            }
        }
        const float gs = action_box ? 2.0f : 1.0f;
        for (int j = 0; j < n; ++j) g[(size_t)u * n + j] = gs * abuf[j];
        free(y32);
    }
}


/*
 * x-only context rows in the order rows_context_from_obs (icnn_amd/csrc/be_picnn_fc_rows_dev.h) applies -- the RL
 * agent's act() path, observation -> action in one launch; layer algebra of RL/src/icnn.py:339-385 without BatchNorm
 * (multi-label-cls/icnn_ebundle.py:339-374).  Stage i: prev_i [K_i] times the column-wise concatenation
 * [ u{i}/W | z{i}_yu_u/W | z{i}_u/W | z{i}_zu_u/W ] ([K_i][ld_i] row-major, ld_i = columns rounded up to 4), per column
 * a chain acc = fma(prev[k], W[k][col], acc), k ascending, then + bias; hidden u layers and gates ReLU'd.
 */
void picnn_context_rows_chain(int B, int n_features, int n, int n_layers, const int *width, const float *const *w_stage,
                              const float *const *b_stage, const float *obs, float *ctx, int C) {
    const int L = n_layers - 1;
    int yu_off[MAXL], zu_off[MAXL], gate_off[MAXL], o = 0, wmax = n_features;
    for (int i = 0; i <= L; ++i) {
        yu_off[i] = o; o += n;
        zu_off[i] = o; o += width[i];
        gate_off[i] = -1;
        if (i > 0) { gate_off[i] = o; o += width[i - 1]; }
        if (width[i] > wmax) wmax = width[i];
    }
    if (o != C) abort();
    float *prev = (float *)malloc(sizeof(float) * wmax), *next = (float *)malloc(sizeof(float) * wmax);
    for (int u = 0; u < B; ++u) {
        float *row = ctx + (size_t)u * C;
        int K = n_features;
        memcpy(prev, obs + (size_t)u * n_features, sizeof(float) * n_features);
        for (int i = 0; i <= L; ++i) {
            const int wu = i < L ? width[i] : 0, wg = i > 0 ? width[i - 1] : 0;
            const int cols = wu + n + width[i] + wg, ld = (cols + 3) & ~3;
            for (int col = 0; col < cols; ++col) {
                float acc = 0.f;
                for (int k = 0; k < K; ++k) acc = fmaf(prev[k], w_stage[i][(size_t)k * ld + col], acc);
                const float v = acc + b_stage[i][col];
                int c = col;
                if (c < wu) { next[c] = i < L - 1 ? (v > 0.f ? v : 0.f) : v; continue; }
                c -= wu;
                if (c < n) { row[yu_off[i] + c] = v; continue; }
                c -= n;
                if (c < width[i]) { row[zu_off[i] + c] = v; continue; }
                c -= width[i];
                row[gate_off[i] + c] = v > 0.f ? v : 0.f;
            }
            float *t = prev; prev = next; next = t;
            K = wu;
        }
    }
    free(prev); free(next);
}
