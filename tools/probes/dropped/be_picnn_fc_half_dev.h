// FC-PICNN energy + y-gradient for a HALF tile: 8 samples per workgroup of 8 waves, on v_mfma_f32_4x4x1_16b_f32.
//
// v_mfma_f32_16x16x4_f32 has M = 16: a tile costs the same whether 4, 8 or 16 of its rows are samples, which is what
// makes anything finer than one 16-sample tile per workgroup pointless on that instruction.  The 16-block form
// (16 blocks of D[4][4] += A[4][1] B[1][4], K = 1; measured on gfx950, tools/probes/mfma_f32_4x4x1_probe.hip: lane 4b+i
// supplies A row i and B column i of block b, D[i][j] of block b lands in lane 4b+j register i; 8.4 cycles per
// instruction over independent accumulators = the same 64 flop/clk/SIMD) used with the SAME four samples in every block
// and 64 different output columns makes the cost of a GEMM proportional to the number of samples in groups of four.
// Two such 8-sample workgroups fit one CU (81 KB of LDS, 128 VGPRs each), unsynchronised: one's phase A overlaps the
// other's dual phase (be_fused.hip).
//
// Arithmetic: per output the same chain of fused multiply-adds as fc_fg_tile -- K = 1 per instruction, issued in the
// order kk = 16 kb + 4 q + s (s outer, q inner) the 16x16x4 form applies --, the same element-wise operations in the
// epilogues: bit-identical to fc_fg_tile, to the VALU rows path and to oracle/picnn_chain.c.  The B operands come from
// the SAME packed weights (16-column tiles, k-block major): lane l of column group g reads the fragment of tile
// 4 g + l/16, position (q, l%16) -- four 256-byte runs per load instruction.
#pragma once
#include "be_picnn_fc_dev.h"

namespace icnn_be {

namespace {

constexpr int HW = 8, HT = HW * 64, HR = 8;    // waves, threads, samples of a half-tile workgroup

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// acc[h][i] += sum_k A[4 h + i][k] W[k][64 g + lane]  (h = sample quad, i = register), k over the ceil(K / 16) k-blocks
// that hold weights.  A: LDS rows of pitch ld (pad columns zero); Wp: a 16x16x4-packed operand of NT column tiles.
// Q0 / Q1: which quads this wave computes (a narrow layer gives each quad of a column group to its own wave).
template <bool Q0, bool Q1>
__device__ __forceinline__ void half_gemm(f4 (&acc)[2], const float *A, int ld, const float *Wp, int K, int NT, int g) {
    const int lane = thread_id() & 63, i = lane & 3, nt = 4 * g + (lane >> 4), r16 = lane & 15;
    const bool valid = nt < NT;
    const f4 *bp = reinterpret_cast<const f4 *>(Wp) + (size_t)(valid ? nt : 0) * 64 + r16;
    const size_t kstride = (size_t)NT * 64;
    const int KB = (K + 15) / 16;
    const float *a0 = A + i * ld, *a1 = A + (4 + i) * ld;
    const f4 zero = {0.f, 0.f, 0.f, 0.f};
    f4 bn[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) bn[q] = valid ? bp[q * 16] : zero;
    for (int kb = 0; kb < KB; ++kb) {
        f4 b[4], x0[4], x1[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b[q] = bn[q];
            if (Q0) x0[q] = *reinterpret_cast<const f4 *>(a0 + 16 * kb + 4 * q);
            if (Q1) x1[q] = *reinterpret_cast<const f4 *>(a1 + 16 * kb + 4 * q);
        }
        const int nk = kb + 1 < KB ? kb + 1 : kb;            // the next k-block's fragments are requested before the MFMAs
#pragma unroll
        for (int q = 0; q < 4; ++q) bn[q] = valid ? bp[(size_t)nk * kstride + q * 16] : zero;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (Q0) acc[0] = mfma4(x0[q][s], b[q][s], acc[0]);
                if (Q1) acc[1] = mfma4(x1[q][s], b[q][s], acc[1]);
            }
    }
}
// mode 0: both quads; 1: quad 0 only; 2: quad 1 only
__device__ __forceinline__ void half_gemm_mode(int mode, f4 (&acc)[2], const float *A, int ld, const float *Wp, int K, int NT, int g) {
    if (mode == 0) half_gemm<true, true>(acc, A, ld, Wp, K, NT, g);
    else if (mode == 1) half_gemm<true, false>(acc, A, ld, Wp, K, NT, g);
    else half_gemm<false, true>(acc, A, ld, Wp, K, NT, g);
}

// Work units of a GEMM with NG column groups on HW waves: with at most HW / 2 groups every (group, quad) pair is a unit
// of its own (mode 1 / 2), otherwise a wave takes whole groups (mode 0).  unit -> (g, mode)
__device__ __forceinline__ int half_units(int NG) { return 2 * NG <= HW ? 2 * NG : NG; }
__device__ __forceinline__ void half_unit(int NG, int unit, int &g, int &mode) {
    if (2 * NG <= HW) { g = unit >> 1; mode = 1 + (unit & 1); }
    else { g = unit; mode = 0; }
}

// One half tile (HR samples, HT threads); LDS buffers laid out by fill_args(m, a, lds, HR).
template <typename ArgsT>
__device__ __forceinline__ void fc_fg_half(const ArgsT &a, int tile, float *lds) {
#pragma clang fp contract(off)
    const int tid = thread_id(), lane = tid & 63, wave = tid >> 6, qi = lane & 3;
    const int s0 = tile * HR;
    const int rows = min(HR, a.batch - s0);
    const int n = a.n, L = a.L, C = a.ctx_width, ldY = a.ldY;
    const int npad = pad16(n);
    float *ybuf = lds + a.ybuf_off;
    if (a.finished) {                    // nothing to do if every sample of the half tile has left the loop
        int live = 0;
        if (tid < rows) live = a.finished[s0 + tid] == 0;
        if (!__syncthreads_or(live)) return;
    }
    const float *ctx = a.ctx + (size_t)s0 * C;
    float *gbuf = lds + a.gbuf_off, *dl = lds + a.dl_off;
    // wave w prepares row w: pad columns, the network input and the y-operands of every layer (as fc_fg_tile)
    {
        auto zero_pad = [&](float *buf, int ld, int width) {
            const int w16 = pad16(width);
            for (int j = w16 + lane; j < ld; j += 64) buf[wave * ld + j] = 0.f;
        };
        for (int i = 0; i < L; ++i) zero_pad(lds + a.aop_off[i], ldY, n);
        for (int i = 0; i < L; ++i) zero_pad(lds + a.zb_off[i], a.zb_ld[i], a.width[i]);
        zero_pad(dl, a.zb_ld[L - 1], a.width[L - 1]);
        const int r = wave;
        const bool row_ok = r < rows;
        for (int j0 = 0; j0 < npad; j0 += 4 * 64) {
            double yd[4];
            float cu[4][ICNN_BE_MAX_LAYERS];
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 64 * k + lane;
                const bool ok = row_ok && j < n;
                yd[k] = ok ? a.y[(size_t)(s0 + r) * n + j] : 0.0;
#pragma unroll
                for (int i = 0; i < ICNN_BE_MAX_LAYERS; ++i)
                    cu[k][i] = ok && i < L ? ctx[(size_t)r * C + a.yu_off[i] + j] : 0.f;
            }
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int j = j0 + 64 * k + lane;
                if (j < npad) {
                    const float v = a.action_box ? (float)(2.0 * yd[k] - 1.0) : (float)yd[k];
                    const bool ok = row_ok && j < n;
                    ybuf[r * ldY + j] = ok ? v : 0.f;
#pragma unroll
                    for (int i = 0; i < ICNN_BE_MAX_LAYERS; ++i)
                        if (i < L) lds[a.aop_off[i] + r * ldY + j] = ok ? v * cu[k][i] : 0.f;
                }
            }
        }
    }
    __syncthreads();

    // ---------------- forward ------------------------------------------------------------
    const float *wyL = a.wpack + a.w_yu_f[L];       // final scalar layer: plain vectors
    const float *wzL = a.wpack + a.w_zu_f[L];
    for (int i = 0; i < L; ++i) {
        const int wi = a.width[i], wpad = pad16(wi);
        const bool last = i == L - 1;
        float *zout = lds + a.zb_off[i];
        const int ldo = a.zb_ld[i], NT = wpad / 16, NG = (wpad + 63) / 64;
        for (int unit = wave; unit < half_units(NG); unit += HW) {
            int g, mode;
            half_unit(NG, unit, g, mode);
            const int col = 64 * g + lane;
            const bool cok = col < wi;
            f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            float czu[2][4], cgt[2][4];
            const float wz = last && cok ? wzL[col] : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * h + r;
                    const bool ok = cok && row < rows && (mode == 0 || mode == 1 + h);
                    const float *c = ctx + (size_t)(ok ? row : 0) * C;
                    czu[h][r] = ok ? c[a.zu_off[i] + col] : 0.f;
                    cgt[h][r] = ok ? c[a.gate_off[i + 1] + col] : 0.f;
                }
            half_gemm_mode(mode, acc, lds + a.aop_off[i], ldY, a.wpack + a.w_yu_f[i], n, NT, g);
            if (i > 0)
                half_gemm_mode(mode, acc, lds + a.zb_off[i - 1], a.zb_ld[i - 1], a.wpack + a.w_zu_f[i], a.width[i - 1], NT, g);
            if (col < wpad) {
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (mode != 0 && mode != 1 + h) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * h + r;
                        float v = 0.f, d = 0.f;
                        if (row < rows && cok) {
                            const float z = act_fn(acc[h][r] + czu[h][r], a.alpha);
                            v = z * cgt[h][r];
                            const float gw = cgt[h][r] * wz;
                            d = gw * (v > 0.f ? 1.f : a.alpha);
                        }
                        zout[row * ldo + col] = v;
                        if (last) dl[row * ldo + col] = d;
                    }
                }
            }
        }
        __syncthreads();
    }

    // ---------------- backward ------------------------------------------------------------
    for (int i = L - 1; i >= 0; --i) {
        const int wi = a.width[i];
        const bool first = i == L - 1;
        const float *delta = first ? dl : lds + a.zb_off[i];
        const int ldd = a.zb_ld[i];
        const int NTy = npad / 16, NGy = (npad + 63) / 64;
        {   // dE/dy (+)= yu_i * (delta_i Wyu_i^T), starting from yu_L * wyu_L
            for (int unit = wave; unit < half_units(NGy); unit += HW) {
                int g, mode;
                half_unit(NGy, unit, g, mode);
                const int col = 64 * g + lane;
                const bool cok = col < n;
                f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                float cyu[2][4], cyL[2][4];
                const float wy = first && cok ? wyL[col] : 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * h + r;
                        const bool ok = cok && row < rows && (mode == 0 || mode == 1 + h);
                        cyu[h][r] = ok ? ctx[(size_t)row * C + a.yu_off[i] + col] : 0.f;
                        cyL[h][r] = ok && first ? ctx[(size_t)row * C + a.yu_off[L] + col] : 0.f;
                    }
                half_gemm_mode(mode, acc, delta, ldd, a.wpack + a.w_yu_b[i], wi, NTy, g);
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    if (mode != 0 && mode != 1 + h) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * h + r;
                        if (row < rows && cok) {
                            const float g_in = first ? cyL[h][r] * wy : gbuf[row * ldY + col];
                            gbuf[row * ldY + col] = __builtin_fmaf(cyu[h][r], acc[h][r], g_in);
                        }
                    }
                }
            }
        }
        if (i > 0) {   // delta_{i-1} = gate_i * (delta_i Wzu_i^T) * act'(pre_{i-1})
            const int wp = a.width[i - 1], wppad = pad16(wp);
            float *zprev = lds + a.zb_off[i - 1];
            const int ldp = a.zb_ld[i - 1], NTp = wppad / 16, NGp = (wppad + 63) / 64;
            for (int unit = wave; unit < half_units(NGp); unit += HW) {
                int g, mode;
                half_unit(NGp, unit, g, mode);
                const int col = 64 * g + lane;
                const bool cok = col < wp;
                f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                float cga[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * h + r;
                        const bool ok = cok && row < rows && (mode == 0 || mode == 1 + h);
                        cga[h][r] = ok ? ctx[(size_t)row * C + a.gate_off[i] + col] : 0.f;
                    }
                half_gemm_mode(mode, acc, delta, ldd, a.wpack + a.w_zu_b[i], wi, NTp, g);
                if (col < wppad) {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        if (mode != 0 && mode != 1 + h) continue;
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int row = 4 * h + r;
                            float d = 0.f;
                            if (row < rows && cok) {
                                const float ga = cga[h][r] * acc[h][r];
                                d = ga * (zprev[row * ldp + col] > 0.f ? 1.f : a.alpha);
                            }
                            zprev[row * ldp + col] = d;
                        }
                    }
                }
            }
        } else {
            // E = z_{L-1} . wzu_L + (y * yu_L) . wyu_L + zu_L: wave w = row w (same operations as fc_fg_tile)
            const float *zl = lds + a.zb_off[L - 1];
            const int ldz = a.zb_ld[L - 1], wl = a.width[L - 1];
            const int r = wave;
            if (r < rows) {
                const float *c = ctx + (size_t)r * C;
                float part = 0.f;
                for (int j = lane; j < wl; j += 64) part = __builtin_fmaf(zl[r * ldz + j], wzL[j], part);
                for (int j = lane; j < n; j += 64) {
                    const float yy = ybuf[r * ldY + j] * c[a.yu_off[L] + j];
                    part = __builtin_fmaf(yy, wyL[j], part);
                }
                const float e = wave_sum_f(part) + c[a.zu_off[L]];
                if (lane == 0) a.f[s0 + r] = e;
            }
        }
        __syncthreads();
    }
    const float gscale = a.action_box ? 2.f : 1.f;
    if (wave < rows)
        for (int j = lane; j < n; j += 64) a.g[(size_t)(s0 + wave) * n + j] = gscale * gbuf[wave * ldY + j];
    (void)qi;
}

}  // namespace
}  // namespace icnn_be
