// DROPPED EXPERIMENT (not built, not included by the product).
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
// the SAME packed weights (16-column tiles, k-block major).
//
// Work unit = 64 output columns x 8 samples: lane l holds column 64 u + l, every block of the instruction carries the
// same four samples (A operand of lane 4 b + i = sample i), two instructions per k -- samples 0-3 and 4-7, two
// independent accumulator chains that keep a SIMD's matrix pipe busy from a single wave.  Every B fragment is loaded
// once.  Measured (tools/probes/half_gemm_probe.hip, cycles per k-block and wave): 635 with two waves per SIMD (85 % of
// the pipe), 405 with one; 32-column units that split the samples over the two halves of the wave need the fragment in
// both halves -- loaded twice (493 at best) or exchanged with v_permlane32_swap (641: a swap between two MFMAs costs ~40
// cycles) -- and lost.
#pragma once
#include "be_picnn_fc_dev.h"

namespace icnn_be {

namespace {

constexpr int HW = 8, HT = HW * 64, HR = 8;    // waves, threads, samples of a half-tile workgroup
constexpr int HALF_MAX_SEGS = 32;
constexpr int HRING = 2;                       // k-blocks of B fragments in flight per wave (16 registers each)

__device__ __forceinline__ f4 mfma4(float a, float b, f4 c) { return __builtin_amdgcn_mfma_f32_4x4x1f32(a, b, c, 0, 0, 0); }

// The weight stream of a wave.  Every GEMM a wave runs in one evaluation is a SEGMENT = (packed operand, 64-column
// unit); the list per wave depends on the model only and is built on the host (half_schedule) in the order the loops of
// fc_fg_half run.  The kernel walks it one entry ahead: the last k-blocks of a segment refill the fragment ring from
// the NEXT segment's first k-blocks -- weights are constants, so the stream runs ahead across epilogues and barriers
// and a segment of ten k-blocks does not start with an empty ring.
struct HalfSched {
    unsigned short seg[HW][HALF_MAX_SEGS];      // (operand id << 8) | unit, id = 4 * layer + kind; after a wave's last
                                                // segment: its first one again (the ring refill never branches)
};
enum { HK_YU_F = 0, HK_ZU_F = 1, HK_YU_B = 2, HK_ZU_B = 3 };

// host: mirror of the unit loops of fc_fg_half; false if the model is outside what the half tile handles (a wave with
// more than HALF_MAX_SEGS - 1 segments, a k-block count that is not a multiple of the ring depth)
inline bool half_schedule(const FcArgs &a, HalfSched &sc) {
    const int L = a.L, npad = pad16(a.n);
    if (kblocks(a.n) % HRING) return false;
    for (int i = 0; i < L; ++i)
        if (kblocks(a.width[i]) % HRING) return false;
    for (int w = 0; w < HW; ++w) {
        int k = 0;
        auto push = [&](int layer, int kind, int unit) {
            if (k >= HALF_MAX_SEGS - 1 || unit > 255) { k = HALF_MAX_SEGS; return; }
            sc.seg[w][k++] = (unsigned short)(((4 * layer + kind) << 8) | unit);
        };
        for (int i = 0; i < L && k < HALF_MAX_SEGS; ++i)
            for (int u = w; u < (pad16(a.width[i]) + 63) / 64; u += HW) {
                push(i, HK_YU_F, u);
                if (i > 0) push(i, HK_ZU_F, u);
            }
        for (int i = L - 1; i >= 0 && k < HALF_MAX_SEGS; --i) {
            const int UY = (npad + 63) / 64, UP = i > 0 ? (pad16(a.width[i - 1]) + 63) / 64 : 0;
            for (int t = w; t < UY + UP; t += HW) {
                if (t < UY) push(i, HK_YU_B, t);
                else push(i, HK_ZU_B, t - UY);
            }
        }
        if (k >= HALF_MAX_SEGS) return false;
        if (k == 0) sc.seg[w][k++] = (unsigned short)(HK_YU_F << 8);       // a wave without work still streams valid addresses
        for (const unsigned short first = sc.seg[w][0]; k < HALF_MAX_SEGS; ++k) sc.seg[w][k] = first;
    }
    return true;
}

// A lane's view of a segment: where its B fragments start and how far apart consecutive k-blocks are.
struct HalfSeg {
    const f4 *bp;
    int kstride, KB;
};
template <typename ArgsT>
__device__ __forceinline__ HalfSeg half_seg(const ArgsT &a, unsigned code) {
    HalfSeg s{nullptr, 0, 0};
    code = __builtin_amdgcn_readfirstlane(code);
    const int id = code >> 8, u = code & 255, i = id >> 2, kind = id & 3;
    const int npad = pad16(a.n);
    long long off;
    int NT;
    if (kind == HK_YU_F) { off = a.w_yu_f[i]; s.KB = kblocks(a.n); NT = pad16(a.width[i]) / 16; }
    else if (kind == HK_ZU_F) { off = a.w_zu_f[i]; s.KB = kblocks(a.width[i - 1]); NT = pad16(a.width[i]) / 16; }
    else if (kind == HK_YU_B) { off = a.w_yu_b[i]; s.KB = kblocks(a.width[i]); NT = npad / 16; }
    else { off = a.w_zu_b[i]; s.KB = kblocks(a.width[i]); NT = pad16(a.width[i - 1]) / 16; }
    const int lane = thread_id() & 63;
    const int nt = min(4 * u + (lane >> 4), NT - 1);        // (columns beyond the operand: any valid address, result unused)
    s.bp = reinterpret_cast<const f4 *>(a.wpack + off) + (size_t)nt * 64 + (lane & 15);
    s.kstride = NT * 64;
    return s;
}

struct HalfRing { f4 r[HRING][4]; };
__device__ __forceinline__ void half_ring_fill(HalfRing &ring, const HalfSeg &s) {
#pragma unroll
    for (int d = 0; d < HRING; ++d)
#pragma unroll
        for (int q = 0; q < 4; ++q) ring.r[d][q] = s.bp[(size_t)d * s.kstride + 16 * q];
}

// acc{0,1}[r] += sum_k A[{0,4} + r][k] W[k][column of the lane] over the KB k-blocks of segment `cur` (KB a multiple of
// HRING; the pack and the LDS pad columns are zero beyond K, as for gemm_loop).  On entry the ring holds k-blocks
// 0 .. HRING-1 of `cur`, on exit those of `nxt`.  The A fragments of the next k-block are requested as soon as the last
// MFMAs that read the present ones have issued.
template <bool TAIL>
__device__ __forceinline__ void half_gemm_blocks(f4 &acc0, f4 &acc1, f4 (&x0)[4], f4 (&x1)[4], const float *a0, const float *a1,
                                                 HalfRing &ring, const f4 *refill, int rstride, int kb0, int KB) {
#pragma unroll
    for (int d = 0; d < HRING; ++d) {
        const int kb = kb0 + d;
        f4 b[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            b[q] = ring.r[d][q];
            ring.r[d][q] = refill[(size_t)(TAIL ? d : kb + HRING) * rstride + 16 * q];
        }
        // the refill loads are issued before the MFMAs of the k-block (left alone, the scheduler sinks them behind the last
        // reader of the slot and the ring runs empty)
        __builtin_amdgcn_sched_barrier(0);
#pragma unroll
        for (int s = 0; s < 3; ++s)
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                acc0 = mfma4(x0[q][s], b[q][s], acc0);
                acc1 = mfma4(x1[q][s], b[q][s], acc1);
            }
        const int nb = 16 * (kb + 1 < KB ? kb + 1 : kb);
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            acc0 = mfma4(x0[q][3], b[q][3], acc0);
            acc1 = mfma4(x1[q][3], b[q][3], acc1);
            x0[q] = *reinterpret_cast<const f4 *>(a0 + nb + 4 * q);
            x1[q] = *reinterpret_cast<const f4 *>(a1 + nb + 4 * q);
        }
    }
}
__device__ __forceinline__ void half_gemm(f4 &acc0, f4 &acc1, const float *A, int ld, const HalfSeg &cur, const HalfSeg &nxt,
                                          HalfRing &ring) {
    const int i = thread_id() & 3;
    const float *a0 = A + i * ld, *a1 = A + (4 + i) * ld;
    f4 x0[4], x1[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        x0[q] = *reinterpret_cast<const f4 *>(a0 + 4 * q);
        x1[q] = *reinterpret_cast<const f4 *>(a1 + 4 * q);
    }
    const int KB = cur.KB;
    for (int kb0 = 0; kb0 < KB - HRING; kb0 += HRING)
        half_gemm_blocks<false>(acc0, acc1, x0, x1, a0, a1, ring, cur.bp, cur.kstride, kb0, KB);
    half_gemm_blocks<true>(acc0, acc1, x0, x1, a0, a1, ring, nxt.bp, nxt.kstride, KB - HRING, KB);
}

// One half tile (HR samples, HT threads); LDS buffers laid out by fill_args(m, a, lds, HR).
template <typename ArgsT>
__device__ __forceinline__ void fc_fg_half(const ArgsT &a, const HalfSched &sched, int tile, float *lds) {
#pragma clang fp contract(off)
    const int tid = thread_id(), lane = tid & 63, wave = tid >> 6;
    const int s0 = tile * HR;
    const int rows = min(HR, a.batch - s0);
    const int n = a.n, L = a.L, C = a.ctx_width, ldY = a.ldY;
    const int npad = pad16(n);
    float *ybuf = lds + a.ybuf_off;
    long long tick = a.prof ? (long long)__builtin_readcyclecounter() : 0;
    auto lap = [&](int phase) {          // diagnostic only (tools/fc_phase_profile.py)
        if (a.prof) {
            const long long now = (long long)__builtin_readcyclecounter();
            if (lane == 0)
                atomicAdd(reinterpret_cast<unsigned long long *>(a.prof) + ((size_t)tile * HW + wave) * FC_PROF_PHASES + phase,
                          (unsigned long long)(now - tick));
            tick = now;
        }
    };
    if (a.finished) {                    // nothing to do if every sample of the half tile has left the loop
        int live = 0;
        if (tid < rows) live = a.finished[s0 + tid] == 0;
        if (!__syncthreads_or(live)) return;
    }
    const float *ctx = a.ctx + (size_t)s0 * C;
    float *gbuf = lds + a.gbuf_off, *dl = lds + a.dl_off;
    // the weight stream starts before anything else: its first fragments arrive while the rows are prepared
    int si = 0;
    HalfRing ring;
    HalfSeg cur = half_seg(a, sched.seg[wave][0]), nxt;
    half_ring_fill(ring, cur);
    auto gemm = [&](f4 (&acc)[2], const float *A, int ld) {
        nxt = half_seg(a, sched.seg[wave][si + 1]);
        half_gemm(acc[0], acc[1], A, ld, cur, nxt, ring);
        cur = nxt;
        ++si;
    };
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
    lap(0);

    // ---------------- forward ------------------------------------------------------------
    const float *wyL = a.wpack + a.w_yu_f[L];       // final scalar layer: plain vectors
    const float *wzL = a.wpack + a.w_zu_f[L];
    for (int i = 0; i < L; ++i) {
        const int wi = a.width[i], wpad = pad16(wi);
        const bool last = i == L - 1;
        float *zout = lds + a.zb_off[i];
        const int ldo = a.zb_ld[i];
        for (int unit = wave; unit < (wpad + 63) / 64; unit += HW) {
            const int col = 64 * unit + lane;
            const bool cok = col < wi;
            f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            gemm(acc, lds + a.aop_off[i], ldY);
            if (i > 0) gemm(acc, lds + a.zb_off[i - 1], a.zb_ld[i - 1]);
            float czu[2][4], cgt[2][4];              // (requested after the MFMA loops: the registers go to the fragment ring)
            const float wz = last && cok ? wzL[col] : 0.f;
#pragma unroll
            for (int h = 0; h < 2; ++h)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * h + r;
                    const bool ok = cok && row < rows;
                    const float *c = ctx + (size_t)(ok ? row : 0) * C;
                    czu[h][r] = ok ? c[a.zu_off[i] + col] : 0.f;
                    cgt[h][r] = ok ? c[a.gate_off[i + 1] + col] : 0.f;
                }
            if (col < wpad) {
#pragma unroll
                for (int h = 0; h < 2; ++h)
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
        lap(2 + 3 * i);
        __syncthreads();
        lap(3 + 3 * i);
    }

    // ---------------- backward ------------------------------------------------------------
    for (int i = L - 1; i >= 0; --i) {
        const int wi = a.width[i];
        const bool first = i == L - 1;
        const float *delta = first ? dl : lds + a.zb_off[i];
        const int ldd = a.zb_ld[i];
        const int UY = (npad + 63) / 64;
        const int wp = i > 0 ? a.width[i - 1] : 0, wppad = pad16(wp);
        const int UP = i > 0 ? (wppad + 63) / 64 : 0;
        // one list of units: dE/dy (+)= yu_i * (delta_i Wyu_i^T), then delta_{i-1} = gate_i * (delta_i Wzu_i^T) * act'
        for (int unit = wave; unit < UY + UP; unit += HW) {
            f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
            if (unit < UY) {
                const int col = 64 * unit + lane;
                const bool cok = col < n;
                gemm(acc, delta, ldd);
                float cyu[2][4], cyL[2][4];
                const float wy = first && cok ? wyL[col] : 0.f;
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * h + r;
                        const bool ok = cok && row < rows;
                        cyu[h][r] = ok ? ctx[(size_t)row * C + a.yu_off[i] + col] : 0.f;
                        cyL[h][r] = ok && first ? ctx[(size_t)row * C + a.yu_off[L] + col] : 0.f;
                    }
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * h + r;
                        if (row < rows && cok) {
                            const float g_in = first ? cyL[h][r] * wy : gbuf[row * ldY + col];
                            gbuf[row * ldY + col] = __builtin_fmaf(cyu[h][r], acc[h][r], g_in);
                        }
                    }
            } else {
                float *zprev = lds + a.zb_off[i - 1];
                const int ldp = a.zb_ld[i - 1];
                const int col = 64 * (unit - UY) + lane;
                const bool cok = col < wp;
                gemm(acc, delta, ldd);
                float cga[2][4];
#pragma unroll
                for (int h = 0; h < 2; ++h)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 4 * h + r;
                        cga[h][r] = cok && row < rows ? ctx[(size_t)row * C + a.gate_off[i] + col] : 0.f;
                    }
                if (col < wppad) {
#pragma unroll
                    for (int h = 0; h < 2; ++h)
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
        lap(i == 0 ? 11 : 8);
        if (i == 0) {
            // E = z_{L-1} . wzu_L + (y * yu_L) . wyu_L + zu_L, rows dealt from the last wave down (the first UY waves
            // have a dE/dy unit in this phase); same operations as fc_fg_tile
            const float *zl = lds + a.zb_off[L - 1];
            const int ldz = a.zb_ld[L - 1], wl = a.width[L - 1];
            for (int r = HW - 1 - wave; r < rows; r += HW) {
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
            lap(7);
        }
        __syncthreads();
        lap(i == 0 ? 12 : 10);
    }
    const float gscale = a.action_box ? 2.f : 1.f;
    if (wave < rows)
        for (int j = lane; j < n; j += 64) a.g[(size_t)(s0 + wave) * n + j] = gscale * gbuf[wave * ldY + j];
    lap(13);
}

}  // namespace
}  // namespace icnn_be
