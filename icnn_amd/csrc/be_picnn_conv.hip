// Energy + y-gradient of the y-dependent part of the convolutional PICNN used for image
// completion (completion/icnn_ebundle.py:376-452; gradient = tf.gradients(E_, y_), :120-121).
//
//   y_red_0 = y,  y_red_{l+1} = conv(y_red_l; k_l, s_l) + b                               (:394-396)
//   z_l = relu( conv(z_{l-1} * gate_l; Wzu_l >= 0)[l>0] + conv(y_red_l * yu_l; Wyu_l) + zu_l ),  l = 0..2
//   z_3 = relu( (flatten(z_2) * gate_3) W_3 + zu_3 ),   E = (z_3 * gate_4) . w_4 + zu_4    (:411-445)
//
// Four launches per evaluation, every contraction with more than one output channel on v_mfma_f32_16x16x4_f32:
//   conv_fwd_kernel     one workgroup per sample: activations in LDS as zero-bordered NHWC maps, the three
//                       convolutions as implicit GEMMs (M = output positions, N = output channels, K = taps x input
//                       channels) whose A fragments are gathered straight from the padded maps -- 16 consecutive
//                       channels of one tap are one k-block, a lane's four k-values one ds_read_b128 -- and whose B
//                       fragments stream from the packed weights (16 B per lane, register ring); writes
//                       A1, A2 (ReLU masks for the backward pass) and flatten(z_2) * gate_3
//   conv_fc_fwd_kernel  the 2048 x 512 layer for a tile of 16 SAMPLES x 64 outputs per workgroup: the 4 MB matrix is
//                       read once per 16 samples instead of once per sample
//   conv_fc_bwd_kernel  energy; delta_3; delta_2 = gate_3 * (delta_3 W_3^T) * [z_2 > 0], 16 samples x 64 outputs
//   conv_bwd_kernel     one workgroup per sample: the transposed convolutions as implicit GEMMs -- stride 1 directly,
//                       stride 2 as four parity classes of positions (2 x 2 valid taps each), and the final
//                       32 -> 1 channel, stride-4 one as a "pixel shuffle": M = the 4 x 4-pixel cells, N = the 16
//                       pixels of a cell, K = the 3 x 3 neighbouring cells x 32 channels (zero weights where a tap
//                       does not reach)
// The single-channel pieces (y_red chain, the yu terms, their transposes) stay on the VALU.  float32 throughout like
// the reference; per output the accumulation is the k-ordered fma chain of the MFMA (kk = 16 kb + 4 q + s, s outer),
// which oracle/picnn_conv_chain.c reproduces bit for bit.  NHWC, 'SAME' padding.
#include <hip/hip_runtime.h>

#include <type_traits>

#include "be_common.h"
#include "be_kernels.h"
#include "icnn_be.h"

namespace icnn_be {

namespace {

constexpr int CT = 512, CW = CT / 64;   // threads / waves per workgroup of the per-sample kernels
constexpr int FT = 256;                 // threads of the fc tile kernels (4 waves, one 16-column tile each)
constexpr int RD = 8;                   // depth of the fragment register rings (k-blocks in flight per wave)
constexpr int FKS = 4;                  // the fc 2048 -> 512 layer splits its K range over the FKS waves of a workgroup
constexpr int PADI = 2, PADM = 1;       // zero borders: image-sized buffers / feature maps

struct ConvArgs {
    int H, W, F[3], K[3], S[3], P[3];
    int oh[3], ow[3];
    int fch, flat, n, C;
    int c_yu[3], c_zu[3], c_gate[5], c_zu3, c_zu4;          // context offsets (floats)
    // weight offsets (floats): raw single-channel pieces, then the packed MFMA operands
    long long w_yu[3], w_yr[2], b_yr[2], w_fc4;
    long long p_l1, p_l2, p_l3, p_l3t, p_l2t[4], p_ps, p_fc3, p_fc3t;
    int kb_ps;                                              // k-blocks of the pixel-shuffle operand (padded to RD)
    const float *wpack, *ctx;
    const double *y;
    float *f, *g;
    const int *skip;
    int batch;
    float *a1s, *a2s, *zflat, *a4, *d3, *d2;                // workspace: [B][p1 F0], [B][p2 F1], [B][flat], 2 x [B][fch], [B][flat]
};

// A zero-bordered [h][w][c] buffer in LDS.  Multi-channel maps keep their pixels CPAD floats apart from a multiple of
// 32: the 16 positions of an A-fragment gather (one ds_read_b128 per lane, same channel quad) then spread over the
// banks instead of all landing on the same four.
constexpr int CPAD = 4;
struct Map {
    float *p;
    int w, c, pad;
    __device__ __forceinline__ int cs() const { return c == 1 ? 1 : c + CPAD; }
    __device__ __forceinline__ int at(int y, int x) const { return ((y + pad) * (w + 2 * pad) + x + pad) * cs(); }
};
__host__ __device__ inline int map_floats(int pixels, int c) { return pixels * (c == 1 ? 1 : c + CPAD); }

// The completion network of the reference (completion/icnn_ebundle.py:344): compile-time layer constants, so that
// the tap / channel-block decoding inside the k-loops is shifts and multiplications, not integer divisions.
struct Net {
    static constexpr int K0 = 8, S0 = 4, P0 = 2, F0 = 32, K1 = 4, S1 = 2, P1 = 1, F1 = 64, K2 = 3, S2 = 1, P2 = 1, F2 = 64;
};

__host__ __device__ inline int kb16(int K) { return (K + 15) / 16; }

// pack[(kb*NT + nt)*256 + lane*4 + s] = W[kb*16 + 4*(lane>>4) + s][nt*16 + (lane&15)]   (zero beyond K, N; KB k-blocks)
template <class Get>
void pack_frag(Get get, int K, int N, int KB, float *dst) {
    const int NT = (N + 15) / 16;
    for (int kb = 0; kb < KB; ++kb)
        for (int nt = 0; nt < NT; ++nt)
            for (int lane = 0; lane < 64; ++lane)
                for (int s = 0; s < 4; ++s) {
                    const int kk = kb * 16 + 4 * (lane >> 4) + s, nn = nt * 16 + (lane & 15);
                    dst[((size_t)(kb * NT + nt) * 64 + lane) * 4 + s] = (kk < K && nn < N) ? get(kk, nn) : 0.f;
                }
}
inline size_t frag_floats(int KB, int N) { return (size_t)KB * ((N + 15) / 16) * 256; }

// acc[t] += A (16 x 16 KB, fragments from `ga`) * packed tiles bp[t] over the k-blocks [kb_lo, kb_hi).  The B fragments
// run an RDN-deep register ring ahead of the MFMAs, and so do the A fragments (LDS gathers, or memory for the fc
// layer): every wave is a latency chain on L2, so the ring depth is what sets its speed.  Per output element: the
// chain kb = kb_lo.., s = 0..3 (instruction), q = 0..3 (inside the instruction).
template <int NTW, int RDN, class GA>
__device__ __forceinline__ void mfma_stream(f4 (&acc)[NTW], GA ga, const f4 *const (&bp)[NTW], size_t kstride, int kb_lo,
                                            int kb_hi) {
    // the k-block range is wave-uniform (a workgroup's waves may take different ranges: the fc layer's K split), and the compiler
    // must KNOW it: derived from the thread index it treated `kb < kb_hi` as divergent, merged the ring slots under exec masks and
    // waited for every refill right behind its request -- a ring of depth one (round 6, conv_fc_fwd_kernel's ISA)
    kb_lo = __builtin_amdgcn_readfirstlane(kb_lo);
    kb_hi = __builtin_amdgcn_readfirstlane(kb_hi);
    f4 br[RDN][NTW], ar[RDN];
#pragma unroll
    for (int d = 0; d < RDN; ++d) {
        const int kb = kb_lo + d < kb_hi ? kb_lo + d : kb_hi - 1;
        ar[d] = ga(kb);
#pragma unroll
        for (int t = 0; t < NTW; ++t) br[d][t] = bp[t][(size_t)kb * kstride];
    }
    auto four = [&](const f4 &a, const f4 (&x)[NTW]) {
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.x, x[t].x, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.y, x[t].y, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.z, x[t].z, acc[t], 0, 0, 0);
#pragma unroll
        for (int t = 0; t < NTW; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a.w, x[t].w, acc[t], 0, 0, 0);
    };
    // Whole turns of the ring WITHOUT a condition around the refill (round 6): with `if (kb < kb_hi)` around every slot the
    // refilled fragment was a conditional value -- the compiler loaded it into scratch registers, waited for it right behind the
    // request and copied it into the slot: eight k-blocks "in flight" that were awaited one by one.  The last partial turn only
    // consumes what the ring holds.  Same MFMA sequence per output element.
    int kb0 = kb_lo;
    for (; kb0 + RDN <= kb_hi; kb0 += RDN) {
#pragma unroll
        for (int d = 0; d < RDN; ++d) {
            const int kb = kb0 + d;
            const f4 a = ar[d];
            f4 x[NTW];
            const int nk = kb + RDN < kb_hi ? kb + RDN : kb_hi - 1;   // ring refill (clamped re-read at the tail)
            ar[d] = ga(nk);
#pragma unroll
            for (int t = 0; t < NTW; ++t) { x[t] = br[d][t]; br[d][t] = bp[t][(size_t)nk * kstride]; }
            four(a, x);
        }
    }
#pragma unroll
    for (int d = 0; d < RDN; ++d)
        if (kb0 + d < kb_hi) {                                    // (wave-uniform) the ring's slots 0 .. hold kb0 ..
            f4 x[NTW];
#pragma unroll
            for (int t = 0; t < NTW; ++t) x[t] = br[d][t];
            four(ar[d], x);
        }
}

// single-channel convolution at one output position: chain over (ky, kx), fused multiply-adds
template <int K, int S, int P>
__device__ __forceinline__ float conv1_at(const Map in, const float *W, int Cout, int ch, int oy, int ox) {
    float acc = 0.f;
#pragma unroll
    for (int ky = 0; ky < K; ++ky)
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
            acc = __builtin_fmaf(in.p[in.at(oy * S + ky - P, ox * S + kx - P)], W[(size_t)(ky * K + kx) * Cout + ch], acc);
    return acc;
}
// transposed convolution at one input position, R source channels, one destination channel `ch` of Cout:
// taps ky = (iy+P) mod S, +S, ..; kx likewise; channel innermost
template <int K, int S, int P, int R>
__device__ __forceinline__ float convt_at(const Map dout, const float *Wt, int Cout, int ch, int iy, int ix) {
    float acc = 0.f;
    const int ry = (iy + P) % S, rx = (ix + P) % S;
#pragma unroll
    for (int ti = 0; ti < K / S; ++ti)
#pragma unroll
        for (int tj = 0; tj < K / S; ++tj) {
            const int ky = ry + S * ti, kx = rx + S * tj;
            const float *src = dout.p + dout.at((iy + P - ky) / S, (ix + P - kx) / S);
            const float *wp = Wt + (size_t)((ky * K + kx) * R) * Cout + ch;
#pragma unroll 8
            for (int r = 0; r < R; ++r) acc = __builtin_fmaf(src[r], wp[(size_t)r * Cout], acc);
        }
    return acc;
}

// Transposed convolution onto ONE destination channel from R source channels, for the output position (iy, ix), by a
// group of eight neighbouring lanes (c = lane & 7): lane c runs the chain over the valid taps (ti, tj) and the channels
// c, c + 8, .., the eight partial sums meet in NumPy's pairwise order ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) through DPP.
// Every lane of the group returns the total.
template <int K, int S, int P, int R>
__device__ __forceinline__ float convt_group8(const Map dout, const float *Wt, int iy, int ix, int c) {
    float acc = 0.f;
    const int ry = (iy + P) % S, rx = (ix + P) % S;
#pragma unroll
    for (int ti = 0; ti < K / S; ++ti)
#pragma unroll
        for (int tj = 0; tj < K / S; ++tj) {
            const int ky = ry + S * ti, kx = rx + S * tj;
            const float *src = dout.p + dout.at((iy + P - ky) / S, (ix + P - kx) / S);
            const float *wp = Wt + (size_t)(ky * K + kx) * R;
#pragma unroll
            for (int m = 0; m < R / 8; ++m) acc = __builtin_fmaf(src[c + 8 * m], wp[c + 8 * m], acc);
        }
    return sum8_numpy_order(acc);
}

// ---------------------------------------------------------------------------------------------------------
// forward convolutions, one workgroup per sample
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CT) void conv_fwd_kernel(ConvArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    if (a.skip && a.skip[u]) return;
    const int n = a.n, H = a.H, W = a.W;
    const int oh0 = a.oh[0], ow0 = a.ow[0], oh1 = a.oh[1], ow1 = a.ow[1], ow2 = a.ow[2];
    const int p1 = oh0 * ow0, p2 = oh1 * ow1, p3 = a.oh[2] * ow2;
    constexpr int F0 = Net::F0, F1 = Net::F1, F2 = Net::F2;
    const int simg = (H + 2 * PADI) * (W + 2 * PADI), sm1 = (oh0 + 2) * (ow0 + 2), sm2 = (oh1 + 2) * (ow1 + 2);
    float *base = lds;
    auto take = [&](int floats) { float *p = base; base += (floats + 3) & ~3; return p; };
    const Map ybuf{take(simg), W, 1, PADI}, a0{take(simg), W, 1, PADI};
    const Map yr1{take(sm1), ow0, 1, PADM}, ay1{take(sm1), ow0, 1, PADM};
    const Map yr2{take(sm2), ow1, 1, PADM}, ay2{take(sm2), ow1, 1, PADM};
    const Map A1{take(map_floats(sm1, F0)), ow0, F0, PADM}, A2{take(map_floats(sm2, F1)), ow1, F1, PADM};
    const int lds_floats = (int)(base - lds);
    const float *ctx = a.ctx + (size_t)u * a.C;
    const float *wp = a.wpack;
    for (int e = tid; e < lds_floats; e += CT) lds[e] = 0.f;       // borders (and everything else) start at zero
    __syncthreads();
    // P0: y (rounded to float32 like a TensorFlow feed), y * yu_0
    for (int j = tid; j < n; j += CT) {
        const float v = (float)a.y[(size_t)u * n + j];
        ybuf.p[ybuf.at(j / W, j % W)] = v;
        a0.p[a0.at(j / W, j % W)] = v * ctx[a.c_yu[0] + j];
    }
    __syncthreads();
    // P1: y_red_1 (VALU); z_0 = relu(conv(y*yu_0; Wyu_0) + zu_0) -> A1 = z_0 * gate_1: M = p1 positions, N = F0, K = K0^2 taps
    if (tid < p1) yr1.p[yr1.at(tid / ow0, tid % ow0)] = conv1_at<Net::K0, Net::S0, Net::P0>(ybuf, wp + a.w_yr[0], 1, 0, tid / ow0, tid % ow0) + wp[a.b_yr[0]];
    {
        constexpr int NT = F0 / 16, KB = Net::K0 * Net::K0 / 16, S = Net::S0, P = Net::P0, K = Net::K0;
        const int MT = p1 / 16;
        for (int mt = wave; mt < MT; mt += CW) {
            const int pos = 16 * mt + r16, oy = pos / ow0, ox = pos % ow0;
            // k = ky*K + kx (one input channel): k-block kb holds 16/K rows of taps; a lane's four k are four neighbouring kx
            auto ga = [&](int kb) -> f4 {
                const int k0 = 16 * kb + 4 * q, ky = k0 / K, kx = k0 % K;          // K is a compile-time power of two
                return *reinterpret_cast<const f4 *>(a0.p + a0.at(oy * S + ky - P, ox * S + kx - P));
            };
            const f4 *wb = reinterpret_cast<const f4 *>(wp + a.p_l1) + lane;
            for (int nt = 0; nt < NT; nt += 2) {
                f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                const f4 *const bp[2] = {wb + (size_t)nt * 64, wb + (size_t)(nt + 1 < NT ? nt + 1 : nt) * 64};
                float zu_[2][4], gt_[2][4];                       // epilogue operands requested ahead of the k-loop
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int e = (16 * mt + 4 * q + r) * F0 + 16 * (nt + t < NT ? nt + t : nt) + r16;
                        zu_[t][r] = ctx[a.c_zu[0] + e];
                        gt_[t][r] = ctx[a.c_gate[1] + e];
                    }
                mfma_stream<2, RD>(acc, ga, bp, (size_t)NT * 64, 0, KB);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (nt + t >= NT) continue;
                    const int ch = 16 * (nt + t) + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int po = 16 * mt + 4 * q + r, e = po * F0 + ch;
                        const float pre = acc[t][r] + zu_[t][r];
                        const float v = (pre > 0.f ? pre : 0.f) * gt_[t][r];
                        A1.p[A1.at(po / ow0, po % ow0) + ch] = v;
                        a.a1s[(size_t)u * p1 * F0 + e] = v;
                    }
                }
            }
        }
    }
    __syncthreads();
    // P2: y_red_1 * yu_1, y_red_2
    if (tid < p1) ay1.p[ay1.at(tid / ow0, tid % ow0)] = yr1.p[yr1.at(tid / ow0, tid % ow0)] * ctx[a.c_yu[1] + tid];
    if (tid < p2) yr2.p[yr2.at(tid / ow1, tid % ow1)] = conv1_at<Net::K1, Net::S1, Net::P1>(yr1, wp + a.w_yr[1], 1, 0, tid / ow1, tid % ow1) + wp[a.b_yr[1]];
    __syncthreads();
    // P3 / P4: z_l = relu(conv(A_l; Wzu_l) + conv(y_red_l * yu_l; Wyu_l) + zu_l), l = 1, 2: M = positions, N = F_l,
    //          K = taps x F_{l-1}; the single-channel yu term is a VALU chain in the epilogue
    auto layer = [&](auto Lc, auto Kc, auto Sc, auto Pc, auto Rc, auto Fc, const Map in, const Map ayl, long long pack) {
        constexpr int l = decltype(Lc)::value, K = decltype(Kc)::value, S = decltype(Sc)::value, P = decltype(Pc)::value;
        constexpr int R = decltype(Rc)::value, Fl = decltype(Fc)::value, NT = Fl / 16, cpb = R / 16, KB = K * K * cpb;
        const int owl = a.ow[l], npos = a.oh[l] * a.ow[l], MT = npos / 16;
        for (int tile = wave; tile < MT * NT; tile += CW) {
            const int mt = tile % MT, nt = tile / MT;
            const int pos = 16 * mt + r16, oy = pos / owl, ox = pos % owl;
            const float *origin = in.p + in.at(oy * S - P, ox * S - P) + 4 * q;
            const int rowf = (in.w + 2 * in.pad) * in.cs(), pixf = in.cs();
            auto ga = [&](int kb) -> f4 {
                const int tap = kb / cpb, cb = (kb - tap * cpb) * 16, ky = tap / K, kx = tap - ky * K;
                return *reinterpret_cast<const f4 *>(origin + ky * rowf + kx * pixf + cb);
            };
            f4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            const f4 *const bp[1] = {reinterpret_cast<const f4 *>(wp + pack) + (size_t)nt * 64 + lane};
            const int ch = 16 * nt + r16;
            float zu_[4], gt_[4], acc2[4];                        // epilogue operands and the single-channel yu chain
#pragma unroll                                                    // are formed ahead of the k-loop
            for (int r = 0; r < 4; ++r) {
                const int po = 16 * mt + 4 * q + r, e = po * Fl + ch;
                zu_[r] = ctx[a.c_zu[l] + e];
                gt_[r] = ctx[a.c_gate[l + 1] + e];
                acc2[r] = conv1_at<K, S, P>(ayl, wp + a.w_yu[l], Fl, ch, po / owl, po % owl);
            }
            mfma_stream<1, RD>(acc, ga, bp, (size_t)NT * 64, 0, KB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int po = 16 * mt + 4 * q + r, e = po * Fl + ch;
                float pre = acc[0][r] + acc2[r];
                pre = pre + zu_[r];
                const float v = (pre > 0.f ? pre : 0.f) * gt_[r];
                if (l == 1) {
                    A2.p[A2.at(po / owl, po % owl) + ch] = v;
                    a.a2s[(size_t)u * npos * Fl + e] = v;
                } else {
                    a.zflat[(size_t)u * a.flat + e] = v;           // flatten(z_2) * gate_3, NHWC row-major
                }
            }
        }
    };
    using std::integral_constant;
    layer(integral_constant<int, 1>{}, integral_constant<int, Net::K1>{}, integral_constant<int, Net::S1>{},
          integral_constant<int, Net::P1>{}, integral_constant<int, F0>{}, integral_constant<int, F1>{}, A1, ay1, a.p_l2);
    if (tid < p2) ay2.p[ay2.at(tid / ow1, tid % ow1)] = yr2.p[yr2.at(tid / ow1, tid % ow1)] * ctx[a.c_yu[2] + tid];
    __syncthreads();
    layer(integral_constant<int, 2>{}, integral_constant<int, Net::K2>{}, integral_constant<int, Net::S2>{},
          integral_constant<int, Net::P2>{}, integral_constant<int, F1>{}, integral_constant<int, F2>{}, A2, ay2, a.p_l3);
    (void)p3;
}

// ---------------------------------------------------------------------------------------------------------
// fc 2048 -> 512 for a tile of 16 samples x 64 outputs; A fragments straight from the flattened activations in memory
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(FT) void conv_fc_fwd_kernel(ConvArgs a) {
#pragma clang fp contract(off)
    // One workgroup = one tile of 16 samples x 16 outputs; its FKS waves take a quarter of the K range each (four
    // times the waves in flight on a layer that is a pure latency chain per wave) and the partial sums are combined
    // in a fixed order: ((p0 + p1) + p2) + p3.
    __shared__ f4 part[FKS][64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const int s0 = blockIdx.y * 16, nt = blockIdx.x, NT = a.fch / 16, KB = a.flat / 16, chunk = KB / FKS;
    const int srow = s0 + r16 < a.batch ? s0 + r16 : a.batch - 1;
    const float *arow = a.zflat + (size_t)srow * a.flat + 4 * q;
    auto ga = [&](int kb) -> f4 { return *reinterpret_cast<const f4 *>(arow + 16 * kb); };
    f4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    const f4 *const bp[1] = {reinterpret_cast<const f4 *>(a.wpack + a.p_fc3) + (size_t)nt * 64 + lane};
    mfma_stream<1, RD>(acc, ga, bp, (size_t)NT * 64, wave * chunk, (wave + 1) * chunk);
    part[wave][lane] = acc[0];
    __syncthreads();
    if (wave != 0) return;
    f4 tot = part[0][lane] + part[1][lane];
#pragma unroll
    for (int w = 2; w < FKS; ++w) tot = tot + part[w][lane];
    const int j = 16 * nt + r16;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = s0 + 4 * q + r;
        if (u >= a.batch || (a.skip && a.skip[u])) continue;
        const float *ctx = a.ctx + (size_t)u * a.C;
        const float pre = tot[r] + ctx[a.c_zu3 + j];
        const float gate = ctx[a.c_gate[4] + j], v = (pre > 0.f ? pre : 0.f) * gate;
        a.a4[(size_t)u * a.fch + j] = v;
        a.d3[(size_t)u * a.fch + j] = v > 0.f ? gate * a.wpack[a.w_fc4 + j] : 0.f;     // delta_3 = gate_4 w_4 [z_3 > 0]
    }
}

// energy, delta_3 = gate_4 * w_4 * [z_3 > 0], delta_2 = gate_3 * (delta_3 W_3^T) * [z_2 > 0]   (16 samples x 64 outputs)
__global__ __launch_bounds__(FT) void conv_fc_bwd_kernel(ConvArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    const int s0 = blockIdx.y * 16, fch = a.fch, ld = fch + 8;           // pitch == 8 (mod 64) for fch = 512
    const float *w4 = a.wpack + a.w_fc4;
    for (int e = tid; e < 16 * fch / 4; e += FT) {            // delta_3 of the tile (written by the forward kernel)
        const int row = e / (fch / 4), j4 = e - row * (fch / 4), u = s0 + row < a.batch ? s0 + row : a.batch - 1;
        *reinterpret_cast<f4 *>(lds + row * ld + 4 * j4) = *reinterpret_cast<const f4 *>(a.d3 + (size_t)u * fch + 4 * j4);
    }
    if (blockIdx.x == 0) {                     // energies of the tile: one wave per sample, lane chains + butterfly
        for (int row = wave; row < 16; row += FT / 64) {
            const int u = s0 + row;
            if (u >= a.batch) break;
            float part = 0.f;
            for (int j = lane; j < fch; j += 64) part = __builtin_fmaf(a.a4[(size_t)u * fch + j], w4[j], part);
            part = wave_sum_f(part);
            if (lane == 0 && !(a.skip && a.skip[u])) a.f[u] = part + a.ctx[(size_t)u * a.C + a.c_zu4];
        }
    }
    __syncthreads();
    const int nt = blockIdx.x * (FT / 64) + wave, NT = a.flat / 16, KB = fch / 16;
    if (nt >= NT) return;
    const float *arow = lds + r16 * ld + 4 * q;
    auto ga = [&](int kb) -> f4 { return *reinterpret_cast<const f4 *>(arow + 16 * kb); };
    f4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
    const f4 *const bp[1] = {reinterpret_cast<const f4 *>(a.wpack + a.p_fc3t) + (size_t)nt * 64 + lane};
    const int k = 16 * nt + r16;
    float gt_[4], mk_[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = s0 + 4 * q + r < a.batch ? s0 + 4 * q + r : a.batch - 1;
        gt_[r] = a.ctx[(size_t)u * a.C + a.c_gate[3] + k];
        mk_[r] = a.zflat[(size_t)u * a.flat + k];
    }
    mfma_stream<1, RD>(acc, ga, bp, (size_t)NT * 64, 0, KB);
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int u = s0 + 4 * q + r;
        if (u >= a.batch) continue;
        const float dz = gt_[r] * acc[0][r];
        a.d2[(size_t)u * a.flat + k] = mk_[r] > 0.f ? dz : 0.f;
    }
}

// ---------------------------------------------------------------------------------------------------------
// transposed convolutions, one workgroup per sample
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(CT) void conv_bwd_kernel(ConvArgs a) {
#pragma clang fp contract(off)
    extern __shared__ __attribute__((aligned(16))) float lds[];
    const int u = blockIdx.x, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, r16 = lane & 15, q = lane >> 4;
    if (a.skip && a.skip[u]) return;
    const int n = a.n, W = a.W;
    const int oh0 = a.oh[0], ow0 = a.ow[0], oh1 = a.oh[1], ow1 = a.ow[1], oh2 = a.oh[2], ow2 = a.ow[2];
    const int p1 = oh0 * ow0, p2 = oh1 * ow1, p3 = oh2 * ow2;
    constexpr int F0 = Net::F0, F1 = Net::F1, F2 = Net::F2;
    const int sm1 = (oh0 + 2) * (ow0 + 2), sm2 = (oh1 + 2) * (ow1 + 2), sm3 = (oh2 + 2) * (ow2 + 2);
    float *base = lds;
    auto take = [&](int floats) { float *p = base; base += (floats + 3) & ~3; return p; };
    const Map D2{take(map_floats(sm3, F2)), ow2, F2, PADM}, D1{take(map_floats(sm2, F1)), ow1, F1, PADM};
    const Map D0{take(map_floats(sm1, F0)), ow0, F0, PADM};
    const Map dyr2{take(sm2), ow1, 1, PADM}, dyr1{take(sm1), ow0, 1, PADM};
    const int lds_floats = (int)(base - lds);
    const float *ctx = a.ctx + (size_t)u * a.C;
    const float *wp = a.wpack;
    for (int e = tid; e < lds_floats; e += CT) lds[e] = 0.f;
    __syncthreads();
    for (int e = tid; e < p3 * F2; e += CT) {
        const int pos = e / F2, f = e - pos * F2;
        D2.p[D2.at(pos / ow2, pos % ow2) + f] = a.d2[(size_t)u * a.flat + e];
    }
    __syncthreads();
    // P9: delta_1 = gate_2 * convT(delta_2; Wzu_2) * [z_1 > 0] (stride 1: every tap valid): M = p2, N = F1, K = taps x F2;
    //     d y_red_2 = yu_2 * convT(delta_2; Wyu_2) on the VALU
    {
        constexpr int K = Net::K2, P = Net::P2, NT = F1 / 16, cpb = F2 / 16, KB = K * K * cpb;
        const int MT = p2 / 16, rowf = (D2.w + 2 * D2.pad) * D2.cs(), pixf = D2.cs();
        for (int tile = wave; tile < MT * NT; tile += CW) {
            const int mt = tile % MT, nt = tile / MT, pos = 16 * mt + r16, iy = pos / ow1, ix = pos % ow1;
            const float *origin = D2.p + D2.at(iy + P, ix + P) + 4 * q;
            auto ga = [&](int kb) -> f4 {
                const int tap = kb / cpb, cb = (kb - tap * cpb) * 16, ky = tap / K, kx = tap - ky * K;
                return *reinterpret_cast<const f4 *>(origin - ky * rowf - kx * pixf + cb);
            };
            f4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            const f4 *const bp[1] = {reinterpret_cast<const f4 *>(wp + a.p_l3t) + (size_t)nt * 64 + lane};
            const int ch = 16 * nt + r16;
            float gt_[4], mk_[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = (16 * mt + 4 * q + r) * F1 + ch;
                gt_[r] = ctx[a.c_gate[2] + e];
                mk_[r] = a.a2s[(size_t)u * p2 * F1 + e];
            }
            mfma_stream<1, RD>(acc, ga, bp, (size_t)NT * 64, 0, KB);
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int po = 16 * mt + 4 * q + r;
                const float dz = gt_[r] * acc[0][r];
                D1.p[D1.at(po / ow1, po % ow1) + ch] = mk_[r] > 0.f ? dz : 0.f;
            }
        }
        for (int pg = tid >> 3; pg < p2; pg += CT / 8) {          // (whole groups of eight lanes share pg)
            const float tot = convt_group8<K, Net::S2, P, F2>(D2, wp + a.w_yu[2], pg / ow1, pg % ow1, tid & 7);
            if ((tid & 7) == 0) dyr2.p[dyr2.at(pg / ow1, pg % ow1)] = ctx[a.c_yu[2] + pg] * tot;
        }
    }
    __syncthreads();
    // P10: delta_0 = gate_1 * convT(delta_1; Wzu_1) * [z_0 > 0] (stride S): positions of one parity class (iy mod S, ix mod S)
    //      share their (K/S)^2 valid taps: per class M = p1 / S^2, N = F0, K = (K/S)^2 x F1;
    //      d y_red_1 = yu_1 * convT(delta_1; Wyu_1) + convT(d y_red_2; Wyr_1) on the VALU
    {
        constexpr int K = Net::K1, S = Net::S1, P = Net::P1, NT = F0 / 16, cpb = F1 / 16, TS = K / S, KB = TS * TS * cpb;
        const int cw = ow0 / S, cpos = p1 / (S * S), MT = cpos / 16, ntiles = S * S * MT;     // class maps are (oh0/S) x (ow0/S)
        const int rowf = (D1.w + 2 * D1.pad) * D1.cs(), pixf = D1.cs();
        for (int tile = wave; tile < ntiles; tile += CW) {
            const int cls = tile / MT, mt = tile - cls * MT, ca = cls / S, cbb = cls - ca * S;
            const int ry = (ca + P) % S, rx = (cbb + P) % S;
            const int cp = 16 * mt + r16, iy = (cp / cw) * S + ca, ix = (cp % cw) * S + cbb;
            // source pixel of tap (ry + S ti, rx + S tj): ((iy + P - ry) / S - ti, (ix + P - rx) / S - tj)
            const float *origin = D1.p + D1.at((iy + P - ry) / S, (ix + P - rx) / S) + 4 * q;
            auto ga = [&](int kb) -> f4 {
                const int tap = kb / cpb, cb = (kb - tap * cpb) * 16, ti = tap / TS, tj = tap - ti * TS;
                return *reinterpret_cast<const f4 *>(origin - ti * rowf - tj * pixf + cb);
            };
            const f4 *wb = reinterpret_cast<const f4 *>(wp + a.p_l2t[cls]) + lane;
#pragma unroll
            for (int nt = 0; nt < NT; nt += 2) {
                f4 acc[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};
                const f4 *const bp[2] = {wb + (size_t)nt * 64, wb + (size_t)(nt + 1 < NT ? nt + 1 : nt) * 64};
                float gt_[2][4], mk_[2][4];
#pragma unroll
                for (int t = 0; t < 2; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cq = 16 * mt + 4 * q + r, py = (cq / cw) * S + ca, px = (cq % cw) * S + cbb;
                        const int e = (py * ow0 + px) * F0 + 16 * (nt + t < NT ? nt + t : nt) + r16;
                        gt_[t][r] = ctx[a.c_gate[1] + e];
                        mk_[t][r] = a.a1s[(size_t)u * p1 * F0 + e];
                    }
                mfma_stream<2, RD>(acc, ga, bp, (size_t)NT * 64, 0, KB);
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    if (nt + t >= NT) continue;
                    const int ch = 16 * (nt + t) + r16;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int cq = 16 * mt + 4 * q + r, py = (cq / cw) * S + ca, px = (cq % cw) * S + cbb;
                        const float dz = gt_[t][r] * acc[t][r];
                        D0.p[D0.at(py, px) + ch] = mk_[t][r] > 0.f ? dz : 0.f;
                    }
                }
            }
        }
        for (int pg = tid >> 3; pg < p1; pg += CT / 8) {
            const int py = pg / ow0, px = pg % ow0;
            const float tot = convt_group8<K, S, P, F1>(D1, wp + a.w_yu[1], py, px, tid & 7);
            if ((tid & 7) == 0) {
                const float t1 = ctx[a.c_yu[1] + pg] * tot;
                dyr1.p[dyr1.at(py, px)] = t1 + convt_at<K, S, P, 1>(dyr2, wp + a.w_yr[1], 1, 0, py, px);
            }
        }
    }
    __syncthreads();
    // P11: dE/dy = yu_0 * convT(delta_0; Wyu_0) + convT(d y_red_1; Wyr_0).  One output channel, stride S0 = K0 / 2:
    //      "pixel shuffle" -- M = the S0 x S0-pixel cells (= positions of delta_0), N = the S0^2 pixels of a cell,
    //      K = 3 x 3 neighbouring cells x F0 channels with zero weights where a tap does not reach.
    {
        constexpr int S = Net::S0, K = Net::K0, P = Net::P0, cpb = F0 / 16;
        const int KB = a.kb_ps, MT = p1 / 16, rowf = (D0.w + 2 * D0.pad) * D0.cs(), pixf = D0.cs();
        for (int mt = wave; mt < MT; mt += CW) {
            const int cell = 16 * mt + r16, cy = cell / ow0, cx = cell % ow0;
            const float *origin = D0.p + D0.at(cy - 1, cx - 1) + 4 * q;
            auto ga = [&](int kb) -> f4 {
                const int nbr = kb / cpb, cb = (kb - nbr * cpb) * 16, ny = nbr / 3, nx = nbr - 3 * ny;
                return *reinterpret_cast<const f4 *>(origin + ny * rowf + nx * pixf + cb);
            };
            f4 acc[1] = {{0.f, 0.f, 0.f, 0.f}};
            const f4 *const bp[1] = {reinterpret_cast<const f4 *>(wp + a.p_ps) + lane};
            mfma_stream<1, RD>(acc, ga, bp, (size_t)64, 0, KB);
            const int pa = r16 / S, pb = r16 - pa * S;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int cl = 16 * mt + 4 * q + r, iy = (cl / ow0) * S + pa, ix = (cl % ow0) * S + pb, j = iy * W + ix;
                const float t0 = ctx[a.c_yu[0] + j] * acc[0][r];
                a.g[(size_t)u * n + j] = t0 + convt_at<K, S, P, 1>(dyr1, wp + a.w_yr[0], 1, 0, iy, ix);
            }
        }
    }
}

struct ConvLayout {
    ConvArgs a;
    size_t pack_floats;
    int lds_fwd, lds_bwd, lds_fcb;
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
        // 'SAME' padding must be symmetric (true for 8/4, 4/2, 3/1 on the reference's maps)
        if (ph < 0 || pw < 0 || ph % 2 || pw % 2 || ph != pw) return ICNN_BE_EINVAL;
        a.F[l] = f; a.K[l] = k; a.S[l] = s; a.P[l] = ph / 2; a.oh[l] = oh; a.ow[l] = ow;
        if (l > 0) { a.c_gate[l] = o; o += h * w * cin; }
        a.c_yu[l] = o; o += h * w;
        a.c_zu[l] = o; o += oh * ow * f;
        a.w_yu[l] = wo; wo += (long long)k * k * f;
        if (l < 2) { a.w_yr[l] = wo; wo += k * k; a.b_yr[l] = wo; wo += 1; }
        h = oh; w = ow; cin = f;
    }
    a.flat = h * w * cin;
    a.c_gate[3] = o; o += a.flat;
    a.c_zu3 = o; o += a.fch;
    a.c_gate[4] = o; o += a.fch;
    a.c_zu4 = o; o += 1;
    if (o != m.ctx_width) return ICNN_BE_EINVAL;
    a.C = o;
    a.w_fc4 = wo; wo += a.fch;
    wo = (wo + 3) & ~3LL;                                  // packed operands are read as 16-byte fragments
    const int p1 = a.oh[0] * a.ow[0], p2 = a.oh[1] * a.ow[1], p3 = a.oh[2] * a.ow[2];
    // what the implicit-GEMM mapping needs (true for the reference's completion network): channel counts and position
    // counts in multiples of 16, a first layer whose 16-tap k-blocks are whole rows of taps read as aligned quads,
    // a stride-1 last conv layer, kernel sizes that are multiples of the stride elsewhere, borders that cover the padding
    // the kernels are compiled for the reference's layer hyper-parameters (struct Net); the image size is free
    if (a.K[0] != Net::K0 || a.S[0] != Net::S0 || a.F[0] != Net::F0 || a.K[1] != Net::K1 || a.S[1] != Net::S1 ||
        a.F[1] != Net::F1 || a.K[2] != Net::K2 || a.S[2] != Net::S2 || a.F[2] != Net::F2 || a.P[0] != Net::P0 ||
        a.P[1] != Net::P1 || a.P[2] != Net::P2)
        return ICNN_BE_ELIMIT;
    for (int l = 0; l < 3; ++l)
        if (a.F[l] % 16 || (a.oh[l] * a.ow[l]) % 16 || a.P[l] > (l == 0 ? PADI : PADM)) return ICNN_BE_ELIMIT;
    if (a.K[0] % 4 || 16 % a.K[0] || a.S[0] % 4 || a.P[0] != PADI || (a.W + 2 * PADI) % 4 || a.K[0] != 2 * a.S[0] ||
        a.S[0] * a.S[0] != 16 || a.K[1] % a.S[1] || a.S[2] != 1 || a.fch % 64 || a.flat % 64 || p1 > CT || a.n % CT)
        return ICNN_BE_ELIMIT;
    if (a.ow[0] % a.S[1] || a.oh[0] % a.S[1] || (p1 / (a.S[1] * a.S[1])) % 16) return ICNN_BE_ELIMIT;
    const int kb1 = a.K[0] * a.K[0] / 16, kb2 = a.K[1] * a.K[1] * a.F[0] / 16, kb3 = a.K[2] * a.K[2] * a.F[1] / 16;
    const int ts = a.K[1] / a.S[1], kb2t = ts * ts * a.F[1] / 16, kb3t = a.K[2] * a.K[2] * a.F[2] / 16;
    a.kb_ps = 9 * a.F[0] / 16;
    if ((a.flat / 16) % FKS || FT != 64 * FKS) return ICNN_BE_ELIMIT;
    a.p_l1 = wo; wo += (long long)frag_floats(kb1, a.F[0]);
    a.p_l2 = wo; wo += (long long)frag_floats(kb2, a.F[1]);
    a.p_l3 = wo; wo += (long long)frag_floats(kb3, a.F[2]);
    a.p_l3t = wo; wo += (long long)frag_floats(kb3t, a.F[1]);
    for (int c = 0; c < a.S[1] * a.S[1]; ++c) {
        if (c >= 4) return ICNN_BE_ELIMIT;
        a.p_l2t[c] = wo; wo += (long long)frag_floats(kb2t, a.F[0]);
    }
    a.p_ps = wo; wo += (long long)frag_floats(a.kb_ps, 16);
    a.p_fc3 = wo; wo += (long long)frag_floats(a.flat / 16, a.fch);
    a.p_fc3t = wo; wo += (long long)frag_floats(a.fch / 16, a.flat);
    L.pack_floats = (size_t)wo;
    auto r4 = [](int v) { return (v + 3) & ~3; };
    const int simg = r4((a.H + 2 * PADI) * (a.W + 2 * PADI)), sm1 = (a.oh[0] + 2) * (a.ow[0] + 2),
              sm2 = (a.oh[1] + 2) * (a.ow[1] + 2), sm3 = (a.oh[2] + 2) * (a.ow[2] + 2);
    L.lds_fwd = 4 * (2 * simg + 2 * r4(sm1) + 2 * r4(sm2) + r4(map_floats(sm1, a.F[0])) + r4(map_floats(sm2, a.F[1])));
    L.lds_bwd = 4 * (r4(map_floats(sm3, a.F[2])) + r4(map_floats(sm2, a.F[1])) + r4(map_floats(sm1, a.F[0])) + r4(sm2) + r4(sm1));
    L.lds_fcb = 4 * 16 * (a.fch + 8);
    if (L.lds_fwd > 160 * 1024 || L.lds_bwd > 160 * 1024 || L.lds_fcb > 160 * 1024) return ICNN_BE_ELIMIT;
    a.wpack = m.wpack;
    return 0;
}

}  // namespace

size_t conv_pack_floats(const icnn_be_conv_model &m) {
    ConvLayout L{};
    return conv_layout(m, L) == 0 ? L.pack_floats : 0;
}

size_t conv_work_floats(const icnn_be_conv_model &m, int batch) {
    ConvLayout L{};
    if (conv_layout(m, L) != 0 || batch < 0) return 0;
    const ConvArgs &a = L.a;
    const size_t p1 = (size_t)a.oh[0] * a.ow[0] * a.F[0], p2 = (size_t)a.oh[1] * a.ow[1] * a.F[1];
    return (size_t)batch * (p1 + p2 + 2 * (size_t)a.flat + 2 * (size_t)a.fch);
}

// w_yu[l]: 'z{l}_yu/W' [k][k][1][F]; w_yr[l], b_yr[l] (l = 0, 1): 'z{l}_y_red/W' [k][k][1][1], '/b';
// w_zu[l] (l = 1, 2): 'z{l}_zu_proj/W' [k][k][Cin][F]; w_fc3 [flat][fch]; w_fc4 [fch][1]
int conv_pack(const icnn_be_conv_model &m, const float *const *w_yu, const float *const *w_yr,
              const float *const *b_yr, const float *const *w_zu, const float *w_fc3, const float *w_fc4,
              float *out) {
    ConvLayout L{};
    if (int rc = conv_layout(m, L)) return rc;
    const ConvArgs &a = L.a;
    for (size_t i = 0; i < L.pack_floats; ++i) out[i] = 0.f;
    for (int l = 0; l < 3; ++l) {
        const int k = a.K[l], f = a.F[l];
        for (int i = 0; i < k * k * f; ++i) out[a.w_yu[l] + i] = w_yu[l][i];
        if (l < 2) {
            for (int i = 0; i < k * k; ++i) out[a.w_yr[l] + i] = w_yr[l][i];
            out[a.b_yr[l]] = b_yr[l][0];
        }
    }
    for (int i = 0; i < a.fch; ++i) out[a.w_fc4 + i] = w_fc4[i];
    const int F0 = a.F[0], F1 = a.F[1], F2 = a.F[2];
    // forward operands: K index = (ky*K + kx)*Cin + r, exactly the tflearn layout [k][k][Cin][F] read as [K][N]
    pack_frag([&](int kk, int nn) { return w_yu[0][(size_t)kk * F0 + nn]; }, a.K[0] * a.K[0], F0, a.K[0] * a.K[0] / 16,
              out + a.p_l1);
    pack_frag([&](int kk, int nn) { return w_zu[1][(size_t)kk * F1 + nn]; }, a.K[1] * a.K[1] * F0, F1,
              a.K[1] * a.K[1] * F0 / 16, out + a.p_l2);
    pack_frag([&](int kk, int nn) { return w_zu[2][(size_t)kk * F2 + nn]; }, a.K[2] * a.K[2] * F1, F2,
              a.K[2] * a.K[2] * F1 / 16, out + a.p_l3);
    // transposed, stride 1: K index = (tap, f_out), N = c_in
    pack_frag([&](int kk, int nn) { const int tap = kk / F2, f = kk % F2; return w_zu[2][((size_t)tap * F1 + nn) * F2 + f]; },
              a.K[2] * a.K[2] * F2, F1, a.K[2] * a.K[2] * F2 / 16, out + a.p_l3t);
    // transposed, stride S: one operand per parity class (a, b) of the output position, K index = (valid tap, f_out)
    {
        const int K = a.K[1], S = a.S[1], P = a.P[1], TS = K / S;
        for (int cls = 0; cls < S * S; ++cls) {
            const int ry = (cls / S + P) % S, rx = (cls % S + P) % S;
            pack_frag([&](int kk, int nn) {
                const int tap = kk / F1, f = kk % F1, ky = ry + S * (tap / TS), kx = rx + S * (tap % TS);
                return w_zu[1][((size_t)(ky * K + kx) * F0 + nn) * F1 + f];
            }, TS * TS * F1, F0, TS * TS * F1 / 16, out + a.p_l2t[cls]);
        }
    }
    // pixel shuffle of the first layer's transpose: K index = ((ny, nx) neighbour cell, r), N = (pa, pb) pixel of the cell;
    // the pixel (S cy + pa) takes from cell cy + ny - 1 through tap ky = pa + P - S (ny - 1) when that is a tap
    {
        const int K = a.K[0], S = a.S[0], P = a.P[0];
        pack_frag([&](int kk, int nn) {
            const int nbr = kk / F0, r = kk % F0, ny = nbr / 3, nx = nbr % 3, pa = nn / S, pb = nn % S;
            const int ky = pa + P - S * (ny - 1), kx = pb + P - S * (nx - 1);
            if (nbr >= 9 || ky < 0 || ky >= K || kx < 0 || kx >= K) return 0.f;
            return w_yu[0][(size_t)(ky * K + kx) * F0 + r];
        }, 9 * F0, S * S, a.kb_ps, out + a.p_ps);
    }
    pack_frag([&](int kk, int nn) { return w_fc3[(size_t)kk * a.fch + nn]; }, a.flat, a.fch, a.flat / 16, out + a.p_fc3);
    pack_frag([&](int kk, int nn) { return w_fc3[(size_t)nn * a.fch + kk]; }, a.fch, a.flat, a.fch / 16, out + a.p_fc3t);
    return 0;
}

int conv_check_model(const icnn_be_conv_model &m) {
    ConvLayout L{};
    return conv_layout(m, L);
}

int conv_ctx_shape(const icnn_be_conv_model &m, ConvCtxShape &g) {
    ConvLayout L{};
    if (int rc = conv_layout(m, L)) return rc;
    const ConvArgs &a = L.a;
    g.H = a.H; g.W = a.W; g.flat = a.flat; g.fch = a.fch; g.ctx_width = a.C;
    for (int l = 0; l < 3; ++l) {
        g.F[l] = a.F[l]; g.K[l] = a.K[l]; g.S[l] = a.S[l]; g.pad[l] = a.P[l]; g.oh[l] = a.oh[l]; g.ow[l] = a.ow[l];
        g.P[l] = a.oh[l] * a.ow[l];
        g.c_yu[l] = a.c_yu[l]; g.c_zu[l] = a.c_zu[l];
    }
    for (int l = 1; l < 5; ++l) g.c_gate[l] = a.c_gate[l];
    g.c_gate[0] = 0;
    g.c_zu3 = a.c_zu3; g.c_zu4 = a.c_zu4;
    return 0;
}

// makeCvx / proj of the completion model (completion/icnn_ebundle.py:145-146, :190, :248-249) on the packed
// 'z{1..4}_zu_proj/W' operands: every orientation the kernels read (forward, transposed, parity classes), in place
hipError_t launch_conv_clamp(const icnn_be_conv_model &m, int mode, hipStream_t stream) {
    ConvLayout L{};
    if (conv_layout(m, L) != 0) return hipErrorInvalidValue;
    const ConvArgs &a = L.a;
    float *w = const_cast<float *>(m.wpack);
    hipError_t e = launch_clamp(w + a.w_fc4, (size_t)a.fch, mode, stream);                           // z4_zu_proj
    if (e == hipSuccess) e = launch_clamp(w + a.p_l2, (size_t)(a.p_ps - a.p_l2), mode, stream);      // z1, z2 (all forms)
    if (e == hipSuccess) e = launch_clamp(w + a.p_fc3, L.pack_floats - (size_t)a.p_fc3, mode, stream);   // z3 (both forms)
    return e;
}

void set_conv_profile_buffer(long long *) {}      // (the phase profiler belonged to the single-kernel version)

hipError_t launch_conv_fg(const icnn_be_conv_model &m, const float *ctx, const double *y, int batch, float *f,
                          float *g, const int *skip, hipStream_t stream) {
    ConvLayout L{};
    if (conv_layout(m, L) != 0) return hipErrorInvalidValue;
    if (!m.work || m.work_batch < batch) return hipErrorInvalidValue;
    ConvArgs a = L.a;
    a.ctx = ctx; a.y = y; a.f = f; a.g = g; a.skip = skip; a.batch = batch;
    const size_t p1 = (size_t)a.oh[0] * a.ow[0] * a.F[0], p2 = (size_t)a.oh[1] * a.ow[1] * a.F[1];
    float *w = m.work;
    a.a1s = w; w += (size_t)batch * p1;
    a.a2s = w; w += (size_t)batch * p2;
    a.zflat = w; w += (size_t)batch * a.flat;
    a.a4 = w; w += (size_t)batch * a.fch;
    a.d3 = w; w += (size_t)batch * a.fch;
    a.d2 = w;
    hipError_t e = ensure_dynamic_lds(reinterpret_cast<const void *>(conv_fwd_kernel), L.lds_fwd);
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void *>(conv_bwd_kernel), L.lds_bwd);
    if (e == hipSuccess) e = ensure_dynamic_lds(reinterpret_cast<const void *>(conv_fc_bwd_kernel), L.lds_fcb);
    if (e != hipSuccess) return e;
    const int tiles = (batch + 15) / 16, per = FT / 64;
    hipLaunchKernelGGL(conv_fwd_kernel, dim3(batch), dim3(CT), L.lds_fwd, stream, a);
    hipLaunchKernelGGL(conv_fc_fwd_kernel, dim3(a.fch / 16, tiles), dim3(FT), 0, stream, a);
    hipLaunchKernelGGL(conv_fc_bwd_kernel, dim3((a.flat / 16 + per - 1) / per, tiles), dim3(FT), L.lds_fcb, stream, a);
    hipLaunchKernelGGL(conv_bwd_kernel, dim3(batch), dim3(CT), L.lds_bwd, stream, a);
    return hipGetLastError();
}

}  // namespace icnn_be
