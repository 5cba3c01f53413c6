// Column phase and Hessian contraction of one Newton update in ONE pass on the VALU, for wide rows split over several
// waves (NW > 1) and bundles of up to 20 cuts, and (round 4) for one-wave samples with rows of up to 192 columns and bundles
// of up to HV_K1MAX = 8 cuts -- the Bibsonomy shapes: three columns per lane, no partial-sum rows to add up.  There it
// replaces the 8x8 / 16x16 MFMA sweep whose operand gathers (ten times the bundle per update) are what the LDS bandwidth of a
// CU bounds when sixteen samples share it: headline solve 1.195 -> 1.075 ms, 512 x 10 0.68 -> 0.61 ms, 128 x 10 0.52 -> 0.48 ms.
//
// Why not the MFMA sweep of be_dual_dev.h here: v_mfma_f64 runs at the vector pipe's float64 rate on gfx950 (78.6 TFLOP/s
// both), so it saves no arithmetic time, it computes the full 8 x 8 block where k (k + 3) / 2 sums are wanted, and every lane
// fetches its own (row, column) operands from LDS: ten times the bundle per update, after the column phase has already
// read it once and parked z, w in LDS behind a barrier.  Here a lane owns its columns for the whole update:
//   a_j = sum_i lam_i A[i][j],  z_j = sigmoid(a_j),  w_j = z_j (1 - z_j)                      (dual :32-33)
//   v[(r, c)] += A[r][j] (A[c][j] w_j)   r <= c < k;        v[k(k+1)/2 + r] += A[r][j] z_j    (dual :35-36)
// with the bundle column in registers, no z / w round trip and no barrier in between.  The k (k + 3) / 2 per-lane sums are
// then reduced over the wave by a TRANSPOSING butterfly: at every level a lane hands half of its values to its partner and
// receives the partner's share of the half it keeps, so six levels cost about as many exchanges as there are values (a plain
// butterfly: six per value) and leave value `idx` in exactly one lane, which writes it to the wave's row of partial sums.
// Levels: lane bit 5 with v_permlane32_swap, bit 4 with v_permlane16_swap (a swap IS the exchange of the two halves: two
// instructions and an add per float64), bits 3..0 with DPP row_mirror / row_half_mirror / quad permutes.
// The summation tree is fixed (columns of a lane in order, then the butterfly, then the waves in order): deterministic, and
// the same in every kernel that uses it.
// (included by be_dual_dev.h inside its namespaces)
#pragma once

constexpr int HV_KMAX = 20;                                  // largest bundle the pass is instantiated for
__host__ __device__ constexpr int hv_cap(int K) { return K <= 12 ? 40 : (K <= 16 ? 24 : 16); }   // sums a lane accumulates and reduces at a time
// Value e of a K-cut system: the upper triangle column by column -- (0,0), (0,1), (1,1), (0,2), .. : e = c (c + 1) / 2 + r --,
// then the K entries of A z.  A chunk of consecutive values touches few columns c.
__host__ __device__ constexpr int hv_wcol(int e) {
    int c = 0;
    while ((c + 1) * (c + 2) / 2 <= e) ++c;
    return c;
}
__host__ __device__ constexpr int hv_wrow(int e) { return e - hv_wcol(e) * (hv_wcol(e) + 1) / 2; }
__host__ __device__ constexpr int hv_nv(int K, bool hess) { return K * (K + 1) / 2 + (hess ? K : 0); }
__host__ __device__ constexpr int hv_chunk_len(int nv, int cap) { return (nv + (nv + cap - 1) / cap - 1) / ((nv + cap - 1) / cap); }
// bundles of up to 7 cuts have their own instance each, larger ones share the next even size (absent rows read the zero row)
__host__ __device__ constexpr int hv_padded(int k) { return k <= 7 ? k : (k + 1) & ~1; }
__host__ __device__ constexpr int hv_pitch(int k) { return (hv_nv(hv_padded(k), true) + 3) & ~3; }      // partial sums per wave
__host__ __device__ constexpr int hv_count(int nv, int level) {
    int c = nv;
    for (int l = 0; l < level; ++l) c = (c + 1) / 2;
    return c;
}

__device__ __forceinline__ void hv_swap32(double &x, double &y) {
    const auto lo = __builtin_amdgcn_permlane32_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}
__device__ __forceinline__ void hv_swap16(double &x, double &y) {
    const auto lo = __builtin_amdgcn_permlane16_swap((unsigned)__double2loint(x), (unsigned)__double2loint(y), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap((unsigned)__double2hiint(x), (unsigned)__double2hiint(y), false, false);
    x = __hiloint2double((int)hi[0], (int)lo[0]);
    y = __hiloint2double((int)hi[1], (int)lo[1]);
}

// One level of the transposing butterfly on the first C values of v: lanes whose bit is clear keep v[0 .. H), the others
// v[H .. C) (renumbered from 0), H = ceil(C / 2); both receive the partner's share.  A slot the upper half does not have is
// carried as garbage that no valid lane ever consumes (hv_index).
template <int NV, int LEVEL, int CTRL, int BIT>
__device__ __forceinline__ void hv_level(double (&v)[NV], int lane) {
    constexpr int C = hv_count(NV, LEVEL), H = (C + 1) / 2;
    const bool up = (lane & BIT) != 0;
#pragma unroll
    for (int i = 0; i < H; ++i) {
        double x = v[i], y = H + i < C ? v[H + i] : 0.0;
        if constexpr (BIT == 32) {
            hv_swap32(x, y);
            v[i] = x + y;
        } else if constexpr (BIT == 16) {
            hv_swap16(x, y);
            v[i] = x + y;
        } else {
            const double send = up ? x : y, keep = up ? y : x;
            v[i] = keep + dpp_move<CTRL>(send);
        }
    }
}
template <int NV>
__device__ __forceinline__ void hv_transpose_reduce(double (&v)[NV], int lane) {
    hv_level<NV, 0, 0, 32>(v, lane);
    hv_level<NV, 1, 0, 16>(v, lane);
    hv_level<NV, 2, 0x140, 8>(v, lane);      // row_mirror: lane ^ 15
    hv_level<NV, 3, 0x141, 4>(v, lane);      // row_half_mirror: lane ^ 7
    hv_level<NV, 4, 0x1B, 2>(v, lane);       // quad_perm [3,2,1,0]: lane ^ 3
    hv_level<NV, 5, 0xB1, 1>(v, lane);       // quad_perm [1,0,3,2]: lane ^ 1
}
// which value lane `lane` holds in v[0] after hv_transpose_reduce (-1: none).  The split points are those of the
// unrolled code (the static counts C -> ceil(C / 2)), whatever the lane's own number of live values.
__device__ __forceinline__ int hv_index(int nv, int lane) {
    int base = 0, live = nv, C = nv;
#pragma unroll
    for (int b = 5; b >= 0; --b) {
        const int H = (C + 1) / 2;
        if ((lane >> b) & 1) { base += H; live = live > H ? live - H : 0; }
        else live = live < H ? live : H;
        C = H;
    }
    return live >= 1 ? base : -1;
}

// The fused pass for a bundle of k <= K cuts (K = hv_padded(k)): this wave's partial sums -> Pw[0 .. K (K + 3) / 2).  `tid` in
// 0 .. 64 NW - 1 owns the columns tid, tid + 64 NW, .. (at most four: n_pad <= 256 NW); lam in row layout (lane i < k holds
// lam_i, 0 beyond; every wave alike).  The sums are formed and reduced hv_cap(K) at a time.  Where a lane's bundle column comes
// from:
//   LR = 0   every row from `As` (the staged bundle: LDS, or device memory in the GLB rounds).  Up to HV_KREG cuts the column
//            stays in registers for the whole pass; larger bundles re-read it per chunk (coalesced rows -- still a fraction of the
//            sweep's operand gathers; out of device memory one column at a time, which is slow: see LR > 0)
//   LR > 0   split staging (dual_step_wide_kernel): rows 0 .. LR - 1 from the LDS mirror `AsL` (in registers up to HV_KREG
//            cuts, else re-read per chunk); the K - LR <= 8 younger rows from `As` in device memory, loaded ONCE per pass (all
//            loads in flight) and kept in registers.
// Rows k .. K - 1 of a padded instance are zeros (the zero row `zrow` of `As`).
// With HESS = false (rank test): the Gram matrix, v[(r, c)] += A[r][j] A[c][j].
constexpr int HV_KREG = 10;
constexpr int HV_K1MAX = 8;     // one wave per sample: the pass up to this many cuts, the MFMA sweep beyond
template <typename CutT, int K, int NC, int E0, int EN, int NV, bool HESS, int KR, typename Load, typename PP>
__device__ __forceinline__ void hv_chunks(const CutT (&av)[KR][NC], Load load, const double (&z)[NC], const double (&w)[NC],
                                          int lane, PP Pw) {
    if constexpr (E0 < NV) {
        constexpr int N = E0 + EN <= NV ? EN : NV - E0, T = K * (K + 1) / 2;
        // rows this chunk needs: 0 .. RMAX - 1 (the triangle's columns up to the chunk's last one; all of them for A z)
        constexpr int LAST = E0 + N - 1, RMAX = LAST >= T ? K : hv_wcol(LAST) + 1;
        double v[N];
#pragma unroll
        for (int e = 0; e < N; ++e) v[e] = 0.0;
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            double ad[RMAX];
#pragma unroll
            for (int i = 0; i < RMAX; ++i) ad[i] = (double)(KR == K ? av[KR == K ? i : 0][c] : load(i, c));
#pragma unroll
            for (int e = 0; e < N; ++e) {
                const int ge = E0 + e;                                // (compile-time after unrolling)
                if (ge < T) v[e] = __builtin_fma(ad[hv_wrow(ge)], ad[hv_wcol(ge)] * w[c], v[e]);
                else v[e] = __builtin_fma(ad[ge - T], z[c], v[e]);
            }
            if (KR != K && RMAX > 12) asm volatile("" ::: "memory");  // one re-read column of a large bundle at a time (registers)
        }
        hv_transpose_reduce<N>(v, lane);
        const int idx = hv_index(N, lane);
        if (idx >= 0) Pw[E0 + idx] = v[0];
        if (KR != K) asm volatile("" ::: "memory");      // the next chunk re-reads its rows: no loads kept live across chunks
        hv_chunks<CutT, K, NC, E0 + EN, EN, NV, HESS, KR>(av, load, z, w, lane, Pw);
    }
}

template <typename CutT, int K, int NW, bool HESS, int LR, typename AP, typename LP, typename PP>
__device__ __forceinline__ void hv_column_pass(AP As, LP AsL, int ldA, int k, int zrow, int n, int n_pad, int tid, double lam, PP Pw) {
    constexpr int NT = 64 * NW, NC = NW == 1 ? 3 : 4, NV = hv_nv(K, HESS);      // (one wave: rows of up to 192 columns)
    constexpr bool SPLIT = LR > 0;
    constexpr int KR = K <= HV_KREG ? K : 1, KG = SPLIT && K > LR ? K - LR : 1;
    static_assert(!SPLIT || K - LR <= 8, "split staging keeps at most eight device-memory rows in registers");
    asm volatile("" : "+v"(tid));       // opaque: the 4 K load addresses are invariant in the Newton loop and would be hoisted
    const int lane = tid & 63;          // out of it for every instance of this pass, then spilled (cf. thread_id(), be_common.h)
    int jc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) jc[c] = tid + c * NT < n_pad ? tid + c * NT : n_pad - 1;
    CutT ag[KG][NC];                                        // SPLIT: the device-memory rows, once per pass
    if constexpr (SPLIT && K > LR) {
#pragma unroll
        for (int i = 0; i < KG; ++i)
#pragma unroll
            for (int c = 0; c < NC; ++c) ag[i][c] = As[(LR + i < k ? LR + i : zrow) * ldA + jc[c]];
    }
    auto load = [&](int i, int c) -> CutT {
        if constexpr (SPLIT) {
            if (i >= LR) return ag[i - LR >= 0 && i - LR < KG ? i - LR : 0][c];
            const CutT v = AsL[(i < k ? i : 0) * ldA + jc[c]];
            return i < k ? v : (CutT)0;                     // (rows beyond the bundle hold older data in the mirror)
        } else {
            // absent rows of a padded instance: the zero row behind the staged bundle where the sample keeps its own constant
            // rows (several waves per sample); one-wave samples share one pair of constant rows per workgroup in the
            // persistent kernels -- zero by a select on an in-bounds read
            if constexpr (NW > 1) return As[(i < k ? i : zrow) * ldA + jc[c]];
            const CutT v = As[(i < k ? i : 0) * ldA + jc[c]];
            return i < k ? v : (CutT)0;
        }
    };
    CutT av[KR][NC];
    double z[NC], w[NC], acc[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) { z[c] = 0.0; w[c] = 1.0; acc[c] = 0.0; }
    if constexpr (KR == K) {
#pragma unroll
        for (int c = 0; c < NC; ++c)
#pragma unroll
            for (int i = 0; i < K; ++i) av[i][c] = load(i, c);
    }
    if (HESS) {
#pragma unroll
        for (int i = 0; i < K; ++i) {                     // plain i = 0 .. k-1 order, as columns_nc (absent rows: + 0 * 0)
            const double li = bcast(lam, i);
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] += li * (double)(KR == K ? av[KR == K ? i : 0][c] : load(i, c));
            if (KR != K && !SPLIT && (i & 3) == 3) asm volatile("" ::: "memory");      // at most 16 streamed loads in flight
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            z[c] = sigmoid_fast(acc[c]);                      // dual :33 (be_dual_dev.h: fast_exp + reciprocal)
            w[c] = z[c] * (1.0 - z[c]);
        }
    }
#pragma unroll
    for (int c = 0; c < NC; ++c)
        if (tid + c * NT >= n) { z[c] = 0.0; w[c] = 0.0; }        // padding and the clamped re-reads beyond n_pad
    if (KR != K) asm volatile("" ::: "memory");
    // (one wave per sample, kernels held to 128 VGPRs: shorter chunks -- 16, 24 sums -- spill less and measure SLOWER, 1.15 /
    //  1.085 ms against 1.075 ms for the headline solve; so do all instances as functions, 1.10 ms)
    hv_chunks<CutT, K, NC, 0, hv_chunk_len(NV, hv_cap(K)), NV, HESS, KR>(av, load, z, w, lane, Pw);
}

// Bundles of 8 and more cuts: the pass as a FUNCTION (called once per Newton update: ~2000 instructions, against which the
// call and its callee-saved registers are noise), so that the thirteen instances do not share the caller's register allocation
// -- inlined they pushed the kernels of the small bundles (BASELINE's nIter = 5 never sees more than 6 cuts) from 240 VGPRs
// without spills to 256 with, 3 % of the solve.  Pointers arrive generic and are restored to what the caller knows: the staged
// bundle in LDS, or (GSRC) in device memory; mirror and partial sums in LDS.
template <typename CutT, int K, int NW, bool HESS, int LR, bool GSRC>
__device__ __noinline__ void hv_column_pass_fn(const CutT *As_, const CutT *AsL_, int ldA, int k, int zrow, int n, int n_pad,
                                               int tid, double lam, double *Pw_) {
    typedef const __attribute__((address_space(3))) CutT *LdsCut;
    typedef const __attribute__((address_space(1))) CutT *GlbCut;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    ldA = uni(ldA); k = uni(k); zrow = uni(zrow); n = uni(n); n_pad = uni(n_pad);
    if constexpr (GSRC) hv_column_pass<CutT, K, NW, HESS, LR>((GlbCut)As_, (LdsCut)AsL_, ldA, k, zrow, n, n_pad, tid, lam, (LdsDbl)Pw_);
    else hv_column_pass<CutT, K, NW, HESS, LR>((LdsCut)As_, (LdsCut)AsL_, ldA, k, zrow, n, n_pad, tid, lam, (LdsDbl)Pw_);
}

// ALLFN: every instance as a function (the 32-slot one-wave kernels: bundles of up to 8 cuts are the first rounds of a long
// solve there, and the inlined instances made the Newton loop of ALL its updates spill -- configs[3] moved 5.4 GB per launch
// against 3.0 GB; the 16-slot kernels, whose bundles mostly ARE that small, are faster with the instances inlined)
template <typename CutT, int NW, bool HESS, int LR, bool GSRC, bool ALLFN = false>
__device__ __forceinline__ void hv_column_pass_k(const CutT *As, const CutT *AsL, int ldA, int k, int zrow, int n, int n_pad,
                                                 int tid, double lam, double *Pw) {
    if constexpr (ALLFN) {
        switch (hv_padded(k)) {
        case 2: hv_column_pass_fn<CutT, 2, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
        case 3: hv_column_pass_fn<CutT, 3, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
        case 4: hv_column_pass_fn<CutT, 4, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
        case 5: hv_column_pass_fn<CutT, 5, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
        case 6: hv_column_pass_fn<CutT, 6, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
        case 7: hv_column_pass_fn<CutT, 7, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
        default: hv_column_pass_fn<CutT, 8, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
        }
        return;
    }
    switch (hv_padded(k)) {                                // wave-uniform
    case 2: hv_column_pass<CutT, 2, NW, HESS, LR>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 3: hv_column_pass<CutT, 3, NW, HESS, LR>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 4: hv_column_pass<CutT, 4, NW, HESS, LR>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 5: hv_column_pass<CutT, 5, NW, HESS, LR>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 6: hv_column_pass<CutT, 6, NW, HESS, LR>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 7: hv_column_pass<CutT, 7, NW, HESS, LR>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 8: hv_column_pass_fn<CutT, 8, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 10: hv_column_pass_fn<CutT, 10, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 12: hv_column_pass_fn<CutT, 12, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 14: hv_column_pass_fn<CutT, 14, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 16: hv_column_pass_fn<CutT, 16, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    case 18: hv_column_pass_fn<CutT, 18, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    default: hv_column_pass_fn<CutT, 20, NW, HESS, LR, GSRC>(As, AsL, ldA, k, zrow, n, n_pad, tid, lam, Pw); break;
    }
}

// The same sums with GIVEN column weights (one wave per sample, rows of up to 192 columns, bundles of up to HV_K1MAX cuts):
//     v[(r, c)] = sum_j A[r][j] A[c][j] w[j],   v[k (k + 1) / 2 + r] = sum_j A[r][j] z[j]
// -- the interior-point variant's M = G Hinv G^T and G Hinv ry (lib/bundle_entropy.py:41, :46) with w = Hinv, z = Hinv ry from
// its column buffers.  `Pw` may alias `zs`: every lane has its w, z in registers before the first sum is stored.
template <typename CutT, int K>
__device__ __noinline__ void hv_weighted_pass_fn(const CutT *As_, int ldA, int k, int n, int n_pad, const double *ws_,
                                                 const double *zs_, double *Pw_) {
    typedef const __attribute__((address_space(3))) CutT *LdsCut;
    typedef const __attribute__((address_space(3))) double *LdsCDbl;
    typedef __attribute__((address_space(3))) double *LdsDbl;
    LdsCut As = (LdsCut)As_;
    LdsCDbl ws = (LdsCDbl)ws_, zs = (LdsCDbl)zs_;
    LdsDbl Pw = (LdsDbl)Pw_;
    ldA = uni(ldA); k = uni(k); n = uni(n); n_pad = uni(n_pad);
    constexpr int NC = 3, NV = hv_nv(K, true);
    const int lane = lane_id();
    CutT av[K][NC];
    double z[NC], w[NC];
#pragma unroll
    for (int c = 0; c < NC; ++c) {
        const int j = lane + 64 * c, jc = j < n_pad ? j : n_pad - 1;
#pragma unroll
        for (int i = 0; i < K; ++i) {
            const CutT v = As[(i < k ? i : 0) * ldA + jc];
            av[i][c] = i < k ? v : (CutT)0;
        }
        const double wv = ws[jc], zv = zs[jc];
        w[c] = j < n ? wv : 0.0;
        z[c] = j < n ? zv : 0.0;
    }
    auto load = [&](int, int) -> CutT { return (CutT)0; };          // (unused: the whole column is in registers)
    hv_chunks<CutT, K, NC, 0, hv_chunk_len(NV, hv_cap(K)), NV, true, K>(av, load, z, w, lane, Pw);
}
template <typename CutT>
__device__ __forceinline__ void hv_weighted_pass_k(const CutT *As, int ldA, int k, int n, int n_pad, const double *ws,
                                                   const double *zs, double *Pw) {
    switch (hv_padded(k)) {                                // wave-uniform
    case 2: hv_weighted_pass_fn<CutT, 2>(As, ldA, k, n, n_pad, ws, zs, Pw); break;
    case 3: hv_weighted_pass_fn<CutT, 3>(As, ldA, k, n, n_pad, ws, zs, Pw); break;
    case 4: hv_weighted_pass_fn<CutT, 4>(As, ldA, k, n, n_pad, ws, zs, Pw); break;
    case 5: hv_weighted_pass_fn<CutT, 5>(As, ldA, k, n, n_pad, ws, zs, Pw); break;
    case 6: hv_weighted_pass_fn<CutT, 6>(As, ldA, k, n, n_pad, ws, zs, Pw); break;
    case 7: hv_weighted_pass_fn<CutT, 7>(As, ldA, k, n, n_pad, ws, zs, Pw); break;
    default: hv_weighted_pass_fn<CutT, 8>(As, ldA, k, n, n_pad, ws, zs, Pw); break;
    }
}

// Sum the NW rows of partial sums (waves in order) into a k x (k + 1) system H | A z (k x k Gram matrix with HESS = false),
// both triangles, with the threads `t`, t + nthreads, ..: a wave into its OWN copy (no second barrier: the copy is read by the
// wave that wrote it), or the whole sample into the shared one.  hv_entry maps entry e of the system to its partial sum and its
// place; the caller of the per-update gather computes its thread's first entry once per round (an integer division by k + 1).
struct HvEntry { int idx, dst; };
template <bool HESS>
__device__ __forceinline__ HvEntry hv_entry(int e, int k, int HP) {
    const int nc = HESS ? k + 1 : k, K = hv_padded(k), T = K * (K + 1) / 2;
    const int r = e / nc, c = e - r * nc;
    return HvEntry{c == k ? T + r : (r <= c ? c * (c + 1) / 2 + r : r * (r + 1) / 2 + c), r * HP + c};
}
template <int NW, bool HESS>
__device__ __forceinline__ void hv_gather(const double *P, int pitch, double *Hw, int HP, int k, int t, int nthreads, HvEntry first) {
    const int total = k * (HESS ? k + 1 : k);
    HvEntry en = first;
    for (int e = t; e < total; e += nthreads) {
        if (e != t) en = hv_entry<HESS>(e, k, HP);
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += P[w * pitch + en.idx];
        Hw[en.dst] = acc;
    }
    sample_sync<1>();
}
