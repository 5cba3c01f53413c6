// Column phase and Hessian contraction of one Newton update in ONE pass on the VALU, for wide rows split over several
// waves (NW > 1) and small bundles (k <= 7 cuts).
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

constexpr int HV_KMAX = 7;                                   // k + 1 <= 8: the range of contract_mfma_8x8
constexpr int HV_PITCH = 36;                                 // >= KMAX (KMAX + 3) / 2 = 35 partial sums per wave
__host__ __device__ constexpr int hv_tri(int K, int r, int c) { return r * K - r * (r - 1) / 2 + (c - r); }   // r <= c
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

// The fused pass for a bundle of exactly K cuts: this wave's partial sums -> Pw[0 .. K (K + 3) / 2).  `tid` in 0 .. 64 NW - 1
// owns the columns tid, tid + 64 NW, ..; lam in row layout (lane i < K holds lam_i, every wave alike).  With HESS = false
// (rank test): the Gram matrix, v[(r, c)] += A[r][j] A[c][j], K (K + 1) / 2 sums.
template <typename CutT, int K, int NW, bool HESS>
__device__ __forceinline__ void hv_column_pass(const CutT *As, int ldA, int n, int n_pad, int tid, double lam, double *Pw) {
    constexpr int NT = 64 * NW, NC = 4, T = K * (K + 1) / 2, NV = HESS ? T + K : T;
    const int lane = tid & 63;
    double v[NV];
#pragma unroll
    for (int e = 0; e < NV; ++e) v[e] = 0.0;
    double li[K];
#pragma unroll
    for (int i = 0; i < K; ++i) li[i] = HESS ? bcast(lam, i) : 0.0;
    for (int j0 = 0; j0 < n_pad; j0 += NC * NT) {
        CutT av[K][NC];
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = j0 + tid + c * NT, jc = j < n_pad ? j : n_pad - 1;
#pragma unroll
            for (int i = 0; i < K; ++i) av[i][c] = As[i * ldA + jc];
        }
        double acc[NC];
        if (HESS) {
#pragma unroll
            for (int c = 0; c < NC; ++c) acc[c] = 0.0;
#pragma unroll
            for (int i = 0; i < K; ++i)                   // plain i = 0 .. K-1 order, as columns_nc
#pragma unroll
                for (int c = 0; c < NC; ++c) acc[c] += li[i] * (double)av[i][c];
        }
#pragma unroll
        for (int c = 0; c < NC; ++c) {
            const int j = j0 + tid + c * NT;
            double z = 0.0, w = 1.0;
            if (HESS) {
                z = 1.0 / (1.0 + exp(-acc[c]));
                w = z * (1.0 - z);
            }
            if (j >= n) { z = 0.0; w = 0.0; }            // padding and the clamped re-reads beyond n_pad
            double ad[K];
#pragma unroll
            for (int i = 0; i < K; ++i) ad[i] = (double)av[i][c];
#pragma unroll
            for (int i = 0; i < K; ++i) {
                const double bw = ad[i] * w;
#pragma unroll
                for (int r = 0; r <= i; ++r) v[hv_tri(K, r, i)] = __builtin_fma(ad[r], bw, v[hv_tri(K, r, i)]);
            }
            if (HESS) {
#pragma unroll
                for (int r = 0; r < K; ++r) v[T + r] = __builtin_fma(ad[r], z, v[T + r]);
            }
        }
    }
    hv_transpose_reduce<NV>(v, lane);
    const int idx = hv_index(NV, lane);
    if (idx >= 0) Pw[idx] = v[0];
}

template <typename CutT, int NW, bool HESS>
__device__ __forceinline__ void hv_column_pass_k(const CutT *As, int ldA, int k, int n, int n_pad, int tid, double lam,
                                                 double *Pw) {
    switch (k) {                                           // wave-uniform
    case 2: hv_column_pass<CutT, 2, NW, HESS>(As, ldA, n, n_pad, tid, lam, Pw); break;
    case 3: hv_column_pass<CutT, 3, NW, HESS>(As, ldA, n, n_pad, tid, lam, Pw); break;
    case 4: hv_column_pass<CutT, 4, NW, HESS>(As, ldA, n, n_pad, tid, lam, Pw); break;
    case 5: hv_column_pass<CutT, 5, NW, HESS>(As, ldA, n, n_pad, tid, lam, Pw); break;
    case 6: hv_column_pass<CutT, 6, NW, HESS>(As, ldA, n, n_pad, tid, lam, Pw); break;
    default: hv_column_pass<CutT, 7, NW, HESS>(As, ldA, n, n_pad, tid, lam, Pw); break;
    }
}

// Every wave sums the NW rows of partial sums (waves in order) into its OWN copy of the k x (k + 1) system H | A z
// (k x k Gram matrix with HESS = false), both triangles: no second barrier, the copy is read by the wave that wrote it.
template <int NW, bool HESS>
__device__ __forceinline__ void hv_gather(const double *P, double *Hw, int HP, int k, int lane) {
    const int nc = HESS ? k + 1 : k, T = k * (k + 1) / 2;
    if (lane < k * nc) {
        const int r = lane / nc, c = lane - r * nc;
        const int idx = c == k ? T + r : (r <= c ? hv_tri(k, r, c) : hv_tri(k, c, r));
        double acc = 0.0;
#pragma unroll
        for (int w = 0; w < NW; ++w) acc += P[w * HV_PITCH + idx];
        Hw[r * HP + c] = acc;
    }
    sample_sync<1>();
}
