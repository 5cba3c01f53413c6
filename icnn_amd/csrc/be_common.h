// Shared device helpers for the bundle-entropy kernels (gfx950, wave64).
#pragma once
// Diagnostic cycle-counter laps (icnn_be_debug_profile*, include/icnn_be.h) are compiled in with -DICNN_BE_PROF=1 only: the
// profiling variant of the library (icnn_amd/build.py --prof).  In the production kernels even the never-taken branches cost
// registers around the 128-VGPR ceiling: 1.041 against 1.022 ms for the benchmark solve on one box.
#ifndef ICNN_BE_PROF
#define ICNN_BE_PROF 0
#endif
#define ICNN_BE_PROF_ON(p) (ICNN_BE_PROF && (p))
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace icnn_be {

typedef double d4 __attribute__((ext_vector_type(4)));
typedef float f4 __attribute__((ext_vector_type(4)));

constexpr int WAVE = 64;

// threadIdx.x through an opaque move.  Inside the round loop of a persistent kernel everything derived from the thread
// index (lane, wave, MFMA fragment coordinates, LDS offsets) is loop invariant: the optimiser hoists it in front of
// the loop, and since each phase alone needs the whole register budget the hoisted values are then spilled to
// scratch memory and reloaded at every use.  Read opaquely, the two or three ALU instructions stay where they are used.
__device__ __forceinline__ int thread_id() {
    int t = threadIdx.x;
    asm volatile("" : "+v"(t));
    return t;
}

// Lane index from the mask counters: needs no work-item id, so a NON-INLINED function that uses it does not make its callers
// keep -- or, at 128 VGPRs, spill and reload before every call -- the packed work-item id the calling convention passes in v31.
__device__ __forceinline__ int lane_id() { return (int)__builtin_amdgcn_mbcnt_hi(~0u, __builtin_amdgcn_mbcnt_lo(~0u, 0u)); }

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}
__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v = fmax(v, __shfl_xor(v, o));
    return v;
}
__device__ __forceinline__ float wave_sum_f(float v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o);
    return v;
}

// Butterfly exchange inside groups of 8 lanes with DPP (no LDS traffic): steps 1 and 2 are
// quad permutes, step 4 is row_half_mirror (lane i <- lane 7-i of its group of eight; after steps
// 1 and 2 every lane of a quad holds the quad's sum, so this adds the other quad's sum).
template <int CTRL> __device__ __forceinline__ float dpp_move(float x) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, x), CTRL, 0xF, 0xF, true));
}
template <int CTRL> __device__ __forceinline__ double dpp_move(double x) {
    const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(x), CTRL, 0xF, 0xF, true);
    const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(x), CTRL, 0xF, 0xF, true);
    return __hiloint2double(hi, lo);
}
template <typename T> __device__ __forceinline__ T sum8_numpy_order(T r) {
    r = r + dpp_move<0xB1>(r);     // quad_perm [1,0,3,2]: (r0+r1), (r2+r3), ...
    r = r + dpp_move<0x4E>(r);     // quad_perm [2,3,0,1]: (r0+r1)+(r2+r3), (r4+r5)+(r6+r7)
    r = r + dpp_move<0x141>(r);    // row_half_mirror:     ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7))
    return r;
}

// NumPy's pairwise summation (numpy/_core/src/umath/loops_utils.h.src): leaves of
// at most 128 elements, each summed with 8 strided accumulators combined as
// ((r0+r1)+(r2+r3))+((r4+r5)+(r6+r7)) plus a sequential tail; leaves combined
// left+right up a binary tree whose split is n/2 rounded down to a multiple of 8.
// The tree depends on n only, so the host flattens it once per launch.
constexpr int PW_MAX_LEAVES = 64;   // n <= 8192
constexpr int PW_MAX_PROG = 2 * PW_MAX_LEAVES;
struct PairwisePlan {
    int n_leaves;
    int n_prog;
    int depth;                       // >= 0: the tree is a perfect binary tree of this depth (n_leaves = 2^depth <= 16, leaves in
                                     // order): the combine step is `depth` butterfly steps inside a 16-lane row; -1: walk `prog`
    short leaf_start[PW_MAX_LEAVES];
    short leaf_len[PW_MAX_LEAVES];
    signed char prog[PW_MAX_PROG];   // postfix: >= 0 push leaf, -1 add the two on top
};

inline void pw_build_rec(PairwisePlan &p, int start, int n, int level) {
    if (n <= 128) {
        p.leaf_start[p.n_leaves] = (short)start;
        p.leaf_len[p.n_leaves] = (short)n;
        p.prog[p.n_prog++] = (signed char)p.n_leaves;
        if (p.n_leaves == 0) p.depth = level;
        else if (p.depth != level) p.depth = -1;
        p.n_leaves++;
        return;
    }
    int n2 = n / 2;
    n2 -= n2 % 8;
    pw_build_rec(p, start, n2, level + 1);
    pw_build_rec(p, start + n2, n - n2, level + 1);
    p.prog[p.n_prog++] = -1;
}
inline bool pw_build(PairwisePlan &p, int n) {
    p.n_leaves = 0;
    p.n_prog = 0;
    p.depth = -1;
    if (n > 128 * PW_MAX_LEAVES / 2) return false;
    pw_build_rec(p, 0, n, 0);
    if (p.depth > 4 || p.n_leaves != (1 << (p.depth < 0 ? 0 : p.depth))) p.depth = -1;
    return true;
}

// Synchronisation of the NW waves that work on one sample.  NW == 1: the sample belongs to a single wave64,
// whose LDS operations execute in program order, so only the compiler has to be kept from reordering them --
// no s_barrier.  That also lets several single-wave samples share one workgroup (be_fused.hip), where a
// workgroup barrier inside the per-sample code would deadlock on its data-dependent control flow.
template <int NW>
__device__ __forceinline__ void sample_sync() {
    if (NW == 1) {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    } else {
        __syncthreads();
    }
}

// Sum `rows` vectors of length plan-n held in LDS in NumPy's order.  `elem(r, j)`
// returns element j of vector r as T.  Result r is left in out[r] (LDS, T).
// `leafbuf` is LDS scratch of rows * n_leaves T's.  All NW waves of the sample participate (thread index
// `tid` in 0 .. 64 NW - 1); all control flow is uniform over them.  Eight lanes (one per accumulator) own a leaf.
template <int NW, typename T, typename Plan, typename Elem>     // Plan: PairwisePlan in whatever address space
__device__ void np_pairwise_rows(const Plan &plan, int rows, Elem elem, T *leafbuf, T *out, int tid) {
    const int lane = tid;                              // rows <= 64: the combine step runs in wave 0
    const int grp = tid >> 3, c = tid & 7;
    const int ngrp = (64 * NW) >> 3;
    const int tasks = rows * plan.n_leaves;
    for (int base = 0; base < tasks; base += ngrp) {
        const int task = base + grp;
        const bool live = task < tasks;
        const int r = live ? task / plan.n_leaves : 0;
        const int lf = live ? task - r * plan.n_leaves : 0;
        const int st = plan.leaf_start[lf], len = plan.leaf_len[lf];
        T res = (T)0;
        if (len < 8) {
            for (int i = 0; i < len; ++i) res = res + elem(r, st + i);
        } else {
            const int body = len - (len % 8);
            T acc = elem(r, st + c);
            int i = 8;
            for (; i + 32 <= body; i += 32) {          // four LDS reads in flight, added in order
                const T e0 = elem(r, st + i + c), e1 = elem(r, st + i + 8 + c), e2 = elem(r, st + i + 16 + c),
                        e3 = elem(r, st + i + 24 + c);
                acc = acc + e0;
                acc = acc + e1;
                acc = acc + e2;
                acc = acc + e3;
            }
            for (; i < body; i += 8) acc = acc + elem(r, st + i + c);
            res = sum8_numpy_order(acc);
            for (i = body; i < len; ++i) res = res + elem(r, st + i);
        }
        if (live && c == 0) leafbuf[task] = res;
    }
    sample_sync<NW>();
    if (plan.depth >= 0) {
        // perfect tree: value (vector r, leaf l) in thread r * n_leaves + l; left + right at every node is the adjacent-pair
        // butterfly (quad permutes, row_half_mirror, row_mirror -- IEEE addition commutes, so which lane adds is immaterial)
        const int depth = plan.depth, L = plan.n_leaves, total = rows * L;
        for (int base = 0; base < total; base += 64 * NW) {
            const int t = base + tid;
            T v = t < total ? leafbuf[t] : (T)0;
            if (depth >= 1) v = v + dpp_move<0xB1>(v);
            if (depth >= 2) v = v + dpp_move<0x4E>(v);
            if (depth >= 3) v = v + dpp_move<0x141>(v);
            if (depth >= 4) v = v + dpp_move<0x140>(v);
            if (t < total && (t & (L - 1)) == 0) out[t >> depth] = v;
        }
        sample_sync<NW>();
        return;
    }
    // combine: lane r walks the postfix program for vector r with a private stack
    // held in registers (depth <= 8 because leaves are >= 64 wide for n > 128)
    if (lane < rows) {
        T s0 = 0, s1 = 0, s2 = 0, s3 = 0, s4 = 0, s5 = 0, s6 = 0, s7 = 0;
        int sp = 0;
        for (int i = 0; i < plan.n_prog; ++i) {
            const int tok = plan.prog[i];
            if (tok >= 0) {
                const T v = leafbuf[lane * plan.n_leaves + tok];
                switch (sp) {
                case 0: s0 = v; break; case 1: s1 = v; break; case 2: s2 = v; break; case 3: s3 = v; break;
                case 4: s4 = v; break; case 5: s5 = v; break; case 6: s6 = v; break; default: s7 = v; break;
                }
                ++sp;
            } else {
                switch (sp) {
                case 2: s0 = s0 + s1; break; case 3: s1 = s1 + s2; break; case 4: s2 = s2 + s3; break;
                case 5: s3 = s3 + s4; break; case 6: s4 = s4 + s5; break; case 7: s5 = s5 + s6; break;
                default: s6 = s6 + s7; break;
                }
                --sp;
            }
        }
        out[lane] = s0;
    }
    sample_sync<NW>();
}

}  // namespace icnn_be
