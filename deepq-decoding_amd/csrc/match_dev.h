// Exact minimum-weight referee without tables ("matching referee"), device side: shared by match.hip (dq_match_decode) and env_big.hip (the
// referee call inside the step of lattices with d >= 9, where the look-up referee's 2^((d^2-1)/2) entries per component no longer fit).
//
// Definition = the look-up referee's (env.hip bfs_* / oracle/referee.py): for one Pauli component predict class 1 iff the lightest error
// with the given syndrome and logical class 1 is STRICTLY lighter than the lightest one with class 0.  Algorithm = oracle/matching_referee.py
// (the numpy restatement this file is tested against bit for bit): the component is a graph (plaquettes = nodes, data qubits = edges, one-
// ended edges go to the boundary, every edge carries the qubit's logical bit); an error is a set of paths pairing the defects with each other
// or the boundary, so  w_c(D) = min over pairings of the summed shortest-path lengths whose classes XOR to c  -- minimum-weight perfect
// matching with class bookkeeping, solved exactly by dynamic programming over the subsets of the (few) defects:
//   f[S][c], S a subset of the defects in index order, one distinguished defect u of S (the oracle: the lowest; here: the highest) ->
//   boundary (class c') or -> partner v in S (class c')
//   w_c = min(f[D][c], f[D][c^1] + w_1(0))        (w_1(0): lightest defect-free class-1 error = the code distance)
// More than DQ_MATCH_MAX_DEFECTS defects: the lowest DQ_MATCH_MAX_DEFECTS are solved exactly, every further one goes to its nearer boundary
// (ties: the class-0 path) and the result is flagged inexact.
//
// One wavefront per (syndrome, component).  The 2^k x 2 table lives in LDS as bytes (255 = unreachable; any reachable entry is below
// 14 x (d + 1) / 2 + d <= 127 for d <= 15: every defect can be sent to a boundary within (d + 1) / 2 edges, and forcing the other class
// costs at most one more crossing); subsets are visited level by level of their highest defect (both predecessors of S lie below 2^h), the
// 2^h subsets of a level one per lane, each lane walking its subset's candidates (boundary or partner v, path class 0 / 1).
#pragma once
#include "common.h"

#define DQ_MATCH_MAX_DEFECTS 14
#define DQ_MATCH_MAX_NODES 128                                   // (d^2 - 1) / 2 <= 112 for d <= 15: two 64-bit words of defects
#define DQ_MATCH_LDS (16 + 16 * 16 * 2 + 16 * 2 + (2 << DQ_MATCH_MAX_DEFECTS))      // bytes per wave: list, pair distances, boundary distances, f

struct MatchComp {
    const u8* dist;        // [n][n][2]  shortest path u -> v with class c (255: none), never through the boundary
    const u8* distB;       // [n][2]     shortest path u -> boundary with class c
    int n, w10;            // nodes; weight of the lightest defect-free class-1 error
};

// Class predicted for the defects `d0 | d1 << 64` (bit i = i-th plaquette of the component in row-major order: the look-up referee's index
// convention).  All lanes of the wave call it with the same arguments; the result is wave-uniform.  `s` = DQ_MATCH_LDS bytes of LDS owned
// by this wave.  *inexact is OR-ed with 1 when the fallback was used.
// Lanes of ONE wave hand values to each other through LDS here (list -> distances -> DP levels -> result): a wave's DS operations retire
// in order, but the ordering the C++ sees must not rest on `volatile` alone -- a wavefront-scope release / acquire fence pair plus a wave
// barrier between producer and consumer phases (no cost: s_waitcnt lgkmcnt(0), which the consumer's first read needs anyway).
static __device__ __forceinline__ void match_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

static __device__ __forceinline__ int match_classify(const MatchComp& T, u64 d0, u64 d1, u8* __restrict__ s, int lane, int* inexact) {
    volatile u8* s_list = s;                                      // [16] node of defect i
    volatile u8* s_pd = s + 16;                                   // [16][16][2] distance defect i -> defect j with class c
    volatile u8* s_pb = s + 16 + 512;                             // [16][2]
    volatile u8* f = s + 16 + 512 + 32;                           // [2^k][2]
    const int n0 = __popcll(d0), total = n0 + __popcll(d1);
    const int k = total < DQ_MATCH_MAX_DEFECTS ? total : DQ_MATCH_MAX_DEFECTS;
    // ---- defect list; the defects beyond the first k go to their nearer boundary -------------------------------------------------
    int extra_add = 0, extra_par = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const u64 m = w ? d1 : d0;
        const int node = lane + 64 * w;
        if ((m >> lane) & 1) {
            const int rank = __popcll(m & ((1ull << lane) - 1)) + (w ? n0 : 0);
            if (rank < k) {
                s_list[rank] = (u8)node;
            } else {
                const int b0 = T.distB[2 * node], b1 = T.distB[2 * node + 1];
                const int cp = b1 < b0;
                extra_add += cp ? b1 : b0;
                extra_par ^= cp;
            }
        }
    }
    if (total > k) {                                              // wave-uniform
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { extra_add += __shfl_xor(extra_add, m); extra_par ^= __shfl_xor(extra_par, m); }
        *inexact |= 1;
    }
    match_wave_sync();                                            // (also orders this call's writes after a previous call's reads of `s`)
    // ---- distances among the k defects ---------------------------------------------------------------------------------------
    for (int t = lane; t < k * k; t += 64) {
        const int i = t / k, j = t - i * k;
        const u8* p = T.dist + ((size_t)s_list[i] * T.n + s_list[j]) * 2;
        s_pd[(i * 16 + j) * 2] = p[0];
        s_pd[(i * 16 + j) * 2 + 1] = p[1];
    }
    if (lane < k) { s_pb[2 * lane] = T.distB[2 * s_list[lane]]; s_pb[2 * lane + 1] = T.distB[2 * s_list[lane] + 1]; }
    if (lane == 0) { f[0] = 0; f[1] = 255; }
    match_wave_sync();
    // ---- subsets by their HIGHEST defect h: every S in [2^h, 2^(h+1)) depends only on sets below 2^h, so the 2^h subsets of a level are
    //      independent -- one subset per lane (the oracle recurses on the lowest defect; the minimum over all pairings is the same) ------
    const int BIG = 1 << 20;
    const int full = (1 << k) - 1;
    for (int h = 0; h < k; ++h) {
        const int base = 1 << h;
        const int b0 = s_pb[2 * h] == 255 ? BIG : s_pb[2 * h], b1 = s_pb[2 * h + 1] == 255 ? BIG : s_pb[2 * h + 1];
        for (int r = lane; r < base; r += 64) {
            int g0 = f[2 * r], g1 = f[2 * r + 1];
            g0 = g0 == 255 ? BIG : g0; g1 = g1 == 255 ? BIG : g1;
            int best0 = min(g0 + b0, g1 + b1), best1 = min(g1 + b0, g0 + b1);      // defect h -> boundary with path class 0 / 1
            for (int m = r; m; m &= m - 1) {                          // ... -> partner v
                const int v = __builtin_ctz(m), rr = r ^ (1 << v);
                int d0 = s_pd[(h * 16 + v) * 2], d1 = s_pd[(h * 16 + v) * 2 + 1];
                d0 = d0 == 255 ? BIG : d0; d1 = d1 == 255 ? BIG : d1;
                int q0 = f[2 * rr], q1 = f[2 * rr + 1];
                q0 = q0 == 255 ? BIG : q0; q1 = q1 == 255 ? BIG : q1;
                best0 = min(best0, min(q0 + d0, q1 + d1));
                best1 = min(best1, min(q1 + d0, q0 + d1));
            }
            f[2 * (base + r)] = (u8)(best0 < 255 ? best0 : 255);
            f[2 * (base + r) + 1] = (u8)(best1 < 255 ? best1 : 255);
        }
        match_wave_sync();                                        // level h complete before level h + 1 (and the final read) looks at it
    }
    int w0 = f[2 * full], w1 = f[2 * full + 1];
    w0 = w0 == 255 ? 1 << 20 : w0;
    w1 = w1 == 255 ? 1 << 20 : w1;
    if (extra_par) { const int t = w0; w0 = w1; w1 = t; }
    w0 += extra_add; w1 += extra_add;
    const int v0 = min(w0, w1 + T.w10), v1 = min(w1, w0 + T.w10);
    match_wave_sync();                                            // every lane has read its result before a following call reuses `s`
    return __builtin_amdgcn_readfirstlane(v1 < v0);
}
