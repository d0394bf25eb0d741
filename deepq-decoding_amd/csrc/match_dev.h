// Exact minimum-weight referee without tables ("matching referee"), device side: shared by match.hip (dq_match_decode) and env_big.hip (the
// referee call inside the step of lattices with d >= 9, where the look-up referee's 2^((d^2-1)/2) entries per component no longer fit).
//
// Definition = the look-up referee's (env.hip bfs_* / oracle/referee.py): for one Pauli component predict class 1 iff the lightest error
// with the given syndrome and logical class 1 is STRICTLY lighter than the lightest one with class 0.  Algorithm = oracle/matching_referee.py
// (the numpy restatement this file is tested against bit for bit): the component is a graph (plaquettes = nodes, data qubits = edges, one-
// ended edges go to the boundary, every edge carries the qubit's logical bit); an error is a set of paths pairing the defects with each other
// or the boundary, so  w_c(D) = min over pairings of the summed shortest-path lengths whose classes XOR to c  -- minimum-weight perfect
// matching with class bookkeeping, solved exactly by dynamic programming over the subsets of the (few) defects:
//   f[S][c], S a subset of the defects in index order, lowest defect u of S -> boundary (class c') or -> partner v in S (class c')
//   w_c = min(f[D][c], f[D][c^1] + w_1(0))        (w_1(0): lightest defect-free class-1 error = the code distance)
// More than DQ_MATCH_MAX_DEFECTS defects: the lowest DQ_MATCH_MAX_DEFECTS are solved exactly, every further one goes to its nearer boundary
// (ties: the class-0 path) and the result is flagged inexact.
//
// One wavefront per (syndrome, component).  The 2^k x 2 table lives in LDS as bytes (255 = unreachable; any reachable entry is below
// 14 x (d + 1) / 2 + d <= 127 for d <= 15: every defect can be sent to a boundary within (d + 1) / 2 edges, and forcing the other class
// costs at most one more crossing); subsets are visited in increasing order (both predecessors of S are smaller); for one S the 64 lanes
// are the candidates (partner j or boundary) x (path class c') x (result class c), min-reduced by butterflies inside each half-wave.
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
    // ---- distances among the k defects ---------------------------------------------------------------------------------------
    for (int t = lane; t < k * k; t += 64) {
        const int i = t / k, j = t - i * k;
        const u8* p = T.dist + ((size_t)s_list[i] * T.n + s_list[j]) * 2;
        s_pd[(i * 16 + j) * 2] = p[0];
        s_pd[(i * 16 + j) * 2 + 1] = p[1];
    }
    if (lane < k) { s_pb[2 * lane] = T.distB[2 * s_list[lane]]; s_pb[2 * lane + 1] = T.distB[2 * s_list[lane] + 1]; }
    if (lane == 0) { f[0] = 0; f[1] = 255; }
    // ---- subsets in increasing order -----------------------------------------------------------------------------------------
    const int j = lane & 15, cp = (lane >> 4) & 1, c = lane >> 5;
    const int full = (1 << k) - 1;
    for (int S = 1; S <= full; ++S) {
        const int i = __builtin_ctz(S), rest = S & (S - 1);
        int cand = 1 << 20;
        if (j == 15) {                                            // lowest defect -> boundary
            const int dd = s_pb[2 * i + cp], fv = f[2 * rest + (c ^ cp)];
            if (dd != 255 && fv != 255) cand = fv + dd;
        } else if ((rest >> j) & 1) {                             // ... -> partner j
            const int dd = s_pd[(i * 16 + j) * 2 + cp], fv = f[2 * (rest ^ (1 << j)) + (c ^ cp)];
            if (dd != 255 && fv != 255) cand = fv + dd;
        }
#pragma unroll
        for (int m = 16; m >= 1; m >>= 1) cand = min(cand, __shfl_xor(cand, m));     // inside the half-wave that shares c
        if ((lane & 31) == 0) f[2 * S + c] = (u8)(cand < 255 ? cand : 255);
    }
    int w0 = f[2 * full], w1 = f[2 * full + 1];
    w0 = w0 == 255 ? 1 << 20 : w0;
    w1 = w1 == 255 ? 1 << 20 : w1;
    if (extra_par) { const int t = w0; w0 = w1; w1 = t; }
    w0 += extra_add; w1 += extra_add;
    const int v0 = min(w0, w1 + T.w10), v1 = min(w1, w0 + T.w10);
    return __builtin_amdgcn_readfirstlane(v1 < v0);
}
