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
//
// CLUSTERS (round 5; oracle/matching_referee.py has the argument): a pair whose shortest path of class c' is not strictly shorter than the best
// two boundary paths of the same total class, for either c', never needs to be matched -- so the DP runs per connected component ("cluster") of the
// graph of the remaining pairs and the clusters' (w_0, w_1) combine by a (min, +) XOR-convolution.  Up to DQ_MATCH_MAX_LIST defects of a component are
// listed and clustered.  A cluster of up to DQ_MATCH_MAX_DEFECTS defects is solved in LDS; one of up to DQ_MATCH_MAX_BIG in a slot of the handle's
// scratch pool in device memory (2^20 words, taken with an atomic and given back: rare -- a learning agent's d = 9 run met more than 14 defects in
// 0.3 % of its referee calls -- and 10-100 x slower than the LDS walk, but EXACT: Environments.py:144-151 decides `done` with this answer).
// Fallbacks, deterministic and flagged inexact: of a cluster beyond DQ_MATCH_MAX_BIG the lowest DQ_MATCH_MAX_BIG are solved exactly, every further
// one goes to its nearer boundary (ties: the class-0 path); so does every defect beyond the first DQ_MATCH_MAX_LIST of the component.
//
// One wavefront per (syndrome, component).  The LDS table holds 2^14 x 2 bytes (255 = unreachable; any reachable entry is below
// 14 x (d + 1) / 2 + d <= 127 for d <= 15: every defect can be sent to a boundary within (d + 1) / 2 edges, and forcing the other class
// costs at most one more crossing); subsets are visited level by level of their highest defect (both predecessors of S lie below 2^h), the
// 2^h subsets of a level one per lane, each lane walking its subset's candidates (boundary or partner v, path class 0 / 1).
#pragma once
#include "common.h"

#define DQ_MATCH_MAX_DEFECTS 14                                  // per cluster, in LDS
#define DQ_MATCH_MAX_BIG 20                                      // per cluster, in a scratch-pool slot
#define DQ_MATCH_MAX_LIST 32                                     // defects of a component that are listed and clustered
#define DQ_MATCH_MAX_NODES 128                                   // (d^2 - 1) / 2 <= 112 for d <= 15: two 64-bit words of defects
#define DQ_MATCH_POOL_SLOTS 8                                    // scratch-pool slots of (1 << DQ_MATCH_MAX_BIG) words (32 MB per handle)
// bytes per wave: list [32], boundary distances [32][2], adjacency [32] words, labels [32], cluster members [32], pair distances [32][32][2], f [2^14][2]
#define DQ_MATCH_O_PB 32
#define DQ_MATCH_O_ADJ (DQ_MATCH_O_PB + 64)
#define DQ_MATCH_O_COMP (DQ_MATCH_O_ADJ + 128)
#define DQ_MATCH_O_CL (DQ_MATCH_O_COMP + 32)
#define DQ_MATCH_O_PD (DQ_MATCH_O_CL + 32)
#define DQ_MATCH_O_F (DQ_MATCH_O_PD + 2048)
#define DQ_MATCH_LDS (DQ_MATCH_O_F + (2 << DQ_MATCH_MAX_DEFECTS))

struct MatchComp {
    const u8* dist;        // [n][n][2]  shortest path u -> v with class c (255: none), never through the boundary
    const u8* distB;       // [n][2]     shortest path u -> boundary with class c
    int n, w10;            // nodes; weight of the lightest defect-free class-1 error
    u32* pool;             // [DQ_MATCH_POOL_SLOTS][1 << DQ_MATCH_MAX_BIG] scratch tables of the clusters beyond DQ_MATCH_MAX_DEFECTS (w_0 | w_1 << 16 per subset)
    u32* pool_lock;        // [DQ_MATCH_POOL_SLOTS] 0 free / 1 taken
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

#define DQ_MATCH_BIGW (1 << 20)

// (w_0, w_1) of the cluster whose members (indices into the defect list, ascending) are s_cl[0 .. k): the subset DP by levels of the subsets' HIGHEST
// member.  LDS form (k <= DQ_MATCH_MAX_DEFECTS): bytes, 255 = unreachable.
static __device__ __forceinline__ void match_dp_lds(volatile u8* s_cl, volatile u8* s_pd, volatile u8* s_pb, volatile u8* f, int k, int lane, int& w0, int& w1) {
    const int BIG = DQ_MATCH_BIGW;
    if (lane == 0) { f[0] = 0; f[1] = 255; }
    match_wave_sync();
    for (int h = 0; h < k; ++h) {
        const int base = 1 << h, ch = s_cl[h];
        const int b0 = s_pb[2 * ch] == 255 ? BIG : s_pb[2 * ch], b1 = s_pb[2 * ch + 1] == 255 ? BIG : s_pb[2 * ch + 1];
        for (int r = lane; r < base; r += 64) {
            int g0 = f[2 * r], g1 = f[2 * r + 1];
            g0 = g0 == 255 ? BIG : g0; g1 = g1 == 255 ? BIG : g1;
            int best0 = min(g0 + b0, g1 + b1), best1 = min(g1 + b0, g0 + b1);      // member h -> boundary with path class 0 / 1
            for (int m = r; m; m &= m - 1) {                          // ... -> partner v
                const int v = __builtin_ctz(m), rr = r ^ (1 << v), cv = s_cl[v];
                int d0 = s_pd[(ch * 32 + cv) * 2], d1 = s_pd[(ch * 32 + cv) * 2 + 1];
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
    const int full = (1 << k) - 1;
    w0 = f[2 * full]; w1 = f[2 * full + 1];
    w0 = w0 == 255 ? BIG : w0; w1 = w1 == 255 ? BIG : w1;
    match_wave_sync();                                            // every lane has read the result before the table is reused
}

// The same walk with the table in device memory (k <= DQ_MATCH_MAX_BIG): one word per subset, w_0 | w_1 << 16 (0xffff = unreachable), written and read
// with agent-scope relaxed atomics (they bypass the CU's vector L1, which another lane's store does not refresh) and an agent-scope fence pair between levels.
static __device__ __forceinline__ void match_dp_pool(volatile u8* s_cl, volatile u8* s_pd, volatile u8* s_pb, u32* g, int k, int lane, int& w0, int& w1) {
    const int BIG = DQ_MATCH_BIGW;
    auto ld = [&](int i, int& a, int& b) {
        const u32 x = __hip_atomic_load(g + i, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        a = (int)(x & 0xffffu); b = (int)(x >> 16);
        a = a == 0xffff ? BIG : a; b = b == 0xffff ? BIG : b;
    };
    auto level_sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    };
    if (lane == 0) __hip_atomic_store(g, 0xffff0000u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    level_sync();
    for (int h = 0; h < k; ++h) {
        const int base = 1 << h, ch = s_cl[h];
        const int b0 = s_pb[2 * ch] == 255 ? BIG : s_pb[2 * ch], b1 = s_pb[2 * ch + 1] == 255 ? BIG : s_pb[2 * ch + 1];
        for (int r = lane; r < base; r += 64) {
            int g0, g1;
            ld(r, g0, g1);
            int best0 = min(g0 + b0, g1 + b1), best1 = min(g1 + b0, g0 + b1);
            for (int m = r; m; m &= m - 1) {
                const int v = __builtin_ctz(m), rr = r ^ (1 << v), cv = s_cl[v];
                int d0 = s_pd[(ch * 32 + cv) * 2], d1 = s_pd[(ch * 32 + cv) * 2 + 1];
                d0 = d0 == 255 ? BIG : d0; d1 = d1 == 255 ? BIG : d1;
                int q0, q1;
                ld(rr, q0, q1);
                best0 = min(best0, min(q0 + d0, q1 + d1));
                best1 = min(best1, min(q1 + d0, q0 + d1));
            }
            const u32 o = (u32)(best0 < 0xffff ? best0 : 0xffff) | (u32)(best1 < 0xffff ? best1 : 0xffff) << 16;
            __hip_atomic_store(g + base + r, o, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        level_sync();
    }
    ld((1 << k) - 1, w0, w1);
}

static __device__ __forceinline__ int match_classify(const MatchComp& T, u64 d0, u64 d1, u8* __restrict__ s, int lane, int* inexact) {
    volatile u8* s_list = s;                                      // [32] node of defect i
    volatile u8* s_pb = s + DQ_MATCH_O_PB;                        // [32][2] boundary distances
    volatile u32* s_adj = reinterpret_cast<volatile u32*>(s + DQ_MATCH_O_ADJ);      // [32] bit j: defects i and j may be worth matching
    volatile u8* s_comp = s + DQ_MATCH_O_COMP;                    // [32] lowest defect of i's cluster
    volatile u8* s_cl = s + DQ_MATCH_O_CL;                        // [32] members of the cluster being solved
    volatile u8* s_pd = s + DQ_MATCH_O_PD;                        // [32][32][2] distance defect i -> defect j with class c
    volatile u8* f = s + DQ_MATCH_O_F;                            // [2^k][2]
    const int BIG = DQ_MATCH_BIGW;
    const int n0 = __popcll(d0), total = n0 + __popcll(d1);
    const int L = total < DQ_MATCH_MAX_LIST ? total : DQ_MATCH_MAX_LIST;
    // boundary fallback of the defects this lane owns that are not solved exactly: accumulated here, reduced over the wave where it is applied
    auto to_boundary = [&](int node, int& add, int& par) {
        const int b0 = T.distB[2 * node], b1 = T.distB[2 * node + 1];
        const int cp = b1 < b0;
        add += cp ? b1 : b0;
        par ^= cp;
    };
    auto wave_sum_xor = [&](int& add, int& par) {
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) { add += __shfl_xor(add, m); par ^= __shfl_xor(par, m); }
    };
    // ---- defect list; the defects beyond the first L go to their nearer boundary ------------------------------------------------------
    int extra_add = 0, extra_par = 0;
#pragma unroll
    for (int w = 0; w < 2; ++w) {
        const u64 m = w ? d1 : d0;
        const int node = lane + 64 * w;
        if ((m >> lane) & 1) {
            const int rank = __popcll(m & ((1ull << lane) - 1)) + (w ? n0 : 0);
            if (rank < L) s_list[rank] = (u8)node;
            else to_boundary(node, extra_add, extra_par);
        }
    }
    if (total > L) {                                              // wave-uniform
        wave_sum_xor(extra_add, extra_par);
        *inexact |= 1;
    }
    match_wave_sync();                                            // (also orders this call's writes after a previous call's reads of `s`)
    // ---- distances among the L defects ----------------------------------------------------------------------------------------------
    for (int t = lane; t < L * L; t += 64) {
        const int i = t / L, j = t - i * L;
        const u8* p = T.dist + ((size_t)s_list[i] * T.n + s_list[j]) * 2;
        s_pd[(i * 32 + j) * 2] = p[0];
        s_pd[(i * 32 + j) * 2 + 1] = p[1];
    }
    if (lane < L) { s_pb[2 * lane] = T.distB[2 * s_list[lane]]; s_pb[2 * lane + 1] = T.distB[2 * s_list[lane] + 1]; }
    match_wave_sync();
    // ---- clusters: lane i owns defect i.  adjacency, then label propagation to the cluster's lowest defect -------------------------------
    int comp = lane;
    if (lane < L) {
        u32 adj = 0;
        const int bi0 = s_pb[2 * lane] == 255 ? BIG : s_pb[2 * lane], bi1 = s_pb[2 * lane + 1] == 255 ? BIG : s_pb[2 * lane + 1];
        for (int j = 0; j < L; ++j) {
            if (j == lane) continue;
            const int bj0 = s_pb[2 * j] == 255 ? BIG : s_pb[2 * j], bj1 = s_pb[2 * j + 1] == 255 ? BIG : s_pb[2 * j + 1];
            const int p0 = s_pd[(lane * 32 + j) * 2], p1 = s_pd[(lane * 32 + j) * 2 + 1];
            const bool c0 = p0 != 255 && p0 < min(bi0 + bj0, bi1 + bj1);      // path class 0: boundary classes (0, 0) or (1, 1)
            const bool c1 = p1 != 255 && p1 < min(bi0 + bj1, bi1 + bj0);      // path class 1: (0, 1) or (1, 0)
            if (c0 || c1) adj |= 1u << j;
        }
        s_adj[lane] = adj;
        s_comp[lane] = (u8)lane;
    }
    match_wave_sync();
    for (int it = 0; it < DQ_MATCH_MAX_LIST; ++it) {              // (a label travels one edge per pass: at most L - 1 passes)
        int nc = comp;
        if (lane < L)
            for (u32 m = s_adj[lane]; m; m &= m - 1) nc = min(nc, (int)s_comp[__builtin_ctz(m)]);
        const bool changed = lane < L && nc != comp;
        match_wave_sync();                                        // every lane has read the labels of this pass
        if (changed) { comp = nc; s_comp[lane] = (u8)nc; }
        match_wave_sync();
        if (!__ballot(changed)) break;                            // wave-uniform
    }
    // ---- the clusters in the order of their lowest defect ---------------------------------------------------------------------------
    int W0 = 0, W1 = BIG;
    u64 reps = __ballot(lane < L && comp == lane);
    while (reps) {                                                // wave-uniform
        const int rep = __builtin_ctzll(reps);
        reps &= reps - 1;
        const u64 members = __ballot(lane < L && comp == rep);
        const int msize = __popcll(members);
        const int k = msize < DQ_MATCH_MAX_BIG ? msize : DQ_MATCH_MAX_BIG;
        int cl_add = 0, cl_par = 0;
        if ((members >> lane) & 1) {
            const int rank = __popcll(members & ((1ull << lane) - 1));
            if (rank < k) s_cl[rank] = (u8)lane;
            else to_boundary(s_list[lane], cl_add, cl_par);
        }
        match_wave_sync();
        int w0, w1;
        if (k <= DQ_MATCH_MAX_DEFECTS) {
            match_dp_lds(s_cl, s_pd, s_pb, f, k, lane, w0, w1);
        } else {
            // a slot of the scratch pool: lane 0 takes the first free one (spinning over the slots: holders always finish -- and a slot left taken by a launch that
            // died is freed by the host in front of the next launch, match.hip match_reset_locks), every lane uses it, lane 0 frees it.  Worst case: a cluster of
            // DQ_MATCH_MAX_BIG = 20 defects walks 2^20 subsets through device memory, a multi-millisecond stall of its wave (and of the vector step that waits for it);
            // the d = 9 fit of the test suite meets 15-20-defect clusters on a few dozen of its ~10^5 lattice-steps
            int slot = 0;
            if (lane == 0) {
                for (;; slot = (slot + 1) % DQ_MATCH_POOL_SLOTS) {
                    if (atomicCAS(T.pool_lock + slot, 0u, 1u) == 0u) break;
                    if (slot == DQ_MATCH_POOL_SLOTS - 1) __builtin_amdgcn_s_sleep(32);
                }
            }
            slot = __builtin_amdgcn_readfirstlane(slot);
            match_dp_pool(s_cl, s_pd, s_pb, T.pool + ((size_t)slot << DQ_MATCH_MAX_BIG), k, lane, w0, w1);
            __builtin_amdgcn_wave_barrier();
            if (lane == 0) __hip_atomic_store(T.pool_lock + slot, 0u, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_AGENT);
        }
        if (msize > k) {                                          // wave-uniform: the cluster's members beyond the exact ones
            wave_sum_xor(cl_add, cl_par);
            if (cl_par) { const int t = w0; w0 = w1; w1 = t; }
            w0 = min(w0 + cl_add, BIG); w1 = min(w1 + cl_add, BIG);
            *inexact |= 1;
        }
        const int n0w = min(min(W0 + w0, W1 + w1), BIG), n1w = min(min(W0 + w1, W1 + w0), BIG);
        W0 = n0w; W1 = n1w;
    }
    if (extra_par) { const int t = W0; W0 = W1; W1 = t; }
    W0 += extra_add; W1 += extra_add;
    const int v0 = min(W0, W1 + T.w10), v1 = min(W1, W0 + T.w10);
    match_wave_sync();                                            // every lane is through with `s` before a following call reuses it
    return __builtin_amdgcn_readfirstlane(v1 < v0);
}
