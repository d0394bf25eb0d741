// Action selection on the device: epsilon-greedy with legal-move exploration and optional
// legal-only greedy choice.
//
// Replaces the keras-rl fork's EpsGreedyQPolicy(masked_greedy=...) / GreedyQPolicy(masked_greedy=True)
// (un-vendored; call sites /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:110-115,
// 166-167).  Semantics follow README.md:168 ("we restrict the agents random choice to actions which
// are either adjacent to violated stabilizer, or adjacent to previously acted on qubits") and
// README.md:262 (masked_greedy: argmax restricted to the legal actions).
//
// 16 lanes cooperate on one lattice (4 lattices per wavefront): coalesced reads of the Q row, then a
// 4-step butterfly over (value, index) keeping the first maximum like np.argmax.
#include "common.h"

__global__ __launch_bounds__(256) void policy_kernel(const float* __restrict__ q, const u64* __restrict__ legal, int n,
                                                     int n_actions, u64 T_eps, int masked_greedy, u32 seed0, u32 seed1,
                                                     u32 env_id_base, u64 t, int32_t* __restrict__ action) {
    const int gid = blockIdx.x * blockDim.x + threadIdx.x;
    const int i = gid >> 4, sub = gid & 15;
    if (i >= n) return;          // whole 16-lane groups leave together
    const u64 lo = legal[2 * (size_t)i], hi = legal[2 * (size_t)i + 1];
    u32 w[4];
    philox4x32_10((u32)t, (u32)(t >> 32), env_id_base + (u32)i, (u32)DQ_STREAM_POLICY << 16, seed0, seed1, w);
    const bool explore = q == nullptr || (u64)w[1] < T_eps;
    int a;
    if (explore) {
        const int n_legal = __popcll(lo) + __popcll(hi);
        a = kth_set_bit128(lo, hi, (int)__umulhi(w[0], (u32)n_legal));
    } else {
        float best = -INFINITY;
        int best_a = 0x7fffffff;
        const float* row = q + (size_t)i * n_actions;
        for (int k = sub; k < n_actions; k += 16) {
            const bool ok = !masked_greedy || (((k < 64 ? lo : hi) >> (k & 63)) & 1);
            const float v = row[k];
            if (ok && (v > best || best_a == 0x7fffffff)) { best = v; best_a = k; }
        }
#pragma unroll
        for (int m = 8; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m, 16);
            const int oa = __shfl_xor(best_a, m, 16);
            if (oa != 0x7fffffff && (best_a == 0x7fffffff || ov > best || (ov == best && oa < best_a))) { best = ov; best_a = oa; }
        }
        a = best_a;
    }
    if (sub == 0) action[i] = a;
}

extern "C" dq_status dq_policy_select(const float* q_dev, const uint64_t* legal_dev, int n, int n_actions, double eps,
                                      int masked_greedy, const uint32_t seed[2], uint32_t env_id_base, uint64_t t,
                                      int32_t* action_dev, void* stream) {
    DQ_REQUIRE(legal_dev && action_dev && seed, DQ_ERR_INVALID, "dq_policy_select: null argument");
    DQ_REQUIRE(n >= 1 && n_actions >= 1 && n_actions <= 128, DQ_ERR_INVALID, "dq_policy_select: bad sizes");
    DQ_REQUIRE(eps >= 0.0 && eps <= 1.0, DQ_ERR_INVALID, "dq_policy_select: eps must be in [0,1]");
    const int threads = n * 16;
    policy_kernel<<<(threads + 255) / 256, 256, 0, (hipStream_t)stream>>>(q_dev, legal_dev, n, n_actions, dq_rate_threshold(eps),
                                                                         masked_greedy, seed[0], seed[1], env_id_base, t, action_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}
