// DQN update pieces around the Q-network: replay minibatch sampling, double-DQN TD target,
// masked squared-error loss + gradient, Keras-style Adam.
//
// Replaces, from the un-vendored keras-rl fork (call sites
// /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:109,119-130):
//   SequentialMemory.sample          -> replay_sample_kernel (time-major device ring, validity rule kept)
//   DQNAgent.backward (double DQN)   -> td_target_kernel, td_loss_grad_kernel
//   keras.optimizers.Adam            -> adam_kernel
#include "common.h"

// Device replay ring: row r = slot * n_envs + env holds (obs s_r, action, reward, terminal of the step taken
// in s_r); the successor observation of row r is row r + n_envs (mod n_slots*n_envs).  keras-rl's rule: a
// transition is not sampled when the PREVIOUS entry of that lattice was terminal (its s0 is then the terminal
// observation the agent only looked at before env.reset()).
__global__ void replay_sample_kernel(const u8* __restrict__ terminal, int n_envs, int n_slots, int head_slot, int filled,
                                     int batch, u32 seed0, u32 seed1, u64 t, u32 sample_base, int32_t* __restrict__ index) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= batch) return;
    const int row = dq_replay_row(terminal, n_envs, n_slots, head_slot, filled, batch, seed0, seed1, t, sample_base + (u32)b);
    index[b] = row;
}

// the minibatches of n_updates consecutive updates t0, t0 + 1, ... on ONE ring state, in one launch: index[u][b] = the row update t0 + u draws for sample b
__global__ void replay_sample_multi_kernel(const u8* __restrict__ terminal, int n_envs, int n_slots, int head_slot, int filled,
                                           int batch, u32 seed0, u32 seed1, u64 t0, int n_updates, u32 sample_base, int32_t* __restrict__ index) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (long long)batch * n_updates) return;
    const int u = (int)(i / batch), b = (int)(i - (long long)u * batch);
    index[i] = dq_replay_row(terminal, n_envs, n_slots, head_slot, filled, batch, seed0, seed1, t0 + (u64)u, sample_base + (u32)b);
}

// y_b = r_b + gamma * (1 - terminal_b) * Q_target(s1_b)[argmax_a Q_online(s1_b)[a]]; one wave per sample
__global__ void td_target_kernel(const float* __restrict__ q_online, const float* __restrict__ q_target,
                                 const float* __restrict__ reward, const u8* __restrict__ terminal,
                                 const int32_t* __restrict__ index, float gamma, int B, int A, float* __restrict__ y) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* row = q_online + (size_t)b * A;
    float best = -INFINITY;
    int best_a = 0x7fffffff;
    for (int a = lane; a < A; a += 64) {
        const float v = row[a];
        if (best_a == 0x7fffffff || v > best) { best = v; best_a = a; }
    }
    for (int m = 32; m >= 1; m >>= 1) {
        const float ov = __shfl_xor(best, m);
        const int oa = __shfl_xor(best_a, m);
        if (oa != 0x7fffffff && (best_a == 0x7fffffff || ov > best || (ov == best && oa < best_a))) { best = ov; best_a = oa; }
    }
    if (lane == 0) {
        const int r = index ? index[b] : b;
        const float qn = q_target[(size_t)b * A + best_a];
        y[b] = reward[r] + (terminal[r] ? 0.f : gamma * qn);
    }
}

// dq[b, a] = grad_scale * (Q[b,a_b] - y_b) at a = a_b, else 0;  metrics[0] = mean_b 0.5 (Q[b,a_b]-y_b)^2,
// metrics[1] = mean_b max_a Q[b,a].  One wave per sample; per-block partials land in metrics[2 + 2*block ...] and a
// second single-block pass adds them in a fixed order (deterministic).
#define TD_MAX_BLOCKS 1024
__global__ __launch_bounds__(256) void td_loss_grad_kernel(const float* __restrict__ q, const int32_t* __restrict__ action,
                                                           const int32_t* __restrict__ index, const float* __restrict__ y, int B,
                                                           int A, float grad_scale, float* __restrict__ dq, float* __restrict__ metrics) {
    __shared__ float s_loss[4], s_q[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f, mq = 0.f;
    for (int b = blockIdx.x * 4 + wave; b < B; b += gridDim.x * 4) {
        const float* row = q + (size_t)b * A;
        float* drow = dq + (size_t)b * A;
        const int a_b = action[index ? index[b] : b];
        float mx = -INFINITY;
        for (int a = lane; a < A; a += 64) {
            const float v = row[a];
            mx = fmaxf(mx, v);
            drow[a] = a == a_b ? (v - y[b]) * grad_scale : 0.f;
        }
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
        const float diff = row[a_b] - y[b];
        loss += 0.5f * diff * diff;
        mq += mx;
    }
    if (lane == 0) { s_loss[wave] = loss; s_q[wave] = mq; }
    __syncthreads();
    if (threadIdx.x == 0 && metrics) {
        metrics[2 + 2 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        metrics[3 + 2 * blockIdx.x] = (s_q[0] + s_q[1]) + (s_q[2] + s_q[3]);
    }
}

struct TdStats { const u8* done; const u8* was_reset; const u32* lifetime; const float* reward; int n; unsigned long long* stats; };

// The whole TD step of one update in one launch (one wave per sample): double-DQN target from Q_online(s1) / Q_target(s1), then the
// masked squared-error loss and its gradient on Q(s0).  Same arithmetic and the same per-block metric partials as
// td_target_kernel + td_loss_grad_kernel; y is also written (nullable) for tests / logging.
__global__ __launch_bounds__(256) void td_update_kernel(const float* __restrict__ q_online, const float* __restrict__ q_target,
                                                        const float* __restrict__ q, const float* __restrict__ reward,
                                                        const u8* __restrict__ terminal, const int32_t* __restrict__ action,
                                                        const int32_t* __restrict__ index, float gamma, int B, int A, float grad_scale,
                                                        float* __restrict__ y_out, float* __restrict__ dq, float* __restrict__ metrics,
                                                        int td_blocks, TdStats st) {
    if ((int)blockIdx.x >= td_blocks) {                             // the episode bookkeeping of the step just taken rides along
        dq_episode_stats_lane(st.done, st.was_reset, st.lifetime, st.reward, st.n, ((int)blockIdx.x - td_blocks) * blockDim.x + threadIdx.x,
                              st.stats);
        return;
    }
    __shared__ float s_loss[4], s_q[4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float loss = 0.f, mq = 0.f;
    for (int b = blockIdx.x * 4 + wave; b < B; b += td_blocks * 4) {
        const float* row1 = q_online + (size_t)b * A;
        float best = -INFINITY;
        int best_a = 0x7fffffff;
        for (int a = lane; a < A; a += 64) {
            const float v = row1[a];
            if (best_a == 0x7fffffff || v > best) { best = v; best_a = a; }
        }
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m);
            const int oa = __shfl_xor(best_a, m);
            if (oa != 0x7fffffff && (best_a == 0x7fffffff || ov > best || (ov == best && oa < best_a))) { best = ov; best_a = oa; }
        }
        const int r = index ? index[b] : b;
        const float yb = reward[r] + (terminal[r] ? 0.f : gamma * q_target[(size_t)b * A + best_a]);
        if (lane == 0 && y_out) y_out[b] = yb;
        const float* row = q + (size_t)b * A;
        float* drow = dq + (size_t)b * A;
        const int a_b = action[r];
        float mx = -INFINITY;
        for (int a = lane; a < A; a += 64) {
            const float v = row[a];
            mx = fmaxf(mx, v);
            drow[a] = a == a_b ? (v - yb) * grad_scale : 0.f;
        }
        for (int m = 32; m >= 1; m >>= 1) mx = fmaxf(mx, __shfl_xor(mx, m));
        const float diff = row[a_b] - yb;
        loss += 0.5f * diff * diff;
        mq += mx;
    }
    if (lane == 0) { s_loss[wave] = loss; s_q[wave] = mq; }
    __syncthreads();
    if (threadIdx.x == 0 && metrics) {
        metrics[2 + 2 * blockIdx.x] = (s_loss[0] + s_loss[1]) + (s_loss[2] + s_loss[3]);
        metrics[3 + 2 * blockIdx.x] = (s_q[0] + s_q[1]) + (s_q[2] + s_q[3]);
    }
}

__global__ __launch_bounds__(256) void td_metrics_kernel(float* __restrict__ metrics, int blocks, int B) {
    __shared__ float s_loss[256], s_q[256];
    float loss = 0.f, mq = 0.f;
    for (int k = threadIdx.x; k < blocks; k += 256) { loss += metrics[2 + 2 * k]; mq += metrics[3 + 2 * k]; }
    s_loss[threadIdx.x] = loss;
    s_q[threadIdx.x] = mq;
    __syncthreads();
    for (int s = 128; s >= 1; s >>= 1) {
        if ((int)threadIdx.x < s) { s_loss[threadIdx.x] += s_loss[threadIdx.x + s]; s_q[threadIdx.x] += s_q[threadIdx.x + s]; }
        __syncthreads();
    }
    if (threadIdx.x == 0) { metrics[0] = s_loss[0] / (float)B; metrics[1] = s_q[0] / (float)B; }
}

// Keras 2.2 Adam.get_updates (common.h dq_adam1).  With a flag word (dq_qnet_adam_step: the several-GPU branch) a non-finite gradient element (the
// fused backward's range guard, include/deepq_hip.h dq_qnet_range_check; behind an all-reduce every rank sees the same ones) leaves its parameter
// and moments untouched and raises *flag -- the rule of the optimizer step that rides on the fused backward's final reduction, so that the branch
// where the all-reduce comes between backward and update behaves the same on EVERY rank.  WITHOUT a flag (the public dq_adam_step) the update is
// Keras': a NaN gradient propagates into the parameter and the divergence is visible, not silently skipped.
__device__ __forceinline__ void dq_adam1_guarded(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps, bool guard, bool& bad) {
    if (!guard || __builtin_isfinite(g)) dq_adam1(p, g, m, v, lr_t, b1, b2, eps);
    else bad = true;
}
__global__ void adam_kernel(float* __restrict__ p, const float* __restrict__ g, float* __restrict__ m, float* __restrict__ v,
                            size_t n, float lr_t, float b1, float b2, float eps, unsigned* __restrict__ flag) {
    const size_t i4 = ((size_t)blockIdx.x * blockDim.x + threadIdx.x) * 4;
    bool bad = false;
    const bool guard = flag != nullptr;
    if (i4 + 3 < n && ((reinterpret_cast<uintptr_t>(p) | reinterpret_cast<uintptr_t>(g) | reinterpret_cast<uintptr_t>(m) |
                        reinterpret_cast<uintptr_t>(v)) & 15) == 0) {
        float4 pp = *reinterpret_cast<float4*>(p + i4);
        const float4 gg = *reinterpret_cast<const float4*>(g + i4);
        float4 mm = *reinterpret_cast<float4*>(m + i4), vv = *reinterpret_cast<float4*>(v + i4);
        dq_adam1_guarded(pp.x, gg.x, mm.x, vv.x, lr_t, b1, b2, eps, guard, bad);
        dq_adam1_guarded(pp.y, gg.y, mm.y, vv.y, lr_t, b1, b2, eps, guard, bad);
        dq_adam1_guarded(pp.z, gg.z, mm.z, vv.z, lr_t, b1, b2, eps, guard, bad);
        dq_adam1_guarded(pp.w, gg.w, mm.w, vv.w, lr_t, b1, b2, eps, guard, bad);
        *reinterpret_cast<float4*>(p + i4) = pp;
        *reinterpret_cast<float4*>(m + i4) = mm;
        *reinterpret_cast<float4*>(v + i4) = vv;
    } else {
        for (size_t i = i4; i < n && i < i4 + 4; ++i) {
            float pi = p[i], mi = m[i], vi = v[i];
            dq_adam1_guarded(pi, g[i], mi, vi, lr_t, b1, b2, eps, guard, bad);
            p[i] = pi; m[i] = mi; v[i] = vi;
        }
    }
    if (bad && flag) atomicOr(flag, 1u);
    // a WHOLE update discarded by the range guard's early half arrives here as NaN in every element (fused_bwd.hip reduce_finish; through the all-reduce
    // on every rank): counted once per optimizer step in the word behind the flag's (dq_qnet_range_discarded)
    if (flag && blockIdx.x == 0 && threadIdx.x == 0 && n > 0 && !__builtin_isfinite(g[0]) && !__builtin_isfinite(g[n - 1])) atomicAdd(flag + 4, 1u);
}

// stats[0] += #episodes that ended this step, stats[1] += sum of their lifetimes, stats[2] += #rewards == 1,
// stats[3] += #lattices stepped (not reset).  Integer atomics => order-independent result.
__global__ void episode_stats_kernel(const u8* __restrict__ done, const u8* __restrict__ was_reset, const u32* __restrict__ lifetime,
                                     const float* __restrict__ reward, int n, unsigned long long* __restrict__ stats) {
    dq_episode_stats_lane(done, was_reset, lifetime, reward, n, blockIdx.x * blockDim.x + threadIdx.x, stats);
}

// Replay sampling for the next update and the episode bookkeeping of the step just taken, in one launch: blocks
// [0, sample_blocks) run replay_sample_kernel's rule, the rest episode_stats_kernel's.
__global__ __launch_bounds__(256) void post_step_kernel(const u8* __restrict__ terminal, int n_envs, int n_slots, int head_slot, int filled,
                                                        int batch, u32 seed0, u32 seed1, u64 t, u32 sample_base, int32_t* __restrict__ index,
                                                        int sample_blocks, const u8* __restrict__ done, const u8* __restrict__ was_reset,
                                                        const u32* __restrict__ lifetime, const float* __restrict__ reward, int n,
                                                        unsigned long long* __restrict__ stats) {
    if ((int)blockIdx.x < sample_blocks) {
        const int b = blockIdx.x * blockDim.x + threadIdx.x;
        if (b < batch) index[b] = dq_replay_row(terminal, n_envs, n_slots, head_slot, filled, batch, seed0, seed1, t, sample_base + (u32)b);
        return;
    }
    dq_episode_stats_lane(done, was_reset, lifetime, reward, n, ((int)blockIdx.x - sample_blocks) * blockDim.x + threadIdx.x, stats);
}

// Episode records of a greedy evaluation (DQNAgent.test, Single_Point_Training_Script.py:206 -> keras-rl Agent.test), kept on the device:
// per lattice the running episode reward / length and the episodes it still owes (its quota); a lattice that ends an episode while it
// owes one appends a record {vector step, lattice, reward, length, lifetime} (slot taken with one atomic; the host sorts by (step,
// lattice), the order keras-rl's serial loop would have produced) and its counters restart.  One thread per lattice.
__global__ void test_bookkeeping_kernel(const u8* __restrict__ done, const u8* __restrict__ was_reset, const float* __restrict__ reward,
                                        const u32* __restrict__ lifetime, int n, int step, int32_t* __restrict__ quota,
                                        float* __restrict__ ep_reward, int32_t* __restrict__ ep_len, int32_t* __restrict__ records,
                                        int capacity, int32_t* __restrict__ counter) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const bool wr = was_reset[i] != 0;
    float r = ep_reward[i];
    int len = ep_len[i];
    if (!wr) { r += reward[i]; len += 1; }
    if (done[i] && !wr) {
        if (quota[i] > 0) {
            quota[i] -= 1;
            const int slot = atomicAdd(counter, 1);
            if (slot < capacity) {
                int32_t* rec = records + 5 * (size_t)slot;
                rec[0] = step; rec[1] = i; rec[2] = __float_as_int(r); rec[3] = len; rec[4] = (int32_t)lifetime[i];
            }
        }
        r = 0.f; len = 0;
    }
    ep_reward[i] = r; ep_len[i] = len;
}

extern "C" {

dq_status dq_post_step(const uint8_t* terminal_ring_dev, int n_envs, int n_slots, int head_slot, int filled_slots, int batch,
                       const uint32_t seed[2], uint64_t t, uint32_t sample_base, int32_t* index_dev, const uint8_t* done_dev,
                       const uint8_t* was_reset_dev, const uint32_t* lifetime_dev, const float* reward_dev, int n, uint64_t* stats_dev,
                       void* stream) {
    DQ_REQUIRE(terminal_ring_dev && index_dev && seed, DQ_ERR_INVALID, "dq_post_step: null argument");
    DQ_REQUIRE(n_envs >= 1 && n_slots >= 4 && batch >= 1 && head_slot >= 0 && head_slot < n_slots, DQ_ERR_INVALID, "dq_post_step: bad sizes");
    DQ_REQUIRE(filled_slots >= DQ_REPLAY_MIN_FILLED && filled_slots <= n_slots, DQ_ERR_STATE, "dq_post_step: need at least three complete transitions per lattice");
    DQ_REQUIRE((long long)n_envs * n_slots < (1ll << 31), DQ_ERR_UNSUPPORTED, "dq_post_step: ring too large for 32-bit rows");
    DQ_REQUIRE(done_dev && lifetime_dev && reward_dev && stats_dev && n >= 1, DQ_ERR_INVALID, "dq_post_step: bad statistics argument");
    const int sb = (batch + 255) / 256;
    post_step_kernel<<<sb + (n + 255) / 256, 256, 0, (hipStream_t)stream>>>(terminal_ring_dev, n_envs, n_slots, head_slot, filled_slots, batch,
                                                                          seed[0], seed[1], t, sample_base, index_dev, sb, done_dev, was_reset_dev,
                                                                          lifetime_dev, reward_dev, n, reinterpret_cast<unsigned long long*>(stats_dev));
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

static dq_status launch_td_update(const float* q_online_s1_dev, const float* q_target_s1_dev, const float* q_s0_dev, const float* reward_dev,
                                  const uint8_t* terminal_dev, const int32_t* action_dev, const int32_t* index_dev, double gamma, int batch,
                                  int n_actions, double grad_scale, float* y_dev, float* dq_dev, float* metrics_dev, const TdStats& ts,
                                  void* stream) {
    DQ_REQUIRE(q_online_s1_dev && q_target_s1_dev && q_s0_dev && reward_dev && terminal_dev && action_dev && dq_dev, DQ_ERR_INVALID,
               "dq_td_update: null argument");
    DQ_REQUIRE(batch >= 1 && n_actions >= 1, DQ_ERR_INVALID, "dq_td_update: bad sizes");
    const int blocks = (batch + 3) / 4 < TD_MAX_BLOCKS ? (batch + 3) / 4 : TD_MAX_BLOCKS;
    const int stat_blocks = ts.n > 0 ? (ts.n + 255) / 256 : 0;
    dq_prof_begin(DQ_K_TD, (hipStream_t)stream);
    td_update_kernel<<<blocks + stat_blocks, 256, 0, (hipStream_t)stream>>>(q_online_s1_dev, q_target_s1_dev, q_s0_dev, reward_dev, terminal_dev,
                                                                            action_dev, index_dev, (float)gamma, batch, n_actions,
                                                                            (float)grad_scale, y_dev, dq_dev, metrics_dev, blocks, ts);
    dq_prof_end(DQ_K_TD, (hipStream_t)stream);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_td_update(const float* q_online_s1_dev, const float* q_target_s1_dev, const float* q_s0_dev, const float* reward_dev,
                       const uint8_t* terminal_dev, const int32_t* action_dev, const int32_t* index_dev, double gamma, int batch, int n_actions,
                       double grad_scale, float* y_dev, float* dq_dev, float* metrics_dev, void* stream) {
    TdStats ts;
    memset(&ts, 0, sizeof(ts));
    return launch_td_update(q_online_s1_dev, q_target_s1_dev, q_s0_dev, reward_dev, terminal_dev, action_dev, index_dev, gamma, batch, n_actions,
                            grad_scale, y_dev, dq_dev, metrics_dev, ts, stream);
}

dq_status dq_td_update_stats(const float* q_online_s1_dev, const float* q_target_s1_dev, const float* q_s0_dev, const float* reward_dev,
                             const uint8_t* terminal_dev, const int32_t* action_dev, const int32_t* index_dev, double gamma, int batch,
                             int n_actions, double grad_scale, float* y_dev, float* dq_dev, float* metrics_dev, const uint8_t* done_dev,
                             const uint8_t* was_reset_dev, const uint32_t* lifetime_dev, const float* step_reward_dev, int n,
                             uint64_t* stats_dev, void* stream) {
    DQ_REQUIRE(done_dev && lifetime_dev && step_reward_dev && stats_dev && n >= 1, DQ_ERR_INVALID, "dq_td_update_stats: bad statistics argument");
    TdStats ts = {done_dev, was_reset_dev, lifetime_dev, step_reward_dev, n, reinterpret_cast<unsigned long long*>(stats_dev)};
    return launch_td_update(q_online_s1_dev, q_target_s1_dev, q_s0_dev, reward_dev, terminal_dev, action_dev, index_dev, gamma, batch, n_actions,
                            grad_scale, y_dev, dq_dev, metrics_dev, ts, stream);
}

dq_status dq_td_metrics(float* metrics_dev, int batch, void* stream) {
    DQ_REQUIRE(metrics_dev && batch >= 1, DQ_ERR_INVALID, "dq_td_metrics: bad argument");
    const int blocks = (batch + 3) / 4 < TD_MAX_BLOCKS ? (batch + 3) / 4 : TD_MAX_BLOCKS;
    td_metrics_kernel<<<1, 256, 0, (hipStream_t)stream>>>(metrics_dev, blocks, batch);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_episode_stats(const uint8_t* done_dev, const uint8_t* was_reset_dev, const uint32_t* lifetime_dev, const float* reward_dev,
                           int n, uint64_t* stats_dev, void* stream) {
    DQ_REQUIRE(done_dev && lifetime_dev && reward_dev && stats_dev && n >= 1, DQ_ERR_INVALID, "dq_episode_stats: bad argument");
    episode_stats_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(done_dev, was_reset_dev, lifetime_dev, reward_dev, n,
                                                                         reinterpret_cast<unsigned long long*>(stats_dev));
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_test_bookkeeping(const uint8_t* done_dev, const uint8_t* was_reset_dev, const float* reward_dev, const uint32_t* lifetime_dev, int n,
                              int step, int32_t* quota_dev, float* ep_reward_dev, int32_t* ep_len_dev, int32_t* records_dev, int capacity,
                              int32_t* counter_dev, void* stream) {
    DQ_REQUIRE(done_dev && was_reset_dev && reward_dev && lifetime_dev && quota_dev && ep_reward_dev && ep_len_dev && records_dev && counter_dev,
               DQ_ERR_INVALID, "dq_test_bookkeeping: null argument");
    DQ_REQUIRE(n >= 1 && capacity >= 1, DQ_ERR_INVALID, "dq_test_bookkeeping: bad sizes");
    test_bookkeeping_kernel<<<(n + 255) / 256, 256, 0, (hipStream_t)stream>>>(done_dev, was_reset_dev, reward_dev, lifetime_dev, n, step, quota_dev,
                                                                            ep_reward_dev, ep_len_dev, records_dev, capacity, counter_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_replay_sample(const uint8_t* terminal_ring_dev, int n_envs, int n_slots, int head_slot, int filled_slots, int batch,
                           const uint32_t seed[2], uint64_t t, uint32_t sample_base, int32_t* index_dev, void* stream) {
    DQ_REQUIRE(terminal_ring_dev && index_dev && seed, DQ_ERR_INVALID, "dq_replay_sample: null argument");
    DQ_REQUIRE(n_envs >= 1 && n_slots >= 4 && batch >= 1 && head_slot >= 0 && head_slot < n_slots, DQ_ERR_INVALID, "dq_replay_sample: bad sizes");
    DQ_REQUIRE(filled_slots >= DQ_REPLAY_MIN_FILLED && filled_slots <= n_slots, DQ_ERR_STATE, "dq_replay_sample: need at least three complete transitions per lattice");
    DQ_REQUIRE((long long)n_envs * n_slots < (1ll << 31), DQ_ERR_UNSUPPORTED, "dq_replay_sample: ring too large for 32-bit rows");
    replay_sample_kernel<<<(batch + 255) / 256, 256, 0, (hipStream_t)stream>>>(terminal_ring_dev, n_envs, n_slots, head_slot, filled_slots,
                                                                             batch, seed[0], seed[1], t, sample_base, index_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_replay_sample_multi(const uint8_t* terminal_ring_dev, int n_envs, int n_slots, int head_slot, int filled_slots, int batch,
                                 const uint32_t seed[2], uint64_t t0, int n_updates, uint32_t sample_base, int32_t* index_dev, void* stream) {
    DQ_REQUIRE(terminal_ring_dev && index_dev && seed, DQ_ERR_INVALID, "dq_replay_sample_multi: null argument");
    DQ_REQUIRE(n_envs >= 1 && n_slots >= 4 && batch >= 1 && n_updates >= 1 && head_slot >= 0 && head_slot < n_slots, DQ_ERR_INVALID, "dq_replay_sample_multi: bad sizes");
    DQ_REQUIRE(filled_slots >= DQ_REPLAY_MIN_FILLED && filled_slots <= n_slots, DQ_ERR_STATE, "dq_replay_sample_multi: need at least three complete transitions per lattice");
    DQ_REQUIRE((long long)n_envs * n_slots < (1ll << 31) && (long long)batch * n_updates < (1ll << 31), DQ_ERR_UNSUPPORTED, "dq_replay_sample_multi: too large for 32-bit rows");
    const long long total = (long long)batch * n_updates;
    replay_sample_multi_kernel<<<(unsigned)((total + 255) / 256), 256, 0, (hipStream_t)stream>>>(terminal_ring_dev, n_envs, n_slots, head_slot, filled_slots,
                                                                                              batch, seed[0], seed[1], t0, n_updates, sample_base, index_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_td_target(const float* q_online_s1_dev, const float* q_target_s1_dev, const float* reward_dev, const uint8_t* terminal_dev,
                       const int32_t* index_dev, double gamma, int batch, int n_actions, float* y_dev, void* stream) {
    DQ_REQUIRE(q_online_s1_dev && q_target_s1_dev && reward_dev && terminal_dev && y_dev, DQ_ERR_INVALID, "dq_td_target: null argument");
    DQ_REQUIRE(batch >= 1 && n_actions >= 1, DQ_ERR_INVALID, "dq_td_target: bad sizes");
    td_target_kernel<<<(batch + 3) / 4, 256, 0, (hipStream_t)stream>>>(q_online_s1_dev, q_target_s1_dev, reward_dev, terminal_dev, index_dev,
                                                                       (float)gamma, batch, n_actions, y_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_td_loss_grad(const float* q_s0_dev, const int32_t* action_dev, const int32_t* index_dev, const float* y_dev, int batch,
                          int n_actions, double grad_scale, float* dq_dev, float* metrics_dev, void* stream) {
    DQ_REQUIRE(q_s0_dev && action_dev && y_dev && dq_dev, DQ_ERR_INVALID, "dq_td_loss_grad: null argument");
    DQ_REQUIRE(batch >= 1 && n_actions >= 1, DQ_ERR_INVALID, "dq_td_loss_grad: bad sizes");
    const int blocks = (batch + 3) / 4 < TD_MAX_BLOCKS ? (batch + 3) / 4 : TD_MAX_BLOCKS;
    td_loss_grad_kernel<<<blocks, 256, 0, (hipStream_t)stream>>>(q_s0_dev, action_dev, index_dev, y_dev, batch, n_actions, (float)grad_scale,
                                                                 dq_dev, metrics_dev);
    if (metrics_dev) td_metrics_kernel<<<1, 256, 0, (hipStream_t)stream>>>(metrics_dev, blocks, batch);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_adam_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, size_t n, double lr, double beta_1,
                       double beta_2, double epsilon, uint64_t t, void* stream) {
    DQ_REQUIRE(params_dev && grads_dev && m_dev && v_dev, DQ_ERR_INVALID, "dq_adam_step: null argument");
    DQ_REQUIRE(t >= 1, DQ_ERR_INVALID, "dq_adam_step: t counts from 1");
    const double lr_t = lr * sqrt(1.0 - pow(beta_2, (double)t)) / (1.0 - pow(beta_1, (double)t));
    const size_t threads = (n + 3) / 4;
    adam_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, (hipStream_t)stream>>>(params_dev, grads_dev, m_dev, v_dev, n, (float)lr_t,
                                                                                  (float)beta_1, (float)beta_2, (float)epsilon, nullptr);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

}  // extern "C"

// dq_qnet_adam_step (qnet.hip): dq_adam_step whose skipped elements raise `flag_dev` (the Q-network handle's range flag)
dq_status adam_step_flagged(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, size_t n, double lr, double beta_1, double beta_2,
                            double epsilon, uint64_t t, unsigned* flag_dev, hipStream_t st) {
    DQ_REQUIRE(params_dev && grads_dev && m_dev && v_dev, DQ_ERR_INVALID, "dq_qnet_adam_step: null argument");
    DQ_REQUIRE(t >= 1, DQ_ERR_INVALID, "dq_qnet_adam_step: t counts from 1");
    const double lr_t = lr * sqrt(1.0 - pow(beta_2, (double)t)) / (1.0 - pow(beta_1, (double)t));
    const size_t threads = (n + 3) / 4;
    adam_kernel<<<(unsigned)((threads + 255) / 256), 256, 0, st>>>(params_dev, grads_dev, m_dev, v_dev, n, (float)lr_t, (float)beta_1, (float)beta_2,
                                                                    (float)epsilon, flag_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

extern "C" {

}  // extern "C"
