// Shared helpers of libdeepq_hip (gfx950 only).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "../../include/deepq_hip.h"

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define DQ_WAVE 64

// ---- error plumbing ---------------------------------------------------------------------------
void dq_set_error(const char* fmt, ...);

#define DQ_HIP(call)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (call);                                                              \
        if (e_ != hipSuccess) {                                                              \
            dq_set_error("%s failed: %s (%s:%d)", #call, hipGetErrorString(e_), __FILE__, __LINE__); \
            return DQ_ERR_HIP;                                                               \
        }                                                                                    \
    } while (0)

#define DQ_REQUIRE(cond, code, ...)  \
    do {                             \
        if (!(cond)) {               \
            dq_set_error(__VA_ARGS__); \
            return (code);           \
        }                            \
    } while (0)

#define DQ_LAUNCH_CHECK()                                                           \
    do {                                                                            \
        hipError_t e_ = hipGetLastError();                                          \
        if (e_ != hipSuccess) {                                                     \
            dq_set_error("kernel launch failed: %s (%s:%d)", hipGetErrorString(e_), __FILE__, __LINE__); \
            return DQ_ERR_HIP;                                                      \
        }                                                                           \
    } while (0)

// ---- live kernel timing (prof.hip; dq_prof_arm / dq_prof_collect in the C ABI) ------------------------
enum { DQ_K_ENV = 0, DQ_K_POLICY, DQ_K_CONV_CHAIN, DQ_K_DENSE_CHAIN, DQ_K_GEMM_FWD, DQ_K_GEMM_WGRAD, DQ_K_REDUCE, DQ_K_TD, DQ_K_ADAM,
       DQ_K_DENSE_BWD, DQ_K_DENSE_WGRAD, DQ_K_CONV_BWD, DQ_K_COUNT };
void dq_prof_begin(int kernel_id, hipStream_t st);
void dq_prof_end(int kernel_id, hipStream_t st);
// The exact form, for the launches of the fused chains and the environment step: the launch itself carries the event pair (hipExtLaunchKernelGGL: both
// events are bound to the dispatch packet, so their distance is the packet's own start -> end timestamps -- the duration rocprofv3 reports; an event
// recorded in front of and one behind the launch also measure the two marker packets, ~3.5 us on a 42 us kernel).
int dq_prof_pair(int kernel_id, hipEvent_t* start, hipEvent_t* stop);      // 1: this launch is timed -> its pair (counted); 2: DQ_PROF_BRACKET=1, the bracketing
                                                                            // form for comparison; 0: an ordinary launch
void dq_prof_note_symbol(int kernel_id, const char* symbol);                // the kernel symbol a family's last launch used (dq_prof_kernel_symbol)
#ifdef __HIPCC__
#include <hip/hip_ext.h>
template <typename K, typename... A>
static inline void dq_launch(int kernel_id, const char* symbol, K kern, dim3 grid, dim3 block, size_t lds, hipStream_t st, A... args) {
    hipEvent_t e0, e1;
    dq_prof_note_symbol(kernel_id, symbol);
    const int how = dq_prof_pair(kernel_id, &e0, &e1);
    if (how == 1) hipExtLaunchKernelGGL(kern, grid, block, lds, st, e0, e1, 0, args...);
    else {
        if (how == 2) dq_prof_begin(kernel_id, st);
        kern<<<grid, block, lds, st>>>(args...);
        if (how == 2) dq_prof_end(kernel_id, st);
    }
}
#endif

// W / 2^32 < p  <=>  W < ceil(p * 2^32)   (oracle/philox.py threshold())
static inline u64 dq_rate_threshold(double p) {
    double t = p * 4294967296.0;
    if (!(t > 0.0)) return 0;
    if (t >= 4294967296.0) return 1ull << 32;
    u64 f = (u64)t;
    return ((double)f < t) ? f + 1 : f;
}
// the same for a 16-bit draw (dropout: eight decisions per Philox call):  H / 2^16 < p  <=>  H < ceil(p * 2^16)
static inline u32 dq_rate_threshold16(double p) {
    double t = p * 65536.0;
    if (!(t > 0.0)) return 0;
    if (t >= 65536.0) return 65536u;
    u32 f = (u32)t;
    return ((double)f < t) ? f + 1 : f;
}

// ---- device helpers -----------------------------------------------------------------------------
#ifdef __HIPCC__

// Philox4x32-10 (Salmon et al., SC'11).  Counter words c0..c3, key k0,k1.
__device__ __forceinline__ void philox4x32_10(u32 c0, u32 c1, u32 c2, u32 c3, u32 k0, u32 k1, u32 (&out)[4]) {
#pragma unroll
    for (int r = 0; r < 10; ++r) {
        // one 32 x 32 -> 64 product per multiplier (v_mad_u64_u32): __umulhi and the low product as two expressions compile to two
        // quarter-rate multiplies each (the dense forward's dropout spent 11K cycles per wave in them)
        const u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;
        const u32 hi0 = (u32)(p0 >> 32), lo0 = (u32)p0, hi1 = (u32)(p1 >> 32), lo1 = (u32)p1;
        c0 = hi1 ^ c1 ^ k0;
        c1 = lo1;
        c2 = hi0 ^ c3 ^ k1;
        c3 = lo0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

// Broadcast lane `src` (compile-time or wave-uniform) of a 64-bit value to the whole wave (SGPR pair).
__device__ __forceinline__ u64 wave_bcast64(u64 v, int src) {
    const u32 lo = __builtin_amdgcn_readlane((int)(u32)v, src);
    const u32 hi = __builtin_amdgcn_readlane((int)(u32)(v >> 32), src);
    return ((u64)hi << 32) | lo;
}

__device__ __forceinline__ u64 wave_uniform64(u64 v) {
    const u32 lo = __builtin_amdgcn_readfirstlane((int)(u32)v);
    const u32 hi = __builtin_amdgcn_readfirstlane((int)(u32)(v >> 32));
    return ((u64)hi << 32) | lo;
}

// Kernel attributes (hipFuncSetAttribute) are PER DEVICE: a process that uses several GPUs must set them on each.  Bit of the current
// device in a call site's "already set" mask.
static inline unsigned long long dq_device_bit() { int dev = 0; (void)hipGetDevice(&dev); return 1ull << (dev & 63); }

__device__ __forceinline__ int lane_id() { return (int)(threadIdx.x & 63u); }

// ---- cross-lane steps on the data-parallel-primitive path (DPP: a source-lane permutation inside the VALU instruction, ~one issue slot)
// instead of ds_bpermute (__shfl_xor: a round trip through the LDS crossbar, ~100 cycles of latency per dependent step; hipcc emits 107 of
// them in the dense backward).  Within a row of 16 lanes: quad_perm [1,0,3,2] and [2,3,0,1] are the xor-1 / xor-2 exchanges; after them
// every quad is uniform, so row_half_mirror (i <-> 7 - i) pairs the quads of an 8-group and row_mirror (i <-> 15 - i) the 8-groups:
// four steps make a row uniform.  The four rows are then combined from v_readlane (SGPR) copies.
#define DQ_DPP_XOR1 0xB1
#define DQ_DPP_XOR2 0x4E
#define DQ_DPP_HALF_MIRROR 0x141
#define DQ_DPP_MIRROR 0x140
template <int CTRL> __device__ __forceinline__ int dq_dpp_i(int v) { return __builtin_amdgcn_update_dpp(v, v, CTRL, 0xf, 0xf, false); }
template <int CTRL> __device__ __forceinline__ float dq_dpp_f(float v) { return __builtin_bit_cast(float, dq_dpp_i<CTRL>(__builtin_bit_cast(int, v))); }

// max over the wave's 64 lanes (wave-uniform result)
__device__ __forceinline__ float dq_wave_max(float v) {
    v = fmaxf(v, dq_dpp_f<DQ_DPP_XOR1>(v));
    v = fmaxf(v, dq_dpp_f<DQ_DPP_XOR2>(v));
    v = fmaxf(v, dq_dpp_f<DQ_DPP_HALF_MIRROR>(v));
    v = fmaxf(v, dq_dpp_f<DQ_DPP_MIRROR>(v));
    const int iv = __builtin_bit_cast(int, v);
    const float r0 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)), r1 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16));
    const float r2 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)), r3 = __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48));
    return fmaxf(fmaxf(r0, r1), fmaxf(r2, r3));
}

// First maximum (value, index) over the candidates of a set of lanes; index 0x7fffffff = "no candidate".  The combination rule -- larger
// value, the smaller index among equals -- is commutative and associative: any exchange order gives the bits of the butterfly it replaces.
__device__ __forceinline__ void dq_argmax_take(float& best, int& best_a, float ov, int oa) {
    if (oa != 0x7fffffff && (best_a == 0x7fffffff || ov > best || (ov == best && oa < best_a))) { best = ov; best_a = oa; }
}
// ... within each row of 16 lanes (every lane of the row ends with the row's result)
__device__ __forceinline__ void dq_row_argmax(float& best, int& best_a) {
    dq_argmax_take(best, best_a, dq_dpp_f<DQ_DPP_XOR1>(best), dq_dpp_i<DQ_DPP_XOR1>(best_a));
    dq_argmax_take(best, best_a, dq_dpp_f<DQ_DPP_XOR2>(best), dq_dpp_i<DQ_DPP_XOR2>(best_a));
    dq_argmax_take(best, best_a, dq_dpp_f<DQ_DPP_HALF_MIRROR>(best), dq_dpp_i<DQ_DPP_HALF_MIRROR>(best_a));
    dq_argmax_take(best, best_a, dq_dpp_f<DQ_DPP_MIRROR>(best), dq_dpp_i<DQ_DPP_MIRROR>(best_a));
}
// ... over the whole wave (wave-uniform result)
__device__ __forceinline__ void dq_wave_argmax(float& best, int& best_a) {
    dq_row_argmax(best, best_a);
    const int ib = __builtin_bit_cast(int, best);
    float v = __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, 0));
    int a = __builtin_amdgcn_readlane(best_a, 0);
#pragma unroll
    for (int r = 16; r < 64; r += 16)
        dq_argmax_take(v, a, __builtin_bit_cast(float, __builtin_amdgcn_readlane(ib, r)), __builtin_amdgcn_readlane(best_a, r));
    best = v; best_a = a;
}

// index of the k-th (0-based) set bit of a 128-bit mask; -1 if fewer bits are set
// (round 6: constant time, no loop -- the word by three comparisons against the words' running bit counts, the bit inside it by a binary search on the bit counts of
// halves; "drop the k lowest set bits" was up to 50 trips of 64-bit arithmetic per lattice, and the two lattices of a wave waited for the longer one)
__device__ __forceinline__ int kth_set_bit128(u64 lo, u64 hi, int k) {
    const u32 w0 = (u32)lo, w1 = (u32)(lo >> 32), w2 = (u32)hi, w3 = (u32)(hi >> 32);
    const int c0 = __popc(w0), c1 = c0 + __popc(w1), c2 = c1 + __popc(w2), c3 = c2 + __popc(w3);
    if (k < 0 || k >= c3) return -1;
    const bool s1 = k >= c0, s2 = k >= c1, s3 = k >= c2;
    u32 m = s3 ? w3 : s2 ? w2 : s1 ? w1 : w0;
    int kk = k - (s3 ? c2 : s2 ? c1 : s1 ? c0 : 0), pos = 32 * ((int)s1 + (int)s2 + (int)s3);
#pragma unroll
    for (int h = 16; h >= 1; h >>= 1) {
        const int t = __popc(m & ((1u << h) - 1u));
        const bool up = kk >= t;
        kk -= up ? t : 0; pos += up ? h : 0; m = up ? m >> h : m;
    }
    return pos;
}

// Replay row of minibatch sample `sample_id` of update t (shared by every launch that carries the sampling along).
// Upstream keras-rl 0.4.2 SequentialMemory.sample (restated in oracle/memory_oracle.py): idx = sample_batch_indexes(window_length,
// nb_entries - 1) + 1, i.e. idx in [2, nb_entries - 1] for window_length 1; the experience is (obs[idx-1], action[idx-1],
// reward[idx-1], obs[idx], terminal[idx-1]); an idx with terminals[idx-2] set is redrawn from the same range (its s0 would be the
// terminal observation the agent only looked at before env.reset()).  The entry appended at this step is counted in nb_entries, its
// successor observation is not in keras-rl's memory yet: the newest transition is never sampled, nor is entry 0 (whose predecessor's
// terminal flag is unknown).  Here `filled` slots hold observations, the newest in slot head_slot, so nb_entries = filled - 1 and entry
// k lives in slot oldest + k, oldest = head - (filled - 1).  Transition idx - 1 in [1, nb_entries - 2]  <=>  slot in [oldest + 1,
// head - 2]: filled - 3 candidates, newest first.  The sampler reads terminal[] only at slots <= head - 3 and the newest row it can
// return is head - 2, so an update's minibatch can be drawn one vector step early (on the environment launch that WRITES slot head - 2
// of the ring as the update will see it) and does not depend on the environment step taken in the update's own vector step.
// DISTINCT ROWS (round 3).  keras-rl's first draw is random.sample(range(low, high), batch_size) -- without replacement -- whenever the
// range holds at least batch_size indexes (else np.random integers with replacement and a warning); only the REDRAW of an idx behind a
// terminal is a fresh independent draw (which may repeat another sample's row).  Over the N lattices of the ring the candidates are the
// M = (filled - 3) * n_envs rows (candidate j of lattice e = flat index j * n_envs + e).  With batch <= M the first draw of sample
// `sample_id` is  pi_t(sample_id mod M),  pi_t a bijection of [0, M) keyed by (seed, t): a four-round unbalanced Feistel network over
// the next power of two (round keys = the four Philox words of the update's key counter; round function = the murmur3 finaliser),
// cycle-walked back into [0, M) -- a minibatch's first draws are distinct rows, each row equally likely, and the row still depends
// only on (seed, t, global sample id).  Redraws (attempt >= 1), and every draw when batch > M, are the independent Philox draws of
// rounds 1-2.  Restated in oracle/memory_oracle.py device_replay_rows.
#define DQ_REPLAY_MIN_FILLED 4                          // nb_entries >= window_length + 2
__device__ __forceinline__ u32 dq_mix32(u32 h) {        // murmur3 fmix32
    h ^= h >> 16; h *= 0x85ebca6bu; h ^= h >> 13; h *= 0xc2b2ae35u; h ^= h >> 16;
    return h;
}
// pi_t(x) for x in [0, M), M >= 1; k[4] = the round keys
__device__ __forceinline__ u32 dq_replay_permute(u32 x, u32 M, const u32 k[4]) {
    if (M < 2u) return 0u;
    const int bits = 32 - __clz((int)(M - 1u));         // 2^(bits-1) < M <= 2^bits
    const int a = bits >> 1, b = bits - a;              // low half: a bits, high half: b bits (a may be 0)
    const u32 ma = (1u << a) - 1u, mb = (1u << b) - 1u;
    u32 v = x;
    do {
        u32 hi = v >> a, lo = v & ma;
        hi ^= dq_mix32(lo ^ k[0]) & mb;
        lo ^= dq_mix32(hi ^ k[1]) & ma;
        hi ^= dq_mix32(lo ^ k[2]) & mb;
        lo ^= dq_mix32(hi ^ k[3]) & ma;
        v = hi << a | lo;
    } while (v >= M);                                   // cycle walking: x lies on the cycle, so this ends; < 2 trips on average
    return v;
}
__device__ __forceinline__ int dq_replay_row(const u8* __restrict__ terminal, int n_envs, int n_slots, int head_slot, int filled, int batch,
                                             u32 seed0, u32 seed1, u64 t, u32 sample_id) {
    const int cand = filled - 3;
    const u32 M = (u32)cand * (u32)n_envs;
    const bool distinct = (u32)batch <= M;
    int row = 0;
    for (u32 attempt = 0; attempt < 64; ++attempt) {
        int j, env;
        if (attempt == 0 && distinct) {
            u32 k[4];
            philox4x32_10((u32)t, (u32)(t >> 32), 0xffffffffu, 0xffffu | ((u32)DQ_STREAM_REPLAY << 16), seed0, seed1, k);
            const u32 flat = dq_replay_permute(sample_id % M, M, k);
            j = (int)(flat / (u32)n_envs);
            env = (int)(flat - (u32)j * (u32)n_envs);
        } else {
            u32 w[4];
            philox4x32_10((u32)t, (u32)(t >> 32), sample_id, attempt | ((u32)DQ_STREAM_REPLAY << 16), seed0, seed1, w);
            j = (int)__umulhi(w[0], (u32)cand);         // 0 = newest candidate
            env = (int)__umulhi(w[1], (u32)n_envs);
        }
        int slot = head_slot - 2 - j;
        if (slot < 0) slot += n_slots;
        row = slot * n_envs + env;
        int prev = slot - 1;                            // >= oldest: always a stored entry
        if (prev < 0) prev += n_slots;
        if (!terminal[(size_t)prev * n_envs + env]) break;
    }
    return row;
}

// Episode bookkeeping of lattice i (one thread per lattice, workgroups of up to 8 waves, every thread calls): stats[0] += #episodes that ended
// this step, stats[1] += sum of their lifetimes, stats[2] += #rewards == 1, stats[3] += #lattices stepped (not reset).  Ballots per
// wave, the waves combined through LDS, then at most four atomics per workgroup (atomics on one word serialise at ~10 ns each:
// per-wave atomics cost the TD launch 2.5 us).  Integer sums => order-independent.
__device__ __forceinline__ void dq_episode_stats_lane(const u8* __restrict__ done, const u8* __restrict__ was_reset,
                                                      const u32* __restrict__ lifetime, const float* __restrict__ reward, int n, int i,
                                                      unsigned long long* __restrict__ stats) {
    __shared__ unsigned long long s_st[8][4];
    const bool in = i < n;
    const bool stepped = in && !(was_reset && was_reset[i]);
    const bool ended = stepped && done[i];
    const u64 m_end = __ballot(ended), m_rew = __ballot(stepped && reward[i] > 0.5f), m_step = __ballot(stepped);
    unsigned long long life = ended ? lifetime[i] : 0;
    for (int m = 32; m >= 1; m >>= 1) life += __shfl_xor(life, m);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0 && wave < 8) {
        s_st[wave][0] = (unsigned long long)__popcll(m_end); s_st[wave][1] = life;
        s_st[wave][2] = (unsigned long long)__popcll(m_rew); s_st[wave][3] = (unsigned long long)__popcll(m_step);
    }
    __syncthreads();
    const int nw = (blockDim.x + 63) >> 6;
    if (threadIdx.x < 4) {
        unsigned long long v = 0;
        for (int w = 0; w < nw && w < 8; ++w) v += s_st[w][threadIdx.x];
        if (v) atomicAdd(&stats[threadIdx.x], v);
    }
}

// Keras 2.2 Adam.get_updates for one parameter: m = b1 m + (1-b1) g; v = b2 v + (1-b2) g^2; p -= lr_t m / (sqrt(v) + eps).
// One definition for adam_kernel (dqn.hip) and the fused backward's final reduction (fused_bwd.hip): the two give the same bits.
// (No multiply-add contraction inside: whether hipcc fuses depends on the surrounding code, and the bits must not.)
__device__ __forceinline__ void dq_adam1(float& p, float g, float& m, float& v, float lr_t, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    m = b1 * m + (1.f - b1) * g;
    v = b2 * v + (1.f - b2) * g * g;
    p = p - lr_t * m / (sqrtf(v) + eps);
}

// ---- optional phase timestamps (development aid): build with DQ_EXTRA_FLAGS="-DDQ_STAMPS=<kernel tag>" and read them back with
// tools/stamp_run.py.  Workgroup DQ_STAMP_BLOCK, lane 0 of every wave, records the shader cycle counter at phase boundaries.
#define DQ_TAG_CONV_FWD 1
#define DQ_TAG_DENSE_FWD 2
#define DQ_TAG_DENSE_BWD 3
#define DQ_TAG_CONV_BWD 4
#define DQ_TAG_DENSE_WGRAD 5
#define DQ_TAG_ENV 6                     // env_block2 riding on the dense backward (fused_bwd.hip; build with -DDQ_STAMP_BLOCK=<a rider block>)
#ifdef DQ_STAMPS
#ifndef DQ_STAMP_BLOCK
#define DQ_STAMP_BLOCK 9
#endif
static __device__ unsigned long long dq_dbg[4096];            // one copy per translation unit (no relocatable device code)
#define DQ_STAMP_READER(name) extern "C" void name(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(dq_dbg), sizeof(dq_dbg)); }
#define DQ_STAMP(tag, i)                                                                                        \
    do {                                                                                                        \
        if ((tag) == DQ_STAMPS && blockIdx.x == DQ_STAMP_BLOCK && (threadIdx.x & 63) == 0)                      \
            dq_dbg[(i) * 8 + (threadIdx.x >> 6)] = __builtin_readcyclecounter();                                \
    } while (0)
// tag + 10: wall-clock (100 MHz) start / end of the first 256 workgroups instead (dispatch spread, launch ramp and drain)
#define DQ_STAMP_WG(tag, end)                                                                                   \
    do {                                                                                                        \
        if ((tag) + 10 == DQ_STAMPS && blockIdx.x < 256 && threadIdx.x == 0)                                    \
            dq_dbg[(end) * 256 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();                                \
    } while (0)
// DQ_STAMPS == 20: wall-clock start / end of workgroups 0..1023 of two consecutive kernels (quarter q = 2 * kernel + end): the gap
// between one kernel's last workgroup and the next kernel's first
#define DQ_STAMP_PAIR(q)                                                                                        \
    do {                                                                                                        \
        if (DQ_STAMPS == 20 && blockIdx.x < 1024 && threadIdx.x == 0)                                           \
            dq_dbg[(q) * 1024 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();                                 \
    } while (0)
// DQ_STAMPS == 21: the same across the backward's launches (fused_bwd.hip): q = 0 dense data gradients end, 1 / 2 dense weight gradients
// start / end, 3 convolutional backward start
#define DQ_STAMP_PAIR2(q)                                                                                       \
    do {                                                                                                        \
        if (DQ_STAMPS == 21 && blockIdx.x < 1024 && threadIdx.x == 0)                                           \
            dq_dbg[(q) * 1024 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();                                 \
    } while (0)
// DQ_STAMPS == 23: wall-clock start / end of EVERY workgroup (< 1024) of the dense backward's launch, riders included (tools/stamp_loop.py)
#define DQ_STAMP_ALL(end)                                                                                       \
    do {                                                                                                        \
        if (DQ_STAMPS == 23 && blockIdx.x < 1024 && threadIdx.x == 0)                                           \
            dq_dbg[(end) * 1024 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();                               \
    } while (0)
#else
#define DQ_STAMP_ALL(end) do { } while (0)
#define DQ_STAMP_PAIR2(q) do { } while (0)
#define DQ_STAMP(tag, i) do { } while (0)
#define DQ_STAMP_WG(tag, end) do { } while (0)
#define DQ_STAMP_PAIR(q) do { } while (0)
#define DQ_STAMP_READER(name)
#endif

#endif  // __HIPCC__
