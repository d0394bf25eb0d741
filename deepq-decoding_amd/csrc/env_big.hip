// Batched surface-code environment for lattices beyond one 64-bit word per bit-plane (d >= 9; any odd 3 <= d <= 15): the "wide" form of
// env.hip / env_dev.h -- same reference (Surface_Code_Environment_Multi_Decoding_Cycles, /root/reference/example_notebooks/Environments.py:
// 10-385, + Function_Library.py:67-326), same formulation (Pauli codes multiply as XOR: two qubit bit-planes; one lattice per wavefront;
// Philox4x32-10 keyed by (round, global lattice id, site); syndromes by popcount + ballot), same semantics call for call (auto-reset,
// sticky done, rejection loop, legal moves), with every plane W = ceil(d^2 / 64) words wide (lane l owns sites l, l + 64, ...), action sets
// LW = ceil(|A| / 64) words wide, and the referee computed per step by the matching decoder (match_dev.h) because a look-up table over
// 2^((d^2-1)/2) syndromes per component no longer exists.  At d <= 7 it reproduces env.hip bit for bit (the reference's golden traces are
// replayed through both; tests/test_env_gpu.py).  Written for correctness first: the lattice record lives in LDS and is updated in place;
// the headline configurations (d = 5, 7) keep env.hip's register-resident kernel.
//
// Record per lattice (uint64 words): x[W] z[W] acted[W] round meta(lifetime | done << 32) completed[LW] legal[LW] volume[depth][W].
// Export (dq_envb_export_state): x[W] z[W] true_syndrome[W] summed[W] acted[W] round completed[LW] legal[LW] meta volume[depth][W].
#include <new>
#include "match_dev.h"
#include "lattice_host.h"

#define BIG_MAX_W 4
#define BIG_EPB 4                       // lattices (waves) per block

struct BigTables {
    const u64* stab_q;                  // [n_stab][W]  qubits of stabilizer s (measurement order)
    const u64* qubit_s;                 // [d2][W]      live stabilizers touched by qubit q (ENV:262-271)
    const u64* neigh;                   // [d2][W]      8-neighbourhood of qubit q (ENV:349-372)
    const u8* stab_isx;                 // [n_stab]     1: type-3 plaquette (parity of the X component)
    const unsigned short* typed;        // [2][128]     stabilizer at position p of component c's row-major order (0xffff: none)
    const u64* col0;                    // [W] FL:312-314
    const u64* row0;                    // [W] FL:315-317
    const u8* cell_static;              // [P] padding_syndrome decoration (ENV:284-298)
    const unsigned short* cell_stab;    // [P] stabilizer shown at an even-even cell (ENV:292-294), 0xffff: none
    const unsigned short* cell_qubit;   // [P] qubit shown at an odd-odd cell of an action plane (ENV:301-314), 0xffff: none
};

struct BigParams {
    BigTables tab;
    MatchComp mx, mz;
    u64* state;
    int n_envs, d2, n_stab, depth, layers, n_actions, identity, model, use_Y, sw, P, W, LW, obs_size;
    u32 env_id_base, seed0, seed1;
    u64 T_phys, T_meas;
    int mode, auto_reset;               // mode 0: reset, 1: step
    const u8* which;
    const int32_t* action;
    u8* obs;
    float* reward;
    u8* done;
    u64* legal;                         // [n_envs][LW]
    u32* lifetime;
    u8* was_reset;
    u8* inexact;                        // [n_envs] or NULL: 1 where the referee's fallback was used in this step
    // fused action selection (the rule of policy.hip, one lattice per wave)
    int policy;
    const float* q;
    u64 T_eps, pt;
    int masked_greedy;
    u32 pseed0, pseed1;
    int32_t* action_out;
};

static inline size_t big_wave_lds(int sw) { return (size_t)((sw * 8 + 15) & ~15) + DQ_MATCH_LDS; }

template <int W>
__global__ __launch_bounds__(64 * BIG_EPB) void env_big_kernel(BigParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int i = blockIdx.x * BIG_EPB + wave;
    if (i >= p.n_envs) return;                                       // wave-uniform; no block-wide barrier below
    const size_t wave_lds = (size_t)((p.sw * 8 + 15) & ~15) + DQ_MATCH_LDS;
    volatile u64* st = reinterpret_cast<volatile u64*>(smem + wave * wave_lds);
    u8* s_match = smem + wave * wave_lds + ((p.sw * 8 + 15) & ~15);
    const int LW = p.LW, depth = p.depth;
    const int O_X = 0, O_Z = W, O_ACT = 2 * W, O_ROUND = 3 * W, O_META = 3 * W + 1, O_COMP = 3 * W + 2, O_LEGAL = O_COMP + LW, O_VOL = O_LEGAL + LW;
    u64* rec = p.state + (size_t)i * p.sw;
    for (int k = lane; k < p.sw; k += 64) st[k] = rec[k];
    match_wave_sync();                                               // the record is handed between lanes through LDS: fences, not `volatile` alone (match_dev.h)
    u64 x[W], z[W];
#pragma unroll
    for (int w = 0; w < W; ++w) { x[w] = st[O_X + w]; z[w] = st[O_Z + w]; }
    u64 round = st[O_ROUND];
    const u64 meta = st[O_META];
    u32 lifetime = (u32)meta;
    int done = (int)((meta >> 32) & 1);

    // this lane's sites: site(w) = lane + 64 w
    auto syndrome = [&](u64 (&tw)[W]) {                               // ENV:139 / ENV:165, FL:152-174
#pragma unroll
        for (int w = 0; w < W; ++w) {
            const int s = lane + 64 * w;
            int par = 0;
            if (s < p.n_stab) {
                const bool isx = p.tab.stab_isx[s] != 0;
#pragma unroll
                for (int v = 0; v < W; ++v) par += __popcll((isx ? x[v] : z[v]) & p.tab.stab_q[(size_t)s * W + v]);
            }
            tw[w] = __ballot(par & 1);
        }
    };

    bool do_reset;
    if (p.mode == 0) do_reset = p.which ? (__builtin_amdgcn_readfirstlane((int)p.which[i]) != 0) : true;
    else do_reset = p.auto_reset && done;
    const bool do_step = p.mode == 1 && !do_reset;
    float reward = 0.f;
    bool need_volume = do_reset;
    int flag = 0;
    if (do_reset) {                                                  // ENV:106-107, 211-213
        done = 0; lifetime = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) { x[w] = 0; z[w] = 0; }
    }
    int a_sel = 0;
    if (p.policy) {                                                  // EpsGreedyQPolicy / GreedyQPolicy(masked_greedy), see policy.hip
        u32 wd[4];
        philox4x32_10((u32)p.pt, (u32)(p.pt >> 32), p.env_id_base + (u32)i, (u32)DQ_STREAM_POLICY << 16, p.pseed0, p.pseed1, wd);
        if (p.q == nullptr || (u64)wd[1] < p.T_eps) {                // explore: k-th smallest legal action
            int n_legal = 0;
            for (int k = 0; k < LW; ++k) n_legal += __popcll(st[O_LEGAL + k]);
            int kk = (int)__umulhi(wd[0], (u32)n_legal);
            a_sel = -1;
            for (int k = 0; k < LW && a_sel < 0; ++k) {
                u64 m = st[O_LEGAL + k];
                const int c = __popcll(m);
                if (kk < c) {
                    for (int t = 0; t < kk; ++t) m &= m - 1;
                    a_sel = 64 * k + __ffsll((long long)m) - 1;
                } else kk -= c;
            }
        } else {                                                     // first maximum of the Q row (optionally over the legal set)
            const float* row = p.q + (size_t)i * p.n_actions;
            float best = -INFINITY;
            int best_a = 0x7fffffff;
            for (int k = lane; k < p.n_actions; k += 64) {
                const bool ok = !p.masked_greedy || ((st[O_LEGAL + (k >> 6)] >> (k & 63)) & 1);
                const float v = row[k];
                if (ok && (v > best || best_a == 0x7fffffff)) { best = v; best_a = k; }
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {
                const float ov = __shfl_xor(best, m);
                const int oa = __shfl_xor(best_a, m);
                if (oa != 0x7fffffff && (best_a == 0x7fffffff || ov > best || (ov == best && oa < best_a))) { best = ov; best_a = oa; }
            }
            a_sel = best_a;
        }
        a_sel = __builtin_amdgcn_readfirstlane(a_sel);
        if (lane == 0) p.action_out[i] = a_sel;
    }
    if (do_step) {
        int a = p.policy ? a_sel : __builtin_amdgcn_readfirstlane(p.action[i]);
        if ((unsigned)a >= (unsigned)p.n_actions) a = p.identity;
        const bool done_identity = a == p.identity || ((st[O_COMP + (a >> 6)] >> (a & 63)) & 1);      // ENV:131
        if (a != p.identity) {                                       // ENV:135-136, FL:243-294
            const int layer = a / p.d2, q = a - layer * p.d2;
            const int pauli = p.model == DQ_MODEL_X ? 1 : (p.use_Y ? layer + 1 : (layer == 0 ? 1 : 3));
#pragma unroll
            for (int w = 0; w < W; ++w)
                if (w == (q >> 6)) {
                    if (pauli != 3) x[w] ^= 1ull << (q & 63);
                    if (pauli != 1) z[w] ^= 1ull << (q & 63);
                }
        }
        u64 tw[W];
        syndrome(tw);
        int px = 0, pz = 0;
        u64 any = 0;
#pragma unroll
        for (int w = 0; w < W; ++w) { px += __popcll(x[w] & p.tab.col0[w]); pz += __popcll(z[w] & p.tab.row0[w]); any |= tw[w]; }
        const int cls = (px & 1) + 2 * (pz & 1);                     // ENV:143
        // the referee (ENV:144): defects of each component in its own row-major order, then the matching decoder
        u64 df[2][2];
#pragma unroll
        for (int comp = 0; comp < 2; ++comp)
#pragma unroll
            for (int ww = 0; ww < 2; ++ww) {
                const int s = p.tab.typed[comp * 128 + ww * 64 + lane];
                bool bit = false;
                if (s != 0xffff) {
                    u64 word = 0;
#pragma unroll
                    for (int w = 0; w < W; ++w) word = (s >> 6) == w ? tw[w] : word;
                    bit = (word >> (s & 63)) & 1;
                }
                df[comp][ww] = __ballot(bit);
            }
        int dec = match_classify(p.mx, df[0][0], df[0][1], s_match, lane, &flag);
        if (p.model != DQ_MODEL_X) dec += 2 * match_classify(p.mz, df[1][0], df[1][1], s_match, lane, &flag);
        if (cls == 0 && any == 0) reward = 1.f;                      // ENV:148-149
        else if (dec != cls) done = 1;                               // ENV:150-151
        if (done_identity) {
            need_volume = true;                                      // ENV:155
        } else {                                                     // ENV:185-196
            const int q = a % p.d2;
            const bool fresh = !((st[O_ACT + (q >> 6)] >> (q & 63)) & 1);
            match_wave_sync();                                       // all lanes have read `acted` before lane 0 sets the bit
            if (lane == 0) {
                st[O_COMP + (a >> 6)] = st[O_COMP + (a >> 6)] | (1ull << (a & 63));
                if (fresh) st[O_ACT + (q >> 6)] = st[O_ACT + (q >> 6)] | (1ull << (q & 63));
            }
            if (fresh) {                                             // every action on a neighbour of q becomes legal, in every layer
                for (int k = 0; k < LW; ++k) {
                    const int t = 64 * k + lane;
                    bool bit = false;
                    if (t < p.layers * p.d2) {
                        const int qq = t % p.d2;
                        bit = (p.tab.neigh[(size_t)q * W + (qq >> 6)] >> (qq & 63)) & 1;
                    }
                    const u64 add = __ballot(bit);
                    if (lane == 0) st[O_LEGAL + k] = st[O_LEGAL + k] | add;
                }
            }
        }
    }
    if (need_volume) {                                               // ENV:157-172 == ENV:216-231
        u64 summed[W];
        u64 any;
        do {
#pragma unroll
            for (int w = 0; w < W; ++w) summed[w] = 0;
            for (int j = 0; j < depth; ++j) {
                u64 ex[W], ez[W], fl[W];
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const int site = lane + 64 * w;
                    u32 wd[4];
                    philox4x32_10((u32)round, (u32)(round >> 32), p.env_id_base + (u32)i, (u32)site, p.seed0, p.seed1, wd);
                    const bool hit = site < p.d2 && (u64)wd[0] < p.T_phys;  // FL:99 / FL:119
                    const int typ = p.model == DQ_MODEL_X ? 1 : 1 + (int)__umulhi(wd[1], 3u);   // FL:100
                    const bool zhit = site < p.d2 && (u64)wd[1] < p.T_phys; // IIDXZ (FL:134-160): the second uniform is an independent Z flip
                    ex[w] = __ballot(p.model == DQ_MODEL_IIDXZ ? hit : hit && typ != 3);
                    ez[w] = __ballot(p.model == DQ_MODEL_IIDXZ ? zhit : hit && typ != 1);
                    fl[w] = __ballot(site < p.n_stab && (u64)wd[2] < p.T_meas);   // FL:191-221
                }
                ++round;
#pragma unroll
                for (int w = 0; w < W; ++w) { x[w] ^= ex[w]; z[w] ^= ez[w]; }     // ENV:164, FL:226-241
                u64 tw[W];
                syndrome(tw);                                        // ENV:165
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const u64 v = tw[w] ^ fl[w];                     // ENV:166
                    if (lane == 0) st[O_VOL + j * W + w] = v;
                    summed[w] |= v;                                  // ENV:168
                }
                ++lifetime;                                          // ENV:169
            }
            any = 0;
#pragma unroll
            for (int w = 0; w < W; ++w) any |= summed[w];
        } while (any == 0);                                          // ENV:171
        // reset_legal_moves, ENV:238-258
        if (lane < W) st[O_ACT + lane] = 0;
        for (int k = 0; k < LW; ++k) {
            const int t = 64 * k + lane;
            bool bit = t == p.identity;
            if (t < p.layers * p.d2) {
                const int qq = t % p.d2;
                u64 touch = 0;
#pragma unroll
                for (int w = 0; w < W; ++w) touch |= p.tab.qubit_s[(size_t)qq * W + w] & summed[w];
                bit = touch != 0;
            }
            const u64 lg = __ballot(bit);
            if (lane == 0) { st[O_LEGAL + k] = lg; st[O_COMP + k] = 0; }
        }
    }
    // ---- state record and scalar outputs ------------------------------------------------------------------------------------
    match_wave_sync();                                               // every lane has read the old record before lane 0 overwrites it
    if (lane == 0) {
#pragma unroll
        for (int w = 0; w < W; ++w) { st[O_X + w] = x[w]; st[O_Z + w] = z[w]; }
        st[O_ROUND] = round;
        st[O_META] = (u64)lifetime | ((u64)done << 32);
        if (p.reward) p.reward[i] = reward;
        if (p.done) p.done[i] = (u8)done;
        if (p.lifetime) p.lifetime[i] = lifetime;
        if (p.was_reset) p.was_reset[i] = (u8)(p.mode == 1 && do_reset);
        if (p.inexact) p.inexact[i] = (u8)flag;
    }
    match_wave_sync();                                               // lane 0's updates (and the action / volume sets written above) are visible to all lanes
    for (int k = lane; k < p.sw; k += 64) rec[k] = st[k];
    if (p.legal && lane < LW) p.legal[(size_t)i * LW + lane] = st[O_LEGAL + lane];
    if (!p.obs) return;
    // ---- observation (ENV:174-175, 200-201, 273-314): syndrome planes then action planes, bytes straight to global memory --------
    u8* ob = p.obs + (size_t)i * p.obs_size;
    for (int j = 0; j < depth; ++j)
        for (int c = lane; c < p.P; c += 64) {
            const int sidx = p.tab.cell_stab[c];
            u32 bit = 0;
            if (sidx != 0xffff) bit = (u32)((st[O_VOL + j * W + (sidx >> 6)] >> (sidx & 63)) & 1);
            ob[j * p.P + c] = (u8)(p.tab.cell_static[c] | bit);
        }
    for (int k = 0; k < p.layers; ++k)
        for (int c = lane; c < p.P; c += 64) {
            const int qi = p.tab.cell_qubit[c];
            u32 bit = 0;
            if (qi != 0xffff) {
                const int a = k * p.d2 + qi;
                bit = (u32)((st[O_COMP + (a >> 6)] >> (a & 63)) & 1);
            }
            ob[(depth + k) * p.P + c] = (u8)bit;
        }
}

// export: x z true summed acted round completed legal meta volume
template <int W>
__global__ void env_big_export_kernel(BigParams p, u64* out, int ew) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= p.n_envs) return;
    const u64* rec = p.state + (size_t)i * p.sw;
    const int LW = p.LW;
    const int O_Z = W, O_ACT = 2 * W, O_ROUND = 3 * W, O_META = 3 * W + 1, O_COMP = 3 * W + 2, O_LEGAL = O_COMP + LW, O_VOL = O_LEGAL + LW;
    u64* o = out + (size_t)i * ew;
    u64 summed[W];
#pragma unroll
    for (int w = 0; w < W; ++w) {
        summed[w] = 0;
        for (int j = 0; j < p.depth; ++j) summed[w] |= rec[O_VOL + j * W + w];
        const int s = lane + 64 * w;
        int par = 0;
        if (s < p.n_stab) {
            const bool isx = p.tab.stab_isx[s] != 0;
            for (int v = 0; v < W; ++v) par += __popcll(rec[(isx ? 0 : O_Z) + v] & p.tab.stab_q[(size_t)s * W + v]);
        }
        const u64 tw = __ballot(par & 1);
        if (lane == 0) { o[w] = rec[w]; o[W + w] = rec[O_Z + w]; o[2 * W + w] = tw; o[3 * W + w] = summed[w]; o[4 * W + w] = rec[O_ACT + w]; }
    }
    if (lane == 0) {
        o[5 * W] = rec[O_ROUND];
        for (int k = 0; k < LW; ++k) { o[5 * W + 1 + k] = rec[O_COMP + k]; o[5 * W + 1 + LW + k] = rec[O_LEGAL + k]; }
        o[5 * W + 1 + 2 * LW] = rec[O_META];
        for (int k = 0; k < p.depth * W; ++k) o[5 * W + 2 + 2 * LW + k] = rec[O_VOL + k];
    }
}

// wide action selection: the rule of dq_policy_select over LW-word legal sets, one lattice per wave
__global__ __launch_bounds__(256) void policy_wide_kernel(const float* __restrict__ q, const u64* __restrict__ legal, int n, int n_actions, int LW,
                                                          u64 T_eps, int masked_greedy, u32 seed0, u32 seed1, u32 env_id_base, u64 t,
                                                          int32_t* __restrict__ action) {
    const int i = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n) return;
    const u64* lg = legal + (size_t)i * LW;
    u32 w[4];
    philox4x32_10((u32)t, (u32)(t >> 32), env_id_base + (u32)i, (u32)DQ_STREAM_POLICY << 16, seed0, seed1, w);
    int a = -1;
    if (q == nullptr || (u64)w[1] < T_eps) {
        int n_legal = 0;
        for (int k = 0; k < LW; ++k) n_legal += __popcll(lg[k]);
        int kk = (int)__umulhi(w[0], (u32)n_legal);
        for (int k = 0; k < LW && a < 0; ++k) {
            u64 m = lg[k];
            const int c = __popcll(m);
            if (kk < c) {
                for (int s = 0; s < kk; ++s) m &= m - 1;
                a = 64 * k + __ffsll((long long)m) - 1;
            } else kk -= c;
        }
    } else {
        const float* row = q + (size_t)i * n_actions;
        float best = -INFINITY;
        int best_a = 0x7fffffff;
        for (int k = lane; k < n_actions; k += 64) {
            const bool ok = !masked_greedy || ((lg[k >> 6] >> (k & 63)) & 1);
            const float v = row[k];
            if (ok && (v > best || best_a == 0x7fffffff)) { best = v; best_a = k; }
        }
#pragma unroll
        for (int m = 32; m >= 1; m >>= 1) {
            const float ov = __shfl_xor(best, m);
            const int oa = __shfl_xor(best_a, m);
            if (oa != 0x7fffffff && (best_a == 0x7fffffff || ov > best || (ov == best && oa < best_a))) { best = ov; best_a = oa; }
        }
        a = best_a;
    }
    if (lane == 0) action[i] = a;
}

// ---- host side --------------------------------------------------------------------------------------------------------------
void match_comp(const struct dq_match* M, int comp, MatchComp* out);     // match.hip
dq_status match_reset_locks(const struct dq_match* M, hipStream_t st);   // match.hip: the scratch pool's lock words, in front of every launch that may take a slot

struct dq_envb {
    dq_env_cfg cfg;
    dq_env_info info;
    int W, LW, sw, ew, P;
    u64 T_phys, T_meas;
    bool rates_set;
    u8* d_blob;                         // all tables in one allocation
    BigTables tab;
    dq_match* match;
    u64* d_state;
};

template <typename T>
static size_t blob_put(std::vector<u8>& blob, const std::vector<T>& v) {
    const size_t off = (blob.size() + 15) & ~(size_t)15;
    blob.resize(off + v.size() * sizeof(T));
    memcpy(blob.data() + off, v.data(), v.size() * sizeof(T));
    return off;
}

extern "C" {

void dq_envb_destroy(dq_envb* E) {
    if (!E) return;
    if (E->d_blob) (void)hipFree(E->d_blob);
    if (E->d_state) (void)hipFree(E->d_state);
    if (E->match) dq_match_destroy(E->match);
    delete E;
}

dq_status dq_envb_create(const dq_env_cfg* cfg, dq_envb** out) {
    DQ_REQUIRE(cfg && out, DQ_ERR_INVALID, "dq_envb_create: null argument");
    *out = nullptr;
    DQ_REQUIRE(cfg->d % 2 == 1, DQ_ERR_INVALID, "for the surface code d must be odd!");     // FL:28-29
    DQ_REQUIRE(cfg->d >= 3 && cfg->d <= 15, DQ_ERR_UNSUPPORTED, "d=%d unsupported: the wide environment covers 3 <= d <= 15", cfg->d);
    DQ_REQUIRE(cfg->error_model == DQ_MODEL_X || cfg->error_model == DQ_MODEL_DP || cfg->error_model == DQ_MODEL_IIDXZ, DQ_ERR_UNSUPPORTED,
               "specified error model not currently supported!");                              // ENV:66-67
    DQ_REQUIRE(cfg->volume_depth >= 1 && cfg->volume_depth <= 16, DQ_ERR_UNSUPPORTED, "volume_depth must be in 1..16");
    DQ_REQUIRE(cfg->n_envs >= 1, DQ_ERR_INVALID, "n_envs must be positive");
    dq_envb* E = new (std::nothrow) dq_envb();
    DQ_REQUIRE(E, DQ_ERR_NOMEM, "out of host memory");
    E->cfg = *cfg; E->d_blob = nullptr; E->d_state = nullptr; E->match = nullptr; E->rates_set = false;
    const int d = cfg->d, d2 = d * d, n = 2 * d + 1;
    const int layers = cfg->error_model == DQ_MODEL_X ? 1 : (cfg->use_Y ? 3 : 2);             // ENV:55-65
    E->info.n_action_layers = layers;
    E->info.num_actions = layers * d2 + 1;
    E->info.identity_index = E->info.num_actions - 1;                                         // ENV:69
    E->info.obs_c = cfg->volume_depth + layers;                                               // ENV:78-82
    E->info.obs_h = E->info.obs_w = n;
    E->info.n_stab = d2 - 1;
    E->W = (d2 + 63) / 64; E->LW = (E->info.num_actions + 63) / 64; E->P = n * n;
    E->sw = 3 * E->W + 2 + 2 * E->LW + cfg->volume_depth * E->W;
    E->ew = 5 * E->W + 2 + 2 * E->LW + cfg->volume_depth * E->W;
    E->info.state_words = E->ew;
    LatticeHost L;
    lattice_build(d, &L);
    const int W = E->W, ns = L.n_stab;
    std::vector<u64> stab_q((size_t)ns * W, 0), qubit_s((size_t)d2 * W, 0), neigh((size_t)d2 * W, 0), col0(W, 0), row0(W, 0);
    std::vector<u8> isx(ns, 0), cstatic(E->P, 0);
    std::vector<unsigned short> typed(256, 0xffff), cstab(E->P, 0xffff), cqubit(E->P, 0xffff);
    for (int s = 0; s < ns; ++s) {
        isx[s] = L.stab_type[s] == 3;
        for (int q : L.stab_qubits[s]) { stab_q[(size_t)s * W + (q >> 6)] |= 1ull << (q & 63); qubit_s[(size_t)q * W + (s >> 6)] |= 1ull << (s & 63); }
        cstab[2 * L.sa[s] * n + 2 * L.sb[s]] = (unsigned short)s;                             // ENV:292-294
    }
    for (int comp = 0; comp < 2; ++comp)
        for (size_t k = 0; k < L.typed[comp].size(); ++k) typed[comp * 128 + k] = (unsigned short)L.typed[comp][k];
    for (int q = 0; q < d2; ++q) {
        for (int m : L.neigh[q]) neigh[(size_t)q * W + (m >> 6)] |= 1ull << (m & 63);
        cqubit[(2 * (q / d) + 1) * n + 2 * (q % d) + 1] = (unsigned short)q;                  // ENV:309-312
    }
    for (int k = 0; k < d; ++k) { const int qc = k * d, qr = k; col0[qc >> 6] |= 1ull << (qc & 63); row0[qr >> 6] |= 1ull << (qr & 63); }
    for (int xx = 0; xx < n; ++xx) for (int yy = 0; yy < n; ++yy) {                          // ENV:284-298
        u8 v = 0;
        if ((xx == 0 || xx == n - 1) && (yy & 1)) v = 1;
        if ((yy == 0 || yy == n - 1) && (xx & 1)) v = 1;
        if ((xx & 1) && (yy & 1) && ((xx + yy) % 4 == 0)) v = 1;
        cstatic[xx * n + yy] = v;
    }
    std::vector<u8> blob;
    const size_t o_sq = blob_put(blob, stab_q), o_qs = blob_put(blob, qubit_s), o_ng = blob_put(blob, neigh), o_ix = blob_put(blob, isx),
                 o_ty = blob_put(blob, typed), o_c0 = blob_put(blob, col0), o_r0 = blob_put(blob, row0), o_cs = blob_put(blob, cstatic),
                 o_cb = blob_put(blob, cstab), o_cq = blob_put(blob, cqubit);
    hipError_t e = hipMalloc(&E->d_blob, blob.size());
    if (e == hipSuccess) e = hipMemcpy(E->d_blob, blob.data(), blob.size(), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&E->d_state, (size_t)cfg->n_envs * E->sw * sizeof(u64));
    if (e == hipSuccess) e = hipMemset(E->d_state, 0, (size_t)cfg->n_envs * E->sw * sizeof(u64));
    if (e != hipSuccess) {
        dq_set_error("dq_envb_create: %s", hipGetErrorString(e));
        dq_envb_destroy(E);
        return DQ_ERR_HIP;
    }
    E->tab.stab_q = reinterpret_cast<const u64*>(E->d_blob + o_sq); E->tab.qubit_s = reinterpret_cast<const u64*>(E->d_blob + o_qs);
    E->tab.neigh = reinterpret_cast<const u64*>(E->d_blob + o_ng); E->tab.stab_isx = E->d_blob + o_ix;
    E->tab.typed = reinterpret_cast<const unsigned short*>(E->d_blob + o_ty);
    E->tab.col0 = reinterpret_cast<const u64*>(E->d_blob + o_c0); E->tab.row0 = reinterpret_cast<const u64*>(E->d_blob + o_r0);
    E->tab.cell_static = E->d_blob + o_cs; E->tab.cell_stab = reinterpret_cast<const unsigned short*>(E->d_blob + o_cb);
    E->tab.cell_qubit = reinterpret_cast<const unsigned short*>(E->d_blob + o_cq);
    const dq_status ms = dq_match_create(d, &E->match);
    if (ms != DQ_OK) { dq_envb_destroy(E); return ms; }
    static unsigned long long attr_devs = 0;                          // per device (common.h dq_device_bit)
    const unsigned long long dev_bit = dq_device_bit();
    if (!(attr_devs & dev_bit)) {
        hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(env_big_kernel<1>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(env_big_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(env_big_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ae == hipSuccess) ae = hipFuncSetAttribute(reinterpret_cast<const void*>(env_big_kernel<4>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
        if (ae != hipSuccess) {
            dq_set_error("dq_envb_create: hipFuncSetAttribute: %s", hipGetErrorString(ae));
            dq_envb_destroy(E);                                         // (frees the tables and the matching referee)
            return DQ_ERR_HIP;
        }
        attr_devs |= dev_bit;
    }
    *out = E;
    return DQ_OK;
}

dq_status dq_envb_get_info(const dq_envb* E, dq_env_info* out, int* legal_words) {
    DQ_REQUIRE(E && out, DQ_ERR_INVALID, "dq_envb_get_info: null argument");
    *out = E->info;
    if (legal_words) *legal_words = E->LW;
    return DQ_OK;
}

dq_status dq_envb_set_rates(dq_envb* E, double p_phys, double p_meas) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_envb_set_rates: null argument");
    DQ_REQUIRE(p_phys >= 0.0 && p_phys <= 1.0 && p_meas >= 0.0 && p_meas <= 1.0, DQ_ERR_INVALID, "error rates must be in [0, 1]");
    E->T_phys = dq_rate_threshold(p_phys);
    E->T_meas = dq_rate_threshold(p_meas);
    E->rates_set = true;
    return DQ_OK;
}

}  // extern "C"

static dq_status big_fill(dq_envb* E, BigParams& p) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "null environment");
    DQ_REQUIRE(E->rates_set, DQ_ERR_STATE, "dq_envb_set_rates has not been called");
    memset(&p, 0, sizeof(p));
    p.tab = E->tab;
    match_comp(E->match, 0, &p.mx);
    match_comp(E->match, 1, &p.mz);
    p.state = E->d_state;
    p.n_envs = E->cfg.n_envs; p.d2 = E->cfg.d * E->cfg.d; p.n_stab = E->info.n_stab; p.depth = E->cfg.volume_depth;
    p.layers = E->info.n_action_layers; p.n_actions = E->info.num_actions; p.identity = E->info.identity_index;
    p.model = E->cfg.error_model; p.use_Y = E->cfg.use_Y; p.sw = E->sw; p.P = E->P; p.W = E->W; p.LW = E->LW;
    p.obs_size = E->info.obs_c * E->P;
    p.env_id_base = E->cfg.env_id_base; p.seed0 = E->cfg.seed[0]; p.seed1 = E->cfg.seed[1];
    p.T_phys = E->T_phys; p.T_meas = E->T_meas;
    return DQ_OK;
}

static dq_status big_launch(dq_envb* E, const BigParams& p, hipStream_t st) {
    const int blocks = (p.n_envs + BIG_EPB - 1) / BIG_EPB;
    const size_t lds = BIG_EPB * big_wave_lds(p.sw);
    DQ_REQUIRE(lds <= 160 * 1024, DQ_ERR_UNSUPPORTED, "lattice record too large for LDS");
    { const dq_status rc = match_reset_locks(E->match, st); if (rc != DQ_OK) return rc; }
    dq_prof_begin(DQ_K_ENV, st);
    switch (E->W) {
        case 1: env_big_kernel<1><<<blocks, 64 * BIG_EPB, lds, st>>>(p); break;
        case 2: env_big_kernel<2><<<blocks, 64 * BIG_EPB, lds, st>>>(p); break;
        case 3: env_big_kernel<3><<<blocks, 64 * BIG_EPB, lds, st>>>(p); break;
        default: env_big_kernel<4><<<blocks, 64 * BIG_EPB, lds, st>>>(p); break;
    }
    dq_prof_end(DQ_K_ENV, st);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

extern "C" {

dq_status dq_envb_reset(dq_envb* E, const uint8_t* which_dev, uint8_t* obs_dev, uint64_t* legal_dev, uint32_t* lifetime_dev, void* stream) {
    BigParams p;
    const dq_status s = big_fill(E, p);
    if (s != DQ_OK) return s;
    p.mode = 0; p.which = which_dev; p.obs = obs_dev; p.legal = legal_dev; p.lifetime = lifetime_dev;
    return big_launch(E, p, (hipStream_t)stream);
}

dq_status dq_envb_step(dq_envb* E, const int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev,
                       uint64_t* legal_dev, uint32_t* lifetime_dev, uint8_t* was_reset_dev, uint8_t* inexact_dev, void* stream) {
    BigParams p;
    const dq_status s = big_fill(E, p);
    if (s != DQ_OK) return s;
    DQ_REQUIRE(action_dev, DQ_ERR_INVALID, "dq_envb_step: null action");
    p.mode = 1; p.auto_reset = auto_reset; p.action = action_dev; p.obs = obs_dev; p.reward = reward_dev; p.done = done_dev;
    p.legal = legal_dev; p.lifetime = lifetime_dev; p.was_reset = was_reset_dev; p.inexact = inexact_dev;
    return big_launch(E, p, (hipStream_t)stream);
}

dq_status dq_envb_act_step(dq_envb* E, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                           int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev,
                           uint32_t* lifetime_dev, uint8_t* was_reset_dev, uint8_t* inexact_dev, void* stream) {
    BigParams p;
    const dq_status s = big_fill(E, p);
    if (s != DQ_OK) return s;
    DQ_REQUIRE(action_dev && seed, DQ_ERR_INVALID, "dq_envb_act_step: null argument");
    DQ_REQUIRE(eps >= 0.0 && eps <= 1.0, DQ_ERR_INVALID, "dq_envb_act_step: eps must be in [0,1]");
    p.mode = 1; p.auto_reset = auto_reset; p.obs = obs_dev; p.reward = reward_dev; p.done = done_dev;
    p.legal = legal_dev; p.lifetime = lifetime_dev; p.was_reset = was_reset_dev; p.inexact = inexact_dev;
    p.policy = 1; p.q = q_dev; p.T_eps = dq_rate_threshold(eps); p.pt = t; p.masked_greedy = masked_greedy;
    p.pseed0 = seed[0]; p.pseed1 = seed[1]; p.action_out = action_dev;
    return big_launch(E, p, (hipStream_t)stream);
}

dq_status dq_envb_export_state(dq_envb* E, uint64_t* state_dev, void* stream) {
    BigParams p;
    const dq_status s = big_fill(E, p);
    if (s != DQ_OK) return s;
    DQ_REQUIRE(state_dev, DQ_ERR_INVALID, "dq_envb_export_state: null argument");
    const int blocks = (p.n_envs + 3) / 4;
    switch (E->W) {
        case 1: env_big_export_kernel<1><<<blocks, 256, 0, (hipStream_t)stream>>>(p, state_dev, E->ew); break;
        case 2: env_big_export_kernel<2><<<blocks, 256, 0, (hipStream_t)stream>>>(p, state_dev, E->ew); break;
        case 3: env_big_export_kernel<3><<<blocks, 256, 0, (hipStream_t)stream>>>(p, state_dev, E->ew); break;
        default: env_big_export_kernel<4><<<blocks, 256, 0, (hipStream_t)stream>>>(p, state_dev, E->ew); break;
    }
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_policy_select_wide(const float* q_dev, const uint64_t* legal_dev, int n, int n_actions, int legal_words, double eps,
                                int masked_greedy, const uint32_t seed[2], uint32_t env_id_base, uint64_t t, int32_t* action_dev,
                                void* stream) {
    DQ_REQUIRE(legal_dev && action_dev && seed, DQ_ERR_INVALID, "dq_policy_select_wide: null argument");
    DQ_REQUIRE(n >= 1 && n_actions >= 1 && n_actions <= 64 * legal_words, DQ_ERR_INVALID, "dq_policy_select_wide: bad sizes");
    DQ_REQUIRE(eps >= 0.0 && eps <= 1.0, DQ_ERR_INVALID, "dq_policy_select_wide: eps must be in [0,1]");
    policy_wide_kernel<<<(n + 3) / 4, 256, 0, (hipStream_t)stream>>>(q_dev, legal_dev, n, n_actions, legal_words, dq_rate_threshold(eps),
                                                                    masked_greedy, seed[0], seed[1], env_id_base, t, action_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

}  // extern "C"
