// Device side of the batched surface-code environment, shared by env.hip (env_kernel) and fused_bwd.hip (the step of the lattices riding
// on the dense backward's launch): tables, launch parameters and the body of one block of lattices.  Formulation: see env.hip.
#pragma once
#include "common.h"

#define DQ_MAX_DEPTH 16
typedef u32 env_u32x4 __attribute__((ext_vector_type(4)));
#define ENVS_PER_BLOCK 4
#define STATE_FIXED 9      // internal record: xmask zmask acted round comp0 comp1 legal0 legal1 meta, then volume
#define EXPORT_FIXED 11

struct EnvTables {
    u64 stab_qmask[64];    // qubits of stabilizer s (measurement order), 0 beyond n_stab
    u64 qubit_smask[64];   // live stabilizers touched by qubit q (ENV:262-271)
    u64 neigh_qmask[64];   // 8-neighbourhood of qubit q (ENV:349-372)
    u8 stab_isx[64];       // 1: type-3 plaquette (parity of the X component), 0: type 1 (Z component)
    u8 stab_type[64];
    u8 ref_src[64];        // lane h <- stabilizer whose bit lands at referee position h (255: none)
    u8 cell_static[256];   // padding_syndrome decoration (ENV:284-298)
    u8 cell_stab[256];     // stabilizer shown at an even-even cell (ENV:292-294), 255: none
    u8 cell_qubit[256];    // qubit shown at an odd-odd cell of an action plane (ENV:301-314), 255: none
    u32 cell_pack[256];    // the three cell tables in one word: static | stab << 8 | qubit << 16 (a lane keeps its cells' words in registers)
    u32 pix_stab[64];      // compact observation: the stabilizers at the four corners (dy, dx) of conv1's 3 x 3 stride-2 patch of output pixel p = oy d + ox,
                           // byte 2 dy + dx = stabilizer shown at padded cell (2 (oy + dy), 2 (ox + dx)), 255: none (a dead corner of the (d+1)^2 grid)
    u64 col0, row0;        // FL:312-317
};

struct EnvParams {
    const EnvTables* tab;
    u64* state;
    const u32 *lut_x, *lut_z;
    const u32* lut_joint;          // != NULL: ONE table over the whole syndrome word, 2 bits (class X + 2Z) per entry: any `.predict` referee
    int n_envs, d2, n_stab, depth, layers, n_actions, identity, model, use_Y, sw, P, C, obs_size;
    u32 env_id_base, seed0, seed1;
    u64 T_phys, T_meas;
    int mode, auto_reset;          // mode 0: reset, 1: step
    const u8* which;
    const int32_t* action;
    u8* obs;
    float* reward;
    u8* done;
    u64* legal;
    u32* lifetime;
    u8* was_reset;
    // fused action selection (dq_env_act_step): the rule of policy.hip's policy_kernel, one lattice per wave
    int policy;                    // 0: actions come from `action`; 1: selected here and written to action_out
    const float* q;                // [n_envs, n_actions] or NULL (explore always)
    u64 T_eps, pt;
    int masked_greedy;
    u32 pseed0, pseed1;
    int32_t* action_out;
    // replay sampling for the update that follows this step (dq_env_act_step_sample): workgroups [env_blocks, env_blocks + s_blocks)
    int env_blocks, s_blocks;      // s_blocks == 0 with s_batch > 0: the lattices' own blocks draw the samples behind their step (s_inline)
    const u8* s_terminal;          // the terminal ring; this step writes slot head - 1, the rule only reads older slots
    int s_n_slots, s_head, s_filled, s_batch;
    u32 s_seed0, s_seed1, s_base;
    u64 s_t;
    int32_t* s_index;
    // episode bookkeeping of THIS step (dq_episode_stats' four sums), done by the blocks themselves when the step rides on a launch
    // that runs before anything could read its results (fused_bwd.hip); NULL: not here
    unsigned long long* stats;
    const u8* dec_in;              // != NULL: the referee's class for each lattice's post-action syndrome, computed by a pre-pass
                                   // (env.hip referee_mlp_kernel: a Dense-stack `static_decoder` evaluated on the device, ENV:144)
    int pair;                      // 1: two lattices per wave (env_block2; d <= 5: qubits, stabilizers and record words all fit 32 lanes)
    int lut_words;                 // > 0: lut_x / lut_z (this many words each; d <= 5: 128) are copied into LDS at the top of a block, so that the
                                   // referee look-up behind the syndrome is an LDS read instead of a dependent round trip to L2
    // compact observation (dq_env_patch_output): one u32 per lattice and conv1 output pixel p = oy d + ox -- the DATA bits of the pixel's 3 x 3
    // stride-2 patch of the padded planes: bit 4 j + 2 dy + dx = faulty syndrome plane j at grid cell (oy + dy, ox + dx) (the patch's corners,
    // ENV:292-294), bit 4 depth + l = action plane l at qubit p (its centre, ENV:309-312); every other cell of the patch is a constant of
    // padding_syndrome / padding_actions (ENV:273-314).  Row i = lattice i, patch_stride words apart.  NULL: not written.
    u32* patch;
    int patch_stride;
    // several agent steps in one launch (env_block2<EPB, true>, dq_env_act_steps): action / reward / done / obs / patch are RING bases ([ring_slots][n_envs] ...),
    // step s writes slot (ring_slot0 + s) mod ring_slots (the successor observation: the slot behind it); policy counter pt + s
    int steps, ring_slots, ring_slot0;
};

// The lattice's patch words (EnvParams.patch): lane p < d2 composes pixel p's word from the volume words `vol` (LDS, lane-shared) and the
// completed-actions mask.  ps = EnvTables.pix_stab[p].
static __device__ __forceinline__ u32 env_patch_word(const EnvParams& p, const volatile u64* vol, u64 comp0, u64 comp1, u32 ps, int pix) {
    u32 w = 0;
    for (int j = 0; j < p.depth; ++j) {
        const u64 v = vol[j];
        const u32 vlo = (u32)v, vhi = (u32)(v >> 32);
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const u32 s = (ps >> (8 * c)) & 0xffu;
            const u32 src = (s & 32u) ? vhi : vlo;
            w |= (s < 64u ? (src >> (s & 31u)) & 1u : 0u) << (4 * j + c);
        }
    }
    for (int l = 0; l < p.layers; ++l) {
        const int a = l * p.d2 + pix;
        w |= (u32)(((a < 64 ? comp0 : comp1) >> (a & 63)) & 1) << (4 * p.depth + l);
    }
    return w;
}

// The same for lattices whose stabilizers fit 32 bits (the two-per-wave form, d <= 5): the corners' shifts and validity are loop invariants of the caller (ps is a
// per-lane constant), a plane costs four bit-field extracts and one masked insert
static __device__ __forceinline__ u32 env_patch_word32(const EnvParams& p, const volatile u64* vol, u64 comp0, u64 comp1, u32 ps, int pix) {
    const u32 s0 = ps & 0xffu, s1 = (ps >> 8) & 0xffu, s2 = (ps >> 16) & 0xffu, s3 = ps >> 24;
    const u32 valid = (s0 < 32u ? 1u : 0u) | (s1 < 32u ? 2u : 0u) | (s2 < 32u ? 4u : 0u) | (s3 < 32u ? 8u : 0u);
    u32 w = 0;
    for (int j = 0; j < p.depth; ++j) {
        const u32 v = (u32)vol[j];
        const u32 nib = ((v >> (s0 & 31u)) & 1u) | (((v >> (s1 & 31u)) & 1u) << 1) | (((v >> (s2 & 31u)) & 1u) << 2) | (((v >> (s3 & 31u)) & 1u) << 3);
        w |= (nib & valid) << (4 * j);
    }
    for (int l = 0; l < p.layers; ++l) {
        const int a = l * p.d2 + pix;
        w |= (u32)(((a < 64 ? comp0 : comp1) >> (a & 63)) & 1) << (4 * p.depth + l);
    }
    return w;
}

// lanes of one wave hand words to each other through LDS (the per-wave referee tables, the volume words): wavefront-scope release / acquire
// + wave barrier, so that the ordering does not rest on the compiler's aliasing analysis (as match_dev.h match_wave_sync)
static __device__ __forceinline__ void env_wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

// (lo, hi) |= v << s over 128 bits, 0 <= s < 128.  Branch-free on VALUES: written as three branches that update one word or the other, hipcc kept
// the pair in SCRATCH memory and selected the word by address -- a load / wait / store round trip per call in the lattices' reset path.
static __device__ __forceinline__ void or_shl128(u64& lo, u64& hi, u64 v, int s) {
    const u64 l = s < 64 ? v << (s & 63) : 0ull;
    const u64 h = s == 0 ? 0ull : (s < 64 ? v >> ((64 - s) & 63) : v << ((s - 64) & 63));
    lo |= l;
    hi |= h;
}

// One block of EPB lattices (64 * EPB threads, one lattice per wave); `block` = the block's number inside the environment part of the
// grid (env_kernel: blockIdx.x; as a rider on another kernel's launch -- fused_bwd.hip -- the offset is subtracted by the caller).
// LDS: env_block_lds(EPB, waves, obs_size, lut_words) bytes at `smem`: [volumes | referee tables, one copy per wave | observation stage].
// (the referee tables: one private copy per WAVE -- `waves` of them --, so that no barrier stands between the copy and the look-up; the stage starts
// on a 16-byte boundary: it leaves by 16-byte stores where the block's rows in global memory are aligned that way)
static inline size_t env_lut_lds(int lut_words) { return ((size_t)2 * lut_words * 4 + 15) & ~(size_t)15; }
static inline size_t env_block_lds(int epb, int waves, int obs_size, int lut_words) {
    return (size_t)epb * DQ_MAX_DEPTH * 8 + (size_t)waves * env_lut_lds(lut_words) + (((size_t)epb * obs_size + 15) & ~(size_t)15);
}
#define ENV_LUT_LDS_MAX 256             // words per table staged in LDS (1 KB: up to 13 stabilizers per type)
#define ENV_CELLS 4                     // cells of an observation plane per lane: P <= 64 * 4 (one lattice per wave), P <= 32 * 4 (two per wave)
#define ENV_QPRE 3                      // Q values per lane requested with the record: n_actions <= 64 * 3 / 32 * 3

// Replay sampling by the lattices' own blocks, behind their step (s_blocks == 0): where the step rides on the dense backward, separate
// sampling workgroups found no free wave slots (one dense + one environment workgroup fill a CU) and started only when a dense
// workgroup ended -- the launch's tail.  The first blocks were dispatched first and finish first: the extra microsecond is theirs.
template <int THREADS>
static __device__ __forceinline__ void env_inline_sampling(const EnvParams& p, const int block) {
    if (p.s_blocks == 0 && p.s_batch > 0) {
        const int b = block * THREADS + (int)threadIdx.x;
        if (b < p.s_batch)
            p.s_index[b] = dq_replay_row(p.s_terminal, p.n_envs, p.s_n_slots, p.s_head, p.s_filled, p.s_batch, p.s_seed0, p.s_seed1, p.s_t, p.s_base + (u32)b);
    }
}

template <int EPB>
static __device__ __forceinline__ void env_block(const EnvParams& p, const int block, u8* __restrict__ smem) {
    constexpr int THREADS = 64 * EPB;
    u64* s_vol = reinterpret_cast<u64*>(smem);                              // [EPB][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lut_bytes = (2 * p.lut_words * 4 + 15) & ~15;
    u32* s_lut = reinterpret_cast<u32*>(smem + EPB * DQ_MAX_DEPTH * 8 + wave * lut_bytes);     // this WAVE's copy of the component referee tables [2][lut_words]
    u8* s_stage = smem + EPB * DQ_MAX_DEPTH * 8 + EPB * lut_bytes;          // [EPB * obs_size]
    __shared__ unsigned long long s_est[EPB][4];                            // episode bookkeeping of the block's lattices (p.stats)

    if (block >= p.env_blocks) {                                            // replay sampling rides along (independent of this step's results)
        const int b = (block - p.env_blocks) * THREADS + tid;
        if (b < p.s_batch)
            p.s_index[b] = dq_replay_row(p.s_terminal, p.n_envs, p.s_n_slots, p.s_head, p.s_filled, p.s_batch, p.s_seed0, p.s_seed1, p.s_t, p.s_base + (u32)b);
        return;
    }
    const int i = block * EPB + wave;
    const bool active = i < p.n_envs;                                        // wave-uniform

    // ---- everything the step reads from memory is requested HERE, in one batch: the lattice's record, the tables' per-lane entries (the cells of
    //      the observation planes this lane composes: registers instead of three LDS tables read byte by byte per plane), the Q row of the greedy
    //      choice (it used to go out behind the policy's Philox call and its explore / exploit branch: a round trip of its own for nine lattices in
    //      ten), the referee tables (copied into LDS where they are small: the look-up behind the syndrome was a dependent round trip to L2).  The
    //      step then is arithmetic and LDS traffic between one load latency at its top and its stores at the end.  (round 3, phase stamps of the
    //      step riding on the dense backward: tables + record 4K, policy 4.5K, referee 2.8K, observation planes 6.3K of 32K cycles)
    const EnvTables* __restrict__ T = p.tab;
    u32 cpk[ENV_CELLS];
#pragma unroll
    for (int k = 0; k < ENV_CELLS; ++k) cpk[k] = T->cell_pack[lane + 64 * k];
    const u64 sq = T->stab_qmask[lane];
    const u64 qs = T->qubit_smask[lane];
    const u64 nq = T->neigh_qmask[lane];
    const bool isx = T->stab_isx[lane] != 0;
    const int rsrc = T->ref_src[lane];
    const u32 pstab = T->pix_stab[lane];                 // (unconditional: a guarded load is a branch with a wait of its own)
    const u64 col0 = T->col0, row0 = T->row0;
    volatile u64* vol = s_vol + wave * DQ_MAX_DEPTH;   // written by lane 0, read by other lanes of the same wave
    u64* rec = p.state + (size_t)(active ? i : 0) * p.sw;
    const u64 word = lane < p.sw ? rec[lane] : 0;
    float qpre[ENV_QPRE];
    const bool have_q = p.policy && p.q != nullptr;
#pragma unroll
    for (int m = 0; m < ENV_QPRE; ++m) qpre[m] = have_q ? p.q[(size_t)(active ? i : 0) * p.n_actions + min(lane + 64 * m, p.n_actions - 1)] : 0.f;
    // (a copy per wave: written and read by the same wave -- no workgroup barrier, a wavefront fence)
    for (int k = lane; k < p.lut_words; k += 64) { s_lut[k] = p.lut_x[k]; s_lut[p.lut_words + k] = p.lut_z[k]; }
    if (p.lut_words) env_wave_sync();

    u64 comp0 = 0, comp1 = 0;
    if (p.stats && lane < 4) s_est[wave][lane] = 0;
    if (active) {
        u64 xmask = wave_bcast64(word, 0), zmask = wave_bcast64(word, 1), acted = wave_bcast64(word, 2);
        u64 round = wave_bcast64(word, 3);
        comp0 = wave_bcast64(word, 4);
        comp1 = wave_bcast64(word, 5);
        u64 legal0 = wave_bcast64(word, 6), legal1 = wave_bcast64(word, 7);
        const u64 meta = wave_bcast64(word, 8);
        u32 lifetime = (u32)meta;
        int done = (int)((meta >> 32) & 1);
        if (lane >= STATE_FIXED && lane < STATE_FIXED + p.depth) vol[lane - STATE_FIXED] = word;

        bool do_reset;
        if (p.mode == 0) {
            do_reset = p.which ? (__builtin_amdgcn_readfirstlane((int)p.which[i]) != 0) : true;
        } else {
            do_reset = p.auto_reset && done;
        }
        const bool do_step = p.mode == 1 && !do_reset;
        float reward = 0.f;
        bool need_volume = do_reset;
        if (do_reset) {                                                     // ENV:106-107, 211-213
            done = 0; lifetime = 0; xmask = 0; zmask = 0;
        }
        int a_sel = 0;
        if (p.policy) {                                                     // EpsGreedyQPolicy / GreedyQPolicy(masked_greedy), see policy.hip
            u32 w[4];
            philox4x32_10((u32)p.pt, (u32)(p.pt >> 32), p.env_id_base + (u32)i, (u32)DQ_STREAM_POLICY << 16, p.pseed0, p.pseed1, w);
            if (p.q == nullptr || (u64)w[1] < p.T_eps) {                    // explore: k-th smallest legal action
                const int n_legal = __popcll(legal0) + __popcll(legal1);
                a_sel = kth_set_bit128(legal0, legal1, (int)__umulhi(w[0], (u32)n_legal));
            } else {                                                        // first maximum of the Q row (optionally over the legal set)
                const float* row = p.q + (size_t)i * p.n_actions;
                float best = -INFINITY;
                int best_a = 0x7fffffff;
#pragma unroll
                for (int m = 0; m < ENV_QPRE; ++m) {                        // (the row's first 64 * ENV_QPRE entries were requested at the top)
                    const int k = lane + 64 * m;
                    const bool ok = k < p.n_actions && (!p.masked_greedy || (((k < 64 ? legal0 : legal1) >> (k & 63)) & 1));
                    if (ok && (qpre[m] > best || best_a == 0x7fffffff)) { best = qpre[m]; best_a = k; }
                }
                for (int k = lane + 64 * ENV_QPRE; k < p.n_actions; k += 64) {
                    const bool ok = !p.masked_greedy || (((k < 64 ? legal0 : legal1) >> (k & 63)) & 1);
                    const float v = row[k];
                    if (ok && (v > best || best_a == 0x7fffffff)) { best = v; best_a = k; }
                }
                dq_wave_argmax(best, best_a);                               // DPP + v_readlane (common.h), the butterfly's result
                a_sel = best_a;
            }
            a_sel = __builtin_amdgcn_readfirstlane(a_sel);
            if (lane == 0) p.action_out[i] = a_sel;
        }
        if (do_step) {
            int a = p.policy ? a_sel : __builtin_amdgcn_readfirstlane(p.action[i]);
            if ((unsigned)a >= (unsigned)p.n_actions) a = p.identity;
            const u64 cw = a < 64 ? comp0 : comp1;
            const bool done_identity = a == p.identity || ((cw >> (a & 63)) & 1);      // ENV:131
            if (a != p.identity) {                                          // ENV:135-136, FL:243-294
                const int layer = a / p.d2, q = a - layer * p.d2;
                const int pauli = p.model == DQ_MODEL_X ? 1 : (p.use_Y ? layer + 1 : (layer == 0 ? 1 : 3));
                if (pauli != 3) xmask ^= 1ull << q;
                if (pauli != 1) zmask ^= 1ull << q;
            }
            const u64 true_word = __ballot(__popcll((isx ? xmask : zmask) & sq) & 1);   // ENV:139, FL:152-174
            const int cls = (__popcll(xmask & col0) & 1) + 2 * (__popcll(zmask & row0) & 1);  // ENV:143
            const u64 refw = __ballot(rsrc < 64 && ((true_word >> (rsrc & 63)) & 1));
            const u32 ix = (u32)refw, iz = (u32)(refw >> 32);
            int dec;
            if (p.dec_in) {                                                 // a Dense-stack static_decoder, evaluated by the pre-pass (env.hip)
                dec = p.dec_in[i];
            } else if (p.lut_joint) {                                       // an arbitrary static_decoder.predict, tabulated (ENV:144,150)
                const u32 sw = (u32)true_word;                              // bit s = stabilizer s in measurement order (n_stab <= 24)
                dec = (p.lut_joint[sw >> 4] >> (2 * (sw & 15))) & 3;
            } else if (p.lut_words) {                                       // the component tables' copies in LDS
                dec = (s_lut[ix >> 5] >> (ix & 31)) & 1;                    // ENV:144
                if (p.model != DQ_MODEL_X) dec += 2 * ((s_lut[p.lut_words + (iz >> 5)] >> (iz & 31)) & 1);
            } else {
                dec = (p.lut_x[ix >> 5] >> (ix & 31)) & 1;                  // ENV:144
                if (p.model != DQ_MODEL_X) dec += 2 * ((p.lut_z[iz >> 5] >> (iz & 31)) & 1);
            }
            dec = __builtin_amdgcn_readfirstlane(dec);
            if (cls == 0 && true_word == 0) reward = 1.f;                   // ENV:148-149
            else if (dec != cls) done = 1;                                  // ENV:150-151
            if (done_identity) {
                need_volume = true;                                         // ENV:155
            } else {                                                        // ENV:185-196
                if (a < 64) comp0 |= 1ull << a; else comp1 |= 1ull << (a - 64);
                const int q = a % p.d2;
                if (!((acted >> q) & 1)) {
                    acted |= 1ull << q;
                    const u64 nm = wave_bcast64(nq, q);                     // (neigh_qmask[q]: lane q holds it)
                    for (int j = 0; j < p.layers; ++j) or_shl128(legal0, legal1, nm, j * p.d2);
                }
            }
        }
        if (need_volume) {                                                  // ENV:157-172 == ENV:216-231
            u64 summed;
            do {
                summed = 0;
                for (int j = 0; j < p.depth; ++j) {
                    u32 w[4];
                    philox4x32_10((u32)round, (u32)(round >> 32), p.env_id_base + (u32)i, (u32)lane, p.seed0, p.seed1, w);
                    const bool hit = lane < p.d2 && (u64)w[0] < p.T_phys;   // FL:99 / FL:119
                    const int typ = p.model == DQ_MODEL_X ? 1 : 1 + (int)__umulhi(w[1], 3u);   // FL:100
                    // IIDXZ (FL:134-160): the qubit's second uniform decides an independent Z flip instead of the Pauli type
                    const bool zhit = lane < p.d2 && (u64)w[1] < p.T_phys;
                    const u64 ex = __ballot(p.model == DQ_MODEL_IIDXZ ? hit : hit && typ != 3);
                    const u64 ez = __ballot(p.model == DQ_MODEL_IIDXZ ? zhit : hit && typ != 1);
                    const u64 flips = __ballot(lane < p.n_stab && (u64)w[2] < p.T_meas);      // FL:191-221
                    ++round;
                    xmask ^= ex;                                            // ENV:164, FL:226-241
                    zmask ^= ez;
                    const u64 tw = __ballot(__popcll((isx ? xmask : zmask) & sq) & 1);        // ENV:165
                    const u64 v = tw ^ flips;                               // ENV:166
                    if (lane == 0) vol[j] = v;
                    summed |= v;                                            // ENV:168
                    ++lifetime;                                             // ENV:169
                }
            } while (summed == 0);                                          // ENV:171
            // reset_legal_moves, ENV:238-258
            comp0 = comp1 = 0; acted = 0;
            const u64 legal_q = __ballot(lane < p.d2 && (qs & summed) != 0);
            legal0 = legal1 = 0;
            or_shl128(legal0, legal1, 1ull, p.identity);
            for (int j = 0; j < p.layers; ++j) or_shl128(legal0, legal1, legal_q, j * p.d2);
        }

        // ---- state record and scalar outputs ----------------------------------------------------
        const u64 meta_out = (u64)lifetime | ((u64)done << 32);
        u64 o = 0;
        o = lane == 0 ? xmask : o;  o = lane == 1 ? zmask : o;  o = lane == 2 ? acted : o;
        o = lane == 3 ? round : o;  o = lane == 4 ? comp0 : o;  o = lane == 5 ? comp1 : o;
        o = lane == 6 ? legal0 : o; o = lane == 7 ? legal1 : o; o = lane == 8 ? meta_out : o;
        if (lane >= STATE_FIXED && lane < STATE_FIXED + p.depth) o = vol[lane - STATE_FIXED];
        if (lane < p.sw) rec[lane] = o;
        if (lane == 0) {
            if (p.reward) p.reward[i] = reward;
            if (p.done) p.done[i] = (u8)done;
            if (p.lifetime) p.lifetime[i] = lifetime;
            if (p.was_reset) p.was_reset[i] = (u8)(p.mode == 1 && do_reset);
            if (p.legal) { p.legal[2 * (size_t)i] = legal0; p.legal[2 * (size_t)i + 1] = legal1; }
            if (p.stats) {                                                  // dq_episode_stats' sums for this lattice (common.h dq_episode_stats_lane)
                const bool stepped = p.mode == 1 && !do_reset, ended = stepped && done;
                s_est[wave][0] = ended; s_est[wave][1] = ended ? lifetime : 0;
                s_est[wave][2] = stepped && reward > 0.5f; s_est[wave][3] = stepped;
            }
        }

        // ---- observation planes into the LDS stage (ENV:174-175, 200-201, 273-314): this lane's cells (lane, lane + 64, ...) of every plane, the
        //      cells' table words from registers -- one LDS read (the plane's syndrome word) and ENV_CELLS byte stores per plane -----------------
        if (p.obs) {
            u8* st = s_stage + wave * p.obs_size;
            for (int j = 0; j < p.depth; ++j) {
                const u64 v = vol[j];
#pragma unroll
                for (int k = 0; k < ENV_CELLS; ++k) {
                    const int c = lane + 64 * k;
                    const u32 sidx = (cpk[k] >> 8) & 0xffu;
                    if (c < p.P) st[j * p.P + c] = (u8)((cpk[k] & 0xffu) | (sidx < 64 ? (u32)((v >> (sidx & 63)) & 1) : 0u));
                }
            }
            for (int k2 = 0; k2 < p.layers; ++k2) {
#pragma unroll
                for (int k = 0; k < ENV_CELLS; ++k) {
                    const int c = lane + 64 * k;
                    const u32 qi = (cpk[k] >> 16) & 0xffu;
                    u32 bit = 0;
                    if (qi < 64) {
                        const int a = k2 * p.d2 + (int)qi;
                        bit = (u32)(((a < 64 ? comp0 : comp1) >> (a & 63)) & 1);
                    }
                    if (c < p.P) st[(p.depth + k2) * p.P + c] = (u8)bit;
                }
            }
        }
        // ---- compact observation: this lane's pixel word, straight to global memory (no stage, no barrier) ----------------------------------
        if (p.patch) {
            env_wave_sync();                                                // (lane 0's volume words)
            const u32 w = env_patch_word(p, vol, comp0, comp1, pstab, lane);
            if (lane < p.d2) p.patch[(size_t)i * p.patch_stride + lane] = w;
        }
    }

    if (!p.obs && !p.stats) { env_inline_sampling<THREADS>(p, block); return; }    // block-uniform
    __syncthreads();                                                        // the stage (and the bookkeeping words) visible
    if (p.stats && tid < 4) {                                               // integer sums: order-independent; at most four atomics per block
        unsigned long long v = 0;
#pragma unroll
        for (int w = 0; w < EPB; ++w) v += s_est[w][tid];
        if (v) atomicAdd(&p.stats[tid], v);
    }
    if (!p.obs) { env_inline_sampling<THREADS>(p, block); return; }
    {
        const int first = block * EPB;
        const int n_valid = min(EPB, p.n_envs - first);
        const int total = n_valid * p.obs_size;
        u8* g = p.obs + (size_t)first * p.obs_size;
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {                   // 16-byte stores (a store instruction moves 1 KB instead of 256 bytes: as dwords
            const env_u32x4* s128 = reinterpret_cast<const env_u32x4*>(s_stage);    // this copy was 5.6K of a riding block's 30K cycles)
            env_u32x4* g128 = reinterpret_cast<env_u32x4*>(g);
            const int nq = total >> 4;
            for (int k = tid; k < nq; k += THREADS) g128[k] = s128[k];
            for (int k = (nq << 4) + tid; k < total; k += THREADS) g[k] = s_stage[k];
        } else if ((reinterpret_cast<uintptr_t>(g) & 3) == 0) {
            const u32* s32 = reinterpret_cast<const u32*>(s_stage);
            u32* g32 = reinterpret_cast<u32*>(g);
            const int ndw = total >> 2;
            for (int k = tid; k < ndw; k += THREADS) g32[k] = s32[k];
            for (int k = (ndw << 2) + tid; k < total; k += THREADS) g[k] = s_stage[k];
        } else {
            for (int k = tid; k < total; k += THREADS) g[k] = s_stage[k];
        }
    }
    env_inline_sampling<THREADS>(p, block);
}

// ---- two lattices per wave (round 3) ------------------------------------------------------------------------------------------------
// At d <= 5 a lattice has at most 25 qubits, 24 stabilizers and 25 record words: it fits the 32 lanes of HALF a wavefront.  env_block2
// runs env_block's arithmetic with lattice 2w + h of the block on half h of wave w -- half as many waves for the same lattices, and no
// idle upper half in the Philox rounds.  That matters most where the step RIDES on the dense backward's launch (fused_bwd.hip): with
// 125 VGPRs a CU holds one dense workgroup plus ONE 8-wave environment workgroup, so 4096 one-per-wave lattices took two rounds of
// environment workgroups (27.7 us per launch); two per wave they take one (measured with half the lattices: 24.1 us).  Same words,
// same bits: lane h*32 + l stands where lane l stood (Philox counters use l), every 64-bit ballot is read as its own 32-bit half,
// wave-uniform values become half-uniform (shuffles of width 32 instead of v_readlane), and the two halves diverge freely (step /
// reset / rejection loop: ordinary EXEC masking).  The referee index needs the second 32 positions of ref_src as a second ballot.
// EPB = lattices per block (32 * EPB threads).  LDS: env_block_lds(EPB, EPB / 2, obs_size, lut_words).
static __device__ __forceinline__ u64 half_bcast64(u64 v, int src) {
    const u32 lo = (u32)__shfl((int)(u32)v, src, 32), hi = (u32)__shfl((int)(u32)(v >> 32), src, 32);
    return ((u64)hi << 32) | lo;
}

// MULTI (round 6, dq_env_act_steps): p.steps consecutive agent steps of the block's lattices in ONE launch -- selection (uniform over the legal moves: no Q row),
// step / auto-reset, transition into the replay ring --, the lattice's state carried in registers from step to step, the outputs of step s going to ring slot
// (ring_slot0 + s) mod ring_slots (action, reward, done) and the successor observation to the slot behind it: an acting loop of T steps costs one launch
// latency instead of T.  Same words, same bits as T launches (policy counter pt + s).
template <int EPB, bool MULTI = false>
static __device__ __forceinline__ void env_block2(const EnvParams& p, const int block, u8* __restrict__ smem) {
    constexpr int THREADS = 32 * EPB;
    u64* s_vol = reinterpret_cast<u64*>(smem);                              // [EPB][16]
    const int tid = threadIdx.x, lane = tid & 63, half = lane >> 5, hl = lane & 31, slot = tid >> 5;
    const int lut_bytes = (2 * p.lut_words * 4 + 15) & ~15;
    u32* s_lut = reinterpret_cast<u32*>(smem + EPB * DQ_MAX_DEPTH * 8 + (tid >> 6) * lut_bytes);      // this WAVE's copy [2][lut_words]
    u8* s_stage = smem + EPB * DQ_MAX_DEPTH * 8 + (EPB / 2) * lut_bytes;    // [EPB * obs_size]
    __shared__ unsigned long long s_est2[EPB][4];

    const int hshift = 32 * half;
    if (block >= p.env_blocks) {                                            // replay sampling rides along (independent of this step's results)
        const int b = (block - p.env_blocks) * THREADS + tid;
        if (b < p.s_batch)
            p.s_index[b] = dq_replay_row(p.s_terminal, p.n_envs, p.s_n_slots, p.s_head, p.s_filled, p.s_batch, p.s_seed0, p.s_seed1, p.s_t, p.s_base + (u32)b);
        return;
    }
    DQ_STAMP(DQ_TAG_ENV, 0);
    const int i = block * EPB + slot;
    const bool active = i < p.n_envs;                                        // half-uniform
    auto hb = [&](bool pred) -> u64 { return (u64)(u32)(__ballot(pred) >> hshift); };   // this lattice's 32 bits of a ballot

    // ---- one batch of loads at the top (env_block's comment): record, per-lane table entries, Q row, referee tables -> LDS ------------------
    const EnvTables* __restrict__ T = p.tab;
    u32 cpk[ENV_CELLS];
#pragma unroll
    for (int k = 0; k < ENV_CELLS; ++k) cpk[k] = T->cell_pack[hl + 32 * k];
    const u64 sq = T->stab_qmask[hl];
    const u64 qs = T->qubit_smask[hl];
    const u64 nq = T->neigh_qmask[hl];
    const bool isx = T->stab_isx[hl] != 0;
    const int rsrc_x = T->ref_src[hl], rsrc_z = T->ref_src[32 + hl];
    const u32 pstab = T->pix_stab[hl];
    const u64 col0 = T->col0, row0 = T->row0;
    volatile u64* vol = s_vol + slot * DQ_MAX_DEPTH;   // written by lane 0 of the half, read by its other lanes
    u64* rec = p.state + (size_t)(active ? i : 0) * p.sw;
    const u64 word = hl < p.sw ? rec[hl] : 0;
    float qpre[ENV_QPRE];
    const bool have_q = p.policy && p.q != nullptr;
#pragma unroll
    for (int m = 0; m < ENV_QPRE; ++m) qpre[m] = have_q ? p.q[(size_t)(active ? i : 0) * p.n_actions + min(hl + 32 * m, p.n_actions - 1)] : 0.f;
    for (int k = lane; k < p.lut_words; k += 64) { s_lut[k] = p.lut_x[k]; s_lut[p.lut_words + k] = p.lut_z[k]; }      // (per wave: a wavefront fence, no barrier)
    if (p.lut_words) env_wave_sync();

    u64 comp0 = 0, comp1 = 0;
    if (p.stats && hl < 4) s_est2[slot][hl] = 0;
    // the lattice's state: loaded once, carried over the steps of a MULTI launch
    u64 xmask = 0, zmask = 0, acted = 0, round = 0, legal0 = 0, legal1 = 0;
    u32 lifetime = 0;
    int done = 0;
    if (active) {
        xmask = half_bcast64(word, 0); zmask = half_bcast64(word, 1); acted = half_bcast64(word, 2);
        round = half_bcast64(word, 3);
        comp0 = half_bcast64(word, 4);
        comp1 = half_bcast64(word, 5);
        legal0 = half_bcast64(word, 6); legal1 = half_bcast64(word, 7);
        const u64 meta = half_bcast64(word, 8);
        lifetime = (u32)meta;
        done = (int)((meta >> 32) & 1);
        if (hl >= STATE_FIXED && hl < STATE_FIXED + p.depth) vol[hl - STATE_FIXED] = word;
    }
    const int n_steps = MULTI ? p.steps : 1;
  for (int step_no = 0; step_no < n_steps; ++step_no) {
    // this step's outputs: the launch's own, or (MULTI) the ring slots of step step_no
    int32_t* const action_out = MULTI ? p.action_out + (size_t)((p.ring_slot0 + step_no) % p.ring_slots) * p.n_envs : p.action_out;
    float* const reward_out = MULTI && p.reward ? p.reward + (size_t)((p.ring_slot0 + step_no) % p.ring_slots) * p.n_envs : p.reward;
    u8* const done_out = MULTI && p.done ? p.done + (size_t)((p.ring_slot0 + step_no) % p.ring_slots) * p.n_envs : p.done;
    u8* const obs_out = MULTI && p.obs ? p.obs + (size_t)((p.ring_slot0 + step_no + 1) % p.ring_slots) * p.n_envs * p.obs_size : p.obs;
    u32* const patch_out = MULTI && p.patch ? p.patch + (size_t)((p.ring_slot0 + step_no + 1) % p.ring_slots) * p.n_envs * p.patch_stride : p.patch;
    const u64 pt = MULTI ? p.pt + (u64)step_no : p.pt;
    const bool last_step = step_no + 1 == n_steps;
    if (active) {
        DQ_STAMP(DQ_TAG_ENV, 1);

        bool do_reset;
        if (p.mode == 0) {
            do_reset = p.which ? (p.which[i] != 0) : true;
        } else {
            do_reset = p.auto_reset && done;
        }
        const bool do_step = p.mode == 1 && !do_reset;
        float reward = 0.f;
        bool need_volume = do_reset;
        if (do_reset) {                                                     // ENV:106-107, 211-213
            done = 0; lifetime = 0; xmask = 0; zmask = 0;
        }
        int a_sel = 0;
        if (p.policy) {                                                     // EpsGreedyQPolicy / GreedyQPolicy(masked_greedy), see policy.hip
            u32 w[4];
            philox4x32_10((u32)pt, (u32)(pt >> 32), p.env_id_base + (u32)i, (u32)DQ_STREAM_POLICY << 16, p.pseed0, p.pseed1, w);
            if (p.q == nullptr || (u64)w[1] < p.T_eps) {                    // explore: k-th smallest legal action
                const int n_legal = __popcll(legal0) + __popcll(legal1);
                a_sel = kth_set_bit128(legal0, legal1, (int)__umulhi(w[0], (u32)n_legal));
            } else {                                                        // first maximum of the Q row (optionally over the legal set)
                const float* row = p.q + (size_t)i * p.n_actions;
                float best = -INFINITY;
                int best_a = 0x7fffffff;
#pragma unroll
                for (int m = 0; m < ENV_QPRE; ++m) {                        // (the row's first 32 * ENV_QPRE entries were requested at the top)
                    const int k = hl + 32 * m;
                    const bool ok = k < p.n_actions && (!p.masked_greedy || (((k < 64 ? legal0 : legal1) >> (k & 63)) & 1));
                    if (ok && (qpre[m] > best || best_a == 0x7fffffff)) { best = qpre[m]; best_a = k; }
                }
                for (int k = hl + 32 * ENV_QPRE; k < p.n_actions; k += 32) {
                    const bool ok = !p.masked_greedy || (((k < 64 ? legal0 : legal1) >> (k & 63)) & 1);
                    const float v = row[k];
                    if (ok && (v > best || best_a == 0x7fffffff)) { best = v; best_a = k; }
                }
                dq_row_argmax(best, best_a);                                // the half's two rows of 16 lanes (DPP, common.h) ...
                dq_argmax_take(best, best_a, __shfl_xor(best, 16), __shfl_xor(best_a, 16));   // ... combined: every lane of the half ends with the result
                a_sel = best_a;
            }
            if (hl == 0) action_out[i] = a_sel;
        }
        DQ_STAMP(DQ_TAG_ENV, 2);
        if (do_step) {
            int a = p.policy ? a_sel : p.action[i];
            if ((unsigned)a >= (unsigned)p.n_actions) a = p.identity;
            const u64 cw = a < 64 ? comp0 : comp1;
            const bool done_identity = a == p.identity || ((cw >> (a & 63)) & 1);      // ENV:131
            if (a != p.identity) {                                          // ENV:135-136, FL:243-294
                const int layer = a / p.d2, q = a - layer * p.d2;
                const int pauli = p.model == DQ_MODEL_X ? 1 : (p.use_Y ? layer + 1 : (layer == 0 ? 1 : 3));
                if (pauli != 3) xmask ^= 1ull << q;
                if (pauli != 1) zmask ^= 1ull << q;
            }
            const u64 true_word = hb(__popcll((isx ? xmask : zmask) & sq) & 1);          // ENV:139, FL:152-174
            const int cls = (__popcll(xmask & col0) & 1) + 2 * (__popcll(zmask & row0) & 1);  // ENV:143
            const u32 ix = (u32)hb(rsrc_x < 64 && ((true_word >> (rsrc_x & 63)) & 1));
            const u32 iz = (u32)hb(rsrc_z < 64 && ((true_word >> (rsrc_z & 63)) & 1));
            int dec;
            if (p.dec_in) {
                dec = p.dec_in[i];
            } else if (p.lut_joint) {                                       // an arbitrary static_decoder.predict, tabulated (ENV:144,150)
                const u32 sw = (u32)true_word;                              // bit s = stabilizer s in measurement order (n_stab <= 24)
                dec = (p.lut_joint[sw >> 4] >> (2 * (sw & 15))) & 3;
            } else if (p.lut_words) {                                       // the component tables' copies in LDS
                dec = (s_lut[ix >> 5] >> (ix & 31)) & 1;                    // ENV:144
                if (p.model != DQ_MODEL_X) dec += 2 * ((s_lut[p.lut_words + (iz >> 5)] >> (iz & 31)) & 1);
            } else {
                dec = (p.lut_x[ix >> 5] >> (ix & 31)) & 1;                  // ENV:144
                if (p.model != DQ_MODEL_X) dec += 2 * ((p.lut_z[iz >> 5] >> (iz & 31)) & 1);
            }
            if (cls == 0 && true_word == 0) reward = 1.f;                   // ENV:148-149
            else if (dec != cls) done = 1;                                  // ENV:150-151
            if (done_identity) {
                need_volume = true;                                         // ENV:155
            } else {                                                        // ENV:185-196
                if (a < 64) comp0 |= 1ull << a; else comp1 |= 1ull << (a - 64);
                const int q = a % p.d2;
                if (!((acted >> q) & 1)) {
                    acted |= 1ull << q;
                    const u64 nm = half_bcast64(nq, q);                     // (neigh_qmask[q]: lane q of the half holds it)
                    for (int j = 0; j < p.layers; ++j) or_shl128(legal0, legal1, nm, j * p.d2);
                }
            }
        }
        DQ_STAMP(DQ_TAG_ENV, 3);
        if (need_volume) {                                                  // ENV:157-172 == ENV:216-231
            u64 summed;
            do {
                summed = 0;
                for (int j = 0; j < p.depth; ++j) {
                    u32 w[4];
                    philox4x32_10((u32)round, (u32)(round >> 32), p.env_id_base + (u32)i, (u32)hl, p.seed0, p.seed1, w);
                    const bool hit = hl < p.d2 && (u64)w[0] < p.T_phys;     // FL:99 / FL:119
                    const int typ = p.model == DQ_MODEL_X ? 1 : 1 + (int)__umulhi(w[1], 3u);   // FL:100
                    const bool zhit = hl < p.d2 && (u64)w[1] < p.T_phys;    // IIDXZ (FL:134-160)
                    const u64 ex = hb(p.model == DQ_MODEL_IIDXZ ? hit : hit && typ != 3);
                    const u64 ez = hb(p.model == DQ_MODEL_IIDXZ ? zhit : hit && typ != 1);
                    const u64 flips = hb(hl < p.n_stab && (u64)w[2] < p.T_meas);          // FL:191-221
                    ++round;
                    xmask ^= ex;                                            // ENV:164, FL:226-241
                    zmask ^= ez;
                    const u64 tw = hb(__popcll((isx ? xmask : zmask) & sq) & 1);           // ENV:165
                    const u64 v = tw ^ flips;                               // ENV:166
                    if (hl == 0) vol[j] = v;
                    summed |= v;                                            // ENV:168
                    ++lifetime;                                             // ENV:169
                }
            } while (summed == 0);                                          // ENV:171
            // reset_legal_moves, ENV:238-258
            comp0 = comp1 = 0; acted = 0;
            const u64 legal_q = hb(hl < p.d2 && (qs & summed) != 0);
            legal0 = legal1 = 0;
            or_shl128(legal0, legal1, 1ull, p.identity);
            for (int j = 0; j < p.layers; ++j) or_shl128(legal0, legal1, legal_q, j * p.d2);
        }
        DQ_STAMP(DQ_TAG_ENV, 4);

        // ---- state record and scalar outputs ----------------------------------------------------
        if (!MULTI || last_step) {                                          // (the record leaves once per launch)
            const u64 meta_out = (u64)lifetime | ((u64)done << 32);
            u64 o = 0;
            o = hl == 0 ? xmask : o;  o = hl == 1 ? zmask : o;  o = hl == 2 ? acted : o;
            o = hl == 3 ? round : o;  o = hl == 4 ? comp0 : o;  o = hl == 5 ? comp1 : o;
            o = hl == 6 ? legal0 : o; o = hl == 7 ? legal1 : o; o = hl == 8 ? meta_out : o;
            if (hl >= STATE_FIXED && hl < STATE_FIXED + p.depth) o = vol[hl - STATE_FIXED];
            if (hl < p.sw) rec[hl] = o;
        }
        if (hl == 0) {
            if (reward_out) reward_out[i] = reward;
            if (done_out) done_out[i] = (u8)done;
            if (p.lifetime && (!MULTI || last_step)) p.lifetime[i] = lifetime;
            if (p.was_reset && (!MULTI || last_step)) p.was_reset[i] = (u8)(p.mode == 1 && do_reset);
            if (p.legal && (!MULTI || last_step)) { p.legal[2 * (size_t)i] = legal0; p.legal[2 * (size_t)i + 1] = legal1; }
            if (p.stats) {                                                  // dq_episode_stats' sums for this lattice
                const bool stepped = p.mode == 1 && !do_reset, ended = stepped && done;
                s_est2[slot][0] = ended; s_est2[slot][1] = ended ? lifetime : 0;
                s_est2[slot][2] = stepped && reward > 0.5f; s_est2[slot][3] = stepped;
            }
        }
        DQ_STAMP(DQ_TAG_ENV, 5);

        // ---- observation planes into the LDS stage (ENV:174-175, 200-201, 273-314): this lane's cells (hl, hl + 32, ...) of every plane, their
        //      table words from registers (env_block's comment) -------------------------------------------------------------------------------
        if (obs_out) {
            u8* st = s_stage + slot * p.obs_size;
            for (int j = 0; j < p.depth; ++j) {
                const u64 v = vol[j];
#pragma unroll
                for (int k = 0; k < ENV_CELLS; ++k) {
                    const int c = hl + 32 * k;
                    const u32 sidx = (cpk[k] >> 8) & 0xffu;
                    if (c < p.P) st[j * p.P + c] = (u8)((cpk[k] & 0xffu) | (sidx < 64 ? (u32)((v >> (sidx & 63)) & 1) : 0u));
                }
            }
            for (int k2 = 0; k2 < p.layers; ++k2) {
#pragma unroll
                for (int k = 0; k < ENV_CELLS; ++k) {
                    const int c = hl + 32 * k;
                    const u32 qi = (cpk[k] >> 16) & 0xffu;
                    u32 bit = 0;
                    if (qi < 64) {
                        const int a = k2 * p.d2 + (int)qi;
                        bit = (u32)(((a < 64 ? comp0 : comp1) >> (a & 63)) & 1);
                    }
                    if (c < p.P) st[(p.depth + k2) * p.P + c] = (u8)bit;
                }
            }
        }
        if (patch_out) {                                                    // compact observation (env_block's comment)
            env_wave_sync();
            const u32 w = env_patch_word32(p, vol, comp0, comp1, pstab, hl);      // (n_stab <= 32 in this form)
            if (hl < p.d2) patch_out[(size_t)i * p.patch_stride + hl] = w;
            if (MULTI) env_wave_sync();                                     // (the next step's volume overwrites the words these lanes have just read)
        }
        DQ_STAMP(DQ_TAG_ENV, 6);
    }

    if constexpr (MULTI) {
        // uint8 planes: the block's stage leaves for this step's ring slot; a barrier each side (the next step refills the stage).  No bookkeeping,
        // no sampling in this form (dq_env_act_steps)
        if (obs_out) {                                                      // block-uniform
            __syncthreads();
            const int first = block * EPB, n_valid = min(EPB, p.n_envs - first), total = n_valid * p.obs_size;
            u8* g = obs_out + (size_t)first * p.obs_size;
            if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {
                const env_u32x4* s128 = reinterpret_cast<const env_u32x4*>(s_stage);
                env_u32x4* g128 = reinterpret_cast<env_u32x4*>(g);
                const int nq = total >> 4;
                for (int k = tid; k < nq; k += THREADS) g128[k] = s128[k];
                for (int k = (nq << 4) + tid; k < total; k += THREADS) g[k] = s_stage[k];
            } else {
                for (int k = tid; k < total; k += THREADS) g[k] = s_stage[k];
            }
            __syncthreads();
        }
        continue;
    }
    if (!p.obs && !p.stats) { env_inline_sampling<THREADS>(p, block); return; }    // block-uniform
    __syncthreads();                                                        // the stage (and the bookkeeping words) visible
    DQ_STAMP(DQ_TAG_ENV, 7);
    if (p.stats && tid < 4) {                                               // integer sums: order-independent; at most four atomics per block
        unsigned long long v = 0;
#pragma unroll
        for (int w = 0; w < EPB; ++w) v += s_est2[w][tid];
        if (v) atomicAdd(&p.stats[tid], v);
    }
    if (!p.obs) { env_inline_sampling<THREADS>(p, block); return; }
    {
        const int first = block * EPB;
        const int n_valid = min(EPB, p.n_envs - first);
        const int total = n_valid * p.obs_size;
        u8* g = p.obs + (size_t)first * p.obs_size;
        if ((reinterpret_cast<uintptr_t>(g) & 15) == 0) {                   // 16-byte stores (a store instruction moves 1 KB instead of 256 bytes: as dwords
            const env_u32x4* s128 = reinterpret_cast<const env_u32x4*>(s_stage);    // this copy was 5.6K of a riding block's 30K cycles)
            env_u32x4* g128 = reinterpret_cast<env_u32x4*>(g);
            const int nq = total >> 4;
            for (int k = tid; k < nq; k += THREADS) g128[k] = s128[k];
            for (int k = (nq << 4) + tid; k < total; k += THREADS) g[k] = s_stage[k];
        } else if ((reinterpret_cast<uintptr_t>(g) & 3) == 0) {
            const u32* s32 = reinterpret_cast<const u32*>(s_stage);
            u32* g32 = reinterpret_cast<u32*>(g);
            const int ndw = total >> 2;
            for (int k = tid; k < ndw; k += THREADS) g32[k] = s32[k];
            for (int k = (ndw << 2) + tid; k < total; k += THREADS) g[k] = s_stage[k];
        } else {
            for (int k = tid; k < total; k += THREADS) g[k] = s_stage[k];
        }
    }
    DQ_STAMP(DQ_TAG_ENV, 8);
    env_inline_sampling<THREADS>(p, block);
    DQ_STAMP(DQ_TAG_ENV, 9);
  }     // (steps of a MULTI launch; one trip otherwise)
}

// env.hip: validates the arguments of dq_env_act_step(_sample) and fills the parameters of a step WITHOUT launching it: the caller
// runs env_block<8> -- or, with p->pair, env_block2<16> -- on blocks [0, p->env_blocks + p->s_blocks) of its own 512-thread grid (256 threads: <4> / <8>)
// (p->env_blocks, p->s_blocks are set for that); *lds = the dynamic LDS those blocks need.
struct dq_env;
dq_status env_fill_act_step(dq_env* E, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                            int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev,
                            uint32_t* lifetime_dev, uint8_t* was_reset_dev, const dq_sample_job* sj, uint64_t* stats_dev, EnvParams* p,
                            size_t* lds, int threads = 512);      // threads per block of the carrying launch: 512 (env_block<8> / env_block2<16>) or 256 (<4> / <8>)
