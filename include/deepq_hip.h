/*
 * deepq_hip.h -- C ABI of the MI355X-native DeepQ-Decoding hot path (libdeepq_hip.so).
 *
 * The reference (R-Sweke/DeepQ-Decoding) is pure Python and has no FFI layer; its boundary for
 * this path is two duck-typed Python protocols: the gym-style environment
 * `Surface_Code_Environment_Multi_Decoding_Cycles` (example_notebooks/Environments.py:10-385) and
 * the keras-rl `DQNAgent` surface used by the driver scripts
 * (cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:109-152,206).  Each entry point
 * below names the reference interface it replaces.  The host-side mirror of those protocols
 * lives in `deepq-decoding_amd/` and reaches this library through ctypes.
 *
 * Conventions
 *   - every `*_dev` pointer is DEVICE memory owned by the caller (e.g. a torch tensor's data_ptr);
 *   - `stream` is a hipStream_t passed as void* (NULL = the null stream); every call only ENQUEUES
 *     work on it and returns; nothing here synchronises unless its comment says so;
 *   - no call allocates or frees device memory except *_create / *_destroy / dq_env_build_referee,
 *     so step-type calls are hipGraph-capturable;
 *   - return value 0 = DQ_OK, negative = error; dq_last_error() gives a thread-local message;
 *   - no exceptions cross the ABI; a handle must not be used from two host threads at once.
 *
 * Bit conventions (shared with oracle/lattice.py)
 *   - qubit q = row*d + col is bit q of a "qubit mask";
 *   - stabilizers are numbered in the order generate_faulty_syndrome draws its uniforms
 *     (Function_Library.py:189-221): bulk plaquettes (a,b), 1<=a,b<=d-1 row-major, then row 0
 *     (b=1,3,..), row d (b=2,4,..), column 0 (a=2,4,..), column d (a=1,3,..); stabilizer s is bit s
 *     of a "syndrome word";
 *   - action a is bit a of a 128-bit "action mask" stored as two uint64 (lo, hi);
 *   - referee tables are bit-packed: entry i is bit (i & 31) of word i >> 5; index bit k is the
 *     k-th live plaquette of that type in row-major (a,b) order.
 *
 * Random numbers: Philox4x32-10, key = seed, counter = (t_lo, t_hi, env_id, lane | stream << 16)
 * (SURVEY.md §8c).  Stream 0 = environment noise with t = the lattice's measurement-round counter:
 * word 0 < ceil(p_phys*2^32) <=> qubit `lane` errs, word 1 -> Pauli type 1 + ((w*3) >> 32),
 * word 2 < ceil(p_meas*2^32) <=> stabilizer `lane` is mis-measured.
 */
#ifndef DEEPQ_HIP_H
#define DEEPQ_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef int dq_status;
enum {
    DQ_OK = 0,
    DQ_ERR_INVALID = -1,     /* bad argument */
    DQ_ERR_UNSUPPORTED = -2, /* configuration outside what the kernels implement */
    DQ_ERR_HIP = -3,         /* a HIP runtime call failed */
    DQ_ERR_NOMEM = -4,
    DQ_ERR_STATE = -5,       /* call made in the wrong state (e.g. step before referee is set) */
    DQ_ERR_RANGE = -6        /* a value left the range the arithmetic carries (dq_qnet_range_check) */
};

/* DQ_MODEL_IIDXZ: independent X and Z flips per qubit (generate_IIDXZ_error, Function_Library.py:134-160: two uniforms per qubit, X
 * first) -- the noise channel generate_error() offers besides "X" and "DP" (Function_Library.py:91-92).  The reference's environment
 * constructor does not accept it (Environments.py:66-67 prints and then fails); here it runs with the depolarising model's action
 * layers (X and Z, or X/Y/Z with use_Y) and four homology classes (Function_Library.py:329-330). */
enum { DQ_MODEL_X = 0, DQ_MODEL_DP = 1, DQ_MODEL_IIDXZ = 2 };
enum { DQ_STREAM_ENV = 0, DQ_STREAM_POLICY = 1, DQ_STREAM_REPLAY = 2, DQ_STREAM_DROPOUT = 3, DQ_STREAM_INIT = 4 };

int dq_version(void);
/* sha256 (hex) over the CODE of the kernel sources this library was built from (csrc/, include/deepq_hip.h; comments and whitespace stripped:
 * deepq-decoding_amd/_digest.py): a binding compares it with the tree's, so that a prebuilt .so carried to another box is known to be the one these sources
 * build (tests/test_abi.py, the -m gpu suite). */
const char* dq_build_digest(void);
const char* dq_last_error(void);
/* sizeof() of the public structs as THIS library was compiled: 0 dq_env_cfg, 1 dq_env_info, 2 dq_sample_job, 3 dq_qnet_cfg, 4 dq_qnet_job, 5 dq_td_job,
 * 6 dq_env_step_job; -1 for any other id.  A binding (the ctypes structures of _lib.py) checks its own layouts against it: a struct of the wrong size handed
 * across the boundary is silent memory corruption (tests/test_abi.py).  No reference counterpart. */
long dq_struct_size(int id);
/* Number of visible HIP devices (0 if none / runtime unavailable).  Never fails. */
int dq_device_count(void);

/* ---------------------------------------------------------------------------------------------
 * Environment: replaces Surface_Code_Environment_Multi_Decoding_Cycles (Environments.py:10-385),
 * batched over n_envs independent lattices, one lattice per wavefront.
 * ------------------------------------------------------------------------------------------- */
typedef struct dq_env dq_env;

typedef struct {
    int32_t d;            /* code distance: 3, 5 or 7 (Environments.py:45; odd, Function_Library.py:28-29) */
    int32_t error_model;  /* DQ_MODEL_X / DQ_MODEL_DP (Environments.py:56-67) / DQ_MODEL_IIDXZ */
    int32_t use_Y;        /* Environments.py:60-65 */
    int32_t volume_depth; /* 1..16 (Environments.py:52) */
    int32_t n_envs;       /* lattices owned by this handle */
    uint32_t env_id_base; /* global id of lattice 0 (rank * n_local): results do not depend on sharding */
    uint32_t seed[2];     /* Philox key */
} dq_env_cfg;

typedef struct {
    int32_t num_actions;     /* Environments.py:57-64 */
    int32_t n_action_layers; /* Environments.py:58-65 */
    int32_t identity_index;  /* Environments.py:69 */
    int32_t obs_c, obs_h, obs_w; /* observation_space.shape, Environments.py:78-82 */
    int32_t n_stab;          /* d*d - 1 live stabilizers */
    int32_t state_words;     /* uint64 words per lattice in dq_env_export_state */
} dq_env_info;

dq_status dq_env_create(const dq_env_cfg* cfg, dq_env** out);     /* Environments.py:45-97 */
void dq_env_destroy(dq_env* env);
dq_status dq_env_get_info(const dq_env* env, dq_env_info* out);

/* env.p_phys / env.p_meas are plain attributes the drivers mutate between test sweeps
 * (Single_Point_Training_Script.py:200-201); converted to integer thresholds on the host. */
dq_status dq_env_set_rates(dq_env* env, double p_phys, double p_meas);

/* Referee ("static_decoder", Environments.py:53,144,150).  The reference's Keras referee blobs are
 * not in the checkout; the library builds the deterministic minimum-weight look-up referee defined
 * in oracle/referee.py ON THE GPU (level-synchronous BFS).  Synchronises `stream`. */
dq_status dq_env_build_referee(dq_env* env, void* stream);
/* ... or the maximum-likelihood referee for independent component flips with probability q_flip per qubit: exact
 * class posteriors by one XOR-convolution pass per qubit over the (syndrome, class) space, same table format.  Synchronises `stream`. */
dq_status dq_env_build_referee_ml(dq_env* env, double q_flip, void* stream);
/* ... or installs caller-provided bit-packed tables (device pointers, 2^((d*d-1)/2) bits each; must
 * stay alive while the handle uses them).  lut_z_dev may be NULL for DQ_MODEL_X. */
dq_status dq_env_set_referee(dq_env* env, const uint32_t* lut_x_dev, const uint32_t* lut_z_dev);
/* ... or ONE table over the whole syndrome: entry s (bit i of s = stabilizer i in generate_faulty_syndrome's draw order,
 * Function_Library.py:189-221) holds the class the referee predicts, X + 2 Z (generate_one_hot_labels_surface_code's index,
 * Function_Library.py:329-334), 2 bits per entry, 16 entries per word: 2^n_stab entries, n_stab <= 24 (d <= 5; 4 MB at d = 5).  This is
 * how an arbitrary `static_decoder.predict` object (Environments.py:144,150 -- only argmax of its output is used) runs inside the
 * kernel: the host tabulates it once (deepq-decoding_amd/env.py VectorEnv.set_referee_predict).  Device pointer, caller-owned. */
dq_status dq_env_set_referee_joint(dq_env* env, const uint32_t* lut_dev);
/* A Dense-stack referee evaluated ON THE DEVICE, for lattices whose syndromes no table holds (d = 7: 48 stabilizers) -- the reference's
 * own referee is such a network, "a fast feed-forward NN homology class predictor" loaded with keras load_model and called once per
 * step on the flattened (d+1)^2 syndrome of the state after the agent's move (Environments.py:53,139-144,
 * Single_Point_Training_Script.py:54-57); only the arg-max of its output is used (Environments.py:150).
 *   n_layers     1 .. 6 Dense layers, ReLU between them (the top layer's softmax is monotone and not evaluated)
 *   dims         int32 [n_layers + 1]: (d+1)^2, the hidden widths (<= 2048), then 2 ("X") or 4 classes
 *   weights_dev  float, caller-owned, must outlive its use: per layer the kernel [in][out] row-major (Keras shape), then the bias
 * Arithmetic (fixed, so that a host restatement gives the same bits -- referee.py FeedForwardReferee.predict_exact): float32, bias
 * first, inputs in increasing index order, one rounded multiply and one rounded add per term, first maximum of the outputs.
 * Every step then runs a pre-pass (one wavefront per lattice) before the environment kernel; dq_env_act_step selects the actions with
 * dq_policy_select first (legal_dev must hold the current legal sets, as every reset / step leaves them); the step does not ride on
 * the dense backward (dq_qnet_td_backward_*_env: DQ_ERR_UNSUPPORTED).  n_layers = 0 uninstalls it; installing any table referee does
 * too.  d <= 7. */
dq_status dq_env_set_referee_mlp(dq_env* env, int n_layers, const int32_t* dims, const float* weights_dev);
/* What the Dense-stack referee says about every lattice as it stands after action_dev[i] (int32 [n_envs]; the identity leaves the state as
 * it is): classes_dev uint8 [n_envs] = arg-max class.  The lattices are not stepped.  Tests and diagnostics (the step runs the same
 * pre-pass itself). */
dq_status dq_env_referee_classes(dq_env* env, const int32_t* action_dev, uint8_t* classes_dev, void* stream);

/* Copies the referee tables to host memory as one byte per entry (tests). Synchronises. */
dq_status dq_env_get_referee(dq_env* env, uint8_t* lut_x_host, uint8_t* lut_z_host, size_t entries);

/* reset(): Environments.py:99-115 + initialize_state 206-235 + reset_legal_moves 238-258.
 * which_dev == NULL resets every lattice, else lattice i iff which_dev[i] != 0 (others keep their
 * state; outputs are written for all lattices).  Nullable outputs are skipped.
 *   obs_dev      uint8 [n_envs, C, H, W]   (0/1 cells)
 *   legal_dev    uint64 [n_envs, 2]        action mask of env.legal_actions
 *   lifetime_dev uint32 [n_envs]           env.lifetime */
dq_status dq_env_reset(dq_env* env, const uint8_t* which_dev, uint8_t* obs_dev, uint64_t* legal_dev,
                       uint32_t* lifetime_dev, void* stream);

/* step(action): Environments.py:118-204.  An action outside [0, num_actions) is treated as the
 * identity.  auto_reset != 0: a lattice whose `done` flag is set when the call starts is reset
 * instead of stepped -- its action is ignored, reward 0, done 0 -- which is keras-rl's convention of
 * spending one agent step on the terminal observation before env.reset(); auto_reset == 0 keeps the
 * reference's sticky `done` (Environments.py:151 never clears it).
 *   action_dev    int32 [n_envs]
 *   reward_dev    float [n_envs]     (Environments.py:146-149)
 *   done_dev      uint8 [n_envs]     (Environments.py:150-151)
 *   was_reset_dev uint8 [n_envs]     1 where auto_reset replaced the step */
dq_status dq_env_step(dq_env* env, const int32_t* action_dev, int auto_reset, uint8_t* obs_dev,
                      float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev, uint32_t* lifetime_dev,
                      uint8_t* was_reset_dev, void* stream);

/* dq_policy_select (declared below; same rule, same Philox stream, on the lattices' CURRENT legal sets) followed by dq_env_step,
 * in one launch: keras-rl's `action = policy.select_action(q_values)` + `env.step(action)` of one agent step.  The selected
 * actions are written to action_dev (the replay memory needs them); q_dev == NULL explores always. */
dq_status dq_env_act_step(dq_env* env, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                          int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev,
                          uint64_t* legal_dev, uint32_t* lifetime_dev, uint8_t* was_reset_dev, void* stream);

/* dq_env_act_step plus the replay sampling (dq_replay_sample's rule, same Philox stream) for a later update, in the same launch: the
 * rule reads terminal flags no newer than slot head_slot - 3, so the job may describe the update that follows this step (head_slot /
 * filled_slots AFTER this step; head_slot = the slot the successor observations go to) or the one after the NEXT step (one more slot;
 * its newest candidate row is then the transition this very launch records). */
typedef struct dq_sample_job {
    const uint8_t* terminal_ring_dev;
    int n_slots, head_slot, filled_slots, batch;
    uint32_t seed[2];
    uint64_t t;                 /* number of the update the minibatch is for (counts from 1) */
    uint32_t sample_base;       /* first global sample id of this rank's minibatch */
    int32_t* index_dev;         /* int32 [batch] out */
} dq_sample_job;
dq_status dq_env_act_step_sample(dq_env* env, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                                 int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev,
                                 uint64_t* legal_dev, uint32_t* lifetime_dev, uint8_t* was_reset_dev, const dq_sample_job* sample,
                                 void* stream);

/* n_steps consecutive agent steps of an ACTING loop in one launch (round 6; SURVEY.md section 7 "hard parts": a 4096-lattice step is launch-latency-bound):
 * per step s = 0 .. n_steps - 1 -- dq_env_act_step with q_dev == NULL (uniform over the legal moves: keras-rl's warm-up / random policy), policy counter
 * t0 + s -- the transition lands in the caller's replay ring: slot c = (slot0 + s) mod n_slots of action_ring_dev int32 [n_slots][n_envs], reward_ring_dev
 * float [n_slots][n_envs] and done_ring_dev uint8 [n_slots][n_envs] (each nullable but the first), the successor observation in slot c + 1 (mod n_slots) of
 * obs_ring_dev uint8 [n_slots][n_envs][C][H][W] and / or patch_ring_dev uint32 [n_slots][n_envs][patch_stride_words] (nullable).  legal_dev / lifetime_dev /
 * was_reset_dev receive the LAST step's values.  Same bits as n_steps calls of dq_env_act_step(NULL, ...) with the same counters (tests run the C oracle's 60-step
 * comparison through this entry).  d <= 5 runs as ONE launch with the lattices' state in registers from step to step; d = 7 (and a Dense-stack referee) as
 * n_steps launches.  Does not consume a dq_env_patch_output arming. */
typedef struct dq_env_ring {
    int32_t* action_ring_dev;
    float* reward_ring_dev;
    uint8_t* done_ring_dev;
    uint8_t* obs_ring_dev;
    uint32_t* patch_ring_dev;
    int32_t patch_stride_words;
    int32_t n_slots, slot0;
} dq_env_ring;
dq_status dq_env_act_steps(dq_env* env, int n_steps, const uint32_t seed[2], uint64_t t0, const dq_env_ring* ring, int auto_reset, uint64_t* legal_dev,
                           uint32_t* lifetime_dev, uint8_t* was_reset_dev, void* stream);

/* Compact observation ("patch words").  The observation the reference builds -- padding_syndrome / padding_actions,
 * Environments.py:273-314 -- feeds Conv2D(64, 3, strides=2) (Function_Library.py:353): output pixel (oy, ox) of that convolution sees
 * the 3 x 3 patch at padded cell (2 oy, 2 ox), of which only the four CORNERS of every syndrome plane (even-even cells: grid cells
 * (oy + dy, ox + dx) of the faulty syndrome, Environments.py:292-294) and the CENTRE of every action plane (the odd-odd cell of qubit
 * oy d + ox, Environments.py:309-312) are data; every other cell is a constant of the embedding (Environments.py:284-298).  A lattice's
 * observation is therefore d * d words, one per output pixel p = oy d + ox:
 *     bit 4 j + 2 dy + dx   = faulty syndrome plane j (j < volume_depth) at grid cell (oy + dy, ox + dx)
 *     bit 4 volume_depth + l = action plane l (l < n_action_layers) at qubit p
 * (the padded uint8 image is a fixed function of these words and back; 4 volume_depth + n_action_layers <= 32 required, else
 * DQ_ERR_UNSUPPORTED).  This call ARMS the handle: the NEXT launch that resets or steps its lattices -- dq_env_reset, dq_env_step,
 * dq_env_act_step(_sample), or the step riding on dq_qnet_td_backward_*_env -- ALSO writes uint32 patch_dev[i * stride_words + p] for every
 * lattice i (stride_words >= d * d; the words between d * d and the stride are left alone); that launch's obs_dev may then be NULL.
 * One call arms one launch; patch_dev == NULL disarms.  The Q-network reads such rows directly (dq_qnet_job.reserved bit 0,
 * dq_qnet_set_patch_input).  No reference counterpart: the reference stores the padded int64 image (6.8 KB per d = 5 observation). */
dq_status dq_env_patch_output(dq_env* env, uint32_t* patch_dev, int stride_words);

/* Hidden state for tests / checkpointing: uint64 [n_envs, state_words], state_words = 11 + volume_depth:
 *   0 xmask (hidden_state codes 1,2)   1 zmask (codes 2,3)
 *   2 current_true_syndrome word       3 OR of the volume's faulty words (summed_syndrome_volume != 0)
 *     (2 and 3 are recomputed on the device at export and ignored by import)
 *   4 acted_on_qubits   5 measurement-round counter   6,7 completed_actions mask   8,9 legal_actions mask
 *   10 lifetime | done << 32           11.. faulty syndrome words of the current volume. */
dq_status dq_env_export_state(dq_env* env, uint64_t* state_dev, void* stream);
dq_status dq_env_import_state(dq_env* env, const uint64_t* state_dev, void* stream);

/* Static lattice tables as the kernels use them (tests compare them with the reference's
 * generateSurfaceCodeLattice / get_stabilizer_list / get_qubit_neighbour_list outputs).
 * Host pointers, each 64 entries: stab_qmask[s], qubit_smask[q], neigh_qmask[q]; stab_type[s] in {1,3,0}. */
dq_status dq_env_get_tables(const dq_env* env, uint64_t* stab_qmask, uint64_t* qubit_smask,
                            uint64_t* neigh_qmask, uint8_t* stab_type);

/* ---------------------------------------------------------------------------------------------
 * Matching referee: the referee of Environments.py:53,144,150 for lattices whose syndrome space no longer fits
 * a look-up table (d >= 9; README.md:278 allows "any perfect-measurement decoding algorithm").  Same definition as dq_env_build_referee's
 * tables -- class 1 iff the lightest error with that syndrome and class 1 is strictly lighter than the lightest with class 0 -- computed
 * per syndrome: minimum-weight perfect matching of the defects (with each other or a boundary) with class bookkeeping, solved exactly by
 * dynamic programming over defect subsets, one wavefront per syndrome (csrc/match_dev.h; numpy restatement oracle/matching_referee.py).
 * Equals the look-up referee at d <= 7 for every syndrome.  3 <= d <= 15, odd.
 *   defects_dev  uint64 [batch][2 components][2 words]: bit i of a component = its i-th plaquette in row-major (a, b) order (the
 *                look-up referee's index convention); component 0 = type-3 plaquettes (X part), 1 = type-1 plaquettes (Z part)
 *   class_dev    uint8 [batch]: X part (+ 2 * Z part when both_components != 0) -- generate_one_hot_labels_surface_code's index
 *   inexact_dev  uint8 [batch] or NULL: 1 where the answer is not exact.  The defects are split into CLUSTERS (pairs that can never be worth
 *                matching -- two boundary paths of the same total class are no longer -- separate them; the clusters' weights combine exactly) and
 *                a cluster of up to max_defects (20) defects is matched exactly: up to 14 in LDS, beyond that in a scratch table in device
 *                memory (rare, slower).  Inexact: a cluster beyond max_defects (its lowest max_defects matched exactly, the others sent to
 *                their nearer boundary) or more than 32 defects in one component (those beyond the 32nd likewise)
 * ------------------------------------------------------------------------------------------- */
typedef struct dq_match dq_match;
dq_status dq_match_create(int d, dq_match** out);
void dq_match_destroy(dq_match* m);
dq_status dq_match_info(const dq_match* m, int* nodes_per_component, int* max_defects, int* w10);
/* host copies of one component's tables (tests): dist uint8 [n][n][2], distB uint8 [n][2] (255: no such path), w10 */
dq_status dq_match_get_tables(const dq_match* m, int component, uint8_t* dist_host, uint8_t* distB_host, int* w10);
dq_status dq_match_decode(const dq_match* m, const uint64_t* defects_dev, int batch, int both_components, uint8_t* class_dev,
                          uint8_t* inexact_dev, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Wide environment: the same Surface_Code_Environment_Multi_Decoding_Cycles (Environments.py:10-385) for lattices beyond one 64-bit word
 * per bit-plane -- any odd 3 <= d <= 15, i.e. the d >= 9 the dq_env_* kernels (one word per plane, look-up referee) cannot hold
 * (csrc/env_big.hip).  Same configuration struct, same call semantics and Philox streams as dq_env_*; differences:
 *   - legal sets are legal_words = ceil(num_actions / 64) uint64 per lattice (dq_envb_get_info), legal_dev is uint64 [n_envs][legal_words];
 *   - the referee is the matching referee above, built by dq_envb_create and evaluated inside every step (inexact_dev, uint8 [n_envs]
 *     or NULL, reports where its more-than-max_defects fallback was used);
 *   - dq_envb_export_state: uint64 [n_envs][state_words] = x[W] z[W] true_syndrome[W] summed_volume[W] acted[W] round completed[LW]
 *     legal[LW] (lifetime | done << 32) volume[depth][W], W = ceil(d*d / 64), LW = legal_words.
 * At d <= 7 every output equals dq_env_*'s (with its default look-up referee) bit for bit.
 * ------------------------------------------------------------------------------------------- */
typedef struct dq_envb dq_envb;
dq_status dq_envb_create(const dq_env_cfg* cfg, dq_envb** out);
void dq_envb_destroy(dq_envb* env);
dq_status dq_envb_get_info(const dq_envb* env, dq_env_info* out, int* legal_words);
dq_status dq_envb_set_rates(dq_envb* env, double p_phys, double p_meas);
dq_status dq_envb_reset(dq_envb* env, const uint8_t* which_dev, uint8_t* obs_dev, uint64_t* legal_dev, uint32_t* lifetime_dev, void* stream);
dq_status dq_envb_step(dq_envb* env, const int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev,
                       uint64_t* legal_dev, uint32_t* lifetime_dev, uint8_t* was_reset_dev, uint8_t* inexact_dev, void* stream);
/* dq_policy_select_wide on the lattices' current legal sets followed by dq_envb_step, in one launch (cf. dq_env_act_step) */
dq_status dq_envb_act_step(dq_envb* env, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                           int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev,
                           uint32_t* lifetime_dev, uint8_t* was_reset_dev, uint8_t* inexact_dev, void* stream);
dq_status dq_envb_export_state(dq_envb* env, uint64_t* state_dev, void* stream);
/* dq_policy_select (below) over legal sets of legal_words uint64 per lattice: same rule, same Philox words */
dq_status dq_policy_select_wide(const float* q_dev, const uint64_t* legal_dev, int n, int n_actions, int legal_words, double eps,
                                int masked_greedy, const uint32_t seed[2], uint32_t env_id_base, uint64_t t, int32_t* action_dev,
                                void* stream);

/* ---------------------------------------------------------------------------------------------
 * Action selection: replaces EpsGreedyQPolicy / GreedyQPolicy(masked_greedy=...) of the keras-rl
 * fork (call sites Single_Point_Training_Script.py:110-115,166-167; README.md:168,262).
 *   q_dev      float [n, n_actions] or NULL (then every lattice explores: uniform over legal)
 *   legal_dev  uint64 [n, 2]
 * Per lattice i, words w = Philox(key=seed, ctr=(t_lo, t_hi, env_id_base+i, DQ_STREAM_POLICY<<16)):
 *   explore <=> w[1] < ceil(eps * 2^32);  explore action = k-th smallest legal action,
 *   k = (w[0] * n_legal) >> 32;  otherwise argmax_a q (first maximum), over the legal set only when
 *   masked_greedy != 0.
 * ------------------------------------------------------------------------------------------- */
dq_status dq_policy_select(const float* q_dev, const uint64_t* legal_dev, int n, int n_actions, double eps,
                           int masked_greedy, const uint32_t seed[2], uint32_t env_id_base, uint64_t t,
                           int32_t* action_dev, void* stream);


/* ---------------------------------------------------------------------------------------------
 * Q-network: replaces the Keras model of build_convolutional_nn (Single_Point_Training_Script.py:61-90)
 * under keras-rl's dueling head (DQNAgent(enable_dueling_network=True), :119-127):
 *   Conv2D(valid, channels_first)+ReLU x n_conv -> Flatten -> [Dense+ReLU+Dropout] x n_ff -> Dense(n_actions)
 *   -> (dueling) Dense(n_actions+1), Q = y0 + y[1:] - mean(y[1:]).
 * Parameters live in ONE flat float buffer owned by the caller, in Keras order and Keras shapes (conv
 * kernels HWIO, dense (in,out)), each layer kernel then bias -- the tensors of a Keras .h5f drop in as is.
 * f32 results throughout (the reference is fp32 and the parity bound is 1e-5).  The per-layer path multiplies on the f32-input MFMA;
 * the fused chains carry every f32 operand as two f16 pieces (22 significant bits, round to nearest) and issue three f16 MFMAs per
 * product with f32 accumulation (two where an operand is binary): error below that of an ordinary f32 GEMM's accumulation, different
 * summation order (csrc/qnet.h "f16x2").  Operands of the fused chains must be finite and below 65504 in magnitude.
 * ------------------------------------------------------------------------------------------- */
typedef struct dq_qnet dq_qnet;

typedef struct {
    int32_t in_c, in_h, in_w;   /* env.observation_space.shape (Single_Point_Training_Script.py:108) */
    int32_t n_conv;             /* 1..4 */
    int32_t conv[4][3];         /* [filters, kernel, stride]  ("c_layers"); stride > 1 only on the first */
    int32_t n_ff;               /* 0..4 hidden dense layers ("ff_layers") */
    int32_t ff_units[4];
    float ff_dropout[4];
    int32_t n_actions;          /* env.num_actions */
    int32_t dueling;            /* enable_dueling_network */
    int32_t max_batch;          /* workspaces are sized for this many samples */
} dq_qnet_cfg;

dq_status dq_qnet_create(const dq_qnet_cfg* cfg, dq_qnet** out);
void dq_qnet_destroy(dq_qnet* net);
size_t dq_qnet_param_count(const dq_qnet* net);
int dq_qnet_num_layers(const dq_qnet* net);
/* Offsets (in floats) of layer `layer`'s kernel and bias in the flat buffer and the kernel's Keras shape. */
dq_status dq_qnet_layer_info(const dq_qnet* net, int layer, int64_t* kernel_offset, int64_t* bias_offset,
                             int32_t shape[4], int32_t* n_dims);

/* Two forward implementations exist, both HIP: per-layer implicit GEMMs, and (default, when the configuration
 * fits) the fused LDS-resident chains of csrc/fused.hip.  dq_qnet_set_fused(net, 0) selects the per-layer path.  Each path's
 * training forward saves its activations in the form ITS backward reads (the fused one mostly as f16 piece planes), so a backward
 * must run on the path its training forward ran on: switching between the two calls makes the backward return DQ_ERR_STATE. */
dq_status dq_qnet_set_fused(dq_qnet* net, int enable);
/* Where the fused path has several FORMS of a kernel (same results to round-off, different summation order), which one this handle runs; a negative value keeps
 * the current choice.  conv_forward_form: 0 = conv_wave_kernel where it applies (patch words, d = 5), 1 = the workgroup-per-group kernels.  conv_backward_form:
 * 0 = conv_bwd16_kernel where it applies and the minibatch is >= 1024, 1 = conv_bwd_chain_kernel always, 2 = conv_bwd16_kernel whatever the minibatch.
 * conv_backward_a1: 0 = where a conv_wave_kernel training forward is followed by conv_bwd16_kernel, the forward does not save the first convolution's output and
 * the backward recomputes it from the patch words (the forward's own instructions: bit-identical gradients, 52 MB less HBM traffic per update at 4096 samples),
 * 1 = every training forward saves it and every backward reads the saved planes.  Set between a training forward and its backward, a change that makes the
 * backward need what the forward did not save is refused there (DQ_ERR_STATE).  The initial values come from DQ_CONV_FORM (=group -> 1) / DQ_CONV_BWD_FORM
 * (=8 -> 1, =16 -> 2) / DQ_CONV_BWD_A1 (=saved -> 1) read ONCE by dq_qnet_create; nothing re-reads the environment per call, so every rank that created its handle
 * under the same environment sums in the same order.  A/B measurements and tests.  No reference counterpart. */
dq_status dq_qnet_set_kernel_forms(dq_qnet* net, int conv_forward_form, int conv_backward_form, int conv_backward_a1);
int dq_qnet_fused_supported(const dq_qnet* net);

/* model.predict_on_batch (training == 0) / the forward half of train_on_batch (training != 0: dropout
 * active, activations kept for dq_qnet_backward).
 *   obs_dev    uint8 [rows, C, H, W]; sample b reads row  b                         if index_dev == NULL,
 *                                                     row (index_dev[b] + index_off) mod index_mod otherwise
 *              (the replay-minibatch gather happens inside the first convolution's loader).  Rows are copied as whole
 *              aligned 16-byte words, so the words that straddle a row's ends are read too: obs_dev must point into an
 *              allocation that starts 16-byte aligned and whose allocated size is a multiple of 16 (any hipMalloc / torch
 *              allocation: both round sizes up to at least 256 bytes; views at arbitrary byte offsets inside it are fine);
 *   q_dev      float [batch, n_actions];
 *   dropout    one Philox call covers eight consecutive units of a sample, 16 bits per decision:
 *              keep(b, j) <=> half-word (j & 7) of Philox(key=seed, ctr=(t_lo, t_hi, sample_base + b, (j>>3) | DQ_STREAM_DROPOUT<<16))
 *              >= ceil(rate * 2^16)   (half-word h = bits 16 (h & 1) .. + 15 of word h >> 1);  kept units are scaled by
 *              1/(1-rate) (Keras K.dropout). */
dq_status dq_qnet_forward(dq_qnet* net, const float* params_dev, const uint8_t* obs_dev, const int32_t* index_dev,
                          int index_off, int index_mod, int batch, int training, const uint32_t seed[2], uint64_t t,
                          uint32_t sample_base, float* q_dev, void* stream);

/* Several forwards in ONE pair of launches (e.g. Q_target(s1), Q_online(s1) and the training forward on s0 of one DQN
 * update): each job has the meaning of one dq_qnet_forward call; at most 4 jobs, at most one of them training.  With the
 * fused chains a lone 4096-sample forward is one workgroup per CU; sharing a grid lets the jobs' phases overlap. */
typedef struct {
    const float* params_dev;
    const uint8_t* obs_dev;
    const int32_t* index_dev;   /* nullable */
    int32_t index_off, index_mod, batch, training;
    uint32_t seed[2];
    uint64_t t;
    uint32_t sample_base;
    uint32_t reserved;          /* bit 0: obs_dev holds PATCH WORDS (dq_env_patch_output) -- uint32 [rows, stride_words], 16-byte aligned rows -- instead
                                 * of padded uint8 images (needs dq_qnet_set_patch_input; every job of a launch in the same form); other bits 0 */
    float* q_dev;
    const void* packed_dev;     /* dq_qnet_pack(params_dev) output, or NULL: the call packs the weights itself (one extra small launch) */
} dq_qnet_job;
dq_status dq_qnet_forward_multi(dq_qnet* net, int n_jobs, const dq_qnet_job* jobs, void* stream);

/* Patch-word input.  Declares that the network's first `n_syndrome_planes` input planes are padding_syndrome planes and the rest
 * padding_actions planes (Environments.py:273-314; the observation of Surface_Code_Environment_Multi_Decoding_Cycles with
 * volume_depth = n_syndrome_planes), so that Conv2D(64, 3, strides=2) (Function_Library.py:353) can read an observation as the d * d
 * patch words of dq_env_patch_output instead of the (2d+1)^2 uint8 planes: one K = 32 matrix block per tile from the words' bits, the
 * embedding's constant cells folded into a per-pixel bias (forward) / five shared gradient columns (backward).  Same function of the
 * same weights: Q-values and gradients agree with the uint8 path to f32 round-off (both meet the 1e-5 bound against the float64 oracle).
 * stride_words = words per observation row (a power of two, max(4, d * d) <= stride_words <= 64).  Fused chains only, d <= 7,
 * 4 * n_syndrome_planes + action planes <= 32 (DQ_ERR_UNSUPPORTED otherwise); n_syndrome_planes = 0 switches it off.  Call it BEFORE
 * dq_qnet_pack: the packed buffer carries the compact first kernel and the per-pixel bias.  Jobs select the form per launch
 * (dq_qnet_job.reserved bit 0); dq_qnet_forward always reads uint8 images.  No reference counterpart. */
dq_status dq_qnet_set_patch_input(dq_qnet* net, int n_syndrome_planes, int stride_words);

/* The fused chains read the conv kernels and Dense(512) as f16 pieces in matrix-core operand order.  A caller that runs several
 * forwards on the same weights packs them once per parameter change (dq_qnet_packed_bytes(net) bytes of device memory) and
 * passes the buffer in dq_qnet_job.packed_dev; it MUST repack after every change of params_dev (dq_adam_step, weight loading).
 * dq_qnet_forward and jobs with packed_dev == NULL pack on every call.  The backward reuses the training forward's pack. */
size_t dq_qnet_packed_bytes(const dq_qnet* net);
dq_status dq_qnet_pack(const dq_qnet* net, const float* params_dev, void* packed_dev, void* stream);

/* Backward half of train_on_batch: grads_dev[n_params] = d/dparams sum(dq * Q) for the last training forward
 * (obs_dev / index_dev of that call must still be valid).  Deterministic (fixed-order reductions). */
dq_status dq_qnet_backward(dq_qnet* net, const float* params_dev, const float* dq_dev, float* grads_dev, void* stream);

/* The fused backward carries its gradients multiplied by a power of two S (undone, exactly, by its final reduction) so that they sit in
 * the f16 pieces' range (csrc/fused_bwd.hip "gradient scale").  With the TD step fused in (dq_qnet_td_backward_*), S follows from
 * td->grad_scale.  A backward that is handed dq_dev measures max |dq| on the device first (one small extra launch) -- unless the caller
 * declares here the loss scale its dq was computed with (dq = TD error x grad_scale, as dq_td_update's argument): then the same S as in
 * the fused TD path is used, and the two paths give the same bits.  0 (default) = not declared.  No reference counterpart. */
dq_status dq_qnet_set_grad_scale(dq_qnet* net, double grad_scale);

/* A mark inside the NEXT fused backward of this network (dq_qnet_backward*, dq_qnet_td_backward*): `hip_event` (a hipEvent_t) is recorded on that
 * call's stream right behind the launch of the convolutional backward, i.e. in front of the final reduction / optimizer step and whatever the
 * caller launches next -- the point from which the device has idle capacity until the next forward (the reduction and the weight repacking are a
 * few hundred small workgroups).  A caller with independent work for another stream (the NEXT update's target-network forward: DQNCore's extra
 * updates of a vector step) makes that stream wait for the mark.  One-shot: consumed by one backward; NULL clears it.  Ignored (and cleared)
 * by backwards that do not run the fused convolutional kernel.  No reference counterpart. */
dq_status dq_qnet_mark_conv_backward(dq_qnet* net, void* hip_event);

/* Range guard of the fused backward.  Its gradients travel as f16 pieces (finite up to 65504 after the scale S above): a TD error so
 * large that some S x gradient leaves that range becomes inf / NaN in the weight gradient, where fp32 arithmetic (the reference's
 * TensorFlow) would still be finite.  That is never silent: the final reduction raises a device-side flag for every non-finite
 * gradient element and, when the optimizer step rides on it (dq_qnet_backward_adam, dq_qnet_td_backward_adam*), leaves that element's
 * parameter and moments untouched.  This call synchronises `stream`, returns DQ_ERR_RANGE if the flag was raised since the last call
 * and clears it; DQ_OK otherwise (always on the per-layer f32 path).  The agent loop calls it at its host synchronisation points.  With
 * S x grad_scale in [4, 8) the guard trips for |TD error| of a few thousand (the reference's recorded losses, trained_models/ * / * /
 * training_history.json, stay below 160, i.e. |TD error| ~ 20).  No reference counterpart. */
/* (round 6) The fused FORWARD is guarded too: its activations travel as f16 pieces as well, and one of 65504 or more would turn into inf, the products it enters
 * into NaN and the next ReLU into 0 -- finite, wrong Q-values.  Every layer's epilogue compares what it splits with the range, the packing launch checks every
 * parameter (finite, < 65504), the dense chain checks the Q-values it stores; any of them raises the same device-side flag and this call returns DQ_ERR_RANGE with
 * a message that starts "dq_qnet_range_check[forward]" (the agent loop treats that one as fatal: no gradient scale helps a diverged network). */
dq_status dq_qnet_range_check(dq_qnet* net, void* stream);
/* How many optimizer steps the guard's early half discarded WHOLE since the last call (the TD step saw a sample beyond the host-known scale's range: the
 * final reduction wrote NaN into every gradient element and moved no parameter -- with several ranks the all-reduce carries the NaNs to all of them and
 * dq_qnet_adam_step counts on each); synchronises `stream`, clears the count.  What the reference (delta_clip = inf, fp32) would have applied and this
 * path did not: the agent loop logs it.  No reference counterpart. */
dq_status dq_qnet_range_discarded(dq_qnet* net, unsigned* count, void* stream);

/* The same backward in two phases, for overlapping the gradient all-reduce with compute on several GPUs (no reference
 * counterpart): phase 0 = dueling + dense layers -> grads_dev[dq_qnet_conv_param_count(net) ..) complete; phase 1 = the
 * convolutions -> grads_dev[0 .. dq_qnet_conv_param_count(net)).  Phase 0 must run first; dq_dev is only read by phase 0. */
dq_status dq_qnet_backward_phase(dq_qnet* net, const float* params_dev, const float* dq_dev, float* grads_dev, int phase,
                                 void* stream);
size_t dq_qnet_conv_param_count(const dq_qnet* net);

/* dq_qnet_backward followed by dq_adam_step on the whole parameter vector, with the optimizer step applied by the backward's
 * final reduction launch (one launch fewer; same bits as the two separate calls).  Single-process training only: with several
 * ranks the gradient has to be all-reduced between the two.  grads_dev still receives the gradient. */
dq_status dq_qnet_backward_adam(dq_qnet* net, float* params_dev, const float* dq_dev, float* grads_dev, float* m_dev, float* v_dev,
                                double lr, double beta_1, double beta_2, double epsilon, uint64_t t, void* stream);

/* Episode records of a greedy evaluation -- keras-rl Agent.test's per-episode log (episode_reward, nb_steps, and the fork's
 * episode_lifetime; Single_Point_Training_Script.py:206-207) -- kept on the device, one call per vector step behind the environment
 * step: lattice i's running reward / length advance (not on a step spent being reset); a lattice that ends an episode while
 * quota_dev[i] > 0 appends int32 {step, i, reward bits (float), length, lifetime} to records_dev [capacity][5] (slot = atomic increment
 * of counter_dev[0]; sort by (step, lattice) for the serial loop's order) and pays one unit of quota.  The host only reads counter_dev
 * every few dozen steps to see whether the evaluation is complete.  No reference counterpart. */
dq_status dq_test_bookkeeping(const uint8_t* done_dev, const uint8_t* was_reset_dev, const float* reward_dev, const uint32_t* lifetime_dev, int n,
                              int step, int32_t* quota_dev, float* ep_reward_dev, int32_t* ep_len_dev, int32_t* records_dev, int capacity,
                              int32_t* counter_dev, void* stream);

/* dq_adam_step on the network's whole parameter vector, for the several-GPU branch (backward with m_dev == v_dev == NULL, all-reduce of
 * grads_dev, then this): an element whose all-reduced gradient is not finite -- the fused backward's range guard, on whichever rank it
 * tripped: the sum carries it to all of them -- is skipped AND raises this handle's range flag, so that dq_qnet_range_check reports
 * DQ_ERR_RANGE on every rank at the same synchronisation point and the replicas stay identical.  No reference counterpart. */
dq_status dq_qnet_adam_step(dq_qnet* net, float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, double lr, double beta_1,
                            double beta_2, double epsilon, uint64_t t, void* stream);

/* The learner half of one DQNAgent.backward in the fewest launches: dq_td_update (+ dq_episode_stats when n > 0) computed in the
 * dense backward's first kernel, then dq_qnet_backward, then dq_adam_step on the final reduction.  Same y / dq / gradient / parameter
 * bits as the separate calls; the loss / mean_q partials are summed in a different order (dq_td_metrics reads them the same way).
 * y_dev, dq_dev (needed only by the per-layer path), metrics_dev nullable.
 * m_dev == v_dev == NULL (dq_qnet_td_backward_adam, _adam_env, dq_qnet_backward_adam): the gradient only, no optimizer step -- the several-GPU
 * path, where the all-reduce of grads_dev comes between the backward and dq_adam_step (one final reduction launch instead of the two of the
 * phase calls). */
typedef struct dq_td_job {
    const float* q_online_s1_dev;
    const float* q_target_s1_dev;
    const float* q_s0_dev;
    const float* reward_dev;        /* replay ring arrays, indexed through index_dev */
    const uint8_t* terminal_dev;
    const int32_t* action_dev;
    const int32_t* index_dev;
    double gamma, grad_scale;
    int batch, n_actions;
    float* y_dev;
    float* dq_dev;
    float* metrics_dev;
    const uint8_t* done_dev;        /* episode bookkeeping of the step just taken (dq_episode_stats arguments); n == 0: none */
    const uint8_t* was_reset_dev;
    const uint32_t* lifetime_dev;
    const float* step_reward_dev;
    int n;
    uint64_t* stats_dev;
    int auto_scale;                 /* fused backward only.  0: its gradients are carried at the power-of-two scale that follows from grad_scale (TD errors up to
                                     * a few thousand; beyond: dq_qnet_range_check).  1: the scale is MEASURED -- max |TD error x grad_scale| of this minibatch, by one
                                     * small launch in front of the backward -- so that any finite TD error fp32 can hold is carried, as in the reference's
                                     * TensorFlow arithmetic (keras-rl delta_clip = inf, Single_Point_Training_Script.py:119-127) */
} dq_td_job;
dq_status dq_qnet_td_backward_adam(dq_qnet* net, float* params_dev, const dq_td_job* td, float* grads_dev, float* m_dev, float* v_dev,
                                   double lr, double beta_1, double beta_2, double epsilon, uint64_t t, void* stream);
/* dq_env_act_step(_sample) of the SAME vector step riding on dq_qnet_td_backward_adam's first launch.  The update does not read what
 * the step writes (its minibatch never holds the newest transition, dq_replay_sample's rule) and the step does not read what the
 * update writes (it acts on q_dev, already computed), so the two are one launch: the lattices' blocks fill the idle issue slots of
 * the dense backward chain instead of taking a launch of their own.  The step's episode bookkeeping (dq_episode_stats' sums into
 * stats_dev, nullable) is done by the lattices' blocks themselves; td->n must be 0.  Same bits as the separate calls.  Fused chains
 * only (DQ_ERR_UNSUPPORTED otherwise: make the separate calls).  Replaces, per vector step of the reference's loop,
 * `action = policy.select_action(q)`, `env.step(action)`, `memory.append` and `DQNAgent.backward`
 * (cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:138-152 -> keras-rl Agent.fit). */
typedef struct dq_env_step_job {
    const float* q_dev;             /* dq_env_act_step's arguments */
    double eps;
    int masked_greedy;
    uint32_t seed[2];
    uint64_t t;
    int32_t* action_dev;
    int auto_reset;
    uint8_t* obs_dev;
    float* reward_dev;
    uint8_t* done_dev;
    uint64_t* legal_dev;
    uint32_t* lifetime_dev;
    uint8_t* was_reset_dev;
    const dq_sample_job* sample;    /* nullable: dq_env_act_step_sample's look-ahead draw */
    uint64_t* stats_dev;            /* nullable: uint64 [4] accumulators of dq_episode_stats */
} dq_env_step_job;
dq_status dq_qnet_td_backward_adam_env(dq_qnet* net, float* params_dev, const dq_td_job* td, float* grads_dev, float* m_dev, float* v_dev,
                                       double lr, double beta_1, double beta_2, double epsilon, uint64_t t, dq_env* env,
                                       const dq_env_step_job* step, void* stream);
/* The several-GPU form: the TD step + phase 0 of dq_qnet_backward_phase (dueling + dense layers) in one call; phase 1, the gradient
 * all-reduce and dq_adam_step follow as separate calls. */
dq_status dq_qnet_td_backward_phase0(dq_qnet* net, const float* params_dev, const dq_td_job* td, float* grads_dev, void* stream);
/* ... with the vector step's environment launch riding on it (as dq_qnet_td_backward_adam_env). */
dq_status dq_qnet_td_backward_phase0_env(dq_qnet* net, const float* params_dev, const dq_td_job* td, float* grads_dev, dq_env* env,
                                         const dq_env_step_job* step, void* stream);

/* ---------------------------------------------------------------------------------------------
 * DQN update: replaces SequentialMemory.sample + DQNAgent.backward + keras Adam of the keras-rl fork
 * (Single_Point_Training_Script.py:109,119-130).
 * Device replay ring: row r = slot*n_envs + env stores (observation, action, reward, terminal) of the
 * step taken from that observation; its successor observation is row r + n_envs (mod n_slots*n_envs).
 * ------------------------------------------------------------------------------------------- */
/* Uniform minibatch rows over exactly the transitions upstream keras-rl 0.4.2 SequentialMemory.sample can return
 * (oracle/memory_oracle.py): idx = sample_batch_indexes(window_length, nb_entries - 1) + 1 and the transition used is idx - 1, i.e.
 * every stored transition except the newest (its successor observation is not in keras-rl's memory yet) and entry 0, redrawing those
 * whose predecessor entry was terminal (terminals[idx - 2]).  head_slot = slot of the newest observation, filled_slots = slots written
 * so far (4 <= filled_slots <= n_slots; nb_entries = filled_slots - 1).
 * As in keras-rl (random.sample, Single_Point_Training_Script.py:109 -> SequentialMemory.sample), the FIRST draws of a minibatch are
 * WITHOUT replacement whenever the M = (filled_slots - 3) * n_envs candidate rows are at least `batch`: sample b takes candidate
 * pi_t((sample_base + b) mod M), pi_t a keyed bijection of [0, M) (four-round Feistel network + cycle walking; round keys =
 * Philox(key=seed, ctr=(t_lo, t_hi, 0xffffffff, 0xffff | DQ_STREAM_REPLAY<<16)); csrc/common.h dq_replay_permute); candidate
 * c = slot head_slot - 2 - c / n_envs, lattice c mod n_envs.  A redraw (attempt >= 1) -- and every draw when batch > M, keras-rl's
 * with-replacement fallback -- is independent: Philox(key=seed, ctr=(t_lo, t_hi, sample_base + b, attempt | DQ_STREAM_REPLAY<<16)),
 * slot = head_slot - 2 - ((w0 * (filled_slots - 3)) >> 32), env = (w1 * n_envs) >> 32.  Terminal flags are read no newer than slot
 * head_slot - 3, so an update's minibatch may be drawn one vector step early (with the head_slot / filled_slots it WILL have). */
dq_status dq_replay_sample(const uint8_t* terminal_ring_dev, int n_envs, int n_slots, int head_slot, int filled_slots,
                           int batch, const uint32_t seed[2], uint64_t t, uint32_t sample_base, int32_t* index_dev,
                           void* stream);

/* The same for n_updates consecutive updates t0, t0 + 1, ... drawn on ONE ring state, in one launch: index_dev int32 [n_updates][batch], row u = what
 * dq_replay_sample gives for t = t0 + u.  (The reference trains one minibatch per environment step, Single_Point_Training_Script.py:119-127; with N
 * lattices per vector step that is several updates per step on the same memory -- their draws do not depend on each other.) */
dq_status dq_replay_sample_multi(const uint8_t* terminal_ring_dev, int n_envs, int n_slots, int head_slot, int filled_slots,
                                 int batch, const uint32_t seed[2], uint64_t t0, int n_updates, uint32_t sample_base,
                                 int32_t* index_dev, void* stream);

/* Double-DQN target: y_b = reward[r_b] + gamma * (1 - terminal[r_b]) * Q_target(s1_b)[argmax_a Q_online(s1_b)[a]],
 * r_b = index_dev ? index_dev[b] : b. */
dq_status dq_td_target(const float* q_online_s1_dev, const float* q_target_s1_dev, const float* reward_dev,
                       const uint8_t* terminal_dev, const int32_t* index_dev, double gamma, int batch, int n_actions,
                       float* y_dev, void* stream);

/* keras-rl clipped_masked_error with delta_clip = inf:  loss = mean_b 0.5 (Q[b,a_b] - y_b)^2.
 *   dq_dev      float [batch, n_actions] = grad_scale * (Q[b,a_b] - y_b) at a_b, 0 elsewhere
 *               (grad_scale = 1 / global batch, so an all-reduce SUM of gradients gives the global mean);
 *   metrics_dev float [DQ_TD_METRICS_FLOATS] (nullable): [0] = loss, [1] = mean_q = mean_b max_a Q[b,a] of this
 *               rank's minibatch; the rest is scratch for the fixed-order two-stage reduction. */
#define DQ_TD_METRICS_FLOATS 2050
dq_status dq_td_loss_grad(const float* q_s0_dev, const int32_t* action_dev, const int32_t* index_dev, const float* y_dev,
                          int batch, int n_actions, double grad_scale, float* dq_dev, float* metrics_dev, void* stream);

/* dq_td_target + dq_td_loss_grad in one launch (same arithmetic, same per-block metric partials), WITHOUT the final metric
 * reduction: call dq_td_metrics(metrics_dev, batch) before reading metrics_dev[0..1] (the training loop only reads them at its
 * logging interval).  y_dev nullable. */
dq_status dq_td_update(const float* q_online_s1_dev, const float* q_target_s1_dev, const float* q_s0_dev, const float* reward_dev,
                       const uint8_t* terminal_dev, const int32_t* action_dev, const int32_t* index_dev, double gamma, int batch,
                       int n_actions, double grad_scale, float* y_dev, float* dq_dev, float* metrics_dev, void* stream);
/* dq_td_update plus dq_episode_stats of the environment step just taken, in one launch. */
dq_status dq_td_update_stats(const float* q_online_s1_dev, const float* q_target_s1_dev, const float* q_s0_dev, const float* reward_dev,
                             const uint8_t* terminal_dev, const int32_t* action_dev, const int32_t* index_dev, double gamma, int batch,
                             int n_actions, double grad_scale, float* y_dev, float* dq_dev, float* metrics_dev, const uint8_t* done_dev,
                             const uint8_t* was_reset_dev, const uint32_t* lifetime_dev, const float* step_reward_dev, int n,
                             uint64_t* stats_dev, void* stream);
dq_status dq_td_metrics(float* metrics_dev, int batch, void* stream);

/* dq_replay_sample (for the next update) + dq_episode_stats (of the step just taken) in one launch. */
dq_status dq_post_step(const uint8_t* terminal_ring_dev, int n_envs, int n_slots, int head_slot, int filled_slots, int batch,
                       const uint32_t seed[2], uint64_t t, uint32_t sample_base, int32_t* index_dev, const uint8_t* done_dev,
                       const uint8_t* was_reset_dev, const uint32_t* lifetime_dev, const float* reward_dev, int n, uint64_t* stats_dev,
                       void* stream);

/* Episode bookkeeping the keras-rl fork does on the host per step (episode ends, env.lifetime of finished
 * episodes -- Single_Point_Training_Script.py:207 reads their rolling average): accumulates into uint64 stats_dev[4]
 * = {episodes ended, sum of their lifetimes, rewards earned, lattices stepped}.  was_reset_dev nullable. */
dq_status dq_episode_stats(const uint8_t* done_dev, const uint8_t* was_reset_dev, const uint32_t* lifetime_dev,
                           const float* reward_dev, int n, uint64_t* stats_dev, void* stream);

/* keras.optimizers.Adam (Keras 2.2): lr_t = lr*sqrt(1-b2^t)/(1-b1^t); m,v EMAs; p -= lr_t*m/(sqrt(v)+epsilon).
 * t = 1 for the first update. */
/* (Keras' update as it is: a non-finite gradient element propagates into its parameter and moments -- a diverged run stays visible.  The guarded
 * form -- such elements skipped and flagged -- is dq_qnet_adam_step and the optimizer step riding on the fused backward.) */
dq_status dq_adam_step(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, size_t n, double lr,
                       double beta_1, double beta_2, double epsilon, uint64_t t, void* stream);

/* ---------------------------------------------------------------------------------------------
 * Live kernel timing (measurement only; no reference counterpart).  dq_prof_arm(id, n) times up to n
 * launches of kernel family `id` (0 <= id < dq_prof_kernel_count(), names from dq_prof_kernel_name) with HIP
 * events on the stream they are launched on -- the fused chains and the environment step carry the event pair
 * on the launch itself (the dispatch's own start / end timestamps), the small per-layer kernels are bracketed
 * by one --; dq_prof_collect synchronises on the last recorded event and returns the number of launches
 * recorded since the last collect and the sum of their durations.  dq_prof_stride(s) (s >= 1; reset to 1 by
 * every dq_prof_arm) times only every s-th launch of the armed family: a timed launch costs the stream a few
 * microseconds, a sample of the launches leaves the timed region undisturbed.
 * dq_prof_arm(-1, 0) disarms.  Not thread-safe; one family at a time.
 * ------------------------------------------------------------------------------------------- */
int dq_prof_kernel_count(void);
const char* dq_prof_kernel_name(int kernel_id);
/* The kernel SYMBOL the family's most recent launch used ("" before its first launch): a family has several forms
 * (conv_chain_kernel: conv_wave_kernel / conv_chain_pkernel / conv_chain_kernel; conv_bwd_chain_kernel: conv_bwd16_kernel /
 * conv_bwd_chain_kernel), and a measurement must name the one that ran. */
const char* dq_prof_kernel_symbol(int kernel_id);
dq_status dq_prof_arm(int kernel_id, int max_launches);
dq_status dq_prof_stride(int stride);
dq_status dq_prof_collect(int* launches, double* total_ms);
/* The same with the shortest and the longest of the recorded launches (either pointer may be NULL): the spread a reader needs to tell a
 * slower box from a slower kernel. */
dq_status dq_prof_collect_spread(int* launches, double* total_ms, double* min_ms, double* max_ms);

#ifdef __cplusplus
}
#endif
#endif /* DEEPQ_HIP_H */
