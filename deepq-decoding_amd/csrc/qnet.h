// Q-network handle shared by qnet.hip (per-layer implicit GEMMs, backward) and fused.hip (fused forward chains).
#pragma once
#include "common.h"

#define QN_MAX_LAYERS 12
#define FWD_MAX_JOBS 4               // forwards served by one fused launch (dq_qnet_forward_multi)

struct Layer {
    int kind;                    // 0 conv, 1 dense
    int cin, cout, k, s, ih, iw, oh, ow;   // conv
    int nin, nout, relu;         // dense
    float dropout;
    size_t w_off, b_off;         // into the flat parameter buffer
    int K, N, rows;              // GEMM view: rows per sample (oh*ow or 1), K, N
};

struct dq_qnet {
    dq_qnet_cfg cfg;
    int n_layers;
    Layer L[QN_MAX_LAYERS];
    size_t n_params;
    int flat_c, flat_hw;         // last conv: channels and oh*ow (Keras Flatten permutation)
    float* act[2][QN_MAX_LAYERS];   // [set][layer] outputs; set 0 = training (kept for backward), 1 = inference
    float* gz[QN_MAX_LAYERS];    // gradient w.r.t. each layer's pre-activation output (what its weight gradient consumes)
    float* partial;              // wgrad slices
    size_t partial_floats;
    int last_train_batch;
    const uint8_t* last_obs;     // inputs of the last training forward (needed by conv1's weight gradient)
    const int32_t* last_index;
    int last_index_off, last_index_mod;
    float* xinf[FWD_MAX_JOBS];   // fused inference forwards: last-convolution output per job slot [max_batch, flat]
    int* kofftab;                // [96] first convolution: weight row k -> byte offset inside an NCHW uint8 observation, -1 past K
    void* pk_scratch[FWD_MAX_JOBS];  // packed weights of jobs that did not bring their own (dq_qnet_job.packed_dev == NULL)
    const void* last_train_packed;   // packed weights of the last training forward (the backward's data gradients read them)
    float* fpartial;             // fused backward workspace (fused_backward_workspace_floats)
    int use_fused;               // bit 0: fused LDS-resident chains when the configuration allows it; bit 1: the forward's convolutions
                                 // through the experimental wave pipeline (conv_pipe.hip) instead of conv_chain_kernel (fused.hip)
};


// ---- shared by the fused chains (fused.hip forward, fused_bwd.hip backward) ---------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned: global_load_dwordx4 accepts it
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define CHAIN_LDS_MAX (160 * 1024)
#define DENSE_THREADS 512
#define DENSE_WAVES 8
#define DENSE_ROWS 16
#define DENSE_HID 512                 // Dense(512): 8 waves x 64 columns; Dense(|A|) splits K = 512 into 8 x 64
static inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }

#ifdef __HIPCC__
// ---- f32-accurate contraction on the bf16 matrix pipe ("bf16x6") ----------------------------------------------------------
// Every f32 value splits EXACTLY into three bf16 pieces x = hi + mid + lo (8+8+8 mantissa bits, by truncation).  A product a*b is then
// the sum of nine exact piece products; the six with piece-index sum <= 4 carry it to ~2^-24 relative (the f32 rounding class), so
//   a.b ~= aH.bH + aH.bM + aM.bH + aM.bM + aH.bL + aL.bH            (f32 accumulation inside v_mfma_f32_16x16x32_bf16)
// costs 6 bf16 MFMAs of K = 32 (~17 cycles each) instead of 8 f32 MFMAs of K = 4 (32 cycles each): 2.5x the matrix-pipe rate at f32
// accuracy.  The price is the VALU work of splitting the operands, so it is used where one operand (the weights) is split once per
// phase and held in registers, and the other costs ~44 VALU per 8 values, issued in the MFMAs' shadow.
// A wave-uniform global pointer made opaque to the optimiser (so that the addresses computed from it are not hoisted out of the
// enclosing loop as VGPR-pair invariants and spilled) and re-marked as global memory (so that the loads stay global_load, not flat).
__device__ __forceinline__ const u32x4* opaque_global(const u32x4* p) {
    const __attribute__((address_space(1))) u32x4* g = (const __attribute__((address_space(1))) u32x4*)p;
    asm volatile("" : "+s"(g));
    return (const u32x4*)g;
}

struct Bf16x3 { u32x4 h, m, l; };       // 8 values: pieces packed two per dword (element e in the low half of dword e/2)

__device__ __forceinline__ Bf16x3 split_bf16x3(const f32x4& x0, const f32x4& x1) {
    Bf16x3 o;
    const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const u32 a0 = __float_as_uint(v[e]), a1 = __float_as_uint(v[e + 1]);
        o.h[e >> 1] = __builtin_amdgcn_perm(a1, a0, 0x07060302u);                   // {a1.hi16, a0.hi16}
        const float r0 = v[e] - __uint_as_float(a0 & 0xffff0000u), r1 = v[e + 1] - __uint_as_float(a1 & 0xffff0000u);     // exact
        const u32 b0 = __float_as_uint(r0), b1 = __float_as_uint(r1);
        o.m[e >> 1] = __builtin_amdgcn_perm(b1, b0, 0x07060302u);
        const float s0 = r0 - __uint_as_float(b0 & 0xffff0000u), s1 = r1 - __uint_as_float(b1 & 0xffff0000u);           // exact
        o.l[e >> 1] = __builtin_amdgcn_perm(__float_as_uint(s1), __float_as_uint(s0), 0x07060302u);
    }
    return o;
}

#define MFMA_BF16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, (a)), __builtin_bit_cast(bf16x8, (b)), (c), 0, 0, 0)

// acc0 takes the three largest piece products, acc1 the three small ones (two independent accumulator chains; summed by the caller)
__device__ __forceinline__ void mma_bf16x6(const Bf16x3& a, const Bf16x3& b, f32x4& acc0, f32x4& acc1) {
    acc0 = MFMA_BF16(a.h, b.h, acc0);
    acc1 = MFMA_BF16(a.m, b.m, acc1);
    acc0 = MFMA_BF16(a.h, b.m, acc0);
    acc1 = MFMA_BF16(a.h, b.l, acc1);
    acc0 = MFMA_BF16(a.m, b.h, acc0);
    acc1 = MFMA_BF16(a.l, b.h, acc1);
}

#endif  // __HIPCC__

// One launch serves up to FWD_MAX_JOBS independent forwards ("jobs": e.g. Q_target(s1), Q_online(s1) and the training forward
// on s0 of one DQN update, plus the acting forward): their serial phases (staging, epilogues, head layers) overlap in one grid.
struct ConvJob {
    const float* params;
    const u32x4* packed;               // bf16 pieces of the conv2 / conv3 kernels (PK_* below)
    const u8* obs;
    const int32_t* index;
    int index_off, index_mod, batch;
    float* act_out[3];                 // global NHWC [batch*oh*ow, cout]; [2] always written
    int write_all;                     // training: write every layer
    int wg0;                           // first workgroup of this job
};

// ---- packed weights (fused.hip: pack_weights_kernel) --------------------------------------------------------------------------
// The bf16x6 contractions read their weight operand as ready-made bf16 pieces in MFMA B-operand order: one "block" = 64 lanes x
// 3 pieces x 16 bytes (lane (kb, j) holds the 8 reduction indices 8kb .. 8kb+7 of its column).  Sections, in u32x4 units:
//   PK_CONV2_FWD  [8 k-blocks][2 column tiles]      B(k = 32 blk + 8kb + e, col = 2j + t)            = W2[k][col]
//   PK_CONV3_FWD  [4][2]                            same for conv3
//   PK_CONV3_DG   [4 taps][2]                       B(n = 8kb + e, c = 16t + j)                      = W3[tap][c][n]
//   PK_CONV2_DG   [2 channel halves][4 taps][2]     B(n = 8kb + e, c = 32 half + 16t + j)            = W2[tap][c][n]
//   PK_CONV1      [3 k-blocks][4 column tiles]      B(k = 32 blk + 8kb + e, col = 4j + t)            = W1[k][col] (conv_pipe.hip)
#define PK_BLOCK 192                  // u32x4 per block (3 pieces x 64 lanes)
#define PK_CONV2_FWD 0
#define PK_CONV3_FWD (PK_CONV2_FWD + 16 * PK_BLOCK)
#define PK_CONV3_DG (PK_CONV3_FWD + 8 * PK_BLOCK)
#define PK_CONV2_DG (PK_CONV3_DG + 8 * PK_BLOCK)
#define PK_CONV1 (PK_CONV2_DG + 16 * PK_BLOCK)      // [3 k-blocks][4 column tiles]: B(k = 32 blk + 8kb + e, col = 4j + t) = W1[k][col], 0 past K1
#define PK_TOTAL_BLOCKS 60
#define PK_TOTAL_U32X4 (PK_TOTAL_BLOCKS * PK_BLOCK)     // the conv sections; then PK_DENSE1 [K1/32 k-blocks][32 column tiles];
// then (f32, for the backward's data gradients) W1T [512][K1] and W2T [N2][512], each section 16-byte aligned:
//   B(k = 32 blk + 8kb + e, col = 64 (ct>>2) + 4j + (ct&3)) = W1[k][col]
size_t fused_packed_u32x4(const dq_qnet* Q);
dq_status fused_pack_weights(const dq_qnet* Q, const float* params_dev, void* packed_dev, hipStream_t st);

// fused.hip: LDS-resident forward (conv chain + dense chain); returns false when the configuration is not covered
bool fused_forward_supported(const dq_qnet* Q);
dq_status fused_forward_multi(dq_qnet* Q, int n_jobs, const dq_qnet_job* jobs, hipStream_t st);
// conv_pipe.hip: the convolutional forward chain as a persistent wave pipeline (jobs[i].wg0 is filled in by the launcher)
bool conv_pipe_supported(const dq_qnet* Q);
dq_status conv_pipe_launch(dq_qnet* Q, int n_jobs, const ConvJob* jobs, int n_cu, hipStream_t st);
// qnet.hip: per-layer backward pieces (also used by the fused backward for layers it does not cover)
dq_status layer_wgrad(dq_qnet* Q, int layer, float* grads_dev, hipStream_t st);
dq_status layer_dgrad(dq_qnet* Q, const float* params_dev, int layer, hipStream_t st);
// fused_bwd.hip: fused backward (data-gradient chains + all-layer weight gradients) for the same configurations
bool fused_backward_supported(const dq_qnet* Q);
size_t fused_backward_workspace_floats(const dq_qnet* Q);
size_t fused_packed_w1t_u32x4(const dq_qnet* Q);           // u32x4 offset of W1T inside a packed buffer
size_t fused_packed_w2t_u32x4(const dq_qnet* Q);           // ... of W2T
// opt != NULL (phases == 3 only): the final reduction also applies the Adam update to p/m/v (one launch fewer per update)
struct AdamOpt { float* p; float* m; float* v; float lr_t, b1, b2, eps; };
// td != NULL: the TD step (dq_td_update's arithmetic) runs in the dense backward's prologue instead of reading dq_dev, and the episode
// bookkeeping of the step just taken (st_n > 0) rides on the same launch
struct TdFused {
    const float *q1o, *q1t, *q0, *reward;
    const u8* terminal;
    const int32_t *action, *index;
    float gamma, grad_scale;
    float *y_out, *dq_out, *metrics;    // nullable
    int metric_slots;                   // partial slots dq_td_metrics will read for this batch
    const u8 *st_done, *st_was_reset;
    const u32* st_lifetime;
    const float* st_reward;
    int st_n;
    unsigned long long* st_stats;
};
// rider != NULL (needs td): the lattices' environment step (env_dev.h parameters, filled by env_fill_act_step) runs as extra
// workgroups of the dense backward's first launch
struct EnvParams;
dq_status fused_backward(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, int phases, hipStream_t st,
                         const AdamOpt* opt = nullptr, const TdFused* td = nullptr, const EnvParams* rider = nullptr, size_t rider_lds = 0);
