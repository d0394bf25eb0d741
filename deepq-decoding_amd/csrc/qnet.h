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
    float* fpartial;             // fused backward workspace (fused_backward_workspace_floats)
    int use_fused;               // 1: fused LDS-resident forward when the configuration allows it
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

// fused.hip: LDS-resident forward (conv chain + dense chain); returns false when the configuration is not covered
bool fused_forward_supported(const dq_qnet* Q);
dq_status fused_forward_multi(dq_qnet* Q, int n_jobs, const dq_qnet_job* jobs, hipStream_t st);
// qnet.hip: per-layer backward pieces (also used by the fused backward for layers it does not cover)
dq_status layer_wgrad(dq_qnet* Q, int layer, float* grads_dev, hipStream_t st);
dq_status layer_dgrad(dq_qnet* Q, const float* params_dev, int layer, hipStream_t st);
// fused_bwd.hip: fused backward (data-gradient chains + all-layer weight gradients) for the same configurations
bool fused_backward_supported(const dq_qnet* Q);
size_t fused_backward_workspace_floats(const dq_qnet* Q);
dq_status fused_backward(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, int phases, hipStream_t st);
