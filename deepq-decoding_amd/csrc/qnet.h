// Q-network handle shared by qnet.hip (per-layer implicit GEMMs, backward) and fused.hip (fused forward chains).
#pragma once
#include "common.h"

#define QN_MAX_LAYERS 12

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
    float* grad[2];              // ping-pong gradient buffers (max activation size)
    float* partial;              // wgrad slices
    size_t partial_floats;
    int last_train_batch;
    const uint8_t* last_obs;     // inputs of the last training forward (needed by conv1's weight gradient)
    const int32_t* last_index;
    int last_index_off, last_index_mod;
    int use_fused;               // 1: fused LDS-resident forward when the configuration allows it
};


// fused.hip: LDS-resident forward (conv chain + dense chain); returns false when the configuration is not covered
bool fused_forward_supported(const dq_qnet* Q);
dq_status fused_forward(dq_qnet* Q, const float* params_dev, const uint8_t* obs_dev, const int32_t* index_dev, int index_off,
                        int index_mod, int batch, int training, const uint32_t seed[2], uint64_t t, uint32_t sample_base,
                        float* q_dev, hipStream_t st);
