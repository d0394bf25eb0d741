// Fused backward of the convolutional Q-network (the backward half of keras-rl's trainable_model.train_on_batch,
// /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:119-130), for the configurations fused.hip covers.
//
//   dense_bwd_chain_kernel   workgroup = 8 waves = 16 samples: dueling backward, then the data gradients of the three dense
//                            layers chained through LDS -- g3 -> gY2 = g3 W3^T -> gH1 = (gY2 W2^T) * [h1 > 0] * 1/(1-rate)
//                            -> gX = (gH1 W1^T) * [x > 0], un-flattened to NHWC.  Every gradient is also written to HBM
//                            (the weight gradients read them from there).  Operand tricks as in fused.hip: the reduction
//                            index of an MFMA group is permuted so that A (LDS) and, for W1, B (global: rows of W1 ARE the
//                            transposed operand's columns) are read as float4.
// Gradients w.r.t. pre-activations are stored per layer in Q->gz[layer]; weight gradients and the convolutional data gradients
// still run through qnet.hip's per-layer kernels (layer_wgrad / layer_dgrad).
#include "qnet.h"

struct DenseBwdArgs {
    const float* params;
    const float* dq;                    // [batch, n_actions]
    const float* h1;                    // saved hidden output (post ReLU + dropout): mask of gH1
    const float* x;                     // saved last-convolution output [batch, K1] (NHWC): mask of gX
    int batch, K1, perm_hw, perm_c;
    int N2, N3, n_actions;
    int w_off[3];
    float mask_scale;                   // 1/(1-rate) of the hidden layer's dropout (1 if none)
    float* g3;                          // [batch, N3] (NULL without a dueling layer)
    float* gy2;                         // [batch, N2]
    float* gh1;                         // [batch, 512]
    float* gx;                          // [batch, K1] NHWC
    int ldg;                            // LDS row stride of the g3 / gY2 images (floats)
    int off_g3, off_gy2, off_gh1;
};

template <int NT2>                      // N2 <= 16*NT2 and N3 <= 16*NT2
__global__ __launch_bounds__(DENSE_THREADS, 2) void dense_bwd_chain_kernel(DenseBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    float* s_g3 = reinterpret_cast<float*>(smem + a.off_g3);
    float* s_gy2 = reinterpret_cast<float*>(smem + a.off_gy2);
    float* s_gh1 = reinterpret_cast<float*>(smem + a.off_gh1);
    constexpr int LDH = DENSE_HID + 4;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, j = lane & 15, kq = lane >> 4;
    const int b0 = blockIdx.x * DENSE_ROWS;
    const int ns = min(DENSE_ROWS, a.batch - b0);
    const int A = a.n_actions, N2 = a.N2, N3 = a.N3, ldg = a.ldg;

    for (int i = tid; i < DENSE_ROWS * ldg; i += DENSE_THREADS) { s_g3[i] = 0.f; s_gy2[i] = 0.f; }
    __syncthreads();
    // ---- dueling backward: g3[b,0] = sum_a dq[b,a];  g3[b,1+a] = dq[b,a] - (1/A) sum_a' dq[b,a'] ----------------------
    for (int row = wave; row < ns; row += DENSE_WAVES) {
        const float* dr = a.dq + (size_t)(b0 + row) * A;
        if (N3 > 0) {
            float s = 0.f;
            for (int c = lane; c < A; c += 64) s += dr[c];
            for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
            float* o = a.g3 + (size_t)(b0 + row) * N3;
            if (lane == 0) { s_g3[row * ldg] = s; o[0] = s; }
            for (int c = lane; c < A; c += 64) {
                const float v = dr[c] - s / (float)A;
                s_g3[row * ldg + 1 + c] = v;
                o[1 + c] = v;
            }
        } else {
            for (int c = lane; c < A; c += 64) {
                const float v = dr[c];
                s_gy2[row * ldg + c] = v;
                a.gy2[(size_t)(b0 + row) * N2 + c] = v;
            }
        }
    }
    __syncthreads();
    // ---- gY2 = g3 W3^T  (K = N3, one column tile per wave) -------------------------------------------------------------
    if (N3 > 0) {
        if (wave < NT2) {
            const float* w3 = a.params + a.w_off[2];
            const int n2 = 16 * wave + j;
            float b[NT2][4];
#pragma unroll
            for (int g = 0; g < NT2; ++g)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int k3 = 16 * g + 4 * kq + s;
                    b[g][s] = (n2 < N2 && k3 < N3) ? w3[(size_t)n2 * N3 + k3] : 0.f;
                }
            f32x4 acc = {0.f, 0.f, 0.f, 0.f};
            const float* grow = s_g3 + j * ldg + 4 * kq;
#pragma unroll
            for (int g = 0; g < NT2; ++g) {
                const f32x4 av = *reinterpret_cast<const f32x4*>(grow + 16 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = MFMA16(av[s], b[g][s], acc);
            }
            if (n2 < N2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kq + r;
                    s_gy2[row * ldg + n2] = acc[r];
                    if (row < ns) a.gy2[(size_t)(b0 + row) * N2 + n2] = acc[r];
                }
            }
        }
        __syncthreads();
    }
    // ---- gH1 = (gY2 W2^T) * [h1 > 0] * scale  (K = N2; wave w owns columns 64w + 4j + t) ----------------------------------
    {
        const float* w2 = a.params + a.w_off[1];
        const int c0 = 64 * wave + 4 * j;
        float b[NT2][4][4];
#pragma unroll
        for (int g = 0; g < NT2; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int k2 = 16 * g + 4 * kq + s;
#pragma unroll
                for (int t = 0; t < 4; ++t) b[g][s][t] = k2 < N2 ? w2[(size_t)(c0 + t) * N2 + k2] : 0.f;      // B(n2, n1) = W2[n1][n2]
            }
        f32x4 hv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            hv[r] = row < ns ? *reinterpret_cast<const f32x4*>(a.h1 + (size_t)(b0 + row) * DENSE_HID + c0) : f32x4{0.f, 0.f, 0.f, 0.f};
        }
        f32x4 acc[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = f32x4{0.f, 0.f, 0.f, 0.f};
        const float* grow = s_gy2 + j * ldg + 4 * kq;
#pragma unroll
        for (int g = 0; g < NT2; ++g) {
            const f32x4 av = *reinterpret_cast<const f32x4*>(grow + 16 * g);
#pragma unroll
            for (int s = 0; s < 4; ++s)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = MFMA16(av[s], b[g][s][t], acc[t]);
        }
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int row = 4 * kq + r;
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = hv[r][t] > 0.f ? acc[t][r] * a.mask_scale : 0.f;
            *reinterpret_cast<f32x4*>(s_gh1 + row * LDH + c0) = v;
            if (row < ns) *reinterpret_cast<f32x4*>(a.gh1 + (size_t)(b0 + row) * DENSE_HID + c0) = v;
        }
    }
    __syncthreads();
    // ---- gX = (gH1 W1^T) * [x > 0]  (K = 512): column k = 16T + j of tile T is ROW k of W1 -> float4 along the reduction ----
    {
        const float* w1 = a.params + a.w_off[0];
        const int tiles = a.K1 >> 4;
        const float* hrow = s_gh1 + j * LDH + 4 * kq;
        for (int T0 = wave; T0 < tiles; T0 += 3 * DENSE_WAVES) {
            bool ok[3];
            const float* wr[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const int T = T0 + u * DENSE_WAVES;
                ok[u] = T < tiles;                                  // wave-uniform
                wr[u] = w1 + (size_t)(16 * (ok[u] ? T : T0) + j) * DENSE_HID + 4 * kq;
            }
            f32x4 acc[3], bA[3], bB[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                acc[u] = f32x4{0.f, 0.f, 0.f, 0.f};
                bA[u] = *reinterpret_cast<const f32x4*>(wr[u]);
            }
            for (int g = 0; g < DENSE_HID / 16; g += 2) {
#pragma unroll
                for (int u = 0; u < 3; ++u) bB[u] = *reinterpret_cast<const f32x4*>(wr[u] + 16 * (g + 1));
                f32x4 av = *reinterpret_cast<const f32x4*>(hrow + 16 * g);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < 3; ++u)
                        if (ok[u]) acc[u] = MFMA16(av[s], bA[u][s], acc[u]);
                if (g + 2 < DENSE_HID / 16) {
#pragma unroll
                    for (int u = 0; u < 3; ++u) bA[u] = *reinterpret_cast<const f32x4*>(wr[u] + 16 * (g + 2));
                }
                av = *reinterpret_cast<const f32x4*>(hrow + 16 * (g + 1));
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int u = 0; u < 3; ++u)
                        if (ok[u]) acc[u] = MFMA16(av[s], bB[u][s], acc[u]);
            }
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                if (!ok[u]) continue;
                const int k = 16 * (T0 + u * DENSE_WAVES) + j;      // Keras Flatten index c*hw + p  ->  NHWC offset p*C + c
                int idx = k;
                if (a.perm_hw > 0) { const int c = k / a.perm_hw, p = k - c * a.perm_hw; idx = p * a.perm_c + c; }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 4 * kq + r;
                    if (row >= ns) continue;
                    const size_t o = (size_t)(b0 + row) * a.K1 + idx;
                    a.gx[o] = a.x[o] > 0.f ? acc[u][r] : 0.f;
                }
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------------------
struct DenseBwdPlan { int NT2, ldg, off_g3, off_gy2, off_gh1; size_t lds; };

static bool plan_dense_bwd(const dq_qnet* Q, DenseBwdPlan* P) {
    const int nc = Q->cfg.n_conv;
    if (Q->cfg.n_ff != 1) return false;
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    if (D1.nout != DENSE_HID || (D1.nin & 15)) return false;
    const int N2 = D2.nout, N3 = Q->cfg.dueling ? Q->L[nc + 2].nout : 0;
    const int widest = N3 > N2 ? N3 : N2;
    if (widest > 112) return false;
    P->NT2 = widest <= 64 ? 4 : 7;
    P->ldg = 16 * P->NT2 + 4;
    size_t off = 0;
    P->off_g3 = (int)off; off += up16((size_t)DENSE_ROWS * P->ldg * 4);
    P->off_gy2 = (int)off; off += up16((size_t)DENSE_ROWS * P->ldg * 4);
    P->off_gh1 = (int)off; off += up16((size_t)DENSE_ROWS * (DENSE_HID + 4) * 4);
    P->lds = off;
    return true;
}

bool fused_backward_supported(const dq_qnet* Q) {
    DenseBwdPlan dp;
    return fused_forward_supported(Q) && plan_dense_bwd(Q, &dp);
}

dq_status fused_backward(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, hipStream_t st) {
    DenseBwdPlan dp;
    DQ_REQUIRE(plan_dense_bwd(Q, &dp), DQ_ERR_UNSUPPORTED, "fused_backward: configuration not covered");
    DQ_REQUIRE((reinterpret_cast<uintptr_t>(params_dev) & 15) == 0, DQ_ERR_INVALID, "fused_backward: params_dev must be 16-byte aligned");
    const int B = Q->last_train_batch, nc = Q->cfg.n_conv, nl = Q->n_layers;
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    DenseBwdArgs da;
    memset(&da, 0, sizeof(da));
    da.params = params_dev; da.dq = dq_dev; da.h1 = Q->act[0][nc]; da.x = Q->act[0][nc - 1];
    da.batch = B; da.K1 = D1.nin; da.perm_hw = Q->flat_hw; da.perm_c = Q->flat_c;
    da.N2 = D2.nout; da.N3 = Q->cfg.dueling ? Q->L[nc + 2].nout : 0; da.n_actions = Q->cfg.n_actions;
    for (int l = 0; l < nl - nc; ++l) da.w_off[l] = (int)Q->L[nc + l].w_off;
    da.mask_scale = D1.dropout > 0.f ? (float)(1.0 / (1.0 - (double)D1.dropout)) : 1.f;
    da.g3 = Q->cfg.dueling ? Q->gz[nc + 2] : nullptr; da.gy2 = Q->gz[nc + 1]; da.gh1 = Q->gz[nc]; da.gx = Q->gz[nc - 1];
    da.ldg = dp.ldg; da.off_g3 = dp.off_g3; da.off_gy2 = dp.off_gy2; da.off_gh1 = dp.off_gh1;
    if (dp.NT2 == 4) dense_bwd_chain_kernel<4><<<(B + DENSE_ROWS - 1) / DENSE_ROWS, DENSE_THREADS, dp.lds, st>>>(da);
    else dense_bwd_chain_kernel<7><<<(B + DENSE_ROWS - 1) / DENSE_ROWS, DENSE_THREADS, dp.lds, st>>>(da);
    DQ_LAUNCH_CHECK();
    for (int i = nl - 1; i >= 0; --i) {
        dq_status rc = layer_wgrad(Q, i, grads_dev, st);
        if (rc != DQ_OK) return rc;
        if (i > 0 && i < nc) {                                      // convolutional data gradients: per-layer kernels
            rc = layer_dgrad(Q, params_dev, i, st);
            if (rc != DQ_OK) return rc;
        }
    }
    return DQ_OK;
}
