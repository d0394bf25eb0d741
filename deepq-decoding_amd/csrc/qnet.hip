// Convolutional dueling Q-network: forward, backward and parameter bookkeeping on gfx950.
//
// Replaces the Keras model built by build_convolutional_nn
// (/root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:61-90 == Function_Library.py:338-377
// of the notebook copy) wrapped by keras-rl's dueling head (DQNAgent(enable_dueling_network=True), :119-127),
// i.e. model.predict_on_batch / trainable_model.train_on_batch of the un-vendored keras-rl fork.
// All contractions run on v_mfma_f32_32x32x2_f32 through the implicit-GEMM pieces of gemm.h.
//
// Device layouts
//   parameters  one flat float buffer, Keras order and Keras shapes (conv kernels HWIO, dense (in,out)):
//               [k1 b1 k2 b2 ... ]  -- the shipped .h5f tensors drop in without transposition;
//   activations NHWC (row m = (sample, oy, ox), col = channel) == the row-major GEMM output; the Keras
//               channels_first Flatten order is applied inside the first dense layer's loaders;
//   observation uint8 NCHW as the environment writes it, optionally gathered through an index vector
//               (replay minibatch rows) inside the first convolution's loader -- no minibatch copy.
#include "gemm.h"
#include <new>

#define BM 128
#define BK 32

// C = A_gather * B (+ epilogue).  Block = WAVES waves; wave w owns rows [32w, 32w+32) x BN columns.
// WAVES = 4 for the big im2col GEMMs, WAVES = 1 when M is small (dense layers) so that the grid still fills the chip.
template <int BN, int WAVES>
__global__ __launch_bounds__(64 * WAVES) void gemm_fwd_kernel(Gather ga, BMap gb, Epilogue ep, int M, int N, int K) {
    constexpr int TBM = 32 * WAVES, NTHR = 64 * WAVES;
    __shared__ float sA[TBM][BK + 1];
    __shared__ float sB[BK][BN];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.x * TBM, n0 = blockIdx.y * BN;
    constexpr int A_PER = TBM * BK / NTHR, A_STEP = NTHR / 32, B_PER = BK * BN / NTHR, B_STEP = NTHR / BN, NT = BN / 32;

    // this thread stages A rows (tid>>5) + A_STEP*i at column tid&31 and B rows (tid / BN) + B_STEP*i at column tid % BN
    RowRef rows[A_PER];
    {
        RowIter it;
        it.init(ga, m0 + (tid >> 5));
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            rows[i] = it.ref(ga, m0 + (tid >> 5) + A_STEP * i < M);
            it.advance(ga, A_STEP);
        }
    }
    const int acol = tid & 31, bcol = tid % BN, brow0 = tid / BN;

    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;

    float ra[A_PER], rb[B_PER];
    auto fetch = [&](int k0) {
        const ColRef c = gather_col(ga, k0 + acol, K);
#pragma unroll
        for (int i = 0; i < A_PER; ++i) ra[i] = gather_load(ga, rows[i], c);
#pragma unroll
        for (int i = 0; i < B_PER; ++i) rb[i] = bmap_load(gb, k0 + brow0 + B_STEP * i, n0 + bcol, K, N);
    };
    fetch(0);
    for (int k0 = 0; k0 < K; k0 += BK) {
        __syncthreads();                       // previous tile fully consumed
#pragma unroll
        for (int i = 0; i < A_PER; ++i) sA[(tid >> 5) + A_STEP * i][acol] = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) sB[brow0 + B_STEP * i][bcol] = rb[i];
        __syncthreads();
        if (k0 + BK < K) fetch(k0 + BK);       // next tile's global loads fly under the MFMAs
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = sA[wave * 32 + (lane & 31)][kk + (lane >> 5)];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float b = sB[kk + (lane >> 5)][t * 32 + (lane & 31)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    // C/D layout of 32x32x2: col = lane & 31, row = (r & 3) + 8 (r >> 2) + 4 (lane >> 5)
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = m0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (m < M && n < N) epilogue_store(ep, m, n, acc[t][r]);
        }
    }
}

// Weight gradient: dW[k, n] = sum_m A(m, k) * dZ[m, n]  (dZ row-major [M, N]) for m in this block's slice;
// partial[slice][k][n] (+ partial bias sums when kt == 0) -- reduced in fixed order by reduce_partials_kernel.
// Block = 4 waves; wave w owns output rows k in [k0 + 32w, k0 + 32w + 32) x BN columns.
template <int BN>
__global__ __launch_bounds__(256) void gemm_wgrad_kernel(Gather ga, const float* __restrict__ dz, float* __restrict__ partial,
                                                         float* __restrict__ partial_bias, int M, int N, int K, int rows_per_slice,
                                                         size_t pstride) {
    __shared__ float sA[BK][BM + 1];           // [m][k]: read transposed by the MFMA A operand
    __shared__ float sB[BK][BN];
    __shared__ float sBias[256];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int k0 = blockIdx.x * BM, slice = blockIdx.y, n0 = blockIdx.z * BN;
    const int m_begin = slice * rows_per_slice, m_end = min(M, m_begin + rows_per_slice);
    constexpr int A_PER = BM * BK / 256, B_PER = BK * BN / 256, NT = BN / 32;

    // A staging: this thread owns column (k) tid & 127 and rows (m) (tid >> 7) + 2i
    const ColRef col = gather_col(ga, k0 + (tid & 127), K);
    const int bcol = tid % BN, brow0 = tid / BN;
    f32x16 acc[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.f;
    float bias_acc = 0.f;

    float ra[A_PER], rb[B_PER];
    auto fetch = [&](int mb) {
        RowIter it;
        it.init(ga, mb + (tid >> 7));
#pragma unroll
        for (int i = 0; i < A_PER; ++i) {
            ra[i] = gather_load(ga, it.ref(ga, mb + (tid >> 7) + 2 * i < m_end), col);
            it.advance(ga, 2);
        }
#pragma unroll
        for (int i = 0; i < B_PER; ++i) {
            const int m = mb + brow0 + (256 / BN) * i, n = n0 + bcol;
            rb[i] = (m < m_end && n < N) ? dz[(size_t)m * N + n] : 0.f;
        }
    };
    fetch(m_begin);
    for (int mb = m_begin; mb < m_end; mb += BK) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < A_PER; ++i) sA[(tid >> 7) + 2 * i][tid & 127] = ra[i];
#pragma unroll
        for (int i = 0; i < B_PER; ++i) { sB[brow0 + (256 / BN) * i][bcol] = rb[i]; bias_acc += rb[i]; }
        __syncthreads();
        if (mb + BK < m_end) fetch(mb + BK);
#pragma unroll
        for (int kk = 0; kk < BK; kk += 2) {
            const float a = sA[kk + (lane >> 5)][wave * 32 + (lane & 31)];
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const float b = sB[kk + (lane >> 5)][t * 32 + (lane & 31)];
                acc[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[t], 0, 0, 0);
            }
        }
    }
    float* out = partial + (size_t)slice * pstride;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int n = n0 + t * 32 + (lane & 31);
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int k = k0 + wave * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
            if (k < K && n < N) out[(size_t)k * N + n] = acc[t][r];
        }
    }
    if (blockIdx.x == 0) {                     // bias gradient = column sums of dZ over this slice
        sBias[tid] = bias_acc;
        __syncthreads();
        if (tid < BN) {
            float s = 0.f;
            for (int r = 0; r < 256 / BN; ++r) s += sBias[r * BN + tid];
            if (n0 + tid < N) partial_bias[(size_t)slice * pstride + n0 + tid] = s;
        }
    }
}

// out[i] = sum_s partial[s * stride + i]; the slices are split over 4 thread groups whose sums are combined in a
// fixed order => deterministic.  block = (64, 4).
__global__ __launch_bounds__(256) void reduce_partials_kernel(const float* __restrict__ partial, float* __restrict__ out, int n,
                                                              int slices, size_t stride) {
    __shared__ float sh[4][64];
    const int i = blockIdx.x * 64 + threadIdx.x, g = threadIdx.y;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;
    if (i < n) {
        int k = g;
        for (; k + 12 < slices; k += 16) {
            s0 += partial[(size_t)k * stride + i];
            s1 += partial[(size_t)(k + 4) * stride + i];
            s2 += partial[(size_t)(k + 8) * stride + i];
            s3 += partial[(size_t)(k + 12) * stride + i];
        }
        for (; k < slices; k += 4) s0 += partial[(size_t)k * stride + i];
    }
    sh[g][threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < n) out[i] = (sh[0][threadIdx.x] + sh[1][threadIdx.x]) + (sh[2][threadIdx.x] + sh[3][threadIdx.x]);
}

// Dueling head (keras-rl dueling_type 'avg'): Q[b,a] = y[b,0] + y[b,1+a] - mean_a' y[b,1+a']
__global__ void dueling_fwd_kernel(const float* __restrict__ y, float* __restrict__ q, int B, int A) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* row = y + (size_t)b * (A + 1);
    float s = 0.f;
    for (int a = lane; a < A; a += 64) s += row[1 + a];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    const float base = row[0] - s / (float)A;
    for (int a = lane; a < A; a += 64) q[(size_t)b * A + a] = base + row[1 + a];
}

// g[b,0] = sum_a dq[b,a];  g[b,1+a] = dq[b,a] - (1/A) sum_a' dq[b,a']
__global__ void dueling_bwd_kernel(const float* __restrict__ dq, float* __restrict__ g, int B, int A) {
    const int b = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (b >= B) return;
    const float* row = dq + (size_t)b * A;
    float s = 0.f;
    for (int a = lane; a < A; a += 64) s += row[a];
    for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
    float* o = g + (size_t)b * (A + 1);
    if (lane == 0) o[0] = s;
    for (int a = lane; a < A; a += 64) o[1 + a] = row[a] - s / (float)A;
}

// ---------------------------------------------------------------------------------------------------
#include "qnet.h"
#include "env_dev.h"

static void launch_fwd(const Gather& ga, const BMap& gb, const Epilogue& ep, int M, int N, int K, hipStream_t st) {
    const bool small = (M + BM - 1) / BM * ((N + 63) / 64) < 256;      // too few 128-row blocks to fill 256 CUs
    dq_prof_begin(DQ_K_GEMM_FWD, st);
    if (N <= 32) {
        if (small) gemm_fwd_kernel<32, 1><<<dim3((M + 31) / 32, 1), 64, 0, st>>>(ga, gb, ep, M, N, K);
        else gemm_fwd_kernel<32, 4><<<dim3((M + BM - 1) / BM, 1), 256, 0, st>>>(ga, gb, ep, M, N, K);
    } else {
        if (small) gemm_fwd_kernel<64, 1><<<dim3((M + 31) / 32, (N + 63) / 64), 64, 0, st>>>(ga, gb, ep, M, N, K);
        else gemm_fwd_kernel<64, 4><<<dim3((M + BM - 1) / BM, (N + 63) / 64), 256, 0, st>>>(ga, gb, ep, M, N, K);
    }
    dq_prof_end(DQ_K_GEMM_FWD, st);
}

// split of the weight-gradient reduction over M into slices (shared by create() for workspace sizing)
static void wgrad_slices(int M, int K, int N, int* rows_per_slice, int* slices) {
    const int tiles = ((K + BM - 1) / BM) * ((N + 63) / 64);
    int want = (768 + tiles - 1) / tiles;                               // aim at >= ~768 blocks
    if (want < 1) want = 1;
    int rows = (M + want - 1) / want;
    rows = (rows + BK - 1) / BK * BK;
    if (rows < 2 * BK) rows = 2 * BK;
    *rows_per_slice = rows;
    *slices = (M + rows - 1) / rows;
}

static Gather dense_gather(const float* src, int K) {
    Gather g;
    memset(&g, 0, sizeof(g));
    g.src = src; g.rows_per_sample = 1; g.RW = 1; g.sb = (unsigned)K;
    g.KC = K > 0 ? K : 1; g.KW = 1 << 30; g.sc = 1;
    return g;
}

static Epilogue plain_epilogue(float* out, int ldo) {
    Epilogue e;
    memset(&e, 0, sizeof(e));
    e.out = out; e.ldo = ldo; e.PC = 1 << 30; e.s_lo = 1;
    return e;
}

// Weight + bias gradient of layer i: dW = A^T g with g = Q->gz[i], A = the layer's (gathered) input of the last training forward.
dq_status layer_wgrad(dq_qnet* Q, int i, float* grads_dev, hipStream_t st) {
    const Layer& L = Q->L[i];
    const int B = Q->last_train_batch, M = B * L.rows;
    const float* g = Q->gz[i];
    Gather ga;
    memset(&ga, 0, sizeof(ga));
    const float* x = i > 0 ? Q->act[0][i - 1] : nullptr;
    if (L.kind == 0) {
        ga.rows_per_sample = L.rows; ga.RW = L.ow; ga.KC = L.cin; ga.KW = L.k;
        if (i == 0) {
            ga.src = Q->last_obs; ga.is_u8 = 1;
            ga.sb = (unsigned)(L.cin * L.ih * L.iw); ga.sy = (unsigned)(L.s * L.iw); ga.sx = (unsigned)L.s;
            ga.sky = L.iw; ga.skx = 1; ga.sc = L.ih * L.iw;
            ga.index = Q->last_index; ga.index_off = Q->last_index_off; ga.index_mod = Q->last_index_mod > 0 ? Q->last_index_mod : 0x7fffffff;
        } else {
            ga.src = x;
            ga.sb = (unsigned)(L.ih * L.iw * L.cin); ga.sy = (unsigned)(L.s * L.iw * L.cin); ga.sx = (unsigned)(L.s * L.cin);
            ga.sky = L.iw * L.cin; ga.skx = L.cin; ga.sc = 1;
        }
    } else {
        ga = dense_gather(x, L.K);
        if (i == Q->cfg.n_conv) { ga.KC = Q->flat_hw; ga.KW = 1 << 30; ga.skx = 1; ga.sc = Q->flat_c; }
    }
    int rows_per_slice, slices;
    wgrad_slices(M, L.K, L.N, &rows_per_slice, &slices);
    const size_t pstride = (size_t)L.K * L.N + L.N;                 // per slice: kernel partial then bias partial
    DQ_REQUIRE((size_t)slices * pstride <= Q->partial_floats, DQ_ERR_STATE, "dq_qnet_backward: workspace too small");
    float* pw = Q->partial;
    float* pb = Q->partial + (size_t)L.K * L.N;
    dq_prof_begin(DQ_K_GEMM_WGRAD, st);
    if (L.N <= 32) {
        dim3 grid((L.K + BM - 1) / BM, slices, 1);
        gemm_wgrad_kernel<32><<<grid, 256, 0, st>>>(ga, g, pw, pb, M, L.N, L.K, rows_per_slice, pstride);
    } else {
        dim3 grid((L.K + BM - 1) / BM, slices, (L.N + 63) / 64);
        gemm_wgrad_kernel<64><<<grid, 256, 0, st>>>(ga, g, pw, pb, M, L.N, L.K, rows_per_slice, pstride);
    }
    dq_prof_end(DQ_K_GEMM_WGRAD, st);
    DQ_LAUNCH_CHECK();
    // bias follows the kernel in the flat buffer (b_off == w_off + K*N): one reduction covers both
    const int nw = L.K * L.N + L.N;
    reduce_partials_kernel<<<(nw + 63) / 64, dim3(64, 4), 0, st>>>(Q->partial, grads_dev + L.w_off, nw, slices, pstride);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

// Data gradient through layer i (i >= 1), masked by the previous layer's activation:
// Q->gz[i-1] = (Q->gz[i] W_i^T) * [y_{i-1} > 0] * scale.
dq_status layer_dgrad(dq_qnet* Q, const float* params_dev, int i, hipStream_t st) {
    const Layer& L = Q->L[i];
    const Layer& Pv = Q->L[i - 1];
    const int B = Q->last_train_batch;
    const float* g = Q->gz[i];
    Gather gd;
    memset(&gd, 0, sizeof(gd));
    BMap gb;
    int Md, Nd, Kd;
    Epilogue ep = plain_epilogue(Q->gz[i - 1], 0);
    if (L.kind == 1) {
        gd = dense_gather(g, L.N);                       // A = g [B, N]
        gb = {params_dev + L.w_off, 1 << 30, 0, 1, L.N}; // B(n, k) = W[k*N + n]
        Md = B; Nd = L.K; Kd = L.N;
        ep.ldo = L.K;
        if (i == Q->cfg.n_conv) { ep.PC = Q->flat_hw; ep.s_lo = Q->flat_c; ep.s_hi = 1; }   // Flatten^-1: k = c*HW + p -> p*C + c
    } else {
        // transposed convolution as a gather: dx[b,iy,ix,c] = sum_{ky,kx,n} g[b,iy-ky,ix-kx,n] W[ky,kx,c,n]   (stride 1)
        gd.src = g; gd.rows_per_sample = L.ih * L.iw; gd.RW = L.iw;
        gd.sb = (unsigned)(L.oh * L.ow * L.N); gd.sy = (unsigned)(L.ow * L.N); gd.sx = (unsigned)L.N;
        gd.KC = L.N; gd.KW = L.k; gd.sky = -(L.ow * L.N); gd.skx = -L.N; gd.sc = 1;
        gd.check = 1; gd.ylim = L.oh; gd.xlim = L.ow;
        gb = {params_dev + L.w_off, L.N, L.cin * L.N, 1, L.N};   // B((ky,kx,n), c) = W[((ky*k+kx)*Cin + c)*N + n]
        Md = B * L.ih * L.iw; Nd = L.cin; Kd = L.k * L.k * L.N;
        ep.ldo = L.cin;
    }
    ep.flags = EPI_MASK;
    ep.mask_src = Q->act[0][i - 1];
    ep.mask_scale = Pv.dropout > 0.f ? (float)(1.0 / (1.0 - (double)Pv.dropout)) : 1.f;
    if (!Pv.relu) { ep.flags = EPI_NONE; }
    launch_fwd(gd, gb, ep, Md, Nd, Kd, st);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

extern "C" {

dq_status dq_qnet_create(const dq_qnet_cfg* cfg, dq_qnet** out) {
    DQ_REQUIRE(cfg && out, DQ_ERR_INVALID, "dq_qnet_create: null argument");
    *out = nullptr;
    DQ_REQUIRE(cfg->n_conv >= 1 && cfg->n_conv <= 4 && cfg->n_ff >= 0 && cfg->n_ff <= 4, DQ_ERR_UNSUPPORTED, "dq_qnet_create: 1..4 conv and 0..4 hidden dense layers");
    DQ_REQUIRE(cfg->n_actions >= 1 && cfg->n_actions <= 1024 && cfg->max_batch >= 1, DQ_ERR_INVALID, "dq_qnet_create: bad n_actions / max_batch");
    dq_qnet* Q = new (std::nothrow) dq_qnet();
    DQ_REQUIRE(Q, DQ_ERR_NOMEM, "out of host memory");
    memset(Q, 0, sizeof(*Q));
    Q->cfg = *cfg;
    int c = cfg->in_c, h = cfg->in_h, w = cfg->in_w, n = 0;
    size_t off = 0;
    for (int i = 0; i < cfg->n_conv; ++i) {
        Layer& L = Q->L[n++];
        L.kind = 0; L.cin = c; L.cout = cfg->conv[i][0]; L.k = cfg->conv[i][1]; L.s = cfg->conv[i][2]; L.ih = h; L.iw = w;
        if (L.k < 1 || L.s < 1 || L.k > h || L.k > w || L.cout < 1) { delete Q; dq_set_error("dq_qnet_create: bad conv layer %d", i); return DQ_ERR_INVALID; }
        if (i > 0 && L.s != 1) { delete Q; dq_set_error("dq_qnet_create: stride > 1 is only implemented for the first convolution"); return DQ_ERR_UNSUPPORTED; }
        L.oh = (h - L.k) / L.s + 1; L.ow = (w - L.k) / L.s + 1;
        L.K = L.k * L.k * c; L.N = L.cout; L.rows = L.oh * L.ow; L.relu = 1;
        L.w_off = off; off += (size_t)L.K * L.N; L.b_off = off; off += L.N;
        c = L.cout; h = L.oh; w = L.ow;
    }
    Q->flat_c = c; Q->flat_hw = h * w;
    int nin = c * h * w;
    for (int i = 0; i <= cfg->n_ff + (cfg->dueling ? 1 : 0); ++i) {
        Layer& L = Q->L[n++];
        L.kind = 1; L.nin = nin; L.rows = 1;
        if (i < cfg->n_ff) { L.nout = cfg->ff_units[i]; L.relu = 1; L.dropout = cfg->ff_dropout[i]; }
        else if (i == cfg->n_ff) { L.nout = cfg->n_actions; }
        else { L.nout = cfg->n_actions + 1; }
        if (L.nout < 1 || L.dropout < 0.f || L.dropout >= 1.f) { delete Q; dq_set_error("dq_qnet_create: bad dense layer %d", i); return DQ_ERR_INVALID; }
        L.K = L.nin; L.N = L.nout;
        L.w_off = off; off += (size_t)L.K * L.N; L.b_off = off; off += L.N;
        nin = L.nout;
    }
    Q->n_layers = n;
    Q->n_params = off;
    Q->use_fused = 1;
    {
        const char* cf = getenv("DQ_CONV_FORM");
        const char* bf = getenv("DQ_CONV_BWD_FORM");
        Q->conv_form = cf && cf[0] == 'g' ? 1 : 0;
        Q->conv_bwd_form = bf && bf[0] == '8' ? 1 : bf && bf[0] == '1' ? 2 : 0;
        const char* a1 = getenv("DQ_CONV_BWD_A1");                  // =saved: the round-5 form (a1 through HBM), for A/B runs
        Q->conv_bwd_a1 = a1 && a1[0] == 's' ? 1 : 0;
        Q->last_a1_saved = 1;
        const char* xp = getenv("DQ_X_PLANES");
        Q->x_planes = xp && xp[0] == '1' ? 1 : 0;
        const char* dl = getenv("DQ_DENSE_LEAN");
        Q->dense_lean = dl ? atoi(dl) : 0;
    }
    // workspaces
    size_t max_partial = 0;
    hipError_t e = hipSuccess;
    for (int i = 0; i < n && e == hipSuccess; ++i) {
        const Layer& L = Q->L[i];
        const size_t floats = (size_t)cfg->max_batch * L.rows * L.N;
        if (floats >= (1ull << 32)) { dq_qnet_destroy(Q); dq_set_error("dq_qnet_create: activation too large for 32-bit offsets"); return DQ_ERR_UNSUPPORTED; }
        // (+ 16 bytes: the dense weight gradient reads rows as dwordx4 that may straddle the last row's end, fused_bwd.hip)
        for (int s = 0; s < 2 && e == hipSuccess; ++s) e = hipMalloc(&Q->act[s][i], floats * sizeof(float) + 16);
        if (e == hipSuccess) e = hipMalloc(&Q->gz[i], floats * sizeof(float) + 16);
        int rps, slices;
        wgrad_slices(cfg->max_batch * L.rows, L.K, L.N, &rps, &slices);
        const size_t p = (size_t)slices * ((size_t)L.K * L.N + L.N);
        max_partial = p > max_partial ? p : max_partial;
    }
    if (e == hipSuccess) e = hipMalloc(&Q->partial, max_partial * sizeof(float));
    if (e == hipSuccess) {
        int tab[96 + (CONV_FWD_TABS + 5) * CONV_ROWTAB];
        const Layer& L1 = Q->L[0];
        for (int k = 0; k < 96; ++k) {
            const int t = k / L1.cin, c = k - t * L1.cin, ky = t / L1.k, kx = t - ky * L1.k;
            tab[k] = k < L1.K ? c * L1.ih * L1.iw + ky * L1.iw + kx : -1;
        }
        memset(tab + 96, 0, sizeof(int) * (CONV_FWD_TABS + 5) * CONV_ROWTAB);
        if (fused_forward_supported(Q)) (void)fused_conv_row_tables(Q, tab + 96);
        if (fused_backward_supported(Q)) (void)fused_conv_bwd_row_tables(Q, tab + 96 + CONV_FWD_TABS * CONV_ROWTAB);
        // (+ PT_TOTAL: the patch-word tables of dq_qnet_set_patch_input live behind them in the SAME allocation -- as an allocation of their own their
        // first touch cost every conv workgroup a translation miss of its own: +3K cycles in the conv backward's prologue)
        e = hipMalloc(&Q->kofftab, sizeof(tab) + PT_TOTAL * sizeof(int));
        if (e == hipSuccess) e = hipMemcpy(Q->kofftab, tab, sizeof(tab), hipMemcpyHostToDevice);
        if (e == hipSuccess) e = hipMemset(Q->kofftab + sizeof(tab) / sizeof(int), 0, PT_TOTAL * sizeof(int));
    }
    if (e == hipSuccess && fused_forward_supported(Q))
        for (int i = 0; i < FWD_MAX_JOBS && e == hipSuccess; ++i) e = hipMalloc(&Q->pk_scratch[i], fused_packed_u32x4(Q) * 16);
    if (e == hipSuccess && fused_forward_supported(Q)) {
        const size_t xf = (size_t)cfg->max_batch * Q->L[cfg->n_conv].nin;
        for (int i = 0; i < FWD_MAX_JOBS && e == hipSuccess; ++i) e = hipMalloc(&Q->xinf[i], xf * sizeof(float));
    }
    if (e == hipSuccess && fused_forward_supported(Q)) {            // piece planes x | h1 | gh1 (qnet.h)
        // (+ 16 rows of slack: the dense backward reads the hidden planes' rows of a ragged last tile unclamped, fused_bwd.hip gh1_phase)
        e = hipMalloc(&Q->planes, (dq_planes_halves(Q) + 16 * DENSE_HID) * sizeof(unsigned short));
    }
    const size_t fws = fused_backward_workspace_floats(Q);
    if (e == hipSuccess && fused_forward_supported(Q)) e = hipMalloc(&Q->keep_bits, (size_t)Q->cfg.max_batch * 16 * sizeof(u32));
    if (e == hipSuccess && fws) e = hipMalloc(&Q->fpartial, fws * sizeof(float));
    if (e == hipSuccess && fws) e = hipMemset(Q->fpartial + fws - 8, 0, 8 * sizeof(float));       // {S, 1/S}, range flag, skip word; td_scale_kernel's two work words, discarded-update count, spare
    Q->partial_floats = max_partial;
    if (e != hipSuccess) { dq_set_error("dq_qnet_create: %s", hipGetErrorString(e)); dq_qnet_destroy(Q); return DQ_ERR_HIP; }
    *out = Q;
    return DQ_OK;
}

void dq_qnet_destroy(dq_qnet* Q) {
    if (!Q) return;
    for (int s = 0; s < 2; ++s)
        for (int i = 0; i < QN_MAX_LAYERS; ++i) if (Q->act[s][i]) (void)hipFree(Q->act[s][i]);
    for (int i = 0; i < QN_MAX_LAYERS; ++i) if (Q->gz[i]) (void)hipFree(Q->gz[i]);
    if (Q->partial) (void)hipFree(Q->partial);
    if (Q->keep_bits) (void)hipFree(Q->keep_bits);
    if (Q->fpartial) (void)hipFree(Q->fpartial);
    if (Q->planes) (void)hipFree(Q->planes);
    if (Q->kofftab) (void)hipFree(Q->kofftab);
    for (int i = 0; i < FWD_MAX_JOBS; ++i) if (Q->pk_scratch[i]) (void)hipFree(Q->pk_scratch[i]);
    for (int i = 0; i < FWD_MAX_JOBS; ++i) if (Q->xinf[i]) (void)hipFree(Q->xinf[i]);
    delete Q;
}

size_t dq_qnet_param_count(const dq_qnet* Q) { return Q ? Q->n_params : 0; }

dq_status dq_qnet_layer_info(const dq_qnet* Q, int layer, int64_t* kernel_offset, int64_t* bias_offset, int32_t shape[4], int32_t* n_dims) {
    DQ_REQUIRE(Q && layer >= 0 && layer < Q->n_layers, DQ_ERR_INVALID, "dq_qnet_layer_info: bad layer");
    const Layer& L = Q->L[layer];
    if (kernel_offset) *kernel_offset = (int64_t)L.w_off;
    if (bias_offset) *bias_offset = (int64_t)L.b_off;
    if (shape && n_dims) {
        if (L.kind == 0) { shape[0] = L.k; shape[1] = L.k; shape[2] = L.cin; shape[3] = L.cout; *n_dims = 4; }
        else { shape[0] = L.nin; shape[1] = L.nout; shape[2] = shape[3] = 0; *n_dims = 2; }
    }
    return DQ_OK;
}

int dq_qnet_num_layers(const dq_qnet* Q) { return Q ? Q->n_layers : 0; }

dq_status dq_qnet_set_fused(dq_qnet* Q, int enable) {
    DQ_REQUIRE(Q, DQ_ERR_INVALID, "dq_qnet_set_fused: null handle");
    Q->use_fused = enable ? 1 : 0;
    return DQ_OK;
}

dq_status dq_qnet_set_kernel_forms(dq_qnet* Q, int conv_forward_form, int conv_backward_form, int conv_backward_a1) {
    DQ_REQUIRE(Q && conv_forward_form <= 1 && conv_backward_form <= 2 && conv_backward_a1 <= 1, DQ_ERR_INVALID, "dq_qnet_set_kernel_forms: bad argument");
    if (conv_forward_form >= 0) Q->conv_form = conv_forward_form;
    if (conv_backward_form >= 0) Q->conv_bwd_form = conv_backward_form;
    if (conv_backward_a1 >= 0) Q->conv_bwd_a1 = conv_backward_a1;
    return DQ_OK;
}

dq_status dq_qnet_set_grad_scale(dq_qnet* Q, double grad_scale) {
    DQ_REQUIRE(Q && grad_scale >= 0.0 && grad_scale < 1e30, DQ_ERR_INVALID, "dq_qnet_set_grad_scale: bad argument");
    Q->grad_scale_hint = (float)grad_scale;
    return DQ_OK;
}

dq_status dq_qnet_mark_conv_backward(dq_qnet* Q, void* hip_event) {
    DQ_REQUIRE(Q, DQ_ERR_INVALID, "dq_qnet_mark_conv_backward: null handle");
    Q->mark_event = hip_event;
    return DQ_OK;
}

dq_status dq_qnet_range_check(dq_qnet* Q, void* stream) {
    DQ_REQUIRE(Q, DQ_ERR_INVALID, "dq_qnet_range_check: null handle");
    unsigned* flag = fused_range_flag(Q);
    if (!flag) return DQ_OK;                                        // per-layer path only: f32 throughout
    hipStream_t st = static_cast<hipStream_t>(stream);
    unsigned host = 0;
    DQ_HIP(hipMemcpyAsync(&host, flag, sizeof(host), hipMemcpyDeviceToHost, st));
    DQ_HIP(hipStreamSynchronize(st));
    if (!host) return DQ_OK;
    DQ_HIP(hipMemsetAsync(flag, 0, sizeof(unsigned), st));
    if (host & 2u) {
        dq_set_error("dq_qnet_range_check[forward]: a parameter or an activation of a fused forward is not finite or reaches 65504, the range of the f16 pieces the "
                     "contractions carry (a diverged run: Q-values of that forward are not to be trusted)%s.  dq_qnet_set_fused(net, 0) selects the f32 path",
                     (host & 1u) ? "; the backward's guard fired as well (dq_qnet_range_discarded)" : "");
        return DQ_ERR_RANGE;
    }
    dq_set_error("dq_qnet_range_check: a gradient of the fused backward left the range of its f16 pieces (TD errors of several thousand with the host-known "
                 "gradient scale): an update whose TD step saw such a sample was discarded WHOLE -- no parameter moved, on any rank; count: "
                 "dq_qnet_range_discarded --, a lone non-finite gradient element behind a passing TD step left only its own parameter and moments untouched.  "
                 "dq_td_job.auto_scale carries any finite TD error; dq_qnet_set_fused(net, 0) selects the f32 path");
    return DQ_ERR_RANGE;
}

dq_status dq_qnet_range_discarded(dq_qnet* Q, unsigned* count, void* stream) {
    DQ_REQUIRE(Q && count, DQ_ERR_INVALID, "dq_qnet_range_discarded: null argument");
    *count = 0;
    unsigned* flag = fused_range_flag(Q);
    if (!flag) return DQ_OK;
    hipStream_t st = static_cast<hipStream_t>(stream);
    DQ_HIP(hipMemcpyAsync(count, flag + 4, sizeof(unsigned), hipMemcpyDeviceToHost, st));
    DQ_HIP(hipStreamSynchronize(st));
    if (*count) DQ_HIP(hipMemsetAsync(flag + 4, 0, sizeof(unsigned), st));
    return DQ_OK;
}

int dq_qnet_fused_supported(const dq_qnet* Q) { return Q && fused_forward_supported(Q) ? 1 : 0; }

dq_status dq_qnet_set_patch_input(dq_qnet* Q, int n_syndrome_planes, int stride_words) {
    DQ_REQUIRE(Q, DQ_ERR_INVALID, "dq_qnet_set_patch_input: null handle");
    if (n_syndrome_planes == 0) { Q->patch_depth = 0; return DQ_OK; }
    DQ_REQUIRE(fused_forward_supported(Q) && fused_backward_supported(Q) && fused_patch_supported(Q, n_syndrome_planes), DQ_ERR_UNSUPPORTED,
               "dq_qnet_set_patch_input: needs the fused chains, Conv2D(64, 3, strides=2) on (2d+1)^2 planes with d <= 7, and 4 * syndrome planes + action planes <= 32");
    const int r1 = Q->L[0].rows;
    DQ_REQUIRE(stride_words >= r1 && stride_words <= 64 && (stride_words & (stride_words - 1)) == 0 && stride_words >= 4, DQ_ERR_INVALID,
               "dq_qnet_set_patch_input: stride_words must be a power of two in [max(4, d * d), 64]");
    // One configuration per handle: the tables below and every packed buffer made under them (the compact first kernel, the per-pixel bias) belong to
    // (planes, stride); a second core on the same handle with another volume depth -- or a disable / re-enable with other values -- would silently leave the first
    // one's packed weights inconsistent with the tables.  Switching it off and on again with the SAME values is allowed.
    DQ_REQUIRE(!Q->patch_cfg_depth || (Q->patch_cfg_depth == n_syndrome_planes && Q->patch_cfg_stride == stride_words), DQ_ERR_STATE,
               "dq_qnet_set_patch_input: this handle is configured for %d syndrome planes, rows of %d words; create another handle for (%d, %d)",
               Q->patch_cfg_depth, Q->patch_cfg_stride, n_syndrome_planes, stride_words);
    int tab[PT_TOTAL];
    fused_patch_tables(Q, n_syndrome_planes, stride_words, tab);
    Q->ptab = Q->kofftab + 96 + (CONV_FWD_TABS + 5) * CONV_ROWTAB;  // (behind the row tables, same allocation: dq_qnet_create)
    DQ_HIP(hipMemcpy(Q->ptab, tab, sizeof(tab), hipMemcpyHostToDevice));
    Q->patch_depth = n_syndrome_planes; Q->patch_kd = 4 * n_syndrome_planes + (Q->L[0].cin - n_syndrome_planes); Q->patch_stride = stride_words;
    Q->patch_cfg_depth = n_syndrome_planes; Q->patch_cfg_stride = stride_words;
    return DQ_OK;
}

dq_status dq_qnet_forward(dq_qnet* Q, const float* params_dev, const uint8_t* obs_dev, const int32_t* index_dev, int index_off,
                          int index_mod, int batch, int training, const uint32_t seed[2], uint64_t t, uint32_t sample_base,
                          float* q_dev, void* stream) {
    DQ_REQUIRE(Q && params_dev && obs_dev && q_dev, DQ_ERR_INVALID, "dq_qnet_forward: null argument");
    DQ_REQUIRE(batch >= 1 && batch <= Q->cfg.max_batch, DQ_ERR_INVALID, "dq_qnet_forward: batch %d outside 1..%d", batch, Q->cfg.max_batch);
    DQ_REQUIRE(!training || seed, DQ_ERR_INVALID, "dq_qnet_forward: training forward needs a seed");
    hipStream_t st = (hipStream_t)stream;
    const int set = training ? 0 : 1;
    const float* x = nullptr;
    // (a training forward takes the fused chains only when its backward can too: each path saves what its own backward reads)
    if (Q->use_fused && fused_forward_supported(Q) && (!training || fused_backward_supported(Q))) {
        dq_qnet_job jb;
        memset(&jb, 0, sizeof(jb));
        jb.params_dev = params_dev; jb.obs_dev = obs_dev; jb.index_dev = index_dev; jb.index_off = index_off; jb.index_mod = index_mod;
        jb.batch = batch; jb.training = training;
        if (seed) { jb.seed[0] = seed[0]; jb.seed[1] = seed[1]; }
        jb.t = t; jb.sample_base = sample_base; jb.q_dev = q_dev;
        return fused_forward_multi(Q, 1, &jb, st);
    }
    for (int i = 0; i < Q->n_layers; ++i) {
        const Layer& L = Q->L[i];
        const int M = batch * L.rows;
        Gather ga;
        memset(&ga, 0, sizeof(ga));
        if (L.kind == 0) {
            ga.rows_per_sample = L.rows; ga.RW = L.ow; ga.KC = L.cin; ga.KW = L.k;
            if (i == 0) {               // uint8 NCHW observation, optional replay gather
                ga.src = obs_dev; ga.is_u8 = 1;
                ga.sb = (unsigned)(L.cin * L.ih * L.iw); ga.sy = (unsigned)(L.s * L.iw); ga.sx = (unsigned)L.s;
                ga.sky = L.iw; ga.skx = 1; ga.sc = L.ih * L.iw;
                ga.index = index_dev; ga.index_off = index_off; ga.index_mod = index_mod > 0 ? index_mod : 0x7fffffff;
            } else {                    // NHWC activation
                ga.src = x;
                ga.sb = (unsigned)(L.ih * L.iw * L.cin); ga.sy = (unsigned)(L.s * L.iw * L.cin); ga.sx = (unsigned)(L.s * L.cin);
                ga.sky = L.iw * L.cin; ga.skx = L.cin; ga.sc = 1;
            }
        } else {
            ga = dense_gather(x, L.K);
            if (i == Q->cfg.n_conv) {   // Keras Flatten (channels_first): k = c*HW + p  ->  NHWC offset p*C + c
                ga.KC = Q->flat_hw; ga.KW = 1 << 30; ga.skx = 1; ga.sc = Q->flat_c;
            }
        }
        BMap gb = {params_dev + L.w_off, 1 << 30, 0, L.N, 1};
        Epilogue ep = plain_epilogue(Q->act[set][i], L.N);
        ep.flags = EPI_BIAS | (L.relu ? EPI_RELU : 0);
        ep.bias = params_dev + L.b_off;
        if (training && L.dropout > 0.f) {
            ep.flags |= EPI_DROPOUT;
            ep.keep_scale = (float)(1.0 / (1.0 - (double)L.dropout));
            ep.drop_T = dq_rate_threshold16((double)L.dropout);
            ep.seed0 = seed[0]; ep.seed1 = seed[1]; ep.t = t; ep.sample_base = sample_base;
        }
        launch_fwd(ga, gb, ep, M, L.N, L.K, st);
        DQ_LAUNCH_CHECK();
        x = Q->act[set][i];
    }
    if (Q->cfg.dueling) {
        dueling_fwd_kernel<<<(batch + 3) / 4, 256, 0, st>>>(x, q_dev, batch, Q->cfg.n_actions);
        DQ_LAUNCH_CHECK();
    } else {
        DQ_HIP(hipMemcpyAsync(q_dev, x, (size_t)batch * Q->cfg.n_actions * sizeof(float), hipMemcpyDeviceToDevice, st));
    }
    if (training) {
        Q->last_train_batch = batch; Q->last_train_fused = 0; Q->last_obs = obs_dev; Q->last_index = index_dev;
        Q->last_index_off = index_off; Q->last_index_mod = index_mod;
    }
    return DQ_OK;
}

size_t dq_qnet_packed_bytes(const dq_qnet* Q) { return Q && fused_forward_supported(Q) ? fused_packed_u32x4(Q) * 16 : 0; }

dq_status dq_qnet_pack(const dq_qnet* Q, const float* params_dev, void* packed_dev, void* stream) {
    return fused_pack_weights(Q, params_dev, packed_dev, (hipStream_t)stream);
}

dq_status dq_qnet_forward_multi(dq_qnet* Q, int n_jobs, const dq_qnet_job* jobs, void* stream) {
    DQ_REQUIRE(Q && jobs && n_jobs >= 1, DQ_ERR_INVALID, "dq_qnet_forward_multi: null argument");
    int any_train = 0;
    for (int i = 0; i < n_jobs && jobs; ++i) any_train |= jobs[i].training ? 1 : 0;
    if (Q->use_fused && fused_forward_supported(Q) && n_jobs <= FWD_MAX_JOBS && (!any_train || fused_backward_supported(Q)))
        return fused_forward_multi(Q, n_jobs, jobs, (hipStream_t)stream);
    int n_train = 0;
    for (int i = 0; i < n_jobs; ++i) n_train += jobs[i].training ? 1 : 0;
    DQ_REQUIRE(n_train <= 1, DQ_ERR_INVALID, "dq_qnet_forward_multi: at most one training job per launch");
    for (int i = 0; i < n_jobs; ++i)
        DQ_REQUIRE(!(jobs[i].reserved & 1u), DQ_ERR_UNSUPPORTED, "dq_qnet_forward_multi: patch-word input is read by the fused chains only");
    for (int i = 0; i < n_jobs; ++i) {                             // per-layer path: one forward after the other
        const dq_qnet_job& jb = jobs[i];
        const dq_status rc = dq_qnet_forward(Q, jb.params_dev, jb.obs_dev, jb.index_dev, jb.index_off, jb.index_mod, jb.batch, jb.training, jb.seed,
                                             jb.t, jb.sample_base, jb.q_dev, stream);
        if (rc != DQ_OK) return rc;
    }
    return DQ_OK;
}

// The backward runs on the path its training forward ran on (dq_qnet_set_fused between the two is refused: each forward saves what its
// own backward reads).
#define DQ_SAME_PATH(Q, fused_now)                                                                                          \
    DQ_REQUIRE((Q)->last_train_fused == ((fused_now) ? 1 : 0), DQ_ERR_STATE,                                               \
               "dq_qnet backward: the training forward ran on the %s path, the backward was asked for the other one (dq_qnet_set_fused between them)", \
               (Q)->last_train_fused ? "fused" : "per-layer")

// phases: bit 0 = dense layers (+ dueling), bit 1 = convolutions; 3 = everything
static dq_status backward_phases(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, int phases, hipStream_t st) {
    DQ_REQUIRE(Q && params_dev && grads_dev && ((phases & 1) == 0 || dq_dev), DQ_ERR_INVALID, "dq_qnet_backward: null argument");
    DQ_REQUIRE(Q->last_train_batch > 0, DQ_ERR_STATE, "dq_qnet_backward: no training forward to differentiate");
    const int B = Q->last_train_batch, nl = Q->n_layers, nc = Q->cfg.n_conv;
    DQ_SAME_PATH(Q, Q->use_fused && fused_backward_supported(Q));
    if (Q->use_fused && fused_backward_supported(Q)) return fused_backward(Q, params_dev, dq_dev, grads_dev, phases, st);
    Q->mark_event = nullptr;                                        // (dq_qnet_mark_conv_backward: the per-layer path has no such point)
    if (phases & 1) {
        // gradient w.r.t. the last layer's (linear) output
        float* g = Q->gz[nl - 1];
        if (Q->cfg.dueling) {
            dueling_bwd_kernel<<<(B + 3) / 4, 256, 0, st>>>(dq_dev, g, B, Q->cfg.n_actions);
            DQ_LAUNCH_CHECK();
        } else {
            DQ_HIP(hipMemcpyAsync(g, dq_dev, (size_t)B * Q->cfg.n_actions * sizeof(float), hipMemcpyDeviceToDevice, st));
        }
    }
    for (int i = nl - 1; i >= 0; --i) {
        const bool dense = i >= nc;
        if (!(phases & (dense ? 1 : 2))) continue;
        dq_status rc = layer_wgrad(Q, i, grads_dev, st);
        if (rc != DQ_OK) return rc;
        if (i == 0) break;
        if (dense || (phases & 2)) {
            if (i == nc && !(phases & 1)) continue;                 // (the dense phase already produced gz[nc-1])
            rc = layer_dgrad(Q, params_dev, i, st);
            if (rc != DQ_OK) return rc;
        }
    }
    return DQ_OK;
}

dq_status dq_qnet_backward(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, void* stream) {
    return backward_phases(Q, params_dev, dq_dev, grads_dev, 3, (hipStream_t)stream);
}

dq_status dq_qnet_backward_phase(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, int phase, void* stream) {
    DQ_REQUIRE(phase == 0 || phase == 1, DQ_ERR_INVALID, "dq_qnet_backward_phase: phase is 0 (dense) or 1 (convolutions)");
    return backward_phases(Q, params_dev, dq_dev, grads_dev, phase == 0 ? 1 : 2, (hipStream_t)stream);
}

size_t dq_qnet_conv_param_count(const dq_qnet* Q) { return Q ? Q->L[Q->cfg.n_conv].w_off : 0; }

static TdFused td_fused_from(const dq_td_job* tdj) {
    TdFused td;
    memset(&td, 0, sizeof(td));
    td.q1o = tdj->q_online_s1_dev; td.q1t = tdj->q_target_s1_dev; td.q0 = tdj->q_s0_dev; td.reward = tdj->reward_dev;
    td.terminal = tdj->terminal_dev; td.action = tdj->action_dev; td.index = tdj->index_dev;
    td.gamma = (float)tdj->gamma; td.grad_scale = (float)tdj->grad_scale;
    td.y_out = tdj->y_dev; td.dq_out = tdj->dq_dev; td.metrics = tdj->metrics_dev;
    td.metric_slots = (tdj->batch + 3) / 4 < 1024 ? (tdj->batch + 3) / 4 : 1024;      // what dq_td_metrics(batch) reads
    td.st_done = tdj->done_dev; td.st_was_reset = tdj->was_reset_dev; td.st_lifetime = tdj->lifetime_dev;
    td.st_reward = tdj->step_reward_dev; td.st_n = tdj->n; td.st_stats = reinterpret_cast<unsigned long long*>(tdj->stats_dev);
    td.auto_scale = tdj->auto_scale ? 1 : 0;
    return td;
}

static dq_status check_td_job(const dq_qnet* Q, const dq_td_job* tdj) {
    DQ_REQUIRE(tdj->q_online_s1_dev && tdj->q_target_s1_dev && tdj->q_s0_dev && tdj->reward_dev && tdj->terminal_dev && tdj->action_dev,
               DQ_ERR_INVALID, "dq_qnet_td_backward: null argument");
    DQ_REQUIRE(tdj->batch == Q->last_train_batch && tdj->n_actions == Q->cfg.n_actions, DQ_ERR_INVALID,
               "dq_qnet_td_backward: the TD job does not match the training forward (batch %d x %d actions)", Q->last_train_batch,
               Q->cfg.n_actions);
    DQ_REQUIRE(tdj->n == 0 || (tdj->done_dev && tdj->lifetime_dev && tdj->step_reward_dev && tdj->stats_dev && tdj->n > 0), DQ_ERR_INVALID,
               "dq_qnet_td_backward: bad statistics argument");
    return DQ_OK;
}

static dq_status separate_td(const dq_td_job* tdj, void* stream) {
    DQ_REQUIRE(tdj->dq_dev, DQ_ERR_INVALID, "dq_qnet_td_backward: the per-layer path needs dq_dev");
    return tdj->n > 0
        ? dq_td_update_stats(tdj->q_online_s1_dev, tdj->q_target_s1_dev, tdj->q_s0_dev, tdj->reward_dev, tdj->terminal_dev, tdj->action_dev,
                             tdj->index_dev, tdj->gamma, tdj->batch, tdj->n_actions, tdj->grad_scale, tdj->y_dev, tdj->dq_dev, tdj->metrics_dev,
                             tdj->done_dev, tdj->was_reset_dev, tdj->lifetime_dev, tdj->step_reward_dev, tdj->n, tdj->stats_dev, stream)
        : dq_td_update(tdj->q_online_s1_dev, tdj->q_target_s1_dev, tdj->q_s0_dev, tdj->reward_dev, tdj->terminal_dev, tdj->action_dev,
                       tdj->index_dev, tdj->gamma, tdj->batch, tdj->n_actions, tdj->grad_scale, tdj->y_dev, tdj->dq_dev, tdj->metrics_dev, stream);
}

static dq_status backward_adam(dq_qnet* Q, float* params_dev, const float* dq_dev, const dq_td_job* tdj, float* grads_dev, float* m_dev,
                               float* v_dev, double lr, double beta_1, double beta_2, double epsilon, uint64_t t, void* stream,
                               const EnvParams* rider = nullptr, size_t rider_lds = 0) {
    const bool no_opt = !m_dev && !v_dev;                           // gradient only (the several-GPU path: the all-reduce comes between backward and Adam)
    DQ_REQUIRE(Q && params_dev && (dq_dev || tdj) && grads_dev && (no_opt || (m_dev && v_dev)), DQ_ERR_INVALID, "dq_qnet_backward_adam: null argument");
    DQ_REQUIRE(t >= 1, DQ_ERR_INVALID, "dq_qnet_backward_adam: t counts from 1");
    DQ_REQUIRE(Q->last_train_batch > 0, DQ_ERR_STATE, "dq_qnet_backward_adam: no training forward to differentiate");
    if (tdj) {
        dq_status rc = check_td_job(Q, tdj);
        if (rc != DQ_OK) return rc;
    }
    DQ_SAME_PATH(Q, Q->use_fused && fused_backward_supported(Q));
    if (Q->use_fused && fused_backward_supported(Q)) {
        AdamOpt opt;
        opt.p = params_dev; opt.m = m_dev; opt.v = v_dev;
        opt.lr_t = (float)(lr * sqrt(1.0 - pow(beta_2, (double)t)) / (1.0 - pow(beta_1, (double)t)));      // as dq_adam_step
        opt.b1 = (float)beta_1; opt.b2 = (float)beta_2; opt.eps = (float)epsilon;
        TdFused td;
        if (tdj) td = td_fused_from(tdj);
        return fused_backward(Q, params_dev, dq_dev, grads_dev, 3, (hipStream_t)stream, no_opt ? nullptr : &opt, tdj ? &td : nullptr, rider, rider_lds);
    }
    DQ_REQUIRE(!rider, DQ_ERR_UNSUPPORTED, "dq_qnet_td_backward_adam_env: only the fused chains carry the environment step");
    if (tdj) {                                                      // per-layer path: the separate launches
        dq_status rc = separate_td(tdj, stream);
        if (rc != DQ_OK) return rc;
        dq_dev = tdj->dq_dev;
    }
    dq_status rc = backward_phases(Q, params_dev, dq_dev, grads_dev, 3, (hipStream_t)stream);
    if (rc != DQ_OK || no_opt) return rc;
    return dq_adam_step(params_dev, grads_dev, m_dev, v_dev, Q->n_params, lr, beta_1, beta_2, epsilon, t, stream);
}

dq_status dq_qnet_backward_adam(dq_qnet* Q, float* params_dev, const float* dq_dev, float* grads_dev, float* m_dev, float* v_dev, double lr,
                                double beta_1, double beta_2, double epsilon, uint64_t t, void* stream) {
    DQ_REQUIRE(dq_dev, DQ_ERR_INVALID, "dq_qnet_backward_adam: null argument");
    return backward_adam(Q, params_dev, dq_dev, nullptr, grads_dev, m_dev, v_dev, lr, beta_1, beta_2, epsilon, t, stream);
}

static dq_status fill_rider(dq_qnet* Q, const dq_td_job* td, dq_env* env, const dq_env_step_job* sj, EnvParams* ep, size_t* lds) {
    DQ_REQUIRE(td && env && sj, DQ_ERR_INVALID, "dq_qnet_td_backward_*_env: null argument");
    DQ_REQUIRE(td->n == 0, DQ_ERR_INVALID, "dq_qnet_td_backward_*_env: the step does its own bookkeeping (td->n must be 0)");
    DQ_REQUIRE(Q && Q->use_fused && fused_backward_supported(Q), DQ_ERR_UNSUPPORTED,
               "dq_qnet_td_backward_*_env: only the fused chains carry the environment step");
    return env_fill_act_step(env, sj->q_dev, sj->eps, sj->masked_greedy, sj->seed, sj->t, sj->action_dev, sj->auto_reset, sj->obs_dev,
                             sj->reward_dev, sj->done_dev, sj->legal_dev, sj->lifetime_dev, sj->was_reset_dev, sj->sample, sj->stats_dev, ep, lds,
                             fused_rider_threads());
}

static dq_status td_backward_phase0(dq_qnet* Q, const float* params_dev, const dq_td_job* tdj, float* grads_dev, void* stream,
                                    const EnvParams* rider, size_t rider_lds) {
    DQ_REQUIRE(Q && params_dev && tdj && grads_dev, DQ_ERR_INVALID, "dq_qnet_td_backward_phase0: null argument");
    DQ_REQUIRE(Q->last_train_batch > 0, DQ_ERR_STATE, "dq_qnet_td_backward_phase0: no training forward to differentiate");
    dq_status rc = check_td_job(Q, tdj);
    if (rc != DQ_OK) return rc;
    DQ_SAME_PATH(Q, Q->use_fused && fused_backward_supported(Q));
    if (Q->use_fused && fused_backward_supported(Q)) {
        const TdFused td = td_fused_from(tdj);
        return fused_backward(Q, params_dev, nullptr, grads_dev, 1, (hipStream_t)stream, nullptr, &td, rider, rider_lds);
    }
    DQ_REQUIRE(!rider, DQ_ERR_UNSUPPORTED, "dq_qnet_td_backward_phase0_env: only the fused chains carry the environment step");
    rc = separate_td(tdj, stream);
    if (rc != DQ_OK) return rc;
    return backward_phases(Q, params_dev, tdj->dq_dev, grads_dev, 1, (hipStream_t)stream);
}

dq_status dq_qnet_td_backward_phase0(dq_qnet* Q, const float* params_dev, const dq_td_job* tdj, float* grads_dev, void* stream) {
    return td_backward_phase0(Q, params_dev, tdj, grads_dev, stream, nullptr, 0);
}

dq_status dq_qnet_td_backward_phase0_env(dq_qnet* Q, const float* params_dev, const dq_td_job* td, float* grads_dev, dq_env* env,
                                         const dq_env_step_job* sj, void* stream) {
    EnvParams ep;
    size_t lds = 0;
    const dq_status rc = fill_rider(Q, td, env, sj, &ep, &lds);
    if (rc != DQ_OK) return rc;
    return td_backward_phase0(Q, params_dev, td, grads_dev, stream, &ep, lds);
}

dq_status dq_qnet_td_backward_adam_env(dq_qnet* Q, float* params_dev, const dq_td_job* td, float* grads_dev, float* m_dev, float* v_dev,
                                       double lr, double beta_1, double beta_2, double epsilon, uint64_t t, dq_env* env,
                                       const dq_env_step_job* sj, void* stream) {
    EnvParams ep;
    size_t lds = 0;
    const dq_status rc = fill_rider(Q, td, env, sj, &ep, &lds);
    if (rc != DQ_OK) return rc;
    return backward_adam(Q, params_dev, nullptr, td, grads_dev, m_dev, v_dev, lr, beta_1, beta_2, epsilon, t, stream, &ep, lds);
}

dq_status dq_qnet_adam_step(dq_qnet* Q, float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, double lr, double beta_1, double beta_2,
                            double epsilon, uint64_t t, void* stream) {
    DQ_REQUIRE(Q, DQ_ERR_INVALID, "dq_qnet_adam_step: null handle");
    return adam_step_flagged(params_dev, grads_dev, m_dev, v_dev, Q->n_params, lr, beta_1, beta_2, epsilon, t, fused_range_flag(Q), (hipStream_t)stream);
}

dq_status dq_qnet_td_backward_adam(dq_qnet* Q, float* params_dev, const dq_td_job* td, float* grads_dev, float* m_dev, float* v_dev, double lr,
                                   double beta_1, double beta_2, double epsilon, uint64_t t, void* stream) {
    DQ_REQUIRE(td, DQ_ERR_INVALID, "dq_qnet_td_backward_adam: null TD job");
    return backward_adam(Q, params_dev, nullptr, td, grads_dev, m_dev, v_dev, lr, beta_1, beta_2, epsilon, t, stream);
}

}  // extern "C"
