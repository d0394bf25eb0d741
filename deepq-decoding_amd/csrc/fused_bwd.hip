// Fused backward of the convolutional Q-network (the backward half of keras-rl's trainable_model.train_on_batch,
// /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:119-130), for the configurations fused.hip covers.
//
//   dense_bwd_chain_kernel   workgroup = 8 waves = 16 samples: (optionally the TD step first) dueling backward, then the data
//                            gradients of the three dense layers chained through LDS -- g3 -> gY2 = g3 W3^T -> gH1 = (gY2 W2^T) *
//                            [h1 > 0] * 1/(1-rate) -> gX = (gH1 W1^T) * [x > 0], un-flattened to NHWC -- on the f16 pipe (f16x2,
//                            qnet.h): W^T as packed pieces, every gradient split once, on write, into f16 piece planes (LDS for the
//                            next layer, HBM for the weight gradients).  Extra workgroups carry the episode bookkeeping or the whole
//                            environment step of the vector step (env_dev.h).
//   dense_wgrad_kernel       weight + bias gradients of all dense layers in one launch from the piece planes.
//   conv_bwd_chain_kernel    persistent: data AND weight gradients of the three convolutions, no convolutional gradient touches HBM.
//   reduce_slices_kernel     fixed-order sum of the partials (+ Adam).
// Configurations the fused chains do not cover run through qnet.hip's per-layer kernels.
#include <type_traits>
#include "qnet.h"
#include "conv_bwd.h"
#include "env_dev.h"

// Build-time switches of one-box A/B comparisons (tools/build_ab.sh): GX_PIN = a sched_barrier pin of the gX weight ring, GX_RING = its
// depth in K blocks.  (Round 2: ring 3 unpinned -- the ISA showed hipcc sinking every load next to its use, one or two in flight per wave, the
// phase a chain of L2 latencies (20K cycles); ring 3 pinned = 139 VGPRs = three waves per SIMD = no room for the riding environment
// workgroup beside the dense one.  Round 3: ring 2 PINNED = 117 VGPRs: next block's six loads issued before this block's MFMAs: gX 20K -> 11-14K
// cycles, dense workgroups end 3.5 us earlier in the loop); WG_ROWS (below) = batch rows per iteration of the dense weight gradients; CB_PRIO / DB_PRIO = static
// issue priority for the second-dispatched half of an 8-wave workgroup's waves (conv backward -0.55 us, dense backward -0.2 us: on; the same in the
// dense forward measured nothing).
#ifndef GX_PIN
#define GX_PIN 1
#endif
#ifndef GX_RING
#define GX_RING 2
#endif
#ifndef ENV_PRIO
#define ENV_PRIO 3                      // the riding environment workgroups (short latency chains) at top issue priority: they end at 14 us instead of 21
#endif
#ifndef CB_PRIO
#define CB_PRIO 1
#endif
#ifndef DB_PRIO
#define DB_PRIO 0
#endif
DQ_STAMP_READER(dq_dbg_read_bwd)

// The data gradients multiply by W^T.  dq_qnet_pack writes W2^T and W1^T as f16 pieces in MFMA operand order (qnet.h dense2t / dense1t,
// the Keras Flatten permutation folded into dense1t's columns), so both run on the f16 pipe (f16x2) with one coalesced 16-byte load per
// lane, tile and piece; the gradients they multiply are split ONCE, when they are produced, into f16 piece planes in LDS.
struct DenseBwdArgs {
    const float* params;
    const u32x4* packed;                // f16 pieces of the training forward's weights (qnet.h)
    int pk_dense2t, pk_dense1t, KB2;    // u32x4 offsets of the transposed dense sections; K = 32 blocks of gH1's reduction (N2 padded)
    const float* dq;                    // [batch, n_actions]
    const unsigned short* h1_pl;        // saved hidden output (post ReLU + dropout) as f16 piece planes [2][plane_rows][512]: mask of gH1
    const float* x;                     // saved last-convolution output [batch, K1] (NHWC): mask of gX
    int batch, K1, perm_hw, perm_c;
    int N2, N3, n_actions;
    int w_off[3];
    float mask_scale;                   // 1/(1-rate) of the hidden layer's dropout (1 if none)
    // (the gradients leave as f16 piece planes only -- what the weight-gradient kernel reads; the per-layer path keeps f32 copies of its own)
    unsigned short* gh1_pl;             // gH1 as f16 piece planes [2][plane_rows][512] (the weight gradient's operand)
    unsigned short* gy2_pl;             // ... [2][plane_rows][small_ld]
    unsigned short* g3_pl;              // ... [2][plane_rows][small_ld]
    int plane_rows, small_ld;
    unsigned short* gx_pl;              // gX [batch][K1] (NHWC) as f16 piece planes (h plane; the l plane gx_lo halves further): split on write --
    size_t gx_lo;                       // it is the convolutional backward's g3 operand, staged there by LDS-DMA and read without arithmetic
    int ldg;                            // LDS row stride of the dq image (floats); the gY2 planes have rows of 32 KB2 + 8 halves
    int off_g3, off_gy2, off_gh1;
    int pk_wc;                          // Wc (qnet.h wc), u32x4 offset
    int pk_w3q, w3q_rows, w3q_pw, off_w3t;  // the folded dueling layer (qnet.h w3q: W3' [w3q_rows + 1][w3q_pw], then W3'^T): u32x4 offset, rows, row stride
                                        // (the FORWARD's tiling: 64 or 128); LDS offset of W3'^T
    float gs;                           // every gradient of the fused backward is carried scaled by this power of two (GradScale below) ...
    const float* gs_dev;                // ... or, when not NULL, by gs_dev[0] (computed on the device from max |dq|)
    int td_on, dense_wgs;               // td_on: dq is computed here from `td`; workgroups >= dense_wgs do the episode bookkeeping ...
    int dense_tiles, col_split;         // dense_wgs = dense_tiles (row tiles of 16 samples) x col_split (1, 2 or 4: small minibatches, see the kernel)
    int env_on;                         // ... or, with a rider, the environment step (+ its replay sampling and bookkeeping): env_block<8>
    TdFused td;
    // range guard, early half (TD launches): a sample whose S x dq leaves the range the chain carries safely (|S dq| >= 2^15: TD errors of several
    // thousand with the host-known scale) stamps this backward's number into *skip_word; the final reduction then discards the WHOLE update -- every
    // gradient element becomes NaN, no parameter moves, the range flag is raised -- instead of applying the elements that happened to stay finite
    unsigned* skip_word;
    unsigned skip_tag;
};

// ---- gradient scale ----------------------------------------------------------------------------------------------------
// The f16 pieces (qnet.h) carry 22 significant bits only for |x| in [2^-14, 65504), and loss gradients are small (dq = TD error / batch).
// The backward is linear in dq, so the fused backward carries S * gradient everywhere, S a power of two (exact), and the final reduction
// multiplies the weight gradient by 1/S.  With the TD step fused in, S = 2^k with S * grad_scale in [4, 8): TD errors between 1.5e-5 and
// 8000 are carried at full precision (smaller ones at 2^-36 / 4 absolute).  With a caller-supplied dq, grad_scale_kernel reads max |dq|
// and picks S with S * max |dq| in (128, 256].  S lives in kernel arguments (host-known) or in two floats behind the workspace.
__global__ __launch_bounds__(1024) void grad_scale_kernel(const float* __restrict__ dq, int n, float* __restrict__ out) {
    __shared__ float sh[16];
    float m = 0.f;
    for (int i = threadIdx.x; i < n; i += 1024) m = fmaxf(m, fabsf(dq[i]));
    for (int k = 32; k >= 1; k >>= 1) m = fmaxf(m, __shfl_xor(m, k));
    if ((threadIdx.x & 63) == 0) sh[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
        int e = 0;
        float S = 1.f;
        if (m > 0.f && m < INFINITY) { (void)frexpf(m, &e); S = ldexpf(1.f, max(-100, min(100, 8 - e))); }      // m = f 2^e, f in [0.5, 1)
        out[0] = S; out[1] = 1.f / S;
    }
}

// The same choice with the TD step fused in and dq_td_job.auto_scale set: max |TD error x grad_scale| of THIS minibatch, measured by a small launch in
// front of the dense backward (one wave per sample: dq_td_update's arithmetic without its stores; block maxima by one atomic each; the last block to
// arrive turns the maximum into {S, 1/S} with S * max in (32, 64] and clears the two work words).  The host-known scale (S * grad_scale in [4, 8)) carries
// TD errors up to a few thousand; with the measured one the backward takes ANY finite TD error fp32 can hold, like the reference's TensorFlow
// (keras-rl delta_clip = inf): what is left to the range guard is a product of weights beyond 2^10, not a large loss.
struct TdScaleArgs { TdFused td; int B, A; float* out; unsigned* work; };     // work[0]: bits of the running maximum, work[1]: blocks arrived
#define TDS_ROWS 4                      // samples per wave (their loads in flight together); 16 waves per block: 64 samples, two atomics per block
__global__ __launch_bounds__(1024) void td_scale_kernel(TdScaleArgs a) {
    __shared__ float sh[16];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, b0 = (blockIdx.x * 16 + wave) * TDS_ROWS;
    float q1[TDS_ROWS];
    int rr[TDS_ROWS];
#pragma unroll
    for (int u = 0; u < TDS_ROWS; ++u) {                            // (A <= 64 * 2 on the fused chains: two entries per lane would be needed beyond 64 -- folded below)
        const int b = min(b0 + u, a.B - 1);
        q1[u] = lane < a.A ? a.td.q1o[(size_t)b * a.A + lane] : -INFINITY;
        rr[u] = a.td.index ? a.td.index[b] : b;
    }
    float mine = 0.f;
#pragma unroll
    for (int u = 0; u < TDS_ROWS; ++u) {
        const int b = min(b0 + u, a.B - 1);
        float best = q1[u];
        int best_a = lane < a.A ? lane : 0x7fffffff;
        for (int c = lane + 64; c < a.A; c += 64) {                 // (more than 64 actions: the rest of the row)
            const float v = a.td.q1o[(size_t)b * a.A + c];
            if (best_a == 0x7fffffff || v > best) { best = v; best_a = c; }
        }
        dq_wave_argmax(best, best_a);
        const int r = rr[u];
        const float qn = a.td.q1t[(size_t)b * a.A + best_a];
        const float y = a.td.reward[r] + (a.td.terminal[r] ? 0.f : a.td.gamma * qn);
        float v = fabsf((a.td.q0[(size_t)b * a.A + a.td.action[r]] - y) * a.td.grad_scale);
        if (!(v < INFINITY)) v = INFINITY;                          // (NaN: the guard's business)
        mine = fmaxf(mine, v);
    }
    if (lane == 0) sh[wave] = mine;
    __syncthreads();
    if (threadIdx.x == 0) {
        float m = sh[0];
#pragma unroll
        for (int w = 1; w < 16; ++w) m = fmaxf(m, sh[w]);
        atomicMax(&a.work[0], __builtin_bit_cast(unsigned, m));     // (non-negative floats order as their bit patterns)
        __threadfence();
        if (atomicAdd(&a.work[1], 1u) == gridDim.x - 1) {
            __threadfence();
            const float mx = __builtin_bit_cast(float, atomicMax(&a.work[0], 0u));
            int e = 0;
            float S = 1.f;
            if (mx > 0.f && mx < INFINITY) { (void)frexpf(mx, &e); S = ldexpf(1.f, max(-100, min(100, 6 - e))); }      // mx = f 2^e, f in [0.5, 1)
            a.out[0] = S; a.out[1] = 1.f / S;
            a.work[0] = 0u; a.work[1] = 0u;
        }
    }
}

// NTP adjacent column tiles [tile0, tile0 + NTP) of gX = (gH1 W1^T) * [x > 0] for this wave, on the f16 pipe: K = 512 in 16 blocks; A = the
// gH1 piece planes in LDS (one ds_read_b128 per piece and block, shared by the tiles), B = dense1t pieces streamed through a ring of three
// blocks (the kernel is bound by this stream: every workgroup reads all of W1^T through the CU's vector-memory path).  Column 16 tile + j
// is the NHWC offset itself (the permutation lives in the packed columns), so masks and results are 64-byte row segments.
template <int NTP>
__device__ __forceinline__ void gx_pass(const DenseBwdArgs& a, const unsigned short* __restrict__ s_gh1p, int tile0, int b0, int ns, int lane) {
    constexpr int LDH = DENSE_HID + 8, NB = DENSE_HID / 32, RING = GX_RING;
    const int j = lane & 15, kq = lane >> 4, K1 = a.K1, tiles = K1 >> 4;
    const u32x4* pk = opaque_global(a.packed + a.pk_dense1t + (size_t)tile0 * PK_BLOCK) + lane;
    const unsigned short* arow = s_gh1p + j * LDH + 8 * kq;
    F16x2 bw[RING][NTP];
    auto load = [&](int blk, F16x2 (&b)[NTP]) {
        const u32x4* pb = pk + (size_t)blk * tiles * PK_BLOCK;
#pragma unroll
        for (int t = 0; t < NTP; ++t) { b[t].h = pb[t * PK_BLOCK]; b[t].l = pb[t * PK_BLOCK + PK_LO]; }
    };
#pragma unroll
    for (int u = 0; u < RING - 1; ++u) load(u, bw[u]);
    float xm[NTP][4];                                               // the mask operand x (requested now, used in the epilogue)
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) xm[t][r] = a.x[(size_t)(b0 + min(4 * kq + r, ns - 1)) * K1 + 16 * (tile0 + t) + j];
    f32x4 acc[NTP][2];
#pragma unroll
    for (int t = 0; t < NTP; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
    static_assert((NB - 1) % RING == 0 || NB % RING == 0, "the block loop is unrolled in whole rings (plus one)");
    auto block = [&](int blk, int u) {
        load(min(blk + RING - 1, NB - 1), bw[(u + RING - 1) % RING]);      // unconditional (clamped): static s_waitcnt counts
        if (GX_PIN) __builtin_amdgcn_sched_barrier(0);              // the requests stay RING - 1 blocks ahead of their use (hipcc sinks them next to it)
        F16x2 av;
        av.h = *reinterpret_cast<const u32x4*>(arow + 32 * blk);
        av.l = *reinterpret_cast<const u32x4*>(arow + 32 * blk + DENSE_ROWS * LDH);
#pragma unroll
        for (int t = 0; t < NTP; ++t) mma_f16x3(av, bw[u][t], acc[t][0], acc[t][1]);
    };
    if constexpr (NB % RING == 0) {
        for (int blk = 0; blk < NB; blk += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) block(blk + u, u);
        }
    } else {
        for (int blk = 0; blk + 1 < NB; blk += RING) {
#pragma unroll
            for (int u = 0; u < RING; ++u) block(blk + u, u);
        }
        block(NB - 1, 0);
    }
#pragma unroll
    for (int t = 0; t < NTP; ++t)
#pragma unroll
        for (int r = 0; r < 4; r += 2) {                            // rows 4kq + r, + 1 as one pair: four instructions split both (qnet.h)
            const float v0 = xm[t][r] > 0.f ? f16x2_sum(acc[t][0][r], acc[t][1][r]) : 0.f;
            const float v1 = xm[t][r + 1] > 0.f ? f16x2_sum(acc[t][0][r + 1], acc[t][1][r + 1]) : 0.f;
            u32 h, l;
            split_f16x2_pair(v0, v1, h, l);
            unsigned short* gp = a.gx_pl + (size_t)(b0 + 4 * kq + r) * K1 + 16 * (tile0 + t) + j;
            if (4 * kq + r < ns) { gp[0] = (unsigned short)h; gp[a.gx_lo] = (unsigned short)l; }
            if (4 * kq + r + 1 < ns) { gp[K1] = (unsigned short)(h >> 16); gp[K1 + a.gx_lo] = (unsigned short)(l >> 16); }
        }
}

// gH1 = (gY2 W2^T) * [h1 > 0] * scale on the f16 pipe: K = N2 padded to KB2 blocks of 32 (a compile-time count: a register ring indexed by a
// run-time block number would live in scratch memory); wave w owns columns 64w + 4j + t.  The result is split on write into the piece
// planes gX and the weight gradients read.
// What gh1_phase needs that does NOT depend on the gradient -- its first K block of W2^T pieces and the mask operand (the saved hidden
// output's pieces) -- is requested at the top of the kernel, under the TD step's own load latencies (round 3: in the loop the phase
// spent most of its 7-10K cycles waiting for exactly these loads).
#ifndef GH1_PRE_W
#define GH1_PRE_W 0                     // 1: the first weight block too (+32 registers across the TD step: 155 VGPRs, three waves per SIMD -- off); 2: behind the TD step, before gY2
#endif
struct Gh1Pre { F16x2 bw0[4]; uint2 hvh[4], hvl[4]; };
__device__ __forceinline__ void gh1_preload(const DenseBwdArgs& a, int b0, int wave, int lane, Gh1Pre& P) {
    const int j = lane & 15, kq = lane >> 4;
    const int c0 = 64 * wave + 4 * j;
    const u32x4* pk = a.packed + a.pk_dense2t + (size_t)(4 * wave) * PK_BLOCK + lane;
    if (GH1_PRE_W == 1) {
#pragma unroll
        for (int t = 0; t < 4; ++t) { P.bw0[t].h = pk[t * PK_BLOCK]; P.bw0[t].l = pk[t * PK_BLOCK + PK_LO]; }
    }
    // the mask operand: the saved hidden output's pieces (h > 0 or, for values below f16's range, l > 0  <=>  the f32 value was > 0)
    // (rows past the batch are read unclamped -- two base addresses + immediate offsets; eight clamped addresses cost 16 registers and
    // with them the fourth wave per SIMD, i.e. the co-residence of the riding environment workgroups -- and ignored: they lie inside the
    // plane buffer, whose sets follow each other, qnet.h)
    const unsigned short* hp = a.h1_pl + (size_t)(b0 + 4 * kq) * DENSE_HID + c0;
    const unsigned short* lp = hp + (size_t)a.plane_rows * DENSE_HID;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        P.hvh[r] = *reinterpret_cast<const uint2*>(hp + r * DENSE_HID);
        P.hvl[r] = *reinterpret_cast<const uint2*>(lp + r * DENSE_HID);
    }
}

template <int KB2>
__device__ __forceinline__ void gh1_phase(const DenseBwdArgs& a, const unsigned short* __restrict__ s_gy2p, unsigned short* __restrict__ s_gh1p,
                                          int b0, int ns, int wave, int lane, const Gh1Pre& P) {
    constexpr int LDH = DENSE_HID + 8, LDY = 32 * KB2 + 8;
    const int j = lane & 15, kq = lane >> 4;
    const int c0 = 64 * wave + 4 * j;
    const u32x4* pk = a.packed + a.pk_dense2t + (size_t)(4 * wave) * PK_BLOCK + lane;
    F16x2 bw[2][4];                                                 // two blocks in flight (the loop below is fully unrolled: static indices)
#pragma unroll
    for (int t = 0; t < 4; ++t) {
        if (GH1_PRE_W) bw[0][t] = P.bw0[t];
        else { bw[0][t].h = pk[t * PK_BLOCK]; bw[0][t].l = pk[t * PK_BLOCK + PK_LO]; }
    }
    const uint2 (&hvh)[4] = P.hvh;
    const uint2 (&hvl)[4] = P.hvl;
    f32x4 acc[4][2];
#pragma unroll
    for (int t = 0; t < 4; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
    const unsigned short* grow = s_gy2p + j * LDY + 8 * kq;
#pragma unroll
    for (int b = 0; b < KB2; ++b) {
        if (b + 1 < KB2) {
#pragma unroll
            for (int t = 0; t < 4; ++t) { bw[(b + 1) & 1][t].h = pk[((b + 1) * 32 + t) * PK_BLOCK]; bw[(b + 1) & 1][t].l = pk[((b + 1) * 32 + t) * PK_BLOCK + PK_LO]; }
        }
        F16x2 av;
        av.h = *reinterpret_cast<const u32x4*>(grow + 32 * b);
        av.l = *reinterpret_cast<const u32x4*>(grow + 32 * b + DENSE_ROWS * LDY);
#pragma unroll
        for (int t = 0; t < 4; ++t) mma_f16x3(av, bw[b & 1][t], acc[t][0], acc[t][1]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const int row = 4 * kq + r;
        f32x4 v;
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const u32 hw = t < 2 ? hvh[r].x : hvh[r].y, lw = t < 2 ? hvl[r].x : hvl[r].y;
            const u32 pos = ((hw | lw) >> (16 * (t & 1))) & 0x7fffu;               // (pieces of a value >= 0: any magnitude bit set)
            v[t] = (pos != 0u && row < ns) ? f16x2_sum(acc[t][0][r], acc[t][1][r]) * a.mask_scale : 0.f;
        }
        u32 hp[2], lp[2];
        split_f16x2_pair(v[0], v[1], hp[0], lp[0]);
        split_f16x2_pair(v[2], v[3], hp[1], lp[1]);
        *reinterpret_cast<uint2*>(s_gh1p + row * LDH + c0) = uint2{hp[0], hp[1]};
        *reinterpret_cast<uint2*>(s_gh1p + (DENSE_ROWS + row) * LDH + c0) = uint2{lp[0], lp[1]};
        if (row < ns) {
            unsigned short* gp = a.gh1_pl + (size_t)(b0 + row) * DENSE_HID + c0;
            *reinterpret_cast<uint2*>(gp) = uint2{hp[0], hp[1]};
            *reinterpret_cast<uint2*>(gp + (size_t)a.plane_rows * DENSE_HID) = uint2{lp[0], lp[1]};
        }
    }
}

#ifndef DB_SHORT_LEAN
#define DB_SHORT_LEAN 0                 // 1: the shortcut without the LDS copy of W3'^T (its row from L2), without clearing gY2's image and without the first barrier
#endif
template <int NT2, bool TD, bool DUEL = true>      // N2 <= 16*NT2 and N3 <= 16*NT2; TD: the TD step in the prologue (a compile-time switch: around loads a run-time one
                                        // is a branch whose merge hipcc guards with s_waitcnt vmcnt(0) -- every preload below was waited for at once); DUEL: a dueling layer
                                        // whose tables pack_weights_kernel built (with TD: the shortcut SHORT below)
__global__ __launch_bounds__(DENSE_THREADS, 2) void dense_bwd_chain_kernel(DenseBwdArgs a, EnvParams env) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    // The vector step's environment launch does not feed this update (the minibatch never holds the newest transition, common.h
    // dq_replay_row) and this update does not feed it (it acts on Q-values the forward already wrote): the lattices' step rides on this
    // launch as extra workgroups.  The dense chain is one 8-wave workgroup per CU with long dependent phases; the environment is a
    // latency chain per lattice: together they fill each other's idle issue slots instead of taking 15 us of their own.
    DQ_STAMP_ALL(0);
    if (a.env_on && (int)blockIdx.x >= a.dense_wgs) {               // block-uniform
        if (ENV_PRIO) __builtin_amdgcn_s_setprio(ENV_PRIO);
        if (env.pair) env_block2<16>(env, (int)blockIdx.x - a.dense_wgs, smem);      // two lattices per wave (d <= 5): ONE round of environment workgroups
        else env_block<8>(env, (int)blockIdx.x - a.dense_wgs, smem);
        DQ_STAMP_ALL(1);
        return;
    }
    float* s_g3 = reinterpret_cast<float*>(smem + a.off_g3);
    unsigned short* s_gy2p = reinterpret_cast<unsigned short*>(smem + a.off_gy2);    // gY2 as f16 piece planes [2][16][LDY]
    unsigned short* s_gh1p = reinterpret_cast<unsigned short*>(smem + a.off_gh1);    // gH1 as planes [2][16][LDH]
    const int LDY = 32 * a.KB2 + 8;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
    // A minibatch of fewer row tiles than half the CUs (c5: 1024 samples = 64 workgroups on 256 CUs, each streaming all of W1^T -- 1.6 MB at d = 7 -- through
    // its CU's vector-memory path: gX was 20 of that launch's 29 us) is spread further: col_split workgroups per row tile, each with the tile's whole
    // prologue (the same values: its stores are duplicates of the same bits; the loss / mean-Q partial is written by the first only) and a quarter
    // or half of gX's column tiles -- a quarter or half of the weight stream.  c3's 256 row tiles run as before (col_split 1).
    const int bt = a.col_split > 1 ? (int)blockIdx.x % a.dense_tiles : (int)blockIdx.x, cs = a.col_split > 1 ? (int)blockIdx.x / a.dense_tiles : 0;
    const int b0 = bt * DENSE_ROWS;
    const int ns = min(DENSE_ROWS, a.batch - b0);
    const int A = a.n_actions, N2 = a.N2, N3 = a.N3, ldg = a.ldg;
    const float GS = a.gs_dev ? a.gs_dev[0] : a.gs;                  // gradient scale (a power of two), wave-uniform

    if ((int)blockIdx.x >= a.dense_wgs) {                           // the episode bookkeeping of the step just taken rides along
        dq_episode_stats_lane(a.td.st_done, a.td.st_was_reset, a.td.st_lifetime, a.td.st_reward, a.td.st_n,
                              ((int)blockIdx.x - a.dense_wgs) * DENSE_THREADS + tid, a.td.st_stats);
        return;
    }
    if (DB_PRIO && wave >= DENSE_WAVES / 2) __builtin_amdgcn_s_setprio(1);   // (as in the conv backward: the younger half of the workgroup's waves)
    DQ_STAMP(DQ_TAG_DENSE_BWD, 0);
    constexpr int RPW = DENSE_ROWS / DENSE_WAVES;                   // rows per wave (2): their loads are issued together
    static_assert(RPW == 2, "two rows per wave");
    // ---- the TD step's three Q rows of this wave's two samples do not hang on the replay rows: requested first of all (round 3, second pass:
    //      they used to go out behind the first barrier, a round trip of their own) ----
    float loss = 0.f, mq = 0.f;
    float yb[RPW] = {0.f, 0.f}, qv[RPW][2] = {{0.f, 0.f}, {0.f, 0.f}};
    int a_b[RPW] = {-1, -1};
    float q1[RPW][2] = {{0.f, 0.f}, {0.f, 0.f}}, q1t[RPW][2] = {{0.f, 0.f}, {0.f, 0.f}}, rw[RPW] = {0.f, 0.f};
    int term[RPW] = {0, 0};
    if constexpr (TD) {
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int b = b0 + min(wave + DENSE_WAVES * u, ns - 1);
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                const bool ok = c < A;
                q1[u][h] = a.td.q1o[(size_t)b * A + (ok ? c : 0)];
                q1t[u][h] = a.td.q1t[(size_t)b * A + (ok ? c : 0)];
                qv[u][h] = a.td.q0[(size_t)b * A + (ok ? c : 0)];
            }
        }
    }
    // ---- requested now, used two / three phases later: the folded dueling layer for gY2, gH1's first weight block and mask operand --------
    // The dueling layer's backward and gY2 = g3 W3^T are ONE linear map of dq: gY2 = dq W3'^T, W3' the forward's folded matrix (qnet.h w3q).  With the
    // TD step fused in, dq has one non-zero per row -- gY2[b] = dq[b][a_b] * W3'^T[a_b] --: a row of the 16 KB table, copied into LDS at the top, times
    // a scalar; no dq image, no matrix phase, no barrier in between (SHORT: tables up to 64 x 64; round 3: that phase and its barrier were 3K of a
    // workgroup's 34K cycles).  A caller's dense dq goes through the matrix pipe as before, K = |A|: the same bits where dq has one non-zero per row.
    constexpr bool SHORT = TD && DUEL;
    constexpr bool W3LDS = SHORT && NT2 == 4 && !DB_SHORT_LEAN;                       // W3'^T (64 x 64) copied into LDS; wider tables (|A| > 64): its row read from the L2-resident table
    constexpr int PW3 = 16 * NT2;
    float w3b[NT2][4];
    f32x4 wt3[2];
    if constexpr (W3LDS) {
        const f32x4* src = reinterpret_cast<const f32x4*>(reinterpret_cast<const float*>(a.packed + a.pk_w3q) + (size_t)(a.w3q_rows + 1) * PW3);
        wt3[0] = N3 > 0 ? src[tid] : f32x4{0.f, 0.f, 0.f, 0.f};      // 64 x 64 floats = 2 x 16 bytes per thread
        wt3[1] = N3 > 0 ? src[tid + DENSE_THREADS] : f32x4{0.f, 0.f, 0.f, 0.f};
    } else if (!SHORT && N3 > 0 && wave < NT2) {
        const float* w3q = reinterpret_cast<const float*>(a.packed + a.pk_w3q);
        const int n2 = 16 * wave + j;                               // (the table is zero-padded: rows past N2, columns past |A|)
#pragma unroll
        for (int g = 0; g < NT2; ++g) {
            const f32x4 v = *reinterpret_cast<const f32x4*>(w3q + (size_t)n2 * a.w3q_pw + 16 * g + 4 * kq);
            w3b[g][0] = v[0]; w3b[g][1] = v[1]; w3b[g][2] = v[2]; w3b[g][3] = v[3];
        }
    }
    // SHORT goes one step further: gH1 = gY2 W2^T = dq[b][a_b] * Wc[a_b], Wc = W3'^T W2^T [|A|][512] built by pack_weights_kernel -- a 2 KB row of an
    // L2-resident table times a scalar, masked; every wave forms the gH1 rows of ITS two samples (whose TD error and action it holds): no matrix
    // phase, no W2^T stream (64 KB per workgroup), no barrier (round 3: that phase was 5.5K of a workgroup's 30K cycles).  Its mask operand -- the
    // saved hidden output's pieces of the wave's rows, eight units per lane -- is requested here.
    Gh1Pre gh1_pre;
    u32x4 mkh[RPW], mkl[RPW];
    if constexpr (SHORT) {
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const unsigned short* hp = a.h1_pl + (size_t)(b0 + min(wave + DENSE_WAVES * u, ns - 1)) * DENSE_HID + 8 * lane;
            mkh[u] = *reinterpret_cast<const u32x4*>(hp);
            mkl[u] = *reinterpret_cast<const u32x4*>(hp + (size_t)a.plane_rows * DENSE_HID);
        }
    } else {
        gh1_preload(a, b0, wave, lane, gh1_pre);
    }
    __builtin_amdgcn_sched_barrier(0);                              // (hipcc sinks a load next to its use: these stay up here)
    // (wave-uniform rows: SCALAR loads through the constant address space -- the index vector is not written while this kernel runs, the
    // riders draw the next update's rows into the other buffer (core.py) --, both addresses formed first so that the two loads go out together,
    // unconditional at a clamped address and selected at the use, BEHIND the vector preloads above and in front of the LDS clearing and the
    // barrier that cover their latency: as `index ? index[b] : b` each was a vector load in a branch of its own with s_waitcnt vmcnt(0) behind
    // it, two serialised round trips = 4K of the 5.4K cycles this workgroup spent in front of its first barrier)
    int ridx[RPW] = {0, 0};
    if constexpr (TD) {
        const bool has_index = a.td.index != nullptr;
        const __attribute__((address_space(4))) int32_t* ip =
            (const __attribute__((address_space(4))) int32_t*)(uintptr_t)(has_index ? (const void*)a.td.index : (const void*)a.params);
        const __attribute__((address_space(4))) int32_t* p0 = ip + (has_index ? b0 + min(wave, ns - 1) : 0);
        const __attribute__((address_space(4))) int32_t* p1 = ip + (has_index ? b0 + min(wave + DENSE_WAVES, ns - 1) : 0);
        asm volatile("" : "+s"(p0), "+s"(p1));
        ridx[0] = *p0; ridx[1] = *p1;
    }
    __shared__ float s_met[DENSE_WAVES][2];
    if constexpr (W3LDS) {
        f32x4* s_w3t = reinterpret_cast<f32x4*>(smem + a.off_w3t);  // W3'^T [a][n2] (waited for here: the loads went out first of all)
        s_w3t[tid] = wt3[0]; s_w3t[tid + DENSE_THREADS] = wt3[1];
    } else if constexpr (!SHORT) {
        for (int i = tid; i < DENSE_ROWS * ldg; i += DENSE_THREADS) s_g3[i] = 0.f;
    }
    // (SHORT writes whole rows of both gradient images, each wave its own: rows past the batch stay undefined -- they only reach MFMA output rows that
    // are never stored --, so DB_SHORT_LEAN drops the clearing and the barrier behind it)
    if (!(SHORT && DB_SHORT_LEAN))
        for (int i = tid; i < DENSE_ROWS * LDY; i += DENSE_THREADS) reinterpret_cast<u32*>(s_gy2p)[i] = 0u;      // both planes (2 x 16 x LDY halves)
    if constexpr (TD) {                                             // the replay rows' fields: in flight across the barrier
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int rr = a.td.index ? ridx[u] : b0 + min(wave + DENSE_WAVES * u, ns - 1);
            rw[u] = a.td.reward[rr];
            term[u] = a.td.terminal[rr];
            a_b[u] = a.td.action[rr];
        }
    }
    if (!(SHORT && DB_SHORT_LEAN)) __syncthreads();
    DQ_STAMP(DQ_TAG_DENSE_BWD, 6);
    f32x4 wcr[RPW][2];                                              // SHORT: this wave's rows of Wc, in flight under the TD arithmetic below
    if constexpr (SHORT) {
        const float* wc = reinterpret_cast<const float*>(a.packed + a.pk_wc);
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int ab = __builtin_amdgcn_readfirstlane(a_b[u]);
            const f32x4* row = reinterpret_cast<const f32x4*>(wc + (size_t)min(max(ab, 0), A - 1) * DENSE_HID + 8 * lane);
            wcr[u][0] = row[0]; wcr[u][1] = row[1];
        }
    }
    float w3r[RPW][2] = {{0.f, 0.f}, {0.f, 0.f}};                   // SHORT beyond 64 actions: row a_b of W3'^T [a][n2], columns lane and lane + 64 (zero past N2: the table's padding)
    if constexpr (SHORT && !W3LDS) {
        const float* w3t = reinterpret_cast<const float*>(a.packed + a.pk_w3q) + (size_t)(a.w3q_rows + 1) * a.w3q_pw;
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int ab = __builtin_amdgcn_readfirstlane(a_b[u]);
            const float* row = w3t + (size_t)min(max(ab, 0), A - 1) * a.w3q_rows;
            w3r[u][0] = row[lane];
            w3r[u][1] = lane + 64 < a.w3q_rows ? row[lane + 64] : 0.f;
        }
    }
    // ---- (TD step: y = r + gamma (1 - terminal) Q_target(s1)[argmax Q_online(s1)], dq = (Q(s0)[a] - y) * scale at the action taken,
    //      dqn.hip td_update_kernel's arithmetic, one wave per sample) then the dueling backward:
    //      g3[b,0] = sum_a dq[b,a];  g3[b,1+a] = dq[b,a] - (1/A) sum_a' dq[b,a'] ---------------------------------------------
    if constexpr (TD) {
        // No load stage of its own any more (round 3: the Q rows are requested at the top of the kernel, the replay rows behind the preloads and
        // their fields in front of the barrier above; before: Q_target(s1)[a*] was fetched behind the arg-max and the replay-row fields behind
        // the row number -- dependent round trips of 3K and 6K cycles in the loop, where 768 workgroups start at once); Q_target(s1) comes as the
        // whole row and its a*-th entry is picked by a shuffle, like Q(s0)[a].
        float qt[RPW];
        DQ_STAMP(DQ_TAG_DENSE_BWD, 7);
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            float best = -INFINITY;
            int best_a = 0x7fffffff;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                if (c < A && (best_a == 0x7fffffff || q1[u][h] > best)) { best = q1[u][h]; best_a = c; }
            }
            dq_wave_argmax(best, best_a);                           // (DPP + v_readlane, common.h: no LDS round trips)
            // Q_target(s1)[best_a] lives in lane best_a & 63, half best_a >> 6 (best_a is wave-uniform: v_readlane)
            qt[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, best_a < 64 ? q1t[u][0] : q1t[u][1]), best_a & 63));
        }
        DQ_STAMP(DQ_TAG_DENSE_BWD, 8);
#pragma unroll
        for (int u = 0; u < RPW; ++u) {
            const int row = wave + DENSE_WAVES * u;
            if (row >= ns) continue;
            const int b = b0 + row;
            yb[u] = rw[u] + (term[u] ? 0.f : a.td.gamma * qt[u]);
            if (lane == 0 && a.td.y_out) a.td.y_out[b] = yb[u];
            float mx = -INFINITY;
#pragma unroll
            for (int h = 0; h < 2; ++h) if (lane + 64 * h < A) mx = fmaxf(mx, qv[u][h]);
            mx = dq_wave_max(mx);
            // Q(s0)[a_b] lives in lane a_b & 63, half a_b >> 6 (a_b: one replay row's action, the same in every lane)
            const int ab = __builtin_amdgcn_readfirstlane(a_b[u]);
            const float diff = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ab < 64 ? qv[u][0] : qv[u][1]), ab & 63)) - yb[u];
            loss += 0.5f * diff * diff;
            mq += mx;
            if (!(fabsf(diff * a.td.grad_scale * GS) < 32768.f) && lane == 0) atomicMax(a.skip_word, a.skip_tag);      // (never taken in a healthy run)
            if (a.td.dq_out)
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int c = lane + 64 * h;
                    if (c < A) a.td.dq_out[(size_t)b * A + c] = c == a_b[u] ? (qv[u][h] - yb[u]) * a.td.grad_scale : 0.f;
                }
        }
    }
#pragma unroll
    for (int u = 0; u < RPW; ++u) {
        const int row = wave + DENSE_WAVES * u;
        if (row >= ns) continue;
        const int b = b0 + row;
        const float* dr = a.dq + (size_t)b * A;
        auto dval = [&](int h) {                                    // dq[b][lane + 64 h]
            const int c = lane + 64 * h;
            return TD ? (c == a_b[u] ? (qv[u][h] - yb[u]) * a.td.grad_scale * GS : 0.f) : dr[c] * GS;
        };
        if (N3 > 0) {
            float s = 0.f;
            if constexpr (TD) {                                         // one non-zero per row: its sum is that value (the butterfly's bits: x + 0 ... + 0)
                const int ab = __builtin_amdgcn_readfirstlane(a_b[u]);
                const float qa = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, ab < 64 ? qv[u][0] : qv[u][1]), ab & 63));
                s = (unsigned)ab < (unsigned)A ? (qa - yb[u]) * a.td.grad_scale * GS : 0.f;
            } else {
#pragma unroll
                for (int h = 0; h < 2; ++h) if (lane + 64 * h < A) s += dval(h);      // (a run-time trip count would index qv dynamically: scratch memory)
                for (int m = 32; m >= 1; m >>= 1) s += __shfl_xor(s, m);
            }
            unsigned short* p3 = a.g3_pl + (size_t)b * a.small_ld;   // the same values as pieces: the dueling layer's weight gradient
            const size_t lo3 = (size_t)a.plane_rows * a.small_ld;
            unsigned short ph, pl;
            if (lane == 0) { split_f16x2_one(s, ph, pl); p3[0] = ph; p3[lo3] = pl; }
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                if (c >= A) continue;
                const float dv = dval(h), v = dv - s / (float)A;    // g3 = the dueling layer's output gradient: its weight gradient's operand (pieces, HBM)
                if constexpr (!SHORT) s_g3[row * ldg + c] = dv;     // the dq image: gY2's operand below
                split_f16x2_one(v, ph, pl);
                p3[1 + c] = ph; p3[lo3 + 1 + c] = pl;
            }
            if constexpr (SHORT) {                                  // gY2[row] = dq[row][a_b] * W3'^T[a_b]: lane = column n2 (zero past N2: the table's padding)
                if constexpr (W3LDS) {
                    const int ab = __builtin_amdgcn_readfirstlane(a_b[u]);
                    const float* s_w3t = reinterpret_cast<const float*>(smem + a.off_w3t);
                    const float gy = s_w3t[min(max(ab, 0), PW3 - 1) * PW3 + lane] * s;
                    const _Float16 vh = (_Float16)gy, vl = (_Float16)((gy - (float)vh) * F16_LO_SCALE);      // split on write (qnet.h)
                    s_gy2p[row * LDY + lane] = __builtin_bit_cast(unsigned short, vh);
                    s_gy2p[(DENSE_ROWS + row) * LDY + lane] = __builtin_bit_cast(unsigned short, vl);
                } else {
#pragma unroll
                    for (int h = 0; h < 2; ++h) {
                        const int n2 = lane + 64 * h;
                        if (n2 >= 32 * a.KB2) continue;
                        const float gy = w3r[u][h] * s;
                        const _Float16 vh = (_Float16)gy, vl = (_Float16)((gy - (float)vh) * F16_LO_SCALE);
                        s_gy2p[row * LDY + n2] = __builtin_bit_cast(unsigned short, vh);
                        s_gy2p[(DENSE_ROWS + row) * LDY + n2] = __builtin_bit_cast(unsigned short, vl);
                    }
                }
                // gH1[row] = dq[row][a_b] * Wc[a_b] * [h1 > 0] / (1 - rate): this lane's eight units, split on write into the planes gX reads
                // (LDS) and the weight gradient reads (HBM)
                constexpr int LDH = DENSE_HID + 8;
                u32x4 oh, ol;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const u32 mw = (e < 2 ? (e == 0 ? mkh[u][0] | mkl[u][0] : mkh[u][1] | mkl[u][1]) : (e == 2 ? mkh[u][2] | mkl[u][2] : mkh[u][3] | mkl[u][3]));
                    const float w0 = wcr[u][e >> 1][2 * (e & 1)], w1 = wcr[u][e >> 1][2 * (e & 1) + 1];
                    const float v0 = (mw & 0x7fffu) ? w0 * s * a.mask_scale : 0.f;          // (pieces of a value >= 0: any magnitude bit set <=> it was > 0)
                    const float v1 = (mw & 0x7fff0000u) ? w1 * s * a.mask_scale : 0.f;
                    u32 hq, lq;
                    split_f16x2_pair(v0, v1, hq, lq);
                    oh[e] = hq; ol[e] = lq;
                }
                *reinterpret_cast<u32x4*>(s_gh1p + row * LDH + 8 * lane) = oh;
                *reinterpret_cast<u32x4*>(s_gh1p + (DENSE_ROWS + row) * LDH + 8 * lane) = ol;
                unsigned short* gp = a.gh1_pl + (size_t)b * DENSE_HID + 8 * lane;
                *reinterpret_cast<u32x4*>(gp) = oh;
                *reinterpret_cast<u32x4*>(gp + (size_t)a.plane_rows * DENSE_HID) = ol;
            }
        } else {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                if (c >= A) continue;
                const float v = dval(h);
                const _Float16 vh = (_Float16)v, vl = (_Float16)((v - (float)vh) * F16_LO_SCALE);
                s_gy2p[row * LDY + c] = __builtin_bit_cast(unsigned short, vh);
                s_gy2p[(DENSE_ROWS + row) * LDY + c] = __builtin_bit_cast(unsigned short, vl);
            }
        }
    }
    DQ_STAMP(DQ_TAG_DENSE_BWD, 9);
    if (TD && lane == 0) { s_met[wave][0] = loss; s_met[wave][1] = mq; }
    __syncthreads();
    if (TD && a.td.metrics && tid == 0 && cs == 0) {           // this row tile's partial, and zeros in the slots nobody owns
        float l = 0.f, q = 0.f;
        for (int w = 0; w < DENSE_WAVES; ++w) { l += s_met[w][0]; q += s_met[w][1]; }
        if (a.dense_tiles <= a.td.metric_slots) {
            a.td.metrics[2 + 2 * bt] = l;
            a.td.metrics[3 + 2 * bt] = q;
            for (int k = bt + a.dense_tiles; k < a.td.metric_slots; k += a.dense_tiles) { a.td.metrics[2 + 2 * k] = 0.f; a.td.metrics[3 + 2 * k] = 0.f; }
        } else {                                                    // more row tiles than slots (batch > 16 384): the launcher zeroed the slots
            atomicAdd(a.td.metrics + 2 + 2 * (bt % a.td.metric_slots), l);
            atomicAdd(a.td.metrics + 3 + 2 * (bt % a.td.metric_slots), q);
        }
    }
    DQ_STAMP(DQ_TAG_DENSE_BWD, 1);
    if (GH1_PRE_W == 2) {                                           // gH1's first weight block: in flight under gY2
        const u32x4* pk = a.packed + a.pk_dense2t + (size_t)(4 * wave) * PK_BLOCK + lane;
#pragma unroll
        for (int t = 0; t < 4; ++t) { gh1_pre.bw0[t].h = pk[t * PK_BLOCK]; gh1_pre.bw0[t].l = pk[t * PK_BLOCK + PK_LO]; }
        __builtin_amdgcn_sched_barrier(0);
    }
    // ---- gY2 = dq W3'^T  (K = |A|, one column tile per wave; SHORT: done above, row by row) ---------------------------------------
    if (!SHORT && N3 > 0) {
        if (wave < NT2) {
            const int n2 = 16 * wave + j;
            float (&b)[NT2][4] = w3b;                               // (requested at the top of the kernel; zero-padded)
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
                    const _Float16 vh = (_Float16)acc[r], vl = (_Float16)((acc[r] - (float)vh) * F16_LO_SCALE);      // split on write (qnet.h)
                    s_gy2p[row * LDY + n2] = __builtin_bit_cast(unsigned short, vh);
                    s_gy2p[(DENSE_ROWS + row) * LDY + n2] = __builtin_bit_cast(unsigned short, vl);
                }
            }
        }
        __syncthreads();
    }
    DQ_STAMP(DQ_TAG_DENSE_BWD, 2);
    // ---- gY2's piece planes also leave for the weight gradient of Dense(|A|): 16 rows x 32 KB2 halves per plane, 16 bytes per thread ----
    {
        const int per_row = 4 * a.KB2, n16 = DENSE_ROWS * per_row;  // 16-byte pieces per row / per plane
        const int prs = a.KB2 == 2 ? 3 : 4;                         // (KB2 is 2 or 4: shifts, not divisions)
        for (int i = tid; i < 2 * n16; i += DENSE_THREADS) {
            const int piece = i >> (prs + 4), r = (i >> prs) & (DENSE_ROWS - 1), c8 = i & (per_row - 1);
            if (r < ns)
                *reinterpret_cast<u32x4*>(a.gy2_pl + ((size_t)piece * a.plane_rows + b0 + r) * a.small_ld + 8 * c8) =
                    *reinterpret_cast<const u32x4*>(s_gy2p + (piece * DENSE_ROWS + r) * LDY + 8 * c8);
        }
    }
    // ---- gH1 (f16x2; K = N2 in 2 or 4 blocks) ---------------------------------------------------------------------------------------
    if constexpr (!SHORT) {                                         // (SHORT: every wave wrote its rows above)
        if (a.KB2 == 2) gh1_phase<2>(a, s_gy2p, s_gh1p, b0, ns, wave, lane, gh1_pre);       // block-uniform
        else gh1_phase<4>(a, s_gy2p, s_gh1p, b0, ns, wave, lane, gh1_pre);
    }
    DQ_STAMP(DQ_TAG_DENSE_BWD, 3);
    __syncthreads();
    DQ_STAMP(DQ_TAG_DENSE_BWD, 4);
    // ---- gX = (gH1 W1T) * [x > 0]  (K = 512): each wave owns a run of adjacent column tiles (counts differ by at most one, the
    //      longer runs on different SIMDs), taken up to three at a time with interleaved columns ---------------------------------
    {
        const int all = a.K1 >> 4, per = (all + a.col_split - 1) / a.col_split, first = cs * per;      // this workgroup's column tiles [first, first + tiles)
        const int tiles = max(0, min(per, all - first)), base = tiles / DENSE_WAVES, extra = tiles - base * DENSE_WAVES;
        int t0 = first + wave * base + min(wave, extra), left = base + (wave < extra ? 1 : 0);
        while (left > 0) {                                          // wave-uniform
            if (left >= 3) { gx_pass<3>(a, s_gh1p, t0, b0, ns, lane); t0 += 3; left -= 3; }
            else if (left == 2) { gx_pass<2>(a, s_gh1p, t0, b0, ns, lane); t0 += 2; left -= 2; }
            else { gx_pass<1>(a, s_gh1p, t0, b0, ns, lane); t0 += 1; left -= 1; }
        }
    }
    DQ_STAMP(DQ_TAG_DENSE_BWD, 5);
    DQ_STAMP_PAIR2(0);
    DQ_STAMP_ALL(1);
}

// ---------------------------------------------------------------------------------------------------------------
// Weight gradients of the dense layers, all layers in ONE launch.  dW[k, n] = sum_b X[b, k] G[b, n]: the batch is the reduction index.
// Workgroup = 4 waves = one 64 x 64 output tile over one batch slice; wave w owns the 32 x 32 quarter (k half w >> 1, n half w & 1) as
// 2 x 2 MFMA tiles.  Every operand (Dense(512)'s input x and its gradient gH1, the hidden output h1, gY2, y2, g3) arrives as ready-made
// row-major f16 piece planes, written by the kernels that produced it (fused.hip dense_chain_kernel, dense_bwd_chain_kernel above).  An
// iteration's 64 batch rows x 64 columns of both operands and pieces are copied global -> LDS by LDS-DMA (no registers, no arithmetic;
// rows of 64 halves + one 16-byte padding slot whose lane stays inactive), and the MFMA operands -- 8 consecutive batch rows of one
// column per lane: the TRANSPOSE of the staged rows -- come out of LDS by transposing reads (lds_tr8 above: two ds_read_b64_tr_b16 per
// piece and tile).  Double-buffered, one barrier per iteration, the next iteration's copies fly under this iteration's MFMAs (since round 3 for real:
// lds_dma16, qnet.h -- 15.5 -> 13.7 us; the nine copies of an iteration issued BETWEEN the MFMA groups instead of in one run before them: 15.1 us, dropped).
// (Before: rows loaded into registers, transposed with 32 byte-permutes per lane and 8 rows, stored into a swizzled [column][batch row]
// image and read back with ds_read_b128: 155 VALU per wave and iteration, 16.4 us; now 15.1 us.)
// Bias gradients (column sums of G) come out of the matrix pipe as well: a tile whose A operand is all ones.  Rows past the slice are
// re-reads of its last row, and zeroed in G only (a zero factor kills the product); columns past K / N are clamped too and only reach
// accumulators that are never stored.
#define WGRAD_THREADS 256
#ifndef DQ_RIDE_WGRAD_DEFAULT
#define DQ_RIDE_WGRAD_DEFAULT 0        // 1: the environment step rides on the dense weight gradients' launch by default (fused_rider_threads)
#endif
#define WGRAD_WAVES 4
#define WG_PS 72                        // halves per LDS row of an operand image: 64 columns + one 16-byte padding slot (rows 36 dwords apart: the four rows of
                                        // a transposing read fall into distinct banks)
#ifndef WG_ROWS
#define WG_ROWS 64                      // batch rows per iteration (64: 72 KB of LDS per workgroup = two workgroups per CU; 32 -- four per CU -- measured 12 % slower)
#endif
#define WG_NCH ((WG_ROWS * 9 + 63) / 64)    // LDS-DMA instructions (1 KB chunks of 64 sixteen-byte slots) per image
#define WG_NK ((4 * WG_NCH + 3) / 4)    // ... per wave and iteration
#define WG_IMG (WG_ROWS * WG_PS)        // halves per image (WG_ROWS rows of one operand and piece)
#define WG_BUF (4 * WG_IMG)             // ... per buffer: [operand][piece]
#ifndef WG_NBUF
#define WG_NBUF 2                       // operand buffers per workgroup: WG_NBUF - 1 iterations' copies in flight under the one being multiplied (2: two workgroups per CU;
                                        // 4 -- 144 KB, one workgroup per CU, three iterations in flight -- with WG_SLICES_TARGET 5: round 6's experiment, NOTEBOOK.md)
#endif
#define DENSE_WGRAD_LDS ((WG_NBUF * WG_BUF * 2) > 64 * 68 * 4 ? (WG_NBUF * WG_BUF * 2) : 64 * 68 * 4)      // the buffers (the finished tile is staged over them)
static_assert(DENSE_WGRAD_LDS <= CHAIN_LDS_MAX && (WG_NBUF - 2) * WG_NK <= 63, "LDS budget / s_waitcnt vmcnt range");

struct WgradOperand {
    const float* f32;                   // [batch, cols] f32 (+ 16 bytes of slack), or NULL when the operand comes as piece planes:
    const unsigned short* planes;       // h plane [plane_rows][cols] then, plane_stride halves further, the l plane (cols a multiple of 8)
    size_t plane_stride;
    int cols, ld;                       // columns that exist / halves (floats) per row
};

struct WgradLayer {
    WgradOperand X;                     // the layer's input in the training forward
    WgradOperand G;                     // gradient w.r.t. the layer's pre-activation output
    int out_w, out_b;                   // offsets into a partial (floats)
    int perm_hw, perm_c;                // > 0: column idx = p*perm_c + c of X is weight row c*perm_hw + p (Keras Flatten)
    int tile0, k_tiles, n_tiles;        // first tile id of this layer, tiling in 64 x 64 tiles
};

struct DenseWgradArgs {
    WgradLayer L[3];
    int n_layers, batch, rows_per_slice, total_tiles, slices;
    float* partial;                     // [slices][pstride]
    size_t pstride;
    int env_on, wg_count, env_wgs;               // env_on: workgroups >= wg_count run the vector step's environment launch (DQ_RIDE_ON=wgrad: env_block<4> / env_block2<8>)
};

// Where the riding environment step rides (round 4): on the dense data gradients' launch (rounds 2-3; DQ_RIDE_ON=dense_bwd) or on the dense weight
// gradients' (DQ_RIDE_ON=wgrad): that launch has two 4-wave workgroups per CU at 121 registers -- issue slots and registers to spare -- while the data
// gradients' workgroups are a latency chain that the riders lengthen (16.3 us alone, 20.2 with them, one box).
static int ride_on_wgrad() {            // 0: dense_bwd_chain's launch; 1: behind the weight-gradient tiles; 2: in front of them
    static const int on_wgrad = !getenv("DQ_RIDE_ON") ? DQ_RIDE_WGRAD_DEFAULT : strcmp(getenv("DQ_RIDE_ON"), "wgrad") == 0 ? 1 : strcmp(getenv("DQ_RIDE_ON"), "wgrad_first") == 0 ? 2 : 0;
    return on_wgrad;
}
int fused_rider_threads() { return ride_on_wgrad() ? 256 : 512; }

// rows per batch slice (a multiple of 64: whole iterations) and the number of slices, at most DENSE_WGRAD_SLICES.  WG_SLICES_TARGET slices are aimed
// for: at c3 the launch is 49 tiles x slices workgroups on 256 CUs with room for two each -- 8 slices = 392 workgroups leave 120 CUs with one
// workgroup and 136 with two (which set the kernel's duration: 8 iterations each); 10 slices of 448 rows = 490 workgroups of 7 iterations fill
// nearly every CU twice.
#define DENSE_WGRAD_SLICES 16
#ifndef WG_SLICES_TARGET
#define WG_SLICES_TARGET 10
#endif
static void wgrad_slicing(const dq_qnet* Q, int B, int* rows_per_slice, int* slices) {
    // as many slices as fill every CU twice -- 512 workgroup slots / the layers' 64 x 64 tiles: c3's 49 tiles -> 10 (WG_SLICES_TARGET, measured there);
    // a wider first dense layer (d = 7: 122 tiles) takes fewer, so that the launch stays ONE round of workgroups and writes fewer partials
    static const int forced = getenv("DQ_WGRAD_SLICES") ? atoi(getenv("DQ_WGRAD_SLICES")) : 0;      // (A/B runs)
    int tiles = 0;
    for (int l = Q->cfg.n_conv; l < Q->n_layers; ++l) tiles += ((Q->L[l].K + 63) / 64) * ((Q->L[l].N + 63) / 64);
    int target = tiles > 0 ? 512 / tiles : WG_SLICES_TARGET;
    if (target > WG_SLICES_TARGET) target = WG_SLICES_TARGET;
    if (target < 2) target = 2;
    if (forced >= 1 && forced <= DENSE_WGRAD_SLICES) target = forced;
    int rps = (B + target - 1) / target;
    rps = (rps + 63) & ~63;
    while ((B + rps - 1) / rps > DENSE_WGRAD_SLICES) rps += 64;
    *rows_per_slice = rps;
    *slices = (B + rps - 1) / rps;
}

// (wblock: the workgroup's index among the gradient tiles' -- blockIdx.x, or less the riding environment blocks in front of them)
__device__ __forceinline__ void dense_wgrad_body(const DenseWgradArgs& a, const int wblock, u8* smem) {
    unsigned short* s_t = reinterpret_cast<unsigned short*>(smem);   // [buf][op][piece][64 rows][WG_PS] row-major piece planes
    // XCD-aware block -> (tile, slice) map: workgroup b runs on XCD b % 8 and each XCD has its own L2, so all tiles of one batch
    // slice are given to ONE XCD (slice = XCD + 8 i): the slice's rows of X and G are then fetched from HBM/MALL once instead of
    // once per XCD.
    // Whole groups of 8 slices are mapped that way; the slices left over (10 slices: two) go round the XCDs tile by tile, so that every XCD gets the
    // same number of workgroups.
    int slice, tile;
    {
        const int full = (a.slices >> 3) * 8 * a.total_tiles;       // workgroups of the whole groups of 8 slices
        if (wblock < full) {
            const int xcd = wblock & 7, within = wblock >> 3;
            slice = xcd + 8 * (within / a.total_tiles); tile = within % a.total_tiles;
        } else {
            const int e = wblock - full;
            slice = (a.slices & ~7) + e / a.total_tiles; tile = e % a.total_tiles;
        }
    }
    if (slice >= a.slices) return;                                  // block-uniform
    int l = 0;
    while (l + 1 < a.n_layers && tile >= a.L[l + 1].tile0) ++l;     // block-uniform
    const WgradLayer& L = a.L[l];
    const int local = tile - L.tile0, kt = local / L.n_tiles, nt = local - kt * L.n_tiles;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kb = lane >> 4;
    const int K = L.X.cols, N = L.G.cols, kbase = 64 * kt, nbase = 64 * nt;
    const int m0 = slice * a.rows_per_slice, m1 = min(a.batch, m0 + a.rows_per_slice);
    const int n_it = (m1 - m0 + WG_ROWS - 1) / WG_ROWS;
    float* out = a.partial + (size_t)slice * a.pstride;
    float* s_o = reinterpret_cast<float*>(smem);                    // the finished tile [64][68], staged for row-wise stores
    DQ_STAMP(DQ_TAG_DENSE_WGRAD, 0);
    DQ_STAMP_WG(DQ_TAG_DENSE_WGRAD, 0);
    DQ_STAMP_PAIR2(1);
    // ---- staging: an iteration's WG_ROWS rows x 64 columns of both operands and both pieces = 4 images of WG_ROWS x 9 sixteen-byte slots (8 of data +
    //      1 of padding, whose lane stays inactive) = 4 x WG_NCH LDS-DMA instructions of 1 KB, WG_NK per wave: instruction c = wave + 4k is
    //      chunk c % WG_NCH of image c / WG_NCH (operand = image >> 1, piece = image & 1).  Per-lane constants of the wave's nine instructions: the row inside
    //      the iteration and the global column offset (columns past the operand's row are clamped to 0: they only reach accumulators that
    //      are never stored); -1 = padding slot.
    int drow[WG_NK];
    const unsigned short* dsrc[WG_NK];
#pragma unroll
    for (int k = 0; k < WG_NK; ++k) {
        const int c = wave + 4 * k, img = min(c / WG_NCH, 3), ch = c - (c / WG_NCH) * WG_NCH, q = ch * 64 + lane, row = q / 9, part = q - row * 9;
        const WgradOperand& O = (img >> 1) ? L.G : L.X;
        const int col = ((img >> 1) ? nbase : kbase) + 8 * part;
        drow[k] = (part < 8 && row < WG_ROWS && c < 4 * WG_NCH) ? row : -1;
        dsrc[k] = O.planes + (size_t)(img & 1) * O.plane_stride + (col < O.ld ? col : 0);
    }
    const int ldx = L.X.ld, ldg = L.G.ld;
    auto issue = [&](int it) {
        unsigned short* dst = s_t + (it % WG_NBUF) * WG_BUF;
#pragma unroll
        for (int k = 0; k < WG_NK; ++k) {
            const int c = wave + 4 * k, img = min(c / WG_NCH, 3), ch = c - (c / WG_NCH) * WG_NCH;      // wave-uniform
            if (drow[k] >= 0) {
                const size_t ro = (size_t)min(m0 + WG_ROWS * it + drow[k], m1 - 1) * ((img >> 1) ? ldg : ldx);      // rows past the slice: re-read its last row (G cleared in mm)
                lds_dma16(dsrc[k] + ro, lds_addr(dst + img * WG_IMG + ch * 512));      // (qnet.h: in flight until the loop's own s_waitcnt vmcnt(0))
            }
        }
    };
    // ---- operands: row-major piece planes in LDS -> MFMA operand order by transposing reads (lds_tr8: this lane points at row ri, column
    //      segment cseg of each four-row read; lane group kb supplies rows (kb >> 1) + 8 (kb & 1) + 2 (e & 3) + 16 (e >> 2) of a 32-row block) ------------------
    const int ri = j >> 2, cseg = 4 * (j & 3);
    f32x4 acc[2][2], accx[2][2], accb[2], accbx[2];                 // [k tile][n tile]: leading products / 2^11-scaled cross terms; bias tiles
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) { acc[ta][tb] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[ta][tb] = acc[ta][tb]; }
#pragma unroll
    for (int tb = 0; tb < 2; ++tb) { accb[tb] = f32x4{0.f, 0.f, 0.f, 0.f}; accbx[tb] = accb[tb]; }
    const u32x4 ones = {0x3c003c00u, 0x3c003c00u, 0x3c003c00u, 0x3c003c00u};     // f16 1.0 x 8
    auto mm = [&](int it) {                                         // the iteration's two blocks: 16 transposing reads + 16 MFMAs each
        const unsigned short* base = s_t + (it % WG_NBUF) * WG_BUF;
        const bool tail = m0 + WG_ROWS * it + WG_ROWS > m1;         // wave-uniform: only the slice's last iteration masks rows
#pragma unroll
        for (int blk = 0; blk < WG_ROWS / 32; ++blk) {
            // (rows of a block dealt so that the eight rows a 32-lane group of a transposing read touches have ONE parity: rows 144 bytes apart, eight same-parity
            // rows put their 32-byte windows on disjoint quarters of the banks -- conv_bwd16.hip; eight consecutive rows overlapped pairwise)
            const int r0 = (32 * blk + (kb >> 1) + 8 * (kb & 1) + 2 * ri) * WG_PS, r1 = r0 + 16 * WG_PS;
            F16x2 xa[2], gb[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                const int cx = 32 * (wave >> 1) + 16 * t + cseg, cg = 32 * (wave & 1) + 16 * t + cseg;
                xa[t] = lds_tr8(base + r0 + cx, base + r1 + cx, WG_IMG);
                gb[t] = lds_tr8(base + 2 * WG_IMG + r0 + cg, base + 2 * WG_IMG + r1 + cg, WG_IMG);
            }
            if (tail) {                                             // rows of G past the slice: their halves cleared (element e = half e of the operand)
#pragma unroll
                for (int d = 0; d < 4; ++d) {
                    const int rb = m0 + WG_ROWS * it + 32 * blk + (kb >> 1) + 8 * (kb & 1);
                    const u32 lo = rb + 2 * ((2 * d) & 3) + 16 * ((2 * d) >> 2) < m1 ? 0xffffu : 0u;
                    const u32 hi = rb + 2 * ((2 * d + 1) & 3) + 16 * ((2 * d + 1) >> 2) < m1 ? 0xffff0000u : 0u;
#pragma unroll
                    for (int t = 0; t < 2; ++t) { gb[t].h[d] &= lo | hi; gb[t].l[d] &= lo | hi; }
                }
            }
#pragma unroll
            for (int ta = 0; ta < 2; ++ta)
#pragma unroll
                for (int tb = 0; tb < 2; ++tb) mma_f16x3(xa[ta], gb[tb], acc[ta][tb], accx[ta][tb]);
            // column sums of G (every wave: a condition around an MFMA makes hipcc copy accumulators; only k-tile 0 stores them)
#pragma unroll
            for (int tb = 0; tb < 2; ++tb) { accb[tb] = MFMA_F16(ones, gb[tb].h, accb[tb]); accbx[tb] = MFMA_F16(ones, gb[tb].l, accbx[tb]); }
        }
    };
    // one step: this wave's copies of iteration `it` have landed; meet (everybody's have, and everybody has left the other buffer); request
    // iteration it + 1 into that buffer; multiply
    // (WG_NBUF buffers: the copies of iterations it + 1 .. it + WG_NBUF - 2 may stay in flight across the wait for iteration it's -- a wave's copies land in order,
    // WG_NK instructions per iteration; the count is an immediate, so the slice's last iterations, with fewer behind them, take their own)
#pragma unroll
    for (int u = 0; u < WG_NBUF - 1; ++u) if (u < n_it) issue(u);
    for (int it = 0; it < n_it; ++it) {
        DQ_STAMP(DQ_TAG_DENSE_WGRAD, 1 + 3 * min(it, 7));
        const int behind = min(n_it - 1 - it, WG_NBUF - 2);         // wave-uniform
        if (WG_NBUF >= 4 && behind == 2) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(2 * WG_NK) : "memory");
        else if (WG_NBUF >= 3 && behind == 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"(WG_NK) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        DQ_STAMP(DQ_TAG_DENSE_WGRAD, 2 + 3 * min(it, 7));
        if (it + WG_NBUF - 1 < n_it) issue(it + WG_NBUF - 1);
        DQ_STAMP(DQ_TAG_DENSE_WGRAD, 3 + 3 * min(it, 7));
        mm(it);
    }
    DQ_STAMP(DQ_TAG_DENSE_WGRAD, 25);
    // ---- the tile goes through LDS (C/D layout: this lane holds rows 4kb + r of column j of each 16 x 16 tile) so that it leaves as
    //      whole 256-byte weight rows ---------------------------------------------------------------------------------------------
    __syncthreads();                                                // every wave is done with the operand planes
#pragma unroll
    for (int ta = 0; ta < 2; ++ta)
#pragma unroll
        for (int tb = 0; tb < 2; ++tb)
#pragma unroll
            for (int r = 0; r < 4; ++r)
                s_o[(32 * (wave >> 1) + 16 * ta + 4 * kb + r) * 68 + 32 * (wave & 1) + 16 * tb + j] = f16x2_sum(acc[ta][tb][r], accx[ta][tb][r]);
    if (kt == 0 && (wave >> 1) == 0 && kb == 0) {                   // bias gradient: row 0 of the all-ones tiles
#pragma unroll
        for (int tb = 0; tb < 2; ++tb) {
            const int n = nbase + 32 * (wave & 1) + 16 * tb + j;
            if (n < N) out[L.out_b + n] = f16x2_sum(accb[tb][0], accbx[tb][0]);
        }
    }
    // ---- one partial per (slice, tile) ------------------------------------------------------------------------------------------
    __syncthreads();
    {
        const int c4 = 4 * (tid & 15);
        const bool nvec = (N & 3) == 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int kr = (tid >> 4) + 16 * q, k = kbase + kr;
            if (k >= K) continue;
            int row = k;
            if (L.perm_hw > 0) { const int pp = k / L.perm_c, c = k - pp * L.perm_c; row = c * L.perm_hw + pp; }
            const f32x4 v = *reinterpret_cast<const f32x4*>(s_o + kr * 68 + c4);
            float* o = out + L.out_w + (size_t)row * N + nbase + c4;
            if (nvec) { if (nbase + c4 < N) *reinterpret_cast<f32x4u*>(o) = v; }
            else {
#pragma unroll
                for (int c = 0; c < 4; ++c) if (nbase + c4 + c < N) o[c] = v[c];
            }
        }
    }
    DQ_STAMP(DQ_TAG_DENSE_WGRAD, 26);
    DQ_STAMP_WG(DQ_TAG_DENSE_WGRAD, 1);
    DQ_STAMP_PAIR2(2);
}

__global__ __launch_bounds__(WGRAD_THREADS, WG_ROWS == 32 ? 4 : WG_NBUF > 2 ? 1 : 2) void dense_wgrad_kernel(DenseWgradArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    dense_wgrad_body(a, (int)blockIdx.x, smem);
}

// DQ_RIDE_ON=wgrad / wgrad_first (A/B runs; NOTEBOOK.md, Round 4 section 9: measured slower): the riding environment step (env_dev.h, 256 threads per
// block) behind the gradient tiles (env_on 1) or in front of them (env_on 2); block-uniform.  A kernel of its own: the default launch keeps its
// registers and its kernel-argument segment.
__global__ __launch_bounds__(WGRAD_THREADS, WG_ROWS == 32 ? 4 : WG_NBUF > 2 ? 1 : 2) void dense_wgrad_ride_kernel(DenseWgradArgs a, EnvParams env) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int wblock = (int)blockIdx.x - (a.env_on == 2 ? a.env_wgs : 0);
    if (a.env_on == 2 ? wblock < 0 : wblock >= a.wg_count) {
        const int eb = a.env_on == 2 ? (int)blockIdx.x : wblock - a.wg_count;
        if (env.pair) env_block2<8>(env, eb, smem);
        else env_block<4>(env, eb, smem);
        return;
    }
    dense_wgrad_body(a, wblock, smem);
}

// Fixed-order reduction of both partial sets in one launch: out[i] = sum_s partial[s * stride + i].
// Block = 64 outputs x 8 slice groups (thread (x, g) sums slices g, g+8, ... in order; the 8 group sums are combined
// pairwise in LDS) -- deterministic, and 8x the loads in flight of a one-thread-per-output loop.
struct ReduceSeg { const float* partial; float* out; int n, slices; size_t stride; int block0; int pidx0; int vec; };   // pidx0: flat index of out[0]; vec: see the kernel
// the hidden layer's dropout keep bits of the NEXT training forward, drawn by the first `wgs` workgroups of this launch (qnet.h keep_bits)
struct DropAhead { u32* bits; u32 seed0, seed1, sample_base, drop_T; u64 t; int batch, wgs; };
struct ReduceArgs { ReduceSeg seg[2]; AdamOpt opt; int adam; float inv_gs; const float* gs_dev; unsigned* range_flag; DropAhead drop;
                    const unsigned* skip_word; unsigned skip_tag; };     // partials carry the gradient scale: x 1/S; skip: DenseBwdArgs.skip_word

// Thread g of the drawing workgroups: word g & 15 of sample g >> 4 = the 32 units 32 (g & 15) .. + 31 = four Philox calls of eight 16-bit draws --
// fused.hip dense_chain_kernel's draw (unit n: half-word n & 7 of call n >> 3, kept iff >= drop_T), the same bits.
__device__ __forceinline__ void dropout_ahead(const DropAhead& d, int g) {
    const int row = g >> 4, word = g & 15;
    if (row >= d.batch) return;
    u32 bits = 0;
#pragma unroll
    for (int c = 0; c < 4; ++c) {
        u32 wd[4];
        philox4x32_10((u32)d.t, (u32)(d.t >> 32), d.sample_base + (u32)row, (u32)(4 * word + c) | ((u32)DQ_STREAM_DROPOUT << 16), d.seed0, d.seed1, wd);
#pragma unroll
        for (int e = 0; e < 8; ++e) bits |= (((wd[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= d.drop_T ? 1u : 0u) << (8 * c + e);
    }
    d.bits[(size_t)row * 16 + word] = bits;
}

// one output's optimizer step and range guard
__device__ __forceinline__ void reduce_finish(const ReduceArgs& a, const ReduceSeg& S, int i, float gsum) {
    // (the dense backward's TD step found a sample beyond the safe range: the whole update is discarded, on every rank once the NaNs are all-reduced)
    if (a.skip_word && __builtin_nontemporal_load(a.skip_word) == a.skip_tag) gsum = __builtin_nanf("");
    S.out[i] = gsum;
    // range guard (dq_qnet_range_check): an S x gradient beyond the f16 pieces' range arrives here as inf / NaN -- reported, never applied
    const bool finite = fabsf(gsum) < INFINITY;
    if (!finite) atomicOr(a.range_flag, 1u);
    if (a.adam && finite) {                                         // the optimizer step rides on the reduction (dq_qnet_backward_adam)
        const size_t k = (size_t)S.pidx0 + i;
        float pk = a.opt.p[k], mk = a.opt.m[k], vk = a.opt.v[k];
        dq_adam1(pk, gsum, mk, vk, a.opt.lr_t, a.opt.b1, a.opt.b2, a.opt.eps);
        a.opt.p[k] = pk; a.opt.m[k] = mk; a.opt.v[k] = vk;
    }
}

// four consecutive outputs (i a multiple of 4, pidx0 too, 16-byte-aligned buffers): gradient, parameters and moments as 16-byte accesses
__device__ __forceinline__ void reduce_finish4(const ReduceArgs& a, const ReduceSeg& S, int i, const f32x4& g4) {
    const bool skip = a.skip_word && __builtin_nontemporal_load(a.skip_word) == a.skip_tag;      // (the update is being discarded: the scalar form writes the NaNs)
    if (i + 3 < S.n && !skip) {
        *reinterpret_cast<f32x4*>(S.out + i) = g4;
        bool finite = true;
#pragma unroll
        for (int c = 0; c < 4; ++c) finite = finite && fabsf(g4[c]) < INFINITY;
        if (!finite) {                                              // (rare: per element, as the scalar form does)
#pragma unroll
            for (int c = 0; c < 4; ++c) reduce_finish(a, S, i + c, g4[c]);
        } else if (a.adam) {
            const size_t k = (size_t)S.pidx0 + i;
            f32x4 pk = *reinterpret_cast<const f32x4*>(a.opt.p + k), mk = *reinterpret_cast<const f32x4*>(a.opt.m + k), vk = *reinterpret_cast<const f32x4*>(a.opt.v + k);
#pragma unroll
            for (int c = 0; c < 4; ++c) { float p1 = pk[c], m1 = mk[c], v1 = vk[c]; dq_adam1(p1, g4[c], m1, v1, a.opt.lr_t, a.opt.b1, a.opt.b2, a.opt.eps); pk[c] = p1; mk[c] = m1; vk[c] = v1; }
            *reinterpret_cast<f32x4*>(a.opt.p + k) = pk; *reinterpret_cast<f32x4*>(a.opt.m + k) = mk; *reinterpret_cast<f32x4*>(a.opt.v + k) = vk;
        }
    } else {
#pragma unroll
        for (int c = 0; c < 4; ++c)
            if (i + c < S.n) reduce_finish(a, S, i + c, g4[c]);
    }
}

__global__ __launch_bounds__(512) void reduce_slices_kernel(ReduceArgs a) {
    __shared__ float sh[8][64];
    // this launch waits for HBM with its vector ALUs idle: its FIRST workgroups (dispatched first, running beside the reduction's) draw the next
    // training forward's dropout keep bits
    if ((int)blockIdx.x < a.drop.wgs) { dropout_ahead(a.drop, (int)blockIdx.x * 512 + threadIdx.y * 64 + threadIdx.x); return; }
    const unsigned bx = blockIdx.x - (unsigned)a.drop.wgs;
    // the optimizer step rides on this launch and the TD step discarded the whole update: counted once (dq_qnet_range_discarded; with a separate optimizer
    // step -- several GPUs -- dqn.hip adam_kernel counts it, on every rank)
    if (bx == 0 && threadIdx.x == 0 && threadIdx.y == 0 && a.adam && a.skip_word && __builtin_nontemporal_load(a.skip_word) == a.skip_tag) atomicAdd(a.range_flag + 4, 1u);
    const ReduceSeg& S = a.seg[bx >= (unsigned)a.seg[1].block0 ? 1 : 0];
    const float inv = a.gs_dev ? a.gs_dev[1] : a.inv_gs;
    if (S.vec == 1) {
        // at most 16 slices (the dense partials): thread = four consecutive outputs, all slices' 16-byte loads in flight together, summed in the
        // scalar form's order (the same bits).  That form spent a 512-thread workgroup, a barrier and an LDS
        // round trip on 64 outputs of 8 loads each: 2763 of this launch's 3020 workgroups.
        const int i = 4 * ((bx - S.block0) * 512 + threadIdx.y * 64 + threadIdx.x);      // block-uniform branch; slices start on 128-byte lines
        if (i >= S.n) return;
        f32x4 v[16];
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const float* p = S.partial + (size_t)min(k, S.slices - 1) * S.stride + i;
            if (i + 3 < S.n) v[k] = *reinterpret_cast<const f32x4*>(p);
            else { v[k] = f32x4{p[0], i + 1 < S.n ? p[1] : 0.f, i + 2 < S.n ? p[2] : 0.f, 0.f}; }
            if (k >= S.slices) v[k] = f32x4{0.f, 0.f, 0.f, 0.f};
        }
        // the scalar form's order (the phased backward of the several-GPU path reduces through it and must give the same bits): slice group g =
        // slices g, g + 8 in turn, then the groups pairwise
#pragma unroll
        for (int k = 0; k < 8; ++k) v[k] = v[k] + v[k + 8];
        const f32x4 g4 = (((v[0] + v[1]) + (v[2] + v[3])) + ((v[4] + v[5]) + (v[6] + v[7]))) * inv;
        reduce_finish4(a, S, i, g4);
        return;
    }
    // (the same treatment of the 256 convolutional partials -- 16 x 32 threads, four outputs each, eight 16-byte loads in flight per thread -- measured
    // slower: 7.7 against 6.6 us for the launch)
    const int i = (bx - S.block0) * 64 + threadIdx.x, g = threadIdx.y;
    float s0 = 0.f, s1 = 0.f, s2 = 0.f, s3 = 0.f;                     // four independent chains: the loop is load-latency-bound
    // (all 32 loads of the 256-slice case in flight at once -- the loop unrolled -- measured slower: 8.9 against 7.6 us)
    if (i < S.n) {
        int k = g;
        for (; k + 24 < S.slices; k += 32) {
            s0 += S.partial[(size_t)k * S.stride + i];
            s1 += S.partial[(size_t)(k + 8) * S.stride + i];
            s2 += S.partial[(size_t)(k + 16) * S.stride + i];
            s3 += S.partial[(size_t)(k + 24) * S.stride + i];
        }
        for (; k < S.slices; k += 8) s0 += S.partial[(size_t)k * S.stride + i];
    }
    sh[g][threadIdx.x] = (s0 + s1) + (s2 + s3);
    __syncthreads();
    if (g == 0 && i < S.n) {
        const int x = threadIdx.x;
        reduce_finish(a, S, i, (((sh[0][x] + sh[1][x]) + (sh[2][x] + sh[3][x])) + ((sh[4][x] + sh[5][x]) + (sh[6][x] + sh[7][x]))) * inv);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Convolutional backward chain: both data gradients AND all three weight gradients of the convolutions in one persistent
// launch.  Workgroup = 8 waves; it loops over groups of S samples.  Per group the saved activations a1 (conv1 output), a2
// (conv2 output), the uint8 observations and the gradient g3 w.r.t. conv3's output are staged in LDS, then
//   dW3 += im2col(a2)^T g3          g2 = (g3 (*) W3^T) * [a2 > 0]  -- written IN PLACE over a2
//   dW2 += im2col(a1)^T g2          g1 = (g2 (*) W2^T) * [a1 > 0]  -- written IN PLACE over a1
//   dW1 += im2col(obs)^T g1
// so no convolutional gradient ever touches HBM.  Weight-gradient accumulators stay in registers across the workgroup's
// groups; each workgroup writes ONE partial at the end (reduced in fixed order by reduce_slices_kernel).
// Everything runs on the f16 pipe (f16x2, qnet.h).  Data gradients: one tap = one K = 32 block, A = the 32 channels of g at the tap's
// pixel (two ds_read_b128, split on the fly), B = the tap's packed weight pieces (registers), a zero row standing in for out-of-range
// taps.  Weight gradients reduce over pixels, so a lane supplies 8 ROWS of its column of each operand per K = 32 block: eight
// ds_read_b32 per operand tile (the rows of a block are assigned to (lane group, element) so that the two 16-lane groups of a
// ds_read_b32 lane group fall 16 banks apart), then one split per operand tile; the LDS reads of block t + 1 are issued before the splits
// and MFMAs of block t; blocks that lie inside the group take a mask-free path whose addresses are affine (immediate offsets).
#define CB_THREADS 512
#define CB_WAVES 8


// Copies `rows` rows of CH floats from global memory into an LDS image with row stride PS.  Loads are issued NB at a time before
// the first store: a load -> store loop body costs one full memory latency per trip.
template <int CH, int PS>
__device__ __forceinline__ void stage_rows(float* __restrict__ dst, const float* __restrict__ src, int rows, int tid) {
    constexpr int Q4 = CH / 4, NB = 4;
    const int total = rows * Q4;
    for (int base = 0; base < total; base += NB * CB_THREADS) {
        f32x4 v[NB];
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = base + u * CB_THREADS + tid;
            v[u] = *reinterpret_cast<const f32x4*>(src + (size_t)(i < total ? i : 0) * 4);      // contiguous source; clamped, unconditional
        }
#pragma unroll
        for (int u = 0; u < NB; ++u) {
            const int i = base + u * CB_THREADS + tid;
            if (i < total) {
                const int r = i / Q4, c4 = (i - r * Q4) * 4;
                *reinterpret_cast<f32x4*>(dst + r * PS + c4) = v[u];
            }
        }
    }
}

// Weights of one 2x2 data gradient for this lane as f16 pieces: [tap ky*2+kx][column tile t]: B(n = 8kb .. 8kb+7, c = c_lo + 16t + j)
// = W[ky,kx, c, n], ready-made in the packed buffer (qnet.h PK_CONV3_DG / PK_CONV2_DG + 8 * PK_BLOCK * half).
__device__ __forceinline__ void dgrad_load_w(F16x2 (&bw)[4][2], const u32x4* __restrict__ pk, int lane) {
    // opaque base: otherwise hipcc hoists the load addresses out of the caller's loop as invariants (VGPR pairs), spills them, and
    // reloads each behind an s_waitcnt vmcnt(0) -- which serialises the loads (measured: 10K cycles for this function)
    pk = opaque_global(pk);
#pragma unroll
    for (int tap = 0; tap < 4; ++tap)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            const u32x4* pb = pk + (tap * 2 + t) * PK_BLOCK + lane;
            bw[tap][t].h = pb[0]; bw[tap][t].l = pb[PK_LO];
        }
}

// Data gradient of a 2x2 stride-1 convolution with 32 output channels, masked by the input activation, in place:
//   act[(s,iy,ix), c] <- (sum_{ky,kx,n} g[(s,iy-ky,ix-kx), n] W[ky,kx,c,n]) * [act > 0]     for c in [c_lo, c_lo + 32)
// One tap = one K = 32 block of the f16 MFMA: A = the 32 channels of g at the tap's pixel, B = the tap's weights (registers).
//   GPL: g is a piece-plane image [rows][PL32] (l plane g_lo halves further): A = one ds_read_b128 per piece; else an f32 image
//        [rows][36], split on the fly.  Both have an all-zero row at index `zero_row`.
//   OPL: act is a piece-plane image [pixels][PSA halves] (l plane act_lo halves further): the mask is "a magnitude bit in either
//        piece", the result is split on write -- same address as the activation it replaces, so in place; else f32 [pixels][PSA].
// dtab[m] (LDS, host-built): row of g under input pixel m's own position, s * oh * ow + iy * ow + ix, | iy << 16 | ix << 24.
// colsum (OPL only): this lane's running column sums of the result (columns c_lo + 2j, c_lo + 2j + 1) -- the bias gradient of the layer below.
template <int PSA, bool GPL, bool OPL>
__device__ __forceinline__ void dgrad_inplace(const F16x2 (&bw)[4][2], const void* __restrict__ gv, int g_lo, int zero_row, void* __restrict__ actv,
                                              int act_lo, int c_lo, const int* __restrict__ dtab, int oh, int ow, int M, int tile_first,
                                              int tile_step, int lane, float (&colsum)[2]) {
    const int j = lane & 15, kb = lane >> 4;
    const int tiles = (M + 15) >> 4;
    for (int tile = tile_first; tile < tiles; tile += tile_step) {
        const int de = dtab[min(tile * 16 + j, M - 1)];
        const int gbase = de & 0xffff, iy = (de >> 16) & 0xff, ix = (de >> 24) & 0xff;
        f32x4 acc[2][2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[t][1] = acc[t][0]; }
        F16x2 av[4];
        f32x4 ga[4][2];
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {                         // all LDS reads first
            const int oy = iy - (tap >> 1), ox = ix - (tap & 1);
            const bool valid = (unsigned)oy < (unsigned)oh && (unsigned)ox < (unsigned)ow;
            const int grow = valid ? gbase - (tap >> 1) * ow - (tap & 1) : zero_row;
            if constexpr (GPL) {
                const unsigned short* gp = static_cast<const unsigned short*>(gv) + grow * PL32 + 8 * kb;
                av[tap].h = *reinterpret_cast<const u32x4*>(gp);
                av[tap].l = *reinterpret_cast<const u32x4*>(gp + g_lo);
            } else {
                const float* gp = static_cast<const float*>(gv) + grow * 36 + 8 * kb;
                ga[tap][0] = *reinterpret_cast<const f32x4*>(gp);
                ga[tap][1] = *reinterpret_cast<const f32x4*>(gp + 4);
            }
        }
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            if constexpr (!GPL) av[tap] = split_f16x2(ga[tap][0], ga[tap][1]);
#pragma unroll
            for (int t = 0; t < 2; ++t) mma_f16x3(av[tap], bw[tap][t], acc[t][0], acc[t][1]);
        }
        // C/D layout: col = lane & 15 -> channel c_lo + 2j + t (the packed weights' column order), row = (lane >> 4) * 4 + reg
        if constexpr (OPL) {
            // this lane's two results are ADJACENT channels c_lo + 2j, + 1 (the packed weights' column order, qnet.h PK_CONV*_DG): the
            // activation's pieces and the result's are one 4-byte LDS access per plane.  The four rows' mask words are read TOGETHER, at
            // clamped rows, before the first result is formed (round 3: under `if (mo >= M) continue` each row was a branch with its own
            // s_waitcnt -- four LDS latencies in a row per tile); only the stores are guarded.
            u32 bits[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const u32* ph = reinterpret_cast<const u32*>(static_cast<const unsigned short*>(actv) + min(tile * 16 + 4 * kb + r, M - 1) * PSA + c_lo + 2 * j);
                bits[r] = ph[0] | ph[act_lo >> 1];
            }
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = tile * 16 + 4 * kb + r;
                const bool in = mo < M;
                const float v0 = (in && (bits[r] & 0x7fffu) != 0u) ? f16x2_sum(acc[0][0][r], acc[0][1][r]) : 0.f;
                const float v1 = (in && (bits[r] & 0x7fff0000u) != 0u) ? f16x2_sum(acc[1][0][r], acc[1][1][r]) : 0.f;
                colsum[0] += v0; colsum[1] += v1;
                u32 h, l;
                split_f16x2_pair(v0, v1, h, l);
                u32* ph = reinterpret_cast<u32*>(static_cast<unsigned short*>(actv) + mo * PSA + c_lo + 2 * j);
                if (in) { ph[0] = h; ph[act_lo >> 1] = l; }
            }
        } else
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mo = tile * 16 + 4 * kb + r;
            if (mo >= M) continue;
            {
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    float* p = static_cast<float*>(actv) + mo * PSA + c_lo + 2 * j + t;
                    *p = *p > 0.f ? f16x2_sum(acc[t][0][r], acc[t][1][r]) : 0.f;
                }
            }
        }
    }
}

#define A1PS 64                         // row stride (halves) of the a1 piece planes: unpadded, because they are filled by LDS-DMA (1 KB contiguous per wave instruction)

// CP: patch-word input.  The first convolution's patch image then has one column per DATA bit of a pixel's patch (K_data <= 32: four corners per
// syndrome plane, the centre per action plane) plus one per constant POSITION (5: the places where padding_syndrome's decoration can put a 1 in a
// 3 x 3 stride-2 patch -- the same on every syndrome plane, so their kernel rows share one gradient): 16 KG1 = 32 or 48 columns where the uint8 image has
// 64 .. 96, built by expanding the pixel's word | its constant mask byte by byte through an LDS table; the result is scattered to the kernel's Keras
// rows at the end (every other row's gradient is 0: its input cell is 0 in every observation).
template <int KG1, bool CP = false>     // first convolution's K padded to 16 * KG1
__global__ __launch_bounds__(CB_THREADS, 2) void conv_bwd_chain_kernel(ConvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u8* s_in = smem;
    int* s_mis = reinterpret_cast<int*>(smem + a.off_mis);
    unsigned short* s_a2 = reinterpret_cast<unsigned short*>(smem + a.off_a2);      // a2, then g2 in place: piece planes [2][S*r2 + 1][PL32]
    unsigned short* s_g3 = reinterpret_cast<unsigned short*>(smem + a.off_g3);      // g3: piece planes [2][S*r3 + 1][PL32]
    u8* s_col = smem + a.off_t1;                                     // observation patch image [S*r1][16*KG1] bytes
    int* s_ko = reinterpret_cast<int*>(smem + a.off_ko);
    int* t2 = reinterpret_cast<int*>(smem + a.off_t2);
    int* t3 = reinterpret_cast<int*>(smem + a.off_t3);
    int* d2 = reinterpret_cast<int*>(smem + a.off_d2);              // dgrad_inplace's tables of the g2 / g1 phases
    int* d1 = reinterpret_cast<int*>(smem + a.off_d1);
    int* t1 = reinterpret_cast<int*>(smem + a.off_tp);              // [S*r1]: sample << 16 | byte offset of the pixel's patch origin inside an observation
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
    const int S = a.S, r1 = a.oh1 * a.ow1, r2 = a.oh2 * a.ow2, r3 = a.oh3 * a.ow3;
    const int in_bytes = CP ? a.slot : a.C * a.H * a.W;
    const int zero2 = S * r2, zero3 = S * r3;                        // all-zero rows of the g2 (= a2) and g3 images
    const int LA2 = (S * r2 + 1) * PL32, LG3 = (S * r3 + 1) * PL32;   // halves from a2's / g3's h plane to its l plane
    constexpr int NW1 = (4 * KG1 + CB_WAVES - 1) / CB_WAVES;        // dW1 tiles (KG1 x 4) per wave
    constexpr int KP = 16 * KG1;                                    // bytes per row of the observation patch image

    if (CB_PRIO && wave >= CB_WAVES / 2) __builtin_amdgcn_s_setprio(1);      // the second-dispatched half loses every issue arbitration by age: static priority evens the pair out
    DQ_STAMP_PAIR2(3);
    DQ_STAMP_WG(DQ_TAG_CONV_BWD, 0);
    DQ_STAMP(DQ_TAG_CONV_BWD, 25);
    // this wave's replay row of the workgroup's FIRST group (obs_row below), requested before anything else: its round trip (the index vector was
    // written by the launch before) then runs under the table copies instead of standing in front of the first group's copies
    int row_first = 0;
    if ((int)blockIdx.x < a.groups) {
        const int gb0 = (int)blockIdx.x * a.S, gns = min(a.S, a.batch - gb0);
        row_first = gb0 + min(wave, gns - 1);
        if (a.index) {
            const __attribute__((address_space(4))) int32_t* idx = (const __attribute__((address_space(4))) int32_t*)(uintptr_t)a.index;
            row_first = idx[row_first] + a.index_off;
            if (row_first >= a.index_mod) row_first -= a.index_mod;
        }
    }
    row_first = __builtin_amdgcn_readfirstlane(row_first);          // (wave-uniform by construction; said so)
    // ---- group-independent tables and zero rows ----------------------------------------------------------------
    // (copied from the host-built tables: computing them here took two integer divisions per entry)
    for (int m = tid; m < S * r1; m += CB_THREADS) {
        const int e1 = a.rowtab1[m], e2 = a.rowtab[CONV_ROWTAB + min(m, S * r2 - 1)], e3 = a.rowtab[2 * CONV_ROWTAB + min(m, S * r3 - 1)];
        const int e4 = a.rowtab[3 * CONV_ROWTAB + min(m, S * r2 - 1)], e5 = a.rowtab[4 * CONV_ROWTAB + m];
        t1[m] = e1;                                                 // sample << 16 | byte offset of the patch origin inside an observation
        d1[m] = e5;
        if (m < S * r2) { t2[m] = e2; d2[m] = e4; }                 // t2: float offset of the a1 row under output pixel m of the second convolution
        if (m < S * r3) t3[m] = e3;                                 // t3: float offset of the a2 row
    }
    if (tid < 96) s_ko[tid] = CP ? a.srctab[tid] : a.kofftab[tid];      // (CP: the scatter map of the kernel's end -- read there from LDS: a dependent global load per
                                                                       // output row serialised eight round trips behind the last group, +2.4 us)
    uint2* s_lut = reinterpret_cast<uint2*>(smem + a.off_lut);       // CP: byte -> its bits as eight bytes
    if (CP && tid < 256) {
        const u32 b = (u32)tid;
        s_lut[tid] = uint2{(b & 1u) | (b & 2u) << 7 | (b & 4u) << 14 | (b & 8u) << 21, ((b >> 4) & 1u) | ((b >> 4) & 2u) << 7 | ((b >> 4) & 4u) << 14 | ((b >> 4) & 8u) << 21};
    }
    if (tid < PL32) { s_a2[zero2 * PL32 + tid] = 0; s_a2[LA2 + zero2 * PL32 + tid] = 0; s_g3[zero3 * PL32 + tid] = 0; s_g3[LG3 + zero3 * PL32 + tid] = 0; }
    DQ_STAMP(DQ_TAG_CONV_BWD, 27);

    // ---- per-lane constants of the weight-gradient phases ----------------------------------------------------------
    // dW3 [128 x 32]: wave w owns k-tile w = (ky,kx) = w>>1, channels 16*(w&1)..; dW2 [256 x 32]: k-tiles 2w, 2w+1 = (ky,kx) = w>>1, channels 16*(2(w&1)+u)
    // dW1 [16 KG1 x 64]: tile id = wave + 8u -> k-tile id>>2, n-tile wave & 3 (the same for every u)
    const int kyx = wave >> 1, ky = kyx >> 1, kx = kyx & 1;
    const int aoff3 = (ky * a.ow2 + kx) * PL32 + 16 * (wave & 1);         // (t3: half offsets of a2's plane rows; t2: float offsets of a1's rows)
    const int aoff2 = (ky * a.ow1 + kx) * A1PS + 32 * (wave & 1);          // (t2: half offsets of a1's plane rows)
    f32x4 acc3[2], acc3l[2], acc2[2][2], acc2l[2][2], acc1[NW1], acc1l[NW1];
    float bs3 = 0.f, bs2[2] = {0.f, 0.f}, bs1[2] = {0.f, 0.f};     // bs2 / bs1: every wave's share of g2's / g1's column sums (its tiles' rows); bs3: this thread's share of g3's column tid & 31
#pragma unroll
    for (int t = 0; t < 2; ++t) { acc3[t] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[0][t] = acc3[t]; acc2[1][t] = acc3[t]; acc2l[0][t] = acc3[t]; acc2l[1][t] = acc3[t]; acc3l[t] = acc3[t]; }
#pragma unroll
    for (int u = 0; u < NW1; ++u) { acc1[u] = f32x4{0.f, 0.f, 0.f, 0.f}; acc1l[u] = acc1[u]; }

    // ---- every input image goes global -> LDS by LDS-DMA (no registers; lds_dma16 / lds_dma4, qnet.h: copies hipcc does not wait for behind our
    //      back -- until round 3 it put an s_waitcnt vmcnt(0) behind every one of them), issued as early as its LDS target is free, so that group
    //      k + 1's inputs land while group k computes.  (All workgroups run in lockstep: a load phase of its own is a burst on HBM
    //      that nothing overlaps.)  One wave instruction writes 64 lanes x 16 B (or x 4 B) CONTIGUOUSLY in LDS from per-lane global
    //      addresses; inactive lanes write nothing.
    // a1 piece planes [M1][64 halves] each: wave w copies 1 KB chunks w, w + 8, ... of the h plane, then of the l plane (LA1 halves further in
    // LDS).  Double-buffered (a1, then g1 in place, is live until the end of dW1).
    const int LA1 = (S * r1 * A1PS + 511) & ~511;                    // halves from a1's h plane to its l plane (whole 1 KB chunks)
    auto issue_a1 = [&](int g, unsigned short* dst) {
        const int gb0 = g * S, gM1 = min(S, a.batch - gb0) * r1;
        const int bytes = gM1 * A1PS * 2, chunks = (bytes + 1023) >> 10;
        for (int c = wave; c < 2 * chunks; c += CB_WAVES) {
            const int piece = c >= chunks ? 1 : 0, ch = c - piece * chunks;
            const char* src = reinterpret_cast<const char*>(a.a1p + piece * a.a1_lo + (size_t)gb0 * r1 * A1PS);
            int off = ch * 1024 + lane * 16;
            if (off >= bytes) off = 0;                                  // tail lanes: a valid address; they land in the image's padding
            lds_dma16(src + off, lds_addr(dst + piece * LA1 + ch * 512));
        }
    };
    // piece planes [rows][32 halves] (a2, g3) -> LDS rows of PL32 halves = 5 lane slots of 16 B (slot 4 of every row is padding), both planes
    auto issue_pl32 = [&](const unsigned short* src, size_t src_lo, int rows, unsigned short* dst, int dst_lo) {
        const int slots = rows * 5, chunks = (slots + 63) >> 6;
        for (int c = wave; c < 2 * chunks; c += CB_WAVES) {
            const int piece = c >= chunks ? 1 : 0, ch = c - piece * chunks;
            const int q = ch * 64 + lane, row = q / 5, part = q - row * 5;
            if (q < slots && part < 4)
                lds_dma16(src + piece * src_lo + (size_t)row * 32 + part * 8, lds_addr(dst + piece * dst_lo + ch * 512));
        }
    };
    auto issue_a2 = [&](int g, int rows) { issue_pl32(a.a2p + (size_t)g * S * r2 * 32, a.a2_lo, rows, s_a2, LA2); };
    auto issue_g3 = [&](int g, int rows) { issue_pl32(a.g3p + (size_t)g * S * r3 * 32, a.g3_lo, rows, s_g3, LG3); };
    // observations: lane l copies aligned dword l of a 256-byte piece of a sample's arbitrarily aligned row -- whole aligned dwords, also
    // where they straddle the neighbouring rows (see fused.hip: the window stays inside the caller's allocation)
    // this wave's sample of group g (one sample per wave: S <= 8 = CB_WAVES): its replay-ring row, read through the CONSTANT address space -- a scalar
    // load the compiler may issue early; as an ordinary load inside the group loop (which stores to global memory) hipcc made it a vector load +
    // s_waitcnt vmcnt(0) + v_readfirstlane right in front of the copies: a memory latency in every group's prefetch
    static_assert(CB_WAVES >= 8, "one observation per wave");
    auto obs_row = [&](int g) {
        const int gb0 = g * S, gns = min(S, a.batch - gb0);
        int row = gb0 + min(wave, gns - 1);
        if (a.index) {
            const __attribute__((address_space(4))) int32_t* idx = (const __attribute__((address_space(4))) int32_t*)(uintptr_t)a.index;
            row = idx[row] + a.index_off;
            if (row >= a.index_mod) row -= a.index_mod;
        }
        return row;
    };
    auto issue_obs = [&](int g, int row) {
        const int gb0 = g * S, gns = min(S, a.batch - gb0);
        const int pieces = (a.slot + 255) >> 8;
        const int s = wave;
        if (CP) {                                                   // patch words: one aligned row of `slot` bytes, 16 bytes per lane
            if (s < gns && lane < (a.slot >> 4)) lds_dma16(a.obs + (size_t)row * in_bytes + 16 * lane, lds_addr(s_in + s * a.slot));
        } else if (s < gns) {                                       // wave-uniform
            const u8* src = a.obs + (size_t)row * in_bytes;
            const int mis = (int)(reinterpret_cast<uintptr_t>(src) & 3);
            for (int pc = 0; pc < pieces; ++pc) {
                const int d = pc * 64 + lane;
                if (4 * d < mis + in_bytes)
                    lds_dma4(reinterpret_cast<const u32*>(src - mis) + d, lds_addr(s_in + s * a.slot + pc * 256));
            }
            if (lane == 0) s_mis[s] = mis;
        }
    };
    DQ_STAMP(DQ_TAG_CONV_BWD, 28);
    if ((int)blockIdx.x < a.groups) {
        const int g = blockIdx.x, gns = min(S, a.batch - g * S);
        issue_obs(g, row_first);
        issue_g3(g, gns * r3);
        issue_a2(g, gns * r2);
        if (a.a1_alt) issue_a1(g, reinterpret_cast<unsigned short*>(smem + a.off_a1));
    }
    DQ_STAMP(DQ_TAG_CONV_BWD, 29);

    F16x2 bw[4][2];                                                 // data-gradient weights (f16 pieces): loaded one phase ahead of their use
    int it = 0;
    for (int grp = blockIdx.x; grp < a.groups; grp += gridDim.x, ++it) {
        unsigned short* s_a1 = reinterpret_cast<unsigned short*>(smem + a.off_a1 + (it & 1) * a.a1_alt);      // a1, then g1 in place: piece planes
        const int b0 = grp * S, ns = min(S, a.batch - b0);
        const int M1 = ns * r1, M2 = ns * r2, M3 = ns * r3;
        const int nxt = grp + (int)gridDim.x;
        const int ns_nxt = min(S, a.batch - nxt * S);
        const int row_nxt = nxt < a.groups ? obs_row(nxt) : 0;      // requested here, used behind g2 (block-uniform)
        const int sb = (grp == (int)blockIdx.x) ? 0 : 12;          // DQ_STAMP slots of the first / a later group
        (void)sb;
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 0);
        // ---- stage: nothing to copy -- wait for this wave's DMA pieces of the group's images, then meet ------------------------------
        if (!a.a1_alt) { __syncthreads(); issue_a1(grp, s_a1); }    // single-buffered a1: free only when every wave has left dW1
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 11);
        __syncthreads();
        dgrad_load_w(bw, a.packed + PK_CONV3_DG, lane);             // first used by g2: lands during the patch image and dW3
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 1);
        // ---- observation patch image: row m = the K1 bytes conv1 multiplies for output pixel m (zeros past K1), so that dW1's A
        //      operand is 16 consecutive bytes per quarter-wave instead of a scattered byte gather -------------------------------
        if constexpr (CP) {
            // row m = the bits of pixel m's word (data), then of its constant mask, one byte each: eight bytes per task from the byte table
            const u32* s_w = reinterpret_cast<const u32*>(s_in);
            for (int task = tid; task < M1 * (KP / 8); task += CB_THREADS) {
                const int m = task / (KP / 8), g = task - m * (KP / 8);
                const int e = t1[m];
                const u64 bits = (u64)s_w[e & 0xffff] | (u64)(u32)(e >> 16) << a.kd;
                *reinterpret_cast<uint2*>(s_col + m * KP + 8 * g) = s_lut[(u32)(bits >> (8 * g)) & 0xffu];
            }
        } else
        for (int task = tid; task < M1 * (KP / 16); task += CB_THREADS) {
            const int m = task / (KP / 16), q = task - m * (KP / 16);
            const int to = t1[m], s = to >> 16;                    // (sample, byte offset of the patch origin inside its observation)
            const u8* op = s_in + s * a.slot + s_mis[s] + (to & 0xffff);
            // the 16 offsets first (four ds_read_b128), then 16 UNCONDITIONAL byte reads at clamped offsets, masked by select afterwards:
            // a read under `off >= 0 ? .. : 0` is a branch with its own s_waitcnt -- 32 serialised LDS latencies per task (measured:
            // 3K cycles per task, 5.9K for this phase)
            int ko[16];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int4 k4 = *reinterpret_cast<const int4*>(s_ko + 16 * q + 4 * e);
                ko[4 * e] = k4.x; ko[4 * e + 1] = k4.y; ko[4 * e + 2] = k4.z; ko[4 * e + 3] = k4.w;
            }
            u32 by[16];
#pragma unroll
            for (int e = 0; e < 16; ++e) by[e] = op[max(ko[e], 0)];
            u32 wd[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                u32 v = 0;
#pragma unroll
                for (int bb = 0; bb < 4; ++bb) v |= (ko[4 * e + bb] >= 0 ? by[4 * e + bb] : 0u) << (8 * bb);
                wd[e] = v;
            }
            *reinterpret_cast<uint4*>(s_col + m * KP + 16 * q) = uint4{wd[0], wd[1], wd[2], wd[3]};
        }

        // the third convolution's bias gradient = column sums of g3, from its pieces: thread (column tid & 31, row class tid >> 5) adds its
        // rows here (one pass per group; the classes are combined at the kernel's end)
        for (int row = tid >> 5; row < M3; row += CB_THREADS / 32) {
            const unsigned short* gp = s_g3 + row * PL32 + (tid & 31);
            bs3 += (float)__builtin_bit_cast(_Float16, gp[0]) + (float)__builtin_bit_cast(_Float16, gp[LG3]) * F16_LO_INV;
        }

        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 2);
        // ---- dW3 += im2col(a2)^T g3 ------------------------------------------------------------------------------
        {
            // f16 pipe: K = 32 rows per block, lane group kq supplies rows m0 + 4kq + (e & 3) + 16 (e >> 2) (e = 0 .. 7) of both operands.
            // A = the wave's 16 columns of a2 at its tap, G = g3's columns j, 16 + j: both ready-made pieces (the forward saved a2 that way,
            // the dense backward writes gX = g3 that way), fetched row-major -> operand order by two transposing reads per piece and tile
            // (this lane points at row ri, column segment cseg of each four-row read) -- no arithmetic beside the MFMAs.
            const int ri = j >> 2, cseg = 4 * (j & 3);
            auto rdA = [&](int m0, F16x2& A) {
                const int r0 = min(m0 + 4 * kq + ri, M3 - 1), r1 = min(m0 + 16 + 4 * kq + ri, M3 - 1);      // (rows past M3: any valid row -- their g is zero)
                A = lds_tr8(s_a2 + t3[r0] + aoff3 + cseg, s_a2 + t3[r1] + aoff3 + cseg, LA2);
            };
            auto rdG = [&](int m0, F16x2& G0, F16x2& G1) {
                const int r0 = min(m0 + 4 * kq + ri, M3 - 1), r1 = min(m0 + 16 + 4 * kq + ri, M3 - 1);      // (rows past M3: cleared in mm)
                const unsigned short* p0 = s_g3 + r0 * PL32 + cseg;
                const unsigned short* p1 = s_g3 + r1 * PL32 + cseg;
                G0 = lds_tr8(p0, p1, LG3);
                G1 = lds_tr8(p0 + 16, p1 + 16, LG3);
            };
            auto mm = [&](int m0, const F16x2& A, F16x2& G0, F16x2& G1) {
                if (m0 + 32 > M3) {                                   // rows past M3: their halves of g's pieces cleared (element e = half e of the operand)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const u32 lo = m0 + 4 * kq + ((2 * d) & 3) + 16 * ((2 * d) >> 2) < M3 ? 0xffffu : 0u;
                        const u32 hi = m0 + 4 * kq + ((2 * d + 1) & 3) + 16 * ((2 * d + 1) >> 2) < M3 ? 0xffff0000u : 0u;
                        G0.h[d] &= lo | hi; G0.l[d] &= lo | hi; G1.h[d] &= lo | hi; G1.l[d] &= lo | hi;
                    }
                }
                mma_f16x3(A, G0, acc3[0], acc3l[0]);
                mma_f16x3(A, G1, acc3[1], acc3l[1]);
            };
            F16x2 aA, aB, g0A, g1A, g0B, g1B;
            rdA(0, aA); rdG(0, g0A, g1A);
            for (int m0 = 0;;) {
                rdA(m0 + 32, aB); rdG(m0 + 32, g0B, g1B);
                mm(m0, aA, g0A, g1A); m0 += 32; if (m0 >= M3) break;
                rdA(m0 + 32, aA); rdG(m0 + 32, g0A, g1A);
                mm(m0, aB, g0B, g1B); m0 += 32; if (m0 >= M3) break;
            }
        }
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 3);
        __syncthreads();                                            // every wave is done reading a2
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 4);
        // ---- g2 = (g3 (*) W3^T) * [a2 > 0], in place over a2 -------------------------------------------------------------
        dgrad_inplace<PL32, true, true>(bw, s_g3, LG3, zero3, s_a2, LA2, 0, d2, a.oh3, a.ow3, M2, wave, CB_WAVES, lane, bs2);
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 5);
        __syncthreads();
        if (nxt < a.groups) {                                       // the observation slots and g3 are dead now; so is the other a1 buffer
            issue_obs(nxt, row_nxt);
            issue_g3(nxt, ns_nxt * r3);
            if (a.a1_alt) issue_a1(nxt, reinterpret_cast<unsigned short*>(smem + a.off_a1 + ((it + 1) & 1) * a.a1_alt));
        }
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 6);
        // ---- dW2 += im2col(a1)^T g2 -----------------------------------------------------------------------------------
        {
            // One K = 32 block = 32 rows, lane group kq supplies rows m0 + 4kq + (e & 3) + 16 (e >> 2) of both operands (any assignment of the
            // block's rows to (kq, e) is a permutation of the reduction index as long as both operands use it).  A = a1's columns of the wave's tap,
            // G = g2's columns j, 16 + j: both ready-made pieces (the forward saved a1 as pieces, the g2 phase split g2 on write), four
            // transposing reads per tile and piece pair -- no arithmetic at all beside the MFMAs.  The reads of trip t + 1 are issued before the
            // MFMAs of trip t.
            const int ri = j >> 2, cseg = 4 * (j & 3);
            auto rdG = [&](int m0, F16x2& G0, F16x2& G1) {
                const int r0 = min(m0 + 4 * kq + ri, M2 - 1), r1 = min(m0 + 16 + 4 * kq + ri, M2 - 1);      // (rows past M2: masked in mm)
                const unsigned short* p0 = s_a2 + r0 * PL32 + cseg;
                const unsigned short* p1 = s_a2 + r1 * PL32 + cseg;
                G0 = lds_tr8(p0, p1, LA2);
                G1 = lds_tr8(p0 + 16, p1 + 16, LA2);
            };
            auto rdA = [&](int m0, F16x2& A0, F16x2& A1) {
                const int r0 = min(m0 + 4 * kq + ri, M2 - 1), r1 = min(m0 + 16 + 4 * kq + ri, M2 - 1);
                const unsigned short* p0 = s_a1 + t2[r0] + aoff2 + cseg;
                const unsigned short* p1 = s_a1 + t2[r1] + aoff2 + cseg;
                A0 = lds_tr8(p0, p1, LA1);
                A1 = lds_tr8(p0 + 16, p1 + 16, LA1);
            };
            auto mm = [&](int m0, const F16x2& A0, const F16x2& A1, F16x2& G0, F16x2& G1) {
                if (m0 + 32 > M2) {                                   // rows past M2: their halves of g's pieces cleared (element e = half e of the operand)
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const u32 lo = m0 + 4 * kq + ((2 * d) & 3) + 16 * ((2 * d) >> 2) < M2 ? 0xffffu : 0u;
                        const u32 hi = m0 + 4 * kq + ((2 * d + 1) & 3) + 16 * ((2 * d + 1) >> 2) < M2 ? 0xffff0000u : 0u;
                        G0.h[d] &= lo | hi; G0.l[d] &= lo | hi; G1.h[d] &= lo | hi; G1.l[d] &= lo | hi;
                    }
                }
                mma_f16x3(A0, G0, acc2[0][0], acc2l[0][0]);
                mma_f16x3(A0, G1, acc2[0][1], acc2l[0][1]);
                mma_f16x3(A1, G0, acc2[1][0], acc2l[1][0]);
                mma_f16x3(A1, G1, acc2[1][1], acc2l[1][1]);
            };
            F16x2 a0A, a1A, a0B, a1B, g0A, g1A, g0B, g1B;
            rdG(0, g0A, g1A); rdA(0, a0A, a1A);
            for (int m0 = 0;;) {
                rdG(m0 + 32, g0B, g1B); rdA(m0 + 32, a0B, a1B);
                mm(m0, a0A, a1A, g0A, g1A); m0 += 32; if (m0 >= M2) break;
                rdG(m0 + 32, g0A, g1A); rdA(m0 + 32, a0A, a1A);
                mm(m0, a0B, a1B, g0B, g1B); m0 += 32; if (m0 >= M2) break;
            }
        }
        dgrad_load_w(bw, a.packed + PK_CONV2_DG + 8 * PK_BLOCK * (wave >> 2), lane);      // (96 registers of pieces: not held across dW2)
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 7);
        __syncthreads();                                            // every wave is done reading a1
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 8);
        // ---- g1 = (g2 (*) W2^T) * [a1 > 0], in place over a1: waves 0-3 channels 0..31, waves 4-7 channels 32..63 -------------
        dgrad_inplace<A1PS, true, true>(bw, s_a2, LA2, zero2, s_a1, LA1, 32 * (wave >> 2), d1, a.oh2, a.ow2, M1, wave & 3, 4, lane, bs1);
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 9);
        __syncthreads();
        DQ_STAMP(DQ_TAG_CONV_BWD, sb + 10);
        if (nxt < a.groups) issue_a2(nxt, ns_nxt * r2);            // g2 (in a2) is dead; lands during dW1
        // ---- dW1 += patches^T g1 ----------------------------------------------------------------------------------------
        {
            // On the f16 pipe: the patch operand is binary (exact in f16), g1 arrives as ready-made pieces (split on write by the g1 phase):
            // one K = 32 MFMA per piece.  Lane group kq supplies rows m0 + kq + 4 (e & 3) + 16 (e >> 2) of both operands (consecutive kq =
            // consecutive patch rows, 16 banks apart).  Both come out of their row-major LDS images by TRANSPOSING reads: g1's columns by
            // lds_tr8 (this lane points at row kq + 4 (j >> 2) (+ 16), column segment 4 (j & 3)), the patch bytes by ds_read_b64_tr_b8
            // (tools/probe/tr8_probe.hip: lanes 2e, 2e+1 of a 16-lane group point at the two 8-byte halves of row e, lane i receives column
            // i of the eight rows) -- ONE read per tile where eight ds_read_u8 were; four byte-permutes + four multiplies make its eight
            // halves.  The LDS reads of trip t + 1 are issued before the MFMAs of trip t.
            const int cseg = 16 * (wave & 3) + 4 * (j & 3), rj = kq + 4 * (j >> 2);
            const int re = j >> 1, rowb = kq + 4 * (re & 3) + 16 * (re >> 2);      // the patch row (inside a block) this lane points at
            const u8* cp = s_col + 16 * (wave >> 2) + 8 * (j & 1);  // + 32 per further tile of this wave (k-tile + 2)
            auto rd = [&](int m0, uint2 (&ab)[NW1], F16x2& G) {
                const int r0 = min(m0 + rj, M1 - 1), r1 = min(m0 + 16 + rj, M1 - 1);      // (rows past M1: masked in mm)
                G = lds_tr8(s_a1 + r0 * A1PS + cseg, s_a1 + r1 * A1PS + cseg, LA1);
                const u8* cb = cp + min(m0 + rowb, M1 - 1) * KP;
#pragma unroll
                for (int u = 0; u < NW1; ++u) {
                    typedef int i32x2 __attribute__((ext_vector_type(2)));
                    const i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(cb + (wave + CB_WAVES * u < 4 * KG1 ? 32 * u : 0)));
                    ab[u] = uint2{(u32)v[0], (u32)v[1]};
                }
            };
            // NO condition around an MFMA, not even a wave-uniform one: hipcc then copies the accumulators after every MFMA
            // (each copy waits for the result) -- a tile this wave does not have accumulates garbage that is never stored
            auto mm = [&](int m0, const uint2 (&ab)[NW1], F16x2& G) {
                if (m0 + 32 > M1) {                                   // rows past M1: their halves of g's pieces cleared
#pragma unroll
                    for (int d = 0; d < 4; ++d) {
                        const u32 lo = m0 + kq + 4 * ((2 * d) & 3) + 16 * ((2 * d) >> 2) < M1 ? 0xffffu : 0u;
                        const u32 hi = m0 + kq + 4 * ((2 * d + 1) & 3) + 16 * ((2 * d + 1) >> 2) < M1 ? 0xffff0000u : 0u;
                        G.h[d] &= lo | hi; G.l[d] &= lo | hi;
                    }
                }
#pragma unroll
                for (int u = 0; u < NW1; ++u) {
                    // bytes (0 / 1) e, e + 1 -> one dword of two halves: byte-permute into the halves' low bytes, times f16(1.0) = 0x3c00
                    // (v_mul_u32_u24: a 32-bit multiply is quarter rate)
                    u32x4 av;
                    av[0] = __umul24(__builtin_amdgcn_perm(0u, ab[u].x, 0x0c010c00u), 0x3c00u);
                    av[1] = __umul24(__builtin_amdgcn_perm(0u, ab[u].x, 0x0c030c02u), 0x3c00u);
                    av[2] = __umul24(__builtin_amdgcn_perm(0u, ab[u].y, 0x0c010c00u), 0x3c00u);
                    av[3] = __umul24(__builtin_amdgcn_perm(0u, ab[u].y, 0x0c030c02u), 0x3c00u);
                    acc1[u] = MFMA_F16(av, G.h, acc1[u]);
                    acc1l[u] = MFMA_F16(av, G.l, acc1l[u]);
                }
            };
            uint2 abA[NW1], abB[NW1];
            F16x2 gA, gB;
            rd(0, abA, gA);
            for (int m0 = 0;;) {
                rd(m0 + 32, abB, gB);
                mm(m0, abA, gA); m0 += 32; if (m0 >= M1) break;
                rd(m0 + 32, abA, gA);
                mm(m0, abB, gB); m0 += 32; if (m0 >= M1) break;
            }
        }
    }

    DQ_STAMP(DQ_TAG_CONV_BWD, 24);
    // ---- one partial per workgroup -----------------------------------------------------------------------------------
    float* out = a.partial + (size_t)blockIdx.x * a.pstride;
#pragma unroll
    for (int t = 0; t < 2; ++t)
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            out[a.w_off[2] + (16 * wave + 4 * kq + r) * 32 + 16 * t + j] = f16x2_sum(acc3[t][r], acc3l[t][r]);
#pragma unroll
            for (int u = 0; u < 2; ++u) out[a.w_off[1] + (16 * (2 * wave + u) + 4 * kq + r) * 32 + 16 * t + j] = f16x2_sum(acc2[u][t][r], acc2l[u][t][r]);
        }
    if constexpr (!CP) {
#pragma unroll
        for (int u = 0; u < NW1; ++u) {
            const int id = wave + CB_WAVES * u, kt = id >> 2, nt = id & 3;
            if (id < 4 * KG1) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int k = 16 * kt + 4 * kq + r;
                    if (k < a.K1) out[a.w_off[0] + k * 64 + 16 * nt + j] = f16x2_sum(acc1[u][r], acc1l[u][r]);
                }
            }
        }
    }
    {   // the second and first convolution's: every wave holds the column sums of the g2 / g1 tiles it produced (g2: columns 2j + t; g1:
        // columns 32 (wave >> 2) + 2j + t) -- combined in fixed order through LDS
        __syncthreads();                                            // (every LDS image is dead)
        float* s_b = reinterpret_cast<float*>(smem);
        float* s_res = s_b + 2048;                                  // CP: the patch image's gradient [16 KG1][64], then scattered to the kernel's Keras rows
        if constexpr (CP) {
#pragma unroll
            for (int u = 0; u < NW1; ++u) {
                const int id = wave + CB_WAVES * u, kt = id >> 2, nt = id & 3;
                if (id < 4 * KG1) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) s_res[(16 * kt + 4 * kq + r) * 64 + 16 * nt + j] = f16x2_sum(acc1[u][r], acc1l[u][r]);
                }
            }
        }
        s_b[256 + tid] = bs3;                                       // the third's: thread (column tid & 31, row class tid >> 5)
#pragma unroll
        for (int t = 0; t < 2; ++t) {
            float v = bs2[t], w1 = bs1[t];
            v += __shfl_xor(v, 16); v += __shfl_xor(v, 32);
            w1 += __shfl_xor(w1, 16); w1 += __shfl_xor(w1, 32);
            if (kq == 0) { s_b[wave * 32 + 2 * j + t] = v; s_b[768 + wave * 32 + 2 * j + t] = w1; }      // (dgrad_inplace: this lane's columns 2j + t)
        }
        __syncthreads();
        if (tid < 32) {
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < CB_WAVES; ++w) v += s_b[w * 32 + tid];
            out[a.b_off[1] + tid] = v;
        } else if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int cl = 0; cl < CB_THREADS / 32; ++cl) v += s_b[256 + cl * 32 + (tid - 32)];
            out[a.b_off[2] + tid - 32] = v;
        } else if (tid < 128) {
            const int c = tid - 64, half = c >> 5;                  // column c of g1: waves 4 half .. 4 half + 3
            float v = 0.f;
#pragma unroll
            for (int w = 0; w < 4; ++w) v += s_b[768 + (4 * half + w) * 32 + (c & 31)];
            out[a.b_off[0] + c] = v;
        }
        if constexpr (CP) {
            for (int i = tid; i < a.K1 * 64; i += CB_THREADS) {
                const int src = s_ko[i >> 6];
                out[a.w_off[0] + i] = src >= 0 ? s_res[src * 64 + (i & 63)] : 0.f;
            }
        }
    }
    DQ_STAMP(DQ_TAG_CONV_BWD, 26);
    DQ_STAMP_WG(DQ_TAG_CONV_BWD, 1);
}

// ---------------------------------------------------------------------------------------------------------------
struct DenseBwdPlan { int NT2, ldg, off_g3, off_gy2, off_gh1, off_w3t; size_t lds; };

static bool plan_dense_bwd(const dq_qnet* Q, DenseBwdPlan* P) {
    const int nc = Q->cfg.n_conv;
    if (Q->cfg.n_ff != 1) return false;
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    if (D1.nout != DENSE_HID || (D1.nin & 15)) return false;
    const int N2 = D2.nout, N3 = Q->cfg.dueling ? Q->L[nc + 2].nout : 0;
    const int widest = N3 > N2 ? N3 : N2;
    if (widest > 112) return false;
    P->NT2 = (widest <= 64 && N3 > 0) ? 4 : 7;                      // (the 4-tile kernel's TD form folds the dueling layer -- SHORT, below --: a network without one takes the general kernel)
    P->ldg = 16 * P->NT2 + 4;
    size_t off = 0;
    P->off_g3 = (int)off; off += up16((size_t)DENSE_ROWS * P->ldg * 4);
    P->off_gy2 = (int)off; off += up16((size_t)2 * DENSE_ROWS * (32 * (N2 <= 64 ? 2 : 4) + 8) * 2);   // f16 piece planes (K padded to 2 or 4 blocks)
    P->off_gh1 = (int)off; off += up16((size_t)2 * DENSE_ROWS * (DENSE_HID + 8) * 2);
    P->off_w3t = (int)off; if (P->NT2 == 4 && N3 > 0) off += (size_t)64 * 64 * 4;      // W3'^T (the TD launch's gY2 shortcut, dense_bwd_chain_kernel SHORT)
    P->lds = off;
    return true;
}

struct ConvBwdPlan { int S, KG1, slot, off_mis, off_a1, off_a2, off_g3, off_t1, off_t2, off_t3, off_ko, off_tp, off_d2, off_d1, off_lut, a1_alt; size_t lds; };

// patch: patch-word input (dq_qnet_set_patch_input): rows of 4 * patch_stride bytes, a patch image of 32 / 48 columns, the byte table behind the tables
static bool plan_conv_bwd(const dq_qnet* Q, ConvBwdPlan* P, bool patch = false) {
    if (Q->cfg.n_conv != 3) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    if (L1.cout != 64 || L1.K > 96) return false;
    if (L2.cin != 64 || L2.cout != 32 || L2.k != 2 || L2.s != 1) return false;
    if (L3.cin != 32 || L3.cout != 32 || L3.k != 2 || L3.s != 1) return false;
    if (patch && !Q->patch_depth) return false;
    P->KG1 = patch ? (Q->patch_kd + 5 + 15) / 16 : (L1.K + 15) / 16;
    if (!patch && P->KG1 < 3) P->KG1 = 3;
    if (patch && P->KG1 < 2) P->KG1 = 2;
    const int in_bytes = Q->cfg.in_c * Q->cfg.in_h * Q->cfg.in_w;
    P->slot = patch ? 4 * Q->patch_stride : (in_bytes + 3 + 3) & ~3;
    if (P->slot > 4 * CB_THREADS) return false;                      // one dword of an observation per thread
    for (int pass = 0; pass < 8; ++pass) {                             // S = 8 double-buffered, S = 8 single, S = 4 double, ...
        const int S = 8 >> (pass >> 1), nbuf = 2 - (pass & 1);
        size_t off = up16((size_t)S * P->slot);
        P->off_mis = (int)off; off += up16((size_t)S * 4);
        off = (off + 1023) & ~(size_t)1023;                                            // LDS-DMA target: whole 1 KB chunks
        const size_t a1_bytes = 2 * (((size_t)S * L1.rows * A1PS * 2 + 1023) & ~(size_t)1023);      // two piece planes, whole 1 KB chunks each
        P->off_a1 = (int)off; off += nbuf * a1_bytes;
        P->a1_alt = nbuf == 2 ? (int)a1_bytes : 0;
        P->off_a2 = (int)off; off += up16((size_t)2 * (S * L2.rows + 1) * PL32 * 2);      // a2 / g2: two f16 piece planes, rows of PL32 halves
        P->off_g3 = (int)off; off += up16((size_t)2 * (S * L3.rows + 1) * PL32 * 2);      // g3: two f16 piece planes
        P->off_t1 = (int)off; off += up16((size_t)S * L1.rows * 16 * P->KG1);      // observation patch image (bytes)
        P->off_t2 = (int)off; off += up16((size_t)S * L2.rows * 4);
        P->off_t3 = (int)off; off += up16((size_t)S * L3.rows * 4);
        P->off_ko = (int)off; off += 96 * 4;
        P->off_tp = (int)off; off += up16((size_t)S * L1.rows * 4);
        P->off_d2 = (int)off; off += up16((size_t)S * L2.rows * 4);
        P->off_d1 = (int)off; off += up16((size_t)S * L1.rows * 4);
        P->off_lut = (int)off; off += patch ? 2048 : 0;
        if (off < 8192 + 48 * 64 * 4) off = 8192 + 48 * 64 * 4;     // (the end of the kernel reuses the first 20 KB for its reductions)
        if (off <= CHAIN_LDS_MAX && S * L1.rows * 16 <= 7 * CB_THREADS && S * L1.rows <= CONV_ROWTAB && S * L2.rows < 65536 && L1.oh < 256 && L1.ow < 256) {
            P->S = S; P->lds = off; return true;
        }
    }
    return false;
}

// Row tables of the fused conv backward (5 x CONV_ROWTAB ints; qnet.hip uploads them behind the forward's at creation), rows m of a
// group of S samples:  [0] first-convolution output pixel: s << 16 | byte offset of its patch inside the observation;  [1] second-
// convolution output pixel: float offset of the a1 row under it;  [2] third: float offset of the a2 row;  [3] / [4] dgrad_inplace's
// tables of the g2 / g1 phases: row of the gradient image at the input pixel's own position | iy << 16 | ix << 24.
bool fused_conv_bwd_row_tables(const dq_qnet* Q, int* tab) {
    ConvBwdPlan P;
    if (!plan_conv_bwd(Q, &P)) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    memset(tab, 0, sizeof(int) * 5 * CONV_ROWTAB);
    // (filled for the largest group, 8 samples: an entry does not depend on the group size, so every plan reads a prefix)
    auto rows_of = [](const Layer& L) { return 8 * L.rows < CONV_ROWTAB ? 8 * L.rows : CONV_ROWTAB; };
    for (int m = 0; m < rows_of(L1); ++m) {
        const int s = m / L1.rows, p = m % L1.rows, oy = p / L1.ow, ox = p % L1.ow;
        tab[m] = s << 16 | (oy * L1.s * L1.iw + ox * L1.s);
        tab[4 * CONV_ROWTAB + m] = (s * L2.rows + oy * L2.ow + ox) | oy << 16 | ox << 24;      // a1 pixel (oy, ox): row of g2 at its own position
    }
    for (int m = 0; m < rows_of(L2); ++m) {
        const int s = m / L2.rows, p = m % L2.rows, oy = p / L2.ow, ox = p % L2.ow;
        tab[CONV_ROWTAB + m] = (s * L1.rows + oy * L1.ow + ox) * A1PS;
        tab[3 * CONV_ROWTAB + m] = (s * L3.rows + oy * L3.ow + ox) | oy << 16 | ox << 24;       // a2 pixel: row of g3
    }
    for (int m = 0; m < rows_of(L3); ++m) {
        const int s = m / L3.rows, p = m % L3.rows, oy = p / L3.ow, ox = p % L3.ow;
        tab[2 * CONV_ROWTAB + m] = (s * L2.rows + oy * L2.ow + ox) * PL32;
    }
    return true;
}

#define CONV_BWD_MAX_WGS 256

bool fused_backward_supported(const dq_qnet* Q) {
    DenseBwdPlan dp;
    ConvBwdPlan cp;
    return fused_forward_supported(Q) && plan_dense_bwd(Q, &dp) && plan_conv_bwd(Q, &cp);
}

// stride (floats) between the dense weight-gradient partials: the parameter count rounded up to whole 128-byte lines, so that every slice starts
// on one (the final reduction reads 16 bytes per lane)
static inline size_t dense_pstride(const dq_qnet* Q) { return (Q->n_params + 31) & ~(size_t)31; }
// floats of workspace the fused backward needs: [DENSE_WGRAD_SLICES][n_params] dense partials, then [CONV_BWD_MAX_WGS][conv params]
size_t fused_backward_workspace_floats(const dq_qnet* Q) {
    if (!fused_backward_supported(Q)) return 0;
    return (size_t)DENSE_WGRAD_SLICES * dense_pstride(Q) + (size_t)CONV_BWD_MAX_WGS * Q->L[Q->cfg.n_conv].w_off + 8;     // + {S, 1/S}: GradScale, the range flag, a spare word, td_scale_kernel's two work words, two spare
}

// the range guard's flag word (include/deepq_hip.h dq_qnet_range_check): the third of the four words behind the partials
bool conv_bwd16_applies(const dq_qnet* Q, int B, bool patch) {
    return patch && conv_bwd16_supported(Q) && B % 8 == 0 && (B >= 1024 || Q->conv_bwd_form == 2) && Q->conv_bwd_form != 1;
}

unsigned* fused_range_flag(const dq_qnet* Q) {
    if (!Q->fpartial) return nullptr;
    return reinterpret_cast<unsigned*>(Q->fpartial + (size_t)DENSE_WGRAD_SLICES * dense_pstride(Q) + (size_t)CONV_BWD_MAX_WGS * Q->L[Q->cfg.n_conv].w_off) + 2;
}

typedef void (*conv_bwd_kernel_t)(ConvBwdArgs);

// phases: bit 0 = dense part (data gradients, dense weight gradients reduced into grads_dev[conv params ..)), bit 1 = convolutional
// part (grads_dev[0 .. conv params)).  3 = whole backward with one reduction launch.
dq_status fused_backward(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, int phases, hipStream_t st,
                         const AdamOpt* opt, const TdFused* td, const EnvParams* rider, size_t rider_lds) {
    DQ_REQUIRE(!rider || (td && (phases & 1)), DQ_ERR_INVALID, "fused_backward: the environment step rides on the TD launch");
    DQ_REQUIRE(!td || (phases & 1), DQ_ERR_INVALID, "fused_backward: the TD step belongs to the dense phase");
    DQ_REQUIRE(!opt || phases == 3, DQ_ERR_INVALID, "fused_backward: the fused optimizer step needs the whole backward in one call");
    DenseBwdPlan dp;
    ConvBwdPlan cp;
    const bool patch = Q->last_patch != 0;                          // the training forward read patch words: so does the first kernel's weight gradient
    DQ_REQUIRE(!patch || Q->patch_depth, DQ_ERR_STATE, "fused_backward: the training forward read patch words, dq_qnet_set_patch_input was reset since");
    DQ_REQUIRE(plan_dense_bwd(Q, &dp) && plan_conv_bwd(Q, &cp, patch), DQ_ERR_UNSUPPORTED, "fused_backward: configuration not covered");
    DQ_REQUIRE((reinterpret_cast<uintptr_t>(params_dev) & 15) == 0, DQ_ERR_INVALID, "fused_backward: params_dev must be 16-byte aligned");
    DQ_REQUIRE(Q->fpartial, DQ_ERR_STATE, "fused_backward: workspace missing");
    static unsigned long long attr_devs = 0;                          // per device (common.h dq_device_bit)
    const unsigned long long dev_bit = dq_device_bit();
    if (!(attr_devs & dev_bit)) {
        const conv_bwd_kernel_t cks[6] = {conv_bwd_chain_kernel<3>, conv_bwd_chain_kernel<4>, conv_bwd_chain_kernel<5>, conv_bwd_chain_kernel<6>,
                                          conv_bwd_chain_kernel<2, true>, conv_bwd_chain_kernel<3, true>};
        for (int i = 0; i < 6; ++i)
            DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cks[i]), hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_MAX));
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wgrad_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_WGRAD_LDS));
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dense_wgrad_ride_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DENSE_WGRAD_LDS));
        attr_devs |= dev_bit;
    }
    const int B = Q->last_train_batch, nc = Q->cfg.n_conv, nl = Q->n_layers;
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    // ---- 0. transposed weights -------------------------------------------------------------------------------------
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    float* dense_partial = Q->fpartial;
    float* conv_partial = dense_partial + (size_t)DENSE_WGRAD_SLICES * dense_pstride(Q);
    float* gs_slot = conv_partial + (size_t)CONV_BWD_MAX_WGS * Q->L[Q->cfg.n_conv].w_off;       // device-computed {S, 1/S}
    DQ_REQUIRE(Q->last_train_packed, DQ_ERR_STATE, "fused_backward: the training forward left no packed weights");
    const u32x4* pkbase = static_cast<const u32x4*>(Q->last_train_packed);
    const PackLayout PL = fused_pack_layout(Q);
    const size_t conv_floats = D1.w_off;
    const int n_dense = (int)(Q->n_params - conv_floats);
    if (phases & 1) {
    // ---- gradient scale (see grad_scale_kernel): host-known with the TD step fused in, else from max |dq| on the device ----------
    const float known = td ? td->grad_scale : Q->grad_scale_hint;  // the loss scale dq carries, when the host knows it
    if (td && td->auto_scale) {                                     // measured on the device from this minibatch's TD errors (td_scale_kernel)
        Q->bwd_scale = 0.f;                                         // = read gs_slot
        TdScaleArgs ta;
        ta.td = *td; ta.B = B; ta.A = Q->cfg.n_actions; ta.out = gs_slot; ta.work = reinterpret_cast<unsigned*>(gs_slot) + 4;
        td_scale_kernel<<<(B + 16 * TDS_ROWS - 1) / (16 * TDS_ROWS), 1024, 0, st>>>(ta);
        DQ_LAUNCH_CHECK();
    } else if (td || known > 0.f) {
        int e = 0;
        (void)frexp((double)known, &e);                             // grad_scale = f 2^e, f in [0.5, 1)
        Q->bwd_scale = known > 0.f ? (float)ldexp(1.0, 3 - e) : 1.f;
    } else {
        Q->bwd_scale = 0.f;                                         // = read gs_slot
        grad_scale_kernel<<<1, 1024, 0, st>>>(dq_dev, B * Q->cfg.n_actions, gs_slot);
        DQ_LAUNCH_CHECK();
    }
    // ---- 1. dense data gradients ----------------------------------------------------------------------------------
    DenseBwdArgs da;
    memset(&da, 0, sizeof(da));
    da.gs = Q->bwd_scale; da.gs_dev = Q->bwd_scale > 0.f ? nullptr : gs_slot;
    da.packed = pkbase; da.pk_dense2t = (int)PL.dense2t; da.pk_dense1t = (int)PL.dense1t; da.KB2 = PL.KB2;
    da.plane_rows = Q->cfg.max_batch; da.small_ld = dq_planes_small_ld(Q);
    da.gh1_pl = dq_plane(Q, 2); da.gy2_pl = dq_plane(Q, 3); da.g3_pl = dq_plane(Q, 4);
    da.params = params_dev; da.dq = dq_dev; da.h1_pl = dq_plane(Q, 1); da.x = Q->act[0][nc - 1];
    da.batch = B; da.K1 = D1.nin; da.perm_hw = Q->flat_hw; da.perm_c = Q->flat_c;
    da.N2 = D2.nout; da.N3 = Q->cfg.dueling ? Q->L[nc + 2].nout : 0; da.n_actions = Q->cfg.n_actions;
    for (int l = 0; l < nl - nc; ++l) da.w_off[l] = (int)Q->L[nc + l].w_off;
    da.mask_scale = D1.dropout > 0.f ? (float)(1.0 / (1.0 - (double)D1.dropout)) : 1.f;
    da.gx_pl = reinterpret_cast<unsigned short*>(Q->gz[nc - 1]); da.gx_lo = (size_t)Q->cfg.max_batch * D1.nin;
    da.ldg = dp.ldg; da.off_g3 = dp.off_g3; da.off_gy2 = dp.off_gy2; da.off_gh1 = dp.off_gh1; da.off_w3t = dp.off_w3t;
    da.pk_w3q = (int)PL.w3q; da.w3q_rows = PL.w3q_rows; da.w3q_pw = 16 * PL.NT2; da.pk_wc = (int)PL.wc;
    da.dense_tiles = (B + DENSE_ROWS - 1) / DENSE_ROWS;
    {
        static const int forced = getenv("DQ_DENSE_BWD_SPLIT") ? atoi(getenv("DQ_DENSE_BWD_SPLIT")) : 0;      // (A/B runs: 1, 2, 4)
        static int n_cu = 0;
        if (!n_cu) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) n_cu = prop.multiProcessorCount;
            if (n_cu <= 0) n_cu = 256;
        }
        da.col_split = 4 * da.dense_tiles <= n_cu ? 4 : 2 * da.dense_tiles <= n_cu ? 2 : 1;          // (8 measured as 4)
        if (forced == 1 || forced == 2 || forced == 4) da.col_split = forced;
    }
    da.dense_wgs = da.dense_tiles * da.col_split;
    da.skip_word = reinterpret_cast<unsigned*>(gs_slot) + 3; da.skip_tag = ++Q->bwd_serial ? Q->bwd_serial : ++Q->bwd_serial;      // (never 0: the word's idle value)
    int stat_wgs = 0;
    if (td) {
        da.td_on = 1; da.td = *td;
        if (td->st_n > 0) stat_wgs = (td->st_n + DENSE_THREADS - 1) / DENSE_THREADS;
    }
    EnvParams ep;
    memset(&ep, 0, sizeof(ep));
    size_t lds = dp.lds;
    const bool ride_wgrad = rider && fused_rider_threads() == 256;  // the step rides on the dense weight gradients' launch instead (below)
    if (rider) {                                                    // (its blocks do their own bookkeeping: no statistics workgroups)
        DQ_REQUIRE(!td->st_n, DQ_ERR_INVALID, "fused_backward: the riding environment step does its own episode bookkeeping");
        if (!ride_wgrad) {
            ep = *rider; da.env_on = 1; stat_wgs = ep.env_blocks + ep.s_blocks;
            if (rider_lds > lds) lds = rider_lds;
        }
    }
    if (td && td->metrics && da.dense_tiles > td->metric_slots)    // (the partials are then summed by atomics, in any order: diagnostics only)
        DQ_HIP(hipMemsetAsync(td->metrics + 2, 0, (size_t)td->metric_slots * 2 * sizeof(float), st));
    {
        // (the TD launch of a dueling network takes the shortcut through the tables pack_weights_kernel built -- W3'^T, Wc --: |A| <= 64 with W3'^T in LDS,
        // wider ones with its row read from L2)
        static const bool wide_off = getenv("DQ_DENSE_BWD_WIDE_SHORT") && getenv("DQ_DENSE_BWD_WIDE_SHORT")[0] == '0';      // (A/B runs)
        const bool wide_short = td && da.N3 > 0 && PL.wc_rows > 0 && !wide_off;
        void (*dbk)(DenseBwdArgs, EnvParams) = dp.NT2 == 4 ? (td ? dense_bwd_chain_kernel<4, true, true> : dense_bwd_chain_kernel<4, false, true>)
                                                            : (td ? (wide_short ? dense_bwd_chain_kernel<7, true, true> : dense_bwd_chain_kernel<7, true, false>)
                                                                  : dense_bwd_chain_kernel<7, false, false>);
        dq_launch(DQ_K_DENSE_BWD, "dense_bwd_chain_kernel", dbk, dim3(da.dense_wgs + stat_wgs), dim3(DENSE_THREADS), lds, st, da, ep);
    }
    DQ_LAUNCH_CHECK();

    // ---- 2. dense weight gradients: all layers, one launch -----------------------------------------------------------
    DenseWgradArgs wa;
    memset(&wa, 0, sizeof(wa));
    wa.n_layers = nl - nc; wa.batch = B;
    int tiles = 0;
    for (int l = 0; l < nl - nc; ++l) {
        const Layer& L = Q->L[nc + l];
        WgradLayer& W = wa.L[l];
        W.X.cols = L.K; W.X.ld = L.K; W.G.cols = L.N; W.G.ld = L.N;
        const size_t mb = (size_t)Q->cfg.max_batch;
        // every operand comes as piece planes (qnet.h dq_plane): x / gH1, h1 / gY2, y2 / g3; the small ones in rows of small_ld halves
        const size_t sl = (size_t)dq_planes_small_ld(Q);
        W.X.f32 = nullptr; W.G.f32 = nullptr;
        if (l == 0) {
            W.X.planes = dq_plane(Q, 0); W.X.plane_stride = mb * D1.nin; W.G.planes = dq_plane(Q, 2); W.G.plane_stride = mb * DENSE_HID;
        } else if (l == 1) {
            W.X.planes = dq_plane(Q, 1); W.X.plane_stride = mb * DENSE_HID; W.G.planes = dq_plane(Q, 3); W.G.plane_stride = mb * sl; W.G.ld = (int)sl;
        } else {
            W.X.planes = dq_plane(Q, 5); W.X.plane_stride = mb * sl; W.X.ld = (int)sl; W.G.planes = dq_plane(Q, 4); W.G.plane_stride = mb * sl; W.G.ld = (int)sl;
        }
        W.out_w = (int)L.w_off; W.out_b = (int)L.b_off;
        if (l == 0) { W.perm_hw = Q->flat_hw; W.perm_c = Q->flat_c; }
        W.k_tiles = (L.K + 63) / 64; W.n_tiles = (L.N + 63) / 64;
        W.tile0 = tiles; tiles += W.k_tiles * W.n_tiles;
    }
    int rps, sy;
    wgrad_slicing(Q, B, &rps, &sy);
    wa.rows_per_slice = rps; wa.partial = dense_partial; wa.pstride = dense_pstride(Q);
    wa.total_tiles = tiles; wa.slices = sy;
    wa.wg_count = tiles * sy;
    EnvParams wep;
    memset(&wep, 0, sizeof(wep));
    int ride_wgs = 0;
    if (ride_wgrad) {
        DQ_REQUIRE(rider_lds <= DENSE_WGRAD_LDS, DQ_ERR_UNSUPPORTED, "fused_backward: the riding environment step needs more LDS than the dense weight gradients' launch has");
        wep = *rider; wa.env_on = ride_on_wgrad(); ride_wgs = wep.env_blocks + wep.s_blocks; wa.env_wgs = ride_wgs;
    }
    if (ride_wgrad) dq_launch(DQ_K_DENSE_WGRAD, "dense_wgrad_ride_kernel", dense_wgrad_ride_kernel, dim3(tiles * sy + ride_wgs), dim3(WGRAD_THREADS), DENSE_WGRAD_LDS, st, wa, wep);
    else dq_launch(DQ_K_DENSE_WGRAD, "dense_wgrad_kernel", dense_wgrad_kernel, dim3(tiles * sy), dim3(WGRAD_THREADS), DENSE_WGRAD_LDS, st, wa);
    DQ_LAUNCH_CHECK();

    if (phases != 3) {                                              // phased: the dense gradients are complete (and reducible) now
        ReduceArgs ra;
        memset(&ra, 0, sizeof(ra));
        ra.inv_gs = Q->bwd_scale > 0.f ? 1.f / Q->bwd_scale : 0.f; ra.gs_dev = Q->bwd_scale > 0.f ? nullptr : gs_slot;
        ra.range_flag = reinterpret_cast<unsigned*>(gs_slot) + 2; ra.skip_word = reinterpret_cast<unsigned*>(gs_slot) + 3; ra.skip_tag = Q->bwd_serial;
        ra.seg[0] = {dense_partial + conv_floats, grads_dev + conv_floats, n_dense, sy, dense_pstride(Q), 0, (int)conv_floats};
        ra.seg[0].vec = sy <= 16 && (conv_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(grads_dev) & 15) == 0;      // (the vector path: same bits, see the kernel)
        ra.seg[1] = ra.seg[0]; ra.seg[1].block0 = 0x7fffffff;
        reduce_slices_kernel<<<ra.seg[0].vec ? (n_dense + 2047) / 2048 : (n_dense + 63) / 64, dim3(64, 8), 0, st>>>(ra);
        DQ_LAUNCH_CHECK();
    }
    }
    if (!(phases & 2)) return DQ_OK;
    // ---- 3. convolutions: data + weight gradients, one persistent launch -----------------------------------------------
    ConvBwdArgs ca;
    memset(&ca, 0, sizeof(ca));
    ca.params = params_dev; ca.obs = Q->last_obs; ca.index = Q->last_index; ca.index_off = Q->last_index_off;
    ca.index_mod = Q->last_index_mod > 0 ? Q->last_index_mod : 0x7fffffff;
    ca.a1p = reinterpret_cast<const unsigned short*>(Q->act[0][0]); ca.a1_lo = (size_t)Q->cfg.max_batch * L1.rows * 64;
    ca.a2p = reinterpret_cast<const unsigned short*>(Q->act[0][1]); ca.a2_lo = (size_t)Q->cfg.max_batch * L2.rows * 32;
    ca.g3p = reinterpret_cast<const unsigned short*>(Q->gz[nc - 1]); ca.g3_lo = (size_t)Q->cfg.max_batch * L3.rows * 32;
    DQ_REQUIRE(Q->last_train_packed, DQ_ERR_STATE, "fused_backward: the training forward left no packed weights");
    ca.packed = static_cast<const u32x4*>(Q->last_train_packed);
    // samples per group: the plan's (8 where LDS allows) -- or fewer when that leaves most CUs without a group (a group is a chain of ten barrier-
    // delimited phases: a lone workgroup per CU is latency-bound, so a small minibatch is better spread thin; the tables are sample-major
    // and every loop runs over the group's own rows, so a smaller S is a prefix of the planned layout)
    int S_run = cp.S;
    {
        static const int forced = getenv("DQ_CONV_BWD_S") ? atoi(getenv("DQ_CONV_BWD_S")) : 0;      // (A/B runs)
        while (S_run > 2 && 2 * ((B + S_run - 1) / S_run) <= CONV_BWD_MAX_WGS) S_run >>= 1;
        if (forced >= 1 && forced <= cp.S) S_run = forced;
    }
    ca.batch = B; ca.S = S_run; ca.groups = (B + S_run - 1) / S_run;
    ca.C = L1.cin; ca.H = L1.ih; ca.W = L1.iw; ca.k1 = L1.k; ca.st1 = L1.s; ca.K1 = L1.K;
    ca.oh1 = L1.oh; ca.ow1 = L1.ow; ca.oh2 = L2.oh; ca.ow2 = L2.ow; ca.oh3 = L3.oh; ca.ow3 = L3.ow;
    for (int l = 0; l < 3; ++l) { ca.w_off[l] = (int)Q->L[l].w_off; ca.b_off[l] = (int)Q->L[l].b_off; }
    ca.partial = conv_partial; ca.pstride = conv_floats;
    ca.slot = cp.slot; ca.off_mis = cp.off_mis; ca.off_a1 = cp.off_a1; ca.a1_alt = cp.a1_alt; ca.off_a2 = cp.off_a2; ca.off_g3 = cp.off_g3;
    ca.off_t1 = cp.off_t1; ca.off_t2 = cp.off_t2; ca.off_t3 = cp.off_t3; ca.off_ko = cp.off_ko; ca.off_tp = cp.off_tp; ca.kofftab = Q->kofftab;
    ca.off_d2 = cp.off_d2; ca.off_d1 = cp.off_d1; ca.rowtab = Q->kofftab + 96 + CONV_FWD_TABS * CONV_ROWTAB;
    ca.rowtab1 = patch ? Q->ptab + PT_BWD : ca.rowtab; ca.srctab = patch ? Q->ptab + PT_SRC : nullptr; ca.kd = Q->patch_kd; ca.off_lut = cp.off_lut;
    DQ_REQUIRE(!patch || (reinterpret_cast<uintptr_t>(ca.obs) & 15) == 0, DQ_ERR_INVALID, "fused_backward: patch-word rows must be 16-byte aligned");
    // the 16-wave form (conv_bwd16.hip) where it applies and the minibatch gives every CU work (groups of 8 samples); conv_bwd_form 1: this file's kernel,
    // 2: the 16-wave form whatever the minibatch (dq_qnet_set_kernel_forms; DQ_CONV_BWD_FORM=8 / 16 at creation)
    const bool form16 = conv_bwd16_applies(Q, B, patch);
    // a1: saved by the training forward, or recomputed here from the patch words when that forward (conv_wave_kernel) left it out -- the two must agree
    DQ_REQUIRE(Q->last_a1_saved || form16, DQ_ERR_STATE, "fused_backward: the training forward did not save the first convolution's output (it expected conv_bwd16_kernel to "
               "recompute it) and this backward takes another form: dq_qnet_set_kernel_forms between a training forward and its backward");
    if (form16) { ca.S = 8; ca.groups = (B + 7) / 8; ca.pk_cdw = (int)PL.cdw; ca.tab16 = Q->ptab + PT_C16; ca.pk_c1w = (int)PL.c1w; ca.a1_recompute = Q->last_a1_saved ? 0 : 1; }
    const int wgs = ca.groups < CONV_BWD_MAX_WGS ? ca.groups : CONV_BWD_MAX_WGS;
    if (form16) {
        const dq_status rc = conv_bwd16_launch(Q, ca, wgs, st);
        if (rc != DQ_OK) return rc;
    } else {
    conv_bwd_kernel_t ck = patch ? (cp.KG1 == 2 ? conv_bwd_chain_kernel<2, true> : conv_bwd_chain_kernel<3, true>)
                         : cp.KG1 == 3 ? conv_bwd_chain_kernel<3> : cp.KG1 == 4 ? conv_bwd_chain_kernel<4>
                         : cp.KG1 == 5 ? conv_bwd_chain_kernel<5> : conv_bwd_chain_kernel<6>;
    dq_launch(DQ_K_CONV_BWD, "conv_bwd_chain_kernel", ck, dim3(wgs), dim3(CB_THREADS), cp.lds, st, ca);
    DQ_LAUNCH_CHECK();
    }
    if (Q->mark_event) {                                            // dq_qnet_mark_conv_backward: the caller's side stream starts from here
        const hipError_t me = hipEventRecord(static_cast<hipEvent_t>(Q->mark_event), st);
        Q->mark_event = nullptr;
        DQ_HIP(me);
    }

    // ---- 4. fixed-order reductions of the partials into the flat gradient ------------------------------------------------
    ReduceArgs ra;
    memset(&ra, 0, sizeof(ra));
    ra.inv_gs = Q->bwd_scale > 0.f ? 1.f / Q->bwd_scale : 0.f; ra.gs_dev = Q->bwd_scale > 0.f ? nullptr : gs_slot;
    ra.range_flag = reinterpret_cast<unsigned*>(gs_slot) + 2; ra.skip_word = reinterpret_cast<unsigned*>(gs_slot) + 3; ra.skip_tag = Q->bwd_serial;
    if (opt) { ra.opt = *opt; ra.adam = 1; }
    ra.seg[0] = {conv_partial, grads_dev, (int)conv_floats, wgs, conv_floats, 0, 0};
    const int blocks0 = ((int)conv_floats + 63) / 64;
    // the next training forward's dropout keep bits, drawn ahead by this launch: the last one's draw at t + 1 (qnet.h keep_bits)
    Q->kb_tag.valid = 0;
    if (Q->keep_bits && Q->last_drop.valid && Q->last_drop.batch <= Q->cfg.max_batch) {
        static const bool ahead_on = !(getenv("DQ_DROP_AHEAD") && getenv("DQ_DROP_AHEAD")[0] == '0');
        if (ahead_on) {
            dq_qnet::DropTag next = Q->last_drop;
            next.t += 1;
            next.stream = (void*)st;                                  // (drawn by THIS launch: only a forward on the same stream is ordered behind it)
            ra.drop = {Q->keep_bits, next.seed0, next.seed1, next.sample_base, next.drop_T, next.t, next.batch, (next.batch * 16 + 511) / 512};
            Q->kb_tag = next;
        }
    }
    if (phases == 3) {
        int rps, sy;
        wgrad_slicing(Q, B, &rps, &sy);
        ra.seg[1] = {dense_partial + conv_floats, grads_dev + conv_floats, n_dense, sy, dense_pstride(Q), blocks0, (int)conv_floats};
        ra.seg[1].vec = sy <= 16 && (conv_floats & 3) == 0 && (reinterpret_cast<uintptr_t>(grads_dev) & 15) == 0 &&
                        (!opt || ((reinterpret_cast<uintptr_t>(opt->p) | reinterpret_cast<uintptr_t>(opt->m) | reinterpret_cast<uintptr_t>(opt->v)) & 15) == 0);          // few slices: four outputs per thread, every slice's 16 bytes in flight at once
        reduce_slices_kernel<<<ra.drop.wgs + blocks0 + (ra.seg[1].vec ? (n_dense + 2047) / 2048 : (n_dense + 63) / 64), dim3(64, 8), 0, st>>>(ra);
    } else {
        ra.seg[1] = ra.seg[0]; ra.seg[1].block0 = 0x7fffffff;
        reduce_slices_kernel<<<ra.drop.wgs + blocks0, dim3(64, 8), 0, st>>>(ra);
    }
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}
