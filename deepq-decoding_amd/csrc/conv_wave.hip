// The three convolutions of the Q-network, wave-private form (round 5): one workgroup of 16 waves per CU, ONE SAMPLE PER WAVE at a time,
// no barrier after the prologue.
//
// Same arithmetic as fused.hip's conv_chain_pkernel<0> (Keras build_convolutional_nn, /root/reference/example_notebooks/Function_Library.py:352-365:
// Conv2D(64, 3, strides=2) - Conv2D(32, 2) - Conv2D(32, 2), ReLU each), f16x2 pieces on the f16 matrix pipe (qnet.h), patch-word observations
// (include/deepq_hip.h dq_env_patch_output).  What changes is who owns what:
//
//   conv_chain_pkernel   workgroup of 4 waves shares 8 samples: three barrier-separated phases per group, every wave streams ALL packed weights
//                        from L2 through a register ring (56 KB per wave and group), row tables map tile rows to (sample, pixel), 193 VGPRs ->
//                        2 waves per SIMD.  Counters (NOTEBOOK.md round 4): MFMA busy 21-26 %, 35 % of the wave cycles parked.
//   conv_wave_kernel     the packed weights of the workgroup's network (56 KB) and the byte -> bits table live in LDS, loaded once; a wave walks its
//                        samples alone: its A operands come from a private LDS image whose geometry is the same for every sample, so every LDS
//                        address is a per-lane constant of the prologue (no row tables, no guards, no address arithmetic in the loop), B operands
//                        are 1 KB lane-ordered blocks read straight from the LDS copy, accumulators <= 128 VGPRs -> 4 waves per SIMD whose
//                        instruction streams are independent (one wave's VALU epilogue runs beside another's MFMAs, LDS latencies are covered
//                        by three other waves instead of by a barrier).
//
// Per sample and wave (d = 5: 25 / 16 / 9 output pixels):
//   conv1   A = one byte of each pixel's patch word (| the pixel's constant cells and the bias bit) expanded through the bits table; ONE K = 32 block
//           whose rows are [K_data data bits | 5 constant positions | bias | 0 ..] (qnet.h c1w: the per-pixel bias of the persistent form is folded
//           into the contraction: no bias registers, no bias add); computed in two halves of 32 output channels: half h goes to the a1 half image,
//           the second convolution consumes it (its K blocks of that channel half), then half 1 takes its place -- a1 costs 4 KB of LDS per wave
//           instead of 8, which is what lets 16 waves fit beside the weights.
//   conv2   per (tap, channel half): A = 2 ds_read_b128 (h / l piece of 8 channels of the tap's pixel), B = 4 ds_read_b128, 6 MFMAs.
//   conv3   the same on the a2 image; its output leaves as f32 rows (the dense chain's input), rows < 9 only.
//   training job: the a1 halves and a2 are copied to the piece planes the convolutional backward reads, 16 bytes per lane, from the LDS images.
//
// LDS (d = 5): [c1w 8 KB | conv2 32 KB | conv3 16 KB | bits table 4 KB | per wave: a1 half (2 planes x 32 rows x 64 B) 4 KB, a2 (2 x 16 x 64 B) 2 KB] = 156 KB.
// Workgroups are dealt to the launch's weight sets in proportion to their samples (a workgroup serves ONE packed buffer); inside a weight set every
// job's samples are split evenly over its workgroups, so the training job's stores are spread over all of them.
#include "qnet.h"

#define CW_THREADS 1024
#define CW_WAVES 16
#define CW_W_C1 0                                   // u32x4 units
#define CW_W_C2 (4 * PK_BLOCK)
#define CW_W_C3 (CW_W_C2 + 16 * PK_BLOCK)
#define CW_W_END (CW_W_C3 + 8 * PK_BLOCK)           // 3584 u32x4 = 56 KB
#define CW_LUT_BYTES 4096

template <int D>
struct CwGeo {
    static constexpr int OW1 = D, OW2 = D - 1, OW3 = D - 2, R1 = D * D, R2 = OW2 * OW2, R3 = OW3 * OW3;
    static constexpr int T1 = (R1 + 15) / 16, T2 = (R2 + 15) / 16, T3 = (R3 + 15) / 16;
    static constexpr int A1_PLANE = T1 * 16 * 64, A2_PLANE = T2 * 16 * 64;          // bytes per piece plane (rows of 32 halves)
    static constexpr int WAVE_BYTES = 2 * A1_PLANE + 2 * A2_PLANE;
    static constexpr int W_BYTES = CW_W_END * 16;
    static constexpr size_t LDS = (size_t)W_BYTES + CW_LUT_BYTES + (size_t)CW_WAVES * WAVE_BYTES;
};

template <int D>
__global__ __launch_bounds__(CW_THREADS) void conv_wave_kernel(ConvWaveArgs a) {
    using G = CwGeo<D>;
    constexpr int T1 = G::T1, T2 = G::T2, T3 = G::T3, R1 = G::R1, R2 = G::R2, R3 = G::R3;
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kb = lane >> 4;
    static_assert(FWD_MAX_JOBS == 4, "three comparisons");

    // ---- this workgroup's weight set and its share of every job of that set (block-uniform scalars) ------------------------------------------------
    const int w = (int)blockIdx.x;
    int kc = 0;
#pragma unroll
    for (int k = FWD_MAX_JOBS - 1; k >= 0; --k)
        if (k < a.n_jobs && w >= a.cls_wg0[k] && w < a.cls_wg0[k] + a.cls_wgs[k]) kc = k;
    const int wg0 = a.cls_wg0[kc], rk = w - wg0;
    int cnt[FWD_MAX_JOBS], first[FWD_MAX_JOBS];
#pragma unroll
    for (int k = 0; k < FWD_MAX_JOBS; ++k) {
        const bool in = k < a.n_jobs && a.cls_wg0[k] == wg0;
        cnt[k] = in ? a.q[k] + (rk < a.m[k] ? 1 : 0) : 0;
        first[k] = rk * a.q[k] + min(rk, a.m[k]);
    }
    const int c0 = cnt[0], c1 = c0 + cnt[1], c2 = c1 + cnt[2], total = c2 + cnt[3];
    if (total == 0) return;                                         // block-uniform

    // ---- prologue: weights and bits table into LDS, per-lane constants -------------------------------------------------------------------------------
    {
        const u32x4* pk = a.job[kc].packed;
        for (int i = wave; i < CW_W_END / 64; i += CW_WAVES) {      // 1 KB units: [c1w 8 | conv2 32 | conv3 16] (PK_CONV2_FWD .. PK_CONV3_FWD are contiguous)
            const u32x4* src = (i < 8 ? pk + a.pk_c1w + 64 * i : pk + PK_CONV2_FWD + 64 * (i - 8)) + lane;
            lds_dma16(src, (u32)(1024 * i));
        }
        static_assert(PK_CONV3_FWD == PK_CONV2_FWD + 16 * PK_BLOCK && PK_CONV2_FWD == 0, "one contiguous source range");
    }
    u32x4* s_w = reinterpret_cast<u32x4*>(smem);
    u32x4* s_lut = reinterpret_cast<u32x4*>(smem + G::W_BYTES);
    if (tid < 256) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (((u32)tid >> (2 * q)) & 1u) * 0x3c00u | (((u32)tid >> (2 * q + 1)) & 1u) * 0x3c000000u;
        s_lut[tid] = v;
    }
    const u32 wbase = (u32)(G::W_BYTES + CW_LUT_BYTES + wave * G::WAVE_BYTES);      // this wave's images: a1 half h / l planes, then a2 h / l
    u8* const s_a1 = smem + wbase;
    u8* const s_a2 = smem + wbase + 2 * G::A1_PLANE;
    // conv1: byte of pixel p's word this lane expands (tile u row j -> pixel min(16u + j, R1 - 1)); the constant cells and the bias bit OR-ed in
    int ob[T1];
    u32 orc[T1];
#pragma unroll
    for (int u = 0; u < T1; ++u) {
        const int p = min(16 * u + j, R1 - 1);
        ob[u] = 4 * p + kb;
        const u32 cb = ((u32)a.ptab[PT_CONST + p] << a.kd) | (1u << (a.kd + 5));
        orc[u] = (cb >> (8 * kb)) & 0xffu;
    }
    // output rows of this lane in a 16-row tile: 4kb + r, its two adjacent channels 2j, 2j + 1 of the 32: byte offset inside a plane
    const int wrow = (4 * kb) * 64 + 4 * j;
    // conv2 / conv3 A operand: tile row j = output pixel n (clamped past the layer's pixels), 8 channels 8kb .. 8kb + 7 of input pixel (y + ky, x + kx)
    int ra2[T2][4], ra3[T3][4];
#pragma unroll
    for (int u = 0; u < T2; ++u) {
        const int n = 16 * u + j < R2 ? 16 * u + j : 0, y = n / G::OW2, x = n - y * G::OW2;
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) ra2[u][tap] = ((y + (tap >> 1)) * G::OW1 + x + (tap & 1)) * 64 + 16 * kb;
    }
#pragma unroll
    for (int u = 0; u < T3; ++u) {
        const int n = 16 * u + j < R3 ? 16 * u + j : 0, y = n / G::OW3, x = n - y * G::OW3;
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) ra3[u][tap] = ((y + (tap >> 1)) * G::OW2 + x + (tap & 1)) * 64 + 16 * kb;
    }
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): this wave's weight copies have landed
    __syncthreads();

    // ---- items of this wave: it = wave, wave + 16, ... ------------------------------------------------------------------------------------------------
    auto job_of = [&](int it) { return (it >= c0) + (it >= c1) + (it >= c2); };
    auto sample_of = [&](int it, int k) { return it - (k == 0 ? 0 : k == 1 ? c0 : k == 2 ? c1 : c2) + (k == 0 ? first[0] : k == 1 ? first[1] : k == 2 ? first[2] : first[3]); };
    auto row_of = [&](int it) {                                     // replay row of item `it` (clamped to the last item): a scalar load through the index vector
        it = min(it, total - 1);
        const int k = job_of(it), b = sample_of(it, k);
        const ConvJob& J = a.job[k];
        const __attribute__((address_space(4))) int32_t* idx = (const __attribute__((address_space(4))) int32_t*)(uintptr_t)J.index;
        int row = b;
        if (J.index) { row = idx[b] + J.index_off; if (row >= J.index_mod) row -= J.index_mod; }
        return row;
    };
    auto load_bytes = [&](int it, int row, u32 (&by)[T1]) {
        const ConvJob& J = a.job[job_of(min(it, total - 1))];
        const u8* src = J.obs + (size_t)row * a.slot;
#pragma unroll
        for (int u = 0; u < T1; ++u) by[u] = src[ob[u]];
    };
    int it = wave;
    if (it >= total) return;                                        // wave-uniform; no barrier follows
    u32 by[T1], byn[T1];
    load_bytes(it, row_of(it), by);
    int row_n = row_of(it + CW_WAVES);
    for (;;) {
        const int k = job_of(it), b = sample_of(it, k);
        const ConvJob& J = a.job[k];
        load_bytes(it + CW_WAVES, row_n, byn);                      // the next item's bytes and the row behind it fly over this item
        row_n = row_of(it + 2 * CW_WAVES);
        const f32x2 bias2 = *reinterpret_cast<const f32x2*>(J.params + a.b_off[1] + 2 * j);
        const f32x2 bias3 = *reinterpret_cast<const f32x2*>(J.params + a.b_off[2] + 2 * j);
        const bool train = J.write_all != 0;                        // wave-uniform

        u32x4 A1[T1];
#pragma unroll
        for (int u = 0; u < T1; ++u) A1[u] = s_lut[by[u] | orc[u]];
        f32x4 acc2[T2][2][2];
#pragma unroll
        for (int u = 0; u < T2; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc2[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[u][t][1] = acc2[u][t][0]; }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            // ---- conv1, output channels 32 hf .. 32 hf + 31 ---------------------------------------------------------------------------------------------
            u32x4 b1h[2], b1l[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                b1h[t] = s_w[CW_W_C1 + (2 * hf + t) * PK_BLOCK + lane];
                b1l[t] = s_w[CW_W_C1 + (2 * hf + t) * PK_BLOCK + PK_LO + lane];
            }
            u32 hp[T1][4], lp[T1][4];
#pragma unroll
            for (int u = 0; u < T1; ++u) {
                f32x4 vs[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const f32x4 ah = MFMA_F16(A1[u], b1h[t], z), al = MFMA_F16(A1[u], b1l[t], z);
                    vs[t] = f16x2_sum(ah, al);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) split_f16x2_pair(relu1(vs[0][r]), relu1(vs[1][r]), hp[u][r], lp[u][r]);
            }
            // the half image (the second convolution's reads of the previous half are older LDS operations of this wave: in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int u = 0; u < T1; ++u)
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    *reinterpret_cast<u32*>(s_a1 + u * 1024 + wrow + 64 * r) = hp[u][r];
                    *reinterpret_cast<u32*>(s_a1 + G::A1_PLANE + u * 1024 + wrow + 64 * r) = lp[u][r];
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (train) {                                            // a1 piece planes [sample pixel][64 halves]: this half's 64 bytes of every row, 16 bytes per lane
#pragma unroll
                for (int piece = 0; piece < 2; ++piece) {
                    unsigned short* dst = J.a1_pl + piece * J.a1_lo + (size_t)b * R1 * 64 + 32 * hf;
#pragma unroll
                    for (int i = 0; 64 * i < 4 * R1; ++i) {
                        const int L = 64 * i + lane, p = L >> 2, c = L & 3;
                        if (L < 4 * R1) *reinterpret_cast<u32x4*>(dst + p * 64 + 8 * c) = *reinterpret_cast<const u32x4*>(s_a1 + piece * G::A1_PLANE + p * 64 + 16 * c);
                    }
                }
            }
            // ---- conv2: the four taps' K blocks of this channel half -----------------------------------------------------------------------------------
#pragma unroll
            for (int tap = 0; tap < 4; ++tap) {
                const int blk = 2 * tap + hf;
                F16x2 bw[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bw[t].h = s_w[CW_W_C2 + (2 * blk + t) * PK_BLOCK + lane];
                    bw[t].l = s_w[CW_W_C2 + (2 * blk + t) * PK_BLOCK + PK_LO + lane];
                }
#pragma unroll
                for (int u = 0; u < T2; ++u) {
                    F16x2 av;
                    av.h = *reinterpret_cast<const u32x4*>(s_a1 + ra2[u][tap]);
                    av.l = *reinterpret_cast<const u32x4*>(s_a1 + G::A1_PLANE + ra2[u][tap]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) mma_f16x3(av, bw[t], acc2[u][t][0], acc2[u][t][1]);
                }
            }
        }
        // ---- conv2 epilogue: a2 image (split on write) --------------------------------------------------------------------------------------------------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
        for (int u = 0; u < T2; ++u) {
            f32x4 vs[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) vs[t] = f16x2_sum(acc2[u][t][0], acc2[u][t][1]) + f32x4{bias2[t], bias2[t], bias2[t], bias2[t]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u32 h, l;
                split_f16x2_pair(relu1(vs[0][r]), relu1(vs[1][r]), h, l);
                *reinterpret_cast<u32*>(s_a2 + u * 1024 + wrow + 64 * r) = h;
                *reinterpret_cast<u32*>(s_a2 + G::A2_PLANE + u * 1024 + wrow + 64 * r) = l;
            }
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (train) {                                                // a2 piece planes [sample pixel][32 halves]
#pragma unroll
            for (int piece = 0; piece < 2; ++piece) {
                unsigned short* dst = J.a2_pl + piece * J.a2_lo + (size_t)b * R2 * 32;
#pragma unroll
                for (int i = 0; 64 * i < 4 * R2; ++i) {
                    const int L = 64 * i + lane;
                    if (L < 4 * R2) *reinterpret_cast<u32x4*>(dst + 8 * L) = *reinterpret_cast<const u32x4*>(s_a2 + piece * G::A2_PLANE + 16 * L);
                }
            }
        }
        // ---- conv3 -> f32 rows [pixel][32] -----------------------------------------------------------------------------------------------------------------
        f32x4 acc3[T3][2][2];
#pragma unroll
        for (int u = 0; u < T3; ++u)
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc3[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc3[u][t][1] = acc3[u][t][0]; }
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            F16x2 bw[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bw[t].h = s_w[CW_W_C3 + (2 * tap + t) * PK_BLOCK + lane];
                bw[t].l = s_w[CW_W_C3 + (2 * tap + t) * PK_BLOCK + PK_LO + lane];
            }
#pragma unroll
            for (int u = 0; u < T3; ++u) {
                F16x2 av;
                av.h = *reinterpret_cast<const u32x4*>(s_a2 + ra3[u][tap]);
                av.l = *reinterpret_cast<const u32x4*>(s_a2 + G::A2_PLANE + ra3[u][tap]);
#pragma unroll
                for (int t = 0; t < 2; ++t) mma_f16x3(av, bw[t], acc3[u][t][0], acc3[u][t][1]);
            }
        }
        float* out = J.act_out[2] + (size_t)b * R3 * 32 + 2 * j;
#pragma unroll
        for (int u = 0; u < T3; ++u) {
            f32x4 vs[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) vs[t] = f16x2_sum(acc3[u][t][0], acc3[u][t][1]) + f32x4{bias3[t], bias3[t], bias3[t], bias3[t]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int n = 16 * u + 4 * kb + r;
                if (n < R3) *reinterpret_cast<f32x2*>(out + n * 32) = f32x2{relu1(vs[0][r]), relu1(vs[1][r])};
            }
        }
        it += CW_WAVES;
        if (it >= total) break;
#pragma unroll
        for (int u = 0; u < T1; ++u) by[u] = byn[u];
    }
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------------
bool conv_wave_supported(const dq_qnet* Q) {
    if (Q->cfg.n_conv != 3 || !Q->patch_depth) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    if (L1.cout != 64 || L1.k != 3 || L1.s != 2 || L1.oh != L1.ow) return false;
    if (L2.cin != 64 || L2.cout != 32 || L2.k != 2 || L2.s != 1) return false;
    if (L3.cin != 32 || L3.cout != 32 || L3.k != 2 || L3.s != 1) return false;
    if (L1.oh != 5) return false;                                   // instantiated geometry
    if (Q->patch_kd + 6 > 32) return false;                         // data bits + 5 constant positions + the bias bit in ONE K = 32 block
    if (4 * Q->patch_stride < 4 * L1.oh * L1.ow) return false;
    return true;
}

// jobs[i].packed groups the jobs into weight sets; the launch's workgroups are dealt to the sets in proportion to their samples
dq_status conv_wave_launch(const dq_qnet* Q, ConvWaveArgs& a, int n_cu, hipStream_t st) {
    static unsigned long long attr_devs = 0;
    const unsigned long long dev_bit = dq_device_bit();
    if (!(attr_devs & dev_bit)) {
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wave_kernel<5>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CwGeo<5>::LDS));
        attr_devs |= dev_bit;
    }
    int cls_of[FWD_MAX_JOBS], n_cls = 0, cls_total[FWD_MAX_JOBS] = {0, 0, 0, 0}, total = 0;
    for (int i = 0; i < a.n_jobs; ++i) {
        cls_of[i] = -1;
        for (int k = 0; k < i; ++k)
            if (a.job[k].packed == a.job[i].packed) { cls_of[i] = cls_of[k]; break; }
        if (cls_of[i] < 0) cls_of[i] = n_cls++;
        cls_total[cls_of[i]] += a.job[i].batch;
        total += a.job[i].batch;
    }
    int grid = total < n_cu ? total : n_cu;
    if (grid < n_cls) grid = n_cls;
    // largest-remainder split of the grid over the weight sets, at least one workgroup each
    int wgs[FWD_MAX_JOBS], given = 0;
    for (int c = 0; c < n_cls; ++c) { wgs[c] = (int)((long long)grid * cls_total[c] / total); if (wgs[c] < 1) wgs[c] = 1; given += wgs[c]; }
    while (given < grid) {
        int best = 0; double need = -1.0;
        for (int c = 0; c < n_cls; ++c) { const double per = (double)cls_total[c] / wgs[c]; if (per > need) { need = per; best = c; } }
        ++wgs[best]; ++given;
    }
    while (given > grid) {
        int best = -1; double need = 1e300;
        for (int c = 0; c < n_cls; ++c) { if (wgs[c] < 2) continue; const double per = (double)cls_total[c] / (wgs[c] - 1); if (per < need) { need = per; best = c; } }
        if (best < 0) break;
        --wgs[best]; --given;
    }
    int wg0[FWD_MAX_JOBS], acc = 0;
    for (int c = 0; c < n_cls; ++c) { wg0[c] = acc; acc += wgs[c]; }
    for (int i = 0; i < FWD_MAX_JOBS; ++i) {
        if (i < a.n_jobs) {
            a.cls_wg0[i] = wg0[cls_of[i]]; a.cls_wgs[i] = wgs[cls_of[i]];
            a.q[i] = a.job[i].batch / wgs[cls_of[i]]; a.m[i] = a.job[i].batch % wgs[cls_of[i]];
        } else { a.cls_wg0[i] = -1; a.cls_wgs[i] = 0; a.q[i] = a.m[i] = 0; }
    }
    dq_launch(DQ_K_CONV_CHAIN, conv_wave_kernel<5>, dim3(acc), dim3(CW_THREADS), CwGeo<5>::LDS, st, a);
    DQ_LAUNCH_CHECK();
    (void)Q;
    return DQ_OK;
}
