// The three convolutions of the Q-network, wave-private form (round 5): one workgroup of 16 waves per CU, TWO SAMPLES PER WAVE at a time,
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
// First version (one sample per wave, a1 in halves of 32 channels): 31.8 us against 41.3 for the step's four forwards, and LDS-bound -- SQ_LDS_IDX_ACTIVE
// 38.9 K cycles per CU of the kernel's 67 K, 56 of a sample's 82 ds_read_b128 being weight blocks that feed 6 MFMAs each.  This version shares every
// weight block between TWO samples (12 MFMAs per 4 block reads) and places the images' 16-byte chunks so that no operand read has a bank conflict:
//
// Per PAIR of samples and wave (d = 5: 25 / 16 / 9 output pixels):
//   conv1   computed TRANSPOSED, one quarter of 16 output channels at a time: first operand = the quarter's weight block (qnet.h c1w; rows =
//           channels), second = one byte of each pixel's patch word (| the pixel's constant cells and the bias bit) expanded through the bits
//           table -- ONE K = 32 block whose rows are [K_data data bits | 5 constant positions | bias | 0 ..]: the per-pixel bias of the
//           persistent form is folded into the contraction.  A lane then holds 4 consecutive channels of one pixel: one 8-byte store per piece.
//           The quarter image (2 samples x 2 pieces x 25 pixels x 32 bytes) is consumed by the second convolution's K blocks of that quarter
//           (qnet.h c2w: a block = the two taps (ky, 0), (ky, 1) x 16 channels) and then overwritten by the next quarter: a1 costs 4 KB of LDS per
//           wave instead of 25.6, which is what lets 16 waves of two samples fit beside the weights.
//   conv2   per (quarter, ky): B = 4 ds_read_b128, A = 2 per sample, 12 MFMAs.  Its output a2 (split on write) OVERLAYS the dead quarter image.
//   conv3   per tap the same on the a2 images; its output leaves as f32 rows (the dense chain's input), rows < 9 only.
//   training job: a2 (and, for a backward that does not recompute it -- ConvJob.write_all bit 1 --, the a1 quarters) are copied to the piece planes the
//           convolutional backward reads, 16 bytes per lane, from the LDS images.
//
// LDS: [c1w 8 KB | c2w 32 KB | conv3 16 KB | bits table 4 KB | per wave 4 KB] = 124 KB.  Inside a wave's 4 KB: sample s at 2048 s, piece l 1024 bytes
// behind piece h, and the 16-byte chunk c of pixel p at slot CW_SLOT1[2 p + c] (quarter image: 2 chunks per pixel) / CW_SLOT2[4 p + c] (a2: 4 chunks)
// -- placements found by annealing (tools/probe/wave_layout.py 5) under which every ds_read_b128 of an operand touches 16 distinct bank quadruples in each
// of its four lane groups (MI355X_MICROARCH.md LDS table) for both taps rows / all four taps; with rows in pixel order every such read took 8 cycles instead of 4.
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
#define CW_WAVE_BYTES 4096
#define CW_LDS (CW_W_END * 16 + CW_LUT_BYTES + CW_WAVES * CW_WAVE_BYTES)

// Development aid (build with -DCW_STAMPS): wall-clock (100 MHz) stamps of every wave -- kernel entry, behind the prologue's barrier, behind its first
// pair, at its end -- read back by tools/probe/wave_stamps.py through dq_dbg_read_wave
#ifdef CW_STAMPS
static __device__ unsigned long long cw_dbg[8 * 4096];
extern "C" void dq_dbg_read_wave(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(cw_dbg), sizeof(cw_dbg)); }
#define CW_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x < 256) cw_dbg[(i) * 4096 + blockIdx.x * 16 + (threadIdx.x >> 6)] = (i) == 7 ? __builtin_readcyclecounter() - cw_c0 : __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CW_STAMP(i) do { } while (0)
#endif

// 16-byte slot (inside a sample's piece plane of 64 slots) of chunk c of pixel p: the quarter image (entry 2 p + c) and a2 (entry 4 p + c)
static const unsigned char CW_SLOT1[64] = {53, 20, 30, 3, 27, 1, 22, 62, 56, 58, 0, 40, 2, 17, 44, 47, 31, 61, 57, 18, 9, 25, 23, 7, 21, 19, 59, 4, 36, 5, 50, 42, 26, 24, 16, 6, 12, 45, 41, 32, 37, 52, 46, 38, 11, 14, 49, 35, 10, 8, 48, 34,
                                           51, 28, 39, 55, 13, 33, 15, 43, 60, 29, 54, 63};      // (entries 50 ..: the padding columns 25 .. 31 of the first convolution's second tile -- slots of their own:
                                                                                                  // clamped to pixel 24 they were eight stores to ONE address per instruction, serialised)
static const unsigned char CW_SLOT2[64] = {53, 7, 57, 61, 58, 34, 37, 23, 22, 47, 43, 8, 28, 17, 3, 50, 12, 49, 16, 35, 11, 24, 63, 6, 14, 48, 52, 26, 20, 46, 45, 25, 56, 15, 44, 4, 54, 36, 1, 33, 9, 29, 27, 62, 51, 41, 0, 32, 30, 21, 13, 40, 42, 19, 2, 59, 5, 18, 60, 39, 38, 31, 55, 10};

__global__ __launch_bounds__(CW_THREADS) void conv_wave_kernel(ConvWaveArgs a) {
    constexpr int D = 5, OW2 = D - 1, OW3 = D - 2, R1 = D * D, R2 = OW2 * OW2, R3 = OW3 * OW3, T1 = 2;
    static_assert(R2 == 16 && R3 <= 16 && R1 <= 32, "one row tile for the second and third convolution, two for the first");
    constexpr int PL = 1024, SM = 2048;                             // bytes: piece l behind piece h, sample 1 behind sample 0
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kb = lane >> 4;
    static_assert(FWD_MAX_JOBS == 4, "three comparisons");
#ifdef CW_STAMPS
    const unsigned long long cw_c0 = __builtin_readcyclecounter();
#endif
    CW_STAMP(0);

    // ---- per-lane constants: one row of the host-built table (conv_wave_lane_table), requested first of all ------------------------------------------------
    const int4* lt = reinterpret_cast<const int4*>(a.ptab + PT_WAVE + lane * PT_WAVE_LD);
    const int4 lt0 = lt[0], lt1 = lt[1], lt2 = lt[2], lt3 = lt[3], lt4 = lt[4];
    // ---- this workgroup's weight set and its pairs of that set's list (block-uniform scalars; the set's record in one load behind the three comparisons) ----
    // (read through the kernel-argument segment's own pointer: indexing the by-value argument's arrays with a run-time value makes hipcc copy them --
    // and the job records -- to scratch memory)
    typedef const __attribute__((address_space(4))) ConvWaveArgs* KArgs;
    const KArgs ka = (KArgs)__builtin_amdgcn_kernarg_segment_ptr();
    const int w = (int)blockIdx.x;
    const int cls = (w >= a.set_wg0[1]) + (w >= a.set_wg0[2]) + (w >= a.set_wg0[3]);
    // ---- prologue: weights into LDS, as early as the set is known (the rest of the set's record is a second round trip) --------------------------------
    {
        const u32x4* pk = cls == 0 ? a.set_packed[0] : cls == 1 ? a.set_packed[1] : cls == 2 ? a.set_packed[2] : a.set_packed[3];
        for (int i = wave; i < CW_W_END / 64; i += CW_WAVES) {      // 1 KB units: [c1w 8 | c2w 32 | conv3 16]
            const u32x4* src = (i < 8 ? pk + a.pk_c1w + 64 * i : i < 40 ? pk + a.pk_c2w + 64 * (i - 8) : pk + PK_CONV3_FWD + 64 * (i - 40)) + lane;
            lds_dma16(src, (u32)(1024 * i));
        }
    }
    const auto& SET = ka->set[cls];
    const int rk = w - SET.wg0, R = SET.wgs;
    const int e0 = SET.end[0], e1 = SET.end[1], e2 = SET.end[2], T = SET.end[3];
    const int j0 = SET.job[0], j1 = SET.job[1], j2 = SET.job[2], j3 = SET.job[3];
    const int n0 = SET.batch[0], n1 = SET.batch[1], n2 = SET.batch[2], n3 = SET.batch[3];
    const int P = T > rk ? (T - rk + R - 1) / R : 0;                // this workgroup's pairs: rk, rk + R, ...
    if (P == 0) { __builtin_amdgcn_s_waitcnt(0x0F70); return; }     // block-uniform (the copies into this workgroup's LDS must not outlive it)
    CW_STAMP(4);
    const float* const bias_p = SET.params + 2 * j;                 // (the set's jobs share their parameters as they share the packed buffer)
    const u32x4* s_w = reinterpret_cast<const u32x4*>(smem);
    u32x4* s_lut = reinterpret_cast<u32x4*>(smem + CW_W_END * 16);
    if (tid < 256) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (((u32)tid >> (2 * q)) & 1u) * 0x3c00u | (((u32)tid >> (2 * q + 1)) & 1u) * 0x3c000000u;
        s_lut[tid] = v;
    }
    u8* const s_img = smem + CW_W_END * 16 + CW_LUT_BYTES + wave * CW_WAVE_BYTES;      // this wave's images
    // (conv_wave_lane_table explains the constants)
    const u32 ob[T1] = {(u32)lt0.x, (u32)lt0.y}, orc[T1] = {(u32)lt0.z, (u32)lt0.w};
    const int wa1[T1] = {lt1.x, lt1.y}, ra2[2] = {lt1.z, lt1.w}, wa2[4] = {lt2.x, lt2.y, lt2.z, lt2.w}, ra3[4] = {lt3.x, lt3.y, lt3.z, lt3.w};
    const int cp1 = lt4.x, cp2 = lt4.y;
    const u32 go3 = (u32)((4 * kb * 32 + 2 * j) * 4);               // conv3's output: row 4kb (+ r: 128 bytes each), channels 2j, 2j + 1
    CW_STAMP(5);
    __builtin_amdgcn_s_waitcnt(0x0F70);                             // vmcnt(0): this wave's weight copies have landed
    CW_STAMP(6);
    __syncthreads();
    CW_STAMP(1);

    // ---- pairs of this wave: p = wave, wave + 16, ... (all of this is wave-uniform scalar arithmetic; nothing below branches on data) ---------------------
    struct Pair { int k, b0, two; };
    // (a macro on plain locals, not a lambda: hipcc turns a select chain over by-reference captures into a run-time index into the closure object, which
    // then lives in scratch memory)
#define CW_PAIR_OF(r, p_)                                                                                            \
    do {                                                                                                             \
        const int g_ = rk + R * min((p_), P - 1);                                                                    \
        const int sl_ = (g_ >= e0) + (g_ >= e1) + (g_ >= e2);                                                        \
        (r).k = sl_ == 0 ? j0 : sl_ == 1 ? j1 : sl_ == 2 ? j2 : j3;                                                  \
        (r).b0 = 2 * (g_ - (sl_ == 0 ? 0 : sl_ == 1 ? e0 : sl_ == 2 ? e1 : e2));                                     \
        (r).two = (r).b0 + 1 < (sl_ == 0 ? n0 : sl_ == 1 ? n1 : sl_ == 2 ? n2 : n3) ? 1 : 0;                         \
    } while (0)
    // replay rows of a pair's samples (the second repeats the first when the job's count is odd): scalar loads through the index vector; a job without one
    // reads a valid dummy (no branch: hipcc duplicates whatever follows a branch into both arms)
    auto rows_of = [&](const Pair& pr, int& r0, int& r1) __attribute__((always_inline)) {
        const auto& J = ka->job[pr.k];
        const bool has = J.index != nullptr;
        const __attribute__((address_space(4))) int32_t* idx = (const __attribute__((address_space(4))) int32_t*)(uintptr_t)(has ? J.index : a.ptab);
        const int i0 = has ? pr.b0 : 0, i1 = has ? pr.b0 + pr.two : 0;
        int x0 = idx[i0] + J.index_off, x1 = idx[i1] + J.index_off;
        x0 -= x0 >= J.index_mod ? J.index_mod : 0; x1 -= x1 >= J.index_mod ? J.index_mod : 0;
        r0 = has ? x0 : pr.b0; r1 = has ? x1 : pr.b0 + pr.two;
    };
    auto load_bytes = [&](int k, int row, u32 (&by)[T1]) __attribute__((always_inline)) {
        const u8* src = ka->job[k].obs + (size_t)row * a.slot;
#pragma unroll
        for (int u = 0; u < T1; ++u) by[u] = src[ob[u]];
    };
    int p = wave;
    if (p >= P) return;                                             // wave-uniform; no barrier follows
    range_mask rmax = 0;                                            // the forward's range guard (qnet.h range_track): every layer's output where it is produced
    u32 by[2][T1], byn[2][T1];
    Pair cur, nxt;
    CW_PAIR_OF(cur, p); CW_PAIR_OF(nxt, p + CW_WAVES);
    int rn0, rn1;
    {
        int r0, r1;
        rows_of(cur, r0, r1);
        load_bytes(cur.k, r0, by[0]); load_bytes(cur.k, r1, by[1]);
        rows_of(nxt, rn0, rn1);
    }
    for (;;) {
        const auto& J = ka->job[cur.k];
        const int b0 = cur.b0;
        const bool two = cur.two != 0;
        load_bytes(nxt.k, rn0, byn[0]); load_bytes(nxt.k, rn1, byn[1]);      // the next pair's bytes and the rows behind it fly over this pair
        Pair nn;
        CW_PAIR_OF(nn, p + 2 * CW_WAVES);
        rows_of(nn, rn0, rn1);
        const bool train = J.write_all != 0, save_a1 = (J.write_all & 2) != 0;      // wave-uniform

#ifndef CW_BITS_LIVE
#define CW_BITS_LIVE 1                                              // 1: the four expanded bytes stay in 16 registers over the quarters; 0: re-read from the table per quarter
#endif
        u32 la[2][T1];                                              // the bytes' table entries
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < T1; ++u) la[s][u] = (by[s][u] << 4) | orc[u];
        u32x4 bitsr[2][T1];
        if (CW_BITS_LIVE) {
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < T1; ++u) bitsr[s][u] = *reinterpret_cast<const u32x4*>(smem + CW_W_END * 16 + la[s][u]);
        }
        f32x4 acc2[2][2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc2[s][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[s][t][1] = acc2[s][t][0]; }
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            // ---- conv1, output channels 16 q .. 16 q + 15 (transposed) -------------------------------------------------------------------------------
            const u32x4 w1h = s_w[CW_W_C1 + q * PK_BLOCK + lane], w1l = s_w[CW_W_C1 + q * PK_BLOCK + PK_LO + lane];
            // (the quarter image's stores: the second convolution's reads of the previous quarter are older LDS operations of this wave: in order)
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int u = 0; u < T1; ++u) {
                    const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                    const u32x4 bits = CW_BITS_LIVE ? bitsr[s][u] : *reinterpret_cast<const u32x4*>(smem + CW_W_END * 16 + la[s][u]);
                    const f32x4 ah = MFMA_F16(w1h, bits, z), al = MFMA_F16(w1l, bits, z);
                    const f32x4 vs = f16x2_sum(ah, al);
                    // (no run-time range check here: the operand is binary, so a1 is bounded by the c1w block's positive column sums, checked when it is packed)
                    uint2 hp, lp;
                    split_f16x2_pair(relu1(vs[0]), relu1(vs[1]), hp.x, lp.x);
                    split_f16x2_pair(relu1(vs[2]), relu1(vs[3]), hp.y, lp.y);
                    *reinterpret_cast<uint2*>(s_img + s * SM + wa1[u]) = hp;
                    *reinterpret_cast<uint2*>(s_img + s * SM + PL + wa1[u]) = lp;
                }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            if (save_a1) {                                          // a1 piece planes [sample pixel][64 halves]: this quarter's 32 bytes of every pixel (only for a
                                                                    // backward that reads them: conv_bwd16_kernel recomputes a1 from the patch words, round 6)
                const u32 cg1 = (u32)((lane >> 1) * 128 + 16 * (lane & 1));      // (recomputed: a register less over the pair)
#pragma unroll
                for (int s = 0; s < 2; ++s)
#pragma unroll
                    for (int piece = 0; piece < 2; ++piece) {
                        u8* dst = reinterpret_cast<u8*>(J.a1_pl + piece * J.a1_lo + (size_t)(b0 + s) * R1 * 64 + 16 * q);
                        if (lane < 2 * R1 && (s == 0 || two)) *reinterpret_cast<u32x4*>(dst + cg1) = *reinterpret_cast<const u32x4*>(s_img + s * SM + piece * PL + cp1);
                    }
            }
            // ---- conv2: the K blocks (quarter, ky) ---------------------------------------------------------------------------------------------------
#pragma unroll
            for (int ky = 0; ky < 2; ++ky) {
                F16x2 bw[2];
#pragma unroll
                for (int t = 0; t < 2; ++t) {
                    bw[t].h = s_w[CW_W_C2 + ((2 * q + ky) * 2 + t) * PK_BLOCK + lane];
                    bw[t].l = s_w[CW_W_C2 + ((2 * q + ky) * 2 + t) * PK_BLOCK + PK_LO + lane];
                }
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    F16x2 av;
                    av.h = *reinterpret_cast<const u32x4*>(s_img + s * SM + ra2[ky]);
                    av.l = *reinterpret_cast<const u32x4*>(s_img + s * SM + PL + ra2[ky]);
#pragma unroll
                    for (int t = 0; t < 2; ++t) mma_f16x3(av, bw[t], acc2[s][t][0], acc2[s][t][1]);
                }
            }
        }
        // ---- conv2 epilogue: a2 images (split on write) over the dead quarter image ------------------------------------------------------------------
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        const f32x2 bias2 = *reinterpret_cast<const f32x2*>(bias_p + a.b_off[1]);      // (per pair, through L1: four registers less over the pair)
        float rm2 = 0.f;                                            // this epilogue's maximum (qnet.h range_max)
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            f32x4 vs[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) vs[t] = f16x2_sum(acc2[s][t][0], acc2[s][t][1]) + f32x4{bias2[t], bias2[t], bias2[t], bias2[t]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                u32 h, l;
                const float r0 = relu1(vs[0][r]), r1 = relu1(vs[1][r]);
                range_max(rm2, r0, r1);
                split_f16x2_pair(r0, r1, h, l);
                *reinterpret_cast<u32*>(s_img + s * SM + wa2[r]) = h;
                *reinterpret_cast<u32*>(s_img + s * SM + PL + wa2[r]) = l;
            }
        }
        range_commit(rmax, rm2);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        if (train) {                                                // a2 piece planes [sample pixel][32 halves]
            const u32 cg2 = (u32)(16 * lane);
#pragma unroll
            for (int s = 0; s < 2; ++s)
#pragma unroll
                for (int piece = 0; piece < 2; ++piece) {
                    u8* dst = reinterpret_cast<u8*>(J.a2_pl + piece * J.a2_lo + (size_t)(b0 + s) * R2 * 32);
                    if (s == 0 || two) *reinterpret_cast<u32x4*>(dst + cg2) = *reinterpret_cast<const u32x4*>(s_img + s * SM + piece * PL + cp2);
                }
        }
        // ---- conv3 -> f32 rows [pixel][32] -----------------------------------------------------------------------------------------------------------------
        const f32x2 bias3 = *reinterpret_cast<const f32x2*>(bias_p + a.b_off[2]);
        f32x4 acc3[2][2][2];
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int t = 0; t < 2; ++t) { acc3[s][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc3[s][t][1] = acc3[s][t][0]; }
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            F16x2 bw[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) {
                bw[t].h = s_w[CW_W_C3 + (2 * tap + t) * PK_BLOCK + lane];
                bw[t].l = s_w[CW_W_C3 + (2 * tap + t) * PK_BLOCK + PK_LO + lane];
            }
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                F16x2 av;
                av.h = *reinterpret_cast<const u32x4*>(s_img + s * SM + ra3[tap]);
                av.l = *reinterpret_cast<const u32x4*>(s_img + s * SM + PL + ra3[tap]);
#pragma unroll
                for (int t = 0; t < 2; ++t) mma_f16x3(av, bw[t], acc3[s][t][0], acc3[s][t][1]);
            }
        }
        float rm3 = 0.f;
#pragma unroll
        for (int s = 0; s < 2; ++s) {
            if (s == 1 && !two) break;
            // split on write (round 6): the dense chain's operand, the dense weight gradients' and the data gradient's mask are these pieces -- the f32 copy had no
            // other reader, and splitting it again cost the dense chain's staging phase its vector ALU work: rows [pixel][32] of halves, 4 bytes per lane and piece
            u8* out = reinterpret_cast<u8*>(J.x_pl + (size_t)(b0 + s) * R3 * 32);
            u8* outf = reinterpret_cast<u8*>(J.act_out[2] + (size_t)(b0 + s) * R3 * 32);      // (training job: the f32 rows too -- the dense data gradient's ReLU mask)
            f32x4 vs[2];
#pragma unroll
            for (int t = 0; t < 2; ++t) vs[t] = f16x2_sum(acc3[s][t][0], acc3[s][t][1]) + f32x4{bias3[t], bias3[t], bias3[t], bias3[t]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float r0 = relu1(vs[0][r]), r1 = relu1(vs[1][r]);
                range_max(rm3, r0, r1);                             // (rows past R3 are clamped copies of real rows)
                u32 h, l;
                split_f16x2_pair(r0, r1, h, l);
                if (4 * kb + r < R3) {
                    if (J.x_pl) {                                   // (wave-uniform)
                        *reinterpret_cast<u32*>(out + (go3 >> 1) + 64 * r) = h;
                        *reinterpret_cast<u32*>(out + (go3 >> 1) + 64 * r + 2 * J.x_lo) = l;
                    }
                    if (train || !J.x_pl) *reinterpret_cast<f32x2*>(outf + go3 + 128 * r) = f32x2{r0, r1};
                }
            }
        }
        range_commit(rmax, rm3);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");     // (the next pair's quarter images overwrite a2: behind conv3's reads, in order)
        if (p == wave) CW_STAMP(2);
        p += CW_WAVES;
        if (p >= P) break;
        cur = nxt; nxt = nn;
#pragma unroll
        for (int s = 0; s < 2; ++s)
#pragma unroll
            for (int u = 0; u < T1; ++u) by[s][u] = byn[s][u];
    }
    range_report(rmax, a.range_flag);
    CW_STAMP(3);
    CW_STAMP(7);                                                    // (shader cycles of this wave's life)
}

// ---- host side ---------------------------------------------------------------------------------------------------------------------------------------
// Per-lane constants of the kernel's prologue (qnet.h PT_WAVE: row = lane, PT_WAVE_LD ints), lane = 16 kb + j:
//   [0..1]   ob[u]    byte offset inside an observation row of the byte this lane expands for tile u: byte kb of the word of pixel min(16u + j, R1 - 1)
//   [2..3]   orc[u]   << 4: that byte of the pixel's constant cells (PT_CONST << K_data) and the bias bit (K_data + 5)
//   [4..5]   wa1[u]   conv1 (transposed): channels 4kb .. 4kb + 3 of the quarter for column j of tile u: 8 bytes at half (kb & 1) of chunk kb >> 1
//   [6..7]   ra2[ky]  conv2: row j = output pixel (y, x); K block (quarter, ky): 8 channels 8 (kb & 1) .. of input pixel (y + ky, x + (kb >> 1))
//   [8..11]  wa2[r]   conv2's output rows of this lane: pixel 4kb + r, channels 2j, 2j + 1 of 32: 4 bytes inside chunk j >> 2
//   [12..15] ra3[tap] conv3: row j = output pixel (clamped past R3), tap (ky, kx), channels 8kb ..
//   [16..17] cp1, cp2 training copies: lane L -> 16 bytes of the quarter image (pixel L >> 1, chunk L & 1; L < 2 R1) / of a2 (chunk L in pixel order)
void conv_wave_lane_table(const dq_qnet* Q, int kd, const int* pt_const, int* out) {
    const int D = Q->L[0].oh, OW1 = D, OW2 = D - 1, OW3 = D - 2, R1 = D * D, R3 = OW3 * OW3;
    memset(out, 0, sizeof(int) * 64 * PT_WAVE_LD);
    if (D != 5 || kd + 6 > 32) return;                              // (conv_wave_supported)
    for (int lane = 0; lane < 64; ++lane) {
        int* o = out + lane * PT_WAVE_LD;
        const int j = lane & 15, kb = lane >> 4;
        for (int u = 0; u < 2; ++u) {
            const int p = 16 * u + j < R1 ? 16 * u + j : R1 - 1;
            o[u] = 4 * p + kb;
            const unsigned cb = ((unsigned)pt_const[p] << kd) | (1u << (kd + 5));
            o[2 + u] = (int)(((cb >> (8 * kb)) & 0xffu) << 4);
            o[4 + u] = 16 * CW_SLOT1[2 * (16 * u + j) + (kb >> 1)] + 8 * (kb & 1);
        }
        {
            const int y = j / OW2, x = j - y * OW2;
            for (int ky = 0; ky < 2; ++ky) o[6 + ky] = 16 * CW_SLOT1[2 * ((y + ky) * OW1 + x + (kb >> 1)) + (kb & 1)];
        }
        for (int r = 0; r < 4; ++r) o[8 + r] = 16 * CW_SLOT2[4 * (4 * kb + r) + (j >> 2)] + 4 * (j & 3);
        {
            const int n = j < R3 ? j : 0, y = n / OW3, x = n - y * OW3;
            for (int tap = 0; tap < 4; ++tap) o[12 + tap] = 16 * CW_SLOT2[4 * ((y + (tap >> 1)) * OW2 + x + (tap & 1)) + kb];
        }
        o[16] = 16 * CW_SLOT1[lane < 2 * R1 ? lane : 2 * R1 - 1];
        o[17] = 16 * CW_SLOT2[lane];
    }
}

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
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_wave_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CW_LDS));
        attr_devs |= dev_bit;
    }
    int cls_of[FWD_MAX_JOBS], n_cls = 0;
    for (int i = 0; i < a.n_jobs; ++i) {
        cls_of[i] = -1;
        for (int k = 0; k < i; ++k)
            if (a.job[k].packed == a.job[i].packed && a.job[k].params == a.job[i].params) { cls_of[i] = cls_of[k]; break; }
        if (cls_of[i] < 0) cls_of[i] = n_cls++;
    }
    int cls_pairs[FWD_MAX_JOBS] = {0, 0, 0, 0}, pairs = 0;
    for (int i = 0; i < a.n_jobs; ++i) { cls_pairs[cls_of[i]] += (a.job[i].batch + 1) / 2; pairs += (a.job[i].batch + 1) / 2; }
    int grid = pairs < n_cu ? pairs : n_cu;                         // (a wave takes two samples at a time)
    if (grid < n_cls) grid = n_cls;
    // the grid over the weight sets in proportion to their pairs, at least one workgroup each
    int wgs[FWD_MAX_JOBS], given = 0;
    for (int c = 0; c < n_cls; ++c) { wgs[c] = (int)((long long)grid * cls_pairs[c] / pairs); if (wgs[c] < 1) wgs[c] = 1; given += wgs[c]; }
    while (given < grid) {
        int best = 0; double need = -1.0;
        for (int c = 0; c < n_cls; ++c) { const double per = (double)cls_pairs[c] / wgs[c]; if (per > need) { need = per; best = c; } }
        ++wgs[best]; ++given;
    }
    while (given > grid) {
        int best = -1; double need = 1e300;
        for (int c = 0; c < n_cls; ++c) { if (wgs[c] < 2) continue; const double per = (double)cls_pairs[c] / (wgs[c] - 1); if (per < need) { need = per; best = c; } }
        if (best < 0) break;
        --wgs[best]; --given;
    }
    int acc = 0;
    for (int c = 0; c < FWD_MAX_JOBS; ++c) {
        ConvWaveArgs::Set& S = a.set[c];
        if (c >= n_cls) { memset(&S, 0, sizeof(S)); S.wg0 = 0x7fffffff; S.wgs = 1; a.set_wg0[c] = S.wg0; a.set_packed[c] = nullptr; continue; }
        S.wg0 = acc; S.wgs = wgs[c]; a.set_wg0[c] = acc; acc += wgs[c];
        int q = 0, end = 0, last = 0;
        for (int i = 0; i < a.n_jobs; ++i)
            if (cls_of[i] == c) {
                if (q == 0) { S.packed = a.job[i].packed; S.params = a.job[i].params; a.set_packed[c] = S.packed; }
                end += (a.job[i].batch + 1) / 2; S.job[q] = i; S.end[q] = end; S.batch[q] = a.job[i].batch; last = i; ++q;
            }
        for (; q < FWD_MAX_JOBS; ++q) { S.job[q] = last; S.end[q] = end; S.batch[q] = a.job[last].batch; }
    }
    dq_launch(DQ_K_CONV_CHAIN, "conv_wave_kernel", conv_wave_kernel, dim3(acc), dim3(CW_THREADS), CW_LDS, st, a);
    DQ_LAUNCH_CHECK();
    (void)Q;
    return DQ_OK;
}
