// Fused forward of the convolutional Q-network: two launches per forward (or per group of up to FWD_MAX_JOBS forwards), activations
// never leave the CU inside a chain.
//
// Same arithmetic as the per-layer path in qnet.hip (Keras model of build_convolutional_nn,
// /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:61-90, + keras-rl dueling head), laid out for MI355X.
// The network is tiny per sample (0.89 MFLOP and ~12 KB of activations at d=5), so a per-layer GEMM launch is dominated by prologue,
// epilogue and HBM round trips, and the f32-input MFMA (64 FLOP/clk/SIMD: one v_mfma_f32_16x16x4_f32 occupies its SIMD for 32 cycles)
// would be the bound.  The chains therefore run on the f16 matrix pipe at f32-class accuracy ("f16x2", qnet.h: every operand as two
// f16 pieces, three v_mfma_f32_16x16x32_f16 per product, f32 accumulation; results within 1e-5 of the float64 oracle like the per-layer
// f32 path) and are organised around what then bounds them -- VALU / LDS issue on the SIMD that also issues the MFMAs:
//
//   conv_chain_kernel   workgroup = 4 waves = S samples (2 workgroups per CU).  uint8 observation rows (optionally gathered from the
//                       replay ring) arrive by LDS-DMA; every convolution's output lives in LDS as READY-MADE f16 piece planes, written
//                       once by the producing layer's epilogue ("split on write": each value is read by four taps).  conv1: binary
//                       operand gathered byte-wise, the kernel's pieces pre-packed (two MFMAs per product are exact); conv2 / conv3:
//                       A = one ds_read_b128 per piece, B = packed weight pieces streamed through a register ring.
//   dense_chain_kernel  workgroup = 8 waves = 16 RT samples.  Both dense layers run TRANSPOSED (weights as the MFMA's first operand,
//                       samples as columns): the hidden layer's accumulators ARE Dense(|A|)'s second operand, so bias, ReLU, dropout
//                       and the split happen in registers and the hidden layer never goes through LDS; Dense(|A|) is split over K
//                       across the 8 waves and reduced in fixed order through LDS; the dueling layer and combination follow.
//
// The order of the reduction inside a dot product differs from the per-layer kernels: the two paths agree to f32 round-off, not bit for
// bit; each is deterministic.  Training forwards additionally write every layer's output to HBM (f32, the layout qnet.hip's backward
// expects) and the dense layers' inputs / outputs as f16 piece planes for the dense weight gradients (fused_bwd.hip).
#include "qnet.h"
#include <type_traits>
#include <utility>

DQ_STAMP_READER(dq_dbg_read_fwd)

// Build-time switches of the one-box A/B comparisons (tools/build_ab.sh; DESIGN.md section 4 records what each measured):
//   CONV_PIN / DENSE_PIN   sched_barrier pins that keep the weight-ring requests a whole block of MFMAs ahead of their use (+1 %: on)
//   CONV_APIPE             conv2 / conv3: the A pieces of K block b + 1 are read before block b's MFMAs (+0.4 us: on)
//   DENSE_XPIPE            the same for the dense forward's input pieces (neutral: on)
//   DQ_EXP_NODROP / DQ_EXP_NOSTORE   TIMING EXPERIMENTS ONLY -- the training forward without its dropout arithmetic / without its saved
//                          activations: WRONG RESULTS, used to price those parts of the training workgroups (never set in a product build)
#ifndef CONV_PIN
#define CONV_PIN 1
#endif
#ifndef DENSE_PIN
#define DENSE_PIN 1
#endif
#ifndef CONV_APIPE
#define CONV_APIPE 1
#endif
#ifndef DENSE_XPIPE
#define DENSE_XPIPE 1
#endif
#ifndef CONV_SWZ
#define CONV_SWZ 0                 // 1: a1 planes unpadded with their 16-byte chunks XOR-swizzled by pixel column (conv2's A reads conflict-free at d = 5): LDS bank
                                   // conflicts -59 %, LDS-active cycles -24 %, VALU +12 %, kernel +1.7 us -- off; 0: rows padded by 8 halves
#endif
#define A1_PS (CONV_SWZ ? 64 : 72)  // halves per a1 pixel row in LDS
#ifndef CONV1_PIPE
#define CONV1_PIPE 0               // persistent conv forward, EXPERIMENT (measured neutral): 1 = the first convolution software-pipelined across tiles (tile
                                   // t + 4's MFMAs in one basic block with tile t's epilogue; a1 planes hold whole tiles, rows past the end are stored too), 2 = the same
                                   // with sched_group_barrier interleaving (one MFMA, six VALU, one LDS read, ...)
#endif
#ifndef DQ_EXP_NODROP
#define DQ_EXP_NODROP 0
#endif
#ifndef DQ_EXP_NOSTORE
#define DQ_EXP_NOSTORE 0
#endif
#if DQ_EXP_NODROP || DQ_EXP_NOSTORE
#warning "timing-experiment build: the training forward's results are wrong"
#endif
#define CONV_THREADS 256
#define CONV_WAVES 4
#define CONV_LDS_2PER_CU (80 * 1024)
#define CONV_LDS_MAX CHAIN_LDS_MAX

struct ConvChainArgs {
    ConvJob job[FWD_MAX_JOBS];
    int n_jobs, S;
    int wg_first[FWD_MAX_JOBS];        // first workgroup of job 1, 2, 3 (INT_MAX for jobs that do not exist; [0] unused): the job of a workgroup by
                                       // three comparisons on one scalar load instead of a loop of dependent ones
    int C, H, W, k1, st1, K1;          // first convolution: input planes, kernel, stride, K = k1*k1*C
    int oh1, ow1, oh2, ow2, oh3, ow3;
    int w_off[3], b_off[3];            // floats into params
    const int* kofftab;                // [96] first convolution: weight row k -> byte offset inside an observation, -1 past K1
    const int* rowtab;                 // [3][CONV_ROWTAB] output row m of a workgroup -> where its input patch starts (fused_conv_row_tables): the
                                       // kernel never divides (every m -> (sample, y, x) was ~30 VALU, ten of them quarter-rate multiplies)
    const int* rowtab0;                // the first convolution's table: rowtab, or with patch-word input qnet.h PT_FWD
    int slot;                          // bytes per sample slot in LDS (multiple of 16, >= C*H*W + 30)
    int off_mis, off_t1, off_a1, off_a2;   // LDS byte offsets (observations at 0; a2 overlays observations + tables)
    int off_fx;                        // CONV_SWZ: byte per first-convolution output row = its chunk swizzle in halves (8 * f, f = 2 (ox & 3))
    int total_groups, off_obs1, off_a2b;   // persistent kernel: groups of S samples over all jobs; second observation buffer; a2 when the group's observations are in it
    // patch-word input (the kernels' KG1 == 0 instances; qnet.h PT_*, c1c, b1p): observation rows are `slot` bytes of u32 words, 16-byte aligned
    int off_lut;                       // LDS: byte -> its eight bits as f16 0 / 1 (256 x 16 bytes), built by the workgroup
    int pk_c1c, pk_b1p;                // u32x4 offsets of the compact first kernel and of the per-pixel bias table inside a job's packed buffer
    unsigned* range_flag;              // the forward's range guard (qnet.h range_report), nullable
};

// Patch-word input: the byte -> eight f16 (0 / 1) table every workgroup builds once (entry b, half e = bit e of b)
__device__ __forceinline__ void conv_build_bit_lut(u32x4* s_lut, int tid, int threads) {
    for (int b = tid; b < 256; b += threads) {
        u32x4 v;
#pragma unroll
        for (int q = 0; q < 4; ++q) v[q] = (((u32)b >> (2 * q)) & 1u) * 0x3c00u | (((u32)b >> (2 * q + 1)) & 1u) * 0x3c000000u;
        s_lut[b] = v;
    }
}

// One stride-1 convolution on the f16 matrix pipe at f32-class accuracy (f16x2, qnet.h).  Its input is an LDS image of READY-MADE
// pieces: two f16 planes [pixel][CIN + 8] (h plane, then the l plane `lo_in` halves further), written once by the producing layer's
// epilogue ("split on write": every value is read by KS*KS taps, so splitting at the read would repeat the arithmetic four times).
// K = KS*KS*CIN is walked in blocks of 32 channels of one tap: A = 8 consecutive channels per lane = one ds_read_b128 per piece (row
// stride CIN + 8 halves: the quarter-wave's 16 rows fall into distinct banks); B = the block's weights as pieces, streamed through a
// register ring.  Each wave takes pairs of 16-row tiles x both column tiles (column tile t, lane j = column 2j + t), two accumulators
// per tile (leading / scaled cross terms).  The output goes to LDS as pieces again (out_lds) and / or to global memory as f32.
template <int CIN, int COUT, int KS>
struct ConvShape {
    static constexpr int NT = COUT / 16, CB = CIN / 32, NB = KS * KS * CB, R = NB < 4 ? NB : 4;     // R: K = 32 weight blocks in flight
};

// One K = 32 block of packed weights (qnet.h) for this lane: NT column tiles x 2 f16 pieces, one coalesced 16-byte load each.
template <int NT>
__device__ __forceinline__ void conv_w_load(F16x2 (&slot)[NT], const u32x4* __restrict__ pk, int blk, int lane) {
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const u32x4* pb = pk + (blk * NT + t) * PK_BLOCK + lane;
        slot[t].h = pb[0]; slot[t].l = pb[PK_LO];
    }
}

// The first R blocks, issued by the caller BEFORE the barrier that publishes the input image.
template <int R, int NT>
__device__ __forceinline__ void conv_w_prefetch(F16x2 (&ring)[R][NT], const u32x4* __restrict__ pk, int lane) {
    pk = opaque_global(pk);
#pragma unroll
    for (int b = 0; b < R; ++b) conv_w_load<NT>(ring[b], pk, b, lane);
}

// SWZ (the second convolution with CONV_SWZ): the input planes' rows are CIN halves, unpadded, and pixel (y, x)'s 16-byte chunk c sits at chunk
// c ^ f(x), f(x) = 2 (x & 3).  With the lane groups ds_read_b128 is served in (MI355X_MICROARCH.md LDS table: rows 0-3, 12-15 of chunk kb together
// with rows 4-11 of chunk kb + 1) the padded rows put two or three rows on the same banks in every group at d = 5 (12 LDS cycles per read against
// 4; tools/probe/lds_swizzle.py enumerates it); the swizzle makes every group's 16 chunks hit 16 different bank quadruples for all four taps.
// The row table then carries ox & 3 in bits 24+ and the lane keeps one address per (row tile, kx, channel half).
template <int CIN, int COUT, int KS, int RR, int NTT, bool SWZ = false, int PADI = 8>
__device__ __forceinline__ void conv_from_lds(const unsigned short* __restrict__ in, int lo_in, int ih, int iw, int oh, int ow, int M,
                                              F16x2 (&ring)[RR][NTT], const u32x4* __restrict__ pk, const float* __restrict__ bias,
                                              unsigned short* __restrict__ out_lds, int lo_out, float* __restrict__ out_g, int wave, int lane,
                                              const int* __restrict__ rowtab, int pre0, int pre1, range_mask& rbad) {
    using SH = ConvShape<CIN, COUT, KS>;
    constexpr int NT = SH::NT, PSI = SWZ ? CIN : CIN + PADI, PSO = COUT + 8, CB = SH::CB, NB = SH::NB, R = SH::R;
    static_assert(NT == 2 && CIN % 32 == 0 && RR == R && NTT == NT, "written for 32 output channels, CIN a multiple of 32");
    static_assert(!SWZ || (CIN == 64 && KS == 2), "the swizzle is laid out for 64 input channels (8 chunks per row) and 2 x 2 taps");
    const int j = lane & 15, kb = lane >> 4;
    const f32x2 bias2 = *reinterpret_cast<const f32x2*>(bias + 2 * j);
    const int tiles = (M + 15) >> 4;
    (void)ih; (void)oh; (void)ow;
    for (int t0 = 2 * wave; t0 < tiles; t0 += 2 * CONV_WAVES) {
        pk = opaque_global(pk);                                     // per trip: see qnet.h
        const bool two = t0 + 1 < tiles;                            // wave-uniform; a missing second tile recomputes clamped rows
        const bool more = t0 + 2 * CONV_WAVES < tiles;              // another pair of row tiles follows: keep the weight stream going
        // patch origins of this lane's two rows: halves into the input image, from the host-built table (rows past M re-read row M - 1:
        // padding rows recompute the last row and are never stored); the first trip's entries were requested by the caller before its barrier
        int ent[2] = {pre0, pre1};
        if (t0 != 2 * wave) {                                       // wave-uniform
#pragma unroll
            for (int u = 0; u < 2; ++u) ent[u] = rowtab[min((t0 + u) * 16 + j, M - 1)];
        }
        int abase[2] = {ent[0] + 8 * kb, ent[1] + 8 * kb};
        int sa[2][2][2];                                            // SWZ: [row tile][kx][channel half]: halves to this lane's chunk of tap (0, kx)
        if (SWZ) {
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int kx = 0; kx < 2; ++kx) {
                    const int f8 = (((ent[u] >> 24) + kx) & 3) << 4;                  // 8 f(x + kx) halves: 0, 16, 32, 48
                    const int low = (ent[u] & 0xffffff) + kx * PSI + ((8 * kb) ^ (f8 & 16));
                    sa[u][kx][0] = low + (f8 & 32);
                    sa[u][kx][1] = low + (32 ^ (f8 & 32));
                }
        }
        f32x4 acc[2][NT][2];
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][t][1] = acc[u][t][0]; }
        // the weights stream through a ring of R blocks: block blk + R is requested as soon as block blk's MFMAs are issued, so the
        // L1/L2 weight traffic (every wave reads the whole layer) overlaps the matrix pipe instead of alternating with it
        // ... and the A pieces of block blk + 1 are read from LDS BEFORE block blk's MFMAs are issued (CONV_APIPE): read next to their use,
        // every block waited out two LDS latencies (one per row tile) with the matrix pipe idle
        auto a_read = [&](int blk, F16x2 (&av)[2]) {
            const int tap = blk / CB, c32 = blk - tap * CB, ky = tap / KS, kx = tap - ky * KS;
            const int off = (ky * iw + kx) * PSI + 32 * c32;
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned short* ap = SWZ ? in + sa[u][kx & 1][c32 & 1] + ky * iw * PSI : in + abase[u] + off;
                av[u].h = *reinterpret_cast<const u32x4*>(ap);
                av[u].l = *reinterpret_cast<const u32x4*>(ap + lo_in);
            }
        };
        F16x2 avr[2][2];
        if (CONV_APIPE) a_read(0, avr[0]);
#pragma unroll
        for (int blk = 0; blk < NB; ++blk) {
            if (CONV_APIPE) {
                if (blk + 1 < NB) { a_read(blk + 1, avr[(blk + 1) & 1]); __builtin_amdgcn_sched_barrier(0); }
            } else a_read(blk, avr[blk & 1]);
#pragma unroll
            for (int u = 0; u < 2; ++u)
#pragma unroll
                for (int t = 0; t < NT; ++t) mma_f16x3(avr[blk & 1][u], ring[blk % R][t], acc[u][t][0], acc[u][t][1]);
            if (blk + R < NB) conv_w_load<NT>(ring[blk % R], pk, blk + R, lane);
            else if (more) conv_w_load<NT>(ring[blk % R], pk, blk + R - NB, lane);
            if (CONV_PIN) __builtin_amdgcn_sched_barrier(0);        // the requests stay R blocks ahead of their use
        }
        // C/D layout of 16x16x32: col = lane & 15, row = (lane >> 4) * 4 + reg; this lane owns columns 2j, 2j+1
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            if (u == 1 && !two) break;
            // pieces' sum and bias as packed operations along an accumulator's own registers (v_pk_fma_f32 / v_pk_add_f32 take adjacent pairs)
            f32x4 vs[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) vs[t] = f16x2_sum(acc[u][t][0], acc[u][t][1]) + f32x4{bias2[t], bias2[t], bias2[t], bias2[t]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = (t0 + u) * 16 + 4 * kb + r;
                if (mo >= M) continue;
                const f32x2 v = {relu1(vs[0][r]), relu1(vs[1][r])};
                range_track(rbad, v[0], v[1]);                      // (the forward's range guard, qnet.h)
                if (out_lds) {
                    u32 h, l;
                    split_f16x2_pair(v[0], v[1], h, l);
                    *reinterpret_cast<u32*>(out_lds + mo * PSO + 2 * j) = h;
                    *reinterpret_cast<u32*>(out_lds + mo * PSO + 2 * j + lo_out) = l;
                }
                if (out_g) *reinterpret_cast<f32x2*>(out_g + (size_t)mo * COUT + 2 * j) = v;
            }
        }
    }
}

// ---- the first convolution over patch words (KG1 == 0 instances of both conv kernels) ---------------------------------------------------
// Output pixel p of Conv2D(64, 3, strides=2) on the padded planes has K_data <= 32 DATA cells in its 3 x 3 patch (four corners per syndrome
// plane, the centre per action plane: include/deepq_hip.h dq_env_patch_output); they arrive as ONE u32 per sample and pixel.  The A operand of
// the single K = 32 block is therefore one byte per lane (lane (kq, j): bits 8kq .. 8kq+7 of row j's word) expanded through a 256-entry LDS table --
// one ds_read_u8 + one ds_read_b128 where the uint8 image took 16 byte gathers, 8 packs and 8 multiplies -- and 8 MFMAs per tile instead of 16 / 24;
// the patch's constant cells are a per-pixel bias (qnet.h b1p).  A wave's tiles are the same rows of every group (tile = wave, wave + 4, ...: row m is
// pixel m mod r1 whatever the group), so the bias vectors of its first PATCH_BT tiles are REGISTERS (conv1_patch_bias: loaded with the kernel pieces, far
// ahead of their use; first version: four 16-byte loads per tile in front of its MFMAs -- an L2 round trip per tile that ate what the shorter gather
// saved: 41.3 against 42.3 us); tiles beyond (a group of more than 16 PATCH_BT rows: one workgroup per CU at d = 7) read them per tile.
//   s_in    the group's staged rows (row s at s * slot);  s_t1[m] & 0x1ffff = s * slot + 4 p for row m = s * r1 + p, padded to whole tiles
//   wb      [piece][column tile] the compact kernel's pieces (qnet.h c1c);  b1p  [r1][64] f32
// Output: a1 piece planes, rows of 64 halves (l plane lo1 halves further), rows < M1 only.
#define PATCH_BT 4
#ifndef PATCH_SPLIT
#define PATCH_SPLIT 0                   // 1: the odd last tile (13 tiles on 4 waves at 8 samples of d = 5) shared by waves 0 and 1, two column tiles each (below).
                                        // Measured (one box, twice each, tools/ab_run.sh): conv forward 43.5-43.7 us with it against 42.0 without -- the critical wave
                                        // does half a tile less, the kernel is 1.6 us SLOWER (four more live registers per lane, two more code paths in the unrolled
                                        // tile loop): off; same bits either way (tests/test_compact_gpu.py passes with both)
#endif
// ent(m): table entry of row m (any source); this lane's rows 4kq .. 4kq+3 of its wave's first PATCH_BT tiles, and (bpx) of the group's LAST tile
template <typename EntFn>
__device__ __forceinline__ void conv1_patch_bias(f32x4 (&bp)[PATCH_BT][4], f32x4 (&bpx)[4], const float* __restrict__ b1p, int slot, int wfirst, int last_tile,
                                                 int j, int kq, EntFn ent) {
#pragma unroll
    for (int u = 0; u < PATCH_BT; ++u)
#pragma unroll
        for (int r = 0; r < 4; ++r)
            bp[u][r] = *reinterpret_cast<const f32x4*>(b1p + ((ent((wfirst + CONV_WAVES * u) * 16 + 4 * kq + r) & (slot - 1)) << 4) + 4 * j);     // 4 p -> p * 64 floats
#pragma unroll
    for (int r = 0; r < 4; ++r) bpx[r] = *reinterpret_cast<const f32x4*>(b1p + ((ent(last_tile * 16 + 4 * kq + r) & (slot - 1)) << 4) + 4 * j);
}

// Tile balance: M1 rows are ceil(M1 / 16) tiles on CONV_WAVES waves, tile t to wave t mod 4 -- 13 tiles at 8 samples of d = 5 (and at 4 of d = 7): wave 0
// has four, the others three, and the barrier behind the phase waits for wave 0 (phase stamps: 6.6K cycles against 4.7K).  When the count is
// 4 n + 1 the LAST tile is shared: wave 0 computes its column tiles 0, 1 (this lane's channels 4j, 4j + 1), wave 1 its column tiles 2, 3 -- half the MFMAs
// and half the epilogue each: 3.5 tile times instead of 4 on the critical wave.
__device__ __forceinline__ void conv1_patch_words(const u8* __restrict__ s_in, const int* __restrict__ s_t1, const u32x4* __restrict__ s_lut,
                                                  const u32x4 (&wb)[2][1][4], const f32x4 (&bpr)[PATCH_BT][4], const f32x4 (&bpx)[4], const float* __restrict__ b1p,
                                                  int slot, unsigned short* __restrict__ s_a1, int lo1, int M1, int wfirst, int j, int kq, int bpx_tile, range_mask& rbad) {
    const int tiles = (M1 + 15) >> 4;
    if (wfirst >= tiles) return;                                    // wave-uniform
    const bool split = PATCH_SPLIT && (tiles & (CONV_WAVES - 1)) == 1 && tiles > CONV_WAVES;     // wave-uniform: the last tile belongs to wave 0 and is shared with wave 1
    auto byte_of = [&](int tile) -> u32 { return s_in[(s_t1[min(tile * 16 + j, M1 - 1)] & 0x1ffff) + kq]; };      // (tiles past the end reread the last row)
    auto tile_out = [&](int tile, u32 byte, const f32x4 (&bp)[4]) {
        const u32x4 av = s_lut[byte];
        f32x4 acc[4], accl[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[t] = acc[t]; }
#pragma unroll
        for (int t = 0; t < 4; ++t) acc[t] = MFMA_F16(av, wb[0][0][t], acc[t]);
#pragma unroll
        for (int t = 0; t < 4; ++t) accl[t] = MFMA_F16(av, wb[1][0][t], accl[t]);
        f32x4 vs[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) vs[t] = f16x2_sum(acc[t], accl[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mo = tile * 16 + 4 * kq + r;
            if (mo >= M1) continue;
            f32x4 v;
#pragma unroll
            for (int t = 0; t < 4; ++t) v[t] = relu1(vs[t][r] + bp[r][t]);
            range_track4(rbad, v);
            u32 hp[2], lp[2];                                       // split on write: this lane's 4 consecutive channels of pixel mo
            split_f16x2_pair(v[0], v[1], hp[0], lp[0]);
            split_f16x2_pair(v[2], v[3], hp[1], lp[1]);
            unsigned short* dst = s_a1 + mo * 64 + 4 * j;
            *reinterpret_cast<uint2*>(dst) = uint2{hp[0], hp[1]};
            *reinterpret_cast<uint2*>(dst + lo1) = uint2{lp[0], lp[1]};
        }
    };
    // half of the shared last tile: column tiles 2 HALF, 2 HALF + 1 = this lane's channels 4j + 2 HALF, + 1 (one dword per plane and row)
    auto tile_half = [&](int tile, u32 byte, const f32x4 (&bp)[4], auto half_c) {
        constexpr int HALF = decltype(half_c)::value;
        const u32x4 av = s_lut[byte];
        f32x4 acc[2], accl[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[t] = acc[t]; }
#pragma unroll
        for (int t = 0; t < 2; ++t) acc[t] = MFMA_F16(av, wb[0][0][2 * HALF + t], acc[t]);
#pragma unroll
        for (int t = 0; t < 2; ++t) accl[t] = MFMA_F16(av, wb[1][0][2 * HALF + t], accl[t]);
        f32x4 vs[2];
#pragma unroll
        for (int t = 0; t < 2; ++t) vs[t] = f16x2_sum(acc[t], accl[t]);
#pragma unroll
        for (int r = 0; r < 4; ++r) {
            const int mo = tile * 16 + 4 * kq + r;
            if (mo >= M1) continue;
            u32 hp, lp;
            const float r0 = relu1(vs[0][r] + bp[r][2 * HALF]), r1 = relu1(vs[1][r] + bp[r][2 * HALF + 1]);
            range_track(rbad, r0, r1);
            split_f16x2_pair(r0, r1, hp, lp);
            unsigned short* dst = s_a1 + mo * 64 + 4 * j + 2 * HALF;
            *reinterpret_cast<u32*>(dst) = hp;
            *reinterpret_cast<u32*>(dst + lo1) = lp;
        }
    };
    u32 by[PATCH_BT + 1];
    by[0] = byte_of(wfirst);
    const u32 byx = byte_of(tiles - 1);                             // (the shared tile's byte: waves 0 and 1 use it when `split`)
#pragma unroll
    for (int u = 0; u < PATCH_BT; ++u) {                            // (bytes one tile ahead)
        const int tile = wfirst + CONV_WAVES * u;
        if (tile >= tiles) break;                                   // wave-uniform
        by[u + 1] = byte_of(tile + CONV_WAVES);
        if (split && tile == tiles - 1) tile_half(tile, by[u], bpr[u], std::integral_constant<int, 0>());      // (wave 0's last)
        else tile_out(tile, by[u], bpr[u]);
    }
    if (split && wfirst == 1) {                                     // wave 1: the other half of wave 0's last tile
        if (tiles - 1 == bpx_tile) tile_half(tiles - 1, byx, bpx, std::integral_constant<int, 1>());
        else {                                                      // (a ragged group of the persistent kernel: bpx holds a full group's last tile -- this tile's bias through L1)
            const int4 e4 = *reinterpret_cast<const int4*>(s_t1 + (tiles - 1) * 16 + 4 * kq);
            const int ent[4] = {e4.x, e4.y, e4.z, e4.w};
            f32x4 bp[4];
#pragma unroll
            for (int r = 0; r < 4; ++r) bp[r] = *reinterpret_cast<const f32x4*>(b1p + ((ent[r] & (slot - 1)) << 4) + 4 * j);
            tile_half(tiles - 1, byx, bp, std::integral_constant<int, 1>());
        }
    }
    u32 bA = by[PATCH_BT];
    for (int tile = wfirst + CONV_WAVES * PATCH_BT; tile < tiles; tile += CONV_WAVES) {     // larger groups: the bias through L1, per tile
        const u32 bB = byte_of(tile + CONV_WAVES);
        const int4 e4 = *reinterpret_cast<const int4*>(s_t1 + tile * 16 + 4 * kq);     // this lane's four output rows (the table is padded to whole tiles)
        const int ent[4] = {e4.x, e4.y, e4.z, e4.w};
        f32x4 bp[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) bp[r] = *reinterpret_cast<const f32x4*>(b1p + ((ent[r] & (slot - 1)) << 4) + 4 * j);
        if (split && tile == tiles - 1) tile_half(tile, bA, bp, std::integral_constant<int, 0>());
        else tile_out(tile, bA, bp);
        bA = bB;
    }
}

template <int KG1>      // first convolution's K padded to 16 * KG1; 0: patch-word input (conv1_patch_words)
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_chain_kernel(ConvChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    u8* s_in = smem;
    int* s_mis = reinterpret_cast<int*>(smem + a.off_mis);
    unsigned short* s_a1 = reinterpret_cast<unsigned short*>(smem + a.off_a1);      // f16 piece planes [2][S*r1][A1_PS]
    u8* s_fx = smem + a.off_fx;                                                     // CONV_SWZ: chunk swizzle of every a1 row, in halves
    unsigned short* s_a2 = reinterpret_cast<unsigned short*>(smem + a.off_a2);      // [2][S*r2][40]; overlays the observations (dead after conv1)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
    range_mask rbad = 0;                                            // the forward's range guard (qnet.h range_track)
    static_assert(FWD_MAX_JOBS == 4, "three comparisons");
    const int jb = ((int)blockIdx.x >= a.wg_first[1]) + ((int)blockIdx.x >= a.wg_first[2]) + ((int)blockIdx.x >= a.wg_first[3]);      // block-uniform
    const ConvJob& J = a.job[jb];
    const int grp = (int)blockIdx.x - J.wg0;
    const int b0 = grp * a.S;
    const int ns = min(a.S, J.batch - b0);
    constexpr bool CP = KG1 == 0;                                   // patch-word input: one u32 per pixel instead of the padded uint8 image
    const int in_bytes = CP ? a.slot : a.C * a.H * a.W;
    constexpr int NH1 = CP ? 1 : (KG1 + 1) / 2;                     // first convolution's K in halves of 32
    constexpr int A1S = CP ? 64 : A1_PS;                            // halves per a1 row in LDS (patch-word input: unpadded, which makes room for the bit table)

    DQ_STAMP(DQ_TAG_CONV_FWD, 0);
    DQ_STAMP_WG(DQ_TAG_CONV_FWD, 0);
    DQ_STAMP_PAIR(0);
    // ---- first convolution's weights -> registers (the loads fly while the observations are staged) ----------------
    // The observation is binary (0/1 is exact in f16), so conv1 needs only the weight's two f16 pieces (qnet.h): TWO MFMAs per
    // product, each a*piece exact, accumulated in f32 by v_mfma_f32_16x16x32_f16 (leading and 2^11-scaled pieces in accumulators of
    // their own).  Layout of 16x16x32: lane (kb = lane >> 4, i = lane & 15) holds A[i][8kb .. 8kb+7] / B[8kb .. 8kb+7][i]; column
    // tile t, lane j is column 4j + t.
    // ready-made pieces from the packed buffer (PK_CONV1, zero past K1): splitting them here cost every workgroup ~350 VALU per wave
    u32x4 wb[2][NH1][4];                                            // [piece][k-half of 32][column tile]: 8 f16 each
    int ko[NH1][8];
    {
        const u32x4* pk1 = J.packed + (CP ? a.pk_c1c : PK_CONV1) + lane;
#pragma unroll
        for (int h = 0; h < NH1; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int piece = 0; piece < 2; ++piece) wb[piece][h][t] = pk1[(h * 4 + t) * PK_BLOCK + PK_LO * piece];
        if constexpr (!CP) {
#pragma unroll
            for (int h = 0; h < NH1; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) ko[h][e] = max(a.kofftab[32 * h + 8 * kq + e], 0);      // Keras HWIO row k -> NCHW uint8 offset (-1 past K1: weights 0)
        }
    }
    const f32x4 bias1 = *reinterpret_cast<const f32x4*>(J.params + a.b_off[0] + 4 * j);
    // row tables (requested here, used behind the barriers): this thread's first-convolution row, this lane's two rows of the wave's first
    // tile pair of the second and third convolution
    const int r1 = a.oh1 * a.ow1, M1 = ns * r1, M2 = ns * a.oh2 * a.ow2, M3 = ns * a.oh3 * a.ow3;
    const int tab1 = a.rowtab0[min(tid, M1 - 1)];
    int tab2[2], tab3[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        tab2[u] = a.rowtab[(CP ? 3 : 1) * CONV_ROWTAB + min((2 * wave + u) * 16 + j, M2 - 1)];      // (table 3: a1 rows of 64 halves)
        tab3[u] = a.rowtab[2 * CONV_ROWTAB + min((2 * wave + u) * 16 + j, M3 - 1)];
    }
    u32x4* s_lut = reinterpret_cast<u32x4*>(smem + a.off_lut);
    f32x4 bp1[PATCH_BT][4], bpx[4];                                 // CP: per-pixel bias of this lane's rows (conv1_patch_words); bpx: of the group's last tile
    if constexpr (CP) {
        conv_build_bit_lut(s_lut, tid, CONV_THREADS);
        conv1_patch_bias(bp1, bpx, reinterpret_cast<const float*>(J.packed + a.pk_b1p), a.slot, wave, (M1 + 15) / 16 - 1, j, kq,
                         [&](int m) { return a.rowtab0[min(m, CONV_ROWTAB - 1)]; });
    }

    DQ_STAMP(DQ_TAG_CONV_FWD, 1);
    // ---- stage the observations by LDS-DMA (global -> LDS, no registers): lane l copies aligned 16-byte word l of a 1 KB piece of a
    //      sample's arbitrarily aligned row -- whole aligned words, also where they straddle the neighbouring rows: the window never
    //      leaves the caller's allocation (dq_qnet_forward: obs_dev rows live in one 16-byte-aligned allocation whose allocated size
    //      is a multiple of 16). ---------------------------------------------------------------------------------------------------
    {
        // wave w takes samples w, w + 4, ...: their ring rows (scalar loads) are all requested BEFORE the first copy -- one after the
        // other, every sample's copy waited for its own index
        constexpr int SPW = 4;                                      // samples per wave at most (S <= 16)
        int rows[SPW];
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const int s = min(wave + CONV_WAVES * q, ns - 1);
            rows[q] = J.index ? J.index[b0 + s] : b0 + s;
        }
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const int s = wave + CONV_WAVES * q;
            if (s >= ns) break;                                     // wave-uniform
            int row = rows[q];
            if (J.index) { row += J.index_off; if (row >= J.index_mod) row -= J.index_mod; }
            const u8* src = J.obs + (size_t)row * in_bytes;
            const int mis = (int)(reinterpret_cast<uintptr_t>(src) & 15);  // (patch words: rows are 16-byte aligned, 0)
            // 16 bytes per lane: ONE instruction copies up to 1 KB (a whole d = 5 observation; four dword copies before)
            const u32x4* gp = reinterpret_cast<const u32x4*>(src - mis) + lane;
            u8* lp = s_in + s * a.slot;
            const int nq = (mis + in_bytes + 15) >> 4;                          // 16-byte pieces of the window
            for (int pc = 0; 64 * pc < nq; ++pc)
                if (64 * pc + lane < nq) __builtin_amdgcn_global_load_lds(gp + 64 * pc, (__attribute__((address_space(3))) u32*)(lp + 1024 * pc), 16, 0, 0);
            if (lane == 0) s_mis[s] = mis;
        }
    }
    __syncthreads();                                                // s_mis
    // byte offset of output pixel m's patch origin inside the staged observations
    const int lo1 = a.S * r1 * A1S, lo2 = a.S * a.oh2 * a.ow2 * 40;     // halves from an h plane to its l plane
    int* s_t1 = reinterpret_cast<int*>(smem + a.off_t1);
    for (int m = tid; m < (CP ? (M1 + 15) & ~15 : M1); m += CONV_THREADS) {     // table entry: sample << 20 | swizzle f << 17 | offset of the patch inside the observation
        const int e = m == tid && m < M1 ? tab1 : a.rowtab0[min(m, M1 - 1)], s = e >> 20;      // (patch words: padded to whole tiles with the last row)
        s_t1[m] = s * a.slot + s_mis[s] + (e & 0x1ffff);
        if (CONV_SWZ) s_fx[m] = (u8)(((e >> 17) & 7) << 3);         // 8 f halves
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                // this wave's DMA pieces have landed
    __syncthreads();

    DQ_STAMP(DQ_TAG_CONV_FWD, 2);
    // ---- convolution 1: A gathered byte-wise from the uint8 image; the bytes of this wave's next tile are requested before the
    //      MFMAs of the current one ----------------------------------------------------------------------------------------------
    if constexpr (CP) {
        conv1_patch_words(s_in, s_t1, s_lut, wb, bp1, bpx, reinterpret_cast<const float*>(J.packed + a.pk_b1p), a.slot, s_a1, lo1, M1, wave, j, kq, (M1 + 15) / 16 - 1, rbad);
    } else {
        const int tiles = (M1 + 15) >> 4;
        auto origin = [&](int tile) { return s_t1[min(tile * 16 + j, M1 - 1)]; };     // rows past the end (and whole tiles past it) reread the last row
        auto rd = [&](int org, u32 (&ab)[NH1][8]) {
            const u8* ap = s_in + org;
#pragma unroll
            for (int h = 0; h < NH1; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) ab[h][e] = ap[ko[h][e]];        // 0 or 1
        };
        auto tile_out = [&](int tile, const u32 (&ab)[NH1][8]) {
            u32 fxh[4];                                             // CONV_SWZ: swizzles of this lane's four output rows (bytes past M1 belong to rows that are not stored)
            if (CONV_SWZ) {
#pragma unroll
                for (int r = 0; r < 4; ++r) fxh[r] = s_fx[tile * 16 + 4 * kq + r];
            }
            f32x4 acc[4], accl[4];
#pragma unroll
            for (int t = 0; t < 4; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[t] = acc[t]; }
#pragma unroll
            for (int h = 0; h < NH1; ++h) {
                u32x4 av;
#pragma unroll
                for (int e = 0; e < 8; e += 2) av[e >> 1] = __umul24(ab[h][e] | (ab[h][e + 1] << 16), 0x3c00u);      // f16(1.0) = 0x3c00; v_mul_u32_u24 (a 32-bit multiply is quarter rate)
#pragma unroll
                for (int t = 0; t < 4; ++t) acc[t] = MFMA_F16(av, wb[0][h][t], acc[t]);
#pragma unroll
                for (int t = 0; t < 4; ++t) accl[t] = MFMA_F16(av, wb[1][h][t], accl[t]);
            }
            f32x4 vs[4];                                            // pieces' sum and bias, packed along each accumulator's own registers
#pragma unroll
            for (int t = 0; t < 4; ++t) vs[t] = f16x2_sum(acc[t], accl[t]) + f32x4{bias1[t], bias1[t], bias1[t], bias1[t]};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int mo = tile * 16 + 4 * kq + r;
                if (mo >= M1) continue;
                f32x4 v;
#pragma unroll
                for (int t = 0; t < 4; ++t) v[t] = relu1(vs[t][r]);
                range_track4(rbad, v);
                u32 hp[2], lp[2];                                   // split on write: this lane's 4 consecutive channels of pixel mo
                split_f16x2_pair(v[0], v[1], hp[0], lp[0]);
                split_f16x2_pair(v[2], v[3], hp[1], lp[1]);
                // this lane's 4 channels 4j .. 4j+3 = half (j & 1) of chunk j >> 1
                unsigned short* dst = CONV_SWZ ? s_a1 + mo * 64 + (((j >> 1) << 3) ^ (int)fxh[r]) + ((j & 1) << 2) : s_a1 + mo * 72 + 4 * j;
                *reinterpret_cast<uint2*>(dst) = uint2{hp[0], hp[1]};
                *reinterpret_cast<uint2*>(dst + lo1) = uint2{lp[0], lp[1]};
            }
        };
        u32 abA[NH1][8], abB[NH1][8];
        if (wave < tiles) {
            // two tiles ahead: the patch origin (one LDS read the byte gathers depend on); one tile ahead: the bytes
            int orgB = origin(wave + CONV_WAVES), orgA;
            rd(origin(wave), abA);
            for (int tile = wave;;) {
                rd(orgB, abB); orgA = origin(tile + 2 * CONV_WAVES); tile_out(tile, abA); tile += CONV_WAVES; if (tile >= tiles) break;
                rd(orgA, abA); orgB = origin(tile + 2 * CONV_WAVES); tile_out(tile, abB); tile += CONV_WAVES; if (tile >= tiles) break;
            }
        }
    }
    DQ_STAMP(DQ_TAG_CONV_FWD, 3);
    // ---- convolution 2 (64 -> 32, 2x2) and 3 (32 -> 32, 2x2): the first weight blocks are requested before the barrier ------------
    F16x2 ring[4][2];
    conv_w_prefetch(ring, J.packed + PK_CONV2_FWD, lane);
    __syncthreads();
    DQ_STAMP(DQ_TAG_CONV_FWD, 4);
    if constexpr (CP) {
        conv_from_lds<64, 32, 2, 4, 2, false, 0>(s_a1, lo1, a.oh1, a.ow1, a.oh2, a.ow2, M2, ring, J.packed + PK_CONV2_FWD, J.params + a.b_off[1], s_a2, lo2,
                                                 nullptr, wave, lane, a.rowtab + 3 * CONV_ROWTAB, tab2[0], tab2[1], rbad);
    } else {
        conv_from_lds<64, 32, 2, 4, 2, CONV_SWZ != 0>(s_a1, lo1, a.oh1, a.ow1, a.oh2, a.ow2, M2, ring, J.packed + PK_CONV2_FWD, J.params + a.b_off[1], s_a2, lo2,
                                                      nullptr, wave, lane, a.rowtab + CONV_ROWTAB, tab2[0], tab2[1], rbad);
    }
    DQ_STAMP(DQ_TAG_CONV_FWD, 5);
    conv_w_prefetch(ring, J.packed + PK_CONV3_FWD, lane);
    __syncthreads();
    DQ_STAMP(DQ_TAG_CONV_FWD, 6);
    {
        const int r3 = a.oh3 * a.ow3;
        conv_from_lds<32, 32, 2, 4, 2>(s_a2, lo2, a.oh2, a.ow2, a.oh3, a.ow3, M3, ring, J.packed + PK_CONV3_FWD, J.params + a.b_off[2], nullptr, 0,
                                 J.act_out[2] + (size_t)b0 * r3 * 32, wave, lane, a.rowtab + 2 * CONV_ROWTAB, tab3[0], tab3[1], rbad);
    }
    // ---- training: a1 and a2 leave as the piece planes they are in LDS (the convolutional backward's operands, fused_bwd.hip) in ONE burst of
    //      16-byte copies at the very end: stores from the layers' epilogues sat in front of the next layer's weight requests (loads and
    //      stores retire in order), +2.5 us on the training workgroups, which set the kernel's duration --------------------------------
    if (J.write_all) {                                              // block-uniform; a1 is dead but intact since conv2, a2 since conv3
        __syncthreads();
        const int r2 = a.oh2 * a.ow2;
        // a1: LDS rows of A1_PS halves (chunks swizzled with CONV_SWZ) -> global rows of 64 (8 slots of 16 B per row and plane); a2: rows of 40 -> 32 (4 slots)
        for (int i = tid; i < 2 * M1 * 8; i += CONV_THREADS) {
            const int piece = i >= M1 * 8 ? 1 : 0, q = i - piece * M1 * 8, row = q >> 3, part = q & 7;
            const int src = CP ? row * 64 + 8 * part : CONV_SWZ ? row * 64 + ((8 * part) ^ (int)s_fx[row]) : row * 72 + 8 * part;
            *reinterpret_cast<u32x4*>(J.a1_pl + piece * J.a1_lo + ((size_t)b0 * r1 + row) * 64 + 8 * part) =
                *reinterpret_cast<const u32x4*>(s_a1 + piece * lo1 + src);
        }
        for (int i = tid; i < 2 * M2 * 4; i += CONV_THREADS) {
            const int piece = i >= M2 * 4 ? 1 : 0, q = i - piece * M2 * 4, row = q >> 2, part = q & 3;
            *reinterpret_cast<u32x4*>(J.a2_pl + piece * J.a2_lo + ((size_t)b0 * r2 + row) * 32 + 8 * part) =
                *reinterpret_cast<const u32x4*>(s_a2 + piece * lo2 + row * 40 + 8 * part);
        }
    }
    range_report(rbad, a.range_flag);
    DQ_STAMP(DQ_TAG_CONV_FWD, 7);
    DQ_STAMP_WG(DQ_TAG_CONV_FWD, 1);
    DQ_STAMP_PAIR(1);
}

// ---- persistent form (round 3) --------------------------------------------------------------------------------------------------
// Phase stamps of conv_chain_kernel in the vector step's launch (2048 workgroups = 4 rounds of 2 per CU) showed every workgroup, in every
// round, spending 1.8-3.0K of its ~20K cycles in chains of dependent scalar loads (kernel arguments -> job record -> pointers -> row tables,
// first-layer weights) and another 1.7-2.6K waiting for its observations (replay-ring index -> LDS-DMA from HBM): a fifth of a workgroup's
// life before the first MFMA, paid again by each of the 2048.  Here 2 workgroups per CU stay resident and walk the groups of S samples
// (group g, g + grid, ...):
//   * row tables and the first layer's byte offsets are read once;
//   * the observations of group g + 1 are requested by LDS-DMA at the top of group g, a whole group ahead, into a second buffer (the a1 planes are
//     unpadded here to make room: 64 halves per pixel; conv2's A reads then take 16 LDS cycles instead of 12, and the LDS has that slack --
//     tools/probe/lds_swizzle.py);
//   * the first layer's weight pieces of group g + 1 are requested when group g's last convolution is under way, and an explicit
//     s_waitcnt vmcnt(0) BEFORE the next DMA is issued retires them: a wave's memory operations retire in order, so a register load waited for
//     behind a younger DMA would wait for the DMA as well (hipcc cannot count the DMA's instructions and falls back to vmcnt(0)).
// LDS: [obs A | core | obs B | a1 planes | alignment offsets]; a2 (conv2's output) overlays the CURRENT group's observation buffer and the core
// (the patch-origin table lives there), never the buffer the next group's observations are landing in.
template <int KG1>
__global__ __launch_bounds__(CONV_THREADS, 2) void conv_chain_pkernel(ConvChainArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    int* s_mis = reinterpret_cast<int*>(smem + a.off_mis);          // [2][16]
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
    range_mask rbad = 0;                                            // the forward's range guard (qnet.h range_track)
    static_assert(FWD_MAX_JOBS == 4, "three comparisons");
    constexpr bool CP = KG1 == 0;                                   // patch-word input (conv1_patch_words)
    const int in_bytes = CP ? a.slot : a.C * a.H * a.W;
    constexpr int NH1 = CP ? 1 : (KG1 + 1) / 2;                     // first convolution's K in halves of 32
    const int r1 = a.oh1 * a.ow1, r2 = a.oh2 * a.ow2, r3 = a.oh3 * a.ow3;
    const int lo1 = (CONV1_PIPE ? (a.S * r1 + 15) & ~15 : a.S * r1) * 64, lo2 = a.S * r2 * 40;      // halves from an h plane to its l plane (CONV1_PIPE: whole 16-row tiles)
    const int total = a.total_groups, gstride = (int)gridDim.x;
    auto job_of = [&](int g) { return (g >= a.wg_first[1]) + (g >= a.wg_first[2]) + (g >= a.wg_first[3]); };     // block-uniform

    DQ_STAMP(DQ_TAG_CONV_FWD, 0);
    // ---- once per workgroup: byte offsets of the first layer's K, row tables of a FULL group (rows past a ragged group's end read stale rows
    //      of LDS whose results are never stored: MFMA rows are independent) -----------------------------------------------------------
    int ko[NH1][8];                                                 // (requested with the first layer's weights, per group: 16 registers that would
                                                                    // otherwise stay live through the other two convolutions)
    const int MF1 = a.S * r1, MF2 = a.S * r2, MF3 = a.S * r3;
    const int tab1 = a.rowtab0[min(tid, MF1 - 1)];
    u32x4* s_lut = reinterpret_cast<u32x4*>(smem + a.off_lut);
    if constexpr (CP) conv_build_bit_lut(s_lut, tid, CONV_THREADS);     // (published by the first group's top barrier)
    int tab2[2], tab3[2];
#pragma unroll
    for (int u = 0; u < 2; ++u) {
        tab2[u] = a.rowtab[3 * CONV_ROWTAB + min((2 * wave + u) * 16 + j, MF2 - 1)];       // (table 3: a1 rows of 64 halves)
        tab3[u] = a.rowtab[2 * CONV_ROWTAB + min((2 * wave + u) * 16 + j, MF3 - 1)];
    }
    // first-layer weight pieces (PK_CONV1, zero past K1) and bias of a job
    u32x4 wb[2][NH1][4];                                            // [piece][k-half of 32][column tile]: 8 f16 each
    f32x4 bias1;
    f32x4 bp1[PATCH_BT][4], bpx[4];                                 // CP: per-pixel bias of this lane's rows (conv1_patch_words), reloaded with the pieces; bpx: of a
                                                                    // FULL group's last tile (a ragged last group: its own last tile's rows are a prefix of the same pixels
                                                                    // only when it ends on the same tile -- see the use)
    auto load_w1 = [&](const ConvJob& Jn) {
        const u32x4* pk1 = opaque_global(Jn.packed + (CP ? a.pk_c1c : PK_CONV1)) + lane;
#pragma unroll
        for (int h = 0; h < NH1; ++h)
#pragma unroll
            for (int t = 0; t < 4; ++t)
#pragma unroll
                for (int piece = 0; piece < 2; ++piece) wb[piece][h][t] = pk1[(h * 4 + t) * PK_BLOCK + PK_LO * piece];
        if constexpr (CP) {
            const int* tc = reinterpret_cast<const int*>(smem + a.off_t1);      // (the constant table: row m -> sample, 4 p; padded to whole tiles)
            conv1_patch_bias(bp1, bpx, reinterpret_cast<const float*>(opaque_global(Jn.packed + a.pk_b1p)), a.slot, wave, (MF1 + 15) / 16 - 1, j, kq,
                             [&](int m) { return tc[min(m, ((MF1 + 15) & ~15) - 1)]; });
        } else {
            bias1 = *reinterpret_cast<const f32x4*>(Jn.params + a.b_off[0] + 4 * j);
            const int* kt = reinterpret_cast<const int*>(opaque_global(reinterpret_cast<const u32x4*>(a.kofftab))) + 8 * kq;
#pragma unroll
            for (int h = 0; h < NH1; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) ko[h][e] = kt[32 * h + e];      // Keras HWIO row k -> NCHW uint8 offset (-1 past K1: weights 0; clamped at the use)
        }
    };
    // LDS-DMA of group g's observations into buffer buf (conv_chain_kernel's staging: whole aligned 16-byte words of arbitrarily aligned rows)
    // Staging in two steps.  prep(g): the scalar part -- job record, this wave's replay rows of group g -- requested TWO groups ahead (at the tail of
    // group g - 2) so that nothing waits for it: loaded at the top of the group that issues the copies, this chain of dependent scalar loads was
    // 1.4K of every group's ~18K cycles (phase stamps).  The index vector is read through the constant address space: it is not written while this
    // kernel runs, and inside the loop hipcc otherwise turns every index[uniform] into a vector load + s_waitcnt vmcnt(0) + v_readfirstlane.
    // issue(P, buf): the copies, LDS-DMA of whole aligned 16-byte words of the arbitrarily aligned rows (conv_chain_kernel's staging).
    constexpr int SPW = 4;                                          // samples per wave at most (S <= 16)
    struct Prep { const u8* obs; int rows[SPW]; int ns, has_index, index_off, index_mod; };
    auto prep = [&](int g) {
        Prep P;
        const ConvJob& Jn = a.job[job_of(g)];
        const int b0n = (g - Jn.wg0) * a.S;
        P.ns = min(a.S, Jn.batch - b0n); P.obs = Jn.obs; P.has_index = Jn.index != nullptr; P.index_off = Jn.index_off; P.index_mod = Jn.index_mod;
        const __attribute__((address_space(4))) int32_t* idx = (const __attribute__((address_space(4))) int32_t*)(uintptr_t)Jn.index;
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const int r = b0n + min(wave + CONV_WAVES * q, P.ns - 1);
            P.rows[q] = P.has_index ? idx[r] : r;
        }
        return P;
    };
    auto issue = [&](const Prep& P, int buf) {
        u8* dst = smem + (buf ? a.off_obs1 : 0);
#pragma unroll
        for (int q = 0; q < SPW; ++q) {
            const int s = wave + CONV_WAVES * q;
            if (s >= P.ns) break;                                   // wave-uniform
            int row = P.rows[q];
            if (P.has_index) { row += P.index_off; if (row >= P.index_mod) row -= P.index_mod; }
            const u8* src = P.obs + (size_t)row * in_bytes;
            const int mis = (int)(reinterpret_cast<uintptr_t>(src) & 15);
            const u32x4* gp = reinterpret_cast<const u32x4*>(src - mis) + lane;
            u8* lp = dst + s * a.slot;
            const int nq = (mis + in_bytes + 15) >> 4;              // 16-byte pieces of the window
            // (lds_dma16, qnet.h: the copies must stay in flight across this whole group; the wave's own s_waitcnt vmcnt(0) at the top of the next
            // group retires them)
            const u32 lds0 = lds_addr(lp);
            for (int pc = 0; 64 * pc < nq; ++pc)
                if (64 * pc + lane < nq) lds_dma16(gp + 64 * pc, lds0 + 1024 * pc);
            if (lane == 0) s_mis[buf * 16 + s] = mis;
        }
    };

    // the patch-origin table, ONCE per workgroup: entry m = sample << 20 | (sample * slot + offset of the patch inside the observation); what changes from
    // group to group is only each sample's alignment offset (s_mis), added at the use.  (Rebuilt per group -- it lived in the core that a2 overlays --
    // it cost every group a pass over the table and a barrier of its own: 0.7K of a group's 17K cycles.)
    {
        int* s_t1c = reinterpret_cast<int*>(smem + a.off_t1);
        for (int m = tid; m < (CP ? (MF1 + 15) & ~15 : MF1); m += CONV_THREADS) {      // (patch words: padded to whole tiles with the last row)
            const int e = m == tid && m < MF1 ? tab1 : a.rowtab0[min(m, MF1 - 1)], s = e >> 20;
            s_t1c[m] = (e & ~0x1ffff) | (s * a.slot + (e & 0x1ffff));
        }
    }
    int gid = (int)blockIdx.x, cur = 0;
    int gk = 0; (void)gk;                                           // group count of this workgroup (stamps: 8 per group)
    if constexpr (CP) __syncthreads();                              // (load_w1 reads the table just built)
    load_w1(a.job[job_of(gid)]);
    issue(prep(gid), 0);
    Prep pre = prep(min(gid + gstride, total - 1));                 // the group whose copies the first trip issues
    for (;;) {
        // LDS bases and table pointers re-derived per group from opaque copies: as loop invariants hipcc keeps every address computed from them
        // in registers of its own across the whole group (spills)
        int o_a1 = a.off_a1, o_t1 = a.off_t1, wv = wave;             // (wv: the row numbers of a wave's tiles, and everything computed from them)
        asm volatile("" : "+s"(o_a1), "+s"(o_t1), "+s"(wv));
        int ln = lane;
        asm volatile("" : "+v"(ln));
        const int j = ln & 15, kq = ln >> 4;
        int* s_t1 = reinterpret_cast<int*>(smem + o_t1);
        unsigned short* s_a1 = reinterpret_cast<unsigned short*>(smem + o_a1);      // f16 piece planes [2][S*r1][64]
        const int* rowtab = reinterpret_cast<const int*>(opaque_global(reinterpret_cast<const u32x4*>(a.rowtab)));
        const ConvJob& J = a.job[job_of(gid)];
        const int b0 = (gid - J.wg0) * a.S;
        const int ns = min(a.S, J.batch - b0);
        const int M1 = ns * r1, M2 = ns * r2, M3 = ns * r3;
        const u8* s_in = smem + (cur ? a.off_obs1 : 0);
        unsigned short* s_a2 = reinterpret_cast<unsigned short*>(smem + (cur ? a.off_a2b : 0));     // [2][S*r2][40]; overlays this group's observations (dead after conv1)
        // this wave's older memory operations retire here: its pieces of this group's observations (requested a group ago) and this group's
        // first-layer weights -- BEFORE the next group's DMA is issued (see the kernel's header)
        __builtin_amdgcn_s_waitcnt(0x0F70);                         // vmcnt(0)
        if constexpr (!CP) {
#pragma unroll
            for (int h = 0; h < NH1; ++h)
#pragma unroll
                for (int e = 0; e < 8; ++e) ko[h][e] = max(ko[h][e], 0);
        }
        __syncthreads();                                            // every wave's pieces have landed; the previous group's LDS images are dead
        DQ_STAMP(DQ_TAG_CONV_FWD, 8 * gk + 1);
        const int nxt = gid + gstride;
        if (nxt < total) issue(pre, cur ^ 1);                       // block-uniform
        DQ_STAMP(DQ_TAG_CONV_FWD, 8 * gk + 2);
        // ---- convolution 1 (conv_chain_kernel's, on rows of 64 halves) ------------------------------------------------------------
        if constexpr (CP) {
            conv1_patch_words(s_in, s_t1, s_lut, wb, bp1, bpx, reinterpret_cast<const float*>(J.packed + a.pk_b1p), a.slot, s_a1, lo1, M1, wv, j, kq, (MF1 + 15) / 16 - 1, rbad);
        } else {
            const int tiles = (M1 + 15) >> 4;
            auto origin = [&](int tile) {                           // (constant table entry + the sample's alignment offset of THIS group)
                const int e = s_t1[min(tile * 16 + j, M1 - 1)];
                return (e & 0x1ffff) + s_mis[cur * 16 + (e >> 20)];
            };
            auto rd = [&](int org, u32 (&ab)[NH1][8]) {
                const u8* ap = s_in + org;
#pragma unroll
                for (int h = 0; h < NH1; ++h)
#pragma unroll
                    for (int e = 0; e < 8; ++e) ab[h][e] = ap[ko[h][e]];        // 0 or 1
            };
            auto tile_out = [&](int tile, const u32 (&ab)[NH1][8]) {
                f32x4 acc[4], accl[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[t] = acc[t]; }
#pragma unroll
                for (int h = 0; h < NH1; ++h) {
                    u32x4 av;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) av[e >> 1] = __umul24(ab[h][e] | (ab[h][e + 1] << 16), 0x3c00u);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = MFMA_F16(av, wb[0][h][t], acc[t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) accl[t] = MFMA_F16(av, wb[1][h][t], accl[t]);
                }
                f32x4 vs[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) vs[t] = f16x2_sum(acc[t], accl[t]) + f32x4{bias1[t], bias1[t], bias1[t], bias1[t]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mo = tile * 16 + 4 * kq + r;
                    if (mo >= M1) continue;
                    f32x4 v;
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = relu1(vs[t][r]);
                    range_track4(rbad, v);
                    u32 hp[2], lp[2];                               // split on write: this lane's 4 consecutive channels of pixel mo
                    split_f16x2_pair(v[0], v[1], hp[0], lp[0]);
                    split_f16x2_pair(v[2], v[3], hp[1], lp[1]);
                    unsigned short* dst = s_a1 + mo * 64 + 4 * j;
                    *reinterpret_cast<uint2*>(dst) = uint2{hp[0], hp[1]};
                    *reinterpret_cast<uint2*>(dst + lo1) = uint2{lp[0], lp[1]};
                }
            };
            u32 abA[NH1][8], abB[NH1][8];
            const int w1 = wv;                                      // (tiles to the waves in reverse order -- the odd 13th tile to the wave the last convolution leaves idle -- measured: +0.4 us)
#if CONV1_PIPE
            // software pipeline across the wave's tiles: bytes two tiles ahead, MFMAs one tile ahead, epilogue of the current tile -- the MFMAs of tile
            // t + 4 and the epilogue of tile t are independent and sit in ONE basic block (no row guards: the planes hold whole tiles)
            auto mma = [&](const u32 (&ab)[NH1][8], f32x4 (&acc)[4], f32x4 (&accl)[4]) {
#pragma unroll
                for (int t = 0; t < 4; ++t) { acc[t] = f32x4{0.f, 0.f, 0.f, 0.f}; accl[t] = acc[t]; }
#pragma unroll
                for (int h = 0; h < NH1; ++h) {
                    u32x4 av;
#pragma unroll
                    for (int e = 0; e < 8; e += 2) av[e >> 1] = __umul24(ab[h][e] | (ab[h][e + 1] << 16), 0x3c00u);
#pragma unroll
                    for (int t = 0; t < 4; ++t) acc[t] = MFMA_F16(av, wb[0][h][t], acc[t]);
#pragma unroll
                    for (int t = 0; t < 4; ++t) accl[t] = MFMA_F16(av, wb[1][h][t], accl[t]);
                }
            };
            auto epi = [&](int tile, const f32x4 (&acc)[4], const f32x4 (&accl)[4]) {
                f32x4 vs[4];
#pragma unroll
                for (int t = 0; t < 4; ++t) vs[t] = f16x2_sum(acc[t], accl[t]) + f32x4{bias1[t], bias1[t], bias1[t], bias1[t]};
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int mo = tile * 16 + 4 * kq + r;
                    f32x4 v;
#pragma unroll
                    for (int t = 0; t < 4; ++t) v[t] = relu1(vs[t][r]);
                    range_track4(rbad, v);
                    u32 hp[2], lp[2];
                    split_f16x2_pair(v[0], v[1], hp[0], lp[0]);
                    split_f16x2_pair(v[2], v[3], hp[1], lp[1]);
                    unsigned short* dst = s_a1 + mo * 64 + 4 * j;
                    *reinterpret_cast<uint2*>(dst) = uint2{hp[0], hp[1]};
                    *reinterpret_cast<uint2*>(dst + lo1) = uint2{lp[0], lp[1]};
                }
            };
            auto interleave = [&]() {
#if CONV1_PIPE == 2
                // 16 MFMAs, ~110 VALU, 17 LDS reads, 8 LDS writes in the block: one MFMA, then what fits into its shadow
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);      // MFMA
                    __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);      // VALU
                    __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);      // DS read
                    if (i & 1) __builtin_amdgcn_sched_group_barrier(0x200, 1, 0);      // DS write
                }
#endif
            };
            if (w1 < tiles) {
                f32x4 accA[4], acclA[4], accB[4], acclB[4];
                int tile = w1;
                rd(origin(tile), abA);
                rd(origin(tile + CONV_WAVES), abB);
                int orgN = origin(tile + 2 * CONV_WAVES);
                mma(abA, accA, acclA);
                for (;;) {
                    if (tile + CONV_WAVES >= tiles) { epi(tile, accA, acclA); break; }
                    rd(orgN, abA); orgN = origin(tile + 3 * CONV_WAVES);
                    mma(abB, accB, acclB); epi(tile, accA, acclA); interleave();
                    tile += CONV_WAVES;
                    if (tile + CONV_WAVES >= tiles) { epi(tile, accB, acclB); break; }
                    rd(orgN, abB); orgN = origin(tile + 3 * CONV_WAVES);
                    mma(abA, accA, acclA); epi(tile, accB, acclB); interleave();
                    tile += CONV_WAVES;
                }
            }
#else
            if (w1 < tiles) {
                int orgB = origin(w1 + CONV_WAVES), orgA;
                rd(origin(w1), abA);
                for (int tile = w1;;) {
                    rd(orgB, abB); orgA = origin(tile + 2 * CONV_WAVES); tile_out(tile, abA); tile += CONV_WAVES; if (tile >= tiles) break;
                    rd(orgA, abA); orgB = origin(tile + 2 * CONV_WAVES); tile_out(tile, abB); tile += CONV_WAVES; if (tile >= tiles) break;
                }
            }
#endif
        }
        DQ_STAMP(DQ_TAG_CONV_FWD, 8 * gk + 3);
        // ---- convolutions 2 and 3 ----------------------------------------------------------------------------------------------------
        F16x2 ring[4][2];
        conv_w_prefetch(ring, J.packed + PK_CONV2_FWD, ln);
        __syncthreads();
        DQ_STAMP(DQ_TAG_CONV_FWD, 8 * gk + 4);
        conv_from_lds<64, 32, 2, 4, 2, false, 0>(s_a1, lo1, a.oh1, a.ow1, a.oh2, a.ow2, M2, ring, J.packed + PK_CONV2_FWD, J.params + a.b_off[1], s_a2, lo2,
                                                 nullptr, wv, ln, rowtab + 3 * CONV_ROWTAB, tab2[0], tab2[1], rbad);
        DQ_STAMP(DQ_TAG_CONV_FWD, 8 * gk + 5);
        conv_w_prefetch(ring, J.packed + PK_CONV3_FWD, ln);
        __syncthreads();
        DQ_STAMP(DQ_TAG_CONV_FWD, 8 * gk + 6);
        conv_from_lds<32, 32, 2, 4, 2>(s_a2, lo2, a.oh2, a.ow2, a.oh3, a.ow3, M3, ring, J.packed + PK_CONV3_FWD, J.params + a.b_off[2], nullptr, 0,
                                       J.act_out[2] + (size_t)b0 * r3 * 32, wv, ln, rowtab + 2 * CONV_ROWTAB, tab3[0], tab3[1], rbad);
        // the next group's first-layer weights fly over the training stores and the next group's top barrier (unconditional -- after the last group
        // the last job's again --: a conditional reload would keep these 84 registers live through the whole group)
        load_w1(a.job[job_of(min(nxt, total - 1))]);
        pre = prep(min(nxt + gstride, total - 1));                  // scalar loads only; consumed at the top of the next group
        // ---- training: a1 and a2 leave as the piece planes they are in LDS, in one burst of 16-byte copies ---------------------------
        if (J.write_all) {                                          // block-uniform; a1 is dead but intact since conv2, a2 since conv3
            __syncthreads();
            for (int i = tid; i < 2 * M1 * 8; i += CONV_THREADS) {
                const int piece = i >= M1 * 8 ? 1 : 0, q = i - piece * M1 * 8;
                *reinterpret_cast<u32x4*>(J.a1_pl + piece * J.a1_lo + (size_t)b0 * r1 * 64 + 8 * q) = *reinterpret_cast<const u32x4*>(s_a1 + piece * lo1 + 8 * q);
            }
            for (int i = tid; i < 2 * M2 * 4; i += CONV_THREADS) {
                const int piece = i >= M2 * 4 ? 1 : 0, q = i - piece * M2 * 4, row = q >> 2, part = q & 3;
                *reinterpret_cast<u32x4*>(J.a2_pl + piece * J.a2_lo + ((size_t)b0 * r2 + row) * 32 + 8 * part) =
                    *reinterpret_cast<const u32x4*>(s_a2 + piece * lo2 + row * 40 + 8 * part);
            }
        }
        DQ_STAMP(DQ_TAG_CONV_FWD, 8 * gk + 7);
        if (nxt >= total) break;
        gid = nxt; cur ^= 1; ++gk;
    }
    range_report(rbad, a.range_flag);
}

// ---------------------------------------------------------------------------------------------------------------
#ifndef DENSE_PR
#define DENSE_PR 32                     // rows per reduction pass of the dense chain's Dense(|A|) partials (64 = one pass: measured)
#endif
struct DenseJob {
    const float* params;
    const u32x4* packed;                // f16 pieces (qnet.h); the hidden layer's blocks start at DenseChainArgs.pk_dense1
    const float* x;                     // [batch, K1]: NHWC flatten of the last convolution
    const unsigned short* xp;           // != NULL (round 6, behind conv_wave_kernel): the same rows as ready-made f16 piece planes [batch][K1] (h plane; the l plane xp_lo
    size_t xp_lo;                       // halves further), staged by LDS-DMA; x is then not read
    int batch;
    float keep_scale;                   // > 0: dropout active on the hidden layer's output
    u32 drop_T;                         // a unit is dropped iff its 16-bit draw < drop_T (dq_rate_threshold16)
    const u32* keep_bits;               // != NULL: the keep bits of this job's samples, drawn ahead ([batch][16] words, qnet.h keep_bits): loaded, not drawn
    u32 seed0, seed1, sample_base;
    u64 t;
    unsigned short* x_pl;               // training: the input rows as f16 piece planes [2][plane_rows][K1] for the dense weight gradients
    unsigned short* h1_pl;              // training: the hidden output as planes [2][plane_rows][512]
    unsigned short* y2_pl;              // training: Dense(|A|)'s output as planes [2][plane_rows][small_ld] (the dueling layer's weight gradient)
    int plane_rows, small_ld;
    float* q_out;                       // [batch, n_actions]
    int wg0;                            // first workgroup of this job
};

struct DenseChainArgs {
    DenseJob job[FWD_MAX_JOBS];
    int n_jobs;
    int wg_first[FWD_MAX_JOBS];         // as in ConvChainArgs
    int K1, perm_hw, perm_c;            // Keras Flatten: k = c*hw + p reads x[p*perm_c + c]
    int pk_dense1;                      // u32x4 offset of the hidden layer's packed blocks [K1/32][32 column tiles]
    int pk_dense2;                      // ... of Dense(|A|)'s [16][NT2]
    int pk_w3q;                         // ... of the folded dueling layer W3' / b3' (f32, PackLayout.w3q)
    int N2, N3, n_actions;              // Dense(|A|) width, dueling layer width (0 = no dueling layer)
    int w_off[3], b_off[3];
    int ldx, ld2, ld3;                  // LDS row strides (floats)
    int off_x, off_h, off_part, off_y2, off_y3;
    unsigned* range_flag;               // the forward's range guard (qnet.h range_report), nullable
};

// LEAN (round 6; NT2 = 4, RT = 2 only; measured SLOWER, off by default -- see fused_forward_multi): the same arithmetic in at most 128 VGPRs, so that TWO workgroups
// share a CU (4 waves per SIMD), in the hope that one's staging, epilogues and reductions run under the other's MFMAs -- the one-workgroup-per-CU form runs its phases
// one after the other with the matrix pipe busy ~40 % of the kernel.  What it gives up: the hidden layer's weights are held as ONE ring of four tiles refilled in place (no second block in flight: the other waves of the SIMD cover the latency),
// Dense(|A|)'s weights come one K block at a time behind the hidden epilogues of BOTH row tiles (the accumulators die first), the folded dueling layer's operand is
// requested where it is used; 32 rows per weight stream instead of 64 (twice the L2 -> CU bytes, under the MFMAs of twice the waves).  Every accumulator sees the
// same products in the same order as in the other forms: the same bits.
template <int NT2, int KG3, int RT, bool LEAN = false>     // N2 <= 16*NT2 (column tiles of Dense(|A|)); N2 <= 16*KG3 (k groups of the dueling layer);
                                        // RT row tiles of 16 samples per workgroup
__global__ __launch_bounds__(DENSE_THREADS, LEAN ? 4 : 2) void dense_chain_kernel(DenseChainArgs a) {
    static_assert(!LEAN || (NT2 == 4 && RT == 2), "the lean form is written for 32-sample workgroups of the 64-action geometry");
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    float* s_x = reinterpret_cast<float*>(smem + a.off_x);
    float* s_part = reinterpret_cast<float*>(smem + a.off_part);
    float* s_y2 = reinterpret_cast<float*>(smem + a.off_y2);
    constexpr int PW = 16 * NT2, ROWS = 16 * RT, PR = ROWS < DENSE_PR ? ROWS : DENSE_PR, UP = PR / 16;      // PR rows (UP row tiles) per reduction pass
    constexpr int PWP = PW + 4;                                     // row stride of the partials: the 8 lanes of a ds_write_b128 lane group are 8 ROWS -- unpadded (a
                                                                    // multiple of 32 banks) every store was an 8-way conflict, 13K cycles of this kernel's LDS time
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
    static_assert(FWD_MAX_JOBS == 4, "three comparisons");
    const int jb = ((int)blockIdx.x >= a.wg_first[1]) + ((int)blockIdx.x >= a.wg_first[2]) + ((int)blockIdx.x >= a.wg_first[3]);      // block-uniform
    const DenseJob& J = a.job[jb];
    const int b0 = ((int)blockIdx.x - J.wg0) * ROWS;
    const int ns = min(ROWS, J.batch - b0);
    const int K1 = a.K1;
    range_mask rbad = 0;                                            // the forward's range guard (qnet.h range_track): the hidden layer and the Q-values (non-finite) here; the
                                                                    // convolutions' outputs -- this kernel's input -- are tracked where they are produced; Dense(|A|)'s output is
                                                                    // split for the backward only, whose own guard sees a non-finite gradient

    DQ_STAMP(DQ_TAG_DENSE_FWD, 0);
    DQ_STAMP_WG(DQ_TAG_DENSE_FWD, 0);
    DQ_STAMP_PAIR(2);
    // ---- hidden layer's first weight blocks start flying before anything else -------------------------------------------
    // Both dense layers run TRANSPOSED on the f16 pipe (f16x2, qnet.h): out^T = W^T x^T, the packed weight pieces are the MFMA's FIRST
    // operand and the samples are its columns.  The accumulator layout (lane (kq, j): rows 4kq .. 4kq+3 of column j) then hands every lane
    // four consecutive hidden units of ONE sample, and with the tiles' rows numbered unit = 64 wave + 32 b + 8 (i >> 2) + 4 s + (i & 3)
    // (tile (b, s), row i: a permutation that lives in the packed weights) the two tiles (b, 0), (b, 1) give lane (kq, j) exactly units
    // 8kq .. 8kq+7 of block b of sample j -- the SECOND operand of Dense(|A|) for that K = 32 block.  So the hidden layer never goes
    // through LDS: bias, ReLU, dropout and the split into pieces happen in registers, and Dense(|A|) (K = 512 split over the 8 waves,
    // each over the 64 units it has just produced) consumes them in place.  LDS holds only the input rows' planes -- which is what lets a
    // workgroup take RT = 4 row tiles (64 samples) per pass over the weights: the kernel is bound by the weight stream (every workgroup
    // reads all 0.6 MB of Dense(512) pieces through the CU's 64 B/clk vector-memory path, and all workgroups together through L2), so
    // bytes per sample are what counts.
    const int KB = K1 >> 5;                                          // k-blocks of 32
    const u32x4* pkw = J.packed + a.pk_dense1 + (size_t)(4 * wave) * PK_BLOCK + lane;
    F16x2 bw[2][4];                                                 // two k-blocks in flight; tile ct = 2 b + s
#pragma unroll
    for (int t = 0; t < 4; ++t) { bw[0][t].h = pkw[t * PK_BLOCK]; bw[0][t].l = pkw[t * PK_BLOCK + PK_LO]; }

    // ---- input rows -> two f16 planes in LDS (zero-filled past the batch), in the rows' own NHWC order: Keras' channels_first
    //      Flatten is a permutation of k, applied ONCE to the packed weight rows by pack_weights_kernel instead of to every input row
    //      here; clear the padded y2 image ---------------------------------------------------------------------------------------
    const int LDP = K1 + 8;                                          // plane row stride in f16 (rows stay 16-byte aligned)
    unsigned short* s_pl = reinterpret_cast<unsigned short*>(s_x);   // [2][ROWS][LDP]
    // dropout keep bits of this lane's units (row 16u + j; byte b = the eight units 64 wave + 32 b + 8 kq .. + 7), drawn HERE, while the input rows
    // are in flight: one Philox call = eight consecutive units of a sample, 16 bits per decision (unit n draws half-word n & 7 of call n >> 3,
    // low half of a word first).  In the epilogue passes the 20 quarter-rate multiplies per call were on the training workgroups' critical
    // path -- the workgroups that set this kernel's duration.
    u32 keepb[RT];
#pragma unroll
    for (int u = 0; u < RT; ++u) keepb[u] = 0xffffu;
    // (drawn ahead by the previous backward's final reduction where the caller's loop lets it guess this forward -- qnet.h keep_bits --: then
    // two words per row tile, requested with the input rows)
    uint2 kbw[RT];
    const bool kb_ahead = J.keep_scale > 0.f && J.keep_bits != nullptr;      // block-uniform
    if (kb_ahead) {
#pragma unroll
        for (int u = 0; u < RT; ++u)
            kbw[u] = *reinterpret_cast<const uint2*>(J.keep_bits + (size_t)min(b0 + 16 * u + j, J.batch - 1) * 16 + 2 * wave);
    }
    auto draw_keep_bits = [&]() {
        if (!(J.keep_scale > 0.f) || DQ_EXP_NODROP) return;         // block-uniform
        if (kb_ahead) {                                             // word 2 wave + b: units 64 wave + 32 b .. + 31; this lane's eight: byte kq
#pragma unroll
            for (int u = 0; u < RT; ++u) keepb[u] = ((kbw[u].x >> (8 * kq)) & 0xffu) | (((kbw[u].y >> (8 * kq)) & 0xffu) << 8);
            return;
        }
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            u32 bits = 0;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                u32 wd[4];
                philox4x32_10((u32)J.t, (u32)(J.t >> 32), J.sample_base + (u32)(b0 + 16 * u + j),
                              ((u32)(64 * wave + 32 * b + 8 * kq) >> 3) | ((u32)DQ_STREAM_DROPOUT << 16), J.seed0, J.seed1, wd);
#pragma unroll
                for (int e = 0; e < 8; ++e) bits |= (((wd[e >> 1] >> (16 * (e & 1))) & 0xffffu) >= J.drop_T ? 1u : 0u) << (8 * b + e);
            }
            keepb[u] = bits;
        }
    };
    if (J.xp) {                                                     // block-uniform
        // the rows arrive as piece planes (conv_wave_kernel splits on write): global -> LDS by LDS-DMA, 16-byte slots in lane order -- slot q of a plane = part
        // q % (K1 / 8 + 1) of row q / (K1 / 8 + 1), the last part of a row being its padding (that lane copies nothing).  No registers, no vector ALU work; rows past the
        // batch re-read its last row (their results are never stored).  In training the planes ARE the weight gradients' operand: nothing to write back.
        const int spr = (K1 >> 3) + 1, slots = ROWS * spr, chunks = (slots + 63) >> 6;
        for (int c = wave; c < 2 * chunks; c += DENSE_WAVES) {
            const int piece = c >= chunks ? 1 : 0, ch = c - piece * chunks;
            const int q = ch * 64 + lane, row = q / spr, part = q - row * spr;
            if (q < slots && part < spr - 1)
                lds_dma16(J.xp + piece * J.xp_lo + (size_t)(b0 + min(row, ns - 1)) * K1 + part * 8, lds_addr(s_pl + piece * ROWS * LDP + ch * 512));
        }
        draw_keep_bits();
        for (int i = tid; i < ROWS * a.ld2; i += DENSE_THREADS) s_y2[i] = 0.f;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // the copies (and the first weight block requested above) have landed
    } else {
        // eight consecutive values per item: two 16-byte loads in, ONE 16-byte store per piece out (LDS plane, and in training the global
        // plane: stores are issue-bound per instruction -- as pairs of 8-byte stores they were twice as many for the same bytes)
        const int q8 = K1 >> 3;                                     // items per row (K1 is a multiple of 32)
        constexpr int NBS = RT == 4 ? 5 : RT == 2 ? 3 : 2;           // items in flight per thread before the first split / LDS store: the whole image
        bool drawn = false;
        for (int i0 = tid; i0 < ROWS * q8; i0 += NBS * DENSE_THREADS) {
            f32x4 vv[NBS][2];
#pragma unroll
            for (int u = 0; u < NBS; ++u) {
                const int i = i0 + u * DENSE_THREADS, r = i / q8, c8 = (i - r * q8) * 8;
                const bool ok = i < ROWS * q8 && r < ns;
                // unconditional loads of a clamped address, masked by multiplication (a select right behind the load serialises them)
                const float* xp = J.x + (ok ? (size_t)(b0 + r) * K1 + c8 : 0);
                vv[u][0] = *reinterpret_cast<const f32x4*>(xp) * (ok ? 1.f : 0.f);
                vv[u][1] = *reinterpret_cast<const f32x4*>(xp + 4) * (ok ? 1.f : 0.f);
            }

            if (i0 == tid) { draw_keep_bits(); drawn = true; }      // (NBS covers the whole image: the loop body runs at most once)
#pragma unroll
            for (int u = 0; u < NBS; ++u) {
                const int i = i0 + u * DENSE_THREADS;
                if (i >= ROWS * q8) break;
                const int r = i / q8, c8 = (i - r * q8) * 8;
                const F16x2 o = split_f16x2(vv[u][0], vv[u][1]);
                unsigned short* d = s_pl + r * LDP + c8;
                *reinterpret_cast<u32x4*>(d) = o.h;
                *reinterpret_cast<u32x4*>(d + ROWS * LDP) = o.l;
                if (J.x_pl && r < ns && !DQ_EXP_NOSTORE) {                             // the same pieces feed the weight gradient of this layer (fused_bwd.hip)
                    unsigned short* gp = J.x_pl + (size_t)(b0 + r) * K1 + c8;
                    *reinterpret_cast<u32x4*>(gp) = o.h;
                    *reinterpret_cast<u32x4*>(gp + (size_t)J.plane_rows * K1) = o.l;
                }
            }
        }
        if (!drawn) draw_keep_bits();                               // (threads that hold no input item: small K1)
        for (int i = tid; i < ROWS * a.ld2; i += DENSE_THREADS) s_y2[i] = 0.f;
    }
    __syncthreads();

    DQ_STAMP(DQ_TAG_DENSE_FWD, 1);
    // ---- Dense(512)^T: wave w owns units [64w, 64w+64) as 4 tiles; two accumulators per tile (leading / cross terms) ---------------
    f32x4 acc[RT][4][2];
#pragma unroll
    for (int u = 0; u < RT; ++u)
#pragma unroll
        for (int t = 0; t < 4; ++t) { acc[u][t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc[u][t][1] = acc[u][t][0]; }
    const unsigned short* xrow = s_pl + j * LDP + 8 * kq;
    auto x_read = [&](int b, int u, F16x2& xv) {
        const unsigned short* xp = xrow + 16 * u * LDP + 32 * b;
        xv.h = *reinterpret_cast<const u32x4*>(xp);
        xv.l = *reinterpret_cast<const u32x4*>(xp + ROWS * LDP);
    };
    // the input pieces of step (b, u) + 1 are read from LDS BEFORE step (b, u)'s MFMAs are issued (DENSE_XPIPE; xq[parity of the step]):
    // read next to their use, every row tile of every block waited out one LDS latency with the matrix pipe idle
    F16x2 xq[2];
    if (DENSE_XPIPE && !LEAN) x_read(0, 0, xq[0]);
    auto do_block = [&](int b, F16x2 (&cur)[4], F16x2 (&nxt)[4], auto ptag) {
        constexpr int P0 = decltype(ptag)::value;                   // parity of step (b, 0)
        const u32x4* pn = pkw + (size_t)min(b + 1, KB - 1) * 32 * PK_BLOCK;      // next block: unconditional, clamped prefetch
#pragma unroll
        for (int t = 0; t < 4; ++t) { nxt[t].h = pn[t * PK_BLOCK]; nxt[t].l = pn[t * PK_BLOCK + PK_LO]; }
        if (DENSE_PIN) __builtin_amdgcn_sched_barrier(0);           // the requests stay HERE, a whole block of MFMAs ahead of their use (hipcc otherwise sinks
                                                                    // each next to its use -- a few MFMAs ahead -- and the wave waits out one L2 latency per pair)
#pragma unroll
        for (int u = 0; u < RT; ++u) {
            const int pp = (P0 + u) & 1;
            if (DENSE_XPIPE) {
                if (u + 1 < RT) x_read(b, u + 1, xq[pp ^ 1]);
                else x_read(min(b + 1, KB - 1), 0, xq[pp ^ 1]);     // (past the last block: re-reads it, unused)
                __builtin_amdgcn_sched_barrier(0);
            } else x_read(b, u, xq[pp]);
#pragma unroll
            for (int t = 0; t < 4; ++t) mma_f16x3(cur[t], xq[pp], acc[u][t][0], acc[u][t][1]);
        }
    };
    using par0 = std::integral_constant<int, 0>;
    using par1 = std::integral_constant<int, (RT & 1)>;             // an odd RT alternates the parity from block to block
    if constexpr (LEAN) {
        // one ring of four tiles (bw[0]), each slot refilled with the NEXT block's tile right behind the MFMAs that read it; the block's input pieces for both
        // row tiles read at its top
        for (int blk = 0; blk < KB; ++blk) {
            const u32x4* pn = pkw + (size_t)min(blk + 1, KB - 1) * 32 * PK_BLOCK;      // (past the last block: re-reads it, unused)
            F16x2 xv[RT];
#pragma unroll
            for (int u = 0; u < RT; ++u) x_read(blk, u, xv[u]);
#pragma unroll
            for (int t = 0; t < 4; ++t) {
#pragma unroll
                for (int u = 0; u < RT; ++u) mma_f16x3(bw[0][t], xv[u], acc[u][t][0], acc[u][t][1]);
                bw[0][t].h = pn[t * PK_BLOCK]; bw[0][t].l = pn[t * PK_BLOCK + PK_LO];
            }
        }
    } else {
    int blk = 0;
    for (; blk + 1 < KB; blk += 2) {                                // no condition around the MFMAs inside the loop
        do_block(blk, bw[0], bw[1], par0{});
        do_block(blk + 1, bw[1], bw[0], par1{});
    }
    if (blk < KB) do_block(blk, bw[0], bw[1], par0{});               // odd block count (K1 = 288: 9 blocks)
    }

    DQ_STAMP(DQ_TAG_DENSE_FWD, 2);
    // ---- Dense(|A|)^T's weights for this wave's 64 units (two K = 32 blocks x NT2 tiles of 16 outputs: qnet.h dense2) start flying
    //      under the hidden layer's epilogue ---------------------------------------------------------------------------------------
    // WIDE2 (|A| > 64 with 32-sample workgroups, round 4; selected by DQ_DENSE_RT=2 only: measured at c5 -- 4096 rows of d = 7 -- it is SLOWER than 16-sample
    // workgroups, 29.1 against 25.9 us: 128 workgroups leave half the CUs idle and the launch is one round of latency-bound workgroups either way): all NT2 = 8 column tiles as pieces are 128 registers, beside the 64 of two row tiles' hidden
    // accumulators they do not fit -- the passes below then form the hidden pieces of BOTH row tiles first (the accumulators die there) and walk
    // Dense(|A|)'s column tiles in two halves of four, the second half's weights requested under the first half's MFMAs
    constexpr bool WIDE2 = NT2 == 8 && RT == 2;
    constexpr int NTW = WIDE2 ? 4 : NT2;                            // column tiles of Dense(|A|) held as pieces at a time
    const u32x4* pk2 = J.packed + a.pk_dense2 + (size_t)(2 * wave) * NT2 * PK_BLOCK + lane;
    F16x2 w2[2][NTW];
    auto load_w2 = [&](F16x2 (&w)[2][NTW], int t0) {
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int t = 0; t < NTW; ++t) { w[b][t].h = pk2[(b * NT2 + t0 + t) * PK_BLOCK]; w[b][t].l = pk2[(b * NT2 + t0 + t) * PK_BLOCK + PK_LO]; }
    };
    if constexpr (!WIDE2 && !LEAN) load_w2(w2, 0);
    // The dueling layer Dense(|A| + 1) and the combination Q = V + A - mean(A) are ONE linear map of Dense(|A|)'s output: pack_weights_kernel
    // folds them into W3' [16 KG3][16 NT2] (zero past N2 / past |A|) and b3' (PackLayout.w3q), so that the last phase's MFMA accumulators ARE the
    // Q-values (round 3: a separate combination -- every wave reading eight rows back from LDS, butterfly sums, a barrier in between -- was 5K of a
    // workgroup's 48K cycles).  Waves 0 .. NTQ-1 x parts: column tile ct3 of Q belongs to waves ct3, ct3 + NTQ, ...: they share its row tiles.
    const int A = a.n_actions;
    const int NTQ = (A + 15) >> 4;
    float b3[KG3][4];
    f32x4 bias1[4];                                                 // this lane's four units of tile ct: 64w + 32 (ct >> 1) + 8kq + 4 (ct & 1) + r
#pragma unroll
    for (int ct = 0; ct < 4; ++ct)
        bias1[ct] = *reinterpret_cast<const f32x4u*>(J.params + a.b_off[0] + 64 * wave + 32 * (ct >> 1) + 8 * kq + 4 * (ct & 1));
    const int nt3 = NTQ > 0 ? NTQ : 1, parts = DENSE_WAVES / nt3, ct3 = wave % nt3, part = wave / nt3;
    const bool act3 = a.N3 > 0 && part < parts;
    const int col3 = 16 * ct3 + j;
    float bias3 = 0.f;
    auto load_b3 = [&]() {
        const float* w3q = reinterpret_cast<const float*>(J.packed + a.pk_w3q);
        bias3 = w3q[16 * KG3 * PW + col3];
#pragma unroll
        for (int g = 0; g < KG3; ++g)
#pragma unroll
            for (int s = 0; s < 4; ++s) b3[g][s] = w3q[(16 * g + 4 * kq + s) * PW + col3];
    };
    if (act3 && !LEAN) load_b3();                                   // (requested here, used three barriers later; unconditional: the table is zero-padded.  LEAN: where it is used)
    const int rcol = tid & 63;                                      // column (+ 64 cc) of the cross-wave reduction below
    float bias2r[(PW + 63) / 64];
#pragma unroll
    for (int cc = 0; cc < (PW + 63) / 64; ++cc) bias2r[cc] = J.params[a.b_off[1] + min(rcol + 64 * cc, a.N2 - 1)];
    __syncthreads();                                                // every wave is done with the input planes: the partials overlay them

    DQ_STAMP(DQ_TAG_DENSE_FWD, 3);
    // ---- per reduction pass (32 rows): hidden epilogue in registers -> Dense(|A|)^T partial over this wave's units -> LDS; then the
    //      fixed-order sum over the 8 waves ------------------------------------------------------------------------------------------
    // the hidden epilogue of row tile u in registers: bias, ReLU, dropout, split -> this wave's two K = 32 blocks of Dense(|A|)'s operand
    auto hidden_pieces = [&](int u, F16x2 (&hb)[2]) {
        const int row = 16 * u + j;                                 // this lane's sample
        float rm = 0.f;                                             // this epilogue's maximum (qnet.h range_max)
#pragma unroll
        for (int b = 0; b < 2; ++b) {
#pragma unroll
            for (int s = 0; s < 2; ++s) {
                const int ct = 2 * b + s;
                f32x4 v;
#pragma unroll
                for (int r = 0; r < 4; ++r) v[r] = fmaxf(f16x2_sum(acc[u][ct][0][r], acc[u][ct][1][r]) + bias1[ct][r], 0.f);
                if (J.keep_scale > 0.f) {
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = ((keepb[u] >> (8 * b + 4 * s + r)) & 1u) ? v[r] * J.keep_scale : 0.f;
                }
                range_max4(rm, v);
                u32 hp[2], lp[2];
                split_f16x2_pair(v[0], v[1], hp[0], lp[0]);
                split_f16x2_pair(v[2], v[3], hp[1], lp[1]);
                hb[b].h[2 * s] = hp[0]; hb[b].h[2 * s + 1] = hp[1];
                hb[b].l[2 * s] = lp[0]; hb[b].l[2 * s + 1] = lp[1];
            }
            if (J.h1_pl && row < ns && !DQ_EXP_NOSTORE) {           // (training: the hidden output as piece planes, see below)
                unsigned short* gp = J.h1_pl + (size_t)(b0 + row) * DENSE_HID + 64 * wave + 32 * b + 8 * kq;
                *reinterpret_cast<u32x4*>(gp) = hb[b].h;
                *reinterpret_cast<u32x4*>(gp + (size_t)J.plane_rows * DENSE_HID) = hb[b].l;
            }
        }
        range_commit(rbad, rm);
    };
    if constexpr (LEAN) {
        static_assert(!LEAN || (ROWS / PR == 1 && UP == 2), "one reduction pass over both row tiles");
        F16x2 hbw[2][2];
        hidden_pieces(0, hbw[0]);
        hidden_pieces(1, hbw[1]);                                   // (the hidden accumulators are dead from here)
#pragma unroll
        for (int uu = 0; uu < 2; ++uu) {
            f32x4 acc2[NT2][2];
#pragma unroll
            for (int t = 0; t < NT2; ++t) { acc2[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t][1] = acc2[t][0]; }
#pragma unroll
            for (int b = 0; b < 2; ++b) {                           // one K block of Dense(|A|)'s weights at a time (re-read per row tile: 16 KB per wave out of L2)
                F16x2 wb[NT2];
#pragma unroll
                for (int t = 0; t < NT2; ++t) { wb[t].h = pk2[(b * NT2 + t) * PK_BLOCK]; wb[t].l = pk2[(b * NT2 + t) * PK_BLOCK + PK_LO]; }
#pragma unroll
                for (int t = 0; t < NT2; ++t) mma_f16x3(wb[t], hbw[uu][b], acc2[t][0], acc2[t][1]);
            }
#pragma unroll
            for (int t = 0; t < NT2; ++t)
                *reinterpret_cast<f32x4*>(s_part + (wave * PR + 16 * uu + j) * PWP + 16 * t + 4 * kq) = f16x2_sum(acc2[t][0], acc2[t][1]);
        }
    }
    if constexpr (WIDE2) {
        static_assert(!WIDE2 || (ROWS / PR == 1 && UP == 2), "one reduction pass over both row tiles");
        F16x2 hbw[2][2];
        F16x2 w2b[2][NTW];
        load_w2(w2, 0);                                             // (both halves' weights fly under the epilogues)
        load_w2(w2b, NTW);
        hidden_pieces(0, hbw[0]);
        hidden_pieces(1, hbw[1]);
#pragma unroll
        for (int hlf = 0; hlf < 2; ++hlf)
#pragma unroll
            for (int uu = 0; uu < 2; ++uu) {
                f32x4 acc2[NTW][2];
#pragma unroll
                for (int t = 0; t < NTW; ++t) { acc2[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t][1] = acc2[t][0]; }
#pragma unroll
                for (int b = 0; b < 2; ++b)
#pragma unroll
                    for (int t = 0; t < NTW; ++t) mma_f16x3(hlf ? w2b[b][t] : w2[b][t], hbw[uu][b], acc2[t][0], acc2[t][1]);
#pragma unroll
                for (int t = 0; t < NTW; ++t)
                    *reinterpret_cast<f32x4*>(s_part + (wave * PR + 16 * uu + j) * PWP + 16 * (NTW * hlf + t) + 4 * kq) = f16x2_sum(acc2[t][0], acc2[t][1]);
            }
    }
#pragma unroll
    for (int pass = 0; pass < ROWS / PR; ++pass) {
#pragma unroll
        for (int uu = 0; uu < (WIDE2 || LEAN ? 0 : UP); ++uu) {
            const int u = pass * UP + uu, row = 16 * u + j;         // this lane's sample
            F16x2 hb[2];                                            // the sample's units 8kq .. 8kq+7 of this wave's two blocks, as pieces
            float rm = 0.f;                                         // this epilogue's maximum (qnet.h range_max)
#pragma unroll
            for (int b = 0; b < 2; ++b) {
#pragma unroll
                for (int s = 0; s < 2; ++s) {
                    const int ct = 2 * b + s;
                    f32x4 v;
#pragma unroll
                    for (int r = 0; r < 4; ++r) v[r] = fmaxf(f16x2_sum(acc[u][ct][0][r], acc[u][ct][1][r]) + bias1[ct][r], 0.f);
                    if (J.keep_scale > 0.f) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) v[r] = ((keepb[u] >> (8 * b + 4 * s + r)) & 1u) ? v[r] * J.keep_scale : 0.f;
                    }
                    range_max4(rm, v);
                    u32 hp[2], lp[2];
                    split_f16x2_pair(v[0], v[1], hp[0], lp[0]);
                    split_f16x2_pair(v[2], v[3], hp[1], lp[1]);
                    hb[b].h[2 * s] = hp[0]; hb[b].h[2 * s + 1] = hp[1];
                    hb[b].l[2 * s] = lp[0]; hb[b].l[2 * s + 1] = lp[1];
                }
                // training: the hidden output is kept as its piece planes ONLY (the weight gradients' operand; the backward takes its ReLU /
                // dropout mask from them too: an f32 copy beside them made the 64 training workgroups' stores the dense forward's critical
                // path, +7 us) -- one 16-byte store per piece for the lane's eight units
                if (J.h1_pl && row < ns && !DQ_EXP_NOSTORE) {
                    unsigned short* gp = J.h1_pl + (size_t)(b0 + row) * DENSE_HID + 64 * wave + 32 * b + 8 * kq;
                    *reinterpret_cast<u32x4*>(gp) = hb[b].h;
                    *reinterpret_cast<u32x4*>(gp + (size_t)J.plane_rows * DENSE_HID) = hb[b].l;
                }
            }
            range_commit(rbad, rm);
            f32x4 acc2[NTW][2];                                      // (NTW == NT2 here: the WIDE2 form ran above)
#pragma unroll
            for (int t = 0; t < NTW; ++t) { acc2[t][0] = f32x4{0.f, 0.f, 0.f, 0.f}; acc2[t][1] = acc2[t][0]; }
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int t = 0; t < NTW; ++t) mma_f16x3(w2[b][t], hb[b], acc2[t][0], acc2[t][1]);
            // C/D layout: this lane holds outputs 16t + 4kq .. + 3 of sample j: one 16-byte store per tile
#pragma unroll
            for (int t = 0; t < NTW; ++t)
                *reinterpret_cast<f32x4*>(s_part + (wave * PR + 16 * uu + j) * PWP + 16 * t + 4 * kq) = f16x2_sum(acc2[t][0], acc2[t][1]);
        }
        if (pass == 0) { DQ_STAMP(DQ_TAG_DENSE_FWD, 4); }
        __syncthreads();
        // thread (rr0 = tid >> 6, col = tid & 63) sums the 8 waves' partials of column col for rows rr0, rr0 + 8, ...: no division, the
        // bias in a register since before the first barrier (a global load per element sat on this loop's critical path), LDS reads
        // conflict-free (a lane group = 32 consecutive columns of one row)
#pragma unroll
        for (int cc = 0; cc < (PW + 63) / 64; ++cc) {
            const int col = rcol + 64 * cc;
            if (col >= a.N2) continue;
#pragma unroll
            for (int rr = tid >> 6; rr < PR; rr += DENSE_THREADS / 64) {
                const int row = pass * PR + rr;
                float v = bias2r[cc];
#pragma unroll
                for (int w = 0; w < DENSE_WAVES; ++w) v += s_part[(w * PR + rr) * PWP + col];
                s_y2[row * a.ld2 + col] = v;
            }
        }
        __syncthreads();
        if (pass == 0) { DQ_STAMP(DQ_TAG_DENSE_FWD, 5); }
    }

    // ---- training: Dense(|A|)'s output leaves as pieces (the dueling layer's weight gradient), eight columns per thread = one 16-byte store
    //      per piece (as single halves from the reduction loop these were 26 two-byte stores per thread); columns past N2 are the image's zeros
    if (J.y2_pl) {                                                  // block-uniform
        const int per_row = J.small_ld >> 3, prs = J.small_ld == 64 ? 3 : 4;        // (small_ld is 64 or 128: shifts, not a division)
        for (int i = tid; i < ROWS * per_row; i += DENSE_THREADS) {
            const int row = i >> prs, c8 = (i - (row << prs)) * 8;
            if (row >= ns) continue;
            const float* yp = s_y2 + row * a.ld2 + c8;
            const F16x2 o = split_f16x2(f32x4{yp[0], yp[1], yp[2], yp[3]}, f32x4{yp[4], yp[5], yp[6], yp[7]});
            unsigned short* gp = J.y2_pl + (size_t)(b0 + row) * J.small_ld + c8;
            *reinterpret_cast<u32x4*>(gp) = o.h;
            *reinterpret_cast<u32x4*>(gp + (size_t)J.plane_rows * J.small_ld) = o.l;
        }
    }
    DQ_STAMP(DQ_TAG_DENSE_FWD, 6);
    // ---- Q = y2 W3' + b3' (the dueling layer and its combination, folded: see above), straight from the accumulators to global memory ----
    if (a.N3 > 0) {
        if (act3) {
            if constexpr (LEAN) load_b3();
            for (int u = part; u < RT; u += parts) {
                f32x4 acc3[2] = {{0.f, 0.f, 0.f, 0.f}, {0.f, 0.f, 0.f, 0.f}};     // two chains: a dependent f32 MFMA waits for its predecessor
                const float* yrow = s_y2 + (16 * u + j) * a.ld2 + 4 * kq;
#pragma unroll
                for (int g = 0; g < KG3; ++g) {
                    const f32x4 av = *reinterpret_cast<const f32x4*>(yrow + 16 * g);
#pragma unroll
                    for (int s = 0; s < 4; ++s) acc3[s & 1] = MFMA16(av[s], b3[g][s], acc3[s & 1]);
                }
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * u + 4 * kq + r;
                    const float qv = (acc3[0][r] + acc3[1][r]) + bias3;
                    const bool live = col3 < A && row < ns;
                    if (live) J.q_out[(size_t)(b0 + row) * A + col3] = qv;
                    range_track_finite(rbad, live ? qv : 0.f);      // (outside the guard: the mask stays wave-uniform)
                }
            }
        }
        DQ_STAMP(DQ_TAG_DENSE_FWD, 7);
    } else {
        // no dueling layer: Q = Dense(|A|)'s output
        DQ_STAMP(DQ_TAG_DENSE_FWD, 7);
        constexpr int RW = ROWS / DENSE_WAVES;
#pragma unroll
        for (int u = 0; u < RW; ++u) {
            const int row = wave + DENSE_WAVES * u;
            if (row >= ns) continue;
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int c = lane + 64 * h;
                if (c < A) J.q_out[(size_t)(b0 + row) * A + c] = s_y2[row * a.ld2 + c];
            }
        }
    }
    range_report(rbad, a.range_flag);
    DQ_STAMP(DQ_TAG_DENSE_FWD, 8);
    DQ_STAMP_WG(DQ_TAG_DENSE_FWD, 1);
    DQ_STAMP_PAIR(3);
}

// ---------------------------------------------------------------------------------------------------------------
// Packs the conv kernels and Dense(512) of one parameter buffer into f16 pieces in MFMA operand order (qnet.h PK_*): one wave per
// block, lane (kb, j) gathers its 8 weights, splits them and writes 2 x 16 bytes.  One launch per parameter change.
struct PackArgs {
    const float* params;
    u32x4* pk;
    int w1_off, K1, w2_off, w3_off, d1_off, d1_blocks;
    int d2_off, N2, NT2, KB2, d2_blocks, d2t_blocks, d1t_blocks, d1k;     // Dense(|A|) kernel offset / width, dense2 / dense2t / dense1t block counts, K1
    int perm_hw, perm_c;                // > 0: plane index k' = p*perm_c + c of the dense forward is Keras weight row c*perm_hw + p
    int pack_wgs, wide_blocks;          // packing workgroups (four 64-lane blocks each); workgroups of pack_wide_wc_block in front of them
    int w3q_off, w3q_rows;              // the folded dueling layer (qnet.h w3q): u32x4 offset, rows 16 KG3 (+ 1: the bias row); 0 rows = no dueling layer
    int w3d_off, b3d_off, N3, n_actions;    // the dueling layer's kernel [N2][N3] and bias [N3] in params
    int wc_off, wc_waves;               // Wc (qnet.h wc: the dense backward's gH1 rows with the TD step fused in): u32x4 offset, waves that build it (0: none)
    const int* ptab;                    // patch-word input (qnet.h PT_*), NULL: not configured -- the two sections below are left alone
    int c1c_off, b1p_off, b1p_rows;     // u32x4 offsets of the compact first kernel (4 blocks) and of the per-pixel bias table; its rows (pixels)
    int p_depth, p_C, b1_off;           // syndrome planes, input planes, the first bias in params
    int c1w_off, c1w_blocks, p_kd;      // the wave-private conv forward's first kernel (qnet.h c1w): u32x4 offset, blocks (4 or 0), data bits per word
    int c2w_off, c2w_blocks;            // ... and its second (qnet.h c2w): 16 blocks or 0
    int cdw_off, cdw_blocks;            // the 16-wave convolutional backward's data-gradient weights (qnet.h cdw): 24 blocks or 0
    unsigned* range_flag;               // the forward's range guard: every parameter is checked finite and < 65504 here (qnet.h range_report), nullable
    int n_params;
};

// Wc for networks with more than 64 actions (N2, N3 <= 112; qnet.h wc): Wc[a][n1] = P[0] + P[1 + a] - mean_a' P[1 + a'],  P = W2[n1] W3 (the plain
// product's row).  Workgroups of their own, FIRST in the launch's grid (theirs is its longest chain): the dueling layer's kernel W3 (40 KB at d = 7) is
// staged in LDS once per workgroup -- read from L2 by every wave it was 13 dependent round trips and the launch 15.6 instead of 6.3 us.  A workgroup forms
// WCW_U = 4 hidden units; wave w takes the w-th quarter of W3's rows for ALL four (every wave reading all of W3 out of LDS, one unit each, was 160 KB of LDS
// reads per workgroup: 9.2 us), two columns per lane (c = lane, lane + 64), W2's values by v_readlane; the four partial rows are summed in wave order through LDS
// and wave u folds and stores unit u.
#define WCW_U 4
__device__ __forceinline__ void pack_wide_wc_block(const PackArgs& a, int block, float* s_w3) {
    const float* __restrict__ params = a.params;
    const int A = a.n_actions, N2 = a.N2, N3 = a.N3, lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane((int)threadIdx.x >> 6);
    float* s_part = s_w3 + N2 * N3;                                 // [wave][unit][128]
    const int n1_0 = min(block * WCW_U, DENSE_HID - WCW_U);
    // the block's WCW_U rows of W2, two values per lane (n2 = lane, lane + 64; zero past N2), requested first of all
    float w2v[WCW_U][2];
#pragma unroll
    for (int q = 0; q < WCW_U; ++q)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const float v = params[a.d2_off + (size_t)(n1_0 + q) * N2 + min(lane + 64 * h, N2 - 1)];
            w2v[q][h] = lane + 64 * h < N2 ? v : 0.f;
        }
    for (int base = 0; base < N2 * N3; base += 16 * 256) {          // (sixteen loads in flight per thread: one by one they are as many round trips)
        float v[16];
#pragma unroll
        for (int u = 0; u < 16; ++u) v[u] = params[a.w3d_off + min(base + u * 256 + (int)threadIdx.x, N2 * N3 - 1)];
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            const int i = base + u * 256 + (int)threadIdx.x;
            if (i < N2 * N3) s_w3[i] = v[u];
        }
    }
    __syncthreads();
    const int c0 = min(lane, N3 - 1), c1 = min(lane + 64, N3 - 1);
    float acc[WCW_U][2];
#pragma unroll
    for (int q = 0; q < WCW_U; ++q) acc[q][0] = acc[q][1] = 0.f;
    const int quarter = (N2 + 3) >> 2, r0 = wave * quarter, r1 = min(N2, r0 + quarter);
    for (int n2 = r0; n2 < r1; ++n2) {                              // wave-uniform bounds
        const float x0 = s_w3[n2 * N3 + c0], x1 = s_w3[n2 * N3 + c1];
#pragma unroll
        for (int q = 0; q < WCW_U; ++q) {
            const float w = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, n2 < 64 ? w2v[q][0] : w2v[q][1]), n2 & 63));
            acc[q][0] = fmaf(w, x0, acc[q][0]);
            acc[q][1] = fmaf(w, x1, acc[q][1]);
        }
    }
#pragma unroll
    for (int q = 0; q < WCW_U; ++q) {
        s_part[(wave * WCW_U + q) * 128 + lane] = acc[q][0];
        s_part[(wave * WCW_U + q) * 128 + 64 + lane] = acc[q][1];
    }
    __syncthreads();
    if (block * WCW_U + wave >= DENSE_HID) return;
    const int n1 = block * WCW_U + wave;                            // this wave's unit: its four partial rows in wave order
    float P[2];
#pragma unroll
    for (int h = 0; h < 2; ++h)
        P[h] = ((s_part[(0 * WCW_U + wave) * 128 + 64 * h + lane] + s_part[(1 * WCW_U + wave) * 128 + 64 * h + lane]) +
                s_part[(2 * WCW_U + wave) * 128 + 64 * h + lane]) + s_part[(3 * WCW_U + wave) * 128 + 64 * h + lane];
    const bool in0 = lane >= 1 && lane <= A, in1 = lane + 64 <= A;  // columns 1 .. A hold the advantages
    float adv = (in0 ? P[0] : 0.f) + (in1 ? P[1] : 0.f);
    for (int m = 32; m >= 1; m >>= 1) adv += __shfl_xor(adv, m);
    const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, P[0])));      // lane 0: the V column
    float* wc = reinterpret_cast<float*>(a.pk + a.wc_off);
    if (in0) wc[(size_t)(lane - 1) * DENSE_HID + n1] = (v0 + P[0]) - adv / (float)A;
    if (in1) wc[(size_t)(lane + 63) * DENSE_HID + n1] = (v0 + P[1]) - adv / (float)A;
}

__global__ __launch_bounds__(256) void pack_weights_kernel(PackArgs a) {
    extern __shared__ __attribute__((aligned(16))) float pack_smem[];
    // (block-uniform; the Wc workgroups come FIRST in the grid: theirs is the launch's longest chain -- stage, barrier, product, scattered stores)
    if ((int)blockIdx.x < a.wide_blocks) { pack_wide_wc_block(a, (int)blockIdx.x, pack_smem); return; }
    const int pblock = (int)blockIdx.x - a.wide_blocks;
    const float* __restrict__ params = a.params;
    // the forward's range guard, parameter half: what the packed pieces are made from must be finite and < 65504 (an h piece of inf / NaN poisons whole
    // contractions and the next ReLU hides it, qnet.h).  The launch's threads walk the flat buffer once beside their packing work (0.77 MB out of L2)
    if (a.range_flag) {
        bool bad = false;
        for (int i = pblock * 256 + (int)threadIdx.x; i < a.n_params; i += a.pack_wgs * 256) bad = bad || !(fabsf(params[i]) < 65504.f);
        if (bad) atomicOr(a.range_flag, 2u);
    }
    u32x4* __restrict__ pk = a.pk;
    const int w2_off = a.w2_off, w3_off = a.w3_off, d1_off = a.d1_off, d1_blocks = a.d1_blocks;
    const int lane = threadIdx.x & 63, blk_id = pblock * 4 + (threadIdx.x >> 6), j = lane & 15, kb = lane >> 4;
    const int e_d1 = PK_TOTAL_BLOCKS + d1_blocks, e_d2 = e_d1 + a.d2_blocks, e_d2t = e_d2 + a.d2t_blocks, e_d1t = e_d2t + a.d1t_blocks;
    if (blk_id >= e_d1t) {
        // the dueling layer folded with its combination (qnet.h w3q): one wave per row k of the dueling kernel (row w3q_rows: its bias),
        // lane = action a (+ 64): out[a] = (V + A_a) - mean_a' A_a', the mean by a fixed-order butterfly
        const int r = blk_id - e_d1t;
        const int n_w3 = (a.w3q_rows ? a.w3q_rows + 1 : 0) + a.wc_waves;
        if (r >= n_w3) {
            // the first convolution over patch words (qnet.h c1c, b1p; include/deepq_hip.h dq_env_patch_output).  Output pixel (oy, ox) of
            // Conv2D(64, 3, strides=2) on the padded planes sees 4 corner cells per syndrome plane + the centre cell per action plane as DATA;
            // every other cell of its 3 x 3 patch is a constant of the embedding, 1 at position-dependent places (ENV:284-298) -- folded into a bias per pixel.
            const int rc = r - n_w3;
            if (!a.ptab) return;
            if (rc < 4) {                                               // B(k = 8kb + e, col = 4j + rc) = W1[PT_KROW[k]][col]
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int row = a.ptab[PT_KROW + 8 * kb + e];
                    v[e] = row >= 0 ? params[a.w1_off + (size_t)row * 64 + 4 * j + rc] : 0.f;
                }
                const F16x2 o = split_f16x2(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
                u32x4* dst = pk + a.c1c_off + (size_t)rc * PK_BLOCK + lane;
                dst[0] = o.h; dst[PK_LO] = o.l;
            } else if (rc - 4 < a.b1p_rows) {                           // one wave per pixel, lane = output channel
                const int pix = rc - 4, mask = a.ptab[PT_CONST + pix];
                double acc = (double)params[a.b1_off + lane];
                float w[5][8];                                          // (all 40 loads in flight: a run-time trip count = one round trip per term; depth <= 8)
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    const int pos = a.ptab[PT_CPOS + c];
#pragma unroll
                    for (int pl = 0; pl < 8; ++pl) w[c][pl] = params[a.w1_off + (size_t)(pos * a.p_C + min(pl, a.p_depth - 1)) * 64 + lane];
                }
#pragma unroll
                for (int c = 0; c < 5; ++c) {
                    if (!((mask >> c) & 1)) continue;
#pragma unroll
                    for (int pl = 0; pl < 8; ++pl) if (pl < a.p_depth) acc += (double)w[c][pl];
                }
                reinterpret_cast<float*>(pk + a.b1p_off)[(size_t)pix * 64 + lane] = (float)acc;
            } else if (rc - 4 - a.b1p_rows < a.c1w_blocks) {            // c1w block (quarter): B(k = 8kb + e, col = 16 quarter + j), bias folded in (qnet.h)
                const int bq = rc - 4 - a.b1p_rows, col = 16 * bq + j;
                float v[8], wc[8][8];                                   // (all loads in flight: a run-time trip count = one round trip per term; depth <= 8)
                int krow[8], cpos[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 8 * kb + e;
                    krow[e] = a.ptab[PT_KROW + k];
                    cpos[e] = a.ptab[PT_CPOS + min(max(k - a.p_kd, 0), 4)];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 8 * kb + e;
                    const bool cst = k >= a.p_kd && k < a.p_kd + 5;
                    v[e] = k < a.p_kd ? params[a.w1_off + (size_t)max(krow[e], 0) * 64 + col] : k == a.p_kd + 5 ? params[a.b1_off + col] : 0.f;
#pragma unroll
                    for (int pl = 0; pl < 8; ++pl) wc[e][pl] = params[a.w1_off + (size_t)((cst ? cpos[e] : 0) * a.p_C + min(pl, a.p_depth - 1)) * 64 + col];
                }
#pragma unroll
                for (int e = 0; e < 8; ++e) {
                    const int k = 8 * kb + e;
                    if (k >= a.p_kd && k < a.p_kd + 5) {
                        double acc = 0.0;
#pragma unroll
                        for (int pl = 0; pl < 8; ++pl) if (pl < a.p_depth) acc += (double)wc[e][pl];
                        v[e] = (float)acc;
                    }
                }
                const F16x2 o = split_f16x2(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
                u32x4* dst = pk + a.c1w_off + (size_t)bq * PK_BLOCK + lane;
                dst[0] = o.h; dst[PK_LO] = o.l;
                // the forward's range guard for conv_wave_kernel's first layer (qnet.h): its operand is binary, so channel `col` of a1 can never exceed the sum of this
                // column's positive entries (data rows, constant rows and the bias row alike) -- checked here, once per parameter change, instead of per activation
                if (a.range_flag) {
                    float pos = 0.f;
#pragma unroll
                    for (int e = 0; e < 8; ++e) pos += fmaxf(v[e], 0.f);
                    pos += __shfl_xor(pos, 16); pos += __shfl_xor(pos, 32);
                    if (!(pos < 65504.f)) atomicOr(a.range_flag, 2u);
                }
            } else if (rc - 4 - a.b1p_rows - a.c1w_blocks < a.c2w_blocks) {    // c2w block (quarter, ky, t) (qnet.h)
                const int b = rc - 4 - a.b1p_rows - a.c1w_blocks, t = b & 1, ky = (b >> 1) & 1, qt = b >> 2;
                const float* w = params + w2_off + (size_t)((2 * ky + (kb >> 1)) * 64 + 16 * qt + 8 * (kb & 1)) * 32 + 2 * j + t;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = w[(size_t)e * 32];
                const F16x2 o = split_f16x2(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
                u32x4* dst = pk + a.c2w_off + (size_t)b * PK_BLOCK + lane;
                dst[0] = o.h; dst[PK_LO] = o.l;
            } else if (rc - 4 - a.b1p_rows - a.c1w_blocks - a.c2w_blocks < a.cdw_blocks) {     // cdw block: conv3 (tap, t) 0 .. 7, conv2 (tap, t) 8 .. 23 (qnet.h)
                const int b = rc - 4 - a.b1p_rows - a.c1w_blocks - a.c2w_blocks;
                const bool c3 = b < 8;
                const int bb = c3 ? b : b - 8, tap = c3 ? bb >> 1 : bb >> 2, t = c3 ? bb & 1 : bb & 3, cin = c3 ? 32 : 64;
                const float* w = params + (c3 ? w3_off : w2_off) + (size_t)(tap * cin + 16 * t + j) * 32 + 8 * kb;
                float v[8];
#pragma unroll
                for (int e = 0; e < 8; ++e) v[e] = w[e];
                const F16x2 o = split_f16x2(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
                u32x4* dst = pk + a.cdw_off + (size_t)b * PK_BLOCK + lane;
                dst[0] = o.h; dst[PK_LO] = o.l;
            }
            return;
        }
        if (a.w3q_rows == 0) return;
        if (r > a.w3q_rows) {
            // Wc [|A|][512] = W3'^T W2^T: row a = what gH1 = gY2 W2^T is for a sample whose dq is 1 at action a (times its TD error: fused_bwd.hip).
            // One wave per hidden unit n1: lane c forms P[c] = sum_n2 W2[n1][n2] W3[n2][c] (the plain product's row: W2's row through scalar loads, W3's
            // rows coalesced, n2 in order), then the dueling fold ALONG the row -- Wc[a][n1] = P[0] + P[1 + a] - mean_a' P[1 + a'] -- by a butterfly.
            const int n1 = r - a.w3q_rows - 1;
            if (n1 >= a.wc_waves) return;
            const int A = a.n_actions, N2 = a.N2, N3 = a.N3;
            const __attribute__((address_space(4))) float* w2 = (const __attribute__((address_space(4))) float*)(uintptr_t)(params + a.d2_off + (size_t)n1 * N2);
            const float* W3 = params + a.w3d_off;
            const int c = min(lane, N3 - 1);
            float acc = 0.f, wv[64];
#pragma unroll
            for (int n2 = 0; n2 < 64; ++n2) wv[n2] = W3[(size_t)min(n2, N2 - 1) * N3 + c];      // all in flight (a run-time trip count = one round trip per term)
#pragma unroll
            for (int n2 = 0; n2 < 64; ++n2) acc = fmaf(n2 < N2 ? w2[n2] : 0.f, wv[n2], acc);
            float adv = (lane >= 1 && lane <= A) ? acc : 0.f;
            for (int m = 32; m >= 1; m >>= 1) adv += __shfl_xor(adv, m);
            const float v0 = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, acc)));      // lane 0: the V column
            if (lane >= 1 && lane <= A) reinterpret_cast<float*>(pk + a.wc_off)[(size_t)(lane - 1) * DENSE_HID + n1] = (v0 + acc) - adv / (float)A;
            return;
        }
        const int pw = 16 * a.NT2, A = a.n_actions;
        const bool live = r == a.w3q_rows || r < a.N2;
        const float* src = r == a.w3q_rows ? params + a.b3d_off : params + a.w3d_off + (size_t)min(r, a.N2 - 1) * a.N3;
        float va[2], sum = 0.f;
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;
            va[h] = (live && c < A) ? src[1 + c] : 0.f;
            sum += va[h];
        }
        for (int m = 32; m >= 1; m >>= 1) sum += __shfl_xor(sum, m);
        const float v0 = live ? src[0] : 0.f, mean = sum / (float)A;
        float* dst = reinterpret_cast<float*>(pk + a.w3q_off) + (size_t)r * pw;
        float* dstT = reinterpret_cast<float*>(pk + a.w3q_off) + (size_t)(a.w3q_rows + 1) * pw;      // W3'^T [a][k] (the dense backward's gY2 = dq W3'^T)
#pragma unroll
        for (int h = 0; h < 2; ++h) {
            const int c = lane + 64 * h;
            const float o = (live && c < A) ? (v0 + va[h]) - mean : 0.f;
            if (c < pw) { dst[c] = o; if (r < a.w3q_rows) dstT[(size_t)c * a.w3q_rows + r] = o; }
        }
        return;
    }
    float v[8];
    if (blk_id >= e_d2t) {                                          // dense1t (gX): B(n1 = 32 blk + 8kb + e, k' = 16 ct + j) = W1[row(k')][n1]
        const int tiles = a.d1k >> 4, b = blk_id - e_d2t, blk = b / tiles, ct = b - blk * tiles;
        int k = 16 * ct + j;
        if (a.perm_hw > 0) { const int pp = k / a.perm_c, c = k - pp * a.perm_c; k = c * a.perm_hw + pp; }
        const float* w = params + d1_off + (size_t)k * DENSE_HID + 32 * blk + 8 * kb;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[e];
    } else if (blk_id >= e_d2) {                                    // dense2t (gH1): B(n2 = 32 blk + 8kb + e, n1 = 64 (ct>>2) + 4j + (ct&3)) = W2[n1][n2]
        const int b = blk_id - e_d2, blk = b >> 5, ct = b & 31, n1 = 64 * (ct >> 2) + 4 * j + (ct & 3);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int n2 = 32 * blk + 8 * kb + e;
            v[e] = n2 < a.N2 ? params[a.d2_off + (size_t)n1 * a.N2 + n2] : 0.f;
        }
    } else if (blk_id >= e_d1) {                                    // dense2 (forward): B(k = 32 blk + 8kb + e, col = NT2 j + t) = W2[k][col]
        const int b = blk_id - e_d1, blk = b / a.NT2, t = b - blk * a.NT2, col = 16 * t + j;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = col < a.N2 ? params[a.d2_off + (size_t)(32 * blk + 8 * kb + e) * a.N2 + col] : 0.f;
    } else if (blk_id >= PK_TOTAL_BLOCKS) {                         // Dense(512): block (kblk, ct): B(k = 32 kblk + 8kb + e, col = 64 (ct>>2) + 4j + (ct&3))
        // (rows of the transposed tile (b, s) = ct & 3: unit = 64 wave + 32 b + 8 (i >> 2) + 4 s + (i & 3), fused.hip dense_chain_kernel)
        const int b = blk_id - PK_TOTAL_BLOCKS, kblk = b >> 5, ct = b & 31;
        const float* w = params + d1_off + 64 * (ct >> 2) + 32 * ((ct >> 1) & 1) + 8 * (j >> 2) + 4 * (ct & 1) + (j & 3);
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            int k = 32 * kblk + 8 * kb + e;                         // the input rows' own (NHWC) order ...
            if (a.perm_hw > 0) { const int pp = k / a.perm_c, c = k - pp * a.perm_c; k = c * a.perm_hw + pp; }      // ... -> Keras Flatten row
            v[e] = w[(size_t)k * DENSE_HID];
        }
    } else if (blk_id < 16 + 8) {                                          // forward: B(k, col = 2j + t) = W[k][col], k = 32 blk + 8kb + e
        const bool c2 = blk_id < 16;
        const int b = c2 ? blk_id : blk_id - 16, blk = b >> 1, t = b & 1;
        const float* w = params + (c2 ? w2_off : w3_off);
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[(size_t)(32 * blk + 8 * kb + e) * 32 + 2 * j + t];
    } else if (blk_id >= 48) {                                      // first convolution: B(k = 32 h + 8kb + e, col = 4j + t), 0 past K1
        const int b = blk_id - 48, h = b >> 2, t = b & 3;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            const int k = 32 * h + 8 * kb + e;
            v[e] = k < a.K1 ? params[a.w1_off + (size_t)k * 64 + 4 * j + t] : 0.f;
        }
    } else {                                                        // data gradient: B(n = 8kb + e, c) = W[tap][c][n], tile t lane j: c = 2j + t
        const bool c3 = blk_id < 32;
        const int b = c3 ? blk_id - 24 : blk_id - 32;               // conv3: [tap][t]; conv2: [half][tap][t]
        const int t = b & 1, tap = (b >> 1) & 3, half = b >> 3, cin = c3 ? 32 : 64;
        const float* w = params + (c3 ? w3_off : w2_off) + (size_t)(tap * cin + 32 * half + 2 * j + t) * 32 + 8 * kb;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[e] = w[e];
    }
    const F16x2 o = split_f16x2(f32x4{v[0], v[1], v[2], v[3]}, f32x4{v[4], v[5], v[6], v[7]});
    u32x4* dst = pk + (size_t)blk_id * PK_BLOCK + lane;
    dst[0] = o.h; dst[PK_LO] = o.l;
}

PackLayout fused_pack_layout(const dq_qnet* Q) {
    const int nc = Q->cfg.n_conv;
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    PackLayout P;
    P.NT2 = D2.nout <= 64 ? 4 : 8;
    P.KB2 = D2.nout <= 64 ? 2 : 4;                                  // gH1's K in blocks of 32, zero-padded (a compile-time count in the kernel)
    P.d1_blocks = (D1.nin >> 5) * 32; P.d2_blocks = 16 * P.NT2; P.d2t_blocks = P.KB2 * 32; P.d1t_blocks = 16 * (D1.nin >> 4);
    P.dense1 = PK_TOTAL_U32X4;
    P.dense2 = P.dense1 + (size_t)P.d1_blocks * PK_BLOCK;
    P.dense2t = P.dense2 + (size_t)P.d2_blocks * PK_BLOCK;
    P.dense1t = P.dense2t + (size_t)P.d2t_blocks * PK_BLOCK;
    P.w3q = P.dense1t + (size_t)P.d1t_blocks * PK_BLOCK;
    P.w3q_rows = Q->cfg.dueling ? (D2.nout <= 64 ? 64 : 128) : 0;  // 16 KG3 (dense_chain_kernel: KG3 = NT2 = 4 or 8)
    P.wc = P.w3q + ((size_t)(2 * P.w3q_rows + 1) * 16 * P.NT2 + 3) / 4;      // W3' rows, the bias row, then W3'^T [16 NT2][w3q_rows]
    // Wc (the dense backward's shortcut, fused_bwd.hip SHORT): up to 64 actions built wave by wave, wider ones through an LDS copy of the dueling kernel
    P.wc_rows = Q->cfg.dueling && (P.NT2 == 4 || (size_t)D2.nout * Q->L[nc + 2].nout * 4 <= 52 * 1024) ? Q->cfg.n_actions : 0;
    P.c1c = P.wc + (size_t)P.wc_rows * DENSE_HID / 4;
    P.b1p_rows = Q->L[0].rows <= 64 ? Q->L[0].rows : 0;              // (patch-word input: one word per pixel and lane, d <= 7)
    P.b1p = P.c1c + (P.b1p_rows ? 4 * PK_BLOCK : 0);
    P.c1w = P.b1p + (size_t)P.b1p_rows * 64 / 4;
    P.c1w_blocks = P.b1p_rows ? 4 : 0;                                // (filled when the patch input has K_data + 6 <= 32; the space is there either way)
    P.c2w = P.c1w + (size_t)P.c1w_blocks * PK_BLOCK;
    P.c2w_blocks = P.c1w_blocks ? 16 : 0;
    P.cdw = P.c2w + (size_t)P.c2w_blocks * PK_BLOCK;
    P.cdw_blocks = P.c1w_blocks ? 24 : 0;
    P.total = P.cdw + (size_t)P.cdw_blocks * PK_BLOCK;
    return P;
}
size_t fused_packed_w1t_u32x4(const dq_qnet* Q) { return fused_pack_layout(Q).total; }
size_t fused_packed_w2t_u32x4(const dq_qnet* Q) {
    const Layer& D1 = Q->L[Q->cfg.n_conv];
    return fused_packed_w1t_u32x4(Q) + ((size_t)D1.K * D1.N + 3) / 4;
}
size_t fused_packed_u32x4(const dq_qnet* Q) {
    const Layer& D2 = Q->L[Q->cfg.n_conv + 1];
    return fused_packed_w2t_u32x4(Q) + ((size_t)D2.K * D2.N + 3) / 4;
}

dq_status fused_pack_weights(const dq_qnet* Q, const float* params_dev, void* packed_dev, hipStream_t st) {
    DQ_REQUIRE(Q && params_dev && packed_dev, DQ_ERR_INVALID, "dq_qnet_pack: null argument");
    DQ_REQUIRE(fused_forward_supported(Q), DQ_ERR_UNSUPPORTED, "dq_qnet_pack: the fused chains do not cover this configuration");
    const int nc = Q->cfg.n_conv;
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    PackArgs a;
    memset(&a, 0, sizeof(a));
    a.params = params_dev; a.pk = static_cast<u32x4*>(packed_dev);
    a.w1_off = (int)Q->L[0].w_off; a.K1 = Q->L[0].K; a.w2_off = (int)Q->L[1].w_off; a.w3_off = (int)Q->L[2].w_off; a.d1_off = (int)D1.w_off; a.d1_blocks = (D1.nin >> 5) * 32;
    a.perm_hw = Q->flat_hw; a.perm_c = Q->flat_c;
    const PackLayout PL = fused_pack_layout(Q);
    a.d2_off = (int)D2.w_off; a.N2 = D2.nout; a.NT2 = PL.NT2; a.KB2 = PL.KB2; a.d2_blocks = PL.d2_blocks; a.d2t_blocks = PL.d2t_blocks;
    a.d1t_blocks = PL.d1t_blocks; a.d1k = D1.nin;
    a.w3q_off = (int)PL.w3q; a.w3q_rows = PL.w3q_rows; a.n_actions = Q->cfg.n_actions;
    if (Q->cfg.dueling) { const Layer& D3 = Q->L[nc + 2]; a.w3d_off = (int)D3.w_off; a.b3d_off = (int)D3.b_off; a.N3 = D3.nout; }
    a.wc_off = (int)PL.wc; a.wc_waves = PL.wc_rows && PL.NT2 == 4 ? DENSE_HID : 0;      // (more than 64 actions: workgroups of their own, below)
    a.ptab = Q->patch_depth && PL.b1p_rows ? Q->ptab : nullptr;
    a.c1c_off = (int)PL.c1c; a.b1p_off = (int)PL.b1p; a.b1p_rows = PL.b1p_rows; a.p_depth = Q->patch_depth; a.p_C = Q->L[0].cin; a.b1_off = (int)Q->L[0].b_off;
    a.c1w_off = (int)PL.c1w; a.p_kd = Q->patch_kd; a.c1w_blocks = a.ptab && Q->patch_kd + 6 <= 32 ? PL.c1w_blocks : 0;
    a.c2w_off = (int)PL.c2w; a.c2w_blocks = a.c1w_blocks ? PL.c2w_blocks : 0;
    a.cdw_off = (int)PL.cdw; a.cdw_blocks = a.c1w_blocks ? PL.cdw_blocks : 0;
    a.pack_wgs = (PK_TOTAL_BLOCKS + PL.d1_blocks + PL.d2_blocks + PL.d2t_blocks + PL.d1t_blocks + (PL.w3q_rows ? PL.w3q_rows + 1 : 0) + a.wc_waves +
                  (a.ptab ? 4 + PL.b1p_rows + a.c1w_blocks + a.c2w_blocks + a.cdw_blocks : 0) + 3) / 4;
    // (the f32 transposes W1T / W2T this kernel used to append are gone with their last reader: both data gradients read packed pieces)
    // more than 64 actions: Wc by workgroups of their own behind the others, W3 staged in LDS (pack_wide_wc_block)
    const bool wide_wc = PL.wc_rows && PL.NT2 == 8;
    const int wide_blocks = wide_wc ? (DENSE_HID + WCW_U - 1) / WCW_U : 0;
    a.wide_blocks = wide_blocks;
    a.range_flag = fused_range_flag(Q); a.n_params = (int)Q->n_params;
    pack_weights_kernel<<<a.pack_wgs + wide_blocks, 256, wide_wc ? (size_t)a.N2 * a.N3 * 4 + 4 * WCW_U * 128 * 4 : 0, st>>>(a);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

// ---------------------------------------------------------------------------------------------------------------
struct ConvPlan { int S, slot, off_mis, off_t1, off_a1, off_a2, off_fx, off_lut, KG1; size_t lds; };

// patch: the observations are patch words (dq_qnet_set_patch_input): rows of 4 * patch_stride bytes, a1 rows unpadded, a byte -> bits table behind a1
static bool plan_conv(const dq_qnet* Q, ConvPlan* P, bool patch = false) {
    if (Q->cfg.n_conv != 3) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    if (L1.cout != 64 || L1.K > 96) return false;
    if (L2.cin != 64 || L2.cout != 32 || L2.k != 2 || L2.s != 1) return false;
    if (L3.cin != 32 || L3.cout != 32 || L3.k != 2 || L3.s != 1) return false;
    if (patch && !Q->patch_depth) return false;
    P->KG1 = (L1.K + 15) / 16;
    if (P->KG1 < 3) P->KG1 = 3;
    const int in_bytes = Q->cfg.in_c * Q->cfg.in_h * Q->cfg.in_w;
    if (in_bytes >= (1 << 17)) return false;                        // (first row table: sample << 20 | swizzle << 17 | offset inside the observation)
    P->slot = patch ? 4 * Q->patch_stride : (in_bytes + 15 + 15 + 15) & ~15;      // the 16-byte-aligned window around an arbitrarily aligned row
    const int a1ps = patch ? 64 : A1_PS;
    for (int pass = 0; pass < 2; ++pass) {                          // prefer two workgroups per CU; else the largest S that fits
        const size_t budget = pass == 0 ? CONV_LDS_2PER_CU : CONV_LDS_MAX;
        for (int S = 8; S >= 1; S >>= 1) {
            // [observations | alignment offsets | patch origins] live until conv1 is done; conv2's output a2 (written after the
            // barrier behind conv1) overlays them.  a1, a2: two f16 piece planes each, rows of 64 + 8 / 32 + 8 halves.
            size_t off = up16((size_t)S * P->slot);
            const size_t mis = off; off += up16((size_t)S * 4);
            const size_t t1 = off; off += up16((size_t)(S * L1.rows + (patch ? 16 : 0)) * 4);      // (+ 16: the patch-word form pads the table to whole tiles)
            const size_t a2_bytes = up16((size_t)2 * S * L2.rows * 40 * 2);
            if (off < a2_bytes) off = a2_bytes;
            const size_t a1 = off; off += up16((size_t)2 * S * L1.rows * a1ps * 2);
            const size_t fx = off; off += CONV_SWZ ? up16((size_t)S * L1.rows + 16) : 0;      // (+ 16: the last tile's rows past the end are read, not used)
            const size_t lut = off; off += patch ? 4096 : 0;
            if (off <= budget && S * L1.rows <= CONV_ROWTAB) {      // (rows per workgroup: the row tables' capacity)
                P->S = S; P->off_mis = (int)mis; P->off_t1 = (int)t1; P->off_a1 = (int)a1; P->off_a2 = 0; P->off_fx = (int)fx; P->off_lut = (int)lut; P->lds = off;
                return true;
            }
        }
    }
    return false;
}

// Persistent conv chain (conv_chain_pkernel): [obs A | core | obs B | a1 planes (rows of 64 halves) | alignment offsets [2][16]]; a2 overlays the
// current group's observation buffer and the core.  Two workgroups per CU where that fits, else one.
struct ConvPlanP { int S, slot, off_t1, off_obs1, off_a2b, off_a1, off_mis, off_lut, per_cu; size_t lds; };
static bool plan_conv_persist(const dq_qnet* Q, ConvPlanP* P, bool patch = false) {
    ConvPlan base;
    if (!plan_conv(Q, &base, patch)) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1];
    for (int pass = 0; pass < 2; ++pass) {
        const size_t budget = pass == 0 ? CONV_LDS_2PER_CU : CONV_LDS_MAX;
        for (int S = 8; S >= 1; S >>= 1) {
            const size_t obs = up16((size_t)S * base.slot), t1b = up16((size_t)(S * L1.rows + (patch ? 16 : 0)) * 4), a2b = up16((size_t)2 * S * L2.rows * 40 * 2);
            size_t core = a2b > obs ? a2b - obs : 0;
            if (core < t1b) core = t1b;
            size_t off = 2 * obs + core;
            const size_t a1 = off; off += up16((size_t)2 * (CONV1_PIPE ? (S * L1.rows + 15) & ~15 : S * L1.rows) * 64 * 2);
            const size_t mis = off; off += 2 * 16 * 4;
            const size_t t1c = off; off += t1b;                     // the patch-origin table, constant over the groups: a place of its own (a2 overlays the core)
            const size_t lut = off; off += patch ? 4096 : 0;
            if (off <= budget && S * L1.rows <= CONV_ROWTAB) {
                P->S = S; P->slot = base.slot; P->off_t1 = (int)t1c; P->off_obs1 = (int)(obs + core); P->off_a2b = (int)(2 * obs + core - a2b);
                P->off_a1 = (int)a1; P->off_mis = (int)mis; P->off_lut = (int)lut; P->lds = off; P->per_cu = pass == 0 ? 2 : 1;
                return true;
            }
        }
    }
    return false;
}

// Row tables of the fused conv forward for this network (CONV_ROWTAB ints each; qnet.hip uploads them behind kofftab at creation):
//   [0] first convolution, row m = s * r1 + oy * ow1 + ox of a workgroup's S samples:  s << 20 | f << 17 | (oy * stride * W + ox * stride), f = 2 (ox & 3)
//       the chunk swizzle of the row's a1 planes (CONV_SWZ)
//   [1] / [2] second / third convolution: halves from the input image's start to row m's patch, ((s * ih + oy) * iw + ox) * row stride (a1: A1_PS, and
//       with CONV_SWZ ox & 3 in bits 24+; a2: 40)
//   [3] the second convolution's over a1 rows of 64 halves (conv_chain_pkernel)
bool fused_conv_row_tables(const dq_qnet* Q, int* tab) {
    ConvPlan P;
    if (!plan_conv(Q, &P)) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    memset(tab, 0, sizeof(int) * CONV_FWD_TABS * CONV_ROWTAB);
    // (every table is filled for the largest group, 8 samples: row m's entry does not depend on the group size, so the plans of both kernels and
    // both input forms read prefixes of the same tables)
    auto rows_of = [](const Layer& L) { return 8 * L.rows < CONV_ROWTAB ? 8 * L.rows : CONV_ROWTAB; };
    for (int m = 0; m < rows_of(L2); ++m) {                         // [3]: the second convolution's rows over UNPADDED a1 planes (persistent kernel; patch-word input)
        const int s = m / L2.rows, pix = m % L2.rows, oy = pix / L2.ow, ox = pix % L2.ow;
        tab[3 * CONV_ROWTAB + m] = ((s * L2.ih + oy) * L2.iw + ox) * 64;
    }
    for (int m = 0; m < rows_of(L1); ++m) {
        const int s = m / L1.rows, pix = m % L1.rows, oy = pix / L1.ow, ox = pix % L1.ow;
        tab[m] = s << 20 | (CONV_SWZ ? (2 * (ox & 3)) << 17 : 0) | (oy * L1.s * L1.iw + ox * L1.s);
    }
    const Layer* Ls[2] = {&L2, &L3};
    for (int l = 0; l < 2; ++l)
        for (int m = 0; m < rows_of(*Ls[l]); ++m) {
            const int s = m / Ls[l]->rows, pix = m % Ls[l]->rows, oy = pix / Ls[l]->ow, ox = pix % Ls[l]->ow;
            tab[(1 + l) * CONV_ROWTAB + m] = l == 0 ? ((s * L2.ih + oy) * L2.iw + ox) * A1_PS | (CONV_SWZ ? (ox & 3) << 24 : 0)
                                                    : ((s * L3.ih + oy) * L3.iw + ox) * (L3.cin + 8);
        }
    return true;
}

// ---- patch-word input (include/deepq_hip.h dq_qnet_set_patch_input) ---------------------------------------------------------------------
// Tables of qnet.h PT_*: which Keras rows of the first kernel the data bits multiply, which constant cells every pixel's patch holds
// (padding_syndrome's decoration, /root/reference/example_notebooks/Environments.py:284-298, on the `depth` syndrome planes; the action
// planes' other cells are 0, ENV:301-314), and the row tables of the two conv kernels over rows of `stride_words` words.
bool fused_patch_supported(const dq_qnet* Q, int depth) {
    ConvPlan P;
    const Layer& L1 = Q->L[0];
    if (!plan_conv(Q, &P)) return false;
    if (L1.k != 3 || L1.s != 2 || L1.ih != L1.iw || !(L1.ih & 1) || L1.oh * L1.ow > 64) return false;     // Conv2D(64, 3, strides=2) on (2d+1)^2 planes, d <= 7
    if (depth < 1 || depth >= L1.cin) return false;
    return 4 * depth + (L1.cin - depth) <= 32;
}
void fused_patch_tables(const dq_qnet* Q, int depth, int stride_words, int* tab) {
    const Layer& L1 = Q->L[0];
    const int C = L1.cin, layers = C - depth, d = L1.oh, n = L1.ih, kd = 4 * depth + layers;
    memset(tab, 0, sizeof(int) * PT_TOTAL);
    for (int k = 0; k < 32; ++k) tab[PT_KROW + k] = -1;
    for (int k = 0; k < 96; ++k) tab[PT_SRC + k] = -1;
    for (int j = 0; j < depth; ++j)
        for (int c = 0; c < 4; ++c) {                                // corner (dy, dx) = (c >> 1, c & 1) of the patch: kernel tap (2 dy, 2 dx)
            const int row = ((2 * (c >> 1)) * 3 + 2 * (c & 1)) * C + j;
            tab[PT_KROW + 4 * j + c] = row; tab[PT_SRC + row] = 4 * j + c;
        }
    for (int l = 0; l < layers; ++l) {                              // the centre tap (1, 1) of the action planes
        const int row = (1 * 3 + 1) * C + depth + l;
        tab[PT_KROW + 4 * depth + l] = row; tab[PT_SRC + row] = 4 * depth + l;
    }
    const int cpos[5] = {0 * 3 + 1, 2 * 3 + 1, 1 * 3 + 0, 1 * 3 + 2, 1 * 3 + 1};
    for (int c = 0; c < 5; ++c) {
        tab[PT_CPOS + c] = cpos[c];
        for (int j = 0; j < depth; ++j) tab[PT_SRC + cpos[c] * C + j] = kd + c;      // the constant cells' kernel rows: one image column per position, every syndrome plane
    }
    auto decoration = [&](int x, int y) {                           // ENV:284-298 (the cells that do not hold a syndrome bit)
        int v = 0;
        if ((x == 0 || x == n - 1) && (y & 1)) v = 1;
        if ((y == 0 || y == n - 1) && (x & 1)) v = 1;
        if ((x & 1) && (y & 1) && ((x + y) % 4 == 0)) v = 1;
        return v;
    };
    for (int oy = 0; oy < d; ++oy)
        for (int ox = 0; ox < d; ++ox) {
            int mask = 0;
            for (int c = 0; c < 5; ++c) mask |= decoration(2 * oy + cpos[c] / 3, 2 * ox + cpos[c] % 3) << c;
            tab[PT_CONST + oy * d + ox] = mask;
        }
    const int r1 = L1.rows, rows = 8 * r1 < CONV_ROWTAB ? 8 * r1 : CONV_ROWTAB;
    for (int m = 0; m < rows; ++m) {
        const int s = m / r1, pix = m % r1;
        tab[PT_FWD + m] = s << 20 | 4 * pix;
        tab[PT_BWD + m] = (s * stride_words + pix) | tab[PT_CONST + pix] << 16;
    }
    conv_wave_lane_table(Q, kd, tab + PT_CONST, tab + PT_WAVE);
    conv_bwd16_tables(Q, kd, stride_words, tab + PT_CONST, tab + PT_SRC, tab + PT_C16);
}

struct DensePlan { int ldx, ld2, ld3, off_x, off_h, off_part, off_y2, off_y3, NT2; size_t lds; };

// rt = row tiles of 16 samples per workgroup (1 or 2)
static bool plan_dense(const dq_qnet* Q, DensePlan* P, int rt = 1) {
    const int nc = Q->cfg.n_conv;
    if (Q->cfg.n_ff != 1) return false;
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    if (D1.nout != DENSE_HID || (D1.nin & 31) || (Q->flat_c & 3)) return false;
    const int N2 = D2.nout, N3 = Q->cfg.dueling ? Q->L[nc + 2].nout : 0;
    if (N2 > 128 || N3 > 128) return false;
    const int rows = 16 * rt;
    P->NT2 = N2 <= 64 ? 4 : 8;
    if (rt > 2 && P->NT2 != 4) return false;                        // (64 samples: the registers only fit four Dense(|A|) tiles; 32 with eight: the WIDE2 form)
    P->ldx = D1.nin + 4;
    P->ld2 = 16 * P->NT2 + 4;
    P->ld3 = 16 * ((N3 + 15) / 16) + 1;
    size_t off = 0;
    const size_t xb = up16((size_t)2 * rows * (D1.nin + 8) * 2), pb = up16((size_t)DENSE_WAVES * (rows < DENSE_PR ? rows : DENSE_PR) * (16 * P->NT2 + 4) * 4);   // f16 planes | partials of one pass (rows padded by 4 floats)
    P->off_x = P->off_part = (int)off; off += xb > pb ? xb : pb;   // the Dense(|A|) partials reuse the input image (dead by then)
    P->off_h = (int)off;                                            // (the hidden output stays in registers)
    P->off_y2 = (int)off; off += up16((size_t)rows * P->ld2 * 4);
    P->off_y3 = (int)off;                                           // (unused since the dueling layer is folded: Q leaves from the accumulators)
    P->lds = off;
    return off <= CONV_LDS_MAX;
}

bool fused_forward_supported(const dq_qnet* Q) {
    ConvPlan cp;
    DensePlan dp;
    return plan_conv(Q, &cp) && plan_dense(Q, &dp);
}

typedef void (*conv_kernel_t)(ConvChainArgs);
typedef void (*dense_kernel_t)(DenseChainArgs);

dq_status fused_forward_multi(dq_qnet* Q, int n_jobs, const dq_qnet_job* jobs, hipStream_t st) {
    ConvPlan cp;
    DensePlan dp;
    DQ_REQUIRE(n_jobs >= 1 && n_jobs <= FWD_MAX_JOBS, DQ_ERR_INVALID, "fused_forward: 1..%d jobs per launch", FWD_MAX_JOBS);
    // every job of a launch reads its observations in the same form: padded uint8 images, or patch words (dq_qnet_job.reserved bit 0)
    const bool patch = (jobs[0].reserved & 1u) != 0;
    for (int i = 1; i < n_jobs; ++i)
        DQ_REQUIRE(((jobs[i].reserved & 1u) != 0) == patch, DQ_ERR_INVALID, "dq_qnet_forward_multi: the jobs of one launch must all read uint8 images or all read patch words");
    DQ_REQUIRE(!patch || Q->patch_depth, DQ_ERR_STATE, "dq_qnet_forward_multi: patch-word input without dq_qnet_set_patch_input");
    DQ_REQUIRE(plan_conv(Q, &cp, patch) && plan_dense(Q, &dp), DQ_ERR_UNSUPPORTED, "fused_forward: configuration not covered");
    conv_kernel_t ck = patch ? conv_chain_kernel<0> : cp.KG1 == 3 ? conv_chain_kernel<3> : cp.KG1 == 4 ? conv_chain_kernel<4> : cp.KG1 == 5 ? conv_chain_kernel<5> : conv_chain_kernel<6>;
    // the persistent form (DQ_CONV_PERSIST=0 selects the one-group-per-workgroup kernel: A/B runs) when the launch has more groups than resident workgroups
    // (read per call: tests flip it.  0: never; 1 / unset: when the launch has more groups than resident workgroups; 2: always, with DQ_CONV_PERSIST_GRID
    // workgroups if that is set -- a small grid makes every workgroup walk many groups)
    ConvPlanP pp;
    const char* pe = getenv("DQ_CONV_PERSIST");
    const int persist_env = pe ? atoi(pe) : 1;
    const char* pg = getenv("DQ_CONV_PERSIST_GRID");
    const int persist_grid = persist_env == 2 && pg ? atoi(pg) : 0;
    const bool can_persist = persist_env != 0 && plan_conv_persist(Q, &pp, patch) && pp.S == cp.S;
    const conv_kernel_t pk = patch ? conv_chain_pkernel<0> : cp.KG1 == 3 ? conv_chain_pkernel<3> : cp.KG1 == 4 ? conv_chain_pkernel<4> : cp.KG1 == 5 ? conv_chain_pkernel<5> : conv_chain_pkernel<6>;
    static unsigned long long attr_devs = 0;                          // per device (common.h dq_device_bit)
    const unsigned long long dev_bit = dq_device_bit();
    if (!(attr_devs & dev_bit)) {
        const conv_kernel_t cks[5] = {conv_chain_kernel<3>, conv_chain_kernel<4>, conv_chain_kernel<5>, conv_chain_kernel<6>, conv_chain_kernel<0>};
        for (int i = 0; i < 5; ++i)
            DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(cks[i]), hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_MAX));
        const conv_kernel_t pks[5] = {conv_chain_pkernel<3>, conv_chain_pkernel<4>, conv_chain_pkernel<5>, conv_chain_pkernel<6>, conv_chain_pkernel<0>};
        for (int i = 0; i < 5; ++i)
            DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(pks[i]), hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_MAX));
        const dense_kernel_t dks[6] = {dense_chain_kernel<4, 4, 1>, dense_chain_kernel<8, 8, 1>, dense_chain_kernel<4, 4, 2>, dense_chain_kernel<4, 4, 4>,
                                       dense_chain_kernel<8, 8, 2>, dense_chain_kernel<4, 4, 2, true>};
        for (int i = 0; i < 6; ++i)
            DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(dks[i]), hipFuncAttributeMaxDynamicSharedMemorySize, CONV_LDS_MAX));
        attr_devs |= dev_bit;
    }
    const int nc = Q->cfg.n_conv;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    const Layer &D1 = Q->L[nc], &D2 = Q->L[nc + 1];
    ConvChainArgs ca;
    DenseChainArgs da;
    memset(&ca, 0, sizeof(ca));
    memset(&da, 0, sizeof(da));
    ca.n_jobs = da.n_jobs = n_jobs; ca.S = cp.S;
    ca.C = L1.cin; ca.H = L1.ih; ca.W = L1.iw; ca.k1 = L1.k; ca.st1 = L1.s; ca.K1 = L1.K;
    ca.oh1 = L1.oh; ca.ow1 = L1.ow; ca.oh2 = L2.oh; ca.ow2 = L2.ow; ca.oh3 = L3.oh; ca.ow3 = L3.ow;
    for (int l = 0; l < 3; ++l) { ca.w_off[l] = (int)Q->L[l].w_off; ca.b_off[l] = (int)Q->L[l].b_off; }
    ca.kofftab = Q->kofftab; ca.rowtab = Q->kofftab + 96; ca.rowtab0 = patch ? Q->ptab + PT_FWD : ca.rowtab;
    ca.slot = cp.slot; ca.off_mis = cp.off_mis; ca.off_t1 = cp.off_t1; ca.off_a1 = cp.off_a1; ca.off_a2 = cp.off_a2; ca.off_fx = cp.off_fx; ca.off_lut = cp.off_lut;
    const PackLayout PL = fused_pack_layout(Q);
    ca.pk_c1c = (int)PL.c1c; ca.pk_b1p = (int)PL.b1p;
    da.K1 = D1.nin; da.perm_hw = Q->flat_hw; da.perm_c = Q->flat_c; da.pk_dense1 = (int)PL.dense1; da.pk_dense2 = (int)PL.dense2; da.pk_w3q = (int)PL.w3q;
    da.N2 = D2.nout; da.N3 = Q->cfg.dueling ? Q->L[nc + 2].nout : 0; da.n_actions = Q->cfg.n_actions;
    for (int l = 0; l < Q->n_layers - nc; ++l) { da.w_off[l] = (int)Q->L[nc + l].w_off; da.b_off[l] = (int)Q->L[nc + l].b_off; }
    // two row tiles (32 samples) per dense workgroup -- half the weight stream per sample -- when one-tile workgroups would
    // outnumber the CUs anyway and the larger images fit in LDS
    int tiles16 = 0;
    for (int i = 0; i < n_jobs; ++i) tiles16 += (jobs[i].batch + 15) / 16;
    int n_cu = 256;
    {
        static int cached_cus = 0;
        if (!cached_cus) {
            int dev = 0;
            hipDeviceProp_t prop;
            if (hipGetDevice(&dev) == hipSuccess && hipGetDeviceProperties(&prop, dev) == hipSuccess) cached_cus = prop.multiProcessorCount;
            if (cached_cus <= 0) cached_cus = 256;
        }
        n_cu = cached_cus;
    }
    // Row tiles per workgroup by a cost model measured at c3 (us per round of n_cu workgroups, DQ_DENSE_RT overrides): the kernel is bound
    // by the weight stream, so a workgroup's time grows slowly with its rows and the bytes per sample fall with them -- but a launch
    // that does not fill the chip with the larger workgroups is better off with smaller ones.
    auto rounds = [&](int rt, int per_cu) { return ((tiles16 + rt - 1) / rt + per_cu * n_cu - 1) / (per_cu * n_cu); };
    const double cost[3] = {(tiles16 <= n_cu ? 11.0 : 15.0 * rounds(1, 2)), 14.5 * rounds(2, 1), 19.0 * rounds(4, 1)};
    int RT = 1;
    double best = cost[0];
    DensePlan dpx;
    if (cost[1] < best && plan_dense(Q, &dpx, 2)) { RT = 2; best = cost[1]; dp = dpx; }
    if (cost[2] < best && plan_dense(Q, &dpx, 4)) { RT = 4; best = cost[2]; dp = dpx; }
    {
        static int forced = -1;
        if (forced < 0) { const char* e = getenv("DQ_DENSE_RT"); forced = e ? atoi(e) : 0; }
        if ((forced == 1 || forced == 2 || forced == 4) && plan_dense(Q, &dpx, forced)) { RT = forced; dp = dpx; }
    }
    // The lean form (round 6; dense_chain_kernel<4, 4, 2, true>: <= 128 VGPRs, two workgroups per CU) -- built, bit-identical, REFUTED: 28.6 against 26.4 us for the
    // step's four forwards (the two co-resident workgroups run the SAME phase at the same time: they share the matrix pipe in the hidden layer and both wait in the
    // staging and reduction phases, while every CU streams the hidden layer's weights twice; NOTEBOOK.md round 6 section 7).  Off; DQ_DENSE_LEAN=1: where 32-row
    // workgroups outnumber the CUs, 2: wherever they fit (A/B runs, the tests' second form; read once per handle at dq_qnet_create)
    bool lean = false;
    {
        const int lean_mode = Q->dense_lean;
        DensePlan dl;
        if (lean_mode > 0 && dp.NT2 == 4 && plan_dense(Q, &dl, 2) && 2 * dl.lds <= CHAIN_LDS_MAX && ((tiles16 + 1) / 2 > n_cu || lean_mode == 2)) {
            lean = true; RT = 2; dp = dl;
        }
    }
    const int dense_rows = 16 * RT;
    const dense_kernel_t dk = lean ? dense_chain_kernel<4, 4, 2, true>
                            : dp.NT2 == 4 ? (RT == 4 ? dense_chain_kernel<4, 4, 4> : RT == 2 ? dense_chain_kernel<4, 4, 2> : dense_chain_kernel<4, 4, 1>)
                                          : (RT == 2 ? dense_chain_kernel<8, 8, 2> : dense_chain_kernel<8, 8, 1>);
    da.ldx = dp.ldx; da.ld2 = dp.ld2; da.ld3 = dp.ld3;
    da.off_x = dp.off_x; da.off_h = dp.off_h; da.off_part = dp.off_part; da.off_y2 = dp.off_y2; da.off_y3 = dp.off_y3;
    int conv_wgs = 0, dense_wgs = 0, n_train = 0;
    for (int i = 0; i < n_jobs; ++i) {
        const dq_qnet_job& jb = jobs[i];
        DQ_REQUIRE(jb.params_dev && jb.obs_dev && jb.q_dev, DQ_ERR_INVALID, "dq_qnet_forward: null argument (job %d)", i);
        DQ_REQUIRE(jb.batch >= 1 && jb.batch <= Q->cfg.max_batch, DQ_ERR_INVALID, "dq_qnet_forward: batch %d outside 1..%d", jb.batch, Q->cfg.max_batch);
        DQ_REQUIRE((reinterpret_cast<uintptr_t>(jb.params_dev) & 15) == 0, DQ_ERR_INVALID, "fused_forward: params_dev must be 16-byte aligned");
        DQ_REQUIRE(!patch || (reinterpret_cast<uintptr_t>(jb.obs_dev) & 15) == 0, DQ_ERR_INVALID, "fused_forward: patch-word rows must be 16-byte aligned");
        const int training = jb.training ? 1 : 0;
        n_train += training;
        DQ_REQUIRE(n_train <= 1, DQ_ERR_INVALID, "dq_qnet_forward_multi: at most one training job per launch");
        float* x = training ? Q->act[0][nc - 1] : Q->xinf[i];      // conv3 output: saved for backward / per-job scratch
        ConvJob& C = ca.job[i];
        const void* packed = jb.packed_dev;
        if (!packed) {                                              // the caller did not pack: do it here (one small launch per job)
            for (int k2 = 0; k2 < i && !packed; ++k2)
                if (jobs[k2].params_dev == jb.params_dev && !jobs[k2].packed_dev) packed = Q->pk_scratch[k2];     // same weights as an earlier job
            if (!packed) {
                const dq_status rc = fused_pack_weights(Q, jb.params_dev, Q->pk_scratch[i], st);
                if (rc != DQ_OK) return rc;
                packed = Q->pk_scratch[i];
            }
        }
        C.params = jb.params_dev; C.packed = static_cast<const u32x4*>(packed); C.obs = jb.obs_dev; C.index = jb.index_dev; C.index_off = jb.index_off;
        C.index_mod = jb.index_mod > 0 ? jb.index_mod : 0x7fffffff;
        // (a conv_wave_kernel training job whose backward is conv_bwd16_kernel does not save a1 -- 26 MB of writes at c3, read straight back --: that backward
        // recomputes it from the patch words, the same bits; qnet.h conv_bwd_a1)
        const bool wave_form = patch && conv_wave_supported(Q) && Q->conv_form == 0;
        const bool a1_saved = !(training && wave_form && Q->conv_bwd_a1 == 0 && conv_bwd16_applies(Q, jb.batch, patch));
        if (training) Q->last_a1_saved = a1_saved ? 1 : 0;
        C.batch = jb.batch; C.write_all = training ? (a1_saved ? 3 : 1) : 0; C.wg0 = conv_wgs; ca.wg_first[i] = conv_wgs; da.wg_first[i] = dense_wgs;
        C.act_out[0] = nullptr; C.act_out[1] = nullptr; C.act_out[2] = x;
        // conv_wave_kernel leaves the last convolution's output as piece planes: the training job's straight into the weight gradients' operand (qnet.h dq_plane 0),
        // an inference job's into its scratch buffer (the same bytes as the f32 rows it held)
        unsigned short* xpl = nullptr;
        if (wave_form && Q->x_planes) xpl = training ? dq_plane(Q, 0) : reinterpret_cast<unsigned short*>(Q->xinf[i]);
        C.x_pl = xpl; C.x_lo = (size_t)Q->cfg.max_batch * D1.nin;
        C.a1_pl = reinterpret_cast<unsigned short*>(Q->act[0][0]); C.a1_lo = (size_t)Q->cfg.max_batch * L1.rows * 64;
        C.a2_pl = reinterpret_cast<unsigned short*>(Q->act[0][1]); C.a2_lo = (size_t)Q->cfg.max_batch * L2.rows * 32;
        conv_wgs += (jb.batch + cp.S - 1) / cp.S;
        DenseJob& D = da.job[i];
        D.params = jb.params_dev; D.packed = static_cast<const u32x4*>(packed); D.x = x; D.xp = xpl; D.xp_lo = (size_t)Q->cfg.max_batch * D1.nin; D.batch = jb.batch; D.wg0 = dense_wgs;
        if (training && D1.dropout > 0.f) {
            D.keep_scale = (float)(1.0 / (1.0 - (double)D1.dropout));
            D.drop_T = dq_rate_threshold16((double)D1.dropout);
            // the draw this forward asks for; the bits the last backward drew ahead if they are exactly it (qnet.h keep_bits)
            const dq_qnet::DropTag want = {jb.seed[0], jb.seed[1], jb.sample_base, D.drop_T, jb.t, jb.batch, 1, (void*)st};
            const dq_qnet::DropTag& have = Q->kb_tag;
            static const bool ahead_on = !(getenv("DQ_DROP_AHEAD") && getenv("DQ_DROP_AHEAD")[0] == '0');
            if (ahead_on && Q->keep_bits && have.valid && have.seed0 == want.seed0 && have.seed1 == want.seed1 && have.sample_base == want.sample_base &&
                have.drop_T == want.drop_T && have.t == want.t && have.batch == want.batch && have.stream == want.stream)
                D.keep_bits = Q->keep_bits;
            Q->last_drop = want;
        } else if (training) {
            Q->last_drop.valid = 0;
        }
        D.seed0 = jb.seed[0]; D.seed1 = jb.seed[1]; D.sample_base = jb.sample_base; D.t = jb.t;
        if (training) {
            D.plane_rows = Q->cfg.max_batch; D.small_ld = dq_planes_small_ld(Q);
            D.x_pl = xpl ? nullptr : dq_plane(Q, 0); D.h1_pl = dq_plane(Q, 1); D.y2_pl = dq_plane(Q, 5);      // (x planes: written by conv_wave_kernel already)
            Q->last_train_batch = jb.batch; Q->last_train_fused = 1; Q->last_obs = jb.obs_dev; Q->last_index = jb.index_dev;
            Q->last_index_off = jb.index_off; Q->last_index_mod = jb.index_mod; Q->last_train_packed = packed; Q->last_patch = patch ? 1 : 0;
        }
        D.q_out = jb.q_dev;
        dense_wgs += (jb.batch + dense_rows - 1) / dense_rows;
    }
    for (int i = n_jobs; i < FWD_MAX_JOBS; ++i) ca.wg_first[i] = da.wg_first[i] = 0x7fffffff;
    ca.range_flag = da.range_flag = fused_range_flag(Q);           // the forward's range guard (qnet.h range_report)
    // the wave-private form (conv_wave.hip) for patch words at d = 5; conv_form 1 (dq_qnet_set_kernel_forms, DQ_CONV_FORM=group at creation) keeps the
    // workgroup-per-group kernels
    if (patch && conv_wave_supported(Q) && Q->conv_form == 0) {
        ConvWaveArgs wa;
        memset(&wa, 0, sizeof(wa));
        for (int i = 0; i < n_jobs; ++i) wa.job[i] = ca.job[i];
        wa.n_jobs = n_jobs;
        for (int l = 0; l < 3; ++l) wa.b_off[l] = ca.b_off[l];
        wa.slot = cp.slot; wa.pk_c1w = (int)PL.c1w; wa.pk_c2w = (int)PL.c2w; wa.kd = Q->patch_kd; wa.ptab = Q->ptab; wa.range_flag = fused_range_flag(Q);
        const dq_status rc = conv_wave_launch(Q, wa, n_cu, st);
        if (rc != DQ_OK) return rc;
    } else if (can_persist && (conv_wgs > pp.per_cu * n_cu || persist_env == 2)) {
        ca.total_groups = conv_wgs; ca.off_t1 = pp.off_t1; ca.off_obs1 = pp.off_obs1; ca.off_a2b = pp.off_a2b; ca.off_a1 = pp.off_a1; ca.off_mis = pp.off_mis;
        ca.off_lut = pp.off_lut;
        int grid = pp.per_cu * n_cu;
        if (persist_grid > 0) grid = persist_grid;
        if (grid > conv_wgs) grid = conv_wgs;
        dq_launch(DQ_K_CONV_CHAIN, "conv_chain_pkernel", pk, dim3(grid), dim3(CONV_THREADS), pp.lds, st, ca);
    } else {
        dq_launch(DQ_K_CONV_CHAIN, "conv_chain_kernel", ck, dim3(conv_wgs), dim3(CONV_THREADS), cp.lds, st, ca);
    }
    DQ_LAUNCH_CHECK();
    dq_launch(DQ_K_DENSE_CHAIN, "dense_chain_kernel", dk, dim3(dense_wgs), dim3(DENSE_THREADS), dp.lds, st, da);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}
