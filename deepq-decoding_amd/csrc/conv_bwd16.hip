// The convolutional backward in workgroups of SIXTEEN waves (round 5): the same arithmetic as fused_bwd.hip's conv_bwd_chain_kernel<2, true> -- data AND weight
// gradients of Conv2D(64, 3, strides=2) - Conv2D(32, 2) - Conv2D(32, 2) (/root/reference/example_notebooks/Function_Library.py:352-365) on f16x2 pieces,
// patch-word observations, groups of 8 samples whose images live in LDS -- re-tiled so that a CU holds four waves per SIMD instead of two:
//
//   conv_bwd_chain_kernel   8 waves x 218 VGPRs: the data gradients' weights sit in registers (64 per wave and phase, streamed from L2 by every wave and
//                           group: 256 KB per group through the CU's 64 B/clk vector-memory path), a wave owns 2 x 2 tiles of dW2 and row tiles of
//                           2 column tiles in g2 / g1 (13 tiles on 4 waves: 4, 3, 3, 3 -- and the two 4-tile waves share a SIMD).  Counters: MFMA busy 17 %,
//                           VALU + MFMA issue 28 K and LDS 24 K of the kernel's 66 K cycles per SIMD / CU, 47 % of the wave cycles parked.
//   conv_bwd16_kernel       16 waves x <= 128 VGPRs (111).  The data gradients' weights (qnet.h cdw: 48 KB, channel-tile order) are copied into LDS once per
//                           workgroup.  The data gradients run TRANSPOSED (first operand = weights, rows = 16 input channels; second = the gradient rows of
//                           16 pixels): a lane then holds four consecutive channels of ONE pixel -- mask and result are one 8-byte LDS access per piece
//                           (they were four 4-byte ones).  g2: 16 units (row tile, channel tile), one per wave, its four weight blocks fetched at the top
//                           of the phase.  g1: by ROW TILE -- the four taps' gradient rows (the reads that can conflict) are read once per pass and held,
//                           the weights stream from the LDS copy (lane-ordered blocks: conflict-free), two channel tiles at a time; the thirteenth row
//                           tile (8 of 16 rows) is split over four waves so that every SIMD carries the same MFMAs.  dW3 / dW2 are split by tile (a wave
//                           owns one 16 x 16 tile of dW3 and one 16 x 32 strip of dW2: 24 accumulator registers instead of 48), dW1 by tile and row-block
//                           parity (the two halves meet in LDS at the end; the first bias gradient is row 31 of dW1: an all-ones patch column).  The rows
//                           of a K block are dealt so that a transposing read touches eight rows of ONE parity: conflict-free G operands (mask_rows).
//                           a1 is single-buffered (its rows padded to 144 bytes: the 8-byte mask / result accesses of 16 consecutive pixels then fall into
//                           distinct banks) and requested behind the group's first barrier -- it is first needed two phases later.  Every LDS address that
//                           depends on a row number comes out of host-built tables (conv_bwd16_tables) copied by LDS-DMA with everything else.
//                           Whole groups only (minibatch a multiple of 8): the row counts are compile-time and every loop unrolls.
//
// LDS (157.8 KB): [cdw 48 KB | a1 / g1 planes 58 KB | a2 / g2 planes 20.2 KB | g3 planes 11.4 KB | patch image 7.8 KB | observations 1 KB | tables 10 KB].
// Steps, counters and what was tried and lost: NOTEBOOK.md Round 5 section 3.
#include <type_traits>
#include "conv_bwd.h"

#define C16_THREADS 1024
#define C16_WAVES 16
#define C16_S 8                         // samples per group
#define C16_R1 25
#define C16_R2 16
#define C16_R3 9
#define C16_OW1 5
#define C16_OW2 4
#define C16_OW3 3
#define A1S 72                          // halves per LDS row of the a1 planes: 64 + 8 of padding (9 LDS-DMA slots of 16 bytes, the last one's lane inactive)
#define C16_A1_CHUNKS ((C16_S * C16_R1 * 9 + 63) / 64)      // 1 KB LDS-DMA pieces per a1 plane (29)
#define C16_LA1 (C16_A1_CHUNKS * 512)   // halves from a1's h plane to its l plane
#define C16_LA2 ((C16_S * C16_R2 + 1) * PL32)
#define C16_LG3 ((C16_S * C16_R3 + 1) * PL32)
#define C16_KP 40                       // bytes per row of the patch image: 32 columns (K_data + 5 <= 31 used, column 31 = 1) + 8 of padding -- the sixteen rows a
                                        // 32-lane group of dW1's 8-bit transposing read touches then fall on distinct banks (32 bytes apart: rows r, r + 8 on the same)
// byte offsets
#define C16_OFF_W 0
#define C16_OFF_A1 (24 * 2048)
#define C16_OFF_A2 (C16_OFF_A1 + 2 * C16_LA1 * 2)
#define C16_OFF_G3 (C16_OFF_A2 + ((2 * C16_LA2 * 2 + 1023) & ~1023))
#define C16_OFF_COL (C16_OFF_G3 + ((2 * C16_LG3 * 2 + 1023) & ~1023))
#define C16_OFF_IN (C16_OFF_COL + C16_S * C16_R1 * C16_KP)
#define C16_OFF_TAB (C16_OFF_IN + C16_S * 128)                // the host-built tables (qnet.h PT_C16), copied as they are: 12 KB
#define C16_LDS (C16_OFF_TAB + 4 * PT_C16_INTS)
// table sections (ints into the blob)
#define TB_D1 0                         // [200] x 4: a1 pixel m -> {g2 row (halves from the image) under taps 0 | 1 << 16, taps 2 | 3 << 16, a1 row of m, 0}; the zero row for a tap outside
#define TB_D2 800                       // [128] x 4: a2 pixel m -> {g3 row under taps 0 | 1, 2 | 3, a2 row of m, 0}
#define TB_T2 1312                      // [128]: second-convolution output pixel m -> a1 row under it | g2 row m << 16
#define TB_T3 1440                      // [72]: third-convolution output pixel m -> a2 row under it | g3 row m << 16
#define TB_TP 1520                      // [200]: first-convolution output pixel m -> (s * stride_words + p) | (its constant-cell mask | the bit of column 31) << 16 (qnet.h PT_BWD)
#define TB_KO 1720                      // [96]: Keras row of the first kernel -> column of the patch image, -1: gradient 0 (qnet.h PT_SRC)
#define TB_LUT 1816                     // [256] x 2: byte -> its bits as eight bytes
static_assert(TB_LUT + 512 <= PT_C16_INTS, "table blob");
static_assert(C16_LDS <= CHAIN_LDS_MAX, "LDS budget");
static_assert(C16_OFF_TAB % 16 == 0 && C16_OFF_COL % 16 == 0, "alignment");

size_t conv_bwd16_lds() { return C16_LDS; }

// Development aid (build with -DC16_STAMPS, tools/probe/build_c16_stamps.sh): shader-cycle stamps of every wave of workgroup 9 at the phase boundaries,
// read back by tools/probe/c16_stamps.py through dq_dbg_read_c16
#ifdef C16_STAMPS
static __device__ unsigned long long c16_dbg[32 * 16];
extern "C" void dq_dbg_read_c16(unsigned long long* out) { (void)hipMemcpyFromSymbol(out, HIP_SYMBOL(c16_dbg), sizeof(c16_dbg)); }
#define C16_STAMP(i) do { if ((threadIdx.x & 63) == 0 && blockIdx.x == 9) c16_dbg[(i) * 16 + (threadIdx.x >> 6)] = __builtin_readcyclecounter(); } while (0)
#else
#define C16_STAMP(i) do { } while (0)
#endif

// RC (round 6): the training forward (conv_wave_kernel) did not save a1; this kernel recomputes it from the patch image it builds anyway (dW1's operand) with the
// forward's own instructions -- first operand = the c1w quarter (per-pixel bias folded into the K = 32 block), second = the pixel's 32 bits as f16 0 / 1, one MFMA per
// weight piece, f16x2_sum, ReLU, split -- so the a1 pieces, the ReLU masks and every gradient are bit-identical to the saved-plane form's, while the forward writes
// 26 MB less (c3, 4096 samples) and this kernel fetches 26 MB less: its largest start-up copy (58 of the first group's 150 KB) is gone.  52 units (13 row tiles x 4
// quarters) of 2 MFMAs per group over the 16 waves, in the g2 phase (a1 is first read two barriers later).
template <bool RC>
__global__ __launch_bounds__(C16_THREADS) void conv_bwd16_kernel(ConvBwdArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const u8* s_wb = smem + C16_OFF_W;                                              // cdw blocks: 2048 bytes each (h pieces, then l)
    unsigned short* s_a1 = reinterpret_cast<unsigned short*>(smem + C16_OFF_A1);   // a1, then g1 in place: piece planes [2][rows][A1S]
    unsigned short* s_a2 = reinterpret_cast<unsigned short*>(smem + C16_OFF_A2);   // a2, then g2 in place: [2][S r2 + 1][PL32]
    unsigned short* s_g3 = reinterpret_cast<unsigned short*>(smem + C16_OFF_G3);   // g3: [2][S r3 + 1][PL32]
    u8* s_col = smem + C16_OFF_COL;                                                // patch image [S r1][32] bytes
    u8* s_in = smem + C16_OFF_IN;                                                  // the group's observation rows (patch words)
    const int* tab = reinterpret_cast<const int*>(smem + C16_OFF_TAB);
    const int4* td1 = reinterpret_cast<const int4*>(tab + TB_D1);
    const int4* td2 = reinterpret_cast<const int4*>(tab + TB_D2);
    const int *t2 = tab + TB_T2, *t3 = tab + TB_T3, *tp = tab + TB_TP, *s_ko = tab + TB_KO;
    const uint2* s_lut = reinterpret_cast<const uint2*>(tab + TB_LUT);
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
    constexpr int S = C16_S, r1 = C16_R1, r2 = C16_R2, r3 = C16_R3;
    constexpr int zero2 = S * r2, zero3 = S * r3, LA1 = C16_LA1, LA2 = C16_LA2, LG3 = C16_LG3, KP = C16_KP;
    const int in_bytes = a.slot;

    // this wave's replay row of the workgroup's first group (waves 0 .. 7: one sample each), requested before anything else
    auto obs_row = [&](int g) {
        int row = g * S + (wave & 7);
        if (a.index) {
            const __attribute__((address_space(4))) int32_t* idx = (const __attribute__((address_space(4))) int32_t*)(uintptr_t)a.index;
            row = idx[row] + a.index_off;
            if (row >= a.index_mod) row -= a.index_mod;
        }
        return row;
    };
    C16_STAMP(0);
    const int row_first = __builtin_amdgcn_readfirstlane(obs_row((int)blockIdx.x));
    // ---- every image goes global -> LDS by LDS-DMA (lds_dma16, qnet.h) --------------------------------------------------------------------------------
    // (the lane number made opaque at every use inside the group loop: hipcc otherwise hoists each copy's per-lane address arithmetic out of the loop as
    // invariants -- two registers per copy instruction -- and spills them)
    auto opq = [](int x) { asm volatile("" : "+v"(x)); return x; };
    auto issue_a1 = [&](int g) {
        const int lane = opq(tid & 63);
        constexpr int rows = S * r1, slots = rows * 9, chunks = (slots + 63) >> 6;
        const int gb0 = g * S;
        for (int c = wave; c < 2 * chunks; c += C16_WAVES) {
            const int piece = c >= chunks ? 1 : 0, ch = c - piece * chunks;
            const int q = ch * 64 + lane, row = q / 9, part = q - row * 9;
            if (q < slots && part < 8)
                lds_dma16(a.a1p + piece * a.a1_lo + (size_t)(gb0 * r1 + row) * 64 + part * 8, lds_addr(s_a1 + piece * LA1 + ch * 512));
        }
    };
    auto issue_pl32 = [&](const unsigned short* src, size_t src_lo, int rows, unsigned short* dst, int dst_lo) {
        const int lane = opq(tid & 63);
        const int slots = rows * 5, chunks = (slots + 63) >> 6;
        for (int c = wave; c < 2 * chunks; c += C16_WAVES) {
            const int piece = c >= chunks ? 1 : 0, ch = c - piece * chunks;
            const int q = ch * 64 + lane, row = q / 5, part = q - row * 5;
            if (q < slots && part < 4)
                lds_dma16(src + piece * src_lo + (size_t)row * 32 + part * 8, lds_addr(dst + piece * dst_lo + ch * 512));
        }
    };
    auto issue_a2 = [&](int g, int rows) { issue_pl32(a.a2p + (size_t)g * S * r2 * 32, a.a2_lo, rows, s_a2, LA2); };
    auto issue_g3 = [&](int g, int rows) { issue_pl32(a.g3p + (size_t)g * S * r3 * 32, a.g3_lo, rows, s_g3, LG3); };
    auto issue_obs = [&](int g, int row) {
        const int lane = opq(tid & 63);
        if (wave < S && lane < (in_bytes >> 4)) lds_dma16(a.obs + (size_t)row * in_bytes + 16 * lane, lds_addr(s_in + wave * in_bytes));
    };
    // RC: this wave's quarter (wave & 3) of the first kernel, both pieces: ordinary loads, issued in front of every copy (a wave's memory operations retire in order)
    u32x4 w1h = {0u, 0u, 0u, 0u}, w1l = w1h;
    if constexpr (RC) {
        const u32x4* c1 = a.packed + a.pk_c1w + (wave & 3) * PK_BLOCK + lane;
        w1h = c1[0]; w1l = c1[PK_LO];
    }
    // Copies of the first group, in the order of their first use: tables, observations, g3, a2 (patch image, dW3) | the data gradients' weights (g2) | a1 (dW2):
    // a wave's copies land in order, so the group loop waits for the first set only before it starts (every CU requests its 150 KB at once: the whole
    // set took 9.5 K cycles to land)
    {
        const int g = blockIdx.x;
        constexpr int gns = S;
        const char* tsrc = reinterpret_cast<const char*>(a.tab16);
        for (int c = wave; c < 4 * PT_C16_INTS / 1024; c += C16_WAVES) lds_dma16(tsrc + c * 1024 + lane * 16, lds_addr(smem + C16_OFF_TAB + c * 1024));
        issue_obs(g, row_first);
        issue_g3(g, gns * r3);
        issue_a2(g, gns * r2);
        const char* wsrc = reinterpret_cast<const char*>(a.packed + a.pk_cdw);
        for (int c = wave; c < 48; c += C16_WAVES) lds_dma16(wsrc + c * 1024 + lane * 16, lds_addr(smem + C16_OFF_W + c * 1024));
        if constexpr (!RC) issue_a1(g);
    }
    if (tid < PL32) { s_a2[zero2 * PL32 + tid] = 0; s_a2[LA2 + zero2 * PL32 + tid] = 0; s_g3[zero3 * PL32 + tid] = 0; s_g3[LG3 + zero3 * PL32 + tid] = 0; }
    C16_STAMP(1);
    // ---- per-wave roles -----------------------------------------------------------------------------------------------------------------------------
    // dW3 [128 x 32]: wave w owns k-tile w >> 1 (tap (w >> 1) >> 1, channels 16 ((w >> 1) & 1) ..) x column tile w & 1
    // dW2 [256 x 32]: wave w owns k-tile w (tap w >> 2, channels 16 (w & 3) ..) x both column tiles
    // dW1 [32 x 64]:  wave w owns tile w & 7 (k-tile (w & 7) >> 2, column tile w & 3) on the row blocks of parity w >> 3
    // g2: unit (row tile w >> 1, channel tile w & 1);  g1: units (row tile (w >> 2) + 4 i, channel tile w & 3)
    const int kt3 = wave >> 1, nt3 = wave & 1;
    const int aoff3 = ((kt3 >> 2) * C16_OW2 + ((kt3 >> 1) & 1)) * PL32 + 16 * (kt3 & 1);
    const int aoff2 = ((wave >> 3) * C16_OW1 + ((wave >> 2) & 1)) * A1S + 16 * (wave & 3);
    f32x4 acc3 = {0.f, 0.f, 0.f, 0.f}, acc3l = acc3, acc2[2] = {acc3, acc3}, acc2l[2] = {acc3, acc3}, acc1 = acc3, acc1l = acc3;
    float bs3 = 0.f;                                                // this thread's share of g3's column tid & 31
    float bs2[2] = {0.f, 0.f};                                      // g2's columns 2 (tid & 15), + 1 (threads < 512): this thread's row class

    // Transposed data gradient of one unit: rows m = 16 T + j of the activation image `act` (row stride AS halves, l plane act_lo halves further), channels
    // 16 nt + 4 kq .. + 3:  act <- (sum over taps  W[tap]^T g[pixel - tap]) * [act > 0], in place; g = piece planes [rows][PL32] with an all-zero row `zero_row`.
    // dtab[m]: row of g at the pixel's own position | iy << 16 | ix << 24.
    auto dgrad_unit = [&](const F16x2 (&bw)[4], const unsigned short* g, int g_lo, unsigned short* act, int act_lo, int nt, const int4* dtab, int M, int T,
                          int j, int kq) {
        const int m = 16 * T + j;
        const int4 de = dtab[min(m, M - 1)];                        // the four taps' gradient rows and this pixel's own row: ready-made offsets (conv_bwd16_tables)
        f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, accx = acc0;
        unsigned short* pa = act + de.z + 16 * nt + 4 * kq;
        const uint2 mh = *reinterpret_cast<const uint2*>(pa), ml = *reinterpret_cast<const uint2*>(pa + act_lo);
#pragma unroll
        for (int tp = 0; tp < 2; ++tp) {                            // two taps' gradient rows in flight (all four: 32 more registers than the 128 allow)
            F16x2 gv[2];
            const u32 e = (u32)(tp ? de.y : de.x);
#pragma unroll
            for (int u = 0; u < 2; ++u) {
                const unsigned short* gp = g + (u ? e >> 16 : e & 0xffffu) + 8 * kq;
                gv[u].h = *reinterpret_cast<const u32x4*>(gp);
                gv[u].l = *reinterpret_cast<const u32x4*>(gp + g_lo);
            }
#pragma unroll
            for (int u = 0; u < 2; ++u) mma_f16x3(bw[2 * tp + u], gv[u], acc0, accx);
        }
        const u32 b0 = mh.x | ml.x, b1 = mh.y | ml.y;
        const bool in = m < M;
        const float v0 = (in && (b0 & 0x7fffu) != 0u) ? f16x2_sum(acc0[0], accx[0]) : 0.f;
        const float v1 = (in && (b0 & 0x7fff0000u) != 0u) ? f16x2_sum(acc0[1], accx[1]) : 0.f;
        const float v2 = (in && (b1 & 0x7fffu) != 0u) ? f16x2_sum(acc0[2], accx[2]) : 0.f;
        const float v3 = (in && (b1 & 0x7fff0000u) != 0u) ? f16x2_sum(acc0[3], accx[3]) : 0.f;
        u32 h0, l0, h1, l1;
        split_f16x2_pair(v0, v1, h0, l0);
        split_f16x2_pair(v2, v3, h1, l1);
        if (in) {
            *reinterpret_cast<uint2*>(pa) = uint2{h0, h1};
            *reinterpret_cast<uint2*>(pa + act_lo) = uint2{l0, l1};
        }
    };
    auto load_bw = [&](F16x2 (&bw)[4], int block0, int stride, int lane) {       // blocks block0 + tap * stride of the LDS copy
#pragma unroll
        for (int tap = 0; tap < 4; ++tap) {
            const u8* p = s_wb + (block0 + tap * stride) * 2048 + lane * 16;
            bw[tap].h = *reinterpret_cast<const u32x4*>(p);
            bw[tap].l = *reinterpret_cast<const u32x4*>(p + 1024);
        }
    };
    // The 32 rows of a K block are dealt to the lanes so that the EIGHT rows a 32-lane group of a transposing read touches (lane groups kq = 0, 1 / 2, 3; four
    // rows per read and lane group) have the same parity: element e of lane group kq is row m0 + (kq >> 1) + 8 (kq & 1) + 2 (e & 3) + 16 (e >> 2).  With rows
    // 80 bytes (a2 / g2, g3) or 144 bytes (a1 / g1) apart, eight same-parity rows put their 32-byte windows on eight disjoint quarters of the 64 banks; eight
    // CONSECUTIVE rows (the first assignment: 4 kq + e) overlapped pairwise -- every such read took twice its cycles (tools/probe/c16_lds_model.py).
    // (Any assignment is a permutation of the reduction index as long as both operands use it.)
    // mask_rows: halves of G's pieces whose rows lie past M cleared (element e = half e of the operand)
    auto mask_rows = [&](F16x2& G, int m0, int M, int kq) {
#pragma unroll
        for (int d = 0; d < 4; ++d) {
            const int rb = m0 + (kq >> 1) + 8 * (kq & 1);
            const u32 lo = rb + 2 * ((2 * d) & 3) + 16 * ((2 * d) >> 2) < M ? 0xffffu : 0u;
            const u32 hi = rb + 2 * ((2 * d + 1) & 3) + 16 * ((2 * d + 1) >> 2) < M ? 0xffff0000u : 0u;
            G.h[d] &= lo | hi; G.l[d] &= lo | hi;
        }
    };
    int sb = 2;
    (void)sb;
    for (int grp = blockIdx.x; grp < a.groups; grp += gridDim.x) {
        // (whole groups only: conv_bwd16_launch refuses a minibatch that is not a multiple of 8 -- the row counts are compile-time, the loops unroll)
        constexpr int ns = S, M1 = S * r1, M2 = S * r2, M3 = S * r3, ns_nxt = S;
        const int nxt = grp + (int)gridDim.x;
        const int row_nxt = nxt < a.groups ? obs_row(nxt) : 0;
        // (lane-derived values re-derived per group from an opaque lane number: as loop invariants they -- and every address formed from them -- would be
        // held in registers across all phases)
        const int lane = opq(tid & 63), j = lane & 15, kq = lane >> 4, cseg = 4 * (j & 3);
        const int rq = (kq >> 1) + 8 * (kq & 1) + 2 * (j >> 2);       // the row (inside a K block) this lane points at in a transposing read; the second read: + 16
        C16_STAMP(sb + 0);
        // this wave's a1 copies of THIS group (the last it issued; 29 one-KB pieces per plane of a full group): they may stay in flight until dW2 -- and, in
        // the first group, so may the weights' until g2.  (vmcnt is an immediate: the exact count or, for a partial group, 0.)
        const int na1 = (2 * C16_A1_CHUNKS - wave + C16_WAVES - 1) / C16_WAVES;
        if (grp == (int)blockIdx.x) {                               // first group: the weights' and a1's copies may stay in flight
            if constexpr (RC) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");      // (three 1 KB pieces of the weights per wave)
            else if (na1 == 4) asm volatile("s_waitcnt vmcnt(7)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(6)" ::: "memory");
        } else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");      // later groups: observations, g3, a2 (requested during the previous group)
        __syncthreads();                                            // ... and every wave has left the previous group's dW1: a1 / g1 is free
        if constexpr (!RC) { if (grp != (int)blockIdx.x) issue_a1(grp); }      // first needed by dW2, two phases on
        C16_STAMP(sb + 1);
        // ---- patch image: row m = the bits of pixel m's word (data), then of its constant mask, one byte each -------------------------------------------
        if (tid < M1 * 4) {
            const int m = tid >> 2, g = tid & 3;
            const int e = tp[m];
            const u64 bits = (u64)reinterpret_cast<const u32*>(s_in)[e & 0xffff] | (u64)(u32)(e >> 16) << a.kd;
            *reinterpret_cast<uint2*>(s_col + m * KP + 8 * g) = s_lut[(u32)(bits >> (8 * g)) & 0xffu];
        }
        // the third convolution's bias gradient = column sums of g3, from its pieces
        for (int row = tid >> 5; row < M3; row += C16_THREADS / 32) {
            const unsigned short* gp = s_g3 + row * PL32 + (tid & 31);
            bs3 += (float)__builtin_bit_cast(_Float16, gp[0]) + (float)__builtin_bit_cast(_Float16, gp[LG3]) * F16_LO_INV;
        }
        C16_STAMP(sb + 2);
        // ---- dW3 += im2col(a2)^T g3 (rows of a K block: see mask_rows) -----------------------------------------
        {
            u32 e3[2 * ((M3 + 31) / 32)];                           // (every block's table entries first: one LDS latency for the phase instead of one per block)
#pragma unroll
            for (int i = 0; i < 2 * ((M3 + 31) / 32); ++i) e3[i] = (u32)t3[min(16 * i + rq, M3 - 1)];      // a2 row under the pixel | g3 row << 16
#pragma unroll
            for (int m0 = 0; m0 < M3; m0 += 32) {
                const u32 ea = e3[m0 >> 4], eb = e3[(m0 >> 4) + 1];
                const F16x2 A = lds_tr8(s_a2 + (ea & 0xffffu) + aoff3 + cseg, s_a2 + (eb & 0xffffu) + aoff3 + cseg, LA2);
                F16x2 G = lds_tr8(s_g3 + (ea >> 16) + 16 * nt3 + cseg, s_g3 + (eb >> 16) + 16 * nt3 + cseg, LG3);
                if (m0 + 32 > M3) mask_rows(G, m0, M3, kq);
                mma_f16x3(A, G, acc3, acc3l);
            }
        }
        C16_STAMP(sb + 3);
        if constexpr (RC) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // (first group: the weights have landed)
        else if (na1 == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory"); // (first group: the weights have landed; a1 may stay in flight)
        else asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
        __syncthreads();                                            // every wave is done reading a2; the patch image is complete
        C16_STAMP(sb + 4);
        // ---- g2 = (g3 (*) W3^T) * [a2 > 0], in place over a2 ----------------------------------------------------------------------------------------------
        {
            F16x2 bw[4];
            load_bw(bw, nt3, 2, lane);
            if (16 * (wave >> 1) < M2) dgrad_unit(bw, s_g3, LG3, s_a2, LA2, nt3, td2, M2, wave >> 1, j, kq);
        }
        if constexpr (RC) {
            // ---- a1 = relu(conv1(patch words)): this wave's quarter of 16 channels on row tiles (wave >> 2) + 4 i -- conv_wave.hip's conv1, instruction for instruction
            //      (transposed: a lane ends up with channels 4 kq .. + 3 of pixel 16 T + j: one 8-byte store per piece into the a1 planes, as g1's in-place writes) ----
            const int q1 = wave & 3;
#pragma unroll
            for (int i = 0; i < 4; ++i) {
                const int T = (wave >> 2) + 4 * i;                  // wave-uniform
                if (T > 12) break;
                const int m = 16 * T + j;
                const uint2 v = *reinterpret_cast<const uint2*>(s_col + min(m, M1 - 1) * KP + 8 * kq);      // the pixel's bits 8 kq .. + 7 as bytes 0 / 1
                u32x4 bits;
                bits[0] = __umul24(__builtin_amdgcn_perm(0u, v.x, 0x0c010c00u), 0x3c00u);
                bits[1] = __umul24(__builtin_amdgcn_perm(0u, v.x, 0x0c030c02u), 0x3c00u);
                bits[2] = __umul24(__builtin_amdgcn_perm(0u, v.y, 0x0c010c00u), 0x3c00u);
                bits[3] = __umul24(__builtin_amdgcn_perm(0u, v.y, 0x0c030c02u), 0x3c00u);
                const f32x4 z = {0.f, 0.f, 0.f, 0.f};
                const f32x4 ah = MFMA_F16(w1h, bits, z), al = MFMA_F16(w1l, bits, z);
                const f32x4 vs = f16x2_sum(ah, al);
                uint2 hp, lp;
                split_f16x2_pair(relu1(vs[0]), relu1(vs[1]), hp.x, lp.x);
                split_f16x2_pair(relu1(vs[2]), relu1(vs[3]), hp.y, lp.y);
                unsigned short* pa = s_a1 + min(m, M1 - 1) * A1S + 16 * q1 + 4 * kq;
                if (m < M1) {
                    *reinterpret_cast<uint2*>(pa) = hp;
                    *reinterpret_cast<uint2*>(pa + LA1) = lp;
                }
            }
        }
        C16_STAMP(sb + 5);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // a1 has landed (RC: nothing is in flight; the barrier publishes the recomputed a1)
        __syncthreads();
        C16_STAMP(sb + 6);
        if (nxt < a.groups) {                                       // the observation rows and g3 are dead now
            issue_obs(nxt, row_nxt);
            issue_g3(nxt, ns_nxt * r3);
        }
        // the second convolution's bias gradient = column sums of g2, from its pieces: thread (column pair tid & 15, row class tid >> 4 < 32) adds its rows
        if (tid < 512) {
            for (int row = tid >> 4; row < M2; row += 32) {
                const unsigned short* gp = s_a2 + row * PL32 + 2 * (tid & 15);
                const f16x2 h = __builtin_bit_cast(f16x2, *reinterpret_cast<const u32*>(gp)), l = __builtin_bit_cast(f16x2, *reinterpret_cast<const u32*>(gp + LA2));
                bs2[0] += (float)h[0] + (float)l[0] * F16_LO_INV; bs2[1] += (float)h[1] + (float)l[1] * F16_LO_INV;
            }
        }
        // ---- dW2 += im2col(a1)^T g2 -------------------------------------------------------------------------------------------------------------------------
        {
            u32 e2[2 * (M2 / 32)];
#pragma unroll
            for (int i = 0; i < 2 * (M2 / 32); ++i) e2[i] = (u32)t2[16 * i + rq];      // a1 row under the pixel | g2 row << 16
#pragma unroll
            for (int m0 = 0; m0 < M2; m0 += 32) {
                const u32 ea = e2[m0 >> 4], eb = e2[(m0 >> 4) + 1];
                const F16x2 A = lds_tr8(s_a1 + (ea & 0xffffu) + aoff2 + cseg, s_a1 + (eb & 0xffffu) + aoff2 + cseg, LA1);
                const unsigned short* p0 = s_a2 + (ea >> 16) + cseg;
                const unsigned short* p1 = s_a2 + (eb >> 16) + cseg;
                F16x2 G0 = lds_tr8(p0, p1, LA2), G1 = lds_tr8(p0 + 16, p1 + 16, LA2);
                mma_f16x3(A, G0, acc2[0], acc2l[0]);
                mma_f16x3(A, G1, acc2[1], acc2l[1]);
            }
        }
        C16_STAMP(sb + 7);
        __syncthreads();                                            // every wave is done reading a1
        C16_STAMP(sb + 8);
        // ---- g1 = (g2 (*) W2^T) * [a1 > 0], in place over a1 ----------------------------------------------------------------------------------------------
        // Wave w < 13 takes row tile w with all four channel tiles: the four taps' gradient rows (the reads that can conflict) are fetched ONCE and held,
        // the weights of a channel tile stream from the LDS copy (lane-ordered blocks: conflict-free).  As (row tile, channel tile) units -- every unit
        // fetching its gradient rows again -- the phase was LDS-bound: 5.2 K of its 5.6 K cycles, 3.6 K of them those reads at 2.2 cycles per ideal one.
        // Row tiles 0 .. 11: one wave each, all four channel tiles.  The thirteenth (8 of its 16 rows exist): waves 12 .. 15, one channel tile each -- every
        // SIMD then carries 3 x 48 + 12 MFMAs (with the whole tile on wave 12 its SIMD carried 4 x 48 and set the phase: 6.3 K cycles against the others' 4.7 K).
        {
            const int T = wave < 12 ? wave : 12;
            const int m = 16 * T + j;
            const int4 de = td1[min(m, M1 - 1)];                    // the four taps' g2 rows and this pixel's own a1 row: ready-made offsets (conv_bwd16_tables)
            const bool in = m < M1;
            auto tiles = [&](int nt0, auto two) {
                constexpr int NT = decltype(two)::value;
                f32x4 acc0[NT], accx[NT];
#pragma unroll
                for (int u = 0; u < NT; ++u) { acc0[u] = f32x4{0.f, 0.f, 0.f, 0.f}; accx[u] = acc0[u]; }
#pragma unroll
                for (int tap = 0; tap < 4; ++tap) {
                    const u32 e = (u32)(tap & 2 ? de.y : de.x);
                    const unsigned short* gp = s_a2 + (tap & 1 ? e >> 16 : e & 0xffffu) + 8 * kq;
                    F16x2 gv, bw[NT];
                    gv.h = *reinterpret_cast<const u32x4*>(gp);
                    gv.l = *reinterpret_cast<const u32x4*>(gp + LA2);
#pragma unroll
                    for (int u = 0; u < NT; ++u) {
                        const u8* p = s_wb + (8 + tap * 4 + nt0 + u) * 2048 + lane * 16;
                        bw[u].h = *reinterpret_cast<const u32x4*>(p);
                        bw[u].l = *reinterpret_cast<const u32x4*>(p + 1024);
                    }
#pragma unroll
                    for (int u = 0; u < NT; ++u) mma_f16x3(bw[u], gv, acc0[u], accx[u]);
                }
#pragma unroll
                for (int u = 0; u < NT; ++u) {
                    unsigned short* pa = s_a1 + de.z + 16 * (nt0 + u) + 4 * kq;
                    const uint2 mh = *reinterpret_cast<const uint2*>(pa), ml = *reinterpret_cast<const uint2*>(pa + LA1);
                    const u32 b0 = mh.x | ml.x, b1 = mh.y | ml.y;
                    const float v0 = (in && (b0 & 0x7fffu) != 0u) ? f16x2_sum(acc0[u][0], accx[u][0]) : 0.f;
                    const float v1 = (in && (b0 & 0x7fff0000u) != 0u) ? f16x2_sum(acc0[u][1], accx[u][1]) : 0.f;
                    const float v2 = (in && (b1 & 0x7fffu) != 0u) ? f16x2_sum(acc0[u][2], accx[u][2]) : 0.f;
                    const float v3 = (in && (b1 & 0x7fff0000u) != 0u) ? f16x2_sum(acc0[u][3], accx[u][3]) : 0.f;
                    u32 h0, l0, h1, l1;
                    split_f16x2_pair(v0, v1, h0, l0);
                    split_f16x2_pair(v2, v3, h1, l1);
                    if (in) {
                        *reinterpret_cast<uint2*>(pa) = uint2{h0, h1};
                        *reinterpret_cast<uint2*>(pa + LA1) = uint2{l0, l1};
                    }
                }
            };
            // (two channel tiles at a time: one accumulator pair per pass made hipcc emit read, wait, MFMA, wait, ... -- an LDS latency in front of every pair
            // of MFMAs --, all four at once spills)
            if (wave < 12) { tiles(0, std::integral_constant<int, 2>{}); tiles(2, std::integral_constant<int, 2>{}); }
            else tiles(wave - 12, std::integral_constant<int, 1>{});
        }
        C16_STAMP(sb + 9);
        __syncthreads();
        C16_STAMP(sb + 10);
        if (nxt < a.groups) issue_a2(nxt, ns_nxt * r2);            // g2 (in a2) is dead
        // (the first convolution's bias gradient comes out of dW1: column 31 of the patch image is 1 for every pixel -- conv_bwd16_tables -- so row 31 of
        // the product is g1's column sums)
        // ---- dW1 += patches^T g1: binary patch operand (one MFMA per piece of g1); rows as in dW2 / dW3 ------------------
        {
            const int w8 = wave & 7;
            const int cs = 16 * (w8 & 3) + 4 * (j & 3), rj = rq;
            const int re = j >> 1, rowb = (kq >> 1) + 8 * (kq & 1) + 2 * (re & 3) + 16 * (re >> 2);      // the patch row (inside a block) this lane points at
            const u8* cp = s_col + 16 * (w8 >> 2) + 8 * (j & 1);
            typedef int i32x2 __attribute__((ext_vector_type(2)));
            auto rd = [&](int m0, F16x2& G, i32x2& v) {                // (rows of g1 and of the patch image are where their numbers say: no table)
                G = lds_tr8(s_a1 + min(m0 + rj, M1 - 1) * A1S + cs, s_a1 + min(m0 + 16 + rj, M1 - 1) * A1S + cs, LA1);
                v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)(cp + min(m0 + rowb, M1 - 1) * KP));
            };
            auto mm = [&](int m0, F16x2& G, const i32x2& v) {
                if (m0 + 32 > M1) mask_rows(G, m0, M1, kq);
                u32x4 av;
                av[0] = __umul24(__builtin_amdgcn_perm(0u, (u32)v[0], 0x0c010c00u), 0x3c00u);
                av[1] = __umul24(__builtin_amdgcn_perm(0u, (u32)v[0], 0x0c030c02u), 0x3c00u);
                av[2] = __umul24(__builtin_amdgcn_perm(0u, (u32)v[1], 0x0c010c00u), 0x3c00u);
                av[3] = __umul24(__builtin_amdgcn_perm(0u, (u32)v[1], 0x0c030c02u), 0x3c00u);
                acc1 = MFMA_F16(av, G.h, acc1);
                acc1l = MFMA_F16(av, G.l, acc1l);
            };
            // two blocks' reads in flight (the seventh block belongs to the even half; the odd half's fourth trip re-reads rows of its third, masked to zero:
            // NO condition around an MFMA -- fused_bwd.hip)
            const int mh0 = 32 * (wave >> 3);
#pragma unroll
            for (int mb = 0; mb < 4; mb += 2) {
                F16x2 Ga, Gb;
                i32x2 va, vb;
                rd(mh0 + 64 * mb, Ga, va);
                rd(mh0 + 64 * mb + 64, Gb, vb);
                mm(mh0 + 64 * mb, Ga, va);
                mm(mh0 + 64 * mb + 64, Gb, vb);
            }
        }
        C16_STAMP(sb + 11);
        sb += 12;
    }
    C16_STAMP(26);

    // ---- one partial per workgroup ----------------------------------------------------------------------------------------------------------------------
    float* out = a.partial + (size_t)blockIdx.x * a.pstride;
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        out[a.w_off[2] + (16 * kt3 + 4 * kq + r) * 32 + 16 * nt3 + j] = f16x2_sum(acc3[r], acc3l[r]);
#pragma unroll
        for (int t = 0; t < 2; ++t) out[a.w_off[1] + (16 * wave + 4 * kq + r) * 32 + 16 * t + j] = f16x2_sum(acc2[t][r], acc2l[t][r]);
    }
    C16_STAMP(27);
    {
        // (these overlay the weights' LDS copy, last read in front of the g1 phase's closing barrier)
        float* s_b = reinterpret_cast<float*>(smem);                // [0, 1024) g3's classes | [1024, 2048) g2's
        float* s_res = s_b + 4096;                                  // the patch image's gradient [32][64], then scattered to the kernel's Keras rows
        const int w8 = wave & 7, kt = w8 >> 2, nt = w8 & 3;
        if (wave >= 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_res[(16 * kt + 4 * kq + r) * 64 + 16 * nt + j] = f16x2_sum(acc1[r], acc1l[r]);
        }
        s_b[tid] = bs3;                                             // thread (column tid & 31, row class tid >> 5)
        if (tid < 512) { s_b[1024 + (tid >> 4) * 32 + 2 * (tid & 15)] = bs2[0]; s_b[1024 + (tid >> 4) * 32 + 2 * (tid & 15) + 1] = bs2[1]; }
        __syncthreads();
        if (wave < 8) {
#pragma unroll
            for (int r = 0; r < 4; ++r) s_res[(16 * kt + 4 * kq + r) * 64 + 16 * nt + j] += f16x2_sum(acc1[r], acc1l[r]);
        }
        if (tid < 32) {
            float v = 0.f;
#pragma unroll
            for (int cl = 0; cl < 32; ++cl) v += s_b[1024 + cl * 32 + tid];
            out[a.b_off[1] + tid] = v;
        } else if (tid < 64) {
            float v = 0.f;
#pragma unroll
            for (int cl = 0; cl < 32; ++cl) v += s_b[cl * 32 + (tid - 32)];
            out[a.b_off[2] + tid - 32] = v;
        }
        __syncthreads();
        for (int i = tid; i < a.K1 * 64; i += C16_THREADS) {
            const int src = s_ko[i >> 6];
            out[a.w_off[0] + i] = src >= 0 ? s_res[src * 64 + (i & 63)] : 0.f;
        }
        if (tid < 64) out[a.b_off[0] + tid] = s_res[31 * 64 + tid];       // the all-ones column of the patch image: g1's column sums
    }
    C16_STAMP(28);
}

// The kernel's LDS-resident tables (qnet.h PT_C16; sections TB_* above), built once per dq_qnet_set_patch_input and copied by LDS-DMA as they are: every
// LDS address the kernel forms from a row number comes out of them, so the placement of the three images' rows is THEIR business (row r of a1 / g1 at
// A1S r halves, of a2 / g2 and g3 at PL32 r).
void conv_bwd16_tables(const dq_qnet* Q, int kd, int stride_words, const int* pt_const, const int* pt_src, int* out) {
    memset(out, 0, sizeof(int) * PT_C16_INTS);
    if (kd + 6 > 32 || Q->cfg.n_conv != 3 || Q->L[0].oh != C16_OW1 || Q->L[0].ow != C16_OW1 || Q->L[1].rows != C16_R2 || Q->L[2].rows != C16_R3) return;
    constexpr int S = C16_S;
    auto a1row = [](int r) { return r * A1S; };
    auto a2row = [](int r) { return r * PL32; };                     // (a2 / g2 image; row S r2 = the all-zero row)
    auto g3row = [](int r) { return r * PL32; };
    for (int m = 0; m < S * C16_R1; ++m) {
        const int s = m / C16_R1, p = m % C16_R1, iy = p / C16_OW1, ix = p % C16_OW1;
        int tapo[4];
        for (int tap = 0; tap < 4; ++tap) {
            const int oy = iy - (tap >> 1), ox = ix - (tap & 1);
            const bool valid = oy >= 0 && oy < C16_OW2 && ox >= 0 && ox < C16_OW2;
            tapo[tap] = a2row(valid ? s * C16_R2 + oy * C16_OW2 + ox : S * C16_R2);
        }
        out[TB_D1 + 4 * m] = tapo[0] | tapo[1] << 16; out[TB_D1 + 4 * m + 1] = tapo[2] | tapo[3] << 16; out[TB_D1 + 4 * m + 2] = a1row(m);
        // (+ column 31 of the patch image: constant 1 -> the bias gradient; + column K_data + 5: the forward's bias row of the c1w block -- a1 recomputed from this
        // image (RC) is conv_wave.hip's a1 bit for bit; no Keras row maps to that column, so dW1 does not see it, and c1w's row 31 is 0)
        out[TB_TP + m] = (s * stride_words + p) | (pt_const[p] | 1 << (31 - kd) | 1 << 5) << 16;
    }
    for (int m = 0; m < S * C16_R2; ++m) {
        const int s = m / C16_R2, p = m % C16_R2, iy = p / C16_OW2, ix = p % C16_OW2;
        int tapo[4];
        for (int tap = 0; tap < 4; ++tap) {
            const int oy = iy - (tap >> 1), ox = ix - (tap & 1);
            const bool valid = oy >= 0 && oy < C16_OW3 && ox >= 0 && ox < C16_OW3;
            tapo[tap] = g3row(valid ? s * C16_R3 + oy * C16_OW3 + ox : S * C16_R3);
        }
        out[TB_D2 + 4 * m] = tapo[0] | tapo[1] << 16; out[TB_D2 + 4 * m + 1] = tapo[2] | tapo[3] << 16; out[TB_D2 + 4 * m + 2] = a2row(m);
        out[TB_T2 + m] = a1row(s * C16_R1 + iy * C16_OW1 + ix) | a2row(m) << 16;
    }
    for (int m = 0; m < S * C16_R3; ++m) {
        const int s = m / C16_R3, p = m % C16_R3, iy = p / C16_OW3, ix = p % C16_OW3;
        out[TB_T3 + m] = a2row(s * C16_R2 + iy * C16_OW2 + ix) | g3row(m) << 16;
    }
    for (int k = 0; k < 96; ++k) out[TB_KO + k] = pt_src[k];
    for (unsigned b = 0; b < 256; ++b) {
        out[TB_LUT + 2 * b] = (int)((b & 1u) | (b & 2u) << 7 | (b & 4u) << 14 | (b & 8u) << 21);
        out[TB_LUT + 2 * b + 1] = (int)(((b >> 4) & 1u) | ((b >> 4) & 2u) << 7 | ((b >> 4) & 4u) << 14 | ((b >> 4) & 8u) << 21);
    }
}

// Patch-word input with K_data + 5 <= 32 columns, the three convolutions 64 x 3 x s2 / 32 x 2 / 32 x 2 on a 5 x 5 first output (d = 5): everything else
// takes conv_bwd_chain_kernel.
bool conv_bwd16_supported(const dq_qnet* Q) {
    if (Q->cfg.n_conv != 3 || !Q->patch_depth || Q->patch_kd + 5 > 32) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    if (L1.cout != 64 || L1.k != 3 || L1.s != 2 || L1.oh != 5 || L1.ow != 5) return false;
    if (L2.cin != 64 || L2.cout != 32 || L2.k != 2 || L2.s != 1 || L3.cin != 32 || L3.cout != 32 || L3.k != 2 || L3.s != 1) return false;
    if (4 * Q->patch_stride > 128 || L2.rows != C16_R2 || L3.rows != C16_R3) return false;
    return fused_pack_layout(Q).cdw_blocks == 24 && Q->patch_kd + 6 <= 32;
}

dq_status conv_bwd16_launch(const dq_qnet* Q, ConvBwdArgs& a, int wgs, hipStream_t st) {
    (void)Q;
    DQ_REQUIRE(a.batch % C16_S == 0 && a.S == C16_S, DQ_ERR_INVALID, "conv_bwd16_launch: whole groups of 8 samples only");
    static unsigned long long attr_devs = 0;
    const unsigned long long dev_bit = dq_device_bit();
    if (!(attr_devs & dev_bit)) {
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bwd16_kernel<false>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS));
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_bwd16_kernel<true>), hipFuncAttributeMaxDynamicSharedMemorySize, C16_LDS));
        attr_devs |= dev_bit;
    }
    if (a.a1_recompute) dq_launch(DQ_K_CONV_BWD, "conv_bwd16_kernel", conv_bwd16_kernel<true>, dim3(wgs), dim3(C16_THREADS), C16_LDS, st, a);
    else dq_launch(DQ_K_CONV_BWD, "conv_bwd16_kernel", conv_bwd16_kernel<false>, dim3(wgs), dim3(C16_THREADS), C16_LDS, st, a);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}
