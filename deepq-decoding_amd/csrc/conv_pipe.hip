// Convolutional forward chain as a persistent, weight-stationary wave pipeline (the same arithmetic as conv_chain_kernel in
// fused.hip: Keras model of build_convolutional_nn, /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:61-90).
//
// conv_chain_kernel re-reads every weight for every 8 samples and splits every activation into bf16 pieces once per tap that
// reads it; per workgroup the matrix pipe is busy a quarter of the time.  Here one workgroup per CU stays resident and walks over
// groups of S samples; its 8 waves are specialised, one pair per SIMD:
//
//   waves 0-3 ("A")   conv1 of group g+1  (first kernel STATIONARY in registers as bf16 pieces, observation bytes gathered from LDS)
//                     conv3 of group g-1  (third kernel's pieces in LDS, copied once per workgroup)
//   waves 4-7 ("B")   conv2 of group g    (second kernel STATIONARY in registers: 8 k-blocks x 2 column tiles x 3 pieces = 192 VGPRs)
//                     + the LDS-DMA of group g+2's observations
//
// with ONE barrier per step.  The layers run TRANSPOSED -- D[cout][pixel] = W^T[cout][k] im2col^T[k][pixel], i.e. the weight
// pieces are the MFMA's first operand and the activation its second -- so that a lane ends up with 8 or 16 CONSECUTIVE output
// channels of one pixel: one split_bf16x3 per 8 channels and three 16-byte LDS stores put the layer's output into three bf16
// planes [pixel][channel] ("split on write"), and the next layer's operand is three ds_read_b128 with no arithmetic at all.  An
// activation is split once, not once per tap and row tile.  a1 / a2 planes and the observations are double-buffered across steps.
#include <type_traits>

#include "qnet.h"

DQ_STAMP_READER(dq_dbg_read_pipe)

#define PIPE_THREADS 512
#define PIPE_PS1 144                    // bytes per pixel of an a1 plane: 64 bf16 + 16 bytes of padding
#define PIPE_PS2 80                     // a2 plane: 32 bf16 + 16
#define PIPE_MAX_S 16
#define PIPE_W3_BYTES (8 * PK_BLOCK * 16)

struct PipeArgs {
    ConvJob job[FWD_MAX_JOBS];          // job[i].wg0 = first workgroup of the job
    int nwg[FWD_MAX_JOBS];              // workgroups that share the job's groups (strided)
    int n_jobs, S;
    int C, H, W, k1, st1, K1;
    int oh1, ow1, oh2, ow2, oh3, ow3;
    int w_off[3], b_off[3];             // floats into params
    const int* kofftab;                 // [96] conv1 weight row k -> byte offset inside an observation, -1 past K1
    int slot;                           // bytes per observation slot in LDS
    int off_obs, obs_buf;               // observation buffer b at off_obs + b * obs_buf
    int off_mis;                        // int [3][PIPE_MAX_S]: misalignment of each staged row
    int off_a1, a1_plane;               // a1 buffer b, piece p at off_a1 + (3 b + p) * a1_plane
    int off_a2, a2_plane;               // B wave bw's private a2 planes (one sample), piece p at off_a2 + (3 bw + p) * a2_plane
    int off_w3, off_bias;               // conv3 pieces (PK_CONV3_FWD section, verbatim); float [128]: b1[64] b2[32] b3[32]
    int off_t1, off_t2, off_t3;         // int tables per output pixel of a group (see the kernel)
};

#define PIPE_FENCE() __builtin_amdgcn_sched_barrier(0)     // keeps "request the next operands" ahead of "multiply the current ones"
#define PIPE_RN (PIPE_MAX_S / 4)        // samples of a group per wave of a role

template <int NH1>                      // first convolution's K in blocks of 32 (K1 <= 32 * NH1)
__global__ __launch_bounds__(PIPE_THREADS, 2) void conv_pipe_kernel(PipeArgs a) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), j = lane & 15, kq = lane >> 4;
    int jb = 0;
    while (jb + 1 < a.n_jobs && (int)blockIdx.x >= a.job[jb + 1].wg0) ++jb;      // block-uniform
    const ConvJob& J = a.job[jb];
    const int wl = (int)blockIdx.x - J.wg0, nw = a.nwg[jb];
    const int S = a.S, r1 = a.oh1 * a.ow1, r2 = a.oh2 * a.ow2, r3 = a.oh3 * a.ow3;
    const int G = (J.batch + S - 1) / S;
    const int Gw = wl < G ? (G - wl + nw - 1) / nw : 0;             // this workgroup's groups: wl, wl + nw, ...
    const int in_bytes = a.C * a.H * a.W;
    int* s_mis = reinterpret_cast<int*>(smem + a.off_mis);
    int* t1 = reinterpret_cast<int*>(smem + a.off_t1);
    int* t2 = reinterpret_cast<int*>(smem + a.off_t2);
    int* t3 = reinterpret_cast<int*>(smem + a.off_t3);
    float* s_bias = reinterpret_cast<float*>(smem + a.off_bias);
    const u32x4* s_w3 = reinterpret_cast<const u32x4*>(smem + a.off_w3);
    typedef __attribute__((address_space(3))) u32* lds_u32;

    DQ_STAMP(DQ_TAG_CONV_PIPE, 0);
    // ---- the three kernels' bf16 pieces: global -> LDS ONCE per workgroup by LDS-DMA (1 KB = one piece of one packed block per wave
    //      instruction); the third kernel's stay there, the first two are staged in the (still unused) activation planes and go on into
    //      the registers of the waves that keep them.  (Every wave loading its own copy costs 4 x 72 KB per CU on a 64 B/clk path.)
    u8* stage_w2 = smem + a.off_a1;                                // 48 KB, then the first kernel's 12 * NH1 KB
    u8* stage_w1 = stage_w2 + 48 * 1024;
    {
        const int n3 = 24, n2 = 48, n1 = 12 * NH1;
        for (int ch = wave; ch < n3 + n2 + n1; ch += 8) {           // wave-uniform
            const u32x4* src;
            u8* dst;
            if (ch < n3) { src = J.packed + PK_CONV3_FWD + ch * 64; dst = smem + a.off_w3 + ch * 1024; }
            else if (ch < n3 + n2) { src = J.packed + PK_CONV2_FWD + (ch - n3) * 64; dst = stage_w2 + (ch - n3) * 1024; }
            else { src = J.packed + PK_CONV1 + (ch - n3 - n2) * 64; dst = stage_w1 + (ch - n3 - n2) * 1024; }
            __builtin_amdgcn_global_load_lds(src + lane, (lds_u32)dst, 16, 0, 0);
        }
    }
    // where the receptive field of an output pixel starts in the layer's input image: conv1 per pixel of a GROUP (observation bytes,
    // sample in the high half: its misalignment is per group), conv2 / conv3 per pixel of a SAMPLE (plane bytes)
    for (int m = tid; m < S * r1; m += PIPE_THREADS) {
        const int s = m / r1, p = m - s * r1, oy = p / a.ow1, ox = p - oy * a.ow1;
        t1[m] = (s << 16) | (s * a.slot + (oy * a.st1) * a.W + ox * a.st1);
    }
    for (int p = tid; p < r2; p += PIPE_THREADS) { const int oy = p / a.ow2, ox = p - oy * a.ow2; t2[p] = (oy * a.ow1 + ox) * PIPE_PS1; }
    for (int p = tid; p < r3; p += PIPE_THREADS) { const int oy = p / a.ow3, ox = p - oy * a.ow3; t3[p] = (oy * a.ow2 + ox) * PIPE_PS2; }
    if (tid < 128) s_bias[tid] = J.params[tid < 64 ? a.b_off[0] + tid : tid < 96 ? a.b_off[1] + tid - 64 : a.b_off[2] + tid - 96];

    if (wave < 4) {
        // =================================================== A waves ===================================================================
        // conv1 of group t + 1 in step t, and the LDS-DMA of the observations three groups ahead.
        // First kernel in registers, ready-made bf16 pieces (PK_CONV1).  The observation is binary, so three MFMAs per K = 32 block are
        // exact (fused.hip).  As the FIRST operand, lane (kb, i) supplies W1[k = 32h + 8kb + e][cout = 4i + c] for tile c: D row 4kq + r
        // of tile c is channel 16kq + 4r + c -- a lane's 16 results of a pixel are the 16 consecutive channels 16kq .. 16kq + 15.
        u32x4 wb[3][NH1][4];
        int ko[NH1][8];
#pragma unroll
        for (int h = 0; h < NH1; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) ko[h][e] = a.kofftab[32 * h + 8 * kq + e];     // -1 past K1: masked below (its weights are 0)
        // ---- observations by LDS-DMA: lane l copies aligned dword l of a 256-byte piece of a sample's arbitrarily aligned row -- whole
        //      aligned dwords, also where they straddle the neighbouring rows (fused.hip: the window stays inside the caller's allocation).
        //      Wave w stages samples w, w + 4, ... of a group, three groups ahead of the convolution that reads them (the copy has a whole
        //      step to land); their ring rows are looked up one step before that (a dependent global round trip).
        int vzero = 0;
        asm volatile("" : "+v"(vzero));
        // rows[]: the RAW index words (ring row before offset / wrap-around), untouched until the copy is issued one step later -- any
        // arithmetic on them here would put the load's latency back on this step
        auto fetch_rows = [&](int k, int (&rows)[PIPE_RN]) {
            const int b0 = (wl + k * nw) * S;
#pragma unroll
            for (int i = 0; i < PIPE_RN; ++i) {
                const int row = max(min(b0 + wave + 4 * i, J.batch - 1), 0);
                // a VECTOR load (lane-opaque zero offset): a scalar load shares its counter with the LDS reads, whose next wait would then
                // wait for this global round trip as well
                rows[i] = J.index ? J.index[row + vzero] : row;
            }
        };
        auto issue_obs = [&](int k, const int (&rows)[PIPE_RN]) {
            const int b0 = (wl + k * nw) * S, ns = min(S, J.batch - b0), buf = k % 3;
            const int pieces = (a.slot + 255) >> 8;
#pragma unroll
            for (int i = 0; i < PIPE_RN; ++i) {
                const int s = wave + 4 * i;
                if (s >= ns) break;
                int row = __builtin_amdgcn_readfirstlane(rows[i]);
                if (J.index) { row += J.index_off; if (row >= J.index_mod) row -= J.index_mod; }
                const u8* src = J.obs + (size_t)row * in_bytes;
                const int mis = (int)(reinterpret_cast<uintptr_t>(src) & 3);
                for (int pc = 0; pc < pieces; ++pc) {
                    const int d = pc * 64 + lane;
                    if (4 * d < mis + in_bytes)
                        __builtin_amdgcn_global_load_lds(reinterpret_cast<const u32*>(src - mis) + d,
                                                         (lds_u32)(smem + a.off_obs + buf * a.obs_buf + s * a.slot + pc * 256), 4, 0, 0);
                }
                if (lane == 0) s_mis[buf * PIPE_MAX_S + s] = mis;
            }
        };
        int rows[PIPE_RN], rows1[PIPE_RN], rows2[PIPE_RN], rows3[PIPE_RN];
        fetch_rows(0, rows);                                        // (group numbers past the end read clamped rows: harmless)
        fetch_rows(1, rows1);
        fetch_rows(2, rows2);
        fetch_rows(3, rows3);
        if (Gw > 0) issue_obs(0, rows);
        if (Gw > 1) issue_obs(1, rows1);
        if (Gw > 2) issue_obs(2, rows2);
#pragma unroll
        for (int i = 0; i < PIPE_RN; ++i) rows[i] = rows3[i];
#pragma unroll
        for (int h = 0; h < NH1; ++h)
#pragma unroll
            for (int e = 0; e < 8; ++e) ko[h][e] = max(ko[h][e], 0);
        DQ_STAMP(DQ_TAG_CONV_PIPE, 18);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's staged pieces and first observations have landed
        DQ_STAMP(DQ_TAG_CONV_PIPE, 17);
        __syncthreads();                                            // tables, biases, staged pieces, the first observations
#pragma unroll
        for (int h = 0; h < NH1; ++h)
#pragma unroll
            for (int c = 0; c < 4; ++c)
#pragma unroll
                for (int piece = 0; piece < 3; ++piece)
                    wb[piece][h][c] = *reinterpret_cast<const u32x4*>(stage_w1 + ((h * 4 + c) * 3 + piece) * 1024 + lane * 16);
        // this wave's first two tiles are the same pixels in every full group: their table entries, once
        const int tvS0 = t1[min(wave * 16 + j, S * r1 - 1)], tvS1 = t1[min((wave + 4) * 16 + j, S * r1 - 1)];
        f32x4 bias1[4];                                             // bias = the accumulators' initial value: channels 16kq + 4r + c
#pragma unroll
        for (int c = 0; c < 4; ++c)
#pragma unroll
            for (int r = 0; r < 4; ++r) bias1[c][r] = s_bias[16 * kq + 4 * r + c];
        DQ_STAMP(DQ_TAG_CONV_PIPE, 19);
        __syncthreads();                                            // the staging area becomes activation planes

        for (int t = -1; t <= Gw; ++t) {
            DQ_STAMP(DQ_TAG_CONV_PIPE, 1 + 3 * min(t + 1, 4));
            if (t + 1 < Gw) {
                // ---- conv1 of group t + 1: observation buffer (t+1) % 3 -> a1 planes, buffer (t+1) & 1.  Software pipeline over this wave's
                //      tiles: the bytes of the tile after next are requested, then the MFMAs of the next tile and the epilogue of the current
                //      one stand in ONE basic block (no branch: rows past the end reread the last row, so their lanes hold -- and store --
                //      exactly the last row's values again), where they overlap ---------------------------------------------------------
                const int k = t + 1, buf = k & 1, ob = k % 3;
                const int b0 = (wl + k * nw) * S, ns = min(S, J.batch - b0), M1 = ns * r1, tiles = (M1 + 15) >> 4;
                const bool full = ns == S;
                const u8* s_in = smem + a.off_obs + ob * a.obs_buf;
                const int* mis = s_mis + ob * PIPE_MAX_S;
                u8* p0 = smem + a.off_a1 + 3 * buf * a.a1_plane;
                float* g1 = J.act_out[0] + (size_t)b0 * r1 * 64;
                auto origin = [&](int tile) { return t1[min(tile * 16 + j, M1 - 1)]; };
                auto gather = [&](int tv, u32 (&ab)[NH1][8]) {
                    const u8* ap = s_in + (tv & 0xffff) + mis[tv >> 16];
#pragma unroll
                    for (int h = 0; h < NH1; ++h)
#pragma unroll
                        for (int e = 0; e < 8; ++e) ab[h][e] = ap[ko[h][e]];       // 0 or 1
                };
                auto mm = [&](const u32 (&ab)[NH1][8], f32x4 (&acc)[4]) {
#pragma unroll
                    for (int h = 0; h < NH1; ++h) {
                        u32x4 av;
#pragma unroll
                        for (int e = 0; e < 8; e += 2) av[e >> 1] = (ab[h][e] | (ab[h][e + 1] << 16)) * 0x3f80u;     // bf16(1.0) = 0x3f80
#pragma unroll
                        for (int piece = 0; piece < 3; ++piece)
#pragma unroll
                            for (int c = 0; c < 4; ++c) acc[c] = MFMA_BF16(wb[piece][h][c], av, (h == 0 && piece == 0) ? bias1[c] : acc[c]);
                    }
                };
                auto run = [&](auto train_tag) {
                    constexpr bool TRAIN = decltype(train_tag)::value;
                    auto ep = [&](int tile, const f32x4 (&acc)[4]) {
                        const int mo = min(tile * 16 + j, M1 - 1);  // D column = lane & 15 = this lane's pixel
                        f32x4 v[4];                                 // channels 16kq + 4q + c = acc[c][q]
#pragma unroll
                        for (int q = 0; q < 4; ++q)
#pragma unroll
                            for (int c = 0; c < 4; ++c) v[q][c] = fmaxf(acc[c][q], 0.f);
                        if constexpr (TRAIN) {
#pragma unroll
                            for (int q = 0; q < 4; ++q) *reinterpret_cast<f32x4*>(g1 + (size_t)mo * 64 + 16 * kq + 4 * q) = v[q];
                        }
                        u8* dst = p0 + mo * PIPE_PS1 + 32 * kq;
#pragma unroll
                        for (int hf = 0; hf < 2; ++hf) {
                            const Bf16x3 o = split_bf16x3(v[2 * hf], v[2 * hf + 1]);
                            *reinterpret_cast<u32x4*>(dst + 16 * hf) = o.h;
                            *reinterpret_cast<u32x4*>(dst + 16 * hf + a.a1_plane) = o.m;
                            *reinterpret_cast<u32x4*>(dst + 16 * hf + 2 * a.a1_plane) = o.l;
                        }
                    };
                    if constexpr (NH1 <= 2) {
                        u32 abA[NH1][8], abB[NH1][8];
                        f32x4 accA[4], accB[4];
                        // the first two tiles' table entries are the same in every full group; their gathers are issued together (two
                        // independent chains of LDS latencies instead of one after the other)
                        const int tv0 = full ? tvS0 : origin(wave), tv1 = full ? tvS1 : origin(wave + 4);
                        gather(tv0, abA); gather(tv1, abB);
                        mm(abA, accA);
                        for (int tile = wave;;) {
                            if (tile + 4 >= tiles) { ep(tile, accA); break; }
                            gather(origin(tile + 8), abA); mm(abB, accB); ep(tile, accA); tile += 4;
                            if (tile + 4 >= tiles) { ep(tile, accB); break; }
                            gather(origin(tile + 8), abB); mm(abA, accA); ep(tile, accB); tile += 4;
                        }
                    } else {                                        // (K1 > 64: 144 weight registers leave no room for the second buffers)
                        u32 ab[NH1][8];
                        f32x4 acc[4];
                        for (int tile = wave; tile < tiles; tile += 4) { gather(origin(tile), ab); mm(ab, acc); ep(tile, acc); }
                    }
                };
                if (wave < tiles) {
                    if (J.write_all) run(std::true_type{}); else run(std::false_type{});
                }
            }
            DQ_STAMP(DQ_TAG_CONV_PIPE, 2 + 3 * min(t + 1, 4));
            // the copy issued during the previous step (group t + 2: conv1 reads it in the next step) has landed by now, and so have the
            // index words fetched then; then the next copy is issued: group t + 3 into buffer (t+3) % 3, which conv1 of group t read during
            // step t - 1
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (t >= 0 && t + 3 < Gw) issue_obs(t + 3, rows);
            if (t >= 0 && t + 4 < Gw) fetch_rows(t + 4, rows);
            DQ_STAMP(DQ_TAG_CONV_PIPE, 3 + 3 * min(t + 1, 4));
            __syncthreads();
        }
    } else {
        // =================================================== B waves ===================================================================
        // conv2 and conv3 of group t in step t, a whole sample at a time: wave bw takes samples bw, bw + 4, ... -- conv2 reads the sample's a1
        // planes (written by the A waves in the previous step), writes its a2 planes into THIS wave's private region, conv3 reads them back
        // and writes the result to global memory: no other wave is involved, no barrier between the two layers.
        const int bw = wave - 4;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");            // this wave's staged pieces have landed
        DQ_STAMP(DQ_TAG_CONV_PIPE, 17);
        __syncthreads();
        // Second kernel -> registers: packed block (blk, c) as the FIRST operand: lane (kb, i) supplies W2[k = 32 blk + 8kb + e][cout = 2i + c],
        // so D row 4kq + r of tile c is channel 8kq + 2r + c: a lane's 8 results of a pixel are the 8 consecutive channels 8kq .. 8kq + 7.
        Bf16x3 w2[8][2];
#pragma unroll
        for (int blk = 0; blk < 8; ++blk)
#pragma unroll
            for (int c = 0; c < 2; ++c) {
                const u8* pb = stage_w2 + (blk * 2 + c) * 3072 + lane * 16;
                w2[blk][c].h = *reinterpret_cast<const u32x4*>(pb);
                w2[blk][c].m = *reinterpret_cast<const u32x4*>(pb + 1024);
                w2[blk][c].l = *reinterpret_cast<const u32x4*>(pb + 2048);
            }
        const int t2S = t2[min(j, r2 - 1)], t3S = t3[min(j, r3 - 1)];      // the first tile of a sample: the same pixels every time
        DQ_STAMP(DQ_TAG_CONV_PIPE, 19);
        __syncthreads();                                            // the staging area becomes activation planes
        u8* q0 = smem + a.off_a2 + bw * 3 * a.a2_plane;            // this wave's a2 planes: [3 pieces][r2 pixels][PIPE_PS2]

        for (int t = -1; t <= Gw; ++t) {
            DQ_STAMP(DQ_TAG_CONV_PIPE, 1 + 3 * min(t + 1, 4));
            if (t >= 0 && t < Gw) {
                const int b0 = (wl + t * nw) * S, ns = min(S, J.batch - b0);
                const u8* p0 = smem + a.off_a1 + 3 * (t & 1) * a.a1_plane;
                for (int s = bw; s < ns; s += 4) {
                    // ---- conv2: block blk + 1's activation pieces are requested before the MFMAs of block blk ------------------------------
                    float* g2 = J.write_all ? J.act_out[1] + (size_t)(b0 + s) * r2 * 32 : nullptr;
                    for (int tile = 0; tile * 16 < r2; ++tile) {
                        const int tv = tile == 0 ? t2S : t2[min(tile * 16 + j, r2 - 1)];
                        const u8* bp = p0 + s * r1 * PIPE_PS1 + tv + 16 * kq;
                        f32x4 acc[2][2];
                        Bf16x3 av[2];
                        auto ld = [&](int blk, Bf16x3& x) {
                            const int tap = blk >> 1;
                            const u8* p = bp + ((tap >> 1) * a.ow1 + (tap & 1)) * PIPE_PS1 + 64 * (blk & 1);
                            x.h = *reinterpret_cast<const u32x4*>(p);
                            x.m = *reinterpret_cast<const u32x4*>(p + a.a1_plane);
                            x.l = *reinterpret_cast<const u32x4*>(p + 2 * a.a1_plane);
                        };
                        ld(0, av[0]);
#pragma unroll
                        for (int c = 0; c < 2; ++c) {                // bias = the accumulators' initial value: channels 8kq + 2r + c
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[c][0][r] = s_bias[64 + 8 * kq + 2 * r + c];
                            acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int blk = 0; blk < 8; ++blk) {
                            if (blk < 7) ld(blk + 1, av[(blk + 1) & 1]);
                            PIPE_FENCE();
#pragma unroll
                            for (int c = 0; c < 2; ++c) mma_bf16x6(w2[blk][c], av[blk & 1], acc[c][0], acc[c][1]);
                            PIPE_FENCE();
                        }
                        const int mo = tile * 16 + j;
                        if (mo < r2) {
                            f32x4 va, vb;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                va[e] = fmaxf(acc[e & 1][0][e >> 1] + acc[e & 1][1][e >> 1], 0.f);
                                vb[e] = fmaxf(acc[e & 1][0][2 + (e >> 1)] + acc[e & 1][1][2 + (e >> 1)], 0.f);
                            }
                            if (g2) {
                                *reinterpret_cast<f32x4*>(g2 + (size_t)mo * 32 + 8 * kq) = va;
                                *reinterpret_cast<f32x4*>(g2 + (size_t)mo * 32 + 8 * kq + 4) = vb;
                            }
                            const Bf16x3 o = split_bf16x3(va, vb);
                            u8* dst = q0 + mo * PIPE_PS2 + 16 * kq;
                            *reinterpret_cast<u32x4*>(dst) = o.h;
                            *reinterpret_cast<u32x4*>(dst + a.a2_plane) = o.m;
                            *reinterpret_cast<u32x4*>(dst + 2 * a.a2_plane) = o.l;
                        }
                    }
                    // ---- conv3: lane (kb, n): 8 channels 8kb.. of pixel n's tap; weight pieces from LDS ---------------------------------------
                    float* out = J.act_out[2] + (size_t)(b0 + s) * r3 * 32;
                    for (int tile = 0; tile * 16 < r3; ++tile) {
                        const int tv = tile == 0 ? t3S : t3[min(tile * 16 + j, r3 - 1)];
                        const u8* bp = q0 + tv + 16 * kq;
                        f32x4 acc[2][2];
#pragma unroll
                        for (int c = 0; c < 2; ++c) {
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc[c][0][r] = s_bias[96 + 8 * kq + 2 * r + c];
                            acc[c][1] = f32x4{0.f, 0.f, 0.f, 0.f};
                        }
#pragma unroll
                        for (int tap = 0; tap < 4; ++tap) {
                            const u8* p = bp + ((tap >> 1) * a.ow2 + (tap & 1)) * PIPE_PS2;
                            Bf16x3 av;
                            av.h = *reinterpret_cast<const u32x4*>(p);
                            av.m = *reinterpret_cast<const u32x4*>(p + a.a2_plane);
                            av.l = *reinterpret_cast<const u32x4*>(p + 2 * a.a2_plane);
#pragma unroll
                            for (int c = 0; c < 2; ++c) {            // (fenced: 192 registers hold the second kernel; hoisted loads would spill)
                                const u32x4* wp = s_w3 + (tap * 2 + c) * PK_BLOCK + lane;
                                Bf16x3 w;
                                w.h = wp[0]; w.m = wp[64]; w.l = wp[128];
                                PIPE_FENCE();
                                mma_bf16x6(w, av, acc[c][0], acc[c][1]);
                                PIPE_FENCE();
                            }
                        }
                        const int mo = tile * 16 + j;
                        if (mo < r3) {                              // channels 8kq + 2r + c = acc[c][r]
                            f32x4 va, vb;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                va[e] = fmaxf(acc[e & 1][0][e >> 1] + acc[e & 1][1][e >> 1], 0.f);
                                vb[e] = fmaxf(acc[e & 1][0][2 + (e >> 1)] + acc[e & 1][1][2 + (e >> 1)], 0.f);
                            }
                            *reinterpret_cast<f32x4*>(out + (size_t)mo * 32 + 8 * kq) = va;
                            *reinterpret_cast<f32x4*>(out + (size_t)mo * 32 + 8 * kq + 4) = vb;
                        }
                    }
                }
            }
            DQ_STAMP(DQ_TAG_CONV_PIPE, 3 + 3 * min(t + 1, 4));
            __syncthreads();
        }
    }
    DQ_STAMP(DQ_TAG_CONV_PIPE, 16);
}

// ---------------------------------------------------------------------------------------------------------------
struct PipePlan { int S, NH1, slot, off_obs, obs_buf, off_mis, off_a1, a1_plane, off_a2, a2_plane, off_w3, off_bias, off_t1, off_t2, off_t3; size_t lds; };

static bool plan_pipe(const dq_qnet* Q, PipePlan* P) {
    if (Q->cfg.n_conv != 3) return false;
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    if (L1.cout != 64 || L1.K > 96) return false;
    if (L2.cin != 64 || L2.cout != 32 || L2.k != 2 || L2.s != 1) return false;
    if (L3.cin != 32 || L3.cout != 32 || L3.k != 2 || L3.s != 1) return false;
    P->NH1 = (L1.K + 31) / 32;
    if (P->NH1 < 2) P->NH1 = 2;
    const int in_bytes = Q->cfg.in_c * Q->cfg.in_h * Q->cfg.in_w;
    P->slot = (in_bytes + 3 + 3) & ~3;
    for (int S = PIPE_MAX_S; S >= 1; S >>= 1) {
        if ((size_t)S * P->slot >= 65536) continue;                  // t1 packs the byte offset into 16 bits
        size_t off = 0;
        P->off_obs = (int)off; P->obs_buf = (int)up16((size_t)S * P->slot + 256); off += 3 * (size_t)P->obs_buf;    // (+256: the last DMA piece of the last slot)
        P->off_mis = (int)off; off += 3 * PIPE_MAX_S * 4;
        off = (off + 1023) & ~(size_t)1023;                            // the plane area doubles as the LDS-DMA target of the weight staging
        P->off_a1 = (int)off; P->a1_plane = (int)up16((size_t)S * L1.rows * PIPE_PS1); off += 6 * (size_t)P->a1_plane;
        P->off_a2 = (int)off; P->a2_plane = (int)up16((size_t)L2.rows * PIPE_PS2); off += 12 * (size_t)P->a2_plane;      // one sample per B wave
        P->off_w3 = (int)off; off += PIPE_W3_BYTES;
        P->off_bias = (int)off; off += 128 * 4;
        P->off_t1 = (int)off; off += up16((size_t)S * L1.rows * 4);
        P->off_t2 = (int)off; off += up16((size_t)L2.rows * 4);
        P->off_t3 = (int)off; off += up16((size_t)L3.rows * 4);
        const size_t staging = (size_t)(48 + 12 * P->NH1) * 1024;       // second and first kernel's pieces, staged in the (contiguous) plane area
        if (off <= CHAIN_LDS_MAX && staging <= 6 * (size_t)P->a1_plane + 12 * (size_t)P->a2_plane) { P->S = S; P->lds = off; return true; }
    }
    return false;
}

bool conv_pipe_supported(const dq_qnet* Q) {
    PipePlan P;
    return plan_pipe(Q, &P);
}

typedef void (*pipe_kernel_t)(PipeArgs);

// jobs[i]: everything but wg0 filled in by the caller (fused_forward_multi)
dq_status conv_pipe_launch(dq_qnet* Q, int n_jobs, const ConvJob* jobs, int n_cu, hipStream_t st) {
    PipePlan P;
    DQ_REQUIRE(plan_pipe(Q, &P), DQ_ERR_UNSUPPORTED, "conv_pipe: configuration not covered");
    static bool attr_set = false;
    if (!attr_set) {
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pipe_kernel<2>), hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_MAX));
        DQ_HIP(hipFuncSetAttribute(reinterpret_cast<const void*>(conv_pipe_kernel<3>), hipFuncAttributeMaxDynamicSharedMemorySize, CHAIN_LDS_MAX));
        attr_set = true;
    }
    const Layer &L1 = Q->L[0], &L2 = Q->L[1], &L3 = Q->L[2];
    PipeArgs a;
    memset(&a, 0, sizeof(a));
    a.n_jobs = n_jobs; a.S = P.S;
    a.C = L1.cin; a.H = L1.ih; a.W = L1.iw; a.k1 = L1.k; a.st1 = L1.s; a.K1 = L1.K;
    a.oh1 = L1.oh; a.ow1 = L1.ow; a.oh2 = L2.oh; a.ow2 = L2.ow; a.oh3 = L3.oh; a.ow3 = L3.ow;
    for (int l = 0; l < 3; ++l) { a.w_off[l] = (int)Q->L[l].w_off; a.b_off[l] = (int)Q->L[l].b_off; }
    a.kofftab = Q->kofftab; a.slot = P.slot;
    a.off_obs = P.off_obs; a.obs_buf = P.obs_buf; a.off_mis = P.off_mis; a.off_a1 = P.off_a1; a.a1_plane = P.a1_plane;
    a.off_a2 = P.off_a2; a.a2_plane = P.a2_plane; a.off_w3 = P.off_w3; a.off_bias = P.off_bias;
    a.off_t1 = P.off_t1; a.off_t2 = P.off_t2; a.off_t3 = P.off_t3;
    // workgroups per job: in proportion to the jobs' groups, one per CU in total (every job at least one, none more than its groups)
    int groups[FWD_MAX_JOBS], total = 0;
    for (int i = 0; i < n_jobs; ++i) { groups[i] = (jobs[i].batch + P.S - 1) / P.S; total += groups[i]; }
    const int budget = total < n_cu ? total : n_cu;
    int wgs = 0;
    for (int i = 0; i < n_jobs; ++i) {
        int w = (int)((long long)budget * groups[i] / total);
        if (w < 1) w = 1;
        if (w > groups[i]) w = groups[i];
        a.job[i] = jobs[i];
        a.job[i].wg0 = wgs; a.nwg[i] = w;
        wgs += w;
    }
    pipe_kernel_t k = P.NH1 == 2 ? conv_pipe_kernel<2> : conv_pipe_kernel<3>;
    dq_prof_begin(DQ_K_CONV_CHAIN, st);
    k<<<wgs, PIPE_THREADS, P.lds, st>>>(a);
    dq_prof_end(DQ_K_CONV_CHAIN, st);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}
