// Q-network handle shared by qnet.hip (per-layer implicit GEMMs, backward) and fused.hip (fused forward chains).
#pragma once
#include "common.h"

#define QN_MAX_LAYERS 12
#define FWD_MAX_JOBS 4               // forwards served by one fused launch (dq_qnet_forward_multi)
#define CONV_ROWTAB 512              // rows per workgroup and layer of the fused conv forward, at most
#define CONV_FWD_TABS 4              // row tables of the fused conv forward (fused_conv_row_tables); the backward's five follow
// Patch-word input (dq_qnet_set_patch_input, include/deepq_hip.h dq_env_patch_output): device tables dq_qnet.ptab, in ints
#define PT_KROW 0                    // [32] compact k -> Keras row of the first kernel ((ky * 3 + kx) * C + plane), -1 past the data bits
#define PT_SRC 32                    // [96] Keras row -> column k' of the backward's patch image (data bits, then the 5 constant positions), -1: gradient 0
#define PT_CPOS 128                  // [8]  constant position c -> ky * 3 + kx   (c = 0 .. 4: (0,1) (2,1) (1,0) (1,2) (1,1))
#define PT_CONST 136                 // [64] pixel p -> 5-bit mask of its constant-1 positions on the syndrome planes
#define PT_FWD 200                   // [CONV_ROWTAB] forward: row m of a workgroup's S samples -> s << 20 | 4 p
#define PT_BWD (PT_FWD + CONV_ROWTAB)    // [CONV_ROWTAB] backward: row m -> (s * stride_words + p) | constant mask << 16
#define PT_WAVE (PT_BWD + CONV_ROWTAB)    // [64][PT_WAVE_LD] conv_wave.hip: the per-lane constants of its prologue (conv_wave_lane_table), 16-byte aligned rows
#define PT_WAVE_LD 20
#define PT_C16 (PT_WAVE + 64 * PT_WAVE_LD)    // [PT_C16_INTS] conv_bwd16.hip: its LDS-resident tables, ready to be copied as they are (conv_bwd16_tables), 16-byte aligned
#define PT_C16_INTS 2560
#define PT_TOTAL (PT_C16 + PT_C16_INTS)

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
    int last_train_fused;        // the last training forward ran on the fused chains (1) / per-layer kernels (0): the backward must take the same path --
                                 // each forward saves what ITS backward reads (the fused one keeps the hidden layer as piece planes only)
    const uint8_t* last_obs;     // inputs of the last training forward (needed by conv1's weight gradient)
    const int32_t* last_index;
    int last_index_off, last_index_mod;
    float* xinf[FWD_MAX_JOBS];   // fused inference forwards: last-convolution output per job slot [max_batch, flat]
    int* kofftab;                // [96] first convolution: weight row k -> byte offset inside an NCHW uint8 observation, -1 past K;
                                 // then [CONV_FWD_TABS][CONV_ROWTAB] the fused conv forward's row tables (fused_conv_row_tables) and [5][CONV_ROWTAB]
                                 // the fused conv backward's (fused_conv_bwd_row_tables)
    void* pk_scratch[FWD_MAX_JOBS];  // packed weights of jobs that did not bring their own (dq_qnet_job.packed_dev == NULL)
    const void* last_train_packed;   // packed weights of the last training forward (the backward's data gradients read them)
    float* fpartial;             // fused backward workspace (fused_backward_workspace_floats)
    unsigned short* planes;      // f16 piece planes (h plane, then l plane) of the training forward's / backward's operands of the dense
                                 // weight gradients, row-major [max_batch][ld]: x (ld K1), h1 (512), gh1 (512), then gy2, g3, y2 (ld
                                 // dq_planes_small_ld: 64 or 128; columns past the tensor's width hold anything finite or not -- they only
                                 // reach weight-gradient accumulators that are never stored), in this order
    // Dropout keep bits drawn AHEAD: the backward's final reduction (an HBM-bound launch whose vector ALUs idle) draws the hidden layer's keep bits
    // of the training forward it expects NEXT -- the last one's seed, sample range and rate at t + 1 -- into keep_bits ([max_batch][16] words: bit
    // unit & 31 of word unit >> 5 of a sample's row), and the next training forward that asks for exactly that (kb_tag) loads them instead of
    // running eight Philox calls per lane in its 64 training workgroups -- the workgroups that set the dense forward's duration.  Any other
    // request draws in the kernel as before: the same bits either way (fused.hip draw_keep_bits, fused_bwd.hip dropout_ahead).
    struct DropTag { u32 seed0, seed1, sample_base, drop_T; u64 t; int batch, valid; void* stream; };     // stream: the one the bits were drawn on -- a forward on
                                 // another stream is not ordered behind the drawing launch and draws in its own kernel
    u32* keep_bits;
    DropTag kb_tag;              // what keep_bits holds (valid = 1)
    DropTag last_drop;           // the last fused training forward's dropout draw (valid = 0: none)
    void* mark_event;            // dq_qnet_mark_conv_backward: hipEvent_t recorded behind the next fused convolutional backward's launch (one-shot)
    float grad_scale_hint;       // dq_qnet_set_grad_scale: loss scale of caller-supplied dq (0 = unknown: measured on the device)
    float bwd_scale;             // fused backward: power-of-two scale the gradients of the last dense phase carry (0: the device-computed one)
    int use_fused;               // fused LDS-resident chains when the configuration allows it
    // which FORM of a kernel family runs where several exist (dq_qnet_set_kernel_forms; initial value from DQ_CONV_FORM / DQ_CONV_BWD_FORM, read ONCE at
    // dq_qnet_create: a handle's summation order does not follow a mutable process environment)
    int conv_form;               // 0: conv_wave_kernel where it applies, 1: the workgroup-per-group kernels (fused.hip)
    int conv_bwd_form;           // 0: conv_bwd16_kernel where it applies and the minibatch is >= 1024, 1: always conv_bwd_chain_kernel, 2: conv_bwd16_kernel whatever the minibatch
    int conv_bwd_a1;             // 0: conv_bwd16_kernel RECOMPUTES the first convolution's output from the patch words where the training forward was conv_wave_kernel
                                 // (which then does not save it: round 6), 1: every training forward saves a1, every backward reads the saved planes
    int x_planes;                // 0 (default): the last convolution's output reaches the dense chain as f32 rows, split by its staging phase; 1 (DQ_X_PLANES=1 at
                                 // dq_qnet_create): conv_wave_kernel splits on write and the dense chain stages the piece planes by LDS-DMA (ConvJob.x_pl) -- round 6, built,
                                 // bit-identical, measured: dense forward -1.15 us, conv forward +0.7 us, the free-running step 0.5-1 % SLOWER (NOTEBOOK.md): off
    int dense_lean;              // DQ_DENSE_LEAN at dq_qnet_create: 0 (default) never, 1 the lean dense forward where 32-row workgroups outnumber the CUs, 2 wherever they fit
    int last_a1_saved;           // the last training forward wrote the a1 piece planes (the backward that recomputes a1 must follow a forward that did NOT, and
                                 // the other way round: dq_qnet_set_kernel_forms between the two is refused)
    // patch-word input (dq_qnet_set_patch_input): observations as d * d words per sample instead of the padded uint8 image
    int patch_depth;             // syndrome planes of the observation (0: not configured); the remaining input planes are action planes
    int patch_kd;                // data bits per pixel = 4 patch_depth + action planes (<= 32)
    int patch_stride;            // words per observation row
    int patch_cfg_depth, patch_cfg_stride;      // the first configuration ever set on this handle: packed buffers and tables made under it stay valid only for it
    int* ptab;                   // device tables (PT_*)
    int last_patch;              // the last training forward read patch words (the backward takes the same form)
    unsigned bwd_serial;         // fused backwards' dense phases launched so far: the tag of the range guard's early half (fused_bwd.hip skip_word)
};


// ---- shared by the fused chains (fused.hip forward, fused_bwd.hip backward) ---------------------------------------
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef u32 u32x4 __attribute__((ext_vector_type(4)));
typedef float f32x4u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned: global_load_dwordx4 accepts it
#define MFMA16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x4f32((a), (b), (c), 0, 0, 0)
#define CHAIN_LDS_MAX (160 * 1024)
#define DENSE_THREADS 512
#define DENSE_WAVES 8
#define DENSE_ROWS 16
#define DENSE_HID 512                 // Dense(512): 8 waves x 64 columns; Dense(|A|) splits K = 512 into 8 x 64
static inline size_t up16(size_t x) { return (x + 15) & ~(size_t)15; }

#ifdef __HIPCC__
// ---- f32-class contraction on the f16 matrix pipe ("f16x2") ---------------------------------------------------------------
// Every f32 operand x is carried as TWO f16 pieces: h = f16(x) (round to nearest even) and l = f16((x - h) * 2^11) -- the residual,
// scaled so that it is a normal f16 wherever h is (x - h is exact in f32; |x - h| <= 2^-11 |x|), i.e. x = h + l 2^-11 + r with
// |r| <= 2^-22 |x|: 22 significant bits and round-to-nearest in both pieces, so the representation error is unbiased and on average
// below f32's own rounding unit.  A product is accumulated in f32 by v_mfma_f32_16x16x32_f16 as
//   a.b ~= aH.bH + 2^-11 (aH.bL + aL.bH)                               (dropped: aL.bL 2^-22, relative 2^-22)
// with the scaled terms in an accumulator of their own, combined once at the end (acc0 + 2^-11 acc1): THREE K = 32 MFMAs per product
// (~16 cycles each) instead of eight v_mfma_f32_16x16x4_f32 (32 cycles each), 5.3x the matrix-pipe rate of the f32 instruction.
// Measured against float64 on this network's layers (tools/f16x2_error.py): max error 1.9e-7 / rms 3.8e-8 on outputs of magnitude ~2,
// against 1.2e-6 / 1.3e-7 for an ordinary f32 GEMM (sgemm) of the same operands -- the scheme is inside the f32 rounding class; the
// fused chains are tested against the float64 oracle at 1e-5 like the per-layer f32 path.  (Round 1 used three bf16 pieces and six
// MFMAs per product; f16 pieces halve both the MFMAs and the splitting arithmetic: ~3 VALU per value.)
// Range: operands must be finite with |x| < 65504 (weights, post-ReLU activations: always; gradients are carried scaled by a power of two
// chosen from the loss scale, fused_bwd.hip).  f16 subnormals are honoured by the matrix pipe on gfx950 (tools/probe/f16_denorm.hip), so a
// piece below 2^-14 only falls back from relative to absolute accuracy (2^-25 on h, 2^-36 on l).
// A wave-uniform global pointer made opaque to the optimiser (so that the addresses computed from it are not hoisted out of the
// enclosing loop as VGPR-pair invariants and spilled) and re-marked as global memory (so that the loads stay global_load, not flat).
__device__ __forceinline__ const u32x4* opaque_global(const u32x4* p) {
    const __attribute__((address_space(1))) u32x4* g = (const __attribute__((address_space(1))) u32x4*)p;
    asm volatile("" : "+s"(g));
    return (const u32x4*)g;
}

typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
#define F16_LO_SCALE 2048.f           // 2^11
#define F16_LO_INV (1.f / 2048.f)
struct F16x2 { u32x4 h, l; };         // 8 values: pieces packed two per dword (element e in the low half of dword e/2)

// two values -> {h pair, l pair} in FOUR instructions: v_cvt_pk_f16_f32 (h pair), v_pk_mul_f32 (x 2^11), then v_fma_mixlo_f16 /
// v_fma_mixhi_f16 = f16(fma(h as f32, -2^11, x 2^11)) straight from the packed h into the two halves of l -- the same roundings as the
// plain form (convert h back, fma, convert: six instructions, which is what hipcc emits for it), so the same bits
__device__ __forceinline__ void split_f16x2_pair(float x0, float x1, u32& h, u32& l) {
    const f16x2 hp = {(_Float16)x0, (_Float16)x1};
    const u32 hu = __builtin_bit_cast(u32, hp);
    const f32x2 sc = f32x2{x0, x1} * F16_LO_SCALE;
    const float ns = -F16_LO_SCALE;
    u32 lu;
    asm("v_fma_mixlo_f16 %0, %1, %2, %3 op_sel_hi:[1,0,0]" : "=v"(lu) : "v"(hu), "v"(ns), "v"(sc[0]));
    asm("v_fma_mixhi_f16 %0, %1, %2, %3 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "+v"(lu) : "v"(hu), "v"(ns), "v"(sc[1]));
    h = hu;
    l = lu;
}

__device__ __forceinline__ F16x2 split_f16x2(const f32x4& x0, const f32x4& x1) {
    F16x2 o;
    const float v[8] = {x0[0], x0[1], x0[2], x0[3], x1[0], x1[1], x1[2], x1[3]};
#pragma unroll
    for (int e = 0; e < 8; e += 2) { u32 h, l; split_f16x2_pair(v[e], v[e + 1], h, l); o.h[e >> 1] = h; o.l[e >> 1] = l; }
    return o;
}

// one f32 value -> its two f16 pieces (qnet.h), as raw halves
__device__ __forceinline__ void split_f16x2_one(float v, unsigned short& h, unsigned short& l) {
    const _Float16 vh = (_Float16)v, vl = (_Float16)((v - (float)vh) * F16_LO_SCALE);
    h = __builtin_bit_cast(unsigned short, vh);
    l = __builtin_bit_cast(unsigned short, vl);
}


// ---- LDS-DMA the compiler does not see ---------------------------------------------------------------------------------------------
// global -> LDS copies (per-lane global address, lane l lands at lds_base + size * l, inactive lanes copy nothing).  Behind the builtin
// (__builtin_amdgcn_global_load_lds) hipcc inserts s_waitcnt vmcnt(0) in front of the NEXT LDS access of any kind -- it cannot tell that the
// access does not touch the copy's destination -- so a copy issued "one stage ahead" was in fact waited for at once: round 3 found every
// LDS-DMA pipeline of this library (dense weight gradients, convolutional backward, the persistent conv forward) serialised that way
// (ISA: s_waitcnt vmcnt(0) right behind each global_load_lds; phase stamps: an iteration = copy latency + compute).  Written as inline assembly
// the copy stays in flight until the kernel's OWN s_waitcnt vmcnt(0) (every pipeline has one, in front of the barrier that publishes the
// stage).  hipcc's vmcnt bookkeeping for other loads stays correct: a wave's memory operations retire in order, so a wait it computes for
// a load can only turn out stricter than needed (it does not count these copies), never weaker.
__device__ __forceinline__ void lds_dma16(const void* gptr, u32 lds_base) {
    u32 m0_saved;                                                   // (m0 = the copy's LDS base; restored: it is not an asm clobber hipcc honours)
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "v"(gptr), "s"(lds_base) : "memory");
}
__device__ __forceinline__ void lds_dma4(const void* gptr, u32 lds_base) {
    u32 m0_saved;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dword %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(m0_saved) : "v"(gptr), "s"(lds_base) : "memory");
}
__device__ __forceinline__ u32 lds_addr(const void* p) { return (u32)(uintptr_t)(const __attribute__((address_space(3))) void*)p; }

#define MFMA_F16(a, b, c) __builtin_amdgcn_mfma_f32_16x16x32_f16(__builtin_bit_cast(f16x8, (a)), __builtin_bit_cast(f16x8, (b)), (c), 0, 0, 0)

// acc0 takes the leading piece product, acc1 the two 2^11-scaled cross terms; the caller combines them with f16x2_sum
__device__ __forceinline__ void mma_f16x3(const F16x2& a, const F16x2& b, f32x4& acc0, f32x4& acc1) {
    acc0 = MFMA_F16(a.h, b.h, acc0);
    acc1 = MFMA_F16(a.h, b.l, acc1);
    acc1 = MFMA_F16(a.l, b.h, acc1);
}
// max(x, 0) as ONE instruction whatever produced x (fmaxf behind a packed operation costs a second v_max_f32 that canonicalises its operand)
__device__ __forceinline__ float relu1(float x) { float y; asm("v_max_f32 %0, 0, %1" : "=v"(y) : "v"(x)); return y; }
__device__ __forceinline__ float f16x2_sum(float acc0, float acc1) { return __builtin_fmaf(acc1, F16_LO_INV, acc0); }
// Range guard of the fused FORWARD (round 6).  An activation of 65504 or more (or a non-finite one) leaves the f16 pieces' range: its h piece is inf, the products
// it enters are inf - inf = NaN, and the next ReLU (v_max_f32 returns the non-NaN operand) turns that into 0 -- the Q-values would come out finite and wrong.  Every
// epilogue that splits activations therefore folds what it splits into a per-lane maximum that lives for THAT epilogue only (post-ReLU values: one v_max3_f32
// per pair -- range_max), compares it with 65504 once at the epilogue's end and ORs the wave's ballot into a wave-uniform mask (range_commit: two SGPRs; a maximum
// kept per lane across the whole kernel cost conv_wave_kernel its 128-register budget -- 66 spills --, a ballot per pair of values cost it 2.9 of its 29.4 us).  A
// wave whose mask is not empty raises bit 1 of the handle's range word at its end (fused_range_flag: bit 0 = the backward's guard) -- dq_qnet_range_check then
// returns DQ_ERR_RANGE for the forward too.  conv_wave_kernel's FIRST layer is guarded without any run-time work: its input is binary, so a1[c] <= the sum of the
// positive entries of column c of the c1w block (bias and constant rows included) -- checked by pack_weights_kernel where it builds that block.  (A sticky hardware flag would have been free: TRAPSTS.EXCP stays 0 across an
// overflowing v_cvt_f16_f32 on gfx950 unless traps are enabled -- tools/probe/trapsts_probe.hip.)  A NaN argument does not move the maximum; it cannot arise
// before an inf that does: observations are binary, parameters are checked finite and < 65504 when they are packed (pack_weights_kernel).
typedef unsigned long long range_mask;     // lanes that met an activation outside the pieces' range (wave-uniform: lives in SGPRs)
#ifdef DQ_NO_RANGE_TRACK                   // (A/B builds, tools/build_ab.sh: what the guard costs)
__device__ __forceinline__ void range_track(range_mask&, float, float) {}
__device__ __forceinline__ void range_track4(range_mask&, const f32x4&) {}
__device__ __forceinline__ void range_track_abs(range_mask&, float, float) {}
__device__ __forceinline__ void range_track_finite(range_mask&, float) {}
__device__ __forceinline__ void range_max(float&, float, float) {}
__device__ __forceinline__ void range_max4(float&, const f32x4&) {}
__device__ __forceinline__ void range_commit(range_mask&, float) {}
#else
__device__ __forceinline__ void range_max(float& m, float a, float b) { m = fmaxf(m, fmaxf(a, b)); }                                       // one v_max3_f32
__device__ __forceinline__ void range_max4(float& m, const f32x4& v) { m = fmaxf(fmaxf(m, fmaxf(v[0], v[1])), fmaxf(v[2], v[3])); }       // two
__device__ __forceinline__ void range_commit(range_mask& bad, float m) { bad |= __ballot(!(m < 65504.f)); }
__device__ __forceinline__ void range_track(range_mask& bad, float a, float b) { bad |= __ballot(!(fmaxf(a, b) < 65504.f)); }
__device__ __forceinline__ void range_track4(range_mask& bad, const f32x4& v) { bad |= __ballot(!(fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3])) < 65504.f)); }
__device__ __forceinline__ void range_track_abs(range_mask& bad, float a, float b) { bad |= __ballot(!(fmaxf(fabsf(a), fabsf(b)) < 65504.f)); }
__device__ __forceinline__ void range_track_finite(range_mask& bad, float q) { bad |= __ballot(!(fabsf(q) < INFINITY)); }      // (an f32 output: NaN counts)
#endif
// (every lane reports its own copy of the mask: where a tracker sits under divergent control flow -- row guards -- the mask is per lane, and every lane that was
// active there holds the ballot's bits)
__device__ __forceinline__ void range_report(range_mask bad, unsigned* flag) { if (flag && bad != 0) atomicOr(flag, 2u); }
__device__ __forceinline__ f32x4 f16x2_sum(const f32x4& acc0, const f32x4& acc1) { return acc1 * F16_LO_INV + acc0; }

#endif  // __HIPCC__

// One launch serves up to FWD_MAX_JOBS independent forwards ("jobs": e.g. Q_target(s1), Q_online(s1) and the training forward
// on s0 of one DQN update, plus the acting forward): their serial phases (staging, epilogues, head layers) overlap in one grid.
struct ConvJob {
    const float* params;
    const u32x4* packed;               // f16 pieces of the conv kernels (PK_* below)
    const u8* obs;
    const int32_t* index;
    int index_off, index_mod, batch;
    float* act_out[3];                 // global NHWC [batch*oh*ow, cout]; [2] always written, [0] in training; [1] unused: in training the second
    unsigned short* a1_pl;             // (the first's the same way: [batch*oh1*ow1][64], l plane a1_lo halves further)
    size_t a1_lo;
    unsigned short* a2_pl;             // convolution's output is kept as f16 piece planes [batch*oh2*ow2][32] (h plane; the l plane a2_lo halves
    size_t a2_lo;                      // further) -- the form the convolutional backward consumes it in (fused_bwd.hip)
    unsigned short* x_pl;              // != NULL (conv_wave_kernel, round 6): the LAST convolution's output leaves as f16 piece planes [batch][oh3*ow3*cout] (h plane; the l
    size_t x_lo;                       // plane x_lo halves further): the dense chain stages them by LDS-DMA (no split, no registers) and the dense weight gradients read
                                       // the same planes; f32 act_out[2] is then written for the training job only (the dense data gradient's ReLU mask)
    int write_all;                     // training: bit 0 = write every layer the backward reads; bit 1 (conv_wave_kernel only) = ... including a1 (else the
                                       // backward recomputes it from the patch words: conv_bwd16.hip)
    int wg0;                           // first workgroup of this job
};

// ---- packed weights (fused.hip: pack_weights_kernel) --------------------------------------------------------------------------
// The f16x2 contractions read their weight operand as ready-made f16 pieces in MFMA operand order: one "block" = 64 lanes x
// 2 pieces x 16 bytes (lane (kb, j) holds the 8 reduction indices 8kb .. 8kb+7 of its column; h pieces at +0, l pieces at +64).
// Sections, in u32x4 units:
//   PK_CONV2_FWD  [8 k-blocks][2 column tiles]      B(k = 32 blk + 8kb + e, col = 2j + t)            = W2[k][col]
//   PK_CONV3_FWD  [4][2]                            same for conv3
//   PK_CONV3_DG   [4 taps][2]                       B(n = 8kb + e, c = 2j + t)                       = W3[tap][c][n]   (a lane's two results are
//   PK_CONV2_DG   [2 channel halves][4 taps][2]     B(n = 8kb + e, c = 32 half + 2j + t)             = W2[tap][c][n]    adjacent channels)
//   PK_CONV1      [3 k-blocks][4 column tiles]      B(k = 32 blk + 8kb + e, col = 4j + t)            = W1[k][col], 0 past K1
#define PK_BLOCK 128                  // u32x4 per block (2 pieces x 64 lanes)
#define PK_LO 64                      // the l pieces of a block
#define PK_CONV2_FWD 0
#define PK_CONV3_FWD (PK_CONV2_FWD + 16 * PK_BLOCK)
#define PK_CONV3_DG (PK_CONV3_FWD + 8 * PK_BLOCK)
#define PK_CONV2_DG (PK_CONV3_DG + 8 * PK_BLOCK)
#define PK_CONV1 (PK_CONV2_DG + 16 * PK_BLOCK)
#define PK_TOTAL_BLOCKS 60
#define PK_TOTAL_U32X4 (PK_TOTAL_BLOCKS * PK_BLOCK)     // the conv sections; then the dense sections (PackLayout, blocks of PK_BLOCK):
//   dense1   [K1/32 k-blocks][32 column tiles]   B(k' = 32 blk + 8kb + e, col = 64 (ct>>2) + 4j + (ct&3)) = W1[row(k')][col]     forward, Dense(512);
//            k' runs in the input rows' own NHWC order, row(k') = the Keras Flatten row c*hw + p of k' = p*C + c
//   dense2   [16][NT2]                           B(k = 32 blk + 8kb + e, col = NT2 j + t) = W2[k][col], 0 past N2                forward, Dense(|A|)
//   dense2t  [KB2][32]                           B(n2 = 32 blk + 8kb + e, n1 = 64 (ct>>2) + 4j + (ct&3)) = W2[n1][n2], 0 past N2   backward, gH1
//   dense1t  [16][K1/16]                         B(n1 = 32 blk + 8kb + e, k' = 16 ct + j) = W1[row(k')][n1]                         backward, gX
//   w3q      f32 [16 KG3 + 1][16 NT2]            the dueling layer folded with its combination (Q = y2 W3' + b3'): W3'[k][a] = W3[k][0] + W3[k][1 + a]
//                                                - mean_a' W3[k][1 + a'], zero past N2 / |A|; the last row is b3' (the same map of the bias)                         forward, Q
//            then f32 [16 NT2][16 KG3]           its transpose W3'^T [a][k]: the dueling backward folded the same way, gY2 = dq W3'^T                             backward, gY2
//   wc       f32 [|A|][512]                      Wc = W3'^T W2^T (Dense(|A|) folded in as well): row a = gH1 of a sample whose dq is 1 at action a               backward, gH1 (TD launch)
//   c1c      [1 k-block][4 column tiles]         the first convolution over patch words: B(k = 8kb + e, col = 4j + t) = W1[PT_KROW[k]][col], 0 past the data bits
//   b1p      f32 [r1][64]                        its per-pixel bias: b1[c] + the kernel rows of the pixel's constant-1 cells (summed in double)
//   c1w      [4 channel quarters]                the same kernel for conv_wave.hip, the per-pixel bias folded into the contraction: rows k < K_data as c1c; row K_data + c
//                                                = the kernel rows of constant position c summed over the syndrome planes (in double); row K_data + 5 = b1; then 0.
//                                                B(k = 8kb + e, col = 16 quarter + j) -- read as the FIRST operand there (rows = channels: the transposed product).
//                                                Present when K_data + 6 <= 32 (c1w_blocks = 4, else 0)
//   c2w      [4 quarters][2 ky][2 column tiles]  the second convolution for conv_wave.hip: a K block = the taps (ky, 0), (ky, 1) x 16 input channels:
//                                                B(slot (kb, e), col = 2j + t) = W2[(2 ky + (kb >> 1)) * 64 + 16 quarter + 8 (kb & 1) + e][col]
//   cdw      [8 + 16 blocks]                     the data gradients' weights for conv_bwd16.hip, read as the FIRST operand (rows = the input channels of a tile of 16:
//                                                the transposed product): conv3 [4 taps][2 tiles], then conv2 [4 taps][4 tiles]: B(n = 8kb + e, c = 16 t + j) = W[tap][c][n]
struct PackLayout { size_t dense1, dense2, dense2t, dense1t, w3q, wc, c1c, b1p, c1w, c2w, cdw, total; int d1_blocks, d2_blocks, d2t_blocks, d1t_blocks, NT2, KB2, w3q_rows, wc_rows, b1p_rows, c1w_blocks, c2w_blocks, cdw_blocks; };   // offsets in u32x4
PackLayout fused_pack_layout(const dq_qnet* Q);
static inline int dq_planes_small_ld(const dq_qnet* Q) { return Q->cfg.n_actions + 1 <= 64 ? 64 : 128; }
static inline size_t dq_planes_halves(const dq_qnet* Q) {      // total size of dq_qnet.planes
    return (size_t)2 * Q->cfg.max_batch * (Q->L[Q->cfg.n_conv].nin + 2 * DENSE_HID + 3 * dq_planes_small_ld(Q));
}
// plane set i (0 x, 1 h1, 2 gh1, 3 gy2, 4 g3, 5 y2): first half (the h plane; the l plane follows max_batch * ld halves further)
static inline unsigned short* dq_plane(const dq_qnet* Q, int i) {
    const size_t mb = (size_t)Q->cfg.max_batch, k1 = Q->L[Q->cfg.n_conv].nin, sl = dq_planes_small_ld(Q);
    const size_t off[6] = {0, k1, k1 + DENSE_HID, k1 + 2 * DENSE_HID, k1 + 2 * DENSE_HID + sl, k1 + 2 * DENSE_HID + 2 * sl};
    return Q->planes + 2 * mb * off[i];
}
size_t fused_packed_u32x4(const dq_qnet* Q);
size_t fused_packed_w1t_u32x4(const dq_qnet* Q);           // u32x4 offset of the f32 transpose W1T behind the pieces
size_t fused_packed_w2t_u32x4(const dq_qnet* Q);           // ... of W2T
dq_status fused_pack_weights(const dq_qnet* Q, const float* params_dev, void* packed_dev, hipStream_t st);

// dqn.hip: dq_adam_step whose skipped (non-finite) gradient elements raise *flag_dev (dq_qnet_adam_step)
dq_status adam_step_flagged(float* params_dev, const float* grads_dev, float* m_dev, float* v_dev, size_t n, double lr, double beta_1, double beta_2,
                            double epsilon, uint64_t t, unsigned* flag_dev, hipStream_t st);
// fused.hip: LDS-resident forward (conv chain + dense chain); returns false when the configuration is not covered
bool fused_forward_supported(const dq_qnet* Q);
bool fused_conv_row_tables(const dq_qnet* Q, int* tab);      // tab: int[CONV_FWD_TABS * CONV_ROWTAB]
bool fused_conv_bwd_row_tables(const dq_qnet* Q, int* tab);  // tab: int[5 * CONV_ROWTAB] (fused_bwd.hip)
bool fused_patch_supported(const dq_qnet* Q, int depth);     // patch-word input possible for this network with `depth` syndrome planes
void fused_patch_tables(const dq_qnet* Q, int depth, int stride_words, int* tab);      // tab: int[PT_TOTAL]
dq_status fused_forward_multi(dq_qnet* Q, int n_jobs, const dq_qnet_job* jobs, hipStream_t st);
// Does a training minibatch of B samples take conv_bwd16_kernel (fused_bwd.hip fused_backward's rule, shared with the forward: a conv_wave_kernel training job
// saves a1 only when its backward will read it)?
bool conv_bwd16_applies(const dq_qnet* Q, int B, bool patch);
// conv_wave.hip: the conv forward's wave-private form (one sample per wave, weights in LDS, no barriers): patch-word input, d = 5
struct ConvWaveArgs {
    ConvJob job[FWD_MAX_JOBS];
    int n_jobs;
    // weight sets: jobs with the same packed buffer and parameters.  A set's pairs of samples -- job after job, a job's pairs (2 i, 2 i + 1) -- form ONE list of
    // end[3] pairs; workgroup r of the set's `wgs` takes the pairs r, r + wgs, ...: every workgroup gets the same number (+- 1) and the same share of every job
    // (the training job's stores are spread over all of them)
    struct Set {
        const u32x4* packed;
        const float* params;
        int wg0, wgs;                                   // first workgroup (INT_MAX: no such set), workgroups
        int job[FWD_MAX_JOBS];                          // its jobs in list order (padded with the last)
        int end[FWD_MAX_JOBS];                          // end of each job's pairs in the list (padded with the total)
        int batch[FWD_MAX_JOBS];                        // the jobs' sample counts
    } set[FWD_MAX_JOBS];
    int set_wg0[FWD_MAX_JOBS];                          // = set[c].wg0: the set of a workgroup by three comparisons on one scalar load ...
    const u32x4* set_packed[FWD_MAX_JOBS];              // ... which also brings the sets' packed buffers: the weight copies go out one round trip after entry
    int b_off[3];                                       // floats into params: the three biases
    int slot;                                           // bytes per observation row (4 * patch_stride)
    int pk_c1w, pk_c2w;                                 // u32x4 offsets of the c1w / c2w sections inside a packed buffer
    int kd;                                             // data bits per patch word
    const int* ptab;                                    // PT_* tables
    unsigned* range_flag;                               // the forward's range guard (range_report), nullable
};
bool conv_wave_supported(const dq_qnet* Q);
dq_status conv_wave_launch(const dq_qnet* Q, ConvWaveArgs& a, int n_cu, hipStream_t st);
void conv_wave_lane_table(const dq_qnet* Q, int kd, const int* pt_const, int* out);
void conv_bwd16_tables(const dq_qnet* Q, int kd, int stride_words, const int* pt_const, const int* pt_src, int* out);      // out: int[PT_C16_INTS] (qnet.h PT_C16)      // out: int[64 * PT_WAVE_LD] (qnet.h PT_WAVE)
// qnet.hip: per-layer backward pieces (also used by the fused backward for layers it does not cover)
dq_status layer_wgrad(dq_qnet* Q, int layer, float* grads_dev, hipStream_t st);
dq_status layer_dgrad(dq_qnet* Q, const float* params_dev, int layer, hipStream_t st);
// fused_bwd.hip: fused backward (data-gradient chains + all-layer weight gradients) for the same configurations
bool fused_backward_supported(const dq_qnet* Q);
size_t fused_backward_workspace_floats(const dq_qnet* Q);
unsigned* fused_range_flag(const dq_qnet* Q);     // device word of the range guard (dq_qnet_range_check), NULL without the fused backward
// opt != NULL (phases == 3 only): the final reduction also applies the Adam update to p/m/v (one launch fewer per update)
struct AdamOpt { float* p; float* m; float* v; float lr_t, b1, b2, eps; };
// td != NULL: the TD step (dq_td_update's arithmetic) runs in the dense backward's prologue instead of reading dq_dev, and the episode
// bookkeeping of the step just taken (st_n > 0) rides on the same launch
struct TdFused {
    const float *q1o, *q1t, *q0, *reward;
    const u8* terminal;
    const int32_t *action, *index;
    float gamma, grad_scale;
    float *y_out, *dq_out, *metrics;    // nullable
    int metric_slots;                   // partial slots dq_td_metrics will read for this batch
    const u8 *st_done, *st_was_reset;
    const u32* st_lifetime;
    const float* st_reward;
    int st_n;
    unsigned long long* st_stats;
    int auto_scale;                     // dq_td_job.auto_scale: the gradient scale from this minibatch's max |TD error| (fused_bwd.hip td_scale_kernel)
};
// rider != NULL (needs td): the lattices' environment step (env_dev.h parameters, filled by env_fill_act_step) runs as extra
// workgroups of the dense backward's first launch
struct EnvParams;
int fused_rider_threads();            // threads per block of the launch that carries the riding environment step: 512 (dense data gradients) or 256 (dense weight gradients)
dq_status fused_backward(dq_qnet* Q, const float* params_dev, const float* dq_dev, float* grads_dev, int phases, hipStream_t st,
                         const AdamOpt* opt = nullptr, const TdFused* td = nullptr, const EnvParams* rider = nullptr, size_t rider_lds = 0);
