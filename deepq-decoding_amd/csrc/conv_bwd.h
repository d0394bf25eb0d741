// Shared by the two forms of the convolutional backward: fused_bwd.hip (conv_bwd_chain_kernel: workgroups of 8 waves, any covered geometry) and
// conv_bwd16.hip (conv_bwd16_kernel: workgroups of 16 waves, patch-word input at d = 5).
#ifndef DEEPQ_CONV_BWD_H
#define DEEPQ_CONV_BWD_H
#include "qnet.h"

struct ConvBwdArgs {
    const float* params;
    const u8* obs;
    const int32_t* index;
    int index_off, index_mod;
    const unsigned short* a1p;          // saved first-convolution output [batch*r1][64] as f16 piece planes (h plane; the l plane a1_lo halves further)
    size_t a1_lo;
    const unsigned short* a2p;          // saved second-convolution output [batch*r2][32] as f16 piece planes (h plane; the l plane a2_lo halves further)
    size_t a2_lo;
    const unsigned short* g3p;          // [batch*r3][32] gradient w.r.t. conv3's pre-activation output, as f16 piece planes (l plane g3_lo halves further)
    size_t g3_lo;
    const u32x4* packed;                // f16 pieces of the conv kernels (qnet.h PK_*), those of the training forward
    int batch, S, groups;
    int C, H, W, k1, st1, K1;
    int oh1, ow1, oh2, ow2, oh3, ow3;
    int w_off[3], b_off[3];
    float* partial;                     // [gridDim.x][pstride]
    size_t pstride;
    int slot;
    int off_mis, off_a1, off_a2, off_g3, off_t1, off_t2, off_t3, off_ko, off_tp, off_d2, off_d1;
    int a1_alt;                         // bytes from the a1 image to its second buffer, 0 = single-buffered
    const int* kofftab;                 // [96] conv1 weight row k -> byte offset inside an observation, -1 past K1
    const int* rowtab;                  // [5][CONV_ROWTAB] host-built row tables (fused_conv_bwd_row_tables): the kernel copies them into LDS and never divides
    // patch-word input (the kernel's CP instances; qnet.h PT_*, include/deepq_hip.h dq_env_patch_output): the observation rows are `slot` bytes of u32
    // words (one per first-convolution output pixel), 16-byte aligned
    const int* rowtab1;                 // the first table: rowtab, or qnet.h PT_BWD (row m -> word index s * stride + p | the pixel's constant-cell mask << 16)
    const int* srctab;                  // qnet.h PT_SRC: Keras row of the first kernel -> column of the patch image, -1: gradient 0
    const int* tab16;                   // conv_bwd16.hip: its table blob (qnet.h PT_C16), device
    int pk_cdw;                         // conv_bwd16.hip: u32x4 offset of the packed data-gradient weights in channel-tile order (qnet.h cdw)
    int pk_c1w;                         // conv_bwd16.hip, a1 recomputed: u32x4 offset of the first kernel's quarters with the per-pixel bias folded in (qnet.h c1w)
    int a1_recompute;                   // conv_bwd16.hip: the training forward (conv_wave_kernel) did not save a1: recompute it from the patch words, the forward's own bits
    int kd, off_lut;                    // data bits per pixel; LDS: byte -> its eight bits as bytes 0 / 1 (256 x 8 bytes), built by the workgroup
};

#ifdef __HIPCC__
// Row stride (halves) of an LDS piece-plane image of 32 channels: 64 data bytes + 16 of padding, filled by LDS-DMA in 16-byte slots
#define PL32 40

// Transposing LDS read (ds_read_b64_tr_b16, tools/probe/tr_probe.hip): within a 16-lane group, lanes 4r .. 4r+3 each point at a 4-half
// segment of row r (r = 0 .. 3, any addresses); lane i receives column i of those four rows.  Two of them are the eight reduction
// indices of a 16 x 16 x 32 MFMA operand held ROW-major in LDS -- what the weight gradients (a reduction over pixels / batch rows) need.
typedef short s16x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ uint2 lds_tr4(const unsigned short* p) {
    const s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    return __builtin_bit_cast(uint2, v);
}
// rows p0 (reduction indices 0 .. 3 of this lane group) and p1 (4 .. 7), both pieces (l plane `lo` halves further)
__device__ __forceinline__ F16x2 lds_tr8(const unsigned short* p0, const unsigned short* p1, int lo) {
    const uint2 a = lds_tr4(p0), b = lds_tr4(p1), c = lds_tr4(p0 + lo), d = lds_tr4(p1 + lo);
    F16x2 o;
    o.h = u32x4{a.x, a.y, b.x, b.y};
    o.l = u32x4{c.x, c.y, d.x, d.y};
    return o;
}

#endif

bool conv_bwd16_supported(const dq_qnet* Q);
size_t conv_bwd16_lds();
dq_status conv_bwd16_launch(const dq_qnet* Q, ConvBwdArgs& a, int wgs, hipStream_t st);
#endif
