// fp32 MFMA implicit-GEMM building blocks for the convolutional Q-network (gfx950).
//
// The Q-network forward/backward of the reference (Keras Conv2D/Dense layers built by
// build_convolutional_nn, /root/reference/cluster_scripts/d5_dp/0.001/Single_Point_Training_Script.py:61-90)
// is a chain of dense contractions.  Every one of them -- convolution forward, its data gradient
// (a transposed convolution), its weight gradient, and the dense layers -- is expressed here as
//     C[m, n] = sum_k A(m, k) * B(k, n)
// where A is GATHERED on the fly (im2col never materialised) through a separable address map
//     A(m, k) = src[ rowoff(m) + koff(k) ],
// B is read through a two-level stride map (so W, W^T and the HWIO->"(ky,kx,n) x c" view of a conv
// kernel need no copies), and the epilogue applies bias / ReLU / dropout / ReLU-mask and an optional
// column permutation (Keras channels_first Flatten <-> the NHWC activations kept on the device).
//
// Arithmetic: v_mfma_f32_32x32x2_f32 -- f32 in, f32 accumulate, bit-exact with an fmaf chain in k
// order (cdna_hip_programming.md §3), so results do not depend on tiling beyond the k order.
#pragma once
#include "common.h"

typedef float f32x16 __attribute__((ext_vector_type(16)));

struct Gather {
    const void* src;
    int is_u8;
    int rows_per_sample, RW;          // m -> (b, y, x): b = m / rows_per_sample, y = r / RW, x = r % RW
    unsigned sb, sy, sx;              // rowoff = bidx*sb + y*sy + x*sx
    const int32_t* index;             // optional sample -> source row map (replay gather): bidx = (index[b] + index_off) % index_mod
    int index_off, index_mod;
    int KC, KW;                       // k -> (ky, kx, c): c = k % KC, kx = (k / KC) % KW, ky = (k / KC) / KW
    int sky, skx, sc;                 // koff = ky*sky + kx*skx + c*sc   (may be negative for transposed convs)
    int check, ylim, xlim;            // transposed conv: element valid iff 0 <= y-ky < ylim and 0 <= x-kx < xlim
};

struct BMap {                         // B(k, n) = w[(k / KD)*s1 + (k % KD)*s2 + n*sn]
    const float* w;
    int KD, s1, s2, sn;
};

enum { EPI_NONE = 0, EPI_BIAS = 1, EPI_RELU = 2, EPI_DROPOUT = 4, EPI_MASK = 8 };

struct Epilogue {
    float* out;
    int ldo;                          // out offset = m*ldo + (n / PC)*s_hi + (n % PC)*s_lo
    int PC, s_hi, s_lo;
    int flags;
    const float* bias;                // EPI_BIAS
    const float* mask_src;            // EPI_MASK: multiply by (mask_src[same offset] > 0) * mask_scale
    float mask_scale;
    float keep_scale;                 // EPI_DROPOUT: y = keep ? y * keep_scale : 0  (keep_scale = 1/(1-rate))
    u32 drop_T;                       //   dropped iff the unit's 16-bit draw < drop_T (dq_rate_threshold16)
    u32 seed0, seed1;                 //   word = Philox(key=seed, ctr=(t_lo, t_hi, sample_base + m, (n>>2) | DROPOUT<<16))[n&3]
    u64 t;
    u32 sample_base;
};

struct RowRef { unsigned off; int yx; };   // yx = y << 16 | x, or -1 for a row past M

__device__ __forceinline__ RowRef gather_row(const Gather& g, int m, int M) {
    RowRef r;
    if (m >= M) { r.off = 0; r.yx = -1; return r; }
    const int b = m / g.rows_per_sample, rr = m - b * g.rows_per_sample;
    const int y = rr / g.RW, x = rr - y * g.RW;
    int bidx = b;
    if (g.index) { bidx = g.index[b] + g.index_off; if (bidx >= g.index_mod) bidx -= g.index_mod; }
    r.off = (unsigned)bidx * g.sb + (unsigned)y * g.sy + (unsigned)x * g.sx;
    r.yx = (y << 16) | x;
    return r;
}

// Walks consecutive rows m, m+step, ... without re-dividing (used where a thread visits many rows).
struct RowIter {
    int b, y, x;
    __device__ __forceinline__ void init(const Gather& g, int m) {
        b = m / g.rows_per_sample;
        const int rr = m - b * g.rows_per_sample;
        y = rr / g.RW;
        x = rr - y * g.RW;
    }
    __device__ __forceinline__ void advance(const Gather& g, int step) {
        x += step;
        const int RH = g.rows_per_sample / g.RW;
        while (x >= g.RW) { x -= g.RW; ++y; }
        while (y >= RH) { y -= RH; ++b; }
    }
    __device__ __forceinline__ RowRef ref(const Gather& g, bool valid) const {
        RowRef r;
        if (!valid) { r.off = 0; r.yx = -1; return r; }
        int bidx = b;
        if (g.index) { bidx = g.index[b] + g.index_off; if (bidx >= g.index_mod) bidx -= g.index_mod; }
        r.off = (unsigned)bidx * g.sb + (unsigned)y * g.sy + (unsigned)x * g.sx;
        r.yx = (y << 16) | x;
        return r;
    }
};

struct ColRef { int off; int kyx; };      // kyx = ky << 16 | kx, or -1 for a column past K

__device__ __forceinline__ ColRef gather_col(const Gather& g, int k, int K) {
    ColRef c;
    if (k >= K) { c.off = 0; c.kyx = -1; return c; }
    const int t = k / g.KC, cc = k - t * g.KC;
    const int ky = t / g.KW, kx = t - ky * g.KW;
    c.off = ky * g.sky + kx * g.skx + cc * g.sc;
    c.kyx = (ky << 16) | kx;
    return c;
}

__device__ __forceinline__ float gather_load(const Gather& g, RowRef r, ColRef c) {
    if (r.yx < 0 || c.kyx < 0) return 0.f;
    if (g.check) {
        const int y = (r.yx >> 16) - (c.kyx >> 16), x = (r.yx & 0xffff) - (c.kyx & 0xffff);
        if ((unsigned)y >= (unsigned)g.ylim || (unsigned)x >= (unsigned)g.xlim) return 0.f;
    }
    const unsigned off = r.off + (unsigned)c.off;
    return g.is_u8 ? (float)static_cast<const u8*>(g.src)[off] : static_cast<const float*>(g.src)[off];
}

__device__ __forceinline__ float bmap_load(const BMap& b, int k, int n, int K, int N) {
    if (k >= K || n >= N) return 0.f;
    if (b.KD >= (1 << 30)) return b.w[k * b.s2 + n * b.sn];      // plain strided view: no division
    const int hi = k / b.KD, lo = k - hi * b.KD;
    return b.w[hi * b.s1 + lo * b.s2 + n * b.sn];
}

__device__ __forceinline__ void epilogue_store(const Epilogue& e, int m, int n, float v) {
    size_t off;
    if (e.PC >= (1 << 30)) {
        off = (size_t)m * e.ldo + (size_t)n * e.s_lo;            // plain row-major: no division
    } else {
        const int hi = n / e.PC, lo = n - hi * e.PC;
        off = (size_t)m * e.ldo + (size_t)hi * e.s_hi + (size_t)lo * e.s_lo;
    }
    if (e.flags & EPI_BIAS) v += e.bias[n];
    if (e.flags & EPI_RELU) v = fmaxf(v, 0.f);
    if (e.flags & EPI_DROPOUT) {
        u32 w[4];
        // one Philox call = eight consecutive units of a sample: unit n draws half-word n & 7 (word (n & 7) >> 1, low half first)
        philox4x32_10((u32)e.t, (u32)(e.t >> 32), e.sample_base + (u32)m, ((u32)n >> 3) | ((u32)DQ_STREAM_DROPOUT << 16), e.seed0, e.seed1, w);
        const int wi = (n & 7) >> 1;
        const u32 word = wi == 0 ? w[0] : wi == 1 ? w[1] : wi == 2 ? w[2] : w[3];
        v = (((word >> (16 * (n & 1))) & 0xffffu) < e.drop_T) ? 0.f : v * e.keep_scale;
    }
    if (e.flags & EPI_MASK) v = e.mask_src[off] > 0.f ? v * e.mask_scale : 0.f;
    e.out[off] = v;
}
