// Batched surface-code decoding environment for gfx950: one lattice per wavefront.
//
// Replaces Surface_Code_Environment_Multi_Decoding_Cycles.reset()/step()
// (/root/reference/example_notebooks/Environments.py:99-235, "ENV") together with the helper loops
// it calls in Function_Library.py ("FL": generate_error FL:67-122, syndrome FL:152-174,
// generate_faulty_syndrome FL:176-223, obtain_new_error_configuration FL:226-241, index_to_move
// FL:243-294, generate_one_hot_labels_surface_code FL:296-326).
//
// Formulation.  Pauli codes multiply as XOR (FL:54-62), so a lattice's hidden state is two qubit
// bit-planes (xmask, zmask) held wave-uniform in SGPRs.  Lane l owns qubit l and stabilizer l:
//   * noise: one Philox4x32-10 evaluation per lane per measurement round gives the lane's error
//     uniform, Pauli type and measurement uniform; three wave ballots turn 64 comparisons into the
//     round's error planes and measurement-flip word;
//   * syndrome: lane s ANDs the matching plane with its plaquette's qubit mask, popcounts, and a
//     ballot assembles the syndrome word (parity reduction by ballot, no LDS traffic);
//   * referee: a per-lane bit pick + ballot permutes the syndrome word into the two look-up
//     indices; the bit-packed tables sit in L2;
//   * observation: the C x (2d+1) x (2d+1) uint8 planes of the block's four lattices are composed
//     in LDS from the volume words / action mask and leave as coalesced dword stores.
// HBM traffic per lattice-step: one 128 B state record in, one out, plus the observation.
#include "env_dev.h"
#include <new>
#include <stdarg.h>
#include <vector>
#include <algorithm>

#define DQ_MLP_MAX_LAYERS 6
// lanes of one wave hand activations to each other through LDS (referee_mlp_kernel): wavefront-scope release / acquire + wave barrier
static __device__ __forceinline__ void match_wave_sync_env() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

struct dq_env {
    dq_env_cfg cfg;
    dq_env_info info;
    int sw;                // internal state stride in u64 words
    int P;                 // cells per observation plane
    u64 T_phys, T_meas;
    bool rates_set;
    EnvTables h_tab;
    EnvTables* d_tab;
    u64* d_state;
    u32 *d_lut_x, *d_lut_z;        // owned tables (dq_env_build_referee)
    const u32 *lut_x, *lut_z;      // tables in use
    const u32* lut_joint;          // caller's joint table (dq_env_set_referee_joint), else NULL
    u32 ref_delta[2][64];          // BFS generators per component (x, z)
    // Dense-stack referee evaluated on the device (dq_env_set_referee_mlp): layer sizes, the caller's flat weights, per-lattice result
    int mlp_layers, mlp_dims[DQ_MLP_MAX_LAYERS + 1];
    const float* mlp_w;
    u8* d_dec;                     // [n_envs] class predicted for each lattice's post-action syndrome
    int* d_mlp_cells;              // [2 * n_stab]: stabilizers in increasing cell order of the (d+1)^2 input vector, then their cells
    bool lut_marker;               // lut_x / lut_z only say "a referee is installed" (they point at the Dense stack's weights; never read)
    u32* patch_next;               // dq_env_patch_output: where the NEXT launch writes the lattices' patch words (one call arms one launch)
    int patch_stride;
};

// ---- Dense-stack referee on the device (round 3) ------------------------------------------------------------------------------------
// The reference's referee is a Keras feed-forward network called once per step on the flattened (d+1)^2 syndrome of the state AFTER the
// agent's move (ENV:139-144); only the arg-max of its output is used (ENV:150).  A joint table over all syndromes stops fitting at d = 7
// (48 stabilizers), so there the stack itself is evaluated: one wavefront per lattice applies the move to the lattice's planes (the
// arithmetic of env_block's step, without touching the record), forms the syndrome word, and runs the layers -- lane o owns outputs
// o, o + 64, ...; the first layer ADDS the kernel rows of the set cells (the input is binary), later layers walk the previous layer's
// activations out of LDS -- in a FIXED arithmetic: float32, bias first, then the inputs in increasing index order, one rounded multiply and
// one rounded add per term (no fused multiply-add), ReLU between layers, first maximum of the last layer's outputs (softmax is monotone).
// referee.py FeedForwardReferee.predict_exact restates exactly that on the host, so device and host classes agree bit for bit.
#define DQ_MLP_THREADS 256
struct RefMlpParams {
    const u64* state; int sw, n_envs, d2, n_stab, n_actions, identity, model, use_Y, auto_reset;
    const EnvTables* tab;
    const int32_t* action;
    const int* cells;                  // [n_stab] stabilizers in increasing cell order, then [n_stab] the cell of each of them
    int layers, dims[DQ_MLP_MAX_LAYERS + 1];
    const float* w;
    int max_width;
    u8* dec;
};

__global__ __launch_bounds__(DQ_MLP_THREADS) void referee_mlp_kernel(RefMlpParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int i = blockIdx.x * (DQ_MLP_THREADS / 64) + wave;
    if (i >= p.n_envs) return;                                       // wave-uniform; no block-wide barrier below
    float* h0 = reinterpret_cast<float*>(smem) + (size_t)wave * 2 * p.max_width;
    float* h1 = h0 + p.max_width;
    const u64* rec = p.state + (size_t)i * p.sw;
    u64 xmask = rec[0], zmask = rec[1];
    const u64 meta = rec[8];
    if (p.auto_reset && ((meta >> 32) & 1)) { if (lane == 0) p.dec[i] = 0; return; }       // this lattice is reset, not stepped: no referee call
    int a = p.action[i];
    if ((unsigned)a >= (unsigned)p.n_actions) a = p.identity;
    if (a != p.identity) {                                          // ENV:135-136, FL:243-294 (env_dev.h env_block)
        const int layer = a / p.d2, q = a - layer * p.d2;
        const int pauli = p.model == DQ_MODEL_X ? 1 : (p.use_Y ? layer + 1 : (layer == 0 ? 1 : 3));
        if (pauli != 3) xmask ^= 1ull << q;
        if (pauli != 1) zmask ^= 1ull << q;
    }
    const u64 sq = p.tab->stab_qmask[lane];
    const bool isx = p.tab->stab_isx[lane] != 0;
    const u64 true_word = __ballot(__popcll((isx ? xmask : zmask) & sq) & 1);   // ENV:139
    // ---- first layer: bias + the kernel rows of the set cells, in increasing cell order -----------------------------------------
    const float* w = p.w;
    int n_in = p.dims[0], n_out = p.dims[1];
    for (int o = lane; o < n_out; o += 64) {
        float acc = w[(size_t)n_in * n_out + o];
        for (int t = 0; t < p.n_stab; ++t) {
            const int s = p.cells[t];
            if ((true_word >> s) & 1) acc = __fadd_rn(acc, w[(size_t)p.cells[p.n_stab + t] * n_out + o]);      // x = 1: the product is the weight itself
        }
        h0[o] = p.layers > 1 ? fmaxf(acc, 0.f) : acc;
    }
    w += (size_t)n_in * n_out + n_out;
    float* src = h0;
    float* dst = h1;
    for (int l = 1; l < p.layers; ++l) {
        match_wave_sync_env();
        n_in = p.dims[l]; n_out = p.dims[l + 1];
        for (int o = lane; o < n_out; o += 64) {
            float acc = w[(size_t)n_in * n_out + o];
            int k = 0;
            for (; k + 8 <= n_in; k += 8) {                           // eight kernel rows requested together; the adds stay in index order
                float wv[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) wv[u] = w[(size_t)(k + u) * n_out + o];
#pragma unroll
                for (int u = 0; u < 8; ++u) acc = __fadd_rn(acc, __fmul_rn(src[k + u], wv[u]));
            }
            for (; k < n_in; ++k) acc = __fadd_rn(acc, __fmul_rn(src[k], w[(size_t)k * n_out + o]));
            dst[o] = l + 1 < p.layers ? fmaxf(acc, 0.f) : acc;
        }
        w += (size_t)n_in * n_out + n_out;
        float* t = src; src = dst; dst = t;
    }
    match_wave_sync_env();
    // ---- first maximum of the outputs (np.argmax) ---------------------------------------------------------------------------------
    const int nc = p.dims[p.layers];
    float best = lane < nc ? src[lane] : -INFINITY;
    int best_a = lane < nc ? lane : 0x7fffffff;
    dq_wave_argmax(best, best_a);
    if (lane == 0) p.dec[i] = (u8)best_a;
}

__global__ __launch_bounds__(256) void env_kernel(EnvParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    if (p.pair) env_block2<2 * ENVS_PER_BLOCK>(p, (int)blockIdx.x, smem);     // block-uniform (kernel argument): two lattices per wave, d <= 5
    else env_block<ENVS_PER_BLOCK>(p, (int)blockIdx.x, smem);
}

// p.steps agent steps of the lattices in one launch (env_dev.h env_block2<EPB, true>; dq_env_act_steps): d <= 5 (two lattices per wave)
__global__ __launch_bounds__(256) void env_multi_kernel(EnvParams p) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    env_block2<2 * ENVS_PER_BLOCK, true>(p, (int)blockIdx.x, smem);
}

// ---- state export / import (tests, checkpointing) ------------------------------------------------
__global__ void env_export_kernel(const EnvTables* tab, const u64* state, u64* out, int n_envs, int sw, int depth) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_envs) return;
    const u64* rec = state + (size_t)i * sw;
    const u64 word = lane < sw ? rec[lane] : 0;
    const u64 xmask = wave_bcast64(word, 0), zmask = wave_bcast64(word, 1);
    const u64 true_word = __ballot(__popcll((tab->stab_isx[lane] ? xmask : zmask) & tab->stab_qmask[lane]) & 1);
    u64 summed = 0;
    for (int j = 0; j < depth; ++j) summed |= wave_bcast64(word, STATE_FIXED + j);
    u64* o = out + (size_t)i * (EXPORT_FIXED + depth);
    // export order: xmask zmask true summed acted round comp0 comp1 legal0 legal1 meta volume...
    if (lane < 2) o[lane] = word;
    if (lane == 2) { o[2] = true_word; o[3] = summed; }
    if (lane >= 2 && lane < STATE_FIXED) o[lane + 2] = word;
    if (lane >= STATE_FIXED && lane < STATE_FIXED + depth) o[lane + 2] = word;
}

__global__ void env_import_kernel(u64* state, const u64* in, int n_envs, int sw, int depth) {
    const int i = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6), lane = threadIdx.x & 63;
    if (i >= n_envs) return;
    const u64* src = in + (size_t)i * (EXPORT_FIXED + depth);
    u64 v = 0;
    if (lane < 2) v = src[lane];
    else if (lane < STATE_FIXED + depth) v = src[lane + 2];
    if (lane < sw) state[(size_t)i * sw + lane] = v;
}

// ---- referee look-up tables: level-synchronous BFS over (syndrome, class) on the GPU --------------
struct BfsDeltas { u32 d[64]; int nq; };

__global__ void bfs_fill_kernel(u8* dist, size_t size) {
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < size; s += (size_t)gridDim.x * blockDim.x)
        dist[s] = s == 0 ? 0 : 255;
}

__global__ void bfs_level_kernel(u8* dist, u32 size, BfsDeltas dl, u8 w, u32* changed) {
    bool any = false;
    for (u32 s = blockIdx.x * blockDim.x + threadIdx.x; s < size; s += gridDim.x * blockDim.x) {
        if (dist[s] != (u8)(w - 1)) continue;
        for (int q = 0; q < dl.nq; ++q) {
            const u32 t = s ^ dl.d[q];
            if (dist[t] == 255) { dist[t] = w; any = true; }     // every racing writer stores the same w
        }
    }
    if (__any(any) && (threadIdx.x & 63) == 0) atomicOr(changed, 1u);
}

__global__ void bfs_pack_kernel(const u8* dist, u32 half, u32* lut) {
    const u32 wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (half + 31) / 32) return;
    u32 bits = 0;
    for (int b = 0; b < 32; ++b) {
        const u32 s = wi * 32 + b;
        if (s < half && dist[half + s] < dist[s]) bits |= 1u << b;   // class 1 iff strictly lighter
    }
    lut[wi] = bits;
}

// ---- host side ------------------------------------------------------------------------------------
static int plaquette_type(int d, int a, int b) {     // FL:32-35, FL:42-50
    if ((a == 0 && b % 2 == 0) || (a == d && b % 2 == 1) || (b == 0 && a % 2 == 1) || (b == d && a % 2 == 0)) return 0;
    return ((a + b) & 1) ? 3 : 1;
}

static void build_tables(dq_env* E) {
    EnvTables& T = E->h_tab;
    memset(&T, 0, sizeof(T));
    memset(T.ref_src, 255, sizeof(T.ref_src));
    memset(T.cell_stab, 255, sizeof(T.cell_stab));
    memset(T.cell_qubit, 255, sizeof(T.cell_qubit));
    const int d = E->cfg.d, n_stab = d * d - 1, half = (d + 1) / 2 - 1, n = 2 * d + 1;
    int sa[64], sb[64], index[16][16];
    for (int a = 0; a <= d; ++a) for (int b = 0; b <= d; ++b) index[a][b] = -1;
    int s = 0;
    for (int a = 1; a < d; ++a) for (int b = 1; b < d; ++b) { sa[s] = a; sb[s] = b; ++s; }   // FL:189-194
    for (int x = 0; x < half; ++x) { sa[s] = 0; sb[s] = 2 * x + 1; ++s; }                    // FL:197-202
    for (int x = 0; x < half; ++x) { sa[s] = d; sb[s] = 2 * x + 2; ++s; }                    // FL:203-208
    for (int x = 0; x < half; ++x) { sa[s] = 2 * x + 2; sb[s] = 0; ++s; }                    // FL:210-215
    for (int x = 0; x < half; ++x) { sa[s] = 2 * x + 1; sb[s] = d; ++s; }                    // FL:216-221
    for (s = 0; s < n_stab; ++s) {
        const int a = sa[s], b = sb[s];
        index[a][b] = s;
        T.stab_type[s] = (u8)plaquette_type(d, a, b);
        T.stab_isx[s] = T.stab_type[s] == 3;
        for (int x = a - 1; x <= a; ++x) for (int y = b - 1; y <= b; ++y)
            if (x >= 0 && x < d && y >= 0 && y < d) {
                T.stab_qmask[s] |= 1ull << (x * d + y);
                T.qubit_smask[x * d + y] |= 1ull << s;
            }
        T.cell_stab[2 * a * n + 2 * b] = (u8)s;                                              // ENV:292-294
    }
    // referee bit order: rank among same-type plaquettes in row-major (a,b); X part -> lanes 0.., Z part -> lanes 32..
    int rank3 = 0, rank1 = 0, ref_bit[64];
    for (int a = 0; a <= d; ++a) for (int b = 0; b <= d; ++b) {
        const int t = plaquette_type(d, a, b);
        if (t == 3) { ref_bit[index[a][b]] = rank3; T.ref_src[rank3++] = (u8)index[a][b]; }
        if (t == 1) { ref_bit[index[a][b]] = rank1; T.ref_src[32 + rank1++] = (u8)index[a][b]; }
    }
    for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
        for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {
            const int rr = r + dr, cc = c + dc;
            if ((dr || dc) && rr >= 0 && rr < d && cc >= 0 && cc < d) T.neigh_qmask[r * d + c] |= 1ull << (rr * d + cc);
        }
        T.cell_qubit[(2 * r + 1) * n + 2 * c + 1] = (u8)(r * d + c);                         // ENV:309-312
    }
    for (int x = 0; x < d; ++x) T.col0 |= 1ull << (x * d);
    for (int y = 0; y < d; ++y) T.row0 |= 1ull << y;
    for (int x = 0; x < n; ++x) for (int y = 0; y < n; ++y) {                                // ENV:284-298
        u8 v = 0;
        if ((x == 0 || x == n - 1) && (y & 1)) v = 1;
        if ((y == 0 || y == n - 1) && (x & 1)) v = 1;
        if ((x & 1) && (y & 1) && ((x + y) % 4 == 0)) v = 1;
        T.cell_static[x * n + y] = v;
    }
    for (int c = 0; c < 256; ++c) T.cell_pack[c] = (u32)T.cell_static[c] | (u32)T.cell_stab[c] << 8 | (u32)T.cell_qubit[c] << 16;
    // compact observation: the data cells of conv1's 3 x 3 stride-2 patch of output pixel (oy, ox) are its four corners on the syndrome planes
    // (even-even cells, ENV:292-294) and its centre on the action planes (the odd-odd cell of qubit oy d + ox, ENV:309-312)
    memset(T.pix_stab, 255, sizeof(T.pix_stab));
    for (int oy = 0; oy < d; ++oy) for (int ox = 0; ox < d; ++ox) {
        u32 w = 0;
        for (int c = 0; c < 4; ++c) w |= (u32)T.cell_stab[2 * (oy + (c >> 1)) * n + 2 * (ox + (c & 1))] << (8 * c);
        T.pix_stab[oy * d + ox] = w;
    }
    // BFS generators: flipping component `comp` of qubit q toggles these referee-index bits (+ the logical bit)
    const int nh = n_stab / 2;
    for (int comp = 0; comp < 2; ++comp) {
        const int typ = comp == 0 ? 3 : 1;
        for (int x = 0; x < d; ++x) for (int y = 0; y < d; ++y) {
            u32 dl = 0;
            for (s = 0; s < n_stab; ++s)
                if (((T.qubit_smask[x * d + y] >> s) & 1) && T.stab_type[s] == typ) dl |= 1u << ref_bit[s];
            const int logical = comp == 0 ? (y == 0) : (x == 0);                             // FL:312-317
            E->ref_delta[comp][x * d + y] = dl | ((u32)logical << nh);
        }
    }
}

static thread_local char g_err[512] = "";
void dq_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

extern "C" {

int dq_version(void) { return 1; }
const char* dq_last_error(void) { return g_err; }
long dq_struct_size(int id) {
    switch (id) {
        case 0: return (long)sizeof(dq_env_cfg);
        case 1: return (long)sizeof(dq_env_info);
        case 2: return (long)sizeof(dq_sample_job);
        case 3: return (long)sizeof(dq_qnet_cfg);
        case 4: return (long)sizeof(dq_qnet_job);
        case 5: return (long)sizeof(dq_td_job);
        case 6: return (long)sizeof(dq_env_step_job);
        case 7: return (long)sizeof(dq_env_ring);
        default: return -1;
    }
}
int dq_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) { (void)hipGetLastError(); return 0; }
    return n;
}

dq_status dq_env_create(const dq_env_cfg* cfg, dq_env** out) {
    DQ_REQUIRE(cfg && out, DQ_ERR_INVALID, "dq_env_create: null argument");
    *out = nullptr;
    DQ_REQUIRE(cfg->d % 2 == 1, DQ_ERR_INVALID, "for the surface code d must be odd!");     // FL:28-29
    DQ_REQUIRE(cfg->d >= 3 && cfg->d <= 7, DQ_ERR_UNSUPPORTED, "d=%d unsupported: one lattice per 64-lane wavefront needs d*d <= 64", cfg->d);
    DQ_REQUIRE(cfg->error_model == DQ_MODEL_X || cfg->error_model == DQ_MODEL_DP || cfg->error_model == DQ_MODEL_IIDXZ, DQ_ERR_UNSUPPORTED,
               "specified error model not currently supported!");                              // ENV:66-67
    DQ_REQUIRE(cfg->volume_depth >= 1 && cfg->volume_depth <= DQ_MAX_DEPTH, DQ_ERR_UNSUPPORTED, "volume_depth must be in 1..%d", DQ_MAX_DEPTH);
    DQ_REQUIRE(cfg->n_envs >= 1, DQ_ERR_INVALID, "n_envs must be positive");
    dq_env* E = new (std::nothrow) dq_env();
    DQ_REQUIRE(E, DQ_ERR_NOMEM, "out of host memory");
    memset(E, 0, sizeof(*E));
    E->cfg = *cfg;
    const int d = cfg->d, d2 = d * d;
    const int layers = cfg->error_model == DQ_MODEL_X ? 1 : (cfg->use_Y ? 3 : 2);             // ENV:55-65
    E->info.n_action_layers = layers;
    E->info.num_actions = layers * d2 + 1;
    E->info.identity_index = E->info.num_actions - 1;                                         // ENV:69
    E->info.obs_c = cfg->volume_depth + layers;                                               // ENV:78-82
    E->info.obs_h = E->info.obs_w = 2 * d + 1;
    E->info.n_stab = d2 - 1;
    E->info.state_words = EXPORT_FIXED + cfg->volume_depth;
    E->sw = STATE_FIXED + cfg->volume_depth <= 16 ? 16 : 32;
    E->P = (2 * d + 1) * (2 * d + 1);
    build_tables(E);
    hipError_t e = hipMalloc(&E->d_tab, sizeof(EnvTables));
    if (e == hipSuccess) e = hipMemcpy(E->d_tab, &E->h_tab, sizeof(EnvTables), hipMemcpyHostToDevice);
    if (e == hipSuccess) e = hipMalloc(&E->d_state, (size_t)cfg->n_envs * E->sw * sizeof(u64));
    if (e == hipSuccess) e = hipMemset(E->d_state, 0, (size_t)cfg->n_envs * E->sw * sizeof(u64));
    if (e != hipSuccess) {
        dq_set_error("dq_env_create: %s", hipGetErrorString(e));
        dq_env_destroy(E);
        return DQ_ERR_HIP;
    }
    *out = E;
    return DQ_OK;
}

void dq_env_destroy(dq_env* E) {
    if (!E) return;
    if (E->d_tab) (void)hipFree(E->d_tab);
    if (E->d_state) (void)hipFree(E->d_state);
    if (E->d_lut_x) (void)hipFree(E->d_lut_x);
    if (E->d_lut_z) (void)hipFree(E->d_lut_z);
    if (E->d_dec) (void)hipFree(E->d_dec);
    if (E->d_mlp_cells) (void)hipFree(E->d_mlp_cells);
    delete E;
}

dq_status dq_env_get_info(const dq_env* E, dq_env_info* out) {
    DQ_REQUIRE(E && out, DQ_ERR_INVALID, "dq_env_get_info: null argument");
    *out = E->info;
    return DQ_OK;
}

dq_status dq_env_set_rates(dq_env* E, double p_phys, double p_meas) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_env_set_rates: null handle");
    DQ_REQUIRE(p_phys >= 0.0 && p_phys <= 1.0 && p_meas >= 0.0 && p_meas <= 1.0, DQ_ERR_INVALID, "rates must be in [0,1]");
    E->T_phys = dq_rate_threshold(p_phys);
    E->T_meas = dq_rate_threshold(p_meas);
    E->rates_set = true;
    return DQ_OK;
}

// device scratch that is released on every exit path of the referee builders
struct DevScratch {
    void* p = nullptr;
    ~DevScratch() { if (p) (void)hipFree(p); }
};

static dq_status build_one_lut(dq_env* E, int comp, u32** out, hipStream_t st) {
    const int nh = E->info.n_stab / 2;
    const size_t size = (size_t)1 << (nh + 1), half = (size_t)1 << nh;
    DevScratch sd, sc;
    DQ_HIP(hipMalloc(&sd.p, size));
    DQ_HIP(hipMalloc(&sc.p, sizeof(u32)));
    u8* dist = static_cast<u8*>(sd.p);
    u32* changed = static_cast<u32*>(sc.p);
    const size_t words = (half + 31) / 32;
    if (!*out) DQ_HIP(hipMalloc(out, words * sizeof(u32)));
    BfsDeltas dl;
    dl.nq = E->cfg.d * E->cfg.d;
    for (int q = 0; q < 64; ++q) dl.d[q] = q < dl.nq ? E->ref_delta[comp][q] : 0;
    const int blocks = (int)((size + 255) / 256 < 4096 ? (size + 255) / 256 : 4096);
    bfs_fill_kernel<<<blocks, 256, 0, st>>>(dist, size);
    for (int w = 1; w <= dl.nq + 1; ++w) {
        u32 h = 0;
        DQ_HIP(hipMemsetAsync(changed, 0, sizeof(u32), st));
        bfs_level_kernel<<<blocks, 256, 0, st>>>(dist, (u32)size, dl, (u8)w, changed);
        DQ_HIP(hipMemcpyAsync(&h, changed, sizeof(u32), hipMemcpyDeviceToHost, st));
        DQ_HIP(hipStreamSynchronize(st));
        if (!h) break;
    }
    bfs_pack_kernel<<<(int)((words + 255) / 256), 256, 0, st>>>(dist, (u32)half, *out);
    DQ_LAUNCH_CHECK();
    DQ_HIP(hipStreamSynchronize(st));
    return DQ_OK;
}

// ---- maximum-likelihood referee: for one Pauli component with independent flip probability q per qubit, the probability of every
// (syndrome, logical class) state is the XOR-convolution of the single-qubit distributions: one pass per qubit,
//   P'[s] = (1 - q) P[s] + q P[s ^ delta_q],
// over the doubled space the minimum-weight search uses (2^(n+1) states, doubles).  The table predicts class 1 iff
// P[s | class 1] > P[s | class 0] (ties -> class 0).  No contraction: the numpy restatement (oracle/referee.py) must give the same bits.
__global__ void ml_init_kernel(double* p, size_t size) {
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < size; s += (size_t)gridDim.x * blockDim.x) p[s] = s == 0 ? 1.0 : 0.0;
}

__global__ void ml_step_kernel(const double* __restrict__ in, double* __restrict__ out, size_t size, u32 delta, double q) {
#pragma clang fp contract(off)
    for (size_t s = (size_t)blockIdx.x * blockDim.x + threadIdx.x; s < size; s += (size_t)gridDim.x * blockDim.x)
        out[s] = (1.0 - q) * in[s] + q * in[s ^ delta];
}

__global__ void ml_pack_kernel(const double* __restrict__ p, u32 half, u32* __restrict__ lut) {
    const u32 wi = blockIdx.x * blockDim.x + threadIdx.x;
    if (wi >= (half + 31) / 32) return;
    u32 w = 0;
    for (u32 b = 0; b < 32; ++b) {
        const u32 s = wi * 32 + b;
        if (s < half && p[(size_t)half + s] > p[s]) w |= 1u << b;
    }
    lut[wi] = w;
}

static dq_status build_one_ml_lut(dq_env* E, int comp, double q, u32** out, hipStream_t st) {
    const int nh = E->info.n_stab / 2, nq = E->cfg.d * E->cfg.d;
    const size_t size = (size_t)1 << (nh + 1), half = (size_t)1 << nh, words = (half + 31) / 32;
    DevScratch s0, s1;
    DQ_HIP(hipMalloc(&s0.p, size * sizeof(double)));
    DQ_HIP(hipMalloc(&s1.p, size * sizeof(double)));
    double* buf[2] = {static_cast<double*>(s0.p), static_cast<double*>(s1.p)};
    if (!*out) DQ_HIP(hipMalloc(out, words * sizeof(u32)));
    const int blocks = (int)((size + 255) / 256 < 8192 ? (size + 255) / 256 : 8192);
    ml_init_kernel<<<blocks, 256, 0, st>>>(buf[0], size);
    for (int k = 0; k < nq; ++k)
        ml_step_kernel<<<blocks, 256, 0, st>>>(buf[k & 1], buf[(k + 1) & 1], size, E->ref_delta[comp][k], q);
    ml_pack_kernel<<<(int)((words + 255) / 256), 256, 0, st>>>(buf[nq & 1], (u32)half, *out);
    DQ_LAUNCH_CHECK();
    DQ_HIP(hipStreamSynchronize(st));
    return DQ_OK;
}

dq_status dq_env_build_referee_ml(dq_env* E, double q_flip, void* stream) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_env_build_referee_ml: null handle");
    DQ_REQUIRE(q_flip > 0.0 && q_flip < 0.5, DQ_ERR_INVALID, "dq_env_build_referee_ml: the flip probability must be in (0, 0.5)");
    hipStream_t st = (hipStream_t)stream;
    dq_status rc = build_one_ml_lut(E, 0, q_flip, &E->d_lut_x, st);
    if (rc != DQ_OK) return rc;
    rc = build_one_ml_lut(E, 1, q_flip, &E->d_lut_z, st);
    if (rc != DQ_OK) return rc;
    E->lut_x = E->d_lut_x;
    E->lut_z = E->d_lut_z;
    E->lut_joint = nullptr; E->mlp_layers = 0; E->lut_marker = false;
    return DQ_OK;
}

dq_status dq_env_build_referee(dq_env* E, void* stream) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_env_build_referee: null handle");
    hipStream_t st = (hipStream_t)stream;
    dq_status rc = build_one_lut(E, 0, &E->d_lut_x, st);
    if (rc != DQ_OK) return rc;
    rc = build_one_lut(E, 1, &E->d_lut_z, st);
    if (rc != DQ_OK) return rc;
    E->lut_x = E->d_lut_x;
    E->lut_z = E->d_lut_z;
    E->lut_joint = nullptr; E->mlp_layers = 0; E->lut_marker = false;
    return DQ_OK;
}

dq_status dq_env_set_referee(dq_env* E, const uint32_t* lut_x_dev, const uint32_t* lut_z_dev) {
    DQ_REQUIRE(E && lut_x_dev, DQ_ERR_INVALID, "dq_env_set_referee: null argument");
    DQ_REQUIRE(lut_z_dev || E->cfg.error_model == DQ_MODEL_X, DQ_ERR_INVALID, "dq_env_set_referee: the DP model needs a Z table");
    E->lut_x = lut_x_dev;
    E->lut_z = lut_z_dev ? lut_z_dev : lut_x_dev;
    E->lut_joint = nullptr; E->mlp_layers = 0; E->lut_marker = false;
    return DQ_OK;
}

dq_status dq_env_set_referee_joint(dq_env* E, const uint32_t* lut_dev) {
    DQ_REQUIRE(E && lut_dev, DQ_ERR_INVALID, "dq_env_set_referee_joint: null argument");
    DQ_REQUIRE(E->info.n_stab <= 24, DQ_ERR_UNSUPPORTED, "dq_env_set_referee_joint: a table over all %d stabilizers does not fit (d <= 5)",
               E->info.n_stab);
    E->lut_joint = lut_dev; E->mlp_layers = 0; E->lut_marker = false;
    E->lut_x = E->lut_z = lut_dev;                                  // "a referee is installed"; the component tables are not read
    return DQ_OK;
}

dq_status dq_env_set_referee_mlp(dq_env* E, int n_layers, const int32_t* dims, const float* weights_dev) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_env_set_referee_mlp: null handle");
    if (n_layers == 0) {                                            // uninstall: the table referee (if any) is in charge again
        if (E->lut_marker) { E->lut_x = E->lut_z = nullptr; E->lut_marker = false; }
        E->mlp_layers = 0; E->mlp_w = nullptr;
        return DQ_OK;
    }
    DQ_REQUIRE(dims && weights_dev && n_layers >= 1 && n_layers <= DQ_MLP_MAX_LAYERS, DQ_ERR_INVALID, "dq_env_set_referee_mlp: 1 .. %d Dense layers",
               DQ_MLP_MAX_LAYERS);
    const int d = E->cfg.d, ns = E->info.n_stab, classes = E->cfg.error_model == DQ_MODEL_X ? 2 : 4;
    DQ_REQUIRE(dims[0] == (d + 1) * (d + 1), DQ_ERR_INVALID, "dq_env_set_referee_mlp: the first layer takes the flattened (d+1)^2 = %d syndrome, not %d inputs",
               (d + 1) * (d + 1), dims[0]);
    DQ_REQUIRE(dims[n_layers] == classes, DQ_ERR_INVALID, "dq_env_set_referee_mlp: %d homology classes for this error model, the stack ends in %d units",
               classes, dims[n_layers]);
    for (int l = 1; l <= n_layers; ++l) DQ_REQUIRE(dims[l] >= 1 && dims[l] <= 2048, DQ_ERR_UNSUPPORTED, "dq_env_set_referee_mlp: layer widths 1 .. 2048 (two activation vectors per wavefront in 64 KB of LDS)");
    if (!E->d_dec) DQ_HIP(hipMalloc(&E->d_dec, (size_t)E->cfg.n_envs));
    if (!E->d_mlp_cells) {
        // stabilizer s (measurement order, FL:189-221) sits at cell a (d+1) + b of the flattened syndrome (ENV:144 reshape)
        std::vector<std::pair<int, int>> byc;
        const int half = (d + 1) / 2 - 1;
        std::vector<int> sa, sb;
        for (int a = 1; a < d; ++a) for (int b = 1; b < d; ++b) { sa.push_back(a); sb.push_back(b); }
        for (int x = 0; x < half; ++x) { sa.push_back(0); sb.push_back(2 * x + 1); }
        for (int x = 0; x < half; ++x) { sa.push_back(d); sb.push_back(2 * x + 2); }
        for (int x = 0; x < half; ++x) { sa.push_back(2 * x + 2); sb.push_back(0); }
        for (int x = 0; x < half; ++x) { sa.push_back(2 * x + 1); sb.push_back(d); }
        DQ_REQUIRE((int)sa.size() == ns, DQ_ERR_STATE, "dq_env_set_referee_mlp: stabilizer count");
        for (int st = 0; st < ns; ++st) byc.push_back({sa[st] * (d + 1) + sb[st], st});
        std::sort(byc.begin(), byc.end());
        std::vector<int> tab(2 * ns);
        for (int t = 0; t < ns; ++t) { tab[t] = byc[t].second; tab[ns + t] = byc[t].first; }
        DQ_HIP(hipMalloc(&E->d_mlp_cells, tab.size() * sizeof(int)));
        DQ_HIP(hipMemcpy(E->d_mlp_cells, tab.data(), tab.size() * sizeof(int), hipMemcpyHostToDevice));
    }
    E->mlp_layers = n_layers;
    for (int l = 0; l <= n_layers; ++l) E->mlp_dims[l] = dims[l];
    E->mlp_w = weights_dev;
    if (!E->lut_x || E->lut_marker) {                               // "a referee is installed"; the tables are not read
        E->lut_x = E->lut_z = reinterpret_cast<const u32*>(weights_dev);
        E->lut_marker = true;
    }
    return DQ_OK;
}

// the pre-pass of a step whose referee is the Dense stack: classes of the lattices' post-action syndromes into E->d_dec
static dq_status launch_referee_mlp(dq_env* E, const int32_t* action_dev, int auto_reset, hipStream_t st) {
    RefMlpParams r;
    memset(&r, 0, sizeof(r));
    r.state = E->d_state; r.sw = E->sw; r.n_envs = E->cfg.n_envs; r.d2 = E->cfg.d * E->cfg.d; r.n_stab = E->info.n_stab;
    r.n_actions = E->info.num_actions; r.identity = E->info.identity_index; r.model = E->cfg.error_model; r.use_Y = E->cfg.use_Y;
    r.auto_reset = auto_reset; r.tab = E->d_tab; r.action = action_dev; r.cells = E->d_mlp_cells; r.layers = E->mlp_layers; r.w = E->mlp_w;
    r.dec = E->d_dec;
    int mw = 1;
    for (int l = 0; l <= E->mlp_layers; ++l) { r.dims[l] = E->mlp_dims[l]; if (l > 0 && E->mlp_dims[l] > mw) mw = E->mlp_dims[l]; }
    r.max_width = (mw + 3) & ~3;
    const int wpb = DQ_MLP_THREADS / 64;
    referee_mlp_kernel<<<(r.n_envs + wpb - 1) / wpb, DQ_MLP_THREADS, (size_t)wpb * 2 * r.max_width * sizeof(float), st>>>(r);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_env_referee_classes(dq_env* E, const int32_t* action_dev, uint8_t* classes_dev, void* stream) {
    DQ_REQUIRE(E && action_dev && classes_dev, DQ_ERR_INVALID, "dq_env_referee_classes: null argument");
    DQ_REQUIRE(E->mlp_layers, DQ_ERR_STATE, "dq_env_referee_classes: no Dense-stack referee installed (dq_env_set_referee_mlp)");
    const dq_status rc = launch_referee_mlp(E, action_dev, 0, (hipStream_t)stream);
    if (rc != DQ_OK) return rc;
    DQ_HIP(hipMemcpyAsync(classes_dev, E->d_dec, (size_t)E->cfg.n_envs, hipMemcpyDeviceToDevice, (hipStream_t)stream));
    return DQ_OK;
}

dq_status dq_env_get_referee(dq_env* E, uint8_t* lut_x_host, uint8_t* lut_z_host, size_t entries) {
    DQ_REQUIRE(E && E->lut_x, DQ_ERR_STATE, "dq_env_get_referee: no referee installed");
    DQ_REQUIRE(!E->lut_joint, DQ_ERR_STATE, "dq_env_get_referee: a joint table is installed (dq_env_set_referee_joint): it has no component tables");
    const size_t half = (size_t)1 << (E->info.n_stab / 2), words = (half + 31) / 32;
    DQ_REQUIRE(entries == half, DQ_ERR_INVALID, "dq_env_get_referee: expected %zu entries", half);
    u32* tmp = new (std::nothrow) u32[words];
    DQ_REQUIRE(tmp, DQ_ERR_NOMEM, "out of host memory");
    const u32* src[2] = {E->lut_x, E->lut_z};
    u8* dst[2] = {lut_x_host, lut_z_host};
    for (int c = 0; c < 2; ++c) {
        if (!dst[c] || !src[c]) continue;
        hipError_t e = hipMemcpy(tmp, src[c], words * sizeof(u32), hipMemcpyDeviceToHost);
        if (e != hipSuccess) { delete[] tmp; dq_set_error("dq_env_get_referee: %s", hipGetErrorString(e)); return DQ_ERR_HIP; }
        for (size_t s = 0; s < half; ++s) dst[c][s] = (tmp[s >> 5] >> (s & 31)) & 1;
    }
    delete[] tmp;
    return DQ_OK;
}

// epb = lattices per block at one lattice per wave; doubled when the lattices go two to a wave (env_dev.h env_block2: everything of
// a lattice fits 32 lanes; DQ_ENV_PAIR=0 switches it off for A/B runs)
static bool env_pairs(const dq_env* E) {
    static const bool enabled = !(getenv("DQ_ENV_PAIR") && getenv("DQ_ENV_PAIR")[0] == '0');
    return enabled && E->cfg.d * E->cfg.d <= 32 && E->info.n_stab <= 32 && E->sw <= 32;
}

static dq_status fill_common(dq_env* E, EnvParams& p, int epb, bool rider = false, int rider_threads = 512) {
    DQ_REQUIRE(E->rates_set, DQ_ERR_STATE, "dq_env_set_rates has not been called");
    p.tab = E->d_tab; p.state = E->d_state; p.lut_x = E->lut_x; p.lut_z = E->lut_z; p.lut_joint = E->lut_joint;
    p.n_envs = E->cfg.n_envs; p.d2 = E->cfg.d * E->cfg.d; p.n_stab = E->info.n_stab; p.depth = E->cfg.volume_depth;
    p.layers = E->info.n_action_layers; p.n_actions = E->info.num_actions; p.identity = E->info.identity_index;
    p.model = E->cfg.error_model; p.use_Y = E->cfg.use_Y; p.sw = E->sw; p.P = E->P; p.C = E->info.obs_c;
    p.obs_size = E->info.obs_c * E->P;
    p.env_id_base = E->cfg.env_id_base; p.seed0 = E->cfg.seed[0]; p.seed1 = E->cfg.seed[1];
    p.T_phys = E->T_phys; p.T_meas = E->T_meas;
    // two to a wave where the lattice fits half a wave (d <= 5): where the step rides on the dense backward it halves the rounds of environment
    // workgroups (-3.7 us per vector step); the stand-alone launch measured 17.4 us one per wave against 17.8 us two per wave in round 3's first pass
    // (round 3, second pass: two to a wave in the stand-alone launch too -- with the step's loads in one batch and the planes composed from registers it
    // measures 15.9 against 16.5 us per 4096-lattice launch, the acting-only loop 47.3 against 48.9 us per vector step; DQ_ENV_PAIR=1: riders only)
    static const bool riders_only = getenv("DQ_ENV_PAIR") && getenv("DQ_ENV_PAIR")[0] == '1';
    p.pair = (rider || !riders_only) && env_pairs(E) ? 1 : 0;
    if (p.pair) epb *= 2;
    // the component referee tables go into LDS where they are small (d <= 5: 512 bytes each) and a step reads them
    {
        const size_t entries = (size_t)1 << (E->info.n_stab / 2), words = (entries + 31) / 32;
        p.lut_words = (p.mode == 1 && E->lut_x && E->lut_z && !E->lut_joint && !E->lut_marker && !E->mlp_layers && words <= ENV_LUT_LDS_MAX) ? (int)words : 0;
    }
    p.patch = E->patch_next; p.patch_stride = E->patch_stride;     // (armed by dq_env_patch_output for this one launch)
    E->patch_next = nullptr;
    p.env_blocks = (p.n_envs + epb - 1) / epb;
    // the riding step's sampling is drawn by the lattices' own blocks when their threads cover the minibatch (env_dev.h env_inline_sampling)
    if (rider && p.s_batch > 0 && (long long)p.env_blocks * rider_threads >= p.s_batch) p.s_blocks = 0;
    return DQ_OK;
}

// dq_env_patch_output arms ONE launch.  Whatever happens to the call that was meant to consume it -- a failed precondition in front of fill_common, an
// error in a pre-pass -- the arming does not survive that call: a later, unrelated reset / step of the handle must not write d * d words per lattice
// into a ring slot that may have been freed or replaced since.
struct PatchDisarm {
    dq_env* E;
    ~PatchDisarm() { if (E) E->patch_next = nullptr; }
};

static dq_status launch_env(dq_env* E, EnvParams& p, hipStream_t st) {
    const dq_status rc = fill_common(E, p, ENVS_PER_BLOCK);
    if (rc != DQ_OK) return rc;
    dq_launch(DQ_K_ENV, "env_kernel", env_kernel, dim3(p.env_blocks + p.s_blocks), dim3(64 * ENVS_PER_BLOCK),
              env_block_lds((p.pair ? 2 : 1) * ENVS_PER_BLOCK, ENVS_PER_BLOCK, p.obs_size, p.lut_words), st, p);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_env_reset(dq_env* E, const uint8_t* which_dev, uint8_t* obs_dev, uint64_t* legal_dev,
                       uint32_t* lifetime_dev, void* stream) {
    PatchDisarm disarm{E};
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_env_reset: null handle");
    EnvParams p;
    memset(&p, 0, sizeof(p));
    p.mode = 0; p.which = which_dev; p.obs = obs_dev; p.legal = legal_dev; p.lifetime = lifetime_dev;
    return launch_env(E, p, (hipStream_t)stream);
}

dq_status dq_env_step(dq_env* E, const int32_t* action_dev, int auto_reset, uint8_t* obs_dev,
                      float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev, uint32_t* lifetime_dev,
                      uint8_t* was_reset_dev, void* stream) {
    PatchDisarm disarm{E};
    DQ_REQUIRE(E && action_dev, DQ_ERR_INVALID, "dq_env_step: null argument");
    DQ_REQUIRE(E->lut_x, DQ_ERR_STATE, "dq_env_step: no referee installed (dq_env_build_referee / dq_env_set_referee)");
    EnvParams p;
    memset(&p, 0, sizeof(p));
    p.mode = 1; p.auto_reset = auto_reset; p.action = action_dev; p.obs = obs_dev; p.reward = reward_dev;
    p.done = done_dev; p.legal = legal_dev; p.lifetime = lifetime_dev; p.was_reset = was_reset_dev;
    if (E->mlp_layers) {                                            // Dense-stack referee: its classes first (pre-pass), then the step reads them
        const dq_status rc = launch_referee_mlp(E, action_dev, auto_reset, (hipStream_t)stream);
        if (rc != DQ_OK) return rc;
        p.dec_in = E->d_dec;
    }
    return launch_env(E, p, (hipStream_t)stream);
}

// threads: sampling threads per block of the launch that will run the parameters (env_kernel: 256)
static dq_status fill_act_step(dq_env* E, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                               int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev,
                               uint32_t* lifetime_dev, uint8_t* was_reset_dev, const dq_sample_job* sj, int threads, EnvParams& p) {
    DQ_REQUIRE(E && action_dev && seed, DQ_ERR_INVALID, "dq_env_act_step: null argument");
    DQ_REQUIRE(E->lut_x, DQ_ERR_STATE, "dq_env_act_step: no referee installed (dq_env_build_referee / dq_env_set_referee)");
    DQ_REQUIRE(eps >= 0.0 && eps <= 1.0, DQ_ERR_INVALID, "dq_env_act_step: eps must be in [0,1]");
    memset(&p, 0, sizeof(p));
    p.mode = 1; p.auto_reset = auto_reset; p.obs = obs_dev; p.reward = reward_dev;
    p.done = done_dev; p.legal = legal_dev; p.lifetime = lifetime_dev; p.was_reset = was_reset_dev;
    p.policy = 1; p.q = q_dev; p.T_eps = dq_rate_threshold(eps); p.masked_greedy = masked_greedy; p.pseed0 = seed[0]; p.pseed1 = seed[1];
    p.pt = t; p.action_out = action_dev;
    if (sj) {
        DQ_REQUIRE(sj->terminal_ring_dev && sj->index_dev, DQ_ERR_INVALID, "dq_env_act_step_sample: null argument");
        DQ_REQUIRE(sj->n_slots >= 4 && sj->batch >= 1 && sj->head_slot >= 0 && sj->head_slot < sj->n_slots, DQ_ERR_INVALID,
                   "dq_env_act_step_sample: bad sizes");
        DQ_REQUIRE(sj->filled_slots >= DQ_REPLAY_MIN_FILLED && sj->filled_slots <= sj->n_slots, DQ_ERR_STATE,
                   "dq_env_act_step_sample: need at least three complete transitions per lattice");
        DQ_REQUIRE((long long)E->cfg.n_envs * sj->n_slots < (1ll << 31), DQ_ERR_UNSUPPORTED, "dq_env_act_step_sample: ring too large for 32-bit rows");
        p.s_blocks = (sj->batch + threads - 1) / threads; p.s_terminal = sj->terminal_ring_dev; p.s_n_slots = sj->n_slots; p.s_head = sj->head_slot;
        p.s_filled = sj->filled_slots; p.s_batch = sj->batch; p.s_seed0 = sj->seed[0]; p.s_seed1 = sj->seed[1]; p.s_base = sj->sample_base;
        p.s_t = sj->t; p.s_index = sj->index_dev;
    }
    return DQ_OK;
}

static dq_status act_step(dq_env* E, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                          int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev,
                          uint32_t* lifetime_dev, uint8_t* was_reset_dev, const dq_sample_job* sj, void* stream) {
    PatchDisarm disarm{E};
    EnvParams p;
    const dq_status rc = fill_act_step(E, q_dev, eps, masked_greedy, seed, t, action_dev, auto_reset, obs_dev, reward_dev, done_dev, legal_dev,
                                       lifetime_dev, was_reset_dev, sj, 64 * ENVS_PER_BLOCK, p);
    if (rc != DQ_OK) return rc;
    if (E->mlp_layers) {
        // Dense-stack referee: the selection (dq_policy_select: the same rule and Philox stream, on the legal sets the last reset / step
        // left in legal_dev) runs first, then the referee's pre-pass on the selected actions, then the step with both handed in
        DQ_REQUIRE(legal_dev, DQ_ERR_INVALID, "dq_env_act_step: a Dense-stack referee needs legal_dev (the selection reads the legal sets from it)");
        dq_status r2 = dq_policy_select(q_dev, legal_dev, E->cfg.n_envs, E->info.num_actions, eps, masked_greedy, seed, E->cfg.env_id_base, t,
                                        action_dev, stream);
        if (r2 != DQ_OK) return r2;
        r2 = launch_referee_mlp(E, action_dev, auto_reset, (hipStream_t)stream);
        if (r2 != DQ_OK) return r2;
        p.policy = 0; p.action = action_dev; p.dec_in = E->d_dec;
    }
    return launch_env(E, p, (hipStream_t)stream);
}

dq_status dq_env_act_step(dq_env* E, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                          int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev,
                          uint32_t* lifetime_dev, uint8_t* was_reset_dev, void* stream) {
    return act_step(E, q_dev, eps, masked_greedy, seed, t, action_dev, auto_reset, obs_dev, reward_dev, done_dev, legal_dev, lifetime_dev,
                    was_reset_dev, nullptr, stream);
}

dq_status dq_env_act_step_sample(dq_env* E, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                                 int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev,
                                 uint64_t* legal_dev, uint32_t* lifetime_dev, uint8_t* was_reset_dev, const dq_sample_job* sample,
                                 void* stream) {
    DQ_REQUIRE(sample, DQ_ERR_INVALID, "dq_env_act_step_sample: null sampling job");
    return act_step(E, q_dev, eps, masked_greedy, seed, t, action_dev, auto_reset, obs_dev, reward_dev, done_dev, legal_dev, lifetime_dev,
                    was_reset_dev, sample, stream);
}

dq_status dq_env_act_steps(dq_env* E, int n_steps, const uint32_t seed[2], uint64_t t0, const dq_env_ring* ring, int auto_reset, uint64_t* legal_dev,
                           uint32_t* lifetime_dev, uint8_t* was_reset_dev, void* stream) {
    DQ_REQUIRE(E && ring && seed && n_steps >= 1, DQ_ERR_INVALID, "dq_env_act_steps: null argument / no steps");
    DQ_REQUIRE(ring->action_ring_dev && ring->n_slots >= 2 && ring->slot0 >= 0 && ring->slot0 < ring->n_slots, DQ_ERR_INVALID, "dq_env_act_steps: bad ring");
    DQ_REQUIRE(!ring->patch_ring_dev || (ring->patch_stride_words >= E->cfg.d * E->cfg.d && 4 * E->cfg.volume_depth + E->info.n_action_layers <= 32),
               DQ_ERR_INVALID, "dq_env_act_steps: patch words need stride_words >= d * d and at most 32 data bits per pixel");
    const size_t n = (size_t)E->cfg.n_envs, obs_size = (size_t)E->info.obs_c * E->P;
    if (env_pairs(E) && !E->mlp_layers) {
        // ONE launch: the lattices' state stays in registers over the steps, every step's transition goes to its ring slot
        PatchDisarm disarm{E};
        EnvParams p;
        dq_status rc = fill_act_step(E, nullptr, 1.0, 0, seed, t0, ring->action_ring_dev, auto_reset, ring->obs_ring_dev, ring->reward_ring_dev,
                                     ring->done_ring_dev, legal_dev, lifetime_dev, was_reset_dev, nullptr, 64 * ENVS_PER_BLOCK, p);
        if (rc != DQ_OK) return rc;
        rc = fill_common(E, p, ENVS_PER_BLOCK);
        if (rc != DQ_OK) return rc;
        DQ_REQUIRE(p.pair, DQ_ERR_STATE, "dq_env_act_steps: DQ_ENV_PAIR=1 restricts the two-per-wave form to riders");
        p.patch = ring->patch_ring_dev; p.patch_stride = ring->patch_stride_words;
        p.steps = n_steps; p.ring_slots = ring->n_slots; p.ring_slot0 = ring->slot0;
        dq_launch(DQ_K_ENV, "env_multi_kernel", env_multi_kernel, dim3(p.env_blocks), dim3(64 * ENVS_PER_BLOCK),
                  env_block_lds(2 * ENVS_PER_BLOCK, ENVS_PER_BLOCK, p.obs_size, p.lut_words), (hipStream_t)stream, p);
        DQ_LAUNCH_CHECK();
        return DQ_OK;
    }
    // lattices past half a wave (d = 7) / a Dense-stack referee: the same steps as n_steps launches of dq_env_act_step -- the same bits
    for (int s = 0; s < n_steps; ++s) {
        const size_t cur = (size_t)((ring->slot0 + s) % ring->n_slots), nxt = (size_t)((ring->slot0 + s + 1) % ring->n_slots);
        if (ring->patch_ring_dev) {
            const dq_status rc = dq_env_patch_output(E, ring->patch_ring_dev + nxt * n * ring->patch_stride_words, ring->patch_stride_words);
            if (rc != DQ_OK) return rc;
        }
        const dq_status rc = dq_env_act_step(E, nullptr, 1.0, 0, seed, t0 + (uint64_t)s, ring->action_ring_dev + cur * n, auto_reset,
                                             ring->obs_ring_dev ? ring->obs_ring_dev + nxt * n * obs_size : nullptr,
                                             ring->reward_ring_dev ? ring->reward_ring_dev + cur * n : nullptr, ring->done_ring_dev ? ring->done_ring_dev + cur * n : nullptr,
                                             legal_dev, lifetime_dev, was_reset_dev, stream);
        if (rc != DQ_OK) return rc;
    }
    return DQ_OK;
}

dq_status dq_env_patch_output(dq_env* E, uint32_t* patch_dev, int stride_words) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_env_patch_output: null handle");
    DQ_REQUIRE(4 * E->cfg.volume_depth + E->info.n_action_layers <= 32, DQ_ERR_UNSUPPORTED,
               "dq_env_patch_output: 4 * volume_depth + action layers = %d data bits per pixel do not fit one word", 4 * E->cfg.volume_depth + E->info.n_action_layers);
    DQ_REQUIRE(!patch_dev || stride_words >= E->cfg.d * E->cfg.d, DQ_ERR_INVALID, "dq_env_patch_output: stride_words must be at least d * d");
    DQ_REQUIRE((reinterpret_cast<uintptr_t>(patch_dev) & 3) == 0, DQ_ERR_INVALID, "dq_env_patch_output: patch_dev must be 4-byte aligned");
    E->patch_next = patch_dev; E->patch_stride = stride_words;
    return DQ_OK;
}

dq_status dq_env_export_state(dq_env* E, uint64_t* state_dev, void* stream) {
    DQ_REQUIRE(E && state_dev, DQ_ERR_INVALID, "dq_env_export_state: null argument");
    const int n = E->cfg.n_envs;
    env_export_kernel<<<(n + 3) / 4, 256, 0, (hipStream_t)stream>>>(E->d_tab, E->d_state, state_dev, n, E->sw, E->cfg.volume_depth);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_env_import_state(dq_env* E, const uint64_t* state_dev, void* stream) {
    DQ_REQUIRE(E && state_dev, DQ_ERR_INVALID, "dq_env_import_state: null argument");
    const int n = E->cfg.n_envs;
    env_import_kernel<<<(n + 3) / 4, 256, 0, (hipStream_t)stream>>>(E->d_state, state_dev, n, E->sw, E->cfg.volume_depth);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

dq_status dq_env_get_tables(const dq_env* E, uint64_t* stab_qmask, uint64_t* qubit_smask, uint64_t* neigh_qmask, uint8_t* stab_type) {
    DQ_REQUIRE(E, DQ_ERR_INVALID, "dq_env_get_tables: null handle");
    if (stab_qmask) memcpy(stab_qmask, E->h_tab.stab_qmask, sizeof(E->h_tab.stab_qmask));
    if (qubit_smask) memcpy(qubit_smask, E->h_tab.qubit_smask, sizeof(E->h_tab.qubit_smask));
    if (neigh_qmask) memcpy(neigh_qmask, E->h_tab.neigh_qmask, sizeof(E->h_tab.neigh_qmask));
    if (stab_type) memcpy(stab_type, E->h_tab.stab_type, sizeof(E->h_tab.stab_type));
    return DQ_OK;
}

}  // extern "C"

dq_status env_fill_act_step(dq_env* E, const float* q_dev, double eps, int masked_greedy, const uint32_t seed[2], uint64_t t,
                            int32_t* action_dev, int auto_reset, uint8_t* obs_dev, float* reward_dev, uint8_t* done_dev, uint64_t* legal_dev,
                            uint32_t* lifetime_dev, uint8_t* was_reset_dev, const dq_sample_job* sj, uint64_t* stats_dev, EnvParams* p,
                            size_t* lds, int threads) {
    PatchDisarm disarm{E};                                          // (fill_common has copied the pointer into *p by the time this runs on the success path)
    DQ_REQUIRE(E && !E->mlp_layers, DQ_ERR_UNSUPPORTED, "a step whose referee is the Dense stack (dq_env_set_referee_mlp) does not ride: make the separate calls");
    DQ_REQUIRE(threads == 512 || threads == 256, DQ_ERR_INVALID, "env_fill_act_step: 256 or 512 threads per carrying block");
    dq_status rc = fill_act_step(E, q_dev, eps, masked_greedy, seed, t, action_dev, auto_reset, obs_dev, reward_dev, done_dev, legal_dev,
                                 lifetime_dev, was_reset_dev, sj, threads, *p);
    if (rc != DQ_OK) return rc;
    const int waves = threads / 64;
    rc = fill_common(E, *p, waves, true, threads);
    if (rc != DQ_OK) return rc;
    p->stats = reinterpret_cast<unsigned long long*>(stats_dev);
    *lds = env_block_lds(p->pair ? 2 * waves : waves, waves, p->obs_size, p->lut_words);
    return DQ_OK;
}
