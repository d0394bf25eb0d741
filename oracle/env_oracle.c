/*
 * Plain-C CPU restatement of the DeepQ-Decoding environment hot path.
 *
 * TEST INFRASTRUCTURE ONLY (see oracle/__init__.py): used by tests/ as the fast checker
 * for large batches and by bench.py's cpu_baseline leg ("port").  It is never linked into
 * or called by the product library.  PINNED: tests/test_oracle_c.py replays the golden
 * traces captured from the reference (tools/gen_golden.py) through this file.
 *
 * References (relative to /root/reference):
 *   ENV = example_notebooks/Environments.py
 *   FL  = cluster_scripts/d5_dp/Function_Library.py
 *
 * Build: make -C oracle      (gcc -O2 -shared -fPIC -> oracle/_build/libenv_oracle.so)
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

typedef uint64_t u64;
typedef uint32_t u32;
typedef uint8_t u8;

#define MAX_DEPTH 16

/* ------------------------------------------------------------------------------------------
 * Philox4x32-10 (Salmon et al. SC'11); site stream of SURVEY.md §8c / oracle/philox.py
 * ---------------------------------------------------------------------------------------- */
static void philox4x32_10(const u32 ctr[4], const u32 key[2], u32 out[4]) {
    u32 c0 = ctr[0], c1 = ctr[1], c2 = ctr[2], c3 = ctr[3], k0 = key[0], k1 = key[1];
    for (int r = 0; r < 10; ++r) {
        u64 p0 = (u64)0xD2511F53u * c0, p1 = (u64)0xCD9E8D57u * c2;
        u32 n0 = (u32)(p1 >> 32) ^ c1 ^ k0, n1 = (u32)p1, n2 = (u32)(p0 >> 32) ^ c3 ^ k1, n3 = (u32)p0;
        c0 = n0; c1 = n1; c2 = n2; c3 = n3;
        k0 += 0x9E3779B9u; k1 += 0xBB67AE85u;
    }
    out[0] = c0; out[1] = c1; out[2] = c2; out[3] = c3;
}

void dqo_philox(const u32* ctr, const u32* key, u32* out) { philox4x32_10(ctr, key, out); }

/* ------------------------------------------------------------------------------------------
 * Lattice tables (FL:13-51, ENV:326-385) in bit-mask form
 * ---------------------------------------------------------------------------------------- */
static int plaquette_type(int d, int a, int b) {           /* FL:32-35, FL:42-50 */
    if ((a == 0 && b % 2 == 0) || (a == d && b % 2 == 1) || (b == 0 && a % 2 == 1) || (b == d && a % 2 == 0)) return 0;
    return ((a + b) & 1) ? 3 : 1;
}

typedef struct {
    int d, d2, n_stab, n_half;
    int stab_a[64], stab_b[64], stab_type[64], ref_bit[64];
    u64 stab_qmask[64];      /* qubits of stabilizer s                                  */
    u64 qubit_smask[64];     /* live stabilizers touched by qubit q (ENV:262-271)       */
    u64 neigh_qmask[64];     /* 8-neighbourhood of qubit q (ENV:349-372)                */
    u64 col0, row0;          /* FL:312-317                                              */
    int stab_index[17][17];  /* (a,b) -> s or -1                                        */
    u8 static_plane[33 * 33];
} lattice_t;

static void lattice_init(lattice_t* L, int d) {
    memset(L, 0, sizeof(*L));
    L->d = d; L->d2 = d * d; L->n_stab = d * d - 1; L->n_half = (d * d - 1) / 2;
    for (int a = 0; a <= d; ++a) for (int b = 0; b <= d; ++b) L->stab_index[a][b] = -1;
    int s = 0, half = (d + 1) / 2 - 1;
    /* measurement order, FL:189-221 */
    for (int a = 1; a < d; ++a) for (int b = 1; b < d; ++b) { L->stab_a[s] = a; L->stab_b[s] = b; ++s; }
    for (int x = 0; x < half; ++x) { L->stab_a[s] = 0; L->stab_b[s] = 2 * x + 1; ++s; }
    for (int x = 0; x < half; ++x) { L->stab_a[s] = d; L->stab_b[s] = 2 * x + 2; ++s; }
    for (int x = 0; x < half; ++x) { L->stab_a[s] = 2 * x + 2; L->stab_b[s] = 0; ++s; }
    for (int x = 0; x < half; ++x) { L->stab_a[s] = 2 * x + 1; L->stab_b[s] = d; ++s; }
    for (s = 0; s < L->n_stab; ++s) {
        int a = L->stab_a[s], b = L->stab_b[s];
        L->stab_index[a][b] = s;
        L->stab_type[s] = plaquette_type(d, a, b);
        for (int x = a - 1; x <= a; ++x) for (int y = b - 1; y <= b; ++y)
            if (x >= 0 && x < d && y >= 0 && y < d) {
                L->stab_qmask[s] |= 1ull << (x * d + y);
                L->qubit_smask[x * d + y] |= 1ull << s;
            }
    }
    /* referee index bit = rank among the plaquettes of the same type in row-major (a,b) order */
    int rank[4] = {0, 0, 0, 0};
    for (int a = 0; a <= d; ++a) for (int b = 0; b <= d; ++b) {
        int t = plaquette_type(d, a, b);
        if (t) L->ref_bit[L->stab_index[a][b]] = rank[t]++;
    }
    for (int r = 0; r < d; ++r) for (int c = 0; c < d; ++c) {
        for (int dr = -1; dr <= 1; ++dr) for (int dc = -1; dc <= 1; ++dc) {
            int rr = r + dr, cc = c + dc;
            if ((dr || dc) && rr >= 0 && rr < d && cc >= 0 && cc < d) L->neigh_qmask[r * d + c] |= 1ull << (rr * d + cc);
        }
    }
    for (int x = 0; x < d; ++x) L->col0 |= 1ull << (x * d);
    for (int y = 0; y < d; ++y) L->row0 |= 1ull << y;
    int n = 2 * d + 1;
    for (int x = 0; x < n; ++x) for (int y = 0; y < n; ++y) {       /* ENV:284-298 */
        u8 v = 0;
        if ((x == 0 || x == n - 1) && (y & 1)) v = 1;
        if ((y == 0 || y == n - 1) && (x & 1)) v = 1;
        if ((x & 1) && (y & 1) && ((x + y) % 4 == 0)) v = 1;
        L->static_plane[x * n + y] = v;
    }
}

static u64 syndrome_word(const lattice_t* L, u64 xm, u64 zm) {  /* FL:152-174 */
    u64 w = 0;
    for (int s = 0; s < L->n_stab; ++s) {
        u64 comp = (L->stab_type[s] == 3) ? xm : zm;
        w |= (u64)(__builtin_popcountll(comp & L->stab_qmask[s]) & 1) << s;
    }
    return w;
}

static u32 referee_index(const lattice_t* L, u64 w, int typ) {
    u32 idx = 0;
    for (int s = 0; s < L->n_stab; ++s)
        if (L->stab_type[s] == typ && ((w >> s) & 1)) idx |= 1u << L->ref_bit[s];
    return idx;
}

/* ------------------------------------------------------------------------------------------
 * Minimum-weight look-up referee (definition: oracle/referee.py)
 * ---------------------------------------------------------------------------------------- */
int dqo_build_lut(int d, int typ, u8* out) {
    lattice_t L; lattice_init(&L, d);
    int n = L.n_half;
    u32 delta[64];
    for (int x = 0; x < d; ++x) for (int y = 0; y < d; ++y) {
        u64 touched = L.qubit_smask[x * d + y];
        u32 dl = 0;
        for (int s = 0; s < L.n_stab; ++s)
            if (((touched >> s) & 1) && L.stab_type[s] == typ) dl |= 1u << L.ref_bit[s];
        int logical = (typ == 3) ? (y == 0) : (x == 0);
        delta[x * d + y] = dl | ((u32)logical << n);
    }
    size_t size = (size_t)1 << (n + 1);
    u8* dist = (u8*)malloc(size);
    u32* queue = (u32*)malloc(size * sizeof(u32));
    if (!dist || !queue) { free(dist); free(queue); return -1; }
    memset(dist, 255, size);
    size_t head = 0, tail = 0;
    dist[0] = 0; queue[tail++] = 0;
    while (head < tail) {
        u32 s = queue[head++];
        u8 w = dist[s] + 1;
        for (int q = 0; q < L.d2; ++q) {
            u32 t = s ^ delta[q];
            if (dist[t] == 255) { dist[t] = w; queue[tail++] = t; }
        }
    }
    size_t half = (size_t)1 << n;
    for (size_t s = 0; s < half; ++s) out[s] = dist[half + s] < dist[s];
    free(dist); free(queue);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Batched environment
 * ---------------------------------------------------------------------------------------- */
typedef struct {
    u64 xmask, zmask, true_word, summed, acted, round;
    u64 volume[MAX_DEPTH];
    u64 completed[2], legal[2];
    u32 lifetime;
    u8 done;
} env_state;

typedef struct {
    lattice_t L;
    int model, use_Y, depth, layers, n_actions, identity, C, n_envs;
    u32 seed[2], env_id_base;
    u64 T_phys, T_meas;
    const u8 *lut_x, *lut_z;
    env_state* st;
} dqo_env;

static u64 rate_threshold(double p) {   /* W/2^32 < p  <=>  W < ceil(p*2^32) */
    double t = p * 4294967296.0;
    if (t <= 0.0) return 0;
    if (t >= 4294967296.0) return 1ull << 32;
    u64 f = (u64)t;
    return ((double)f < t) ? f + 1 : f;
}

dqo_env* dqo_env_create(int d, int model, int use_Y, int depth, int n_envs, u32 env_id_base, u32 seed0, u32 seed1) {
    if (d < 3 || d > 7 || !(d & 1) || depth < 1 || depth > MAX_DEPTH || n_envs < 1) return NULL;
    dqo_env* E = (dqo_env*)calloc(1, sizeof(dqo_env));
    lattice_init(&E->L, d);
    E->model = model; E->use_Y = use_Y; E->depth = depth; E->n_envs = n_envs;
    E->layers = model == 0 ? 1 : (use_Y ? 3 : 2);                 /* ENV:55-65 */
    E->n_actions = E->layers * d * d + 1;
    E->identity = E->n_actions - 1;                               /* ENV:69 */
    E->C = depth + E->layers;
    E->seed[0] = seed0; E->seed[1] = seed1; E->env_id_base = env_id_base;
    E->st = (env_state*)calloc(n_envs, sizeof(env_state));
    return E;
}

void dqo_env_destroy(dqo_env* E) { if (E) { free(E->st); free(E); } }
void dqo_env_set_rates(dqo_env* E, double p_phys, double p_meas) { E->T_phys = rate_threshold(p_phys); E->T_meas = rate_threshold(p_meas); }
void dqo_env_set_referee(dqo_env* E, const u8* lut_x, const u8* lut_z) { E->lut_x = lut_x; E->lut_z = lut_z; }
int dqo_env_num_actions(const dqo_env* E) { return E->n_actions; }
int dqo_env_obs_size(const dqo_env* E) { int n = 2 * E->L.d + 1; return E->C * n * n; }

static inline void bit_set(u64 m[2], int i) { m[i >> 6] |= 1ull << (i & 63); }
static inline int bit_get(const u64 m[2], int i) { return (int)((m[i >> 6] >> (i & 63)) & 1); }

static void new_volume(dqo_env* E, env_state* S, u32 env_id) {       /* ENV:157-172 == ENV:216-231 */
    const lattice_t* L = &E->L;
    do {
        S->summed = 0;
        for (int j = 0; j < E->depth; ++j) {
            u64 ex = 0, ez = 0, flips = 0;
            for (int lane = 0; lane < L->d2; ++lane) {
                u32 ctr[4] = {(u32)S->round, (u32)(S->round >> 32), env_id, (u32)lane}, w[4];
                philox4x32_10(ctr, E->seed, w);
                if (E->model == 2) {                                  /* IIDXZ, FL:134-160: X flip, then an independent Z flip */
                    if ((u64)w[0] < E->T_phys) ex |= 1ull << lane;
                    if ((u64)w[1] < E->T_phys) ez |= 1ull << lane;
                } else if ((u64)w[0] < E->T_phys) {                   /* FL:99 / FL:119 */
                    int t = E->model == 0 ? 1 : 1 + (int)(((u64)w[1] * 3) >> 32);   /* FL:100 */
                    if (t == 1 || t == 2) ex |= 1ull << lane;
                    if (t == 2 || t == 3) ez |= 1ull << lane;
                }
                if (lane < L->n_stab && (u64)w[2] < E->T_meas) flips |= 1ull << lane;   /* FL:191-221 */
            }
            S->round++;
            S->xmask ^= ex; S->zmask ^= ez;                           /* ENV:164 */
            S->true_word = syndrome_word(L, S->xmask, S->zmask);      /* ENV:165 */
            S->volume[j] = S->true_word ^ flips;                      /* ENV:166 */
            S->summed |= S->volume[j];                                /* ENV:168 */
            S->lifetime++;                                            /* ENV:169 */
        }
    } while (S->summed == 0);                                         /* ENV:171 */
}

static void reset_legal(dqo_env* E, env_state* S) {                   /* ENV:238-258 */
    const lattice_t* L = &E->L;
    S->completed[0] = S->completed[1] = 0; S->acted = 0;
    S->legal[0] = S->legal[1] = 0;
    bit_set(S->legal, E->identity);
    for (int q = 0; q < L->d2; ++q)
        if (L->qubit_smask[q] & S->summed)
            for (int j = 0; j < E->layers; ++j) bit_set(S->legal, q + j * L->d2);
}

static void write_obs(const dqo_env* E, const env_state* S, u8* obs) {
    const lattice_t* L = &E->L;
    int n = 2 * L->d + 1, plane = n * n;
    for (int j = 0; j < E->depth; ++j) {                              /* ENV:174-175, 273-299 */
        u8* p = obs + j * plane;
        memcpy(p, L->static_plane, plane);
        for (int s = 0; s < L->n_stab; ++s) p[2 * L->stab_a[s] * n + 2 * L->stab_b[s]] = (u8)((S->volume[j] >> s) & 1);
    }
    for (int k = 0; k < E->layers; ++k) {                             /* ENV:200-201, 301-314 */
        u8* p = obs + (E->depth + k) * plane;
        memset(p, 0, plane);
        for (int q = 0; q < L->d2; ++q)
            if (bit_get(S->completed, k * L->d2 + q)) p[(2 * (q / L->d) + 1) * n + 2 * (q % L->d) + 1] = 1;
    }
}

static void env_reset_one(dqo_env* E, env_state* S, u32 env_id) {     /* ENV:99-115, 206-235 */
    S->done = 0; S->lifetime = 0; S->xmask = S->zmask = 0; S->true_word = 0;
    new_volume(E, S, env_id);
    reset_legal(E, S);
}

static float env_step_one(dqo_env* E, env_state* S, u32 env_id, int action) {   /* ENV:118-204 */
    const lattice_t* L = &E->L;
    if (action < 0 || action >= E->n_actions) action = E->identity;
    int done_identity = action == E->identity || bit_get(S->completed, action);  /* ENV:131 */
    if (action < E->layers * L->d2) {                                 /* ENV:135-136, FL:243-294 */
        int layer = action / L->d2, q = action % L->d2;
        int pauli = E->model == 0 ? 1 : (E->use_Y ? layer + 1 : (layer == 0 ? 1 : 3));
        if (pauli == 1 || pauli == 2) S->xmask ^= 1ull << q;
        if (pauli == 2 || pauli == 3) S->zmask ^= 1ull << q;
    }
    S->true_word = syndrome_word(L, S->xmask, S->zmask);              /* ENV:139 */
    int X = __builtin_popcountll(S->xmask & L->col0) & 1, Z = __builtin_popcountll(S->zmask & L->row0) & 1;
    int correct = X + 2 * Z;                                          /* ENV:143, FL:296-326 */
    int decoded = E->lut_x[referee_index(L, S->true_word, 3)];        /* ENV:144 */
    if (E->model != 0) decoded += 2 * E->lut_z[referee_index(L, S->true_word, 1)];
    float reward = 0.f;
    if (correct == 0 && S->true_word == 0) reward = 1.f;              /* ENV:148-149 */
    else if (decoded != correct) S->done = 1;                         /* ENV:150-151 */
    if (done_identity) {                                              /* ENV:155-182 */
        new_volume(E, S, env_id);
        reset_legal(E, S);
    } else {                                                          /* ENV:185-196 */
        bit_set(S->completed, action);
        int q = action % L->d2;
        if (!((S->acted >> q) & 1)) {
            S->acted |= 1ull << q;
            for (int j = 0; j < E->layers; ++j)
                for (int nb = 0; nb < L->d2; ++nb)
                    if ((L->neigh_qmask[q] >> nb) & 1) bit_set(S->legal, nb + j * L->d2);
        }
    }
    return reward;
}

static void emit(const dqo_env* E, const env_state* S, int i, u8* obs, u8* done, u64* legal, u32* lifetime) {
    if (obs) write_obs(E, S, obs + (size_t)i * dqo_env_obs_size(E));
    if (done) done[i] = S->done;
    if (legal) { legal[2 * i] = S->legal[0]; legal[2 * i + 1] = S->legal[1]; }
    if (lifetime) lifetime[i] = S->lifetime;
}

/* which == NULL: reset all; else reset env i iff which[i] != 0 (others untouched but still emitted) */
void dqo_env_reset(dqo_env* E, const u8* which, u8* obs, u64* legal, u32* lifetime) {
    for (int i = 0; i < E->n_envs; ++i) {
        if (!which || which[i]) env_reset_one(E, &E->st[i], E->env_id_base + i);
        emit(E, &E->st[i], i, obs, NULL, legal, lifetime);
    }
}

/* auto_reset: an env whose done flag is set when the call starts is reset instead of stepped
 * (its action is ignored, reward 0) -- the keras-rl convention of spending one agent step on
 * the terminal observation.  was_reset (nullable) reports which. */
void dqo_env_step(dqo_env* E, const int32_t* action, int auto_reset, u8* obs, float* reward, u8* done,
                  u64* legal, u32* lifetime, u8* was_reset) {
    for (int i = 0; i < E->n_envs; ++i) {
        env_state* S = &E->st[i];
        float r = 0.f;
        u8 wr = 0;
        if (auto_reset && S->done) { env_reset_one(E, S, E->env_id_base + i); wr = 1; }
        else r = env_step_one(E, S, E->env_id_base + i, action[i]);
        if (reward) reward[i] = r;
        if (was_reset) was_reset[i] = wr;
        emit(E, S, i, obs, done, legal, lifetime);
    }
}

/* state export for tests: per env 8 u64 = {xmask, zmask, true_word, summed, acted, round, completed0, completed1} */
void dqo_env_export(const dqo_env* E, u64* out, u64* volume /* n_envs*depth, nullable */) {
    for (int i = 0; i < E->n_envs; ++i) {
        const env_state* S = &E->st[i];
        u64* o = out + 8 * (size_t)i;
        o[0] = S->xmask; o[1] = S->zmask; o[2] = S->true_word; o[3] = S->summed; o[4] = S->acted; o[5] = S->round;
        o[6] = S->completed[0]; o[7] = S->completed[1];
        if (volume) for (int j = 0; j < E->depth; ++j) volume[(size_t)i * E->depth + j] = S->volume[j];
    }
}

/* test hook: overwrite the hidden state / done flag of env i */
void dqo_env_poke(dqo_env* E, int i, u64 xmask, u64 zmask, int done) {
    E->st[i].xmask = xmask; E->st[i].zmask = zmask; E->st[i].done = (u8)done;
    E->st[i].true_word = syndrome_word(&E->L, xmask, zmask);
}

/* uniform-over-legal policy from STREAM_POLICY (same rule as the device policy kernel):
 * word0 of Philox(key=seed, ctr=(t_lo, t_hi, env_id, 1<<16)); k = (w*n_legal)>>32; k-th set bit */
void dqo_policy_uniform_legal(const dqo_env* E, u64 t, const u64* legal, int32_t* action) {
    for (int i = 0; i < E->n_envs; ++i) {
        u32 ctr[4] = {(u32)t, (u32)(t >> 32), E->env_id_base + i, 1u << 16}, w[4];
        philox4x32_10(ctr, E->seed, w);
        u64 lo = legal[2 * i], hi = legal[2 * i + 1];
        int n = __builtin_popcountll(lo) + __builtin_popcountll(hi);
        int k = (int)(((u64)w[0] * (u64)n) >> 32), a = -1;
        for (int b = 0; b < 128; ++b) {
            int set = b < 64 ? (int)((lo >> b) & 1) : (int)((hi >> (b - 64)) & 1);
            if (set && k-- == 0) { a = b; break; }
        }
        action[i] = a;
    }
}
