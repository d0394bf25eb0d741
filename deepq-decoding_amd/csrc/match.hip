// Matching referee: the exact minimum-weight homology-class decoder for lattices too large for look-up tables,
// as a stand-alone batched decode (dq_match_decode) -- the reference consumes its referee only through
// argmax(static_decoder.predict(true_syndrome)) (/root/reference/example_notebooks/Environments.py:144,150; README.md:278 allows "any
// perfect-measurement decoding algorithm").  Algorithm and LDS layout: match_dev.h; tables: lattice_host.h; numpy restatement:
// oracle/matching_referee.py.  At d <= 7 the predictions equal the look-up referee's for every syndrome (tests).
#include "match_dev.h"
#include "lattice_host.h"

struct dq_match {
    int d, n[2], w10[2];
    u8* dist_dev[2];
    u8* distB_dev[2];
    u32* pool;              // match_dev.h: DQ_MATCH_POOL_SLOTS scratch tables of the clusters beyond DQ_MATCH_MAX_DEFECTS defects, then their lock words
    std::vector<u8> dist[2], distB[2];
};

// one wave per syndrome: defects [batch][2 components][2 words], class = X part + 2 * Z part
__global__ __launch_bounds__(64) void match_decode_kernel(MatchComp cx, MatchComp cz, const u64* __restrict__ defects, int batch, int both,
                                                          u8* __restrict__ cls, u8* __restrict__ inexact) {
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int i = blockIdx.x, lane = threadIdx.x;
    if (i >= batch) return;
    const u64* dp = defects + (size_t)i * 4;
    int flag = 0;
    int c = match_classify(cx, dp[0], dp[1], smem, lane, &flag);
    if (both) c += 2 * match_classify(cz, dp[2], dp[3], smem, lane, &flag);
    if (lane == 0) { cls[i] = (u8)c; if (inexact) inexact[i] = (u8)flag; }
}

void match_comp(const dq_match* M, int comp, MatchComp* out) {
    out->dist = M->dist_dev[comp]; out->distB = M->distB_dev[comp]; out->n = M->n[comp]; out->w10 = M->w10[comp];
    out->pool = M->pool; out->pool_lock = M->pool + ((size_t)DQ_MATCH_POOL_SLOTS << DQ_MATCH_MAX_BIG);
}

// The scratch pool's lock words are re-zeroed on the launch's own stream in front of every launch that may take a slot (dq_match_decode, the wide environment's
// step): a kernel that faulted or was killed while holding one would otherwise leave it taken and every later big-cluster decode on this handle spinning for ever
// (ADVICE r5).  Launches of ONE handle are stream-ordered (the handle is not thread-safe), so no live holder can be wiped.
dq_status match_reset_locks(const dq_match* M, hipStream_t st) {
    if (M && M->pool) DQ_HIP(hipMemsetAsync(M->pool + ((size_t)DQ_MATCH_POOL_SLOTS << DQ_MATCH_MAX_BIG), 0, DQ_MATCH_POOL_SLOTS * sizeof(u32), st));
    return DQ_OK;
}

extern "C" {

void dq_match_destroy(dq_match* M);
dq_status dq_match_create(int d, dq_match** out) {
    DQ_REQUIRE(out, DQ_ERR_INVALID, "dq_match_create: null argument");
    DQ_REQUIRE(d >= 3 && d <= 15 && (d & 1), DQ_ERR_INVALID, "for the surface code d must be odd! (3 <= d <= 15)");
    dq_match* M = new dq_match();
    M->d = d;
    M->pool = nullptr;
    LatticeHost L;
    lattice_build(d, &L);
    for (int comp = 0; comp < 2; ++comp) {
        M->n[comp] = (int)L.typed[comp].size();
        lattice_match_tables(L, comp, &M->dist[comp], &M->distB[comp], &M->w10[comp]);
        M->dist_dev[comp] = nullptr; M->distB_dev[comp] = nullptr;
    }
    for (int comp = 0; comp < 2; ++comp) {
        if (hipMalloc(&M->dist_dev[comp], M->dist[comp].size()) != hipSuccess || hipMalloc(&M->distB_dev[comp], M->distB[comp].size()) != hipSuccess ||
            hipMemcpy(M->dist_dev[comp], M->dist[comp].data(), M->dist[comp].size(), hipMemcpyHostToDevice) != hipSuccess ||
            hipMemcpy(M->distB_dev[comp], M->distB[comp].data(), M->distB[comp].size(), hipMemcpyHostToDevice) != hipSuccess) {
            dq_set_error("dq_match_create: device allocation / upload failed (is a GPU present?)");
            for (int c2 = 0; c2 < 2; ++c2) { (void)hipFree(M->dist_dev[c2]); (void)hipFree(M->distB_dev[c2]); }
            delete M;
            return DQ_ERR_HIP;
        }
    }
    {
        const size_t words = ((size_t)DQ_MATCH_POOL_SLOTS << DQ_MATCH_MAX_BIG) + DQ_MATCH_POOL_SLOTS;
        if (hipMalloc(&M->pool, words * sizeof(u32)) != hipSuccess ||
            hipMemset(M->pool + ((size_t)DQ_MATCH_POOL_SLOTS << DQ_MATCH_MAX_BIG), 0, DQ_MATCH_POOL_SLOTS * sizeof(u32)) != hipSuccess) {
            dq_set_error("dq_match_create: scratch pool allocation failed");
            dq_match_destroy(M);
            return DQ_ERR_HIP;
        }
    }
    static unsigned long long attr_devs = 0;                          // per device (common.h dq_device_bit)
    const unsigned long long dev_bit = dq_device_bit();
    if (!(attr_devs & dev_bit)) {
        const hipError_t ae = hipFuncSetAttribute(reinterpret_cast<const void*>(match_decode_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, DQ_MATCH_LDS);
        if (ae != hipSuccess) {
            dq_set_error("dq_match_create: hipFuncSetAttribute: %s", hipGetErrorString(ae));
            dq_match_destroy(M);
            return DQ_ERR_HIP;
        }
        attr_devs |= dev_bit;
    }
    *out = M;
    return DQ_OK;
}

void dq_match_destroy(dq_match* M) {
    if (!M) return;
    for (int comp = 0; comp < 2; ++comp) { (void)hipFree(M->dist_dev[comp]); (void)hipFree(M->distB_dev[comp]); }
    (void)hipFree(M->pool);
    delete M;
}

dq_status dq_match_info(const dq_match* M, int* nodes_per_component, int* max_defects, int* w10) {
    DQ_REQUIRE(M, DQ_ERR_INVALID, "dq_match_info: null argument");
    if (nodes_per_component) *nodes_per_component = M->n[0];
    if (max_defects) *max_defects = DQ_MATCH_MAX_BIG;          // (defects of one CLUSTER that are matched exactly)
    if (w10) *w10 = M->w10[0];
    return DQ_OK;
}

dq_status dq_match_get_tables(const dq_match* M, int comp, uint8_t* dist_host, uint8_t* distB_host, int* w10) {
    DQ_REQUIRE(M && (comp == 0 || comp == 1), DQ_ERR_INVALID, "dq_match_get_tables: bad argument");
    if (dist_host) memcpy(dist_host, M->dist[comp].data(), M->dist[comp].size());
    if (distB_host) memcpy(distB_host, M->distB[comp].data(), M->distB[comp].size());
    if (w10) *w10 = M->w10[comp];
    return DQ_OK;
}

dq_status dq_match_decode(const dq_match* M, const uint64_t* defects_dev, int batch, int both_components, uint8_t* class_dev,
                          uint8_t* inexact_dev, void* stream) {
    DQ_REQUIRE(M && defects_dev && class_dev, DQ_ERR_INVALID, "dq_match_decode: null argument");
    DQ_REQUIRE(batch >= 1, DQ_ERR_INVALID, "dq_match_decode: batch must be positive");
    MatchComp cx, cz;
    match_comp(M, 0, &cx);
    match_comp(M, 1, &cz);
    { const dq_status rc = match_reset_locks(M, (hipStream_t)stream); if (rc != DQ_OK) return rc; }
    match_decode_kernel<<<batch, 64, DQ_MATCH_LDS, (hipStream_t)stream>>>(cx, cz, defects_dev, batch, both_components, class_dev, inexact_dev);
    DQ_LAUNCH_CHECK();
    return DQ_OK;
}

}  // extern "C"
