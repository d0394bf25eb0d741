// Live per-kernel timing with HIP events on the stream the kernel is launched on (bench.py's roofline object:
// "achieved = algorithmic work per launch / that kernel's average launch duration, measured live").
// One kernel family is armed at a time; every launch of it carries an event pair (dq_launch, common.h: the dispatch's own timestamps) or, for the
// small per-layer kernels, is bracketed by one (dq_prof_begin / dq_prof_end) until the pool is full.
// Nothing is recorded (and nothing costs anything beyond one compare) while no kernel is armed.
#include "common.h"
#include <cstdlib>
#include <vector>

static const char* const kNames[DQ_K_COUNT] = {
    "env_kernel", "policy_kernel", "conv_chain_kernel", "dense_chain_kernel", "gemm_fwd_kernel", "gemm_wgrad_kernel",
    "reduce_partials_kernel", "td_kernels", "adam_kernel", "dense_bwd_chain_kernel", "dense_wgrad_kernel", "conv_bwd_chain_kernel",
};

static const char* g_symbol[DQ_K_COUNT];    // the symbol each family's LAST launch used (string literals of the launch sites)
void dq_prof_note_symbol(int id, const char* symbol) { if (id >= 0 && id < DQ_K_COUNT) g_symbol[id] = symbol; }

static int g_armed = -1;
static int g_used = 0;
static int g_stride = 1, g_seen = 0;          // every g_stride-th launch of the armed family is timed
static std::vector<hipEvent_t> g_events;      // 2 per launch

static bool g_open = false;                    // a bracket is open (dq_prof_begin recorded its first event)

void dq_prof_begin(int id, hipStream_t st) {
    g_open = false;
    if (id != g_armed || 2 * g_used + 1 >= (int)g_events.size()) return;
    if (g_seen++ % g_stride) return;
    (void)hipEventRecord(g_events[2 * g_used], st);
    g_open = true;
}

void dq_prof_end(int id, hipStream_t st) {
    if (id != g_armed || !g_open) return;
    (void)hipEventRecord(g_events[2 * g_used + 1], st);
    ++g_used;
    g_open = false;
}

int dq_prof_pair(int id, hipEvent_t* start, hipEvent_t* stop) {
    static const bool bracket = getenv("DQ_PROF_BRACKET") && getenv("DQ_PROF_BRACKET")[0] == '1';
    if (id != g_armed) return 0;
    if (bracket) return 2;
    if (2 * g_used + 1 >= (int)g_events.size() || g_seen++ % g_stride) return 0;
    *start = g_events[2 * g_used]; *stop = g_events[2 * g_used + 1];
    ++g_used;
    return 1;
}

static void release_events() {
    for (hipEvent_t e : g_events) (void)hipEventDestroy(e);
    g_events.clear();
    g_used = 0;
}

extern "C" {

int dq_prof_kernel_count(void) { return DQ_K_COUNT; }

const char* dq_prof_kernel_name(int id) { return id >= 0 && id < DQ_K_COUNT ? kNames[id] : ""; }

const char* dq_prof_kernel_symbol(int id) { return id >= 0 && id < DQ_K_COUNT && g_symbol[id] ? g_symbol[id] : ""; }

dq_status dq_prof_stride(int stride) {
    DQ_REQUIRE(stride >= 1, DQ_ERR_INVALID, "dq_prof_stride: stride must be >= 1");
    g_stride = stride; g_seen = 0;
    return DQ_OK;
}

dq_status dq_prof_arm(int kernel_id, int max_launches) {
    release_events();
    g_armed = -1;
    g_stride = 1; g_seen = 0; g_open = false;
    if (kernel_id < 0) return DQ_OK;
    DQ_REQUIRE(kernel_id < DQ_K_COUNT && max_launches > 0, DQ_ERR_INVALID, "dq_prof_arm: bad kernel id / capacity");
    g_events.resize(2 * (size_t)max_launches);
    for (size_t i = 0; i < g_events.size(); ++i) {
        hipError_t e = hipEventCreate(&g_events[i]);
        if (e != hipSuccess) {
            g_events.resize(i);
            release_events();
            dq_set_error("dq_prof_arm: hipEventCreate: %s", hipGetErrorString(e));
            return DQ_ERR_HIP;
        }
    }
    g_armed = kernel_id;
    return DQ_OK;
}

dq_status dq_prof_collect_spread(int* launches, double* total_ms, double* min_ms, double* max_ms) {
    DQ_REQUIRE(launches && total_ms, DQ_ERR_INVALID, "dq_prof_collect: null argument");
    *launches = 0;
    *total_ms = 0.0;
    double lo = 0.0, hi = 0.0;
    if (g_used > 0) DQ_HIP(hipEventSynchronize(g_events[2 * g_used - 1]));
    for (int i = 0; i < g_used; ++i) {
        float ms = 0.f;
        DQ_HIP(hipEventElapsedTime(&ms, g_events[2 * i], g_events[2 * i + 1]));
        *total_ms += ms;
        if (i == 0 || ms < lo) lo = ms;
        if (i == 0 || ms > hi) hi = ms;
    }
    if (min_ms) *min_ms = lo;
    if (max_ms) *max_ms = hi;
    *launches = g_used;
    g_used = 0;
    return DQ_OK;
}

dq_status dq_prof_collect(int* launches, double* total_ms) { return dq_prof_collect_spread(launches, total_ms, nullptr, nullptr); }

}  // extern "C"
