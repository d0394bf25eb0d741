// What does one SIMD of gfx950 issue per cycle when its waves mix matrix, vector, scalar and LDS instructions?  (Round 5: conv_wave_kernel runs four
// waves per SIMD and its time did not move when 12 % of its VALU and 15 % of its SALU instructions were removed.)
// GPU box:  hipcc --offload-arch=gfx950 -O3 tools/probe/issue_mix.hip -o /tmp/mix && /tmp/mix
//
// Every wave runs `iters` trips of a body of NM v_mfma_f32_16x16x32_f16 (four rotating accumulators), NV v_fma_f32 (eight independent registers),
// NS s_add_u32 and NL ds_read_b128 (lane-linear, conflict-free), spread evenly over the body (one MFMA, then its share of the others), pinned by
// inline assembly.  Workgroups of 256 / 512 / 1024 threads = 1 / 2 / 4 waves per SIMD, one workgroup per CU, 256 workgroups.
// Printed: shader cycles per trip (slowest wave of a workgroup, mean over workgroups).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));

template <int NM, int NV, int NS, int NL, int NN>
__global__ void k(int iters, float* out, unsigned long long* cyc) {
    __shared__ u32x4 buf[1024];
    const int lane = threadIdx.x & 63;
    f32x4 c[4];
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) c[i][e] = 0.f;
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)(0.001f * (lane + e)); b[e] = (_Float16)(0.002f * (lane - e)); }
    float r[8];
    for (int e = 0; e < 8; ++e) r[e] = 0.5f + 0.01f * lane + e;
    const float s = 0.999f;
    unsigned sr = 1;
    buf[threadIdx.x] = u32x4{1u, 2u, 3u, (unsigned)threadIdx.x};
    const unsigned laddr = (unsigned)(threadIdx.x * 16);
    u32x4 lv[4] = {};
    __syncthreads();
    constexpr int STEPS = NM > 0 ? NM : 16;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < STEPS; ++i) {
            if (NM > 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(c[i & 3]) : "v"(a), "v"(b));
#pragma unroll
            for (int q = (i * NV) / STEPS; q < ((i + 1) * NV) / STEPS; ++q) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(r[q & 7]) : "v"(s));
#pragma unroll
            for (int q = (i * NS) / STEPS; q < ((i + 1) * NS) / STEPS; ++q) asm volatile("s_add_u32 %0, %0, 1" : "+s"(sr));
#pragma unroll
            for (int q = (i * NL) / STEPS; q < ((i + 1) * NL) / STEPS; ++q) asm volatile("ds_read_b128 %0, %1" : "=v"(lv[q & 3]) : "v"(laddr));
#pragma unroll
            for (int q = (i * NN) / STEPS; q < ((i + 1) * NN) / STEPS; ++q) asm volatile("s_nop 0");
        }
        if (NL > 0) asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    float acc = sr * 1e-9f;
    for (int i = 0; i < 4; ++i) for (int e = 0; e < 4; ++e) acc += c[i][e];
    for (int e = 0; e < 8; ++e) acc += r[e];
    for (int i = 0; i < 4; ++i) acc += (float)lv[i][0];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (lane == 0) cyc[blockIdx.x * 16 + (threadIdx.x >> 6)] = t1 - t0;
}

template <int NM, int NV, int NS, int NL, int NN>
static void run(const char* name) {
    const int iters = 200, wgs = 256;
    float* out; unsigned long long* cyc;
    hipMalloc(&out, wgs * 1024 * 4); hipMalloc(&cyc, wgs * 16 * 8);
    printf("%-28s", name);
    for (int threads = 256; threads <= 1024; threads *= 2) {
        hipMemset(cyc, 0, wgs * 16 * 8);
        k<NM, NV, NS, NL, NN><<<wgs, threads>>>(iters, out, cyc);
        k<NM, NV, NS, NL, NN><<<wgs, threads>>>(iters, out, cyc);
        hipDeviceSynchronize();
        std::vector<unsigned long long> h(wgs * 16);
        hipMemcpy(h.data(), cyc, wgs * 16 * 8, hipMemcpyDeviceToHost);
        double sum = 0;
        for (int g = 0; g < wgs; ++g) { unsigned long long m = 0; for (int w = 0; w < threads / 64; ++w) m = std::max(m, h[g * 16 + w]); sum += (double)m; }
        printf("  %d w/SIMD: %8.0f", threads / 256, sum / wgs / iters);
    }
    printf("   cycles per trip\n");
    hipFree(out); hipFree(cyc);
}

int main() {
    printf("per trip and wave: M = v_mfma_f32_16x16x32_f16, V = v_fma_f32, S = s_add_u32, L = ds_read_b128, N = s_nop 0\n");
    run<48, 0, 0, 0, 0>("M48");
    run<0, 96, 0, 0, 0>("V96");
    run<0, 0, 96, 0, 0>("S96");
    run<0, 0, 0, 48, 0>("L48");
    run<0, 0, 0, 0, 96>("N96");
    run<48, 96, 0, 0, 0>("M48 V96");
    run<48, 0, 96, 0, 0>("M48 S96");
    run<48, 0, 0, 48, 0>("M48 L48");
    run<48, 0, 0, 0, 96>("M48 N96");
    run<0, 96, 96, 0, 0>("V96 S96");
    run<0, 96, 0, 48, 0>("V96 L48");
    run<48, 96, 96, 0, 0>("M48 V96 S96");
    run<48, 96, 96, 48, 0>("M48 V96 S96 L48");
    run<48, 96, 96, 48, 96>("M48 V96 S96 L48 N96");
    run<48, 192, 0, 0, 0>("M48 V192");
    run<48, 48, 0, 0, 0>("M48 V48");
    return 0;
}
