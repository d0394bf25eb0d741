// Probe: does ds_read_b32 / ds_read_b64 at an UNALIGNED byte address return the bytes at that address (SH_MEM_CONFIG alignment mode)?
// hipcc --offload-arch=gfx950 -O3 tools/probe/lds_unaligned.hip -o /tmp/lds_unaligned && /tmp/lds_unaligned
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
__global__ void k(uint32_t* out32, uint64_t* out64, long long* cyc) {
    __shared__ __attribute__((aligned(16))) unsigned char s[1024];
    for (int i = threadIdx.x; i < 1024; i += 64) s[i] = (unsigned char)(i * 7 + 3);
    __syncthreads();
    const uint32_t addr = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(s + threadIdx.x * 3 + 1);
    uint32_t v; uint64_t w;
    asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(v) : "v"(addr));
    asm volatile("ds_read_b64 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(w) : "v"(addr));
    out32[threadIdx.x] = v; out64[threadIdx.x] = w;
    // timing: 64 unaligned dword reads vs 64 aligned
    const uint32_t a0 = (uint32_t)(uintptr_t)(const __attribute__((address_space(3))) void*)(s + (threadIdx.x * 4));
    long long t0 = __builtin_readcyclecounter();
    uint32_t acc = 0;
    for (int r = 0; r < 64; ++r) { uint32_t x; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a0 + (r & 7) * 4)); acc += x; }
    long long t1 = __builtin_readcyclecounter();
    for (int r = 0; r < 64; ++r) { uint32_t x; asm volatile("ds_read_b32 %0, %1\n\ts_waitcnt lgkmcnt(0)" : "=v"(x) : "v"(a0 + (r & 7) * 4 + 1)); acc += x; }
    long long t2 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cyc[0] = t1 - t0; cyc[1] = t2 - t1; out32[64] = acc; }
}
int main() {
    uint32_t* o32; uint64_t* o64; long long* c;
    hipMalloc(&o32, 65 * 4); hipMalloc(&o64, 64 * 8); hipMalloc(&c, 16);
    k<<<1, 64>>>(o32, o64, c);
    uint32_t h32[65]; uint64_t h64[64]; long long hc[2];
    hipMemcpy(h32, o32, 65 * 4, hipMemcpyDeviceToHost); hipMemcpy(h64, o64, 64 * 8, hipMemcpyDeviceToHost); hipMemcpy(hc, c, 16, hipMemcpyDeviceToHost);
    int bad32 = 0, bad64 = 0;
    for (int t = 0; t < 64; ++t) {
        uint64_t e = 0;
        for (int b = 7; b >= 0; --b) e = (e << 8) | (unsigned char)((t * 3 + 1 + b) * 7 + 3);
        if (h32[t] != (uint32_t)e) ++bad32;
        if (h64[t] != e) ++bad64;
    }
    printf("unaligned ds_read_b32: %d of 64 lanes wrong; ds_read_b64: %d wrong; 64 dependent reads aligned %lld cycles, unaligned %lld cycles\n", bad32, bad64, hc[0], hc[1]);
    return 0;
}
