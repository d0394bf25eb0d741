// Does TRAPSTS.EXCP accumulate f32 -> f16 conversion overflow (sticky, per wave, without enabling traps) on gfx950?  (round 6: the forward's range guard)
// hipcc --offload-arch=gfx950 -O2 tools/probe/trapsts_probe.hip -o /tmp/trapsts_probe && /tmp/trapsts_probe
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 f16x2 __attribute__((ext_vector_type(2)));
__global__ void probe(const float* in, unsigned* out, unsigned* bits) {
    const int t = threadIdx.x + blockIdx.x * blockDim.x;
    unsigned before, after;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(before));
    const float x = in[t];
    const f16x2 h = {(_Float16)x, (_Float16)(x * 0.5f)};
    out[t] = __builtin_bit_cast(unsigned, h);
    asm volatile("s_nop 4\n\ts_getreg_b32 %0, hwreg(HW_REG_TRAPSTS, 0, 9)" : "=s"(after) : "v"(out[t]));
    if ((threadIdx.x & 63) == 0) { bits[2 * (t >> 6)] = before; bits[2 * (t >> 6) + 1] = after; }
}
int main() {
    const int n = 256;                     // 4 waves: wave 0 small values, wave 1 one lane at 70000, wave 2 one lane NaN, wave 3 denormal-range
    float h[n];
    for (int i = 0; i < n; ++i) h[i] = 1.0f + i;
    h[64 + 17] = 70000.f;
    h[128 + 3] = __builtin_nanf("");
    for (int i = 192; i < 256; ++i) h[i] = 1e-9f;
    float* d; unsigned *o, *b;
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, n * 4); hipMalloc(&b, 8 * 4);
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    probe<<<1, n>>>(d, o, b);
    unsigned hb[8], ho[n];
    hipMemcpy(hb, b, sizeof(hb), hipMemcpyDeviceToHost); hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const char* what[4] = {"finite", "one lane 70000", "one lane NaN", "1e-9 (underflow)"};
    for (int w = 0; w < 4; ++w) printf("wave %d (%s): EXCP before 0x%03x after 0x%03x  (bit 0 invalid, 3 overflow, 4 underflow, 5 inexact)\n", w, what[w], hb[2 * w], hb[2 * w + 1]);
    printf("h(70000) = 0x%04x\n", ho[64 + 17] & 0xffff);
    return 0;
}
