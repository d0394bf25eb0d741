// ds_read_b64_tr_b16 lane map probe (GPU box): hipcc --offload-arch=gfx950 tools/probe/tr_probe.hip -o tools/probe/tr/tr_probe
// Result: within a 16-lane group, lanes 4r .. 4r+3 point at the four 4-half segments of row r (r = 0 .. 3, any 8-byte-aligned addresses); lane i
// receives half i of each of the four rows (element r = row r).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short s16x4 __attribute__((ext_vector_type(4)));
__global__ void k(unsigned* o, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned short sm[4096];
    for (int i = threadIdx.x; i < 4096; i += 64) sm[i] = i;
    __syncthreads();
    const int lane = threadIdx.x;
    // mode 0: standard; mode 1: rows scattered: row r of group g at element offset 100*(r + 4 g) (+ 4 (i & 3))
    const unsigned short* p = mode == 0 ? sm + (lane >> 4) * 64 + (lane & 15) * 4
                                        : sm + 100 * (((lane & 15) >> 2) + 4 * (lane >> 4)) + 4 * (lane & 3);
    s16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4*)p);
    for (int e = 0; e < 4; ++e) o[4 * lane + e] = (unsigned short)v[e];
}
int main() {
    unsigned* d; hipMalloc(&d, 64 * 4 * 4); unsigned h[256];
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; l += (l < 18 ? 1 : 15)) printf("lane %d: %u %u %u %u\n", l, h[4*l], h[4*l+1], h[4*l+2], h[4*l+3]);
    }
    return 0;
}
