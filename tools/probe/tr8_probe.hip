// ds_read_b64_tr_b8 lane map probe (GPU box): hipcc --offload-arch=gfx950 tools/probe/tr8_probe.hip -o tools/probe/tr/tr8_probe
// Result: within a 16-lane group, lanes 2e, 2e+1 point at the two 8-byte halves of row e (e = 0 .. 7; addresses are aligned DOWN to 8 bytes); lane i
// receives byte i of each of the eight rows (element e = row e).
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x2 __attribute__((ext_vector_type(2)));
__global__ void k(unsigned* o, int mode) {
    __shared__ __attribute__((aligned(16))) unsigned char sm[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) sm[i] = (unsigned char)(i & 255);
    __syncthreads();
    const int lane = threadIdx.x;
    // mode 0: lane l points at sm + 8 l (contiguous);  mode 1: lanes 2r, 2r+1 of a 16-lane group at row r = 100 r (+ 8 (l & 1)), groups 1000 apart
    const unsigned char* p = mode == 0 ? sm + 8 * lane : sm + 1000 * (lane >> 4) + 100 * ((lane & 15) >> 1) + 8 * (lane & 1);
    i32x2 v = __builtin_amdgcn_ds_read_tr8_b64_v2i32((__attribute__((address_space(3))) i32x2*)p);
    o[2 * lane] = v[0]; o[2 * lane + 1] = v[1];
}
int main() {
    unsigned* d; (void)hipMalloc(&d, 64 * 2 * 4); unsigned h[128];
    for (int mode = 0; mode < 2; ++mode) {
        k<<<1, 64>>>(d, mode); (void)hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("mode %d\n", mode);
        for (int l = 0; l < 64; l += (l < 18 ? 1 : 15)) {
            printf("lane %d:", l);
            for (int e = 0; e < 8; ++e) printf(" %u", (h[2 * l + (e >> 2)] >> (8 * (e & 3))) & 255);
            printf("\n");
        }
    }
    return 0;
}
