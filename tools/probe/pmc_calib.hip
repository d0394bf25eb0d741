// Calibration of rocprofv3's FETCH_SIZE / WRITE_SIZE on gfx950 for THIS library's access widths (MI355X_MICROARCH.md, HBM section: "calibrate on a
// known byte count in your own access pattern"): each kernel moves exactly 64 MiB.
//   hipcc --offload-arch=gfx950 -O3 tools/probe/pmc_calib.hip -o tools/probe/bin/pmc_calib
//   rocprofv3 --pmc FETCH_SIZE --kernel-trace ... -- tools/probe/bin/pmc_calib      (and again with WRITE_SIZE)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
#define N_BYTES (64u << 20)
__global__ void read4(const unsigned* p, unsigned* out, size_t n) {           // 4 bytes per lane, coalesced (the final reduction's loads)
    unsigned s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) s += p[i];
    if (s == 0x12345678u) out[0] = s;
}
__global__ void read16(const u32x4* p, unsigned* out, size_t n) {             // 16 bytes per lane
    unsigned s = 0;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) { const u32x4 v = p[i]; s += v.x ^ v.y ^ v.z ^ v.w; }
    if (s == 0x12345678u) out[0] = s;
}
__global__ void write4(unsigned* p, size_t n) {                               // 4 bytes per lane (the environment's observation rows)
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = (unsigned)i;
}
__global__ void write16(u32x4* p, size_t n) {
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) p[i] = u32x4{(unsigned)i, 1u, 2u, 3u};
}
__global__ void write4_rows847(unsigned char* p, size_t rows) {               // rows of 847 bytes back to back (unaligned starts), written as whole dwords of the
    const size_t total = rows * 847 / 4;                                      // byte stream: one wave = 256 contiguous bytes
    unsigned* q = reinterpret_cast<unsigned*>(p);
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) q[i] = (unsigned)i;
}
__global__ void write4_blocks3388(unsigned char* p, size_t blocks) {           // env_kernel's pattern: workgroup b writes the 3388 bytes (4 observations of 847) at
    for (size_t b = blockIdx.x; b < blocks; b += gridDim.x) {                  // b * 3388 as dwords: its 256-byte wave chunks are not 64-byte aligned
        unsigned* q = reinterpret_cast<unsigned*>(p + b * 3388);
        for (int k = threadIdx.x; k < 847; k += blockDim.x) q[k] = (unsigned)k;
    }
}
int main() {
    void *a, *o;
    hipMalloc(&a, N_BYTES + 4096); hipMalloc(&o, 64); hipMemset(a, 1, N_BYTES);
    for (int rep = 0; rep < 3; ++rep) {
        read4<<<2048, 256>>>((const unsigned*)a, (unsigned*)o, N_BYTES / 4);
        read16<<<2048, 256>>>((const u32x4*)a, (unsigned*)o, N_BYTES / 16);
        write4<<<2048, 256>>>((unsigned*)a, N_BYTES / 4);
        write16<<<2048, 256>>>((u32x4*)a, N_BYTES / 16);
        write4_rows847<<<2048, 256>>>((unsigned char*)a, N_BYTES / 847);
        write4_blocks3388<<<2048, 256>>>((unsigned char*)a, N_BYTES / 3388);
        write4_blocks3388<<<1024, 256>>>((unsigned char*)a, 1024);                // one launch of the environment: 1024 workgroups, 3.47 MB
    }
    hipDeviceSynchronize();
    printf("done: every kernel moves %u bytes\n", N_BYTES);
    return 0;
}
