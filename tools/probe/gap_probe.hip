// Development probe (not part of the product): wall-clock gap between two dependent kernels on one stream, by launch form.
//   hipcc --offload-arch=gfx950 -O2 -o gap_probe gap_probe.hip && ./gap_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#include <algorithm>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s -> %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

struct Big { unsigned long long pad[64]; };   // 512-byte kernarg

template <bool BIG>
__global__ void probe(unsigned long long* ts, int which, float* sink, long write_floats, int spin_us, Big b) {
    extern __shared__ float lds[];
    const unsigned long long t0 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) ts[(which * 2 + 0) * 1024 + blockIdx.x] = t0;
    if (write_floats) {
        const long per = write_floats / gridDim.x;
        float* o = sink + (long)blockIdx.x * per;
        for (long i = threadIdx.x; i < per; i += blockDim.x) o[i] = (float)i;
    }
    if (spin_us) while (__builtin_amdgcn_s_memrealtime() - t0 < (unsigned long long)spin_us * 100) __builtin_amdgcn_s_sleep(8);
    if (BIG && b.pad[3] == 77) lds[threadIdx.x] = 1.f;
    __syncthreads();
    if (threadIdx.x == 0) ts[(which * 2 + 1) * 1024 + blockIdx.x] = __builtin_amdgcn_s_memrealtime();
}

int main() {
    unsigned long long* ts; float* sink;
    CK(hipMalloc(&ts, 4 * 1024 * 8)); CK(hipMalloc(&sink, 64l << 20));
    hipStream_t s1; CK(hipStreamCreateWithFlags(&s1, hipStreamNonBlocking));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<true>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(probe<false>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024));
    Big b; memset(&b, 0, sizeof b);
    struct Case { const char* name; hipStream_t st; int wgs, threads; size_t lds; long wfloats; int spin; };
    Case cases[] = {
        {"null stream, 256x256, no LDS", 0, 256, 256, 0, 0, 0},
        {"own stream,  256x256, no LDS", s1, 256, 256, 0, 0, 0},
        {"own stream,  512x256, 80 KB LDS", s1, 512, 256, 80 * 1024 - 256, 0, 0},
        {"null stream, 512x256, 80 KB LDS", 0, 512, 256, 80 * 1024 - 256, 0, 0},
        {"own stream,  512x256, 80 KB LDS, spin 15us", s1, 512, 256, 80 * 1024 - 256, 0, 15},
        {"null stream, 512x256, 80 KB LDS, spin 15us", 0, 512, 256, 80 * 1024 - 256, 0, 15},
        {"own stream,  512x256, 80 KB LDS, spin 15us, 4 MB written", s1, 512, 256, 80 * 1024 - 256, 1l << 20, 15},
        {"own stream,  512x256, 80 KB LDS, spin 15us, 40 MB written", s1, 512, 256, 80 * 1024 - 256, 10l << 20, 15},
        {"own stream,  256x512, 150 KB LDS, spin 15us", s1, 256, 512, 150 * 1024, 0, 15},
    };
    for (auto& c : cases) {
        std::vector<double> gaps;
        for (int rep = 0; rep < 6; ++rep) {
            for (int k = 0; k < 2; ++k) probe<true><<<c.wgs, c.threads, c.lds, c.st>>>(ts, k, sink, c.wfloats, c.spin, b);
            CK(hipStreamSynchronize(c.st));
            std::vector<unsigned long long> h(4 * 1024);
            CK(hipMemcpy(h.data(), ts, h.size() * 8, hipMemcpyDeviceToHost));
            unsigned long long endA = 0, startB = ~0ull, startA = ~0ull;
            for (int i = 0; i < c.wgs; ++i) { endA = std::max(endA, h[1 * 1024 + i]); startB = std::min(startB, h[2 * 1024 + i]); startA = std::min(startA, h[i]); }
            if (rep) gaps.push_back(((double)startB - (double)endA) / 100.0);
            if (rep == 5) printf("%-62s A lasted %.2f us; ", c.name, (endA - startA) / 100.0);
        }
        std::sort(gaps.begin(), gaps.end());
        printf("gap last-end(A) -> first-start(B): min %.2f median %.2f max %.2f us\n", gaps.front(), gaps[gaps.size() / 2], gaps.back());
    }
    return 0;
}
