// Does v_mfma_f32_16x16x32_f16 honour f16 subnormal inputs on gfx950?  (development aid; DESIGN.md section 4, "f16x2")
//   hipcc --offload-arch=gfx950 -O2 -o f16_denorm f16_denorm.hip && ./f16_denorm
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
__global__ void k(float a_val, float b_val, float* out) {
    f16x8 a, b;
    for (int e = 0; e < 8; ++e) { a[e] = (_Float16)a_val; b[e] = (_Float16)b_val; }
    f32x4 acc = {0.f, 0.f, 0.f, 0.f};
    acc = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = acc[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}
int main() {
    float* d; hipMalloc(&d, 16);
    const float cases[][2] = {{1.f, 1.f}, {3e-5f, 1024.f}, {1e-6f, 1024.f}, {5.96e-8f, 4096.f}, {3e-5f, 3e-5f}};
    for (auto& c : cases) {
        k<<<1, 64>>>(c[0], c[1], d);
        float h[3]; hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a=%g (f16 %g) b=%g (f16 %g): mfma sum over K=32 = %.9g, expected %.9g\n", c[0], h[1], c[1], h[2], h[0], 32.0 * (double)h[1] * (double)h[2]);
    }
    return 0;
}
