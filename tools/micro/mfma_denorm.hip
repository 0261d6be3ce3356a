// Does v_mfma_f32_16x16x32_f16 keep f16 SUBNORMAL inputs (|v| < 2^-14)?  The f16x3 mode stores lo = f16(x - f16(x)), which is subnormal
// for |x| < 2^-3; if the matrix pipe flushed it the mode would lose the low half of every small activation.
//   hipcc --offload-arch=gfx950 -O2 tools/micro/mfma_denorm.hip -o tools/micro/mfma_denorm && tools/micro/mfma_denorm
#include <hip/hip_runtime.h>
#include <hip/hip_fp16.h>
#include <cstdio>
typedef _Float16 f16x8 __attribute__((ext_vector_type(8)));
typedef float f32x4 __attribute__((ext_vector_type(4)));

__global__ void k(float a_val, float b_val, float* out)
{
    f16x8 a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)0.0f; b[i] = (_Float16)0.0f; }
    // one non-zero K position per lane group: element 0 of every lane -> 4 products per output (k = 0, 8, 16, 24)
    a[0] = (_Float16)a_val;
    b[0] = (_Float16)b_val;
    f32x4 c = {0.f, 0.f, 0.f, 0.f};
    c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
    if (threadIdx.x == 0) { out[0] = c[0]; out[1] = (float)a[0]; out[2] = (float)b[0]; }
}

int main()
{
    float* d;
    hipMalloc(&d, 64);
    const float cases[][2] = {{1.0f, 1.0f}, {3.0e-5f, 1.0f}, {6.0e-8f, 1.0f}, {5.96e-8f, 1024.0f}, {3.0e-5f, 3.0e-5f}, {1.0e-6f, 2.0f}};
    for (auto& c : cases) {
        k<<<1, 64>>>(c[0], c[1], d);
        float h[3];
        hipMemcpy(h, d, 12, hipMemcpyDeviceToHost);
        printf("a = %.9g (as f16 %.9g)  b = %.9g (as f16 %.9g)  mfma sum of 4 products = %.9g   expected %.9g\n", c[0], h[1], c[1], h[2], h[0],
               4.0 * (double)h[1] * (double)h[2]);
    }
    return 0;
}
