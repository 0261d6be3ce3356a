// Do 64-byte pieces cost HBM bandwidth?  out = a + b over float32-sized "pre-split" rows, two ways of handing the bytes to the lanes:
//   full    every wavefront instruction covers 64 lanes x 16 B = 1 KB of CONTIGUOUS bytes (what an elementwise kernel does)
//   pieces  the epilogue pattern of the convolution kernels: lane (row = lane & 15, q = lane >> 4) takes the 16 bytes at
//           row * pitch + q * 16 with one instruction and the 16 bytes 64 further on with the next -- per instruction 16 rows x 64 B,
//           every 128-byte line touched by two different instructions
// Same bytes, same arithmetic.   build: hipcc -O3 --offload-arch=gfx950 tools/micro/partial_line.hip -o gpurun_out/partial_line
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef __attribute__((ext_vector_type(4))) float f32x4;
#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__global__ __launch_bounds__(256) void add_full(const f32x4* a, const f32x4* b, f32x4* o, size_t n16)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += (size_t)gridDim.x * blockDim.x) o[i] = a[i] + b[i];
}

// rows of `pitch` bytes (a multiple of 128); a wavefront owns 16 rows x 128 bytes at a time
__global__ __launch_bounds__(256) void add_pieces(const char* a, const char* b, char* o, size_t rows, int pitch)
{
    const int lane = threadIdx.x & 63, r = lane & 15, q = lane >> 4;
    const size_t waves = (size_t)gridDim.x * (blockDim.x / 64), wave = (size_t)blockIdx.x * (blockDim.x / 64) + threadIdx.x / 64;
    const int lines = pitch / 128;
    const size_t units = (rows / 16) * lines;                       // (16 rows, one 128-byte column) per step
    for (size_t u = wave; u < units; u += waves) {
        const size_t rb = (u / lines) * 16, col = (u % lines) * 128;
        const size_t off = (rb + r) * (size_t)pitch + col + q * 16;
        const f32x4 a0 = *(const f32x4*)(a + off), a1 = *(const f32x4*)(a + off + 64);
        const f32x4 b0 = *(const f32x4*)(b + off), b1 = *(const f32x4*)(b + off + 64);
        *(f32x4*)(o + off) = a0 + b0;
        *(f32x4*)(o + off + 64) = a1 + b1;
    }
}

int main()
{
    const int pitch = 2048;                                          // 512 float32-sized channels per row (res3 branch2c's output)
    const size_t rows = 68136 / 16 * 16, bytes = rows * pitch;       // 139.5 MB per array
    char *a, *b, *o, *flush;
    CHECK(hipMalloc(&a, bytes)); CHECK(hipMalloc(&b, bytes)); CHECK(hipMalloc(&o, bytes)); CHECK(hipMalloc(&flush, 600 << 20));
    CHECK(hipMemset(a, 1, bytes)); CHECK(hipMemset(b, 2, bytes));
    hipEvent_t e0, e1;
    CHECK(hipEventCreate(&e0)); CHECK(hipEventCreate(&e1));
    for (int cold = 0; cold < 2; ++cold)
        for (int which = 0; which < 2; ++which) {
            float best = 1e30f;
            for (int rep = 0; rep < 7; ++rep) {
                if (cold) CHECK(hipMemset(flush, rep, 600 << 20));
                CHECK(hipEventRecord(e0));
                const int n = cold ? 1 : 10;
                for (int i = 0; i < n; ++i) {
                    if (which == 0) add_full<<<256 * 8, 256>>>((const f32x4*)a, (const f32x4*)b, (f32x4*)o, bytes / 16);
                    else add_pieces<<<256 * 8, 256>>>(a, b, o, rows, pitch);
                }
                CHECK(hipEventRecord(e1)); CHECK(hipEventSynchronize(e1));
                float ms; CHECK(hipEventElapsedTime(&ms, e0, e1));
                if (rep && ms / n < best) best = ms / n;
            }
            printf("%-6s %-4s: %7.1f us = %.2f TB/s (3 x %.1f MB)\n", which ? "pieces" : "full", cold ? "cold" : "hot", best * 1e3, 3.0 * bytes / best / 1e9, bytes / 1e6);
        }
    return 0;
}
