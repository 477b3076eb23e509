// Developer tool: effective shader clock under different instruction mixes (s_memrealtime = 100 MHz reference).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_clock.bin tools/ubench_clock.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256) void burn(float* out, long long* clk, int iters) {
    const long long c0 = clock64(), w0 = wall_clock64();
    floatx4_t acc[4] = {{0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}, {0, 0, 0, 0}};
    half8_t a, b;
    for (int i = 0; i < 8; ++i) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(1.0f - i * 0.01f); }
    float x = threadIdx.x * 1e-3f, y = 0.5f, z = 0.25f;
    for (int it = 0; it < iters; ++it) {
        if (MODE & 1) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[q], 0, 0, 0);
        }
        if (MODE & 2) {
#pragma unroll
            for (int q = 0; q < 8; ++q) { x = fmaf(x, y, z); y = fmaf(y, z, x); }
        }
        if (MODE & 4) {
#pragma unroll
            for (int q = 0; q < 2; ++q) { x = __builtin_amdgcn_sinf(x); y = __builtin_amdgcn_cosf(y); }
        }
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    out[blockIdx.x * 256 + threadIdx.x] = acc[0][0] + acc[1][1] + acc[2][2] + acc[3][3] + x + y;
    if (threadIdx.x == 0) { clk[2 * blockIdx.x] = c1 - c0; clk[2 * blockIdx.x + 1] = w1 - w0; }
}

int main() {
    const int blocks = 256 * 8;
    float* out; long long* clk; hipMalloc((void**)&out, blocks * 256 * 4); hipMalloc((void**)&clk, blocks * 16);
    long long h[2 * 2048];
    auto run = [&](auto k, const char* name, int iters) {
        for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, clk, iters);
        hipDeviceSynchronize();
        hipMemcpy(h, clk, sizeof h, hipMemcpyDeviceToHost);
        double c = 0, w = 0; for (int i = 0; i < blocks; ++i) { c += h[2 * i]; w += h[2 * i + 1]; }
        printf("%-28s shader cycles / 100MHz ticks = %.2f  -> %.0f MHz  (%.0f us per block)\n", name, c / w, c / w * 100.0, w / blocks / 100.0);
    };
    run(burn<1>, "MFMA f16 16x16x32 only", 20000);
    run(burn<2>, "fp32 FMA only", 20000);
    run(burn<4>, "sin/cos only", 20000);
    run(burn<3>, "MFMA + FMA", 20000);
    run(burn<7>, "MFMA + FMA + sin/cos", 10000);
    run(burn<2>, "fp32 FMA only (again)", 20000);
    return 0;
}
