// Developer tool: sustained FLOP/s of v_mfma_f32_16x16x32_f16 vs v_mfma_f32_32x32x16_f16 on pseudo-random operands,
// every SIMD busy (the accumulate kernels run at the board's power cap, where operand traffic per flop matters).
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_mfma_shapes.bin tools/ubench_mfma_shapes.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));
typedef float floatx16_t __attribute__((ext_vector_type(16)));

__device__ inline half8_t rnd8(unsigned& s) {
    half8_t v;
    for (int i = 0; i < 8; ++i) { s = s * 1664525u + 1013904223u; v[i] = (_Float16)((float)(int)(s >> 8) * (1.0f / 8388608.0f) - 1.0f); }
    return v;
}

template <int SHAPE>
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
    unsigned s = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 1u;
    half8_t a[4], b[4];
    for (int i = 0; i < 4; ++i) { a[i] = rnd8(s); b[i] = rnd8(s); }
    float acc_sum = 0.f;
    if (SHAPE == 16) {
        floatx4_t acc[8];
        for (int q = 0; q < 8; ++q) acc[q] = floatx4_t{0, 0, 0, 0};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[q & 3], b[(q + 1) & 3], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) acc_sum += acc[q][0] + acc[q][3];
    } else if (SHAPE == 1616) {      // the CDNA3 shape (K = 16): does gfx950 run it in half the time of 16x16x32?
        floatx4_t acc[8];
        for (int q = 0; q < 8; ++q) acc[q] = floatx4_t{0, 0, 0, 0};
        half4_t a4[4], b4[4];
        for (int i = 0; i < 4; ++i) { a4[i] = half4_t{a[i][0], a[i][1], a[i][2], a[i][3]}; b4[i] = half4_t{b[i][0], b[i][1], b[i][2], b[i][3]}; }
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 8; ++q) acc[q] = __builtin_amdgcn_mfma_f32_16x16x16f16(a4[q & 3], b4[(q + 1) & 3], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 8; ++q) acc_sum += acc[q][0] + acc[q][3];
    } else {
        floatx16_t acc[4];
        for (int q = 0; q < 4; ++q) for (int e = 0; e < 16; ++e) acc[q][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[q & 3], b[(q + 1) & 3], acc[q], 0, 0, 0);
        }
        for (int q = 0; q < 4; ++q) acc_sum += acc[q][0] + acc[q][15];
    }
    out[blockIdx.x * 256 + threadIdx.x] = acc_sum;
}

int main() {
    const int blocks = 256 * 8;   // 8 waves per SIMD
    float* out; hipMalloc((void**)&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto k, const char* name, int iters, double flop_per_iter_per_wave) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);   // warm
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double flops = flop_per_iter_per_wave * iters * blocks * 4;
        printf("%-28s %8.2f ms  %8.1f TFLOP/s\n", name, ms, flops / (ms * 1e-3) / 1e12);
    };
    for (int rep = 0; rep < 2; ++rep) {
        run(burn<16>, "16x16x32 f16 (8 acc tiles)", 400000, 8.0 * 16 * 16 * 32 * 2);
        run(burn<32>, "32x32x16 f16 (4 acc tiles)", 400000, 4.0 * 32 * 32 * 16 * 2);
        run(burn<1616>, "16x16x16 f16 (8 acc tiles)", 400000, 8.0 * 16 * 16 * 16 * 2);
    }
    return 0;
}
