// Developer tool: what would fp8 correction products buy under the power cap?  Per two K-steps of the lattice kernels
// (one 16-row tile, one 16-column tile) today's fp16 hi/lo split issues 6 v_mfma_f32_16x16x32_f16 (hi*hi, lo*hi, hi*lo
// for each K-step); with e4m3 operands for the two correction products they fit ONE v_mfma_scale_f32_16x16x128_f8f6f4
// (tools/sim_fp8_correction.py).  Every SIMD busy, pseudo-random operands.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_mfma_fp8corr.bin tools/ubench_mfma_fp8corr.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));
typedef int intx8_t __attribute__((ext_vector_type(8)));

__device__ inline unsigned lcg(unsigned& s) { s = s * 1664525u + 1013904223u; return s; }
__device__ inline half8_t rnd8(unsigned& s) {
    half8_t v;
    for (int i = 0; i < 8; ++i) v[i] = (_Float16)((float)(int)(lcg(s) >> 8) * (1.0f / 8388608.0f) - 1.0f);
    return v;
}
__device__ inline intx8_t rnd_fp8(unsigned& s) {   // random e4m3 bytes with the exponent kept in the normal range
    intx8_t v;
    for (int i = 0; i < 8; ++i) v[i] = (int)((lcg(s) & 0xBFBFBFBFu) | 0x20202020u);
    return v;
}

template <int MODE>   // 0: 6 fp16 MFMAs per step; 1: 2 fp16 + 1 fp8 (K = 128); 2: 2 fp16 only (the hi*hi floor); 3 / 4: 2 fp16 + 1 fp6 / fp4 (K = 128)
__global__ __launch_bounds__(256) void burn(float* out, int iters) {
    unsigned s = threadIdx.x * 747796405u + blockIdx.x * 2891336453u + 1u;
    half8_t ah[2], al[2], bh[2], bl[2];
    for (int i = 0; i < 2; ++i) { ah[i] = rnd8(s); al[i] = rnd8(s) * (_Float16)0.0005f; bh[i] = rnd8(s); bl[i] = rnd8(s) * (_Float16)0.0005f; }
    intx8_t a8 = rnd_fp8(s), b8 = rnd_fp8(s);
    constexpr int T = 4;   // independent accumulator tiles (row tiles of a wave)
    floatx4_t acc[T];
    for (int q = 0; q < T; ++q) acc[q] = floatx4_t{0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int q = 0; q < T; ++q) {
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bh[k], acc[q], 0, 0, 0);
                if (MODE == 0) {
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[k], bh[k], acc[q], 0, 0, 0);
                    acc[q] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[k], bl[k], acc[q], 0, 0, 0);
                }
            }
            if (MODE == 1) acc[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[q], 0, 0, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            // MODE 3 / 4: the correction product with FP6 (E2M3, format 2) / FP4 (E2M1, format 4) operands -- the same instruction at the rate of the
            // narrower formats (only the TIMING is of interest here: the operand bytes are the e4m3 ones, reinterpreted)
            if (MODE == 3) acc[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[q], 2, 2, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
            if (MODE == 4) acc[q] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[q], 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
        }
    }
    float sum = 0.f;
    for (int q = 0; q < T; ++q) sum += acc[q][0] + acc[q][3];
    out[blockIdx.x * 256 + threadIdx.x] = sum;
}

int main() {
    const int blocks = 256 * 8;
    float* out; hipMalloc((void**)&out, blocks * 256 * 4);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto run = [&](auto k, const char* name, int iters) {
        hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipDeviceSynchronize();
        hipEventRecord(e0);
        for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k, dim3(blocks), dim3(256), 0, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= 5;
        const double steps = (double)iters * 4 /*tiles*/ * blocks * 4 /*waves*/;
        printf("%-46s %8.2f ms   %7.2f ns per (tile, 2 K-steps) per SIMD\n", name, ms, ms * 1e6 / (steps / 1024.0));
    };
    for (int rep = 0; rep < 2; ++rep) {
        run(burn<0>, "6 x f16 16x16x32 (today)", 100000);
        run(burn<1>, "2 x f16 16x16x32 + 1 x fp8 16x16x128", 100000);
        run(burn<2>, "2 x f16 16x16x32 (hi*hi only)", 100000);
        run(burn<3>, "2 x f16 16x16x32 + 1 x fp6 (E2M3) 16x16x128", 100000);
        run(burn<4>, "2 x f16 16x16x32 + 1 x fp4 (E2M1) 16x16x128", 100000);
    }
    return 0;
}
