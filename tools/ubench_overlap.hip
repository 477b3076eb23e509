// Micro-benchmark: does the matrix pipe overlap with a transcendental-heavy VALU stream on gfx950?
// One K-step-like loop body = 16 geometry-term evaluations (the VALU mix of kernel 2c) plus either no
// MFMA, 12 x v_mfma_f32_16x16x32_f16 or 6 x v_mfma_f32_32x32x16_f16 (same MAC count).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
union H8 { half8_t h; unsigned w[4]; };
constexpr int ITER = 2048;

template <int KIND, int VALU>
__global__ __launch_bounds__(256) void k(float* out, float seed) {
    float r2[4], zz[4];
    for (int i = 0; i < 4; ++i) { r2[i] = seed + threadIdx.x * 1e-3f + i; zz[i] = 0.5f + i; }
    f4 acc4[4] = {}; f16v acc16[2] = {};
    H8 bh, bl; for (int i = 0; i < 4; ++i) { bh.w[i] = 0x3c003c00u; bl.w[i] = 0x10001000u; }
    for (int it = 0; it < ITER; ++it) {
        H8 ah[4], al[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (VALU) {
                    const float d2 = r2[q] + zz[t];
                    const float ri = __builtin_amdgcn_rsqf(d2);
                    const float ph = d2 * ri;
                    const float rs = ri * 1024.f;
                    const float gr = rs * __builtin_amdgcn_cosf(ph), gi = rs * __builtin_amdgcn_sinf(ph);
                    const auto hi = __builtin_amdgcn_cvt_pkrtz(gr, gi);
                    const auto lo = __builtin_amdgcn_cvt_pkrtz(gr - (float)hi[0], gi - (float)hi[1]);
                    ah[t].w[q] = __builtin_bit_cast(unsigned, hi); al[t].w[q] = __builtin_bit_cast(unsigned, lo);
                } else { ah[t].w[q] = 0x3c003c00u + it; al[t].w[q] = 0x10001000u; }
            }
            if (KIND == 1) {
                acc4[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t].h, bh.h, acc4[t], 0, 0, 0);
                acc4[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t].h, bh.h, acc4[t], 0, 0, 0);
                acc4[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t].h, bl.h, acc4[t], 0, 0, 0);
            } else if (KIND == 2 && (t & 1)) {
                acc16[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t].h, bh.h, acc16[t >> 1], 0, 0, 0);
                acc16[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al[t].h, bh.h, acc16[t >> 1], 0, 0, 0);
                acc16[t >> 1] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[t - 1].h, bl.h, acc16[t >> 1], 0, 0, 0);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(ah[t].w[q]), "v"(al[t].w[q]));
            }
        }
        r2[it & 3] += 1e-6f;
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) s += acc4[t][0] + acc4[t][3];
    s += acc16[0][0] + acc16[1][5];
    out[blockIdx.x * 256 + threadIdx.x] = s;
}

template <int KIND, int VALU>
void run(const char* name, int wps, float* d) {
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL((k<KIND, VALU>), dim3(256 * wps), dim3(256), 0, 0, d, 0.3f);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<KIND, VALU>), dim3(256 * wps), dim3(256), 0, 0, d, 0.3f);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    printf("%-34s waves/SIMD=%d  %8.3f ms   %7.1f cycles@2.3GHz per loop body per wave-slot\n", name, wps, ms,
           ms * 1e-3 * 2.3e9 / (ITER * wps));
}

int main() {
    float* d; CHK(hipMalloc(&d, sizeof(float) * 256 * 8 * 256));
    for (int w : {1, 2, 4}) {
        if (w == 1) { run<0, 1>("VALU only", 1, d); run<1, 0>("12 x mfma16x16x32 only", 1, d); run<2, 0>("6 x mfma32x32x16 only", 1, d); run<1, 1>("VALU + 12 x 16x16x32", 1, d); run<2, 1>("VALU + 6 x 32x32x16", 1, d); }
        if (w == 2) { run<0, 1>("VALU only", 2, d); run<1, 0>("12 x mfma16x16x32 only", 2, d); run<2, 0>("6 x mfma32x32x16 only", 2, d); run<1, 1>("VALU + 12 x 16x16x32", 2, d); run<2, 1>("VALU + 6 x 32x32x16", 2, d); }
        if (w == 4) { run<0, 1>("VALU only", 4, d); run<1, 0>("12 x mfma16x16x32 only", 4, d); run<2, 0>("6 x mfma32x32x16 only", 4, d); run<1, 1>("VALU + 12 x 16x16x32", 4, d); run<2, 1>("VALU + 6 x 32x32x16", 4, d); }
    }
    return 0;
}
