// checks: (1) lane (row=l&15, g=l>>4), byte i of A pairs with lane (col=l&15, g), byte i of B; (2) E8M0 scale semantics; (3) cvt_pk_fp8_f32 = OCP e4m3
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
typedef float floatx4_t __attribute__((ext_vector_type(4)));
typedef int intx8_t __attribute__((ext_vector_type(8)));
__global__ void k(const unsigned char* A, const unsigned char* B, float* D, int sa, int sb, float* cv) {
    const int l = threadIdx.x;
    intx8_t a, b;
    for (int d = 0; d < 8; ++d) { a[d] = ((const int*)A)[l * 8 + d]; b[d] = ((const int*)B)[l * 8 + d]; }
    floatx4_t acc = {0, 0, 0, 0};
    acc = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a, b, acc, 0, 0, 0, sa, 0, sb);
    for (int r = 0; r < 4; ++r) D[l * 4 + r] = acc[r];
    if (l == 0) {
        const float tv[8] = {1.0f, -0.3f, 300.f, 0.001f, 17.f, 448.f, 500.f, 0.0137f};
        for (int q = 0; q < 8; q += 2) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32(tv[q], tv[q + 1], 0, false);
            cv[q] = (float)(w & 0xFF); cv[q + 1] = (float)((w >> 8) & 0xFF);
        }
    }
}
static float dec(unsigned char v) {
    const int s = v >> 7, e = (v >> 3) & 15, m = v & 7;
    float x = e == 0 ? ldexpf((float)m, -9) : ldexpf(1.0f + m / 8.0f, e - 7);
    return s ? -x : x;
}
int main() {
    unsigned char hA[64 * 32], hB[64 * 32];
    srand(5);
    for (int i = 0; i < 64 * 32; ++i) { hA[i] = (rand() & 0xBF) | 0x08; hB[i] = (rand() & 0xBF) | 0x08; if ((hA[i] & 0x7F) == 0x7F) hA[i] = 0x38; if ((hB[i] & 0x7F) == 0x7F) hB[i] = 0x38; }
    unsigned char *dA, *dB; float *dD, *dcv; float hD[256], hcv[8];
    hipMalloc((void**)&dA, sizeof hA); hipMalloc((void**)&dB, sizeof hB); hipMalloc((void**)&dD, sizeof hD); hipMalloc((void**)&dcv, 32);
    hipMemcpy(dA, hA, sizeof hA, hipMemcpyHostToDevice); hipMemcpy(dB, hB, sizeof hB, hipMemcpyHostToDevice);
    for (int pass = 0; pass < 3; ++pass) {
        const int sa = pass == 1 ? 128 : 127, sb = pass == 2 ? 125 : 127;
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, dA, dB, dD, sa, sb, dcv);
        hipMemcpy(hD, dD, sizeof hD, hipMemcpyDeviceToHost); hipMemcpy(hcv, dcv, 32, hipMemcpyDeviceToHost);
        double worst = 0, big = 0;
        for (int row = 0; row < 16; ++row)
            for (int col = 0; col < 16; ++col) {
                double ref = 0;
                for (int g = 0; g < 4; ++g)
                    for (int i = 0; i < 32; ++i) ref += (double)dec(hA[(16 * g + row) * 32 + i]) * dec(hB[(16 * g + col) * 32 + i]);
                ref *= ldexp(1.0, (sa - 127) + (sb - 127));
                const double got = hD[(16 * (row >> 2) + col) * 4 + (row & 3)];
                worst = fmax(worst, fabs(got - ref)); big = fmax(big, fabs(ref));
            }
        printf("scale_a %d scale_b %d: max |got - ref| = %.3e (max |ref| %.3e)\n", sa, sb, worst, big);
    }
    printf("cvt_pk_fp8_f32: ");
    const float tv[8] = {1.0f, -0.3f, 300.f, 0.001f, 17.f, 448.f, 500.f, 0.0137f};
    for (int q = 0; q < 8; ++q) printf("%g->0x%02X(%g) ", tv[q], (int)hcv[q], dec((unsigned char)hcv[q]));
    printf("\n");
    return 0;
}
