// probe: semantics of v_cvt_scalef32_pk_fp8_f32 on gfx950 (does the scale operand multiply or divide?)
#include <hip/hip_runtime.h>
#include <cstdio>
typedef short v2s __attribute__((ext_vector_type(2)));
__global__ void k(const float* in, unsigned* o) {
    const float a = in[0], b = in[1];
    v2s old = {0, 0};
    o[0] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a, b, 0, false);
    o[1] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, 2.0f, false));
    o[2] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, 0.5f, false));
    o[3] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, a, b, 1.0f, true));
    o[4] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a * 0.5f, b * 0.5f, 0, false);
    o[5] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(a * 2.0f, b * 2.0f, 0, false);
    o[6] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(old, in[2], in[3], 1.0f, false));
    o[7] = (unsigned)__builtin_amdgcn_cvt_pk_fp8_f32(in[2], in[3], 0, false);
}
int main() {
    float h[4] = {1.0f, -3.0f, 500.0f, 0.3f}, *d; unsigned *o, r[8];
    hipMalloc(&d, 16); hipMalloc(&o, 32); hipMemcpy(d, h, 16, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, 1, 1, 0, 0, d, o); hipMemcpy(r, o, 32, hipMemcpyDeviceToHost);
    const char* nm[8] = {"plain(a,b)", "scale 2.0", "scale 0.5", "scale 1.0 hi", "plain(a/2,b/2)", "plain(2a,2b)", "scaled(500,0.3)", "plain(500,0.3)"};
    for (int i = 0; i < 8; ++i) printf("%-16s %08x\n", nm[i], r[i]);
    return 0;
}
