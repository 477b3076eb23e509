// Micro-benchmark: issue cost of the conversion / packing instructions considered for the
// hi+lo operand split of kernel 2c (gfx950).  Same method as ubench_valu.hip.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 4096;
#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed, unsigned long long* clk) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
#define OPA(i) asm volatile(INSTR : "+v"(v[i]) : "v"(v[(i + 1) & 15]));
        if (KIND == 0) {
#define INSTR "v_cvt_pkrtz_f16_f32 %0, %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 1) {
#define INSTR "v_cvt_pk_bf16_f32 %0, %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 2) {
#define INSTR "v_fma_mix_f32 %0, %0, %1, %0 op_sel_hi:[0,0,1]"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 3) {
#define INSTR "v_cvt_f32_f16 %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 4) {
#define INSTR "v_lshlrev_b32 %0, 16, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 5) {
#define INSTR "v_and_b32 %0, 0xffff0000, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 6) {
#define INSTR "v_perm_b32 %0, %0, %1, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 7) {
#define INSTR "v_sub_f32 %0, %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 8) {
#define INSTR "v_ldexp_f32 %0, %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 9) {
#define INSTR "v_cvt_pk_f16_f32 %0, %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 10) {
#define INSTR "v_pack_b32_f16 %0, %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 11) {
#define INSTR "v_cvt_f16_f32 %0, %1"
            REP16(OPA)
#undef INSTR
        } else if (KIND == 12) {
#define INSTR "v_bfi_b32 %0, %0, %1, %1"
            REP16(OPA)
#undef INSTR
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, float* d_out, unsigned long long* d_clk) {
    const int w = 8, blocks = 256 * w;
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.37f, d_clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.37f, d_clk);
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    unsigned long long clk[2]; CHK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);
    printf("%-24s %8.3f ms clk=%.3f GHz  cycles per wave-instr per SIMD = %.2f\n", name, ms, ghz, ms * 1e-3 * ghz * 1e9 / ((double)w * ITER * 16));
}

int main() {
    float* d_out; unsigned long long* d_clk;
    CHK(hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256)); CHK(hipMalloc(&d_clk, 16));
    run<0>("v_cvt_pkrtz_f16_f32", d_out, d_clk);
    run<9>("v_cvt_pk_f16_f32", d_out, d_clk);
    run<1>("v_cvt_pk_bf16_f32", d_out, d_clk);
    run<11>("v_cvt_f16_f32", d_out, d_clk);
    run<2>("v_fma_mix_f32", d_out, d_clk);
    run<3>("v_cvt_f32_f16", d_out, d_clk);
    run<4>("v_lshlrev_b32", d_out, d_clk);
    run<5>("v_and_b32", d_out, d_clk);
    run<6>("v_perm_b32", d_out, d_clk);
    run<12>("v_bfi_b32", d_out, d_clk);
    run<10>("v_pack_b32_f16", d_out, d_clk);
    run<7>("v_sub_f32", d_out, d_clk);
    run<8>("v_ldexp_f32", d_out, d_clk);
    return 0;
}
