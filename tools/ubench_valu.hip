// Micro-benchmark: VALU / transcendental issue rates on gfx950, the ceiling of the
// per-pair pressure-field kernel.  Standalone:
//   hipcc -O3 --offload-arch=gfx950 -o ubench_valu tools/ubench_valu.hip && ./ubench_valu
// Each kernel runs ITER iterations of 16 independent instructions per lane on every SIMD
// (grid = 256 CUs x 8 blocks x 256 threads -> 8 waves/SIMD); prints cycles per
// wave-instruction per SIMD derived from wall time and the measured shader clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

constexpr int ITER = 4096;

#define REP16(OP) OP(0) OP(1) OP(2) OP(3) OP(4) OP(5) OP(6) OP(7) OP(8) OP(9) OP(10) OP(11) OP(12) OP(13) OP(14) OP(15)

template <int KIND>
__global__ __launch_bounds__(256) void k(float* out, float seed, unsigned long long* clk) {
    float v[16];
#pragma unroll
    for (int i = 0; i < 16; ++i) v[i] = seed + threadIdx.x * 1e-3f + i;
    float2 p[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) p[i] = make_float2(v[2 * i], v[2 * i + 1]);
    unsigned long long t0 = __builtin_amdgcn_s_memtime();
    unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < ITER; ++it) {
        if (KIND == 0) {  // v_fma_f32
#define OP(i) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if (KIND == 1) {  // v_sin_f32
#define OP(i) asm volatile("v_sin_f32 %0, %0" : "+v"(v[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 2) {  // v_rsq_f32
#define OP(i) asm volatile("v_rsq_f32 %0, %0" : "+v"(v[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 3) {  // v_pk_fma_f32 (8 x 2 floats)
#define OP(i) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
            REP16(OP)
#undef OP
        } else if (KIND == 4) {  // interleaved 8 fma + 8 sin
#define OP(i) if ((i) & 1) asm volatile("v_sin_f32 %0, %0" : "+v"(v[i])); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if (KIND == 5) {  // the kernel's mix per pair: 6 plain + 3 trans (x2 -> 18; use 12 plain + 4... see host)
            // 10 plain + 6 trans  (ratio 5:3)
#define OP(i) if ((i) % 8 < 3) asm volatile("v_cos_f32 %0, %0" : "+v"(v[i])); else asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(v[i]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if (KIND == 6) {  // v_mul_f32
#define OP(i) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(v[i]) : "v"(seed));
            REP16(OP)
#undef OP
        } else if (KIND == 7) {  // v_exp_f32
#define OP(i) asm volatile("v_exp_f32 %0, %0" : "+v"(v[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 8) {  // v_fract_f32
#define OP(i) asm volatile("v_fract_f32 %0, %0" : "+v"(v[i]));
            REP16(OP)
#undef OP
        } else if (KIND == 9) {  // v_pk_mul_f32
#define OP(i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(p[i & 7]) : "v"(p[(i + 1) & 7]));
            REP16(OP)
#undef OP
        }
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime();
    unsigned long long r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 16; ++i) s += v[i];
#pragma unroll
    for (int i = 0; i < 8; ++i) s += p[i].x + p[i].y;
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { clk[0] = t1 - t0; clk[1] = r1 - r0; }
}

template <int KIND>
void run(const char* name, int waves_per_simd, float* d_out, unsigned long long* d_clk) {
    const int blocks = 256 * waves_per_simd;  // 256 threads = 4 waves = 1 wave per SIMD per block
    hipEvent_t a, b;
    CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.37f, d_clk);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(a));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL(k<KIND>, dim3(blocks), dim3(256), 0, 0, d_out, 0.37f, d_clk);
    CHK(hipEventRecord(b));
    CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= 5;
    unsigned long long clk[2];
    CHK(hipMemcpy(clk, d_clk, sizeof clk, hipMemcpyDeviceToHost));
    const double ghz = (double)clk[0] / ((double)clk[1] * 10.0);  // memrealtime = 100 MHz
    // wave-instructions per SIMD = waves_per_simd * ITER * 16
    const double winstr = (double)waves_per_simd * ITER * 16;
    const double cyc = ms * 1e-3 * ghz * 1e9 / winstr;
    printf("%-28s waves/SIMD=%d  %8.3f ms  clk=%.3f GHz  cycles per wave-instr per SIMD = %.2f\n", name,
           waves_per_simd, ms, ghz, cyc);
}

int main() {
    float* d_out; unsigned long long* d_clk;
    CHK(hipMalloc(&d_out, sizeof(float) * 256 * 8 * 256));
    CHK(hipMalloc(&d_clk, 16));
    for (int w : {1, 2, 4, 8}) {
        run<0>("v_fma_f32", w, d_out, d_clk);
        run<6>("v_mul_f32", w, d_out, d_clk);
        run<3>("v_pk_fma_f32", w, d_out, d_clk);
        run<9>("v_pk_mul_f32", w, d_out, d_clk);
        run<1>("v_sin_f32", w, d_out, d_clk);
        run<2>("v_rsq_f32", w, d_out, d_clk);
        run<7>("v_exp_f32", w, d_out, d_clk);
        run<8>("v_fract_f32", w, d_out, d_clk);
        run<4>("8 fma + 8 sin interleaved", w, d_out, d_clk);
        run<5>("10 fma + 6 cos (5:3)", w, d_out, d_clk);
    }
    return 0;
}
