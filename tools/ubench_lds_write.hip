// Micro-benchmark: does a 2-way bank conflict of ds_write_b32 cost time on gfx950?  (kernel 2f's table WRITES collide 2-way -- 32
// consecutive lanes span two table rows of stride 32 words -- and the counter shows 16-24 % bank-conflict cycles; the claim is that they are free
// because a ds_write_b32 occupies its issue slot longer than the LDS array needs for two passes.)  Standalone:
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_lds_write.bin tools/ubench_lds_write.hip && tools/ubench_lds_write.bin
// Every block = 512 threads (8 waves), two blocks per CU, ITER x 16 stores per lane; lane -> word address by pattern:
//   0  conflict free: lane l -> word l (64 lanes on 64 banks)
//   1  2-way: lanes 0..31 -> words 0..31, lanes 32..63 -> words 64..95 (the upper half wave lands on the banks of the lower one)
//   2  kernel 2f's shape: lane l -> row (l / 26) of stride 32 words, column l % 26: rows 0 / 1 / 2 overlap on 20 banks
//   3  4-way: lane l -> word (l % 16) + 64 (l / 16)
//   4  ds_write_b64, conflict free (lane l -> 8-byte slot l)
// Prints shader cycles per wave-instruction per CU (all 8 waves issuing) from s_memtime and the wall clock.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 2048;

template <int KIND>
__global__ __launch_bounds__(512) void k(unsigned* out, unsigned seed) {
    __shared__ unsigned lds[8 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    int w;
    if (KIND == 0) w = lane;
    else if (KIND == 1) w = lane < 32 ? lane : 64 + (lane - 32);
    else if (KIND == 2) w = (lane / 26) * 32 + lane % 26;
    else if (KIND == 3) w = (lane % 16) + 64 * (lane / 16);
    else w = 2 * lane;
    unsigned* p = lds + wave * 1024 + w;
    unsigned v = seed + threadIdx.x;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (KIND == 4) asm volatile("ds_write_b64 %0, %1 offset:%2" :: "v"((unsigned)(size_t)p), "v"((unsigned long long)v), "n"(0) : "memory");
            else asm volatile("ds_write_b32 %0, %1" :: "v"((unsigned)(size_t)p), "v"(v) : "memory");
        }
    }
    __syncthreads();
    out[blockIdx.x * 512 + threadIdx.x] = lds[threadIdx.x];
}

template <int KIND>
static void run(const char* name, unsigned* d_out, int n_cu) {
    const int blocks = 2 * n_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(512), 0, 0, d_out, 1u);      // warm
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<KIND>), dim3(blocks), dim3(512), 0, 0, d_out, 1u);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    int khz = 0; CHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double inst_per_cu = 16.0 * ITER * 16;      // 16 waves per CU x ITER x 16 wave-instructions each
    printf("%-46s %8.3f ms  %6.2f ns per wave-instruction per CU  (= %5.2f cycles at the %d MHz nominal clock)\n", name, ms, ms * 1e6 / inst_per_cu,
           ms * 1e-3 * khz * 1e3 / inst_per_cu, khz / 1000);
}

int main() {
    int n_cu = 0; CHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned* d_out; CHK(hipMalloc(&d_out, sizeof(unsigned) * 512 * 2 * n_cu));
    run<0>("ds_write_b32, conflict free", d_out, n_cu);
    run<1>("ds_write_b32, 2-way (half waves collide)", d_out, n_cu);
    run<2>("ds_write_b32, kernel 2f's rows of 26 / stride 32", d_out, n_cu);
    run<3>("ds_write_b32, 4-way", d_out, n_cu);
    run<4>("ds_write_b64, conflict free", d_out, n_cu);
    return 0;
}
