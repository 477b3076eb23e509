// Micro-benchmark: LDS read issue rate per CU on gfx950 (ds_read_b32 / b64 / b128, conflict free, 16 waves per CU all reading) -- the other resource of
// kernel 2g's K loop: every MFMA tile and K-step reads 4 x ds_read_b64 of geometry fragments per lane, every K-step 4 x ds_read_b128 of steering fragments.
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_lds_read.bin tools/ubench_lds_read.hip && tools/ubench_lds_read.bin
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)
constexpr int ITER = 2048;

template <int BYTES>
__global__ __launch_bounds__(512) void k(unsigned* out) {
    __shared__ __attribute__((aligned(16))) unsigned lds[8 * 1024];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 8 * 1024; i += 512) lds[i] = i;
    __syncthreads();
    const unsigned addr = (unsigned)(size_t)(lds + wave * 1024) + lane * BYTES;
    unsigned acc = 0;
    for (int it = 0; it < ITER; ++it) {
#pragma unroll
        for (int u = 0; u < 16; ++u) {
            if (BYTES == 4) { unsigned v; asm volatile("ds_read_b32 %0, %1" : "=v"(v) : "v"(addr) : "memory"); acc ^= v; }
            else if (BYTES == 8) { unsigned long long v; asm volatile("ds_read_b64 %0, %1" : "=v"(v) : "v"(addr) : "memory"); acc ^= (unsigned)v; }
            else { uint4 v; asm volatile("ds_read_b128 %0, %1" : "=v"(v) : "v"(addr) : "memory"); acc ^= v.x; }
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    out[blockIdx.x * 512 + threadIdx.x] = acc;
}

template <int BYTES>
static void run(const char* name, unsigned* d_out, int n_cu) {
    const int blocks = 2 * n_cu;
    hipEvent_t e0, e1;
    CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    hipLaunchKernelGGL((k<BYTES>), dim3(blocks), dim3(512), 0, 0, d_out);
    CHK(hipDeviceSynchronize());
    CHK(hipEventRecord(e0));
    for (int r = 0; r < 5; ++r) hipLaunchKernelGGL((k<BYTES>), dim3(blocks), dim3(512), 0, 0, d_out);
    CHK(hipEventRecord(e1)); CHK(hipEventSynchronize(e1));
    float ms = 0; CHK(hipEventElapsedTime(&ms, e0, e1));
    ms /= 5;
    int khz = 0; CHK(hipDeviceGetAttribute(&khz, hipDeviceAttributeClockRate, 0));
    const double inst_per_cu = 16.0 * ITER * 16;
    const double cyc = ms * 1e-3 * khz * 1e3 / inst_per_cu;
    printf("%-30s %8.3f ms  %5.2f cycles per wave-instruction per CU at the %d MHz nominal clock = %6.1f bytes per cycle per CU\n", name, ms, cyc, khz / 1000, 64.0 * BYTES / cyc);
}

int main() {
    int n_cu = 0; CHK(hipDeviceGetAttribute(&n_cu, hipDeviceAttributeMultiprocessorCount, 0));
    unsigned* d_out; CHK(hipMalloc(&d_out, sizeof(unsigned) * 512 * 2 * n_cu));
    run<4>("ds_read_b32, conflict free", d_out, n_cu);
    run<8>("ds_read_b64, conflict free", d_out, n_cu);
    run<16>("ds_read_b128, conflict free", d_out, n_cu);
    return 0;
}
