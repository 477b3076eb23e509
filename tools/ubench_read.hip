// Read-bandwidth microbenchmark (developer tool): what do the streaming scans of the path (csrc/k_small.hip.h) have to work with?
//   hipcc -O3 --offload-arch=gfx950 -o tools/ubench_read.bin tools/ubench_read.hip && tools/ubench_read.bin
// (a) one stream of 1.07 GB, 16-byte non-temporal loads, grid-stride, several grid sizes; (b) the aggregate's pattern: every lane reads the same
// offset of NV volumes that lie `stride` bytes apart (64 MiB = 256^3 floats: a power of two) -- with and without padding between the volumes;
// (c) read + write of one stream (the scale pass).
#include <hip/hip_runtime.h>
#include <cstdio>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)
typedef float f4 __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(256) void read1_k(const f4* __restrict__ in, size_t n4, float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) s += __builtin_nontemporal_load(in + i);
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = 1.f;
}
template <int NV>
__global__ __launch_bounds__(256) void readnv_k(const f4* __restrict__ in, size_t vol4 /*f4 per volume*/, size_t stride4 /*f4 between volumes*/, float* __restrict__ out) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    f4 s = {0.f, 0.f, 0.f, 0.f};
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < vol4; i += stride) {
#pragma unroll
        for (int v = 0; v < NV; ++v) s += __builtin_nontemporal_load(in + v * stride4 + i);
    }
    if (s.x + s.y + s.z + s.w == 123.456f) out[0] = 1.f;
}
__global__ __launch_bounds__(256) void scale_k(f4* __restrict__ io, size_t n4, float sc) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) { f4 v = __builtin_nontemporal_load(io + i); v *= sc; io[i] = v; }
}

int main() {
    const size_t vox = 256ull * 256 * 256, nv = 16, pad_max = 1 << 20;
    const size_t bytes = (vox * 4 + pad_max) * nv;
    f4* d; float* o; CHK(hipMalloc((void**)&d, bytes)); CHK(hipMalloc((void**)&o, 4)); CHK(hipMemset(d, 0, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* name, double gb) {
        for (int i = 0; i < 5; ++i) launch();
        (void)hipEventRecord(e0);
        const int it = 30;
        for (int i = 0; i < it; ++i) launch();
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms; (void)hipEventElapsedTime(&ms, e0, e1); ms /= it;
        printf("%-64s %8.3f ms  %8.1f GB/s\n", name, ms, gb / (ms * 1e-3));
    };
    const double gb = vox * 4.0 * nv / 1e9;
    char nm[128];
    for (int blocks : {1024, 2048, 4096, 8192, 16384}) {
        snprintf(nm, sizeof nm, "one stream, %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(read1_k, dim3(blocks), dim3(256), 0, 0, d, vox * nv / 4, o); }, nm, gb);
    }
    for (size_t pad : {(size_t)0, (size_t)256, (size_t)4096, (size_t)4096 + 256, (size_t)65536 + 4096 + 256}) {
        for (int blocks : {2048, 8192}) {
            snprintf(nm, sizeof nm, "16 volumes at the same offset, %zu B between volumes, %d blocks", pad, blocks);
            time([&] { hipLaunchKernelGGL(readnv_k<16>, dim3(blocks), dim3(256), 0, 0, d, vox / 4, (vox * 4 + pad) / 16, o); }, nm, gb);
        }
    }
    for (int blocks : {2048, 8192})  {
        snprintf(nm, sizeof nm, "read + write one stream (scale), %d blocks", blocks);
        time([&] { hipLaunchKernelGGL(scale_k, dim3(blocks), dim3(256), 0, 0, d, vox * nv / 4, 1.0f); }, nm, 2 * gb);
    }
    return 0;
}
