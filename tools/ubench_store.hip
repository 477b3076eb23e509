// Store-bandwidth microbenchmark (developer tool): how fast can MI355X absorb the field kernels' output pattern?
//   hipcc -O3 --offload-arch=gfx950 -o /tmp/ubench_store tools/ubench_store.hip && /tmp/ubench_store
// (a) streaming float4 stores; (b) runs of R contiguous bytes, consecutive runs of a block landing in different
// volumes / rows like the lattice kernel's read-out (run r of block b -> volume r % NV, row offset scattered).
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void stream_k(float4* out, size_t n4) {
    const size_t stride = (size_t)gridDim.x * blockDim.x;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) out[i] = make_float4(1.f, 2.f, 3.f, (float)i);
}

// each block writes `runs` runs of RUN4 float4; run q goes to volume (q % nv) at row (block * rows_per_block + q / nv)
template <int RUN4>
__global__ void runs_k(float4* out, size_t vol4, int nv, int rows_total) {
    const int lanes_per_run = RUN4, runs_per_pass = blockDim.x / lanes_per_run;
    const int l = threadIdx.x % lanes_per_run, rp = threadIdx.x / lanes_per_run;
    const int rows_per_block = 8;
    for (int q = rp; q < rows_per_block * nv; q += runs_per_pass) {
        const int v = q % nv, row = q / nv;
        // scatter rows of one block across the volume like (i, j) rows one pitch apart: row * 12 * 256 runs apart
        const size_t grow = ((size_t)blockIdx.x * 1 + (size_t)row * 3072 * 16) % (size_t)rows_total;
        out[(size_t)v * vol4 + grow * RUN4 + l] = make_float4(1.f, 2.f, 3.f, 4.f);
    }
}

int main() {
    const size_t vox = 256ull * 256 * 256, nv = 16;          // 16 volumes of 67 MB = 1.07 GB
    const size_t bytes = vox * 4 * nv;
    float4* d; CHK(hipMalloc((void**)&d, bytes));
    hipEvent_t e0, e1; CHK(hipEventCreate(&e0)); CHK(hipEventCreate(&e1));
    auto time = [&](auto launch, const char* name, double gb) {
        for (int i = 0; i < 3; ++i) launch();
        hipEventRecord(e0);
        const int it = 20;
        for (int i = 0; i < it; ++i) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1); ms /= it;
        printf("%-40s %8.3f ms  %8.1f GB/s\n", name, ms, gb / (ms * 1e-3));
    };
    const double gb = bytes / 1e9;
    time([&] { hipLaunchKernelGGL(stream_k, dim3(256 * 16), dim3(256), 0, 0, d, bytes / 16); }, "streaming float4 (4096 blocks)", gb);
    time([&] { hipLaunchKernelGGL(stream_k, dim3(256 * 64), dim3(256), 0, 0, d, bytes / 16); }, "streaming float4 (16384 blocks)", gb);
    time([&] { hipMemsetAsync(d, 0, bytes, 0); }, "hipMemsetAsync", gb);
    // runs: total runs = bytes / R; each block writes 8 rows x nv volumes
    {
        const int R4 = 4; const size_t rows = vox * 4 / (R4 * 16); const unsigned blocks = (unsigned)(rows / 8);
        time([&] { hipLaunchKernelGGL(runs_k<4>, dim3(blocks), dim3(512), 0, 0, d, vox / 4, (int)nv, (int)rows); }, "64-B runs, 16 volumes", gb);
    }
    {
        const int R4 = 8; const size_t rows = vox * 4 / (R4 * 16); const unsigned blocks = (unsigned)(rows / 8);
        time([&] { hipLaunchKernelGGL(runs_k<8>, dim3(blocks), dim3(512), 0, 0, d, vox / 4, (int)nv, (int)rows); }, "128-B runs, 16 volumes", gb);
    }
    {
        const int R4 = 16; const size_t rows = vox * 4 / (R4 * 16); const unsigned blocks = (unsigned)(rows / 8);
        time([&] { hipLaunchKernelGGL(runs_k<16>, dim3(blocks), dim3(512), 0, 0, d, vox / 4, (int)nv, (int)rows); }, "256-B runs, 16 volumes", gb);
    }
    {
        const int R4 = 64; const size_t rows = vox * 4 / (R4 * 16); const unsigned blocks = (unsigned)(rows / 8);
        time([&] { hipLaunchKernelGGL(runs_k<64>, dim3(blocks), dim3(512), 0, 0, d, vox / 4, (int)nv, (int)rows); }, "1-KiB runs, 16 volumes", gb);
    }
    return 0;
}
