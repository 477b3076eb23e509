// Kernel experiment harness (developer tool): times field kernels of olx_kernels.hip.h on the
// headline workload (16x16 array, 256^3 grid, single on-axis focus) without the host library, so
// that compile flags / -D macro variants can be A/B-ed in one gpurun call.
//   hipcc -O3 --offload-arch=gfx950 -std=c++17 [-DVARIANT flags] -o exp tools/exp_field.hip
#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <functional>
#include <vector>

#include "../openlifu-python_amd/csrc/olx_kernels.hip.h"

using namespace olx;
#define CHK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); exit(1); } } while (0)

#ifndef EXP_MX
#define EXP_MX 2
#endif
#ifndef EXP_MY
#define EXP_MY 2
#endif
#ifndef EXP_DX
#define EXP_DX EXP_MX
#endif
#ifndef EXP_DY
#define EXP_DY EXP_MY
#endif
#ifndef EXP_NF
#define EXP_NF 1
#endif
#ifndef EXP_FLAGS
#define EXP_FLAGS 3u
#endif
#ifndef EXP_NT
#define EXP_NT 1
#endif
#ifndef EXP_MT
#define EXP_MT 4
#endif
#ifndef EXP_ZPL
#define EXP_ZPL 4
#endif

int main(int argc, char** argv) {
    const int G = argc > 1 ? atoi(argv[1]) : 256;
    const int NE = 16, N = NE * NE, F = EXP_NF;
    const double pitch = 3e-3, h = 0.25e-3 * 256 / G, f0 = 400e3, c0 = 1500.0;
    std::vector<double> pos(3 * N), area(N, 7.29e-6), delays(F * N), apod(F * N, 1.0);
    for (int i = 0; i < N; ++i) {
        pos[i] = ((i / NE) - (NE - 1) / 2.0) * pitch;
        pos[N + i] = -((i % NE) - (NE - 1) / 2.0) * pitch;
        pos[2 * N + i] = 0.0;
    }
    for (int f = 0; f < F; ++f) {
        double fx = 0.001 * f, fy = 0, fz = 0.04, mx = 0;
        std::vector<double> tof(N);
        for (int i = 0; i < N; ++i) {
            tof[i] = std::sqrt((pos[i] - fx) * (pos[i] - fx) + (pos[N + i] - fy) * (pos[N + i] - fy) + fz * fz) / c0;
            mx = std::max(mx, tof[i]);
        }
        for (int i = 0; i < N; ++i) delays[f * N + i] = mx - tof[i];
    }
    // mirror permutations for the centred array
    constexpr int NM = EXP_DX * EXP_DY;
    std::vector<int> perm(NM * N);
    for (int m = 0; m < NM; ++m)
        for (int e = 0; e < N; ++e) {
            int ix = e / NE, iy = e % NE;
            const bool fx = EXP_DX == 2 && (m & 1), fy = EXP_DY == 2 && (EXP_DX == 2 ? (m >> 1) : (m & 1));
            if (fx) ix = NE - 1 - ix;
            if (fy) iy = NE - 1 - iy;
            perm[m * N + e] = ix * NE + iy;
        }
    double *d_pos, *d_area, *d_delays, *d_apod; int* d_perm; float *d_tab, *d_pm, *d_it;
    CHK(hipMalloc(&d_pos, sizeof(double) * 3 * N)); CHK(hipMalloc(&d_area, sizeof(double) * N));
    CHK(hipMalloc(&d_delays, sizeof(double) * F * N)); CHK(hipMalloc(&d_apod, sizeof(double) * F * N));
    CHK(hipMalloc(&d_perm, sizeof(int) * NM * N));
    constexpr int NOUT = NM * EXP_NF, STRIDE = 4 + 2 * NOUT;
    CHK(hipMalloc(&d_tab, sizeof(float) * N * STRIDE));
    const long long vox = (long long)G * G * G;
    CHK(hipMalloc(&d_pm, sizeof(float) * vox * F)); CHK(hipMalloc(&d_it, sizeof(float) * vox * F));
    CHK(hipMemcpy(d_pos, pos.data(), sizeof(double) * 3 * N, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_area, area.data(), sizeof(double) * N, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_delays, delays.data(), sizeof(double) * F * N, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_apod, apod.data(), sizeof(double) * F * N, hipMemcpyHostToDevice));
    CHK(hipMemcpy(d_perm, perm.data(), sizeof(int) * NM * N, hipMemcpyHostToDevice));
    const double ox = EXP_MX == 2 ? 0.0 : -(G - 1) / 2.0 * h, oy = EXP_MY == 2 ? 0.0 : -(G - 1) / 2.0 * h, oz = 5e-3;
    hipLaunchKernelGGL(steer_pack_shared_k, dim3((N + 127) / 128, 1), dim3(128), 0, 0, d_pos, d_area, N, d_delays, d_apod,
                       d_perm, ox, oy, oz, f0, 1e5 / (c0 / f0), f0 / c0, F, EXP_NF, NM, d_tab);
    SharedParams S{};
    S.nx = S.ny = S.nz = G; S.n_el = N; S.x_begin = 0; S.n_foci = F;
    S.hx = S.hy = S.hz = (float)(h * f0 / c0); S.dmin2 = 0.f; S.inten_scale = 1e-4f / 3e6f;
    S.flat_ez = (float)((0.0 - oz) * f0 / c0); S.vox = vox; S.flags = 3u;
    const long long cpr = (G + EXP_ZPL - 1) / EXP_ZPL;
    const long long lanes = (long long)(G - (EXP_MX == 2 ? G / 2 : 0)) * (G - (EXP_MY == 2 ? G / 2 : 0)) * cpr;
    dim3 grid((unsigned)((lanes + FIELD_THREADS - 1) / FIELD_THREADS), 1);
    std::function<void()> launch = [&]() {
        hipLaunchKernelGGL((field_shared_k<EXP_ZPL, EXP_MX, EXP_MY, EXP_DX, EXP_DY, EXP_NF, true, false>), grid, dim3(FIELD_THREADS), 0, 0,
                           d_tab, d_pm, d_it, (float*)nullptr, S);
    };
#ifdef EXP_MFMA
    // kernel 2c on the same workload: 8 columns = EXP_NF2 foci x distinct mirror columns
    constexpr int MT = EXP_MT;
    const int n_pad = (N + 15) / 16 * 16;
    float4* d_coords; uint4* d_bfrag; int* d_colinfo;
    CHK(hipMalloc(&d_coords, sizeof(float4) * n_pad)); CHK(hipMalloc(&d_bfrag, sizeof(uint4) * (n_pad / 16) * 128 * EXP_NT));
    int colinfo[64], tgts[128]; MfmaParams M{};
    for (int o = 0; o < 32; ++o) {   // one column per (focus, distinct mirror column); its targets = the images mapped to it
        const int fl = o / NM, cm = o % NM;
        colinfo[2 * o] = o < NM * EXP_NF ? fl : -1; colinfo[2 * o + 1] = cm;
        int nt_ = 0; for (int q = 0; q < 4; ++q) tgts[4 * o + q] = -1;
        if (o < NM * EXP_NF)
            for (int m = 0; m < EXP_MX * EXP_MY; ++m) {
                const bool fx = EXP_MX == 2 && (m & 1), fy = EXP_MY == 2 && (EXP_MX == 2 ? (m >> 1) : (m & 1));
                const int col = ((EXP_DX == 2 && fx) ? 1 : 0) + EXP_DX * ((EXP_DY == 2 && fy) ? 1 : 0);
                if (col == cm) tgts[4 * o + nt_++] = fl * 4 + m;
            }
    }
    int* d_tgts;
    CHK(hipMalloc(&d_colinfo, sizeof colinfo)); CHK(hipMemcpy(d_colinfo, colinfo, sizeof colinfo, hipMemcpyHostToDevice));
    CHK(hipMalloc(&d_tgts, sizeof tgts)); CHK(hipMemcpy(d_tgts, tgts, sizeof tgts, hipMemcpyHostToDevice));
    const double sg = 16384.0, sw = 1024.0 * 16;
    hipLaunchKernelGGL(mfma_pack_k, dim3(n_pad / 16, 1, EXP_NT), dim3(64), 0, 0, d_pos, d_area, N, n_pad, d_delays, d_apod, d_perm, ox, oy,
                       oz, f0, 1e5 / (c0 / f0) * (f0 / c0) * sw, f0 / c0, F, d_colinfo, d_coords, d_bfrag);
    M.nx = M.ny = M.nz = G; M.n_el_pad = n_pad; M.x_begin = 0; M.n_tiles = 1; M.hx = M.hy = M.hz = S.hx; M.dmin2 = 0.f;
    M.flat_ez = S.flat_ez; M.g_scale = (float)sg; M.out_scale = (float)(1.0 / (sg * sw)); M.inten_scale = S.inten_scale;
    M.vox = vox; M.flags = EXP_FLAGS;
    const long long rpr = (G + MT * 16 - 1) / (MT * 16);
    const long long runs = (long long)(G - (EXP_MX == 2 ? G / 2 : 0)) * (G - (EXP_MY == 2 ? G / 2 : 0)) * rpr;
    dim3 mgrid((unsigned)((runs + 3) / 4), 1);
    launch = [&]() {
        hipLaunchKernelGGL((field_mfma_k<MT, EXP_NT, EXP_MX, EXP_MY, true, false>), mgrid, dim3(FIELD_THREADS), 0, 0, d_coords, d_bfrag,
                           d_pm, d_it, (float*)nullptr, d_tgts, M);
    };
#endif
    for (int i = 0; i < 3; ++i) launch();
    CHK(hipDeviceSynchronize());
    hipEvent_t a, b; CHK(hipEventCreate(&a)); CHK(hipEventCreate(&b));
    const int R = 20;
    CHK(hipEventRecord(a));
    for (int i = 0; i < R; ++i) launch();
    CHK(hipEventRecord(b)); CHK(hipEventSynchronize(b));
    float ms; CHK(hipEventElapsedTime(&ms, a, b)); ms /= R;
    std::vector<float> hp(1024);
    CHK(hipMemcpy(hp.data(), d_pm + ((long long)(G / 2) * G + G / 2) * G, sizeof(float) * std::min(G, 1024), hipMemcpyDeviceToHost));
    double cks = 0; for (int i = 0; i < std::min(G, 1024); ++i) cks += hp[i];
    printf("%s  mx%d my%d nf%d zpl%d grid %d^3: %.4f ms  -> %.3f T pairs/s   (checksum %.6e)\n",
#ifdef EXP_TAG
           EXP_TAG,
#else
           "base",
#endif
           EXP_MX, EXP_MY, EXP_NF, EXP_ZPL, G, ms, (double)vox * N * F / (ms * 1e-3) / 1e12, cks);
    return 0;
}
