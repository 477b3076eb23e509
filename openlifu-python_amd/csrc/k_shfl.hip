// kernel 2s (field_shfl_k): ELEMENTS ACROSS LANES with wavefront-level __shfl reductions -- BASELINE north_star's stated
// design for the accumulate, built to evidence the choice against the one-voxel-per-lane kernel 2a with rocprof
// (profiles/r02_shfl_vs_accum.*; OLX_FIELD_VARIANT=shfl selects it, the planner never does).
// gfx950 (CDNA4, wave64) only.
#ifdef OLX_AB_VARIANTS   // measured-slower A/B form: compiled only into the developer library (build.py -DOLX_AB_VARIANTS), never into libolx.so
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// Work map: a wave owns 64 consecutive z voxels of one grid row.  Lane l holds up to EP = 4 elements (e = 64 p + l) in
// registers -- coalesced 32-byte loads of the packed table, once per chunk of 256 elements -- and evaluates ITS elements'
// terms for one voxel at a time (the voxel's coordinates are wave-uniform); the 64 partial sums meet in a butterfly of six
// __shfl_xor steps per component, and the lane whose index equals the voxel's keeps the total, so the results leave as one
// coalesced 256-byte store per output.  Same arithmetic per pair as kernel 2a (v_rsq, v_sin, v_cos on the phase in
// revolutions); what is added per voxel: 12 cross-lane shuffles + adds, and the element loop covers only n / 64 terms per
// lane, so its latency is exposed 64 times per wave instead of once.
template <bool CLAMP>
__global__ __launch_bounds__(FIELD_THREADS) void field_shfl_k(
    const float* __restrict__ tab, float* __restrict__ pmag, float* __restrict__ inten,
    float* __restrict__ cplx, const FieldParams P) {
    constexpr int EP = 4;
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int cpr = (P.nz + 63) / 64;                       // 64-voxel chunks per row
    const long long wave_id = ((long long)blockIdx.x * FIELD_THREADS + threadIdx.x) >> 6;
    const long long rows = (long long)P.nx * P.ny;
    const long long row = wave_id / cpr;
    if (row >= rows) return;                                // wave-uniform
    const int chunk = (int)(wave_id - row * cpr);
    const int i = (int)(row / P.ny), j = (int)(row - (long long)i * P.ny);
    const int k0 = chunk * 64;
    const float x = (float)(i + P.x_begin) * P.hx, y = (float)j * P.hy;
    const float* t = tab + (size_t)f * P.n_el * TAB_STRIDE;
    float my_re = 0.f, my_im = 0.f;                         // lane l: voxel k0 + l
    for (int e0 = 0; e0 < P.n_el; e0 += 64 * EP) {
        float r2[EP], ez[EP], w[EP], phi[EP];
#pragma unroll
        for (int p = 0; p < EP; ++p) {
            const int e = e0 + 64 * p + lane;
            const bool ok = e < P.n_el;
            const float4 a = ok ? *reinterpret_cast<const float4*>(t + (size_t)e * TAB_STRIDE) : make_float4(0.f, 0.f, 0.f, 0.f);
            const float ph = ok ? t[(size_t)e * TAB_STRIDE + 4] : 0.f;
            const float dx = x - a.x, dy = y - a.y;
            r2[p] = fmaf(dy, dy, dx * dx); ez[p] = a.z; w[p] = a.w; phi[p] = ph;   // out-of-range slots: zero weight
        }
        const int nv = min(64, P.nz - k0);
        for (int v = 0; v < nv; ++v) {                      // wave-uniform voxel
            const float z = (float)(k0 + v) * P.hz;
            float sr = 0.f, si = 0.f;
#pragma unroll
            for (int p = 0; p < EP; ++p) {
                const float dz = z - ez[p];
                float d2 = fmaf(dz, dz, r2[p]);
                if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                const float ri = __builtin_amdgcn_rsqf(d2);
                const float ph = fmaf(d2, ri, phi[p]);
                const float a = w[p] * ri;
                sr = fmaf(a, __builtin_amdgcn_cosf(ph), sr);
                si = fmaf(a, __builtin_amdgcn_sinf(ph), si);
            }
#pragma unroll
            for (int m = 32; m >= 1; m >>= 1) {             // wavefront reduction over the elements
                sr += __shfl_xor(sr, m, 64);
                si += __shfl_xor(si, m, 64);
            }
            if (lane == v) { my_re += sr; my_im += si; }
        }
    }
    if (k0 + lane >= P.nz) return;
    const long long o = (long long)f * P.vox + row * P.nz + k0 + lane;
    const float m2 = fmaf(my_re, my_re, my_im * my_im);
    if (P.flags & 1u) pmag[o] = __builtin_sqrtf(m2);
    if (P.flags & 2u) inten[o] = m2 * P.inten_scale;
    if (P.flags & 4u) { cplx[2 * o] = my_re; cplx[2 * o + 1] = my_im; }
}

}  // namespace olx

using namespace olx;

void olx_launch_shfl(olx_ctx* c, float* pm) {
    const FieldParams& P = c->fp;
    const long long waves = (long long)P.nx * P.ny * ((P.nz + 63) / 64);
    dim3 grid((unsigned)((waves * 64 + FIELD_THREADS - 1) / FIELD_THREADS), c->plan_foci), blk(FIELD_THREADS);
    if (c->clamp) hipLaunchKernelGGL((field_shfl_k<true>), grid, blk, 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, P);
    else hipLaunchKernelGGL((field_shfl_k<false>), grid, blk, 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, P);
}
#endif  // OLX_AB_VARIANTS
