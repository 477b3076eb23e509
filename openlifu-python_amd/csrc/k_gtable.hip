// Precomputed geometry table of the lattice kernels (gtable_gen_k): what kernel 2g's blocks used to evaluate for themselves.
// gfx950 (CDNA4, wave64) only.  Layout: olx_params.h (CosetParams::gt_*), DESIGN.md section 4.
//
// A table entry G(U, W, k) = S_G exp(j 2 pi d) / d, d^2 = dx(U)^2 + dy(W)^2 + dz(k)^2 [wavelengths], split into the fp16 hi / lo (or e4m3) words
// of the matrix operand, depends on the array lattice, the grid and the frequency -- not on foci, steering, launch tile or call.  Kernel 2g
// evaluated each one 7.5 x per launch on the headline grid (every block its own 26 x 12 x 16 window, 9 216 blocks), and again for every launch
// tile of a sweep and every calc_solution: rsq + sin + cos + 2 converts + 2 mixed fmas per entry, a fifth of the kernel's vector issue cycles.
// Here every entry is evaluated ONCE per plan with exactly the expression (and the operation order the compiler gives it) of the in-kernel
// generation -- the GT instantiations of field_cosetp_k are bit-identical to the generating ones (tests/test_gpu_field.py) -- and streamed
// to HBM: 8 bytes per entry, (offsets per class) x planes x 144 classes = 223 MB on the headline grid, written at store rate in ~0.1 ms.
// Entry order: class (U mod mx, W mod my) slowest -- a block only ever reads ONE class --, then plane, table row (W), table column in LDS
// order (U descending): a table row is 96 contiguous bytes, the planes of a block 16 strides of NW NU entries apart.
//
// MEASURED SLOWER (round 4, DESIGN.md 5.4, profiles/r04_gtable_*): the copies cost more in the vector-memory pipe than the evaluation costs on
// the vector ALU, whose time the other resident block's matrix instructions were covering anyway.  Developer library only (OLX_GTABLE=1).
#ifdef OLX_AB_VARIANTS
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

template <bool CLAMP, bool FP8>
__global__ __launch_bounds__(256) void gtable_gen_k(GtEntry* __restrict__ tab, const CosetParams P) {
#pragma clang fp contract(off)      // every fused operation below is written as one: the same ones the compiler forms in field_cosetp_k
    const int cls = blockIdx.z, k = blockIdx.y;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= P.gt_nw * P.gt_nu) return;
    const int w = idx / P.gt_nu, ur = idx - w * P.gt_nu;
    const int cx = cls / P.my, cy = cls - cx * P.my;
    const float U = (float)(P.gt_ulo + cx + P.mx * (P.gt_nu - 1 - ur));
    const float W = (float)(P.gt_wlo + cy + P.my * w);
    const float dx = __builtin_fmaf(U, P.hx_hi, __builtin_fmaf(U, P.hx_lo, P.fx0));
    const float dy = __builtin_fmaf(W, P.hy_hi, __builtin_fmaf(W, P.hy_lo, P.fy0));
    const float dz = __builtin_fmaf(P.hz, (float)k, -P.flat_ez);
    const float dx2 = dx * dx, dz2 = dz * dz;
    const float r2 = __builtin_fmaf(dy, dy, dx2);
    float d2 = r2 + dz2;
    if (CLAMP) d2 = fmaxf(d2, P.dmin2);
    const float ri = __builtin_amdgcn_rsqf(d2);
    const float ph = d2 * ri;
    const float rs = ri * P.g_scale;
    const float gr = rs * __builtin_amdgcn_cosf(ph);
    const float gi = rs * __builtin_amdgcn_sinf(ph);
    half2_t hi;
    if constexpr (FP8) hi = __builtin_convertvector(float2_t{gr, gi}, half2_t);
    else hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
    float lr, li;
    const unsigned hw = __builtin_bit_cast(unsigned, hi);
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
    unsigned lo_word;
    if constexpr (FP8) {
        short2_t wq;
        wq = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wq, lr, li, 1.0f / COS_F8_LO, false);
        wq = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(wq, gr, gi, 1.0f / COS_F8_HI, true);
        lo_word = __builtin_bit_cast(unsigned, wq);
    } else {
        lo_word = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lr, li));
    }
    const long long e = (((long long)cls * P.gt_nzp + k) * P.gt_nw + w) * P.gt_nu + ur;
    tab[e] = GtEntry{hw, lo_word};
}

}  // namespace olx

using namespace olx;

// (Re)build the context's geometry table for the planned kernel-2g launch unless the resident one was built from the same parameters.
int olx_gtable_prepare(olx_ctx* c) {
    const CosetParams& Q = c->cp;
    const bool clamp = c->clamp || c->lat.clamp;
    struct Key {
        int mx, my, nu, nw, nzp, ulo, wlo, clamp, fp8;
        float fx0, fy0, hx_hi, hx_lo, hy_hi, hy_lo, hz, dmin2, flat_ez, g_scale;
    } key{Q.mx, Q.my, Q.gt_nu, Q.gt_nw, Q.gt_nzp, Q.gt_ulo, Q.gt_wlo, clamp ? 1 : 0, c->fp8corr ? 1 : 0,
          Q.fx0, Q.fy0, Q.hx_hi, Q.hx_lo, Q.hy_hi, Q.hy_lo, Q.hz, clamp ? Q.dmin2 : 0.f, Q.flat_ez, Q.g_scale};
    const std::string ks(reinterpret_cast<const char*>(&key), sizeof key);
    const size_t need = (size_t)Q.mx * Q.my * Q.gt_nzp * Q.gt_nw * Q.gt_nu;
    if (c->d_gtab && c->gtab_key == ks && c->gtab_cap >= need) return OLX_OK;
    c->gtab_key.clear();
    if (c->gtab_cap < need) {
        HIPCHK(c, hipStreamSynchronize(c->stream));       // (a launch in flight may still read the old table)
        if (c->d_gtab) hipFree(c->d_gtab);
        c->d_gtab = nullptr; c->gtab_cap = 0;
        HIPCHK(c, hipMalloc((void**)&c->d_gtab, sizeof(GtEntry) * need));
        c->gtab_cap = need;
    }
    const dim3 grid((unsigned)((Q.gt_nw * Q.gt_nu + 255) / 256), (unsigned)Q.gt_nzp, (unsigned)(Q.mx * Q.my));
    if (c->fp8corr) { if (clamp) hipLaunchKernelGGL((gtable_gen_k<true, true>), grid, dim3(256), 0, c->stream, c->d_gtab, Q); else hipLaunchKernelGGL((gtable_gen_k<false, true>), grid, dim3(256), 0, c->stream, c->d_gtab, Q); }
    else            { if (clamp) hipLaunchKernelGGL((gtable_gen_k<true, false>), grid, dim3(256), 0, c->stream, c->d_gtab, Q); else hipLaunchKernelGGL((gtable_gen_k<false, false>), grid, dim3(256), 0, c->stream, c->d_gtab, Q); }
    HIPCHK(c, hipGetLastError());
    c->gtab_key = ks;
    return OLX_OK;
}
#endif  // OLX_AB_VARIANTS
