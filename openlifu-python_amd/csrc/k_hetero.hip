// kernel 2h (field_hetero_k): heterogeneous medium, straight-ray layered model
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// kernel 2h: heterogeneous medium, straight-ray layered model (definition: oracle/field_oracle.c
// olo_field_grid_hetero / olo_field_grid_hetero_layers).  Per (voxel, element) the ray is sampled
// where it crosses each NON-TRIVIAL grid plane (planes whose excess slowness and absorption are identically zero are
// skipped; the host lists the others) lying between the element and the voxel: bilinear sample (clamped to the border)
// of {sig, a'} -> E' = l' sum sig (extra path, wavelengths), A = l' sum a' (nepers), l' = hz d / |dz|.  Then the usual
// term with phase d + E' + phi and amplitude w exp(-A) / d.
//
//  * The ray integrals E', A depend on (voxel, element) only -- not on the focus.  One lane therefore evaluates them
//    ONCE and feeds NF foci (template; the steering table of a launch tile carries NF (w, phi) pairs per element):
//    per extra focus only the phase add, sin, cos and two fma are repeated, not the ~33 gathers of a skull layer.
//  * LAYERS (opt-in, olx_field_medium_layering G > 1): two-level quadrature.  Runs of non-trivial planes are cut into
//    layers of <= G planes carrying the column sums of their planes; a layer that lies wholly between element and voxel
//    is sampled once at its mid height (a thin phase / absorption screen), the planes of a layer that is only partly
//    between (the one the voxel sits in) individually.  G = 1 (default) is the one-sample-per-plane model.
//
// The medium is stored as a PRE-GATHERED bilinear stencil: texel (p, i, j) = 8 floats { sig, a' } x {(i,j), (i,j+1),
// (i+1,j), (i+1,j+1)} (edge-clamped), 32 B aligned, so one sample = two 16-B loads from one cache line instead of four
// 8-B gathers from two rows.  Work map: a wave = an 8 x 8 (x, y) tile of voxels x ZPL consecutive z per lane, the four
// waves of a block = four consecutive z chunks of the same tile; for a fixed plane and element the crossing points of
// the wave's 64 rays form a compact (shrunken) image of the tile, so the gathers of one wave-instruction fall into a
// few cache lines.  Table entry (tile, e) = { x, y, z, kfirst, klast, 0, 0, 0, (w_f, phi_f) f < NF } with kfirst /
// klast = first / last plane index strictly above / below the element (bit-cast ints, decided on the host in fp64).
// ------------------------------------------------------------------------------------
typedef float het_f2_t __attribute__((ext_vector_type(2)));
typedef float het_f4u_t __attribute__((ext_vector_type(4), aligned(16)));
// (every cell carries its own 2 x 2 stencil with the border already folded in, so u = n - 1 is a valid cell with weight 0; the plane
// base is wave-uniform -- scalar registers -- and the lane adds a 32-bit byte offset: a plane of stencils, 32 bytes per cell, stays far
// below 4 GiB for any lateral grid that fits the volumes themselves)
__device__ __forceinline__ void hetero_sample(const float4* __restrict__ plane, float tt, float dxu, float dyv, float eu, float ev,
                                              float umax, float vmax, int nyg, float& ss, float& as) {
    const float u = __builtin_amdgcn_fmed3f(fmaf(tt, dxu, eu), 0.f, umax);   // border values extend outwards
    const float v = __builtin_amdgcn_fmed3f(fmaf(tt, dyv, ev), 0.f, vmax);
    const unsigned i0 = (unsigned)(int)u, j0 = (unsigned)(int)v;
    const float fu = __builtin_amdgcn_fractf(u), fv = __builtin_amdgcn_fractf(v);
    const unsigned off = (__umul24(i0, (unsigned)nyg) + j0) << 5;            // 32 bytes per cell
    const char* base = reinterpret_cast<const char*>(plane);
    const het_f4u_t lo = *reinterpret_cast<const het_f4u_t*>(base + off), hi = *reinterpret_cast<const het_f4u_t*>(base + off + 16);
    // {s00,a00,s01,a01}, {s10,a10,s11,a11}: the (s, a) pairs interpolate in packed fp32
    const het_f2_t l0 = {lo.x, lo.y}, l1 = {lo.z, lo.w}, h0 = {hi.x, hi.y}, h1 = {hi.z, hi.w};
    const het_f2_t c0 = fv * (l1 - l0) + l0, c1 = fv * (h1 - h0) + h0;
    const het_f2_t sa = fu * (c1 - c0) + c0;
    ss += sa.x;
    as += sa.y;
}

template <int ZPL, int NF, bool CLAMP, bool LAYERS>
__global__ __launch_bounds__(FIELD_THREADS) void field_hetero_k(
    const float* __restrict__ tab, const float4* __restrict__ med, const float4* __restrict__ med_layer,
    const int* __restrict__ plane_k, const int* __restrict__ plane_of_k, const int* __restrict__ layer_lo,
    const int* __restrict__ layer_hi, const float* __restrict__ inv2z, float* __restrict__ pmag,
    float* __restrict__ inten, float* __restrict__ cplx, const FieldParams P, const HeteroParams H) {
    constexpr int STRIDE = HET_TAB_HEAD + 2 * NF;
    const int ftile = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_y = (P.ny + 7) >> 3, zblocks = (P.nz + 4 * ZPL - 1) / (4 * ZPL);
    const int zb = blockIdx.x % zblocks;
    const int tile = blockIdx.x / zblocks;
    const int ti = tile / tiles_y, tj = tile - ti * tiles_y;
    const int i = ti * 8 + (lane >> 3), j = tj * 8 + (lane & 7);
    const int k0 = (zb * 4 + wave) * ZPL;
    const bool live = i < P.nx && j < P.ny && k0 < P.nz;
    const int ic = min(i, P.nx - 1), jc = min(j, P.ny - 1);
    const float xi = (float)(ic + P.x_begin), yj = (float)jc;      // voxel indices: differences to an element from exact index differences (table slots 5 .. 10)
    float z[ZPL], re[ZPL][NF], im[ZPL][NF], sv[ZPL], av[ZPL];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        const int kq = min(k0 + q, P.nz - 1);
        z[q] = (float)kq;                                   // plane index
#pragma unroll
        for (int f = 0; f < NF; ++f) { re[q][f] = 0.f; im[q][f] = 0.f; }
        const int pq = plane_of_k[kq];                   // the voxel's own half layer
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pq >= 0) m = med[(((size_t)pq * H.nxg + (ic + H.xg_begin)) * H.nyg + jc) * 2];
        sv[q] = 0.5f * m.x; av[q] = 0.5f * m.y;
    }
    const size_t plane_sz = (size_t)H.nxg * H.nyg * 2;
    const float umax = (float)(H.nxg - 1), vmax = (float)(H.nyg - 1);
    const float* t = tab + (size_t)ftile * P.n_el * STRIDE;
    for (int e = 0; e < P.n_el; ++e) {
        const float* te = t + (size_t)e * STRIDE;
        const float ex = te[0], ey = te[1];
        const int kfirst = __float_as_int(te[3]), klast = __float_as_int(te[4]);
        const float kez = te[7], fez = te[10];
        const float dx = fmaf(xi - te[5], P.hx, -te[8]), dy = fmaf(yj - te[6], P.hy, -te[9]);
        const float r2 = fmaf(dy, dy, dx * dx);
        const float eu = fmaf(ex, H.inv_hx, H.u0), ev = fmaf(ey, H.inv_hy, H.v0);   // element in grid index space
        const float dxu = dx * H.inv_hx, dyv = dy * H.inv_hy;
        float dz[ZPL], idz[ZPL], ss[ZPL], as[ZPL];
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            dz[q] = fmaf(z[q] - kez, P.hz, -fez);
            idz[q] = dz[q] != 0.f ? __builtin_amdgcn_rcpf(dz[q]) : 0.f;
            ss[q] = sv[q]; as[q] = av[q];
        }
        // planes between element and voxel: k in [kfirst, kv) (voxel above) or (kv, klast] (voxel below).
        // k0..k0+ZPL-1 are wave-uniform, so every trip bound and branch below is too.
        if constexpr (LAYERS) {
            for (int g = 0; g < H.n_layers; ++g) {
                const int lo = layer_lo[g], hi = layer_hi[g];
                if ((hi < kfirst || lo >= k0 + ZPL - 1) && (lo > klast || hi <= k0)) continue;   // no plane of it is between for any q
                const float zg = fmaf(0.5f * (float)(lo + hi) - kez, P.hz, -fez);
                bool part = false;
#pragma unroll
                for (int q = 0; q < ZPL; ++q) {
                    const int kv = k0 + q;
                    const bool full = (lo >= kfirst && hi < kv) || (lo > kv && hi <= klast);
                    if (full) hetero_sample(med_layer + (size_t)g * plane_sz, zg * idz[q], dxu, dyv, eu, ev, umax, vmax, H.nyg, ss[q], as[q]);
                    else part = true;
                }
                if (!part) continue;
                for (int k = lo; k <= hi; ++k) {          // the layer the voxels sit in (or an element plane cuts): plane by plane
                    const float zk = fmaf((float)k - kez, P.hz, -fez);
                    const float4* plane = med + (size_t)plane_of_k[k] * plane_sz;
#pragma unroll
                    for (int q = 0; q < ZPL; ++q) {
                        const int kv = k0 + q;
                        const bool full = (lo >= kfirst && hi < kv) || (lo > kv && hi <= klast);
                        const bool between = (k >= kfirst && k < kv) || (k <= klast && k > kv);
                        if (full || !between) continue;
                        hetero_sample(plane, zk * idz[q], dxu, dyv, eu, ev, umax, vmax, H.nyg, ss[q], as[q]);
                    }
                }
            }
        } else {
            for (int p = 0; p < H.n_planes; ++p) {
                const int k = plane_k[p];                    // wave-uniform
                const bool any_above = k >= kfirst && k < k0 + ZPL - 1, any_below = k <= klast && k > k0;
                if (!any_above && !any_below) continue;
                const float zk = fmaf((float)k - kez, P.hz, -fez);
                const float4* plane = med + (size_t)p * plane_sz;
#pragma unroll
                for (int q = 0; q < ZPL; ++q) {
                    const int kv = k0 + q;
                    const bool between = (k >= kfirst && k < kv) || (k <= klast && k > kv);   // wave-uniform
                    if (!between) continue;
                    hetero_sample(plane, zk * idz[q], dxu, dyv, eu, ev, umax, vmax, H.nyg, ss[q], as[q]);
                }
            }
        }
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            float d2 = fmaf(dz[q], dz[q], r2);
            if (CLAMP) d2 = fmaxf(d2, P.dmin2);
            const float ri = __builtin_amdgcn_rsqf(d2);
            const float d = d2 * ri;
            const float l = dz[q] != 0.f ? P.hz * d * fabsf(idz[q]) : 0.f;   // path per layer [wavelengths]
            const float ph0 = fmaf(l, ss[q], d);
            const float a0 = ri * __expf(-l * as[q]);
#pragma unroll
            for (int f = 0; f < NF; ++f) {               // the ray integrals above serve every focus of the tile
                const float ph = ph0 + te[HET_TAB_HEAD + 2 * f + 1];
                const float a = a0 * te[HET_TAB_HEAD + 2 * f];
                re[q][f] = fmaf(a, __builtin_amdgcn_cosf(ph), re[q][f]);
                im[q][f] = fmaf(a, __builtin_amdgcn_sinf(ph), im[q][f]);
            }
        }
    }
    if (!live) return;
    const long long vrow = ((long long)i * P.ny + j) * P.nz + k0;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int fg = ftile * NF + f;
        if (fg >= H.n_foci) break;
        const long long base = (long long)fg * P.vox + vrow;
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            if (k0 + q >= P.nz) continue;
            const float m2 = fmaf(re[q][f], re[q][f], im[q][f] * im[q][f]);
            if (P.flags & 1u) pmag[base + q] = __builtin_sqrtf(m2);
            if (P.flags & 2u) inten[base + q] = m2 * (inv2z ? inv2z[vrow + q] : P.inten_scale);
            if (P.flags & 4u) { cplx[2 * (base + q)] = re[q][f]; cplx[2 * (base + q) + 1] = im[q][f]; }
        }
    }
}

// steering pack for kernel 2h: fp64 (pos, area, delays, apod) -> [tiles][n][HET_TAB_HEAD + 2 NF] floats (see above);
// lengths in wavelengths, w = a P0 S / lambda^2, phi = frac(f0 tau) [revolutions]; foci past the last one get w = 0.
__global__ void steer_pack_hetero_k(const double* __restrict__ pos, const double* __restrict__ area, int n,
                                    const double* __restrict__ delays, const double* __restrict__ apod, double ox, double oy,
                                    double oz, double freq, double p0_over_lambda, double rev, const int* __restrict__ kfirst,
                                    const int* __restrict__ klast, int n_foci, int nf, double hx_m, double hy_m, double hz_m /* spacing [m] */, float* __restrict__ tab) {
    const int tile = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    float* t = tab + ((size_t)tile * n + e) * (HET_TAB_HEAD + 2 * nf);
    t[0] = (float)((pos[e] - ox) * rev);
    t[1] = (float)((pos[n + e] - oy) * rev);
    t[2] = (float)((pos[2 * n + e] - oz) * rev);
    t[3] = __int_as_float(kfirst[e]);
    t[4] = __int_as_float(klast[e]);
    {   // the element as (voxel index, offset from that voxel): voxel - element differences from exact index differences next to the elements (round 6: kernel 2h
        // throughout, kernel 2m in the planes below the medium; as kernels 2a - 2c)
        const double q[3] = {pos[e] - ox, pos[n + e] - oy, pos[2 * n + e] - oz}, h[3] = {hx_m, hy_m, hz_m};
        for (int a = 0; a < 3; ++a) { const double k = rint(q[a] / h[a]); t[5 + a] = (float)k; t[8 + a] = (float)((q[a] - k * h[a]) * rev); }
        t[11] = 0.f;
    }
    for (int fl = 0; fl < nf; ++fl) {
        const int f = tile * nf + fl;
        float w = 0.f, ph = 0.f;
        if (f < n_foci) {
            const size_t o = (size_t)f * n + e;
            const double cyc = freq * delays[o];
            w = (float)(apod[o] * area[e] * p0_over_lambda * rev);
            ph = (float)(cyc - floor(cyc));
        }
        t[HET_TAB_HEAD + 2 * fl] = w;
        t[HET_TAB_HEAD + 2 * fl + 1] = ph;
    }
}

}  // namespace olx

using namespace olx;

void olx_pack_hetero(olx_ctx* c) {
    const double lambda = c->c / c->freq;
    const int tiles = (c->plan_foci + c->nf - 1) / c->nf;
    dim3 g((c->n_el + 127) / 128, tiles);
    hipLaunchKernelGGL(steer_pack_hetero_k, g, dim3(128), 0, c->stream, c->d_pos, c->d_area, c->n_el, c->d_delays, c->d_apod,
                       c->grid.origin[0], c->grid.origin[1], c->grid.origin[2], c->freq, c->p0_pa / lambda, c->freq / c->c,
                       c->d_kfirst, c->d_klast, c->plan_foci, c->nf, c->grid.spacing[0], c->grid.spacing[1], c->grid.spacing[2], c->d_tab);
}

template <int NF>
static void launch_hetero_nf(olx_ctx* c, float* pm) {
    const FieldParams& P = c->fp;
    const long long nblk = (long long)((P.nx + 7) / 8) * ((P.ny + 7) / 8) * ((P.nz + 15) / 16);  // 8x8 tile x 16 z
    dim3 grid((unsigned)nblk, (c->plan_foci + NF - 1) / NF), blk(FIELD_THREADS);
    const bool layers = c->hp.n_layers > 0;
#define OLX_HET(CL, LY) hipLaunchKernelGGL((field_hetero_k<4, NF, CL, LY>), grid, blk, 0, c->stream, c->d_tab, c->d_med, c->d_med_layer, \
                                           c->d_plane_k, c->d_plane_of_k, c->d_layer_lo, c->d_layer_hi, c->d_inv2z, pm, c->d_inten, c->d_cplx, P, c->hp)
    if (layers) { if (c->clamp) OLX_HET(true, true); else OLX_HET(false, true); }
    else        { if (c->clamp) OLX_HET(true, false); else OLX_HET(false, false); }
#undef OLX_HET
}

void olx_launch_hetero(olx_ctx* c, float* pm) {
    if (c->nf >= 8) launch_hetero_nf<8>(c, pm);
    else if (c->nf >= 4) launch_hetero_nf<4>(c, pm);
    else if (c->nf >= 2) launch_hetero_nf<2>(c, pm);
    else launch_hetero_nf<1>(c, pm);
}
