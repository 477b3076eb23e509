// kernel 2h (field_hetero_k): heterogeneous medium, straight-ray layered model
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// kernel 2h: heterogeneous medium, straight-ray layered model (definition: oracle/field_oracle.c,
// DESIGN.md section 7).  Per (voxel, element) the ray is sampled where it crosses each NON-TRIVIAL grid plane
// (planes whose excess slowness and absorption are identically zero are skipped; the host lists the others)
// lying between the element and the voxel: bilinear gather (clamped to the border) of {sig, a'} (float2, plane-major [np][nx][ny],
// L2 / Infinity-Cache resident) -> E' = l' sum sig (extra path, wavelengths), A = l' sum a' (nepers),
// l' = hz d / |dz|.  Then the usual term with phase d + E' + phi and amplitude w exp(-A) / d.
// Table entry: kernel-2a layout with slots 5 / 6 = first / last plane index strictly above / below the
// element (bit-cast ints, decided on the host in fp64).  Work map as kernel 2a (ZPL z voxels per lane).
// ------------------------------------------------------------------------------------

// Work map of kernel 2h: a wave = an 8 x 8 (x, y) tile of voxels x ZPL consecutive z per lane, the four
// waves of a block = four consecutive z chunks of the same tile.  For a fixed plane and element the
// crossing points of the wave's 64 rays then form a compact (shrunken) image of the tile, so the gathers
// of one wave-instruction fall into a few cache lines.  The medium is stored as a PRE-GATHERED bilinear
// stencil: texel (p, i, j) = 8 floats { sig, a' } x {(i,j), (i,j+1), (i+1,j), (i+1,j+1)} (edge-clamped),
// 32 B aligned, so one sample = two 16-B loads from one cache line instead of four 8-B gathers from two rows.
template <int ZPL, bool CLAMP>
__global__ __launch_bounds__(FIELD_THREADS) void field_hetero_k(
    const float* __restrict__ tab, const float4* __restrict__ med, const int* __restrict__ plane_k,
    const int* __restrict__ plane_of_k, const float* __restrict__ inv2z, float* __restrict__ pmag,
    float* __restrict__ inten, float* __restrict__ cplx, const FieldParams P, const HeteroParams H) {
    const int f = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int tiles_y = (P.ny + 7) >> 3, zblocks = (P.nz + 4 * ZPL - 1) / (4 * ZPL);
    const int zb = blockIdx.x % zblocks;
    const int tile = blockIdx.x / zblocks;
    const int ti = tile / tiles_y, tj = tile - ti * tiles_y;
    const int i = ti * 8 + (lane >> 3), j = tj * 8 + (lane & 7);
    const int k0 = (zb * 4 + wave) * ZPL;
    const bool live = i < P.nx && j < P.ny && k0 < P.nz;
    const int ic = min(i, P.nx - 1), jc = min(j, P.ny - 1);
    const float x = (float)(ic + P.x_begin) * P.hx, y = (float)jc * P.hy;
    float z[ZPL], re[ZPL], im[ZPL], sv[ZPL], av[ZPL];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        const int kq = min(k0 + q, P.nz - 1);
        z[q] = (float)kq * P.hz;
        re[q] = 0.f; im[q] = 0.f;
        const int pq = plane_of_k[kq];                   // the voxel's own half layer
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f);
        if (pq >= 0) m = med[(((size_t)pq * H.nxg + (ic + H.xg_begin)) * H.nyg + jc) * 2];
        sv[q] = 0.5f * m.x; av[q] = 0.5f * m.y;
    }
    const float* t = tab + (size_t)f * P.n_el * TAB_STRIDE;
    for (int e = 0; e < P.n_el; ++e) {
        const float ex = t[e * TAB_STRIDE + 0], ey = t[e * TAB_STRIDE + 1], ez = t[e * TAB_STRIDE + 2];
        const float w = t[e * TAB_STRIDE + 3], phi = t[e * TAB_STRIDE + 4];
        const int kfirst = __float_as_int(t[e * TAB_STRIDE + 5]), klast = __float_as_int(t[e * TAB_STRIDE + 6]);
        const float dx = x - ex, dy = y - ey;
        const float r2 = fmaf(dy, dy, dx * dx);
        const float eu = fmaf(ex, H.inv_hx, H.u0), ev = fmaf(ey, H.inv_hy, H.v0);   // element in grid index space
        const float dxu = dx * H.inv_hx, dyv = dy * H.inv_hy;
        const float umax = (float)(H.nxg - 1), vmax = (float)(H.nyg - 1);
        float dz[ZPL], idz[ZPL], ss[ZPL], as[ZPL];
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            dz[q] = z[q] - ez;
            idz[q] = dz[q] != 0.f ? __builtin_amdgcn_rcpf(dz[q]) : 0.f;
            ss[q] = sv[q]; as[q] = av[q];
        }
        // planes between element and voxel: k in [kfirst, kv) (voxel above) or (kv, klast] (voxel below).
        // k0..k0+ZPL-1 are wave-uniform, so the trip bounds are too.
        for (int p = 0; p < H.n_planes; ++p) {
            const int k = plane_k[p];                    // wave-uniform
            const bool any_above = k >= kfirst && k < k0 + ZPL - 1, any_below = k <= klast && k > k0;
            if (!any_above && !any_below) continue;
            const float zk = (float)k * P.hz - ez;
            const float4* plane = med + (size_t)p * H.nxg * H.nyg * 2;
#pragma unroll
            for (int q = 0; q < ZPL; ++q) {
                const int kv = k0 + q;
                const bool between = (k >= kfirst && k < kv) || (k <= klast && k > kv);   // wave-uniform
                if (!between) continue;
                const float tt = zk * idz[q];
                const float u = fminf(fmaxf(fmaf(tt, dxu, eu), 0.f), umax);   // border values extend outwards
                const float v = fminf(fmaxf(fmaf(tt, dyv, ev), 0.f), vmax);
                const int i0 = (int)u, j0 = (int)v;
                const float fu = u - (float)i0, fv = v - (float)j0;
                const float4* tx = plane + ((size_t)i0 * H.nyg + j0) * 2;
                const float4 lo = tx[0], hi = tx[1];     // {s00,a00,s01,a01}, {s10,a10,s11,a11}
                const float s0 = fmaf(fv, lo.z - lo.x, lo.x), a0 = fmaf(fv, lo.w - lo.y, lo.y);
                const float s1 = fmaf(fv, hi.z - hi.x, hi.x), a1 = fmaf(fv, hi.w - hi.y, hi.y);
                ss[q] += fmaf(fu, s1 - s0, s0);
                as[q] += fmaf(fu, a1 - a0, a0);
            }
        }
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            float d2 = fmaf(dz[q], dz[q], r2);
            if (CLAMP) d2 = fmaxf(d2, P.dmin2);
            const float ri = __builtin_amdgcn_rsqf(d2);
            const float d = d2 * ri;
            const float l = dz[q] != 0.f ? P.hz * d * fabsf(idz[q]) : 0.f;   // path per layer [wavelengths]
            const float ph = fmaf(l, ss[q], d) + phi;
            const float a = w * ri * __expf(-l * as[q]);
            re[q] = fmaf(a, __builtin_amdgcn_cosf(ph), re[q]);
            im[q] = fmaf(a, __builtin_amdgcn_sinf(ph), im[q]);
        }
    }
    if (!live) return;
    const long long vrow = ((long long)i * P.ny + j) * P.nz + k0;
    const long long base = (long long)f * P.vox + vrow;
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        if (k0 + q >= P.nz) continue;
        const float m2 = fmaf(re[q], re[q], im[q] * im[q]);
        if (P.flags & 1u) pmag[base + q] = __builtin_sqrtf(m2);
        if (P.flags & 2u) inten[base + q] = m2 * (inv2z ? inv2z[vrow + q] : P.inten_scale);
        if (P.flags & 4u) { cplx[2 * (base + q)] = re[q]; cplx[2 * (base + q) + 1] = im[q]; }
    }
}


}  // namespace olx

using namespace olx;

void olx_launch_hetero(olx_ctx* c, float* pm) {
    const FieldParams& P = c->fp;
    const long long nblk = (long long)((P.nx + 7) / 8) * ((P.ny + 7) / 8) * ((P.nz + 15) / 16);  // 8x8 tile x 16 z
    dim3 grid((unsigned)nblk, c->plan_foci), blk(FIELD_THREADS);
    if (c->clamp) hipLaunchKernelGGL((field_hetero_k<4, true>), grid, blk, 0, c->stream, c->d_tab, c->d_med, c->d_plane_k, c->d_plane_of_k, c->d_inv2z, pm, c->d_inten, c->d_cplx, P, c->hp);
    else          hipLaunchKernelGGL((field_hetero_k<4, false>), grid, blk, 0, c->stream, c->d_tab, c->d_med, c->d_plane_k, c->d_plane_of_k, c->d_inv2z, pm, c->d_inten, c->d_cplx, P, c->hp);
}
