// kernel 2m (field_hmarch_k): heterogeneous medium, MARCHED ray integrals -- one bilinear look-up per (voxel, element)
// gfx950 (CDNA4, wave64) only.  Definition: oracle/field_oracle.c olo_field_columns_hetero_march, DESIGN.md section 7.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"
#include <string>
#include <type_traits>

namespace olx {

// ------------------------------------------------------------------------------------
// Kernel 2h samples the medium on EVERY non-trivial plane between element and voxel (~32 bilinear gathers per ray on
// the skull-slab phantom, ~30 VALU instructions each: 137 ms per focus at 256 el x 256^3).  Here the ray sums are
// carried upwards through the non-trivial planes m_0 < m_1 < ... ON THE GRID, per element:
//     U_0(i,j) = med(i,j,m_0),   U_p(i,j) = B[U_{p-1}](crossing of the ray e -> (i,j,m_p) with plane m_{p-1}) + med(i,j,m_p)
// (B = bilinear, border values extended outwards) and a voxel of plane k takes ONE look-up into U_{p*}, p* = the last
// non-trivial plane strictly below k.  Work per ray: one gather instead of one per plane.  Precondition (host-checked):
// every element lies strictly below plane m_0 -- rays then cross the non-trivial planes upwards only, and voxels level with
// or below an element see none.
//
// The host walks the planes in order and issues one launch per SEGMENT = a run of voxel planes that read the same U_p:
//   * planes (m_p, m_{p+1}) -- trivial planes, any count -- and the final run above the last non-trivial plane: look-ups
//     only (ES = 1: the four waves of a block are four consecutive planes of one 4 x 16 lateral tile, so their gathers
//     share cache lines);
//   * each non-trivial plane m_{p+1} on its own: the look-up value + the plane's own medium term IS U_{p+1}, so the same
//     launch writes the next running sums (U is double-buffered) -- over the WHOLE lateral grid even in an x-slab launch,
//     because rays cross slab boundaries (field outputs stay masked to the slab).  One plane is only 1024 tiles, so the
//     elements are split over the 16 waves of a 1024-thread block (ES = 16) and their partial sums meet in LDS.
// U layout: [i][element][j] float2 {sum sig, sum a'} (< 4 GiB, host-checked); a look-up = two 16-byte loads (rows i0, i0 + 1; 8-byte
// aligned).  The writers of one tile row -- concurrent blocks, consecutive in y -- fill 2 KB x n_el contiguous bytes between them.
// The ray sums are focus-independent: up to NF = 8 foci of a launch tile share every look-up (only sin, cos and two fma
// per extra focus).  Table entry as in kernel 2h: { x, y, z, kfirst, klast, 0, 0, 0, (w_f, phi_f) f < NF }.
// ------------------------------------------------------------------------------------
struct MarchSeg {
    int k_lo, k_hi;      // voxel planes of this launch (inclusive)
    int k_src;           // grid plane the source sums U_src live on (-1: none -- nothing non-trivial below these planes)
    int write;           // 1: k_lo == k_hi is a non-trivial plane; write U_dst (launch covers the whole lateral grid)
    int i0, ni;          // lateral extent of the launch in GLOBAL x indices: the slab, or the whole grid when writing
    unsigned nblocks;    // logical blocks of the launch (the grid is rounded up to a multiple of 8, one residue per XCD)
    int reverse;         // walk the logical blocks downwards: consecutive writer launches alternate, so that what one wrote LAST the next
                         // reads FIRST -- out of the Infinity Cache (two U buffers are 268 MB at 256 el x 256^2, just past its 256 MiB)
};

typedef float float4u_t __attribute__((ext_vector_type(4), aligned(8)));
typedef float float2u_t __attribute__((ext_vector_type(2), aligned(4)));      // (also a 4-byte-aligned 8-byte load: a stencil row of the one-sum form)

// ES = 1: 4 waves = 4 consecutive planes, every wave walks all elements.  ES = 16: 16 waves = 16 element subsets of ONE plane.
template <int ES> constexpr int hm_waves() { return ES == 1 ? 4 : ES; }
#ifndef HM_EU1
#define HM_EU1 4
#endif
#ifndef HM_EUW
#define HM_EUW 4
#endif
constexpr int HM_TJ = 16, HM_TI = 4;       // lateral tile of a wave: 4 x 16 voxels -- 16 consecutive y = whole 128-byte lines of a U row of {sum sig, sum a'} pairs
                                           // (one-sum form: 64 bytes; a 2 x 32 tile for its writers measured the same 42 us per launch)

// ONE (round 5): media whose absorption is PROPORTIONAL to their slowness perturbation -- a' = kappa sig in every voxel: every two-material
// segmentation over a lossless reference medium (water + skull: UniformWater's alpha is 0), the shape of BASELINE configs[4] -- carry ONE running
// sum.  U is then [i][element][j] float {sum sig}, a look-up two 8-byte loads instead of two 16-byte ones, the bilinear step plain fp32 instead
// of packed pairs, and sum a' = kappa sum sig: half the bytes of the HBM-bound writer launches and of the look-up launches' L1 traffic.  The host
// detects the property from the stencil values (olx_field_set_medium); OLX_MARCH_SUMS=2 pins the general form for A/B runs and tests.
// SRC: the launch reads running sums (k_src >= 0).  Every element lies strictly below the first non-trivial plane (host-checked), so
// with SRC every ray of the launch crosses the source plane upwards (0 < tt < 1) and without it no ray sees anything: the look-up is
// compiled in or out as a whole -- no per-element branch, no zero fill of the gather registers.
// INSIDE (round 5): every element lies at least half a cell inside the lateral grid, so every crossing point -- a convex combination of
// an element and a voxel with 0 < tt < 1 -- does too: the two clamps of the look-up coordinates are compiled out (host-checked, olx_launch_hmarch).
// TEX (round 5; ONE, look-up launches): the source sums come as ROW PAIRS -- cell (i, e, j) holds {U(i,j), U(i+1,j)} (u_texel_k below, once per
// launch sequence for the last non-trivial plane), so the cells j0 and j0 + 1 are the whole 2 x 2 stencil in 16 contiguous, 8-byte-aligned
// bytes: a look-up is ONE 16-byte load instead of two 8-byte ones.  (Whole 2 x 2 texels per cell -- 16 bytes, aligned -- measured the same
// time and twice the HBM traffic: 4.4 GB per launch, every cell's bytes being its own.)  The long run of planes above the medium is bound by the vector-memory instruction rate (a wave's load occupies the CU's
// address unit ~14-16 cycles whatever its width: 2 loads x 4 SIMDs per CU > the ~100 issue cycles a SIMD needs per pair), not by bytes.
template <int NF, int ES, bool CLAMP, bool SRC, bool ONE, bool INSIDE, bool TEX = false>
__global__ __launch_bounds__(64 * hm_waves<ES>()) void field_hmarch_k(
    const float* __restrict__ tab, const float4* __restrict__ med, const int* __restrict__ plane_of_k,
    const float2* __restrict__ U_src, float2* __restrict__ U_dst, const float* __restrict__ inv2z,
    float* __restrict__ pmag, float* __restrict__ inten, float* __restrict__ cplx, const FieldParams P,
    const HeteroParams H, const MarchSeg S) {
    constexpr int STRIDE = HET_TAB_HEAD + 2 * NF;
    constexpr int WAVES = hm_waves<ES>(), PB = WAVES / ES;  // planes per block
    constexpr int EU = ES == 1 ? HM_EU1 : HM_EUW;                             // elements in flight per wave: their gathers are issued back to back
    // per-wave ray table: what depends on (element, plane) only -- evaluated once per wave, 64 elements at a time across the
    // lanes, instead of per lane per element: { tt, cu, cv, lf | dz2, ex, ey, - } with the crossing u = tt i + cu (grid cells),
    // lf = hz / |dz|.  In the look-up launches (ES = 1, issue-bound) the steering weights { w_f, phi_f } of the tile's foci ride in
    // the same table (RW more float4s per element): a scalar load per element and focus would stall the wave on its latency.
    constexpr int RW = ES == 1 ? (NF + 1) / 2 : 0;
    constexpr int ECH = ES == 1 ? (NF > 2 ? 64 : 128) : 64; // elements per chunk of the wave's ray table
    __shared__ float4 s_ray[WAVES][ECH][2 + RW];
    __shared__ float s_red[ES > 1 ? (ES - 1) * 2 * NF * 64 : 1];
    const int ftile = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_y = (P.ny + HM_TJ - 1) / HM_TJ, zblocks = (S.k_hi - S.k_lo + PB) / PB;
    // Workgroups go to the 8 XCDs round robin; each XCD takes one contiguous eighth of the logical blocks, so that the blocks that
    // read the same lines of U -- the plane groups of a tile, the tiles next to it -- meet in ONE L2 instead of fetching them into eight
    unsigned lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);
    if (lb >= S.nblocks) return;
    if (S.reverse) lb = S.nblocks - 1u - lb;
    const int zb = (int)(lb % (unsigned)zblocks), tile = (int)(lb / (unsigned)zblocks);
    const int ti = tile / tiles_y, tj = tile - ti * tiles_y;
    const int ig = S.i0 + ti * HM_TI + (lane >> 4), j = tj * HM_TJ + (lane & 15);   // global x index, y index
    const int k = S.k_lo + zb * PB + wave / ES, es = wave % ES;            // wave-uniform (scalar registers)
    const bool in_grid = ig < S.i0 + S.ni && j < P.ny;
    const bool live_k = k <= S.k_hi;                        // (ES > 1: PB = 1, always true)
    const int ic = min(ig, S.i0 + S.ni - 1), jc = min(j, P.ny - 1), kc = min(k, S.k_hi);
    const float igf = (float)ic, jgf = (float)jc;
    const float x = igf * P.hx, y = jgf * P.hy, z = (float)kc * P.hz;
    // the voxel's own half layer
    float sv = 0.f, av = 0.f;
    if constexpr (ES > 1) {                                 // (look-up launches, ES = 1, cover trivial planes only: no own term -- the host's segments)
        const int pq = plane_of_k[kc];                      // wave-uniform
        if (pq >= 0) {
            const float4 m = med[(((size_t)pq * H.nxg + ic) * H.nyg + jc) * 2];
            sv = 0.5f * m.x; av = 0.5f * m.y;
        }
    }
    float re[NF], im[NF];
#pragma unroll
    for (int f = 0; f < NF; ++f) { re[f] = 0.f; im[f] = 0.f; }
    const unsigned row_cells = (unsigned)P.n_el * (unsigned)H.nyg;         // U is [i][element][j]: one grid row of every element, then the next
    // look-up coordinates are clamped just inside the last cell, so that (i0, i0 + 1) / (j0, j0 + 1) always exist and the
    // fraction is fract(u): border values extend outwards (the oracle's clamp) to one ulp of the coordinate
    const float umax = (float)(H.nxg - 1) * (1.f - 0x1p-23f), vmax = (float)(H.nyg - 1) * (1.f - 0x1p-23f);
    const float zsrc = (float)S.k_src * P.hz;
    const float* t = tab + (size_t)ftile * P.n_el * STRIDE;
    const bool writer = ES > 1 && S.write && ftile == 0 && in_grid;        // (ES = 1 launches never write)
    const unsigned own = (unsigned)ig * row_cells + (unsigned)j;           // this voxel's cell in element 0's rows
    const float sv2 = 2.f * sv, av2 = 2.f * av;
    const float k2 = -1.4426950408889634f * H.kappa;       // ONE: a' = kappa sig -> exp(-l sum a') = exp2(k2 l sum sig)
    if (live_k) {
        const int n_mine = (P.n_el - es + ES - 1) / ES;     // elements es, es + ES, ...
        for (int c0 = 0; c0 < n_mine; c0 += ECH) {
            const int n_ch = min(ECH, n_mine - c0);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");      // the previous chunk's reads are done (same wave, in order)
            __builtin_amdgcn_wave_barrier();
            for (int q = lane; q < n_ch; q += 64) {          // ray table of this chunk: one element per lane
                const float* te = t + (size_t)(es + ES * (c0 + q)) * STRIDE;
                const float ex = te[0], ey = te[1], ez = te[2];
                // Planes WITHOUT source sums (!SRC: at or below the first non-trivial plane -- the only planes that come near the elements, which all lie
                // below the medium): voxel - element differences from exact index differences, the element as (voxel index, offset) in table slots
                // 5 .. 10 (round 6, as kernels 2a - 2c: absolute fp32 coordinates of ~ 10 wavelengths lose 1e-6 wavelengths, 1.5e-5 of a term at the clamp
                // distance of a fine grid).  The crossing coefficients of the look-up (r0.y, r0.z) are not needed there: the slots carry the offsets.
                const float dz = SRC ? z - ez : fmaf((float)kc - te[7], P.hz, -te[10]);
                const float idz = dz != 0.f ? 1.0f / dz : 0.f;
                const float tt = SRC ? (zsrc - ez) * idz : 0.f;
                const float eu = fmaf(ex, H.inv_hx, H.u0), ev = fmaf(ey, H.inv_hy, H.v0);   // element in grid cells
                s_ray[wave][q][0] = SRC ? make_float4(tt, eu - tt * eu, ev - tt * ev, P.hz * fabsf(idz)) : make_float4(0.f, te[8], te[9], P.hz * fabsf(idz));
                s_ray[wave][q][1] = SRC ? make_float4(dz * dz, ex, ey, 0.f) : make_float4(dz * dz, te[5], te[6], 0.f);
#pragma unroll
                for (int r = 0; r < RW; ++r) {
                    const bool two = 2 * r + 1 < NF;
                    float w0 = te[HET_TAB_HEAD + 4 * r];
                    if constexpr (ONE && NF == 1) w0 = __builtin_amdgcn_logf(w0);      // (log2; the single focus' weight rides in the exponent: v_log_f32(0) = -inf)
                    s_ray[wave][q][2 + r] = make_float4(w0, te[HET_TAB_HEAD + 4 * r + 1],
                                                        two ? te[HET_TAB_HEAD + 4 * r + 2] : 0.f, two ? te[HET_TAB_HEAD + 4 * r + 3] : 0.f);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            // EU elements at a time: all their gathers are issued before the first is consumed (one basic block, pinned by the
            // scheduling barrier -- with a branch between the two phases the compiler sinks every gather down to its use and the
            // wave waits out each L1 round trip); a ragged tail goes element by element
            auto group = [&](const int q0, auto count) {
                constexpr int E = decltype(count)::value;
                static_assert(!TEX || (ONE && SRC && ES == 1), "texel look-ups: one-sum look-up launches only");
                typedef typename std::conditional<ONE, float2u_t, float4u_t>::type Row;      // a stencil row: {s00, s01} | {s00, a00, s01, a01}
                Row lo[E], hi[E];
                float fu[E], fv[E];
                if constexpr (SRC) {
#pragma unroll
                    for (int s = 0; s < E; ++s) {            // phase 1: addresses and gathers
                        const float4 r0 = s_ray[wave][q0 + s][0];     // same address in every lane: an LDS broadcast
                        float u = fmaf(r0.x, igf, r0.y), v = fmaf(r0.x, jgf, r0.z);
                        if constexpr (!INSIDE) { u = __builtin_amdgcn_fmed3f(u, 0.f, umax); v = __builtin_amdgcn_fmed3f(v, 0.f, vmax); }
                        const unsigned i0 = (unsigned)(int)u, j0 = (unsigned)(int)v;
                        fu[s] = __builtin_amdgcn_fractf(u); fv[s] = __builtin_amdgcn_fractf(v);
                        // wave-uniform 64-bit bases (scalar registers) + one 32-bit byte offset per lane (a U plane is < 4 GiB)
                        if constexpr (TEX) {
                            const char* Te = reinterpret_cast<const char*>(U_src) + (size_t)(es + ES * (c0 + q0 + s)) * H.nyg * 8;
                            const unsigned offt = (__umul24(i0, row_cells) + j0) * 8u;
                            if (!OLX_IN((long long)(es + ES * (c0 + q0 + s)) * H.nyg * 8 + offt + 15, (long long)H.nxg * row_cells * 8 + 8, 6)) { lo[s] = Row{}; hi[s] = lo[s]; continue; }
                            const float4u_t t4 = *reinterpret_cast<const float4u_t*>(Te + offt);      // {U(i0,j0), U(i0+1,j0), U(i0,j0+1), U(i0+1,j0+1)}
                            lo[s] = Row{t4.x, t4.z}; hi[s] = Row{t4.y, t4.w};
                            continue;
                        }
                        constexpr int CB = ONE ? 4 : 8;      // bytes per cell of U
                        const char* Ue = reinterpret_cast<const char*>(U_src) + (size_t)(es + ES * (c0 + q0 + s)) * H.nyg * CB;
                        const unsigned off = (__umul24(i0, row_cells) + j0) * CB;
                        if (!OLX_IN((long long)(es + ES * (c0 + q0 + s)) * H.nyg * CB + off + (long long)row_cells * CB + 2 * CB - 1, (long long)H.nxg * row_cells * CB, 6)) { lo[s] = Row{}; hi[s] = lo[s]; continue; }
                        lo[s] = *reinterpret_cast<const Row*>(Ue + off);                             // {s00, a00, s01, a01} | {s00, s01}
                        hi[s] = *reinterpret_cast<const Row*>(Ue + (size_t)row_cells * CB + off);     // {s10, a10, s11, a11} | {s10, s11}
                    }
                    __builtin_amdgcn_sched_barrier(0);
                }
#pragma unroll
                for (int s = 0; s < E; ++s) {                // phase 2: interpolate, (write,) accumulate
                    const int e = es + ES * (c0 + q0 + s);
                    const float4 r0 = s_ray[wave][q0 + s][0], r1 = s_ray[wave][q0 + s][1];
                    const float dx = SRC ? x - r1.y : fmaf(igf - r1.y, P.hx, -r0.y), dy = SRC ? y - r1.z : fmaf(jgf - r1.z, P.hy, -r0.z);
                    float2u_t sa = {0.f, 0.f};               // { sum sig, sum a' } at the crossing: (s, a) pairs in packed fp32
                    if constexpr (SRC) {
                        if constexpr (ONE) {
                            const float c0v = fmaf(fv[s], lo[s].y - lo[s].x, lo[s].x), c1v = fmaf(fv[s], hi[s].y - hi[s].x, hi[s].x);
                            sa.x = fmaf(fu[s], c1v - c0v, c0v);
                        } else {
                            const float2u_t l0 = {lo[s].x, lo[s].y}, l1 = {lo[s].z, lo[s].w}, h0 = {hi[s].x, hi[s].y}, h1 = {hi[s].z, hi[s].w};
                            const float2u_t c0v = fv[s] * (l1 - l0) + l0, c1v = fv[s] * (h1 - h0) + h0;
                            sa = fu[s] * (c1v - c0v) + c0v;
                        }
                    }
                    if (writer && OLX_IN((long long)e * H.nyg + own, (long long)H.nxg * row_cells, 7)) {
                        if constexpr (ONE) (reinterpret_cast<float*>(U_dst) + (size_t)e * H.nyg)[own] = sa.x + sv2;
                        else (U_dst + (size_t)e * H.nyg)[own] = make_float2(sa.x + sv2, sa.y + av2);
                    }
                    float d2 = fmaf(dy, dy, fmaf(dx, dx, r1.x));
                    if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                    const float ri = __builtin_amdgcn_rsqf(d2);
                    const float d = d2 * ri;
                    const float l = d * r0.w;                // path per layer [wavelengths]; 0 level with the element
                    const float ssum = ES > 1 ? sa.x + sv : sa.x;
                    const float q = l * ssum;                // extra acoustic path of the ray [wavelengths]
                    const float ph0 = d + q;
                    // exp(-l (sum a' + a'(v) / 2)) as one exp2: ONE: a' = kappa sig -> exp2(k2 q), k2 = -kappa log2(e); with a single focus per
                    // tile its weight rides in the exponent too (log2 w from the ray table; w = 0 -> -inf -> 0)
                    constexpr bool WEXP = ONE && NF == 1 && RW > 0;
                    float amp;
                    if constexpr (WEXP) amp = ri * __builtin_amdgcn_exp2f(fmaf(k2, q, s_ray[wave][q0 + s][2].x));
                    else if constexpr (ONE) amp = ri * __builtin_amdgcn_exp2f(k2 * q);
                    else amp = ri * __builtin_amdgcn_exp2f(-1.4426950408889634f * l * (ES > 1 ? sa.y + av : sa.y));
                    float wf[2 * NF];                         // { w_f, phi_f }: from the ray table (ES = 1) or the steering table
                    if constexpr (RW > 0) {
#pragma unroll
                        for (int r = 0; r < RW; ++r) {
                            const float4 w = s_ray[wave][q0 + s][2 + r];
                            wf[4 * r] = w.x; wf[4 * r + 1] = w.y;
                            if (2 * r + 1 < NF) { wf[4 * r + 2] = w.z; wf[4 * r + 3] = w.w; }
                        }
                    } else {
                        const float* te = t + (size_t)e * STRIDE;
#pragma unroll
                        for (int f = 0; f < 2 * NF; ++f) wf[f] = te[HET_TAB_HEAD + f];
                    }
#pragma unroll
                    for (int f = 0; f < NF; ++f) {           // the ray sums above serve every focus of the tile
                        const float ph = ph0 + wf[2 * f + 1];
                        const float a = WEXP ? amp : amp * wf[2 * f];
                        re[f] = fmaf(a, __builtin_amdgcn_cosf(ph), re[f]);
                        im[f] = fmaf(a, __builtin_amdgcn_sinf(ph), im[f]);
                    }
                }
            };
            int q0 = 0;
            for (; q0 + EU <= n_ch; q0 += EU) group(q0, std::integral_constant<int, EU>{});
            for (; q0 < n_ch; ++q0) group(q0, std::integral_constant<int, 1>{});
        }
    }
    if constexpr (ES > 1) {                                  // partial sums of the element subsets meet in wave 0
        if (es > 0) {
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                s_red[(((es - 1) * NF + f) * 2 + 0) * 64 + lane] = re[f];
                s_red[(((es - 1) * NF + f) * 2 + 1) * 64 + lane] = im[f];
            }
        }
        __syncthreads();
        if (es > 0) return;
#pragma unroll 1
        for (int q = 0; q < ES - 1; ++q)
#pragma unroll
            for (int f = 0; f < NF; ++f) {
                re[f] += s_red[((q * NF + f) * 2 + 0) * 64 + lane];
                im[f] += s_red[((q * NF + f) * 2 + 1) * 64 + lane];
            }
    }
    const int il = ig - P.x_begin;                           // slab-local x
    if (!live_k || !in_grid || il < 0 || il >= P.nx) return;
    const long long vrow = ((long long)il * P.ny + j) * P.nz + k;
#pragma unroll
    for (int f = 0; f < NF; ++f) {
        const int fg = ftile * NF + f;
        if (fg >= H.n_foci) break;
        const long long o = (long long)fg * P.vox + vrow;
        if (!OLX_IN(o, (long long)H.n_foci * P.vox, 8)) continue;
        const float m2 = fmaf(re[f], re[f], im[f] * im[f]);
        if (P.flags & 1u) pmag[o] = __builtin_sqrtf(m2);
        if (P.flags & 2u) inten[o] = m2 * (inv2z ? inv2z[vrow] : P.inten_scale);
        if (P.flags & 4u) { cplx[2 * o] = re[f]; cplx[2 * o + 1] = im[f]; }
    }
}

// ------------------------------------------------------------------------------------
// FUSED WRITERS (round 6): G consecutive non-trivial planes per launch, the running sums in between carried in LDS.
// A writer launch of round 5 read U_p and wrote U_{p+1} over the whole lateral grid for ONE plane: 134 MB of HBM traffic per plane for 16.7 M
// look-ups, 31 times per focus -- 4.65 of the 7.26 GB a focus moved, and bandwidth-bound (33 us per launch against 13 us for the same number of
// look-ups above the medium).  Here a block takes a 16 x 16 lateral tile T through the G planes m_{p+1} .. m_{p+G} of a run of CONSECUTIVE grid
// planes, per element:
//   level 0   the cells of U_p that the chain needs, read once from HBM into the wave's arena;
//   level g   U_{p+g} = B[U_{p+g-1}](crossing) + med(., m_{p+g}) on the rectangle H_g, from LDS to LDS;  H_G = T,  H_{g-1} = T u stencil cells of H_g:
//             the crossing map is a contraction towards the element, so H_g is T stretched towards it by (1 - tt) |i - e_u| cells per level
//             (<= 6.6 per level at the corner of BASELINE's configs[4], 2.6 on average: 1.7 x the look-ups of four separate launches);
//   the cells of T take part at EVERY level: their look-up value + the plane's own half term is the ray sum of voxel (i, j, m_{p+g}) -- the field
//   of the G planes is accumulated over the elements in registers (4 cells per lane x G planes), exactly as the single-plane writers do;
//   level G   U_{p+G} on T goes to HBM: one read of ~U and one write of U per G planes.
// Same arithmetic per look-up and per pair as field_hmarch_k<NF = 1, ES = 4, ONE> (same fmas in the same order, the same split of the elements
// over four waves): bit-identical volumes.  One-sum form, one focus per launch tile, a source plane below the group; everything else keeps the
// single-plane launches.  The own terms come from the compact copy sig[plane][i][j] (olx_field_set_medium).
// ------------------------------------------------------------------------------------
constexpr int HF_T = 16;                   // lateral tile
constexpr int HF_WAVES = 4;                // element subsets per block (as the one-sum writers)
constexpr int HF_ECH = 16;                 // elements per chunk of a wave's ray table
constexpr int HF_GMAX = 4;
struct FusedSeg {
    int k_src;           // grid plane of the source sums U_src
    int k_first;         // planes k_first .. k_first + G - 1 (consecutive grid planes, all non-trivial)
    int pq_first;        // index of k_first among the non-trivial planes (sig[pq_first + g - 1])
    unsigned nblocks;    // logical blocks = lateral tiles
    int reverse;
    int cap;             // capacity of one arena [words]; a wave owns two
};
struct HfRay {             // per element of a chunk: what depends on (element, level) and (element, tile) only
    float ex, ey, w, phi;
    float tt[HF_GMAX], cu[HF_GMAX], cv[HF_GMAX], lf[HF_GMAX], dz2[HF_GMAX];
    int lx[HF_GMAX + 1], ly[HF_GMAX + 1], nxr[HF_GMAX + 1], nyr[HF_GMAX + 1];      // rectangle H_g: first cell, rows, columns
};

template <int G, int TI, bool CLAMP, bool INSIDE>
__global__ __launch_bounds__(64 * HF_WAVES, TI == 4 ? 6 : 2) void field_hmarch_fused_k(
    const float* __restrict__ tab, const float* __restrict__ sig, const float* __restrict__ U_src, float* __restrict__ U_dst,
    const float* __restrict__ inv2z, float* __restrict__ pmag, float* __restrict__ inten, float* __restrict__ cplx, const FieldParams P,
    const HeteroParams H, const FusedSeg S) {
    static_assert(G >= 2 && G <= HF_GMAX, "2 .. 4 planes per launch");
    static_assert(TI == 4 || TI == 16, "lateral tile TI x 16: one or four cells per lane");
    constexpr int SL = TI / 4;                        // cells of T per lane
    constexpr int STRIDE = HET_TAB_HEAD + 2;          // one focus per launch tile
    extern __shared__ __attribute__((aligned(16))) unsigned char hf_smem[];
    HfRay* const s_ray = reinterpret_cast<HfRay*>(hf_smem);                                   // [HF_WAVES][HF_ECH]
    float* const s_arena = reinterpret_cast<float*>(hf_smem + sizeof(HfRay) * HF_WAVES * HF_ECH);   // [HF_WAVES][2][cap]; afterwards the partial sums
    const int ftile = blockIdx.y;
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int tiles_y = (H.nyg + HF_T - 1) / HF_T;
    unsigned lb = (blockIdx.x & 7u) * (gridDim.x >> 3) + (blockIdx.x >> 3);      // one contiguous eighth of the tiles per XCD (as field_hmarch_k)
    if (lb >= S.nblocks) return;
    if (S.reverse) lb = S.nblocks - 1u - lb;
    const int ti = (int)(lb / (unsigned)tiles_y), tj = (int)(lb - (unsigned)ti * (unsigned)tiles_y);
    const int tax = ti * TI, tay = tj * HF_T, tbx = min(tax + TI - 1, H.nxg - 1), tby = min(tay + HF_T - 1, H.nyg - 1);      // T (inclusive)
    // this lane's four cells of T: rows (lane >> 4) + 4 s, column lane & 15 (clamped copies at a ragged edge: computed, never stored)
    const int jraw = tay + (lane & 15), jc = min(jraw, H.nyg - 1);
    const float jgf = (float)jc, y = jgf * P.hy;
    int iraw[SL], icl[SL];
    float igf[SL], x[SL];
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        iraw[s] = tax + (lane >> 4) + 4 * s; icl[s] = min(iraw[s], H.nxg - 1);
        igf[s] = (float)icl[s]; x[s] = igf[s] * P.hx;
    }
    float re[SL][G], im[SL][G];
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int g = 0; g < G; ++g) { re[s][g] = 0.f; im[s][g] = 0.f; }
    // the planes' own half terms of this lane's cells
    float sv[SL][G];
#pragma unroll
    for (int s = 0; s < SL; ++s)
#pragma unroll
        for (int g = 0; g < G; ++g) sv[s][g] = 0.5f * sig[((size_t)(S.pq_first + g) * H.nxg + icl[s]) * H.nyg + jc];
    const unsigned row_cells = (unsigned)P.n_el * (unsigned)H.nyg;         // U is [i][element][j]
    const float umax = (float)(H.nxg - 1) * (1.f - 0x1p-23f), vmax = (float)(H.nyg - 1) * (1.f - 0x1p-23f);
    const float k2 = -1.4426950408889634f * H.kappa;
    const float* t = tab + (size_t)ftile * P.n_el * STRIDE;
    const bool writer = ftile == 0;
    const int es = wave;
    HfRay* const my_ray = s_ray + wave * HF_ECH;
    float* const arena = s_arena + (size_t)wave * 2 * S.cap;
    const int n_mine = (P.n_el - es + HF_WAVES - 1) / HF_WAVES;           // elements es, es + 4, ...
    auto cross = [&](const float tt, const float c, const float idx, const float cmax) __attribute__((always_inline)) {
        float u = fmaf(tt, idx, c);
        if constexpr (!INSIDE) u = __builtin_amdgcn_fmed3f(u, 0.f, cmax);
        return u;
    };
    for (int c0 = 0; c0 < n_mine; c0 += HF_ECH) {
        const int n_ch = min(HF_ECH, n_mine - c0);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        if (lane < n_ch) {       // ray table of this chunk: one element per lane
            const float* te = t + (size_t)(es + HF_WAVES * (c0 + lane)) * STRIDE;
            const float ex = te[0], ey = te[1], ez = te[2];
            const float eu = fmaf(ex, H.inv_hx, H.u0), ev = fmaf(ey, H.inv_hy, H.v0);      // element in grid cells
            HfRay R;
            R.ex = ex; R.ey = ey; R.w = te[HET_TAB_HEAD]; R.phi = te[HET_TAB_HEAD + 1];
            float zprev = (float)S.k_src * P.hz;
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const float z = (float)(S.k_first + g) * P.hz;
                const float dz = z - ez;
                const float idz = dz != 0.f ? 1.0f / dz : 0.f;
                const float tt = (zprev - ez) * idz;
                R.tt[g] = tt; R.cu[g] = eu - tt * eu; R.cv[g] = ev - tt * ev; R.lf[g] = P.hz * fabsf(idz); R.dz2[g] = dz * dz;
                zprev = z;
            }
            // rectangles, from the top level down: H_G = T, H_{g-1} = T u stencil cells of H_g (the crossing map is monotonic in the cell index)
            int lx = tax, rx = tbx, ly = tay, ry = tby;
            R.lx[G] = lx; R.ly[G] = ly; R.nxr[G] = rx - lx + 1; R.nyr[G] = ry - ly + 1;
#pragma unroll
            for (int g = G; g >= 1; --g) {
                const int l0 = (int)cross(R.tt[g - 1], R.cu[g - 1], (float)lx, umax), r0 = (int)cross(R.tt[g - 1], R.cu[g - 1], (float)rx, umax) + 1;
                const int l1 = (int)cross(R.tt[g - 1], R.cv[g - 1], (float)ly, vmax), r1 = (int)cross(R.tt[g - 1], R.cv[g - 1], (float)ry, vmax) + 1;
                lx = l0; rx = r0; ly = l1; ry = r1;
                if (g - 1 >= 1) { lx = min(lx, tax); rx = max(rx, tbx); ly = min(ly, tay); ry = max(ry, tby); }
                R.lx[g - 1] = lx; R.ly[g - 1] = ly; R.nxr[g - 1] = rx - lx + 1; R.nyr[g - 1] = ry - ly + 1;
            }
            my_ray[lane] = R;
        }
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        // ---- level 0: the source sums of H_0 come from HBM (rows of consecutive j) ONE ELEMENT AHEAD, into registers: a wave walks its elements one
        // after the other and only two waves share a SIMD (the arenas fill the LDS), so a load waited for on the spot costs its whole latency
        constexpr int HF_PRE = TI == 4 ? 12 : 24;        // registers of the prefetch: rectangles up to 768 / 1536 cells (larger ones load the rest on the spot)
        float pre[HF_PRE];
        auto fetch0 = [&](const int q) __attribute__((always_inline)) {
            const HfRay& R = my_ray[q];
            const int lx = __builtin_amdgcn_readfirstlane(R.lx[0]), ly = __builtin_amdgcn_readfirstlane(R.ly[0]);
            const int nyr = __builtin_amdgcn_readfirstlane(R.nyr[0]), n0 = __builtin_amdgcn_readfirstlane(R.nxr[0]) * nyr;
            const float inv = 1.0f / (float)nyr;
            const float* Ue = U_src + (size_t)(es + HF_WAVES * (c0 + q)) * H.nyg;
#pragma unroll
            for (int k = 0; k < HF_PRE; ++k) {
                const int idx = lane + 64 * k;
                const int r = (int)(((float)idx + 0.5f) * inv), cidx = idx - r * nyr;       // exact for these small integers
                const unsigned off = (unsigned)(lx + r) * row_cells + (unsigned)(ly + cidx);
                pre[k] = 0.f;
                if (idx < n0 && OLX_IN((long long)(es + HF_WAVES * (c0 + q)) * H.nyg + off, (long long)H.nxg * row_cells, 9)) pre[k] = Ue[off];
            }
        };
        auto store0 = [&](const int q) __attribute__((always_inline)) {
            const HfRay& R = my_ray[q];
            const int lx = __builtin_amdgcn_readfirstlane(R.lx[0]), ly = __builtin_amdgcn_readfirstlane(R.ly[0]);
            const int nyr = __builtin_amdgcn_readfirstlane(R.nyr[0]), n0 = __builtin_amdgcn_readfirstlane(R.nxr[0]) * nyr;
#pragma unroll
            for (int k = 0; k < HF_PRE; ++k) {
                const int idx = lane + 64 * k;
                if (idx < n0 && OLX_IN(idx, S.cap, 10)) arena[idx] = pre[k];
            }
            if (n0 > 64 * HF_PRE) {                      // (a rectangle beyond the prefetch registers: the rest on the spot)
                const float inv = 1.0f / (float)nyr;
                const float* Ue = U_src + (size_t)(es + HF_WAVES * (c0 + q)) * H.nyg;
                for (int idx = lane + 64 * HF_PRE; idx < n0; idx += 64) {
                    const int r = (int)(((float)idx + 0.5f) * inv), cidx = idx - r * nyr;
                    const unsigned off = (unsigned)(lx + r) * row_cells + (unsigned)(ly + cidx);
                    if (OLX_IN(idx, S.cap, 10) && OLX_IN((long long)(es + HF_WAVES * (c0 + q)) * H.nyg + off, (long long)H.nxg * row_cells, 9)) arena[idx] = Ue[off];
                }
            }
        };
        fetch0(0);
        for (int q = 0; q < n_ch; ++q) {
            const int e = es + HF_WAVES * (c0 + q);
            const HfRay& R = my_ray[q];                  // (same address in every lane: LDS broadcasts)
            store0(q);                                   // (the previous element's last level read the OTHER arena, or is done with this one: in-order LDS)
            if (q + 1 < n_ch) fetch0(q + 1);             // in flight during this element's levels
#pragma unroll
            for (int g = 1; g <= G; ++g) {
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                const float* const src = arena + ((g - 1) & 1) * S.cap;
                float* const dst = arena + (g & 1) * S.cap;
                const float tt = R.tt[g - 1], cu = R.cu[g - 1], cv = R.cv[g - 1];
                const int slx = __builtin_amdgcn_readfirstlane(R.lx[g - 1]), sly = __builtin_amdgcn_readfirstlane(R.ly[g - 1]);
                const int snyr = __builtin_amdgcn_readfirstlane(R.nyr[g - 1]), scells = __builtin_amdgcn_readfirstlane(R.nxr[g - 1]) * snyr;
                const int dlx = __builtin_amdgcn_readfirstlane(R.lx[g]), dly = __builtin_amdgcn_readfirstlane(R.ly[g]);
                const int dnyr = __builtin_amdgcn_readfirstlane(R.nyr[g]), dnxr = __builtin_amdgcn_readfirstlane(R.nxr[g]);
                // one look-up: the running sums of level g - 1 at the crossing of the ray element -> (i, j, plane g) with the plane below
                auto lookup = [&](const float fi, const float fj) __attribute__((always_inline)) {
                    const float u = cross(tt, cu, fi, umax), v = cross(tt, cv, fj, vmax);
                    const int i0 = (int)u, j0 = (int)v;
                    const float fu = __builtin_amdgcn_fractf(u), fv = __builtin_amdgcn_fractf(v);
                    const int o = (i0 - slx) * snyr + (j0 - sly);
                    float s00 = 0.f, s01 = 0.f, s10 = 0.f, s11 = 0.f;
                    if (OLX_IN(o, scells - snyr - 1, 11)) { s00 = src[o]; s01 = src[o + 1]; s10 = src[o + snyr]; s11 = src[o + snyr + 1]; }
                    const float c0v = fmaf(fv, s01 - s00, s00), c1v = fmaf(fv, s11 - s10, s10);
                    return fmaf(fu, c1v - c0v, c0v);
                };
                // the cells of T: look-up, next running sum, and the pair's term of the field of plane k_first + g - 1
                const float dz2 = R.dz2[g - 1], lf = R.lf[g - 1], ex = R.ex, ey = R.ey, w = R.w, phi = R.phi;
                const float dy = y - ey;
#pragma unroll
                for (int s = 0; s < SL; ++s) {
                    const float sa = lookup(igf[s], jgf);
                    const float un = sa + 2.f * sv[s][g - 1];
                    if (g < G) {
                        const int o = (icl[s] - dlx) * dnyr + (jc - dly);
                        if (OLX_IN(o, S.cap, 10)) dst[o] = un;
                    } else if (writer && iraw[s] < H.nxg && jraw < H.nyg) {
                        const unsigned off = (unsigned)iraw[s] * row_cells + (unsigned)jraw;
                        if (OLX_IN((long long)e * H.nyg + off, (long long)H.nxg * row_cells, 7)) (U_dst + (size_t)e * H.nyg)[off] = un;
                    }
                    const float dx = x[s] - ex;
                    float d2 = fmaf(dy, dy, fmaf(dx, dx, dz2));
                    if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                    const float ri = __builtin_amdgcn_rsqf(d2);
                    const float d = d2 * ri;
                    const float l = d * lf;
                    const float qq = l * (sa + sv[s][g - 1]);
                    const float ph = (d + qq) + phi;
                    const float a = (ri * __builtin_amdgcn_exp2f(k2 * qq)) * w;
                    re[s][g - 1] = fmaf(a, __builtin_amdgcn_cosf(ph), re[s][g - 1]);
                    im[s][g - 1] = fmaf(a, __builtin_amdgcn_sinf(ph), im[s][g - 1]);
                }
                if (g < G) {
                    // the rest of H_g (four bands around T): look-up + own term -> next running sum
                    const int ra = tax - dlx, rb = dlx + dnxr - 1 - tbx, rt = tbx - tax + 1, cl = tay - dly, cr = dly + dnyr - 1 - tby;
                    const int nA = ra * dnyr, nB = rb * dnyr, nC = rt * cl, nD = rt * cr, nX = nA + nB + nC + nD;
                    const float inv_w = 1.0f / (float)dnyr, inv_l = cl > 0 ? 1.0f / (float)cl : 0.f, inv_r = cr > 0 ? 1.0f / (float)cr : 0.f;
                    const float* const sg_p = sig + (size_t)(S.pq_first + g - 1) * H.nxg * H.nyg;
                    for (int idx = lane; idx < nX; idx += 64) {
                        int i, j;
                        if (idx < nA + nB) {
                            const int tq = idx < nA ? idx : idx - nA;
                            const int r = (int)(((float)tq + 0.5f) * inv_w);
                            i = (idx < nA ? dlx : tbx + 1) + r; j = dly + (tq - r * dnyr);
                        } else if (idx < nA + nB + nC) {
                            const int tq = idx - nA - nB, r = (int)(((float)tq + 0.5f) * inv_l);
                            i = tax + r; j = dly + (tq - r * cl);
                        } else {
                            const int tq = idx - nA - nB - nC, r = (int)(((float)tq + 0.5f) * inv_r);
                            i = tax + r; j = tby + 1 + (tq - r * cr);
                        }
                        const float sa = lookup((float)i, (float)j);
                        float own = 0.f;
                        if (OLX_IN((long long)i * H.nyg + j, (long long)H.nxg * H.nyg, 12)) own = sg_p[(size_t)i * H.nyg + j];
                        const int o = (i - dlx) * dnyr + (j - dly);
                        if (OLX_IN(o, S.cap, 10)) dst[o] = sa + 2.f * (0.5f * own);
                    }
                }
            }
        }
    }
    // ---- partial sums of the four element subsets meet in wave 0 (the arenas are free now), in the single-plane writers' order
    __syncthreads();
    float* const s_red = s_arena;
    if (es > 0) {
#pragma unroll
        for (int s = 0; s < SL; ++s)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                s_red[((((es - 1) * SL + s) * G + g) * 2 + 0) * 64 + lane] = re[s][g];
                s_red[((((es - 1) * SL + s) * G + g) * 2 + 1) * 64 + lane] = im[s][g];
            }
    }
    __syncthreads();
    if (es > 0) return;
#pragma unroll 1
    for (int q = 0; q < HF_WAVES - 1; ++q)
#pragma unroll
        for (int s = 0; s < SL; ++s)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                re[s][g] += s_red[(((q * SL + s) * G + g) * 2 + 0) * 64 + lane];
                im[s][g] += s_red[(((q * SL + s) * G + g) * 2 + 1) * 64 + lane];
            }
    if (jraw >= P.ny) return;
#pragma unroll
    for (int s = 0; s < SL; ++s) {
        const int il = iraw[s] - P.x_begin;                  // slab-local x
        if (iraw[s] >= H.nxg || il < 0 || il >= P.nx) continue;
#pragma unroll
        for (int g = 0; g < G; ++g) {
            const long long vrow = ((long long)il * P.ny + jraw) * P.nz + (S.k_first + g);
            const long long o = (long long)ftile * P.vox + vrow;
            if (ftile >= H.n_foci || !OLX_IN(o, (long long)H.n_foci * P.vox, 8)) continue;
            const float m2 = fmaf(re[s][g], re[s][g], im[s][g] * im[s][g]);
            if (P.flags & 1u) pmag[o] = __builtin_sqrtf(m2);
            if (P.flags & 2u) inten[o] = m2 * (inv2z ? inv2z[vrow] : P.inten_scale);
            if (P.flags & 4u) { cplx[2 * o] = re[s][g]; cplx[2 * o + 1] = im[s][g]; }
        }
    }
}

// U [i][element][j] float  ->  T [i][element][j] float2 = {U(i,j), U(i+1,j)}, the last row repeated (the look-ups stay just inside the last cell, so
// the repeated value carries weight ~0).  One thread per cell, 8-byte stores; one cell of padding behind the array (the 16-byte load of the last cell).
__global__ __launch_bounds__(256) void u_texel_k(const float* __restrict__ U, float2* __restrict__ T, int nxg, int n_el, int nyg) {
    const long long cells = (long long)nxg * n_el * nyg;
    const long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (q > cells) return;
    if (q == cells) { T[q] = make_float2(0.f, 0.f); return; }
    const long long r = q / nyg;            // i * n_el + e
    const int i = (int)(r / n_el);
    const long long up = i + 1 < nxg ? (long long)n_el * nyg : 0;
    T[q] = make_float2(U[q], U[q + up]);
}

}  // namespace olx

using namespace olx;
OLX_BOUNDS_READER(hmarch)

// Arena capacity [words] of a fused launch over planes k_first .. k_first + G - 1 with the source sums on plane k_src: the largest rectangle any
// (tile, element) pair needs, from the kernel's own recursion evaluated in fp64 for the tiles at and next to the four corners of the lateral grid
// (the rectangles grow with the distance between tile and element), + 3 cells per axis for the kernel's fp32 rounding of the same expressions.
static int fused_capacity(const olx_ctx* c, int k_src, int k_first, int G, int TI) {
    const int n = c->n_el, nxg = c->hp.nxg, nyg = c->hp.nyg;
    const double hz = c->grid.spacing[2];
    const int ntx = (nxg + TI - 1) / TI, nty = (nyg + HF_T - 1) / HF_T;
    long long best = 0;
    for (int e = 0; e < n; ++e) {
        const double eu = (c->h_pos[e] - c->grid.origin[0]) / c->grid.spacing[0], ev = (c->h_pos[(size_t)n + e] - c->grid.origin[1]) / c->grid.spacing[1];
        const double ez = c->h_pos[2 * (size_t)n + e] - c->grid.origin[2];
        double tt[HF_GMAX];
        double zprev = k_src * hz;
        for (int g = 0; g < G; ++g) { const double z = (k_first + g) * hz; tt[g] = (zprev - ez) / (z - ez); zprev = z; }
        for (int a = 0; a < 4; ++a)
            for (int b = 0; b < 4; ++b) {
                const int ti = a < 2 ? std::min(a, ntx - 1) : std::max(0, ntx - 1 - (a - 2)), tj = b < 2 ? std::min(b, nty - 1) : std::max(0, nty - 1 - (b - 2));
                const int tax = ti * TI, tay = tj * HF_T, tbx = std::min(tax + TI - 1, nxg - 1), tby = std::min(tay + HF_T - 1, nyg - 1);
                int lx = tax, rx = tbx, ly = tay, ry = tby;
                for (int g = G; g >= 1; --g) {
                    auto map = [&](double idx, double ee, double hi) { return std::min(std::max(tt[g - 1] * idx + (ee - tt[g - 1] * ee), 0.0), hi); };
                    lx = (int)map(lx, eu, nxg - 1.0); rx = (int)map(rx, eu, nxg - 1.0) + 1; ly = (int)map(ly, ev, nyg - 1.0); ry = (int)map(ry, ev, nyg - 1.0) + 1;
                    if (g - 1 >= 1) { lx = std::min(lx, tax); rx = std::max(rx, tbx); ly = std::min(ly, tay); ry = std::max(ry, tby); }
                    best = std::max(best, (long long)(rx - lx + 1 + 3) * (ry - ly + 1 + 3));
                }
            }
    }
    return (int)std::min<long long>(best, 1 << 20);
}

template <int NF>
static void launch_hmarch_nf(olx_ctx* c, float* pm) {
    const FieldParams& P = c->fp;
    const int nz = P.nz, np = (int)c->h_plane_k.size();
    const int ftiles = (c->plan_foci + NF - 1) / NF;
    // every element at least half a cell inside the lateral grid -> no crossing point needs a clamp (the look-up coordinates are convex
    // combinations of element and voxel positions; half a cell dwarfs their fp32 rounding)
    bool inside = true;
    for (int e = 0; e < c->n_el && inside; ++e) {
        const double eu = (c->h_pos[e] - c->grid.origin[0]) / c->grid.spacing[0], ev = (c->h_pos[(size_t)c->n_el + e] - c->grid.origin[1]) / c->grid.spacing[1];
        inside = eu >= 0.5 && eu <= c->hp.nxg - 1.5 && ev >= 0.5 && ev <= c->hp.nyg - 1.5;
    }
    int fused_planes = 0, fused_launches = 0;
    int ub = 0;           // the U buffer that holds the latest running sums (a writer reads d_U[ub], writes d_U[ub ^ 1])
    int n_written = 0;    // writer launches so far (consecutive ones walk the blocks in alternate directions)
    // p_src: index of the non-trivial plane the source sums live on (-1: none); write: k_lo == k_hi is non-trivial, the next sums go to the other buffer
    auto go = [&](int k_lo, int k_hi, int p_src, bool write, bool tex = false) {
        if (k_hi < k_lo) return;
        MarchSeg S;
        S.k_lo = k_lo; S.k_hi = k_hi; S.k_src = p_src >= 0 ? c->h_plane_k[p_src] : -1; S.write = write ? 1 : 0;
        S.i0 = write ? 0 : c->slab.x_begin; S.ni = write ? c->hp.nxg : P.nx;
        S.reverse = write && (n_written & 1);
        const float2* src = tex ? reinterpret_cast<const float2*>(c->d_Utex) : (p_src >= 0 ? c->d_U[ub] : nullptr);
        float2* dst = write ? c->d_U[p_src >= 0 ? ub ^ 1 : ub] : nullptr;       // (the first non-trivial plane has no source: it fills d_U[ub] itself)
        const long long tiles = (long long)((S.ni + HM_TI - 1) / HM_TI) * ((P.ny + HM_TJ - 1) / HM_TJ);
#define OLX_HM___(ES_, CL, SR, ON, IN) S.nblocks = (unsigned)(tiles * ((ES_ == 1) ? (k_hi - k_lo + 4) / 4 : 1)); \
                             hipLaunchKernelGGL((field_hmarch_k<NF, ES_, CL, SR, ON, IN>), dim3((S.nblocks + 7u) / 8u * 8u, ftiles), \
                                           dim3(64 * hm_waves<ES_>()), 0, c->stream, c->d_tab, c->d_med, c->d_plane_of_k, src, dst, c->d_inv2z, pm, \
                                           c->d_inten, c->d_cplx, P, c->hp, S)
#define OLX_HM__(ES_, CL, SR, ON) do { if (inside) { OLX_HM___(ES_, CL, SR, ON, true); } else { OLX_HM___(ES_, CL, SR, ON, false); } } while (0)
#define OLX_HM_(ES_, CL, SR) do { if (c->march_one) { OLX_HM__(ES_, CL, SR, true); } else { OLX_HM__(ES_, CL, SR, false); } } while (0)
#define OLX_HM(ES_, CL) do { if (p_src >= 0) { OLX_HM_(ES_, CL, true); } else { OLX_HM_(ES_, CL, false); } } while (0)
        // element subsets per writer block: 16 for the two-sum form (4, 8, 16 measured the same in round 3: HBM-bound on U), 4 for the one-sum form
        // (round 5, same box, alternating: 4.18 / 3.99 / 3.87 / 4.38 ms per focus with 16 / 8 / 4 / 2 subsets -- with half the bytes the launch
        // is no longer bandwidth-bound and 256-thread blocks of 64 elements per wave keep more independent work per CU)
        if (write) {
            if (c->march_one) { if (c->clamp) OLX_HM(4, true); else OLX_HM(4, false); }
            else              { if (c->clamp) OLX_HM(16, true); else OLX_HM(16, false); }
            if (p_src >= 0) ub ^= 1;
            ++n_written;
        }
        else if (tex) {   // one-sum look-ups out of the texel form of the last running sums
#define OLX_HMT(CL, IN) hipLaunchKernelGGL((field_hmarch_k<NF, 1, CL, true, true, IN, true>), dim3((S.nblocks + 7u) / 8u * 8u, ftiles), dim3(64 * hm_waves<1>()), 0, \
                                           c->stream, c->d_tab, c->d_med, c->d_plane_of_k, src, dst, c->d_inv2z, pm, c->d_inten, c->d_cplx, P, c->hp, S)
            S.nblocks = (unsigned)(tiles * ((k_hi - k_lo + 4) / 4));
            if (c->clamp) { if (inside) OLX_HMT(true, true); else OLX_HMT(true, false); }
            else          { if (inside) OLX_HMT(false, true); else OLX_HMT(false, false); }
#undef OLX_HMT
        }
        else       { if (c->clamp) OLX_HM(1, true); else OLX_HM(1, false); }
#undef OLX_HM
#undef OLX_HM_
#undef OLX_HM__
#undef OLX_HM___
    };
    // fused writers (field_hmarch_fused_k): planes p .. p + G - 1 of a run of consecutive grid planes in ONE launch.  One-sum form, one focus per
    // launch tile, a source plane below.  OPT-IN (OLX_MARCH_FUSE=2 / 3 / 4 planes per launch; same bits as the single-plane writers): they cut the
    // writers' HBM traffic by G but measured SLOWER on BASELINE configs[4] -- 6.45 / 6.22 / 6.15 ms per focus with 4 / 3 / 2 planes per launch on
    // 16 x 16 tiles, 5.50 with 2 planes on 4 x 16 tiles (OLX_MARCH_FUSE_TI=4), against 3.85 ms (profiles/r06_hmarch_fused.txt): a wave walks its
    // elements one after the other and every (element, level) step is one or two wave-iterations of work behind its own set-up -- 2.4 to 4.7
    // times the instructions of the single-plane writers, whose 33 us per plane (bandwidth-bound) the fused form would have to beat.
    int gmax = 1;
    if (const char* e = getenv("OLX_MARCH_FUSE")) { const int v = atoi(e); gmax = (NF == 1 && c->march_one && c->d_sig && v >= 2 && v <= HF_GMAX) ? v : 1; }
    int fti = 16;          // rows of the fused writers' lateral tile (x 16 columns): OLX_MARCH_FUSE_TI=4 pins the single-plane writers' 4 x 16 tile (A/B)
    if (const char* e = getenv("OLX_MARCH_FUSE_TI")) fti = atoi(e) == 4 ? 4 : 16;
    auto go_fused = [&](int p_first, int G) -> bool {      // false: does not fit (the caller takes single planes)
        FusedSeg S;
        S.k_src = c->h_plane_k[p_first - 1]; S.k_first = c->h_plane_k[p_first]; S.pq_first = p_first;
        S.cap = std::max(fused_capacity(c, S.k_src, S.k_first, G, fti), (HF_WAVES - 1) * (fti / 4) * G * 2 * 64 / (HF_WAVES * 2) + 1);     // (the partial sums reuse the arenas)
        const size_t lds = sizeof(HfRay) * HF_WAVES * HF_ECH + sizeof(float) * (size_t)HF_WAVES * 2 * S.cap;
        if (lds > 78 * 1024) return false;             // two blocks per CU
        S.nblocks = (unsigned)(((c->hp.nxg + fti - 1) / fti) * ((c->hp.nyg + HF_T - 1) / HF_T));
        S.reverse = n_written & 1;
        const float* src = reinterpret_cast<const float*>(c->d_U[ub]);
        float* dst = reinterpret_cast<float*>(c->d_U[ub ^ 1]);
        const dim3 grid((S.nblocks + 7u) / 8u * 8u, ftiles), blk(64 * HF_WAVES);
#define OLX_HF___(G_, TI_, CL, IN) do { auto kern = field_hmarch_fused_k<G_, TI_, CL, IN>; \
            static size_t attr_set = 0; if (lds > attr_set) { hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds); attr_set = lds; } \
            hipLaunchKernelGGL(kern, grid, blk, lds, c->stream, c->d_tab, c->d_sig, src, dst, c->d_inv2z, pm, c->d_inten, c->d_cplx, P, c->hp, S); } while (0)
#define OLX_HF__(G_, CL, IN) do { if (fti == 4) OLX_HF___(G_, 4, CL, IN); else OLX_HF___(G_, 16, CL, IN); } while (0)
#define OLX_HF_(G_) do { if (c->clamp) { if (inside) OLX_HF__(G_, true, true); else OLX_HF__(G_, true, false); } \
                         else          { if (inside) OLX_HF__(G_, false, true); else OLX_HF__(G_, false, false); } } while (0)
        if (G == 2) OLX_HF_(2); else if (G == 3) OLX_HF_(3); else OLX_HF_(4);
#undef OLX_HF_
#undef OLX_HF__
#undef OLX_HF___
        ub ^= 1; ++n_written;
        fused_planes += G; ++fused_launches;
        return true;
    };
    if (np == 0) { go(0, nz - 1, -1, false); return; }
    go(0, c->h_plane_k[0] - 1, -1, false);                   // below the first non-trivial plane: homogeneous rays
    go(c->h_plane_k[0], c->h_plane_k[0], -1, true);          // U_0 = the plane's own term
    for (int p = 1; p < np;) {
        go(c->h_plane_k[p - 1] + 1, c->h_plane_k[p] - 1, p - 1, false);      // trivial planes in between: look-ups only
        int G = 1;
        while (G < gmax && p + G < np && c->h_plane_k[p + G] == c->h_plane_k[p + G - 1] + 1) ++G;
        while (G >= 2 && !go_fused(p, G)) --G;
        if (G < 2) { go(c->h_plane_k[p], c->h_plane_k[p], p - 1, true); G = 1; }
        p += G;
    }
    // the run of planes above the medium: with the one-sum form and enough planes to pay for it, the last running sums are first spread into
    // row pairs (134 MB at 256 elements x 256^2: ~0.06 ms) so that every look-up is one load
    const int top_lo = c->h_plane_k[np - 1] + 1;
    const bool tex = c->march_one && c->d_Utex && nz - top_lo >= 16 && !getenv("OLX_MARCH_NO_TEXELS");
    if (tex) {
        const long long cells = (long long)c->hp.nxg * c->n_el * c->hp.nyg;
        hipLaunchKernelGGL(u_texel_k, dim3((unsigned)((cells + 1 + 255) / 256)), dim3(256), 0, c->stream, reinterpret_cast<const float*>(c->d_U[ub]),
                           reinterpret_cast<float2*>(c->d_Utex), c->hp.nxg, c->n_el, c->hp.nyg);
    }
    go(top_lo, nz - 1, np - 1, false, tex);
    {   // olx_field_variant names what ran: "...; fused writers: 28 planes in 7 launches"
        const size_t cut = c->variant.find("; fused writers");
        if (cut != std::string::npos) c->variant.erase(cut);
        if (fused_launches) c->variant += "; fused writers: " + std::to_string(fused_planes) + " planes in " + std::to_string(fused_launches) + " launches";
    }
}

void olx_launch_hmarch(olx_ctx* c, float* pm) {
    if (c->nf >= 8) launch_hmarch_nf<8>(c, pm);
    else if (c->nf >= 4) launch_hmarch_nf<4>(c, pm);
    else if (c->nf >= 2) launch_hmarch_nf<2>(c, pm);
    else launch_hmarch_nf<1>(c, pm);
}
