// kernels 2a (field_accum_k) and 2b (field_shared_k): exact per-pair accumulate on the VALU
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// kernel 2: pressure-field accumulate, exact per voxel-element pair, fp32.
//
// Work map: the [nx,ny,nz] slab is rows of nz voxels (z fastest).  A lane owns ZPL
// consecutive z voxels of one row, so dx, dy and rho^2 = dx^2 + dy^2 are formed once per
// (lane, element) and shared by its ZPL voxels; 64/ (nz/ZPL) rows per wave.  Lanes of a
// wave write ZPL*4-byte pieces that tile whole rows: for nz = 256, ZPL = 4 a wave stores
// one contiguous 1 KiB row with one dwordx4 store per lane.
// Element data is wave-uniform: read with scalar loads (s_load_dwordx8) from the packed
// table, served by the scalar cache -- no VGPR, LDS or vector-memory traffic in the loop.
// Per pair: v_rsq_f32 (1/d), v_sin_f32 + v_cos_f32 on the phase in REVOLUTIONS
// (t = d + frac(f tau), d in wavelengths), ~5 plain VALU.  Transcendental issue is the bound
// (DESIGN.md section 5); HBM sees only the output stream.
//   FLAT : every element has the same z -> (z_v - z_e)^2 hoisted out of the element loop.
//   CLAMP: apply d >= dmin (needed only if a voxel can come within dmin of an element;
//          decided on the host from the element / slab bounding boxes).
// ------------------------------------------------------------------------------------


//   NEAR : some voxel comes within a quarter wavelength of an element (host: min_dist) -- coordinates as (voxel index, residual), see below; else absolute fp32
//          coordinates (one subtraction per axis: the split form costs 5 % of the launch -- jittered array, single focus, 256^3: 1.074 -> 1.128 ms).
template <int ZPL, bool FLAT, bool CLAMP, bool NEAR>
__global__ __launch_bounds__(FIELD_THREADS) void field_accum_k(
    const float* __restrict__ tab, float* __restrict__ pmag, float* __restrict__ inten,
    float* __restrict__ cplx, const FieldParams P) {
    const int f = blockIdx.y;
    const int cpr = (P.nz + ZPL - 1) / ZPL;  // chunks per row
    const long long lane_id = (long long)blockIdx.x * FIELD_THREADS + threadIdx.x;
    const long long rows = (long long)P.nx * P.ny;
    const long long row = lane_id / cpr;
    if (row >= rows) return;
    const int chunk = (int)(lane_id - row * cpr);
    const int i = (int)(row / P.ny), j = (int)(row - (long long)i * P.ny);
    const int k0 = chunk * ZPL;
    // Coordinates as (voxel index, residual): an element is { nearest voxel index, offset from that voxel } per axis (steer_pack_k, split), so that
    // x_v - x_e = (i - i_e) h - f_e with an EXACT index difference -- formed from absolute fp32 coordinates the difference carried the rounding of a
    // coordinate of ~ 10 wavelengths (1e-6) into distances of a fraction of a wavelength: 1.07e-5 of the volume maximum on a 72 mm wide 0.5 mm grid
    // through the element plane (round 6, tools/probe/kernel2a_near_plane.py; the lattice kernels always worked from index differences).
    // Away from the elements (every distance >= a quarter wavelength) the rounding is <= 1e-6 of a term and the kernel keeps its absolute coordinates.
    const float xi = NEAR ? (float)(i + P.x_begin) : (float)(i + P.x_begin) * P.hx, yj = NEAR ? (float)j : (float)j * P.hy;
    // (the spacings in VECTOR registers: an fma reads ONE scalar operand on this part -- with h and f_e both scalar the compiler copies one per element)
    float hxv = P.hx, hyv = P.hy, hzv = P.hz;
    asm volatile("" : "+v"(hxv), "+v"(hyv), "+v"(hzv));
    float z[ZPL], re[ZPL], im[ZPL];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        z[q] = NEAR ? (float)(k0 + q) : (float)(k0 + q) * P.hz;  // plane index | coordinate
        if (FLAT) { const float dz = NEAR ? fmaf(z[q] - P.flat_kz, P.hz, -P.flat_fz) : z[q] - P.flat_ez; z[q] = dz * dz; }
        re[q] = 0.f; im[q] = 0.f;
    }
    const float* t = tab + (size_t)f * P.n_el * TAB_STRIDE;
#pragma unroll 2
    for (int e = 0; e < P.n_el; ++e) {
        const float ex = t[e * TAB_STRIDE + 0], ey = t[e * TAB_STRIDE + 1];      // voxel indices of the element (integers) ...
        const float ez = t[e * TAB_STRIDE + 2], w = t[e * TAB_STRIDE + 3];
        const float phi = t[e * TAB_STRIDE + 4];
        const float fx = t[e * TAB_STRIDE + 5], fy = t[e * TAB_STRIDE + 6], fz = t[e * TAB_STRIDE + 7];      // ... and its offsets from them [wavelengths]
        const float dx = NEAR ? fmaf(xi - ex, hxv, -fx) : xi - ex, dy = NEAR ? fmaf(yj - ey, hyv, -fy) : yj - ey;
        const float r2 = fmaf(dy, dy, dx * dx);
        // Two voxels per packed fp32 instruction (v_pk_add / v_pk_fma / v_pk_mul: the five plain operations of a pair; the three transcendentals
        // stay scalar) -- same bits; jittered array, single focus, 256^3: 1.154 -> 1.080 ms (same box, alternating: round 6)
        if constexpr (ZPL % 2 == 0) {
#pragma unroll
            for (int q = 0; q < ZPL; q += 2) {
                float2_t d2;
                if (FLAT) d2 = float2_t{r2, r2} + float2_t{z[q], z[q + 1]};
                else {
                    float2_t dz = float2_t{z[q], z[q + 1]} - float2_t{ez, ez};
                    if constexpr (NEAR) dz = __builtin_elementwise_fma(dz, float2_t{hzv, hzv}, float2_t{-fz, -fz});
                    d2 = __builtin_elementwise_fma(dz, dz, float2_t{r2, r2});
                }
                if (CLAMP) { d2.x = fmaxf(d2.x, P.dmin2); d2.y = fmaxf(d2.y, P.dmin2); }
                const float2_t ri = {__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
                const float2_t ph = __builtin_elementwise_fma(d2, ri, float2_t{phi, phi});
                const float2_t s = {__builtin_amdgcn_sinf(ph.x), __builtin_amdgcn_sinf(ph.y)};
                const float2_t c = {__builtin_amdgcn_cosf(ph.x), __builtin_amdgcn_cosf(ph.y)};
                const float2_t a = float2_t{w, w} * ri;
                const float2_t rr = __builtin_elementwise_fma(a, c, float2_t{re[q], re[q + 1]});
                const float2_t ii = __builtin_elementwise_fma(a, s, float2_t{im[q], im[q + 1]});
                re[q] = rr.x; re[q + 1] = rr.y; im[q] = ii.x; im[q + 1] = ii.y;
            }
        } else
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            float d2;
            if (FLAT) {
                d2 = r2 + z[q];
            } else {
                const float dz = NEAR ? fmaf(z[q] - ez, hzv, -fz) : z[q] - ez;
                d2 = fmaf(dz, dz, r2);
            }
            if (CLAMP) d2 = fmaxf(d2, P.dmin2);
            const float ri = __builtin_amdgcn_rsqf(d2);
            const float ph = fmaf(d2, ri, phi);  // d [wavelengths] + phi = phase [revolutions]
            const float s = __builtin_amdgcn_sinf(ph);
            const float c = __builtin_amdgcn_cosf(ph);
            const float a = w * ri;
            re[q] = fmaf(a, c, re[q]);
            im[q] = fmaf(a, s, im[q]);
        }
    }
    // epilogue: fused |p|, intensity, optional complex
    const long long base = (long long)f * P.vox + row * P.nz + k0;
    float pm[ZPL], it[ZPL];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        const float m2 = fmaf(re[q], re[q], im[q] * im[q]);
        pm[q] = __builtin_sqrtf(m2);
        it[q] = m2 * P.inten_scale;
    }
    const bool full = (k0 + ZPL <= P.nz);
    if (ZPL == 4 && full) {      // (dword-aligned 16-byte stores: rows of odd length are fine)
        if (P.flags & 1u) *reinterpret_cast<floatx4u_t*>(pmag + base) = floatx4u_t{pm[0], pm[1], pm[2], pm[3]};
        if (P.flags & 2u) *reinterpret_cast<floatx4u_t*>(inten + base) = floatx4u_t{it[0], it[1], it[2], it[3]};
        if (P.flags & 4u) {
            floatx4u_t* c4 = reinterpret_cast<floatx4u_t*>(cplx + 2 * base);
            c4[0] = floatx4u_t{re[0], im[0], re[1], im[1]};
            c4[1] = floatx4u_t{re[2], im[2], re[3], im[3]};
        }
    } else {
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            if (k0 + q < P.nz) {
                if (P.flags & 1u) pmag[base + q] = pm[q];
                if (P.flags & 2u) inten[base + q] = it[q];
                if (P.flags & 4u) { cplx[2 * (base + q)] = re[q]; cplx[2 * (base + q) + 1] = im[q]; }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// kernel 2a-d: kernel 2a with the optional far-field piston directivity (OLX_FIELD_DIRECTIVITY; definition:
// oracle/field_oracle.py piston_directivity).  Per pair, on top of kernel 2a: the direction cosines along the element's
// local axes (6 fma + 2 mul with 1/d), two sinc factors -- v_sin on the argument in revolutions, one v_rcp for both
// denominators, a series value next to the origin -- i.e. 3 more transcendentals and ~16 more plain instructions.  The second
// table entry { ex, w / (2 lambda) | ey, l / (2 lambda) } arrives through the scalar cache like the first.
// ------------------------------------------------------------------------------------
template <int ZPL, bool CLAMP>
__global__ __launch_bounds__(FIELD_THREADS) void field_accum_dir_k(
    const float* __restrict__ tab, const float* __restrict__ tab2, float* __restrict__ pmag, float* __restrict__ inten,
    float* __restrict__ cplx, const FieldParams P) {
    const int f = blockIdx.y;
    const int cpr = (P.nz + ZPL - 1) / ZPL;  // chunks per row
    const long long lane_id = (long long)blockIdx.x * FIELD_THREADS + threadIdx.x;
    const long long rows = (long long)P.nx * P.ny;
    const long long row = lane_id / cpr;
    if (row >= rows) return;
    const int chunk = (int)(lane_id - row * cpr);
    const int i = (int)(row / P.ny), j = (int)(row - (long long)i * P.ny);
    const int k0 = chunk * ZPL;
    const float xi = (float)(i + P.x_begin), yj = (float)j;      // (index, residual) coordinates as in field_accum_k
    float z[ZPL], re[ZPL], im[ZPL];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) { z[q] = (float)(k0 + q); re[q] = 0.f; im[q] = 0.f; }
    const float* t = tab + (size_t)f * P.n_el * TAB_STRIDE;
    constexpr float TWO_PI = 6.283185307179586f;
    for (int e = 0; e < P.n_el; ++e) {
        const float ex = t[e * TAB_STRIDE + 0], ey = t[e * TAB_STRIDE + 1];
        const float ez = t[e * TAB_STRIDE + 2], w = t[e * TAB_STRIDE + 3];
        const float phi = t[e * TAB_STRIDE + 4];
        const float fx = t[e * TAB_STRIDE + 5], fy = t[e * TAB_STRIDE + 6], fz = t[e * TAB_STRIDE + 7];
        const float* a = tab2 ? tab2 + (size_t)e * 8 : t;      // (no frames: never read)
        const float dx = fmaf(xi - ex, P.hx, -fx), dy = fmaf(yj - ey, P.hy, -fy);
        const float r2 = fmaf(dy, dy, dx * dx);
        const float px = tab2 ? fmaf(dy, a[1], dx * a[0]) : 0.f, py = tab2 ? fmaf(dy, a[5], dx * a[4]) : 0.f;   // lateral part of r . ex, r . ey
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            const float dz = fmaf(z[q] - ez, P.hz, -fz);
            float d2 = fmaf(dz, dz, r2);
            if (CLAMP) d2 = fmaxf(d2, P.dmin2);
            const float ri = __builtin_amdgcn_rsqf(d2);
            const float ph = fmaf(d2, ri, phi);  // d [wavelengths] + phi = phase [revolutions]
            // sinc(pi w u_x / lambda) = sin(2 pi tx) / (2 pi tx), tx = u_x w / (2 lambda) [revolutions]
            float D = 1.0f;
            if (tab2) {                                      // (uniform: a launch has the element frames or it has not)
                const float tx = fmaf(dz, a[2], px) * ri * a[3], ty = fmaf(dz, a[6], py) * ri * a[7];
                const float sx = __builtin_amdgcn_sinf(tx), sy = __builtin_amdgcn_sinf(ty);
                const float ax = TWO_PI * tx, ay = TWO_PI * ty;
                const float inv = __builtin_amdgcn_rcpf(ax * ay);
                const bool nx0 = fabsf(ax) < 1e-3f, ny0 = fabsf(ay) < 1e-3f;       // next to the axis: sinc = 1 - t^2 / 6
                if (!nx0 && !ny0) D = sx * sy * inv;
                else D = (nx0 ? fmaf(ax * ax, -1.0f / 6.0f, 1.0f) : sx / ax) * (ny0 ? fmaf(ay * ay, -1.0f / 6.0f, 1.0f) : sy / ay);
            }
            if (P.absorb_l2 > 0.f) D *= __builtin_amdgcn_exp2f(-P.absorb_l2 * (d2 * ri));       // uniform absorbing medium: exp(-a d)
            const float s = __builtin_amdgcn_sinf(ph);
            const float c = __builtin_amdgcn_cosf(ph);
            const float amp = w * ri * D;
            re[q] = fmaf(amp, c, re[q]);
            im[q] = fmaf(amp, s, im[q]);
        }
    }
    const long long base = (long long)f * P.vox + row * P.nz + k0;
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        if (k0 + q >= P.nz) continue;
        const float m2 = fmaf(re[q], re[q], im[q] * im[q]);
        if (P.flags & 1u) pmag[base + q] = __builtin_sqrtf(m2);
        if (P.flags & 2u) inten[base + q] = m2 * P.inten_scale;
        if (P.flags & 4u) { cplx[2 * (base + q)] = re[q]; cplx[2 * (base + q) + 1] = im[q]; }
    }
}

// ------------------------------------------------------------------------------------
// kernel 2b: shared-geometry accumulate.  The geometry term G(v,e) = exp(j k d)/d does not
// depend on the focus, and for an element set that is mirror-symmetric about the grid's
// centre plane(s) G(v,e) = G(sigma v, sigma e).  One lane therefore evaluates G once
// (the 3 transcendentals) and feeds NOUT = MX*MY*NF complex accumulators
//     P_f(sigma_m v) += W[sigma_m e, f] * G(v, e)
// with 4 plain fma each (weights wave-uniform in SGPRs).  MX/MY = 2 folds the x / y mirror
// (lanes cover only the upper half of that axis and also write the mirrored voxel),
// NF = foci per tile (blockIdx.y = tile).  Exact: no approximation is involved, only
// re-association of which (voxel, element) pair is evaluated where.
// Table entry (tile, e) = { kx, ky, kz, 0 | fx, fy, fz, 0 | (wr_k, wi_k) k < NOUT },  k = f_local*NM + cx + DX*cy: the element as (index of the nearest
// coordinate step, offset from it [wavelengths]) per axis -- round 6, as kernel 2a: x_v - x_e = (k_v - k_e) s - f_e with an exact index difference,
// NM = DX*DY distinct mirror columns (perm[m] passed to the pack kernel lists exactly those).
// Coordinates on a mirrored axis are taken relative to the grid centre and formed as
// (2 i - (n-1)) * h/2 so that x(n-1-i) == -x(i) bit for bit.
// ------------------------------------------------------------------------------------

// DX / DY (1 or 2) = distinct weight columns along a folded axis: when the steering itself is
// mirror-symmetric (W[sigma e] == W[e] bit for bit, e.g. an on-axis focus) the mirrored voxel's
// value is the same sum, so it is accumulated once (D = 1) and stored twice.
template <int ZPL, int MX, int MY, int DX, int DY, int NF, bool FLAT, bool CLAMP>
__global__ __launch_bounds__(FIELD_THREADS) void field_shared_k(
    const float* __restrict__ tab, float* __restrict__ pmag, float* __restrict__ inten,
    float* __restrict__ cplx, const SharedParams P) {
    static_assert(DX <= MX && DY <= MY, "distinct columns cannot exceed the fold");
    constexpr int NM = DX * DY, NOUT = NM * NF, STRIDE = SH_HEAD + 2 * NOUT;
    const int tile = blockIdx.y;
    const unsigned cpr = (unsigned)(P.nz + ZPL - 1) / ZPL;
    const unsigned lane_id = blockIdx.x * FIELD_THREADS + threadIdx.x;
    const int x_lo = (MX == 2) ? P.nx / 2 : 0, y_lo = (MY == 2) ? P.ny / 2 : 0;
    const unsigned hyn = (unsigned)(P.ny - y_lo);
    const unsigned rows = (unsigned)(P.nx - x_lo) * hyn;
    const unsigned row = lane_id / cpr;
    if (row >= rows) return;
    const int chunk = (int)(lane_id - row * cpr);
    const int ii = (int)(row / hyn);
    const int i = ii + x_lo, j = (int)(row - (unsigned)ii * hyn) + y_lo;
    const int k0 = chunk * ZPL;
    // NEAR (= the CLAMP instantiations: the host launches them wherever a voxel comes within a quarter wavelength of an element): coordinates as (index, residual);
    // coordinate steps: half a voxel on a folded axis (x = (2 i - (n - 1)) h / 2 about the grid centre), a voxel otherwise; indices are exact in fp32.
    // Elsewhere absolute fp32 coordinates as before (the split form costs the tilted array's single focus 9 %: one fma per pair for dz).
    constexpr bool NEAR = CLAMP;
    float sxv = (MX == 2) ? 0.5f * P.hx : P.hx, syv = (MY == 2) ? 0.5f * P.hy : P.hy, hzv = P.hz;
    const float x = ((MX == 2) ? (float)(2 * i - (P.nx - 1)) : (float)(i + P.x_begin)) * (NEAR ? 1.0f : sxv);
    const float y = ((MY == 2) ? (float)(2 * j - (P.ny - 1)) : (float)j) * (NEAR ? 1.0f : syv);
    asm volatile("" : "+v"(sxv), "+v"(syv), "+v"(hzv));      // (vector registers: an fma reads one scalar operand)
    float z[ZPL], re[ZPL][NOUT], im[ZPL][NOUT];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        z[q] = NEAR ? (float)(k0 + q) : (float)(k0 + q) * P.hz;
        if (FLAT) { const float dz = NEAR ? fmaf(z[q] - P.flat_kz, P.hz, -P.flat_fz) : z[q] - P.flat_ez; z[q] = dz * dz; }
#pragma unroll
        for (int k = 0; k < NOUT; ++k) { re[q][k] = 0.f; im[q][k] = 0.f; }
    }
    const float* t = tab + (size_t)tile * P.n_el * STRIDE;
    for (int e = 0; e < P.n_el; ++e) {
        const float* te = t + (size_t)e * STRIDE;
        const float dx = NEAR ? fmaf(x - te[0], sxv, -te[4]) : x - te[0], dy = NEAR ? fmaf(y - te[1], syv, -te[5]) : y - te[1];
        const float ez = te[2], fz = te[6];
        const float r2 = fmaf(dy, dy, dx * dx);
        float gr[ZPL], gi[ZPL];
        if constexpr (ZPL % 2 == 0) {      // two voxels per packed fp32 instruction, as kernel 2a (same bits): tilted array, single focus 0.372 -> 0.348 ms
#pragma unroll
            for (int q = 0; q < ZPL; q += 2) {
                float2_t d2;
                if (FLAT) d2 = float2_t{r2, r2} + float2_t{z[q], z[q + 1]};
                else {
                    float2_t dz = float2_t{z[q], z[q + 1]} - float2_t{ez, ez};
                    if constexpr (NEAR) dz = __builtin_elementwise_fma(dz, float2_t{hzv, hzv}, float2_t{-fz, -fz});
                    d2 = __builtin_elementwise_fma(dz, dz, float2_t{r2, r2});
                }
                if (CLAMP) { d2.x = fmaxf(d2.x, P.dmin2); d2.y = fmaxf(d2.y, P.dmin2); }
                const float2_t ri = {__builtin_amdgcn_rsqf(d2.x), __builtin_amdgcn_rsqf(d2.y)};
                const float2_t ph = d2 * ri;
                const float2_t g0 = ri * float2_t{__builtin_amdgcn_cosf(ph.x), __builtin_amdgcn_cosf(ph.y)};
                const float2_t g1 = ri * float2_t{__builtin_amdgcn_sinf(ph.x), __builtin_amdgcn_sinf(ph.y)};
                gr[q] = g0.x; gr[q + 1] = g0.y; gi[q] = g1.x; gi[q + 1] = g1.y;
            }
#pragma unroll
            for (int k = 0; k < NOUT; ++k) {
                const float wr = te[SH_HEAD + 2 * k], wi = te[SH_HEAD + 1 + 2 * k];
#pragma unroll
                for (int q = 0; q < ZPL; q += 2) {
                    const float2_t g0 = {gr[q], gr[q + 1]}, g1 = {gi[q], gi[q + 1]};
                    float2_t rr = {re[q][k], re[q + 1][k]}, ii = {im[q][k], im[q + 1][k]};
                    rr = __builtin_elementwise_fma(g0, float2_t{wr, wr}, rr);
                    rr = __builtin_elementwise_fma(-g1, float2_t{wi, wi}, rr);
                    ii = __builtin_elementwise_fma(g0, float2_t{wi, wi}, ii);
                    ii = __builtin_elementwise_fma(g1, float2_t{wr, wr}, ii);
                    re[q][k] = rr.x; re[q + 1][k] = rr.y; im[q][k] = ii.x; im[q + 1][k] = ii.y;
                }
            }
        } else {
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            float d2;
            if (FLAT) {
                d2 = r2 + z[q];
            } else {
                const float dz = NEAR ? fmaf(z[q] - ez, hzv, -fz) : z[q] - ez;
                d2 = fmaf(dz, dz, r2);
            }
            if (CLAMP) d2 = fmaxf(d2, P.dmin2);
            const float ri = __builtin_amdgcn_rsqf(d2);
            const float ph = d2 * ri;  // distance in wavelengths = phase in revolutions
            gr[q] = ri * __builtin_amdgcn_cosf(ph);
            gi[q] = ri * __builtin_amdgcn_sinf(ph);
        }
#pragma unroll
        for (int k = 0; k < NOUT; ++k) {
            const float wr = te[SH_HEAD + 2 * k], wi = te[SH_HEAD + 1 + 2 * k];
#pragma unroll
            for (int q = 0; q < ZPL; ++q) {
                re[q][k] = fmaf(gr[q], wr, re[q][k]);
                re[q][k] = fmaf(-gi[q], wi, re[q][k]);
                im[q][k] = fmaf(gr[q], wi, im[q][k]);
                im[q][k] = fmaf(gi[q], wr, im[q][k]);
            }
        }
        }
    }
    const bool full = k0 + ZPL <= P.nz;      // (dword-aligned 16-byte stores: rows of odd length are fine)
#pragma unroll
    for (int kk = 0; kk < MX * MY * NF; ++kk) {      // every stored volume slice: (focus, mirror image)
        const int f = tile * NF + kk / (MX * MY);
        if (f >= P.n_foci) continue;
        const int ms = kk % (MX * MY);                // store mirror: bit 0 = x (if MX == 2), next = y
        const bool fx = (MX == 2) && (ms & 1), fy = (MY == 2) && ((MX == 2) ? (ms >> 1) : (ms & 1));
        // weight column that holds this image's sum (collapsed along axes with symmetric steering)
        const int cx = (DX == 2 && fx) ? 1 : 0, cy = (DY == 2 && fy) ? 1 : 0;
        const int k = (kk / (MX * MY)) * NM + cx + DX * cy;
        const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
        const long long base = (long long)f * P.vox + ((long long)io * P.ny + jo) * P.nz + k0;
        float pm[ZPL], it[ZPL];
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            const float m2 = fmaf(re[q][k], re[q][k], im[q][k] * im[q][k]);
            pm[q] = __builtin_sqrtf(m2);
            it[q] = m2 * P.inten_scale;
        }
        if (ZPL == 4 && full) {
            if (P.flags & 1u) *reinterpret_cast<floatx4u_t*>(pmag + base) = floatx4u_t{pm[0], pm[1], pm[2], pm[3]};
            if (P.flags & 2u) *reinterpret_cast<floatx4u_t*>(inten + base) = floatx4u_t{it[0], it[1], it[2], it[3]};
            if (P.flags & 4u) {
                floatx4u_t* c4 = reinterpret_cast<floatx4u_t*>(cplx + 2 * base);
                c4[0] = floatx4u_t{re[0][k], im[0][k], re[1][k], im[1][k]};
                c4[1] = floatx4u_t{re[2][k], im[2][k], re[3][k], im[3][k]};
            }
        } else {
#pragma unroll
            for (int q = 0; q < ZPL; ++q) {
                if (k0 + q < P.nz) {
                    if (P.flags & 1u) pmag[base + q] = pm[q];
                    if (P.flags & 2u) inten[base + q] = it[q];
                    if (P.flags & 4u) { cplx[2 * (base + q)] = re[q][k]; cplx[2 * (base + q) + 1] = im[q][k]; }
                }
            }
        }
    }
}


}  // namespace olx

using namespace olx;

template <int MX, int MY, int DX, int DY, int NF>
static void launch_shared(olx_ctx* c, float* pm) {
    const SharedParams& S = c->sp;
    constexpr int ZPL = 4;
    const long long cpr = (S.nz + ZPL - 1) / ZPL;
    const long long lanes = (long long)(S.nx - (MX == 2 ? S.nx / 2 : 0)) * (S.ny - (MY == 2 ? S.ny / 2 : 0)) * cpr;
    dim3 grid((unsigned)((lanes + FIELD_THREADS - 1) / FIELD_THREADS), (c->plan_foci + NF - 1) / NF);
    dim3 blk(FIELD_THREADS);
    if (c->flat) {
        if (c->near) hipLaunchKernelGGL((field_shared_k<ZPL, MX, MY, DX, DY, NF, true, true>), grid, blk, 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, S);
        else          hipLaunchKernelGGL((field_shared_k<ZPL, MX, MY, DX, DY, NF, true, false>), grid, blk, 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, S);
    } else {
        if (c->near) hipLaunchKernelGGL((field_shared_k<ZPL, MX, MY, DX, DY, NF, false, true>), grid, blk, 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, S);
        else          hipLaunchKernelGGL((field_shared_k<ZPL, MX, MY, DX, DY, NF, false, false>), grid, blk, 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, S);
    }
}

static bool dispatch_shared(olx_ctx* c, float* pm) {
#define OLX_CASE(MX_, MY_, DX_, DY_, NF_) \
    if (c->mx == MX_ && c->my == MY_ && c->dx == DX_ && c->dy == DY_ && c->nf == NF_) { launch_shared<MX_, MY_, DX_, DY_, NF_>(c, pm); return true; }
    // no fold: foci tiles only
    OLX_CASE(1, 1, 1, 1, 2) OLX_CASE(1, 1, 1, 1, 4) OLX_CASE(1, 1, 1, 1, 8)
    // one fold
    OLX_CASE(2, 1, 1, 1, 1) OLX_CASE(2, 1, 1, 1, 2) OLX_CASE(2, 1, 1, 1, 4) OLX_CASE(2, 1, 1, 1, 8)
    OLX_CASE(2, 1, 2, 1, 1) OLX_CASE(2, 1, 2, 1, 2) OLX_CASE(2, 1, 2, 1, 4)
    OLX_CASE(1, 2, 1, 1, 1) OLX_CASE(1, 2, 1, 1, 2) OLX_CASE(1, 2, 1, 1, 4) OLX_CASE(1, 2, 1, 1, 8)
    OLX_CASE(1, 2, 1, 2, 1) OLX_CASE(1, 2, 1, 2, 2) OLX_CASE(1, 2, 1, 2, 4)
    // two folds
    OLX_CASE(2, 2, 1, 1, 1) OLX_CASE(2, 2, 1, 1, 2) OLX_CASE(2, 2, 1, 1, 4) OLX_CASE(2, 2, 1, 1, 8)
    OLX_CASE(2, 2, 2, 1, 1) OLX_CASE(2, 2, 2, 1, 2) OLX_CASE(2, 2, 2, 1, 4)
    OLX_CASE(2, 2, 1, 2, 1) OLX_CASE(2, 2, 1, 2, 2) OLX_CASE(2, 2, 1, 2, 4)
    OLX_CASE(2, 2, 2, 2, 1) OLX_CASE(2, 2, 2, 2, 2)
#undef OLX_CASE
    return false;
}

template <bool FLAT, bool CLAMP>
static void launch_field(olx_ctx* c, float* pm) {
    const FieldParams& P = c->fp;
    constexpr int ZPL = 4;
    const long long cpr = (P.nz + ZPL - 1) / ZPL;
    const long long lanes = (long long)P.nx * P.ny * cpr;
    dim3 grid((unsigned)((lanes + FIELD_THREADS - 1) / FIELD_THREADS), c->plan_foci);
    // (the table carries split coordinates where a voxel comes within a quarter wavelength of an element: olx.hip, steer_pack_k)
    if (CLAMP || c->near) hipLaunchKernelGGL((field_accum_k<ZPL, FLAT, CLAMP, true>), grid, dim3(FIELD_THREADS), 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, P);
    else hipLaunchKernelGGL((field_accum_k<ZPL, FLAT, CLAMP, CLAMP>), grid, dim3(FIELD_THREADS), 0, c->stream, c->d_tab, pm, c->d_inten, c->d_cplx, P);
}

bool olx_launch_shared(olx_ctx* c, float* pm) { return dispatch_shared(c, pm); }

void olx_launch_accum_dir(olx_ctx* c, float* pm) {
    const FieldParams& P = c->fp;
    const long long lanes = (long long)P.nx * P.ny * ((P.nz + 3) / 4);
    dim3 grid((unsigned)((lanes + FIELD_THREADS - 1) / FIELD_THREADS), c->plan_foci), blk(FIELD_THREADS);
    const float* frames = c->directivity ? c->d_tab2 : nullptr;      // (absorption only: no piston factor)
    if (c->clamp) hipLaunchKernelGGL((field_accum_dir_k<4, true>), grid, blk, 0, c->stream, c->d_tab, frames, pm, c->d_inten, c->d_cplx, P);
    else hipLaunchKernelGGL((field_accum_dir_k<4, false>), grid, blk, 0, c->stream, c->d_tab, frames, pm, c->d_inten, c->d_cplx, P);
}

void olx_launch_accum(olx_ctx* c, float* pm) {
    if (c->flat) { if (c->clamp) launch_field<true, true>(c, pm); else launch_field<true, false>(c, pm); }
    else         { if (c->clamp) launch_field<false, true>(c, pm); else launch_field<false, false>(c, pm); }
}
