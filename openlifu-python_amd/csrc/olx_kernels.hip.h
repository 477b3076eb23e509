// Device kernels of the openlifu hot path for gfx950 (CDNA4, wave64).
// Written for MI355X only: no other-architecture paths.
//
//  kernel 1  bf_solve_k        per-element geometric delay / apodization solve (fp64)
//            steer_pack_k      fp64 steering + element table -> fp32 kernel-2 table
//  kernel 2  field_accum_k     per-voxel complex pressure accumulate over elements (fp32)
//            field_aggregate_k max / mean over foci;  field_scale_k  per-focus scaling
//
// Data layout in HBM is documented in DESIGN.md section 4.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace olx {

// ------------------------------------------------------------------------------------
// kernel 1: F blocks (one per focus) x 256 threads striding over elements.
// Element table is SoA fp64 (pos[3][N], nrm[3][N]) so that lane e reads pos[a][e]:
// consecutive lanes -> consecutive 8-byte words (coalesced).  The focus and the 4x4
// transform are staged once per block in LDS and broadcast from there.
// Restates  xdc/element.py:239-246 (distance), :248-260 (angle),
//           bf/delay_methods/direct.py:36-38, bf/apod_methods/maxangle.py:37-38,
//           bf/apod_methods/piecewiselinear.py:46-48.
// ------------------------------------------------------------------------------------
constexpr int BF_THREADS = 256;

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off, 64));
    return v;
}

__global__ __launch_bounds__(BF_THREADS) void bf_solve_k(
    const double* __restrict__ pos,  // [3][N]
    const double* __restrict__ nrm,  // [3][N]
    int n, const double* __restrict__ foci /*[F][3]*/, const double* __restrict__ M /*[16]*/,
    double c, int apod_kind, double angle_scale, double p0, double p1,
    double* __restrict__ delays /*[F][N]*/, double* __restrict__ apod /*[F][N]*/) {
    __shared__ double s_focus[3];
    __shared__ double s_M[16];
    __shared__ double s_red[BF_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    if (tid < 3) s_focus[tid] = foci[3 * f + tid];
    if (tid >= 64 && tid < 80) s_M[tid - 64] = M[tid - 64];
    __syncthreads();
    const double fx = s_focus[0], fy = s_focus[1], fz = s_focus[2];
    double* dl = delays + (size_t)f * n;
    double* ap = apod + (size_t)f * n;
    double lmax = -1.0;
    for (int e = tid; e < n; e += BF_THREADS) {
        const double px = pos[e], py = pos[n + e], pz = pos[2 * n + e];
        // gpos = (M . [p,1])[:3]
        const double gx = s_M[0] * px + s_M[1] * py + s_M[2] * pz + s_M[3];
        const double gy = s_M[4] * px + s_M[5] * py + s_M[6] * pz + s_M[7];
        const double gz = s_M[8] * px + s_M[9] * py + s_M[10] * pz + s_M[11];
        const double vx = fx - gx, vy = fy - gy, vz = fz - gz;
        const double d = sqrt(vx * vx + vy * vy + vz * vz);
        const double tof = d / c;
        dl[e] = tof;
        lmax = fmax(lmax, tof);
        double a;
        if (apod_kind == 0) {
            a = p0;
        } else {
            const double nx0 = nrm[e], ny0 = nrm[n + e], nz0 = nrm[2 * n + e];
            // v2 = (M . pose)[:3,2] = M[:3,:3] . normal
            double wx = s_M[0] * nx0 + s_M[1] * ny0 + s_M[2] * nz0;
            double wy = s_M[4] * nx0 + s_M[5] * ny0 + s_M[6] * nz0;
            double wz = s_M[8] * nx0 + s_M[9] * ny0 + s_M[10] * nz0;
            const double wn = sqrt(wx * wx + wy * wy + wz * wz);
            wx /= wn; wy /= wn; wz /= wn;
            const double ux = vx / d, uy = vy / d, uz = vz / d;
            const double cx = uy * wz - uz * wy, cy = uz * wx - ux * wz, cz = ux * wy - uy * wx;
            double sn = sqrt(cx * cx + cy * cy + cz * cz);
            const double theta_deg = asin(sn) * angle_scale;  // 180/pi (np.degrees) or 1
            if (apod_kind == 1) {
                a = (theta_deg <= p0) ? 1.0 : 0.0;
            } else {
                const double fr = (p0 - theta_deg) / (p0 - p1);
                a = fmax(0.0, fmin(1.0, fr));
            }
        }
        ap[e] = a;
    }
    lmax = wave_max(lmax);
    if ((tid & 63) == 0) s_red[tid >> 6] = lmax;
    __syncthreads();
    double bmax = s_red[0];
#pragma unroll
    for (int w = 1; w < BF_THREADS / 64; ++w) bmax = fmax(bmax, s_red[w]);
    for (int e = tid; e < n; e += BF_THREADS) dl[e] = bmax - dl[e];  // same thread wrote dl[e]
}

// ------------------------------------------------------------------------------------
// hardware hand-off of the steering table (io/LIFUTXDevice.py:1357-1372, 1874): per focus and
// element the beamformer-clock delay count int(delay * 1.0 * bf_clk) -- the reference's own fp64 expression,
// truncated toward zero, so the ticks are bit-exact -- and the apodization-off bit int(1 - apod)
// (LIFUTXDevice.py:1811); per focus max(apod) (the duty-cycle factor of :1358) and the number of delays that
// do not fit `width` bits (set_register_value would raise, :1500-1501).  F blocks x 256 threads.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(BF_THREADS) void bf_quantize_k(const double* __restrict__ delays, const double* __restrict__ apod,
                                                             int n, double bf_clk, unsigned max_ticks,
                                                             unsigned short* __restrict__ ticks, unsigned char* __restrict__ apod_off,
                                                             double* __restrict__ max_apod, int* __restrict__ n_overflow) {
    __shared__ double s_red[BF_THREADS / 64];
    __shared__ int s_ovf[BF_THREADS / 64];
    const int f = blockIdx.x, tid = threadIdx.x;
    double amax = -1.0e300;
    int ovf = 0;
    for (int e = tid; e < n; e += BF_THREADS) {
        const size_t o = (size_t)f * n + e;
        const double prod = delays[o] * 1.0 * bf_clk;
        const long long t = (long long)prod;                 // int(): toward zero
        if (t < 0 || t > (long long)max_ticks) ++ovf;
        ticks[o] = (unsigned short)(t < 0 ? 0 : (t > 65535 ? 65535 : t));
        const double a = apod[o];
        apod_off[o] = (unsigned char)(int)(1.0 - a);
        amax = fmax(amax, a);
    }
    amax = wave_max(amax);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) ovf += __shfl_xor(ovf, off, 64);
    if ((tid & 63) == 0) { s_red[tid >> 6] = amax; s_ovf[tid >> 6] = ovf; }
    __syncthreads();
    if (tid == 0) {
        double m = s_red[0]; int v = s_ovf[0];
#pragma unroll
        for (int w = 1; w < BF_THREADS / 64; ++w) { m = fmax(m, s_red[w]); v += s_ovf[w]; }
        max_apod[f] = m;
        n_overflow[f] = v;
    }
}

// ------------------------------------------------------------------------------------
// steering pack: fp64 (pos, area, delays, apod) -> the fp32 table kernel 2 streams through
// the scalar cache.  Entry (f, e) = 8 floats (32 B, one s_load_dwordx8):
//   { (x_e - ox)/lambda, (y_e - oy)/lambda, (z_e - oz)/lambda, w_ef, phi_ef, 0, 0, 0 }
// Lengths are in WAVELENGTHS (x f0/c) so that the phase in revolutions is the distance itself:
// t = d2 * rsq(d2) + phi is ONE fma.  w_ef = a_ef P0 S_e / lambda^2 [Pa] (amplitude w/d with d in
// wavelengths);  phi_ef = frac(f0 tau_ef) [revolutions].
// Differences and products are formed in fp64 and rounded once.
// ------------------------------------------------------------------------------------
constexpr int TAB_STRIDE = 8;

__global__ void steer_pack_k(const double* __restrict__ pos, const double* __restrict__ area, int n,
                             const double* __restrict__ delays, const double* __restrict__ apod,
                             double ox, double oy, double oz, double freq, double p0_over_lambda,
                             double rev, const int* __restrict__ kfirst, const int* __restrict__ klast,
                             float* __restrict__ tab) {
    const int f = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const size_t o = ((size_t)f * n + e);
    const double cyc = freq * delays[o];
    float* t = tab + o * TAB_STRIDE;
    t[0] = (float)((pos[e] - ox) * rev);
    t[1] = (float)((pos[n + e] - oy) * rev);
    t[2] = (float)((pos[2 * n + e] - oz) * rev);
    t[3] = (float)(apod[o] * area[e] * p0_over_lambda * rev);
    t[4] = (float)(cyc - floor(cyc));
    t[5] = kfirst ? __int_as_float(kfirst[e]) : 0.f;  // kernel 2h: planes strictly above / below the element
    t[6] = klast ? __int_as_float(klast[e]) : 0.f;
    t[7] = 0.f;
}

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
struct FieldParams {
    int nx, ny, nz;        // slab extent in voxels (nx = slab x_count)
    int n_el;
    int x_begin;           // slab start (global x index): coordinates are formed from GLOBAL indices so
                           // that a slab launch is bit-identical to the same voxels of a whole-grid launch
    float hx, hy, hz;      // spacing [wavelengths]
    float dmin2;           // dmin^2 [wavelengths^2]
    float inten_scale;     // 1e-4 / (2 rho c)
    float flat_ez;         // common element z (FLAT only), relative to table origin [wavelengths]
    long long vox;         // voxels per focus volume (nx*ny*nz)
    unsigned flags;        // OLX_OUT_*
};

constexpr int FIELD_THREADS = 256;

template <int ZPL, bool FLAT, bool CLAMP>
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
    const float x = (float)(i + P.x_begin) * P.hx;
    const float y = (float)j * P.hy;
    float z[ZPL], re[ZPL], im[ZPL];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        z[q] = (float)(k0 + q) * P.hz;
        if (FLAT) { const float dz = z[q] - P.flat_ez; z[q] = dz * dz; }
        re[q] = 0.f; im[q] = 0.f;
    }
    const float* t = tab + (size_t)f * P.n_el * TAB_STRIDE;
#pragma unroll 2
    for (int e = 0; e < P.n_el; ++e) {
        const float ex = t[e * TAB_STRIDE + 0], ey = t[e * TAB_STRIDE + 1];
        const float ez = t[e * TAB_STRIDE + 2], w = t[e * TAB_STRIDE + 3];
        const float phi = t[e * TAB_STRIDE + 4];
        const float dx = x - ex, dy = y - ey;
        const float r2 = fmaf(dy, dy, dx * dx);
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            float d2;
            if (FLAT) {
                d2 = r2 + z[q];
            } else {
                const float dz = z[q] - ez;
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
    if (ZPL == 4 && full && (P.nz & 3) == 0) {
        if (P.flags & 1u) *reinterpret_cast<float4*>(pmag + base) = make_float4(pm[0], pm[1], pm[2], pm[3]);
        if (P.flags & 2u) *reinterpret_cast<float4*>(inten + base) = make_float4(it[0], it[1], it[2], it[3]);
        if (P.flags & 4u) {
            float4* c4 = reinterpret_cast<float4*>(cplx + 2 * base);
            c4[0] = make_float4(re[0], im[0], re[1], im[1]);
            c4[1] = make_float4(re[2], im[2], re[3], im[3]);
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
// kernel 2b: shared-geometry accumulate.  The geometry term G(v,e) = exp(j k d)/d does not
// depend on the focus, and for an element set that is mirror-symmetric about the grid's
// centre plane(s) G(v,e) = G(sigma v, sigma e).  One lane therefore evaluates G once
// (the 3 transcendentals) and feeds NOUT = MX*MY*NF complex accumulators
//     P_f(sigma_m v) += W[sigma_m e, f] * G(v, e)
// with 4 plain fma each (weights wave-uniform in SGPRs).  MX/MY = 2 folds the x / y mirror
// (lanes cover only the upper half of that axis and also write the mirrored voxel),
// NF = foci per tile (blockIdx.y = tile).  Exact: no approximation is involved, only
// re-association of which (voxel, element) pair is evaluated where.
// Table entry (tile, e) = { x_e, y_e, z_e, 0, (wr_k, wi_k) k < NOUT },  k = f_local*NM + cx + DX*cy,
// NM = DX*DY distinct mirror columns (perm[m] passed to the pack kernel lists exactly those).
// Coordinates on a mirrored axis are taken relative to the grid centre and formed as
// (2 i - (n-1)) * h/2 so that x(n-1-i) == -x(i) bit for bit.
// ------------------------------------------------------------------------------------
struct SharedParams {
    int nx, ny, nz, n_el;
    int x_begin, n_foci;
    float hx, hy, hz;                     // [wavelengths]
    float dmin2, inten_scale, flat_ez;
    long long vox;
    unsigned flags;
};

// DX / DY (1 or 2) = distinct weight columns along a folded axis: when the steering itself is
// mirror-symmetric (W[sigma e] == W[e] bit for bit, e.g. an on-axis focus) the mirrored voxel's
// value is the same sum, so it is accumulated once (D = 1) and stored twice.
template <int ZPL, int MX, int MY, int DX, int DY, int NF, bool FLAT, bool CLAMP>
__global__ __launch_bounds__(FIELD_THREADS) void field_shared_k(
    const float* __restrict__ tab, float* __restrict__ pmag, float* __restrict__ inten,
    float* __restrict__ cplx, const SharedParams P) {
    static_assert(DX <= MX && DY <= MY, "distinct columns cannot exceed the fold");
    constexpr int NM = DX * DY, NOUT = NM * NF, STRIDE = 4 + 2 * NOUT;
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
    const float x = (MX == 2) ? (float)(2 * i - (P.nx - 1)) * (0.5f * P.hx) : (float)(i + P.x_begin) * P.hx;
    const float y = (MY == 2) ? (float)(2 * j - (P.ny - 1)) * (0.5f * P.hy) : (float)j * P.hy;
    float z[ZPL], re[ZPL][NOUT], im[ZPL][NOUT];
#pragma unroll
    for (int q = 0; q < ZPL; ++q) {
        z[q] = (float)(k0 + q) * P.hz;
        if (FLAT) { const float dz = z[q] - P.flat_ez; z[q] = dz * dz; }
#pragma unroll
        for (int k = 0; k < NOUT; ++k) { re[q][k] = 0.f; im[q][k] = 0.f; }
    }
    const float* t = tab + (size_t)tile * P.n_el * STRIDE;
#ifdef OLX_EXP_UNROLL
#pragma unroll OLX_EXP_UNROLL
#endif
    for (int e = 0; e < P.n_el; ++e) {
        const float* te = t + (size_t)e * STRIDE;
        const float dx = x - te[0], dy = y - te[1];
        const float ez = te[2];
        const float r2 = fmaf(dy, dy, dx * dx);
        float gr[ZPL], gi[ZPL];
#pragma unroll
        for (int q = 0; q < ZPL; ++q) {
            float d2;
            if (FLAT) {
                d2 = r2 + z[q];
            } else {
                const float dz = z[q] - ez;
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
            const float wr = te[4 + 2 * k], wi = te[5 + 2 * k];
#pragma unroll
            for (int q = 0; q < ZPL; ++q) {
                re[q][k] = fmaf(gr[q], wr, re[q][k]);
                re[q][k] = fmaf(-gi[q], wi, re[q][k]);
                im[q][k] = fmaf(gr[q], wi, im[q][k]);
                im[q][k] = fmaf(gi[q], wr, im[q][k]);
            }
        }
    }
    const bool full = (k0 + ZPL <= P.nz) && (P.nz % ZPL == 0);
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
            if (P.flags & 1u) *reinterpret_cast<float4*>(pmag + base) = make_float4(pm[0], pm[1], pm[2], pm[3]);
            if (P.flags & 2u) *reinterpret_cast<float4*>(inten + base) = make_float4(it[0], it[1], it[2], it[3]);
            if (P.flags & 4u) {
                float4* c4 = reinterpret_cast<float4*>(cplx + 2 * base);
                c4[0] = make_float4(re[0][k], im[0][k], re[1][k], im[1][k]);
                c4[1] = make_float4(re[2][k], im[2][k], re[3][k], im[3][k]);
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

// pack for kernel 2b: complex weights W[sigma_m(e), f] = a P0 S / lambda * exp(j 2 pi frac(f0 tau)),
// evaluated in fp64 and rounded once.  perm[m][e] = index of the mirror image of element e.
__global__ void steer_pack_shared_k(const double* __restrict__ pos, const double* __restrict__ area, int n,
                                    const double* __restrict__ delays, const double* __restrict__ apod,
                                    const int* __restrict__ perm, double ox, double oy, double oz, double freq,
                                    double p0_over_lambda, double rev, int n_foci, int nf, int nm,
                                    float* __restrict__ tab) {
    const int tile = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int nout = nf * nm, stride = 4 + 2 * nout;
    float* t = tab + ((size_t)tile * n + e) * stride;
    t[0] = (float)((pos[e] - ox) * rev);
    t[1] = (float)((pos[n + e] - oy) * rev);
    t[2] = (float)((pos[2 * n + e] - oz) * rev);
    t[3] = 0.f;
    for (int k = 0; k < nout; ++k) {
        const int f = tile * nf + k / nm, m = k % nm;
        float wr = 0.f, wi = 0.f;
        if (f < n_foci) {
            const int es = perm[m * n + e];
            const size_t o = (size_t)f * n + es;
            const double cyc = freq * delays[o];
            const double ph = 6.283185307179586476925286766559 * (cyc - floor(cyc));
            const double w = apod[o] * area[es] * p0_over_lambda * rev;
            wr = (float)(w * cos(ph));
            wi = (float)(w * sin(ph));
        }
        t[4 + 2 * k] = wr;
        t[5 + 2 * k] = wi;
    }
}

// ------------------------------------------------------------------------------------
// kernel 2c: shared-geometry accumulate with the contraction on the matrix cores.
//
//   out[v, c] = sum_k A[v, k] B[k, c]      k = (element e, part in {re, im}),  c = (output o, part)
//   A[v,(e,re)] = Re G(v,e), A[v,(e,im)] = Im G(v,e),   G = exp(j k d)/d  (focus independent)
//   B[(e,re)][(o,re)] = wr, B[(e,im)][(o,re)] = -wi, B[(e,re)][(o,im)] = wi, B[(e,im)][(o,im)] = wr
// "output o" = one distinct steering vector (a focus seen through one mirror image; images and foci whose
// vectors coincide share it and become its store targets); up to 8 of them fill the 16 columns of
// v_mfma_f32_16x16x32_f16.  The VALU produces G (the transcendentals) directly in the MFMA A-operand
// layout -- lane l owns voxel row l&15 and the four elements 4*(l>>4)..+3 of the 16-element K-step,
// i.e. exactly its eight k values -- so no LDS transpose is needed; the matrix pipe runs concurrently
// with the VALU.  fp32 accuracy from fp16 matrix math: both operands are split hi + lo
// (x*S = hi + lo, |lo| <= 2^-11 |hi|) and three products are accumulated in fp32,
//   A B ~= Ah Bh + Al Bh + Ah Bl        (dropped term Al Bl ~ 2^-22 relative);
// power-of-two scales S_G, S_W keep hi and lo in fp16's normal range and are undone in the epilogue.
// A wave owns MT tiles of 16 consecutive z voxels of one grid row; B fragments (pre-packed in
// lane order by mfma_pack_k) and element coordinates are staged per 256-element chunk in LDS.
// ------------------------------------------------------------------------------------
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float floatx4_t __attribute__((ext_vector_type(4)));

constexpr int MFMA_COLS = 8;        // complex output columns per 16-column MFMA tile
constexpr int MFMA_MAX_NT = 4;      // column tiles per launch tile (A fragments are reused across them)
constexpr int MFMA_ELEMS_LDS = 256; // elements * NT staged in LDS at a time (32 KiB of B fragments)

struct MfmaParams {
    int nx, ny, nz, n_el_pad;      // n_el_pad: elements padded to a multiple of 16 (zero weights)
    int x_begin, n_tiles;
    float hx, hy, hz;              // [wavelengths]
    float dmin2, flat_ez, g_scale; // g_scale = S_G
    float out_scale;               // 1 / (S_G S_W)
    float inten_scale;
    long long vox;
    unsigned flags;
};

union Half8Bits { half8_t h; uint4 u; unsigned w[4]; };

template <int MT, int NT, int MX, int MY, bool FLAT, bool CLAMP>
__global__ __launch_bounds__(FIELD_THREADS) void field_mfma_k(
    const float4* __restrict__ coords /*[n_el_pad]*/, const uint4* __restrict__ bfrag /*[tiles][ks][NT][2][64]*/,
    float* __restrict__ pmag, float* __restrict__ inten, float* __restrict__ cplx,
    const int* __restrict__ targets /*[tiles][32][4]: focus*4 + image of every store target of a column, -1 = none*/,
    const MfmaParams P) {
    constexpr int CH = MFMA_ELEMS_LDS / NT;              // elements per LDS chunk
    __shared__ float4 s_xyz[CH];
    __shared__ uint4 s_B[CH / 16][NT][2][64];
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, r16 = lane & 15;
    constexpr int RUN = MT * 16;                         // z voxels per wave
    const unsigned rpr = (unsigned)(P.nz + RUN - 1) / RUN;  // runs per row
    const int x_lo = (MX == 2) ? P.nx / 2 : 0, y_lo = (MY == 2) ? P.ny / 2 : 0;
    const unsigned hyn = (unsigned)(P.ny - y_lo);
    const unsigned rows = (unsigned)(P.nx - x_lo) * hyn;
    const unsigned run = blockIdx.x * (FIELD_THREADS / 64) + wave;
    const unsigned row = run / rpr;
    const bool active = row < rows;                      // inactive waves still help staging LDS
    const unsigned rowc = active ? row : 0;
    const int zb = (int)(run - row * rpr) * RUN;
    const int ii = (int)(rowc / hyn);
    const int i = ii + x_lo, j = (int)(rowc - (unsigned)ii * hyn) + y_lo;
    const float x = (MX == 2) ? (float)(2 * i - (P.nx - 1)) * (0.5f * P.hx) : (float)(i + P.x_begin) * P.hx;
    const float y = (MY == 2) ? (float)(2 * j - (P.ny - 1)) * (0.5f * P.hy) : (float)j * P.hy;
    float zz[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        zz[t] = (float)(zb + 16 * t + r16) * P.hz;
        if (FLAT) { const float dz = zz[t] - P.flat_ez; zz[t] = dz * dz; }
    }
    floatx4_t acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};

    const int ks_total = P.n_el_pad / 16;
    for (int chunk = 0; chunk < P.n_el_pad; chunk += CH) {
        const int n_here = min(CH, P.n_el_pad - chunk), nks = n_here / 16;
        __syncthreads();
        if (tid < n_here) s_xyz[tid] = coords[chunk + tid];
        const uint4* src = bfrag + ((size_t)tile * ks_total + chunk / 16) * (NT * 128);
        for (int q = tid; q < nks * NT * 128; q += FIELD_THREADS) (&s_B[0][0][0][0])[q] = src[q];
        __syncthreads();
        if (!active) continue;
        for (int ks = 0; ks < nks; ++ks) {
            float r2[4], ez[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 e = s_xyz[16 * ks + 4 * g + q];
                const float dx = x - e.x, dy = y - e.y;
                r2[q] = fmaf(dy, dy, dx * dx);
                ez[q] = e.z;
            }
            // B fragments of this K-step stay in registers and are reused by all MT voxel tiles.
            Half8Bits bh[NT], bl[NT];
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                bh[nt].u = s_B[ks][nt][0][lane];
                bl[nt].u = s_B[ks][nt][1][lane];
            }
            // Per voxel tile: the VALU builds the A fragment (hi, lo), then its 3*NT MFMAs are issued.
            // (Interleaving the MFMAs of tile t-1 into tile t's VALU stream with sched_group_barrier was
            // measured: no gain -- on gfx950 the 16x16x32 MFMAs and this VALU mix add up)
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                Half8Bits ah, al;
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float d2;
                    if (FLAT) {
                        d2 = r2[q] + zz[t];
                    } else {
                        const float dz = zz[t] - ez[q];
                        d2 = fmaf(dz, dz, r2[q]);
                    }
                    if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                    const float ri = __builtin_amdgcn_rsqf(d2);
                    const float ph = d2 * ri;            // distance in wavelengths = phase in revolutions
                    const float rs = ri * P.g_scale;
                    const float gr = rs * __builtin_amdgcn_cosf(ph);
                    const float gi = rs * __builtin_amdgcn_sinf(ph);
                    const auto hi = __builtin_amdgcn_cvt_pkrtz(gr, gi);
                    const auto lo = __builtin_amdgcn_cvt_pkrtz(gr - (float)hi[0], gi - (float)hi[1]);
                    ah.w[q] = __builtin_bit_cast(unsigned, hi);
                    al.w[q] = __builtin_bit_cast(unsigned, lo);
                }
#ifdef OLX_EXP_NOMFMA
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt][0] += (float)bh[nt].h[0] + (float)bl[nt].h[0];
#pragma unroll
                for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(ah.w[q]), "v"(al.w[q]));
#else
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bl[nt].h, acc[t][nt], 0, 0, 0);
#endif
            }
        }
    }
    if (!active) return;
    // epilogue.  D layout: lane holds rows 4*(lane>>4)+r (r = 0..3) of column lane&15 = (o, part):
    // even lanes own Re, odd lanes Im of output o; after one cross-lane add both know |p|^2, the even
    // lane stores |p| and the odd lane the intensity (16-B pieces, 64 B contiguous per column and tile).
    const int part = r16 & 1;
    float* const dst_arr = part == 0 ? pmag : inten;
    const bool want = part == 0 ? (P.flags & 1u) != 0 : (P.flags & 2u) != 0;
    const bool fast = (P.nz % RUN) == 0;  // wave-uniform: every z of the run exists, 16-B aligned
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int4 tg = reinterpret_cast<const int4*>(targets)[(size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt * MFMA_COLS + (r16 >> 1)];
        float w[MT][4], v[MT][4];
#pragma unroll
        for (int t = 0; t < MT; ++t)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                v[t][r] = acc[t][nt][r] * P.out_scale;
                const float sq = v[t][r] * v[t][r];
                const float m2 = sq + __shfl_xor(sq, 1, 64);  // re^2 + im^2 (partner lane holds the other part)
                w[t][r] = part == 0 ? __builtin_sqrtf(m2) : m2 * P.inten_scale;
            }
        const int tgs[4] = {tg.x, tg.y, tg.z, tg.w};
#pragma unroll
        for (int s4 = 0; s4 < 4; ++s4) {
            const int code = tgs[s4];
            if (code < 0) continue;
            const int f = code >> 2, m = code & 3;
            const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
            const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
            const long long base = (long long)f * P.vox + ((long long)io * P.ny + jo) * P.nz + zb + 4 * g;
            if (fast) {
                if (want)
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        *reinterpret_cast<float4*>(dst_arr + base + 16 * t) = make_float4(w[t][0], w[t][1], w[t][2], w[t][3]);
            } else if (want) {
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (zb + 16 * t + 4 * g + r < P.nz) dst_arr[base + 16 * t + r] = w[t][r];
            }
            if (P.flags & 4u)
#pragma unroll
                for (int t = 0; t < MT; ++t)
#pragma unroll
                    for (int r = 0; r < 4; ++r)
                        if (zb + 16 * t + 4 * g + r < P.nz) cplx[2 * (base + 16 * t + r) + part] = v[t][r];
        }
    }
}

// ------------------------------------------------------------------------------------
// kernel 2d: lattice accumulate.  Matrix arrays (Transducer.gen_matrix_array, xdc/transducer.py:372-406)
// put their elements on a regular (a, b) lattice with pitch (px, py); when the pitch is a whole number of
// voxels (px = mx hx, py = my hy) the geometry term depends only on the INTEGER offset between voxel and
// element:      G(v, e) = g(i - mx a, j - my b, k),
// i.e. the contraction over elements is a dilated 2-D convolution and the A operand of kernel 2c is
// block-Toeplitz.  An MFMA tile's 16 voxel rows are therefore pitch-strided:
//     row = (sxp, sy, sz):  i = ibase + 2 mx sxp (sxp < 2),  j = jbase + my sy (sy < 4),  plane k0 + sz MT + t,
// and a wave owns MT such tiles (2 MT consecutive planes).  Against one 8 x 8 "super-block" of elements
// (4 K-steps of 4 x 4 elements) the rows of a plane see only 10 x 11 distinct offsets, so the wave evaluates
// 110 G values per plane (the transcendentals + the fp16 hi/lo split) instead of 512, writes them to a
// wave-private LDS table and reads its A fragments back from there: lane (row, k-group g) needs the four
// elements (aa = 0..3, bb = g) of K-step (ka, kb) = table row (sy - 4 kb - g + 7), entries p .. p+3 with
// p = 2 - 2 sxp + 4 ka (the table row is stored reversed).  The x stride of two pitches makes p EVEN, so a
// fragment is two aligned ds_read_b64 per part with immediate offsets -- half the LDS cycles of 4-byte reads
// and no address arithmetic.  B fragments, the hi/lo three-product scheme and the columns / store targets
// are kernel 2c's.  Exact in the same sense: only WHERE a (voxel, element) term is evaluated changes.
// Array edges are padded to whole super-blocks with zero-weight virtual elements on the same lattice.
// VALU work per K-step drops ~8x against kernel 2c (which is VALU-bound); the limiters become the matrix
// pipe and LDS bandwidth.
// ------------------------------------------------------------------------------------
struct LatParams {
    int nx, ny, nz;            // slab extent in voxels
    int x_lo, y_lo;            // first computed voxel per folded axis (n/2) or 0
    int x_begin;               // slab start (global x index of local voxel 0)
    int mx, my;                // pitch / spacing (whole numbers)
    int tiles_x, tiles_y;      // row tiles per axis: blocks of 4 pitches x (2 mx | my) residues
    int kgroups;               // ceil(nz / (2 MT))
    int nsa, nsb;              // element super-blocks (8 x 8) per lattice axis
    int nsbp;                  // rows of the K-slot map (= nsb here; kernel 2e's NT = 2 shape pads it to an even count)
    int ux0, uy0;              // dx(i, a) = fx0 + (i_global + ux0 - mx a) hx   (integer part folded into ux0)
    float fx0, fy0;            // [wavelengths], |f| <= h/2
    float hx_hi, hx_lo, hy_hi, hy_lo, hz;  // spacing [wavelengths]; hi + lo = the fp64 value to ~2^-48
    float dmin2, flat_ez, g_scale, out_scale, inten_scale;
    long long vox;
    unsigned flags;
};

constexpr int LAT_ELEMS_LDS = 128;         // elements * NT of B fragments staged in LDS at a time (16 KiB)
constexpr int LAT_TW = 10;                 // words per table row (p = 0..9)
constexpr int LAT_PSZ = 116;               // words per plane table (11 rows x 10, padded so that the fragment reads of a
                                           // 32-lane group -- both sz planes, 5 table rows, 2 sxp -- fall on distinct banks)

__device__ __forceinline__ float quad_swap1(float v) {  // value of lane ^ 1 (DPP quad_perm [1,0,3,2]: no LDS traffic)
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xF, 0xF, true));
}

#ifdef OLX_EXP_STAMPS
__device__ unsigned long long g_stamps[4096][8];
#define OLX_STAMP(k) do { if (lane == 0 && wave < 4 && blockIdx.y == 0 && blockIdx.x % 37 == 0 && blockIdx.x / 37 < 1024) g_stamps[(blockIdx.x / 37) * 4 + wave][k] = __builtin_readcyclecounter(); } while (0)
#else
#define OLX_STAMP(k)
#endif

constexpr int LAT_THREADS = 512;           // 8 waves share one copy of the B fragments: 4 waves / SIMD at 2 blocks / CU

template <int MT, int NT, int MX, int MY, bool CLAMP>
__global__ __launch_bounds__(LAT_THREADS, NT >= 4 ? 2 : 4) void field_lattice_k(
    const uint4* __restrict__ bfrag /*[tiles][ks][NT][2][64]*/, float* __restrict__ pmag, float* __restrict__ inten,
    float* __restrict__ cplx, const int* __restrict__ targets, const LatParams P) {
    constexpr int SB_PER_CHUNK = (LAT_ELEMS_LDS / NT) / 64 > 0 ? (LAT_ELEMS_LDS / NT) / 64 : 1;  // super-blocks of B per LDS stage
    constexpr int ZW = 2 * MT;                       // planes per wave
    constexpr int NW = LAT_THREADS / 64;             // waves per block
    constexpr int ZB = NW * ZW;                      // planes per block
    // one LDS arena: [B fragments | per-wave G tables] during the K loop, re-used as the output staging buffer
    constexpr int CS = ZB + 4;                       // staging column stride [floats] (+4: conflict-free 16-B writes)
    constexpr int RS = 16 * CS;                      // staging row stride (8 (x, y) rows)
    constexpr int B_BYTES = SB_PER_CHUNK * 4 * NT * 2 * 64 * 16, T_BYTES = NW * 2 * ZW * LAT_PSZ * 4;
    constexpr int OUT_BYTES = 8 * RS * 4;
    constexpr int ARENA = B_BYTES + T_BYTES > OUT_BYTES ? B_BYTES + T_BYTES : OUT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[ARENA];
    typedef uint4 (*BArr)[NT][2][64];
    BArr s_B = reinterpret_cast<BArr>(smem);
    unsigned* const s_T = reinterpret_cast<unsigned*>(smem + B_BYTES);
    float* const s_out = reinterpret_cast<float*>(smem);
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, sxp = lane & 1, sy = (lane >> 1) & 3, sz = (lane >> 3) & 1;   // A-operand row = lane & 15
    // wave -> (row tile in x, row tile in y, group of ZW planes); the NW waves of a block take NW
    // consecutive plane groups of one row tile, so a block owns ZB consecutive floats per (voxel row, column).
    const unsigned kblocks = (unsigned)(P.kgroups + NW - 1) / NW;
    const unsigned kblock = blockIdx.x % kblocks, txy = blockIdx.x / kblocks;
    const int ty = (int)(txy % (unsigned)P.tiles_y), tx = (int)(txy / (unsigned)P.tiles_y);
    const int kgroup = (int)kblock * NW + wave;
    const bool active = kgroup < P.kgroups;
    const int k0 = kgroup * ZW;
    const int qx = tx / (2 * P.mx), qy = ty / P.my;
    const int ibase = P.x_lo + qx * 4 * P.mx + (tx - qx * 2 * P.mx);   // local voxel index of row sxp = 0 (sxp = 1: + 2 mx)
    const int jbase = P.y_lo + qy * 4 * P.my + (ty - qy * P.my);
    float dz2[ZW];
#pragma unroll
    for (int z = 0; z < ZW; ++z) {
        const float dz = (float)(k0 + z) * P.hz - P.flat_ez;
        dz2[z] = dz * dz;
    }
    // table-generation role of this lane: entries n = lane and lane + 64 of the 11 x 10 offset table of a plane
    int Ur[2], Wr[2], toff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int n = lane + 64 * r;
        const int nn = n < 110 ? n : 109;
        const int wi = nn / 10, p = nn - 10 * wi;
        Ur[r] = ibase + P.x_begin + P.ux0 + P.mx * (2 - p);
        Wr[r] = jbase + P.uy0 + P.my * (wi - 7);
        toff[r] = n < 110 ? nn : LAT_PSZ - 1;          // spare lanes write the pad word
    }
    unsigned* const Thi = s_T + (wave * 2 + 0) * ZW * LAT_PSZ;
    unsigned* const Tlo = s_T + (wave * 2 + 1) * ZW * LAT_PSZ;
    // fragment read base of K-step (0, 0), tile 0: plane sz MT, row (sy - g + 7), entry 2 - 2 sxp  (even -> 8-B aligned)
    const int rbase = sz * MT * LAT_PSZ + (sy - g + 7) * LAT_TW + (2 - 2 * sxp);

    floatx4_t acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};

    const int nsbp = P.nsbp;                    // (= nsb: this kernel is never planned with the padded K-slot map)
    const int n_sb = P.nsa * nsbp;
    OLX_STAMP(0);
    // B fragments: chunk c+1 is fetched into registers while chunk c is contracted (the loads stay in flight across
    // the K-steps), then handed to LDS between two barriers -- no wave waits for global memory inside the loop.
    constexpr int CHUNK_U4 = SB_PER_CHUNK * 4 * NT * 128, PRE = CHUNK_U4 / LAT_THREADS;
    static_assert(CHUNK_U4 % LAT_THREADS == 0, "chunk must split evenly over the block");
    uint4 pre[PRE];
    const uint4* const bsrc = bfrag + (size_t)tile * n_sb * (4 * NT * 128);
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int idx = tid + q * LAT_THREADS;
        pre[q] = idx < n_sb * 4 * NT * 128 ? bsrc[idx] : make_uint4(0, 0, 0, 0);
    }
    for (int sb0 = 0; sb0 < n_sb; sb0 += SB_PER_CHUNK) {
        const int sb_here = min(SB_PER_CHUNK, n_sb - sb0);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PRE; ++q) reinterpret_cast<uint4*>(smem)[tid + q * LAT_THREADS] = pre[q];
        __syncthreads();
        {
            const int nxt = (sb0 + SB_PER_CHUNK) * 4 * NT * 128, lim = n_sb * 4 * NT * 128;
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int idx = nxt + tid + q * LAT_THREADS;
                if (idx < lim) pre[q] = bsrc[idx];
            }
        }
        if (!active) continue;
        if (sb0 == 0) OLX_STAMP(1);
        for (int sbl = 0; sbl < sb_here; ++sbl) {
            const int sb = sb0 + sbl;
            const int sa = sb / nsbp, sbb = sb - sa * nsbp;      // sa-major order (host slot map)
            // ---- G table of this super-block: 110 offsets x ZW planes (2 rounds of 64 lanes per plane)
#ifndef OLX_EXP_NOTGEN
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float U = (float)(Ur[r] - 8 * P.mx * sa), W = (float)(Wr[r] - 8 * P.my * sbb);
                const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
                const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
                const float r2 = fmaf(dy, dy, dx * dx);
#pragma unroll
                for (int z = 0; z < ZW; ++z) {
                    float d2 = r2 + dz2[z];
                    if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                    const float ri = __builtin_amdgcn_rsqf(d2);
                    const float ph = d2 * ri;            // distance in wavelengths = phase in revolutions
                    const float rs = ri * P.g_scale;
                    const float gr = rs * __builtin_amdgcn_cosf(ph);
                    const float gi = rs * __builtin_amdgcn_sinf(ph);
                    const auto hi = __builtin_amdgcn_cvt_pkrtz(gr, gi);
                    const auto lo = __builtin_amdgcn_cvt_pkrtz(gr - (float)hi[0], gi - (float)hi[1]);
                    Thi[z * LAT_PSZ + toff[r]] = __builtin_bit_cast(unsigned, hi);
                    Tlo[z * LAT_PSZ + toff[r]] = __builtin_bit_cast(unsigned, lo);
                }
            }
#endif
            // the table is wave-private: DS operations of one wave execute in order, only the compiler must not
            // move the fragment reads above the table writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (sb == 0) OLX_STAMP(2);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ka = ks & 1, kb = ks >> 1;
                Half8Bits bh[NT], bl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt].u = s_B[sbl * 4 + ks][nt][0][lane];
                    bl[nt].u = s_B[sbl * 4 + ks][nt][1][lane];
                }
                const int roff = rbase - 4 * kb * LAT_TW + 4 * ka;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    Half8Bits ah, al;
#ifdef OLX_EXP_NOAREAD
#pragma unroll
                    for (int q = 0; q < 4; ++q) { ah.w[q] = 0x3c003c00u + t + q + ks; al.w[q] = 0x1c001c00u + t + q; }
#else
                    // four separate ds_read_b64 (2 LDS cycles each, 64-bank mode).  Relaxed atomic loads keep the
                    // compiler from fusing them into ds_read2_b64, which runs at a quarter of that rate.
                    const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(Thi + t * LAT_PSZ + roff);
                    const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(Tlo + t * LAT_PSZ + roff);
                    const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    ah.w[0] = (unsigned)h0; ah.w[1] = (unsigned)(h0 >> 32); ah.w[2] = (unsigned)h1; ah.w[3] = (unsigned)(h1 >> 32);
                    al.w[0] = (unsigned)l0; al.w[1] = (unsigned)(l0 >> 32); al.w[2] = (unsigned)l1; al.w[3] = (unsigned)(l1 >> 32);
#endif
#ifdef OLX_EXP_NOMFMA
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt][0] += (float)bh[nt].h[0] + (float)bl[nt].h[0];
#pragma unroll
                    for (int q = 0; q < 4; ++q) asm volatile("" :: "v"(ah.w[q]), "v"(al.w[q]));
                    continue;
#endif
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bl[nt].h, acc[t][nt], 0, 0, 0);
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (sb == 0) OLX_STAMP(3);
        }
    }
    OLX_STAMP(4);
    // epilogue.  D layout: lane holds rows 4 (lane >> 4) + r of column lane & 15 = (o, part); even lanes own Re, odd
    // lanes Im of output o, one DPP swap gives |p|^2 to both; the even lane keeps |p|, the odd lane the intensity.
    // A lane's MT planes are only 4 MT contiguous bytes and its neighbours belong to other columns / rows, so the
    // values are transposed through LDS: the block's 4 waves hold ZB consecutive planes of the same 8 (x, y) rows x
    // 16 columns, and every (row, column, store target) leaves as one contiguous 4 ZB-byte run written by ZB / 4
    // adjacent lanes (full 128-B lines at ZB = 32).
    const int c16 = lane & 15, part = c16 & 1, gy = lane >> 4;
    const int kb0 = (int)kblock * ZB;                // first plane of the block
    const bool fast = (P.nz % ZB) == 0;              // whole 16-B aligned runs
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        __syncthreads();                             // K loop / previous read-out done with the arena
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * gy + r;          // (sxp, sy, sz) = (row & 1, (row >> 1) & 3, row >> 3)
                float w[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const float v = acc[t][nt][r] * P.out_scale;
                    const float sq = v * v;
                    const float m2 = sq + quad_swap1(sq);          // re^2 + im^2 (partner lane holds the other part)
                    w[t] = part == 0 ? __builtin_amdgcn_sqrtf(m2) : m2 * P.inten_scale;
                }
                float* dst = s_out + (row & 7) * RS + c16 * CS + wave * ZW + (row >> 3) * MT;
                if (MT % 4 == 0) {
#pragma unroll
                    for (int t4 = 0; t4 < MT / 4; ++t4)
                        *reinterpret_cast<float4*>(dst + 4 * t4) = make_float4(w[4 * t4], w[4 * t4 + 1], w[4 * t4 + 2], w[4 * t4 + 3]);
                } else {
#pragma unroll
                    for (int t = 0; t < MT; ++t) dst[t] = w[t];
                }
            }
        }
        __syncthreads();
        if (nt == 0) OLX_STAMP(5);
        // read-out: thread -> (column, (x, y) rows r0, r0 + RSTEP, ..., piece of 4 planes); ZB / 4 adjacent lanes write
        // one contiguous run.  Column, piece and the column's store targets are fixed per thread, so the
        // (focus, mirror image) bases are formed once and each store needs one row offset.
        constexpr int PIECES = ZB / 4, RSTEP = LAT_THREADS / (PIECES * 16);
        static_assert(LAT_THREADS % (PIECES * 16) == 0 && 8 % RSTEP == 0, "read-out map");
        {
            const int piece = tid % PIECES, col = (tid / PIECES) & 15, r0 = tid / (PIECES * 16);
            const int kz = kb0 + 4 * piece;
            const bool is_p = (col & 1) == 0;
            float* const arr = is_p ? pmag : inten;
            const bool want = (is_p ? (P.flags & 1u) : (P.flags & 2u)) != 0 && kz < P.nz;
            const int4 tg = reinterpret_cast<const int4*>(targets)[(size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt * MFMA_COLS + (col >> 1)];
            const int tgs[4] = {tg.x, tg.y, tg.z, tg.w};
            const float* src = s_out + col * CS + 4 * piece;
#ifdef OLX_EXP_NOSTORE
            if (kz == 123456)
#endif
            if (want && fast) {
                float* tb[4]; bool tfx[4], tfy[4];   // per store target: volume base, mirror flags (hoisted out of the rows)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int code = tgs[s4], m = code & 3;
                    tb[s4] = code < 0 ? nullptr : arr + (long long)(code >> 2) * P.vox + kz;
                    tfx[s4] = (MX == 2) && (m & 1);
                    tfy[s4] = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                }
#pragma unroll
                for (int it = 0; it < 8 / RSTEP; ++it) {
                    const int row = r0 + RSTEP * it;
                    const int i = ibase + (row & 1) * 2 * P.mx, j = jbase + (row >> 1) * P.my;
                    if (i >= P.nx || j >= P.ny) continue;
                    const float4 val = *reinterpret_cast<const float4*>(src + row * RS);
                    const int ai = i * P.ny, aX = (P.nx - 1 - i) * P.ny, bY = P.ny - 1 - j;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        if (!tb[s4]) continue;
                        const int off = ((tfx[s4] ? aX : ai) + (tfy[s4] ? bY : j)) * P.nz;
                        *reinterpret_cast<float4*>(tb[s4] + off) = val;
                    }
                }
            } else if (want) {                       // ragged nz: guarded scalar stores (not a throughput path)
#pragma unroll 1
                for (int row = r0; row < 8; row += RSTEP) {
                    const int i = ibase + (row & 1) * 2 * P.mx, j = jbase + (row >> 1) * P.my;
                    if (i >= P.nx || j >= P.ny) continue;
#pragma unroll 1
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int code = tgs[s4];
                        if (code < 0) continue;
                        const int m = code & 3;
                        const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                        const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
                        float* o = arr + (long long)(code >> 2) * P.vox + ((long long)io * P.ny + jo) * P.nz + kz;
#pragma unroll 1
                        for (int q = 0; q < 4; ++q) if (kz + q < P.nz) o[q] = src[row * RS + q];
                    }
                }
            }
        }
        if (nt == NT - 1) OLX_STAMP(6);
        if ((P.flags & 4u) && active) {              // complex output (not a throughput path): direct scalar stores
            const int4 tg = reinterpret_cast<const int4*>(targets)[(size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt * MFMA_COLS + (c16 >> 1)];
            const int tgs[4] = {tg.x, tg.y, tg.z, tg.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * gy + r;
                const int i = ibase + (row & 1) * 2 * P.mx, j = jbase + ((row >> 1) & 3) * P.my, kz0 = k0 + (row >> 3) * MT;
                if (i >= P.nx || j >= P.ny) continue;
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int code = tgs[s4];
                    if (code < 0) continue;
                    const int f = code >> 2, m = code & 3;
                    const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                    const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
                    const long long base = (long long)f * P.vox + ((long long)io * P.ny + jo) * P.nz + kz0;
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        if (kz0 + t < P.nz) cplx[2 * (base + t) + part] = acc[t][nt][r] * P.out_scale;
                }
            }
        }
    }
}

// ------------------------------------------------------------------------------------
// kernel 2e: lattice accumulate, whole cosets per wave.  Same mathematics and operands as kernel 2d; the row map is
// changed to remove 2d's two costs: its row tiles overhang the (half) axis (27 % of all MFMA rows on BASELINE's
// grids) and each 8-position tile evaluates its own G table.
//   * The voxels of one plane that share a lattice coset -- x = xbase + 2 mx kx, y = ybase + my ky -- form a
//     KX x KY grid of positions (KX <= 6, KY <= 11: the whole half axis at 128 voxels / 12-voxel pitch; longer
//     axes are cut into equal parts on the host).  ALL of them see the same offsets against an 8 x 8 element
//     super-block: ud = 2 kx - a in [-7, 10], wd = ky - b in [-7, 10], i.e. ONE 18 x 18 table per plane serves up
//     to 66 positions (4.9 - 6.5 entries per position against 13.75 in kernel 2d).
//   * A wave owns such a position grid on 2 consecutive planes; its MFMA rows are simply n = 0 .. 2 KX KY - 1
//     (plane-major, then kx, then ky), 16 per tile, so only the last tile of a wave can hold padding (2 - 6 %).
//     MT = ceil(2 KX KY / 16) <= 9 tiles, all sharing the K-step's B fragments.
//   * Fragment of row n, K-step (ka, kb), k-group g: table row ky - 4 kb - g + 7, entries p .. p+3,
//     p = 10 - 2 kx + 4 ka (even: two aligned ds_read_b64 per part, as in 2d).  TW = 20, plane stride 378 words:
//     conflict-free for the row sets that occur (brute-forced, 1.03 LDS cycles per access).
//   * 8 waves = 16 consecutive planes per block; the epilogue transposes through LDS in two halves of the position
//     grid and writes 64-byte z runs.
//   * NT <= 2 (SHARE): the tables of the super-blocks (sa, 2p) and (sa, 2p + 1) overlap in 10 of 18 rows, so one 26-row
//     table per PAIR is evaluated and the second super-block reads it 8 rows lower (K-slot map sa-major, rows padded to even).
//   * NT <= 2, FP8: the two hi x lo correction products of the fp16 split take e4m3 operands -- one
//     v_mfma_scale_f32_16x16x128_f8f6f4 per two K-steps instead of four fp16 MFMAs (cos_fp8 below; host-gated).
// ------------------------------------------------------------------------------------
struct CosetParams {
    int nx, ny, nz;
    int x_lo, y_lo, x_begin;
    int mx, my;
    int nsx, nsy;              // parts the coset's positions are cut into along x / y
    int kblocks;               // plane blocks of COS_ZB planes
    int nsa, nsb;
    int nsbp;                  // rows of the K-slot map: nsb, padded to an even count for the NT = 2 shape (shared pair tables)
    int ux0, uy0;
    float fx0, fy0, hx_hi, hx_lo, hy_hi, hy_lo, hz;
    float dmin2, flat_ez, g_scale, out_scale, inten_scale;
    long long vox;
    unsigned flags;
};

constexpr int COS_NW = 8;                  // waves per block
constexpr int COS_P = 2;                   // planes per wave
constexpr int COS_ZB = COS_NW * COS_P;     // planes per block
constexpr int COS_KYW = 11;                // positions per wave along y (18 table rows)
constexpr int COS_JOBS = 64;               // store jobs per column tile: 16 (column, part) x up to 4 targets
// positions per wave along x: 6 (the whole half axis at 128 voxels / 12-voxel pitch; table 18 x 18, 9 MFMA tiles) when
// one column tile leaves registers for 36 accumulators, 3 (table 18 x 12, 5 tiles) with two column tiles, 2 (table
// 18 x 10, 3 tiles) with four
constexpr int cos_kxw(int nt) { return nt >= 4 ? 2 : (nt >= 2 ? 3 : 6); }
// fp8 correction products (NT <= 2; the NT = 4 shape has no registers for the second operand set): the two hi x lo terms of
// the fp16 hi/lo split only need their hi factor to 2^-4, so both go through ONE v_mfma_scale_f32_16x16x128_f8f6f4 per two
// K-steps with e4m3 operands -- bytes [lo re, lo im, hi re, hi im] per element against [hi(k0), hi(k1), lo(k0), lo(k1)] of
// the steering column.  hi parts are <= 2^14 and lo parts < 8 in both operands (host scales), so lo * 2^5 and hi * 2^-6 stay
// <= 256 (e4m3 overflows to NaN above 448); the instruction's E8M0 block scales (2^1, 2^0) undo the 2^-1 of each product.
constexpr bool cos_fp8(int nt) { return nt <= 2; }
constexpr float COS_F8_LO = 32.0f, COS_F8_HI = 1.0f / 64.0f;
typedef int intx8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

template <int V> struct IntC { static constexpr int value = V; };

template <int NT, int MX, int MY, bool CLAMP, bool FP8>
__global__ __launch_bounds__(COS_NW * 64, 4) void field_coset_k(
    const uint4* __restrict__ bfrag, float* __restrict__ pmag, float* __restrict__ inten, float* __restrict__ cplx,
    const int* __restrict__ jobs /*[tiles][MFMA_MAX_NT][COS_JOBS + 1]: dense (column, focus, image) store jobs, [COS_JOBS] = log2 count*/,
    const CosetParams P) {
    constexpr int THREADS = COS_NW * 64;
    static_assert(!FP8 || cos_fp8(NT), "fp8 correction products need NT <= 2");
    constexpr int COS_KXW = cos_kxw(NT);
    constexpr int UW = 8 + 2 * (COS_KXW - 1);                       // table columns: ud = 2 kx - a in [-7, 2 (KXW - 1)]
    constexpr int COS_MT = (COS_P * COS_KXW * COS_KYW + 15) / 16;   // MFMA tiles per wave
    // table row / plane stride [words], conflict-free for the row sets that occur (brute-forced per shape)
    // SHARE (NT <= 2): the tables of the two super-blocks (sa, 2p) and (sa, 2p + 1)
    // overlap in 10 of their 18 rows (offsets wd = ky - b), so ONE 26-row table serves both: 26 instead of 36 rows to evaluate
    constexpr bool SHARE = NT <= 2;                                 // (NT = 4: measured +1 %, it spills 7 registers)
    constexpr int TROWS = SHARE ? 26 : 18, ROW0 = SHARE ? 15 : 7;   // table rows; row of offset wd = 0
    // (plane strides keep the residues mod 64 of the brute-forced 378 / 216 / 184 of the 18-row tables)
    constexpr int COS_TW = COS_KXW == 6 ? 20 : (COS_KXW == 3 ? 12 : 10), COS_PSZ = COS_KXW == 6 ? 570 : (COS_KXW == 3 ? 344 : 184);
    static_assert(TROWS * COS_TW <= COS_PSZ, "table does not fit its plane stride");
    constexpr int RPR = 64 / UW, NROUND = (TROWS + RPR - 1) / RPR;  // table rows per generation round, rounds
    // super-blocks of B fragments per LDS stage: 2 for every NT but 4 (NT = 2 has LDS to spare at its 2 blocks / CU; two
    // super-blocks per stage halve the barriers)
    constexpr int SB_PER_CHUNK = NT == 2 ? 2 : 1;                // (NT = 1: its 26 x 18 pair tables leave LDS for one super-block of B only)
    // staging strides [floats]: odd column stride and row stride = 4 (mod 8) spread the 64 lanes of a staging write
    // (16 columns x 4 row groups) over all 32 banks (2-way, which is free for ds_write_b32)
    constexpr int CS = COS_ZB + 1, RS = 16 * CS + 4;
    constexpr int B_BYTES = SB_PER_CHUNK * 4 * NT * 2 * 64 * 16, T_BYTES = COS_NW * 2 * COS_P * COS_PSZ * 4;
    constexpr int GROUP = NT >= 4 ? 2 : NT;                      // column tiles staged per epilogue pass
    constexpr int SLAB = COS_KXW * COS_KYW * RS;                 // floats per staged column tile (whole position grid)
    constexpr int OUT_BYTES = GROUP * SLAB * 4;
    constexpr int ARENA = B_BYTES + T_BYTES > OUT_BYTES ? B_BYTES + T_BYTES : OUT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[ARENA];
    typedef uint4 (*BArr)[NT][2][64];
    BArr s_B = reinterpret_cast<BArr>(smem);
    unsigned* const s_T = reinterpret_cast<unsigned*>(smem + B_BYTES);
    float* const s_out = reinterpret_cast<float*>(smem);
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4;
    // block -> (x coset, y coset, x part, y part, plane block)
    unsigned b = blockIdx.x;
    // The two blocks that write the two 64-byte halves of the same 128-byte lines get block ids 8 apart (same XCD, i.e.
    // the same L2, under round-robin dispatch over the 8 XCDs) instead of adjacent ids (measured -1...-2 %).
    int kblock;
    if ((P.kblocks & 1) == 0 && gridDim.x % 16 == 0) {
        const unsigned xcd = b % 8, sft = b / 8, kb_lo = sft % 2, u = (sft / 2) * 8 + xcd, half = (unsigned)P.kblocks / 2;
        kblock = (int)(2 * (u % half) + kb_lo); b = u / half;
    } else { kblock = (int)(b % (unsigned)P.kblocks); b /= (unsigned)P.kblocks; }
    const int sy_part = (int)(b % (unsigned)P.nsy); b /= (unsigned)P.nsy;
    const int sx_part = (int)(b % (unsigned)P.nsx); b /= (unsigned)P.nsx;
    const int ry = (int)(b % (unsigned)P.my), rx = (int)(b / (unsigned)P.my);          // rx < 2 mx
    const int wx = P.nx - P.x_lo, wy = P.ny - P.y_lo;
    const int kx_all = rx < wx ? (wx - 1 - rx) / (2 * P.mx) + 1 : 0, ky_all = ry < wy ? (wy - 1 - ry) / P.my + 1 : 0;
    // equal parts: part s of n covers [s K / n, (s+1) K / n)
    const int kx0 = sx_part * kx_all / P.nsx, KX = (sx_part + 1) * kx_all / P.nsx - kx0;
    const int ky0 = sy_part * ky_all / P.nsy, KY = (sy_part + 1) * ky_all / P.nsy - ky0;
    const int npos = KX * KY, nrow = COS_P * npos;
    const int ibase = P.x_lo + rx + 2 * P.mx * kx0, jbase = P.y_lo + ry + P.my * ky0;
    const int k0 = (kblock * COS_NW + wave) * COS_P;
    const bool active = npos > 0 && k0 < P.nz;
    const int ntile = (nrow + 15) >> 4;              // block-uniform (<= COS_MT)
    const float inv_ky = KY > 0 ? 1.0f / (float)KY : 0.f;
    float dz2[COS_P];
#pragma unroll
    for (int z = 0; z < COS_P; ++z) {
        const float dz = (float)(k0 + z) * P.hz - P.flat_ez;
        dz2[z] = dz * dz;
    }
    // table generation role: lane -> (wl = lane / UW < RPR, ui = lane % UW); round r covers table rows RPR r + wl
    const int wl = lane / UW, ui = lane - UW * wl;
    const bool gen_lane = wl < RPR;
    const int Ulane = ibase + P.x_begin + P.ux0 + P.mx * (ui - 7);
    const int Wlane = jbase + P.uy0 + P.my * (wl - ROW0);
    const int tw_off = wl * COS_TW + (UW - 1 - ui);   // + RPR r TW per round
    unsigned* const Thi = s_T + (wave * 2 + 0) * COS_P * COS_PSZ;
    unsigned* const Tlo = s_T + (wave * 2 + 1) * COS_P * COS_PSZ;
    // fragment read offset of every tile's row for K-step (0, 0)
    int roffT[COS_MT];
#pragma unroll
    for (int t = 0; t < COS_MT; ++t) {
        int n = 16 * t + (lane & 15);
        n = n < nrow ? n : (nrow > 0 ? nrow - 1 : 0);
        const int plane = n >= npos ? 1 : 0, pos = n - plane * npos;             // COS_P == 2
        const int kx = (int)(((float)pos + 0.5f) * inv_ky), ky = pos - kx * KY;   // exact for these small integers
        roffT[t] = plane * COS_PSZ + (ky - g + ROW0) * COS_TW + (UW - 8 - 2 * kx);
    }
    floatx4_t acc[COS_MT][NT];
#pragma unroll
    for (int t = 0; t < COS_MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};

    const int nsbp = P.nsbp;                    // SHARE: nsb padded to an even count, so chunks = table pairs never straddle sa
    const int n_sb = P.nsa * nsbp;
    OLX_STAMP(0);
#ifdef OLX_EXP_STAGGER
    if (blockIdx.y == 0 && blockIdx.x < 768) {   // first-round blocks start staggered (A/B: do store bursts of lock-stepped blocks add up?)
        const long long t0 = __builtin_readcyclecounter(), wait = (long long)((blockIdx.x * 7) % 16) * (OLX_EXP_STAGGER);
        while ((long long)__builtin_readcyclecounter() - t0 < wait) __builtin_amdgcn_s_sleep(32);
    }
#endif
    constexpr int CHUNK_U4 = SB_PER_CHUNK * 4 * NT * 128, PRE = CHUNK_U4 / THREADS;
    static_assert(CHUNK_U4 % THREADS == 0, "chunk must split evenly over the block");
    uint4 pre[PRE];
    const uint4* const bsrc = bfrag + (size_t)tile * n_sb * (4 * NT * 128);
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int idx = tid + q * THREADS;
        pre[q] = idx < n_sb * 4 * NT * 128 ? bsrc[idx] : make_uint4(0, 0, 0, 0);
    }
    for (int sb0 = 0; sb0 < n_sb; sb0 += SB_PER_CHUNK) {
        const int sb_here = min(SB_PER_CHUNK, n_sb - sb0);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PRE; ++q) reinterpret_cast<uint4*>(smem)[tid + q * THREADS] = pre[q];
        __syncthreads();
        if (sb0 == 0) OLX_STAMP(1);
        {
            const int nxt = (sb0 + SB_PER_CHUNK) * 4 * NT * 128, lim = n_sb * 4 * NT * 128;
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int idx = nxt + tid + q * THREADS;
                if (idx < lim) pre[q] = bsrc[idx];
            }
        }
        if (!active) continue;
#pragma unroll                                          // (unrolled: the pair position becomes part of the immediate table offsets)
        for (int sbl = 0; sbl < SB_PER_CHUNK; ++sbl) {
            if (SB_PER_CHUNK > 1 && sbl >= sb_here) break;
            const int sb = sb0 + sbl;
            const int sa = sb / nsbp, sbb = sb - sa * nsbp;      // sa-major order (host slot map)
            if constexpr (SHARE) { if (sbb >= P.nsb) continue; }   // padding super-block of an odd count: zero weights, nothing to do
            // position in the table pair: the chunk index when a chunk is a pair (compile-time after unrolling: it goes into
            // the immediate table offsets), else the parity of sbb (then the table pointers move)
            const int sl = !SHARE ? 0 : (SB_PER_CHUNK == 2 ? sbl : (sbb & 1));
            const int sl_imm = SB_PER_CHUNK == 2 ? sl : 0;
            // ---- G table: TROWS x UW offsets x 2 planes (SHARE: once per super-block pair)
            if (sl == 0) {
                const float U = (float)(Ulane - 8 * P.mx * sa);
                const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
                const float dx2 = dx * dx;
                const int Wsb = Wlane - 8 * P.my * sbb;
#pragma unroll 2
                for (int r = 0; r < NROUND; ++r) {
                    const bool row_ok = gen_lane && RPR * r + wl < TROWS;  // the last round may run past the table
                    const float W = (float)(Wsb + RPR * P.my * r);
                    const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
                    const float r2 = fmaf(dy, dy, dx2);
#pragma unroll
                    for (int z = 0; z < COS_P; ++z) {
                        float d2 = r2 + dz2[z];
                        if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                        const float ri = __builtin_amdgcn_rsqf(d2);
                        const float ph = d2 * ri;
                        const float rs = ri * P.g_scale;
                        const float gr = rs * __builtin_amdgcn_cosf(ph);
                        const float gi = rs * __builtin_amdgcn_sinf(ph);
                        // fp8 corrections: hi rounded to nearest (v_cvt_pk_f16_f32) so that |lo| <= half an ulp
                        half2_t hi;
                        if constexpr (FP8) hi = __builtin_convertvector(float2_t{gr, gi}, half2_t);
                        else hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
                        unsigned lo_word;
                        if constexpr (FP8) {             // e4m3 bytes [lo re, lo im | hi re, hi im], |.| <= 256 (448 overflows to NaN)
                            int w = __builtin_amdgcn_cvt_pk_fp8_f32((gr - (float)hi[0]) * COS_F8_LO, (gi - (float)hi[1]) * COS_F8_LO, 0, false);
                            w = __builtin_amdgcn_cvt_pk_fp8_f32(gr * COS_F8_HI, gi * COS_F8_HI, w, true);
                            lo_word = (unsigned)w;
                        } else {
                            lo_word = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(gr - (float)hi[0], gi - (float)hi[1]));
                        }
                        if (row_ok) {                    // spare lanes / rows past the table do not store
                            const int o = z * COS_PSZ + tw_off + RPR * r * COS_TW;
                            Thi[o] = __builtin_bit_cast(unsigned, hi);
                            Tlo[o] = lo_word;
                        }
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (sb == 0) OLX_STAMP(2);
            const unsigned* const Th = Thi - (SB_PER_CHUNK == 2 ? 0 : 8 * sl * COS_TW);
            const unsigned* const Tl = Tlo - (SB_PER_CHUNK == 2 ? 0 : 8 * sl * COS_TW);
            if constexpr (FP8) {
#pragma unroll
            for (int kb = 0; kb < 2; ++kb) {         // K-step pairs (ka = 0, 1): two fp16 hi*hi products + ONE fp8 product
                Half8Bits bh[2][NT];                 // for both correction terms of both K-steps (K = 128 e4m3 values)
                intx8_t b8[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                    for (int ka = 0; ka < 2; ++ka) {
                        bh[ka][nt].u = s_B[sbl * 4 + 2 * kb + ka][nt][0][lane];
                        const uint4 q = s_B[sbl * 4 + 2 * kb + ka][nt][1][lane];
                        b8[nt][4 * ka + 0] = (int)q.x; b8[nt][4 * ka + 1] = (int)q.y; b8[nt][4 * ka + 2] = (int)q.z; b8[nt][4 * ka + 3] = (int)q.w;
                    }
                }
#pragma unroll
                for (int t = 0; t < COS_MT; ++t) {
                    if (t >= ntile) continue;            // block-uniform
                    Half8Bits ah[2];
                    intx8_t a8;
#pragma unroll
                    for (int ka = 0; ka < 2; ++ka) {
                        const int kso = 4 * ka - (4 * kb + 8 * sl_imm) * COS_TW;   // the pair's second super-block reads 8 table rows lower
                        const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(Th + roffT[t] + kso);
                        const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(Tl + roffT[t] + kso);
                        const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        ah[ka].w[0] = (unsigned)h0; ah[ka].w[1] = (unsigned)(h0 >> 32); ah[ka].w[2] = (unsigned)h1; ah[ka].w[3] = (unsigned)(h1 >> 32);
                        a8[4 * ka + 0] = (int)(unsigned)l0; a8[4 * ka + 1] = (int)(unsigned)(l0 >> 32);
                        a8[4 * ka + 2] = (int)(unsigned)l1; a8[4 * ka + 3] = (int)(unsigned)(l1 >> 32);
                    }
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[0].h, bh[0][nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[1].h, bh[1][nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt)      // E8M0 scales undo the operand scaling: 2^(128 - 127) * COS_F8_LO * COS_F8_HI = 1
                        acc[t][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8[nt], acc[t][nt], 0, 0, 0, 128, 0, 127);
                }
            }
            } else {
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {         // unrolled: the K-step's table offset becomes an immediate
                const int ka = ks & 1, kb = ks >> 1;
                Half8Bits bh[NT], bl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt].u = s_B[sbl * 4 + ks][nt][0][lane];
                    bl[nt].u = s_B[sbl * 4 + ks][nt][1][lane];
                }
                const int kso = 4 * ka - (4 * kb + 8 * sl_imm) * COS_TW;   // the pair's second super-block reads 8 table rows lower
#pragma unroll
                for (int t = 0; t < COS_MT; ++t) {
                    if (t >= ntile) continue;            // block-uniform
                    Half8Bits ah, al;
                    const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(Th + roffT[t] + kso);
                    const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(Tl + roffT[t] + kso);
                    const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                    ah.w[0] = (unsigned)h0; ah.w[1] = (unsigned)(h0 >> 32); ah.w[2] = (unsigned)h1; ah.w[3] = (unsigned)(h1 >> 32);
                    al.w[0] = (unsigned)l0; al.w[1] = (unsigned)(l0 >> 32); al.w[2] = (unsigned)l1; al.w[3] = (unsigned)(l1 >> 32);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bl[nt].h, acc[t][nt], 0, 0, 0);
                }
            }
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (sb == 0) OLX_STAMP(3);
        }
    }
    OLX_STAMP(4);
    // ---- epilogue.  D layout: lane holds rows 4 (lane >> 4) + r of tile t = rows n = 16 t + 4 gy + r -> (plane, position),
    // column lane & 15 = (o, part).  Staging [position][column][plane of the block], two halves of the position grid.
    const int c16 = lane & 15, part = c16 & 1, gy = lane >> 4;
    const int kb0 = kblock * COS_ZB;
    const bool fast = (P.nz % COS_ZB) == 0;
    // |p| / intensity in place, then one staged pass per column tile (complex output is served by kernel 2d: the host
    // does not select this kernel when OLX_OUT_COMPLEX is planned)
    // The |p| lane (part 0) and its partner, the intensity lane (part 1), hold the same (S re)^2 + (S im)^2 for every row, and
    // only the |p| lane needs its root: per pair of rows the |p| lane takes the root of the first and the partner lane of the
    // second (handed back through the quad swap) -- one quarter-rate instruction per two rows instead of two.
    const float s_lane = part == 0 ? P.out_scale : P.out_scale * P.out_scale * P.inten_scale;   // scales applied after the square
#pragma unroll
    for (int t = 0; t < COS_MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt)
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float a0 = acc[t][nt][r], a1 = acc[t][nt][r + 1];
                const float sq0 = a0 * a0, sq1 = a1 * a1;
                const float m0 = sq0 + quad_swap1(sq0), m1 = sq1 + quad_swap1(sq1);
                const float y = __builtin_amdgcn_sqrtf(part == 0 ? m0 : m1);
                const float ys = quad_swap1(y);
                acc[t][nt][r] = (part == 0 ? y : m0) * s_lane;
                acc[t][nt][r + 1] = (part == 0 ? ys : m1) * s_lane;
            }
    // (the column tile is a compile-time argument so that the accumulators keep static indices; the pass loop is rolled)
    auto stage_and_store = [&](auto nt_c) {          // one pass: GROUP column tiles nt0 .. nt0 + GROUP - 1, two barriers
            constexpr int nt0 = decltype(nt_c)::value;
            __syncthreads();                         // arena free (K loop / previous read-out done)
            if (nt0 == 0) OLX_STAMP(5);
            if (active) {
                // (the row -> position arithmetic is loop-invariant; the opaque copy keeps the compiler from hoisting all
                // 36 of them out of the pass loop, which costs > 100 registers)
                int n0 = 4 * gy;
                asm volatile("" : "+v"(n0));
                // staging address of row n = n0 + 16 t + r: one multiply per lane, then immediates; rows of the second plane
                // (n >= npos) sit one float further and npos rows back
                float* const a0 = s_out + n0 * RS + c16 * CS + wave * COS_P;
                const int dplane = 1 - npos * RS;
#pragma unroll
                for (int t = 0; t < COS_MT; ++t) {
                    if (t >= ntile) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int n = 16 * t + n0 + r;
                        if (n < nrow) {
                            float* a = a0 + (16 * t + r) * RS + (n >= npos ? dplane : 0);
#pragma unroll
                            for (int gq = 0; gq < GROUP; ++gq) a[gq * SLAB] = acc[t][nt0 + gq][r];
                        }
                    }
                }
            }
            __syncthreads();
            if (nt0 == 0) OLX_STAMP(6);
#pragma unroll
            for (int gq = 0; gq < GROUP; ++gq) {
            const int nt = nt0 + gq;
            // read-out.  The host lists the (column, focus, mirror image) store jobs of this column tile densely
            // (count padded to a power of two), so thread -> (piece of 4 planes, job, positions q0, q0 + step, ...) keeps
            // every lane of a store instruction busy whatever the number of targets per column is.
            constexpr int PIECES = COS_ZB / 4;
            static_assert(PIECES == 4, "piece index is two bits");
            const int* jb = jobs + ((size_t)tile * MFMA_MAX_NT + nt) * (COS_JOBS + 1);
            const int lg = jb[COS_JOBS];
            const int piece = tid & 3, jidx = (tid >> 2) & ((1 << lg) - 1), q0 = tid >> (2 + lg), qstep = THREADS >> (2 + lg);
            const int job = jb[jidx];
            const int kz = kb0 + 4 * piece;
            if (job >= 0 && kz < P.nz && npos > 0) {
                const int col = job & 15, m = (job >> 4) & 3;
                const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                float* const base = ((col & 1) ? inten : pmag) + (long long)(job >> 6) * P.vox + kz;
                const float* sv = s_out + gq * SLAB + col * CS + 4 * piece;
                int kx = (int)(((float)q0 + 0.5f) * inv_ky), ky = q0 - kx * KY;     // then carried
                // RU positions per trip: their LDS reads and address arithmetic first, then RU stores back to back, so that
                // RU stores are in flight per lane instead of one (a store holds its data registers until it is sent; the
                // one-store loop spent ~12 k cycles per wave here waiting, which keeps the block's slot on the CU busy)
                constexpr int RU = 4;
                // position step as (kx, ky) increments: one conditional wrap per step instead of a divergent loop
                const int skx = (int)(((float)qstep + 0.5f) * inv_ky), sky = qstep - skx * KY;
                // (the ragged-nz variant is a separate copy of the loop: with both store forms in one body the compiler merges
                // them and splits every 16-byte store into a 12-byte and a 4-byte instruction)
                auto readout = [&](auto fast_c) {
                constexpr bool FAST = decltype(fast_c)::value != 0;
#pragma unroll 1
                for (int q = q0; q < npos; q += RU * qstep) {
                    float4 val[RU]; float* dst[RU];
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        const int qu = q + u * qstep;
                        const int i = ibase + 2 * P.mx * kx, j = jbase + P.my * ky;
                        kx += skx; ky += sky;
                        if (ky >= KY) { ky -= KY; ++kx; }
                        const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
#ifdef OLX_EXP_L2STORE
                        dst[u] = qu < npos ? base + ((long long)(io * P.ny + jo) * P.nz & 0xFFFFF) - (long long)(job >> 6) * P.vox : nullptr;  // A/B: stores stay cache resident
#else
                        dst[u] = qu < npos ? base + (long long)(io * P.ny + jo) * P.nz : nullptr;
#endif
                        const float* v = sv + (qu < npos ? qu : q) * RS;
                        val[u] = make_float4(v[0], v[1], v[2], v[3]);
                    }
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        if (!dst[u]) continue;
                        if constexpr (FAST) {
                            *reinterpret_cast<float4*>(dst[u]) = val[u];
                        } else {
                            const float vv[4] = {val[u].x, val[u].y, val[u].z, val[u].w};
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (kz + e < P.nz) dst[u][e] = vv[e];
                        }
                    }
                }
                };
                if (fast) readout(IntC<1>{}); else readout(IntC<0>{});
            }
            }
    };
#ifdef OLX_EXP_NOEPILOGUE
    {   // A/B build: keep every accumulator live, skip staging + stores
        float live = 0.f;
#pragma unroll
        for (int t = 0; t < COS_MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) live += acc[t][nt][0] + acc[t][nt][1] + acc[t][nt][2] + acc[t][nt][3];
        if (live != 123.456f) return;
    }
#endif
    stage_and_store(IntC<0>{});
    if constexpr (NT > GROUP) stage_and_store(IntC<GROUP>{});
    OLX_STAMP(7);
}

// pack for kernel 2c: element coordinates (wavelengths, padded) and B fragments in MFMA lane order.
// grid (n_el_pad/16, tiles, NT), block 64: thread = lane.  Column o (< 8*NT) of tile T carries the steering
// vector of its representative (focus, mirror image): W[perm[image][e], focus]; unused columns are zero.
__global__ void mfma_pack_k(const double* __restrict__ pos, const double* __restrict__ area, int n, int n_pad,
                            const double* __restrict__ delays, const double* __restrict__ apod,
                            const int* __restrict__ perm, double ox, double oy, double oz, double freq,
                            double w_scale /* P0/lambda * rev * S_W */, double rev, int n_foci,
                            const int* __restrict__ colinfo /*[tiles][32][2]: representative focus, mirror image (-1 = unused)*/,
                            const int* __restrict__ slot_elem /*kernel 2d: element of K slot s (-1 = virtual), NULL = identity*/,
                            int fp8corr /*kernel 2e, NT <= 2: the second fragment holds e4m3 [hi(k0), hi(k1), lo(k0), lo(k1)] per element*/,
                            float4* __restrict__ coords, uint4* __restrict__ bfrag) {
    const int ks = blockIdx.x, tile = blockIdx.y, nt = blockIdx.z, NT = gridDim.z, lane = threadIdx.x;
    if (!slot_elem && tile == 0 && nt == 0 && lane < 16) {
        const int e = 16 * ks + lane;
        coords[e] = (e < n) ? make_float4((float)((pos[e] - ox) * rev), (float)((pos[n + e] - oy) * rev),
                                          (float)((pos[2 * n + e] - oz) * rev), 0.f)
                            : make_float4(1.0e4f, 1.0e4f, 1.0e4f, 0.f);  // padding: far away, zero weight
    }
    const int g = lane >> 4, c = lane & 15, o = nt * 8 + (c >> 1), part_c = c & 1;
    const int col_focus = colinfo[((size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + o) * 2];
    const int col_mirror = colinfo[((size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + o) * 2 + 1];
    Half8Bits hi, lo;
#pragma unroll
    for (int jj = 0; jj < 8; ++jj) {
        const int k = 8 * g + jj, slot = 16 * ks + (k >> 1), part_k = k & 1;
        const int e = slot_elem ? slot_elem[slot] : (slot < n ? slot : -1);
        double val = 0.0;
        const int f = col_focus;
        if (e >= 0 && f >= 0 && f < n_foci) {
            const int es = perm[col_mirror * n + e];
            const size_t off = (size_t)f * n + es;
            const double cyc = freq * delays[off];
            const double ph = 6.283185307179586476925286766559 * (cyc - floor(cyc));
            const double w = apod[off] * area[es] * w_scale;
            const double wr = w * cos(ph), wi = w * sin(ph);
            val = part_k == 0 ? (part_c == 0 ? wr : wi) : (part_c == 0 ? -wi : wr);
        }
        const _Float16 h = (_Float16)(float)val;
        const _Float16 l = (_Float16)(float)(val - (double)(float)h);
        hi.h[jj] = h;
        lo.h[jj] = l;
    }
    if (fp8corr) {
        Half8Bits q;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            int w = __builtin_amdgcn_cvt_pk_fp8_f32((float)hi.h[2 * e] * COS_F8_HI, (float)hi.h[2 * e + 1] * COS_F8_HI, 0, false);
            w = __builtin_amdgcn_cvt_pk_fp8_f32((float)lo.h[2 * e] * COS_F8_LO, (float)lo.h[2 * e + 1] * COS_F8_LO, w, true);
            q.w[e] = (unsigned)w;
        }
        lo.u = q.u;
    }
    uint4* dst = bfrag + (((size_t)tile * (n_pad / 16) + ks) * NT + nt) * 128;
    dst[lane] = hi.u;
    dst[64 + lane] = lo.u;
}

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
struct HeteroParams {
    int n_planes;          // non-trivial planes
    float u0, v0;          // (table origin - grid x0) / hx, same for y: index-space offset of the table frame
    float inv_hx, inv_hy;  // 1 / spacing [1/wavelengths]
    int nxg, nyg;          // whole-grid lateral size of the medium planes
    int xg_begin;          // slab start (voxel i of the slab is grid column i + xg_begin)
};

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

// ------------------------------------------------------------------------------------
// aggregation over foci (plan/protocol.py:384-387) and per-focus scaling
// (plan/solution.py:331-337).  HBM-bound streaming: float4 per lane, grid-stride.
// ------------------------------------------------------------------------------------
__global__ void field_aggregate_k(const float* __restrict__ pmag, const float* __restrict__ inten,
                                  int n_foci, long long vox, float inv /* 1 / total foci (all ranks) */,
                                  float* __restrict__ pmax, float* __restrict__ imean) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {
        float m = 0.f, s = 0.f;
        for (int f = 0; f < n_foci; ++f) {
            if (pmag) m = fmaxf(m, pmag[(long long)f * vox + v]);
            if (inten) s += inten[(long long)f * vox + v];
        }
        if (pmax) pmax[v] = m;
        if (imean) imean[v] = s * inv;
    }
}

// Same aggregate from the |p| volumes alone: the intensity of a launched (not uploaded) result is scale(v) |p|^2 by
// construction (kwave_if.py:140-141), so the mean intensity is scale(v) mean_f |p_f|^2 and the intensity volumes
// need not be read back -- half the HBM traffic of field_aggregate_k (4 B per voxel and focus, float4 per lane).
__global__ __launch_bounds__(256) void field_aggregate_p_k(const float* __restrict__ pmag, int n_foci, long long vox, float inv,
                                                            float inten_scale, const float* __restrict__ inv2z,
                                                            float* __restrict__ pmax, float* __restrict__ imean) {
    const long long stride = (long long)gridDim.x * blockDim.x, v4 = vox >> 2;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < v4; q += stride) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), s = m;
#pragma unroll 8
        for (int f = 0; f < n_foci; ++f) {          // unrolled: up to 8 independent 16-byte loads in flight per lane
            const float4 p = reinterpret_cast<const float4*>(pmag + (long long)f * vox)[q];
            m.x = fmaxf(m.x, p.x); m.y = fmaxf(m.y, p.y); m.z = fmaxf(m.z, p.z); m.w = fmaxf(m.w, p.w);
            s.x = fmaf(p.x, p.x, s.x); s.y = fmaf(p.y, p.y, s.y); s.z = fmaf(p.z, p.z, s.z); s.w = fmaf(p.w, p.w, s.w);
        }
        reinterpret_cast<float4*>(pmax)[q] = m;
        if (imean) {
            float4 k = make_float4(inten_scale, inten_scale, inten_scale, inten_scale);
            if (inv2z) k = reinterpret_cast<const float4*>(inv2z)[q];
            reinterpret_cast<float4*>(imean)[q] = make_float4(s.x * k.x * inv, s.y * k.y * inv, s.z * k.z * inv, s.w * k.w * inv);
        }
    }
    for (long long v = (v4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {   // tail
        float m = 0.f, s = 0.f;
        for (int f = 0; f < n_foci; ++f) { const float p = pmag[(long long)f * vox + v]; m = fmaxf(m, p); s = fmaf(p, p, s); }
        pmax[v] = m;
        if (imean) imean[v] = s * (inv2z ? inv2z[v] : inten_scale) * inv;
    }
}

__global__ void field_scale_k(float* __restrict__ pmag, float* __restrict__ inten,
                              float* __restrict__ cplx, const float* __restrict__ scale,
                              long long vox) {
    const int f = blockIdx.y;
    const float s = scale[f];
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {
        const long long o = (long long)f * vox + v;
        if (pmag) pmag[o] *= s;
        if (inten) inten[o] *= s * s;
        if (cplx) { cplx[2 * o] *= s; cplx[2 * o + 1] *= s; }
    }
}

// ------------------------------------------------------------------------------------
// masked peak per focus (get_mask + max; plan/solution_analysis.py:384-442).  HBM-bound
// scan of one float per voxel; the focal-frame affine is evaluated in fp64 so that the
// mask edge matches the fp64 oracle.  Non-negative floats order like their bit patterns,
// so the cross-block reduction is an integer atomicMax.
// ------------------------------------------------------------------------------------
struct PeakParams {
    int nx, ny, nz;
    double ox, oy, oz, hx, hy, hz;  // slab voxel (0,0,0) position and spacing [m]
    double ia0, ia1, ia2;           // 1 / aspect
    double radius; int op; int use_zmin; double zmin;
    long long vox;
    long long vol_stride;           // vox for per-focus volumes, 0 when every focus mask scans ONE volume
};

__global__ __launch_bounds__(256) void field_masked_peak_k(const float* __restrict__ vol,
                                                            const double* __restrict__ A,
                                                            const PeakParams P,
                                                            unsigned* __restrict__ out) {
    const int f = blockIdx.y;
    __shared__ double sA[12];
    __shared__ float s_red[4];
    if (threadIdx.x < 12) sA[threadIdx.x] = A[f * 12 + threadIdx.x];
    __syncthreads();
    const float* v = vol + (long long)f * P.vol_stride;
    float m = 0.f;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long nyz = (long long)P.ny * P.nz;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P.vox; i += stride) {
        const int ix = (int)(i / nyz);
        const int rem = (int)(i - ix * nyz);
        const int iy = rem / P.nz, iz = rem - iy * P.nz;
        const double x = P.ox + ix * P.hx, y = P.oy + iy * P.hy, z = P.oz + iz * P.hz;
        bool sel = true;
        if (P.op != 4) {
            const double q0 = (sA[0] * x + sA[1] * y + sA[2] * z + sA[3]) * P.ia0;
            const double q1 = (sA[4] * x + sA[5] * y + sA[6] * z + sA[7]) * P.ia1;
            const double q2 = (sA[8] * x + sA[9] * y + sA[10] * z + sA[11]) * P.ia2;
            const double dist = sqrt(q0 * q0 + q1 * q1 + q2 * q2);
            sel = (P.op == 0) ? (dist < P.radius) : (P.op == 1) ? (dist <= P.radius)
                : (P.op == 2) ? (dist > P.radius) : (dist >= P.radius);
        }
        if (P.use_zmin) sel = sel && (z > P.zmin);
        if (sel) m = fmaxf(m, v[i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]));
        atomicMax(out + f, __float_as_uint(m));
    }
}

// ------------------------------------------------------------------------------------
// masked first moments per focus (find_centroid, plan/solution_analysis.py:306-317): over voxels inside
// the focal ellipsoid (dist < radius) whose |p| exceeds cutoff_f:  S0 = sum p, S1 = sum p * (x, y, z).
// fp64 sums, block-reduced, one atomicAdd(double) per block and component.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void field_masked_moments_k(const float* __restrict__ vol,
                                                               const double* __restrict__ A,
                                                               const float* __restrict__ cutoff,
                                                               const PeakParams P, double* __restrict__ out /*[F][4]*/) {
    const int f = blockIdx.y;
    __shared__ double sA[12];
    __shared__ double s_red[4][4];
    if (threadIdx.x < 12) sA[threadIdx.x] = A[f * 12 + threadIdx.x];
    __syncthreads();
    const float* v = vol + (long long)f * P.vol_stride;
    const float cut = cutoff[f];
    double s0 = 0, sx = 0, sy = 0, sz = 0;
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long nyz = (long long)P.ny * P.nz;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < P.vox; i += stride) {
        const int ix = (int)(i / nyz);
        const int rem = (int)(i - ix * nyz);
        const int iy = rem / P.nz, iz = rem - iy * P.nz;
        const double x = P.ox + ix * P.hx, y = P.oy + iy * P.hy, z = P.oz + iz * P.hz;
        const double q0 = (sA[0] * x + sA[1] * y + sA[2] * z + sA[3]) * P.ia0;
        const double q1 = (sA[4] * x + sA[5] * y + sA[6] * z + sA[7]) * P.ia1;
        const double q2 = (sA[8] * x + sA[9] * y + sA[10] * z + sA[11]) * P.ia2;
        const float p = v[i];
        if (sqrt(q0 * q0 + q1 * q1 + q2 * q2) < P.radius && p > cut) {
            s0 += p; sx += p * x; sy += p * y; sz += p * z;
        }
    }
    double comp[4] = {s0, sx, sy, sz};
#pragma unroll
    for (int k = 0; k < 4; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) comp[k] += __shfl_xor(comp[k], off, 64);
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = comp[k];
    }
    __syncthreads();
    if (threadIdx.x < 4) atomicAdd(out + 4 * f + threadIdx.x, s_red[0][threadIdx.x] + s_red[1][threadIdx.x] +
                                                                s_red[2][threadIdx.x] + s_red[3][threadIdx.x]);
}

// ------------------------------------------------------------------------------------
// trilinear samples of one resident volume at arbitrary points (interp_transformed_axis,
// plan/solution_analysis.py:444-486: xarray linear interpolation, NaN outside the grid).
// ------------------------------------------------------------------------------------
__global__ void field_sample_k(const float* __restrict__ vol, const double* __restrict__ pts, int npts,
                               const PeakParams P, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= npts) return;
    const double c[3] = {(pts[3 * i] - P.ox) / P.hx, (pts[3 * i + 1] - P.oy) / P.hy, (pts[3 * i + 2] - P.oz) / P.hz};
    const int n[3] = {P.nx, P.ny, P.nz};
    int i0[3]; double w[3];
    bool inside = true;
#pragma unroll
    for (int a = 0; a < 3; ++a) {
        const double tol = 1e-9 * (n[a] > 1 ? n[a] - 1 : 1);
        if (!(c[a] >= -tol && c[a] <= n[a] - 1 + tol)) inside = false;
        double cc = fmin(fmax(c[a], 0.0), (double)(n[a] - 1));
        i0[a] = (int)fmin(floor(cc), (double)max(n[a] - 2, 0));
        w[a] = cc - i0[a];
    }
    if (!inside) { out[i] = __builtin_nanf(""); return; }
    double acc = 0;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) {
                const int ix = min(i0[0] + dx, P.nx - 1), iy = min(i0[1] + dy, P.ny - 1), iz = min(i0[2] + dz, P.nz - 1);
                const double ww = (dx ? w[0] : 1 - w[0]) * (dy ? w[1] : 1 - w[1]) * (dz ? w[2] : 1 - w[2]);
                acc += ww * vol[((long long)ix * P.ny + iy) * P.nz + iz];
            }
    out[i] = (float)acc;
}

// ------------------------------------------------------------------------------------
// offset grid (get_gridded_transformed_coords / get_offset_grid / calc_dist_from_focus,
// plan/solution_analysis.py:344-403): per voxel q = A . [x, y, z, 1] in fp64 (A = first three rows of
// inv(get_focus_matrix)), optionally dist = sqrt(sum (q_a / aspect_a)^2).  Pure HBM write stream: 24 (+8) bytes
// per voxel, coordinates come from the three axis vectors (a few KB, cache resident).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void offset_grid_k(const double* __restrict__ xs, const double* __restrict__ ys,
                                                      const double* __restrict__ zs, int nx, int ny, int nz,
                                                      const double* __restrict__ A, double ia0, double ia1, double ia2,
                                                      double* __restrict__ coords, double* __restrict__ dist) {
    __shared__ double sA[12];
    if (threadIdx.x < 12) sA[threadIdx.x] = A[threadIdx.x];
    __syncthreads();
    const long long vox = (long long)nx * ny * nz, nyz = (long long)ny * nz;
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < vox; i += stride) {
        const int ix = (int)(i / nyz);
        const int rem = (int)(i - ix * nyz);
        const int iy = rem / nz, iz = rem - iy * nz;
        const double x = xs[ix], y = ys[iy], z = zs[iz];
        // same association as the reference's np.dot row: ((a0 x + a1 y) + a2 z) + a3, no fused contraction
        const double q0 = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(sA[0], x), __dmul_rn(sA[1], y)), __dmul_rn(sA[2], z)), sA[3]);
        const double q1 = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(sA[4], x), __dmul_rn(sA[5], y)), __dmul_rn(sA[6], z)), sA[7]);
        const double q2 = __dadd_rn(__dadd_rn(__dadd_rn(__dmul_rn(sA[8], x), __dmul_rn(sA[9], y)), __dmul_rn(sA[10], z)), sA[11]);
        if (coords) { coords[3 * i] = q0; coords[3 * i + 1] = q1; coords[3 * i + 2] = q2; }
        if (dist) {
            const double d0 = q0 * ia0, d1 = q1 * ia1, d2 = q2 * ia2;
            dist[i] = sqrt(d0 * d0 + d1 * d1 + d2 * d2);
        }
    }
}

// ------------------------------------------------------------------------------------
// time-of-flight spread (SimSetup.get_max_cycle_offset, sim/sim_setup.py:132-143): per voxel
// tof_e = ||r_v - r_e|| / c0 + delay_e, dtof = max_e tof - min_e tof; result = max over voxels (fp64, the reference's
// arithmetic).  Element positions / delays are wave-uniform scalar loads; block max via __shfl_xor, then one
// atomicMax on the bit pattern (non-negative doubles order like their uint64 images).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void tof_spread_k(const double* __restrict__ xs, const double* __restrict__ ys,
                                                     const double* __restrict__ zs, int nx, int ny, int nz,
                                                     const double* __restrict__ pos /*[3][N]*/, const double* __restrict__ delays,
                                                     int n, double c0, unsigned long long* __restrict__ out) {
    __shared__ double s_red[4];
    const long long vox = (long long)nx * ny * nz, nyz = (long long)ny * nz;
    const long long stride = (long long)gridDim.x * blockDim.x;
    double best = 0.0;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < vox; i += stride) {
        const int ix = (int)(i / nyz);
        const int rem = (int)(i - ix * nyz);
        const int iy = rem / nz, iz = rem - iy * nz;
        const double x = xs[ix], y = ys[iy], z = zs[iz];
        double tmax = -1.0e300, tmin = 1.0e300;
        for (int e = 0; e < n; ++e) {
            const double dx = x - pos[e], dy = y - pos[n + e], dz = z - pos[2 * n + e];
            const double t = __dadd_rn(__ddiv_rn(sqrt(__dadd_rn(__dadd_rn(__dmul_rn(dx, dx), __dmul_rn(dy, dy)), __dmul_rn(dz, dz))), c0),
                                       delays ? delays[e] : 0.0);
            tmax = fmax(tmax, t); tmin = fmin(tmin, t);
        }
        best = fmax(best, tmax - tmin);
    }
    best = wave_max(best);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = best;
    __syncthreads();
    if (threadIdx.x == 0) {
        best = fmax(fmax(s_red[0], s_red[1]), fmax(s_red[2], s_red[3]));
        atomicMax(out, (unsigned long long)__double_as_longlong(best));
    }
}

// weighted sum over foci into one volume: out[v] = sum_f w_f vol_f[v]  (get_ita, plan/solution.py:365-388)
__global__ void field_weighted_sum_k(const float* __restrict__ vol, const float* __restrict__ wts, int n_foci,
                                     long long vox, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    for (long long v = (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {
        float s = 0.f;
        for (int f = 0; f < n_foci; ++f) s += wts[f] * vol[(long long)f * vox + v];
        out[v] = s;
    }
}

}  // namespace olx
