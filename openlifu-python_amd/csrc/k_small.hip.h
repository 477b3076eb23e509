// Small kernels of the openlifu hot path, compiled into the host translation unit (olx.hip):
//  kernel 1  bf_solve_k / bf_quantize_k   per-element geometric delay / apodization solve (fp64), hardware hand-off
//            steer_pack_k, steer_pack_shared_k, mfma_pack_k   fp64 steering + element table -> kernel-2 operands
//  scans     field_aggregate(_p)_k, field_scale_k, field_masked_peak_k, field_masked_moments_k, field_sample_k,
//            offset_grid_k, tof_spread_k, field_weighted_sum_k   (HBM-bound streaming)
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4.
#pragma once
#include "k_types.hip.h"

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

__global__ void steer_pack_k(const double* __restrict__ pos, const double* __restrict__ area, int n,
                             const double* __restrict__ delays, const double* __restrict__ apod,
                             double ox, double oy, double oz, double freq, double p0_over_lambda,
                             double rev, const int* __restrict__ kfirst, const int* __restrict__ klast,
                             double hx_m, double hy_m, double hz_m /* spacing [m]: > 0 = SPLIT coordinates (kernel 2a) */,
                             float* __restrict__ tab) {
    const int f = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const size_t o = ((size_t)f * n + e);
    const double cyc = freq * delays[o];
    float* t = tab + o * TAB_STRIDE;
    t[3] = (float)(apod[o] * area[e] * p0_over_lambda * rev);
    t[4] = (float)(cyc - floor(cyc));
    if (hx_m > 0.0) {
        // kernel 2a: { voxel index nearest the element (an integer, exact in fp32), ... } and { offset from that voxel [wavelengths] }: the kernel forms
        // x_v - x_e = (i - i_e) h - f_e from an exact index difference instead of subtracting two rounded coordinates of ~ 10 wavelengths
        const double q[3] = {pos[e] - ox, pos[n + e] - oy, pos[2 * n + e] - oz}, h[3] = {hx_m, hy_m, hz_m};
        for (int a = 0; a < 3; ++a) {
            const double k = rint(q[a] / h[a]);
            t[a] = (float)k;
            t[5 + a] = (float)((q[a] - k * h[a]) * rev);
        }
        return;
    }
    t[0] = (float)((pos[e] - ox) * rev);
    t[1] = (float)((pos[n + e] - oy) * rev);
    t[2] = (float)((pos[2 * n + e] - oz) * rev);
    t[5] = kfirst ? __int_as_float(kfirst[e]) : 0.f;  // kernel 2h: planes strictly above / below the element
    t[6] = klast ? __int_as_float(klast[e]) : 0.f;
    t[7] = 0.f;
}

// pack for kernel 2b: complex weights W[sigma_m(e), f] = a P0 S / lambda * exp(j 2 pi frac(f0 tau)),
// evaluated in fp64 and rounded once.  perm[m][e] = index of the mirror image of element e.
__global__ void steer_pack_shared_k(const double* __restrict__ pos, const double* __restrict__ area, int n,
                                    const double* __restrict__ delays, const double* __restrict__ apod,
                                    const int* __restrict__ perm, double ox, double oy, double oz, double freq,
                                    double p0_over_lambda, double rev, int n_foci, int nf, int nm,
                                    double sx_m, double sy_m, double sz_m /* coordinate steps [m]: half a voxel on a folded axis, a voxel otherwise */,
                                    float* __restrict__ tab) {
    const int tile = blockIdx.y;
    const int e = blockIdx.x * blockDim.x + threadIdx.x;
    if (e >= n) return;
    const int nout = nf * nm, stride = SH_HEAD + 2 * nout;
    float* t = tab + ((size_t)tile * n + e) * stride;
    {   // the element as (index of the nearest coordinate step -- exact in fp32 --, offset from it [wavelengths]) per axis
        const double q[3] = {pos[e] - ox, pos[n + e] - oy, pos[2 * n + e] - oz}, st[3] = {sx_m, sy_m, sz_m};
        for (int a = 0; a < 3; ++a) {
            if (sx_m > 0.0) { const double k = rint(q[a] / st[a]); t[a] = (float)k; t[4 + a] = (float)((q[a] - k * st[a]) * rev); }
            else { t[a] = (float)(q[a] * rev); t[4 + a] = 0.f; }      // (absolute coordinates [wavelengths]: no voxel within a wavelength of an element)
        }
        t[3] = 0.f; t[7] = 0.f;
    }
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
        t[SH_HEAD + 2 * k] = wr;
        t[SH_HEAD + 1 + 2 * k] = wi;
    }
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
                            int fp8corr /*1 = kernel 2e / 2g, NT <= 2: the second fragment holds e4m3 [hi(k0), hi(k1), lo(k0), lo(k1)] per element*/,
                            int split_reim /*1 = kernel 2g (NT = 2): matrix column c of column tile 0 is Re, of column tile 1 Im of steering column c --
                                             a lane's accumulators then hold both parts of its voxels; 0 = (Re, Im) in adjacent matrix columns*/,
                            double sx_m, double sy_m, double sz_m /* kernel 2c: coordinate steps [m] (half a voxel on a folded axis, a voxel otherwise) */,
                            float4* __restrict__ coords /*[n_pad][2], kernel 2c*/, uint4* __restrict__ bfrag) {
    const int ks = blockIdx.x, tile = blockIdx.y, nt = blockIdx.z, NT = gridDim.z, lane = threadIdx.x;
    if (!slot_elem && tile == 0 && nt == 0 && lane < 16) {
        // the element as (index of the nearest coordinate step -- exact in fp32 --, offset from it [wavelengths]) per axis: field_mfma_k
        const int e = 16 * ks + lane;
        float kf[3] = {1.0e5f, 1.0e5f, 1.0e5f}, ff[3] = {0.f, 0.f, 0.f};      // padding: far away (steps or wavelengths), zero weight
        if (e < n) {
            const double q[3] = {pos[e] - ox, pos[n + e] - oy, pos[2 * n + e] - oz}, st[3] = {sx_m, sy_m, sz_m};
            for (int a = 0; a < 3; ++a) {
                if (sx_m > 0.0) { const double k = rint(q[a] / st[a]); kf[a] = (float)k; ff[a] = (float)((q[a] - k * st[a]) * rev); }
                else { kf[a] = (float)(q[a] * rev); ff[a] = 0.f; }      // (absolute coordinates: no voxel within a wavelength of an element)
            }
        }
        coords[2 * e] = make_float4(kf[0], kf[1], kf[2], 0.f);
        coords[2 * e + 1] = make_float4(ff[0], ff[1], ff[2], 0.f);
    }
    const int g = lane >> 4, c = lane & 15, o = split_reim ? c : nt * 8 + (c >> 1), part_c = split_reim ? nt : c & 1;
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
// aggregation over foci (plan/protocol.py:384-387) and per-focus scaling
// (plan/solution.py:331-337).  HBM-bound streaming: float4 per lane, grid-stride.
// ------------------------------------------------------------------------------------
// The streaming scans touch every byte of the result volumes once per pass: non-temporal 16-byte loads / stores (the volumes are far larger
// than L2 + Infinity Cache, and a line kept for them evicts one somebody will read) -- aggregate 64 -> 68 %, scale 66 -> 69 % of 8 TB/s.
typedef float olx_f4_t __attribute__((ext_vector_type(4)));
// Minimum waves per SIMD of the register-heavy scans (round 6, same box, alternating): field_masked_peak_k at 4 (<= 128 registers instead of 132:
// 4 instead of 3 waves per SIMD keep more loads in flight) 126.6 -> 116 us = 53 -> 58 % of 8 TB/s; at 5 it spills (162 us).  The six-peak scan and
// the fused post-pass LOSE with the same setting (201 -> 242 us, 470 -> 756 us: their exact-test branches spill) and keep the compiler's choice.
#define OLX_SCAN_WPE_FIELD_MASKED_PEAK_K 4
#define OLX_SCAN_WPE_FIELD_ANALYSIS_PEAKS4_K 1
#define OLX_SCAN_WPE_FIELD_SCALE_AGG_ANALYZE_K 1
__device__ __forceinline__ float4 ld4s(const float4* p) {
    const olx_f4_t v = __builtin_nontemporal_load(reinterpret_cast<const olx_f4_t*>(p));
    return make_float4(v.x, v.y, v.z, v.w);
}
__device__ __forceinline__ void st4s(float4* p, const float4 v) {
    __builtin_nontemporal_store(olx_f4_t{v.x, v.y, v.z, v.w}, reinterpret_cast<olx_f4_t*>(p));
}
// (the one-pass scale + aggregate + analyze kernel: non-temporal LOADS only -- 62.4 -> 64.4 %; with non-temporal stores it drops to 30 %)
#define OLX_SAA_LD(p) ld4s(p)
#define OLX_SAA_ST(p, v) (*(p) = (v))
// row quads: four consecutive z voxels of ONE row starting at a dword-aligned address, the last quad of a row possibly partial (nz % 4 != 0: every grid of
// the reference's SimSetup has odd voxel counts) -- the 16-byte form wherever the quad is whole, element by element (zero-filled) at a row's end
__device__ __forceinline__ float4 ld4u(const float* p, const int cnt) {
    if (cnt == 4) { const olx::floatx4u_t v = __builtin_nontemporal_load(reinterpret_cast<const olx::floatx4u_t*>(p)); return make_float4(v.x, v.y, v.z, v.w); }
    float4 r = make_float4(p[0], 0.f, 0.f, 0.f);
    if (cnt > 1) r.y = p[1];
    if (cnt > 2) r.z = p[2];
    return r;
}
__device__ __forceinline__ void st4u(float* p, const float4 v, const int cnt) {
    if (cnt == 4) { *reinterpret_cast<olx::floatx4u_t*>(p) = olx::floatx4u_t{v.x, v.y, v.z, v.w}; return; }
    p[0] = v.x;
    if (cnt > 1) p[1] = v.y;
    if (cnt > 2) p[2] = v.z;
}
#define OLX_LD4(p) __builtin_nontemporal_load(p)
#define OLX_ST4(p, v) __builtin_nontemporal_store(v, p)
__global__ __launch_bounds__(256) void field_aggregate_k(const float* __restrict__ pmag, const float* __restrict__ inten,
                                  int n_foci, long long vox, float inv /* 1 / total foci (all ranks) */,
                                  float* __restrict__ pmax, float* __restrict__ imean) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    // 16-byte main loop (every focus volume starts 16-byte aligned when vox % 4 == 0), up to 8 foci = 16 independent loads in
    // flight per lane; the max / sum run over f in the same order as the scalar tail, so both forms give the same bits
    const long long v4 = (vox & 3) == 0 ? (vox >> 2) : 0;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < v4; q += stride) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), sm = m;
#pragma unroll 8
        for (int f = 0; f < n_foci; ++f) {
            if (pmag) { const olx_f4_t p = OLX_LD4(reinterpret_cast<const olx_f4_t*>(pmag + (long long)f * vox) + q); m.x = fmaxf(m.x, p.x); m.y = fmaxf(m.y, p.y); m.z = fmaxf(m.z, p.z); m.w = fmaxf(m.w, p.w); }
            if (inten) { const olx_f4_t w = OLX_LD4(reinterpret_cast<const olx_f4_t*>(inten + (long long)f * vox) + q); sm.x += w.x; sm.y += w.y; sm.z += w.z; sm.w += w.w; }
        }
        if (pmax) st4s(reinterpret_cast<float4*>(pmax) + q, m);
        if (imean) st4s(reinterpret_cast<float4*>(imean) + q, make_float4(sm.x * inv, sm.y * inv, sm.z * inv, sm.w * inv));
    }
    for (long long v = (v4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {
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
            const float4 p = ld4s(reinterpret_cast<const float4*>(pmag + (long long)f * vox) + q);
            m.x = fmaxf(m.x, p.x); m.y = fmaxf(m.y, p.y); m.z = fmaxf(m.z, p.z); m.w = fmaxf(m.w, p.w);
            s.x = fmaf(p.x, p.x, s.x); s.y = fmaf(p.y, p.y, s.y); s.z = fmaf(p.z, p.z, s.z); s.w = fmaf(p.w, p.w, s.w);
        }
        st4s(reinterpret_cast<float4*>(pmax) + q, m);
        if (imean) {
            float4 k = make_float4(inten_scale, inten_scale, inten_scale, inten_scale);
            if (inv2z) k = ld4s(reinterpret_cast<const float4*>(inv2z) + q);
            st4s(reinterpret_cast<float4*>(imean) + q, make_float4(s.x * k.x * inv, s.y * k.y * inv, s.z * k.z * inv, s.w * k.w * inv));
        }
    }
    for (long long v = (v4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {   // tail
        float m = 0.f, s = 0.f;
        for (int f = 0; f < n_foci; ++f) { const float p = pmag[(long long)f * vox + v]; m = fmaxf(m, p); s = fmaf(p, p, s); }
        pmax[v] = m;
        if (imean) imean[v] = s * (inv2z ? inv2z[v] : inten_scale) * inv;
    }
}

// Solution.scale followed by the aggregation over foci (plan/solution.py:331-337, plan/protocol.py:382-387) in ONE pass:
// p_f *= s_f, I_f *= s_f^2 written back in place, and max_f p_f / mean_f I_f of the SCALED values written beside them -- the
// same products, the same order of the max / sum over f as field_scale_k followed by field_aggregate_k (bit-identical), but
// the volumes cross HBM twice (read + write) instead of three times.  float4 per lane; vox % 4 == 0 (host-checked).
__global__ __launch_bounds__(256) void field_scale_aggregate_k(float* __restrict__ pmag, float* __restrict__ inten, const float* __restrict__ scale,
                                                                int n_foci, long long vox, float inv, float* __restrict__ pmax, float* __restrict__ imean) {
    const long long stride = (long long)gridDim.x * blockDim.x, v4 = vox >> 2;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < v4; q += stride) {
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), sm = m;
#pragma unroll 4
        for (int f = 0; f < n_foci; ++f) {
#pragma clang fp contract(off)      // the sums take the ROUNDED products that are stored (what field_aggregate_k reads back), never an fma
            const float s = scale[f], s2 = s * s;
            float4* pp = reinterpret_cast<float4*>(pmag + (long long)f * vox) + q;
            float4* ip = reinterpret_cast<float4*>(inten + (long long)f * vox) + q;
            float4 p = ld4s(pp), w = ld4s(ip);
            p.x *= s; p.y *= s; p.z *= s; p.w *= s;
            w.x *= s2; w.y *= s2; w.z *= s2; w.w *= s2;
            st4s(pp, p); st4s(ip, w);
            m.x = fmaxf(m.x, p.x); m.y = fmaxf(m.y, p.y); m.z = fmaxf(m.z, p.z); m.w = fmaxf(m.w, p.w);
            sm.x += w.x; sm.y += w.y; sm.z += w.z; sm.w += w.w;
        }
        st4s(reinterpret_cast<float4*>(pmax) + q, m);
        st4s(reinterpret_cast<float4*>(imean) + q, make_float4(sm.x * inv, sm.y * inv, sm.z * inv, sm.w * inv));
    }
}

__global__ __launch_bounds__(256) void field_scale_k(float* __restrict__ pmag, float* __restrict__ inten,
                              float* __restrict__ cplx, const float* __restrict__ scale,
                              long long vox) {
    const int f = blockIdx.y;
    const float s = scale[f];
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long v4 = (vox & 3) == 0 ? (vox >> 2) : 0;          // 16-byte main loop, scalar tail (same products)
    const float s2 = s * s;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < v4; q += stride) {
        if (pmag) { olx_f4_t* p = reinterpret_cast<olx_f4_t*>(pmag + (long long)f * vox) + q; olx_f4_t v = OLX_LD4(p); v *= s; OLX_ST4(p, v); }
        if (inten) { olx_f4_t* p = reinterpret_cast<olx_f4_t*>(inten + (long long)f * vox) + q; olx_f4_t v = OLX_LD4(p); v *= s2; OLX_ST4(p, v); }
        if (cplx) {
            float4* p = reinterpret_cast<float4*>(cplx + 2 * (long long)f * vox) + 2 * q;
            float4 a = p[0], b = p[1];
            a.x *= s; a.y *= s; a.z *= s; a.w *= s; b.x *= s; b.y *= s; b.z *= s; b.w *= s;
            p[0] = a; p[1] = b;
        }
    }
    for (long long v = (v4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {
        const long long o = (long long)f * vox + v;
        if (pmag) pmag[o] *= s;
        if (inten) inten[o] *= s2;
        if (cplx) { cplx[2 * o] *= s; cplx[2 * o + 1] *= s; }
    }
}

// ------------------------------------------------------------------------------------
// Two helpers that keep the fp64 mask tests of the scans cheap without changing a single decision.
//  * quad_decode: (ix, iy, first z) of quad iq without integer divisions (there is no hardware integer divide): float
//    reciprocal estimate, corrected by one step -- exact for every iq < 2^31.
//  * mask_cmp: dist OP radius with dist = sqrt(d2).  sqrt is monotonic and correctly rounded, so outside a guard band of
//    1e-12 r^2 around r^2 the comparison of the squares decides; only inside the band (a voxel within ~1e-12 of the mask
//    surface) is the square root taken -- the same answer as taking it always, for ~10 fp64 instructions less per voxel.
// ------------------------------------------------------------------------------------
__device__ __forceinline__ void quad_decode(int iq, int nzq, int nyzq, float inv_nyzq, float inv_nzq, int& ix, int& iy, int& iz0) {
    ix = (int)((float)iq * inv_nyzq);
    int rem = iq - ix * nyzq;
    if (rem < 0) { --ix; rem += nyzq; } else if (rem >= nyzq) { ++ix; rem -= nyzq; }
    iy = (int)((float)rem * inv_nzq);
    int r2 = rem - iy * nzq;
    if (r2 < 0) { --iy; r2 += nzq; } else if (r2 >= nzq) { ++iy; r2 -= nzq; }
    iz0 = r2 << 2;
}

template <int OP>   // 0 '<', 1 '<=', 2 '>', 3 '>='
__device__ __forceinline__ bool mask_cmp(double d2, double r, double r2lo, double r2hi) {
    if (d2 < r2lo) return OP <= 1;
    if (d2 > r2hi) return OP >= 2;
    const double dist = sqrt(d2);
    return OP == 0 ? dist < r : OP == 1 ? dist <= r : OP == 2 ? dist > r : dist >= r;
}

// fp32 pre-test of the same masks.  fp64 runs at half rate and the focal-frame test is ~25 fp64 instructions per voxel: the scans were
// bound by it (29 - 43 % of the HBM roofline).  One thread per block prepares an fp32 copy of the frame (1 / aspect folded in)
// and a band [r_in^2, r_out^2] around the mask surface that is wider than anything the fp32 evaluation can be off by
// (32 roundings of relative size 2^-24 on terms bounded by max_a sum_k |A_ak| |c_k|_max); a voxel whose fp32 squared distance
// lies outside the band is decided there, a voxel inside it (within ~1e-6 m of the surface) takes the exact fp64 test -- every
// decision is the fp64 one.  z > zmin becomes an integer plane test (first plane with oz + k hz > zmin, found with that
// very expression: it is monotonic in k).
struct MaskFast {
    float a[12];            // rows of diag(1 / aspect) . A
    float rin2[2], rout2[2];// bands of the two radii (main '<', side '>'); rin2 < 0: no voxel is surely inside
    float ox, oy, oz, hx, hy, hz;
    int iz_first;           // planes iz >= iz_first satisfy z > zmin (INT_MAX: none; 0: all)
};
__device__ inline void mask_fast_prepare(MaskFast& M, const double* sA, const PeakParams& P, double r_main, double r_side, bool use_zmin) {
    const double ia[3] = {P.ia0, P.ia1, P.ia2};
    const double cm[3] = {fmax(fabs(P.ox), fabs(P.ox + (P.nx - 1) * P.hx)), fmax(fabs(P.oy), fabs(P.oy + (P.ny - 1) * P.hy)),
                          fmax(fabs(P.oz), fabs(P.oz + (P.nz - 1) * P.hz))};
    double bound = 0;
    for (int a = 0; a < 3; ++a) {
        double mag = fabs(sA[4 * a + 3]);
        for (int k = 0; k < 3; ++k) mag += fabs(sA[4 * a + k]) * cm[k];
        bound = fmax(bound, mag * fabs(ia[a]));
        for (int k = 0; k < 4; ++k) M.a[4 * a + k] = (float)(sA[4 * a + k] * ia[a]);
    }
    const double band = 2.0 * 32.0 * 5.9604645e-8 * bound;      // > sqrt(3) x the per-component error
    const double rr[2] = {r_main, r_side};
    for (int q = 0; q < 2; ++q) {
        const double rin = rr[q] - band, rout = rr[q] + band;
        M.rin2[q] = rin > 0 ? (float)(rin * rin * (1.0 - 1e-5)) : -1.f;
        M.rout2[q] = (float)(rout * rout * (1.0 + 1e-5)) * 1.000001f + 1e-37f;
    }
    M.ox = (float)P.ox; M.oy = (float)P.oy; M.oz = (float)P.oz; M.hx = (float)P.hx; M.hy = (float)P.hy; M.hz = (float)P.hz;
    int k0 = 0;
    if (use_zmin) {
        double est = floor((P.zmin - P.oz) / P.hz);
        est = fmin(fmax(est, -2.0), (double)P.nz + 1.0);
        int k = (int)est;
        while (k >= 0 && P.oz + k * P.hz > P.zmin) --k;
        while (k + 1 < P.nz + 2 && P.oz + (k + 1) * P.hz <= P.zmin) ++k;
        k0 = k + 1 < 0 ? 0 : k + 1;
    }
    M.iz_first = k0;
}
// 1 = surely selected side "inside" (d < r), -1 = surely outside (d > r), 0 = inside the band: take the exact test
__device__ __forceinline__ int mask_fast_side(float d2, float rin2, float rout2) { return d2 < rin2 ? 1 : (d2 > rout2 ? -1 : 0); }

// ------------------------------------------------------------------------------------
// masked peak per focus (get_mask + max; plan/solution_analysis.py:384-442).  HBM-bound
// scan of one float per voxel; the focal-frame affine is evaluated in fp64 so that the
// mask edge matches the fp64 oracle.  Non-negative floats order like their bit patterns,
// so the cross-block reduction is an integer atomicMax.
// ------------------------------------------------------------------------------------

__global__ __launch_bounds__(256, OLX_SCAN_WPE_FIELD_MASKED_PEAK_K) void field_masked_peak_k(const float* __restrict__ vol,
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
    // 16-byte main loop: a lane owns four consecutive z voxels (quads never straddle a z row), the index decode is paid once per
    // quad; the per-voxel test is the one of the scalar loop below (which serves grids with nz % 4 != 0)
    const bool quads = (P.nz & 3) == 0 && (P.vol_stride & 3) == 0 && P.vox < (1ll << 33);
    if (quads) {
        const int nzq = P.nz >> 2, nyzq = P.ny * nzq, nq = (int)(P.vox >> 2);
        const float inv_nyzq = 1.0f / (float)nyzq, inv_nzq = 1.0f / (float)nzq;
        const double r2 = P.radius * P.radius, r2lo = r2 * (1.0 - 1e-12), r2hi = r2 * (1.0 + 1e-12);
        __shared__ MaskFast sM;
        if (threadIdx.x == 0) mask_fast_prepare(sM, sA, P, P.radius, P.radius, P.use_zmin != 0);
        __syncthreads();
        const MaskFast M = sM;
        const bool want_in = P.op <= 1;              // '<' / '<=' select the inside, '>' / '>=' the outside (op 4: no distance test)
        constexpr int UQ = 4;                        // quads per lane and iteration: their loads are issued before any of the arithmetic
        for (int ib = blockIdx.x * blockDim.x + threadIdx.x; ib < nq; ib += UQ * (int)stride) {
          float4 pq[UQ];
#pragma unroll
          for (int u = 0; u < UQ; ++u) {
              const long long iu = (long long)ib + (long long)u * stride;
              pq[u] = iu < nq ? ld4s(reinterpret_cast<const float4*>(v) + iu) : make_float4(0.f, 0.f, 0.f, 0.f);
          }
#pragma unroll
          for (int u = 0; u < UQ; ++u) {
            const int iq = (int)min((long long)ib + (long long)u * stride, (long long)nq - 1);      // (past the end: a zero quad, harmless under max)
            int ix, iy, iz0;
            quad_decode(iq, nzq, nyzq, inv_nyzq, inv_nzq, ix, iy, iz0);
            const float4 p4 = pq[u];
            const float pv[4] = {p4.x, p4.y, p4.z, p4.w};
            const float fx = fmaf((float)ix, M.hx, M.ox), fy = fmaf((float)iy, M.hy, M.oy);
            const float b0 = fmaf(M.a[1], fy, fmaf(M.a[0], fx, M.a[3])), b1 = fmaf(M.a[5], fy, fmaf(M.a[4], fx, M.a[7])), b2 = fmaf(M.a[9], fy, fmaf(M.a[8], fx, M.a[11]));
            // branch-free fp32 pass over the quad's four voxels; the (rare) voxels inside the band are settled by ONE exact pass behind it
            int side[4], undecided = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float fz = fmaf((float)(iz0 + e), M.hz, M.oz);
                const float g0 = fmaf(M.a[2], fz, b0), g1 = fmaf(M.a[6], fz, b1), g2 = fmaf(M.a[10], fz, b2);
                side[e] = mask_fast_side(fmaf(g2, g2, fmaf(g1, g1, g0 * g0)), M.rin2[0], M.rout2[0]);
                undecided |= side[e] == 0 ? (1 << e) : 0;
            }
            if (P.op != 4 && undecided) {
                const double x = P.ox + ix * P.hx, y = P.oy + iy * P.hy;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!((undecided >> e) & 1)) continue;
                    const double z = P.oz + (iz0 + e) * P.hz;
                    const double q0 = (sA[0] * x + sA[1] * y + sA[2] * z + sA[3]) * P.ia0;
                    const double q1 = (sA[4] * x + sA[5] * y + sA[6] * z + sA[7]) * P.ia1;
                    const double q2 = (sA[8] * x + sA[9] * y + sA[10] * z + sA[11]) * P.ia2;
                    const double d2 = q0 * q0 + q1 * q1 + q2 * q2;
                    // "inside" in the sense the op needs: '<' selects dist < r, '>=' its complement; '<=' selects dist <= r, '>' its complement
                    const bool in = (P.op == 0 || P.op == 3) ? mask_cmp<0>(d2, P.radius, r2lo, r2hi) : mask_cmp<1>(d2, P.radius, r2lo, r2hi);
                    side[e] = in ? 1 : -1;
                }
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                bool sel = P.op == 4 ? true : ((side[e] > 0) == want_in);
                if (P.use_zmin) sel = sel && (iz0 + e) >= M.iz_first;
                if (sel) m = fmaxf(m, pv[e]);
            }
          }
        }
    } else
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
// The six masked peaks Solution.analyze reads off |p_f| and intensity_f (plan/solution.py:205-262) in ONE pass over the two
// volumes: mainlobe (dist < r_main), sidelobe (dist > r_side, z > zmin) and global (z > zmin) maxima of each.  Same fp64
// focal-frame arithmetic and comparisons as field_masked_peak_k (ops 0, 2, 4), so the six numbers are bit-identical to six
// separate scans; out[f][6] = (main p, main I, side p, side I, global p, global I), integer atomicMax of the float bits.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void field_analysis_peaks_k(const float* __restrict__ pmag, const float* __restrict__ inten,
                                                               const double* __restrict__ A, const PeakParams P /*radius = r_main*/,
                                                               const double r_side, unsigned* __restrict__ out /*[F][6]*/) {
    const int f = blockIdx.y;
    __shared__ double sA[12];
    __shared__ float s_red[4][6];
    if (threadIdx.x < 12) sA[threadIdx.x] = A[f * 12 + threadIdx.x];
    __syncthreads();
    const float* vp = pmag + (long long)f * P.vox;
    const float* vi = inten + (long long)f * P.vox;
    float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
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
        const double dist = sqrt(q0 * q0 + q1 * q1 + q2 * q2);
        const bool zok = z > P.zmin;
        const float p = vp[i], w = vi[i];
        if (dist < P.radius) { m[0] = fmaxf(m[0], p); m[1] = fmaxf(m[1], w); }
        if (zok && dist > r_side) { m[2] = fmaxf(m[2], p); m[3] = fmaxf(m[3], w); }
        if (zok) { m[4] = fmaxf(m[4], p); m[5] = fmaxf(m[5], w); }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m[k] = fmaxf(m[k], __shfl_xor(m[k], off, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const float r = fmaxf(fmaxf(s_red[0][k], s_red[1][k]), fmaxf(s_red[2][k], s_red[3][k]));
        atomicMax(out + f * 6 + k, __float_as_uint(r));
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
        if (coords) { coords[3 * i] = q0; coords[3 * i + 1] = q1; coords[3 * i + 2] = q2; }      // (8-byte stores at a 24-byte stride: non-temporal, they would not merge in L2 -- 63 -> 35 % of 8 TB/s, measured)
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

// "time-average" intensity volume of Solution.analyze: out[v] = max_f w_f vol_f[v].  The reference's get_ita (plan/solution.py:365-388) means a
// pulse-count-weighted average over the foci, but on its [focal_point_index, x, y, z] arrays its broadcast (`expand_dims(..., -1) * counts`,
// counts shaped [1, 1, 1, F]) returns every focus' OWN intensity times the two duty cycles, and analyze takes `.where(mask).max()` /
// `(ita * z_mask).max()` over that whole stack (plan/solution.py:243, 274) -- i.e. the maximum over foci AND voxels.  This volume is the
// maximum over foci, so that the masked peaks taken from it are the reference's numbers.  (The kernel keeps its round-2 name.)
__global__ __launch_bounds__(256) void field_weighted_sum_k(const float* __restrict__ vol, const float* __restrict__ wts, int n_foci,
                                     long long vox, float* __restrict__ out) {
    const long long stride = (long long)gridDim.x * blockDim.x;
    const long long v4 = (vox & 3) == 0 ? (vox >> 2) : 0;          // 16-byte main loop, scalar tail (same sums)
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < v4; q += stride) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int f = 0; f < n_foci; ++f) {
            const float w = wts[f];
            const float4 v = ld4s(reinterpret_cast<const float4*>(vol + (long long)f * vox) + q);
            s.x = fmaxf(s.x, w * v.x); s.y = fmaxf(s.y, w * v.y); s.z = fmaxf(s.z, w * v.z); s.w = fmaxf(s.w, w * v.w);
        }
        st4s(reinterpret_cast<float4*>(out) + q, s);
    }
    for (long long v = (v4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; v < vox; v += stride) {
        float s = 0.f;
        for (int f = 0; f < n_foci; ++f) s = fmaxf(s, wts[f] * vol[(long long)f * vox + v]);
        out[v] = s;
    }
}

// ------------------------------------------------------------------------------------
// The same six peaks with 16-byte loads: a thread owns FOUR consecutive z voxels (nz % 4 == 0), so the index decode (32-bit
// divisions: there is no hardware integer divide) is paid once per 32 bytes of traffic instead of once per 8.  The fp64 focal-frame
// expression and the comparisons per voxel are the ones of field_analysis_peaks_k, so the peaks are bit-identical.
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, OLX_SCAN_WPE_FIELD_ANALYSIS_PEAKS4_K) void field_analysis_peaks4_k(const float* __restrict__ pmag, const float* __restrict__ inten,
                                                                const double* __restrict__ A, const PeakParams P /*radius = r_main*/,
                                                                const double r_side, unsigned* __restrict__ out /*[F][6]*/) {
    const int f = blockIdx.y;
    __shared__ double sA[12];
    __shared__ float s_red[4][6];
    if (threadIdx.x < 12) sA[threadIdx.x] = A[f * 12 + threadIdx.x];
    __syncthreads();
    const float4* vp = reinterpret_cast<const float4*>(pmag + (long long)f * P.vox);
    const float4* vi = reinterpret_cast<const float4*>(inten + (long long)f * P.vox);
    float m[6] = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    const int nzq = P.nz >> 2, nyzq = P.ny * nzq, nq = (int)(P.vox >> 2);
    const int stride = gridDim.x * blockDim.x;
    const float inv_nyzq = 1.0f / (float)nyzq, inv_nzq = 1.0f / (float)nzq;
    const double rm2 = P.radius * P.radius, rm2lo = rm2 * (1.0 - 1e-12), rm2hi = rm2 * (1.0 + 1e-12);
    const double rs2 = r_side * r_side, rs2lo = rs2 * (1.0 - 1e-12), rs2hi = rs2 * (1.0 + 1e-12);
    __shared__ MaskFast sM;
    if (threadIdx.x == 0) mask_fast_prepare(sM, sA, P, P.radius, r_side, true);
    __syncthreads();
    const MaskFast M = sM;
    constexpr int UQ = 4;                            // quads per lane and iteration: 8 independent 16-byte loads before any arithmetic
    for (int ib = blockIdx.x * blockDim.x + threadIdx.x; ib < nq; ib += UQ * stride) {
      float4 pq[UQ], wq[UQ];
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
          const long long iu = (long long)ib + (long long)u * stride;
          const bool ok = iu < nq;
          pq[u] = ok ? ld4s(vp + iu) : make_float4(0.f, 0.f, 0.f, 0.f);
          wq[u] = ok ? ld4s(vi + iu) : make_float4(0.f, 0.f, 0.f, 0.f);
      }
#pragma unroll
      for (int u = 0; u < UQ; ++u) {
        const int iq = (int)min((long long)ib + (long long)u * stride, (long long)nq - 1);          // (past the end: zero quads, harmless under max)
        int ix, iy, iz0;
        quad_decode(iq, nzq, nyzq, inv_nyzq, inv_nzq, ix, iy, iz0);
        const float4 p4 = pq[u], w4 = wq[u];
        const float pv[4] = {p4.x, p4.y, p4.z, p4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w};
        const float fx = fmaf((float)ix, M.hx, M.ox), fy = fmaf((float)iy, M.hy, M.oy);
        const float b0 = fmaf(M.a[1], fy, fmaf(M.a[0], fx, M.a[3])), b1 = fmaf(M.a[5], fy, fmaf(M.a[4], fx, M.a[7])), b2 = fmaf(M.a[9], fy, fmaf(M.a[8], fx, M.a[11]));
        // branch-free fp32 pass over the quad's four voxels; voxels inside a band of either mask surface (rare) are settled by ONE exact pass
        int in_main[4], in_side[4], undecided = 0;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float fz = fmaf((float)(iz0 + e), M.hz, M.oz);
            const float g0 = fmaf(M.a[2], fz, b0), g1 = fmaf(M.a[6], fz, b1), g2 = fmaf(M.a[10], fz, b2);
            const float d2f = fmaf(g2, g2, fmaf(g1, g1, g0 * g0));
            in_main[e] = mask_fast_side(d2f, M.rin2[0], M.rout2[0]);
            in_side[e] = mask_fast_side(d2f, M.rin2[1], M.rout2[1]);
            undecided |= (in_main[e] == 0 || in_side[e] == 0) ? (1 << e) : 0;
        }
        if (undecided) {
            const double x = P.ox + ix * P.hx, y = P.oy + iy * P.hy;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                if (!((undecided >> e) & 1)) continue;
                const double z = P.oz + (iz0 + e) * P.hz;
                const double q0 = (sA[0] * x + sA[1] * y + sA[2] * z + sA[3]) * P.ia0;
                const double q1 = (sA[4] * x + sA[5] * y + sA[6] * z + sA[7]) * P.ia1;
                const double q2 = (sA[8] * x + sA[9] * y + sA[10] * z + sA[11]) * P.ia2;
                const double d2 = q0 * q0 + q1 * q1 + q2 * q2;
                in_main[e] = mask_cmp<0>(d2, P.radius, rm2lo, rm2hi) ? 1 : -1;
                in_side[e] = mask_cmp<2>(d2, r_side, rs2lo, rs2hi) ? -1 : 1;      // (-1 = outside the side radius = selected)
            }
        }
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const bool zok = (iz0 + e) >= M.iz_first;
            const float p = pv[e], w = wv[e];
            if (in_main[e] > 0) { m[0] = fmaxf(m[0], p); m[1] = fmaxf(m[1], w); }
            if (zok && in_side[e] < 0) { m[2] = fmaxf(m[2], p); m[3] = fmaxf(m[3], w); }
            if (zok) { m[4] = fmaxf(m[4], p); m[5] = fmaxf(m[5], w); }
        }
      }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) m[k] = fmaxf(m[k], __shfl_xor(m[k], off, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][k] = m[k];
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int k = threadIdx.x;
        const float r = fmaxf(fmaxf(s_red[0][k], s_red[1][k]), fmaxf(s_red[2][k], s_red[3][k]));
        atomicMax(out + f * 6 + k, __float_as_uint(r));
    }
}

// ------------------------------------------------------------------------------------
// Mainlobe masks are small: a 2.5 mm x 2.5 mm x 12.5 mm ellipsoid holds ~0.3 % of a 64 mm cube.  The host hands every focus
// the index box that encloses its ellipsoid (one voxel of margin); these two kernels evaluate the SAME per-voxel expressions as
// field_masked_peak_k / field_masked_moments_k (ops '<' and '<=') on the box only -- voxels outside it cannot be selected, so the
// peak is bit-identical and the moments sum the same terms.  box[f] = {x0, x1, y0, y1, z0, z1} (half-open, slab indices).
// ------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void field_masked_peak_box_k(const float* __restrict__ vol, const double* __restrict__ A, const PeakParams P,
                                                                const int* __restrict__ boxes, unsigned* __restrict__ out) {
    const int f = blockIdx.y;
    __shared__ double sA[12];
    __shared__ float s_red[4];
    if (threadIdx.x < 12) sA[threadIdx.x] = A[f * 12 + threadIdx.x];
    __syncthreads();
    const int* b = boxes + 6 * f;
    const int bx = b[1] - b[0], by = b[3] - b[2], bz = b[5] - b[4];
    const float* v = vol + (long long)f * P.vol_stride;
    float m = 0.f;
    const int total = (bx > 0 && by > 0 && bz > 0) ? bx * by * bz : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lx = i / (by * bz), rem = i - lx * (by * bz), ly = rem / bz, lz = rem - ly * bz;
        const int ix = b[0] + lx, iy = b[2] + ly, iz = b[4] + lz;
        const double x = P.ox + ix * P.hx, y = P.oy + iy * P.hy, z = P.oz + iz * P.hz;
        const double q0 = (sA[0] * x + sA[1] * y + sA[2] * z + sA[3]) * P.ia0;
        const double q1 = (sA[4] * x + sA[5] * y + sA[6] * z + sA[7]) * P.ia1;
        const double q2 = (sA[8] * x + sA[9] * y + sA[10] * z + sA[11]) * P.ia2;
        const double dist = sqrt(q0 * q0 + q1 * q1 + q2 * q2);
        bool sel = (P.op == 0) ? (dist < P.radius) : (dist <= P.radius);
        if (P.use_zmin) sel = sel && (z > P.zmin);
        if (sel) m = fmaxf(m, v[((long long)ix * P.ny + iy) * P.nz + iz]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(out + f, __float_as_uint(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]))));
}

__global__ __launch_bounds__(256) void field_masked_moments_box_k(const float* __restrict__ vol, const double* __restrict__ A,
                                                                   const float* __restrict__ cutoff, const PeakParams P,
                                                                   const int* __restrict__ boxes, double* __restrict__ out /*[F][4]*/) {
    const int f = blockIdx.y;
    __shared__ double sA[12];
    __shared__ double s_red[4][4];
    if (threadIdx.x < 12) sA[threadIdx.x] = A[f * 12 + threadIdx.x];
    __syncthreads();
    const int* b = boxes + 6 * f;
    const int bx = b[1] - b[0], by = b[3] - b[2], bz = b[5] - b[4];
    const float* v = vol + (long long)f * P.vol_stride;
    const float cut = cutoff[f];
    double s0 = 0, sx = 0, sy = 0, sz = 0;
    const int total = (bx > 0 && by > 0 && bz > 0) ? bx * by * bz : 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < total; i += gridDim.x * blockDim.x) {
        const int lx = i / (by * bz), rem = i - lx * (by * bz), ly = rem / bz, lz = rem - ly * bz;
        const int ix = b[0] + lx, iy = b[2] + ly, iz = b[4] + lz;
        const double x = P.ox + ix * P.hx, y = P.oy + iy * P.hy, z = P.oz + iz * P.hz;
        const double q0 = (sA[0] * x + sA[1] * y + sA[2] * z + sA[3]) * P.ia0;
        const double q1 = (sA[4] * x + sA[5] * y + sA[6] * z + sA[7]) * P.ia1;
        const double q2 = (sA[8] * x + sA[9] * y + sA[10] * z + sA[11]) * P.ia2;
        const float p = v[((long long)ix * P.ny + iy) * P.nz + iz];
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

// out[v] = max_f w_f I_f[v] (field_weighted_sum_k) and, in the same pass, the maximum of out over the voxels with z > zmin
// (field_masked_peak_k with op 4 on that volume): the global time-average intensity peak costs no second scan.
__global__ __launch_bounds__(256) void field_weighted_sum_peak_k(const float* __restrict__ vol, const float* __restrict__ wts, int n_foci,
                                                                  const PeakParams P, float* __restrict__ out, unsigned* __restrict__ peak) {
    __shared__ float s_red[4];
    const long long stride = (long long)gridDim.x * blockDim.x;
    float m = 0.f;
    const long long v4 = ((P.vox & 3) == 0 && (P.nz & 3) == 0 && P.vox < (1ll << 33)) ? (P.vox >> 2) : 0;   // quads never straddle a z row
    const int nzq = P.nz >> 2;
    for (long long q = (long long)blockIdx.x * blockDim.x + threadIdx.x; q < v4; q += stride) {
        float4 s = make_float4(0.f, 0.f, 0.f, 0.f);
#pragma unroll 8
        for (int f = 0; f < n_foci; ++f) {
            const float w = wts[f];
            const float4 v = ld4s(reinterpret_cast<const float4*>(vol + (long long)f * P.vox) + q);
            s.x = fmaxf(s.x, w * v.x); s.y = fmaxf(s.y, w * v.y); s.z = fmaxf(s.z, w * v.z); s.w = fmaxf(s.w, w * v.w);
        }
        st4s(reinterpret_cast<float4*>(out) + q, s);
        const int iz0 = (int)((unsigned)q % (unsigned)nzq) << 2;
        const float sv[4] = {s.x, s.y, s.z, s.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const double z = P.oz + (iz0 + e) * P.hz;
            if (z > P.zmin) m = fmaxf(m, sv[e]);
        }
    }
    for (long long v = (v4 << 2) + (long long)blockIdx.x * blockDim.x + threadIdx.x; v < P.vox; v += stride) {
        float s = 0.f;
        for (int f = 0; f < n_foci; ++f) s = fmaxf(s, wts[f] * vol[(long long)f * P.vox + v]);
        out[v] = s;
        const int iz = P.vox < (1ll << 31) ? (int)((unsigned)v % (unsigned)P.nz) : (int)(v % P.nz);
        const double z = P.oz + iz * P.hz;
        if (z > P.zmin) m = fmaxf(m, s);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) m = fmaxf(m, __shfl_xor(m, off, 64));
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = m;
    __syncthreads();
    if (threadIdx.x == 0) atomicMax(peak, __float_as_uint(fmaxf(fmaxf(s_red[0], s_red[1]), fmaxf(s_red[2], s_red[3]))));
}

// ------------------------------------------------------------------------------------
// Protocol.calc_solution(scale=True) in ONE pass over the focus volumes: Solution.scale (p_f *= s_f, I_f *= s_f^2, in place), the
// aggregation over foci (max |p|, mean intensity), the time-average intensity volume sum_f w_f I_f with its global peak above zmin,
// and the six masked peaks per focus of Solution.analyze -- each the arithmetic of its own kernel (field_scale_aggregate_k,
// field_weighted_sum_peak_k, field_analysis_peaks4_k: same products, same order over f, same mask decisions), but the 1.07 GB of
// volumes cross HBM twice (read, write back) instead of five times.  Voxel-major: a lane owns a quad of z voxels of ALL (<= 8) foci -- round 6:
// a ROW quad (ld4u / st4u), so rows of any length take this pass (every grid of the reference's SimSetup has odd voxel counts; until then such grids
// fell back to five scalar passes: calc_solution 2.9 instead of 1.7 ms at 241 x 241 x 257 x 8 foci).
// ------------------------------------------------------------------------------------
constexpr int SAA_MAXF = 8;
template <bool ROWQ /*rows of any length: row quads with a partial last quad (the 16-byte aligned form measured 470 us against 790 us with the general one)*/>
__global__ __launch_bounds__(256, OLX_SCAN_WPE_FIELD_SCALE_AGG_ANALYZE_K) void field_scale_agg_analyze_k(float* __restrict__ pmag, float* __restrict__ inten, const float* __restrict__ scale,
                                                                  const float* __restrict__ wts, const double* __restrict__ A, int n_foci,
                                                                  const PeakParams P /*radius = r_main*/, double r_side, float inv_n,
                                                                  float* __restrict__ pmax, float* __restrict__ imean, float* __restrict__ wint,
                                                                  unsigned* __restrict__ peaks /*[F][6]*/, unsigned* __restrict__ wpeak) {
    __shared__ double sA[SAA_MAXF][12];
    __shared__ MaskFast sM[SAA_MAXF];
    __shared__ float s_red[4][SAA_MAXF * 6 + 1];
    for (int q = threadIdx.x; q < 12 * n_foci; q += blockDim.x) sA[q / 12][q % 12] = A[q];
    __syncthreads();
    if ((int)threadIdx.x < n_foci) mask_fast_prepare(sM[threadIdx.x], sA[threadIdx.x], P, P.radius, r_side, true);
    __syncthreads();
    const int nzq = P.nz >> 2, nyzq = P.ny * nzq, nq = P.nx * nyzq;            // WHOLE quads of the rows (host: nx ny ceil(nz / 4) < 2^31)
    const float inv_nyzq = 1.0f / (float)nyzq, inv_nzq = 1.0f / (float)nzq;
    const double rm2 = P.radius * P.radius, rm2lo = rm2 * (1.0 - 1e-12), rm2hi = rm2 * (1.0 + 1e-12);
    const double rs2 = r_side * r_side, rs2lo = rs2 * (1.0 - 1e-12), rs2hi = rs2 * (1.0 + 1e-12);
    const int iz_first = sM[0].iz_first;                 // (zmin and the grid are the same for every focus)
    const float ox = sM[0].ox, oy = sM[0].oy, oz = sM[0].oz, hx = sM[0].hx, hy = sM[0].hy, hz = sM[0].hz;
    float pk[SAA_MAXF][6];
#pragma unroll
    for (int f = 0; f < SAA_MAXF; ++f)
#pragma unroll
        for (int k = 0; k < 6; ++k) pk[f][k] = 0.f;
    float wmax = 0.f;
    const int stride = gridDim.x * blockDim.x;
    // one quad of z voxels of every focus; WHOLE: four voxels (the 16-byte forms), else the 1 - 3 voxels behind a row's last whole quad
    auto quad = [&](const int ix, const int iy, const int iz0, auto whole_c) __attribute__((always_inline)) {
        constexpr bool WHOLE = decltype(whole_c)::value != 0;
        const int cnt = WHOLE ? 4 : P.nz - iz0;                               // < 4 only behind a row's last whole quad (nz % 4 != 0)
        const long long vo = ((long long)ix * P.ny + iy) * P.nz + iz0;        // first voxel of the quad (= 4 iq when nz % 4 == 0)
        const float fx = fmaf((float)ix, hx, ox), fy = fmaf((float)iy, hy, oy);
        float fz[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) fz[e] = fmaf((float)(iz0 + e), hz, oz);
        float4 m = make_float4(0.f, 0.f, 0.f, 0.f), sm = m, ws = m;
#pragma unroll
        for (int f = 0; f < SAA_MAXF; ++f) {
            if (f >= n_foci) break;                      // uniform
            const float s = scale[f], s2 = s * s, w = wts[f];
            float4 p4, w4;
            {
#pragma clang fp contract(off)      // sums of the ROUNDED scaled values (the stored ones), as the separate kernels form them
            float* const pp = pmag + (long long)f * P.vox + vo;
            float* const ip = inten + (long long)f * P.vox + vo;
            if constexpr (ROWQ && WHOLE) { p4 = ld4u(pp, 4); w4 = ld4u(ip, 4); }
            else if constexpr (ROWQ) { p4 = ld4u(pp, cnt); w4 = ld4u(ip, cnt); }
            else { p4 = OLX_SAA_LD(reinterpret_cast<const float4*>(pp)); w4 = OLX_SAA_LD(reinterpret_cast<const float4*>(ip)); }
            p4.x *= s; p4.y *= s; p4.z *= s; p4.w *= s;
            w4.x *= s2; w4.y *= s2; w4.z *= s2; w4.w *= s2;
            if constexpr (ROWQ) { st4u(pp, p4, cnt); st4u(ip, w4, cnt); }
            else { OLX_SAA_ST(reinterpret_cast<float4*>(pp), p4); OLX_SAA_ST(reinterpret_cast<float4*>(ip), w4); }
            m.x = fmaxf(m.x, p4.x); m.y = fmaxf(m.y, p4.y); m.z = fmaxf(m.z, p4.z); m.w = fmaxf(m.w, p4.w);
            sm.x += w4.x; sm.y += w4.y; sm.z += w4.z; sm.w += w4.w;
            }
            ws.x = fmaxf(ws.x, w * w4.x); ws.y = fmaxf(ws.y, w * w4.y); ws.z = fmaxf(ws.z, w * w4.z); ws.w = fmaxf(ws.w, w * w4.w);
            // the six masked peaks of this focus on the SCALED values (field_analysis_peaks4_k's decisions)
            const MaskFast& M = sM[f];
            const float b0 = fmaf(M.a[1], fy, fmaf(M.a[0], fx, M.a[3])), b1 = fmaf(M.a[5], fy, fmaf(M.a[4], fx, M.a[7])), b2 = fmaf(M.a[9], fy, fmaf(M.a[8], fx, M.a[11]));
            int in_main[4], in_side[4], undecided = 0;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float g0 = fmaf(M.a[2], fz[e], b0), g1 = fmaf(M.a[6], fz[e], b1), g2 = fmaf(M.a[10], fz[e], b2);
                const float d2f = fmaf(g2, g2, fmaf(g1, g1, g0 * g0));
                in_main[e] = mask_fast_side(d2f, M.rin2[0], M.rout2[0]);
                in_side[e] = mask_fast_side(d2f, M.rin2[1], M.rout2[1]);
                undecided |= (in_main[e] == 0 || in_side[e] == 0) ? (1 << e) : 0;
            }
            if (undecided) {
                const double x = P.ox + ix * P.hx, y = P.oy + iy * P.hy;
                const double* a = sA[f];
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    if (!((undecided >> e) & 1)) continue;
                    const double z = P.oz + (iz0 + e) * P.hz;
                    const double q0 = (a[0] * x + a[1] * y + a[2] * z + a[3]) * P.ia0;
                    const double q1 = (a[4] * x + a[5] * y + a[6] * z + a[7]) * P.ia1;
                    const double q2 = (a[8] * x + a[9] * y + a[10] * z + a[11]) * P.ia2;
                    const double d2 = q0 * q0 + q1 * q1 + q2 * q2;
                    in_main[e] = mask_cmp<0>(d2, P.radius, rm2lo, rm2hi) ? 1 : -1;
                    in_side[e] = mask_cmp<2>(d2, r_side, rs2lo, rs2hi) ? -1 : 1;
                }
            }
            const float pv[4] = {p4.x, p4.y, p4.z, p4.w}, wv[4] = {w4.x, w4.y, w4.z, w4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const bool zok = (iz0 + e) >= iz_first;
                if (in_main[e] > 0) { pk[f][0] = fmaxf(pk[f][0], pv[e]); pk[f][1] = fmaxf(pk[f][1], wv[e]); }
                if (zok && in_side[e] < 0) { pk[f][2] = fmaxf(pk[f][2], pv[e]); pk[f][3] = fmaxf(pk[f][3], wv[e]); }
                if (zok) { pk[f][4] = fmaxf(pk[f][4], pv[e]); pk[f][5] = fmaxf(pk[f][5], wv[e]); }
            }
        }
        const float4 mean4 = make_float4(sm.x * inv_n, sm.y * inv_n, sm.z * inv_n, sm.w * inv_n);
        if constexpr (ROWQ) { st4u(pmax + vo, m, cnt); st4u(imean + vo, mean4, cnt); st4u(wint + vo, ws, cnt); }
        else { OLX_SAA_ST(reinterpret_cast<float4*>(pmax + vo), m); OLX_SAA_ST(reinterpret_cast<float4*>(imean + vo), mean4); OLX_SAA_ST(reinterpret_cast<float4*>(wint + vo), ws); }
        const float wsv[4] = {ws.x, ws.y, ws.z, ws.w};
#pragma unroll
        for (int e = 0; e < 4; ++e) if ((iz0 + e) >= iz_first) wmax = fmaxf(wmax, wsv[e]);
    };
    for (int iq = blockIdx.x * blockDim.x + threadIdx.x; iq < nq; iq += stride) {
        int ix, iy, iz0;
        quad_decode(iq, nzq, nyzq, inv_nyzq, inv_nzq, ix, iy, iz0);
        quad(ix, iy, iz0, IntC<1>{});
    }
    if constexpr (ROWQ) {
        if (P.nz & 3) {      // the rows' tails: one thread per row
            const float inv_ny = 1.0f / (float)P.ny;
            for (int r = blockIdx.x * blockDim.x + threadIdx.x; r < P.nx * P.ny; r += stride) {
                int ix = (int)((float)r * inv_ny), iy = r - ix * P.ny;
                if (iy < 0) { --ix; iy += P.ny; } else if (iy >= P.ny) { ++ix; iy -= P.ny; }
                quad(ix, iy, P.nz & ~3, IntC<0>{});
            }
        }
    }
    // block reduction: 6 F + 1 maxima
#pragma unroll
    for (int f = 0; f < SAA_MAXF; ++f)
#pragma unroll
        for (int k = 0; k < 6; ++k) {
            float v = pk[f][k];
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
            if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][f * 6 + k] = v;
        }
    {
        float v = wmax;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v = fmaxf(v, __shfl_xor(v, off, 64));
        if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6][SAA_MAXF * 6] = v;
    }
    __syncthreads();
    if ((int)threadIdx.x < 6 * n_foci) {
        const int k = threadIdx.x;
        atomicMax(peaks + k, __float_as_uint(fmaxf(fmaxf(s_red[0][k], s_red[1][k]), fmaxf(s_red[2][k], s_red[3][k]))));
    }
    if (threadIdx.x == 255) atomicMax(wpeak, __float_as_uint(fmaxf(fmaxf(s_red[0][SAA_MAXF * 6], s_red[1][SAA_MAXF * 6]), fmaxf(s_red[2][SAA_MAXF * 6], s_red[3][SAA_MAXF * 6]))));
}

// ------------------------------------------------------------------------------------
// Pieces of the one-call analysis (olx_solution_analyze): everything Solution.analyze reads off the resident volumes is
// enqueued back to back on the context's stream, the intermediate numbers (mainlobe peaks -> -3 dB centroid cut-offs and beam
// width cut-offs) never leave the device, and ONE copy brings the per-focus reports to the host.
// ------------------------------------------------------------------------------------
// cut[f] = mainlobe |p| peak * factor, in fp32 (what NumPy's float32 array * Python float does on the host path)
__global__ void analysis_cutoffs_k(const unsigned* __restrict__ peaks /*[F][6]*/, int n_foci, float factor, float* __restrict__ cut) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < n_foci) cut[f] = __fmul_rn(__uint_as_float(peaks[6 * f]), factor);
}

// trilinear samples of every focus' |p| volume along its three focal-axis lines: pts [F][npts][3] (fp64, formed on the host with the
// expression Solution.analyze always used), out [F][npts]; same arithmetic as field_sample_k
__global__ __launch_bounds__(128) void field_sample_lines_k(const float* __restrict__ vol, const double* __restrict__ pts, int npts,
                                                            const PeakParams P, float* __restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x, f = blockIdx.y;
    if (i >= npts) return;
    const double* q = pts + ((size_t)f * npts + i) * 3;
    const float* v = vol + (long long)f * P.vol_stride;
    const double c[3] = {(q[0] - P.ox) / P.hx, (q[1] - P.oy) / P.hy, (q[2] - P.oz) / P.hz};
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
    float* o = out + (size_t)f * npts + i;
    if (!inside) { *o = __builtin_nanf(""); return; }
    double acc = 0;
#pragma unroll
    for (int dx = 0; dx < 2; ++dx)
#pragma unroll
        for (int dy = 0; dy < 2; ++dy)
#pragma unroll
            for (int dz = 0; dz < 2; ++dz) {
                const int ix = min(i0[0] + dx, P.nx - 1), iy = min(i0[1] + dy, P.ny - 1), iz = min(i0[2] + dz, P.nz - 1);
                const double ww = (dx ? w[0] : 1 - w[0]) * (dy ? w[1] : 1 - w[1]) * (dz ? w[2] : 1 - w[2]);
                acc += ww * v[((long long)ix * P.ny + iy) * P.nz + iz];
            }
    *o = (float)acc;
}

// get_beam_bounds on the sampled lines (plan/solution_analysis.py:488-535): per (focus, axis, level) the LAST sample at an
// offset <= 0 and the FIRST at an offset >= 0 whose value lies below the level's cut-off (NaN never does); indices into the
// axis line, -1 = none.  One 64-lane block per (axis, focus); cut-off = (float)(main peak * factor) as on the host path.
struct BeamLines { int start[3], n[3], n_le[3], i_ge[3]; double factor[2]; };
__global__ __launch_bounds__(64) void beam_bounds_k(const float* __restrict__ samples /*[F][npts]*/, int npts, const unsigned* __restrict__ peaks,
                                                    const BeamLines L, int* __restrict__ bounds /*[F][3][2][2]*/) {
    const int a = blockIdx.x, f = blockIdx.y, lane = threadIdx.x;
    const float* v = samples + (size_t)f * npts + L.start[a];
    const double mp = (double)__uint_as_float(peaks[6 * f]);
#pragma unroll
    for (int lv = 0; lv < 2; ++lv) {
        const float cut = (float)(mp * L.factor[lv]);
        int neg = -1, pos = 0x7fffffff;
        for (int i = lane; i < L.n[a]; i += 64) {
            const bool below = v[i] < cut;
            if (below && i < L.n_le[a]) neg = max(neg, i);
            if (below && i >= L.i_ge[a]) pos = min(pos, i);
        }
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) { neg = max(neg, __shfl_xor(neg, off, 64)); pos = min(pos, __shfl_xor(pos, off, 64)); }
        if (lane == 0) {
            int* b = bounds + ((f * 3 + a) * 2 + lv) * 2;
            b[0] = neg; b[1] = pos == 0x7fffffff ? -1 : pos;
        }
    }
}


}  // namespace olx
