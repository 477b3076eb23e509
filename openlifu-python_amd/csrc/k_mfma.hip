// kernel 2c (field_mfma_k): shared-geometry accumulate, contraction on the matrix cores
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

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




template <int MT, int NT, int MX, int MY, bool FLAT, bool CLAMP>
__global__ __launch_bounds__(FIELD_THREADS) void field_mfma_k(
    const float4* __restrict__ coords /*[n_el_pad][2]: { kx, ky, kz, 0 }, { fx, fy, fz, 0 } -- the element as (index of the nearest coordinate step, offset from it [wavelengths])*/, const uint4* __restrict__ bfrag /*[tiles][ks][NT][2][64]*/,
    float* __restrict__ pmag, float* __restrict__ inten, float* __restrict__ cplx,
    const int* __restrict__ targets /*[tiles][32][4]: focus*4 + image of every store target of a column, -1 = none*/,
    const MfmaParams P) {
    constexpr int CH = MFMA_ELEMS_LDS / NT;              // elements per LDS chunk
    __shared__ float4 s_xyz[CH][2];
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
    // Coordinates as (index, residual), round 6 (as kernels 2a / 2b: x_v - x_e = (k_v - k_e) s - f_e with an exact index difference -- absolute fp32 coordinates
    // lose 1e-6 wavelengths, which is 1e-5 of a term a tenth of a wavelength from its element); steps: half a voxel on a folded axis, a voxel otherwise
    // NEAR = the CLAMP instantiations (the host launches them wherever a voxel comes within a quarter wavelength of an element); elsewhere absolute coordinates as before
    constexpr bool NEAR = CLAMP;
    const float sx = (MX == 2) ? 0.5f * P.hx : P.hx, sy = (MY == 2) ? 0.5f * P.hy : P.hy;
    const float x = ((MX == 2) ? (float)(2 * i - (P.nx - 1)) : (float)(i + P.x_begin)) * (NEAR ? 1.0f : sx);
    const float y = ((MY == 2) ? (float)(2 * j - (P.ny - 1)) : (float)j) * (NEAR ? 1.0f : sy);
    float zz[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        zz[t] = NEAR ? (float)(zb + 16 * t + r16) : (float)(zb + 16 * t + r16) * P.hz;
        if (FLAT) { const float dz = NEAR ? fmaf(zz[t] - P.flat_kz, P.hz, -P.flat_fz) : zz[t] - P.flat_ez; zz[t] = dz * dz; }
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
        for (int q = tid; q < 2 * n_here; q += FIELD_THREADS) (&s_xyz[0][0])[q] = coords[2 * chunk + q];
        const uint4* src = bfrag + ((size_t)tile * ks_total + chunk / 16) * (NT * 128);
        for (int q = tid; q < nks * NT * 128; q += FIELD_THREADS) (&s_B[0][0][0][0])[q] = src[q];
        __syncthreads();
        if (!active) continue;
        for (int ks = 0; ks < nks; ++ks) {
            float r2[4], ez[4], fz[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const float4 e = s_xyz[16 * ks + 4 * g + q][0], f = s_xyz[16 * ks + 4 * g + q][1];
                const float dx = NEAR ? fmaf(x - e.x, sx, -f.x) : x - e.x, dy = NEAR ? fmaf(y - e.y, sy, -f.y) : y - e.y;
                r2[q] = fmaf(dy, dy, dx * dx);
                ez[q] = e.z; fz[q] = f.z;
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
                // (the plain operations of two G values in packed fp32 instructions, as kernels 2a / 2b have them, measured +5 % here -- jittered
                // shard 2.00 -> 2.10 ms, tilted 0.705 -> 0.750: beside matrix instructions packed fp32 loses, MI355X_MICROARCH.md)
#pragma unroll
                for (int q = 0; q < 4; ++q) {
                    float d2;
                    if (FLAT) {
                        d2 = r2[q] + zz[t];
                    } else {
                        const float dz = NEAR ? fmaf(zz[t] - ez[q], P.hz, -fz[q]) : zz[t] - ez[q];
                        d2 = fmaf(dz, dz, r2[q]);
                    }
                    if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                    const float ri = __builtin_amdgcn_rsqf(d2);
                    const float ph = d2 * ri;            // distance in wavelengths = phase in revolutions
                    const float rs = ri * P.g_scale;
                    const float gr = rs * __builtin_amdgcn_cosf(ph);
                    const float gi = rs * __builtin_amdgcn_sinf(ph);
                    const auto hi = __builtin_amdgcn_cvt_pkrtz(gr, gi);
                    // lo = g - (float)hi: with >= 2 column tiles in ONE mixed-precision fma per component (the compiler's form is a convert and a
                    // subtract -- same bits): tilted 8-focus shard 0.731 -> 0.711 ms, 64-focus sweeps -1.8 %; with one column tile the same change
                    // measured +7 % (jittered shard 2.005 -> 2.146 ms, same box, alternating), so that shape keeps the compiler's form
                    float lr, li;
                    if constexpr (NT >= 2) {
                        const unsigned hw = __builtin_bit_cast(unsigned, hi);
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
                        asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
                    } else { lr = gr - (float)hi[0]; li = gi - (float)hi[1]; }
                    const auto lo = __builtin_amdgcn_cvt_pkrtz(lr, li);
                    ah.w[q] = __builtin_bit_cast(unsigned, hi);
                    al.w[q] = __builtin_bit_cast(unsigned, lo);
                }
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bh[nt].h, acc[t][nt], 0, 0, 0);
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bl[nt].h, acc[t][nt], 0, 0, 0);
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
    const bool fast = zb + RUN <= P.nz;   // wave-uniform: every z of the run exists (dword-aligned 16-byte stores: rows of odd length are fine)
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
                        *reinterpret_cast<floatx4u_t*>(dst_arr + base + 16 * t) = floatx4u_t{w[t][0], w[t][1], w[t][2], w[t][3]};
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


}  // namespace olx

using namespace olx;

template <int MT, int NT, int MX, int MY>
static void launch_mfma(olx_ctx* c, float* pm) {
    const MfmaParams& M = c->mp;
    const long long rpr = (M.nz + MT * 16 - 1) / (MT * 16);
    const long long runs = (long long)(M.nx - (MX == 2 ? M.nx / 2 : 0)) * (M.ny - (MY == 2 ? M.ny / 2 : 0)) * rpr;
    dim3 grid((unsigned)((runs + 3) / 4), M.n_tiles), blk(FIELD_THREADS);
#define OLX_MF(FL, CL) hipLaunchKernelGGL((field_mfma_k<MT, NT, MX, MY, FL, CL>), grid, blk, 0, c->stream, c->d_coords, c->d_bfrag, pm, c->d_inten, c->d_cplx, c->d_targets, M)
    if (c->flat) { if (c->near) OLX_MF(true, true); else OLX_MF(true, false); }      // (near: a voxel within a quarter wavelength of an element -- the clamp instantiations carry the split coordinates)
    else         { if (c->near) OLX_MF(false, true); else OLX_MF(false, false); }
#undef OLX_MF
}

template <int MX, int MY>
static void dispatch_mfma_nt(olx_ctx* c, float* pm) {
    const bool big = c->mp.nz >= 48;
    if (big) { if (c->nt == 1) launch_mfma<4, 1, MX, MY>(c, pm); else if (c->nt == 2) launch_mfma<4, 2, MX, MY>(c, pm); else launch_mfma<4, 4, MX, MY>(c, pm); }
    else     { if (c->nt == 1) launch_mfma<1, 1, MX, MY>(c, pm); else if (c->nt == 2) launch_mfma<1, 2, MX, MY>(c, pm); else launch_mfma<1, 4, MX, MY>(c, pm); }
}

void olx_launch_mfma(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) dispatch_mfma_nt<2, 2>(c, pm);
    else if (c->mx == 2) dispatch_mfma_nt<2, 1>(c, pm);
    else if (c->my == 2) dispatch_mfma_nt<1, 2>(c, pm);
    else dispatch_mfma_nt<1, 1>(c, pm);
}
