// kernel 2f, wave-specialised form (field_toepws_k): one steering column on a lattice array -- persistent blocks, one team of
// waves evaluates the NEXT geometry tables while the other contracts the CURRENT ones on the matrix pipe
// gfx950 (CDNA4, wave64) only.  Mathematics, operands and table layout: k_toep.hip (kernel 2f).
#ifdef OLX_AB_VARIANTS   // measured-slower A/B form: compiled only into the developer library (build.py -DOLX_AB_VARIANTS), never into libolx.so
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"
#include "k_toep.hip.h"

namespace olx {

// ------------------------------------------------------------------------------------
// In field_toep_k a block alternates between table generation (VALU: 3 transcendentals per entry and plane) and the
// contraction (matrix pipe), separated by barriers; with two 75 KB blocks per CU the two pipes overlap only by chance
// (matrix pipe 31 - 41 % busy, tools/stamps_toep.py).  Here ONE 8-wave block per CU owns BOTH table buffers (2 x 74.75 KB) and
// walks a static list of work items (item = coset x position part x 16-plane block; step = item x element super-block):
//     waves 4-7 (generators):  step q + 1's tables -> buffer (q + 1) & 1          VALU
//     waves 0-3 (contractors): step q's contraction from buffer q & 1             LDS reads + MFMA; the item's stores at its last step
// one __syncthreads() per step hands the buffers over.  Every SIMD hosts one wave of each team, so its VALU and its matrix pipe
// both stay busy; a step lasts max(tables, contraction) instead of their sum.
// RESULT (tools/stamps_toepws.py): 13 % (single focus, 256^3) to 35 % (1024 el x 512^3) SLOWER than field_toep_k.
// A generator wave needs 6.7 k cycles per step beside a contractor (4.8 k) on the same SIMD, not the 2.2 k of its instruction
// count: an MFMA holds the SIMD's vector issue for 8 of its 16 cycles, so table generation (VALU-issue bound: 14 instructions
// per entry and plane, 3 of them transcendental) and the contraction share one issue port rather than overlapping -- the
// sum is what four interleaved waves per SIMD already achieve in field_toep_k.  Kept for A/B runs: OLX_FIELD_VARIANT=toepws.
//   * Contractor w owns the y positions ky = w, w + 4, w + 8 and BOTH K-steps (no partial sums to exchange).  Toeplitz-weight
//     fragments come through L2 in a three-slot register ring, two element rows ahead; the first two of the next step are
//     requested before the hand-over barrier, which orders LDS only (lds_barrier: loads and the epilogue's stores stay in flight).
//   * Epilogue per accumulator tile through a wave-private 1.25 KB LDS tile (no block barrier): lane = (output, kx, plane quad)
//     -> one 16-byte store per target, as in field_toep_k.
//   * All items of a launch use the same table extent (the largest position part): columns / rows beyond an item's own are
//     finite geometry values that only meet unused accumulator rows; the two never-written pad columns are zeroed once.
// ------------------------------------------------------------------------------------
#ifdef OLX_EXP_STAMPS
#define TWS_STAMP(k) do { if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 512) g_stamps[blockIdx.x * 8 + wave][k] = __builtin_readcyclecounter(); } while (0)
#else
#define TWS_STAMP(k)
#endif
constexpr int TWS_WAVES = 8, TWS_TEAM = 4;
constexpr int TWS_BUF = 2 * TOEP_ZB * TOEP_PSZ;              // words per table buffer: [hi | lo][plane][row][ud']
constexpr int TWS_RING = 3;

struct ToepWsParams {
    ToepParams t;
    int n_items;               // 2 mx my nsx nsy kblocks
    int ncmax, nrmax;          // table columns / rows generated for every item
};

struct ToepItem { int ibase, jbase, KX, KY, k0; };

__device__ __forceinline__ ToepItem toep_item(const CosetParams& P, unsigned b) {
    ToepItem it;
    const int kblock = (int)(b % (unsigned)P.kblocks); b /= (unsigned)P.kblocks;
    const int sy_part = (int)(b % (unsigned)P.nsy); b /= (unsigned)P.nsy;
    const int sx_part = (int)(b % (unsigned)P.nsx); b /= (unsigned)P.nsx;
    const int ry = (int)(b % (unsigned)P.my), rx = (int)(b / (unsigned)P.my);          // rx < 2 mx
    const int wx = P.nx - P.x_lo, wy = P.ny - P.y_lo;
    const int kx_all = rx < wx ? (wx - 1 - rx) / (2 * P.mx) + 1 : 0, ky_all = ry < wy ? (wy - 1 - ry) / P.my + 1 : 0;
    const int kx0 = sx_part * kx_all / P.nsx, ky0 = sy_part * ky_all / P.nsy;
    it.KX = (sx_part + 1) * kx_all / P.nsx - kx0;
    it.KY = (sy_part + 1) * ky_all / P.nsy - ky0;
    it.ibase = P.x_lo + rx + 2 * P.mx * kx0; it.jbase = P.y_lo + ry + P.my * ky0;
    it.k0 = kblock * TOEP_ZB;
    return it;
}

template <int MX, int MY, bool CLAMP>
__global__ __launch_bounds__(TWS_WAVES * 64, 2) void field_toepws_k(const uint4* __restrict__ afrag, float* __restrict__ pmag,
                                                                     float* __restrict__ inten, const ToepWsParams W) {
    const ToepParams& T = W.t;
    const CosetParams& P = T.q;
    __shared__ __attribute__((aligned(16))) unsigned s_T[2 * TWS_BUF];
    __shared__ __attribute__((aligned(16))) float s_x[TWS_TEAM][16 * TOEP_XS];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool generator = wave >= TWS_TEAM;
    const int w4 = wave & (TWS_TEAM - 1);
    const int n_sb = T.nsa16 * P.nsb;
    const int n_mine = ((int)W.n_items - (int)blockIdx.x + (int)gridDim.x - 1) / (int)gridDim.x;
    const int Q = n_mine * n_sb;
    // pad columns (ncmax .. 31) of both buffers: never written, their Toeplitz weights are zero -- 0 x garbage must stay 0
    for (int idx = tid; idx < 2 * TOEP_ROWS * (TOEP_TW - W.ncmax); idx += TWS_WAVES * 64) {
        const int bsel = idx / (TOEP_ROWS * (TOEP_TW - W.ncmax)), rem = idx - bsel * TOEP_ROWS * (TOEP_TW - W.ncmax);
        const int row = rem / (TOEP_TW - W.ncmax), col = W.ncmax + rem - row * (TOEP_TW - W.ncmax);
#pragma unroll
        for (int z = 0; z < TOEP_ZB; ++z) {
            s_T[bsel * TWS_BUF + z * TOEP_PSZ + row * TOEP_TW + col] = 0u;
            s_T[bsel * TWS_BUF + TOEP_ZB * TOEP_PSZ + z * TOEP_PSZ + row * TOEP_TW + col] = 0u;
        }
    }
    const float inv_nc = 1.0f / (float)W.ncmax;

    // ---- generator: tables of step q into buffer q & 1
    auto generate = [&](int q) {
        const int item = (int)blockIdx.x + (q / n_sb) * (int)gridDim.x, sb = q % n_sb;
        const ToepItem it = toep_item(P, (unsigned)item);
        if (it.KX <= 0 || it.KY <= 0) return;           // block-uniform
        const int sa = sb / P.nsb, sbb = sb - sa * P.nsb;
        unsigned* const hi_t = s_T + (q & 1) * TWS_BUF;
        unsigned* const lo_t = hi_t + TOEP_ZB * TOEP_PSZ;
        const float dz0 = (float)it.k0 * P.hz - P.flat_ez;
        for (int idx = tid - TWS_TEAM * 64; idx < W.nrmax * W.ncmax; idx += TWS_TEAM * 64) {
            const int row = (int)(((float)idx + 0.5f) * inv_nc), col = idx - row * W.ncmax;     // exact for these small integers
            const float U = (float)(it.ibase + P.x_begin + P.ux0 + P.mx * (col - 15) - TOEP_SA * P.mx * sa);
            const float Wd = (float)(it.jbase + P.uy0 + P.my * (row - 7) - TOEP_SB * P.my * sbb);
            const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
            const float dy = fmaf(Wd, P.hy_hi, fmaf(Wd, P.hy_lo, P.fy0));
            const float r2 = fmaf(dy, dy, dx * dx);
            const int o = row * TOEP_TW + col;
#pragma unroll
            for (int z = 0; z < TOEP_ZB; ++z) {
                const float dz = fmaf((float)z, P.hz, dz0);
                float d2 = fmaf(dz, dz, r2);
                if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                const float ri = __builtin_amdgcn_rsqf(d2);
                const float ph = d2 * ri;
                const float rs = ri * P.g_scale;
                const float gr = rs * __builtin_amdgcn_cosf(ph);
                const float gi = rs * __builtin_amdgcn_sinf(ph);
                const half2_t hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
                hi_t[z * TOEP_PSZ + o] = __builtin_bit_cast(unsigned, hi);
                lo_t[z * TOEP_PSZ + o] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(gr - (float)hi[0], gi - (float)hi[1]));
            }
        }
    };

    // ---- contractor state
    const int n16 = lane & 15, g = lane >> 4;
    const unsigned bbase = (unsigned)(n16 * TOEP_PSZ + 4 * g);
    floatx4_t acc[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) acc[t] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    uint4 ring[TWS_RING][4];                            // Toeplitz weights of one element row: {hi s0, hi s1, lo s0, lo s1}
    auto a_src = [&](int q) {                           // fragments of step q's super-block (the same for every item)
        const int sb = q % n_sb, sa = sb / P.nsb, sbb = sb - sa * P.nsb;
        return afrag + ((size_t)(blockIdx.y * T.nsa16 + sa) * T.ay_pad + TOEP_SB * sbb) * 4 * 64 + lane;
    };
    auto a_load = [&](const uint4* ab, int bl) {
#pragma unroll
        for (int c = 0; c < 4; ++c) ring[bl % TWS_RING][c] = ab[(bl * 4 + c) * 64];
    };

    auto contract = [&](int q) {
        const int item = (int)blockIdx.x + (q / n_sb) * (int)gridDim.x, sb = q % n_sb;
        const ToepItem it = toep_item(P, (unsigned)item);
        const bool live = it.KX > 0 && it.KY > 0;       // block-uniform
        const unsigned* const hi_t = s_T + (q & 1) * TWS_BUF;
        const unsigned* const lo_t = hi_t + TOEP_ZB * TOEP_PSZ;
        const uint4* ab = a_src(q);
        if (live) {
            // One contractor per SIMD: nothing else hides its LDS latency, so the table fragments of element row bl + 1 (3 y
            // positions x 2 K-steps x {hi, lo} = 12 ds_read_b128) are requested before the 18 MFMAs of row bl issue.  Straight-line
            // code: a y position past the part's own (ky >= KY) reads the last valid row and feeds an accumulator nobody stores.
            unsigned w0[3];
#pragma unroll
            for (int t = 0; t < 3; ++t) w0[t] = bbase + (unsigned)(min(w4 + TWS_TEAM * t, it.KY - 1) * TOEP_TW);
            uint4 fb[2][3][2][2];                       // [buffer][y position][K-step][hi | lo]
            auto b_load = [&](int bl) {
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {       // table row ky - bl + 7: the (7 - bl) part and the K-step are immediates
                        fb[bl & 1][t][s][0] = *reinterpret_cast<const uint4*>(hi_t + w0[t] + (7 - bl) * TOEP_TW + 16 * s);
                        fb[bl & 1][t][s][1] = *reinterpret_cast<const uint4*>(lo_t + w0[t] + (7 - bl) * TOEP_TW + 16 * s);
                    }
            };
            b_load(0);
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int bl = 0; bl < TOEP_SB; ++bl) {
                Half8Bits ah[2], al[2];
                ah[0].u = ring[bl % TWS_RING][0]; ah[1].u = ring[bl % TWS_RING][1];
                al[0].u = ring[bl % TWS_RING][2]; al[1].u = ring[bl % TWS_RING][3];
                if (bl + TWS_RING - 1 < TOEP_SB) a_load(ab, bl + TWS_RING - 1);     // two element rows ahead
                if (bl + 1 < TOEP_SB) b_load(bl + 1);
                __builtin_amdgcn_sched_barrier(0);      // (left alone, the scheduler sinks every read to its use and waits for it there)
#pragma unroll
                for (int t = 0; t < 3; ++t)
#pragma unroll
                    for (int s = 0; s < 2; ++s) {
                        Half8Bits bh, bw;
                        bh.u = fb[bl & 1][t][s][0]; bw.u = fb[bl & 1][t][s][1];
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s].h, bh.h, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s].h, bh.h, acc[t], 0, 0, 0);
                        acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s].h, bw.h, acc[t], 0, 0, 0);
                    }
            }
        }
        if (q + 1 < Q) {                                // next step's first rows: in flight across the hand-over barrier
            const uint4* nb = a_src(q + 1);
#pragma unroll
            for (int bl = 0; bl < TWS_RING - 1; ++bl) a_load(nb, bl);
        }
        if (sb != n_sb - 1) return;
        // ---- the item is complete: |p| / intensity and stores, one accumulator tile at a time through this wave's LDS tile
        if (live) {
            float* const xt = s_x[w4];
            const int out = lane >> 5, kx = (lane >> 2) & 7, pq = lane & 3;
            const int kz = it.k0 + 4 * pq;
            const float sc = out ? P.out_scale * P.out_scale * P.inten_scale : P.out_scale;
            float* const vol = out ? inten : pmag;
            const bool want = (P.flags & (out ? 2u : 1u)) != 0 && kx < it.KX && kz < P.nz;
            auto readout = [&](auto full_c) {           // (separate copies: one body would split the 16-byte stores into 12 + 4)
                constexpr bool FULL4 = decltype(full_c)::value != 0;
#pragma unroll
                for (int t = 0; t < 3; ++t) {
                    const int ky = w4 + TWS_TEAM * t;
                    if (ky >= it.KY) continue;
#pragma unroll
                    for (int r = 0; r < 4; ++r) xt[(4 * g + r) * TOEP_XS + n16] = acc[t][r];
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    const float4 re = *reinterpret_cast<const float4*>(xt + (2 * kx) * TOEP_XS + 4 * pq);
                    const float4 im = *reinterpret_cast<const float4*>(xt + (2 * kx + 1) * TOEP_XS + 4 * pq);
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                    if (!want) continue;
                    const float rr[4] = {re.x, re.y, re.z, re.w}, ii[4] = {im.x, im.y, im.z, im.w};
                    float v[4];
#pragma unroll
                    for (int e = 0; e < 4; ++e) {
                        const float m2 = fmaf(rr[e], rr[e], ii[e] * ii[e]);
                        v[e] = (out ? m2 : __builtin_amdgcn_sqrtf(m2)) * sc;
                    }
                    const int i = it.ibase + 2 * P.mx * kx, j = it.jbase + P.my * ky;
#pragma unroll
                    for (int c = 0; c < 4; ++c) {
                        const int code = T.targets[c];
                        if (code < 0) continue;             // uniform
                        const int m = code & 3;
                        const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                        const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
                        float* dst = vol + (long long)(code >> 2) * P.vox + (unsigned)((io * P.ny + jo) * P.nz + kz);
                        if constexpr (FULL4) *reinterpret_cast<float4*>(dst) = make_float4(v[0], v[1], v[2], v[3]);
                        else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (kz + e < P.nz) dst[e] = v[e];
                        }
                    }
                }
            };
            if ((P.nz & 3) == 0) readout(IntC<1>{}); else readout(IntC<0>{});
        }
#pragma unroll
        for (int t = 0; t < 3; ++t) acc[t] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    };

    // ---- the pipeline
    lds_barrier();                                      // pad columns zeroed
    if (generator) { if (Q > 0) generate(0); }
    else if (Q > 0) {
        const uint4* nb = a_src(0);
#pragma unroll
        for (int bl = 0; bl < TWS_RING - 1; ++bl) a_load(nb, bl);
    }
    for (int q = 0; q < Q; ++q) {
        if (q == 4) TWS_STAMP(0);
        lds_barrier();
        if (q == 4) TWS_STAMP(1);                                  // buffer q & 1 is complete, buffer (q + 1) & 1 is free (LDS only: the
                                                        // contractors' weight loads and stores stay in flight across it)
        if (generator) { if (q + 1 < Q) generate(q + 1); }
        else contract(q);
        if (q == 4) TWS_STAMP(2);
        if (q == 5) TWS_STAMP(3);
    }
    TWS_STAMP(4);
}

}  // namespace olx

using namespace olx;

#ifdef OLX_EXP_STAMPS
extern "C" int olx_exp_read_stamps_toepws(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(olx::g_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif

template <int MX, int MY>
static void launch_toepws(olx_ctx* c, float* pm) {
    ToepWsParams W;
    W.t.q = c->cp; W.t.nsa16 = c->toep_nsa16; W.t.ay_pad = 8 * c->lat.nsb;
    for (int q = 0; q < 4; ++q) W.t.targets[q] = c->toep_targets[q];
    const CosetParams& Q = W.t.q;
    W.n_items = (int)((long long)2 * Q.mx * Q.my * Q.nsx * Q.nsy * Q.kblocks);
    // largest position part of any coset: kx_all <= ceil(wx / (2 mx)), parts of ceil(kx_all / nsx)
    const int wx = Q.nx - Q.x_lo, wy = Q.ny - Q.y_lo;
    const int kxm = ((wx + 2 * Q.mx - 1) / (2 * Q.mx) + Q.nsx - 1) / Q.nsx, kym = ((wy + Q.my - 1) / Q.my + Q.nsy - 1) / Q.nsy;
    W.ncmax = 15 + 2 * (std::max(kxm, 1) - 1) + 1; W.nrmax = std::max(kym, 1) + TOEP_SB - 1;
    const int blocks = std::min(W.n_items, c->n_cu > 0 ? c->n_cu : 256);
    dim3 grid((unsigned)blocks, c->mp.n_tiles), blk(TWS_WAVES * 64);
    if (c->clamp || c->lat.clamp) hipLaunchKernelGGL((field_toepws_k<MX, MY, true>), grid, blk, 0, c->stream, c->d_afrag, pm, c->d_inten, W);
    else hipLaunchKernelGGL((field_toepws_k<MX, MY, false>), grid, blk, 0, c->stream, c->d_afrag, pm, c->d_inten, W);
}

void olx_launch_toepws(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) launch_toepws<2, 2>(c, pm);
    else if (c->mx == 2) launch_toepws<2, 1>(c, pm);
    else if (c->my == 2) launch_toepws<1, 2>(c, pm);
    else launch_toepws<1, 1>(c, pm);
}
#endif  // OLX_AB_VARIANTS
