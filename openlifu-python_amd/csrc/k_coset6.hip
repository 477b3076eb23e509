// kernel 2g, four column tiles (field_cosetp4_k): kernel 2g's row map -- planes in the MFMA rows, stores straight from the
// accumulators -- for launch tiles of 17 - 32 steering columns (focal-pattern sweeps, shards without shared mirror images)
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#ifdef OLX_AB_VARIANTS   // measured-slower A/B form: compiled only into the developer library (build.py -DOLX_AB_VARIANTS), never into libolx.so
#include <algorithm>
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// With 17 - 32 columns per launch tile (NT = 4) kernel 2e lists (plane, position) down the MFMA rows and transposes every block's
// output through LDS (|p| -> staging -> barrier -> job-list read-out).  Kernel 2g's row map removes that phase for NT = 2; this is
// the same map with four column tiles:
//   * block = one coset part of KX <= 2 x KY <= 11 positions x 16 planes (kernel 2e's NT = 4 footprint and block records);
//     MFMA tile = ONE position x 16 planes; wave w takes positions w, w + 8, w + 16 (<= 3 tiles x 4 column tiles = 48 accumulators).
//   * tables: 18 rows x 10 offsets per plane and 8 x 8 element super-block (no pair tables: one super-block of steering
//     fragments, 32 KB, per LDS stage), evaluated by the waves for planes 2 w, 2 w + 1 and shared by the block.  Row stride 10 words
//     (5 eight-byte slots), plane stride 236 words (118 = 22 (mod 32) slots): the 32 lanes of a ds_read_b64 group -- 16 planes x 2
//     k-groups -- hit 32 distinct slots, as in kernel 2g.
//   * K-step: the geometry fragments of the wave's three tiles stay in registers while the four column tiles' steering
//     fragments stream through (8 registers at a time): 12 matrix instructions per tile, three fp16 hi/lo products.
//   * epilogue as kernel 2g: |p| / intensity in place (quad swap), one 16-byte store per store target from the accumulators.
// MEASURED (round 3, same box, alternating runs; profiles/r03_cosetp4_ab.txt): SLOWER than kernel 2e's NT = 4 shape -- 64-focus sweep
// 4.40 vs 3.80 ms, off-axis 8-focus shard 0.945 vs 0.847 ms.  With four column tiles the K-steps are twice as long as in 2g's NT = 2 shape
// and the output phase it removes weighs half as much, while the block-shared tables put a second barrier wait (the slowest of eight
// table generators) in front of every one of the four single-super-block stages; 2e's tables are wave-private.  Kept in the developer
// library (OLX_FIELD_VARIANT=cosetp4) as evidence for DESIGN.md 5.4; the planner keeps kernel 2e for NT = 4.
// ------------------------------------------------------------------------------------
constexpr int C4_NT = 4, C4_KXW = 2;
constexpr int C4_UW = 8 + 2 * (C4_KXW - 1);         // 10 table columns: ud = 2 kx - a in [-7, 2]
constexpr int C4_TW = 10, C4_TROWS = 18, C4_ROW0 = 7, C4_PSZ = 236;
constexpr int C4_MT = 3;                            // tiles (positions) per wave: ceil(22 / 8)
static_assert(C4_TROWS * C4_TW <= C4_PSZ, "table does not fit its plane stride");

template <int MX, int MY, bool CLAMP>
__global__ __launch_bounds__(COS_NW * 64, 4) void field_cosetp4_k(
    const uint4* __restrict__ bfrag, float* __restrict__ pmag, float* __restrict__ inten,
    const int* __restrict__ targets /*[tiles][32 columns][4]: focus * 4 + mirror image, -1 = none*/,
    const CosetBlock* __restrict__ blocks /*[gridDim.x]*/, const CosetParams P) {
    constexpr int NT = C4_NT, THREADS = COS_NW * 64;
    constexpr int RPR = 64 / C4_UW, NROUND = (C4_TROWS + RPR - 1) / RPR;        // 6 table rows per generation round, 3 rounds
    constexpr int B_BYTES = 4 * NT * 2 * 64 * 16;                               // ONE super-block of steering fragments (32 KB)
    constexpr int T_WORDS = COS_ZB * C4_PSZ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES + 2 * T_WORDS * 4 + 64];
    unsigned* const s_hi = reinterpret_cast<unsigned*>(smem + B_BYTES);
    unsigned* const s_lo = s_hi + T_WORDS;
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const CosetBlock BK = blocks[blockIdx.x];
    const int npos = BK.npos, KY = BK.KY, ky_magic = BK.ky_magic;
    if (npos <= 0) return;                              // block-uniform
    const int ibase = BK.ibase, jbase = BK.jbase, k0 = BK.k0;
    const int ntile = __builtin_amdgcn_readfirstlane((npos - wave + COS_NW - 1) / COS_NW);      // this wave's positions: wave, wave + 8, wave + 16
    int toff[C4_MT];                                    // wave-uniform: the tile's position in the table
#pragma unroll
    for (int t = 0; t < C4_MT; ++t) {
        const int pos = min(wave + COS_NW * t, npos - 1);
        const int kx = (pos * ky_magic) >> 16, ky = pos - kx * KY;                // scalar: pos / KY, exact for pos <= 40
        toff[t] = (ky + C4_ROW0) * C4_TW + (C4_UW - 8 - 2 * kx);
    }
    floatx4_t acc[C4_MT][NT];
#pragma unroll
    for (int t = 0; t < C4_MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    const int n_sb = P.nsa * P.nsbp;
    constexpr int CHUNK_U4 = 4 * NT * 128, PRE = CHUNK_U4 / THREADS;
    static_assert(CHUNK_U4 % THREADS == 0, "chunk must split evenly over the block");
    uint4 pre[PRE];
    const uint4* const bsrc = bfrag + (size_t)tile * n_sb * CHUNK_U4;
#pragma unroll
    for (int q = 0; q < PRE; ++q) pre[q] = bsrc[tid + q * THREADS];
    for (int sb = 0; sb < n_sb; ++sb) {
        const int sa = sb / P.nsbp, sbb = sb - sa * P.nsbp;
        if (sb > 0) __syncthreads();                     // previous super-block consumed: steering stage and tables are free
        // ---- G tables of planes 2 wave, 2 wave + 1: 18 rows x 10 offsets
        if (k0 + wave * COS_P < P.nz) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int wl = lane_o / C4_UW, ui = lane_o - C4_UW * wl;
            const bool gen_lane = wl < RPR;
            const int Ulane = ibase + P.x_begin + P.ux0 + P.mx * (ui - 7);
            const int Wlane = jbase + P.uy0 + P.my * (wl - C4_ROW0);
            const int tw_off = (wave * COS_P) * C4_PSZ + wl * C4_TW + (C4_UW - 1 - ui);   // + z PSZ + RPR r TW
            float dz2[COS_P];
#pragma unroll
            for (int z = 0; z < COS_P; ++z) {
                const float dz = (float)(k0 + wave * COS_P + z) * P.hz - P.flat_ez;
                dz2[z] = dz * dz;
            }
            const float U = (float)(Ulane - 8 * P.mx * sa);
            const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
            const float dx2 = dx * dx;
            const int Wsb = Wlane - 8 * P.my * sbb;
#pragma unroll
            for (int r = 0; r < NROUND; ++r) {
                const bool row_ok = gen_lane && RPR * r + wl < C4_TROWS;
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
                    const half2_t hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
                    float lr, li;                        // lo = g - (float)hi in ONE mixed-precision fma per component
                    const unsigned hw = __builtin_bit_cast(unsigned, hi);
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
                    if (row_ok) {
                        const int o = z * C4_PSZ + tw_off + RPR * r * C4_TW;
                        s_hi[o] = hw;
                        s_lo[o] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lr, li));
                    }
                }
            }
        }
        // this super-block's steering fragments (requested one super-block ahead; the first ones arrive behind the table generation above)
#pragma unroll
        for (int q = 0; q < PRE; ++q) reinterpret_cast<uint4*>(smem)[tid + q * THREADS] = pre[q];
        __syncthreads();
        if (sb + 1 < n_sb) {   // next super-block's fragments: in flight during the K-steps, drained by the next barrier
#pragma unroll
            for (int q = 0; q < PRE; ++q) pre[q] = bsrc[(size_t)(sb + 1) * CHUNK_U4 + tid + q * THREADS];
        }
        if (sbb >= P.nsb) continue;                      // (padding super-block of a padded slot map: zero weights)
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        const int a_off = (lane_k & 15) * C4_PSZ - (lane_k >> 4) * C4_TW;         // plane, k-group
        const uint4* const b_lane = reinterpret_cast<const uint4*>(smem) + lane_k;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {                 // unrolled: the K-step's table offset becomes an immediate
            const int ka = ks & 1, kb = ks >> 1;
            const int kso = 4 * ka - 4 * kb * C4_TW;
            Half8Bits ah[C4_MT], al[C4_MT];
#pragma unroll
            for (int t = 0; t < C4_MT; ++t) {
                if (t >= ntile) continue;                // wave-uniform
                const int ro = a_off + toff[t];
                const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(s_hi + ro + kso);
                const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(s_lo + ro + kso);
                const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                ah[t].w[0] = (unsigned)h0; ah[t].w[1] = (unsigned)(h0 >> 32); ah[t].w[2] = (unsigned)h1; ah[t].w[3] = (unsigned)(h1 >> 32);
                al[t].w[0] = (unsigned)l0; al[t].w[1] = (unsigned)(l0 >> 32); al[t].w[2] = (unsigned)l1; al[t].w[3] = (unsigned)(l1 >> 32);
            }
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) {
                Half8Bits bh, bl;
                bh.u = b_lane[(ks * NT + nt) * 128];
                bl.u = b_lane[(ks * NT + nt) * 128 + 64];
#pragma unroll
                for (int t = 0; t < C4_MT; ++t) {        // (product outermost -- consecutive instructions to different accumulators -- measured 7 % slower)
                    if (t >= ntile) continue;            // wave-uniform
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t].h, bh.h, acc[t][nt], 0, 0, 0);
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[t].h, bh.h, acc[t][nt], 0, 0, 0);
                    acc[t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[t].h, bl.h, acc[t][nt], 0, 0, 0);
                }
            }
        }
    }
    // ---- epilogue, straight from the accumulators (as kernel 2g): lane (g, c16) holds rows 4 g .. 4 g + 3 = planes k0 + 4 g .. + 3 of the
    // tile's position, column c16 = (o, re | im) of each column tile
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int c16 = lane_e & 15, part = c16 & 1;
    const int kz = k0 + 4 * (lane_e >> 4);
    if (kz >= P.nz) return;
    const float s_lane = part == 0 ? P.out_scale : P.out_scale * P.out_scale * P.inten_scale;
    float* const vol = part ? inten : pmag;
    const bool want = (P.flags & (part ? 2u : 1u)) != 0;
#pragma unroll
    for (int t = 0; t < C4_MT; ++t) {
        if (t >= ntile) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float a0 = acc[t][nt][r], a1 = acc[t][nt][r + 1];
                const float sq0 = a0 * a0, sq1 = a1 * a1;
                const float m0 = __builtin_fmaf(a0, a0, quad_swap1(sq0)), m1 = __builtin_fmaf(a1, a1, quad_swap1(sq1));
                const float y = __builtin_amdgcn_sqrtf(part == 0 ? m0 : m1);
                const float ys = quad_swap1(y);
                acc[t][nt][r] = (part == 0 ? y : m0) * s_lane;
                acc[t][nt][r + 1] = (part == 0 ? ys : m1) * s_lane;
            }
        }
    }
    const int xm = P.nx - 1, ym = P.ny - 1;
    const int sxz = P.ny * P.nz;
    auto readout = [&](auto full_c) {
        constexpr bool FULL4 = decltype(full_c)::value != 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            const int4 tq = *reinterpret_cast<const int4*>(targets + ((size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt * MFMA_COLS + (c16 >> 1)) * 4);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int code = want ? (q == 0 ? tq.x : q == 1 ? tq.y : q == 2 ? tq.z : tq.w) : -1;
                if (code < 0) continue;
                const unsigned m = (unsigned)code & 3u;
                const bool fx = (MX == 2) && (m & 1u), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1u));
                unsigned fxm = fx ? 0xFFFFFFFFu : 0u, fym = fy ? 0xFFFFFFFFu : 0u;
                asm volatile("" : "+v"(fxm), "+v"(fym));      // (opaque: kept as masks -- one v_and per term instead of a move and a select)
                float* const base = vol + (long long)(code >> 2) * P.vox + kz;
#pragma unroll
                for (int t = 0; t < C4_MT; ++t) {
                    if (t >= ntile) continue;
                    const int pos = wave + COS_NW * t;
                    const int kx = (pos * ky_magic) >> 16, ky = pos - kx * KY;
                    const int i = ibase + 2 * P.mx * kx, j = jbase + P.my * ky;      // wave-uniform (scalar ALU)
                    const unsigned o00 = (unsigned)(i * sxz + j * P.nz);
                    const unsigned DX = (unsigned)((xm - 2 * i) * sxz), DY = (unsigned)((ym - 2 * j) * P.nz);
                    float* dst = base + (o00 + (fxm & DX) + (fym & DY));
                    if constexpr (FULL4) *reinterpret_cast<float4*>(dst) = make_float4(acc[t][nt][0], acc[t][nt][1], acc[t][nt][2], acc[t][nt][3]);
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (kz + e < P.nz) dst[e] = acc[t][nt][e];
                    }
                }
            }
        }
    };
    if ((P.nz & 3) == 0) readout(IntC<1>{}); else readout(IntC<0>{});
}

}  // namespace olx

using namespace olx;

template <int MX, int MY>
static void launch_cosetp4(olx_ctx* c, float* pm) {
    const CosetParams& Q = c->cp;
    const bool clamp = c->clamp || c->lat.clamp;
    dim3 grid((unsigned)c->cp_nblocks, c->mp.n_tiles), blk(COS_NW * 64);
    if (clamp) hipLaunchKernelGGL((field_cosetp4_k<MX, MY, true>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_targets, c->d_cpblocks, Q);
    else hipLaunchKernelGGL((field_cosetp4_k<MX, MY, false>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_targets, c->d_cpblocks, Q);
}

void olx_launch_cosetp4(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) launch_cosetp4<2, 2>(c, pm);
    else if (c->mx == 2) launch_cosetp4<2, 1>(c, pm);
    else if (c->my == 2) launch_cosetp4<1, 2>(c, pm);
    else launch_cosetp4<1, 1>(c, pm);
}
#endif  // OLX_AB_VARIANTS
