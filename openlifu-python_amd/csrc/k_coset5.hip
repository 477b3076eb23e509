// kernel 2g, 32 x 32 form (field_cosetp32_k): kernel 2g with v_mfma_f32_32x32x16_f16 -- TWO positions x 16 planes x all 32 columns per matrix instruction
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#ifdef OLX_AB_VARIANTS   // measured-slower A/B form: compiled only into the developer library (build.py -DOLX_AB_VARIANTS), never into libolx.so
#include <algorithm>
#include <cstdlib>
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// Kernel 2g issues v_mfma_f32_16x16x32_f16: one position x 16 planes (rows) x 16 of the launch tile's 32 output columns, twice
// (NT = 2) per geometry fragment.  Under this workload the chip sits at its power cap and the matrix pipe sustains 30 % more
// FLOP/s in the 32 x 32 x 16 shape (profiles/ubench_mfma_shapes_r01.txt: half the operand traffic per flop).  Same block
// footprint, same geometry tables (26-row pair tables, plane stride 364 words, row stride 14), same steering-fragment staging
// and operands, same fp16 hi/lo (or e4m3-correction) arithmetic as field_cosetp_k; what changes is the tile:
//   * rows  = (position of a PAIR, plane): row = 16 p2 + plane.  Wave w takes the position pairs w', w' + 8, ... (<= 3 tiles;
//     w' = w rotated by the block index, so that the wave with one tile more is not always on the same SIMD).
//   * cols  = all 32 output columns: col = 16 nt + c16 (c16 = 2 o + re|im as before).
//   * K     = 16 = 4 x 2 elements x (re, im): lane (row, h) holds elements (aa = 0..3, bb = 2 kb2 + h) of the old 4 x 4 K-step --
//     the same four consecutive table words as before, from the table row of bb; two instructions (kb2 = 0, 1) per old K-step.
//     B operand: lane (col, h) needs what lane (c16, g = 2 kb2 + h) of column tile nt held -- read from the SAME LDS stage with
//     one ds_read_b128 (16 consecutive lanes -> 16 consecutive 16-byte slots: conflict-free).
//   * fp8 corrections: v_mfma_scale_f32_32x32x64_f8f6f4, lane (row, h) = the 32 operand bytes lane (row, g = 2 kb2 + h) held
//     (elements aa = 0..3 of the K-steps ka = 0, 1): one instruction per (kb, kb2).
//   * D: lane (col, h) holds rows 8 j + 4 h .. + 3 (j = 0..3): planes 4 h + 8 (j & 1) .. + 3 of position p2 = j >> 1 -- four
//     16-byte runs per tile; |p| / intensity in place with the same quad swap; stores straight from the accumulators.
// MEASURED (round 3, same box, alternating runs; profiles/r03_cosetp32_ab.txt): SLOWER than the 16 x 16 x 32 form.  Headline shard
// (33 / 30 / 22 / 20 positions per block part): 0.573 vs 0.446 ms (fp16 corrections), 0.484 vs 0.384 (fp8) -- 17 position pairs
// over 8 waves leave one wave with 3 tiles = 6 positions where the 16 x 16 form's busiest wave has 5, and the block's K phase
// lasts as long as its busiest wave.  On a shape where both forms are balanced (192^3: 16 positions = 8 pairs per block part)
// it is still 5 % slower (0.227 vs 0.216, 0.198 vs 0.187): a wave's matrix instructions all chain through its 1 - 3 accumulators,
// and the table / epilogue phases the kernel spends half its time in are unchanged.  Kept in the developer library
// (OLX_FIELD_VARIANT=cosetp32) as the evidence for DESIGN.md 5.4; the planner never selects it.
// The sum over elements is associated differently from field_cosetp_k (K = 16 instead of 32 per instruction), so results agree
// with it to rounding (~1e-7 of the maximum), not to the bit; the parity gate is the fp64 oracle's.
// ------------------------------------------------------------------------------------
constexpr int C5_TW = 14, C5_TROWS = 26, C5_ROW0 = 15, C5_PSZ = 364, C5_UW = 12;     // = kernel 2g's table geometry
constexpr int C5_MT = 3;                            // tiles (position pairs) per wave: ceil(ceil(40 / 2) / 8)
typedef float floatx16_t __attribute__((ext_vector_type(16)));

template <int MX, int MY, bool CLAMP, bool FP8>
__global__ __launch_bounds__(COS_NW * 64, 4) void field_cosetp32_k(
    const uint4* __restrict__ bfrag, float* __restrict__ pmag, float* __restrict__ inten,
    const int* __restrict__ targets /*[tiles][32 columns][4]: focus * 4 + mirror image, -1 = none*/,
    const CosetBlock* __restrict__ blocks /*[gridDim.x]*/, const CosetParams P) {
    constexpr int NT = 2, THREADS = COS_NW * 64;
    constexpr int RPR = 64 / C5_UW, NROUND = (C5_TROWS + RPR - 1) / RPR;       // 5 table rows per generation round, 6 rounds
    constexpr int B_BYTES = 2 * 4 * NT * 2 * 64 * 16;                           // two super-blocks of steering fragments
    constexpr int T_WORDS = COS_ZB * C5_PSZ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES + 2 * T_WORDS * 4 + 64];
    unsigned* const s_hi = reinterpret_cast<unsigned*>(smem + B_BYTES);
    unsigned* const s_lo = s_hi + T_WORDS;
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const CosetBlock BK = blocks[blockIdx.x];
    const int npos = BK.npos, KY = BK.KY, ky_magic = BK.ky_magic;
    if (npos <= 0) return;                              // block-uniform
    const int ibase = BK.ibase, jbase = BK.jbase, k0 = BK.k0;
    const int npair = (npos + 1) >> 1;
    const int wrot = (wave + (int)(blockIdx.x & 7u)) & 7;                       // which pairs this wave takes: wrot, wrot + 8, ...
    const int ntile = __builtin_amdgcn_readfirstlane((npair - wrot + COS_NW - 1) / COS_NW);      // wave-uniform, <= C5_MT
    // fragment read offset [words] of a row for K-step (0, 0, kb2 = 0): per-lane part (plane, k-group h) + the row's position (p2 of the
    // tile's pair: the two halves of a 32-lane group read the tables at the two positions' offsets)
    int toff[C5_MT];
    {
        int lane_o = lane;
        asm volatile("" : "+v"(lane_o));
        const int p2 = (lane_o >> 4) & 1;
#pragma unroll
        for (int t = 0; t < C5_MT; ++t) {
            const int pos = min(2 * (wrot + COS_NW * t) + p2, npos - 1);         // (an odd count's last pair computes its position twice)
            const int kx = (pos * ky_magic) >> 16, ky = pos - kx * KY;           // pos / KY, exact for pos <= 40 (host checks)
            toff[t] = (ky + C5_ROW0) * C5_TW + (C5_UW - 8 - 2 * kx);
        }
    }
    floatx16_t acc[C5_MT];
#pragma unroll
    for (int t = 0; t < C5_MT; ++t)
#pragma unroll
        for (int i = 0; i < 16; ++i) acc[t][i] = 0.f;
    const int nsbp = P.nsbp;                    // even: chunks = table pairs never straddle sa
    const int n_sb = P.nsa * nsbp;
    constexpr int CHUNK_U4 = 2 * 4 * NT * 128, PRE = CHUNK_U4 / THREADS;
    static_assert(CHUNK_U4 % THREADS == 0, "chunk must split evenly over the block");
    uint4 pre[PRE];
    const uint4* const bsrc = bfrag + (size_t)tile * n_sb * (4 * NT * 128);
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int idx = tid + q * THREADS;
        pre[q] = idx < n_sb * 4 * NT * 128 ? bsrc[idx] : make_uint4(0, 0, 0, 0);
    }
    for (int sb0 = 0; sb0 < n_sb; sb0 += 2) {
        const int sa = sb0 / nsbp, sbb0 = sb0 - sa * nsbp;       // the pair (sa, sbb0), (sa, sbb0 + 1)
        if (sb0 > 0) __syncthreads();                             // previous pair consumed: steering stage and tables are free
        // ---- G tables of planes 2 wave, 2 wave + 1: 26 rows x 12 offsets, shared by the pair's two super-blocks (as kernel 2g)
        if constexpr (FP8) __builtin_amdgcn_s_setprio(1);
        if (k0 + wave * COS_P < P.nz) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int wl = lane_o / C5_UW, ui = lane_o - C5_UW * wl;
            const bool gen_lane = wl < RPR;
            const int Ulane = ibase + P.x_begin + P.ux0 + P.mx * (ui - 7);
            const int Wlane = jbase + P.uy0 + P.my * (wl - C5_ROW0);
            const int tw_off = (wave * COS_P) * C5_PSZ + wl * C5_TW + (C5_UW - 1 - ui);   // + z PSZ + RPR r TW
            float dz2[COS_P];
#pragma unroll
            for (int z = 0; z < COS_P; ++z) {
                const float dz = (float)(k0 + wave * COS_P + z) * P.hz - P.flat_ez;
                dz2[z] = dz * dz;
            }
            const float U = (float)(Ulane - 8 * P.mx * sa);
            const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
            const float dx2 = dx * dx;
            const int Wsb = Wlane - 8 * P.my * sbb0;
#pragma unroll 2
            for (int r = 0; r < NROUND; ++r) {
                const bool row_ok = gen_lane && RPR * r + wl < C5_TROWS;  // the last round may run past the table
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
                    half2_t hi;
                    if constexpr (FP8) hi = __builtin_convertvector(float2_t{gr, gi}, half2_t);      // to nearest: |lo| <= half an ulp
                    else hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
                    float lr, li;
                    const unsigned hw = __builtin_bit_cast(unsigned, hi);
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
                    unsigned lo_word;
                    if constexpr (FP8) {             // e4m3 bytes [lo re, lo im | hi re, hi im], |.| <= 256 (448 overflows to NaN)
                        short2_t w;                  // (both halves are written below)
                        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lr, li, 1.0f / COS_F8_LO, false);
                        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, gr, gi, 1.0f / COS_F8_HI, true);
                        lo_word = __builtin_bit_cast(unsigned, w);
                    } else {
                        lo_word = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lr, li));
                    }
                    if (row_ok) {
                        const int o = z * C5_PSZ + tw_off + RPR * r * C5_TW;
                        s_hi[o] = __builtin_bit_cast(unsigned, hi);
                        s_lo[o] = lo_word;
                    }
                }
            }
        }
        if constexpr (FP8) __builtin_amdgcn_s_setprio(0);
        // this pair's steering fragments (requested one pair ahead; the first ones arrive behind the table generation above)
#pragma unroll
        for (int q = 0; q < PRE; ++q) reinterpret_cast<uint4*>(smem)[tid + q * THREADS] = pre[q];
        __syncthreads();
        {   // next pair's fragments: in flight during the K-steps, drained by the next barrier
            const int nxt = (sb0 + 2) * 4 * NT * 128, lim = n_sb * 4 * NT * 128;
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int idx = nxt + tid + q * THREADS;
                if (idx < lim) pre[q] = bsrc[idx];
            }
        }
        // per-lane operand addresses, formed here (opaque) so that nothing of them is live across the table generation
        int lane_k = lane;
        asm volatile("" : "+v"(lane_k));
        const int hk = lane_k >> 5;
        const int a_off = (lane_k & 15) * C5_PSZ - hk * C5_TW;                                    // plane, k-group h
        // steering stage [ks][nt][part][64] uint4: lane (col = 16 nt + c16, h) of step kb2 <- old lane c16 + 16 (2 kb2 + h) of column tile nt
        const uint4* const b_lane = reinterpret_cast<const uint4*>(smem) + ((lane_k >> 4) & 1) * 128 + (lane_k & 15) + 16 * hk;
#pragma unroll                                          // (unrolled: the pair position becomes part of the immediate table offsets)
        for (int sl = 0; sl < 2; ++sl) {
            if (sbb0 + sl >= P.nsb) break;              // padding super-block of an odd count: zero weights, nothing to do
            if constexpr (FP8) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {
#pragma unroll
                    for (int kb2 = 0; kb2 < 2; ++kb2) {      // both K-steps ka = 0, 1 of row pair bb = 2 kb2 + h: two fp16 hi*hi products + ONE fp8 product
                        Half8Bits bh[2];
                        intx8_t b8;
#pragma unroll
                        for (int ka = 0; ka < 2; ++ka) {
                            const uint4* bp = b_lane + (sl * 4 + 2 * kb + ka) * (NT * 128) + 32 * kb2;
                            bh[ka].u = bp[0];
                            const uint4 q = bp[64];
                            b8[4 * ka + 0] = (int)q.x; b8[4 * ka + 1] = (int)q.y; b8[4 * ka + 2] = (int)q.z; b8[4 * ka + 3] = (int)q.w;
                        }
#pragma unroll
                        for (int t = 0; t < C5_MT; ++t) {
                            if (t >= ntile) continue;            // wave-uniform
                            Half8Bits ah[2];
                            intx8_t a8;
                            const int ro = a_off + toff[t];
#pragma unroll
                            for (int ka = 0; ka < 2; ++ka) {
                                const int kso = 4 * ka - (4 * kb + 2 * kb2 + 8 * sl) * C5_TW;
                                const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(s_hi + ro + kso);
                                const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(s_lo + ro + kso);
                                const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                ah[ka].w[0] = (unsigned)h0; ah[ka].w[1] = (unsigned)(h0 >> 32); ah[ka].w[2] = (unsigned)h1; ah[ka].w[3] = (unsigned)(h1 >> 32);
                                a8[4 * ka + 0] = (int)(unsigned)l0; a8[4 * ka + 1] = (int)(unsigned)(l0 >> 32);
                                a8[4 * ka + 2] = (int)(unsigned)l1; a8[4 * ka + 3] = (int)(unsigned)(l1 >> 32);
                            }
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[0].h, bh[0].h, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah[1].h, bh[1].h, acc[t], 0, 0, 0);
                            // E8M0 scales undo the operand scaling: 2^(128 - 127) * COS_F8_LO * COS_F8_HI = 1
                            acc[t] = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(a8, b8, acc[t], 0, 0, 0, 128, 0, 127);
                        }
                    }
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < 4; ++ks) {         // unrolled: the K-step's table offset becomes an immediate
                    const int ka = ks & 1, kb = ks >> 1;
#pragma unroll
                    for (int kb2 = 0; kb2 < 2; ++kb2) {
                        Half8Bits bh, bl;
                        const uint4* bp = b_lane + (sl * 4 + ks) * (NT * 128) + 32 * kb2;
                        bh.u = bp[0];
                        bl.u = bp[64];
                        const int kso = 4 * ka - (4 * kb + 2 * kb2 + 8 * sl) * C5_TW;
#pragma unroll
                        for (int t = 0; t < C5_MT; ++t) {        // (products outermost -- consecutive instructions to different accumulators -- measured 8 % slower)
                            if (t >= ntile) continue;            // wave-uniform
                            Half8Bits ah, al;
                            const int ro = a_off + toff[t];
                            const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(s_hi + ro + kso);
                            const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(s_lo + ro + kso);
                            const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            ah.w[0] = (unsigned)h0; ah.w[1] = (unsigned)(h0 >> 32); ah.w[2] = (unsigned)h1; ah.w[3] = (unsigned)(h1 >> 32);
                            al.w[0] = (unsigned)l0; al.w[1] = (unsigned)(l0 >> 32); al.w[2] = (unsigned)l1; al.w[3] = (unsigned)(l1 >> 32);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bh.h, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(al.h, bh.h, acc[t], 0, 0, 0);
                            acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(ah.h, bl.h, acc[t], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- epilogue, straight from the accumulators.  Lane (col = 16 nt + c16, h): acc[t][4 j .. 4 j + 3] = planes k0 + 8 (j & 1) + 4 h .. + 3
    // of the pair's position j >> 1, column c16 = (o, re | im) of column tile nt.  |p| lane (part 0) and its partner (part 1, the
    // intensity lane) exchange squares with one quad swap, one square root per two rows (as kernel 2g).
    int lane_e = lane;
    asm volatile("" : "+v"(lane_e));
    const int c16 = lane_e & 15, part = c16 & 1, nt_l = (lane_e >> 4) & 1, he = lane_e >> 5;
    const float s_lane = part == 0 ? P.out_scale : P.out_scale * P.out_scale * P.inten_scale;
    float* const vol = part ? inten : pmag;
    const bool want = (P.flags & (part ? 2u : 1u)) != 0;
    const int4 tq = *reinterpret_cast<const int4*>(targets + ((size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt_l * MFMA_COLS + (c16 >> 1)) * 4);
#pragma unroll
    for (int t = 0; t < C5_MT; ++t) {
        if (t >= ntile) continue;
#pragma unroll
        for (int r = 0; r < 16; r += 2) {
            const float a0 = acc[t][r], a1 = acc[t][r + 1];
            const float sq0 = a0 * a0, sq1 = a1 * a1;
            const float m0 = __builtin_fmaf(a0, a0, quad_swap1(sq0)), m1 = __builtin_fmaf(a1, a1, quad_swap1(sq1));
            const float y = __builtin_amdgcn_sqrtf(part == 0 ? m0 : m1);
            const float ys = quad_swap1(y);
            acc[t][r] = (part == 0 ? y : m0) * s_lane;
            acc[t][r + 1] = (part == 0 ? ys : m1) * s_lane;
        }
    }
    const int xm = P.nx - 1, ym = P.ny - 1;
    const int sxz = P.ny * P.nz;
    auto readout = [&](auto full_c) {
        constexpr bool FULL4 = decltype(full_c)::value != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int code = want ? (q == 0 ? tq.x : q == 1 ? tq.y : q == 2 ? tq.z : tq.w) : -1;
            if (code < 0) continue;
            const unsigned m = (unsigned)code & 3u;
            const bool fx = (MX == 2) && (m & 1u), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1u));
            unsigned fxm = fx ? 0xFFFFFFFFu : 0u, fym = fy ? 0xFFFFFFFFu : 0u;
            asm volatile("" : "+v"(fxm), "+v"(fym));      // (opaque: kept as masks -- one v_and per term instead of a move and a select)
            float* const base = vol + (long long)(code >> 2) * P.vox + k0 + 4 * he;
#pragma unroll
            for (int t = 0; t < C5_MT; ++t) {
                if (t >= ntile) continue;
#pragma unroll
                for (int p2 = 0; p2 < 2; ++p2) {
                    const int pos = 2 * (wrot + COS_NW * t) + p2;
                    if (pos >= npos) continue;                                       // wave-uniform: the dummy half of an odd count's last pair
                    const int kx = (pos * ky_magic) >> 16, ky = pos - kx * KY;
                    const int i = ibase + 2 * P.mx * kx, j = jbase + P.my * ky;      // wave-uniform (scalar ALU)
                    const unsigned o00 = (unsigned)(i * sxz + j * P.nz);
                    const unsigned DX = (unsigned)((xm - 2 * i) * sxz), DY = (unsigned)((ym - 2 * j) * P.nz);
                    const unsigned off = o00 + (fxm & DX) + (fym & DY);
#pragma unroll
                    for (int zh = 0; zh < 2; ++zh) {
                        const int jj = 2 * p2 + zh, kz = k0 + 8 * zh + 4 * he;
                        float* dst = base + off + 8 * zh;
                        if constexpr (FULL4) {
                            if (kz < P.nz) *reinterpret_cast<float4*>(dst) = make_float4(acc[t][4 * jj], acc[t][4 * jj + 1], acc[t][4 * jj + 2], acc[t][4 * jj + 3]);
                        } else {
#pragma unroll
                            for (int e = 0; e < 4; ++e) if (kz + e < P.nz) dst[e] = acc[t][4 * jj + e];
                        }
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
static void launch_cosetp32(olx_ctx* c, float* pm) {
    const CosetParams& Q = c->cp;
    const bool clamp = c->clamp || c->lat.clamp;
    dim3 grid((unsigned)c->cp_nblocks, c->mp.n_tiles), blk(COS_NW * 64);
#define OLX_CP(CL, F8) hipLaunchKernelGGL((field_cosetp32_k<MX, MY, CL, F8>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_targets, c->d_cpblocks, Q)
    if (c->fp8corr) { if (clamp) OLX_CP(true, true); else OLX_CP(false, true); }
    else            { if (clamp) OLX_CP(true, false); else OLX_CP(false, false); }
#undef OLX_CP
}

void olx_launch_cosetp32(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) launch_cosetp32<2, 2>(c, pm);
    else if (c->mx == 2) launch_cosetp32<2, 1>(c, pm);
    else if (c->my == 2) launch_cosetp32<1, 2>(c, pm);
    else launch_cosetp32<1, 1>(c, pm);
}
#endif  // OLX_AB_VARIANTS
