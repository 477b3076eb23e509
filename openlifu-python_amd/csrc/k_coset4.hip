// kernel 2r (field_cosetr_k): kernel 2g as ONE persistent block per CU -- the next tables are generated INSIDE the K-steps
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#ifdef OLX_AB_VARIANTS   // measured-slower A/B form: compiled only into the developer library (build.py -DOLX_AB_VARIANTS), never into libolx.so
#include <algorithm>
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// Kernel 2g's phases -- table generation (vector ALU), K-steps (matrix pipe), |p| + stores -- alternate behind block barriers and
// overlap only through the second block of the CU: the matrix pipe is busy 35 %, the vector issue port 50 %, and a slot idles
// ~13 % of the time at block turn-over (tools/cutrace_cosetp.py).  Here the same tiles, tables and arithmetic run as one
// persistent 8-wave block per CU (two waves per SIMD, 256 VGPRs each) that walks the block records blockIdx.x, + gridDim.x, ...:
//   * TWO table buffers: while the K-steps of pair p read buffer p & 1, every wave evaluates its two planes of the NEXT pair's
//     tables (the next record's first pair behind the last one) into the other buffer, one 5-row round after a tile's matrix
//     instructions -- the vector ALU works in the shadow of the matrix pipe.  One LDS-only barrier per pair.
//   * the steering fragments of all (<= 4) super-blocks stay resident in LDS (64 KB), staged once per block;
//   * the stores of a record drain behind the next record's K-steps (the barrier does not wait for vmcnt).
// LDS: 64 KB steering + 2 x 46.6 KB tables = 157 KB.  Shape: NT = 2, n_sb <= 4 (arrays up to 16 x 16 elements), one launch tile.
// ------------------------------------------------------------------------------------
constexpr int CR_NW = 8;                           // waves per block
constexpr int CR_P = 2;                            // planes per wave (table generation)
constexpr int CR_ZB = CR_NW * CR_P;                // planes per block
constexpr int CR_TW = 14;                          // words per table row (12 in use)
constexpr int CR_TROWS = 26, CR_ROW0 = 15;         // pair table rows; row of offset wd = 0
constexpr int CR_PSZ = 364;                        // words per plane table; PSZ / 2 = 22 (mod 32): see kernel 2g
constexpr int CR_UW = 12;                          // table columns: ud = 2 kx - a in [-7, 4]
constexpr int CR_MT = 5;                           // tiles (positions) per wave: ceil(33 / 8)
constexpr int CR_MAXSB = 4;                        // super-blocks whose steering fragments stay resident

template <int MX, int MY, bool CLAMP, bool FP8>
__global__ __launch_bounds__(CR_NW * 64, 2) void field_cosetr_k(
    const uint4* __restrict__ bfrag, float* __restrict__ pmag, float* __restrict__ inten,
    const int* __restrict__ targets /*[tiles][32 columns][4]: focus * 4 + mirror image, -1 = none*/,
    const CosetBlock* __restrict__ blocks /*[n_items], all with npos > 0*/, const CosetParams P, const int n_items) {
    constexpr int NT = 2, THREADS = CR_NW * 64;
    constexpr int RPR = 64 / CR_UW, NROUND = (CR_TROWS + RPR - 1) / RPR;       // 5 table rows per generation round, 6 rounds
    constexpr int SB_U4 = 4 * NT * 128;                                         // uint4 per super-block of steering fragments (16 KB)
    constexpr int B_BYTES = CR_MAXSB * SB_U4 * 16;
    constexpr int T_WORDS = CR_ZB * CR_PSZ;                                     // one part (hi or lo) of one buffer
    __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES + 4 * T_WORDS * 4 + 64];
    typedef uint4 (*BArr)[NT][2][64];
    BArr s_B = reinterpret_cast<BArr>(smem);                                    // [sb * 4 + K-step][nt][hi | lo][lane]
    unsigned* const s_tab = reinterpret_cast<unsigned*>(smem + B_BYTES);        // buffer b: hi at b * 2 T_WORDS, lo at + T_WORDS
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p16 = lane & 15;
    const int nsbp = P.nsbp;                    // even: table pairs never straddle sa
    const int n_sb = P.nsa * nsbp;              // <= CR_MAXSB (host)
    const int n_pair = n_sb >> 1;
    // ---- steering fragments: all super-blocks, once
    for (int idx = tid; idx < n_sb * SB_U4; idx += THREADS) reinterpret_cast<uint4*>(smem)[idx] = bfrag[idx];

    // ---- table generation (this wave: planes 2 wave, 2 wave + 1 of the record's 16): lane -> (wl = lane / UW < RPR, ui = lane % UW);
    // round r: rows RPR r + wl.  `TG` = the per-lane terms of one (record, pair); a round = 2 evaluations per lane.
    struct TG { float dx2, dz2[CR_P]; int Wsb, tw_off; bool gen_lane, live; };
    auto tg_setup = [&](const CosetBlock& R, int pair, int buf) {
        TG t;
        const int sb0 = 2 * pair, sa = sb0 / nsbp, sbb0 = sb0 - sa * nsbp;
        const int wl = lane / CR_UW, ui = lane - CR_UW * wl;
        t.gen_lane = wl < RPR;
        t.live = R.k0 + wave * CR_P < P.nz;
        const int Ulane = R.ibase + P.x_begin + P.ux0 + P.mx * (ui - 7);
        const int Wlane = R.jbase + P.uy0 + P.my * (wl - CR_ROW0);
        t.tw_off = buf * 2 * T_WORDS + (wave * CR_P) * CR_PSZ + wl * CR_TW + (CR_UW - 1 - ui);   // + z PSZ + RPR r TW
#pragma unroll
        for (int z = 0; z < CR_P; ++z) {
            const float dz = (float)(R.k0 + wave * CR_P + z) * P.hz - P.flat_ez;
            t.dz2[z] = dz * dz;
        }
        const float U = (float)(Ulane - 8 * P.mx * sa);
        const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
        t.dx2 = dx * dx;
        t.Wsb = Wlane - 8 * P.my * sbb0;
        return t;
    };
    auto tg_round = [&](const TG& t, int r) {
        if (!t.live) return;                                      // wave-uniform
        const bool row_ok = t.gen_lane && RPR * r + (lane / CR_UW) < CR_TROWS;  // the last round may run past the table
        const float W = (float)(t.Wsb + RPR * P.my * r);
        const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
        const float r2 = fmaf(dy, dy, t.dx2);
#pragma unroll
        for (int z = 0; z < CR_P; ++z) {
            float d2 = r2 + t.dz2[z];
            if (CLAMP) d2 = fmaxf(d2, P.dmin2);
            const float ri = __builtin_amdgcn_rsqf(d2);
            const float ph = d2 * ri;
            const float rs = ri * P.g_scale;
            const float gr = rs * __builtin_amdgcn_cosf(ph);
            const float gi = rs * __builtin_amdgcn_sinf(ph);
            half2_t hi;
            if constexpr (FP8) hi = __builtin_convertvector(float2_t{gr, gi}, half2_t);      // to nearest: |lo| <= half an ulp
            else hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
            float lr, li;                      // lo = g - (float)hi: one mixed-precision fma per component
            const unsigned hw = __builtin_bit_cast(unsigned, hi);
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
            asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
            unsigned lo_word;
            if constexpr (FP8) {               // e4m3 bytes [lo re, lo im | hi re, hi im]; the scale operand DIVIDES (tools/probe/cvt_scale_probe.hip)
                short2_t w;
                w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lr, li, 1.0f / COS_F8_LO, false);
                w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, gr, gi, 1.0f / COS_F8_HI, true);
                lo_word = __builtin_bit_cast(unsigned, w);
            } else {
                lo_word = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lr, li));
            }
            if (row_ok) {
                const int o = z * CR_PSZ + t.tw_off + RPR * r * CR_TW;
                s_tab[o] = __builtin_bit_cast(unsigned, hi);
                s_tab[o + T_WORDS] = lo_word;
            }
        }
    };

    // ---- store side: per-lane constants of the 8 store-target slots of this lane's column (record-invariant; 256 registers per wave)
    const int c16 = lane & 15, part = c16 & 1;
    const float s_lane = part == 0 ? P.out_scale : P.out_scale * P.out_scale * P.inten_scale;
    float* const vol = part ? inten : pmag;
    const bool want = (P.flags & (part ? 2u : 1u)) != 0;
    unsigned tgt[NT][2];                          // store targets of this lane's column, two 16-bit codes (focus * 4 + mirror image, 0xFFFF = none) per register
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        const int4 tq = *reinterpret_cast<const int4*>(targets + ((size_t)nt * MFMA_COLS + (c16 >> 1)) * 4);
        tgt[nt][0] = want ? (((unsigned)tq.x & 0xFFFFu) | ((unsigned)tq.y << 16)) : 0xFFFFFFFFu;
        tgt[nt][1] = want ? (((unsigned)tq.z & 0xFFFFu) | ((unsigned)tq.w << 16)) : 0xFFFFFFFFu;
    }
    const int xm = P.nx - 1, ym = P.ny - 1;
    const int sxz = P.ny * P.nz;
    struct RP { int ibase, jbase, k0, KY, ky_magic, ntile; };      // a record's wave-uniform terms (scalar registers)
    // |p| / intensity and stores of ONE tile of a finished record, straight from its accumulators (as kernel 2g's two passes, per tile)
    auto epi_tile = [&](floatx4_t (&A)[NT], const RP& R, int t) {
        if (t >= R.ntile) return;                                   // wave-uniform
        const int kz = R.k0 + 4 * g;
        if (kz >= P.nz) return;
        const int pos = wave + CR_NW * t;
        const int kx = (pos * R.ky_magic) >> 16, ky = pos - kx * R.KY;
        const int i = R.ibase + 2 * P.mx * kx, j = R.jbase + P.my * ky;      // wave-uniform (scalar ALU)
        const unsigned o00 = (unsigned)(i * sxz + j * P.nz) + (unsigned)kz;
        const unsigned DX = (unsigned)((xm - 2 * i) * sxz), DY = (unsigned)((ym - 2 * j) * P.nz);
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            float v[4];
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float a0 = A[nt][r], a1 = A[nt][r + 1];
                const float sq0 = a0 * a0, sq1 = a1 * a1;
                const float m0 = __builtin_fmaf(a0, a0, quad_swap1(sq0)), m1 = __builtin_fmaf(a1, a1, quad_swap1(sq1));   // (pinned: own square unrounded, partner's rounded)
                const float y = __builtin_amdgcn_sqrtf(part == 0 ? m0 : m1);
                const float ys = quad_swap1(y);
                v[r] = (part == 0 ? y : m0) * s_lane;
                v[r + 1] = (part == 0 ? ys : m1) * s_lane;
            }
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const unsigned code = (tgt[nt][q >> 1] >> (16 * (q & 1))) & 0xFFFFu;
                if (code == 0xFFFFu) continue;
                const unsigned m = code & 3u;
                const bool fx = (MX == 2) && (m & 1u), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1u));
                const unsigned off = (code >> 2) * (unsigned)P.vox + o00 + ((fx ? DX : 0u) + (fy ? DY : 0u));     // (host: foci * voxels < 2^32)
                *reinterpret_cast<float4*>(vol + off) = make_float4(v[0], v[1], v[2], v[3]);                      // (host: nz % 4 == 0)
            }
        }
    };

    int item = blockIdx.x;                      // < n_items (host: grid <= n_items)
    CosetBlock rec = blocks[item];
    {   // first tables of the first record: nothing to hide them behind
        const TG t0 = tg_setup(rec, 0, 0);
#pragma unroll 2
        for (int r = 0; r < NROUND; ++r) tg_round(t0, r);
    }
    __syncthreads();
    int pb = 0;                                 // table buffer of the current pair
    const int lane_off = p16 * CR_PSZ - g * CR_TW;
    floatx4_t acc[2][CR_MT][NT];                // two accumulator sets: the finished record's results are stored behind the next record's K-steps
    RP prev{0, 0, 0, 1, 0, 0};
    bool prev_valid = false, has_next = true;
    // one record: K-steps into accumulator set PAR, next tables and (pair 0) the previous record's |p| + stores in their shadow
    auto record = [&](auto par_c) {
        constexpr int PAR = decltype(par_c)::value;
        const int npos = rec.npos;
        RP cur;
        cur.ibase = rec.ibase; cur.jbase = rec.jbase; cur.k0 = rec.k0; cur.KY = rec.KY; cur.ky_magic = rec.ky_magic;
        cur.ntile = __builtin_amdgcn_readfirstlane((npos - wave + CR_NW - 1) / CR_NW);   // this wave's positions: wave, wave + 8, ... (<= CR_MT)
        const int ntile = cur.ntile;
        const int item_n = item + (int)gridDim.x;
        has_next = item_n < n_items;
        const CosetBlock rec_n = blocks[has_next ? item_n : item];
        int toff[CR_MT];
#pragma unroll
        for (int t = 0; t < CR_MT; ++t) {
            const int pos = min(wave + CR_NW * t, npos - 1);
            const int kx = (pos * cur.ky_magic) >> 16, ky = pos - kx * cur.KY;        // scalar: pos / KY, exact for pos <= 40 (host checks)
            toff[t] = (ky + CR_ROW0) * CR_TW + (CR_UW - 8 - 2 * kx);
        }
#pragma unroll
        for (int t = 0; t < CR_MT; ++t)
#pragma unroll
            for (int nt = 0; nt < NT; ++nt) acc[PAR][t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};
        OLX_STAMP(0);
        for (int pr = 0; pr < n_pair; ++pr) {
            const int sb0 = 2 * pr, sa = sb0 / nsbp, sbb0 = sb0 - sa * nsbp;
            // the tables to evaluate behind this pair's matrix instructions: the record's next pair, else the next record's first
            const bool last_pair = pr + 1 >= n_pair;
            const bool gen = !last_pair || has_next;
            const TG tg = tg_setup(last_pair ? rec_n : rec, last_pair ? 0 : pr + 1, pb ^ 1);
            const int tbase = pb * 2 * T_WORDS;
            const bool two = sbb0 + 1 < P.nsb;       // (else: padding super-block of an odd count -- zero weights, nothing to do)
            const bool epi = pr == 0 && prev_valid;  // the previous record's results leave behind this pair's first super-block
#pragma unroll
            for (int sl = 0; sl < 2; ++sl) {
                if (sl == 1 && !two) break;
                const int sbq = (sb0 + sl) * 4;
                if constexpr (FP8) {
#pragma unroll
                    for (int kb = 0; kb < 2; ++kb) {         // K-step pairs (ka = 0, 1): two fp16 hi*hi products + ONE fp8 product
                        Half8Bits bh[2][NT];
                        intx8_t b8[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                            for (int ka = 0; ka < 2; ++ka) {
                                bh[ka][nt].u = s_B[sbq + 2 * kb + ka][nt][0][lane];
                                const uint4 q = s_B[sbq + 2 * kb + ka][nt][1][lane];
                                b8[nt][4 * ka + 0] = (int)q.x; b8[nt][4 * ka + 1] = (int)q.y; b8[nt][4 * ka + 2] = (int)q.z; b8[nt][4 * ka + 3] = (int)q.w;
                            }
                        }
                        Half8Bits ah[1][2];                  // [register set][ka]
                        intx8_t a8[1];
                        auto loadA = [&](int t, int set) {
                            const int ro = lane_off + toff[t] + tbase;
#pragma unroll
                            for (int ka = 0; ka < 2; ++ka) {
                                const int kso = 4 * ka - (4 * kb + 8 * sl) * CR_TW;   // the pair's second super-block reads 8 table rows lower
                                const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(s_tab + ro + kso);
                                const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(s_tab + T_WORDS + ro + kso);
                                const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                                ah[set][ka].w[0] = (unsigned)h0; ah[set][ka].w[1] = (unsigned)(h0 >> 32); ah[set][ka].w[2] = (unsigned)h1; ah[set][ka].w[3] = (unsigned)(h1 >> 32);
                                a8[set][4 * ka + 0] = (int)(unsigned)l0; a8[set][4 * ka + 1] = (int)(unsigned)(l0 >> 32);
                                a8[set][4 * ka + 2] = (int)(unsigned)l1; a8[set][4 * ka + 3] = (int)(unsigned)(l1 >> 32);
                            }
                        };
#pragma unroll
                        for (int t = 0; t < CR_MT; ++t) {
                            if (t < ntile) {                     // wave-uniform
                                loadA(t, 0);
                                const int s = 0;
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[PAR][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s][0].h, bh[0][nt].h, acc[PAR][t][nt], 0, 0, 0);
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[PAR][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s][1].h, bh[1][nt].h, acc[PAR][t][nt], 0, 0, 0);
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt)      // E8M0 scales undo the operand scaling: 2^(128 - 127) * COS_F8_LO * COS_F8_HI = 1
                                    acc[PAR][t][nt] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8[s], b8[nt], acc[PAR][t][nt], 0, 0, 0, 128, 0, 127);
                            }
                            // in the shadow of the tile's matrix instructions: a round of the next tables behind tiles 0, 1 of each K-step
                            // pair (rounds 0 - 3 with sl = 0, 4 - 5 with sl = 1); the previous record's tiles behind tiles 2 .. 4 (sl = 0)
                            if (t < 2) { const int r = (sl * 2 + kb) * 2 + t; if (r < NROUND && gen) tg_round(tg, r); }
                            else if (sl == 0) { const int e = kb * 3 + (t - 2); if (e < CR_MT && epi) epi_tile(acc[PAR ^ 1][e], prev, e); }
                        }
                    }
                } else {
#pragma unroll
                    for (int ks = 0; ks < 4; ++ks) {         // unrolled: the K-step's table offset becomes an immediate
                        const int ka = ks & 1, kb = ks >> 1;
                        Half8Bits bh[NT], bl[NT];
#pragma unroll
                        for (int nt = 0; nt < NT; ++nt) {
                            bh[nt].u = s_B[sbq + ks][nt][0][lane];
                            bl[nt].u = s_B[sbq + ks][nt][1][lane];
                        }
                        const int kso = 4 * ka - (4 * kb + 8 * sl) * CR_TW;
                        Half8Bits ah[1], al[1];
                        auto loadA = [&](int t, int set) {
                            const int ro = lane_off + toff[t] + tbase;
                            const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(s_tab + ro + kso);
                            const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(s_tab + T_WORDS + ro + kso);
                            const unsigned long long h0 = __hip_atomic_load(ph2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            const unsigned long long h1 = __hip_atomic_load(ph2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            const unsigned long long l0 = __hip_atomic_load(pl2, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            const unsigned long long l1 = __hip_atomic_load(pl2 + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                            ah[set].w[0] = (unsigned)h0; ah[set].w[1] = (unsigned)(h0 >> 32); ah[set].w[2] = (unsigned)h1; ah[set].w[3] = (unsigned)(h1 >> 32);
                            al[set].w[0] = (unsigned)l0; al[set].w[1] = (unsigned)(l0 >> 32); al[set].w[2] = (unsigned)l1; al[set].w[3] = (unsigned)(l1 >> 32);
                        };
#pragma unroll
                        for (int t = 0; t < CR_MT; ++t) {
                            if (t < ntile) {                     // wave-uniform
                                loadA(t, 0);
                                const int s = 0;
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[PAR][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s].h, bh[nt].h, acc[PAR][t][nt], 0, 0, 0);
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[PAR][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[s].h, bh[nt].h, acc[PAR][t][nt], 0, 0, 0);
#pragma unroll
                                for (int nt = 0; nt < NT; ++nt) acc[PAR][t][nt] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[s].h, bl[nt].h, acc[PAR][t][nt], 0, 0, 0);
                            }
                            // in the shadow: a round of the next tables behind tile 0 of each K-step (rounds 0 - 3 with sl = 0, 4 - 5 with
                            // sl = 1); the previous record's tiles behind tiles 1 .. 4 of K-step 0 and tile 1 of K-step 1 (sl = 0)
                            if (t == 0) { const int r = sl * 4 + ks; if (r < NROUND && gen) tg_round(tg, r); }
                            else if (sl == 0 && ks < 2) { const int e = ks * 4 + (t - 1); if (e < CR_MT && epi) epi_tile(acc[PAR ^ 1][e], prev, e); }
                        }
                    }
                }
            }
            if (!two && gen) { tg_round(tg, 4); tg_round(tg, 5); }    // (the rounds that ride on the skipped super-block)
            if (pr == 0) OLX_STAMP(1); else OLX_STAMP(3);
            lds_barrier();                   // every wave has left this pair's tables; the next pair's are complete.  LDS only: stores keep draining
            if (pr == 0) OLX_STAMP(2); else OLX_STAMP(4);
            pb ^= 1;
        }
        OLX_STAMP(5);
        prev = cur; prev_valid = true;
        item = item_n; rec = rec_n;
    };
    while (true) {
        record(IntC<0>{});
        if (!has_next) {
#pragma unroll
            for (int t = 0; t < CR_MT; ++t) epi_tile(acc[0][t], prev, t);      // the last record: nothing left to hide behind
            break;
        }
        record(IntC<1>{});
        if (!has_next) {
#pragma unroll
            for (int t = 0; t < CR_MT; ++t) epi_tile(acc[1][t], prev, t);
            break;
        }
    }
}

}  // namespace olx

using namespace olx;

#ifdef OLX_EXP_STAMPS
extern "C" int olx_exp_read_stamps_cosetr(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(olx::g_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif

template <int MX, int MY>
static void launch_cosetr(olx_ctx* c, float* pm) {
    const CosetParams& Q = c->cp;
    const int n_items = (int)c->cpr_nblocks;
    dim3 grid((unsigned)std::min(c->n_cu, n_items)), blk(CR_NW * 64);
    const bool clamp = c->clamp || c->lat.clamp;
#define OLX_CR(CL, F8) hipLaunchKernelGGL((field_cosetr_k<MX, MY, CL, F8>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_targets, c->d_cprblocks, Q, n_items)
    if (c->fp8corr) { if (clamp) OLX_CR(true, true); else OLX_CR(false, true); }
    else            { if (clamp) OLX_CR(true, false); else OLX_CR(false, false); }
#undef OLX_CR
}

void olx_launch_cosetr(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) launch_cosetr<2, 2>(c, pm);
    else if (c->mx == 2) launch_cosetr<2, 1>(c, pm);
    else if (c->my == 2) launch_cosetr<1, 2>(c, pm);
    else launch_cosetr<1, 1>(c, pm);
}
#endif  // OLX_AB_VARIANTS
