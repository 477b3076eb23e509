// kernel 2g (field_cosetp_k): kernel 2e's NT = 2 shape with the PLANES in the MFMA rows -- no output staging
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include <algorithm>
#include <cstdlib>
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// Kernel 2e gives a wave a position grid on 2 planes and lists (plane, position) down the MFMA rows; its results then sit
// in registers with the z neighbours of a voxel in OTHER waves, so every block transposes its whole output through LDS
// (|p| -> staging -> barrier -> read-out over a job list -> 64-byte runs): 29 % of a wave's lifetime on the headline shard.
// Here the block's footprint, operands, tables and arithmetic are the same, but the row map is turned:
//   * MFMA tile = ONE position x the block's 16 planes (row = plane).  Wave w takes the positions w, w + 8, ... of the
//     block's position grid (<= 5 tiles).  A lane then holds, for its column, rows 4 g .. 4 g + 3 = FOUR CONSECUTIVE PLANES
//     of one voxel column: |p| / intensity in place, one 16-byte store per store target straight from the accumulators
//     (after one lane transpose four CONSECUTIVE lanes write 64 contiguous bytes, as kernel 2e's read-out does).
//     No staging arena, no epilogue barriers, no job-list pass.
//   * The geometry tables of the 16 planes are shared by the block: wave w still evaluates planes 2 w, 2 w + 1 (same
//     26-row pair tables, same arithmetic), behind the block barriers that the steering-fragment staging needs anyway.
//   * A fragment of row p (plane), k-group g: table of plane p, row ky - g + ROW0 - 4 kb - 8 sl, entries UW - 8 - 2 kx + 4 ka ..:
//     a per-lane base (plane, g) + a wave-uniform tile offset.  Plane stride 364 words (182 = 22 (mod 32) 8-byte slots: 16 planes ->
//     16 distinct even slots) and row stride 14 words (7 slots: the second k-group of a ds_read_b64 lane group -> the odd slots)
//     put the 32 lanes of a ds_read_b64 group (16 planes x 2 k-groups) on 32 distinct 8-byte slots.
// NT = 2 (9 - 16 distinct steering columns), KX <= 3: the shape of BASELINE's 8-focus shard; the other shapes stay with 2e.
// Measured on the headline shard (8 foci, 256 el x 256^3, alternating runs on one box): 0.404 vs 0.430 ms with fp8 corrections,
// 0.456 vs 0.499 ms with fp16 corrections (kernel 2e).  A first version was 12 % SLOWER: the read-out loop carried the ragged-nz
// store form in the same body, which makes the compiler split every 16-byte store into 12 + 4 bytes (twice the store
// instructions, and the vector memory pipe takes ~16 cycles per wave-instruction whatever its width or exec mask).
// ------------------------------------------------------------------------------------
constexpr int CP_TW = 14;                          // words per table row (12 in use): = 2 (mod 4)
constexpr int CP_TROWS = 26, CP_ROW0 = 15;         // pair table rows; row of offset wd = 0
constexpr int CP_PSZ = 364;                        // words per plane table = 26 x 14; PSZ / 2 = 22 (mod 32): see above
constexpr int CP_UW = 12;                          // table columns: ud = 2 kx - a in [-7, 4]
constexpr int CP_MT = 5;                           // tiles (positions) per wave: ceil(33 / 8)

#ifdef OLX_EXP_CUTRACE
// developer build only (tools/cutrace_cosetp.py): per block and wave {HW_ID, XCC_ID, cycle at entry, at the last store issued, at the
// last store acknowledged} -- the occupancy timeline of a CU (how long a slot idles between two blocks)
static __device__ unsigned long long g_cutrace[16384][8][5];
#define OLX_CUTRACE(k, v) do { if (lane == 0 && blockIdx.y == 0 && blockIdx.x < 16384) g_cutrace[blockIdx.x][wave][k] = (v); } while (0)
#else
#define OLX_CUTRACE(k, v)
#endif

template <int MX, int MY, bool CLAMP, bool FP8, bool DIR = false, bool BOTH = true /*|p| and intensity both wanted (the product's case): no flag tests between the stores*/>
__global__ __launch_bounds__(COS_NW * 64, 4) void field_cosetp_k(
    const uint4* __restrict__ bfrag, float* __restrict__ pmag, float* __restrict__ inten,
    const int* __restrict__ targets /*[tiles][32 columns][4]: focus * 4 + mirror image, -1 = none*/,
    const CosetBlock* __restrict__ blocks /*[n_items]*/, const CosetParams P) {
    constexpr int NT = 2, THREADS = COS_NW * 64, PAIR = 2;
    constexpr int TROWS = CP_TROWS, ROW0 = CP_ROW0, PSZ = CP_PSZ, MT = CP_MT;   // 26-row pair tables, 364 words per plane, <= 5 tiles per wave
    static_assert((PSZ / 2) % 2 == 0 && ((PSZ / 2) % 32 == 22 || (PSZ / 2) % 32 == 30), "16 planes on 16 distinct even 8-byte slots");
    constexpr int RPR = 64 / CP_UW, NROUND = (TROWS + RPR - 1) / RPR;       // 5 table rows per generation round: 6 rounds (pairs) / 4
    constexpr int B_KS_U4 = 128;                                                // uint4 per K-step and column tile: hi, lo (64 lanes each)
    constexpr int B_BYTES = PAIR * 4 * NT * B_KS_U4 * 16;                       // PAIR super-blocks of steering fragments
    constexpr int T_WORDS = COS_ZB * PSZ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES + T_WORDS * 4 + T_WORDS * 4 + 64];
    // steering stage: K-step record (sl * 4 + ks, nt) = B_KS_U4 uint4: [0] hi, [1] lo of the 64 lanes
    auto s_B = [&](int rec, int nt, int part) -> const uint4* { return reinterpret_cast<const uint4*>(smem) + ((size_t)(rec * NT + nt) * B_KS_U4 + part * 64); };
    unsigned* const s_hi = reinterpret_cast<unsigned*>(smem + B_BYTES);
    unsigned* const s_lo = s_hi + T_WORDS;
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p16 = lane & 15;
    OLX_CUTRACE(2, __builtin_readcyclecounter());
    OLX_CUTRACE(0, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 4));      // HW_REG_HW_ID
    OLX_CUTRACE(1, (unsigned long long)__builtin_amdgcn_s_getreg((31 << 11) | 20));     // HW_REG_XCC_ID
    // the block's share of the coset decomposition: one scalar load of the host's record (olx.hip; blockIdx order as in kernel 2e,
    // the two blocks that write the two 64-byte halves of the same 128-byte lines 8 ids apart = same XCD).  Decoded here it was
    // ~350 VALU instructions per wave -- a fifth of the wave's vector instructions: integer divisions have no scalar form.
    const int item = blockIdx.x;
    {
    const CosetBlock BK = blocks[item];
    const int npos = BK.npos, KY = BK.KY, ky_magic = BK.ky_magic;
    if (npos <= 0) return;                              // block-uniform
    const int ibase = BK.ibase, jbase = BK.jbase;
    const int k0 = BK.k0;
    const int ntile = __builtin_amdgcn_readfirstlane((npos - wave + COS_NW - 1) / COS_NW);      // this wave's positions: wave, wave + 8, ... (wave-uniform, <= MT)
    // table generation role (planes 2 wave, 2 wave + 1): lane -> (wl = lane / UW < RPR, ui = lane % UW); round r: rows RPR r + wl.
    // (Its per-lane constants are formed inside the pair loop from an opaque copy of the lane index: hoisted, they would be live
    // across the K-steps, where the fp8 shape has no register to spare -- 5 spilled registers cost 190 MB of scratch traffic.)
    // fragment read offset [words] of a tile's row for K-step (0, 0) = per-lane part (plane, k-group) + the tile's position
    // (wave-uniform: kept in scalar registers, added per tile and K-step group -- five registers fewer across the K-steps)
    const int lane_off = p16 * PSZ - g * CP_TW;
    int toff[MT];
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        const int pos = min(wave + COS_NW * t, npos - 1);
        const int kx = (pos * ky_magic) >> 16, ky = pos - kx * KY;                // scalar: pos / KY, exact for pos <= 40 (host checks)
        toff[t] = (ky + ROW0) * CP_TW + (CP_UW - 8 - 2 * kx);
    }
    floatx4_t acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    const int nsbp = P.nsbp;                    // even: chunks = table pairs never straddle sa
    const int n_sb = P.nsa * nsbp;
    constexpr int CHUNK_U4 = PAIR * 4 * NT * B_KS_U4, PRE = CHUNK_U4 / THREADS;
    static_assert(CHUNK_U4 % THREADS == 0, "chunk must split evenly over the block");
    uint4 pre[PRE];
    const uint4* const bsrc = bfrag + (size_t)tile * n_sb * (4 * NT * B_KS_U4);
    // (the steering fragments do not depend on the item: all items of a launch tile read the same chunks)
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int idx = tid + q * THREADS;
        pre[q] = idx < n_sb * 4 * NT * B_KS_U4 ? bsrc[idx] : make_uint4(0, 0, 0, 0);
    }
    OLX_STAMP(0);
    for (int sb0 = 0; sb0 < n_sb; sb0 += PAIR) {
        const int sa = sb0 / nsbp, sbb0 = sb0 - sa * nsbp;       // the pair (sa, sbb0), (sa, sbb0 + 1)
        // previous pair consumed: steering stage and tables are free.  (Not before the first pair: nothing to protect yet, and
        // __syncthreads() drains vmcnt -- the wave would wait for its first steering loads before the tables instead of behind them.)
        if (sb0 > 0) __syncthreads();
        if (sb0 == 0) OLX_STAMP(1);
        if (sb0 == 2) OLX_STAMP(7);
        // ---- G tables of planes 2 wave, 2 wave + 1: 26 rows x 12 offsets, shared by the pair's two super-blocks
        // (fp8 shape: table generation at raised priority -- the waves a block's K-steps wait for get the VALU first: -2 ... -4 %;
        // with fp16 corrections the matrix pipe is the scarcer resource and the same setting costs 1.5 %, priority on the K-steps 2.5 %)
        if constexpr (FP8) __builtin_amdgcn_s_setprio(1);
        if (k0 + wave * COS_P < P.nz) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int wl = lane_o / CP_UW, ui = lane_o - CP_UW * wl;
            const bool gen_lane = wl < RPR;
            const int Ulane = ibase + P.x_begin + P.ux0 + P.mx * (ui - 7);
            const int Wlane = jbase + P.uy0 + P.my * (wl - ROW0);
            const int tw_off = (wave * COS_P) * PSZ + wl * CP_TW + (CP_UW - 1 - ui);   // + z PSZ + RPR r TW
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
            // one table entry: G of squared lateral distance r2 on a plane dz2v above the elements -> hi / lo words at table offset o
            auto entry = [&](const float r2, const float dz2v, const float dy, const int o, const bool ok) __attribute__((always_inline)) {
                float d2 = r2 + dz2v;
                if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                const float ri = __builtin_amdgcn_rsqf(d2);
                const float ph = d2 * ri;
                float rs = ri * P.g_scale;
                if constexpr (DIR) rs *= table_mod(dx, dy, ph, ri, P.dir_wx, P.dir_wy, P.absorb_l2);      // (own instantiations: the default path never sees this)
                float gr = rs * __builtin_amdgcn_cosf(ph), gi = rs * __builtin_amdgcn_sinf(ph);
                asm volatile("" : "+v"(gr), "+v"(gi));      // (two plain multiplies: packed fp32 beside the partner block's matrix instructions measured slower)
                half2_t hi;
                if constexpr (FP8) hi = __builtin_convertvector(float2_t{gr, gi}, half2_t);      // to nearest: |lo| <= half an ulp
                else hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
                // lo = g - (float)hi in ONE mixed-precision fma per component (the compiler's form: a convert and a subtract)
                float lr, li;
                const unsigned hw = __builtin_bit_cast(unsigned, hi);
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
                unsigned lo_word;
                if constexpr (FP8) {             // e4m3 bytes [lo re, lo im | hi re, hi im], |.| <= 256 (448 overflows to NaN)
                    // (v_cvt_scalef32_pk_fp8_f32 DIVIDES by its power-of-two scale operand -- tools/probe/cvt_scale_probe.hip --
                    // and rounds / saturates as the unscaled convert: the four operand scalings cost no instruction)
                    short2_t w;                  // (both halves are written below)
                    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lr, li, 1.0f / COS_F8_LO, false);
                    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, gr, gi, 1.0f / COS_F8_HI, true);
                    lo_word = __builtin_bit_cast(unsigned, w);
                } else {
                    lo_word = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lr, li));
                }
                if (ok && OLX_IN(o, T_WORDS, 0)) {
                    s_hi[o] = __builtin_bit_cast(unsigned, hi);
                    s_lo[o] = lo_word;
                }
            };
            // rows 0 .. TROWS - 2 in whole rounds of RPR rows x the wave's two planes; the LAST row once for both planes (row group wl takes
            // plane wl): 11 instead of 12 evaluations per lane and pair
            static_assert((TROWS - 1) % RPR == 0 && COS_P <= RPR, "the last table row is shared by the planes of a wave");
#pragma unroll 2
            for (int r = 0; r < NROUND - 1; ++r) {
                const float W = (float)(Wsb + RPR * P.my * r);
                const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
                const float r2 = fmaf(dy, dy, dx2);
#pragma unroll
                for (int z = 0; z < COS_P; ++z) entry(r2, dz2[z], dy, z * PSZ + tw_off + RPR * r * CP_TW, gen_lane);
            }
            if (KY + 14 >= TROWS - 1) {      // (block-uniform: the fragments of KY positions reach table rows 0 .. KY + 14)
                const float W = (float)(Wsb - P.my * wl + (TROWS - 1) * P.my);
                const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
                float dzl = dz2[0];
#pragma unroll
                for (int z = 1; z < COS_P; ++z) dzl = wl == z ? dz2[z] : dzl;
                entry(fmaf(dy, dy, dx2), dzl, dy, (wave * COS_P + wl) * PSZ + (TROWS - 1) * CP_TW + (CP_UW - 1 - ui), wl < COS_P);
            }
        }
        if constexpr (FP8) __builtin_amdgcn_s_setprio(0);
        // this pair's steering fragments (requested one pair ahead; the first ones arrive behind the table generation above)
#pragma unroll
        for (int q = 0; q < PRE; ++q) reinterpret_cast<uint4*>(smem)[tid + q * THREADS] = pre[q];
        if (sb0 == 0) OLX_STAMP(2);
        __syncthreads();
        if (sb0 == 0) OLX_STAMP(3);
        {   // next pair's fragments: in flight during the K-steps, drained by the next barrier
            const int nxt = (sb0 + PAIR) * 4 * NT * B_KS_U4, lim = n_sb * 4 * NT * B_KS_U4;
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int idx = nxt + tid + q * THREADS;
                if (idx < lim) pre[q] = bsrc[idx];
            }
        }
#pragma unroll                                          // (unrolled: the pair position becomes part of the immediate table offsets)
        for (int sl = 0; sl < PAIR; ++sl) {
            if (sbb0 + sl >= P.nsb) break;              // padding super-block of an odd count: zero weights, nothing to do
            if constexpr (FP8) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {         // K-step pairs (ka = 0, 1): two fp16 hi*hi products + ONE fp8 product
                    Half8Bits bh[2][NT];
                    intx8_t b8[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                        for (int ka = 0; ka < 2; ++ka) {
                            bh[ka][nt].u = s_B(sl * 4 + 2 * kb + ka, nt, 0)[lane];
                            const uint4 q = s_B(sl * 4 + 2 * kb + ka, nt, 1)[lane];
                            b8[nt][4 * ka + 0] = (int)q.x; b8[nt][4 * ka + 1] = (int)q.y; b8[nt][4 * ka + 2] = (int)q.z; b8[nt][4 * ka + 3] = (int)q.w;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        if (t >= ntile) continue;            // wave-uniform
                        Half8Bits ah[2];
                        intx8_t a8;
                        int lo_t = lane_off;                 // (opaque: formed here, not hoisted into five live registers)
                        asm volatile("" : "+v"(lo_t));
                        const int ro = lo_t + toff[t];
#pragma unroll
                        for (int ka = 0; ka < 2; ++ka) {
                            const int kso = 4 * ka - (4 * kb + 8 * sl) * CP_TW;   // the pair's second super-block reads 8 table rows lower
                            if (!OLX_IN(ro + kso, T_WORDS - 3, 1)) continue;
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
                        bh[nt].u = s_B(sl * 4 + ks, nt, 0)[lane];
                        bl[nt].u = s_B(sl * 4 + ks, nt, 1)[lane];
                    }
                    const int kso = 4 * ka - (4 * kb + 8 * sl) * CP_TW;
#pragma unroll
                    for (int t = 0; t < MT; ++t) {
                        if (t >= ntile) continue;            // wave-uniform
                        Half8Bits ah, al;
                        int lo_t = lane_off;
                        asm volatile("" : "+v"(lo_t));
                        const int ro = lo_t + toff[t];
                        if (!OLX_IN(ro + kso, T_WORDS - 3, 1)) continue;
                        const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(s_hi + ro + kso);
                        const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(s_lo + ro + kso);
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
        }
        if (sb0 == 0) OLX_STAMP(4);
    }
    OLX_STAMP(5);
    // ---- epilogue, straight from the accumulators.  The steering fragments carry Re of column c in column tile 0 and Im of the same column
    // in column tile 1 (mfma_pack_k, split_reim), so lane (g, c16) holds BOTH parts of column c16 for rows 4 g .. 4 g + 3 = planes
    // k0 + 4 g .. + 3 of the tile's position: |p| and intensity in place without a lane exchange, two 16-byte stores per store target.
    // (Rounds 2 - 4 kept (Re, Im) in adjacent matrix columns as kernels 2c / 2e do: a quad swap per square, selects between the |p| lane
    // and the intensity lane, every address formed once per lane pair -- 48 instead of 12 vector instructions per position, 5 instead
    // of 3.5 per store: 939 -> 734 vector instructions per wave.  That alone changed the launch time by nothing -- profiles/r05_store_path.txt.)
    //   A  |p| / intensity over the accumulators (two rows per packed fp32 instruction), then the lane transpose below;
    //   B  per store-target slot of the lane's column (outer) the 64-bit bases of its focus volumes and its mirror masks once, then
    //      per tile (inner) offset = o00 + (fx & DX) + (fy & DY) with the three terms wave-uniform (scalar ALU), shared by both stores.
    int lane_e = lane;                                   // (opaque: the epilogue's per-lane constants -- targets, bases -- are formed here,
    asm volatile("" : "+v"(lane_e));                     // not kept in registers across the K-steps)
    // The accumulators put the four plane quads of a column 16 lanes apart; the vector memory unit takes a 16-byte store four LANES at a
    // time, so from there every 64-byte run goes out as four separate 16-byte requests.  One lane transpose first (ds_bpermute: the LDS
    // crossbar, no vector-ALU work): lane 4 c + g takes over column c, planes 4 g .. 4 g + 3 -- four consecutive lanes now write 64
    // contiguous bytes: -7 % on the launch (profiles/r05_store_path.txt).
    const int c16 = lane_e >> 2;
    const int kz = k0 + 4 * (lane_e & 3);
    const int src_lane4 = (16 * (lane_e & 3) + (lane_e >> 2)) * 4;      // byte index of the lane that computed this lane's values
    const float s_p = P.out_scale, s_i = P.out_scale * P.out_scale * P.inten_scale;
    const bool want_p = (P.flags & 1u) != 0, want_i = (P.flags & 2u) != 0;      // (uniform)
#pragma unroll
    for (int t = 0; t < MT; ++t) {
        if (t >= ntile) continue;
        float o[NT][4];
#pragma unroll
        for (int r = 0; r < 4; r += 2) {
            const float2_t re = {acc[t][0][r], acc[t][0][r + 1]}, im = {acc[t][1][r], acc[t][1][r + 1]};
            const float2_t m = __builtin_elementwise_fma(re, re, im * im);
            o[0][r] = __builtin_amdgcn_sqrtf(m.x) * s_p;
            o[0][r + 1] = __builtin_amdgcn_sqrtf(m.y) * s_p;
            const float2_t in = m * float2_t{s_i, s_i};
            o[1][r] = in.x;
            o[1][r + 1] = in.y;
        }
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
            floatx4_t v;
            v[0] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, o[nt][0])));
            v[1] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, o[nt][1])));
            v[2] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, o[nt][2])));
            v[3] = __builtin_bit_cast(float, __builtin_amdgcn_ds_bpermute(src_lane4, __builtin_bit_cast(int, o[nt][3])));
            acc[t][nt] = v;
        }
    }
    if (kz < P.nz) {
    // store targets of this lane's column: focus * 4 + mirror image, -1 = none
    const int4 tq = *reinterpret_cast<const int4*>(targets + ((size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + c16) * 4);
    const int xm = P.nx - 1, ym = P.ny - 1;
    const int sxz = P.ny * P.nz;
    // (the ragged-nz variant is a separate copy of the loop: with both store forms in one body the compiler merges them and
    // splits every 16-byte store into a 12-byte and a 4-byte instruction)
    auto readout = [&](auto full_c) {
        constexpr bool FULL4 = decltype(full_c)::value != 0;
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int code = q == 0 ? tq.x : q == 1 ? tq.y : q == 2 ? tq.z : tq.w;
            if (code < 0) continue;
            const unsigned m = (unsigned)code & 3u;
            const bool fx = (MX == 2) && (m & 1u), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1u));
            unsigned fxm = fx ? 0xFFFFFFFFu : 0u, fym = fy ? 0xFFFFFFFFu : 0u;
            asm volatile("" : "+v"(fxm), "+v"(fym));      // (opaque: kept as masks -- one v_and per term instead of a move and a select)
            const long long fb = (long long)(code >> 2) * P.vox + kz;
            float* const base_p = pmag + fb;
            float* const base_i = inten + fb;
#pragma unroll
            for (int t = 0; t < MT; ++t) {
                if (t >= ntile) continue;
                const int pos = wave + COS_NW * t;
                const int kx = (pos * ky_magic) >> 16, ky = pos - kx * KY;
                const int i = ibase + 2 * P.mx * kx, j = jbase + P.my * ky;      // wave-uniform (scalar ALU)
                const unsigned o00 = (unsigned)(i * sxz + j * P.nz);
                const unsigned DX = (unsigned)((xm - 2 * i) * sxz), DY = (unsigned)((ym - 2 * j) * P.nz);
                const unsigned off = o00 + (fxm & DX) + (fym & DY);
                if (!OLX_IN(fb + off + (FULL4 ? 3 : 0), (long long)P.n_foci * P.vox, 2)) continue;
                if constexpr (FULL4) {
                    if (BOTH || want_p) *reinterpret_cast<floatx4u_t*>(base_p + off) = floatx4u_t{acc[t][0][0], acc[t][0][1], acc[t][0][2], acc[t][0][3]};
                    if (BOTH || want_i) *reinterpret_cast<floatx4u_t*>(base_i + off) = floatx4u_t{acc[t][1][0], acc[t][1][1], acc[t][1][2], acc[t][1][3]};
                } else {
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (kz + e < P.nz) {
                            if (BOTH || want_p) base_p[off + e] = acc[t][0][e];
                            if (BOTH || want_i) base_i[off + e] = acc[t][1][e];
                        }
                }
            }
        }
    };
    if (k0 + COS_ZB <= P.nz) readout(IntC<1>{}); else readout(IntC<0>{});      // (block-uniform: only a LAST, partial plane block stores plane by plane)
    }
    OLX_STAMP(6);
#ifdef OLX_EXP_CUTRACE
    OLX_CUTRACE(3, __builtin_readcyclecounter());
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    OLX_CUTRACE(4, __builtin_readcyclecounter());
#endif
    }
}

}  // namespace olx

using namespace olx;
OLX_BOUNDS_READER(cosetp)

#ifdef OLX_EXP_CUTRACE
extern "C" int olx_exp_read_cutrace_cosetp(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(olx::g_cutrace), sizeof(unsigned long long) * 16384 * 8 * 5);
}
#endif
#ifdef OLX_EXP_STAMPS
extern "C" int olx_exp_read_stamps_cosetp(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(olx::g_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif

template <int MX, int MY>
static void launch_cosetp(olx_ctx* c, float* pm) {
    CosetParams Q = c->cp;
#ifdef OLX_DEBUG_BOUNDS   // self-test of the debug build: pretend the output arrays hold one focus less -- the last focus' stores must be reported (and skipped)
    if (getenv("OLX_DEBUG_BOUNDS_SELFTEST")) Q.n_foci -= 1;
#endif
    float* inten_exp = c->d_inten;
    const bool clamp = c->clamp || c->lat.clamp;
    dim3 grid((unsigned)c->cp_nblocks, c->mp.n_tiles), blk(COS_NW * 64);
    const bool both = (Q.flags & 3u) == 3u;
#define OLX_CP(CL, F8, DR) do { if (both) hipLaunchKernelGGL((field_cosetp_k<MX, MY, CL, F8, DR, true>), grid, blk, 0, c->stream, c->d_bfrag, pm, inten_exp, c->d_targets, c->d_cpblocks, Q); \
                                else hipLaunchKernelGGL((field_cosetp_k<MX, MY, CL, F8, DR, false>), grid, blk, 0, c->stream, c->d_bfrag, pm, inten_exp, c->d_targets, c->d_cpblocks, Q); } while (0)
    if (c->dir_lattice) {   // piston directivity / uniform absorption folded into the geometry tables (fp16 corrections only)
        if (clamp) OLX_CP(true, false, true); else OLX_CP(false, false, true);
    } else if (c->fp8corr) {
        if (clamp) OLX_CP(true, true, false); else OLX_CP(false, true, false);
    } else {
        if (clamp) OLX_CP(true, false, false); else OLX_CP(false, false, false);
    }
#undef OLX_CP
}

void olx_launch_cosetp(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) launch_cosetp<2, 2>(c, pm);
    else if (c->mx == 2) launch_cosetp<2, 1>(c, pm);
    else if (c->my == 2) launch_cosetp<1, 2>(c, pm);
    else launch_cosetp<1, 1>(c, pm);
}
