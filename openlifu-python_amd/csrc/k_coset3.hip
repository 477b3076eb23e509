// kernel 2q (field_cosetq_k): kernel 2g cut into blocks of FOUR waves x EIGHT planes -- four blocks per CU instead of two
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#ifdef OLX_AB_VARIANTS   // measured-slower A/B form: compiled only into the developer library (build.py -DOLX_AB_VARIANTS), never into libolx.so
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// Kernel 2g runs two 8-wave blocks per CU (79 KB of LDS each).  Its phases -- table generation (VALU), K-steps (matrix pipe, LDS
// reads), |p| + stores (vector memory) -- are separated by block barriers, so a SIMD sees at most two different phases at a time,
// and the CU trace (tools/cutrace_cosetp.py) shows a slot idle ~19 % of the time at block turn-over (the block's eight waves finish
// 4.8 k cycles apart, the next block arrives 2.8 - 17 k cycles after the last store is acknowledged).
// Here a block is 4 waves and owns 8 planes:
//   * MFMA tile = TWO y-adjacent positions x the block's 8 planes: row p16 -> plane p16 & 7, position (kx, 2 kyp + (p16 >> 3)).
//     The second position's fragments are the first one's, one table row higher: the per-lane base carries (p16 >> 3) * TW, the
//     tile offset stays wave-uniform.  Plane stride 334 words, row stride 12: the 24 distinct addresses of a ds_read_b64 lane
//     group (8 planes x row offsets {-1, 0, 1}) fall on 24 distinct 8-byte slots.
//   * A lane's rows 4 g .. 4 g + 3 are planes 4 (g & 1) .. + 3 of position g >> 1: still one 16-byte store per target.
//   * Tables: wave w evaluates planes 2 w, 2 w + 1 (same 26-row pair tables and arithmetic as 2g); steering fragments are staged one
//     super-block (16 KB) at a time: 37.4 KB of LDS per block, four blocks (16 waves, 128 VGPRs) per CU, each wave of a block on
//     its own SIMD beside waves of three other blocks in other phases.
// NT = 2 (9 - 16 distinct steering columns), KX <= 3, KY <= 12 as kernel 2g.
// ------------------------------------------------------------------------------------
constexpr int CQ_NW = 4;                           // waves per block
constexpr int CQ_P = 2;                            // planes per wave (table generation)
constexpr int CQ_ZB = CQ_NW * CQ_P;                // planes per block
constexpr int CQ_TW = 12;                          // words per table row
constexpr int CQ_TROWS = 26, CQ_ROW0 = 15;         // pair table rows; row of offset wd = 0
constexpr int CQ_PSZ = 334;                        // words per plane table (>= 27 rows: the idle second position of an odd KY reads one row past the table)
constexpr int CQ_UW = 12;                          // table columns: ud = 2 kx - a in [-7, 4]
constexpr int CQ_MT = 5;                           // tiles (position pairs) per wave: ceil(3 * 6 / 4)

template <int MX, int MY, bool CLAMP, bool FP8>
__global__ __launch_bounds__(CQ_NW * 64, 4) void field_cosetq_k(
    const uint4* __restrict__ bfrag, float* __restrict__ pmag, float* __restrict__ inten,
    const int* __restrict__ targets /*[tiles][32 columns][4]: focus * 4 + mirror image, -1 = none*/,
    const CosetBlock* __restrict__ blocks /*[gridDim.x]*/, const CosetParams P) {
    constexpr int NT = 2, THREADS = CQ_NW * 64;
    constexpr int RPR = 64 / CQ_UW, NROUND = (CQ_TROWS + RPR - 1) / RPR;       // 5 table rows per generation round, 6 rounds
    constexpr int B_BYTES = 4 * NT * 2 * 64 * 16;                               // ONE super-block of steering fragments
    constexpr int T_WORDS = CQ_ZB * CQ_PSZ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[B_BYTES + 2 * T_WORDS * 4 + 64];
    typedef uint4 (*BArr)[NT][2][64];
    BArr s_B = reinterpret_cast<BArr>(smem);
    unsigned* const s_hi = reinterpret_cast<unsigned*>(smem + B_BYTES);
    unsigned* const s_lo = s_hi + T_WORDS;
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int g = lane >> 4, p16 = lane & 15;
    // the block's share of the coset decomposition: one scalar load of the host's record (olx.hip)
    const CosetBlock BK = blocks[blockIdx.x];
    const int npair = BK.npos, KY = BK.KY, kyp_magic = BK.ky_magic;          // npos = KX * KYP position pairs, KYP = ceil(KY / 2)
    if (npair <= 0) return;                             // block-uniform
    const int KYP = (KY + 1) >> 1;
    const int ibase = BK.ibase, jbase = BK.jbase;
    const int k0 = BK.k0;
    const int ntile = __builtin_amdgcn_readfirstlane((npair - wave + CQ_NW - 1) / CQ_NW);      // this wave's pairs: wave, wave + 4, ... (<= CQ_MT)
    // fragment read offset [words] of a tile's row for K-step (0, 0) = per-lane part (plane, second position, k-group) + the tile's
    // first position (wave-uniform: scalar registers, added per tile and K-step group)
    const int lane_off = (p16 & 7) * CQ_PSZ + ((p16 >> 3) - g) * CQ_TW;
    int toff[CQ_MT];
#pragma unroll
    for (int t = 0; t < CQ_MT; ++t) {
        const int pr = min(wave + CQ_NW * t, npair - 1);
        const int kx = (pr * kyp_magic) >> 16, kyp = pr - kx * KYP;               // scalar: pr / KYP, exact for pr <= 40 (host checks)
        toff[t] = (2 * kyp + CQ_ROW0) * CQ_TW + (CQ_UW - 8 - 2 * kx);
    }
    floatx4_t acc[CQ_MT][NT];
#pragma unroll
    for (int t = 0; t < CQ_MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    const int nsbp = P.nsbp;                    // even: table pairs never straddle sa
    const int n_sb = P.nsa * nsbp;
    constexpr int CHUNK_U4 = 4 * NT * 128, PRE = CHUNK_U4 / THREADS;
    static_assert(CHUNK_U4 % THREADS == 0, "chunk must split evenly over the block");
    uint4 pre[PRE];
    const uint4* const bsrc = bfrag + (size_t)tile * n_sb * CHUNK_U4;
#pragma unroll
    for (int q = 0; q < PRE; ++q) pre[q] = bsrc[tid + q * THREADS];
    for (int sb0 = 0; sb0 < n_sb; sb0 += 2) {
        const int sa = sb0 / nsbp, sbb0 = sb0 - sa * nsbp;       // the pair (sa, sbb0), (sa, sbb0 + 1)
        // previous pair consumed: steering stage and tables are free.  (Not before the first pair: nothing to protect yet, and
        // __syncthreads() drains vmcnt -- the wave would wait for its first steering loads before the tables instead of behind them.)
        if (sb0 > 0) __syncthreads();
        // ---- G tables of planes 2 wave, 2 wave + 1: 26 rows x 12 offsets, shared by the pair's two super-blocks
        if constexpr (FP8) __builtin_amdgcn_s_setprio(1);
        if (k0 + wave * CQ_P < P.nz) {
            int lane_o = lane;
            asm volatile("" : "+v"(lane_o));
            const int wl = lane_o / CQ_UW, ui = lane_o - CQ_UW * wl;
            const bool gen_lane = wl < RPR;
            const int Ulane = ibase + P.x_begin + P.ux0 + P.mx * (ui - 7);
            const int Wlane = jbase + P.uy0 + P.my * (wl - CQ_ROW0);
            const int tw_off = (wave * CQ_P) * CQ_PSZ + wl * CQ_TW + (CQ_UW - 1 - ui);   // + z PSZ + RPR r TW
            float dz2[CQ_P];
#pragma unroll
            for (int z = 0; z < CQ_P; ++z) {
                const float dz = (float)(k0 + wave * CQ_P + z) * P.hz - P.flat_ez;
                dz2[z] = dz * dz;
            }
            const float U = (float)(Ulane - 8 * P.mx * sa);
            const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
            const float dx2 = dx * dx;
            const int Wsb = Wlane - 8 * P.my * sbb0;
#pragma unroll 2
            for (int r = 0; r < NROUND; ++r) {
                const bool row_ok = gen_lane && RPR * r + wl < CQ_TROWS;  // the last round may run past the table
                const float W = (float)(Wsb + RPR * P.my * r);
                const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
                const float r2 = fmaf(dy, dy, dx2);
#pragma unroll
                for (int z = 0; z < CQ_P; ++z) {
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
                    // lo = g - (float)hi in ONE mixed-precision fma per component (the compiler's form: a convert and a subtract)
                    float lr, li;
                    const unsigned hw = __builtin_bit_cast(unsigned, hi);
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
                    unsigned lo_word;
                    if constexpr (FP8) {             // e4m3 bytes [lo re, lo im | hi re, hi im], |.| <= 256 (448 overflows to NaN)
                        short2_t w;                  // (both halves are written below; the scale operand DIVIDES: tools/probe/cvt_scale_probe.hip)
                        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lr, li, 1.0f / COS_F8_LO, false);
                        w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, gr, gi, 1.0f / COS_F8_HI, true);
                        lo_word = __builtin_bit_cast(unsigned, w);
                    } else {
                        lo_word = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lr, li));
                    }
                    if (row_ok) {
                        const int o = z * CQ_PSZ + tw_off + RPR * r * CQ_TW;
                        s_hi[o] = __builtin_bit_cast(unsigned, hi);
                        s_lo[o] = lo_word;
                    }
                }
            }
        }
        if constexpr (FP8) __builtin_amdgcn_s_setprio(0);
#pragma unroll                                          // (unrolled: the pair position becomes part of the immediate table offsets)
        for (int sl = 0; sl < 2; ++sl) {
            if (sbb0 + sl >= P.nsb) break;              // padding super-block of an odd count: zero weights, nothing to do
            // this super-block's steering fragments (requested one super-block ahead); the stage is free: the barrier at the top of the
            // pair loop (sl = 0) or the one below (sl = 1) follows the K-steps that read it
            if (sl == 1) __syncthreads();
#pragma unroll
            for (int q = 0; q < PRE; ++q) reinterpret_cast<uint4*>(smem)[tid + q * THREADS] = pre[q];
            __syncthreads();
            {   // next ACTIVE super-block's fragments (a padding super-block is skipped): in flight during the K-steps, drained by the next barrier
                const int nxt_sb = (sl == 0 && sbb0 + 1 < P.nsb) ? sb0 + 1 : sb0 + 2;
                const int nxt = nxt_sb * CHUNK_U4, lim = n_sb * CHUNK_U4;
#pragma unroll
                for (int q = 0; q < PRE; ++q) {
                    const int idx = nxt + tid + q * THREADS;
                    if (idx < lim) pre[q] = bsrc[idx];
                }
            }
            if constexpr (FP8) {
#pragma unroll
                for (int kb = 0; kb < 2; ++kb) {         // K-step pairs (ka = 0, 1): two fp16 hi*hi products + ONE fp8 product
                    Half8Bits bh[2][NT];
                    intx8_t b8[NT];
#pragma unroll
                    for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
                        for (int ka = 0; ka < 2; ++ka) {
                            bh[ka][nt].u = s_B[2 * kb + ka][nt][0][lane];
                            const uint4 q = s_B[2 * kb + ka][nt][1][lane];
                            b8[nt][4 * ka + 0] = (int)q.x; b8[nt][4 * ka + 1] = (int)q.y; b8[nt][4 * ka + 2] = (int)q.z; b8[nt][4 * ka + 3] = (int)q.w;
                        }
                    }
#pragma unroll
                    for (int t = 0; t < CQ_MT; ++t) {
                        if (t >= ntile) continue;            // wave-uniform
                        Half8Bits ah[2];
                        intx8_t a8;
                        int lo_t = lane_off;                 // (opaque: formed here, not hoisted into five live registers)
                        asm volatile("" : "+v"(lo_t));
                        const int ro = lo_t + toff[t];
#pragma unroll
                        for (int ka = 0; ka < 2; ++ka) {
                            const int kso = 4 * ka - (4 * kb + 8 * sl) * CQ_TW;   // the pair's second super-block reads 8 table rows lower
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
                        bh[nt].u = s_B[ks][nt][0][lane];
                        bl[nt].u = s_B[ks][nt][1][lane];
                    }
                    const int kso = 4 * ka - (4 * kb + 8 * sl) * CQ_TW;
#pragma unroll
                    for (int t = 0; t < CQ_MT; ++t) {
                        if (t >= ntile) continue;            // wave-uniform
                        Half8Bits ah, al;
                        int lo_t = lane_off;
                        asm volatile("" : "+v"(lo_t));
                        const int ro = lo_t + toff[t];
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
    }
    // ---- epilogue, straight from the accumulators (as kernel 2g).  Lane (g, c16): rows 4 g .. 4 g + 3 = planes k0 + 4 (g & 1) .. + 3
    // of the tile's position g >> 1, column c16 = (o, re | im).  Two passes: A |p| / intensity in place; B per store-target slot of
    // the lane's column (outer) the base of its focus volume and its mirror masks once, then per tile (inner) the voxel offset
    // from wave-uniform terms, the second position one y pitch further.
    const int c16 = lane & 15, part = c16 & 1;
    const int kz = k0 + 4 * (g & 1);
    const int psel = g >> 1;                              // which of the tile's two positions this lane stores
    if (kz >= P.nz) return;
    const float s_lane = part == 0 ? P.out_scale : P.out_scale * P.out_scale * P.inten_scale;
    float* const vol = part ? inten : pmag;
    const bool want = (P.flags & (part ? 2u : 1u)) != 0;
    int4 tq[NT];                                         // store targets of this lane's column: focus * 4 + mirror image, -1 = none
#pragma unroll
    for (int nt = 0; nt < NT; ++nt)
        tq[nt] = *reinterpret_cast<const int4*>(targets + ((size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt * MFMA_COLS + (c16 >> 1)) * 4);
#pragma unroll
    for (int t = 0; t < CQ_MT; ++t) {
        if (t >= ntile) continue;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int r = 0; r < 4; r += 2) {
                const float a0 = acc[t][nt][r], a1 = acc[t][nt][r + 1];
                const float sq0 = a0 * a0, sq1 = a1 * a1;
                const float m0 = __builtin_fmaf(a0, a0, quad_swap1(sq0)), m1 = __builtin_fmaf(a1, a1, quad_swap1(sq1));   // (pinned: own square unrounded, partner's rounded)
                const float y = __builtin_amdgcn_sqrtf(part == 0 ? m0 : m1);
                const float ys = quad_swap1(y);
                acc[t][nt][r] = (part == 0 ? y : m0) * s_lane;
                acc[t][nt][r + 1] = (part == 0 ? ys : m1) * s_lane;
            }
        }
    }
    const int xm = P.nx - 1, ym = P.ny - 1;
    const int sxz = P.ny * P.nz;
    // second position of a tile: j + my -> the voxel offset moves by my nz, its mirror term by -2 my nz (per-lane constants)
    const unsigned pj = psel ? (unsigned)(P.my * P.nz) : 0u;
    // (the ragged-nz variant is a separate copy of the loop: with both store forms in one body the compiler merges them and
    // splits every 16-byte store into a 12-byte and a 4-byte instruction)
    auto readout = [&](auto full_c) {
        constexpr bool FULL4 = decltype(full_c)::value != 0;
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) {
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int code = want ? (q == 0 ? tq[nt].x : q == 1 ? tq[nt].y : q == 2 ? tq[nt].z : tq[nt].w) : -1;
                if (code < 0) continue;
                const unsigned m = (unsigned)code & 3u;
                const bool fx = (MX == 2) && (m & 1u), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1u));
                unsigned fxm = fx ? 0xFFFFFFFFu : 0u;
                // y terms of this lane: offset of its position within the tile, +my nz unmirrored, -my nz mirrored
                unsigned pjl = fy ? 0u - pj : pj;
                unsigned fym = fy ? 0xFFFFFFFFu : 0u;
                asm volatile("" : "+v"(fxm), "+v"(fym), "+v"(pjl));      // (opaque: kept as masks -- one v_and per term instead of a move and a select)
                float* const base = vol + (long long)(code >> 2) * P.vox + kz;
#pragma unroll
                for (int t = 0; t < CQ_MT; ++t) {
                    if (t >= ntile) continue;
                    const int pr = wave + CQ_NW * t;
                    const int kx = (pr * kyp_magic) >> 16, kyp = pr - kx * KYP;
                    const int i = ibase + 2 * P.mx * kx, j = jbase + P.my * 2 * kyp;  // the tile's first position: wave-uniform (scalar ALU)
                    if (2 * kyp + psel >= KY) continue;                               // idle second position of an odd KY
                    const unsigned o00 = (unsigned)(i * sxz + j * P.nz);
                    const unsigned DX = (unsigned)((xm - 2 * i) * sxz), DY = (unsigned)((ym - 2 * j) * P.nz);
                    const unsigned off = o00 + (fxm & DX) + ((fym & DY) + pjl);
                    float* dst = base + off;
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
static void launch_cosetq(olx_ctx* c, float* pm) {
    const CosetParams& Q = c->cp;
    dim3 grid(c->cp_nblocks, c->mp.n_tiles), blk(CQ_NW * 64);
    const bool clamp = c->clamp || c->lat.clamp;
#define OLX_CQ(CL, F8) hipLaunchKernelGGL((field_cosetq_k<MX, MY, CL, F8>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_targets, c->d_cpblocks, Q)
    if (c->fp8corr) { if (clamp) OLX_CQ(true, true); else OLX_CQ(false, true); }
    else            { if (clamp) OLX_CQ(true, false); else OLX_CQ(false, false); }
#undef OLX_CQ
}

void olx_launch_cosetq(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) launch_cosetq<2, 2>(c, pm);
    else if (c->mx == 2) launch_cosetq<2, 1>(c, pm);
    else if (c->my == 2) launch_cosetq<1, 2>(c, pm);
    else launch_cosetq<1, 1>(c, pm);
}
#endif  // OLX_AB_VARIANTS
