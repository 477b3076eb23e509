// kernel 2e (field_coset_k): lattice arrays, whole cosets per wave (the headline kernel)
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

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


template <int NT, int MX, int MY, bool CLAMP, bool FP8, bool DIR>
__global__ __launch_bounds__(COS_NW * 64, 4) void field_coset_k(
    const uint4* __restrict__ bfrag, float* __restrict__ pmag, float* __restrict__ inten, float* __restrict__ cplx,
    const int* __restrict__ jobs /*[tiles][MFMA_MAX_NT][COS_JOBS + 1]: dense (column, focus, image) store jobs, [COS_JOBS] = log2 count*/,
    const CosetBlock* __restrict__ blocks /*[gridDim.x]*/, const CosetParams P) {
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
    // the block's share of the coset decomposition: one scalar load of the host's record (olx.hip, as kernels 2g / 2f; the two
    // blocks that write the two 64-byte halves of the same 128-byte lines have ids 8 apart = same XCD, i.e. the same L2, under
    // round-robin dispatch: measured -1...-2 %).  Decoded here, the chain of integer divisions was ~350 vector instructions per wave.
    const CosetBlock BK = blocks[blockIdx.x];
    const int KX = BK.KX, KY = BK.KY;
    const int ibase = BK.ibase, jbase = BK.jbase;
    const int npos = BK.npos, nrow = COS_P * npos;
    const int k0 = BK.k0 + wave * COS_P;
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
                        float rs = ri * P.g_scale;
                        if constexpr (DIR) rs *= table_mod(dx, dy, ph, ri, P.dir_wx, P.dir_wy, P.absorb_l2);      // (own instantiations: the default path never sees this)
                        const float gr = rs * __builtin_amdgcn_cosf(ph);
                        const float gi = rs * __builtin_amdgcn_sinf(ph);
                        // fp8 corrections: hi rounded to nearest (v_cvt_pk_f16_f32) so that |lo| <= half an ulp
                        half2_t hi;
                        if constexpr (FP8) hi = __builtin_convertvector(float2_t{gr, gi}, half2_t);
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
    const int kb0 = BK.k0;
    const bool fast = kb0 + COS_ZB <= P.nz;      // (block-uniform: only a LAST, partial plane block stores plane by plane)
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
                        dst[u] = qu < npos ? base + (long long)(io * P.ny + jo) * P.nz : nullptr;
                        if (dst[u] && !OLX_IN((long long)(job >> 6) * P.vox + kz + (long long)(io * P.ny + jo) * P.nz + (FAST ? 3 : 0), (long long)P.n_foci * P.vox, 3)) dst[u] = nullptr;
                        const float* v = sv + (qu < npos ? qu : q) * RS;
                        val[u] = make_float4(v[0], v[1], v[2], v[3]);
                    }
#pragma unroll
                    for (int u = 0; u < RU; ++u) {
                        if (!dst[u]) continue;
                        if constexpr (FAST) {
                            *reinterpret_cast<floatx4u_t*>(dst[u]) = floatx4u_t{val[u].x, val[u].y, val[u].z, val[u].w};
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
    stage_and_store(IntC<0>{});
    if constexpr (NT > GROUP) stage_and_store(IntC<GROUP>{});
    OLX_STAMP(7);
}


}  // namespace olx

using namespace olx;
OLX_BOUNDS_READER(coset)

template <int NT, int MX, int MY>
static void launch_coset(olx_ctx* c, float* pm, bool clamp) {
    const CosetParams& Q = c->cp;
    dim3 grid(c->cp_nblocks, c->mp.n_tiles), blk(COS_NW * 64);      // (the block records of this launch: all of them, or one side of a launch split at fp8_kcut)
#define OLX_COS(CL, F8) hipLaunchKernelGGL((field_coset_k<NT, MX, MY, CL, F8, false>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_cplx, c->d_jobs, c->d_cpblocks, Q)
    if (c->dir_lattice) {   // piston directivity folded into the geometry tables (fp16 corrections only)
        if (clamp) hipLaunchKernelGGL((field_coset_k<NT, MX, MY, true, false, true>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_cplx, c->d_jobs, c->d_cpblocks, Q);
        else hipLaunchKernelGGL((field_coset_k<NT, MX, MY, false, false, true>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_cplx, c->d_jobs, c->d_cpblocks, Q);
        return;
    }
    if constexpr (cos_fp8(NT)) {
        if (c->fp8corr) { if (clamp) OLX_COS(true, true); else OLX_COS(false, true); return; }
    }
    if (clamp) OLX_COS(true, false); else OLX_COS(false, false);
#undef OLX_COS
}

template <int MX, int MY>
static void dispatch_coset_nt(olx_ctx* c, float* pm) {
    const bool clamp = c->clamp || c->lat.clamp;
    if (c->nt == 1) launch_coset<1, MX, MY>(c, pm, clamp); else if (c->nt == 2) launch_coset<2, MX, MY>(c, pm, clamp);
    else launch_coset<4, MX, MY>(c, pm, clamp);
}

void olx_launch_coset(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) dispatch_coset_nt<2, 2>(c, pm);
    else if (c->mx == 2) dispatch_coset_nt<2, 1>(c, pm);
    else if (c->my == 2) dispatch_coset_nt<1, 2>(c, pm);
    else dispatch_coset_nt<1, 1>(c, pm);
}

#ifdef OLX_EXP_STAMPS
// developer build only (tools/stamps.py): per-wave phase time stamps of this kernel
extern "C" int olx_exp_read_stamps(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(olx::g_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif
