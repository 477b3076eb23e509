// kernel 2f (field_toep_k): lattice arrays, ONE steering column -- Toeplitz weights stationary, geometry tables stream
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"
#include "k_toep.hip.h"

namespace olx {

// ------------------------------------------------------------------------------------
// Kernel 2e contracts G[voxel rows, elements] with W[elements, columns]: with ONE distinct steering vector (an on-axis
// SinglePoint focus -- the reference's default focal pattern -- on a mirror-symmetric array: all four images share it)
// 2 of the matrix pipe's 16 output columns carry data.  For a lattice array the same sum is a dilated 2-D convolution
//     out(kx, ky, k) = sum_{a, b} W(a, b) g(2 kx - a, ky - b, k)           (coset positions, table offsets as in 2e)
// so the ROLES can be swapped: for one element row b the weights form a Toeplitz matrix over the table's x offsets,
//     A_b[(kx, o), (ud, c)] = coefficient of g_c(ud) in out_o(kx):  (wr, -wi | wi, wr) of W(2 kx - ud, b), else 0,
// and the table row wd = ky - b of SIXTEEN PLANES is the other operand, B[(ud, c), plane] = g_c(ud, ky - b, plane):
//     acc_ky[(kx, o), plane] += A_b . B_{ky - b}          v_mfma_f32_16x16x32_f16, fp16 hi/lo split, 3 products.
// M = 8 positions x (re, im) = 16/16 rows, N = 16 planes = 16/16 columns, K = 30 offsets x (re, im) -> 64 (47 - 81 % dense,
// the Toeplitz band): ~2.6 x fewer matrix instructions than 2e's NT = 1 shape on BASELINE's grids.
//   * Block = (coset, x part <= 8 NM positions -- NM = 1, 2, 3 row tiles that share tables and weights: ToepShape, k_toep.hip.h --, y part <= 11 positions,
//     16 planes), 8 waves = 4 y-position groups x 2 halves of the contraction.  Elements are walked in
//     super-blocks of 16 (a) x 8 (b); per super-block the block evaluates ONE table of 18 x <= 30 offsets for each of its 16
//     planes -- shared by all waves, (row, offset) pairs across the threads, the 16 planes in a register loop so that dx^2 +
//     dy^2 is formed once per pair -- as fp16 (re, im) hi and lo words in LDS.
//   * B fragment of lane (plane n = lane & 15, k-group g): offsets 16 s + 4 g .. + 3 of the table row = 16 contiguous
//     bytes: one ds_read_b128 per part and K-step.  Plane stride = 8 (mod 64) words puts the 16 lanes of every b128 lane
//     group on 16 distinct 16-byte slots (planes {0-3, 12-15} on the even ones, {4-11} one k-group further on the odd ones).
//   * A fragments (the Toeplitz weights: 4 x 16 bytes per lane and element row) are packed once per steering table by
//     toep_pack_k in lane order and arrive through L2, one element row ahead.
//   * Wave w owns the y positions ky = 3 w, 3 w + 1, 3 w + 2 (<= 3 accumulator tiles; round 6: consecutive, so that the pairs (ky, b) of a diagonal share their table row).  D layout: lane (g, n) holds rows
//     4 g .. 4 g + 3 = (kx = 2 g, re), (2 g, im), (2 g + 1, re), (2 g + 1, im) of plane n -- |p| and the intensity need
//     no cross-lane step, and the 16 lanes of a k-group write 64 contiguous bytes of z per (position, target).
// ------------------------------------------------------------------------------------
// FP8 (round 5; the planner's default where the foci lie inside the planned volume and N_eff >= 256, as for kernels 2e / 2g): the two hi x lo
// correction products of BOTH K-steps of an element row go through ONE v_mfma_scale_f32_16x16x128_f8f6f4 (e4m3 operands, 32 cycles)
// instead of four fp16 products (64 cycles): 64 instead of 96 matrix cycles per (element row, y position).  The table keeps fp16
// (re, im) of hi (rounded to nearest) and four e4m3 bytes [lo re, lo im | hi re, hi im] per entry (same two words as before); the
// Toeplitz weights carry [hi(c0), hi(c1), lo(c0), lo(c1)] * (2^-6, 2^5) in the same byte positions (toep_pack_k), and the instruction's
// E8M0 block scales (2^1, 2^0) undo the 2^-1 of each product.  A lane's 32 operand bytes are the entries 4 g .. 4 g + 3 and
// 16 + 4 g .. 16 + 4 g + 3 of the table row -- the two 16-byte pieces the hi fragments of the two K-steps sit at, so the e4m3 reads hit
// the same conflict-free slots.  The instruction spans both K-steps, so the block's two wave groups split the ELEMENT ROWS of a
// super-block (b = 0 .. 3 | 4 .. 7) instead of the K-steps; their partial sums meet in LDS as before.
// M2 (round 5; arrays up to 17 elements wide, i.e. every 16 x 16 array): TWO row tiles per block -- positions kx = 0 .. 7 and 8 .. 15 of the coset, the
// whole half axis on BASELINE's grids -- that share the block's tables AND its Toeplitz weights: A[(kx + 8, o), ud'] = A[(kx, o), ud' - 8], so the
// second tile is the same A fragment against the table row read 8 columns further on (32 bytes: the fragment reads stay aligned and
// conflict-free; what they overrun -- the first columns of the next row, the plane's pad -- meets zero weights and is finite because every
// word no generation round writes is cleared at block entry).  Same matrix instructions as two blocks of <= 8 positions; one table of <= 31 columns instead of two of
// <= 23, half the block prologues, barriers and weight loads per position.
// NM = 3 (round 6; arrays wider than 17 elements, e.g. BASELINE configs[3]): THREE row tiles -- 24 positions along x -- on 48-word table rows, one block per CU
// (ToepShape<3>, k_toep.hip.h): a table entry then serves 2.1 positions along x instead of one in four.
template <int MX, int MY, bool CLAMP, bool DIR = false, bool FP8 = false, int NM = 1>
__global__ __launch_bounds__(ToepShape<NM>::WAVES * 64, ToepShape<NM>::MINW) void field_toep_k(const uint4* __restrict__ afrag, float* __restrict__ pmag,
                                                                    float* __restrict__ inten, const CosetBlock* __restrict__ blocks /*[gridDim.x]*/,
                                                                    const ToepParams T) {
    using S = ToepShape<NM>;
    constexpr int TOEP_PSZ = S::PSZ, TOEP_TW = S::TW, TOEP_ROWS = S::ROWS, TOEP_WAVES = S::WAVES, NKY = S::NKY, KYG = S::KYG;
    constexpr bool M2 = NM >= 2;                         // more than one row tile: all of them are always computed
    const CosetParams& P = T.q;
    __shared__ __attribute__((aligned(16))) unsigned s_T[2 * TOEP_ZB * TOEP_PSZ];     // [hi | lo][plane][row][ud']
    unsigned* const s_hi = s_T;
    unsigned* const s_lo = s_T + TOEP_ZB * TOEP_PSZ;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int kyg = wave % KYG, ks = wave / KYG;              // ks: this wave's K-step -- FP8: its half of the super-block's element rows
    // the block's share of the coset decomposition: one scalar load of the host's record (olx.hip, as kernel 2g; the two blocks
    // that write the two 64-byte halves of the same 128-byte lines have ids 8 apart -- the same XCD, i.e. the same L2, under
    // round-robin dispatch).  Decoded here, the chain of integer divisions was ~160 vector and ~400 scalar instructions per wave.
    // NM = 3 (one block per CU: nobody covers a block's store drain, launch and first loads): the grid is one block per CU and a block WALKS the records
    // blockIdx.x, + gridDim.x, ... (gridDim.x is a multiple of 8: a block keeps to the records of its XCD, in the host's order) -- the stores of a record
    // drain under the next record's table generation.
    constexpr bool WALK = NM == 3;
    unsigned rec = blockIdx.x;
    do {                                                // (one pass where WALK is false: the loop form alone costs the 128-register shapes 15 - 22 spills)
    if (WALK && rec != blockIdx.x) lds_barrier();       // the previous record's exchange tiles are read: the arena is free (LDS only: its stores drain under this record's tables)
    const CosetBlock BK = blocks[rec];
    const int KX = BK.KX, KY = BK.KY;
    if (BK.npos <= 0) { if constexpr (WALK) continue; else return; }      // block-uniform
    const int ibase = BK.ibase, jbase = BK.jbase;
    const int k0 = BK.k0;
    const int SAW = T.sa_w;                             // elements of a super-block along x
    const int NC = SAW + P.xs * (KX - 1);               // table columns in use: ud' = xs kx - al + (SAW - 1) in [0, NC)
    const int NR = KY + TOEP_SB - 1;                    // table rows in use: wd = -7 .. KY - 1
    // What no generation round writes and the fragment reads of a STORED position meet must be finite (it meets zero Toeplitz weights: 0 x garbage must stay 0; the
    // arena holds the previous block's exchange tiles): the columns NC .. TW - 1 of every row, the rows NR .. ROWS - 1 (a second row tile reads 8 columns past its
    // row into the next one; the ring of rows later rotates generated rows over them) and the plane's pad.  16-byte stores over (half, plane) x row items --
    // 1.3 k of them on BASELINE's 16 x 16 shapes instead of the whole arena's 4.7 k (profiles/r06_toep_phases.txt: the whole-arena clear was 7 % of the launch).
    // (Rows of y positions beyond KY -- computed by a wave that also holds stored ones, never stored -- may read anything.)
    {
        const int nc4 = NC & ~3;
        for (int idx = tid; idx < 32 * (TOEP_ROWS + 1); idx += TOEP_WAVES * 64) {
            const int hz = idx & 31, row = idx >> 5;            // (hi | lo half, plane), table row -- row ROWS = the pad behind the last row
            unsigned* const p = s_T + hz * TOEP_PSZ + row * TOEP_TW;
            const int c1 = row == TOEP_ROWS ? TOEP_PSZ - TOEP_ROWS * TOEP_TW : TOEP_TW;
            for (int c = row < NR ? nc4 : 0; c < c1; c += 4) *reinterpret_cast<uint4*>(p + c) = make_uint4(0u, 0u, 0u, 0u);
        }
    }
    // dz^2 of the block's 16 planes: wave-uniform, held in scalar registers (one v_add per evaluation instead of two fmas)
    float dz2[TOEP_ZB];
#pragma unroll
    for (int z = 0; z < TOEP_ZB; ++z) {
        const float dz = (float)(k0 + z) * P.hz - P.flat_ez;
        dz2[z] = __builtin_bit_cast(float, __builtin_amdgcn_readfirstlane(__builtin_bit_cast(int, dz * dz)));
    }
    // B fragment base of this lane [words]: plane (lane & 15), k-group (lane >> 4), this wave's K-step
    const int n16 = lane & 15, g = lane >> 4;
    const unsigned bbase = (unsigned)(n16 * TOEP_PSZ + 4 * g + (FP8 ? 0 : 16 * ks));
    (void)KX;                                           // (row tiles: positions 8 m .. 8 m + 7)
    floatx4_t acc[NM][NKY];
#pragma unroll
    for (int m = 0; m < NM; ++m)
#pragma unroll
        for (int t = 0; t < NKY; ++t) acc[m][t] = floatx4_t{0.f, 0.f, 0.f, 0.f};
    const int n_sb = T.nsa * P.nsb;
    OLX_STAMP(0);
    for (int sb = 0; sb < n_sb; ++sb) {
        const int sa = sb / P.nsb, sbb = sb - sa * P.nsb;
        // Toeplitz weights of the super-block's 8 element rows (this wave's K-step: hi + lo = 2 x 16 bytes per lane and row),
        // requested before the table is generated so that they arrive from L2 behind it
        const uint4* ab = afrag + ((size_t)(blockIdx.y * T.nsa + sa) * T.ay_pad + TOEP_SB * sbb + (FP8 ? 4 * ks : 0)) * 4 * 64 + (FP8 ? 0 : ks * 64) + lane;
        constexpr int NB = FP8 ? TOEP_SB / 2 : TOEP_SB, NA = FP8 ? 4 : 2;      // element rows per wave and super-block; 16-byte weight pieces per row
        uint4 afr[NB][NA];
        __syncthreads();                                // table free (previous super-block consumed)
        if (sb == 0) OLX_STAMP(1);
        auto load_weights = [&]() __attribute__((always_inline)) {
#pragma unroll
            for (int bl = 0; bl < NB; ++bl) {
                if constexpr (FP8) {       // hi of both K-steps, then the 32 e4m3 bytes of the row
#pragma unroll
                    for (int q = 0; q < 4; ++q) afr[bl][q] = ab[(bl * 4 + q) * 64];
                } else { afr[bl][0] = ab[bl * 4 * 64]; afr[bl][1] = ab[(bl * 4 + 2) * 64]; }
            }
        };
        // (requested behind the barrier: __syncthreads() drains vmcnt, the loads would be waited for right there; LDS-only barriers here and behind the fill: +- 0)
        load_weights();
        // ---- G tables of the 16 planes: (row, column) pairs across the threads
        // Sliding rows: logical row r of super-block (sa, sbb) is the offset wd = r - 7 - 8 sbb against element row 0 of the
        // column of super-blocks, so rows 8 .. NR - 1 of super-block sbb + 1 ARE rows 0 .. NR - 9 of super-block sbb.  The table is a
        // ring of 18 physical rows, phys(r) = (r - 8 sbb) mod 18: only the 8 new rows are evaluated for sbb >= 1 (they overwrite the
        // 8 rows the previous super-block no longer needs) -- 18 + 8 (nsb - 1) instead of 18 nsb rows per column of super-blocks.
        const int n_new = sbb == 0 ? NR : min(NR, TOEP_SB);
        const int rot = (TOEP_ROWS * 64 - TOEP_SB * sbb) % TOEP_ROWS;         // phys(r) = (r + rot) mod 18
        // columns that meet a weight in this column of super-blocks: ud' = xs kx - al + (SAW - 1) with al < (elements of the column) -- a narrow LAST
        // column (32 elements = 24 + 8) uses the upper ones only; the others keep the previous column's entries (finite: 0 x them stays 0)
        const int c_lo = SAW - min(SAW, T.ax - SAW * sa), ncol = NC - c_lo;
        const float inv_ncol = 1.0f / (float)ncol;
        auto fill = [&](const int idx, auto z0_c, auto zn_c) __attribute__((always_inline)) {
            constexpr int Z0 = decltype(z0_c)::value, ZN = decltype(zn_c)::value;
            const int row = (int)(((float)idx + 0.5f) * inv_ncol), col = c_lo + idx - row * ncol;      // exact for these small integers
            const float U = (float)(ibase + P.x_begin + P.ux0 + P.mx * (col - (SAW - 1)) - SAW * P.mx * sa);
            const float W = (float)(jbase + P.uy0 + P.my * (row - 7) - TOEP_SB * P.my * sbb);
            const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
            const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
            const float r2 = fmaf(dy, dy, dx * dx);
            int prow = row + rot;
            prow = prow >= TOEP_ROWS ? prow - TOEP_ROWS : prow;
            const int o = prow * TOEP_TW + col;
#pragma unroll
            for (int z = Z0; z < Z0 + ZN; ++z) {
                float d2 = r2 + dz2[z];
                if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                const float ri = __builtin_amdgcn_rsqf(d2);
                const float ph = d2 * ri;
                float rs = ri * P.g_scale;
                if constexpr (DIR) rs *= table_mod(dx, dy, ph, ri, P.dir_wx, P.dir_wy, P.absorb_l2);      // (own instantiations: the default path never sees this)
                float gr = rs * __builtin_amdgcn_cosf(ph), gi = rs * __builtin_amdgcn_sinf(ph);
                asm volatile("" : "+v"(gr), "+v"(gi));      // (two plain multiplies: a packed one measured slower in kernel 2g's fill)
                half2_t hi;
                if constexpr (FP8) hi = __builtin_convertvector(float2_t{gr, gi}, half2_t);      // to nearest: |lo| <= half an ulp
                else hi = __builtin_bit_cast(half2_t, __builtin_amdgcn_cvt_pkrtz(gr, gi));
                if (!OLX_IN(z * TOEP_PSZ + o, TOEP_ZB * TOEP_PSZ, 4)) continue;
                s_hi[z * TOEP_PSZ + o] = __builtin_bit_cast(unsigned, hi);
                // lo = g - (float)hi in ONE mixed-precision fma per component (the compiler's form: a convert and a subtract)
                float lr, li;
                const unsigned hw = __builtin_bit_cast(unsigned, hi);
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
                asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
                if constexpr (FP8) {   // e4m3 bytes [lo re, lo im | hi re, hi im] (the scaled convert DIVIDES by its scale operand: k_coset2.hip)
                    short2_t w;
                    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, lr, li, 1.0f / COS_F8_LO, false);
                    w = __builtin_amdgcn_cvt_scalef32_pk_fp8_f32(w, gr, gi, 1.0f / COS_F8_HI, true);
                    s_lo[z * TOEP_PSZ + o] = __builtin_bit_cast(unsigned, w);
                } else s_lo[z * TOEP_PSZ + o] = __builtin_bit_cast(unsigned, __builtin_amdgcn_cvt_pkrtz(lr, li));
            }
        };
        // (where one round leaves half of the block idle -- <= 256 pairs: every super-block after the first -- splitting a pair's 16 planes over the
        // two wave groups measured +1.5 % on configs[3], as in round 5: idle waves cost nothing, the partner block's waves take the SIMDs)
        // (a wave taking four of the 16 planes of its entries, so that a super-block's 248 - 376 new entries fill whole rounds of 128 threads: +4 ... +8 % on every
        // shape -- the index and dx^2 + dy^2 arithmetic repeats per quarter, and what counts is the instructions issued, not the idle lanes: profiles/r06_toep_phases.txt)
        for (int idx = tid; idx < n_new * ncol; idx += TOEP_WAVES * 64) fill(idx, IntC<0>{}, IntC<TOEP_ZB>{});
        if (sb == 0) OLX_STAMP(2);
        __syncthreads();
        if (sb == 0) OLX_STAMP(3);
        // ---- contraction: element rows b of the super-block, this wave's y positions and K-step
        // Round 6: a wave owns THREE CONSECUTIVE y positions ky = 3 kyg + t, and the (y position, element row) pairs are walked along their
        // diagonals d = t - bi: every pair of a diagonal meets the SAME table row wd = ky - b, so its B fragments are read once and serve up to
        // three matrix-instruction groups (48 instead of 96 ds_read_b128 per wave and super-block with two row tiles: the fragment reads had made
        // the LDS, not the matrix pipe, the busiest unit of this kernel -- 54 % against 33 %).  The schedule is static -- no wave-uniform branch
        // inside, the K-step mask and the second row tile are compile-time cases -- so the compiler can issue a diagonal's reads under the
        // previous diagonal's matrix instructions instead of waiting out every read (round 5: one s_waitcnt lgkmcnt(0) per group).  A y position
        // beyond KY reads rows that exist (the ring has 18) and its sums are never stored.
        const unsigned ksm = (T.ks_mask >> (2 * sa)) & 3u;      // K-steps of this column of super-blocks that carry weights (wave-uniform)
        if (!FP8 && !((ksm >> ks) & 1u)) continue;              // (fp16 corrections: this wave's K-step is all zeros here; the barriers sit above)
        if (NKY * kyg >= KY) continue;                            // (no y position of this wave exists: wave-uniform)
        const int p0 = NKY * kyg + 7 - (FP8 ? 4 * ks : 0) + rot + TOEP_ROWS;      // logical table row of (t = 0, bi = 0) + ROWS (p0 + d stays in [0, 3 ROWS))
        auto contract = [&](auto two_c, auto ksm_c) __attribute__((always_inline)) {
            constexpr int NMM = decltype(two_c)::value;
            constexpr unsigned KSM = decltype(ksm_c)::value;
            constexpr int ND = NB + NKY - 1, NSTEP = NMM * ND;      // diagonals d = -(NB - 1) .. NKY - 1 per row tile
            if constexpr (NMM == 3) {
                // three row tiles: a diagonal's fragments for ALL tiles in one step -- 3 - 6 independent accumulators per step (a tile alone has one or two: the
                // matrix instructions of an accumulator are a dependent chain), and tile m's K-step s sits 8 m + 16 s words into the row, so tile 2's
                // K-step 0 IS tile 0's K-step 1: 5 pieces x (hi, e4m3) = 10 ds_read_b128 per diagonal instead of 12
                constexpr int NP = FP8 ? 5 : 3;
                struct BF3 { uint4 h[NP], q[NP]; };          // FP8: piece j = words 8 j .. of the row (hi fp16 | e4m3 bytes); fp16 x 3: tile j, this wave's K-step (hi | lo)
                auto load3 = [&](const int di) __attribute__((always_inline)) {
                    int prow = p0 + di - (NB - 1);           // physical row of the ring (scalar arithmetic: wave-uniform)
                    prow = prow >= 2 * TOEP_ROWS ? prow - 2 * TOEP_ROWS : (prow >= TOEP_ROWS ? prow - TOEP_ROWS : prow);
                    const unsigned wm = bbase + (unsigned)(prow * TOEP_TW);
                    BF3 f;
#pragma unroll
                    for (int j = 0; j < NP; ++j) {
                        if (FP8 && KSM == 2u && j < 2) f.h[j] = make_uint4(0u, 0u, 0u, 0u);      // (K-step 0 carries no weights here: its hi pieces are not read)
                        else f.h[j] = *reinterpret_cast<const uint4*>(s_hi + wm + 8 * j);
                        f.q[j] = *reinterpret_cast<const uint4*>(s_lo + wm + 8 * j);
                    }
                    return f;
                };
                if constexpr (FP8 && KSM == 2u) {
                    // A narrow LAST column of super-blocks (32 elements = 24 + 8) carries weights in K-step 1 only: half of every K = 128 e4m3 instruction met zeros.
                    // Its element rows are taken in PAIRS instead: the instruction's K-step-0 half gets row b + 1 -- its K-step-1 weight bytes against the K-step-1
                    // piece of ITS table row, the previous diagonal's (the byte order inside a half is the same for both halves) -- so a pair costs 2 fp16 products +
                    // one e4m3 instruction instead of 2 + 2: 4 matrix units per two rows instead of 6, and 6 instead of 10 fragment reads per diagonal.
                    static_assert(NB % 2 == 0, "element rows in pairs");
                    struct BP { uint4 h[3], q[3]; };             // the K-step-1 pieces of the three tiles: words 8 (m + 2) .. of the row
                    auto loadp = [&](const int di) __attribute__((always_inline)) {
                        int prow = p0 + di - (NB - 1);
                        prow = prow >= 2 * TOEP_ROWS ? prow - 2 * TOEP_ROWS : (prow >= TOEP_ROWS ? prow - TOEP_ROWS : prow);
                        const unsigned wm = bbase + (unsigned)(prow * TOEP_TW) + 16u;
                        BP f;
#pragma unroll
                        for (int m = 0; m < 3; ++m) { f.h[m] = *reinterpret_cast<const uint4*>(s_hi + wm + 8 * m); f.q[m] = *reinterpret_cast<const uint4*>(s_lo + wm + 8 * m); }
                        return f;
                    };
                    BP prv = loadp(0), cur = loadp(1);
#pragma unroll
                    for (int di = 1; di < ND; ++di) {            // the even row b of a pair sits on diagonal d, its partner b + 1 on d - 1
                        const int d = di - (NB - 1);
                        BP nxt = cur;
                        if (di + 1 < ND) nxt = loadp(di + 1);
#pragma unroll
                        for (int pass = 0; pass < 3; ++pass)
#pragma unroll
                            for (int t = 0; t < NKY; ++t) {
                                const int bi = t - d;
                                if (bi < 0 || bi >= NB || (bi & 1)) continue;    // (compile-time)
#pragma unroll
                                for (int m = 0; m < 3; ++m) {
                                    Half8Bits a0, a1, b0, b1;
                                    a0.u = afr[bi][1]; a1.u = afr[bi + 1][1]; b0.u = cur.h[m]; b1.u = prv.h[m];
                                    if (pass == 0) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0.h, b0.h, acc[m][t], 0, 0, 0);
                                    else if (pass == 1) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1.h, b1.h, acc[m][t], 0, 0, 0);
                                    else {
                                        intx8_t a8, b8;
                                        a8[0] = (int)afr[bi + 1][3].x; a8[1] = (int)afr[bi + 1][3].y; a8[2] = (int)afr[bi + 1][3].z; a8[3] = (int)afr[bi + 1][3].w;
                                        a8[4] = (int)afr[bi][3].x; a8[5] = (int)afr[bi][3].y; a8[6] = (int)afr[bi][3].z; a8[7] = (int)afr[bi][3].w;
                                        b8[0] = (int)prv.q[m].x; b8[1] = (int)prv.q[m].y; b8[2] = (int)prv.q[m].z; b8[3] = (int)prv.q[m].w;
                                        b8[4] = (int)cur.q[m].x; b8[5] = (int)cur.q[m].y; b8[6] = (int)cur.q[m].z; b8[7] = (int)cur.q[m].w;
                                        acc[m][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[m][t], 0, 0, 0, 128, 0, 127);
                                    }
                                }
                            }
                        __builtin_amdgcn_sched_barrier(0);
                        prv = cur; cur = nxt;
                    }
                    return;
                }
                BF3 cur3 = load3(0);
#pragma unroll
                for (int di = 0; di < ND; ++di) {
                    const int d = di - (NB - 1);
                    BF3 nxt3 = cur3;
                    if (di + 1 < ND) nxt3 = load3(di + 1);   // one diagonal ahead (256 registers per lane in this shape)
                    const uint4* fh = cur3.h; const uint4* fq = cur3.q;
#pragma unroll
                    for (int pass = 0; pass < 3; ++pass)     // the three products of a group one after the other ACROSS the groups
#pragma unroll
                        for (int t = 0; t < NKY; ++t) {
                            const int bi = t - d;
                            if (bi < 0 || bi >= NB) continue;    // (compile-time)
#pragma unroll
                            for (int m = 0; m < 3; ++m) {
                                Half8Bits ah, al, bh, bw;
                                ah.u = afr[bi][0]; al.u = afr[bi][1];
                                if constexpr (FP8) {
                                    bh.u = fh[m]; bw.u = fh[m + 2];
                                    if (pass == 0) { if constexpr ((KSM & 1u) != 0) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh.h, acc[m][t], 0, 0, 0); }
                                    else if (pass == 1) { if constexpr ((KSM & 2u) != 0) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bw.h, acc[m][t], 0, 0, 0); }
                                    else {
                                        intx8_t a8, b8;
                                        a8[0] = (int)afr[bi][2].x; a8[1] = (int)afr[bi][2].y; a8[2] = (int)afr[bi][2].z; a8[3] = (int)afr[bi][2].w;
                                        a8[4] = (int)afr[bi][3].x; a8[5] = (int)afr[bi][3].y; a8[6] = (int)afr[bi][3].z; a8[7] = (int)afr[bi][3].w;
                                        b8[0] = (int)fq[m].x; b8[1] = (int)fq[m].y; b8[2] = (int)fq[m].z; b8[3] = (int)fq[m].w;
                                        b8[4] = (int)fq[m + 2].x; b8[5] = (int)fq[m + 2].y; b8[6] = (int)fq[m + 2].z; b8[7] = (int)fq[m + 2].w;
                                        acc[m][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[m][t], 0, 0, 0, 128, 0, 127);
                                    }
                                } else {
                                    bh.u = fh[m]; bw.u = fq[m];
                                    if (pass == 0) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh.h, acc[m][t], 0, 0, 0);
                                    else if (pass == 1) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bh.h, acc[m][t], 0, 0, 0);
                                    else acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bw.h, acc[m][t], 0, 0, 0);
                                }
                            }
                        }
                    __builtin_amdgcn_sched_barrier(0);       // (one diagonal ahead, not more)
                    cur3 = nxt3;
                }
                return;
            }
            if constexpr (NMM == 2 && FP8) {
                // Arrays up to 17 elements wide: a row of 8 positions meets at most 7 table offsets in K-step 1 (ud' = kx - al + sa_w - 1 <= 23), i.e. 14 of its 32 K
                // slots.  Two element rows b, b + 1 share ONE K-step-1 operand instead: k-groups 0, 1 hold row b's offsets 16 .. 23 against ITS table row (this
                // diagonal's), k-groups 2, 3 row b + 1's against the previous diagonal's -- a per-lane row in the fragment address, the weights packed to match
                // (toep_pack_k, pair_k1).  Per y position and FOUR element rows: 4 + 2 fp16 products and THREE e4m3 instructions whose halves are
                //   [K0 b0 | K1 (b0, b1)]   [K0 b1 | K0 b2]   [K0 b3 | K1 (b2, b3)]
                // (a half's bytes follow the same order in both operands, whichever rows it holds) = 12 matrix units instead of 16; every half meets fragments of this
                // diagonal or the previous one, whose e4m3 piece stays in four registers.  Fragment reads per diagonal as before (four 16-byte pieces).
                static_assert(NB == 4 && NKY == 3, "rows in quads, three y positions per wave");
                const unsigned pofs = (unsigned)(n16 * TOEP_PSZ + 16 + 4 * (g & 1));
                const bool odd_row = g >= 2;
                struct BF2 { uint4 h0, q0, ph, pq; };        // this diagonal's K-step-0 pieces (hi fp16 | e4m3), the paired K-step-1 pieces of (this | previous) diagonal
                auto load2 = [&](const int step) __attribute__((always_inline)) {
                    const int m = step / ND, d = step - m * ND - (NB - 1);
                    int prow = p0 + d, prow1 = p0 + d - 1;   // physical rows of the ring (scalar arithmetic: wave-uniform)
                    prow = prow >= 2 * TOEP_ROWS ? prow - 2 * TOEP_ROWS : (prow >= TOEP_ROWS ? prow - TOEP_ROWS : prow);
                    prow1 = prow1 >= 2 * TOEP_ROWS ? prow1 - 2 * TOEP_ROWS : (prow1 >= TOEP_ROWS ? prow1 - TOEP_ROWS : prow1);
                    const unsigned wm = bbase + (unsigned)(prow * TOEP_TW) + 8u * (unsigned)m;
                    BF2 f;
                    f.h0 = *reinterpret_cast<const uint4*>(s_hi + wm); f.q0 = *reinterpret_cast<const uint4*>(s_lo + wm);
                    if (d >= -(NB - 2)) {                    // (compile-time: the first diagonal holds the last row of a quad only)
                        const unsigned wp = pofs + (unsigned)((odd_row ? prow1 : prow) * TOEP_TW) + 8u * (unsigned)m;
                        f.ph = *reinterpret_cast<const uint4*>(s_hi + wp); f.pq = *reinterpret_cast<const uint4*>(s_lo + wp);
                    } else { f.ph = make_uint4(0u, 0u, 0u, 0u); f.pq = f.ph; }
                    return f;
                };
                auto e4 = [&](floatx4_t c, const uint4 a_lo, const uint4 a_hi, const uint4 b_lo, const uint4 b_hi) __attribute__((always_inline)) {
                    intx8_t a8, b8;
                    a8[0] = (int)a_lo.x; a8[1] = (int)a_lo.y; a8[2] = (int)a_lo.z; a8[3] = (int)a_lo.w; a8[4] = (int)a_hi.x; a8[5] = (int)a_hi.y; a8[6] = (int)a_hi.z; a8[7] = (int)a_hi.w;
                    b8[0] = (int)b_lo.x; b8[1] = (int)b_lo.y; b8[2] = (int)b_lo.z; b8[3] = (int)b_lo.w; b8[4] = (int)b_hi.x; b8[5] = (int)b_hi.y; b8[6] = (int)b_hi.z; b8[7] = (int)b_hi.w;
                    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, c, 0, 0, 0, 128, 0, 127);      // (E8M0 scales as below)
                };
                BF2 cur = load2(0);
                uint4 qprev = make_uint4(0u, 0u, 0u, 0u);    // the previous diagonal's K-step-0 e4m3 piece
#pragma unroll
                for (int step = 0; step < NSTEP; ++step) {
                    const int m = step / ND, d = step - m * ND - (NB - 1);
                    __builtin_amdgcn_sched_barrier(0);
#pragma unroll
                    for (int t = 0; t < NKY; ++t) {
                        const int bi = t - d;                // element row of the quad: (t, bi) on the diagonal d
                        if (bi < 0 || bi >= NB) continue;    // (compile-time)
                        Half8Bits a0, b0;
                        a0.u = afr[bi][0]; b0.u = cur.h0;
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a0.h, b0.h, acc[m][t], 0, 0, 0);          // K-step 0, hi x hi
                        if (bi == 0 || bi == 2) {            // K-step 1 of the pair (bi, bi + 1), hi x hi
                            Half8Bits a1, b1;
                            a1.u = afr[bi][1]; b1.u = cur.ph;
                            acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a1.h, b1.h, acc[m][t], 0, 0, 0);
                        }
                        if (bi == 0) acc[m][t] = e4(acc[m][t], afr[0][2], afr[0][3], cur.q0, cur.pq);
                        else if (bi == 1) acc[m][t] = e4(acc[m][t], afr[1][2], afr[2][2], cur.q0, qprev);
                        else if (bi == 2) acc[m][t] = e4(acc[m][t], afr[3][2], afr[2][3], qprev, cur.pq);
                    }
                    __builtin_amdgcn_sched_barrier(0);
                    qprev = cur.q0;
                    if (step + 1 < NSTEP) cur = load2(step + 1);
                }
                return;
            }
            // B fragments of one diagonal: hi of this wave's K-step(s) and -- FP8 -- the 32 e4m3 bytes of the table row's two pieces
            struct BF { uint4 h0, h1, q0, q1; };
            auto load = [&](const int step) __attribute__((always_inline)) {
                const int m = step / ND, d = step - m * ND - (NB - 1);
                int prow = p0 + d;                           // physical row of the ring (scalar arithmetic: wave-uniform)
                prow = prow >= 2 * TOEP_ROWS ? prow - 2 * TOEP_ROWS : (prow >= TOEP_ROWS ? prow - TOEP_ROWS : prow);
                const unsigned wm = bbase + (unsigned)(prow * TOEP_TW) + 8u * (unsigned)m;       // second row tile: the same weights, 8 columns on
                BF f;
                f.h0 = *reinterpret_cast<const uint4*>(s_hi + wm);
                if constexpr (FP8) {
                    f.h1 = *reinterpret_cast<const uint4*>(s_hi + wm + 16);                      // hi, K-step 1
                    f.q0 = *reinterpret_cast<const uint4*>(s_lo + wm); f.q1 = *reinterpret_cast<const uint4*>(s_lo + wm + 16);
                } else { f.h1 = *reinterpret_cast<const uint4*>(s_lo + wm); f.q0 = f.h1; f.q1 = f.h1; }
                return f;
            };
            // one diagonal ahead with one row tile (12 accumulator registers); with two, accumulators + weights + two sets of fragments fill all
            // 128 registers for no gain (same box, alternating: 0.0695 either way) -- the block's other wave on the SIMD covers the read
            constexpr bool AHEAD = !M2;
            BF cur = load(0);
#pragma unroll
            for (int step = 0; step < NSTEP; ++step) {
                const int m = step / ND, d = step - m * ND - (NB - 1);
                BF nxt = cur;
                if constexpr (AHEAD) { if (step + 1 < NSTEP) nxt = load(step + 1); }      // the next diagonal's reads go out under this one's matrix instructions
                if constexpr (!AHEAD) __builtin_amdgcn_sched_barrier(0);
                Half8Bits bh, bw;
                bh.u = cur.h0; bw.u = cur.h1;
                intx8_t b8;
                if constexpr (FP8) {
                    b8[0] = (int)cur.q0.x; b8[1] = (int)cur.q0.y; b8[2] = (int)cur.q0.z; b8[3] = (int)cur.q0.w;
                    b8[4] = (int)cur.q1.x; b8[5] = (int)cur.q1.y; b8[6] = (int)cur.q1.z; b8[7] = (int)cur.q1.w;
                }
#pragma unroll
                for (int t = 0; t < NKY; ++t) {
                    const int bi = t - d;                    // element row of this wave's share: (t, bi) on the diagonal d
                    if (bi < 0 || bi >= NB) continue;        // (compile-time)
                    Half8Bits ah, al;
                    ah.u = afr[bi][0]; al.u = afr[bi][1];    // (FP8: hi of K-step 0 and 1)
                    if constexpr (FP8) {
                        intx8_t a8;
                        a8[0] = (int)afr[bi][2].x; a8[1] = (int)afr[bi][2].y; a8[2] = (int)afr[bi][2].z; a8[3] = (int)afr[bi][2].w;
                        a8[4] = (int)afr[bi][3].x; a8[5] = (int)afr[bi][3].y; a8[6] = (int)afr[bi][3].z; a8[7] = (int)afr[bi][3].w;
                        if constexpr ((KSM & 1u) != 0) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh.h, acc[m][t], 0, 0, 0);
                        if constexpr ((KSM & 2u) != 0) acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bw.h, acc[m][t], 0, 0, 0);
                        // E8M0 scales undo the operand scaling: 2^(128 - 127) * COS_F8_LO * COS_F8_HI = 1
                        acc[m][t] = __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(a8, b8, acc[m][t], 0, 0, 0, 128, 0, 127);
                    } else {
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bh.h, acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al.h, bh.h, acc[m][t], 0, 0, 0);
                        acc[m][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah.h, bw.h, acc[m][t], 0, 0, 0);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);           // (one diagonal ahead, not more: the fragments of all of them would not fit the registers)
                if constexpr (AHEAD) cur = nxt;
                else if (step + 1 < NSTEP) cur = load(step + 1);
            }
        };
        if constexpr (NM == 2) {     // (arrays up to 17 wide: one column of super-blocks, both K-steps carry weights)
            contract(IntC<2>{}, IntC<3>{});      // (both row tiles always: the host picks this shape only where some part has more than 8 positions -- a second contraction body beside it costs 34 spilled registers)
        } else if constexpr (FP8) {
            if (ksm == 3u) contract(IntC<NM>{}, IntC<3>{}); else if (ksm == 2u) contract(IntC<NM>{}, IntC<2>{}); else contract(IntC<NM>{}, IntC<1>{});
        } else contract(IntC<NM>{}, IntC<3>{});
        if (sb == 0) OLX_STAMP(4);
    }
    OLX_STAMP(5);
    // ---- the two K-step halves of a y-position group meet in LDS (the table arena is free now); tile = 16 rows (kx, re | im)
    // x 16 planes.  D layout: lane (g, n16) holds rows 4 g .. 4 g + 3 of plane n16.
    __syncthreads();
    float* const s_x = reinterpret_cast<float*>(s_T);
    {
        float* xo = s_x + (wave * NKY * NM) * 16 * TOEP_XS + n16;
#pragma unroll
        for (int t = 0; t < NKY; ++t)
#pragma unroll
            for (int m = 0; m < NM; ++m)
#pragma unroll
                for (int r = 0; r < 4; ++r) xo[((t * NM + m) * 16 + 4 * g + r) * TOEP_XS] = acc[m][t][r];
    }
    __syncthreads();
    // read-out: the (y position, row tile) pairs q = t NM + m of a y-position group are split evenly between its two waves (group 0: the first
    // half).  Lane = (output, kx, plane quad): four consecutive planes of one position -> one 16-byte store per target (64 contiguous bytes per four lanes)
    const int out = lane >> 5, kx = (lane >> 2) & 7, pq = lane & 3;
    const int kz = k0 + 4 * pq;
    const float sc = out ? P.out_scale * P.out_scale * P.inten_scale : P.out_scale;
    float* const vol = out ? inten : pmag;
    const bool want0 = (P.flags & (out ? 2u : 1u)) != 0 && kz < P.nz;
    // store addresses: the targets (focus, mirror image) are block-uniform, so per lane only the two x forms of the voxel
    // offset are formed once; per (y position, target) the y term is scalar: one add + one 64-bit add per 16-byte store
    const int sxz = P.ny * P.nz;
    // (the ragged-nz variant is a separate copy of the loop: with both store forms in one body the compiler merges them and
    // splits every 16-byte store into a 12-byte and a 4-byte instruction)
    auto readout = [&](auto full_c) {
        constexpr bool FULL4 = decltype(full_c)::value != 0;
        constexpr int NQ = NKY * NM, QH = (NQ + 1) / 2;     // pairs per group; the first wave takes QH of them
#pragma unroll
        for (int qq = 0; qq < QH; ++qq) {
            const int q = ks ? QH + qq : qq;
            if (q >= NQ) break;
            const int t = q / NM, m = q - t * NM;
            const int ky = NKY * kyg + t;
            const int kxg = 8 * m + kx;                      // position of the coset along x
            const bool want = want0 && kxg < KX;
            if (ky >= KY || !want) continue;
            const int i = ibase + P.xs * P.mx * kxg;
            const unsigned ox0 = (unsigned)(i * sxz + kz), ox1 = (unsigned)((P.nx - 1 - i) * sxz + kz);
            const float* xa = s_x + (((kyg * NKY + t) * NM + m) * 16 + 2 * kx) * TOEP_XS + 4 * pq;   // first wave group's partial
            const float* xb = xa + KYG * NKY * NM * 16 * TOEP_XS;                                // second wave group's partial (wave + KYG)
            const float4 ra = *reinterpret_cast<const float4*>(xa), ia = *reinterpret_cast<const float4*>(xa + TOEP_XS);
            const float4 rb = *reinterpret_cast<const float4*>(xb), ib = *reinterpret_cast<const float4*>(xb + TOEP_XS);
            const float re[4] = {ra.x + rb.x, ra.y + rb.y, ra.z + rb.z, ra.w + rb.w};
            const float im[4] = {ia.x + ib.x, ia.y + ib.y, ia.z + ib.z, ia.w + ib.w};
            float v[4];
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float m2 = fmaf(re[e], re[e], im[e] * im[e]);
                v[e] = (out ? m2 : __builtin_amdgcn_sqrtf(m2)) * sc;
            }
            const int j = jbase + P.my * ky;                 // wave-uniform
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int code = T.targets[q];
                if (code < 0) continue;                     // uniform
                const int m = code & 3;
                const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                const unsigned oy = (unsigned)((fy ? (P.ny - 1 - j) : j) * P.nz);                 // scalar
                float* dst = vol + (long long)(code >> 2) * P.vox + ((fx ? ox1 : ox0) + oy);      // (32-bit offset within the focus volume)
                if (!OLX_IN((long long)(code >> 2) * P.vox + ((fx ? ox1 : ox0) + oy) + (FULL4 ? 3 : 0), (long long)P.n_foci * P.vox, 5)) continue;
                if constexpr (FULL4) {
                    // the walking shape stores non-temporally: its blocks overflow the XCD's L2 between the two halves of a 128-byte line (19.6 M instead of 16.8 M
                    // write requests per launch on configs[3]) -- 1.263 -> 1.215 ms, same box, alternating.  The two-tile shapes keep plain stores: non-temporal ones
                    // gain 1.5 % at 128^3 and LOSE 5 % at 256^3, where the lines do meet in L2 (profiles/r06_toep_phases.txt (11)).
                    if constexpr (NM == 3) __builtin_nontemporal_store(floatx4u_t{v[0], v[1], v[2], v[3]}, reinterpret_cast<floatx4u_t*>(dst));
                    else *reinterpret_cast<floatx4u_t*>(dst) = floatx4u_t{v[0], v[1], v[2], v[3]};
                }
                else {
#pragma unroll
                    for (int e = 0; e < 4; ++e) if (kz + e < P.nz) dst[e] = v[e];
                }
            }
        }
    };
    if (k0 + TOEP_ZB <= P.nz) readout(IntC<1>{}); else readout(IntC<0>{});      // (block-uniform: only a LAST, partial plane block stores plane by plane)
    OLX_STAMP(6);
    } while (WALK && (rec += gridDim.x) < T.n_rec);      // records
}

// Toeplitz weights of kernel 2f in MFMA lane order: afrag[(((tile nsa + sa) ay_pad + b) 4 + {hi s0, hi s1, lo s0, lo s1}) 64 + lane].
// Lane (m = lane & 15 -> kx = m >> 1, o = m & 1; k-group g): k = 32 s + 8 g + jj -> offset ud' = k >> 1, part c = k & 1,
// element column a = sa_w sa + xs kx - (ud' - (sa_w - 1)).  grid (nsa * ay_pad, tiles), block 64.
__global__ void toep_pack_k(const double* __restrict__ area, int n, const double* __restrict__ delays, const double* __restrict__ apod,
                            const int* __restrict__ perm, double freq, double w_scale, int n_foci,
                            const int* __restrict__ colinfo /*[tiles][32][2]*/, const int* __restrict__ cell /*[ax][ay] -> element*/,
                            int ax, int ay, int ay_pad, int fp8corr /*1: the two lo pieces hold the row's e4m3 bytes instead*/, int sa_w, int xs,
                            int pair_k1 /*1: the K-step-1 pieces of an EVEN row hold rows (b | b + 1), offsets 16 .. 23, in k-groups (0, 1 | 2, 3)*/, uint4* __restrict__ afrag) {
    const int lane = threadIdx.x, tile = blockIdx.y;
    const int sa = blockIdx.x / ay_pad, b = blockIdx.x - sa * ay_pad;
    const int m = lane & 15, g = lane >> 4, kx = m >> 1, o = m & 1;
    const int f = colinfo[(size_t)tile * (MFMA_COLS * MFMA_MAX_NT) * 2], mirror = colinfo[(size_t)tile * (MFMA_COLS * MFMA_MAX_NT) * 2 + 1];
    uint4* dst = afrag + ((size_t)(tile * gridDim.x + blockIdx.x)) * 4 * 64;
    for (int s = 0; s < 2; ++s) {
        Half8Bits hi, lo;
#pragma unroll
        for (int jj = 0; jj < 8; ++jj) {
            const int k = 32 * s + 8 * g + jj, c = k & 1;
            const bool paired = pair_k1 && s == 1;       // (odd rows' K-step-1 pieces are not read in that layout: zeros)
            const int udp = paired ? 16 + 4 * (g & 1) + (jj >> 1) : (k >> 1);
            const int bb = paired ? b + (g >> 1) : b;
            const int al = xs * kx - (udp - (sa_w - 1)), a = sa_w * sa + al;
            double val = 0.0;
            if (al >= 0 && al < sa_w && a < ax && bb < ay && f >= 0 && f < n_foci && !(paired && (b & 1))) {
                const int e = cell[(size_t)a * ay + bb];
                if (e >= 0) {
                    const int es = perm[mirror * n + e];
                    const size_t off = (size_t)f * n + es;
                    const double cyc = freq * delays[off];
                    const double ph = 6.283185307179586476925286766559 * (cyc - floor(cyc));
                    const double w = apod[off] * area[es] * w_scale;
                    const double wr = w * cos(ph), wi = w * sin(ph);
                    val = o == 0 ? (c == 0 ? wr : -wi) : (c == 0 ? wi : wr);
                }
            }
            const _Float16 h = (_Float16)(float)val;
            hi.h[jj] = h;
            lo.h[jj] = (_Float16)(float)(val - (double)(float)h);
        }
        dst[s * 64 + lane] = hi.u;
        if (fp8corr) {      // e4m3 [hi(c0), hi(c1), lo(c0), lo(c1)] * (2^-6, 2^5) of this K-step's four offsets: 16 of the lane's 32 operand bytes
            Half8Bits q;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                int w = __builtin_amdgcn_cvt_pk_fp8_f32((float)hi.h[2 * e] * COS_F8_HI, (float)hi.h[2 * e + 1] * COS_F8_HI, 0, false);
                w = __builtin_amdgcn_cvt_pk_fp8_f32((float)lo.h[2 * e] * COS_F8_LO, (float)lo.h[2 * e + 1] * COS_F8_LO, w, true);
                q.w[e] = (unsigned)w;
            }
            lo.u = q.u;
        }
        dst[(2 + s) * 64 + lane] = lo.u;
    }
}

}  // namespace olx

using namespace olx;
OLX_BOUNDS_READER(toep)

#ifdef OLX_EXP_STAMPS
// developer build only (tools/stamps_toep.py): per-wave phase time stamps of this kernel
extern "C" int olx_exp_read_stamps_toep(unsigned long long* out) {
    return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(olx::g_stamps), sizeof(unsigned long long) * 4096 * 8);
}
#endif

void olx_pack_toep(olx_ctx* c) {
    const olx_ctx::Lattice& A = c->lat;
    dim3 g(c->toep_nsa * 8 * A.nsb, c->mp.n_tiles);
    hipLaunchKernelGGL(toep_pack_k, g, dim3(64), 0, c->stream, c->d_area, c->n_el, c->d_delays, c->d_apod, c->d_perm, c->freq,
                       c->mfma_wscale, c->plan_foci, c->d_colinfo, c->d_cell, A.ax, A.ay, 8 * A.nsb, c->fp8corr ? 1 : 0, c->toep_saw, c->cp.xs,
                       (c->fp8corr && c->toep_nm == 2) ? 1 : 0, c->d_afrag);
}

template <int MX, int MY>
static void launch_toep(olx_ctx* c, float* pm) {
    ToepParams T;
    T.q = c->cp; T.nsa = c->toep_nsa; T.sa_w = c->toep_saw; T.ax = c->lat.ax; T.ks_mask = c->toep_ksmask; T.ay_pad = 8 * c->lat.nsb;
    for (int q = 0; q < 4; ++q) T.targets[q] = c->toep_targets[q];
    const int nm = c->toep_nm;                          // row tiles per block (the planner's choice: olx.hip)
    T.n_rec = c->cp_nblocks;
    // (three row tiles: one block per CU walks the records; a multiple of 8 blocks so that a block's records stay on its XCD)
    const unsigned walkers = (unsigned)std::max(8, c->n_cu / 8 * 8);
    dim3 grid(nm == 3 ? std::min(c->cp_nblocks, walkers) : c->cp_nblocks, c->mp.n_tiles), blk((nm == 3 ? ToepShape<3>::WAVES : ToepShape<1>::WAVES) * 64);      // (the block records of this launch: all of them, or one side of a launch split at fp8_kcut)
    if (c->dir_lattice) {   // piston directivity folded into the geometry tables (one row tile: the planner keeps nm = 1 here)
        if (c->clamp || c->lat.clamp) hipLaunchKernelGGL((field_toep_k<MX, MY, true, true>), grid, blk, 0, c->stream, c->d_afrag, pm, c->d_inten, c->d_cpblocks, T);
        else hipLaunchKernelGGL((field_toep_k<MX, MY, false, true>), grid, blk, 0, c->stream, c->d_afrag, pm, c->d_inten, c->d_cpblocks, T);
        return;
    }
#define OLX_TP(CL, F8, NM_) hipLaunchKernelGGL((field_toep_k<MX, MY, CL, false, F8, NM_>), grid, blk, 0, c->stream, c->d_afrag, pm, c->d_inten, c->d_cpblocks, T)
#define OLX_TPN(CL, F8) do { if (nm == 3) OLX_TP(CL, F8, 3); else if (nm == 2) OLX_TP(CL, F8, 2); else OLX_TP(CL, F8, 1); } while (0)
    const bool cl = c->clamp || c->lat.clamp;
    if (c->fp8corr) { if (cl) OLX_TPN(true, true); else OLX_TPN(false, true); }      // e4m3 correction products (the planner's gated default)
    else            { if (cl) OLX_TPN(true, false); else OLX_TPN(false, false); }
#undef OLX_TPN
#undef OLX_TP
}

void olx_launch_toep(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) launch_toep<2, 2>(c, pm);
    else if (c->mx == 2) launch_toep<2, 1>(c, pm);
    else if (c->my == 2) launch_toep<1, 2>(c, pm);
    else launch_toep<1, 1>(c, pm);
}
