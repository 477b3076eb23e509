// kernel 2d (field_lattice_k): lattice arrays, block-Toeplitz geometry tables (serves complex output)
// gfx950 (CDNA4, wave64) only.  Data layout in HBM: DESIGN.md section 4; launchers declared in olx_launch.h.
#include "k_types.hip.h"
#include "olx_ctx.h"
#include "olx_launch.h"

namespace olx {

// ------------------------------------------------------------------------------------
// kernel 2d: lattice accumulate.  Matrix arrays (Transducer.gen_matrix_array, xdc/transducer.py:372-406)
// put their elements on a regular (a, b) lattice with pitch (px, py); when the pitch is a whole number of
// voxels (px = mx hx, py = my hy) the geometry term depends only on the INTEGER offset between voxel and
// element:      G(v, e) = g(i - mx a, j - my b, k),
// i.e. the contraction over elements is a dilated 2-D convolution and the A operand of kernel 2c is
// block-Toeplitz.  An MFMA tile's 16 voxel rows are therefore pitch-strided:
//     row = (sxp, sy, sz):  i = ibase + 2 mx sxp (sxp < 2),  j = jbase + my sy (sy < 4),  plane k0 + sz MT + t,
// and a wave owns MT such tiles (2 MT consecutive planes).  Against one 8 x 8 "super-block" of elements
// (4 K-steps of 4 x 4 elements) the rows of a plane see only 10 x 11 distinct offsets, so the wave evaluates
// 110 G values per plane (the transcendentals + the fp16 hi/lo split) instead of 512, writes them to a
// wave-private LDS table and reads its A fragments back from there: lane (row, k-group g) needs the four
// elements (aa = 0..3, bb = g) of K-step (ka, kb) = table row (sy - 4 kb - g + 7), entries p .. p+3 with
// p = 2 - 2 sxp + 4 ka (the table row is stored reversed).  The x stride of two pitches makes p EVEN, so a
// fragment is two aligned ds_read_b64 per part with immediate offsets -- half the LDS cycles of 4-byte reads
// and no address arithmetic.  B fragments, the hi/lo three-product scheme and the columns / store targets
// are kernel 2c's.  Exact in the same sense: only WHERE a (voxel, element) term is evaluated changes.
// Array edges are padded to whole super-blocks with zero-weight virtual elements on the same lattice.
// VALU work per K-step drops ~8x against kernel 2c (which is VALU-bound); the limiters become the matrix
// pipe and LDS bandwidth.
// ------------------------------------------------------------------------------------





template <int MT, int NT, int MX, int MY, bool CLAMP>
__global__ __launch_bounds__(LAT_THREADS, NT >= 4 ? 2 : 4) void field_lattice_k(
    const uint4* __restrict__ bfrag /*[tiles][ks][NT][2][64]*/, float* __restrict__ pmag, float* __restrict__ inten,
    float* __restrict__ cplx, const int* __restrict__ targets, const LatParams P) {
    constexpr int SB_PER_CHUNK = (LAT_ELEMS_LDS / NT) / 64 > 0 ? (LAT_ELEMS_LDS / NT) / 64 : 1;  // super-blocks of B per LDS stage
    constexpr int ZW = 2 * MT;                       // planes per wave
    constexpr int NW = LAT_THREADS / 64;             // waves per block
    constexpr int ZB = NW * ZW;                      // planes per block
    // one LDS arena: [B fragments | per-wave G tables] during the K loop, re-used as the output staging buffer
    constexpr int CS = ZB + 4;                       // staging column stride [floats] (+4: conflict-free 16-B writes)
    constexpr int RS = 16 * CS;                      // staging row stride (8 (x, y) rows)
    constexpr int B_BYTES = SB_PER_CHUNK * 4 * NT * 2 * 64 * 16, T_BYTES = NW * 2 * ZW * LAT_PSZ * 4;
    constexpr int OUT_BYTES = 8 * RS * 4;
    constexpr int ARENA = B_BYTES + T_BYTES > OUT_BYTES ? B_BYTES + T_BYTES : OUT_BYTES;
    __shared__ __attribute__((aligned(16))) unsigned char smem[ARENA];
    typedef uint4 (*BArr)[NT][2][64];
    BArr s_B = reinterpret_cast<BArr>(smem);
    unsigned* const s_T = reinterpret_cast<unsigned*>(smem + B_BYTES);
    float* const s_out = reinterpret_cast<float*>(smem);
    const int tile = blockIdx.y;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int g = lane >> 4, sxp = lane & 1, sy = (lane >> 1) & 3, sz = (lane >> 3) & 1;   // A-operand row = lane & 15
    // wave -> (row tile in x, row tile in y, group of ZW planes); the NW waves of a block take NW
    // consecutive plane groups of one row tile, so a block owns ZB consecutive floats per (voxel row, column).
    const unsigned kblocks = (unsigned)(P.kgroups + NW - 1) / NW;
    const unsigned kblock = blockIdx.x % kblocks, txy = blockIdx.x / kblocks;
    const int ty = (int)(txy % (unsigned)P.tiles_y), tx = (int)(txy / (unsigned)P.tiles_y);
    const int kgroup = (int)kblock * NW + wave;
    const bool active = kgroup < P.kgroups;
    const int k0 = kgroup * ZW;
    const int qx = tx / (2 * P.mx), qy = ty / P.my;
    const int ibase = P.x_lo + qx * 4 * P.mx + (tx - qx * 2 * P.mx);   // local voxel index of row sxp = 0 (sxp = 1: + 2 mx)
    const int jbase = P.y_lo + qy * 4 * P.my + (ty - qy * P.my);
    float dz2[ZW];
#pragma unroll
    for (int z = 0; z < ZW; ++z) {
        const float dz = (float)(k0 + z) * P.hz - P.flat_ez;
        dz2[z] = dz * dz;
    }
    // table-generation role of this lane: entries n = lane and lane + 64 of the 11 x 10 offset table of a plane
    int Ur[2], Wr[2], toff[2];
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const int n = lane + 64 * r;
        const int nn = n < 110 ? n : 109;
        const int wi = nn / 10, p = nn - 10 * wi;
        Ur[r] = ibase + P.x_begin + P.ux0 + P.mx * (2 - p);
        Wr[r] = jbase + P.uy0 + P.my * (wi - 7);
        toff[r] = n < 110 ? nn : LAT_PSZ - 1;          // spare lanes write the pad word
    }
    unsigned* const Thi = s_T + (wave * 2 + 0) * ZW * LAT_PSZ;
    unsigned* const Tlo = s_T + (wave * 2 + 1) * ZW * LAT_PSZ;
    // fragment read base of K-step (0, 0), tile 0: plane sz MT, row (sy - g + 7), entry 2 - 2 sxp  (even -> 8-B aligned)
    const int rbase = sz * MT * LAT_PSZ + (sy - g + 7) * LAT_TW + (2 - 2 * sxp);

    floatx4_t acc[MT][NT];
#pragma unroll
    for (int t = 0; t < MT; ++t)
#pragma unroll
        for (int nt = 0; nt < NT; ++nt) acc[t][nt] = floatx4_t{0.f, 0.f, 0.f, 0.f};

    const int nsbp = P.nsbp;                    // (= nsb: this kernel is never planned with the padded K-slot map)
    const int n_sb = P.nsa * nsbp;
    OLX_STAMP(0);
    // B fragments: chunk c+1 is fetched into registers while chunk c is contracted (the loads stay in flight across
    // the K-steps), then handed to LDS between two barriers -- no wave waits for global memory inside the loop.
    constexpr int CHUNK_U4 = SB_PER_CHUNK * 4 * NT * 128, PRE = CHUNK_U4 / LAT_THREADS;
    static_assert(CHUNK_U4 % LAT_THREADS == 0, "chunk must split evenly over the block");
    uint4 pre[PRE];
    const uint4* const bsrc = bfrag + (size_t)tile * n_sb * (4 * NT * 128);
#pragma unroll
    for (int q = 0; q < PRE; ++q) {
        const int idx = tid + q * LAT_THREADS;
        pre[q] = idx < n_sb * 4 * NT * 128 ? bsrc[idx] : make_uint4(0, 0, 0, 0);
    }
    for (int sb0 = 0; sb0 < n_sb; sb0 += SB_PER_CHUNK) {
        const int sb_here = min(SB_PER_CHUNK, n_sb - sb0);
        __syncthreads();
#pragma unroll
        for (int q = 0; q < PRE; ++q) reinterpret_cast<uint4*>(smem)[tid + q * LAT_THREADS] = pre[q];
        __syncthreads();
        {
            const int nxt = (sb0 + SB_PER_CHUNK) * 4 * NT * 128, lim = n_sb * 4 * NT * 128;
#pragma unroll
            for (int q = 0; q < PRE; ++q) {
                const int idx = nxt + tid + q * LAT_THREADS;
                if (idx < lim) pre[q] = bsrc[idx];
            }
        }
        if (!active) continue;
        if (sb0 == 0) OLX_STAMP(1);
        for (int sbl = 0; sbl < sb_here; ++sbl) {
            const int sb = sb0 + sbl;
            const int sa = sb / nsbp, sbb = sb - sa * nsbp;      // sa-major order (host slot map)
            // ---- G table of this super-block: 110 offsets x ZW planes (2 rounds of 64 lanes per plane)
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const float U = (float)(Ur[r] - 8 * P.mx * sa), W = (float)(Wr[r] - 8 * P.my * sbb);
                const float dx = fmaf(U, P.hx_hi, fmaf(U, P.hx_lo, P.fx0));
                const float dy = fmaf(W, P.hy_hi, fmaf(W, P.hy_lo, P.fy0));
                const float r2 = fmaf(dy, dy, dx * dx);
#pragma unroll
                for (int z = 0; z < ZW; ++z) {
                    float d2 = r2 + dz2[z];
                    if (CLAMP) d2 = fmaxf(d2, P.dmin2);
                    const float ri = __builtin_amdgcn_rsqf(d2);
                    const float ph = d2 * ri;            // distance in wavelengths = phase in revolutions
                    const float rs = ri * P.g_scale;
                    const float gr = rs * __builtin_amdgcn_cosf(ph);
                    const float gi = rs * __builtin_amdgcn_sinf(ph);
                    const auto hi = __builtin_amdgcn_cvt_pkrtz(gr, gi);
                    // lo = g - (float)hi in ONE mixed-precision fma per component (the compiler's form: a convert and a subtract -- same bits)
                    float lr, li;
                    const unsigned hw = __builtin_bit_cast(unsigned, hi);
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(lr) : "v"(hw), "v"(gr));
                    asm("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(li) : "v"(hw), "v"(gi));
                    const auto lo = __builtin_amdgcn_cvt_pkrtz(lr, li);
                    Thi[z * LAT_PSZ + toff[r]] = __builtin_bit_cast(unsigned, hi);
                    Tlo[z * LAT_PSZ + toff[r]] = __builtin_bit_cast(unsigned, lo);
                }
            }
            // the table is wave-private: DS operations of one wave execute in order, only the compiler must not
            // move the fragment reads above the table writes
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (sb == 0) OLX_STAMP(2);
#pragma unroll
            for (int ks = 0; ks < 4; ++ks) {
                const int ka = ks & 1, kb = ks >> 1;
                Half8Bits bh[NT], bl[NT];
#pragma unroll
                for (int nt = 0; nt < NT; ++nt) {
                    bh[nt].u = s_B[sbl * 4 + ks][nt][0][lane];
                    bl[nt].u = s_B[sbl * 4 + ks][nt][1][lane];
                }
                const int roff = rbase - 4 * kb * LAT_TW + 4 * ka;
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    Half8Bits ah, al;
                    // four separate ds_read_b64 (2 LDS cycles each, 64-bank mode).  Relaxed atomic loads keep the
                    // compiler from fusing them into ds_read2_b64, which runs at a quarter of that rate.
                    const unsigned long long* ph2 = reinterpret_cast<const unsigned long long*>(Thi + t * LAT_PSZ + roff);
                    const unsigned long long* pl2 = reinterpret_cast<const unsigned long long*>(Tlo + t * LAT_PSZ + roff);
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
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            if (sb == 0) OLX_STAMP(3);
        }
    }
    OLX_STAMP(4);
    // epilogue.  D layout: lane holds rows 4 (lane >> 4) + r of column lane & 15 = (o, part); even lanes own Re, odd
    // lanes Im of output o, one DPP swap gives |p|^2 to both; the even lane keeps |p|, the odd lane the intensity.
    // A lane's MT planes are only 4 MT contiguous bytes and its neighbours belong to other columns / rows, so the
    // values are transposed through LDS: the block's 4 waves hold ZB consecutive planes of the same 8 (x, y) rows x
    // 16 columns, and every (row, column, store target) leaves as one contiguous 4 ZB-byte run written by ZB / 4
    // adjacent lanes (full 128-B lines at ZB = 32).
    const int c16 = lane & 15, part = c16 & 1, gy = lane >> 4;
    const int kb0 = (int)kblock * ZB;                // first plane of the block
    const bool fast = kb0 + ZB <= P.nz;              // every plane of the block exists (dword-aligned 16-byte stores: rows of odd length are fine)
#pragma unroll
    for (int nt = 0; nt < NT; ++nt) {
        __syncthreads();                             // K loop / previous read-out done with the arena
        if (active) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * gy + r;          // (sxp, sy, sz) = (row & 1, (row >> 1) & 3, row >> 3)
                float w[MT];
#pragma unroll
                for (int t = 0; t < MT; ++t) {
                    const float v = acc[t][nt][r] * P.out_scale;
                    const float sq = v * v;
                    const float m2 = sq + quad_swap1(sq);          // re^2 + im^2 (partner lane holds the other part)
                    w[t] = part == 0 ? __builtin_amdgcn_sqrtf(m2) : m2 * P.inten_scale;
                }
                float* dst = s_out + (row & 7) * RS + c16 * CS + wave * ZW + (row >> 3) * MT;
                if (MT % 4 == 0) {
#pragma unroll
                    for (int t4 = 0; t4 < MT / 4; ++t4)
                        *reinterpret_cast<float4*>(dst + 4 * t4) = make_float4(w[4 * t4], w[4 * t4 + 1], w[4 * t4 + 2], w[4 * t4 + 3]);
                } else {
#pragma unroll
                    for (int t = 0; t < MT; ++t) dst[t] = w[t];
                }
            }
        }
        __syncthreads();
        if (nt == 0) OLX_STAMP(5);
        // read-out: thread -> (column, (x, y) rows r0, r0 + RSTEP, ..., piece of 4 planes); ZB / 4 adjacent lanes write
        // one contiguous run.  Column, piece and the column's store targets are fixed per thread, so the
        // (focus, mirror image) bases are formed once and each store needs one row offset.
        constexpr int PIECES = ZB / 4, RSTEP = LAT_THREADS / (PIECES * 16);
        static_assert(LAT_THREADS % (PIECES * 16) == 0 && 8 % RSTEP == 0, "read-out map");
        {
            const int piece = tid % PIECES, col = (tid / PIECES) & 15, r0 = tid / (PIECES * 16);
            const int kz = kb0 + 4 * piece;
            const bool is_p = (col & 1) == 0;
            float* const arr = is_p ? pmag : inten;
            const bool want = (is_p ? (P.flags & 1u) : (P.flags & 2u)) != 0 && kz < P.nz;
            const int4 tg = reinterpret_cast<const int4*>(targets)[(size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt * MFMA_COLS + (col >> 1)];
            const int tgs[4] = {tg.x, tg.y, tg.z, tg.w};
            const float* src = s_out + col * CS + 4 * piece;
            if (want && fast) {
                float* tb[4]; bool tfx[4], tfy[4];   // per store target: volume base, mirror flags (hoisted out of the rows)
#pragma unroll
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int code = tgs[s4], m = code & 3;
                    tb[s4] = code < 0 ? nullptr : arr + (long long)(code >> 2) * P.vox + kz;
                    tfx[s4] = (MX == 2) && (m & 1);
                    tfy[s4] = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                }
#pragma unroll
                for (int it = 0; it < 8 / RSTEP; ++it) {
                    const int row = r0 + RSTEP * it;
                    const int i = ibase + (row & 1) * 2 * P.mx, j = jbase + (row >> 1) * P.my;
                    if (i >= P.nx || j >= P.ny) continue;
                    const float4 val = *reinterpret_cast<const float4*>(src + row * RS);
                    const int ai = i * P.ny, aX = (P.nx - 1 - i) * P.ny, bY = P.ny - 1 - j;
#pragma unroll
                    for (int s4 = 0; s4 < 4; ++s4) {
                        if (!tb[s4]) continue;
                        const int off = ((tfx[s4] ? aX : ai) + (tfy[s4] ? bY : j)) * P.nz;
                        *reinterpret_cast<floatx4u_t*>(tb[s4] + off) = floatx4u_t{val.x, val.y, val.z, val.w};
                    }
                }
            } else if (want) {                       // ragged nz: guarded scalar stores (not a throughput path)
#pragma unroll 1
                for (int row = r0; row < 8; row += RSTEP) {
                    const int i = ibase + (row & 1) * 2 * P.mx, j = jbase + (row >> 1) * P.my;
                    if (i >= P.nx || j >= P.ny) continue;
#pragma unroll 1
                    for (int s4 = 0; s4 < 4; ++s4) {
                        const int code = tgs[s4];
                        if (code < 0) continue;
                        const int m = code & 3;
                        const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                        const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
                        float* o = arr + (long long)(code >> 2) * P.vox + ((long long)io * P.ny + jo) * P.nz + kz;
#pragma unroll 1
                        for (int q = 0; q < 4; ++q) if (kz + q < P.nz) o[q] = src[row * RS + q];
                    }
                }
            }
        }
        if (nt == NT - 1) OLX_STAMP(6);
        if ((P.flags & 4u) && active) {              // complex output (not a throughput path): direct scalar stores
            const int4 tg = reinterpret_cast<const int4*>(targets)[(size_t)tile * (MFMA_COLS * MFMA_MAX_NT) + nt * MFMA_COLS + (c16 >> 1)];
            const int tgs[4] = {tg.x, tg.y, tg.z, tg.w};
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int row = 4 * gy + r;
                const int i = ibase + (row & 1) * 2 * P.mx, j = jbase + ((row >> 1) & 3) * P.my, kz0 = k0 + (row >> 3) * MT;
                if (i >= P.nx || j >= P.ny) continue;
                for (int s4 = 0; s4 < 4; ++s4) {
                    const int code = tgs[s4];
                    if (code < 0) continue;
                    const int f = code >> 2, m = code & 3;
                    const bool fx = (MX == 2) && (m & 1), fy = (MY == 2) && ((MX == 2) ? (m >> 1) : (m & 1));
                    const int io = fx ? (P.nx - 1 - i) : i, jo = fy ? (P.ny - 1 - j) : j;
                    const long long base = (long long)f * P.vox + ((long long)io * P.ny + jo) * P.nz + kz0;
#pragma unroll
                    for (int t = 0; t < MT; ++t)
                        if (kz0 + t < P.nz) cplx[2 * (base + t) + part] = acc[t][nt][r] * P.out_scale;
                }
            }
        }
    }
}


}  // namespace olx

using namespace olx;

template <int MT, int NT, int MX, int MY>
static void launch_lattice(olx_ctx* c, float* pm, bool clamp) {
    const LatParams& L = c->lp;
    constexpr int NW = LAT_THREADS / 64;
    const long long blocks = (long long)L.tiles_x * L.tiles_y * ((L.kgroups + NW - 1) / NW);
    dim3 grid((unsigned)blocks, c->mp.n_tiles), blk(LAT_THREADS);
    if (clamp) hipLaunchKernelGGL((field_lattice_k<MT, NT, MX, MY, true>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_cplx, c->d_targets, L);
    else       hipLaunchKernelGGL((field_lattice_k<MT, NT, MX, MY, false>), grid, blk, 0, c->stream, c->d_bfrag, pm, c->d_inten, c->d_cplx, c->d_targets, L);
}

template <int MX, int MY>
static void dispatch_lattice_nt(olx_ctx* c, float* pm) {
    const bool clamp = c->clamp || c->lat.clamp;
    if (c->nt == 1) launch_lattice<4, 1, MX, MY>(c, pm, clamp);
    else if (c->nt == 2) launch_lattice<4, 2, MX, MY>(c, pm, clamp);
    else launch_lattice<4, 4, MX, MY>(c, pm, clamp);
}

void olx_launch_lattice(olx_ctx* c, float* pm) {
    if (c->mx == 2 && c->my == 2) dispatch_lattice_nt<2, 2>(c, pm);
    else if (c->mx == 2) dispatch_lattice_nt<2, 1>(c, pm);
    else if (c->my == 2) dispatch_lattice_nt<1, 2>(c, pm);
    else dispatch_lattice_nt<1, 1>(c, pm);
}
