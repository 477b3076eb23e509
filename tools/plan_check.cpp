// CPU-side checker of the lattice kernels' host planning (openlifu-python_amd/csrc/olx_plan.cpp), built by tests/test_plan_host.py with
//   g++ -std=c++17 -O1 -g -fsanitize=address,undefined -fno-sanitize-recover=all tools/plan_check.cpp openlifu-python_amd/csrc/olx_plan.cpp
// and run over seeded fuzz shapes: arrays that pad to super-blocks, pitches of 1 .. 12 voxels, grids that cut cosets into unequal
// parts, ragged plane counts, folded / unfolded axes, x-slabs, 1 .. 64 foci with shared and distinct steering vectors.  Invariants:
//   lattice      every element sits in exactly one K slot, virtual slots are -1, the slot map has nsa x nsbp x 64 entries
//   columns      every (focus, image) is a store target of exactly one column, all targets of a column carry the same steering vector,
//                tiles hold at most maxc columns, at most 4 (2 after balancing, where slots were free) targets per column
//   blocks       the positions of all records x their plane blocks cover every voxel of the computed region exactly once, lie inside it,
//                respect the per-part limit, and the kernels' magic division pos / KY is exact for every position
//   store jobs   the dense job lists name exactly the targets of their columns
//   foci         geometric delays are recognised and reproduce the foci; scrambled delays are refused
// Exit code 0 = all shapes passed; any violation prints the shape and exits 1 (sanitizer reports abort on their own).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <random>
#include <set>
#include <string>
#include <vector>

#include "../openlifu-python_amd/csrc/olx_plan.h"

using namespace olx;
using namespace olxplan;

static int g_fail = 0;
static long long g_lattices = 0, g_records = 0, g_columns = 0;
#define CHECK(cond, ...) do { if (!(cond)) { ++g_fail; fprintf(stderr, "FAIL %s:%d (%s): ", __FILE__, __LINE__, #cond); fprintf(stderr, __VA_ARGS__); fputc('\n', stderr); if (g_fail > 20) exit(1); } } while (0)

// the checker's own notion of "the same steering vector" (independent of the planner's): equal drive weights and equal phases (mod one period)
static bool ref_same_vector(const Steering& S, int f1, int m1, int f2, int m2) {
    for (int e = 0; e < S.n; ++e) {
        const int e1 = S.perm[(size_t)m1 * S.n + e], e2 = S.perm[(size_t)m2 * S.n + e];
        const double w1 = S.apod[(size_t)f1 * S.n + e1] * S.area[e1], w2 = S.apod[(size_t)f2 * S.n + e2] * S.area[e2];
        if (std::fabs(w1 - w2) > 1e-9 * std::max(std::fabs(w1), std::fabs(w2))) return false;
        if (w1 == 0.0) continue;
        const double turns = (S.delays[(size_t)f1 * S.n + e1] - S.delays[(size_t)f2 * S.n + e2]) * S.freq;
        if (std::fabs(turns - std::round(turns)) > 1e-6) return false;
    }
    return true;
}

struct Shape {
    int nax, nay, mxv, myv;        // elements per axis, pitch in voxels
    int n[3];                      // grid
    int x_begin, x_count;          // slab
    bool fold_x, fold_y;           // mirror folds (grid centred on the array)
    int nf;                        // foci
    double h;                      // spacing [m]
};

static void check_shape(const Shape& S, std::mt19937_64& rng, int nt_force) {
    const int n = S.nax * S.nay;
    // ---- the array: lattice points in (a, b) order shuffled (element order must not matter), z = 0
    std::vector<int> order(n);
    for (int e = 0; e < n; ++e) order[e] = e;
    std::shuffle(order.begin(), order.end(), rng);
    std::vector<double> pos(3 * (size_t)n), area(n, 1e-6);
    const double px = S.mxv * S.h, py = S.myv * S.h;
    for (int q = 0; q < n; ++q) {
        const int a = order[q] / S.nay, b = order[q] % S.nay;
        pos[q] = (a - 0.5 * (S.nax - 1)) * px; pos[(size_t)n + q] = (b - 0.5 * (S.nay - 1)) * py; pos[2 * (size_t)n + q] = 0.0;
    }
    const double spacing[3] = {S.h, S.h, S.h};
    // grid: centred on the array where folded, shifted by a whole number of voxels plus a fraction otherwise
    double origin[3] = {-(S.n[0] - 1) * 0.5 * S.h + (S.fold_x ? 0.0 : 2.25 * S.h), -(S.n[1] - 1) * 0.5 * S.h + (S.fold_y ? 0.0 : -1.5 * S.h), 4e-3};
    double lo[3], hi[3];
    for (int a = 0; a < 3; ++a) {
        const int b0 = a == 0 ? S.x_begin : 0, cnt = a == 0 ? S.x_count : S.n[a];
        lo[a] = origin[a] + b0 * spacing[a]; hi[a] = origin[a] + (b0 + cnt - 1) * spacing[a];
    }
    Lattice L;
    detect_lattice(L, true, n, pos.data(), spacing, lo, hi, 0.5 * S.h);
    if ((long long)((S.nax + 7) / 8) * ((S.nay + 7) / 8) * 64 > 2LL * n) { CHECK(!L.ok, "padding rule"); return; }
    CHECK(L.ok && L.ax == S.nax && L.ay == S.nay && L.mx == S.mxv && L.my == S.myv, "lattice %dx%d pitch %dx%d not recognised", S.nax, S.nay, S.mxv, S.myv);
    if (!L.ok) return;
    ++g_lattices;
    for (int nsbp : {L.nsb, (L.nsb + 1) & ~1}) {
        build_slot_map(L, nsbp);
        CHECK((int)L.slot_elem.size() == L.nsa * nsbp * 64 && L.n_pad == L.nsa * nsbp * 64, "slot map size");
        std::vector<int> seen(n, 0);
        for (int v : L.slot_elem) { CHECK(v >= -1 && v < n, "slot value %d", v); if (v >= 0) seen[v]++; }
        for (int e = 0; e < n; ++e) CHECK(seen[e] == 1, "element %d in %d slots", e, seen[e]);
    }
    // ---- steering: foci with geometric delays; on folded axes some foci are mirror partners / on the axis (shared columns)
    const int F = S.nf;
    const bool whole_x = S.x_begin == 0 && S.x_count == S.n[0];
    const int mxf = (S.fold_x && whole_x) ? 2 : 1, myf = S.fold_y ? 2 : 1, n_img = mxf * myf;
    std::uniform_real_distribution<double> U(-4e-3, 4e-3), Z(20e-3, 40e-3);
    std::vector<double> foci(3 * (size_t)F);
    for (int f = 0; f < F; ++f) {
        double x = U(rng), y = U(rng), z = Z(rng);
        if (f % 5 == 0) { x = 0; y = 0; }                 // on the axis: all images share one vector
        else if (f % 5 == 1) x = 0;                       // on a symmetry plane
        else if (f % 5 == 2 && f >= 3) { x = -foci[3 * (size_t)(f - 1)]; y = foci[3 * (size_t)(f - 1) + 1]; z = foci[3 * (size_t)(f - 1) + 2]; }   // mirror partner of the previous focus
        foci[3 * (size_t)f] = x; foci[3 * (size_t)f + 1] = y; foci[3 * (size_t)f + 2] = z;
    }
    const double c0 = 1500.0, freq = 400e3;
    std::vector<double> delays((size_t)F * n), apod((size_t)F * n, 1.0);
    for (int f = 0; f < F; ++f) {
        double tmax = 0;
        std::vector<double> tof(n);
        for (int e = 0; e < n; ++e) {
            const double dx = foci[3 * (size_t)f] - pos[e], dy = foci[3 * (size_t)f + 1] - pos[(size_t)n + e], dz = foci[3 * (size_t)f + 2];
            tof[e] = std::sqrt(dx * dx + dy * dy + dz * dz) / c0; tmax = std::max(tmax, tof[e]);
        }
        for (int e = 0; e < n; ++e) delays[(size_t)f * n + e] = tmax - tof[e];
    }
    // mirror permutations of the element set (exact: lattice symmetric about 0)
    auto mirror = [&](int axis) {
        std::vector<int> p(n, -1);
        for (int e = 0; e < n; ++e)
            for (int o = 0; o < n; ++o)
                if (std::fabs(pos[(size_t)axis * n + o] + pos[(size_t)axis * n + e]) < 1e-12 && std::fabs(pos[(size_t)(1 - axis) * n + o] - pos[(size_t)(1 - axis) * n + e]) < 1e-12) p[e] = o;
        return p;
    };
    const std::vector<int> hpx = mirror(0), hpy = mirror(1);
    std::vector<int> perm((size_t)4 * n);
    for (int m = 0; m < 4; ++m)
        for (int e = 0; e < n; ++e) {
            int o = e;
            const bool fx = mxf == 2 && (m & 1), fy = myf == 2 && (mxf == 2 ? (m >> 1) : (m & 1));
            if (m < n_img && fx) o = hpx[o];
            if (m < n_img && fy) o = hpy[o];
            perm[(size_t)m * n + e] = o;
        }
    Steering SV; SV.n = n; SV.F = F; SV.n_img = n_img; SV.perm = perm.data(); SV.delays = delays.data(); SV.apod = apod.data(); SV.area = area.data(); SV.freq = freq;
    for (int maxc : {32, 16, 8}) {
        Tiles tiles = pack_columns(SV, maxc);
        std::map<int, int> hits;
        for (auto& t : tiles) {
            CHECK((int)t.size() <= maxc && !t.empty(), "tile of %zu columns (max %d)", t.size(), maxc);
            for (auto& col : t) {
                ++g_columns;
                CHECK(col.ntgt >= 1 && col.ntgt <= 4, "column with %d targets", col.ntgt);
                for (int q = 0; q < 4; ++q) {
                    CHECK((q < col.ntgt) == (col.tgt[q] >= 0), "target list not dense");
                    if (col.tgt[q] < 0) continue;
                    hits[col.tgt[q]]++;
                    CHECK(ref_same_vector(SV, col.f, col.m, col.tgt[q] >> 2, col.tgt[q] & 3), "target %d stored by a column with another steering vector", col.tgt[q]);
                }
            }
        }
        for (int f = 0; f < F; ++f) for (int m = 0; m < n_img; ++m) CHECK(hits[f * 4 + m] == 1, "(focus %d, image %d) stored %d times (maxc %d)", f, m, hits[f * 4 + m], maxc);
        CHECK((int)hits.size() == F * n_img, "spurious targets");
        // sharing really happens: an on-axis focus on a doubly folded grid needs ONE column
        if (n_img == 4 && F >= 1) { int cnt = 0; for (auto& t : tiles) for (auto& col : t) if (col.f == 0) ++cnt; CHECK(cnt == 1, "on-axis focus in %d columns", cnt); }
        Tiles bal = tiles;
        balance_store_targets(bal, maxc);
        std::map<int, int> hits2;
        for (size_t t = 0; t < bal.size(); ++t) {
            CHECK((int)bal[t].size() <= maxc, "balanced tile overflows");
            for (auto& col : bal[t]) { for (int q = 0; q < col.ntgt; ++q) hits2[col.tgt[q]]++; if ((int)bal[t].size() < maxc) CHECK(col.ntgt <= 2, "unbalanced column although a slot is free"); }
        }
        CHECK(hits2 == hits, "balancing changed the target set");
        // store jobs of kernel 2e
        const std::vector<int> jobs = build_store_jobs(tiles, MFMA_MAX_NT, MFMA_COLS, COS_JOBS, true, true);
        for (size_t t = 0; t < tiles.size(); ++t)
            for (int nt = 0; nt < MFMA_MAX_NT; ++nt) {
                const int* jb = &jobs[(t * MFMA_MAX_NT + nt) * (COS_JOBS + 1)];
                std::multiset<int> want, got;
                for (int c16 = 0; c16 < 16; ++c16) {
                    const size_t o = (size_t)nt * MFMA_COLS + (c16 >> 1);
                    if (o < tiles[t].size()) for (int q = 0; q < tiles[t][o].ntgt; ++q) want.insert(c16 | ((tiles[t][o].tgt[q] & 3) << 4) | ((tiles[t][o].tgt[q] >> 2) << 6));
                }
                int cnt = 0;
                while (cnt < COS_JOBS && jb[cnt] >= 0) got.insert(jb[cnt++]);
                CHECK(got == want, "store jobs of tile %zu / %d", t, nt);
                CHECK((1 << jb[COS_JOBS]) >= cnt && (cnt <= 1 || (1 << (jb[COS_JOBS] - 1)) < cnt), "job count log2");
            }
    }
    // focus inference (flat array, geometric delays) ... and its refusal of anything else
    {
        std::vector<double> got;
        CHECK(infer_foci(true, n, F, pos.data(), delays.data(), c0, origin[2] + 0.5 * (S.n[2] - 1) * S.h, got), "geometric delays not recognised");
        for (size_t q = 0; q < got.size() && q < foci.size(); ++q) CHECK(std::fabs(got[q] - foci[q]) < 1e-6, "inferred focus off by %g", got[q] - foci[q]);
        std::vector<double> bad = delays;
        for (int e = 0; e < n; e += 3) bad[e] += 1e-7 * (1 + e % 5);
        CHECK(!infer_foci(true, n, F, pos.data(), bad.data(), c0, origin[2], got), "scrambled delays accepted");
    }
    // ---- block records of every shape the kernels use: (kxw, zb, limit) = 2g / 2e NT = 2 (3, 16, 40), 2e NT = 1 (6, 16, 0), 2e NT = 4 (2, 16, 0), 2f (8 | 16 | 24, 16, 0; positions one pitch apart),
    // each in every order of the records over the XCDs the host may choose (grp plane blocks -- or plane blocks x y cosets -- in a row on one XCD)
    struct Form { const char* name; int kxw, zb; unsigned grp; int max_pos; int xs; };
    std::vector<Form> forms;
    for (unsigned grp : {1u, 2u, 4u, 16u, 48u, 192u})
        for (const Form& f0 : {Form{"2g", 3, 16, 2, 40, 2}, Form{"2e nt1", 6, 16, 2, 0, 2}, Form{"2e nt4", 2, 16, 2, 0, 2}, Form{"2f", 8, 16, 2, 0, 1}, Form{"2f m2", 16, 16, 2, 0, 1}, Form{"2f m3", 24, 16, 2, 0, 1}}) {
            Form f = f0; f.grp = grp; forms.push_back(f);
        }
    for (const Form& fm : forms) {
        if (nt_force && fm.kxw != nt_force) continue;
        CosetParams Q{};
        Q.nx = S.x_count; Q.ny = S.n[1]; Q.nz = S.n[2]; Q.x_begin = S.x_begin;
        Q.x_lo = mxf == 2 ? Q.nx / 2 : 0; Q.y_lo = myf == 2 ? Q.ny / 2 : 0;
        Q.mx = L.mx; Q.my = L.my; Q.nsa = L.nsa; Q.nsb = L.nsb; Q.nsbp = (L.nsb + 1) & ~1; Q.xs = fm.xs;
        Q.ux0 = (int)std::llround((origin[0] - L.x0) / S.h); Q.uy0 = (int)std::llround((origin[1] - L.y0) / S.h);
        coset_partition(Q, fm.kxw, fm.zb);
        std::vector<CosetBlock> blk;
        std::string why;
        {
            const bool ok = build_coset_blocks(Q, fm.zb, fm.grp, fm.max_pos, blk, why);
            CHECK(ok, "%s: %s", fm.name, why.c_str());
            if (!ok) continue;
            CHECK(blk.size() == (size_t)(Q.xs * Q.mx * Q.my * Q.nsx * Q.nsy * Q.kblocks), "record count");
            const int wx = Q.nx - Q.x_lo, wy = Q.ny - Q.y_lo;
            std::vector<unsigned char> cover((size_t)wx * wy * Q.kblocks, 0);
            for (const CosetBlock& B : blk) {
                if (B.npos <= 0) continue;
                ++g_records;
                CHECK(B.k0 % fm.zb == 0 && B.k0 >= 0 && B.k0 < Q.kblocks * fm.zb, "plane block %d", B.k0);
                CHECK(B.KX >= 1 && B.KX <= fm.kxw && B.KY >= 1 && B.KY <= COS_KYW && B.npos == B.KX * B.KY, "part %d x %d (npos %d)", B.KX, B.KY, B.npos);
                for (int pq = 0; pq < B.npos; ++pq) CHECK(((pq * B.ky_magic) >> 16) == pq / B.KY, "magic division %d / %d", pq, B.KY);
                for (int kx = 0; kx < B.KX; ++kx)
                    for (int ky = 0; ky < B.KY; ++ky) {
                        const int i = B.ibase + Q.xs * Q.mx * kx, j = B.jbase + Q.my * ky;
                        CHECK(i >= Q.x_lo && i < Q.nx && j >= Q.y_lo && j < Q.ny, "position (%d, %d) outside the computed region", i, j);
                        if (i >= Q.x_lo && i < Q.nx && j >= Q.y_lo && j < Q.ny) cover[((size_t)(i - Q.x_lo) * wy + (j - Q.y_lo)) * Q.kblocks + B.k0 / fm.zb]++;
                    }
            }
            size_t bad = 0;
            for (unsigned char v : cover) bad += v != 1;
            CHECK(bad == 0, "%s: %zu (voxel column, plane block) cells not covered exactly once (grid %dx%dx%d pitch %dx%d fold %d%d slab %d+%d)", fm.name, bad, S.n[0], S.n[1], S.n[2],
                  S.mxv, S.myv, mxf, myf, S.x_begin, S.x_count);
            // the records that share z lines (plane blocks j, j + 1, ... of one part) sit 8 ids apart = one XCD (when the id space allows it)
            if (fm.grp >= 2 && fm.grp <= (unsigned)Q.kblocks && (Q.kblocks % fm.grp) == 0 && blk.size() % (8 * fm.grp) == 0)
                for (size_t id = 0; id + 8 < blk.size(); ++id)
                    if ((id / 8) % fm.grp != fm.grp - 1 && blk[id].npos > 0)
                        CHECK(blk[id + 8].ibase == blk[id].ibase && blk[id + 8].jbase == blk[id].jbase && blk[id + 8].k0 == blk[id].k0 + fm.zb, "line partners not 8 ids apart");
        }
    }
}

// ---- e4m3 error rule (olx_plan.h FP8_ERR_K / FP8_ERR_BOUND, nearfield_s2): known decisions on BASELINE's 16 x 16 @ 3 mm array, uniform drive, focus (0, 0, 40) mm
// (ratios from tools/emul_fp8_bound.py), and nearfield_s2 against the brute-force maximum over every voxel of the grid's first planes.
static void check_fp8_rule() {
    const int na = 16, n = na * na;
    std::vector<double> pos(3 * (size_t)n, 0.0);
    for (int a = 0; a < na; ++a) for (int b = 0; b < na; ++b) { pos[(size_t)a * na + b] = (a - 7.5) * 3e-3; pos[(size_t)n + a * na + b] = (b - 7.5) * 3e-3; }
    double peak = 0;
    for (int e = 0; e < n; ++e) peak += 1.0 / std::sqrt(pos[e] * pos[e] + pos[(size_t)n + e] * pos[(size_t)n + e] + 40e-3 * 40e-3);
    struct G { double h, z0; int nxy, nz; bool admit; double ratio; } grids[] = {
        {0.25e-3, 5e-3, 256, 256, true, 0.188},       // the headline grid
        {0.5e-3, 5e-3, 128, 128, true, 0.188},        // configs[1]
        {1e-3, -4e-3, 61, 65, false, 0.340},          // the reference's default SimSetup: through the element plane
        {0.5e-3, -4e-3, 121, 129, false, 0.730},      // (odd counts: a voxel sits ON every element -- the clamp distance)
        {0.25e-3, -4e-3, 241, 257, false, 1.403},
        {0.25e-3, 0.25e-3, 256, 256, false, 0.61},    // one voxel above the element plane
        {0.25e-3, 1e-3, 256, 256, false, 0.285},
    };
    for (const G& g : grids) {
        const double origin[3] = {-(g.nxy - 1) / 2.0 * g.h, -(g.nxy - 1) / 2.0 * g.h, g.z0}, spacing[3] = {g.h, g.h, g.h};
        const int b0[3] = {0, 0, 0}, cnt[3] = {g.nxy, g.nxy, g.nz};
        const double s2 = nearfield_s2(n, pos.data(), origin, spacing, b0, cnt, 0.5 * g.h);
        const double ratio = std::sqrt(s2) / peak;
        CHECK(std::fabs(ratio - g.ratio) <= 0.03 * g.ratio + 0.005, "fp8 rule: ratio %.4f, expected %.3f (h %.2g z0 %.2g)", ratio, g.ratio, g.h, g.z0);
        CHECK((FP8_ERR_K * ratio <= FP8_ERR_BOUND) == g.admit, "fp8 rule: h %.2g z0 %.2g admitted = %d, expected %d", g.h, g.z0, (int)(FP8_ERR_K * ratio <= FP8_ERR_BOUND), (int)g.admit);
        // brute force over the three planes nearest to the elements (a quadrant: the array and the grid are symmetric)
        double brute = 0;
        int kn = (int)std::llround((0.0 - g.z0) / g.h); kn = std::max(0, std::min(kn, g.nz - 1));
        for (int k = std::max(0, kn - 1); k <= std::min(g.nz - 1, kn + 1); ++k)
            for (int i = g.nxy / 2; i < g.nxy; ++i)
                for (int j = g.nxy / 2; j < g.nxy; ++j) {
                    const double x = origin[0] + i * g.h, y = origin[1] + j * g.h, z = origin[2] + k * g.h;
                    double sum = 0;
                    for (int e = 0; e < n; ++e) {
                        const double dx = x - pos[e], dy = y - pos[(size_t)n + e];
                        sum += 1.0 / std::max(dx * dx + dy * dy + z * z, 0.25 * g.h * g.h);
                    }
                    brute = std::max(brute, sum);
                }
        CHECK(s2 <= brute * (1 + 1e-12) && s2 >= 0.97 * brute, "nearfield_s2 %.6g vs brute-force maximum %.6g (h %.2g z0 %.2g)", s2, brute, g.h, g.z0);
    }
    {   // a slab beside the array sees the clamped nearest voxel, not the element's own position
        const double origin[3] = {-31.875e-3, -31.875e-3, 5e-3}, spacing[3] = {0.25e-3, 0.25e-3, 0.25e-3};
        const int b0[3] = {192, 0, 0}, cnt[3] = {64, 256, 256}, whole0[3] = {0, 0, 0}, whole[3] = {256, 256, 256};
        CHECK(nearfield_s2(n, pos.data(), origin, spacing, b0, cnt, 0.125e-3) < nearfield_s2(n, pos.data(), origin, spacing, whole0, whole, 0.125e-3), "slab beside the centre must see a smaller sum");
    }
}

int main(int argc, char** argv) {
    check_fp8_rule();
    const int cases = argc > 1 ? atoi(argv[1]) : 200;
    const unsigned long long seed = argc > 2 ? strtoull(argv[2], nullptr, 10) : 147;
    std::mt19937_64 rng(seed);
    auto ri = [&](int lo, int hi) { return (int)(lo + rng() % (unsigned long long)(hi - lo + 1)); };
    int done = 0;
    // BASELINE's shapes first: 16 x 16 @ 12 voxels on 256^3 (8-focus shard, 64-focus sweep, 4-GPU x-slab), 32 x 32 @ 12 on 512^3 (coarse: plane count cut)
    const Shape fixed[] = {{16, 16, 12, 12, {256, 256, 256}, 0, 256, true, true, 8, 0.25e-3}, {16, 16, 12, 12, {256, 256, 256}, 0, 256, true, true, 64, 0.25e-3},
                           {16, 16, 12, 12, {256, 256, 256}, 64, 64, true, true, 8, 0.25e-3}, {32, 32, 12, 12, {512, 512, 48}, 0, 512, true, true, 1, 0.125e-3},
                           {16, 16, 6, 6, {128, 128, 128}, 0, 128, true, true, 1, 0.5e-3}, {8, 8, 4, 4, {61, 61, 65}, 0, 61, true, true, 4, 1e-3}};
    for (const Shape& S : fixed) { check_shape(S, rng, 0); ++done; }
    for (int k = 0; k < cases; ++k) {
        Shape S{};
        S.nax = ri(4, 20); S.nay = ri(4, 20); S.mxv = ri(1, 12); S.myv = ri(1, 12);
        S.n[0] = ri(9, 140); S.n[1] = ri(9, 140); S.n[2] = ri(3, 70);
        S.fold_x = rng() & 1; S.fold_y = rng() & 1;
        if (rng() % 4 == 0) { S.x_begin = ri(0, S.n[0] / 2); S.x_count = ri(1, S.n[0] - S.x_begin); } else { S.x_begin = 0; S.x_count = S.n[0]; }
        S.nf = ri(1, 20); S.h = 0.5e-3;
        check_shape(S, rng, 0);
        ++done;
    }
    printf("plan_check: %d shapes (%lld recognised lattices, %lld columns, %lld block records), %d violations\n", done, g_lattices, g_columns, g_records, g_fail);
    return g_fail ? 1 : 0;
}
