// Pure host planning of the lattice kernels: see olx_plan.h.  No HIP in this file -- it is also compiled by plain g++ with
// -fsanitize=address,undefined into the CPU-side checker (tools/plan_check.cpp).
#include "olx_plan.h"

#include <algorithm>
#include <climits>
#include <cmath>
#include <cstdlib>

namespace olxplan {

using namespace olx;

void build_slot_map(Lattice& L, int nsbp) {
    L.nsbp = nsbp; L.n_pad = L.nsa * nsbp * 64;
    L.slot_elem.assign((size_t)L.nsa * nsbp * 64, -1);
    for (int sa = 0; sa < L.nsa; ++sa)
        for (int sbb = 0; sbb < L.nsb; ++sbb)
            for (int ks = 0; ks < 4; ++ks)
                for (int bb = 0; bb < 4; ++bb)
                    for (int aa = 0; aa < 4; ++aa) {
                        const int a = 8 * sa + 4 * (ks & 1) + aa, b = 8 * sbb + 4 * (ks >> 1) + bb;
                        if (a < L.ax && b < L.ay)
                            L.slot_elem[((size_t)(sa * nsbp + sbb) * 4 + ks) * 16 + 4 * bb + aa] = L.cell[(size_t)a * L.ay + b];
                    }
}

// Kernel 2d precondition: the elements fill a regular ax x ay lattice in one z plane and the pitch is a whole number
// of voxels along x and y.  Fills L (slot map in 8 x 8 super-blocks of four 4 x 4 K-steps, padded with
// virtual elements) and the clamp / minimum-distance bounds including the virtual lattice points.
void detect_lattice(Lattice& L, bool flat, int n, const double* pos, const double spacing[3], const double lo[3], const double hi[3], double dmin) {
    L = Lattice();
    if (!flat || n < 16 || n > 16384) return;
    const double* X = pos;
    const double* Y = X + n;
    const double tol = 1e-10;
    auto axis = [&](const double* v, std::vector<double>& u) {
        u.assign(v, v + n);
        std::sort(u.begin(), u.end());
        size_t m = 0;
        for (size_t q = 0; q < u.size(); ++q)
            if (m == 0 || u[q] - u[m - 1] > tol) u[m++] = u[q];
        u.resize(m);
    };
    std::vector<double> xs, ys;
    axis(X, xs); axis(Y, ys);
    const int ax = (int)xs.size(), ay = (int)ys.size();
    if (ax < 2 || ay < 2 || (long long)ax * ay != n) return;
    const double px = (xs.back() - xs.front()) / (ax - 1), py = (ys.back() - ys.front()) / (ay - 1);
    for (int a = 0; a < ax; ++a) if (std::fabs(xs[a] - (xs[0] + a * px)) > tol) return;
    for (int b = 0; b < ay; ++b) if (std::fabs(ys[b] - (ys[0] + b * py)) > tol) return;
    const double rx = px / spacing[0], ry = py / spacing[1];
    const int mx = (int)std::llround(rx), my = (int)std::llround(ry);
    if (mx < 1 || my < 1 || std::fabs(rx - mx) > 1e-9 * mx || std::fabs(ry - my) > 1e-9 * my) return;
    const int nsa = (ax + 7) / 8, nsb = (ay + 7) / 8;
    if ((long long)nsa * nsb * 64 > 2LL * n) return;          // padding would more than double the contraction
    std::vector<int> cell((size_t)ax * ay, -1);
    for (int e = 0; e < n; ++e) {
        const int a = (int)std::llround((X[e] - xs[0]) / px), b = (int)std::llround((Y[e] - ys[0]) / py);
        if (a < 0 || a >= ax || b < 0 || b >= ay || cell[(size_t)a * ay + b] >= 0) return;
        cell[(size_t)a * ay + b] = e;
    }
    L.ax = ax; L.ay = ay; L.nsa = nsa; L.nsb = nsb;
    L.cell.swap(cell);
    build_slot_map(L, nsb);
    const int nsbp = (nsb + 1) & ~1;               // the bounds below also cover the padding rows of kernel 2e's pair tables
    // distance bounds over every lattice point of the padded array (virtual ones included: their G must stay finite)
    const double ez = pos[2 * (size_t)n];
    double min_d2 = 1e300; bool clamp = false;
    const double guard = 2.0 * dmin;
    for (int a = 0; a < std::max(8 * nsa, 16 * ((ax + 15) / 16)); ++a)   // (kernel 2f walks the columns in super-blocks of 16)
        for (int b = 0; b < 8 * nsbp; ++b) {          // (kernel 2e's pair table also covers the padding rows)
            const double p[3] = {xs[0] + a * px, ys[0] + b * py, ez};
            double d2 = 0;
            for (int k = 0; k < 3; ++k) {
                const double d = p[k] < lo[k] ? lo[k] - p[k] : (p[k] > hi[k] ? p[k] - hi[k] : 0.0);
                d2 += d * d;
            }
            min_d2 = std::min(min_d2, d2);
            if (d2 < guard * guard) clamp = true;
        }
    L.mx = mx; L.my = my;
    L.x0 = xs[0]; L.y0 = ys[0]; L.px = px; L.py = py; L.min_d2 = min_d2; L.clamp = clamp;
    L.ok = true;
}

// Kernel 2e: MFMA row tiles (16 rows) one plane pair needs over all cosets and parts -- per (coset, part)
// ceil(COS_P KX KY / 16) -- for a computed region of wx x wy voxels at lattice pitch (mx, my) voxels.
long long coset_tiles16(int wx, int wy, int mx, int my, int nt) {
    const int kxw = cos_kxw(nt);
    const int nsx = ((wx + 2 * mx - 1) / (2 * mx) + kxw - 1) / kxw, nsy = ((wy + my - 1) / my + COS_KYW - 1) / COS_KYW;
    long long t16 = 0;
    for (int rx = 0; rx < 2 * mx; ++rx)
        for (int ry = 0; ry < my; ++ry) {
            const int kxa = rx < wx ? (wx - 1 - rx) / (2 * mx) + 1 : 0, kya = ry < wy ? (wy - 1 - ry) / my + 1 : 0;
            for (int sx = 0; sx < nsx; ++sx)
                for (int sy = 0; sy < nsy; ++sy) {
                    const int KX = (sx + 1) * kxa / nsx - sx * kxa / nsx, KY = (sy + 1) * kya / nsy - sy * kya / nsy;
                    t16 += (COS_P * KX * KY + 15) / 16;
                }
        }
    return t16;
}

bool same_vector(const Steering& S, int f1, int m1, int f2, int m2) {
    const int n = S.n;
    for (int e = 0; e < n; ++e) {
        const size_t a = (size_t)f1 * n + S.perm[(size_t)m1 * n + e], b = (size_t)f2 * n + S.perm[(size_t)m2 * n + e];
        double dph = (S.delays[a] - S.delays[b]) * S.freq;
        dph -= std::nearbyint(dph);
        const double wa = S.apod[a] * S.area[a % n], wb = S.apod[b] * S.area[b % n];
        if (std::fabs(wa - wb) > 1e-12 * std::max(std::fabs(wa), std::fabs(wb))) return false;
        if (wa != 0.0 && std::fabs(dph) > 1e-9) return false;
    }
    return true;
}

// A column may store to ANY focus volume, so the search for an equal vector runs over every tile packed so far: mirror-partner
// foci share their columns wherever they sit in the sweep (a Wheel in its natural order has them at opposite ends).
Tiles pack_columns(const Steering& S, int maxc) {
    Tiles tl(1);
    for (int f = 0; f < S.F; ++f)
        for (int m = 0; m < S.n_img; ++m) {
            Col* hit = nullptr;
            for (size_t t = 0; t < tl.size() && !hit; ++t)
                for (size_t q = 0; q < tl[t].size() && !hit; ++q)
                    if (tl[t][q].ntgt < 4 && same_vector(S, tl[t][q].f, tl[t][q].m, f, m)) hit = &tl[t][q];
            if (!hit) {
                if ((int)tl.back().size() >= maxc) tl.emplace_back();
                tl.back().push_back(Col{f, m, 0, {-1, -1, -1, -1}});
                hit = &tl.back().back();
            }
            hit->tgt[hit->ntgt++] = f * 4 + m;
        }
    return tl;
}

void balance_store_targets(Tiles& tiles, int max_cols) {
    for (auto& t : tiles)
        for (size_t o = 0; o < t.size() && (int)t.size() < max_cols; ++o)
            if (t[o].ntgt > 2) {
                Col extra{t[o].f, t[o].m, 0, {-1, -1, -1, -1}};
                while (t[o].ntgt > 2) { extra.tgt[extra.ntgt++] = t[o].tgt[--t[o].ntgt]; t[o].tgt[t[o].ntgt] = -1; }
                t.push_back(extra);
            }
}

void coset_partition(CosetParams& Q, int kxw, int zb, int kyw) {
    const int px = Q.xs * Q.mx;      // voxels between two positions of a coset along x
    const int kx_max = (Q.nx - Q.x_lo + px - 1) / px, ky_max = (Q.ny - Q.y_lo + Q.my - 1) / Q.my;
    Q.nsx = (kx_max + kxw - 1) / kxw; Q.nsy = (ky_max + kyw - 1) / kyw;
    Q.kblocks = (Q.nz + zb - 1) / zb;
}

// blockIdx.x -> (coset, part, plane block), in the kernels' former decode order (the two blocks that write the two 64-byte halves of the
// same 128-byte lines get ids 8 apart = same XCD under round-robin dispatch)
bool build_coset_blocks(const CosetParams& Q, int zb, unsigned grp, int max_pos,
                        std::vector<CosetBlock>& blk, std::string& msg) {
    const int px = Q.xs * Q.mx;
    const unsigned nblk = (unsigned)(px * Q.my * Q.nsx * Q.nsy * Q.kblocks);
    blk.assign(nblk, CosetBlock{});
    const int wx = Q.nx - Q.x_lo, wy = Q.ny - Q.y_lo;
    for (unsigned id = 0; id < nblk; ++id) {
        unsigned b = id;
        int kblock, ry_lo = 0;
        unsigned gy = 1;      // y cosets that follow each other on an XCD (grp > kblocks: grp = kblocks * gy)
        if (grp > (unsigned)Q.kblocks && grp % (unsigned)Q.kblocks == 0 && (unsigned)Q.my % (grp / (unsigned)Q.kblocks) == 0 && nblk % (8 * grp) == 0) {
            gy = grp / (unsigned)Q.kblocks;
            const unsigned xcd = b % 8, sft = b / 8, inner = sft % grp;
            kblock = (int)(inner % (unsigned)Q.kblocks); ry_lo = (int)(inner / (unsigned)Q.kblocks);
            b = (sft / grp) * 8 + xcd;
        } else if (grp <= (unsigned)Q.kblocks && (Q.kblocks % grp) == 0 && nblk % (8 * grp) == 0) {
            const unsigned xcd = b % 8, sft = b / 8, kb_lo = sft % grp, u = (sft / grp) * 8 + xcd, part = (unsigned)Q.kblocks / grp;
            kblock = (int)(grp * (u % part) + kb_lo); b = u / part;
        } else { kblock = (int)(b % (unsigned)Q.kblocks); b /= (unsigned)Q.kblocks; }
        const int sy_part = (int)(b % (unsigned)Q.nsy); b /= (unsigned)Q.nsy;
        const int sx_part = (int)(b % (unsigned)Q.nsx); b /= (unsigned)Q.nsx;
        const unsigned myh = (unsigned)Q.my / gy;
        const int ry = (int)(b % myh) * (int)gy + ry_lo, rx = (int)(b / myh);
        const int kx_all = rx < wx ? (wx - 1 - rx) / px + 1 : 0, ky_all = ry < wy ? (wy - 1 - ry) / Q.my + 1 : 0;
        const int kx0 = sx_part * kx_all / Q.nsx, KX = (sx_part + 1) * kx_all / Q.nsx - kx0;
        const int ky0 = sy_part * ky_all / Q.nsy, KY = (sy_part + 1) * ky_all / Q.nsy - ky0;
        CosetBlock& B = blk[id];
        B.ibase = Q.x_lo + rx + px * kx0; B.jbase = Q.y_lo + ry + Q.my * ky0; B.k0 = kblock * zb;
        B.npos = (KX > 0 && KY > 0) ? KX * KY : 0; B.KY = KY > 0 ? KY : 1; B.ky_magic = 65536 / B.KY + 1; B.KX = KX > 0 ? KX : 0; B.pad_ = 0;
        if (max_pos > 0 && B.npos > max_pos) { msg = "a block part holds more than " + std::to_string(max_pos) + " positions"; return false; }
    }
    return true;
}

std::vector<int> build_store_jobs(const Tiles& tiles, int max_nt, int cols_per_nt, int jobs_per_tile, bool want_p, bool want_i) {
    const int ntiles = (int)tiles.size();
    std::vector<int> jobs((size_t)ntiles * max_nt * (jobs_per_tile + 1), -1);
    for (int t = 0; t < ntiles; ++t)
        for (int nt = 0; nt < max_nt; ++nt) {
            int* jb = &jobs[((size_t)t * max_nt + nt) * (jobs_per_tile + 1)];
            int cnt = 0;
            for (int c16 = 0; c16 < 16; ++c16) {
                const bool wantp = (c16 & 1) ? want_i : want_p;
                const size_t o = (size_t)nt * cols_per_nt + (c16 >> 1);
                if (!wantp || o >= tiles[t].size()) continue;
                for (int q = 0; q < 4; ++q) {
                    const int code = tiles[t][o].tgt[q];
                    if (code >= 0) jb[cnt++] = c16 | ((code & 3) << 4) | ((code >> 2) << 6);
                }
            }
            int lg = 0;
            while ((1 << lg) < cnt) ++lg;
            jb[jobs_per_tile] = lg;
        }
    return jobs;
}

// Foci of an externally supplied steering table (olx_set_steering; the run_simulation seam hands over delays only).  For the
// reference's geometric delays (bf/delay_methods/direct.py:28-38) tau_e = max(tof) - tof_e, every element satisfies
// |x - r_e| = S - s_e with s_e = c tau_e and one unknown S per focus; subtracting element 0's equation leaves a LINEAR system in
// (x, y, S) for a flat array:  -2 (r_e - r_0) . x + 2 (s_e - s_0) S = (s_e^2 - s_0^2) - (|r_e|^2 - |r_0|^2);  z follows from
// element 0 on the grid's side of the array.  Accepted only if the point reproduces all delays to 1e-6 m (lambda / 3750 at
// 400 kHz) -- arbitrary delay patterns have no such point and are reported as unknown.
bool infer_foci(bool flat, int n, int F, const double* pos, const double* delays, double c, double grid_z_mid, std::vector<double>& foci) {
    if (!flat || n < 4 || !delays) return false;
    const double* X = pos; const double* Y = X + n; const double* Z = Y + n;
    foci.assign(3 * (size_t)F, 0.0);
    for (int f = 0; f < F; ++f) {
        const double* tau = delays + (size_t)f * n;
        double A[3][4] = {{0}};   // normal equations [A | b] for u = (x, y, S)
        const double s0 = c * tau[0], q0 = X[0] * X[0] + Y[0] * Y[0];
        for (int e = 1; e < n; ++e) {
            const double se = c * tau[e];
            const double row[3] = {-2.0 * (X[e] - X[0]), -2.0 * (Y[e] - Y[0]), 2.0 * (se - s0)};
            const double rhs = (se * se - s0 * s0) - (X[e] * X[e] + Y[e] * Y[e] - q0);
            for (int i = 0; i < 3; ++i) {
                for (int j = 0; j < 3; ++j) A[i][j] += row[i] * row[j];
                A[i][3] += row[i] * rhs;
            }
        }
        for (int i = 0; i < 3; ++i) {   // Gaussian elimination with partial pivoting
            int piv = i;
            for (int r = i + 1; r < 3; ++r) if (std::fabs(A[r][i]) > std::fabs(A[piv][i])) piv = r;
            if (!(std::fabs(A[piv][i]) > 1e-300)) return false;
            if (piv != i) for (int j = 0; j < 4; ++j) std::swap(A[i][j], A[piv][j]);
            for (int r = 0; r < 3; ++r) {
                if (r == i) continue;
                const double m = A[r][i] / A[i][i];
                for (int j = i; j < 4; ++j) A[r][j] -= m * A[i][j];
            }
        }
        const double x = A[0][3] / A[0][0], y = A[1][3] / A[1][1], S = A[2][3] / A[2][2];
        const double dz2 = (S - s0) * (S - s0) - (x - X[0]) * (x - X[0]) - (y - Y[0]) * (y - Y[0]);
        if (!(dz2 > 0) || !std::isfinite(dz2)) return false;
        const double side = grid_z_mid >= Z[0] ? 1.0 : -1.0;
        const double z = Z[0] + side * std::sqrt(dz2);
        for (int e = 0; e < n; ++e) {
            const double d = std::sqrt((x - X[e]) * (x - X[e]) + (y - Y[e]) * (y - Y[e]) + (z - Z[e]) * (z - Z[e]));
            if (!(std::fabs(d - (S - c * tau[e])) <= 1e-6)) return false;
        }
        foci[3 * (size_t)f] = x; foci[3 * (size_t)f + 1] = y; foci[3 * (size_t)f + 2] = z;
    }
    return true;
}

double nearfield_s2(int n, const double* pos, const double origin[3], const double spacing[3], const int begin[3], const int count[3], double dclamp) {
    if (n <= 0) return 0.0;
    const double* P[3] = {pos, pos + n, pos + 2 * (size_t)n};
    const double dc2 = dclamp * dclamp;
    // candidates: every element for n <= 1024, else every stride-th one and the element nearest to the array's centroid
    // (regular arrays: the sum peaks next to the innermost elements, which have the most neighbours)
    const int stride = n <= 1024 ? 1 : (n + 1023) / 1024;
    int centre = 0;
    {
        double m[3] = {0, 0, 0}, best = 1e300;
        for (int a = 0; a < 3; ++a) { for (int e = 0; e < n; ++e) m[a] += P[a][e]; m[a] /= n; }
        for (int e = 0; e < n; ++e) {
            double d2 = 0;
            for (int a = 0; a < 3; ++a) d2 += (P[a][e] - m[a]) * (P[a][e] - m[a]);
            if (d2 < best) { best = d2; centre = e; }
        }
    }
    std::vector<int> cand;
    for (int e = 0; e < n; e += stride) cand.push_back(e);
    cand.push_back(centre);
    double s2max = 0.0;
    for (const int e0 : cand) {
        double v[3];
        for (int a = 0; a < 3; ++a) {
            long long i = std::llround((P[a][e0] - origin[a]) / spacing[a]);
            i = std::max<long long>(begin[a], std::min<long long>(i, (long long)begin[a] + count[a] - 1));
            v[a] = origin[a] + (double)i * spacing[a];
        }
        double s2 = 0.0;
        for (int e = 0; e < n; ++e) {
            const double dx = v[0] - P[0][e], dy = v[1] - P[1][e], dz = v[2] - P[2][e];
            s2 += 1.0 / std::max(dx * dx + dy * dy + dz * dz, dc2);
        }
        s2max = std::max(s2max, s2);
    }
    return s2max;
}

}  // namespace olxplan
