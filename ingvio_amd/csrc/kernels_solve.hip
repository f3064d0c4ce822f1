// kernels_solve.hip — K8 / K9 / K11 of the factored MSCKF path in SYMMETRIC form, on the FP64 matrix cores.
//
// The information-form update needs  M = (A Pcc + s^2 I)^-1 A  and  t = (A Pcc + s^2 I)^-1 b   (StateManager.cpp:399-405 after the
// push-through identity, see kernels_factored.hip).  A Pcc + s^2 I is a product of two symmetric matrices plus a shift: its
// eigenvalues are those of the SPD innovation matrix, but it is far from normal (A has rank n - 6), and Gauss-Jordan on it loses
// cond(A^1/2) more digits than the reference's SPD solve (measured: dx off by 3e-5 at a 100x inflated prior, tests/test_gpu_pinning.py).
// With  Pcc = L D L^T  (L unit BLOCK lower, D block diagonal with SPD 4x4 blocks: block pivots keep the dependent chain per
// panel at two reciprocals):
//        A Pcc + s^2 I = L^-T (L^T A L + s^2 D^-1) D L^T,      W := L^T A L + s^2 D^-1   SPD, eigenvalues >= s^2 / max(D)
//        M = (L^-T D^-1) W^-1 (L^T A),   t = (L^-T D^-1) W^-1 (L^T b)
// and with  W = L2 D2 L2^T:   M = R2' D2^-1 R1'^T,   R2' = (L^-T D^-1) L2^-T,   R1' = (A L) L2^-T.
// Both factorisations are blocked LDL^T (panels of 4 pivots) on 16x16 tiles in the MFMA C/D layout, as in the gate kernel
// (gate_kernel.h), with CARRIED ROWS: rows C appended below a matrix that is being factorised come out as C L^-T D^-1 (their
// "L entries") / C L^-T (before the D scaling) — so carrying the identity through the first factorisation yields L^-T D^-1, and
// carrying [A L ; b^T L] and L^-T D^-1 through the second yields R1' D2^-1 and R2' without any triangular solve.
// The three products (A L, L^T (A L), R2' D2^-1 R1'^T) are MFMA GEMMs on operands staged in LDS.  No pivot search, no atomics, one
// workgroup barrier per panel (the panel buffer is double-buffered).
// One workgroup of 4 waves per filter; tiles are dealt round-robin to the waves.  gfx950 only.
#include <stdlib.h>
#include <string.h>
#include "launch_factored.h"

typedef double double4_f __attribute__((ext_vector_type(4)));

namespace {

template <int NC>
struct SolveCfg {
    static constexpr int NT = (NC + 15) / 16, NP = 16 * NT;          // tile rows / padded size of the n x n matrices
    static constexpr int NR1 = (NC + 1 + 15) / 16, R1ROWS = 16 * NR1; // rows of [A L ; b^T L]: the b row sits at row NC
    static constexpr int NW = 8, RW = 2;                              // waves per filter, tile rows a wave may own
    static constexpr int MROWS = R1ROWS > NP ? R1ROWS : NP, LDM = NP + 1;
    static constexpr int PANROWS = NP + R1ROWS + NP;
    static constexpr int MP = (NC + 3) & ~3;
    static constexpr int LDSS = NC + 6 + 1;                          // row stride of the staged window block of P (up to NC + 6 columns; odd)
    static constexpr size_t lds_bytes() { return sizeof(double) * (2 * (size_t)MROWS * LDM + 2 * (size_t)PANROWS * 4 + 4 * NP) + sizeof(int) * NP; }
};

// A tile ROW owned by a wave: its 16 rows live at pan rows [arow, arow + 16), its tiles are the column tiles c0..c1.
// kind 0: row of the matrix being factorised (tile row rt, tiles 0..rt); 1: carried row with a full set of tiles (rt = its index);
// 2: carried row of an upper-triangular block (tiles rt..NT-1; nothing happens to it before the panel reaches its diagonal tile).
struct TileRow { int kind, rt, arow, c0, c1; bool valid; };

// The tile rows of the two factorisations, in descending tile count.
template <int NC, int PHASE>
constexpr TileRow row_desc(int i)
{
    using C = SolveCfg<NC>;
    TileRow r{ 0, 0, 0, 1, 0, true };
    if (PHASE == 2 && i < C::NR1) { r.kind = 1; r.rt = i; r.arow = C::NP + 16 * i; r.c0 = 0; r.c1 = C::NT - 1; return r; }
    const int q = PHASE == 2 ? i - C::NR1 : i, m = C::NT - (q >> 1);
    if (q & 1) { r.kind = 2; r.rt = C::NT - m; r.arow = C::NP + (PHASE == 2 ? C::R1ROWS : 0) + 16 * r.rt; r.c0 = r.rt; r.c1 = C::NT - 1; }
    else { r.kind = 0; r.rt = m - 1; r.arow = 16 * (m - 1); r.c0 = 0; r.c1 = m - 1; }
    return r;
}

// Deals the tile rows to the NW waves at compile time: every row goes to the wave that owns the fewest tiles so far.
template <int NC, int PHASE>
struct Deal {
    using C = SolveCfg<NC>;
    static constexpr int NROWS = (PHASE == 2 ? C::NR1 : 0) + 2 * C::NT;
    TileRow rows[C::NW][C::RW];
    int count[C::NW];
    constexpr Deal() : rows{}, count{}
    {
        int load[C::NW] = {};
        for (int w = 0; w < C::NW; ++w) { count[w] = 0; for (int q = 0; q < C::RW; ++q) rows[w][q] = TileRow{ 0, 0, 0, 1, 0, false }; }
        for (int i = 0; i < NROWS; ++i) {
            const TileRow r = row_desc<NC, PHASE>(i);
            int best = 0;
            for (int w = 1; w < C::NW; ++w) if (load[w] < load[best]) best = w;
            load[best] += r.c1 - r.c0 + 1;
            rows[best][count[best] < C::RW ? count[best] : C::RW - 1] = r;
            ++count[best];
        }
    }
    constexpr bool fits() const { for (int w = 0; w < C::NW; ++w) if (count[w] > C::RW) return false; return true; }
};

// One blocked LDL^T sweep (panels of 4 pivots) as seen by wave W: every structural decision (which tiles exist, which are still
// live at a given panel) is a compile-time constant, the only run-time loop is over the four panels of a tile column.
// emit(own row w, pan row, pivot column, x, x / d): the entry of the row transformed by L_d^-T and its factor entry.
template <int NC, int PHASE, int W, class Emit>
__device__ __forceinline__ void ldl_sweep(double4_f (&T)[SolveCfg<NC>::RW][SolveCfg<NC>::NT], double (*pan)[SolveCfg<NC>::PANROWS][4],
                                          int lane, Emit emit, double* __restrict__ dinv, int* bad)
{
    using C = SolveCfg<NC>;
    constexpr int NT = C::NT, RW = C::RW;
    constexpr Deal<NC, PHASE> D{};
    const int kq = lane >> 4, l15 = lane & 15;
#pragma unroll
    for (int tj0 = 0; tj0 < NT; ++tj0) {
#pragma unroll 1
        for (int kk = 0; kk < 4; ++kk) {
            const int k = 4 * tj0 + kk, buf = k & 1, cb = 4 * kk;
#ifdef INGVIO_DBG_STAMPS
            if (PHASE == 1 && W == 0 && k < 8) dbg_stamp(8 + 5 * k);
#endif
            if (l15 >= cb && l15 < cb + 4) {
#pragma unroll
                for (int w = 0; w < RW; ++w) {
                    if (D.rows[W][w].valid && tj0 >= D.rows[W][w].c0 && tj0 <= D.rows[W][w].c1) {
#pragma unroll
                        for (int r = 0; r < 4; ++r) pan[buf][D.rows[W][w].arow + kq + 4 * r][l15 - cb] = T[w][tj0][r];
                    }
                }
            }
            // The inverse of the 4 x 4 pivot block is formed by ONE wave - the owner of the tile row it lies in, from the panel rows it
            // has just written itself (a hand-over inside the wave, no workgroup barrier) - and left in dinv[16 k ..] for the others,
            // who read their column after the panel's barrier.  Formed by every wave for itself (round 2-3) it was the longest
            // stretch of a panel: ~100 instructions x 4 waves per SIMD, issue-bound at 1500-1700 cycles of a panel's 3700
            // (shader-clock stamps of a workgroup in the middle of the grid).
            bool owner = false;                             // folds to a constant: tj0 is unrolled, the deal is constexpr
#pragma unroll
            for (int w = 0; w < RW; ++w) owner |= D.rows[W][w].valid && D.rows[W][w].kind == 0 && D.rows[W][w].rt == tj0;
            if (owner) {
                __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
                __builtin_amdgcn_wave_barrier();
                double a[4][4];
#pragma unroll
                for (int ra = 0; ra < 4; ++ra) {
                    const double2* pr = reinterpret_cast<const double2*>(pan[buf][4 * k + ra]);
                    const double2 u0 = pr[0], u1 = pr[1];
                    a[ra][0] = u0.x; a[ra][1] = u0.y; a[ra][2] = u1.x; a[ra][3] = u1.y;
                }
                // inverse of the SPD 4x4 pivot block [[E, F], [F^T, G]] by 2x2 blocks (uniform data): two dependent reciprocals
                const double e00 = a[0][0], e10 = a[1][0], e11 = a[1][1];
                const double f00 = a[2][0], f01 = a[3][0], f10 = a[2][1], f11 = a[3][1];          // F[i][j] = block(i, 2 + j) = a[2 + j][i]
                const double g00 = a[2][2], g10 = a[3][2], g11 = a[3][3];
                const double detE = e00 * e11 - e10 * e10, rE = fast_rcp(detE);
                const double ei00 = e11 * rE, ei10 = -e10 * rE, ei11 = e00 * rE;                   // E^-1
                const double h00 = ei00 * f00 + ei10 * f10, h01 = ei00 * f01 + ei10 * f11;         // H = E^-1 F
                const double h10 = ei10 * f00 + ei11 * f10, h11 = ei10 * f01 + ei11 * f11;
                const double s00 = g00 - (f00 * h00 + f10 * h10), s10 = g10 - (f01 * h00 + f11 * h10), s11 = g11 - (f01 * h01 + f11 * h11);   // S = G - F^T H
                const double detS = s00 * s11 - s10 * s10, rS = fast_rcp(detS);
                const double si00 = s11 * rS, si10 = -s10 * rS, si11 = s00 * rS;                   // S^-1
                if (!(e00 > 0.0) || !(detE > 0.0) || !(s00 > 0.0) || !(detS > 0.0)) *bad = 1;
                const double q00 = h00 * si00 + h01 * si10, q01 = h00 * si10 + h01 * si11;         // Q = H S^-1
                const double q10 = h10 * si00 + h11 * si10, q11 = h10 * si10 + h11 * si11;
                // block^-1 = [[E^-1 + Q H^T, -Q], [-Q^T, S^-1]]
                const double v00 = ei00 + q00 * h00 + q01 * h01, v10 = ei10 + q10 * h00 + q11 * h01, v11 = ei11 + q10 * h10 + q11 * h11;
                if (lane == 0) {                            // row-major 4 x 4 (symmetric)
                    double2* o = reinterpret_cast<double2*>(dinv + 16 * k);
                    o[0] = make_double2(v00, v10); o[1] = make_double2(-q00, -q01);
                    o[2] = make_double2(v10, v11); o[3] = make_double2(-q10, -q11);
                    o[4] = make_double2(-q00, -q10); o[5] = make_double2(si00, si10);
                    o[6] = make_double2(-q01, -q11); o[7] = make_double2(si10, si11);
                }
            }
            lds_barrier();
#ifdef INGVIO_DBG_STAMPS
            if (PHASE == 1 && W == 0 && k < 8) dbg_stamp(9 + 5 * k);
#endif
            // column kq of the inverse (= its row kq): this lane's coefficients for (row) block^-1
            const double2* ci = reinterpret_cast<const double2*>(dinv + 16 * k + 4 * kq);
            const double2 ca = ci[0], cbb = ci[1];
            const double c0 = ca.x, c1 = ca.y, c2 = cbb.x, c3 = cbb.y;
#ifdef INGVIO_DBG_STAMPS
            if (PHASE == 1 && W == 0 && k < 8) { if (c0 + c1 + c2 + c3 == 1.2345e-300) *bad = 1; dbg_stamp(10 + 5 * k); }
#endif
            auto xinv = [&](int row) {                      // entry kq of (pan row) block^-1
                const double2* pr = reinterpret_cast<const double2*>(pan[buf][row]);
                const double2 u0 = pr[0], u1 = pr[1];
                return fma(u1.y, c3, fma(u1.x, c2, fma(u0.y, c1, u0.x * c0)));
            };
            double xb[NT], xa[RW];
#pragma unroll
            for (int c = tj0; c < NT; ++c) {               // B operands: -(matrix rows of the live column tiles) block^-1
                const int rb = 16 * c + l15;
                const double x = xinv(rb);
                xb[c] = (c == tj0 && rb <= 4 * k + 3) ? 0.0 : -x;
            }
#pragma unroll
            for (int w = 0; w < RW; ++w) {
                const bool live = D.rows[W][w].valid && D.rows[W][w].c1 >= tj0 && (D.rows[W][w].kind != 2 || D.rows[W][w].rt <= tj0);
                xa[w] = 0.0;
                if (live) {
                    const int ra = D.rows[W][w].arow + l15;
                    const double x = pan[buf][ra][kq];      // A operand: the row's panel entries as they are
                    if (D.rows[W][w].kind != 0 || ra >= 4 * k) emit(w, ra, 4 * k + kq, k, x, xinv(ra));
                    xa[w] = (D.rows[W][w].kind == 0 && ra <= 4 * k + 3) ? 0.0 : x;      // pivot rows and everything above: finished
                }
            }
#ifdef INGVIO_DBG_STAMPS
            if (PHASE == 1 && W == 0 && k < 8) dbg_stamp(11 + 5 * k);
#endif
            // trailing update: T(row, c) -= R_row block^-1 R_c^T
#pragma unroll
            for (int w = 0; w < RW; ++w) {
                const bool live = D.rows[W][w].valid && (D.rows[W][w].kind != 2 || D.rows[W][w].rt <= tj0);
#pragma unroll
                for (int c = tj0; c < NT; ++c)
                    if (live && c >= D.rows[W][w].c0 && c <= D.rows[W][w].c1)
                        T[w][c] = __builtin_amdgcn_mfma_f64_16x16x4f64(xa[w], xb[c], T[w][c], 0, 0, 0);
            }
        }
    }
    lds_barrier();
}

template <int NC>
struct SolveArgs {
    const double* P; const int* sCol; double *X, *Y; double (*pan)[SolveCfg<NC>::PANROWS][4]; double *sD1inv, *sD2inv;
    const double* Apart; const int* chunk_used; int G, rstride, bl, ncol, ld, lane; double var;
    // gauge-reduced form (see k_info_solve): the solve runs on the ncol = ncolF - 6 difference coordinates; index i of the
    // reduced system is column i + (i >= ref6 ? 6 : 0) of the gram partials (row stride ncolF + 1), sRef[i] is the state column
    // of the same component in the reference clone's block
    const int* sRef; int ncolF, ref6;
};

// Everything wave W does between the set-up and the final product: both factorisations and the two products in between.
template <int NC, int W, bool RED>
__device__ __forceinline__ void solve_wave(const SolveArgs<NC>& a, int* bad)
{
    using C = SolveCfg<NC>;
    constexpr int NT = C::NT, NP = C::NP, R1ROWS = C::R1ROWS, RW = C::RW, LDM = C::LDM, MROWS = C::MROWS, NTH = 64 * C::NW;
    constexpr Deal<NC, 1> D1{};
    constexpr Deal<NC, 2> D2{};
    static_assert(D1.fits() && D2.fits(), "a wave would own more tile rows than RW");
    const int lane = a.lane, kq = lane >> 4, l15 = lane & 15, ncol = a.ncol, ld = a.ld;
    double* X = a.X; double* Y = a.Y;
    double4_f T[RW][NT];
    // The A fragments of this wave's [A ; b^T] tile rows (product R1 = [A ; b^T] L below): the loads are issued now and
    // complete under factorisation 1.  A is exactly symmetric (k_feat_gram2 builds both halves from the same sums): element
    // (row, col) is read at (col, row), which runs along the 16 lanes of a fragment (coalesced); the b row is column ncol.
    double af[RW][NT][4];
    unsigned long long used_mask = 0;                          // G <= 16 chunks (ingvio_ctx_create; INGVIO_GRAM_CHUNKS of the variant build: <= 64)
    for (int g = 0; g < a.G; ++g) used_mask |= (a.chunk_used[a.bl * a.G + g] != 0 ? 1ull : 0ull) << g;
    // Round 6 (last day): with the chunk loop INSIDE the element loop and a run-time trip count every element was a load -> wait ->
    // select -> add of its own - 16 to 32 dependent round trips per lane, 17.6 k of the kernel's 129 k cycles (shader-clock stamps),
    // nothing of it under the factorisation.  ONE partial (what a full batch and, behind k_chunk_sum, a few filters have): every load is
    // issued here from a clamped address and NOT looked at - the padding is selected away right before the product, after factorisation
    // 1 (the barriers in between are LDS-only: __syncthreads would drain the loads).  Several partials: the chunk loop outermost, a
    // chunk's loads in flight together (G round trips), added in chunk order as before.
    unsigned okm[RW];
    const bool one_partial = a.G == 1;
#pragma unroll
    for (int w = 0; w < RW; ++w) {
        okm[w] = 0u;
        if (D2.rows[W][w].valid && D2.rows[W][w].kind == 1) {
            const int row = 16 * D2.rows[W][w].rt + l15;
            int eoff[NT][4];
#pragma unroll
            for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const int col = 16 * kt + 4 * s + kq;
                    const bool ok = col < ncol && (row < ncol || row == NC);      // row NC = b^T
                    const int colF = RED ? col + (col >= a.ref6 ? 6 : 0) : col, rowF = RED ? row + (row >= a.ref6 ? 6 : 0) : row;
                    eoff[kt][s] = ok ? colF * (a.ncolF + 1) + (row == NC ? a.ncolF : rowF) : 0;
                    okm[w] |= (ok ? 1u : 0u) << (4 * kt + s);
                    af[w][kt][s] = 0.0;
                }
            if (one_partial) {
                const double* Ap = a.Apart + (size_t)a.bl * a.rstride;
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[w][kt][s] = Ap[eoff[kt][s]];
            } else {
                for (int g = 0; g < a.G; ++g) {
                    const double* Ap = a.Apart + ((size_t)a.bl * a.G + g) * a.rstride;
                    const bool used = (used_mask >> g) & 1ull;
                    double x[NT][4];
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int s = 0; s < 4; ++s) x[kt][s] = Ap[eoff[kt][s]];
#pragma unroll
                    for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                        for (int s = 0; s < 4; ++s) af[w][kt][s] += (used && ((okm[w] >> (4 * kt + s)) & 1u)) ? x[kt][s] : 0.0;
                }
            }
        }
    }
    // ================= factorisation 1: Pcc = L D L^T, identity carried =================
    {
        // The window block of P was staged in LDS by the whole workgroup (k_info_solve: S, in the X / Y area, row stride LDS): every
        // entry of Pdd is four of its elements.  Read from global memory - 128 loads of 8 bytes per lane, 2048 load instructions per
        // CU with two workgroups resident - the set-up took as long as a factorisation (the address path, not the bandwidth).
        constexpr int LDS = SolveCfg<NC>::LDSS;
        const double* S = X;
        auto wcol = [&](int c) { return RED ? c + (c >= a.ref6 ? 6 : 0) : c; };          // reduced index -> window column
        auto wref = [&](int c) { return a.ref6 + wcol(c) % 6; };                          // same component of the reference clone
        int scol_c[NT], scol_r[RW][4], sref_c[NT], sref_r[RW][4];
#pragma unroll
        for (int c = 0; c < NT; ++c) { const int e = min(16 * c + l15, ncol - 1); scol_c[c] = wcol(e); sref_c[c] = RED ? wref(e) : 0; }
#pragma unroll
        for (int w = 0; w < RW; ++w)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int e = min((D1.rows[W][w].valid ? 16 * D1.rows[W][w].rt : 0) + kq + 4 * r, ncol - 1);
                scol_r[w][r] = wcol(e); sref_r[w][r] = RED ? wref(e) : 0;
            }
#pragma unroll
        for (int w = 0; w < RW; ++w)
#pragma unroll
            for (int c = 0; c < NT; ++c) {
                const int col = 16 * c + l15;
                if (D1.rows[W][w].valid && D1.rows[W][w].kind == 0 && c <= D1.rows[W][w].c1) {      // (compile-time: the tile exists)
                    // Round 6: the tile's reads are unconditional - the indices are clamped into the window - and issued together, the
                    // padding is selected afterwards: under `row < ncol && col < ncol` every element was an exec-masked branch with its
                    // own LDS round trip (32 in a row per lane).  One pin per tile: all 128 reads at once spill (128 VGPRs at 4 waves).
                    double x[4];
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        x[r] = S[scol_r[w][r] * LDS + scol_c[c]];
                        if (RED) {                                               // covariance of the DIFFERENCES to the reference clone
                            const double x1 = S[scol_r[w][r] * LDS + sref_c[c]], x2 = S[sref_r[w][r] * LDS + scol_c[c]], x3 = S[sref_r[w][r] * LDS + sref_c[c]];
                            x[r] = (x[r] - x1) - (x2 - x3);
                        }
                    }
                    asm volatile("" : "+v"(x[0]), "+v"(x[1]), "+v"(x[2]), "+v"(x[3]));
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int row = 16 * D1.rows[W][w].rt + kq + 4 * r;
                        T[w][c][r] = (row < ncol && col < ncol) ? x[r] : (row == col ? 1.0 : 0.0);
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 4; ++r) T[w][c][r] = (16 * D1.rows[W][w].rt + kq + 4 * r == col) ? 1.0 : 0.0;
                }
            }
        lds_barrier();                                        // every wave holds its tiles of Pdd: the staging area becomes X and Y (LDS only: the A fragments stay in flight)
        for (int e = W * 64 + lane; e < 2 * MROWS * LDM; e += NTH) X[e] = 0.0;
        lds_barrier();
        auto emit = [&](int w, int row, int col, int k, double x, double xs) {
            if (D1.rows[W][w].kind == 0) X[row * LDM + col] = row > 4 * k + 3 ? xs : (row == col ? 1.0 : 0.0);   // L: identity pivot blocks
            else Y[(row - NP) * LDM + col] = xs;                                                             // L^-T D^-1 (upper)
        };
        dbg_stamp(1);
        ldl_sweep<NC, 1, W>(T, a.pan, lane, emit, a.sD1inv, bad);
    }
    dbg_stamp(2);
    // ================= second matrix W = L^T A L + s^2 D^-1 and its carried rows [A L ; b^T L], L^-T D^-1 =================
#pragma unroll
    for (int w = 0; w < RW; ++w)
#pragma unroll
        for (int c = 0; c < NT; ++c) {
            T[w][c] = double4_f{ 0.0, 0.0, 0.0, 0.0 };
            if (D2.rows[W][w].valid && D2.rows[W][w].kind == 2 && c >= D2.rows[W][w].c0) {        // R2 = L^-T D^-1 from factorisation 1
#pragma unroll
                for (int r = 0; r < 4; ++r) T[w][c][r] = Y[(16 * D2.rows[W][w].rt + kq + 4 * r) * LDM + 16 * c + l15];
            }
        }
    lds_barrier();                                            // every wave has its R2 tiles: Y may be overwritten
    // ---- R1 = [A ; b^T ; 0] L  (A symmetric, from the gram partials in global memory; L in X) ----
#pragma unroll
    for (int w = 0; w < RW; ++w) {
        if (D2.rows[W][w].valid && D2.rows[W][w].kind == 1) {
            constexpr int dummy = 0; (void)dummy;
            const int i = D2.rows[W][w].rt;
            if (one_partial) {                                  // the fragments requested before factorisation 1: padding selected away now
#pragma unroll
                for (int kt = 0; kt < NT; ++kt)
#pragma unroll
                    for (int s = 0; s < 4; ++s) af[w][kt][s] = ((okm[w] >> (4 * kt + s)) & 1u) ? af[w][kt][s] : 0.0;
            }
#pragma unroll
            for (int j = 0; j < NT; ++j) {
                double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
                for (int kt = j; kt < NT; ++kt)
#pragma unroll
                    for (int s = 0; s < 4; ++s)
                        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[w][kt][s], X[(16 * kt + 4 * s + kq) * LDM + 16 * j + l15], acc, 0, 0, 0);
                T[w][j] = acc;
#pragma unroll
                for (int r = 0; r < 4; ++r) Y[(16 * i + kq + 4 * r) * LDM + 16 * j + l15] = acc[r];
            }
        }
    }
    __syncthreads();
    dbg_stamp(3);
    // ---- W = L^T (A L) + s^2 D^-1, lower tiles ----
#pragma unroll
    for (int w = 0; w < RW; ++w) {
        if (D2.rows[W][w].valid && D2.rows[W][w].kind == 0) {
            constexpr int dummy = 0; (void)dummy;
            const int i = D2.rows[W][w].rt;
            double4_f acc[NT];
#pragma unroll
            for (int j = 0; j < NT; ++j) acc[j] = double4_f{ 0.0, 0.0, 0.0, 0.0 };
#pragma unroll
            for (int kt = 0; kt < NT; ++kt) {
                if (kt >= i) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        const int kr = 16 * kt + 4 * s + kq;
                        const double afv = X[kr * LDM + 16 * i + l15];                  // A[i'][k'] = L[k][i]
#pragma unroll
                        for (int j = 0; j < NT; ++j)
                            if (j <= i) {
                                const double bfv = (NC < NP && kr >= NC) ? 0.0 : Y[kr * LDM + 16 * j + l15];   // row NC of Y is b^T L, not a row of A L
                                acc[j] = __builtin_amdgcn_mfma_f64_16x16x4f64(afv, bfv, acc[j], 0, 0, 0);
                            }
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < NT; ++j)
                if (j <= i) {
                    if (j == i) {                             // + s^2 D^-1: the 4x4 inverse pivot blocks of factorisation 1
#pragma unroll
                        for (int r = 0; r < 4; ++r) {
                            const int rr = kq + 4 * r;        // row within the tile; its pivot block is r (rows 4 r .. 4 r + 3), row kq inside
                            if ((l15 >> 2) == r) acc[j][r] += a.var * a.sD1inv[16 * (4 * i + r) + 4 * kq + (l15 & 3)];
                            (void)rr;
                        }
                    }
                    T[w][j] = acc[j];
                }
        }
    }
    __syncthreads();                                          // X (L) and Y (A L) are dead: they become the outputs of sweep 2
    dbg_stamp(4);
    for (int e = W * 64 + lane; e < MROWS * LDM; e += NTH) Y[e] = 0.0;   // R2' is upper triangular by tiles
    // ================= factorisation 2: W = L2 D2 L2^T with [A L ; b^T L] and L^-T D^-1 carried =================
    {
        auto emit = [&](int w, int row, int col, int k, double x, double xs) {
            if (D2.rows[W][w].kind == 1) X[(row - NP) * LDM + col] = xs;                          // R1' D2^-1
            else if (D2.rows[W][w].kind == 2) Y[(row - NP - R1ROWS) * LDM + col] = x;             // R2'      (L2 itself is not needed)
        };
        ldl_sweep<NC, 2, W>(T, a.pan, lane, emit, a.sD1inv, bad);      // the inverse blocks of factorisation 1 are dead by now: same slots
    }
}

// Gauge-reduced form (RED, the RemoveLost form of the Jacobians): every row of H_j = V^T Hx annihilates a common 6-vector added to
// all clones of the window (a rigid motion of the window: D_o u = [p_f]x w - [p_f]x w = 0 for the rotation part,
// RemoveLostUpdate.cpp:476-482, and -t = Hf (-t) for the translation part, projected out by V), so in exact arithmetic
// A (1_C (x) I_6) = 0 and b^T (1_C (x) I_6) = 0: A is a block Laplacian.  The rounded A of k_feat_gram2 (block-sparse term minus a
// rank-3 term) violates this at eps |A|, and with an inflated prior the posterior amplifies exactly that leak (measured and
// reproduced in numpy, DESIGN 4.1b: an eps-sized symmetric perturbation of A moves the window block of P by 1e-5 relative at a
// 1e4 x prior and s = 1e-3, the cancellation in P - K H P itself only by 1e-9).  In the difference coordinates d_c = u_c - u_ref
// the information is A with the reference clone's block row / column deleted (full rank, no gauge), the covariance is
//     Pdd = P(c,c') - P(c,ref) - P(ref,c') + P(ref,ref)     (exact differences of the given entries)
// and with Mr = (Ar Pdd + s^2 I)^-1 Ar, tr = (Ar Pdd + s^2 I)^-1 br the n x n solution the apply kernel consumes is Mr bordered by
// the reference block that makes every block row / column sum vanish: M = T^-T diag(0, Mr) T^-1.  The reference clone is the one
// with the largest translation information.  The solve shrinks from 6 C to 6 (C - 1) columns (5 -> 4 tile rows at 11 clones).
template <int NCF, bool RED>
__global__ __launch_bounds__(512) __attribute__((amdgpu_waves_per_eu(4, 4))) void k_info_solve(
    CovView cv, FrameView fv, int b0, const double* __restrict__ Apart, const int* __restrict__ chunk_used, int G, int rstride,
    const double* __restrict__ noise_all, double* __restrict__ Mall, int mstride, double* __restrict__ Pcall, int ystride,
    double* __restrict__ dx_all, int* __restrict__ m_out, int* __restrict__ nc_out, int* __restrict__ status,
    const int* __restrict__ marg_idx, int* __restrict__ pc_base_out)
{
    constexpr int NC = RED ? NCF - 6 : NCF;
    using Cfg = SolveCfg<NC>;
    constexpr int NT = Cfg::NT, NP = Cfg::NP, NR1 = Cfg::NR1, NW = Cfg::NW;
    constexpr int LDM = Cfg::LDM, MROWS = Cfg::MROWS, PANROWS = Cfg::PANROWS, NTH = 64 * NW;
    constexpr int MPF = (NCF + 3) & ~3, NPF = 16 * ((NCF + 15) / 16), CF = NCF / 6;
    extern __shared__ __attribute__((aligned(16))) char smem_raw[];
    double* X = reinterpret_cast<double*>(smem_raw);                 // MROWS x LDM: L, later R1' D2^-1
    double* Y = X + (size_t)MROWS * LDM;                             // MROWS x LDM: L^-T D^-1, then A L, later R2'
    double (*pan)[PANROWS][4] = reinterpret_cast<double (*)[PANROWS][4]>(Y + (size_t)MROWS * LDM);
    double* sD1inv = reinterpret_cast<double*>(pan) + 2 * (size_t)PANROWS * 4;      // NP / 4 inverse pivot blocks, 16 doubles each
    double* sD2inv = nullptr;
    int* sCol = reinterpret_cast<int*>(sD1inv + 4 * NP);                             // NP
    __shared__ int sBad, sRefSlot;
    // set-up tables, dead once every wave holds its indices in registers: they borrow the second panel buffer, which is first
    // written in panel 1, i.e. after the barrier of panel 0 that every wave reaches with its set-up done (two workgroups per CU
    // need the kernel's LDS under 80 KB)
    int* sColF = reinterpret_cast<int*>(&pan[1][0][0]);              // NPF: state column of every window column
    int* sRef = sColF + NPF;                                         // NP: same component in the reference clone's block
    double* sTr = reinterpret_cast<double*>(sRef + NPF + (NPF & 1)); // 16 clones x 16 chunk lanes
    static_assert(sizeof(int) * (2 * NPF + 2) + sizeof(double) * 256 <= sizeof(double) * PANROWS * 4, "set-up tables exceed a panel buffer");
    dbg_stamp(60);
    const int bl = blockIdx.x, b = b0 + bl, tid = threadIdx.x, wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;
    const int kq = lane >> 4, l15 = lane & 15;
    const int C = fv.n_clones[b], ncolF = 6 * C, n = cv.n[b], ld = cv.ldp;
    double* dx = dx_all + (size_t)b * ld;
    int total = 0;
    for (int g = 0; g < G; ++g) total += chunk_used[bl * G + g];
    if (total == 0) {
        for (int r = tid; r < n; r += NTH) dx[r] = 0.0;
        if (tid == 0) { m_out[bl] = 0; nc_out[bl] = ncolF; pc_base_out[bl] = -1; }
        return;
    }
    const double* P = cov_ptr(cv, b);
    for (int c = tid; c < NPF; c += NTH) { const int cc = c < ncolF ? c : 0; sColF[c] = fv.clone_idx[(size_t)b * fv.cmax + cc / 6] + cc % 6; }
#ifndef SOLVE_REF_BY_INFO
#define SOLVE_REF_BY_INFO 0      // 1: the reference clone of the gauge reduction = the one with the largest translation information (rounds 3-5)
#endif
    if (SOLVE_REF_BY_INFO && RED && tid < 256) {             // translation information of every clone: trace of its (p, p) block
        // lane (clone c, chunk g): three loads; the per-clone sums over the chunks are taken in chunk order below, so the choice
        // of the reference clone - and with it every bit of the result - does not depend on the order the lanes ran in
        const int c = tid >> 4, g0 = tid & 15;
        double tr = 0.0;
        if (c < C) {
            for (int g = g0; g < G; g += 16) {
                const double* Ap = Apart + ((size_t)bl * G + g) * rstride;
                const double u = chunk_used[bl * G + g] ? 1.0 : 0.0;
                const size_t e = (size_t)(6 * c + 3) * (ncolF + 1) + 6 * c + 3;
                tr += u * ((Ap[e] + Ap[e + ncolF + 2]) + Ap[e + 2 * (ncolF + 2)]);
            }
        }
        sTr[tid] = tr;
    }
    if (tid == 0) sBad = 0;
    __syncthreads();
    dbg_stamp(61);
    static_assert((NC + 6) * Cfg::LDSS <= 2 * MROWS * LDM, "the staged window block exceeds the X / Y area");
    {
        // S[i][j] = P(window column j, window column i): j runs along the coalesced direction.  Round 6: a thread's loads are ALL issued
        // before its first LDS store (as a loop `X[..] = P[..]` the stores to X - LDS, like the index table the addresses come from - kept
        // every iteration's loads behind the previous store: 9 dependent global round trips per thread in front of everything else)
        constexpr int SPT = (NCF * NCF + NTH - 1) / NTH;
        const int nn = ncolF * ncolF;
        const float inv = 1.0f / (float)ncolF;
        double sv[SPT];
        int so[SPT];
#pragma unroll
        for (int u = 0; u < SPT; ++u) {
            const int e = min(tid + NTH * u, nn - 1);
            const int i = (int)(((float)e + 0.5f) * inv), j = e - i * ncolF;      // e / ncolF (exact: e < 2^13, the quotient's fraction is >= 0.5 / 72 off an integer)
            sv[u] = P[sColF[j] + (size_t)sColF[i] * ld];
            so[u] = i * Cfg::LDSS + j;
        }
#pragma unroll
        for (int u = 0; u < SPT; ++u) if (tid + NTH * u < nn) X[so[u]] = sv[u];
    }
    // Round 6: the reference clone is the NEWEST one (the clone of this frame).  Any reference is exact algebra (the differences
    // d_c = u_c - u_ref span the same space); the largest-information rule of rounds 3-5 cost three dependent loads per lane, a
    // barrier and a 176-term serial loop of one thread in front of every solve (~6 k of the set-up's 35 k cycles) for no measurable
    // change of the result (tests/test_gpu_pinning.py::test_sigma_and_prior_scale_sweep passes with either).
    if (!SOLVE_REF_BY_INFO && RED && tid == 0) sRefSlot = C - 1;
    if (SOLVE_REF_BY_INFO && RED && tid == 0) {
        int best = 0; double tb = -1.0;
        for (int c = 0; c < C; ++c) {
            double t = 0.0;
            for (int g = 0; g < 16; ++g) t += sTr[16 * c + g];
            if (t > tb) { tb = t; best = c; }
        }
        sRefSlot = best;
    }
    const bool fused = marg_idx && marg_idx[bl] >= 0;
    const int contig = __syncthreads_and(tid >= ncolF || sColF[tid < NPF ? tid : 0] == sColF[0] + tid);
    const bool zero_copy = fused && contig && sColF[0] + MPF <= ld;
    const int ref6 = RED ? 6 * sRefSlot : (1 << 20), ncol = RED ? ncolF - 6 : ncolF;
    for (int c = tid; c < NP; c += NTH) {
        const int cf = c < ncol ? c + (c >= ref6 ? 6 : 0) : 0;
        sCol[c] = sColF[cf];
        if (RED) sRef[c] = sColF[ref6 + cf % 6];
    }
    __syncthreads();
    int bad = 0;
    dbg_stamp(0);
    SolveArgs<NC> sa{ P, sCol, X, Y, pan, sD1inv, sD2inv, Apart, chunk_used, G, rstride, bl, ncol, ld, lane, noise_all[bl], sRef, ncolF, ref6 };
    switch (wave) {                                           // wave-uniform: every wave runs the code specialised for its tile rows
    case 0: solve_wave<NC, 0, RED>(sa, &bad); break;
    case 1: solve_wave<NC, 1, RED>(sa, &bad); break;
    case 2: solve_wave<NC, 2, RED>(sa, &bad); break;
    case 3: solve_wave<NC, 3, RED>(sa, &bad); break;
    case 4: solve_wave<NC, 4, RED>(sa, &bad); break;
    case 5: solve_wave<NC, 5, RED>(sa, &bad); break;
    case 6: solve_wave<NC, 6, RED>(sa, &bad); break;
    default: solve_wave<NC, 7, RED>(sa, &bad); break;
    }
    dbg_stamp(5);
    if (bad) sBad = 1;
    // ---- [M | t] = R2' (R1' D2^-1)^T : tile (i, j), j over the NR1 row tiles of R1' (column NC carries t) ----
    double* Mg = Mall + (size_t)bl * mstride;
    constexpr int NQ = (NT * NR1 + NW - 1) / NW;                     // tiles per wave
    double4_f accs[NQ];
#pragma unroll
    for (int u = 0; u < NQ; ++u) {
        const int q = wave + NW * u;
        double4_f acc = { 0.0, 0.0, 0.0, 0.0 };
        if (q < NT * NR1) {
            const int i = q / NR1, j = q % NR1;
            for (int kt = i; kt < NT; ++kt) {
                double af[4], bf[4];
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    af[s] = Y[(16 * i + l15) * LDM + 16 * kt + 4 * s + kq];
                    bf[s] = X[(16 * j + l15) * LDM + 16 * kt + 4 * s + kq];          // B[k'][j'] = (R1' D2^-1)[j][k]
                }
#pragma unroll
                for (int s = 0; s < 4; ++s) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(af[s], bf[s], acc, 0, 0, 0);
            }
            const int col = 16 * j + l15;
            if (RED) {                                            // scattered into the full layout; the reference block follows below
                const int colF = col + (col >= ref6 ? 6 : 0);
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + kq + 4 * r, rowF = row + (row >= ref6 ? 6 : 0);
                    if (row < NC && col < NC) Mg[(size_t)rowF * MPF + colF] = acc[r];
                    if (row < NC && col == NC) Mg[(size_t)MPF * MPF + rowF] = acc[r];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + kq + 4 * r;
                    const double v = (row < NC) ? acc[r] : 0.0;
                    if (row < MPF && col < MPF) Mg[(size_t)row * MPF + col] = col < NC ? v : 0.0;
                    if (col == NC && row < MPF) Mg[(size_t)MPF * MPF + row] = v;
                }
            }
        }
        accs[u] = acc;
    }
    if (RED) {
        // The reference clone's block row and column: minus the sums over the other clones' blocks, component by component -
        // taken from a copy of [Mr | tr] in LDS (X is free once every wave has its tiles; reading the sums back from the global
        // M just written cost two store -> load round trips behind workgroup barriers: 38 k of the kernel's 150 k cycles).
        lds_barrier();
#pragma unroll
        for (int u = 0; u < NQ; ++u) {
            const int q = wave + NW * u;
            if (q < NT * NR1) {
                const int i = q / NR1, j = q % NR1, col = 16 * j + l15;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const int row = 16 * i + kq + 4 * r;
                    if (col < LDM) X[row * LDM + col] = accs[u][r];
                }
            }
        }
        lds_barrier();
        for (int e = tid; e < 2 * 6 * (MPF + 1); e += NTH) {
            const int side = e / (6 * (MPF + 1)), q = e - side * 6 * (MPF + 1), k = q / (MPF + 1), J = q - k * (MPF + 1);
            const bool inref = J >= ref6 && J < ref6 + 6;
            if (J == MPF) {                                   // t (row task only)
                if (side == 0) {
                    double s = 0.0;
                    for (int cr = 0; cr < CF - 1; ++cr) s += X[(6 * cr + k) * LDM + NC];
                    Mg[(size_t)MPF * MPF + ref6 + k] = -s;
                }
            } else if (!inref) {
                double s = 0.0;
                if (J < NCF) {
                    const int Jr = J < ref6 ? J : J - 6;
                    for (int cr = 0; cr < CF - 1; ++cr) s += side == 0 ? X[(6 * cr + k) * LDM + Jr] : X[Jr * LDM + 6 * cr + k];
                }
                if (side == 0) Mg[(size_t)(ref6 + k) * MPF + J] = -s; else Mg[(size_t)J * MPF + ref6 + k] = -s;
                if (side == 0 && J < NCF) Y[k * LDM + (J < ref6 ? J : J - 6)] = s;      // the column sums of block row k, for the corner below (Y is dead)
            }
        }
        if (MPF > NCF) {                                      // zero padding of the other rows / columns
            for (int e = tid; e < NCF * (MPF - NCF); e += NTH) {
                const int I = e / (MPF - NCF), J = NCF + e % (MPF - NCF);
                if (I < ref6 || I >= ref6 + 6) { Mg[(size_t)I * MPF + J] = 0.0; Mg[(size_t)J * MPF + I] = 0.0; }
            }
            for (int e = tid; e < (MPF - NCF) * (MPF - NCF); e += NTH)
                Mg[(size_t)(NCF + e / (MPF - NCF)) * MPF + NCF + e % (MPF - NCF)] = 0.0;
            if (tid < MPF - NCF) Mg[(size_t)MPF * MPF + NCF + tid] = 0.0;
        }
        lds_barrier();
        if (tid < 36) {                                       // corner: + the sum over all the other clones' blocks = the block row's column sums added over
            const int k = tid / 6, l = tid - 6 * k;           // the clones (round 6: as a double sum 36 lanes walked 100 LDS reads each at the very end of the kernel)
            double s = 0.0;
            for (int c2 = 0; c2 < CF - 1; ++c2) s += Y[k * LDM + 6 * c2 + l];
            Mg[(size_t)(ref6 + k) * MPF + ref6 + l] = s;
        }
    }
    dbg_stamp(6);
    if (sBad && tid == 0) atomicOr(&status[b], 4);
    // Pc = P[:, clone cols] for the in-place update (the apply kernel must read the PRE-update columns)
    double* Pc = Pcall + (size_t)bl * ystride;
    if (!zero_copy) {
        const int tx = tid & 63, ty = tid >> 6;
        for (int k = ty; k < MPF; k += NW) {
            const bool real = k < ncolF;
            const int gk = real ? fv.clone_idx[(size_t)b * fv.cmax + k / 6] + k % 6 : 0;
            for (int r = tx; r < n; r += 64) Pc[r + (size_t)k * ld] = real ? P[r + (size_t)gk * ld] : 0.0;
        }
    }
    if (tid == 0) { m_out[bl] = ncolF; nc_out[bl] = ncolF; pc_base_out[bl] = zero_copy ? fv.clone_idx[(size_t)b * fv.cmax] : -1; }
    dbg_stamp(7);
}

}  // namespace

// returns 0 when the window class is handled here (6 C <= 72), non-zero otherwise (caller falls back to k_info_update)
int launch_info_solve(const FactoredLaunch& L, hipStream_t st)
{
    const int ncm = 6 * ((L.c_used > 0 && L.c_used <= L.fv.cmax) ? L.c_used : L.fv.cmax);      // window class by the frames (launch_factored.h)
#define SOLVE_DISPATCH(NCF, RED)                                                                                              \
    {                                                                                                                         \
        const size_t sm = SolveCfg<(RED ? NCF - 6 : NCF)>::lds_bytes();                                                       \
        static bool attr_set = false;                                                                                         \
        if (!attr_set) { hipFuncSetAttribute((const void*)k_info_solve<NCF, RED>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)sm); attr_set = true; } \
        hipLaunchKernelGGL((k_info_solve<NCF, RED>), dim3(L.nb), dim3(64 * SolveCfg<NCF>::NW), sm, st, L.cv, L.fv, L.b0, L.Apart, L.chunk_used, L.G, L.rstride, \
                           L.noise, L.T, L.mstride, L.Pc, L.ystride, L.dx, L.m_out, L.nc_out, L.status, L.marg_idx, L.pc_base);       \
        return 0;                                                                                                             \
    }
    // The gauge-reduced form needs the block-Laplacian structure of A: the RemoveLost form of the Jacobians.  The Selected-timestamp
    // variants overwrite the anchor's translation columns (quirk Q10, SwMargUpdate.cpp:127,302), which breaks it; INGVIO_INFO_GAUGE=off
    // selects the unreduced solve for comparison.
#ifdef INGVIO_ALT_KERNELS
    static const bool no_gauge = [] { const char* e = getenv("INGVIO_INFO_GAUGE"); return e && !strcmp(e, "off"); }();
#else
    constexpr bool no_gauge = false;
#endif
    const bool red = !L.op.selected_variant && !no_gauge;
    if (ncm <= 36) { if (red) SOLVE_DISPATCH(36, true) else SOLVE_DISPATCH(36, false) }
    if (ncm <= 66) { if (red) SOLVE_DISPATCH(66, true) else SOLVE_DISPATCH(66, false) }
    // 12 clones (round 6): the same five tile rows as the 66 class - 66 reduced / 72 unreduced columns; 121 KB of LDS unreduced: one
    // workgroup per CU, which is what a single real-time filter in sliding-window mode needs (it used to take the Gauss-Jordan
    // route of the 16-clone class: 0.4 ms per update)
    if (ncm <= 72) { if (red) SOLVE_DISPATCH(72, true) else SOLVE_DISPATCH(72, false) }
#undef SOLVE_DISPATCH
    return 1;
}

int dbg_read_solve(long long* out, int n) { return dbg_read_local(out, n); }
