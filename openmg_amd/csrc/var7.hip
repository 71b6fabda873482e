// 7-point grid stencils with per-row coefficients: each half of the cycle over such a level as ONE launch
// (common.h Var7Plan).  Replaces, for the ordinary variable-coefficient input of openmg.mgSolve (openmg/__init__.py:28,
// operators.py:178), the eight level launches per V(1,1) cycle of the set-by-set schedule; arithmetic per row is the row
// kernels' (csr_kernels.hip): the row's fma chain in column order from +0 — -K, -J, -I, diagonal, +I, +J, +K; a neighbour
// outside the grid is a zero coefficient on a zero value —, x + (b - s) / a_ii (openmg/solvers.py:68), r = b - s, the
// restriction's chain over the eight children in R's column order (openmg/__init__.py:210), x + fma(w, e, 0) (:214).
//
// A workgroup owns a TX x TY tile of cells and a chunk of planes.  Step s of its march:
//   interval 1 (red rows):   sweep of plane s on tile + 2 (old black neighbours), residual of plane s - 3 on the tile
//   interval 2 (black rows): sweep of plane s - 1 on tile + 1 (new red neighbours) and its residual — the row's chain
//                            once more on the new value, as the fused sweep + residual launch of the set schedule forms it;
//                            the restriction of a finished plane pair; plane s + 2 (requested at the step's top) into LDS
// with one barrier after each.  The iterate's planes s - 4 .. s + 1 are LDS images updated in place (red at step p, black
// at step p + 1: the last reader of an old value is always earlier); residuals of four planes wait in LDS for their
// pair.  Ring cells are relaxed redundantly, so no workgroup waits for another; what a workgroup writes to HBM is its own
// tile.  Coefficients are never staged: every row loads its seven (unit stride across a wave) where it uses them.
#include <algorithm>
#include <cstring>
#include <thread>
#include <type_traits>
#include <vector>

#include "common.h"

namespace omg {
namespace {

constexpr int V7_HX = 4;      // cells of halo in x (three are needed; four keep the pairs of a line aligned)
constexpr int V7_HY = 3;
constexpr int V7_SLOTS = 6;   // planes of the iterate in LDS
constexpr int V7_RES = 4;     // planes of residuals in LDS

template <typename V>
struct Var7Args {
    int nx, ny, nz, hx;
    long long nh;                 // slots per colour
    const V *cD, *cM0, *cM1, *cM2, *cP0, *cP1, *cP2;
    const V *x_old;
    V *x_new;
    const V *b;
    const int32_t *cmap;
    V *cb;
    const V *e;
    V w;
    double *partials;
    int ntx, nty, lz;
    int sweep, x_zero;
    int dbg;                      // (timing experiments: bit 0 no coefficient loads, 1 no arithmetic, 2 no plane loads, 3 no stores)
};

__device__ __forceinline__ double v7_madd(double v, double x, double acc) { return fma(v, x, acc); }
__device__ __forceinline__ float v7_madd(float v, float x, float acc) { return fmaf(v, x, acc); }

template <typename V, int TX, int TY, bool DOWN, bool SYM, int NT>
__global__ __launch_bounds__(NT) void var7_pass_kernel(const Var7Args<V> a) {
    constexpr int RX = TX + 2 * V7_HX, RY = TY + 2 * V7_HY;
    constexpr int PLANE = RX * RY;
    // rows per role (below) and the split of the workgroup: the first NC threads relax, the waves behind them move the
    // iterate between HBM and LDS
    constexpr int WA = TX + 4, HA = TY + 4, WB = TX + 2, HB = TY + 2;
    constexpr int ROWS_A = (WA / 2) * HA, ROWS_B = (WB / 2) * HB, ROWS_C = (TX / 2) * TY;
    constexpr int NC = (ROWS_A + 63) / 64 * 64;
    constexpr int NIO = NT - NC;
    static_assert(ROWS_B <= NC && ROWS_C <= NC && NIO >= 64, "one row per relaxing thread and role; at least one wave for the copies");
    constexpr int PER = (PLANE + NIO - 1) / NIO;               // cells of a plane image per copying thread
    constexpr int PER_T = (TX * TY + NIO - 1) / NIO;           // cells of the tile per copying thread
    extern __shared__ unsigned char v7_raw[];
    V *const X = reinterpret_cast<V *>(v7_raw);                // [V7_SLOTS][RY][RX]
    V *const RS = X + V7_SLOTS * PLANE;                        // [V7_RES][TY][TX] (down); up: the coarse correction under the image
    constexpr int ECX = RX / 2, ECY = RY / 2 + 1, ECN = ECX * ECY;   // coarse cells under an image (j0 is odd: one more line)
    V *const EC = RS;                                          // [2][ECY][ECX]: two coarse planes
    __shared__ double s_red[NT / 64];

    // workgroup b runs on XCD b % 8: consecutive tiles (one z chunk's neighbours in x and y) share that XCD's L2
    const int nwg = int(gridDim.x);
    int wg = int(blockIdx.x);
    if ((nwg & 7) == 0) wg = (wg & 7) * (nwg >> 3) + (wg >> 3);
    const int tix = wg % a.ntx, tiy = (wg / a.ntx) % a.nty, tiz = wg / (a.ntx * a.nty);
    const int i0 = tix * TX - V7_HX, j0 = tiy * TY - V7_HY;     // the image's cell (0, 0)
    const int z0 = tiz * a.lz, z1 = min(a.nz, z0 + a.lz);
    const int tid = int(threadIdx.x);
    const int hy = a.ny >> 1;
    // Everything below indexes with 32 bits (build() refuses levels of 2^31 unknowns).  A cell's slot in the colour layout:
    // colour * nh + k * PS + j * hx + (i >> 1); a role's colour is fixed, and so is a thread's (j, i >> 1) in each role —
    // only the plane moves, and with it which cell of the thread's pair (i even / odd) has the role's colour.
    const int nh = int(a.nh), PS = a.ny * a.hx;
    auto img = [&](int k) -> V * { return X + ((k + 2 * V7_SLOTS) % V7_SLOTS) * PLANE; };
    auto res = [&](int k) -> V * { return RS + ((k + 2 * V7_RES) % V7_RES) * (TX * TY); };
    double sq = 0.0;

    if (tid >= NC) {
        // ================= the copying waves: HBM -> images (two planes ahead), finished planes -> HBM, restriction =========
        const int t = tid - NC;
        int f_static[PER], f_coarse[PER], f_par[PER];
        bool f_on[PER];
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = t + u * NIO;
            const int ly = c / RX, lx = c - ly * RX;
            const int i = i0 + lx, j = j0 + ly;
            f_on[u] = c < PLANE && i >= 0 && i < a.nx && j >= 0 && j < a.ny;
            f_static[u] = j * a.hx + (i >> 1);
            f_coarse[u] = (j >> 1) * a.hx + (i >> 1);
            f_par[u] = (i + j) & 1;
        }
        int g_img[PER_T], g_static[PER_T], g_par[PER_T];       // the tile's own cells: image index, slot part, parity
        bool g_on[PER_T];
#pragma unroll
        for (int u = 0; u < PER_T; ++u) {
            const int c = t + u * NIO;
            const int ty = c / TX, tx = c - ty * TX;
            const int i = i0 + V7_HX + tx, j = j0 + V7_HY + ty;
            g_on[u] = c < TX * TY && i < a.nx && j < a.ny;
            g_img[u] = (ty + V7_HY) * RX + tx + V7_HX;
            g_static[u] = j * a.hx + (i >> 1);
            g_par[u] = (i + j) & 1;
        }
        // up: the coarse correction e (openmg/__init__.py:214) under the image, one coarse plane at a time in LDS — a fine
        // cell of either plane of the pair takes its aggregate's value from there (fetched per fine cell it was two dependent
        // gathers per cell and plane: 240 us of the 256^3 pass)
        constexpr int PER_E = (ECN + NIO - 1) / NIO;
        const int ci0 = i0 >> 1, cj0 = j0 >> 1;                  // coarse cell under the image's cell (0, 0)
        auto fetch_coarse = [&](int K, V (&v)[PER_E]) {
#pragma unroll
            for (int u = 0; u < PER_E; ++u) {
                const int c = t + u * NIO;
                const int cy = c / ECX, cx = c - cy * ECX;
                const int I = ci0 + cx, J = cj0 + cy;
                V val = V(0);
                if (c < ECN && K >= 0 && K < (a.nz >> 1) && I >= 0 && I < a.hx && J >= 0 && J < hy) {
                    const int ci = (K * hy + J) * a.hx + I;
                    val = a.e[a.cmap ? a.cmap[ci] : ci];
                }
                v[u] = val;
            }
        };
        auto store_coarse = [&](int K, const V (&v)[PER_E]) {
            V *const dst = EC + (K & 1) * ECN;
#pragma unroll
            for (int u = 0; u < PER_E; ++u) {
                const int c = t + u * NIO;
                if (c < ECN) dst[c] = v[u];
            }
        };
        int f_ec[PER];                                           // a fine cell's aggregate in the coarse image
#pragma unroll
        for (int u = 0; u < PER; ++u) {
            const int c = t + u * NIO;
            const int ly = c / RX, lx = c - ly * RX;
            f_ec[u] = (((j0 + ly) >> 1) - cj0) * ECX + (((i0 + lx) >> 1) - ci0);
        }
        // one plane of the iterate as the pass starts from it: x_old, zero outside the grid; up: + R^T e where it lands in LDS
        auto fetch_plane = [&](int k, V (&v)[PER]) {
            const bool k_on = k >= 0 && k < a.nz;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                V val = V(0);
                if (k_on && f_on[u] && !(a.dbg & 4)) {
                    const int p = ((f_par[u] ^ k) & 1) * nh + k * PS + f_static[u];
                    val = (DOWN && a.x_zero) ? V(0) : a.x_old[p];
                }
                v[u] = val;
            }
        };
        auto store_plane = [&](int k, const V (&v)[PER]) {
            V *const dst = img(k);
            const V *const ec = EC + ((k >> 1) & 1) * ECN;
            const bool k_on = k >= 0 && k < a.nz;
#pragma unroll
            for (int u = 0; u < PER; ++u) {
                const int c = t + u * NIO;
                if (c < PLANE) {
                    V val = v[u];
                    if (!DOWN && k_on && f_on[u]) val = val + v7_madd(a.w, ec[f_ec[u]], V(0));   // :214 as the row kernels' y += R^T e
                    dst[c] = val;
                }
            }
        };
        if (!DOWN) {
            // the coarse planes under the prologue's three fine planes, then the one the loop starts with
            V ce[PER_E];
            fetch_coarse((z0 - 3) >> 1, ce);
            store_coarse((z0 - 3) >> 1, ce);
            fetch_coarse((z0 - 1) >> 1, ce);
            store_coarse((z0 - 1) >> 1, ce);
        }
        if (!DOWN) __syncthreads();
        {
            V v[PER];
            for (int k = z0 - 3; k <= z0 - 1; ++k) {
                fetch_plane(k, v);
                store_plane(k, v);
            }
        }
        if (!DOWN) {
            __syncthreads();                                     // (every reader of the prologue's coarse planes is done)
            V ce[PER_E];
            fetch_coarse(z0 >> 1, ce);
            store_coarse(z0 >> 1, ce);
        }
        __syncthreads();
        const bool write_back = DOWN ? a.sweep != 0 : true;      // (a down pass without its sweep leaves x_new alone)
        for (int s = z0 - 2; s <= z1 + 2; ++s) {
            V nxt[PER];
            fetch_plane(s + 2, nxt);                             // (in flight across interval 1)
            V ce[PER_E];
            const bool new_coarse = !DOWN && ((s + 3) & 1) == 0; // the NEXT step's plane s + 3 is the first of its pair
            if (new_coarse) fetch_coarse((s + 3) >> 1, ce);
            __syncthreads();
            // (first the loads' landing, then every store of the step: a wave's one counter covers both kinds, and the wait
            // for the loads would otherwise sit behind stores that have just been issued)
            store_plane(s + 2, nxt);                             // (over plane s - 4: its last readers were in interval 1)
            if (new_coarse) store_coarse((s + 3) >> 1, ce);      // (over coarse plane (s - 1) / 2: its last reader stored plane s)
            // plane s - 2 is final (its black rows were relaxed in the step before): the tile's cells of it leave
            if (write_back && !(a.dbg & 8) && s - 2 >= z0 && s - 2 < z1) {
                const int k = s - 2;
                const V *const src = img(k);
#pragma unroll
                for (int u = 0; u < PER_T; ++u)
                    if (g_on[u]) a.x_new[((g_par[u] ^ k) & 1) * nh + k * PS + g_static[u]] = src[g_img[u]];
            }
            if (DOWN) {
                // plane pair (q - 1, q), q = s - 3 odd: its last residuals (red of plane q) were formed in interval 1
                const int q = s - 3;
                if ((q & 1) && q >= z0 && q < z1) {
                    const V *const r0 = res(q - 1), *const r1 = res(q);
                    const int K = q >> 1;
                    for (int c = t; c < (TX / 2) * (TY / 2); c += NIO) {
                        const int cy = c / (TX / 2), cx = c - cy * (TX / 2);
                        const int I = (tix * TX >> 1) + cx, J = (tiy * TY >> 1) + cy;
                        if (I >= a.hx || J >= hy) continue;
                        const int o = (2 * cy) * TX + 2 * cx;
                        V acc = v7_madd(a.w, r0[o], V(0));                               // R's columns ascend: i fastest, then j, then k
                        acc = v7_madd(a.w, r0[o + 1], acc);
                        acc = v7_madd(a.w, r0[o + TX], acc);
                        acc = v7_madd(a.w, r0[o + TX + 1], acc);
                        acc = v7_madd(a.w, r1[o], acc);
                        acc = v7_madd(a.w, r1[o + 1], acc);
                        acc = v7_madd(a.w, r1[o + TX], acc);
                        acc = v7_madd(a.w, r1[o + TX + 1], acc);
                        const int ci = (K * hy + J) * a.hx + I;
                        a.cb[a.cmap ? a.cmap[ci] : ci] = acc;
                    }
                }
            }
            __syncthreads();
        }
    } else {
        // ================= the relaxing waves: loads of coefficients only, no stores to HBM ==================================
        // (a wave's vector-memory counter counts loads and stores alike, and only loads return in order: with no store in
        // flight the wait for one role's coefficients is a COUNTED wait that leaves the next role's requests in flight)
        // the row's chain (column order), on the images; own: the value standing for x_i
        struct Row { V aMK, aMJ, aMI, aD, aPI, aPJ, aPK, b; };
        auto chain = [&](const Row &r, const V *lo, const V *mid, const V *hi, int c, V own) -> V {
            if (a.dbg & 2) return r.aD * own;
            V s = v7_madd(r.aMK, lo[c], V(0));
            s = v7_madd(r.aMJ, mid[c - RX], s);
            s = v7_madd(r.aMI, mid[c - 1], s);
            s = v7_madd(r.aD, own, s);
            s = v7_madd(r.aPI, mid[c + 1], s);
            s = v7_madd(r.aPJ, mid[c + RX], s);
            s = v7_madd(r.aPK, hi[c], s);
            return s;
        };
        // A thread has at most ONE row in each of the step's three roles — A: red sweep of plane s on tile + 2, C: red
        // residual of plane s - 3 on the tile, B: black sweep + residual of plane s - 1 on tile + 1 — and requests a role's
        // coefficients one interval before it uses them (B's during interval 1, the next step's A and C during interval 2).
        // What of a role's row does not move with the plane: the thread's line and pair of the image rectangle
        // [lx0, lx0 + W) x [ly0, ly0 + H) (W even: every line holds W / 2 rows of a colour)
        struct Role {
            bool on;          // the thread has a row, its line is inside the grid
            bool own;         // ... inside the tile's lines (and, loosely, columns: refined per cell)
            int cE, iE, j;    // image index and grid x of the pair's first cell; grid y
            int lxE, ly;      // ... its image coordinates
            int par0;         // the row's cell is iE + (par0 ^ (k & 1))
            int base;         // j * hx + (first cell's i >> 1)
            int odd;          // iE & 1: the two cells of the pair lie in different slots
        };
        auto make_role = [&](int col, int lx0, int ly0, int W, int H) -> Role {
            Role q;
            const int half = W >> 1;
            const int ly = ly0 + tid / half, m = tid - (tid / half) * half;
            const int lxE = lx0 + 2 * m;
            q.j = j0 + ly;
            q.lxE = lxE;
            q.ly = ly;
            q.iE = i0 + lxE;
            q.cE = ly * RX + lxE;
            q.par0 = (col ^ (q.iE + q.j)) & 1;
            q.odd = q.iE & 1;
            q.base = q.j * a.hx + ((q.iE - q.odd) >> 1);
            q.on = tid < half * H && q.j >= 0 && q.j < a.ny;
            q.own = ly >= V7_HY && ly < V7_HY + TY;
            return q;
        };
        const Role RA = make_role(0, 2, 1, WA, HA), RC = make_role(0, V7_HX, V7_HY, TX, TY), RB = make_role(1, 3, 2, WB, HB);
        struct At { bool on; int c, p, lx, t, i; };    // this plane's cell of the role: image index, slot, image x, index in the tile, grid x
        auto at = [&](const Role &q, int col, int k, bool plane_on) -> At {
            At w;
            const int par = (q.par0 ^ k) & 1;
            w.i = q.iE + par;
            w.on = plane_on && q.on && w.i >= 0 && w.i < a.nx;
            w.c = q.cE + par;
            w.lx = q.lxE + par;
            w.t = (q.ly - V7_HY) * TX + (w.lx - V7_HX);
            w.p = col * nh + k * PS + q.base + (q.odd & par);
            return w;
        };
        // every load unconditional (a row that does not exist reads slot 0 and is never used; a coupling that does not exist
        // reads the row's own slot and becomes zero): the same number of requests on every path
        auto request = [&](const Role &q, const At &w, int col, int k, Row &r) {
            if (a.dbg & 1) { r.aD = V(6); r.aPI = r.aPJ = r.aPK = r.aMI = r.aMJ = r.aMK = V(-1); r.b = V(1); return; }
            const int p = w.on ? w.p : 0;
            r.aD = a.cD[p];
            r.aPI = a.cP0[p];
            r.aPJ = a.cP1[p];
            r.aPK = a.cP2[p];
            if (SYM) {
                const int o = p + (1 - 2 * col) * nh;          // the other colour's slot of the same pair position
                const bool hasI = w.on && w.i > 0, hasJ = w.on && q.j > 0, hasK = w.on && k > 0;
                const V vI = a.cP0[hasI ? o - ((w.i & 1) ^ 1) : p];
                const V vJ = a.cP1[hasJ ? o - a.hx : p];
                const V vK = a.cP2[hasK ? o - PS : p];
                r.aMI = hasI ? vI : V(0);
                r.aMJ = hasJ ? vJ : V(0);
                r.aMK = hasK ? vK : V(0);
            } else {
                r.aMI = a.cM0[p];
                r.aMJ = a.cM1[p];
                r.aMK = a.cM2[p];
            }
            r.b = a.b[p];
        };
        auto planeA = [&](int sA) { return a.sweep && sA >= 0 && sA < a.nz && sA <= z1 + 1; };
        auto planeC = [&](int q) { return q >= z0 && q < z1; };
        auto planeB = [&](int q) { return q >= 0 && q < a.nz && q >= z0 - 1 && q <= z1; };

        // With its sweep a pass forms the red residual of plane q (role C, step q + 3) from the coefficients role A loaded for
        // that row at step q — the same cell —, kept in registers for the three steps between (only b is read again): a
        // quarter of the pass's coefficient traffic.  Role C then takes A's rows (those inside the tile).
        const bool carry = a.sweep != 0;
        const Role &RCx = carry ? RA : RC;
        const bool c_own_line = RCx.ly >= V7_HY && RCx.ly < V7_HY + TY;
        auto atC = [&](int k) -> At {
            At w = at(RCx, 0, k, planeC(k));
            w.on = w.on && c_own_line && w.lx >= V7_HX && w.lx < V7_HX + TX;
            return w;
        };
        Row rA, rB, rC, k1, k2, k3;                              // k1 .. k3: role A's rows of the last three steps
        At wA = at(RA, 0, z0 - 2, planeA(z0 - 2)), wC = atC(z0 - 5);
        request(RA, wA, 0, z0 - 2, rA);
        request(RCx, wC, 0, z0 - 5, rC);
        k1 = k2 = k3 = rC;
        if (!DOWN) { __syncthreads(); __syncthreads(); }         // (the copying waves' two barriers around the prologue's coarse planes)
        __syncthreads();
        for (int s = z0 - 2; s <= z1 + 2; ++s) {
            const At wB = at(RB, 1, s - 1, planeB(s - 1));
            request(RB, wB, 1, s - 1, rB);                       // (in flight across interval 1)
            // ---- interval 1: red rows ----
            if (wA.on) {
                V *const mid = img(s);
                const V *const lo = img(s - 1), *const hi = img(s + 1);
                const V xv = mid[wA.c];
                mid[wA.c] = xv + (rA.b - chain(rA, lo, mid, hi, wA.c, xv)) / rA.aD;    // openmg/solvers.py:68
            }
            if (wC.on) {
                const int k = s - 3;
                const V *const mid = img(k), *const lo = img(k - 1), *const hi = img(k + 1);
                Row r = carry ? k3 : rC;
                r.b = rC.b;
                const V rv = r.b - chain(r, lo, mid, hi, wC.c, mid[wC.c]);               // :209 / :227
                if (DOWN) res(k)[wC.t] = rv;
                else sq = fma(double(rv), double(rv), sq);
            }
            k3 = k2; k2 = k1; k1 = rA;
            __syncthreads();
            wA = at(RA, 0, s + 1, planeA(s + 1));
            wC = atC(s - 2);
            request(RA, wA, 0, s + 1, rA);                       // (in flight across interval 2)
            if (carry) rC.b = a.b[wC.on ? wC.p : 0];
            else request(RCx, wC, 0, s - 2, rC);
            // ---- interval 2: black rows of plane s - 1 ----
            if (wB.on) {
                const int k = s - 1;
                V *const mid = img(k);
                const V *const lo = img(k - 1), *const hi = img(k + 1);
                V xv = mid[wB.c];
                if (a.sweep) {
                    xv = xv + (rB.b - chain(rB, lo, mid, hi, wB.c, xv)) / rB.aD;
                    mid[wB.c] = xv;
                }
                if (k >= z0 && k < z1 && RB.own && wB.lx >= V7_HX && wB.lx < V7_HX + TX) {
                    const V rv = rB.b - chain(rB, lo, mid, hi, wB.c, xv);               // the same chain on the updated vector
                    if (DOWN) res(k)[wB.t] = rv;
                    else sq = fma(double(rv), double(rv), sq);
                }
            }
            __syncthreads();
        }
    }
    if (!DOWN && a.partials) {
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if ((tid & 63) == 0) s_red[tid >> 6] = sq;
        __syncthreads();
        if (tid == 0) {
            double t = 0.0;
            for (int wv = 0; wv < NT / 64; ++wv) t += s_red[wv];
            a.partials[blockIdx.x] = t;
        }
    }
}

template <typename V, int TX, int TY, int NT>
void launch_pass(const Var7Args<V> &a, bool down, bool sym, int64_t n_wg, hipStream_t s) {
    constexpr size_t lds = (size_t(V7_SLOTS) * (TX + 2 * V7_HX) * (TY + 2 * V7_HY) + size_t(V7_RES) * TX * TY) * sizeof(V);
    auto go = [&](auto kernel) {
        static bool once = false;         // (per instantiation)
        if (!once && lds > size_t(48) << 10) {
            OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
            once = true;
        }
        hipLaunchKernelGGL(kernel, dim3(unsigned(n_wg)), dim3(NT), lds, s, a);
    };
    if (down && sym) go(var7_pass_kernel<V, TX, TY, true, true, NT>);
    else if (down) go(var7_pass_kernel<V, TX, TY, true, false, NT>);
    else if (sym) go(var7_pass_kernel<V, TX, TY, false, true, NT>);
    else go(var7_pass_kernel<V, TX, TY, false, false, NT>);
    OMG_HIP(hipGetLastError());
}

}  // namespace

namespace {
// every row of a natural-numbered device CSR against the 7-point pattern of its grid position; its coefficients into the
// seven colour-layout arrays (a neighbour outside the grid: an explicit zero); the first offending row into *err
template <typename V>
__global__ void var7_scatter_kernel(const int32_t *indptr, const int32_t *indices, const double *data, int nx, int ny, int nz,
                                    V *cD, V *cM0, V *cM1, V *cM2, V *cP0, V *cP1, V *cP2, unsigned long long *err) {
    const long long n = (long long)nx * ny * nz;
    const long long r = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (r >= n) return;
    const long long sj = nx, sk = (long long)nx * ny;
    const int i = int(r % nx), j = int((r / nx) % ny), k = int(r / sk);
    const long long sl = (long long)((i + j + k) & 1) * (n / 2) + r / 2;
    const long long want[7] = {k > 0 ? r - sk : -1, j > 0 ? r - sj : -1, i > 0 ? r - 1 : -1, r,
                               i + 1 < nx ? r + 1 : -1, j + 1 < ny ? r + sj : -1, k + 1 < nz ? r + sk : -1};
    V *const out[7] = {cM2, cM1, cM0, cD, cP0, cP1, cP2};
    int p = indptr[r];
    const int pe = indptr[r + 1];
    bool good = true;
#pragma unroll
    for (int e = 0; e < 7; ++e) {
        V v = V(0);
        if (want[e] >= 0) {
            if (p < pe && (long long)indices[p] == want[e]) {
                v = V(data[p]);
                ++p;
                if (!isfinite(v) || (e == 3 && v == V(0))) good = false;
            } else {
                good = false;
            }
        }
        out[e][sl] = v;
    }
    if (p != pe) good = false;
    if (!good) atomicMin(err, (unsigned long long)(r + 1));
}

template <typename V>
void var7_tiling(Var7Plan<V> &P) {
    // tiles: 64 x 16 on wide grids, narrower below
    P.tx = P.nx >= 64 ? 64 : P.nx >= 32 ? 32 : 16;
    if (P.tx == 64 && int64_t((P.nx + 63) / 64) * ((P.ny + 15) / 16) * ((P.nz + 15) / 16) < 256) P.tx = 32;    // (a launch of fewer workgroups than compute units)
    if (const char *e = experiment_env("OMG_VAR7_TX")) { const int v = atoi(e); if (v == 16 || v == 32 || v == 64) P.tx = v; }
    P.ty = 16;
    P.threads = P.tx == 64 ? 1024 : P.tx == 32 ? 512 : 320;
    P.ntx = (P.nx + P.tx - 1) / P.tx;
    P.nty = (P.ny + P.ty - 1) / P.ty;
    // chunks of planes: ONE workgroup per compute unit (its images fill the unit's LDS) — 256^3 fp64: 64 tiles x 4 chunks of
    // 64 planes 0.94 ms per cycle, 8 chunks of 34 (two rounds of workgroups, 15 % more ring planes) 1.02, 2 chunks 1.46
    {
        const int64_t tiles = int64_t(P.ntx) * P.nty;
        const int64_t want = std::max<int64_t>(1, 256 / tiles);
        P.lz = int((P.nz + want - 1) / want);
        P.lz = std::max(8, (P.lz + 1) & ~1);
    }
    if (const char *e = experiment_env("OMG_VAR7_LZ")) P.lz = std::max(2, atoi(e) & ~1);
    P.ntz = (P.nz + P.lz - 1) / P.lz;
    P.n_wg = int64_t(P.ntx) * P.nty * P.ntz;
}

int64_t var7_min_rows() {
    // (below 128^3 the set-by-set schedule's launches are short and a pass of a few workgroups is not: measured at 64^3 and
    // 32^3, profiles/r06_var7.txt.  OMG_VAR7_MIN: tests put small levels through the passes)
    int64_t n_min = int64_t(1) << 21;
    if (const char *e = getenv("OMG_VAR7_MIN")) n_min = std::max<int64_t>(4096, atoll(e));
    return n_min;
}
}  // namespace

template <typename V>
bool Var7Plan<V>::build_device(const DevCsrPlain &A, int gx, int gy, int gz, double wv, Ordering &ord, hipStream_t s) {
    const int64_t n = A.n_rows;
    if (n < var7_min_rows() || A.n_cols != n || int64_t(gx) * gy * gz != n) return false;
    if (gx < 4 || gy < 4 || gz < 4 || (gx & 1) || (gy & 1) || (gz & 1) || gx > (1 << 14) || gy > (1 << 14) || gz > (1 << 14)) return false;
    if (n >= (int64_t(1) << 31)) return false;
    cD.alloc(size_t(n));
    for (int d = 0; d < 3; ++d) { cM[d].alloc(size_t(n)); cP[d].alloc(size_t(n)); }
    DevBuf<unsigned long long> d_err(1);
    OMG_HIP(hipMemsetAsync(d_err.p, 0xFF, sizeof(unsigned long long), s));
    hipLaunchKernelGGL((var7_scatter_kernel<V>), dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, A.indptr.p, A.indices.p, A.data.p, gx, gy, gz,
                       cD.p, cM[0].p, cM[1].p, cM[2].p, cP[0].p, cP[1].p, cP[2].p, d_err.p);
    OMG_HIP(hipGetLastError());
    unsigned long long err = 0;
    OMG_HIP(hipMemcpyAsync(&err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    if (err != ~0ull) {
        cD.release();
        for (int d = 0; d < 3; ++d) { cM[d].release(); cP[d].release(); }
        return false;
    }
    nx = gx; ny = gy; nz = gz;
    w = wv;
    sym = false;
    var7_tiling(*this);
    partials.alloc(size_t(n_wg) + 64);
    partials.zero(s);
    OMG_HIP(hipStreamSynchronize(s));
    ord = Ordering();
    ord.identity = false;
    ord.sets = {0, n / 2, n};
    ord.closed_form = 2; ord.cf_nx = nx; ord.cf_ny = ny; ord.cf_nz = nz;     // perm / inv: on the device (fill_ordering_device), on the host on demand
    return true;
}

template <typename V>
bool Var7Plan<V>::build(const omg_csr &A, const omg_csr &R, Ordering &ord, hipStream_t s) {
    const int64_t n = A.n_rows;
    if (n < var7_min_rows() || A.n_cols != n || (n & 1)) return false;
    // grid extents from the first missing coupling (as the other plans read them)
    auto has = [&](int64_t r, int64_t col) {
        for (int32_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p)
            if (A.indices[p] == col) return true;
        return false;
    };
    int64_t gx = n;
    for (int64_t r = 1; r < n; ++r)
        if (!has(r, r - 1)) { gx = r; break; }
    if (gx < 4 || n % gx) return false;
    const int64_t lines = n / gx;
    int64_t gy = lines;
    for (int64_t q = 1; q < lines; ++q)
        if (!has(q * gx, (q - 1) * gx)) { gy = q; break; }
    if (gy < 4 || lines % gy) return false;
    const int64_t gz = lines / gy;
    if (gz < 4 || (gx & 1) || (gy & 1) || (gz & 1) || gx > (1 << 14) || gy > (1 << 14) || gz > (1 << 14)) return false;
    if (n >= (int64_t(1) << 31)) return false;
    const int64_t sj = gx, sk = gx * gy, nh_ = n / 2, hx_ = gx / 2;
    // the restriction: the plain 2 x 2 x 2 aggregation with one weight, columns ascending
    if (R.n_rows != n / 8 || R.n_cols != n || R.nnz != n) return false;
    const double wv = R.data[0];
    std::vector<char> ok_flag(64, 1);
    std::vector<V> hD((size_t)(n)), hM[3], hP[3];
    for (int d = 0; d < 3; ++d) { hM[d].assign(size_t(n), V(0)); hP[d].assign(size_t(n), V(0)); }
    auto slot = [&](int64_t i, int64_t j, int64_t k) { return ((i + j + k) & 1) * nh_ + (k * gy + j) * hx_ + (i >> 1); };
    {
        // rows in parallel; every chunk reports through its own flag
        const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
        const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(std::min<int64_t>(hw, 64), n / 65536));
        std::vector<std::thread> th;
        for (int64_t t = 0; t < nt; ++t) {
            th.emplace_back([&, t] {
                const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
                bool good = true;
                for (int64_t r = lo; r < hi && good; ++r) {
                    const int64_t i = r % gx, j = (r / gx) % gy, k = r / sk;
                    const int64_t sl = slot(i, j, k);
                    const int64_t want[7] = {k > 0 ? r - sk : -1, j > 0 ? r - sj : -1, i > 0 ? r - 1 : -1, r,
                                             i + 1 < gx ? r + 1 : -1, j + 1 < gy ? r + sj : -1, k + 1 < gz ? r + sk : -1};
                    int32_t p = A.indptr[r];
                    const int32_t pe = A.indptr[r + 1];
                    for (int sl7 = 0; sl7 < 7 && good; ++sl7) {
                        if (want[sl7] < 0) continue;
                        if (p >= pe || int64_t(A.indices[p]) != want[sl7]) { good = false; break; }
                        const double v = A.data[p];
                        const V vv = V(v);
                        if (!(std::isfinite(double(vv)))) { good = false; break; }
                        switch (sl7) {
                            case 0: hM[2][size_t(sl)] = vv; break;
                            case 1: hM[1][size_t(sl)] = vv; break;
                            case 2: hM[0][size_t(sl)] = vv; break;
                            case 3: hD[size_t(sl)] = vv; if (vv == V(0)) good = false; break;
                            case 4: hP[0][size_t(sl)] = vv; break;
                            case 5: hP[1][size_t(sl)] = vv; break;
                            default: hP[2][size_t(sl)] = vv; break;
                        }
                        ++p;
                    }
                    if (p != pe) good = false;
                }
                ok_flag[size_t(t)] = good;
            });
        }
        for (auto &x : th) x.join();
        for (int64_t t = 0; t < nt; ++t)
            if (!ok_flag[size_t(t)]) return false;
    }
    // the restriction's rows
    {
        const int64_t cx = gx / 2, cy = gy / 2;
        bool good = true;
        for (int64_t I = 0; I < n / 8 && good; ++I) {
            if (R.indptr[I + 1] - R.indptr[I] != 8) { good = false; break; }
            const int64_t ci = I % cx, cj = (I / cx) % cy, ck = I / (cx * cy);
            int32_t p = R.indptr[I];
            for (int dk = 0; dk < 2 && good; ++dk)
                for (int dj = 0; dj < 2 && good; ++dj)
                    for (int di = 0; di < 2; ++di, ++p)
                        if (int64_t(R.indices[p]) != ((2 * ck + dk) * gy + 2 * cj + dj) * gx + 2 * ci + di || R.data[p] != wv) { good = false; break; }
        }
        if (!good) return false;
    }
    // symmetric bit for bit?  (-I of cell i + 1 is +I of cell i, and so on)
    {
        bool symm = true;
        for (int64_t r = 0; r < n && symm; ++r) {
            const int64_t i = r % gx, j = (r / gx) % gy, k = r / sk;
            const int64_t sl = slot(i, j, k);
            if (i + 1 < gx && std::memcmp(&hP[0][size_t(sl)], &hM[0][size_t(slot(i + 1, j, k))], sizeof(V)) != 0) symm = false;
            if (j + 1 < gy && std::memcmp(&hP[1][size_t(sl)], &hM[1][size_t(slot(i, j + 1, k))], sizeof(V)) != 0) symm = false;
            if (k + 1 < gz && std::memcmp(&hP[2][size_t(sl)], &hM[2][size_t(slot(i, j, k + 1))], sizeof(V)) != 0) symm = false;
        }
        // Measured at 256^3 fp64 (profiles/r06_var7.txt): the four-array form is no faster than the seven-array one (0.940
        // against 0.925 ms per cycle) — a coupling's second reader comes an interval or two after its first, and by then an
        // XCD's 32 workgroups have pushed it out of the L2 — so the seven arrays are the default and OMG_VAR7_SYM=1 asks for
        // the symmetric form where the operator allows it.
        const char *e = getenv("OMG_VAR7_SYM");
        sym = symm && e && e[0] == '1';
    }
    nx = int(gx); ny = int(gy); nz = int(gz);
    w = wv;
    var7_tiling(*this);
    auto put = [&](DevBuf<V> &d, const std::vector<V> &h) {
        d.alloc(size_t(n));
        d.upload(h.data(), size_t(n), s);
    };
    put(cD, hD);
    for (int d = 0; d < 3; ++d) put(cP[d], hP[d]);
    if (!sym) for (int d = 0; d < 3; ++d) put(cM[d], hM[d]);
    partials.alloc(size_t(n_wg) + 64);
    partials.zero(s);
    OMG_HIP(hipStreamSynchronize(s));
    // the ordering: parity colours, red (even i + j + k) first, each colour in natural order — slot = colour n / 2 + row / 2
    ord = Ordering();
    ord.identity = false;
    ord.sets = {0, n / 2, n};
    ord.closed_form = 2; ord.cf_nx = nx; ord.cf_ny = ny; ord.cf_nz = nz;
    materialise_ordering(ord);
    return true;
}

template <typename V>
HostCsr Var7Plan<V>::operator_csr(hipStream_t s) const {
    const int64_t gx = nx, gy = ny, gz = nz, n = gx * gy * gz, nh_ = n / 2, hx_ = gx / 2, sj = gx, sk = gx * gy;
    std::vector<V> hD((size_t)(n)), hM[3], hP[3];
    cD.download(hD.data(), size_t(n), s);
    for (int d = 0; d < 3; ++d) {
        hP[d].resize(size_t(n));
        cP[d].download(hP[d].data(), size_t(n), s);
        if (!sym) { hM[d].resize(size_t(n)); cM[d].download(hM[d].data(), size_t(n), s); }
    }
    OMG_HIP(hipStreamSynchronize(s));
    auto slot = [&](int64_t i, int64_t j, int64_t k) { return ((i + j + k) & 1) * nh_ + (k * gy + j) * hx_ + (i >> 1); };
    HostCsr A;
    A.n_rows = A.n_cols = n;
    A.indptr.resize(size_t(n) + 1);
    A.indptr[0] = 0;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t i = r % gx, j = (r / gx) % gy, k = r / sk;
        A.indptr[size_t(r) + 1] = A.indptr[size_t(r)] + int32_t(1 + (i > 0) + (i + 1 < gx) + (j > 0) + (j + 1 < gy) + (k > 0) + (k + 1 < gz));
    }
    A.nnz = A.indptr[size_t(n)];
    A.indices.resize(size_t(A.nnz));
    A.data.resize(size_t(A.nnz));
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int64_t nt = std::max<int64_t>(1, std::min<int64_t>(hw, n / 65536));
    std::vector<std::thread> th;
    for (int64_t t = 0; t < nt; ++t)
        th.emplace_back([&, t] {
            for (int64_t r = n * t / nt; r < n * (t + 1) / nt; ++r) {
                const int64_t i = r % gx, j = (r / gx) % gy, k = r / sk;
                const size_t sl = size_t(slot(i, j, k));
                int32_t p = A.indptr[size_t(r)];
                auto put = [&](int64_t col, V v) { A.indices[size_t(p)] = int32_t(col); A.data[size_t(p)] = double(v); ++p; };
                if (k > 0) put(r - sk, sym ? hP[2][size_t(slot(i, j, k - 1))] : hM[2][sl]);
                if (j > 0) put(r - sj, sym ? hP[1][size_t(slot(i, j - 1, k))] : hM[1][sl]);
                if (i > 0) put(r - 1, sym ? hP[0][size_t(slot(i - 1, j, k))] : hM[0][sl]);
                put(r, hD[sl]);
                if (i + 1 < gx) put(r + 1, hP[0][sl]);
                if (j + 1 < gy) put(r + sj, hP[1][sl]);
                if (k + 1 < gz) put(r + sk, hP[2][sl]);
            }
        });
    for (auto &x : th) x.join();
    return A;
}

template <typename V>
HostCsr Var7Plan<V>::restriction_csr() const {
    const int64_t gx = nx, gy = ny, gz = nz, n = gx * gy * gz, sj = gx, sk = gx * gy;
    const int64_t nxc = gx / 2, nyc = gy / 2, nc = n / 8;
    HostCsr R;
    R.n_rows = nc;
    R.n_cols = n;
    R.nnz = n;
    R.indptr.resize(size_t(nc) + 1);
    R.indices.resize(size_t(n));
    R.data.resize(size_t(n));
    for (int64_t cr = 0; cr <= nc; ++cr) R.indptr[size_t(cr)] = int32_t(8 * cr);
    for (int64_t cr = 0; cr < nc; ++cr) {
        const int64_t I = cr % nxc, J = (cr / nxc) % nyc, K = cr / (nxc * nyc);
        int64_t p = 8 * cr;
        for (int dk = 0; dk < 2; ++dk)
            for (int dj = 0; dj < 2; ++dj)
                for (int di = 0; di < 2; ++di, ++p) {
                    R.indices[size_t(p)] = int32_t((2 * K + dk) * sk + (2 * J + dj) * sj + 2 * I + di);
                    R.data[size_t(p)] = w;
                }
    }
    (void)gz;
    return R;
}

namespace {
template <typename V>
void run(const Var7Plan<V> &P, Var7Args<V> a, bool down, hipStream_t s) {
    a.nx = P.nx; a.ny = P.ny; a.nz = P.nz; a.hx = P.nx / 2;
    a.nh = (long long)P.nx * P.ny * P.nz / 2;
    a.cD = P.cD.p;
    a.cP0 = P.cP[0].p; a.cP1 = P.cP[1].p; a.cP2 = P.cP[2].p;
    a.cM0 = P.cM[0].p; a.cM1 = P.cM[1].p; a.cM2 = P.cM[2].p;
    a.w = V(P.w);
    a.ntx = P.ntx; a.nty = P.nty; a.lz = P.lz;
    if (const char *e = experiment_env("OMG_VAR7_DBG")) a.dbg = atoi(e);      // (tools/var7_dbg.py; a build with -DOMG_EXPERIMENTS)
    if (P.tx == 64) launch_pass<V, 64, 16, 1024>(a, down, P.sym, P.n_wg, s);
    else if (P.tx == 32) launch_pass<V, 32, 16, 512>(a, down, P.sym, P.n_wg, s);
    else launch_pass<V, 16, 16, 320>(a, down, P.sym, P.n_wg, s);
}
}  // namespace

template <typename V>
void Var7Plan<V>::down(const V *x_old, V *x_new, const V *b, bool x_zero, const Coarse &c, hipStream_t s, bool sweep) const {
    Var7Args<V> a;
    std::memset(&a, 0, sizeof(a));
    a.x_old = x_old; a.x_new = x_new; a.b = b;
    a.cmap = c.map; a.cb = c.b;
    a.sweep = sweep ? 1 : 0;
    a.x_zero = x_zero ? 1 : 0;
    run(*this, a, true, s);
}

template <typename V>
void Var7Plan<V>::up(const V *x_old, V *x_new, const V *b, const Coarse &c, double *out, hipStream_t s, bool sweep) const {
    Var7Args<V> a;
    std::memset(&a, 0, sizeof(a));
    a.x_old = x_old; a.x_new = x_new; a.b = b;
    a.cmap = c.map; a.e = c.e;
    a.partials = out;
    a.sweep = sweep ? 1 : 0;
    run(*this, a, false, s);
}

template struct Var7Plan<double>;
template struct Var7Plan<float>;

}  // namespace omg
