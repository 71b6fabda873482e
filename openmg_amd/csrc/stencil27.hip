// 27-point grid stencils with per-row coefficients (common.h Stencil27Plan) — gfx950 only.
//
// BASELINE configs[4]: the Q1 stiffness operator of -div(kappa grad u) and its Galerkin products, fp32, 8-colour
// Gauss-Seidel.  The octant colouring (colour = (i & 1) + 2 (j & 1) + 4 (k & 1), what the greedy colouring finds)
// puts cell (i, j, k) of colour c at slot c na + a, a = ((k >> 1) hy + (j >> 1)) hx + (i >> 1): every colour's part
// of a level vector is a grid of AGGREGATES.  For a cell with parity bit p along an axis and a neighbour at offset
// d in {-1, 0, +1} along it, the neighbour's bit is p ^ (d != 0) and its aggregate index moves by (p + d) >> 1 — all
// compile-time facts of (colour, slot), so a row's 27 operands are 27 unit-stride vector loads.
//
// Thread mapping: a wave owns G = 64 / L whole grid LINES of one colour (L = lanes per line, a lane RG consecutive
// aggregates of its line: 16-byte accesses), so the only dependency inside a pair of colours (2m, 2m + 1) — the two
// in-line neighbours — is a neighbouring element of the same lane or the neighbouring LANE (one wave shuffle); no
// workgroup ever waits for another.  The coefficients are stored for exactly that mapping: [colour][wave][slot][lane][RG].
//
// Arithmetic: the row kernels' (csr_kernels.hip) for a row of 27 stored entries — four fma chains, entry e to chain
// e & 3, ((s0 + s1) + s2) + s3, x + (b - sum) / a_ii — on the operator PADDED with explicit zeros for the neighbours
// a boundary row lacks (common.h): same bits as the set-by-set schedule of the same hierarchy.
#include <algorithm>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <type_traits>

#include "common.h"

namespace omg {

bool is_plain_aggregation(const omg_csr &R, int64_t nx, int64_t ny, int64_t nz, double &w) {
    const int64_t n = nx * ny * nz, sj = nx, sk = nx * ny;
    const int64_t nxc = nx / 2, nyc = ny / 2, nzc = nz / 2;
    if ((nx & 1) || (ny & 1) || (nz & 1) || R.n_cols != n || R.n_rows != nxc * nyc * nzc || R.nnz != n || n == 0) return false;
    w = R.data[0];
    const double w0 = w;
    const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const int nt = (int)std::min<int64_t>(hw, std::max<int64_t>(1, R.n_rows / 8192));
    std::atomic<bool> ok(true);
    auto scan = [&](int tnum) {
        const int64_t clo = R.n_rows * tnum / nt, chi = R.n_rows * (tnum + 1) / nt;
        for (int64_t cr = clo; cr < chi; ++cr) {
            const int64_t I = cr % nxc, J = (cr / nxc) % nyc, K = cr / (nxc * nyc);
            int64_t p = R.indptr[cr];
            if (R.indptr[cr + 1] - p != 8) { ok = false; return; }
            for (int dk = 0; dk < 2; ++dk)
                for (int dj = 0; dj < 2; ++dj)
                    for (int di = 0; di < 2; ++di, ++p) {
                        const int64_t f = (2 * K + dk) * sk + (2 * J + dj) * sj + 2 * I + di;
                        if (R.indices[p] != f || R.data[p] != w0) { ok = false; return; }
                    }
        }
    };
    std::vector<std::thread> th;
    for (int tnum = 1; tnum < nt; ++tnum) th.emplace_back(scan, tnum);
    scan(0);
    for (auto &q : th) q.join();
    return ok;
}

namespace {

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));

// tuning switches (A/B builds: make EXTRA=-DS27_...)
#ifndef S27_COEF_AUX
#define S27_COEF_AUX 2                    // cache policy of the coefficient loads (gfx950: 1 = sc0, 2 = nt, 16 = sc1): they are read once; nt
                                          // keeps them from evicting the iterate from L2 (256^3 fp32 sweep: 447 -> 419 us)
#endif
#ifndef S27_SG
#define S27_SG 14                         // slots per load group (measured: 5: 462 us per sweep, 9: 419, 14: 399, 18: 419, 27: 449)
#endif
#ifndef S27_WAVES
#define S27_WAVES 2                       // launch bound: waves per SIMD the sweep / residual kernels are compiled for
#endif
constexpr int S27_OOB = 0x7FFFFFF0;       // byte offset behind every buffer: the range check answers 0 / drops the store

__device__ __forceinline__ double madd(double v, double x, double acc) { return fma(v, x, acc); }
__device__ __forceinline__ float madd(float v, float x, float acc) { return fmaf(v, x, acc); }

// RG consecutive values of a buffer, starting at element `idx` (any 4-byte aligned offset; outside the buffer: zeros)
template <typename V, int RG>
struct Vec {
    V v[RG];
};
template <int AUX = 0>
__device__ __forceinline__ Vec<float, 4> ldv(__amdgpu_buffer_rsrc_t rs, int idx, Vec<float, 4> *) {
    const v4f q = __builtin_bit_cast(v4f, __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 4, 0, AUX));
    return {{q.x, q.y, q.z, q.w}};
}
template <int AUX = 0>
__device__ __forceinline__ Vec<double, 2> ldv(__amdgpu_buffer_rsrc_t rs, int idx, Vec<double, 2> *) {
    const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, idx * 8, 0, AUX);
    return {{__builtin_bit_cast(double, v2u{q.x, q.y}), __builtin_bit_cast(double, v2u{q.z, q.w})}};
}
template <int AUX = 0>
__device__ __forceinline__ Vec<float, 2> ldv(__amdgpu_buffer_rsrc_t rs, int idx, Vec<float, 2> *) {
    const v2f q = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, idx * 4, 0, AUX));
    return {{q.x, q.y}};
}
template <int AUX = 0>
__device__ __forceinline__ Vec<float, 1> ldv(__amdgpu_buffer_rsrc_t rs, int idx, Vec<float, 1> *) {
    return {{__builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, idx * 4, 0, AUX))}};
}
template <int AUX = 0>
__device__ __forceinline__ Vec<double, 1> ldv(__amdgpu_buffer_rsrc_t rs, int idx, Vec<double, 1> *) {
    return {{__builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, idx * 8, 0, AUX))}};
}
// element-wise masked store (mask bit r: element r is a row of the grid)
__device__ __forceinline__ void stv(__amdgpu_buffer_rsrc_t rs, int idx, const Vec<float, 4> &x, unsigned mask) {
    if (mask == 0xFu) {
        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, v4f{x.v[0], x.v[1], x.v[2], x.v[3]}), rs, idx * 4, 0, 0);
    } else {
#pragma unroll
        for (int r = 0; r < 4; ++r)
            if ((mask >> r) & 1u) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x.v[r]), rs, (idx + r) * 4, 0, 0);
    }
}
__device__ __forceinline__ void stv(__amdgpu_buffer_rsrc_t rs, int idx, const Vec<double, 2> &x, unsigned mask) {
    if (mask == 0x3u) {
        const v2u lo = __builtin_bit_cast(v2u, x.v[0]), hi = __builtin_bit_cast(v2u, x.v[1]);
        __builtin_amdgcn_raw_buffer_store_b128(v4u{lo.x, lo.y, hi.x, hi.y}, rs, idx * 8, 0, 0);
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if ((mask >> r) & 1u) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x.v[r]), rs, (idx + r) * 8, 0, 0);
    }
}

__device__ __forceinline__ void stv(__amdgpu_buffer_rsrc_t rs, int idx, const Vec<float, 2> &x, unsigned mask) {
    if (mask == 0x3u) {
        __builtin_amdgcn_raw_buffer_store_b64(v2u{__builtin_bit_cast(unsigned, x.v[0]), __builtin_bit_cast(unsigned, x.v[1])}, rs, idx * 4, 0, 0);
    } else {
#pragma unroll
        for (int r = 0; r < 2; ++r)
            if ((mask >> r) & 1u) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x.v[r]), rs, (idx + r) * 4, 0, 0);
    }
}
__device__ __forceinline__ void stv(__amdgpu_buffer_rsrc_t rs, int idx, const Vec<float, 1> &x, unsigned mask) {
    if (mask & 1u) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, x.v[0]), rs, idx * 4, 0, 0);
}
__device__ __forceinline__ void stv(__amdgpu_buffer_rsrc_t rs, int idx, const Vec<double, 1> &x, unsigned mask) {
    if (mask & 1u) __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, x.v[0]), rs, idx * 8, 0, 0);
}

template <typename V>
struct S27KArgs {
    const V *coef;                   // [8][ng][27][64 RG]
    const V *x_old;
    V *x_new;                        // sweep: the iterate being written; residual / prolong: unused / the iterate
    const V *b;
    V *res;                          // [2][na]: residuals of colours 6, 7
    const V *e;                      // prolong: coarse correction
    V *bc;                           // restrict: coarse right-hand side
    const int32_t *cmap;             // coarse natural index -> slot (null: identity)
    double *part_a, *part_b;         // block partials: sweep: old-residual squares / new-residual squares; residual: squares
    int hx, hy, L, G;
    int na, nl, ng;                  // rows per colour, lines per colour, waves per colour
    // lines [line_lo, line_hi) are relaxed / evaluated by this launch, squares are taken of lines [sq_lo, sq_hi): all of
    // them on a whole grid; on a slab with ghost aggregate planes (dist27.hip) the owned planes' lines, for the first two
    // pair launches of a sweep also the upper ghost plane's
    int line_lo, line_hi, sq_lo, sq_hi;
    unsigned vec_bytes, coef_bytes;  // 8 na sizeof(V); bytes of ONE colour's coefficients
    V w;
};

// the neighbour of a cell with parity bits (px, py, pz) at slot s: its colour and the shift of its aggregate
struct Nb {
    int colour, si, sj, sk;
};
__host__ __device__ constexpr Nb neighbour(int px, int py, int pz, int s) {
    const int dx = s % 3 - 1, dy = (s / 3) % 3 - 1, dz = s / 9 - 1;
    // (p + d) >> 1 with an arithmetic shift: p = 0: -1 -> -1, +1 -> 0; p = 1: -1 -> 0, +1 -> +1
    const int si = (px + dx) < 0 ? -1 : (px + dx) >> 1, sj = (py + dy) < 0 ? -1 : (py + dy) >> 1, sk = (pz + dz) < 0 ? -1 : (pz + dz) >> 1;
    return Nb{(px ^ (dx != 0 ? 1 : 0)) | ((py ^ (dy != 0 ? 1 : 0)) << 1) | ((pz ^ (dz != 0 ? 1 : 0)) << 2), si, sj, sk};
}

template <typename V, int RG>
struct Lane {
    int line, i0, a0, cbase;         // grid line of the colour, first aggregate in the line, a0 = line hx + i0, first coefficient
    unsigned mask;                   // bit r: aggregate i0 + r exists
    int lane;
    bool counts;                     // the line's squares count towards the norm
};
template <typename V, int RG>
__device__ __forceinline__ Lane<V, RG> lane_of(const S27KArgs<V> &a) {
    Lane<V, RG> t;
    const int wave = int(blockIdx.x) * (int(blockDim.x) >> 6) + (int(threadIdx.x) >> 6);     // (workgroups of four waves, or of one on small levels)
    t.lane = int(threadIdx.x) & 63;
    const int lw = t.lane / a.L;                      // line of the wave
    t.line = wave * a.G + lw;
    t.i0 = (t.lane - lw * a.L) * RG;
    const bool live = wave < a.ng && lw < a.G && t.line >= a.line_lo && t.line < a.line_hi;
    t.counts = t.line >= a.sq_lo && t.line < a.sq_hi;
    t.a0 = t.line * a.hx + t.i0;
    t.cbase = (wave * 27 * 64 + t.lane) * RG;
    t.mask = 0u;
#pragma unroll
    for (int r = 0; r < RG; ++r)
        if (live && t.i0 + r < a.hx) t.mask |= 1u << r;
    if (wave >= a.ng) t.cbase = S27_OOB / int(sizeof(V));          // (a launch is padded to whole workgroups)
    return t;
}

// a workgroup none of whose lines is in the launch's range (slabs: the ghost planes' lines) leaves at once
template <typename V>
__device__ __forceinline__ bool block_dead(const S27KArgs<V> &a) {
    const int wpb = int(blockDim.x) >> 6;
    const int first = int(blockIdx.x) * wpb * a.G, end = first + wpb * a.G;
    return end <= a.line_lo || first >= a.line_hi;
}

template <typename V>
struct Chains4 {
    V s[4];
    __device__ __forceinline__ void clear() { s[0] = s[1] = s[2] = s[3] = V(0); }
    __device__ __forceinline__ V total() const { return ((s[0] + s[1]) + s[2]) + s[3]; }
};

__device__ __forceinline__ void add_sq(double &sq, double r) { sq = fma(r, r, sq); }

// between two groups of slots: no load of the next group is issued before the fmas of this one (a compiler fence for
// memory operations — the machine scheduler's own barrier alone left all 54 loads of a row pair at the top — and a
// scheduling barrier)
// ... and the lane's two base indices pass through an empty asm, so that no ADDRESS of the next group is computed
// ahead of it either (54 offsets with their selects were live at once)
// ... and the group's accumulators too: its fmas are then in front of the fence, not sunk behind the next loads
template <typename C>
__device__ __forceinline__ void pin_chains(C &c) {
    asm volatile("" : "+v"(c.s[0]), "+v"(c.s[1]), "+v"(c.s[2]), "+v"(c.s[3]));
}
__device__ __forceinline__ void group_fence(int &a0, int &cbase) {
    asm volatile("" : "+v"(a0), "+v"(cbase) : : "memory");
    __builtin_amdgcn_sched_barrier(0);
}

// fixed order: lanes of a wave (shuffle tree), then the workgroup's waves in turn
__device__ __forceinline__ void block_partial(double sq, double *out) {
    __shared__ double s_red[4];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
    if (threadIdx.x < 4) s_red[threadIdx.x] = 0.0;
    __syncthreads();
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = sq;
    __syncthreads();
    if (threadIdx.x == 0) out[blockIdx.x] = ((s_red[0] + s_red[1]) + s_red[2]) + s_red[3];
}

// ---- one sweep launch: colours (2 PAIR, 2 PAIR + 1) ---------------------------------------------------------------
// XZ: x_old is zero and not read.  NOLD: also the squares of b - A x_old (both colours) -> part_a.  LAST (PAIR == 3):
// the residuals of both colours with the FINAL iterate -> res, their squares -> part_b when it is given.
// The 27 slots of a row are taken in groups of SG: the group's coefficient and operand loads are issued together,
// then its fmas; a scheduling barrier between the groups keeps the compiler from hoisting every load of the unrolled
// row to the top (54 sixteen-byte loads in flight: 256 registers and scratch).
constexpr int SG = S27_SG;
template <typename V, int RG, int PAIR, bool XZ, bool NOLD, bool LAST>
__global__ __launch_bounds__(256, NOLD ? 1 : S27_WAVES) void s27_sweep_kernel(const S27KArgs<V> a) {
    static_assert(!LAST || PAIR == 3, "only the last pair's rows are final when their launch ends");
    static_assert(!(XZ && NOLD), "a zero iterate has no predecessor cycle");
    typedef Vec<V, RG> VR;
    constexpr int CA = 2 * PAIR, CB = CA + 1, PY = PAIR & 1, PZ = PAIR >> 1;
    if (block_dead(a)) {
        if (threadIdx.x == 0) {
            if (NOLD) a.part_a[blockIdx.x] = 0.0;
            if (LAST && a.part_b) a.part_b[blockIdx.x] = 0.0;
        }
        return;
    }
    Lane<V, RG> tt = lane_of<V, RG>(a);
    const Lane<V, RG> &t = tt;
    const int lhx = a.hx, lhyhx = a.hx * a.hy;
    const __amdgpu_buffer_rsrc_t xo = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.x_old), 0, XZ ? 0u : a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t xn = __builtin_amdgcn_make_buffer_rsrc(a.x_new, 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bs = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.b), 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ka = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.coef + size_t(CA) * (a.coef_bytes / sizeof(V))), 0, a.coef_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t kb = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.coef + size_t(CB) * (a.coef_bytes / sizeof(V))), 0, a.coef_bytes, 0x00020000);
    const int ok = t.mask ? 1 : 0;                     // (a lane without rows asks for nothing)
    constexpr int NONE = S27_OOB / int(sizeof(V));
    auto cload = [&](const __amdgpu_buffer_rsrc_t &rs, int s) -> VR { return ldv<S27_COEF_AUX>(rs, ok ? t.cbase + s * 64 * RG : NONE, (VR *)nullptr); };
    // operand of a slot: current = x_new for colours already relaxed in this sweep (< CA), x_old otherwise
    auto operand = [&](const __amdgpu_buffer_rsrc_t &rs, const Nb &nb) -> VR {
        return ldv(rs, ok ? nb.colour * a.na + t.a0 + nb.sk * lhyhx + nb.sj * lhx + nb.si : NONE, (VR *)nullptr);
    };
    double sq_old = 0.0, sq_new = 0.0;
    Chains4<V> acc[RG], aold[RG];

    // ---- colour CA (x even) ----
    VR xa = {}, ba = ldv(bs, ok ? CA * a.na + t.a0 : NONE, (VR *)nullptr);
    if (!XZ) xa = ldv(xo, ok ? CA * a.na + t.a0 : NONE, (VR *)nullptr);
#pragma unroll
    for (int r = 0; r < RG; ++r) { acc[r].clear(); aold[r].clear(); }
    VR diag_a = {};
    V pre_a[3][RG], a3[RG];                            // LAST: colour CA's chains 0, 1, 2 in front of slots 12, 13, 14; its chain 3
#pragma unroll
    for (int r = 0; r < RG; ++r) pre_a[0][r] = pre_a[1][r] = pre_a[2][r] = a3[r] = V(0);
#pragma unroll
    for (int g0 = 0; g0 < 27; g0 += SG) {
        VR c[SG], x[SG], xold[SG];
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            const int s = g0 + u;
            if (s >= 27) continue;
            const Nb nb = neighbour(0, PY, PZ, s);
            const bool from_new = nb.colour < CA;
            if (XZ && !from_new && s != 13) continue;  // (c * 0 leaves a chain as it is; the coefficient is not even loaded)
            c[u] = cload(ka, s);
            if (s == 13) { diag_a = c[u]; x[u] = xa; xold[u] = xa; continue; }
            x[u] = operand(from_new ? xn : xo, nb);
            if (NOLD) xold[u] = from_new ? operand(xo, nb) : x[u];
        }
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            const int s = g0 + u;
            if (s >= 27) continue;
            const Nb nb = neighbour(0, PY, PZ, s);
            if (LAST && (s == 12 || s == 13 || s == 14)) {      // (before the skip below) chains 0, 1, 2 in front of slots 12, 13, 14
#pragma unroll
                for (int r = 0; r < RG; ++r) pre_a[s - 12][r] = acc[r].s[s & 3];
            }
            if (XZ && !(nb.colour < CA)) continue;     // (slot 13 too: x_i = 0)
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                acc[r].s[s & 3] = madd(c[u].v[r], x[u].v[r], acc[r].s[s & 3]);
                if (NOLD) aold[r].s[s & 3] = madd(c[u].v[r], xold[u].v[r], aold[r].s[s & 3]);
            }
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) { pin_chains(acc[r]); if (NOLD) pin_chains(aold[r]); }
        group_fence(tt.a0, tt.cbase);
    }
    VR na_;                                            // the relaxed values of colour CA
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        // openmg/solvers.py:68   x[i] = x[i] + (b[i] - Aix) / A[i, i]
        const V v = xa.v[r] + (ba.v[r] - acc[r].total()) / diag_a.v[r];
        na_.v[r] = ((t.mask >> r) & 1u) ? v : V(0);
        if (NOLD && ((t.mask >> r) & 1u) && t.counts) add_sq(sq_old, double(ba.v[r] - aold[r].total()));
        if (LAST) a3[r] = acc[r].s[3];
    }
    // the in-line neighbours of colour CB: aggregate i + 1 of this lane or the next lane's first; of colour CA's
    // residual: aggregate i - 1 of this lane or the previous lane's last (a line's first / last: no such cell, zero
    // coefficient — whatever finite value the neighbouring lane holds)
    const V na_next = __shfl_down(na_.v[0], 1, 64);

    // ---- colour CB (x odd) ----
    VR xb = {}, bb = ldv(bs, ok ? CB * a.na + t.a0 : NONE, (VR *)nullptr);
    if (!XZ) xb = ldv(xo, ok ? CB * a.na + t.a0 : NONE, (VR *)nullptr);
#pragma unroll
    for (int r = 0; r < RG; ++r) { acc[r].clear(); aold[r].clear(); }
    VR diag_b = {};
    V keep1[RG];                                       // LAST: chain 1 in front of slot 13
#pragma unroll
    for (int r = 0; r < RG; ++r) keep1[r] = V(0);
#pragma unroll
    for (int g0 = 0; g0 < 27; g0 += SG) {
        VR c[SG], x[SG], xold[SG];
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            const int s = g0 + u;
            if (s >= 27) continue;
            const Nb nb = neighbour(1, PY, PZ, s);
            const bool in_line = s == 12 || s == 14;   // colour CA of the own line: relaxed above
            const bool from_new = nb.colour < CA;
            if (XZ && !from_new && !in_line && s != 13) continue;
            c[u] = cload(kb, s);
            if (s == 13) { diag_b = c[u]; x[u] = xb; xold[u] = xb; }
            else if (s == 12) { x[u] = na_; xold[u] = xa; }          // dx = -1: the same aggregate's colour CA cell
            else if (s == 14) {                                        // dx = +1: the next aggregate's
#pragma unroll
                for (int r = 0; r + 1 < RG; ++r) x[u].v[r] = na_.v[r + 1];
                x[u].v[RG - 1] = na_next;
                if (NOLD) xold[u] = operand(xo, nb);
            } else {
                x[u] = operand(from_new ? xn : xo, nb);
                if (NOLD) xold[u] = from_new ? operand(xo, nb) : x[u];
            }
        }
#pragma unroll
        for (int u = 0; u < SG; ++u) {
            const int s = g0 + u;
            if (s >= 27) continue;
            const Nb nb = neighbour(1, PY, PZ, s);
            const bool in_line = s == 12 || s == 14;
            if (LAST && s == 13) {                     // (before the skip below: with XZ slot 13 itself adds nothing)
#pragma unroll
                for (int r = 0; r < RG; ++r) keep1[r] = acc[r].s[1];
            }
            if (XZ && !(nb.colour < CA) && !in_line) continue;
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                acc[r].s[s & 3] = madd(c[u].v[r], x[u].v[r], acc[r].s[s & 3]);
                if (NOLD) aold[r].s[s & 3] = madd(c[u].v[r], xold[u].v[r], aold[r].s[s & 3]);
            }
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) { pin_chains(acc[r]); if (NOLD) pin_chains(aold[r]); }
        group_fence(tt.a0, tt.cbase);
    }
    VR nb_;
#pragma unroll
    for (int r = 0; r < RG; ++r) {
        const V v = xb.v[r] + (bb.v[r] - acc[r].total()) / diag_b.v[r];
        nb_.v[r] = ((t.mask >> r) & 1u) ? v : V(0);
        if (NOLD && ((t.mask >> r) & 1u) && t.counts) add_sq(sq_old, double(bb.v[r] - aold[r].total()));
    }
    stv(xn, CA * a.na + t.a0, na_, t.mask);
    stv(xn, CB * a.na + t.a0, nb_, t.mask);

    if (LAST) {
        // The rows of colours 6 and 7 are final: their residuals with the new iterate, as a residual pass would form
        // them — the same chains, with the operands that have changed since the row was relaxed.
        // Colour 7 (CB): only its own value (slot 13, chain 1): chain 1 again from slot 13 on.
        const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(a.res, 0, unsigned(2 * size_t(a.na) * sizeof(V)), 0x00020000);
        VR rb;
        {
            V c1[RG];
#pragma unroll
            for (int r = 0; r < RG; ++r) c1[r] = madd(diag_b.v[r], nb_.v[r], keep1[r]);
            VR c[3], x[3];
#pragma unroll
            for (int u = 0; u < 3; ++u) {
                const Nb nb = neighbour(1, PY, PZ, 17 + 4 * u);
                c[u] = cload(kb, 17 + 4 * u);
                x[u] = operand(xn, nb);                   // (colours 0 .. 5: relaxed by the earlier launches of this sweep)
            }
#pragma unroll
            for (int u = 0; u < 3; ++u)
#pragma unroll
                for (int r = 0; r < RG; ++r) c1[r] = madd(c[u].v[r], x[u].v[r], c1[r]);
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const V tot = ((acc[r].s[0] + c1[r]) + acc[r].s[2]) + acc[r].s[3];
                rb.v[r] = ((t.mask >> r) & 1u) ? bb.v[r] - tot : V(0);
            }
        }
        group_fence(tt.a0, tt.cbase);
        // Colour 6 (CA): its own value (slot 13) and its two in-line neighbours of colour 7 (slots 12, 14) have changed
        // since its rows were relaxed: chains 0, 1, 2 again from those slots on (their prefixes and chain 3 were kept),
        // the nine later coefficients and operands loaded again (operands: what the sweep itself used)
        const V nb_prev = __shfl_up(nb_.v[RG - 1], 1, 64);
        VR ra;
        {
            V ch[3][RG];
            const VR c12 = cload(ka, 12), c14 = cload(ka, 14);
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const V x12 = r ? nb_.v[r - 1] : nb_prev;                    // dx = -1: the previous aggregate's colour 7 cell
                ch[0][r] = madd(c12.v[r], x12, pre_a[0][r]);
                ch[1][r] = madd(diag_a.v[r], na_.v[r], pre_a[1][r]);
                ch[2][r] = madd(c14.v[r], nb_.v[r], pre_a[2][r]);            // dx = +1: the same aggregate's
            }
            VR c[9], x[9];
#pragma unroll
            for (int u = 0; u < 9; ++u) {
                const int sl = 16 + 4 * (u / 3) + (u % 3);                   // 16, 17, 18, 20, 21, 22, 24, 25, 26
                c[u] = cload(ka, sl);
                x[u] = operand(xn, neighbour(0, PY, PZ, sl));                // (colours 0 .. 5)
            }
#pragma unroll
            for (int u = 0; u < 9; ++u)
#pragma unroll
                for (int r = 0; r < RG; ++r) ch[u % 3][r] = madd(c[u].v[r], x[u].v[r], ch[u % 3][r]);
#pragma unroll
            for (int r = 0; r < RG; ++r) {
                const V tot = ((ch[0][r] + ch[1][r]) + ch[2][r]) + a3[r];
                ra.v[r] = ((t.mask >> r) & 1u) ? ba.v[r] - tot : V(0);
            }
        }
        stv(rr, t.a0, ra, t.mask);
        stv(rr, a.na + t.a0, rb, t.mask);
        if (a.part_b && t.counts) {
#pragma unroll
            for (int r = 0; r < RG; ++r) { add_sq(sq_new, double(ra.v[r])); add_sq(sq_new, double(rb.v[r])); }
        }
    }
    if (NOLD) block_partial(sq_old, a.part_a);
    if (LAST && a.part_b) block_partial(sq_new, a.part_b);
}

// ---- residual of the colours [0, NC) (NC = 6: those of colours 6, 7 are read from res), then ----------------------
// MODE 0: the restriction, openmg/__init__.py:210 — a coarse cell's eight fine residuals in R's column order, which
//         IS the colour order — into the coarse right-hand side;  MODE 1: their squares -> part_a (NC colours only).
// The colours are a run-time loop (the colour is wave-uniform: its neighbour table is scalar arithmetic), the 27 slots
// of a row unrolled in groups of SG like the sweep's.
template <typename V, int RG, int NC, int MODE>
__global__ __launch_bounds__(256, S27_WAVES) void s27_residual_kernel(const S27KArgs<V> a) {
    typedef Vec<V, RG> VR;
    if (block_dead(a)) {
        if (MODE == 1 && threadIdx.x == 0) a.part_a[blockIdx.x] = 0.0;
        return;
    }
    Lane<V, RG> tt = lane_of<V, RG>(a);
    const Lane<V, RG> &t = tt;
    const int lhx = a.hx, lhyhx = a.hx * a.hy;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.x_old), 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bs = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.b), 0, a.vec_bytes, 0x00020000);
    const int ok = t.mask ? 1 : 0;
    constexpr int NONE = S27_OOB / int(sizeof(V));
    V rsum[RG];                                        // MODE 0: the restriction's running chain
#pragma unroll
    for (int r = 0; r < RG; ++r) rsum[r] = V(0);
    double sq = 0.0;
#pragma unroll 1
    for (int c = 0; c < NC; ++c) {
        const int px = c & 1, py = (c >> 1) & 1, pz = c >> 2;
        const __amdgpu_buffer_rsrc_t kc = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.coef + size_t(c) * (a.coef_bytes / sizeof(V))), 0, a.coef_bytes, 0x00020000);
        const VR bv = ldv(bs, ok ? c * a.na + t.a0 : NONE, (VR *)nullptr);
        Chains4<V> acc[RG];
#pragma unroll
        for (int r = 0; r < RG; ++r) acc[r].clear();
#pragma unroll
        for (int g0 = 0; g0 < 27; g0 += SG) {
            VR cf[SG], x[SG];
#pragma unroll
            for (int u = 0; u < SG; ++u) {
                const int s = g0 + u;
                if (s >= 27) continue;
                const Nb nb = neighbour(px, py, pz, s);
                cf[u] = ldv<S27_COEF_AUX>(kc, ok ? t.cbase + s * 64 * RG : NONE, (VR *)nullptr);
                x[u] = ldv(xs, ok ? nb.colour * a.na + t.a0 + nb.sk * lhyhx + nb.sj * lhx + nb.si : NONE, (VR *)nullptr);
            }
#pragma unroll
            for (int u = 0; u < SG; ++u) {
                if (g0 + u >= 27) continue;
#pragma unroll
                for (int r = 0; r < RG; ++r) acc[r].s[(g0 + u) & 3] = madd(cf[u].v[r], x[u].v[r], acc[r].s[(g0 + u) & 3]);
            }
#pragma unroll
            for (int r = 0; r < RG; ++r) pin_chains(acc[r]);
            group_fence(tt.a0, tt.cbase);
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            const V res = ((t.mask >> r) & 1u) ? bv.v[r] - acc[r].total() : V(0);
            if (MODE == 1) { if (t.counts) add_sq(sq, double(res)); }
            else rsum[r] = madd(a.w, res, rsum[r]);
        }
    }
    if (MODE == 0) {
        if (NC < 8) {
            const __amdgpu_buffer_rsrc_t rr = __builtin_amdgcn_make_buffer_rsrc(a.res, 0, unsigned(2 * size_t(a.na) * sizeof(V)), 0x00020000);
            const VR r6 = ldv(rr, ok ? t.a0 : NONE, (VR *)nullptr), r7 = ldv(rr, ok ? a.na + t.a0 : NONE, (VR *)nullptr);
#pragma unroll
            for (int r = 0; r < RG; ++r) rsum[r] = madd(a.w, r7.v[r], madd(a.w, r6.v[r], rsum[r]));
        }
#pragma unroll
        for (int r = 0; r < RG; ++r) {
            if (!((t.mask >> r) & 1u)) continue;
            const int ca = t.a0 + r;                   // the coarse cell's natural index IS the aggregate's
            a.bc[a.cmap ? a.cmap[ca] : ca] = rsum[r];
        }
    } else {
        block_partial(sq, a.part_a);
    }
}

// x += R^T e, openmg/__init__.py:214,220 — the product rounded, then added (ROW_SCATTER's two roundings)
template <typename V, int RG>
__global__ __launch_bounds__(256) void s27_prolong_kernel(const S27KArgs<V> a) {
    typedef Vec<V, RG> VR;
    if (block_dead(a)) return;
    const Lane<V, RG> t = lane_of<V, RG>(a);
    if (!t.mask) return;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(a.x_new, 0, a.vec_bytes, 0x00020000);
    VR e;
#pragma unroll
    for (int r = 0; r < RG; ++r) e.v[r] = ((t.mask >> r) & 1u) ? a.e[a.cmap ? a.cmap[t.a0 + r] : t.a0 + r] : V(0);
    VR x[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) x[c] = ldv(xs, c * a.na + t.a0, (VR *)nullptr);
#pragma unroll
    for (int c = 0; c < 8; ++c) {
#pragma unroll
        for (int r = 0; r < RG; ++r) x[c].v[r] = x[c].v[r] + madd(a.w, e.v[r], V(0));
        stv(xs, c * a.na + t.a0, x[c], t.mask);
    }
}

// ---- setup: the caller's CSR (natural numbering) -> the wave-tiled coefficient array; every row checked ----------
// err: 0 fine, otherwise 1 + the first offending row (any of them)
template <typename V>
// Slabs (dist27.hip): the CSR is the rank's extended slab (ghost planes' rows empty); planes [kz_lo, kz_hi) of it exist in
// the global grid (a neighbour outside is absent), the rows of planes [kb_lo, kb_hi) are tiled, the others left alone.
__global__ __launch_bounds__(256) void s27_build_kernel(const int32_t *indptr, const int32_t *indices, const double *data,
                                                        int nx, int ny, int nz, int kz_lo, int kz_hi, int kb_lo, int kb_hi, int L, int G, int RG, int64_t ng,
                                                        V *coef, unsigned long long *err) {
    const int64_t n = int64_t(nx) * ny * nz;
    const int64_t row = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (row >= n) return;
    const int i = int(row % nx), j = int((row / nx) % ny), k = int(row / (int64_t(nx) * ny));
    if (k < kb_lo || k >= kb_hi) return;
    const int hy = ny / 2;
    const int c = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2);
    const int64_t line = int64_t(k >> 1) * hy + (j >> 1);
    const int64_t wave = line / G;
    const int lane = int(line - wave * G) * L + (i >> 1) / RG, r = (i >> 1) % RG;
    V *out = coef + ((size_t(c) * size_t(ng) + size_t(wave)) * 27 * 64 + size_t(lane)) * size_t(RG) + size_t(r);
    int64_t p = indptr[row];
    const int64_t pe = indptr[row + 1];
    bool bad = false;
    for (int s = 0; s < 27; ++s) {
        const int dx = s % 3 - 1, dy = (s / 3) % 3 - 1, dz = s / 9 - 1;
        const bool present = i + dx >= 0 && i + dx < nx && j + dy >= 0 && j + dy < ny && k + dz >= kz_lo && k + dz < kz_hi;
        V v = V(0);
        if (present) {
            if (p < pe && int64_t(indices[p]) == row + (int64_t(dz) * ny + dy) * nx + dx) {
                v = V(data[p]);
                const double back = double(v);
                if (!(back - back == 0.0)) bad = true;                 // not finite (as V)
                if (s == 13 && !(back != 0.0)) bad = true;             // a zero diagonal
                ++p;
            } else {
                bad = true;
            }
        }
        out[size_t(s) * 64 * size_t(RG)] = v;
    }
    if (p != pe) bad = true;
    if (bad) atomicMin(err, (unsigned long long)(row + 1));
}

template <typename V>
S27KArgs<V> base_args(const Stencil27Plan<V> &P) {
    S27KArgs<V> k;
    std::memset(&k, 0, sizeof(k));
    const S27Geom &g = P.g;
    k.coef = P.coef.p;
    k.res = P.res67.p;
    k.hx = g.hx; k.hy = g.hy; k.L = g.L; k.G = g.G;
    k.na = int(g.na); k.nl = int(g.nl); k.ng = int(g.ng);
    k.vec_bytes = unsigned(size_t(8) * size_t(g.na) * sizeof(V));
    k.coef_bytes = unsigned(size_t(g.ng) * 27 * 64 * size_t(g.rg) * sizeof(V));
    k.w = V(g.w);
    k.line_lo = int(P.live_lo); k.line_hi = int(P.live_hi);
    k.sq_lo = int(P.live_lo); k.sq_hi = int(P.live_hi);
    return k;
}

template <typename K, typename V>
void launch_s27(K kernel, const S27Geom &g, const S27KArgs<V> &k, hipStream_t s) {
    hipLaunchKernelGGL(kernel, dim3(unsigned(g.n_wg)), dim3(unsigned(64 * g.wpb)), 0, s, k);
    OMG_HIP(hipGetLastError());
}

template <typename V, int RG, int PAIR>
void launch_pair(const S27Geom &g, const S27KArgs<V> &k, bool x_zero, bool nold, bool last, hipStream_t s) {
    if (PAIR == 3 && last) {
        if (x_zero) launch_s27(s27_sweep_kernel<V, RG, 3, true, false, true>, g, k, s);
        else if (nold) launch_s27(s27_sweep_kernel<V, RG, 3, false, true, true>, g, k, s);
        else launch_s27(s27_sweep_kernel<V, RG, 3, false, false, true>, g, k, s);
    } else {
        if (x_zero) launch_s27(s27_sweep_kernel<V, RG, PAIR, true, false, false>, g, k, s);
        else if (nold) launch_s27(s27_sweep_kernel<V, RG, PAIR, false, true, false>, g, k, s);
        else launch_s27(s27_sweep_kernel<V, RG, PAIR, false, false, false>, g, k, s);
    }
}

template <typename V, int RG>
void sweep_rg(const Stencil27Plan<V> &P, const V *x_old, V *x_new, const V *b, bool x_zero, double *norm_old, bool last, double *norm_new,
              hipStream_t s) {
    S27KArgs<V> k = base_args(P);
    k.x_old = x_old; k.x_new = x_new; k.b = b;
    const int nw = P.g.n_wg;
    k.part_b = (last && norm_new) ? norm_new + 3 * size_t(nw) : nullptr;
    // (a slab: the colours 0 .. 3 of the upper ghost aggregate plane are relaxed here too — the neighbour's rows, its bits —
    // so that colours 4 .. 7 of the last owned plane find them without an exchange inside the sweep)
    k.line_lo = int(P.live_lo01); k.line_hi = int(P.live_hi01);
    k.part_a = norm_old;                      launch_pair<V, RG, 0>(P.g, k, x_zero, norm_old != nullptr, false, s);
    k.part_a = norm_old ? norm_old + nw : nullptr;     launch_pair<V, RG, 1>(P.g, k, x_zero, norm_old != nullptr, false, s);
    k.line_lo = int(P.live_lo); k.line_hi = int(P.live_hi);
    k.part_a = norm_old ? norm_old + 2 * size_t(nw) : nullptr; launch_pair<V, RG, 2>(P.g, k, x_zero, norm_old != nullptr, false, s);
    k.part_a = norm_old ? norm_old + 3 * size_t(nw) : nullptr; launch_pair<V, RG, 3>(P.g, k, x_zero, norm_old != nullptr, last, s);
}

template <typename V, int RG>
void residual_rg(const Stencil27Plan<V> &P, const V *x, const V *b, bool use67, int mode, const int32_t *cmap, V *bc, double *out, hipStream_t s) {
    S27KArgs<V> k = base_args(P);
    k.x_old = x; k.b = b; k.cmap = cmap; k.bc = bc; k.part_a = out;
    if (mode == 0) {
        if (use67) launch_s27(s27_residual_kernel<V, RG, 6, 0>, P.g, k, s);
        else launch_s27(s27_residual_kernel<V, RG, 8, 0>, P.g, k, s);
    } else {
        if (use67) launch_s27(s27_residual_kernel<V, RG, 6, 1>, P.g, k, s);
        else launch_s27(s27_residual_kernel<V, RG, 8, 1>, P.g, k, s);
    }
}

// RG: 16-byte accesses (4 floats, 2 doubles) on the large levels, fewer aggregates per lane — more waves — on the small ones
template <typename V> constexpr int rg_max() { return sizeof(V) == 4 ? 4 : 2; }
template <typename V, typename F>
void with_rg(int rg, F &&f) {
    if (rg == 1) f(std::integral_constant<int, 1>());
    else if (rg == 2) f(std::integral_constant<int, 2>());
    else if (sizeof(V) == 4) f(std::integral_constant<int, rg_max<V>()>());
    else f(std::integral_constant<int, 2>());
}

}  // namespace

template <typename V>
bool Stencil27Plan<V>::build(const omg_csr &A, const omg_csr &R, Ordering &ord, hipStream_t s) {
    {
        const char *e = getenv("OMG_STENCIL27");
        if (e && e[0] == '0') return false;
    }
    const int64_t n = A.n_rows;
    if (n < 64 || n != A.n_cols || (n & 7)) return false;
    auto has = [&](int64_t r, int64_t c) {
        for (int64_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p)
            if (A.indices[p] == c) return true;
        return false;
    };
    // the grid, read off the couplings (as PlanePlan::build does)
    int64_t nx = n;
    for (int64_t r = 1; r < n; ++r)
        if (!has(r, r - 1)) { nx = r; break; }
    if (nx < 2 || n % nx) return false;
    const int64_t lines = n / nx;
    int64_t ny = lines;
    for (int64_t q = 1; q < lines; ++q)
        if (!has(q * nx, (q - 1) * nx)) { ny = q; break; }
    if (lines % ny) return false;
    const int64_t nz = lines / ny;
    if ((nx & 1) || (ny & 1) || (nz & 1) || ny < 2 || nz < 2) return false;
    // every row holds exactly its in-grid neighbours: the entry count is then (3 nx - 2)(3 ny - 2)(3 nz - 2)
    if (A.nnz != (3 * nx - 2) * (3 * ny - 2) * (3 * nz - 2)) return false;
    // a 27-point row next to the grid's corner: the diagonal neighbour must be there (a 7-point operator is not this)
    if (!has(nx * ny + nx + 1, 0)) return false;
    double w = 0.0;
    if (!is_plain_aggregation(R, nx, ny, nz, w)) return false;
    // the operator onto the device as it is, every row checked and scattered into the tiles there
    require_device();
    DevCsrPlain D;
    {
        SetupTimer tm("27-point level: upload CSR");
        D.n_rows = D.n_cols = n; D.nnz = A.nnz;
        D.indptr.alloc(size_t(n) + 1);
        D.indices.alloc(size_t(std::max<int64_t>(A.nnz, 1)));
        D.data.alloc(size_t(std::max<int64_t>(A.nnz, 1)));
        OMG_HIP(hipMemcpyAsync(D.indptr.p, A.indptr, (size_t(n) + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
        OMG_HIP(hipMemcpyAsync(D.indices.p, A.indices, size_t(A.nnz) * sizeof(int32_t), hipMemcpyHostToDevice, s));
        OMG_HIP(hipMemcpyAsync(D.data.p, A.data, size_t(A.nnz) * sizeof(double), hipMemcpyHostToDevice, s));
        OMG_HIP(hipStreamSynchronize(s));
    }
    if (!build_device(D, int(nx), int(ny), int(nz), w, ord, s)) return false;
    materialise_ordering(ord);
    return true;
}

// the tiling of an nx x ny x nz grid (false: a line does not fit a wave, or the arrays would outgrow 32-bit byte offsets)
template <typename V>
static bool s27_geometry(int nx, int ny, int nz, double w, S27Geom &q) {
    const int64_t n = int64_t(nx) * ny * nz;
    if (n < 64 || (nx & 1) || (ny & 1) || (nz & 1) || ny < 2 || nz < 2) return false;
    q.nx = int(nx); q.ny = int(ny); q.nz = int(nz);
    q.hx = q.nx / 2; q.hy = q.ny / 2; q.hz = q.nz / 2;
    q.na = int64_t(q.hx) * q.hy * q.hz;
    q.rg = rg_max<V>();
    if (q.hx > 64 * q.rg) return false;                          // a grid line must fit one wave (nx <= 512 in float, 256 in double)
    q.nl = int64_t(q.hy) * q.hz;
    {
        // small levels: fewer aggregates per lane and one-wave workgroups, so that the launch still spreads over the chip
        const char *e = experiment_env("OMG_S27_RG");
        const int forced = e ? atoi(e) : 0;
        for (;;) {
            q.L = (q.hx + q.rg - 1) / q.rg;
            q.G = 64 / q.L;
            q.ng = (q.nl + q.G - 1) / q.G;
            if (forced == 1 || forced == 2 || forced == 4) {
                if (q.rg == forced || q.rg == 1 || forced > rg_max<V>()) break;
            } else if (q.ng >= 1024 || q.rg == 1) {
                break;
            }
            q.rg /= 2;
        }
    }
    q.wpb = q.ng >= 1024 ? 4 : 1;
    q.n_wg = int((q.ng + q.wpb - 1) / q.wpb);
    q.w = double(V(w));
    const uint64_t coef_colour_bytes = uint64_t(q.ng) * 27 * 64 * uint64_t(q.rg) * sizeof(V);
    if (uint64_t(n) * sizeof(V) >= (uint64_t(1) << 31) || coef_colour_bytes >= (uint64_t(1) << 31)) return false;
    return true;
}

template <typename V>
bool Stencil27Plan<V>::tile(const DevCsrPlain &A, const S27Geom &q, int kz_lo, int kz_hi, int kb_lo, int kb_hi, hipStream_t s) {
    SetupTimer tm("27-point level: check + tile the coefficients on the device");
    const int64_t n = int64_t(q.nx) * q.ny * q.nz;
    DevBuf<unsigned long long> d_err(1);
    OMG_HIP(hipMemsetAsync(d_err.p, 0xFF, sizeof(unsigned long long), s));
    // (OMG_S27_PLACE: how a large level's tiles are placed — common.h DevBuf::alloc; experiment, default ordinary)
    static const int place = [] { const char *e = experiment_env("OMG_S27_PLACE"); return e && e[0] ? atoi(e) : 0; }();
    coef.alloc(size_t(8) * size_t(q.ng) * 27 * 64 * size_t(q.rg), 0, n >= (int64_t(1) << 23) ? place : 0);
    coef.zero(s);
    hipLaunchKernelGGL(s27_build_kernel<V>, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, A.indptr.p, A.indices.p, A.data.p, q.nx, q.ny, q.nz,
                       kz_lo, kz_hi, kb_lo, kb_hi, q.L, q.G, q.rg, q.ng, coef.p, d_err.p);
    OMG_HIP(hipGetLastError());
    unsigned long long err = 0;
    OMG_HIP(hipMemcpyAsync(&err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    if (err != ~0ull) { coef.release(); return false; }
    g = q;
    res67.alloc(size_t(2) * size_t(g.na));
    partials.alloc(size_t(4) * size_t(g.n_wg) + SUM_FOLD);
    partials.zero(s);
    have67 = false;
    live_lo = live_lo01 = 0;
    live_hi = live_hi01 = g.nl;
    return true;
}

template <typename V>
bool Stencil27Plan<V>::build_device(const DevCsrPlain &A, int nx, int ny, int nz, double w, Ordering &ord, hipStream_t s) {
    {
        const char *e = getenv("OMG_STENCIL27");
        if (e && e[0] == '0') return false;
    }
    const int64_t n = int64_t(nx) * ny * nz;
    if (n != A.n_rows || n != A.n_cols) return false;
    if (A.nnz != (3 * int64_t(nx) - 2) * (3 * int64_t(ny) - 2) * (3 * int64_t(nz) - 2)) return false;
    S27Geom q;
    if (!s27_geometry<V>(nx, ny, nz, w, q)) return false;
    if (!tile(A, q, 0, nz, 0, nz, s)) return false;
    // the ordering: colour = octant, inside a colour the aggregates in natural order (what the greedy colouring of
    // such an operator gives: every lower-numbered neighbour of a cell lies in another octant position); its host
    // arrays are written on demand (materialise_ordering)
    ord = Ordering();
    ord.identity = false;
    ord.sets.resize(9);
    for (int c = 0; c <= 8; ++c) ord.sets[size_t(c)] = int64_t(c) * g.na;
    ord.closed_form = 8; ord.cf_nx = nx; ord.cf_ny = ny; ord.cf_nz = nz;
    return true;
}

// One rank's slab of a 27-point level (dist27.hip).  A: the rows of the rank's OWNED planes in the numbering of its
// extended slab — nz_ext = owned + 4 planes: one ghost AGGREGATE plane on either side, whose rows are empty here (the
// upper one's even plane is filled from the neighbour afterwards: copy_plane_rows / pack_plane_rows).  first / last: the
// slab lies at the global grid's lower / upper end (the ghost planes there do not exist: couplings to them are absent).
template <typename V>
void Stencil27Plan<V>::build_slab(const DevCsrPlain &A, int nx, int ny, int nz_ext, bool first, bool last, double w, hipStream_t s) {
    const int64_t n = int64_t(nx) * ny * nz_ext;
    OMG_REQUIRE(n == A.n_rows && n == A.n_cols && nz_ext >= 6, "27-point slab: operator does not match the extended slab");
    S27Geom q;
    OMG_REQUIRE(s27_geometry<V>(nx, ny, nz_ext, w, q), "27-point slab: the grid does not fit the kernels (even extents, nx <= 512 (float) / 256 (double), "
                                                       "one colour's vector and coefficients below 2 GiB)");
    OMG_REQUIRE(tile(A, q, first ? 2 : 0, last ? nz_ext - 2 : nz_ext, 2, nz_ext - 2, s),
                "27-point slab: a row of the rank's planes is not the full 27-point stencil of its in-grid neighbours (sorted columns, finite, nonzero diagonal)");
    // owned aggregate planes 1 .. hz - 2; colours 0 .. 3 are relaxed on the upper ghost plane too (where there is one)
    live_lo = live_lo01 = int64_t(g.hy);
    live_hi = int64_t(g.hy) * (g.hz - 1);
    live_hi01 = last ? live_hi : int64_t(g.hy) * g.hz;
}

// colours 0 .. 3 of aggregate plane K, every slot: [colour][line J][slot][aggregate I] <-> the tiles
template <typename V>
__global__ void s27_plane_rows_kernel(V *coef, V *buf, int hx, int hy, int L, int G, int RG, int64_t ng, int K, int to_buf) {
    const int64_t total = int64_t(4) * hy * 27 * hx;
    for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
        const int I = int(e % hx), sl = int((e / hx) % 27), J = int((e / (int64_t(hx) * 27)) % hy), c = int(e / (int64_t(hx) * 27 * hy));
        const int64_t line = int64_t(K) * hy + J, wave = line / G;
        const int lane = int(line - wave * G) * L + I / RG, r = I % RG;
        V *t = coef + ((size_t(c) * size_t(ng) + size_t(wave)) * 27 * 64 + size_t(lane)) * size_t(RG) + size_t(r) + size_t(sl) * 64 * size_t(RG);
        if (to_buf) buf[e] = *t;
        else *t = buf[e];
    }
}

template <typename V>
size_t Stencil27Plan<V>::plane_rows_count() const { return size_t(4) * size_t(g.hy) * 27 * size_t(g.hx); }

template <typename V>
void Stencil27Plan<V>::pack_plane_rows(int K, V *buf, hipStream_t s) const {
    hipLaunchKernelGGL(s27_plane_rows_kernel<V>, dim3(unsigned(std::min<size_t>(4096, (plane_rows_count() + 255) / 256))), dim3(256), 0, s,
                       const_cast<V *>(coef.p), buf, g.hx, g.hy, g.L, g.G, g.rg, g.ng, K, 1);
    OMG_HIP(hipGetLastError());
}

template <typename V>
void Stencil27Plan<V>::unpack_plane_rows(int K, const V *buf, hipStream_t s) {
    hipLaunchKernelGGL(s27_plane_rows_kernel<V>, dim3(unsigned(std::min<size_t>(4096, (plane_rows_count() + 255) / 256))), dim3(256), 0, s,
                       coef.p, const_cast<V *>(buf), g.hx, g.hy, g.L, g.G, g.rg, g.ng, K, 0);
    OMG_HIP(hipGetLastError());
}

template <typename V>
void Stencil27Plan<V>::sweep(const V *x_old, V *x_new, const V *b, bool x_zero, double *norm_old, bool last, double *norm_new, hipStream_t s) {
    with_rg<V>(g.rg, [&](auto RGC) { sweep_rg<V, decltype(RGC)::value>(*this, x_old, x_new, b, x_zero, norm_old, last, norm_new, s); });
    have67 = last;
}

template <typename V>
void Stencil27Plan<V>::residual_restrict(const V *x, const V *b, bool use67, const int32_t *cmap, V *bc, hipStream_t s) {
    OMG_REQUIRE(!use67 || have67, "internal: residuals of colours 6, 7 asked for but not there");
    with_rg<V>(g.rg, [&](auto RGC) { residual_rg<V, decltype(RGC)::value>(*this, x, b, use67, 0, cmap, bc, nullptr, s); });
}

template <typename V>
void Stencil27Plan<V>::norm(const V *x, const V *b, bool use67, double *out, hipStream_t s) {
    OMG_REQUIRE(!use67 || have67, "internal: squares of colours 6, 7 asked for but not there");
    const size_t nw = size_t(g.n_wg);
    OMG_HIP(hipMemsetAsync(out + nw, 0, (use67 ? 2 : 3) * nw * sizeof(double), s));
    with_rg<V>(g.rg, [&](auto RGC) { residual_rg<V, decltype(RGC)::value>(*this, x, b, use67, 1, nullptr, nullptr, out, s); });
}

template <typename V>
void Stencil27Plan<V>::prolong(V *x, const V *e, const int32_t *cmap, hipStream_t s) {
    S27KArgs<V> k = base_args(*this);
    k.x_new = x; k.e = e; k.cmap = cmap;
    k.line_lo = 0; k.line_hi = int(g.nl);                          // (a slab's ghost planes too: their coarse cells' corrections are known)
    with_rg<V>(g.rg, [&](auto RGC) { launch_s27(s27_prolong_kernel<V, decltype(RGC)::value>, g, k, s); });
    have67 = false;
}

template <typename V>
HostCsr Stencil27Plan<V>::operator_csr(hipStream_t s) const {
    const int64_t nx = g.nx, ny = g.ny, nz = g.nz, n = nx * ny * nz;
    // (the padded operator has 27 n entries behind int32 row pointers: a level that large has no set-by-set twin)
    OMG_REQUIRE(27 * n < (int64_t(1) << 31), "27-point level: the row-kernel format of the padded operator needs 27 n < 2^31 entries");
    std::vector<V> host(coef.n);
    OMG_HIP(hipMemcpyAsync(host.data(), coef.p, coef.n * sizeof(V), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    HostCsr A;
    A.n_rows = A.n_cols = n;
    A.nnz = 27 * n;
    A.indptr.resize(size_t(n) + 1);
    A.indices.resize(size_t(A.nnz));
    A.data.resize(size_t(A.nnz));
    const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const int nt = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 16384));
    auto fill = [&](int tnum) {
        const int64_t lo = n * tnum / nt, hi = n * (tnum + 1) / nt;
        for (int64_t row = lo; row < hi; ++row) {
            const int i = int(row % nx), j = int((row / nx) % ny), k = int(row / (nx * ny));
            const int c = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2);
            const int64_t line = int64_t(k >> 1) * g.hy + (j >> 1), wave = line / g.G;
            const int lane = int(line - wave * g.G) * g.L + (i >> 1) / g.rg, r = (i >> 1) % g.rg;
            const V *in = host.data() + ((size_t(c) * size_t(g.ng) + size_t(wave)) * 27 * 64 + size_t(lane)) * size_t(g.rg) + size_t(r);
            A.indptr[size_t(row)] = int32_t(27 * row);
            for (int sl = 0; sl < 27; ++sl) {
                const int dx = sl % 3 - 1, dy = (sl / 3) % 3 - 1, dz = sl / 9 - 1;
                const bool present = i + dx >= 0 && i + dx < nx && j + dy >= 0 && j + dy < ny && k + dz >= 0 && k + dz < nz;
                // an absent neighbour: an explicit zero on the row's own column (it adds +0 to the diagonal's sum)
                A.indices[size_t(27 * row + sl)] = int32_t(present ? row + (int64_t(dz) * ny + dy) * nx + dx : row);
                A.data[size_t(27 * row + sl)] = present ? double(in[size_t(sl) * 64 * size_t(g.rg)]) : 0.0;
            }
        }
    };
    std::vector<std::thread> th;
    for (int tnum = 1; tnum < nt; ++tnum) th.emplace_back(fill, tnum);
    fill(0);
    for (auto &t2 : th) t2.join();
    A.indptr[size_t(n)] = int32_t(27 * n);
    return A;
}

template <typename V>
HostCsr Stencil27Plan<V>::restriction_csr() const {
    const int64_t nx = g.nx, ny = g.ny, nz = g.nz, n = nx * ny * nz, sj = nx, sk = nx * ny;
    const int64_t nxc = nx / 2, nyc = ny / 2, nc = nxc * nyc * (nz / 2);
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
                    R.data[size_t(p)] = g.w;
                }
    }
    return R;
}

// ---- new coefficients into an existing level, and the Galerkin product in closed form (round 5) ---------------------
// openmg/operators.py:184-186 for the plain 2 x 2 x 2 aggregation of a 27-point level: A_c(I, I + D) = sum over the fine
// cells j of aggregate I + D, ascending, of w (R A)(I, j), and (R A)(I, j) = sum over the fine rows k of aggregate I,
// ascending, of w a(k, j) — SciPy's accumulation order, which the one-wave-per-coarse-row hash kernel of setup_device.hip
// reproduces for ANY aggregation at 7.2 ms for the 256^3 operator.  On a grid the j that a coarse row touches are the 4 x 4 x 4
// cells around its aggregate, each fed by fixed (child, slot) pairs: a thread takes one coarse row, streams its eight fine
// rows once (27 values each; a neighbour outside the grid is a zero), adds w a(k, j) into 64 registers in the order
// above, and forms the 27 coarse entries from them.  The same thread writes the fine rows' coefficients into the fine
// level's tiles (as V) — new values for an existing level — and the coarse row into the next level's tiles and into a
// dense [row][27] double array, which is the next product's input: the whole chain of a hierarchy reads every operator once.
// Same additions in the same order as the hash kernel, products by w exact (w is a power of two, checked by the caller):
// the same bits (tests/test_gpu_update.py).
template <typename V>
struct S27RapArgs {
    const int32_t *indptr;           // CSR input (the caller's operator): row pointers; null: dense [row][27] input
    const double *vals;
    int nx, ny, nz;                  // the fine grid
    V *fine_coef;                    // the fine level's tiles (null: not written)
    int fL, fG, fRG; int64_t fng;
    double *coarse_dense;            // [coarse row][27] (null: not written)
    V *coarse_coef;                  // the coarse level's tiles (null: the coarse level is not a 27-point level)
    int cL, cG, cRG; int64_t cng;
    double w;
    unsigned long long *err;         // smallest offending fine row + 1
};

template <typename F, int... Is>
__device__ __forceinline__ void static_for_impl(F &&f, std::integer_sequence<int, Is...>) { (f(std::integral_constant<int, Is>()), ...); }
template <int N, typename F>
__device__ __forceinline__ void static_for(F &&f) { static_for_impl(f, std::make_integer_sequence<int, N>()); }
typedef double rap_d2 __attribute__((ext_vector_type(2)));
typedef rap_d2 rap_d2u __attribute__((aligned(8)));

// Threads -> coarse rows: first the rows with 1 <= I <= hx - 2, line by line, then the two ends of every line.  Of the
// first kind a thread's two fine rows of a (dj, dk) are 54 CONSECUTIVE values of the CSR input wherever their lines are
// inside the grid in y and z, and the chunks of consecutive threads follow one another in memory.  A lane that walked its
// own chunk (432 bytes from its neighbour's) touched a cache line of its own with every load, 64 lines per instruction,
// and with eight waves on a compute unit the lines were gone from the L1 before the lane's next load came back to them:
// 16.5 GB from L2 to L1 for 3.7 GB of input (PMC), 3.7 ms with 8-byte loads, 2.9 ms with 16-byte ones.  So the WAVE fetches
// its threads' chunks together — lane l of load u takes the 16 bytes at position 128 u + 2 l of the wave's concatenated
// chunks (unit stride wherever the chunks are contiguous: every line is requested once), through 27.6 KB of LDS per
// wave — and every thread then takes its 54 values out of LDS (27 16-byte reads, the lanes 432 bytes apart:
// conflict-free) in the order the sums want them: 1.5-1.8 ms.  (Staged 16 / 22 / 32 threads at a time — less LDS, eight
// waves per compute unit instead of four, the next group's loads in flight — 2.0 / 2.0 / 1.8-2.1 ms: the threads that wait
// for their group cost more than the occupancy gives; one-wave workgroups 3.5 ms.)  Rows at the grid's faces have fewer
// entries: their threads walk them one by one as before, by themselves.
constexpr int RAP_WAVES = 4, RAP_LOADS = 64 * 54 / 128;      // 27 loads of 64 x 16 bytes: the 54 values of 64 threads
template <typename V, bool CSR_IN, bool WRITE_FINE>
__global__ __launch_bounds__(64 * RAP_WAVES) void s27_rap_kernel(const S27RapArgs<V> a) {
    __shared__ __attribute__((aligned(16))) double s_stage[RAP_WAVES][RAP_LOADS * 128];
    __shared__ int64_t s_base[RAP_WAVES][64];
    const int hx = a.nx / 2, hy = a.ny / 2, hz = a.nz / 2;
    const int64_t lines = int64_t(hy) * hz, nc = lines * hx;
    const int wv = int(threadIdx.x) >> 6, ln = int(threadIdx.x) & 63;
    int64_t idx = int64_t(blockIdx.x) * (64 * RAP_WAVES) + threadIdx.x;
    const bool valid = idx < nc;                       // (a thread behind the last row still fetches for its wave)
    if (!valid) idx = nc - 1;
    const int hx2 = hx > 2 ? hx - 2 : 0;
    const int64_t n_mid = lines * hx2;
    int I;
    int64_t line;
    if (idx < n_mid) { line = idx / hx2; I = 1 + int(idx - line * hx2); }
    else {
        const int64_t e = idx - n_mid;                 // the lines' ends: 2 per line (hx > 2), else all hx cells of it
        const int per = hx - hx2;
        line = e / per;
        const int o = int(e - line * per);
        I = (hx2 && o) ? hx - 1 : o;
    }
    const int J = int(line % hy), K = int(line / hy);
    const int64_t crow = line * hx + I;
    const bool mid_x = I >= 1 && I <= hx - 2;
    double RA[64];
#pragma unroll
    for (int q = 0; q < 64; ++q) RA[q] = 0.0;
    const int64_t wave = line / a.fG;
    const int lane = int(line - wave * a.fG) * a.fL + I / a.fRG, rr = I % a.fRG;
    unsigned bad = 0;                                  // bit c: a value of child c is wrong (branch-free throughout: a select per value, not a jump)
    auto wave_sync = [] {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
    };
    static_for<4>([&](auto PAIR) __attribute__((always_inline)) {
        constexpr int pair = decltype(PAIR)::value;
        constexpr int dj = pair & 1, dk = pair >> 1;
        const int j = 2 * J + dj, k = 2 * K + dk;
        const int64_t row0 = (int64_t(k) * a.ny + j) * a.nx + 2 * I;
        const bool ym = j > 0, yp = j + 1 < a.ny, zm = k > 0, zp = k + 1 < a.nz;
        V *tp = a.fine_coef + ((size_t(2 * pair) * size_t(a.fng) + size_t(wave)) * 27 * 64 + size_t(lane)) * size_t(a.fRG) + size_t(rr);
        // one value of child (di, dj, dk), slot sl: into the fine tiles, and w a(k, j) into (R A)(I, j) — cell j at
        // (dk + dz + 1, dj + dy + 1, di + dx + 1) of the 4 x 4 x 4 cells around the aggregate; the rows k in ascending order =
        // the children in this loop's order (di fastest)
        auto consume = [&](auto DI, auto SL, double v) __attribute__((always_inline)) {
            constexpr int di = decltype(DI)::value, sl = decltype(SL)::value;
            constexpr int dx = sl % 3 - 1, dy = (sl / 3) % 3 - 1, dz = sl / 9 - 1;
            if (WRITE_FINE) {
                const V t = V(v);
                const bool wrong = !__builtin_isfinite(t) || (sl == 13 && t == V(0));
                bad |= wrong ? 1u << (2 * pair + di) : 0u;
                // (the values arrive in the tiles' order — child di, slot by slot —: ONE running address, not 54 of them in registers)
                *tp = t;
                tp += sl == 26 ? (ptrdiff_t(a.fng) * 27 - 26) * 64 * ptrdiff_t(a.fRG) : 64 * ptrdiff_t(a.fRG);
                asm volatile("" : "+v"(tp));
            }
            const int q = ((dk + dz + 1) * 4 + (dj + dy + 1)) * 4 + (di + dx + 1);
            RA[q] = __dadd_rn(RA[q], __dmul_rn(a.w, v));
            asm volatile("" : "+v"(RA[q]));            // (the sum is formed HERE: left to itself the compiler keeps a pair's 54 loaded values until all have arrived)
        };
        const bool fast = valid && (CSR_IN ? (mid_x && ym && yp && zm && zp) : true);
        wave_sync();                                   // (the previous pair's readers are done with the bases and the stage)
        s_base[wv][ln] = fast ? (CSR_IN ? int64_t(a.indptr[row0]) : row0 * 27) : int64_t(-1);
        wave_sync();
        // lane ln of load u: the 16 bytes at doubles 128 u + 2 ln of the wave's concatenated chunks = thread (128 u + 2 ln) / 54's;
        // all 27 requested before the first is put down (one latency, not 27)
        rap_d2 got[RAP_LOADS];
#pragma unroll
        for (int u = 0; u < RAP_LOADS; ++u) {
            const unsigned f = 128u * unsigned(u) + 2u * unsigned(ln);
            const unsigned owner = f / 54u, o = f - 54u * owner;
            const int64_t base = s_base[wv][owner];
            got[u] = base >= 0 ? *reinterpret_cast<const rap_d2u *>(a.vals + base + o) : rap_d2{0.0, 0.0};
        }
#pragma unroll
        for (int u = 0; u < RAP_LOADS; ++u) *reinterpret_cast<rap_d2 *>(&s_stage[wv][128 * u + 2 * ln]) = got[u];
        wave_sync();
        if (fast) {
            const double *const src = &s_stage[wv][54 * ln];
            static_for<27>([&](auto M) __attribute__((always_inline)) {
                constexpr int m = decltype(M)::value;
                const rap_d2 c = *reinterpret_cast<const rap_d2 *>(src + 2 * m);
                consume(std::integral_constant<int, (2 * m) / 27>(), std::integral_constant<int, (2 * m) % 27>(), c.x);
                consume(std::integral_constant<int, (2 * m + 1) / 27>(), std::integral_constant<int, (2 * m + 1) % 27>(), c.y);
                if (m % 6 == 5) asm volatile("" ::: "memory");     // (six reads in flight at a time: all 27 would take 108 registers)
            });
        }
        if (valid && !fast) {
            static_for<2>([&](auto DI) __attribute__((always_inline)) {
                constexpr int di = decltype(DI)::value;
                const int i = 2 * I + di;
                const bool xm = i > 0, xp = i + 1 < a.nx;
                const int64_t p0 = int64_t(a.indptr[row0 + di]);
                int64_t p = p0;
                static_for<27>([&](auto SL) __attribute__((always_inline)) {
                    constexpr int sl = decltype(SL)::value;
                    constexpr int dx = sl % 3 - 1, dy = (sl / 3) % 3 - 1, dz = sl / 9 - 1;
                    const bool present = (dx < 0 ? xm : dx > 0 ? xp : true) && (dy < 0 ? ym : dy > 0 ? yp : true) && (dz < 0 ? zm : dz > 0 ? zp : true);
                    const double got1 = a.vals[present ? p : p0];      // (a slot that is not there: the row's first entry, not used)
                    p += present ? 1 : 0;
                    consume(DI, SL, present ? got1 : 0.0);
                });
            });
        }
    });
    if (!valid) return;
    if (bad) {
        const int child = __builtin_ctz(bad);          // (the children's rows ascend with their number)
        const int64_t row = (int64_t(2 * K + (child >> 2)) * a.ny + 2 * J + ((child >> 1) & 1)) * a.nx + 2 * I + (child & 1);
        atomicMin(a.err, (unsigned long long)(row + 1));
    }
    // the coarse row: slot D = (Dz, Dy, Dx) takes the cells of aggregate I + D — along an axis cell 0 for -1, cells 1, 2
    // for 0, cell 3 for +1 —, ascending
    const int c_colour = (I & 1) | ((J & 1) << 1) | ((K & 1) << 2);
    V *ctile = nullptr;
    if (a.coarse_coef) {
        const int chy = hy / 2;
        const int64_t cline = int64_t(K >> 1) * chy + (J >> 1), cwave = cline / a.cG;
        const int clane = int(cline - cwave * a.cG) * a.cL + (I >> 1) / a.cRG, cr = (I >> 1) % a.cRG;
        ctile = a.coarse_coef + ((size_t(c_colour) * size_t(a.cng) + size_t(cwave)) * 27 * 64 + size_t(clane)) * size_t(a.cRG) + size_t(cr);
    }
#pragma unroll
    for (int sl = 0; sl < 27; ++sl) {
        const int Dx = sl % 3 - 1, Dy = (sl / 3) % 3 - 1, Dz = sl / 9 - 1;
        const int x0 = Dx < 0 ? 0 : Dx == 0 ? 1 : 3, x1 = Dx == 0 ? 3 : x0 + 1;
        const int y0 = Dy < 0 ? 0 : Dy == 0 ? 1 : 3, y1 = Dy == 0 ? 3 : y0 + 1;
        const int z0 = Dz < 0 ? 0 : Dz == 0 ? 1 : 3, z1 = Dz == 0 ? 3 : z0 + 1;
        double sum = 0.0;
#pragma unroll
        for (int z = z0; z < z1; ++z)
#pragma unroll
            for (int y = y0; y < y1; ++y)
#pragma unroll
                for (int x = x0; x < x1; ++x) sum = __dadd_rn(sum, __dmul_rn(RA[(z * 4 + y) * 4 + x], a.w));
        if (a.coarse_dense) a.coarse_dense[crow * 27 + sl] = sum;
        if (ctile) {
            // what s27_build_kernel checks on a fresh level: finite as the level's type, a diagonal that is not zero
            const V cv = V(sum);
            if (!isfinite(cv) || (sl == 13 && cv == V(0))) atomicMin(a.err, (1ull << 62) | (unsigned long long)(crow + 1));
            ctile[size_t(sl) * 64 * size_t(a.cRG)] = cv;
        }
    }
}

template <typename V>
void Stencil27Plan<V>::rap_from(const int32_t *indptr, const double *vals, bool write_fine, double *coarse_dense, Stencil27Plan<V> *coarse, hipStream_t s) {
    S27RapArgs<V> a;
    std::memset(&a, 0, sizeof(a));
    a.indptr = indptr; a.vals = vals;
    a.nx = g.nx; a.ny = g.ny; a.nz = g.nz;
    a.fine_coef = write_fine ? coef.p : nullptr;
    a.fL = g.L; a.fG = g.G; a.fRG = g.rg; a.fng = g.ng;
    a.coarse_dense = coarse_dense;
    if (coarse) {
        OMG_REQUIRE(coarse->g.nx == g.hx && coarse->g.ny == g.hy && coarse->g.nz == g.hz, "internal: the coarse 27-point level does not sit under this one");
        a.coarse_coef = coarse->coef.p;
        a.cL = coarse->g.L; a.cG = coarse->g.G; a.cRG = coarse->g.rg; a.cng = coarse->g.ng;
    }
    a.w = g.w;
    DevBuf<unsigned long long> d_err(1);
    OMG_HIP(hipMemsetAsync(d_err.p, 0xFF, sizeof(unsigned long long), s));
    a.err = d_err.p;
    const int64_t nc = g.na;
    constexpr int T = 64 * RAP_WAVES;
    const dim3 grid(unsigned((nc + T - 1) / T));
    if (indptr && write_fine) hipLaunchKernelGGL((s27_rap_kernel<V, true, true>), grid, dim3(T), 0, s, a);
    else if (indptr) hipLaunchKernelGGL((s27_rap_kernel<V, true, false>), grid, dim3(T), 0, s, a);
    else if (write_fine) hipLaunchKernelGGL((s27_rap_kernel<V, false, true>), grid, dim3(T), 0, s, a);
    else hipLaunchKernelGGL((s27_rap_kernel<V, false, false>), grid, dim3(T), 0, s, a);
    OMG_HIP(hipGetLastError());
    unsigned long long err = 0;
    OMG_HIP(hipMemcpyAsync(&err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    if (err != ~0ull && (err >> 62))
        throw Error(OMG_ERR_INVALID, "27-point level: the Galerkin product's row " + std::to_string((err & ~(3ull << 62)) - 1) +
                                         " of the next level is not finite as the level's type, or its diagonal is zero");
    if (err != ~0ull)
        throw Error(OMG_ERR_INVALID, "27-point level: new coefficient of row " + std::to_string(err - 1) + " is not finite as the level's type, or its diagonal is zero");
    have67 = false;
    if (coarse) coarse->have67 = false;
}

// WHERE a large level's coefficient tiles (1.8 GB at 256^3 fp32: what its sweeps stream) lie in HBM moves the sweep: 388-391 us
// on some allocations and 432-448 us on others (profiles/r05_pool_placement.txt, section 7; where the level's VECTORS lie
// moves it by 1 %) — as for the plane levels' vectors (hierarchy.hip place_finest_pool), nothing a process can ask the driver
// for decides it.  So: candidates — a copy of the tiles in another allocation, hipMalloc and scattered pieces in turn — are
// timed with the level's own sweep until one is good (the sweep at 5.5 TB/s of its needed bytes) or OMG_S27_TRIALS (4) are
// tried; no more than 8 GB of them are held.
template <typename V>
void Stencil27Plan<V>::place_tiles(V *x, V *tmp, V *b, hipStream_t s) {
    const char *e = getenv("OMG_S27_TRIALS");
    const int trials = e && e[0] ? atoi(e) : 4;
    const bool keep_last = getenv("OMG_PLACE_KEEP_LAST") != nullptr;       // (tests: the newest candidate is kept whatever its time)
    if (trials < 2 || !coef.p || int64_t(g.nx) * g.ny * g.nz < (int64_t(1) << 23)) return;
    SetupTimer tm("placement of a large 27-point level's tiles (timed)");
    const int64_t n = int64_t(g.nx) * g.ny * g.nz;
    const int max_trials = int(std::min<size_t>(size_t(trials), std::max<size_t>(2, (size_t(8) << 30) / (coef.n * sizeof(V)))));
    hipEvent_t e0, e1;
    OMG_HIP(hipEventCreate(&e0));
    OMG_HIP(hipEventCreate(&e1));
    auto timed = [&]() -> float {
        OMG_HIP(hipMemsetAsync(x, 0, size_t(n) * sizeof(V), s));
        OMG_HIP(hipMemsetAsync(tmp, 0, size_t(n) * sizeof(V), s));
        OMG_HIP(hipMemsetAsync(b, 0, size_t(n) * sizeof(V), s));
        sweep(x, tmp, b, false, nullptr, false, nullptr, s);
        OMG_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < 2; ++r) {
            sweep(tmp, x, b, false, nullptr, false, nullptr, s);
            sweep(x, tmp, b, false, nullptr, false, nullptr, s);
        }
        OMG_HIP(hipEventRecord(e1, s));
        OMG_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        OMG_HIP(hipEventElapsedTime(&ms, e0, e1));
        return 1e3f * ms / 4.0f;
    };
    const bool debug = SetupTimer::on();
    // "good": a sweep moves its (27 + 6) w n bytes at 5.5 TB/s or more (403 us at 256^3 fp32: the fast kind 367-395, the others 416-448)
    const double sweep_bytes = 33.0 * double(sizeof(V)) * double(n);
    auto good = [&](float us) { return sweep_bytes / (double(us) * 1e-6) >= 5.5e12; };
    float best = timed(), worst = best;
    if (debug) fprintf(stderr, "[omg setup] 27-point tiles, candidate 0 (as built): %.1f us per sweep\n", best);
    std::vector<DevBuf<V>> held;
    for (int k = 1; k < max_trials; ++k) {
        if (!keep_last && (good(best) || (k >= 3 && best <= 0.93f * worst))) break;
        DevBuf<V> alt;
        try { alt.alloc(coef.n, 0, pool_placement(k + 1)); }           // (k = 1: hipMalloc again, then 2 MiB pieces, 32 MiB pieces, ...)
        catch (const Error &) { (void)hipGetLastError(); break; }       // (no memory for another candidate: what there is stays)
        OMG_HIP(hipMemcpyAsync(alt.p, coef.p, coef.n * sizeof(V), hipMemcpyDeviceToDevice, s));
        OMG_HIP(hipStreamSynchronize(s));
        std::swap(coef, alt);                                           // coef: the candidate, alt: the best so far
        const float t = timed();
        if (debug) fprintf(stderr, "[omg setup] 27-point tiles, candidate %d (placement %d): %.1f us per sweep\n", k, pool_placement(k + 1), t);
        worst = std::max(worst, t);
        if (t < best || keep_last) best = std::min(best, t);
        else std::swap(coef, alt);
        held.push_back(std::move(alt));
    }
    OMG_HIP(hipStreamSynchronize(s));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

template struct Stencil27Plan<double>;
template struct Stencil27Plan<float>;

}  // namespace omg
