// Plane-pipelined red-black passes of grid star stencils (common.h PlanePlan) — gfx950 only.
//
// The two halves of a V(1,1) cycle over a red-black ordered level (openmg/__init__.py:201-227) as
// ONE launch each.  Level vectors are in the colour ordering the set-by-set kernels use: the red
// cells ((i + j + k) even) in natural order, then the black ones, so a grid line (j, k) of nx cells
// is two runs of hx = nx / 2 values — its reds at [(k ny + j) hx, +hx) and its blacks nr further —
// and in that "half-index" space the seven-point stencil is regular: a cell h of one colour has its
// four j / k neighbours at the SAME h of the other colour's run of the neighbouring line, and its
// two i neighbours at h - 1 + p, h + p (red) or h - p, h + 1 - p (black), p = (j + k) & 1.
//
// A thread owns four consecutive cells of two adjacent lines — per line one pair (h = 2q, 2q + 1)
// of either colour, 16-byte accesses throughout — and keeps every plane it still needs of them in
// registers while the workgroup marches in z.  Step s:
//     B  red   sweep of plane s      (black: old values of planes s-1, s, s+1)
//     C  black sweep of plane s-1    (red: new values of s-2, s-1, s), and that row's residual
//     D  red residual of plane s-2   (black: new values of s-3, s-2, s-1)
//     restriction of a finished pair of planes (down) / squares for the norm (up)
// Values of the neighbouring lines and of the neighbouring threads' cells come from three LDS
// images written at the end of the previous step (double buffered: one barrier per step).  The
// (x, y) tile carries a ring that is relaxed redundantly — red two cells deep, black one — so no
// workgroup waits for another; everything is out of place (x_old read, x_new written by the owner).
// All global traffic of a step is issued at its top (the loads the NEXT step consumes, the stores
// of what the previous step finished), so the one wait per step finds them done.
//
// What else lives here:
//   * instantiations: PEER — a slab whose neighbours are reached by peer stores (PlanePlan::Peer, dist.hip); FIRST —
//     the coarse level's first relaxation is written too (only when that level does not start from zero); MAXT /
//     LA = 2 — small workgroups with two steps of lookahead; the loop runs two steps per iteration where the
//     registers allow it, so that the step's parity is a compile-time constant;
//   * block_kernel — levels of <= 64^3 cells below the finest: a block of the grid per workgroup, whole in LDS;
//   * choose_tiles (two cost models) and PlanePlan::tune, which times their winners on the level's own vectors.
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstring>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>

#include "common.h"

// -DOMG_PLANE_STAMPS: in-kernel cycle stamps per stage (tools/plane_stamps.py); -DOMG_PLANE_DBGARGS: only the switches
// of OMG_PLANE_DBG that leave out part of a pass's memory traffic (tools/plane_dbg_times.py; wrong results, timing only)
#if defined(OMG_PLANE_STAMPS) || defined(OMG_PLANE_DBGARGS)
#define OMG_PLANE_DBG_ON 1
#endif

namespace omg {
namespace {

// cache policy of the x_new stores: 0 = plain; 2 = nt; 16 = sc1 (write-through: the lines leave the XCD's L2 while the
// kernel runs instead of in one write-back at its end — the 5-6 us gap behind every level-0 pass in the
// trace); measured at 256^3 over whole cycles: plain 3276 V-cycles/s, nt 3304, sc1 3254, sc0 | sc1 3136; plain and
// nt interleaved in one call, three times: nt ahead by 0.3-0.7 % each time (down pass -1 to -2 us): nt it is
#ifndef PLANE_STORE_AUX
#define PLANE_STORE_AUX 2
#endif
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));

__device__ __forceinline__ double madd(double v, double x, double acc) { return fma(v, x, acc); }
__device__ __forceinline__ float madd(float v, float x, float acc) { return fmaf(v, x, acc); }

template <typename V>
struct P2 {
    V x, y;
};
template <typename V> struct VecOf;
template <> struct VecOf<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    typedef type gtype __attribute__((aligned(8)));       // global: aligned like one element
};
template <> struct VecOf<float> {
    typedef float type __attribute__((ext_vector_type(2)));
    typedef type gtype __attribute__((aligned(4)));
};

// (Measured and removed in round 5, docs/HISTORY.md has the numbers: the step's stores of x_new at its top behind or in front
// of its loads instead of behind stages B and C (+ 1 to + 3 us per pass); the chunk's last planes written through the L2;
// non-temporal vector loads (+ 30 us: the rings stop sharing the L2).)
__device__ __forceinline__ P2<double> bload2(__amdgpu_buffer_rsrc_t rs, int off, double) {
    const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, off, 0, 0);
    return {__builtin_bit_cast(double, v2u{q.x, q.y}), __builtin_bit_cast(double, v2u{q.z, q.w})};
}
__device__ __forceinline__ P2<float> bload2(__amdgpu_buffer_rsrc_t rs, int off, float) {
    // (the whole vector is cast: hipcc 7.2 turns __builtin_bit_cast(float, q.y) of a vector ELEMENT into
    // element 0 and narrows the load to one dword)
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v2f q = __builtin_bit_cast(v2f, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
    return {q.x, q.y};
}
__device__ __forceinline__ double bload1(__amdgpu_buffer_rsrc_t rs, int off, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, off, 0, 0));
}
__device__ __forceinline__ float bload1(__amdgpu_buffer_rsrc_t rs, int off, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, off, 0, 0));
}
template <typename V>
__device__ __forceinline__ P2<V> lds_pair(const V *p) {
    const typename VecOf<V>::type q = *reinterpret_cast<const typename VecOf<V>::type *>(p);
    return {q.x, q.y};
}
template <typename V>
__device__ __forceinline__ void lds_put(V *p, const P2<V> &v) {
    typename VecOf<V>::type q;
    q.x = v.x;
    q.y = v.y;
    *reinterpret_cast<typename VecOf<V>::type *>(p) = q;
}
// AUX: the store's cache policy bits (gfx950: 1 = sc0, 2 = nt, 16 = sc1; sc0 | sc1 = written through at system scope)
template <int AUX = PLANE_STORE_AUX>
__device__ __forceinline__ void bstore2(__amdgpu_buffer_rsrc_t rs, int off, const P2<double> &v, bool both) {
    if (both) {
        const v2u lo = __builtin_bit_cast(v2u, v.x), hi = __builtin_bit_cast(v2u, v.y);
        __builtin_amdgcn_raw_buffer_store_b128(v4u{lo.x, lo.y, hi.x, hi.y}, rs, off, 0, AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v.x), rs, off, 0, AUX);
    }
}
// (float: plain stores — with nt the fp32 cycle dropped from 4890 to 4480 V-cycles/s; its stores are 8 bytes a lane)
template <int AUX = 0>
__device__ __forceinline__ void bstore2(__amdgpu_buffer_rsrc_t rs, int off, const P2<float> &v, bool both) {
    if (both) {
        __builtin_amdgcn_raw_buffer_store_b64(v2u{__builtin_bit_cast(unsigned, v.x), __builtin_bit_cast(unsigned, v.y)}, rs, off, 0, AUX);
    } else {
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v.x), rs, off, 0, AUX);
    }
}
template <int AUX>
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t rs, int off, double v) {
    __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, v), rs, off, 0, AUX);
}
template <int AUX>
__device__ __forceinline__ void bstore1(__amdgpu_buffer_rsrc_t rs, int off, float v) {
    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), rs, off, 0, AUX);
}
constexpr int PEER_PLANES = 3, PEER_CPLANES = 2;   // boundary planes of x / of the coarse right-hand side a neighbour takes (a chunk holds at least four planes)
constexpr int PEER_AUX = 1 | 16;     // stores into a neighbour's memory: through this GPU's caches, acknowledged when they are out
template <typename V>
__device__ __forceinline__ void store2(V *p, const P2<V> &v, bool both) {
    if (both) {
        typename VecOf<V>::gtype q;
        q.x = v.x;
        q.y = v.y;
        *reinterpret_cast<typename VecOf<V>::gtype *>(p) = q;
    } else {
        p[0] = v.x;
    }
}

#ifndef PLANE_LA_BIG
#define PLANE_LA_BIG 1               // lookahead of the whole-grid passes of large levels (2: experiments; the registers do not allow it)
#endif
#ifndef PLANE_LA
#define PLANE_LA 1
#endif
#ifndef PLANE_WAVE_SYNC
#define PLANE_WAVE_SYNC 1            // 0: a workgroup barrier per step
#endif
constexpr unsigned WAVE_SYNC_SPIN = 1u << 22;     // polls of a neighbour wave's step count (~0.1 us each) before a wave gives up
constexpr int OOB = 0x7FFFFFF0;      // a byte offset behind every vector: the buffer's range check answers 0

template <typename V>
struct PlaneKArgs {
    const V *x_old;
    V *x_new;
    const V *b;
    unsigned vec_bytes;              // n * sizeof(V)
    int nr;                          // slot of the first black cell
    int hx, ny, nz;
    // a slab of a larger grid (plane_dist.hip): the planes [z_base, z_end) are this launch's (relaxed, stored), the
    // planes [kv0, kv1) exist in the global grid (the others — ghost planes beyond its first / last plane — are
    // zeros and stay zeros); a whole grid: 0, nz, 0, nz.  kc_off: coarse plane of fine plane k = (k >> 1) + kc_off
    int z_base, z_end, kv0, kv1, kc_off;
    int TXq, TY, LZ, PX, PY, ntx, nty, ntz;
    // the z chunks of this launch: chunk i = planes [zc_base + i zc_stride, + zc_len) clipped to zc_end.  A whole pass:
    // z_base, LZ, LZ, z_end.  A pass made of two launches (PlanePlan::PART_*): the EDGE launch has two short chunks — the
    // slab's first and last PLANE_EDGE planes —, the INNER launch chunks of LZ planes between them; the norm's partial of a
    // workgroup then goes to slot part_slot0 + its number (the two launches fill one array)
    int zc_base, zc_stride, zc_len, zc_end, part_slot0;
    // ONE launch of inner and edge chunks (PlanePlan::Gate: a slab whose ghost planes arrive by an exchange on another stream
    // while the pass runs): workgroups [0, edge_wg0) take the inner chunks above, the others — dispatched last — the slab's
    // first and last PLANE_EDGE planes, and wait for wait_flag[] before they read anything.  0: no such launch
    int edge_wg0;
    V c0, c1, c2, c3, c4, c5, c6, w;
    int x_zero;
    int fast_div;                    // the diagonal's exponent is within 2^-400 .. 2^400 (quotients())
    // coarse level
    int nxc, nyc, nzc;
    unsigned cvec_bytes, cmap_bytes;
    const int32_t *cmap;
    V *bc, *xc;
    const V *cdiag;
    int first_end;
    const V *ec;
    double *partials;
    // slab neighbours (PEER kernels; PlanePlan::Peer)
    V *peer_x[2], *peer_bc[2];
    int peer_shift, peer_cshift;
    const uint32_t *wait_flag[4];
    uint32_t wait_seq[4];
    int fused_wait;
    uint32_t *done, *peer_flag[2];
    uint32_t flag_seq, spin;
    uint32_t *status;
#ifdef OMG_PLANE_STAMPS
    unsigned long long *stamps;      // diagnostic build: per workgroup and wave, cycles spent waiting for loads / computing / at the barrier
#endif
#ifdef OMG_PLANE_DBG_ON
    int dbg;                         // ... OMG_PLANE_DBG bits: 1 no x stores, 2 no coarse stores, 4 no loads of x, 8 no loads of b, 16 no loads of the neighbouring chunks' planes (wrong results, timing only)
#endif
};

// The row of a cell as the row kernels walk it: one fma chain over the seven slots in column order
// (-K, -J, -I, diagonal, +I, +J, +K), started from +0.  The first three terms do not involve the
// cell itself, so a sweep's chain and the residual chain after it share them.
template <typename V>
__device__ __forceinline__ V chain_head(const PlaneKArgs<V> &a, V km, V jm, V im) {
    V s = madd(a.c0, km, V(0));
    s = madd(a.c1, jm, s);
    return madd(a.c2, im, s);
}
template <typename V>
__device__ __forceinline__ V chain_tail(const PlaneKArgs<V> &a, V s, V d, V ip, V jp, V kp) {
    s = madd(a.c3, d, s);
    s = madd(a.c4, ip, s);
    s = madd(a.c5, jp, s);
    return madd(a.c6, kp, s);
}

// n / c3 for the relaxation.  The compiler's expansion of a double division — v_div_scale x 2, v_rcp_f64,
// two Newton steps on the reciprocal, quotient, one correction, v_div_fmas, v_div_fixup — spends most of
// its dozen dependent instructions on the DENOMINATOR, which here is one constant per level.  For a
// numerator v_div_scale_f64 leaves unscaled (exponent within 2^-400 .. 2^400, or zero) and such a
// denominator, the sequence reduces to q0 = n r, q = fma(fma(-c, q0, n), r, q0) with r = the refined
// reciprocal: the same instructions on the same operands, hence the same bits (march.hip does the same;
// tests/test_gpu_plane.py compares with the row kernels, which divide).  A wave that meets any other
// numerator redoes its four quotients with the division itself.
__device__ __forceinline__ double refined_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ float refined_rcp(float) { return 0.0f; }
__device__ __forceinline__ bool plain_numerator(double v) {
    const unsigned e = ((unsigned)__double2hiint(v) >> 20) & 0x7ffu;
    return e - 623u <= 800u || v == 0.0;
}
__device__ __forceinline__ bool plain_numerator(float) { return false; }
// q[i] = n[i] / c for the lanes' four numerators (every lane's: a cell outside the grid has a finite numerator like any other)
template <typename V>
__device__ __forceinline__ void quotients(const V (&n)[4], V c, V r, bool fast, V (&q)[4]) {
    if (sizeof(V) == 8 && fast) {
        bool bad = false;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const V q0 = n[i] * r;
            q[i] = madd(madd(-c, q0, n[i]), r, q0);
            bad = bad || !plain_numerator(n[i]);
        }
        if (__builtin_amdgcn_ballot_w64(bad) == 0) return;
    }
#pragma unroll
    for (int i = 0; i < 4; ++i) q[i] = n[i] / c;
}

// The in-line neighbours of a pair (h = 2q, 2q + 1) of one colour: O = the pair of the other colour
// at the same h, nb = the one value beyond it — rule 0: the LEFT thread's second value, rule 1: the
// RIGHT thread's first (rule = line parity for a red cell, its complement for a black one).
template <typename V>
struct Inline {
    V imx, ipx, imy, ipy;
};
template <typename V>
__device__ __forceinline__ Inline<V> in_line(int rule, const P2<V> &O, V nb) {
    Inline<V> r;
    r.imx = rule ? O.x : nb;
    r.ipx = rule ? O.y : O.x;
    r.imy = rule ? O.y : O.x;
    r.ipy = rule ? nb : O.y;
    return r;
}

// Wait until a flag another GPU (or another process's kernel) stores into this GPU's memory holds at least seq
// (wrapping compare): system-scope loads, a bounded number of them — a wait that gives up sets bit 0 of *status
// and lets the caller run on (its results are then wrong and the host says so) instead of hanging the device.
__device__ __forceinline__ void peer_wait(const uint32_t *flag, uint32_t seq, uint32_t *status, uint32_t spin) {
    if (!flag) return;
    for (uint32_t n = 0;; ++n) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (int32_t(v - seq) >= 0) break;
        if (n >= spin) {
            if (status) __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            break;
        }
        __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");      // what the flag's writer stored before it: not from this CU's L1 / this XCD's L2
}
// the pass's workgroups are counted at its end; the last one tells the neighbours
// (The stores into the neighbours are write-through — PEER_AUX — so "out" is their acknowledgement: a wait for the
// wave's outstanding stores, NOT a release fence at system scope, which would also write this XCD's whole L2 back —
// the pass's own 134 MB of stores — once per workgroup: measured +30 us per pass.)
__device__ __forceinline__ void peer_done(uint32_t *done, uint32_t n_wg, uint32_t *const (&flag)[2], uint32_t seq) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");         // every wave: s_waitcnt vmcnt(0)
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(done, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (before == n_wg - 1) {
            __hip_atomic_store(done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            for (int i = 0; i < 2; ++i)
                if (flag[i]) __hip_atomic_store(flag[i], seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// (one workgroup; the caller's stream then launches the pass: for neighbours that share this GPU, where a pass
// that waited itself would hold the compute units the neighbour's pass needs)
__global__ void plane_wait_kernel(const uint32_t *f0, const uint32_t *f1, const uint32_t *f2, const uint32_t *f3,
                                  uint32_t s0, uint32_t s1, uint32_t s2, uint32_t s3, uint32_t *status, uint32_t spin) {
    if (threadIdx.x == 0) {
        peer_wait(f0, s0, status, spin);
        peer_wait(f1, s1, status, spin);
        peer_wait(f2, s2, status, spin);
        peer_wait(f3, s3, status, spin);
    }
}

// MODE 0: down (sweep, residual, restriction), 1: up (prolongation, sweep, optionally the norm)
// PEER: a slab with neighbours reached by peer stores (PlanePlan::Peer).
// XZ (down): x_old is zero and is not read.
// Every load of the loop is UNCONDITIONAL (a lane that needs nothing asks for an offset behind the buffer,
// which costs no memory traffic) and is committed at the top of the NEXT step: a load inside a branch is
// followed by s_waitcnt vmcnt(0) where the branch rejoins — which also waits for the whole step's
// prefetch (measured: 4600 of a step's 12000 cycles).
// LA: how many steps ahead a step's loads are requested (2: two sets of registers in flight — the vector
// memory pipe then streams while a step computes and waits at its barrier; the loop runs two steps per
// iteration so that every register has a fixed role and the step's parity is a compile-time constant).
// MAXT: the largest workgroup the instantiation is launched with (small tiles: more registers per lane, which LA = 2 needs)
// FIRST (down): the coarse level's initial iterate is written too (zeros, or its first relaxation where a diagonal
// is given) — not needed when the coarse level's own down pass takes its iterate as zero, the usual case
// SWEEP = false: the pass without its relaxation — a cycle with preIterations = 0 (down: residual of the iterate as
// it is + restriction; x_new is not written) or postIterations = 0, the reference's default (openmg/__init__.py:22-23;
// up: x_new = x_old + R^T e and the squares of ITS residual): the same pipeline, stages B and C form no quotient
// MIRROR (up; whole grids, and slabs whose neighbours are reached by exchanges): the march runs from the LAST plane to the first — plane index and coarse plane index
// mirrored where global memory is addressed, the two k neighbours exchanged in the row chains, the in-line rule's parity
// flipped (nz is even) — so that the pass starts where the down pass before it ended and ends where the next cycle's
// down pass starts: what those passes touched last is the likeliest to be still in the memory-side cache.  A sweep of
// one colour does not depend on the order of its cells: the same bits in x; the norm's squares are added in another order.
template <typename V, int MODE, bool NORM, bool XZ, int LA, bool PEER = false, int MAXT = 512, bool FIRST = false, bool SWEEP = true, bool MIRROR = false>
__global__ __launch_bounds__(MAXT) void plane_kernel(const PlaneKArgs<V> a) {
    static_assert(!MIRROR || (MODE == 1 && !PEER && !FIRST && !XZ), "mirrored march: up passes of whole grids");
#ifdef OMG_PLANE_STAMPS
    const unsigned long long st_entry = __builtin_amdgcn_s_memtime();
#endif
    extern __shared__ __attribute__((aligned(16))) unsigned char plane_smem[];
    V *const lds = reinterpret_cast<V *>(plane_smem);
    __shared__ double s_red[8];
    __shared__ int s_step[8];            // per wave: steps finished (PLANE_WAVE_SYNC)
    if (threadIdx.x < 8) s_step[threadIdx.x] = 0;

    const int PX = a.PX, PY = a.PY;
    const int S = 2 * PX + 4;                        // LDS row: guard pair, PX pairs, guard pair
    const int BUF = (2 * PY + 2) * S;                // guard row, 2 PY lines, guard row
    const int t = int(threadIdx.x);
    const bool live = t < PX * PY;
    const int px = live ? t % PX : 0, py = live ? t / PX : 0;
    // workgroup -> (tx, ty, tz).  Blocks b and b + 8 share an XCD (observed, speed only): give an
    // XCD a contiguous run of z chunks so that the tiles sharing a ring also share an L2.
    int L = int(blockIdx.x);
    int nwg = int(gridDim.x);
    const bool edge_wg = !PEER && a.edge_wg0 > 0 && L >= a.edge_wg0;      // (uniform; decided once, before the loop)
    if (!PEER && a.edge_wg0 > 0) {
        if (edge_wg) { L -= a.edge_wg0; nwg = 1; }
        else nwg = a.edge_wg0;
    }
    if ((nwg & 7) == 0) L = (L & 7) * (nwg >> 3) + (L >> 3);
    const int nxy = a.ntx * a.nty;
    const int tz = L / nxy, rem = L - tz * nxy;
    const int ty = rem / a.ntx, tx = rem - ty * a.ntx;
    const int z0 = edge_wg ? (tz ? a.z_end - PLANE_EDGE : a.z_base) : a.zc_base + tz * a.zc_stride;
    const int z1 = edge_wg ? z0 + PLANE_EDGE : min(a.zc_end, z0 + a.zc_len);
    const int q = tx * a.TXq + px - 1;               // pair index in the line
    const int ja = ty * a.TY + 2 * py - 4;           // the thread's lines ja (even), ja + 1
    const bool vx0 = live && q >= 0 && 2 * q < a.hx, vx1 = live && q >= 0 && 2 * q + 1 < a.hx;
    const bool vl[2] = {ja >= 0 && ja < a.ny, ja + 1 >= 0 && ja + 1 < a.ny};
    const bool inner = live && px >= 1 && px <= PX - 2 && py >= 2 && py <= PY - 3;
    const int ps = a.ny * a.hx;                      // one colour's values per plane
    const int lb[2] = {ja * a.hx + 2 * q, (ja + 1) * a.hx + 2 * q};

    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.x_old), 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t bs = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.b), 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ws = __builtin_amdgcn_make_buffer_rsrc(a.x_new, 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t es = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(MODE == 1 ? a.ec : a.cdiag), 0, a.cvec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ms = __builtin_amdgcn_make_buffer_rsrc(const_cast<int32_t *>(a.cmap), 0, a.cmap_bytes, 0x00020000);

    // ---- cells outside the grid ------------------------------------------------------------------------------------
    // Nothing in the loop is masked per lane.  A thread relaxes all its cells; for a cell outside the grid that gives a
    // finite value nobody may use, so the couplings of the cells INSIDE to such cells are switched off instead: per-lane
    // copies of the four coefficients that can point outside, zero where they do (a zero coefficient times a finite
    // value adds +-0 to a chain that started from +0 — the bits of the same chain over a neighbour that holds zero).
    // The OLD values of the cells outside are exact zeros as before (loads from behind the buffer), nothing of them is
    // stored (offsets behind the buffer again), their residuals end in coarse slots that do not exist, their squares are
    // dropped at the end.  PLANES outside the grid are uniform: a scalar branch clears them after stages B and C.
    // (ja and ny are even: a thread's two lines are in the grid together)
    constexpr int W = int(sizeof(V));
    const bool vcell = vl[0] && vx0;
    V c1a = ja > 0 ? a.c1 : V(0);                    // line ja: its -J neighbour is the thread above's second line
    V c5b = ja + 2 < a.ny ? a.c5 : V(0);             // line ja + 1: its +J neighbour is the thread below's first line
    V c2x = q > 0 ? a.c2 : V(0);                     // first value under rule 0: -I is the left thread's second value
    V c4x = 2 * q + 1 < a.hx ? a.c4 : V(0);          // first value under rule 1: +I is the other colour's second value (outside in the last pair of a line of odd hx)
    V c4y = 2 * q + 2 < a.hx ? a.c4 : V(0);          // second value under rule 1: +I is the right thread's first value
    // byte offsets of the thread's pairs within a plane of one colour: loads (every cell of the grid), stores (the tile's own cells)
    unsigned lbo[2] = {vcell ? unsigned(lb[0] * W) : unsigned(OOB), vcell ? unsigned(lb[1] * W) : unsigned(OOB)};
    unsigned lso[2] = {(vcell && inner) ? unsigned(lb[0] * W) : unsigned(OOB), (vcell && inner) ? unsigned(lb[1] * W) : unsigned(OOB)};
    // (opaque from here on: kept in registers, not re-derived from the conditions inside the loop)
    asm volatile("" : "+v"(c1a), "+v"(c5b), "+v"(c2x), "+v"(c4x), "+v"(c4y));
    asm volatile("" : "+v"(lbo[0]), "+v"(lbo[1]), "+v"(lso[0]), "+v"(lso[1]));
    const bool oddx = (a.hx & 1) != 0;               // the last pair of a line holds one value: its store is a single one

    // the chain of a row: the seven slots in column order from +0, the lane's coefficient where the neighbour may be outside.
    // l: the thread's line, v: first / second value of the pair, rule: see in_line()
    auto head = [&](int l, int v, int rule, V km, V jm, V im) -> V {
        V sum = madd(a.c0, km, V(0));
        sum = madd(l == 0 ? c1a : a.c1, jm, sum);
        return madd((v == 0 && rule == 0) ? c2x : a.c2, im, sum);
    };
    auto tail = [&](int l, int v, int rule, V sum, V d, V ip, V jp, V kp) -> V {
        sum = madd(a.c3, d, sum);
        sum = madd(rule == 0 ? a.c4 : (v == 0 ? c4x : c4y), ip, sum);
        sum = madd(l == 1 ? c5b : a.c5, jp, sum);
        return madd(a.c6, kp, sum);
    };

    // a pair of the level vector: colour (0 red, 1 black), plane k (uniform), line ja + l; 0 outside the grid
    auto zphys = [&](int k) -> int { return MIRROR ? a.nz - 1 - k : k; };          // the plane a march index stands for
    // the coarse plane (index into the coarse vectors, which may carry more ghost planes than half the fine ones: kc_off)
    // under the fine planes 2 kc, 2 kc + 1 of the march
    auto cplane = [&](int kc) -> int { return MIRROR ? (a.nz / 2 - 1 + a.kc_off) - kc : kc + a.kc_off; };
    auto plane_off = [&](int colour, int k) -> unsigned {
        return (k >= a.kv0 && k < a.kv1) ? unsigned(((colour ? a.nr : 0) + zphys(k) * ps) * W) : unsigned(OOB);
    };
    // (the plane's part of the offset as ONE scalar value — the asm keeps the compiler from branching around its select)
    auto fetch = [&](const __amdgpu_buffer_rsrc_t &rs, int colour, int k, int l) -> P2<V> {
        unsigned po = plane_off(colour, k);
        asm volatile("" : "+s"(po));
        return bload2(rs, int(lbo[l] + po), V(0));
    };
    auto fetch_at = [&](const __amdgpu_buffer_rsrc_t &rs, unsigned po, int l) -> P2<V> { return bload2(rs, int(lbo[l] + po), V(0)); };
    auto fetch_x = [&](int colour, int k, int l) -> P2<V> {
        if (XZ) return {V(0), V(0)};
        return fetch(xs, colour, k, l);
    };
    // (a line of odd hx: the pair as two stores, the second one dropped where the line has ended)
    constexpr int XAUX = sizeof(V) == 8 ? PLANE_STORE_AUX : 0;
    auto put_pair = [&](const __amdgpu_buffer_rsrc_t &rs, unsigned off, const P2<V> &v) {
#ifdef OMG_PLANE_DBG_ON
        if (a.dbg & 1) return;
#endif
        if (oddx) {
            bstore1<XAUX>(rs, int(off), v.x);
            bstore1<XAUX>(rs, int(vx1 ? off + unsigned(W) : unsigned(OOB)), v.y);
        } else {
            bstore2(rs, int(off), v, true);
        }
    };
    // the coarse cells (2q, J, kc), (2q + 1, J, kc) of this thread, J = ja / 2, as BYTE offsets into the coarse vectors
    // (>= OOB: none — the cell or the plane is outside, or, going down, the thread is in the tile's ring and stores nothing)
    const bool vcoarse = vcell && (MODE == 1 || inner);
    const int cbase = (ja >> 1) * a.nxc + 2 * q;
    unsigned cb4 = vcoarse ? unsigned(cbase * 4) : unsigned(OOB);
    unsigned lmx = vcoarse ? 0u : unsigned(OOB), lmy = (vcoarse && vx1) ? 0u : unsigned(OOB);
    asm volatile("" : "+v"(cb4), "+v"(lmx), "+v"(lmy));
    const v2u none = {unsigned(OOB), unsigned(OOB)};
    // ... in two halves, so that the load has no consumer in the step that issues it: the request
    // (raw words of the slot map; nothing is fetched for a lane that needs none) and, a step later, the offsets
    auto slots_request = [&](int kc, bool want) -> v2u {
        kc = cplane(kc);
        const bool ok = want && a.cmap && kc >= 0 && kc < a.nzc;
        return __builtin_amdgcn_raw_buffer_load_b64(ms, int(cb4 + (ok ? unsigned(kc * a.nyc * a.nxc * 4) : unsigned(OOB))), 0, 0);   // (second word unused when !vx1)
    };
    auto slots_commit = [&](const v2u &m, int kc, bool want) -> v2u {
        kc = cplane(kc);
        const bool ok = want && kc >= 0 && kc < a.nzc;
        const unsigned u = ok ? 0u : unsigned(OOB), cn = ok ? unsigned(kc * a.nyc * a.nxc * W) : 0u;
        // (no slot map: the coarse level in natural order.  cb4 of a lane without coarse cells is an offset behind the
        // buffers already, and stays one under the masks)
        const unsigned s0_ = a.cmap ? m.x * unsigned(W) : cn + cb4 * unsigned(W / 4), s1_ = a.cmap ? m.y * unsigned(W) : cn + cb4 * unsigned(W / 4) + unsigned(W);
        return v2u{s0_ | lmx | u, s1_ | lmy | u};
    };
    auto coarse_slots = [&](int kc) -> v2u { return slots_commit(slots_request(kc, true), kc, true); };
    auto coarse_vals = [&](const v2u &sl) -> P2<V> { return {bload1(es, int(sl.x), V(0)), bload1(es, int(sl.y), V(0))}; };
    auto prolonged = [&](const P2<V> &x, const P2<V> &e) -> P2<V> {
        // openmg/__init__.py:214,220: x + R^T e — the product rounded, then added (ROW_SCATTER's two roundings)
        return {x.x + madd(a.w, e.x, V(0)), x.y + madd(a.w, e.y, V(0))};
    };

    const V rc3 = refined_rcp(a.c3);
    const bool fast = a.fast_div != 0;
    const int row_a = 1 + 2 * py, col = 2 + 2 * px;
    const int idx[2] = {row_a * S + col, (row_a + 1) * S + col};
    for (int i = t; i < 6 * BUF; i += int(blockDim.x)) lds[i] = V(0);
    if (((PEER && a.fused_wait) || edge_wg) && t == 0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) peer_wait(a.wait_flag[i], a.wait_seq[i], a.status, a.spin);
    }
    __syncthreads();
    // ONE scalar says what this workgroup owes the neighbours (everything else about them is fetched from the
    // kernel's arguments only in the steps that store there: the loop has no scalar registers to spare, and an
    // argument it has to re-read costs a scalar-load round trip per step): bit 0 / 1: its chunk holds the slab's
    // first / last planes and rank - 1 / rank + 1 exists; bits 2 / 3: ... and takes the coarse right-hand side
    int pside = 0;
    if (PEER) {
        const bool lo = z0 < a.z_base + 2 * PEER_CPLANES, hi = z1 > a.z_end - 2 * PEER_CPLANES;    // (a short chunk may be the second of them)
        pside = (lo && a.peer_x[0] ? 1 : 0) | (hi && a.peer_x[1] ? 2 : 0) | (lo && a.peer_bc[0] ? 4 : 0) | (hi && a.peer_bc[1] ? 8 : 0);
    }
    // plane k (uniform) of colour-offset slot `base` (+ lb[l]): to the neighbours whose ghost planes it is.  The
    // neighbour's buffer descriptor is made HERE, in the few steps that use it (the asm keeps it from being hoisted
    // out of the loop): kept live over the loop it does not fit the scalar registers any more, moves into vector
    // registers, and every store becomes a loop over the lanes' "different" descriptors (+5 us per pass).
    // A missing neighbour: an empty range, its stores are dropped.
    auto peer_store = [&](int k, int base, const P2<V> (&v)[2]) {
#pragma unroll
        for (int side = 0; side < 2; ++side) {
            const bool hit = side == 0 ? (pside & 1) && k < z0 + PEER_PLANES && k < a.z_base + PEER_PLANES
                                       : (pside & 2) && k >= z1 - PEER_PLANES && k >= a.z_end - PEER_PLANES;
            if (!hit) continue;
            V *px = a.peer_x[side];
            asm volatile("" : "+s"(px));
            const __amdgpu_buffer_rsrc_t pw = __builtin_amdgcn_make_buffer_rsrc(px ? px : a.x_new, 0, px ? a.vec_bytes : 0u, 0x00020000);
            int sh = side == 0 ? a.peer_shift : -a.peer_shift;
            asm volatile("" : "+s"(sh));                            // (... nor the addresses computed ahead of the branch)
#pragma unroll
            for (int l = 0; l < 2; ++l)
                if (vl[l] && vx0) bstore2<PEER_AUX>(pw, (base + sh + lb[l]) * int(sizeof(V)), v[l], vx1);
        }
    };
    int CO_side = 0;                 // the pending coarse pair is also rank - 1's (1) / rank + 1's (2) ghost
    // (the coarse right-hand side through a descriptor too: a slot that is none is an offset behind it, no branch)
    const __amdgpu_buffer_rsrc_t bcs = __builtin_amdgcn_make_buffer_rsrc(a.bc, 0, a.bc ? a.cvec_bytes : 0u, 0x00020000);
    auto coarse_store = [&](const v2u &sl, const P2<V> &co, const P2<V> &cx, auto PB) {
        bstore1<0>(bcs, int(sl.x), co.x);
        bstore1<0>(bcs, int(sl.y), co.y);
        if (FIRST && a.xc) {
            const __amdgpu_buffer_rsrc_t xcs = __builtin_amdgcn_make_buffer_rsrc(a.xc, 0, a.cvec_bytes, 0x00020000);
            bstore1<0>(xcs, int(sl.x), cx.x);
            bstore1<0>(xcs, int(sl.y), cx.y);
        }
        if (PEER && decltype(PB)::value) {
#pragma unroll
            for (int side = 0; side < 2; ++side) {
                if (!(CO_side & (1 << side))) continue;
                // (pointer and shift pinned inside the branch, as in peer_store: nothing of it is computed ahead)
                V *pb = a.peer_bc[side];
                int sh = side == 0 ? a.peer_cshift : -a.peer_cshift;
                asm volatile("" : "+s"(pb), "+s"(sh));
                // (through a buffer descriptor with 32-bit offsets, like the planes of x: 64-bit addresses per value
                // cost this kernel forty registers)
                const __amdgpu_buffer_rsrc_t pwb = __builtin_amdgcn_make_buffer_rsrc(pb, 0, a.cvec_bytes, 0x00020000);
                bstore1<PEER_AUX>(pwb, sl.x < unsigned(OOB) ? int(sl.x) + sh * W : OOB, co.x);
                bstore1<PEER_AUX>(pwb, sl.y < unsigned(OOB) ? int(sl.y) + sh * W : OOB, co.y);
            }
        }
    };

    const P2<V> zero2 = {V(0), V(0)};
    P2<V> XR[3][2], XB[5][2], BR[3][2], BB[2], LXB[LA][2], LXR[LA][2], LBR[LA][2], LBB[LA][2], RB[2];
#pragma unroll
    for (int l = 0; l < 2; ++l) {
#pragma unroll
        for (int p = 0; p < 3; ++p) { XR[p][l] = zero2; BR[p][l] = zero2; }
#pragma unroll
        for (int p = 0; p < 5; ++p) XB[p][l] = zero2;
        BB[l] = zero2;
        RB[l] = zero2;
    }
    P2<V> ACC = zero2;               // down: the restriction's running chain
    P2<V> CO = zero2, CX = zero2;    // ... a finished coarse pair waiting for its store: right-hand side, first iterate
    v2u SLd = none, SLo = none;
    v2u SLt = {0u, 0u};              // SLt, DGt, Et: requested in one step, committed at the top of the next
    int SLt_kc = -1;
    P2<V> DG = zero2, DGt = zero2;   // ... the coarse diagonal at those slots
    P2<V> E0 = zero2, E1n = zero2, Et = zero2;   // up: coarse correction of coarse planes s >> 1 and (s >> 1) + 1
    v2u SLn = none;
    double sq = 0.0, sqy = 0.0;      // squares of the pairs' first / second values (every lane's; the lanes that count are picked at the end)

    // ---- prologue: the state step s0 = z0 - 2 expects ---------------------------------------------
    const int s0 = z0 - 2;
    {
        P2<V> em = zero2;
        if (MODE == 1) {
            const v2u sl0 = coarse_slots(s0 >> 1), slm = coarse_slots((s0 >> 1) - 1);
            SLt_kc = (s0 >> 1) + 1;
            SLt = slots_request(SLt_kc, true);        // committed to SLn at the top of the first step
            E0 = coarse_vals(sl0);
            em = coarse_vals(slm);
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            XB[0][l] = fetch_x(1, s0, l);            // becomes XB[1] at the first shift
            XB[1][l] = fetch_x(1, s0 - 1, l);        // becomes XB[2]
            if (MODE == 1) {
                XB[0][l] = prolonged(XB[0][l], E0);
                XB[1][l] = prolonged(XB[1][l], em);
            }
#pragma unroll
            for (int d = 0; d < LA; ++d) {            // what steps s0 (, s0 + 1) take at their top
                LXB[d][l] = fetch_x(1, s0 + d + 1, l);
                LXR[d][l] = fetch_x(0, s0 + d, l);
                LBR[d][l] = fetch(bs, 0, s0 + d, l);
                LBB[d][l] = fetch(bs, 1, s0 + d - 1, l);
            }
            if (live) lds_put(lds + (0 * 2 + (s0 & 1)) * BUF + idx[l], XB[0][l]);
        }
    }
    __syncthreads();

#ifdef OMG_PLANE_STAMPS
    unsigned long long st_mem = 0, st_cmp = 0, st_bar = 0, st_top = 0, st_B = 0, st_C = 0, st_t = __builtin_amdgcn_s_memtime();
#define PLANE_STAMP(acc) { const unsigned long long now_ = __builtin_amdgcn_s_memtime(); acc += now_ - st_t; st_t = now_; }
#else
#define PLANE_STAMP(acc)
#endif
    // one step; PARC: the parity of s as a type (s0 is even)
    // PB (PEER kernels): the step may hold planes the neighbours take — only those steps carry the peer stores;
    // the others are the code of a pass without neighbours
    static_assert(!(PEER && LA == 2), "the two-step loop has no boundary steps");
    // The waves of a workgroup, step by step.  A wave reads only LDS cells written by threads t +- 1 and t +- PX, i.e. by
    // the waves within (PX + 63) / 64 of it: instead of a barrier of the whole workgroup — where eight waves waited for the
    // slowest at every step — it PUBLISHES its progress once it has written the next step's images and needs nothing of
    // this step's any more (stage D's operands are taken first), and AWAITS those neighbours' progress only where the
    // next step first reads an image: stage D, the restriction and the next step's top (loads taken, requests issued)
    // lie between the two, so a neighbour that is a little behind costs nothing.  The images are double buffered: one
    // count per step covers both hazards — what a neighbour wrote for step s + 1 is there, and what it read for step s it
    // has read.  (double only: the fp32 passes were 10 % slower with it than with the barrier.)
#ifndef PLANE_WSYNC_F32
#define PLANE_WSYNC_F32 0
#endif
    constexpr bool WSYNC = PLANE_WAVE_SYNC && (sizeof(V) == 8 || PLANE_WSYNC_F32);
    int n_waves = int(blockDim.x) >> 6;
    asm volatile("" : "+s"(n_waves));                    // (not re-read from the dispatch packet in every step)
    const int my_wave = t >> 6, reach = (PX + 63) >> 6;
    auto publish = [&](int cnt) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if ((t & 63) == 0) __hip_atomic_store(&s_step[my_wave], cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
    };
    auto await = [&](int cnt) {
        if ((t & 63) == 0) {
            // a bounded wait, like the peer waits and the wavefront sweep's polls: a wave that never arrives (it
            // cannot, short of a fault) sets bit 1 of *status — the host raises — instead of hanging the queue
            for (int o = max(0, my_wave - reach); o <= min(n_waves - 1, my_wave + reach); ++o)
                for (unsigned polls = 0; __hip_atomic_load(&s_step[o], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP) < cnt; ++polls) {
                    if (polls >= WAVE_SYNC_SPIN) {
                        if (a.status) __hip_atomic_fetch_or(a.status, 2u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        break;
                    }
                    __builtin_amdgcn_s_sleep(1);
                }
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
    };
    auto step = [&](auto PARC, const int s, auto PB) {
        constexpr bool PEER_STEP = PEER && decltype(PB)::value;
        const int par = PARC;                            // (a compile-time constant after inlining when LA == 2)
        const int set = LA == 2 ? par : 0;               // the registers this step's loads arrive in, and its requests go to
        // The step's loads are taken HERE, behind the barrier — not where the compiler would sink the copies
        // into the state registers (the bottom of the previous step, in front of the barrier): a whole step
        // to arrive instead of the step's arithmetic only.
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            asm volatile("" : "+v"(LXB[set][l].x), "+v"(LXB[set][l].y), "+v"(LXR[set][l].x), "+v"(LXR[set][l].y));
            asm volatile("" : "+v"(LBR[set][l].x), "+v"(LBR[set][l].y), "+v"(LBB[set][l].x), "+v"(LBB[set][l].y));
        }
#ifdef OMG_PLANE_STAMPS
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        PLANE_STAMP(st_mem)
#endif
        // ---- top: take what has arrived, request what the next step needs, store what is final ----
        if (MODE == 1) {
            // commit what the previous step requested: an even step asked for the next coarse plane's values,
            // an odd one for the slots of the one after
            E1n.x = par ? Et.x : E1n.x; E1n.y = par ? Et.y : E1n.y;
            const v2u sl = slots_commit(SLt, SLt_kc, true);
            SLn.x = par ? SLn.x : sl.x; SLn.y = par ? SLn.y : sl.y;
        } else {
            // an odd step asked for the slots of the coarse plane finished two steps later, the even step
            // after it for the coarse diagonal there
            const v2u sl = slots_commit(SLt, SLt_kc, true);
            SLd.x = par ? SLd.x : sl.x; SLd.y = par ? SLd.y : sl.y;
            if (FIRST) { DG.x = par ? DGt.x : DG.x; DG.y = par ? DGt.y : DG.y; }
        }
#pragma unroll
        for (int l = 0; l < 2; ++l) {
            XB[4][l] = XB[3][l]; XB[3][l] = XB[2][l]; XB[2][l] = XB[1][l]; XB[1][l] = XB[0][l];
            XR[2][l] = XR[1][l]; XR[1][l] = XR[0][l];
            BR[2][l] = BR[1][l]; BR[1][l] = BR[0][l];
            if (MODE == 1) {
                XB[0][l] = prolonged(LXB[set][l], par ? E1n : E0);            // plane s + 1: coarse plane (s + 1) >> 1
                XR[0][l] = prolonged(LXR[set][l], E0);                        // plane s
            } else {
                XB[0][l] = LXB[set][l];
                XR[0][l] = LXR[set][l];
            }
            BR[0][l] = LBR[set][l];
            BB[l] = LBB[set][l];
        }
        if (PEER_STEP && inner && (((pside & 1) && s < z0 + PEER_PLANES + 2) || ((pside & 2) && s >= z1 - PEER_PLANES + 1))) {
            // ONE uniform branch per step, taken by the few steps that hold a neighbour's ghost planes: red of plane
            // s - 1 and black of plane s - 2, final since the previous step.  HERE — the last step's loads are consumed,
            // this step's not yet requested — the fewest registers are live (at the step's end the branch cost the
            // kernels with neighbours their two-steps-per-iteration form).
            if (s - 1 >= z0 && s - 1 < z1) peer_store(s - 1, (s - 1) * ps, XR[1]);
            if (s - 2 >= z0 && s - 2 < z1) peer_store(s - 2, a.nr + (s - 2) * ps, XB[3]);
        }
        if (MODE == 1) {
            E0.x = par ? E1n.x : E0.x; E0.y = par ? E1n.y : E0.y;
            SLt_kc = (s >> 1) + 2;
            SLt = slots_request(SLt_kc, par != 0);
            Et = coarse_vals(par ? none : SLn);                   // coarse plane (s >> 1) + 1, slots committed above
        } else {
            SLt_kc = (s - 1) >> 1;                                // the coarse plane finished by step s + 2
            SLt = slots_request(SLt_kc, par != 0);
            if (FIRST) DGt = coarse_vals((par || !a.cdiag) ? none : SLd);
        }
        {
            unsigned pXB = plane_off(1, s + LA + 1), pR = plane_off(0, s + LA), pBB = plane_off(1, s + LA - 1);
            asm volatile("" : "+s"(pXB), "+s"(pR), "+s"(pBB));
#ifdef OMG_PLANE_DBG_ON
            if (a.dbg & 16) {                                     // (timing experiment: the chunk's fill planes — those of its neighbours in z — not loaded)
                if (s + LA + 1 < z0 || s + LA + 1 >= z1) pXB = unsigned(OOB);
                if (s + LA < z0 || s + LA >= z1) pR = unsigned(OOB);
                if (s + LA - 1 < z0 || s + LA - 1 >= z1) pBB = unsigned(OOB);
            }
            if (a.dbg & 4) { pXB = unsigned(OOB); }               // (timing experiments: no loads of x / of b — wrong results)
            unsigned pR_b = pR;
            if (a.dbg & 4) pR = unsigned(OOB);
            if (a.dbg & 8) { pR_b = unsigned(OOB); pBB = unsigned(OOB); }
#else
            const unsigned pR_b = pR;
#endif
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                LXB[set][l] = XZ ? zero2 : fetch_at(xs, pXB, l);
                LXR[set][l] = XZ ? zero2 : fetch_at(xs, pR, l);
                LBR[set][l] = fetch_at(bs, pR_b, l);
                LBB[set][l] = fetch_at(bs, pBB, l);
            }
        }
        // the coarse pair the previous (odd) step finished (uniform)
        if (MODE == 0 && !par && s - 3 >= z0 && s - 3 < z1) {
#ifdef OMG_PLANE_DBG_ON
            if (!(a.dbg & 2))
#endif
            coarse_store(SLo, CO, CX, PB);
        }

        PLANE_STAMP(st_top)
        // the images this step reads are complete, and the neighbours have read the ones it overwrites (see `publish` below)
        if (WSYNC) await(s - s0);
        PLANE_STAMP(st_bar)
#ifdef PLANE_SETPRIO
        __builtin_amdgcn_s_setprio(PLANE_SETPRIO);      // (experiment, VERDICT r4 item 5 (ii): the wave that computes ahead of the wave that waits)
#endif
        P2<V> d_jm = zero2, d_jp = zero2;                         // stage D's values from the images, taken before they are handed over
        V d_nb[2] = {V(0), V(0)};
        P2<V> rb[2] = {zero2, zero2};                             // stage C's residuals (black rows of plane s - 1)
        if (live) {
            const V *const E1 = lds + (0 * 2 + par) * BUF;        // black, old, plane s
            const V *const E2 = lds + (1 * 2 + par) * BUF;        // red, new, plane s - 1
            const V *const E3 = lds + (2 * 2 + par) * BUF;        // black, new, plane s - 2
            const bool pvB = s >= a.kv0 && s < a.kv1, pvC = s - 1 >= a.kv0 && s - 1 < a.kv1;
            // B: red sweep of plane s
            if (SWEEP) {
                const P2<V> jm = lds_pair(E1 + idx[0] - S), jp = lds_pair(E1 + idx[1] + S);
                const P2<V> o0 = XB[1][0], o1 = XB[1][1];
                V num[4], quo[4];
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const int rule = (l + par + (MIRROR ? 1 : 0)) & 1;    // (par, not s: a constant in the two-steps form)
                    const P2<V> O = l ? o1 : o0, Ojm = l ? o0 : jm, Ojp = l ? jp : o1;
                    const Inline<V> n = in_line(rule, O, E1[idx[l] + (rule ? 2 : -1)]);
                    const P2<V> D = XR[0][l], Bv = BR[0][l], Km = XB[MIRROR ? 0 : 2][l], Kp = XB[MIRROR ? 2 : 0][l];
                    num[2 * l] = Bv.x - tail(l, 0, rule, head(l, 0, rule, Km.x, Ojm.x, n.imx), D.x, n.ipx, Ojp.x, Kp.x);
                    num[2 * l + 1] = Bv.y - tail(l, 1, rule, head(l, 1, rule, Km.y, Ojm.y, n.imy), D.y, n.ipy, Ojp.y, Kp.y);
                }
                quotients(num, a.c3, rc3, fast, quo);
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    // openmg/solvers.py:68   x[i] = x[i] + (b[i] - Aix) / A[i, i]
                    XR[0][l].x = XR[0][l].x + quo[2 * l];
                    XR[0][l].y = XR[0][l].y + quo[2 * l + 1];
                }
                if (!pvB) {                                       // (uniform: a plane outside the grid stays zero)
                    asm volatile("" ::: "memory");
                    XR[0][0] = zero2; XR[0][1] = zero2;
                }
            }
            if ((SWEEP || MODE == 1) && s - 1 >= z0 && s - 1 < z1) {
                // red of plane s - 1 became final in the previous step
                const unsigned base = unsigned(zphys(s - 1) * ps * W);
#pragma unroll
                for (int l = 0; l < 2; ++l) put_pair(ws, lso[l] + base, XR[1][l]);
            }
            PLANE_STAMP(st_B)
            // C: black sweep of plane s - 1, and the residual of the rows it has just relaxed
            {
                const P2<V> jm = lds_pair(E2 + idx[0] - S), jp = lds_pair(E2 + idx[1] + S);
                const P2<V> o0 = XR[1][0], o1 = XR[1][1];
                V num[4], quo[4], hd[4];
                Inline<V> nl[2];
                P2<V> Jp[2];
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const int rule = (l + par + (MIRROR ? 1 : 0)) & 1;    // (par, not s: a constant in the two-steps form)
                    const P2<V> O = l ? o1 : o0, Ojm = l ? o0 : jm, Ojp = l ? jp : o1;
                    nl[l] = in_line(rule, O, E2[idx[l] + (rule ? 2 : -1)]);
                    Jp[l] = Ojp;
                    const P2<V> D = XB[2][l], Bv = BB[l], Km = XR[MIRROR ? 0 : 2][l], Kp = XR[MIRROR ? 2 : 0][l];
                    hd[2 * l] = head(l, 0, rule, Km.x, Ojm.x, nl[l].imx);
                    hd[2 * l + 1] = head(l, 1, rule, Km.y, Ojm.y, nl[l].imy);
                    num[2 * l] = Bv.x - tail(l, 0, rule, hd[2 * l], D.x, nl[l].ipx, Ojp.x, Kp.x);
                    num[2 * l + 1] = Bv.y - tail(l, 1, rule, hd[2 * l + 1], D.y, nl[l].ipy, Ojp.y, Kp.y);
                }
                if (SWEEP) quotients(num, a.c3, rc3, fast, quo);
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    if (!SWEEP) {
                        // the row's residual with the iterate as it is: the chain above IS a residual pass's
                        rb[l].x = num[2 * l];
                        rb[l].y = num[2 * l + 1];
                        continue;
                    }
                    const int rule = (l + par + (MIRROR ? 1 : 0)) & 1;    // (par, not s: a constant in the two-steps form)
                    const P2<V> Bv = BB[l], Kp = XR[MIRROR ? 2 : 0][l];
                    const V nx_ = XB[2][l].x + quo[2 * l];
                    const V ny_ = XB[2][l].y + quo[2 * l + 1];
                    // the same chain with the new x_i: what a residual pass over the updated vector computes
                    rb[l].x = Bv.x - tail(l, 0, rule, hd[2 * l], nx_, nl[l].ipx, Jp[l].x, Kp.x);
                    rb[l].y = Bv.y - tail(l, 1, rule, hd[2 * l + 1], ny_, nl[l].ipy, Jp[l].y, Kp.y);
                    XB[2][l].x = nx_;
                    XB[2][l].y = ny_;
                }
                if (!pvC) {                                       // (uniform)
                    asm volatile("" ::: "memory");
                    XB[2][0] = zero2; XB[2][1] = zero2;
                    rb[0] = zero2; rb[1] = zero2;
                }
            }
            if ((SWEEP || MODE == 1) && s - 2 >= z0 && s - 2 < z1) {
                // black of plane s - 2 became final in the previous step
                const unsigned base = unsigned((a.nr + zphys(s - 2) * ps) * W);
#pragma unroll
                for (int l = 0; l < 2; ++l) put_pair(ws, lso[l] + base, XB[3][l]);
            }
            PLANE_STAMP(st_C)
            // D's operands from the images, then the images the next step reads: after that this wave needs nothing of
            // the old images any more and says so; the residuals and the restriction run while the neighbours catch up
#pragma unroll
            for (int l = 0; l < 2; ++l) d_nb[l] = E3[idx[l] + (((l + par + (MIRROR ? 1 : 0)) & 1) ? 2 : -1)];
            d_jm = lds_pair(E3 + idx[0] - S);
            d_jp = lds_pair(E3 + idx[1] + S);
            V *const W1 = lds + (0 * 2 + (par ^ 1)) * BUF;
            V *const W2 = lds + (1 * 2 + (par ^ 1)) * BUF;
            V *const W3 = lds + (2 * 2 + (par ^ 1)) * BUF;
#pragma unroll
            for (int l = 0; l < 2; ++l) {
                lds_put(W1 + idx[l], XB[0][l]);
                lds_put(W2 + idx[l], XR[0][l]);
                lds_put(W3 + idx[l], XB[2][l]);
            }
        }
        PLANE_STAMP(st_cmp)
#ifdef PLANE_SETPRIO
        __builtin_amdgcn_s_setprio(0);
#endif
        if (WSYNC) publish(s - s0 + 1);
        else __syncthreads();
        PLANE_STAMP(st_bar)
        if (live) {
            // D: residual of the red rows of plane s - 2 (of a plane outside the grid: a finite value that ends in no coarse
            // slot and no norm)
            P2<V> rr[2];
            {
                const P2<V> jm = d_jm, jp = d_jp;
                const P2<V> o0 = XB[3][0], o1 = XB[3][1];
#pragma unroll
                for (int l = 0; l < 2; ++l) {
                    const int rule = (l + par + (MIRROR ? 1 : 0)) & 1;    // (par, not s: a constant in the two-steps form)
                    const P2<V> O = l ? o1 : o0, Ojm = l ? o0 : jm, Ojp = l ? jp : o1;
                    const Inline<V> n = in_line(rule, O, d_nb[l]);
                    const P2<V> D = XR[2][l], Bv = BR[2][l], Km = XB[MIRROR ? 2 : 4][l], Kp = XB[MIRROR ? 4 : 2][l];
                    rr[l].x = Bv.x - tail(l, 0, rule, head(l, 0, rule, Km.x, Ojm.x, n.imx), D.x, n.ipx, Ojp.x, Kp.x);
                    rr[l].y = Bv.y - tail(l, 1, rule, head(l, 1, rule, Km.y, Ojm.y, n.imy), D.y, n.ipy, Ojp.y, Kp.y);
                }
            }
            if (MODE == 0) {
                // openmg/__init__.py:210: the coarse cell's eight fine residuals in column order — plane
                // 2K: (line a: red, black), (line b: black, red); plane 2K + 1: the colours swapped.  RB holds
                // the black residuals of plane s - 2 (formed one step ago).
                if (!par) {
                    ACC.x = madd(a.w, rr[0].x, V(0)); ACC.y = madd(a.w, rr[0].y, V(0));
                    ACC.x = madd(a.w, RB[0].x, ACC.x); ACC.y = madd(a.w, RB[0].y, ACC.y);
                    ACC.x = madd(a.w, RB[1].x, ACC.x); ACC.y = madd(a.w, RB[1].y, ACC.y);
                    ACC.x = madd(a.w, rr[1].x, ACC.x); ACC.y = madd(a.w, rr[1].y, ACC.y);
                } else {
                    ACC.x = madd(a.w, RB[0].x, ACC.x); ACC.y = madd(a.w, RB[0].y, ACC.y);
                    ACC.x = madd(a.w, rr[0].x, ACC.x); ACC.y = madd(a.w, rr[0].y, ACC.y);
                    ACC.x = madd(a.w, rr[1].x, ACC.x); ACC.y = madd(a.w, rr[1].y, ACC.y);
                    ACC.x = madd(a.w, RB[1].x, ACC.x); ACC.y = madd(a.w, RB[1].y, ACC.y);
                    if (s - 2 >= z0 && s - 2 < z1) {              // (uniform; the slots of a thread in the ring are none)
                        CO = ACC;
                        SLo = SLd;
                        if (PEER) CO_side = (((pside & 4) && s - 2 < z0 + 2 * PEER_CPLANES && s - 2 < a.z_base + 2 * PEER_CPLANES) ? 1 : 0) |
                                            (((pside & 8) && s - 2 >= z1 - 2 * PEER_CPLANES && s - 2 >= a.z_end - 2 * PEER_CPLANES) ? 2 : 0);
                        // the coarse level's first relaxation of a zero iterate, spelled like row_epilogue's
                        if (FIRST) {
                            CX.x = (a.cdiag && SLd.x < unsigned(a.first_end) * unsigned(W)) ? V(0) + (ACC.x - V(0)) / DG.x : V(0);
                            CX.y = (a.cdiag && SLd.y < unsigned(a.first_end) * unsigned(W)) ? V(0) + (ACC.y - V(0)) / DG.y : V(0);
                        }
                    }
                }
                RB[0] = rb[0];
                RB[1] = rb[1];
            }
            if (NORM) {
                if (s - 1 >= z0 && s - 1 < z1) {
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        sq = fma(double(rb[l].x), double(rb[l].x), sq);
                        sqy = fma(double(rb[l].y), double(rb[l].y), sqy);
                    }
                }
                if (s - 2 >= z0 && s - 2 < z1) {
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        sq = fma(double(rr[l].x), double(rr[l].x), sq);
                        sqy = fma(double(rr[l].y), double(rr[l].y), sqy);
                    }
                }
            }
        }
        PLANE_STAMP(st_cmp)
    };
    if (LA == 2) {
        for (int s = s0; s <= z1 + 1; s += 2) {          // s0 even, z1 + 1 odd: whole pairs of steps
            step(std::integral_constant<int, 0>(), s, std::false_type());
            step(std::integral_constant<int, 1>(), s + 1, std::false_type());
        }
    } else {
        // (with neighbours, the finest level's two passes run out of registers that way: 16-68 bytes of scratch per lane)
        constexpr bool PAIRS = !FIRST;
        if (PAIRS) {
            // two steps per iteration: the step's parity — which neighbour an in-line pair takes, which coarse plane
            // a fine one lies over — is a compile-time constant in each copy (s0 is even, the step count too): 10 us of
            // 117 / 124 for the fine level's down / up pass.
            for (int s = s0; s <= z1 + 1; s += 2) {
                step(std::integral_constant<int, 0>(), s, std::integral_constant<bool, PEER>());
                step(std::integral_constant<int, 1>(), s + 1, std::integral_constant<bool, PEER>());
            }
        } else {
            for (int s = s0; s <= z1 + 1; ++s) step(s & 1, s, std::integral_constant<bool, PEER>());
        }
    }
#ifdef OMG_PLANE_STAMPS
    if ((t & 63) == 0) {
        unsigned long long *o = a.stamps + (size_t(blockIdx.x) * 8 + (t >> 6)) * 8;
        o[0] = st_mem; o[1] = st_cmp; o[2] = st_bar; o[3] = unsigned(z1 + 2 - s0); o[4] = st_top; o[5] = st_B; o[6] = st_C;
        o[7] = __builtin_amdgcn_s_memtime() - st_entry;
    }
#endif
    if (MODE == 0 && z1 > z0) coarse_store(SLo, CO, CX, std::true_type());       // (the last step, z1 + 1, finished the chunk's last coarse plane)
    if (NORM) {
        // the squares of the tile's own cells inside the grid; fixed order: lanes of a wave (shuffle tree), then the waves in turn
        sq = (inner && vcell) ? sq + (vx1 ? sqy : 0.0) : 0.0;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if ((t & 63) == 0) s_red[t >> 6] = sq;
        __syncthreads();
        if (t == 0) {
            double tot = 0.0;
            for (int wv = 0; wv < int(blockDim.x) >> 6; ++wv) tot += s_red[wv];
            a.partials[a.part_slot0 + int(blockIdx.x)] = tot;
        }
    }
    if (PEER) peer_done(a.done, gridDim.x, a.peer_flag, a.flag_seq);
}

// ---- y = A x of a plane level, matrix-free (the "fine-grid SpMV" of BASELINE's metric on the default path) -----------
// A thread forms one PAIR (h = 2q, 2q + 1) of one colour of one grid line: the row kernels' chain over the seven slots in
// column order from +0, a neighbour outside the grid a zero value — the bits of rows_*_kernel<ROW_SPMV> on the same
// operator.  Six 16-byte loads and one store per pair; x is read once from HBM (the neighbours come out of L2).
template <typename V>
struct PlaneSpmvArgs {
    const V *x;
    V *y;
    unsigned vec_bytes;
    int nr, hx, ny, nz, hq;              // first black slot; pairs per half line = (hx + 1) / 2
    V c0, c1, c2, c3, c4, c5, c6;
};
template <typename V>
__global__ __launch_bounds__(256) void plane_spmv_kernel(const PlaneSpmvArgs<V> a) {
    const int64_t t = int64_t(blockIdx.x) * 256 + threadIdx.x;
    const int64_t per_colour = int64_t(a.hq) * a.ny * a.nz;
    if (t >= 2 * per_colour) return;
    const int colour = t >= per_colour ? 1 : 0;
    const int64_t u = t - colour * per_colour;
    const int q = int(u % a.hq);
    const int line = int(u / a.hq), j = line % a.ny, k = line / a.ny;
    const int h = 2 * q;
    const bool two = h + 1 < a.hx;
    const int p = (j + k) & 1;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.x), 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.vec_bytes, 0x00020000);
    const int same = colour ? a.nr : 0, other = colour ? 0 : a.nr;
    const int lb = line * a.hx + h;                              // the pair's place in a colour's run
    const int ps = a.ny * a.hx;
    auto pair_at = [&](int base, bool ok) -> P2<V> {
        P2<V> r = bload2(xs, ok ? (base + lb) * int(sizeof(V)) : OOB, V(0));
        if (!two) r.y = V(0);
        return r;
    };
    const P2<V> D = pair_at(same, true);
    const P2<V> Km = pair_at(other - ps, k > 0), Kp = pair_at(other + ps, k + 1 < a.nz);
    const P2<V> Jm = pair_at(other - a.hx, j > 0), Jp = pair_at(other + a.hx, j + 1 < a.ny);
    const P2<V> O = pair_at(other, true);
    // the in-line neighbours (plane.hip, top): rule 0: i - 1 at h - 1, i + 1 at h of the other colour; rule 1: at h, h + 1
    const int rule = colour ? 1 - p : p;
    const int nbh = rule ? h + 2 : h - 1;
    const V nb = bload1(xs, (nbh >= 0 && nbh < a.hx) ? (other + line * a.hx + nbh) * int(sizeof(V)) : OOB, V(0));
    const Inline<V> n = in_line(rule, O, nb);
    PlaneKArgs<V> c;                                              // (chain_head / chain_tail take the coefficients from there)
    c.c0 = a.c0; c.c1 = a.c1; c.c2 = a.c2; c.c3 = a.c3; c.c4 = a.c4; c.c5 = a.c5; c.c6 = a.c6;
    P2<V> out;
    out.x = chain_tail(c, chain_head(c, Km.x, Jm.x, n.imx), D.x, n.ipx, Jp.x, Kp.x);
    out.y = chain_tail(c, chain_head(c, Km.y, Jm.y, n.imy), D.y, n.ipy, Jp.y, Kp.y);
    bstore2<0>(ys, (same + lb) * int(sizeof(V)), out, two);
}

// The same product, marching along z (round 5).  plane_spmv_kernel above asks the L2 for every operand of every row: 6.5
// loads per output pair, the k neighbours a plane apart — 4.0 TB/s of needed bytes, 0.51 of the HBM peak, bound by requests
// not by HBM.  Here a thread owns ONE position (line j, pair q) of BOTH colours and walks a chunk of planes with the three
// planes it needs of them in registers: the k neighbours, the same-position operand and the row's own value are its own
// registers (each loaded once per chunk), only the j neighbours and the one in-line value beyond the pair come from the
// cache (the neighbouring threads' own loads of the same step: L1 / L2 hits) — 3.5 loads per output pair.  Workgroups with
// consecutive numbers inside an XCD take neighbouring line groups of one z chunk, so those hits stay in that XCD's L2.
// Same chain, same operands: the bits of plane_spmv_kernel.
template <typename V>
struct PlaneSpmvMarchArgs {
    PlaneSpmvArgs<V> s;
    int qt, lpw, njg, nqt, lz;           // pairs per workgroup row, lines per workgroup, line groups, q tiles, planes per chunk
};
template <typename V>
__global__ __launch_bounds__(512) void plane_spmv_march_kernel(const PlaneSpmvMarchArgs<V> m) {
    const PlaneSpmvArgs<V> &a = m.s;
    int L = int(blockIdx.x);
    const int nwg = int(gridDim.x);
    if ((nwg & 7) == 0) L = (L & 7) * (nwg >> 3) + (L >> 3);
    const int per_chunk = m.njg * m.nqt;
    const int zc = L / per_chunk, rem = L - zc * per_chunk;
    const int jg = rem / m.nqt, qg = rem - jg * m.nqt;
    const int t = int(threadIdx.x);
    const int q = qg * m.qt + t % m.qt, j = jg * m.lpw + t / m.qt;
    const bool live = t < m.qt * m.lpw && q < a.hq && j < a.ny;
    const int h = 2 * q;
    const bool two = h + 1 < a.hx;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.x), 0, a.vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t ys = __builtin_amdgcn_make_buffer_rsrc(a.y, 0, a.vec_bytes, 0x00020000);
    const int ps = a.ny * a.hx;
    const int W = int(sizeof(V));
    const int lb = j * a.hx + h;                                   // the pair's place in a plane of one colour
    auto pair_at = [&](int colour, int k, int dj) -> P2<V> {
        const bool ok = live && k >= 0 && k < a.nz && j + dj >= 0 && j + dj < a.ny;
        P2<V> r = bload2(xs, ok ? ((colour ? a.nr : 0) + k * ps + lb + dj * a.hx) * W : OOB, V(0));
        if (!two) r.y = V(0);
        return r;
    };
    auto one_at = [&](int colour, int k, int hh) -> V {
        const bool ok = live && hh >= 0 && hh < a.hx;
        return bload1(xs, ok ? ((colour ? a.nr : 0) + k * ps + j * a.hx + hh) * W : OOB, V(0));
    };
    PlaneKArgs<V> c;                                              // (chain_head / chain_tail take the coefficients from there)
    c.c0 = a.c0; c.c1 = a.c1; c.c2 = a.c2; c.c3 = a.c3; c.c4 = a.c4; c.c5 = a.c5; c.c6 = a.c6;
    const int z0 = zc * m.lz, z1 = min(a.nz, z0 + m.lz);
    P2<V> X[2][3];                                                // [colour][k - 1, k, k + 1]
#pragma unroll
    for (int col = 0; col < 2; ++col) { X[col][0] = pair_at(col, z0 - 1, 0); X[col][1] = pair_at(col, z0, 0); }
    for (int k = z0; k < z1; ++k) {
        P2<V> Jm[2], Jp[2];
        V nb[2];
        const int p = (j + k) & 1;
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            X[col][2] = pair_at(col, k + 1, 0);
            Jm[col] = pair_at(col, k, -1);
            Jp[col] = pair_at(col, k, 1);
            // the value beyond the pair that the OTHER colour's row takes from this colour: rule of that row
            const int rule = col ? p : 1 - p;                      // (row colour 1 - col: rule = (1 - col) ? 1 - p : p)
            nb[col] = one_at(col, k, rule ? h + 2 : h - 1);
        }
#pragma unroll
        for (int col = 0; col < 2; ++col) {
            const int o = 1 - col;
            const int rule = col ? 1 - p : p;
            const Inline<V> n = in_line(rule, X[o][1], nb[o]);
            P2<V> out;
            out.x = chain_tail(c, chain_head(c, X[o][0].x, Jm[o].x, n.imx), X[col][1].x, n.ipx, Jp[o].x, X[o][2].x);
            out.y = chain_tail(c, chain_head(c, X[o][0].y, Jm[o].y, n.imy), X[col][1].y, n.ipy, Jp[o].y, X[o][2].y);
            if (live) bstore2<0>(ys, ((col ? a.nr : 0) + k * ps + lb) * W, out, two);
        }
#pragma unroll
        for (int col = 0; col < 2; ++col) { X[col][0] = X[col][1]; X[col][1] = X[col][2]; }
    }
}

// ---- small levels: a block of the grid per workgroup, whole in LDS ----------------------------------------------
// A level of <= 64^3 cells is pure latency for the marching kernel above: LZ + 4 = 6 dependent steps of one wave each
// (load round trip, red chain -> black chain -> residual chain, barrier), 12-15 us per pass whatever the size
// (profiles/r03_small_levels.txt).  Here a workgroup takes a BX x BY x BZ block with a ring of two cells, loads it
// ONCE (all its loads in flight together), and walks the pass's stages over the whole block at a time:
//   down (the level's iterate is zero: every level below the finest on the way down): red sweep on block + 2,
//        black sweep on block + 1, residual on the block, the eight residuals of a coarse cell -> coarse b;
//   up:  x + w e on block + 2, red sweep on block + 1, black sweep on the block.
// The ring is relaxed redundantly (same operations on the same operands as the owning block: same bits); every
// row is the marching kernel's chain — slots -K, -J, -I, diagonal, +I, +J, +K from +0, x + (b - s) / a_ii, cells
// outside the grid as zeros — so the two kernels are interchangeable bit for bit (tests/test_gpu_plane.py).
template <typename V>
struct BlockKArgs {
    const V *x_old;
    V *x_new;
    const V *b;
    int nx, ny, nz, nr;
    int nbx, nby;
    V c0, c1, c2, c3, c4, c5, c6, w;
    int fast_div;
    int nxc, nyc;
    const int32_t *cmap;
    V *bc;
    const V *ec;
};

__device__ __forceinline__ double block_quotient(double n, double c, double r, bool fast) {
    if (fast && plain_numerator(n)) {
        const double q0 = n * r;
        return fma(fma(-c, q0, n), r, q0);
    }
    return n / c;
}
__device__ __forceinline__ float block_quotient(float n, float c, float, bool) { return n / c; }

// BX, BY, BZ as template parameters: every index of the stages is then a division by a constant and every loop has a
// known trip count (with run-time block extents the integer divisions alone cost more than the marching kernel)
// SWEEP = false: the pass of a cycle without pre- / post-smoothing (down: the residual of a zero iterate is b itself)
template <typename V, int MODE, int BX, int BY, int BZ, bool SWEEP = true>
__global__ __launch_bounds__(256) void block_kernel(const BlockKArgs<V> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plane_smem[];
    constexpr int EX = BX + 4, EY = BY + 4, EZ = BZ + 4, vol = EX * EY * EZ;
    constexpr int NT = vol >= 1024 ? 256 : 128;       // the launch's workgroup size (choose_block)
    V *const X = reinterpret_cast<V *>(plane_smem);
    V *const B = X + vol;
    V *const E = B + vol;                             // up: the coarse correction under the block + 2 region; down: the block's residuals
    const int t = int(threadIdx.x);
    const int bi = int(blockIdx.x) % a.nbx, bj = (int(blockIdx.x) / a.nbx) % a.nby, bk = int(blockIdx.x) / (a.nbx * a.nby);
    const int i0 = bi * BX - 2, j0 = bj * BY - 2, k0 = bk * BZ - 2;            // grid coordinates of LDS cell (0, 0, 0): even
    constexpr int CX = EX / 2, CY = EY / 2, CZ = EZ / 2;                      // coarse cells under the LDS region
    constexpr int hx = BX / 2, hy = BY / 2, hz = BZ / 2;                      // ... under the block
    const int ci0 = i0 >> 1, cj0 = j0 >> 1, ck0 = k0 >> 1;
    const int nzc = a.nz >> 1;
    const V rc3 = refined_rcp(a.c3);
    const bool fast = a.fast_div != 0;
    auto slot_of = [&](int gi, int gj, int gk) { return (((gi + gj + gk) & 1) ? a.nr : 0) + (((gk * a.ny + gj) * a.nx + gi) >> 1); };
    auto in_grid = [&](int gi, int gj, int gk) { return gi >= 0 && gi < a.nx && gj >= 0 && gj < a.ny && gk >= 0 && gk < a.nz; };

    // ALL the block's loads are requested before the first one is used (a load per loop iteration with its consumer
    // in the same iteration is a memory round trip per iteration: 11 of them for a 16 x 8 x 8 block)
    constexpr int NL = (vol + NT - 1) / NT, NC = MODE == 1 ? (CX * CY * CZ + NT - 1) / NT : (hx * hy * hz + NT - 1) / NT;
    int cs[NC];                                      // coarse slots: up: of the cells under block + 2; down: of the block's coarse cells
#pragma unroll
    for (int n = 0; n < NC; ++n) {
        const int c = t + n * NT;
        int I, J, K;
        bool ok;
        if (MODE == 1) {
            I = ci0 + c % CX; J = cj0 + (c / CX) % CY; K = ck0 + c / (CX * CY);
            ok = c < CX * CY * CZ && I >= 0 && J >= 0 && K >= 0;
        } else {
            I = (bi * BX >> 1) + c % hx; J = (bj * BY >> 1) + (c / hx) % hy; K = (bk * BZ >> 1) + c / (hx * hy);
            ok = c < hx * hy * hz;
        }
        ok = ok && I < a.nxc && J < a.nyc && K < nzc;
        const int ce = (K * a.nyc + J) * a.nxc + I;
        cs[n] = ok ? (a.cmap ? a.cmap[ce] : ce) : -1;
    }
    V xv[NL], bv[NL];
#pragma unroll
    for (int n = 0; n < NL; ++n) {
        const int c = t + n * NT;
        const int gi = i0 + c % EX, gj = j0 + (c / EX) % EY, gk = k0 + c / (EX * EY);
        const bool ok = c < vol && in_grid(gi, gj, gk);
        const int slot = ok ? slot_of(gi, gj, gk) : 0;
        bv[n] = a.b[slot];
        xv[n] = MODE == 1 ? a.x_old[slot] : V(0);
        if (!ok) { bv[n] = V(0); xv[n] = V(0); }      // cells outside the grid are zeros and stay so
    }
    if (MODE == 1) {
        V ev[NC];
#pragma unroll
        for (int n = 0; n < NC; ++n) ev[n] = a.ec[cs[n] >= 0 ? cs[n] : 0];
#pragma unroll
        for (int n = 0; n < NC; ++n)
            if (t + n * NT < CX * CY * CZ) E[t + n * NT] = cs[n] >= 0 ? ev[n] : V(0);
    }
#pragma unroll
    for (int n = 0; n < NL; ++n)
        if (t + n * NT < vol) { X[t + n * NT] = xv[n]; B[t + n * NT] = bv[n]; }
    __syncthreads();
    if (MODE == 1) {
        // openmg/__init__.py:214,220: x + R^T e — the product rounded, then added
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int c = t + n * NT;
            const int li = c % EX, lj = (c / EX) % EY, lk = c / (EX * EY);
            if (c < vol && in_grid(i0 + li, j0 + lj, k0 + lk))
                X[c] = xv[n] + madd(a.w, E[((lk >> 1) * CY + (lj >> 1)) * CX + (li >> 1)], V(0));
        }
        __syncthreads();
    }
    auto row = [&](int c) -> V {
        V s = madd(a.c0, X[c - EX * EY], V(0));
        s = madd(a.c1, X[c - EX], s);
        s = madd(a.c2, X[c - 1], s);
        s = madd(a.c3, X[c], s);
        s = madd(a.c4, X[c + 1], s);
        s = madd(a.c5, X[c + EX], s);
        return madd(a.c6, X[c + EX * EY], s);
    };
    // cells of one colour inside the block widened by RING, clipped to the grid
    auto sweep = [&](int colour, auto RING, auto FROM_ZERO) {
        constexpr int ring = decltype(RING)::value;
        constexpr int wx = BX + 2 * ring, wy = BY + 2 * ring, wz = BZ + 2 * ring, off = 2 - ring;
        // a thread takes PAIRS of cells along x and relaxes the one of the sweep's colour
        constexpr int np = wx / 2 * wy * wz, NI = (np + NT - 1) / NT;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int p = t + n * NT;
            const int pj = (p / (wx / 2)) % wy, pk = p / (wx / 2 * wy);
            const int lj = off + pj, lk = off + pk;
            // off + i0 + j + k parity: i0, j0, k0 are even, so the grid parity is the local one
            const int li = off + 2 * (p % (wx / 2)) + ((off + lj + lk + colour) & 1);
            const int gi = i0 + li, gj = j0 + lj, gk = k0 + lk;
            if (p >= np || !in_grid(gi, gj, gk)) continue;
            const int l = (lk * EY + lj) * EX + li;
            // (FROM_ZERO: the first sweep of a zero iterate — the chain over zeros is +0 exactly, and the region's
            // outermost cells have no neighbours in LDS to read)
            const V sum = decltype(FROM_ZERO)::value ? V(0) : row(l);
            // openmg/solvers.py:68   x[i] = x[i] + (b[i] - Aix) / A[i, i]
            X[l] = X[l] + block_quotient(B[l] - sum, a.c3, rc3, fast);
        }
    };
    typedef std::integral_constant<int, 0> R0;
    typedef std::integral_constant<int, 1> R1;
    typedef std::integral_constant<int, 2> R2;
    constexpr int NB = (BX * BY * BZ + NT - 1) / NT;
    if (MODE == 0) {
        if (SWEEP) {
            sweep(0, R2(), std::true_type());
            __syncthreads();
            sweep(1, R1(), std::false_type());
            __syncthreads();
        }
        // the block's cells: the new iterate out, the residual kept for the restriction
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int c = t + n * NT;
            const int li = 2 + c % BX, lj = 2 + (c / BX) % BY, lk = 2 + c / (BX * BY);
            const int gi = i0 + li, gj = j0 + lj, gk = k0 + lk;
            if (c >= BX * BY * BZ || gi >= a.nx || gj >= a.ny || gk >= a.nz) continue;
            const int l = (lk * EY + lj) * EX + li;
            E[c] = B[l] - row(l);
            if (SWEEP) a.x_new[slot_of(gi, gj, gk)] = X[l];
        }
        __syncthreads();
        // openmg/__init__.py:210: a coarse cell's eight fine residuals in column order
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const int c = t + n * NT;
            if (cs[n] < 0) continue;
            const int I = c % hx, J = (c / hx) % hy, K = c / (hx * hy);
            V acc = V(0);
#pragma unroll
            for (int d = 0; d < 8; ++d)
                acc = madd(a.w, E[((2 * K + (d >> 2)) * BY + 2 * J + ((d >> 1) & 1)) * BX + 2 * I + (d & 1)], acc);
            a.bc[cs[n]] = acc;
        }
    } else {
        if (SWEEP) {
            sweep(0, R1(), std::false_type());
            __syncthreads();
            sweep(1, R0(), std::false_type());
            __syncthreads();
        }
#pragma unroll
        for (int n = 0; n < NB; ++n) {
            const int c = t + n * NT;
            const int li = 2 + c % BX, lj = 2 + (c / BX) % BY, lk = 2 + c / (BX * BY);
            const int gi = i0 + li, gj = j0 + lj, gk = k0 + lk;
            if (c >= BX * BY * BZ || gi >= a.nx || gj >= a.ny || gk >= a.nz) continue;
            a.x_new[slot_of(gi, gj, gk)] = X[(lk * EY + lj) * EX + li];
        }
    }
}

// ---- 2-D levels (five-point stencils, 2 x 2 aggregation): a tile of the grid per workgroup, whole in LDS ---------
// BASELINE configs[1]'s grids (1024^2 and its Galerkin products).  A 2-D level has nothing to march along, so each half
// of the cycle is one launch of independent tiles: the workgroup loads its BX x BY cells with a ring of three (x; b: two),
// relaxes red on tile + 2 and black on tile + 1 redundantly — the same operations on the same operands as the owning
// tiles, so no workgroup waits for another —, and finishes on the tile itself:
//   down: x_new out, residual, the four residuals of a coarse cell in R's column order -> coarse right-hand side;
//   up:   the tile's x is x_old + w e first (openmg/__init__.py:214,220); x_new out; optionally the residual's squares.
// A row is the row kernels' chain over the five slots in column order (-J, -I, diagonal, +I, +J) from +0, a cell outside
// the grid a zero value, and the same quotient: the bits of the set-by-set schedule (tests/test_gpu_plane2d.py).
template <typename V>
struct Tile2KArgs {
    const V *x_old;
    V *x_new;
    const V *b;
    int nx, ny, nr;
    int nbx;
    V c1, c2, c3, c4, c5, w, omega;
    int fast_div;
    int nxc, nyc;
    const int32_t *cmap;
    V *bc;
    const V *ec;
    double *partials;
};

// JAC: weighted Jacobi (the smoother BASELINE configs[1] names) instead of red-black Gauss-Seidel: the level keeps its
// natural ordering; one out-of-place sweep over tile + 1 (x_new in a second LDS image), x + omega ((b - A x) / a_ii)
// spelled like the row kernels' ROW_JACOBI
template <typename V, int MODE, bool XZ, bool NORM, bool SWEEP, int BX, int BY, bool JAC = false>
__global__ __launch_bounds__(256) void tile2d_kernel(const Tile2KArgs<V> a) {
    extern __shared__ __attribute__((aligned(16))) unsigned char plane_smem[];
    constexpr int EX = BX + 6, EY = BY + 6, vol = EX * EY, NT = 256;
    V *const X = reinterpret_cast<V *>(plane_smem);
    V *const B = X + vol;
    V *const XN = B + vol;                                                 // JAC: the relaxed values (tile + 1); otherwise X itself
    V *const E = JAC ? XN + vol : XN;                                      // up: coarse correction under the region; down / norm: the tile's residuals
    __shared__ double s_red[NT / 64];
    const int t = int(threadIdx.x);
    const int bi = int(blockIdx.x) % a.nbx, bj = int(blockIdx.x) / a.nbx;
    const int i0 = bi * BX - 3, j0 = bj * BY - 3;                          // grid coordinates of LDS cell (0, 0): odd
    const V rc3 = refined_rcp(a.c3);
    const bool fast = a.fast_div != 0;
    auto slot_of = [&](int gi, int gj) { return JAC ? gj * a.nx + gi : (((gi + gj) & 1) ? a.nr : 0) + ((gj * a.nx + gi) >> 1); };
    auto in_grid = [&](int gi, int gj) { return gi >= 0 && gi < a.nx && gj >= 0 && gj < a.ny; };
    // all loads requested before the first is used
    constexpr int NL = (vol + NT - 1) / NT;
    V xv[NL], bv[NL];
#pragma unroll
    for (int n = 0; n < NL; ++n) {
        const int c = t + n * NT;
        const int gi = i0 + c % EX, gj = j0 + c / EX;
        const bool ok = c < vol && in_grid(gi, gj);
        const int slot = ok ? slot_of(gi, gj) : 0;
        bv[n] = a.b[slot];
        xv[n] = XZ ? V(0) : a.x_old[slot];
        if (!ok) { bv[n] = V(0); xv[n] = V(0); }                           // cells outside the grid are zeros and stay so
    }
    if (MODE == 1) {
        // the coarse cells under the region: LDS cell (li, lj) lies over coarse cell ((i0 + li) >> 1, (j0 + lj) >> 1), i0, j0 odd
        constexpr int NC = (((EX + 1) / 2 + 1) * ((EY + 1) / 2 + 1) + NT - 1) / NT;
        constexpr int CW = (EX + 1) / 2 + 1, CH = (EY + 1) / 2 + 1;
        const int ci0 = (i0 - 1) / 2, cj0 = (j0 - 1) / 2;                  // floor(i0 / 2), floor(j0 / 2): i0, j0 are odd
#pragma unroll
        for (int n = 0; n < NC; ++n) {
            const int c = t + n * NT;
            const int I = ci0 + c % CW, J = cj0 + c / CW;
            const bool ok = c < CW * CH && I >= 0 && J >= 0 && I < a.nxc && J < a.nyc;
            const int ce = J * a.nxc + I;
            const V ev = ok ? a.ec[a.cmap ? a.cmap[ce] : ce] : V(0);
            if (c < CW * CH) E[c] = ev;
        }
        __syncthreads();
        // openmg/__init__.py:214,220: x + R^T e — the product rounded, then added
#pragma unroll
        for (int n = 0; n < NL; ++n) {
            const int c = t + n * NT;
            const int gi = i0 + c % EX, gj = j0 + c / EX;
            if (c < vol && in_grid(gi, gj)) xv[n] = xv[n] + madd(a.w, E[((gj >> 1) - cj0) * CW + ((gi >> 1) - ci0)], V(0));
        }
        __syncthreads();
    }
#pragma unroll
    for (int n = 0; n < NL; ++n)
        if (t + n * NT < vol) { X[t + n * NT] = xv[n]; B[t + n * NT] = bv[n]; }
    __syncthreads();
    auto row_of = [&](const V *Y, int c) -> V {
        V s = madd(a.c1, Y[c - EX], V(0));
        s = madd(a.c2, Y[c - 1], s);
        s = madd(a.c3, Y[c], s);
        s = madd(a.c4, Y[c + 1], s);
        return madd(a.c5, Y[c + EX], s);
    };
    auto row = [&](int c) -> V { return row_of(X, c); };
    // cells of one colour inside the tile widened by RING (<= 2), clipped to the grid
    auto sweep = [&](int colour, auto RING) {
        constexpr int ring = decltype(RING)::value;
        constexpr int wx = BX + 2 * ring, wy = BY + 2 * ring, off = 3 - ring;
        constexpr int np = wx / 2 * wy, NI = (np + NT - 1) / NT;
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int p = t + n * NT;
            const int lj = off + p / (wx / 2);
            // i0, j0 are odd: grid parity of LDS cell (li, lj) = (li + lj) & 1
            const int li = off + 2 * (p % (wx / 2)) + ((off + lj + colour) & 1);
            const int gi = i0 + li, gj = j0 + lj;
            if (p >= np || !in_grid(gi, gj)) continue;
            const int l = lj * EX + li;
            // openmg/solvers.py:68   x[i] = x[i] + (b[i] - Aix) / A[i, i]
            X[l] = X[l] + block_quotient(B[l] - row(l), a.c3, rc3, fast);
        }
    };
    const V *Y = X;                                                        // the iterate the tile is finished from
    if (SWEEP && JAC) {
        // one weighted-Jacobi sweep over tile + 1, out of place (cells outside the grid stay zero)
        constexpr int wx = BX + 2, wy = BY + 2, np = wx * wy, NI = (np + NT - 1) / NT;
        for (int i = t; i < vol; i += NT) XN[i] = V(0);
        __syncthreads();
#pragma unroll
        for (int n = 0; n < NI; ++n) {
            const int p = t + n * NT;
            const int li = 2 + p % wx, lj = 2 + p / wx;
            const int gi = i0 + li, gj = j0 + lj;
            if (p >= np || !in_grid(gi, gj)) continue;
            const int l = lj * EX + li;
            XN[l] = X[l] + a.omega * block_quotient(B[l] - row(l), a.c3, rc3, fast);
        }
        __syncthreads();
        Y = XN;
    } else if (SWEEP) {
        sweep(0, std::integral_constant<int, 2>());
        __syncthreads();
        sweep(1, std::integral_constant<int, 1>());
        __syncthreads();
    }
    constexpr int NB = (BX * BY + NT - 1) / NT;
    double sq = 0.0;
#pragma unroll
    for (int n = 0; n < NB; ++n) {
        const int c = t + n * NT;
        const int li = 3 + c % BX, lj = 3 + c / BX;
        const int gi = i0 + li, gj = j0 + lj;
        if (c >= BX * BY || gi >= a.nx || gj >= a.ny) continue;
        const int l = lj * EX + li;
        if (SWEEP || MODE == 1) a.x_new[slot_of(gi, gj)] = Y[l];
        if (MODE == 0 || NORM) {
            const V res = B[l] - row_of(Y, l);
            if (MODE == 0) E[c] = res;
            if (NORM) sq = fma(double(res), double(res), sq);
        }
    }
    if (MODE == 0) {
        __syncthreads();
        // openmg/__init__.py:210: a coarse cell's four fine residuals in column order
        constexpr int hx = BX / 2, hy = BY / 2, NCC = (hx * hy + NT - 1) / NT;
#pragma unroll
        for (int n = 0; n < NCC; ++n) {
            const int c = t + n * NT;
            const int I = c % hx, J = c / hx;
            const int GI = (bi * BX >> 1) + I, GJ = (bj * BY >> 1) + J;
            if (c >= hx * hy || GI >= a.nxc || GJ >= a.nyc) continue;
            V acc = V(0);
#pragma unroll
            for (int d = 0; d < 4; ++d) acc = madd(a.w, E[(2 * J + (d >> 1)) * BX + 2 * I + (d & 1)], acc);
            const int ce = GJ * a.nxc + GI;
            a.bc[a.cmap ? a.cmap[ce] : ce] = acc;
        }
    }
    if (NORM) {
        // fixed order: lanes of a wave (shuffle tree), then the waves in turn
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) sq += __shfl_down(sq, off, 64);
        if ((t & 63) == 0) s_red[t >> 6] = sq;
        __syncthreads();
        if (t == 0) {
            double tot = 0.0;
            for (int wv = 0; wv < NT / 64; ++wv) tot += s_red[wv];
            a.partials[blockIdx.x] = tot;
        }
    }
}

// ---- host: does the level qualify, and how is it tiled -----------------------------------------------
int env_int3(const char *name, int out[3]) {
    const char *e = getenv(name);
    if (!e || !e[0]) return 0;
    return sscanf(e, "%d,%d,%d", &out[0], &out[1], &out[2]) == 3 ? 1 : 0;
}

inline int round_up(int v, int m) { return (v + m - 1) / m * m; }

// (decided once per plan, so that a process can hold hierarchies built with different switches: the tests do)
inline void small_level_switches(PlaneGeom &g) {
    auto env = [](const char *name, long long dflt) { const char *e = getenv(name); return e && e[0] ? atoll(e) : dflt; };
    const bool whole = g.z_base == 0 && g.z_end == g.nz && g.kv0 == 0 && g.kv1 == g.nz;
    g.block = env("OMG_PLANE_BLOCK", 1) != 0 && whole && int64_t(g.nx) * g.ny * g.nz <= env("OMG_PLANE_BLOCK_CELLS", int64_t(64) * 64 * 64);
    g.la2 = env("OMG_PLANE_LA2", 1) != 0;
}

// model 0: one workgroup per CU, cost ~ rounds x steps x (threads + a fixed per-step price).
// model 1: the CU's four SIMDs evenly loaded — a workgroup of 6 waves costs a step what one of 8 does (two of its
// SIMDs carry two waves), and several small workgroups share a CU.  Neither predicts the other's winner
// (256^3: 64 x 32 x 32 at 132 us against 128 x 22 x 26 at 122 us per down pass; other shapes the other way):
// PlanePlan::tune() times both on the level's own vectors.
// `more` (nullable): the best few tilings of the model beyond the winner, best first (PlanePlan::tune times them too)
void choose_tiles(PlaneGeom &g, size_t value_bytes, int model = 0, std::vector<std::array<int, 3>> *more = nullptr, int keep = 0) {
    const int nzo = g.z_end - g.z_base;             // planes the launch relaxes
    auto set = [&](int TX, int TY, int LZ) {
        g.TX = TX; g.TY = TY; g.LZ = LZ;
        g.PX = TX / 4 + 2; g.PY = TY / 2 + 4;
        g.ntx = (g.nx + TX - 1) / TX; g.nty = (g.ny + TY - 1) / TY; g.ntz = (nzo + LZ - 1) / LZ;
        g.n_wg = g.ntx * g.nty * g.ntz;
        g.threads = round_up(g.PX * g.PY, 64);
        g.lds_bytes = size_t(6) * size_t(2 * g.PY + 2) * size_t(2 * g.PX + 4) * value_bytes;
    };
    int forced[3];
    if (env_int3("OMG_PLANE_TILE", forced)) {
        const int TX = std::max(4, round_up(forced[0], 4)), TY = std::max(2, round_up(forced[1], 2)), LZ = std::max(2, round_up(forced[2], 2));
        set(TX, TY, LZ);
        if (g.threads <= 512 && g.lds_bytes <= size_t(160) * 1024) return;
    }
    // Few, fat workgroups: one per CU (256), each as large a tile as its registers hold (<= 512 threads of
    // two lines x four cells), the z chunk as long as the count allows.  cost ~ rounds of workgroups x
    // steps x (threads + a fixed per-step price).
    std::vector<std::pair<double, std::array<int, 3>>> all;
    for (int ntx = 1; ntx <= (g.nx + 3) / 4; ++ntx) {
        const int TX = round_up((g.nx + ntx - 1) / ntx, 4);
        if ((g.nx + TX - 1) / TX != ntx) continue;
        for (int nty = 1; nty <= g.ny / 2; ++nty) {
            const int TY = round_up((g.ny + nty - 1) / nty, 2);
            if ((g.ny + TY - 1) / TY != nty) continue;
            const int thr = round_up((TX / 4 + 2) * (TY / 2 + 4), 64);
            if (thr > 512) continue;
            if (size_t(6) * size_t(TY + 10) * size_t(TX / 2 + 8) * value_bytes > size_t(150) * 1024) continue;
            for (int ntz = 1; ntz <= nzo / 2; ++ntz) {
                const int LZ = round_up((nzo + ntz - 1) / ntz, 2);
                if ((nzo + LZ - 1) / LZ != ntz) continue;
                double cost;
                if (model == 0) {
                    const double rounds = std::ceil(double(ntx) * nty * ntz / 256.0);
                    cost = rounds * (LZ + 4) * (thr + 192.0);
                } else {
                    const int waves = thr / 64;
                    const size_t lds = size_t(6) * size_t(TY + 10) * size_t(TX / 2 + 8) * value_bytes;
                    const int per_cu = std::max(1, std::min(8 / waves, int(size_t(160) * 1024 / lds)));
                    const int simd = (waves * per_cu + 3) / 4;
                    const double rounds = std::ceil(double(ntx) * nty * ntz / (256.0 * per_cu));
                    cost = rounds * (LZ + 4) * (256.0 * simd + 192.0);
                }
                all.push_back({cost, {TX, TY, LZ}});
            }
        }
    }
    if (all.empty()) { g.TX = 0; return; }
    std::stable_sort(all.begin(), all.end(), [](const auto &u, const auto &v) { return u.first < v.first; });
    set(all[0].second[0], all[0].second[1], all[0].second[2]);
    if (more)
        for (size_t i = 1; i < all.size() && int(more->size()) < keep; ++i) more->push_back(all[i].second);
}

// 2-D levels: the tile shapes tile2d_kernel is instantiated for, largest first; the largest that still gives 128 workgroups
constexpr int TILE2_SHAPES[3][2] = {{64, 32}, {32, 16}, {16, 8}};
void choose_tile2d(PlaneGeom &g, size_t value_bytes) {
    int pick = 2;
    for (int i = 0; i < 3; ++i) {
        const int64_t n = int64_t((g.nx + TILE2_SHAPES[i][0] - 1) / TILE2_SHAPES[i][0]) * ((g.ny + TILE2_SHAPES[i][1] - 1) / TILE2_SHAPES[i][1]);
        if (n >= 128) { pick = i; break; }
    }
    g.block = false;
    g.TX = TILE2_SHAPES[pick][0]; g.TY = TILE2_SHAPES[pick][1]; g.LZ = 1;
    g.ntx = (g.nx + g.TX - 1) / g.TX; g.nty = (g.ny + g.TY - 1) / g.TY; g.ntz = 1;
    g.n_wg = g.ntx * g.nty;
    g.threads = 256;
    const size_t vol = size_t(g.TX + 6) * size_t(g.TY + 6);
    g.lds_bytes = ((g.jacobi ? 3 : 2) * vol + std::max(size_t(g.TX) * g.TY, size_t(g.TX / 2 + 4) * size_t(g.TY / 2 + 4))) * value_bytes;
    (void)pick;
}

}  // namespace

template <typename V>
bool PlanePlan<V>::build(const omg_csr &A, const omg_csr &R, Ordering &ord, bool jacobi, double omega) {
    {
        const char *e = getenv("OMG_PLANE");
        if (e && e[0] == '0') return false;
    }
    const int64_t n = A.n_rows;
    if (n < 8 || n != A.n_cols || uint64_t(n) * sizeof(V) >= (uint64_t(1) << 31)) return false;
    auto has = [&](int64_t r, int64_t c) {
        for (int64_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p)
            if (A.indices[p] == c) return true;
        return false;
    };
    // the grid, read off the couplings: the first row not coupled to its predecessor starts the second
    // line, the first line not coupled to the previous one the second plane
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
    // nz == 1: a 2-D grid (five-point stencil, 2 x 2 aggregation): tile2d_kernel instead of the marching kernel
    const bool dim2 = nz == 1;
    if ((nx & 1) || (ny & 1) || (!dim2 && (nz & 1)) || ny < 2) return false;
    if (jacobi && !dim2) return false;                 // weighted Jacobi: the 2-D tile passes only
    {
        const char *e = experiment_env("OMG_PLANE_2D");
        if (dim2 && e && e[0] == '0') return false;
    }
    const int64_t sj = nx, sk = nx * ny;
    // (the ordering — two colours by parity, red first — is WRITTEN below once the level has qualified: it is
    // what the greedy smallest-free-colour pass in natural order finds on such a stencil, every lower
    // neighbour of a cell having the other parity; tests/test_gpu_plane.py compares the two)
    // the coefficients: from the first row that has each slot
    bool havec[7] = {false, false, false, false, false, false, false};
    double c[7] = {0, 0, 0, 0, 0, 0, 0};
    auto slot_of = [&](int64_t r, int64_t col) -> int {
        const int64_t i = r % nx, jl = (r / nx) % ny, kl = r / sk, off = col - r;
        if (off == 0) return 3;
        if (off == -1 && i > 0) return 2;
        if (off == 1 && i + 1 < nx) return 4;
        if (off == -sj && jl > 0) return 1;
        if (off == sj && jl + 1 < ny) return 5;
        if (off == -sk && kl > 0) return 0;
        if (off == sk && kl + 1 < nz) return 6;
        return -1;
    };
    {
        const int64_t probe = std::min<int64_t>(n - 1, sk + sj + 1);      // cell (1, 1, 1): an interior row when every extent is >= 3
        for (int64_t r : {probe, int64_t(0), n - 1}) {
            for (int64_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p) {
                const int sl = slot_of(r, A.indices[p]);
                if (sl < 0) return false;
                if (!havec[sl]) { havec[sl] = true; c[sl] = A.data[p]; }
            }
        }
        if (dim2) havec[0] = havec[6] = true;               // (no such slots: their coefficients stay 0)
        for (int e = 0; e < 7; ++e)
            if (!havec[e]) return false;
        if (!(std::fabs(c[3]) > 0.0) || !std::isfinite(c[3])) return false;
    }
    // every row: exactly its in-grid neighbours, in slot (= column) order, with those coefficients (as V
    // they are what the device format holds); its slot in the ordering; and R's row of every coarse cell
    const int64_t nxc = nx / 2, nyc = ny / 2, nzc = dim2 ? 1 : nz / 2;
    const int per = dim2 ? 4 : 8;                            // fine cells of an aggregate
    if (R.n_cols != n || R.n_rows != nxc * nyc * nzc || R.nnz != n) return false;
    const double w = R.nnz ? R.data[0] : 0.0;
    // (memory-bound: the scan reads every byte of the operator once; 64 threads on a many-core host: 64 -> ~15 ms at 256^3)
    const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const int nt = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 65536));
    std::atomic<bool> ok(true);
    auto scan = [&](int tnum) {
        const int64_t lo = n * tnum / nt, hi = n * (tnum + 1) / nt;
        int64_t i = lo % nx, jl = (lo / nx) % ny, kl = lo / sk;
        const int64_t off7[7] = {-sk, -sj, -1, 0, 1, sj, sk};
        for (int64_t r = lo; r < hi; ++r) {
            if (r > lo && ++i == nx) {                    // (the cell's coordinates, carried along instead of divided out)
                i = 0;
                if (++jl == ny) { jl = 0; ++kl; }
            }
            const bool present[7] = {kl > 0, jl > 0, i > 0, true, i + 1 < nx, jl + 1 < ny, kl + 1 < nz};
            int64_t p = A.indptr[r];
            const int64_t pe = A.indptr[r + 1];
            for (int e = 0; e < 7; ++e) {
                if (!present[e]) continue;
                // (the column itself: slot_of() costs three integer divisions per ENTRY — the scan was bound by them, not by memory)
                if (p >= pe || int64_t(A.indices[p]) != r + off7[e] || A.data[p] != c[e]) { ok = false; return; }
                ++p;
            }
            if (p != pe) { ok = false; return; }
        }
        const int64_t clo = R.n_rows * tnum / nt, chi = R.n_rows * (tnum + 1) / nt;
        for (int64_t cr = clo; cr < chi; ++cr) {
            const int64_t I = cr % nxc, J = (cr / nxc) % nyc, K = cr / (nxc * nyc);
            int64_t p = R.indptr[cr];
            if (R.indptr[cr + 1] - p != per) { ok = false; return; }
            for (int dk = 0; dk < (dim2 ? 1 : 2); ++dk)
                for (int dj = 0; dj < 2; ++dj)
                    for (int di = 0; di < 2; ++di, ++p) {
                        const int64_t f = (2 * K + dk) * sk + (2 * J + dj) * sj + 2 * I + di;
                        if (R.indices[p] != f || R.data[p] != w) { ok = false; return; }
                    }
        }
    };
    {
        std::vector<std::thread> th;
        for (int tnum = 1; tnum < nt; ++tnum) th.emplace_back(scan, tnum);
        scan(0);
        for (auto &q : th) q.join();
    }
    if (!ok) return false;
    ord = Ordering();
    if (jacobi) {
        ord.identity = true;                           // one set, natural numbering
        ord.sets = {0, n};
    } else {
    ord.identity = false;
    ord.sets = {0, n / 2, n};
    ord.perm.resize(size_t(n));
    ord.inv.resize(size_t(n));
    {
        auto fill = [&](int tnum) {
            const int64_t lo = n * tnum / nt, hi = n * (tnum + 1) / nt;
            for (int64_t r = lo; r < hi; ++r) {
                const int64_t i = r % nx, jl = (r / nx) % ny, kl = r / sk;
                const int64_t slot = ((i + jl + kl) & 1 ? n / 2 : 0) + r / 2;
                ord.inv[size_t(r)] = int32_t(slot);
                ord.perm[size_t(slot)] = int32_t(r);
            }
        };
        std::vector<std::thread> th;
        for (int tnum = 1; tnum < nt; ++tnum) th.emplace_back(fill, tnum);
        fill(0);
        for (auto &q : th) q.join();
    }
    }
    return finish_geometry(nx, ny, nz, c, w, jacobi, omega);
}

// the plan of a whole nx x ny x nz grid (nz = 1: 2-D) with these coefficients
template <typename V>
bool PlanePlan<V>::finish_geometry(int64_t nx, int64_t ny, int64_t nz, const double (&c)[7], double w, bool jacobi, double omega) {
    const bool dim2 = nz == 1;
    // (a float level holds the rounded coefficients: every entry with one value rounds to one value)
    g = PlaneGeom();
    g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz; g.hx = (int)(nx / 2);
    for (int e = 0; e < 7; ++e) g.c[e] = double(V(c[e]));
    g.w = double(V(w));
    g.z_base = 0; g.z_end = g.nz; g.kv0 = 0; g.kv1 = g.nz; g.kc_off = 0; g.nzc = g.nz / 2;
    g.dim2 = dim2;
    g.jacobi = jacobi;
    g.omega = jacobi ? double(V(omega)) : 1.0;
    if (dim2) {
        choose_tile2d(g, sizeof(V));
    } else {
        choose_tiles(g, sizeof(V));
        small_level_switches(g);
        if (g.TX <= 0 || g.threads > 512) return false;
    }
    partials.alloc(size_t(g.n_wg) + SUM_FOLD);
    return true;
}

namespace {
// every row of a device CSR: exactly its in-grid neighbours of the 5 / 7-point stencil in slot order with the level's
// ONE coefficient per slot?  err: 0, or 1 + the smallest offending row
__global__ __launch_bounds__(256) void plane_check_kernel(const int32_t *indptr, const int32_t *indices, const double *data, int nx, int ny, int nz,
                                                          double c0, double c1, double c2, double c3, double c4, double c5, double c6,
                                                          unsigned long long *err) {
    const int64_t n = int64_t(nx) * ny * nz, sj = nx, sk = int64_t(nx) * ny;
    const int64_t r = int64_t(blockIdx.x) * 256 + threadIdx.x;
    if (r >= n) return;
    const int64_t i = r % nx, j = (r / nx) % ny, k = r / sk;
    const bool present[7] = {k > 0, j > 0, i > 0, true, i + 1 < nx, j + 1 < ny, k + 1 < nz};
    const int64_t off[7] = {-sk, -sj, -1, 0, 1, sj, sk};
    const double c[7] = {c0, c1, c2, c3, c4, c5, c6};
    int64_t p = indptr[r];
    const int64_t pe = indptr[r + 1];
    bool bad = false;
    for (int e = 0; e < 7; ++e) {
        if (!present[e]) continue;
        if (p >= pe || int64_t(indices[p]) != r + off[e] || data[p] != c[e]) { bad = true; break; }
        ++p;
    }
    if (bad || p != pe) atomicMin(err, (unsigned long long)(r + 1));
}
}  // namespace

template <typename V>
bool PlanePlan<V>::build_device(const DevCsrPlain &A, int nx, int ny, int nz, double w, Ordering &ord, bool jacobi, double omega, hipStream_t s) {
    {
        const char *e = getenv("OMG_PLANE");
        if (e && e[0] == '0') return false;
    }
    const int64_t n = int64_t(nx) * ny * nz;
    const bool dim2 = nz == 1;
    if (n < 8 || n != A.n_rows || n != A.n_cols || uint64_t(n) * sizeof(V) >= (uint64_t(1) << 31)) return false;
    if ((nx & 1) || (ny & 1) || (!dim2 && (nz & 1)) || ny < 2) return false;
    if (jacobi && !dim2) return false;
    {
        const char *e = experiment_env("OMG_PLANE_2D");
        if (dim2 && e && e[0] == '0') return false;
    }
    const int64_t want = dim2 ? 5 * n - 2 * nx - 2 * ny : 7 * n - 2 * (int64_t(nx) * ny + int64_t(ny) * nz + int64_t(nx) * nz);
    if (A.nnz != want) return false;
    // the coefficients: from cell (1, 1, 1) (an interior row when every extent is >= 3), the first and the last row
    const int64_t sj = nx, sk = int64_t(nx) * ny;
    bool havec[7] = {false, false, false, false, false, false, false};
    double c[7] = {0, 0, 0, 0, 0, 0, 0};
    const int64_t probe = std::min<int64_t>(n - 1, sk + sj + 1);
    for (int64_t r : {probe, int64_t(0), n - 1}) {
        int32_t pp[2];
        OMG_HIP(hipMemcpyAsync(pp, A.indptr.p + r, sizeof(pp), hipMemcpyDeviceToHost, s));
        OMG_HIP(hipStreamSynchronize(s));
        const int len = pp[1] - pp[0];
        if (len < 1 || len > 7) return false;
        int32_t idx[7];
        double val[7];
        OMG_HIP(hipMemcpyAsync(idx, A.indices.p + pp[0], size_t(len) * sizeof(int32_t), hipMemcpyDeviceToHost, s));
        OMG_HIP(hipMemcpyAsync(val, A.data.p + pp[0], size_t(len) * sizeof(double), hipMemcpyDeviceToHost, s));
        OMG_HIP(hipStreamSynchronize(s));
        const int64_t i = r % nx, jl = (r / nx) % ny, kl = r / sk;
        for (int q = 0; q < len; ++q) {
            const int64_t off = int64_t(idx[q]) - r;
            int sl = -1;
            if (off == 0) sl = 3;
            else if (off == -1 && i > 0) sl = 2;
            else if (off == 1 && i + 1 < nx) sl = 4;
            else if (off == -sj && jl > 0) sl = 1;
            else if (off == sj && jl + 1 < ny) sl = 5;
            else if (off == -sk && kl > 0) sl = 0;
            else if (off == sk && kl + 1 < nz) sl = 6;
            if (sl < 0) return false;
            if (!havec[sl]) { havec[sl] = true; c[sl] = val[q]; }
        }
    }
    if (dim2) havec[0] = havec[6] = true;
    for (int e = 0; e < 7; ++e)
        if (!havec[e]) return false;
    if (!(std::fabs(c[3]) > 0.0) || !std::isfinite(c[3])) return false;
    DevBuf<unsigned long long> d_err(1);
    OMG_HIP(hipMemsetAsync(d_err.p, 0xFF, sizeof(unsigned long long), s));
    hipLaunchKernelGGL(plane_check_kernel, dim3(unsigned((n + 255) / 256)), dim3(256), 0, s, A.indptr.p, A.indices.p, A.data.p, nx, ny, nz,
                       c[0], c[1], c[2], c[3], c[4], c[5], c[6], d_err.p);
    OMG_HIP(hipGetLastError());
    unsigned long long err = 0;
    OMG_HIP(hipMemcpyAsync(&err, d_err.p, sizeof(err), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    if (err != ~0ull) return false;
    ord = Ordering();
    if (jacobi) {
        ord.identity = true;
        ord.sets = {0, n};
    } else {
        ord.identity = false;
        ord.sets = {0, n / 2, n};
        ord.closed_form = 2; ord.cf_nx = nx; ord.cf_ny = ny; ord.cf_nz = nz;     // perm / inv: on the device (fill_ordering_device), on the host on demand
    }
    return finish_geometry(nx, ny, nz, c, w, jacobi, omega);
}

template <typename V>
void PlanePlan<V>::build_slab(int nx, int ny, int nz_own, int ghost, int ghost_c, bool first, bool last, const double (&c)[7], double w) {
    OMG_REQUIRE(nx >= 2 && ny >= 2 && nz_own >= 2 && !(nx & 1) && !(ny & 1) && !(nz_own & 1) && !(ghost & 1) && ghost >= 4 && ghost_c >= 2,
                "plane slab: extents must be even, at least four ghost planes (two on the coarse slab)");
    g = PlaneGeom();
    g.nx = nx; g.ny = ny; g.nz = nz_own + 2 * ghost; g.hx = nx / 2;
    for (int e = 0; e < 7; ++e) g.c[e] = double(V(c[e]));
    g.w = double(V(w));
    g.z_base = ghost; g.z_end = ghost + nz_own;
    g.kv0 = first ? ghost : 0;
    g.kv1 = last ? ghost + nz_own : g.nz;
    g.nzc = nz_own / 2 + 2 * ghost_c;
    g.kc_off = ghost_c - ghost / 2;
    OMG_REQUIRE(uint64_t(nx) * ny * g.nz * sizeof(V) < (uint64_t(1) << 31), "plane slab: vector exceeds 2 GiB");
    choose_tiles(g, sizeof(V));
    small_level_switches(g);
    OMG_REQUIRE(g.TX > 0 && g.threads <= 512, "plane slab: no tiling");
    partials.alloc(size_t(std::max(g.n_wg, split_partials())) + SUM_FOLD);
}

// Times the candidate tilings on the caller's vectors (their contents are destroyed; zeros are a fair input) and
// keeps the fastest.  Large levels only: a few launches, ~3 ms.
template <typename V>
void PlanePlan<V>::tune(V *x, V *tmp, const V *b, const Coarse &c, hipStream_t s, bool finest) {
    {
        const char *e = experiment_env("OMG_PLANE_TUNE");
        int forced[3];
        if ((e && e[0] == '0') || env_int3("OMG_PLANE_TILE", forced)) return;
    }
    if (g.dim2 || int64_t(g.nx) * g.ny * (g.z_end - g.z_base) < (int64_t(1) << 21)) return;
    // Candidates: the winner of either cost model and the next few of each (OMG_PLANE_TUNE_K per model, default 0: measured in round 4, profiles/r04_plane_tune_candidates.txt: none beat the winners) —
    // neither model predicts the other's winner, nor always the fastest tiling; all are TIMED on the level's own vectors.
    static const int keep = [] { const char *e = experiment_env("OMG_PLANE_TUNE_K"); return e ? std::max(0, atoi(e)) : 0; }();
    static const bool debug = [] { const char *e = experiment_env("OMG_PLANE_TUNE_DEBUG"); return e && e[0] == '1'; }();
    std::vector<PlaneGeom> cand;
    {
        std::vector<std::array<int, 3>> more0, more1;
        PlaneGeom g0 = g, g1 = g;
        choose_tiles(g0, sizeof(V), 0, &more0, keep);
        choose_tiles(g1, sizeof(V), 1, &more1, keep);
        auto add = [&](const PlaneGeom &q) {
            if (q.TX <= 0 || q.threads > 512) return;
            for (const PlaneGeom &c : cand)
                if (c.TX == q.TX && c.TY == q.TY && c.LZ == q.LZ) return;
            cand.push_back(q);
        };
        add(g);
        add(g0);
        add(g1);
        auto from = [&](const std::array<int, 3> &t) {
            PlaneGeom q = g;
            q.TX = t[0]; q.TY = t[1]; q.LZ = t[2];
            q.PX = q.TX / 4 + 2; q.PY = q.TY / 2 + 4;
            const int nzo = q.z_end - q.z_base;
            q.ntx = (q.nx + q.TX - 1) / q.TX; q.nty = (q.ny + q.TY - 1) / q.TY; q.ntz = (nzo + q.LZ - 1) / q.LZ;
            q.n_wg = q.ntx * q.nty * q.ntz;
            q.threads = round_up(q.PX * q.PY, 64);
            q.lds_bytes = size_t(6) * size_t(2 * q.PY + 2) * size_t(2 * q.PX + 4) * sizeof(V);
            return q;
        };
        for (const auto &t : more0) add(from(t));
        for (const auto &t : more1) add(from(t));
        // OMG_PLANE_TUNE_EXTRA="TX,TY,LZ;TX,TY,LZ;...": more tilings to time on the large levels (a tiling that does not fit is skipped)
        if (const char *e = experiment_env("OMG_PLANE_TUNE_EXTRA")) {
            const char *q = e;
            while (*q) {
                int t3[3];
                if (sscanf(q, "%d,%d,%d", &t3[0], &t3[1], &t3[2]) == 3 && t3[0] >= 4 && t3[1] >= 2 && t3[2] >= 2) {
                    PlaneGeom c = from({round_up(t3[0], 4), round_up(t3[1], 2), round_up(t3[2], 2)});
                    if (c.lds_bytes <= size_t(160) * 1024) add(c);
                }
                while (*q && *q != ';') ++q;
                if (*q == ';') ++q;
            }
        }
    }
    if (cand.size() < 2) return;
    // One decision per process and shape: the norm's partial sums follow the tiles, and two hierarchies of one process
    // must not differ in its last bit because a timing came out the other way (where the candidates are close).
    static std::mutex mu;
    static std::map<std::array<int, 8>, std::array<int, 3>> decided;
    const std::array<int, 8> key = {g.nx, g.ny, g.nz, g.z_base, g.z_end, int(sizeof(V)), finest ? 1 : 0, g.kv1 - g.kv0};
    auto adopt = [&](const std::array<int, 3> &t) {
        for (const PlaneGeom &c : cand)
            if (c.TX == t[0] && c.TY == t[1] && c.LZ == t[2]) { g = c; break; }
        partials.alloc(size_t(std::max(g.n_wg, split_partials())) + SUM_FOLD);
    };
    {
        std::lock_guard<std::mutex> lock(mu);
        const auto it = decided.find(key);
        if (it != decided.end()) {
            adopt(it->second);
            return;
        }
    }
    int max_wg = 0;
    for (const PlaneGeom &c : cand) max_wg = std::max(max_wg, c.n_wg);
    partials.alloc(size_t(std::max(max_wg, split_partials())) + SUM_FOLD);
    hipEvent_t e0, e1;
    OMG_HIP(hipEventCreate(&e0));
    OMG_HIP(hipEventCreate(&e1));
    float best = 0.0f;
    size_t pick = 0;
    for (size_t i = 0; i < cand.size(); ++i) {
        g = cand[i];
        // (the finest level's passes read the iterate and square the residual; the others' start from zero)
        down(x, tmp, b, !finest, c, s);                           // (not timed: the kernel's first launch)
        up(tmp, x, b, c, finest ? partials.p : nullptr, s);
        OMG_HIP(hipEventRecord(e0, s));
        for (int r = 0; r < 2; ++r) {
            down(x, tmp, b, !finest, c, s);
            up(tmp, x, b, c, finest ? partials.p : nullptr, s);
        }
        OMG_HIP(hipEventRecord(e1, s));
        OMG_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        OMG_HIP(hipEventElapsedTime(&ms, e0, e1));
        if (debug)
            fprintf(stderr, "[plane tune] %dx%dx%d%s: tile %3d x %3d x %3d  %4d workgroups of %3d threads  %7.1f us per down + up\n", g.nx, g.ny, g.nz,
                    finest ? " (finest)" : "", g.TX, g.TY, g.LZ, g.n_wg, g.threads, 1e3 * ms / 2);
        if (i == 0 || ms < best) { best = ms; pick = i; }
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    std::array<int, 3> chosen = {cand[pick].TX, cand[pick].TY, cand[pick].LZ};
    {
        std::lock_guard<std::mutex> lock(mu);
        chosen = decided.emplace(key, chosen).first->second;
    }
    adopt(chosen);
}

template <typename V>
float PlanePlan<V>::time_pair(V *x, V *tmp, const V *b, const Coarse &c, hipStream_t s, bool finest, int pairs) {
    hipEvent_t e0, e1;
    OMG_HIP(hipEventCreate(&e0));
    OMG_HIP(hipEventCreate(&e1));
    down(x, tmp, b, !finest, c, s);
    up(tmp, x, b, c, finest ? partials.p : nullptr, s);
    OMG_HIP(hipEventRecord(e0, s));
    for (int r = 0; r < pairs; ++r) {
        down(x, tmp, b, !finest, c, s);
        up(tmp, x, b, c, finest ? partials.p : nullptr, s);
    }
    OMG_HIP(hipEventRecord(e1, s));
    OMG_HIP(hipEventSynchronize(e1));
    float ms = 0.0f;
    OMG_HIP(hipEventElapsedTime(&ms, e0, e1));
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
    return 1e3f * ms / float(std::max(pairs, 1));
}

template <typename V>
HostCsr PlanePlan<V>::operator_csr() const {
    const int64_t nx = g.nx, ny = g.ny, nz = g.nz, n = nx * ny * nz, sj = nx, sk = nx * ny;
    HostCsr A;
    A.n_rows = A.n_cols = n;
    A.nnz = 7 * n - 2 * (nx * ny + ny * nz + nx * nz);
    A.indptr.resize(size_t(n) + 1);
    A.indices.resize(size_t(A.nnz));
    A.data.resize(size_t(A.nnz));
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())), nz));
    auto fill = [&](int tnum) {
        // rows of planes [k0, k1): entries before plane k = 7 (rows) - 2 per missing neighbour
        const int64_t k0 = nz * tnum / nt, k1 = nz * (tnum + 1) / nt;
        auto before_plane = [&](int64_t k) {       // stored entries of the rows of planes < k
            // per plane: 7 nx ny - 2 nx (j faces) - 2 ny (i faces); - nx ny for each of -K (plane 0) and +K (last plane)
            int64_t t = k * (7 * nx * ny - 2 * nx - 2 * ny) - (k > 0 ? nx * ny : 0) - (k == nz ? nx * ny : 0);
            return t;
        };
        int64_t p = before_plane(k0);
        for (int64_t r = k0 * sk; r < k1 * sk; ++r) {
            const int64_t i = r % nx, jl = (r / nx) % ny, kl = r / sk;
            A.indptr[size_t(r)] = int32_t(p);
            const bool present[7] = {kl > 0, jl > 0, i > 0, true, i + 1 < nx, jl + 1 < ny, kl + 1 < nz};
            const int64_t off[7] = {-sk, -sj, -1, 0, 1, sj, sk};
            for (int e = 0; e < 7; ++e)
                if (present[e]) { A.indices[size_t(p)] = int32_t(r + off[e]); A.data[size_t(p)] = g.c[e]; ++p; }
        }
        if (k1 == nz) A.indptr[size_t(n)] = int32_t(p);
    };
    std::vector<std::thread> th;
    for (int tnum = 1; tnum < nt; ++tnum) th.emplace_back(fill, tnum);
    fill(0);
    for (auto &q : th) q.join();
    OMG_REQUIRE(A.indptr[size_t(n)] == A.nnz, "internal: synthesised operator has the wrong number of entries");
    return A;
}

template <typename V>
HostCsr PlanePlan<V>::restriction_csr() const {
    const int64_t nx = g.nx, ny = g.ny, nz = g.nz, n = nx * ny * nz, sj = nx, sk = nx * ny;
    const int per = g.dim2 ? 4 : 8;
    const int64_t nxc = nx / 2, nyc = ny / 2, nc = nxc * nyc * (g.dim2 ? 1 : nz / 2);
    HostCsr R;
    R.n_rows = nc;
    R.n_cols = n;
    R.nnz = n;
    R.indptr.resize(size_t(nc) + 1);
    R.indices.resize(size_t(n));
    R.data.resize(size_t(n));
    for (int64_t cr = 0; cr <= nc; ++cr) R.indptr[size_t(cr)] = int32_t(per * cr);
    for (int64_t cr = 0; cr < nc; ++cr) {
        const int64_t I = cr % nxc, J = (cr / nxc) % nyc, K = cr / (nxc * nyc);
        int64_t p = per * cr;
        for (int dk = 0; dk < (g.dim2 ? 1 : 2); ++dk)
            for (int dj = 0; dj < 2; ++dj)
                for (int di = 0; di < 2; ++di, ++p) {
                    R.indices[size_t(p)] = int32_t((2 * K + dk) * sk + (2 * J + dj) * sj + 2 * I + di);
                    R.data[size_t(p)] = g.w;
                }
    }
    return R;
}

namespace {

template <typename V>
PlaneKArgs<V> plane_args(const PlaneGeom &g, const V *x_old, V *x_new, const V *b, const typename PlanePlan<V>::Coarse &c, uint32_t *status) {
    PlaneKArgs<V> k;
    std::memset(&k, 0, sizeof(k));
    const int64_t n = int64_t(g.nx) * g.ny * g.nz;
    k.x_old = x_old; k.x_new = x_new; k.b = b;
    k.vec_bytes = unsigned(n * int64_t(sizeof(V)));
    k.nr = int(n / 2);
    k.hx = g.hx; k.ny = g.ny; k.nz = g.nz;
    k.z_base = g.z_base; k.z_end = g.z_end; k.kv0 = g.kv0; k.kv1 = g.kv1; k.kc_off = g.kc_off;
    k.TXq = g.TX / 4; k.TY = g.TY; k.LZ = g.LZ; k.PX = g.PX; k.PY = g.PY; k.ntx = g.ntx; k.nty = g.nty; k.ntz = g.ntz;
    k.c0 = V(g.c[0]); k.c1 = V(g.c[1]); k.c2 = V(g.c[2]); k.c3 = V(g.c[3]); k.c4 = V(g.c[4]); k.c5 = V(g.c[5]); k.c6 = V(g.c[6]);
    k.w = V(g.w);
    k.fast_div = (std::fabs(g.c[3]) >= 0x1p-400 && std::fabs(g.c[3]) <= 0x1p400) ? 1 : 0;
    k.nxc = g.nx / 2; k.nyc = g.ny / 2; k.nzc = g.nzc;
    const int64_t nc = int64_t(k.nxc) * k.nyc * k.nzc;
    k.cvec_bytes = unsigned(nc * int64_t(sizeof(V)));
    k.cmap_bytes = unsigned(nc * 4);
    k.cmap = c.map;
    k.bc = c.b; k.xc = c.x; k.cdiag = c.diag; k.first_end = c.first_end; k.ec = c.e;
    k.status = status;
    k.zc_base = g.z_base; k.zc_stride = g.LZ; k.zc_len = g.LZ; k.zc_end = g.z_end; k.part_slot0 = 0;
#ifdef OMG_PLANE_DBG_ON
    {
        const char *e = getenv("OMG_PLANE_DBG");
        k.dbg = e ? atoi(e) : 0;
    }
#endif
    return k;
}

#ifdef OMG_PLANE_STAMPS
template <typename V>
void stamps_begin(PlaneKArgs<V> &k, const PlaneGeom &g, DevBuf<unsigned long long> &buf) {
    buf.alloc(size_t(g.n_wg) * 8 * 8);
    OMG_HIP(hipMemset(buf.p, 0, buf.n * 8));
    k.stamps = buf.p;
}
inline void stamps_end(const char *what, const PlaneGeom &g, DevBuf<unsigned long long> &buf, hipStream_t s) {
    OMG_HIP(hipStreamSynchronize(s));
    std::vector<unsigned long long> hst(buf.n);
    OMG_HIP(hipMemcpy(hst.data(), buf.p, buf.n * 8, hipMemcpyDeviceToHost));
    double m = 0, c = 0, b = 0, steps = 0, tp = 0, sB = 0, sC = 0, tot = 0, nw = 0;
    const int waves = g.threads / 64;
    for (int w = 0; w < g.n_wg; ++w)
        for (int v = 0; v < waves; ++v) {
            const unsigned long long *o = &hst[(size_t(w) * 8 + v) * 8];
            m += o[0]; c += o[1]; b += o[2]; steps += o[3]; tp += o[4]; sB += o[5]; sC += o[6];
            tot += o[7]; nw += 1;
        }
    fprintf(stderr, "[plane stamps] %-5s ... per wave: %.1f steps, %.0f cycles entry to last step's end, %.0f of them outside the steps (100 MHz counter)\n",
            what, steps / nw, tot / nw, (tot - m - c - b - tp - sB - sC) / nw);
    fprintf(stderr, "[plane stamps] %-5s n %dx%dx%d wg %d thr %d: cycles per step and wave: wait-for-loads %.0f | top (shift, issue loads + stores) %.0f | B %.0f | C %.0f | D + rest %.0f | barrier %.0f\n",
            what, g.nx, g.ny, g.nz, g.n_wg, g.threads, m / steps, tp / steps, sB / steps, sC / steps, c / steps, b / steps);
}
#endif

template <typename K>
void allow_lds(K kernel, size_t bytes) {
    if (bytes > size_t(64) * 1024)
        OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, int(bytes)));
}

}  // namespace

// the neighbours of a slab into the kernel's arguments; the wait as a launch of its own where asked for
template <typename V>
void peer_args(PlaneKArgs<V> &k, const typename PlanePlan<V>::Peer &p, hipStream_t s) {
    for (int i = 0; i < 2; ++i) { k.peer_x[i] = p.x[i]; k.peer_bc[i] = p.bc[i]; k.peer_flag[i] = p.flag[i]; }
    k.peer_shift = int(p.shift); k.peer_cshift = int(p.cshift);
    OMG_REQUIRE(p.planes == PEER_PLANES && (p.cplanes == PEER_CPLANES || (!p.bc[0] && !p.bc[1])), "plane pass with neighbours: three planes of x, two of the coarse right-hand side");
    for (int i = 0; i < 4; ++i) { k.wait_flag[i] = p.wait_flag[i]; k.wait_seq[i] = p.wait_seq[i]; }
    k.fused_wait = p.fused_wait ? 1 : 0;
    k.done = p.done; k.flag_seq = p.seq; k.status = p.status; k.spin = p.spin;
    OMG_REQUIRE(p.done, "plane pass with neighbours: no workgroup counter");
    if (!p.fused_wait && (p.wait_flag[0] || p.wait_flag[1] || p.wait_flag[2] || p.wait_flag[3])) {
        hipLaunchKernelGGL(plane_wait_kernel, dim3(1), dim3(64), 0, s, p.wait_flag[0], p.wait_flag[1], p.wait_flag[2], p.wait_flag[3],
                           p.wait_seq[0], p.wait_seq[1], p.wait_seq[2], p.wait_seq[3], p.status, p.spin);
        OMG_HIP(hipGetLastError());
    }
}

// small levels (whole grids of at most 64^3 cells): block_kernel
struct BlockGeom {
    int shape, BX, BY, BZ, nbx, nby, nbz, threads;
    size_t lds;
};
inline bool block_level(const PlaneGeom &g) { return g.block; }
// the block extents block_kernel is instantiated for, largest first
constexpr int BLOCK_SHAPES[4][3] = {{16, 8, 8}, {8, 8, 8}, {8, 4, 4}, {4, 4, 4}};
inline BlockGeom choose_block(const PlaneGeom &g, size_t value_bytes) {
    BlockGeom k;
    int pick = 3;
    // enough workgroups to spread over the chip, from the largest block (least ring per cell) down
    for (int i = 0; i < 4; ++i) {
        const int64_t n = int64_t((g.nx + BLOCK_SHAPES[i][0] - 1) / BLOCK_SHAPES[i][0]) * ((g.ny + BLOCK_SHAPES[i][1] - 1) / BLOCK_SHAPES[i][1]) *
                          ((g.nz + BLOCK_SHAPES[i][2] - 1) / BLOCK_SHAPES[i][2]);
        if (n >= 128) { pick = i; break; }
    }
    k.shape = pick;
    k.BX = BLOCK_SHAPES[pick][0]; k.BY = BLOCK_SHAPES[pick][1]; k.BZ = BLOCK_SHAPES[pick][2];
    k.nbx = (g.nx + k.BX - 1) / k.BX; k.nby = (g.ny + k.BY - 1) / k.BY; k.nbz = (g.nz + k.BZ - 1) / k.BZ;
    const size_t vol = size_t(k.BX + 4) * size_t(k.BY + 4) * size_t(k.BZ + 4);
    k.threads = vol >= 1024 ? 256 : 128;
    k.lds = (2 * vol + std::max(size_t(k.BX) * k.BY * k.BZ, vol / 8)) * value_bytes;
    return k;
}
template <typename V, int MODE, int S, bool SWEEP>
void launch_block_shape(const BlockGeom &k, const BlockKArgs<V> &a, hipStream_t s) {
    auto kernel = block_kernel<V, MODE, BLOCK_SHAPES[S][0], BLOCK_SHAPES[S][1], BLOCK_SHAPES[S][2], SWEEP>;
    allow_lds(kernel, k.lds);
    hipLaunchKernelGGL(kernel, dim3(unsigned(k.nbx * k.nby * k.nbz)), dim3(unsigned(k.threads)), k.lds, s, a);
}
template <typename V, int MODE, bool SWEEP = true>
void launch_block(const PlaneGeom &g, const V *x_old, V *x_new, const V *b, const typename PlanePlan<V>::Coarse &c, hipStream_t s) {
    const BlockGeom k = choose_block(g, sizeof(V));
    BlockKArgs<V> a;
    std::memset(&a, 0, sizeof(a));
    a.x_old = x_old; a.x_new = x_new; a.b = b;
    a.nx = g.nx; a.ny = g.ny; a.nz = g.nz; a.nr = int(int64_t(g.nx) * g.ny * g.nz / 2);
    a.nbx = k.nbx; a.nby = k.nby;
    a.c0 = V(g.c[0]); a.c1 = V(g.c[1]); a.c2 = V(g.c[2]); a.c3 = V(g.c[3]); a.c4 = V(g.c[4]); a.c5 = V(g.c[5]); a.c6 = V(g.c[6]);
    a.w = V(g.w);
    a.fast_div = (std::fabs(g.c[3]) >= 0x1p-400 && std::fabs(g.c[3]) <= 0x1p400) ? 1 : 0;
    a.nxc = g.nx / 2; a.nyc = g.ny / 2;
    a.cmap = c.map; a.bc = c.b; a.ec = c.e;
    switch (k.shape) {
        case 0: launch_block_shape<V, MODE, 0, SWEEP>(k, a, s); break;
        case 1: launch_block_shape<V, MODE, 1, SWEEP>(k, a, s); break;
        case 2: launch_block_shape<V, MODE, 2, SWEEP>(k, a, s); break;
        default: launch_block_shape<V, MODE, 3, SWEEP>(k, a, s); break;
    }
    OMG_HIP(hipGetLastError());
}


template <typename V, int MODE, bool XZ, bool NORM, bool SWEEP>
void launch_tile2d(const PlaneGeom &g, const V *x_old, V *x_new, const V *b, const typename PlanePlan<V>::Coarse &c, double *out, hipStream_t s) {
    Tile2KArgs<V> a;
    std::memset(&a, 0, sizeof(a));
    a.x_old = x_old; a.x_new = x_new; a.b = b;
    a.nx = g.nx; a.ny = g.ny; a.nr = int(int64_t(g.nx) * g.ny / 2);
    a.nbx = g.ntx;
    a.c1 = V(g.c[1]); a.c2 = V(g.c[2]); a.c3 = V(g.c[3]); a.c4 = V(g.c[4]); a.c5 = V(g.c[5]);
    a.w = V(g.w);
    a.omega = V(g.omega);
    a.fast_div = (std::fabs(g.c[3]) >= 0x1p-400 && std::fabs(g.c[3]) <= 0x1p400) ? 1 : 0;
    a.nxc = g.nx / 2; a.nyc = g.ny / 2;
    a.cmap = c.map; a.bc = c.b; a.ec = c.e;
    a.partials = out;
    auto go = [&](auto kernel) {
        allow_lds(kernel, g.lds_bytes);
        hipLaunchKernelGGL(kernel, dim3(unsigned(g.n_wg)), dim3(256), g.lds_bytes, s, a);
    };
    if (g.jacobi) {
        if (g.TX == 64) go(tile2d_kernel<V, MODE, XZ, NORM, SWEEP, 64, 32, true>);
        else if (g.TX == 32) go(tile2d_kernel<V, MODE, XZ, NORM, SWEEP, 32, 16, true>);
        else go(tile2d_kernel<V, MODE, XZ, NORM, SWEEP, 16, 8, true>);
    } else {
        if (g.TX == 64) go(tile2d_kernel<V, MODE, XZ, NORM, SWEEP, 64, 32>);
        else if (g.TX == 32) go(tile2d_kernel<V, MODE, XZ, NORM, SWEEP, 32, 16>);
        else go(tile2d_kernel<V, MODE, XZ, NORM, SWEEP, 16, 8>);
    }
    OMG_HIP(hipGetLastError());
}

inline bool small_tile(const PlaneGeom &g) { return g.la2 && g.threads <= 128; }

// part (PlanePlan::PART_*): the whole pass, or only the slab's first and last PLANE_EDGE planes, or only the planes between
template <typename V>
int plane_part(const PlaneGeom &g, PlaneKArgs<V> &k, int part) {
    const int nxy = g.ntx * g.nty, nzo = g.z_end - g.z_base;
    if (part == 0) return g.n_wg;
    if (part == 1) {
        k.zc_base = g.z_base; k.zc_stride = nzo - PLANE_EDGE; k.zc_len = PLANE_EDGE; k.zc_end = g.z_end; k.part_slot0 = 0;
        return 2 * nxy;
    }
    k.zc_base = g.z_base + PLANE_EDGE; k.zc_stride = g.LZ; k.zc_len = g.LZ; k.zc_end = g.z_end - PLANE_EDGE; k.part_slot0 = 2 * nxy;
    return nxy * ((nzo - 2 * PLANE_EDGE + g.LZ - 1) / g.LZ);
}
template <typename K, typename V>
void launch_plane(K kernel, const PlaneGeom &g, const PlaneKArgs<V> &k, hipStream_t s, int n_wg = -1) {
    if (n_wg == 0) return;
    allow_lds(kernel, g.lds_bytes);
    hipLaunchKernelGGL(kernel, dim3(unsigned(n_wg < 0 ? g.n_wg : n_wg)), dim3(unsigned(g.threads)), g.lds_bytes, s, k);
    OMG_HIP(hipGetLastError());
}

// the one-launch form of a pass whose ghost planes are still on their way (PlanePlan::Gate): inner chunks, then edge chunks
template <typename V>
int plane_gate(const PlaneGeom &g, PlaneKArgs<V> &k, const typename PlanePlan<V>::Gate &gate, int lz) {
    const int nxy = g.ntx * g.nty, nzo = g.z_end - g.z_base;
    k.zc_base = g.z_base + PLANE_EDGE; k.zc_stride = lz; k.zc_len = lz; k.zc_end = g.z_end - PLANE_EDGE; k.part_slot0 = 0;
    k.edge_wg0 = nxy * ((nzo - 2 * PLANE_EDGE + lz - 1) / lz);
    for (int i = 0; i < 4; ++i) { k.wait_flag[i] = gate.flag[i]; k.wait_seq[i] = gate.seq[i]; }
    k.spin = gate.spin;
    if (gate.status) k.status = gate.status;
    return k.edge_wg0 + 2 * nxy;
}

template <typename V>
int PlanePlan<V>::gate_lz() const {
    // as many inner chunks as leave most of the edge workgroups a compute unit of their own from the start (the rest
    // follow the first workgroups that finish): 256 compute units, one workgroup each
    const int nxy = g.ntx * g.nty, nzo = g.z_end - g.z_base;
    static const int forced = [] { const char *e = experiment_env("OMG_PLANE_GATE_LZ"); return e ? atoi(e) : 0; }();
    if (forced >= 2) return forced / 2 * 2;
    const int chunks = std::max(1, (256 - (2 * nxy * 5 + 5) / 6) / nxy);
    const int lz = ((nzo - 2 * PLANE_EDGE + chunks - 1) / chunks + 1) / 2 * 2;
    return std::max(lz, g.LZ);
}

template <typename V>
int PlanePlan<V>::gate_partials() const {
    const int nxy = g.ntx * g.nty, nzo = g.z_end - g.z_base;
    return can_split() ? nxy * (2 + (nzo - 2 * PLANE_EDGE + gate_lz() - 1) / gate_lz()) : g.n_wg;
}

template <typename V>
void PlanePlan<V>::down(const V *x_old, V *x_new, const V *b, bool x_zero, const Coarse &c, hipStream_t s, const Peer *peer, bool sweep, int part,
                        const Gate *gate) const {
    PlaneKArgs<V> k = plane_args<V>(g, x_old, x_new, b, c, status);
    k.x_zero = x_zero ? 1 : 0;
    OMG_REQUIRE(part == 0 || (!peer && sweep && !g.dim2 && !block_level(g) && !c.x && !c.diag && !small_tile(g)),
                "a pass in two launches: slabs' marching passes only");
    OMG_REQUIRE(!gate || (part == 0 && !peer && sweep && can_gate() && !c.x && !c.diag), "a gated pass: thick slabs' marching passes only");
    const int wgs = gate ? plane_gate<V>(g, k, *gate, gate_lz()) : plane_part(g, k, part);
#ifdef OMG_PLANE_STAMPS
    DevBuf<unsigned long long> sb;
    stamps_begin(k, g, sb);
#endif
    if (g.dim2) {
        OMG_REQUIRE(!peer && !c.diag, "2-D plane level: no slab neighbours, no fused first relaxation of the coarse level");
        if (c.x) OMG_HIP(hipMemsetAsync(c.x, 0, size_t(g.nx / 2) * size_t(g.ny / 2) * sizeof(V), s));      // (coarse zero iterate, :191-192)
        if (sweep) {
            if (x_zero) launch_tile2d<V, 0, true, false, true>(g, x_old, x_new, b, c, nullptr, s);
            else launch_tile2d<V, 0, false, false, true>(g, x_old, x_new, b, c, nullptr, s);
        } else {
            if (x_zero) launch_tile2d<V, 0, true, false, false>(g, x_old, x_new, b, c, nullptr, s);
            else launch_tile2d<V, 0, false, false, false>(g, x_old, x_new, b, c, nullptr, s);
        }
    } else if (!sweep) {
        // preIterations = 0: residual of the iterate as it is + restriction; x_new is not written
        OMG_REQUIRE(!peer, "plane pass without its sweep: not built for slabs with neighbours");
        if (x_zero && !c.x && !c.diag && block_level(g)) launch_block<V, 0, false>(g, x_old, x_new, b, c, s);
        else if (x_zero) launch_plane(plane_kernel<V, 0, false, true, PLANE_LA, false, 512, true, false>, g, k, s);
        else launch_plane(plane_kernel<V, 0, false, false, PLANE_LA, false, 512, true, false>, g, k, s);
    } else if (!peer && x_zero && !c.x && !c.diag && block_level(g)) {
        launch_block<V, 0>(g, x_old, x_new, b, c, s);
    } else if (peer) {
        peer_args<V>(k, *peer, s);
        if (x_zero) launch_plane(plane_kernel<V, 0, false, true, PLANE_LA, true>, g, k, s);
        else launch_plane(plane_kernel<V, 0, false, false, PLANE_LA, true>, g, k, s);
    } else if (c.x || c.diag) {
        if (x_zero) launch_plane(plane_kernel<V, 0, false, true, PLANE_LA, false, 512, true>, g, k, s);
        else launch_plane(plane_kernel<V, 0, false, false, PLANE_LA, false, 512, true>, g, k, s);
    } else if (small_tile(g)) {
        // latency-bound small levels: loads two steps ahead (the registers are there for workgroups this small)
        if (x_zero) launch_plane(plane_kernel<V, 0, false, true, 2, false, 128>, g, k, s);
        else launch_plane(plane_kernel<V, 0, false, false, 2, false, 128>, g, k, s);
    } else if (x_zero) {
        launch_plane(plane_kernel<V, 0, false, true, PLANE_LA_BIG>, g, k, s, wgs);
    } else {
        launch_plane(plane_kernel<V, 0, false, false, PLANE_LA_BIG>, g, k, s, wgs);
    }
#ifdef OMG_PLANE_STAMPS
    stamps_end("down", g, sb, s);
#endif
}

template <typename V>
void PlanePlan<V>::up(const V *x_old, V *x_new, const V *b, const Coarse &c, double *out, hipStream_t s, const Peer *peer, bool sweep, int part,
                      const Gate *gate) const {
    PlaneKArgs<V> k = plane_args<V>(g, x_old, x_new, b, c, status);
    k.partials = out;
    OMG_REQUIRE(part == 0 || (!peer && sweep && !g.dim2 && !block_level(g) && !(small_tile(g) && !out)),
                "a pass in two launches: slabs' marching passes only");
    OMG_REQUIRE(!gate || (part == 0 && !peer && sweep && can_gate()), "a gated pass: thick slabs' marching passes only");
    const int wgs = gate ? plane_gate<V>(g, k, *gate, gate_lz()) : plane_part(g, k, part);
    // the up pass of a whole grid marches from the last plane down (plane_kernel MIRROR); OMG_PLANE_MIRROR=0: upwards like the down pass
    const char *mirror_env = getenv("OMG_PLANE_MIRROR");                 // (read per call: A/B runs flip it inside one process)
    const bool mirror_on = !(mirror_env && mirror_env[0] == '0');
    // (slabs too, where the neighbours are reached by exchanges and not by the pass's own stores: as many ghost planes in
    // front of the owned ones as behind them, so the mirrored march covers the same planes; the planes that exist in the
    // global grid, kv0 .. kv1, are handed over in march coordinates)
    const bool mirror = mirror_on && !peer && part == 0 && !g.dim2 && !block_level(g) && !small_tile(g) && (g.nz & 1) == 0 && g.z_base == g.nz - g.z_end;
    if (mirror) { k.kv0 = g.nz - g.kv1; k.kv1 = g.nz - g.kv0; }
#ifdef OMG_PLANE_STAMPS
    DevBuf<unsigned long long> sb;
    stamps_begin(k, g, sb);
#endif
    if (g.dim2) {
        OMG_REQUIRE(!peer, "2-D plane level: no slab neighbours");
        if (sweep) {
            if (out) launch_tile2d<V, 1, false, true, true>(g, x_old, x_new, b, c, out, s);
            else launch_tile2d<V, 1, false, false, true>(g, x_old, x_new, b, c, nullptr, s);
        } else {
            if (out) launch_tile2d<V, 1, false, true, false>(g, x_old, x_new, b, c, out, s);
            else launch_tile2d<V, 1, false, false, false>(g, x_old, x_new, b, c, nullptr, s);
        }
    } else if (!sweep) {
        // postIterations = 0 (the reference's default): x_new = x_old + R^T e (+ the squares of its residual)
        OMG_REQUIRE(!peer, "plane pass without its sweep: not built for slabs with neighbours");
        if (!out && block_level(g)) launch_block<V, 1, false>(g, x_old, x_new, b, c, s);
        else if (mirror && out) launch_plane(plane_kernel<V, 1, true, false, PLANE_LA, false, 512, false, false, true>, g, k, s);
        else if (mirror) launch_plane(plane_kernel<V, 1, false, false, PLANE_LA, false, 512, false, false, true>, g, k, s);
        else if (out) launch_plane(plane_kernel<V, 1, true, false, PLANE_LA, false, 512, false, false>, g, k, s);
        else launch_plane(plane_kernel<V, 1, false, false, PLANE_LA, false, 512, false, false>, g, k, s);
    } else if (!peer && !out && block_level(g)) {
        launch_block<V, 1>(g, x_old, x_new, b, c, s);
    } else if (peer) {
        peer_args<V>(k, *peer, s);
        if (out) launch_plane(plane_kernel<V, 1, true, false, PLANE_LA, true>, g, k, s);
        else launch_plane(plane_kernel<V, 1, false, false, PLANE_LA, true>, g, k, s);
    } else if (small_tile(g) && !out) {
        launch_plane(plane_kernel<V, 1, false, false, 2, false, 128>, g, k, s);
    } else if (mirror) {
        if (out) launch_plane(plane_kernel<V, 1, true, false, PLANE_LA_BIG, false, 512, false, true, true>, g, k, s, wgs);
        else launch_plane(plane_kernel<V, 1, false, false, PLANE_LA_BIG, false, 512, false, true, true>, g, k, s, wgs);
    } else if (out) {
        launch_plane(plane_kernel<V, 1, true, false, PLANE_LA_BIG>, g, k, s, wgs);
    } else {
        launch_plane(plane_kernel<V, 1, false, false, PLANE_LA_BIG>, g, k, s, wgs);
    }
#ifdef OMG_PLANE_STAMPS
    stamps_end("up", g, sb, s);
#endif
}

template <typename V>
void PlanePlan<V>::spmv(const V *x, V *y, hipStream_t s) const {
    OMG_REQUIRE(!g.jacobi && g.z_base == 0 && g.z_end == g.nz, "matrix-free SpMV: whole red-black ordered grids only");
    PlaneSpmvArgs<V> a;
    std::memset(&a, 0, sizeof(a));
    const int64_t n = int64_t(g.nx) * g.ny * g.nz;
    a.x = x; a.y = y;
    a.vec_bytes = unsigned(n * int64_t(sizeof(V)));
    a.nr = int(n / 2); a.hx = g.hx; a.ny = g.ny; a.nz = g.nz; a.hq = (g.hx + 1) / 2;
    a.c0 = V(g.c[0]); a.c1 = V(g.c[1]); a.c2 = V(g.c[2]); a.c3 = V(g.c[3]); a.c4 = V(g.c[4]); a.c5 = V(g.c[5]); a.c6 = V(g.c[6]);
    // the z-marching form (OMG_PLANE_SPMV=0: one thread per output pair, every operand from the cache); OMG_PLANE_SPMV_LZ:
    // planes per chunk
    static const int mode = [] { const char *e = getenv("OMG_PLANE_SPMV"); return e ? atoi(e) : 1; }();
    static const int lz_env = [] { const char *e = getenv("OMG_PLANE_SPMV_LZ"); return e ? atoi(e) : 0; }();
    if (mode != 0 && a.hq <= 256) {
        PlaneSpmvMarchArgs<V> m;
        m.s = a;
        int qt = 1;
        while (qt < a.hq) qt *= 2;                                  // pairs per workgroup row: a power of two >= the half line's pairs
        static const int wg_threads = [] { const char *e = experiment_env("OMG_PLANE_SPMV_T"); return (e && atoi(e) == 512) ? 512 : 256; }();
        qt = std::min(qt, wg_threads);
        m.qt = qt; m.lpw = wg_threads / qt;
        m.nqt = (a.hq + qt - 1) / qt;
        m.njg = (g.ny + m.lpw - 1) / m.lpw;
        // enough workgroups to fill the chip several times, chunks as long as that allows (two ring planes per chunk)
        int lz = lz_env > 0 ? lz_env : 64;
        while (lz > 4 && int64_t(m.njg) * m.nqt * ((g.nz + lz - 1) / lz) < 512) lz /= 2;
        m.lz = lz;
        const int64_t wgs = int64_t(m.njg) * m.nqt * ((g.nz + lz - 1) / lz);
        hipLaunchKernelGGL(plane_spmv_march_kernel<V>, dim3(unsigned(wgs)), dim3(unsigned(wg_threads)), 0, s, m);
        OMG_HIP(hipGetLastError());
        return;
    }
    const int64_t threads = 2 * int64_t(a.hq) * g.ny * g.nz;
    hipLaunchKernelGGL(plane_spmv_kernel<V>, dim3(unsigned((threads + 255) / 256)), dim3(256), 0, s, a);
    OMG_HIP(hipGetLastError());
}

template struct PlanePlan<double>;
template struct PlanePlan<float>;

}  // namespace omg
