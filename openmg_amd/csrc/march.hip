// Lexicographic Gauss-Seidel of a grid star stencil as ONE launch per sweep (common.h MarchPlan).
//
// Reference: openmg/solvers.py:56-68 — for i in range(n): x[i] += (b[i] - A[i, :] x) / A[i, i], rows
// in their natural order.  The level schedule of hierarchy.hip runs that as one launch per set of
// mutually uncoupled rows (766 sets for 256^3); here the dependent steps stay inside a launch:
//
//   * a wave owns TJ x TK grid lines; lane (jj, kk) relaxes row i = t - jj - kk of its line at step
//     t, so the -I neighbour is the lane's own previous result, the -J / -K neighbours are what
//     lane - 1 / lane - TJ produced one step earlier (two shuffles), the +I neighbour is the next
//     old value of the lane's own line, and the +J / +K neighbours are the +I operands of
//     lane + 1 / lane + TJ (two more shuffles, off the dependent chain);
//   * per step and lane that is one row: the seven-slot fma chain in stored (= column) order with
//     the coefficients of the row's pattern from LDS, then x_i + (b_i - sum) / a_ii — the
//     expression of csr_kernels.hip's ROW_GS, hence the same bits as the level schedule;
//   * operands that live in another tile's lines: the not yet relaxed ones are old values of x, loaded a
//     block of steps ahead like the lane's own line and b; the relaxed ones come through FACE SLOTS in
//     HBM, one per row of a tile's last lines: unset (a marker NaN) between sweeps, written
//     write-through by the owning tile, loaded a block ahead by the tile that needs them and polled
//     again while still unset (all of that with agent-scope accesses that bypass the L1 and the other
//     XCDs' L2: MI355X_MICROARCH.md, inter-workgroup visibility), then reset.  Tiles are numbered by
//     a ticket counter, so the tiles a tile waits for have always started; a wave gives up after a
//     bounded number of polls (MarchPlan::timed_out) instead of hanging the device.
//
// Measured (profiles/r02_march_*): one step of a tile alone is 0.20 us (the dependent chain: a shuffle
// round trip through LDS, 72 cycles, + ~25 dependent double operations — tools/lat_probe.hip), a hop
// from tile to tile 8-13 us (the 7 steps of skew, whole blocks published and consumed, and mostly the
// chain of latencies of a hand-over through HBM: finer-grained and streaming variants, DESIGN.md
// section 6, did not shorten it); 256^3: 62 hops, 1.26 ms per sweep against 3.4 ms for the 766
// launches of the level schedule.
#include <algorithm>
#include <array>
#include <atomic>
#include <cmath>
#include <cstring>
#include <thread>
#include <type_traits>

#include "common.h"

namespace omg {

namespace {

#ifndef MARCH_NAP_MAX
#define MARCH_NAP_MAX 64             // longest sleep (x 64 cycles) between two polls of a tile that waits for its first faces
#endif
constexpr int SYNC_WORDS = 16;     // ticket, finished tiles, error flag
constexpr int SPIN_LIMIT = 1 << 20;
constexpr int STORE_SPIN_LIMIT = 1 << 22;   // the storing wave's polls of its own computing wave (LDS), most of them ~1 us apart
constexpr int FACE_PAD = MARCH_FACE_PAD;

template <typename V>
struct MarchArgs {
    V *x;
    const V *b;
    const uint32_t *codes;
    const V *coef;
    const V *rowc;       // per-row mode: [tile][step][8][lane]
    const int32_t *order;
    uint32_t *sync;
    V *faceJ, *faceK;
    long long *dbg;      // OMG_MARCH_DEBUG=1: per tile start, end, time in face waits, polls (wall_clock64 ticks)
    int nx, ny, nz, TJ, ntj, n_tiles, T, n_grp, n_pat, n;
};

__device__ __forceinline__ double madd(double v, double x, double acc) { return fma(v, x, acc); }
__device__ __forceinline__ float madd(float v, float x, float acc) { return fmaf(v, x, acc); }

template <typename T>
__device__ __forceinline__ T load_through(const T *p) {
    return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}
template <typename T>
__device__ __forceinline__ void store_through(T *p, T v) {
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// "Not written yet" in a face slot: a NaN no arithmetic produces (hardware NaNs are canonical).
template <typename V> struct Unset;
template <> struct Unset<double> {
    static constexpr uint64_t bits = 0x7ff8c0dec0de0001ull;
    __host__ __device__ static double value() { union { uint64_t u; double d; } c; c.u = bits; return c.d; }
    __device__ static bool is(double v) { return (uint64_t)__double_as_longlong(v) == bits; }
};
template <> struct Unset<float> {
    static constexpr uint32_t bits = 0x7fc0c0deu;
    __host__ __device__ static float value() { union { uint32_t u; float f; } c; c.u = bits; return c.f; }
    __device__ static bool is(float v) { return (uint32_t)__float_as_int(v) == bits; }
};

// 1 / d as the compiler's expansion of a double division forms it from d alone (v_rcp_f64 and two
// Newton steps), for operands that v_div_scale_f64 leaves unscaled
__device__ __forceinline__ double refined_rcp(double d) {
    double r = __builtin_amdgcn_rcp(d);
    double e = fma(-d, r, 1.0);
    r = fma(r, e, r);
    e = fma(-d, r, 1.0);
    return fma(r, e, r);
}
__device__ __forceinline__ float refined_rcp(float) { return 0.0f; }
// exponent within 2^-400 .. 2^400: neither operand is scaled, no fix-up case applies
__device__ __forceinline__ bool plain_range(double v) {
    const unsigned e = ((unsigned)__double2hiint(v) >> 20) & 0x7ffu;
    return e - 623u <= 800u;
}

__device__ __forceinline__ bool plain_range(float) { return true; }

// n / d for the relaxation.  double, FAST: the quotient the compiler's division sequence forms, with
// its denominator half (r = refined_rcp(d), per pattern) taken off the dependent chain — valid for
// numerators in plain_range(); a block that met another one is repeated with the division itself.
template <bool FAST>
__device__ __forceinline__ double quotient(double n, double d, double r) {
    if (!FAST) return n / d;
    const double q = n * r;
    const double rem = fma(-d, q, n);
    return fma(rem, r, q);
}
template <bool FAST>
__device__ __forceinline__ float quotient(float n, float d, float) { return n / d; }

// the eight table entries of a pattern by 16-byte LDS reads
__device__ __forceinline__ void load_coefs(const double *c, double (&o)[8]) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d *p = reinterpret_cast<const v2d *>(c);
    const v2d a = p[0], b = p[1], e = p[2], f = p[3];
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = e.x; o[5] = e.y; o[6] = f.x; o[7] = f.y;
}
__device__ __forceinline__ void load_coefs(const float *c, float (&o)[8]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    const v4f *p = reinterpret_cast<const v4f *>(c);
    const v4f a = p[0], b = p[1];
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = b.z; o[7] = b.w;
}

// x[q] / b[q] through a buffer descriptor: one instruction per load, and a row index outside the
// vector (the steps of a lane before / behind its line, wrapped when negative) returns 0
__device__ __forceinline__ double buffer_at(__amdgpu_buffer_rsrc_t rs, int q, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, int(unsigned(q) * 8u), 0, 0));
}
__device__ __forceinline__ float buffer_at(__amdgpu_buffer_rsrc_t rs, int q, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, int(unsigned(q) * 4u), 0, 0));
}

// value of lane - 1 / lane + 1 (lane 0 / 63 keep their own): one DPP move per dword, no LDS round trip
template <int CTRL>
__device__ __forceinline__ double dpp_wave_shift(double v) {
    int lo = __double2loint(v), hi = __double2hiint(v);
    lo = __builtin_amdgcn_update_dpp(lo, lo, CTRL, 0xf, 0xf, false);
    hi = __builtin_amdgcn_update_dpp(hi, hi, CTRL, 0xf, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL>
__device__ __forceinline__ float dpp_wave_shift(float v) {
    const int w = __float_as_int(v);
    return __int_as_float(__builtin_amdgcn_update_dpp(w, w, CTRL, 0xf, 0xf, false));
}
constexpr int DPP_WAVE_SHR1 = 0x138, DPP_WAVE_SHL1 = 0x130;

template <typename V, int U>
struct BlockData {
    V xs[U];     // own line, old value of row i + 1 at step t
    V bv[U];     // right-hand side of row i
    V ej[U];     // the -J (lane jj == 0: another tile's face) or +J (jj == TJ - 1: old value) operand
    V ek[U];     // the same for K
    uint32_t codes[U / 4];
};

constexpr int RING = 4;            // blocks of results between the computing wave and the storing wave

// One workgroup = one tile = two waves with the same lane -> line mapping.
//   * The COMPUTING wave loads, relaxes, and never stores to global memory: with stores in flight every
//     wait for a load would also wait for them (one counter for both kinds, out of order between
//     them), and a write-through face store takes about a microsecond.  It hands each block's results
//     to the STORING wave through an LDS ring; that wave writes x, the face slots the +J / +K tiles
//     wait for, and "unset" into the slots this tile has consumed.
//   * ROWS (per-row coefficients): a third, LOADING wave streams the rows' coefficients — [tile][step][8][lane] in HBM, unit
//     stride across the wave — into an LDS ring CRING blocks deep (the launch's dynamic LDS, which also keeps a second
//     worker off the compute unit), one block ahead of the computing wave, and forms each row's refined reciprocal on the way;
//     the computing wave reads its row's eight entries where the pattern table's entries were read.
constexpr int CRING = 3;
template <typename V, int U, bool ROWS>
__global__ __launch_bounds__(ROWS ? 192 : 128) void march_gs_kernel(MarchArgs<V> a) {
    constexpr int NTH = ROWS ? 192 : 128;
    __shared__ V s_coef[ROWS ? 8 : 256 * 8];
    __shared__ V s_ring[RING][U][64];
    __shared__ int s_tile;
    __shared__ int s_ready, s_taken;           // blocks the computing wave has put into the ring / the storing wave has taken out
    __shared__ int s_cready;                   // ROWS: blocks of coefficients the loading wave has put into its ring
    extern __shared__ __attribute__((aligned(16))) unsigned char march_dyn[];
    V *const s_cring = reinterpret_cast<V *>(march_dyn);      // ROWS: [CRING][U][8][64]
    const int lane = threadIdx.x & 63;
    const bool storer = threadIdx.x >= 64 && threadIdx.x < 128;
    const bool loader = ROWS && threadIdx.x >= 128;
    if (!ROWS) {
        for (int q = threadIdx.x; q < a.n_pat * 8; q += NTH) {
            V c = a.coef[q];
            if ((q & 7) == 7) c = refined_rcp(a.coef[q - 4]);
            s_coef[q] = c;
        }
    }
    // A workgroup is a WORKER: it takes tiles by ticket until none is left.  Tickets run along the wavefront (a.order:
    // anti-diagonals of the tile grid), so the tiles a tile takes faces from (earlier tickets) are done or in some
    // worker's hands, and with about one worker per CU the tiles that hold a CU are the ones next to run.  (With every
    // tile a workgroup of its own, 512 of them were resident two to a CU from the start — in tile-number order the
    // first sixteen tile rows — polling beside the few that could run: a tile took 98 us instead of 62, a hop 13.7
    // instead of 9, and a tile of row 18 that could have run at 250 us found no CU before 340.)
    for (;;) {
    __syncthreads();                                   // (the previous tile's last use of the LDS flags and rings)
    if (threadIdx.x == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.sync + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        s_tile = int(t) < a.n_tiles ? (a.order ? a.order[t] : (int)t) : -1;
        s_ready = 0;
        s_taken = 0;
        s_cready = 0;
    }
    __syncthreads();
    const int tile = __builtin_amdgcn_readfirstlane(s_tile);
    if (tile < 0) break;
    const int TJ = a.TJ, TK = 64 / TJ;
    const int J = tile % a.ntj, K = tile / a.ntj;
    const int jj = lane % TJ, kk = lane / TJ;
    const int j = J * TJ + jj, k = K * TK + kk;
    const bool valid = j < a.ny && k < a.nz;
    const bool whole = (J + 1) * TJ <= a.ny && (K + 1) * TK <= a.nz;     // every lane of the tile has a line
    const int skew = jj + kk, max_skew = TJ + TK - 2;
    const int nx = a.nx, nxp = nx + 2 * FACE_PAD;
    const int line = valid ? (k * a.ny + j) * nx : 0;
    // Operands in other tiles' lines.  Relaxed ones (-J, -K) come from the face slots the owning tile
    // fills; not yet relaxed ones (+J, +K) are old values of x.
    const bool lowJ = valid && jj == 0 && j > 0, highJ = valid && jj == TJ - 1 && j + 1 < a.ny;
    const bool lowK = valid && kk == 0 && k > 0, highK = valid && kk == TK - 1 && k + 1 < a.nz;
    const int offJ = nx, offK = nx * a.ny;
    // (a face line carries FACE_PAD slots in front of row 0 and behind row nx - 1: the steps of a lane
    // that fall outside its line address those, so that nothing there is predicated on the row)
    V *inJ = a.faceJ + (size_t(lowJ ? tile - 1 : tile) * TK + kk) * nxp + FACE_PAD;
    V *inK = a.faceK + (size_t(lowK ? tile - a.ntj : tile) * TJ + jj) * nxp + FACE_PAD;
    V *outJ = a.faceJ + (size_t(tile) * TK + kk) * nxp + FACE_PAD;
    V *outK = a.faceK + (size_t(tile) * TJ + jj) * nxp + FACE_PAD;
    const V unset = Unset<V>::value();
    const int n_blk = (a.T + U - 1) / U;
    int spins = 0;
    long long t_begin = 0, t_wait = 0;
    if (a.dbg) t_begin = wall_clock64();
    auto lds_flag = [&](int *f) { return __hip_atomic_load(f, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };
    auto lds_set = [&](int *f, int v) { __hip_atomic_store(f, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); };

    if (loader) {
        // block blk of this tile's coefficients into ring slot blk % CRING, once the computing wave is done with block
        // blk - CRING (s_ready counts the blocks it has finished)
        const V *src = a.rowc + size_t(tile) * size_t(n_blk * U) * 512 + lane;
        for (int blk = 0; blk < n_blk; ++blk) {
            int polls = 0;
            while (lds_flag(&s_ready) < blk - CRING + 1 && polls < STORE_SPIN_LIMIT) {
                if (polls < 4096) __builtin_amdgcn_s_sleep(1);
                else __builtin_amdgcn_s_sleep(32);
                ++polls;
            }
            if (polls >= STORE_SPIN_LIMIT) spins = SPIN_LIMIT;
            asm volatile("" ::: "memory");
            V *const dst = s_cring + size_t(blk % CRING) * (U * 512) + lane;
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                V v[16];
#pragma unroll
                for (int e = 0; e < 16; ++e) v[e] = src[(size_t(blk) * U + u) * 512 + e * 64];
                v[7] = refined_rcp(v[3]);
                v[15] = refined_rcp(v[11]);
#pragma unroll
                for (int e = 0; e < 16; ++e) dst[u * 512 + e * 64] = v[e];
            }
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_set(&s_cready, blk + 1);
        }
        if (__any(spins >= SPIN_LIMIT) && lane == 0) store_through(a.sync + 2, 1u);
    } else if (storer) {
        const __amdgpu_buffer_rsrc_t xw = __builtin_amdgcn_make_buffer_rsrc(a.x, 0, unsigned(a.n) * unsigned(sizeof(V)), 0x00020000);
        for (int blk = 0; blk < n_blk; ++blk) {
            const int i0 = blk * U - skew;
            long long t0 = 0;
            if (a.dbg) t0 = wall_clock64();
            // (this wave starts polling when the tile starts and has to outlast the computing wave's whole wait
            // for the tile's predecessors: back off to ~1 us per poll after a while, and give it seconds)
            int polls = 0;
            while (lds_flag(&s_ready) <= blk && polls < STORE_SPIN_LIMIT) {
                if (polls < 4096) __builtin_amdgcn_s_sleep(1);
                else __builtin_amdgcn_s_sleep(32);
                ++polls;
            }
            if (polls >= STORE_SPIN_LIMIT) spins = SPIN_LIMIT;
            if (a.dbg) t_wait += wall_clock64() - t0;
            asm volatile("" ::: "memory");
            V out[U];
#pragma unroll
            for (int u = 0; u < U; ++u) out[u] = s_ring[blk % RING][u][lane];
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (lane == 0) lds_set(&s_taken, blk + 1);
            // the faces the +J / +K tiles wait for
            if (highJ) {
#pragma unroll
                for (int u = 0; u < U; ++u) store_through(outJ + i0 + u, out[u]);
            }
            if (highK) {
#pragma unroll
                for (int u = 0; u < U; ++u) store_through(outK + i0 + u, out[u]);
            }
            const bool full = whole && blk * U >= max_skew && blk * U + U <= nx;
            if (full) {
                // 16 bytes per lane and instruction: every lane writes its own cache line, and the
                // address unit takes a lane per cycle whatever the width — it is what several tiles
                // sharing a CU run short of
                typedef int v4i_t __attribute__((ext_vector_type(4)));
                constexpr int PER = 16 / int(sizeof(V));
#pragma unroll
                for (int u = 0; u < U; u += PER) {
                    union { V v[PER]; v4i_t q; } pack;
#pragma unroll
                    for (int e = 0; e < PER; ++e) pack.v[e] = out[u + e];
                    __builtin_amdgcn_raw_buffer_store_b128(pack.q, xw, int(unsigned(line + i0 + u) * unsigned(sizeof(V))), 0, 0);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u)
                    if (valid && i0 + u >= 0 && i0 + u < nx) a.x[line + i0 + u] = out[u];
            }
            // the slots the computing wave consumed for this block are left unset for the next sweep
            if (lowJ) {
#pragma unroll
                for (int u = 0; u < U; ++u) store_through(inJ + i0 + u, unset);
            }
            if (lowK) {
#pragma unroll
                for (int u = 0; u < U; ++u) store_through(inK + i0 + u, unset);
            }
        }
        if (__any(spins >= SPIN_LIMIT) && lane == 0) store_through(a.sync + 2, 1u);
        if (a.dbg && lane == 0) {
            a.dbg[8 * tile + 5] = wall_clock64();
            a.dbg[8 * tile + 6] = t_wait;
        }
    } else {
    long long t_ring = 0;

    const unsigned vec_bytes = unsigned(a.n) * unsigned(sizeof(V));
    const __amdgpu_buffer_rsrc_t xr = __builtin_amdgcn_make_buffer_rsrc(a.x, 0, vec_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t br = __builtin_amdgcn_make_buffer_rsrc(const_cast<V *>(a.b), 0, vec_bytes, 0x00020000);
    typedef BlockData<V, U> Block;
    // what a block needs of the tile's own lines and of lines nobody has relaxed yet (nothing here waits for another tile) ...
    auto prefetch_own = [&](int blk, Block &d) {
        const int i0 = blk * U - skew;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            d.xs[u] = buffer_at(xr, line + i0 + u + 1, V(0));
            d.bv[u] = buffer_at(br, line + i0 + u, V(0));
            d.ej[u] = V(0);
            d.ek[u] = V(0);
        }
        if (highJ) {
#pragma unroll
            for (int u = 0; u < U; ++u) d.ej[u] = buffer_at(xr, line + i0 + u + offJ, V(0));
        }
        if (highK) {
#pragma unroll
            for (int u = 0; u < U; ++u) d.ek[u] = buffer_at(xr, line + i0 + u + offK, V(0));
        }
        if (!ROWS) {
#pragma unroll
            for (int w = 0; w < U / 4; ++w) d.codes[w] = a.codes[(size_t(tile) * a.n_grp + blk * (U / 4) + w) * 64 + lane];
        }
    };
    // ... and the relaxed values of the -J / -K tiles' last lines (face slots)
    auto prefetch_faces = [&](int blk, Block &d) {
        const int i0 = blk * U - skew;
        if (lowJ) {
#pragma unroll
            for (int u = 0; u < U; ++u) d.ej[u] = load_through(inJ + i0 + u);
        }
        if (lowK) {
#pragma unroll
            for (int u = 0; u < U; ++u) d.ek[u] = load_through(inK + i0 + u);
        }
    };
    auto prefetch = [&](int blk, Block &d) {
        prefetch_own(blk, d);
        prefetch_faces(blk, d);
    };

    V xcur = a.x[line];                    // old value of the row of the lane's next step (row 0 first)
    V xlast = V(0);                        // the lane's newest result

    // U steps; FAST (double only): returns whether some row's numerator left quotient<true>'s range
    // UNI: every row of the block has the same pattern, its table entries are in cu[] (no LDS reads)
    V cu[8];
    int cu_code = -1;
    const V *cring_blk = s_cring + lane;       // ROWS: the current block's coefficients in the loading wave's ring
    auto steps = [&](auto full_tag, auto fast_tag, auto uni_tag, int i0, const Block &cur, V (&out)[U]) {
        constexpr bool FULL = decltype(full_tag)::value;
        constexpr bool FAST = decltype(fast_tag)::value;
        constexpr bool UNI = decltype(uni_tag)::value;
        bool bad = false;
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = i0 + u;
            const bool act = FULL || (valid && i >= 0 && i < nx);
            V c[8];
            if (ROWS) {
#pragma unroll
                for (int e = 0; e < 8; ++e) c[e] = cring_blk[u * 512 + e * 64];
            } else if (UNI) {
#pragma unroll
                for (int e = 0; e < 8; ++e) c[e] = cu[e];
            } else {
                const int code = int((cur.codes[u / 4] >> (8 * (u & 3))) & 255u);
                load_coefs(s_coef + code * 8, c);
            }
            const V c0 = c[0], c1 = c[1], c2 = c[2], c3 = c[3], c4 = c[4], c5 = c[5], c6 = c[6], rc = c[7];
            V xjm = dpp_wave_shift<DPP_WAVE_SHR1>(xlast), xkm = xlast;
            if (TK > 1) xkm = __shfl_up(xlast, TJ);
            if (jj == 0) xjm = cur.ej[u];
            if (kk == 0) xkm = cur.ek[u];
            const V xip = cur.xs[u];
            V xjp = dpp_wave_shift<DPP_WAVE_SHL1>(xip), xkp = xip;
            if (TK > 1) xkp = __shfl_down(xip, TJ);
            if (jj == TJ - 1) xjp = cur.ej[u];
            if (kk == TK - 1) xkp = cur.ek[u];
            V sum = madd(c0, xkm, V(0));
            sum = madd(c1, xjm, sum);
            sum = madd(c2, xlast, sum);
            sum = madd(c3, xcur, sum);
            sum = madd(c4, xip, sum);
            sum = madd(c5, xjp, sum);
            sum = madd(c6, xkp, sum);
            const V num = cur.bv[u] - sum;
            if (FAST) bad = bad || (act && !plain_range(num));
            const V xn = xcur + quotient<FAST>(num, c3, rc);      // csr_kernels.hip ROW_GS
            out[u] = xn;
            xlast = act ? xn : xlast;
            xcur = xip;
        }
        return bad;
    };
    constexpr bool HAS_FAST = std::is_same<V, double>::value;
    auto steps_checked = [&](auto full_tag, auto uni_tag, int i0, const Block &cur, V (&out)[U]) {
        if constexpr (HAS_FAST) {
            const V xl0 = xlast, xc0 = xcur;
            if (__any(steps(full_tag, std::true_type(), uni_tag, i0, cur, out))) {
                xlast = xl0;
                xcur = xc0;
                steps(full_tag, std::false_type(), uni_tag, i0, cur, out);
            }
        } else {
            steps(full_tag, std::false_type(), uni_tag, i0, cur, out);
        }
    };
    // one pattern for all rows of the block?  (then its entries are fetched once, and kept while it stays the same)
    auto uniform_pattern = [&](const Block &cur) {
        const uint32_t w0 = __builtin_amdgcn_readfirstlane(cur.codes[0]);
        bool same = w0 == (w0 & 255u) * 0x01010101u;
#pragma unroll
        for (int w = 0; w < U / 4; ++w) same = same && cur.codes[w] == w0;
        if (!__all(same)) return false;
        if (int(w0 & 255u) != cu_code) {
            cu_code = int(w0 & 255u);
            load_coefs(s_coef + cu_code * 8, cu);
        }
        return true;
    };

    int taken_seen = 0, cready_seen = 0;
    auto wait_at_least = [&](int *flag, int need, int &seen) {
        if (seen >= need) return;
        long long t0 = 0;
        if (a.dbg) t0 = wall_clock64();
        while (seen < need && spins < SPIN_LIMIT) {
            seen = lds_flag(flag);
            if (seen < need) { __builtin_amdgcn_s_sleep(1); ++spins; }
        }
        if (a.dbg) t_ring += wall_clock64() - t0;
        asm volatile("" ::: "memory");
    };
    auto block = [&](int blk, Block &cur, Block &nxt) {
        const int T0 = blk * U, i0 = T0 - skew;
        // 1. the face operands of this block were loaded a block ago: wait for those their tiles had not written yet
        if (lowJ || lowK) {
            auto missing = [&]() {
                bool m = false;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool in = i0 + u >= 0 && i0 + u < nx;
                    m = m || (in && ((lowJ && Unset<V>::is(cur.ej[u])) || (lowK && Unset<V>::is(cur.ek[u]))));
                }
                return m;
            };
            long long t0 = 0;
            if (a.dbg) t0 = wall_clock64();
            int nap = 0;
            while (__any(missing()) && spins < SPIN_LIMIT) {
                if (nap == 0) __builtin_amdgcn_s_sleep(1);
                else if (nap == 1) __builtin_amdgcn_s_sleep(4);
                else __builtin_amdgcn_s_sleep(16);
                ++nap;
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    if (lowJ) cur.ej[u] = load_through(inJ + i0 + u);
                    if (lowK) cur.ek[u] = load_through(inK + i0 + u);
                }
                ++spins;
            }
            if (a.dbg) t_wait += wall_clock64() - t0;
        }
        // 2. operands of the next block
        if (blk + 1 < n_blk) prefetch(blk + 1, nxt);
        // 3. U steps
        V out[U];
        const bool full = whole && T0 >= max_skew && T0 + U <= nx;     // every lane is inside its line for all U steps
        if (ROWS) {
            wait_at_least(&s_cready, blk + 1, cready_seen);
            cring_blk = s_cring + size_t(blk % CRING) * (U * 512) + lane;
        }
        if (!ROWS && full && uniform_pattern(cur)) steps_checked(std::true_type(), std::true_type(), i0, cur, out);
        else if (full) steps_checked(std::true_type(), std::false_type(), i0, cur, out);
        else steps_checked(std::false_type(), std::false_type(), i0, cur, out);
        // 4. hand the results to the storing wave
        wait_at_least(&s_taken, blk - RING + 1, taken_seen);
#pragma unroll
        for (int u = 0; u < U; ++u) s_ring[blk % RING][u][lane] = out[u];
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        if (lane == 0) lds_set(&s_ready, blk + 1);
    };

    // The first block's own operands are requested BEFORE the wait for the neighbours: they arrive while it lasts
    // (a hop from tile to tile is ~2 us of load latency shorter: 62 hops at 256^3).
    Block A, B;
    prefetch_own(0, A);
    // Then: wait, politely, until the tiles this one takes faces from have produced
    // their first rows.  Hundreds of tiles sit here when a sweep starts; polling all the slots of a
    // block from each of them takes memory bandwidth from the few tiles that can run ("255 pollers cut
    // chip bandwidth 37-71 %", MI355X_MICROARCH.md), so this polls ONE slot per lane and sleeps
    // longer and longer (up to ~1.7 us) in between.
    if (lowJ || lowK) {
        long long t0 = 0;
        if (a.dbg) t0 = wall_clock64();
        int nap = 1;
        for (;;) {
            bool wait = false;
            if (lowJ) wait = Unset<V>::is(load_through(inJ));
            if (lowK) wait = wait || Unset<V>::is(load_through(inK));
            if (!__any(wait) || spins >= SPIN_LIMIT) break;
            if (nap <= 1) __builtin_amdgcn_s_sleep(2);
            else if (nap <= 2) __builtin_amdgcn_s_sleep(8);
            else if (nap <= 4) __builtin_amdgcn_s_sleep(24);
            else __builtin_amdgcn_s_sleep(MARCH_NAP_MAX);
            ++nap;
            ++spins;
        }
        if (a.dbg) t_wait += wall_clock64() - t0;
    }
    prefetch_faces(0, A);
    for (int blk = 0; blk < n_blk; blk += 2) {
        block(blk, A, B);
        if (blk + 1 < n_blk) block(blk + 1, B, A);
    }
    if (__any(spins >= SPIN_LIMIT) && lane == 0) store_through(a.sync + 2, 1u);
    if (a.dbg && lane == 0) {
        a.dbg[8 * tile + 0] = t_begin;
        a.dbg[8 * tile + 1] = wall_clock64();
        a.dbg[8 * tile + 2] = t_wait;
        a.dbg[8 * tile + 3] = spins;
        a.dbg[8 * tile + 4] = t_ring;
    }
    }   // computing wave
    }   // next tile
    // the last worker out leaves the counters as the next sweep expects them
    if (threadIdx.x == 0) {
        const unsigned d = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (d == gridDim.x - 1) {
            store_through(a.sync + 1, 0u);
            store_through(a.sync + 0, 0u);
        }
    }
}

// ---- 1-D grids: the sweep as a first-order recurrence on ONE wave ---------------------------------------------------
// openmg/solvers.py:56-68 on a tridiagonal operator: x_i <- x_i + (b_i - (a_i,i-1 x_i-1 + a_ii x_i + a_i,i+1 x_i+1)) / a_ii
// with the NEW x_i-1 — every row waits for its predecessor, so the sweep is one chain of n links (a scan would associate
// differently: not the reference's bits).  The tiled wavefront kernel above spends 250 ns per link on a 1-D grid (face
// slots, shuffles, one lane busy); here a wave holds a block of 64 rows in registers (lane l: row base + l, loaded
// coalesced, the next block requested a block ahead), every lane runs the SAME chain on wave-uniform operands
// (v_readlane), and what is left on the dependent path per row is eight double operations: the stored-order chain from
// +0 (three fmas), b - s, the quotient with the denominator's half hoisted (refined reciprocal per row, formed when the
// block is loaded: the same instructions on the same operands as the division, march_gs_kernel's rule), x + q.
// Every lane runs the SAME chain on operands all lanes read from ONE LDS address (a broadcast read: no bank conflict, no
// vector-ALU instruction — with v_readlane the row's twelve operand dwords cost twelve more issue slots than the chain's
// eight operations): the wave stages a block of 64 rows in LDS as [row][lo, di, up, r, x, b] (coalesced global loads, the
// block after next requested meanwhile), walks it, leaves each row's result in LDS, and stores the block coalesced.
// The quotient: the hoisted-reciprocal form for every row; afterwards the lanes form their own rows' numerators once
// more IN PARALLEL (the same chain on the same operands: the same bits) and test them — a block in which some numerator
// is outside the range that needs no scaling (or zero) is walked again with the division itself (march_gs_kernel's rule).
constexpr int LINE_W = 8;            // values per staged row (six used; 64-byte rows: four 16-byte reads)
template <typename V>
struct LineRow {
    V lo, di, up, r, x, b;
};
template <typename V>
__device__ __forceinline__ LineRow<V> line_fetch(const V *x, const V *b, const V *tri, int n, int row) {
    LineRow<V> k;
    const bool ok = row < n;
    k.x = ok ? x[row] : V(0);
    k.b = ok ? b[row] : V(0);
    k.lo = ok ? tri[row] : V(0);
    k.di = ok ? tri[size_t(n) + row] : V(1);
    k.up = ok ? tri[2 * size_t(n) + row] : V(0);
    k.r = V(0);
    return k;
}
template <typename V>
__device__ __forceinline__ void line_stage(V *st, const LineRow<V> &k) {
    st[0] = k.lo; st[1] = k.di; st[2] = k.up; st[3] = k.r; st[4] = k.x; st[5] = k.b;
}

template <typename V>
__global__ __launch_bounds__(64) void line_gs_kernel(V *x, const V *b, const V *tri, int n) {
    __shared__ __attribute__((aligned(16))) V s_rows[2][65][LINE_W];      // two blocks; row 64: the next block's first x
    __shared__ V s_out[64];
    const int lane = int(threadIdx.x);
    V prev = V(0);                                     // the relaxed value of the previous row (wave-uniform)
    LineRow<V> cur = line_fetch(x, b, tri, n, lane);
    LineRow<V> nxt = line_fetch(x, b, tri, n, 64 + lane);
    cur.r = refined_rcp(cur.di);
    line_stage(&s_rows[0][lane][0], cur);
    for (int base = 0, buf = 0; base < n; base += 64, buf ^= 1) {
        // the next block into the other half (its first x is row 63's +I operand), the one after it requested
        nxt.r = refined_rcp(nxt.di);
        line_stage(&s_rows[buf ^ 1][lane][0], nxt);
        if (lane == 0) s_rows[buf][64][4] = nxt.x;
        const LineRow<V> after = line_fetch(x, b, tri, n, base + 128 + lane);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        const int rows = min(64, n - base);
        const V (*R)[LINE_W] = s_rows[buf];
        auto walk = [&](auto FAST, V p) -> V {
            struct Ops { V lo, di, up, r, xi, bi; };
            auto ops_of = [&](int j) -> Ops { return Ops{R[j][0], R[j][1], R[j][2], R[j][3], R[j][4], R[j][5]}; };
            auto link = [&](int j, const Ops &o, V xn) {
                V s = madd(o.lo, p, V(0));
                s = madd(o.di, o.xi, s);
                s = madd(o.up, xn, s);
                p = o.xi + quotient<decltype(FAST)::value>(o.bi - s, o.di, o.r);    // openmg/solvers.py:68
                s_out[j] = p;
            };
            if (rows == 64) {
                // a row's operands are read LA rows ahead of their use (an LDS round trip is longer than a row's chain)
                constexpr int LA = 3;
                Ops q[LA + 1];
#pragma unroll
                for (int j = 0; j < LA; ++j) q[j] = ops_of(j);
#pragma unroll
                for (int j = 0; j < 64; ++j) {
                    q[(j + LA) % (LA + 1)] = ops_of(j + LA < 64 ? j + LA : 64);      // (row 64: only its x is there, and only that is used)
                    link(j, q[j % (LA + 1)], q[(j + 1) % (LA + 1)].xi);
                }
            } else {
                for (int j = 0; j < rows; ++j) link(j, ops_of(j), R[j + 1][4]);
            }
            return p;
        };
        V last = walk(std::true_type(), prev);
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
        __builtin_amdgcn_wave_barrier();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
        if (sizeof(V) == 8) {
            // every lane's own row once more: was its numerator one the hoisted reciprocal covers?
            const V left = lane ? s_out[(lane - 1) & 63] : prev;
            // (both shuffles with every lane active; lane 63's upper neighbour is row base + 64: the NEXT block's lane 0)
            const V x_down = __shfl_down(cur.x, 1, 64);
            const V x_next0 = __shfl(nxt.x, 0, 64);
            const V xn = lane + 1 < 64 ? x_down : x_next0;
            V s = madd(cur.lo, left, V(0));
            s = madd(cur.di, cur.x, s);
            s = madd(cur.up, (base + lane + 1 < n) ? xn : V(0), s);
            // (+0 — b_i equal to the row's sum, every row of a converged iterate — goes through the hoisted form like the
            // division: q = +0 or -0 with the diagonal's sign, either way)
            const V num = cur.b - s;
            const bool bad = lane < rows && !plain_range(num) && __double_as_longlong(double(num)) != 0;
            if (__builtin_amdgcn_ballot_w64(bad)) {
                last = walk(std::false_type(), prev);
                __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                __builtin_amdgcn_wave_barrier();
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
            }
        }
        prev = last;
        if (lane < rows) x[base + lane] = s_out[lane];
        cur = nxt;
        nxt = after;
    }
}

template <typename V>
__global__ void fill_kernel(V *p, size_t n, V v) {
    for (size_t q = blockIdx.x * size_t(blockDim.x) + threadIdx.x; q < n; q += size_t(gridDim.x) * blockDim.x) p[q] = v;
}


// ---- the line-scan sweep (opt-in: OMG_MARCH_SCAN=1) ----------------------------------------------
// Not the bits of the level schedule: within a grid line the sweep is the first-order recurrence
//   x_i = c_i + alpha_i x_{i-1},  alpha_i = -a_{i,i-1} / a_ii,  c_i = (b_i - the other five neighbours' terms) / a_ii
// (the row's own old value drops out), and a wave resolves a whole line at once by a scan over (alpha, c) pairs — every lane
// holds C consecutive rows and composes them in order, the 64 lane totals are composed by six DPP steps (row_shr 1 2 4 8,
// row_bcast 15 31), and each lane applies its predecessor's end value.  The additions associate differently from the
// sequential loop and the division is a multiplication by the reciprocal (differences of a few ulp per row, |alpha| < 1):
// an option beside march_gs_kernel, tested against it to a tolerance.
//   * i leaves the wavefront: a line (j, k) needs the relaxed lines (j - 1, k) and (j, k - 1): ny + nz - 1 line steps
//     instead of nx + ny + nz - 2 row steps, and every lane busy at every step.
//   * A workgroup owns SCAN_W consecutive planes, one COMPUTING wave per plane; wave w relaxes line j = t - w of its plane at
//     step t: the (j - 1, k) line is its own previous result (registers), the (j, k - 1) line what wave w - 1 left in LDS one
//     step earlier.  Operands that are old values (own line, line j + 1, plane k + 1) and b are loaded D steps ahead into
//     registers; the computing waves never store to global memory (one counter for loads and stores: march_gs_kernel's header).
//     A step is split at the point where the (j, k - 1) line is needed: the rows' sums without that neighbour and the whole
//     alpha half of the scan (products of coefficients) come before it.
//   * No barrier per step: every wave publishes the steps it has finished in LDS (s_prog) and waits for the counts it needs —
//     its predecessor's step t - 1, and that the slot it is about to overwrite (SCAN_SLOTS steps old) has been read by its
//     successor and stored.
//   * A STORING wave writes finished lines from LDS to x — the workgroup's last plane first and, write-through, into the face
//     slots the next workgroup waits for (unset = a marker NaN between sweeps, as march_gs_kernel's faces) — and sets the face
//     slots this workgroup has consumed back to unset.
//   * A GHOST wave polls the previous workgroup's face lines, SCAN_Q lines per round, round after round, and hands complete
//     lines to wave 0 through an LDS ring.  Workgroups take their position from a ticket, so the one a workgroup waits for has
//     always started; every wait is bounded, then the error flag (MarchPlan::timed_out) and no more waiting.
// Measured (profiles/r06_scan_*): 256^3 0.61 ms per sweep against 0.92 (march_gs_kernel), 128^3 0.21 against 0.38, 64^3 0.10
// against 0.18, 32^3 0.05 against 0.08.  A workgroup alone: 0.8 us per step at nx = 256 (259 steps: 0.21 ms); a hop to the
// next workgroup 4.5 us before its first line is out (0.65 us per wave of the chain + 1.8 us through HBM and the poll), and
// each workgroup runs a little slower than the one that feeds it (0.28 ms for the 4th, 0.33 for the 60th of 64).  The ghost
// wave polls two lines per round and naps when it is half a ring ahead: eight lines per round cost 12 % of the sweep (the
// wave shares a SIMD with a computing wave).
constexpr int SCAN_W = 4;          // planes = computing waves per workgroup
constexpr int SCAN_RING = 16;      // ghost lines in LDS
constexpr int SCAN_Q = 2;          // face lines polled per round
constexpr int SCAN_SLOTS = 5;      // steps a computing wave's lines stay in LDS
constexpr int SCAN_ROUND_LIMIT = 1 << 18;   // rounds of the ghost wave without a new line (~1 us each)
constexpr int SCAN_SPIN_LIMIT = 1 << 24;    // polls of an LDS count (~0.1 us each)

template <typename V>
struct ScanArgs {
    V *x;
    const V *b;
    const uint8_t *code;   // pattern of every row, natural order
    const V *coef;         // [pattern][8]
    V *face;               // [workgroup][line][64 C]
    uint32_t *sync;        // ticket, finished workgroups, error flag
    int nx, ny, nz, G, n_pat;
    long long *trace;      // OMG_SCAN_TRACE=1 (experiments): per workgroup, wall_clock64 ticks of [0] start, [1] ghost has line 0,
                           // [2] face line 0 stored, [3] ghost has the last line, [4] last face line stored, [5] end
};

template <int CTRL, int RM>
__device__ __forceinline__ double dpp_take(double old, double src) {
    const int lo = __builtin_amdgcn_update_dpp(__double2loint(old), __double2loint(src), CTRL, RM, 0xf, false);
    const int hi = __builtin_amdgcn_update_dpp(__double2hiint(old), __double2hiint(src), CTRL, RM, 0xf, false);
    return __hiloint2double(hi, lo);
}
template <int CTRL, int RM>
__device__ __forceinline__ float dpp_take(float old, float src) {
    return __int_as_float(__builtin_amdgcn_update_dpp(__float_as_int(old), __float_as_int(src), CTRL, RM, 0xf, false));
}
// the seven table entries the scan uses (the eighth is not read: a register nobody needs would be handed to the next read
// while this one is still in flight, and the compiler would wait between the rows' reads)
__device__ __forceinline__ void load_coefs7(const double *c, double (&o)[8]) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    const v2d *p = reinterpret_cast<const v2d *>(c);
    const v2d a = p[0], b = p[1], e = p[2];
    o[0] = a.x; o[1] = a.y; o[2] = b.x; o[3] = b.y; o[4] = e.x; o[5] = e.y; o[6] = c[6];
}
__device__ __forceinline__ void load_coefs7(const float *c, float (&o)[8]) {
    typedef float v4f __attribute__((ext_vector_type(4)));
    typedef float v2f __attribute__((ext_vector_type(2)));
    const v4f a = *reinterpret_cast<const v4f *>(c);
    const v2f b = *reinterpret_cast<const v2f *>(c + 4);
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = a.w; o[4] = b.x; o[5] = b.y; o[6] = c[6];
}

__device__ __forceinline__ int lds_peek(int *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP); }
__device__ __forceinline__ void lds_post(int *p, int v) {
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");      // (the lines written before the count that announces them)
    __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}

// a bounded wait for an LDS count (another wave's progress): false = gave up
__device__ __forceinline__ bool lds_await(int *p, int at_least) {
    int spins = 0;
    while (lds_peek(p) < at_least) {
        __builtin_amdgcn_s_sleep(1);
        if (++spins > SCAN_SPIN_LIMIT) return false;
    }
    asm volatile("" ::: "memory");
    return true;
}

template <typename V, int C, bool PAIR>
__global__ __launch_bounds__((SCAN_W + 2) * 64) void scan_gs_kernel(ScanArgs<V> a) {
    constexpr int W = SCAN_W, NXP = 64 * C, R = C >= 8 ? SCAN_RING / 2 : SCAN_RING, R2 = SCAN_SLOTS, Q = SCAN_Q, D = C >= 8 ? 2 : 3, NTH = (SCAN_W + 2) * 64;
    extern __shared__ __attribute__((aligned(16))) unsigned char scan_dyn[];
    V *const s_coef = reinterpret_cast<V *>(scan_dyn);         // [256][8]
    V *const s_line = s_coef + 256 * 8;                         // [R2][W][NXP]: the waves' relaxed lines, step t in slot t % R2
    V *const s_ghost = s_line + R2 * W * NXP;                   // [R][NXP]
    __shared__ int s_ctl[2];                                    // ghost lines ready, position
    __shared__ int s_prog[W + 1];                               // steps finished by computing wave w; [W]: steps stored
    const int tid = threadIdx.x, wave = tid >> 6, lane = tid & 63;
    if (tid == 0) {
        s_ctl[1] = (int)atomicAdd(&a.sync[0], 1u);
        s_ctl[0] = 0;
    }
    if (tid <= W) s_prog[tid] = 0;
    // the table as the scan uses it: every coefficient divided by its row's diagonal, the reciprocal itself in the diagonal's
    // place (x_i = b_i / d - sum over the six neighbours of (c / d) x: the row's own old value drops out); pattern n_pat: zeros,
    // for the lanes' rows beyond the end of a line
    for (int q = tid; q < (a.n_pat + 1) * 8; q += NTH) {
        V v = V(0);
        if (q < a.n_pat * 8 && (q & 7) != 7) {
            const V r = V(1) / a.coef[(q & ~7) + 3];
            v = (q & 7) == 3 ? r : a.coef[q] * r;
        }
        s_coef[q] = v;
    }
    __syncthreads();                                            // (the only barrier: from here on the waves wait for counts)
    const int g = s_ctl[1];
    const int nx = a.nx, ny = a.ny, nz = a.nz;
    const int T = (ny + W - 1 + D - 1) / D * D;                 // steps: a whole number of unrolled groups
    if (a.trace && tid == NTH - 1) a.trace[8 * g + 0] = wall_clock64();

    if (wave < W) {
        // ---- computing wave: plane k, line t - wave at step t
        const int k = g * W + wave;
        const bool plane = k < nz;
        const int kc = min(k, nz - 1), ku = min(k + 1, nz - 1);
        int ic[C];
        bool cell[C];
#pragma unroll
        for (int s = 0; s < C; ++s) {
            const int i = lane * C + s;
            cell[s] = i < nx;
            ic[s] = min(i, nx - 1);
        }
        V sxo[D][C], sbb[D][C], sxu[D][C];
        uint32_t scd[D][C];
        V xprev[C];
#pragma unroll
        for (int s = 0; s < C; ++s) xprev[s] = V(0);
        bool gave_up = false;              // (reported after the loop: a store inside it would cost every load its counted wait)
        // PAIR (even nx, C >= 2): a lane's rows two at a time — half the load instructions, each with twice the bytes per
        // cache line it touches (a lane's C rows are consecutive: 16-byte pieces 8 C bytes apart)
        typedef V vpair __attribute__((ext_vector_type(2)));
        int ip[C / 2 > 0 ? C / 2 : 1];
#pragma unroll
        for (int q = 0; q < C / 2; ++q) ip[q] = min(lane * C + 2 * q, nx - 2);
        auto fetch = [&](int u, int t) {
            const int jc = min(max(t - wave, 0), ny - 1);
            const int base = (kc * ny + jc) * nx, baseu = (ku * ny + jc) * nx;
            // (the patterns first: they address the table in LDS before anything else of the step can start)
            if constexpr (PAIR) {
#pragma unroll
                for (int q = 0; q < C / 2; ++q) {
                    const uint32_t two = *reinterpret_cast<const uint16_t *>(a.code + base + ip[q]);
                    scd[u][2 * q] = two & 255u;
                    scd[u][2 * q + 1] = two >> 8;
                }
#pragma unroll
                for (int q = 0; q < C / 2; ++q) {
                    const vpair xo = *reinterpret_cast<const vpair *>(a.x + base + ip[q]);
                    const vpair bb = *reinterpret_cast<const vpair *>(a.b + base + ip[q]);
                    const vpair xu = *reinterpret_cast<const vpair *>(a.x + baseu + ip[q]);
                    sxo[u][2 * q] = xo.x; sxo[u][2 * q + 1] = xo.y;
                    sbb[u][2 * q] = bb.x; sbb[u][2 * q + 1] = bb.y;
                    sxu[u][2 * q] = xu.x; sxu[u][2 * q + 1] = xu.y;
                }
            } else {
#pragma unroll
                for (int s = 0; s < C; ++s) scd[u][s] = a.code[base + ic[s]];
#pragma unroll
                for (int s = 0; s < C; ++s) {
                    sxo[u][s] = a.x[base + ic[s]];
                    sbb[u][s] = a.b[base + ic[s]];
                    sxu[u][s] = a.x[baseu + ic[s]];
                }
            }
        };
#pragma unroll
        for (int u = 0; u < D; ++u) fetch(u, u);
        for (int t0 = 0; t0 < T; t0 += D) {
#pragma unroll
            for (int u = 0; u < D; ++u) {
                const int t = t0 + u, j = t - wave;
                const bool active = plane && j >= 0 && j < ny;
                if (active) {
                    // ---- before the line (j, k - 1) is needed: everything that does not depend on it — the rows' sums without
                    // that neighbour, and the whole alpha half of the scan (products of coefficients: no x in them).  What a
                    // plane's line costs the next plane is then only the part below the wait.
                    const V xe_next = dpp_take<DPP_WAVE_SHL1, 0xf>(V(0), sxo[u][0]);     // lane + 1's first row (lane 63: 0)
                    const int un = (u + 1) % D;                                          // the stage of line j + 1
                    V Ap[C], ccp[C], aa[C], ck[C];
                    // four rows at a time: all their table reads first (one exposed LDS latency per group, not per row)
                    constexpr int CG = C < 4 ? C : 4;
#pragma unroll
                    for (int s0 = 0; s0 < C; s0 += CG) {
                        V co[CG][8];
#pragma unroll
                        for (int q = 0; q < CG; ++q)
                            load_coefs7(s_coef + (cell[s0 + q] ? scd[u][s0 + q] : (uint32_t)a.n_pat) * 8, co[q]);
#pragma unroll
                        for (int q = 0; q < CG; ++q) {
                            const int s = s0 + q;
                            const V xe = s + 1 < C ? sxo[u][s + 1 < C ? s + 1 : s] : xe_next;
                            V sum = co[q][4] * xe;
                            sum = madd(co[q][5], sxo[un][s], sum);
                            sum = madd(co[q][6], sxu[u][s], sum);
                            sum = madd(co[q][1], xprev[s], sum);
                            ccp[s] = madd(sbb[u][s], co[q][3], -sum);
                            ck[s] = -co[q][0];
                            aa[s] = -co[q][2];
                            Ap[s] = s == 0 ? aa[0] : aa[s] * Ap[s > 0 ? s - 1 : 0];
                        }
                    }
                    // the lanes' products as the six steps of the scan will want them
                    V As[6];
                    {
                        V A = Ap[C - 1];
#define OMG_SCAN_A(I, CTRL, RM)                                  \
    {                                                            \
        As[I] = A;                                               \
        A = A * dpp_take<CTRL, RM>(V(1), A);                     \
    }
                        OMG_SCAN_A(0, 0x111, 0xf)
                        OMG_SCAN_A(1, 0x112, 0xf)
                        OMG_SCAN_A(2, 0x114, 0xf)
                        OMG_SCAN_A(3, 0x118, 0xf)
                        OMG_SCAN_A(4, 0x142, 0xa)
                        OMG_SCAN_A(5, 0x143, 0xc)
#undef OMG_SCAN_A
                    }
                    // ---- the line (j, k - 1): wave - 1 has finished step t - 1 / the ghost wave has line j; the slot this step
                    // writes held step t - R2, read by wave + 1 at its step t - R2 + 1 and by the storing wave.  All three counts
                    // are read at once (one LDS latency when nothing is missing, the usual case); after one wait has run out the
                    // sweep is lost — MarchPlan::timed_out — and nothing waits any more
                    if (!gave_up) {
                        int *const p_prev = wave > 0 ? &s_prog[wave - 1] : &s_ctl[0];
                        const int n_prev = wave > 0 ? t : (g > 0 ? j + 1 : 0);
                        int *const p_next = &s_prog[wave + 1 < W ? wave + 1 : W];
                        const int n_next = wave + 1 < W ? t - R2 + 2 : 0, n_store = t - R2 + 1;
                        const int c_prev = lds_peek(p_prev), c_next = lds_peek(p_next), c_store = lds_peek(&s_prog[W]);
                        if (c_prev < n_prev || c_next < n_next || c_store < n_store)
                            gave_up = !(lds_await(p_prev, n_prev) && lds_await(p_next, n_next) && lds_await(&s_prog[W], n_store));
                        asm volatile("" ::: "memory");
                    }
                    V Cp[C];
                    if (wave == 0 && g == 0) {
#pragma unroll
                        for (int s = 0; s < C; ++s) Cp[s] = ccp[s];
                    } else {
                        const V *src = wave == 0 ? s_ghost + (j % R) * NXP + lane * C
                                                 : s_line + (((t - 1) % R2) * W + (wave - 1)) * NXP + lane * C;
#pragma unroll
                        for (int s = 0; s < C; ++s) Cp[s] = madd(ck[s], src[s], ccp[s]);
                    }
#pragma unroll
                    for (int s = 1; s < C; ++s) Cp[s] = madd(aa[s], Cp[s - 1], Cp[s]);
                    V Cs = Cp[C - 1];
#define OMG_SCAN_C(I, CTRL, RM) Cs = madd(As[I], dpp_take<CTRL, RM>(V(0), Cs), Cs);
                    OMG_SCAN_C(0, 0x111, 0xf)
                    OMG_SCAN_C(1, 0x112, 0xf)
                    OMG_SCAN_C(2, 0x114, 0xf)
                    OMG_SCAN_C(3, 0x118, 0xf)
                    OMG_SCAN_C(4, 0x142, 0xa)
                    OMG_SCAN_C(5, 0x143, 0xc)
#undef OMG_SCAN_C
                    const V xin = dpp_take<DPP_WAVE_SHR1, 0xf>(V(0), Cs);                // the end value of lane - 1 (lane 0: 0)
                    V *dst = s_line + ((t % R2) * W + wave) * NXP + lane * C;
#pragma unroll
                    for (int s = 0; s < C; ++s) {
                        xprev[s] = madd(Ap[s], xin, Cp[s]);
                        dst[s] = xprev[s];
                    }
                }
                if (lane == 0) lds_post(&s_prog[wave], t + 1);          // (first: the next plane's wave waits for this)
                fetch(u, t + D);
            }
        }
        if (gave_up) store_through(&a.sync[2], 1u);
    } else if (wave == W) {
        // ---- storing wave: every wave's line of step t as soon as that wave has finished the step
        bool gave_up = false;
        // the face lines the ghost wave has taken over: back to "unset" for the next sweep (here, not there: a wave that polls
        // with loads would wait for its own write-through stores at every round)
        int reset_next = 0;
        V *const fin = a.face + size_t(g > 0 ? g - 1 : 0) * ny * NXP;
        auto reset = [&](int upto) {
            for (; reset_next < upto; ++reset_next)
#pragma unroll
                for (int c = 0; c < C; ++c) store_through(fin + size_t(reset_next) * NXP + c * 64 + lane, Unset<V>::value());
        };
        auto put = [&](int w, int t) {
            const int jl = t - w, k = g * W + w;
            if (k >= nz || jl < 0 || jl >= ny) return;
            const V *src = s_line + ((t % R2) * W + w) * NXP;
            V *xo = a.x + (size_t(k) * ny + jl) * nx;
            const bool face = w == W - 1 && g + 1 < a.G;
            V *fo = a.face + (size_t(g) * ny + jl) * NXP;
            if constexpr (PAIR) {
                typedef V vpair __attribute__((ext_vector_type(2)));
#pragma unroll
                for (int c = 0; c < C / 2; ++c) {
                    const int idx = (c * 64 + lane) * 2;
                    const vpair v = *reinterpret_cast<const vpair *>(src + idx);
                    if (face) {
                        store_through(fo + idx, idx < nx ? v.x : V(0));
                        store_through(fo + idx + 1, idx < nx ? v.y : V(0));
                    }
                    if (idx < nx) *reinterpret_cast<vpair *>(xo + idx) = v;
                }
            } else {
#pragma unroll
                for (int c = 0; c < C; ++c) {
                    const int idx = c * 64 + lane;
                    const V v = src[idx];
                    if (face) store_through(fo + idx, idx < nx ? v : V(0));
                    if (idx < nx) xo[idx] = v;
                }
            }
        };
        // Iteration t: the last plane's line of step t (the next workgroup waits for it) as soon as wave W - 1 has finished that
        // step — whatever the other waves are doing —, then the other planes' lines of step t - 1.  s_prog[W] = steps of which
        // every line is stored.
        const int TS = ny + W - 1;
        for (int t = 0; t <= TS; ++t) {
            if (t < TS) {
                if (!gave_up) gave_up = !lds_await(&s_prog[W - 1], t + 1);
                if (a.trace && lane == 0 && t == W - 1) a.trace[8 * g + 2] = wall_clock64();
                put(W - 1, t);
                if (a.trace && lane == 0 && t == TS - 1) a.trace[8 * g + 4] = wall_clock64();
            }
            if (t > 0) {
                if (!gave_up) {
                    int cnt[W - 1];
#pragma unroll
                    for (int w = 0; w < W - 1; ++w) cnt[w] = lds_peek(&s_prog[w]);
                    bool all = true;
#pragma unroll
                    for (int w = 0; w < W - 1; ++w) all = all && cnt[w] >= t;
                    if (!all)
#pragma unroll
                        for (int w = 0; w < W - 1; ++w) gave_up = gave_up || !lds_await(&s_prog[w], t);
                    asm volatile("" ::: "memory");
                }
#pragma unroll
                for (int w = W - 2; w >= 0; --w) put(w, t - 1);
                if (lane == 0) lds_post(&s_prog[W], t);
                if (g > 0) reset(min(lds_peek(&s_ctl[0]), reset_next + 1 + (t & 1)));
            }
        }
        if (lane == 0) lds_post(&s_prog[W], T + R2);            // (steps without lines: nobody waits for them)
        if (g > 0) {
            if (!gave_up) gave_up = !lds_await(&s_ctl[0], ny);
            reset(ny);
        }
        if (a.trace && lane == 0) a.trace[8 * g + 5] = wall_clock64();
        if (gave_up) store_through(&a.sync[2], 1u);
    } else if (g > 0) {
        // ---- ghost wave: the previous workgroup's face lines into the LDS ring, ahead of wave 0, round after round
        int jn = 0, base_line = 0, idle = 0;
        V v[Q][C];
        V *const fin = a.face + size_t(g - 1) * ny * NXP;
        while (jn < ny) {
            base_line = jn;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int line = min(jn + q, ny - 1);
#pragma unroll
                for (int c = 0; c < C; ++c) v[q][c] = load_through(fin + size_t(line) * NXP + c * 64 + lane);
            }
            // wave 0 has finished its steps < s_prog[0], i.e. is done with the lines < s_prog[0]: room for R lines from there
            const int limit = min(ny, lds_peek(&s_prog[0]) + R);
            const int before = jn;
#pragma unroll
            for (int q = 0; q < Q; ++q) {
                const int line = base_line + q;
                bool set = true;
#pragma unroll
                for (int c = 0; c < C; ++c) set = set && !Unset<V>::is(v[q][c]);
                const bool all = __ballot(set) == ~0ull;
                if (line == jn && line < limit && all) {
                    V *dst = s_ghost + (line % R) * NXP;
#pragma unroll
                    for (int c = 0; c < C; ++c) dst[c * 64 + lane] = v[q][c];      // (the storing wave resets the slots)
                    jn = line + 1;
                }
            }
            if (jn != before) {
                if (lane == 0) lds_post(&s_ctl[0], jn);
                if (jn - (limit - R) > R / 2) __builtin_amdgcn_s_sleep(16);    // (well ahead of wave 0: leave the SIMD to the wave it shares it with)
                if (a.trace && lane == 0 && before == 0) a.trace[8 * g + 1] = wall_clock64();
                if (a.trace && lane == 0 && jn == ny) a.trace[8 * g + 3] = wall_clock64();
                idle = 0;
            } else {
                if (jn >= limit) __builtin_amdgcn_s_sleep(8);   // (the ring is full: wave 0 is behind, nothing to poll for)
                if (++idle > SCAN_ROUND_LIMIT) {
                    // give up: the error flag, and whatever the slots hold, so that nobody waits for ever
                    store_through(&a.sync[2], 1u);
                    jn = ny;
                    if (lane == 0) lds_post(&s_ctl[0], jn);
                }
            }
        }
    }
    // the last workgroup to finish rewinds the ticket for the next sweep
    if (tid == 0) {
        __threadfence();
        const uint32_t done = atomicAdd(&a.sync[1], 1u);
        if (done + 1u == (uint32_t)a.G) {
            store_through(&a.sync[1], 0u);
            store_through(&a.sync[0], 0u);
        }
    }
}

}  // namespace

template <typename V>
bool MarchPlan<V>::build(const omg_csr &A, hipStream_t s) {
    const int64_t n = A.n_rows;
    if (n < 2 || n != A.n_cols || uint64_t(n) * sizeof(V) >= (uint64_t(1) << 32)) return false;   // 32-bit byte offsets
    auto has = [&](int64_t r, int64_t c) {
        for (int64_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p)
            if (A.indices[p] == c) return true;
        return false;
    };
    // line length: the first row without a coupling to its predecessor starts the second line
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
    if (ny == 1 && nz > 1) return false;
    const int64_t sj = nx, sk = nx * ny;
    {
        // a 1-D grid: any tridiagonal operator with ascending columns and diagonals the hoisted reciprocal covers
        const char *e = getenv("OMG_MARCH_LINE");
        if (ny == 1 && nz == 1 && !(e && e[0] == '0')) {
            std::vector<V> t(size_t(3) * size_t(n), V(0));
            for (int64_t r = 0; r < n; ++r) {
                int last = -1;
                for (int64_t q = A.indptr[r]; q < A.indptr[r + 1]; ++q) {
                    const int64_t off = int64_t(A.indices[q]) - r;
                    const int slot = off == -1 ? 0 : off == 0 ? 1 : off == 1 ? 2 : -1;
                    if (slot <= last) return false;                       // not a neighbour, or not in column order
                    last = slot;
                    t[size_t(slot) * size_t(n) + size_t(r)] = V(A.data[q]);
                }
                const double d = double(t[size_t(n) + size_t(r)]);
                if (!(std::fabs(d) >= 0x1p-400 && std::fabs(d) <= 0x1p400)) return false;
            }
            g.nx = int(nx); g.ny = 1; g.nz = 1;
            g.TJ = 64; g.TK = 1; g.ntj = g.ntk = g.n_tiles = 1; g.T = g.nx; g.n_grp = 0; g.n_pat = 0;
            tri.alloc(t.size());
            tri.upload(t.data(), t.size(), s);
            sync.alloc(SYNC_WORDS);
            sync.zero(s);
            OMG_HIP(hipStreamSynchronize(s));
            line1 = true;
            return true;
        }
    }

    // every row as seven coefficients in slot order; stored order must be slot order
    typedef std::array<double, 7> Pat;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int nt = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 65536));
    std::vector<uint8_t> code((size_t)n);
    std::vector<std::vector<Pat>> local(nt);
    std::atomic<bool> ok(true);
    std::atomic<bool> many(false);            // more than 256 distinct rows: the per-row form (below) instead of the pattern table
    auto scan = [&](int t) {
        const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
        std::vector<Pat> &pats = local[t];
        size_t hit = 0;
        for (int64_t r = lo; r < hi && ok.load(std::memory_order_relaxed); ++r) {
            const int64_t i = r % nx, jl = (r / nx) % ny, kl = r / sk;
            Pat p;
            p.fill(0.0);
            int last = -1;
            for (int64_t q = A.indptr[r]; q < A.indptr[r + 1]; ++q) {
                const int64_t off = int64_t(A.indices[q]) - r;
                int slot = -1;
                if (off == 0) slot = 3;
                else if (off == -1 && i > 0) slot = 2;
                else if (off == 1 && i + 1 < nx) slot = 4;
                else if (off == -sj && jl > 0) slot = 1;
                else if (off == sj && jl + 1 < ny) slot = 5;
                else if (off == -sk && kl > 0) slot = 0;
                else if (off == sk && kl + 1 < nz) slot = 6;
                if (slot <= last) { ok = false; return; }      // not a neighbour, or not in column order
                last = slot;
                p[slot] = A.data[q];
            }
            // (the diagonal's exponent: the range in which march_gs_kernel's division needs no scaling)
            if (!(std::fabs(p[3]) >= 0x1p-400 && std::fabs(p[3]) <= 0x1p400)) { ok = false; return; }
            if (hit < pats.size() && !memcmp(&pats[hit], &p, sizeof(Pat))) { code[r] = (uint8_t)hit; continue; }
            size_t f = 0;
            while (f < pats.size() && memcmp(&pats[f], &p, sizeof(Pat))) ++f;
            if (f == pats.size()) {
                if (pats.size() >= 256) { many = true; ok = false; return; }
                pats.push_back(p);
            }
            hit = f;
            code[r] = (uint8_t)f;
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(scan, t);
        scan(0);
        for (auto &q : th) q.join();
    }
    std::vector<Pat> pats;
    std::vector<std::vector<uint8_t>> remap(nt);
    if (ok) {
        for (int t = 0; t < nt && !many; ++t)
            for (const Pat &p : local[t]) {
                size_t f = 0;
                while (f < pats.size() && memcmp(&pats[f], &p, sizeof(Pat))) ++f;
                if (f == pats.size()) {
                    if (pats.size() >= 256) { many = true; break; }
                    pats.push_back(p);
                }
                remap[t].push_back((uint8_t)f);
            }
    }
    if (!ok && !many) return false;
    const bool rows_mode = many.load();
    if (rows_mode) {
        // (OMG_MARCH_ROWS=0: operators with more than 256 distinct rows keep the level schedule, as before round 6)
        const char *e = getenv("OMG_MARCH_ROWS");
        if (e && e[0] == '0') return false;
        // the whole operator once more: the same structural test, no table
        std::atomic<bool> fine(true);
        auto check = [&](int t) {
            const int64_t lo = n * t / nt, hi = n * (t + 1) / nt;
            for (int64_t r = lo; r < hi && fine.load(std::memory_order_relaxed); ++r) {
                const int64_t i = r % nx, jl = (r / nx) % ny, kl = r / sk;
                int last = -1;
                double diag = 0.0;
                for (int64_t q = A.indptr[r]; q < A.indptr[r + 1]; ++q) {
                    const int64_t off = int64_t(A.indices[q]) - r;
                    int slot = -1;
                    if (off == 0) { slot = 3; diag = A.data[q]; }
                    else if (off == -1 && i > 0) slot = 2;
                    else if (off == 1 && i + 1 < nx) slot = 4;
                    else if (off == -sj && jl > 0) slot = 1;
                    else if (off == sj && jl + 1 < ny) slot = 5;
                    else if (off == -sk && kl > 0) slot = 0;
                    else if (off == sk && kl + 1 < nz) slot = 6;
                    if (slot <= last) { fine = false; return; }
                    last = slot;
                }
                if (!(std::fabs(diag) >= 0x1p-400 && std::fabs(diag) <= 0x1p400)) { fine = false; return; }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(check, t);
        check(0);
        for (auto &q : th) q.join();
        if (!fine) return false;
    }

    g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz;
    g.TJ = nz > 1 ? 8 : 64;
    g.TK = 64 / g.TJ;
    g.ntj = (g.ny + g.TJ - 1) / g.TJ;
    g.ntk = (g.nz + g.TK - 1) / g.TK;
    g.n_tiles = g.ntj * g.ntk;
    g.T = g.nx + g.TJ + g.TK - 2;
    g.n_grp = 2 * ((g.T + 7) / 8);
    g.n_pat = (int)pats.size();

    per_row = rows_mode;
    if (rows_mode) {
        // every row's seven coefficients as the tiles consume them: [tile][step][8][lane], steps padded to whole blocks of 8
        // (entry 7 — the refined reciprocal of the diagonal — is formed by the loading wave); a step outside the lane's line:
        // zeros and a diagonal of one
        const int Tpad = (g.T + 7) / 8 * 8;
        if (uint64_t(g.n_tiles) * uint64_t(Tpad) * 512u * sizeof(V) > (uint64_t(24) << 30)) return false;
        std::vector<V> rc(size_t(g.n_tiles) * size_t(Tpad) * 512, V(0));
        auto fill = [&](int t) {
            for (int tile = t; tile < g.n_tiles; tile += nt) {
                const int J = tile % g.ntj, K = tile / g.ntj;
                for (int lane = 0; lane < 64; ++lane) {
                    const int jj = lane % g.TJ, kk = lane / g.TJ;
                    const int j = J * g.TJ + jj, k = K * g.TK + kk;
                    const bool valid = j < g.ny && k < g.nz;
                    const int64_t line = valid ? (int64_t(k) * ny + j) * nx : 0;
                    for (int st = 0; st < Tpad; ++st) {
                        V *const o = rc.data() + (size_t(tile) * size_t(Tpad) + size_t(st)) * 512 + size_t(lane);
                        const int i = st - jj - kk;
                        o[3 * 64] = V(1);
                        if (!valid || i < 0 || i >= nx) continue;
                        const int64_t r = line + i;
                        const int64_t jl = (r / nx) % ny, kl = r / sk;
                        for (int64_t q = A.indptr[r]; q < A.indptr[r + 1]; ++q) {
                            const int64_t off = int64_t(A.indices[q]) - r;
                            int slot = 3;
                            if (off == -1) slot = 2;
                            else if (off == 1) slot = 4;
                            else if (off == -sj && jl > 0) slot = 1;
                            else if (off == sj && jl + 1 < ny) slot = 5;
                            else if (off == -sk && kl > 0) slot = 0;
                            else if (off == sk && kl + 1 < nz) slot = 6;
                            o[size_t(slot) * 64] = V(A.data[q]);
                        }
                    }
                }
            }
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(fill, t);
        fill(0);
        for (auto &q : th) q.join();
        rowc.alloc(rc.size());
        rowc.upload(rc.data(), rc.size(), s);
        OMG_HIP(hipStreamSynchronize(s));
    }
    // the codes as the tiles consume them: one 4-byte word per (tile, group of four steps, lane)
    std::vector<uint32_t> words(rows_mode ? size_t(64) : size_t(g.n_tiles) * g.n_grp * 64);
    auto arrange = [&](int t) {
        if (rows_mode) return;
        for (int tile = t; tile < g.n_tiles; tile += nt) {
            const int J = tile % g.ntj, K = tile / g.ntj;
            for (int lane = 0; lane < 64; ++lane) {
                const int jj = lane % g.TJ, kk = lane / g.TJ;
                const int j = J * g.TJ + jj, k = K * g.TK + kk;
                const bool valid = j < g.ny && k < g.nz;
                const int64_t line = valid ? (int64_t(k) * ny + j) * nx : 0;
                int owner = 0;                       // the scanning thread of the line's rows (for the remap)
                for (int grp = 0; grp < g.n_grp; ++grp) {
                    uint32_t w = 0;
                    for (int u = 0; u < 4; ++u) {
                        const int i = grp * 4 + u - jj - kk;
                        if (!valid || i < 0 || i >= nx) continue;
                        const int64_t r = line + i;
                        while (r >= n * (owner + 1) / nt) ++owner;
                        while (r < n * owner / nt) --owner;
                        w |= uint32_t(remap[owner][code[r]]) << (8 * u);
                    }
                    words[(size_t(tile) * g.n_grp + grp) * 64 + lane] = w;
                }
            }
        }
    };
    {
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(arrange, t);
        arrange(0);
        for (auto &q : th) q.join();
    }
    std::vector<V> cf(size_t(g.n_pat) * 8, V(0));
    for (int q = 0; q < g.n_pat; ++q)
        for (int e = 0; e < 7; ++e) cf[size_t(q) * 8 + e] = V(pats[q][e]);
    {
        std::vector<int32_t> ord;
        ord.reserve(size_t(g.n_tiles));
        for (int d = 0; d <= g.ntj + g.ntk - 2; ++d)
            for (int K = std::max(0, d - (g.ntj - 1)); K <= std::min(d, g.ntk - 1); ++K) ord.push_back(K * g.ntj + (d - K));
        const char *e = experiment_env("OMG_MARCH_ORDER");
        if (!(e && e[0] == '0')) {
            order.alloc(ord.size());
            order.upload(ord.data(), ord.size(), s);
            OMG_HIP(hipStreamSynchronize(s));
        }
    }
    codes.alloc(words.size());
    coef.alloc(cf.size());
    sync.alloc(SYNC_WORDS);
    // one slot per row of every tile's +J / +K face lines, all unset between sweeps
    faceJ.alloc(size_t(g.n_tiles) * g.TK * (g.nx + 2 * FACE_PAD));
    faceK.alloc(size_t(g.n_tiles) * g.TJ * (g.nx + 2 * FACE_PAD));
    codes.upload(words.data(), words.size(), s);
    coef.upload(cf.data(), cf.size(), s);
    sync.zero(s);
    hipLaunchKernelGGL(fill_kernel<V>, dim3(1024), dim3(256), 0, s, faceJ.p, faceJ.n, Unset<V>::value());
    hipLaunchKernelGGL(fill_kernel<V>, dim3(1024), dim3(256), 0, s, faceK.p, faceK.n, Unset<V>::value());
    OMG_HIP(hipGetLastError());
    OMG_HIP(hipStreamSynchronize(s));
    // OMG_MARCH_SCAN=1 (read when the plan is made): the line-scan sweep for 3-D pattern-table levels with lines of at most
    // 512 rows — rounding-level differences from the sequential loop, see scan_gs_kernel
    line_scan = false;
    {
        const char *e = getenv("OMG_MARCH_SCAN");
        if (e && e[0] == '1' && !rows_mode && nz >= 2 && nx <= 512 && g.n_pat < 256) {
            std::vector<uint8_t> rc((size_t)n);
            for (int t = 0; t < nt; ++t)
                for (int64_t r = n * t / nt; r < n * (t + 1) / nt; ++r) rc[(size_t)r] = remap[t][code[r]];
            rowcode.alloc(rc.size());
            rowcode.upload(rc.data(), rc.size(), s);
            scan_c = nx <= 64 ? 1 : nx <= 128 ? 2 : nx <= 256 ? 4 : 8;
            scan_g = (g.nz + SCAN_W - 1) / SCAN_W;
            face_scan.alloc(std::max<size_t>(1, size_t(scan_g - 1) * size_t(g.ny) * size_t(64 * scan_c)));
            hipLaunchKernelGGL(fill_kernel<V>, dim3(256), dim3(256), 0, s, face_scan.p, face_scan.n, Unset<V>::value());
            OMG_HIP(hipGetLastError());
            OMG_HIP(hipStreamSynchronize(s));
            line_scan = true;
        }
    }
    return true;
}

template <typename V, int C, bool PAIR>
static void launch_scan_as(const ScanArgs<V> &a, hipStream_t s) {
    const size_t lds = (size_t(256) * 8 + size_t(SCAN_SLOTS) * SCAN_W * 64 * C + size_t(C >= 8 ? SCAN_RING / 2 : SCAN_RING) * 64 * C) * sizeof(V) + 16;
    static bool allowed = false;                              // (per instantiation)
    if (!allowed) {
        OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(scan_gs_kernel<V, C, PAIR>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
        allowed = true;
    }
    hipLaunchKernelGGL((scan_gs_kernel<V, C, PAIR>), dim3((unsigned)a.G), dim3((SCAN_W + 2) * 64), lds, s, a);
    OMG_HIP(hipGetLastError());
}
template <typename V, int C>
static void launch_scan(const ScanArgs<V> &a, hipStream_t s) {
    if (C >= 2 && a.nx % 2 == 0) launch_scan_as<V, C, (C >= 2)>(a, s);
    else launch_scan_as<V, C, false>(a, s);
}

template <typename V>
void MarchPlan<V>::sweep(V *x, const V *b, hipStream_t s) const {
    if (line1) {
        hipLaunchKernelGGL(line_gs_kernel<V>, dim3(1), dim3(64), 0, s, x, b, tri.p, g.nx);
        OMG_HIP(hipGetLastError());
        return;
    }
    if (line_scan) {
        ScanArgs<V> sa;
        sa.x = x; sa.b = b; sa.code = rowcode.p; sa.coef = coef.p; sa.face = face_scan.p; sa.sync = sync.p;
        sa.nx = g.nx; sa.ny = g.ny; sa.nz = g.nz; sa.G = scan_g; sa.n_pat = g.n_pat;
        static const bool trace = [] { const char *e = experiment_env("OMG_SCAN_TRACE"); return e && e[0] == '1'; }();
        DevBuf<long long> tr;
        sa.trace = nullptr;
        if (trace) { tr.alloc(size_t(8) * scan_g); tr.zero(s); sa.trace = tr.p; }
        if (scan_c == 1) launch_scan<V, 1>(sa, s);
        else if (scan_c == 2) launch_scan<V, 2>(sa, s);
        else if (scan_c == 4) launch_scan<V, 4>(sa, s);
        else launch_scan<V, 8>(sa, s);
        if (trace) {
            std::vector<long long> h(size_t(8) * scan_g);
            tr.download(h.data(), h.size(), s);
            OMG_HIP(hipStreamSynchronize(s));
            long long t0 = h[0];
            for (int q = 0; q < scan_g; ++q) t0 = std::min(t0, h[8 * q]);
            fprintf(stderr, "[omg scan] %d x %d x %d, %d workgroups (us: start, ghost has line 0, face line 0 stored, ghost has last line, last face line stored, end)\n", g.nx, g.ny, g.nz, scan_g);
            for (int q = 0; q < scan_g; q += (scan_g > 16 ? scan_g / 16 : 1))
                fprintf(stderr, "[omg scan]   wg %3d: %8.2f %8.2f %8.2f %8.2f %8.2f %8.2f\n", q, (h[8 * q] - t0) / 100.0, (h[8 * q + 1] - t0) / 100.0,
                        (h[8 * q + 2] - t0) / 100.0, (h[8 * q + 3] - t0) / 100.0, (h[8 * q + 4] - t0) / 100.0, (h[8 * q + 5] - t0) / 100.0);
        }
        return;
    }
    MarchArgs<V> a;
    a.x = x; a.b = b; a.codes = codes.p; a.coef = coef.p; a.rowc = rowc.p; a.sync = sync.p; a.order = order.p;
    a.nx = g.nx; a.ny = g.ny; a.nz = g.nz; a.TJ = g.TJ; a.ntj = g.ntj; a.n_tiles = g.n_tiles;
    a.T = g.T; a.n_grp = g.n_grp; a.n_pat = g.n_pat; a.n = g.nx * g.ny * g.nz;
    a.faceJ = faceJ.p; a.faceK = faceK.p;
    static const int steps = [] { const char *e = experiment_env("OMG_MARCH_STEPS"); return e ? atoi(e) : 8; }();   // steps per block
    static const bool debug = [] { const char *e = getenv("OMG_MARCH_DEBUG"); return e && e[0] == '1'; }();
    DevBuf<long long> dbg;
    a.dbg = nullptr;
    if (debug) { dbg.alloc(size_t(8) * g.n_tiles); dbg.zero(s); a.dbg = dbg.p; }
    // workers: one per CU (a pad of dynamic LDS keeps a second one off the CU), each taking tiles until none is left
    static const int workers = [] { const char *e = experiment_env("OMG_MARCH_WORKERS"); return e ? atoi(e) : 256; }();
    const bool persistent = workers > 0 && g.n_tiles > workers;
    const dim3 grid((unsigned)(persistent ? workers : g.n_tiles));
    const size_t pad = persistent ? size_t(56) * 1024 : 0;
    if (per_row) {
        // the loading wave's ring IS the dynamic LDS (98 KB in double: nothing else fits beside it on the compute unit)
        const size_t ring = size_t(CRING) * 8 * 512 * sizeof(V);
        static bool allowed_rows = false;
        if (!allowed_rows) {
            OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(march_gs_kernel<V, 8, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(ring)));
            allowed_rows = true;
        }
        hipLaunchKernelGGL((march_gs_kernel<V, 8, true>), grid, dim3(192), ring, s, a);
        OMG_HIP(hipGetLastError());
    } else {
    if (pad) {
        static bool allowed = false;                          // (per instantiation of sweep<V>)
        if (!allowed) {
            OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(march_gs_kernel<V, 4, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(pad)));
            OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(march_gs_kernel<V, 8, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(pad)));
            allowed = true;
        }
    }
    if (steps == 4) hipLaunchKernelGGL((march_gs_kernel<V, 4, false>), grid, dim3(128), pad, s, a);
    else hipLaunchKernelGGL((march_gs_kernel<V, 8, false>), grid, dim3(128), pad, s, a);
    OMG_HIP(hipGetLastError());
    }
    if (debug) {
        std::vector<long long> h(size_t(8) * g.n_tiles);
        dbg.download(h.data(), h.size(), s);
        OMG_HIP(hipStreamSynchronize(s));
        long long t0 = h[0];
        for (int q = 0; q < g.n_tiles; ++q) t0 = std::min(t0, h[8 * q]);
        fprintf(stderr, "[omg march] %d x %d x %d, %d tiles (100 MHz ticks: start, end, in face waits, polls, ring full | storing wave: end, idle)\n",
                g.nx, g.ny, g.nz, g.n_tiles);
        if (g.n_tiles <= 16) {
            for (int q = 0; q < g.n_tiles; ++q)
                fprintf(stderr, "[omg march]   tile %4d (J %2d K %2d): %8lld %8lld %8lld %6lld %8lld | %8lld %8lld\n", q, q % g.ntj,
                        q / g.ntj, h[8 * q] - t0, h[8 * q + 1] - t0, h[8 * q + 2], h[8 * q + 3], h[8 * q + 4], h[8 * q + 5] - t0,
                        h[8 * q + 6]);
        } else {
            // start / end / face-wait of every tile in microseconds, rows = K, columns = J (subsampled to 16 x 16)
            const int sj = std::max(1, g.ntj / 16), sk = std::max(1, g.ntk / 16);
            for (int what = 0; what < 3; ++what) {
                fprintf(stderr, "[omg march]  %s (us):\n", what == 0 ? "start" : what == 1 ? "end" : "in face waits");
                for (int K = 0; K < g.ntk; K += sk) {
                    fprintf(stderr, "[omg march]   ");
                    for (int J = 0; J < g.ntj; J += sj) {
                        const int q = K * g.ntj + J;
                        const long long v = what == 2 ? h[8 * q + 2] : h[8 * q + what] - t0;
                        fprintf(stderr, "%5lld", v / 100);
                    }
                    fprintf(stderr, "\n");
                }
            }
        }
    }
}

template <typename V>
bool MarchPlan<V>::timed_out(hipStream_t s) const {
    uint32_t flag = 0;
    OMG_HIP(hipMemcpyAsync(&flag, sync.p + 2, sizeof(flag), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    return flag != 0;
}

template struct MarchPlan<double>;
template struct MarchPlan<float>;

}  // namespace omg
