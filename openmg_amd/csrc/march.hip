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
//   * operands that live in another tile's lines are loaded MARCH_U steps ahead: not yet relaxed
//     ones at any time, relaxed ones once the owning tile has published enough steps
//     (write-through stores -> s_waitcnt vmcnt(0) -> flag; the loads and polls bypass L1/L2
//     staleness with agent-scope loads: MI355X_MICROARCH.md, inter-workgroup visibility).
#include <algorithm>
#include <array>
#include <atomic>
#include <cstring>
#include <thread>

#include "common.h"

namespace omg {

namespace {

constexpr int U = MARCH_U;
constexpr int SYNC_HEAD = 32;      // uint32 words in front of the progress slots
constexpr int SYNC_STRIDE = 16;    // uint32 words per progress slot (64 bytes)

template <typename V>
struct MarchArgs {
    V *x;
    const V *b;
    const uint64_t *codes;
    const V *coef;
    uint32_t *sync;
    int nx, ny, nz, TJ, ntj, n_tiles, T, n_blk, n_pat;
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

// until the tile behind `flag` has completed `need` steps (uniform over the wave)
__device__ __forceinline__ void wait_steps(const uint32_t *flag, uint32_t need, uint32_t &seen) {
    while (seen < need) {
        seen = load_through(flag);
        if (seen < need) __builtin_amdgcn_s_sleep(1);
    }
    asm volatile("" ::: "memory");
}

template <typename V>
struct BlockData {
    V xs[U];     // own line, old value of row i + 1 at step t
    V bv[U];     // right-hand side of row i
    V ej[U];     // the -J (lane jj == 0) or +J (jj == TJ - 1) operand from another tile's line
    V ek[U];     // the same for K
    uint64_t codes;
};

template <typename V>
__global__ __launch_bounds__(64) void march_gs_kernel(MarchArgs<V> a) {
    __shared__ V s_coef[256 * 8];
    __shared__ int s_tile;
    const int lane = threadIdx.x;
    if (lane == 0) s_tile = (int)__hip_atomic_fetch_add(a.sync + 0, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    for (int q = lane; q < a.n_pat * 8; q += 64) s_coef[q] = a.coef[q];
    __syncthreads();
    const int tile = __builtin_amdgcn_readfirstlane(s_tile);
    const int TJ = a.TJ, TK = 64 / TJ;
    const int J = tile % a.ntj, K = tile / a.ntj;
    const int jj = lane % TJ, kk = lane / TJ;
    const int j = J * TJ + jj, k = K * TK + kk;
    const bool valid = j < a.ny && k < a.nz;
    const int skew = jj + kk;
    const int nx = a.nx;
    const int line = valid ? (k * a.ny + j) * nx : 0;
    // operands in other tiles' lines
    const bool lowJ = valid && jj == 0 && j > 0, highJ = valid && jj == TJ - 1 && j + 1 < a.ny;
    const bool lowK = valid && kk == 0 && k > 0, highK = valid && kk == TK - 1 && k + 1 < a.nz;
    const bool extJ = lowJ || highJ, extK = lowK || highK;
    const int offJ = lowJ ? -nx : nx;
    const int offK = lowK ? -nx * a.ny : nx * a.ny;
    const bool face = highJ || highK;                       // another tile will load this lane's results
    uint32_t *progress = a.sync + SYNC_HEAD;
    const uint32_t *flagJ = J > 0 ? progress + size_t(tile - 1) * SYNC_STRIDE : nullptr;
    const uint32_t *flagK = K > 0 ? progress + size_t(tile - a.ntj) * SYNC_STRIDE : nullptr;
    uint32_t seenJ = 0, seenK = 0;
    const int T = a.T;

    BlockData<V> cur, nxt;
    auto prefetch = [&](int blk, BlockData<V> &d) {
        const int T0 = blk * U;
        // the -J tile relaxes row i of its last lane TJ - 1 steps after this tile's step for row i
        if (flagJ) wait_steps(flagJ, (uint32_t)min(T, T0 + U + TJ - 1), seenJ);
        if (flagK) wait_steps(flagK, (uint32_t)min(T, T0 + U + TK - 1), seenK);
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = T0 + u - skew;
            const bool in = valid && i >= 0 && i < nx;
            d.xs[u] = (valid && i + 1 >= 0 && i + 1 < nx) ? a.x[line + i + 1] : V(0);
            d.bv[u] = in ? a.b[line + i] : V(0);
            d.ej[u] = (extJ && in) ? load_through(a.x + line + i + offJ) : V(0);
            d.ek[u] = (extK && in) ? load_through(a.x + line + i + offK) : V(0);
        }
        d.codes = a.codes[(size_t(tile) * a.n_blk + blk) * 64 + lane];
    };

    prefetch(0, nxt);
    V xcur = valid ? a.x[line] : V(0);     // old value of the row of the lane's next step (row 0 first)
    V xlast = V(0);                        // the lane's newest result
    for (int blk = 0; blk < a.n_blk; ++blk) {
        const int T0 = blk * U;
        cur = nxt;
        if (blk > 0) {
            // steps < T0 are complete once the stores behind them have left the wave
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            if (lane == 0) store_through(progress + size_t(tile) * SYNC_STRIDE, (uint32_t)T0);
        }
        if (blk + 1 < a.n_blk) prefetch(blk + 1, nxt);
        V out[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = T0 + u - skew;
            const bool act = valid && i >= 0 && i < nx;
            const int code = int((cur.codes >> (8 * u)) & 255u);
            const V *c = s_coef + code * 8;
            V xjm = __shfl_up(xlast, 1), xkm = __shfl_up(xlast, TJ);
            if (jj == 0) xjm = cur.ej[u];
            if (kk == 0) xkm = cur.ek[u];
            const V xip = cur.xs[u];
            V xjp = __shfl_down(xip, 1), xkp = __shfl_down(xip, TJ);
            if (jj == TJ - 1) xjp = cur.ej[u];
            if (kk == TK - 1) xkp = cur.ek[u];
            V sum = madd(c[0], xkm, V(0));
            sum = madd(c[1], xjm, sum);
            sum = madd(c[2], xlast, sum);
            sum = madd(c[3], xcur, sum);
            sum = madd(c[4], xip, sum);
            sum = madd(c[5], xjp, sum);
            sum = madd(c[6], xkp, sum);
            const V xn = xcur + (cur.bv[u] - sum) / c[3];      // csr_kernels.hip ROW_GS
            out[u] = xn;
            if (act) xlast = xn;
            xcur = xip;
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int i = T0 + u - skew;
            if (valid && i >= 0 && i < nx) {
                if (face) store_through(a.x + line + i, out[u]);
                else a.x[line + i] = out[u];
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (lane == 0) store_through(progress + size_t(tile) * SYNC_STRIDE, (uint32_t)T);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // the last tile to finish leaves the counters as the next sweep expects them
    uint32_t done = 0;
    if (lane == 0) done = __hip_atomic_fetch_add(a.sync + 1, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    done = __builtin_amdgcn_readfirstlane(done);
    if (done == uint32_t(a.n_tiles - 1)) {
        for (int q = lane; q < a.n_tiles; q += 64) store_through(progress + size_t(q) * SYNC_STRIDE, 0u);
        if (lane == 0) {
            store_through(a.sync + 0, 0u);
            store_through(a.sync + 1, 0u);
        }
    }
}

}  // namespace

template <typename V>
bool MarchPlan<V>::build(const omg_csr &A, hipStream_t s) {
    const int64_t n = A.n_rows;
    if (n < 2 || n != A.n_cols || n >= (int64_t(1) << 30)) return false;
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

    // every row as seven coefficients in slot order; stored order must be slot order
    typedef std::array<double, 7> Pat;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    const int nt = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 65536));
    std::vector<uint8_t> code((size_t)n);
    std::vector<std::vector<Pat>> local(nt);
    std::atomic<bool> ok(true);
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
            if (p[3] == 0.0) { ok = false; return; }
            if (hit < pats.size() && !memcmp(&pats[hit], &p, sizeof(Pat))) { code[r] = (uint8_t)hit; continue; }
            size_t f = 0;
            while (f < pats.size() && memcmp(&pats[f], &p, sizeof(Pat))) ++f;
            if (f == pats.size()) {
                if (pats.size() >= 256) { ok = false; return; }
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
    if (!ok) return false;
    std::vector<Pat> pats;
    std::vector<std::vector<uint8_t>> remap(nt);
    for (int t = 0; t < nt; ++t)
        for (const Pat &p : local[t]) {
            size_t f = 0;
            while (f < pats.size() && memcmp(&pats[f], &p, sizeof(Pat))) ++f;
            if (f == pats.size()) {
                if (pats.size() >= 256) return false;
                pats.push_back(p);
            }
            remap[t].push_back((uint8_t)f);
        }

    g.nx = (int)nx; g.ny = (int)ny; g.nz = (int)nz;
    g.TJ = nz > 1 ? 8 : 64;
    g.TK = 64 / g.TJ;
    g.ntj = (g.ny + g.TJ - 1) / g.TJ;
    g.ntk = (g.nz + g.TK - 1) / g.TK;
    g.n_tiles = g.ntj * g.ntk;
    g.T = g.nx + g.TJ + g.TK - 2;
    g.n_blk = (g.T + U - 1) / U;
    g.n_pat = (int)pats.size();

    // the codes as the tiles consume them: one 8-byte word per (tile, block, lane)
    std::vector<uint64_t> words(size_t(g.n_tiles) * g.n_blk * 64);
    auto arrange = [&](int t) {
        for (int tile = t; tile < g.n_tiles; tile += nt) {
            const int J = tile % g.ntj, K = tile / g.ntj;
            for (int lane = 0; lane < 64; ++lane) {
                const int jj = lane % g.TJ, kk = lane / g.TJ;
                const int j = J * g.TJ + jj, k = K * g.TK + kk;
                const bool valid = j < g.ny && k < g.nz;
                const int64_t line = valid ? (int64_t(k) * ny + j) * nx : 0;
                int owner = 0;                       // the scanning thread of the line's rows (for the remap)
                for (int blk = 0; blk < g.n_blk; ++blk) {
                    uint64_t w = 0;
                    for (int u = 0; u < U; ++u) {
                        const int i = blk * U + u - jj - kk;
                        if (!valid || i < 0 || i >= nx) continue;
                        const int64_t r = line + i;
                        while (r >= n * (owner + 1) / nt) ++owner;
                        while (r < n * owner / nt) --owner;
                        w |= uint64_t(remap[owner][code[r]]) << (8 * u);
                    }
                    words[(size_t(tile) * g.n_blk + blk) * 64 + lane] = w;
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
    codes.alloc(words.size());
    coef.alloc(cf.size());
    sync.alloc(size_t(SYNC_HEAD) + size_t(g.n_tiles) * SYNC_STRIDE);
    codes.upload(words.data(), words.size(), s);
    coef.upload(cf.data(), cf.size(), s);
    sync.zero(s);
    OMG_HIP(hipStreamSynchronize(s));
    return true;
}

template <typename V>
void MarchPlan<V>::sweep(V *x, const V *b, hipStream_t s) const {
    MarchArgs<V> a;
    a.x = x; a.b = b; a.codes = codes.p; a.coef = coef.p; a.sync = sync.p;
    a.nx = g.nx; a.ny = g.ny; a.nz = g.nz; a.TJ = g.TJ; a.ntj = g.ntj; a.n_tiles = g.n_tiles;
    a.T = g.T; a.n_blk = g.n_blk; a.n_pat = g.n_pat;
    hipLaunchKernelGGL(march_gs_kernel<V>, dim3((unsigned)g.n_tiles), dim3(64), 0, s, a);
    OMG_HIP(hipGetLastError());
}

template struct MarchPlan<double>;
template struct MarchPlan<float>;

}  // namespace omg
