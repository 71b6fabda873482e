// Coarsest-level direct solve: explicit inverse, or substructuring along the band (common.h
// CoarseSolver).  The reference re-factorises with SuperLU on every cycle
// (openmg/solvers.py:23); here the factors are built once at setup.
#include <algorithm>
#include <cmath>
#include <cstdlib>
#include <type_traits>

#include "common.h"

namespace omg {
namespace {

constexpr int64_t LDS_DOUBLES = 6144;      // 48 KB: g and the longest interior block must fit (as V)

int grid1d(int64_t n, int cap = 65536) {
    int64_t g = (n + 255) / 256;
    if (g > cap) g = cap;
    if (g < 1) g = 1;
    return (int)g;
}

__device__ __forceinline__ double wsum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}
__device__ __forceinline__ float wsum(float v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// T[i, c] = sum_r B[i, r] Aig[r, c] for the interior block: thread (i, c) walks column c of the
// coupling block (given column-wise: cptr / cidx / cval, rows local to the block).
__global__ void couple_kernel(const double *B, int64_t m, const int32_t *cptr, const int32_t *cidx,
                              const double *cval, int64_t g, double *T) {
    const int64_t tot = m * g;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = t / g, c = t % g;
        double acc = 0.0;
        for (int32_t p = cptr[c]; p < cptr[c + 1]; ++p) acc += B[i * m + cidx[p]] * cval[p];
        T[t] = acc;
    }
}

// S[q, c] -= sum_i Agi[q, i] T[i, c]: rows of A_GI restricted to this block (CSR, columns local)
// (S: leading dimension ld — the left half of the augmented matrix its inversion works on)
__global__ void schur_kernel(const int32_t *rptr, const int32_t *ridx, const double *rval, int64_t g,
                             const double *T, double *S, int64_t ld) {
    const int64_t tot = g * g;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < tot; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t q = t / g, c = t % g;
        double acc = 0.0;
        for (int32_t p = rptr[q]; p < rptr[q + 1]; ++p) acc += rval[p] * T[int64_t(ridx[p]) * g + c];
        if (acc != 0.0) S[q * ld + c] -= acc;
    }
}

template <typename V>
__global__ void narrow_kernel(const double *src, V *dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = V(src[i]);
}

// ---- the solve ---------------------------------------------------------------------------
// y_k = B_k b_I_k: one wave per interior row, four rows of one block per workgroup.
template <typename V>
__global__ __launch_bounds__(256) void interior_solve_kernel(const V *__restrict__ binv, const int64_t *__restrict__ binv_off,
                                                             const int64_t *__restrict__ blk_off,
                                                             const int32_t *__restrict__ wg_blk, const int32_t *__restrict__ wg_row,
                                                             const int32_t *__restrict__ perm, const V *__restrict__ b,
                                                             V *__restrict__ y) {
    const int k = wg_blk[blockIdx.x];
    const int64_t o = blk_off[k], m = blk_off[k + 1] - o;
    const int64_t i = wg_row[blockIdx.x] + (threadIdx.x >> 6);
    if (i >= m) return;
    const int lane = threadIdx.x & 63;
    const V *row = binv + binv_off[k] + i * m;
    V acc = V(0);
    for (int64_t j = lane; j < m; j += 64) acc += row[j] * b[perm[o + j]];
    acc = wsum(acc);
    if (lane == 0) y[o + i] = acc;
}

// x_G = S^-1 (b_G - A_GI y): the right-hand side is formed in LDS by the whole workgroup, then
// one wave per row of S^-1.  Also leaves x_G in xg (permuted) for the back substitution.
template <typename V>
__global__ __launch_bounds__(256) void separator_solve_kernel(const V *__restrict__ sinv, int64_t g, int64_t n_int,
                                                              const int32_t *__restrict__ gptr, const int32_t *__restrict__ gidx,
                                                              const V *__restrict__ gval, const int32_t *__restrict__ perm,
                                                              const V *__restrict__ b, const V *__restrict__ y,
                                                              V *__restrict__ xg, V *__restrict__ x) {
    extern __shared__ unsigned char s_raw[];
    V *s_rhs = reinterpret_cast<V *>(s_raw);
    for (int64_t q = threadIdx.x; q < g; q += 256) {
        V acc = V(0);
        for (int32_t p = gptr[q]; p < gptr[q + 1]; ++p) acc += gval[p] * y[gidx[p]];
        s_rhs[q] = b[perm[n_int + q]] - acc;
    }
    __syncthreads();
    const int64_t q = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (q >= g) return;
    const int lane = threadIdx.x & 63;
    const V *row = sinv + q * g;
    V acc = V(0);
    for (int64_t c = lane; c < g; c += 64) acc += row[c] * s_rhs[c];
    acc = wsum(acc);
    if (lane == 0) { xg[q] = acc; x[perm[n_int + q]] = acc; }
}

// x_I_k = y_k - B_k t, t = A_I_kG x_G: t is nonzero only within w of the block's ends, so only
// those columns of B_k are read.  t is formed in LDS by the whole workgroup.
template <typename V>
__global__ __launch_bounds__(256) void interior_correct_kernel(const V *__restrict__ binv, const int64_t *__restrict__ binv_off,
                                                               const int64_t *__restrict__ blk_off,
                                                               const int32_t *__restrict__ wg_blk, const int32_t *__restrict__ wg_row,
                                                               const int32_t *__restrict__ iptr, const int32_t *__restrict__ iidx,
                                                               const V *__restrict__ ival, const V *__restrict__ xg,
                                                               const int32_t *__restrict__ perm, const V *__restrict__ y,
                                                               int64_t w, V *__restrict__ x) {
    extern __shared__ unsigned char s_raw[];
    V *s_t = reinterpret_cast<V *>(s_raw);          // 2 w entries: the first w and the last w rows of the block
    const int k = wg_blk[blockIdx.x];
    const int64_t o = blk_off[k], m = blk_off[k + 1] - o;
    const int64_t head = min(w, m), tail0 = max(head, m - w);       // columns [0, head) and [tail0, m)
    const int64_t nt = head + (m - tail0);
    for (int64_t c = threadIdx.x; c < nt; c += 256) {
        const int64_t j = c < head ? c : tail0 + (c - head);
        V acc = V(0);
        for (int32_t p = iptr[o + j]; p < iptr[o + j + 1]; ++p) acc += ival[p] * xg[iidx[p]];
        s_t[c] = acc;
    }
    __syncthreads();
    const int64_t i = wg_row[blockIdx.x] + (threadIdx.x >> 6);
    if (i >= m) return;
    const int lane = threadIdx.x & 63;
    const V *row = binv + binv_off[k] + i * m;
    V acc = V(0);
    for (int64_t c = lane; c < nt; c += 64) {
        const int64_t j = c < head ? c : tail0 + (c - head);
        acc += row[j] * s_t[c];
    }
    acc = wsum(acc);
    if (lane == 0) x[perm[o + i]] = y[o + i] - acc;
}

// One step of the block chain (CoarseSolver P == -1) over block [r0, r0 + m): the workgroup forms the step's vector in LDS
// — forward t = b_k - A_{k,k-1} z_{k-1}, backward t = A_{k,k+1} x_{k+1}, from the rows' entries outside their block —, then
// a wave takes four rows of D'_k^-1 at a time.  Forward: z_k = D'^-1 t (and x_k = z_k in the last block); backward:
// x_k = z_k - D'^-1 t.
constexpr int CHAIN_ROWS = 16;      // rows of the block per workgroup (four per wave)
template <typename V, bool BACK>
__global__ __launch_bounds__(256) void chain_step_kernel(const V *__restrict__ dinv, int64_t m, int64_t r0, const int32_t *__restrict__ ptr,
                                                         const int32_t *__restrict__ idx, const V *__restrict__ val,
                                                         const V *__restrict__ b, const V *src, V *z, V *x, int last) {
    extern __shared__ unsigned char s_raw[];
    V *s_t = reinterpret_cast<V *>(s_raw);
    for (int64_t q = threadIdx.x; q < m; q += 256) {
        V acc = V(0);
        for (int32_t p = ptr[r0 + q]; p < ptr[r0 + q + 1]; ++p) acc += val[p] * src[idx[p]];
        s_t[q] = BACK ? acc : b[r0 + q] - acc;
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    const int64_t q0 = int64_t(blockIdx.x) * CHAIN_ROWS + (threadIdx.x >> 6) * 4;
    V acc[4] = {V(0), V(0), V(0), V(0)};
    const V *row[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) row[r] = dinv + min(q0 + r, m - 1) * m;
    for (int64_t j = lane; j < m; j += 64) {
        const V t = s_t[j];
#pragma unroll
        for (int r = 0; r < 4; ++r) acc[r] += row[r][j] * t;
    }
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        const V v = wsum(acc[r]);
        if (lane == 0 && q0 + r < m) {
            if (BACK) x[r0 + q0 + r] = z[r0 + q0 + r] - v;
            else { z[r0 + q0 + r] = v; if (last) x[r0 + q0 + r] = v; }
        }
    }
}

struct HostSub {            // small host CSR / CSC piece
    std::vector<int32_t> ptr, idx;
    std::vector<double> val;
};

template <typename T, typename Al>
void put(DevBuf<T> &d, const std::vector<T, Al> &h, hipStream_t s) {
    d.alloc(std::max<size_t>(h.size(), 1));
    d.upload(h.data(), h.size(), s);
}

template <typename V>
void put_values(DevBuf<V> &d, const std::vector<double> &h, hipStream_t s, std::vector<V> &keep) {
    keep.assign(h.begin(), h.end());
    d.alloc(std::max<size_t>(keep.size(), 1));
    d.upload(keep.data(), keep.size(), s);
}

// ---- sine-transform solve (CoarseSolver P == 0) ---------------------------------------------------------
constexpr int SINE_THREADS = 1024;
// x = S ((S b) / lambda), S = Sz (x) Sy (x) Sx: six passes of short dense transforms through LDS.  A pass is
// out(ne x lines) = S(ne x ne) in(ne x lines) — a small dense matrix product, and the one place of this path where the
// matrix cores earn their keep: v_mfma_f64_16x16x4 (A: lane 16 k + i holds S[i][k]; B: lane 16 k + j holds in[k][line j];
// D: register r of lane l holds row 4 r + l / 16 of column l % 16).  A wave takes 16 lines at a time, all of their outputs, so a
// pass works in place; one workgroup barrier per pass.  E (16 or 32) >= every extent: the tables' row stride.
typedef double v4d __attribute__((ext_vector_type(4)));
template <typename V, int E>
__global__ __launch_bounds__(SINE_THREADS) void sine_solve_kernel(const V *__restrict__ b, V *__restrict__ x, int nx, int ny, int nz,
                                                                 const double *__restrict__ tables, const double *__restrict__ lambda) {
    extern __shared__ double sine_buf[];              // (i, j, k) at (k ny + j) (nx + 1) + i: lines along x one bank apart
    const int px = nx + 1, n = nx * ny * nz;
    double *const lt = sine_buf + ((px * ny * nz + 1) & ~1);       // the three tables behind the vector
    const int nt = (nx + ny + nz) * E;
    for (int r = int(threadIdx.x); r < nt; r += SINE_THREADS) lt[r] = tables[r];
    // a thread's elements r = tid + 1024 q (n <= 8192: q < 8): their places in the padded image and their eigenvalues,
    // fetched once (the kernel is a chain of latencies: nothing global is left between the passes)
    constexpr int PER = 8192 / SINE_THREADS;
    int place[PER];
    double lam[PER];
#pragma unroll
    for (int q = 0; q < PER; ++q) {
        const int r = int(threadIdx.x) + q * SINE_THREADS;
        place[q] = r < n ? (r / nx) * px + r % nx : -1;
        lam[q] = r < n ? lambda[r] : 1.0;
        if (r < n) sine_buf[place[q]] = double(b[r]);
    }
    __syncthreads();
    const int lane = int(threadIdx.x) & 63, wave = int(threadIdx.x) >> 6;
    const int c16 = lane & 15, k4 = lane >> 4;
#pragma unroll 1
    for (int pass = 0; pass < 6; ++pass) {
        const int axis = pass % 3;
        // a line's first element: (line / da) sa + (line % da) sb; its elements `stride` apart
        const int ne = axis == 0 ? nx : axis == 1 ? ny : nz;
        const int n_lines = n / ne;
        const int da = axis == 0 ? 1 : nx, sb = axis == 0 ? 0 : 1;
        const int sa = axis == 0 ? px : axis == 1 ? ny * px : px;
        const int stride = axis == 0 ? 1 : axis == 1 ? px : ny * px;
        const double *S = lt + (axis == 0 ? 0 : axis == 1 ? nx : nx + ny) * E;
        if (pass == 3) {
#pragma unroll
            for (int q = 0; q < PER; ++q)
                if (place[q] >= 0) sine_buf[place[q]] = sine_buf[place[q]] / lam[q];
            __syncthreads();
        }
        if (ne <= 1) continue;                        // a 1 x 1 transform is the identity
        const int n_blocks = (n_lines + 15) >> 4, k_steps = (ne + 3) >> 2;
#pragma unroll 1
        for (int bl = wave; bl < n_blocks; bl += SINE_THREADS / 64) {
            const int line = 16 * bl + c16;
            const bool live = line < n_lines;
            const int bs = live ? (line / da) * sa + (line % da) * sb : 0;
            v4d acc[E / 16];
#pragma unroll
            for (int ib = 0; ib < E / 16; ++ib) acc[ib] = v4d{0.0, 0.0, 0.0, 0.0};
            // all the operands first (E / 4 steps whatever the extent: the tables are zero beyond it), then the chain of
            // matrix instructions: the LDS round trips overlap instead of one per step
            double bvs[E / 4], avs[E / 16][E / 4];
#pragma unroll
            for (int ks = 0; ks < E / 4; ++ks) {
                const int k = 4 * ks + k4;
                bvs[ks] = (live && k < ne) ? sine_buf[bs + k * stride] : 0.0;
#pragma unroll
                for (int ib = 0; ib < E / 16; ++ib) avs[ib][ks] = S[(16 * ib + c16) * E + k];
            }
#pragma unroll
            for (int ks = 0; ks < E / 4; ++ks)
#pragma unroll
                for (int ib = 0; ib < E / 16; ++ib)
                    if (ks < k_steps) acc[ib] = __builtin_amdgcn_mfma_f64_16x16x4f64(avs[ib][ks], bvs[ks], acc[ib], 0, 0, 0);
            // (the wave has read all it needs of its 16 lines: LDS operations of a wave complete in order)
            if (live) {
#pragma unroll
                for (int ib = 0; ib < E / 16; ++ib)
#pragma unroll
                    for (int r = 0; r < 4; ++r) {
                        const int i = 16 * ib + 4 * r + k4;       // (measured layout: tools/mfma_probe.hip)
                        if (i < ne) sine_buf[bs + i * stride] = acc[ib][r];
                    }
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int q = 0; q < PER; ++q)
        if (place[q] >= 0) x[int(threadIdx.x) + q * SINE_THREADS] = V(sine_buf[place[q]]);
}

// The same solve for the 16 x 16 x 16 grid (the coarsest level of 256^3 with five grids and of 512^3 with six) with two
// workgroup barriers instead of eight.  Sixteen waves; wave w owns PLANE k = w for the transforms along x and y (its own
// 16 lines either way: what it wrote it reads back itself, LDS operations of a wave complete in order) and the z-LINES
// (j = w, every i) for the transforms along z.  The right-hand side and the tables come straight from global memory
// into the matrix instructions' operands, the result goes straight out of them; the forward transform along z, the
// division by the eigenvalues and the inverse transform along z happen in registers (the D layout of
// v_mfma_f64_16x16x4 — register r of lane l: row 4 r + l / 16, column l % 16 — IS the B layout of k-step r).
// Order of the six transforms: x y z | z x y (the general kernel: x y z | x y z) — the same sums, associated as before
// within each transform; 15.5 -> see DESIGN.md section 5e.
// Lanes of ONE wave exchange values through LDS: the stores of all lanes must be complete and visible before any lane's
// loads of the transposed addresses (nothing kept the compiler from reordering the may-alias accesses).
__device__ __forceinline__ void wave_lds_exchange() {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
}

template <typename V>
__global__ __launch_bounds__(SINE_THREADS) void sine_cube16_kernel(const V *__restrict__ b, V *__restrict__ x, const double *__restrict__ tables,
                                                                   const double *__restrict__ lambda) {
    constexpr int N = 16, PX = N + 1;
    __shared__ double buf[N * N * PX];                 // (i, j, k) at (k N + j) PX + i
    const int lane = int(threadIdx.x) & 63, w = int(threadIdx.x) >> 6;
    const int c = lane & 15, k4 = lane >> 4;
    // A operands: lane 16 k + i holds S[i][k]; k-step ks: k = 4 ks + k4
    double sx[4], sy[4], sz[4], lam[4], rhs[4];
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) {
        sx[ks] = tables[(0 * N + c) * N + 4 * ks + k4];
        sy[ks] = tables[(1 * N + c) * N + 4 * ks + k4];
        sz[ks] = tables[(2 * N + c) * N + 4 * ks + k4];
        rhs[ks] = double(b[(w * N + c) * N + 4 * ks + k4]);                 // line (k = w, j = c), element i = 4 ks + k4
        lam[ks] = lambda[((4 * ks + k4) * N + w) * N + c];                  // element (i = c, j = w, k = 4 ks + k4)
    }
    auto transform = [&](const double (&S)[4], const double (&in)[4]) -> v4d {
        v4d acc = {0.0, 0.0, 0.0, 0.0};
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) acc = __builtin_amdgcn_mfma_f64_16x16x4f64(S[ks], in[ks], acc, 0, 0, 0);
        return acc;
    };
    double in[4];
    // ---- forward along x and y, plane k = w -------------------------------------------------------------------
    v4d acc = transform(sx, rhs);                      // rows i' = 4 r + k4 of line j = c
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(w * N + c) * PX + 4 * r + k4] = acc[r];
    wave_lds_exchange();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) in[ks] = buf[(w * N + 4 * ks + k4) * PX + c];       // line (k = w, i = c), element j
    acc = transform(sy, in);                           // rows j' = 4 r + k4 of line i = c
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(w * N + 4 * r + k4) * PX + c] = acc[r];
    __syncthreads();
    // ---- along z: forward, the division, inverse — lines (j = w, i = c), in registers -----------------------------
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) in[ks] = buf[((4 * ks + k4) * N + w) * PX + c];
    acc = transform(sz, in);
#pragma unroll
    for (int r = 0; r < 4; ++r) in[r] = acc[r] / lam[r];
    acc = transform(sz, in);
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[((4 * r + k4) * N + w) * PX + c] = acc[r];
    __syncthreads();
    // ---- inverse along x and y, plane k = w; the result leaves from the matrix instruction's registers --------------
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) in[ks] = buf[(w * N + c) * PX + 4 * ks + k4];
    acc = transform(sx, in);
#pragma unroll
    for (int r = 0; r < 4; ++r) buf[(w * N + c) * PX + 4 * r + k4] = acc[r];
    wave_lds_exchange();
#pragma unroll
    for (int ks = 0; ks < 4; ++ks) in[ks] = buf[(w * N + 4 * ks + k4) * PX + c];
    acc = transform(sy, in);
#pragma unroll
    for (int r = 0; r < 4; ++r) x[(w * N + 4 * r + k4) * N + c] = V(acc[r]);
}

// Is A a constant-coefficient SYMMETRIC star stencil on a lexicographically numbered grid whose boundary rows
// drop the entries of missing neighbours?  (extents and the four coefficients: diagonal, x, y, z couplings)
bool detect_symmetric_star(const HostCsr &A, int64_t &nx, int64_t &ny, int64_t &nz, double (&c)[4]) {
    const int64_t n = A.n_rows;
    if (n < 2) return false;
    auto has = [&](int64_t r, int64_t col) {
        for (int32_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p)
            if (A.indices[p] == col) return true;
        return false;
    };
    nx = n;
    for (int64_t r = 1; r < n; ++r)
        if (!has(r, r - 1)) { nx = r; break; }
    if (nx < 2 || n % nx) return false;
    const int64_t lines = n / nx;
    ny = lines;
    for (int64_t q = 1; q < lines; ++q)
        if (!has(q * nx, (q - 1) * nx)) { ny = q; break; }
    if (lines % ny) return false;
    nz = lines / ny;
    const int64_t sj = nx, sk = nx * ny;
    bool have[4] = {false, false, false, false};
    c[0] = c[1] = c[2] = c[3] = 0.0;
    for (int64_t r = 0; r < n; ++r) {
        const int64_t i = r % nx, jl = (r / nx) % ny, kl = r / sk;
        const int want = 1 + (i > 0) + (i + 1 < nx) + (jl > 0) + (jl + 1 < ny) + (kl > 0) + (kl + 1 < nz);
        if (A.indptr[r + 1] - A.indptr[r] != want) return false;
        int seen = 0;
        for (int32_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p) {
            const int64_t off = int64_t(A.indices[p]) - r;
            int slot, bit;
            if (off == 0) { slot = 0; bit = 1; }
            else if (off == -1 && i > 0) { slot = 1; bit = 2; }
            else if (off == 1 && i + 1 < nx) { slot = 1; bit = 4; }
            else if (off == -sj && jl > 0 && ny > 1) { slot = 2; bit = 8; }
            else if (off == sj && jl + 1 < ny) { slot = 2; bit = 16; }
            else if (off == -sk && kl > 0 && nz > 1) { slot = 3; bit = 32; }
            else if (off == sk && kl + 1 < nz) { slot = 3; bit = 64; }
            else return false;
            if (seen & bit) return false;
            seen |= bit;
            if (!have[slot]) { have[slot] = true; c[slot] = A.data[p]; }
            else if (A.data[p] != c[slot]) return false;
        }
    }
    return have[0];
}

}  // namespace

template <typename V>
void CoarseSolver<V>::build(const HostCsr &A, hipStream_t s) {
    OMG_REQUIRE(A.n_rows == A.n_cols, "coarse operator must be square");
    n = A.n_rows;
    if (n == 0) return;
    {   // OMG_COARSE_BLOCKS forces the inverse / substructuring, OMG_COARSE_SINE=0 switches the sine solve off
        const char *e = getenv("OMG_COARSE_BLOCKS"), *q = getenv("OMG_COARSE_SINE");
        if (!(e && atoi(e) > 0) && !(q && q[0] == '0') && build_sine(A, s)) return;
    }
    // half-bandwidth
    int64_t band = 0;
    for (int64_t i = 0; i < n; ++i)
        for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) band = std::max<int64_t>(band, std::llabs(int64_t(A.indices[p]) - i));
    w = std::max<int64_t>(band, 1);
    // number of interior blocks: least stored bytes subject to the LDS limits of the solve kernels
    const int64_t lds_cap = LDS_DOUBLES * int64_t(sizeof(double) / sizeof(V));      // elements of V in 48 KB
    P = 1;
    {
        const char *e = getenv("OMG_COARSE_BLOCKS");         // 1 forces the explicit inverse
        const int forced = e ? atoi(e) : 0;
        // Measured at 256^3 / 5 grids (n = 4096, w = 256): the inverse's one 134 MB mat-vec takes
        // 26 us, the three dependent launches of the substructured solve 39 us although they read
        // 40 MB — so substructuring is for operators whose inverse would not even fit the 256 MiB
        // Infinity Cache (n > 5792 in double): 128^2 5-point, 0.5 ms -> 0.05 ms per solve.
        const bool big = double(n) * double(n) * sizeof(V) > double(size_t(256) << 20);
        double best = double(n) * double(n);
        if (forced > 1 || (forced == 0 && big))
            for (int q = 2; q <= 64; ++q) {
                const int64_t gq = int64_t(q - 1) * w;
                if (gq >= n || gq > lds_cap || 2 * w > lds_cap) break;
                const int64_t mq = (n - gq + q - 1) / q;
                if (mq < 2 * w) break;                        // blocks shorter than their couplings: no gain
                const double cost = double(q) * double(mq) * double(mq) + double(gq) * double(gq);
                if ((forced == q) || (forced == 0 && cost < best)) { best = cost; P = q; }
                if (forced == q) break;
            }
    }
    std::vector<int32_t> plain_ptr(A.indptr.begin(), A.indptr.end());
    DevBuf<int32_t> d_ptr, d_idx;
    DevBuf<double> d_val;
    put(d_ptr, plain_ptr, s);
    put(d_idx, A.indices, s);
    put(d_val, A.data, s);
    {   // OMG_COARSE_CHAIN=1 forces the block chain (tests: small operators through the path large ones take)
        const char *e = getenv("OMG_COARSE_CHAIN");
        if ((P == 1 && n > 16384) || (e && e[0] == '1' && n >= 4)) {
            build_chain(A, d_ptr.p, d_idx.p, d_val.p, s);
            return;
        }
    }
    if (P == 1) {
        std::vector<int32_t> ident((size_t)(n));
        for (int64_t i = 0; i < n; ++i) ident[i] = int32_t(i);
        DevBuf<int32_t> d_map;
        put(d_map, ident, s);
        const size_t nn = (size_t)(n) * (size_t)(n);
        DevBuf<double> W_own, inv64_own;
        DevBuf<double> &W = retain_workspace ? ws_aug : W_own, &inv64 = retain_workspace ? ws_inv64 : inv64_own;
        if (W.n != 2 * nn) W.alloc(2 * nn);
        if (inv64.n != nn) inv64.alloc(nn);
        if (retain_workspace && ws_keep.n != nn) ws_keep.alloc(nn);
        fill_augmented_from_csr(d_ptr.p, d_idx.p, d_val.p, n, d_map.p, d_map.p, n, W.p, s);
        gauss_jordan_inverse(W.p, n, inv64.p, s, retain_workspace ? ws_keep.p : nullptr);
        if (inv.n != nn) inv.alloc(nn);
        hipLaunchKernelGGL((narrow_kernel<V>), dim3(grid1d(n * n)), dim3(256), 0, s, inv64.p, inv.p, n * n);
        OMG_HIP(hipStreamSynchronize(s));
        bytes = (size_t)(n) * (size_t)(n) * sizeof(V);
        return;
    }
    // ---- partition: [I_0][G_0][I_1][G_1] ... [I_{P-1}] in the original numbering -------------
    g = int64_t(P - 1) * w;
    n_int = n - g;
    std::vector<int64_t> h_off((size_t)(P) + 1, 0), h_boff((size_t)(P) + 1, 0), start((size_t)(P), 0);
    std::vector<int32_t> h_perm((size_t)(n)), to_int((size_t)(n), -1), to_gam((size_t)(n), -1), blk_of((size_t)(n), -1);
    {
        int64_t pos = 0;
        for (int k = 0; k < P; ++k) {
            const int64_t m = n_int / P + (k < n_int % P ? 1 : 0);
            start[k] = pos;
            h_off[k + 1] = h_off[k] + m;
            h_boff[k + 1] = h_boff[k] + m * m;
            for (int64_t j = 0; j < m; ++j) {
                h_perm[h_off[k] + j] = int32_t(pos + j);
                to_int[pos + j] = int32_t(h_off[k] + j);
                blk_of[pos + j] = k;
            }
            pos += m;
            if (k + 1 < P)
                for (int64_t j = 0; j < w; ++j) {
                    h_perm[n_int + int64_t(k) * w + j] = int32_t(pos + j);
                    to_gam[pos + j] = int32_t(int64_t(k) * w + j);
                }
            pos += (k + 1 < P) ? w : 0;
        }
    }
    // sparse couplings (host): A_GI by rows (columns permuted interior), A_IG by rows (columns
    // Gamma-local); per block: A_IG column-wise with block-local rows, A_GI rows with block-local columns
    HostSub gi, ig;
    gi.ptr.assign((size_t)(g) + 1, 0);
    ig.ptr.assign((size_t)(n_int) + 1, 0);
    std::vector<HostSub> col_k((size_t)(P)), row_k((size_t)(P));
    for (int k = 0; k < P; ++k) { col_k[k].ptr.assign((size_t)(g) + 1, 0); row_k[k].ptr.assign((size_t)(g) + 1, 0); }
    for (int64_t q = 0; q < g; ++q) {
        const int64_t i = h_perm[n_int + q];
        for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) {
            const int32_t c = A.indices[p];
            if (to_int[c] >= 0) {
                gi.idx.push_back(to_int[c]);
                gi.val.push_back(A.data[p]);
                HostSub &rk = row_k[blk_of[c]];
                rk.idx.push_back(int32_t(to_int[c] - h_off[blk_of[c]]));
                rk.val.push_back(A.data[p]);
                rk.ptr[q + 1]++;
            }
        }
        gi.ptr[q + 1] = int32_t(gi.idx.size());
    }
    for (int k = 0; k < P; ++k)
        for (int64_t q = 0; q < g; ++q) row_k[k].ptr[q + 1] += row_k[k].ptr[q];
    {
        std::vector<std::vector<std::pair<int32_t, double>>> cols((size_t)(g));
        for (int64_t r = 0; r < n_int; ++r) {
            const int64_t i = h_perm[r];
            for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) {
                const int32_t c = A.indices[p];
                if (to_gam[c] >= 0) {
                    ig.idx.push_back(to_gam[c]);
                    ig.val.push_back(A.data[p]);
                    OMG_REQUIRE(blk_of[i] >= 0, "coarse solve: partition error");
                }
            }
            ig.ptr[r + 1] = int32_t(ig.idx.size());
        }
        for (int k = 0; k < P; ++k) {
            for (auto &c : cols) c.clear();
            for (int64_t r = h_off[k]; r < h_off[k + 1]; ++r)
                for (int32_t p = ig.ptr[r]; p < ig.ptr[r + 1]; ++p) cols[ig.idx[p]].emplace_back(int32_t(r - h_off[k]), ig.val[p]);
            HostSub &ck = col_k[k];
            for (int64_t c = 0; c < g; ++c) {
                for (auto &e : cols[c]) { ck.idx.push_back(e.first); ck.val.push_back(e.second); }
                ck.ptr[c + 1] = int32_t(ck.idx.size());
            }
        }
    }
    // ---- device: B_k, Schur complement, its inverse ----------------------------------------
    DevBuf<int32_t> d_toint, d_togam;
    put(d_toint, to_int, s);
    put(d_togam, to_gam, s);
    DevBuf<double> b64((size_t)(h_boff[P])), S((size_t)(g) * (size_t)(g) * 2), sinv64((size_t)(g) * (size_t)(g));
    // S starts as A_GG (built as the left half of an augmented [A_GG | I], reused for its inversion)
    fill_augmented_from_csr(d_ptr.p, d_idx.p, d_val.p, n, d_togam.p, d_togam.p, g, S.p, s);
    int64_t m_max = 0;
    for (int k = 0; k < P; ++k) m_max = std::max(m_max, h_off[k + 1] - h_off[k]);
    DevBuf<double> W((size_t)(m_max) * (size_t)(2 * m_max)), T((size_t)(m_max) * (size_t)(g));
    std::vector<int32_t> rmap((size_t)(n));
    DevBuf<int32_t> d_rmap((size_t)(n)), c_ptr, c_idx, r_ptr, r_idx;
    DevBuf<double> c_val, r_val;
    for (int k = 0; k < P; ++k) {
        const int64_t m = h_off[k + 1] - h_off[k];
        for (int64_t i = 0; i < n; ++i) rmap[i] = (blk_of[i] == k) ? int32_t(to_int[i] - h_off[k]) : -1;
        d_rmap.upload(rmap.data(), (size_t)(n), s);
        OMG_HIP(hipStreamSynchronize(s));
        fill_augmented_from_csr(d_ptr.p, d_idx.p, d_val.p, n, d_rmap.p, d_rmap.p, m, W.p, s);
        gauss_jordan_inverse(W.p, m, b64.p + h_boff[k], s);
        put(c_ptr, col_k[k].ptr, s); put(c_idx, col_k[k].idx, s); put(c_val, col_k[k].val, s);
        put(r_ptr, row_k[k].ptr, s); put(r_idx, row_k[k].idx, s); put(r_val, row_k[k].val, s);
        hipLaunchKernelGGL(couple_kernel, dim3(grid1d(m * g)), dim3(256), 0, s, b64.p + h_boff[k], m, c_ptr.p, c_idx.p,
                           c_val.p, g, T.p);
        // S -= A_GI_k T
        hipLaunchKernelGGL(schur_kernel, dim3(grid1d(g * g)), dim3(256), 0, s, r_ptr.p, r_idx.p, r_val.p, g, T.p, S.p, 2 * g);
        OMG_HIP(hipGetLastError());
        OMG_HIP(hipStreamSynchronize(s));          // host staging of this block may be reused
    }
    gauss_jordan_inverse(S.p, g, sinv64.p, s);
    // ---- stored factors (V) ----------------------------------------------------------------
    binv.alloc((size_t)(h_boff[P]));
    sinv.alloc((size_t)(g) * (size_t)(g));
    hipLaunchKernelGGL((narrow_kernel<V>), dim3(grid1d(h_boff[P])), dim3(256), 0, s, b64.p, binv.p, h_boff[P]);
    hipLaunchKernelGGL((narrow_kernel<V>), dim3(grid1d(g * g)), dim3(256), 0, s, sinv64.p, sinv.p, g * g);
    put(blk_off, h_off, s);
    put(binv_off, h_boff, s);
    put(perm, h_perm, s);
    std::vector<int32_t> h_wblk, h_wrow;
    for (int k = 0; k < P; ++k)
        for (int64_t r = 0; r < h_off[k + 1] - h_off[k]; r += 4) { h_wblk.push_back(k); h_wrow.push_back(int32_t(r)); }
    n_wg = int64_t(h_wblk.size());
    put(wg_blk, h_wblk, s);
    put(wg_row, h_wrow, s);
    std::vector<V> keep1, keep2;
    put(gi_ptr, gi.ptr, s); put(gi_idx, gi.idx, s); put_values(gi_val, gi.val, s, keep1);
    put(ig_ptr, ig.ptr, s); put(ig_idx, ig.idx, s); put_values(ig_val, ig.val, s, keep2);
    y.alloc((size_t)(n_int));
    xg.alloc((size_t)(g));
    OMG_HIP(hipStreamSynchronize(s));
    size_t corr = 0;
    for (int k = 0; k < P; ++k) {
        const int64_t m = h_off[k + 1] - h_off[k];
        corr += (size_t)(m) * (size_t)(std::min<int64_t>(2 * w, m));
    }
    bytes = ((size_t)(h_boff[P]) + (size_t)(g) * (size_t)(g) + corr) * sizeof(V);
}

// The block chain (common.h CoarseSolver, P == -1).  openmg/solvers.py:16-26 solves ANY coarsest operator; this is the
// path that has no size limit but memory: n bs values of V for the K inverses.
template <typename V>
void CoarseSolver<V>::build_chain(const HostCsr &A, const int32_t *d_ptr, const int32_t *d_idx, const double *d_val, hipStream_t s) {
    // block size: the half-bandwidth, in multiples of 64 (forced runs on tiny operators: at least 2 blocks where possible)
    bs = std::max<int64_t>((w + 63) / 64 * 64, 64);
    {
        const char *e = getenv("OMG_COARSE_CHAIN_BS");      // (tests: several blocks on small operators; must be >= the half-bandwidth)
        if (e && atol(e) >= w) bs = atol(e);
    }
    bs = std::min(bs, n);
    K = int((n + bs - 1) / bs);
    const size_t lds_need = size_t(bs) * sizeof(V);
    size_t free_b = 0, total_b = 0;
    OMG_HIP(hipMemGetInfo(&free_b, &total_b));
    const double need = double(K) * double(bs) * double(bs) * sizeof(V) + 4.0 * double(bs) * double(bs) * sizeof(double);
    if (lds_need > size_t(128) << 10 || need > 0.8 * double(free_b))
        throw Error(OMG_ERR_UNSUPPORTED, "coarsest level has " + std::to_string(n) + " unknowns and a half-bandwidth of " + std::to_string(w) +
                                             ": the block elimination along the band would need " + std::to_string(int64_t(need / 1e9)) +
                                             " GB of device memory (" + std::to_string(int64_t(free_b / 1e9)) + " GB free) and " +
                                             std::to_string(lds_need >> 10) + " KB of LDS per step (128) — use more gridLevels");
    P = -1;
    // every row's entries outside its own block, original columns (the solve's couplings)
    HostSub lo, up;
    lo.ptr.assign(size_t(n) + 1, 0);
    up.ptr.assign(size_t(n) + 1, 0);
    for (int64_t r = 0; r < n; ++r) {
        const int64_t k = r / bs, c0 = k * bs, c1 = std::min(n, c0 + bs);
        for (int32_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p) {
            const int64_t c = A.indices[p];
            if (c < c0) { OMG_REQUIRE(c >= c0 - bs, "block chain: an entry reaches beyond the neighbouring block"); lo.idx.push_back(int32_t(c)); lo.val.push_back(A.data[p]); }
            else if (c >= c1) { OMG_REQUIRE(c < c1 + bs, "block chain: an entry reaches beyond the neighbouring block"); up.idx.push_back(int32_t(c)); up.val.push_back(A.data[p]); }
        }
        lo.ptr[size_t(r) + 1] = int32_t(lo.idx.size());
        up.ptr[size_t(r) + 1] = int32_t(up.idx.size());
    }
    std::vector<V> keep1, keep2;
    put(lo_ptr, lo.ptr, s); put(lo_idx, lo.idx, s); put_values(lo_val, lo.val, s, keep1);
    put(up_ptr, up.ptr, s); put(up_idx, up.idx, s); put_values(up_val, up.val, s, keep2);
    dinv.alloc(size_t(K) * size_t(bs) * size_t(bs));
    z.alloc(size_t(n));
    DevBuf<double> W(size_t(bs) * size_t(2 * bs)), inv64(size_t(bs) * size_t(bs)), T(size_t(bs) * size_t(bs)), keep(size_t(bs) * size_t(bs));
    std::vector<int32_t> rmap(size_t(n), -1);
    DevBuf<int32_t> d_rmap{size_t(n)}, c_ptr, c_idx, r_ptr, r_idx;
    DevBuf<double> c_val, r_val;
    for (int k = 0; k < K; ++k) {
        const int64_t r0 = int64_t(k) * bs, m = std::min(n, r0 + bs) - r0;
        if (k > 0) for (int64_t i = r0 - bs; i < r0; ++i) rmap[size_t(i)] = -1;
        for (int64_t i = 0; i < m; ++i) rmap[size_t(r0 + i)] = int32_t(i);
        d_rmap.upload(rmap.data(), size_t(n), s);
        OMG_HIP(hipStreamSynchronize(s));
        fill_augmented_from_csr(d_ptr, d_idx, d_val, n, d_rmap.p, d_rmap.p, m, W.p, s);          // [D_k | I]
        if (k > 0) {
            // D'_k = D_k - A_{k,k-1} (D'_{k-1}^-1 A_{k-1,k}): the couplings are sparse, so the update is two passes over the
            // previous inverse's entries, not a dense product.  X = D'_{k-1}^-1 A_{k-1,k} (mp x m), column-wise access to
            // A_{k-1,k}; then the left half of W -= A_{k,k-1} X.
            const int64_t p0 = r0 - bs, mp = bs;
            HostSub col, row;                                  // A_{k-1,k} by columns (rows local to block k-1); A_{k,k-1} by rows (columns local)
            std::vector<std::vector<std::pair<int32_t, double>>> cols{size_t(m)};
            for (int64_t r = p0; r < r0; ++r)
                for (int32_t p = up.ptr[size_t(r)]; p < up.ptr[size_t(r) + 1]; ++p) cols[size_t(up.idx[size_t(p)] - r0)].emplace_back(int32_t(r - p0), up.val[size_t(p)]);
            col.ptr.assign(size_t(m) + 1, 0);
            for (int64_t c = 0; c < m; ++c) {
                for (auto &e : cols[size_t(c)]) { col.idx.push_back(e.first); col.val.push_back(e.second); }
                col.ptr[size_t(c) + 1] = int32_t(col.idx.size());
            }
            row.ptr.assign(size_t(m) + 1, 0);
            for (int64_t q = 0; q < m; ++q) {
                for (int32_t p = lo.ptr[size_t(r0 + q)]; p < lo.ptr[size_t(r0 + q) + 1]; ++p) { row.idx.push_back(int32_t(lo.idx[size_t(p)] - p0)); row.val.push_back(lo.val[size_t(p)]); }
                row.ptr[size_t(q) + 1] = int32_t(row.idx.size());
            }
            put(c_ptr, col.ptr, s); put(c_idx, col.idx, s); put(c_val, col.val, s);
            put(r_ptr, row.ptr, s); put(r_idx, row.idx, s); put(r_val, row.val, s);
            hipLaunchKernelGGL(couple_kernel, dim3(grid1d(mp * m)), dim3(256), 0, s, inv64.p, mp, c_ptr.p, c_idx.p, c_val.p, m, T.p);
            hipLaunchKernelGGL(schur_kernel, dim3(grid1d(m * m)), dim3(256), 0, s, r_ptr.p, r_idx.p, r_val.p, m, T.p, W.p, 2 * m);
            OMG_HIP(hipGetLastError());
        }
        gauss_jordan_inverse(W.p, m, inv64.p, s, keep.p);
        hipLaunchKernelGGL((narrow_kernel<V>), dim3(grid1d(m * m)), dim3(256), 0, s, inv64.p, dinv.p + size_t(k) * size_t(bs) * size_t(bs), m * m);
        OMG_HIP(hipGetLastError());
        OMG_HIP(hipStreamSynchronize(s));                      // (the host staging of this block is reused)
    }
    if (lds_need > size_t(48) << 10) {
        OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_step_kernel<V, false>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_need)));
        OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(chain_step_kernel<V, true>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds_need)));
    }
    bytes = (size_t(2 * K - 1) * size_t(bs) * size_t(bs)) * sizeof(V);
}

template <typename V>
bool CoarseSolver<V>::build_sine(const HostCsr &A, hipStream_t s) {
    int64_t gx = 0, gy = 0, gz = 0;
    double c[4];
    if (A.n_rows > 8192 || !detect_symmetric_star(A, gx, gy, gz, c) || gx > 32 || gy > 32 || gz > 32) return false;
    const double pi = 3.14159265358979323846264338327950288;
    const int64_t E = std::max(gx, std::max(gy, gz)) <= 16 ? 16 : 32;                 // row stride of the tables (zeros behind a row)
    std::vector<double> tab(size_t((gx + gy + gz) * E), 0.0), lam(size_t(A.n_rows));
    auto table = [&](double *t, int64_t m) {
        const long double f = std::sqrt(2.0L / (long double)(m + 1));
        for (int64_t i = 0; i < m; ++i)
            for (int64_t p = 0; p < m; ++p)
                t[i * E + p] = m == 1 ? 1.0 : double(f * sinl((long double)pi * (long double)((i + 1) * (p + 1)) / (long double)(m + 1)));
    };
    table(tab.data(), gx);
    table(tab.data() + gx * E, gy);
    table(tab.data() + (gx + gy) * E, gz);
    double smallest = 1e300, largest = 0.0;
    for (int64_t r = 0; r < A.n_rows; ++r) {
        const int64_t p = r % gx, q = (r / gx) % gy, t = r / (gx * gy);
        double l = c[0] + 2.0 * c[1] * std::cos(pi * double(p + 1) / double(gx + 1));
        if (gy > 1) l += 2.0 * c[2] * std::cos(pi * double(q + 1) / double(gy + 1));
        if (gz > 1) l += 2.0 * c[3] * std::cos(pi * double(t + 1) / double(gz + 1));
        lam[size_t(r)] = l;
        smallest = std::min(smallest, std::fabs(l));
        largest = std::max(largest, std::fabs(l));
    }
    if (!(smallest > 1e-13 * largest)) return false;     // (numerically) singular: let the factorisation say so
    P = 0;
    sx = int(gx); sy = int(gy); sz = int(gz);
    sine.alloc(tab.size());
    lambda.alloc(lam.size());
    sine.upload(tab.data(), tab.size(), s);
    lambda.upload(lam.data(), lam.size(), s);
    OMG_HIP(hipStreamSynchronize(s));
    w = gx * gy;
    bytes = (tab.size() + lam.size()) * sizeof(double);
    return true;
}

template <typename V>
void CoarseSolver<V>::solve(const V *b, V *x, hipStream_t s) const {
    if (n == 0) return;
    if (P == 0) {
        const int E = std::max(sx, std::max(sy, sz)) <= 16 ? 16 : 32;
        const size_t lds = (size_t(sx + 1) * size_t(sy) * size_t(sz) + 2 + size_t(sx + sy + sz) * size_t(E)) * sizeof(double);
        static const bool cube16 = [] { const char *e = experiment_env("OMG_SINE_CUBE16"); return !(e && e[0] == '0'); }();
        if (E == 16 && sx == 16 && sy == 16 && sz == 16 && cube16) {
            hipLaunchKernelGGL((sine_cube16_kernel<V>), dim3(1), dim3(SINE_THREADS), 0, s, b, x, sine.p, lambda.p);
        } else if (E == 16) {
            hipLaunchKernelGGL((sine_solve_kernel<V, 16>), dim3(1), dim3(SINE_THREADS), lds, s, b, x, sx, sy, sz, sine.p, lambda.p);
        } else {
            if (lds > size_t(64) * 1024)
                OMG_HIP(hipFuncSetAttribute(reinterpret_cast<const void *>(sine_solve_kernel<V, 32>), hipFuncAttributeMaxDynamicSharedMemorySize, int(lds)));
            hipLaunchKernelGGL((sine_solve_kernel<V, 32>), dim3(1), dim3(SINE_THREADS), lds, s, b, x, sx, sy, sz, sine.p, lambda.p);
        }
        OMG_HIP(hipGetLastError());
        return;
    }
    if (P == 1) {
        launch_dense_gemv<V>(inv.p, b, x, n, s);
        return;
    }
    if (P == -1) {
        for (int k = 0; k < K; ++k) {
            const int64_t r0 = int64_t(k) * bs, m = std::min(n, r0 + bs) - r0;
            hipLaunchKernelGGL((chain_step_kernel<V, false>), dim3(unsigned((m + CHAIN_ROWS - 1) / CHAIN_ROWS)), dim3(256), size_t(m) * sizeof(V), s,
                               dinv.p + size_t(k) * size_t(bs) * size_t(bs), m, r0, lo_ptr.p, lo_idx.p, lo_val.p, b, z.p, z.p, x, k == K - 1 ? 1 : 0);
        }
        for (int k = K - 2; k >= 0; --k) {
            const int64_t r0 = int64_t(k) * bs;
            hipLaunchKernelGGL((chain_step_kernel<V, true>), dim3(unsigned((bs + CHAIN_ROWS - 1) / CHAIN_ROWS)), dim3(256), size_t(bs) * sizeof(V), s,
                               dinv.p + size_t(k) * size_t(bs) * size_t(bs), bs, r0, up_ptr.p, up_idx.p, up_val.p, b, x, z.p, x, 0);
        }
        OMG_HIP(hipGetLastError());
        return;
    }
    hipLaunchKernelGGL((interior_solve_kernel<V>), dim3((unsigned)n_wg), dim3(256), 0, s, binv.p, binv_off.p, blk_off.p,
                       wg_blk.p, wg_row.p, perm.p, b, y.p);
    hipLaunchKernelGGL((separator_solve_kernel<V>), dim3((unsigned)((g + 3) / 4)), dim3(256), (size_t)(g) * sizeof(V), s,
                       sinv.p, g, n_int, gi_ptr.p, gi_idx.p, gi_val.p, perm.p, b, y.p, xg.p, x);
    hipLaunchKernelGGL((interior_correct_kernel<V>), dim3((unsigned)n_wg), dim3(256), (size_t)(2 * w) * sizeof(V), s, binv.p,
                       binv_off.p, blk_off.p, wg_blk.p, wg_row.p, ig_ptr.p, ig_idx.p, ig_val.p, xg.p, perm.p, y.p, w, x);
    OMG_HIP(hipGetLastError());
}

template struct CoarseSolver<double>;
template struct CoarseSolver<float>;

}  // namespace omg
