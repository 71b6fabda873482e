// Coarsest-level direct solve.  The reference calls SuperLU on every cycle
// (openmg/solvers.py:23); here the coarsest operator is inverted ONCE at setup on the
// device (Gauss-Jordan with partial pivoting on the augmented matrix [A | I], fp64) and
// every cycle's coarse solve is one dense mat-vec with the stored inverse — a pure HBM
// stream of 8 n^2 bytes (n = 4096 for a 256^3 problem with 5 grids: 134 MB, ~25 us).
#include <cmath>

#include "common.h"

namespace omg {

namespace {

// W is row-major n x (2n): [A | I].
__global__ void fill_aug_kernel(double *W, int64_t n) {
    const int64_t total = n * 2 * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (2 * n), c = i % (2 * n);
        W[i] = (c == n + r) ? 1.0 : 0.0;
    }
}

__global__ void csr_scatter_dense_kernel(const int32_t *indptr, const int32_t *indices,
                                         const double *data, int64_t n_rows, double *W,
                                         int64_t ld) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows;
         r += (int64_t)gridDim.x * blockDim.x)
        for (int32_t p = indptr[r]; p < indptr[r + 1]; ++p)
            atomicAdd(&W[r * ld + indices[p]], data[p]);   // duplicates add, like A[i, j] in SciPy
}

// The same for a sub-block: entry (r, c) goes to W[rmap[r] * ld + cmap[c]] when both maps are >= 0.
__global__ void csr_scatter_sub_kernel(const int32_t *indptr, const int32_t *indices, const double *data,
                                       int64_t n_rows, const int32_t *rmap, const int32_t *cmap, double *W,
                                       int64_t ld) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows;
         r += (int64_t)gridDim.x * blockDim.x) {
        const int32_t lr = rmap[r];
        if (lr < 0) continue;
        for (int32_t p = indptr[r]; p < indptr[r + 1]; ++p) {
            const int32_t lc = cmap[indices[p]];
            if (lc >= 0) atomicAdd(&W[int64_t(lr) * ld + lc], data[p]);
        }
    }
}

// argmax_{i >= k} |W[i, k]|  ->  piv[0] = row, flag set when the column is exactly zero.
__global__ __launch_bounds__(1024) void pivot_kernel(const double *W, int64_t n, int64_t ld,
                                                     int64_t k, int *piv, int *singular) {
    __shared__ double s_v[1024];
    __shared__ int s_i[1024];
    double best = -1.0;
    int bi = (int)k;
    for (int64_t i = k + threadIdx.x; i < n; i += 1024) {
        const double v = fabs(W[i * ld + k]);
        if (v > best) { best = v; bi = (int)i; }
    }
    s_v[threadIdx.x] = best;
    s_i[threadIdx.x] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (threadIdx.x < off) {
            const double o = s_v[threadIdx.x + off];
            const int oi = s_i[threadIdx.x + off];
            if (o > s_v[threadIdx.x] || (o == s_v[threadIdx.x] && oi < s_i[threadIdx.x])) {
                s_v[threadIdx.x] = o;
                s_i[threadIdx.x] = oi;
            }
        }
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        piv[0] = s_i[0];
        if (!(s_v[0] > 0.0) || !isfinite(s_v[0])) *singular = 1;
    }
}

// colk[i] = column k as it will look AFTER rows k and piv are swapped.
__global__ void column_kernel(const double *W, int64_t n, int64_t ld, int64_t k, const int *piv,
                              double *colk) {
    const int64_t p = piv[0];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = (i == k) ? p : (i == p ? k : i);
        colk[i] = W[src * ld + k];
    }
}

// Swap rows k and piv over all columns and divide the new row k by the pivot colk[k].
__global__ void swap_scale_kernel(double *W, int64_t ld, int64_t k, const int *piv,
                                  const double *colk) {
    const int64_t p = piv[0];
    const double pv = colk[k];
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < ld;
         c += (int64_t)gridDim.x * blockDim.x) {
        const double a = W[k * ld + c], b = W[p * ld + c];
        W[p * ld + c] = a;            // p == k: overwritten by the next line
        W[k * ld + c] = b / pv;
    }
}

// W[i, c] -= colk[i] * W[k, c] for every row i != k.  Columns where the pivot row is zero
// (left of k, and the not-yet-touched identity columns on the right) exit at once, so the
// cost per step is ~ (n + 1) columns x n rows.
__global__ __launch_bounds__(256) void eliminate_kernel(double *W, int64_t n, int64_t ld,
                                                        int64_t k, const double *colk) {
    const int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x;
    if (c >= ld) return;
    const double pk = W[k * ld + c];
    if (pk == 0.0) return;
    for (int64_t i = blockIdx.y; i < n; i += gridDim.y) {
        if (i == k) continue;
        const double f = colk[i];
        if (f != 0.0) W[i * ld + c] -= f * pk;
    }
}

__global__ void extract_inverse_kernel(const double *W, int64_t n, double *Minv) {
    const int64_t total = n * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / n, c = i % n;
        Minv[i] = W[r * 2 * n + n + c];
    }
}

int grid1d(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 4096) g = 4096;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- blocked Gauss-Jordan (round 5) -----------------------------------------------------------------------------------
// The column-by-column elimination above sweeps the whole augmented matrix once per PIVOT: 4 n launches and O(n^3) bytes of
// HBM traffic — 190 ms for the 4096 unknowns of a 16^3 coarsest level with per-row coefficients (BASELINE configs[4]; the
// constant-coefficient headline takes the sine transform instead).  Here 64 pivots at a time: the 64 x 64 diagonal block is
// inverted in LDS (partial pivoting INSIDE the block: just the method for its inverse), row block K becomes D^-1 R_K, and
// every other row block R_i <- R_i - A_iK R_K as 64 x 64 x 64 tile products — one sweep of the active columns (left of
// the pivots the matrix is already the identity, right of them the inverse's columns are still untouched: n columns in
// all) per 64 pivots, 3 launches per block.  No pivoting ACROSS blocks: exact for the symmetric positive definite /
// diagonally dominant operators a Galerkin hierarchy ends in; the caller checks the result against the operator and falls
// back to the pivoted elimination where it is not an inverse.
constexpr int GJB = 64;

// D^-1 of the diagonal block [k0, k0 + nb) by Gauss-Jordan with partial pivoting in LDS -> dinv (GJB x GJB, row-major)
__global__ __launch_bounds__(256) void gjb_diag_kernel(const double *W, int64_t ld, int64_t k0, int nb, double *dinv, int *singular) {
    __shared__ double D[GJB][2 * GJB + 1];
    __shared__ int s_piv;
    __shared__ double s_col[GJB];
    const int t = int(threadIdx.x);
    for (int q = t; q < GJB * 2 * GJB; q += 256) {
        const int r = q / (2 * GJB), c = q % (2 * GJB);
        double v;
        if (c < GJB) v = (r < nb && c < nb) ? W[(k0 + r) * ld + k0 + c] : (r == c ? 1.0 : 0.0);
        else v = (c - GJB == r) ? 1.0 : 0.0;
        D[r][c] = v;
    }
    __syncthreads();
    for (int k = 0; k < nb; ++k) {
        if (t < 64) {
            // lanes = rows: the largest magnitude of column k among rows >= k (ties: the smallest row)
            double v = (t >= k && t < nb) ? fabs(D[t][k]) : -1.0;
            int idx = t;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) {
                const double ov = __shfl_down(v, off, 64);
                const int oi = __shfl_down(idx, off, 64);
                if (ov > v || (ov == v && oi < idx)) { v = ov; idx = oi; }
            }
            if (t == 0) {
                s_piv = idx;
                if (!(v > 0.0) || !isfinite(v)) *singular = 1;
            }
        }
        __syncthreads();
        const int p = s_piv;
        if (p != k && t < 2 * GJB) {
            const double a = D[k][t], b2 = D[p][t];
            D[k][t] = b2;
            D[p][t] = a;
        }
        __syncthreads();
        const double pv = D[k][k];
        if (t < GJB) s_col[t] = D[t][k];
        __syncthreads();
        if (t < 2 * GJB) D[k][t] = D[k][t] / pv;
        __syncthreads();
        for (int q = t; q < GJB * 2 * GJB; q += 256) {
            const int r = q / (2 * GJB), c = q % (2 * GJB);
            if (r != k && r < nb) D[r][c] -= s_col[r] * D[k][c];
        }
        __syncthreads();
    }
    for (int q = t; q < GJB * GJB; q += 256) dinv[q] = D[q / GJB][GJB + q % GJB];
}

// column tile `ct` of the active columns of block step k0 .. k1: left of the matrix [k1, n), then the inverse's [n, n + k1)
__device__ __forceinline__ int64_t gjb_col0(int64_t ct, int64_t n, int64_t k1, int64_t &c_end) {
    const int64_t left_tiles = (n - k1 + GJB - 1) / GJB;
    if (ct < left_tiles) { c_end = n; return k1 + ct * GJB; }
    c_end = n + k1;
    return n + (ct - left_tiles) * GJB;
}

// blockIdx.x < col_tiles: row block K <- D^-1 (row block K) on that column tile; the others: the pivot columns of 64 rows
// into `panel` (n x GJB), which the elimination reads while the matrix's own copy is being overwritten
__global__ __launch_bounds__(256) void gjb_rowblock_kernel(double *W, int64_t n, int64_t ld, int64_t k0, int nb, const double *dinv, double *panel,
                                                           int col_tiles) {
    __shared__ double Dv[GJB][GJB + 1];
    __shared__ double Rk[GJB][GJB + 1];
    const int t = int(threadIdx.x);
    if (int(blockIdx.x) >= col_tiles) {
        const int64_t r0 = int64_t(int(blockIdx.x) - col_tiles) * GJB;
        for (int q = t; q < GJB * GJB; q += 256) {
            const int64_t r = r0 + q / GJB;
            const int c = q % GJB;
            if (r < n) panel[r * GJB + c] = c < nb ? W[r * ld + k0 + c] : 0.0;
        }
        return;
    }
    int64_t c_end;
    const int64_t c0 = gjb_col0(blockIdx.x, n, k0 + nb, c_end);
    for (int q = t; q < GJB * GJB; q += 256) {
        const int r = q / GJB, c = q % GJB;
        Dv[r][c] = dinv[q];
        Rk[r][c] = (r < nb && c0 + c < c_end) ? W[(k0 + r) * ld + c0 + c] : 0.0;
    }
    __syncthreads();
    const int ty = t / 16, tx = t % 16;
    double acc[4][4] = {{0.0}};
    for (int k = 0; k < nb; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = Dv[ty * 4 + r][k];
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = Rk[k][tx * 4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fma(av[r], bv[c], acc[r][c]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int rr = ty * 4 + r;
            const int64_t cc = c0 + tx * 4 + c;
            if (rr < nb && cc < c_end) W[(k0 + rr) * ld + cc] = acc[r][c];
        }
}

// R_i <- R_i - A_iK R_K on a 64 x 64 tile (row tile blockIdx.y, skipping block K; column tile blockIdx.x of the active columns)
__global__ __launch_bounds__(256) void gjb_eliminate_kernel(double *W, int64_t n, int64_t ld, int64_t k0, int nb, const double *panel) {
    const int64_t r0 = int64_t(blockIdx.y) * GJB;
    if (r0 == k0) return;
    __shared__ double Ap[GJB][GJB + 1];
    __shared__ double Rk[GJB][GJB + 1];
    const int t = int(threadIdx.x);
    int64_t c_end;
    const int64_t c0 = gjb_col0(blockIdx.x, n, k0 + nb, c_end);
    for (int q = t; q < GJB * GJB; q += 256) {
        const int r = q / GJB, c = q % GJB;
        Ap[r][c] = (r0 + r < n) ? panel[(r0 + r) * GJB + c] : 0.0;
        Rk[r][c] = (r < nb && c0 + c < c_end) ? W[(k0 + r) * ld + c0 + c] : 0.0;
    }
    __syncthreads();
    const int ty = t / 16, tx = t % 16;
    double acc[4][4];
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t rr = r0 + ty * 4 + r, cc = c0 + tx * 4 + c;
            acc[r][c] = (rr < n && cc < c_end) ? W[rr * ld + cc] : 0.0;
        }
    for (int k = 0; k < nb; ++k) {
        double av[4], bv[4];
#pragma unroll
        for (int r = 0; r < 4; ++r) av[r] = Ap[ty * 4 + r][k];
#pragma unroll
        for (int c = 0; c < 4; ++c) bv[c] = Rk[k][tx * 4 + c];
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int c = 0; c < 4; ++c) acc[r][c] = fma(-av[r], bv[c], acc[r][c]);
    }
#pragma unroll
    for (int r = 0; r < 4; ++r)
#pragma unroll
        for (int c = 0; c < 4; ++c) {
            const int64_t rr = r0 + ty * 4 + r, cc = c0 + tx * 4 + c;
            if (rr < n && cc < c_end) W[rr * ld + cc] = acc[r][c];
        }
}

// || M (Minv v) - v ||_inf / || v ||_inf for one fixed vector: is Minv an inverse of M?  (M: n x n row-major)
__global__ __launch_bounds__(256) void gjb_matvec_kernel(const double *M, int64_t n, int64_t ld, const double *x, double *y) {
    __shared__ double s_red[256];
    const int64_t r = blockIdx.x;
    double acc = 0.0;
    for (int64_t c = threadIdx.x; c < n; c += 256) acc = fma(M[r * ld + c], x[c], acc);
    s_red[threadIdx.x] = acc;
    __syncthreads();
    for (int off = 128; off > 0; off >>= 1) {
        if (int(threadIdx.x) < off) s_red[threadIdx.x] += s_red[threadIdx.x + off];
        __syncthreads();
    }
    if (threadIdx.x == 0) y[r] = s_red[0];
}

// true: Minv holds the inverse.  W = [M | I] is destroyed either way; `keep` (n x n) must hold a copy of M.
bool blocked_inverse(double *W, int64_t n, const double *keep, double *Minv, hipStream_t s) {
    const int64_t ld = 2 * n;
    DevBuf<double> dinv(size_t(GJB) * GJB), panel(size_t(n) * GJB);
    DevBuf<int> flag(1);
    flag.zero(s);
    const int64_t row_tiles = (n + GJB - 1) / GJB;
    for (int64_t k0 = 0; k0 < n; k0 += GJB) {
        const int nb = int(std::min<int64_t>(GJB, n - k0));
        const int64_t k1 = k0 + nb;
        const int col_tiles = int((n - k1 + GJB - 1) / GJB + (k1 + GJB - 1) / GJB);
        hipLaunchKernelGGL(gjb_diag_kernel, dim3(1), dim3(256), 0, s, W, ld, k0, nb, dinv.p, flag.p);
        hipLaunchKernelGGL(gjb_rowblock_kernel, dim3(unsigned(col_tiles + row_tiles)), dim3(256), 0, s, W, n, ld, k0, nb, dinv.p, panel.p, col_tiles);
        hipLaunchKernelGGL(gjb_eliminate_kernel, dim3(unsigned(col_tiles), unsigned(row_tiles)), dim3(256), 0, s, W, n, ld, k0, nb, panel.p);
    }
    OMG_HIP(hipGetLastError());
    hipLaunchKernelGGL(extract_inverse_kernel, dim3(grid1d(n * n)), dim3(256), 0, s, W, n, Minv);
    // the check: v = (1, -1/2, 1/3, ...): M (Minv v) against v
    std::vector<double> v((size_t)(n));
    for (int64_t i = 0; i < n; ++i) v[size_t(i)] = ((i & 1) ? -1.0 : 1.0) / double(1 + i % 7);
    DevBuf<double> dv((size_t)(n)), dy((size_t)(n)), dz((size_t)(n));
    dv.upload(v.data(), size_t(n), s);
    hipLaunchKernelGGL(gjb_matvec_kernel, dim3(unsigned(n)), dim3(256), 0, s, Minv, n, n, dv.p, dy.p);
    hipLaunchKernelGGL(gjb_matvec_kernel, dim3(unsigned(n)), dim3(256), 0, s, keep, n, n, dy.p, dz.p);
    std::vector<double> z((size_t)(n));
    int bad = 0;
    dz.download(z.data(), size_t(n), s);
    OMG_HIP(hipMemcpyAsync(&bad, flag.p, sizeof(int), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    if (bad) return false;
    double err = 0.0;
    for (int64_t i = 0; i < n; ++i) {
        const double d = z[size_t(i)] - v[size_t(i)];
        if (!(d == d)) return false;
        err = std::max(err, std::fabs(d));
    }
    return err <= 1e-9;
}

__global__ void copy_block_kernel(const double *W, int64_t n, int64_t ld, double *out) {
    const int64_t total = n * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) out[i] = W[(i / n) * ld + i % n];
}
__global__ void restore_aug_kernel(const double *keep, int64_t n, double *W) {
    const int64_t total = n * 2 * n;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int64_t r = i / (2 * n), c = i % (2 * n);
        W[i] = c < n ? keep[r * n + c] : (c - n == r ? 1.0 : 0.0);
    }
}

}  // namespace

// Gauss-Jordan with partial pivoting on a prepared augmented matrix W = [M | I] (row-major
// n x 2n, destroyed); the inverse goes to Minv (n*n doubles).  ~4 n dependent launches — or, for n >= 256, the blocked
// form above first (OMG_DENSE_BLOCKED=0: never), checked against M.
void gauss_jordan_inverse(double *W, int64_t n, double *Minv, hipStream_t s) {
    if (n == 0) return;
    const int64_t ld = 2 * n;
    {
        static const bool blocked = [] { const char *e = getenv("OMG_DENSE_BLOCKED"); return !(e && e[0] == '0'); }();
        if (blocked && n >= 256) {
            DevBuf<double> keep((size_t)(n) * (size_t)(n));
            hipLaunchKernelGGL(copy_block_kernel, dim3(grid1d(n * n)), dim3(256), 0, s, W, n, ld, keep.p);
            if (blocked_inverse(W, n, keep.p, Minv, s)) return;
            hipLaunchKernelGGL(restore_aug_kernel, dim3(grid1d(n * ld)), dim3(256), 0, s, keep.p, n, W);
            OMG_HIP(hipGetLastError());
        }
    }
    DevBuf<double> colk(n);
    DevBuf<int> piv(2);
    piv.zero(s);
    int *singular = piv.p + 1;
    const dim3 egrid((unsigned)((ld + 255) / 256), (unsigned)std::min<int64_t>(n, 64));
    for (int64_t k = 0; k < n; ++k) {
        hipLaunchKernelGGL(pivot_kernel, dim3(1), dim3(1024), 0, s, W, n, ld, k, piv.p, singular);
        hipLaunchKernelGGL(column_kernel, dim3(grid1d(n)), dim3(256), 0, s, W, n, ld, k, piv.p, colk.p);
        hipLaunchKernelGGL(swap_scale_kernel, dim3(grid1d(ld)), dim3(256), 0, s, W, ld, k, piv.p, colk.p);
        hipLaunchKernelGGL(eliminate_kernel, egrid, dim3(256), 0, s, W, n, ld, k, colk.p);
    }
    OMG_HIP(hipGetLastError());
    hipLaunchKernelGGL(extract_inverse_kernel, dim3(grid1d(n * n)), dim3(256), 0, s, W, n, Minv);
    int flag = 0;
    OMG_HIP(hipMemcpyAsync(&flag, singular, sizeof(int), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    if (flag) throw Error(OMG_ERR_SINGULAR, "coarsest operator is singular to working precision");
}

// W = [sub-block of A | I]: rows / columns of the plain device CSR (indptr, indices, data) whose
// maps are >= 0 land at (rmap[r], cmap[c]); m = size of the sub-block.
void fill_augmented_from_csr(const int32_t *indptr, const int32_t *indices, const double *data, int64_t n_rows,
                             const int32_t *rmap, const int32_t *cmap, int64_t m, double *W, hipStream_t s) {
    if (m == 0) return;
    hipLaunchKernelGGL(fill_aug_kernel, dim3(grid1d(m * 2 * m)), dim3(256), 0, s, W, m);
    hipLaunchKernelGGL(csr_scatter_sub_kernel, dim3(grid1d(n_rows)), dim3(256), 0, s, indptr, indices, data, n_rows,
                       rmap, cmap, W, 2 * m);
    OMG_HIP(hipGetLastError());
}

// Build the dense inverse of the (square) device CSR matrix A into Minv (n*n doubles).
void dense_inverse_from_csr(const DevCsr &A, double *Minv, hipStream_t s) {
    const int64_t n = A.n_rows;
    OMG_REQUIRE(A.n_rows == A.n_cols, "coarse operator must be square");
    if (n == 0) return;
    if (n > 16384)
        throw Error(OMG_ERR_UNSUPPORTED,
                    "coarsest level has " + std::to_string(n) +
                        " unknowns; the dense direct solve is limited to 16384 — use more gridLevels");
    const int64_t ld = 2 * n;
    DevBuf<double> W(size_t(n) * size_t(ld));
    hipLaunchKernelGGL(fill_aug_kernel, dim3(grid1d(n * ld)), dim3(256), 0, s, W.p, n);
    hipLaunchKernelGGL(csr_scatter_dense_kernel, dim3(grid1d(n)), dim3(256), 0, s, A.indptr.p,
                       A.indices.p, A.data.p, n, W.p, ld);
    gauss_jordan_inverse(W.p, n, Minv, s);
}

}  // namespace omg
