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

}  // namespace

// Gauss-Jordan with partial pivoting on a prepared augmented matrix W = [M | I] (row-major
// n x 2n, destroyed); the inverse goes to Minv (n*n doubles).  ~4 n dependent launches.
void gauss_jordan_inverse(double *W, int64_t n, double *Minv, hipStream_t s) {
    if (n == 0) return;
    const int64_t ld = 2 * n;
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
