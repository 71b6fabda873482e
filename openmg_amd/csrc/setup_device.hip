// Device-side hierarchy setup: sparse products (Galerkin R A R^T, openmg/operators.py:184-186)
// and the aggregation restriction operator (openmg/operators.py:15-89).
//
// SpGEMM C = X Y is row-wise Gustavson with one thread per row of X.  The products of a row
// are generated in SciPy's order (k in stored order of X's row, then j in stored order of
// Y's row k) and accumulated into a per-row list kept sorted by column, so every C(i, j) is
// summed in exactly the order SciPy's csr_matmat uses and the output has sorted columns;
// entries whose sum is exactly zero are dropped, as SciPy drops them.
// The list lives in a global scratch slice sized by the row's upper bound sum_k nnz(Y_k).
// Setup runs once per hierarchy; it is latency- not bandwidth-bound and is not on the
// V-cycle's critical path.
#include <algorithm>
#include <memory>
#include <numeric>

#include <hipcub/hipcub.hpp>

#include "common.h"

using namespace omg;

struct omg_csr_result {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    DevBuf<int32_t> indptr, indices;
    DevBuf<double> data;
};

namespace {

struct DevMat {            // plain device CSR without row blocks
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    DevBuf<int32_t> indptr, indices;
    DevBuf<double> data;
};

void upload(DevMat &M, const omg_csr &A, hipStream_t s) {
    M.n_rows = A.n_rows; M.n_cols = A.n_cols; M.nnz = A.nnz;
    M.indptr.alloc(A.n_rows + 1);
    M.indices.alloc(std::max<int64_t>(A.nnz, 1));
    M.data.alloc(std::max<int64_t>(A.nnz, 1));
    M.indptr.upload(A.indptr, A.n_rows + 1, s);
    M.indices.upload(A.indices, A.nnz, s);
    M.data.upload(A.data, A.nnz, s);
}

__global__ void upper_bound_kernel(int64_t n, const int32_t *xp, const int32_t *xi,
                                   const int32_t *yp, int64_t *ub) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int64_t t = 0;
        for (int32_t p = xp[i]; p < xp[i + 1]; ++p) {
            const int32_t k = xi[p];
            t += yp[k + 1] - yp[k];
        }
        ub[i] = t;
    }
}

__global__ void gustavson_rows_kernel(int64_t n, const int32_t *xp, const int32_t *xi,
                                      const double *xv, const int32_t *yp, const int32_t *yi,
                                      const double *yv, const int64_t *off, int32_t *scols,
                                      double *svals, int32_t *row_nnz) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        int32_t *cols = scols + off[i];
        double *vals = svals + off[i];
        int cnt = 0;
        for (int32_t p = xp[i]; p < xp[i + 1]; ++p) {
            const int32_t k = xi[p];
            const double a = xv[p];
            for (int32_t q = yp[k]; q < yp[k + 1]; ++q) {
                const int32_t j = yi[q];
                const double prod = __dmul_rn(a, yv[q]);      // no FMA contraction: SciPy's order and rounding
                int pos = cnt;                                 // search from the back: stencil rows arrive nearly sorted
                while (pos > 0 && cols[pos - 1] > j) --pos;
                if (pos > 0 && cols[pos - 1] == j) {
                    vals[pos - 1] = __dadd_rn(vals[pos - 1], prod);
                } else {
                    for (int m = cnt; m > pos; --m) { cols[m] = cols[m - 1]; vals[m] = vals[m - 1]; }
                    cols[pos] = j;
                    vals[pos] = prod;
                    ++cnt;
                }
            }
        }
        // SciPy's csr_matmat keeps an accumulated entry only `if (sums[head] != 0)`: sums that
        // cancel to exactly 0.0 are not stored (they would change nnz, add couplings to the
        // colouring / level schedule and cost bytes)
        int kept = 0;
        for (int m = 0; m < cnt; ++m)
            if (vals[m] != 0.0) { cols[kept] = cols[m]; vals[kept] = vals[m]; ++kept; }
        row_nnz[i] = kept;
    }
}

__global__ void compact_rows_kernel(int64_t n, const int64_t *off, const int32_t *scols,
                                    const double *svals, const int32_t *cp, int32_t *ci, double *cv) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x) {
        const int32_t *cols = scols + off[i];
        const double *vals = svals + off[i];
        const int32_t dst = cp[i], len = cp[i + 1] - dst;
        for (int32_t m = 0; m < len; ++m) { ci[dst + m] = cols[m]; cv[dst + m] = vals[m]; }
    }
}

int grid1d(int64_t n) {
    int64_t g = (n + 255) / 256;
    if (g > 65536) g = 65536;
    if (g < 1) g = 1;
    return (int)g;
}

// ---- transpose on the device (R^T for the Galerkin product) --------------------------------
// Counting sort by column: counts (atomics: order-free), exclusive scan, scatter through per-column
// cursors, then every column's short segment is put in ascending row order — the order SciPy's
// csc view / the host transpose_csr give (entries with the same (row, column) tie-break by value
// bits: deterministic).
__global__ void count_columns_kernel(int64_t nnz, const int32_t *ix, int32_t *cnt) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x)
        atomicAdd(&cnt[ix[p]], 1);
}

__global__ void scatter_transpose_kernel(int64_t n_rows, const int32_t *ip, const int32_t *ix, const double *dv,
                                         const int32_t *tp, int32_t *cursor, int32_t *ti, double *tv) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
        for (int32_t p = ip[r]; p < ip[r + 1]; ++p) {
            const int32_t c = ix[p];
            const int32_t pos = tp[c] + atomicAdd(&cursor[c], 1);
            ti[pos] = int32_t(r);
            tv[pos] = dv[p];
        }
}

__global__ void sort_segments_kernel(int64_t n_seg, const int32_t *tp, int32_t *ti, double *tv) {
    for (int64_t c = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; c < n_seg; c += (int64_t)gridDim.x * blockDim.x) {
        const int32_t b = tp[c], e = tp[c + 1];
        for (int32_t i = b + 1; i < e; ++i) {                      // insertion sort: segments are a few entries long
            const int32_t ki = ti[i];
            const double kv = tv[i];
            int32_t j = i - 1;
            while (j >= b && (ti[j] > ki || (ti[j] == ki && __double_as_longlong(tv[j]) > __double_as_longlong(kv)))) {
                ti[j + 1] = ti[j];
                tv[j + 1] = tv[j];
                --j;
            }
            ti[j + 1] = ki;
            tv[j + 1] = kv;
        }
    }
}

void device_transpose(const DevMat &X, DevMat &T, hipStream_t s) {
    T.n_rows = X.n_cols; T.n_cols = X.n_rows; T.nnz = X.nnz;
    const int64_t nc = X.n_cols;
    T.indptr.alloc(nc + 2);
    T.indices.alloc(std::max<int64_t>(X.nnz, 1));
    T.data.alloc(std::max<int64_t>(X.nnz, 1));
    DevBuf<int32_t> cnt(nc + 2);
    OMG_HIP(hipMemsetAsync(cnt.p, 0, size_t(nc + 2) * sizeof(int32_t), s));
    hipLaunchKernelGGL(count_columns_kernel, dim3(grid1d(X.nnz)), dim3(256), 0, s, X.nnz, X.indices.p, cnt.p);
    size_t bytes = 0;
    OMG_HIP(hipcub::DeviceScan::ExclusiveSum(nullptr, bytes, cnt.p, T.indptr.p, int(nc + 1), s));
    DevBuf<unsigned char> tmp(bytes + 16);
    OMG_HIP(hipcub::DeviceScan::ExclusiveSum(tmp.p, bytes, cnt.p, T.indptr.p, int(nc + 1), s));
    OMG_HIP(hipMemsetAsync(cnt.p, 0, size_t(nc + 2) * sizeof(int32_t), s));
    hipLaunchKernelGGL(scatter_transpose_kernel, dim3(grid1d(X.n_rows)), dim3(256), 0, s, X.n_rows, X.indptr.p,
                       X.indices.p, X.data.p, T.indptr.p, cnt.p, T.indices.p, T.data.p);
    hipLaunchKernelGGL(sort_segments_kernel, dim3(grid1d(nc)), dim3(256), 0, s, nc, T.indptr.p, T.indices.p, T.data.p);
    OMG_HIP(hipGetLastError());
    OMG_HIP(hipStreamSynchronize(s));       // cnt / tmp die here
}

// C = X Y on the device.
void spgemm(const DevMat &X, const DevMat &Y, omg_csr_result &C, hipStream_t s) {
    OMG_REQUIRE(X.n_cols == Y.n_rows, "spgemm: inner dimensions differ");
    const int64_t n = X.n_rows;
    C.n_rows = n;
    C.n_cols = Y.n_cols;
    DevBuf<int64_t> ub(std::max<int64_t>(n, 1));
    hipLaunchKernelGGL(upper_bound_kernel, dim3(grid1d(n)), dim3(256), 0, s, n, X.indptr.p,
                       X.indices.p, Y.indptr.p, ub.p);
    std::vector<int64_t> h_ub(n), h_off(n + 1, 0);
    ub.download(h_ub.data(), n, s);
    OMG_HIP(hipStreamSynchronize(s));
    for (int64_t i = 0; i < n; ++i) h_off[i + 1] = h_off[i] + h_ub[i];
    const int64_t total = h_off[n];
    DevBuf<int64_t> off(n + 1);
    off.upload(h_off.data(), n + 1, s);
    DevBuf<int32_t> scols(std::max<int64_t>(total, 1)), row_nnz(std::max<int64_t>(n, 1));
    DevBuf<double> svals(std::max<int64_t>(total, 1));
    hipLaunchKernelGGL(gustavson_rows_kernel, dim3(grid1d(n)), dim3(256), 0, s, n, X.indptr.p,
                       X.indices.p, X.data.p, Y.indptr.p, Y.indices.p, Y.data.p, off.p, scols.p,
                       svals.p, row_nnz.p);
    std::vector<int32_t> h_nnz(n), h_cp(n + 1, 0);
    row_nnz.download(h_nnz.data(), n, s);
    OMG_HIP(hipStreamSynchronize(s));
    int64_t acc = 0;
    for (int64_t i = 0; i < n; ++i) {
        acc += h_nnz[i];
        OMG_REQUIRE(acc < INT32_MAX, "spgemm: result exceeds int32 nnz");
        h_cp[i + 1] = (int32_t)acc;
    }
    C.nnz = acc;
    C.indptr.alloc(n + 1);
    C.indptr.upload(h_cp.data(), n + 1, s);
    C.indices.alloc(std::max<int64_t>(acc, 1));
    C.data.alloc(std::max<int64_t>(acc, 1));
    hipLaunchKernelGGL(compact_rows_kernel, dim3(grid1d(n)), dim3(256), 0, s, n, off.p, scols.p,
                       svals.p, C.indptr.p, C.indices.p, C.data.p);
    OMG_HIP(hipGetLastError());
    OMG_HIP(hipStreamSynchronize(s));
}

void as_devmat(omg_csr_result &&R, DevMat &M) {
    M.n_rows = R.n_rows; M.n_cols = R.n_cols; M.nnz = R.nnz;
    M.indptr = std::move(R.indptr);
    M.indices = std::move(R.indices);
    M.data = std::move(R.data);
}

// openmg/operators.py:73-84.  Row r <-> the r-th coarse cell in C-order over ceil(s/2)
// extents; its first fine column is the C-order index of the cell's even corner; the other
// members sit at +1, +NX, +NX+1, +NX*NY, ... with NX = shape[0], NY = shape[1] (Q6).
__global__ void restriction_kernel(int dim, int64_t s0, int64_t s1, int64_t s2, int64_t rows,
                                   int32_t *indptr, int32_t *indices, double *data) {
    const int per = 1 << dim;
    const int64_t c0 = (s0 + 1) / 2, c1 = dim >= 2 ? (s1 + 1) / 2 : 1, c2 = dim >= 3 ? (s2 + 1) / 2 : 1;
    (void)c0;
    const double w = 1.0 / (double)per;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= rows;
         r += (int64_t)gridDim.x * blockDim.x) {
        indptr[r] = (int32_t)(r * per);
        if (r == rows) break;
        int64_t first;
        if (dim == 1) first = 2 * r;
        else if (dim == 2) { const int64_t i0 = r / c1, i1 = r % c1; first = (2 * i0) * s1 + 2 * i1; }
        else { const int64_t i0 = r / (c1 * c2), rem = r % (c1 * c2), i1 = rem / c2, i2 = rem % c2;
               first = ((2 * i0) * s1 + 2 * i1) * s2 + 2 * i2; }
        const int64_t NX = s0, NXY = s0 * s1;
        int64_t offs[8] = {0, 1, NX, NX + 1, NXY, NXY + 1, NXY + NX, NXY + NX + 1};
        // ascending column order inside the row needs offs sorted; they are whenever NX >= 2.
        for (int m = 0; m < per; ++m) {
            indices[r * per + m] = (int32_t)(first + offs[m]);
            data[r * per + m] = w;
        }
    }
}

template <typename F>
int guarded(F &&f) {
    try {
        f();
        return OMG_OK;
    } catch (const Error &e) {
        set_last_error(e.what());
        return e.code;
    } catch (const std::bad_alloc &) {
        set_last_error("host allocation failed");
        return OMG_ERR_ALLOC;
    } catch (const std::exception &e) {
        set_last_error(e.what());
        return OMG_ERR_INVALID;
    }
}

struct Stream {
    hipStream_t s = nullptr;
    Stream() { require_device(); OMG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~Stream() { if (s) (void)hipStreamDestroy(s); }
};

}  // namespace

extern "C" {

int omg_spgemm(const omg_csr *X, const omg_csr *Y, omg_csr_result **out, int64_t *n_rows,
               int64_t *n_cols, int64_t *nnz) {
    return guarded([&] {
        OMG_REQUIRE(X && Y && out && n_rows && n_cols && nnz, "null argument");
        validate_csr(*X, "X");
        validate_csr(*Y, "Y");
        Stream st;
        DevMat dX, dY;
        upload(dX, *X, st.s);
        upload(dY, *Y, st.s);
        std::unique_ptr<omg_csr_result> C(new omg_csr_result);
        spgemm(dX, dY, *C, st.s);
        *n_rows = C->n_rows; *n_cols = C->n_cols; *nnz = C->nnz;
        *out = C.release();
    });
}

int omg_rap(const omg_csr *R, const omg_csr *A, omg_csr_result **out, int64_t *n_rows,
            int64_t *n_cols, int64_t *nnz) {
    return guarded([&] {
        OMG_REQUIRE(R && A && out && n_rows && n_cols && nnz, "null argument");
        validate_csr(*R, "R");
        validate_csr(*A, "A");
        OMG_REQUIRE(A->n_rows == A->n_cols && R->n_cols == A->n_rows, "rap: shapes do not chain");
        Stream st;
        DevMat dR, dA, dRt, dRA;
        { SetupTimer tm("rap: upload R, A"); upload(dR, *R, st.s); upload(dA, *A, st.s); OMG_HIP(hipStreamSynchronize(st.s)); }
        { SetupTimer tm("rap: R^T on the device"); device_transpose(dR, dRt, st.s); }   // index shuffle only
        omg_csr_result RA;
        SetupTimer tm("rap: two sparse products");
        spgemm(dR, dA, RA, st.s);                          // (R A)        operators.py:185
        as_devmat(std::move(RA), dRA);
        std::unique_ptr<omg_csr_result> C(new omg_csr_result);
        spgemm(dRA, dRt, *C, st.s);                        // (R A) R^T    operators.py:184-186
        *n_rows = C->n_rows; *n_cols = C->n_cols; *nnz = C->nnz;
        *out = C.release();
    });
}

int omg_csr_result_fetch(omg_csr_result *res, int32_t *indptr, int32_t *indices, double *data) {
    return guarded([&] {
        OMG_REQUIRE(res && indptr, "null argument");
        OMG_REQUIRE(res->nnz == 0 || (indices && data), "null output array");
        res->indptr.download(indptr, res->n_rows + 1, nullptr);
        res->indices.download(indices, res->nnz, nullptr);
        res->data.download(data, res->nnz, nullptr);
        OMG_HIP(hipStreamSynchronize(nullptr));
        delete res;
    });
}

int omg_csr_result_free(omg_csr_result *res) {
    delete res;
    return OMG_OK;
}

int omg_restriction(int dim, const int64_t *shape, int32_t *indptr, int32_t *indices, double *data,
                    int64_t *n_rows, int64_t *nnz) {
    return guarded([&] {
        OMG_REQUIRE(shape && indptr && indices && data && n_rows && nnz, "null argument");
        OMG_REQUIRE(dim >= 1 && dim <= 3, "restriction(): Greater than 3 dimensions is not implemented");
        int64_t N = 1, cells = 1;
        for (int d = 0; d < dim; ++d) {
            OMG_REQUIRE(shape[d] >= 1, "non-positive extent");
            N *= shape[d];
            cells *= (shape[d] + 1) / 2;
        }
        OMG_REQUIRE(N < INT32_MAX, "int32 index range exceeded");
        const int per = 1 << dim;
        const int64_t n = N / per;
        const int64_t rows = std::min(n, cells);               // zip() truncation, operators.py:74
        Stream st;
        DevBuf<int32_t> dp(n + 1), di(std::max<int64_t>(rows * per, 1));
        DevBuf<double> dv(std::max<int64_t>(rows * per, 1));
        hipLaunchKernelGGL(restriction_kernel, dim3(grid1d(rows + 1)), dim3(256), 0, st.s, dim,
                           shape[0], dim >= 2 ? shape[1] : 1, dim >= 3 ? shape[2] : 1, rows, dp.p,
                           di.p, dv.p);
        OMG_HIP(hipGetLastError());
        dp.download(indptr, rows + 1, st.s);
        di.download(indices, rows * per, st.s);
        dv.download(data, rows * per, st.s);
        OMG_HIP(hipStreamSynchronize(st.s));
        for (int64_t r = rows + 1; r <= n; ++r) indptr[r] = (int32_t)(rows * per);   // trailing empty rows
        *n_rows = n;
        *nnz = rows * per;
    });
}

}  // extern "C"
