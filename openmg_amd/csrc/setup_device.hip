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

#include "common.h"

using namespace omg;

struct omg_csr_result {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    DevBuf<int32_t> indptr, indices;
    DevBuf<double> data;
};

namespace {

typedef DevCsrPlain DevMat;   // plain device CSR without row blocks (common.h)

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

// indptr[0..n] = exclusive scan of len[0..n) on the device (below); returns the total
int64_t device_row_pointers(int64_t n, const int32_t *len, int32_t *indptr, hipStream_t s);

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
    (void)device_row_pointers(nc, cnt.p, T.indptr.p, s);          // (own wave-scan kernels; no library scan)
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
    C.indptr.alloc(n + 1);
    const int64_t acc = device_row_pointers(n, row_nnz.p, C.indptr.p, s);
    C.nnz = acc;
    C.indices.alloc(std::max<int64_t>(acc, 1));
    C.data.alloc(std::max<int64_t>(acc, 1));
    hipLaunchKernelGGL(compact_rows_kernel, dim3(grid1d(n)), dim3(256), 0, s, n, off.p, scols.p,
                       svals.p, C.indptr.p, C.indices.p, C.data.p);
    OMG_HIP(hipGetLastError());
    OMG_HIP(hipStreamSynchronize(s));
}

// ---- row pointers: exclusive scan of the row lengths on the device -------------------------------------
// Three small launches (a workgroup's total, the totals' scan by one workgroup, the final offsets); the
// lengths of a workgroup's 1024 rows are scanned by its waves (shuffle scan) and its 16 wave totals in LDS.
constexpr int SCAN_WG = 1024;
__device__ __forceinline__ int wave_inclusive_scan(int v) {
    const int lane = int(threadIdx.x) & 63;
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int t = __shfl_up(v, off, 64);
        if (lane >= off) v += t;
    }
    return v;
}
// out[i] = base + exclusive scan of in[] inside the workgroup's 1024 entries; total[wg] = their sum
__global__ __launch_bounds__(SCAN_WG) void scan_local_kernel(int64_t n, const int32_t *in, int32_t *out, int64_t *total) {
    __shared__ int s_wave[SCAN_WG / 64];
    const int64_t i = int64_t(blockIdx.x) * SCAN_WG + threadIdx.x;
    const int v = i < n ? in[i] : 0;
    const int inc = wave_inclusive_scan(v);
    const int lane = int(threadIdx.x) & 63, wave = int(threadIdx.x) >> 6;
    if (lane == 63) s_wave[wave] = inc;
    __syncthreads();
    int before = 0;
    for (int w = 0; w < wave; ++w) before += s_wave[w];
    if (i < n) out[i] = before + inc - v;
    if (threadIdx.x == SCAN_WG - 1) total[blockIdx.x] = int64_t(before) + inc;
}
// exclusive scan of the workgroup totals (one workgroup, sequential over chunks of 1024), grand total behind them
__global__ __launch_bounds__(SCAN_WG) void scan_totals_kernel(int64_t n_wg, int64_t *total) {
    __shared__ long long s_wave[SCAN_WG / 64];
    __shared__ long long s_carry;
    if (threadIdx.x == 0) s_carry = 0;
    __syncthreads();
    for (int64_t base = 0; base < n_wg; base += SCAN_WG) {
        const int64_t i = base + threadIdx.x;
        const long long v = i < n_wg ? total[i] : 0;
        long long inc = v;
        const int lane = int(threadIdx.x) & 63, wave = int(threadIdx.x) >> 6;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const long long t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        if (lane == 63) s_wave[wave] = inc;
        __syncthreads();
        long long before = s_carry;
        for (int w = 0; w < wave; ++w) before += s_wave[w];
        if (i < n_wg) total[i] = before + inc - v;
        __syncthreads();
        if (threadIdx.x == SCAN_WG - 1) s_carry = before + inc;
        __syncthreads();
    }
    if (threadIdx.x == 0) total[n_wg] = s_carry;
}
__global__ __launch_bounds__(SCAN_WG) void scan_add_kernel(int64_t n, int32_t *out, const int64_t *total) {
    const int64_t i = int64_t(blockIdx.x) * SCAN_WG + threadIdx.x;
    if (i < n) out[i] += int32_t(total[blockIdx.x]);
    if (i == n - 1 || (n == 0 && i == 0)) out[n] = int32_t(total[gridDim.x]);
}
// indptr[0..n] = exclusive scan of len[0..n); returns the total (synchronises)
int64_t device_row_pointers(int64_t n, const int32_t *len, int32_t *indptr, hipStream_t s) {
    if (n == 0) { OMG_HIP(hipMemsetAsync(indptr, 0, sizeof(int32_t), s)); return 0; }
    const int64_t n_wg = (n + SCAN_WG - 1) / SCAN_WG;
    DevBuf<int64_t> total(n_wg + 1);
    hipLaunchKernelGGL(scan_local_kernel, dim3(unsigned(n_wg)), dim3(SCAN_WG), 0, s, n, len, indptr, total.p);
    hipLaunchKernelGGL(scan_totals_kernel, dim3(1), dim3(SCAN_WG), 0, s, n_wg, total.p);
    hipLaunchKernelGGL(scan_add_kernel, dim3(unsigned(n_wg)), dim3(SCAN_WG), 0, s, n, indptr, total.p);
    OMG_HIP(hipGetLastError());
    int64_t grand = 0;
    OMG_HIP(hipMemcpyAsync(&grand, total.p + n_wg, sizeof(int64_t), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    OMG_REQUIRE(grand < INT32_MAX, "sparse product: result exceeds int32 nnz");
    return grand;
}

// ---- fused Galerkin product for an aggregation restriction ------------------------------------------------
// (R A) R^T of openmg/operators.py:184-186 where every column of R holds exactly ONE entry (cell
// aggregation: a fine unknown belongs to one coarse one) and A's rows have ascending columns — the
// hierarchy path.  One WAVE per coarse row I, both products in LDS, no intermediate in HBM:
//   1. (R A)(I, j) for the columns j of the rows k of I's aggregate: k in R's stored order (one after the
//      other), the entries of row k by the lanes, product R(I,k) * A(k,j) rounded, then ADDED to the
//      column's sum in a hash table keyed by j — per column the additions of SciPy's csr_matmat in its
//      order; sums that are exactly zero are dropped, as SciPy drops them;
//   2. C(I, J) = sum over the j of aggregate J, ascending j, of (R A)(I, j) * R(J, j): the kept columns
//      are ranked by (J, j) (a rank sort: a few hundred keys per row), a lane adds up one J's run in that
//      order, zero sums are dropped, and the row comes out with ascending columns.
// Exactly the arithmetic, in exactly the order, of spgemm(spgemm(R, A), R^T) above (tests: bit for bit
// against it, and against the reference's products of golden g3).  Two passes — row lengths, then the
// entries straight into the result — instead of a scratch list per row.
constexpr int RAP_WAVES = 4;          // coarse rows per workgroup
constexpr int RAP_CAP = 512;          // hash slots per wave (distinct columns of a row's aggregate: at most RAP_CAP / 2)
constexpr int RAP_EMPTY = -1;

__global__ void column_owner_kernel(int64_t n_rows, const int32_t *rp, const int32_t *ri, const double *rv,
                                    int32_t *owner, double *weight, int32_t *seen) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n_rows; r += (int64_t)gridDim.x * blockDim.x)
        for (int32_t p = rp[r]; p < rp[r + 1]; ++p) {
            const int32_t c = ri[p];
            owner[c] = int32_t(r);
            weight[c] = rv[p];
            atomicAdd(&seen[c], 1);
        }
}
// flag[0] |= 1 when some column of R is owned more than once (or never: handled by owner = -1);
// flag[0] |= 2 when a row of A has columns that do not ascend strictly; flag[1] = max over coarse rows of
// the number of entries of its aggregate's rows
__global__ void rap_check_kernel(int64_t n, int64_t nc, const int32_t *seen, const int32_t *ap, const int32_t *ai,
                                 const int32_t *rp, const int32_t *ri, int32_t *flag) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        if (seen[i] > 1) atomicOr(&flag[0], 1);
        for (int32_t p = ap[i] + 1; p < ap[i + 1]; ++p)
            if (ai[p] <= ai[p - 1]) { atomicOr(&flag[0], 2); break; }
        if (i < nc) {
            int32_t t = 0;
            for (int32_t p = rp[i]; p < rp[i + 1]; ++p) t += ap[ri[p] + 1] - ap[ri[p]];
            atomicMax(&flag[1], t);
        }
    }
}

template <bool FILL, int CAP>
__global__ __launch_bounds__(RAP_WAVES * 64) void rap_aggregation_kernel(
    int64_t nc, const int32_t *rp, const int32_t *ri, const double *rv, const int32_t *ap, const int32_t *ai,
    const double *av, const int32_t *owner, const double *weight, int32_t *row_len, const int32_t *cp, int32_t *ci,
    double *cv, int32_t *stash_i, double *stash_v, int stash_k, int32_t *overflow) {
    __shared__ int s_key[RAP_WAVES][CAP];
    __shared__ double s_val[RAP_WAVES][CAP];
    __shared__ unsigned long long s_sort[RAP_WAVES][CAP / 2];     // (J << 32 | j) of the kept columns, then ranked
    __shared__ double s_prod[RAP_WAVES][CAP / 2];
    const int lane = int(threadIdx.x) & 63, wave = int(threadIdx.x) >> 6;
    const int64_t I = int64_t(blockIdx.x) * RAP_WAVES + wave;
    if (I >= nc) return;                                             // (whole waves: no workgroup barrier below)
    int *key = s_key[wave];
    double *val = s_val[wave];
    unsigned long long *srt = s_sort[wave];
    double *prd = s_prod[wave];
    for (int q = lane; q < CAP; q += 64) key[q] = RAP_EMPTY;
    // 1. (R A)(I, :).  All of the aggregate's rows are fetched at once (two dependent loads instead of three
    // per row): lane m takes R's entry m and row k_m's extent, a wave scan lays the rows' entries end to end,
    // every lane then fetches entries of that list (products into LDS); the additions follow row by row.
    const int32_t r0 = rp[I], nk = rp[I + 1] - r0;
    int32_t *ent_j = reinterpret_cast<int32_t *>(srt);               // staging (the sort arrays are free until step 2)
    double *ent_p = prd;
    int done = 0;                                                    // rows of the aggregate handled so far (64 per round)
    while (done < nk) {
        const int m = done + lane;
        int32_t k = 0, beg = 0, len = 0;
        double r = 0.0;
        if (m < nk) { k = ri[r0 + m]; r = rv[r0 + m]; beg = ap[k]; len = ap[k + 1] - beg; }
        int inc = len;
#pragma unroll
        for (int off = 1; off < 64; off <<= 1) {
            const int t = __shfl_up(inc, off, 64);
            if (lane >= off) inc += t;
        }
        const int total = __shfl(inc, 63, 64);                      // <= CAP / 2 (host check)
        const int start = inc - len;
        const int rows_here = min(64, nk - done);
        for (int e = lane; e < total; e += 64) {
            // the row this entry belongs to: the last one whose start is <= e
            int row = 0;
            for (int q = 1; q < rows_here; ++q) row = __shfl(start, q, 64) <= e ? q : row;
            const int32_t p = __shfl(beg, row, 64) + (e - __shfl(start, row, 64));
            ent_j[e] = ai[p];
            ent_p[e] = __dmul_rn(__shfl(r, row, 64), av[p]);         // no FMA contraction: SciPy's order and rounding
        }
        __builtin_amdgcn_wave_barrier();
        for (int q = 0; q < rows_here; ++q) {                        // the additions, one row of the aggregate after the other
            const int qs = __shfl(start, q, 64), ql = __shfl(len, q, 64);
            for (int e = lane; e < ql; e += 64) {
                const int32_t j = ent_j[qs + e];
                const double prod = ent_p[qs + e];
                unsigned h = ((unsigned(j) * 2654435761u) >> 23) & (CAP - 1);
                for (;;) {
                    const int old = atomicCAS(&key[h], RAP_EMPTY, j);
                    if (old == RAP_EMPTY) { val[h] = prod; break; }  // csr_matmat: sums[j] starts at 0 and 0 + prod == prod
                    if (old == j) { val[h] = __dadd_rn(val[h], prod); break; }
                    h = (h + 1) & (CAP - 1);
                }
            }
            __builtin_amdgcn_wave_barrier();
        }
        done += rows_here;
    }
    // 2. the kept columns, compacted; key (J, j), product with R(J, j)
    int count = 0;
    for (int base = 0; base < CAP; base += 64) {
        const int j = key[base + lane];
        const double v = j != RAP_EMPTY ? val[base + lane] : 0.0;
        const bool keep = j != RAP_EMPTY && v != 0.0;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int pos = count + __popcll(mask & ((1ull << lane) - 1ull));
            srt[pos] = (unsigned long long)(unsigned)owner[j] << 32 | (unsigned)j;
            prd[pos] = __dmul_rn(v, weight[j]);
        }
        count += __popcll(mask);
    }
    __builtin_amdgcn_wave_barrier();
    // rank sort by (J, j) into key[] (ranks) — keys are distinct
    int *rank_of = key;                                              // reuse: rank -> position
    for (int e = lane; e < count; e += 64) {
        const unsigned long long mine = srt[e];
        int rank = 0;
        for (int f = 0; f < count; ++f) rank += srt[f] < mine ? 1 : 0;
        rank_of[rank] = e;
    }
    __builtin_amdgcn_wave_barrier();
    // runs of equal J in rank order: a lane per run head adds the run up in order
    int out_count = 0;
    const int32_t dst0 = FILL ? cp[I] : 0;
    for (int base = 0; base < count; base += 64) {
        const int rk = base + lane;
        bool head = false;
        unsigned J = 0;
        if (rk < count) {
            J = unsigned(srt[rank_of[rk]] >> 32);
            head = rk == 0 || unsigned(srt[rank_of[rk - 1]] >> 32) != J;
        }
        double sum = 0.0;
        if (head) {
            for (int q = rk; q < count && unsigned(srt[rank_of[q]] >> 32) == J; ++q) sum = __dadd_rn(sum, prd[rank_of[q]]);
        }
        const bool keep = head && sum != 0.0;
        const unsigned long long mask = __ballot(keep);
        if (keep) {
            const int pos = out_count + __popcll(mask & ((1ull << lane) - 1ull));
            if (FILL) {
                ci[dst0 + pos] = int32_t(J);
                cv[dst0 + pos] = sum;
            } else if (pos < stash_k) {
                // the counting pass keeps what it has computed, stash_k entries per row: if every row fits, the second
                // pass is a copy (rap_rows_kernel) instead of the same computation again
                stash_i[I * stash_k + pos] = int32_t(J);
                stash_v[I * stash_k + pos] = sum;
            }
        }
        out_count += __popcll(mask);
    }
    if (!FILL && lane == 0) {
        row_len[I] = out_count;
        if (out_count > stash_k) atomicOr(overflow, 1);
    }
}

// the rows the counting pass has kept, into CSR order (a lane per row: a few tens of contiguous bytes each way)
__global__ void rap_rows_kernel(int64_t nc, int k, const int32_t *cp, const int32_t *si, const double *sv, int32_t *ci, double *cv) {
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < nc; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t src = r * k, dst = cp[r];
        const int len = cp[r + 1] - cp[r];
        for (int e = 0; e < len; ++e) {
            ci[dst + e] = si[src + e];
            cv[dst + e] = sv[src + e];
        }
    }
}

// true: C = (R A) R^T by the fused kernel; false: the operands do not qualify (the caller takes the two products)
bool rap_aggregation(const DevMat &R, const DevMat &A, omg_csr_result &C, hipStream_t s) {
    {
        const char *e = getenv("OMG_RAP_FUSED");
        if (e && e[0] == '0') return false;
    }
    const int64_t nc = R.n_rows, n = A.n_rows;
    if (nc == 0 || n == 0 || R.nnz == 0) return false;
    DevBuf<int32_t> owner(n), seen(n), flag(3);
    DevBuf<double> weight(n);
    OMG_HIP(hipMemsetAsync(owner.p, 0xff, size_t(n) * sizeof(int32_t), s));
    OMG_HIP(hipMemsetAsync(seen.p, 0, size_t(n) * sizeof(int32_t), s));
    OMG_HIP(hipMemsetAsync(flag.p, 0, 3 * sizeof(int32_t), s));
    hipLaunchKernelGGL(column_owner_kernel, dim3(grid1d(nc)), dim3(256), 0, s, nc, R.indptr.p, R.indices.p, R.data.p, owner.p,
                       weight.p, seen.p);
    hipLaunchKernelGGL(rap_check_kernel, dim3(grid1d(n)), dim3(256), 0, s, n, nc, seen.p, A.indptr.p, A.indices.p, R.indptr.p,
                       R.indices.p, flag.p);
    int32_t h_flag[2] = {0, 0};
    OMG_HIP(hipMemcpyAsync(h_flag, flag.p, sizeof(h_flag), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    // (a column of R that nobody owns contributes nothing to R^T: owner -1 would be read — require full cover)
    if (h_flag[0] != 0 || h_flag[1] > RAP_CAP / 2 || R.nnz != n) return false;
    C.n_rows = nc;
    C.n_cols = nc;
    DevBuf<int32_t> row_len(nc);
    const dim3 grid(unsigned((nc + RAP_WAVES - 1) / RAP_WAVES)), block(RAP_WAVES * 64);
    // the hash table is sized by the longest aggregate (the compaction walks every slot)
    const int cap = h_flag[1] <= 64 ? 128 : h_flag[1] <= 128 ? 256 : RAP_CAP;
    // what the counting pass keeps per row: about a fine row's length, which is what a Galerkin row of a grid stencil
    // under 2^d aggregation has (7 -> 8, 27 -> 32 entries; longer rows: the second pass computes again)
    const int stash_k = std::min(32, ((int(h_flag[1]) + 7) / 8 + 7) / 8 * 8);
    DevBuf<int32_t> stash_i(size_t(nc) * size_t(stash_k));
    DevBuf<double> stash_v(size_t(nc) * size_t(stash_k));
    auto pass = [&](bool fill) {
#define OMG_RAP_LAUNCH(F, CAPV)                                                                                              \
        hipLaunchKernelGGL((rap_aggregation_kernel<F, CAPV>), grid, block, 0, s, nc, R.indptr.p, R.indices.p, R.data.p,        \
                           A.indptr.p, A.indices.p, A.data.p, owner.p, weight.p, row_len.p, C.indptr.p, C.indices.p, C.data.p,  \
                           stash_i.p, stash_v.p, stash_k, flag.p + 2)
        if (fill) { if (cap == 128) OMG_RAP_LAUNCH(true, 128); else if (cap == 256) OMG_RAP_LAUNCH(true, 256); else OMG_RAP_LAUNCH(true, RAP_CAP); }
        else { if (cap == 128) OMG_RAP_LAUNCH(false, 128); else if (cap == 256) OMG_RAP_LAUNCH(false, 256); else OMG_RAP_LAUNCH(false, RAP_CAP); }
#undef OMG_RAP_LAUNCH
    };
    pass(false);
    C.indptr.alloc(nc + 1);
    C.nnz = device_row_pointers(nc, row_len.p, C.indptr.p, s);
    C.indices.alloc(std::max<int64_t>(C.nnz, 1));
    C.data.alloc(std::max<int64_t>(C.nnz, 1));
    int32_t overflow = 0;
    OMG_HIP(hipMemcpyAsync(&overflow, flag.p + 2, sizeof(int32_t), hipMemcpyDeviceToHost, s));
    OMG_HIP(hipStreamSynchronize(s));
    if (overflow) pass(true);                                        // some row is longer than the stash: the computation again
    else hipLaunchKernelGGL(rap_rows_kernel, dim3(grid1d(nc)), dim3(256), 0, s, nc, stash_k, C.indptr.p, stash_i.p, stash_v.p, C.indices.p, C.data.p);
    OMG_HIP(hipGetLastError());
    OMG_HIP(hipStreamSynchronize(s));
    return true;
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

// perm[slot] = natural row, inv[row] = slot of a closed-form ordering (either pointer may be null)
__global__ void ordering_fill_kernel(int kind, int nx, int ny, int nz, int32_t *perm, int32_t *inv) {
    const int64_t n = int64_t(nx) * ny * nz;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = r % nx, j = (r / nx) % ny, k = r / (int64_t(nx) * ny);
        int64_t slot;
        if (kind == 2) {
            slot = (((i + j + k) & 1) ? n / 2 : 0) + r / 2;
        } else {
            const int64_t c = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2), na = n / 8;
            slot = c * na + ((k >> 1) * (ny / 2) + (j >> 1)) * (nx / 2) + (i >> 1);
        }
        if (inv) inv[r] = int32_t(slot);
        if (perm) perm[slot] = int32_t(r);
    }
}

}  // namespace

namespace omg {

void fill_ordering_device(int kind, int nx, int ny, int nz, int32_t *perm, int32_t *inv, hipStream_t s) {
    const int64_t n = int64_t(nx) * ny * nz;
    hipLaunchKernelGGL(ordering_fill_kernel, dim3(grid1d(n)), dim3(256), 0, s, kind, nx, ny, nz, perm, inv);
    OMG_HIP(hipGetLastError());
}

bool rap_aggregation_device(const DevCsrPlain &R, const DevCsrPlain &A, DevCsrPlain &C, hipStream_t s) {
    omg_csr_result T;
    if (!rap_aggregation(R, A, T, s)) return false;
    as_devmat(std::move(T), C);
    return true;
}

HostCsr download_csr(const DevCsrPlain &M, hipStream_t s) {
    HostCsr H;
    H.n_rows = M.n_rows; H.n_cols = M.n_cols; H.nnz = M.nnz;
    H.indptr.resize(size_t(M.n_rows) + 1);
    H.indices.resize(size_t(M.nnz));
    H.data.resize(size_t(M.nnz));
    download_staged(H.indptr.data(), M.indptr.p, (size_t(M.n_rows) + 1) * sizeof(int32_t), s);
    download_staged(H.indices.data(), M.indices.p, size_t(M.nnz) * sizeof(int32_t), s);
    download_staged(H.data.data(), M.data.p, size_t(M.nnz) * sizeof(double), s);
    return H;
}

void galerkin_chain_device(const omg_csr &A0, int dim, const int64_t *shape, int n_restrictions, std::vector<DevCsrPlain> &A,
                           std::vector<DevCsrPlain> &R, hipStream_t s) {
    OMG_REQUIRE(dim >= 1 && dim <= 3 && n_restrictions >= 1, "bad argument");
    A.clear();
    R.clear();
    A.resize(size_t(n_restrictions) + 1);
    R.resize(size_t(n_restrictions));
    { SetupTimer tm("device setup: upload the fine operator"); upload(A[0], A0, s); OMG_HIP(hipStreamSynchronize(s)); }
    int64_t ext[3] = {1, 1, 1};
    for (int d = 0; d < dim; ++d) ext[d] = shape[d];
    for (int l = 0; l < n_restrictions; ++l) {
        int64_t N = 1;
        for (int d = 0; d < dim; ++d) {
            OMG_REQUIRE(ext[d] >= 2 && !(ext[d] & 1), "device setup needs even extents on every level");
            N *= ext[d];
        }
        OMG_REQUIRE(N == A[size_t(l)].n_rows, "problemShape does not match the operator");
        const int per = 1 << dim;
        const int64_t rows = N / per;
        DevCsrPlain &Rl = R[size_t(l)];
        Rl.n_rows = rows; Rl.n_cols = N; Rl.nnz = N;
        Rl.indptr.alloc(size_t(rows) + 1);
        Rl.indices.alloc(size_t(N));
        Rl.data.alloc(size_t(N));
        hipLaunchKernelGGL(restriction_kernel, dim3(grid1d(rows + 1)), dim3(256), 0, s, dim, ext[0], dim >= 2 ? ext[1] : 1, dim >= 3 ? ext[2] : 1,
                           rows, Rl.indptr.p, Rl.indices.p, Rl.data.p);
        OMG_HIP(hipGetLastError());
        SetupTimer tm("device setup: Galerkin product of a level");
        omg_csr_result C;
        if (!rap_aggregation(Rl, A[size_t(l)], C, s)) {
            DevMat Rt, RA;
            device_transpose(Rl, Rt, s);
            omg_csr_result T;
            spgemm(Rl, A[size_t(l)], T, s);
            as_devmat(std::move(T), RA);
            spgemm(RA, Rt, C, s);
        }
        as_devmat(std::move(C), A[size_t(l) + 1]);
        for (int d = 0; d < dim; ++d) ext[d] /= 2;
    }
}

}  // namespace omg

namespace {

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
        {
            SetupTimer tm("rap: fused (R A) R^T, one wave per coarse row");
            std::unique_ptr<omg_csr_result> F(new omg_csr_result);
            if (rap_aggregation(dR, dA, *F, st.s)) {
                *n_rows = F->n_rows; *n_cols = F->n_cols; *nnz = F->nnz;
                *out = F.release();
                return;
            }
        }
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
        download_staged(indptr, res->indptr.p, size_t(res->n_rows + 1) * sizeof(int32_t), nullptr);
        download_staged(indices, res->indices.p, size_t(res->nnz) * sizeof(int32_t), nullptr);
        download_staged(data, res->data.p, size_t(res->nnz) * sizeof(double), nullptr);
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
        download_staged(indptr, dp.p, size_t(rows + 1) * sizeof(int32_t), st.s);
        download_staged(indices, di.p, size_t(rows * per) * sizeof(int32_t), st.s);
        download_staged(data, dv.p, size_t(rows * per) * sizeof(double), st.s);
        for (int64_t r = rows + 1; r <= n; ++r) indptr[r] = (int32_t)(rows * per);   // trailing empty rows
        *n_rows = n;
        *nnz = rows * per;
    });
}

}  // extern "C"
