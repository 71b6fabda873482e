/*
 * ORACLE — TEST INFRASTRUCTURE ONLY.  Not part of the product path.
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this.
 *
 * Plain-C restatement of the reference's sequential kernels so that the CPU
 * checker finishes in seconds at sizes where the reference's Python loop takes
 * minutes.  Each function names the reference lines it follows
 * (reference = tsbertalan/openmg, mounted at /root/reference in the build container).
 *
 * Build: make -C oracle      ->  oracle/libmg_oracle.so
 */
#include <math.h>
#include <stdint.h>
#include <stddef.h>

/* openmg/solvers.py:56-68 — one or more in-place lexicographic Gauss-Seidel sweeps.
 * Row sum runs over the STORED column order (solvers.py:63-65, np.dot of the row
 * slice with x gathered at the row's indices) and includes the diagonal term;
 * the update is x[i] += (b[i] - sum) / A[i,i] (solvers.py:68).  A[i,i] is the sum of
 * all stored entries of row i whose column is i (SciPy's A[i, i] adds duplicates).
 * Returns 0, or 1+i if row i has no stored diagonal (the reference would divide by 0). */
int oracle_gs_lex(int64_t n, const int32_t *indptr, const int32_t *indices,
                  const double *data, const double *b, double *x, int sweeps)
{
    for (int s = 0; s < sweeps; ++s) {
        for (int64_t i = 0; i < n; ++i) {
            double sum = 0.0, diag = 0.0;
            int have = 0;
            for (int32_t p = indptr[i]; p < indptr[i + 1]; ++p) {
                sum += data[p] * x[indices[p]];
                if (indices[p] == i) { diag += data[p]; have = 1; }
            }
            if (!have) return (int)(1 + i);
            x[i] = x[i] + (b[i] - sum) / diag;
        }
    }
    return 0;
}

/* Same sweep, but rows are visited in the order given by `order` (a permutation of
 * 0..n-1).  With order = "all rows of colour 0, then colour 1, ..." this is
 * multi-colour Gauss-Seidel; it equals oracle_gs_lex applied to the symmetrically
 * permuted system, which is how the real reference pins it (tests/golden g4). */
int oracle_gs_ordered(int64_t n, const int32_t *indptr, const int32_t *indices,
                      const double *data, const double *b, double *x,
                      const int32_t *order, int sweeps)
{
    for (int s = 0; s < sweeps; ++s) {
        for (int64_t k = 0; k < n; ++k) {
            int64_t i = order[k];
            double sum = 0.0, diag = 0.0;
            int have = 0;
            for (int32_t p = indptr[i]; p < indptr[i + 1]; ++p) {
                sum += data[p] * x[indices[p]];
                if (indices[p] == i) { diag += data[p]; have = 1; }
            }
            if (!have) return (int)(1 + i);
            x[i] = x[i] + (b[i] - sum) / diag;
        }
    }
    return 0;
}

/* openmg/tools.py:12-15 + :26 — r = b - A x, row sums in stored order
 * (scipy sparsetools csr_matvec accumulates in stored order too). */
void oracle_residual(int64_t n, const int32_t *indptr, const int32_t *indices,
                     const double *data, const double *b, const double *x, double *r)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int32_t p = indptr[i]; p < indptr[i + 1]; ++p)
            sum += data[p] * x[indices[p]];
        r[i] = b[i] - sum;
    }
}

/* y = A x (tools.py:26 -> csr_matvec). */
void oracle_spmv(int64_t n, const int32_t *indptr, const int32_t *indices,
                 const double *data, const double *x, double *y)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0;
        for (int32_t p = indptr[i]; p < indptr[i + 1]; ++p)
            sum += data[p] * x[indices[p]];
        y[i] = sum;
    }
}

/* Weighted Jacobi sweep, x_new = x + omega * D^-1 (b - A x).  NOT in the reference
 * (SURVEY D3): parity unpinned by the reference, checked only against this restatement. */
int oracle_jacobi(int64_t n, const int32_t *indptr, const int32_t *indices,
                  const double *data, const double *b, const double *x, double *xnew,
                  double omega)
{
    for (int64_t i = 0; i < n; ++i) {
        double sum = 0.0, diag = 0.0;
        int have = 0;
        for (int32_t p = indptr[i]; p < indptr[i + 1]; ++p) {
            sum += data[p] * x[indices[p]];
            if (indices[p] == i) { diag += data[p]; have = 1; }
        }
        if (!have) return (int)(1 + i);
        xnew[i] = x[i] + omega * ((b[i] - sum) / diag);
    }
    return 0;
}
