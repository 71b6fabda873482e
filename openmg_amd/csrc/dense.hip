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
// inverted in registers (partial pivoting INSIDE the block: just the method for its inverse), row block K becomes D^-1 R_K, and
// every other row block R_i <- R_i - A_iK R_K as 64 x 64 x 64 tile products on the fp64 matrix cores — one sweep of the
// active columns (left of the pivots the matrix is already the identity, right of them the inverse's columns are still
// untouched: n columns in all) per 64 pivots, 3 launches per block.  Tiles that hold only zeros — below and right of a
// banded operator's band, which is what a grid's coarsest operator is — are found as the step's operands are staged and
// skipped: a third of the products remain.  No pivoting ACROSS blocks: exact for the symmetric positive definite /
// diagonally dominant operators a Galerkin hierarchy ends in; the caller checks the result against the operator and falls
// back to the pivoted elimination where it is not an inverse.
constexpr int GJB = 64;
typedef double gj_v4d __attribute__((ext_vector_type(4)));

// max over the wave of an unsigned key -> every lane (row_shr 1, 2, 4, 8 within the rows of 16, then the rows' last lanes
// broadcast along: the total is in lane 63)
__device__ __forceinline__ unsigned wave_max_u32(unsigned x) {
    x = max(x, unsigned(__builtin_amdgcn_update_dpp(0, int(x), 0x111, 0xf, 0xf, false)));
    x = max(x, unsigned(__builtin_amdgcn_update_dpp(0, int(x), 0x112, 0xf, 0xf, false)));
    x = max(x, unsigned(__builtin_amdgcn_update_dpp(0, int(x), 0x114, 0xf, 0xf, false)));
    x = max(x, unsigned(__builtin_amdgcn_update_dpp(0, int(x), 0x118, 0xf, 0xf, false)));
    x = max(x, unsigned(__builtin_amdgcn_update_dpp(0, int(x), 0x142, 0xa, 0xf, false)));
    x = max(x, unsigned(__builtin_amdgcn_update_dpp(0, int(x), 0x143, 0xc, 0xf, false)));
    return unsigned(__builtin_amdgcn_readlane(int(x), 63));
}
__device__ __forceinline__ double lane_value(double v, int lane_uniform) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane_uniform), hi = __builtin_amdgcn_readlane(__double2hiint(v), lane_uniform);
    return __hiloint2double(hi, lo);
}

// D^-1 of the diagonal block [k0, k0 + nb) -> dinv (GJB x GJB, row-major).  Gauss-Jordan on [D | I], a lane per ROW and a
// wave per 32 of the 128 columns, all in registers (the loop over the pivots is unrolled: the pivot's column is a fixed
// register).  Pivoting is implicit — the pivot of column k is the unused row with the largest magnitude as far as the
// upper word of the double tells (exponent and 20 bits: any such row is as good a pivot) — rows stay in their lanes and the
// lane that was pivot k ends up holding row k of the inverse.  Per pivot: the wave that owns column k finds the pivot's
// lane (a DPP reduction), publishes the column, the lane and the pivot's reciprocal (ONE workgroup barrier, double
// buffered), every wave takes the pivot row's 32 values of its own columns out of that lane (v_readlane), scales them and
// eliminates.  (The first version kept [D | I] in LDS with five barriers per pivot: 261 us per block; this one: see DESIGN.md.)
// (FRESH: the block was written by this workgroup a moment ago — loads that do not stop at this compute unit's L1)
template <bool FRESH>
__device__ __forceinline__ void gjb_diag_block(const double *W, int64_t ld, int64_t k0, int nb, double *dinv, int *singular) {
    __shared__ double s_col[2][GJB];
    __shared__ double s_rp[2];
    __shared__ int s_p[2];
    const int t = int(threadIdx.x), w = t >> 6, lane = t & 63;
    double row[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) {
        const int col = 32 * w + c;
        double v;
        if (col < GJB) {
            v = lane == col ? 1.0 : 0.0;
            if (lane < nb && col < nb) {
                const double *const src = W + (k0 + lane) * ld + k0 + col;
                v = FRESH ? __hip_atomic_load(src, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : *src;
            }
        } else {
            v = (col - GJB == lane) ? 1.0 : 0.0;
        }
        row[c] = v;
    }
    bool used = lane >= nb;
    int mine = lane;                                   // the pivot this lane's row was (rows beyond nb: identity rows)
#pragma unroll
    for (int k = 0; k < GJB; ++k) {
        if (k >= nb) continue;                         // (uniform)
        const int par = k & 1;
        if (w == (k >> 5)) {
            const double v = row[k & 31];
            const unsigned key = used ? 0u : (unsigned(__double2hiint(v)) & 0x7fffffffu);
            const unsigned m = wave_max_u32(key);
            const unsigned long long eq = __builtin_amdgcn_ballot_w64(!used && key == m);
            const int p = eq ? int(__builtin_ctzll(eq)) : 0;
            const double pv = lane_value(v, p);
            double r = __builtin_amdgcn_rcp(pv);
            r = fma(r, fma(-pv, r, 1.0), r);
            r = fma(r, fma(-pv, r, 1.0), r);
            s_col[par][lane] = v;
            if (lane == 0) {
                s_p[par] = p;
                s_rp[par] = r;
                if (m == 0u || m >= 0x7ff00000u) *singular = 1;      // no pivot to speak of (zero / denormal column), or inf / nan
            }
        }
        __syncthreads();
        const int p = __builtin_amdgcn_readfirstlane(s_p[par]);
        const double rp = s_rp[par];
        const double f = s_col[par][lane];
        const bool piv = lane == p;
#pragma unroll
        for (int c = 0; c < 32; ++c) {
            const double pr = lane_value(row[c], p) * rp;      // the scaled pivot row: the same in every lane
            const double e = fma(-f, pr, row[c]);
            row[c] = piv ? pr : e;
        }
        if (piv) { used = true; mine = k; }
    }
    if (w >= 2) {
#pragma unroll
        for (int c = 0; c < 32; ++c) dinv[mine * GJB + 32 * (w - 2) + c] = row[c];
    }
}
__global__ __launch_bounds__(256) void gjb_diag_kernel(const double *W, int64_t ld, int64_t k0, int nb, double *dinv, int *singular) {
    gjb_diag_block<false>(W, ld, k0, nb, dinv, singular);
}

// column tile `ct` of the active columns of block step k0 .. k1: left of the matrix [k1, n), then the inverse's [n, n + k1)
__device__ __forceinline__ int64_t gjb_col0(int64_t ct, int64_t n, int64_t k1, int64_t &c_end) {
    const int64_t left_tiles = (n - k1 + GJB - 1) / GJB;
    if (ct < left_tiles) { c_end = n; return k1 + ct * GJB; }
    c_end = n + k1;
    return n + (ct - left_tiles) * GJB;
}

// The matrix-core tiles below: v_mfma_f64_16x16x4 — A: lane 16 k + i holds A[i][k], B: lane 16 k + j holds B[k][j],
// D: register r of lane l holds row 4 r + l / 16 of column l % 16 (coarse.hip uses the same layout).

// Workgroups [0, col_groups): two column tiles each, two waves per tile (the upper / lower 32 rows of the result) — row
// block K <- D^-1 (row block K) there, unless the tile holds only zeros (col_flag says which).  Both waves hold the whole
// tile before either overwrites it.  The others, one per row tile: the pivot columns of its 64 rows, NEGATED and transposed,
// into `panel` (GJB x n_pad: the elimination's A operand, read while the matrix's own copy is overwritten), row_flag: any nonzero.
__global__ __launch_bounds__(256) void gjb_rowblock_kernel(double *W, int64_t n, int64_t ld, int64_t k0, int nb, const double *dinv, double *panel,
                                                           int64_t n_pad, int col_tiles, int col_groups, int *row_flag, int *col_flag) {
    const int t = int(threadIdx.x), w = t >> 6, lane = t & 63;
    if (int(blockIdx.x) >= col_groups) {
        __shared__ double tile[GJB][GJB + 1];
        const int rt = int(blockIdx.x) - col_groups;
        const int64_t r0 = int64_t(rt) * GJB;
        int any = 0;
        for (int q = t; q < GJB * GJB; q += 256) {
            const int r = q / GJB, c = q % GJB;
            const double v = (r0 + r < n && c < nb) ? W[(r0 + r) * ld + k0 + c] : 0.0;
            any |= v != 0.0;
            tile[r][c] = -v;
        }
        any = __syncthreads_or(any);
        for (int q = t; q < GJB * GJB; q += 256) {
            const int c = q / GJB, r = q % GJB;
            panel[int64_t(c) * n_pad + r0 + r] = tile[r][c];
        }
        if (t == 0) row_flag[rt] = any;
        return;
    }
    const int ct = 2 * int(blockIdx.x) + (w >> 1), half = w & 1;
    const bool have = ct < col_tiles;
    int64_t c_end = 0;
    const int64_t c0 = have ? gjb_col0(ct, n, k0 + nb, c_end) : 0;
    const int q4 = lane >> 4, j = lane & 15;
    // the whole 64 x 64 tile of row block K first (it is overwritten in place): b[ks][jt] = R[4 ks + q4][16 jt + j]
    double b[16][4], a[16][2];
    bool any = false;
#pragma unroll
    for (int ks = 0; ks < 16; ++ks) {
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) {
            const int r = 4 * ks + q4;
            const int64_t cc = c0 + 16 * jt + j;
            const double v = (have && r < nb && cc < c_end) ? W[(k0 + r) * ld + cc] : 0.0;
            any = any || v != 0.0;
            b[ks][jt] = v;
        }
#pragma unroll
        for (int it = 0; it < 2; ++it) a[ks][it] = dinv[(32 * half + 16 * it + j) * GJB + 4 * ks + q4];      // (lane 16 k + i: here i = j, k = q4)
    }
    const bool some = __builtin_amdgcn_ballot_w64(any) != 0;
    if (have && half == 0 && lane == 0) col_flag[ct] = some ? 1 : 0;
    __syncthreads();
    if (!some) return;
    gj_v4d acc[2][4];
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt) acc[it][jt] = gj_v4d{0.0, 0.0, 0.0, 0.0};
#pragma unroll
    for (int ks = 0; ks < 16; ++ks)
#pragma unroll
        for (int it = 0; it < 2; ++it)
#pragma unroll
            for (int jt = 0; jt < 4; ++jt) acc[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks][it], b[ks][jt], acc[it][jt], 0, 0, 0);
#pragma unroll
    for (int it = 0; it < 2; ++it)
#pragma unroll
        for (int jt = 0; jt < 4; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int rr = 32 * half + 16 * it + 4 * r + q4;
                const int64_t cc = c0 + 16 * jt + j;
                if (rr < nb && cc < c_end) W[(k0 + rr) * ld + cc] = acc[it][jt][r];
            }
}

// R_i <- R_i - A_iK R_K, a wave per 64 x 32 half tile (a workgroup: one row tile, two column tiles; row block K and the
// tiles either of whose operands is all zeros are skipped).  Operands go straight from global memory / L2 into the matrix
// instructions' registers, half of the k-steps' at a time; two waves per SIMD, one's loads under the other's products.
// next_rt >= 0: the FIRST workgroup takes the next step's row tile (it exchanges places with that tile's own workgroup) and,
// its tiles done, inverts the next diagonal block — column tile 0 there — into dinv: the chain of 64 dependent pivots
// runs beside this step's other tiles instead of behind them.  (On a second stream with events it cost more than it hid:
// 14.8 against 10.7 ms per 4096 x 4096 inverse.)
template <typename = void>
__device__ __forceinline__ void gjb_eliminate_tile(double *W, int64_t n, int64_t ld, int64_t k0, int nb, const double *panel, int64_t n_pad, int rt, int ct,
                                                   int half, int lane, int64_t c0, int64_t c_end) {
    const int64_t r0 = int64_t(rt) * GJB;
    const int q4 = lane >> 4, j = lane & 15;
    gj_v4d acc[4][2];
    const double *const pa = panel + r0 + j;           // A[i][k] = panel[k * n_pad + r0 + i] (already negated)
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t rr = r0 + 16 * it + 4 * r + q4, cc = c0 + 16 * jt + j;
                acc[it][jt][r] = (rr < n && cc < c_end) ? W[rr * ld + cc] : 0.0;
            }
    // (two phases of eight k-steps: all 256 registers would be needed for the operands of all sixteen at once)
#pragma unroll
    for (int ph = 0; ph < 2; ++ph) {
        double a[8][4], b[8][2];
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) {
            const int kk = 4 * (8 * ph + ks) + q4;
#pragma unroll
            for (int it = 0; it < 4; ++it) a[ks][it] = pa[int64_t(kk) * n_pad + 16 * it];
#pragma unroll
            for (int jt = 0; jt < 2; ++jt) {
                const int64_t cc = c0 + 16 * jt + j;
                b[ks][jt] = (kk < nb && cc < c_end) ? W[(k0 + kk) * ld + cc] : 0.0;
            }
        }
#pragma unroll
        for (int ks = 0; ks < 8; ++ks)
#pragma unroll
            for (int it = 0; it < 4; ++it)
#pragma unroll
                for (int jt = 0; jt < 2; ++jt) acc[it][jt] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[ks][it], b[ks][jt], acc[it][jt], 0, 0, 0);
    }
#pragma unroll
    for (int it = 0; it < 4; ++it)
#pragma unroll
        for (int jt = 0; jt < 2; ++jt)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const int64_t rr = r0 + 16 * it + 4 * r + q4, cc = c0 + 16 * jt + j;
                if (rr < n && cc < c_end) W[rr * ld + cc] = acc[it][jt][r];
            }
}

__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(2, 2)))
void gjb_eliminate_kernel(double *W, int64_t n, int64_t ld, int64_t k0, int nb, const double *panel, int64_t n_pad,
                          int row_tiles, int col_tiles, const int *row_flag, const int *col_flag, int next_rt, int next_nb, double *dinv, int *singular) {
    const int t = int(threadIdx.x), w = t >> 6, lane = t & 63;
    int rt = int(blockIdx.y);
    if (next_rt >= 0) rt = rt == 0 ? next_rt : rt == next_rt ? 0 : rt;
    const bool inverts = next_rt >= 0 && blockIdx.x == 0 && blockIdx.y == 0;      // (uniform over the workgroup)
    const int ct = 2 * int(blockIdx.x) + (w >> 1), half = w & 1;
    if (rt < row_tiles && ct < col_tiles && int64_t(rt) * GJB != k0 && row_flag[rt] && col_flag[ct]) {
        int64_t c_end;
        const int64_t c0 = gjb_col0(ct, n, k0 + nb, c_end) + 32 * half;
        gjb_eliminate_tile(W, n, ld, k0, nb, panel, n_pad, rt, ct, half, lane, c0, c_end);
    }
    if (!inverts) return;
    __threadfence();                                   // the tile's new values: out of this compute unit before any wave reads them back
    __syncthreads();
    gjb_diag_block<true>(W, ld, k0 + nb, next_nb, dinv, singular);
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
    const int64_t row_tiles = (n + GJB - 1) / GJB, n_pad = row_tiles * GJB;
    DevBuf<double> dinv(size_t(GJB) * GJB), panel(size_t(n_pad) * GJB);
    DevBuf<int> flag(1), tile_flag(size_t(2 * row_tiles + 2));
    flag.zero(s);
    int *const row_flag = tile_flag.p, *const col_flag = tile_flag.p + row_tiles;
    // Look-ahead: the next diagonal block is final as soon as ITS tile of this step's elimination is — the first workgroup of
    // the elimination takes that tile and then inverts the block (gjb_eliminate_kernel); OMG_DENSE_LOOKAHEAD=0: a launch of its own
    static const bool lookahead = [] { const char *e = experiment_env("OMG_DENSE_LOOKAHEAD"); return !(e && e[0] == '0'); }();
    hipLaunchKernelGGL(gjb_diag_kernel, dim3(1), dim3(256), 0, s, W, ld, int64_t(0), int(std::min<int64_t>(GJB, n)), dinv.p, flag.p);
    for (int64_t k0 = 0; k0 < n; k0 += GJB) {
        const int nb = int(std::min<int64_t>(GJB, n - k0));
        const int64_t k1 = k0 + nb;
        const int col_tiles = int((n - k1 + GJB - 1) / GJB + (k1 + GJB - 1) / GJB);
        const int col_groups = (col_tiles + 1) / 2;
        const bool more = k1 < n;
        const int nb1 = more ? int(std::min<int64_t>(GJB, n - k1)) : 0;
        hipLaunchKernelGGL(gjb_rowblock_kernel, dim3(unsigned(col_groups + row_tiles)), dim3(256), 0, s, W, n, ld, k0, nb, dinv.p, panel.p, n_pad,
                           col_tiles, col_groups, row_flag, col_flag);
        const dim3 egrid(unsigned((col_tiles + 1) / 2), unsigned(row_tiles));
        const bool fused = lookahead && more;
        hipLaunchKernelGGL(gjb_eliminate_kernel, egrid, dim3(256), 0, s, W, n, ld, k0, nb, panel.p, n_pad, int(row_tiles), col_tiles, row_flag, col_flag,
                           fused ? int(k1 / GJB) : -1, nb1, dinv.p, flag.p);
        if (more && !fused) hipLaunchKernelGGL(gjb_diag_kernel, dim3(1), dim3(256), 0, s, W, ld, k1, nb1, dinv.p, flag.p);
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
void gauss_jordan_inverse(double *W, int64_t n, double *Minv, hipStream_t s, double *keep_ws) {
    if (n == 0) return;
    const int64_t ld = 2 * n;
    {
        static const bool blocked = [] { const char *e = getenv("OMG_DENSE_BLOCKED"); return !(e && e[0] == '0'); }();
        if (blocked && n >= 256) {
            DevBuf<double> keep_own;
            if (!keep_ws) keep_own.alloc((size_t)(n) * (size_t)(n));
            double *const keep = keep_ws ? keep_ws : keep_own.p;
            hipLaunchKernelGGL(copy_block_kernel, dim3(grid1d(n * n)), dim3(256), 0, s, W, n, ld, keep);
            if (blocked_inverse(W, n, keep, Minv, s)) return;
            hipLaunchKernelGGL(restore_aug_kernel, dim3(grid1d(n * ld)), dim3(256), 0, s, keep, n, W);
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
