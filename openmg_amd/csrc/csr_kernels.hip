// CSR row kernels for gfx950 (MI355X).  Bandwidth-bound sparse path: no MFMA.
//
// One workgroup (256 threads = 4 wave64) owns one ROW BLOCK: a run of at most 256
// consecutive rows holding at most ROWBLK_NNZ stored entries (partition made at setup,
// setup_host.cpp:make_row_blocks).  Two phases:
//   1. the block's slice of `indices` and `data` is streamed from HBM into LDS with
//      16-byte-per-lane loads (1 KiB per wave instruction, fully coalesced, each byte of
//      the matrix is fetched exactly once);
//   2. one thread per row walks its entries in LDS IN STORED ORDER (the summation order
//      of openmg/solvers.py:63-65 and of SciPy's csr_matvec), gathering x from global
//      memory — for stencil-like matrices consecutive lanes touch consecutive x, so
//      each gather is itself a coalesced 512-byte access served by L1/L2.
// The epilogue is selected by MODE (SpMV, residual, residual norm, Gauss-Seidel set
// sweep, weighted Jacobi, y += A x).
//
// Algorithmic HBM bytes per launch (fp64 values, int32 indices; DESIGN.md "Kernels"):
//   SpMV      12*nnz + 4*(n+1) + 16*n
//   residual / GS set sweep / Jacobi   SpMV + 8*n
#include "common.h"

namespace omg {

namespace {

constexpr int NT = ROWBLK_THREADS;
constexpr int T = ROWBLK_NNZ;
// LDS image: entry k lives at k + (k >> 5).  One pad slot per 32 entries makes the
// thread-per-row reads conflict-free for EVERY row length up to 32 (row length 8 or 16
// would otherwise hit 4 or 2 banks).  +4 entries of slack for the aligned-down start.
constexpr int LDS_SLOTS = (T + 8) + ((T + 8) >> 5) + 1;

__device__ __forceinline__ int slot(int k) { return k + (k >> 5); }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef double v2d __attribute__((ext_vector_type(2)));

struct KArgs {
    const int32_t *blk_info;   // (first row, first entry) per row block, + end sentinel
    const int32_t *indptr;
    const int32_t *indices;
    const double *data;
    const double *x;
    const double *b;
    double *y;
    double *partials;
    double *zero;              // SpMV only: also clear zero[r] (next level's initial iterate)
    const int32_t *ymap;       // SpMV only: row r is stored to y[ymap[r]] (NULL: y[r])
    double omega;
};

__device__ __forceinline__ double wave_sum(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// Sum over the workgroup, result valid in thread 0.  Fixed order => deterministic.
__device__ __forceinline__ double block_sum(double v, double *s_red) {
    v = wave_sum(v);
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();
    if (lane == 0) s_red[wave] = v;
    __syncthreads();
    double tot = 0.0;
    if (threadIdx.x == 0) {
#pragma unroll
        for (int w = 0; w < NT / 64; ++w) tot += s_red[w];
    }
    return tot;
}

// Per-row operands that do not depend on the staged matrix entries.  They are loaded BEFORE
// the staging loop so that their latency overlaps the matrix stream instead of adding a
// third dependent memory round trip behind the workgroup barrier.
struct RowPre {
    int beg, end;     // CSR extent of the row (absolute entry offsets)
    int out;          // SpMV: where the result goes (ymap)
    double bv, xv;    // b[r]; x[r] (GS, Jacobi) or y[r] (y += A x)
};

template <int MODE>
__device__ __forceinline__ RowPre row_preload(const KArgs &a, int r) {
    RowPre p;
    p.beg = a.indptr[r];
    p.end = a.indptr[r + 1];
    p.bv = 0.0;
    p.xv = 0.0;
    if constexpr (MODE != ROW_SPMV && MODE != ROW_AXPY) p.bv = a.b[r];
    if constexpr (MODE == ROW_GS || MODE == ROW_JACOBI || MODE == ROW_GS_RES || MODE == ROW_GS_NORM)
        p.xv = a.x[r];
    if constexpr (MODE == ROW_AXPY) p.xv = a.y[r];
    p.out = r;
    if constexpr (MODE == ROW_SPMV) { if (a.ymap) p.out = a.ymap[r]; }
    return p;
}

// Row sums are spelled with an explicit fma everywhere so that the fused sweep+residual
// modes reproduce the plain residual kernel bit for bit.
__device__ __forceinline__ double madd(double v, double x, double acc) { return fma(v, x, acc); }

template <int MODE>
__device__ __forceinline__ void row_epilogue(const KArgs &a, int r, const RowPre &p, double sum,
                                             double diag, double &sq) {
    if constexpr (MODE == ROW_SPMV) {
        a.y[p.out] = sum;
        if (a.zero) a.zero[r] = 0.0;
    } else if constexpr (MODE == ROW_RESIDUAL) {
        a.y[r] = p.bv - sum;
    } else if constexpr (MODE == ROW_RESNORM) {
        const double res = p.bv - sum;
        a.y[r] = res;
        sq += res * res;
    } else if constexpr (MODE == ROW_NORM_ONLY) {
        const double res = p.bv - sum;
        sq += res * res;
    } else if constexpr (MODE == ROW_GS) {
        // openmg/solvers.py:68   x[i] = x[i] + (b[i] - Aix) / A[i, i]
        a.y[r] = p.xv + (p.bv - sum) / diag;
    } else if constexpr (MODE == ROW_JACOBI) {
        a.y[r] = p.xv + a.omega * ((p.bv - sum) / diag);
    } else if constexpr (MODE == ROW_AXPY) {
        a.y[r] = p.xv + sum;
    }
}

// SHORT selects the phase-2 variant for operators whose rows are so short (prolongation:
// one entry per row) that a block holds several rows per thread: the rows of a thread are
// then processed eight at a time with all their loads in flight together.
template <int MODE, bool nt, bool SHORT, int LPR>
__device__ void process_block(const KArgs &a, int blk, double *s_val, int *s_idx, double *s_red) {
    constexpr bool FUSED = (MODE == ROW_GS_RES || MODE == ROW_GS_NORM);
    constexpr bool NEED_DIAG = (MODE == ROW_GS || MODE == ROW_JACOBI || FUSED);
    constexpr bool NEED_NORM = (MODE == ROW_RESNORM || MODE == ROW_NORM_ONLY || MODE == ROW_GS_NORM);
    const int tid = threadIdx.x;
    // (first row, first entry) of this block and of the next: one round trip instead of
    // block table -> indptr
    const int2 *info = reinterpret_cast<const int2 *>(a.blk_info);
    const int2 lo = info[blk], hi = info[blk + 1];
    const int r0 = lo.x, p0 = lo.y, r1 = hi.x, p1 = hi.y;
    double sq = 0.0;

    if (p1 - p0 <= T) {
        int r = r0 + tid / LPR;
        RowPre pre;
        if (!SHORT && r < r1) pre = row_preload<MODE>(a, r);
        // ---- phase 1: stream the block's entries into LDS, 16 B per lane per load ------
        const int base = p0 & ~3;              // 16-B aligned for int32, 32-B for fp64
        const int cnt = p1 - base;
        {
            const v4i *gi = reinterpret_cast<const v4i *>(a.indices + base);
            for (int k = 4 * tid; k < cnt; k += 4 * NT) {
                // matrix entries are read exactly once: non-temporal so that they do not push
                // the re-used x lines out of L2 / Infinity Cache
                const v4i v = nt ? __builtin_nontemporal_load(gi + (k >> 2)) : gi[k >> 2];
                const int s = slot(k);          // k % 4 == 0: the four slots are contiguous
                s_idx[s] = v[0]; s_idx[s + 1] = v[1]; s_idx[s + 2] = v[2]; s_idx[s + 3] = v[3];
            }
            const v2d *gd = reinterpret_cast<const v2d *>(a.data + base);
            for (int k = 2 * tid; k < cnt; k += 2 * NT) {
                const v2d v = nt ? __builtin_nontemporal_load(gd + (k >> 1)) : gd[k >> 1];
                const int s = slot(k);
                s_val[s] = v[0]; s_val[s + 1] = v[1];
            }
        }
        __syncthreads();
        // ---- phase 2: one thread per row, stored order ---------------------------------
        // (blocks of very short rows — prolongation has one entry per row — hold up to
        // ROWBLK_NNZ rows, so a thread may take several, NT apart: still coalesced)
        if constexpr (SHORT) {
            constexpr int U = 8;
            for (int rb = r; rb < r1; rb += U * NT) {
                RowPre q[U];
                int c0[U];
                double v0[U], x0[U], sum[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ru = rb + u * NT;
                    q[u].beg = q[u].end = 0;
                    if (ru < r1) q[u] = row_preload<MODE>(a, ru);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {           // first entry of every row, batched
                    const bool has = q[u].beg < q[u].end;
                    const int sl = slot(has ? q[u].beg - base : 0);
                    c0[u] = has ? s_idx[sl] : 0;
                    v0[u] = has ? s_val[sl] : 0.0;
                }
#pragma unroll
                for (int u = 0; u < U; ++u) x0[u] = a.x[c0[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ru = rb + u * NT;
                    if (ru >= r1) continue;
                    sum[u] = (q[u].beg < q[u].end) ? madd(v0[u], x0[u], 0.0) : 0.0;
                    for (int k = q[u].beg + 1 - base; k < q[u].end - base; ++k) {   // rest, stored order
                        const int sl = slot(k);
                        sum[u] = madd(s_val[sl], a.x[s_idx[sl]], sum[u]);
                    }
                    double unused = 0.0;
                    row_epilogue<MODE>(a, ru, q[u], sum[u], 0.0, unused);
                }
            }
        } else {
            // LPR lanes share a row (LPR = 1: one thread per row, strictly stored order).  With
            // LPR = 4 — operators with long rows, e.g. 27-point stencils, where one thread per
            // row would leave most of the workgroup idle — lane q of a row's quad takes entries
            // q, q+4, q+8, ... in stored order and the four partial sums are added in lane
            // order: a fixed, run-to-run reproducible association, the same in every mode.
            const int sub = tid % LPR;
            const int lane0 = (threadIdx.x & 63) & ~(LPR - 1);     // first lane of the quad
            auto quad_sum = [&](double p) {
                if constexpr (LPR == 1) return p;
                double t = __shfl(p, lane0, 64);
#pragma unroll
                for (int q = 1; q < LPR; ++q) t += __shfl(p, lane0 + q, 64);
                return t;
            };
            while (r < r1) {
                const int beg = pre.beg - base, end = pre.end - base;
                double sum = 0.0, diag = 0.0;
                int c[8];
                double v[8], xv[8];
                for (int k = beg + sub; k < end; k += 8 * LPR) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        const int kk = min(k + j * LPR, end - 1);
                        const int s = slot(kk);
                        c[j] = s_idx[s];
                        v[j] = s_val[s];
                    }
#pragma unroll
                    for (int j = 0; j < 8; ++j) xv[j] = a.x[c[j]];
#pragma unroll
                    for (int j = 0; j < 8; ++j) {
                        if (k + j * LPR < end) {
                            sum = madd(v[j], xv[j], sum);
                            if (NEED_DIAG && c[j] == r) diag += v[j];
                        }
                    }
                }
                sum = quad_sum(sum);
                if constexpr (NEED_DIAG) diag = quad_sum(diag);
                if constexpr (FUSED) {
                    // relax the row, then its residual with the NEW x_i: same entries, same
                    // order, same fma chain as ROW_RESIDUAL would run on the updated vector
                    const double xnew = pre.xv + (pre.bv - sum) / diag;
                    double sum2 = 0.0;
                    if (end - beg <= 8 * LPR) {      // the row's entries are still in registers
#pragma unroll
                        for (int j = 0; j < 8; ++j)
                            if (beg + sub + j * LPR < end) sum2 = madd(v[j], c[j] == r ? xnew : xv[j], sum2);
                    } else {
                        for (int k = beg + sub; k < end; k += 8 * LPR) {
#pragma unroll
                            for (int j = 0; j < 8; ++j) {
                                const int kk = k + j * LPR;
                                if (kk < end) {
                                    const int s = slot(kk);
                                    const int cc = s_idx[s];
                                    sum2 = madd(s_val[s], cc == r ? xnew : a.x[cc], sum2);
                                }
                            }
                        }
                    }
                    sum2 = quad_sum(sum2);
                    const double res = pre.bv - sum2;
                    if (sub == 0) {
                        a.y[r] = xnew;
                        if constexpr (MODE == ROW_GS_RES) a.zero[r] = res;
                        else sq += res * res;
                    }
                } else if (sub == 0) {
                    row_epilogue<MODE>(a, r, pre, sum, diag, sq);
                }
                r += NT / LPR;
                if (r < r1) pre = row_preload<MODE>(a, r);
            }
        }
    } else {
        // ---- one long row: the whole workgroup strides over it (summation order differs
        // from the stored order here; only rows longer than ROWBLK_NNZ take this path) ---
        const int r = r0;
        double part = 0.0, dpart = 0.0;
        for (int p = p0 + tid; p < p1; p += NT) {
            const int c = a.indices[p];
            const double v = a.data[p];
            part = madd(v, a.x[c], part);
            if (NEED_DIAG && c == r) dpart += v;
        }
        const double sum = block_sum(part, s_red);
        double diag = 0.0;
        if (NEED_DIAG) diag = block_sum(dpart, s_red);
        if constexpr (FUSED) {
            const RowPre pre = row_preload<MODE>(a, r);
            __syncthreads();
            if (tid == 0) s_red[0] = pre.xv + (pre.bv - sum) / diag;
            __syncthreads();
            const double xnew = s_red[0];
            double part2 = 0.0;
            for (int p = p0 + tid; p < p1; p += NT) {
                const int c = a.indices[p];
                part2 = madd(a.data[p], c == r ? xnew : a.x[c], part2);
            }
            const double sum2 = block_sum(part2, s_red);
            if (tid == 0) {
                a.y[r] = xnew;
                const double res = pre.bv - sum2;
                if constexpr (MODE == ROW_GS_RES) a.zero[r] = res;
                else sq += res * res;
            }
        } else if (tid == 0) {
            const RowPre pre = row_preload<MODE>(a, r);
            row_epilogue<MODE>(a, r, pre, sum, diag, sq);
        }
    }
    if constexpr (NEED_NORM) {
        const double tot = block_sum(sq, s_red);
        if (tid == 0) a.partials[blk] = tot;
    }
}

// XCD-aware block mapping: hardware deals workgroups round-robin over the 8 XCDs
// (MI355X_MICROARCH.md "Workgroup dispatch"), so workgroups b and b+8 share an L2.  Give
// every XCD one CONTIGUOUS eighth of the row blocks: the x entries a stencil row needs
// from the planes above and below were fetched into the same L2 a few hundred blocks
// earlier.  Bijective for any grid size.  Speed only; results do not depend on it.
__device__ __forceinline__ int xcd_remap(int b, int n) {
    const int q = n >> 3, r = n & 7;
    const int xcd = b & 7, idx = b >> 3;
    return (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + idx;
}

// Grouped variant: inside every window of 8*G consecutive row blocks, XCD k (workgroups with
// id % 8 == k) takes the k-th run of G consecutive blocks, so neighbouring blocks — whose x
// windows overlap by the stencil's in-plane reach — share an L2, while all eight XCDs still
// stream from one 8*G-block region of the matrix.  The ragged tail keeps the identity map.
__device__ __forceinline__ int xcd_remap_grouped(int b, int n, int G) {
    const int span = 8 * G;
    const int base = (b / span) * span;
    if (base + span > n) return b;
    const int r = b - base;
    return base + (r & 7) * G + (r >> 3);
}

template <int MODE, bool NTL, bool SHORT, int LPR>
// (NT, 6): six workgroups per CU is what the 25 KB LDS image allows; keep every variant within
// 80 VGPRs so that registers do not cut that to five (the fused sweeps wanted 82-94).
__global__ __launch_bounds__(NT, 6) void rows_kernel(KArgs a, int blk0, int remap) {
    __shared__ double s_val[LDS_SLOTS];
    __shared__ int s_idx[LDS_SLOTS];
    __shared__ double s_red[NT / 64];
    const int local = remap == 0 ? int(blockIdx.x)
                    : remap == 1 ? xcd_remap(blockIdx.x, gridDim.x)
                                 : xcd_remap_grouped(blockIdx.x, gridDim.x, remap);
    process_block<MODE, NTL, SHORT, LPR>(a, blk0 + local, s_val, s_idx, s_red);
}

// A run of consecutive tiny sets (one row block each), executed back to back by ONE
// workgroup with a workgroup barrier between sets: the level-scheduled lexicographic
// sweep of a 1-D operator is n single-row sets, and the first/last hyperplanes of a 3-D
// grid are tiny too.  Stores of set s are visible to set s+1 through the barrier's
// workgroup-scope fence (same CU, same L1).
template <int MODE>
__global__ __launch_bounds__(NT) void rows_serial_kernel(KArgs a, int blk_begin, int blk_end) {
    __shared__ double s_val[LDS_SLOTS];
    __shared__ int s_idx[LDS_SLOTS];
    __shared__ double s_red[NT / 64];
    for (int blk = blk_begin; blk < blk_end; ++blk) {
        process_block<MODE, false, false, 1>(a, blk, s_val, s_idx, s_red);
        __threadfence_block();
        __syncthreads();
    }
}

// Tuning switches (speed only): OMG_XCD_REMAP=1 enables the XCD-chunked block mapping
// (measured slower than the hardware's round-robin on the 256^3 stencil: 390 vs 374 us for the
// residual), OMG_NT_LOADS=0 disables the non-temporal matrix loads (measured 3-4 % slower).
int launch_flags() {
    static int v = -1;
    if (v < 0) {
        const char *e = getenv("OMG_XCD_REMAP");
        const char *n = getenv("OMG_NT_LOADS");
        v = ((e && e[0] == '1') ? 1 : 0) | ((n && n[0] == '0') ? 0 : 2);
    }
    return v;
}

template <int MODE>
void launch_mode(const DevCsr &A, int64_t blk0, int64_t nblk, const KArgs &k, hipStream_t s) {
    if (nblk <= 0) return;
    const int flags = launch_flags();
    static const int group = [] { const char *e = getenv("OMG_XCD_GROUP"); return e ? atoi(e) : 0; }();
    const int remap = (nblk < 64) ? 0 : (group > 1 ? group : (flags & 1));
    // small operators live in L2 / Infinity Cache across cycles: keep them cacheable
    const bool ntl = (flags & 2) && A.nnz * 12 > (int64_t(192) << 20);
    // several rows per thread (short rows): batched variant, built for the modes such
    // operators are used with; other modes fall back to the generic one-row-at-a-time loop
    constexpr bool HAS_SHORT = (MODE == ROW_SPMV || MODE == ROW_AXPY || MODE == ROW_RESIDUAL);
    const dim3 grid((unsigned)nblk), block(NT);
    if constexpr (HAS_SHORT) {
        if (A.rows_cap > NT) {
            if (ntl) hipLaunchKernelGGL((rows_kernel<MODE, true, true, 1>), grid, block, 0, s, k, (int)blk0, remap);
            else hipLaunchKernelGGL((rows_kernel<MODE, false, true, 1>), grid, block, 0, s, k, (int)blk0, remap);
            OMG_HIP(hipGetLastError());
            return;
        }
    }
    if (A.lanes_per_row == 4) {       // long rows: four lanes per row
        if (ntl) hipLaunchKernelGGL((rows_kernel<MODE, true, false, 4>), grid, block, 0, s, k, (int)blk0, remap);
        else hipLaunchKernelGGL((rows_kernel<MODE, false, false, 4>), grid, block, 0, s, k, (int)blk0, remap);
        OMG_HIP(hipGetLastError());
        return;
    }
    if (ntl) hipLaunchKernelGGL((rows_kernel<MODE, true, false, 1>), grid, block, 0, s, k, (int)blk0, remap);
    else hipLaunchKernelGGL((rows_kernel<MODE, false, false, 1>), grid, block, 0, s, k, (int)blk0, remap);
    OMG_HIP(hipGetLastError());
}

}  // namespace

void launch_rows(const DevCsr &A, int mode, int set, const RowArgs &args, hipStream_t s) {
    if (set < 0) launch_rows_range(A, mode, 0, (int)A.n_sets(), args, s);
    else launch_rows_range(A, mode, set, set + 1, args, s);
}

void launch_rows_range(const DevCsr &A, int mode, int set_begin, int set_end, const RowArgs &args,
                       hipStream_t s) {
    KArgs k;
    k.blk_info = A.blk_rows.p;
    k.indptr = A.indptr.p;
    k.indices = A.indices.p;
    k.data = A.data.p;
    k.x = args.x;
    k.b = args.b;
    k.y = args.y;
    k.partials = args.partials;
    k.zero = args.zero;
    k.ymap = args.ymap;
    k.omega = args.omega;
    OMG_REQUIRE(set_begin >= 0 && set_begin <= set_end && size_t(set_end) <= A.n_sets(),
                "launch_rows: set range out of bounds");
    const int64_t blk0 = A.set_blk[set_begin];
    const int64_t nblk = A.set_blk[set_end] - blk0;
    switch (mode) {
        case ROW_GS_RES: launch_mode<ROW_GS_RES>(A, blk0, nblk, k, s); break;
        case ROW_GS_NORM: launch_mode<ROW_GS_NORM>(A, blk0, nblk, k, s); break;
        case ROW_SPMV: launch_mode<ROW_SPMV>(A, blk0, nblk, k, s); break;
        case ROW_RESIDUAL: launch_mode<ROW_RESIDUAL>(A, blk0, nblk, k, s); break;
        case ROW_RESNORM: launch_mode<ROW_RESNORM>(A, blk0, nblk, k, s); break;
        case ROW_NORM_ONLY: launch_mode<ROW_NORM_ONLY>(A, blk0, nblk, k, s); break;
        case ROW_GS: launch_mode<ROW_GS>(A, blk0, nblk, k, s); break;
        case ROW_JACOBI: launch_mode<ROW_JACOBI>(A, blk0, nblk, k, s); break;
        case ROW_AXPY: launch_mode<ROW_AXPY>(A, blk0, nblk, k, s); break;
        default: throw Error(OMG_ERR_INVALID, "launch_rows: unknown mode");
    }
}

void launch_gs_serial(const DevCsr &A, int set_begin, int set_end, const RowArgs &args,
                      hipStream_t s) {
    KArgs k;
    k.blk_info = A.blk_rows.p;
    k.indptr = A.indptr.p;
    k.indices = A.indices.p;
    k.data = A.data.p;
    k.x = args.x;
    k.b = args.b;
    k.y = args.y;
    k.partials = nullptr;
    k.zero = nullptr;
    k.ymap = nullptr;
    k.omega = args.omega;
    const int b0 = (int)A.set_blk[set_begin], b1 = (int)A.set_blk[set_end];
    if (b1 <= b0) return;
    hipLaunchKernelGGL(rows_serial_kernel<ROW_GS>, dim3(1), dim3(NT), 0, s, k, b0, b1);
    OMG_HIP(hipGetLastError());
}

// ---- small vector kernels --------------------------------------------------------------
namespace {

__global__ __launch_bounds__(1024) void sum_kernel(const double *__restrict__ p, int64_t n,
                                                   double *__restrict__ out, int take_sqrt) {
    __shared__ double s_red[16];
    double acc = 0.0;
    for (int64_t i = threadIdx.x; i < n; i += 1024) acc += p[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += s_red[w];
        *out = take_sqrt ? sqrt(t) : t;
    }
}

__global__ __launch_bounds__(256) void fold_kernel(const double *__restrict__ p, int64_t n,
                                                   double *__restrict__ scratch) {
    __shared__ double s_red[4];
    const int64_t chunk = (n + gridDim.x - 1) / gridDim.x;
    const int64_t lo = blockIdx.x * chunk, hi = min(n, lo + chunk);
    double acc = 0.0;
    for (int64_t i = lo + threadIdx.x; i < hi; i += 256) acc += p[i];
    acc = wave_sum(acc);
    if ((threadIdx.x & 63) == 0) s_red[threadIdx.x >> 6] = acc;
    __syncthreads();
    if (threadIdx.x == 0) scratch[blockIdx.x] = (s_red[0] + s_red[1]) + (s_red[2] + s_red[3]);
}

__global__ void gather_kernel(const double *__restrict__ src, const int32_t *__restrict__ idx,
                              double *__restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = src[idx[i]];
}

__global__ void scatter_kernel(const double *__restrict__ src, const int32_t *__restrict__ idx,
                               double *__restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[idx[i]] = src[i];
}

// out = M v, M row-major rows x n.  One wave per row, 16-byte loads, 4 rows per workgroup.
__global__ __launch_bounds__(256) void dense_gemv_kernel(const double *__restrict__ M,
                                                         const double *__restrict__ v,
                                                         double *__restrict__ out, int64_t rows,
                                                         int64_t n) {
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const double *m = M + row * n;
    double acc = 0.0;
    if ((n & 1) == 0) {
        const double2 *m2 = reinterpret_cast<const double2 *>(m);
        const double2 *v2 = reinterpret_cast<const double2 *>(v);
        for (int64_t j = lane; j < (n >> 1); j += 64) {
            const double2 a = m2[j], b = v2[j];
            acc += a.x * b.x;
            acc += a.y * b.y;
        }
    } else {
        for (int64_t j = lane; j < n; j += 64) acc += m[j] * v[j];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

int grid_for(int64_t n, int threads) {
    int64_t g = (n + threads - 1) / threads;
    if (g > 256 * 8) g = 256 * 8;
    if (g < 1) g = 1;
    return (int)g;
}

}  // namespace

// Deterministic two-stage sum: SUM_FOLD workgroups each add one contiguous chunk in a fixed
// order into scratch[0..SUM_FOLD), then one workgroup adds those.  `partials` must have
// room for n + SUM_FOLD doubles (the scratch lives behind the data).
void launch_sum_impl(double *partials, int64_t n, double *out, int take_sqrt, hipStream_t s) {
    if (n > 8192) {
        double *scratch = partials + n;
        hipLaunchKernelGGL(fold_kernel, dim3(SUM_FOLD), dim3(256), 0, s, partials, n, scratch);
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, scratch, (int64_t)SUM_FOLD, out, take_sqrt);
    } else {
        hipLaunchKernelGGL(sum_kernel, dim3(1), dim3(1024), 0, s, partials, n, out, take_sqrt);
    }
    OMG_HIP(hipGetLastError());
}

void launch_sum(double *partials, int64_t n, double *out, hipStream_t s) {
    launch_sum_impl(partials, n, out, 0, s);
}

void launch_sum_sqrt(double *partials, int64_t n, double *out, hipStream_t s) {
    launch_sum_impl(partials, n, out, 1, s);
}

void launch_gather(const double *src, const int32_t *idx, double *dst, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(gather_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, idx, dst, n);
    OMG_HIP(hipGetLastError());
}

void launch_scatter(const double *src, const int32_t *idx, double *dst, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL(scatter_kernel, dim3(grid_for(n, 256)), dim3(256), 0, s, src, idx, dst, n);
    OMG_HIP(hipGetLastError());
}

void launch_dense_gemv(const double *M, const double *v, double *out, int64_t n, hipStream_t s) {
    launch_dense_gemv_rows(M, v, out, n, n, s);
}

void launch_dense_gemv_rows(const double *M, const double *v, double *out, int64_t rows, int64_t n,
                            hipStream_t s) {
    if (rows <= 0 || n <= 0) return;
    hipLaunchKernelGGL(dense_gemv_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, M, v, out, rows, n);
    OMG_HIP(hipGetLastError());
}

}  // namespace omg
