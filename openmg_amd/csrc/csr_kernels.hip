// CSR row kernels for gfx950 (MI355X).  Bandwidth-bound sparse path: no MFMA.
//
// The operators sit in HBM in the block-coded form of common.h (plain CSR, per-entry
// dictionaries, row patterns, offset patterns + ELL values — a lossless recoding chosen per row
// block at upload).  Two kernels walk it; both take a row's entries in STORED order (the
// summation order of openmg/solvers.py:63-65 and of SciPy's csr_matvec; rows longer than
// ASSOC_LEN as four interleaved chains, common.h), so every coding and either kernel give the
// same bits.
//
// rows_kernel — any block.  One workgroup (256 threads = 4 wave64) owns one ROW BLOCK: a run of
// at most 256 consecutive rows holding at most ROWBLK_NNZ stored entries (partition made at
// setup, setup_host.cpp:make_row_blocks).  Two phases:
//   1. an LDS image of the block's (column, value) pairs is built: plain sides are streamed from
//      HBM with 16-byte-per-lane loads (1 KiB per wave instruction, fully coalesced, each byte
//      of the matrix fetched exactly once), coded sides are expanded through the block's
//      dictionary;
//   2. one thread per row (or four lanes per row) walks its entries in LDS, gathering x from
//      global memory — for stencil-like matrices consecutive lanes touch consecutive x, so
//      each gather is itself a coalesced 512-byte access served by L1/L2.
// rows_pattern_kernel — launches whose blocks are all row-pattern coded with wave-sized
// dictionaries: no LDS image, one thread per row, the pattern's offsets and values as
// wave-uniform operands (see the comment at the kernel).
// The epilogue is selected by MODE (SpMV, residual, residual norm, Gauss-Seidel set
// sweep, weighted Jacobi, y += A x, the fused last-set sweeps, and the transpose scatter).
//
// Everything is a template on the value type V (double: the reference's precision; float:
// BASELINE configs[4]); indices are int32; norms are accumulated in double for either V.
//
// Algorithmic HBM bytes per launch (w = sizeof(V), int32 indices; DESIGN.md "Kernels"):
//   SpMV      (w+4)*nnz + 4*(n+1) + 2*w*n
//   residual / GS set sweep / Jacobi   SpMV + w*n
#include "common.h"

namespace omg {

namespace {

constexpr int NT = ROWBLK_THREADS;
constexpr int T = ROWBLK_NNZ;
// LDS image: entry k (counted from the block's 8-aligned first entry) lives at k + (k >> 5).  One pad slot per 32 entries makes the
// thread-per-row reads conflict-free for EVERY row length up to 32 (row length 8 or 16
// would otherwise hit 4 or 2 banks).  +8 entries of slack for the aligned-down start.
constexpr int LDS_SLOTS = (T + 8) + ((T + 8) >> 5) + 1;

__device__ __forceinline__ int slot(int k) { return k + (k >> 5); }

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v2i __attribute__((ext_vector_type(2)));
// 16-byte vector of values: 2 doubles or 4 floats
template <typename V> struct Vec16;
template <> struct Vec16<double> {
    typedef double type __attribute__((ext_vector_type(2)));
    static constexpr int N = 2;
};
template <> struct Vec16<float> {
    typedef float type __attribute__((ext_vector_type(4)));
    static constexpr int N = 4;
};

template <typename V>
struct KArgs {
    const int32_t *blk_info;   // (first row, first entry, column dict, value dict) per row block, + end sentinel
    const uint8_t *ccode;      // per entry: index into the block's column-offset dictionary
    const uint8_t *vcode;      // per entry: index into the block's value dictionary
    const int32_t *cdict;      // dictionary pools (common.h "Block-dictionary coding")
    const V *vdict;
    const uint8_t *rcode;      // per row: its pattern in the block's row-pattern dictionary
    const int32_t *pidx;       // row-pattern pools: (column - row) offsets, values, pattern starts
    const V *pval;
    const int32_t *pbeg;
    const V *vell;             // block-transposed values of offset-pattern blocks
    const int32_t *indptr;
    const int32_t *indices;
    const V *data;
    const V *x;
    const V *b;
    V *y;
    double *partials;
    V *zero;                   // SpMV: also clear zero[r] (next level's initial iterate); GS_RES: residual
    const int32_t *ymap;       // SpMV only: row r is stored to y[ymap[r]] (NULL: y[r])
    V omega;
    const V *first_diag;       // SpMV with `zero`: see RowArgsT::first_diag
    int first_end;
    int first_jacobi;
};

template <typename S>
__device__ __forceinline__ S wave_sum(S v) {
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
template <typename V>
struct RowPre {
    int beg, end;     // CSR extent of the row (absolute entry offsets)
    int out;          // SpMV: where the result goes (ymap)
    V bv, xv;         // b[r]; x[r] (GS, Jacobi) or y[r] (y += A x)
};

// extent: where the row's entries are — EXT_ROWPTR: indptr; EXT_PATTERN: the block is
// row-pattern coded, beg holds the row's pattern code until row_resolve() turns it into the
// pattern's extent in the LDS image; EXT_NONE: the caller knows (one entry per row).
enum : int { EXT_ROWPTR = 0, EXT_PATTERN = 1, EXT_NONE = 2 };
template <int MODE, typename V>
__device__ __forceinline__ RowPre<V> row_preload(const KArgs<V> &a, int r, int extent = EXT_ROWPTR) {
    RowPre<V> p;
    if (extent == EXT_PATTERN) {
        p.beg = a.rcode[r];
        p.end = 0;
    } else if (extent == EXT_NONE) {
        p.beg = p.end = 0;
    } else {
        p.beg = a.indptr[r];
        p.end = a.indptr[r + 1];
    }
    p.bv = V(0);
    p.xv = V(0);
    if constexpr (MODE != ROW_SPMV && MODE != ROW_AXPY && MODE != ROW_SCATTER) p.bv = a.b[r];
    if constexpr (mode_relaxes(MODE))
        p.xv = a.x[r];
    if constexpr (MODE == ROW_AXPY) p.xv = a.y[r];
    p.out = r;
    if constexpr (MODE == ROW_SPMV) { if (a.ymap) p.out = a.ymap[r]; }
    if constexpr (MODE == ROW_SCATTER) p.xv = a.x[a.ymap ? a.ymap[r] : r];     // the row's multiplier
    return p;
}

template <typename V>
__device__ __forceinline__ void row_resolve(RowPre<V> &p, const int *s_cd) {
    const int code = p.beg;
    p.beg = s_cd[code];
    p.end = s_cd[code + 1];
}

// Row sums are spelled with an explicit fma everywhere so that the fused sweep+residual
// modes reproduce the plain residual kernel bit for bit.
__device__ __forceinline__ double madd(double v, double x, double acc) { return fma(v, x, acc); }
__device__ __forceinline__ float madd(float v, float x, float acc) { return fmaf(v, x, acc); }
// sq += res^2 for the residual norms, spelled as ONE fma everywhere: left to the compiler the
// contraction of "sq += r * r" came out differently in different epilogues (last-bit differences
// between a norm formed by a PRENORM launch and the same norm formed by a NORM_ONLY launch)
template <typename V>
__device__ __forceinline__ void add_square(double &sq, V res) { sq = fma(double(res), double(res), sq); }

// Row-sum association (common.h ASSOC_LEN).  four: the row is long — entry e (counted from the
// row's first) goes to chain e & 3; callers pass e's low bits as a compile-time j.
template <typename V>
struct Chains {
    V s0 = V(0), s1 = V(0), s2 = V(0), s3 = V(0);
    __device__ __forceinline__ void fma_at(int j, bool four, V v, V x) {
        if (four) {
            switch (j & 3) {
                case 0: s0 = madd(v, x, s0); break;
                case 1: s1 = madd(v, x, s1); break;
                case 2: s2 = madd(v, x, s2); break;
                default: s3 = madd(v, x, s3); break;
            }
        } else {
            s0 = madd(v, x, s0);
        }
    }
    __device__ __forceinline__ void add_at(int j, bool four, V v) {
        if (four) {
            switch (j & 3) {
                case 0: s0 += v; break;
                case 1: s1 += v; break;
                case 2: s2 += v; break;
                default: s3 += v; break;
            }
        } else {
            s0 += v;
        }
    }
    __device__ __forceinline__ V total(bool four) const { return four ? ((s0 + s1) + s2) + s3 : s0; }
};

// Modes whose epilogue is ONE store of one value to y[r]: that value (and the row's share of sq).
constexpr bool mode_one_store(int m) {
    return m == ROW_RESIDUAL || m == ROW_RESNORM || m == ROW_GS || m == ROW_JACOBI || m == ROW_GS_PRENORM ||
           m == ROW_JACOBI_PRENORM || m == ROW_AXPY;
}
template <int MODE, typename V>
__device__ __forceinline__ V row_result(const KArgs<V> &a, const RowPre<V> &p, V sum, V diag, double &sq) {
    if constexpr (MODE == ROW_RESIDUAL) {
        return p.bv - sum;
    } else if constexpr (MODE == ROW_RESNORM) {
        const V res = p.bv - sum;
        add_square(sq, res);
        return res;
    } else if constexpr (MODE == ROW_GS) {
        // openmg/solvers.py:68   x[i] = x[i] + (b[i] - Aix) / A[i, i]
        return p.xv + (p.bv - sum) / diag;
    } else if constexpr (MODE == ROW_JACOBI) {
        return p.xv + a.omega * ((p.bv - sum) / diag);
    } else if constexpr (MODE == ROW_GS_PRENORM) {
        const V res = p.bv - sum;                  // the expression ROW_GS divides by the diagonal, and ROW_NORM_ONLY squares
        add_square(sq, res);
        return p.xv + res / diag;
    } else if constexpr (MODE == ROW_JACOBI_PRENORM) {
        const V res = p.bv - sum;
        add_square(sq, res);
        return p.xv + a.omega * (res / diag);
    } else {
        static_assert(MODE == ROW_AXPY, "row_result: mode has no single stored value");
        return p.xv + sum;
    }
}

template <int MODE, typename V>
__device__ __forceinline__ void row_epilogue(const KArgs<V> &a, int r, const RowPre<V> &p, V sum,
                                             V diag, double &sq) {
    if constexpr (MODE == ROW_SPMV) {
        a.y[p.out] = sum;
        if (a.zero) {
            V x0 = V(0);
            if (a.first_diag && p.out < a.first_end) {
                // the next level's first relaxation of a zero iterate, spelled like row_epilogue's
                // ROW_GS / ROW_JACOBI with x_i = 0 and a row sum of +0
                const V q = (sum - V(0)) / a.first_diag[p.out];
                x0 = V(0) + (a.first_jacobi ? a.omega * q : q);
            }
            a.zero[p.out] = x0;
        }
    } else if constexpr (MODE == ROW_NORM_ONLY) {
        const V res = p.bv - sum;
        add_square(sq, res);
    } else if constexpr (mode_one_store(MODE)) {
        a.y[r] = row_result<MODE>(a, p, sum, diag, sq);
    }
}

// SHORT selects the phase-2 variant for operators whose rows are so short (prolongation:
// one entry per row) that a block holds several rows per thread: the rows of a thread are
// then processed eight at a time with all their loads in flight together.
// LONG: the launch may hold rows of more than ASSOC_LEN entries (host: set_maxlen); only then does a
// one-thread-per-row instantiation carry the four accumulators of a long row (registers).
template <int MODE, bool nt, bool SHORT, int LPR, bool LONG, typename V>
__device__ void process_block(const KArgs<V> &a, int blk, V *s_val, int *s_idx, double *s_red, int *s_cd,
                              V *s_vd) {
    constexpr bool FUSED = (MODE == ROW_GS_RES || MODE == ROW_GS_NORM);
    constexpr bool NEED_DIAG = mode_relaxes(MODE);
    constexpr bool NEED_NORM = mode_norm(MODE);
    using vdat = typename Vec16<V>::type;
    constexpr int VN = Vec16<V>::N;
    const int tid = threadIdx.x;
    // (first row, first entry) of this block and of the next: one round trip instead of
    // block table -> indptr
    const v4i *info = reinterpret_cast<const v4i *>(a.blk_info);
    const v4i lo = info[2 * blk], lp = info[2 * blk + 1], hi = info[2 * blk + 2];
    const int r0 = lo[0], p0 = lo[1], r1 = hi[0], p1 = hi[1];
    // row-pattern dictionary: lp = (entry offset, entries, table offset, patterns), patterns 0 = none
    const int npat = lp[3];
    const bool pat = npat != 0;
    // per-entry dictionaries: (pool offset << DICT_SHIFT) | entries, 0 = this block reads the plain
    // array (in a pattern block the two words mean something else: common.h)
    const int cinfo = pat ? 0 : lo[2], vinfo = pat ? 0 : lo[3];
    const bool crel = cinfo != 0 || pat;   // LDS then holds (column - row), not the column
    // plain / per-entry coded block whose rows all hold exactly one entry (prolongation):
    // entry p0 + (row - r0), the row pointers are not read
    const bool unit = SHORT && !pat && lp[2] == 1;
    double sq = 0.0;

    if (pat || p1 - p0 <= T) {             // (a pattern block stages its dictionary, not its entries)
        int r = r0 + tid / LPR;
        RowPre<V> pre;
        if (!SHORT && r < r1) pre = row_preload<MODE>(a, r, pat ? EXT_PATTERN : EXT_ROWPTR);
        // ---- phase 1: the block's entries into LDS --------------------------------------
        // plain side: streamed from HBM, 16 B per lane per load; coded side: 8 one-byte codes
        // per lane per load, expanded through the block's dictionary (held in LDS)
        // (a pattern block copies its dictionary — a few hundred bytes, shared between blocks,
        // L2-resident — instead of streaming entries of its own)
        const int base = pat ? 0 : (p0 & ~7);  // 8-entry aligned: 8 B of codes, 32 B of int32, 64 B of fp64
        const int cnt = pat ? lp[1] : p1 - base;
        if (pat) {
            if (tid <= npat) s_cd[tid] = a.pbeg[lp[2] + tid];
            const int32_t *di = a.pidx + lp[0];
            const V *dv = a.pval + lp[0];
            for (int k = tid; k < cnt; k += NT) {
                const int s = slot(k);
                s_idx[s] = di[k];
                s_val[s] = dv[k];
            }
        }
        const int k0 = 8 * tid;
        v2i ccw = {0, 0}, vcw = {0, 0};
        if (cinfo | vinfo) {
            const int ncd = cinfo & (2 * DICT_MAX - 1), nvd = vinfo & (2 * DICT_MAX - 1);
            if (tid < ncd) s_cd[tid] = a.cdict[(cinfo >> DICT_SHIFT) + tid];
            if (tid < nvd) s_vd[tid] = a.vdict[(vinfo >> DICT_SHIFT) + tid];
            if (cinfo && k0 < cnt) ccw = *reinterpret_cast<const v2i *>(a.ccode + base + k0);
            if (vinfo && k0 < cnt) vcw = *reinterpret_cast<const v2i *>(a.vcode + base + k0);
        }
        if (!cinfo && !pat) {
            const v4i *gi = reinterpret_cast<const v4i *>(a.indices + base);
            for (int k = 4 * tid; k < cnt; k += 4 * NT) {
                // matrix entries are read exactly once: non-temporal so that they do not push
                // the re-used x lines out of L2 / Infinity Cache
                const v4i v = nt ? __builtin_nontemporal_load(gi + (k >> 2)) : gi[k >> 2];
                const int s = slot(k);          // k % 4 == 0: the four slots are contiguous
                s_idx[s] = v[0]; s_idx[s + 1] = v[1]; s_idx[s + 2] = v[2]; s_idx[s + 3] = v[3];
            }
        }
        if (!vinfo && !pat) {
            const vdat *gd = reinterpret_cast<const vdat *>(a.data + base);
            for (int k = VN * tid; k < cnt; k += VN * NT) {
                const vdat v = nt ? __builtin_nontemporal_load(gd + (k / VN)) : gd[k / VN];
                const int s = slot(k);          // k % VN == 0 and VN <= 4: contiguous slots
#pragma unroll
                for (int j = 0; j < VN; ++j) s_val[s + j] = v[j];
            }
        }
        if (cinfo | vinfo) {
            __syncthreads();                    // dictionaries are in LDS
            // entries base+k .. base+k+7 sit in one 32-entry group: contiguous slots
            if (cinfo) {
                for (int k = k0; k < cnt; k += 8 * NT) {
                    if (k != k0) ccw = *reinterpret_cast<const v2i *>(a.ccode + base + k);
                    const int s = slot(k);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        s_idx[s + j] = s_cd[(ccw[j >> 2] >> (8 * (j & 3))) & (DICT_MAX - 1)];
                }
            }
            if (vinfo) {
                for (int k = k0; k < cnt; k += 8 * NT) {
                    if (k != k0) vcw = *reinterpret_cast<const v2i *>(a.vcode + base + k);
                    const int s = slot(k);
#pragma unroll
                    for (int j = 0; j < 8; ++j)
                        s_val[s + j] = s_vd[(vcw[j >> 2] >> (8 * (j & 3))) & (DICT_MAX - 1)];
                }
            }
        }
        __syncthreads();
        if (!SHORT && pat && r < r1) row_resolve(pre, s_cd);
        // ---- phase 2: one thread per row, stored order ---------------------------------
        // (blocks of very short rows — prolongation has one entry per row — hold up to
        // ROWBLK_NNZ rows, so a thread may take several, NT apart: still coalesced)
        if constexpr (SHORT) {
            constexpr int U = 8;
            for (int rb = r; rb < r1; rb += U * NT) {
                RowPre<V> q[U];
                int c0[U];
                V v0[U], x0[U], sum[U];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ru = rb + u * NT;
                    q[u].beg = q[u].end = 0;
                    if (ru < r1) {
                        q[u] = row_preload<MODE>(a, ru, unit ? EXT_NONE : EXT_ROWPTR);
                        if (unit) {                           // one entry per row: no row pointers read
                            q[u].beg = p0 + (ru - r0);
                            q[u].end = q[u].beg + 1;
                        }
                    }
                }
#pragma unroll
                for (int u = 0; u < U; ++u) {           // first entry of every row, batched
                    const bool has = q[u].beg < q[u].end;
                    const int sl = slot(has ? q[u].beg - base : 0);
                    c0[u] = has ? s_idx[sl] + (crel ? rb + u * NT : 0) : 0;
                    v0[u] = has ? s_val[sl] : V(0);
                }
#pragma unroll
                for (int u = 0; u < U; ++u) x0[u] = a.x[c0[u]];
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const int ru = rb + u * NT;
                    if (ru >= r1) continue;
                    sum[u] = (q[u].beg < q[u].end) ? madd(v0[u], x0[u], V(0)) : V(0);
                    if (q[u].end - q[u].beg <= ASSOC_LEN) {
                        for (int k = q[u].beg + 1 - base; k < q[u].end - base; ++k) {   // rest, stored order
                            const int sl = slot(k);
                            sum[u] = madd(s_val[sl], a.x[s_idx[sl] + (crel ? ru : 0)], sum[u]);
                        }
                    } else {
                        // a long row among the short ones: four chains (common.h ASSOC_LEN); the
                        // batched first entry started chain 0
                        V ch[4] = {sum[u], V(0), V(0), V(0)};
#pragma unroll
                        for (int t = 0; t < 4; ++t)
                            for (int k = q[u].beg - base + (t == 0 ? 4 : t); k < q[u].end - base; k += 4) {
                                const int sl = slot(k);
                                ch[t] = madd(s_val[sl], a.x[s_idx[sl] + (crel ? ru : 0)], ch[t]);
                            }
                        sum[u] = ((ch[0] + ch[1]) + ch[2]) + ch[3];
                    }
                    double unused = 0.0;
                    row_epilogue<MODE>(a, ru, q[u], sum[u], V(0), unused);
                }
            }
        } else {
            // LPR lanes share a row.  The association of a row's sum is fixed by its length
            // (common.h ASSOC_LEN), not by LPR: a row of <= 16 entries is ONE chain in stored
            // order (LPR = 4: the quad's first lane walks it alone), a longer row FOUR chains —
            // chain q takes entries q, q+4, ... — which with LPR = 4 are the quad's four lanes
            // and with LPR = 1 four accumulators of the row's thread; added as ((s0+s1)+s2)+s3.
            const int sub = tid % LPR;
            const int lane0 = (threadIdx.x & 63) & ~(LPR - 1);     // first lane of the quad
            while (r < r1) {
                const int beg = pre.beg - base, end = pre.end - base;
                const int radd = crel ? r : 0;
                const bool four = end - beg > ASSOC_LEN;
                const bool four_here = LPR == 1 && LONG && four;  // the four chains live in this thread
                const int step = (LPR == 4 && four) ? 4 : 1;
                const int first = (LPR == 4 && four) ? sub : 0;
                const bool live = LPR == 1 || four || sub == 0;
                // a quad's partial sums -> the row's sum, in lane order (or lane 0's chain alone)
                auto fold = [&](const Chains<V> &ch) {
                    if constexpr (LPR == 1) {
                        return ch.total(four);
                    } else {
                        V t = __shfl(ch.s0, lane0, 64);
                        if (four) {
#pragma unroll
                            for (int q = 1; q < LPR; ++q) t += __shfl(ch.s0, lane0 + q, 64);
                        }
                        return t;
                    }
                };
                Chains<V> acc, dacc;
                int c[8];
                V v[8], xv[8];
                if (live) {
                    for (int k = beg + first; k < end; k += 8 * step) {   // (k - beg - first) % (8 step) == 0
#pragma unroll
                        for (int j = 0; j < 8; ++j) {
                            const int kk = min(k + j * step, end - 1);
                            const int s = slot(kk);
                            c[j] = s_idx[s] + radd;
                            v[j] = s_val[s];
                        }
#pragma unroll
                        for (int j = 0; j < 8; ++j) xv[j] = a.x[c[j]];
                        if (four_here) {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (k + j * step < end) {
                                    acc.fma_at(j, true, v[j], xv[j]);
                                    if (NEED_DIAG && c[j] == r) dacc.add_at(j, true, v[j]);
                                }
                        } else {
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (k + j * step < end) {
                                    acc.s0 = madd(v[j], xv[j], acc.s0);
                                    if (NEED_DIAG && c[j] == r) dacc.s0 += v[j];
                                }
                        }
                    }
                }
                const V sum = fold(acc);
                V diag = V(0);
                if constexpr (NEED_DIAG) diag = fold(dacc);
                if constexpr (FUSED) {
                    // relax the row, then its residual with the NEW x_i: same entries, same
                    // chains as ROW_RESIDUAL would run on the updated vector
                    const V xnew = pre.xv + (pre.bv - sum) / diag;
                    Chains<V> acc2;
                    if (live) {
                        if (end - beg <= 8 * step) {      // the lane's entries are still in registers: one chain here
#pragma unroll
                            for (int j = 0; j < 8; ++j)
                                if (beg + first + j * step < end) acc2.s0 = madd(v[j], c[j] == r ? xnew : xv[j], acc2.s0);
                        } else {
                            for (int k = beg + first; k < end; k += 8 * step) {
#pragma unroll
                                for (int j = 0; j < 8; ++j) {
                                    const int kk = k + j * step;
                                    if (kk < end) {
                                        const int s = slot(kk);
                                        const int cc = s_idx[s] + radd;
                                        acc2.fma_at(j, four_here, s_val[s], cc == r ? xnew : a.x[cc]);
                                    }
                                }
                            }
                        }
                    }
                    const V sum2 = fold(acc2);
                    const V res = pre.bv - sum2;
                    if (sub == 0) {
                        a.y[r] = xnew;
                        if constexpr (MODE == ROW_GS_RES) a.zero[r] = res;
                        else add_square(sq, res);
                    }
                } else if (sub == 0) {
                    row_epilogue<MODE>(a, r, pre, sum, diag, sq);
                }
                r += NT / LPR;
                if (r < r1) {
                    pre = row_preload<MODE>(a, r, pat ? EXT_PATTERN : EXT_ROWPTR);
                    if (pat) row_resolve(pre, s_cd);
                }
            }
        }
    } else {
        // ---- one long row: the whole workgroup strides over it (summation order differs
        // from the stored order here; only rows longer than ROWBLK_NNZ take this path; the
        // per-thread partial sums are combined in double for either V) --------------------
        const int r = r0;
        V part = V(0), dpart = V(0);
        for (int p = p0 + tid; p < p1; p += NT) {
            const int c = a.indices[p];
            const V v = a.data[p];
            part = madd(v, a.x[c], part);
            if (NEED_DIAG && c == r) dpart += v;
        }
        const V sum = V(block_sum(double(part), s_red));
        V diag = V(0);
        if (NEED_DIAG) diag = V(block_sum(double(dpart), s_red));
        if constexpr (FUSED) {
            const RowPre<V> pre = row_preload<MODE>(a, r);
            __syncthreads();
            if (tid == 0) s_red[0] = double(pre.xv + (pre.bv - sum) / diag);
            __syncthreads();
            const V xnew = V(s_red[0]);
            V part2 = V(0);
            for (int p = p0 + tid; p < p1; p += NT) {
                const int c = a.indices[p];
                part2 = madd(a.data[p], c == r ? xnew : a.x[c], part2);
            }
            const V sum2 = V(block_sum(double(part2), s_red));
            if (tid == 0) {
                a.y[r] = xnew;
                const V res = pre.bv - sum2;
                if constexpr (MODE == ROW_GS_RES) a.zero[r] = res;
                else add_square(sq, res);
            }
        } else if (tid == 0) {
            const RowPre<V> pre = row_preload<MODE>(a, r);
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

// (NT, 6): six workgroups per CU is what the 25 KB fp64 LDS image allows; keep every variant
// within 80 VGPRs so that registers do not cut that to five (the fused sweeps wanted 82-94).
template <int MODE, bool NTL, bool SHORT, int LPR, bool LONG, typename V>
__global__ __launch_bounds__(NT, 6) void rows_kernel(KArgs<V> a, int blk0, int remap) {
    __shared__ V s_val[LDS_SLOTS];
    __shared__ int s_idx[LDS_SLOTS];
    __shared__ double s_red[NT / 64];
    __shared__ int s_cd[2 * DICT_MAX];     // column dictionary, or the npat + 1 pattern starts
    __shared__ V s_vd[DICT_MAX];
    const int local = remap == 0 ? int(blockIdx.x)
                    : remap == 1 ? xcd_remap(blockIdx.x, gridDim.x)
                                 : xcd_remap_grouped(blockIdx.x, gridDim.x, remap);
    process_block<MODE, NTL, SHORT, LPR, LONG>(a, blk0 + local, s_val, s_idx, s_red, s_cd, s_vd);
}

// A run of consecutive tiny sets (one row block each), executed back to back by ONE
// workgroup with a workgroup barrier between sets: the level-scheduled lexicographic
// sweep of a 1-D operator is n single-row sets, and the first/last hyperplanes of a 3-D
// grid are tiny too.  Stores of set s are visible to set s+1 through the barrier's
// workgroup-scope fence (same CU, same L1).
template <int MODE, int LPR, bool LONG, typename V>
__global__ __launch_bounds__(NT) void rows_serial_kernel(KArgs<V> a, int blk_begin, int blk_end) {
    __shared__ V s_val[LDS_SLOTS];
    __shared__ int s_idx[LDS_SLOTS];
    __shared__ double s_red[NT / 64];
    __shared__ int s_cd[2 * DICT_MAX];
    __shared__ V s_vd[DICT_MAX];
    for (int blk = blk_begin; blk < blk_end; ++blk) {
        process_block<MODE, false, false, LPR, LONG>(a, blk, s_val, s_idx, s_red, s_cd, s_vd);
        __threadfence_block();
        __syncthreads();
    }
}

// ---- pure row-pattern launches -------------------------------------------------------------
// When every row block of a launch is row-pattern coded (common.h), the matrix arrives as one
// byte per row plus a dictionary that is the same for (nearly) all blocks.  This kernel then
// skips LDS altogether: one thread per row; the lanes of a wave are grouped by pattern
// (interior waves: a single group) and each group walks ITS pattern with wave-uniform
// operands — offsets and values sit in SGPRs, every length test is a scalar branch — so a
// stored entry costs one address add, one gather of x and one fma.  Entries are taken in
// stored order with the same fma chain as rows_kernel: identical bits.
//
// The dependent memory round trips per block are what bound it, so the dictionary does not
// wait for the row codes: a dictionary of at most 64 entries is fetched with ONE vector load
// (lane i holds entry i) next to the codes, b and x_i, and the wave-uniform operands are then
// picked out of those registers with v_readlane.  (Blocks with larger dictionaries — e.g. the
// hyperplane sets of a lexicographic ordering — go through rows_kernel's LDS image: a
// scalar-load variant of this kernel was measured at half its speed.)
// No LDS image and <= 64 VGPRs: eight workgroups per CU.
__device__ __forceinline__ int lane_pick(int v, int lane) { return __builtin_amdgcn_readlane(v, lane); }
__device__ __forceinline__ float lane_pick(float v, int lane) {
    return __int_as_float(__builtin_amdgcn_readlane(__float_as_int(v), lane));
}
__device__ __forceinline__ double lane_pick(double v, int lane) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), lane);
    const int hi = __builtin_amdgcn_readlane(__double2hiint(v), lane);
    return __hiloint2double(hi, lo);
}

// Dictionary of one wave's block: lane i of didx[h] / dval[h] holds entry 64 h + i, lane c of
// dbeg the start of pattern c (a pattern-kernel block has at most PAT_LANE_ENTRIES entries in at
// most 63 patterns); the wave-uniform operands come out with v_readlane.  ell != NULL: the block
// keeps offset patterns only and its values sit transposed in the ELL array — entry j of row i
// of the block at ell[j * rows + i], read coalesced by the one-thread-per-row kernel.
template <bool BIG, typename V>
struct PatDictRef {
    int didx[2], dbeg;
    V dval[2];
    const V *__restrict__ ell;     // + row in block
    int rows;                      // rows of the block (ELL stride)
    // BIG: more than 64 entries, the second register is in use (block-uniform; a template
    // parameter so that the common small dictionary pays no test per operand)
    __device__ __forceinline__ int start(int code) const { return lane_pick(dbeg, code); }
    __device__ __forceinline__ int off(int k) const {
        if (!BIG) return lane_pick(didx[0], k);
        return k < 64 ? lane_pick(didx[0], k) : lane_pick(didx[1], k - 64);
    }
    __device__ __forceinline__ V val(int k) const {
        if (!BIG) return lane_pick(dval[0], k);
        return k < 64 ? lane_pick(dval[0], k) : lane_pick(dval[1], k - 64);
    }
};

// N consecutive pattern entries starting at k (everything about them is wave-uniform: no
// clamps, no predicates): operands, then all gathers, then the fma chain in stored order.
// FUSED and `whole` (the chunk is the entire row): also the relaxed value and the row's
// residual with it, from the operands still in registers.
template <int MODE, int N, bool BIG, typename V>
__device__ __forceinline__ void pattern_chunk(const PatDictRef<BIG, V> &d, const V *__restrict__ xrow, int k, int pb,
                                              bool whole, bool four, const RowPre<V> &pre, Chains<V> &acc,
                                              Chains<V> &dacc, V &sum2, V &xnew) {
    constexpr bool FUSED = (MODE == ROW_GS_RES || MODE == ROW_GS_NORM);
    constexpr bool NEED_DIAG = mode_relaxes(MODE);
    int off[N];
    V val[N], xg[N];
#pragma unroll
    for (int j = 0; j < N; ++j) off[j] = d.off(k + j);
    if (d.ell) {                                              // block-uniform
#pragma unroll
        for (int j = 0; j < N; ++j) val[j] = d.ell[(k - pb + j) * d.rows];
    } else {
#pragma unroll
        for (int j = 0; j < N; ++j) val[j] = d.val(k + j);
    }
    if constexpr (MODE == ROW_SCATTER) {
        // y[r + off] += val * x_row: product rounded, then added — the two roundings of
        // ROW_AXPY over the explicit transpose (y + fma(val, x, 0)), so the same bits
        V *__restrict__ ycol = const_cast<V *>(xrow);         // xrow = y + r here (pattern_rows)
#pragma unroll
        for (int j = 0; j < N; ++j) xg[j] = ycol[off[j]];
#pragma unroll
        for (int j = 0; j < N; ++j) ycol[off[j]] = xg[j] + madd(val[j], pre.xv, V(0));
        return;
    }
#pragma unroll
    for (int j = 0; j < N; ++j) xg[j] = xrow[off[j]];
#pragma unroll
    for (int j = 0; j < N; ++j) {                             // (k - pb) % 8 == 0: entry k + j feeds chain j & 3
        acc.fma_at(j, four, val[j], xg[j]);
        if (NEED_DIAG && off[j] == 0) dacc.add_at(j, four, val[j]);
    }
    if constexpr (FUSED) {
        if (whole) {                                          // <= 8 entries: one chain
            xnew = pre.xv + (pre.bv - acc.s0) / dacc.s0;
#pragma unroll
            for (int j = 0; j < N; ++j) sum2 = madd(val[j], off[j] == 0 ? xnew : xg[j], sum2);
        }
    }
}

// One wave's rows of a pattern block: lanes grouped by pattern, one group at a time.
template <int MODE, bool LONG, bool BIG, typename V>
__device__ __forceinline__ void pattern_rows(const KArgs<V> &a, int r, bool active, int code, const RowPre<V> &pre,
                                             const PatDictRef<BIG, V> &d, V &sum, V &diag, V &sum2, V &xnew) {
    constexpr bool FUSED = (MODE == ROW_GS_RES || MODE == ROW_GS_NORM);
    // x[r + offset]; ROW_SCATTER walks (and updates) y there instead
    const V *__restrict__ xrow = (MODE == ROW_SCATTER ? a.y : a.x) + r;
    unsigned long long todo = __ballot(active);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int ucode = __builtin_amdgcn_readlane(code, leader);      // wave-uniform from here on
        const bool mine = active && code == ucode;
        const int pb = d.start(ucode), pe = d.start(ucode + 1);
        if (mine) {
            const bool whole = pe - pb <= 8;
            // common.h: long rows sum as four chains (LONG: the launch holds such rows at all;
            // the lean instantiation keeps a single accumulator)
            const bool four = LONG && pe - pb > ASSOC_LEN;
            Chains<V> acc, dacc;
            int k = pb;
            for (; k + 8 <= pe; k += 8) pattern_chunk<MODE, 8>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew);
            switch (pe - k) {                                 // uniform: one scalar jump
                case 1: pattern_chunk<MODE, 1>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew); break;
                case 2: pattern_chunk<MODE, 2>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew); break;
                case 3: pattern_chunk<MODE, 3>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew); break;
                case 4: pattern_chunk<MODE, 4>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew); break;
                case 5: pattern_chunk<MODE, 5>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew); break;
                case 6: pattern_chunk<MODE, 6>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew); break;
                case 7: pattern_chunk<MODE, 7>(d, xrow, k, pb, whole, four, pre, acc, dacc, sum2, xnew); break;
                default: break;
            }
            sum = acc.total(four);
            diag = dacc.total(four);
            if constexpr (FUSED) {
                if (!whole) {
                    // longer rows: relax, then walk the pattern again (rows_kernel's FUSED branch),
                    // chain by chain
                    xnew = pre.xv + (pre.bv - sum) / diag;
                    V ch[4] = {V(0), V(0), V(0), V(0)};
                    const int nch = four ? 4 : 1;
#pragma unroll
                    for (int t = 0; t < 4; ++t) {
                        if (t >= nch) break;
                        for (int q = pb + t; q < pe; q += nch) {
                            const int o = d.off(q);
                            const V v = d.ell ? d.ell[(q - pb) * d.rows] : d.val(q);
                            ch[t] = madd(v, o == 0 ? xnew : xrow[o], ch[t]);
                        }
                    }
                    sum2 = four ? ((ch[0] + ch[1]) + ch[2]) + ch[3] : ch[0];
                }
            }
        }
        todo &= ~__ballot(mine);
    }
}

// (Variants measured and dropped: several blocks per workgroup with all their operands
// requested together — no gain; XCD-chunked block mapping — 6 % slower, as for rows_kernel.)
// ELL: the launch may hold blocks whose values sit in the ELL array (host: set_ell)
template <int MODE, bool LONG, bool ELL, typename V>
__global__ __launch_bounds__(NT, (MODE == ROW_GS_RES || MODE == ROW_GS_NORM) ? (LONG ? 4 : 6)
                                                                              : (MODE == ROW_GS || MODE == ROW_JACOBI || mode_prenorm(MODE)) ? 7 : 8)
void rows_pattern_kernel(KArgs<V> a, int blk0) {
    constexpr bool FUSED = (MODE == ROW_GS_RES || MODE == ROW_GS_NORM);
    constexpr bool NEED_NORM = mode_norm(MODE);
    __shared__ double s_red[NT / 64];
    const int blk = blk0 + int(blockIdx.x);
    const v4i *info = reinterpret_cast<const v4i *>(a.blk_info);
    const v4i lo = info[2 * blk], lp = info[2 * blk + 1], hi = info[2 * blk + 2];
    const int r0 = lo[0], r1 = hi[0];
    const int32_t *__restrict__ pidx = a.pidx + lp[0];       // this block's dictionary
    const V *__restrict__ pval = a.pval + lp[0];
    const int32_t *__restrict__ pbeg = a.pbeg + lp[2];
    const int cnt = lp[1], npat = lp[3];                      // cnt <= PAT_LANE_ENTRIES, npat <= 63 (setup: set_pattern)
    const int r = r0 + int(threadIdx.x);                      // a pattern block has <= NT rows
    const bool active = r < r1;
    const int lane = int(threadIdx.x) & 63;
    double sq = 0.0;
    RowPre<V> pre;
    int code = 0;
    // the dictionary with the codes, not after them
    const V *ell = (ELL && lo[2]) ? a.vell + (lo[2] - 1) + int(threadIdx.x) : nullptr;   // offset patterns, values in the ELL array
    const int didx0 = pidx[min(lane, cnt - 1)];
    const int dbeg = pbeg[min(lane, npat)];
    V dval0 = V(0);
    if (!ell) dval0 = pval[min(lane, cnt - 1)];
    if (active) {
        pre = row_preload<MODE>(a, r, EXT_PATTERN);
        code = pre.beg;
    }
    V sum = V(0), diag = V(0), sum2 = V(0), xnew = V(0);
    if (cnt > 64) {                                           // block-uniform: second dictionary register
        PatDictRef<true, V> d;
        d.didx[0] = didx0; d.dbeg = dbeg; d.dval[0] = dval0; d.ell = ell; d.rows = r1 - r0;
        d.didx[1] = pidx[min(64 + lane, cnt - 1)];
        d.dval[1] = ell ? V(0) : pval[min(64 + lane, cnt - 1)];
        pattern_rows<MODE, LONG>(a, r, active, code, pre, d, sum, diag, sum2, xnew);
    } else {
        PatDictRef<false, V> d;
        d.didx[0] = didx0; d.dbeg = dbeg; d.dval[0] = dval0; d.ell = ell; d.rows = r1 - r0;
        d.didx[1] = 0; d.dval[1] = V(0);
        pattern_rows<MODE, LONG>(a, r, active, code, pre, d, sum, diag, sum2, xnew);
    }
    if (active) {
        if constexpr (FUSED) {
            const V res = pre.bv - sum2;
            a.y[r] = xnew;
            if constexpr (MODE == ROW_GS_RES) a.zero[r] = res;
            else add_square(sq, res);
        } else if constexpr (MODE != ROW_SCATTER) {
            row_epilogue<MODE>(a, r, pre, sum, diag, sq);
        }
    }
    if constexpr (NEED_NORM) {
        const double tot = block_sum(sq, s_red);
        if (threadIdx.x == 0) a.partials[blk] = tot;
    }
}

// ---- union walk: several rows per thread, one pass per wave ----------------------------------
// Blocks whose dictionary carries a UNION (common.h UNION_MAX: every row pattern of the block is a
// subsequence of one short sequence of (offset, value) pairs, e.g. a stencil's boundary rows and
// the two parities of a red-black ordering).  One workgroup = one block of up to U x 256 rows, a
// thread owns rows r0 + tid + 256 u.  The dependent memory round trips per wave are what bounds
// the one-row-per-thread kernel above (block record -> codes -> gathers, with the lanes taken one
// pattern group at a time); here a wave
//   1. reads the block record and the union (offsets, values: lane j holds slot j; masks: lane c
//      holds pattern c) — shared by all blocks of a launch, L2-resident;
//   2. requests, for ALL its rows at once, the row code, b, x_i, then — masks looked up with one
//      ds_bpermute per row — the gathers of every union slot a row owns: x[r + off_j] through a
//      buffer descriptor (32-bit offset r + off_j in one VALU add, no 64-bit address arithmetic),
//      U x (entries + 2) loads in flight per lane and no grouping of the lanes by pattern;
//   3. walks the union once with SCALAR operands (v_readlane), each lane skipping the slots its
//      pattern's mask lacks: the row's entries in stored order, one fma chain — the bits of
//      rows_kernel / rows_pattern_kernel (rows here hold at most UNION_MAX <= ASSOC_LEN entries).
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
// two adjacent elements of a vector as one access, aligned like ONE element
typedef double pair_d_raw __attribute__((ext_vector_type(2)));
typedef float pair_f_raw __attribute__((ext_vector_type(2)));
typedef pair_d_raw pair_d __attribute__((aligned(8)));
typedef pair_f_raw pair_f __attribute__((aligned(4)));
template <typename V> struct pair_of;
template <> struct pair_of<double> { typedef pair_d type; };
template <> struct pair_of<float> { typedef pair_f type; };
template <typename V> using pair_t = typename pair_of<V>::type;
template <typename V>
__device__ __forceinline__ void store_pair(V *p, V lo, V hi) {
    pair_t<V> q;
    q.x = lo;
    q.y = hi;
    *reinterpret_cast<pair_t<V> *>(p) = q;
}
__device__ __forceinline__ double buffer_gather(__amdgpu_buffer_rsrc_t rs, int byte_off, double) {
    return __builtin_bit_cast(double, __builtin_amdgcn_raw_buffer_load_b64(rs, byte_off, 0, 0));
}
__device__ __forceinline__ float buffer_gather(__amdgpu_buffer_rsrc_t rs, int byte_off, float) {
    return __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
}
// two consecutive elements in ONE load: the vector L1 serves 16 bytes per lane at twice the rate of
// 8 (MI355X_MICROARCH.md: "8-B accesses 0.54-0.70x the 16-B rate"), and these launches are short of it
__device__ __forceinline__ void buffer_gather2(__amdgpu_buffer_rsrc_t rs, int byte_off, double &lo, double &hi) {
    const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, byte_off, 0, 0);
    lo = __builtin_bit_cast(double, v2u{q.x, q.y});
    hi = __builtin_bit_cast(double, v2u{q.z, q.w});
}
// (float rows are not paired: the paired fp32 variant was no faster — 2915 against 2890 V-cycles/s — and
// did not reproduce the other kernels' bits in tests/test_gpu_parity.py's fp32 format case.  The
// hardware is not the reason: tools/align_probe.hip shows 8- and 16-byte buffer and global loads
// served correctly at any 4-byte aligned offset.)
__device__ __forceinline__ void buffer_gather2(__amdgpu_buffer_rsrc_t rs, int byte_off, float &lo, float &hi) {
    lo = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off, 0, 0));
    hi = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(rs, byte_off + 4, 0, 0));
}

template <int MODE, int U, int UMAX, typename V>
__global__ __launch_bounds__(NT) void rows_union_kernel(KArgs<V> a, int blk0, unsigned x_bytes) {
    constexpr bool FUSED = (MODE == ROW_GS_RES || MODE == ROW_GS_NORM);
    constexpr bool NEED_DIAG = mode_relaxes(MODE);
    constexpr bool NEED_NORM = mode_norm(MODE);
    __shared__ double s_red[NT / 64];
    const int blk = blk0 + int(blockIdx.x);
    const v4i *info = reinterpret_cast<const v4i *>(a.blk_info);
    const v4i lo = info[2 * blk], lp = info[2 * blk + 1], hi = info[2 * blk + 2];
    const int r0 = lo[0], r1 = hi[0];
    const int ul = lo[3];                                       // 1 <= ul <= UMAX (host: set_union, union_max)
    const int cnt = lp[1], npat = lp[3];
    const int32_t *__restrict__ uidx = a.pidx + lp[0] + cnt;    // the union sits behind the pattern entries
    const V *__restrict__ uval = a.pval + lp[0] + cnt;
    const int32_t *__restrict__ pmask = a.pbeg + lp[2] + npat + 1;   // the masks behind the pattern starts
    const int lane = int(threadIdx.x) & 63;
    const int uo = uidx[min(lane, ul - 1)];
    const V uv = uval[min(lane, ul - 1)];
    const int mk = pmask[min(lane, npat - 1)];
    // The gathered vector as a buffer: out-of-range offsets (negative ones wrap to huge) read 0.
    // A thread's rows come in ADJACENT pairs (U even) and a pair's operands of one slot, x[c] and
    // x[c + 1], in one load; the descriptor starts PAIR_PAD elements in front of the vector and ends as
    // many behind it (every device vector is a DevBuf: 64 bytes of slack on both sides), so that a
    // pair whose first or second member alone leaves the vector by one element is still served.
    constexpr bool PAIRS = U % 2 == 0 && sizeof(V) == 8;
    constexpr int PAIR_PAD = PAIRS ? 2 : 0;
    const __amdgpu_buffer_rsrc_t xs = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<V *>(a.x) - PAIR_PAD, 0, x_bytes + 2 * PAIR_PAD * unsigned(sizeof(V)), 0x00020000);

    int row[U], code[U];
    bool act[U];
    RowPre<V> pre[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        row[u] = PAIRS ? r0 + 2 * (int(threadIdx.x) + (u >> 1) * NT) + (u & 1) : r0 + int(threadIdx.x) + u * NT;
        act[u] = row[u] < r1;
        code[u] = 0;
        pre[u].beg = pre[u].end = 0; pre[u].out = row[u]; pre[u].bv = V(0); pre[u].xv = V(0);
        if constexpr (!PAIRS) {
            if (act[u]) {
                pre[u] = row_preload<MODE>(a, row[u], EXT_PATTERN);
                code[u] = pre[u].beg;
            }
        }
    }
    if constexpr (PAIRS) {
        // row_preload for a pair of adjacent rows: b, x_i (or y_i) as ONE 16-byte load each (the second
        // element is read even when the pair's second row is beyond the block: it exists — the next
        // block's row, or the slack behind the vector — and is not used)
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            if (act[u]) {
                const int r = row[u];
                code[u] = a.rcode[r];
                if (act[u + 1]) code[u + 1] = a.rcode[r + 1];
                if constexpr (MODE != ROW_SPMV && MODE != ROW_AXPY && MODE != ROW_SCATTER) {
                    const pair_t<V> q = *reinterpret_cast<const pair_t<V> *>(a.b + r);
                    pre[u].bv = q.x; pre[u + 1].bv = q.y;
                }
                if constexpr (mode_relaxes(MODE)) {
                    const pair_t<V> q = *reinterpret_cast<const pair_t<V> *>(a.x + r);
                    pre[u].xv = q.x; pre[u + 1].xv = q.y;
                }
                if constexpr (MODE == ROW_AXPY) {
                    const pair_t<V> q = *reinterpret_cast<const pair_t<V> *>(a.y + r);
                    pre[u].xv = q.x; pre[u + 1].xv = q.y;
                }
                if constexpr (MODE == ROW_SPMV) {
                    if (a.ymap) { pre[u].out = a.ymap[r]; if (act[u + 1]) pre[u + 1].out = a.ymap[r + 1]; }
                }
            }
        }
    }
    // the pattern masks: lane c holds pattern c's.  ds_bpermute outside any divergent branch — an
    // inactive SOURCE lane would read as 0 (a block's last wave may have fewer live rows than
    // the block has patterns)
    int mask[U];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int m = __builtin_amdgcn_ds_bpermute(code[u] << 2, mk);
        mask[u] = act[u] ? m : 0;
    }
    // gathers of every slot a row owns, all rows of the thread, before any is used.  (Requesting
    // every union slot for every lane without waiting for the masks — offsets that leave the
    // vector read 0 by the descriptor's range check — shortens the dependent chain by one round
    // trip but moves more bytes through the vector L1, which is what these launches are short
    // of: measured 3-8 % slower at 256^3.)
    V xg[U][UMAX];
#pragma unroll
    for (int j = 0; j < UMAX; ++j) {
        if (j < ul) {                                           // block-uniform
            const int oj = __builtin_amdgcn_readlane(uo, j);
            if constexpr (PAIRS) {
#pragma unroll
                for (int u = 0; u < U; u += 2) {
                    xg[u][j] = xg[u + 1][j] = V(0);
                    if (((mask[u] | mask[u + 1]) >> j) & 1)
                        buffer_gather2(xs, (row[u] + oj + PAIR_PAD) * int(sizeof(V)), xg[u][j], xg[u + 1][j]);
                }
            } else {
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    xg[u][j] = V(0);
                    if ((mask[u] >> j) & 1) xg[u][j] = buffer_gather(xs, (row[u] + oj) * int(sizeof(V)), V(0));
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u) xg[u][j] = V(0);
        }
    }
    V sum[U], diag[U];
#pragma unroll
    for (int u = 0; u < U; ++u) { sum[u] = V(0); diag[u] = V(0); }
#pragma unroll
    for (int j = 0; j < UMAX; ++j) {
        if (j < ul) {
            const V vj = lane_pick(uv, j);
            const int oj = __builtin_amdgcn_readlane(uo, j);
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const bool on = (mask[u] >> j) & 1;
                sum[u] = on ? madd(vj, xg[u][j], sum[u]) : sum[u];
                if (NEED_DIAG && oj == 0) diag[u] = on ? diag[u] + vj : diag[u];
            }
        }
    }
    double sq = 0.0;
    if constexpr (FUSED) {
        // relax, then the row's residual with the NEW x_i from the operands still in registers:
        // the same chain ROW_RESIDUAL would run on the updated vector
        V xnew[U], sum2[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { xnew[u] = pre[u].xv + (pre[u].bv - sum[u]) / diag[u]; sum2[u] = V(0); }
#pragma unroll
        for (int j = 0; j < UMAX; ++j) {
            if (j < ul) {
                const V vj = lane_pick(uv, j);
                const int oj = __builtin_amdgcn_readlane(uo, j);
#pragma unroll
                for (int u = 0; u < U; ++u) {
                    const bool on = (mask[u] >> j) & 1;
                    const V xj = oj == 0 ? xnew[u] : xg[u][j];
                    sum2[u] = on ? madd(vj, xj, sum2[u]) : sum2[u];
                }
            }
        }
        if constexpr (PAIRS) {
#pragma unroll
            for (int u = 0; u < U; u += 2) {
                const V res0 = pre[u].bv - sum2[u], res1 = pre[u + 1].bv - sum2[u + 1];
                if (act[u + 1]) {
                    store_pair(a.y + row[u], xnew[u], xnew[u + 1]);
                    if constexpr (MODE == ROW_GS_RES) store_pair(a.zero + row[u], res0, res1);
                    else { add_square(sq, res0); add_square(sq, res1); }
                } else if (act[u]) {
                    a.y[row[u]] = xnew[u];
                    if constexpr (MODE == ROW_GS_RES) a.zero[row[u]] = res0;
                    else add_square(sq, res0);
                }
            }
        } else {
#pragma unroll
            for (int u = 0; u < U; ++u)
                if (act[u]) {
                    const V res = pre[u].bv - sum2[u];
                    a.y[row[u]] = xnew[u];
                    if constexpr (MODE == ROW_GS_RES) a.zero[row[u]] = res;
                    else add_square(sq, res);
                }
        }
    } else if constexpr (PAIRS && mode_one_store(MODE)) {
#pragma unroll
        for (int u = 0; u < U; u += 2) {
            if (act[u + 1]) {
                const V v0 = row_result<MODE>(a, pre[u], sum[u], diag[u], sq);
                const V v1 = row_result<MODE>(a, pre[u + 1], sum[u + 1], diag[u + 1], sq);
                store_pair(a.y + row[u], v0, v1);
            } else if (act[u]) {
                a.y[row[u]] = row_result<MODE>(a, pre[u], sum[u], diag[u], sq);
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < U; ++u)
            if (act[u]) row_epilogue<MODE>(a, row[u], pre[u], sum[u], diag[u], sq);
    }
    if constexpr (NEED_NORM) {
        const double tot = block_sum(sq, s_red);
        if (threadIdx.x == 0) a.partials[blk] = tot;
    }
}

// Tuning switches (speed only): OMG_XCD_REMAP=1 enables the XCD-chunked block mapping
// (measured slower than the hardware's round-robin on the 256^3 stencil: 390 vs 374 us for the
// residual), OMG_NT_LOADS=0 disables the non-temporal matrix loads (measured 3-4 % slower).
int launch_flags() {
    static int v = -1;
    if (v < 0) {
        const char *e = experiment_env("OMG_XCD_REMAP");
        const char *n = experiment_env("OMG_NT_LOADS");
        v = ((e && e[0] == '1') ? 1 : 0) | ((n && n[0] == '0') ? 0 : 2);
    }
    return v;
}

template <int MODE, int U, typename V>
void launch_union(const DevCsrT<V> &A, int64_t blk0, int64_t nblk, const KArgs<V> &k, hipStream_t s) {
    const dim3 g((unsigned)nblk), b(NT);
    const unsigned x_bytes = unsigned(size_t(A.n_cols) * sizeof(V));
    if (A.union_max <= 10) hipLaunchKernelGGL((rows_union_kernel<MODE, U, 10, V>), g, b, 0, s, k, (int)blk0, x_bytes);
    else hipLaunchKernelGGL((rows_union_kernel<MODE, U, UNION_MAX, V>), g, b, 0, s, k, (int)blk0, x_bytes);
    OMG_HIP(hipGetLastError());
}

// kind: 0 rows_kernel, 1 rows_pattern_kernel, 2 rows_union_kernel
template <int MODE, typename V>
void launch_mode(const DevCsrT<V> &A, int64_t blk0, int64_t nblk, const KArgs<V> &k, int kind,
                 bool long_rows, bool ell, hipStream_t s) {
    if (nblk <= 0) return;
    if (kind == 2) {
        if constexpr (MODE != ROW_SCATTER) {
            const int u = (A.rows_cap + NT - 1) / NT;
            if (u <= 1) launch_union<MODE, 1>(A, blk0, nblk, k, s);
            else if (u == 2) launch_union<MODE, 2>(A, blk0, nblk, k, s);
            else launch_union<MODE, 4>(A, blk0, nblk, k, s);
        }
        return;
    }
    const bool all_pattern = kind == 1;
    if (all_pattern) {
        OMG_REQUIRE(A.rows_cap <= NT, "rows_pattern_kernel cannot run blocks of more than 256 rows");
        const dim3 g((unsigned)nblk), b(NT);
        if (ell) {
            if (long_rows) hipLaunchKernelGGL((rows_pattern_kernel<MODE, true, true, V>), g, b, 0, s, k, (int)blk0);
            else hipLaunchKernelGGL((rows_pattern_kernel<MODE, false, true, V>), g, b, 0, s, k, (int)blk0);
        } else {
            if (long_rows) hipLaunchKernelGGL((rows_pattern_kernel<MODE, true, false, V>), g, b, 0, s, k, (int)blk0);
            else hipLaunchKernelGGL((rows_pattern_kernel<MODE, false, false, V>), g, b, 0, s, k, (int)blk0);
        }
        OMG_HIP(hipGetLastError());
        return;
    }
    const int flags = launch_flags();
    static const int group = [] { const char *e = experiment_env("OMG_XCD_GROUP"); return e ? atoi(e) : 0; }();
    const int remap = (nblk < 64) ? 0 : (group > 1 ? group : (flags & 1));
    // small operators live in L2 / Infinity Cache across cycles: keep them cacheable
    const bool ntl = (flags & 2) && A.nnz * int64_t(4 + sizeof(V)) > (int64_t(192) << 20);
    // several rows per thread (short rows): batched variant, built for the modes such
    // operators are used with; other modes fall back to the generic one-row-at-a-time loop
    constexpr bool HAS_SHORT = (MODE == ROW_SPMV || MODE == ROW_AXPY || MODE == ROW_RESIDUAL);
    const dim3 grid((unsigned)nblk), block(NT);
    if constexpr (HAS_SHORT) {
        if (A.rows_cap > NT && A.union_blocks == 0) {
            if (ntl) hipLaunchKernelGGL((rows_kernel<MODE, true, true, 1, true, V>), grid, block, 0, s, k, (int)blk0, remap);
            else hipLaunchKernelGGL((rows_kernel<MODE, false, true, 1, true, V>), grid, block, 0, s, k, (int)blk0, remap);
            OMG_HIP(hipGetLastError());
            return;
        }
    }
    if (A.lanes_per_row == 4) {       // long rows: four lanes per row
        if (ntl) hipLaunchKernelGGL((rows_kernel<MODE, true, false, 4, true, V>), grid, block, 0, s, k, (int)blk0, remap);
        else hipLaunchKernelGGL((rows_kernel<MODE, false, false, 4, true, V>), grid, block, 0, s, k, (int)blk0, remap);
        OMG_HIP(hipGetLastError());
        return;
    }
    if (long_rows) {
        if (ntl) hipLaunchKernelGGL((rows_kernel<MODE, true, false, 1, true, V>), grid, block, 0, s, k, (int)blk0, remap);
        else hipLaunchKernelGGL((rows_kernel<MODE, false, false, 1, true, V>), grid, block, 0, s, k, (int)blk0, remap);
    } else {
        if (ntl) hipLaunchKernelGGL((rows_kernel<MODE, true, false, 1, false, V>), grid, block, 0, s, k, (int)blk0, remap);
        else hipLaunchKernelGGL((rows_kernel<MODE, false, false, 1, false, V>), grid, block, 0, s, k, (int)blk0, remap);
    }
    OMG_HIP(hipGetLastError());
}

template <typename V>
KArgs<V> make_kargs(const DevCsrT<V> &A, const RowArgsT<V> &args) {
    KArgs<V> k;
    k.blk_info = A.blk_rows.p;
    k.indptr = A.indptr.p;
    k.indices = A.indices.p;
    k.data = A.data.p;
    k.ccode = A.ccode.p;
    k.vcode = A.vcode.p;
    k.cdict = A.cdict.p;
    k.vdict = A.vdict.p;
    k.rcode = A.rcode.p;
    k.pidx = A.pidx.p;
    k.pval = A.pval.p;
    k.pbeg = A.pbeg.p;
    k.vell = A.vell.p;
    k.x = args.x;
    k.b = args.b;
    k.y = args.y;
    k.partials = args.partials;
    k.zero = args.zero;
    k.ymap = args.ymap;
    k.omega = V(args.omega);
    k.first_diag = args.first_diag;
    k.first_end = args.first_end;
    k.first_jacobi = args.first_jacobi ? 1 : 0;
    return k;
}

}  // namespace

template <typename V>
void launch_rows(const DevCsrT<V> &A, int mode, int set, const RowArgsT<V> &args, hipStream_t s) {
    if (set < 0) launch_rows_range(A, mode, 0, (int)A.n_sets(), args, s);
    else launch_rows_range(A, mode, set, set + 1, args, s);
}

namespace {

// [set_begin, set_end): all sets go to the same kernel (kind: 0 rows_kernel, 1 rows_pattern_kernel,
// 2 rows_union_kernel)
template <typename V>
void launch_rows_uniform(const DevCsrT<V> &A, int mode, int set_begin, int set_end, int kind,
                         const RowArgsT<V> &args, hipStream_t s) {
    const KArgs<V> k = make_kargs(A, args);
    const int64_t blk0 = A.set_blk[set_begin];
    const int64_t nblk = A.set_blk[set_end] - blk0;
    bool long_rows = false;                    // any row of the range summed as four chains (common.h ASSOC_LEN)
    bool ell = false;                          // any block of the range keeps its values in the ELL array
    for (int q = set_begin; q < set_end; ++q) {
        long_rows = long_rows || A.set_maxlen.empty() || A.set_maxlen[q] > ASSOC_LEN;
        ell = ell || (!A.set_ell.empty() && A.set_ell[q]);
    }
    if (mode == ROW_SCATTER) {                 // exists in the pattern kernel only (common.h)
        OMG_REQUIRE(kind == 1 && A.rows_cap <= NT, "ROW_SCATTER needs an operator whose blocks are all row-pattern coded");
        if (nblk > 0) {
            const dim3 g((unsigned)nblk), b(NT);
            if (ell) hipLaunchKernelGGL((rows_pattern_kernel<ROW_SCATTER, false, true, V>), g, b, 0, s, k, (int)blk0);
            else hipLaunchKernelGGL((rows_pattern_kernel<ROW_SCATTER, false, false, V>), g, b, 0, s, k, (int)blk0);
            OMG_HIP(hipGetLastError());
        }
        return;
    }
    switch (mode) {
        case ROW_GS_RES: launch_mode<ROW_GS_RES>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_GS_NORM: launch_mode<ROW_GS_NORM>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_SPMV: launch_mode<ROW_SPMV>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_RESIDUAL: launch_mode<ROW_RESIDUAL>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_RESNORM: launch_mode<ROW_RESNORM>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_NORM_ONLY: launch_mode<ROW_NORM_ONLY>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_GS: launch_mode<ROW_GS>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_JACOBI: launch_mode<ROW_JACOBI>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_GS_PRENORM: launch_mode<ROW_GS_PRENORM>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_JACOBI_PRENORM: launch_mode<ROW_JACOBI_PRENORM>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        case ROW_AXPY: launch_mode<ROW_AXPY>(A, blk0, nblk, k, kind, long_rows, ell, s); break;
        default: throw Error(OMG_ERR_INVALID, "launch_rows: unknown mode");
    }
}

}  // namespace

template <typename V>
void launch_rows_range(const DevCsrT<V> &A, int mode, int set_begin, int set_end, const RowArgsT<V> &args,
                       hipStream_t s) {
    OMG_REQUIRE(set_begin >= 0 && set_begin <= set_end && size_t(set_end) <= A.n_sets(),
                "launch_rows: set range out of bounds");
    if (set_begin == set_end) return;
    // Which kernel a set runs: rows_union_kernel where every block carries a union (common.h
    // set_union; OMG_UNION_KERNEL=0 clears it at upload); else by set_pattern: 2 = the LDS-free
    // pattern kernel, always; 1 = the pattern kernel unless OMG_PATTERN_KERNEL=0 asks for
    // rows_kernel's walk of the same dictionaries through LDS (identical bits); 0 = rows_kernel.
    // Consecutive sets with the same answer share a launch.
    const char *e = getenv("OMG_PATTERN_KERNEL");
    const bool prefer = !(e && e[0] == '0');
    const bool x_fits = size_t(A.n_cols) * sizeof(V) < (size_t(1) << 31);       // 32-bit buffer offsets
    auto kind_of = [&](int q) {
        // (operators A_l only — square, or a rank's rows with their halo columns behind: restriction
        // / prolongation rows measured faster in the pattern kernel)
        if (mode != ROW_SCATTER && x_fits && A.operator_like() && !A.set_union.empty() && A.set_union[q]) return 2;
        const int v = A.set_pattern.empty() ? 0 : A.set_pattern[q];
        if (A.rows_cap > NT && A.union_blocks) return 0;                        // big pattern blocks: rows_kernel walks them
        return (v == 2 || (v == 1 && (prefer || mode == ROW_SCATTER))) ? 1 : 0;   // the scatter exists in the pattern kernel only
    };
    auto empty = [&](int q) { return A.set_blk[q + 1] == A.set_blk[q]; };
    int q0 = set_begin;
    while (q0 < set_end) {
        if (empty(q0)) { ++q0; continue; }
        const int kind = kind_of(q0);
        int q1 = q0 + 1;
        while (q1 < set_end && (empty(q1) || kind_of(q1) == kind)) ++q1;
        launch_rows_uniform(A, mode, q0, q1, kind, args, s);
        q0 = q1;
    }
}

template <typename V>
void launch_gs_serial(const DevCsrT<V> &A, int set_begin, int set_end, const RowArgsT<V> &args,
                      hipStream_t s) {
    KArgs<V> k = make_kargs(A, args);
    k.partials = nullptr;
    k.zero = nullptr;
    k.ymap = nullptr;
    const int b0 = (int)A.set_blk[set_begin], b1 = (int)A.set_blk[set_end];
    if (b1 <= b0) return;
    // the serial kernel walks rows_kernel's LDS image: sets that only the pattern kernel can run
    // are launched one by one instead
    for (int q = set_begin; q < set_end; ++q)
        if (!A.set_pattern.empty() && A.set_pattern[q] == 2) {
            for (int t = set_begin; t < set_end; ++t) launch_rows_range(A, ROW_GS, t, t + 1, args, s);
            return;
        }
    // the same lanes-per-row association as launch_rows uses for this operator: a row's sum
    // must not depend on which of the two kernels relaxes it
    bool long_rows = false;
    for (int q = set_begin; q < set_end; ++q) long_rows = long_rows || A.set_maxlen.empty() || A.set_maxlen[q] > ASSOC_LEN;
    if (A.lanes_per_row == 4) hipLaunchKernelGGL((rows_serial_kernel<ROW_GS, 4, true, V>), dim3(1), dim3(NT), 0, s, k, b0, b1);
    else if (long_rows) hipLaunchKernelGGL((rows_serial_kernel<ROW_GS, 1, true, V>), dim3(1), dim3(NT), 0, s, k, b0, b1);
    else hipLaunchKernelGGL((rows_serial_kernel<ROW_GS, 1, false, V>), dim3(1), dim3(NT), 0, s, k, b0, b1);
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

// The two launches above for `count` partial arrays at once, one workgroup each (array c at
// p_all + c * stride, result out[c]): stage 1 is fold_kernel's sum of SUM_FOLD chunks — four chunks
// at a time, one per 256 threads — stage 2 sum_kernel's; every addition in the same order, so
// out[c] has the bits launch_sum / launch_sum_sqrt would give for that array.
__global__ __launch_bounds__(1024) void sum_batch_kernel(const double *__restrict__ p_all, int64_t stride, int64_t n,
                                                         double *__restrict__ out, int take_sqrt) {
    __shared__ double s_fold[SUM_FOLD];
    __shared__ double s_w[16];
    const double *__restrict__ p = p_all + blockIdx.x * stride;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    double acc = 0.0;
    if (n > 8192) {
        const int64_t chunk = (n + SUM_FOLD - 1) / SUM_FOLD;
        for (int g = 0; g < SUM_FOLD / 4; ++g) {
            const int c = 4 * g + (tid >> 8), t = tid & 255;
            const int64_t lo = c * chunk, hi = min(n, lo + chunk);
            double a = 0.0;
            for (int64_t i = lo + t; i < hi; i += 256) a += p[i];
            a = wave_sum(a);
            if (lane == 0) s_w[wave] = a;
            __syncthreads();
            if (t == 0) {
                const int w0 = 4 * (tid >> 8);
                s_fold[c] = (s_w[w0] + s_w[w0 + 1]) + (s_w[w0 + 2] + s_w[w0 + 3]);
            }
            __syncthreads();
        }
        for (int i = tid; i < SUM_FOLD; i += 1024) acc += s_fold[i];
    } else {
        for (int64_t i = tid; i < n; i += 1024) acc += p[i];
    }
    acc = wave_sum(acc);
    if (lane == 0) s_w[wave] = acc;
    __syncthreads();
    if (tid == 0) {
        double t = 0.0;
        for (int w = 0; w < 16; ++w) t += s_w[w];
        out[blockIdx.x] = take_sqrt ? sqrt(t) : t;
    }
}

template <typename S, typename D>
__global__ void gather_kernel(const S *__restrict__ src, const int32_t *__restrict__ idx,
                              D *__restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[i] = D(src[idx ? idx[i] : i]);
}

template <typename S, typename D>
__global__ void scatter_kernel(const S *__restrict__ src, const int32_t *__restrict__ idx,
                               D *__restrict__ dst, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n;
         i += (int64_t)gridDim.x * blockDim.x)
        dst[idx ? idx[i] : i] = D(src[i]);
}

// out = M v, M row-major rows x n.  One wave per row, 16-byte loads, 4 rows per workgroup.
template <typename V>
__global__ __launch_bounds__(256) void dense_gemv_kernel(const V *__restrict__ M, const V *__restrict__ v,
                                                         V *__restrict__ out, int64_t rows, int64_t n) {
    using vdat = typename Vec16<V>::type;
    constexpr int VN = Vec16<V>::N;
    const int lane = threadIdx.x & 63;
    const int64_t row = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= rows) return;
    const V *m = M + row * n;
    V acc = V(0);
    if (n % VN == 0) {
        const vdat *m2 = reinterpret_cast<const vdat *>(m);
        const vdat *v2 = reinterpret_cast<const vdat *>(v);
        for (int64_t j = lane; j < n / VN; j += 64) {
            const vdat a = m2[j], b = v2[j];
#pragma unroll
            for (int q = 0; q < VN; ++q) acc += a[q] * b[q];
        }
    } else {
        for (int64_t j = lane; j < n; j += 64) acc += m[j] * v[j];
    }
    acc = wave_sum(acc);
    if (lane == 0) out[row] = acc;
}

template <typename V>
__global__ void diagonal_kernel(const int32_t *__restrict__ ip, const int32_t *__restrict__ ix, const V *__restrict__ dv,
                                V *__restrict__ diag, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        V d = V(0);
        for (int32_t p = ip[i]; p < ip[i + 1]; ++p)
            if (ix[p] == i) d = d + dv[p];
        diag[i] = d;
    }
}

template <typename V>
__global__ void first_relaxation_kernel(const V *__restrict__ b, const V *__restrict__ diag, V *__restrict__ x, int64_t n,
                                        int64_t first_end, int jacobi, V omega) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        V x0 = V(0);
        if (i < first_end) {
            const V q = (b[i] - V(0)) / diag[i];
            x0 = V(0) + (jacobi ? omega * q : q);
        }
        x[i] = x0;
    }
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

void launch_sum_batch(const double *partials, int64_t stride, int64_t n, int count, double *out, bool take_sqrt,
                      hipStream_t s) {
    if (count <= 0) return;
    hipLaunchKernelGGL(sum_batch_kernel, dim3((unsigned)count), dim3(1024), 0, s, partials, stride, n, out, take_sqrt ? 1 : 0);
    OMG_HIP(hipGetLastError());
}

void launch_sum(double *partials, int64_t n, double *out, hipStream_t s) {
    launch_sum_impl(partials, n, out, 0, s);
}

void launch_sum_sqrt(double *partials, int64_t n, double *out, hipStream_t s) {
    launch_sum_impl(partials, n, out, 1, s);
}

template <typename S, typename D>
void launch_gather(const S *src, const int32_t *idx, D *dst, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((gather_kernel<S, D>), dim3(grid_for(n, 256)), dim3(256), 0, s, src, idx, dst, n);
    OMG_HIP(hipGetLastError());
}

template <typename S, typename D>
void launch_scatter(const S *src, const int32_t *idx, D *dst, int64_t n, hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((scatter_kernel<S, D>), dim3(grid_for(n, 256)), dim3(256), 0, s, src, idx, dst, n);
    OMG_HIP(hipGetLastError());
}

template <typename V>
void launch_diagonal(const DevCsrT<V> &A, V *diag, hipStream_t s) {
    if (A.n_rows <= 0) return;
    hipLaunchKernelGGL((diagonal_kernel<V>), dim3(grid_for(A.n_rows, 256)), dim3(256), 0, s, A.indptr.p, A.indices.p, A.data.p,
                       diag, A.n_rows);
    OMG_HIP(hipGetLastError());
}

template <typename V>
void launch_first_relaxation(const V *b, const V *diag, V *x, int64_t n, int64_t first_end, bool jacobi, double omega,
                             hipStream_t s) {
    if (n <= 0) return;
    hipLaunchKernelGGL((first_relaxation_kernel<V>), dim3(grid_for(n, 256)), dim3(256), 0, s, b, diag, x, n, first_end,
                       jacobi ? 1 : 0, V(omega));
    OMG_HIP(hipGetLastError());
}

template <typename V>
void launch_dense_gemv(const V *M, const V *v, V *out, int64_t n, hipStream_t s) {
    launch_dense_gemv_rows(M, v, out, n, n, s);
}

template <typename V>
void launch_dense_gemv_rows(const V *M, const V *v, V *out, int64_t rows, int64_t n, hipStream_t s) {
    if (rows <= 0 || n <= 0) return;
    hipLaunchKernelGGL((dense_gemv_kernel<V>), dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, M, v, out, rows, n);
    OMG_HIP(hipGetLastError());
}

// ---- explicit instantiations (V = double, float) -------------------------------------------
#define OMG_INSTANTIATE(V)                                                                                \
    template void launch_rows<V>(const DevCsrT<V> &, int, int, const RowArgsT<V> &, hipStream_t);         \
    template void launch_rows_range<V>(const DevCsrT<V> &, int, int, int, const RowArgsT<V> &, hipStream_t); \
    template void launch_gs_serial<V>(const DevCsrT<V> &, int, int, const RowArgsT<V> &, hipStream_t);    \
    template void launch_dense_gemv<V>(const V *, const V *, V *, int64_t, hipStream_t);                   \
    template void launch_diagonal<V>(const DevCsrT<V> &, V *, hipStream_t);                               \
    template void launch_first_relaxation<V>(const V *, const V *, V *, int64_t, int64_t, bool, double, hipStream_t); \
    template void launch_dense_gemv_rows<V>(const V *, const V *, V *, int64_t, int64_t, hipStream_t);
OMG_INSTANTIATE(double)
OMG_INSTANTIATE(float)
#undef OMG_INSTANTIATE
#define OMG_INSTANTIATE_CVT(S, D)                                                                         \
    template void launch_gather<S, D>(const S *, const int32_t *, D *, int64_t, hipStream_t);             \
    template void launch_scatter<S, D>(const S *, const int32_t *, D *, int64_t, hipStream_t);
OMG_INSTANTIATE_CVT(double, double)
OMG_INSTANTIATE_CVT(double, float)
OMG_INSTANTIATE_CVT(float, double)
OMG_INSTANTIATE_CVT(float, float)
#undef OMG_INSTANTIATE_CVT

}  // namespace omg
