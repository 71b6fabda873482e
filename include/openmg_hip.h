/*
 * openmg_hip.h — C ABI of libopenmg_hip.so, the MI355X (gfx950) implementation of the
 * multigrid V-cycle hot path of tsbertalan/openmg.
 *
 * The reference is pure Python and has no FFI of its own, so every entry
 * point below is defined by the reference CALL SITE it replaces; the citation after
 * "replaces:" is a path under the reference root.  The host side (openmg_amd/, Python,
 * ctypes) keeps the reference's function names and dict semantics and forwards here.
 *
 * Conventions
 *   - every pointer is a HOST pointer borrowed for the duration of the call unless the
 *     name ends in _dev; nothing is retained except inside an omg_hierarchy;
 *   - matrices are CSR: int32 indptr[n_rows+1], int32 indices[nnz], float64 data[nnz];
 *     column order inside a row is free (SciPy's SpGEMM leaves it unsorted) and row sums
 *     run in STORED order like the reference's (openmg/solvers.py:63-65);
 *   - vectors are float64;
 *   - every function returns OMG_OK (0) or an OMG_ERR_* code; omg_last_error() gives text;
 *   - there is NO CPU fallback: without a usable GPU every compute call returns
 *     OMG_ERR_NO_DEVICE;
 *   - what the device keeps of an operator is a LOSSLESS recoding of the caller's CSR (row
 *     patterns, dictionaries, block-transposed values; csrc/common.h):
 *     a storage choice per row block that never changes a result bit
 *     (omg_hierarchy_format_info reports it, env OMG_COMPRESS=0 turns it off).
 */
#ifndef OPENMG_HIP_H
#define OPENMG_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define OMG_OK                0
#define OMG_ERR_INVALID       1   /* bad argument / shape mismatch                         */
#define OMG_ERR_NO_DEVICE     2   /* no HIP device visible                                 */
#define OMG_ERR_HIP           3   /* a HIP runtime call failed                             */
#define OMG_ERR_SINGULAR      4   /* coarse matrix is singular (reference: SuperLU error)  */
#define OMG_ERR_NO_DIAGONAL   5   /* a row has no stored diagonal (reference: divide by 0) */
#define OMG_ERR_ALLOC         6
#define OMG_ERR_UNSUPPORTED   7

/* Smoother orderings.  The reference has exactly one smoother: lexicographic in-place
 * Gauss-Seidel (openmg/solvers.py:34-75).
 *   GS_LEX    the same iterate as the reference: rows are level-scheduled (row i runs after
 *             every coupled row j<i and before every coupled row j>i), so a sweep is a
 *             sequence of independent row sets — identical arithmetic per row.
 *   GS_COLOUR greedy multi-colour Gauss-Seidel (red-black on 5/7-point stencils, 8 colours
 *             on 27-point); equals the reference's sweep on the colour-permuted system.
 *   JACOBI    weighted Jacobi, x += omega D^-1 (b - A x); no reference counterpart.      */
#define OMG_SMOOTH_GS_LEX     0
#define OMG_SMOOTH_GS_COLOUR  1
#define OMG_SMOOTH_JACOBI     2

typedef struct {
    int64_t n_rows, n_cols, nnz;
    const int32_t *indptr;
    const int32_t *indices;
    const double  *data;
} omg_csr;

typedef struct omg_hierarchy omg_hierarchy;   /* device-resident level hierarchy            */
typedef struct omg_csr_result omg_csr_result; /* device-built CSR waiting to be fetched     */

const char *omg_last_error(void);
int omg_device_count(int *count);
int omg_set_device(int device);
/* free / total memory of the current device (hipMemGetInfo): what a caller sizes its problem by; tests check that
 * destroying a hierarchy gives everything back */
int omg_device_mem_info(int64_t *free_bytes, int64_t *total_bytes);
/* every stream of the current device idle (before device arrays another library produced are handed to omg_vcycle_dev) */
int omg_device_synchronize(void);
/* Build identification (git-independent): returns a static string. */
const char *omg_version(void);

/* ---- hierarchy --------------------------------------------------------------------
 * replaces: the A and R lists built by openmg/__init__.py:103-109 and carried through
 * every mgCycle call (openmg/__init__.py:151).  A[0..n_levels-1] are the level operators,
 * R[0..n_levels-2] the restrictions (R[l]: level l -> l+1); prolongation is R[l]^T
 * (openmg/__init__.py:214).  The coarsest operator A[n_levels-1] is factorised once here
 * (the reference refactorises it on every cycle, openmg/solvers.py:23).
 * n_levels == 1 is allowed (direct solve only).                                          */
int omg_hierarchy_create(int n_levels, const omg_csr *A, const omg_csr *R,
                         int smoother, double omega, omg_hierarchy **out);
/* Same, choosing the precision the levels are STORED and COMPUTED in on the device
 * (BASELINE.json configs[4] is fp32).  OMG_DTYPE_F64 is omg_hierarchy_create.  With
 * OMG_DTYPE_F32 the operators are rounded to float once at upload, every vector lives in HBM
 * as float, row sums run in float fma; residual norms are still accumulated in double, and
 * the coarsest operator is inverted in double and then rounded.  Every entry point keeps its
 * double host (and omg_hierarchy_cycle_dev device) vectors: the conversion happens on the
 * device.  The reference has no such switch (it is fp64 throughout); parity for fp32 is
 * "within single-precision rounding of the fp64 iterate", stated per test.               */
#define OMG_DTYPE_F64 0
#define OMG_DTYPE_F32 1
int omg_hierarchy_create_ex(int n_levels, const omg_csr *A, const omg_csr *R, int smoother,
                            double omega, int dtype, omg_hierarchy **out);
/* replaces: the setup of openmg.mgSolve — operators.restrictionList(problemShape, ...) + operators.coeffecientList(A_in, R)
 * (openmg/__init__.py:103-109) — and omg_hierarchy_create_ex in ONE call that keeps everything on the device: A_in is
 * uploaded once, every R[l] is built and every Galerkin product (R A) R^T formed in HBM, every smoothed level is
 * qualified there for a fused path (plane passes, 2-D tile passes, 27-point kernels; the grid is the caller's
 * problemShape = shape[0..dim-1], dim 2 or 3, shape[0] == shape[dim-1], extents divisible by 2^n_restrictions), and only
 * the coarsest operator visits the host (its factorisation).  n_restrictions: len(restrictionList(...)) by the
 * reference's depth rule (the caller applies it: it needs the shapes only).  If some level does not qualify its operators
 * are fetched and the hierarchy is built the ordinary way: the same object either way.  The lists mgSolve returns in
 * infoDict['A'], ['R'] are NOT kept: a caller with giveInfo takes the ordinary route.                                  */
int omg_hierarchy_create_from_fine(const omg_csr *A_in, int dim, const int64_t *shape, int n_restrictions, int smoother, double omega,
                                   int dtype, omg_hierarchy **out);

/* New values for the fine operator of a hierarchy made by omg_hierarchy_create_from_fine whose smoothed levels all run the
 * 27-point kernels (BASELINE configs[4]: "Galerkin RAP rebuilt on-device"; what the reference would do is run
 * operators.coeffecientList again, openmg/operators.py:144-188): data = the CSR's value array in the SAME pattern, nnz
 * doubles, in host memory or — on_device != 0 — already in HBM.  Level 0 is re-tiled, every Galerkin product re-formed
 * on the device in one pass per level (closed form of the aggregation on a grid, SciPy's accumulation order: the bits of
 * a fresh setup), the coarsest operator factorised anew.  The resident right-hand side and iterate stay. */
int omg_hierarchy_update_fine(omg_hierarchy *h, const double *data, int64_t nnz, int on_device);
int omg_hierarchy_dtype(const omg_hierarchy *h, int *dtype);
int omg_hierarchy_destroy(omg_hierarchy *h);
/* Run on a caller-owned hipStream_t instead of the hierarchy's own stream (NULL = own). */
int omg_hierarchy_set_stream(omg_hierarchy *h, void *hip_stream);
int omg_hierarchy_sync(omg_hierarchy *h);
/* Rows of level l; number of independent row sets one smoother sweep is split into. */
int omg_hierarchy_level_rows(const omg_hierarchy *h, int level, int64_t *n_rows);
int omg_hierarchy_level_sets(const omg_hierarchy *h, int level, int64_t *n_sets);
/* 1 when the smoother's last set launch of this level also produces that set's residual /
 * norm share (ROW_GS_RES / ROW_GS_NORM in csrc/common.h), so that the residual and norm
 * launches cover only the other sets.  Bit-identical results either way.                  */
int omg_hierarchy_level_fused(const omg_hierarchy *h, int level, int *fused);
/* Schedule choices of a level as a bit mask: 1 = fused last set (as above); 2 = prolongation
 * runs as a scatter over the restriction's row patterns (csrc/common.h ROW_SCATTER) instead of
 * a pass over the explicit transpose.  Bit-identical results either way.                   */
#define OMG_LEVEL_FUSED_LAST_SET   1
#define OMG_LEVEL_SCATTER_PROLONG  2
#define OMG_LEVEL_UNION_WALK       16  /* every smoother set of A runs rows_union_kernel (several rows per thread) */
#define OMG_LEVEL_MARCH            32  /* lexicographic Gauss-Seidel runs as one wavefront launch per sweep (march.hip) */
#define OMG_LEVEL_MARCH_SCAN       512 /* ... and that launch is the line-scan kernel (OMG_MARCH_SCAN=1 when the hierarchy was made: 3-D,
                                           at most 255 distinct rows, lines of at most 512 rows): rounding-level differences from
                                           the sequential loop, march.hip scan_gs_kernel */
#define OMG_LEVEL_PLANE            64  /* red-black sweeps of a grid star stencil: each half of a V(>=1, >=1) cycle over this
                                        * level (openmg/__init__.py:201-210 and :214-227) is ONE plane-pipelined launch (plane.hip) */
#define OMG_LEVEL_VAR7             256 /* 7-point grid stencil with per-row coefficients under the 2x2x2 aggregation, red-black:
                                        * each half of the cycle over this level is one launch (var7.hip) */
#define OMG_LEVEL_STENCIL27        128 /* 27-point grid stencil with per-row coefficients under the 2x2x2 aggregation, 8-colour
                                        * Gauss-Seidel (BASELINE configs[4]): the cycle over this level runs the octant-layout
                                        * kernels of stencil27.hip — four launches per sweep, the coefficients streamed once each */
int omg_hierarchy_level_flags(const omg_hierarchy *h, int level, int *flags);
/* Switch the plane-pipelined passes of a hierarchy off (enable = 0) or back on: the cycle then runs set
 * by set (same iterate, bit for bit; the norm's partial sums are associated differently).  A/B only.
 * OMG_PLANE=0 in the environment at creation never builds them.                                      */
int omg_hierarchy_use_plane(omg_hierarchy *h, int enable);
/* Tiling of a level's plane-pipelined passes: out[8] = cells per line, lines per plane, planes, a
 * workgroup's interior cells in x / lines / planes, workgroups, threads per workgroup (all 0 when the
 * level has none).                                                                                   */
int omg_hierarchy_plane_info(const omg_hierarchy *h, int level, int64_t *out8);
/* Rows and stored entries of one smoother set (for byte accounting of per-set launches). */
int omg_hierarchy_set_info(const omg_hierarchy *h, int level, int set, int64_t *rows, int64_t *nnz);
/* How an operator of level l sits in HBM (csrc/common.h "Block-dictionary coding": the device
 * format is a lossless recoding of the caller's CSR, chosen per row block).  op: 0 = A[l],
 * 1 = R[l], 2 = R[l]^T; set: one smoother set of A[l], or -1 for the whole operator.  out[]:
 *   0 rows  1 stored entries  2 row blocks
 *   3 row blocks / 4 rows / 5 entries held as row patterns (one byte per row)
 *   6 entries with a one-byte column code   7 entries with a one-byte value code
 *   8 bytes of the operator one launch over these rows reads from HBM in this format
 *   9 the same for plain int32 CSR: entries * (4 + sizeof value) + 4 * rows
 *  10 row blocks / 11 entries whose values sit block-transposed beside offset-only row patterns
 *     (variable coefficients: the one-thread-per-row kernel reads them coalesced)              */
#define OMG_FORMAT_FIELDS 12
int omg_hierarchy_format_info(const omg_hierarchy *h, int level, int op, int set, int64_t *out);
/* Host-only check of that recoding (needs no GPU): codes the operator exactly as an upload
 * would (one smoother set, dtype as in omg_hierarchy_create_ex), decodes it the way the kernels
 * do and compares every column and every value bit with the input; OMG_ERR_INVALID on any
 * difference.  out[] as above.                                                             */
int omg_format_selftest(const omg_csr *A, int dtype, int64_t *out);

/* replaces: openmg.mgCycle(A, b, level, R, parameters, initial) — openmg/__init__.py:151-236.
 * One V-cycle entered at `level` with pre/post = parameters['preIterations'|'postIterations'].
 * x holds `initial` on entry (zeros for initial=None, :191-192) and uOut on return.
 * *norm (nullable) = ||b - A[level] uOut||_2 (:227), 0 when level is the coarsest (:232). */
int omg_vcycle(omg_hierarchy *h, int level, const double *b, double *x,
               int pre, int post, double *norm);
/* The same cycle with the reference's in-place pre-smoother made explicit (Q2): gaussSeidel writes x[i] of the
 * caller's `initial` (openmg/solvers.py:68) and returns the same object (:75), so after mgCycle the array passed as
 * `initial` holds the PRE-SMOOTHED iterate (openmg/__init__.py:201) while uOut is a new array (:220 / :224).
 * x_in: initial iterate (NULL: zeros, :191-192; nothing is uploaded); x_out: uOut; x_pre (nullable): the iterate
 * after the pre-smoothing sweeps of `level` (equal to x_in when pre = 0).  One call, each vector over PCIe once. */
int omg_vcycle_ex(omg_hierarchy *h, int level, const double *b, const double *x_in, double *x_out, double *x_pre,
                  int pre, int post, double *norm);

/* replaces: the cycle loop of openmg.mgSolve — openmg/__init__.py:112-138.  At least one
 * cycle, then until cycle >= max_cycles (if max_cycles > 0) or norm < threshold (if
 * threshold > 0).  x: initial iterate in, solution out.  The both-off ValueError
 * (:118-119) is raised by the Python caller after the first cycle; here both-off is
 * OMG_ERR_INVALID.                                                                        */
int omg_solve(omg_hierarchy *h, const double *b, double *x, int pre, int post,
              int max_cycles, double threshold, int *cycles_done, double *norm);

/* replaces: openmg.mgCycle(A, b, level, R, parameters, initial) (openmg/__init__.py:151-236) for a caller whose b, initial
 * and uOut are DEVICE arrays (double, natural numbering; e.g. the result of the previous call): omg_vcycle_ex without the
 * PCIe copies — only the norm comes back to the host; the call returns when the outputs are complete.  x_in_dev NULL:
 * zeros (:191-192); x_pre_dev (NULL: not wanted; may be x_in_dev) receives the pre-smoothed iterate (Q2).               */
int omg_vcycle_dev(omg_hierarchy *h, int level, const double *b_dev, const double *x_in_dev, double *x_out_dev,
                   double *x_pre_dev, int pre, int post, double *norm);
/* omg_resident_load / omg_resident_fetch for device arrays (mgSolve for a caller on the GPU: openmg/__init__.py:112-138) */
int omg_resident_load_dev(omg_hierarchy *h, const double *b_dev, const double *x0_dev /* NULL = zeros */);
int omg_resident_fetch_dev(omg_hierarchy *h, double *x_dev);

/* Device-pointer cycle: b_dev, x_dev are level-0 DEVICE vectors in natural numbering, x starts
 * from zero, work is enqueued on hip_stream (NULL = the hierarchy's own) without host sync. */
int omg_hierarchy_cycle_dev(omg_hierarchy *h, const double *b_dev, double *x_dev, int pre, int post,
                            void *hip_stream);

/* Device-resident variants used by bench.py and by repeated mgCycle calls: level-0 b and x
 * live in HBM between calls.                                                             */
int omg_resident_load(omg_hierarchy *h, const double *b, const double *x0 /* NULL = zeros */);
int omg_resident_cycle(omg_hierarchy *h, int pre, int post, double *norm /* NULL = no readback */);
/* n_cycles cycles back to back with EVERY cycle's residual norm computed (openmg/__init__.py:227)
 * and returned in norms[n_cycles]; one host synchronisation at the end.  mgSolve's loop
 * (openmg/__init__.py:132-138) when it stops on a cycle count.  Where the ordering allows it
 * (two colour sets, or Jacobi) the norm of cycle k is finished inside the first launch of cycle
 * k + 1, which forms the same residuals anyway — same bits as n_cycles omg_resident_cycle calls;
 * OMG_NO_PRENORM=1 switches that off.                                                        */
int omg_resident_cycles(omg_hierarchy *h, int pre, int post, int n_cycles, double *norms /* nullable */);
int omg_resident_fetch(omg_hierarchy *h, double *x);
/* Fine-grid SpMV y = A[0] x (tools.flexibleMmult, openmg/tools.py:26) on the resident
 * operator, `reps` back-to-back launches inside one hipEvent bracket; *avg_ms per launch. */
int omg_resident_spmv_time(omg_hierarchy *h, int reps, double *avg_ms);
/* Capture one resident cycle into a hipGraph and replay it on later omg_resident_cycle
 * calls with the same (pre, post).  enable = 0 drops the graph.                          */
int omg_resident_use_graph(omg_hierarchy *h, int enable);

/* Per-kernel timing of level-0 launches with hipEvents on the hierarchy's stream.
 * Classes: 0 smoother set-sweep, 1 residual, 2 restrict, 3 prolong-add, 4 residual+norm,
 * 5 plane-pipelined down pass (sweep + residual + restriction), 6 plane-pipelined up pass
 * (prolongation + sweep + norm).
 * omg_profile_enable(h, mask): bit c of mask switches class c on (-1 = all, 0 = off) and
 * clears the totals; each timed launch costs two hipEventRecords (~10 us of stream gap), so
 * bench.py times only class 1 inside its timed region.  omg_profile_read syncs and returns,
 * per class, launches and total milliseconds since the last omg_profile_enable.           */
#define OMG_PROFILE_CLASSES 7
int omg_profile_enable(omg_hierarchy *h, int mask);
int omg_profile_read(omg_hierarchy *h, int64_t *launches, double *total_ms);

/* ---- single operations on one level of a hierarchy (kernel-level parity tests) ------- */
/* replaces: solvers.smooth(A[l], b, x, iterations) — openmg/solvers.py:28-29; x in place. */
int omg_level_smooth(omg_hierarchy *h, int level, const double *b, double *x, int iterations);
/* replaces: tools.getresidual(b, A[l], x, N) — openmg/tools.py:12-15 (+ norm, __init__.py:227). */
int omg_level_residual(omg_hierarchy *h, int level, const double *b, const double *x,
                       double *r, double *norm /* nullable */);
/* replaces: tools.flexibleMmult(A[level], x) — openmg/tools.py:26 (csr_matvec) — on the operator as the hierarchy
 * holds it: a plane level (constant-coefficient grid stencil, red-black) applies it matrix-free, any other level walks
 * its device format; the same bits either way.                                                                        */
int omg_level_spmv(omg_hierarchy *h, int level, const double *x, double *y);

/* replaces: flexibleMmult(R[l], residual) — openmg/__init__.py:210. */
int omg_level_restrict(omg_hierarchy *h, int level, const double *fine, double *coarse);
/* replaces: uApx + flexibleMmult(R[l].transpose(), coarseCorrection) — openmg/__init__.py:214,220,224. */
int omg_level_prolong_add(omg_hierarchy *h, int level, const double *coarse, double *fine_inout);
/* replaces: solvers.coarseSolve(A[-1], b) — openmg/solvers.py:16-26. */
int omg_coarse_solve(omg_hierarchy *h, const double *b, double *x);
/* How the coarsest operator is factored (the reference re-runs SuperLU on every cycle,
 * openmg/solvers.py:23; here once at setup): out[0] = interior blocks P (1 = explicit dense
 * inverse; > 1 = substructuring along the band into P blocks and P - 1 separators), out[1] =
 * unknowns, out[2] = half-bandwidth, out[3] = device bytes one solve reads.                  */
int omg_hierarchy_coarse_info(const omg_hierarchy *h, int64_t *out4);

/* ---- standalone operations ----------------------------------------------------------- */
/* replaces: tools.flexibleMmult(A, x) for sparse A, dense vector x — openmg/tools.py:26. */
int omg_spmv(const omg_csr *A, const double *x, double *y);
/* replaces: tools.getresidual — openmg/tools.py:12-15; *norm nullable. */
int omg_residual(const omg_csr *A, const double *b, const double *x, double *r, double *norm);
/* replaces: solvers.gaussSeidel(A, b, x, iterations, threshold) — openmg/solvers.py:34-75,
 * incl. smoothToThreshold (:31-32).  iterations < 0 = None, threshold < 0 = None; both
 * None = one sweep (:39-40).  The norm test runs before the first sweep and after every
 * sweep (:43-54).  x in place.  *sweeps_done nullable.                                    */
int omg_gauss_seidel(const omg_csr *A, const double *b, double *x, int smoother, double omega,
                     int iterations, double threshold, int *sweeps_done);
/* replaces: solvers.coarseSolve(A, b) for a one-off solve — openmg/solvers.py:16-26. */
int omg_direct_solve(const omg_csr *A, const double *b, double *x);

/* replaces: flexibleMmult(flexibleMmult(R, A), R.T) — openmg/operators.py:184-186.
 * Galerkin triple product on the device.  Two steps because the output size is unknown:
 * omg_rap() computes it and reports the shape, omg_csr_result_fetch() copies it into
 * caller-allocated arrays (columns sorted ascending inside each row) and frees it.       */
int omg_rap(const omg_csr *R, const omg_csr *A, omg_csr_result **out,
            int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
/* replaces: tools.flexibleMmult(X, Y) with both operands sparse — openmg/tools.py:26 (SciPy
 * csr_matmat).  C = X Y on the device; every C(i,j) is accumulated in SciPy's order.      */
int omg_spgemm(const omg_csr *X, const omg_csr *Y, omg_csr_result **out,
               int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
int omg_csr_result_fetch(omg_csr_result *res, int32_t *indptr, int32_t *indices, double *data);
int omg_csr_result_free(omg_csr_result *res);

/* replaces: operators.restriction(shape) — openmg/operators.py:15-89 — built on the device.
 * shape[0..dim-1], dim in 1..3; same index quirks as the reference (second-axis offset is
 * shape[0], third-axis offset shape[0]*shape[1]).  Output arrays are caller-allocated:
 * indptr[n/2^dim + 1], indices[n], data[n] with n = prod(shape) when all extents are even. */
int omg_restriction(int dim, const int64_t *shape, int32_t *indptr, int32_t *indices,
                    double *data, int64_t *n_rows, int64_t *nnz);

/* Host-side helper of the mgCycle binding (no reference counterpart, no device involved): mgCycle(A, b, level, R,
 * parameters, initial) — openmg/__init__.py:151 — is handed the operator lists on EVERY call; the binding keeps the
 * device hierarchy between calls and recognises the lists by a checksum of every byte (an operator edited in place
 * must not meet a stale device copy).  64-bit digest of buf[0..bytes), chunks hashed on all host threads.           */
int omg_host_checksum(const void *buf, int64_t bytes, uint64_t *out);

/* ---- multi-GPU: one process per GPU, 1-D slabs, RCCL halo exchange ---------------------------
 * No reference counterpart (the reference is single-threaded): the same
 * mgCycle (openmg/__init__.py:151-236) run on a row-partitioned hierarchy.  Each rank owns a
 * contiguous row range of every level.  A level of one rank is described by:
 *   A      owned rows; columns 0..n_loc-1 are the owned unknowns (local numbering), columns
 *          n_loc..n_loc+n_halo-1 the remote unknowns its rows touch (the halo), grouped by
 *          owning peer in the order of `peers`;
 *   R      restriction to the next level, (coarse owned rows) x (fine owned rows) — slabs
 *          must be cut on aggregate boundaries (ignored on the coarsest level);
 *   keys   smoother set of every owned row, 0 <= key < n_sets, consistent across ranks
 *          (NULL: one set, e.g. Jacobi; ignored on the coarsest level);
 *   peers[n_peers], send_off[n_peers+1], send_idx[send_off[n_peers]] (owned rows, local
 *          numbering, sent to each peer), recv_off[n_peers+1] (slots of the halo region
 *          filled by each peer; recv_off[n_peers] == n_halo).
 * coarse_global is the WHOLE coarsest operator (replicated direct solve); coarse_counts[q] =
 * coarsest rows owned by rank q.                                                           */
typedef struct {
    omg_csr A;
    omg_csr R;
    int64_t n_halo;
    const int32_t *keys;
    int32_t n_sets;
    int32_t n_peers;
    const int32_t *peers;
    const int64_t *send_off;
    const int32_t *send_idx;
    const int64_t *recv_off;
    /* 1: plain sets.  2: sets come in (boundary, interior) pairs of one colour — rows that
     * other ranks need first, the rest second; the halo exchange of a pair then runs on a
     * second stream while the interior rows are still being relaxed.                        */
    int32_t set_group;
    /* Optional, one per (peer) entry: the smoother colour whose values the entry carries
     * (peers[] then repeats a rank once per colour).  After relaxing colour c only entries of
     * group c travel; the exchange after prolongation moves all of them.  NULL: every entry
     * is sent every time.                                                                  */
    const int32_t *entry_group;
} omg_dist_level;

typedef struct omg_dist omg_dist;
typedef struct omg_dist_group omg_dist_group;

int omg_dist_create(int rank, int n_ranks, int n_levels, const omg_dist_level *levels,
                    const omg_csr *coarse_global, const int64_t *coarse_counts,
                    int smoother, double omega, omg_dist **out);
/* Same with the levels stored / computed in `dtype` (OMG_DTYPE_F64 is omg_dist_create); halo
 * messages and the coarse all-gather then travel in that type.  Host vectors stay double.  */
int omg_dist_create_ex(int rank, int n_ranks, int n_levels, const omg_dist_level *levels,
                       const omg_csr *coarse_global, const int64_t *coarse_counts,
                       int smoother, double omega, int dtype, omg_dist **out);
int omg_dist_destroy(omg_dist *d);
/* Replicated tail: below the last distributed level every rank runs `tail` — an ordinary
 * omg_hierarchy whose finest operator is the WHOLE operator of that level — on the gathered
 * right-hand side and keeps its own slice of the correction; no exchange happens down there.
 * Pass coarse_global = NULL to omg_dist_create when a tail will be set.  `tail` is borrowed. */
int omg_dist_set_tail(omg_dist *d, omg_hierarchy *tail);
int omg_dist_set_stream(omg_dist *d, void *hip_stream);
int omg_dist_sync(omg_dist *d);
/* RCCL bootstrap: rank 0 makes a 128-byte id, the host broadcasts it (torch.distributed),
 * every rank connects.  librccl is dlopen'ed on first use.                                */
int omg_rccl_unique_id(void *out128);
int omg_dist_connect(omg_dist *d, const void *unique_id128);
/* Microseconds per grouped ncclSend + ncclRecv of `bytes` to the calling rank itself on a one-rank
 * communicator, each followed by a small kernel, `reps` back to back on one stream (hipEvents): the floor
 * of a halo exchange on this GPU (no link involved) — tools/exchange_probe.py, DESIGN.md section 7. */
int omg_rccl_self_exchange_time(int64_t bytes, int reps, double *avg_us);
/* Number of ranks of the RCCL communicator this rank is connected to (ncclCommCount); 0 before
 * omg_dist_connect.  bench.py prints it (config.rccl_ranks) as evidence that the N-GPU run
 * really is one N-rank communicator.                                                        */
int omg_dist_rccl_ranks(omg_dist *d, int *count);
/* owned part of b and of the initial iterate (NULL = zeros), natural local numbering */
int omg_dist_load(omg_dist *d, const double *b_local, const double *x0_local);
int omg_dist_fetch(omg_dist *d, double *x_local);
/* One V-cycle over all ranks (collective: every rank calls it).  *norm (nullable) = the
 * GLOBAL ||b - A x||_2 (openmg/__init__.py:227).                                          */
int omg_dist_cycle(omg_dist *d, int pre, int post, double *norm);
/* n_cycles cycles with every cycle's GLOBAL norm computed, norms[n_cycles] (nullable) returned at
 * the end — the multi-GPU form of omg_resident_cycles (same deferral of the first colour's share
 * of the norm into the next cycle's first launches, same bits as n_cycles omg_dist_cycle calls). */
int omg_dist_cycles(omg_dist *d, int pre, int post, int n_cycles, double *norms);
/* Per-GPU measurement helpers of a multi-GPU run: `reps` launches of y = A_0 x over this rank's
 * rows in one hipEvent bracket (average ms per launch), and omg_hierarchy_format_info for this
 * rank's operators.                                                                          */
int omg_dist_spmv_time(omg_dist *d, int reps, double *avg_ms);
int omg_dist_format_info(omg_dist *d, int level, int op, int set, int64_t *out);
/* Schedule of a smoothed level of this rank: OMG_LEVEL_SCATTER_PROLONG, 4 = (boundary, interior)
 * set pairs, 8 = the scatter prolongation is split the same way (boundary aggregates first). */
int omg_dist_level_flags(omg_dist *d, int level, int *flags);
/* Loopback group: ALL ranks of a decomposition inside one process on one GPU, halos moved by
 * device-to-device copies.  Same schedule as omg_dist_cycle; used to verify the distributed
 * algorithm where only one GPU is available.                                               */
int omg_dist_group_create(int n, omg_dist **ranks, omg_dist_group **out);
int omg_dist_group_destroy(omg_dist_group *g);
int omg_dist_group_cycle(omg_dist_group *g, int pre, int post, double *norm);
int omg_dist_group_cycles(omg_dist_group *g, int pre, int post, int n_cycles, double *norms);

/* ---- plane-pipelined slabs (csrc/dist.hip, round 3): the multi-GPU cycle of constant-coefficient grid stencils ----
 * A rank owns nz_global / n_ranks planes of every distributed level (halved per level) plus ghost planes; each half of
 * the cycle over a level is the single-GPU plane-pipelined launch on that slab (csrc/plane.hip), the exchanges are
 * ghost PLANES (three of x before a pass, two of the right-hand side / the coarse correction), the level below the
 * slabs is gathered and run replicated (`tail`, an ordinary hierarchy, borrowed).  coef7: seven coefficients per
 * distributed level (-K, -J, -I, diagonal, +I, +J, +K), weight: the aggregation's.  double, V(1,1).  The iterate is
 * bit-identical to the single-GPU cycle for every number of ranks.                                                  */
typedef struct omg_pdist omg_pdist;
typedef struct omg_pdist_group omg_pdist_group;
int omg_pdist_create(int rank, int n_ranks, int nx, int ny, int nz_global, int n_levels, const double *coef7, double weight,
                     omg_pdist **out);
int omg_pdist_destroy(omg_pdist *d);
int omg_pdist_set_tail(omg_pdist *d, omg_hierarchy *tail);
/* two communicators: the cycle's, and one for the exchanges that run on a second stream beside the coarser levels */
int omg_pdist_connect(omg_pdist *d, const void *unique_id128, const void *unique_id128_side);
int omg_pdist_rccl_ranks(omg_pdist *d, int *count);
int omg_pdist_load(omg_pdist *d, const double *b_local, const double *x0_local /* NULL = zeros */);   /* collective */
int omg_pdist_fetch(omg_pdist *d, double *x_local);
int omg_pdist_sync(omg_pdist *d);
/* out8: distributed levels; 1 when the finest level's passes run GATED between neighbours (one launch of inner chunks and
 * flag-gated edge chunks, the exchange beside it: csrc/dist.hip PlaneDist::gate); its tiling (cells per line, lines,
 * planes per chunk), workgroups, threads per workgroup; planes per inner chunk of a gated pass */
int omg_pdist_info(omg_pdist *d, int64_t *out8);
/* gated passes on / off (off at creation unless OMG_PDIST_GATE=1): the caller switches them on after a checked cycle */
int omg_pdist_set_gate(omg_pdist *d, int enable);
/* omg_pdist_trace(1): the stream writes a progress word (pinned host memory) between the phases of a cycle;
 * omg_pdist_progress reads it without synchronising: (cycle << 16) | (level << 8) | phase, phase 1 halo of x, 2 halo of
 * b, 3 down pass, 4 halo of x for the up pass, 5 gather + replicated tail, 6 halo of the correction, 7 up pass —
 * what a rank reports when a collective never completes (bench.py's preflight).                                   */
int omg_pdist_trace(omg_pdist *d, int enable);
int omg_pdist_progress(omg_pdist *d, unsigned *word);
int omg_pdist_cycles(omg_pdist *d, int n_cycles, double *norms /* nullable */);
/* V(pre, post) with pre, post in {0, 1} — the reference's default is V(1, 0), openmg/__init__.py:22-23 — as the passes
 * without their relaxation; over the RCCL exchanges (peer mode and split passes: V(1, 1) only)                      */
int omg_pdist_cycles_ex(omg_pdist *d, int pre, int post, int n_cycles, double *norms /* nullable */);

/* ---- peer mode: the slab exchanges as stores into the neighbours' memory over xGMI peer mappings ------------------
 * (no reference counterpart: openmg is single-process.)  The passes write their boundary planes straight into the
 * neighbours' ghost planes, raise a flag there when all their workgroups are done, and wait for their neighbours'
 * flags before reading a ghost plane: no exchange launches inside a cycle.  Setup: every rank exports
 * omg_pdist_p2p_handle_count() IPC handles of 64 bytes (omg_pdist_p2p_handles), the control plane hands them round,
 * every rank opens every other rank's (omg_pdist_p2p_open; ranks of one process: omg_pdist_p2p_local), then
 * omg_pdist_p2p_enable(mode) — 1: the passes wait themselves (one GPU per rank); 2: one-workgroup wait launches
 * (ranks sharing a GPU); 0: back to RCCL.  Every distributed level needs >= 4 planes per rank.  A wait that gives up
 * (OMG_P2P_SPIN polls, default 2^21) sets bit 0 of omg_pdist_p2p_status and lets the device run on.
 * omg_pdist_cycles_squares: the cycles without the norm's collective — this rank's sums of squared residuals. */
int omg_peer_access(int device, int peer_device, int *can);     /* hipDeviceCanAccessPeer (same device: 1): ask before mapping */
int omg_pdist_p2p_handle_count(omg_pdist *d, int *count);
int omg_pdist_p2p_handles(omg_pdist *d, void *handles64, int capacity);
int omg_pdist_p2p_open(omg_pdist *d, int peer_rank, const void *handles64, int count);
int omg_pdist_p2p_local(omg_pdist *d, omg_pdist *other);
int omg_pdist_p2p_enable(omg_pdist *d, int mode);
int omg_pdist_p2p_status(omg_pdist *d, unsigned *status);
int omg_pdist_cycles_squares(omg_pdist *d, int n_cycles, double *squares);                         /* collective */
/* all ranks in one process on one GPU, device copies in place of RCCL (verification) */
int omg_pdist_group_create(int n, omg_pdist **ranks, omg_pdist_group **out);
int omg_pdist_group_destroy(omg_pdist_group *g);
int omg_pdist_group_cycles(omg_pdist_group *g, int n_cycles, double *norms /* nullable */);
int omg_pdist_group_cycles_ex(omg_pdist_group *g, int pre, int post, int n_cycles, double *norms /* nullable */);

/* ---- 27-point slabs (csrc/dist27.hip, round 5): the multi-GPU cycle of 27-point grid stencils with per-row
 * coefficients (BASELINE configs[4]) on the kernels of csrc/stencil27.hip.  (No reference counterpart: openmg is
 * single-process; the cycle is openmg/__init__.py:151-236.)  A rank owns nz_global / n_ranks planes of each of the
 * n_levels distributed levels (halved per level, even on every one) plus one ghost AGGREGATE plane on either side; the
 * sweeps relax colours 0 .. 3 of the upper ghost plane redundantly, so ONE exchange of ghost planes serves a whole
 * 8-colour sweep: per V(p, q) cycle p + q exchanges on the finest level, 1 + p + q on the others, one all-gather above
 * the replicated `tail` (an ordinary hierarchy over the levels below the slabs, borrowed), one all-reduce per batch.
 * A_rows: this rank's rows of the finest operator (its planes, C order) with GLOBAL column indices — every row the full
 * 27-point stencil of its in-grid neighbours, ascending columns; the Galerkin products of the distributed levels are
 * made per rank on the device.  omg_sdist_coarse_*: this rank's rows (global columns) of the operator below the
 * slabs — the control plane gathers them and builds the tail.  dtype: OMG_DTYPE_F64 / OMG_DTYPE_F32 (vectors,
 * coefficients and messages).  The iterate is bit-identical to the single-GPU hierarchy's for every number of ranks
 * (tests/test_gpu_dist27.py). */
typedef struct omg_sdist omg_sdist;
typedef struct omg_sdist_group omg_sdist_group;
int omg_sdist_create(int rank, int n_ranks, int nx, int ny, int nz_global, int n_levels, const omg_csr *A_rows, double weight, int dtype,
                     omg_sdist **out);
int omg_sdist_destroy(omg_sdist *d);
int omg_sdist_coarse_size(omg_sdist *d, int64_t *n_rows, int64_t *n_cols, int64_t *nnz);
int omg_sdist_coarse_fetch(omg_sdist *d, int32_t *indptr, int32_t *indices, double *data);
int omg_sdist_set_tail(omg_sdist *d, omg_hierarchy *tail);
/* joins the communicator; collective: also fetches the neighbour's coefficient rows of the ghost planes */
int omg_sdist_connect(omg_sdist *d, const void *unique_id128);
int omg_sdist_rccl_ranks(omg_sdist *d, int *count);
/* out8: nx, ny, owned planes, aggregates per lane, workgroups, waves per workgroup of `level`; distributed levels;
 * halo exchanges the last omg_sdist_cycles call enqueued on this rank */
int omg_sdist_info(omg_sdist *d, int level, int64_t *out8);
int omg_sdist_load(omg_sdist *d, const double *b_local, const double *x0_local /* NULL = zeros */);
int omg_sdist_fetch(omg_sdist *d, double *x_local);
int omg_sdist_sync(omg_sdist *d);
/* n_cycles V(pre, post) cycles (openmg/__init__.py:22-23: the reference's default is V(1, 0)), every cycle's GLOBAL
 * residual norm (:227) computed and returned; collective */
int omg_sdist_cycles(omg_sdist *d, int pre, int post, int n_cycles, double *norms /* nullable */);
/* all ranks in one process on one GPU, device copies in place of RCCL (verification) */
/* Peer mode for the halo exchanges of the 27-point slabs (round 6): stores into the neighbours' ghost planes ordered by
 * flags, in place of the grouped ncclSend / ncclRecv launches — the calls of omg_pdist_p2p_* with 1 + 3 per level handles
 * (flag words, then x / tmp / b of every level); a rank opens its two NEIGHBOURS' only.  The gather below the slabs and the
 * norm's reduction stay on the communicator (omg_sdist_connect).  status bit 0: a bounded wait (OMG_P2P_SPIN polls) gave up. */
int omg_sdist_p2p_handle_count(omg_sdist *d, int *count);
int omg_sdist_p2p_handles(omg_sdist *d, void *handles64, int capacity);
int omg_sdist_p2p_open(omg_sdist *d, int peer_rank, const void *handles64, int count);
int omg_sdist_p2p_local(omg_sdist *d, omg_sdist *other);
int omg_sdist_p2p_enable(omg_sdist *d, int mode /* 0 | 1 */);
int omg_sdist_p2p_status(omg_sdist *d, unsigned *status);
int omg_sdist_group_create(int n, omg_sdist **ranks, omg_sdist_group **out);
int omg_sdist_group_destroy(omg_sdist_group *g);
int omg_sdist_group_cycles(omg_sdist_group *g, int pre, int post, int n_cycles, double *norms /* nullable */);

#ifdef __cplusplus
}
#endif
#endif /* OPENMG_HIP_H */
