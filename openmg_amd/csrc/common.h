// Shared declarations for libopenmg_hip.so (MI355X / gfx950 only).
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <stdexcept>
#include <memory>
#include <string>
#include <utility>
#include <vector>

#include "../../include/openmg_hip.h"

// Compile-time tuning switches (each a measured choice, docs/HISTORY.md): overriding one is an EXPERIMENT and has to say so —
//     make EXTRA="-DOMG_EXPERIMENTS -DS27_SG=9"
#if !defined(OMG_EXPERIMENTS) && (defined(PLANE_STORE_AUX) || defined(PLANE_LA_BIG) || defined(PLANE_LA) || defined(PLANE_WAVE_SYNC) || \
                                  defined(PLANE_WSYNC_F32) || defined(S27_COEF_AUX) || defined(S27_SG) || defined(S27_WAVES) ||        \
                                  defined(MARCH_NAP_MAX) || defined(OMG_ROWBLK_NNZ) || defined(PLANE_SETPRIO))
#error "tuning switches are experiments: add -DOMG_EXPERIMENTS"
#endif

namespace omg {

// Run-time switches come in two kinds.  The SUPPORTED ones (INTEGRATION.md lists them: paths a test, a tool or the bench
// selects) are read with getenv.  The switches of past tuning experiments are read through experiment_env: in an ordinary
// build that is nullptr — the shipped default — and only a library built with -DOMG_EXPERIMENTS listens to them.
inline const char *experiment_env(const char *name) {
#ifdef OMG_EXPERIMENTS
    return getenv(name);
#else
    (void)name;
    return nullptr;
#endif
}
inline bool experiment_flag0(const char *name) { const char *e = experiment_env(name); return e && e[0] == '0'; }   // switched OFF

// ---- errors -------------------------------------------------------------------------
struct Error : std::runtime_error {
    int code;
    Error(int c, const std::string &m) : std::runtime_error(m), code(c) {}
};
void set_last_error(const std::string &m);

#define OMG_HIP(call)                                                                       \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess)                                                               \
            throw omg::Error(e_ == hipErrorOutOfMemory ? OMG_ERR_ALLOC : OMG_ERR_HIP,       \
                             std::string(#call) + ": " + hipGetErrorString(e_));            \
    } while (0)

#define OMG_REQUIRE(cond, msg)                                                              \
    do {                                                                                    \
        if (!(cond)) throw omg::Error(OMG_ERR_INVALID, std::string(msg));                   \
    } while (0)

void require_device();   // throws OMG_ERR_NO_DEVICE when no GPU is visible

// OMG_SETUP_TIMING=1: phases of the hierarchy setup, with their wall time, on stderr
struct SetupTimer {
    const char *what;
    double t0;
    static bool on() { static const bool v = [] { const char *e = getenv("OMG_SETUP_TIMING"); return e && e[0] == '1'; }(); return v; }
    static double now();
    explicit SetupTimer(const char *w) : what(w), t0(on() ? now() : 0.0) {}
    ~SetupTimer() { if (on()) fprintf(stderr, "[omg setup] %-34s %8.1f ms\n", what, 1e3 * (now() - t0)); }
};

// ---- device buffers -----------------------------------------------------------------
// Plain hipMalloc RAII.  Every allocation gets 64 bytes of slack on BOTH sides so that the
// 16-byte vector loads of the row kernels may run past the logical end of an array, and the
// paired gathers of rows_union_kernel (x[c], x[c + 1] in one load, c = -1 for a first row whose
// neighbour does not exist) may start one element in front of it.
constexpr size_t DEVBUF_SLACK = 64;
// Device memory whose 2 MiB pieces are separate physical allocations, mapped into ONE virtual range in a shuffled order
// (HIP's virtual-memory API).  Why anybody would want that: hierarchy.hip place_finest_pool — where the large vectors of
// the finest level land PHYSICALLY decides between pass speeds 7 % apart, a physically contiguous allocation is the worst
// case (another 12 %), and what the driver hands out for an ordinary hipMalloc lies somewhere in between, differently in
// every process.  scattered_alloc returns nullptr where the API is not available (the caller allocates ordinarily).
struct ScatteredBlock {
    void *va = nullptr;
    size_t total = 0;
    size_t chunk = 0;                                           // every hipMemMap of the range covers this many bytes
    std::vector<hipMemGenericAllocationHandle_t> handles;
};
std::vector<ScatteredBlock> &scattered_registry();
void *scattered_alloc(size_t bytes, size_t chunk_bytes);
bool scattered_free(void *va);                                  // false: not one of ours

template <typename T>
struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    size_t shift = 0;
    bool owned = true;               // false: a view into another buffer (borrow())
    DevBuf() = default;
    explicit DevBuf(size_t count) { alloc(count); }
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n), shift(o.shift), owned(o.owned) { o.p = nullptr; o.n = 0; o.shift = 0; o.owned = true; }
    DevBuf &operator=(DevBuf &&o) noexcept {
        if (this != &o) { release(); p = o.p; n = o.n; shift = o.shift; owned = o.owned; o.p = nullptr; o.n = 0; o.shift = 0; o.owned = true; }
        return *this;
    }
    // count elements at ptr, inside an allocation somebody else owns and keeps alive (with the slack this type promises)
    // (shift_bytes: how far behind the owner's first element the view starts — the allocation's base is then
    // p - DEVBUF_SLACK - shift for views as for owners: dist.hip's IPC export)
    void borrow(T *ptr, size_t count, size_t shift_bytes = 0) {
        release();
        p = ptr;
        n = count;
        shift = shift_bytes;
        owned = false;
    }
    ~DevBuf() { release(); }
    // shift (bytes, a multiple of 64): the vector starts that much further into its allocation — large vectors that a
    // kernel streams side by side are staggered so that they do not walk the memory channels in lockstep
    // placement: 0 ordinary; 1 physically contiguous (hipDeviceMallocContiguous: an experiment, the slowest there is);
    // >= 2: scattered pieces of that many MiB (scattered_alloc); either falls back to an ordinary allocation
    void alloc(size_t count, size_t shift_bytes = 0, int placement = 0) {
        release();
        n = count;
        char *raw = nullptr;
        const size_t total = count * sizeof(T) + 2 * DEVBUF_SLACK + shift_bytes;
        if (placement == 1 && hipExtMallocWithFlags(reinterpret_cast<void **>(&raw), total, hipDeviceMallocContiguous) != hipSuccess) {
            (void)hipGetLastError();
            raw = nullptr;
        }
        if (placement >= 2) raw = static_cast<char *>(scattered_alloc(total, size_t(placement) << 20));
        if (!raw) OMG_HIP(hipMalloc(reinterpret_cast<void **>(&raw), total));
        shift = shift_bytes;
        p = reinterpret_cast<T *>(raw + DEVBUF_SLACK + shift_bytes);
        // OMG_POISON=1 (tests): fresh device memory holds NaN patterns instead of whatever the allocator hands out (zeros
        // in a young process, anything later) — a kernel that reads what nobody wrote shows in the results
        // (OMG_POISON=2: only the slack around the arrays; 3: only the arrays)
        static const int poison = [] { const char *e = getenv("OMG_POISON"); return e ? atoi(e) : 0; }();
        if (poison == 1) OMG_HIP(hipMemset(raw, 0xFF, count * sizeof(T) + 2 * DEVBUF_SLACK + shift_bytes));
        if (poison == 2) {
            OMG_HIP(hipMemset(raw, 0xFF, DEVBUF_SLACK + shift_bytes));
            OMG_HIP(hipMemset(raw + DEVBUF_SLACK + shift_bytes + count * sizeof(T), 0xFF, DEVBUF_SLACK));
        }
        if (poison == 3 && count) OMG_HIP(hipMemset(p, 0xFF, count * sizeof(T)));
        if (poison) OMG_HIP(hipDeviceSynchronize());      // (the fill is in place before any stream writes the array)
    }
    void release() {
        if (p && owned) {
            char *const raw = reinterpret_cast<char *>(p) - DEVBUF_SLACK - shift;
            if (!scattered_free(raw)) (void)hipFree(raw);
        }
        p = nullptr;
        n = 0;
        shift = 0;
        owned = true;
    }
    void upload(const T *host, size_t count, hipStream_t s) {
        if (count) OMG_HIP(hipMemcpyAsync(p, host, count * sizeof(T), hipMemcpyHostToDevice, s));
    }
    void download(T *host, size_t count, hipStream_t s) const {
        if (count) OMG_HIP(hipMemcpyAsync(host, p, count * sizeof(T), hipMemcpyDeviceToHost, s));
    }
    void zero(hipStream_t s) { if (n) OMG_HIP(hipMemsetAsync(p, 0, n * sizeof(T), s)); }
};

// How the k-th candidate of a placement search (hierarchy.hip place_finest_pool, Stencil27Plan::place_tiles, ...) is
// allocated: 0 ordinary hipMalloc, n >= 2: scattered pieces of n MiB (DevBuf::alloc; 1: physically contiguous — an
// experiment).  OMG_POOL_PLACE=n: every candidate that way.  The order (round 6, profiles/r06_pool_piece_size.txt): where the
// ordinary allocation is the slow kind for the plane passes, pieces of 2, 4 and 8 MiB are fast for them on the first try (173 us
// per down + up against 191), 16 and 32 MiB pieces once in three to six; the level's other stream over x, the matrix-free
// SpMV, reads an x that lies in 2 or 4 MiB pieces at 68 - 72 us per launch whatever its destination, one in 8 MiB pieces at 44
// us in four processes of five (all of a process's 8 MiB pools alike), one in 16 / 32 MiB pieces or an ordinary allocation
// at 44.  place_finest_pool asks both: 8 MiB pieces first, then 16 / 32 among them.
inline int pool_placement(int k) {
    const char *e = getenv("OMG_POOL_PLACE");
    if (e && e[0]) return atoi(e);
    static const int kinds[] = {0, 8, 16, 8, 32, 16, 2, 32};
    return kinds[size_t(k) % (sizeof(kinds) / sizeof(kinds[0]))];
}

// The vectors a plane pass streams side by side (x_old, x_new, b: plane.hip) start k * 256 bytes into their allocations
// (hipMalloc returns them all at the same offset of a 2 MiB page, so at every index they would otherwise sit on the
// same memory channel): measured at 256^3 over five processes each, down pass 103.4-104.1 us in four of five with
// 256 bytes against 106-111 us (two populations) with 0, 128, 512, 1 Ki, 2 Ki, 4 Ki, 64 Ki or 1 Mi.  OMG_VEC_STAGGER
// overrides the 256.
inline size_t vector_stagger(int k) {
    static const long bytes = [] { const char *e = experiment_env("OMG_VEC_STAGGER"); return e ? atol(e) : 256L; }();
    return size_t(bytes > 0 ? bytes : 0) / 64 * 64 * size_t(k);
}

// ---- host-side CSR (setup only) -----------------------------------------------------
// Allocator whose construct() default-initialises: resize() of a vector of ints / doubles then
// leaves the memory untouched instead of zero-filling it on one thread (1.4 GB for the 256^3
// operator) — the parallel loops that fill the arrays do the first touch.
template <typename T>
struct NoInitAlloc : std::allocator<T> {
    template <typename U> struct rebind { typedef NoInitAlloc<U> other; };
    NoInitAlloc() = default;
    template <typename U> NoInitAlloc(const NoInitAlloc<U> &) {}
    template <typename U> void construct(U *p) noexcept { ::new (static_cast<void *>(p)) U; }
    template <typename U, typename... Args> void construct(U *p, Args &&...args) { ::new (static_cast<void *>(p)) U(std::forward<Args>(args)...); }
};
typedef std::vector<int32_t, NoInitAlloc<int32_t>> IndexVec;
typedef std::vector<double, NoInitAlloc<double>> ValueVec;

struct HostCsr {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    IndexVec indptr, indices;
    ValueVec data;
};

// A smoother ordering: rows renumbered so that each independent set is a contiguous range.
//   perm[new] = old;  sets = offsets into the new numbering (size n_sets + 1).
struct Ordering {
    std::vector<int32_t> perm, inv;
    std::vector<int64_t> sets;
    bool identity = true;
    // A closed-form ordering of an nx x ny x nz grid whose host arrays have not been written (setup on the device):
    // 2 = red-black by parity (plane levels), 8 = octants (27-point levels); materialise_ordering() fills perm / inv
    int closed_form = 0, cf_nx = 0, cf_ny = 0, cf_nz = 0;
};
void materialise_ordering(Ordering &ord);                     // (setup_host.cpp) no-op unless perm / inv are pending
// perm (slot -> natural row) of a closed-form ordering, and for a coarse level with such an ordering the map natural
// index -> slot, written on the device (setup_device.hip)
void fill_ordering_device(int kind, int nx, int ny, int nz, int32_t *perm, int32_t *inv, hipStream_t s);

// setup_host.cpp
// Touch every page of a freshly allocated host buffer that is about to RECEIVE a large device-to-host copy, on many
// threads: the copy into untouched pages of a new NumPy array runs at page-fault speed (175 MB in 33 ms, measured)
void prefault_host(void *p, size_t bytes);
// Large device-to-host copy into PAGEABLE memory (a caller's NumPy array): the runtime's own path ran at 5-12 GB/s
// (175 MB in 33 ms).  Here: chunks through two pinned staging buffers (kept for the process), the copy of chunk i + 1
// over PCIe while several host threads move chunk i into the destination.  Synchronous: returns when `host` is
// complete; everything enqueued on `s` before the call is waited for.  (setup_host.cpp)
void download_staged(void *host, const void *dev, size_t bytes, hipStream_t s);
void validate_csr(const omg_csr &A, const char *what);
// Smallest row whose stored diagonal entries are missing or sum to zero; -1 if none.
int64_t first_row_without_diagonal(const omg_csr &A);
Ordering make_ordering(const omg_csr &A, int smoother);
// col_inv relabels only columns < n_inv (n_inv < 0: all of them); a distributed level keeps
// its halo columns (>= number of owned rows) where they are.
HostCsr permute_csr(const omg_csr &A, const int32_t *row_perm /* new->old or null */,
                    const int32_t *col_inv /* old->new or null */, int64_t n_inv = -1);
// Ordering from caller-supplied set keys (0 <= key < n_sets) instead of a local colouring:
// distributed runs need keys that agree across ranks.  Empty sets are kept.
Ordering ordering_from_keys(const int32_t *keys, int64_t n, int32_t n_sets);
HostCsr transpose_csr(const HostCsr &A);
// Row blocks for the streaming kernels: greedy split of each set into blocks of at most
// `max_rows` rows and `max_nnz` entries (a single longer row gets a block of its own).
void make_row_blocks(const IndexVec &indptr, const std::vector<int64_t> &sets,
                     int max_rows, int max_nnz, std::vector<int32_t> &blk_rows,
                     std::vector<int64_t> &set_blk);

// ---- device CSR ---------------------------------------------------------------------
constexpr int ROWBLK_THREADS = 256;   // threads per workgroup of the row kernels
constexpr int ROWBLK_ROWS = 256;      // <= one row per thread
#ifndef OMG_ROWBLK_NNZ
#define OMG_ROWBLK_NNZ 2048
#endif
constexpr int ROWBLK_NNZ = OMG_ROWBLK_NNZ;   // entries staged through LDS per workgroup
constexpr int DICT_MAX = 64;                 // entries of a block dictionary (power of two)
constexpr int DICT_SHIFT = 7;                // table word = (pool offset << DICT_SHIFT) | entries
// How a row's sum is associated — a property of the ROW, so that every kernel, partition and
// rank count gives the same bits: rows of at most ASSOC_LEN stored entries are one fma chain in
// stored order (the reference's order, openmg/solvers.py:63-65); longer rows are four chains —
// chain q takes entries q, q+4, q+8, ... in stored order — added as ((s0 + s1) + s2) + s3.
// (Four lanes can then share a long row, and a single thread keeps four fmas in flight.)
constexpr int ASSOC_LEN = 16;
constexpr int PAT_LANE_ENTRIES = 128;         // pattern-kernel dictionary: entries held in the lanes of a wave (2 registers)
constexpr int MIN_CODED_ENTRIES = 64;         // smaller blocks stay plain CSR
constexpr int BLK_INFO_INTS = 8;             // ints per row-block table record
// Union walk (rows_union_kernel): when every row pattern of a block is a SUBSEQUENCE of one short
// sequence of (offset, value) pairs — a stencil's boundary rows drop entries of the interior row,
// the two parities of a red-black ordering interleave — the block's dictionary also carries that
// common supersequence (at most UNION_MAX entries, behind its pattern entries) and one bit mask
// per pattern (behind its pattern starts).  A wave then walks the union ONCE with scalar operands
// for all its rows, each lane skipping the slots its mask lacks: stored order, hence the bits,
// unchanged, and no per-pattern grouping of the lanes.
constexpr int UNION_MAX = 16;

// Host image of the device format of one operator (setup_host.cpp:encode_csr): row-block
// table, code arrays and dictionary pools exactly as they are uploaded.
template <typename V>
struct HostFormat {
    std::vector<int64_t> sets, set_blk, set_nnz;
    std::vector<int32_t> set_maxlen;
    std::vector<char> set_pattern, set_ell;
    std::vector<char> set_union;                // per set: every block carries a union (rows_union_kernel can run it)
    int union_max = 0;                          // longest union of any block
    int union_blocks = 0;                       // 1: several-rows-per-thread partition (blocks of rows_cap > 256 rows, all with unions)
    int rows_cap = 256, lanes_per_row = 1;
    std::vector<V> narrowed;                    // float operators: the entries rounded once
    std::vector<uint8_t> cc, vc, rc;            // per-entry column / value codes, per-row pattern codes
    std::vector<int32_t> cpool, ppool_idx, ppool_beg, info;
    std::vector<V> vpool, ppool_val;
    std::vector<V> vell;                        // block-transposed values of the offset-pattern blocks
    int64_t blocks_ccoded = 0, blocks_vcoded = 0, blocks_pcoded = 0, blocks_ell = 0;
    int64_t nnz_ccoded = 0, nnz_vcoded = 0, nnz_pcoded = 0, rows_pcoded = 0, nnz_ell = 0;
    bool wide_failed = false;                   // encode_csr's 256-row attempt for long-row operators did not qualify
    bool union_failed = false;                  // encode_csr's several-rows-per-thread attempt (union walk) did not qualify
};
template <typename V>
HostFormat<V> encode_csr(const HostCsr &A, const std::vector<int64_t> &sets);
// encode -> decode -> compare bit for bit (throws on a mismatch); out: OMG_FORMAT_FIELDS statistics
template <typename V>
void format_selftest(const omg_csr &A, int64_t *out);

// V = value type of the stored entries and of the vectors the operator is applied to:
// double (the reference's precision) or float (BASELINE configs[4]); indices are int32.
template <typename V>
struct DevCsrT {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    DevBuf<int32_t> indptr, indices;
    DevBuf<V> data;
    // Row-block table, BLK_INFO_INTS ints per block (+ end sentinel): [0] first row, [1] first
    // entry, [2] / [3] the block's column / value dictionary as (pool offset << DICT_SHIFT) |
    // entries (0 = none), [4..7] its row-pattern dictionary: entry offset into pidx / pval,
    // entries, offset into pbeg, patterns (0 = none; [6] is then 1 when every row of the
    // block holds exactly one entry, which lets the kernels skip the row pointers).
    // In a row-pattern block [2] / [3] mean something else: [2] != 0: offset patterns, values in
    // vell at [2] - 1 and [3] = the block's longest row; [2] == 0: [3] = entries of the block's
    // UNION (0 = none), which sits at pidx / pval [4] + [5] .. and its masks at pbeg [6] + [7] + 1 ..
    DevBuf<int32_t> blk_rows;
    // Block-dictionary coding (lossless; DESIGN.md "Device format").  Stencil-like operators
    // repeat a handful of (column - row) offsets and of values inside a row block: such a
    // block keeps the distinct ones in a dictionary of at most DICT_MAX entries and one byte
    // per stored entry for each, and the kernels rebuild the int32 column / V value in LDS —
    // 2 bytes from HBM per entry instead of 4 + sizeof(V).  Blocks that do not fit (irregular
    // sparsity, variable coefficients) read the plain arrays; the choice is per block and
    // separate for columns and values.  The plain arrays above are always present.
    //
    // Row patterns go one step further for operators whose ROWS repeat (constant-coefficient
    // stencils, Galerkin products of them): the block's dictionary holds each distinct row —
    // its (column - row) offsets and values in stored order — once, and HBM carries one byte
    // per ROW.  The kernels stream the dictionary (a few hundred bytes, shared between blocks,
    // L2-resident) into the same LDS image a plain block would fill and walk it per row, so the
    // summation order, and with it every bit of the result, is unchanged.
    //
    // Variable coefficients repeat the OFFSETS of their rows but not the values.  Where a whole
    // smoother set can run the LDS-free pattern kernel, such a block keeps offset-only patterns
    // (one byte per row) and its values move to `vell`, transposed inside the block: entry j of
    // every row side by side, so that the one-thread-per-row kernel reads them coalesced
    // (record word [2] = offset into vell + 1, [3] = the block's longest row).  sizeof(V) bytes
    // per entry instead of sizeof(V) + 1 (+ padding for the shorter boundary rows).
    DevBuf<uint8_t> ccode, vcode;      // one byte per stored entry (meaningful in coded blocks)
    DevBuf<int32_t> cdict;             // pool of column-offset dictionaries (shared between blocks)
    DevBuf<V> vdict;                   // pool of value dictionaries
    DevBuf<uint8_t> rcode;             // one byte per row: its pattern (meaningful in pattern blocks)
    DevBuf<int32_t> pidx, pbeg;        // pattern pools: offsets; per dictionary npat + 1 starts into them
    DevBuf<V> pval;                    //                values
    DevBuf<V> vell;                    // block-transposed values of offset-pattern blocks
    int64_t blocks_ell = 0, nnz_ell = 0;
    int64_t blocks_ccoded = 0, blocks_vcoded = 0, blocks_pcoded = 0;   // blocks using each coding (host, stats)
    int64_t nnz_ccoded = 0, nnz_vcoded = 0, nnz_pcoded = 0;            // their stored entries
    int64_t rows_pcoded = 0;
    // per set (host): 0 = rows_kernel; 1 = every block is row-pattern coded with a wave-sized
    // dictionary: rows_pattern_kernel (OMG_PATTERN_KERNEL=0 may still choose rows_kernel); 2 = the
    // same, and rows_kernel could not run it (values in vell, or 256-row blocks of long rows)
    std::vector<char> set_pattern;
    std::vector<int32_t> blk_host;     // host copy of the row-block table (format statistics)
    // out[OMG_FORMAT_FIELDS] of include/openmg_hip.h for the blocks of set `set` (-1: all)
    void format_info(int set, int64_t *out) const;
    std::vector<int64_t> set_blk;      // block offsets of the independent sets (host)
    std::vector<int64_t> sets;         // row offsets of the sets (host)
    int rows_cap = ROWBLK_ROWS;        // most rows a block may hold (> ROWBLK_THREADS: short rows)
    std::vector<int64_t> set_nnz;      // stored entries of each set (host)
    std::vector<int32_t> set_maxlen;   // longest row of each set (host): > ASSOC_LEN picks the four-chain kernels
    std::vector<char> set_ell;         // per set: some block keeps its values in vell (host)
    std::vector<char> set_union;       // per set: every block carries a union -> rows_union_kernel (host)
    int union_max = 0;                 // longest union of any block (picks the kernel instantiation)
    int union_blocks = 0;              // 1: blocks of rows_cap > 256 rows made for rows_union_kernel (not SHORT rows)
    int lanes_per_row = 1;             // 4 for operators with long rows (avg > 16 entries)
    void upload(const HostCsr &A, const std::vector<int64_t> &sets, hipStream_t s);   // converts to V
    size_t n_sets() const { return sets.empty() ? 0 : sets.size() - 1; }
    bool all_pattern() const {         // every (non-empty) set can run rows_pattern_kernel
        if (set_pattern.empty() || union_blocks) return false;
        for (size_t q = 0; q < set_pattern.size(); ++q)
            if (!set_pattern[q] && sets[q + 1] > sets[q]) return false;
        return true;
    }
    // a level operator: square, or one rank's rows of it with the halo columns behind the owned ones
    bool operator_like() const { return n_cols >= n_rows && n_cols < 2 * n_rows; }
    bool all_union() const {           // every (non-empty) set runs rows_union_kernel (csr_kernels.hip launch_rows_range)
        if (set_union.empty() || !operator_like() || size_t(n_cols) * sizeof(V) >= (size_t(1) << 31)) return false;
        for (size_t q = 0; q < set_union.size(); ++q)
            if (!set_union[q] && sets[q + 1] > sets[q]) return false;
        return true;
    }
    int64_t n_blocks() const { return set_blk.empty() ? 0 : set_blk.back(); }
};
using DevCsr = DevCsrT<double>;

// ---- kernel launchers (csr_kernels.hip) ---------------------------------------------
enum RowMode : int {
    ROW_SPMV = 0,       // y = A x
    ROW_RESIDUAL = 1,   // y = b - A x
    ROW_RESNORM = 2,    // y = b - A x, partial sums of y^2 per block
    ROW_GS = 3,         // x += (b - A x) / diag      (in place, rows of one set)
    ROW_JACOBI = 4,     // y = x + omega (b - A x) / diag
    ROW_AXPY = 5,       // y += A x
    ROW_NORM_ONLY = 6,  // partial sums of (b - A x)^2, nothing stored
    // Gauss-Seidel sweep of the LAST set of a sweep, fused with the residual of the rows it
    // has just relaxed.  Rows of a set are mutually uncoupled, so after its own update a
    // row's residual b_i - sum_j a_ij x_j depends only on values no other row of this launch
    // changes: it is recomputed from the entries already in LDS with x_i replaced by its new
    // value — bit for bit what a separate residual pass would produce — and that pass then
    // only has to visit the rows of the other sets.
    ROW_GS_RES = 7,     // ... residual stored to `zero`-slot pointer aux
    ROW_GS_NORM = 8,    // ... residual squared into the block partials
    // y[col] += a_rc * x[row] for every stored entry: the TRANSPOSE applied from the rows of
    // the operator — prolongation straight from the restriction's row patterns, no explicit
    // R^T.  Only for operators whose columns each hold at most one entry (aggregation: no two
    // rows write the same y) and whose blocks are all pattern coded (rows_pattern_kernel).
    ROW_SCATTER = 9,
    // First launch of a cycle, also finishing the PREVIOUS cycle's residual norm: a Gauss-Seidel
    // sweep of the first of two colour sets, or a Jacobi sweep, forms b_i - sum_j a_ij x_j with the
    // iterate the previous cycle left — exactly the residual its norm launch would compute for
    // these rows (same chain, same bits) — so the launch also leaves the squares in the block
    // partials and the separate norm launch over these rows is not needed
    // (hierarchy.hip omg_resident_cycles).
    ROW_GS_PRENORM = 10,      // x += (b - A x) / diag, block partials of (b - A x)^2
    ROW_JACOBI_PRENORM = 11,  // y = x + omega (b - A x) / diag, block partials of (b - A x)^2
};
constexpr bool mode_fused(int m) { return m == ROW_GS_RES || m == ROW_GS_NORM; }
constexpr bool mode_prenorm(int m) { return m == ROW_GS_PRENORM || m == ROW_JACOBI_PRENORM; }
constexpr bool mode_relaxes(int m) { return m == ROW_GS || m == ROW_JACOBI || mode_fused(m) || mode_prenorm(m); }   // needs x_i and the diagonal
constexpr bool mode_norm(int m) { return m == ROW_RESNORM || m == ROW_NORM_ONLY || m == ROW_GS_NORM || mode_prenorm(m); }

template <typename V>
struct RowArgsT {
    const V *x = nullptr;        // gathered vector
    const V *b = nullptr;
    V *y = nullptr;              // output (GS: the same pointer as x)
    double omega = 1.0;
    double *partials = nullptr;  // one DOUBLE per row block (RESNORM / NORM_ONLY), any V
    V *zero = nullptr;           // ROW_SPMV: zero[r] = 0 alongside y[r] (fused clear);
                                 // ROW_GS_RES: residual of the relaxed rows goes here
    const int32_t *ymap = nullptr;   // ROW_SPMV: result of row r goes to y[ymap[r]]
    // ROW_SPMV with `zero` (restriction: y = next level's right-hand side, zero = its initial
    // iterate): instead of 0, the slots below first_end get what the next level's FIRST smoothing
    // launch would make of a zero iterate — 0 + (y - 0) / diag for the first Gauss-Seidel set,
    // 0 + omega ((y - 0) / diag) for a Jacobi sweep (every entry of the sum is v * 0) — and that
    // launch is skipped.  diag: the diagonal of the next level's operator (its ordering).
    const V *first_diag = nullptr;
    int first_end = 0;
    bool first_jacobi = false;
};
using RowArgs = RowArgsT<double>;

// Launch `mode` over row set `set` of A (set < 0: all sets in one launch).
// (all launchers are instantiated for V = double and V = float in csr_kernels.hip)
template <typename V>
void launch_rows(const DevCsrT<V> &A, int mode, int set, const RowArgsT<V> &args, hipStream_t s);
// ... over the consecutive sets [set_begin, set_end) in one launch.
template <typename V>
void launch_rows_range(const DevCsrT<V> &A, int mode, int set_begin, int set_end, const RowArgsT<V> &args,
                       hipStream_t s);
// Consecutive single-block sets [set_begin, set_end) of a Gauss-Seidel sweep, run back to
// back by one workgroup (workgroup barrier between sets).
template <typename V>
void launch_gs_serial(const DevCsrT<V> &A, int set_begin, int set_end, const RowArgsT<V> &args,
                      hipStream_t s);
// sum of partials[0..n) -> *out (device), fixed order => deterministic.  `partials` must
// have SUM_FOLD spare doubles behind its n entries (stage-1 scratch).
constexpr int SUM_FOLD = 64;
void launch_sum(double *partials, int64_t n, double *out, hipStream_t s);
void launch_sum_sqrt(double *partials, int64_t n, double *out, hipStream_t s);
// `count` arrays of n partials, `stride` doubles apart, summed by ONE launch into out[count] with
// exactly the additions of launch_sum / launch_sum_sqrt (same bits); no scratch needed
void launch_sum_batch(const double *partials, int64_t stride, int64_t n, int count, double *out, bool take_sqrt,
                      hipStream_t s);
// dst[i] = src[idx[i]] / dst[idx[i]] = src[i]; idx == NULL is the identity, S -> D converts
// (the host boundary of a float hierarchy is double)
template <typename S, typename D>
void launch_gather(const S *src, const int32_t *idx, D *dst, int64_t n, hipStream_t s);
template <typename S, typename D>
void launch_scatter(const S *src, const int32_t *idx, D *dst, int64_t n, hipStream_t s);
// diag[i] = 0 + the stored entries (i, i) of row i in stored order, summed in V — the diagonal as the
// relaxation kernels form it (plain CSR arrays of the operator, which stay on the device)
template <typename V>
void launch_diagonal(const DevCsrT<V> &A, V *diag, hipStream_t s);
// x[i] = i < first_end ? 0 + scale_q((b[i] - 0) / diag[i]) : 0 — the first relaxation launch of a
// zero iterate (first Gauss-Seidel set, or with jacobi a whole weighted-Jacobi sweep), spelled like
// the row kernels' epilogues
template <typename V>
void launch_first_relaxation(const V *b, const V *diag, V *x, int64_t n, int64_t first_end, bool jacobi, double omega,
                             hipStream_t s);
template <typename V>
void launch_dense_gemv(const V *M, const V *v, V *out, int64_t n, hipStream_t s);    // out = M v (row-major n x n)
template <typename V>
void launch_dense_gemv_rows(const V *M, const V *v, V *out, int64_t rows, int64_t n,
                            hipStream_t s);                                          // M: rows x n slab
// Dense inverse (row-major n x n) of a square device CSR matrix by Gauss-Jordan with
// partial pivoting on the device.  Throws OMG_ERR_SINGULAR / OMG_ERR_UNSUPPORTED (n > 16384).
void dense_inverse_from_csr(const DevCsr &A, double *Minv, hipStream_t s);
// dense.hip building blocks: Gauss-Jordan on a prepared [M | I] (n x 2n, destroyed), and the
// preparation of that matrix from a sub-block of a plain device CSR (maps < 0: not in the block)
void gauss_jordan_inverse(double *W, int64_t n, double *Minv, hipStream_t s, double *keep_ws = nullptr);      // keep_ws: n * n doubles of scratch, or null
void fill_augmented_from_csr(const int32_t *indptr, const int32_t *indices, const double *data, int64_t n_rows,
                             const int32_t *rmap, const int32_t *cmap, int64_t m, double *W, hipStream_t s);

// ---- coarsest-level direct solve (coarse.hip) ----------------------------------------------
// openmg/solvers.py:16-26 calls SuperLU on every cycle, any size.  Here the operator is factored
// ONCE and a solve is two or three massively parallel launches:
//   * operators whose inverse fits the 256 MiB Infinity Cache (n <= 5792 in double), or whose
//     band leaves no room for separators: the explicit inverse, one dense mat-vec;
//   * otherwise SUBSTRUCTURING along the band.  With half-bandwidth w, index ranges of width w
//     placed regularly are separators Gamma that cut the unknowns into P mutually uncoupled
//     interior blocks I_k.  Stored: B_k = A[I_k, I_k]^-1 (dense), the inverse of the Schur
//     complement S = A_GG - sum_k A_GI_k B_k A_I_kG (dense, (P-1) w square), and the sparse
//     couplings.  Solve:  y_k = B_k b_I_k  |  x_G = S^-1 (b_G - A_GI y)  |  x_I_k = y_k - B_k A_I_kG x_G.
//     128^2 5-point (w = 128): 164 MB read per solve instead of 2.1 GB, 0.3 s of setup instead of 9.
//     OMG_COARSE_BLOCKS = 1 forces the inverse, = P > 1 that many blocks.
// Everything is computed in double; a float solver stores the rounded factors.
template <typename V>
struct CoarseSolver {
    int64_t n = 0;
    int P = 1;                        // interior blocks; 1 = explicit inverse; 0 = sine transforms, -1 = block chain (below)
    DevBuf<V> inv;                    // P == 1: n x n
    int64_t n_int = 0, g = 0, w = 0;  // interior / separator unknowns, half-bandwidth
    DevBuf<V> binv, sinv;             // concatenated B_k (row-major m_k x m_k); S^-1 (g x g)
    DevBuf<int64_t> blk_off, binv_off;   // P + 1 interior offsets (permuted numbering); offsets of B_k
    DevBuf<int32_t> perm;             // permuted [I_0 .. I_{P-1} | Gamma] -> original
    DevBuf<int32_t> wg_blk, wg_row;   // workgroup -> (interior block, first local row) for the batched mat-vecs
    int64_t n_wg = 0;
    DevBuf<int32_t> gi_ptr, gi_idx;   // rows Gamma x columns interior (permuted)
    DevBuf<V> gi_val;
    DevBuf<int32_t> ig_ptr, ig_idx;   // rows interior x columns Gamma (local 0..g)
    DevBuf<V> ig_val;
    DevBuf<V> y, xg;                  // work vectors
    // P == 0: fast sine-transform solve — the operator is a constant-coefficient symmetric star stencil on
    // an sx x sy x sz grid with dropped boundary entries (the Galerkin products of the Poisson hierarchy):
    // A = S diag(lambda) S with S = Sz (x) Sy (x) Sx the orthogonal sine matrices, so x = S ((S b) / lambda):
    // one launch of one workgroup, six passes of short dense transforms through LDS — a direct solve like the
    // inverse (no iteration, error at rounding level), 3 us instead of the 134 MB mat-vec's 30
    int sx = 0, sy = 0, sz = 0;
    DevBuf<double> sine, lambda;      // the three tables (sx^2, sy^2, sz^2 doubles, row-major), the n eigenvalues
    size_t bytes = 0;                 // device bytes one solve reads
    // the explicit inverse's work arrays ([A | I], the inverse in double, a copy of A for the check): kept between builds once
    // a hierarchy takes new coefficients (omg_hierarchy_update_fine) — three allocations of hundreds of MB cost ~2 ms per build
    bool retain_workspace = false;
    DevBuf<double> ws_aug, ws_inv64, ws_keep;
    // P == -1: block elimination ALONG THE BAND, factors in HBM — what is left when neither the explicit inverse (n <= 16384)
    // nor substructuring (separators and interior blocks through 48 KB of LDS) applies: with blocks of bs >= half-bandwidth
    // rows the operator is block tridiagonal, D'_0 = D_0, D'_k = D_k - A_{k,k-1} D'_{k-1}^-1 A_{k-1,k}; the K explicit
    // inverses D'_k^-1 (bs x bs each: n bs values in all) are kept.  A solve is 2 K - 1 dependent launches: forward
    // z_k = D'_k^-1 (b_k - A_{k,k-1} z_{k-1}), backward x_k = z_k - D'_k^-1 A_{k,k+1} x_{k+1}.  Limited by memory only.
    int64_t bs = 0;
    int K = 0;
    DevBuf<V> dinv;                   // block k at k bs bs, row-major m_k x m_k
    DevBuf<int32_t> lo_ptr, lo_idx, up_ptr, up_idx;   // every row's entries left of / right of its own block (original columns)
    DevBuf<V> lo_val, up_val;
    DevBuf<V> z;
    void build_chain(const HostCsr &A, const int32_t *d_ptr, const int32_t *d_idx, const double *d_val, hipStream_t s);
    void build(const HostCsr &A, hipStream_t s);
    bool build_sine(const HostCsr &A, hipStream_t s);       // P = 0 when the operator qualifies
    void solve(const V *b, V *x, hipStream_t s) const;     // original numbering, device pointers
};

// ---- lexicographic Gauss-Seidel of grid stencils (march.hip) ---------------------------------
// The reference's own smoother (openmg/solvers.py:56-68) relaxes the rows in their natural order.
// The general path runs it as a level schedule: one launch per set of mutually uncoupled rows —
// 766 dependent launches per sweep of a 256^3 seven-point operator.  Where the operator is a
// star stencil on a lexicographically numbered grid — every stored entry couples row r with
// r, r -+ 1, r -+ nx or r -+ nx ny, which the plan reads off the CSR structure itself — the sweep
// runs as ONE launch instead: a wave owns a TJ x TK tile of grid LINES (64 lanes, a line = the nx
// consecutive rows of one (j, k)), lane (jj, kk) relaxes row i = t - jj - kk of its line at step t,
// so that the three already relaxed neighbours of a row were relaxed one step earlier by this lane,
// lane - 1 and lane - TJ (wave shuffles), and the three not yet relaxed ones are old values.
// Tiles hand their faces over through HBM: one slot per row of a tile's last lines, unset (a marker
// NaN) between sweeps, written write-through by the owning tile's storing wave and polled by the
// +J / +K tile, which starts a block of steps once every row it needs has arrived (tile indices
// are taken from a ticket counter, so a tile's predecessors have always started).  Every row is the
// same fma chain in stored order as in the row kernels (absent neighbours contribute 0 * x to the
// chain, which leaves it unchanged), and the same division.
constexpr int MARCH_FACE_PAD = 72;    // slots in front of / behind the rows of a face line (>= largest skew 63 + steps per block)
struct MarchGeom {
    int nx = 0, ny = 0, nz = 0;       // rows per line, lines per plane, planes
    int TJ = 0, TK = 0;               // tile of lines held by one wave (TJ * TK = 64)
    int ntj = 0, ntk = 0, n_tiles = 0;
    int T = 0, n_grp = 0;             // steps per tile = nx + TJ + TK - 2; groups of four steps (even count)
    int n_pat = 0;                    // distinct rows (7 coefficients)
};
template <typename V>
struct MarchPlan {
    MarchGeom g;
    DevBuf<uint32_t> codes;           // [tile][group][lane]: the pattern codes of the lane's rows of four consecutive steps
    DevBuf<V> coef;                   // [pattern][8]: -K, -J, -I, diagonal, +I, +J, +K, unused
    DevBuf<uint32_t> sync;            // ticket, finished tiles, error flag
    DevBuf<int32_t> order;            // ticket -> tile: along the wavefront (anti-diagonals of the tile grid), so that the
                                      // tiles holding a CU are the ones next to run (empty: tickets are tile numbers)
    DevBuf<V> faceJ, faceK;           // [tile][line of the +J / +K face][row]: hand-over slots, unset (a marker NaN) between sweeps
    // A 1-D grid (tridiagonal operator, any coefficients: openmg's own demo and test operators, BASELINE configs[0]): the
    // sweep is a first-order recurrence x_i = f_i(x_{i-1}); ONE wave walks it (line_gs_kernel), the rows' coefficients as
    // three arrays (a row without a neighbour: a zero coefficient)
    bool line1 = false;
    DevBuf<V> tri;                    // [3][n]: -I, diagonal, +I
    // More than 256 distinct rows (per-row coefficients: the ordinary variable-coefficient input): no pattern table — every
    // row's seven coefficients as the tiles consume them, [tile][step][8][lane], streamed by a third wave of the workgroup
    // into an LDS ring one block ahead of the computing wave (round 6)
    bool per_row = false;
    DevBuf<V> rowc;
    // OMG_MARCH_SCAN=1 (opt-in, 3-D pattern-table levels, nx <= 512): a wave resolves a whole grid line by a scan over the
    // line's first-order recurrence — ny + nz - 1 line steps per sweep; NOT the bits of the sequential loop (march.hip)
    bool line_scan = false;
    int scan_c = 0, scan_g = 0;       // rows per lane (64 scan_c >= nx); workgroups (SCAN_W planes each)
    DevBuf<uint8_t> rowcode;          // every row's pattern, natural order
    DevBuf<V> face_scan;              // [workgroup][line][64 scan_c]: the last plane's relaxed lines for the next workgroup
    // false: the operator is not such a stencil (the caller keeps the level schedule)
    bool build(const omg_csr &A, hipStream_t s);
    void sweep(V *x, const V *b, hipStream_t s) const;   // one in-place lexicographic sweep
    bool timed_out(hipStream_t s) const;                 // (synchronises) some sweep gave up waiting for a face
};

// ---- setup that stays on the device (round 4: omg_hierarchy_create_from_fine) -----------------------------------
// A plain device CSR (double values, natural numbering): what the Galerkin chain leaves per level.
struct DevCsrPlain {
    int64_t n_rows = 0, n_cols = 0, nnz = 0;
    DevBuf<int32_t> indptr, indices;
    DevBuf<double> data;
};
// operators.restrictionList + coeffecientList (openmg/operators.py:92-141, 144-188) without the host in between: A[0] =
// the caller's operator uploaded once, R[l] built on the device (omg_restriction's kernel), A[l + 1] = (R[l] A[l]) R[l]^T
// by omg_rap's kernels, everything left in HBM.  n_restrictions operators R are made (the caller has applied the
// reference's depth rule).  (setup_device.hip)
void galerkin_chain_device(const omg_csr &A0, int dim, const int64_t *shape, int n_restrictions, std::vector<DevCsrPlain> &A,
                           std::vector<DevCsrPlain> &R, hipStream_t s);
HostCsr download_csr(const DevCsrPlain &M, hipStream_t s);
// C = (R A) R^T for an aggregation R (every column of R owned by exactly one row) by the fused kernel, SciPy's
// accumulation order, ascending columns; false: the operands do not qualify (setup_device.hip)
bool rap_aggregation_device(const DevCsrPlain &R, const DevCsrPlain &A, DevCsrPlain &C, hipStream_t s);

// Is R the plain 2 x 2 x 2 aggregation of an nx x ny x nz grid (openmg/operators.py:73-84: eight entries per coarse
// cell in (dk, dj, di) order, ONE weight, ascending columns)?  w: that weight.  Host scan on many threads.
bool is_plain_aggregation(const omg_csr &R, int64_t nx, int64_t ny, int64_t nz, double &w);   // (stencil27.hip)

// ---- plane-pipelined red-black passes of grid stencils (plane.hip) -----------------------------
// A V(1,1) cycle over a red-black ordered level streams x and b through HBM eight times (red
// sweep, black sweep + residual, red residual, restriction; prolongation, red, black + norm, red
// norm).  Where the level is a constant-coefficient star stencil on a lexicographically numbered
// grid (boundary rows drop the entries whose neighbour does not exist; read off the CSR itself) and
// the restriction is the 2x2x2 aggregation of openmg/operators.py:73-84, the two halves of the
// cycle run as ONE launch each:
//   down:  x_new = RB-sweep(x_old),  b_c = R (b - A x_new)  (+ the coarse level's first relaxation)
//   up:    x_new = RB-sweep(x_old + R^T e),  block partials of ||b - A x_new||^2
// A workgroup owns an (x, y) tile with an overlapped ring and marches in z over a chunk of planes:
// red on plane s, black on plane s-1, the residual of plane s-2 and the restriction of finished
// plane pairs, out of place (x_old is only read, x_new only written), every value a row needs from
// its own column in registers, from its in-plane neighbours through LDS.  A row is the same fma
// chain in stored (= slot) order as in the row kernels — an absent neighbour is 0 * c, which leaves
// the chain as it is — and the same division, so the iterate has the bits of the set-by-set
// schedule (tests/test_gpu_plane.py; OMG_PLANE=0 switches the path off).
constexpr int PLANE_EDGE = 4;         // planes at either end of a slab that the EDGE launch of a split pass covers
struct PlaneGeom {
    int nx = 0, ny = 0, nz = 0;       // cells per line, lines per plane, planes (all even)
    int hx = 0;                       // cells of one colour per line = nx / 2
    // a slab with ghost planes (plane_dist.hip): planes [z_base, z_end) are relaxed and stored, [kv0, kv1) exist in
    // the global grid; the coarse grid has nzc planes and fine plane k restricts into coarse plane (k >> 1) + kc_off
    int z_base = 0, z_end = 0, kv0 = 0, kv1 = 0, kc_off = 0, nzc = 0;
    int TX = 0, TY = 0, LZ = 0;       // a workgroup's interior: cells in x (multiple of 4), lines (even), planes (even)
    int PX = 0, PY = 0;               // its threads: TX / 4 + 2 (four cells each) x TY / 2 + 4 (two lines each)
    int ntx = 0, nty = 0, ntz = 0, n_wg = 0;
    int threads = 0;                  // PX * PY rounded up to whole waves
    size_t lds_bytes = 0;
    double c[7] = {0, 0, 0, 0, 0, 0, 0};   // -K, -J, -I, diagonal, +I, +J, +K
    double w = 0.0;                   // the restriction's weight
    // small levels (set when the plan is built, from OMG_PLANE_BLOCK / OMG_PLANE_BLOCK_CELLS / OMG_PLANE_LA2):
    bool dim2 = false;                // nz == 1: a 2-D grid (five-point stencil, 2 x 2 aggregation): tile2d_kernel
    bool jacobi = false;              // ... smoothed with weighted Jacobi (weight omega) in its natural ordering instead of red-black
    double omega = 1.0;
    bool block = false;               // whole grid of <= 64^3 cells: block_kernel where the pass allows it
    bool la2 = true;                  // marching kernel with two steps of lookahead for workgroups of <= 128 threads
};
template <typename V>
struct PlanePlan {
    PlaneGeom g;
    DevBuf<double> partials;          // one per workgroup (+ SUM_FOLD): the up pass's share of ||b - A x||^2
    // device word (not owned; null: none) in which a pass records that it gave up a bounded wait: bit 0 a neighbour
    // GPU's flag (Peer::status overrides it), bit 1 a neighbouring wave of its own workgroup.  Whoever reads results
    // back looks at it (hierarchy.hip check_march, omg_pdist_p2p_status)
    uint32_t *status = nullptr;
    // false: the level does not qualify (the caller keeps the set-by-set schedule; ord untouched).  A, R:
    // the caller's CSR in natural numbering.  true: ord = the level's colour ordering, written in closed
    // form (parity colours, red first — what the greedy colouring of such a stencil gives, without its
    // sequential pass over the rows).
    // jacobi: the level is smoothed with weighted Jacobi (2-D levels only): natural ordering, one set.
    bool build(const omg_csr &A, const omg_csr &R, Ordering &ord, bool jacobi = false, double omega = 1.0);
    // The same qualification for an operator that is already in HBM, on a grid the caller knows (nz = 1: 2-D), restricted by
    // the plain aggregation with weight w: the rows are checked by a kernel.  ord gets its sets only (perm / inv empty:
    // Ordering::parity_pending; the device copies of perm and of the coarse slot map are made by fill_parity_maps).
    bool build_device(const DevCsrPlain &A, int nx, int ny, int nz, double w, Ordering &ord, bool jacobi, double omega, hipStream_t s);
    bool finish_geometry(int64_t nx, int64_t ny, int64_t nz, const double (&c)[7], double w, bool jacobi, double omega);
    // y = A x of the whole level (colour-ordered vectors), matrix-free; the bits of the row kernels' SpMV
    void spmv(const V *x, V *y, hipStream_t s) const;
    // The operator and the restriction the plan stands for, as the caller's CSR had them (natural numbering,
    // ascending columns; doubles — of the rounded coefficients for a float plan): what the row kernels'
    // format of a plane level is built from when something asks for it (hierarchy.hip ensure_format).
    HostCsr operator_csr() const;
    HostCsr restriction_csr() const;
    // A slab of such a level with `ghost` planes on either side (plane_dist.hip): nz_own planes of nx x ny cells
    // behind `ghost` ghost planes; first / last: the slab holds the global grid's first / last plane (its outer
    // ghost planes do not exist and stay zero).  The coarse slab has `ghost_c` ghost planes of its own.
    void build_slab(int nx, int ny, int nz_own, int ghost, int ghost_c, bool first, bool last, const double (&c)[7], double w);
    // Slab neighbours reached by peer stores over xGMI (dist.hip, p2p mode): a pass writes its first / last planes of
    // x_new (and, going down, of the coarse right-hand side) straight into the neighbours' ghost planes, raises a
    // flag in each neighbour's memory when ALL its workgroups have done so, and — before it reads a ghost plane —
    // waits for the flags its neighbours' producing passes raise in its own memory.  Index 0: rank - 1, 1: rank + 1.
    struct Peer {
        V *x[2] = {nullptr, nullptr};             // the neighbour's vector in the role of this pass's x_new (null: no neighbour)
        V *bc[2] = {nullptr, nullptr};            // down: its coarse right-hand side (null: none, or not exchanged)
        int64_t shift = 0, cshift = 0;            // slots per slab: my slot + shift = the same cell in rank - 1's slab, - shift: rank + 1's
        int planes = 0, cplanes = 0;              // boundary planes written there
        int zc_lo = 0, zc_hi = 0;                 // my owned coarse planes [zc_lo, zc_hi) of the extended coarse slab
        const uint32_t *wait_flag[4] = {nullptr, nullptr, nullptr, nullptr};   // in MY memory (null: nothing to wait for)
        uint32_t wait_seq[4] = {0, 0, 0, 0};      // ... until it holds at least this (wrapping compare)
        bool fused_wait = true;                   // the pass waits itself (false: the caller has launched wait_kernel)
        uint32_t *done = nullptr;                 // my workgroup counter (zero between launches)
        uint32_t *flag[2] = {nullptr, nullptr};   // in the NEIGHBOURS' memory
        uint32_t seq = 0;                         // what is stored there
        uint32_t *status = nullptr;               // my error word: bit 0 = a wait gave up
        uint32_t spin = 1u << 21;                 // polls before a wait gives up
    };
    struct Coarse {
        const int32_t *map = nullptr; // coarse natural index -> slot in the coarse ordering (null: identity)
        V *b = nullptr;               // down: coarse right-hand side
        V *x = nullptr;               // down: coarse initial iterate (null: not written)
        const V *diag = nullptr;      // down: first relaxation of the coarse zero iterate for slots < first_end (null: zeros)
        int first_end = 0;
        const V *e = nullptr;         // up: coarse correction
    };
    // x_zero: x_old is known to be zero and is not read
    // Large levels: times the candidate tilings on these vectors (contents destroyed) and keeps the fastest
    // (OMG_PLANE_TUNE=0 / OMG_PLANE_TILE: no timing).
    void tune(V *x, V *tmp, const V *b, const Coarse &c, hipStream_t s, bool finest = true);
    // microseconds per down + up pass with the chosen tiling on these vectors (one untimed pair, then `pairs` timed ones)
    float time_pair(V *x, V *tmp, const V *b, const Coarse &c, hipStream_t s, bool finest, int pairs);
    // sweep = false: the pass without its relaxation (a cycle with preIterations = 0 / postIterations = 0): down then
    // leaves x_new untouched (the iterate stays in x_old), up writes x_new = x_old + R^T e
    // part: a pass as TWO launches (slabs with RCCL exchanges, dist.hip): PART_EDGE = the slab's first and last PLANE_EDGE
    // planes — what the neighbours are waiting for — as two short chunks, PART_INNER = the planes between them; the two
    // write disjoint planes and may run side by side on two streams.
    enum { PART_ALL = 0, PART_EDGE = 1, PART_INNER = 2 };
    // (PLANE_EDGE = 4 planes at either end: the three planes of x and the two coarse planes a neighbour takes)
    bool can_split() const {
        return !g.dim2 && !g.block && !(g.la2 && g.threads <= 128) && (g.z_end - g.z_base) >= 3 * PLANE_EDGE;
    }
    // workgroups = norm partials of a pass made of the two launches
    int split_partials() const {
        const int nzo = g.z_end - g.z_base;
        return nzo >= 3 * PLANE_EDGE ? g.ntx * g.nty * (2 + (nzo - 2 * PLANE_EDGE + g.LZ - 1) / g.LZ) : g.n_wg;
    }
    // Slab neighbours reached by an exchange that runs on ANOTHER stream while the pass does (dist.hip, RCCL): the pass is
    // ONE launch of inner chunks and — with the highest workgroup numbers, dispatched last — the slab's first and last
    // PLANE_EDGE planes as short chunks of their own, which wait (bounded, status bit 0) until flag[i] holds seq[i] before
    // they read a ghost plane; the inner chunks read none and never wait.
    struct Gate {
        const uint32_t *flag[4] = {nullptr, nullptr, nullptr, nullptr};
        uint32_t seq[4] = {0, 0, 0, 0};
        uint32_t *status = nullptr;
        uint32_t spin = 1u << 21;
    };
    // Does such a launch pay, and can it run at all?  Its edge workgroups spin on their compute units until the exchange
    // has landed, and the exchange's own kernels (RCCL's, a copy's) need compute units too: with 2 x (xy tiles) >= 256
    // edge workgroups every unit ends up holding a spinning one and nothing can raise the flag (measured: the 512 x 512 x 64
    // slabs of the 8-rank shape, 128 xy tiles, ran into the bounded wait) — at most 192 of them; and a slab thinner than
    // 96 planes spends more on its eight edge planes as chunks of their own (+ 25 % workgroup-steps at 64 planes) than the
    // exchange it hides.
    bool can_gate() const { return can_split() && 2 * g.ntx * g.nty <= 192 && (g.z_end - g.z_base) >= 96; }
    int gate_lz() const;              // planes per inner chunk of such a launch
    int gate_partials() const;        // its workgroups = norm partials
    void down(const V *x_old, V *x_new, const V *b, bool x_zero, const Coarse &c, hipStream_t s, const Peer *peer = nullptr,
              bool sweep = true, int part = PART_ALL, const Gate *gate = nullptr) const;
    // out (nullable): also the squares of the residual of the NEW iterate, one partial per workgroup
    void up(const V *x_old, V *x_new, const V *b, const Coarse &c, double *out, hipStream_t s, const Peer *peer = nullptr,
            bool sweep = true, int part = PART_ALL, const Gate *gate = nullptr) const;
};

// ---- 7-point grid stencils with PER-ROW coefficients (var7.hip) ---------------------------------------------------
// The ordinary real input of mgSolve (openmg/__init__.py:28: any A_in; operators.py:178): -div(kappa grad u) on a grid.
// A level whose operator holds, in every row, exactly the in-grid neighbours of a 7-point stencil on a lexicographically
// numbered grid of even extents (columns ascending: -K, -J, -I, diagonal, +I, +J, +K) with ARBITRARY coefficients,
// restricted by the plain 2 x 2 x 2 aggregation, smoothed red-black (parity colours, what the greedy colouring finds).
// The set-by-set schedule walks the operator 3.5 times per V(1,1) cycle in eight level launches; here each half of the
// cycle is ONE launch, as for the constant-coefficient levels (PlanePlan): down = last pre-smoothing sweep + residual +
// restriction (openmg/__init__.py:201, :209, :210), up = prolongation + correction + first post-smoothing sweep + the
// norm's squares (:214, :220/:224, :216-222, :227).  A workgroup owns an (x, y) tile with a ring relaxed redundantly
// (red two cells, black one) and marches along z; the iterate's planes live in LDS, the coefficients — seven arrays in the
// vectors' own colour layout; or, where the operator is symmetric bit for bit, the diagonal and the three "+" couplings,
// each read by both rows it couples — are streamed from HBM with unit stride.  Same chains, same division, same bits as
// the row kernels on the same operator (tests/test_gpu_var7.py against omg_hierarchy_use_plane(0)).
template <typename V>
struct Var7Plan {
    int nx = 0, ny = 0, nz = 0;
    double w = 0.0;                   // the aggregation's one weight
    bool sym = false;                 // a(i, j) == a(j, i) bit for bit: the "-" arrays are not kept
    int tx = 0, ty = 0, lz = 0, ntx = 0, nty = 0, ntz = 0, threads = 0;
    int64_t n_wg = 0;
    DevBuf<V> cD, cM[3], cP[3];       // [colour-ordered slot]; M / P index: 0 = I, 1 = J, 2 = K
    DevBuf<double> partials;          // one per workgroup: the up pass's share of ||b - A x||^2
    // false: the level does not qualify (nothing changed).  A, R: the caller's CSR in natural numbering.  true: ord = the
    // level's colour ordering in closed form (parity colours, red first — what the greedy colouring of such a stencil gives,
    // without its sequential pass over the rows; OMG_PLANE_CHECK_ORDER=1 compares the two).
    bool build(const omg_csr &A, const omg_csr &R, Ordering &ord, hipStream_t s);
    // The same for an operator that is already in HBM (mgSolve's setup on the device, omg_hierarchy_create_from_fine), on a
    // grid the caller knows, restricted by the plain aggregation with weight w: one kernel checks the rows and scatters their
    // coefficients into the seven arrays.  ord gets its sets and closed form only (perm / inv stay on the device).
    bool build_device(const DevCsrPlain &A, int nx, int ny, int nz, double w, Ordering &ord, hipStream_t s);
    // The operator and the restriction as the caller's CSR had them (natural numbering, ascending columns; the values as the
    // level holds them): what the row kernels' format of the level is built from when something asks for it
    // (hierarchy.hip ensure_format) — a cycle over the fused passes never does.
    HostCsr operator_csr(hipStream_t s) const;
    HostCsr restriction_csr() const;
    struct Coarse {
        const int32_t *map = nullptr; // coarse natural index -> slot in the coarse ordering (null: identity)
        V *b = nullptr;               // down: coarse right-hand side
        const V *e = nullptr;         // up: coarse correction
    };
    // sweep = false: the pass without its relaxation (down leaves x_new untouched: the iterate stays in x_old)
    void down(const V *x_old, V *x_new, const V *b, bool x_zero, const Coarse &c, hipStream_t s, bool sweep = true) const;
    void up(const V *x_old, V *x_new, const V *b, const Coarse &c, double *out, hipStream_t s, bool sweep = true) const;
};

// ---- 27-point grid stencils with per-row coefficients: BASELINE configs[4] (stencil27.hip) ----------------
// A level whose operator holds, in every row, exactly the in-grid neighbours of a 27-point stencil on a
// lexicographically numbered grid of even extents (columns ascending: slot s = 9 (dz + 1) + 3 (dy + 1) + dx + 1) with
// ARBITRARY coefficients, restricted by the plain 2 x 2 x 2 aggregation, smoothed with the 8-colour Gauss-Seidel the
// greedy colouring finds on it (colour = (i & 1) + 2 (j & 1) + 4 (k & 1): the octant of a cell inside its aggregate).
// In that ordering each colour's part of a level vector is indexed by the AGGREGATE (I, J, K) = (i, j, k) >> 1, so
//   * a row's 27 operands are unit-stride runs of eight "coarse-shaped" arrays at shifts of -1 / 0 / +1 aggregates,
//   * the restriction (openmg/__init__.py:210) and the prolongation (:214) are element-wise over the eight arrays,
//   * colours 2m and 2m + 1 (x even / x odd of one line) depend on each other only inside their own grid line,
// and the set-by-set schedule (8 launches per sweep, a 4-byte gather + a 4-byte coefficient load per stored entry)
// becomes 4 launches per sweep that stream the coefficients — held per colour and wave in [slot][lane] tiles of
// 16-byte accesses — once each:
//   sweep_pair    colours (2m, 2m + 1), out of place (x_old only read); the last pair also leaves the residuals of
//                 its rows (their sweep is final), the first sweep of a batched cycle also squares the residuals
//                 with respect to x_old — the PREVIOUS cycle's norm (openmg/__init__.py:227) for free;
//   residual      b - A x of the other six colours fused with the restriction (r never exists in HBM), or squared
//                 for the norm;
//   prolong       x += R^T e.
// Every row is summed as the row kernels sum a row of 27 stored entries — four fma chains, entry e to chain e & 3,
// ((s0 + s1) + s2) + s3 — with an absent neighbour as an explicit zero coefficient: the level's row-kernel format is
// built from that PADDED operator (27 entries in every row), so the set-by-set schedule of the same hierarchy
// (omg_hierarchy_use_plane(0)) produces the same bits.
struct S27Geom {
    int nx = 0, ny = 0, nz = 0;       // cells (even)
    int hx = 0, hy = 0, hz = 0;       // aggregates per line / lines per plane / planes
    int64_t na = 0;                   // aggregates = rows per colour
    int rg = 0;                       // aggregates per lane (16-byte accesses: 4 floats, 2 doubles; fewer on small levels)
    int L = 0, G = 0;                 // lanes per grid line, lines per wave
    int64_t nl = 0, ng = 0;           // lines per colour; waves (line groups) per colour
    int wpb = 4;                      // waves per workgroup (1 on small levels)
    int n_wg = 0;                     // workgroups
    double w = 0.0;                   // the restriction's weight
};
template <typename V>
struct Stencil27Plan {
    S27Geom g;
    DevBuf<V> coef;                   // [colour][wave group][slot][lane][rg]
    DevBuf<V> res67;                  // residuals of colours 6 and 7, left by a sweep's last pair launch
    DevBuf<double> partials;          // 4 segments of n_wg block partials (+ SUM_FOLD): squared residuals
    bool have67 = false;              // res67 / segment 3 belong to the current iterate
    // The grid lines (line = aggregate plane * hy + aggregate line) the launches work on: all of them on a whole grid.
    // One rank's slab with a ghost aggregate plane on either side (dist27.hip): the owned planes' lines [live_lo,
    // live_hi) — residuals, squares, the pair launches of colours 4 .. 7 —, for the pair launches of colours 0 .. 3
    // [live_lo01, live_hi01): also the upper ghost plane's (relaxed redundantly); the prolongation takes every line.
    int64_t live_lo = 0, live_hi = 0, live_lo01 = 0, live_hi01 = 0;
    // false: the level does not qualify (ord untouched).  A, R: the caller's CSR in natural numbering.
    bool build(const omg_csr &A, const omg_csr &R, Ordering &ord, hipStream_t s);
    // ... for an operator already in HBM on a known grid, restricted by the plain aggregation with weight w
    bool build_device(const DevCsrPlain &A, int nx, int ny, int nz, double w, Ordering &ord, hipStream_t s);
    // ... for one rank's slab (stencil27.hip; throws when the rows do not qualify)
    void build_slab(const DevCsrPlain &A, int nx, int ny, int nz_ext, bool first, bool last, double w, hipStream_t s);
    bool tile(const DevCsrPlain &A, const S27Geom &q, int kz_lo, int kz_hi, int kb_lo, int kb_hi, hipStream_t s);
    // the coefficient rows of colours 0 .. 3 of aggregate plane K as one contiguous array (plane_rows_count() values):
    // a slab's upper ghost plane takes them from the neighbour's first owned plane
    size_t plane_rows_count() const;
    void pack_plane_rows(int K, V *buf, hipStream_t s) const;
    void unpack_plane_rows(int K, const V *buf, hipStream_t s);
    // One sweep x_old -> x_new.  x_zero: x_old is zero and is not read.  norm_old (nullable: 4 n_wg doubles): also
    // the squares of b - A x_old.  last: leave the residuals of colours 6, 7 (res67) and, with norm_new (nullable,
    // same shape; only segment 3 is written), their squares.
    void sweep(const V *x_old, V *x_new, const V *b, bool x_zero, double *norm_old, bool last, double *norm_new, hipStream_t s);
    // a large level's tiles placed by timing (see the definition); x, tmp, b: the level's vectors, overwritten
    void place_tiles(V *x, V *tmp, V *b, hipStream_t s);
    // coarse right-hand side = R (b - A x) (cmap: coarse natural index -> slot in the coarse ordering, null: identity).
    // use67: the residuals of colours 6, 7 are taken from res67.
    void residual_restrict(const V *x, const V *b, bool use67, const int32_t *cmap, V *bc, hipStream_t s);
    // squares of b - A x into segment 0 of `out` (4 n_wg doubles; segments 1, 2 are cleared, segment 3 too unless
    // use67: then it holds the squares of colours 6, 7 already)
    void norm(const V *x, const V *b, bool use67, double *out, hipStream_t s);
    void prolong(V *x, const V *e, const int32_t *cmap, hipStream_t s);
    // The operator the row kernels' format of this level is built from: every row with all 27 slots, an absent
    // neighbour as a zero entry on the row's own column (doubles of the stored V); and the aggregation.
    HostCsr operator_csr(hipStream_t s) const;
    HostCsr restriction_csr() const;
    // New coefficients, and the Galerkin product of this level in closed form (stencil27.hip s27_rap_kernel: one pass over
    // the operator, SciPy's accumulation order).  The operator comes as the caller's CSR values behind `indptr` (the
    // level's own pattern: every row its in-grid neighbours in column order) or, indptr == null, as a dense [row][27]
    // double array (an absent neighbour a zero).  write_fine: its entries go into this level's tiles (as V);
    // coarse_dense (nullable): the coarse operator as such an array; coarse (nullable): ... and into that level's tiles.
    // Throws when a value is not finite as V or a diagonal is zero.
    void rap_from(const int32_t *indptr, const double *vals, bool write_fine, double *coarse_dense, Stencil27Plan<V> *coarse, hipStream_t s);
};

}  // namespace omg
