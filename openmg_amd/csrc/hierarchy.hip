// Device-resident level hierarchy and the V-cycle driver (openmg/__init__.py:151-236),
// plus the extern "C" surface declared in include/openmg_hip.h.
//
// The implementation is a template on the value type V the levels are stored and computed
// in (double: the reference's precision; float: BASELINE configs[4]).  The C handle holds
// one of the two instantiations; the host boundary is double either way.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstring>
#include <memory>
#include <array>
#include <map>
#include <mutex>
#include <thread>
#include <type_traits>

#include "common.h"

namespace omg {

// ---- scattered device memory (common.h) -------------------------------------------------------------------------------
std::vector<ScatteredBlock> &scattered_registry() {
    static std::vector<ScatteredBlock> r;
    return r;
}
static std::mutex &scattered_mutex() {
    static std::mutex m;
    return m;
}
// Tear a block down the way it was put up: ONE hipMemUnmap per hipMemMap (the runtime unmaps per mapping — a single call over
// a range of many mappings need not undo any of them, and releasing the handles would then leave the pages mapped: a leak
// of the whole block for the life of the process), then the handles, then the address range.  Every failure is reported.
static void scattered_release(ScatteredBlock &blk, size_t mapped) {
    auto report = [](const char *what, hipError_t e) {
        if (e == hipSuccess) return;
        (void)hipGetLastError();
        fprintf(stderr, "libopenmg_hip: scattered block: %s failed: %s (device memory may stay allocated)\n", what, hipGetErrorString(e));
    };
    for (size_t i = 0; i < mapped; ++i) report("hipMemUnmap", hipMemUnmap(static_cast<char *>(blk.va) + i * blk.chunk, blk.chunk));
    for (auto h : blk.handles) report("hipMemRelease", hipMemRelease(h));
    blk.handles.clear();
    if (blk.va) report("hipMemAddressFree", hipMemAddressFree(blk.va, blk.total));
    blk.va = nullptr;
}
void *scattered_alloc(size_t bytes, size_t chunk_bytes) {
    int device = 0;
    if (hipGetDevice(&device) != hipSuccess) return nullptr;
    hipMemAllocationProp prop;
    std::memset(&prop, 0, sizeof(prop));
    prop.type = hipMemAllocationTypePinned;
    prop.location.type = hipMemLocationTypeDevice;
    prop.location.id = device;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess || gran == 0) { (void)hipGetLastError(); return nullptr; }
    const size_t chunk = (std::max(chunk_bytes, gran) + gran - 1) / gran * gran;
    const size_t n = (bytes + chunk - 1) / chunk;
    ScatteredBlock blk;
    blk.total = n * chunk;
    blk.chunk = chunk;
    auto undo = [&](size_t mapped) { scattered_release(blk, mapped); };
    if (hipMemAddressReserve(&blk.va, blk.total, chunk, nullptr, 0) != hipSuccess) { (void)hipGetLastError(); return nullptr; }
    blk.handles.reserve(n);
    for (size_t i = 0; i < n; ++i) {
        hipMemGenericAllocationHandle_t h;
        if (hipMemCreate(&h, chunk, &prop, 0) != hipSuccess) { undo(0); return nullptr; }
        blk.handles.push_back(h);
    }
    // piece i of the virtual range <- the perm(i)-th allocation: a fixed pseudo-random permutation (Fisher-Yates over an LCG)
    std::vector<size_t> perm(n);
    for (size_t i = 0; i < n; ++i) perm[i] = i;
    uint64_t state = 0x9E3779B97F4A7C15ull;
    for (size_t i = n; i > 1; --i) {
        state = state * 6364136223846793005ull + 1442695040888963407ull;
        std::swap(perm[i - 1], perm[size_t((state >> 33) % i)]);
    }
    for (size_t i = 0; i < n; ++i)
        if (hipMemMap(static_cast<char *>(blk.va) + i * chunk, chunk, 0, blk.handles[perm[i]], 0) != hipSuccess) { undo(i); return nullptr; }
    hipMemAccessDesc acc;
    std::memset(&acc, 0, sizeof(acc));
    acc.location.type = hipMemLocationTypeDevice;
    acc.location.id = device;
    acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(blk.va, blk.total, &acc, 1) != hipSuccess) { undo(n); return nullptr; }
    void *const va = blk.va;
    std::lock_guard<std::mutex> lock(scattered_mutex());
    scattered_registry().push_back(std::move(blk));
    return va;
}
bool scattered_free(void *va) {
    ScatteredBlock blk;
    {
        std::lock_guard<std::mutex> lock(scattered_mutex());
        auto &r = scattered_registry();
        size_t i = 0;
        while (i < r.size() && r[i].va != va) ++i;
        if (i == r.size()) return false;
        blk = std::move(r[i]);
        r.erase(r.begin() + long(i));
    }
    // unlike hipFree, unmapping does not wait for work that still uses the range
    if (hipDeviceSynchronize() != hipSuccess) (void)hipGetLastError();
    scattered_release(blk, blk.handles.size());
    return true;
}

static thread_local std::string g_last_error;
void set_last_error(const std::string &m) { g_last_error = m; }

double SetupTimer::now() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}

void require_device() {
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess || n <= 0)
        throw Error(OMG_ERR_NO_DEVICE,
                    "no HIP device visible: libopenmg_hip.so has no CPU fallback "
                    "(hipGetDeviceCount: " + std::string(hipGetErrorString(e)) + ")");
}

namespace {

inline bool getenv_flag(const char *name) { const char *e = getenv(name); return e && e[0] == '1'; }
inline bool getenv_flag0(const char *name) { const char *e = getenv(name); return e && e[0] == '0'; }   // switched OFF

struct SweepStep {
    int set_begin, set_end;   // [begin, end)
    bool serial;              // run of single-block sets -> one workgroup
};

template <typename V>
struct Level {
    int64_t n = 0;
    DevCsrT<V> A;             // P A P^T in this level's smoother ordering
    DevCsrT<V> R;             // to level+1: rows in next ordering, columns in this ordering
    DevCsrT<V> P;             // R^T: rows in this ordering, columns in next ordering
    Ordering ord;
    DevBuf<int32_t> perm;     // new -> old (empty when identity)
    DevBuf<int32_t> r_out;    // restriction row (natural coarse numbering) -> slot in the next level's ordering
    DevBuf<V> x, b, r, tmp;
    DevBuf<V> diag;           // diagonal of A in this level's ordering (levels entered with a zero iterate: restrict_level)
    DevBuf<double> partials;
    DevBuf<double> nat;       // natural-order (double) staging for host I/O at this level
    V *xp = nullptr, *tp = nullptr;   // current iterate / Jacobi scratch (swap)
    std::vector<SweepStep> plan;
    // prolongation as a scatter over R's row patterns (ROW_SCATTER) instead of a pass over the
    // explicit transpose P: possible when no two rows of R share a column (aggregation) and R
    // is row-pattern coded; reads 1 byte per COARSE row instead of ~9 per fine row
    bool scatter_prolong = false;
    // lexicographic Gauss-Seidel of a grid star stencil: one launch per sweep (common.h MarchPlan);
    // the level then keeps its natural ordering (one "set") for every other kernel
    std::unique_ptr<MarchPlan<V>> march;
    // red-black sweeps of a constant-coefficient grid star stencil under the 2x2x2 aggregation: each half
    // of a V(pre >= 1, post >= 1) cycle is one plane-pipelined launch (common.h PlanePlan), out of place
    // between x and tmp
    std::unique_ptr<PlanePlan<V>> plane;
    // 27-point grid stencil with per-row coefficients under the 2x2x2 aggregation, 8-colour Gauss-Seidel (BASELINE
    // configs[4]): the cycle over this level runs the octant-layout kernels of stencil27.hip (common.h Stencil27Plan)
    std::unique_ptr<Stencil27Plan<V>> s27;
    // 7-point grid stencil with per-row coefficients, red-black (the ordinary variable-coefficient input): each half of the
    // cycle over this level is one launch of var7.hip (common.h Var7Plan); the row-kernel format exists beside it
    std::unique_ptr<Var7Plan<V>> var7;
    DevBuf<char> pool;               // OMG_VEC_POOL=1: x, tmp, b of a large plane level as three views into ONE allocation
    size_t pool_span = 0, pool_off1 = 0, pool_off2 = 0;      // ... its layout (pool_views)
    // where the matrix-free SpMV of a plane level leaves its product: an allocation of its OWN, made on first use.  Written
    // into the level's scratch vector — which shares an allocation with x, two spans further on — the same launch takes
    // 57-62 us at 256^3 where it takes 45 into any separate allocation (profiles/r05_spmv_placement.txt: one process, the
    // destinations in turn): read and write streams a fixed physical distance apart meet on the same banks
    DevBuf<V> spmv_y;
    // A plane level's cycle never touches the row-kernel format of A and R (nor r, the block partials, the
    // sweep plan): they are built on first use — by a cycle with pre = 0 or post = 0, a single-level
    // operation, a format query — from the operator the plan describes (ensure_format; OMG_PLANE_LAZY=0: at creation)
    bool format_pending = false;
};

struct ProfEvent {
    int cls;
    hipEvent_t a, b;
};

template <typename V>
struct Hier {
    using value_type = V;
    std::vector<Level<V>> lv;
    CoarseSolver<V> coarse;   // direct solve of the coarsest operator (common.h)
    DevBuf<double> norm_dev;
    DevBuf<uint32_t> plane_status; // PlanePlan::status of every plane level: bit 1 = a wave gave up waiting for its neighbour wave
    // omg_vcycle_ex: cycle_body leaves a copy of level pre_level's iterate after its pre-smoothing here (-1: nowhere)
    DevBuf<V> pre_buf;
    int pre_level = -1;
    DevBuf<double> norms_dev;      // omg_resident_cycles: one norm per cycle of the batch
    DevBuf<double> batch_partials; // ... and the block partials of up to 64 deferred norms
    int smoother = OMG_SMOOTH_GS_LEX;
    double omega = 1.0;
    hipStream_t own = nullptr, stream = nullptr;
    // omg_hierarchy_update_fine (a hierarchy set up on the device whose smoothed levels all run the 27-point kernels): the
    // fine operator's row pointers (the values come with every update), its grid, and per level >= 1 the Galerkin
    // operator as a dense [row][27] double array — the next product's input
    DevBuf<int32_t> indptr0;
    int64_t nnz0 = 0;
    std::vector<DevBuf<double>> dense27;
    // resident state
    bool resident = false;
    // OMG_NO_FUSE=1: never fuse the last smoother set with the residual / norm (A/B switch;
    // results are bit-identical either way, tests/test_gpu_parity.py checks that)
    bool no_fuse = [] { const char *e = getenv("OMG_NO_FUSE"); return e && e[0] == '1'; }();
    // OMG_PLANE=0 (at creation): no plane-pipelined passes; set by omg_hierarchy_use_plane afterwards
    bool no_plane = false;
    // graph: a captured cycle bakes the levels' current / scratch vector pointers in, and a cycle may leave them
    // swapped (out-of-place sweeps: Jacobi, the 27-point pair launches, the plane passes) — so a graph is keyed by the
    // pointer state it was captured FROM and records the state it leaves; a cycle with an odd number of swaps gets a
    // second graph for the other phase (one graph replayed from the wrong phase recomputed cycle 1 forever)
    struct GraphSlot {
        hipGraphExec_t exec = nullptr;
        int pre = -1, post = -1;
        std::vector<V *> before, after;      // xp, tp of every level
    };
    bool want_graph = false;
    std::vector<GraphSlot> graphs;
    // profile
    unsigned profiling = 0;   // bit c set: time level-0 launches of class c
    std::vector<ProfEvent> events;
    int64_t prof_n[OMG_PROFILE_CLASSES] = {0};
    double prof_ms[OMG_PROFILE_CLASSES] = {0};
    std::vector<hipEvent_t> event_pool;

    Hier() = default;
    Hier(const Hier &) = delete;
    Hier &operator=(const Hier &) = delete;
    ~Hier() {
        for (auto &g : graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
        for (auto &e : events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
        for (auto &e : event_pool) (void)hipEventDestroy(e);
        if (own) (void)hipStreamDestroy(own);
    }
};

}  // namespace
}  // namespace omg

struct omg_hierarchy {
    std::unique_ptr<omg::Hier<double>> d;
    std::unique_ptr<omg::Hier<float>> f;
};

namespace omg {
namespace {

void check_diagonal(const omg_csr &A, int level) {
    const int64_t bad = first_row_without_diagonal(A);
    if (bad >= 0)
        throw Error(OMG_ERR_NO_DIAGONAL, "level " + std::to_string(level) + ": row " +
                                             std::to_string(bad) + " has no nonzero diagonal entry");
}

omg_csr view(const HostCsr &A) {
    return omg_csr{A.n_rows, A.n_cols, A.nnz, A.indptr.data(), A.indices.data(), A.data.data()};
}

// The smoother ordering of a level.  The reference's lexicographic sweep first tries the
// one-launch wavefront of march.hip (OMG_MARCH=0: always the level schedule; same bits, tested).
template <typename V>
void order_level(Level<V> &L, const omg_csr &A, int smoother, hipStream_t s) {
    L.march.reset();
    if (smoother == OMG_SMOOTH_GS_LEX && !getenv_flag0("OMG_MARCH")) {
        std::unique_ptr<MarchPlan<V>> plan(new MarchPlan<V>);
        if (plan->build(A, s)) {
            L.march = std::move(plan);
            L.ord = Ordering();
            L.ord.identity = true;
            L.ord.sets = {0, A.n_rows};
            return;
        }
    }
    L.ord = make_ordering(A, smoother);
}

template <typename V>
void build_plan(Level<V> &L) {
    L.plan.clear();
    const int ns = (int)L.A.n_sets();
    int s = 0;
    while (s < ns) {
        auto one_block = [&](int k) { return L.A.set_blk[k + 1] - L.A.set_blk[k] == 1; };
        if (one_block(s)) {
            int e = s + 1;
            while (e < ns && one_block(e)) ++e;
            L.plan.push_back({s, e, e - s >= 2});
            s = e;
        } else {
            L.plan.push_back({s, s + 1, false});
            ++s;
        }
    }
}

template <typename V>
struct Prof {
    Hier<V> *h;
    int idx = -1;
    Prof(Hier<V> *hh, int level, int cls) : h(hh) {
        if (level != 0 || !((h->profiling >> cls) & 1u)) return;
        ProfEvent e;
        e.cls = cls;
        auto get = [&]() {
            hipEvent_t ev;
            if (!h->event_pool.empty()) { ev = h->event_pool.back(); h->event_pool.pop_back(); }
            else OMG_HIP(hipEventCreate(&ev));
            return ev;
        };
        e.a = get();
        e.b = get();
        OMG_HIP(hipEventRecord(e.a, h->stream));
        h->events.push_back(e);
        idx = (int)h->events.size() - 1;
    }
    ~Prof() {
        if (idx >= 0) (void)hipEventRecord(h->events[idx].b, h->stream);
    }
};

template <typename V>
void ensure_format(Hier<V> *h, int l);            // (a plane level's row-kernel side is built on first use: below)

// ---- level operations (device vectors in the level's ordering) ---------------------------
// What the LAST set launch of a Gauss-Seidel smoothing call also produces for its own rows
// (RowMode ROW_GS_RES / ROW_GS_NORM): the residual pass that follows then skips that set.
enum Fuse { FUSE_NONE = 0, FUSE_RESIDUAL = 1, FUSE_NORM = 2 };

// Fusion needs a Gauss-Seidel ordering whose final step is an ordinary set launch.
template <typename V>
bool can_fuse(const Hier<V> *h, const Level<V> &L) {
    if (h->smoother == OMG_SMOOTH_JACOBI || L.plan.empty() || h->no_fuse || L.march) return false;
    return !L.plan.back().serial;
}

// Can the FIRST launch of a cycle entered at this level also finish the previous cycle's residual
// norm (RowMode ROW_GS_PRENORM / ROW_JACOBI_PRENORM)?  Jacobi: its one launch sees the iterate
// the previous cycle left, for every row.  Gauss-Seidel: only the first set does, so the other
// sets' share must already be there — two sets, the second one's squares left by the fused
// post-smoothing launch (ROW_GS_NORM).
template <typename V>
bool can_prenorm(const Hier<V> *h, const Level<V> &L, int pre, int post) {
    if (pre <= 0 || L.plan.empty() || L.march) return false;
    if (h->smoother == OMG_SMOOTH_JACOBI) return true;
    return L.A.n_sets() == 2 && post > 0 && can_fuse(h, L) && !L.plan.front().serial && L.plan.size() == 2;
}

// Returns true when the last set launch was fused (its rows' residual / norm is done).
// pre_slot (block-partials array, nullable): the first launch runs in the PRENORM mode and leaves
// the squares of ITS rows' residuals — the previous cycle's — there (caller checked can_prenorm;
// the other set's were put there by that cycle's fused post-smoothing launch, post_slot).
// post_slot (nullable = the level's own partials): where a FUSE_NORM launch puts its squares.
// first_done: the first launch of the first sweep has already been applied (restrict_level wrote
// its result for a zero iterate): skip it.
template <typename V>
bool smooth_level(Hier<V> *h, int l, int iterations, Fuse fuse = FUSE_NONE, double *pre_slot = nullptr,
                  bool first_done = false, double *post_slot = nullptr) {
    ensure_format(h, l);
    Level<V> &L = h->lv[l];
    bool fused = false;
    for (int it = 0; it < iterations; ++it) {
        if (h->smoother == OMG_SMOOTH_JACOBI) {
            if (it == 0 && first_done) continue;
            Prof<V> p(h, l, 0);
            RowArgsT<V> a;
            a.x = L.xp; a.b = L.b.p; a.y = L.tp; a.omega = h->omega;
            if (it == 0 && pre_slot) {
                a.partials = pre_slot;
                launch_rows(L.A, ROW_JACOBI_PRENORM, -1, a, h->stream);
            } else {
                launch_rows(L.A, ROW_JACOBI, -1, a, h->stream);
            }
            std::swap(L.xp, L.tp);
        } else if (L.march) {
            Prof<V> p(h, l, 0);
            L.march->sweep(L.xp, L.b.p, h->stream);
        } else {
            RowArgsT<V> a;
            a.x = L.xp; a.b = L.b.p; a.y = L.xp;
            const bool last_it = it + 1 == iterations;
            for (size_t k = 0; k < L.plan.size(); ++k) {
                const SweepStep &st = L.plan[k];
                if (it == 0 && k == 0 && first_done) continue;
                Prof<V> p(h, l, 0);
                if (st.serial) {
                    launch_gs_serial(L.A, st.set_begin, st.set_end, a, h->stream);
                } else if (last_it && k + 1 == L.plan.size() && fuse != FUSE_NONE && can_fuse(h, L)) {
                    RowArgsT<V> f = a;
                    f.zero = L.r.p;
                    f.partials = (fuse == FUSE_NORM && post_slot) ? post_slot : L.partials.p;
                    launch_rows(L.A, fuse == FUSE_RESIDUAL ? ROW_GS_RES : ROW_GS_NORM, st.set_begin, f, h->stream);
                    fused = true;
                } else if (it == 0 && k == 0 && pre_slot) {
                    RowArgsT<V> f = a;
                    f.partials = pre_slot;                // the first set's blocks; the last set's are in place (ROW_GS_NORM)
                    launch_rows(L.A, ROW_GS_PRENORM, st.set_begin, f, h->stream);
                } else {
                    launch_rows(L.A, ROW_GS, st.set_begin, a, h->stream);
                }
            }
        }
    }
    return fused;
}

// r = b - A x over all sets, or over all but the last one when the smoother has already
// produced the last set's residual.
template <typename V>
void residual_level(Hier<V> *h, int l, V *r_out, bool last_set_done = false) {
    ensure_format(h, l);
    Level<V> &L = h->lv[l];
    Prof<V> p(h, l, 1);
    RowArgsT<V> a;
    a.x = L.xp; a.b = L.b.p; a.y = r_out;
    const int ns = (int)L.A.n_sets();
    launch_rows_range(L.A, ROW_RESIDUAL, 0, last_set_done ? ns - 1 : ns, a, h->stream);
}

// ||b - A x||_2 of level l into h->norm_dev (device scalar); r_out optional.  With
// last_set_done the last set's block partials are already in place.
template <typename V>
void norm_level(Hier<V> *h, int l, V *r_out, bool last_set_done = false, double *out = nullptr) {
    ensure_format(h, l);
    Level<V> &L = h->lv[l];
    {
        Prof<V> p(h, l, 4);
        RowArgsT<V> a;
        a.x = L.xp; a.b = L.b.p; a.y = r_out; a.partials = L.partials.p;
        const int ns = (int)L.A.n_sets();
        launch_rows_range(L.A, r_out ? ROW_RESNORM : ROW_NORM_ONLY, 0, last_set_done ? ns - 1 : ns, a, h->stream);
    }
    launch_sum_sqrt(L.partials.p, L.A.n_blocks(), out ? out : h->norm_dev.p, h->stream);
}

// Can the restriction INTO level C also apply C's first smoothing launch?  A cycle enters C with
// a zero iterate, so that launch — the first set of a Gauss-Seidel sweep, or a whole Jacobi sweep
// — computes x_i = 0 + (b_i - 0) / a_ii (every term of the row sum is v * 0): no gather needed, and
// the restriction's thread holds b_i (RowArgsT::first_diag).  OMG_NO_FIRST_SWEEP=1 switches it off.
template <typename V>
bool first_sweep_in_restrict(const Hier<V> *h, const Level<V> &C, int pre) {
    if (pre <= 0 || !C.diag.p || C.plan.empty() || C.march || getenv_flag("OMG_NO_FIRST_SWEEP")) return false;
    // (a captured graph bakes the Jacobi ping-pong pointers in: keep two swaps per V(1,1) level there)
    if (h->smoother == OMG_SMOOTH_JACOBI) return !h->want_graph;
    const SweepStep &st = C.plan.front();
    return !st.serial && st.set_begin == 0 && st.set_end == 1;
}

// coarse = R fine; `clear` (nullable, coarse-sized) is zeroed by the same launch — or, with
// first_sweep, set to what the coarse level's first smoothing launch makes of a zero iterate.
template <typename V>
void restrict_level(Hier<V> *h, int l, const V *fine, V *coarse, V *clear = nullptr, bool first_sweep = false) {
    ensure_format(h, l);
    if (first_sweep) ensure_format(h, l + 1);
    Level<V> &L = h->lv[l];
    Prof<V> p(h, l, 2);
    RowArgsT<V> a;
    a.x = fine; a.y = coarse; a.zero = clear; a.ymap = L.r_out.p;
    if (first_sweep && clear) {
        const Level<V> &C = h->lv[l + 1];
        a.first_diag = C.diag.p;
        a.first_jacobi = h->smoother == OMG_SMOOTH_JACOBI;
        a.first_end = int(a.first_jacobi ? C.n : C.A.sets[1]);
        a.omega = h->omega;
    }
    launch_rows(L.R, ROW_SPMV, -1, a, h->stream);
}

template <typename V>
void prolong_add_level(Hier<V> *h, int l, const V *coarse, V *fine) {
    ensure_format(h, l);
    Level<V> &L = h->lv[l];
    Prof<V> p(h, l, 3);
    RowArgsT<V> a;
    a.x = coarse; a.y = fine;
    if (L.scatter_prolong) {
        a.ymap = L.r_out.p;                   // R's rows are in natural coarse order (create())
        launch_rows(L.R, ROW_SCATTER, -1, a, h->stream);
    } else {
        launch_rows(L.P, ROW_AXPY, -1, a, h->stream);
    }
}

template <typename V>
void coarse_solve_level(Hier<V> *h) {
    Level<V> &L = h->lv.back();
    h->coarse.solve(L.b.p, L.xp, h->stream);
}

// What a cycle leaves of the entry level's residual norm (openmg/__init__.py:227)
enum NormState { NORM_NONE = 0, NORM_LAST_SET = 1, NORM_PLANE = 2, NORM_S27 = 3, NORM_VAR7 = 4 };

// Does the cycle over this level run the 27-point kernels of stencil27.hip?  (Switched with the plane passes.)
// The three vectors a large fused level's passes stream side by side (x, its out-of-place twin, b) come out of ONE
// allocation, each starting 2 MiB-aligned + its stagger (common.h vector_stagger): one request to the driver instead of
// three.  Measured at 256^3 over seven alternating processes (tools/ab_libs.py): 0.2758 against 0.2832 ms per cycle, and
// the spread between processes shrinks from 1.3 % to 0.7 % — where three separate allocations land relative to each
// other is what made a process "fast" or "slow".  OMG_VEC_POOL=0: three allocations.
template <typename V>
void pool_views(Level<V> &L, char *base, size_t span, size_t off1, size_t off2);
struct PoolLayout { size_t span, off1, off2; };
template <typename V>
PoolLayout pool_layout(int64_t n) {
    auto env = [](const char *name, size_t dflt) { const char *e = getenv(name); return e && e[0] ? (size_t(atoll(e)) < 64 ? size_t(atoll(e)) : size_t(atoll(e)) / 64 * 64) : dflt; };
    const size_t off1 = env("OMG_POOL_OFF1", vector_stagger(1)), off2 = env("OMG_POOL_OFF2", vector_stagger(2)), pad = env("OMG_POOL_PAD", 0);
    const size_t MB2 = size_t(2) << 20, bytes = size_t(n) * sizeof(V);
    return {(bytes + 2 * DEVBUF_SLACK + std::max(off1, off2) + MB2 - 1) / MB2 * MB2 + pad, off1, off2};
}
constexpr int64_t POOL_TRIAL_MIN = int64_t(1) << 23;       // levels from 8 M unknowns: the placement of their pool is timed (place_finest_pool)
template <typename V>
bool pooled_vectors(Level<V> &L) {
    static const bool on = [] { const char *e = experiment_env("OMG_VEC_POOL"); return !(e && e[0] == '0'); }();
    if (!on || !(L.plane || L.s27) || L.n < (int64_t(1) << 20)) return false;
    if (L.pool.p || L.x.p) return true;
    const PoolLayout q = pool_layout<V>(L.n);
    // (every pooled level of a hierarchy out of ONE allocation — OMG_POOL_ARENA, round 4 — measured no different: removed)
    L.pool.alloc(3 * q.span);
    pool_views(L, L.pool.p, q.span, q.off1, q.off2);
    return true;
}

template <typename V>
void pool_views(Level<V> &L, char *base, size_t span, size_t off1, size_t off2) {
    auto env = [](const char *name, size_t dflt) { const char *e = getenv(name); return e && e[0] ? size_t(atoll(e)) : dflt; };
    // order: x, b, x's twin (OMG_POOL_ORDER=0: x, twin, b).  With b in the middle four of four processes ran 0.260 ms per
    // cycle where the other order gave 0.266-0.274 on the same box; on a second box both orders fell into two populations
    // (0.260 / 0.273) from process to process: where the allocation lands physically still matters, and is not ours to choose
    const bool b_mid = env("OMG_POOL_ORDER", 1) != 0;
    L.x.borrow(reinterpret_cast<V *>(base + DEVBUF_SLACK), size_t(L.n));
    // OMG_POOL_TMP_OWN=1 (experiment): x and b from the pool, x's twin an allocation of its own
    static const bool tmp_own = [] { const char *e = experiment_env("OMG_POOL_TMP_OWN"); return e && e[0] == '1'; }();
    if (tmp_own) L.tmp.alloc(size_t(L.n), off1);
    else
    L.tmp.borrow(reinterpret_cast<V *>(base + (b_mid ? 2 : 1) * span + DEVBUF_SLACK + off1), size_t(L.n));
    L.b.borrow(reinterpret_cast<V *>(base + (b_mid ? 1 : 2) * span + DEVBUF_SLACK + off2), size_t(L.n));
    L.pool_span = span; L.pool_off1 = off1; L.pool_off2 = off2;
}

// WHERE the finest level's pool lands in HBM decides the speed of its passes: at 256^3 fp64 a down + up pass takes 174-179 us
// on some allocations and 187-197 us on others — the "two populations of processes" of rounds 3 and 4, in truth a
// property of each allocation (tools/population_probe.py: three hierarchies made in turn by one process: slow, fast, fast;
// neither the virtual addresses nor the clocks say anything; re-placing the level's other arrays changes nothing; a
// physically CONTIGUOUS pool is the slowest of all, 210-220 us; pieces of 2 / 8 / 32 MiB mapped in a shuffled order are
// fast on some boxes every time and on others no better than hipMalloc: profiles/r05_pool_placement.txt).  Nothing a
// process can ask for decides it, so the placement is MEASURED like the tiling: candidates are allocated one after another
// (ordinary and scattered ones in turn, all held until the end so that each is different memory), the level's own down +
// up pass is timed on each, and the search ends when a candidate is good (the passes at 4.5 TB/s of needed bytes), or as fast
// as the best this process has ever had for the shape, or 7 % faster than the slowest of three or more, or after
// OMG_POOL_TRIALS (8); then, where none was good, vector by vector (below).
template <typename V>
void place_finest_pool(Hier<V> *h) {
    // (read at every call, like OMG_PLACE_KEEP_LAST=1 — tests: the newest candidate is kept whatever its time, no early stop)
    const char *env_trials = getenv("OMG_POOL_TRIALS");
    const int trials = env_trials && env_trials[0] ? atoi(env_trials) : 8;
    const bool keep_last = getenv("OMG_PLACE_KEEP_LAST") != nullptr;
    if ((trials < 2 && !getenv("OMG_POOL_REFINE")) || h->lv.size() < 2) return;
    Level<V> &L = h->lv[0], &C = h->lv[1];
    if (!L.plane || !L.pool.p || !L.pool_span || L.n < POOL_TRIAL_MIN || L.tmp.owned) return;
    SetupTimer tm("placement of the finest level's vectors (timed)");
    typename PlanePlan<V>::Coarse c;
    c.map = L.r_out.p;
    c.b = C.b.p;
    c.e = C.xp;
    auto timed = [&]() -> float {
        L.x.zero(h->stream); L.tmp.zero(h->stream); L.b.zero(h->stream); C.x.zero(h->stream);
        return L.plane->time_pair(L.x.p, L.tmp.p, L.b.p, c, h->stream, true, 4);
    };
    static std::mutex mu;
    static std::map<std::array<int64_t, 2>, float> best_ever;       // (unknowns, bytes per value) -> us
    const std::array<int64_t, 2> key = {L.n, int64_t(sizeof(V))};
    float known = 0.0f;
    {
        std::lock_guard<std::mutex> lock(mu);
        const auto it = best_ever.find(key);
        if (it != best_ever.end()) known = it->second;
    }
    // (the candidates are held together: no more than 16 GB of them — five for a 512^3 level)
    const int max_trials = int(std::min<size_t>(size_t(trials), std::max<size_t>(2, (size_t(16) << 30) / (3 * L.pool_span))));
    // "good": the pair of passes moves its 6 w n bytes at 4.5 TB/s or more (179 us at 256^3 fp64: what all three vectors on the
    // fast kind of memory give; one of them elsewhere: 180-186 us) — where a level's shape never reaches that, every candidate is tried
    const double pair_bytes = 6.0 * double(sizeof(V)) * double(L.n);
    auto good = [&](float us) { return pair_bytes / (double(us) * 1e-6) >= 4.5e12; };
    // Round 6: the level's OTHER stream over x — the matrix-free fine-grid SpMV of the metric (omg_resident_spmv_time) — has
    // its own opinion of where x lies: from a pool in 2 or 4 MiB pieces it takes 68 - 72 us whatever its destination, from
    // one in 8 MiB pieces 44 or 60 - 70 (a property of the process: all of its 8 MiB pools or none), from 32 MiB pieces or an
    // ordinary allocation 44 (profiles/r06_pool_piece_size.txt) — while the passes want the small pieces.  So a candidate
    // that is good for the passes is also asked for its SpMV time (into an ordinary allocation), and the search goes on
    // while the best pool is not good for both; between two pools within 2 % of each other for the passes the one the SpMV
    // likes wins.  (OMG_POOL_SPMV=0: the passes alone decide, as before.)
    DevBuf<V> spmv_probe_y;
    hipEvent_t se0 = nullptr, se1 = nullptr;
    const bool ask_spmv = !L.plane->g.jacobi && !getenv_flag0("OMG_POOL_SPMV");
    const bool debug = SetupTimer::on();
    auto spmv_ok = [&]() -> bool {
        if (!ask_spmv) return true;
        if (!spmv_probe_y.p) {
            try { spmv_probe_y.alloc(size_t(L.n)); } catch (const Error &) { (void)hipGetLastError(); return true; }
            OMG_HIP(hipEventCreate(&se0));
            OMG_HIP(hipEventCreate(&se1));
        }
        L.plane->spmv(L.x.p, spmv_probe_y.p, h->stream);
        OMG_HIP(hipEventRecord(se0, h->stream));
        for (int r = 0; r < 6; ++r) L.plane->spmv(L.x.p, spmv_probe_y.p, h->stream);
        OMG_HIP(hipEventRecord(se1, h->stream));
        OMG_HIP(hipEventSynchronize(se1));
        float ms = 0.0f;
        OMG_HIP(hipEventElapsedTime(&ms, se0, se1));
        const double us = 1e3 * double(ms) / 6.0;
        if (debug) fprintf(stderr, "[omg setup]   ... its matrix-free SpMV: %.1f us per launch\n", us);
        return 2.0 * double(sizeof(V)) * double(L.n) / (us * 1e-6) >= 5.15e12;        // (ensure_spmv_y's line: 52 us at 256^3 fp64)
    };
    float best = timed(), worst = best;
    std::vector<DevBuf<char>> held;                       // (the losers: kept until the end, so that a candidate is not the memory just given back)
    if (debug) fprintf(stderr, "[omg setup] finest level's pool, candidate 0 (hipMalloc): %.1f us per down + up\n", best);
    bool best_spmv = true;
    auto settled = [&](int k) { return good(best) || (k >= 3 && best <= 0.93f * worst) || (known > 0.0f && best <= 1.02f * known); };
    if (!keep_last && settled(1)) best_spmv = spmv_ok();
    for (int k = 1; k < max_trials; ++k) {
        if (!keep_last && settled(k) && best_spmv) break;
        DevBuf<char> alt;
        try { alt.alloc(3 * L.pool_span, 0, pool_placement(k)); } catch (const Error &) { (void)hipGetLastError(); break; }       // (no memory for another candidate: what there is stays)
        std::swap(L.pool, alt);                           // L.pool: the candidate, alt: the best so far
        pool_views(L, L.pool.p, L.pool_span, L.pool_off1, L.pool_off2);
        const float t = timed();
        if (debug) fprintf(stderr, "[omg setup] finest level's pool, candidate %d (placement %d): %.1f us per down + up\n", k, pool_placement(k), t);
        worst = std::max(worst, t);
        // (the SpMV is asked only where the passes' answer alone does not decide)
        bool take = keep_last || t < 0.98f * best;
        bool cand_spmv = true;
        if (!keep_last && t <= 1.02f * best) {
            cand_spmv = spmv_ok();
            take = (cand_spmv && !best_spmv) || (t < best && (cand_spmv || !best_spmv)) || t < 0.98f * best;
        }
        if (take) { best = keep_last ? std::min(best, t) : t; best_spmv = cand_spmv; }
        else std::swap(L.pool, alt);                      // the candidate lost
        held.push_back(std::move(alt));
    }
    if (se0) { (void)hipEventDestroy(se0); (void)hipEventDestroy(se1); }
    {
        std::lock_guard<std::mutex> lock(mu);
        float &e = best_ever[key];
        if (e == 0.0f || best < e) e = best;
    }
    pool_views(L, L.pool.p, L.pool_span, L.pool_off1, L.pool_off2);
    // Where no candidate was good, the search goes on vector by vector: an allocation
    // of its own for one of the three (OMG_POOL_REFINE = candidates per vector, default 4), kept where the passes get 1.5 %
    // faster.  (The times BETWEEN the two kinds are steps: 174 us with all three vectors on the fast kind of memory, + 6 us
    // for each one that is not — profiles/r05_pool_placement.txt, section 10.)
    {
        const char *e = getenv("OMG_POOL_REFINE");
        const bool established = good(best) || (known > 0.0f && best <= 1.02f * known);
        const int K = e && e[0] ? atoi(e) : ((keep_last || trials < 2 || established) ? 0 : 4);
        DevBuf<V> *vecs[3] = {&L.tmp, &L.b, &L.x};
        const size_t shifts[3] = {L.pool_off1, L.pool_off2, 0};
        std::vector<DevBuf<V>> losers;
        for (int v = 0; v < 3 && K > 0; ++v) {
            for (int k = 1; k <= K; ++k) {
                DevBuf<V> alt;
                try { alt.alloc(size_t(L.n), shifts[v], pool_placement(k + v)); } catch (const Error &) { (void)hipGetLastError(); break; }
                std::swap(*vecs[v], alt);
                const float t = timed();
                if (debug) fprintf(stderr, "[omg setup] finest level's vector %d, candidate %d (placement %d): %.1f us (best so far %.1f)\n", v, k, pool_placement(k + v), t, best);
                if (t <= 0.985f * best) { best = t; losers.push_back(std::move(alt)); break; }
                std::swap(*vecs[v], alt);
                losers.push_back(std::move(alt));
            }
        }
    }
    L.x.zero(h->stream); L.tmp.zero(h->stream); L.b.zero(h->stream); C.x.zero(h->stream);
    L.xp = L.x.p;
    L.tp = L.tmp.p;
    OMG_HIP(hipStreamSynchronize(h->stream));
}

// The same for a large 27-point level's coefficient tiles (Stencil27Plan::place_tiles, stencil27.hip)
template <typename V>
void place_s27_tiles(Hier<V> *h) {
    if (h->lv.size() < 2) return;
    Level<V> &L = h->lv[0];
    if (!L.s27 || !L.tmp.p || L.n < POOL_TRIAL_MIN) return;
    L.s27->place_tiles(L.x.p, L.tmp.p, L.b.p, h->stream);
    L.x.zero(h->stream); L.tmp.zero(h->stream); L.b.zero(h->stream);
    OMG_HIP(hipStreamSynchronize(h->stream));
}

// The matrix-free SpMV's destination (an allocation of its own: DESIGN.md section 4) is placed like the level's pool — by
// timing: 45 us per launch on some allocations, 57-69 on others at 256^3 fp64.
template <typename V>
void ensure_spmv_y(Hier<V> *h, Level<V> &L) {
    if (L.spmv_y.p) return;
    L.spmv_y.alloc(size_t(L.n));
    // (16 candidates, 0.5 ms each: the first ones a fresh hipMalloc hands out are what the pool's search has just given back —
    // its losers, the slow kind; a bench process whose eight candidates were all of them ran the SpMV at 68 us, round 6)
    static const int trials = [] { const char *e = getenv("OMG_SPMV_TRIALS"); return e && e[0] ? atoi(e) : 16; }();
    if (trials < 2 || !L.plane || L.n < POOL_TRIAL_MIN) return;
    hipEvent_t e0, e1;
    OMG_HIP(hipEventCreate(&e0));
    OMG_HIP(hipEventCreate(&e1));
    auto timed = [&]() -> float {
        L.plane->spmv(L.xp, L.spmv_y.p, h->stream);
        OMG_HIP(hipEventRecord(e0, h->stream));
        for (int r = 0; r < 8; ++r) L.plane->spmv(L.xp, L.spmv_y.p, h->stream);
        OMG_HIP(hipEventRecord(e1, h->stream));
        OMG_HIP(hipEventSynchronize(e1));
        float ms = 0.0f;
        OMG_HIP(hipEventElapsedTime(&ms, e0, e1));
        return 1e3f * ms / 8.0f;
    };
    float best = timed(), worst = best;
    const bool debug = SetupTimer::on();
    if (debug) fprintf(stderr, "[omg setup] SpMV destination, candidate 0: %.1f us per launch\n", best);
    std::vector<DevBuf<V>> held;
    for (int k = 1; k < trials; ++k) {
        if (2.0 * double(sizeof(V)) * double(L.n) / (double(best) * 1e-6) >= 5.15e12) break;     // good: x read and y written at 5.15 TB/s (52 us at 256^3 fp64; the fast kind 43-45 — 47-50 in these short timings —, the others 57-70)
        DevBuf<V> alt;
        try { alt.alloc(size_t(L.n), 0, pool_placement(k)); } catch (const Error &) { (void)hipGetLastError(); break; }
        std::swap(L.spmv_y, alt);
        const float t = timed();
        if (debug) fprintf(stderr, "[omg setup] SpMV destination, candidate %d (placement %d): %.1f us per launch\n", k, pool_placement(k), t);
        worst = std::max(worst, t);
        if (t < best) best = t;
        else std::swap(L.spmv_y, alt);
        held.push_back(std::move(alt));
    }
    (void)hipEventDestroy(e0);
    (void)hipEventDestroy(e1);
}

template <typename V>
bool use_s27(const Hier<V> *h, const Level<V> &L) {
    return L.s27 && !h->no_plane;
}

template <typename V>
bool use_var7(const Hier<V> *h, const Level<V> &L) {
    return L.var7 && !h->no_plane;
}

// Do both halves of a cycle over this level run as plane-pipelined launches?  (Any sweep counts: with
// pre = 0 or post = 0 — the reference's default is V(1,0), openmg/__init__.py:22-23 — the pass runs
// without its relaxation.)
template <typename V>
bool use_plane(const Hier<V> *h, const Level<V> &L, int pre, int post) {
    (void)pre; (void)post;
    return L.plane && !h->no_plane;
}

// openmg/__init__.py:199-234 with the dead work removed: R[l]*b (:205-206) is computed by the
// reference only for its length, and the norm at levels > entry (:227) is discarded by the
// caller (:213 takes [0]); neither changes any returned value.
//
// want_norm: the caller will ask for ||b - A x|| of THIS level right after the cycle.  Returns what
// is already there of it: NORM_LAST_SET — the post-smoother's last set launch has left that set's
// share in the block partials (norm_level(..., last_set_done = true) finishes it); NORM_PLANE — the
// plane-pipelined up pass has left ALL of it in its workgroup partials (post_slot, or the plan's own
// array: finish_plane_norm).
// x_zero: the level's iterate is zero and has not been written (plane levels only).
template <typename V>
int cycle_body(Hier<V> *h, int l, int pre, int post, bool want_norm = false, double *pre_slot = nullptr,
               bool first_done = false, double *post_slot = nullptr, bool x_zero = false) {
    const int last = (int)h->lv.size() - 1;
    if (l >= last) {
        coarse_solve_level(h);
        return NORM_NONE;
    }
    Level<V> &L = h->lv[l];
    Level<V> &C = h->lv[l + 1];
    if (use_s27(h, L)) {
        // 27-point per-row-coefficient level (stencil27.hip): sweeps as four pair launches, out of place; the last
        // pre-smoothing sweep leaves the residuals of its final rows, the others' are formed inside the restriction
        Stencil27Plan<V> &P = *L.s27;
        // (a 27-point child's first sweep never reads its zero iterate; a plane child's down pass only when it IS the
        // cycle's one pre-smoothing sweep — with more, smooth_level runs before it on the iterate as stored)
        const bool child_zero = l + 1 < last && ((use_s27(h, C) && pre >= 1) || (use_plane(h, C, pre, post) && pre == 1));
        bool zero_now = x_zero;
        if (x_zero && pre == 0) {
            OMG_HIP(hipMemsetAsync(L.xp, 0, size_t(L.n) * sizeof(V), h->stream));
            zero_now = false;
        }
        for (int it = 0; it < pre; ++it) {                      // :201
            Prof<V> p(h, l, 0);
            P.sweep(L.xp, L.tp, L.b.p, zero_now, (it == 0 && !zero_now) ? pre_slot : nullptr, it + 1 == pre, nullptr, h->stream);
            std::swap(L.xp, L.tp);
            zero_now = false;
        }
        if (l == h->pre_level) OMG_HIP(hipMemcpyAsync(h->pre_buf.p, L.xp, size_t(L.n) * sizeof(V), hipMemcpyDeviceToDevice, h->stream));
        if (l + 1 < last && !child_zero) OMG_HIP(hipMemsetAsync(C.xp, 0, size_t(C.n) * sizeof(V), h->stream));   // :191-192
        {
            Prof<V> p(h, l, 1);
            P.residual_restrict(L.xp, L.b.p, pre >= 1, L.r_out.p, C.b.p, h->stream);                  // :209, :210
        }
        cycle_body(h, l + 1, pre, post, false, nullptr, false, nullptr, child_zero);                   // :213
        {
            Prof<V> p(h, l, 3);
            P.prolong(L.xp, C.xp, L.r_out.p, h->stream);                                               // :214, :220 / :224
        }
        double *nslot = want_norm ? (post_slot ? post_slot : P.partials.p) : nullptr;
        for (int it = 0; it < post; ++it) {                     // :216-222
            Prof<V> p(h, l, 0);
            const bool fin = it + 1 == post && want_norm;
            P.sweep(L.xp, L.tp, L.b.p, false, nullptr, fin, fin ? nslot : nullptr, h->stream);
            std::swap(L.xp, L.tp);
        }
        return want_norm ? NORM_S27 : NORM_NONE;
    }
    // OMG_PLANE_HALVES=1|2 (debugging, not under hipGraph): only the down / only the up pass plane-pipelined
    static const int halves = [] { const char *e = experiment_env("OMG_PLANE_HALVES"); return (e && (e[0] == '1' || e[0] == '2')) ? e[0] - '0' : 3; }();
    if (use_plane(h, L, pre, post) && halves == 2 && !x_zero && pre >= 1 && post >= 1) {
        const bool res_done = smooth_level(h, l, pre, FUSE_RESIDUAL, nullptr, first_done);
        residual_level(h, l, L.r.p, res_done);
        if (l + 1 < last) ensure_format(h, l + 1);
        const bool child_first = l + 1 < last && first_sweep_in_restrict(h, C, pre);
        restrict_level<V>(h, l, L.r.p, C.b.p, l + 1 < last ? C.xp : nullptr, child_first);
        cycle_body(h, l + 1, pre, post, false, nullptr, child_first);
        typename PlanePlan<V>::Coarse c;
        c.map = L.r_out.p;
        c.e = C.xp;
        double *out = (want_norm && post == 1) ? (post_slot ? post_slot : L.plane->partials.p) : nullptr;
        L.plane->up(L.xp, L.tp, L.b.p, c, out, h->stream);
        std::swap(L.xp, L.tp);
        if (post > 1)
            return smooth_level(h, l, post - 1, want_norm ? FUSE_NORM : FUSE_NONE, nullptr, false, post_slot) ? NORM_LAST_SET : NORM_NONE;
        return out ? NORM_PLANE : NORM_NONE;
    }
    if (use_plane(h, L, pre, post) && halves == 1 && pre >= 1 && post >= 1) {
        if (pre > 1) smooth_level(h, l, pre - 1, FUSE_NONE, nullptr, first_done);
        if (l + 1 < last) ensure_format(h, l + 1);
        const bool child_first = l + 1 < last && first_sweep_in_restrict(h, C, pre);
        typename PlanePlan<V>::Coarse c;
        c.map = L.r_out.p;
        c.b = C.b.p;
        c.x = l + 1 < last ? C.xp : nullptr;
        c.diag = child_first ? C.diag.p : nullptr;
        c.first_end = child_first ? int(C.A.sets[1]) : 0;
        L.plane->down(L.xp, L.tp, L.b.p, x_zero, c, h->stream);
        std::swap(L.xp, L.tp);
        cycle_body(h, l + 1, pre, post, false, nullptr, child_first);
        prolong_add_level<V>(h, l, C.xp, L.xp);
        return smooth_level(h, l, post, want_norm ? FUSE_NORM : FUSE_NONE, nullptr, false, post_slot) ? NORM_LAST_SET : NORM_NONE;
    }
    if (use_plane(h, L, pre, post)) {
        // :201 all but the last pre-smoothing sweep set by set (in place); the last one inside the down pass
        if (pre > 1) smooth_level(h, l, pre - 1, FUSE_NONE, nullptr, first_done);
        const bool child_plane = l + 1 < last && use_plane(h, C, pre, post);
        const bool child_zero = child_plane && pre <= 1 && halves == 3;       // the child's down pass never reads its zero iterate
        if (l + 1 < last && !child_zero) ensure_format(h, l + 1);
        const bool child_first = l + 1 < last && !child_zero && !use_s27(h, C) && !L.plane->g.dim2 && first_sweep_in_restrict(h, C, pre);
        typename PlanePlan<V>::Coarse c;
        c.map = L.r_out.p;
        c.b = C.b.p;
        c.x = (l + 1 < last && !child_zero) ? C.xp : nullptr;
        c.diag = child_first ? C.diag.p : nullptr;
        c.first_end = child_first ? int(C.A.sets[1]) : 0;
        // (pre = 0: the child's down pass does not write its iterate either, and its up pass reads it)
        if (child_zero && pre == 0) OMG_HIP(hipMemsetAsync(C.xp, 0, size_t(C.n) * sizeof(V), h->stream));
        {
            Prof<V> p(h, l, 5);
            L.plane->down(L.xp, L.tp, L.b.p, x_zero, c, h->stream, nullptr, pre >= 1);      // :201 (last sweep), :209, :210
        }
        if (pre >= 1) std::swap(L.xp, L.tp);
        if (l == h->pre_level) OMG_HIP(hipMemcpyAsync(h->pre_buf.p, L.xp, size_t(L.n) * sizeof(V), hipMemcpyDeviceToDevice, h->stream));
        cycle_body(h, l + 1, pre, post, false, nullptr, child_first, nullptr, child_zero);   // :213
        c.e = C.xp;
        double *out = (want_norm && post <= 1) ? (post_slot ? post_slot : L.plane->partials.p) : nullptr;
        {
            Prof<V> p(h, l, 6);
            L.plane->up(L.xp, L.tp, L.b.p, c, out, h->stream, nullptr, post >= 1);           // :214, :220/:224, first sweep of :216-222 (, :227)
        }
        std::swap(L.xp, L.tp);
        if (post > 1)
            return smooth_level(h, l, post - 1, want_norm ? FUSE_NORM : FUSE_NONE, nullptr, false, post_slot) ? NORM_LAST_SET : NORM_NONE;
        return out ? NORM_PLANE : NORM_NONE;
    }
    if (use_var7(h, L)) {
        // 7-point per-row-coefficient level (var7.hip): all but the last pre-smoothing sweep set by set (in place), the last
        // one inside the down pass; the up pass holds the first post-smoothing sweep and the norm's squares
        Var7Plan<V> &P = *L.var7;
        if (pre > 1) smooth_level(h, l, pre - 1, FUSE_NONE, nullptr, first_done);
        const bool child_zero = l + 1 < last && ((use_var7(h, C) && pre == 1) || (use_plane(h, C, pre, post) && pre == 1) || (use_s27(h, C) && pre >= 1));
        if (l + 1 < last && !child_zero) {
            if (!use_s27(h, C)) ensure_format(h, l + 1);
            OMG_HIP(hipMemsetAsync(C.xp, 0, size_t(C.n) * sizeof(V), h->stream));      // :191-192
        }
        typename Var7Plan<V>::Coarse c;
        c.map = L.r_out.p;
        c.b = C.b.p;
        {
            Prof<V> p(h, l, 5);
            P.down(L.xp, L.tp, L.b.p, x_zero && pre == 1, c, h->stream, pre >= 1);       // :201 (last sweep), :209, :210
        }
        if (pre >= 1) std::swap(L.xp, L.tp);
        else if (x_zero) OMG_HIP(hipMemsetAsync(L.xp, 0, size_t(L.n) * sizeof(V), h->stream));
        if (l == h->pre_level) OMG_HIP(hipMemcpyAsync(h->pre_buf.p, L.xp, size_t(L.n) * sizeof(V), hipMemcpyDeviceToDevice, h->stream));
        cycle_body(h, l + 1, pre, post, false, nullptr, false, nullptr, child_zero);      // :213
        c.e = C.xp;
        double *out = (want_norm && post <= 1) ? (post_slot ? post_slot : P.partials.p) : nullptr;
        {
            Prof<V> p(h, l, 6);
            P.up(L.xp, L.tp, L.b.p, c, out, h->stream, post >= 1);                       // :214, :220/:224, first sweep of :216-222 (, :227)
        }
        std::swap(L.xp, L.tp);
        if (post > 1)
            return smooth_level(h, l, post - 1, want_norm ? FUSE_NORM : FUSE_NONE, nullptr, false, post_slot) ? NORM_LAST_SET : NORM_NONE;
        return out ? NORM_VAR7 : NORM_NONE;
    }
    const bool res_done = smooth_level(h, l, pre, FUSE_RESIDUAL, pre_slot, first_done);   // :201 (+ last set's share of :209)
    if (l == h->pre_level) OMG_HIP(hipMemcpyAsync(h->pre_buf.p, L.xp, size_t(L.n) * sizeof(V), hipMemcpyDeviceToDevice, h->stream));
    residual_level(h, l, L.r.p, res_done);                          // :209
    // :210, and the coarse cycle's initial=None -> zeros (:191-192) cleared by the same launch —
    // or already relaxed once (first_sweep_in_restrict)
    const bool child_zero = l + 1 < last && ((use_plane(h, C, pre, post) && pre == 1 && halves == 3) || (use_s27(h, C) && pre >= 1));
    if (l + 1 < last && !child_zero && !use_s27(h, C)) ensure_format(h, l + 1);
    const bool child_first = l + 1 < last && !child_zero && !use_s27(h, C) && first_sweep_in_restrict(h, C, pre);
    restrict_level<V>(h, l, L.r.p, C.b.p, (l + 1 < last && !child_zero) ? C.xp : nullptr, child_first);
    cycle_body(h, l + 1, pre, post, false, nullptr, child_first, nullptr, child_zero);   // :213
    prolong_add_level<V>(h, l, C.xp, L.xp);                         // :214, :220/:224
    if (post > 0)
        return smooth_level(h, l, post, want_norm ? FUSE_NORM : FUSE_NONE, nullptr, false, post_slot) ? NORM_LAST_SET : NORM_NONE;   // :216-222
    return NORM_NONE;
}

// ||b - A x|| of level l into out (device scalar; null: h->norm_dev) after a cycle that returned `state`
template <typename V>
void finish_norm(Hier<V> *h, int l, int state, double *out = nullptr) {
    if (state == NORM_PLANE) {
        Level<V> &L = h->lv[l];
        launch_sum_sqrt(L.plane->partials.p, L.plane->g.n_wg, out ? out : h->norm_dev.p, h->stream);
    } else if (state == NORM_VAR7) {
        Level<V> &L = h->lv[l];
        launch_sum_sqrt(L.var7->partials.p, L.var7->n_wg, out ? out : h->norm_dev.p, h->stream);
    } else if (state == NORM_S27) {
        // the rows the last post-smoothing launch made final have left their squares (segment 3); the others' now
        Level<V> &L = h->lv[l];
        Stencil27Plan<V> &P = *L.s27;
        {
            Prof<V> p(h, l, 4);
            P.norm(L.xp, L.b.p, P.have67, P.partials.p, h->stream);
        }
        launch_sum_sqrt(P.partials.p, int64_t(4) * P.g.n_wg, out ? out : h->norm_dev.p, h->stream);
    } else {
        norm_level<V>(h, l, nullptr, state == NORM_LAST_SET, out);
    }
}

// Host vectors are double for either V.  A double level in its natural ordering copies
// straight through; otherwise the vector is staged in `nat` and permuted / converted by one
// gather (idx == NULL: conversion only).
template <typename V>
constexpr bool direct_io(const Level<V> &L) { return std::is_same<V, double>::value && L.ord.identity; }

template <typename V>
void ensure_nat(Hier<V> *h, int l) {
    Level<V> &L = h->lv[l];
    if (!direct_io(L) && L.nat.n < size_t(std::max<int64_t>(L.n, 1))) L.nat.alloc(std::max<int64_t>(L.n, 1));
}

template <typename V>
void load_vec(Hier<V> *h, int l, const double *host, V *dst) {
    Level<V> &L = h->lv[l];
    if (direct_io(L)) {
        OMG_HIP(hipMemcpyAsync(dst, host, L.n * sizeof(double), hipMemcpyHostToDevice, h->stream));
    } else {
        ensure_nat(h, l);
        L.nat.upload(host, L.n, h->stream);
        launch_gather<double, V>(L.nat.p, L.ord.identity ? nullptr : L.perm.p, dst, L.n, h->stream);
    }
}

// The wavefront sweeps give up waiting for a face after a bounded number of polls instead of
// hanging the device; whoever reads results back asks whether that has happened.
template <typename V>
void check_plane_status(Hier<V> *h, uint32_t st) {
    if (!st) return;
    OMG_HIP(hipMemsetAsync(h->plane_status.p, 0, sizeof(st), h->stream));
    throw Error(OMG_ERR_HIP, "a plane-pipelined pass gave up waiting for a neighbouring wave of its workgroup (status " +
                             std::to_string(st) + "): the results since the last read-back are not valid");
}

// (have_plane_status: the caller has already copied the plane passes' status word back with its own results)
template <typename V>
void check_march(Hier<V> *h, bool have_plane_status = false, uint32_t plane_st = 0) {
    for (Level<V> &L : h->lv)
        if (L.march && L.march->timed_out(h->stream))
            throw Error(OMG_ERR_HIP, "lexicographic wavefront sweep timed out waiting for a neighbouring tile's face; the iterate and "
                                     "the hand-over slots of this hierarchy are no longer consistent: destroy it and build a new one");
    // the plane passes' waves wait for their neighbour waves a bounded number of polls (plane.hip WAVE_SYNC_SPIN)
    if (h->plane_status.p && !have_plane_status) {
        OMG_HIP(hipMemcpyAsync(&plane_st, h->plane_status.p, sizeof(plane_st), hipMemcpyDeviceToHost, h->stream));
        OMG_HIP(hipStreamSynchronize(h->stream));
    }
    check_plane_status(h, plane_st);
}

template <typename V>
void fetch_vec(Hier<V> *h, int l, const V *src, double *host) {
    Level<V> &L = h->lv[l];
    // (large vectors through the pinned staging buffers of download_staged: the runtime's own copy into a pageable
    // NumPy array ran at ~10 GB/s)
    if (direct_io(L)) {
        download_staged(host, src, size_t(L.n) * sizeof(double), h->stream);
    } else {
        ensure_nat(h, l);
        launch_scatter<V, double>(src, L.ord.identity ? nullptr : L.perm.p, L.nat.p, L.n, h->stream);
        download_staged(host, L.nat.p, size_t(L.n) * sizeof(double), h->stream);
    }
    check_march(h);
}

// The same two for a caller whose vectors already live in HBM (double, natural numbering): device-to-device
template <typename V>
void load_vec_dev(Hier<V> *h, int l, const double *dev, V *dst) {
    Level<V> &L = h->lv[l];
    if (direct_io(L)) OMG_HIP(hipMemcpyAsync(dst, dev, L.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    else launch_gather<double, V>(dev, L.ord.identity ? nullptr : L.perm.p, dst, L.n, h->stream);
}
template <typename V>
void fetch_vec_dev(Hier<V> *h, int l, const V *src, double *dev) {
    Level<V> &L = h->lv[l];
    if (direct_io(L)) OMG_HIP(hipMemcpyAsync(dev, src, L.n * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
    else launch_scatter<V, double>(src, L.ord.identity ? nullptr : L.perm.p, dev, L.n, h->stream);
}

template <typename V>
double read_norm(Hier<V> *h) {
    double v = 0.0;
    uint32_t st = 0;
    OMG_HIP(hipMemcpyAsync(&v, h->norm_dev.p, sizeof(double), hipMemcpyDeviceToHost, h->stream));
    if (h->plane_status.p) OMG_HIP(hipMemcpyAsync(&st, h->plane_status.p, sizeof(st), hipMemcpyDeviceToHost, h->stream));
    OMG_HIP(hipStreamSynchronize(h->stream));
    check_march(h, true, st);
    return v;
}

template <typename V>
void drop_graph(Hier<V> *h) {
    for (auto &g : h->graphs) if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->graphs.clear();
}

template <typename V>
std::vector<V *> pointer_state(const Hier<V> *h) {
    std::vector<V *> s;
    s.reserve(2 * h->lv.size());
    for (const Level<V> &L : h->lv) { s.push_back(L.xp); s.push_back(L.tp); }
    return s;
}

// One level-0 cycle + residual norm into norm_dev; optionally replayed from a hipGraph.
template <typename V>
void run_cycle0(Hier<V> *h, int pre, int post) {
    const bool single = h->lv.size() == 1;
    auto body = [&]() {
        const int part = cycle_body(h, 0, pre, post, !single);
        if (!single) finish_norm(h, 0, part);                 // :227
        else OMG_HIP(hipMemsetAsync(h->norm_dev.p, 0, sizeof(double), h->stream));   // :232
    };
    if (!h->want_graph || h->profiling) {
        body();
        return;
    }
    const std::vector<V *> cur = pointer_state(h);
    auto restore = [&](const std::vector<V *> &st) {
        for (size_t l = 0; l < h->lv.size(); ++l) { h->lv[l].xp = st[2 * l]; h->lv[l].tp = st[2 * l + 1]; }
    };
    typename Hier<V>::GraphSlot *slot = nullptr;
    for (auto &g : h->graphs)
        if (g.pre == pre && g.post == post && g.before == cur) slot = &g;
    if (!slot) {
        // (two phases of one (pre, post) at most: anything else — other sweep counts, a state no captured cycle
        // leaves — starts over)
        if (h->graphs.size() >= 2 || (!h->graphs.empty() && (h->graphs[0].pre != pre || h->graphs[0].post != post))) drop_graph(h);
        hipGraph_t g = nullptr;
        OMG_HIP(hipStreamBeginCapture(h->stream, hipStreamCaptureModeThreadLocal));
        try {
            body();
        } catch (...) {
            (void)hipStreamEndCapture(h->stream, &g);
            if (g) (void)hipGraphDestroy(g);
            restore(cur);
            throw;
        }
        OMG_HIP(hipStreamEndCapture(h->stream, &g));
        typename Hier<V>::GraphSlot ns;
        ns.pre = pre;
        ns.post = post;
        ns.before = cur;
        ns.after = pointer_state(h);               // (the capture ran the host side of the cycle: the swaps are made)
        hipError_t e = hipGraphInstantiate(&ns.exec, g, nullptr, nullptr, 0);
        (void)hipGraphDestroy(g);
        if (e != hipSuccess) {
            restore(cur);
            throw Error(OMG_ERR_HIP, std::string("hipGraphInstantiate: ") + hipGetErrorString(e));
        }
        h->graphs.push_back(std::move(ns));
        slot = &h->graphs.back();
    }
    OMG_HIP(hipGraphLaunch(slot->exec, h->stream));
    restore(slot->after);
}

template <typename V>
void check_level(const Hier<V> *h, int level) {
    OMG_REQUIRE(level >= 0 && level < (int)h->lv.size(), "level out of range");
}


// The row-kernel side of a smoothed level: A and R (and P where the prolongation cannot scatter over R) in
// the device format of common.h, the residual vector, the block partials, the sweep plan.  A, R: natural
// numbering (the caller's, or — for a plane level, later — what its plan describes).
template <typename V>
void build_format(Hier<V> *h, int l, const omg_csr &A, const omg_csr &R) {
    Level<V> &L = h->lv[l];
    const bool id = L.ord.identity;
    {
        HostCsr Ap;
        { SetupTimer tm("permute A"); Ap = permute_csr(A, id ? nullptr : L.ord.perm.data(), id ? nullptr : L.ord.inv.data()); }
        SetupTimer tm("encode + upload A");
        L.A.upload(Ap, L.ord.sets, h->stream);
        // the diagonal as the sweeps form it: restrict_level / omg_hierarchy_cycle_dev apply this
        // level's first relaxation of a zero iterate with it (level 0: only when it is needed)
        if (l >= 1) {
            L.diag.alloc(std::max<int64_t>(L.n, 1));
            launch_diagonal(L.A, L.diag.p, h->stream);
        }
    }
    const Ordering &co = h->lv[l + 1].ord;
    SetupTimer tm("R (and P): permute, encode, upload");
    HostCsr Rp, Rn;
    const HostCsr *r_used = &Rp;
    if (co.identity) {
        Rp = permute_csr(R, nullptr, id ? nullptr : L.ord.inv.data());
        L.R.upload(Rp, {}, h->stream);
    } else {
        // Restriction rows stay in NATURAL coarse order and write through a map (r_out) into the
        // coarse level's ordering: consecutive rows are then consecutive aggregates,
        // whose fine unknowns are consecutive inside each colour segment (dense
        // gathers).  In the coarse COLOUR order consecutive rows are every other
        // aggregate and each gather used half of every cache line (measured: 480 MB
        // read for 343 MB algorithmic).
        Rn = permute_csr(R, nullptr, id ? nullptr : L.ord.inv.data());
        r_used = &Rn;
        L.R.upload(Rn, {}, h->stream);
    }
    {   // OMG_PROLONG_SCATTER=0: always the explicit transpose (identical bits, tested)
        const char *e = getenv("OMG_PROLONG_SCATTER");
        bool ok = !(e && e[0] == '0') && L.R.all_pattern();
        if (ok) {
            std::vector<char> seen(size_t(L.n), 0);
            for (int32_t c : r_used->indices) {
                if (seen[c]) { ok = false; break; }
                seen[c] = 1;
            }
        }
        L.scatter_prolong = ok;
    }
    if (!L.scatter_prolong) {
        // the explicit transpose P = R^T (rows in this level's ordering, columns in the next
        // level's) is only built where the prolongation cannot run as a scatter over R's rows
        if (!co.identity) Rp = permute_csr(R, co.perm.data(), id ? nullptr : L.ord.inv.data());
        HostCsr Pt = transpose_csr(Rp);
        L.P.upload(Pt, {}, h->stream);
    }
    L.r.alloc(L.n);
    if (h->smoother == OMG_SMOOTH_JACOBI && !L.tmp.p) { L.tmp.alloc(L.n); L.tp = L.tmp.p; }
    L.partials.alloc(L.A.n_blocks() + SUM_FOLD);
    build_plan(L);
    OMG_HIP(hipStreamSynchronize(h->stream));
}

// A plane level's row-kernel side, on first use.
template <typename V>
void ensure_format(Hier<V> *h, int l) {
    if (l < 0 || l >= (int)h->lv.size() || !h->lv[l].format_pending) return;
    Level<V> &L = h->lv[l];
    SetupTimer tm("plane / 27-point level: row-kernel format on first use");
    materialise_ordering(L.ord);                                   // (a hierarchy set up on the device keeps its closed-form
    if (l + 1 < (int)h->lv.size()) materialise_ordering(h->lv[l + 1].ord);   //  orderings as device arrays only until here)
    // (a 27-point level: from the operator PADDED to 27 entries per row, so that the row kernels associate every row's
    // sum as the kernels of stencil27.hip do)
    const HostCsr A = L.s27 ? L.s27->operator_csr(h->stream) : L.var7 ? L.var7->operator_csr(h->stream) : L.plane->operator_csr();
    const HostCsr R = L.s27 ? L.s27->restriction_csr() : L.var7 ? L.var7->restriction_csr() : L.plane->restriction_csr();
    L.format_pending = false;
    build_format(h, l, view(A), view(R));
}
template <typename V>
void ensure_format(const Hier<V> *h, int l) { ensure_format(const_cast<Hier<V> *>(h), l); }


template <typename V>
std::unique_ptr<Hier<V>> create(int n_levels, const omg_csr *A, const omg_csr *R, int smoother,
                                double omega) {
    using H = Hier<V>;
    using Lv = Level<V>;
    OMG_REQUIRE(n_levels >= 1, "n_levels must be >= 1");
    OMG_REQUIRE(A != nullptr && (n_levels == 1 || R != nullptr), "A / R array is null");
    OMG_REQUIRE(smoother >= OMG_SMOOTH_GS_LEX && smoother <= OMG_SMOOTH_JACOBI, "unknown smoother");
    for (int l = 0; l < n_levels; ++l) {
        validate_csr(A[l], ("A[" + std::to_string(l) + "]").c_str());
        OMG_REQUIRE(A[l].n_rows == A[l].n_cols, "A[l] must be square");
        if (l + 1 < n_levels) {
            validate_csr(R[l], ("R[" + std::to_string(l) + "]").c_str());
            OMG_REQUIRE(R[l].n_cols == A[l].n_rows && R[l].n_rows == A[l + 1].n_rows,
                        "R[l] shape does not match A[l], A[l+1]");
            check_diagonal(A[l], l);
        }
    }
    require_device();
    std::unique_ptr<H> h(new H);
    h->smoother = smoother;
    h->omega = omega;
    OMG_HIP(hipStreamCreateWithFlags(&h->own, hipStreamNonBlocking));
    h->stream = h->own;
    h->lv.resize(n_levels);
    h->norm_dev.alloc(1);
    // The coarsest operator is factored once (reference: SuperLU factorisation on every
    // cycle; here: explicit inverse or substructuring along the band, common.h CoarseSolver).
    // That is thousands of tiny dependent launches, so it runs on a helper thread with its own
    // stream while this thread does the index work of the smoothed levels.  The factors are
    // always computed in double; a float hierarchy stores their rounding.
    int device = 0;
    OMG_HIP(hipGetDevice(&device));
    {
        Lv &L = h->lv.back();
        L.n = A[n_levels - 1].n_rows;
        L.ord.identity = true;
        L.ord.sets = {0, L.n};
    }
    int inv_code = OMG_OK;
    std::string inv_msg;
    std::thread inverter([&] {
        hipStream_t s = nullptr;
        SetupTimer tm("coarse factorisation (helper thread)");
        try {
            OMG_HIP(hipSetDevice(device));
            OMG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            Lv &L = h->lv.back();
            HostCsr Ap = permute_csr(A[n_levels - 1], nullptr, nullptr);
            L.A.upload(Ap, L.ord.sets, s);
            h->coarse.build(Ap, s);               // factors in double, stored in V
        } catch (const Error &e) {
            inv_code = e.code;
            inv_msg = e.what();
        } catch (const std::exception &e) {
            inv_code = OMG_ERR_HIP;
            inv_msg = e.what();
        }
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{inverter};
    // orderings first (restrictions need both neighbours')
    for (int l = 0; l + 1 < n_levels; ++l) {
        Lv &L = h->lv[l];
        L.n = A[l].n_rows;
        if (smoother == OMG_SMOOTH_GS_COLOUR || smoother == OMG_SMOOTH_JACOBI) {
            SetupTimer tp("plane-pipelined passes: does the level qualify (+ its parity ordering)");
            std::unique_ptr<PlanePlan<V>> plan(new PlanePlan<V>);
            L.march.reset();
            // (weighted Jacobi: 2-D levels only — BASELINE configs[1] — in their natural ordering)
            if (plan->build(A[l], R[l], L.ord, smoother == OMG_SMOOTH_JACOBI, omega)) {
                if (!h->plane_status.p) { h->plane_status.alloc(1); h->plane_status.zero(h->stream); }
                plan->status = h->plane_status.p;
                L.plane = std::move(plan);
                // OMG_PLANE_CHECK_ORDER=1 (tests): the closed-form ordering is the greedy colouring's
                if (getenv_flag("OMG_PLANE_CHECK_ORDER") && smoother == OMG_SMOOTH_GS_COLOUR) {
                    const Ordering g = make_ordering(A[l], smoother);
                    OMG_REQUIRE(g.sets == L.ord.sets && g.perm == L.ord.perm && g.inv == L.ord.inv,
                                "internal: the parity ordering differs from the greedy colouring");
                }
                continue;
            }
            std::unique_ptr<Stencil27Plan<V>> s27(new Stencil27Plan<V>);
            if (smoother == OMG_SMOOTH_GS_COLOUR && s27->build(A[l], R[l], L.ord, h->stream)) {
                L.s27 = std::move(s27);
                if (getenv_flag("OMG_PLANE_CHECK_ORDER")) {
                    const Ordering g = make_ordering(A[l], smoother);
                    OMG_REQUIRE(g.sets == L.ord.sets && g.perm == L.ord.perm && g.inv == L.ord.inv,
                                "internal: the octant ordering differs from the greedy colouring");
                }
                continue;
            }
        }
        // (OMG_PLANE=0 asks for the set-by-set schedule: no fused path of any kind)
        if (smoother == OMG_SMOOTH_GS_COLOUR && !getenv_flag0("OMG_VAR7") && !getenv_flag0("OMG_PLANE")) {
            SetupTimer tv("7-point level with per-row coefficients: does it qualify (+ its coefficient arrays, its parity ordering)");
            std::unique_ptr<Var7Plan<V>> v7(new Var7Plan<V>);
            if (v7->build(A[l], R[l], L.ord, h->stream)) {
                L.var7 = std::move(v7);
                if (getenv_flag("OMG_PLANE_CHECK_ORDER")) {
                    const Ordering g = make_ordering(A[l], smoother);
                    OMG_REQUIRE(g.sets == L.ord.sets && g.perm == L.ord.perm && g.inv == L.ord.inv,
                                "internal: the parity ordering differs from the greedy colouring");
                }
                continue;
            }
        }
        SetupTimer tm("ordering (colouring / level schedule / wavefront plan)");
        order_level(L, A[l], smoother, h->stream);
    }
    for (int l = 0; l < n_levels; ++l) {
        Lv &L = h->lv[l];
        const bool id = L.ord.identity;
        if (!id) {
            L.perm.alloc(L.n);
            L.perm.upload(L.ord.perm.data(), L.n, h->stream);
            OMG_HIP(hipStreamSynchronize(h->stream));
        }
        if (l + 1 < n_levels) {
            const Ordering &co = h->lv[l + 1].ord;
            if (!co.identity) {
                // restriction rows are in NATURAL coarse order and write through this map into the coarse
                // level's ordering (build_format; the plane passes use it too)
                L.r_out.alloc(co.inv.size());
                L.r_out.upload(co.inv.data(), co.inv.size(), h->stream);
                OMG_HIP(hipStreamSynchronize(h->stream));
            }
            if (L.plane && !pooled_vectors(L)) L.tmp.alloc(L.n, vector_stagger(1));
            if (L.s27 && !pooled_vectors(L)) L.tmp.alloc(L.n);
            if (L.var7 && !L.tmp.p) L.tmp.alloc(L.n);
            if ((L.plane || L.s27 || L.var7) && !experiment_flag0("OMG_PLANE_LAZY")) L.format_pending = true;
            else if (L.s27) { L.format_pending = true; ensure_format(h.get(), l); }      // (from the padded operator)
            else build_format(h.get(), l, A[l], R[l]);
        }
        if (!L.x.p) {
            L.x.alloc(std::max<int64_t>(L.n, 1));
            L.b.alloc(std::max<int64_t>(L.n, 1), L.plane ? vector_stagger(2) : 0);
        }
        // (finite from the start: the 27-point sweeps multiply the slot of a neighbour that does not exist — a zero
        // coefficient — with whatever the vectors hold at the index the slot's shift lands on)
        L.x.zero(h->stream);
        L.tmp.zero(h->stream);
        L.xp = L.x.p;
        L.tp = L.tmp.p;
        if (SetupTimer::on() && l == 0) fprintf(stderr, "[omg setup] level 0 vectors at x %p  tmp %p  b %p\n", (void *)L.x.p, (void *)L.tmp.p, (void *)L.b.p);
    }
    for (int l = 0; l + 1 < n_levels; ++l) {
        // the large levels' tilings: measured, not modelled (plane.hip choose_tiles; tune() skips the small ones)
        if (!h->lv[l].plane || !h->lv[l].tmp.p) continue;
        SetupTimer tm("plane tiling of a large level");
        Level<V> &L = h->lv[l], &C = h->lv[l + 1];
        L.x.zero(h->stream); L.tmp.zero(h->stream); L.b.zero(h->stream); C.x.zero(h->stream);
        typename PlanePlan<V>::Coarse c;
        c.map = L.r_out.p;
        c.b = C.b.p;
        c.e = C.xp;
        L.plane->tune(L.xp, L.tp, L.b.p, c, h->stream, l == 0);
        OMG_HIP(hipStreamSynchronize(h->stream));
    }
    // (first: the coarse factorisation's kernels on their own stream beside the timed passes would make a candidate look slow
    // and the decision depend on the overlap)
    { SetupTimer tm("wait for the coarse factorisation"); inverter.join(); }
    place_finest_pool(h.get());
    place_s27_tiles(h.get());
    if (inv_code != OMG_OK) throw Error(inv_code, inv_msg);
    OMG_HIP(hipStreamSynchronize(h->stream));
    return h;
}

// mgSolve's setup (openmg/__init__.py:103-109) without the host in between: restrictionList + coeffecientList on the
// device (galerkin_chain_device), every smoothed level qualified THERE for a fused path (plane passes / 2-D tile passes /
// 27-point kernels) — the grid is the caller's problemShape —, orderings and slot maps written by kernels.  What comes
// back to the host: the coarsest operator (a few thousand entries) for its factorisation.  A hierarchy with a level
// that does not qualify takes the ordinary route (its operators are fetched once): the result is the same object.
template <typename V>
std::unique_ptr<Hier<V>> create_from_fine(const omg_csr &A0, int dim, const int64_t *shape, int n_restrictions, int smoother, double omega) {
    using H = Hier<V>;
    using Lv = Level<V>;
    OMG_REQUIRE(dim >= 2 && dim <= 3 && shape && n_restrictions >= 1, "device setup: 2-D / 3-D grids, at least one restriction");
    OMG_REQUIRE(smoother >= OMG_SMOOTH_GS_LEX && smoother <= OMG_SMOOTH_JACOBI, "unknown smoother");
    validate_csr(A0, "A_in");
    OMG_REQUIRE(A0.n_rows == A0.n_cols, "A_in must be square");
    // the reference's restriction is the plain aggregation when its quirky second-axis offset shape[0] (Q6) equals the
    // true one: first and last extent equal
    OMG_REQUIRE(shape[0] == shape[dim - 1], "device setup: first and last extent must be equal");
    require_device();
    std::unique_ptr<H> h(new H);
    h->smoother = smoother;
    h->omega = omega;
    OMG_HIP(hipStreamCreateWithFlags(&h->own, hipStreamNonBlocking));
    h->stream = h->own;
    h->norm_dev.alloc(1);
    std::vector<DevCsrPlain> dA, dR;
    galerkin_chain_device(A0, dim, shape, n_restrictions, dA, dR, h->stream);
    const int n_levels = n_restrictions + 1;
    h->lv.resize(size_t(n_levels));
    const double w = 1.0 / double(1 << dim);
    auto dims = [&](int l, int &nx, int &ny, int &nz) {
        nx = int(shape[dim - 1] >> l);
        ny = int(shape[dim - 2] >> l);
        nz = dim == 3 ? int(shape[0] >> l) : 1;
    };
    bool all = smoother == OMG_SMOOTH_GS_COLOUR || (smoother == OMG_SMOOTH_JACOBI && dim == 2);
    // Round 6: a hierarchy whose LARGE levels qualify (7-point levels with per-row coefficients take the fused passes from
    // 128^3 up) keeps them on the device; the small levels below — a few hundred thousand rows — are fetched and ordered / coded
    // by the host as the ordinary route would (host_level), each a few milliseconds.
    constexpr int64_t HOST_LEVEL_MAX = int64_t(1) << 19;
    std::vector<char> host_level((size_t)(n_levels), 0);
    std::vector<HostCsr> hostA((size_t)(n_levels)), hostR((size_t)(n_levels));
    bool any_var7 = false;
    for (int l = 0; all && l + 1 < n_levels; ++l) {
        Lv &L = h->lv[size_t(l)];
        L.n = dA[size_t(l)].n_rows;
        int nx, ny, nz;
        dims(l, nx, ny, nz);
        SetupTimer tp("device setup: does the level qualify for a fused path");
        std::unique_ptr<PlanePlan<V>> plan(new PlanePlan<V>);
        if (plan->build_device(dA[size_t(l)], nx, ny, nz, w, L.ord, smoother == OMG_SMOOTH_JACOBI, omega, h->stream)) {
            if (!h->plane_status.p) { h->plane_status.alloc(1); h->plane_status.zero(h->stream); }
            plan->status = h->plane_status.p;
            L.plane = std::move(plan);
            continue;
        }
        std::unique_ptr<Stencil27Plan<V>> s27(new Stencil27Plan<V>);
        if (smoother == OMG_SMOOTH_GS_COLOUR && dim == 3 && s27->build_device(dA[size_t(l)], nx, ny, nz, w, L.ord, h->stream)) {
            L.s27 = std::move(s27);
            continue;
        }
        if (smoother == OMG_SMOOTH_GS_COLOUR && dim == 3 && !getenv_flag0("OMG_VAR7") && !getenv_flag0("OMG_PLANE")) {
            std::unique_ptr<Var7Plan<V>> v7(new Var7Plan<V>);
            if (v7->build_device(dA[size_t(l)], nx, ny, nz, w, L.ord, h->stream)) {
                L.var7 = std::move(v7);
                any_var7 = true;
                continue;
            }
        }
        if (any_var7 && L.n <= HOST_LEVEL_MAX) {
            host_level[size_t(l)] = 1;
            hostA[size_t(l)] = download_csr(dA[size_t(l)], h->stream);
            hostR[size_t(l)] = download_csr(dR[size_t(l)], h->stream);
            check_diagonal(view(hostA[size_t(l)]), l);
            SetupTimer tm("ordering (colouring / level schedule / wavefront plan)");
            order_level(L, view(hostA[size_t(l)]), smoother, h->stream);
            continue;
        }
        all = false;
    }
    if (!all) {
        // some level needs the host's orderings and codings: fetch the operators once, then the ordinary route
        SetupTimer tm("device setup: a level does not qualify: operators fetched, ordinary route");
        std::vector<HostCsr> hA, hR;
        for (auto &M : dA) hA.push_back(download_csr(M, h->stream));
        for (auto &M : dR) hR.push_back(download_csr(M, h->stream));
        dA.clear();
        dR.clear();
        std::vector<omg_csr> vA, vR;
        for (auto &M : hA) vA.push_back(view(M));
        for (auto &M : hR) vR.push_back(view(M));
        h.reset();
        return create<V>(n_levels, vA.data(), vR.data(), smoother, omega);
    }
    // the coarsest operator: to the host for its factorisation (helper thread, as in create())
    int device = 0;
    OMG_HIP(hipGetDevice(&device));
    HostCsr Ac = download_csr(dA.back(), h->stream);
    {
        Lv &L = h->lv.back();
        L.n = Ac.n_rows;
        L.ord.identity = true;
        L.ord.sets = {0, L.n};
    }
    check_diagonal(view(Ac), n_levels - 1);
    int inv_code = OMG_OK;
    std::string inv_msg;
    std::thread inverter([&] {
        hipStream_t s = nullptr;
        SetupTimer tm("coarse factorisation (helper thread)");
        try {
            OMG_HIP(hipSetDevice(device));
            OMG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
            Lv &L = h->lv.back();
            L.A.upload(Ac, L.ord.sets, s);
            h->coarse.build(Ac, s);
        } catch (const Error &e) {
            inv_code = e.code;
            inv_msg = e.what();
        } catch (const std::exception &e) {
            inv_code = OMG_ERR_HIP;
            inv_msg = e.what();
        }
        if (s) { (void)hipStreamSynchronize(s); (void)hipStreamDestroy(s); }
    });
    struct Joiner { std::thread &t; ~Joiner() { if (t.joinable()) t.join(); } } joiner{inverter};
    {
        // (new coefficients later: omg_hierarchy_update_fine needs the fine pattern's row pointers, nothing else of the operators)
        bool all27 = dim == 3;
        for (int l = 0; l + 1 < n_levels; ++l) all27 = all27 && bool(h->lv[size_t(l)].s27);
        if (all27) {
            h->indptr0 = std::move(dA[0].indptr);
            h->nnz0 = dA[0].nnz;
        }
    }
    dA.clear();                                                    // (the fused paths hold what they need of the operators)
    dR.clear();
    for (int l = 0; l < n_levels; ++l) {
        Lv &L = h->lv[size_t(l)];
        if (!L.ord.identity) {
            L.perm.alloc(L.n);
            if (L.ord.closed_form) {
                // slot -> natural row, for the gathers at the host boundary: written by a kernel
                fill_ordering_device(L.ord.closed_form, L.ord.cf_nx, L.ord.cf_ny, L.ord.cf_nz, L.perm.p, nullptr, h->stream);
            } else {
                L.perm.upload(L.ord.perm.data(), L.n, h->stream);         // (a level the host ordered)
                OMG_HIP(hipStreamSynchronize(h->stream));
            }
        }
        if (l + 1 < n_levels) {
            Ordering &co = h->lv[size_t(l) + 1].ord;
            if (!co.identity) {
                L.r_out.alloc(size_t(h->lv[size_t(l) + 1].n));
                if (co.closed_form) {
                    fill_ordering_device(co.closed_form, co.cf_nx, co.cf_ny, co.cf_nz, nullptr, L.r_out.p, h->stream);
                } else {
                    L.r_out.upload(co.inv.data(), co.inv.size(), h->stream);
                    OMG_HIP(hipStreamSynchronize(h->stream));
                }
            }
            if (L.plane && !pooled_vectors(L)) L.tmp.alloc(L.n, vector_stagger(1));
            if (L.s27 && !pooled_vectors(L)) L.tmp.alloc(L.n);
            if (L.var7 && !L.tmp.p) L.tmp.alloc(L.n);
            if (host_level[size_t(l)]) {
                // (the restriction's rows land in the coarse level's ordering: a closed-form one is written out for that)
                materialise_ordering(co);
                build_format(h.get(), l, view(hostA[size_t(l)]), view(hostR[size_t(l)]));
            } else {
                L.format_pending = true;
            }
        }
        if (!L.x.p) {
            L.x.alloc(std::max<int64_t>(L.n, 1));
            L.b.alloc(std::max<int64_t>(L.n, 1), L.plane ? vector_stagger(2) : 0);
        }
        // (finite from the start: the 27-point sweeps multiply the slot of a neighbour that does not exist — a zero
        // coefficient — with whatever the vectors hold at the index the slot's shift lands on)
        L.x.zero(h->stream);
        L.tmp.zero(h->stream);
        L.xp = L.x.p;
        L.tp = L.tmp.p;
        if (SetupTimer::on() && l == 0) fprintf(stderr, "[omg setup] level 0 vectors at x %p  tmp %p  b %p\n", (void *)L.x.p, (void *)L.tmp.p, (void *)L.b.p);
    }
    OMG_HIP(hipStreamSynchronize(h->stream));
    for (int l = 0; l + 1 < n_levels; ++l) {
        if (!h->lv[size_t(l)].plane || !h->lv[size_t(l)].tmp.p) continue;
        SetupTimer tm("plane tiling of a large level");
        Level<V> &L = h->lv[size_t(l)], &C = h->lv[size_t(l) + 1];
        L.x.zero(h->stream); L.tmp.zero(h->stream); L.b.zero(h->stream); C.x.zero(h->stream);
        typename PlanePlan<V>::Coarse c;
        c.map = L.r_out.p;
        c.b = C.b.p;
        c.e = C.xp;
        L.plane->tune(L.xp, L.tp, L.b.p, c, h->stream, l == 0);
        OMG_HIP(hipStreamSynchronize(h->stream));
    }
    // (first: the coarse factorisation's kernels on their own stream beside the timed passes would make a candidate look slow
    // and the decision depend on the overlap)
    { SetupTimer tm("wait for the coarse factorisation"); inverter.join(); }
    place_finest_pool(h.get());
    place_s27_tiles(h.get());
    if (inv_code != OMG_OK) throw Error(inv_code, inv_msg);
    OMG_HIP(hipStreamSynchronize(h->stream));
    return h;
}

// New coefficients for the fine operator of a hierarchy that create_from_fine set up with 27-point levels throughout (same
// pattern, new values: BASELINE configs[4]'s "Galerkin RAP rebuilt on-device"; openmg/operators.py:144-188 is what the
// reference would run again): level 0 is re-tiled and every Galerkin product re-formed in HBM by s27_rap_kernel — each
// operator of the chain read once —, the coarsest operator factorised anew.  vals: the CSR's data array, host or device.
template <typename V>
void update_fine(Hier<V> *h, const double *vals, int64_t nnz, bool on_device) {
    OMG_REQUIRE(h->indptr0.p, "omg_hierarchy_update_fine: needs a hierarchy made by omg_hierarchy_create_from_fine whose smoothed levels all run the 27-point kernels");
    OMG_REQUIRE(vals && nnz == h->nnz0, "omg_hierarchy_update_fine: the value array must have the fine operator's number of entries");
    const int last = (int)h->lv.size() - 1;
    OMG_HIP(hipStreamSynchronize(h->stream));
    drop_graph(h);
    DevBuf<double> up;
    if (!on_device) {
        SetupTimer tm("update: upload the new values");
        up.alloc(size_t(nnz));
        OMG_HIP(hipMemcpyAsync(up.p, vals, size_t(nnz) * sizeof(double), hipMemcpyHostToDevice, h->stream));
        vals = up.p;
    }
    if (h->dense27.size() != h->lv.size()) h->dense27.resize(h->lv.size());
    for (int l = 0; l < last; ++l) {
        SetupTimer tm("update: tiles + Galerkin product of a level (one pass)");
        Level<V> &L = h->lv[size_t(l)];
        DevBuf<double> &out = h->dense27[size_t(l) + 1];
        const size_t need = size_t(h->lv[size_t(l) + 1].n) * 27;
        if (out.n < need) out.alloc(need);
        Stencil27Plan<V> *C = l + 1 < last ? h->lv[size_t(l) + 1].s27.get() : nullptr;
        // (a level's tiles below the finest were written as the `coarse` of the level above)
        L.s27->rap_from(l == 0 ? h->indptr0.p : nullptr, l == 0 ? vals : h->dense27[size_t(l)].p, l == 0, out.p, C, h->stream);
        L.format_pending = true;                                   // (its row-kernel side, if it was ever built, is of the old operator)
    }
    // the coarsest operator: its present, nonzero entries as CSR (what the Galerkin chain hands create_from_fine), factorised anew
    Level<V> &Lc = h->lv.back();
    const Stencil27Plan<V> &P = *h->lv[size_t(last) - 1].s27;
    const int cx = P.g.hx, cy = P.g.hy, cz = P.g.hz;
    std::vector<double> host(size_t(Lc.n) * 27);
    OMG_HIP(hipMemcpyAsync(host.data(), h->dense27[size_t(last)].p, host.size() * sizeof(double), hipMemcpyDeviceToHost, h->stream));
    OMG_HIP(hipStreamSynchronize(h->stream));
    HostCsr Ac;
    Ac.n_rows = Ac.n_cols = Lc.n;
    Ac.indptr.resize(size_t(Lc.n) + 1);
    Ac.indices.reserve(host.size());
    Ac.data.reserve(host.size());
    for (int64_t r = 0; r < Lc.n; ++r) {
        Ac.indptr[size_t(r)] = int32_t(Ac.indices.size());
        const int i = int(r % cx), j = int((r / cx) % cy), k = int(r / (int64_t(cx) * cy));
        for (int sl = 0; sl < 27; ++sl) {
            const int dx = sl % 3 - 1, dy = (sl / 3) % 3 - 1, dz = sl / 9 - 1;
            const bool present = i + dx >= 0 && i + dx < cx && j + dy >= 0 && j + dy < cy && k + dz >= 0 && k + dz < cz;
            const double v = host[size_t(r) * 27 + size_t(sl)];
            if (!present || v == 0.0) continue;                       // (exact zeros are not stored: csr_matmat drops them)
            Ac.indices.push_back(int32_t(r + (int64_t(dz) * cy + dy) * cx + dx));
            Ac.data.push_back(v);
        }
    }
    Ac.indptr[size_t(Lc.n)] = int32_t(Ac.indices.size());
    Ac.nnz = int64_t(Ac.indices.size());
    check_diagonal(view(Ac), last);
    {
        SetupTimer tm("update: coarse factorisation");
        Lc.A.upload(Ac, Lc.ord.sets, h->stream);
        h->coarse.retain_workspace = true;                          // (a hierarchy that is updated once is updated again)
        h->coarse.build(Ac, h->stream);
    }
    OMG_HIP(hipStreamSynchronize(h->stream));
}

// A throw-away single operator for the standalone entry points.
struct OneShot {
    DevCsr A;
    hipStream_t s = nullptr;
    OneShot() { require_device(); OMG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking)); }
    ~OneShot() { if (s) (void)hipStreamDestroy(s); }
};

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

// Runs f(Hier<V> *) on whichever instantiation the handle holds.
template <typename F>
void with(omg_hierarchy *h, F &&f) {
    OMG_REQUIRE(h != nullptr && (h->d || h->f), "null hierarchy");
    if (h->f) f(h->f.get());
    else f(h->d.get());
}

template <typename F>
void with(const omg_hierarchy *h, F &&f) {
    OMG_REQUIRE(h != nullptr && (h->d || h->f), "null hierarchy");
    if (h->f) f(static_cast<const Hier<float> *>(h->f.get()));
    else f(static_cast<const Hier<double> *>(h->d.get()));
}

template <typename HP>
using value_of = typename std::remove_cv<typename std::remove_pointer<HP>::type>::type::value_type;

}  // namespace
}  // namespace omg

using namespace omg;

extern "C" {

const char *omg_last_error(void) { return g_last_error.c_str(); }
const char *omg_version(void) { return "openmg_hip 0.2 (gfx950; f64, f32)"; }

int omg_device_count(int *count) {
    return guarded([&] {
        OMG_REQUIRE(count, "count is null");
        int n = 0;
        hipError_t e = hipGetDeviceCount(&n);
        *count = (e == hipSuccess) ? n : 0;
    });
}

int omg_set_device(int device) {
    return guarded([&] { require_device(); OMG_HIP(hipSetDevice(device)); });
}

int omg_device_mem_info(int64_t *free_bytes, int64_t *total_bytes) {
    return guarded([&] {
        OMG_REQUIRE(free_bytes && total_bytes, "null argument");
        require_device();
        size_t f = 0, t = 0;
        OMG_HIP(hipMemGetInfo(&f, &t));
        *free_bytes = int64_t(f);
        *total_bytes = int64_t(t);
    });
}

int omg_device_synchronize(void) {
    return guarded([&] { require_device(); OMG_HIP(hipDeviceSynchronize()); });
}

int omg_hierarchy_create_ex(int n_levels, const omg_csr *A, const omg_csr *R, int smoother,
                            double omega, int dtype, omg_hierarchy **out) {
    return guarded([&] {
        OMG_REQUIRE(out, "out is null");
        *out = nullptr;
        OMG_REQUIRE(dtype == OMG_DTYPE_F64 || dtype == OMG_DTYPE_F32, "unknown dtype");
        std::unique_ptr<omg_hierarchy> h(new omg_hierarchy);
        if (dtype == OMG_DTYPE_F32) h->f = create<float>(n_levels, A, R, smoother, omega);
        else h->d = create<double>(n_levels, A, R, smoother, omega);
        *out = h.release();
    });
}

int omg_hierarchy_create(int n_levels, const omg_csr *A, const omg_csr *R, int smoother,
                         double omega, omg_hierarchy **out) {
    return omg_hierarchy_create_ex(n_levels, A, R, smoother, omega, OMG_DTYPE_F64, out);
}

int omg_hierarchy_create_from_fine(const omg_csr *A_in, int dim, const int64_t *shape, int n_restrictions, int smoother, double omega,
                                   int dtype, omg_hierarchy **out) {
    return guarded([&] {
        OMG_REQUIRE(out && A_in, "null argument");
        *out = nullptr;
        OMG_REQUIRE(dtype == OMG_DTYPE_F64 || dtype == OMG_DTYPE_F32, "unknown dtype");
        std::unique_ptr<omg_hierarchy> h(new omg_hierarchy);
        if (dtype == OMG_DTYPE_F32) h->f = create_from_fine<float>(*A_in, dim, shape, n_restrictions, smoother, omega);
        else h->d = create_from_fine<double>(*A_in, dim, shape, n_restrictions, smoother, omega);
        *out = h.release();
    });
}

int omg_hierarchy_update_fine(omg_hierarchy *h, const double *data, int64_t nnz, int on_device) {
    return guarded([&] {
        with(h, [&](auto *hh) { update_fine(hh, data, nnz, on_device != 0); });
    });
}

int omg_hierarchy_dtype(const omg_hierarchy *h, int *dtype) {
    return guarded([&] {
        OMG_REQUIRE(h && dtype, "null argument");
        *dtype = h->f ? OMG_DTYPE_F32 : OMG_DTYPE_F64;
    });
}

int omg_hierarchy_destroy(omg_hierarchy *h) {
    return guarded([&] {
        if (!h) return;
        if (h->d || h->f) with(h, [&](auto *hh) { (void)hipStreamSynchronize(hh->stream); });
        delete h;
    });
}

int omg_hierarchy_set_stream(omg_hierarchy *h, void *hip_stream) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            OMG_HIP(hipStreamSynchronize(hh->stream));
            drop_graph(hh);
            hh->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : hh->own;
        });
    });
}

int omg_hierarchy_sync(omg_hierarchy *h) {
    return guarded([&] { with(h, [&](auto *hh) { OMG_HIP(hipStreamSynchronize(hh->stream)); }); });
}

int omg_hierarchy_level_rows(const omg_hierarchy *h, int level, int64_t *n_rows) {
    return guarded([&] {
        with(h, [&](auto *hh) { check_level(hh, level); OMG_REQUIRE(n_rows, "null"); *n_rows = hh->lv[level].n; });
    });
}

int omg_hierarchy_level_sets(const omg_hierarchy *h, int level, int64_t *n_sets) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, level);
            OMG_REQUIRE(n_sets, "null");
            *n_sets = (int64_t)hh->lv[level].ord.sets.size() - 1;      // (of the ordering: the format may not be built yet)
        });
    });
}

int omg_hierarchy_level_fused(const omg_hierarchy *h, int level, int *fused) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, level);
            OMG_REQUIRE(fused, "null");
            ensure_format(hh, level);
            *fused = (level + 1 < (int)hh->lv.size() && can_fuse(hh, hh->lv[level])) ? 1 : 0;
        });
    });
}

int omg_hierarchy_level_flags(const omg_hierarchy *h, int level, int *flags) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, level);
            OMG_REQUIRE(flags, "null");
            ensure_format(hh, level);
            const bool smoothed = level + 1 < (int)hh->lv.size();
            *flags = ((smoothed && can_fuse(hh, hh->lv[level])) ? OMG_LEVEL_FUSED_LAST_SET : 0) |
                     ((smoothed && hh->lv[level].scatter_prolong) ? OMG_LEVEL_SCATTER_PROLONG : 0) |
                     ((smoothed && hh->lv[level].A.all_union()) ? OMG_LEVEL_UNION_WALK : 0) |
                     ((smoothed && hh->lv[level].march) ? OMG_LEVEL_MARCH : 0) |
                     ((smoothed && hh->lv[level].march && hh->lv[level].march->line_scan) ? OMG_LEVEL_MARCH_SCAN : 0) |
                     ((smoothed && hh->lv[level].plane && !hh->no_plane) ? OMG_LEVEL_PLANE : 0) |
                     ((smoothed && hh->lv[level].s27 && !hh->no_plane) ? OMG_LEVEL_STENCIL27 : 0) |
                     ((smoothed && hh->lv[level].var7 && !hh->no_plane) ? OMG_LEVEL_VAR7 : 0);
        });
    });
}

int omg_hierarchy_use_plane(omg_hierarchy *h, int enable) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            OMG_HIP(hipStreamSynchronize(hh->stream));
            drop_graph(hh);
            hh->no_plane = enable == 0;
        });
    });
}

int omg_hierarchy_plane_info(const omg_hierarchy *h, int level, int64_t *out8) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, level);
            OMG_REQUIRE(out8, "null");
            for (int i = 0; i < 8; ++i) out8[i] = 0;
            const auto &L = hh->lv[level];
            if (!L.plane) return;
            const PlaneGeom &g = L.plane->g;
            const int64_t v[8] = {g.nx, g.ny, g.nz, g.TX, g.TY, g.LZ, g.n_wg, g.threads};
            for (int i = 0; i < 8; ++i) out8[i] = v[i];
        });
    });
}

int omg_hierarchy_set_info(const omg_hierarchy *h, int level, int set, int64_t *rows, int64_t *nnz) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, level);
            ensure_format(hh, level);
            const auto &A = hh->lv[level].A;
            OMG_REQUIRE(set >= 0 && size_t(set) < A.n_sets() && rows && nnz, "set out of range / null");
            *rows = A.sets[set + 1] - A.sets[set];
            *nnz = A.set_nnz[set];
        });
    });
}

int omg_hierarchy_format_info(const omg_hierarchy *h, int level, int op, int set, int64_t *out) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, level);
            OMG_REQUIRE(out && op >= 0 && op <= 2, "null / unknown operator");
            OMG_REQUIRE(op == 0 || level + 1 < (int)hh->lv.size(), "the coarsest level has no restriction");
            ensure_format(hh, level);
            const auto &L = hh->lv[level];
            (op == 0 ? L.A : op == 1 ? L.R : L.P).format_info(set, out);
        });
    });
}

int omg_format_selftest(const omg_csr *A, int dtype, int64_t *out) {
    return guarded([&] {
        OMG_REQUIRE(A && out, "null argument");
        OMG_REQUIRE(dtype == OMG_DTYPE_F64 || dtype == OMG_DTYPE_F32, "unknown dtype");
        validate_csr(*A, "A");
        if (dtype == OMG_DTYPE_F32) format_selftest<float>(*A, out);
        else format_selftest<double>(*A, out);
    });
}

int omg_vcycle(omg_hierarchy *h, int level, const double *b, double *x, int pre, int post,
               double *norm) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(b && x, "b / x is null");
            OMG_REQUIRE(pre >= 0 && post >= 0, "negative sweep count");
            auto &L = hh->lv[level];
            hh->resident = false;
            load_vec(hh, level, b, L.b.p);
            load_vec(hh, level, x, L.xp);
            const int last = (int)hh->lv.size() - 1;
            const int part = cycle_body(hh, level, pre, post, level < last);
            double nv = 0.0;
            if (level < last) {
                finish_norm(hh, level, part);
                nv = read_norm(hh);
            }
            fetch_vec<V>(hh, level, L.xp, x);
            if (norm) *norm = nv;
        });
    });
}

int omg_vcycle_ex(omg_hierarchy *h, int level, const double *b, const double *x_in, double *x_out, double *x_pre,
                  int pre, int post, double *norm) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(b && x_out, "b / x_out is null");
            OMG_REQUIRE(pre >= 0 && post >= 0, "negative sweep count");
            auto &L = hh->lv[level];
            const int last = (int)hh->lv.size() - 1;
            hh->resident = false;
            load_vec(hh, level, b, L.b.p);
            if (x_in) load_vec(hh, level, x_in, L.xp);
            else OMG_HIP(hipMemsetAsync(L.xp, 0, size_t(L.n) * sizeof(V), hh->stream));
            const bool want_pre = x_pre && level < last && pre > 0;
            if (want_pre) {
                if (hh->pre_buf.n < size_t(L.n)) hh->pre_buf.alloc(size_t(L.n));
                hh->pre_level = level;
            }
            int part = NORM_NONE;
            try {
                part = cycle_body(hh, level, pre, post, level < last);
            } catch (...) {
                hh->pre_level = -1;
                throw;
            }
            hh->pre_level = -1;
            double nv = 0.0;
            if (level < last) {
                finish_norm(hh, level, part);
                nv = read_norm(hh);
            }
            fetch_vec<V>(hh, level, L.xp, x_out);
            if (want_pre) fetch_vec<V>(hh, level, hh->pre_buf.p, x_pre);
            else if (x_pre && x_in) { if (x_pre != x_in) std::memcpy(x_pre, x_in, size_t(L.n) * sizeof(double)); }
            else if (x_pre) std::memset(x_pre, 0, size_t(L.n) * sizeof(double));
            if (norm) *norm = nv;
        });
    });
}

// omg_vcycle_ex for a caller whose b / initial / uOut are DEVICE arrays (double, natural numbering): openmg.mgCycle
// (openmg/__init__.py:151-236) chained on the GPU — nothing but the norm (8 bytes) crosses PCIe.  x_in_dev NULL: zeros;
// x_pre_dev (may be x_in_dev): the iterate after the pre-smoothing (Q2), the input's copy / zeros where none ran.
int omg_vcycle_dev(omg_hierarchy *h, int level, const double *b_dev, const double *x_in_dev, double *x_out_dev, double *x_pre_dev,
                   int pre, int post, double *norm) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(b_dev && x_out_dev, "b / x_out is null");
            OMG_REQUIRE(pre >= 0 && post >= 0, "negative sweep count");
            auto &L = hh->lv[level];
            const int last = (int)hh->lv.size() - 1;
            hh->resident = false;
            load_vec_dev(hh, level, b_dev, L.b.p);
            if (x_in_dev) load_vec_dev(hh, level, x_in_dev, L.xp);
            else OMG_HIP(hipMemsetAsync(L.xp, 0, size_t(L.n) * sizeof(V), hh->stream));
            const bool want_pre = x_pre_dev && level < last && pre > 0;
            if (want_pre) {
                if (hh->pre_buf.n < size_t(L.n)) hh->pre_buf.alloc(size_t(L.n));
                hh->pre_level = level;
            }
            int part = NORM_NONE;
            try {
                part = cycle_body(hh, level, pre, post, level < last);
            } catch (...) {
                hh->pre_level = -1;
                throw;
            }
            hh->pre_level = -1;
            if (level < last) finish_norm(hh, level, part);
            fetch_vec_dev<V>(hh, level, L.xp, x_out_dev);
            if (want_pre) fetch_vec_dev<V>(hh, level, hh->pre_buf.p, x_pre_dev);
            else if (x_pre_dev && x_in_dev) { if (x_pre_dev != x_in_dev) OMG_HIP(hipMemcpyAsync(x_pre_dev, x_in_dev, size_t(L.n) * sizeof(double), hipMemcpyDeviceToDevice, hh->stream)); }
            else if (x_pre_dev) OMG_HIP(hipMemsetAsync(x_pre_dev, 0, size_t(L.n) * sizeof(double), hh->stream));
            const double nv = level < last ? read_norm(hh) : 0.0;     // (synchronises the stream: the outputs are complete on return)
            if (level >= last) { OMG_HIP(hipStreamSynchronize(hh->stream)); check_march(hh); }
            if (norm) *norm = nv;
        });
    });
}

// omg_resident_load / omg_resident_fetch with device arrays (double, natural numbering): mgSolve for a caller on the GPU
int omg_resident_load_dev(omg_hierarchy *h, const double *b_dev, const double *x0_dev) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, 0);
            OMG_REQUIRE(b_dev, "b is null");
            auto &L = hh->lv[0];
            load_vec_dev(hh, 0, b_dev, L.b.p);
            if (x0_dev) load_vec_dev(hh, 0, x0_dev, L.xp);
            else OMG_HIP(hipMemsetAsync(L.xp, 0, L.n * sizeof(V), hh->stream));
            OMG_HIP(hipStreamSynchronize(hh->stream));
            hh->resident = true;
        });
    });
}

int omg_resident_fetch_dev(omg_hierarchy *h, double *x_dev) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, 0);
            OMG_REQUIRE(hh->resident && x_dev, "nothing resident / x is null");
            fetch_vec_dev<V>(hh, 0, hh->lv[0].xp, x_dev);
            OMG_HIP(hipStreamSynchronize(hh->stream));
            check_march(hh);
        });
    });
}

// Device-pointer V-cycle for callers that already live on the GPU (the multi-GPU runner uses
// a replicated hierarchy as its coarse solver): b_dev, x_dev are level-0 DOUBLE vectors in
// natural numbering; x starts from zero (openmg/__init__.py:191-192); enqueued on
// `hip_stream`, no host synchronisation.
int omg_hierarchy_cycle_dev(omg_hierarchy *h, const double *b_dev, double *x_dev, int pre, int post,
                            void *hip_stream) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, 0);
            OMG_REQUIRE(b_dev && x_dev && pre >= 0 && post >= 0, "bad argument");
            auto &L = hh->lv[0];
            hipStream_t keep = hh->stream;
            hh->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : hh->own;
            hh->resident = false;
            try {
                const int32_t *perm = L.ord.identity ? nullptr : L.perm.p;
                if (direct_io(L)) OMG_HIP(hipMemcpyAsync(L.b.p, b_dev, L.n * sizeof(double), hipMemcpyDeviceToDevice, hh->stream));
                else launch_gather<double, V>(b_dev, perm, L.b.p, L.n, hh->stream);
                bool first = false, zero_in = false;
                if (hh->lv.size() > 1) {
                    zero_in = (use_plane(hh, L, pre, post) && pre == 1) || (use_s27(hh, L) && pre >= 1) || (use_var7(hh, L) && pre == 1);   // the down pass / first sweep does not read a zero iterate
                    if (!zero_in) {
                        // x starts from zero: the first relaxation launch is a pointwise b / diag (restrict_level).  The
                        // diagonal comes from the row-kernel format, which a plane level builds on first use: BEFORE it
                        // (with the format still pending the diagonal stayed uninitialised)
                        if (pre > 0) {
                            ensure_format(hh, 0);
                            if (!L.diag.p && !getenv_flag("OMG_NO_FIRST_SWEEP")) {
                                L.diag.alloc(std::max<int64_t>(L.n, 1));
                                launch_diagonal(L.A, L.diag.p, hh->stream);
                            }
                        }
                        first = pre > 0 && first_sweep_in_restrict(hh, L, pre);
                        if (first) {
                            const bool jac = hh->smoother == OMG_SMOOTH_JACOBI;
                            launch_first_relaxation<V>(L.b.p, L.diag.p, L.xp, L.n, jac ? L.n : L.A.sets[1], jac, hh->omega, hh->stream);
                        } else {
                            OMG_HIP(hipMemsetAsync(L.xp, 0, L.n * sizeof(V), hh->stream));
                        }
                    }
                }
                cycle_body(hh, 0, pre, post, false, nullptr, first, nullptr, zero_in);
                if (direct_io(L)) OMG_HIP(hipMemcpyAsync(x_dev, L.xp, L.n * sizeof(double), hipMemcpyDeviceToDevice, hh->stream));
                else launch_scatter<V, double>(L.xp, perm, x_dev, L.n, hh->stream);
            } catch (...) {
                hh->stream = keep;
                throw;
            }
            hh->stream = keep;
        });
    });
}

int omg_resident_load(omg_hierarchy *h, const double *b, const double *x0) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, 0);
            OMG_REQUIRE(b, "b is null");
            auto &L = hh->lv[0];
            load_vec(hh, 0, b, L.b.p);
            if (x0) load_vec(hh, 0, x0, L.xp);
            else OMG_HIP(hipMemsetAsync(L.xp, 0, L.n * sizeof(V), hh->stream));
            OMG_HIP(hipStreamSynchronize(hh->stream));
            hh->resident = true;
        });
    });
}

int omg_resident_cycle(omg_hierarchy *h, int pre, int post, double *norm) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, 0);
            OMG_REQUIRE(hh->resident, "omg_resident_load has not been called");
            OMG_REQUIRE(pre >= 0 && post >= 0, "negative sweep count");
            run_cycle0(hh, pre, post);
            if (norm) *norm = read_norm(hh);
        });
    });
}

// n cycles back to back, EVERY cycle's residual norm computed and returned (norms[n], host).  Where
// can_prenorm() holds, the norm of cycle k is finished inside cycle k + 1's first launch instead of
// by a launch of its own (same partial sums, same order: the same bits as n omg_resident_cycle
// calls); the last cycle's by the usual norm launch.  One host synchronisation at the end.
int omg_resident_cycles(omg_hierarchy *h, int pre, int post, int n_cycles, double *norms) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            check_level(hh, 0);
            OMG_REQUIRE(hh->resident, "omg_resident_load has not been called");
            OMG_REQUIRE(pre >= 0 && post >= 0 && n_cycles >= 0, "negative argument");
            if (n_cycles == 0) return;
            if (hh->norms_dev.n < size_t(n_cycles)) hh->norms_dev.alloc(size_t(n_cycles));
            const bool single = hh->lv.size() == 1;
            if (!single && use_plane(hh, hh->lv[0], pre, post) && post <= 1) {
                // every cycle's up pass leaves its workgroup partials in a slot of the batch buffer; one
                // launch per chunk adds the slots up (the additions of launch_sum_sqrt: same bits as
                // omg_resident_cycle)
                constexpr int CHUNK = 64;
                const int64_t nb = hh->lv[0].plane->g.n_wg;
                if (hh->batch_partials.n < size_t(CHUNK) * size_t(nb)) hh->batch_partials.alloc(size_t(CHUNK) * size_t(nb));
                for (int k0 = 0; k0 < n_cycles; k0 += CHUNK) {
                    const int cnt = std::min(CHUNK, n_cycles - k0);
                    for (int j = 0; j < cnt; ++j) {
                        const int st = cycle_body(hh, 0, pre, post, true, nullptr, false, hh->batch_partials.p + size_t(j) * size_t(nb));
                        OMG_REQUIRE(st == NORM_PLANE, "internal: plane-pipelined cycle without its norm partials");
                    }
                    launch_sum_batch(hh->batch_partials.p, nb, nb, cnt, hh->norms_dev.p + k0, true, hh->stream);
                }
                if (norms) OMG_HIP(hipMemcpyAsync(norms, hh->norms_dev.p, size_t(n_cycles) * sizeof(double), hipMemcpyDeviceToHost, hh->stream));
                OMG_HIP(hipStreamSynchronize(hh->stream));
                check_march(hh);
                return;
            }
            if (!single && use_s27(hh, hh->lv[0])) {
                // 27-point level: with pre >= 1 the first sweep of cycle j + 1 also squares the residuals with respect
                // to the iterate it starts from — cycle j's norm (openmg/__init__.py:227) without another pass over
                // the operator; the last cycle of a chunk (and every cycle when pre = 0) runs the norm kernel
                constexpr int CHUNK = 64;
                using V = value_of<decltype(hh)>;
                auto &L0 = hh->lv[0];
                Stencil27Plan<V> &P = *L0.s27;
                const int64_t nb = int64_t(4) * P.g.n_wg;
                if (hh->batch_partials.n < size_t(CHUNK) * size_t(nb)) hh->batch_partials.alloc(size_t(CHUNK) * size_t(nb));
                for (int k0 = 0; k0 < n_cycles; k0 += CHUNK) {
                    const int cnt = std::min(CHUNK, n_cycles - k0);
                    for (int j = 0; j < cnt; ++j) {
                        double *slot_prev = (pre >= 1 && j > 0) ? hh->batch_partials.p + size_t(j - 1) * size_t(nb) : nullptr;
                        double *slot_this = hh->batch_partials.p + size_t(j) * size_t(nb);
                        const bool own_norm = j + 1 == cnt || pre == 0;
                        cycle_body(hh, 0, pre, post, own_norm, slot_prev, false, own_norm ? slot_this : nullptr);
                        if (own_norm) {
                            Prof<V> p(hh, 0, 4);
                            P.norm(L0.xp, L0.b.p, P.have67, slot_this, hh->stream);
                        }
                    }
                    launch_sum_batch(hh->batch_partials.p, nb, nb, cnt, hh->norms_dev.p + k0, true, hh->stream);
                }
                if (norms) OMG_HIP(hipMemcpyAsync(norms, hh->norms_dev.p, size_t(n_cycles) * sizeof(double), hipMemcpyDeviceToHost, hh->stream));
                OMG_HIP(hipStreamSynchronize(hh->stream));
                check_march(hh);
                return;
            }
            const bool v7 = !single && use_var7(hh, hh->lv[0]);       // (its cycle never touches the row-kernel format)
            if (!single && !v7) ensure_format(hh, 0);
            // (a plane-pipelined cycle with post > 1 ends set by set: its first launch is no PRENORM launch)
            const bool defer = !single && !v7 && !getenv_flag("OMG_NO_PRENORM") && can_prenorm(hh, hh->lv[0], pre, post) &&
                               !use_plane(hh, hh->lv[0], pre, post);
            // Deferred norms: cycle k's block partials are collected in slot k of a batch buffer — the
            // last set's by its fused post-smoothing launch, the first set's by cycle k + 1's first
            // launch — and ALL slots of a chunk are added up by one launch at the chunk's end (the
            // additions of launch_sum_sqrt, same bits).  The last cycle of a chunk has no successor
            // in it: the usual norm launch.
            constexpr int CHUNK = 64;
            const int64_t nb = (single || v7) ? 0 : hh->lv[0].A.n_blocks();
            if (defer && hh->batch_partials.n < size_t(CHUNK) * size_t(nb)) hh->batch_partials.alloc(size_t(CHUNK) * size_t(nb));
            for (int k0 = 0; k0 < n_cycles; k0 += CHUNK) {
                const int cnt = std::min(CHUNK, n_cycles - k0);
                for (int j = 0; j < cnt; ++j) {
                    const int k = k0 + j;
                    if (single) {
                        cycle_body(hh, 0, pre, post);
                        OMG_HIP(hipMemsetAsync(hh->norms_dev.p + k, 0, sizeof(double), hh->stream));   // :232
                        continue;
                    }
                    const bool last_of_chunk = j + 1 == cnt;
                    double *slot_prev = (defer && j > 0) ? hh->batch_partials.p + size_t(j - 1) * size_t(nb) : nullptr;
                    double *slot_this = (defer && !last_of_chunk) ? hh->batch_partials.p + size_t(j) * size_t(nb) : nullptr;
                    const int part = cycle_body(hh, 0, pre, post, true, slot_prev, false, slot_this);
                    if (!defer || last_of_chunk) {
                        finish_norm(hh, 0, part, hh->norms_dev.p + k);                                 // :227
                    } else {
                        OMG_REQUIRE(part || hh->smoother == OMG_SMOOTH_JACOBI, "internal: deferred norm without the fused post-smoothing launch");
                    }
                }
                if (defer && cnt > 1)
                    launch_sum_batch(hh->batch_partials.p, nb, nb, cnt - 1, hh->norms_dev.p + k0, true, hh->stream);
            }
            if (norms) OMG_HIP(hipMemcpyAsync(norms, hh->norms_dev.p, size_t(n_cycles) * sizeof(double), hipMemcpyDeviceToHost, hh->stream));
            OMG_HIP(hipStreamSynchronize(hh->stream));
            check_march(hh);       // (the batched entry returned norms of a sweep that had given up)
        });
    });
}

// Plain fine-grid SpMV y = A_0 x over the WHOLE operator (x: the resident iterate, y: the
// level's residual buffer, which every cycle overwrites anyway), `reps` launches timed as one
// hipEvent bracket on the hierarchy's stream: the "fine-grid SpMV GB/s" of BASELINE.json's
// metric, measured on the operator as it sits in HBM inside the V-cycle.
int omg_resident_spmv_time(omg_hierarchy *h, int reps, double *avg_ms) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, 0);
            OMG_REQUIRE(hh->resident && reps > 0 && avg_ms, "nothing resident / bad argument");
            OMG_REQUIRE(hh->lv.size() > 1, "single-level hierarchy has no smoothed operator");
            auto &L = hh->lv[0];
            // a plane level (red-black ordered constant-coefficient stencil): matrix-free, like its cycle; y = the level's
            // scratch vector.  Anything else: the row kernels on the operator as it sits in HBM.
            const bool mf = use_plane(hh, L, 1, 1) && !L.plane->g.jacobi;
            if (!mf) ensure_format(hh, 0);
            RowArgsT<V> a;
            a.x = L.xp; a.y = mf ? L.tp : L.r.p;
            // (OMG_SPMV_Y=b / own, read at every call — placement experiments: the product written over the right-hand side, or
            // into an allocation of its own, instead of the level's scratch vector)
            if (mf) ensure_spmv_y(hh, L);
            V *ymf = L.spmv_y.p;
            DevBuf<V> own_y;
            if (const char *e = getenv("OMG_SPMV_Y")) {
                if (e[0] == 'b') ymf = L.b.p;
                else if (e[0] == 't') ymf = L.tp;
                else if (e[0] == 'o') { own_y.alloc(size_t(L.n), size_t(atoi(e + 1)) * 64); ymf = own_y.p; }
            }
            auto once = [&] {
                if (mf) L.plane->spmv(L.xp, ymf, hh->stream);
                else launch_rows(L.A, ROW_SPMV, -1, a, hh->stream);
            };
            once();                                                     // warm-up
            hipEvent_t e0, e1;
            OMG_HIP(hipEventCreate(&e0));
            OMG_HIP(hipEventCreate(&e1));
            OMG_HIP(hipEventRecord(e0, hh->stream));
            for (int i = 0; i < reps; ++i) once();
            OMG_HIP(hipEventRecord(e1, hh->stream));
            OMG_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            OMG_HIP(hipEventElapsedTime(&ms, e0, e1));
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            *avg_ms = double(ms) / reps;
        });
    });
}

int omg_resident_fetch(omg_hierarchy *h, double *x) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, 0);
            OMG_REQUIRE(hh->resident && x, "nothing resident / x is null");
            fetch_vec<V>(hh, 0, hh->lv[0].xp, x);
        });
    });
}

int omg_resident_use_graph(omg_hierarchy *h, int enable) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            OMG_HIP(hipStreamSynchronize(hh->stream));
            hh->want_graph = enable != 0;
            if (!enable) drop_graph(hh);
        });
    });
}

int omg_solve(omg_hierarchy *h, const double *b, double *x, int pre, int post, int max_cycles,
              double threshold, int *cycles_done, double *norm) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, 0);
            OMG_REQUIRE(b && x, "b / x is null");
            OMG_REQUIRE(max_cycles > 0 || threshold > 0.0,
                        "Either threshold or cycles must be > 0");   // openmg/__init__.py:118-119
            auto &L = hh->lv[0];
            load_vec(hh, 0, b, L.b.p);
            load_vec(hh, 0, x, L.xp);
            hh->resident = true;
            int cycle = 0;
            double nv = 0.0;
            for (;;) {                                                // :112, :132-138
                run_cycle0(hh, pre, post);
                nv = read_norm(hh);
                ++cycle;
                const bool by_cycles = max_cycles > 0 && cycle >= max_cycles;
                const bool by_norm = threshold > 0.0 && nv < threshold;
                if (by_cycles || by_norm) break;
            }
            fetch_vec<V>(hh, 0, L.xp, x);
            if (cycles_done) *cycles_done = cycle;
            if (norm) *norm = nv;
        });
    });
}

int omg_profile_enable(omg_hierarchy *h, int enable) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            OMG_HIP(hipStreamSynchronize(hh->stream));
            for (auto &e : hh->events) { hh->event_pool.push_back(e.a); hh->event_pool.push_back(e.b); }
            hh->events.clear();
            std::memset(hh->prof_n, 0, sizeof(hh->prof_n));
            std::memset(hh->prof_ms, 0, sizeof(hh->prof_ms));
            hh->profiling = (unsigned)enable;   // bit c = class c; -1 = all classes
        });
    });
}

int omg_profile_read(omg_hierarchy *h, int64_t *launches, double *total_ms) {
    return guarded([&] {
        OMG_REQUIRE(launches && total_ms, "null argument");
        with(h, [&](auto *hh) {
            OMG_HIP(hipStreamSynchronize(hh->stream));
            for (auto &e : hh->events) {
                float ms = 0.f;
                OMG_HIP(hipEventElapsedTime(&ms, e.a, e.b));
                hh->prof_n[e.cls] += 1;
                hh->prof_ms[e.cls] += ms;
                hh->event_pool.push_back(e.a);
                hh->event_pool.push_back(e.b);
            }
            hh->events.clear();
            for (int c = 0; c < OMG_PROFILE_CLASSES; ++c) { launches[c] = hh->prof_n[c]; total_ms[c] = hh->prof_ms[c]; }
        });
    });
}

// ---- single level operations ------------------------------------------------------------
int omg_level_smooth(omg_hierarchy *h, int level, const double *b, double *x, int iterations) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(level < (int)hh->lv.size() - 1, "the coarsest level has no smoother");
            OMG_REQUIRE(b && x && iterations >= 0, "bad argument");
            auto &L = hh->lv[level];
            hh->resident = false;
            load_vec(hh, level, b, L.b.p);
            load_vec(hh, level, x, L.xp);
            smooth_level(hh, level, iterations);
            fetch_vec<V>(hh, level, L.xp, x);
        });
    });
}

int omg_level_residual(omg_hierarchy *h, int level, const double *b, const double *x, double *r,
                       double *norm) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(level < (int)hh->lv.size() - 1, "use omg_residual for the coarsest operator");
            OMG_REQUIRE(b && x && r, "null vector");
            auto &L = hh->lv[level];
            hh->resident = false;
            load_vec(hh, level, b, L.b.p);
            load_vec(hh, level, x, L.xp);
            if (norm) {
                norm_level<V>(hh, level, L.r.p);
                *norm = read_norm(hh);
            } else {
                residual_level<V>(hh, level, L.r.p);
            }
            fetch_vec<V>(hh, level, L.r.p, r);
        });
    });
}

int omg_level_spmv(omg_hierarchy *h, int level, const double *x, double *y) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(level < (int)hh->lv.size() - 1, "use omg_spmv for the coarsest operator");
            OMG_REQUIRE(x && y, "null vector");
            auto &L = hh->lv[level];
            hh->resident = false;
            load_vec(hh, level, x, L.xp);
            const bool mf = use_plane(hh, L, 1, 1) && !L.plane->g.jacobi;
            if (mf) {
                ensure_spmv_y(hh, L);
                L.plane->spmv(L.xp, L.spmv_y.p, hh->stream);
                fetch_vec<V>(hh, level, L.spmv_y.p, y);
            } else {
                ensure_format(hh, level);
                RowArgsT<V> a;
                a.x = L.xp; a.y = L.r.p;
                launch_rows(L.A, ROW_SPMV, -1, a, hh->stream);
                fetch_vec<V>(hh, level, L.r.p, y);
            }
        });
    });
}

int omg_level_restrict(omg_hierarchy *h, int level, const double *fine, double *coarse) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(level < (int)hh->lv.size() - 1, "no restriction below the coarsest level");
            OMG_REQUIRE(fine && coarse, "null vector");
            auto &L = hh->lv[level];
            auto &C = hh->lv[level + 1];
            hh->resident = false;
            load_vec(hh, level, fine, L.r.p);
            restrict_level<V>(hh, level, L.r.p, C.b.p);
            fetch_vec<V>(hh, level + 1, C.b.p, coarse);
        });
    });
}

int omg_level_prolong_add(omg_hierarchy *h, int level, const double *coarse, double *fine_inout) {
    return guarded([&] {
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            check_level(hh, level);
            OMG_REQUIRE(level < (int)hh->lv.size() - 1, "no prolongation below the coarsest level");
            OMG_REQUIRE(coarse && fine_inout, "null vector");
            auto &L = hh->lv[level];
            auto &C = hh->lv[level + 1];
            hh->resident = false;
            load_vec(hh, level + 1, coarse, C.xp);
            load_vec(hh, level, fine_inout, L.xp);
            prolong_add_level<V>(hh, level, C.xp, L.xp);
            fetch_vec<V>(hh, level, L.xp, fine_inout);
        });
    });
}

int omg_coarse_solve(omg_hierarchy *h, const double *b, double *x) {
    return guarded([&] {
        OMG_REQUIRE(b && x, "null argument");
        with(h, [&](auto *hh) {
            using V = value_of<decltype(hh)>;
            const int last = (int)hh->lv.size() - 1;
            auto &L = hh->lv[last];
            hh->resident = last != 0 ? hh->resident : false;
            load_vec(hh, last, b, L.b.p);
            coarse_solve_level(hh);
            fetch_vec<V>(hh, last, L.xp, x);
        });
    });
}

int omg_hierarchy_coarse_info(const omg_hierarchy *h, int64_t *out4) {
    return guarded([&] {
        OMG_REQUIRE(out4, "null argument");
        with(h, [&](auto *hh) {
            out4[0] = hh->coarse.P;
            out4[1] = hh->coarse.n;
            out4[2] = hh->coarse.w;
            out4[3] = int64_t(hh->coarse.bytes);
        });
    });
}

// ---- standalone (double only: these mirror the reference's fp64 operators one to one) -------
static void standalone_rows(const omg_csr *A, int mode, const double *x, const double *b,
                            double *y, double *norm) {
    OMG_REQUIRE(A && x && y, "null argument");
    validate_csr(*A, "A");
    OneShot os;
    HostCsr Ah = permute_csr(*A, nullptr, nullptr);
    os.A.upload(Ah, {}, os.s);
    DevBuf<double> dx(std::max<int64_t>(A->n_cols, 1)), dy(std::max<int64_t>(A->n_rows, 1)), db, part, nrm;
    dx.upload(x, A->n_cols, os.s);
    RowArgs a;
    a.x = dx.p; a.y = dy.p;
    if (b) { db.alloc(std::max<int64_t>(A->n_rows, 1)); db.upload(b, A->n_rows, os.s); a.b = db.p; }
    if (norm) { part.alloc(os.A.n_blocks() + SUM_FOLD); nrm.alloc(1); a.partials = part.p; }
    launch_rows(os.A, mode, -1, a, os.s);
    if (norm) {
        launch_sum_sqrt(part.p, os.A.n_blocks(), nrm.p, os.s);
        nrm.download(norm, 1, os.s);
    }
    dy.download(y, A->n_rows, os.s);
    OMG_HIP(hipStreamSynchronize(os.s));
}

int omg_spmv(const omg_csr *A, const double *x, double *y) {
    return guarded([&] { standalone_rows(A, ROW_SPMV, x, nullptr, y, nullptr); });
}

int omg_residual(const omg_csr *A, const double *b, const double *x, double *r, double *norm) {
    return guarded([&] {
        OMG_REQUIRE(b, "b is null");
        OMG_REQUIRE(A && A->n_rows == A->n_cols, "residual needs a square operator");
        standalone_rows(A, norm ? ROW_RESNORM : ROW_RESIDUAL, x, b, r, norm);
    });
}

int omg_gauss_seidel(const omg_csr *A, const double *b, double *x, int smoother, double omega,
                     int iterations, double threshold, int *sweeps_done) {
    return guarded([&] {
        using H = Hier<double>;
        OMG_REQUIRE(A && b && x, "null argument");
        OMG_REQUIRE(A->n_rows == A->n_cols, "smoother needs a square operator");
        validate_csr(*A, "A");
        check_diagonal(*A, 0);
        if (iterations < 0 && threshold < 0.0) iterations = 1;      // openmg/solvers.py:39-40
        // a two-"level" shell so that the level machinery (ordering, plan) is reused
        require_device();
        std::unique_ptr<H> h(new H);
        h->smoother = smoother;
        h->omega = omega;
        OMG_HIP(hipStreamCreateWithFlags(&h->own, hipStreamNonBlocking));
        h->stream = h->own;
        h->lv.resize(2);           // level 1 is a dummy so that level 0 counts as "smoothed"
        h->norm_dev.alloc(1);
        Level<double> &L = h->lv[0];
        L.n = A->n_rows;
        order_level(L, *A, smoother, h->stream);
        const bool id = L.ord.identity;
        {
            HostCsr Ap = permute_csr(*A, id ? nullptr : L.ord.perm.data(), id ? nullptr : L.ord.inv.data());
            L.A.upload(Ap, L.ord.sets, h->stream);
        }
        if (!id) { L.perm.alloc(L.n); L.perm.upload(L.ord.perm.data(), L.n, h->stream); }
        L.x.alloc(std::max<int64_t>(L.n, 1));
        L.b.alloc(std::max<int64_t>(L.n, 1));
        if (smoother == OMG_SMOOTH_JACOBI) L.tmp.alloc(std::max<int64_t>(L.n, 1));
        L.partials.alloc(L.A.n_blocks() + SUM_FOLD);
        L.xp = L.x.p;
        L.tp = L.tmp.p;
        build_plan(L);
        load_vec(h.get(), 0, b, L.b.p);
        load_vec(h.get(), 0, x, L.xp);
        int it = 0;
        auto stop = [&]() {                                          // openmg/solvers.py:43-50
            const bool by_iter = iterations >= 0 && it >= iterations;
            bool by_norm = false;
            if (threshold >= 0.0) {
                norm_level<double>(h.get(), 0, nullptr);
                by_norm = read_norm(h.get()) < threshold;
            }
            return by_iter || by_norm;
        };
        while (!stop()) {                                            // :52-54, :72-74
            smooth_level(h.get(), 0, 1);
            ++it;
        }
        fetch_vec<double>(h.get(), 0, L.xp, x);
        if (sweeps_done) *sweeps_done = it;
    });
}

int omg_direct_solve(const omg_csr *A, const double *b, double *x) {
    return guarded([&] {
        OMG_REQUIRE(A && b && x, "null argument");
        OMG_REQUIRE(A->n_rows == A->n_cols, "direct solve needs a square operator");
        validate_csr(*A, "A");
        OneShot os;
        HostCsr Ah = permute_csr(*A, nullptr, nullptr);
        os.A.upload(Ah, {}, os.s);
        const int64_t n = A->n_rows;
        DevBuf<double> db(std::max<int64_t>(n, 1)), dx(std::max<int64_t>(n, 1));
        CoarseSolver<double> cs;
        cs.build(Ah, os.s);
        db.upload(b, n, os.s);
        cs.solve(db.p, dx.p, os.s);
        dx.download(x, n, os.s);
        OMG_HIP(hipStreamSynchronize(os.s));
    });
}

}  // extern "C"
