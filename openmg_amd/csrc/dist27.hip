// Multi-GPU V-cycle of 27-point grid stencils with per-row coefficients (BASELINE configs[4]): 1-D slabs on the kernels
// of stencil27.hip — gfx950 only.  (No reference counterpart: openmg is single-process; the cycle is
// openmg/__init__.py:151-236, the rows' arithmetic openmg/solvers.py:63-68 as stencil27.hip states it.)
//
// A rank owns nz / n_ranks planes of every distributed level (an even number: slabs are cut on aggregate boundaries, so
// the restriction, the prolongation and the Galerkin products are rank-local).  Its vectors are the octant layout of
// its EXTENDED slab: one ghost AGGREGATE plane (two grid planes) below and above the owned ones.  With colour =
// (i & 1) + 2 (j & 1) + 4 (k & 1), colours 0 .. 3 live on even planes and 4 .. 7 on odd ones, and a sweep takes the
// colours in turn; therefore
//   * the first owned plane (even) needs the OLD values of colours 4 .. 7 of the lower ghost plane;
//   * the last owned plane (odd) needs the NEW values of colours 0 .. 3 of the upper ghost plane: the sweep's first two
//     pair launches relax those rows too (the neighbour's rows — their coefficients are copied over once at setup —, the
//     neighbour's bits), which needs the old values of all colours of that plane.
// So ONE exchange per sweep (colours 4 .. 7 of the boundary aggregate planes, both ways, contiguous runs of the
// vectors: no pack kernels) keeps the invariant "lower ghost: colours 4 .. 7 current; upper ghost: all colours current",
// and nothing else of a cycle needs one except the coarse right-hand side's colours 0 .. 3 of the upper ghost plane
// (those redundantly relaxed rows read it): the residual reads what the invariant holds, the restriction is local, the
// prolongation corrects the ghost planes from the coarse level's ghost cells, which the coarse level's invariant covers.
// Exchanges per V(p, q) cycle: p + q on the finest level, 1 + p + q on every other distributed level, one all-gather
// above the replicated tail, one all-reduce per BATCH of cycles for the norms.
//
// Setup per rank, on the device: the rank's CSR rows (global columns) are embedded in the extended slab's numbering,
// checked and tiled (s27_build_kernel), and multiplied down the hierarchy with the aggregation of the extended slab
// (rap_aggregation_device: SciPy's accumulation order; the embedding is monotone, so every coarse entry has the bits of
// the global product).  The level below the slabs is handed back as CSR rows with global columns; the caller gathers
// them for the replicated tail (an ordinary hierarchy, as in dist.hip).
//
// The same schedule runs over a LOOPBACK group (all ranks in one process on one GPU, device copies in place of RCCL):
// tests/test_gpu_dist27.py compares it bit for bit with the single-GPU hierarchy.
#include <algorithm>
#include <array>
#include <cstring>
#include <memory>
#include <type_traits>

#include "common.h"
#include "rccl_dyn.h"

namespace omg {
namespace {

template <typename V>
struct SLevel {
    int nx = 0, ny = 0, nzo = 0;              // cells per line, lines per plane, OWNED planes
    Stencil27Plan<V> plan;                    // on the extended slab: nzo + 4 planes
    DevBuf<char> pool;                        // large levels: x, b, tmp as views into ONE allocation (hierarchy.hip pooled_vectors)
    DevBuf<V> x, tmp, b;
    V *xp = nullptr, *tp = nullptr;
    DevBuf<int32_t> cmap;                     // aggregate of the extended slab (natural index) -> slot in the next level's vectors
    int64_t pl() const { return int64_t(plan.g.hx) * plan.g.hy; }       // one colour's values of an aggregate plane
};

template <typename V>
struct SDist {
    using value_type = V;
    int rank = 0, n_ranks = 1;
    std::vector<SLevel<V>> lv;
    // the level below the slabs: this rank's planes in natural order with one ghost plane on either side (cb: the
    // restricted residual, ce: the correction), the gathered right-hand side and the replicated tail's solution
    int cnx = 0, cny = 0, cnzo = 0;
    DevBuf<V> cb, ce, gathered;
    DevBuf<double> full_b, full_x;
    DevCsrPlain coarse_rows;                  // its operator: this rank's rows, global columns
    omg_hierarchy *tail = nullptr;
    DevBuf<double> batch_partials, squares, nat;
    DevBuf<V> rows_out, rows_in;              // setup: the coefficient rows of one aggregate plane
    hipStream_t own = nullptr, stream = nullptr;
    ncclComm_t comm = nullptr;
    // Peer mode (round 6; dist.hip's PlaneDist has its own form): the halo exchanges as stores into the neighbours' ghost
    // planes — hipIpc mappings between processes, plain pointers inside one — ordered by flags instead of grouped
    // ncclSend / ncclRecv launches.  pflags (in MY memory, written by the neighbours): [0] / [1] "data of exchange k has
    // landed" from rank - 1 / rank + 1, [2] / [3] "I am done reading my ghost planes of exchange k - 1: overwrite them" from
    // rank - 1 / rank + 1, [8] status (bit 0: a bounded wait gave up), [9] the push launch's workgroup counter.
    struct PeerBufs {
        uint32_t *flags = nullptr;
        std::vector<V *> x, tmp, b;
        std::vector<void *> mapped;           // IPC mappings to close
    };
    DevBuf<uint32_t> pflags;
    PeerBufs peer[2];                         // 0: rank - 1, 1: rank + 1
    int p2p = 0;
    uint32_t pseq = 0;                        // exchanges so far (the same on every rank: one schedule)
    uint32_t p2p_spin = 1u << 22;
    bool rows_exchanged = false;              // the upper ghost planes hold the neighbour's coefficient rows
    bool ghosts_current = false;              // the ghost planes of b and x on the finest level are exchanged since the last load
    bool loaded = false;
    int exchanges = 0;                        // halo exchanges enqueued by the last cycle (DESIGN section 7 quotes it)

    SDist() = default;
    SDist(const SDist &) = delete;
    SDist &operator=(const SDist &) = delete;
    ~SDist() {
        for (PeerBufs &P : peer)
            for (void *m : P.mapped) (void)hipIpcCloseMemHandle(m);
        if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
        if (own) (void)hipStreamDestroy(own);
    }
};

int grid_of(int64_t n) { return int(std::max<int64_t>(1, std::min<int64_t>(65536, (n + 255) / 256))); }

// rows [row_off, row_off + n_src) of the destination are the source's, the others empty
__global__ void embed_rows_kernel(int64_t n_src, const int32_t *src, int64_t row_off, int64_t n_dst, int32_t *dst) {
    const int32_t nnz = src[n_src];
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= n_dst; r += (int64_t)gridDim.x * blockDim.x)
        dst[r] = r <= row_off ? 0 : r >= row_off + n_src ? nnz : src[r - row_off];
}
__global__ void shift_columns_kernel(int64_t nnz, int32_t *idx, int64_t delta, int64_t n_cols, int32_t *bad) {
    for (int64_t p = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; p < nnz; p += (int64_t)gridDim.x * blockDim.x) {
        const int64_t c = int64_t(idx[p]) + delta;
        if (c < 0 || c >= n_cols) { atomicOr(bad, 1); continue; }
        idx[p] = int32_t(c);
    }
}
// the plain 2 x 2 x 2 aggregation of an nz x ny x nx grid (C order), weight w: rows = coarse cells in C order, a row's
// columns ascending — openmg/operators.py:73-84 with the TRUE strides (the reference's equal them when shape[0] ==
// shape[2], which its callers' grids satisfy; a slab's extents do not)
__global__ void aggregation_kernel(int nx, int ny, int nz, double w, int32_t *indptr, int32_t *indices, double *data) {
    const int hx = nx / 2, hy = ny / 2;
    const int64_t rows = int64_t(hx) * hy * (nz / 2);
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r <= rows; r += (int64_t)gridDim.x * blockDim.x) {
        indptr[r] = int32_t(8 * r);
        if (r == rows) break;
        const int64_t I = r % hx, J = (r / hx) % hy, K = r / (int64_t(hx) * hy);
        int m = 0;
        for (int dk = 0; dk < 2; ++dk)
            for (int dj = 0; dj < 2; ++dj)
                for (int di = 0; di < 2; ++di, ++m) {
                    indices[8 * r + m] = int32_t(((2 * K + dk) * ny + 2 * J + dj) * nx + 2 * I + di);
                    data[8 * r + m] = w;
                }
    }
}
// aggregate (K, J, I) of a level's extended slab = cell (K + 1, J, I) of the next level's extended slab -> its slot there
__global__ void coarse_slot_kernel(int hx, int hy, int hz, int32_t *cmap) {
    const int64_t na = int64_t(hx) * hy * hz;                       // aggregates of this level's extended slab
    const int cz = hz + 2;                                           // planes of the next level's extended slab
    const int64_t nac = int64_t(hx / 2) * (hy / 2) * (cz / 2);
    for (int64_t a = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; a < na; a += (int64_t)gridDim.x * blockDim.x) {
        const int I = int(a % hx), J = int((a / hx) % hy), kc = int(a / (int64_t(hx) * hy)) + 1;
        const int c = (I & 1) | ((J & 1) << 1) | ((kc & 1) << 2);
        cmap[a] = int32_t(c * nac + (int64_t(kc >> 1) * (hy / 2) + (J >> 1)) * (hx / 2) + (I >> 1));
    }
}
// owned planes in natural order (host side, double) <-> the extended slab's octant layout (V)
template <typename V>
__global__ void slab_layout_kernel(double *nat, V *ext, int nx, int ny, int nzo, int to_ext) {
    const int64_t n = int64_t(nx) * ny * nzo;
    const int hx = nx / 2, hy = ny / 2;
    const int64_t na = int64_t(hx) * hy * ((nzo + 4) / 2);
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int i = int(r % nx), j = int((r / nx) % ny), k = int(r / (int64_t(nx) * ny)) + 2;
        const int c = (i & 1) | ((j & 1) << 1) | ((k & 1) << 2);
        const int64_t slot = c * na + (int64_t(k >> 1) * hy + (j >> 1)) * hx + (i >> 1);
        if (to_ext) ext[slot] = V(nat[r]);
        else nat[r] = double(ext[slot]);
    }
}
__global__ void add_arrays_kernel(double *acc, const double *v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) acc[i] += v[i];
}
__global__ void sqrt_arrays_kernel(const double *v, double *out, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) out[i] = sqrt(v[i]);
}

// ---- peer-store exchange ---------------------------------------------------------------------------------------------
__device__ __forceinline__ void sp_wait(const uint32_t *flag, uint32_t seq, uint32_t *status, uint32_t spin) {
    if (!flag) return;
    for (uint32_t n = 0;; ++n) {
        const uint32_t v = __hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if (int32_t(v - seq) >= 0) break;
        if (n >= spin) { __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// "overwrite my ghost planes": everything this rank's stream has done before (all readers of the ghost values of the
// previous exchange) is complete when this one-thread launch runs
__global__ void sp_ack_kernel(uint32_t *lo, uint32_t *hi, uint32_t seq) {
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    if (lo) __hip_atomic_store(lo, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (hi) __hip_atomic_store(hi, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
constexpr int SP_RUNS = 16;
template <typename V>
struct SpPush {
    const V *src[SP_RUNS];
    V *dst[SP_RUNS];
    int dir[SP_RUNS];                 // 0: into rank - 1, 1: into rank + 1
    int n_runs;
    long long count;                  // values per run
    const uint32_t *ack[2];           // my flags: the neighbour allows the overwrite
    uint32_t *data[2];                // the neighbours' flags: my data has landed
    uint32_t *done;                   // my workgroup counter
    uint32_t *status;
    uint32_t seq, spin;
};
// one workgroup row per run (blockIdx.y): wait for the target's permission, copy, and — the launch's last workgroup —
// tell the neighbours
template <typename V>
__global__ __launch_bounds__(256) void sp_push_kernel(const SpPush<V> a) {
    const int run = int(blockIdx.y);
    if (threadIdx.x == 0) sp_wait(a.ack[a.dir[run]], a.seq, a.status, a.spin);
    __syncthreads();
    const V *const src = a.src[run];
    V *const dst = a.dst[run];
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < a.count; i += (long long)gridDim.x * blockDim.x) dst[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t n_wg = gridDim.x * gridDim.y;
        const uint32_t before = __hip_atomic_fetch_add(a.done, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == n_wg - 1) {
            __hip_atomic_store(a.done, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            for (int e = 0; e < 2; ++e)
                if (a.data[e]) __hip_atomic_store(a.data[e], a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
__global__ void sp_wait_kernel(const uint32_t *lo, const uint32_t *hi, uint32_t seq, uint32_t *status, uint32_t spin) {
    sp_wait(lo, seq, status, spin);
    sp_wait(hi, seq, status, spin);
}

template <typename V>
std::unique_ptr<SDist<V>> sd_create(int rank, int n_ranks, int nx, int ny, int nz_global, int n_levels, const omg_csr &A_rows, double w) {
    OMG_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks && n_levels >= 1, "bad rank / level count");
    OMG_REQUIRE(nz_global % n_ranks == 0, "planes must divide evenly over the ranks");
    require_device();
    std::unique_ptr<SDist<V>> d(new SDist<V>);
    d->rank = rank;
    d->n_ranks = n_ranks;
    OMG_HIP(hipStreamCreateWithFlags(&d->own, hipStreamNonBlocking));
    d->stream = d->own;
    hipStream_t s = d->stream;
    const int nzo0 = nz_global / n_ranks;
    const int64_t plane0 = int64_t(nx) * ny, n_own = plane0 * nzo0, n_glob = plane0 * nz_global;
    validate_csr(A_rows, "A_rows");
    OMG_REQUIRE(A_rows.n_rows == n_own && A_rows.n_cols == n_glob, "A_rows must hold this rank's planes' rows with GLOBAL column indices");
    OMG_REQUIRE(n_glob < (int64_t(1) << 31), "int32 index range exceeded");
    d->lv.resize(size_t(n_levels));
    // the rank's rows in the numbering of its extended slab
    DevCsrPlain A;
    {
        SetupTimer tm("27-point slab: upload the rank's rows, embed them in the extended slab");
        const int64_t n_ext = plane0 * (nzo0 + 4);
        DevBuf<int32_t> own_ptr(size_t(n_own) + 1);
        own_ptr.upload(A_rows.indptr, size_t(n_own) + 1, s);
        A.n_rows = A.n_cols = n_ext;
        A.nnz = A_rows.nnz;
        A.indptr.alloc(size_t(n_ext) + 1);
        A.indices.alloc(size_t(std::max<int64_t>(A.nnz, 1)));
        A.data.alloc(size_t(std::max<int64_t>(A.nnz, 1)));
        A.indices.upload(A_rows.indices, size_t(A.nnz), s);
        A.data.upload(A_rows.data, size_t(A.nnz), s);
        hipLaunchKernelGGL(embed_rows_kernel, dim3(grid_of(n_ext + 1)), dim3(256), 0, s, n_own, own_ptr.p, 2 * plane0, n_ext, A.indptr.p);
        DevBuf<int32_t> bad(1);
        bad.zero(s);
        hipLaunchKernelGGL(shift_columns_kernel, dim3(grid_of(A.nnz)), dim3(256), 0, s, A.nnz, A.indices.p, (2 - int64_t(rank) * nzo0) * plane0, n_ext, bad.p);
        OMG_HIP(hipGetLastError());
        int32_t h_bad = 0;
        bad.download(&h_bad, 1, s);
        OMG_HIP(hipStreamSynchronize(s));
        OMG_REQUIRE(!h_bad, "27-point slab: a row couples to a plane further than one away from the rank's slab");
    }
    int lx = nx, ly = ny, lz = nzo0;
    for (int l = 0; l < n_levels; ++l) {
        SLevel<V> &L = d->lv[size_t(l)];
        OMG_REQUIRE(lz >= 2 && !(lz & 1) && !(lx & 1) && !(ly & 1), "every distributed level needs an even number (>= 2) of planes per rank and even extents");
        L.nx = lx; L.ny = ly; L.nzo = lz;
        L.plan.build_slab(A, lx, ly, lz + 4, rank == 0, rank == n_ranks - 1, w, s);
        const S27Geom &g = L.plan.g;
        const size_t nv = size_t(8) * size_t(g.na);
        {
            // the three vectors the sweeps stream side by side out of one allocation, each 2 MiB-aligned + its stagger, b in the
            // middle: what hierarchy.hip measured for whole grids (configs[4]: + 3 %); OMG_VEC_POOL=0: three allocations
            static const bool pool_on = [] { const char *e = experiment_env("OMG_VEC_POOL"); return !(e && e[0] == '0'); }();
            if (pool_on && nv >= (size_t(1) << 20)) {
                const size_t MB2 = size_t(2) << 20, bytes = nv * sizeof(V);
                const size_t span = (bytes + 2 * DEVBUF_SLACK + vector_stagger(2) + MB2 - 1) / MB2 * MB2;
                L.pool.alloc(3 * span);
                L.x.borrow(reinterpret_cast<V *>(L.pool.p + DEVBUF_SLACK), nv);
                L.b.borrow(reinterpret_cast<V *>(L.pool.p + span + DEVBUF_SLACK + vector_stagger(2)), nv);
                L.tmp.borrow(reinterpret_cast<V *>(L.pool.p + 2 * span + DEVBUF_SLACK + vector_stagger(1)), nv);
            } else {
                L.x.alloc(nv); L.tmp.alloc(nv); L.b.alloc(nv);
            }
        }
        L.x.zero(s); L.tmp.zero(s); L.b.zero(s);
        L.xp = L.x.p; L.tp = L.tmp.p;
        if (l == 0) {
            // where the finest slab's tiles lie moves its sweeps by up to 15 % (Stencil27Plan::place_tiles), and the slowest rank sets the cycle
            L.plan.place_tiles(L.x.p, L.tmp.p, L.b.p, s);
            L.x.zero(s); L.tmp.zero(s); L.b.zero(s);
        }
        // the next level's operator: (R A) R^T with the aggregation of the extended slab; it comes out with ONE ghost plane
        // on either side and is embedded in the next extended slab (two)
        SetupTimer tm("27-point slab: Galerkin product of a level");
        const int64_t n_ext = int64_t(lx) * ly * (lz + 4), nc = n_ext / 8;
        DevCsrPlain R, C;
        R.n_rows = nc; R.n_cols = n_ext; R.nnz = n_ext;
        R.indptr.alloc(size_t(nc) + 1);
        R.indices.alloc(size_t(n_ext));
        R.data.alloc(size_t(n_ext));
        hipLaunchKernelGGL(aggregation_kernel, dim3(grid_of(nc + 1)), dim3(256), 0, s, lx, ly, lz + 4, w, R.indptr.p, R.indices.p, R.data.p);
        OMG_HIP(hipGetLastError());
        OMG_REQUIRE(rap_aggregation_device(R, A, C, s), "27-point slab: the Galerkin product of a level does not fit the fused kernel");
        const int cx = lx / 2, cy = ly / 2, cz = lz / 2;
        const int64_t cplane = int64_t(cx) * cy;
        if (l + 1 < n_levels) {
            const int64_t nn = cplane * (cz + 4);
            DevCsrPlain N;
            N.n_rows = N.n_cols = nn;
            N.nnz = C.nnz;
            N.indptr.alloc(size_t(nn) + 1);
            hipLaunchKernelGGL(embed_rows_kernel, dim3(grid_of(nn + 1)), dim3(256), 0, s, C.n_rows, C.indptr.p, cplane, nn, N.indptr.p);
            DevBuf<int32_t> bad(1);
            bad.zero(s);
            hipLaunchKernelGGL(shift_columns_kernel, dim3(grid_of(C.nnz)), dim3(256), 0, s, C.nnz, C.indices.p, cplane, nn, bad.p);
            OMG_HIP(hipGetLastError());
            OMG_HIP(hipStreamSynchronize(s));
            N.indices = std::move(C.indices);
            N.data = std::move(C.data);
            A = std::move(N);
            L.cmap.alloc(size_t(g.na));
            hipLaunchKernelGGL(coarse_slot_kernel, dim3(grid_of(g.na)), dim3(256), 0, s, g.hx, g.hy, g.hz, L.cmap.p);
            OMG_HIP(hipGetLastError());
        } else {
            // the level below the slabs: the rank's rows with global columns (its ghost planes' rows are empty, the first
            // owned row is row `cplane` of C)
            DevCsrPlain &G = d->coarse_rows;
            G.n_rows = cplane * cz;
            G.n_cols = cplane * cz * n_ranks;
            G.nnz = C.nnz;
            G.indptr.alloc(size_t(G.n_rows) + 1);
            OMG_HIP(hipMemcpyAsync(G.indptr.p, C.indptr.p + cplane, (size_t(G.n_rows) + 1) * sizeof(int32_t), hipMemcpyDeviceToDevice, s));
            DevBuf<int32_t> bad(1);
            bad.zero(s);
            hipLaunchKernelGGL(shift_columns_kernel, dim3(grid_of(C.nnz)), dim3(256), 0, s, C.nnz, C.indices.p, (int64_t(rank) * cz - 1) * cplane, G.n_cols, bad.p);
            OMG_HIP(hipGetLastError());
            int32_t h_bad = 0;
            bad.download(&h_bad, 1, s);
            OMG_HIP(hipStreamSynchronize(s));
            OMG_REQUIRE(!h_bad, "internal: a coarse row of the rank couples outside the global grid");
            G.indices = std::move(C.indices);
            G.data = std::move(C.data);
            d->cnx = cx; d->cny = cy; d->cnzo = cz;
        }
        lx = cx; ly = cy; lz = cz;
    }
    OMG_REQUIRE(d->cnzo >= 1, "the level below the slabs needs at least one plane per rank");
    const int64_t cplane = int64_t(d->cnx) * d->cny;
    d->cb.alloc(size_t(cplane * (d->cnzo + 2)));
    d->ce.alloc(size_t(cplane * (d->cnzo + 2)));
    d->cb.zero(s); d->ce.zero(s);
    d->gathered.alloc(size_t(cplane * d->cnzo * n_ranks));
    d->full_b.alloc(size_t(cplane * d->cnzo * n_ranks));
    d->full_x.alloc(size_t(cplane * d->cnzo * n_ranks));
    d->squares.alloc(64);
    d->nat.alloc(size_t(n_own));
    {
        const Stencil27Plan<V> &P = d->lv[0].plan;
        d->batch_partials.alloc(size_t(64) * size_t(4) * size_t(P.g.n_wg));
        d->batch_partials.zero(s);
    }
    size_t most = 0;
    for (const SLevel<V> &L : d->lv) most = std::max(most, L.plan.plane_rows_count());
    d->rows_out.alloc(most);
    d->rows_in.alloc(most);
    OMG_HIP(hipStreamSynchronize(s));
    return d;
}

// The schedule over the ranks this process drives: ONE rank with RCCL exchanges, or all of them (loopback group: the
// same launches per rank in the same order on one stream, device copies in place of the exchanges).
template <typename V>
struct SExchange {
    using D = SDist<V>;
    std::vector<D *> ranks;                   // ascending
    bool loopback = false;

    D *at(int rank) const { return loopback ? ranks[size_t(rank)] : nullptr; }

    // setup: colours 0 .. 3 of every level's upper ghost aggregate plane take the coefficient rows of the neighbour's
    // first owned plane
    void exchange_rows() {
        const int nd = int(ranks[0]->lv.size());
        for (int l = 0; l < nd; ++l) {
            for (D *d : ranks) {
                Stencil27Plan<V> &P = d->lv[size_t(l)].plan;
                const size_t cnt = P.plane_rows_count();
                if (d->rank > 0) P.pack_plane_rows(1, d->rows_out.p, d->stream);
                if (!loopback && d->n_ranks > 1) {
                    OMG_NCCL(g_rccl.GroupStart());
                    if (d->rank > 0) OMG_NCCL(g_rccl.Send(d->rows_out.p, cnt, NcclType<V>::value, d->rank - 1, d->comm, d->stream));
                    if (d->rank + 1 < d->n_ranks) OMG_NCCL(g_rccl.Recv(d->rows_in.p, cnt, NcclType<V>::value, d->rank + 1, d->comm, d->stream));
                    OMG_NCCL(g_rccl.GroupEnd());
                }
            }
            for (D *d : ranks) {
                if (d->rank + 1 >= d->n_ranks) continue;
                Stencil27Plan<V> &P = d->lv[size_t(l)].plan;
                const V *src = loopback ? at(d->rank + 1)->rows_out.p : d->rows_in.p;
                P.unpack_plane_rows(P.g.hz - 1, src, d->stream);
            }
            for (D *d : ranks) OMG_HIP(hipStreamSynchronize(d->stream));      // (rows_out is reused by the next level)
        }
        for (D *d : ranks) d->rows_exchanged = true;
    }

    // what: 0 — x after a sweep: colours 4 .. 7 of the boundary aggregate planes, both ways; 1 — x after a load: also
    // colours 0 .. 3 downwards; 2 — b: colours 0 .. 3 of the first owned plane downwards (into the neighbour's upper ghost)
    void halo(int l, int what) {
        for (D *d : ranks) {
            SLevel<V> &L = d->lv[size_t(l)];
            const int64_t na = L.plan.g.na, pl = L.pl();
            const int hz = L.plan.g.hz;
            V *mine = what == 2 ? L.b.p : L.xp;
            const int up_lo = what == 2 ? 8 : 4, up_hi = 8;                          // colours that travel upwards (rank -> rank + 1)
            const int dn_lo = what == 0 ? 4 : 0, dn_hi = what == 2 ? 4 : 8;         // ... downwards
            const bool lo_nb = d->rank > 0, hi_nb = d->rank + 1 < d->n_ranks;
            if (lo_nb || hi_nb) ++d->exchanges;
            if (d->p2p) continue;                                                    // (below: in phases over the ranks)
            if (loopback) {
                // (only the receives: every rank pulls from its neighbours' owned planes)
                if (lo_nb) {
                    SLevel<V> &O = at(d->rank - 1)->lv[size_t(l)];
                    const V *theirs = what == 2 ? O.b.p : O.xp;
                    for (int c = up_lo; c < up_hi; ++c)
                        OMG_HIP(hipMemcpyAsync(mine + c * na, theirs + c * na + int64_t(hz - 2) * pl, size_t(pl) * sizeof(V), hipMemcpyDeviceToDevice, d->stream));
                }
                if (hi_nb) {
                    SLevel<V> &O = at(d->rank + 1)->lv[size_t(l)];
                    const V *theirs = what == 2 ? O.b.p : O.xp;
                    for (int c = dn_lo; c < dn_hi; ++c)
                        OMG_HIP(hipMemcpyAsync(mine + c * na + int64_t(hz - 1) * pl, theirs + c * na + pl, size_t(pl) * sizeof(V), hipMemcpyDeviceToDevice, d->stream));
                }
            } else if (d->n_ranks > 1) {
                OMG_NCCL(g_rccl.GroupStart());
                if (hi_nb) {
                    for (int c = up_lo; c < up_hi; ++c)
                        OMG_NCCL(g_rccl.Send(mine + c * na + int64_t(hz - 2) * pl, size_t(pl), NcclType<V>::value, d->rank + 1, d->comm, d->stream));
                    for (int c = dn_lo; c < dn_hi; ++c)
                        OMG_NCCL(g_rccl.Recv(mine + c * na + int64_t(hz - 1) * pl, size_t(pl), NcclType<V>::value, d->rank + 1, d->comm, d->stream));
                }
                if (lo_nb) {
                    for (int c = dn_lo; c < dn_hi; ++c)
                        OMG_NCCL(g_rccl.Send(mine + c * na + pl, size_t(pl), NcclType<V>::value, d->rank - 1, d->comm, d->stream));
                    for (int c = up_lo; c < up_hi; ++c)
                        OMG_NCCL(g_rccl.Recv(mine + c * na, size_t(pl), NcclType<V>::value, d->rank - 1, d->comm, d->stream));
                }
                OMG_NCCL(g_rccl.GroupEnd());
            }
        }
        halo_p2p(l, what);
    }

    // Peer mode: the same planes as stores into the neighbours' ghost planes.  Three launches per rank — permission, push
    // (which waits for the neighbours' permission itself), wait for the neighbours' data — enqueued phase by phase over the
    // ranks this process drives, so that a loopback group (every rank on ONE stream) never waits for a flag a later launch
    // of the same stream would raise; with one rank per process the phases are simply consecutive on its stream.
    void halo_p2p(int l, int what) {
        const int up_lo = what == 2 ? 8 : 4, up_hi = 8;
        const int dn_lo = what == 0 ? 4 : 0, dn_hi = what == 2 ? 4 : 8;
        bool any = false;
        for (D *d : ranks) any = any || d->p2p;
        if (!any) return;
        for (D *d : ranks) {
            OMG_REQUIRE(d->p2p, "peer mode must be on for every rank of the group");
            ++d->pseq;
            const bool lo_nb = d->rank > 0, hi_nb = d->rank + 1 < d->n_ranks;
            if (!lo_nb && !hi_nb) continue;
            // my permission lands in rank - 1's "ack from rank + 1" [3] and in rank + 1's "ack from rank - 1" [2]
            hipLaunchKernelGGL(sp_ack_kernel, dim3(1), dim3(1), 0, d->stream, lo_nb ? d->peer[0].flags + 3 : nullptr,
                               hi_nb ? d->peer[1].flags + 2 : nullptr, d->pseq);
        }
        for (D *d : ranks) {
            const bool lo_nb = d->rank > 0, hi_nb = d->rank + 1 < d->n_ranks;
            if (!lo_nb && !hi_nb) continue;
            SLevel<V> &L = d->lv[size_t(l)];
            const int64_t na = L.plan.g.na, pl = L.pl();
            const int hz = L.plan.g.hz;
            const bool in_x = L.xp == L.x.p;                                         // (the neighbours have swapped as often as I have)
            V *mine = what == 2 ? L.b.p : L.xp;
            auto theirs = [&](int e) -> V * {
                const typename D::PeerBufs &P = d->peer[e];
                return what == 2 ? P.b[size_t(l)] : in_x ? P.x[size_t(l)] : P.tmp[size_t(l)];
            };
            SpPush<V> a;
            std::memset(&a, 0, sizeof(a));
            int n = 0;
            if (hi_nb)
                for (int c = up_lo; c < up_hi; ++c, ++n) { a.src[n] = mine + c * na + int64_t(hz - 2) * pl; a.dst[n] = theirs(1) + c * na; a.dir[n] = 1; }
            if (lo_nb)
                for (int c = dn_lo; c < dn_hi; ++c, ++n) { a.src[n] = mine + c * na + pl; a.dst[n] = theirs(0) + c * na + int64_t(hz - 1) * pl; a.dir[n] = 0; }
            OMG_REQUIRE(n <= SP_RUNS, "internal: too many runs in one exchange");
            a.n_runs = n;
            a.count = pl;
            a.ack[0] = lo_nb ? d->pflags.p + 2 : nullptr;
            a.ack[1] = hi_nb ? d->pflags.p + 3 : nullptr;
            a.data[0] = lo_nb ? d->peer[0].flags + 1 : nullptr;                      // rank - 1's "data from rank + 1"
            a.data[1] = hi_nb ? d->peer[1].flags + 0 : nullptr;                      // rank + 1's "data from rank - 1"
            a.done = d->pflags.p + 9;
            a.status = d->pflags.p + 8;
            a.seq = d->pseq;
            a.spin = d->p2p_spin;
            if (n) {
                const unsigned gx = unsigned(std::max<int64_t>(1, std::min<int64_t>(32, (pl + 2047) / 2048)));
                hipLaunchKernelGGL((sp_push_kernel<V>), dim3(gx, unsigned(n)), dim3(256), 0, d->stream, a);
            } else {
                // (nothing travels from this rank in this exchange, but its neighbours wait for its flag)
                hipLaunchKernelGGL(sp_ack_kernel, dim3(1), dim3(1), 0, d->stream, a.data[0], a.data[1], d->pseq);
            }
        }
        for (D *d : ranks) {
            const bool lo_nb = d->rank > 0, hi_nb = d->rank + 1 < d->n_ranks;
            if (!lo_nb && !hi_nb) continue;
            hipLaunchKernelGGL(sp_wait_kernel, dim3(1), dim3(1), 0, d->stream, lo_nb ? d->pflags.p + 0 : nullptr, hi_nb ? d->pflags.p + 1 : nullptr,
                               d->pseq, d->pflags.p + 8, d->p2p_spin);
        }
        OMG_HIP(hipGetLastError());
    }

    // right-hand side of the level below the slabs: gathered, solved by the replicated tail, this rank's planes (and
    // one ghost plane on either side) of the correction taken out of it
    void tail_solve(int pre, int post) {
        for (D *d : ranks) {
            const int64_t plane = int64_t(d->cnx) * d->cny, own = plane * d->cnzo;
            const V *mine = d->cb.p + plane;
            if (loopback) {
                for (D *o : ranks)
                    OMG_HIP(hipMemcpyAsync(o->gathered.p + int64_t(d->rank) * own, mine, size_t(own) * sizeof(V), hipMemcpyDeviceToDevice, d->stream));
            } else if (d->n_ranks > 1) {
                OMG_NCCL(g_rccl.AllGather(mine, d->gathered.p, size_t(own), NcclType<V>::value, d->comm, d->stream));
            } else {
                OMG_HIP(hipMemcpyAsync(d->gathered.p, mine, size_t(own) * sizeof(V), hipMemcpyDeviceToDevice, d->stream));
            }
        }
        for (D *d : ranks) {
            const int64_t plane = int64_t(d->cnx) * d->cny, all = plane * d->cnzo * d->n_ranks;
            launch_gather<V, double>(d->gathered.p, nullptr, d->full_b.p, all, d->stream);         // (the tail's device boundary is double)
            if (omg_hierarchy_cycle_dev(d->tail, d->full_b.p, d->full_x.p, pre, post, d->stream) != OMG_OK)
                throw Error(OMG_ERR_HIP, std::string("replicated tail cycle: ") + omg_last_error());
            // planes [k0 - 1, k0 + own + 1) of the correction, clipped to the grid (the rest stays zero)
            const int64_t k0 = int64_t(d->rank) * d->cnzo, nzg = int64_t(d->n_ranks) * d->cnzo;
            const int64_t lo = std::max<int64_t>(0, k0 - 1), hi = std::min<int64_t>(nzg, k0 + d->cnzo + 1);
            launch_gather<double, V>(d->full_x.p + lo * plane, nullptr, d->ce.p + (lo - (k0 - 1)) * plane, (hi - lo) * plane, d->stream);
        }
    }

    // openmg/__init__.py:199-227 over level l of every rank (dead work removed as hierarchy.hip's cycle_body does)
    // slot_prev (level 0, per rank, nullable): the first sweep also leaves the squares of b - A x_old there;
    // slot_this: the last post-smoothing launch leaves its rows' squares there
    void cycle(int l, int pre, int post, const std::vector<double *> *slot_prev, const std::vector<double *> *slot_this, bool x_zero) {
        const int nd = int(ranks[0]->lv.size());
        const bool last = l + 1 == nd;
        bool zero_now = x_zero;
        if (x_zero && pre == 0) {
            for (D *d : ranks) { SLevel<V> &L = d->lv[size_t(l)]; OMG_HIP(hipMemsetAsync(L.xp, 0, L.x.n * sizeof(V), d->stream)); }    // :191-192
            zero_now = false;
        }
        for (int it = 0; it < pre; ++it) {                                                        // :201
            for (size_t r = 0; r < ranks.size(); ++r) {
                D *d = ranks[r];
                SLevel<V> &L = d->lv[size_t(l)];
                double *old = (it == 0 && !zero_now && slot_prev) ? (*slot_prev)[r] : nullptr;
                L.plan.sweep(L.xp, L.tp, L.b.p, zero_now, old, it + 1 == pre, nullptr, d->stream);
                std::swap(L.xp, L.tp);
            }
            halo(l, 0);
            zero_now = false;
        }
        for (D *d : ranks) {                                                                      // :209, :210
            SLevel<V> &L = d->lv[size_t(l)];
            L.plan.residual_restrict(L.xp, L.b.p, pre >= 1, last ? nullptr : L.cmap.p, last ? d->cb.p : d->lv[size_t(l) + 1].b.p, d->stream);
        }
        if (last) {
            tail_solve(pre, post);
        } else {
            halo(l + 1, 2);
            cycle(l + 1, pre, post, nullptr, nullptr, true);                                      // :213
        }
        for (D *d : ranks) {                                                                      // :214, :220 / :224
            SLevel<V> &L = d->lv[size_t(l)];
            L.plan.prolong(L.xp, last ? d->ce.p : d->lv[size_t(l) + 1].xp, last ? nullptr : L.cmap.p, d->stream);
        }
        for (int it = 0; it < post; ++it) {                                                       // :216-222
            const bool fin = it + 1 == post && slot_this;
            for (size_t r = 0; r < ranks.size(); ++r) {
                D *d = ranks[r];
                SLevel<V> &L = d->lv[size_t(l)];
                L.plan.sweep(L.xp, L.tp, L.b.p, false, nullptr, fin, fin ? (*slot_this)[r] : nullptr, d->stream);
                std::swap(L.xp, L.tp);
            }
            halo(l, 0);
        }
    }

    // n cycles from the loaded vectors; every cycle's GLOBAL norm (:227) -> norms_out (host, nullable).  As
    // omg_resident_cycles does for a 27-point level: with pre >= 1 the first sweep of cycle j + 1 squares the residuals
    // of the iterate it starts from; the last cycle of a chunk (every cycle when pre = 0) runs the norm kernel.
    void run(int pre, int post, int n, double *norms_out) {
        if (n <= 0) return;
        for (D *d : ranks) {
            OMG_REQUIRE(d->loaded, "omg_sdist_load has not been called");
            OMG_REQUIRE(d->tail, "omg_sdist_set_tail has not been called");
            OMG_REQUIRE(d->rows_exchanged || d->n_ranks == 1, "the neighbours' coefficient rows have not been exchanged (omg_sdist_connect / group)");
        }
        bool current = true;
        for (D *d : ranks) current = current && d->ghosts_current;
        if (!current) {
            halo(0, 2);
            halo(0, 1);
            for (D *d : ranks) d->ghosts_current = true;
        }
        for (D *d : ranks) d->exchanges = 0;
        constexpr int CHUNK = 64;
        std::vector<double> host(size_t(n), 0.0);
        for (int k0 = 0; k0 < n; k0 += CHUNK) {
            const int cnt = std::min(CHUNK, n - k0);
            for (int j = 0; j < cnt; ++j) {
                std::vector<double *> prev, now;
                for (D *d : ranks) {
                    const size_t nb = size_t(4) * size_t(d->lv[0].plan.g.n_wg);
                    prev.push_back(d->batch_partials.p + size_t(std::max(j - 1, 0)) * nb);
                    now.push_back(d->batch_partials.p + size_t(j) * nb);
                }
                const bool own_norm = j + 1 == cnt || pre == 0;
                cycle(0, pre, post, (pre >= 1 && j > 0) ? &prev : nullptr, own_norm ? &now : nullptr, false);
                if (own_norm)
                    for (size_t r = 0; r < ranks.size(); ++r) {
                        SLevel<V> &L = ranks[r]->lv[0];
                        L.plan.norm(L.xp, L.b.p, L.plan.have67, now[r], ranks[r]->stream);
                    }
            }
            for (D *d : ranks) {
                const int64_t nb = int64_t(4) * d->lv[0].plan.g.n_wg;
                launch_sum_batch(d->batch_partials.p, nb, nb, cnt, d->squares.p, false, d->stream);
            }
            D *z = ranks[0];
            if (loopback) {
                for (size_t r = 1; r < ranks.size(); ++r)                                          // ranks in ascending order (one stream)
                    hipLaunchKernelGGL(add_arrays_kernel, dim3(1), dim3(64), 0, z->stream, z->squares.p, ranks[r]->squares.p, cnt);
            } else if (z->n_ranks > 1) {
                OMG_NCCL(g_rccl.AllReduce(z->squares.p, z->squares.p, size_t(cnt), ncclDouble, ncclSum, z->comm, z->stream));
            }
            hipLaunchKernelGGL(sqrt_arrays_kernel, dim3(1), dim3(64), 0, z->stream, z->squares.p, z->squares.p, cnt);
            OMG_HIP(hipGetLastError());
            OMG_HIP(hipMemcpyAsync(host.data() + k0, z->squares.p, size_t(cnt) * sizeof(double), hipMemcpyDeviceToHost, z->stream));
            for (D *d : ranks) OMG_HIP(hipStreamSynchronize(d->stream));
        }
        if (norms_out) std::memcpy(norms_out, host.data(), size_t(n) * sizeof(double));
    }
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

}  // namespace
}  // namespace omg

using namespace omg;

struct omg_sdist {
    std::unique_ptr<omg::SDist<double>> d;
    std::unique_ptr<omg::SDist<float>> f;
};
struct omg_sdist_group {
    std::vector<omg_sdist *> ranks;
};

namespace {
template <typename F>
void with(omg_sdist *d, F &&f) {
    OMG_REQUIRE(d != nullptr && (d->d || d->f), "null handle");
    if (d->f) f(d->f.get());
    else f(d->d.get());
}
template <typename HP>
using value_of = typename std::remove_pointer<HP>::type::value_type;
}  // namespace

namespace {
template <typename V>
void sd_own_buffers(SDist<V> *d, std::vector<void *> &out, std::vector<size_t> &shift) {
    if (!d->pflags.p) {
        d->pflags.alloc(64);
        d->pflags.zero(d->stream);
        OMG_HIP(hipStreamSynchronize(d->stream));
    }
    out = {d->pflags.p};
    shift = {d->pflags.shift};
    for (auto &L : d->lv) {
        out.push_back(L.x.p); out.push_back(L.tmp.p); out.push_back(L.b.p);
        shift.push_back(L.x.shift); shift.push_back(L.tmp.shift); shift.push_back(L.b.shift);
    }
}
template <typename V>
void sd_attach(SDist<V> *d, int peer_rank, const std::vector<void *> &bufs) {
    OMG_REQUIRE(peer_rank == d->rank - 1 || peer_rank == d->rank + 1, "peer mode maps the slab's two neighbours only");
    auto &P = d->peer[peer_rank == d->rank - 1 ? 0 : 1];
    P.flags = static_cast<uint32_t *>(bufs[0]);
    P.x.clear(); P.tmp.clear(); P.b.clear();
    for (size_t l = 0; l < d->lv.size(); ++l) {
        P.x.push_back(static_cast<V *>(bufs[1 + 3 * l]));
        P.tmp.push_back(static_cast<V *>(bufs[2 + 3 * l]));
        P.b.push_back(static_cast<V *>(bufs[3 + 3 * l]));
    }
}
}  // namespace

extern "C" {

int omg_sdist_create(int rank, int n_ranks, int nx, int ny, int nz_global, int n_levels, const omg_csr *A_rows, double weight, int dtype,
                     omg_sdist **out) {
    return guarded([&] {
        OMG_REQUIRE(out && A_rows, "null argument");
        *out = nullptr;
        OMG_REQUIRE(dtype == OMG_DTYPE_F64 || dtype == OMG_DTYPE_F32, "unknown dtype");
        std::unique_ptr<omg_sdist> h(new omg_sdist);
        if (dtype == OMG_DTYPE_F32) h->f = sd_create<float>(rank, n_ranks, nx, ny, nz_global, n_levels, *A_rows, weight);
        else h->d = sd_create<double>(rank, n_ranks, nx, ny, nz_global, n_levels, *A_rows, weight);
        *out = h.release();
    });
}

int omg_sdist_destroy(omg_sdist *d) {
    return guarded([&] {
        if (!d) return;
        if (d->d || d->f) with(d, [&](auto *dd) { (void)hipStreamSynchronize(dd->stream); });
        delete d;
    });
}

int omg_sdist_coarse_size(omg_sdist *d, int64_t *n_rows, int64_t *n_cols, int64_t *nnz) {
    return guarded([&] {
        OMG_REQUIRE(n_rows && n_cols && nnz, "null argument");
        with(d, [&](auto *dd) { *n_rows = dd->coarse_rows.n_rows; *n_cols = dd->coarse_rows.n_cols; *nnz = dd->coarse_rows.nnz; });
    });
}

int omg_sdist_coarse_fetch(omg_sdist *d, int32_t *indptr, int32_t *indices, double *data) {
    return guarded([&] {
        OMG_REQUIRE(indptr && indices && data, "null argument");
        with(d, [&](auto *dd) {
            const DevCsrPlain &G = dd->coarse_rows;
            download_staged(indptr, G.indptr.p, (size_t(G.n_rows) + 1) * sizeof(int32_t), dd->stream);
            download_staged(indices, G.indices.p, size_t(G.nnz) * sizeof(int32_t), dd->stream);
            download_staged(data, G.data.p, size_t(G.nnz) * sizeof(double), dd->stream);
        });
    });
}

int omg_sdist_set_tail(omg_sdist *d, omg_hierarchy *tail) {
    return guarded([&] {
        OMG_REQUIRE(tail, "null argument");
        with(d, [&](auto *dd) {
            int64_t n = 0;
            OMG_REQUIRE(omg_hierarchy_level_rows(tail, 0, &n) == OMG_OK && n == int64_t(dd->cnx) * dd->cny * dd->cnzo * dd->n_ranks,
                        "tail hierarchy's finest level must be the level below the slabs");
            dd->tail = tail;
        });
    });
}

/* joins the communicator and — collective — fetches the neighbour's coefficient rows for the ghost planes */
int omg_sdist_connect(omg_sdist *d, const void *unique_id128) {
    return guarded([&] {
        OMG_REQUIRE(unique_id128, "null argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            g_rccl.load();
            ncclUniqueId id;
            std::memcpy(&id, unique_id128, sizeof(id));
            OMG_NCCL(g_rccl.CommInitRank(&dd->comm, dd->n_ranks, id, dd->rank));
            SExchange<V> ex;
            ex.ranks = {dd};
            ex.exchange_rows();
        });
    });
}

int omg_sdist_rccl_ranks(omg_sdist *d, int *count) {
    return guarded([&] {
        OMG_REQUIRE(count, "null argument");
        with(d, [&](auto *dd) {
            *count = 0;
            if (dd->comm) OMG_NCCL(g_rccl.CommCount(dd->comm, count));
        });
    });
}

int omg_sdist_info(omg_sdist *d, int level, int64_t *out8) {
    return guarded([&] {
        OMG_REQUIRE(out8, "null argument");
        with(d, [&](auto *dd) {
            OMG_REQUIRE(level >= 0 && level < int(dd->lv.size()), "level out of range");
            const auto &L = dd->lv[size_t(level)];
            const S27Geom &g = L.plan.g;
            const int64_t v[8] = {L.nx, L.ny, L.nzo, g.rg, g.n_wg, g.wpb, int64_t(dd->lv.size()), dd->exchanges};
            for (int i = 0; i < 8; ++i) out8[i] = v[i];
        });
    });
}

/* this rank's planes of the finest level in natural order (x0 NULL: zeros); the ghost planes are exchanged by the next
 * omg_sdist_cycles */
int omg_sdist_load(omg_sdist *d, const double *b_local, const double *x0_local) {
    return guarded([&] {
        OMG_REQUIRE(b_local, "null argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            auto &L = dd->lv[0];
            const int64_t n = int64_t(L.nx) * L.ny * L.nzo;
            auto put = [&](const double *host, V *ext) {
                OMG_HIP(hipMemsetAsync(ext, 0, L.x.n * sizeof(V), dd->stream));
                if (!host) return;
                dd->nat.upload(host, size_t(n), dd->stream);
                hipLaunchKernelGGL(slab_layout_kernel<V>, dim3(grid_of(n)), dim3(256), 0, dd->stream, dd->nat.p, ext, L.nx, L.ny, L.nzo, 1);
                OMG_HIP(hipGetLastError());
                OMG_HIP(hipStreamSynchronize(dd->stream));             // (nat is reused by the next vector)
            };
            put(b_local, L.b.p);
            put(x0_local, L.xp);
            L.plan.have67 = false;
            OMG_HIP(hipStreamSynchronize(dd->stream));
            dd->ghosts_current = false;
            dd->loaded = true;
        });
    });
}

int omg_sdist_fetch(omg_sdist *d, double *x_local) {
    return guarded([&] {
        OMG_REQUIRE(x_local, "null argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            OMG_REQUIRE(dd->loaded, "nothing loaded");
            auto &L = dd->lv[0];
            const int64_t n = int64_t(L.nx) * L.ny * L.nzo;
            hipLaunchKernelGGL(slab_layout_kernel<V>, dim3(grid_of(n)), dim3(256), 0, dd->stream, dd->nat.p, L.xp, L.nx, L.ny, L.nzo, 0);
            OMG_HIP(hipGetLastError());
            dd->nat.download(x_local, size_t(n), dd->stream);
            OMG_HIP(hipStreamSynchronize(dd->stream));
        });
    });
}

int omg_sdist_sync(omg_sdist *d) {
    return guarded([&] { with(d, [&](auto *dd) { OMG_HIP(hipStreamSynchronize(dd->stream)); }); });
}

/* n_cycles V(pre, post) cycles, every cycle's GLOBAL residual norm computed and returned; collective */
int omg_sdist_cycles(omg_sdist *d, int pre, int post, int n_cycles, double *norms) {
    return guarded([&] {
        OMG_REQUIRE(pre >= 0 && post >= 0 && n_cycles >= 0, "bad argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            OMG_REQUIRE(dd->n_ranks == 1 || dd->comm, "omg_sdist_connect has not been called");
            SExchange<V> ex;
            ex.ranks = {dd};
            ex.run(pre, post, n_cycles, norms);
        });
    });
}

/* ---- peer mode for the halo exchanges ----------------------------------------------------------
 * As omg_pdist_p2p_*: every rank exports IPC handles of what its neighbours store into — its flag words, then x / tmp / b
 * of every level: 1 + 3 per level handles of 64 bytes —, the control plane hands them round, a rank opens its NEIGHBOURS'
 * (omg_sdist_p2p_open; same process: omg_sdist_p2p_local), then omg_sdist_p2p_enable(1).  The gather below the slabs and
 * the norm's reduction stay on the communicator. */
int omg_sdist_p2p_handle_count(omg_sdist *d, int *count) {
    return guarded([&] {
        OMG_REQUIRE(count, "null argument");
        with(d, [&](auto *dd) { *count = 1 + 3 * int(dd->lv.size()); });
    });
}

int omg_sdist_p2p_handles(omg_sdist *d, void *handles64, int capacity) {
    return guarded([&] {
        OMG_REQUIRE(handles64, "null argument");
        with(d, [&](auto *dd) {
            std::vector<void *> bufs;
            std::vector<size_t> shift;
            sd_own_buffers(dd, bufs, shift);
            OMG_REQUIRE(capacity >= int(bufs.size()), "handle buffer too small");
            OMG_HIP(hipStreamSynchronize(dd->stream));
            for (size_t i = 0; i < bufs.size(); ++i) {
                hipIpcMemHandle_t h;
                OMG_HIP(hipIpcGetMemHandle(&h, static_cast<char *>(bufs[i]) - DEVBUF_SLACK - shift[i]));     // (the allocation's base)
                std::memcpy(static_cast<char *>(handles64) + 64 * i, &h, 64);
            }
        });
    });
}

int omg_sdist_p2p_open(omg_sdist *d, int peer_rank, const void *handles64, int count) {
    return guarded([&] {
        OMG_REQUIRE(handles64, "null argument");
        with(d, [&](auto *dd) {
            OMG_REQUIRE(peer_rank == dd->rank - 1 || peer_rank == dd->rank + 1, "peer mode maps the slab's two neighbours only");
            OMG_REQUIRE(peer_rank >= 0 && peer_rank < dd->n_ranks && count == 1 + 3 * int(dd->lv.size()), "bad peer rank / handle count");
            auto &P = dd->peer[peer_rank == dd->rank - 1 ? 0 : 1];
            OMG_REQUIRE(P.mapped.empty(), "peer already opened");
            std::vector<void *> own, bufs;
            std::vector<size_t> shift;
            sd_own_buffers(dd, own, shift);                  // (the neighbour's vectors sit in their allocations as mine do)
            std::vector<std::pair<std::array<char, 64>, void *>> opened;
            for (int i = 0; i < count; ++i) {
                std::array<char, 64> key;
                std::memcpy(key.data(), static_cast<const char *>(handles64) + 64 * i, 64);
                void *base = nullptr;
                for (const auto &o : opened)
                    if (o.first == key) base = o.second;
                if (!base) {
                    hipIpcMemHandle_t h;
                    std::memcpy(&h, key.data(), 64);
                    OMG_HIP(hipIpcOpenMemHandle(&base, h, hipIpcMemLazyEnablePeerAccess));
                    P.mapped.push_back(base);
                    opened.emplace_back(key, base);
                }
                bufs.push_back(static_cast<char *>(base) + DEVBUF_SLACK + shift[size_t(i)]);
            }
            sd_attach(dd, peer_rank, bufs);
        });
    });
}

int omg_sdist_p2p_local(omg_sdist *d, omg_sdist *other) {
    return guarded([&] {
        OMG_REQUIRE(d && other && bool(d->f) == bool(other->f), "null argument / mixed dtypes");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            SDist<V> *oo;
            if constexpr (std::is_same<V, double>::value) oo = other->d.get();
            else oo = other->f.get();
            OMG_REQUIRE(oo && oo->n_ranks == dd->n_ranks && oo->lv.size() == dd->lv.size(), "not another rank of the same decomposition");
            std::vector<void *> bufs;
            std::vector<size_t> shift;
            sd_own_buffers(oo, bufs, shift);
            sd_attach(dd, oo->rank, bufs);
        });
    });
}

int omg_sdist_p2p_enable(omg_sdist *d, int mode) {
    return guarded([&] {
        OMG_REQUIRE(mode == 0 || mode == 1, "bad argument");
        with(d, [&](auto *dd) {
            if (mode) {
                std::vector<void *> bufs;
                std::vector<size_t> shift;
                sd_own_buffers(dd, bufs, shift);                // (my own flag words exist)
                if (dd->rank > 0) OMG_REQUIRE(dd->peer[0].flags, "peer mode: rank - 1's buffers have not been opened");
                if (dd->rank + 1 < dd->n_ranks) OMG_REQUIRE(dd->peer[1].flags, "peer mode: rank + 1's buffers have not been opened");
                if (const char *e = getenv("OMG_P2P_SPIN")) dd->p2p_spin = uint32_t(std::max(1l, atol(e)));
            }
            OMG_HIP(hipStreamSynchronize(dd->stream));
            dd->p2p = mode;
        });
    });
}

/* bit 0: a wait for a neighbour's flag gave up since the last call (the results since then are not to be used) */
int omg_sdist_p2p_status(omg_sdist *d, unsigned *status) {
    return guarded([&] {
        OMG_REQUIRE(status, "null argument");
        with(d, [&](auto *dd) {
            uint32_t v = 0;
            if (dd->pflags.p) {
                OMG_HIP(hipMemcpyAsync(&v, dd->pflags.p + 8, 4, hipMemcpyDeviceToHost, dd->stream));
                OMG_HIP(hipMemsetAsync(dd->pflags.p + 8, 0, 4, dd->stream));
                OMG_HIP(hipStreamSynchronize(dd->stream));
            }
            *status = v;
        });
    });
}

/* All ranks of a decomposition in ONE process on one GPU, device copies in place of the RCCL exchanges (verification
 * of the schedule without several GPUs).  The ranks run on rank 0's stream; the neighbours' coefficient rows are
 * copied here. */
int omg_sdist_group_create(int n, omg_sdist **ranks, omg_sdist_group **out) {
    return guarded([&] {
        OMG_REQUIRE(n >= 1 && ranks && out, "bad argument");
        std::unique_ptr<omg_sdist_group> g(new omg_sdist_group);
        for (int r = 0; r < n; ++r) {
            OMG_REQUIRE(ranks[r] && (ranks[r]->d || ranks[r]->f) && bool(ranks[r]->f) == bool(ranks[0]->f), "null rank / mixed dtypes");
            g->ranks.push_back(ranks[r]);
        }
        with(g->ranks[0], [&](auto *first) {
            using V = value_of<decltype(first)>;
            SExchange<V> ex;
            ex.loopback = true;
            for (int r = 0; r < n; ++r) {
                SDist<V> *dd;
                if constexpr (std::is_same<V, double>::value) dd = g->ranks[size_t(r)]->d.get();
                else dd = g->ranks[size_t(r)]->f.get();
                OMG_REQUIRE(dd->rank == r && dd->n_ranks == n && dd->lv.size() == first->lv.size(), "ranks must be 0 .. n-1 of an n-rank decomposition");
                OMG_HIP(hipStreamSynchronize(dd->stream));
                dd->stream = first->own;
                ex.ranks.push_back(dd);
            }
            ex.exchange_rows();
        });
        *out = g.release();
    });
}

int omg_sdist_group_destroy(omg_sdist_group *g) {
    delete g;
    return OMG_OK;
}

int omg_sdist_group_cycles(omg_sdist_group *g, int pre, int post, int n_cycles, double *norms) {
    return guarded([&] {
        OMG_REQUIRE(g && !g->ranks.empty() && pre >= 0 && post >= 0 && n_cycles >= 0, "bad argument");
        with(g->ranks[0], [&](auto *first) {
            using V = value_of<decltype(first)>;
            SExchange<V> ex;
            ex.loopback = true;
            for (omg_sdist *r : g->ranks) {
                if constexpr (std::is_same<V, double>::value) ex.ranks.push_back(r->d.get());
                else ex.ranks.push_back(r->f.get());
            }
            ex.run(pre, post, n_cycles, norms);
        });
    });
}

}  // extern "C"
