// Multi-GPU V-cycle: 1-D slab decomposition, one process per GPU, RCCL over xGMI.
//
// Each rank owns a contiguous range of rows of every level.  Its local operator A_l has the
// owned columns first (local numbering) and the remote columns it touches — the HALO — behind
// them; x_l is stored as [owned | halo].  The hot path of a cycle is the single-GPU one
// (csr_kernels.hip) plus three exchanges:
//   * after every smoother set and after prolongation: boundary values of x_l to the slab
//     neighbours (pack kernel -> grouped ncclSend/ncclRecv straight into the halo region);
//   * before the coarse solve: all-gather of the coarsest right-hand side, then every rank
//     applies its own rows of the replicated inverse;
//   * for the residual norm: one 8-byte all-reduce.
// Restriction and prolongation are local: slabs are cut on aggregate boundaries (the host
// checks it).  Rows of one smoother set are mutually uncoupled, so the iterate does not
// depend on the number of ranks.
//
// The same schedule runs over a LOOPBACK group — several ranks living in one process on one
// GPU, halos moved with device-to-device copies — which is how the distributed algorithm is
// verified on a single-GPU box (tests/test_gpu_dist.py); only the thin RCCL calls differ.

#include <algorithm>
#include <array>
#include <cmath>
#include <cstring>
#include <memory>
#include <type_traits>

#include "common.h"
#include "rccl_dyn.h"

namespace omg {
Rccl g_rccl;
namespace {

template <typename V>
struct DLevel {
    int64_t n_loc = 0, n_halo = 0;
    DevCsrT<V> A, R, P;
    Ordering ord;
    DevBuf<int32_t> perm;
    DevBuf<int32_t> r_out;                     // restriction row -> slot in the next level's ordering
    DevBuf<V> x, tmp, b, r;
    DevBuf<V> diag;                            // owned rows' diagonal, this level's ordering (levels entered with a zero iterate)
    DevBuf<double> partials, nat;              // block sums of squares; double staging for host I/O
    V *xp = nullptr, *tp = nullptr;
    // halo plan
    std::vector<int> peers;
    std::vector<int64_t> send_off, recv_off;   // per peer, size peers + 1
    DevBuf<int32_t> send_idx;                  // rows (this level's ordering) to send, all peers
    DevBuf<V> send_buf;
    // prolongation as a scatter over R's row patterns (csrc/common.h ROW_SCATTER) instead of a
    // pass over P; with (boundary, interior) set pairs R's rows are then split into [first
    // coarse planes | interior | last coarse planes] so that the boundary part still goes first
    bool scatter_prolong = false;
    int set_group = 1;                         // 2: sets are (boundary, interior) pairs
    std::vector<int> entry_group;              // per plan entry: colour it carries, -1 = any
    std::vector<int64_t> send_start;           // per entry: first row in x if contiguous, else -1
};

__global__ void sqrt_kernel(double *v) { *v = sqrt(*v); }
__global__ void sqrt_array_kernel(const double *v, double *out, int n) {
    for (int i = threadIdx.x; i < n; i += blockDim.x) out[i] = sqrt(v[i]);
}

// One rank's slab, levels stored and computed in V (double: the reference's precision; float:
// BASELINE configs[4] — halo messages are then half the bytes too).
template <typename V>
struct Dist {
    using value_type = V;
    int rank = 0, n_ranks = 1;
    std::vector<DLevel<V>> lv;
    int smoother = OMG_SMOOTH_GS_COLOUR;
    double omega = 1.0;
    hipStream_t own = nullptr, stream = nullptr;
    // coarsest level: replicated inverse, gathered right-hand side
    int64_t n_coarse = 0, coarse_lo = 0;
    std::vector<int64_t> coarse_counts;        // rows per rank at the coarsest level
    CoarseSolver<V> coarse;                    // replicated direct solver of the last distributed level
    bool have_coarse = false;
    DevBuf<V> coarse_rhs, coarse_sol;
    DevBuf<double> tail_rhs, tail_sol;         // the tail hierarchy's device boundary is double
    // Optional replicated TAIL: instead of one direct solve, every rank runs the levels below
    // the last distributed one as an ordinary single-GPU hierarchy on the gathered right-hand
    // side (no communication down there; the coarse levels of a slab decomposition are pure
    // exchange latency otherwise).  Borrowed, owned by the caller.
    omg_hierarchy *tail = nullptr;
    DevBuf<double> sumsq;                      // device scalar
    DevBuf<double> norms;                      // omg_dist_cycles: one norm per cycle of the batch
    DevBuf<double> batch_partials, batch_sums; // ... block partials / local sums of squares of up to 64 deferred norms
    ncclComm_t comm = nullptr;
    // RCCL calls go to a second stream so that an exchange can overlap the interior rows of
    // a (boundary, interior) set pair; two events order it against the compute stream.
    hipStream_t cstream = nullptr;
    hipEvent_t ev_go = nullptr, ev_done = nullptr;
    bool halo_dirty = true;
    bool loaded = false;

    Dist() = default;
    Dist(const Dist &) = delete;
    Dist &operator=(const Dist &) = delete;
    ~Dist() {
        if (comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(comm);
        if (ev_go) (void)hipEventDestroy(ev_go);
        if (ev_done) (void)hipEventDestroy(ev_done);
        if (cstream) (void)hipStreamDestroy(cstream);
        if (own) (void)hipStreamDestroy(own);
    }
};

}  // namespace
}  // namespace omg

using namespace omg;

struct omg_dist {
    std::unique_ptr<omg::Dist<double>> d;
    std::unique_ptr<omg::Dist<float>> f;
};

struct omg_dist_group {
    std::vector<omg_dist *> ranks;             // borrowed; all on one device, one shared stream, one dtype
};

namespace omg {
namespace {

template <typename V>
std::unique_ptr<Dist<V>> create(int rank, int n_ranks, int n_levels, const omg_dist_level *lv,
                                const omg_csr *coarse_global, const int64_t *coarse_counts, int smoother,
                                double omega) {
    using D = Dist<V>;
    OMG_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks, "bad rank / n_ranks");
    OMG_REQUIRE(n_levels >= 1 && lv && coarse_counts, "null argument");
    require_device();
    std::unique_ptr<D> d(new D);
    d->rank = rank;
    d->n_ranks = n_ranks;
    d->smoother = smoother;
    d->omega = omega;
    OMG_HIP(hipStreamCreateWithFlags(&d->own, hipStreamNonBlocking));
    d->stream = d->own;
    OMG_HIP(hipStreamCreateWithFlags(&d->cstream, hipStreamNonBlocking));
    OMG_HIP(hipEventCreateWithFlags(&d->ev_go, hipEventDisableTiming));
    OMG_HIP(hipEventCreateWithFlags(&d->ev_done, hipEventDisableTiming));
    d->lv.resize(n_levels);
    d->sumsq.alloc(1);
    for (int l = 0; l < n_levels; ++l) {
        const omg_dist_level &in = lv[l];
        DLevel<V> &L = d->lv[l];
        validate_csr(in.A, ("A[" + std::to_string(l) + "]").c_str());
        L.n_loc = in.A.n_rows;
        L.n_halo = in.n_halo;
        OMG_REQUIRE(in.A.n_cols == L.n_loc + L.n_halo, "local operator must have n_loc + n_halo columns");
        const bool last = l + 1 == n_levels;
        if (!last && in.keys) L.ord = ordering_from_keys(in.keys, L.n_loc, in.n_sets);
        else { L.ord.identity = true; L.ord.sets = {0, L.n_loc}; }
    }
    for (int l = 0; l < n_levels; ++l) {
        const omg_dist_level &in = lv[l];
        DLevel<V> &L = d->lv[l];
        const bool last = l + 1 == n_levels;
        const bool id = L.ord.identity;
        {
            HostCsr Ap = permute_csr(in.A, id ? nullptr : L.ord.perm.data(), id ? nullptr : L.ord.inv.data(), L.n_loc);
            L.A.upload(Ap, L.ord.sets, d->stream);
            if (l >= 1 && !last) {
                // the diagonal as the sweeps form it (hierarchy.hip): the restriction into this level
                // applies its first relaxation of a zero iterate
                L.diag.alloc(std::max<int64_t>(L.n_loc, 1));
                launch_diagonal(L.A, L.diag.p, d->stream);
            }
        }
        if (!id) { L.perm.alloc(L.n_loc); L.perm.upload(L.ord.perm.data(), L.n_loc, d->stream); }
        if (l == 0) L.nat.alloc(std::max<int64_t>(L.n_loc, 1));
        if (!last) {
            // smoothed level: every owned row needs a diagonal
            for (int64_t i = 0; i < L.n_loc; ++i) {
                bool have = false;
                for (int32_t p = in.A.indptr[i]; p < in.A.indptr[i + 1]; ++p) if (in.A.indices[p] == i) { have = true; break; }
                if (!have) throw Error(OMG_ERR_NO_DIAGONAL, "level " + std::to_string(l) + ": local row " + std::to_string(i) + " has no diagonal entry");
            }
            validate_csr(in.R, ("R[" + std::to_string(l) + "]").c_str());
            const DLevel<V> &C = d->lv[l + 1];
            OMG_REQUIRE(in.R.n_rows == C.n_loc && in.R.n_cols == L.n_loc,
                        "R[l] must be (coarse owned rows) x (fine owned rows): slabs must be cut on aggregate boundaries");
            HostCsr Rp = permute_csr(in.R, C.ord.identity ? nullptr : C.ord.perm.data(), id ? nullptr : L.ord.inv.data());
            HostCsr Pt = transpose_csr(Rp);
            // R's rows as [aggregates holding boundary rows | interior | ... boundary]: possible when
            // those aggregates are a prefix and a suffix of the natural coarse order (slabs: the
            // first and last coarse planes); needed only where the fine sets are paired
            std::vector<int64_t> rsets;
            bool scatter_ok = true;
            {
                const char *e = getenv("OMG_PROLONG_SCATTER");
                scatter_ok = !(e && e[0] == '0');
                std::vector<char> seen(size_t(L.n_loc), 0);                // aggregation: no shared columns
                for (int64_t p = 0; scatter_ok && p < in.R.nnz; ++p) {
                    const int32_t c = in.R.indices[p];
                    if (seen[c]) scatter_ok = false;
                    seen[c] = 1;
                }
                if (scatter_ok && in.set_group == 2 && in.keys) {
                    const int64_t nc = in.R.n_rows;
                    std::vector<char> bnd(size_t(nc), 0);
                    for (int64_t j = 0; j < nc; ++j)
                        for (int32_t p = in.R.indptr[j]; p < in.R.indptr[j + 1]; ++p)
                            if (in.keys[in.R.indices[p]] % 2 == 0) bnd[j] = 1;   // key = 2 colour + interior
                    int64_t a = 0, b = nc;
                    while (a < nc && bnd[a]) ++a;
                    while (b > a && bnd[b - 1]) --b;
                    for (int64_t j = a; scatter_ok && j < b; ++j) if (bnd[j]) scatter_ok = false;
                    if (scatter_ok) rsets = {0, a, b, nc};
                }
            }
            if (C.ord.identity) {
                L.R.upload(Rp, rsets, d->stream);
            } else {
                // rows in natural coarse order + output map: dense gathers (see hierarchy.hip)
                HostCsr Rn = permute_csr(in.R, nullptr, id ? nullptr : L.ord.inv.data());
                L.R.upload(Rn, rsets, d->stream);
                L.r_out.alloc(C.ord.inv.size());
                L.r_out.upload(C.ord.inv.data(), C.ord.inv.size(), d->stream);
                OMG_HIP(hipStreamSynchronize(d->stream));
            }
            L.scatter_prolong = scatter_ok && L.R.all_pattern();
            // prolongation rows carry the fine level's sets: with (boundary, interior) pairs the
            // boundary rows are corrected first and travel while the interior ones are corrected
            L.P.upload(Pt, L.ord.sets, d->stream);
            L.r.alloc(std::max<int64_t>(L.n_loc, 1));
            if (smoother == OMG_SMOOTH_JACOBI) L.tmp.alloc(std::max<int64_t>(L.n_loc + L.n_halo, 1));
            L.partials.alloc(L.A.n_blocks() + SUM_FOLD);
        }
        L.x.alloc(std::max<int64_t>(L.n_loc + L.n_halo, 1));
        L.x.zero(d->stream);
        if (L.tmp.p) L.tmp.zero(d->stream);
        L.b.alloc(std::max<int64_t>(L.n_loc, 1));
        L.xp = L.x.p;
        L.tp = L.tmp.p;
        // halo plan
        OMG_REQUIRE(in.n_peers >= 0, "negative peer count");
        L.set_group = (in.set_group == 2 && !last && in.keys) ? 2 : 1;
        OMG_REQUIRE(L.set_group == 1 || in.n_sets % 2 == 0, "paired sets need an even set count");
        L.peers.assign(in.peers, in.peers + in.n_peers);
        L.send_off.assign(in.send_off, in.send_off + in.n_peers + 1);
        L.recv_off.assign(in.recv_off, in.recv_off + in.n_peers + 1);
        OMG_REQUIRE(in.n_peers == 0 || L.recv_off.back() == L.n_halo, "recv offsets do not cover the halo");
        const int64_t n_send = in.n_peers ? L.send_off.back() : 0;
        std::vector<int32_t> idx(n_send);
        for (int64_t k = 0; k < n_send; ++k) {
            const int32_t i = in.send_idx[k];
            OMG_REQUIRE(i >= 0 && i < L.n_loc, "send index out of range");
            idx[k] = id ? i : L.ord.inv[i];
        }
        // Per entry: its message group, and where its rows start in x when they happen to be
        // one ascending run of this level's ordering — a slab's first / last plane inside a
        // colour is — so that it can be sent straight from x without a pack kernel.
        L.entry_group.assign(in.n_peers, -1);
        if (in.entry_group) L.entry_group.assign(in.entry_group, in.entry_group + in.n_peers);
        L.send_start.assign(in.n_peers, -1);
        for (int e = 0; e < in.n_peers; ++e) {
            const int64_t a = L.send_off[e], b = L.send_off[e + 1];
            bool run = b > a;
            for (int64_t k = a + 1; k < b && run; ++k) run = idx[k] == idx[k - 1] + 1;
            if (run) L.send_start[e] = idx[a];
        }
        L.send_idx.alloc(std::max<int64_t>(n_send, 1));
        L.send_idx.upload(idx.data(), n_send, d->stream);
        L.send_buf.alloc(std::max<int64_t>(n_send, 1));
        OMG_HIP(hipStreamSynchronize(d->stream));
    }
    // coarsest distributed level: either invert the replicated global operator here (kept
    // whole, n_L is small) or, with coarse_global == NULL, wait for omg_dist_set_tail
    {
        d->coarse_counts.assign(coarse_counts, coarse_counts + n_ranks);
        int64_t lo = 0, tot = 0;
        for (int q = 0; q < n_ranks; ++q) { if (q < rank) lo += coarse_counts[q]; tot += coarse_counts[q]; }
        d->n_coarse = tot;
        OMG_REQUIRE(coarse_counts[rank] == d->lv.back().n_loc, "coarse row count of this rank differs from its level");
        d->coarse_lo = lo;
        if (coarse_global) {
            validate_csr(*coarse_global, "coarse_global");
            OMG_REQUIRE(coarse_global->n_rows == coarse_global->n_cols && coarse_global->n_rows == tot,
                        "coarse operator must be square and match the coarse row counts");
            // factored in double whatever V is; the same solver (explicit inverse, or substructuring
            // along the band for big operators) as a single-GPU hierarchy builds: same bits
            HostCsr Gh = permute_csr(*coarse_global, nullptr, nullptr);
            d->coarse.build(Gh, d->stream);
            d->have_coarse = true;
        }
        d->coarse_rhs.alloc(std::max<int64_t>(d->n_coarse, 1));
        d->coarse_sol.alloc(std::max<int64_t>(d->n_coarse, 1));
        d->tail_rhs.alloc(std::max<int64_t>(d->n_coarse, 1));
        d->tail_sol.alloc(std::max<int64_t>(d->n_coarse, 1));
    }
    OMG_HIP(hipStreamSynchronize(d->stream));
    return d;
}

// ---- the SPMD schedule over the ranks that live in this process ------------------------------
template <typename V>
struct Runner {
    using D = Dist<V>;
    using RowArgs = RowArgsT<V>;
    std::vector<D *> rs;
    bool rccl;     // true: exactly one local rank, peers reached through RCCL

    D *find(int rank) const {
        for (D *d : rs) if (d->rank == rank) return d;
        throw Error(OMG_ERR_INVALID, "loopback group lacks rank " + std::to_string(rank));
    }

    // Does level l's exchange run on the second (comm) stream so that it overlaps the
    // interior launch?  Only where that pays: each cross-stream event pair costs a few us of
    // bubble (measured: 25 of them slowed a one-rank 256^3 cycle from 1.53 to 1.80 ms), so
    // small levels, single-rank runs and loopback groups keep everything on one stream.
    static int64_t overlap_min_rows() {
        const char *e = getenv("OMG_OVERLAP_MIN_ROWS");
        return e ? atoll(e) : (int64_t(1) << 19);
    }
    // OMG_FORCE_OVERLAP=1 (tests): take the two-stream schedule in a loopback group too, so
    // that its ordering is exercised on one GPU; all loopback ranks then share rank 0's
    // comm stream the way they share its compute stream.
    static bool force_overlap() {
        const char *e = getenv("OMG_FORCE_OVERLAP");
        return e && e[0] == '1';
    }
    bool on_comm_stream(int l) const {
        if (!rccl && !force_overlap()) return false;
        const DLevel<V> &L = rs[0]->lv[l];
        return L.set_group == 2 && !L.peers.empty() && L.n_loc >= overlap_min_rows();
    }
    hipStream_t comm_stream_of(D *d) const { return rccl ? d->cstream : rs[0]->cstream; }
    hipStream_t cs(D *d, int l) const { return on_comm_stream(l) ? comm_stream_of(d) : d->stream; }
    // comm stream waits for everything enqueued so far on the compute stream ...
    void comm_begin(D *d, int l) {
        if (!on_comm_stream(l)) return;
        OMG_HIP(hipEventRecord(d->ev_go, d->stream));
        OMG_HIP(hipStreamWaitEvent(comm_stream_of(d), d->ev_go, 0));
    }
    // ... and the compute stream waits for the communication enqueued since.
    void comm_end(D *d, int l) {
        if (!on_comm_stream(l)) return;
        OMG_HIP(hipEventRecord(d->ev_done, comm_stream_of(d)));
        OMG_HIP(hipStreamWaitEvent(d->stream, d->ev_done, 0));
    }

    void pack(D *d, int l) {
        DLevel<V> &L = d->lv[l];
        const int64_t n = L.peers.empty() ? 0 : L.send_off.back();
        if (n) launch_gather(L.xp, L.send_idx.p, L.send_buf.p, n, cs(d, l));
    }

    // boundary values of x_l -> neighbours' halo regions.  exchange_start enqueues the
    // transfer behind the work already on the compute stream; work enqueued on the compute
    // stream between start and finish overlaps it (it must not read halo entries nor write
    // rows that are being sent); exchange_finish makes the compute stream wait for it.
    // group >= 0: only the entries that carry that colour (after relaxing it nothing else has
    // changed); group < 0: everything.
    void exchange(int l, int group = -1) {
        exchange_start(l, group);
        exchange_finish(l);
    }

    void exchange_finish(int l) {
        for (D *d : rs) comm_end(d, l);
    }

    static bool selected(const DLevel<V> &L, size_t e, int group) {
        return group < 0 || L.entry_group[e] < 0 || L.entry_group[e] == group;
    }

    // Where entry e of level L is sent from: straight out of x when its rows are one run of
    // the ordering, the packed copy otherwise.
    static const V *send_ptr(const DLevel<V> &L, size_t e) {
        return L.send_start[e] >= 0 ? L.xp + L.send_start[e] : L.send_buf.p + L.send_off[e];
    }

    void exchange_start(int l, int group = -1) {
        for (D *d : rs) comm_begin(d, l);
        for (D *d : rs) {                        // pack only what cannot be sent in place
            DLevel<V> &L = d->lv[l];
            bool need = false;
            for (size_t e = 0; e < L.peers.size(); ++e)
                need = need || (selected(L, e, group) && L.send_start[e] < 0 && L.send_off[e + 1] > L.send_off[e]);
            if (need) pack(d, l);
        }
        if (rccl) {
            D *d = rs[0];
            DLevel<V> &L = d->lv[l];
            if (L.peers.empty()) return;
            hipStream_t s = cs(d, l);
            OMG_NCCL(g_rccl.GroupStart());
            for (size_t k = 0; k < L.peers.size(); ++k) {
                if (!selected(L, k, group)) continue;
                const int64_t ns = L.send_off[k + 1] - L.send_off[k], nr = L.recv_off[k + 1] - L.recv_off[k];
                if (ns) OMG_NCCL(g_rccl.Send(send_ptr(L, k), (size_t)ns, NcclType<V>::value, L.peers[k], d->comm, s));
                if (nr) OMG_NCCL(g_rccl.Recv(L.xp + L.n_loc + L.recv_off[k], (size_t)nr, NcclType<V>::value, L.peers[k], d->comm, s));
            }
            OMG_NCCL(g_rccl.GroupEnd());
        } else {
            for (D *d : rs) {
                DLevel<V> &L = d->lv[l];
                for (size_t k = 0; k < L.peers.size(); ++k) {
                    if (!selected(L, k, group)) continue;
                    D *p = find(L.peers[k]);
                    DLevel<V> &PL = p->lv[l];
                    // the peer's entry towards this rank with the same group tag
                    size_t j = 0;
                    while (j < PL.peers.size() && !(PL.peers[j] == d->rank && PL.entry_group[j] == L.entry_group[k])) ++j;
                    OMG_REQUIRE(j < PL.peers.size(), "halo plan is not symmetric");
                    const int64_t nr = L.recv_off[k + 1] - L.recv_off[k];
                    OMG_REQUIRE(nr == PL.send_off[j + 1] - PL.send_off[j], "send/recv counts differ");
                    if (nr) OMG_HIP(hipMemcpyAsync(L.xp + L.n_loc + L.recv_off[k], send_ptr(PL, j),
                                                   nr * sizeof(V), hipMemcpyDeviceToDevice, cs(d, l)));
                }
            }
        }
    }

    // fuse: 0 none, 1 the last set launch also stores its rows' residual into r, 2 it adds
    // their squared residual to the block partials (RowMode ROW_GS_RES / ROW_GS_NORM: exact,
    // because a relaxed row's residual involves no other row of its own set).  Returns true
    // when that happened.
    // Can the first launches of a cycle also finish the PREVIOUS cycle's norm (RowMode ROW_GS_PRENORM /
    // ROW_JACOBI_PRENORM, hierarchy.hip can_prenorm)?  Two colours (each possibly a (boundary,
    // interior) pair), the second one's squares left by the fused post-smoothing launches; or Jacobi.
    bool can_prenorm(int pre, int post) const {
        if (pre <= 0 || rs[0]->lv.size() < 2) return false;
        if (rs[0]->smoother == OMG_SMOOTH_JACOBI) return true;
        if (rs[0]->smoother != OMG_SMOOTH_GS_COLOUR || post <= 0) return false;
        const DLevel<V> &L = rs[0]->lv[0];
        return (int)L.A.n_sets() == 2 * L.set_group;
    }

    // the slot-th block-partials array of rank d's batch buffer (level 0)
    static double *slot_of(D *d, int slot) { return d->batch_partials.p + size_t(slot) * size_t(d->lv[0].A.n_blocks()); }

    // Can the restriction into level l also apply its first smoothing launches (hierarchy.hip
    // first_sweep_in_restrict)?  First colour of a colour ordering, or a Jacobi sweep.
    bool first_sweep_in_restrict(int l, int pre) const {
        const char *e = getenv("OMG_NO_FIRST_SWEEP");
        if (pre <= 0 || (e && e[0] == '1')) return false;
        for (D *d : rs) if (!d->lv[l].diag.p) return false;
        if (rs[0]->smoother == OMG_SMOOTH_JACOBI) return true;
        return rs[0]->smoother == OMG_SMOOTH_GS_COLOUR && (int)rs[0]->lv[l].A.n_sets() >= rs[0]->lv[l].set_group;
    }

    // pre_slot >= 0: level 0, first sweep of the cycle: the first colour's launches run in the PRENORM
    // mode and leave the squares of their rows' residuals — the PREVIOUS cycle's — in that slot of the
    // batch buffer; post_slot >= 0: where the fused post-smoothing launches (fuse == 2) leave theirs.
    // first_done: the first colour's launches (or the first Jacobi sweep) of the first iteration have
    // been applied by the restriction: only their exchange is left
    bool smooth(int l, int iterations, int fuse = 0, int pre_slot = -1, bool first_done = false, int post_slot = -1) {
        bool fused = false;
        for (int it = 0; it < iterations; ++it) {
            const bool prenorm = pre_slot >= 0 && it == 0;
            if (rs[0]->smoother == OMG_SMOOTH_JACOBI) {
                if (it == 0 && first_done) { exchange(l); continue; }
                for (D *d : rs) {
                    DLevel<V> &L = d->lv[l];
                    RowArgs a;
                    a.x = L.xp; a.b = L.b.p; a.y = L.tp; a.omega = d->omega;
                    if (prenorm) a.partials = slot_of(d, pre_slot);
                    launch_rows(L.A, prenorm ? ROW_JACOBI_PRENORM : ROW_JACOBI, -1, a, d->stream);
                    std::swap(L.xp, L.tp);
                }
                exchange(l);
            } else {
                const int n_sets = (int)rs[0]->lv[l].A.n_sets();
                const int grp = rs[0]->lv[l].set_group;
                for (D *d : rs)
                    OMG_REQUIRE((int)d->lv[l].A.n_sets() == n_sets && d->lv[l].set_group == grp,
                                "ranks disagree on the smoother sets of a level");
                auto sweep_set = [&](int s, bool last_group) {
                    for (D *d : rs) {
                        DLevel<V> &L = d->lv[l];
                        RowArgs a;
                        a.x = L.xp; a.b = L.b.p; a.y = L.xp;
                        if (prenorm && s < grp) {             // first colour: sees the iterate the previous cycle left
                            a.partials = slot_of(d, pre_slot);
                            launch_rows(L.A, ROW_GS_PRENORM, s, a, d->stream);
                        } else if (fuse && it + 1 == iterations && last_group) {
                            a.zero = L.r.p;
                            a.partials = (fuse == 2 && post_slot >= 0) ? slot_of(d, post_slot) : L.partials.p;
                            launch_rows(L.A, fuse == 1 ? ROW_GS_RES : ROW_GS_NORM, s, a, d->stream);
                            fused = true;
                        } else {
                            launch_rows(L.A, ROW_GS, s, a, d->stream);
                        }
                    }
                };
                for (int s = 0; s < n_sets; s += grp) {
                    const bool last_group = s + grp == n_sets;
                    if (it == 0 && s == 0 && first_done) { exchange(l, 0); continue; }   // relaxed by the restriction
                    sweep_set(s, last_group);            // plain set, or the BOUNDARY rows of a colour
                    exchange_start(l, s / grp);          // only this colour's values have changed
                    // interior rows of the same colour: touch no halo entry, are not sent
                    if (grp == 2) sweep_set(s + 1, last_group);
                    exchange_finish(l);
                }
            }
        }
        return fused;
    }

    void coarse(int pre, int post) {
        // all-gather the coarsest right-hand side, then each rank applies ITS rows of the
        // inverse — or runs the replicated tail hierarchy and keeps its slice of the result
        if (rccl) {
            D *d = rs[0];
            DLevel<V> &L = d->lv.back();
            bool equal = true;
            for (int64_t c : d->coarse_counts) equal = equal && c == d->coarse_counts[0];
            if (d->n_ranks == 1) {
                OMG_HIP(hipMemcpyAsync(d->coarse_rhs.p, L.b.p, L.n_loc * sizeof(V), hipMemcpyDeviceToDevice, d->stream));
            } else if (equal) {
                OMG_NCCL(g_rccl.AllGather(L.b.p, d->coarse_rhs.p, (size_t)L.n_loc, NcclType<V>::value, d->comm, d->stream));
            } else {
                OMG_NCCL(g_rccl.GroupStart());
                int64_t off = 0;
                for (int q = 0; q < d->n_ranks; ++q) {
                    if (L.n_loc) OMG_NCCL(g_rccl.Send(L.b.p, (size_t)L.n_loc, NcclType<V>::value, q, d->comm, d->stream));
                    if (d->coarse_counts[q]) OMG_NCCL(g_rccl.Recv(d->coarse_rhs.p + off, (size_t)d->coarse_counts[q], NcclType<V>::value, q, d->comm, d->stream));
                    off += d->coarse_counts[q];
                }
                OMG_NCCL(g_rccl.GroupEnd());
            }
        } else {
            for (D *d : rs)
                for (D *p : rs) {
                    DLevel<V> &PL = p->lv.back();
                    if (PL.n_loc) OMG_HIP(hipMemcpyAsync(d->coarse_rhs.p + p->coarse_lo, PL.b.p, PL.n_loc * sizeof(V),
                                                         hipMemcpyDeviceToDevice, d->stream));
                }
        }
        for (D *d : rs) {
            DLevel<V> &L = d->lv.back();
            if (d->tail) {
                // the tail's device boundary is double (it converts to its own dtype inside)
                const double *rhs64;
                double *sol64;
                if constexpr (std::is_same<V, double>::value) {
                    rhs64 = d->coarse_rhs.p;
                    sol64 = d->coarse_sol.p;
                } else {
                    launch_gather<V, double>(d->coarse_rhs.p, nullptr, d->tail_rhs.p, d->n_coarse, d->stream);
                    rhs64 = d->tail_rhs.p;
                    sol64 = d->tail_sol.p;
                }
                const int rc = omg_hierarchy_cycle_dev(d->tail, rhs64, sol64, pre, post, d->stream);
                if (rc != OMG_OK) throw Error(rc, std::string("tail hierarchy: ") + omg_last_error());
                if (L.n_loc) launch_gather<double, V>(sol64 + d->coarse_lo, nullptr, L.xp, L.n_loc, d->stream);
            } else {
                OMG_REQUIRE(d->have_coarse, "no coarse solver: pass coarse_global or call omg_dist_set_tail");
                if (d->coarse.P == 1) {          // explicit inverse: this rank's rows of it
                    launch_dense_gemv_rows<V>(d->coarse.inv.p + d->coarse_lo * d->n_coarse, d->coarse_rhs.p, L.xp, L.n_loc,
                                              d->n_coarse, d->stream);
                } else {                         // substructured: the whole (replicated) solve, keep this rank's slice
                    d->coarse.solve(d->coarse_rhs.p, d->coarse_sol.p, d->stream);
                    if (L.n_loc) OMG_HIP(hipMemcpyAsync(L.xp, d->coarse_sol.p + d->coarse_lo, L.n_loc * sizeof(V),
                                                        hipMemcpyDeviceToDevice, d->stream));
                }
            }
        }
    }

    bool cycle(int l, int pre, int post, bool want_norm = false, int pre_slot = -1, bool first_done = false,
               int post_slot = -1) {
        const int last = (int)rs[0]->lv.size() - 1;
        if (l >= last) { coarse(pre, post); return false; }
        const bool res_done = smooth(l, pre, 1, pre_slot, first_done);
        const bool child_first = l + 1 < last && first_sweep_in_restrict(l + 1, pre);
        for (D *d : rs) {
            DLevel<V> &L = d->lv[l];
            DLevel<V> &C = d->lv[l + 1];
            RowArgs a;
            a.x = L.xp; a.b = L.b.p; a.y = L.r.p;
            const int ns = (int)L.A.n_sets();
            launch_rows_range(L.A, ROW_RESIDUAL, 0, res_done ? ns - L.set_group : ns, a, d->stream);
            RowArgs q;
            q.x = L.r.p; q.y = C.b.p; q.zero = (l + 1 < last) ? C.xp : nullptr; q.ymap = L.r_out.p;
            if (child_first) {
                q.first_diag = C.diag.p;
                q.first_jacobi = d->smoother == OMG_SMOOTH_JACOBI;
                q.first_end = int(q.first_jacobi ? C.n_loc : C.A.sets[C.set_group]);
                q.omega = d->omega;
            }
            launch_rows(L.R, ROW_SPMV, -1, q, d->stream);
            if (l + 1 < last && C.n_halo)
                OMG_HIP(hipMemsetAsync(C.xp + C.n_loc, 0, C.n_halo * sizeof(V), d->stream));
        }
        cycle(l + 1, pre, post, false, -1, child_first);
        // x_l += R^T x_{l+1} (:214, :220/:224), then everybody needs the corrected boundary values
        // step 0: everything; first 0 / 1 with step 2: the boundary / the interior part
        auto prolong_sets = [&](int first, int step) {
            for (D *d : rs) {
                DLevel<V> &L = d->lv[l];
                DLevel<V> &C = d->lv[l + 1];
                RowArgs a;
                a.x = C.xp; a.y = L.xp;
                if (L.scatter_prolong) {             // over R's rows: [boundary | interior | boundary] when split
                    a.ymap = L.r_out.p;
                    if (step == 0 || L.R.n_sets() != 3) {
                        if (step == 0 || first == 0) launch_rows(L.R, ROW_SCATTER, -1, a, d->stream);
                    } else if (first == 0) {
                        launch_rows(L.R, ROW_SCATTER, 0, a, d->stream);
                        launch_rows(L.R, ROW_SCATTER, 2, a, d->stream);
                    } else {
                        launch_rows(L.R, ROW_SCATTER, 1, a, d->stream);
                    }
                    continue;
                }
                if (step == 0) { launch_rows(L.P, ROW_AXPY, -1, a, d->stream); continue; }
                for (int s = first; s < (int)L.P.n_sets(); s += step) launch_rows(L.P, ROW_AXPY, s, a, d->stream);
            }
        };
        if (on_comm_stream(l)) {
            prolong_sets(0, 2);          // boundary rows of every colour
            exchange_start(l);
            prolong_sets(1, 2);          // interior rows, beside the exchange
            exchange_finish(l);
        } else {
            prolong_sets(0, 0);
            exchange(l);
        }
        if (post > 0) return smooth(l, post, want_norm ? 2 : 0, -1, false, post_slot);
        return false;
    }

    // The first count slots of every rank's batch buffer are complete: local sums (the additions
    // launch_sum makes), ONE reduction over the ranks for all of them, sqrt -> norms[first ...].
    void finish_batch(int count, int first) {
        if (count <= 0) return;
        for (D *d : rs) {
            const int64_t nb = d->lv[0].A.n_blocks();
            launch_sum_batch(d->batch_partials.p, nb, nb, count, d->batch_sums.p, false, d->stream);
        }
        if (rccl) {
            D *d = rs[0];
            if (d->n_ranks > 1)
                OMG_NCCL(g_rccl.AllReduce(d->batch_sums.p, d->batch_sums.p, (size_t)count, ncclDouble, ncclSum, d->comm, d->stream));
        } else if (rs.size() > 1) {
            // loopback (test path): rank order, on the host
            std::vector<double> tot(count, 0.0), v(count);
            for (D *d : rs) {
                OMG_HIP(hipMemcpyAsync(v.data(), d->batch_sums.p, count * sizeof(double), hipMemcpyDeviceToHost, d->stream));
                OMG_HIP(hipStreamSynchronize(d->stream));
                for (int i = 0; i < count; ++i) tot[i] += v[i];
            }
            for (D *d : rs) OMG_HIP(hipMemcpyAsync(d->batch_sums.p, tot.data(), count * sizeof(double), hipMemcpyHostToDevice, d->stream));
            for (D *d : rs) OMG_HIP(hipStreamSynchronize(d->stream));
        }
        for (D *d : rs)
            hipLaunchKernelGGL(sqrt_array_kernel, dim3(1), dim3(64), 0, d->stream, d->batch_sums.p, d->norms.p + first, count);
    }

    // sum over all ranks of the local sums of squares -> sqrt, left in every rank's sumsq
    void norm(bool last_set_done) {
        const bool single = rs[0]->lv.size() == 1;
        for (D *d : rs) {
            if (single) { OMG_HIP(hipMemsetAsync(d->sumsq.p, 0, sizeof(double), d->stream)); continue; }
            DLevel<V> &L = d->lv[0];
            RowArgs a;
            a.x = L.xp; a.b = L.b.p; a.partials = L.partials.p;
            const int ns = (int)L.A.n_sets();
            launch_rows_range(L.A, ROW_NORM_ONLY, 0, last_set_done ? ns - L.set_group : ns, a, d->stream);
            launch_sum(L.partials.p, L.A.n_blocks(), d->sumsq.p, d->stream);
        }
        if (single) return;
        reduce_sumsq();
        for (D *d : rs) hipLaunchKernelGGL(sqrt_kernel, dim3(1), dim3(1), 0, d->stream, d->sumsq.p);
    }

    // every rank's sumsq <- the sum over all ranks
    void reduce_sumsq() {
        if (rccl) {
            D *d = rs[0];
            if (d->n_ranks > 1)
                OMG_NCCL(g_rccl.AllReduce(d->sumsq.p, d->sumsq.p, 1, ncclDouble, ncclSum, d->comm, d->stream));
        } else if (rs.size() > 1) {
            // loopback (test path): rank order, on the host
            double tot = 0.0;
            for (D *d : rs) {
                double v = 0.0;
                OMG_HIP(hipMemcpyAsync(&v, d->sumsq.p, sizeof(double), hipMemcpyDeviceToHost, d->stream));
                OMG_HIP(hipStreamSynchronize(d->stream));
                tot += v;
            }
            for (D *d : rs) OMG_HIP(hipMemcpyAsync(d->sumsq.p, &tot, sizeof(double), hipMemcpyHostToDevice, d->stream));
            for (D *d : rs) OMG_HIP(hipStreamSynchronize(d->stream));
        }
    }

    // n cycles, every cycle's global norm computed; norms_out (host, nullable) gets all of them.
    // Where can_prenorm() holds, cycles go in chunks of up to 64: the squares of cycle k's norm are
    // finished by cycle k + 1's first launches into slot k of the batch buffer, and the chunk's
    // local sums are reduced over the ranks by ONE all-reduce instead of one per cycle (with more
    // than two ranks RCCL may then add the ranks' sums in another order than the one-value
    // all-reduce of omg_dist_cycle does: the norms are reported values, the iterate is untouched).
    void run_batch(int pre, int post, int n, double *norms_out) {
        for (D *d : rs) OMG_REQUIRE(d->loaded, "omg_dist_load has not been called");
        if (n <= 0) return;
        for (D *d : rs) if (d->norms.n < size_t(n)) d->norms.alloc(size_t(n));
        bool dirty = false;
        for (D *d : rs) dirty = dirty || d->halo_dirty;
        if (dirty) { exchange(0); for (D *d : rs) d->halo_dirty = false; }
        const bool multi = rs[0]->lv.size() > 1;
        const char *e = getenv("OMG_NO_PRENORM");
        const bool defer = multi && !(e && e[0] == '1') && can_prenorm(pre, post);
        const int CHUNK = 64;
        if (defer)
            for (D *d : rs) {
                const size_t need = size_t(CHUNK) * size_t(d->lv[0].A.n_blocks());
                if (d->batch_partials.n < need) d->batch_partials.alloc(need);
                if (d->batch_sums.n < size_t(CHUNK)) d->batch_sums.alloc(CHUNK);
            }
        for (int k0 = 0; k0 < n; k0 += CHUNK) {
            const int cnt = std::min(CHUNK, n - k0);
            for (int j = 0; j < cnt; ++j) {
                const bool last_of_chunk = j + 1 == cnt;
                const bool part = cycle(0, pre, post, multi, (defer && j > 0) ? j - 1 : -1, false,
                                        (defer && !last_of_chunk) ? j : -1);
                if (!defer || last_of_chunk) {
                    norm(part);
                    for (D *d : rs)
                        OMG_HIP(hipMemcpyAsync(d->norms.p + k0 + j, d->sumsq.p, sizeof(double), hipMemcpyDeviceToDevice, d->stream));
                }
            }
            if (defer) finish_batch(cnt - 1, k0);
        }
        D *d0 = rs[0];
        if (norms_out) OMG_HIP(hipMemcpyAsync(norms_out, d0->norms.p, size_t(n) * sizeof(double), hipMemcpyDeviceToHost, d0->stream));
        for (D *d : rs) OMG_HIP(hipStreamSynchronize(d->stream));
    }

    void run(int pre, int post, double *norm_out) {
        for (D *d : rs) OMG_REQUIRE(d->loaded, "omg_dist_load has not been called");
        bool dirty = false;
        for (D *d : rs) dirty = dirty || d->halo_dirty;
        if (dirty) { exchange(0); for (D *d : rs) d->halo_dirty = false; }
        const bool part = cycle(0, pre, post, rs[0]->lv.size() > 1);
        norm(part);
        if (norm_out) {
            D *d = rs[0];
            OMG_HIP(hipMemcpyAsync(norm_out, d->sumsq.p, sizeof(double), hipMemcpyDeviceToHost, d->stream));
            OMG_HIP(hipStreamSynchronize(d->stream));
        }
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

// Runs f(Dist<V> *) on whichever instantiation the handle holds.
template <typename F>
void with(omg_dist *d, F &&f) {
    OMG_REQUIRE(d != nullptr && (d->d || d->f), "null handle");
    if (d->f) f(d->f.get());
    else f(d->d.get());
}

template <typename HP>
using value_of = typename std::remove_pointer<HP>::type::value_type;


// ============================================================================================================
// Plane-pipelined slabs (round 3).  The runner above exchanges one boundary plane per colour after every
// smoother set: ~8 exchanges per level and cycle, each at least 10-12 us (profiles/r03_exchange_probe.txt) — more
// than a level's arithmetic on the small levels.  For the constant-coefficient grid stencils the single-GPU
// cycle runs as plane-pipelined passes (plane.hip), a rank's slab — its planes of every distributed level plus
// GHOST planes on either side, in the red-black ordering of that extended grid — runs the SAME kernel: the
// passes' overlapped ring does in z what it does in x and y, the neighbour's planes take the place of the
// redundantly relaxed ring.  Exchanges per cycle and distributed level: three ghost planes of x before each pass
// (the one before the up pass travels while the coarser levels run), two ghost planes of the right-hand side
// (levels >= 1) and of the coarse correction — contiguous runs of the plane-major vectors, no pack kernels.
// Below the last distributed level every rank all-gathers the right-hand side and runs the replicated tail
// hierarchy (as above).  The iterate has the bits of the single-GPU cycle for every number of ranks
// (tests/test_gpu_plane_dist.py: loopback groups of 2, 4, 8 slabs on one GPU against the hierarchy).
// double only; V(1,1).
constexpr int PD_GHOST = 4;                   // ghost planes on either side of a slab (three are read; even: colour parity)

struct PDLevel {
    int nx = 0, ny = 0, nzo = 0;              // cells per line, lines per plane, owned planes
    int64_t pc = 0, n_ext = 0;                // one colour's values per plane; values of the extended slab
    PlanePlan<double> plan;
    DevBuf<char> pool;                        // x, tmp, b are views into it (one allocation), or empty
    DevBuf<double> x, tmp, b;
    double *xp = nullptr, *tp = nullptr;
    DevBuf<int32_t> cmap;                     // coarse slab (natural, extended) -> slot in the next level's ordering (not the last level)
};

// ---- peer mode: the exchanges as stores into the neighbours' memory (xGMI peer mappings) ----------------------
// Flags live in each rank's own memory and are written by the others: flag (slot, from) of a rank, from = 0: by
// rank - 1, 1: by rank + 1.  Values only grow: cycle numbers (PF_DOWN/PF_UP/PF_GATHER), batch numbers (the rest).
constexpr int PD_FLAGS = 256;
constexpr int PF_DOWN = 0;                    // + level: the neighbour's down pass of cycle c has written its boundary planes here
constexpr int PF_UP = 8;                      // + level: ... its up pass
constexpr int PF_PRIME = 16;                  // the ghost planes a batch of cycles starts from have been written
constexpr int PF_READY = 17;                  // the neighbour's stream has reached the start of the batch
// gated passes (RCCL exchanges on the side stream while the pass runs): LOCAL words, raised by a one-thread launch on the
// side stream behind the exchange — the ghost planes of x for the finest level's down pass / for its up pass, of the
// correction for its up pass
constexpr int PF_GATE_X = 18, PF_GATE_XU = 19, PF_GATE_C = 20;
constexpr int PF_GATHER = 64;                 // + source rank (no direction): its planes of the tail's right-hand side
constexpr int PD_STATUS = PD_FLAGS, PD_DONE = PD_FLAGS + 1, PD_COUNT = PD_FLAGS + 8;   // local words behind the flags
inline int pf(int slot, int from) { return slot * 2 + from; }

struct PDPeer {                               // another rank's buffers as this process addresses them
    std::vector<double *> x, tmp, b;          // per distributed level
    double *full_b[2] = {nullptr, nullptr};
    uint32_t *flags = nullptr;
    std::vector<void *> mapped;               // hipIpcOpenMemHandle results (closed with the rank)
};

struct PlaneDist {
    int rank = 0, n_ranks = 1;
    int p2p = 0;                              // 0: RCCL / loopback copies; 1: peer stores, the passes wait themselves; 2: ... wait launches
    uint32_t spin = 1u << 21;
    std::vector<PDPeer> peers;                // by rank (own entry unused)
    DevBuf<uint32_t> flags;                   // PD_FLAGS flags + status + counters
    DevBuf<double> full_b2;                   // the gathered right-hand side of odd cycles (a rank may still be reading the even one)
    DevBuf<double *> ag_dst[2];               // per parity: where every rank wants my planes
    DevBuf<uint32_t *> ag_flag;
    uint32_t batch_no = 0;
    std::vector<PDLevel> lv;
    // the level below the last distributed one: every rank's planes of its right-hand side / of the correction in
    // natural order with ghost planes (cb, ce), and the whole of them for the replicated tail (full_b, full_x)
    int cnx = 0, cny = 0, cnzo = 0;
    DevBuf<double> cb, ce, full_b, full_x;
    omg_hierarchy *tail = nullptr;
    DevBuf<double> norm2, norms;              // this rank's sum of squares; the batch's norms
    DevBuf<double> nat;                       // host I/O staging (owned planes, natural order)
    hipStream_t own = nullptr, stream = nullptr;
    ncclComm_t comm = nullptr;
    // The ghost planes of x for a level's UP pass are exchanged right after its DOWN pass, on a second stream (and
    // a communicator of its own), while the coarser levels run: two events per level order the streams.
    hipStream_t side_own = nullptr, side = nullptr;
    ncclComm_t comm_side = nullptr;
    std::vector<hipEvent_t> ev_down, ev_halo;
    // Split passes (round 4): a pass of a slab level runs as TWO launches — the z chunks that hold the slab's first and
    // last planes on the side stream, followed there by the exchanges of what they produced, and the other chunks on the
    // main stream beside them — so that an exchange is on the critical path only for as long as it outlasts the inner
    // chunks.  ev_main[l][0 / 1]: the main stream has everything the down / up pass of level l reads; ev_edge[l][0 / 1]:
    // the side stream has run the pass's edge chunks and the exchanges behind them.
    std::vector<hipEvent_t> ev_main[2], ev_edge[2];
    // OMG_PDIST_SPLIT=1 (or 2: even without neighbours).  OFF by default: measured on one GPU (tools/pdist_split_cost.py,
    // profiles/r04_pdist_split_cost.txt) the two launches + their stream hand-overs cost a rank ~27 us of device time per
    // pass, ~165 us per cycle — more than the ~120 us of exchange latency and wire time they can hide at N = 8.
    bool split = [] { const char *e = getenv("OMG_PDIST_SPLIT"); return e && (e[0] == '1' || e[0] == '2'); }();
    bool x0_posted = false;                   // the ghost planes of x for the next cycle's first pass are already on their way (ev_edge[1][0] / PF_GATE_X)
    // Gated passes (round 5): the finest level's two passes run as ONE launch each whose edge chunks — the slab's first and
    // last four planes, the highest workgroup numbers — wait on a device flag for their ghost planes, while the exchange
    // that brings them runs on the side stream BESIDE the pass's inner chunks: the 3-plane exchange between up(k) and
    // down(k + 1) and the correction's exchange in front of the finest up pass leave the critical path without a second
    // launch (split passes: + 27 us each) and without peer stores.  OMG_PDIST_GATE=0: exchanges in stream order.
    // (=2: also with ONE slab — no neighbour, no exchange: what the one-launch inner + edge form costs a rank in device time)
    // Off until somebody switches it on (omg_pdist_set_gate: bench.py does after one checked cycle with it reproduces the
    // stream-ordered cycle's norm on every rank, as for peer mode; OMG_PDIST_GATE=1: from creation — the tests): whether the
    // exchange's kernels find compute units beside the spinning edge workgroups cannot be verified on a one-GPU box.
    bool gate = [] { const char *e = getenv("OMG_PDIST_GATE"); return e && (e[0] == '1' || e[0] == '2'); }();
    bool gate_forced = [] { const char *e = getenv("OMG_PDIST_GATE"); return e && e[0] == '2'; }();
    // progress of the DEVICE through a cycle, for a caller whose collective never completes (bench.py's preflight):
    // a word in pinned host memory the stream writes between the phases — (cycle << 16) | (level << 8) | phase,
    // phase 1 halo x, 2 halo b, 3 down pass, 4 halo x (for the up pass), 5 gather + tail, 6 halo of the correction,
    // 7 up pass, 8 norm; only while tracing is on (omg_pdist_trace)
    uint32_t *progress = nullptr;
    bool trace = false;
    uint32_t cycle_no = 0;
    void mark(int level, int phase) {
        if (trace && progress) (void)hipStreamWriteValue32(stream, progress, (cycle_no << 16) | (uint32_t(level) << 8) | uint32_t(phase), 0);
    }
    ~PlaneDist() {
        for (PDPeer &p : peers)
            for (void *m : p.mapped) (void)hipIpcCloseMemHandle(m);
        if (progress) (void)hipHostFree(progress);
        for (hipEvent_t e : ev_down) (void)hipEventDestroy(e);
        for (hipEvent_t e : ev_halo) (void)hipEventDestroy(e);
        for (int i = 0; i < 2; ++i) {
            for (hipEvent_t e : ev_main[i]) (void)hipEventDestroy(e);
            for (hipEvent_t e : ev_edge[i]) (void)hipEventDestroy(e);
        }
        if (comm_side) (void)g_rccl.CommDestroy(comm_side);
        if (side_own) (void)hipStreamDestroy(side_own);
        if (comm) (void)g_rccl.CommDestroy(comm);
        if (own) (void)hipStreamDestroy(own);
    }
};

// owned planes, natural order <-> the extended slab's red-black ordering
__global__ void pd_scatter_kernel(const double *__restrict__ nat, double *__restrict__ ext, int nx, int ny, int nzo, int ghost, int to_ext) {
    const int64_t n = int64_t(nx) * ny * nzo, nr = int64_t(nx) * ny * (nzo + 2 * ghost) / 2;
    for (int64_t r = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; r < n; r += (int64_t)gridDim.x * blockDim.x) {
        const int64_t i = r % nx, j = (r / nx) % ny, k = r / (int64_t(nx) * ny) + ghost;
        const int64_t e = (k * ny + j) * nx + i;                  // natural index in the extended slab
        const int64_t slot = (((i + j + k) & 1) ? nr : 0) + e / 2;
        if (to_ext) ext[slot] = nat[r];
        else const_cast<double *>(nat)[r] = ext[slot];
    }
}
__global__ void pd_add_kernel(double *acc, const double *v) { *acc += *v; }
__global__ void pd_sqrt_kernel(const double *v, double *out) { *out = sqrt(*v); }
__global__ void pd_sqrt_batch_kernel(double *v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) v[i] = sqrt(v[i]);
}

// peer mode's own launches (the passes themselves store and wait in plane.hip)
__device__ __forceinline__ void pd_spin(const uint32_t *flag, uint32_t seq, uint32_t *status, uint32_t spin) {
    for (uint32_t n = 0;; ++n) {
        if (int32_t(__hip_atomic_load(flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) - seq) >= 0) break;
        if (n >= spin) { __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); break; }
        __builtin_amdgcn_s_sleep(16);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
}
// thread i waits for flags[first + i * stride] (i < count)
__global__ void pd_wait_kernel(const uint32_t *flags, int first, int stride, int count, uint32_t seq, uint32_t *status, uint32_t spin) {
    if (int(threadIdx.x) < count) pd_spin(flags + first + int(threadIdx.x) * stride, seq, status, spin);
}
__global__ void pd_signal_kernel(uint32_t *f0, uint32_t *f1, uint32_t seq) {
    if (f0) __hip_atomic_store(f0, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    if (f1) __hip_atomic_store(f1, seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
struct PDPush {
    const double *src[4];
    double *dst[4];
    int64_t n[4];                             // doubles (even)
    int nseg;
    uint32_t *counter, *flag;
    uint32_t seq;
};
// copies the segments into another rank's memory, then raises the flag there (the last workgroup does)
__global__ __launch_bounds__(256) void pd_push_kernel(const PDPush a) {
    typedef double v2d __attribute__((ext_vector_type(2)));
    for (int g = 0; g < a.nseg; ++g) {
        const v2d *s = reinterpret_cast<const v2d *>(a.src[g]);
        v2d *d = reinterpret_cast<v2d *>(a.dst[g]);
        const int64_t n = a.n[g] / 2;
        for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = s[i];
    }
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(a.counter, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1) {
            __hip_atomic_store(a.counter, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(a.flag, a.seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}
// my planes of the tail's right-hand side into every rank's gathered vector (blockIdx.y = destination rank)
__global__ __launch_bounds__(256) void pd_gather_kernel(const double *src, int64_t n, double *const *dst, uint32_t *const *flag, uint32_t *counters, uint32_t seq) {
    const int r = int(blockIdx.y);
    double *d = dst[r];
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) d[i] = src[i];
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
    __syncthreads();
    if (threadIdx.x == 0) {
        const uint32_t before = __hip_atomic_fetch_add(counters + r, 1u, __ATOMIC_ACQ_REL, __HIP_MEMORY_SCOPE_AGENT);
        if (before == gridDim.x - 1) {
            __hip_atomic_store(counters + r, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
            __hip_atomic_store(flag[r], seq, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
        }
    }
}

// The schedule over a set of ranks: ONE rank with RCCL exchanges, or all of them in one process with device copies
// in their place (loopback group: the same launches in the same order per rank).
struct PDExchange {
    std::vector<PlaneDist *> ranks;           // the ranks this process drives, ascending
    bool loopback = false;
    // sweep counts of the running cycle (0 or 1 each: the reference's default is V(1, 0), openmg/__init__.py:22-23): the
    // passes without their relaxation (plane.hip SWEEP = false) run on slabs whose neighbours are reached by exchanges
    int pre = 1, post = 1;

    // ghost planes of a plane-major vector of level l (both colours): `count` planes from either neighbour
    bool exchanges() const { return loopback ? ranks.size() > 1 : ranks[0]->n_ranks > 1; }
    void halo(int l, int which /* 0 x (current), 1 b */, int count, bool on_side = false) {
        for (PlaneDist *d : ranks) {
            PDLevel &L = d->lv[l];
            const hipStream_t st = on_side ? d->side : d->stream;
            const ncclComm_t cm = on_side ? d->comm_side : d->comm;
            if (!loopback && d->n_ranks > 1) OMG_NCCL(g_rccl.GroupStart());
            for (int colour = 0; colour < 2; ++colour) {
                double *mine = (which == 0 ? L.xp : L.b.p) + (colour ? L.n_ext / 2 : 0);
                const size_t cnt = size_t(count) * size_t(L.pc);
                // up: my last owned planes -> rank + 1's lower ghosts; down: my first owned planes -> rank - 1's upper ghosts
                double *send_up = mine + int64_t(PD_GHOST + L.nzo - count) * L.pc, *recv_lo = mine + int64_t(PD_GHOST - count) * L.pc;
                double *send_dn = mine + int64_t(PD_GHOST) * L.pc, *recv_hi = mine + int64_t(PD_GHOST + L.nzo) * L.pc;
                if (loopback) {
                    // (only the receives: every rank pulls from its neighbours' owned planes, which no launch between
                    // the passes writes)
                    if (d->rank > 0) {
                        PlaneDist *o = ranks[d->rank - 1];
                        PDLevel &O = o->lv[l];
                        const double *src = (which == 0 ? O.xp : O.b.p) + (colour ? O.n_ext / 2 : 0) + int64_t(PD_GHOST + O.nzo - count) * O.pc;
                        OMG_HIP(hipMemcpyAsync(recv_lo, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
                    }
                    if (d->rank + 1 < d->n_ranks) {
                        PlaneDist *o = ranks[d->rank + 1];
                        PDLevel &O = o->lv[l];
                        const double *src = (which == 0 ? O.xp : O.b.p) + (colour ? O.n_ext / 2 : 0) + int64_t(PD_GHOST) * O.pc;
                        OMG_HIP(hipMemcpyAsync(recv_hi, src, cnt * sizeof(double), hipMemcpyDeviceToDevice, st));
                    }
                } else if (d->n_ranks > 1) {
                    if (d->rank + 1 < d->n_ranks) {
                        OMG_NCCL(g_rccl.Send(send_up, cnt, ncclDouble, d->rank + 1, cm, st));
                        OMG_NCCL(g_rccl.Recv(recv_hi, cnt, ncclDouble, d->rank + 1, cm, st));
                    }
                    if (d->rank > 0) {
                        OMG_NCCL(g_rccl.Send(send_dn, cnt, ncclDouble, d->rank - 1, cm, st));
                        OMG_NCCL(g_rccl.Recv(recv_lo, cnt, ncclDouble, d->rank - 1, cm, st));
                    }
                }
            }
            if (!loopback && d->n_ranks > 1) OMG_NCCL(g_rccl.GroupEnd());
        }
    }
    // loopback: all ranks of a group share one stream order only per rank; a copy must not start before the
    // source rank's producing kernel has finished -> the group runs on ONE stream (set at creation)

    bool p2p() const { return ranks[0]->p2p != 0; }
    // peer mode: what a pass of level l (down or up, cycle c) stores into its neighbours and waits for
    PlanePlan<double>::Peer peer_of(PlaneDist *d, int l, bool going_down, bool first_of_batch) const {
        PDLevel &L = d->lv[l];
        const int nd = (int)d->lv.size();
        const bool last = l + 1 == nd;
        const uint32_t c = d->cycle_no;
        PlanePlan<double>::Peer p;
        p.planes = 3;
        p.shift = int64_t(L.nzo) * L.pc;
        if (going_down && !last) {
            const PDLevel &C = d->lv[l + 1];
            p.cplanes = 2;
            p.cshift = int64_t(C.nzo) * C.pc;
            p.zc_lo = PD_GHOST;
            p.zc_hi = PD_GHOST + C.nzo;
        }
        const bool new_is_tmp = L.tp == L.tmp.p;
        for (int i = 0; i < 2; ++i) {
            const int nb = d->rank + (i ? 1 : -1);
            if (nb < 0 || nb >= d->n_ranks) continue;
            const PDPeer &P = d->peers[size_t(nb)];
            p.x[i] = new_is_tmp ? P.tmp[size_t(l)] : P.x[size_t(l)];
            if (going_down && !last) p.bc[i] = P.b[size_t(l) + 1];
            // (I am rank - 1's NEXT neighbour and rank + 1's PREVIOUS one)
            p.flag[i] = P.flags + pf((going_down ? PF_DOWN : PF_UP) + l, i ? 0 : 1);
            if (going_down) {
                if (l == 0) {
                    p.wait_flag[i] = d->flags.p + pf(first_of_batch ? PF_PRIME : PF_UP, i);
                    p.wait_seq[i] = first_of_batch ? d->batch_no : c - 1;
                } else {
                    p.wait_flag[i] = d->flags.p + pf(PF_DOWN + l - 1, i);
                    p.wait_seq[i] = c;
                }
            } else {
                p.wait_flag[i] = d->flags.p + pf(PF_DOWN + l, i);
                p.wait_seq[i] = c;
                if (!last) {
                    p.wait_flag[2 + i] = d->flags.p + pf(PF_UP + l + 1, i);
                    p.wait_seq[2 + i] = c;
                }
            }
        }
        p.fused_wait = d->p2p == 1;
        p.done = d->flags.p + PD_DONE;
        p.status = d->flags.p + PD_STATUS;
        p.seq = c;
        p.spin = d->spin;
        return p;
    }
    // part: PlanePlan::PART_ALL, or one of the two launches of a split pass (the EDGE launch goes to the side stream;
    // the vectors are swapped once, after the pass's last launch has been enqueued: swap)
    // the finest level's passes gated on exchanges that run beside them (PlaneDist::gate)?  V(1,1) over RCCL / loopback copies,
    // more than one slab, a slab thick enough for edge chunks of its own
    bool gated0() const {
        if (p2p() || pre != 1 || post != 1) return false;
        bool forced = true;
        for (PlaneDist *d : ranks) {
            if (!d->gate || d->split || !d->lv[0].plan.can_gate()) return false;
            forced = forced && d->gate_forced;
        }
        return exchanges() || forced;
    }
    PlanePlan<double>::Gate gate_of(PlaneDist *d, int slot_a, int slot_b = -1) const {
        PlanePlan<double>::Gate g;
        g.flag[0] = d->flags.p + pf(slot_a, 0);
        g.seq[0] = d->cycle_no;
        if (slot_b >= 0) { g.flag[1] = d->flags.p + pf(slot_b, 0); g.seq[1] = d->cycle_no; }
        g.status = d->flags.p + PD_STATUS;
        g.spin = d->spin;
        return g;
    }
    // OMG_PDIST_GATE_DEBUG=1: after a gated pass, wait for it and say what its waits saw (stderr)
    void gate_report(const char *what) {
        static const bool on = [] { const char *e = experiment_env("OMG_PDIST_GATE_DEBUG"); return e && e[0] == '1'; }();
        if (!on) return;
        for (PlaneDist *d : ranks) {
            OMG_HIP(hipStreamSynchronize(d->stream));
            uint32_t f[PD_FLAGS + 2];
            OMG_HIP(hipMemcpy(f, d->flags.p, sizeof(f), hipMemcpyDeviceToHost));
            fprintf(stderr, "[pdist gate] cycle %u rank %d after %s: status %u  GX %u  GXU %u  GC %u  (tile %d x %d x %d, gate chunk %d, %d workgroups)\n", d->cycle_no, d->rank,
                    what, f[PD_STATUS], f[pf(PF_GATE_X, 0)], f[pf(PF_GATE_XU, 0)], f[pf(PF_GATE_C, 0)], d->lv[0].plan.g.TX, d->lv[0].plan.g.TY, d->lv[0].plan.g.LZ,
                    d->lv[0].plan.gate_lz(), d->lv[0].plan.gate_partials());
        }
    }
    void raise(PlaneDist *d, hipStream_t st, int slot, uint32_t seq) {
        hipLaunchKernelGGL(pd_signal_kernel, dim3(1), dim3(1), 0, st, d->flags.p + pf(slot, 0), static_cast<uint32_t *>(nullptr), seq);
        OMG_HIP(hipGetLastError());
    }
    void down(int l, bool first_of_batch = false, int part = PlanePlan<double>::PART_ALL, bool swap = true, bool on_side = false, bool gated = false) {
        for (PlaneDist *d : ranks) {
            PDLevel &L = d->lv[l];
            const bool last = l + 1 == (int)d->lv.size();
            PlanePlan<double>::Coarse c;
            c.map = last ? nullptr : L.cmap.p;
            c.b = last ? d->cb.p : d->lv[l + 1].b.p;
            c.x = nullptr;                                        // the level below takes its iterate as zero
            if (d->p2p) {
                const PlanePlan<double>::Peer p = peer_of(d, l, true, first_of_batch);
                L.plan.down(L.xp, L.tp, L.b.p, l > 0, c, d->stream, &p);
            } else {
                // (pre = 0: the residual of the iterate as it is + the restriction; x_new is not written.  A level below the finest
                // enters with a zero iterate that the pass does not write either, and the up pass reads it: cleared here)
                if (pre == 0 && l > 0) OMG_HIP(hipMemsetAsync(L.xp, 0, size_t(L.n_ext) * sizeof(double), d->stream));
                const PlanePlan<double>::Gate gt = gate_of(d, PF_GATE_X);
                L.plan.down(L.xp, L.tp, L.b.p, l > 0, c, (on_side || part == PlanePlan<double>::PART_EDGE) ? d->side : d->stream, nullptr, pre >= 1, part,
                            gated ? &gt : nullptr);
            }
            if (swap && pre >= 1) std::swap(L.xp, L.tp);
        }
    }
    void up(int l, double *partials, int part = PlanePlan<double>::PART_ALL, bool swap = true, bool on_side = false, bool gated = false) {
        for (PlaneDist *d : ranks) {
            PDLevel &L = d->lv[l];
            const bool last = l + 1 == (int)d->lv.size();
            PlanePlan<double>::Coarse c;
            c.map = last ? nullptr : L.cmap.p;
            c.e = last ? d->ce.p : d->lv[l + 1].xp;
            double *out = l == 0 ? (partials ? partials : L.plan.partials.p) : nullptr;
            if (d->p2p) {
                const PlanePlan<double>::Peer p = peer_of(d, l, false, false);
                L.plan.up(L.xp, L.tp, L.b.p, c, out, d->stream, &p);
            } else {
                const PlanePlan<double>::Gate gt = gate_of(d, PF_GATE_XU, last ? -1 : PF_GATE_C);
                L.plan.up(L.xp, L.tp, L.b.p, c, out, (on_side || part == PlanePlan<double>::PART_EDGE) ? d->side : d->stream, nullptr, post >= 1, part,
                          gated ? &gt : nullptr);
            }
            if (swap) std::swap(L.xp, L.tp);
        }
    }
    // split passes: main -> side ("the pass's inputs are complete") and side -> main ("edge chunks and exchanges done")
    void main_to_side(int l, int up_) {
        for (PlaneDist *d : ranks) {
            OMG_HIP(hipEventRecord(d->ev_main[up_][size_t(l)], d->stream));
            OMG_HIP(hipStreamWaitEvent(d->side, d->ev_main[up_][size_t(l)], 0));
        }
    }
    void side_done(int l, int up_) {
        for (PlaneDist *d : ranks) OMG_HIP(hipEventRecord(d->ev_edge[up_][size_t(l)], d->side));
    }
    void main_waits_side(int l, int up_) {
        for (PlaneDist *d : ranks) OMG_HIP(hipStreamWaitEvent(d->stream, d->ev_edge[up_][size_t(l)], 0));
    }
    bool split() const {
        // (OMG_PDIST_SPLIT=2: also with ONE slab — no neighbour, no exchange: what the two-launch schedule costs a rank
        // in device time, measurable on one GPU)
        static const bool force = [] { const char *e = getenv("OMG_PDIST_SPLIT"); return e && e[0] == '2'; }();
        if (!exchanges() && !force) return false;                 // one slab: no neighbour, nothing to overlap
        for (PlaneDist *d : ranks)
            if (!d->split) return false;
        return true;
    }
    // a level too thin for two launches runs its whole pass where the edge chunks would run
    bool two_launches(int l) const {
        for (PlaneDist *d : ranks)
            if (!d->lv[size_t(l)].plan.can_split()) return false;
        return true;
    }
    // One V(1,1) cycle with split passes.  Per pass: side stream: edge chunks, then the exchanges of what they produced
    // (ghost planes of the new iterate for this level's own up pass / for the finer level's; ghost planes of the coarse
    // right-hand side for the next level's down pass) — main stream: the inner chunks meanwhile.  Left on the critical
    // path: the tail's all-gather, and whatever part of an exchange outlasts its pass's inner chunks.
    void cycle_split(double *squares_out) {
        typedef PlanePlan<double> PP;
        const int nd = (int)ranks[0]->lv.size();
        // the ghost planes of x the first pass reads: posted behind the previous cycle's last edge launch, or (first
        // cycle after a load) exchanged here
        bool posted = true;
        for (PlaneDist *d : ranks) posted = posted && d->x0_posted;
        if (posted) main_waits_side(0, 1);
        else halo(0, 0, 3);
        mark(0, 1);
        for (int l = 0; l < nd; ++l) {
            if (l > 0) main_waits_side(l - 1, 0);                 // ghost planes of this level's right-hand side
            main_to_side(l, 0);
            if (two_launches(l)) {
                down(l, false, PP::PART_EDGE, false);
                down(l, false, PP::PART_INNER, true);             // (swapped: the exchanges below send the NEW iterate)
            } else {
                down(l, false, PP::PART_ALL, true, true);
            }
            halo(l, 0, 3, true);                                  // for this level's up pass
            if (l + 1 < nd) halo(l + 1, 1, 2, true);              // for the next level's down pass
            side_done(l, 0);
            mark(l, 3);
        }
        main_waits_side(nd - 1, 0);                               // the tail gathers ALL of the last level's coarse right-hand side
        tail_solve();
        mark(nd, 5);
        for (int l = nd - 1; l >= 0; --l) {
            main_waits_side(l, 0);                                // ghost planes of the iterate the down pass left
            if (l + 1 < nd) main_waits_side(l + 1, 1);            // ... and of the correction
            mark(l, 4);
            main_to_side(l, 1);
            if (two_launches(l)) {
                up(l, nullptr, PP::PART_EDGE, false);
                up(l, nullptr, PP::PART_INNER, true);
            } else {
                up(l, nullptr, PP::PART_ALL, true, true);
            }
            halo(l, 0, l > 0 ? 2 : 3, true);                      // the correction's ghost planes for level l - 1 / the next cycle's first pass
            side_done(l, 1);
            mark(l, 7);
        }
        for (PlaneDist *d : ranks) d->x0_posted = true;
        main_waits_side(0, 1);                                    // (the edge chunks' share of the norm's partials)
        const bool two0 = two_launches(0);
        for (PlaneDist *d : ranks)
            launch_sum(d->lv[0].plan.partials.p, two0 ? d->lv[0].plan.split_partials() : d->lv[0].plan.g.n_wg, squares_out ? squares_out : d->norm2.p,
                       d->stream);
    }
    // peer mode, once per batch of cycles: the ghost planes of b and of the incoming x, handed over only when the
    // neighbour's stream has reached its own batch start (it may have been loading new vectors until then)
    void prime() {
        for (PlaneDist *d : ranks) ++d->batch_no;
        for (PlaneDist *d : ranks) {
            uint32_t *f[2] = {nullptr, nullptr};
            for (int i = 0; i < 2; ++i) {
                const int nb = d->rank + (i ? 1 : -1);
                if (nb >= 0 && nb < d->n_ranks) f[i] = d->peers[size_t(nb)].flags + pf(PF_READY, i ? 0 : 1);
            }
            hipLaunchKernelGGL(pd_signal_kernel, dim3(1), dim3(1), 0, d->stream, f[0], f[1], d->batch_no);
        }
        for (PlaneDist *d : ranks) wait_neighbours(d, PF_READY, d->batch_no);
        for (PlaneDist *d : ranks) {
            PDLevel &L = d->lv[0];
            const bool cur_is_x = L.xp == L.x.p;
            for (int i = 0; i < 2; ++i) {
                const int nb = d->rank + (i ? 1 : -1);
                if (nb < 0 || nb >= d->n_ranks) continue;
                const PDPeer &P = d->peers[size_t(nb)];
                double *px = cur_is_x ? P.x[0] : P.tmp[0], *pb = P.b[0];
                PDPush a;
                a.nseg = 4;
                for (int colour = 0; colour < 2; ++colour) {
                    const int64_t half = colour ? L.n_ext / 2 : 0;
                    // to rank + 1: my last owned planes -> its lower ghosts; to rank - 1: my first owned planes -> its upper ghosts
                    const int64_t from_x = half + int64_t(i ? PD_GHOST + L.nzo - 3 : PD_GHOST) * L.pc, to_x = half + int64_t(i ? PD_GHOST - 3 : PD_GHOST + L.nzo) * L.pc;
                    const int64_t from_b = half + int64_t(i ? PD_GHOST + L.nzo - 2 : PD_GHOST) * L.pc, to_b = half + int64_t(i ? PD_GHOST - 2 : PD_GHOST + L.nzo) * L.pc;
                    a.src[colour] = L.xp + from_x; a.dst[colour] = px + to_x; a.n[colour] = 3 * L.pc;
                    a.src[2 + colour] = L.b.p + from_b; a.dst[2 + colour] = pb + to_b; a.n[2 + colour] = 2 * L.pc;
                }
                a.counter = d->flags.p + PD_DONE + 1 + i;
                a.flag = P.flags + pf(PF_PRIME, i ? 0 : 1);
                a.seq = d->batch_no;
                const int64_t bytes = 5 * L.pc * 2 * 8;
                const unsigned blocks = unsigned(std::max<int64_t>(1, std::min<int64_t>(64, bytes / 32768)));
                hipLaunchKernelGGL(pd_push_kernel, dim3(blocks), dim3(256), 0, d->stream, a);
            }
        }
        OMG_HIP(hipGetLastError());
    }
    void wait_neighbours(PlaneDist *d, int slot, uint32_t seq) {
        const bool lo = d->rank > 0, hi = d->rank + 1 < d->n_ranks;
        if (!lo && !hi) return;
        hipLaunchKernelGGL(pd_wait_kernel, dim3(1), dim3(64), 0, d->stream, d->flags.p, pf(slot, lo ? 0 : 1), 1, (lo && hi) ? 2 : 1, seq,
                           d->flags.p + PD_STATUS, d->spin);
    }
    // peer mode: every rank's planes of the tail's right-hand side stored into every rank's gathered vector
    void gather_p2p() {
        for (PlaneDist *d : ranks) {
            const int64_t own = int64_t(d->cnx) * d->cny * d->cnzo;
            const int par = int(d->cycle_no & 1u);
            const unsigned blocks = unsigned(std::max<int64_t>(1, std::min<int64_t>(16, own * 8 / 16384)));
            hipLaunchKernelGGL(pd_gather_kernel, dim3(blocks, unsigned(d->n_ranks)), dim3(256), 0, d->stream,
                               d->cb.p + int64_t(PD_GHOST) * d->cnx * d->cny, own, d->ag_dst[par].p, d->ag_flag.p, d->flags.p + PD_COUNT, d->cycle_no);
        }
        for (PlaneDist *d : ranks)
            hipLaunchKernelGGL(pd_wait_kernel, dim3(1), dim3(unsigned((d->n_ranks + 63) / 64 * 64)), 0, d->stream, d->flags.p, PF_GATHER * 2, 1, d->n_ranks,
                               d->cycle_no, d->flags.p + PD_STATUS, d->spin);
        OMG_HIP(hipGetLastError());
    }
    // right-hand side of the level below the slabs: gathered, solved by the replicated tail, the slab's planes
    // (and ghosts) of the correction taken out of it
    void tail_solve() {
        if (p2p()) gather_p2p();
        for (PlaneDist *d : ranks) {
            const int64_t plane = int64_t(d->cnx) * d->cny, own = plane * d->cnzo;
            const double *mine = d->cb.p + int64_t(PD_GHOST) * plane;
            if (d->p2p) {
            } else if (loopback) {
                for (PlaneDist *o : ranks)
                    OMG_HIP(hipMemcpyAsync(o->full_b.p + int64_t(d->rank) * own, mine, size_t(own) * sizeof(double), hipMemcpyDeviceToDevice, d->stream));
            } else if (d->n_ranks > 1) {
                OMG_NCCL(g_rccl.AllGather(mine, d->full_b.p, size_t(own), ncclDouble, d->comm, d->stream));
            } else {
                OMG_HIP(hipMemcpyAsync(d->full_b.p, mine, size_t(own) * sizeof(double), hipMemcpyDeviceToDevice, d->stream));
            }
        }
        for (PlaneDist *d : ranks) {
            const int64_t plane = int64_t(d->cnx) * d->cny;
            const double *gathered = (d->p2p && (d->cycle_no & 1u)) ? d->full_b2.p : d->full_b.p;
            if (omg_hierarchy_cycle_dev(d->tail, gathered, d->full_x.p, pre, post, d->stream) != OMG_OK)
                throw Error(OMG_ERR_HIP, std::string("replicated tail cycle: ") + omg_last_error());
            // planes [k0 - ghost, k0 + own + ghost) of the correction, clipped to the grid (the rest stays zero)
            const int64_t k0 = int64_t(d->rank) * d->cnzo, nzg = int64_t(d->n_ranks) * d->cnzo;
            const int64_t lo = std::max<int64_t>(0, k0 - PD_GHOST), hi = std::min<int64_t>(nzg, k0 + d->cnzo + PD_GHOST);
            OMG_HIP(hipMemcpyAsync(d->ce.p + (lo - (k0 - PD_GHOST)) * plane, d->full_x.p + lo * plane, size_t(hi - lo) * size_t(plane) * sizeof(double),
                                   hipMemcpyDeviceToDevice, d->stream));
        }
    }
    // one V(1,1) cycle; the norm's squares of every rank in its norm2 (level-0 up pass partials summed)
    void mark(int level, int phase) {
        for (PlaneDist *d : ranks) d->mark(level, phase);
    }
    void cycle(double *squares_out = nullptr /* one rank only: where its sum of squares goes instead of norm2 */, bool first_of_batch = false) {
        const int nd = (int)ranks[0]->lv.size();
        for (PlaneDist *d : ranks) ++d->cycle_no;
        OMG_REQUIRE(pre >= 0 && pre <= 1 && post >= 0 && post <= 1, "plane slabs: sweep counts of 0 or 1 (more sweeps: the set-by-set runner, omg_dist_*)");
        OMG_REQUIRE((pre == 1 && post == 1) || (!p2p() && !split()), "plane slabs: cycles other than V(1,1) run over exchanges (no peer mode, no split passes)");
        if (p2p()) {
            // peer mode: no exchange launches — the passes store into their neighbours and wait for them (prime() has
            // run before the batch's first cycle)
            for (int l = 0; l < nd; ++l) { down(l, first_of_batch); mark(l, 3); }
            tail_solve();
            mark(nd, 5);
            for (int l = nd - 1; l >= 0; --l) { up(l, nullptr); mark(l, 7); }
            for (PlaneDist *d : ranks)
                launch_sum(d->lv[0].plan.partials.p, d->lv[0].plan.g.n_wg, squares_out ? squares_out : d->norm2.p, d->stream);
            return;
        }
        if (split()) { cycle_split(squares_out); return; }
        const bool g0 = gated0();
        if (g0) {
            // the ghost planes of x the first pass reads were posted behind the previous cycle's last pass (side stream), or
            // — the first cycle after a load — are exchanged here, in stream order
            bool posted = true;
            for (PlaneDist *d : ranks) posted = posted && d->x0_posted;
            if (!posted) {
                halo(0, 0, 3);
                for (PlaneDist *d : ranks) raise(d, d->stream, PF_GATE_X, d->cycle_no);
            }
        } else {
            halo(0, 0, 3);
        }
        mark(0, 1);
        for (int l = 0; l < nd; ++l) {
            if (l > 0) { halo(l, 1, 2); mark(l, 2); }
            down(l, false, PlanePlan<double>::PART_ALL, true, false, g0 && l == 0);
            if (g0 && l == 0) gate_report("down(0)");
            mark(l, 3);
            // the ghost planes for the up pass: on the side stream, beside the coarser levels (nothing before the up
            // pass reads or writes them, nor the planes they are copied from)
            if (!exchanges() && !(g0 && l == 0)) continue;        // one slab: no neighbour, no second stream
            for (PlaneDist *d : ranks) {
                OMG_HIP(hipEventRecord(d->ev_down[size_t(l)], d->stream));
                OMG_HIP(hipStreamWaitEvent(d->side, d->ev_down[size_t(l)], 0));
            }
            halo(l, 0, 3, true);
            if (g0 && l == 0) { for (PlaneDist *d : ranks) raise(d, d->side, PF_GATE_XU, d->cycle_no); }
            else { for (PlaneDist *d : ranks) OMG_HIP(hipEventRecord(d->ev_halo[size_t(l)], d->side)); }
        }
        tail_solve();
        mark(nd, 5);
        for (int l = nd - 1; l >= 0; --l) {
            const bool gl = g0 && l == 0;
            if (l + 1 < nd) {                                     // ghost planes of the correction
                if (gl) {
                    // ... beside the finest up pass's inner chunks: its edge chunks wait for PF_GATE_C
                    for (PlaneDist *d : ranks) {
                        OMG_HIP(hipEventRecord(d->ev_main[1][1], d->stream));
                        OMG_HIP(hipStreamWaitEvent(d->side, d->ev_main[1][1], 0));
                    }
                    halo(l + 1, 0, 2, true);
                    for (PlaneDist *d : ranks) raise(d, d->side, PF_GATE_C, d->cycle_no);
                } else {
                    halo(l + 1, 0, 2);
                }
                mark(l, 6);
            }
            if (exchanges() && !gl)
                for (PlaneDist *d : ranks) OMG_HIP(hipStreamWaitEvent(d->stream, d->ev_halo[size_t(l)], 0));
            mark(l, 4);
            up(l, nullptr, PlanePlan<double>::PART_ALL, true, false, gl);
            if (gl) gate_report("up(0)");
            mark(l, 7);
        }
        if (g0) {
            // the next cycle's first exchange, posted now: it runs beside that cycle's down pass, whose edge chunks wait for it
            const char *pe = getenv("OMG_PDIST_GATE_POISON");
            const bool poison = pe && pe[0] == '1';
            for (PlaneDist *d : ranks) {
                OMG_HIP(hipEventRecord(d->ev_main[1][0], d->stream));
                OMG_HIP(hipStreamWaitEvent(d->side, d->ev_main[1][0], 0));
            }
            halo(0, 0, 3, true);
            // (OMG_PDIST_GATE_POISON=1, tests: the flag is never raised — the next pass's bounded wait must give up and say so)
            if (!poison) for (PlaneDist *d : ranks) raise(d, d->side, PF_GATE_X, d->cycle_no + 1);
            for (PlaneDist *d : ranks) d->x0_posted = true;
        }
        for (PlaneDist *d : ranks)
            launch_sum(d->lv[0].plan.partials.p, g0 ? d->lv[0].plan.gate_partials() : d->lv[0].plan.g.n_wg, squares_out ? squares_out : d->norm2.p, d->stream);
    }
    // a batch of cycles of ONE rank: slot k holds this rank's squares of cycle k; one all-reduce for all of them
    // (nothing on the device waits for a norm, so none is needed inside a cycle), then the roots in place
    void norms_of_batch(double *slots, int n) {
        PlaneDist *d = ranks[0];
        if (d->n_ranks > 1) OMG_NCCL(g_rccl.AllReduce(slots, slots, size_t(n), ncclDouble, ncclSum, d->comm, d->stream));
        hipLaunchKernelGGL(pd_sqrt_batch_kernel, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, d->stream, slots, n);
        OMG_HIP(hipGetLastError());
    }
    // ||b - A x|| of the cycle just run into out (device) on every rank
    void norm(double *const *out_per_rank) {
        if (loopback) {
            // ranks in ascending order into rank 0's accumulator, then to everybody (one stream)
            PlaneDist *z = ranks[0];
            for (size_t r = 1; r < ranks.size(); ++r) hipLaunchKernelGGL(pd_add_kernel, dim3(1), dim3(1), 0, z->stream, z->norm2.p, ranks[r]->norm2.p);
            for (size_t r = 0; r < ranks.size(); ++r) hipLaunchKernelGGL(pd_sqrt_kernel, dim3(1), dim3(1), 0, z->stream, z->norm2.p, out_per_rank[r]);
        } else {
            PlaneDist *d = ranks[0];
            if (d->n_ranks > 1) OMG_NCCL(g_rccl.AllReduce(d->norm2.p, d->norm2.p, 1, ncclDouble, ncclSum, d->comm, d->stream));
            hipLaunchKernelGGL(pd_sqrt_kernel, dim3(1), dim3(1), 0, d->stream, d->norm2.p, out_per_rank[0]);
        }
        OMG_HIP(hipGetLastError());
    }
};

std::unique_ptr<PlaneDist> pd_create(int rank, int n_ranks, int nx, int ny, int nz_global, int n_levels, const double *coef7, double w) {
    OMG_REQUIRE(n_ranks >= 1 && rank >= 0 && rank < n_ranks && n_levels >= 1 && coef7, "bad argument");
    OMG_REQUIRE(nz_global % n_ranks == 0, "planes must divide evenly over the ranks");
    require_device();
    std::unique_ptr<PlaneDist> d(new PlaneDist);
    d->rank = rank;
    d->n_ranks = n_ranks;
    OMG_HIP(hipStreamCreateWithFlags(&d->own, hipStreamNonBlocking));
    d->stream = d->own;
    {
        // the side stream carries the edge chunks of split passes and the exchanges behind them: ahead of the main
        // stream's inner chunks when both have workgroups to place
        int least = 0, greatest = 0;
        OMG_HIP(hipDeviceGetStreamPriorityRange(&least, &greatest));
        OMG_HIP(hipStreamCreateWithPriority(&d->side_own, hipStreamNonBlocking, greatest));
    }
    d->side = d->side_own;
    d->lv.resize(size_t(n_levels));
    for (int l = 0; l < n_levels; ++l) {
        hipEvent_t a, b;
        OMG_HIP(hipEventCreateWithFlags(&a, hipEventDisableTiming));
        OMG_HIP(hipEventCreateWithFlags(&b, hipEventDisableTiming));
        d->ev_down.push_back(a);
        d->ev_halo.push_back(b);
        for (int i = 0; i < 2; ++i) {
            hipEvent_t m, e;
            OMG_HIP(hipEventCreateWithFlags(&m, hipEventDisableTiming));
            OMG_HIP(hipEventCreateWithFlags(&e, hipEventDisableTiming));
            d->ev_main[i].push_back(m);
            d->ev_edge[i].push_back(e);
        }
    }
    int lx = nx, ly = ny, lz = nz_global / n_ranks;
    for (int l = 0; l < n_levels; ++l) {
        PDLevel &L = d->lv[size_t(l)];
        OMG_REQUIRE(lz >= 2 && !(lz & 1) && !(lx & 1) && !(ly & 1), "every distributed level needs an even number (>= 2) of planes per rank and even extents");
        L.nx = lx; L.ny = ly; L.nzo = lz;
        L.pc = int64_t(lx / 2) * ly;
        L.n_ext = int64_t(lx) * ly * (lz + 2 * PD_GHOST);
        double c[7];
        for (int e = 0; e < 7; ++e) c[e] = coef7[7 * l + e];
        L.plan.build_slab(lx, ly, lz, PD_GHOST, PD_GHOST, rank == 0, rank == n_ranks - 1, c, w);
        {
            // OMG_DIST_VEC_POOL=1: one allocation for the three vectors the passes stream side by side, as hierarchy.hip
            // pooled_vectors does for whole grids (there: -2.6 % per cycle).  Measured on a slab with its ghost planes
            // (`bench.py --dist 1`): 0.2944 against 0.2905 ms per cycle — slower, so three allocations stay the default here.
            static const bool pool_on = [] { const char *e = experiment_env("OMG_DIST_VEC_POOL"); return e && e[0] == '1'; }();
            if (pool_on) {
                const size_t MB2 = size_t(2) << 20, bytes = size_t(L.n_ext) * sizeof(double);
                const size_t span = (bytes + 2 * DEVBUF_SLACK + vector_stagger(2) + MB2 - 1) / MB2 * MB2;
                L.pool.alloc(3 * span);
                // (order x, b, x's twin, as hierarchy.hip has it)
                L.x.borrow(reinterpret_cast<double *>(L.pool.p), size_t(L.n_ext), 0);
                L.b.borrow(reinterpret_cast<double *>(L.pool.p + span + vector_stagger(2)), size_t(L.n_ext), span + vector_stagger(2));
                L.tmp.borrow(reinterpret_cast<double *>(L.pool.p + 2 * span + vector_stagger(1)), size_t(L.n_ext), 2 * span + vector_stagger(1));
            } else {
                L.x.alloc(size_t(L.n_ext)); L.tmp.alloc(size_t(L.n_ext), vector_stagger(1)); L.b.alloc(size_t(L.n_ext), vector_stagger(2));
            }
        }
        L.x.zero(d->stream); L.tmp.zero(d->stream); L.b.zero(d->stream);
        L.xp = L.x.p; L.tp = L.tmp.p;
        lx /= 2; ly /= 2; lz /= 2;
    }
    // coarse slab (natural, extended) -> slot in the next level's red-black ordering
    for (int l = 0; l + 1 < n_levels; ++l) {
        const PDLevel &C = d->lv[size_t(l) + 1];
        std::vector<int32_t> map(size_t(C.n_ext));
        const int64_t nr = C.n_ext / 2;
        for (int64_t e = 0; e < C.n_ext; ++e) {
            const int64_t i = e % C.nx, j = (e / C.nx) % C.ny, k = e / (int64_t(C.nx) * C.ny);
            map[size_t(e)] = int32_t((((i + j + k) & 1) ? nr : 0) + e / 2);
        }
        d->lv[size_t(l)].cmap.alloc(map.size());
        d->lv[size_t(l)].cmap.upload(map.data(), map.size(), d->stream);
        OMG_HIP(hipStreamSynchronize(d->stream));
    }
    OMG_REQUIRE(lz >= 1, "the level below the slabs needs at least one plane per rank");
    d->cnx = lx; d->cny = ly; d->cnzo = lz;
    const int64_t cplane = int64_t(lx) * ly;
    d->cb.alloc(size_t(cplane * (lz + 2 * PD_GHOST)));
    d->ce.alloc(size_t(cplane * (lz + 2 * PD_GHOST)));
    d->cb.zero(d->stream); d->ce.zero(d->stream);
    d->full_b.alloc(size_t(cplane * lz * n_ranks));
    d->full_x.alloc(size_t(cplane * lz * n_ranks));
    d->norm2.alloc(1);
    d->norms.alloc(64);
    d->flags.alloc(size_t(PD_COUNT + std::max(n_ranks, 8)));
    d->flags.zero(d->stream);
    for (PDLevel &L : d->lv) L.plan.status = d->flags.p + PD_STATUS;       // (bit 1: a wave gave up waiting for its neighbour wave)
    d->full_b2.alloc(size_t(cplane * lz * n_ranks));
    d->peers.resize(size_t(n_ranks));
    if (const char *e = getenv("OMG_P2P_SPIN")) d->spin = uint32_t(std::max(1L, atol(e)));
    OMG_HIP(hipHostMalloc(reinterpret_cast<void **>(&d->progress), sizeof(uint32_t), hipHostMallocDefault));
    *d->progress = 0;
    d->nat.alloc(size_t(int64_t(nx) * ny * (nz_global / n_ranks)));
    OMG_HIP(hipStreamSynchronize(d->stream));
    {
        // the finest slab's tiling: measured on its own (zero) vectors
        PDLevel &L = d->lv[0];
        const bool last = n_levels == 1;
        PlanePlan<double>::Coarse c;
        c.map = last ? nullptr : L.cmap.p;
        c.b = last ? d->cb.p : d->lv[1].b.p;
        c.e = last ? d->ce.p : d->lv[1].xp;
        L.plan.tune(L.xp, L.tp, L.b.p, c, d->stream);
        // ... and WHERE its three vectors lie: the passes run 7 % apart on different allocations of the same vectors
        // (hierarchy.hip place_finest_pool, profiles/r05_pool_placement.txt), and with one process per GPU the slowest rank
        // sets the cycle.  Candidates (hipMalloc only: a neighbour may map these vectors over hipIpc) until one is good
        // (the passes at 4.5 TB/s of needed bytes) or OMG_PDIST_TRIALS (5) are tried, no more than 16 GB held.
        static const int trials = [] { const char *e = experiment_env("OMG_PDIST_TRIALS"); return e && e[0] ? atoi(e) : 5; }();
        const size_t triple = 3 * size_t(L.n_ext) * sizeof(double);
        const int max_trials = int(std::min<size_t>(size_t(std::max(trials, 1)), std::max<size_t>(2, (size_t(16) << 30) / std::max<size_t>(triple, 1))));
        if (!L.pool.p && trials >= 2 && L.n_ext >= (int64_t(1) << 23)) {
            auto timed = [&]() -> float {
                L.x.zero(d->stream); L.tmp.zero(d->stream); L.b.zero(d->stream);
                if (last) d->cb.zero(d->stream); else d->lv[1].b.zero(d->stream);
                return L.plan.time_pair(L.x.p, L.tmp.p, L.b.p, c, d->stream, true, 4);
            };
            const bool debug = SetupTimer::on();
            float best = timed(), worst = best;
            if (debug) fprintf(stderr, "[omg setup] rank %d: slab vectors, candidate 0: %.1f us per down + up\n", rank, best);
            struct Triple { DevBuf<double> x, tmp, b; };
            std::vector<Triple> held;
            for (int k = 1; k < max_trials; ++k) {
                if (6.0 * 8.0 * double(L.n_ext) / (double(best) * 1e-6) >= 4.5e12 || (k >= 3 && best <= 0.93f * worst)) break;     // (good: as hierarchy.hip has it)
                Triple t;
                try { t.x.alloc(size_t(L.n_ext)); t.tmp.alloc(size_t(L.n_ext), vector_stagger(1)); t.b.alloc(size_t(L.n_ext), vector_stagger(2)); }
                catch (const Error &) { (void)hipGetLastError(); break; }      // (no memory for another candidate: what there is stays)
                std::swap(L.x, t.x); std::swap(L.tmp, t.tmp); std::swap(L.b, t.b);
                const float us = timed();
                if (debug) fprintf(stderr, "[omg setup] rank %d: slab vectors, candidate %d: %.1f us per down + up\n", rank, k, us);
                worst = std::max(worst, us);
                if (us < best) best = us;
                else { std::swap(L.x, t.x); std::swap(L.tmp, t.tmp); std::swap(L.b, t.b); }
                held.push_back(std::move(t));
            }
            L.b.zero(d->stream);
            L.xp = L.x.p; L.tp = L.tmp.p;
        }
        L.x.zero(d->stream); L.tmp.zero(d->stream);
        if (last) d->cb.zero(d->stream); else d->lv[1].b.zero(d->stream);
        OMG_HIP(hipStreamSynchronize(d->stream));
    }
    return d;
}

}  // namespace
}  // namespace omg

struct omg_pdist {
    std::unique_ptr<omg::PlaneDist> d;
};
struct omg_pdist_group {
    std::vector<omg_pdist *> ranks;
};

namespace omg {
namespace {
}  // namespace
}  // namespace omg

extern "C" {

int omg_dist_create_ex(int rank, int n_ranks, int n_levels, const omg_dist_level *levels,
                       const omg_csr *coarse_global, const int64_t *coarse_counts, int smoother,
                       double omega, int dtype, omg_dist **out) {
    return guarded([&] {
        OMG_REQUIRE(out, "out is null");
        *out = nullptr;
        OMG_REQUIRE(dtype == OMG_DTYPE_F64 || dtype == OMG_DTYPE_F32, "unknown dtype");
        std::unique_ptr<omg_dist> h(new omg_dist);
        if (dtype == OMG_DTYPE_F32) h->f = create<float>(rank, n_ranks, n_levels, levels, coarse_global, coarse_counts, smoother, omega);
        else h->d = create<double>(rank, n_ranks, n_levels, levels, coarse_global, coarse_counts, smoother, omega);
        *out = h.release();
    });
}

int omg_dist_create(int rank, int n_ranks, int n_levels, const omg_dist_level *levels,
                    const omg_csr *coarse_global, const int64_t *coarse_counts, int smoother,
                    double omega, omg_dist **out) {
    return omg_dist_create_ex(rank, n_ranks, n_levels, levels, coarse_global, coarse_counts, smoother, omega,
                              OMG_DTYPE_F64, out);
}

int omg_dist_destroy(omg_dist *d) {
    return guarded([&] {
        if (!d) return;
        if (d->d || d->f) with(d, [&](auto *dd) { (void)hipStreamSynchronize(dd->stream); });
        delete d;
    });
}

int omg_dist_set_stream(omg_dist *d, void *hip_stream) {
    return guarded([&] {
        with(d, [&](auto *dd) {
            OMG_HIP(hipStreamSynchronize(dd->stream));
            dd->stream = hip_stream ? reinterpret_cast<hipStream_t>(hip_stream) : dd->own;
        });
    });
}

int omg_dist_set_tail(omg_dist *d, omg_hierarchy *tail) {
    return guarded([&] {
        with(d, [&](auto *dd) {
            if (tail) {
                int64_t n = 0;
                OMG_REQUIRE(omg_hierarchy_level_rows(tail, 0, &n) == OMG_OK && n == dd->n_coarse,
                            "tail hierarchy's finest level must have as many rows as the last distributed level has in total");
            }
            dd->tail = tail;
        });
    });
}

int omg_dist_sync(omg_dist *d) {
    return guarded([&] { with(d, [&](auto *dd) { OMG_HIP(hipStreamSynchronize(dd->stream)); }); });
}

int omg_rccl_unique_id(void *out128) {
    return guarded([&] {
        OMG_REQUIRE(out128, "null buffer");
        g_rccl.load();
        ncclUniqueId id;
        OMG_NCCL(g_rccl.GetUniqueId(&id));
        std::memcpy(out128, &id, sizeof(id));
    });
}

// What one grouped halo exchange costs on this GPU, measured with the calls the runner makes: a ONE-rank
// communicator, ncclSend + ncclRecv of `bytes` to the rank itself inside ncclGroupStart / ncclGroupEnd on a
// stream, `reps` of them back to back, each followed by a small kernel (as a smoother set follows an exchange
// in the cycle), one hipEvent bracket.  No xGMI hop is involved, so this is the FLOOR of an exchange: API
// and launch cost of the grouped pair plus a device copy — what N = 8 adds on top is the link.
__global__ void exchange_probe_touch(double *p) { p[threadIdx.x] += 1.0; }
int omg_rccl_self_exchange_time(int64_t bytes, int reps, double *avg_us) {
    return guarded([&] {
        OMG_REQUIRE(bytes > 0 && reps > 0 && avg_us, "bad argument");
        require_device();
        g_rccl.load();
        ncclUniqueId id;
        OMG_NCCL(g_rccl.GetUniqueId(&id));
        ncclComm_t comm = nullptr;
        OMG_NCCL(g_rccl.CommInitRank(&comm, 1, id, 0));
        hipStream_t s = nullptr;
        OMG_HIP(hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
        {
            DevBuf<char> a, b;
            a.alloc(size_t(bytes));
            b.alloc(size_t(bytes));
            DevBuf<double> t(64);
            OMG_HIP(hipMemsetAsync(a.p, 1, size_t(bytes), s));
            OMG_HIP(hipMemsetAsync(t.p, 0, 64 * sizeof(double), s));
            auto once = [&]() {
                OMG_NCCL(g_rccl.GroupStart());
                OMG_NCCL(g_rccl.Send(a.p, size_t(bytes), ncclChar, 0, comm, s));
                OMG_NCCL(g_rccl.Recv(b.p, size_t(bytes), ncclChar, 0, comm, s));
                OMG_NCCL(g_rccl.GroupEnd());
                hipLaunchKernelGGL(exchange_probe_touch, dim3(1), dim3(64), 0, s, t.p);
            };
            for (int i = 0; i < 5; ++i) once();
            hipEvent_t e0, e1;
            OMG_HIP(hipEventCreate(&e0));
            OMG_HIP(hipEventCreate(&e1));
            OMG_HIP(hipEventRecord(e0, s));
            for (int i = 0; i < reps; ++i) once();
            OMG_HIP(hipEventRecord(e1, s));
            OMG_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            OMG_HIP(hipEventElapsedTime(&ms, e0, e1));
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            *avg_us = 1e3 * double(ms) / reps;
        }
        (void)hipStreamSynchronize(s);
        (void)g_rccl.CommDestroy(comm);
        (void)hipStreamDestroy(s);
    });
}

int omg_dist_connect(omg_dist *d, const void *unique_id128) {
    return guarded([&] {
        OMG_REQUIRE(unique_id128, "null argument");
        with(d, [&](auto *dd) {
            g_rccl.load();
            ncclUniqueId id;
            std::memcpy(&id, unique_id128, sizeof(id));
            OMG_NCCL(g_rccl.CommInitRank(&dd->comm, dd->n_ranks, id, dd->rank));
        });
    });
}

int omg_dist_rccl_ranks(omg_dist *d, int *count) {
    return guarded([&] {
        OMG_REQUIRE(count, "null argument");
        with(d, [&](auto *dd) {
            *count = 0;
            if (dd->comm) OMG_NCCL(g_rccl.CommCount(dd->comm, count));
        });
    });
}

int omg_dist_load(omg_dist *d, const double *b_local, const double *x0_local) {
    return guarded([&] {
        OMG_REQUIRE(b_local, "null argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            auto &L = dd->lv[0];
            auto put = [&](const double *host, V *dst) {
                if (std::is_same<V, double>::value && L.ord.identity) {
                    OMG_HIP(hipMemcpyAsync(dst, host, L.n_loc * sizeof(double), hipMemcpyHostToDevice, dd->stream));
                } else {                                       // permute and / or narrow on the device
                    L.nat.upload(host, L.n_loc, dd->stream);
                    launch_gather<double, V>(L.nat.p, L.ord.identity ? nullptr : L.perm.p, dst, L.n_loc, dd->stream);
                }
            };
            put(b_local, L.b.p);
            if (x0_local) put(x0_local, L.xp);
            else OMG_HIP(hipMemsetAsync(L.xp, 0, (L.n_loc + L.n_halo) * sizeof(V), dd->stream));
            OMG_HIP(hipStreamSynchronize(dd->stream));
            dd->halo_dirty = x0_local != nullptr;
            dd->loaded = true;
        });
    });
}

int omg_dist_fetch(omg_dist *d, double *x_local) {
    return guarded([&] {
        OMG_REQUIRE(x_local, "null argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            OMG_REQUIRE(dd->loaded, "nothing loaded");
            auto &L = dd->lv[0];
            if (std::is_same<V, double>::value && L.ord.identity) {
                OMG_HIP(hipMemcpyAsync(x_local, L.xp, L.n_loc * sizeof(double), hipMemcpyDeviceToHost, dd->stream));
            } else {
                launch_scatter<V, double>(L.xp, L.ord.identity ? nullptr : L.perm.p, L.nat.p, L.n_loc, dd->stream);
                L.nat.download(x_local, L.n_loc, dd->stream);
            }
            OMG_HIP(hipStreamSynchronize(dd->stream));
        });
    });
}

int omg_dist_cycle(omg_dist *d, int pre, int post, double *norm) {
    return guarded([&] {
        OMG_REQUIRE(pre >= 0 && post >= 0, "bad argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            OMG_REQUIRE(dd->n_ranks == 1 || dd->comm, "omg_dist_connect has not been called");
            Runner<V> r;
            r.rs = {dd};
            r.rccl = true;
            r.run(pre, post, norm);
        });
    });
}

int omg_dist_cycles(omg_dist *d, int pre, int post, int n_cycles, double *norms) {
    return guarded([&] {
        OMG_REQUIRE(pre >= 0 && post >= 0 && n_cycles >= 0, "bad argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            OMG_REQUIRE(dd->n_ranks == 1 || dd->comm, "omg_dist_connect has not been called");
            Runner<V> r;
            r.rs = {dd};
            r.rccl = true;
            r.run_batch(pre, post, n_cycles, norms);
        });
    });
}

// Plain y = A_0 x over this rank's rows (x: the resident iterate with its halo, y: the residual
// buffer), `reps` launches in one hipEvent bracket on the rank's stream: the per-GPU fine-grid
// SpMV rate of a multi-GPU run.
int omg_dist_spmv_time(omg_dist *d, int reps, double *avg_ms) {
    return guarded([&] {
        OMG_REQUIRE(reps > 0 && avg_ms, "bad argument");
        with(d, [&](auto *dd) {
            using V = value_of<decltype(dd)>;
            OMG_REQUIRE(dd->loaded && dd->lv.size() > 1, "nothing loaded / single level");
            auto &L = dd->lv[0];
            RowArgsT<V> a;
            a.x = L.xp; a.y = L.r.p;
            launch_rows(L.A, ROW_SPMV, -1, a, dd->stream);
            hipEvent_t e0, e1;
            OMG_HIP(hipEventCreate(&e0));
            OMG_HIP(hipEventCreate(&e1));
            OMG_HIP(hipEventRecord(e0, dd->stream));
            for (int i = 0; i < reps; ++i) launch_rows(L.A, ROW_SPMV, -1, a, dd->stream);
            OMG_HIP(hipEventRecord(e1, dd->stream));
            OMG_HIP(hipEventSynchronize(e1));
            float ms = 0.f;
            OMG_HIP(hipEventElapsedTime(&ms, e0, e1));
            (void)hipEventDestroy(e0);
            (void)hipEventDestroy(e1);
            *avg_ms = double(ms) / reps;
        });
    });
}

int omg_dist_level_flags(omg_dist *d, int level, int *flags) {
    return guarded([&] {
        OMG_REQUIRE(flags, "null");
        with(d, [&](auto *dd) {
            OMG_REQUIRE(level >= 0 && level + 1 < (int)dd->lv.size(), "level out of range (smoothed levels only)");
            const auto &L = dd->lv[level];
            *flags = (L.scatter_prolong ? OMG_LEVEL_SCATTER_PROLONG : 0) | (L.set_group == 2 ? 4 : 0) |
                     (L.scatter_prolong && L.R.n_sets() == 3 ? 8 : 0);
        });
    });
}

int omg_dist_format_info(omg_dist *d, int level, int op, int set, int64_t *out) {
    return guarded([&] {
        OMG_REQUIRE(out && op >= 0 && op <= 2, "null / unknown operator");
        with(d, [&](auto *dd) {
            OMG_REQUIRE(level >= 0 && level < (int)dd->lv.size(), "level out of range");
            OMG_REQUIRE(op == 0 || level + 1 < (int)dd->lv.size(), "the last level has no restriction");
            const auto &L = dd->lv[level];
            (op == 0 ? L.A : op == 1 ? L.R : L.P).format_info(set, out);
        });
    });
}

int omg_dist_group_create(int n, omg_dist **ranks, omg_dist_group **out) {
    return guarded([&] {
        OMG_REQUIRE(n >= 1 && ranks && out, "bad argument");
        std::unique_ptr<omg_dist_group> g(new omg_dist_group);
        for (int i = 0; i < n; ++i) {
            OMG_REQUIRE(ranks[i] && (ranks[i]->d || ranks[i]->f), "null rank");
            OMG_REQUIRE(bool(ranks[i]->f) == bool(ranks[0]->f), "all ranks of a group must have one dtype");
            with(ranks[i], [&](auto *dd) { OMG_REQUIRE(dd->n_ranks == n, "group must hold every rank of the decomposition"); });
            g->ranks.push_back(ranks[i]);
        }
        // one shared stream: the loopback exchange relies on in-order execution
        hipStream_t shared = nullptr;
        with(g->ranks[0], [&](auto *dd) { shared = dd->own; });
        for (omg_dist *d : g->ranks)
            with(d, [&](auto *dd) { OMG_HIP(hipStreamSynchronize(dd->stream)); dd->stream = shared; });
        *out = g.release();
    });
}

int omg_dist_group_destroy(omg_dist_group *g) {
    delete g;
    return OMG_OK;
}

int omg_dist_group_cycles(omg_dist_group *g, int pre, int post, int n_cycles, double *norms) {
    return guarded([&] {
        OMG_REQUIRE(g && !g->ranks.empty() && pre >= 0 && post >= 0 && n_cycles >= 0, "bad argument");
        with(g->ranks[0], [&](auto *first) {
            using V = value_of<decltype(first)>;
            Runner<V> r;
            for (omg_dist *d : g->ranks) {
                if constexpr (std::is_same<V, double>::value) r.rs.push_back(d->d.get());
                else r.rs.push_back(d->f.get());
            }
            r.rccl = false;
            r.run_batch(pre, post, n_cycles, norms);
        });
    });
}

int omg_dist_group_cycle(omg_dist_group *g, int pre, int post, double *norm) {
    return guarded([&] {
        OMG_REQUIRE(g && !g->ranks.empty() && pre >= 0 && post >= 0, "bad argument");
        with(g->ranks[0], [&](auto *first) {
            using V = value_of<decltype(first)>;
            Runner<V> r;
            for (omg_dist *d : g->ranks) {
                if constexpr (std::is_same<V, double>::value) r.rs.push_back(d->d.get());
                else r.rs.push_back(d->f.get());
            }
            r.rccl = false;
            r.run(pre, post, norm);
        });
    });
}

// ---- plane-pipelined slabs -----------------------------------------------------------------------------------
int omg_pdist_create(int rank, int n_ranks, int nx, int ny, int nz_global, int n_levels, const double *coef7, double weight,
                     omg_pdist **out) {
    return guarded([&] {
        OMG_REQUIRE(out, "out is null");
        *out = nullptr;
        std::unique_ptr<omg_pdist> h(new omg_pdist);
        h->d = pd_create(rank, n_ranks, nx, ny, nz_global, n_levels, coef7, weight);
        *out = h.release();
    });
}

int omg_pdist_destroy(omg_pdist *d) {
    return guarded([&] {
        if (!d) return;
        if (d->d) (void)hipStreamSynchronize(d->d->stream);
        delete d;
    });
}

int omg_pdist_set_tail(omg_pdist *d, omg_hierarchy *tail) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && tail, "null argument");
        int64_t n = 0;
        OMG_REQUIRE(omg_hierarchy_level_rows(tail, 0, &n) == OMG_OK &&
                        n == int64_t(d->d->cnx) * d->d->cny * d->d->cnzo * d->d->n_ranks,
                    "tail hierarchy's finest level must be the level below the slabs");
        d->d->tail = tail;
    });
}

int omg_pdist_connect(omg_pdist *d, const void *unique_id128, const void *unique_id128_side) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && unique_id128 && unique_id128_side, "null argument");
        g_rccl.load();
        ncclUniqueId id;
        std::memcpy(&id, unique_id128, sizeof(id));
        OMG_NCCL(g_rccl.CommInitRank(&d->d->comm, d->d->n_ranks, id, d->d->rank));
        std::memcpy(&id, unique_id128_side, sizeof(id));
        OMG_NCCL(g_rccl.CommInitRank(&d->d->comm_side, d->d->n_ranks, id, d->d->rank));
    });
}

int omg_pdist_rccl_ranks(omg_pdist *d, int *count) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && count, "null argument");
        *count = 0;
        if (d->d->comm) OMG_NCCL(g_rccl.CommCount(d->d->comm, count));
    });
}

/* ---- peer mode (xGMI peer stores instead of RCCL launches) ------------------------------------------------------
 * Every rank exports IPC handles of the buffers its neighbours store into (omg_pdist_p2p_handles: 3 + 3 per level
 * handles of 64 bytes: flags, the two gathered right-hand sides, then x / tmp / b of every level), the control plane
 * hands them round, every rank opens every other rank's (omg_pdist_p2p_open; same process: omg_pdist_p2p_local),
 * then omg_pdist_p2p_enable(mode): 1 = the passes wait for their neighbours' flags themselves (one GPU per rank),
 * 2 = a one-workgroup wait launch before each pass (ranks that share a GPU: a pass that waited itself would hold the
 * compute units the neighbour's pass needs), 0 = back to RCCL.  Needs >= 4 planes per rank on every level. */
// (with how far each vector starts into its allocation: common.h vector_stagger — the same on every rank)
static void pd_own_buffers(PlaneDist *d, std::vector<void *> &out, std::vector<size_t> *shift = nullptr) {
    out = {d->flags.p, d->full_b.p, d->full_b2.p};
    if (shift) *shift = {d->flags.shift, d->full_b.shift, d->full_b2.shift};
    for (PDLevel &L : d->lv) {
        out.push_back(L.x.p); out.push_back(L.tmp.p); out.push_back(L.b.p);
        if (shift) { shift->push_back(L.x.shift); shift->push_back(L.tmp.shift); shift->push_back(L.b.shift); }
    }
}
static void pd_attach(PlaneDist *d, int peer_rank, const std::vector<void *> &bufs) {
    PDPeer &P = d->peers[size_t(peer_rank)];
    P.flags = static_cast<uint32_t *>(bufs[0]);
    P.full_b[0] = static_cast<double *>(bufs[1]);
    P.full_b[1] = static_cast<double *>(bufs[2]);
    P.x.clear(); P.tmp.clear(); P.b.clear();
    for (size_t l = 0; l < d->lv.size(); ++l) {
        P.x.push_back(static_cast<double *>(bufs[3 + 3 * l]));
        P.tmp.push_back(static_cast<double *>(bufs[4 + 3 * l]));
        P.b.push_back(static_cast<double *>(bufs[5 + 3 * l]));
    }
}

/* Can `device` address `peer_device`'s memory (hipDeviceCanAccessPeer; the same device: yes)?  Asked BEFORE peer mode
 * is tried: a store through a mapping the hardware cannot serve is a memory fault, not an error code. */
int omg_peer_access(int device, int peer_device, int *can) {
    return guarded([&] {
        OMG_REQUIRE(can, "null argument");
        *can = 0;
        if (device == peer_device) { *can = 1; return; }
        int c = 0;
        OMG_HIP(hipDeviceCanAccessPeer(&c, device, peer_device));
        *can = c;
    });
}

int omg_pdist_p2p_handle_count(omg_pdist *d, int *count) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && count, "null argument");
        *count = 3 + 3 * int(d->d->lv.size());
    });
}

int omg_pdist_p2p_handles(omg_pdist *d, void *handles64, int capacity) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && handles64, "null argument");
        static_assert(sizeof(hipIpcMemHandle_t) == 64, "IPC handle size");
        std::vector<void *> bufs;
        std::vector<size_t> shift;
        pd_own_buffers(d->d.get(), bufs, &shift);
        OMG_REQUIRE(capacity >= int(bufs.size()), "handle buffer too small");
        OMG_HIP(hipStreamSynchronize(d->d->stream));
        for (size_t i = 0; i < bufs.size(); ++i) {
            hipIpcMemHandle_t h;
            OMG_HIP(hipIpcGetMemHandle(&h, static_cast<char *>(bufs[i]) - DEVBUF_SLACK - shift[i]));     // (the allocation's base)
            std::memcpy(static_cast<char *>(handles64) + 64 * i, &h, 64);
        }
    });
}

int omg_pdist_p2p_open(omg_pdist *d, int peer_rank, const void *handles64, int count) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && handles64, "null argument");
        PlaneDist *dd = d->d.get();
        OMG_REQUIRE(peer_rank >= 0 && peer_rank < dd->n_ranks && peer_rank != dd->rank, "bad peer rank");
        OMG_REQUIRE(count == 3 + 3 * int(dd->lv.size()), "handle count does not match the levels");
        PDPeer &P = dd->peers[size_t(peer_rank)];
        OMG_REQUIRE(P.mapped.empty(), "peer already opened");
        std::vector<void *> bufs, own;
        std::vector<size_t> shift;
        pd_own_buffers(dd, own, &shift);                      // (the peer's vectors sit in their allocations as mine do)
        // (vectors that share an allocation share a handle: it is opened once)
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
        pd_attach(dd, peer_rank, bufs);
    });
}

int omg_pdist_p2p_local(omg_pdist *d, omg_pdist *other) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && other && other->d, "null argument");
        OMG_REQUIRE(other->d->n_ranks == d->d->n_ranks && other->d->lv.size() == d->d->lv.size() && other->d->rank != d->d->rank,
                    "not another rank of the same decomposition");
        std::vector<void *> bufs;
        pd_own_buffers(other->d.get(), bufs);
        pd_attach(d->d.get(), other->d->rank, bufs);
    });
}

int omg_pdist_p2p_enable(omg_pdist *d, int mode) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && mode >= 0 && mode <= 2, "bad argument");
        PlaneDist *dd = d->d.get();
        if (mode == 0) { dd->p2p = 0; return; }
        for (const PDLevel &L : dd->lv) OMG_REQUIRE(L.nzo >= 4, "peer mode needs at least four planes per rank on every distributed level");
        const int64_t own = int64_t(dd->cnx) * dd->cny * dd->cnzo;
        std::vector<double *> dst[2];
        std::vector<uint32_t *> fl;
        for (int r = 0; r < dd->n_ranks; ++r) {
            const bool self = r == dd->rank;
            if (!self) OMG_REQUIRE(dd->peers[size_t(r)].flags, "peer mode: a rank's buffers have not been opened");
            dst[0].push_back((self ? dd->full_b.p : dd->peers[size_t(r)].full_b[0]) + int64_t(dd->rank) * own);
            dst[1].push_back((self ? dd->full_b2.p : dd->peers[size_t(r)].full_b[1]) + int64_t(dd->rank) * own);
            fl.push_back((self ? dd->flags.p : dd->peers[size_t(r)].flags) + PF_GATHER * 2 + dd->rank);
        }
        OMG_REQUIRE(PF_GATHER * 2 + dd->n_ranks <= PD_FLAGS, "too many ranks for the flag table");
        for (int par = 0; par < 2; ++par) {
            dd->ag_dst[par].alloc(dst[par].size());
            dd->ag_dst[par].upload(dst[par].data(), dst[par].size(), dd->stream);
        }
        dd->ag_flag.alloc(fl.size());
        dd->ag_flag.upload(fl.data(), fl.size(), dd->stream);
        OMG_HIP(hipStreamSynchronize(dd->stream));
        dd->p2p = mode;
    });
}

/* bit 0: a wait for a neighbour's flag gave up since the last call (the results since then are not to be used);
 * bit 1: a wave of a pass gave up waiting for a neighbouring wave of its own workgroup */
int omg_pdist_p2p_status(omg_pdist *d, unsigned *status) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && status, "null argument");
        PlaneDist *dd = d->d.get();
        uint32_t v = 0;
        OMG_HIP(hipMemcpyAsync(&v, dd->flags.p + PD_STATUS, 4, hipMemcpyDeviceToHost, dd->stream));
        OMG_HIP(hipMemsetAsync(dd->flags.p + PD_STATUS, 0, 4, dd->stream));
        OMG_HIP(hipStreamSynchronize(dd->stream));
        *status = v;
    });
}

static void pd_put(PlaneDist *d, const double *host, double *ext) {
    PDLevel &L = d->lv[0];
    const int64_t n = int64_t(L.nx) * L.ny * L.nzo;
    if (host) {
        d->nat.upload(host, size_t(n), d->stream);
        hipLaunchKernelGGL(pd_scatter_kernel, dim3(1024), dim3(256), 0, d->stream, d->nat.p, ext, L.nx, L.ny, L.nzo, PD_GHOST, 1);
        OMG_HIP(hipGetLastError());
        OMG_HIP(hipStreamSynchronize(d->stream));           // (nat is reused by the next vector)
    } else {
        OMG_HIP(hipMemsetAsync(ext, 0, size_t(L.n_ext) * sizeof(double), d->stream));
    }
}

/* b_local, x0_local: this rank's planes of the finest level in natural order (x0 NULL: zeros).  With a
 * communicator of more than one rank the call is COLLECTIVE (the ghost planes of b are exchanged). */
int omg_pdist_load(omg_pdist *d, const double *b_local, const double *x0_local) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && b_local, "null argument");
        PlaneDist *dd = d->d.get();
        OMG_HIP(hipStreamSynchronize(dd->side));                  // (an exchange posted behind the last cycle may still be writing ghost planes)
        dd->x0_posted = false;                                    // (the next cycle exchanges the ghost planes of the new iterate itself)
        pd_put(dd, b_local, dd->lv[0].b.p);
        pd_put(dd, x0_local, dd->lv[0].xp);
        if (dd->comm && dd->n_ranks > 1 && !dd->p2p) {            // (peer mode hands the ghost planes over at every batch start)
            PDExchange ex;
            ex.ranks = {dd};
            ex.halo(0, 1, 2);
        }
        OMG_HIP(hipStreamSynchronize(dd->stream));
    });
}

int omg_pdist_fetch(omg_pdist *d, double *x_local) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && x_local, "null argument");
        PlaneDist *dd = d->d.get();
        PDLevel &L = dd->lv[0];
        const int64_t n = int64_t(L.nx) * L.ny * L.nzo;
        hipLaunchKernelGGL(pd_scatter_kernel, dim3(1024), dim3(256), 0, dd->stream, dd->nat.p, L.xp, L.nx, L.ny, L.nzo, PD_GHOST, 0);
        OMG_HIP(hipGetLastError());
        dd->nat.download(x_local, size_t(n), dd->stream);
        OMG_HIP(hipStreamSynchronize(dd->stream));
    });
}

/* Tracing on: the stream writes a progress word between the phases of a cycle; omg_pdist_progress reads it WITHOUT
 * synchronising — (cycle << 16) | (level << 8) | phase (1 halo x, 2 halo b, 3 down pass, 4 halo x for the up pass,
 * 5 gather + replicated tail, 6 halo of the correction, 7 up pass): where the device is when a collective hangs. */
int omg_pdist_trace(omg_pdist *d, int enable) {
    return guarded([&] { OMG_REQUIRE(d && d->d, "null"); d->d->trace = enable != 0; });
}
int omg_pdist_progress(omg_pdist *d, unsigned *word) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && word, "null");
        *word = d->d->progress ? *reinterpret_cast<volatile uint32_t *>(d->d->progress) : 0u;
    });
}

/* out8: distributed levels; 1 when the finest level's passes can run gated (inner + edge chunks in one launch); its tile
 * (cells per line, lines, planes per chunk), workgroups, threads per workgroup, planes per inner chunk of a gated pass */
int omg_pdist_info(omg_pdist *d, int64_t *out8) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && out8, "null");
        const PlaneDist *dd = d->d.get();
        const PlanePlan<double> &P = dd->lv[0].plan;
        const int64_t v[8] = {int64_t(dd->lv.size()), (dd->gate && !dd->split && P.can_gate()) ? 1 : 0, P.g.TX, P.g.TY, P.g.LZ, P.g.n_wg, P.g.threads,
                              P.can_split() ? P.gate_lz() : 0};
        for (int i = 0; i < 8; ++i) out8[i] = v[i];
    });
}

int omg_pdist_set_gate(omg_pdist *d, int enable) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d, "null");
        PlaneDist *dd = d->d.get();
        OMG_HIP(hipStreamSynchronize(dd->stream));
        OMG_HIP(hipStreamSynchronize(dd->side));
        dd->gate = enable != 0;
        dd->x0_posted = false;                                    // (the next cycle exchanges its ghost planes in stream order first)
    });
}

int omg_pdist_sync(omg_pdist *d) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d, "null");
        OMG_HIP(hipStreamSynchronize(d->d->stream));
        OMG_HIP(hipStreamSynchronize(d->d->side));
    });
}

/* n_cycles V(1,1) cycles, every cycle's global residual norm computed and returned; collective. */
static int pd_cycles(omg_pdist *d, int pre, int post, int n_cycles, double *norms, bool squares_only) {
    return guarded([&] {
        OMG_REQUIRE(d && d->d && n_cycles >= 0, "bad argument");
        PlaneDist *dd = d->d.get();
        OMG_REQUIRE(dd->tail, "omg_pdist_set_tail has not been called");
        OMG_REQUIRE(dd->n_ranks == 1 || dd->comm || (dd->p2p && squares_only), "omg_pdist_connect has not been called");
        if (n_cycles == 0) return;
        if (dd->norms.n < size_t(n_cycles)) dd->norms.alloc(size_t(n_cycles));
        PDExchange ex;
        ex.ranks = {dd};
        ex.pre = pre; ex.post = post;
        if (dd->p2p) ex.prime();
        for (int k = 0; k < n_cycles; ++k) ex.cycle(dd->norms.p + k, k == 0);
        if (squares_only) { OMG_REQUIRE(norms, "null argument"); }
        else ex.norms_of_batch(dd->norms.p, n_cycles);
        if (norms) OMG_HIP(hipMemcpyAsync(norms, dd->norms.p, size_t(n_cycles) * sizeof(double), hipMemcpyDeviceToHost, dd->stream));
        OMG_HIP(hipStreamSynchronize(dd->stream));
    });
}

int omg_pdist_cycles(omg_pdist *d, int n_cycles, double *norms) { return pd_cycles(d, 1, 1, n_cycles, norms, false); }
/* V(pre, post) with pre, post in {0, 1} (openmg/__init__.py:22-23: the reference's default is V(1, 0)); RCCL exchanges only */
int omg_pdist_cycles_ex(omg_pdist *d, int pre, int post, int n_cycles, double *norms) { return pd_cycles(d, pre, post, n_cycles, norms, false); }
/* The same cycles; squares[k] = THIS rank's sum of squared residuals after cycle k — no collective (peer mode without
 * a communicator: the caller adds the ranks' values and takes the root). */
int omg_pdist_cycles_squares(omg_pdist *d, int n_cycles, double *squares) { return pd_cycles(d, 1, 1, n_cycles, squares, true); }

/* All ranks of a decomposition in ONE process on one GPU: the same launches per rank, device copies in place of
 * the RCCL exchanges (verification of the schedule without several GPUs).  The ranks run on rank 0's stream. */
int omg_pdist_group_create(int n, omg_pdist **ranks, omg_pdist_group **out) {
    return guarded([&] {
        OMG_REQUIRE(n >= 1 && ranks && out, "bad argument");
        std::unique_ptr<omg_pdist_group> g(new omg_pdist_group);
        for (int r = 0; r < n; ++r) {
            OMG_REQUIRE(ranks[r] && ranks[r]->d && ranks[r]->d->rank == r && ranks[r]->d->n_ranks == n, "ranks must be 0 .. n-1 of an n-rank decomposition");
            OMG_HIP(hipStreamSynchronize(ranks[r]->d->stream));
            ranks[r]->d->stream = ranks[0]->d->own;
            ranks[r]->d->side = ranks[0]->d->side_own;
            g->ranks.push_back(ranks[r]);
        }
        *out = g.release();
    });
}

int omg_pdist_group_destroy(omg_pdist_group *g) {
    delete g;
    return OMG_OK;
}

int omg_pdist_group_cycles_ex(omg_pdist_group *g, int pre, int post, int n_cycles, double *norms);
int omg_pdist_group_cycles(omg_pdist_group *g, int n_cycles, double *norms) { return omg_pdist_group_cycles_ex(g, 1, 1, n_cycles, norms); }

int omg_pdist_group_cycles_ex(omg_pdist_group *g, int pre, int post, int n_cycles, double *norms) {
    return guarded([&] {
        OMG_REQUIRE(g && !g->ranks.empty() && n_cycles >= 0, "bad argument");
        PDExchange ex;
        ex.loopback = true;
        ex.pre = pre; ex.post = post;
        for (omg_pdist *r : g->ranks) {
            OMG_REQUIRE(r->d->tail, "omg_pdist_set_tail has not been called on every rank");
            ex.ranks.push_back(r->d.get());
        }
        if (n_cycles == 0) return;
        PlaneDist *z = ex.ranks[0];
        if (z->norms.n < size_t(n_cycles) * ex.ranks.size()) z->norms.alloc(size_t(n_cycles) * ex.ranks.size());
        for (PlaneDist *r : ex.ranks) OMG_REQUIRE(r->p2p == z->p2p, "the ranks of a group must all be in the same mode");
        if (ex.p2p()) ex.prime();
        else ex.halo(0, 1, 2);                                // ghost planes of the right-hand side
        std::vector<double *> outs(ex.ranks.size());
        for (int k = 0; k < n_cycles; ++k) {
            ex.cycle(nullptr, k == 0);
            for (size_t r = 0; r < ex.ranks.size(); ++r) outs[r] = z->norms.p + size_t(k) * ex.ranks.size() + r;
            ex.norm(outs.data());
        }
        std::vector<double> all(size_t(n_cycles) * ex.ranks.size());
        OMG_HIP(hipMemcpyAsync(all.data(), z->norms.p, all.size() * sizeof(double), hipMemcpyDeviceToHost, z->stream));
        OMG_HIP(hipStreamSynchronize(z->stream));
        for (int k = 0; k < n_cycles; ++k) {
            for (size_t r = 1; r < ex.ranks.size(); ++r)
                OMG_REQUIRE(all[size_t(k) * ex.ranks.size() + r] == all[size_t(k) * ex.ranks.size()], "internal: ranks disagree about the norm");
            if (norms) norms[k] = all[size_t(k) * ex.ranks.size()];
        }
    });
}

}  // extern "C"
