// Host-side setup: smoother orderings, symmetric permutation of CSR matrices, transpose,
// row-block partition.  Index work only — no floating-point arithmetic happens here.
#include <algorithm>
#include <atomic>
#include <cstring>
#include <string>
#include <map>
#include <memory>
#include <mutex>
#include <thread>
#include <unordered_map>
#include <type_traits>

#include "common.h"

namespace omg {

static unsigned host_threads() {
    unsigned t = std::thread::hardware_concurrency();
    if (t == 0) t = 4;
    return std::min(t, 32u);
}

template <typename F>
static void parallel_rows(int64_t n, F &&body) {
    unsigned nt = host_threads();
    if (n < (1 << 16) || nt == 1) {
        body(int64_t(0), n);
        return;
    }
    std::vector<std::thread> pool;
    int64_t chunk = (n + nt - 1) / nt;
    for (unsigned t = 0; t < nt; ++t) {
        int64_t lo = t * chunk, hi = std::min<int64_t>(n, lo + chunk);
        if (lo >= hi) break;
        pool.emplace_back([&body, lo, hi] { body(lo, hi); });
    }
    for (auto &th : pool) th.join();
}

void validate_csr(const omg_csr &A, const char *what) {
    std::string w(what);
    OMG_REQUIRE(A.n_rows >= 0 && A.n_cols >= 0 && A.nnz >= 0, w + ": negative dimension");
    OMG_REQUIRE(A.n_rows < INT32_MAX && A.n_cols < INT32_MAX && A.nnz < INT32_MAX,
                w + ": int32 index range exceeded (shard the operator first)");
    OMG_REQUIRE(A.indptr != nullptr, w + ": indptr is null");
    OMG_REQUIRE(A.nnz == 0 || (A.indices && A.data), w + ": indices/data is null");
    OMG_REQUIRE(A.indptr[0] == 0 && A.indptr[A.n_rows] == A.nnz, w + ": indptr does not span nnz");
    bool ok = true;
    parallel_rows(A.n_rows, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi && ok; ++i) {
            if (A.indptr[i] > A.indptr[i + 1]) { ok = false; break; }
            for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p)
                if (A.indices[p] < 0 || A.indices[p] >= A.n_cols) { ok = false; break; }
        }
    });
    OMG_REQUIRE(ok, w + ": malformed CSR (indptr not monotone or column out of range)");
}

int64_t first_row_without_diagonal(const omg_csr &A) {
    std::atomic<int64_t> bad(INT64_MAX);
    parallel_rows(A.n_rows, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            double d = 0.0;
            bool have = false;
            for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p)
                if (A.indices[p] == i) { d += A.data[p]; have = true; }
            if (!have || d == 0.0) {
                int64_t cur = bad.load();
                while (i < cur && !bad.compare_exchange_weak(cur, i)) {}
                return;
            }
        }
    });
    const int64_t v = bad.load();
    return v == INT64_MAX ? -1 : v;
}

// ---- orderings ------------------------------------------------------------------------
// GS_LEX: level schedule of the lexicographic sweep.  level(i) is the smallest value that is
// larger than level(j) for every coupled j < i; "coupled" means a_ij != 0 or a_ji != 0
// structurally.  Row i's own pattern supplies the j < i lower bounds directly and pushes a
// lower bound onto every j > i it references, so unsymmetric patterns are covered in one
// ascending pass.  Rows of one level are mutually uncoupled, every row sees the NEW value
// of its lower neighbours and the OLD value of its upper neighbours: the same iterate as
// the sequential loop at openmg/solvers.py:56-68.
static void lex_levels(const omg_csr &A, std::vector<int32_t> &key, int32_t &n_keys) {
    const int64_t n = A.n_rows;
    key.assign(n, 0);
    int32_t top = 0;
    for (int64_t i = 0; i < n; ++i) {
        int32_t lv = key[i];
        for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) {
            int32_t j = A.indices[p];
            if (j < i && key[j] + 1 > lv) lv = key[j] + 1;
        }
        key[i] = lv;
        if (lv > top) top = lv;
        for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) {
            int32_t j = A.indices[p];
            if (j > i && j < n && key[j] < lv + 1) key[j] = lv + 1;
        }
    }
    n_keys = n ? top + 1 : 0;
}

// GS_COLOUR: smallest-free-colour greedy colouring in natural row order over the graph of
// A + A^T (same push trick for the transposed couplings).  Up to 64 colours.
static void greedy_colours(const omg_csr &A, std::vector<int32_t> &key, int32_t &n_keys) {
    const int64_t n = A.n_rows;
    key.assign(n, 0);
    std::vector<uint64_t> forbidden(n, 0);
    int32_t top = 0;
    for (int64_t i = 0; i < n; ++i) {
        uint64_t f = forbidden[i];
        for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) {
            int32_t j = A.indices[p];
            if (j < i) f |= (uint64_t(1) << key[j]);
        }
        if (f == ~uint64_t(0))
            throw Error(OMG_ERR_UNSUPPORTED, "greedy colouring needs more than 64 colours; use GS_LEX");
        int32_t c = __builtin_ctzll(~f);
        key[i] = c;
        if (c > top) top = c;
        for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) {
            int32_t j = A.indices[p];
            if (j > i && j < n) forbidden[j] |= (uint64_t(1) << c);
        }
    }
    n_keys = n ? top + 1 : 0;
}

Ordering make_ordering(const omg_csr &A, int smoother) {
    const int64_t n = A.n_rows;
    Ordering o;
    if (smoother == OMG_SMOOTH_JACOBI || n == 0) {
        o.sets = {0, n};
        o.identity = true;
        return o;
    }
    OMG_REQUIRE(A.n_rows == A.n_cols, "smoother needs a square operator");
    std::vector<int32_t> key;
    int32_t n_keys = 0;
    if (smoother == OMG_SMOOTH_GS_LEX) lex_levels(A, key, n_keys);
    else if (smoother == OMG_SMOOTH_GS_COLOUR) greedy_colours(A, key, n_keys);
    else throw Error(OMG_ERR_INVALID, "unknown smoother kind");
    return ordering_from_keys(key.data(), n, n_keys);
}

Ordering ordering_from_keys(const int32_t *keys, int64_t n, int32_t n_keys) {
    Ordering o;
    if (keys == nullptr || n_keys <= 0) {
        o.sets = {0, n};
        o.identity = true;
        return o;
    }
    for (int64_t i = 0; i < n; ++i)
        OMG_REQUIRE(keys[i] >= 0 && keys[i] < n_keys, "ordering key out of range");
    // stable counting sort by key
    o.sets.assign(size_t(n_keys) + 1, 0);
    for (int64_t i = 0; i < n; ++i) o.sets[keys[i] + 1]++;
    for (int32_t k = 0; k < n_keys; ++k) o.sets[k + 1] += o.sets[k];
    std::vector<int64_t> cursor(o.sets.begin(), o.sets.end() - 1);
    o.perm.resize(n);
    o.inv.resize(n);
    bool ident = true;
    for (int64_t i = 0; i < n; ++i) {
        int64_t pos = cursor[keys[i]]++;
        o.perm[pos] = int32_t(i);
        o.inv[i] = int32_t(pos);
        if (pos != i) ident = false;
    }
    o.identity = ident;
    if (ident) { o.perm.clear(); o.inv.clear(); }
    return o;
}

// ---- permutation / transpose ----------------------------------------------------------
HostCsr permute_csr(const omg_csr &A, const int32_t *row_perm, const int32_t *col_inv,
                    int64_t n_inv) {
    HostCsr out;
    out.n_rows = A.n_rows;
    out.n_cols = A.n_cols;
    out.nnz = A.nnz;
    out.indptr.resize(A.n_rows + 1);
    out.indices.resize(A.nnz);
    out.data.resize(A.nnz);
    out.indptr[0] = 0;
    for (int64_t i = 0; i < A.n_rows; ++i) {
        int64_t src = row_perm ? row_perm[i] : i;
        out.indptr[i + 1] = out.indptr[i] + (A.indptr[src + 1] - A.indptr[src]);
    }
    parallel_rows(A.n_rows, [&](int64_t lo, int64_t hi) {
        for (int64_t i = lo; i < hi; ++i) {
            int64_t src = row_perm ? row_perm[i] : i;
            int32_t q = out.indptr[i];
            for (int32_t p = A.indptr[src]; p < A.indptr[src + 1]; ++p, ++q) {
                int32_t c = A.indices[p];
                out.indices[q] = (col_inv && (n_inv < 0 || c < n_inv)) ? col_inv[c] : c;   // stored order inside the row kept
                out.data[q] = A.data[p];
            }
        }
    });
    return out;
}

HostCsr transpose_csr(const HostCsr &A) {
    HostCsr T;
    T.n_rows = A.n_cols;
    T.n_cols = A.n_rows;
    T.nnz = A.nnz;
    T.indptr.assign(T.n_rows + 1, 0);
    T.indices.resize(A.nnz);
    T.data.resize(A.nnz);
    for (int64_t p = 0; p < A.nnz; ++p) T.indptr[A.indices[p] + 1]++;
    for (int64_t i = 0; i < T.n_rows; ++i) T.indptr[i + 1] += T.indptr[i];
    std::vector<int32_t> cursor(T.indptr.begin(), T.indptr.end() - 1);
    // ascending source row => ascending column inside every transposed row: the order in
    // which SciPy's csc_matvec accumulates R^T e (openmg/__init__.py:214).
    for (int64_t i = 0; i < A.n_rows; ++i)
        for (int32_t p = A.indptr[i]; p < A.indptr[i + 1]; ++p) {
            int32_t q = cursor[A.indices[p]]++;
            T.indices[q] = int32_t(i);
            T.data[q] = A.data[p];
        }
    return T;
}

void make_row_blocks(const IndexVec &indptr, const std::vector<int64_t> &sets,
                     int max_rows, int max_nnz, std::vector<int32_t> &blk_rows,
                     std::vector<int64_t> &set_blk) {
    blk_rows.clear();
    set_blk.assign(1, 0);
    for (size_t s = 0; s + 1 < sets.size(); ++s) {
        int64_t r = sets[s];
        const int64_t end = sets[s + 1];
        while (r < end) {
            blk_rows.push_back(int32_t(r));
            int64_t r1 = r + 1;                       // a block always takes at least one row
            const int64_t p0 = indptr[r];
            while (r1 < end && r1 - r < max_rows && indptr[r1 + 1] - p0 <= max_nnz) ++r1;
            r = r1;
        }
        set_blk.push_back(int64_t(blk_rows.size()));
    }
    blk_rows.push_back(int32_t(sets.empty() ? 0 : sets.back()));
}

namespace {

// OMG_COMPRESS, a bit mask: 1 = per-entry column dictionaries, 2 = per-entry value
// dictionaries, 4 = whole-row pattern dictionaries, 8 = offset patterns with the values in a
// block-transposed (ELL) array (needs 4); 0 = plain CSR only; default 15.  Results are
// bit-identical in every mode (tests/test_gpu_parity.py).
int compress_mode() {
    const char *e = getenv("OMG_COMPRESS");
    if (!e || !e[0]) return 15;
    const int v = atoi(e);
    return (v < 0 || v > 15) ? 15 : v;
}

// rows per thread of the union walk (csr_kernels.hip rows_union_kernel): 0 = kernel off
int union_rows() {
    const char *k = getenv("OMG_UNION_KERNEL");
    if (k && k[0] == '0') return 0;
    const char *p = getenv("OMG_PATTERN_KERNEL");            // rows_kernel everywhere: standard blocks
    if (p && p[0] == '0') return 0;
    const char *e = getenv("OMG_UNION_ROWS");
    const int v = e ? atoi(e) : 2;          // measured at 256^3 (7-point, red-black): 2 > 4 > 1
    return v >= 4 ? 4 : v >= 2 ? 2 : 1;
}

template <typename V> struct Bits;
template <> struct Bits<double> {
    typedef uint64_t type;
    static uint64_t of(double v) { uint64_t u; std::memcpy(&u, &v, sizeof u); return u; }
};
template <> struct Bits<float> {
    typedef uint32_t type;
    static uint32_t of(float v) { uint32_t u; std::memcpy(&u, &v, sizeof u); return u; }
};

// At most DICT_MAX distinct keys in order of first appearance; code() returns -1 once full.
template <typename K>
struct SmallDict {
    K key[DICT_MAX];
    int n = 0, last = 0;
    int code(K k) {
        if (n && key[last] == k) return last;
        for (int i = 0; i < n; ++i)
            if (key[i] == k) { last = i; return i; }
        if (n == DICT_MAX) return -1;
        key[n] = k;
        last = n;
        return n++;
    }
};

}  // namespace

namespace {

// One attempt at coding A with a given row-block partition (see encode_csr).
template <typename V>
HostFormat<V> encode_with(const HostCsr &A, const std::vector<int64_t> &sets_in, int max_rows, int max_nnz,
                          int lanes, int mode, bool need_union = false) {
    HostFormat<V> F;
    auto &sets = F.sets;
    auto &set_blk = F.set_blk;
    const int64_t n_rows = A.n_rows, nnz = A.nnz;
    sets = sets_in;
    if (sets.empty()) sets = {0, n_rows};
    F.rows_cap = max_rows;
    F.lanes_per_row = lanes;
    std::vector<int32_t> blocks;
    F.set_nnz.assign(sets.size() - 1, 0);
    for (size_t k = 0; k + 1 < sets.size(); ++k) F.set_nnz[k] = A.indptr[sets[k + 1]] - A.indptr[sets[k]];
    F.set_maxlen.assign(sets.size() - 1, 0);
    for (size_t k = 0; k + 1 < sets.size(); ++k)
        for (int64_t r = sets[k]; r < sets[k + 1]; ++r)
            F.set_maxlen[k] = std::max<int32_t>(F.set_maxlen[k], A.indptr[r + 1] - A.indptr[r]);
    make_row_blocks(A.indptr, sets, max_rows, max_nnz, blocks, set_blk);
    // float operator: round the fp64 entries once, here
    if constexpr (!std::is_same<V, double>::value) F.narrowed.assign(A.data.begin(), A.data.end());
    const V *vals;
    if constexpr (std::is_same<V, double>::value) vals = A.data.data();
    else vals = F.narrowed.data();

    // ---- phase A (parallel over blocks): what each block COULD be coded as -----------------------
    const int64_t nblk = int64_t(blocks.size()) - 1;          // `blocks` ends with a sentinel row
    auto &cc = F.cc;
    auto &vc = F.vc;
    auto &rc = F.rc;
    const size_t nb = size_t(std::max<int64_t>(nblk, 0));
    struct PatDict {
        std::vector<int32_t> beg, idx;     // beg: npat + 1 offsets into idx / val
        std::vector<V> val;                // empty: offsets only (values go to the ELL array)
        int maxlen = 0;
        // union walk (common.h UNION_MAX): a common supersequence of all patterns and, per pattern,
        // the mask of the union slots it occupies; uidx empty = none
        std::vector<int32_t> uidx, umask;
        std::vector<V> uval;
        bool wave_sized() const { return !idx.empty() && int(beg.size()) - 1 < 64 && int(idx.size()) <= PAT_LANE_ENTRIES; }
        // Shortest common supersequence, pattern by pattern (each pattern stays a subsequence of the
        // growing union because the union only ever gains entries); gives up beyond UNION_MAX.
        void make_union() {
            uidx.clear(); uval.clear(); umask.clear();
            if (idx.empty() || val.size() != idx.size()) return;
            const int npat = int(beg.size()) - 1;
            if (npat > 64) return;
            auto eq = [&](int ui, int pi) { return uidx[ui] == idx[pi] && Bits<V>::of(uval[ui]) == Bits<V>::of(val[pi]); };
            // start from the longest pattern: most others are subsequences of it
            int longest = 0;
            for (int c = 1; c < npat; ++c) if (beg[c + 1] - beg[c] > beg[longest + 1] - beg[longest]) longest = c;
            uidx.assign(idx.begin() + beg[longest], idx.begin() + beg[longest + 1]);
            uval.assign(val.begin() + beg[longest], val.begin() + beg[longest + 1]);
            for (int c = 0; c < npat; ++c) {
                const int pb = beg[c], len = beg[c + 1] - beg[c], ul = int(uidx.size());
                int k = 0;
                for (int i = 0; i < ul && k < len; ++i) if (eq(i, pb + k)) ++k;
                if (k == len) continue;                              // already a subsequence
                // LCS table, then merge
                std::vector<int> L(size_t(ul + 1) * size_t(len + 1), 0);
                auto at = [&](int i, int j) -> int & { return L[size_t(i) * size_t(len + 1) + size_t(j)]; };
                for (int i = ul - 1; i >= 0; --i)
                    for (int j = len - 1; j >= 0; --j)
                        at(i, j) = eq(i, pb + j) ? at(i + 1, j + 1) + 1 : std::max(at(i + 1, j), at(i, j + 1));
                std::vector<int32_t> ni;
                std::vector<V> nv;
                int i = 0, j = 0;
                while (i < ul || j < len) {
                    if (i < ul && j < len && eq(i, pb + j)) { ni.push_back(uidx[i]); nv.push_back(uval[i]); ++i; ++j; }
                    else if (j == len || (i < ul && at(i + 1, j) >= at(i, j + 1))) { ni.push_back(uidx[i]); nv.push_back(uval[i]); ++i; }
                    else { ni.push_back(idx[pb + j]); nv.push_back(val[pb + j]); ++j; }
                }
                if (int(ni.size()) > UNION_MAX) { uidx.clear(); uval.clear(); return; }
                uidx.swap(ni);
                uval.swap(nv);
            }
            if (int(uidx.size()) > UNION_MAX) { uidx.clear(); uval.clear(); return; }
            umask.assign(size_t(npat), 0);
            for (int c = 0; c < npat; ++c) {
                const int pb = beg[c], len = beg[c + 1] - beg[c];
                int k = 0;
                for (int i = 0; i < int(uidx.size()) && k < len; ++i)
                    if (eq(i, pb + k)) { umask[c] |= int32_t(1) << i; ++k; }
                if (k != len) { uidx.clear(); uval.clear(); umask.clear(); return; }   // cannot happen
            }
        }
    };
    std::vector<PatDict> pfull(nb), pcols(nb);
    std::vector<std::vector<int32_t>> cdicts(nb);
    std::vector<std::vector<V>> vdicts(nb);
    // row patterns: not for the several-rows-per-thread operators (prolongation, restriction)
    const bool try_pat = (mode & 4) && (max_rows <= ROWBLK_THREADS || need_union);
    const bool try_ell = try_pat && (mode & 8) && lanes == 1;
    if (mode != 0 && nblk > 0 && nnz > 0) {
        if (mode & 1) cc.assign(size_t(nnz), 0);
        if (mode & 2) vc.assign(size_t(nnz), 0);
        if (try_pat) rc.assign(size_t(n_rows), 0);
        // rows of [r0, r1) as patterns of (column - row) offsets, with their values (full) or without
        auto patterns = [&](int64_t r0, int64_t r1, bool with_values, int64_t budget, PatDict &out) {
            PatDict d;
            d.beg.push_back(0);
            int last = -1;
            for (int64_t r = r0; r < r1; ++r) {
                const int64_t q0 = A.indptr[r];
                const int len = int(A.indptr[r + 1] - q0);
                auto same = [&](int c) {
                    const int b = d.beg[c];
                    if (d.beg[c + 1] - b != len) return false;
                    for (int j = 0; j < len; ++j) {
                        if (d.idx[b + j] != A.indices[q0 + j] - int32_t(r)) return false;
                        if (with_values && Bits<V>::of(d.val[b + j]) != Bits<V>::of(vals[q0 + j])) return false;
                    }
                    return true;
                };
                int code = -1;
                const int npat = int(d.beg.size()) - 1;
                if (last >= 0 && same(last)) code = last;
                for (int c = 0; c < npat && code < 0; ++c)
                    if (c != last && same(c)) code = c;
                if (code < 0) {
                    if (npat == DICT_MAX || int64_t(d.idx.size()) + len > budget) return false;
                    for (int j = 0; j < len; ++j) {
                        d.idx.push_back(A.indices[q0 + j] - int32_t(r));
                        if (with_values) d.val.push_back(vals[q0 + j]);
                    }
                    d.beg.push_back(int32_t(d.idx.size()));
                    code = npat;
                }
                rc[r] = uint8_t(code);
                last = code;
                d.maxlen = std::max(d.maxlen, len);
            }
            out = std::move(d);
            return true;
        };
        auto work = [&](int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; ++k) {
                const int64_t r0 = blocks[k], r1 = blocks[k + 1];
                const int64_t p0 = A.indptr[r0], p1 = A.indptr[r1];
                if (p1 <= p0 || p1 - p0 > max_nnz) continue;          // empty, or one long row: plain
                // a small block is latency, not bytes: a dictionary would only add a round trip
                // (the single-row sets of a 1-D lexicographic sweep ran 20 % slower coded)
                if (p1 - p0 < MIN_CODED_ENTRIES) continue;
                const int64_t budget = (p1 - p0) / 2;                 // a dictionary must be a real saving
                if (try_pat && patterns(r0, r1, true, budget, pfull[k])) { pfull[k].make_union(); continue; }
                if (try_ell) patterns(r0, r1, false, budget, pcols[k]);
                if (p1 - p0 > ROWBLK_NNZ) continue;                   // (wide partition: no LDS fallback exists)
                if (mode & 1) {
                    SmallDict<int32_t> d;
                    bool ok = true;
                    for (int64_t r = r0; r < r1 && ok; ++r)
                        for (int64_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p) {
                            const int c = d.code(A.indices[p] - int32_t(r));
                            if (c < 0) { ok = false; break; }
                            cc[p] = uint8_t(c);
                        }
                    if (ok) cdicts[k].assign(d.key, d.key + d.n);
                }
                if (mode & 2) {
                    SmallDict<typename Bits<V>::type> d;
                    bool ok = true;
                    for (int64_t p = p0; p < p1; ++p) {
                        const int c = d.code(Bits<V>::of(vals[p]));
                        if (c < 0) { ok = false; break; }
                        vc[p] = uint8_t(c);
                    }
                    if (ok) {
                        vdicts[k].resize(d.n);
                        std::memcpy(vdicts[k].data(), d.key, size_t(d.n) * sizeof(V));
                    }
                }
            }
        };
        const int nthreads = int(std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())),
                                                   std::max<int64_t>(1, nblk / 256)));
        if (nthreads <= 1) {
            work(0, nblk);
        } else {
            std::vector<std::thread> pool;
            for (int t = 0; t < nthreads; ++t)
                pool.emplace_back(work, nblk * t / nthreads, nblk * (t + 1) / nthreads);
            for (auto &t : pool) t.join();
        }
    }

    // ---- phase B: which sets can run the LDS-free pattern kernel; final coding of every block ----
    // A set runs rows_pattern_kernel when EVERY block of it is a row-pattern block whose
    // dictionary fits the lanes of a wave (one row per thread: LPR 1).  Only there may a block
    // keep offset patterns with its values in the ELL array (rows_kernel has no path for it).
    F.set_pattern.assign(sets.size() - 1, 0);
    F.set_ell.assign(sets.size() - 1, 0);
    if (lanes == 1 && max_rows <= ROWBLK_THREADS)
        for (size_t q = 0; q + 1 < sets.size(); ++q) {
            bool all = set_blk[q + 1] > set_blk[q], ell = false;
            for (int64_t k = set_blk[q]; all && k < set_blk[q + 1]; ++k) {
                if (pfull[k].wave_sized()) continue;
                if (pcols[k].wave_sized()) { ell = true; continue; }
                all = false;
            }
            F.set_pattern[q] = all ? (ell ? 2 : 1) : 0;
            F.set_ell[q] = (all && ell) ? 1 : 0;
        }
    // union walk: every block of the set is a full (offsets + values) pattern block with a union
    F.set_union.assign(sets.size() - 1, 0);
    for (size_t q = 0; q + 1 < sets.size(); ++q) {
        bool all = set_blk[q + 1] > set_blk[q];
        for (int64_t k = set_blk[q]; all && k < set_blk[q + 1]; ++k) {
            all = !pfull[k].uidx.empty() && int(pfull[k].beg.size()) - 1 <= 64;
            if (all) F.union_max = std::max(F.union_max, int(pfull[k].uidx.size()));
        }
        F.set_union[q] = all ? 1 : 0;
    }
    if (need_union) {
        // several-rows-per-thread partition (encode_csr): blocks of more than ROWBLK_THREADS rows,
        // which only rows_union_kernel (and rows_kernel's walk of the dictionary) can run
        bool all = true;
        for (size_t q = 0; q + 1 < sets.size(); ++q) all = all && (F.set_union[q] != 0 || set_blk[q + 1] == set_blk[q]);
        if (!all) { F.union_failed = true; return F; }
        for (size_t q = 0; q + 1 < sets.size(); ++q) F.set_pattern[q] = 1;
        F.union_blocks = 1;
    } else if (max_nnz > ROWBLK_NNZ) {
        // wide partition (encode_csr): usable only if the pattern kernel takes every set
        bool all = true;
        for (size_t q = 0; q + 1 < sets.size(); ++q) all = all && (F.set_pattern[q] != 0 || set_blk[q + 1] == set_blk[q]);
        if (!all) { F.set_pattern.assign(sets.size() - 1, 0); F.wide_failed = true; return F; }
        for (auto &c : F.set_pattern) c = 2;     // mandatory: these blocks do not fit rows_kernel's LDS image
    }
    auto &cpool = F.cpool;
    auto &ppool_idx = F.ppool_idx;
    auto &ppool_beg = F.ppool_beg;
    auto &vpool = F.vpool;
    auto &ppool_val = F.ppool_val;
    auto &info = F.info;
    info.assign(size_t(BLK_INFO_INTS) * blocks.size(), 0);
    std::vector<int64_t> ell_off(nb, -1);
    int64_t ell_total = 0;
    {
        std::unordered_map<std::string, int32_t> cseen, vseen;
        std::unordered_map<std::string, std::pair<int32_t, int32_t>> pseen;    // -> (entry offset, table offset)
        const int64_t pool_cap = (int64_t(1) << (31 - DICT_SHIFT)) - DICT_MAX;
        size_t set_of = 0;
        for (int64_t k = 0; k <= nblk; ++k) {
            int32_t *rec = info.data() + size_t(BLK_INFO_INTS) * size_t(k);
            rec[0] = blocks[k];
            rec[1] = A.indptr[blocks[k]];
            if (k == nblk) break;
            while (set_of + 1 < set_blk.size() - 1 && k >= set_blk[set_of + 1]) ++set_of;
            const int64_t rows = blocks[k + 1] - blocks[k];
            const int64_t entries = A.indptr[blocks[k + 1]] - A.indptr[blocks[k]];
            const bool use_ell = pfull[k].idx.empty() && !pcols[k].idx.empty() && F.set_pattern[set_of] == 2;
            const PatDict &pd = use_ell ? pcols[k] : pfull[k];
            if (!pd.idx.empty()) {
                std::vector<V> padv(pd.idx.size(), V(0));                      // offsets-only dictionaries: zeros
                const V *dv = pd.val.empty() ? padv.data() : pd.val.data();
                std::string key(use_ell ? "E" : "F");               // an offsets-only dictionary carries no union behind it
                key.append(reinterpret_cast<const char *>(pd.beg.data()), pd.beg.size() * sizeof(int32_t));
                key.append(reinterpret_cast<const char *>(pd.idx.data()), pd.idx.size() * sizeof(int32_t));
                key.append(reinterpret_cast<const char *>(dv), pd.idx.size() * sizeof(V));
                auto it = pseen.find(key);
                if (it == pseen.end() && int64_t(ppool_idx.size()) < (int64_t(1) << 30)) {
                    it = pseen.emplace(std::move(key), std::make_pair(int32_t(ppool_idx.size()), int32_t(ppool_beg.size()))).first;
                    ppool_idx.insert(ppool_idx.end(), pd.idx.begin(), pd.idx.end());
                    ppool_val.insert(ppool_val.end(), dv, dv + pd.idx.size());
                    if (!use_ell && !pd.uidx.empty()) {                        // the union, then (below) its masks
                        ppool_idx.insert(ppool_idx.end(), pd.uidx.begin(), pd.uidx.end());
                        ppool_val.insert(ppool_val.end(), pd.uval.begin(), pd.uval.end());
                    }
                    while (ppool_idx.size() % 8) { ppool_idx.push_back(0); ppool_val.push_back(V(0)); }   // 16-B vector loads
                    ppool_beg.insert(ppool_beg.end(), pd.beg.begin(), pd.beg.end());
                    if (!use_ell && !pd.uidx.empty()) ppool_beg.insert(ppool_beg.end(), pd.umask.begin(), pd.umask.end());
                }
                OMG_REQUIRE(it != pseen.end(), "pattern dictionary pool overflow");
                rec[4] = it->second.first;
                rec[5] = int32_t(pd.idx.size());
                rec[6] = it->second.second;
                rec[7] = int32_t(pd.beg.size()) - 1;
                if (!use_ell) rec[3] = int32_t(pd.uidx.size());                 // union entries (0: none)
                ++F.blocks_pcoded;
                F.rows_pcoded += rows;
                F.nnz_pcoded += entries;
                if (use_ell) {
                    // values of the block transposed: entry j of every row side by side (coalesced)
                    OMG_REQUIRE(ell_total + rows * pd.maxlen < (int64_t(1) << 31) - 1, "ELL value array exceeds int32 offsets");
                    ell_off[k] = ell_total;
                    rec[2] = int32_t(ell_total) + 1;
                    rec[3] = pd.maxlen;
                    ell_total += rows * pd.maxlen;
                    ++F.blocks_ell;
                    F.nnz_ell += entries;
                }
                continue;
            }
            OMG_REQUIRE(max_nnz <= ROWBLK_NNZ, "a wide row block fell back to the LDS kernels");
            {   // exactly one entry in every row (prolongation): the kernels then skip the row pointers
                bool unit = entries == rows;
                for (int64_t r = blocks[k]; unit && r < blocks[k + 1]; ++r) unit = A.indptr[r + 1] - A.indptr[r] == 1;
                rec[6] = unit ? 1 : 0;
            }
            if (!cdicts[k].empty()) {
                std::string key(reinterpret_cast<const char *>(cdicts[k].data()), cdicts[k].size() * sizeof(int32_t));
                auto it = cseen.find(key);
                if (it == cseen.end() && int64_t(cpool.size()) < pool_cap) {
                    it = cseen.emplace(std::move(key), int32_t(cpool.size())).first;
                    cpool.insert(cpool.end(), cdicts[k].begin(), cdicts[k].end());
                }
                if (it != cseen.end()) {
                    rec[2] = (it->second << DICT_SHIFT) | int32_t(cdicts[k].size());
                    ++F.blocks_ccoded;
                    F.nnz_ccoded += entries;
                }
            }
            if (!vdicts[k].empty()) {
                std::string key(reinterpret_cast<const char *>(vdicts[k].data()), vdicts[k].size() * sizeof(V));
                auto it = vseen.find(key);
                if (it == vseen.end() && int64_t(vpool.size()) < pool_cap) {
                    it = vseen.emplace(std::move(key), int32_t(vpool.size())).first;
                    vpool.insert(vpool.end(), vdicts[k].begin(), vdicts[k].end());
                }
                if (it != vseen.end()) {
                    rec[3] = (it->second << DICT_SHIFT) | int32_t(vdicts[k].size());
                    ++F.blocks_vcoded;
                    F.nnz_vcoded += entries;
                }
            }
        }
    }
    // ---- phase C (parallel): the ELL value array -------------------------------------------------
    if (ell_total > 0) {
        F.vell.assign(size_t(ell_total), V(0));
        auto fill = [&](int64_t k0, int64_t k1) {
            for (int64_t k = k0; k < k1; ++k) {
                if (ell_off[k] < 0) continue;
                const int64_t r0 = blocks[k], rows = blocks[k + 1] - blocks[k];
                V *dst = F.vell.data() + ell_off[k];
                for (int64_t r = r0; r < r0 + rows; ++r)
                    for (int64_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p)
                        dst[(p - A.indptr[r]) * rows + (r - r0)] = vals[p];
            }
        };
        const int nthreads = int(std::min<int64_t>(std::max(1u, std::min(16u, std::thread::hardware_concurrency())),
                                                   std::max<int64_t>(1, nblk / 256)));
        std::vector<std::thread> pool;
        for (int t = 0; t < nthreads; ++t) pool.emplace_back(fill, nblk * t / nthreads, nblk * (t + 1) / nthreads);
        for (auto &t : pool) t.join();
    }
    return F;
}

}  // namespace

template <typename V>
HostFormat<V> encode_csr(const HostCsr &A, const std::vector<int64_t> &sets_in) {
    int mode = compress_mode();
    {   // OMG_PATTERN_KERNEL=0 (rows_kernel everywhere): no ELL blocks, rows_kernel cannot read them
        const char *e = getenv("OMG_PATTERN_KERNEL");
        if (e && e[0] == '0') mode &= ~8;
    }
    // One row per thread for stencil-like rows; for very short rows (prolongation: one entry
    // per row) let a block take as many rows as fit its LDS budget so that each workgroup
    // still moves tens of KB.
    int max_rows = ROWBLK_ROWS, lanes = 1;
    if (A.n_rows > 0) {
        const double avg = double(A.nnz) / double(A.n_rows);
        while (max_rows < ROWBLK_NNZ && avg * (2 * max_rows) <= ROWBLK_NNZ) max_rows *= 2;
        // long rows (27-point stencils ...): one thread per row would leave most of the
        // workgroup idle in rows_kernel's row phase, so four lanes share a row and a block holds
        // at most ROWBLK_THREADS / 4 rows per pass
        const char *e = experiment_env("OMG_LANES_PER_ROW");
        lanes = e ? atoi(e) : (avg > 16.0 ? 4 : 1);
        if (lanes != 4) lanes = 1;
        if (lanes == 4) max_rows = ROWBLK_THREADS / 4;
        // ... unless the LDS-free pattern kernel can take the WHOLE operator: it has no LDS image
        // to fit, so its blocks are a full 256 rows whatever the row length.  Tried first.
        if (lanes == 4 && (mode & 4) && !(getenv("OMG_PATTERN_KERNEL") && getenv("OMG_PATTERN_KERNEL")[0] == '0')) {
            HostFormat<V> W = encode_with<V>(A, sets_in, ROWBLK_THREADS, 1 << 24, 1, mode);
            if (!W.wide_failed) return W;
        }
        // Short-row level operators (square A_l, or a rank's rows of it with halo columns) whose blocks all carry a
        // union: blocks of U x 256 rows, U rows per thread in rows_union_kernel (all gathers of a
        // thread's rows in flight together).  Tried first; OMG_UNION_ROWS = 1, 2 or 4 (default 2),
        // OMG_UNION_KERNEL=0 switches the union walk off altogether.
        const int urows = union_rows();
        if (lanes == 1 && max_rows == ROWBLK_ROWS && urows > 1 && (mode & 4) && A.n_cols >= A.n_rows &&
            A.n_cols < 2 * A.n_rows && A.n_rows >= int64_t(4) * urows * ROWBLK_THREADS) {
            HostFormat<V> W = encode_with<V>(A, sets_in, urows * ROWBLK_THREADS, 1 << 24, 1, mode, true);
            if (!W.union_failed) return W;
        }
    }
    return encode_with<V>(A, sets_in, max_rows, ROWBLK_NNZ, lanes, mode);
}

template <typename V>
void DevCsrT<V>::upload(const HostCsr &A, const std::vector<int64_t> &sets_in, hipStream_t s) {
    HostFormat<V> F;
    { SetupTimer tm("  encode_csr"); F = encode_csr<V>(A, sets_in); }       // host only (setup_host.cpp)
    SetupTimer tm_up("  upload");
    n_rows = A.n_rows;
    n_cols = A.n_cols;
    nnz = A.nnz;
    sets = std::move(F.sets);
    set_blk = std::move(F.set_blk);
    set_nnz = std::move(F.set_nnz);
    set_maxlen = std::move(F.set_maxlen);
    set_ell = std::move(F.set_ell);
    set_union = std::move(F.set_union);
    if (union_rows() == 0) set_union.assign(set_union.size(), 0);
    union_max = F.union_max;
    union_blocks = F.union_blocks;
    set_pattern = std::move(F.set_pattern);
    rows_cap = F.rows_cap;
    lanes_per_row = F.lanes_per_row;
    blocks_ccoded = F.blocks_ccoded; blocks_vcoded = F.blocks_vcoded; blocks_pcoded = F.blocks_pcoded;
    nnz_ccoded = F.nnz_ccoded; nnz_vcoded = F.nnz_vcoded; nnz_pcoded = F.nnz_pcoded;
    rows_pcoded = F.rows_pcoded;
    blocks_ell = F.blocks_ell; nnz_ell = F.nnz_ell;
    indptr.alloc(A.indptr.size());
    indices.alloc(std::max<size_t>(A.indices.size(), 1));
    data.alloc(std::max<size_t>(A.data.size(), 1));
    indptr.upload(A.indptr.data(), A.indptr.size(), s);
    indices.upload(A.indices.data(), A.indices.size(), s);
    if constexpr (std::is_same<V, double>::value) data.upload(A.data.data(), A.data.size(), s);
    else data.upload(F.narrowed.data(), F.narrowed.size(), s);
    auto put = [&](auto &dev, const auto &host) {
        dev.alloc(host.size());
        dev.upload(host.data(), host.size(), s);
    };
    if (blocks_ccoded) { put(ccode, F.cc); put(cdict, F.cpool); }
    if (blocks_vcoded) { put(vcode, F.vc); put(vdict, F.vpool); }
    if (blocks_pcoded) { put(rcode, F.rc); put(pidx, F.ppool_idx); put(pval, F.ppool_val); put(pbeg, F.ppool_beg); }
    if (!F.vell.empty()) put(vell, F.vell);
    put(blk_rows, F.info);
    OMG_HIP(hipStreamSynchronize(s));   // host staging vectors may die after return
    blk_host = std::move(F.info);
}

// The (column, value) of every stored entry rebuilt from the coded form the way the kernels do
// it (csr_kernels.hip process_block / pattern_rows).  Self-test and documentation of the format.
template <typename V>
void decode_format(const HostFormat<V> &F, const HostCsr &A, std::vector<int32_t> &cols, std::vector<V> &vals) {
    cols.assign(size_t(A.nnz), -1);
    vals.assign(size_t(A.nnz), V(0));
    const int64_t nblk = int64_t(F.info.size() / BLK_INFO_INTS) - 1;
    for (int64_t k = 0; k < nblk; ++k) {
        const int32_t *rec = F.info.data() + size_t(BLK_INFO_INTS) * size_t(k);
        const int64_t r0 = rec[0], r1 = rec[BLK_INFO_INTS];
        if (rec[7]) {
            const int32_t *beg = F.ppool_beg.data() + rec[6];
            for (int64_t r = r0; r < r1; ++r) {
                const int code = F.rc[r];
                OMG_REQUIRE(code < rec[7], "decode: pattern code out of range");
                const int pb = beg[code], pe = beg[code + 1];
                OMG_REQUIRE(pe - pb == A.indptr[r + 1] - A.indptr[r], "decode: pattern length differs from the row's");
                OMG_REQUIRE(pe <= rec[5], "decode: pattern runs past its dictionary");
                if (!rec[2] && rec[3]) {
                    // union walk (rows_union_kernel): the slots of the block's union that the
                    // pattern's mask selects, in order, must be exactly the row's entries
                    const int ul = rec[3];
                    OMG_REQUIRE(ul <= UNION_MAX, "decode: union longer than UNION_MAX");
                    const int32_t *ui = F.ppool_idx.data() + rec[4] + rec[5];
                    const V *uv = F.ppool_val.data() + rec[4] + rec[5];
                    const int32_t mask = F.ppool_beg[rec[6] + rec[7] + 1 + code];
                    int j = 0;
                    for (int sl = 0; sl < ul; ++sl) {
                        if (!((mask >> sl) & 1)) continue;
                        OMG_REQUIRE(j < pe - pb && ui[sl] == F.ppool_idx[rec[4] + pb + j] &&
                                        Bits<V>::of(uv[sl]) == Bits<V>::of(F.ppool_val[rec[4] + pb + j]),
                                    "decode: union slot differs from the pattern entry");
                        ++j;
                    }
                    OMG_REQUIRE(j == pe - pb && (mask >> ul) == 0, "decode: mask does not cover the pattern");
                }
                for (int j = 0; j < pe - pb; ++j) {
                    cols[A.indptr[r] + j] = F.ppool_idx[rec[4] + pb + j] + int32_t(r);
                    if (rec[2]) {                                 // values in the ELL array, entry j of the block's rows side by side
                        OMG_REQUIRE(j < rec[3], "decode: row longer than the block's ELL width");
                        vals[A.indptr[r] + j] = F.vell[size_t(rec[2] - 1) + size_t(j) * size_t(r1 - r0) + size_t(r - r0)];
                    } else {
                        vals[A.indptr[r] + j] = F.ppool_val[rec[4] + pb + j];
                    }
                }
            }
            continue;
        }
        const int ncd = rec[2] & (2 * DICT_MAX - 1), nvd = rec[3] & (2 * DICT_MAX - 1);
        for (int64_t r = r0; r < r1; ++r)
            for (int64_t p = A.indptr[r]; p < A.indptr[r + 1]; ++p) {
                if (rec[2]) {
                    OMG_REQUIRE(F.cc[p] < ncd, "decode: column code out of range");
                    cols[p] = F.cpool[(rec[2] >> DICT_SHIFT) + F.cc[p]] + int32_t(r);
                } else {
                    cols[p] = A.indices[p];
                }
                if (rec[3]) {
                    OMG_REQUIRE(F.vc[p] < nvd, "decode: value code out of range");
                    vals[p] = F.vpool[(rec[3] >> DICT_SHIFT) + F.vc[p]];
                } else {
                    vals[p] = std::is_same<V, double>::value ? V(A.data[p]) : F.narrowed[p];
                }
            }
    }
}

void format_stats(const std::vector<int32_t> &info, const std::vector<int64_t> &set_blk, int rows_cap, int64_t w,
                  int set, int64_t *out) {
    for (int i = 0; i < OMG_FORMAT_FIELDS; ++i) out[i] = 0;
    const int64_t n_sets = set_blk.empty() ? 0 : int64_t(set_blk.size()) - 1;
    OMG_REQUIRE(set >= -1 && set < n_sets, "format_info: set out of range");
    const int64_t k0 = set < 0 ? 0 : set_blk[set], k1 = set < 0 ? (n_sets ? set_blk.back() : 0) : set_blk[set + 1];
    for (int64_t k = k0; k < k1; ++k) {
        const int32_t *rec = info.data() + size_t(BLK_INFO_INTS) * size_t(k);
        const int64_t rows = rec[BLK_INFO_INTS] - rec[0], ent = rec[BLK_INFO_INTS + 1] - rec[1];
        out[0] += rows;
        out[1] += ent;
        out[2] += 1;
        out[8] += 4 * BLK_INFO_INTS;                              // the block's table record
        if (rec[7]) {                                             // row patterns: a byte per row
            out[3] += 1;
            out[4] += rows;
            out[5] += ent;
            out[8] += rows;
            if (rec[2]) { out[8] += w * rows * rec[3]; out[10] += 1; out[11] += ent; }   // + the block's (padded) ELL values
        } else {
            if (!(rec[6] == 1 && rows_cap > ROWBLK_THREADS)) out[8] += 4 * rows;   // row pointers (not read for one-entry rows)
            if (rec[2]) { out[6] += ent; out[8] += ent; } else out[8] += 4 * ent;
            if (rec[3]) { out[7] += ent; out[8] += ent; } else out[8] += w * ent;
        }
    }
    out[9] = out[1] * (4 + w) + 4 * out[0];
}

template <typename V>
void DevCsrT<V>::format_info(int set, int64_t *out) const {
    format_stats(blk_host, set_blk, rows_cap, int64_t(sizeof(V)), set, out);
}

// Encode, decode, compare bit for bit; no device involved.
template <typename V>
void format_selftest(const omg_csr &A, int64_t *out) {
    HostCsr H = permute_csr(A, nullptr, nullptr);
    HostFormat<V> F = encode_csr<V>(H, {});
    std::vector<int32_t> cols;
    std::vector<V> vals;
    decode_format(F, H, cols, vals);
    for (int64_t p = 0; p < H.nnz; ++p) {
        const V want = V(H.data[p]);
        OMG_REQUIRE(cols[p] == H.indices[p], "format self-test: column of entry " + std::to_string(p) + " not reproduced");
        OMG_REQUIRE(Bits<V>::of(vals[p]) == Bits<V>::of(want), "format self-test: value of entry " + std::to_string(p) + " not reproduced");
    }
    format_stats(F.info, F.set_blk, F.rows_cap, int64_t(sizeof(V)), -1, out);
}

template HostFormat<double> encode_csr<double>(const HostCsr &, const std::vector<int64_t> &);
template HostFormat<float> encode_csr<float>(const HostCsr &, const std::vector<int64_t> &);
template void format_selftest<double>(const omg_csr &, int64_t *);
template void format_selftest<float>(const omg_csr &, int64_t *);
template struct DevCsrT<double>;
template struct DevCsrT<float>;

}  // namespace omg

namespace omg {
// ---- host-side checksum of a caller's array (no device involved) -----------------------------------------
// openmg_amd.mgCycle receives the A and R lists on every call (openmg/__init__.py:151) and keys its cache of device
// hierarchies on a checksum of EVERY byte of them.  One Python thread hashes ~5 GB/s (1.9 GB at 256^3: 0.3 s per
// call, and the xxhash module holds the interpreter lock, so a thread pool does not help); here the buffer is cut
// into 4 MiB chunks hashed on as many threads as the host has (xxh64's four-lane stripe loop per chunk: 32 bytes
// per iteration, bound by memory bandwidth) and the chunk digests are folded in order.
namespace {
inline uint64_t rotl64(uint64_t v, int r) { return (v << r) | (v >> (64 - r)); }
constexpr uint64_t CK_P1 = 0x9E3779B185EBCA87ull, CK_P2 = 0xC2B2AE3D27D4EB4Full, CK_P3 = 0x165667B19E3779F9ull,
                   CK_P4 = 0x85EBCA77C2B2AE63ull, CK_P5 = 0x27D4EB2F165667C5ull;
inline uint64_t ck_round(uint64_t acc, uint64_t lane) { return rotl64(acc + lane * CK_P2, 31) * CK_P1; }
inline uint64_t ck_avalanche(uint64_t h) {
    h ^= h >> 33; h *= CK_P2; h ^= h >> 29; h *= CK_P3; h ^= h >> 32;
    return h;
}
uint64_t chunk_hash(const unsigned char *p, size_t n, uint64_t seed) {
    uint64_t a0 = seed + CK_P1 + CK_P2, a1 = seed + CK_P2, a2 = seed, a3 = seed - CK_P1;
    size_t i = 0;
    for (; i + 32 <= n; i += 32) {
        uint64_t w[4];
        std::memcpy(w, p + i, 32);
        a0 = ck_round(a0, w[0]); a1 = ck_round(a1, w[1]); a2 = ck_round(a2, w[2]); a3 = ck_round(a3, w[3]);
    }
    uint64_t h = rotl64(a0, 1) + rotl64(a1, 7) + rotl64(a2, 12) + rotl64(a3, 18);
    h = (h ^ ck_round(0, a0)) * CK_P1 + CK_P4;
    h = (h ^ ck_round(0, a1)) * CK_P1 + CK_P4;
    h = (h ^ ck_round(0, a2)) * CK_P1 + CK_P4;
    h = (h ^ ck_round(0, a3)) * CK_P1 + CK_P4;
    h += uint64_t(n);
    for (; i < n; ++i) h = rotl64(h ^ (uint64_t(p[i]) * CK_P5), 11) * CK_P1;
    return ck_avalanche(h);
}
}  // namespace

void materialise_ordering(Ordering &ord) {
    if (!ord.closed_form || !ord.perm.empty()) return;
    const int64_t nx = ord.cf_nx, ny = ord.cf_ny, nz = ord.cf_nz, n = nx * ny * nz;
    ord.perm.resize(size_t(n));
    ord.inv.resize(size_t(n));
    const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const int nt = (int)std::min<int64_t>(hw, std::max<int64_t>(1, n / 65536));
    const int kind = ord.closed_form;
    auto fill = [&](int tnum) {
        const int64_t lo = n * tnum / nt, hi = n * (tnum + 1) / nt;
        for (int64_t r = lo; r < hi; ++r) {
            const int64_t i = r % nx, j = (r / nx) % ny, k = r / (nx * ny);
            int64_t slot;
            if (kind == 2) slot = (((i + j + k) & 1) ? n / 2 : 0) + r / 2;
            else slot = ((i & 1) | ((j & 1) << 1) | ((k & 1) << 2)) * (n / 8) + ((k >> 1) * (ny / 2) + (j >> 1)) * (nx / 2) + (i >> 1);
            ord.inv[size_t(r)] = int32_t(slot);
            ord.perm[size_t(slot)] = int32_t(r);
        }
    };
    std::vector<std::thread> th;
    for (int tnum = 1; tnum < nt; ++tnum) th.emplace_back(fill, tnum);
    fill(0);
    for (auto &q : th) q.join();
}

void prefault_host(void *p, size_t bytes) {
    if (bytes < (size_t(8) << 20)) return;
    volatile unsigned char *c = static_cast<volatile unsigned char *>(p);
    constexpr size_t PAGE = 4096;
    const size_t pages = (bytes + PAGE - 1) / PAGE;
    const unsigned hw = std::max(1u, std::min(32u, std::thread::hardware_concurrency()));
    const int nt = (int)std::max<size_t>(1, std::min<size_t>(hw, pages / 1024));
    auto work = [&](int t) {
        const size_t lo = pages * size_t(t) / size_t(nt), hi = pages * size_t(t + 1) / size_t(nt);
        for (size_t q = lo; q < hi; ++q) {
            const size_t off = q * PAGE;
            c[off] = 0;                                       // (the buffer is an OUTPUT: its contents are about to be overwritten)
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
    work(0);
    for (auto &q : th) q.join();
}

void download_staged(void *host, const void *dev, size_t bytes, hipStream_t s) {
    constexpr size_t CH = size_t(32) << 20;
    if (bytes < (size_t(8) << 20)) {
        if (bytes) OMG_HIP(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
        OMG_HIP(hipStreamSynchronize(s));
        return;
    }
    // two pinned buffers and their events per DEVICE, made on first use and shared (one copy at a time per device: events
    // belong to the device that was current when they were made, and a process may drive several — PlaneDistGroup,
    // replicated hierarchies)
    struct Staging { std::mutex mu; void *pin[2] = {nullptr, nullptr}; hipEvent_t ev[2] = {nullptr, nullptr}; };
    static std::mutex table_mu;
    static std::map<int, std::unique_ptr<Staging>> table;
    int device = 0;
    OMG_HIP(hipGetDevice(&device));
    Staging *st;
    {
        std::lock_guard<std::mutex> lock(table_mu);
        auto &slot = table[device];
        if (!slot) slot.reset(new Staging);
        st = slot.get();
    }
    std::lock_guard<std::mutex> lock(st->mu);
    void **pin = st->pin;
    hipEvent_t *ev = st->ev;
    if (!pin[0]) {
        for (int i = 0; i < 2; ++i) {
            OMG_HIP(hipHostMalloc(&pin[i], CH, hipHostMallocPortable));
            OMG_HIP(hipEventCreateWithFlags(&ev[i], hipEventDisableTiming));
        }
    }
    const size_t n_chunks = (bytes + CH - 1) / CH;
    const unsigned hw = std::max(1u, std::min(16u, std::thread::hardware_concurrency()));
    auto issue = [&](size_t c) {
        const size_t off = c * CH, len = std::min(CH, bytes - off);
        OMG_HIP(hipMemcpyAsync(pin[c & 1], static_cast<const char *>(dev) + off, len, hipMemcpyDeviceToHost, s));
        OMG_HIP(hipEventRecord(ev[c & 1], s));
    };
    issue(0);
    for (size_t c = 0; c < n_chunks; ++c) {
        if (c + 1 < n_chunks) issue(c + 1);                    // (buffer (c + 1) & 1 was emptied in the previous iteration)
        OMG_HIP(hipEventSynchronize(ev[c & 1]));
        const size_t off = c * CH, len = std::min(CH, bytes - off);
        const int nt = (int)std::max<size_t>(1, std::min<size_t>(hw, len >> 20));
        auto work = [&](int t) {
            const size_t lo = len * size_t(t) / size_t(nt), hi = len * size_t(t + 1) / size_t(nt);
            std::memcpy(static_cast<char *>(host) + off + lo, static_cast<const char *>(pin[c & 1]) + lo, hi - lo);
        };
        std::vector<std::thread> th;
        for (int t = 1; t < nt; ++t) th.emplace_back(work, t);
        work(0);
        for (auto &q : th) q.join();
    }
}

uint64_t host_checksum(const void *buf, int64_t bytes) {
    const unsigned char *p = static_cast<const unsigned char *>(buf);
    constexpr int64_t CH = int64_t(4) << 20;
    const int64_t n_chunks = std::max<int64_t>(1, (bytes + CH - 1) / CH);
    std::vector<uint64_t> dig(size_t(n_chunks), 0);
    const unsigned hw = std::max(1u, std::min(64u, std::thread::hardware_concurrency()));
    const int nt = (int)std::max<int64_t>(1, std::min<int64_t>(hw, n_chunks / 4));
    std::atomic<int64_t> next(0);
    auto work = [&] {
        for (;;) {
            const int64_t c = next.fetch_add(1);
            if (c >= n_chunks) return;
            const int64_t lo = c * CH, hi = std::min(bytes, lo + CH);
            dig[size_t(c)] = chunk_hash(p + lo, size_t(std::max<int64_t>(0, hi - lo)), uint64_t(c));
        }
    };
    std::vector<std::thread> th;
    for (int t = 1; t < nt; ++t) th.emplace_back(work);
    work();
    for (auto &q : th) q.join();
    return chunk_hash(reinterpret_cast<const unsigned char *>(dig.data()), dig.size() * 8, uint64_t(bytes));
}

}  // namespace omg

extern "C" int omg_host_checksum(const void *buf, int64_t bytes, uint64_t *out) {
    if (!out || bytes < 0 || (!buf && bytes > 0)) {
        omg::set_last_error("omg_host_checksum: null / negative argument");
        return OMG_ERR_INVALID;
    }
    *out = omg::host_checksum(buf, bytes);
    return OMG_OK;
}

