// TEST INFRASTRUCTURE — not part of the product, never loaded unless OMG_RCCL_LIB names it.
//
// A stand-in for librccl that lets N PROCESSES SHARING ONE GPU run the product's N > 1 RCCL call sites
// (csrc/dist.hip, csrc/dist27.hip): RCCL itself refuses two ranks on one device, and the pool has one-GPU boxes.
// It implements the eleven symbols csrc/rccl_dyn.h resolves, with RCCL's semantics where they can deadlock a
// first multi-GPU run:
//   * everything is STREAM-ORDERED device work (no host synchronisation in Send / Recv / AllGather / AllReduce);
//   * ncclSend BLOCKS THE STREAM until the matching ncclRecv has taken the data (no eager completion), ops of one
//     ncclGroupStart / ncclGroupEnd section progress together, ops between a pair of ranks match in posting order;
//   * ncclCommInitRank is a collective over all ranks of the id.
// Mechanics: every rank of a communicator owns a "mailbox" in device memory (hipMalloc, exported with hipIpc and
// mapped by every other rank at ncclCommInitRank; the rendezvous is a POSIX shared-memory block named after the
// unique id).  A mailbox holds, for every source rank, a ring of NSLOT data slots and two 64-bit counters: `full`
// (chunks the source has put down) and `ack` (chunks this rank has taken).  One chunk = one launch of xfer_kernel
// on the caller's stream: wait (bounded by wall-clock ticks) until a counter reaches a value, copy, release-store
// another counter.  A wait that gives up — a deadlock in the caller's schedule — sets a status word instead of
// hanging the box: frccl_status() reports it, and every later call returns ncclSystemError.
// AllReduce adds the ranks' contributions in rank order on every rank (all ranks get the same bits, as a ring
// all-reduce gives them).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <thread>
#include <vector>

namespace {

constexpr int MAX_RANKS = 16;
constexpr int NSLOT = 4;
constexpr uint64_t MAGIC = 0x4652434c31ull;   // "FRCL1"

size_t slot_bytes() {
    static size_t v = [] {
        const char *e = getenv("FRCCL_SLOT_KB");
        return size_t(e ? atol(e) : 4096) * 1024;
    }();
    return v;
}
double timeout_s() {
    static double v = [] {
        const char *e = getenv("FRCCL_TIMEOUT_S");
        return e ? atof(e) : 30.0;
    }();
    return v;
}

struct Ctl {                                   // the rendezvous block in /dev/shm
    std::atomic<int> ready[MAX_RANKS];
    std::atomic<int> opened;
    std::atomic<int> closed;
    hipIpcMemHandle_t handle[MAX_RANKS];
    int device[MAX_RANKS];
};

// mailbox layout: [MAX_RANKS x {full, ack} counters, 64 B apart][n_ranks x NSLOT x slot_bytes of data]
constexpr size_t CTR_STRIDE = 8;               // in uint64: 64 bytes between counters
constexpr size_t HEADER_BYTES = 2 * MAX_RANKS * CTR_STRIDE * sizeof(uint64_t);

struct Mailbox {
    char *base = nullptr;
    uint64_t *full(int src) const { return reinterpret_cast<uint64_t *>(base) + size_t(2 * src) * CTR_STRIDE; }
    uint64_t *ack(int src) const { return reinterpret_cast<uint64_t *>(base) + size_t(2 * src + 1) * CTR_STRIDE; }
    char *slot(int src, uint64_t seq, int n_ranks) const {
        (void)n_ranks;
        return base + HEADER_BYTES + (size_t(src) * NSLOT + size_t(seq % NSLOT)) * slot_bytes();
    }
};

uint32_t *g_status = nullptr;                  // host-pinned: bit 0 = a bounded wait gave up
std::atomic<int> g_sticky{0};

struct Op {
    bool send;
    char *ptr;
    size_t bytes;
    int peer;
    hipStream_t stream;
    ncclComm *comm;
};

}  // namespace

struct ncclComm {
    uint64_t magic = MAGIC;
    int rank = 0, n_ranks = 1;
    Mailbox box[MAX_RANKS];                    // box[rank] is mine; the others are hipIpc mappings
    Ctl *ctl = nullptr;
    char shm_name[64] = {0};
    uint64_t sent[MAX_RANKS] = {0};            // chunks put into box[peer] so far
    uint64_t taken[MAX_RANKS] = {0};           // chunks taken out of box[rank] from each source so far
    char *scratch = nullptr;                   // AllReduce: every rank's contribution
    size_t scratch_bytes = 0;
};

namespace {
thread_local int t_depth = 0;
thread_local std::vector<Op> t_pending;

__global__ void __launch_bounds__(1024) xfer_kernel(const uint64_t *wait_ctr, uint64_t wait_value, const char *src, char *dst, size_t bytes,
                                                    uint64_t *set_ctr, uint64_t set_value, uint32_t *status, long long ticks) {
    if (wait_ctr) {
        if (threadIdx.x == 0) {
            const long long t0 = wall_clock64();
            // (once a wait has given up the schedule is broken: the launches still queued behind it do not wait their turn out)
            const bool broken = __hip_atomic_load(status, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != 0;
            for (; !broken;) {
                const uint64_t v = __hip_atomic_load(wait_ctr, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                if (v >= wait_value) break;
                if (wall_clock64() - t0 > ticks) {
                    __hip_atomic_fetch_or(status, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                    break;
                }
                __builtin_amdgcn_s_sleep(32);
            }
        }
        __syncthreads();
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "");
    }
    if (bytes) {
        if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | bytes) & 15) == 0) {
            const uint4 *s = reinterpret_cast<const uint4 *>(src);
            uint4 *d = reinterpret_cast<uint4 *>(dst);
            for (size_t i = threadIdx.x; i < bytes / 16; i += blockDim.x) d[i] = s[i];
        } else if (((reinterpret_cast<uintptr_t>(src) | reinterpret_cast<uintptr_t>(dst) | bytes) & 3) == 0) {
            const uint32_t *s = reinterpret_cast<const uint32_t *>(src);
            uint32_t *d = reinterpret_cast<uint32_t *>(dst);
            for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = s[i];
        } else {
            for (size_t i = threadIdx.x; i < bytes; i += blockDim.x) dst[i] = src[i];
        }
    }
    if (set_ctr) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "");
        __syncthreads();
        if (threadIdx.x == 0) __hip_atomic_store(set_ctr, set_value, __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}

template <typename T>
__global__ void sum_ranks_kernel(const T *parts, T *out, size_t count, int n_ranks) {
    const size_t i = size_t(blockIdx.x) * blockDim.x + threadIdx.x;
    if (i >= count) return;
    T s = parts[i];
    for (int r = 1; r < n_ranks; ++r) s += parts[size_t(r) * count + i];
    out[i] = s;
}

long long ticks() { return (long long)(timeout_s() * 100e6); }   // wall_clock64 counts at 100 MHz on gfx950

bool ok(hipError_t e, const char *what) {
    if (e == hipSuccess) return true;
    fprintf(stderr, "fake_rccl: %s: %s\n", what, hipGetErrorString(e));
    return false;
}

size_t type_bytes(ncclDataType_t t) {
    switch (t) {
        case ncclInt8: case ncclUint8: return 1;
        case ncclFloat16: case ncclBfloat16: return 2;
        case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
        case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
        default: return 0;
    }
}

bool status_bad() {
    if (g_status && *reinterpret_cast<volatile uint32_t *>(g_status)) g_sticky.store(1);
    return g_sticky.load() != 0;
}

void launch(hipStream_t st, const uint64_t *wait_ctr, uint64_t wait_value, const char *src, char *dst, size_t bytes, uint64_t *set_ctr,
            uint64_t set_value) {
    hipLaunchKernelGGL(xfer_kernel, dim3(1), dim3(1024), 0, st, wait_ctr, wait_value, src, dst, bytes, set_ctr, set_value, g_status, ticks());
}

struct Piece {                                 // one chunk of one posted operation
    ncclComm *c;
    bool send;
    int peer;
    char *ptr;
    size_t bytes;
    hipStream_t st;
    bool done = false;
};

// Run a section's operations: sends and receives in batches of at most NSLOT chunks per direction and pair (the ring's
// capacity), sends of a batch first; afterwards every send waits for its last chunk's acknowledgement.
ncclResult_t run_section(std::vector<Op> &ops) {
    if (status_bad()) return ncclSystemError;
    std::vector<Piece> pieces;
    // a rank's operations with itself: the k-th send meets the k-th receive, one copy
    std::vector<size_t> self_send, self_recv;
    for (size_t i = 0; i < ops.size(); ++i)
        if (ops[i].peer == ops[i].comm->rank) (ops[i].send ? self_send : self_recv).push_back(i);
    if (self_send.size() != self_recv.size()) {
        fprintf(stderr, "fake_rccl: %zu sends to self against %zu receives in one section\n", self_send.size(), self_recv.size());
        return ncclInvalidUsage;
    }
    for (size_t k = 0; k < self_send.size(); ++k) {
        const Op &s = ops[self_send[k]], &r = ops[self_recv[k]];
        if (s.bytes != r.bytes) return ncclInvalidUsage;
        launch(r.stream, nullptr, 0, s.ptr, r.ptr, s.bytes, nullptr, 0);
    }
    for (const Op &o : ops) {
        if (o.peer == o.comm->rank) continue;
        const size_t sb = slot_bytes();
        size_t off = 0;
        do {                                   // (a zero-byte operation is still one chunk: it orders)
            const size_t n = o.bytes - off < sb ? o.bytes - off : sb;
            pieces.push_back({o.comm, o.send, o.peer, o.ptr + off, n, o.stream});
            off += n;
        } while (off < o.bytes);
    }
    struct Last { ncclComm *c; int peer; hipStream_t st; uint64_t seq; };
    std::vector<Last> lasts;
    size_t left = pieces.size();
    while (left) {
        for (int pass = 0; pass < 2; ++pass) {          // 0: sends, 1: receives
            std::vector<std::pair<ncclComm *, int>> seen;   // (communicator, peer) -> pieces issued in this batch
            std::vector<int> count;
            for (Piece &p : pieces) {
                if (p.done || p.send != (pass == 0)) continue;
                size_t k = 0;
                for (; k < seen.size(); ++k)
                    if (seen[k].first == p.c && seen[k].second == p.peer) break;
                if (k == seen.size()) { seen.push_back({p.c, p.peer}); count.push_back(0); }
                if (count[k] >= NSLOT) continue;
                ++count[k];
                ncclComm &c = *p.c;
                if (p.send) {
                    const uint64_t seq = ++c.sent[p.peer];
                    const Mailbox &box = c.box[p.peer];
                    // slot seq % NSLOT is free once chunk seq - NSLOT has been taken
                    launch(p.st, seq > NSLOT ? box.ack(c.rank) : nullptr, seq > NSLOT ? seq - NSLOT : 0, p.ptr, box.slot(c.rank, seq, c.n_ranks),
                           p.bytes, box.full(c.rank), seq);
                    bool found = false;
                    for (Last &l : lasts)
                        if (l.c == &c && l.peer == p.peer && l.st == p.st) { l.seq = seq; found = true; }
                    if (!found) lasts.push_back({&c, p.peer, p.st, seq});
                } else {
                    const uint64_t seq = ++c.taken[p.peer];
                    const Mailbox &box = c.box[c.rank];
                    launch(p.st, box.full(p.peer), seq, box.slot(p.peer, seq, c.n_ranks), p.ptr, p.bytes, box.ack(p.peer), seq);
                }
                p.done = true;
                --left;
            }
        }
    }
    for (const Last &l : lasts)                          // ncclSend returns the stream when the receiver has the data
        launch(l.st, l.c->box[l.peer].ack(l.c->rank), l.seq, nullptr, nullptr, 0, nullptr, 0);
    return ok(hipGetLastError(), "kernel launch") ? ncclSuccess : ncclUnhandledCudaError;
}

ncclResult_t post(Op op) {
    if (!op.comm || op.comm->magic != MAGIC) return ncclInvalidArgument;
    if (op.peer < 0 || op.peer >= op.comm->n_ranks) return ncclInvalidArgument;
    t_pending.push_back(op);
    if (t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_pending);
    return run_section(ops);
}

}  // namespace

extern "C" {

__attribute__((visibility("default"))) int frccl_status() { return status_bad() ? 1 : 0; }
__attribute__((visibility("default"))) const char *frccl_identity() { return "fake_rccl test shim (tests/fake_rccl)"; }

ncclResult_t ncclGetUniqueId(ncclUniqueId *id) {
    if (!id) return ncclInvalidArgument;
    memset(id, 0, sizeof(*id));
    unsigned long long r[2] = {0, 0};
    FILE *f = fopen("/dev/urandom", "rb");
    if (f) { if (fread(r, sizeof(r), 1, f) != 1) r[0] = 0; fclose(f); }
    if (!r[0]) r[0] = (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count() ^ ((unsigned long long)getpid() << 32);
    snprintf(id->internal, sizeof(id->internal), "frccl_%016llx%08llx", r[0], r[1] & 0xffffffffull);
    return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t *out, int n_ranks, ncclUniqueId id, int rank) {
    if (!out || n_ranks < 1 || n_ranks > MAX_RANKS || rank < 0 || rank >= n_ranks) return ncclInvalidArgument;
    if (strncmp(id.internal, "frccl_", 6) != 0) return ncclInvalidArgument;
    if (!g_status) {
        if (!ok(hipHostMalloc(reinterpret_cast<void **>(&g_status), 64, hipHostMallocMapped), "hipHostMalloc")) return ncclUnhandledCudaError;
        memset(g_status, 0, 64);
    }
    ncclComm *c = new ncclComm();
    c->rank = rank;
    c->n_ranks = n_ranks;
    const size_t bytes = HEADER_BYTES + size_t(n_ranks) * NSLOT * slot_bytes();
    if (!ok(hipMalloc(reinterpret_cast<void **>(&c->box[rank].base), bytes), "hipMalloc(mailbox)")) { delete c; return ncclUnhandledCudaError; }
    if (!ok(hipMemset(c->box[rank].base, 0, HEADER_BYTES), "hipMemset") || !ok(hipDeviceSynchronize(), "sync")) { delete c; return ncclUnhandledCudaError; }
    if (n_ranks > 1) {
        snprintf(c->shm_name, sizeof(c->shm_name), "/%.40s", id.internal);
        const int fd = shm_open(c->shm_name, O_CREAT | O_RDWR, 0600);
        if (fd < 0 || ftruncate(fd, sizeof(Ctl)) != 0) { perror("fake_rccl: shm_open"); delete c; return ncclSystemError; }
        c->ctl = static_cast<Ctl *>(mmap(nullptr, sizeof(Ctl), PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0));
        close(fd);
        if (c->ctl == MAP_FAILED) { delete c; return ncclSystemError; }
        int dev = 0;
        (void)hipGetDevice(&dev);
        if (!ok(hipIpcGetMemHandle(&c->ctl->handle[rank], c->box[rank].base), "hipIpcGetMemHandle")) { delete c; return ncclUnhandledCudaError; }
        c->ctl->device[rank] = dev;
        c->ctl->ready[rank].store(1, std::memory_order_release);
        const auto t0 = std::chrono::steady_clock::now();
        auto late = [&] { return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 120.0; };
        for (int r = 0; r < n_ranks; ++r) {
            while (!c->ctl->ready[r].load(std::memory_order_acquire)) {
                if (late()) { fprintf(stderr, "fake_rccl: rank %d never arrived at ncclCommInitRank\n", r); return ncclSystemError; }
                std::this_thread::sleep_for(std::chrono::milliseconds(1));
            }
            if (r == rank) continue;
            if (c->ctl->device[r] != dev) {
                fprintf(stderr, "fake_rccl: rank %d is on device %d, rank %d on %d — this shim is for ranks SHARING one GPU\n", r, c->ctl->device[r], rank, dev);
                return ncclInvalidUsage;
            }
            void *p = nullptr;
            if (!ok(hipIpcOpenMemHandle(&p, c->ctl->handle[r], hipIpcMemLazyEnablePeerAccess), "hipIpcOpenMemHandle")) return ncclUnhandledCudaError;
            c->box[r].base = static_cast<char *>(p);
        }
        c->ctl->opened.fetch_add(1);
        while (c->ctl->opened.load() < n_ranks) {
            if (late()) return ncclSystemError;
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        }
        if (rank == 0) shm_unlink(c->shm_name);     // everybody has it mapped; the name can go
    }
    *out = c;
    return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
    ncclComm *c = comm;
    if (!c || c->magic != MAGIC) return ncclInvalidArgument;
    (void)hipDeviceSynchronize();
    if (status_bad()) fprintf(stderr, "fake_rccl: rank %d: a bounded wait gave up earlier (a send without its receive, or a rank that left)\n", c->rank);
    if (c->ctl) {
        for (int r = 0; r < c->n_ranks; ++r)
            if (r != c->rank && c->box[r].base) (void)hipIpcCloseMemHandle(c->box[r].base);
        c->ctl->closed.fetch_add(1);
        const auto t0 = std::chrono::steady_clock::now();   // nobody frees a mailbox a neighbour still has mapped
        while (c->ctl->closed.load() < c->n_ranks && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() < 20.0)
            std::this_thread::sleep_for(std::chrono::milliseconds(1));
        munmap(c->ctl, sizeof(Ctl));
    }
    (void)hipFree(c->box[c->rank].base);
    if (c->scratch) (void)hipFree(c->scratch);
    c->magic = 0;
    delete c;
    return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t comm, int *count) {
    if (!comm || comm->magic != MAGIC || !count) return ncclInvalidArgument;
    *count = comm->n_ranks;
    return ncclSuccess;
}

const char *ncclGetErrorString(ncclResult_t r) {
    switch (r) {
        case ncclSuccess: return "no error";
        case ncclUnhandledCudaError: return "unhandled HIP error (fake_rccl)";
        case ncclSystemError: return "system error (fake_rccl: a bounded wait gave up, or the rendezvous failed)";
        case ncclInvalidArgument: return "invalid argument (fake_rccl)";
        case ncclInvalidUsage: return "invalid usage (fake_rccl)";
        default: return "error (fake_rccl)";
    }
}

ncclResult_t ncclGroupStart() {
    ++t_depth;
    return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
    if (t_depth <= 0) return ncclInvalidUsage;
    if (--t_depth > 0) return ncclSuccess;
    std::vector<Op> ops;
    ops.swap(t_pending);
    return ops.empty() ? (status_bad() ? ncclSystemError : ncclSuccess) : run_section(ops);
}

ncclResult_t ncclSend(const void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!w) return ncclInvalidArgument;
    return post({true, const_cast<char *>(static_cast<const char *>(buf)), count * w, peer, stream, comm});
}

ncclResult_t ncclRecv(void *buf, size_t count, ncclDataType_t type, int peer, ncclComm_t comm, hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!w) return ncclInvalidArgument;
    return post({false, static_cast<char *>(buf), count * w, peer, stream, comm});
}

ncclResult_t ncclAllGather(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t type, ncclComm_t comm, hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!w || !comm || comm->magic != MAGIC) return ncclInvalidArgument;
    if (t_depth > 0) return ncclInvalidUsage;          // (the product never groups collectives)
    const size_t bytes = count * w;
    char *mine = static_cast<char *>(recvbuf) + size_t(comm->rank) * bytes;
    if (mine != sendbuf) launch(stream, nullptr, 0, static_cast<const char *>(sendbuf), mine, bytes, nullptr, 0);
    std::vector<Op> ops;
    for (int d = 1; d < comm->n_ranks; ++d) {
        const int to = (comm->rank + d) % comm->n_ranks, from = (comm->rank - d + comm->n_ranks) % comm->n_ranks;
        ops.push_back({true, const_cast<char *>(static_cast<const char *>(sendbuf)), bytes, to, stream, comm});
        ops.push_back({false, static_cast<char *>(recvbuf) + size_t(from) * bytes, bytes, from, stream, comm});
    }
    return ops.empty() ? ncclSuccess : run_section(ops);
}

ncclResult_t ncclAllReduce(const void *sendbuf, void *recvbuf, size_t count, ncclDataType_t type, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
    const size_t w = type_bytes(type);
    if (!comm || comm->magic != MAGIC) return ncclInvalidArgument;
    if (op != ncclSum || (type != ncclFloat64 && type != ncclFloat32)) {
        fprintf(stderr, "fake_rccl: ncclAllReduce only adds floats and doubles\n");
        return ncclInvalidArgument;
    }
    if (t_depth > 0) return ncclInvalidUsage;
    ncclComm *c = comm;
    const size_t bytes = count * w;
    if (c->scratch_bytes < bytes * size_t(c->n_ranks)) {
        // (a synchronising call, but only the first time a size is seen; every rank sees the sizes in the same order)
        if (c->scratch) { (void)hipDeviceSynchronize(); (void)hipFree(c->scratch); }
        c->scratch_bytes = bytes * size_t(c->n_ranks) * 2;
        if (!ok(hipMalloc(reinterpret_cast<void **>(&c->scratch), c->scratch_bytes), "hipMalloc(scratch)")) return ncclUnhandledCudaError;
    }
    const ncclResult_t r = ncclAllGather(sendbuf, c->scratch, count, type, comm, stream);
    if (r != ncclSuccess) return r;
    const unsigned blocks = unsigned((count + 255) / 256);
    if (type == ncclFloat64)
        hipLaunchKernelGGL(sum_ranks_kernel<double>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const double *>(c->scratch),
                           static_cast<double *>(recvbuf), count, c->n_ranks);
    else
        hipLaunchKernelGGL(sum_ranks_kernel<float>, dim3(blocks), dim3(256), 0, stream, reinterpret_cast<const float *>(c->scratch),
                           static_cast<float *>(recvbuf), count, c->n_ranks);
    return ok(hipGetLastError(), "kernel launch") ? ncclSuccess : ncclUnhandledCudaError;
}

}  // extern "C"
