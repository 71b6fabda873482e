// Microbenchmark behind DESIGN.md §6 "the tail": does ONE persistent kernel with grid barriers
// beat a chain of small dependent launches?  Both variants run the same phase body N times: a
// 3-deep dependent load chain per thread (index -> table entry -> gathered value -> store), i.e.
// the block record -> codes / dictionary -> gathers -> store chain of a coarse-level row kernel,
// on vectors that live in L2 / Infinity Cache.
//   (a) N launches of phase_kernel on one stream;
//   (b) one launch of persistent_kernel: N phases separated by a grid barrier (monotonic counter,
//       lane-0 release fence before the arrive, relaxed agent-scope poll with s_sleep, acquire fence
//       after: the "barrier-counter" form of MI355X_MICROARCH.md; spins are bounded).
// Build: hipcc -O3 --offload-arch=gfx950 tools/tail_probe.hip -o tools/build/tail_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <vector>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

__device__ int g_depth = 3;     // dependent loads in front of the gathers (3: record -> dictionary -> gathers; 2: no dictionary hop)

__device__ __forceinline__ void phase_body(const int *__restrict__ idx, const int *__restrict__ table,
                                           const double *__restrict__ src, double *__restrict__ dst, int n, int phase) {
    const int depth = g_depth;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        const int a = idx[i];                       // "block record / code"
        const int off = depth >= 3 ? table[a & 1023] : (a & 3) - 1;   // "dictionary"
        int j = i + off + phase;
        j = j < 0 ? 0 : (j >= n ? n - 1 : j);
        const double v = src[j] + src[(j + 1) % n];  // "gathers"
        dst[i] = 0.5 * v + 1e-3;                     // store
    }
}

__global__ __launch_bounds__(256) void phase_kernel(const int *idx, const int *table, const double *src, double *dst, int n, int phase) {
    phase_body(idx, table, src, dst, n, phase);
}

__device__ __forceinline__ bool grid_barrier(unsigned *counter, unsigned target, unsigned *timeout) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // every storing wave
    __syncthreads();
    bool ok = true;
    if (threadIdx.x == 0) {
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __hip_atomic_fetch_add(counter, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        unsigned spins = 0;
        while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
            __builtin_amdgcn_s_sleep(2);
            if (++spins > (1u << 22)) { *timeout = 1; ok = false; break; }     // bounded: never hang the box
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    return ok;
}

__global__ __launch_bounds__(256) void persistent_kernel(const int *idx, const int *table, double *a, double *b, int n,
                                                          int phases, unsigned *counter, unsigned *timeout) {
    for (int p = 0; p < phases; ++p) {
        phase_body(idx, table, (p & 1) ? b : a, (p & 1) ? a : b, n, p);
        if (p + 1 < phases && !grid_barrier(counter, unsigned(p + 1) * gridDim.x, timeout)) return;
    }
}

int main(int argc, char **argv) {
    const int phases = argc > 1 ? atoi(argv[1]) : 21, reps = 50;
    const int depth = argc > 2 ? atoi(argv[2]) : 3;
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_depth), &depth, sizeof(int)));
    printf("dependent loads before the store: %d\n", depth);
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    printf("%8s %8s %22s %22s\n", "rows", "wgs", "chain of launches us/phase", "persistent us/phase");
    for (int wgs : {32, 256, 2048}) {
        const int n = wgs * 1024;                  // 4 rows per thread, like a 512-row block per 2 x 256
        std::vector<int> h_idx(n), h_tab(1024);
        for (int i = 0; i < n; ++i) h_idx[i] = (i * 7919) & 0xffff;
        for (int i = 0; i < 1024; ++i) h_tab[i] = (i % 7) - 3;
        int *idx, *tab;
        double *a, *b;
        unsigned *ctl;
        CK(hipMalloc(&idx, n * sizeof(int)));
        CK(hipMalloc(&tab, 1024 * sizeof(int)));
        CK(hipMalloc(&a, n * sizeof(double)));
        CK(hipMalloc(&b, n * sizeof(double)));
        CK(hipMalloc(&ctl, 64));
        CK(hipMemcpy(idx, h_idx.data(), n * sizeof(int), hipMemcpyHostToDevice));
        CK(hipMemcpy(tab, h_tab.data(), 1024 * sizeof(int), hipMemcpyHostToDevice));
        CK(hipMemset(a, 0, n * sizeof(double)));
        CK(hipMemset(b, 0, n * sizeof(double)));
        float ms_chain = 0.f, ms_pers = 0.f;
        for (int variant = 0; variant < 2; ++variant) {
            for (int rep = -5; rep < reps; ++rep) {
                if (rep == 0) CK(hipEventRecord(e0, s));
                if (variant == 0) {
                    for (int p = 0; p < phases; ++p)
                        hipLaunchKernelGGL(phase_kernel, dim3(wgs), dim3(256), 0, s, idx, tab, (p & 1) ? b : a, (p & 1) ? a : b, n, p);
                } else {
                    if (wgs > 1024) continue;       // must be co-resident: 256 CUs x 4 blocks
                    CK(hipMemsetAsync(ctl, 0, 64, s));
                    hipLaunchKernelGGL(persistent_kernel, dim3(wgs), dim3(256), 0, s, idx, tab, a, b, n, phases, ctl, ctl + 8);
                }
            }
            CK(hipEventRecord(e1, s));
            CK(hipEventSynchronize(e1));
            float ms = 0.f;
            CK(hipEventElapsedTime(&ms, e0, e1));
            (variant == 0 ? ms_chain : ms_pers) = ms;
        }
        unsigned tmo[16];
        CK(hipMemcpy(tmo, ctl, 64, hipMemcpyDeviceToHost));
        if (wgs > 1024) printf("%8d %8d %22.2f %22s\n", n, wgs, 1e3 * ms_chain / reps / phases, "(not co-resident)");
        else printf("%8d %8d %22.2f %22.2f%s\n", n, wgs, 1e3 * ms_chain / reps / phases, 1e3 * ms_pers / reps / phases, tmo[8] ? "  TIMEOUT" : "");
        hipFree(idx); hipFree(tab); hipFree(a); hipFree(b); hipFree(ctl);
    }
    return 0;
}
