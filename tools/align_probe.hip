// Which wide per-lane loads does gfx950 serve at sub-natural alignment?  (rows_union_kernel pairs rows
// only where the answer is yes.)  Build: hipcc -O3 --offload-arch=gfx950 tools/align_probe.hip -o build/align_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef float f2 __attribute__((ext_vector_type(2)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef f2 f2a __attribute__((aligned(4)));
typedef f4 f4a __attribute__((aligned(4)));

// out[t] = number of wrong dwords seen by thread t; every thread loads at byte offset 4 * (k * t + shift)
template <int KIND>
__global__ void probe(const unsigned *src, unsigned n_bytes, int shift, int *bad) {
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    const int first = 4 * t + shift;                      // index of the first dword
    const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned *>(src), 0, n_bytes, 0x00020000);
    unsigned got[4] = {0, 0, 0, 0};
    int n = 0;
    if (KIND == 0) { const v2u q = __builtin_amdgcn_raw_buffer_load_b64(rs, first * 4, 0, 0); got[0] = q.x; got[1] = q.y; n = 2; }
    if (KIND == 1) { const v4u q = __builtin_amdgcn_raw_buffer_load_b128(rs, first * 4, 0, 0); got[0] = q.x; got[1] = q.y; got[2] = q.z; got[3] = q.w; n = 4; }
    if (KIND == 2) { const f2a q = *reinterpret_cast<const f2a *>(src + first); got[0] = __float_as_uint(q.x); got[1] = __float_as_uint(q.y); n = 2; }
    if (KIND == 3) { const f4a q = *reinterpret_cast<const f4a *>(src + first); got[0] = __float_as_uint(q.x); got[1] = __float_as_uint(q.y); got[2] = __float_as_uint(q.z); got[3] = __float_as_uint(q.w); n = 4; }
    int wrong = 0;
    for (int e = 0; e < n; ++e) wrong += got[e] != unsigned(first + e) * 2654435761u;
    bad[t] = wrong;
}

int main() {
    const int threads = 1 << 16, n = 4 * threads + 64;
    std::vector<unsigned> h(n);
    for (int i = 0; i < n; ++i) h[i] = unsigned(i) * 2654435761u;
    unsigned *d;
    int *bad;
    hipMalloc(&d, n * 4);
    hipMalloc(&bad, threads * 4);
    hipMemcpy(d, h.data(), n * 4, hipMemcpyHostToDevice);
    const char *names[4] = {"raw_buffer_load_b64", "raw_buffer_load_b128", "global 8-byte vector load", "global 16-byte vector load"};
    for (int kind = 0; kind < 4; ++kind)
        for (int shift = 0; shift < 4; ++shift) {
            if (kind == 0) probe<0><<<threads / 256, 256>>>(d, n * 4, shift, bad);
            if (kind == 1) probe<1><<<threads / 256, 256>>>(d, n * 4, shift, bad);
            if (kind == 2) probe<2><<<threads / 256, 256>>>(d, n * 4, shift, bad);
            if (kind == 3) probe<3><<<threads / 256, 256>>>(d, n * 4, shift, bad);
            std::vector<int> r(threads);
            hipMemcpy(r.data(), bad, threads * 4, hipMemcpyDeviceToHost);
            long w = 0;
            for (int v : r) w += v;
            printf("%-28s byte offset = 16 t + %2d: %s (%ld wrong dwords)\n", names[kind], 4 * shift, w ? "WRONG" : "ok", w);
        }
    return 0;
}
