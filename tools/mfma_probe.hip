#include <hip/hip_runtime.h>
#include <cstdio>
typedef double v4d __attribute__((ext_vector_type(4)));
__global__ void k(int i0, int k0, int j0, double* out) {
    const int lane = threadIdx.x;
    // hypothesis: A lane = 16*k + i ; B lane = 16*k + j
    double a = (lane % 16 == i0 && lane / 16 == k0) ? 1.0 : 0.0;
    double b = (lane % 16 == j0 && lane / 16 == k0) ? 1.0 : 0.0;
    v4d c = {0, 0, 0, 0};
    c = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c, 0, 0, 0);
    for (int r = 0; r < 4; ++r) out[lane * 4 + r] = c[r];
}
int main() {
    double* d; (void)hipMalloc(&d, 256 * 8);
    int tests[][3] = {{0,0,0},{1,0,0},{0,0,1},{5,2,9},{12,3,7},{15,1,15},{4,0,2},{8,0,3}};
    for (auto& t : tests) {
        hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, t[0], t[1], t[2], d);
        double h[256]; (void)hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost);
        for (int q = 0; q < 256; ++q) if (h[q] != 0.0) printf("i0=%d k0=%d j0=%d -> lane %d reg %d val %g\n", t[0], t[1], t[2], q / 4, q % 4, h[q]);
    }
    return 0;
}
