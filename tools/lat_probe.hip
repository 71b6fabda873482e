// Dependent-chain latencies on one wave of an otherwise idle CU (gfx950): what bounds one step of
// march.hip's wavefront sweep.  Build: hipcc -O3 --offload-arch=gfx950 tools/lat_probe.hip -o build/lat_probe
#include <hip/hip_runtime.h>
#include <cstdio>

#define N 4096

__global__ void fma64_chain(double *out, double a, double b) {
    double x = out[threadIdx.x];
    long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fma(x, a, b);
    long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = double(t1 - t0) / N;
}
__global__ void fma32_chain(float *out, float a, float b) {
    float x = out[threadIdx.x];
    long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fmaf(x, a, b);
    long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = float(t1 - t0) / N;
}
__global__ void bperm_chain(double *out) {
    int x = threadIdx.x;
    long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = __builtin_amdgcn_ds_bpermute(((x + 1) & 63) << 2, x);
    long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = double(t1 - t0) / N;
}
__global__ void shfl64_fma_chain(double *out, double a, double b) {
    double x = out[threadIdx.x];
    long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = fma(__shfl_up(x, 1), a, b);
    long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = double(t1 - t0) / N;
}
__global__ void dpp_fma_chain(double *out, double a, double b) {
    double x = out[threadIdx.x];
    long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) {
        int lo = __double2loint(x), hi = __double2hiint(x);
        lo = __builtin_amdgcn_update_dpp(lo, lo, 0x138, 0xf, 0xf, false);   // wave_shr:1
        hi = __builtin_amdgcn_update_dpp(hi, hi, 0x138, 0xf, 0xf, false);
        x = fma(__hiloint2double(hi, lo), a, b);
    }
    long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = double(t1 - t0) / N;
}
__global__ void lds_read_chain(double *out) {
    __shared__ int s[256];
    s[threadIdx.x] = (threadIdx.x + 1) & 63;
    __syncthreads();
    int x = threadIdx.x;
    long t0 = clock64();
#pragma unroll 16
    for (int i = 0; i < N; ++i) x = s[x];
    long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = double(t1 - t0) / N;
}
__global__ void div64_chain(double *out, double d) {
    double x = out[threadIdx.x];
    long t0 = clock64();
#pragma unroll 4
    for (int i = 0; i < N; ++i) x = 1.0 + x / d;
    long t1 = clock64();
    out[threadIdx.x] = x;
    if (threadIdx.x == 0) out[64] = double(t1 - t0) / N;
}

int main() {
    double *d;
    hipMalloc(&d, 1024);
    hipMemset(d, 0, 1024);
    double h = 0;
    float hf = 0;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    auto report = [&](const char *name, bool is_float) {
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms;
        hipEventElapsedTime(&ms, e0, e1);
        if (is_float) { hipMemcpy(&hf, (float *)d + 64, 4, hipMemcpyDeviceToHost); h = hf; }
        else hipMemcpy(&h, d + 64, 8, hipMemcpyDeviceToHost);
        printf("%-34s %8.1f clock64 ticks per link   (%.3f us per link by events, launch included)\n", name, h, 1e3 * ms / N);
    };
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0); fma64_chain<<<1, 64>>>(d, 0.999, 0.001); report("v_fma_f64 chain", false);
        hipEventRecord(e0); fma32_chain<<<1, 64>>>((float *)d, 0.999f, 0.001f); report("v_fma_f32 chain", true);
        hipEventRecord(e0); bperm_chain<<<1, 64>>>(d); report("ds_bpermute_b32 chain", false);
        hipEventRecord(e0); lds_read_chain<<<1, 64>>>(d); report("ds_read_b32 chain", false);
        hipEventRecord(e0); shfl64_fma_chain<<<1, 64>>>(d, 0.999, 0.001); report("__shfl_up(f64) + fma chain", false);
        hipEventRecord(e0); dpp_fma_chain<<<1, 64>>>(d, 0.999, 0.001); report("dpp wave_shr:1 (f64) + fma chain", false);
        hipEventRecord(e0); div64_chain<<<1, 64>>>(d, 3.0); report("1 + x / d (f64) chain", false);
    }
    return 0;
}
