// Which XCD (and CU) does workgroup b of a 240 x 512 launch with a large LDS footprint land on?  (plane.hip maps tiles to
// workgroups assuming b % 8.)  Built on the GPU box: hipcc --offload-arch=gfx950 -shared -fPIC -o /tmp/xcc_probe.so tools/xcc_probe.hip
#include <hip/hip_runtime.h>
#include <cstdint>
__global__ void xcc_kernel(unsigned *out) {
    extern __shared__ char lds[];
    if (threadIdx.x == 0) {
        unsigned x, hw;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(x));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        out[2 * blockIdx.x] = x;
        out[2 * blockIdx.x + 1] = hw;
    }
    lds[threadIdx.x] = 1;
    const unsigned long long t0 = wall_clock64();
    while (wall_clock64() - t0 < 3000) {}          // ~30 us at 100 MHz: every workgroup of the launch is resident at once
}
extern "C" int xcc_map(int n_wg, int threads, int lds_bytes, unsigned *host_out) {
    unsigned *d = nullptr;
    if (hipMalloc(&d, size_t(n_wg) * 8) != hipSuccess) return 1;
    if (lds_bytes > 64 * 1024) hipFuncSetAttribute(reinterpret_cast<const void *>(xcc_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
    hipLaunchKernelGGL(xcc_kernel, dim3(n_wg), dim3(threads), lds_bytes, 0, d);
    if (hipDeviceSynchronize() != hipSuccess) return 2;
    hipMemcpy(host_out, d, size_t(n_wg) * 8, hipMemcpyDeviceToHost);
    hipFree(d);
    return 0;
}
