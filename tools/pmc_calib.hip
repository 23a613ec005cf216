// pmc_calib.hip -- known-byte-count kernels to calibrate FETCH_SIZE / WRITE_SIZE on gfx950 for the
// access widths the ORB kernels use (MI355X_MICROARCH.md, HBM section: the counters are only
// calibrated for 16 B/lane streams; "calibrate on a known byte count in your own access pattern").
//   copy4  : 4 B/lane coalesced loads + 4 B/lane stores   (resize / fast_blur staging + outputs)
//   copy16 : 16 B/lane loads + stores                      (guide's reference pattern)
// Buffer 1 GiB (>> 256 MiB Infinity Cache) so hits do not hide traffic.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ void copy4(const unsigned* __restrict__ a, unsigned* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] + 1u;
}
__global__ void copy16(const uint4* __restrict__ a, uint4* __restrict__ b, size_t n)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        uint4 v = a[i];
        v.x += 1u;
        b[i] = v;
    }
}
int main()
{
    const size_t bytes = 1ull << 30;
    void *a, *b;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
    hipMemset(a, 1, bytes);
    hipMemset(b, 0, bytes);
    hipDeviceSynchronize();
    for (int r = 0; r < 2; r++) {
        hipLaunchKernelGGL(copy4, dim3(8192), dim3(256), 0, 0, (const unsigned*)a, (unsigned*)b, bytes / 4);
        hipLaunchKernelGGL(copy16, dim3(8192), dim3(256), 0, 0, (const uint4*)a, (uint4*)b, bytes / 16);
    }
    hipDeviceSynchronize();
    printf("calib bytes per kernel: read %zu write %zu\n", bytes, bytes);
    return 0;
}
