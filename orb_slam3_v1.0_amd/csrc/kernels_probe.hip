// kernels_probe.hip -- in-kernel clock probe (diagnostics; not part of the extraction or matching path).
//
// The shader clock the chip actually holds under a workload is not what sysfs reports (MI355X_MICROARCH.md, "DVFS give-back",
// item 6): it is delta(s_memtime) / delta(s_memrealtime) x 100 MHz -- s_memtime ticks once per shader cycle, s_memrealtime at a
// constant 100 MHz.  One wave stamps both counters, sleeps until `ticks` of the constant clock have gone by, stamps again and
// writes the two differences.  bench.py launches it on a stream of its own every few steps of the sustained region, so the
// figure is the clock the extraction kernels ran at; the wave occupies one slot of one SIMD for the `ticks` it is asked for.
#include "orbfe_internal.h"

namespace orbfe {

namespace {

__global__ __launch_bounds__(64) void clock_probe_kernel(unsigned long long* __restrict__ out, unsigned ticks)
{
    if (threadIdx.x != 0) return;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime();
    const unsigned long long r0 = __builtin_amdgcn_s_memrealtime();
    unsigned long long c1, r1;
    for (int guard = 0; guard < (1 << 22); guard++) {  // bounded: every wave leaves, whatever the counters do
        __builtin_amdgcn_s_sleep(32);
        c1 = __builtin_amdgcn_s_memtime();
        r1 = __builtin_amdgcn_s_memrealtime();
        if (r1 - r0 >= (unsigned long long)ticks) break;
    }
    out[0] = c1 - c0;  // shader cycles
    out[1] = r1 - r0;  // 100 MHz ticks
}

}  // namespace

void launch_clock_probe(hipStream_t s, unsigned long long* dOut2, unsigned ticks)
{
    hipLaunchKernelGGL(clock_probe_kernel, dim3(1), dim3(64), 0, s, dOut2, ticks);
}

}  // namespace orbfe
