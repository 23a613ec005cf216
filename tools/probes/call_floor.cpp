// The fixed cost of a single-frame call on this system: upload of a 752x480 frame (pinned) + a 130 KB block, a captured graph
// of N empty kernels + one 130 KB download, one stream synchronisation.  What orbfe_track_frame would cost with kernels that
// take no time.  build: hipcc --offload-arch=gfx950 -O2 -o call_floor.bin call_floor.cpp
#include <hip/hip_runtime.h>
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <vector>

__global__ void empty_kernel(int* p) { if (p && threadIdx.x == 1000) *p = 1; }

static double median_us(std::vector<double>& v) { std::sort(v.begin(), v.end()); return v[v.size() / 2]; }

int main()
{
    const size_t frame = 752 * 480, small = 130 * 1024, out = 130 * 1024;
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    void *hF, *hS, *hO, *dF, *dS, *dO;
    (void)hipHostMalloc(&hF, frame); (void)hipHostMalloc(&hS, small); (void)hipHostMalloc(&hO, out);
    (void)hipMalloc(&dF, frame); (void)hipMalloc(&dS, small); (void)hipMalloc(&dO, out);
    for (int nk : {0, 1, 11, 16}) {
        hipStream_t cs;
        (void)hipStreamCreateWithFlags(&cs, hipStreamNonBlocking);
        hipGraph_t g;
        hipGraphExec_t ge;
        (void)hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        for (int k = 0; k < nk; k++) hipLaunchKernelGGL(empty_kernel, dim3(64), dim3(256), 0, cs, (int*)nullptr);
        (void)hipMemcpyAsync(hO, dO, out, hipMemcpyDeviceToHost, cs);
        (void)hipStreamEndCapture(cs, &g);
        (void)hipGraphInstantiate(&ge, g, nullptr, nullptr, 0);
        (void)hipStreamDestroy(cs);
        std::vector<double> t;
        for (int i = 0; i < 600; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipMemcpyAsync(dF, hF, frame, hipMemcpyHostToDevice, s);
            (void)hipMemcpyAsync(dS, hS, small, hipMemcpyHostToDevice, s);
            (void)hipGraphLaunch(ge, s);
            (void)hipStreamSynchronize(s);
            const auto t1 = std::chrono::steady_clock::now();
            if (i >= 100) t.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::printf("upload 361 KB + 130 KB, graph of %2d empty kernels + 130 KB download, sync: median %.1f us\n", nk, median_us(t));
        std::vector<double> t2;
        for (int i = 0; i < 600; i++) {
            const auto t0 = std::chrono::steady_clock::now();
            (void)hipGraphLaunch(ge, s);
            (void)hipStreamSynchronize(s);
            const auto t1 = std::chrono::steady_clock::now();
            if (i >= 100) t2.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
        }
        std::printf("   the graph alone (no uploads): median %.1f us\n", median_us(t2));
        (void)hipGraphExecDestroy(ge);
        (void)hipGraphDestroy(g);
    }
    std::vector<double> t3;
    for (int i = 0; i < 600; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        (void)hipMemcpyAsync(dF, hF, frame, hipMemcpyHostToDevice, s);
        (void)hipStreamSynchronize(s);
        const auto t1 = std::chrono::steady_clock::now();
        if (i >= 100) t3.push_back(std::chrono::duration<double, std::micro>(t1 - t0).count());
    }
    std::printf("upload of the frame alone + sync: median %.1f us\n", median_us(t3));
    return 0;
}
