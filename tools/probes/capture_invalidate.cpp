// What a stream looks like after another thread's NULL-stream copy met its capture (ROCm 7.x), per capture mode:
// does the other thread's hipMemcpy fail, is the capture invalidated, does the stream work again, can it be destroyed.
// build: hipcc --offload-arch=gfx950 -O1 -o capture_invalidate.bin capture_invalidate.cpp -lpthread
#include <hip/hip_runtime.h>
#include <atomic>
#include <cstdio>
#include <thread>
#include <vector>

static void run(hipStreamCaptureMode mode, const char* name, unsigned flags, const char* fname, bool otherAsync)
{
    std::printf("---- capture mode %s, stream %s, other thread uses %s ----\n", name, fname,
                otherAsync ? "hipMemcpyAsync on its own stream + sync" : "hipMemcpy (NULL stream)");
    hipStream_t s;
    (void)hipStreamCreateWithFlags(&s, flags);
    void *d = nullptr, *d2 = nullptr;
    (void)hipMalloc(&d, 1 << 20);
    (void)hipMalloc(&d2, 1 << 16);
    std::vector<unsigned char> host(1 << 16);
    std::atomic<int> phase{0};
    std::thread other([&] {
        hipStream_t so = nullptr;
        if (otherAsync) (void)hipStreamCreateWithFlags(&so, hipStreamNonBlocking);
        while (phase.load() == 0) {}
        hipError_t e;
        if (otherAsync) {
            e = hipMemcpyAsync(d2, host.data(), host.size(), hipMemcpyHostToDevice, so);
            if (e == hipSuccess) e = hipStreamSynchronize(so);
        } else {
            e = hipMemcpy(d2, host.data(), host.size(), hipMemcpyHostToDevice);
        }
        std::printf("  other thread: copy -> %d (%s)\n", (int)e, hipGetErrorString(e));
        void* tmp = nullptr;
        e = hipMalloc(&tmp, 1 << 20);
        hipError_t e2 = hipFree(tmp);
        std::printf("  other thread: hipMalloc -> %d, hipFree -> %d (%s)\n", (int)e, (int)e2, hipGetErrorString(e2));
        (void)hipGetLastError();
        phase.store(2);
    });
    hipError_t e = hipStreamBeginCapture(s, mode);
    e = hipMemsetAsync(d, 0, 1 << 20, s);
    phase.store(1);
    while (phase.load() != 2) {}
    e = hipMemsetAsync(d, 0, 1 << 20, s);
    std::printf("  memset in capture after the other thread's calls -> %d (%s)\n", (int)e, hipGetErrorString(e));
    hipGraph_t g = nullptr;
    e = hipStreamEndCapture(s, &g);
    std::printf("  end capture -> %d (%s), graph %s\n", (int)e, hipGetErrorString(e), g ? "yes" : "no");
    (void)hipGetLastError();
    e = hipMemsetAsync(d, 0, 1 << 20, s);
    std::printf("  plain memset on the stream afterwards -> %d\n", (int)e);
    (void)hipGetLastError();
    e = hipStreamDestroy(s);
    std::printf("  destroy the stream -> %d (%s)\n", (int)e, hipGetErrorString(e));
    (void)hipGetLastError();
    other.join();
    (void)hipFree(d);
    (void)hipFree(d2);
}

int main()
{
    run(hipStreamCaptureModeThreadLocal, "ThreadLocal", hipStreamNonBlocking, "non-blocking", false);
    run(hipStreamCaptureModeRelaxed, "Relaxed", hipStreamNonBlocking, "non-blocking", false);
    run(hipStreamCaptureModeGlobal, "Global", hipStreamNonBlocking, "non-blocking", false);
    run(hipStreamCaptureModeThreadLocal, "ThreadLocal", hipStreamNonBlocking, "non-blocking", true);
    return 0;
}
