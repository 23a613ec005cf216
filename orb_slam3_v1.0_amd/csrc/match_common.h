// match_common.h -- pieces shared by the two matcher translation units.
#pragma once
#include <string>

#include "match.h"

namespace orbfe {

constexpr unsigned long long kKeyNone = ~0ull;

// ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1375-1391) as 4 x 64-bit popcounts
__device__ __forceinline__ int hamming256(const uint2* a, const unsigned long long* b4)
{
    const unsigned long long* a4 = reinterpret_cast<const unsigned long long*>(a);
    return __popcll(a4[0] ^ b4[0]) + __popcll(a4[1] ^ b4[1]) + __popcll(a4[2] ^ b4[2]) + __popcll(a4[3] ^ b4[3]);
}

__device__ __forceinline__ unsigned long long shfl_xor_u64(unsigned long long v, int m)
{
    unsigned lo = (unsigned)v, hi = (unsigned)(v >> 32);
    lo = __shfl_xor(lo, m);
    hi = __shfl_xor(hi, m);
    return ((unsigned long long)hi << 32) | lo;
}

// merge two (smallest, second smallest) pairs of distinct keys
__device__ __forceinline__ void top2_merge(unsigned long long& k1, unsigned long long& k2, unsigned long long o1,
                                           unsigned long long o2)
{
    const unsigned long long lo = k1 < o1 ? k1 : o1;
    const unsigned long long hi = k1 < o1 ? o1 : k1;
    const unsigned long long s2 = k2 < o2 ? k2 : o2;
    k1 = lo;
    k2 = hi < s2 ? hi : s2;
}

// wave-wide top-2 (all 64 lanes receive the result)
__device__ __forceinline__ void wave_top2(unsigned long long& k1, unsigned long long& k2)
{
#pragma unroll
    for (int d = 32; d >= 1; d >>= 1) {
        const unsigned long long o1 = shfl_xor_u64(k1, d), o2 = shfl_xor_u64(k2, d);
        top2_merge(k1, k2, o1, o2);
    }
}

[[maybe_unused]] static __global__ void fill_kernel(int* p, int v, size_t n)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}

// grow-only arenas
inline int ensure(MatchScratch& m, size_t dBytes, size_t hBytes, std::string& err)
{
    // an asynchronous *_device launch on another stream may still be using the arenas: wait before replacing them
    if ((dBytes > m.dBytes || hBytes > m.hBytes) && m.busy && (m.d || m.hpin)) (void)hipEventSynchronize(m.busy);
    if (dBytes > m.dBytes) {
        if (m.d) (void)hipFree(m.d);
        m.d = nullptr;
        m.dBytes = 0;
        const size_t want = dBytes + dBytes / 2;
        if (hipMalloc(&m.d, want) != hipSuccess) { err = "hipMalloc(match scratch) failed"; return ORBFE_ERR_OUT_OF_MEMORY; }
        m.dBytes = want;
    }
    if (hBytes > m.hBytes) {
        if (m.hpin) (void)hipHostFree(m.hpin);
        m.hpin = nullptr;
        m.hBytes = 0;
        const size_t want = hBytes + hBytes / 2;
        if (hipHostMalloc(&m.hpin, want) != hipSuccess) { err = "hipHostMalloc(match scratch) failed"; return ORBFE_ERR_OUT_OF_MEMORY; }
        m.hBytes = want;
    }
    return ORBFE_OK;
}

struct Carver {
    size_t off = 0;
    size_t take(size_t bytes) { const size_t o = off; off = (off + bytes + 255) / 256 * 256; return o; }
};

#define MCHK(call)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return ORBFE_ERR_HIP; } \
    } while (0)

}  // namespace orbfe
