// kernels_vocab.hip -- DBoW2 vocabulary-tree descent on gfx950.
//
// Replaces the per-feature part of Frame::ComputeBoW (src/Frame.cc:483-495) ->
// TemplatedVocabulary::transform(feature, word_id, weight, nid, levelsup)
// (Thirdparty/DBoW2/include/DBoW2/TemplatedVocabulary.h:1227-1270): starting at the root, at every
// level take the child with the smallest FORB::distance (256-bit Hamming, Thirdparty/DBoW2/src/
// DBoW2/FORB.cpp:81-101), FIRST minimum wins (:1250-1258, strict <), remember the node reached at
// level L - levelsup, stop at a leaf.  The BowVector / FeatureVector maps (:1136-1204) are assembled on
// the host by the adaptor from the per-feature (word, weight, node) triples this kernel returns.
// One group of G lanes per feature (G = 16 for branching factors <= 16, else 64): lane j takes child j,
// the arg-min is a butterfly over (distance << 20 | child position).
// SPEC DECISION S7: if a leaf is reached above level L - levelsup the reference leaves `nid`
// uninitialised (:1141-1148 declare it without a value); here nid = the leaf's node id.
#include <cstring>
#include <vector>

#include "match_common.h"
#include "vocab.h"

namespace orbfe {

namespace {

template <int G>
__global__ __launch_bounds__(256) void vocab_transform_kernel(const int* __restrict__ childOff, const int* __restrict__ childIdx,
                                                              const uint8_t* __restrict__ nodeDesc, const int* __restrict__ wordId,
                                                              const uint8_t* __restrict__ desc, int n, const int* __restrict__ nDev,
                                                              int nidLevel, int* __restrict__ out /* [n][2]: word id, node id */,
                                                              int* __restrict__ leafOut /* [n]: leaf node (weight lookup) */,
                                                              int* __restrict__ clearOut /* optional [capacity]: set to -1 */)
{
    const int gid = (blockIdx.x * 256 + threadIdx.x) / G;  // feature
    const int gl = threadIdx.x % G;
    // (fused per-frame chain: the matcher's output array starts as "no match" -- one launch less than a separate fill)
    if (clearOut && gl == 0 && gid < n) clearOut[gid] = -1;
    if (nDev) n = min(n, *nDev);  // the count is still on the device (fused per-frame chain): n is the capacity
    if (gid >= n) return;  // whole groups exit together (256 % G == 0)
    unsigned long long d4[4];
    const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(desc + (size_t)gid * 32);
    d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
    int node = 0, nid = nidLevel <= 0 ? 0 : -1, level = 0;
    for (int guard = 0; guard < 64; guard++) {
        const int c0 = childOff[node], c1 = childOff[node + 1];
        if (c1 == c0) break;  // leaf (only possible at entry for a degenerate root)
        ++level;
        unsigned int best = 0xffffffffu;
        for (int j = gl; j < c1 - c0; j += G) {
            const int child = childIdx[c0 + j];
            const unsigned d = (unsigned)hamming256(reinterpret_cast<const uint2*>(nodeDesc + (size_t)child * 32), d4);
            const unsigned key = (d << 20) | (unsigned)j;  // first minimum wins
            best = key < best ? key : best;
        }
#pragma unroll
        for (int m = G / 2; m >= 1; m >>= 1) {
            const unsigned o = __shfl_xor(best, m);
            best = o < best ? o : best;
        }
        node = childIdx[c0 + (int)(best & 0xFFFFFu)];
        if (level == nidLevel) nid = node;
        if (childOff[node + 1] == childOff[node]) break;  // isLeaf()
    }
    if (nid < 0) nid = node;  // S7
    if (gl == 0) {
        out[2 * gid] = wordId[node];
        out[2 * gid + 1] = nid;
        leafOut[gid] = node;
    }
}

#define VCHK(call)                                                                          \
    do {                                                                                    \
        hipError_t e_ = (call);                                                             \
        if (e_ != hipSuccess) { err = std::string(#call) + ": " + hipGetErrorString(e_); return ORBFE_ERR_HIP; } \
    } while (0)

}  // namespace

int vocab_create(int nNodes, const int* childOff, const int* childIdx, const uint8_t* nodeDesc, const int* wordId,
                 const double* weight, int L, hipStream_t s, Vocab** out, std::string& err)
{
    *out = nullptr;
    if (nNodes < 2 || childOff[0] != 0 || childOff[1] == 0) return ORBFE_ERR_INVALID_ARG;  // root must have children
    int maxCh = 0;
    std::vector<char> seen(nNodes, 0);
    for (int i = 0; i < nNodes; i++) {
        const int nc = childOff[i + 1] - childOff[i];
        if (nc < 0) return ORBFE_ERR_INVALID_ARG;
        maxCh = std::max(maxCh, nc);
        for (int j = childOff[i]; j < childOff[i + 1]; j++) {
            const int c = childIdx[j];
            if (c <= 0 || c >= nNodes || seen[c]) return ORBFE_ERR_INVALID_ARG;  // a tree: every node has one parent
            seen[c] = 1;
        }
    }
    if (maxCh >= (1 << 20)) return ORBFE_ERR_UNSUPPORTED;
    Vocab* v = new Vocab();
    v->nNodes = nNodes; v->L = L; v->maxChildren = maxCh;
    v->hWeight.assign(weight, weight + nNodes);
    const int nEdges = childOff[nNodes];
#define VC(call) do { if ((call) != hipSuccess) { err = #call; vocab_destroy(v); return ORBFE_ERR_OUT_OF_MEMORY; } } while (0)
    VC(hipMalloc(&v->dChildOff, (size_t)(nNodes + 1) * sizeof(int)));
    VC(hipMalloc(&v->dChildIdx, (size_t)std::max(nEdges, 1) * sizeof(int)));
    VC(hipMalloc(&v->dDesc, (size_t)nNodes * 32));
    VC(hipMalloc(&v->dWordId, (size_t)nNodes * sizeof(int)));
    VC(hipMalloc(&v->dWeight, (size_t)nNodes * sizeof(double)));
    VC(copy_sync(v->dChildOff, childOff, (size_t)(nNodes + 1) * sizeof(int), hipMemcpyHostToDevice, s));
    VC(copy_sync(v->dChildIdx, childIdx, (size_t)nEdges * sizeof(int), hipMemcpyHostToDevice, s));
    VC(copy_sync(v->dDesc, nodeDesc, (size_t)nNodes * 32, hipMemcpyHostToDevice, s));
    VC(copy_sync(v->dWordId, wordId, (size_t)nNodes * sizeof(int), hipMemcpyHostToDevice, s));
    VC(copy_sync(v->dWeight, weight, (size_t)nNodes * sizeof(double), hipMemcpyHostToDevice, s));
#undef VC
    *out = v;
    return ORBFE_OK;
}

void vocab_destroy(Vocab* v)
{
    if (!v) return;
    void* d[] = {v->dChildOff, v->dChildIdx, v->dDesc, v->dWordId, v->dWeight, v->dIn, v->dOut};
    for (void* p : d)
        if (p) (void)hipFree(p);
    if (v->hpin) (void)hipHostFree(v->hpin);
    delete v;
}

int vocab_transform(Vocab* v, hipStream_t s, const uint8_t* desc, int n, int levelsup, int* wordOut, int* nodeOut,
                    double* weightOut, std::string& err)
{
    if (n == 0) return ORBFE_OK;
    if ((size_t)n > v->cap) {
        if (v->dIn) (void)hipFree(v->dIn);
        if (v->dOut) (void)hipFree(v->dOut);
        if (v->hpin) (void)hipHostFree(v->hpin);
        v->dIn = nullptr; v->dOut = nullptr; v->hpin = nullptr; v->cap = 0;
        const size_t cap = (size_t)n + n / 2;
        VCHK(hipMalloc(&v->dIn, cap * 32));
        VCHK(hipMalloc(&v->dOut, cap * 3 * sizeof(int)));
        VCHK(hipHostMalloc(&v->hpin, cap * (32 + 3 * sizeof(int))));
        v->cap = cap;
    }
    uint8_t* hp = static_cast<uint8_t*>(v->hpin);
    int* hOut = reinterpret_cast<int*>(hp + v->cap * 32);
    memcpy(hp, desc, (size_t)n * 32);
    VCHK(hipMemcpyAsync(v->dIn, hp, (size_t)n * 32, hipMemcpyHostToDevice, s));
    const int nidLevel = v->L - levelsup;  // TemplatedVocabulary.h:1235
    int* dLeaf = v->dOut + 2 * (size_t)n;
    if (v->maxChildren <= 16)
        hipLaunchKernelGGL(vocab_transform_kernel<16>, dim3((n * 16 + 255) / 256), dim3(256), 0, s, v->dChildOff, v->dChildIdx,
                           v->dDesc, v->dWordId, v->dIn, n, nullptr, nidLevel, v->dOut, dLeaf, nullptr);
    else
        hipLaunchKernelGGL(vocab_transform_kernel<64>, dim3((n * 64 + 255) / 256), dim3(256), 0, s, v->dChildOff, v->dChildIdx,
                           v->dDesc, v->dWordId, v->dIn, n, nullptr, nidLevel, v->dOut, dLeaf, nullptr);
    VCHK(hipGetLastError());
    VCHK(hipMemcpyAsync(hOut, v->dOut, (size_t)n * 3 * sizeof(int), hipMemcpyDeviceToHost, s));
    VCHK(hipStreamSynchronize(s));
    for (int i = 0; i < n; i++) {
        wordOut[i] = hOut[2 * i];
        nodeOut[i] = hOut[2 * i + 1];
        if (weightOut) weightOut[i] = v->hWeight[hOut[2 * (size_t)n + i]];  // m_nodes[final_id].weight, :1268
    }
    return ORBFE_OK;
}

// the descent on descriptors that are already on the device, their count included (orbfe_track_reference_keyframe: the
// extraction chain wrote them a kernel earlier); no copy, no synchronisation -- capturable
int vocab_transform_launch_dev(const Vocab* v, hipStream_t s, const uint8_t* dDesc, const int* dN, int cap, int levelsup,
                               int* dOut, int* dLeaf, int* dClear, std::string& err)
{
    if (cap <= 0) return ORBFE_OK;
    const int nidLevel = v->L - levelsup;
    if (v->maxChildren <= 16)
        hipLaunchKernelGGL(vocab_transform_kernel<16>, dim3((cap * 16 + 255) / 256), dim3(256), 0, s, v->dChildOff, v->dChildIdx,
                           v->dDesc, v->dWordId, dDesc, cap, dN, nidLevel, dOut, dLeaf, dClear);
    else
        hipLaunchKernelGGL(vocab_transform_kernel<64>, dim3((cap * 64 + 255) / 256), dim3(256), 0, s, v->dChildOff, v->dChildIdx,
                           v->dDesc, v->dWordId, dDesc, cap, dN, nidLevel, dOut, dLeaf, dClear);
    VCHK(hipGetLastError());
    return ORBFE_OK;
}

}  // namespace orbfe
