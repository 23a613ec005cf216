// kernels_distinct.hip -- MapPoint::ComputeDistinctiveDescriptors for a batch of map points (SURVEY.md section 8f,
// honourable mention): the N x N Hamming-median selection of src/MapPoint.cc:343-416.
//
// One thread per (set, row i): the median of row i is its k-th smallest distance, k = (size_t)(0.5 * (N - 1)), row
// including the zero self-distance.  Distances lie in 0..256, so the k-th smallest is found by a 9-step binary
// search over the VALUE (count of distances <= v, recomputing the popcounts from L2-resident descriptors) instead
// of materialising and sorting the row; the representative = min over rows of (median << 20 | i) by one atomicMin
// per row -- the first minimum wins, as the strict "<" of the reference (:403).  Sets are small (observations of
// one map point), the batch supplies the parallelism.
#include <algorithm>
#include <cstring>

#include "match_common.h"

namespace orbfe {

namespace {

__global__ __launch_bounds__(256) void distinct_kernel(int nRows, const int* __restrict__ rowSet, const int* __restrict__ setOff,
                                                       const uint8_t* __restrict__ desc, unsigned int* __restrict__ best)
{
    const int r = blockIdx.x * 256 + threadIdx.x;
    if (r >= nRows) return;
    const int s = rowSet[r];
    const int lo = setOff[s], hi = setOff[s + 1];
    const int N = hi - lo, i = r - lo;
    const int k = (int)(size_t)(0.5 * (double)(N - 1));
    unsigned long long d4[4];
    const unsigned long long* dp = reinterpret_cast<const unsigned long long*>(desc + (size_t)r * 32);
    d4[0] = dp[0]; d4[1] = dp[1]; d4[2] = dp[2]; d4[3] = dp[3];
    int vlo = 0, vhi = 256;  // smallest v with #{j : dist(i, j) <= v} >= k + 1
    while (vlo < vhi) {
        const int mid = (vlo + vhi) >> 1;
        int c = 0;
        for (int j = lo; j < hi; j++) {
            const int dist = j == r ? 0 : hamming256(reinterpret_cast<const uint2*>(desc + (size_t)j * 32), d4);
            c += dist <= mid;
        }
        if (c >= k + 1) vhi = mid;
        else vlo = mid + 1;
    }
    atomicMin(&best[s], ((unsigned int)vlo << 20) | (unsigned int)i);
}

}  // namespace

int distinctive_run(MatchScratch& m, hipStream_t s, int nSets, const int* setOff, const uint8_t* desc, int* bestIdx,
                    int* bestMedian, std::string& err)
{
    if (nSets == 0) return ORBFE_OK;
    const int nRows = setOff[nSets];
    for (int q = 0; q < nSets; q++) {
        const int n = setOff[q + 1] - setOff[q];
        if (n < 0 || n >= (1 << 20)) return ORBFE_ERR_INVALID_ARG;
        bestIdx[q] = -1;
        if (bestMedian) bestMedian[q] = 0;
    }
    if (nRows == 0) return ORBFE_OK;
    Carver in;
    const size_t oOff = in.take((size_t)(nSets + 1) * sizeof(int));
    const size_t oRowSet = in.take((size_t)nRows * sizeof(int));
    const size_t oDesc = in.take((size_t)nRows * 32);
    const size_t oBest = in.take((size_t)nSets * sizeof(unsigned int));
    const size_t inBytes = in.off;
    int rc = ensure(m, inBytes, inBytes + (size_t)nSets * sizeof(unsigned int) + 256, err);
    if (rc != ORBFE_OK) return rc;
    uint8_t* hp = static_cast<uint8_t*>(m.hpin);
    uint8_t* dp = static_cast<uint8_t*>(m.d);
    memcpy(hp + oOff, setOff, (size_t)(nSets + 1) * sizeof(int));
    int* rowSet = reinterpret_cast<int*>(hp + oRowSet);
    for (int q = 0; q < nSets; q++)
        for (int r = setOff[q]; r < setOff[q + 1]; r++) rowSet[r] = q;
    memcpy(hp + oDesc, desc, (size_t)nRows * 32);
    memset(hp + oBest, 0xff, (size_t)nSets * sizeof(unsigned int));
    MCHK(hipMemcpyAsync(dp, hp, inBytes, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(distinct_kernel, dim3((nRows + 255) / 256), dim3(256), 0, s, nRows, reinterpret_cast<const int*>(dp + oRowSet),
                       reinterpret_cast<const int*>(dp + oOff), dp + oDesc, reinterpret_cast<unsigned int*>(dp + oBest));
    MCHK(hipGetLastError());
    unsigned int* hBest = reinterpret_cast<unsigned int*>(hp + inBytes);
    MCHK(hipMemcpyAsync(hBest, dp + oBest, (size_t)nSets * sizeof(unsigned int), hipMemcpyDeviceToHost, s));
    MCHK(hipStreamSynchronize(s));
    for (int q = 0; q < nSets; q++)
        if (setOff[q + 1] > setOff[q]) {
            bestIdx[q] = (int)(hBest[q] & 0xFFFFFu);
            if (bestMedian) bestMedian[q] = (int)(hBest[q] >> 20);
        }
    return ORBFE_OK;
}

}  // namespace orbfe
