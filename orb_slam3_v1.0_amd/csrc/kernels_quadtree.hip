// kernels_quadtree.hip -- DistributeOctTree + best-per-node on gfx950, one workgroup per
// (frame, level), bit-exact against the sequential std::list algorithm of the reference.
//
// Replaces ORBextractor::DistributeOctTree (src/ORBextractor.cc:226-431), ExtractorNode::DivideNode
// (:151-207), compareNodes (:209-224), the two-threshold bookkeeping of ComputeKeyPointsOctTree
// (:449-482) and the best-response-per-node loop (:505-527), all of which run on the CPU in the
// reference.
//
// How the list choreography becomes data-parallel:
//  * The node list is kept as an ARRAY in list order (index 0 == list head).  A pass that splits
//    the nodes e_0..e_{J-1} (in processing order) and push_front()s their non-empty children
//    n1..n4 produces   new list = reverse(flatmap(e_r -> children ascending)) ++ (old list minus
//    the split nodes, order kept).  Positions follow from two prefix sums.
//  * Uniform phase (:283-361): processing order == list order, every node with > 1 point splits.
//  * Sorted phase (:362-427): processing order == descending (count, UL.x, UL.y, creation seq)
//    (S4 total order); the mid-pass early break (:419-420) is the first r whose running size
//    n + sum(children-1) reaches N (children-1 >= 0, so the running size is monotone).
//  * Points never move: each candidate carries the index of its node (u16 in HBM scratch) and is
//    re-labelled once per round; per-child counts are integer LDS atomics (order-independent).
//  * The reference feeds the tree the concatenation [high-threshold list, low-threshold list], so
//    a strong corner appears twice.  Duplicates only ever act through vKeys.size(): a candidate
//    gets weight (score >= iniTh) + (retry pass taken && inside the low-list cap).
//  * Best point per node: max response, ties -> first in vKeys order == smallest raster key.
#include "launch.h"
#include "fast_common.h"

namespace orbfe {

constexpr int kQtThreads = 512;   // block size of the throughput launches (many frames: four blocks share a CU)
constexpr int kQtThreadsSmall = 1024;  // a single frame's call: half as many trips through every candidate loop
constexpr int kQtMaxRounds = 48;
constexpr uint32_t kKeyInf = 0x01000000u;  // larger than any raster key

struct Box {
    int16_t ulx, urx, uly, bry;
};

__device__ __forceinline__ int child_code(const Box& b, int x, int y)
{
    const int halfX = (b.urx - b.ulx + 1) >> 1;  // ceil(float(UR.x-UL.x)/2), :153
    const int halfY = (b.bry - b.uly + 1) >> 1;  // :154
    const int right = !(x < b.ulx + halfX);      // kp.pt.x < n1.UR.x, :188
    const int bottom = !(y < b.uly + halfY);     // kp.pt.y < n1.BR.y, :190
    return right + 2 * bottom;                   // n1=0 n2=1 n3=2 n4=3
}

__device__ __forceinline__ Box child_box(const Box& b, int c)
{
    const int halfX = (b.urx - b.ulx + 1) >> 1;
    const int halfY = (b.bry - b.uly + 1) >> 1;
    Box r;
    r.ulx = (c & 1) ? (int16_t)(b.ulx + halfX) : b.ulx;
    r.urx = (c & 1) ? b.urx : (int16_t)(b.ulx + halfX);
    r.uly = (c & 2) ? (int16_t)(b.uly + halfY) : b.uly;
    r.bry = (c & 2) ? b.bry : (int16_t)(b.uly + halfY);
    return r;
}

// exclusive scan of one int per thread across the block (tid order); all threads must call.
template <int NT>
__device__ __forceinline__ int block_excl_scan(int v, int& total, int* sWave)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    // wave-level inclusive scan with DPP row shifts (one VALU instruction per step; the LDS permute of __shfl_up costs
    // a round trip per step and this kernel is a chain of scans): Hillis-Steele inside the 16-lane rows, then the row
    // totals ripple through row_bcast 15 / 31.  Shifted-in and masked-off lanes read 0.
    int incl = v;
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x111, 0xF, 0xF, true);  // row_shr:1
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x112, 0xF, 0xF, true);  // row_shr:2
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x114, 0xF, 0xF, true);  // row_shr:4
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x118, 0xF, 0xF, true);  // row_shr:8
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x142, 0xA, 0xF, true);  // row_bcast:15 into rows 1 and 3
    incl += __builtin_amdgcn_update_dpp(0, incl, 0x143, 0xC, 0xF, true);  // row_bcast:31 into rows 2 and 3
    __syncthreads();  // protect sWave from the previous use
    if (lane == 63) sWave[wave] = incl;
    __syncthreads();
    int base = 0, tot = 0;
#pragma unroll
    for (int i = 0; i < NT / 64; i++) {
        const int s = sWave[i];
        if (i < wave) base += s;
        tot += s;
    }
    total = tot;
    return base + incl - v;
}

// Exclusive raster-key bound of one FAST pass' pre-NMS cap: key of the nFast-th corner (raster order)
// of this level, plus one.  The cutoff row comes from the per-tile-row counts the FAST kernel
// stored; the cutoff column from re-running the segment test along that single row.  Block-wide,
// every thread returns the same value.  Only called when the level has more than nFast corners.
template <int NT>
__device__ uint32_t pre_nms_cut_key(const PipelineDesc* __restrict__ P, const LevelDesc& L, int f, int l, bool high,
                                    int nFast, const uint8_t* __restrict__ gray0, size_t gray0FrameStride, int gray0Pitch,
                                    const uint8_t* __restrict__ ws, const uint16_t* __restrict__ tileRows, int* sWave,
                                    int* sRed)
{
    const int tid = threadIdx.x;
    const uint16_t* tr = tileRows + ((size_t)f * P->totalTiles + L.tileBase) * kFastTH;  // low pass | high pass << 8 per tile row
    if (tid == 0) { sRed[0] = -1; sRed[1] = 0; sRed[2] = -1; }
    __syncthreads();
    int carry = 0;
    for (int base = 0; base < L.h; base += NT) {  // row where the running corner count reaches nFast
        const int y = base + tid;
        int c = 0;
        if (y < L.h) {
            const int ty = y / kFastTH, r = y % kFastTH;
            for (int tx = 0; tx < L.tilesX; tx++) {
                const uint32_t v = tr[(size_t)(ty * L.tilesX + tx) * kFastTH + r];
                c += high ? (int)(v >> 8) : (int)(v & 0xffu);
            }
        }
        int tot;
        const int ex = block_excl_scan<NT>(c, tot, sWave) + carry;
        if (c > 0 && ex < nFast && ex + c >= nFast) { sRed[0] = y; sRed[1] = nFast - ex; }  // exactly one thread
        carry += tot;
        __syncthreads();
        if (sRed[0] >= 0) break;
    }
    const int ys = sRed[0], q = sRed[1];
    if (ys < 0) return kKeyInf;
    const uint8_t* img;
    int pitch;
    if (l == 0) {
        img = gray0 + (size_t)f * gray0FrameStride;
        pitch = gray0Pitch;
    } else {
        img = ws + L.imgOff + (size_t)f * L.imgFrameStride;
        pitch = L.pitch;
    }
    const int th = high ? P->iniTh : P->minTh;  // corner of the pass <=> score >= its threshold
    carry = 0;
    for (int base = 0; base < L.w; base += NT) {  // the q-th corner of that row
        const int x = base + tid;
        int flag = 0;
        if (x > kEdge && x < L.w - kEdge) flag = corner_at_global(img, pitch, x, ys, th) ? 1 : 0;
        int tot;
        const int ex = block_excl_scan<NT>(flag, tot, sWave) + carry;
        if (flag && ex + 1 == q) sRed[2] = x;
        carry += tot;
        __syncthreads();
        if (sRed[2] >= 0) break;
    }
    const int xs = sRed[2];
    __syncthreads();
    if (xs < 0) return kKeyInf;  // cannot happen: the row holds at least q corners
    return (((uint32_t)ys << kCoordBits) | (uint32_t)xs) + 1u;
}

// bytes of the per-block HBM slab of the GLOBAL variant (see the carve in the kernel)
__host__ __device__ constexpr size_t qt_scratch_bytes(int nc) { return (size_t)nc * (8 + 8 + 8 + 4 + 4 + 16 + 4 + 4 + 4 + 1 + 1); }

template <int NC, bool GLOBAL, int NT = kQtThreads>
__global__ __launch_bounds__(NT) __attribute__((amdgpu_waves_per_eu(NC <= 512 ? 8 : 1))) void quadtree_kernel(const PipelineDesc* __restrict__ P,
                                                             const uint32_t* __restrict__ cand,
                                                             uint16_t* __restrict__ nodeOfAll,
                                                             uint32_t* __restrict__ counters,
                                                             uint32_t* __restrict__ lvlKp,
                                                             const uint8_t* __restrict__ gray0, size_t gray0FrameStride,
                                                             int gray0Pitch, const uint8_t* __restrict__ ws,
                                                             const uint16_t* __restrict__ tileRows,
                                                             uint8_t* __restrict__ scratch)
{
    constexpr int IPT = NC / NT;  // nodes per thread in node-parallel steps
    static_assert(NC % NT == 0, "NC must be a multiple of the block size");

    // Node tables: LDS for the usual per-level budgets (<= 2048 nodes); for the reference's large
    // configurations (e.g. 10000 features in one level, mono_inertial_node.cpp:87-93) the same
    // tables live in an HBM scratch slab per block (L2-resident, slower, same algorithm).
    Box* sBox[2];
    int* sCnt[2];
    int* sCC;                         // child counts, then child new positions
    int* sKeep;                       // new position of a non-split node
    int* sS;                          // flat-map offset of a split node's first child
    uint8_t* sSplit;
    uint8_t* sNch;
    unsigned long long* sKey;
    int* sProc;                       // sorted phase: rank -> node
    if constexpr (!GLOBAL) {
        __shared__ Box lBox[2][NC];
        __shared__ int lCnt[2][NC];
        __shared__ int lCC[NC * 4];
        __shared__ int lKeep[NC];
        __shared__ int lS[NC];
        __shared__ uint8_t lSplit[NC];
        __shared__ uint8_t lNch[NC];
        __shared__ unsigned long long lKey[NC];
        __shared__ int lProc[NC];
        sBox[0] = lBox[0]; sBox[1] = lBox[1]; sCnt[0] = lCnt[0]; sCnt[1] = lCnt[1];
        sCC = lCC; sKeep = lKeep; sS = lS; sSplit = lSplit; sNch = lNch; sKey = lKey; sProc = lProc;
    } else {
        uint8_t* base = scratch + ((size_t)blockIdx.y * gridDim.x + blockIdx.x) * qt_scratch_bytes(NC);
        sKey = reinterpret_cast<unsigned long long*>(base); base += (size_t)NC * 8;
        sBox[0] = reinterpret_cast<Box*>(base); base += (size_t)NC * 8;
        sBox[1] = reinterpret_cast<Box*>(base); base += (size_t)NC * 8;
        sCnt[0] = reinterpret_cast<int*>(base); base += (size_t)NC * 4;
        sCnt[1] = reinterpret_cast<int*>(base); base += (size_t)NC * 4;
        sCC = reinterpret_cast<int*>(base); base += (size_t)NC * 16;
        sKeep = reinterpret_cast<int*>(base); base += (size_t)NC * 4;
        sS = reinterpret_cast<int*>(base); base += (size_t)NC * 4;
        sProc = reinterpret_cast<int*>(base); base += (size_t)NC * 4;
        sSplit = base; base += NC;
        sNch = base;
    }
    __shared__ int sWave[NT / 64];
    __shared__ int sRed[4];

    const int f = blockIdx.x, l = blockIdx.y;
    const int nL = P->nLevels;
    const LevelDesc& L = P->lv[l];
    const int tid = threadIdx.x;
    uint32_t* cnt = counters + ((size_t)f * nL + l) * kCntWords;
    const uint32_t* C = cand + L.candOff + (size_t)f * L.candCap;
    uint16_t* nodeOf = nodeOfAll + L.candOff + (size_t)f * L.candCap;
    uint32_t* out = lvlKp + (size_t)f * P->kpCapFrame + L.kpBase;

    const int N = L.nFeatures;
    const int nFast = P->nFast;
    const int iniTh = P->iniTh;
    const int nAll = (int)min(cnt[kCntCand], (uint32_t)L.candCap);  // NMS survivors stored by the FAST kernel
    // ---- the reference's caps (S2b: raster-first entries survive) ----
    // (i) GpuFast::detect keeps only the first maxKeypoints PRE-NMS corners of a pass
    //     (src/cuda/Fast_gpu.cu:278-281,377); a survivor counts for a pass iff its raster key is below the
    //     key of that pass' nFast-th corner.  Rare: only when a level has more than nFast corners.
    uint32_t cutHi = kKeyInf, cutLo = kKeyInf;
    int cH = (int)cnt[kCntHigh];
    if ((int)cnt[kCntPreHigh] > nFast) {
        cutHi = pre_nms_cut_key<NT>(P, L, f, l, true, nFast, gray0, gray0FrameStride, gray0Pitch, ws, tileRows, sWave, sRed);
        int c = 0;
        for (int p = tid; p < nAll; p += NT) c += cand_score(C[p]) >= iniTh && cand_key(C[p]) < cutHi;
        block_excl_scan<NT>(c, cH, sWave);
    }
    // retry rule, src/ORBextractor.cc:440,463-465 (unsigned diff, double compare)
    const unsigned diff = (unsigned)nFast - (unsigned)cH;
    const bool retry = (double)diff > 0.25 * (double)nFast;
    int cL = nAll;
    if (retry && (int)cnt[kCntPreLow] > nFast) {
        cutLo = pre_nms_cut_key<NT>(P, L, f, l, false, nFast, gray0, gray0FrameStride, gray0Pitch, ws, tileRows, sWave, sRed);
        int c = 0;
        for (int p = tid; p < nAll; p += NT) c += cand_key(C[p]) < cutLo;
        block_excl_scan<NT>(c, cL, sWave);
    }
    // (ii) high + low lists together are capped at nFast by trimming the low list's tail (:470-473)
    uint32_t keyCut = cutLo;
    int lowKept = 0;
    if (retry && cL > 0) {
        lowKept = cL;
        if (cH + cL > nFast) lowKept = max(nFast - cH, 0);
    }
    if (retry && lowKept < cL) {
        // smallest key K with #{key < K} >= lowKept  (binary search over the 24-bit key)
        uint32_t lo = 0, hi = kKeyInf;
        while (lo < hi) {
            const uint32_t mid = lo + ((hi - lo) >> 1);
            int c = 0;
            for (int p = tid; p < nAll; p += NT) c += cand_key(C[p]) < mid;
            int tot;
            block_excl_scan<NT>(c, tot, sWave);
            if (tot >= lowKept) hi = mid; else lo = mid + 1;
        }
        keyCut = min(lo, cutLo);
    }
    cL = nAll;  // loops below run over every stored survivor; the weights apply the cuts
    const int totalPts = cH + lowKept;
    if (totalPts == 0) {
        if (tid == 0) cnt[kCntKp] = 0;
        return;
    }
    if (L.nodeCap > NC) {  // host picks NC >= nodeCap; guard anyway
        if (tid == 0) { cnt[kCntKp] = 0; atomicOr(&cnt[kCntStatus], (uint32_t)kFlagNodeOverflow); }
        return;
    }

#define PT_WEIGHT(cw) ((int)(cand_score(cw) >= iniTh && cand_key(cw) < cutHi) + (int)(retry && cand_key(cw) < keyCut))

    // ---- initial nodes, :231-274 ----
    const int nIni = L.nIni;
    const float hX = L.hX;
    for (int i = tid; i < NC; i += NT) sCnt[0][i] = 0;
    __syncthreads();
    for (int p = tid; p < cL; p += NT) {
        const uint32_t cw = C[p];
        const int wgt = PT_WEIGHT(cw);
        if (wgt) atomicAdd(&sCnt[0][(int)((float)cand_x(cw) / hX)], wgt);
    }
    __syncthreads();
    // compact away empty initial nodes (nIni is tiny: serial on one thread)
    if (tid == 0) {
        int n = 0;
        for (int i = 0; i < nIni; i++) {
            const int c = sCnt[0][i];
            sKeep[i] = n;
            if (c > 0) {
                Box b;
                b.ulx = (int16_t)(int)(hX * (float)i);
                b.urx = (int16_t)(int)(hX * (float)(i + 1));
                b.uly = 0;
                b.bry = (int16_t)L.h;
                sBox[1][n] = b;
                sCnt[1][n] = c;
                n++;
            }
        }
        sRed[0] = n;
    }
    __syncthreads();
    int n = sRed[0];
    for (int p = tid; p < cL; p += NT) {
        const uint32_t cw = C[p];
        if (PT_WEIGHT(cw)) nodeOf[p] = (uint16_t)sKeep[(int)((float)cand_x(cw) / hX)];
    }
    int cur = 1;  // buffer holding the current list
    bool sortedMode = false;
    bool overflow = false;

    for (int round = 0; round < kQtMaxRounds; round++) {
        const int prevSize = n;
        // constant indices + select (not sBox[cur]): keeps the pointer arrays in registers so the
        // compiler can still prove the LDS address space (otherwise every access becomes a flat op)
        Box* box = cur ? sBox[1] : sBox[0];
        int* ncnt = cur ? sCnt[1] : sCnt[0];
        Box* nbox = cur ? sBox[0] : sBox[1];
        int* nncnt = cur ? sCnt[0] : sCnt[1];

        // 1. child counts
        for (int i = tid; i < n * 4; i += NT) sCC[i] = 0;
        __syncthreads();
        // (four candidate words and node labels in flight per thread: the passes over the candidates are
        // latency-bound, and a load / use pair per iteration would wait for every L2 round trip in turn)
        for (int p0 = tid; p0 < cL; p0 += 4 * NT) {
            uint32_t cw4[4];
            int nd4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int p = p0 + k * NT;
                cw4[k] = p < cL ? C[p] : 0u;
                nd4[k] = p < cL ? (int)nodeOf[p] : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const uint32_t cw = cw4[k];
                const int wgt = (p0 + k * NT < cL) ? PT_WEIGHT(cw) : 0;
                if (wgt) {
                    const int i = nd4[k];
                    if (ncnt[i] > 1) atomicAdd(&sCC[4 * i + child_code(box[i], cand_x(cw), cand_y(cw))], wgt);
                }
            }
        }
        __syncthreads();

        // 2. per node: expandable, number of non-empty children
        int myExp[IPT], myNch[IPT];
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const int i = tid * IPT + k;
            int e = 0, nc = 0;
            if (i < n && ncnt[i] > 1) {
                e = 1;
                nc = (sCC[4 * i] > 0) + (sCC[4 * i + 1] > 0) + (sCC[4 * i + 2] > 0) + (sCC[4 * i + 3] > 0);
            }
            myExp[k] = e;
            myNch[k] = nc;
            if (i < NC) { sNch[i] = (uint8_t)nc; sSplit[i] = 0; }
        }

        int T = 0;  // total children created this round
        if (!sortedMode) {
            // 3a. uniform: every expandable node splits, processing order == list order
            int loc = 0;
#pragma unroll
            for (int k = 0; k < IPT; k++) loc += myNch[k];
            int base = block_excl_scan<NT>(loc, T, sWave);
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int i = tid * IPT + k;
                if (i < n) {
                    sS[i] = base;
                    sSplit[i] = (uint8_t)myExp[k];
                }
                base += myNch[k];
            }
        } else {
            // 3b. sorted: rank expandable nodes by descending (count, UL.x, UL.y, creation seq)
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int i = tid * IPT + k;
                if (i < n) {
                    unsigned long long key = 0;
                    if (myExp[k])
                        key = ((unsigned long long)(uint32_t)ncnt[i] << 40) |
                              ((unsigned long long)(uint16_t)box[i].ulx << 28) |
                              ((unsigned long long)(uint16_t)box[i].uly << 16) |
                              (unsigned long long)(0xFFFFu - (uint32_t)i);
                    sKey[i] = key;  // 0 == not expandable (any real key is > 0)
                }
            }
            __syncthreads();
            int locM = 0;
#pragma unroll
            for (int k = 0; k < IPT; k++) locM += myExp[k];
            int m;
            block_excl_scan<NT>(locM, m, sWave);
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int i = tid * IPT + k;
                if (i < n && myExp[k]) {
                    const unsigned long long key = sKey[i];
                    int r = 0;
                    for (int j = 0; j < n; j++) r += sKey[j] > key;
                    sProc[r] = i;
                }
            }
            __syncthreads();
            // running size after processing rank r: n + sum_{q<=r} (nch-1); J = first r reaching N
            int locInc[IPT], locSum = 0;
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int r = tid * IPT + k;
                locInc[k] = (r < m) ? (int)sNch[sProc[r]] - 1 : 0;
                locSum += locInc[k];
            }
            int totInc;
            int run = block_excl_scan<NT>(locSum, totInc, sWave);
            int myJ = m;  // candidate: first rank where the inclusive running size >= N
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int r = tid * IPT + k;
                run += locInc[k];
                if (r < m && n + run >= N && myJ == m) myJ = r + 1;
            }
            if (tid == 0) sRed[1] = m;
            __syncthreads();
            if (myJ < m) atomicMin(&sRed[1], myJ);
            __syncthreads();
            const int J = sRed[1];
            // flat-map offsets over the processed prefix (rank order)
            int locN[IPT], locT = 0;
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int r = tid * IPT + k;
                locN[k] = (r < J) ? (int)sNch[sProc[r]] : 0;
                locT += locN[k];
            }
            int base = block_excl_scan<NT>(locT, T, sWave);
#pragma unroll
            for (int k = 0; k < IPT; k++) {
                const int r = tid * IPT + k;
                if (r < J) {
                    const int i = sProc[r];
                    sS[i] = base;
                    sSplit[i] = 1;
                }
                base += locN[k];
            }
        }
        __syncthreads();

        // 4. positions of the nodes that stay
        int locK = 0;
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const int i = tid * IPT + k;
            locK += (i < n && !sSplit[i]) ? 1 : 0;
        }
        int K;
        int kbase = block_excl_scan<NT>(locK, K, sWave);
        const int newSize = T + K;
        if (newSize > NC) overflow = true;  // uniform over the block (T, K are block-wide totals)
        if (overflow) break;
        if (tid == 0) sRed[2] = 0;
        __syncthreads();

        // 5. build the new list
        int locExpand = 0;
#pragma unroll
        for (int k = 0; k < IPT; k++) {
            const int i = tid * IPT + k;
            if (i < n) {
                if (sSplit[i]) {
                    const Box b = box[i];
                    int kk = 0;
                    const int s = sS[i];
#pragma unroll
                    for (int c = 0; c < 4; c++) {
                        const int cc = sCC[4 * i + c];
                        if (cc > 0) {
                            const int pos = T - 1 - (s + kk);
                            nbox[pos] = child_box(b, c);
                            nncnt[pos] = cc;
                            sCC[4 * i + c] = pos;
                            locExpand += cc > 1;
                            kk++;
                        }
                    }
                } else {
                    const int pos = T + kbase;
                    nbox[pos] = box[i];
                    nncnt[pos] = ncnt[i];
                    sKeep[i] = pos;
                    kbase++;
                }
            }
        }
        if (locExpand) atomicAdd(&sRed[2], locExpand);
        __syncthreads();
        const int nToExpand = sRed[2];

        // 6. re-label the points
        for (int p0 = tid; p0 < cL; p0 += 4 * NT) {
            uint32_t cw4[4];
            int nd4[4];
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int p = p0 + k * NT;
                cw4[k] = p < cL ? C[p] : 0u;
                nd4[k] = p < cL ? (int)nodeOf[p] : 0;
            }
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const int p = p0 + k * NT;
                const uint32_t cw = cw4[k];
                if (p < cL && PT_WEIGHT(cw)) {
                    const int i = nd4[k];
                    nodeOf[p] = (uint16_t)(sSplit[i] ? sCC[4 * i + child_code(box[i], cand_x(cw), cand_y(cw))] : sKeep[i]);
                }
            }
        }
        __syncthreads();
        n = newSize;
        cur ^= 1;

        // 7. termination, :358-362 / :424-425
        if (n >= N || n == prevSize) break;
        if (!sortedMode && n + 3 * nToExpand > N) sortedMode = true;
        if (round == kQtMaxRounds - 1 && tid == 0) atomicOr(&cnt[kCntStatus], (uint32_t)kFlagRoundLimit);
    }

    if (overflow) {
        if (tid == 0) { cnt[kCntKp] = 0; atomicOr(&cnt[kCntStatus], (uint32_t)kFlagNodeOverflow); }
        return;
    }

    // ---- best point per node: max response, first (smallest raster key) wins ties, :515-527 ----
    int* bestResp = sCC;
    uint32_t* bestKey = reinterpret_cast<uint32_t*>(sCC + NC);
    for (int i = tid; i < n; i += NT) { bestResp[i] = 0; bestKey[i] = 0xFFFFFFFFu; }
    __syncthreads();
    for (int p = tid; p < cL; p += NT) {
        const uint32_t cw = C[p];
        if (PT_WEIGHT(cw)) atomicMax(&bestResp[nodeOf[p]], cand_score(cw));
    }
    __syncthreads();
    for (int p = tid; p < cL; p += NT) {
        const uint32_t cw = C[p];
        if (PT_WEIGHT(cw)) {
            const int i = nodeOf[p];
            if (cand_score(cw) == bestResp[i]) atomicMin(&bestKey[i], cand_key(cw));
        }
    }
    __syncthreads();
    for (int i = tid; i < n; i += NT) out[i] = ((uint32_t)bestResp[i] << 24) | bestKey[i];
    if (tid == 0) cnt[kCntKp] = (uint32_t)n;
#undef PT_WEIGHT
}

// node-table capacity the launcher picks for a given per-level maximum (0 == unsupported)
int quadtree_node_capacity(int maxNodeCap)
{
    const int caps[] = {512, 2048, 8192, 32768, 65536};
    for (int c : caps)
        if (maxNodeCap <= c && maxNodeCap <= 65535) return c;  // node ids are u16
    return 0;
}

// bytes of HBM scratch one (frame, level) block needs (0 for the LDS variants)
size_t quadtree_scratch_bytes_per_block(int maxNodeCap)
{
    const int nc = quadtree_node_capacity(maxNodeCap);
    return nc > 2048 ? qt_scratch_bytes(nc) : 0;
}

void launch_quadtree(hipStream_t s, int frames, int nLevels, int maxNodeCap, const PipelineDesc* dP,
                     const uint32_t* cand, uint16_t* nodeOf, uint32_t* counters, uint32_t* lvlKp,
                     const uint8_t* gray0, size_t gray0FrameStride, int gray0Pitch, const uint8_t* ws,
                     const uint16_t* tileRows, uint8_t* scratch)
{
    dim3 grid(frames, nLevels);
    // A call on one or a few frames is a handful of blocks on an empty chip and its time is the largest level's block:
    // 1024 threads walk every candidate loop in half the trips (0.5 candidates-per-thread passes dominate a level-0 block).
    if ((long long)frames * nLevels <= 64 && quadtree_node_capacity(maxNodeCap) <= 2048) {
        dim3 blockS(kQtThreadsSmall);
        if (quadtree_node_capacity(maxNodeCap) <= 1024)
            hipLaunchKernelGGL((quadtree_kernel<1024, false, kQtThreadsSmall>), grid, blockS, 0, s, dP, cand, nodeOf, counters, lvlKp, gray0,
                               gray0FrameStride, gray0Pitch, ws, tileRows, scratch);
        else
            hipLaunchKernelGGL((quadtree_kernel<2048, false, kQtThreadsSmall>), grid, blockS, 0, s, dP, cand, nodeOf, counters, lvlKp, gray0,
                               gray0FrameStride, gray0Pitch, ws, tileRows, scratch);
        return;
    }
    dim3 block(kQtThreads);
#define ORBFE_QT(NCV, GLB)                                                                                        \
    hipLaunchKernelGGL((quadtree_kernel<NCV, GLB>), grid, block, 0, s, dP, cand, nodeOf, counters, lvlKp, gray0, \
                       gray0FrameStride, gray0Pitch, ws, tileRows, scratch)
    switch (quadtree_node_capacity(maxNodeCap)) {
    case 512: ORBFE_QT(512, false); break;
    case 2048: ORBFE_QT(2048, false); break;
    case 8192: ORBFE_QT(8192, true); break;
    case 32768: ORBFE_QT(32768, true); break;
    default: ORBFE_QT(65536, true); break;
    }
#undef ORBFE_QT
}

}  // namespace orbfe
