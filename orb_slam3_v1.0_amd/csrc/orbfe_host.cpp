// orbfe_host.cpp -- host orchestration + C ABI (include/orbfe.h) of the MI355X ORB front-end.
//
// Mirrors the host side of the reference extractor (src/ORBextractor.cc): constructor tables
// (:82-149), AllocatePyramid (:587-605), ComputePyramid (:607-623), ComputeKeyPointsOctTree
// (:433-541) and extractFeatures (:543-585) -- but as ONE ordered chain of asynchronous kernel
// launches on one HIP stream, with no host synchronisation between stages (the reference syncs
// >= 8 times per level) and no per-frame allocation.
#include <atomic>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <mutex>
#include <new>
#include <condition_variable>
#include <functional>
#include <string>
#include <thread>
#include <vector>

#include "launch.h"
#include "match.h"
#include "match_common.h"
#include "match_proj.h"
#include "keyframe.h"
#include "prep.h"
#include "vocab.h"

using namespace orbfe;

namespace {

constexpr int kEventSets = 128;

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

const char* kStageNames[ORBFE_NUM_STAGES] = {"pyramid_resize", "fast_nms_blur",
                                             "quadtree", "orient_brief", "total"};

}  // namespace

struct orbfe_map;
struct orbfe_stream;

struct orbfe_handle {
    orbfe_params prm{};
    int device = 0;
    int nLevels = 0;
    double scaleFactorD = 0.0;  // the reference stores scaleFactor as double (ORBextractor.h:108)
    float sf[kMaxLevels]{}, inv[kMaxLevels]{}, sig2[kMaxLevels]{}, invSig2[kMaxLevels]{};
    PipelineDesc P{};
    PipelineDesc* dP = nullptr;
    int maxBatch = 1;
    int maxNodeCap = 0;

    uint8_t* ws = nullptr;          // pyramid + blurred pyramid, all frames
    size_t wsBytes = 0;
    uint32_t* dCand = nullptr;      // [level][frame][candCap]
    uint16_t* dNodeOf = nullptr;
    size_t candWordsPerBatch = 0;
    uint32_t* dCounters = nullptr;  // [frame][level][kCntWords]
    uint32_t* dLvlKp = nullptr;     // [frame][kpCapFrame]
    uint16_t* dTileRows = nullptr;  // [frame][FAST tile][32] pre-NMS corner counts per tile row: low pass | high pass << 8
    uint8_t* dQtScratch = nullptr;  // node tables of the large-N quadtree variant, [level][frame] slabs
    uint32_t* dTabs = nullptr;      // resize tables
    uint32_t* dTileInfo = nullptr;  // FAST tile -> (level, tile column, tile row)
    bool pyrFits[kMaxLevels]{};     // level is produced by the row-streaming pyramid kernel (else: the table-driven tile kernel)
    float* dSf = nullptr;           // mvScaleFactor on the device (batched matcher)

    // staging for the host-pointer API
    uint8_t* dIn = nullptr;
    int dInPitch = 0;
    uint8_t* hIn = nullptr;         // pinned
    // results of the host-pointer API: ONE device block [n | status | per-level | keypoints | descriptors] with a
    // pinned mirror of the same layout, so a full batch comes back in a single D2H copy
    uint8_t* dOutBlock = nullptr;
    uint8_t* hOutBlock = nullptr;   // pinned
    size_t outBlockBytes = 0, offStatus = 0, offPer = 0, offKp = 0, offDesc = 0;
    orbfe_keypoint* dKp = nullptr;
    uint8_t* dDesc = nullptr;
    int* dN = nullptr;
    int* dStatus = nullptr;
    int* dPer = nullptr;
    orbfe_keypoint* hKp = nullptr;
    uint8_t* hDesc = nullptr;
    int* hN = nullptr;
    int* hStatus = nullptr;
    int* hPer = nullptr;
    // the whole host-API call (H2D, kernel chain, D2H) as a captured hipGraph per batch size (latency path)
    std::map<int, hipGraphExec_t> graphs;
    bool useGraph = true;
    int graphCaptures = 0, captureFailures = 0;  // orbfe_debug_graph_stats (totals since create / the last orbfe_set_graph_capture(1))
    int captureFailStreak = 0;                   // CONSECUTIVE failed captures: a successful one resets it

    hipStream_t stream = nullptr;
    bool timing = false;
    hipEvent_t ev[kEventSets][ORBFE_NUM_STAGES]{};
    int evHead = 0, evCount = 0;

    // last call (for the pyramid / candidate getters)
    const uint8_t* lastGray = nullptr;
    size_t lastStride = 0;
    int lastPitch = 0, lastBatch = 0;

    MatchScratch match;

    // orbfe_track_frame: the fused per-frame chain.  Its captured graphs hold the addresses of these blocks and of this
    // matcher arena, so nothing else uses them and a regrow drops every graph.
    struct TrackKey {
        int Mb, inPitch, gridCols, gridRows, farPoints;
        float minX, minY, invW, invH, th, nnRatio, thFar;
        const void* map;  // null: explicit points in the block; else the resident map the ids refer to (its addresses are in the graph)
    };
    struct TrackGraph {
        TrackKey key;
        hipGraphExec_t exec;
    };
    MatchScratch trackMatch;
    uint8_t* dTrkIn = nullptr;   // [image | frustum | points (Mb) | descriptors (Mb)]
    uint8_t* hTrkIn = nullptr;   // pinned mirror
    uint8_t* dTrkOut = nullptr;  // [n, status, n_matches | per-level | keypoints | descriptors | match | map-point records (Mb) | xr (Mb)]
    uint8_t* hTrkOut = nullptr;  // pinned mirror
    int trkCapM = 0;
    std::vector<TrackGraph> trackGraphs;
    // orbfe_track_reference_keyframe: extract -> vocabulary descent -> SearchByBoW against a resident key frame.  The key
    // frame is named by a record inside the input block, so a graph depends only on what is in RefKey.
    struct RefKey {
        int inPitch, levelsup, checkOri;
        float nnRatio;
        unsigned long long vocabSerial;
    };
    struct RefGraph {
        RefKey key;
        hipGraphExec_t exec;
    };
    uint8_t* dRefIn = nullptr;   // [image | key-frame record | key-frame flags (refCapFlags)]
    uint8_t* hRefIn = nullptr;
    uint8_t* dRefOut = nullptr;  // [n, status, n_matches | per-level | keypoints | descriptors | match | (word, node) | leaf | bins]
    uint8_t* hRefOut = nullptr;
    int refCapFlags = 0;
    std::vector<RefGraph> refGraphs;
    // orbfe_track_initialization: extract -> SearchForInitialization(resident initial frame, frame).  A graph holds the
    // addresses of the initial frame it was captured for (named by its serial) and of these blocks.
    struct IniKey {
        int inPitch, gridCols, gridRows, window, checkOri;
        float minX, minY, invW, invH, nnRatio;
        unsigned long long frameSerial;
    };
    struct IniGraph {
        IniKey key;
        hipGraphExec_t exec;
    };
    uint8_t* dIniIn = nullptr;    // [image]
    uint8_t* hIniIn = nullptr;
    uint8_t* dIniOut = nullptr;   // [n, status, n_matches | per-level | keypoints | descriptors | matches12 (iniCapN1)] + matcher scratch
    uint8_t* hIniOut = nullptr;
    int iniCapN1 = -1;
    size_t iniScratchBytes = 0;
    std::vector<IniGraph> iniGraphs;
    std::mutex mu;
    std::string err;
    // objects that keep a pointer to this handle: orbfe_destroy releases their device memory and orphans them (h = null), so
    // that destroying them afterwards -- a binding's finalisers run in any order -- only frees the empty shell
    std::vector<orbfe_map*> maps;
    std::vector<orbfe_stream*> rings;

    // Cross-stream ordering of the scratch the handle owns: an event recorded behind the last enqueue that used the
    // extraction scratch (pyramid, candidates, counters, level keypoints) / the matcher arena, and the stream it ran on.
    // The next enqueue on a different stream waits for it first; the getters wait for it on the host.
    hipEvent_t evExtract = nullptr, evMatch = nullptr;
    hipStream_t extractStream = nullptr, matchStream = nullptr;
    bool extractUsed = false, matchUsed = false;
};

namespace {

int fail_hip(orbfe_handle* h, hipError_t e, const char* what, int line)
{
    char buf[256];
    snprintf(buf, sizeof buf, "%s failed at orbfe_host.cpp:%d: %s", what, line, hipGetErrorString(e));
    if (h) h->err = buf;
    return ORBFE_ERR_HIP;
}

#define HIPCHK(h, call)                                                   \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) return fail_hip((h), e_, #call, __LINE__);  \
    } while (0)

// scratch hand-over between streams (see the handle): call with h->mu held
int scratch_acquire(orbfe_handle* h, bool used, hipStream_t last, hipEvent_t ev, hipStream_t s)
{
    if (used && last != s) HIPCHK(h, hipStreamWaitEvent(s, ev, 0));
    return ORBFE_OK;
}

int extract_scratch_release(orbfe_handle* h, hipStream_t s)
{
    HIPCHK(h, hipEventRecord(h->evExtract, s));
    h->extractStream = s;
    h->extractUsed = true;
    return ORBFE_OK;
}

// RAII pair around a matcher launch: waits for the arena's previous user on another stream, records behind this one
struct MatchScope {
    orbfe_handle* h;
    hipStream_t s;
    int rc;
    MatchScope(orbfe_handle* h_, hipStream_t s_) : h(h_), s(s_), rc(scratch_acquire(h_, h_->matchUsed, h_->matchStream, h_->evMatch, s_)) {}
    ~MatchScope()
    {
        if (hipEventRecord(h->evMatch, s) == hipSuccess) {
            h->matchStream = s;
            h->matchUsed = true;
        }
    }
};

int cv_round_f(float v) { return (int)lrintf(v); }

// resize tables of SPEC DECISION S1: entry = x1 | (Q11 weight << 16)
void fill_resize_table(std::vector<uint32_t>& t, size_t off, int srcN, int dstN)
{
    for (int x = 0; x < dstN; x++) {
        const int64_t sx = (int64_t)x * srcN;
        const int x1 = (int)(sx / dstN);
        const int fx = (int)(sx % dstN);
        const int wx = (int)(((int64_t)fx * 2048 + dstN / 2) / dstN);
        t[off + x] = (uint32_t)x1 | ((uint32_t)wx << 16);
    }
    // the kernel evaluates whole groups of four: pad with the last entry (its results are never stored)
    for (size_t x = (size_t)dstN; x < align_up((size_t)dstN, 4); x++) t[off + x] = t[off + dstN - 1];
}

void orphan_children(orbfe_handle* h);  // defined behind orbfe_map / orbfe_stream

void destroy_impl(orbfe_handle* h)
{
    if (!h) return;
    (void)hipSetDevice(h->device);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    orphan_children(h);
    for (auto& set : h->ev)
        for (auto& e : set)
            if (e) (void)hipEventDestroy(e);
    if (h->evExtract) (void)hipEventDestroy(h->evExtract);
    if (h->evMatch) (void)hipEventDestroy(h->evMatch);
    h->match.busy = nullptr;
    match_scratch_free(h->match);
    for (auto& g : h->trackGraphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    match_scratch_free(h->trackMatch);
    for (auto& g : h->refGraphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    for (auto& g : h->iniGraphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    if (h->dIniIn) (void)hipFree(h->dIniIn);
    if (h->dIniOut) (void)hipFree(h->dIniOut);
    if (h->hIniIn) (void)hipHostFree(h->hIniIn);
    if (h->hIniOut) (void)hipHostFree(h->hIniOut);
    if (h->dRefIn) (void)hipFree(h->dRefIn);
    if (h->dRefOut) (void)hipFree(h->dRefOut);
    if (h->hRefIn) (void)hipHostFree(h->hRefIn);
    if (h->hRefOut) (void)hipHostFree(h->hRefOut);
    if (h->dTrkIn) (void)hipFree(h->dTrkIn);
    if (h->dTrkOut) (void)hipFree(h->dTrkOut);
    if (h->hTrkIn) (void)hipHostFree(h->hTrkIn);
    if (h->hTrkOut) (void)hipHostFree(h->hTrkOut);
    for (auto& g : h->graphs)
        if (g.second) (void)hipGraphExecDestroy(g.second);
    void* dptrs[] = {h->dP, h->ws, h->dCand, h->dNodeOf, h->dCounters, h->dLvlKp, h->dTileRows, h->dQtScratch, h->dTabs,
                     h->dTileInfo, h->dSf, h->dIn, h->dOutBlock};
    for (void* p : dptrs)
        if (p) (void)hipFree(p);
    void* hptrs[] = {h->hIn, h->hOutBlock};
    for (void* p : hptrs)
        if (p) (void)hipHostFree(p);
    if (h->stream) (void)hipStreamDestroy(h->stream);
    delete h;
}

}  // namespace

extern "C" {

const char* orbfe_version(void) { return "orbfe 0.1 (gfx950, HIP)"; }

const char* orbfe_status_string(int s)
{
    switch (s) {
    case ORBFE_OK: return "ok";
    case ORBFE_ERR_INVALID_ARG: return "invalid argument";
    case ORBFE_ERR_UNSUPPORTED: return "unsupported configuration";
    case ORBFE_ERR_NO_DEVICE: return "no usable gfx950 device";
    case ORBFE_ERR_HIP: return "HIP runtime error";
    case ORBFE_ERR_OUT_OF_MEMORY: return "out of memory";
    case ORBFE_ERR_INTERNAL: return "internal device-side guard tripped";
    case ORBFE_ERR_BUSY: return "every slot of the stream ring is in flight";
    default: return "unknown status";
    }
}

const char* orbfe_last_error(const orbfe_handle* h) { return h ? h->err.c_str() : ""; }
const char* orbfe_stage_name(int s) { return (s >= 0 && s < ORBFE_NUM_STAGES) ? kStageNames[s] : ""; }

int orbfe_create(const orbfe_params* p, orbfe_handle** out)
{
    if (!p || !out) return ORBFE_ERR_INVALID_ARG;
    *out = nullptr;
    if (p->n_levels < 1 || p->n_levels > kMaxLevels || p->image_width < 16 || p->image_height < 16 ||
        p->n_features < 0 || p->n_fast_features < 1 || p->max_batch < 1 || !(p->scale_factor > 1.0f))
        return ORBFE_ERR_INVALID_ARG;
    // one fused FAST pass serves both thresholds only when iniThFAST >= minThFAST >= 1
    if (p->min_th_fast < 1 || p->ini_th_fast < p->min_th_fast || p->ini_th_fast > 254)
        return ORBFE_ERR_UNSUPPORTED;
    if (p->image_width > (int)kCoordMask || p->image_height > (int)kCoordMask) return ORBFE_ERR_UNSUPPORTED;

    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0 || p->device_id < 0 || p->device_id >= ndev)
        return ORBFE_ERR_NO_DEVICE;
    if (hipSetDevice(p->device_id) != hipSuccess) return ORBFE_ERR_NO_DEVICE;
    hipDeviceProp_t prop;
    if (hipGetDeviceProperties(&prop, p->device_id) != hipSuccess) return ORBFE_ERR_NO_DEVICE;
    if (strncmp(prop.gcnArchName, "gfx950", 6) != 0) return ORBFE_ERR_NO_DEVICE;  // code object is gfx950 only

    orbfe_handle* h = new (std::nothrow) orbfe_handle();
    if (!h) return ORBFE_ERR_OUT_OF_MEMORY;
    h->prm = *p;
    h->device = p->device_id;
    h->nLevels = p->n_levels;
    h->maxBatch = p->max_batch;
    const int nL = p->n_levels;

    // ---- scale tables, src/ORBextractor.cc:92-108 ----
    h->scaleFactorD = p->scale_factor;
    h->sf[0] = 1.0f;
    h->sig2[0] = 1.0f;
    for (int i = 1; i < nL; i++) {
        h->sf[i] = (float)(h->sf[i - 1] * h->scaleFactorD);
        h->sig2[i] = h->sf[i] * h->sf[i];
    }
    for (int i = 0; i < nL; i++) {
        h->inv[i] = 1.0f / h->sf[i];
        h->invSig2[i] = 1.0f / h->sig2[i];
    }
    // ---- features per level, :112-124 ----
    int fpl[kMaxLevels];
    {
        const float factor = (float)(1.0f / h->scaleFactorD);
        float nDesired = (float)p->n_features * (1 - factor) / (1 - (float)std::pow((double)factor, (double)nL));
        int sum = 0;
        for (int l = 0; l < nL - 1; l++) {
            fpl[l] = cv_round_f(nDesired);
            sum += fpl[l];
            nDesired *= factor;
        }
        fpl[nL - 1] = std::max(p->n_features - sum, 0);
    }

    // ---- level geometry (AllocatePyramid :587-605) + workspace layout ----
    PipelineDesc& P = h->P;
    memset(&P, 0, sizeof P);
    P.nLevels = nL;
    P.nFast = p->n_fast_features;
    P.iniTh = p->ini_th_fast;
    P.minTh = p->min_th_fast;
    size_t wsOff = 0, candOff = 0, tabOff = 0;
    int tileBase = 0, kpBase = 0;
    const size_t B = (size_t)p->max_batch;
    int status = ORBFE_OK;
    for (int l = 0; l < nL; l++) {
        LevelDesc& L = P.lv[l];
        if (l == 0) {
            L.w = p->image_width;
            L.h = p->image_height;
        } else {
            L.w = cv_round_f(h->inv[l] * (float)p->image_width);
            L.h = cv_round_f(h->inv[l] * (float)p->image_height);
        }
        if (L.w < 16 || L.h < 16) { status = ORBFE_ERR_UNSUPPORTED; break; }
        L.pitch = (int)align_up((size_t)L.w, kPitchAlign);
        L.nFeatures = fpl[l];
        L.nIni = (int)roundf((float)L.w / (float)L.h);  // :231
        if (L.nIni < 1) { status = ORBFE_ERR_UNSUPPORTED; break; }  // the reference divides by zero here
        L.hX = (float)L.w / (float)L.nIni;                // :233
        L.nodeCap = std::max(L.nFeatures + 3, 4 * L.nIni);
        L.candCap = ((L.w + 1) / 2) * ((L.h + 1) / 2);
        fast_tiles_for(L.w, L.h, &L.tilesX, &L.tilesY);
        L.tileBase = tileBase;
        tileBase += L.tilesX * L.tilesY;
        L.kpBase = kpBase;
        kpBase += L.nodeCap;
        L.invScale = h->inv[l];
        L.scaledPatch = (int)(kPatch * h->inv[l]);         // :511
        const size_t frameBytes = align_up((size_t)L.pitch * L.h, 256);
        L.imgFrameStride = frameBytes;
        L.blurFrameStride = frameBytes;
        if (l > 0) {
            L.imgOff = wsOff;
            wsOff += frameBytes * B;
        }
        L.blurOff = wsOff;
        wsOff += frameBytes * B;
        L.candOff = candOff;
        candOff += (size_t)L.candCap * B;
        tabOff = align_up(tabOff, 4);  // uint4 table loads in resize_kernel
        L.xtabOff = tabOff;
        tabOff += align_up((size_t)L.w, 4);
        L.ytabOff = tabOff;
        tabOff += align_up((size_t)L.h, 4);
        h->maxNodeCap = std::max(h->maxNodeCap, L.nodeCap);
    }
    if (status == ORBFE_OK && quadtree_node_capacity(h->maxNodeCap) == 0) status = ORBFE_ERR_UNSUPPORTED;  // > 65535 nodes/level
    if (status != ORBFE_OK) {
        delete h;
        return status;
    }
    P.kpCapFrame = kpBase;
    P.totalTiles = tileBase;
    for (int l = 0; l < 8; l++) P.tileBaseTab[l] = l < nL ? P.lv[l].tileBase : 0x7fffffff;
    h->wsBytes = wsOff;
    h->candWordsPerBatch = candOff;

    std::vector<uint32_t> tabs(tabOff ? tabOff : 1, 0);
    for (int l = 1; l < nL; l++) {
        fill_resize_table(tabs, P.lv[l].xtabOff, P.lv[l - 1].w, P.lv[l].w);
        fill_resize_table(tabs, P.lv[l].ytabOff, P.lv[l - 1].h, P.lv[l].h);
    }

    for (int l = 1; l < nL; l++)
        if (pyramid_level_fits(tabs.data() + P.lv[l].xtabOff, tabs.data() + P.lv[l].ytabOff, P.lv[l - 1].w, P.lv[l - 1].h, P.lv[l].w,
                               P.lv[l].h))
            h->pyrFits[l] = true;

#define CREATE_CHK(call)                                                  \
    do {                                                                  \
        hipError_t e_ = (call);                                           \
        if (e_ != hipSuccess) {                                           \
            int rc_ = e_ == hipErrorOutOfMemory ? ORBFE_ERR_OUT_OF_MEMORY : ORBFE_ERR_HIP; \
            destroy_impl(h);                                              \
            return rc_;                                                   \
        }                                                                 \
    } while (0)

    CREATE_CHK(hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking));
    CREATE_CHK(hipMalloc(&h->dP, sizeof(PipelineDesc)));
    CREATE_CHK(hipMalloc(&h->ws, h->wsBytes + 256));  // + slack: the pyramid kernel's 12-byte row windows may end past a level's last row
    CREATE_CHK(hipMalloc(&h->dCand, candOff * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc(&h->dNodeOf, candOff * sizeof(uint16_t)));
    CREATE_CHK(hipMalloc(&h->dCounters, B * nL * kCntWords * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc(&h->dLvlKp, B * (size_t)P.kpCapFrame * sizeof(uint32_t)));
    CREATE_CHK(hipMalloc(&h->dTileRows, B * (size_t)P.totalTiles * 32 * sizeof(uint16_t)));
    if (quadtree_scratch_bytes_per_block(h->maxNodeCap))
        CREATE_CHK(hipMalloc(&h->dQtScratch, quadtree_scratch_bytes_per_block(h->maxNodeCap) * B * nL));
    CREATE_CHK(hipMalloc(&h->dTabs, tabs.size() * sizeof(uint32_t)));
    {
        std::vector<uint32_t> info((size_t)P.totalTiles);
        for (int l = 0; l < nL; l++)
            for (int ty = 0; ty < P.lv[l].tilesY; ty++)
                for (int tx = 0; tx < P.lv[l].tilesX; tx++) info[P.lv[l].tileBase + ty * P.lv[l].tilesX + tx] = fast_tile_info(l, tx, ty);
        CREATE_CHK(hipMalloc(&h->dTileInfo, info.size() * sizeof(uint32_t)));
        CREATE_CHK(copy_sync(h->dTileInfo, info.data(), info.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));
    }
    CREATE_CHK(hipMalloc(&h->dSf, kMaxLevels * sizeof(float)));
    CREATE_CHK(copy_sync(h->dSf, h->sf, kMaxLevels * sizeof(float), hipMemcpyHostToDevice, h->stream));
    CREATE_CHK(copy_sync(h->dP, &P, sizeof P, hipMemcpyHostToDevice, h->stream));
    CREATE_CHK(copy_sync(h->dTabs, tabs.data(), tabs.size() * sizeof(uint32_t), hipMemcpyHostToDevice, h->stream));

    // staging for the host-pointer entry points
    h->dInPitch = (int)align_up((size_t)p->image_width, kPitchAlign);
    const size_t inFrame = (size_t)h->dInPitch * p->image_height;
    const size_t cap = (size_t)P.kpCapFrame;
    CREATE_CHK(hipMalloc(&h->dIn, inFrame * B));
    CREATE_CHK(hipHostMalloc(&h->hIn, inFrame * B));
    {
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
        take(B * sizeof(int));  // n at offset 0
        h->offStatus = take(B * sizeof(int));
        h->offPer = take(B * nL * sizeof(int));
        h->offKp = take(B * cap * sizeof(orbfe_keypoint));
        h->offDesc = take(B * cap * ORBFE_DESC_BYTES);
        h->outBlockBytes = off;
    }
    CREATE_CHK(hipMalloc(&h->dOutBlock, h->outBlockBytes));
    CREATE_CHK(hipHostMalloc(&h->hOutBlock, h->outBlockBytes));
    h->dN = reinterpret_cast<int*>(h->dOutBlock);
    h->dStatus = reinterpret_cast<int*>(h->dOutBlock + h->offStatus);
    h->dPer = reinterpret_cast<int*>(h->dOutBlock + h->offPer);
    h->dKp = reinterpret_cast<orbfe_keypoint*>(h->dOutBlock + h->offKp);
    h->dDesc = h->dOutBlock + h->offDesc;
    h->hN = reinterpret_cast<int*>(h->hOutBlock);
    h->hStatus = reinterpret_cast<int*>(h->hOutBlock + h->offStatus);
    h->hPer = reinterpret_cast<int*>(h->hOutBlock + h->offPer);
    h->hKp = reinterpret_cast<orbfe_keypoint*>(h->hOutBlock + h->offKp);
    h->hDesc = h->hOutBlock + h->offDesc;
#ifdef ORBFE_DIAG
    h->useGraph = getenv("ORBFE_NO_GRAPH") == nullptr;  // liborbfe_diag.so only: plain launches, e.g. under a debugger
#endif
    CREATE_CHK(hipEventCreateWithFlags(&h->evExtract, hipEventDisableTiming));
    CREATE_CHK(hipEventCreateWithFlags(&h->evMatch, hipEventDisableTiming));
    h->match.busy = h->evMatch;  // the arena is not regrown while a launch still uses it
    for (auto& set : h->ev)
        for (auto& e : set) CREATE_CHK(hipEventCreate(&e));
#undef CREATE_CHK
    *out = h;
    return ORBFE_OK;
}

void orbfe_destroy(orbfe_handle* h) { destroy_impl(h); }

int orbfe_get_levels(const orbfe_handle* h) { return h ? h->nLevels : 0; }
float orbfe_get_scale_factor(const orbfe_handle* h) { return h ? (float)h->scaleFactorD : 0.f; }

int orbfe_get_scale_tables(const orbfe_handle* h, float* sf, float* inv, float* s2, float* is2)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < h->nLevels; i++) {
        if (sf) sf[i] = h->sf[i];
        if (inv) inv[i] = h->inv[i];
        if (s2) s2[i] = h->sig2[i];
        if (is2) is2[i] = h->invSig2[i];
    }
    return ORBFE_OK;
}

int orbfe_get_level_info(const orbfe_handle* h, int* fpl, int* lw, int* lh)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < h->nLevels; i++) {
        if (fpl) fpl[i] = h->P.lv[i].nFeatures;
        if (lw) lw[i] = h->P.lv[i].w;
        if (lh) lh[i] = h->P.lv[i].h;
    }
    return ORBFE_OK;
}

int orbfe_max_keypoints(const orbfe_handle* h) { return h ? h->P.kpCapFrame : 0; }

int orbfe_set_stage_timing(orbfe_handle* h, int enabled)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    h->timing = enabled != 0;
    h->evHead = h->evCount = 0;
    return ORBFE_OK;
}

int orbfe_get_stage_ms(orbfe_handle* h, float ms[ORBFE_NUM_STAGES], int* n_calls)
{
    if (!h || !ms || !n_calls) return ORBFE_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    for (int s = 0; s < ORBFE_NUM_STAGES; s++) ms[s] = 0.f;
    const int n = h->evCount < kEventSets ? h->evCount : kEventSets;
    for (int i = 0; i < n; i++) {
        hipEvent_t* e = h->ev[i];
        HIPCHK(h, hipEventSynchronize(e[ORBFE_NUM_STAGES - 1]));
        for (int s = 0; s + 1 < ORBFE_NUM_STAGES; s++) {
            float t = 0.f;
            HIPCHK(h, hipEventElapsedTime(&t, e[s], e[s + 1]));
            ms[s] += t;
        }
        float t = 0.f;
        HIPCHK(h, hipEventElapsedTime(&t, e[0], e[ORBFE_NUM_STAGES - 1]));
        ms[ORBFE_NUM_STAGES - 1] += t;
    }
    *n_calls = n;
    return ORBFE_OK;
}

static int extract_chain(orbfe_handle* h, const uint8_t* d_gray, size_t frame_stride, int pitch, int batch,
                         orbfe_keypoint* d_kp, uint8_t* d_desc, int* d_n, int* d_per, int* d_status, void* stream_);

int orbfe_extract_batch_device(orbfe_handle* h, const uint8_t* d_gray, size_t frame_stride, int pitch,
                               int batch, orbfe_keypoint* d_kp, uint8_t* d_desc, int* d_n, int* d_per,
                               void* stream_)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    return extract_chain(h, d_gray, frame_stride, pitch, batch, d_kp, d_desc, d_n, d_per, nullptr, stream_);
}

}  // extern "C"

static int extract_chain(orbfe_handle* h, const uint8_t* d_gray, size_t frame_stride, int pitch, int batch,
                         orbfe_keypoint* d_kp, uint8_t* d_desc, int* d_n, int* d_per, int* d_status, void* stream_)
{
    if (!h || !d_gray || !d_kp || !d_desc || !d_n) return ORBFE_ERR_INVALID_ARG;
    if (batch < 1 || batch > h->maxBatch || pitch < h->prm.image_width || pitch >= (1 << 24)) return ORBFE_ERR_INVALID_ARG;
    const int aligned4 = ((reinterpret_cast<uintptr_t>(d_gray) | (uintptr_t)pitch | (uintptr_t)frame_stride) & 3u) == 0;
    // frame extent the kernels read (orbfe.h, input contract): whole dwords on the aligned path
    const size_t rowEnd = aligned4 ? (size_t)((h->prm.image_width + 3) & ~3) : (size_t)h->prm.image_width;
    if (batch > 1 && frame_stride < (size_t)pitch * (h->prm.image_height - 1) + rowEnd) return ORBFE_ERR_INVALID_ARG;
    // the kernels address a frame with 32-bit byte offsets and send dropped lanes to offset 0x7ffffff0 of a buffer
    // descriptor: a frame must end below that (480 rows at an 8 MiB pitch would wrap the row offsets)
    if ((size_t)pitch * (h->prm.image_height - 1) + (size_t)((h->prm.image_width + 3) & ~3) >= (size_t)0x7ffffff0u)
        return ORBFE_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = stream_ ? (hipStream_t)stream_ : h->stream;
    // while the chain is being captured into a hipGraph the hand-over events stay outside it: the replay path waits
    // and records around hipGraphLaunch
    hipStreamCaptureStatus capturing = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &capturing) != hipSuccess) (void)hipGetLastError();
    if (capturing == hipStreamCaptureStatusNone) {
        const int rc = scratch_acquire(h, h->extractUsed, h->extractStream, h->evExtract, s);
        if (rc != ORBFE_OK) return rc;
    }
    const PipelineDesc& P = h->P;
    const int nL = P.nLevels;
    hipEvent_t* ev = nullptr;
    if (h->timing) {
        ev = h->ev[h->evHead];
        h->evHead = (h->evHead + 1) % kEventSets;
        h->evCount++;
    }
    const bool chainPyr = (long long)batch * nL <= 64 && !ev && nL > 1;  // one to eight frames: see below
    if (!chainPyr) HIPCHK(h, hipMemsetAsync(h->dCounters, 0, (size_t)batch * nL * kCntWords * sizeof(uint32_t), s));
    if (ev) HIPCHK(h, hipEventRecord(ev[0], s));
    // ComputePyramid (:607-623): level l from the UNBLURRED level l-1
    // (one to eight frames: two or three levels per launch -- a dependent launch costs more than the small levels' work)
    for (int l = 1; chainPyr && l < nL;) {
        const int left = nL - l;
        const int depth = left >= 5 ? 2 : std::min(left, 3);  // 7 levels to build: 2 + 2 + 3
        launch_pyramid_chain(s, batch, h->dP, P, l - 1, depth, d_gray, frame_stride, pitch, h->ws, h->dTabs,
                             l == 1 ? h->dCounters : nullptr, nL * kCntWords);  // the first launch clears the level counters
        l += depth;
    }
    for (int l = chainPyr ? nL : 1; l < nL; l++) {
        const LevelDesc& S = P.lv[l - 1];
        const LevelDesc& D = P.lv[l];
        if (h->pyrFits[l] && (l > 1 || aligned4)) {
            launch_pyramid_level(s, batch, h->dP, l, D.w, D.h, d_gray, frame_stride, pitch, h->ws, h->dTabs);
            continue;
        }
        const uint8_t* src = l == 1 ? d_gray : h->ws + S.imgOff;
        const size_t sstride = l == 1 ? frame_stride : S.imgFrameStride;
        const int spitch = l == 1 ? pitch : S.pitch;
        launch_resize(s, batch, src, sstride, S.w, S.h, spitch, l == 1 ? aligned4 : 1, h->ws + D.imgOff, D.imgFrameStride, D.w, D.h,
                      D.pitch, h->dTabs + D.xtabOff, h->dTabs + D.ytabOff);
    }
    if (ev) HIPCHK(h, hipEventRecord(ev[1], s));
    launch_fast_blur(s, batch, P.totalTiles, h->dP, h->dTileInfo, d_gray, frame_stride, pitch, aligned4, h->ws, h->dCand, h->dCounters,
                     h->dTileRows);
    if (ev) HIPCHK(h, hipEventRecord(ev[2], s));
    launch_quadtree(s, batch, nL, h->maxNodeCap, h->dP, h->dCand, h->dNodeOf, h->dCounters, h->dLvlKp, d_gray, frame_stride,
                    pitch, h->ws, h->dTileRows, h->dQtScratch);
    if (ev) HIPCHK(h, hipEventRecord(ev[3], s));
    int kpBase[kMaxLevels];
    for (int l = 0; l < nL; l++) kpBase[l] = P.lv[l].kpBase;
    launch_orient_brief(s, batch, P.kpCapFrame, h->dP, d_gray, frame_stride, pitch, h->ws, h->dCounters, h->dLvlKp,
                        d_kp, d_desc, d_n, d_per, d_status, kpBase, nL);
    if (ev) HIPCHK(h, hipEventRecord(ev[4], s));
    HIPCHK(h, hipGetLastError());
    h->lastGray = d_gray;
    h->lastStride = frame_stride;
    h->lastPitch = pitch;
    h->lastBatch = batch;
    if (capturing == hipStreamCaptureStatusNone) return extract_scratch_release(h, s);
    return ORBFE_OK;
}

extern "C" {

static int check_device_flags(orbfe_handle* h, int batch)
{
    // caller has synchronised the stream; hStatus holds the OR of the per-level guard flags of every frame
    for (int b = 0; b < batch; b++)
        if (h->hStatus[b]) {
            char buf[128];
            snprintf(buf, sizeof buf, "device guard flags 0x%x at frame %d", (unsigned)h->hStatus[b], b);
            h->err = buf;
            return ORBFE_ERR_INTERNAL;
        }
    return ORBFE_OK;
}

// kernel chain on the frames already sent to dIn, D2H of the result block (one copy for a full batch)
static int extract_host_enqueue(orbfe_handle* h, int batch, int inPitch, hipStream_t s)
{
    const int nL = h->nLevels;
    const size_t inFrame = (size_t)h->dInPitch * h->prm.image_height;
    const size_t cap = (size_t)h->P.kpCapFrame;
    const int rc = extract_chain(h, h->dIn, inFrame, inPitch, batch, h->dKp, h->dDesc, h->dN, h->dPer, h->dStatus, s);
    if (rc != ORBFE_OK) return rc;
    if (batch == h->maxBatch) {
        HIPCHK(h, hipMemcpyAsync(h->hOutBlock, h->dOutBlock, h->outBlockBytes, hipMemcpyDeviceToHost, s));
    } else {
        HIPCHK(h, hipMemcpyAsync(h->hN, h->dN, batch * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->hStatus, h->dStatus, batch * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->hPer, h->dPer, (size_t)batch * nL * sizeof(int), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->hKp, h->dKp, batch * cap * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost, s));
        HIPCHK(h, hipMemcpyAsync(h->hDesc, h->dDesc, batch * cap * ORBFE_DESC_BYTES, hipMemcpyDeviceToHost, s));
    }
    return ORBFE_OK;
}

// Graphs are captured on a THROW-AWAY stream, never on the handle's (or a caller's) own.  On this runtime (ROCm 7.2) a
// NULL-stream operation of any other thread of the process -- a plain hipMemcpy of the application -- that meets a capture in
// flight fails with hipErrorStreamCaptureImplicit AND leaves the capturing stream unusable for good, whatever the capture mode
// and although the stream is non-blocking (tools/probes/capture_invalidate.cpp): every later call on it returns
// hipErrorStreamCaptureInvalidated.  With a throw-away stream such an encounter costs the call that was capturing its graph,
// not the handle: the stream is destroyed, the call runs on plain launches, a later call captures again.  Only a handle
// whose captures fail eight times IN A ROW stops trying (a success resets the count).  (The library itself makes no NULL-stream call: copy_sync / memset_sync.)
struct CaptureStream {
    hipStream_t s = nullptr;
    bool capturing = false;
    bool begin()
    {
        if (hipStreamCreateWithFlags(&s, hipStreamNonBlocking) != hipSuccess) {
            s = nullptr;
            (void)hipGetLastError();
            return false;
        }
        if (hipStreamBeginCapture(s, hipStreamCaptureModeThreadLocal) != hipSuccess) {
            (void)hipGetLastError();
            return false;
        }
        capturing = true;
        return true;
    }
    // the captured graph, or null if the capture did not survive
    hipGraph_t end()
    {
        hipGraph_t g = nullptr;
        if (capturing) {
            capturing = false;
            if (hipStreamEndCapture(s, &g) != hipSuccess) {
                (void)hipGetLastError();
                if (g) (void)hipGraphDestroy(g);
                g = nullptr;
            }
        }
        return g;
    }
    ~CaptureStream()
    {
        if (capturing) {
            hipGraph_t g = end();
            if (g) (void)hipGraphDestroy(g);
        }
        if (s) (void)hipStreamDestroy(s);
        (void)hipGetLastError();
    }
};

static void graph_capture_failed(orbfe_handle* h)
{
    (void)hipGetLastError();
    h->captureFailures++;
    if (++h->captureFailStreak >= 8) h->useGraph = false;  // eight in a row: this handle's captures keep failing
}

static void graph_capture_succeeded(orbfe_handle* h)
{
    h->graphCaptures++;
    h->captureFailStreak = 0;  // an occasional foreign NULL-stream call over a long run never adds up to the limit
}

// extract_host_enqueue, replayed from a hipGraph when possible: every pointer behind the upload is owned by the
// handle, so the enqueue sequence (the kernels of extract_chain, result copy) is captured once per (batch, input pitch) and
// replayed with a single hipGraphLaunch
static int extract_enqueue_replay(orbfe_handle* h, int batch, int inPitch, hipStream_t s)
{
    const size_t inFrame = (size_t)h->dInPitch * h->prm.image_height;
    int rc = ORBFE_OK;
    bool viaGraph = h->useGraph && !h->timing && batch < 4096;
    if (viaGraph) {
        // every pointer behind the upload is owned by the handle, so the enqueue sequence (kernels, result
        // copy) is captured once per batch size and replayed with a single hipGraphLaunch
        const int gkey = batch | (inPitch << 12);  // batch <= 4095 frames per call in graph mode, else plain launches
        hipGraphExec_t& exec = h->graphs[gkey];
        if (!exec) {
            hipGraph_t graph = nullptr;
            {
                CaptureStream cs;
                rc = cs.begin() ? extract_host_enqueue(h, batch, inPitch, cs.s) : ORBFE_ERR_HIP;
                graph = cs.end();
            }
            if (rc != ORBFE_OK || !graph) {
                rc = ORBFE_OK;
                if (graph) (void)hipGraphDestroy(graph);
                h->graphs.erase(gkey);
                graph_capture_failed(h);
                viaGraph = false;  // plain launches below
            } else {
                const hipError_t ei = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
                (void)hipGraphDestroy(graph);
                if (ei != hipSuccess) {
                    h->graphs.erase(gkey);
                    graph_capture_failed(h);
                    viaGraph = false;
                } else {
                    graph_capture_succeeded(h);
                }
            }
        }
        if (viaGraph) {
            {
                const int rca = scratch_acquire(h, h->extractUsed, h->extractStream, h->evExtract, s);
                if (rca != ORBFE_OK) return rca;
            }
            HIPCHK(h, hipGraphLaunch(h->graphs[gkey], s));
            {
                const int rcr = extract_scratch_release(h, s);
                if (rcr != ORBFE_OK) return rcr;
            }
            h->lastGray = h->dIn;  // what extract_chain records on a plain launch (pyramid / candidate getters)
            h->lastStride = inFrame;
            h->lastPitch = inPitch;
            h->lastBatch = batch;
        }
    }
    if (!viaGraph) {
        rc = extract_host_enqueue(h, batch, inPitch, s);
        if (rc != ORBFE_OK) return rc;
    }
    return ORBFE_OK;
}

int orbfe_extract_batch(orbfe_handle* h, const uint8_t* const* grays, int pitch, int batch, orbfe_keypoint* kp_out,
                        uint8_t* desc_out, int* n_out, int* per_level)
{
    if (!h || !grays || !kp_out || !desc_out || !n_out) return ORBFE_ERR_INVALID_ARG;
    if (batch < 1 || batch > h->maxBatch || pitch < h->prm.image_width) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    const int W = h->prm.image_width, H = h->prm.image_height, nL = h->nLevels;
    const size_t inFrame = (size_t)h->dInPitch * H;
    hipStream_t s = h->stream;
    // Upload.  Pinned sources (the reference hands over cv::cuda::HostMem, include/ORBextractor.h:62) with a
    // dword-aligned pitch that fits the device staging rows are copied by the DMA engine straight from the caller's
    // buffer, keeping the caller's pitch (level 0 is read with an arbitrary pitch anyway); everything else is
    // re-pitched through the handle's pinned staging block first.
    bool direct = pitch <= h->dInPitch && (pitch & 3) == 0;
    for (int b = 0; b < batch; b++) {
        if (!grays[b]) return ORBFE_ERR_INVALID_ARG;
        if (direct) {
            hipPointerAttribute_t attr;
            if ((reinterpret_cast<uintptr_t>(grays[b]) & 3u) != 0 || hipPointerGetAttributes(&attr, grays[b]) != hipSuccess ||
                attr.type != hipMemoryTypeHost) {
                (void)hipGetLastError();
                direct = false;
            }
        }
    }
    const int inPitch = direct ? pitch : h->dInPitch;
    if (direct) {
        const size_t bytes = (size_t)pitch * (H - 1) + (size_t)W;
        for (int b = 0; b < batch; b++)
            HIPCHK(h, hipMemcpyAsync(h->dIn + b * inFrame, grays[b], bytes, hipMemcpyHostToDevice, s));
    } else {
        for (int b = 0; b < batch; b++)
            for (int y = 0; y < H; y++)
                memcpy(h->hIn + b * inFrame + (size_t)y * h->dInPitch, grays[b] + (size_t)y * pitch, (size_t)W);
        HIPCHK(h, hipMemcpyAsync(h->dIn, h->hIn, inFrame * batch, hipMemcpyHostToDevice, s));
    }
    int rc = extract_enqueue_replay(h, batch, inPitch, s);
    if (rc != ORBFE_OK) return rc;
    HIPCHK(h, hipStreamSynchronize(s));
    rc = check_device_flags(h, batch);
    if (rc != ORBFE_OK) return rc;
    const size_t cap = (size_t)h->P.kpCapFrame;
    for (int b = 0; b < batch; b++) {
        const int n = h->hN[b];
        n_out[b] = n;
        memcpy(kp_out + b * cap, h->hKp + b * cap, (size_t)n * sizeof(orbfe_keypoint));
        memcpy(desc_out + b * cap * ORBFE_DESC_BYTES, h->hDesc + b * cap * ORBFE_DESC_BYTES, (size_t)n * ORBFE_DESC_BYTES);
        if (per_level) memcpy(per_level + (size_t)b * nL, h->hPer + (size_t)b * nL, nL * sizeof(int));
    }
    return ORBFE_OK;
}

int orbfe_extract(orbfe_handle* h, const uint8_t* gray, int pitch, orbfe_keypoint* kp_out, uint8_t* desc_out,
                  int* n_out, int* per_level)
{
    const uint8_t* one[1] = {gray};
    return orbfe_extract_batch(h, one, pitch, 1, kp_out, desc_out, n_out, per_level);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// orbfe_track_frame: extract -> isInFrustum -> SearchByProjection as one captured hipGraph (orbfe.h)
// ---------------------------------------------------------------------------------------------
struct orbfe_map {  // map points resident in HBM (orbfe_map_*, further down)
    orbfe_handle* h = nullptr;
    int cap = 0;
    orbfe_world_point* dPts = nullptr;
    uint8_t* dDesc = nullptr;
};

namespace {

// map points per graph: rounded up so that a local map that grows by a few points replays the same graph; the padding
// records are "bad" (isInFrustum leaves them out of view, SearchByProjection skips them), so results do not change
int track_bucket(int M)
{
    int g = 256;
    while (g * 8 <= M) g <<= 1;  // <= 8 buckets per octave
    return std::max(g, (M + g - 1) / g * g);
}

struct TrackLayout {
    size_t inFrame, oFr, oPts, oMpDesc, inBytes;                              // input block
    size_t oPer, oKp, oDesc, oMatch, oMps, oXr, outBytes;                     // result block ([n, status, n_matches] at 0)
};

TrackLayout track_layout(const orbfe_handle* h, int Mb)
{
    TrackLayout L{};
    const size_t cap = (size_t)h->P.kpCapFrame;
    L.inFrame = align_up((size_t)h->dInPitch * h->prm.image_height, 256);
    L.oFr = L.inFrame;
    L.oPts = L.oFr + align_up(sizeof(orbfe_frustum), 256);
    L.oMpDesc = L.oPts + (size_t)Mb * sizeof(orbfe_world_point);
    L.inBytes = L.oMpDesc + (size_t)Mb * ORBFE_DESC_BYTES;
    size_t off = 256;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.oPer = take((size_t)h->nLevels * sizeof(int));
    L.oKp = take(cap * sizeof(orbfe_keypoint));
    L.oDesc = take(cap * ORBFE_DESC_BYTES);
    L.oMatch = take(cap * sizeof(int));
    L.oMps = take((size_t)Mb * sizeof(orbfe_map_point));
    L.oXr = take((size_t)Mb * sizeof(float));
    L.outBytes = off;
    return L;
}

void track_drop_graphs(orbfe_handle* h)
{
    for (auto& g : h->trackGraphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->trackGraphs.clear();
}

// (re)allocate the blocks for Mb map points; the caller holds h->mu and nothing of this path is in flight
int track_reserve(orbfe_handle* h, int Mb)
{
    if (Mb <= h->trkCapM) return ORBFE_OK;
    track_drop_graphs(h);
    if (h->dTrkIn) (void)hipFree(h->dTrkIn);
    if (h->dTrkOut) (void)hipFree(h->dTrkOut);
    if (h->hTrkIn) (void)hipHostFree(h->hTrkIn);
    if (h->hTrkOut) (void)hipHostFree(h->hTrkOut);
    h->dTrkIn = h->dTrkOut = h->hTrkIn = h->hTrkOut = nullptr;
    h->trkCapM = 0;
    const int capM = Mb + Mb / 2;  // head-room: a growing local map does not reallocate (and re-capture) at every bucket
    const TrackLayout L = track_layout(h, capM);
    if (hipMalloc(&h->dTrkIn, L.inBytes) != hipSuccess || hipMalloc(&h->dTrkOut, L.outBytes) != hipSuccess ||
        hipHostMalloc(&h->hTrkIn, L.inBytes) != hipSuccess || hipHostMalloc(&h->hTrkOut, L.outBytes) != hipSuccess) {
        (void)hipGetLastError();
        h->err = "orbfe_track_frame: allocation of the staging blocks failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    h->trkCapM = capM;
    return ORBFE_OK;
}

// the device side of one call, enqueued on s (directly, or under stream capture): extraction chain on the uploaded
// frame, projection of the uploaded map points with the uploaded frustum, SearchByProjection on the fresh keypoints,
// download of the result block
int track_enqueue(orbfe_handle* h, const TrackLayout& L, int Mb, int inPitch, proj::ProjArgs& A, const orbfe_map* map, hipStream_t s)
{
    int* dHead = reinterpret_cast<int*>(h->dTrkOut);  // [n, status, n_matches]
    int rc = extract_chain(h, h->dTrkIn, L.inFrame, inPitch, 1, reinterpret_cast<orbfe_keypoint*>(h->dTrkOut + L.oKp),
                           h->dTrkOut + L.oDesc, dHead, reinterpret_cast<int*>(h->dTrkOut + L.oPer), dHead + 1, s);
    if (rc != ORBFE_OK) return rc;
    std::string err;
    const orbfe_frustum* dF = reinterpret_cast<const orbfe_frustum*>(h->dTrkIn + L.oFr);
    orbfe_map_point* dMps = reinterpret_cast<orbfe_map_point*>(h->dTrkOut + L.oMps);
    float* dXr = reinterpret_cast<float*>(h->dTrkOut + L.oXr);
    if (map)  // the block carries ids (at the points' offset): gather record + descriptor from the resident map, then project
        rc = frustum_gather_launch(s, 1, dF, reinterpret_cast<const int*>(h->dTrkIn + L.oPts), Mb, map->cap, map->dPts, map->dDesc, dMps,
                                   h->dTrkIn + L.oMpDesc, dXr, err);
    else
        rc = frustum_launch_dev(s, dF, Mb, reinterpret_cast<const orbfe_world_point*>(h->dTrkIn + L.oPts), dMps, dXr, err);
    if (rc == ORBFE_OK) rc = proj::proj_launch(s, A, err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->hTrkOut, h->dTrkOut, L.outBytes, hipMemcpyDeviceToHost, s));
    return ORBFE_OK;
}

}  // namespace

static int track_frame_impl(orbfe_handle* h, const uint8_t* gray, int pitch, const orbfe_frustum* frustum,
                            const orbfe_track_params* tp, int M, const orbfe_world_point* points, const uint8_t* mp_desc,
                            const orbfe_map* map, const int* ids, orbfe_keypoint* kp_out, uint8_t* desc_out, int* n_out, int* per_level,
                            orbfe_map_point* mp_out, float* proj_xr_out, int* match_out, int* n_matches)
{
    if (!h || !gray || !tp || !kp_out || !desc_out || !n_out || !match_out || !n_matches || M < 0) return ORBFE_ERR_INVALID_ARG;
    if (map ? (map->h != h || (M > 0 && !ids)) : (M > 0 && (!points || !mp_desc))) return ORBFE_ERR_INVALID_ARG;
    if (pitch < h->prm.image_width || pitch >= (1 << 24)) return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    std::lock_guard<std::mutex> lk(h->mu);
    if (tp->struct_size != (int)sizeof(orbfe_track_params)) {
        h->err = "orbfe_track_params.struct_size does not match this library (rebuild the caller against include/orbfe.h)";
        return ORBFE_ERR_INVALID_ARG;
    }
    if (tp->grid_cols < 1 || tp->grid_rows < 1 || frustum->n_levels > h->nLevels) return ORBFE_ERR_INVALID_ARG;
    if (h->P.kpCapFrame >= (1 << 20) || tp->grid_cols > 65535 || tp->grid_rows > 32767 ||
        (long long)tp->grid_cols * tp->grid_rows > proj::kMaxCells || M > (1 << 24))
        return ORBFE_ERR_UNSUPPORTED;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int W = h->prm.image_width, H = h->prm.image_height, nL = h->nLevels;
    const int cap = h->P.kpCapFrame;
    const int Mb = track_bucket(M);
    int rc = track_reserve(h, Mb);
    if (rc != ORBFE_OK) return rc;
    const TrackLayout L = track_layout(h, Mb);

    // ---- matcher arguments; the arena is sized here, outside any capture ----
    proj::ProjArgs A{};
    A.B = 1; A.M = Mb; A.kpStride = cap;
    A.g = proj::GridDesc{tp->grid_cols, tp->grid_rows, tp->min_x, tp->min_y, tp->grid_inv_w, tp->grid_inv_h};
    A.th = tp->th; A.thFar = tp->th_far_points; A.nnRatio = tp->nn_ratio; A.farPoints = tp->far_points; A.bFactor = tp->th != 1.0;
    A.kp = reinterpret_cast<const orbfe_keypoint*>(h->dTrkOut + L.oKp);
    A.desc = h->dTrkOut + L.oDesc;
    A.nKp = reinterpret_cast<const int*>(h->dTrkOut);
    A.mps = reinterpret_cast<const orbfe_map_point*>(h->dTrkOut + L.oMps);
    A.mpDesc = h->dTrkIn + L.oMpDesc;
    A.initObs = nullptr;
    A.scaleFactors = h->dSf; A.nLevels = nL;
    A.matchOut = reinterpret_cast<int*>(h->dTrkOut + L.oMatch);
    A.nMatches = reinterpret_cast<int*>(h->dTrkOut) + 2;
    {
        std::string err;
        void* before = h->trackMatch.d;
        rc = proj::proj_setup(h->trackMatch, A, Carver(), 64, err);
        if (rc != ORBFE_OK) {
            h->err = err;
            return rc;
        }
        if (before && h->trackMatch.d != before) track_drop_graphs(h);  // the arena moved: graphs hold its old address
    }

    // ---- stage the small block: [frustum | points | descriptors] with the padding records marked bad, or -- resident map --
    //      [frustum | ids] with the padding ids outside the map (the gather kernel turns those into bad records) ----
    memcpy(h->hTrkIn + L.oFr, frustum, sizeof(orbfe_frustum));
    size_t smallEnd;  // end of what has to go up behind the frame
    if (map) {
        int* hid = reinterpret_cast<int*>(h->hTrkIn + L.oPts);
        if (M) memcpy(hid, ids, (size_t)M * sizeof(int));
        for (int i = M; i < Mb; i++) hid[i] = 0x7fffffff;
        smallEnd = L.oPts + (size_t)Mb * sizeof(int);
    } else {
        if (M) memcpy(h->hTrkIn + L.oPts, points, (size_t)M * sizeof(orbfe_world_point));
        orbfe_world_point pad{};
        pad.bad = 1;
        pad.skip = 1;
        orbfe_world_point* hp = reinterpret_cast<orbfe_world_point*>(h->hTrkIn + L.oPts);
        for (int i = M; i < Mb; i++) hp[i] = pad;
        if (M) memcpy(h->hTrkIn + L.oMpDesc, mp_desc, (size_t)M * ORBFE_DESC_BYTES);
        if (Mb > M) memset(h->hTrkIn + L.oMpDesc + (size_t)M * ORBFE_DESC_BYTES, 0, (size_t)(Mb - M) * ORBFE_DESC_BYTES);
        smallEnd = L.inBytes;
    }

    // ---- upload: pinned frames straight from the caller's buffer + the small block; pageable ones through the
    //      pinned mirror, where frame and small block are ONE copy ----
    bool direct = pitch <= h->dInPitch && (pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(gray) & 3u) == 0;
    if (direct) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, gray) != hipSuccess || attr.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            direct = false;
        }
    }
    int inPitch;
    if (direct) {
        inPitch = pitch;
        HIPCHK(h, hipMemcpyAsync(h->dTrkIn, gray, (size_t)pitch * (H - 1) + (size_t)W, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->dTrkIn + L.oFr, h->hTrkIn + L.oFr, smallEnd - L.oFr, hipMemcpyHostToDevice, s));
    } else {
        if ((pitch & 3) == 0 && pitch <= h->dInPitch) {  // a dword-aligned pitch is kept: the frame is one contiguous copy
            inPitch = pitch;
            memcpy(h->hTrkIn, gray, (size_t)pitch * (H - 1) + (size_t)W);
        } else {
            inPitch = h->dInPitch;
            for (int y = 0; y < H; y++) memcpy(h->hTrkIn + (size_t)y * inPitch, gray + (size_t)y * pitch, (size_t)W);
        }
        HIPCHK(h, hipMemcpyAsync(h->dTrkIn, h->hTrkIn, smallEnd, hipMemcpyHostToDevice, s));
    }

    // ---- kernels + download: replay the graph of this (bucket, pitch, parameters), capturing it first if needed ----
    bool viaGraph = h->useGraph && !h->timing;
    if (viaGraph) {
        const orbfe_handle::TrackKey key{Mb, inPitch, tp->grid_cols, tp->grid_rows, tp->far_points, tp->min_x, tp->min_y,
                                         tp->grid_inv_w, tp->grid_inv_h, tp->th, tp->nn_ratio, tp->th_far_points, map};
        hipGraphExec_t exec = nullptr;
        for (auto& g : h->trackGraphs)
            if (memcmp(&g.key, &key, sizeof key) == 0) exec = g.exec;
        if (!exec) {
            hipGraph_t graph = nullptr;
            {
                CaptureStream cs;
                rc = cs.begin() ? track_enqueue(h, L, Mb, inPitch, A, map, cs.s) : ORBFE_ERR_HIP;
                graph = cs.end();
            }
            if (rc == ORBFE_OK && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
            if (!exec) {
                graph_capture_failed(h);  // plain launches for this call
                viaGraph = false;
            } else {
                graph_capture_succeeded(h);
                if (h->trackGraphs.size() >= 64) track_drop_graphs(h);  // a caller cycling through parameters: bounded cache
                h->trackGraphs.push_back({key, exec});
            }
        }
        if (viaGraph) {
            rc = scratch_acquire(h, h->extractUsed, h->extractStream, h->evExtract, s);
            if (rc != ORBFE_OK) return rc;
            HIPCHK(h, hipGraphLaunch(exec, s));
            rc = extract_scratch_release(h, s);
            if (rc != ORBFE_OK) return rc;
            h->lastGray = h->dTrkIn;
            h->lastStride = L.inFrame;
            h->lastPitch = inPitch;
            h->lastBatch = 1;
        }
    }
    if (!viaGraph) {
        rc = track_enqueue(h, L, Mb, inPitch, A, map, s);
        if (rc != ORBFE_OK) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(s));

    // ---- hand over ----
    const int* head = reinterpret_cast<const int*>(h->hTrkOut);
    if (head[1]) {
        char buf[96];
        snprintf(buf, sizeof buf, "device guard flags 0x%x in orbfe_track_frame", (unsigned)head[1]);
        h->err = buf;
        return ORBFE_ERR_INTERNAL;
    }
    const int n = head[0];
    *n_out = n;
    *n_matches = n > 0 ? head[2] : 0;
    memcpy(kp_out, h->hTrkOut + L.oKp, (size_t)n * sizeof(orbfe_keypoint));
    memcpy(desc_out, h->hTrkOut + L.oDesc, (size_t)n * ORBFE_DESC_BYTES);
    memcpy(match_out, h->hTrkOut + L.oMatch, (size_t)n * sizeof(int));
    if (per_level) memcpy(per_level, h->hTrkOut + L.oPer, (size_t)nL * sizeof(int));
    if (mp_out && M) memcpy(mp_out, h->hTrkOut + L.oMps, (size_t)M * sizeof(orbfe_map_point));
    if (proj_xr_out && M) memcpy(proj_xr_out, h->hTrkOut + L.oXr, (size_t)M * sizeof(float));
    return ORBFE_OK;
}

extern "C" int orbfe_track_frame(orbfe_handle* h, const uint8_t* gray, int pitch, const orbfe_frustum* frustum,
                                 const orbfe_track_params* tp, int M, const orbfe_world_point* points, const uint8_t* mp_desc,
                                 orbfe_keypoint* kp_out, uint8_t* desc_out, int* n_out, int* per_level, orbfe_map_point* mp_out,
                                 float* proj_xr_out, int* match_out, int* n_matches)
{
    return track_frame_impl(h, gray, pitch, frustum, tp, M, points, mp_desc, nullptr, nullptr, kp_out, desc_out, n_out, per_level, mp_out,
                            proj_xr_out, match_out, n_matches);
}

extern "C" int orbfe_track_frame_map(orbfe_handle* h, const uint8_t* gray, int pitch, const orbfe_frustum* frustum,
                                     const orbfe_track_params* tp, const orbfe_map* map, int M, const int* ids, orbfe_keypoint* kp_out,
                                     uint8_t* desc_out, int* n_out, int* per_level, orbfe_map_point* mp_out, float* proj_xr_out,
                                     int* match_out, int* n_matches)
{
    if (!map) return ORBFE_ERR_INVALID_ARG;
    return track_frame_impl(h, gray, pitch, frustum, tp, M, nullptr, nullptr, map, ids, kp_out, desc_out, n_out, per_level, mp_out,
                            proj_xr_out, match_out, n_matches);
}

// ---------------------------------------------------------------------------------------------
// Pipelined host-pointer extraction (orbfe.h: orbfe_stream_*)
// ---------------------------------------------------------------------------------------------
namespace {

// a few persistent workers for the row-by-row re-pitch of pageable frames into pinned memory (one thread moves
// ~8 GB/s of 752-byte rows: a third of what the PCIe link takes)
class RowCopyPool {
public:
    explicit RowCopyPool(int n)
    {
        for (int i = 0; i < n; i++) workers_.emplace_back([this] { loop(); });
    }
    ~RowCopyPool()
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            stop_ = true;
        }
        cv_.notify_all();
        for (auto& t : workers_) t.join();
    }
    // run fn(i) for i in [0, n) on the workers and the caller; returns when all are done
    void parallel_for(int n, const std::function<void(int)>& fn)
    {
        {
            std::lock_guard<std::mutex> lk(mu_);
            fn_ = &fn;
            next_ = 0;
            end_ = n;
            pending_ = n;
        }
        cv_.notify_all();
        run_some();
        std::unique_lock<std::mutex> lk(mu_);
        done_.wait(lk, [this] { return pending_ == 0; });
        fn_ = nullptr;
    }

private:
    void run_some()
    {
        for (;;) {
            int i;
            const std::function<void(int)>* fn;
            {
                std::lock_guard<std::mutex> lk(mu_);
                if (!fn_ || next_ >= end_) return;
                i = next_++;
                fn = fn_;
            }
            (*fn)(i);
            std::lock_guard<std::mutex> lk(mu_);
            if (--pending_ == 0) done_.notify_all();
        }
    }
    void loop()
    {
        for (;;) {
            {
                std::unique_lock<std::mutex> lk(mu_);
                cv_.wait(lk, [this] { return stop_ || (fn_ && next_ < end_); });
                if (stop_) return;
            }
            run_some();
        }
    }
    std::vector<std::thread> workers_;
    std::mutex mu_;
    std::condition_variable cv_, done_;
    const std::function<void(int)>* fn_ = nullptr;
    int next_ = 0, end_ = 0, pending_ = 0;
    bool stop_ = false;
};

struct StreamSlot {
    uint8_t* hIn = nullptr;   // pinned, slotFrames x inFrame
    uint8_t* dIn = nullptr;
    uint8_t* dOut = nullptr;  // [n | status | per-level | keypoints | descriptors], the handle's block layout
    uint8_t* hOut = nullptr;  // pinned mirror
    // extract-and-match submissions (orbfe_stream_enable_track): per frame frustum + map-point ids up, matches down
    uint8_t* hTrkIn = nullptr;   // pinned [frusta (slotFrames) | ids (slotFrames x maxPoints)]
    uint8_t* dTrkIn = nullptr;
    uint8_t* dTrkWork = nullptr; // [map-point records | gathered descriptors]
    uint8_t* dTrkOut = nullptr;  // [n_matches (slotFrames) | match (slotFrames x cap)]
    uint8_t* hTrkOut = nullptr;  // pinned mirror
    hipEvent_t evIn = nullptr, evDone = nullptr, evOut = nullptr;
    int n = 0;
    bool track = false;
};

}  // namespace

struct orbfe_stream {
    orbfe_handle* h = nullptr;
    int nSlots = 0, slotFrames = 0;
    size_t inFrame = 0, outBytes = 0, offStatus = 0, offPer = 0, offKp = 0, offDesc = 0;
    hipStream_t sIn = nullptr, sOut = nullptr;  // upload / download; the kernels run on the handle's stream
    std::vector<StreamSlot> slots;
    // one producer (submit) and one consumer (collect) may run concurrently: a slot belongs to its submission from the
    // moment `submitted` passes it until `collected` does
    std::atomic<unsigned long long> submitted{0}, collected{0};
    RowCopyPool* pool = nullptr;
    std::mutex poolMu;  // RowCopyPool::parallel_for is not re-entrant: the two sides take turns
    // extract-and-match
    orbfe_map* map = nullptr;
    int maxPoints = 0;
    size_t trkInBytes = 0, offIds = 0, trkWorkBytes = 0, offGatherDesc = 0, trkOutBytes = 0, offMatch = 0;
};

extern "C" {

}  // extern "C"

// everything a ring owns except the struct itself; the handle is still alive
static void stream_release(orbfe_stream* st)
{
    orbfe_handle* h = st->h;
    (void)hipSetDevice(h->device);
    if (st->sIn) (void)hipStreamSynchronize(st->sIn);
    (void)hipStreamSynchronize(h->stream);
    if (st->sOut) (void)hipStreamSynchronize(st->sOut);
    for (auto& sl : st->slots) {
        if (sl.hIn) (void)hipHostFree(sl.hIn);
        if (sl.dIn) (void)hipFree(sl.dIn);
        if (sl.dOut) (void)hipFree(sl.dOut);
        if (sl.hOut) (void)hipHostFree(sl.hOut);
        if (sl.hTrkIn) (void)hipHostFree(sl.hTrkIn);
        if (sl.dTrkIn) (void)hipFree(sl.dTrkIn);
        if (sl.dTrkWork) (void)hipFree(sl.dTrkWork);
        if (sl.dTrkOut) (void)hipFree(sl.dTrkOut);
        if (sl.hTrkOut) (void)hipHostFree(sl.hTrkOut);
        for (hipEvent_t e : {sl.evIn, sl.evDone, sl.evOut})
            if (e) (void)hipEventDestroy(e);
    }
    st->slots.clear();
    if (st->sIn) (void)hipStreamDestroy(st->sIn);
    if (st->sOut) (void)hipStreamDestroy(st->sOut);
    st->sIn = st->sOut = nullptr;
    delete st->pool;
    st->pool = nullptr;
    st->map = nullptr;
}

extern "C" {

void orbfe_stream_destroy(orbfe_stream* st)
{
    if (!st) return;
    if (st->h) {  // (null: the handle went first and released the ring's memory, orphan_children)
        orbfe_handle* h = st->h;
        {
            std::lock_guard<std::mutex> lk(h->mu);
            for (size_t i = 0; i < h->rings.size(); i++)
                if (h->rings[i] == st) h->rings.erase(h->rings.begin() + (long)i--);
        }
        stream_release(st);
    }
    delete st;
}

int orbfe_stream_create(orbfe_handle* h, int slots, int slot_frames, orbfe_stream** out)
{
    if (!h || !out || slots < 2 || slots > 64 || slot_frames < 1 || slot_frames > h->maxBatch) return ORBFE_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    orbfe_stream* st = new (std::nothrow) orbfe_stream();
    if (!st) return ORBFE_ERR_OUT_OF_MEMORY;
    st->h = h;
    st->nSlots = slots;
    st->slotFrames = slot_frames;
    st->inFrame = (size_t)h->dInPitch * h->prm.image_height;
    const size_t B = (size_t)slot_frames, cap = (size_t)h->P.kpCapFrame;
    {
        size_t off = 0;
        auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
        take(B * sizeof(int));
        st->offStatus = take(B * sizeof(int));
        st->offPer = take(B * h->nLevels * sizeof(int));
        st->offKp = take(B * cap * sizeof(orbfe_keypoint));
        st->offDesc = take(B * cap * ORBFE_DESC_BYTES);
        st->outBytes = off;
    }
    st->slots.resize((size_t)slots);
    bool ok = hipStreamCreateWithFlags(&st->sIn, hipStreamNonBlocking) == hipSuccess &&
              hipStreamCreateWithFlags(&st->sOut, hipStreamNonBlocking) == hipSuccess;
    for (auto& sl : st->slots) {
        ok = ok && hipHostMalloc(&sl.hIn, st->inFrame * B) == hipSuccess && hipMalloc(&sl.dIn, st->inFrame * B) == hipSuccess &&
             hipMalloc(&sl.dOut, st->outBytes) == hipSuccess && hipHostMalloc(&sl.hOut, st->outBytes) == hipSuccess &&
             hipEventCreateWithFlags(&sl.evIn, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.evDone, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&sl.evOut, hipEventDisableTiming) == hipSuccess;
    }
    if (!ok) {
        (void)hipGetLastError();
        h->err = "orbfe_stream_create: allocation failed";
        stream_release(st);  // (not yet registered with the handle; h->mu is held)
        delete st;
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    const unsigned hw = std::thread::hardware_concurrency();
    st->pool = new RowCopyPool((int)std::min<unsigned>(7u, hw > 2 ? hw / 2 : 1));
    h->rings.push_back(st);
    *out = st;
    return ORBFE_OK;
}

int orbfe_stream_in_flight(const orbfe_stream* st)
{
    return st ? (int)(st->submitted.load(std::memory_order_acquire) - st->collected.load(std::memory_order_acquire)) : 0;
}

// the common submission: upload, extraction chain, optionally projection + SearchByProjection of the slot's frames, download
static int stream_submit_impl(orbfe_stream* st, const uint8_t* const* grays, int pitch, int n, const orbfe_track_params* tp,
                              const orbfe_frustum* frusta, int nPoints, const int* ids)
{
    if (!st || !st->h || !grays || n < 1 || n > st->slotFrames) return ORBFE_ERR_INVALID_ARG;
    orbfe_handle* h = st->h;
    if (pitch < h->prm.image_width) return ORBFE_ERR_INVALID_ARG;
    const bool track = tp != nullptr;
    if (track) {
        if (!st->map || !frusta || nPoints < 0 || nPoints > st->maxPoints || (nPoints > 0 && !ids)) return ORBFE_ERR_INVALID_ARG;
        if (tp->struct_size != (int)sizeof(orbfe_track_params) || tp->grid_cols < 1 || tp->grid_rows < 1) return ORBFE_ERR_INVALID_ARG;
        for (int b = 0; b < n; b++) {
            const int rcf = frustum_validate(&frusta[b]);
            if (rcf != ORBFE_OK) return rcf;
            if (frusta[b].n_levels > h->nLevels) return ORBFE_ERR_INVALID_ARG;
        }
    }
    const unsigned long long seq = st->submitted.load(std::memory_order_relaxed);
    if (seq - st->collected.load(std::memory_order_acquire) >= (unsigned long long)st->nSlots) return ORBFE_ERR_BUSY;
    StreamSlot& sl = st->slots[seq % st->nSlots];
    const int W = h->prm.image_width, H = h->prm.image_height;
    // (the slot was collected: its previous upload, kernels and download have completed -- evOut was waited for)
    bool direct = pitch <= h->dInPitch && (pitch & 3) == 0;
    for (int b = 0; b < n; b++) {
        if (!grays[b]) return ORBFE_ERR_INVALID_ARG;
        if (direct) {
            hipPointerAttribute_t attr;
            if ((reinterpret_cast<uintptr_t>(grays[b]) & 3u) != 0 || hipPointerGetAttributes(&attr, grays[b]) != hipSuccess ||
                attr.type != hipMemoryTypeHost) {
                (void)hipGetLastError();
                direct = false;
            }
        }
    }
    int stagedPitch = h->dInPitch;
    size_t frameStride = st->inFrame;
    if (!direct) {
        // pageable sources go through the slot's pinned staging block on a pool of copy threads -- BEFORE the handle is
        // locked: a consumer thread's collect may need the handle meanwhile.  A dword-aligned source pitch is kept (the
        // kernels read it as it is), so a band of rows is ONE contiguous copy; other pitches are re-pitched row by row.
        const bool keep = (pitch & 3) == 0 && pitch <= h->dInPitch;
        const int dp = keep ? pitch : h->dInPitch;
        const size_t inFrame = st->inFrame;
        uint8_t* hIn = sl.hIn;
        const int bands = 4;  // row bands per frame so that a handful of frames still spreads over the pool
        std::lock_guard<std::mutex> pk(st->poolMu);
        st->pool->parallel_for(n * bands, [&](int job) {
            const int b = job / bands, band = job - b * bands;
            const int y0 = H * band / bands, y1 = H * (band + 1) / bands;
            if (keep) {
                const size_t bytes = (size_t)(y1 - y0 - 1) * pitch + (size_t)W;
                memcpy(hIn + b * inFrame + (size_t)y0 * dp, grays[b] + (size_t)y0 * pitch, bytes);
            } else {
                for (int y = y0; y < y1; y++) memcpy(hIn + b * inFrame + (size_t)y * dp, grays[b] + (size_t)y * pitch, (size_t)W);
            }
        });
        stagedPitch = dp;
    }
    if (track) {  // [frusta | ids] of the slot into its pinned block
        memcpy(sl.hTrkIn, frusta, (size_t)n * sizeof(orbfe_frustum));
        if (nPoints) memcpy(sl.hTrkIn + st->offIds, ids, (size_t)n * nPoints * sizeof(int));
    }
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    if (direct) {
        // pinned sources: DMA straight from the caller's memory.  Frames that sit at a constant distance (a packed
        // [n][H][pitch] block is the usual case) go as ONE copy and keep that distance on the device.
        const size_t bytes = (size_t)pitch * (H - 1) + (size_t)((W + 3) & ~3);
        bool block = n > 1 && grays[1] > grays[0];
        const size_t dist = block ? (size_t)(grays[1] - grays[0]) : 0;
        for (int b = 2; b < n && block; b++) block = grays[b] == grays[b - 1] + dist;
        block = block && (dist & 3) == 0 && dist >= bytes && dist * (size_t)(n - 1) + bytes <= st->inFrame * (size_t)st->slotFrames;
        if (block) {
            frameStride = dist;
            HIPCHK(h, hipMemcpyAsync(sl.dIn, grays[0], dist * (size_t)(n - 1) + bytes, hipMemcpyHostToDevice, st->sIn));
        } else {
            for (int b = 0; b < n; b++)
                HIPCHK(h, hipMemcpyAsync(sl.dIn + b * st->inFrame, grays[b], std::min(bytes, st->inFrame), hipMemcpyHostToDevice, st->sIn));
        }
    } else {
        HIPCHK(h, hipMemcpyAsync(sl.dIn, sl.hIn, st->inFrame * (size_t)n, hipMemcpyHostToDevice, st->sIn));
    }
    if (track) {
        HIPCHK(h, hipMemcpyAsync(sl.dTrkIn, sl.hTrkIn, (size_t)n * sizeof(orbfe_frustum), hipMemcpyHostToDevice, st->sIn));
        if (nPoints)
            HIPCHK(h, hipMemcpyAsync(sl.dTrkIn + st->offIds, sl.hTrkIn + st->offIds, (size_t)n * nPoints * sizeof(int), hipMemcpyHostToDevice,
                                     st->sIn));
    }
    HIPCHK(h, hipEventRecord(sl.evIn, st->sIn));
    // kernels on the handle's stream, behind the upload
    HIPCHK(h, hipStreamWaitEvent(h->stream, sl.evIn, 0));
    const int inPitch = direct ? pitch : stagedPitch;
    orbfe_keypoint* dKp = reinterpret_cast<orbfe_keypoint*>(sl.dOut + st->offKp);
    const int rc = extract_chain(h, sl.dIn, frameStride, inPitch, n, dKp, sl.dOut + st->offDesc, reinterpret_cast<int*>(sl.dOut),
                                 reinterpret_cast<int*>(sl.dOut + st->offPer), reinterpret_cast<int*>(sl.dOut + st->offStatus), h->stream);
    if (rc != ORBFE_OK) return rc;
    if (track) {
        std::string err;
        orbfe_map_point* dMps = reinterpret_cast<orbfe_map_point*>(sl.dTrkWork);
        uint8_t* dMpDesc = sl.dTrkWork + st->offGatherDesc;
        int rcm = frustum_gather_launch(h->stream, n, reinterpret_cast<const orbfe_frustum*>(sl.dTrkIn),
                                        reinterpret_cast<const int*>(sl.dTrkIn + st->offIds), nPoints, st->map->cap, st->map->dPts,
                                        st->map->dDesc, dMps, dMpDesc, nullptr, err);
        if (rcm == ORBFE_OK) {
            MatchScope scope_(h, h->stream);
            rcm = scope_.rc;
            if (rcm == ORBFE_OK)
                rcm = match_projection_batch_device(h->match, h->stream, n, dKp, sl.dOut + st->offDesc, reinterpret_cast<const int*>(sl.dOut),
                                                    h->P.kpCapFrame, tp->grid_cols, tp->grid_rows, tp->min_x, tp->min_y, tp->grid_inv_w,
                                                    tp->grid_inv_h, h->dSf, h->nLevels, nPoints, dMps, dMpDesc, nullptr, tp->th, tp->far_points,
                                                    tp->th_far_points, tp->nn_ratio, reinterpret_cast<int*>(sl.dTrkOut + st->offMatch),
                                                    reinterpret_cast<int*>(sl.dTrkOut), err);
        }
        if (rcm != ORBFE_OK) {
            h->err = err;
            return rcm;
        }
    }
    HIPCHK(h, hipEventRecord(sl.evDone, h->stream));
    // download behind the kernels: the whole block for a full slot (one copy), the used parts otherwise
    HIPCHK(h, hipStreamWaitEvent(st->sOut, sl.evDone, 0));
    if (n == st->slotFrames) {
        HIPCHK(h, hipMemcpyAsync(sl.hOut, sl.dOut, st->outBytes, hipMemcpyDeviceToHost, st->sOut));
    } else {
        const size_t cap = (size_t)h->P.kpCapFrame;
        HIPCHK(h, hipMemcpyAsync(sl.hOut, sl.dOut, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st->sOut));
        HIPCHK(h, hipMemcpyAsync(sl.hOut + st->offStatus, sl.dOut + st->offStatus, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st->sOut));
        HIPCHK(h, hipMemcpyAsync(sl.hOut + st->offPer, sl.dOut + st->offPer, (size_t)n * h->nLevels * sizeof(int), hipMemcpyDeviceToHost, st->sOut));
        HIPCHK(h, hipMemcpyAsync(sl.hOut + st->offKp, sl.dOut + st->offKp, (size_t)n * cap * sizeof(orbfe_keypoint), hipMemcpyDeviceToHost, st->sOut));
        HIPCHK(h, hipMemcpyAsync(sl.hOut + st->offDesc, sl.dOut + st->offDesc, (size_t)n * cap * ORBFE_DESC_BYTES, hipMemcpyDeviceToHost, st->sOut));
    }
    if (track) {
        const size_t cap = (size_t)h->P.kpCapFrame;
        if (n == st->slotFrames) {
            HIPCHK(h, hipMemcpyAsync(sl.hTrkOut, sl.dTrkOut, st->trkOutBytes, hipMemcpyDeviceToHost, st->sOut));
        } else {
            HIPCHK(h, hipMemcpyAsync(sl.hTrkOut, sl.dTrkOut, (size_t)n * sizeof(int), hipMemcpyDeviceToHost, st->sOut));
            HIPCHK(h, hipMemcpyAsync(sl.hTrkOut + st->offMatch, sl.dTrkOut + st->offMatch, (size_t)n * cap * sizeof(int), hipMemcpyDeviceToHost,
                                     st->sOut));
        }
    }
    HIPCHK(h, hipEventRecord(sl.evOut, st->sOut));
    sl.n = n;
    sl.track = track;
    st->submitted.store(seq + 1, std::memory_order_release);
    return ORBFE_OK;
}

int orbfe_stream_submit(orbfe_stream* st, const uint8_t* const* grays, int pitch, int n)
{
    return stream_submit_impl(st, grays, pitch, n, nullptr, nullptr, 0, nullptr);
}

int orbfe_stream_submit_track(orbfe_stream* st, const uint8_t* const* grays, int pitch, int n, const orbfe_track_params* tp,
                              const orbfe_frustum* frusta, int n_points, const int* ids)
{
    if (!tp) return ORBFE_ERR_INVALID_ARG;
    return stream_submit_impl(st, grays, pitch, n, tp, frusta, n_points, ids);
}

int orbfe_stream_enable_track(orbfe_stream* st, orbfe_map* map, int max_points)
{
    if (!st || !st->h || !map || map->h != st->h || max_points < 1 || max_points > (1 << 24)) return ORBFE_ERR_INVALID_ARG;
    if (st->map || st->slots[0].dTrkIn || st->submitted.load() != st->collected.load()) return ORBFE_ERR_INVALID_ARG;  // once, on an idle ring
    orbfe_handle* h = st->h;
    if (h->P.kpCapFrame >= (1 << 20)) return ORBFE_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    const size_t F = (size_t)st->slotFrames, M = (size_t)max_points, cap = (size_t)h->P.kpCapFrame;
    st->offIds = align_up(F * sizeof(orbfe_frustum), 256);
    st->trkInBytes = st->offIds + F * M * sizeof(int);
    st->offGatherDesc = align_up(F * M * sizeof(orbfe_map_point), 256);
    st->trkWorkBytes = st->offGatherDesc + F * M * ORBFE_DESC_BYTES;
    st->offMatch = align_up(F * sizeof(int), 256);
    st->trkOutBytes = st->offMatch + F * cap * sizeof(int);
    bool ok = true;
    for (auto& sl : st->slots)
        ok = ok && hipHostMalloc(&sl.hTrkIn, st->trkInBytes) == hipSuccess && hipMalloc(&sl.dTrkIn, st->trkInBytes) == hipSuccess &&
             hipMalloc(&sl.dTrkWork, st->trkWorkBytes) == hipSuccess && hipMalloc(&sl.dTrkOut, st->trkOutBytes) == hipSuccess &&
             hipHostMalloc(&sl.hTrkOut, st->trkOutBytes) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        // give back what was allocated: the ring stays usable for plain submissions, and a retry starts from empty slots
        for (auto& sl : st->slots) {
            if (sl.hTrkIn) (void)hipHostFree(sl.hTrkIn);
            if (sl.dTrkIn) (void)hipFree(sl.dTrkIn);
            if (sl.dTrkWork) (void)hipFree(sl.dTrkWork);
            if (sl.dTrkOut) (void)hipFree(sl.dTrkOut);
            if (sl.hTrkOut) (void)hipHostFree(sl.hTrkOut);
            sl.hTrkIn = sl.dTrkIn = sl.dTrkWork = sl.dTrkOut = sl.hTrkOut = nullptr;
        }
        h->err = "orbfe_stream_enable_track: allocation failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    st->map = map;
    st->maxPoints = max_points;
    return ORBFE_OK;
}

// waits for the oldest submission and checks its guard flags; *slot_out stays in flight until the caller bumps `collected`
static int stream_wait_oldest(orbfe_stream* st, StreamSlot** slot_out)
{
    if (!st || !st->h) return ORBFE_ERR_INVALID_ARG;
    const unsigned long long seq = st->collected.load(std::memory_order_relaxed);
    if (st->submitted.load(std::memory_order_acquire) == seq) return ORBFE_ERR_INVALID_ARG;
    orbfe_handle* h = st->h;
    StreamSlot& sl = st->slots[seq % st->nSlots];
    HIPCHK(h, hipSetDevice(h->device));
    {
        const hipError_t e = hipEventSynchronize(sl.evOut);  // (no handle lock: a producer may be submitting right now)
        if (e != hipSuccess) {
            std::lock_guard<std::mutex> lk(h->mu);
            return fail_hip(h, e, "hipEventSynchronize(slot)", __LINE__);
        }
    }
    *slot_out = &sl;
    const int* status = reinterpret_cast<const int*>(sl.hOut + st->offStatus);
    for (int b = 0; b < sl.n; b++)
        if (status[b]) {
            char buf[128];
            snprintf(buf, sizeof buf, "device guard flags 0x%x at frame %d of the collected submission", (unsigned)status[b], b);
            std::lock_guard<std::mutex> lk(h->mu);
            h->err = buf;
            st->collected.store(seq + 1, std::memory_order_release);
            return ORBFE_ERR_INTERNAL;
        }
    return ORBFE_OK;
}

int orbfe_stream_collect_view(orbfe_stream* st, const orbfe_keypoint** kp, const uint8_t** desc, const int** n,
                              const int** per_level, int* n_frames)
{
    StreamSlot* sl = nullptr;
    const int rc = stream_wait_oldest(st, &sl);
    if (rc != ORBFE_OK) return rc;
    if (kp) *kp = reinterpret_cast<const orbfe_keypoint*>(sl->hOut + st->offKp);
    if (desc) *desc = sl->hOut + st->offDesc;
    if (n) *n = reinterpret_cast<const int*>(sl->hOut);
    if (per_level) *per_level = reinterpret_cast<const int*>(sl->hOut + st->offPer);
    if (n_frames) *n_frames = sl->n;
    st->collected.fetch_add(1, std::memory_order_release);
    return ORBFE_OK;
}

static int stream_collect_impl(orbfe_stream* st, orbfe_keypoint* kp_out, uint8_t* desc_out, int* n_out, int* per_level, int* match_out,
                               int* n_matches, int* n_frames, bool wantTrack)
{
    if (!st || !kp_out || !desc_out || !n_out || (wantTrack && (!match_out || !n_matches))) return ORBFE_ERR_INVALID_ARG;
    StreamSlot* sl = nullptr;
    const int rc = stream_wait_oldest(st, &sl);
    if (rc != ORBFE_OK) return rc;
    orbfe_handle* h = st->h;
    if (wantTrack && !sl->track) {  // the oldest submission carried no map points: leave it in flight for orbfe_stream_collect
        std::lock_guard<std::mutex> lk(h->mu);
        h->err = "orbfe_stream_collect_track: the oldest submission was made with orbfe_stream_submit";
        return ORBFE_ERR_INVALID_ARG;
    }
    const size_t cap = (size_t)h->P.kpCapFrame;
    const int nL = h->nLevels;
    const int* hn = reinterpret_cast<const int*>(sl->hOut);
    const orbfe_keypoint* hkp = reinterpret_cast<const orbfe_keypoint*>(sl->hOut + st->offKp);
    const uint8_t* hdesc = sl->hOut + st->offDesc;
    const int* hper = reinterpret_cast<const int*>(sl->hOut + st->offPer);
    const int* hnm = wantTrack ? reinterpret_cast<const int*>(sl->hTrkOut) : nullptr;
    const int* hmatch = wantTrack ? reinterpret_cast<const int*>(sl->hTrkOut + st->offMatch) : nullptr;
    const int nfr = sl->n;
    {
        std::lock_guard<std::mutex> pk(st->poolMu);
        st->pool->parallel_for(nfr, [&](int b) {
            const int k = hn[b];
            n_out[b] = k;
            memcpy(kp_out + b * cap, hkp + b * cap, (size_t)k * sizeof(orbfe_keypoint));
            memcpy(desc_out + b * cap * ORBFE_DESC_BYTES, hdesc + b * cap * ORBFE_DESC_BYTES, (size_t)k * ORBFE_DESC_BYTES);
            if (per_level) memcpy(per_level + (size_t)b * nL, hper + (size_t)b * nL, nL * sizeof(int));
            if (wantTrack) {
                n_matches[b] = k > 0 ? hnm[b] : 0;
                memcpy(match_out + b * cap, hmatch + b * cap, (size_t)k * sizeof(int));
            }
        });
    }
    if (n_frames) *n_frames = nfr;
    st->collected.fetch_add(1, std::memory_order_release);
    return ORBFE_OK;
}

int orbfe_stream_collect(orbfe_stream* st, orbfe_keypoint* kp_out, uint8_t* desc_out, int* n_out, int* per_level, int* n_frames)
{
    return stream_collect_impl(st, kp_out, desc_out, n_out, per_level, nullptr, nullptr, n_frames, false);
}

int orbfe_stream_collect_track(orbfe_stream* st, orbfe_keypoint* kp_out, uint8_t* desc_out, int* n_out, int* per_level, int* match_out,
                               int* n_matches, int* n_frames)
{
    return stream_collect_impl(st, kp_out, desc_out, n_out, per_level, match_out, n_matches, n_frames, true);
}

// ---------------------------------------------------------------------------------------------
// Map points resident in HBM (orbfe.h: orbfe_map_*)
// ---------------------------------------------------------------------------------------------
int orbfe_map_create(orbfe_handle* h, int capacity, orbfe_map** out)
{
    if (!h || !out || capacity < 1 || capacity > (1 << 26)) return ORBFE_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    orbfe_map* m = new (std::nothrow) orbfe_map();
    if (!m) return ORBFE_ERR_OUT_OF_MEMORY;
    m->h = h;
    m->cap = capacity;
    if (hipMalloc(&m->dPts, (size_t)capacity * sizeof(orbfe_world_point)) != hipSuccess ||
        hipMalloc(&m->dDesc, (size_t)capacity * ORBFE_DESC_BYTES) != hipSuccess) {
        (void)hipGetLastError();
        if (m->dPts) (void)hipFree(m->dPts);
        delete m;
        h->err = "orbfe_map_create: allocation failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    // an entry that was never written is a bad point: skipped by isInFrustum and by the matcher
    std::vector<orbfe_world_point> init((size_t)std::min(capacity, 1 << 16));
    for (auto& p : init) { p = orbfe_world_point{}; p.bad = 1; }
    for (size_t o = 0; o < (size_t)capacity; o += init.size())
        if (copy_sync(m->dPts + o, init.data(), std::min(init.size(), (size_t)capacity - o) * sizeof(orbfe_world_point),
                      hipMemcpyHostToDevice, h->stream) != hipSuccess) {
            (void)hipGetLastError();
            (void)hipFree(m->dPts);
            (void)hipFree(m->dDesc);
            delete m;
            return ORBFE_ERR_HIP;
        }
    (void)memset_sync(m->dDesc, 0, (size_t)capacity * ORBFE_DESC_BYTES, h->stream);
    h->maps.push_back(m);
    *out = m;
    return ORBFE_OK;
}

// the map's device memory and every reference to it; the caller holds m->h->mu (or is the handle's destruction)
static void map_release(orbfe_map* m)
{
    orbfe_handle* h = m->h;
    (void)hipSetDevice(h->device);
    (void)hipStreamSynchronize(h->stream);
    track_drop_graphs(h);  // graphs of orbfe_track_frame_map hold this map's addresses
    for (orbfe_stream* st : h->rings)
        if (st->map == m) {  // a ring that was given this map: its later track submissions are refused, not served from freed memory
            st->map = nullptr;
            st->maxPoints = 0;
        }
    if (m->dPts) (void)hipFree(m->dPts);
    if (m->dDesc) (void)hipFree(m->dDesc);
    m->dPts = nullptr;
    m->dDesc = nullptr;
}

void orbfe_map_destroy(orbfe_map* m)
{
    if (!m) return;
    if (m->h) {  // (null: the handle went first, orphan_children)
        std::lock_guard<std::mutex> lk(m->h->mu);
        for (size_t i = 0; i < m->h->maps.size(); i++)
            if (m->h->maps[i] == m) m->h->maps.erase(m->h->maps.begin() + (long)i--);
        map_release(m);
    }
    delete m;
}

int orbfe_map_update(orbfe_handle* h, orbfe_map* m, int n, const int* ids, const orbfe_world_point* points, const uint8_t* desc)
{
    if (!h || !m || m->h != h || n < 0 || (n > 0 && (!ids || !points || !desc))) return ORBFE_ERR_INVALID_ARG;
    for (int i = 0; i < n; i++)
        if (ids[i] < 0 || ids[i] >= m->cap) return ORBFE_ERR_INVALID_ARG;
    if (n == 0) return ORBFE_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    // staged through the matcher's grow-only arenas: [ids | points | descriptors]; on the handle's stream, i.e. behind
    // every submission made so far (their kernels run there) and in front of every later one
    Carver c;
    const size_t oIds = c.take((size_t)n * sizeof(int));
    const size_t oPts = c.take((size_t)n * sizeof(orbfe_world_point));
    const size_t oDesc = c.take((size_t)n * ORBFE_DESC_BYTES);
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    int rc = ensure(h->match, c.off, c.off, err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    uint8_t* hp = static_cast<uint8_t*>(h->match.hpin);
    uint8_t* dp = static_cast<uint8_t*>(h->match.d);
    memcpy(hp + oIds, ids, (size_t)n * sizeof(int));
    memcpy(hp + oPts, points, (size_t)n * sizeof(orbfe_world_point));
    memcpy(hp + oDesc, desc, (size_t)n * ORBFE_DESC_BYTES);
    HIPCHK(h, hipMemcpyAsync(dp, hp, c.off, hipMemcpyHostToDevice, h->stream));
    rc = map_scatter_launch(h->stream, n, reinterpret_cast<const int*>(dp + oIds), reinterpret_cast<const orbfe_world_point*>(dp + oPts),
                            dp + oDesc, m->cap, m->dPts, m->dDesc, err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    HIPCHK(h, hipStreamSynchronize(h->stream));
    return ORBFE_OK;
}

}  // extern "C"

namespace {
// orbfe_destroy with maps / rings still alive: release their memory now (the device context of the handle is still
// current) and cut the back pointers; their own destroy calls then only delete the shells
void orphan_children(orbfe_handle* h)
{
    for (orbfe_stream* st : h->rings) {
        stream_release(st);
        st->h = nullptr;
    }
    h->rings.clear();
    for (orbfe_map* m : h->maps) {
        map_release(m);
        m->h = nullptr;
    }
    h->maps.clear();
}
}  // namespace

extern "C" {

int orbfe_get_pyramid_level(orbfe_handle* h, int frame, int level, int blurred, uint8_t* out, int out_pitch)
{
    if (!h || !out || level < 0 || level >= h->nLevels || frame < 0) return ORBFE_ERR_INVALID_ARG;
    const LevelDesc& L = h->P.lv[level];
    if (out_pitch < L.w) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (frame >= h->lastBatch) return ORBFE_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->extractUsed) HIPCHK(h, hipEventSynchronize(h->evExtract));  // the last call, whatever stream it ran on
    const uint8_t* src;
    size_t spitch;
    if (blurred) {
        src = h->ws + L.blurOff + (size_t)frame * L.blurFrameStride;
        spitch = L.pitch;
    } else if (level == 0) {
        src = h->lastGray + (size_t)frame * h->lastStride;
        spitch = h->lastPitch;
    } else {
        src = h->ws + L.imgOff + (size_t)frame * L.imgFrameStride;
        spitch = L.pitch;
    }
    HIPCHK(h, hipMemcpy2D(out, out_pitch, src, spitch, L.w, L.h, hipMemcpyDeviceToHost));
    return ORBFE_OK;
}

int orbfe_debug_get_candidates(orbfe_handle* h, int frame, int level, uint32_t* packed, int cap, int* n_out,
                               int counters[4])
{
    if (!h || !n_out || level < 0 || level >= h->nLevels || frame < 0) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (frame >= h->lastBatch) return ORBFE_ERR_INVALID_ARG;
    HIPCHK(h, hipSetDevice(h->device));
    if (h->extractUsed) HIPCHK(h, hipEventSynchronize(h->evExtract));
    uint32_t c[kCntWords];
    HIPCHK(h, copy_sync(c, h->dCounters + ((size_t)frame * h->nLevels + level) * kCntWords, sizeof c, hipMemcpyDeviceToHost, h->stream));
    const LevelDesc& L = h->P.lv[level];
    int n = (int)std::min<uint32_t>(c[kCntCand], (uint32_t)L.candCap);
    if (counters) {
        counters[0] = (int)c[kCntCand];
        counters[1] = (int)c[kCntHigh];
        counters[2] = (int)c[kCntPreLow];
        counters[3] = (int)c[kCntPreHigh];
    }
    const int m = std::min(n, cap);
    if (packed && m > 0)
        HIPCHK(h, copy_sync(packed, h->dCand + L.candOff + (size_t)frame * L.candCap, (size_t)m * sizeof(uint32_t),
                            hipMemcpyDeviceToHost, h->stream));
    *n_out = m;
    return ORBFE_OK;
}

int orbfe_set_graph_capture(orbfe_handle* h, int enable)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    h->useGraph = enable != 0;
    if (enable) h->captureFailures = h->captureFailStreak = 0;
    return ORBFE_OK;
}

int orbfe_set_stream_priority(orbfe_handle* h, int high)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    int least = 0, greatest = 0;
    HIPCHK(h, hipDeviceGetStreamPriorityRange(&least, &greatest));  // numerically lower = higher priority
    hipStream_t ns = nullptr;
    HIPCHK(h, hipStreamCreateWithPriority(&ns, hipStreamNonBlocking, high ? greatest : least));
    // the handle must be idle: everything it enqueued has finished before the stream goes away
    if (hipStreamSynchronize(h->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipStreamDestroy(ns);
        return ORBFE_ERR_HIP;
    }
    (void)hipStreamDestroy(h->stream);
    h->stream = ns;
    h->extractUsed = h->matchUsed = false;  // the hand-over events belonged to work that is complete
    h->extractStream = h->matchStream = nullptr;
    return ORBFE_OK;
}

int orbfe_debug_clock_probe(orbfe_handle* h, int spin_us, unsigned long long* d_out, void* stream)
{
    if (!h || !d_out || spin_us < 1 || spin_us > 100000) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    launch_clock_probe(stream ? static_cast<hipStream_t>(stream) : h->stream, d_out, (unsigned)spin_us * 100u);
    HIPCHK(h, hipGetLastError());
    return ORBFE_OK;
}

int orbfe_debug_graph_stats(orbfe_handle* h, int* captured, int* failed)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (captured) *captured = h->graphCaptures;
    if (failed) *failed = h->captureFailures;
    return ORBFE_OK;
}

int orbfe_get_device_status(orbfe_handle* h, unsigned* flags_out)
{
    if (!h) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    unsigned flags = 0;
    if (h->extractUsed && h->lastBatch > 0) {
        HIPCHK(h, hipEventSynchronize(h->evExtract));
        std::vector<uint32_t> c((size_t)h->lastBatch * h->nLevels * kCntWords);
        HIPCHK(h, copy_sync(c.data(), h->dCounters, c.size() * sizeof(uint32_t), hipMemcpyDeviceToHost, h->stream));
        for (size_t i = 0; i < (size_t)h->lastBatch * h->nLevels; i++) flags |= c[i * kCntWords + kCntStatus];
    }
    if (flags_out) *flags_out = flags;
    if (flags) {
        char buf[96];
        snprintf(buf, sizeof buf, "device guard flags 0x%x in the last extract call", flags);
        h->err = buf;
        return ORBFE_ERR_INTERNAL;
    }
    return ORBFE_OK;
}

int orbfe_hamming(const uint8_t* a, const uint8_t* b)
{
    // ORBmatcher::DescriptorDistance (src/ORBmatcher.cc:1375-1391): popcount of the 256-bit XOR
    int d = 0;
    for (int i = 0; i < 4; i++) {
        uint64_t x, y;
        memcpy(&x, a + 8 * i, 8);
        memcpy(&y, b + 8 * i, 8);
        d += __builtin_popcountll(x ^ y);
    }
    return d;
}

int orbfe_match_projection(orbfe_handle* h, const orbfe_frame_view* F, int M, const orbfe_map_point* mps,
                           const uint8_t* mp_desc, const int* init_obs, float th, int far_points, float th_far,
                           float nn_ratio, int* match_out, int* n_matches)
{
    if (!h || !F || !match_out || !n_matches || M < 0 || F->n < 0) return ORBFE_ERR_INVALID_ARG;
    if ((M > 0 && (!mps || !mp_desc)) || (F->n > 0 && (!F->kp || !F->desc)) || !F->scale_factors) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    int rc = match_projection_run(h->match, h->stream, F, M, mps, mp_desc, init_obs, th, far_points, th_far, nn_ratio,
                                  match_out, n_matches, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_match_projection_batch_device(orbfe_handle* h, int batch, const orbfe_keypoint* d_kp, const uint8_t* d_desc,
                                        const int* d_n, int kp_stride, int grid_cols, int grid_rows, float min_x,
                                        float min_y, float inv_w, float inv_h, int M, const orbfe_map_point* d_mps,
                                        const uint8_t* d_mp_desc, const int* d_init_obs, float th, int far_points,
                                        float th_far, float nn_ratio, int* d_match_out, int* d_n_matches, void* stream_)
{
    if (!h || !d_kp || !d_desc || !d_n || !d_match_out || !d_n_matches || batch < 1 || M < 0 || kp_stride < 1 ||
        grid_cols < 1 || grid_rows < 1)
        return ORBFE_ERR_INVALID_ARG;
    if (M > 0 && (!d_mps || !d_mp_desc)) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = stream_ ? (hipStream_t)stream_ : h->stream;
    std::string err;
    MatchScope scope_(h, s);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    int rc = match_projection_batch_device(h->match, s, batch, d_kp, d_desc, d_n, kp_stride, grid_cols, grid_rows, min_x,
                                           min_y, inv_w, inv_h, h->dSf, h->nLevels, M, d_mps, d_mp_desc, d_init_obs, th,
                                           far_points, th_far, nn_ratio, d_match_out, d_n_matches, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_match_initialization(orbfe_handle* h, const orbfe_frame_view* F1, const orbfe_frame_view* F2, int window_size,
                               float nn_ratio, int check_orientation, int* matches12_out, int* n_matches)
{
    if (!h || !F1 || !F2 || !matches12_out || !n_matches || F1->n < 0 || F2->n < 0 || window_size < 0)
        return ORBFE_ERR_INVALID_ARG;
    if ((F1->n > 0 && (!F1->kp || !F1->desc)) || (F2->n > 0 && (!F2->kp || !F2->desc)) || F2->grid_cols < 1 || F2->grid_rows < 1)
        return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    int rc = match_initialization_run(h->match, h->stream, F1, F2, window_size, nn_ratio, check_orientation, matches12_out,
                                      n_matches, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_project_map_points_device(orbfe_handle* h, const orbfe_frustum* frustum, int n, const orbfe_world_point* d_points,
                                    orbfe_map_point* d_out, float* d_proj_xr, void* stream)
{
    if (!h || n < 0 || (n > 0 && (!d_points || !d_out))) return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    const int rc = frustum_launch(stream ? static_cast<hipStream_t>(stream) : h->stream, frustum, n, d_points, d_out, d_proj_xr, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_project_map_points(orbfe_handle* h, const orbfe_frustum* frustum, int n, const orbfe_world_point* points,
                             orbfe_map_point* out, float* proj_xr)
{
    if (!h || n < 0 || (n > 0 && (!points || !out))) return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    if (n == 0) return ORBFE_OK;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    // staging through the matcher's grow-only arenas: [points | out | xr] on both sides
    const size_t bIn = (size_t)n * sizeof(orbfe_world_point), bOut = (size_t)n * sizeof(orbfe_map_point), bXr = (size_t)n * sizeof(float);
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    int rc = ensure(h->match, bIn + bOut + bXr + 256, bIn + bOut + bXr + 256, err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    uint8_t* dp = static_cast<uint8_t*>(h->match.d);
    uint8_t* hp = static_cast<uint8_t*>(h->match.hpin);
    memcpy(hp, points, bIn);
    HIPCHK(h, hipMemcpyAsync(dp, hp, bIn, hipMemcpyHostToDevice, h->stream));
    rc = frustum_launch(h->stream, frustum, n, reinterpret_cast<const orbfe_world_point*>(dp),
                        reinterpret_cast<orbfe_map_point*>(dp + bIn), reinterpret_cast<float*>(dp + bIn + bOut), err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    HIPCHK(h, hipMemcpyAsync(hp + bIn, dp + bIn, bOut + bXr, hipMemcpyDeviceToHost, h->stream));
    HIPCHK(h, hipStreamSynchronize(h->stream));
    memcpy(out, hp + bIn, bOut);
    if (proj_xr) memcpy(proj_xr, hp + bIn + bOut, bXr);
    return ORBFE_OK;
}

int orbfe_fuse_search(orbfe_handle* h, const orbfe_frame_view* KF, const float* inv_level_sigma2, const float* u_right,
                      const orbfe_frustum* frustum, float th, int M, const orbfe_world_point* points, const uint8_t* mp_desc,
                      int* best_idx_out, int* best_dist_out)
{
    if (!h || !KF || !inv_level_sigma2 || M < 0 || KF->n < 0 || (KF->n > 0 && (!KF->kp || !KF->desc || !KF->scale_factors)) ||
        (M > 0 && (!points || !mp_desc || !best_idx_out || !best_dist_out)))
        return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = fuse_search_run(h->match, h->stream, KF, inv_level_sigma2, u_right, frustum, th, M, points, mp_desc, 1, -1,
                                   best_idx_out, best_dist_out, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_fuse_search_right(orbfe_handle* h, const orbfe_frame_view* KF, int n_right, const float* inv_level_sigma2,
                            const float* u_right, const orbfe_frustum* frustum, float th, int M, const orbfe_world_point* points,
                            const uint8_t* mp_desc, int* best_idx_out, int* best_dist_out)
{
    if (!h || !KF || !inv_level_sigma2 || M < 0 || KF->n < 0 || n_right < 0 ||
        (KF->n > 0 && (!KF->kp || !KF->desc || !KF->scale_factors)) ||
        (M > 0 && (!points || !mp_desc || !best_idx_out || !best_dist_out)))
        return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = fuse_search_run(h->match, h->stream, KF, inv_level_sigma2, u_right, frustum, th, M, points, mp_desc, 1,
                                   n_right, best_idx_out, best_dist_out, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

static bool view_args_bad(const orbfe_frame_view* V)
{
    return !V || V->n < 0 || (V->n > 0 && (!V->kp || !V->desc || !V->scale_factors));
}

int orbfe_fuse_search_sim3(orbfe_handle* h, const orbfe_frame_view* KF, const orbfe_frustum* frustum, float th, int M,
                           const orbfe_world_point* points, const uint8_t* mp_desc, int* best_idx_out, int* best_dist_out)
{
    if (!h || view_args_bad(KF) || M < 0 || (M > 0 && (!points || !mp_desc || !best_idx_out || !best_dist_out)))
        return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = fuse_search_run(h->match, h->stream, KF, nullptr, nullptr, frustum, th, M, points, mp_desc, 0, -1,
                                   best_idx_out, best_dist_out, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

static bool sim3_view_bad(const orbfe_sim3_view* D)
{
    return !D || D->n_levels < 1 || !(D->log_scale_factor > 0.0f);
}

int orbfe_search_by_sim3(orbfe_handle* h, const orbfe_frame_view* KF1, const orbfe_frame_view* KF2,
                         const orbfe_sim3_view* dir12, const orbfe_sim3_view* dir21, const orbfe_world_point* mp1,
                         const uint8_t* mp_desc1, const orbfe_world_point* mp2, const uint8_t* mp_desc2, float th,
                         int* match12_out, int* n_found)
{
    if (!h || view_args_bad(KF1) || view_args_bad(KF2) || sim3_view_bad(dir12) || sim3_view_bad(dir21) || !n_found ||
        (KF1->n > 0 && (!mp1 || !mp_desc1 || !match12_out)) || (KF2->n > 0 && (!mp2 || !mp_desc2)))
        return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = search_by_sim3_run(h->match, h->stream, KF1, KF2, dir12, dir21, mp1, mp_desc1, mp2, mp_desc2, th,
                                      match12_out, n_found, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_match_projection_keyframe(orbfe_handle* h, const orbfe_frame_view* frame, const orbfe_frustum* frustum,
                                    int n_points, const orbfe_world_point* points, const uint8_t* mp_desc,
                                    const float* kf_angle, const uint8_t* frame_has_mp, float th, int check_orientation,
                                    int* match_out, int* n_matches)
{
    if (!h || view_args_bad(frame) || n_points < 0 || !n_matches || (frame->n > 0 && !match_out) ||
        (n_points > 0 && (!points || !mp_desc || (check_orientation && !kf_angle))))
        return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = match_projection_kf_run(h->match, h->stream, frame, frustum, n_points, points, mp_desc, kf_angle,
                                           frame_has_mp, th, check_orientation, match_out, n_matches, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_match_triangulation(orbfe_handle* h, int n_groups, const int* kf1_off, const int* kf1_idx, const int* kf2_off,
                              const int* kf2_idx, int n1, const orbfe_keypoint* kp1, const uint8_t* desc1,
                              const uint8_t* has_mp1, const uint8_t* stereo1, int n2, const orbfe_keypoint* kp2,
                              const uint8_t* desc2, const uint8_t* has_mp2, const uint8_t* stereo2,
                              const float* scale_factors2, int n_levels2, const orbfe_tri_params* params,
                              int* matches12_out, int* n_matches)
{
    if (!h || !params || !n_matches || n_groups < 0 || n1 < 0 || n2 < 0 || n_levels2 < 1 || !scale_factors2 ||
        (n_groups > 0 && (!kf1_off || !kf2_off)) || (n1 > 0 && (!kp1 || !desc1 || !has_mp1 || !matches12_out)) ||
        (n2 > 0 && (!kp2 || !desc2 || !has_mp2)))
        return ORBFE_ERR_INVALID_ARG;
    if (n_groups > 0 && ((kf1_off[n_groups] > 0 && !kf1_idx) || (kf2_off[n_groups] > 0 && !kf2_idx))) return ORBFE_ERR_INVALID_ARG;
    if (params->struct_size != (int)sizeof(orbfe_tri_params)) {  // a caller built against another layout of the block
        std::lock_guard<std::mutex> lk(h->mu);
        h->err = "orbfe_tri_params.struct_size does not match this library (rebuild the caller against include/orbfe.h)";
        return ORBFE_ERR_INVALID_ARG;
    }
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = match_triangulation_run(h->match, h->stream, n_groups, kf1_off, kf1_idx, kf2_off, kf2_idx, n1, kp1, desc1,
                                           has_mp1, stereo1, n2, kp2, desc2, has_mp2, stereo2, scale_factors2, n_levels2,
                                           params, matches12_out, n_matches, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

struct orbfe_keyframe {
    orbfe::KeyFrameDev* k;
    int device;
};

int orbfe_keyframe_create(orbfe_handle* h, int n, const orbfe_keypoint* kp, const uint8_t* desc, const int* node_id,
                          const uint8_t* stereo, const float* scale_factors, int n_levels, orbfe_keyframe** out)
{
    if (!h || !out) return ORBFE_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    orbfe::KeyFrameDev* k = nullptr;
    const int rc = keyframe_create(n, kp, desc, node_id, stereo, scale_factors, n_levels, h->stream, &k, err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    *out = new orbfe_keyframe{k, h->device};
    return ORBFE_OK;
}

void orbfe_keyframe_destroy(orbfe_keyframe* kf)
{
    if (!kf) return;
    (void)hipSetDevice(kf->device);
    keyframe_destroy(kf->k);
    delete kf;
}

int orbfe_keyframe_size(const orbfe_keyframe* kf) { return kf ? kf->k->n : 0; }

int orbfe_keyframe_set_grid(orbfe_handle* h, orbfe_keyframe* kf, int grid_cols, int grid_rows, float min_x, float min_y,
                            float grid_inv_w, float grid_inv_h, const float* inv_level_sigma2, const float* u_right)
{
    if (!h || !kf || kf->device != h->device || !inv_level_sigma2 || grid_cols < 1 || grid_rows < 1) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    const int rc = keyframe_set_grid(kf->k, h->stream, grid_cols, grid_rows, min_x, min_y, grid_inv_w, grid_inv_h, inv_level_sigma2,
                                     u_right, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_fuse_search_keyframe(orbfe_handle* h, const orbfe_keyframe* kf, const orbfe_map* map, int M, const int* ids,
                               const orbfe_frustum* frustum, float th, int* best_idx_out, int* best_dist_out)
{
    if (!h || !kf || !map || kf->device != h->device || map->h != h || M < 0 || (M > 0 && (!ids || !best_idx_out || !best_dist_out)))
        return ORBFE_ERR_INVALID_ARG;
    const int rc0 = frustum_validate(frustum);
    if (rc0 != ORBFE_OK) return rc0;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);  // also orders the call behind an orbfe_map_update of another stream's making
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = fuse_search_keyframe_run(h->match, h->stream, kf->k, map->cap, map->dPts, map->dDesc, M, ids, frustum, th,
                                            best_idx_out, best_dist_out, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_match_triangulation_batch(orbfe_handle* h, const orbfe_keyframe* kf1, const uint8_t* has_mp1, int K,
                                    const orbfe_keyframe* const* kf2, const uint8_t* const* has_mp2,
                                    const orbfe_tri_params* params, int* raw_match12, uint8_t* raw_bin)
{
    if (!h || !kf1 || K < 0 || K > 4096 || (K > 0 && (!kf2 || !has_mp2 || !params)) ||
        (kf1->k->n > 0 && K > 0 && (!has_mp1 || !raw_match12 || !raw_bin)))
        return ORBFE_ERR_INVALID_ARG;
    std::vector<const orbfe::KeyFrameDev*> k2((size_t)K);
    for (int k = 0; k < K; k++) {
        if (!kf2[k] || kf2[k]->device != h->device) return ORBFE_ERR_INVALID_ARG;
        k2[(size_t)k] = kf2[k]->k;
    }
    if (kf1->device != h->device) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = match_triangulation_batch_run(h->match, h->stream, kf1->k, has_mp1, K, k2.data(), has_mp2, params, raw_match12,
                                                 raw_bin, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_triangulation_select(int n1, const int* raw_match12, const uint8_t* raw_bin, const uint8_t* has_mp1_now,
                               int check_orientation, int* matches12_out, int* n_matches)
{
    if (n1 < 0 || !n_matches || (n1 > 0 && (!raw_match12 || !raw_bin || !has_mp1_now || !matches12_out))) return ORBFE_ERR_INVALID_ARG;
    int hist[ORBFE_HISTO_LENGTH] = {0};
    int n = 0;
    for (int i = 0; i < n1; i++) {
        int m = has_mp1_now[i] ? -1 : raw_match12[i];  // src/ORBmatcher.cc:506-509
        if (m >= 0) {
            if (check_orientation && raw_bin[i] >= ORBFE_HISTO_LENGTH) return ORBFE_ERR_INVALID_ARG;
            n++;
            if (check_orientation) hist[raw_bin[i]]++;
        }
        matches12_out[i] = m;
    }
    if (check_orientation) {  // :633-661 with ComputeThreeMaxima :1328-1370
        int max1 = 0, max2 = 0, max3 = 0, ind1 = -1, ind2 = -1, ind3 = -1;
        for (int i = 0; i < ORBFE_HISTO_LENGTH; i++) {
            const int s = hist[i];
            if (s > max1) { max3 = max2; max2 = max1; max1 = s; ind3 = ind2; ind2 = ind1; ind1 = i; }
            else if (s > max2) { max3 = max2; max2 = s; ind3 = ind2; ind2 = i; }
            else if (s > max3) { max3 = s; ind3 = i; }
        }
        if ((float)max2 < 0.1f * (float)max1) { ind2 = -1; ind3 = -1; }
        else if ((float)max3 < 0.1f * (float)max1) { ind3 = -1; }
        for (int i = 0; i < n1; i++)
            if (matches12_out[i] >= 0) {
                const int b = raw_bin[i];
                if (b != ind1 && b != ind2 && b != ind3) {
                    matches12_out[i] = -1;
                    n--;
                }
            }
    }
    *n_matches = n;
    return ORBFE_OK;
}

int orbfe_distinctive_descriptors(orbfe_handle* h, int n_sets, const int* set_off, const uint8_t* desc, int* best_idx_out,
                                  int* best_median_out)
{
    if (!h || n_sets < 0 || (n_sets > 0 && (!set_off || !best_idx_out || set_off[0] != 0)) ||
        (n_sets > 0 && set_off[n_sets] > 0 && !desc))
        return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    const int rc = distinctive_run(h->match, h->stream, n_sets, set_off, desc, best_idx_out, best_median_out, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

// ---------------------------------------------------------------------------------------------
// Node-side image preparation (image_grabber.hpp:96-110): kernels_prep.hip
// ---------------------------------------------------------------------------------------------
struct orbfe_prep {
    int device = 0;
    int srcW = 0, srcH = 0, dstW = 0, dstH = 0;
    float* dMap1 = nullptr;
    float* dMap2 = nullptr;
    uint8_t* dSrc = nullptr;   // srcH x srcPitch, BGR
    uint8_t* hSrc = nullptr;   // pinned staging for pageable sources
    int srcPitch = 0;
    uint8_t* dGray = nullptr;  // dstH x grayPitch
    uint8_t* hGray = nullptr;  // pinned
    int grayPitch = 0;
};

void orbfe_prep_destroy(orbfe_prep* p)
{
    if (!p) return;
    (void)hipSetDevice(p->device);
    if (p->dMap1) (void)hipFree(p->dMap1);
    if (p->dMap2) (void)hipFree(p->dMap2);
    if (p->dSrc) (void)hipFree(p->dSrc);
    if (p->dGray) (void)hipFree(p->dGray);
    if (p->hSrc) (void)hipHostFree(p->hSrc);
    if (p->hGray) (void)hipHostFree(p->hGray);
    delete p;
}

int orbfe_prep_create(orbfe_handle* h, int src_w, int src_h, const float* map1, const float* map2, int dst_w, int dst_h,
                      orbfe_prep** out)
{
    if (!h || !map1 || !map2 || !out || src_w < 2 || src_h < 2 || dst_w < 1 || dst_h < 1 || src_w > 16384 || src_h > 16384 ||
        dst_w > 16384 || dst_h > 16384)
        return ORBFE_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    orbfe_prep* p = new (std::nothrow) orbfe_prep();
    if (!p) return ORBFE_ERR_OUT_OF_MEMORY;
    p->device = h->device;
    p->srcW = src_w; p->srcH = src_h; p->dstW = dst_w; p->dstH = dst_h;
    p->srcPitch = (int)align_up((size_t)src_w * 3, 256);
    p->grayPitch = (int)align_up((size_t)dst_w, 256);
    const size_t mapBytes = (size_t)src_w * src_h * sizeof(float);
    bool ok = hipMalloc(&p->dMap1, mapBytes) == hipSuccess && hipMalloc(&p->dMap2, mapBytes) == hipSuccess &&
              hipMalloc(&p->dSrc, (size_t)p->srcPitch * src_h) == hipSuccess &&
              hipHostMalloc(&p->hSrc, (size_t)p->srcPitch * src_h) == hipSuccess &&
              hipMalloc(&p->dGray, (size_t)p->grayPitch * dst_h) == hipSuccess &&
              hipHostMalloc(&p->hGray, (size_t)p->grayPitch * dst_h) == hipSuccess;
    ok = ok && copy_sync(p->dMap1, map1, mapBytes, hipMemcpyHostToDevice, h->stream) == hipSuccess &&
         copy_sync(p->dMap2, map2, mapBytes, hipMemcpyHostToDevice, h->stream) == hipSuccess;
    if (!ok) {
        (void)hipGetLastError();
        orbfe_prep_destroy(p);
        h->err = "orbfe_prep_create: allocation or map upload failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    *out = p;
    return ORBFE_OK;
}

static orbfe::PrepArgs prep_args(const orbfe_prep* p, const uint8_t* dSrc, int srcPitch, uint8_t* dDst, int dstPitch)
{
    orbfe::PrepArgs P{};
    P.src = dSrc; P.srcPitch = srcPitch; P.srcFrameStride = 0;
    P.srcW = p->srcW; P.srcH = p->srcH;
    P.map1 = p->dMap1; P.map2 = p->dMap2;
    P.dst = dDst; P.dstPitch = dstPitch; P.dstFrameStride = 0;
    P.dstW = p->dstW; P.dstH = p->dstH;
    P.fx = orbfe::prep_scale(p->srcW, p->dstW);
    P.fy = orbfe::prep_scale(p->srcH, p->dstH);
    return P;
}

// BGR frame -> p->dSrc on stream s (pinned sources go straight through the DMA engine with their own pitch)
static int prep_upload(orbfe_handle* h, orbfe_prep* p, const uint8_t* bgr, int pitch, hipStream_t s, int* devPitch)
{
    const size_t rowBytes = (size_t)p->srcW * 3;
    hipPointerAttribute_t attr;
    const bool pinned = pitch <= p->srcPitch && hipPointerGetAttributes(&attr, bgr) == hipSuccess && attr.type == hipMemoryTypeHost;
    if (!pinned) (void)hipGetLastError();
    if (pinned) {
        HIPCHK(h, hipMemcpyAsync(p->dSrc, bgr, (size_t)pitch * (p->srcH - 1) + rowBytes, hipMemcpyHostToDevice, s));
        *devPitch = pitch;
    } else {
        for (int y = 0; y < p->srcH; y++) memcpy(p->hSrc + (size_t)y * p->srcPitch, bgr + (size_t)y * pitch, rowBytes);
        HIPCHK(h, hipMemcpyAsync(p->dSrc, p->hSrc, (size_t)p->srcPitch * p->srcH, hipMemcpyHostToDevice, s));
        *devPitch = p->srcPitch;
    }
    return ORBFE_OK;
}

int orbfe_prepare_image_device(orbfe_handle* h, orbfe_prep* p, const uint8_t* d_bgr, int pitch, uint8_t* d_gray, int gray_pitch,
                               void* stream)
{
    if (!h || !p || !d_bgr || !d_gray || pitch < p->srcW * 3 || gray_pitch < p->dstW) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = stream ? static_cast<hipStream_t>(stream) : h->stream;
    orbfe::prep_launch(s, prep_args(p, d_bgr, pitch, d_gray, gray_pitch), 1);
    HIPCHK(h, hipGetLastError());
    return ORBFE_OK;
}

int orbfe_prepare_image(orbfe_handle* h, orbfe_prep* p, const uint8_t* bgr, int pitch, uint8_t* gray_out, int gray_pitch)
{
    if (!h || !p || !bgr || !gray_out || pitch < p->srcW * 3 || gray_pitch < p->dstW) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int devPitch = 0;
    const int rc = prep_upload(h, p, bgr, pitch, s, &devPitch);
    if (rc != ORBFE_OK) return rc;
    orbfe::prep_launch(s, prep_args(p, p->dSrc, devPitch, p->dGray, p->grayPitch), 1);
    HIPCHK(h, hipGetLastError());
    HIPCHK(h, hipMemcpyAsync(p->hGray, p->dGray, (size_t)p->grayPitch * p->dstH, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    for (int y = 0; y < p->dstH; y++) memcpy(gray_out + (size_t)y * gray_pitch, p->hGray + (size_t)y * p->grayPitch, (size_t)p->dstW);
    return ORBFE_OK;
}

int orbfe_prepare_and_extract(orbfe_handle* h, orbfe_prep* p, const uint8_t* bgr, int pitch, orbfe_keypoint* kp_out,
                              uint8_t* desc_out, int* n_out, int* per_level, uint8_t* gray_out, int gray_pitch)
{
    if (!h || !p || !bgr || !kp_out || !desc_out || !n_out || pitch < p->srcW * 3 || (gray_out && gray_pitch < p->dstW))
        return ORBFE_ERR_INVALID_ARG;
    if (p->dstW != h->prm.image_width || p->dstH != h->prm.image_height) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    int devPitch = 0;
    int rc = prep_upload(h, p, bgr, pitch, s, &devPitch);
    if (rc != ORBFE_OK) return rc;
    // the grey frame is written straight into the extractor's level-0 input rows: no host round trip in between
    orbfe::prep_launch(s, prep_args(p, p->dSrc, devPitch, h->dIn, h->dInPitch), 1);
    HIPCHK(h, hipGetLastError());
    rc = extract_enqueue_replay(h, 1, h->dInPitch, s);
    if (rc != ORBFE_OK) return rc;
    if (gray_out)
        HIPCHK(h, hipMemcpy2DAsync(p->hGray, p->grayPitch, h->dIn, h->dInPitch, p->dstW, p->dstH, hipMemcpyDeviceToHost, s));
    HIPCHK(h, hipStreamSynchronize(s));
    rc = check_device_flags(h, 1);
    if (rc != ORBFE_OK) return rc;
    const int n = h->hN[0];
    n_out[0] = n;
    memcpy(kp_out, h->hKp, (size_t)n * sizeof(orbfe_keypoint));
    memcpy(desc_out, h->hDesc, (size_t)n * ORBFE_DESC_BYTES);
    if (per_level) memcpy(per_level, h->hPer, h->nLevels * sizeof(int));
    if (gray_out)
        for (int y = 0; y < p->dstH; y++) memcpy(gray_out + (size_t)y * gray_pitch, p->hGray + (size_t)y * p->grayPitch, (size_t)p->dstW);
    return ORBFE_OK;
}

struct orbfe_vocab {
    orbfe::Vocab* v;
    int device;
    unsigned long long serial;  // names the vocabulary in graph keys (an address could be reused after a destroy)
};
static std::atomic<unsigned long long> g_vocabSerial{1};

int orbfe_vocab_create(orbfe_handle* h, int n_nodes, const int* child_off, const int* child_idx, const uint8_t* node_desc,
                       const int* word_id, const double* weight, int L, orbfe_vocab** out)
{
    if (!h || !child_off || !child_idx || !node_desc || !word_id || !weight || !out || n_nodes < 2 || L < 1)
        return ORBFE_ERR_INVALID_ARG;
    *out = nullptr;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    orbfe::Vocab* v = nullptr;
    const int rc = vocab_create(n_nodes, child_off, child_idx, node_desc, word_id, weight, L, h->stream, &v, err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    *out = new orbfe_vocab{v, h->device, g_vocabSerial.fetch_add(1)};
    return ORBFE_OK;
}

void orbfe_vocab_destroy(orbfe_vocab* v)
{
    if (!v) return;
    (void)hipSetDevice(v->device);
    vocab_destroy(v->v);
    delete v;
}

int orbfe_bow_transform(orbfe_handle* h, orbfe_vocab* v, const uint8_t* desc, int n, int levelsup, int* word_id_out,
                        int* node_id_out, double* weight_out)
{
    if (!h || !v || n < 0 || (n > 0 && (!desc || !word_id_out || !node_id_out))) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    const int rc = vocab_transform(v->v, h->stream, desc, n, levelsup, word_id_out, node_id_out, weight_out, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_match_bow_rig(orbfe_handle* h, int G, const int* kf_off, const int* kf_idx, const int* f_off, const int* f_idx,
                        int n_kf, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_has_mp, int n_f,
                        const uint8_t* f_desc, const float* f_angle, int n_left, float nn_ratio, int check_orientation,
                        int* match_out, int* n_matches)
{
    if (!h || !match_out || !n_matches || G < 0 || n_kf < 0 || n_f < 0 || n_left < -1 || n_left > n_f) return ORBFE_ERR_INVALID_ARG;
    if (G > 0 && (!kf_off || !kf_idx || !f_off || !f_idx || !kf_desc || !f_desc || !kf_has_mp)) return ORBFE_ERR_INVALID_ARG;
    if (check_orientation && G > 0 && (!kf_angle || !f_angle)) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::string err;
    MatchScope scope_(h, h->stream);
    if (scope_.rc != ORBFE_OK) return scope_.rc;
    int rc = match_bow_run(h->match, h->stream, G, kf_off, kf_idx, f_off, f_idx, n_kf, kf_desc, kf_angle, kf_has_mp, n_f,
                           f_desc, f_angle, n_left, nn_ratio, check_orientation, match_out, n_matches, err);
    if (rc != ORBFE_OK) h->err = err;
    return rc;
}

int orbfe_match_bow(orbfe_handle* h, int G, const int* kf_off, const int* kf_idx, const int* f_off, const int* f_idx,
                    int n_kf, const uint8_t* kf_desc, const float* kf_angle, const uint8_t* kf_has_mp, int n_f,
                    const uint8_t* f_desc, const float* f_angle, float nn_ratio, int check_orientation, int* match_out,
                    int* n_matches)
{
    return orbfe_match_bow_rig(h, G, kf_off, kf_idx, f_off, f_idx, n_kf, kf_desc, kf_angle, kf_has_mp, n_f, f_desc, f_angle, -1,
                               nn_ratio, check_orientation, match_out, n_matches);
}

}  // extern "C"

// ---------------------------------------------------------------------------------------------
// orbfe_track_reference_keyframe: extract -> vocabulary descent -> SearchByBoW(resident key frame) as one captured hipGraph
// ---------------------------------------------------------------------------------------------
namespace {

struct RefLayout {
    size_t inFrame, oRef, oFlags, inBytes;
    size_t oPer, oKp, oDesc, oMatch, oBow, oLeaf, outBytes /* what is downloaded */, oBin, devBytes;
};

RefLayout ref_layout(const orbfe_handle* h, int capFlags)
{
    RefLayout L{};
    const size_t cap = (size_t)h->P.kpCapFrame;
    L.inFrame = align_up((size_t)h->dInPitch * h->prm.image_height, 256);
    L.oRef = L.inFrame;
    L.oFlags = L.oRef + align_up(sizeof(BowKfRef), 256);
    L.inBytes = L.oFlags + align_up((size_t)capFlags, 256);
    size_t off = 256;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.oPer = take((size_t)h->nLevels * sizeof(int));
    L.oKp = take(cap * sizeof(orbfe_keypoint));
    L.oDesc = take(cap * ORBFE_DESC_BYTES);
    L.oMatch = take(cap * sizeof(int));
    L.oBow = take(cap * 2 * sizeof(int));
    L.oLeaf = take(cap * sizeof(int));
    L.outBytes = off;
    L.oBin = take(cap * sizeof(int));
    L.devBytes = off;
    return L;
}

void ref_drop_graphs(orbfe_handle* h)
{
    for (auto& g : h->refGraphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->refGraphs.clear();
}

int ref_reserve(orbfe_handle* h, int nFlags)
{
    if (h->dRefIn && nFlags <= h->refCapFlags) return ORBFE_OK;
    ref_drop_graphs(h);
    if (h->dRefIn) (void)hipFree(h->dRefIn);
    if (h->dRefOut) (void)hipFree(h->dRefOut);
    if (h->hRefIn) (void)hipHostFree(h->hRefIn);
    if (h->hRefOut) (void)hipHostFree(h->hRefOut);
    h->dRefIn = h->dRefOut = h->hRefIn = h->hRefOut = nullptr;
    h->refCapFlags = 0;
    const int capFlags = std::max(4096, nFlags + nFlags / 2);
    const RefLayout L = ref_layout(h, capFlags);
    if (hipMalloc(&h->dRefIn, L.inBytes) != hipSuccess || hipMalloc(&h->dRefOut, L.devBytes) != hipSuccess ||
        hipHostMalloc(&h->hRefIn, L.inBytes) != hipSuccess || hipHostMalloc(&h->hRefOut, L.outBytes) != hipSuccess) {
        (void)hipGetLastError();
        h->err = "orbfe_track_reference_keyframe: allocation of the staging blocks failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    h->refCapFlags = capFlags;
    return ORBFE_OK;
}

// the device side of one call, enqueued on s (directly, or under stream capture)
int ref_enqueue(orbfe_handle* h, const RefLayout& L, int inPitch, const orbfe::Vocab* v, int levelsup, float nnRatio, int checkOri,
                hipStream_t s)
{
    const int cap = h->P.kpCapFrame;
    int* dHead = reinterpret_cast<int*>(h->dRefOut);  // [n, status, n_matches]
    orbfe_keypoint* dKp = reinterpret_cast<orbfe_keypoint*>(h->dRefOut + L.oKp);
    int rc = extract_chain(h, h->dRefIn, L.inFrame, inPitch, 1, dKp, h->dRefOut + L.oDesc, dHead, reinterpret_cast<int*>(h->dRefOut + L.oPer),
                           dHead + 1, s);
    if (rc != ORBFE_OK) return rc;
    std::string err;
    int* dBow = reinterpret_cast<int*>(h->dRefOut + L.oBow);
    rc = vocab_transform_launch_dev(v, s, h->dRefOut + L.oDesc, dHead, cap, levelsup, dBow, reinterpret_cast<int*>(h->dRefOut + L.oLeaf),
                                    reinterpret_cast<int*>(h->dRefOut + L.oMatch), err);
    if (rc == ORBFE_OK) {
        BowTrackArgs A{};
        A.ref = reinterpret_cast<const BowKfRef*>(h->dRefIn + L.oRef);
        A.kfHasMP = h->dRefIn + L.oFlags;
        A.fKp = dKp;
        A.fDesc = h->dRefOut + L.oDesc;
        A.fBow = dBow;
        A.fLeaf = reinterpret_cast<const int*>(h->dRefOut + L.oLeaf);
        A.weight = v->dWeight;
        A.nF = dHead;
        A.cap = cap;
        A.nnRatio = nnRatio;
        A.checkOrientation = checkOri;
        A.matchOut = reinterpret_cast<int*>(h->dRefOut + L.oMatch);
        A.binOf = reinterpret_cast<int*>(h->dRefOut + L.oBin);
        A.nMatches = dHead + 2;
        rc = bow_track_launch(s, A, err);
    }
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    HIPCHK(h, hipMemcpyAsync(h->hRefOut, h->dRefOut, L.outBytes, hipMemcpyDeviceToHost, s));
    return ORBFE_OK;
}

}  // namespace

extern "C" int orbfe_track_reference_keyframe(orbfe_handle* h, const uint8_t* gray, int pitch, const orbfe_vocab* vocab, int levelsup,
                                              const orbfe_keyframe* kf, const uint8_t* kf_has_mp, float nn_ratio,
                                              int check_orientation, orbfe_keypoint* kp_out, uint8_t* desc_out, int* n_out,
                                              int* per_level, int* word_id_out, int* node_id_out, double* weight_out,
                                              int* match_out, int* n_matches)
{
    if (!h || !gray || !vocab || !kf || !kp_out || !desc_out || !n_out || !word_id_out || !node_id_out || !match_out || !n_matches)
        return ORBFE_ERR_INVALID_ARG;
    if (pitch < h->prm.image_width || pitch >= (1 << 24)) return ORBFE_ERR_INVALID_ARG;
    if (vocab->device != h->device || kf->device != h->device) return ORBFE_ERR_INVALID_ARG;
    const KeyFrameDev* K = kf->k;
    if (K->n > 0 && !kf_has_mp) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (h->P.kpCapFrame > 7168) return ORBFE_ERR_UNSUPPORTED;  // the matcher's per-frame LDS arrays (bow_track_launch)
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int W = h->prm.image_width, H = h->prm.image_height, nL = h->nLevels;
    int rc = ref_reserve(h, K->n);
    if (rc != ORBFE_OK) return rc;
    const RefLayout L = ref_layout(h, h->refCapFlags);

    // ---- the small block: which key frame (addresses of its resident arrays) + its flags as they stand now ----
    BowKfRef R{K->desc, K->kp, K->order, K->nodeList, K->nodeOff, K->G, K->n};
    memcpy(h->hRefIn + L.oRef, &R, sizeof R);
    if (K->n) memcpy(h->hRefIn + L.oFlags, kf_has_mp, (size_t)K->n);
    const size_t smallEnd = L.oFlags + (size_t)K->n;

    // ---- upload (as orbfe_track_frame): pinned frames straight from the caller's buffer, pageable ones through the mirror ----
    bool direct = pitch <= h->dInPitch && (pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(gray) & 3u) == 0;
    if (direct) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, gray) != hipSuccess || attr.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            direct = false;
        }
    }
    int inPitch;
    if (direct) {
        inPitch = pitch;
        HIPCHK(h, hipMemcpyAsync(h->dRefIn, gray, (size_t)pitch * (H - 1) + (size_t)W, hipMemcpyHostToDevice, s));
        HIPCHK(h, hipMemcpyAsync(h->dRefIn + L.oRef, h->hRefIn + L.oRef, smallEnd - L.oRef, hipMemcpyHostToDevice, s));
    } else {
        if ((pitch & 3) == 0 && pitch <= h->dInPitch) {
            inPitch = pitch;
            memcpy(h->hRefIn, gray, (size_t)pitch * (H - 1) + (size_t)W);
        } else {
            inPitch = h->dInPitch;
            for (int y = 0; y < H; y++) memcpy(h->hRefIn + (size_t)y * inPitch, gray + (size_t)y * pitch, (size_t)W);
        }
        HIPCHK(h, hipMemcpyAsync(h->dRefIn, h->hRefIn, smallEnd, hipMemcpyHostToDevice, s));
    }

    // ---- kernels + download: replay the graph of this (pitch, vocabulary, parameters), capturing it first if needed ----
    bool viaGraph = h->useGraph && !h->timing;
    if (viaGraph) {
        orbfe_handle::RefKey key;
        memset(&key, 0, sizeof key);
        key.inPitch = inPitch; key.levelsup = levelsup; key.checkOri = check_orientation; key.nnRatio = nn_ratio;
        key.vocabSerial = vocab->serial;
        hipGraphExec_t exec = nullptr;
        for (auto& g : h->refGraphs)
            if (memcmp(&g.key, &key, sizeof key) == 0) exec = g.exec;
        if (!exec) {
            hipGraph_t graph = nullptr;
            {
                CaptureStream cs;
                rc = cs.begin() ? ref_enqueue(h, L, inPitch, vocab->v, levelsup, nn_ratio, check_orientation, cs.s) : ORBFE_ERR_HIP;
                graph = cs.end();
            }
            if (rc == ORBFE_OK && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
            if (!exec) {
                graph_capture_failed(h);  // plain launches for this call
                viaGraph = false;
            } else {
                graph_capture_succeeded(h);
                if (h->refGraphs.size() >= 16) ref_drop_graphs(h);  // vocabularies / parameters cycling: bounded cache
                h->refGraphs.push_back({key, exec});
            }
        }
        if (viaGraph) {
            rc = scratch_acquire(h, h->extractUsed, h->extractStream, h->evExtract, s);
            if (rc != ORBFE_OK) return rc;
            HIPCHK(h, hipGraphLaunch(exec, s));
            rc = extract_scratch_release(h, s);
            if (rc != ORBFE_OK) return rc;
            h->lastGray = h->dRefIn;
            h->lastStride = L.inFrame;
            h->lastPitch = inPitch;
            h->lastBatch = 1;
        }
    }
    if (!viaGraph) {
        rc = ref_enqueue(h, L, inPitch, vocab->v, levelsup, nn_ratio, check_orientation, s);
        if (rc != ORBFE_OK) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(s));

    // ---- hand over ----
    const int* head = reinterpret_cast<const int*>(h->hRefOut);
    if (head[1]) {
        char buf[112];
        snprintf(buf, sizeof buf, "device guard flags 0x%x in orbfe_track_reference_keyframe", (unsigned)head[1]);
        h->err = buf;
        return ORBFE_ERR_INTERNAL;
    }
    const int n = head[0];
    *n_out = n;
    *n_matches = n > 0 ? head[2] : 0;
    memcpy(kp_out, h->hRefOut + L.oKp, (size_t)n * sizeof(orbfe_keypoint));
    memcpy(desc_out, h->hRefOut + L.oDesc, (size_t)n * ORBFE_DESC_BYTES);
    memcpy(match_out, h->hRefOut + L.oMatch, (size_t)n * sizeof(int));
    if (per_level) memcpy(per_level, h->hRefOut + L.oPer, (size_t)nL * sizeof(int));
    const int* bow = reinterpret_cast<const int*>(h->hRefOut + L.oBow);
    const int* leaf = reinterpret_cast<const int*>(h->hRefOut + L.oLeaf);
    for (int i = 0; i < n; i++) {
        word_id_out[i] = bow[2 * i];
        node_id_out[i] = bow[2 * i + 1];
        if (weight_out) weight_out[i] = vocab->v->hWeight[(size_t)leaf[i]];  // m_nodes[final_id].weight, TemplatedVocabulary.h:1268
    }
    return ORBFE_OK;
}


// ---------------------------------------------------------------------------------------------
// orbfe_track_initialization: extract -> SearchForInitialization(resident initial frame, frame) as one captured hipGraph
// ---------------------------------------------------------------------------------------------
struct orbfe_init_frame {
    int device;
    unsigned long long serial;  // names the frame in graph keys (an address could be reused after a destroy)
    int n, n0;                  // keypoints, level-0 keypoints (the only ones SearchForInitialization walks, ORBmatcher.cc:346-347)
    void* block;
    const orbfe_keypoint* kp;
    const uint8_t* desc;
    const int* list0;           // level-0 keypoints in index order
};
static std::atomic<unsigned long long> g_initFrameSerial{1};

extern "C" int orbfe_init_frame_create(orbfe_handle* h, int n, const orbfe_keypoint* kp, const uint8_t* desc, orbfe_init_frame** out)
{
    if (!h || !out || n < 0 || (n > 0 && (!kp || !desc))) return ORBFE_ERR_INVALID_ARG;
    *out = nullptr;
    if (n >= (1 << 20)) return ORBFE_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lk(h->mu);
    HIPCHK(h, hipSetDevice(h->device));
    std::vector<int> list0;
    for (int i = 0; i < n; i++)
        if (kp[i].octave <= 0) list0.push_back(i);  // level1 > 0 -> continue (:346-347)
    Carver c;
    const size_t oKp = c.take((size_t)std::max(n, 1) * sizeof(orbfe_keypoint));
    const size_t oDesc = c.take((size_t)std::max(n, 1) * ORBFE_DESC_BYTES);
    const size_t oList = c.take((size_t)std::max(n, 1) * sizeof(int));  // n entries: the sequential matcher kernel rebuilds the list in place
    std::vector<uint8_t> img(c.off, 0);
    if (n) {
        memcpy(&img[oKp], kp, (size_t)n * sizeof(orbfe_keypoint));
        memcpy(&img[oDesc], desc, (size_t)n * ORBFE_DESC_BYTES);
    }
    if (!list0.empty()) memcpy(&img[oList], list0.data(), list0.size() * sizeof(int));
    void* block = nullptr;
    if (hipMalloc(&block, c.off) != hipSuccess) {
        (void)hipGetLastError();
        h->err = "orbfe_init_frame_create: allocation failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    if (copy_sync(block, img.data(), c.off, hipMemcpyHostToDevice, h->stream) != hipSuccess) {
        (void)hipGetLastError();
        (void)hipFree(block);
        h->err = "orbfe_init_frame_create: upload failed";
        return ORBFE_ERR_HIP;
    }
    uint8_t* b = static_cast<uint8_t*>(block);
    *out = new orbfe_init_frame{h->device, g_initFrameSerial.fetch_add(1), n, (int)list0.size(), block,
                                reinterpret_cast<const orbfe_keypoint*>(b + oKp), b + oDesc, reinterpret_cast<const int*>(b + oList)};
    return ORBFE_OK;
}

extern "C" void orbfe_init_frame_destroy(orbfe_init_frame* f)
{
    if (!f) return;
    (void)hipSetDevice(f->device);
    (void)hipDeviceSynchronize();  // (a replay that reads the block may still be running on some handle's stream)
    if (f->block) (void)hipFree(f->block);
    delete f;
}

extern "C" int orbfe_init_frame_size(const orbfe_init_frame* f) { return f ? f->n : 0; }

namespace {

struct IniLayout {
    size_t inFrame, inBytes;
    size_t oPer, oKp, oDesc, oMatch, outBytes /* what is downloaded */, oScratch, devBytes;
};

IniLayout ini_layout(const orbfe_handle* h, int capN1, size_t scratchBytes)
{
    IniLayout L{};
    const size_t cap = (size_t)h->P.kpCapFrame;
    L.inFrame = align_up((size_t)h->dInPitch * h->prm.image_height, 256);
    L.inBytes = L.inFrame;
    size_t off = 256;
    auto take = [&](size_t bytes) { const size_t o = off; off = align_up(off + bytes, 256); return o; };
    L.oPer = take((size_t)h->nLevels * sizeof(int));
    L.oKp = take(cap * sizeof(orbfe_keypoint));
    L.oDesc = take(cap * ORBFE_DESC_BYTES);
    L.oMatch = take((size_t)std::max(capN1, 1) * sizeof(int));
    L.outBytes = off;
    L.oScratch = take(scratchBytes);
    L.devBytes = off;
    return L;
}

void ini_drop_graphs(orbfe_handle* h)
{
    for (auto& g : h->iniGraphs)
        if (g.exec) (void)hipGraphExecDestroy(g.exec);
    h->iniGraphs.clear();
}

int ini_reserve(orbfe_handle* h, int n1, size_t scratchBytes)
{
    if (h->dIniIn && n1 <= h->iniCapN1 && scratchBytes <= h->iniScratchBytes) return ORBFE_OK;
    ini_drop_graphs(h);
    if (h->dIniIn) (void)hipFree(h->dIniIn);
    if (h->dIniOut) (void)hipFree(h->dIniOut);
    if (h->hIniIn) (void)hipHostFree(h->hIniIn);
    if (h->hIniOut) (void)hipHostFree(h->hIniOut);
    h->dIniIn = h->dIniOut = h->hIniIn = h->hIniOut = nullptr;
    h->iniCapN1 = -1;
    h->iniScratchBytes = 0;
    const int capN1 = std::max(n1 + n1 / 2, h->P.kpCapFrame);  // the initial frame is a frame of this extractor: <= kpCapFrame keypoints
    const size_t capScratch = std::max(scratchBytes + scratchBytes / 2, init_track_scratch_bytes(capN1, std::min(capN1, 1024), h->P.kpCapFrame));
    const IniLayout L = ini_layout(h, capN1, capScratch);
    if (hipMalloc(&h->dIniIn, L.inBytes) != hipSuccess || hipMalloc(&h->dIniOut, L.devBytes) != hipSuccess ||
        hipHostMalloc(&h->hIniIn, L.inBytes) != hipSuccess || hipHostMalloc(&h->hIniOut, L.outBytes) != hipSuccess) {
        (void)hipGetLastError();
        h->err = "orbfe_track_initialization: allocation of the staging blocks failed";
        return ORBFE_ERR_OUT_OF_MEMORY;
    }
    h->iniCapN1 = capN1;
    h->iniScratchBytes = capScratch;
    return ORBFE_OK;
}

// the device side of one call, enqueued on s (directly, or under stream capture)
int ini_enqueue(orbfe_handle* h, const IniLayout& L, int inPitch, const orbfe_init_frame* f1, const orbfe_track_params* tp, int window,
                float nnRatio, int checkOri, hipStream_t s)
{
    const int cap = h->P.kpCapFrame;
    int* dHead = reinterpret_cast<int*>(h->dIniOut);  // [n, status, n_matches]
    orbfe_keypoint* dKp = reinterpret_cast<orbfe_keypoint*>(h->dIniOut + L.oKp);
    int rc = extract_chain(h, h->dIniIn, L.inFrame, inPitch, 1, dKp, h->dIniOut + L.oDesc, dHead, reinterpret_cast<int*>(h->dIniOut + L.oPer),
                           dHead + 1, s);
    if (rc != ORBFE_OK) return rc;
    std::string err;
    rc = init_track_launch(s, f1->n, f1->n0, f1->kp, f1->desc, f1->list0, dKp, h->dIniOut + L.oDesc, dHead, cap, tp->grid_cols, tp->grid_rows,
                           tp->min_x, tp->min_y, tp->grid_inv_w, tp->grid_inv_h, window, nnRatio, checkOri,
                           reinterpret_cast<int*>(h->dIniOut + L.oMatch), dHead + 2, h->dIniOut + L.oScratch, err);
    if (rc != ORBFE_OK) {
        h->err = err;
        return rc;
    }
    // [head | per-level | keypoints | descriptors | matches12 of THIS initial frame]: the block is laid out for iniCapN1 entries,
    // the copy stops behind the n1 that exist
    HIPCHK(h, hipMemcpyAsync(h->hIniOut, h->dIniOut, L.oMatch + (size_t)std::max(f1->n, 1) * sizeof(int), hipMemcpyDeviceToHost, s));
    return ORBFE_OK;
}

}  // namespace

extern "C" int orbfe_track_initialization(orbfe_handle* h, const uint8_t* gray, int pitch, const orbfe_init_frame* f1,
                                          const orbfe_track_params* tp, int window_size, float nn_ratio, int check_orientation,
                                          orbfe_keypoint* kp_out, uint8_t* desc_out, int* n_out, int* per_level, int* matches12_out,
                                          int* n_matches)
{
    if (!h || !gray || !f1 || !tp || !kp_out || !desc_out || !n_out || !n_matches || window_size < 0 || (f1->n > 0 && !matches12_out))
        return ORBFE_ERR_INVALID_ARG;
    if (pitch < h->prm.image_width || pitch >= (1 << 24) || f1->device != h->device) return ORBFE_ERR_INVALID_ARG;
    std::lock_guard<std::mutex> lk(h->mu);
    if (tp->struct_size != (int)sizeof(orbfe_track_params)) {
        h->err = "orbfe_track_params.struct_size does not match this library (rebuild the caller against include/orbfe.h)";
        return ORBFE_ERR_INVALID_ARG;
    }
    if (tp->grid_cols < 1 || tp->grid_rows < 1) return ORBFE_ERR_INVALID_ARG;
    if (h->P.kpCapFrame >= (1 << 20) || tp->grid_cols > 65535 || tp->grid_rows > 32767) return ORBFE_ERR_UNSUPPORTED;
    HIPCHK(h, hipSetDevice(h->device));
    hipStream_t s = h->stream;
    const int W = h->prm.image_width, H = h->prm.image_height, nL = h->nLevels;
    int rc = ini_reserve(h, f1->n, init_track_scratch_bytes(f1->n, f1->n0, h->P.kpCapFrame));
    if (rc != ORBFE_OK) return rc;
    const IniLayout L = ini_layout(h, h->iniCapN1, h->iniScratchBytes);

    // ---- upload (as orbfe_track_frame): pinned frames straight from the caller's buffer, pageable ones through the mirror ----
    bool direct = pitch <= h->dInPitch && (pitch & 3) == 0 && (reinterpret_cast<uintptr_t>(gray) & 3u) == 0;
    if (direct) {
        hipPointerAttribute_t attr;
        if (hipPointerGetAttributes(&attr, gray) != hipSuccess || attr.type != hipMemoryTypeHost) {
            (void)hipGetLastError();
            direct = false;
        }
    }
    int inPitch;
    if (direct) {
        inPitch = pitch;
        HIPCHK(h, hipMemcpyAsync(h->dIniIn, gray, (size_t)pitch * (H - 1) + (size_t)W, hipMemcpyHostToDevice, s));
    } else {
        if ((pitch & 3) == 0 && pitch <= h->dInPitch) {
            inPitch = pitch;
            memcpy(h->hIniIn, gray, (size_t)pitch * (H - 1) + (size_t)W);
        } else {
            inPitch = h->dInPitch;
            for (int y = 0; y < H; y++) memcpy(h->hIniIn + (size_t)y * inPitch, gray + (size_t)y * pitch, (size_t)W);
        }
        HIPCHK(h, hipMemcpyAsync(h->dIniIn, h->hIniIn, (size_t)inPitch * (H - 1) + (size_t)W, hipMemcpyHostToDevice, s));
    }

    // ---- kernels + download: replay the graph of this (initial frame, pitch, parameters), capturing it first if needed ----
    bool viaGraph = h->useGraph && !h->timing;
    if (viaGraph) {
        orbfe_handle::IniKey key;
        memset(&key, 0, sizeof key);
        key.inPitch = inPitch; key.gridCols = tp->grid_cols; key.gridRows = tp->grid_rows; key.window = window_size;
        key.checkOri = check_orientation; key.minX = tp->min_x; key.minY = tp->min_y; key.invW = tp->grid_inv_w; key.invH = tp->grid_inv_h;
        key.nnRatio = nn_ratio; key.frameSerial = f1->serial;
        hipGraphExec_t exec = nullptr;
        for (auto& g : h->iniGraphs)
            if (memcmp(&g.key, &key, sizeof key) == 0) exec = g.exec;
        if (!exec) {
            hipGraph_t graph = nullptr;
            {
                CaptureStream cs;
                rc = cs.begin() ? ini_enqueue(h, L, inPitch, f1, tp, window_size, nn_ratio, check_orientation, cs.s) : ORBFE_ERR_HIP;
                graph = cs.end();
            }
            if (rc == ORBFE_OK && graph && hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0) != hipSuccess) exec = nullptr;
            if (graph) (void)hipGraphDestroy(graph);
            if (!exec) {
                graph_capture_failed(h);  // plain launches for this call
                viaGraph = false;
            } else {
                graph_capture_succeeded(h);
                if (h->iniGraphs.size() >= 8) ini_drop_graphs(h);  // initial frames come and go (:569-602): bounded cache
                h->iniGraphs.push_back({key, exec});
            }
        }
        if (viaGraph) {
            rc = scratch_acquire(h, h->extractUsed, h->extractStream, h->evExtract, s);
            if (rc != ORBFE_OK) return rc;
            HIPCHK(h, hipGraphLaunch(exec, s));
            rc = extract_scratch_release(h, s);
            if (rc != ORBFE_OK) return rc;
            h->lastGray = h->dIniIn;
            h->lastStride = L.inFrame;
            h->lastPitch = inPitch;
            h->lastBatch = 1;
        }
    }
    if (!viaGraph) {
        rc = ini_enqueue(h, L, inPitch, f1, tp, window_size, nn_ratio, check_orientation, s);
        if (rc != ORBFE_OK) return rc;
    }
    HIPCHK(h, hipStreamSynchronize(s));

    // ---- hand over ----
    const int* head = reinterpret_cast<const int*>(h->hIniOut);
    if (head[1]) {
        char buf[112];
        snprintf(buf, sizeof buf, "device guard flags 0x%x in orbfe_track_initialization", (unsigned)head[1]);
        h->err = buf;
        return ORBFE_ERR_INTERNAL;
    }
    const int n = head[0];
    *n_out = n;
    memcpy(kp_out, h->hIniOut + L.oKp, (size_t)n * sizeof(orbfe_keypoint));
    memcpy(desc_out, h->hIniOut + L.oDesc, (size_t)n * ORBFE_DESC_BYTES);
    if (per_level) memcpy(per_level, h->hIniOut + L.oPer, (size_t)nL * sizeof(int));
    if (n > 0 && f1->n > 0) {
        *n_matches = head[2];
        memcpy(matches12_out, h->hIniOut + L.oMatch, (size_t)f1->n * sizeof(int));
    } else {  // the reference returns early on either empty frame (vnMatches12 all -1)
        *n_matches = 0;
        for (int i = 0; i < f1->n; i++) matches12_out[i] = -1;
    }
    return ORBFE_OK;
}
