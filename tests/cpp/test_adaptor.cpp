// Drives include/orbfe_adaptor.hpp the way src/Frame.cc:178-189 and src/Tracking.cc:1115 drive the
// reference classes, with light mock Frame / MapPoint types that carry the members those functions read.
//   usage: test_adaptor <W> <H> <gray.raw> <mps.bin> <M> <out.bin> [<voc.txt> <bow_out.txt> [<world.bin> <M2> <track_out.bin>]]
#include <chrono>
#include <cmath>
#include <cstdio>
#include <fstream>
#include <iostream>
#include <set>
#include <tuple>

#include "orbfe_adaptor.hpp"

using namespace ORB_SLAM3;

struct MapPoint {
    bool mbTrackInView = true;
    float mTrackDepth = 1.f, mTrackViewCos = 1.f, mTrackProjX = 0.f, mTrackProjY = 0.f;
    int mnTrackScaleLevel = 0, obs = 1;
    float mTrackProjXR = 0.f, mfMinDistance = 0.5f, mfMaxDistance = 30.f;
    long unsigned int mnLastFrameSeen = 0;
    float wp[3] = {0.f, 0.f, 1.f};
    const float* GetWorldPos() const { return wp; }
    bool bad = false;
    uint8_t desc[32];
    bool isBad() const { return bad; }
    int Observations() const { return obs; }
};

struct KeyFrame;
struct MapPoint3D {  // the MapPoint members ORBmatcher::Fuse touches
    float wp[3] = {0.f, 0.f, 1.f};
    float mfMinDistance = 0.1f, mfMaxDistance = 100.f;
    uint8_t desc[32];
    bool bad = false;
    int obs = 1;
    std::vector<const KeyFrame*> inKF;
    const float* GetWorldPos() const { return wp; }
    bool isBad() const { return bad; }
    int Observations() const { return obs; }
    bool IsInKeyFrame(const std::shared_ptr<KeyFrame>& kf) const
    {
        for (auto k : inKF) if (k == kf.get()) return true;
        return false;
    }
    void AddObservation(const std::shared_ptr<KeyFrame>& kf, size_t) { inKF.push_back(kf.get()); obs++; }
    void Replace(const std::shared_ptr<MapPoint3D>& other) { bad = true; other->obs += obs; }
    std::tuple<int, int> GetIndexInKeyFrame(const std::shared_ptr<KeyFrame>&) const { return {-1, -1}; }
};

struct KeyFrame {
    int N = 0, NLeft = -1, NRight = -1;
    std::shared_ptr<std::vector<KeyPoint>> mvKeysUn;
    std::vector<uint8_t> mDescriptors;
    std::map<unsigned, std::vector<unsigned>> mFeatVec;
    std::vector<std::shared_ptr<MapPoint3D>> mvpMapPoints;
    std::vector<float> mvuRight, mvScaleFactors, mvInvLevelSigma2;
    int mnGridCols = 64, mnGridRows = 48;  // src/KeyFrame.cc:45 copies the grid geometry from the Frame
    float mnMinX = 0, mnMinY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    std::shared_ptr<MapPoint3D> GetMapPoint(size_t i) const { return mvpMapPoints[i]; }
    void AddMapPoint(const std::shared_ptr<MapPoint3D>& mp, size_t i) { mvpMapPoints[i] = mp; }
    std::vector<std::shared_ptr<MapPoint3D>> GetMapPointMatches() const { return mvpMapPoints; }
    std::set<std::shared_ptr<MapPoint3D>> GetMapPoints() const
    {
        std::set<std::shared_ptr<MapPoint3D>> s;
        for (auto& m : mvpMapPoints) if (m && !m->bad) s.insert(m);
        return s;
    }
};

struct RelocFrame {  // the Frame members the relocalisation overload writes
    std::vector<std::shared_ptr<MapPoint3D>> mvpMapPoints;
};

struct Frame {
    int mNumKeypoints = 0, Nleft = -1;
    std::shared_ptr<std::vector<KeyPoint>> mvKeysUn;
    std::vector<uint8_t> mDescriptors;
    std::vector<std::shared_ptr<MapPoint>> mvpMapPoints;
    std::map<unsigned, double> mBowVec;                    // DBoW2::BowVector
    std::map<unsigned, std::vector<unsigned>> mFeatVec;    // DBoW2::FeatureVector
    std::vector<float> mvScaleFactors;
    float mnMinX = 0, mnMinY = 0, mfGridElementWidthInv = 0, mfGridElementHeightInv = 0;
    int cols = 64, rows = 48;
    int getFrameGridCols() const { return cols; }
    int getFrameGridRows() const { return rows; }
};

static std::vector<uint8_t> slurp(const char* p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

int main(int argc, char** argv)
{
    if (argc < 7) { std::printf("%s\n", orbfe_version()); return 0; }  // link smoke test (no GPU needed)
    const int W = atoi(argv[1]), H = atoi(argv[2]), M = atoi(argv[5]);
    const auto img = slurp(argv[3]);
    const auto mpraw = slurp(argv[4]);  // M x (orbfe_map_point + 32 desc bytes)
    ORBextractor ex(1000, 40000, 1.2f, 8, 20, 7, W, H);
    auto r = ex.extractFeatures(GrayImageView{img.data(), W});
    if (!r) { std::puts("nullopt"); return 2; }
    auto [keys, desc] = *r;

    {   // the upstream-style call operator forwards to the same extraction
        std::vector<KeyPoint> k2;
        std::vector<uint8_t> d2;
        const int n2 = ex(GrayImageView{img.data(), W}, k2, d2);
        const bool same = n2 == (int)keys->size() && std::memcmp(k2.data(), keys->data(), (size_t)n2 * sizeof(KeyPoint)) == 0 && d2 == desc;
        std::vector<KeyPoint> k3;
        std::vector<uint8_t> d3;
        ex(GrayImageView{img.data(), W}, k3, d3, true);
        bool scaled = k3.size() == k2.size();
        const auto sf = ex.GetScaleFactors();
        for (size_t i = 0; i < k3.size() && scaled; i++) scaled = k3[i].pt.x == k2[i].pt.x * sf[k2[i].octave] && k3[i].pt.y == k2[i].pt.y * sf[k2[i].octave];
        std::printf("call_operator same=%d scaled=%d\n", (int)same, (int)scaled);
    }
    auto F = std::make_shared<Frame>();  // Frame::Frame, src/Frame.cc:56-135 (the parts the matcher reads)
    F->mNumKeypoints = (int)keys->size();
    F->mvKeysUn = keys;
    F->mDescriptors = desc;
    F->mvpMapPoints.assign(keys->size(), nullptr);
    F->mvScaleFactors = ex.GetScaleFactors();
    F->mfGridElementWidthInv = (float)F->cols / (float)(W - 0.f);
    F->mfGridElementHeightInv = (float)F->rows / (float)(H - 0.f);

    std::vector<std::shared_ptr<MapPoint>> mps;
    const size_t rec = sizeof(orbfe_map_point) + 32;
    for (int i = 0; i < M; i++) {
        orbfe_map_point q;
        std::memcpy(&q, mpraw.data() + i * rec, sizeof q);
        auto p = std::make_shared<MapPoint>();
        p->mTrackProjX = q.proj_x; p->mTrackProjY = q.proj_y; p->mTrackViewCos = q.view_cos; p->mTrackDepth = q.track_depth;
        p->mnTrackScaleLevel = q.level; p->mbTrackInView = q.in_view; p->bad = q.bad; p->obs = q.observations;
        std::memcpy(p->desc, mpraw.data() + i * rec + sizeof q, 32);
        mps.push_back(p);
    }
    const int nm = ORBmatcher::SearchByProjection(
        ex.handle(), F, mps, 20.f, false, 0.f, 0.85f, false, [](const std::shared_ptr<Frame>& f) { return f->mDescriptors.data(); },
        [](const std::shared_ptr<MapPoint>& p) { return (const uint8_t*)p->desc; });

    std::ofstream o(argv[6], std::ios::binary);
    const int n = (int)keys->size();
    o.write((const char*)&n, 4);
    o.write((const char*)&nm, 4);
    o.write((const char*)keys->data(), (std::streamsize)n * sizeof(KeyPoint));
    o.write((const char*)desc.data(), (std::streamsize)n * 32);
    for (int i = 0; i < n; i++) {
        int idx = -1;
        for (int j = 0; j < M && F->mvpMapPoints[i]; j++)
            if (mps[j] == F->mvpMapPoints[i]) { idx = j; break; }
        o.write((const char*)&idx, 4);
    }
    if (argc >= 9) {  // Frame::ComputeBoW, src/Frame.cc:483-495: mpORBvocabulary->transform(vCurrentDesc, mBowVec, mFeatVec, 4)
        ORBVocabulary voc(ex.handle());
        if (!voc.loadFromTextFile(argv[7])) { std::puts("vocabulary load failed"); return 3; }
        std::map<unsigned, double> bowVec;                    // DBoW2::BowVector
        std::map<unsigned, std::vector<unsigned>> featVec;    // DBoW2::FeatureVector
        voc.transform(desc.data(), n, bowVec, featVec, 4);
        FILE* fo = fopen(argv[8], "w");
        std::fprintf(fo, "%zu %zu %u\n", bowVec.size(), featVec.size(), voc.size());
        for (auto& e : bowVec) std::fprintf(fo, "%u %a\n", e.first, e.second);
        for (auto& e : featVec) {
            std::fprintf(fo, "%u %zu", e.first, e.second.size());
            for (unsigned i : e.second) std::fprintf(fo, " %u", i);
            std::fprintf(fo, "\n");
        }
        fclose(fo);
        {   // Tracking::TrackReferenceKeyFrame up to the pose solver as ONE submission: this image's own features as the
            // reference key frame (6 of 7 with a map point); the fused chain must leave the frame with the BoW of transform()
            // and the matches of ORBmatcher::SearchByBoW on the same FeatureVectors
            auto kfr = std::make_shared<KeyFrame>();
            kfr->N = n;
            kfr->mvKeysUn = keys;
            kfr->mDescriptors = desc;
            kfr->mFeatVec = featVec;
            kfr->mvuRight.assign(n, -1.f);
            kfr->mvScaleFactors = ex.GetScaleFactors();
            kfr->mvpMapPoints.resize(n);
            for (int i = 0; i < n; i++)
                if (i % 7) kfr->mvpMapPoints[i] = std::make_shared<MapPoint3D>();
            auto kfDesc = [](const std::shared_ptr<KeyFrame>& k) { return k->mDescriptors.data(); };
            ResidentKeyFrame rk(ex.handle(), kfr, kfDesc);
            auto F3 = std::make_shared<Frame>();
            std::vector<std::shared_ptr<MapPoint3D>> vm, vm2;
            const int nmR = ReferenceKeyFrameTracker::ExtractAndSearchByBoW(
                ex, GrayImageView{img.data(), W}, F3, voc, kfr, rk, vm, 0.75f, true,
                [](const std::shared_ptr<Frame>& f, const uint8_t* rows, int m) { f->mDescriptors.assign(rows, rows + (size_t)m * 32); });
            const int nmS = ORBmatcher::SearchByBoW(ex.handle(), kfr, F3, vm2, kfr->mFeatVec, F3->mFeatVec, 0.75f, true, kfDesc,
                                                    [](const std::shared_ptr<Frame>& f) { return f->mDescriptors.data(); });
            int self = 0;
            for (int i = 0; i < n && i < (int)vm.size(); i++) self += vm[i] && vm[i] == kfr->mvpMapPoints[i];
            std::printf("refkf n=%d seq=%d same=%d self=%d bow_same=%d frame_same=%d\n", nmR, nmS, (int)(vm == vm2), self,
                        (int)(F3->mBowVec == bowVec && F3->mFeatVec == featVec),
                        (int)(F3->mNumKeypoints == n && F3->mDescriptors == desc &&
                              std::memcmp(F3->mvKeysUn->data(), keys->data(), (size_t)n * sizeof(KeyPoint)) == 0));
        }
    }
    if (argc >= 12) {  // the whole per-frame chain through FrameTracker (orbfe_track_frame): frame + local map -> matches
        const auto wraw = slurp(argv[9]);  // M2 x (orbfe_world_point + 32 descriptor bytes + mnLastFrameSeen flag in .skip)
        const int M2 = atoi(argv[10]);
        const size_t wrec = sizeof(orbfe_world_point) + 32;
        std::vector<std::shared_ptr<MapPoint>> local;
        for (int i = 0; i < M2; i++) {
            orbfe_world_point q;
            std::memcpy(&q, wraw.data() + i * wrec, sizeof q);
            auto p = std::make_shared<MapPoint>();
            p->wp[0] = q.x; p->wp[1] = q.y; p->wp[2] = q.z;
            p->mfMinDistance = q.min_distance; p->mfMaxDistance = q.max_distance;
            p->bad = q.bad; p->obs = q.observations; p->mnLastFrameSeen = q.skip ? 7 : 0;
            p->mbTrackInView = false;
            std::memcpy(p->desc, wraw.data() + i * wrec + sizeof q, 32);
            local.push_back(p);
        }
        orbfe_frustum fr{};
        fr.rcw[0] = fr.rcw[4] = fr.rcw[8] = 1.0f;
        fr.min_x = 0.f; fr.max_x = (float)W; fr.min_y = 0.f; fr.max_y = (float)H;
        fr.fx = fr.fy = 400.f; fr.cx = 0.5f * (float)W; fr.cy = 0.5f * (float)H;
        fr.mbf = 40.f; fr.log_scale_factor = 0.18232156f; fr.n_levels = 8; fr.camera_model = ORBFE_CAMERA_PINHOLE;
        auto F2 = std::make_shared<Frame>();
        F2->mvScaleFactors = ex.GetScaleFactors();
        F2->mfGridElementWidthInv = (float)F2->cols / (float)(W - 0.f);
        F2->mfGridElementHeightInv = (float)F2->rows / (float)(H - 0.f);
        int nToMatch = 0;
        const int nmT = FrameTracker::ExtractAndSearchLocalPoints(
            ex, GrayImageView{img.data(), W}, F2, fr, 7, local, 40.f, false, 0.f, 0.75f,
            [](const std::shared_ptr<MapPoint>& p) { return (const uint8_t*)p->desc; },
            [](const std::shared_ptr<Frame>& f, const uint8_t* rows, int n) { f->mDescriptors.assign(rows, rows + (size_t)n * 32); },
            &nToMatch);
        std::ofstream ot(argv[11], std::ios::binary);
        const int nT = F2->mNumKeypoints;
        ot.write((const char*)&nT, 4);
        ot.write((const char*)&nmT, 4);
        ot.write((const char*)&nToMatch, 4);
        ot.write((const char*)F2->mvKeysUn->data(), (std::streamsize)nT * sizeof(KeyPoint));
        ot.write((const char*)F2->mDescriptors.data(), (std::streamsize)nT * 32);
        for (int i = 0; i < nT; i++) {
            int idx = -1;
            for (int j = 0; j < M2 && F2->mvpMapPoints[i]; j++)
                if (local[j] == F2->mvpMapPoints[i]) { idx = j; break; }
            ot.write((const char*)&idx, 4);
        }
        for (auto& p : local) {
            const int v = p->mbTrackInView ? p->mnTrackScaleLevel : -1;
            ot.write((const char*)&v, 4);
        }
    }
    {  // Tracking::SearchLocalPoints, src/Tracking.cc:1059-1077: project a deterministic cloud (identity pose)
        std::vector<std::shared_ptr<MapPoint>> cloud;
        for (int j = 0; j < 3000; j++) {
            auto mp = std::make_shared<MapPoint>();
            mp->wp[0] = (float)(j % 37 - 18) * 0.25f;
            mp->wp[1] = (float)(j % 23 - 11) * 0.2f;
            mp->wp[2] = 2.0f + (float)(j % 11);
            mp->bad = j % 97 == 0;
            mp->mnLastFrameSeen = j % 53 == 0 ? 7 : 0;
            cloud.push_back(mp);
        }
        orbfe_frustum fr{};
        fr.rcw[0] = fr.rcw[4] = fr.rcw[8] = 1.0f;
        fr.min_x = 0.f; fr.max_x = (float)W; fr.min_y = 0.f; fr.max_y = (float)H;
        fr.fx = fr.fy = 400.f; fr.cx = 0.5f * (float)W; fr.cy = 0.5f * (float)H;
        fr.mbf = 40.f; fr.log_scale_factor = 0.18232156f; fr.n_levels = 8; fr.camera_model = ORBFE_CAMERA_PINHOLE;
        const int nToMatch = LocalPointProjector::ProjectLocalMapPoints(ex.handle(), fr, 7, cloud);
        long lvlSum = 0;
        for (auto& mp : cloud) if (mp->mbTrackInView) lvlSum += mp->mnTrackScaleLevel;
        std::printf("frustum nToMatch=%d levelSum=%ld\n", nToMatch, lvlSum);
    }
    {  // LocalMapping::CreateNewMapPoints / SearchInNeighbors: key frame against itself
        auto kf = std::make_shared<KeyFrame>();
        kf->N = n;
        kf->mvKeysUn = keys;
        kf->mDescriptors = desc;
        kf->mvpMapPoints.assign(n, nullptr);
        kf->mvuRight.assign(n, -1.f);
        kf->mvScaleFactors = ex.GetScaleFactors();
        kf->mvInvLevelSigma2 = ex.GetInverseScaleSigmaSquares();
        for (int i = 0; i < n; i++) kf->mFeatVec[(unsigned)(i % 50)].push_back((unsigned)i);
        orbfe_tri_params tp = ORBFE_TRI_PARAMS_INIT;
        tp.ep_x = -1e4f; tp.ep_y = 0.f; tp.coarse = 1; tp.check_orientation = 1;
        std::vector<std::pair<size_t, size_t>> pairs;
        const int nt = KeyFrameMatcher::SearchForTriangulation(ex.handle(), kf, kf, tp, pairs,
                                                               [](const std::shared_ptr<KeyFrame>& k) { return k->mDescriptors.data(); });
        size_t self = 0;
        for (auto& pr : pairs) self += pr.first == pr.second;
        {   // the same pair through resident key frames + the batched entry point (K = 3 copies of the neighbour): every
            // neighbour's Matches() must equal the sequential call as long as pKF1 gains no map point in between; then
            // give a few features map points and check that they drop out
            ResidentKeyFrame rk(ex.handle(), kf, [](const std::shared_ptr<KeyFrame>& k) { return k->mDescriptors.data(); });
            std::vector<std::shared_ptr<KeyFrame>> neigh{kf, kf, kf};
            std::vector<const ResidentKeyFrame*> rn{&rk, &rk, &rk};
            std::vector<orbfe_tri_params> prms{tp, tp, tp};
            TriangulationBatch tb(ex.handle(), kf, rk, neigh, rn, prms);
            std::vector<std::pair<size_t, size_t>> p0, p2;
            const int n0 = tb.Matches(0, kf, p0);
            int dropped = 0;
            for (size_t q = 0; q < p0.size() && q < 40; q += 4) {
                kf->mvpMapPoints[p0[q].first] = std::make_shared<MapPoint3D>();
                dropped++;
            }
            const int n2b = tb.Matches(2, kf, p2);
            std::vector<std::pair<size_t, size_t>> seq;
            const int nseq = KeyFrameMatcher::SearchForTriangulation(ex.handle(), kf, kf, tp, seq,
                                                                     [](const std::shared_ptr<KeyFrame>& k) { return k->mDescriptors.data(); });
            std::printf("tri_batch first=%d same_first=%d dropped=%d after=%d seq_after=%d same_after=%d\n", n0, (int)(p0 == pairs && n0 == nt),
                        dropped, n2b, nseq, (int)(p2 == seq));
            // host cost of one select (n1 features): the replay runs once per neighbour
            std::vector<uint8_t> nowv(n, 0), binv(n, 3);
            std::vector<int> rawv(n), outv(n);
            for (int i = 0; i < n; i++) rawv[i] = i % 3 ? i : -1;
            int nm2 = 0;
            const auto t0 = std::chrono::steady_clock::now();
            for (int r = 0; r < 2000; r++) orbfe_triangulation_select(n, rawv.data(), binv.data(), nowv.data(), 1, outv.data(), &nm2);
            const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count() / 2000.0;
            std::printf("tri_select_us=%.2f n1=%d\n", us, n);
            kf->mvpMapPoints.assign(n, nullptr);
        }
        // Fuse: map points sitting exactly on every 5th keypoint (identity pose, pinhole), descriptor copied
        orbfe_frustum fr{};
        fr.rcw[0] = fr.rcw[4] = fr.rcw[8] = 1.0f;
        fr.min_x = 0.f; fr.max_x = (float)W; fr.min_y = 0.f; fr.max_y = (float)H;
        fr.fx = fr.fy = 400.f; fr.cx = 0.5f * (float)W; fr.cy = 0.5f * (float)H;
        fr.mbf = 40.f; fr.log_scale_factor = 0.18232156f; fr.n_levels = ex.GetLevels(); fr.camera_model = ORBFE_CAMERA_PINHOLE;
        std::vector<std::shared_ptr<MapPoint3D>> mpts;
        for (int i = 0; i < n; i += 5) {
            auto mp = std::make_shared<MapPoint3D>();
            const float z = 4.0f;
            mp->wp[0] = ((*keys)[i].pt.x - fr.cx) / fr.fx * z;
            mp->wp[1] = ((*keys)[i].pt.y - fr.cy) / fr.fy * z;
            mp->wp[2] = z;
            mp->mfMaxDistance = 5.0f * std::pow(1.2f, (float)(*keys)[i].octave);
            mp->mfMinDistance = 0.05f;
            std::memcpy(mp->desc, &desc[(size_t)i * 32], 32);
            mpts.push_back(mp);
        }
        orbfe_frame_view kv{};
        kv.n = n; kv.kp = reinterpret_cast<const orbfe_keypoint*>(keys->data()); kv.desc = desc.data();
        kv.grid_cols = 64; kv.grid_rows = 48; kv.min_x = 0.f; kv.min_y = 0.f;
        kv.grid_inv_w = 64.f / (float)W; kv.grid_inv_h = 48.f / (float)H;
        kv.n_levels = ex.GetLevels(); kv.scale_factors = kf->mvScaleFactors.data();
        const int nf = KeyFrameMatcher::Fuse(ex.handle(), kf, mpts, 3.0f, fr, kv, 0,
                                             [](const std::shared_ptr<MapPoint3D>& m) { return m->desc; });
        std::printf("triangulation n=%d self=%zu fuse=%d of %zu\n", nt, self, nf, mpts.size());
        {   // the same Fuse on RESIDENT data (LocalMapping::SearchInNeighbors' shape): fresh copies of the map points in a
            // ResidentMap, the key frame with its grid tables on the GPU; ids and a frustum go up, nothing else
            auto kfq = std::make_shared<KeyFrame>(*kf);
            kfq->mvpMapPoints.assign(n, nullptr);
            kfq->mfGridElementWidthInv = kv.grid_inv_w;
            kfq->mfGridElementHeightInv = kv.grid_inv_h;
            std::vector<std::shared_ptr<MapPoint3D>> fresh;
            for (auto& m : mpts) {
                auto c = std::make_shared<MapPoint3D>(*m);
                c->inKF.clear(); c->bad = false; c->obs = 1;
                fresh.push_back(c);
            }
            ResidentKeyFrame rq(ex.handle(), kfq, [](const std::shared_ptr<KeyFrame>& k) { return k->mDescriptors.data(); });
            rq.SetGrid(ex.handle(), kfq);
            ResidentMap rmap(ex.handle(), (int)fresh.size() + 8);
            std::vector<int> ids(fresh.size());
            for (size_t i = 0; i < ids.size(); i++) ids[i] = (int)i + 3;
            rmap.Update(fresh, ids, [](const std::shared_ptr<MapPoint3D>& m) { return (const uint8_t*)m->desc; });
            const int nfr2 = ResidentFuse::Fuse(ex.handle(), kfq, rq, rmap, fresh, ids, 3.0f, fr);
            int sameSlots = 0;
            for (int i = 0; i < n; i++) sameSlots += (kfq->mvpMapPoints[i] != nullptr) == (kf->mvpMapPoints[i] != nullptr);
            std::printf("fuse_resident n=%d same_as_host_pointer_fuse=%d slots_equal=%d\n", nfr2, (int)(nfr2 == nf), (int)(sameSlots == n));
        }
        {   // the chain of a frame during monocular initialisation: the frame's own features as mInitialFrame
            InitializationTracker ini(ex);
            auto F1 = std::make_shared<Frame>();
            F1->mvKeysUn = keys;
            F1->mDescriptors = desc;
            F1->mvScaleFactors = ex.GetScaleFactors();
            F1->mfGridElementWidthInv = (float)F1->cols / (float)W;
            F1->mfGridElementHeightInv = (float)F1->rows / (float)H;
            ini.Seed(F1, [](const std::shared_ptr<Frame>& f) { return f->mDescriptors.data(); });
            auto F2 = std::make_shared<Frame>();
            F2->mfGridElementWidthInv = (float)F2->cols / (float)W;
            F2->mfGridElementHeightInv = (float)F2->rows / (float)H;
            auto [nIni, m12] = ini.ExtractAndSearch(GrayImageView{img.data(), W}, F2,
                                                    [](const std::shared_ptr<Frame>& f, const uint8_t* rows, int k) { f->mDescriptors.assign(rows, rows + (size_t)k * 32); });
            const auto two = ORBmatcher::SearchForInitialization(ex.handle(), F1, F, 40, 0.45f, true,
                                                                 [](const std::shared_ptr<Frame>& f) { return f->mDescriptors.data(); });
            int self0 = 0;
            for (size_t i = 0; i < m12.size(); i++) self0 += m12[i] == (int)i;
            std::printf("track_initialization n=%d same_as_two_calls=%d self=%d keypoints=%d\n", nIni, (int)(nIni == two.first && m12 == two.second),
                        self0, F2->mNumKeypoints);
            ini.Reset();
        }
        {   // Fuse(..., bRight = true) on a two-camera key frame whose right features are copies of the left ones seen
            // from the same pose: the same map points fuse, with indices shifted by NLeft (:820)
            auto kfr = std::make_shared<KeyFrame>(*kf);
            kfr->NLeft = n; kfr->NRight = n; kfr->N = 2 * n;
            kfr->mDescriptors.insert(kfr->mDescriptors.end(), desc.begin(), desc.end());
            kfr->mvpMapPoints.assign(2 * (size_t)n, nullptr);
            std::vector<std::shared_ptr<MapPoint3D>> fresh;
            for (auto& m : mpts) { auto c = std::make_shared<MapPoint3D>(*m); c->bad = false; c->inKF.clear(); fresh.push_back(c); }
            orbfe_frame_view kvr = kv;
            kvr.desc = kfr->mDescriptors.data();
            const int nfr = KeyFrameMatcher::Fuse(ex.handle(), kfr, fresh, 3.0f, fr, kvr, 0,
                                                  [](const std::shared_ptr<MapPoint3D>& m) { return m->desc; }, true);
            int shifted = 0, left = 0;
            for (int i = 0; i < 2 * n; i++)
                if (kfr->mvpMapPoints[i]) (i >= n ? shifted : left)++;
            std::printf("fuse_right n=%d shifted=%d left=%d\n", nfr, shifted, left);
        }

        // the remaining ORBmatcher statics on a key frame observed from its own pose (identity Sim3):
        // every 3rd keypoint owns a map point sitting exactly on it
        auto kf2 = std::make_shared<KeyFrame>(*kf);
        kf2->mvpMapPoints.assign(n, nullptr);
        kf->mvpMapPoints.assign(n, nullptr);
        auto on_keypoint = [&](int i) {
            auto mp = std::make_shared<MapPoint3D>();
            const float z = 4.0f;
            mp->wp[0] = ((*keys)[i].pt.x - fr.cx) / fr.fx * z;
            mp->wp[1] = ((*keys)[i].pt.y - fr.cy) / fr.fy * z;
            mp->wp[2] = z;
            mp->mfMaxDistance = 5.0f * std::pow(1.2f, (float)(*keys)[i].octave);
            mp->mfMinDistance = 0.05f;
            std::memcpy(mp->desc, &desc[(size_t)i * 32], 32);
            return mp;
        };
        for (int i = 0; i < n; i += 3) {
            kf->mvpMapPoints[i] = on_keypoint(i);
            kf2->mvpMapPoints[i] = on_keypoint(i);
        }
        auto descOfMP = [](const std::shared_ptr<MapPoint3D>& m) { return m->desc; };
        orbfe_sim3_view dv{};
        dv.rcw[0] = dv.rcw[4] = dv.rcw[8] = 1.0f;
        dv.sr[0] = dv.sr[4] = dv.sr[8] = 1.0f;
        dv.fx = fr.fx; dv.fy = fr.fy; dv.cx = fr.cx; dv.cy = fr.cy;
        dv.min_x = 0.f; dv.max_x = (float)W; dv.min_y = 0.f; dv.max_y = (float)H;
        dv.log_scale_factor = fr.log_scale_factor; dv.n_levels = ex.GetLevels();
        std::vector<std::shared_ptr<MapPoint3D>> vpMatches12(n, nullptr);
        const int nSim3 = KeyFrameMatcher::SearchBySim3(ex.handle(), kf, kf2, vpMatches12, dv, dv, 7.5f, kv, kv, descOfMP);
        int sameSlot = 0;
        for (int i = 0; i < n; i++) sameSlot += vpMatches12[i] && vpMatches12[i] == kf2->mvpMapPoints[i];
        // Sim3 Fuse: candidates = key frame 2's points; key frame 1 holds points on the same keypoints -> replacements
        std::vector<std::shared_ptr<MapPoint3D>> cand, vpReplace;
        for (int i = 0; i < n; i += 3) cand.push_back(kf2->mvpMapPoints[i]);
        for (int i = 1; i < n; i += 6) cand.push_back(on_keypoint(i));  // free keypoints -> new observations
        vpReplace.assign(cand.size(), nullptr);
        const int nFuse3 = KeyFrameMatcher::Fuse(ex.handle(), kf, fr, cand, 4.0f, vpReplace, kv, descOfMP);
        int nRepl = 0, nAdded = 0;
        for (auto& r : vpReplace) nRepl += r != nullptr;
        for (int i = 1; i < n; i += 3) nAdded += kf->mvpMapPoints[i] != nullptr;
        // relocalisation overload: key frame 2's points into an empty frame with the same keypoints
        auto rf = std::make_shared<RelocFrame>();
        rf->mvpMapPoints.assign(n, nullptr);
        std::set<std::shared_ptr<MapPoint3D>> sFound;
        sFound.insert(kf2->mvpMapPoints[0]);
        const int nReloc = KeyFrameMatcher::SearchByProjection(ex.handle(), rf, kf2, sFound, 10.0f, true, fr, kv, descOfMP);
        int relocSame = 0;
        for (int i = 0; i < n; i++) relocSame += rf->mvpMapPoints[i] && rf->mvpMapPoints[i] == kf2->mvpMapPoints[i];
        std::printf("sim3 found=%d same=%d fuse3=%d repl=%d added=%d reloc=%d same=%d slot0=%d\n", nSim3, sameSlot, nFuse3, nRepl,
                    nAdded, nReloc, relocSame, rf->mvpMapPoints[0] ? 1 : 0);
    }
    {  // ImagePreparer: identity maps, dst == src, grey replicated into B, G, R -> the prepared frame is the input frame
        std::vector<float> m1((size_t)W * H), m2((size_t)W * H);
        for (int y = 0; y < H; y++)
            for (int x = 0; x < W; x++) { m1[(size_t)y * W + x] = (float)x; m2[(size_t)y * W + x] = (float)y; }
        std::vector<uint8_t> bgr((size_t)W * H * 3);
        for (size_t i = 0; i < (size_t)W * H; i++) bgr[3 * i] = bgr[3 * i + 1] = bgr[3 * i + 2] = img[i];
        ImagePreparer prep(ex, W, H, m1.data(), m2.data(), W, H);
        const auto grey = prep.ConvertImageToGPU(bgr.data(), W * 3);
        std::vector<uint8_t> grey2;
        auto r2 = prep.extractFeatures(bgr.data(), W * 3, &grey2);
        const bool same = r2 && std::get<0>(*r2)->size() == keys->size() && std::get<1>(*r2) == desc &&
                          std::memcmp(std::get<0>(*r2)->data(), keys->data(), keys->size() * sizeof(KeyPoint)) == 0;
        std::printf("prep grey=%d grey2=%d same=%d\n", (int)(std::memcmp(grey.data(), img.data(), img.size()) == 0),
                    (int)(grey2 == grey), (int)same);
    }
    std::printf("adaptor ok: %d keypoints, %d matches, levels=%d scale=%g\n", n, nm, ex.GetLevels(), ex.GetScaleFactor());
    return 0;
}
