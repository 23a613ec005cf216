// Per-call latency of the tracking thread's entry points through the C ABI itself (no Python in the timed path):
// orbfe_extract, orbfe_track_frame, orbfe_track_frame_map on ONE 752x480 host frame per call (src/Frame.cc:178-189,
// src/Tracking.cc:152-173,1059-1115).  Inputs come from files written by tests/test_track_latency_cpp.py.
//   usage: track_latency <W> <H> <gray.raw> <world.bin> <M> <frustum.bin> [reps]
#include <algorithm>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <fstream>
#include <vector>

#include "orbfe.h"

static std::vector<uint8_t> slurp(const char* p)
{
    std::ifstream f(p, std::ios::binary);
    return std::vector<uint8_t>((std::istreambuf_iterator<char>(f)), std::istreambuf_iterator<char>());
}

template <class Fn>
static double median_us(Fn fn, int reps)
{
    for (int i = 0; i < 10; i++) fn();
    std::vector<double> t((size_t)reps);
    for (int i = 0; i < reps; i++) {
        const auto t0 = std::chrono::steady_clock::now();
        fn();
        t[(size_t)i] = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main(int argc, char** argv)
{
    if (argc < 7) { std::printf("%s\n", orbfe_version()); return 0; }
    const int W = atoi(argv[1]), H = atoi(argv[2]), M = atoi(argv[5]);
    const int reps = argc > 7 ? atoi(argv[7]) : 300;
    const auto img = slurp(argv[3]);
    const auto wraw = slurp(argv[4]);  // M x (orbfe_world_point + 32 descriptor bytes)
    const auto fraw = slurp(argv[6]);
    if ((int)img.size() < W * H || wraw.size() < (size_t)M * 64 || fraw.size() < sizeof(orbfe_frustum)) return 2;
    orbfe_frustum fr;
    std::memcpy(&fr, fraw.data(), sizeof fr);
    std::vector<orbfe_world_point> pts((size_t)M);
    std::vector<uint8_t> mpd((size_t)M * 32);
    std::vector<int> ids((size_t)M);
    for (int i = 0; i < M; i++) {
        std::memcpy(&pts[(size_t)i], wraw.data() + (size_t)i * 64, 32);
        std::memcpy(&mpd[(size_t)i * 32], wraw.data() + (size_t)i * 64 + 32, 32);
        ids[(size_t)i] = pts[(size_t)i].skip ? ~i : i;
    }
    orbfe_params p = {1000, 40000, 1.2f, 8, 20, 7, W, H, 0, 1};
    orbfe_handle* h = nullptr;
    if (orbfe_create(&p, &h) != ORBFE_OK) { std::puts("orbfe_create failed"); return 3; }
    const int cap = orbfe_max_keypoints(h);
    std::vector<orbfe_keypoint> kp((size_t)cap);
    std::vector<uint8_t> desc((size_t)cap * 32);
    std::vector<int> match((size_t)cap), match2((size_t)cap);
    std::vector<orbfe_map_point> rec((size_t)M);
    int n = 0, nm = 0, nm2 = 0;
    orbfe_track_params tp = ORBFE_TRACK_PARAMS_INIT;
    tp.grid_cols = 64; tp.grid_rows = 48; tp.grid_inv_w = 64.f / (float)W; tp.grid_inv_h = 48.f / (float)H;
    tp.th = 20.f; tp.nn_ratio = 0.85f;
    orbfe_map* map = nullptr;
    if (orbfe_map_create(h, M, &map) != ORBFE_OK) return 4;
    {
        std::vector<int> all((size_t)M);
        for (int i = 0; i < M; i++) all[(size_t)i] = i;
        if (orbfe_map_update(h, map, M, all.data(), pts.data(), mpd.data()) != ORBFE_OK) return 5;
    }
    int rc = 0;
    const double e = median_us([&] { rc |= orbfe_extract(h, img.data(), W, kp.data(), desc.data(), &n, nullptr); }, reps);
    const double t = median_us([&] {
        rc |= orbfe_track_frame(h, img.data(), W, &fr, &tp, M, pts.data(), mpd.data(), kp.data(), desc.data(), &n, nullptr, rec.data(), nullptr,
                                match.data(), &nm);
    }, reps);
    const double tm = median_us([&] {
        rc |= orbfe_track_frame_map(h, img.data(), W, &fr, &tp, map, M, ids.data(), kp.data(), desc.data(), &n, nullptr, rec.data(), nullptr,
                                    match2.data(), &nm2);
    }, reps);
    const bool same = nm == nm2 && std::memcmp(match.data(), match2.data(), (size_t)n * sizeof(int)) == 0;
    std::printf("c_abi_latency_us extract=%.1f track_frame=%.1f track_frame_map=%.1f keypoints=%d matches=%d same=%d rc=%d reps=%d\n", e, t, tm, n,
                nm, (int)same, rc, reps);
    if (argc > 8) {  // the chain against the reference key frame: vocabulary from a binary dump, the frame's own features as key frame
        const auto vraw = slurp(argv[8]);  // int32 [nNodes, nEdges, L, levelsup] | childOff | childIdx | wordId | nodeDesc (32 B rows) | weight (f64)
        const int* hd = reinterpret_cast<const int*>(vraw.data());
        const int nNodes = hd[0], nEdges = hd[1], L = hd[2], levelsup = hd[3];
        const int* childOff = hd + 4;
        const int* childIdx = childOff + nNodes + 1;
        const int* wordId = childIdx + nEdges;
        const uint8_t* nodeDesc = reinterpret_cast<const uint8_t*>(wordId + nNodes);
        std::vector<double> weight((size_t)nNodes);
        std::memcpy(weight.data(), nodeDesc + (size_t)nNodes * 32, (size_t)nNodes * sizeof(double));
        orbfe_vocab* voc = nullptr;
        if (orbfe_vocab_create(h, nNodes, childOff, childIdx, nodeDesc, wordId, weight.data(), L, &voc) != ORBFE_OK) return 7;
        rc |= orbfe_extract(h, img.data(), W, kp.data(), desc.data(), &n, nullptr);
        std::vector<int> word((size_t)cap), node((size_t)cap), matchR((size_t)cap);
        std::vector<double> w((size_t)cap);
        rc |= orbfe_bow_transform(h, voc, desc.data(), n, levelsup, word.data(), node.data(), w.data());
        for (int i = 0; i < n; i++)  // the key frame's mFeatVec: a feature on a stopped word (weight 0) is in no node
            if (!(w[(size_t)i] > 0.0)) node[(size_t)i] = -1;
        std::vector<float> sf(8);
        {
            float s = 1.f;
            for (int i = 0; i < 8; i++) { sf[(size_t)i] = s; s *= 1.2f; }
        }
        orbfe_keyframe* kf = nullptr;
        if (orbfe_keyframe_create(h, n, kp.data(), desc.data(), node.data(), nullptr, sf.data(), 8, &kf) != ORBFE_OK) return 8;
        std::vector<uint8_t> has((size_t)n, 1);
        std::vector<orbfe_keypoint> kp2((size_t)cap);
        std::vector<uint8_t> desc2((size_t)cap * 32);
        int n2 = 0, nmR = 0;
        const double tr = median_us([&] {
            rc |= orbfe_track_reference_keyframe(h, img.data(), W, voc, levelsup, kf, has.data(), 0.75f, 1, kp2.data(), desc2.data(), &n2, nullptr,
                                                 word.data(), node.data(), w.data(), matchR.data(), &nmR);
        }, reps);
        int self = 0;
        for (int i = 0; i < n2; i++) self += matchR[(size_t)i] == i;
        std::printf("c_abi_latency_us track_reference_keyframe=%.1f keypoints=%d ref_matches=%d self=%d rc=%d\n", tr, n2, nmR, self, rc);
        orbfe_keyframe_destroy(kf);
        orbfe_vocab_destroy(voc);
    }
    orbfe_map_destroy(map);
    orbfe_destroy(h);
    return rc == 0 && same ? 0 : 6;
}
