"""tests/cpp/track_latency.cpp: the tracking thread's entry points through the C ABI from a plain C++ program (what a
maintainer's binding pays per call, no Python in the timed path); results of orbfe_track_frame and orbfe_track_frame_map
must agree, and match the oracle's chain."""
import ctypes as C
import os
import re
import subprocess

import numpy as np
import pytest

import frustum_scenarios as FS
import oracle_py as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb_slam3_v1.0_amd", "csrc")
BIN = os.path.join(ROOT, "tests", "cpp", "track_latency.bin")


def _build():
    subprocess.check_call(["g++", "-O2", "-std=c++17", "-I", os.path.join(ROOT, "include"), os.path.join(ROOT, "tests", "cpp", "track_latency.cpp"),
                           "-o", BIN, "-L", CSRC, "-lorbfe", "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])


def test_track_latency_program_links(built):
    _build()
    assert "gfx950" in subprocess.check_output([BIN]).decode()


@pytest.mark.gpu
def test_track_latency_program_runs_and_agrees_with_the_oracle(built, tmp_path):
    import orbfe
    from orbfe import synth
    from test_frustum import ON, PN
    _build()
    W, H, M = 752, 480, 2000
    img = synth.frame(W, H, 4)
    eo = O.Extractor(1000, 40000, 1.2, 8, 20, 7, W, H)
    kp, desc, _ = eo.extract(img)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, W=float(W), H=float(H), seed=21)
    FS.fill_frustum(Fp, PN, W=float(W), H=float(H), seed=21)
    pts, mpd = FS.world_points_on_keypoints(kp, desc, v, M, np.random.default_rng(12), 8)
    rec = np.zeros(M, np.dtype([("wp", O.WP_DTYPE), ("d", np.uint8, 32)]))
    rec["wp"], rec["d"] = pts, mpd
    (tmp_path / "g.raw").write_bytes(img.tobytes())
    (tmp_path / "w.bin").write_bytes(rec.tobytes())
    (tmp_path / "f.bin").write_bytes(bytes(Fp))
    # a vocabulary (k = 10, L = 4, levelsup 2: ~100 nodes at the FeatureVector level) as a binary dump for the fourth timing
    import vocab_synth as vs
    t = vs.spread_first_level(vs.make_tree(10, 4, seed=3, early_leaf_p=0.03), 4)
    levelsup = 2
    with open(tmp_path / "voc.bin", "wb") as fvoc:
        fvoc.write(np.array([len(t["wordId"]), len(t["childIdx"]), 4, levelsup], np.int32).tobytes())
        for a, dt in ((t["childOff"], np.int32), (t["childIdx"], np.int32), (t["wordId"], np.int32), (t["nodeDesc"], np.uint8), (t["weight"], np.float64)):
            fvoc.write(np.ascontiguousarray(a, dt).tobytes())
    out = subprocess.check_output([BIN, str(W), str(H), str(tmp_path / "g.raw"), str(tmp_path / "w.bin"), str(M), str(tmp_path / "f.bin"), "100",
                                   str(tmp_path / "voc.bin")]).decode()
    m = re.search(r"c_abi_latency_us extract=([0-9.]+) track_frame=([0-9.]+) track_frame_map=([0-9.]+) keypoints=(\d+) matches=(\d+) same=1 rc=0", out)
    assert m, out
    mps, _ = O.is_in_frustum(Fo, pts)
    fv = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    n_ref, _ = O.search_by_projection(fv, mps, mpd, None, 20.0, 0.85)
    assert int(m.group(4)) == len(kp) and int(m.group(5)) == n_ref and n_ref > 500
    assert float(m.group(1)) < float(m.group(2)) < 5000.0
    print(m.group(0))
    # the chain against the reference key frame from C++: the frame's own features as the key frame, checked against the oracle
    mr = re.search(r"c_abi_latency_us track_reference_keyframe=([0-9.]+) keypoints=(\d+) ref_matches=(\d+) self=(\d+) rc=0", out)
    assert mr, out
    _, node, wt = O.vocab_transform(t["childOff"], t["childIdx"], t["nodeDesc"], t["wordId"], t["weight"], 4, desc, levelsup)
    node = np.where(wt > 0, node, -1)  # mFeatVec holds features whose word has weight > 0 only (TemplatedVocabulary.h:1168-1172)
    kfOff, kfIdx = [0], []
    for g in sorted(set(node.tolist()) - {-1}):
        kfIdx += list(np.flatnonzero(node == g))
        kfOff.append(len(kfIdx))
    n_b, m_b = O.search_by_bow(kfOff, kfIdx, kfOff, kfIdx, desc, kp["angle"], np.ones(len(kp), np.uint8), desc, kp["angle"], 0.75, True)
    assert int(mr.group(2)) == len(kp) and int(mr.group(3)) == n_b and int(mr.group(4)) == int((m_b == np.arange(len(kp))).sum()) and n_b > 500
    assert float(mr.group(1)) < float(m.group(2))  # cheaper than the projection chain at 2000 map points
    print(mr.group(0))
