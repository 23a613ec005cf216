"""The reference-signature C++ adaptor (include/orbfe_adaptor.hpp): compiles + links without a GPU,
and on the GPU box reproduces the oracle through the ORBextractor / ORBmatcher classes."""
import os
import re
import subprocess

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "orb_slam3_v1.0_amd", "csrc")
BIN = os.path.join(ROOT, "tests", "cpp", "test_adaptor.bin")


def _build():
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           os.path.join(ROOT, "tests", "cpp", "test_adaptor.cpp"), "-o", BIN, "-L", CSRC, "-lorbfe",
                           "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])


BIN_CV = os.path.join(ROOT, "tests", "cpp", "test_adaptor_opencv.bin")


def _build_cv():
    """the ORBFE_WITH_OPENCV branch (exact reference signatures) against tests/cpp/mock_opencv: OpenCV is not in the image"""
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"),
                           "-I", os.path.join(ROOT, "tests", "cpp", "mock_opencv"),
                           os.path.join(ROOT, "tests", "cpp", "test_adaptor_opencv.cpp"), "-o", BIN_CV, "-L", CSRC, "-lorbfe",
                           "-Wl,-rpath," + CSRC, "-Wl,-rpath,/opt/rocm/lib"])


def test_reference_signature_branch_type_checks(built):
    _build_cv()  # the static_asserts in the source pin the return types of include/ORBextractor.h:62
    assert "gfx950" in subprocess.check_output([BIN_CV]).decode()


@pytest.mark.gpu
def test_hostmem_overload_equals_view_overload(built, tmp_path):
    from orbfe import synth
    _build_cv()
    W, H = 752, 480
    (tmp_path / "g.raw").write_bytes(synth.frame(W, H, 11).tobytes())
    out = subprocess.check_output([BIN_CV, str(W), str(H), str(tmp_path / "g.raw")]).decode()
    assert "hostmem overload ok" in out


def test_adaptor_compiles_and_links(built):
    _build()
    out = subprocess.check_output([BIN]).decode()
    assert "gfx950" in out


@pytest.mark.gpu
def test_adaptor_matches_oracle(built, tmp_path):
    import match_scenarios as S
    import oracle_py as O
    import orbfe
    from orbfe import synth
    _build()
    W, H, M = 752, 480, 1500
    img = synth.frame(W, H, 4)
    e = O.Extractor(1000, 40000, 1.2, 8, 20, 7, W, H)
    kp_r, desc_r, _ = e.extract(img)
    names = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")
    mps, mpd, _ = S.projection_scenario(kp_r, desc_r, M, 3, O.MP_DTYPE, names, e.nLevels)
    fv = O.make_frame_view(kp_r, desc_r, 64, 48, 0.0, 0.0, float(W), float(H), e.scaleFactors)
    n_ref, out_ref = O.search_by_projection(fv, mps, mpd, None, 20.0, 0.85)
    (tmp_path / "g.raw").write_bytes(img.tobytes())
    rec = np.zeros(M, np.dtype([("mp", O.MP_DTYPE), ("d", np.uint8, 32)]))
    rec["mp"], rec["d"] = mps, mpd
    (tmp_path / "m.bin").write_bytes(rec.tobytes())
    import vocab_synth as vs
    from test_vocab import ref_bow
    tree = vs.make_tree(10, 5, seed=9, early_leaf_p=0.03)
    vs.write_text(tree, str(tmp_path / "voc.txt"))
    # FrameTracker::ExtractAndSearchLocalPoints (orbfe_track_frame): a local map that re-projects onto the frame's keypoints
    import frustum_scenarios as FS
    M2 = 1700
    Ft = O.Frustum()
    Ft.rcw[0] = Ft.rcw[4] = Ft.rcw[8] = 1.0
    Ft.minX, Ft.maxX, Ft.minY, Ft.maxY = 0.0, float(W), 0.0, float(H)
    Ft.fx = Ft.fy = 400.0
    Ft.cx, Ft.cy, Ft.mbf, Ft.logScaleFactor, Ft.nLevels = 0.5 * W, 0.5 * H, 40.0, 0.18232156, 8
    vt = dict(rcw=np.eye(3).reshape(-1), tcw=np.zeros(3), twc=np.zeros(3), fx=400.0, fy=400.0, cx=0.5 * W, cy=0.5 * H)
    wpts, wdesc = FS.world_points_on_keypoints(kp_r, desc_r, vt, M2, np.random.default_rng(8), 8)
    wrec = np.zeros(M2, np.dtype([("wp", O.WP_DTYPE), ("d", np.uint8, 32)]))
    wrec["wp"], wrec["d"] = wpts, wdesc
    (tmp_path / "w.bin").write_bytes(wrec.tobytes())
    stdout = subprocess.check_output([BIN, str(W), str(H), str(tmp_path / "g.raw"), str(tmp_path / "m.bin"), str(M),
                                      str(tmp_path / "o.bin"), str(tmp_path / "voc.txt"), str(tmp_path / "bow.txt"),
                                      str(tmp_path / "w.bin"), str(M2), str(tmp_path / "t.bin")]).decode()
    mps_t, _ = O.is_in_frustum(Ft, wpts)
    n_t, match_t = O.search_by_projection(fv, mps_t, wdesc, None, 40.0, 0.75)
    rawt = (tmp_path / "t.bin").read_bytes()
    nT, nmT, nToMatch = np.frombuffer(rawt[:12], np.int32)
    assert nT == len(kp_r) and rawt[12:12 + 24 * nT] == kp_r.tobytes() and rawt[12 + 24 * nT:12 + 56 * nT] == desc_r.tobytes()
    assert nmT == n_t and n_t > 500 and nToMatch == int(mps_t["inView"].sum())
    assert np.array_equal(np.frombuffer(rawt[12 + 56 * nT:12 + 60 * nT], np.int32), match_t)
    lvl = np.frombuffer(rawt[12 + 60 * nT:], np.int32)
    untouched = (wpts["skip"] != 0) | (wpts["bad"] != 0)  # the reference does not visit them (src/Tracking.cc:1066-1069)
    assert np.array_equal(lvl[~untouched], np.where(mps_t["inView"] == 1, mps_t["level"], -1)[~untouched]) and (lvl[untouched] == -1).all()
    # LocalPointProjector::ProjectLocalMapPoints on the driver's deterministic cloud == the oracle
    Fo = O.Frustum()
    Fo.rcw[0] = Fo.rcw[4] = Fo.rcw[8] = 1.0
    Fo.minX, Fo.maxX, Fo.minY, Fo.maxY = 0.0, float(W), 0.0, float(H)
    Fo.fx = Fo.fy = 400.0
    Fo.cx, Fo.cy, Fo.mbf, Fo.logScaleFactor, Fo.nLevels = 0.5 * W, 0.5 * H, 40.0, 0.18232156, 8
    j = np.arange(3000)
    cloud = np.zeros(3000, O.WP_DTYPE)
    cloud["x"], cloud["y"], cloud["z"] = (j % 37 - 18) * np.float32(0.25), (j % 23 - 11) * np.float32(0.2), 2.0 + j % 11
    cloud["minDistance"], cloud["maxDistance"] = 0.5, 30.0
    cloud["bad"], cloud["skip"], cloud["observations"] = j % 97 == 0, j % 53 == 0, 1
    fo, _ = O.is_in_frustum(Fo, cloud)
    assert "frustum nToMatch=%d levelSum=%d" % (fo["inView"].sum(), fo["level"][fo["inView"] == 1].sum()) in stdout
    assert fo["inView"].sum() > 500
    raw = (tmp_path / "o.bin").read_bytes()
    n, nm = np.frombuffer(raw[:8], np.int32)
    kp = np.frombuffer(raw[8:8 + 24 * n], orbfe.KP_DTYPE)
    desc = np.frombuffer(raw[8 + 24 * n:8 + 56 * n], np.uint8).reshape(n, 32)
    match = np.frombuffer(raw[8 + 56 * n:], np.int32)
    assert n == len(kp_r) and kp.tobytes() == kp_r.tobytes() and np.array_equal(desc, desc_r)
    assert nm == n_ref and np.array_equal(match, out_ref)
    # Frame::ComputeBoW through ORBVocabulary::transform (levelsup 4, TF_IDF + L1 as ORBvoc.txt declares)
    ow, on, owt = O.vocab_transform(tree["childOff"], tree["childIdx"], tree["nodeDesc"], tree["wordId"], tree["weight"],
                                    tree["L"], desc_r, 4)
    rb, rf = ref_bow(ow, on, owt, 0, 0)
    lines = (tmp_path / "bow.txt").read_text().split("\n")
    nb, nf, nw = (int(t) for t in lines[0].split())
    assert nw == tree["nWords"] and nb == len(rb) and nf == len(rf)
    got_b = [(int(l.split()[0]), float.fromhex(l.split()[1])) for l in lines[1:1 + nb]]
    assert got_b == list(rb.items())
    got_f = {int(l.split()[0]): [int(t) for t in l.split()[2:]] for l in lines[1 + nb:1 + nb + nf]}
    assert got_f == rf
    # Tracking::TrackReferenceKeyFrame through ReferenceKeyFrameTracker (orbfe_track_reference_keyframe): == the adaptor's
    # own sequential SearchByBoW, == the frame transform() leaves, and nearly every flagged feature finds itself
    mk = re.search(r"refkf n=(\d+) seq=(\d+) same=(\d+) self=(\d+) bow_same=(\d+) frame_same=(\d+)", stdout)
    assert mk and mk.group(1) == mk.group(2) and mk.group(3) == "1" and mk.group(5) == "1" and mk.group(6) == "1"
    assert int(mk.group(4)) > 0.7 * len(kp_r) * 6 // 7 and int(mk.group(4)) <= int(mk.group(1))
    # KeyFrameMatcher::SearchForTriangulation (key frame against itself, bCoarse) == the oracle on the same groups
    n_kp = len(kp_r)
    groups = [np.arange(g, n_kp, 50) for g in range(min(50, n_kp))]
    off = np.concatenate([[0], np.cumsum([len(g) for g in groups])])
    idx = np.concatenate(groups)
    z8 = np.zeros(n_kp, np.uint8)
    nt, m12 = O.search_for_triangulation(off, idx, off, idx, kp_r, desc_r, z8, None, kp_r, desc_r, z8, None, e.scaleFactors,
                                         np.zeros(9, np.float32), (-1e4, 0.0), False, True, True)
    mt = re.search(r"triangulation n=(\d+) self=(\d+) fuse=(\d+) of (\d+)", stdout)
    assert mt and int(mt.group(1)) == nt and int(mt.group(2)) == int((m12 == np.arange(n_kp)).sum())
    assert int(mt.group(3)) > int(mt.group(4)) // 2  # most on-keypoint map points fuse
    mq = re.search(r"fuse_resident n=(\d+) same_as_host_pointer_fuse=1 slots_equal=1", stdout)  # resident key frame + resident map
    assert mq and int(mq.group(1)) == int(mt.group(3)), stdout
    mi = re.search(r"track_initialization n=(\d+) same_as_two_calls=1 self=(\d+) keypoints=(\d+)", stdout)  # one submission == extract + match
    assert mi and int(mi.group(1)) > 100 and int(mi.group(2)) >= 0.9 * int(mi.group(1)) and int(mi.group(3)) == n_kp, stdout
    mr = re.search(r"fuse_right n=(\d+) shifted=(\d+) left=(\d+)", stdout)   # bRight: same matches, indices + NLeft
    assert mr and int(mr.group(1)) == int(mt.group(3)) and int(mr.group(2)) > 0 and int(mr.group(3)) == 0
    mb = re.search(r"tri_batch first=(\d+) same_first=(\d+) dropped=(\d+) after=(\d+) seq_after=(\d+) same_after=(\d+)", stdout)
    assert mb and int(mb.group(1)) == nt and mb.group(2) == "1" and mb.group(6) == "1" and int(mb.group(4)) == int(mb.group(5))
    assert int(mb.group(3)) > 0 and int(mb.group(4)) <= nt - int(mb.group(3))  # features that got a map point drop out (:506-509)
    print(re.search(r"tri_select_us=[0-9.]+ n1=\d+", stdout).group(0))
    assert "call_operator same=1 scaled=1" in stdout  # the upstream-style operator() overload
    assert "prep grey=1 grey2=1 same=1" in stdout  # ImagePreparer with identity maps reproduces the plain extraction
    # SearchBySim3 / Sim3 Fuse / relocalisation SearchByProjection on a key frame seen from its own pose
    ms = re.search(r"sim3 found=(\d+) same=(\d+) fuse3=(\d+) repl=(\d+) added=(\d+) reloc=(\d+) same=(\d+) slot0=(\d+)", stdout)
    assert ms
    found, same, fuse3, repl, added, reloc, rsame, slot0 = (int(g) for g in ms.groups())
    n_mp = (n_kp + 2) // 3
    assert found == same and found > n_mp * 0.8       # mutual best matches land on the construction's pairs
    assert fuse3 == repl + added and repl > n_mp * 0.8 and added > 0
    assert reloc == rsame and reloc > n_mp * 0.6 and slot0 == 0  # the point in sAlreadyFound is not searched
