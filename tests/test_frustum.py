"""Frame::isInFrustum for a batch of map points (src/Frame.cc:272-331) with
Pinhole::project and MapPoint::PredictScale, SPEC DECISION S8."""
import math

import numpy as np
import pytest

import frustum_scenarios as FS
import oracle_py as O

ON = dict(rcw="rcw", tcw="tcw", twc="twc", min_x="minX", max_x="maxX", min_y="minY", max_y="maxY", fx="fx", fy="fy", cx="cx",
          cy="cy", k1="k1", k2="k2", k3="k3", k4="k4", mbf="mbf", log_scale_factor="logScaleFactor", n_levels="nLevels",
          camera_model="cameraModel")
OP = dict(x="x", y="y", z="z", min_distance="minDistance", max_distance="maxDistance", bad="bad", observations="observations",
          skip="skip")
PN = {k: k for k in ON}
PP = {k: k for k in OP}
f32 = np.float32


def py_spec_logf(x):
    """numpy binary32 restatement of orc_spec_logf (pins the C code)."""
    x = f32(x)
    if not x > 0:
        return -math.inf
    m, e = np.frexp(x)
    m, e = f32(m), int(e)
    if m < f32(float.fromhex("0x1.6a09e6p-1")):
        m = f32(m * f32(2))
        e -= 1
    s = f32(f32(m - f32(1)) / f32(m + f32(1)))
    z = f32(s * s)
    p = f32(float.fromhex("0x1.c71c72p-4"))
    for c in ("0x1.24924ap-3", "0x1.99999ap-3", "0x1.555556p-2"):
        p = f32(f32(p * z) + f32(float.fromhex(c)))
    p = f32(p * z)
    t = f32(s + s)
    r = f32(t + f32(t * p))
    ef = f32(e)
    return float(f32(f32(ef * f32(float.fromhex("0x1.62ep-1"))) + f32(r + f32(ef * f32(float.fromhex("0x1.0bfbe8p-15"))))))


def py_in_frustum(v, p):
    """Straight-line binary32 restatement of src/Frame.cc:272-331 for one point (S8 evaluation order)."""
    R, t, c = [f32(x) for x in v["rcw"]], [f32(x) for x in v["tcw"]], [f32(x) for x in v["twc"]]
    o = dict(projX=f32(-1), projY=f32(-1), viewCos=f32(0), trackDepth=f32(0), level=0, inView=0, bad=int(p["bad"]),
             observations=int(p["observations"]))
    xr = f32(0)
    if p["skip"] or p["bad"]:
        return o, xr
    X, Y, Z = f32(p["x"]), f32(p["y"]), f32(p["z"])
    pc = [f32(f32(f32(f32(R[3 * i] * X) + f32(R[3 * i + 1] * Y)) + f32(R[3 * i + 2] * Z)) + t[i]) for i in range(3)]
    dist_c = f32(np.sqrt(f32(f32(f32(pc[0] * pc[0]) + f32(pc[1] * pc[1])) + f32(pc[2] * pc[2]))))
    if pc[2] < 0:
        return o, xr
    with np.errstate(divide="ignore", invalid="ignore"):
        invz = f32(f32(1) / pc[2])
        if v["camera_model"] == 0:
            u = f32(f32(f32(f32(v["fx"]) * pc[0]) / pc[2]) + f32(v["cx"]))
            w = f32(f32(f32(f32(v["fy"]) * pc[1]) / pc[2]) + f32(v["cy"]))
        else:  # KannalaBrandt8.cpp:66-83 on the S5 primitives (pinned by tests/test_oracle_kat.py)
            th = f32(O.spec_atan2f(np.sqrt(f32(f32(pc[0] * pc[0]) + f32(pc[1] * pc[1]))), pc[2]))
            psi = f32(O.spec_atan2f(pc[1], pc[0]))
            t2 = f32(th * th)
            t3 = f32(th * t2)
            t5 = f32(t3 * t2)
            t7 = f32(t5 * t2)
            t9 = f32(t7 * t2)
            r = f32(f32(f32(f32(th + f32(f32(v["k1"]) * t3)) + f32(f32(v["k2"]) * t5)) + f32(f32(v["k3"]) * t7)) + f32(f32(v["k4"]) * t9))
            deg = f32(psi * f32(float.fromhex("0x1.ca5dc2p+5")))
            if deg < 0:
                deg = f32(deg + f32(360))
            cs_, s_ = O.cos_sin_deg(deg)
            u = f32(f32(f32(f32(v["fx"]) * r) * f32(cs_)) + f32(v["cx"]))
            w = f32(f32(f32(f32(v["fy"]) * r) * f32(s_)) + f32(v["cy"]))
    if u < f32(v["min_x"]) or u > f32(v["max_x"]) or w < f32(v["min_y"]) or w > f32(v["max_y"]):
        return o, xr
    o["projX"], o["projY"] = u, w
    maxD, minD = f32(f32(1.1) * f32(p["maxDistance"])), f32(f32(0.9) * f32(p["minDistance"]))
    d = [f32(a - b) for a, b in zip((X, Y, Z), c)]
    dist = f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))
    if dist < minD or dist > maxD:
        return o, xr
    q = f32(f32(py_spec_logf(f32(f32(p["maxDistance"]) / dist))) / f32(v["log_scale_factor"]))
    nL = v["n_levels"]
    if not q > 0:
        lvl = 0
    elif q >= nL:
        lvl = nL - 1
    else:
        lvl = min(int(math.ceil(float(q))), nL - 1)
    o.update(inView=1, level=lvl, viewCos=f32(1), trackDepth=dist_c)
    return o, f32(u - f32(f32(v["mbf"]) * invz))


def test_spec_logf_pin_and_accuracy():
    rng = np.random.default_rng(0)
    xs = np.concatenate([np.exp(rng.uniform(-20, 20, 4000)), [1.0, 0.5, 2.0, 1.2, 1.44, 1e-38, 3e38]]).astype(np.float32)
    worst = 0.0
    for x in xs:
        got = O.spec_logf(x)
        assert got == py_spec_logf(x)
        ref = math.log(float(x))
        worst = max(worst, abs(got - ref) / max(float(np.spacing(f32(abs(ref)))), 1e-45))
    assert worst < 2.5  # ulps; the tolerance of SPEC DECISION S8 against an exact logarithm
    assert O.spec_logf(0.0) == -math.inf and O.spec_logf(-1.0) == -math.inf


@pytest.mark.parametrize("kb8", [False, True])
def test_oracle_frustum_matches_restatement(kb8):
    F = O.Frustum()
    v = FS.fill_frustum(F, ON, seed=3, kb8=kb8)
    pts = FS.world_points(1500, O.WP_DTYPE, OP, seed=4)
    out, xr = O.is_in_frustum(F, pts)
    assert 0.15 < out["inView"].mean() < 0.9
    assert len(set(out["level"][out["inView"] == 1])) == v["n_levels"]
    for i in range(len(pts)):
        o, x = py_in_frustum(v, pts[i])
        for k, val in o.items():
            assert np.array_equal(np.asarray(out[k][i]), np.asarray(val, out[k].dtype), equal_nan=True), (i, k)
        assert np.array_equal(xr[i], x, equal_nan=True)


@pytest.mark.gpu
@pytest.mark.parametrize("n,seed,kb8", [(1, 0, False), (257, 1, False), (5000, 2, False), (100000, 3, False), (20000, 4, True)])
def test_gpu_frustum_matches_oracle(built, n, seed, kb8):
    import orbfe
    e = orbfe.ORBextractor(500, 2000, 1.2, 8, 20, 7, 320, 240)
    m = orbfe.ORBmatcher(e)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    FS.fill_frustum(Fo, ON, seed=seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=seed, kb8=kb8)
    pts = FS.world_points(n, O.WP_DTYPE, OP, seed=seed + 10)
    ref, ref_xr = O.is_in_frustum(Fo, pts)
    out, xr = m.isInFrustum_batch(Fp, pts.view(orbfe.WP_DTYPE))
    assert out.tobytes() == ref.tobytes()
    assert xr.tobytes() == ref_xr.tobytes()
    o0, _ = m.isInFrustum_batch(Fp, pts[:0].view(orbfe.WP_DTYPE))
    assert len(o0) == 0
    Fp.camera_model = 7
    with pytest.raises(orbfe.OrbfeError):
        m.isInFrustum_batch(Fp, pts.view(orbfe.WP_DTYPE))


@pytest.mark.gpu
def test_gpu_project_then_match_chain(built):
    """SearchLocalPoints end to end: isInFrustum for every local map point, then SearchByProjection on the records it
    wrote (src/Tracking.cc:1059-1115) == the oracle's chain."""
    import match_scenarios as S
    import orbfe
    from orbfe import synth
    W, H = 752, 480
    args = (1000, 40000, 1.2, 8, 20, 7, W, H)
    eo = O.Extractor(*args)
    kp, desc, _ = eo.extract(synth.frame(W, H, 5))
    ex = orbfe.ORBextractor(*args)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=7)
    FS.fill_frustum(Fp, PN, seed=7)
    # back-project keypoints (level-0 coordinates) to random depths so that they re-project near themselves
    rng = np.random.default_rng(11)
    M = 1500
    src = rng.integers(0, len(kp), M)
    sf = eo.scaleFactors
    u = kp["x"][src] * sf[kp["octave"][src]] + rng.uniform(-2, 2, M)
    w = kp["y"][src] * sf[kp["octave"][src]] + rng.uniform(-2, 2, M)
    z = rng.uniform(1.5, 9.0, M)
    pc = np.stack([(u - v["cx"]) / v["fx"] * z, (w - v["cy"]) / v["fy"] * z, z], 1)
    R = np.asarray(v["rcw"], np.float64).reshape(3, 3)
    pw = (R.T @ (pc - np.asarray(v["tcw"], np.float64)).T).T
    pts = np.zeros(M, O.WP_DTYPE)
    pts["x"], pts["y"], pts["z"] = pw[:, 0], pw[:, 1], pw[:, 2]
    d = np.linalg.norm(pw - np.asarray(v["twc"], np.float64), axis=1)
    pts["maxDistance"] = d * np.float32(1.2) ** kp["octave"][src].astype(np.float32) * rng.uniform(0.95, 1.05, M)
    pts["minDistance"] = pts["maxDistance"] / np.float32(1.2) ** 7
    pts["observations"] = rng.integers(0, 4, M)
    pts["bad"] = rng.random(M) < 0.02
    mpd = np.stack([S.flip_bits(desc[s], int(rng.integers(0, 20)), rng) for s in src])
    # NOTE: the matcher works in LEVEL coordinates in this fork (S6); the chain test only needs both sides to agree
    mps_o, _ = O.is_in_frustum(Fo, pts)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    n_ref, out_ref = O.search_by_projection(fvo, mps_o, mpd, None, 3.0, 0.8)
    mps_g, _ = m.isInFrustum_batch(Fp, pts.view(orbfe.WP_DTYPE))
    assert mps_g.tobytes() == mps_o.tobytes()
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    n, out = m.SearchByProjection(fv, mps_g, mpd, 3.0, False, 0.0, 0.8, None)
    assert n == n_ref and np.array_equal(out, out_ref)
    assert n_ref > 50
