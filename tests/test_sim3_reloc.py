"""The three remaining ORBmatcher statics (include/ORBmatcher.h:48,63,69): the Sim3 overload of Fuse
(src/ORBmatcher.cc:864-975), SearchBySim3 (:977-1200) and the relocalisation overload of SearchByProjection
(:1202-1326).  CPU: the C oracle against plain binary32 restatements; GPU: the HIP path against the oracle."""
import math

import numpy as np
import pytest

import frustum_scenarios as FS
import match_scenarios as S
import oracle_py as O
from test_frustum import ON, PN, py_spec_logf
from test_fuse import scenario

W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)
GRID = (64, 48)
f32 = np.float32
K = dict(fx=458.654, fy=457.296, cx=367.215, cy=248.375)


# ---------------------------------------------------------------------------------------------
# plain restatements (numpy binary32 scalars, one rounding per operation == SPEC DECISION S8)
# ---------------------------------------------------------------------------------------------
def mat_apply(R, t, p):
    R, t = [f32(x) for x in R], [f32(x) for x in t]
    return [f32(f32(f32(f32(R[3 * i] * p[0]) + f32(R[3 * i + 1] * p[1])) + f32(R[3 * i + 2] * p[2])) + t[i]) for i in range(3)]


def norm3(d):
    return f32(np.sqrt(f32(f32(f32(d[0] * d[0]) + f32(d[1] * d[1])) + f32(d[2] * d[2]))))


def py_predict_scale(max_distance, dist, log_sf, n_levels):
    with np.errstate(divide="ignore", invalid="ignore"):
        q = f32(f32(py_spec_logf(f32(f32(max_distance) / dist))) / f32(log_sf))
    if not q > 0:
        return 0
    if q >= n_levels:
        return n_levels - 1
    return min(int(math.ceil(float(q))), n_levels - 1)


def cells_of(kp, bounds, grid):
    """Frame::PosInGrid (src/Frame.cc:470-480) per keypoint: (cell x, cell y) or None."""
    cols, rows = grid
    min_x, max_x, min_y, max_y = [f32(b) for b in bounds]
    invw, invh = f32(cols) / f32(max_x - min_x), f32(rows) / f32(max_y - min_y)
    out = []
    for k in kp:
        px = int(np.round(f32(f32(k["x"] - min_x) * invw)))
        py = int(np.round(f32(f32(k["y"] - min_y) * invh)))
        lin = py * cols + px
        out.append((lin % cols, lin // cols) if 0 <= lin < cols * rows else None)
    return out, invw, invh


def area_candidates(kp, cell, invw, invh, bounds, grid, u, w, r):
    """GetFeaturesInArea visit order (cell x, cell y, index) with the square test (src/Frame.cc:413-466)."""
    cols, rows = grid
    min_x, min_y = f32(bounds[0]), f32(bounds[2])
    lo_x = max(0, int(np.floor(f32(f32(f32(u - min_x) - r) * invw))))
    hi_x = min(cols - 1, int(np.ceil(f32(f32(f32(u - min_x) + r) * invw))))
    lo_y = max(0, int(np.floor(f32(f32(f32(w - min_y) - r) * invh))))
    hi_y = min(rows - 1, int(np.ceil(f32(f32(f32(w - min_y) + r) * invh))))
    if lo_x >= cols or hi_x < 0 or lo_y >= rows or hi_y < 0:
        return []
    cand = sorted((c[0], c[1], i) for i, c in enumerate(cell) if c and lo_x <= c[0] <= hi_x and lo_y <= c[1] <= hi_y)
    return [i for _, _, i in cand if abs(f32(kp[i]["x"] - u)) < r and abs(f32(kp[i]["y"] - w)) < r]


def hamming(a, b):
    return int(np.unpackbits(a ^ b).sum())


def py_sim3_direction(kpT, descT, sfT, D, pts, mpd, th):
    bounds = (D["min_x"], D["max_x"], D["min_y"], D["max_y"])
    cell, invw, invh = cells_of(kpT, bounds, GRID)
    out = []
    for p, d in zip(pts, mpd):
        if p["skip"] or p["bad"]:
            out.append(-1)
            continue
        a = mat_apply(D["rcw"], D["tcw"], [f32(p["x"]), f32(p["y"]), f32(p["z"])])
        b = mat_apply(D["sr"], D["t"], a)
        if b[2] < 0:
            out.append(-1)
            continue
        with np.errstate(divide="ignore", invalid="ignore"):
            invz = f32(f32(1) / b[2])
            u = f32(f32(f32(D["fx"]) * f32(b[0] * invz)) + f32(D["cx"]))
            w = f32(f32(f32(D["fy"]) * f32(b[1] * invz)) + f32(D["cy"]))
        if not (u >= f32(bounds[0]) and u < f32(bounds[1]) and w >= f32(bounds[2]) and w < f32(bounds[3])):
            out.append(-1)
            continue
        dist = norm3(b)
        if dist < f32(f32(0.9) * f32(p["minDistance"])) or dist > f32(f32(1.1) * f32(p["maxDistance"])):
            out.append(-1)
            continue
        lvl = py_predict_scale(p["maxDistance"], dist, D["log_scale_factor"], D["n_levels"])
        r = f32(f32(th) * sfT[lvl])
        best, best_i = 257, -1
        for i in area_candidates(kpT, cell, invw, invh, bounds, GRID, u, w, r):
            if kpT[i]["octave"] < lvl - 1 or kpT[i]["octave"] > lvl:
                continue
            dd = hamming(descT[i], d)
            if dd < best:
                best, best_i = dd, i
        out.append(best_i if best <= 100 else -1)
    return out


def py_reloc(kp, desc, sf, v, pts, mpd, kf_angle, has_mp, th, check):
    bounds = (v["min_x"], v["max_x"], v["min_y"], v["max_y"])
    cell, invw, invh = cells_of(kp, bounds, GRID)
    taken = [bool(x) for x in (has_mp if has_mp is not None else np.zeros(len(kp)))]
    match = [-1] * len(kp)
    hist = [[] for _ in range(30)]
    n = 0
    for i, (p, d) in enumerate(zip(pts, mpd)):
        if p["skip"] or p["bad"]:
            continue
        pc = mat_apply(v["rcw"], v["tcw"], [f32(p["x"]), f32(p["y"]), f32(p["z"])])
        with np.errstate(divide="ignore", invalid="ignore"):
            u = f32(f32(f32(f32(v["fx"]) * pc[0]) / pc[2]) + f32(v["cx"]))
            w = f32(f32(f32(f32(v["fy"]) * pc[1]) / pc[2]) + f32(v["cy"]))
        if u < f32(bounds[0]) or u > f32(bounds[1]) or w < f32(bounds[2]) or w > f32(bounds[3]) or u != u or w != w:
            continue
        dist = norm3([f32(f32(p[k]) - f32(c)) for k, c in zip("xyz", v["twc"])])
        if dist < f32(f32(0.9) * f32(p["minDistance"])) or dist > f32(f32(1.1) * f32(p["maxDistance"])):
            continue
        lvl = py_predict_scale(p["maxDistance"], dist, v["log_scale_factor"], v["n_levels"])
        r = f32(f32(th) * sf[lvl])
        best, best_i = 256, -1
        for i2 in area_candidates(kp, cell, invw, invh, bounds, GRID, u, w, r):
            if kp[i2]["octave"] < lvl - 1 or kp[i2]["octave"] > lvl + 1 or taken[i2]:
                continue
            dd = hamming(desc[i2], d)
            if dd < best:
                best, best_i = dd, i2
        if best <= 100:
            match[best_i] = i
            taken[best_i] = True
            n += 1
            if check:
                rot = f32(f32(kf_angle[i]) - kp[best_i]["angle"])
                if rot < 0:
                    rot = f32(rot + f32(360))
                x = float(f32(rot * f32(f32(1) / f32(30))))
                b = int(math.floor(abs(x) + 0.5)) * (1 if x >= 0 else -1)  # roundf: half away from zero
                hist[0 if b == 30 else b].append(best_i)
    if check:
        sizes = [len(h) for h in hist]
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for b, sz in enumerate(sizes):  # ComputeThreeMaxima :1328-1370
            if sz > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, sz, i2, i1, b
            elif sz > m2:
                m3, m2, i3, i2 = m2, sz, i2, b
            elif sz > m3:
                m3, i3 = sz, b
        if f32(m2) < f32(f32(0.1) * f32(m1)):
            i2 = i3 = -1
        elif f32(m3) < f32(f32(0.1) * f32(m1)):
            i3 = -1
        keep = {i1, i2, i3}
        for b in range(30):
            if b not in keep:
                for j in hist[b]:
                    match[j] = -1
                    n -= 1
    return n, match


# ---------------------------------------------------------------------------------------------
# scenarios
# ---------------------------------------------------------------------------------------------
def extraction(seed):
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, seed))
    return eo, kp, desc


def pose(seed):
    rng = np.random.default_rng(seed)
    R = FS.rot(*(rng.uniform(-0.15, 0.15, 3))).astype(np.float64)
    t = rng.uniform(-0.4, 0.4, 3)
    return R, t


def sim3_scenario(kp1, desc1, sf, seed, n_levels=8):
    """Key frame 2 sees the map points of key frame 1 from another pose: its keypoints are the projections of those
    points (noise, bit flips, shuffled, some dropped); S12 is the exact relative pose times a scale near one."""
    rng = np.random.default_rng(seed)
    n1 = len(kp1)
    R1, t1 = pose(seed + 100)
    R2, t2 = pose(seed + 200)
    z = rng.uniform(2.0, 9.0, n1)
    pc1 = np.stack([(kp1["x"] - K["cx"]) / K["fx"] * z, (kp1["y"] - K["cy"]) / K["fy"] * z, z], 1).astype(np.float64)
    pw = (R1.T @ (pc1 - t1).T).T
    pc2 = (R2 @ pw.T).T + t2
    u2 = K["fx"] * pc2[:, 0] / pc2[:, 2] + K["cx"] + rng.normal(0, 0.8, n1)
    v2 = K["fy"] * pc2[:, 1] / pc2[:, 2] + K["cy"] + rng.normal(0, 0.8, n1)
    ok = (pc2[:, 2] > 0.1) & (u2 > 8) & (u2 < W - 8) & (v2 > 8) & (v2 < H - 8) & (rng.random(n1) < 0.9)
    src2 = np.flatnonzero(ok)
    rng.shuffle(src2)
    n2 = len(src2)
    kp2 = kp1[src2].copy()
    kp2["x"], kp2["y"] = u2[src2], v2[src2]
    desc2 = (np.stack([S.flip_bits(desc1[s], int(rng.integers(0, 30)), rng) for s in src2]) if n2
             else np.zeros((0, 32), np.uint8))

    def points(n, src_world, octave, centre, keep):
        p = np.zeros(n, O.WP_DTYPE)
        p["x"], p["y"], p["z"] = src_world[:, 0], src_world[:, 1], src_world[:, 2]
        d = np.linalg.norm(src_world - centre, axis=1)
        p["maxDistance"] = d * np.float32(1.2) ** octave.astype(np.float32) * rng.uniform(0.92, 1.08, n)
        p["minDistance"] = p["maxDistance"] / np.float32(1.2) ** 7
        p["bad"] = rng.random(n) < 0.02
        p["skip"] = rng.random(n) > keep  # no map point / already matched
        return p

    c2 = -(R2.T @ t2)
    c1 = -(R1.T @ t1)
    mp1 = points(n1, pw, kp1["octave"], c2, 0.85)
    mp2 = points(n2, pw[src2] + rng.normal(0, 0.003, (n2, 3)), kp2["octave"], c1, 0.85)
    # S12 = T1w * T2w^-1 with a scale: x1 = s * R12 x2 + t12
    s12 = 1.0 + rng.uniform(-0.01, 0.01)
    R12 = R1 @ R2.T
    t12 = t1 - R12 @ t2
    sR12, sR21 = s12 * R12, (1.0 / s12) * R12.T
    t21 = -(sR21 @ t12)

    def direction(D, names, Rs, ts, sR, tt):
        vals = dict(rcw=Rs.reshape(-1), tcw=ts, sr=sR.reshape(-1), t=tt, min_x=0.0, max_x=float(W), min_y=0.0,
                    max_y=float(H), log_scale_factor=float(np.log(np.float32(1.2))), n_levels=n_levels, **K)
        for k, val in vals.items():
            if k in ("rcw", "tcw", "sr", "t"):
                arr = getattr(D, names[k])
                for i, x in enumerate(np.asarray(val, np.float32)):
                    arr[i] = float(x)
                vals[k] = np.asarray(val, np.float32)
            else:
                setattr(D, names[k], val)
        return vals

    return dict(kp2=kp2, desc2=desc2, mp1=mp1, mp2=mp2, mpd1=desc1.copy(), mpd2=desc2.copy(), src2=src2,
                fill12=lambda D, names: direction(D, names, R1, t1, sR21, t21),
                fill21=lambda D, names: direction(D, names, R2, t2, sR12, t12))


SN_O = dict(rcw="rcw", tcw="tcw", sr="sr", t="t", fx="fx", fy="fy", cx="cx", cy="cy", min_x="minX", max_x="maxX",
            min_y="minY", max_y="maxY", log_scale_factor="logScaleFactor", n_levels="nLevels")
SN_P = {k: k for k in SN_O}


def reloc_scenario(kp, desc, sf, v, M, seed):
    pts, mpd, _, _ = scenario(kp, desc, sf, v, M, seed, False)
    rng = np.random.default_rng(seed + 7)
    # the map point of row i was back-projected from frame keypoint src[i]; recover src to derive an angle
    src = np.random.default_rng(seed).integers(0, len(kp), M)
    jump = rng.choice([0.0, 0.0, 0.0, 0.0, 95.0, 200.0, 310.0], M)
    ang = (kp["angle"][src] + jump + rng.uniform(-6, 6, M)) % 360.0
    has = (rng.random(len(kp)) < 0.05).astype(np.uint8)
    return pts, mpd, ang.astype(np.float32), has


# ---------------------------------------------------------------------------------------------
# CPU: oracle == restatement
# ---------------------------------------------------------------------------------------------
def test_oracle_fuse_sim3_matches_restatement(built):
    from test_fuse import py_fuse_one
    eo, kp, desc = extraction(6)
    Fo = O.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=21)
    pts, mpd, _, inv_s2 = scenario(kp, desc, eo.scaleFactors, v, 140, 5, False)
    fv = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    bi, bd = O.fuse_search_sim3(fv, Fo, 4.0, pts, mpd)
    bi_g, bd_g = O.fuse_search(fv, inv_s2, None, Fo, 4.0, pts, mpd)
    assert (bd <= 30).sum() > 30
    assert not (np.array_equal(bi, bi_g) and np.array_equal(bd, bd_g))  # the chi-square gate does bite in this scenario
    for i in range(len(pts)):
        assert py_fuse_one(kp, desc, eo.scaleFactors, inv_s2, None, v, 4.0, pts[i], mpd[i], chi2=False) == (int(bi[i]), int(bd[i])), i


def test_oracle_search_by_sim3_matches_restatement(built):
    eo, kp1, desc1 = extraction(9)
    kp1, desc1 = kp1[:260], desc1[:260]
    sc = sim3_scenario(kp1, desc1, eo.scaleFactors, 3)
    d12, d21 = O.Sim3Dir(), O.Sim3Dir()
    v12, v21 = sc["fill12"](d12, SN_O), sc["fill21"](d21, SN_O)
    fv1 = O.make_frame_view(kp1, desc1, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    fv2 = O.make_frame_view(sc["kp2"], sc["desc2"], 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    n, m12 = O.search_by_sim3(fv1, fv2, d12, d21, sc["mp1"], sc["mpd1"], sc["mp2"], sc["mpd2"], 7.5)
    vn1 = py_sim3_direction(sc["kp2"], sc["desc2"], eo.scaleFactors, v12, sc["mp1"], sc["mpd1"], 7.5)
    vn2 = py_sim3_direction(kp1, desc1, eo.scaleFactors, v21, sc["mp2"], sc["mpd2"], 7.5)
    want = [i2 if i2 >= 0 and vn2[i2] == i1 else -1 for i1, i2 in enumerate(vn1)]
    assert list(m12) == want and n == sum(w >= 0 for w in want)
    assert n > 60
    # found pairs are the construction's pairs
    hit = np.flatnonzero(m12 >= 0)
    assert (sc["src2"][m12[hit]] == hit).mean() > 0.95


@pytest.mark.parametrize("check", [True, False])
def test_oracle_reloc_matches_restatement(built, check):
    eo, kp, desc = extraction(11)
    kp, desc = kp[:400], desc[:400]
    Fo = O.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=31)
    pts, mpd, ang, has = reloc_scenario(kp, desc, eo.scaleFactors, v, 220, 4)
    fv = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    n, m = O.search_by_projection_kf(fv, Fo, pts, mpd, ang, has, 12.0, check)
    n_py, m_py = py_reloc(kp, desc, eo.scaleFactors, v, pts, mpd, ang, has, 12.0, check)
    assert n == n_py and list(m) == m_py
    assert n > 60
    if check:
        n_all, _ = O.search_by_projection_kf(fv, Fo, pts, mpd, ang, has, 12.0, False)
        assert n < n_all  # the rotation histogram removed something
    # behind-the-camera points are matched by this overload (no depth test in the reference)
    R, t = np.asarray(v["rcw"], np.float64).reshape(3, 3), np.asarray(v["tcw"], np.float64)
    zc = (R @ np.stack([pts["x"], pts["y"], pts["z"]]).astype(np.float64)).T[:, 2] + t[2]
    assert any(zc[i] < 0 for i in m if i >= 0)


# ---------------------------------------------------------------------------------------------
# GPU: HIP path == oracle
# ---------------------------------------------------------------------------------------------
@pytest.mark.gpu
@pytest.mark.parametrize("M,th,seed,kb8", [(2000, 4.0, 1, False), (900, 10.0, 2, False), (1, 3.0, 3, False), (1500, 4.0, 4, True)])
def test_gpu_fuse_sim3_matches_oracle(built, M, th, seed, kb8):
    import orbfe
    eo, kp, desc = extraction(6 + seed)
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=20 + seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=20 + seed, kb8=kb8)
    pts, mpd, _, _ = scenario(kp, desc, eo.scaleFactors, v, M, seed, False)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    bi_r, bd_r = O.fuse_search_sim3(fvo, Fo, th, pts, mpd)
    bi, bd = m.Fuse_search_sim3(fv, Fp, th, pts.view(orbfe.WP_DTYPE), mpd)
    assert np.array_equal(bd, bd_r) and np.array_equal(bi, bi_r)
    assert len(m.Fuse_search_sim3(fv, Fp, th, pts[:0].view(orbfe.WP_DTYPE), mpd[:0])[0]) == 0


@pytest.mark.gpu
@pytest.mark.parametrize("seed,th,n_keep", [(1, 7.5, None), (2, 7.5, None), (3, 15.0, None), (4, 7.5, 300), (5, 7.5, 1), (6, 7.5, 3)])
def test_gpu_search_by_sim3_matches_oracle(built, seed, th, n_keep):
    import orbfe
    eo, kp1, desc1 = extraction(20 + seed)
    if n_keep:
        kp1, desc1 = kp1[:n_keep], desc1[:n_keep]
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    sc = sim3_scenario(kp1, desc1, eo.scaleFactors, seed)
    d12o, d21o, d12, d21 = O.Sim3Dir(), O.Sim3Dir(), orbfe.Sim3View(), orbfe.Sim3View()
    sc["fill12"](d12o, SN_O), sc["fill21"](d21o, SN_O), sc["fill12"](d12, SN_P), sc["fill21"](d21, SN_P)
    mk = lambda mod, sf: (mod.make_frame_view(kp1, desc1, 64, 48, 0.0, 0.0, float(W), float(H), sf),
                          mod.make_frame_view(sc["kp2"], sc["desc2"], 64, 48, 0.0, 0.0, float(W), float(H), sf))
    fv1o, fv2o = mk(O, eo.scaleFactors)
    fv1, fv2 = mk(orbfe, ex.mvScaleFactor)
    n_r, m_r = O.search_by_sim3(fv1o, fv2o, d12o, d21o, sc["mp1"], sc["mpd1"], sc["mp2"], sc["mpd2"], th)
    n, m12 = m.SearchBySim3(fv1, fv2, d12, d21, sc["mp1"].view(orbfe.WP_DTYPE), sc["mpd1"], sc["mp2"].view(orbfe.WP_DTYPE),
                            sc["mpd2"], th)
    assert n == n_r and np.array_equal(m12, m_r)
    if n_keep is None:
        assert n > 200


@pytest.mark.gpu
@pytest.mark.parametrize("M,th,seed,check,kb8", [(2000, 12.0, 1, True, False), (2000, 12.0, 2, False, False),
                                                 (5000, 25.0, 3, True, False), (1, 12.0, 4, True, False),
                                                 (1500, 12.0, 5, True, True), (3000, 60.0, 6, True, False),
                                                 (40000, 6.0, 7, True, False)])  # >= 128 blocks: the thread-per-map-point top-K
def test_gpu_reloc_projection_matches_oracle(built, M, th, seed, check, kb8):
    import orbfe
    eo, kp, desc = extraction(40 + seed)
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=50 + seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=50 + seed, kb8=kb8)
    pts, mpd, ang, has = reloc_scenario(kp, desc, eo.scaleFactors, v, M, seed)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    n_r, m_r = O.search_by_projection_kf(fvo, Fo, pts, mpd, ang, has, th, check)
    n, mm = m.SearchByProjection_keyframe(fv, Fp, pts.view(orbfe.WP_DTYPE), mpd, ang, has, th, check)
    assert n == n_r and np.array_equal(mm, m_r)
    if M >= 1000 and not kb8:
        assert n > 150
    # no occupied slots / no points
    n0, m0 = m.SearchByProjection_keyframe(fv, Fp, pts[:0].view(orbfe.WP_DTYPE), mpd[:0], ang[:0], None, th, check)
    assert n0 == 0 and (m0 == -1).all()
    n1r, m1r = O.search_by_projection_kf(fvo, Fo, pts, mpd, ang, None, th, check)
    n1, m1 = m.SearchByProjection_keyframe(fv, Fp, pts.view(orbfe.WP_DTYPE), mpd, ang, None, th, check)
    assert n1 == n1r and np.array_equal(m1, m1r)


@pytest.mark.gpu
@pytest.mark.parametrize("seed", range(8))
def test_gpu_random_configurations(built, seed):
    """Random frame sizes, pyramid depths, grids (8x6 ... 200x150), radii and map-point counts through the three entry
    points: HIP == oracle on each."""
    import orbfe
    from orbfe import synth
    rng = np.random.default_rng(1000 + seed)
    w, h = int(rng.integers(160, 900)), int(rng.integers(120, 600))
    levels = int(rng.integers(1, 9))
    while min(w, h) / 1.2 ** (levels - 1) < 50:
        levels -= 1
    args = (int(rng.integers(200, 1500)), 30000, 1.2, levels, 20, 7, w, h)
    cols, rows = int(rng.integers(8, 201)), int(rng.integers(6, 151))
    eo = O.Extractor(*args)
    kp, desc, _ = eo.extract(synth.frame(w, h, 70 + seed))
    assert len(kp) > 20
    ex = orbfe.ORBextractor(*args)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, W=float(w), H=float(h), n_levels=levels, seed=60 + seed)
    FS.fill_frustum(Fp, PN, W=float(w), H=float(h), n_levels=levels, seed=60 + seed)
    for F in (Fo, Fp):  # principal point inside this frame
        F.cx, F.cy = 0.49 * w, 0.52 * h
    v["cx"], v["cy"] = 0.49 * w, 0.52 * h
    M = int(rng.integers(1, 3000))
    th = float(rng.uniform(2.0, 40.0))
    pts, mpd, ang, has = reloc_scenario(kp, desc, eo.scaleFactors, v, M, seed)
    fvo = O.make_frame_view(kp, desc, cols, rows, 0.0, 0.0, float(w), float(h), eo.scaleFactors)
    fv = orbfe.make_frame_view(kp, desc, cols, rows, 0.0, 0.0, float(w), float(h), ex.mvScaleFactor)
    check = bool(seed & 1)
    n_r, m_r = O.search_by_projection_kf(fvo, Fo, pts, mpd, ang, has, th, check)
    n, mm = m.SearchByProjection_keyframe(fv, Fp, pts.view(orbfe.WP_DTYPE), mpd, ang, has, th, check)
    assert n == n_r and np.array_equal(mm, m_r)
    bi_r, bd_r = O.fuse_search_sim3(fvo, Fo, th, pts, mpd)
    bi, bd = m.Fuse_search_sim3(fv, Fp, th, pts.view(orbfe.WP_DTYPE), mpd)
    assert np.array_equal(bd, bd_r) and np.array_equal(bi, bi_r)
    # SearchBySim3 on the same key frame pair construction, this frame size / grid
    global W, H
    W0, H0 = W, H
    try:
        W, H = w, h  # the scenario keeps key frame 2's keypoints inside these bounds
        sc = sim3_scenario(kp, desc, eo.scaleFactors, seed, n_levels=levels)
    finally:
        W, H = W0, H0
    d12o, d21o, d12, d21 = O.Sim3Dir(), O.Sim3Dir(), orbfe.Sim3View(), orbfe.Sim3View()
    for D, names in ((d12o, SN_O), (d12, SN_P)):
        sc["fill12"](D, names)
        setattr(D, names["max_x"], float(w)), setattr(D, names["max_y"], float(h))
    for D, names in ((d21o, SN_O), (d21, SN_P)):
        sc["fill21"](D, names)
        setattr(D, names["max_x"], float(w)), setattr(D, names["max_y"], float(h))
    if len(sc["kp2"]) == 0:
        return
    fv2o = O.make_frame_view(sc["kp2"], sc["desc2"], cols, rows, 0.0, 0.0, float(w), float(h), eo.scaleFactors)
    fv2 = orbfe.make_frame_view(sc["kp2"], sc["desc2"], cols, rows, 0.0, 0.0, float(w), float(h), ex.mvScaleFactor)
    ns_r, ms_r = O.search_by_sim3(fvo, fv2o, d12o, d21o, sc["mp1"], sc["mpd1"], sc["mp2"], sc["mpd2"], th)
    ns, ms = m.SearchBySim3(fv, fv2, d12, d21, sc["mp1"].view(orbfe.WP_DTYPE), sc["mpd1"], sc["mp2"].view(orbfe.WP_DTYPE),
                            sc["mpd2"], th)
    assert ns == ns_r and np.array_equal(ms, ms_r)
