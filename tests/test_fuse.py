"""SURVEY.md section 8f row f2 (first half): the search part of ORBmatcher::Fuse(pKF, vpMapPoints, th)
(src/ORBmatcher.cc:678-836)."""
import numpy as np
import pytest

import frustum_scenarios as FS
import match_scenarios as S
import oracle_py as O
from test_frustum import ON, PN, py_in_frustum  # noqa: F401

W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)


def scenario(kp, desc, sf, v, M, seed, stereo):
    """Map points back-projected from key-frame keypoints (image coordinates == the key frame's keypoint
    coordinates, as Fuse compares them), plus outliers: behind the camera, outside the image, wrong distance."""
    rng = np.random.default_rng(seed)
    src = rng.integers(0, len(kp), M)
    u = kp["x"][src] + rng.normal(0, 1.2, M)
    w = kp["y"][src] + rng.normal(0, 1.2, M)
    z = rng.uniform(1.5, 9.0, M)
    z[rng.random(M) < 0.04] *= -1
    pc = np.stack([(u - v["cx"]) / v["fx"] * z, (w - v["cy"]) / v["fy"] * z, z], 1)
    R = np.asarray(v["rcw"], np.float64).reshape(3, 3)
    pw = (R.T @ (pc - np.asarray(v["tcw"], np.float64)).T).T
    pts = np.zeros(M, O.WP_DTYPE)
    pts["x"], pts["y"], pts["z"] = pw[:, 0], pw[:, 1], pw[:, 2]
    d = np.linalg.norm(pw - np.asarray(v["twc"], np.float64), axis=1)
    pts["maxDistance"] = d * np.float32(1.2) ** kp["octave"][src].astype(np.float32) * rng.uniform(0.9, 1.1, M)
    pts["minDistance"] = pts["maxDistance"] / np.float32(1.2) ** 7
    far = rng.random(M) < 0.05
    pts["maxDistance"][far] *= 0.3
    pts["observations"] = rng.integers(0, 4, M)
    pts["bad"] = rng.random(M) < 0.02
    pts["skip"] = rng.random(M) < 0.05
    mpd = np.stack([S.flip_bits(desc[s], int(rng.integers(0, 40)), rng) for s in src])
    u_right = None
    if stereo:
        u_right = np.where(rng.random(len(kp)) < 0.5, kp["x"] - rng.uniform(2, 30, len(kp)), -1.0).astype(np.float32)
    inv_sigma2 = (1.0 / (sf.astype(np.float32) ** 2)).astype(np.float32)
    return pts, mpd, u_right, inv_sigma2


def py_fuse_one(kp, desc, sf, inv_s2, u_right, v, th, p, d, grid=(64, 48), chi2=True, n_right=None):
    """Plain restatement of :722-826 for one map point on top of the pinned isInFrustum arithmetic; candidates are
    visited in GetFeaturesInArea order (cell x, cell y, index) with the strict '<' of :819."""
    f32 = np.float32
    if p["skip"] or p["bad"]:
        return -1, 256
    o, _ = py_in_frustum(dict(v, min_x=-1e30, max_x=1e30, min_y=-1e30, max_y=1e30), {k: p[k] for k in p.dtype.names})
    # py_in_frustum rejects by distance / depth exactly as Fuse does; the image test differs (IsInImage, half-open)
    if o["projX"] == -1 and o["projY"] == -1 and not o["inView"]:
        return -1, 256
    u, w = o["projX"], o["projY"]
    if not (u >= f32(v["min_x"]) and u < f32(v["max_x"]) and w >= f32(v["min_y"]) and w < f32(v["max_y"])):
        return -1, 256
    if not o["inView"]:
        return -1, 256
    lvl = o["level"]
    r = f32(f32(th) * sf[lvl])
    cols, rows = grid
    invw, invh = f32(cols) / f32(f32(v["max_x"]) - f32(v["min_x"])), f32(rows) / f32(f32(v["max_y"]) - f32(v["min_y"]))
    cell = []
    for k in kp:
        px, py = int(np.round(f32(f32(k["x"] - f32(v["min_x"])) * invw))), int(np.round(f32(f32(k["y"] - f32(v["min_y"])) * invh)))
        lin = py * cols + px
        cell.append((lin % cols, lin // cols) if 0 <= lin < cols * rows else None)
    lo_x = max(0, int(np.floor(f32(f32(f32(u - f32(v["min_x"])) - r) * invw))))
    hi_x = min(cols - 1, int(np.ceil(f32(f32(f32(u - f32(v["min_x"])) + r) * invw))))
    lo_y = max(0, int(np.floor(f32(f32(f32(w - f32(v["min_y"])) - r) * invh))))
    hi_y = min(rows - 1, int(np.ceil(f32(f32(f32(w - f32(v["min_y"])) + r) * invh))))
    if lo_x >= cols or hi_x < 0 or lo_y >= rows or hi_y < 0:
        return -1, 256
    cand = sorted((c[0], c[1], i) for i, c in enumerate(cell) if c and lo_x <= c[0] <= hi_x and lo_y <= c[1] <= hi_y)
    best, best_i = 256, -1
    for _, _, i in cand:
        k = kp[i]
        if not (abs(f32(k["x"] - u)) < r and abs(f32(k["y"] - w)) < r):
            continue
        if k["octave"] < lvl - 1 or k["octave"] > lvl:
            continue
        ex, ey = f32(u - k["x"]), f32(w - k["y"])
        assert u_right is None  # the stereo gate (:792-805) is covered by the GPU == oracle comparison
        e2 = f32(f32(ex * ex) + f32(ey * ey))
        if chi2 and float(f32(e2 * inv_s2[k["octave"]])) > 5.99:  # the Sim3 overload (:864-975) has no gate
            continue
        row = i
        if n_right is not None:   # bRight (:820): desc holds NLeft + n_right rows, the right twin's row is compared
            if i >= n_right:
                continue
            row = i + len(kp)
        dist = int(np.unpackbits(desc[row] ^ d).sum())
        if dist < best:
            best, best_i = dist, row
    return best_i, best


def test_oracle_fuse_matches_restatement(built):
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 6))
    Fo = O.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=21)
    pts, mpd, _, inv_s2 = scenario(kp, desc, eo.scaleFactors, v, 160, 5, False)
    fv = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    bi, bd = O.fuse_search(fv, inv_s2, None, Fo, 3.0, pts, mpd)
    assert (bd <= 30).sum() > 40
    for i in range(len(pts)):
        assert py_fuse_one(kp, desc, eo.scaleFactors, inv_s2, None, v, 3.0, pts[i], mpd[i]) == (int(bi[i]), int(bd[i])), i


@pytest.mark.gpu
@pytest.mark.parametrize("M,th,stereo,seed,kb8", [(2000, 3.0, False, 1, False), (1500, 3.0, True, 2, False), (700, 8.0, True, 3, False),
                                                  (1, 3.0, False, 4, False), (1500, 4.0, False, 5, True)])
def test_gpu_fuse_search_matches_oracle(built, M, th, stereo, seed, kb8):
    import orbfe
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 6 + seed))
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=20 + seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=20 + seed, kb8=kb8)
    pts, mpd, u_right, inv_s2 = scenario(kp, desc, eo.scaleFactors, v, M, seed, stereo)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    bi_r, bd_r = O.fuse_search(fvo, inv_s2, u_right, Fo, th, pts, mpd)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    bi, bd = m.Fuse_search(fv, inv_s2, u_right, Fp, th, pts.view(orbfe.WP_DTYPE), mpd)
    assert np.array_equal(bd, bd_r) and np.array_equal(bi, bi_r)
    if M > 100 and not kb8:  # the scenario back-projects with the pinhole model
        assert (bd_r <= 30).sum() > M // 8
    # empty inputs
    bi0, bd0 = m.Fuse_search(fv, inv_s2, u_right, Fp, th, pts[:0].view(orbfe.WP_DTYPE), mpd[:0])
    assert len(bi0) == 0


def right_scenario(kp, desc, seed, n_right):
    """mDescriptors of a two-camera key frame: NLeft left rows followed by n_right right rows (right twin i of left
    feature i: a few flipped bits, some unrelated)"""
    rng = np.random.default_rng(seed)
    right = np.stack([S.flip_bits(desc[i], int(rng.integers(0, 50)), rng) for i in range(n_right)])
    bad = rng.random(n_right) < 0.2
    right[bad] = rng.integers(0, 256, (int(bad.sum()), 32), dtype=np.uint8)
    return np.ascontiguousarray(np.concatenate([desc, right]))


def test_oracle_fuse_right_matches_restatement(built):
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 9))
    n_right = len(kp) - 150   # fewer right than left features: rows beyond the matrix are skipped
    all_desc = right_scenario(kp, desc, 3, n_right)
    Fo = O.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=27)
    pts, mpd, _, inv_s2 = scenario(kp, desc, eo.scaleFactors, v, 160, 8, False)
    fv = O.make_frame_view(kp, all_desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    bi, bd = O.fuse_search_right(fv, n_right, inv_s2, None, Fo, 3.0, pts, mpd)
    assert (bd <= 30).sum() > 15 and bi[bi >= 0].min() >= len(kp) and bi.max() < len(kp) + n_right
    for i in range(len(pts)):
        assert py_fuse_one(kp, all_desc, eo.scaleFactors, inv_s2, None, v, 3.0, pts[i], mpd[i], n_right=n_right) == (int(bi[i]), int(bd[i])), i
    # no right features at all
    bi0, bd0 = O.fuse_search_right(fv, 0, inv_s2, None, Fo, 3.0, pts, mpd)
    assert (bi0 == -1).all() and (bd0 == 256).all()


@pytest.mark.gpu
@pytest.mark.parametrize("M,th,stereo,seed,kb8,short", [(1500, 3.0, False, 1, False, 0), (1200, 4.0, True, 2, True, 150), (600, 8.0, False, 3, True, 700)])
def test_gpu_fuse_search_right_matches_oracle(built, M, th, stereo, seed, kb8, short):
    import orbfe
    from orbfe import synth
    eo = O.Extractor(*ARGS)
    kp, desc, _ = eo.extract(synth.frame(W, H, 16 + seed))
    n_right = len(kp) - short
    all_desc = right_scenario(kp, desc, seed, n_right)
    ex = orbfe.ORBextractor(*ARGS)
    m = orbfe.ORBmatcher(ex)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, seed=40 + seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=40 + seed, kb8=kb8)
    pts, mpd, u_right, inv_s2 = scenario(kp, desc, eo.scaleFactors, v, M, seed, stereo)
    fvo = O.make_frame_view(kp, all_desc, 64, 48, 0.0, 0.0, float(W), float(H), eo.scaleFactors)
    bi_r, bd_r = O.fuse_search_right(fvo, n_right, inv_s2, u_right, Fo, th, pts, mpd)
    fv = orbfe.make_frame_view(kp, all_desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    bi, bd = m.Fuse_search_right(fv, n_right, inv_s2, u_right, Fp, th, pts.view(orbfe.WP_DTYPE), mpd)
    assert np.array_equal(bd, bd_r) and np.array_equal(bi, bi_r)
    assert (bi_r >= len(kp)).sum() > 20 and bi_r.max() < len(kp) + n_right
    # the left call on the same view is unchanged by the extra rows
    bi_l, bd_l = m.Fuse_search(fv, inv_s2, u_right, Fp, th, pts.view(orbfe.WP_DTYPE), mpd)
    bi_lr, bd_lr = O.fuse_search(fvo, inv_s2, u_right, Fo, th, pts, mpd)
    assert np.array_equal(bi_l, bi_lr) and np.array_equal(bd_l, bd_lr)
    bi0, bd0 = m.Fuse_search_right(fv, 0, inv_s2, u_right, Fp, th, pts.view(orbfe.WP_DTYPE), mpd)
    assert (bi0 == -1).all() and (bd0 == 256).all()
