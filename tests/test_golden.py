"""Committed golden vectors (tests/golden/*.npz, made by tests/golden/make_golden.py from the oracle):
the oracle must keep reproducing them (no GPU), and the HIP path must reproduce them (GPU)."""
import glob
import os

import numpy as np
import pytest

import oracle_py as O

HERE = os.path.dirname(os.path.abspath(__file__))
FIXTURES = sorted(glob.glob(os.path.join(HERE, "golden", "*.npz")))
assert len(FIXTURES) >= 3


def _args(g):
    a = g["args"]
    return (int(a[0]), int(a[1]), float(a[2]), int(a[3]), int(a[4]), int(a[5]), int(a[6]), int(a[7]))


@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_oracle_reproduces_golden(path):
    g = np.load(path)
    args = _args(g)
    e = O.Extractor(*args)
    kp, desc, per = e.extract(g["image"])
    assert kp.tobytes() == g["kp"].tobytes() and np.array_equal(desc, g["desc"]) and np.array_equal(per, g["per_level"])
    assert np.array_equal(e.level_image(min(1, args[3] - 1), False), g["level1"])
    assert np.array_equal(e.level_image(0, True), g["blur0"])
    if "proj_n" in g:
        fv = O.make_frame_view(kp, desc, 16, 12, 0.0, 0.0, float(args[6]), float(args[7]), e.scaleFactors)
        n, m = O.search_by_projection(fv, g["mps"], g["mpd"], g["init_obs"], 20.0, 0.85)
        assert n == int(g["proj_n"]) and np.array_equal(m, g["proj_match"])
        nb, mb = O.search_by_bow(g["kfOff"], g["kfIdx"], g["fOff"], g["fIdx"], desc, kp["angle"], g["has"], desc,
                                 kp["angle"][::-1].copy(), 0.75, True)
        assert nb == int(g["bow_n"]) and np.array_equal(mb, g["bow_match"])


@pytest.mark.gpu
@pytest.mark.parametrize("path", FIXTURES, ids=[os.path.basename(p)[:-4] for p in FIXTURES])
def test_hip_reproduces_golden(built, path):
    import orbfe
    g = np.load(path)
    args = _args(g)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    got = ex.extractFeatures(g["image"])
    assert got is not None
    kp, desc = got
    assert kp.tobytes() == g["kp"].tobytes() and np.array_equal(desc, g["desc"])
    assert np.array_equal(ex.last_per_level, g["per_level"])
    assert np.array_equal(ex.pyramid_level(min(1, args[3] - 1), False), g["level1"])
    assert np.array_equal(ex.pyramid_level(0, True), g["blur0"])
    if "proj_n" in g:
        m = orbfe.ORBmatcher(ex)
        fv = orbfe.make_frame_view(kp, desc, 16, 12, 0.0, 0.0, float(args[6]), float(args[7]), ex.mvScaleFactor)
        n, out = m.SearchByProjection(fv, g["mps"].view(orbfe.MP_DTYPE), g["mpd"], 20.0, False, 0.0, 0.85, g["init_obs"])
        assert n == int(g["proj_n"]) and np.array_equal(out, g["proj_match"])
        nb, mb = m.SearchByBoW(g["kfOff"], g["kfIdx"], g["fOff"], g["fIdx"], desc, kp["angle"], g["has"], desc,
                               kp["angle"][::-1].copy(), 0.75, True)
        assert nb == int(g["bow_n"]) and np.array_equal(mb, g["bow_match"])
