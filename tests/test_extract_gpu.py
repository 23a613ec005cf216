"""GPU parity: HIP path (through the C ABI) vs the CPU oracle, bit-exact, stage by stage."""
import numpy as np
import pytest

import oracle_py as O

pytestmark = pytest.mark.gpu

CONFIGS = {
    # name: (nFeatures, nFast, scale, levels, iniTh, minTh, W, H)
    "c1_752x480": (1000, 40000, 1.2, 8, 20, 7, 752, 480),
    "small_160x120": (200, 8000, 1.2, 4, 20, 7, 160, 120),
    "odd_333x251": (500, 30000, 1.2, 6, 20, 7, 333, 251),
    "c5_1024sq": (1500, 100000, 1.2, 12, 20, 7, 1024, 1024),
    "c4_1280x720": (2000, 100000, 1.2, 8, 20, 7, 1280, 720),
    "ref_node_1level": (10000, 16000, 2.0, 1, 100, 80, 614, 460),  # mono_inertial_node.cpp:87-93
    # pre-NMS / final caps active (S2b): SURVEY's default nFast = 16 * nFeatures, and a tight one
    "c1_nfast16k": (1000, 16000, 1.2, 8, 20, 7, 752, 480),
    "cap_tight": (500, 2500, 1.2, 6, 20, 7, 333, 251),
    "cap_hi": (300, 900, 1.2, 4, 9, 7, 320, 240),
    "ref_gnss_6level": (50000, 86000, 1.2, 6, 40, 35, 752, 480),  # mono_inertial_gnss_node.cpp:96-101
    "dense_1level": (10000, 60000, 1.2, 1, 12, 7, 752, 480),       # > 2048 nodes actually reached in one level
}


def _mk(cfg, max_batch=1):
    import orbfe
    return orbfe.ORBextractor(*CONFIGS[cfg], device=0, max_batch=max_batch)


def _img(cfg, idx=0):
    from orbfe import synth
    a = CONFIGS[cfg]
    return synth.frame(a[6], a[7], idx)


def _unpack(packed):
    return packed & 0xFFF, (packed >> 12) & 0xFFF, packed >> 24


@pytest.mark.parametrize("cfg", ["small_160x120", "odd_333x251", "c1_752x480"])
def test_stages_match_oracle(built, cfg):
    a = CONFIGS[cfg]
    ex = _mk(cfg)
    ref = O.Extractor(*a)
    img = _img(cfg)
    got = ex.extractFeatures(img)
    kp_r, desc_r, per_r = ref.extract(img)
    # tables
    assert np.array_equal(ex.mvScaleFactor, ref.scaleFactors)
    assert np.array_equal(ex.mvInvScaleFactor, ref.invScaleFactors)
    assert np.array_equal(ex.mvLevelSigma2, ref.levelSigma2)
    assert np.array_equal(ex.mvInvLevelSigma2, ref.invLevelSigma2)
    assert np.array_equal(ex.mnFeaturesPerLevel, ref.featuresPerLevel)
    assert np.array_equal(ex.levelW, ref.levelW) and np.array_equal(ex.levelH, ref.levelH)
    for l in range(a[3]):
        assert np.array_equal(ex.pyramid_level(l, False), ref.level_image(l, False)), "pyramid level %d" % l
        assert np.array_equal(ex.pyramid_level(l, True), ref.level_image(l, True)), "blurred level %d" % l
        packed, cnt = ex.debug_candidates(l)
        xy, resp, pre = O.fast_detect(ref.level_image(l, False), a[5], 1 << 22)
        x, y, s = _unpack(np.sort(packed & 0xFFFFFF | (packed & 0xFF000000)))
        order = np.argsort((packed & 0xFFFFFF))
        x, y, s = _unpack(packed[order])
        assert cnt[2] == pre, "pre-NMS count level %d" % l
        assert len(packed) == len(resp), "NMS survivors level %d: %d vs %d" % (l, len(packed), len(resp))
        assert np.array_equal(x, xy[:, 0]) and np.array_equal(y, xy[:, 1]) and np.array_equal(s, resp)
        assert cnt[1] == int((resp >= a[4]).sum())
    assert got is not None
    kp, desc = got
    assert np.array_equal(ex.last_per_level, per_r)
    for fld in ("x", "y", "response", "size", "octave"):
        assert np.array_equal(kp[fld], kp_r[fld]), fld
    assert kp["angle"].tobytes() == kp_r["angle"].tobytes(), "angle bits"
    assert np.array_equal(desc, desc_r)


@pytest.mark.parametrize("cfg", ["c5_1024sq", "c4_1280x720", "c1_nfast16k", "cap_tight", "cap_hi", "ref_node_1level", "ref_gnss_6level", "dense_1level"])
def test_extract_bit_exact(built, cfg):
    ex = _mk(cfg)
    ref = O.Extractor(*CONFIGS[cfg])
    for idx in (0, 3):
        img = _img(cfg, idx)
        got = ex.extractFeatures(img)
        kp_r, desc_r, per_r = ref.extract(img)
        assert got is not None
        kp, desc = got
        assert np.array_equal(ex.last_per_level, per_r)
        assert kp.tobytes() == kp_r.tobytes()
        assert np.array_equal(desc, desc_r)


def test_batch_equals_single_and_is_deterministic(built):
    cfg = "small_160x120"
    from orbfe import synth
    a = CONFIGS[cfg]
    B = 6
    ex = _mk(cfg, max_batch=B)
    ims = [synth.frame(a[6], a[7], 10 + i) for i in range(B)]
    r1 = ex.extract_batch(ims)
    r2 = ex.extract_batch(ims)
    ref = O.Extractor(*a)
    for b in range(B):
        kp_r, desc_r, per_r = ref.extract(ims[b])
        assert r1[b][0].tobytes() == kp_r.tobytes() and np.array_equal(r1[b][1], desc_r)
        assert r1[b][0].tobytes() == r2[b][0].tobytes() and np.array_equal(r1[b][1], r2[b][1])
        assert np.array_equal(r1[b][2], per_r)


def test_blank_image_gives_nullopt(built):
    ex = _mk("small_160x120")
    assert ex.extractFeatures(np.full((120, 160), 77, np.uint8)) is None
