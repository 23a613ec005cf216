"""Parity on hostile image classes: the HIP path against the CPU oracle, bit for bit, on content the
rectangle / disc generator never produces -- white noise (maximum corner density: queue fill, candidate caps, quadtree
node tables), checkerboards of period 1..4 and lattices of identical blobs (equal scores under the strict `>` of the
NMS, src/cuda/Fast_gpu.cu:300-310), saturated frames with isolated extreme pixels (score 254, :193-216), contrasts of
exactly th / th + 1 (:60-65), step edges on the 64x32 tile seams and on the tested-region border (:275,365-368), and the
two streams `bench.py --texture-sweep` times (1/f noise, low texture).  Geometries: the headline 752x480 8-level
config, a 12-level one and a scale-2.0 one (the tile resize kernel instead of the strip kernel).
A mismatch is reported with the first stage that differs (pyramid pixels, blurred pixels, FAST candidates, counts)."""
import numpy as np
import pytest

import oracle_py as O
from orbfe import synth

pytestmark = pytest.mark.gpu

GEOM = {
    # name: (nFeatures, nFast, scale, levels, iniTh, minTh, W, H)
    "c1": (1000, 40000, 1.2, 8, 20, 7, 752, 480),
    "l12": (1500, 100000, 1.2, 12, 20, 7, 640, 480),
    "s20": (800, 30000, 2.0, 5, 20, 7, 640, 400),
}


def first_difference(ex, ref, args, frame=0):
    """Stage-by-stage comparison of the LAST extraction of both sides -> text naming the first stage that differs."""
    for l in range(args[3]):
        for blurred in (False, True):
            g, r = ex.pyramid_level(l, blurred, frame=frame), ref.level_image(l, blurred)
            if not np.array_equal(g, r):
                bad = np.argwhere(g != r)
                return "%s level %d: %d pixels differ, first at (y, x) = %s: gpu %d oracle %d" % (
                    "blurred" if blurred else "pyramid", l, len(bad), tuple(bad[0]), g[tuple(bad[0])], r[tuple(bad[0])])
        packed, cnt = ex.debug_candidates(l, frame=frame)
        xy, resp, pre = O.fast_detect(ref.level_image(l, False), args[5], 1 << 22)
        key_g = np.sort(packed & 0xFFFFFF)
        key_r = np.sort((xy[:, 1].astype(np.uint32) << 12) | xy[:, 0].astype(np.uint32)) if len(xy) else np.zeros(0, np.uint32)
        if cnt[2] != pre:
            return "level %d: pre-NMS corner count gpu %d oracle %d" % (l, cnt[2], pre)
        if not np.array_equal(key_g, key_r):
            only_g, only_r = np.setdiff1d(key_g, key_r), np.setdiff1d(key_r, key_g)
            return "level %d: NMS survivors differ: %d only on gpu %s, %d only in oracle %s" % (
                l, len(only_g), [(int(k & 0xFFF), int(k >> 12)) for k in only_g[:4]],
                len(only_r), [(int(k & 0xFFF), int(k >> 12)) for k in only_r[:4]])
        order = np.argsort(packed & 0xFFFFFF)
        if not np.array_equal(packed[order] >> 24, resp.astype(np.uint32)):
            i = int(np.nonzero((packed[order] >> 24) != resp.astype(np.uint32))[0][0])
            return "level %d: score of (%d, %d): gpu %d oracle %d" % (l, xy[i, 0], xy[i, 1], packed[order][i] >> 24, resp[i])
    return "pyramid, blur and FAST candidates agree: the difference is in the quadtree / orientation / descriptor stages"


def check_frame(ex, ref, args, img, what):
    got = ex.extractFeatures(img)
    kp_r, desc_r, per_r = ref.extract(img)
    if got is None:
        assert len(kp_r) == 0, "%s: gpu found nothing, oracle %d keypoints; %s" % (what, len(kp_r), first_difference(ex, ref, args))
        return 0
    kp, desc = got
    ok = (len(kp) == len(kp_r) and np.array_equal(ex.last_per_level, per_r) and kp.tobytes() == kp_r.tobytes()
          and np.array_equal(desc, desc_r))
    if not ok:
        detail = first_difference(ex, ref, args)
        if len(kp) == len(kp_r):
            for fld in ("x", "y", "response", "size", "octave"):
                if not np.array_equal(kp[fld], kp_r[fld]):
                    i = int(np.nonzero(kp[fld] != kp_r[fld])[0][0])
                    detail += "; keypoint %d field %s gpu %r oracle %r" % (i, fld, kp[fld][i], kp_r[fld][i])
                    break
            else:
                if kp["angle"].tobytes() != kp_r["angle"].tobytes():
                    detail += "; angle bits differ on %d keypoints" % int((kp["angle"].view(np.uint32) != kp_r["angle"].view(np.uint32)).sum())
                elif not np.array_equal(desc, desc_r):
                    detail += "; %d descriptors differ" % int((desc != desc_r).any(axis=1).sum())
        assert False, "%s: gpu %d keypoints %s, oracle %d %s; %s" % (what, len(kp), list(ex.last_per_level), len(kp_r), list(per_r), detail)
    return len(kp)


@pytest.mark.parametrize("geom", sorted(GEOM))
@pytest.mark.parametrize("kind", synth.HOSTILE_KINDS)
def test_hostile_class_bit_exact(built, kind, geom):
    import orbfe
    args = GEOM[geom]
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    ref = O.Extractor(*args)
    total = 0
    for seed in (0, 1):
        total += check_frame(ex, ref, args, synth.hostile(kind, args[6], args[7], seed), "%s/%s seed %d" % (kind, geom, seed))
    # (a period-3 checkerboard halved four times is flat: no keypoints is the right answer for checker3/s20)
    assert total > 0 or kind.startswith("checker"), "the class produced no keypoints at all: it tests nothing"


@pytest.mark.parametrize("nfast", [400000, 16000, 3000])
@pytest.mark.parametrize("kind", ["noise", "saltpepper", "pink"])
def test_dense_classes_with_and_without_caps(built, kind, nfast):
    """The densest classes with the pre-NMS cap out of reach (every corner of a 752x480 level reaches the quadtree:
    ~27 k NMS survivors in level 0), with SURVEY's default cap (active on these images), and with a tight one."""
    import orbfe
    args = (1000, nfast, 1.2, 8, 20, 7, 752, 480)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    ref = O.Extractor(*args)
    assert check_frame(ex, ref, args, synth.hostile(kind, 752, 480, 5), "%s nFast %d" % (kind, nfast)) > 900


@pytest.mark.parametrize("kind", ["noise", "plateau", "checker3"])
def test_hostile_reference_node_budgets(built, kind):
    """The reference nodes' own budgets (mono_inertial_gnss_node.cpp:96-101: 50000 features, 6 levels, FAST 40/35) on dense
    content: tens of thousands of quadtree nodes per level (the HBM-slab node tables)."""
    import orbfe
    args = (50000, 86000, 1.2, 6, 40, 35, 752, 480)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    ref = O.Extractor(*args)
    check_frame(ex, ref, args, synth.hostile(kind, 752, 480, 2), "%s gnss budgets" % kind)


def test_hostile_batch_mixed_classes(built):
    """One batch holding every class at once (the batched launch geometry: blockIdx.x = frame), against per-frame oracle runs."""
    import orbfe
    args = GEOM["c1"]
    ims = [synth.hostile(k, args[6], args[7], 3) for k in synth.HOSTILE_KINDS]
    ex = orbfe.ORBextractor(*args, device=0, max_batch=len(ims))
    ref = O.Extractor(*args)
    res = ex.extract_batch(ims)
    for k, im, (kp, desc, per) in zip(synth.HOSTILE_KINDS, ims, res):
        kp_r, desc_r, per_r = ref.extract(im)
        assert len(kp) == len(kp_r) and np.array_equal(per, per_r), (k, list(per), list(per_r))
        assert kp.tobytes() == kp_r.tobytes() and np.array_equal(desc, desc_r), k


def test_hostile_matching_exact(built):
    """SearchByProjection on the keypoints of the dense classes: thousands of look-alike descriptors per window
    (checkerboards, blob lattices) drive the top-K lists into their overflow / exact-rescan paths."""
    import bench
    import orbfe
    args = GEOM["c1"]
    W, H = args[6], args[7]
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    ref = O.Extractor(*args)
    m = orbfe.ORBmatcher(ex)
    rng = np.random.default_rng(17)
    for kind in ("noise", "checker3", "plateau", "pink", "saltpepper"):
        kp, desc, _ = ref.extract(synth.hostile(kind, W, H, 4))
        mps, mpd = bench.make_map_points(kp.view(orbfe.KP_DTYPE), len(kp), desc, 2000, rng, ref.nLevels, orbfe.MP_DTYPE)
        fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), ref.scaleFactors)
        n_ref, out_ref = O.search_by_projection(fvo, mps.view(O.MP_DTYPE), mpd, None, 20.0, 0.85)
        fv = orbfe.make_frame_view(kp.view(orbfe.KP_DTYPE), desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
        n, out = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
        assert n == n_ref and np.array_equal(out, out_ref), kind
