"""Node-side image preparation (ros2_ws/src/mono-inertial/include/image_grabber.hpp:96-110): fisheye remap INTER_CUBIC
-> resize INTER_LINEAR -> BGR2GRAY, SPEC DECISION S9 (parity against OpenCV-CUDA unpinned, see oracle/prep_oracle.c).
CPU: the C oracle against a numpy binary32 restatement; GPU: the fused HIP kernel against the oracle, bit for bit."""
import math

import numpy as np
import pytest

import oracle_py as O

f32 = np.float32


from orbfe.synth import colour_image, fisheye_maps  # noqa: E402,F401  (also used by bench.py and tests/tools/prep_latency.py)


def py_cubic(t):
    t = f32(abs(f32(t)))
    if t <= 1:
        return f32(f32(f32(t * t) * f32(f32(f32(1.5) * t) - f32(2.5))) + f32(1))
    if t < 2:
        return f32(f32(t * f32(f32(t * f32(f32(f32(-0.5) * t) + f32(2.5))) - f32(4))) + f32(2))
    return f32(0)


def py_sat(v):
    if not v > 0:
        return 0
    if v >= 255:
        return 255
    return int(np.rint(f32(v)))  # half to even


def py_remap_pixel(img, x, y):
    h, w, _ = img.shape
    x, y = f32(x), f32(y)
    if not (x > -3 and x < w + 2 and y > -3 and y < h + 2):
        return [0, 0, 0]
    s = [f32(0)] * 3
    ws = f32(0)
    for cy in range(int(math.ceil(float(f32(y - f32(2))))), int(math.floor(float(f32(y + f32(2))))) + 1):
        for cx in range(int(math.ceil(float(f32(x - f32(2))))), int(math.floor(float(f32(x + f32(2))))) + 1):
            wgt = f32(py_cubic(f32(x - f32(cx))) * py_cubic(f32(y - f32(cy))))
            if 0 <= cx < w and 0 <= cy < h:
                for c in range(3):
                    s[c] = f32(s[c] + f32(wgt * f32(img[cy, cx, c])))
            ws = f32(ws + wgt)
    if ws == 0:
        return [0, 0, 0]
    return [py_sat(f32(v / ws)) for v in s]


def py_prepare_pixel(img, m1, m2, dst_w, dst_h, dx, dy):
    h, w, _ = img.shape
    fx, fy = f32(1.0 / (dst_w / w)), f32(1.0 / (dst_h / h))
    sx, sy = f32(f32(dx) * fx), f32(f32(dy) * fy)
    x1, y1 = int(math.floor(float(sx))), int(math.floor(float(sy)))
    x2, y2 = x1 + 1, y1 + 1
    xr, yr = [min(x1, w - 1), min(x2, w - 1)], [min(y1, h - 1), min(y2, h - 1)]
    px = [py_remap_pixel(img, m1[yr[k >> 1], xr[k & 1]], m2[yr[k >> 1], xr[k & 1]]) for k in range(4)]
    wts = [f32(f32(f32(x2) - sx) * f32(f32(y2) - sy)), f32(f32(sx - f32(x1)) * f32(f32(y2) - sy)),
           f32(f32(f32(x2) - sx) * f32(sy - f32(y1))), f32(f32(sx - f32(x1)) * f32(sy - f32(y1)))]
    ch = []
    for c in range(3):
        o = f32(0)
        for k in range(4):
            o = f32(o + f32(f32(px[k][c]) * wts[k]))
        ch.append(py_sat(o))
    return (ch[0] * 1868 + ch[1] * 9617 + ch[2] * 4899 + 8192) >> 14


@pytest.mark.parametrize("w,h,dw,dh,seed", [(160, 120, 48, 36, 1), (97, 61, 97, 61, 2), (64, 48, 80, 60, 3)])
def test_oracle_prepare_matches_restatement(built, w, h, dw, dh, seed):
    img = colour_image(w, h, seed)
    m1, m2 = fisheye_maps(w, h, seed)
    grey, und = O.prepare_image(img, m1, m2, dw, dh, want_undistorted=True)
    assert float(O.prep_scale(w, dw)) == float(f32(1.0 / (dw / w)))
    rng = np.random.default_rng(seed)
    # the intermediate (undistorted, 8-bit) image on the special map entries and a random sample
    special = np.argwhere(~np.isfinite(m1) | ~np.isfinite(m2) | (m1 == np.round(m1)) | (m1 < 0) | (m2 < 0) | (m1 > w - 1) | (m2 > h - 1))
    sample = np.concatenate([special, np.stack([rng.integers(0, h, 150), rng.integers(0, w, 150)], 1)])
    for y, x in sample:
        assert list(und[y, x]) == py_remap_pixel(img, m1[y, x], m2[y, x]), (y, x, m1[y, x], m2[y, x])
    pts = [(0, 0), (dw - 1, 0), (0, dh - 1), (dw - 1, dh - 1)] + [(int(rng.integers(0, dw)), int(rng.integers(0, dh))) for _ in range(120)]
    for dx, dy in pts:
        assert int(grey[dy, dx]) == py_prepare_pixel(img, m1, m2, dw, dh, dx, dy), (dx, dy)
    assert grey.std() > 20  # a real picture, not a constant


def test_identity_map_is_the_plain_resize_and_grey(built):
    """Integer map entries: the cubic kernel has weight 1 at the centre tap and 0 elsewhere, so the undistorted image
    is the input; with dst == src the bilinear weights are (1, 0, 0, 0): the output is the BGR2GRAY formula alone."""
    w, h = 40, 30
    img = colour_image(w, h, 5)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float32)
    grey, und = O.prepare_image(img, xx, yy, w, h, want_undistorted=True)
    assert np.array_equal(und, img)
    want = (img[..., 0].astype(np.uint32) * 1868 + img[..., 1].astype(np.uint32) * 9617 + img[..., 2].astype(np.uint32) * 4899 + 8192) >> 14
    assert np.array_equal(grey, want.astype(np.uint8))


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,dw,dh,seed", [(160, 120, 48, 36, 1), (97, 61, 97, 61, 2), (64, 48, 80, 60, 3), (640, 480, 192, 144, 4),
                                            (2048, 1536, 614, 460, 5)])
def test_gpu_prepare_matches_oracle(built, w, h, dw, dh, seed):
    import orbfe
    img = colour_image(w, h, seed)
    m1, m2 = fisheye_maps(w, h, seed)
    ref = O.prepare_image(img, m1, m2, dw, dh)
    ex = orbfe.ORBextractor(300, 4000, 1.2, 3, 20, 7, max(dw, 64), max(dh, 64))
    prep = orbfe.ImagePreparer(ex, m1, m2, dw, dh)
    got = prep.prepare(img)
    assert np.array_equal(got, ref)
    # a padded (pitched) source view gives the same result
    padded = np.zeros((h, w + 5, 3), np.uint8)
    padded[:, :w] = img
    assert np.array_equal(prep.prepare(padded[:, :w]), ref)
    prep.close()


@pytest.mark.gpu
def test_gpu_prepare_and_extract_chain(built):
    """ConvertImageToGPU -> extractFeatures in one call == the two steps with the grey frame through the host, and ==
    the oracle chain (prepare oracle -> extraction oracle)."""
    import orbfe
    w, h, dw, dh = 1024, 768, 614, 460
    args = (1000, 20000, 1.2, 8, 20, 7, dw, dh)
    from orbfe import synth
    base = synth.frame(w, h, 3)
    img = np.stack([base, np.roll(base, 1, 1), 255 - base // 2], 2).astype(np.uint8)
    m1, m2 = fisheye_maps(w, h, 9, strength=0.1)
    ex = orbfe.ORBextractor(*args)
    prep = orbfe.ImagePreparer(ex, m1, m2, dw, dh)
    kp, desc, grey = prep.extract(img, want_gray=True)
    grey_ref = O.prepare_image(img, m1, m2, dw, dh)
    assert np.array_equal(grey, grey_ref)
    kp_r, desc_r, _ = O.Extractor(*args).extract(grey_ref)
    assert len(kp) == len(kp_r) > 300 and kp.tobytes() == kp_r.tobytes() and np.array_equal(desc, desc_r)
    kp2, desc2 = ex.extractFeatures(prep.prepare(img))
    assert kp2.tobytes() == kp.tobytes() and np.array_equal(desc2, desc)
    # an extractor of another size is refused
    other = orbfe.ORBextractor(500, 4000, 1.2, 4, 20, 7, 320, 240)
    with pytest.raises(Exception):
        orbfe.ImagePreparer(other, m1, m2, dw, dh).extract(img)
