"""Seeded synthetic frames: EuRoC / TUM-VI images are not available offline.

frame(seed): mid-grey 128 background, K ~ 0.15% * W * H axis-aligned rectangles and K/4 filled
discs with uniform grey levels, one 3x3 box blur, additive uniform noise in [-3, 3], clamp to u8.
stream(): frame t+1 = frame t translated by a seeded (dx, dy) in [-4, 4]^2 plus new noise.
"""
import numpy as np

SEED0 = 0x0B5EED


def _base(w, h, rng, density=0.0015):
    img = np.full((h, w), 128, np.int32)
    k = max(4, int(round(density * w * h)))
    xs = rng.integers(0, w, k)
    ys = rng.integers(0, h, k)
    ws = rng.integers(4, max(5, w // 8), k)
    hs = rng.integers(4, max(5, h // 8), k)
    gs = rng.integers(0, 256, k)
    for x, y, ww, hh, g in zip(xs, ys, ws, hs, gs):
        img[y:y + hh, x:x + ww] = g
    kd = k // 4
    cx = rng.integers(0, w, kd)
    cy = rng.integers(0, h, kd)
    rr = rng.integers(3, max(4, min(w, h) // 16), kd)
    gd = rng.integers(0, 256, kd)
    for x, y, r, g in zip(cx, cy, rr, gd):
        x0, x1 = max(0, x - r), min(w, x + r + 1)
        y0, y1 = max(0, y - r), min(h, y + r + 1)
        yy, xx = np.ogrid[y0:y1, x0:x1]
        m = (xx - x) ** 2 + (yy - y) ** 2 <= r * r
        img[y0:y1, x0:x1][m] = g
    # 3x3 box blur (edge replicated)
    p = np.pad(img, 1, mode="edge")
    s = sum(p[dy:dy + h, dx:dx + w] for dy in range(3) for dx in range(3))
    return (s + 4) // 9


def _noise(img, rng):
    n = rng.integers(-3, 4, img.shape)
    return np.clip(img + n, 0, 255).astype(np.uint8)


def frame(w, h, index=0, density=0.0015):
    rng = np.random.default_rng(SEED0 + index)
    return _noise(_base(w, h, rng, density), rng)


def stream(w, h, count, index0=0, density=0.0015):
    """Yield `count` frames of a translating scene (one seeded base, shifted inside a padded canvas)."""
    rng = np.random.default_rng(SEED0 + 7919 * (index0 + 1))
    pad = 64
    big = _base(w + 2 * pad, h + 2 * pad, rng, density)
    ox, oy = pad, pad
    for _ in range(count):
        yield _noise(big[oy:oy + h, ox:ox + w], rng)
        ox = min(max(ox + int(rng.integers(-4, 5)), 0), 2 * pad)
        oy = min(max(oy + int(rng.integers(-4, 5)), 0), 2 * pad)


def batch(w, h, count, index0=0, density=0.0015):
    return np.stack([frame(w, h, index0 + i, density) for i in range(count)])


def _pink(w, h, rng, sigma=40.0):
    """1/f ("pink") noise canvas: random phases, amplitude 1 / spatial frequency -- texture at every scale."""
    fy = np.fft.fftfreq(h)[:, None]
    fx = np.fft.rfftfreq(w)[None, :]
    f = np.sqrt(fx * fx + fy * fy)
    f[0, 0] = 1.0
    spec = (rng.standard_normal((h, w // 2 + 1)) + 1j * rng.standard_normal((h, w // 2 + 1))) / f
    spec[0, 0] = 0.0
    img = np.fft.irfft2(spec, s=(h, w))
    img = img / img.std() * sigma + 128.0
    return np.clip(np.rint(img), 0, 255).astype(np.int32)


def pink_stream(w, h, count, index0=0):
    """Like stream(), on a 1/f-noise scene (texture-rich: many weak corners at every pyramid level)."""
    rng = np.random.default_rng(SEED0 + 104729 * (index0 + 1))
    pad = 64
    big = _pink(w + 2 * pad, h + 2 * pad, rng)
    ox, oy = pad, pad
    for _ in range(count):
        yield _noise(big[oy:oy + h, ox:ox + w], rng)
        ox = min(max(ox + int(rng.integers(-4, 5)), 0), 2 * pad)
        oy = min(max(oy + int(rng.integers(-4, 5)), 0), 2 * pad)


# --------------------------------------------------------------------------------------------------------------------
# Hostile image classes ("parity on hostile inputs"): content the rectangle / disc generator never makes and
# on which the reference's FAST / NMS edge cases are decided (src/cuda/Fast_gpu.cu:193-216 score saturation, :222-267
# threshold boundaries, :289-319 strict-> ties), plus maximum corner density for the queues, caps and node tables.
# Every class is a pure function of (kind, w, h, seed).
# --------------------------------------------------------------------------------------------------------------------
HOSTILE_KINDS = ("noise", "checker1", "checker2", "checker3", "checker4", "plateau", "extreme", "seams", "thresh",
                 "pink", "lowtex", "saltpepper")


def _stamp(img, x, y, bw, bh, g):
    h, w = img.shape
    img[max(0, y):min(h, y + bh), max(0, x):min(w, x + bw)] = g


def hostile(kind, w, h, seed=0, tile=(64, 32)):
    """One u8 frame of the named hostile class."""
    rng = np.random.default_rng(SEED0 + 15485863 * (seed + 1) + sum(map(ord, kind)))
    if kind == "noise":  # uniform white noise: maximum corner density at every level
        return rng.integers(0, 256, (h, w), dtype=np.uint8)
    if kind.startswith("checker"):  # period 1..4; full contrast on the left half, contrast 24 (> iniTh) on the right
        p = int(kind[7:])
        yy, xx = np.mgrid[0:h, 0:w]
        on = ((xx // p + yy // p) & 1).astype(bool)
        img = np.where(on, 255, 0)
        soft = np.where(on, 140, 116)
        img[:, w // 2:] = soft[:, w // 2:]
        return img.astype(np.uint8)
    if kind == "plateau":
        # lattice of IDENTICAL blobs (1x1, 2x1, 1x2, 2x2, 3x3, 3x2 in rotation) on a flat background: all pixels of a
        # blob score the same, so strict > lets neighbours suppress each other (Fast_gpu.cu:300-310); the blob grey
        # flips polarity per lattice row; pitch 7 x 6 px so blobs drift across every tile and level phase
        bg, shapes = 90, ((1, 1), (2, 1), (1, 2), (2, 2), (3, 3), (3, 2))
        img = np.full((h, w), bg, np.int32)
        k = 0
        for j, y in enumerate(range(2, h - 3, 6)):
            g = 200 if j & 1 else 20
            for x in range(2, w - 3, 7):
                bw, bh = shapes[k % len(shapes)]
                _stamp(img, x, y, bw, bh, g)
                k += 1
        return img.astype(np.uint8)
    if kind == "extreme":
        # all-0 top half with isolated 255 pixels, all-255 bottom half with isolated 0 pixels (score 254, the largest
        # the binary search of :193-216 can return), and 2-px pairs of them (equal maximal scores side by side)
        img = np.zeros((h, w), np.int32)
        img[h // 2:] = 255
        n = max(8, w * h // 400)
        xs, ys = rng.integers(0, w, n), rng.integers(0, h, n)
        for i, (x, y) in enumerate(zip(xs, ys)):
            v = 255 if y < h // 2 else 0
            img[y, x] = v
            if i % 3 == 0 and x + 1 < w:
                img[y, x + 1] = v
        return img.astype(np.uint8)
    if kind == "seams":
        # rectangle corners and step edges ON the FAST tile seams (multiples of tile w / h and the pixels either side)
        # and on the tested-region border (x, y in {5, 6, w-6, w-5}); grey levels alternate so that both polarities occur
        tw, th_ = tile
        img = np.full((h, w), 128, np.int32)
        k = 0
        for y0 in list(range(0, h, th_)) + [5, 6, h - 6, h - 5]:
            for x0 in list(range(0, w, tw)) + [5, 6, w - 6, w - 5]:
                dx, dy = (k % 3) - 1, ((k // 3) % 3) - 1  # corner at the seam or one pixel either side
                g = (30, 220, 70, 180)[k % 4]
                _stamp(img, x0 + dx, y0 + dy, 9 + k % 5, 7 + k % 4, g)
                k += 1
        # isolated single pixels exactly at the first / last tested coordinates of level 0
        for (x, y) in ((6, 6), (w - 6, 6), (6, h - 6), (w - 6, h - 6), (5, 5), (w - 5, h - 5), (6, h // 2), (w // 2, h - 6)):
            img[y, x] = 255
        return img.astype(np.uint8)
    if kind == "thresh":
        # blobs whose contrast is exactly th and th + 1 for the usual thresholds (7, 20) and near saturation: the
        # compare is strict (`> th`), and v + th passes 255 for bright centres
        img = np.full((h, w), 100, np.int32)
        img[:, 2 * w // 3:] = 245  # bright background: v + th > 255
        img[: h // 4, : w // 3] = 8  # dark background: v - th < 0
        deltas = (7, 8, -7, -8, 20, 21, -20, -21, 6, -6, 9, -9, 19, -19, 10, -10)
        k = 0
        for y in range(3, h - 4, 9):
            for x in range(3, w - 4, 9):
                d = deltas[k % len(deltas)]
                s = 1 + (k // len(deltas)) % 3
                base = int(img[y, x])
                _stamp(img, x, y, s, s, min(255, max(0, base + d)))
                k += 1
        return img.astype(np.uint8)
    if kind == "pink":
        return _noise(_pink(w, h, rng), rng)
    if kind == "lowtex":  # bench.py --texture-sweep's low-texture density
        return _noise(_base(w, h, rng, 0.0002), rng)
    if kind == "saltpepper":  # flat grey with 2 % salt and 2 % pepper: isolated maximal corners of both polarities
        img = np.full((h, w), 128, np.uint8)
        r = rng.random((h, w))
        img[r < 0.02] = 0
        img[r > 0.98] = 255
        return img
    raise ValueError(kind)


def lowtex_stream(w, h, count, index0=0):
    return stream(w, h, count, index0=index0, density=0.0002)


# ---- inputs of the node-side image preparation (ImageGrabber::ConvertImageToGPU, image_grabber.hpp:96-110) ----
def colour_image(w, h, seed):
    """Smooth colour gradients + blocks + noise (three different channels so a channel swap shows)."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    img = np.stack([(xx * 255 // max(w - 1, 1)), (yy * 255 // max(h - 1, 1)), ((xx + yy) * 3) % 256], 2).astype(np.int32)
    for _ in range(max(4, w * h // 4000)):
        x0, y0 = int(rng.integers(0, w)), int(rng.integers(0, h))
        img[y0:y0 + int(rng.integers(3, 40)), x0:x0 + int(rng.integers(3, 40))] = rng.integers(0, 256, 3)
    img += rng.integers(-6, 7, img.shape)
    return np.clip(img, 0, 255).astype(np.uint8)


def fisheye_maps(w, h, seed, strength=0.18):
    """Equidistant-fisheye style maps (what cv::fisheye::initUndistortRectifyMap produces for the node, CV_32F) plus the
    special entries the operators must survive: exact integer coordinates (5 taps, two of weight 0), points outside the
    image on every side, NaN and infinities, huge values."""
    rng = np.random.default_rng(seed)
    yy, xx = np.mgrid[0:h, 0:w].astype(np.float64)
    cx, cy, f = w * 0.498, h * 0.512, 0.65 * w
    a, b = (xx - cx) / f, (yy - cy) / f
    r = np.sqrt(a * a + b * b) + 1e-12
    th = np.arctan(r)
    thd = th * (1 + strength * th ** 2 - 0.3 * strength * th ** 4)
    m1 = (f * thd / r * a + cx).astype(np.float32)
    m2 = (f * thd / r * b + cy).astype(np.float32)
    k = max(32, w * h // 300)
    ys, xs = rng.integers(0, h, k), rng.integers(0, w, k)
    m1[ys[: k // 4], xs[: k // 4]] = np.round(m1[ys[: k // 4], xs[: k // 4]])           # integer x
    m2[ys[k // 8: k // 3], xs[k // 8: k // 3]] = np.round(m2[ys[k // 8: k // 3], xs[k // 8: k // 3]])  # integer y (some both)
    edge = [(-2.5, 3.0), (-1.0, -1.0), (w - 0.5, h - 0.5), (w + 1.0, 5.0), (3.0, h + 1.5), (-3.5, 2.0), (w + 2.5, 2.0),
            (0.0, 0.0), (w - 1.0, h - 1.0), (np.nan, 4.0), (4.0, np.inf), (-np.inf, 1.0), (1e30, 2.0), (2.0, -1e30)]
    for i, (ex, ey) in enumerate(edge):
        m1[ys[-1 - i], xs[-1 - i]], m2[ys[-1 - i], xs[-1 - i]] = ex, ey
    return m1, m2
