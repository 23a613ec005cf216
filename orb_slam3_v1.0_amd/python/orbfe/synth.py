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
