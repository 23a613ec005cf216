"""Synthetic camera + map-point clouds for the f3 (isInFrustum) parity tests."""
import ctypes as C
import math

import numpy as np


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = math.cos(rx), math.sin(rx), math.cos(ry), math.sin(ry), math.cos(rz), math.sin(rz)
    Rx = np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]])
    Ry = np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]])
    Rz = np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]])
    return (Rz @ Ry @ Rx).astype(np.float32)


def fill_frustum(F, names, W=752.0, H=480.0, n_levels=8, scale=1.2, seed=0, kb8=False):
    """`names` maps the logical field names to the struct's (oracle: camelCase, product: snake_case)."""
    rng = np.random.default_rng(seed)
    R = rot(*(rng.uniform(-0.2, 0.2, 3)))
    t = rng.uniform(-0.5, 0.5, 3).astype(np.float32)
    twc = (-(R.T.astype(np.float64) @ t.astype(np.float64))).astype(np.float32)
    vals = dict(rcw=R.reshape(-1), tcw=t, twc=twc, min_x=0.0, max_x=W, min_y=0.0, max_y=H, fx=458.654, fy=457.296,
                cx=367.215, cy=248.375, mbf=47.9, log_scale_factor=float(np.log(np.float32(scale))), n_levels=n_levels,
                camera_model=1 if kb8 else 0, k1=-0.0135 if kb8 else 0.0, k2=0.021 if kb8 else 0.0,
                k3=-0.0107 if kb8 else 0.0, k4=0.0023 if kb8 else 0.0)
    for k, v in vals.items():
        f = names[k]
        if k in ("rcw", "tcw", "twc"):
            arr = getattr(F, f)
            for i, x in enumerate(v):
                arr[i] = float(x)
        else:
            setattr(F, f, v)
    return vals


def world_points(n, dtype, names, seed=1):
    """Points in front of, behind and beside the camera, with distance ranges that exercise both distance
    rejections and every predicted level."""
    rng = np.random.default_rng(seed)
    p = np.zeros(n, dtype)
    p[names["x"]] = rng.uniform(-6, 6, n)
    p[names["y"]] = rng.uniform(-4, 4, n)
    p[names["z"]] = rng.uniform(-2, 12, n)
    d = np.sqrt(p[names["x"]] ** 2 + p[names["y"]] ** 2 + p[names["z"]] ** 2) + 0.01
    lo = d * rng.uniform(0.2, 1.3, n)
    p[names["min_distance"]] = lo
    p[names["max_distance"]] = lo * rng.uniform(1.0, 6.0, n)
    p[names["bad"]] = rng.random(n) < 0.03
    p[names["observations"]] = rng.integers(0, 5, n)
    p[names["skip"]] = rng.random(n) < 0.05
    # exact boundaries: ratio == scale^k (ceil edge), depth 0, on the image border
    k = min(n, 8)
    p[names["max_distance"]][:k] = (d[:k] * np.float32(1.2) ** np.arange(k)).astype(np.float32)
    p[names["min_distance"]][:k] = 0.0
    return p
