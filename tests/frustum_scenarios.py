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


def world_points_on_keypoints(kp, desc, v, M, rng, n_levels, wp_dtype=None):
    """map points that re-project onto keypoints of the frame (in the coordinates the matcher compares, S6): pose and
    intrinsics of `v`, depths 1.5..9 m, distance range chosen so that PredictScale lands near the keypoint's octave"""
    import match_scenarios as S
    if wp_dtype is None:
        import oracle_py
        wp_dtype = oracle_py.WP_DTYPE
    lo, hi = wp_dtype.names[3], wp_dtype.names[4]  # minDistance / maxDistance (oracle) or min_distance / max_distance (orbfe)
    src = rng.integers(0, len(kp), M)
    u = kp["x"][src] + rng.uniform(-2, 2, M)
    w = kp["y"][src] + rng.uniform(-2, 2, M)
    z = rng.uniform(1.5, 9.0, M)
    pc = np.stack([(u - v["cx"]) / v["fx"] * z, (w - v["cy"]) / v["fy"] * z, z], 1)
    R = np.asarray(v["rcw"], np.float64).reshape(3, 3)
    pw = (R.T @ (pc - np.asarray(v["tcw"], np.float64)).T).T
    pts = np.zeros(M, wp_dtype)
    pts["x"], pts["y"], pts["z"] = pw[:, 0], pw[:, 1], pw[:, 2]
    d = np.linalg.norm(pw - np.asarray(v["twc"], np.float64), axis=1)
    pts[hi] = d * np.float32(1.2) ** kp["octave"][src].astype(np.float32) * rng.uniform(0.95, 1.05, M)
    pts[lo] = pts[hi] / np.float32(1.2) ** (n_levels - 1)
    pts["observations"] = rng.integers(0, 4, M)
    pts["bad"] = rng.random(M) < 0.02
    pts["skip"] = rng.random(M) < 0.02
    mpd = np.stack([S.flip_bits(desc[s], int(rng.integers(0, 20)), rng) for s in src])
    return pts, mpd
