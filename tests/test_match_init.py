"""SearchForInitialization: oracle vs an independent Python restatement (no GPU) and the
HIP kernel vs the oracle (GPU), exact indices."""
import math
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "orb_slam3_v1.0_amd", "python"))  # when run as a script
import oracle_py as O  # noqa: E402
from orbfe import synth  # noqa: E402

f32 = np.float32


def _two_frames(W=320, H=240, nfeat=400, levels=4, idx0=7):
    e = O.Extractor(nfeat, 20000, 1.2, levels, 20, 7, W, H)
    fr = list(synth.stream(W, H, 2, index0=idx0))
    kp1, d1, _ = e.extract(fr[0])
    kp2, d2, _ = e.extract(fr[1])
    return e, kp1, d1, kp2, d2


def py_round(v):
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def py_init(kp1, d1, kp2, d2, cols, rows, invW, invH, window, nn, orient):
    n1, n2 = len(kp1), len(kp2)
    grid = [[] for _ in range(cols * rows)]
    for i in range(n2):
        px = py_round(float(f32(f32(kp2["x"][i]) * f32(invW))))
        py = py_round(float(f32(f32(kp2["y"][i]) * f32(invH))))
        lin = py * cols + px
        if 0 <= lin < cols * rows:
            grid[lin].append(i)
    m12 = [-1] * n1
    m21 = [-1] * n2
    md = [2 ** 31 - 1] * n2
    hist = [[] for _ in range(30)]
    nm = 0
    r = f32(window)
    for i1 in range(n1):
        if kp1["octave"][i1] > 0:
            continue
        x, y = f32(kp1["x"][i1]), f32(kp1["y"][i1])
        c0 = max(0, math.floor(float(f32(f32(x - r) * f32(invW)))))
        c1 = min(cols - 1, math.ceil(float(f32(f32(x + r) * f32(invW)))))
        r0 = max(0, math.floor(float(f32(f32(y - r) * f32(invH)))))
        r1 = min(rows - 1, math.ceil(float(f32(f32(y + r) * f32(invH)))))
        if c0 >= cols or c1 < 0 or r0 >= rows or r1 < 0:
            continue
        b1 = b2 = 2 ** 31 - 1
        bi = -1
        for ix in range(c0, c1 + 1):
            for iy in range(r0, r1 + 1):
                for j in grid[iy * cols + ix]:
                    if kp2["octave"][j] != 0:
                        continue
                    if not (abs(f32(kp2["x"][j] - x)) < r and abs(f32(kp2["y"][j] - y)) < r):
                        continue
                    d = int(np.unpackbits(d1[i1] ^ d2[j]).sum())
                    if md[j] <= d:
                        continue
                    if d < b1:
                        b2, b1, bi = b1, d, j
                    elif d < b2:
                        b2 = d
        if b1 <= 30 and f32(b1) < f32(f32(b2) * f32(nn)):
            if m21[bi] >= 0:
                m12[m21[bi]] = -1
                nm -= 1
            m12[i1] = bi
            m21[bi] = i1
            md[bi] = b1
            nm += 1
            if orient:
                rot = f32(kp1["angle"][i1] - kp2["angle"][bi])
                if rot < 0:
                    rot = f32(rot + f32(360.0))
                b = py_round(float(f32(rot * f32(f32(1.0) / f32(30)))))
                hist[0 if b == 30 else b].append(i1)
    if orient:
        mx = [0, 0, 0]
        ind = [-1, -1, -1]
        for i in range(30):
            s = len(hist[i])
            if s > mx[0]:
                mx = [s, mx[0], mx[1]]; ind = [i, ind[0], ind[1]]
            elif s > mx[1]:
                mx = [mx[0], s, mx[1]]; ind = [ind[0], i, ind[1]]
            elif s > mx[2]:
                mx[2] = s; ind[2] = i
        if f32(mx[1]) < f32(f32(0.1) * f32(mx[0])):
            ind[1] = ind[2] = -1
        elif f32(mx[2]) < f32(f32(0.1) * f32(mx[0])):
            ind[2] = -1
        for i in range(30):
            if i in ind:
                continue
            for i1 in hist[i]:
                if m12[i1] >= 0:
                    m12[i1] = -1
                    nm -= 1
    return nm, m12


@pytest.mark.parametrize("window,nn,orient,grid", [(40, 0.45, True, (64, 48)), (40, 0.9, True, (64, 48)),
                                                   (100, 0.9, False, (16, 12)), (15, 0.95, True, (512, 512))])
def test_init_oracle_equals_python(window, nn, orient, grid):
    e, kp1, d1, kp2, d2 = _two_frames()
    fv1 = O.make_frame_view(kp1, d1, grid[0], grid[1], 0.0, 0.0, 320.0, 240.0, e.scaleFactors)
    fv2 = O.make_frame_view(kp2, d2, grid[0], grid[1], 0.0, 0.0, 320.0, 240.0, e.scaleFactors)
    n, m = O.search_for_initialization(fv1, fv2, window, nn, orient)
    n2, m2 = py_init(kp1, d1, kp2, d2, grid[0], grid[1], fv2.gridInvW, fv2.gridInvH, window, nn, orient)
    assert n == n2 and list(m) == m2
    assert n == sum(1 for v in m if v >= 0)
    if nn >= 0.9 and window >= 40:
        assert n > 10


@pytest.mark.gpu
@pytest.mark.parametrize("window,nn,orient,grid", [(40, 0.45, True, (64, 48)), (40, 0.9, True, (64, 48)),
                                                   (100, 0.9, False, (16, 12)), (100, 0.99, True, (64, 48))])
def test_init_hip_equals_oracle(built, window, nn, orient, grid):
    import orbfe
    W, H = 752, 480
    args = (1000, 40000, 1.2, 8, 20, 7, W, H)
    e = O.Extractor(*args)
    fr = list(synth.stream(W, H, 2, index0=9))
    kp1, d1, _ = e.extract(fr[0])
    kp2, d2, _ = e.extract(fr[1])
    f1 = O.make_frame_view(kp1, d1, grid[0], grid[1], 0.0, 0.0, float(W), float(H), e.scaleFactors)
    f2 = O.make_frame_view(kp2, d2, grid[0], grid[1], 0.0, 0.0, float(W), float(H), e.scaleFactors)
    n_ref, m_ref = O.search_for_initialization(f1, f2, window, nn, orient)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    g1 = orbfe.make_frame_view(kp1, d1, grid[0], grid[1], 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    g2 = orbfe.make_frame_view(kp2, d2, grid[0], grid[1], 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    n, m = orbfe.ORBmatcher(ex).SearchForInitialization(g1, g2, window, nn, orient)
    assert n == n_ref and np.array_equal(m, m_ref)
    # identical frames: every level-0 keypoint matches itself at distance 0 unless the ratio test rejects it
    n_s, m_s = orbfe.ORBmatcher(ex).SearchForInitialization(g1, g1, window, nn, orient)
    n_so, m_so = O.search_for_initialization(f1, f1, window, nn, orient)
    assert n_s == n_so and np.array_equal(m_s, m_so)


INIT_CASES = [(1, 1000, 100, 0.9), (1, 2000, 60, 0.9), (2, 1500, 400, 0.95), (8, 1000, 1000, 0.9),
              (1, 3000, 100, 0.9), (1, 1100, 400, 0.95)]  # the last two: > 2048 keypoints in frame 2, > 1024 level-0 keypoints with a huge window


def _init_case(levels, nfeat, window, nn):
    """-> (frame views of the oracle side, GPU-side constructor arguments, reference result)"""
    W, H = 752, 480
    args = (nfeat, 40000, 1.2, levels, 20, 7, W, H)
    e = O.Extractor(*args)
    fr = list(synth.stream(W, H, 2, index0=4))
    kp1, d1, _ = e.extract(fr[0])
    kp2, d2, _ = e.extract(fr[1])
    f1 = O.make_frame_view(kp1, d1, 64, 48, 0.0, 0.0, float(W), float(H), e.scaleFactors)
    f2 = O.make_frame_view(kp2, d2, 64, 48, 0.0, 0.0, float(W), float(H), e.scaleFactors)
    n_ref, m_ref = O.search_for_initialization(f1, f2, window, nn, True)
    assert n_ref > 20
    return args, (kp1, d1, kp2, d2), (n_ref, m_ref)


def _init_gpu(orbfe, args, kd, window, nn):
    W, H = args[6], args[7]
    kp1, d1, kp2, d2 = kd
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    g1 = orbfe.make_frame_view(kp1, d1, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    g2 = orbfe.make_frame_view(kp2, d2, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    return orbfe.ORBmatcher(ex).SearchForInitialization(g1, g2, window, nn, True)


@pytest.mark.gpu
@pytest.mark.parametrize("levels,nfeat,window,nn", INIT_CASES)
def test_init_fast_and_sequential_kernels_agree(built, levels, nfeat, window, nn):
    """Both device paths of orbfe_match_initialization against the oracle, selected by the host from the input sizes as in
    production: the two-phase kernel (candidate keys in parallel, one wave for the order-dependent part) and the
    sequential block kernel that takes over for large inputs.  One pyramid level puts every keypoint on level 0 (long
    candidate rows, rows that overflow the LDS image with a huge window; > 1024 level-0 keypoints or > 2048 keypoints in
    frame 2 -> the sequential kernel)."""
    import orbfe
    args, kd, (n_ref, m_ref) = _init_case(levels, nfeat, window, nn)
    n, m = _init_gpu(orbfe, args, kd, window, nn)
    assert n == n_ref and np.array_equal(m, m_ref)


@pytest.mark.gpu
def test_init_sequential_kernel_forced_on_small_inputs(built):
    """The sequential kernel on the inputs the fast kernel normally takes.  The switch that forces it (ORBFE_INIT_SLOW)
    exists only in the diagnostics build liborbfe_diag.so (`make diag`, -DORBFE_DIAG) -- the shipped library reads no such
    variable -- so this runs once in a child process that loads that build."""
    import subprocess
    import __graft_entry__ as g
    g.build_variant("diag")  # built on demand: build() only makes the shipped library
    env = dict(os.environ, ORBFE_INIT_SLOW="1", ORBFE_TEST_LIB="liborbfe_diag.so")
    p = subprocess.run([sys.executable, os.path.abspath(__file__)], env=env, capture_output=True, text=True, timeout=600)
    assert p.returncode == 0 and "forced sequential kernel: 4 cases exact" in p.stdout, p.stdout[-1000:] + p.stderr[-2000:]


if __name__ == "__main__":  # child of test_init_sequential_kernel_forced_on_small_inputs
    import orbfe
    orbfe.LIB_PATH = os.path.join(orbfe.CSRC, os.environ["ORBFE_TEST_LIB"])
    assert os.environ.get("ORBFE_INIT_SLOW") == "1"
    for case in INIT_CASES[:4]:
        a, kd, (n_ref, m_ref) = _init_case(*case)
        n, m = _init_gpu(orbfe, a, kd, case[2], case[3])
        assert n == n_ref and np.array_equal(m, m_ref), case
    print("forced sequential kernel: 4 cases exact")
