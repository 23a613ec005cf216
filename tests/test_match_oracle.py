"""Pins the matcher oracle (C) against a second, brute-force restatement in Python (no GPU)."""
import math

import numpy as np
import pytest

import match_scenarios as S
import oracle_py as O
from orbfe import synth

NAMES_O = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")
f32 = np.float32


def _frame(W=320, H=240, idx=2, nfeat=300, levels=4):
    e = O.Extractor(nfeat, 20000, 1.2, levels, 20, 7, W, H)
    kp, desc, _ = e.extract(synth.frame(W, H, idx))
    return e, kp, desc


def py_round(v):  # C round(): half away from zero
    return int(math.floor(abs(v) + 0.5)) * (1 if v >= 0 else -1)


def py_search_by_projection(kp, desc, cols, rows, minX, minY, invW, invH, sf, mps, mpd, init_obs, th, nnRatio,
                            far=False, thFar=0.0):
    n = len(kp)
    grid = [[] for _ in range(cols * rows)]
    for i in range(n):
        px = py_round(float(f32(f32(kp["x"][i] - f32(minX)) * f32(invW))))
        py = py_round(float(f32(f32(kp["y"][i] - f32(minY)) * f32(invH))))
        lin = py * cols + px
        if 0 <= lin < cols * rows:
            grid[lin].append(i)
    slot = list(init_obs)
    out = [-1] * n
    nm = 0
    for i, mp in enumerate(mps):
        if not mp["inView"] or (far and mp["trackDepth"] > thFar) or mp["bad"]:
            continue
        lvl = int(mp["level"])
        r = f32(2.5) if float(mp["viewCos"]) > 0.998 else f32(4.0)
        if th != 1.0:
            r = f32(r * f32(th))
        r = f32(r * sf[lvl])
        x, y = f32(mp["projX"]), f32(mp["projY"])
        c0 = max(0, math.floor(float(f32(f32(f32(x - f32(minX)) - r) * f32(invW)))))
        c1 = min(cols - 1, math.ceil(float(f32(f32(f32(x - f32(minX)) + r) * f32(invW)))))
        r0 = max(0, math.floor(float(f32(f32(f32(y - f32(minY)) - r) * f32(invH)))))
        r1 = min(rows - 1, math.ceil(float(f32(f32(f32(y - f32(minY)) + r) * f32(invH)))))
        if c0 >= cols or c1 < 0 or r0 >= rows or r1 < 0:
            continue
        cand = []
        for ix in range(c0, c1 + 1):
            for iy in range(r0, r1 + 1):
                for j in grid[iy * cols + ix]:
                    o = int(kp["octave"][j])
                    if o < lvl - 1 or o > lvl:
                        continue
                    if abs(f32(kp["x"][j] - x)) < r and abs(f32(kp["y"][j] - y)) < r:
                        cand.append(j)
        if not cand:
            continue
        bd, bl, bd2, bl2, bi = 256, -1, 256, -1, -1
        for j in cand:
            if slot[j] > 0:
                continue
            d = int(np.unpackbits(mpd[i] ^ desc[j]).sum())
            if d < bd:
                bd2, bd, bl2, bl, bi = bd, d, bl, int(kp["octave"][j]), j
            elif d < bd2:
                bl2, bd2 = int(kp["octave"][j]), d
        if bd <= 100:
            if bl == bl2 and f32(bd) > f32(f32(nnRatio) * f32(bd2)):
                continue
            out[bi] = i
            slot[bi] = int(mp["observations"])
            nm += 1
    return nm, out


@pytest.mark.parametrize("grid,th,nn,seed", [((64, 48), 20.0, 0.85, 1), ((64, 48), 40.0, 0.75, 2),
                                             ((16, 12), 1.0, 0.85, 3), ((512, 512), 20.0, 0.85, 4)])
def test_projection_oracle_equals_python(grid, th, nn, seed):
    e, kp, desc = _frame()
    W, H = 320, 240
    mps, mpd, init_obs = S.projection_scenario(kp, desc, 400, seed, O.MP_DTYPE, NAMES_O, e.nLevels)
    fv = O.make_frame_view(kp, desc, grid[0], grid[1], 0.0, 0.0, float(W), float(H), e.scaleFactors)
    n, out = O.search_by_projection(fv, mps, mpd, init_obs, th, nn)
    n2, out2 = py_search_by_projection(kp, desc, grid[0], grid[1], 0.0, 0.0, fv.gridInvW, fv.gridInvH, e.scaleFactors,
                                       mps, mpd, init_obs, th, nn)
    assert n == n2 and list(out) == out2
    assert n > 50


def test_projection_far_points_and_empty():
    e, kp, desc = _frame()
    mps, mpd, init_obs = S.projection_scenario(kp, desc, 200, 9, O.MP_DTYPE, NAMES_O, e.nLevels)
    fv = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 320.0, 240.0, e.scaleFactors)
    n_all, _ = O.search_by_projection(fv, mps, mpd, init_obs, 20.0, 0.85)
    n_far, out = O.search_by_projection(fv, mps, mpd, init_obs, 20.0, 0.85, True, 10.0)
    n_py, out_py = py_search_by_projection(kp, desc, 64, 48, 0.0, 0.0, fv.gridInvW, fv.gridInvH, e.scaleFactors, mps, mpd,
                                           init_obs, 20.0, 0.85, True, 10.0)
    assert n_far == n_py and list(out) == out_py and n_far < n_all
    n0, out0 = O.search_by_projection(fv, mps[:0], mpd[:0], init_obs, 20.0, 0.85)
    assert n0 == 0 and (out0 == -1).all()


def test_grid_wrap_quirk():
    """PosInGrid validates only the linear index (src/Frame.cc:470-480): posX == cols wraps to the next row."""
    kp = np.zeros(3, O.KP_DTYPE)
    kp["x"] = [319.9, 10.0, 319.9]
    kp["y"] = [10.0, 10.0, 239.9]
    fv = O.make_frame_view(kp, np.zeros((3, 32), np.uint8), 64, 48, 0.0, 0.0, 320.0, 240.0, np.ones(1, np.float32))
    cells = O.assign_grid(fv)
    assert cells[0] == 2 * 64 + 64  # column 64 of row 2 == column 0 of row 3
    assert cells[1] == 2 * 64 + 2
    assert cells[2] == -1  # row 48 -> outside the linear range


def py_bow(kfOff, kfIdx, fOff, fIdx, kfDesc, kfAng, has, fDesc, fAng, nn, orient, nLeft=-1):
    nF = len(fDesc)
    out = [-1] * nF
    hist = [[] for _ in range(30)]
    nm = 0
    for g in range(len(kfOff) - 1):
        for ik in kfIdx[kfOff[g]:kfOff[g + 1]]:
            if not has[ik]:
                continue
            b1, b2, bi = 256, 256, -1
            r1, ri = 256, -1   # right camera of a rig (src/ORBmatcher.cc:222-229)
            for jf in fIdx[fOff[g]:fOff[g + 1]]:
                if out[jf] >= 0:
                    continue
                d = int(np.unpackbits(kfDesc[ik] ^ fDesc[jf]).sum())
                if nLeft == -1 or jf < nLeft:
                    if d < b1:
                        b2, b1, bi = b1, d, jf
                    elif d < b2:
                        b2 = d
                elif d < r1:
                    r1, ri = d, jf

            def accept(idx):
                out[idx] = int(ik)
                if orient:
                    rot = f32(kfAng[ik] - fAng[idx])
                    if rot < 0:
                        rot = f32(rot + f32(360.0))
                    b = py_round(float(f32(rot * f32(f32(1.0) / f32(30)))))
                    hist[0 if b == 30 else b].append(idx)

            if b1 <= 30:
                if f32(b1) < f32(f32(nn) * f32(b2)):
                    accept(bi)
                    nm += 1
                if r1 <= 30:   # :263-286, ratio test "|| true"
                    accept(ri)
                    nm += 1
    if orient:
        m1 = m2 = m3 = 0
        i1 = i2 = i3 = -1
        for i in range(30):
            s = len(hist[i])
            if s > m1:
                m3, m2, m1, i3, i2, i1 = m2, m1, s, i2, i1, i
            elif s > m2:
                m3, m2, i3, i2 = m2, s, i2, i
            elif s > m3:
                m3, i3 = s, i
        if f32(m2) < f32(f32(0.1) * f32(m1)):
            i2 = i3 = -1
        elif f32(m3) < f32(f32(0.1) * f32(m1)):
            i3 = -1
        for i in range(30):
            if i in (i1, i2, i3):
                continue
            for j in hist[i]:
                out[j] = -1
                nm -= 1
    return nm, out


@pytest.mark.parametrize("orient,nodes,seed", [(True, 40, 1), (False, 40, 2), (True, 6, 3)])
def test_bow_oracle_equals_python(orient, nodes, seed):
    W, H = 320, 240
    e = O.Extractor(300, 20000, 1.2, 4, 20, 7, W, H)
    frames = list(synth.stream(W, H, 2, index0=5))
    kpk, dk, _ = e.extract(frames[0])
    kpf, df, _ = e.extract(frames[1])
    kfOff, kfIdx, fOff, fIdx, has = S.bow_scenario(kpk, dk, kpf, df, nodes, seed)
    n, out = O.search_by_bow(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient)
    n2, out2 = py_bow(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient)
    assert n == n2 and list(out) == out2
    assert n == sum(1 for v in out if v >= 0)


@pytest.mark.parametrize("orient,nodes,seed,left_frac", [(True, 40, 4, 0.5), (False, 25, 5, 0.7), (True, 8, 6, 0.3)])
def test_bow_rig_oracle_equals_python(orient, nodes, seed, left_frac):
    """F->Nleft != -1 (src/ORBmatcher.cc:205-233, 263-286): frame features >= Nleft are the right camera's."""
    W, H = 320, 240
    e = O.Extractor(300, 20000, 1.2, 4, 20, 7, W, H)
    frames = list(synth.stream(W, H, 2, index0=9))
    kpk, dk, _ = e.extract(frames[0])
    kpf, df, _ = e.extract(frames[1])
    kfOff, kfIdx, fOff, fIdx, has = S.bow_scenario(kpk, dk, kpf, df, nodes, seed)
    nLeft = int(len(df) * left_frac)
    n, out = O.search_by_bow(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient, nLeft=nLeft)
    n2, out2 = py_bow(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient, nLeft)
    assert n == n2 and list(out) == out2
    n1, out1 = O.search_by_bow(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient)
    assert list(out1) != out2  # the rig branch changes the result on this scenario
