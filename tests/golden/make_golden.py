#!/usr/bin/env python3
"""Generate the committed golden vectors (tests/golden/*.npz) from the CPU oracle.

The reference has no tests or golden data and cannot be built here, so these vectors
pin the ORACLE's output on small seeded inputs: the non-GPU suite checks the oracle still reproduces
them (guards against silent spec drift), the GPU suite checks the HIP path against them.
Inputs are stored in the fixture, so the check does not depend on the synthetic generator.
Run from the repo root:  python tests/golden/make_golden.py   (writes the fixtures that do not exist yet; --all rewrites all)
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
import match_scenarios as S  # noqa: E402
import oracle_py as O  # noqa: E402
from orbfe import synth  # noqa: E402

CASES = {
    # name: (nFeatures, nFast, scale, levels, iniTh, minTh, W, H, image index)
    "g160x120_l4": (200, 8000, 1.2, 4, 20, 7, 160, 120, 21),
    "g201x97_l3_s15": (120, 6000, 1.5, 3, 25, 9, 201, 97, 22),
    "g96x96_l1_cap": (60, 150, 1.2, 1, 20, 7, 96, 96, 23),  # nFast small: exercises the caps (S2b)
    # hostile classes (orbfe.synth.hostile; a 10th entry names the class): white noise with both caps active, a lattice of
    # equal-score blobs (strict-> NMS ties), a period-3 checkerboard over a tile seam
    "g144x100_l3_noise": (150, 600, 1.2, 3, 20, 7, 144, 100, 24, "noise"),
    "g136x72_l2_plateau": (120, 6000, 1.2, 2, 20, 7, 136, 72, 25, "plateau"),
    "g130x70_l3_checker3": (100, 5000, 1.25, 3, 20, 7, 130, 70, 26, "checker3"),
}
NAMES = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")


def main():
    for name, c in CASES.items():
        args, idx = c[:8], c[8]
        if os.path.exists(os.path.join(HERE, name + ".npz")) and "--all" not in sys.argv:
            continue  # committed fixtures are data: they are only rewritten on request
        img = synth.frame(args[6], args[7], idx) if len(c) == 9 else synth.hostile(c[9], args[6], args[7], idx)
        e = O.Extractor(*args)
        kp, desc, per = e.extract(img)
        out = dict(args=np.array(args, np.float64), image=img, kp=kp, desc=desc, per_level=per,
                   level1=e.level_image(min(1, args[3] - 1), False), blur0=e.level_image(0, True))
        if len(kp) > 20:
            mps, mpd, init_obs = S.projection_scenario(kp, desc, 150, 5, O.MP_DTYPE, NAMES, e.nLevels)
            fv = O.make_frame_view(kp, desc, 16, 12, 0.0, 0.0, float(args[6]), float(args[7]), e.scaleFactors)
            n, m = O.search_by_projection(fv, mps, mpd, init_obs, 20.0, 0.85)
            out.update(mps=mps, mpd=mpd, init_obs=init_obs, proj_n=np.int32(n), proj_match=m)
            kfOff, kfIdx, fOff, fIdx, has = S.bow_scenario(kp, desc, kp, np.roll(desc, 1, axis=1) ^ desc * 0 + desc, 9, 3)
            nb, mb = O.search_by_bow(kfOff, kfIdx, fOff, fIdx, desc, kp["angle"], has, desc, kp["angle"][::-1].copy(), 0.75, True)
            out.update(kfOff=kfOff, kfIdx=kfIdx, fOff=fOff, fIdx=fIdx, has=has, bow_n=np.int32(nb), bow_match=mb)
        np.savez_compressed(os.path.join(HERE, name + ".npz"), **out)
        print(name, len(kp), per, out.get("proj_n"), out.get("bow_n"))


if __name__ == "__main__":
    main()
