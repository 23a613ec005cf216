#!/usr/bin/env python3
"""Per-frame wall time of the node-side image preparation at the node's own configuration
(mono_inertial_node.cpp:20,59-71: 2048x1536 BGR -> 614x460 grey) through the host-pointer API, with and without the
chained extraction, next to the single-thread C oracle.  DESIGN.md section 6 quotes it; never the bench's `value`."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import torch  # noqa: E402  (pinned host buffers + device memory for the device-pointer entry)

import oracle_py as O  # noqa: E402
import orbfe  # noqa: E402
import test_prep as TP  # noqa: E402

W, H, DW, DH = 2048, 1536, 614, 460
ARGS = (1000, 20000, 1.2, 8, 20, 7, DW, DH)


def timed(fn, reps):
    fn()
    t = time.perf_counter()
    for _ in range(reps):
        fn()
    return (time.perf_counter() - t) / reps * 1e3


def main():
    img = TP.colour_image(W, H, 1)
    m1, m2 = TP.fisheye_maps(W, H, 1)
    ex = orbfe.ORBextractor(*ARGS)
    prep = orbfe.ImagePreparer(ex, m1, m2, DW, DH)
    ref = O.prepare_image(img, m1, m2, DW, DH)
    assert np.array_equal(prep.prepare(img), ref)
    pinned = torch.from_numpy(img.copy()).pin_memory().numpy()
    print("prepare, pageable source : %.3f ms/frame" % timed(lambda: prep.prepare(img), 50))
    print("prepare, pinned source   : %.3f ms/frame" % timed(lambda: prep.prepare(pinned), 50))
    print("prepare + extract, pinned: %.3f ms/frame" % timed(lambda: prep.extract(pinned), 50))
    grey = prep.prepare(pinned)
    print("extract alone (614x460)  : %.3f ms/frame" % timed(lambda: ex.extractFeatures(grey), 50))
    d_img = torch.from_numpy(img).cuda()
    d_grey = torch.zeros((DH, DW), dtype=torch.uint8, device="cuda")
    s = torch.cuda.current_stream()

    def dev():
        for _ in range(20):
            prep.prepare_device(d_img.data_ptr(), W * 3, d_grey.data_ptr(), DW, s.cuda_stream)
        torch.cuda.synchronize()
    print("kernel only (device ptrs): %.3f ms/frame" % (timed(dev, 10) / 20))
    assert np.array_equal(d_grey.cpu().numpy(), ref)
    print("oracle, 1 thread (full undistorted image as the reference computes it): %.1f ms/frame"
          % timed(lambda: O.prepare_image(img, m1, m2, DW, DH), 3))


if __name__ == "__main__":
    main()
