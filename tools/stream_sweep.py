#!/usr/bin/env python3
"""frames/s of the host-pointer ring (orbfe_stream_*) against ring depth and slot size (pinned source)"""
import os
import sys
import time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402
import orbfe  # noqa: E402
from orbfe import synth  # noqa: E402

ARGS = (1000, 40000, 1.2, 8, 20, 7, 752, 480)
frames = np.stack(list(synth.stream(752, 480, 1024)))
src = torch.from_numpy(frames).pin_memory().numpy()
for slots, sf, pin in ((3, 256, 1), (3, 256, 0), (4, 256, 0), (3, 512, 0), (3, 256, 1), (3, 256, 0)):
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=sf)
    st = ex.stream(slots=slots, slot_frames=sf)
    chunks = [(i, min(sf, len(src) - i)) for i in range(0, len(src), sf)]

    def run(rounds, view):
        todo = [c for _ in range(rounds) for c in chunks]
        pos = done = 0
        while pos < len(todo) or st.in_flight():
            while pos < len(todo):
                lo, n = todo[pos]
                if not st.submit((src if pin else frames)[lo:lo + n]):
                    break
                pos += 1
            done += st.collect_raw()[0]
        return done
    run(1, False)
    t = time.perf_counter()
    d = run(24, False)
    dt = time.perf_counter() - t
    print("slots %d x %3d frames, %s source: %.0f frames/s (copy-out)" % (slots, sf, "pinned" if pin else "pageable", d / dt), flush=True)
    st.close()
    del ex
