#!/usr/bin/env python3
"""How far is SPEC DECISION S5 (fixed fp32 polynomial atan2 / sin / cos) from a libm?  

The reference computes the keypoint orientation with the CUDA math library's atan2f (src/cuda/Angle_gpu.cu:73-75) and
the descriptor rotation with its cosf / sinf (src/cuda/Orb_gpu.cu:327-329).  Those functions are not available here and
are documented to 2 ulp / 2 ulp / 2 ulp (with CUDA_FAST_MATH even less), so their last bits are not pinned by anything;
S5 replaces them by fixed operation sequences (~2 ulp).  This tool measures what that choice can move: the CPU oracle is
built a second time with the HOST libm (glibc: < 1 ulp) in place of the polynomials (`make -C oracle
liborb_oracle_libm.so`, -DORC_LIBM) and both builds run on the same C1 / C4 / C5 frames:
  * fraction of keypoints whose angle BITS differ, and the largest angle difference in degrees and ulps;
  * fraction of keypoints whose descriptor differs by >= 1 bit, and the mean Hamming distance of those;
  * downstream: SearchByProjection of the same 2000 map points against the frame's keypoints with either descriptor set --
    fraction of map points whose match index is the same.
Keypoint positions, levels and responses cannot differ (integer stages).  Usage: s5_libm_study.py [out.json] [frames]"""
import json
import os
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

WORKLOADS = {
    "C1_752x480": (1000, 40000, 1.2, 8, 20, 7, 752, 480),
    "C4_1280x720": (2000, 100000, 1.2, 8, 20, 7, 1280, 720),
    "C5_1024x1024_L12": (1500, 100000, 1.2, 12, 20, 7, 1024, 1024),
}


def child(out_path, n_frames):
    """extract the study frames with whichever oracle build ORB_ORACLE_VARIANT selects"""
    import oracle_py as O
    from orbfe import synth
    res = {}
    for name, a in WORKLOADS.items():
        e = O.Extractor(*a)
        for i, img in enumerate(synth.stream(a[6], a[7], n_frames, index0=500)):
            kp, desc, _ = e.extract(img)
            res["%s/%d/kp" % (name, i)] = kp
            res["%s/%d/desc" % (name, i)] = desc
    np.savez(out_path, **res)


def main():
    out_json = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r03_s5_libm_distance.json")
    n_frames = int(sys.argv[2]) if len(sys.argv) > 2 else 8
    tmp = tempfile.mkdtemp()
    paths = {}
    for variant in ("spec", "libm"):
        env = dict(os.environ)
        if variant == "libm":
            env["ORB_ORACLE_VARIANT"] = "libm"
        paths[variant] = os.path.join(tmp, variant + ".npz")
        subprocess.check_call([sys.executable, os.path.abspath(__file__), "--child", paths[variant], str(n_frames)], env=env)
    A, B = np.load(paths["spec"]), np.load(paths["libm"])
    import bench
    import oracle_py as O
    import orbfe  # dtypes only: nothing of the HIP library is called
    report = {"what": __doc__.split("\n\n")[0], "frames_per_workload": n_frames, "libm": "glibc " + os.confstr("CS_GNU_LIBC_VERSION"),
              "workloads": {}}
    for name, a in WORKLOADS.items():
        W, H = a[6], a[7]
        e = O.Extractor(*a)
        tot = ang_diff = desc_diff = 0
        max_deg, max_ulp, ham = 0.0, 0, []
        mp_total = mp_same = 0
        rng = np.random.default_rng(3)
        for i in range(n_frames):
            ka, da, kb, db = A["%s/%d/kp" % (name, i)], A["%s/%d/desc" % (name, i)], B["%s/%d/kp" % (name, i)], B["%s/%d/desc" % (name, i)]
            assert len(ka) == len(kb)
            for f in ("x", "y", "response", "size", "octave"):
                assert np.array_equal(ka[f], kb[f]), f  # integer stages: identical by construction
            tot += len(ka)
            ba, bb = ka["angle"].view(np.int32).astype(np.int64), kb["angle"].view(np.int32).astype(np.int64)
            ang_diff += int((ba != bb).sum())
            d = np.abs(ka["angle"].astype(np.float64) - kb["angle"].astype(np.float64))
            d = np.minimum(d, 360.0 - d)
            max_deg = max(max_deg, float(d.max()))
            max_ulp = max(max_ulp, int(np.abs(ba - bb)[d < 1.0].max(initial=0)))
            hd = np.unpackbits(da ^ db, axis=1).sum(axis=1)
            desc_diff += int((hd > 0).sum())
            ham += hd[hd > 0].tolist()
            # downstream: the same map points (made from the spec keypoints) against either descriptor set
            mps, mpd = bench.make_map_points(ka, len(ka), da, 2000, rng, e.nLevels, orbfe.MP_DTYPE)
            mps = mps.view(O.MP_DTYPE)
            fva = O.make_frame_view(ka, da, 64, 48, 0.0, 0.0, float(W), float(H), e.scaleFactors)
            fvb = O.make_frame_view(kb, db, 64, 48, 0.0, 0.0, float(W), float(H), e.scaleFactors)
            _, ma = O.search_by_projection(fva, mps, mpd, None, 20.0, 0.85)
            _, mb = O.search_by_projection(fvb, mps, mpd, None, 20.0, 0.85)
            mp_total += len(ma)
            mp_same += int((ma == mb).sum())
        report["workloads"][name] = {
            "keypoints": tot,
            "angle_bits_differ_frac": ang_diff / tot,
            "max_angle_difference_deg": max_deg,
            "max_angle_difference_ulp": max_ulp,
            "descriptor_differs_frac": desc_diff / tot,
            "mean_hamming_of_differing_descriptors": float(np.mean(ham)) if ham else 0.0,
            "max_hamming": int(max(ham)) if ham else 0,
            "projection_match_slots_identical_frac": mp_same / mp_total,
        }
        print(name, json.dumps(report["workloads"][name]))
    with open(out_json, "w") as f:
        json.dump(report, f, indent=1)
    print("wrote", out_json)


if __name__ == "__main__":
    if len(sys.argv) > 1 and sys.argv[1] == "--child":
        child(sys.argv[2], int(sys.argv[3]))
    else:
        main()
