#!/usr/bin/env python3
"""How far is SPEC DECISION S1 (fixed-point resize + Gaussian) from the published OpenCV-CUDA forms the reference calls?

The reference builds its pyramid with cv::cuda::resize(INTER_LINEAR) and cv::cuda::createGaussianFilter(5x5, sigma 1.2,
BORDER_REFLECT_101) (src/ORBextractor.cc:145,612,620-621) -- third-party code that is neither in /root/reference nor in
this image, so it cannot be run here (parity unpinned).  What CAN be done is to restate the
PUBLISHED form of those two operators in binary32, one operation per line, and measure the distance:

  resize   (opencv_contrib cudawarping, resize_linear kernel): src_x = dst_x * fx with fx = float(1 / (dst_w / src_w)),
           x1 = floor(src_x), x2 = x1 + 1 (read index clamped to the last column), four products accumulated from 0
           in the order (y1,x1) (y1,x2) (y2,x1) (y2,x2) with the weights (x2 - src_x)(y2 - src_y) ..., stored through
           saturate_cast<uchar> (round half to even).  No half-pixel shift.
  Gaussian (cudafilters separable linear filter): kernel = getGaussianKernel(5, 1.2) computed in double, normalised,
           stored as float; row pass u8 -> f32 (sum += src * k, k = 0..4), column pass f32 -> u8 (saturate_cast), borders
           REFLECT_101.  The reference image is built with CUDA_FAST_MATH, so the multiply-adds may be contracted: both
           the separately rounded and the fused (FMA) accumulation are evaluated.

Outputs (JSON): per workload and level the histogram of pixel differences S1 - float form (unblurred and blurred
pyramids), and downstream -- the same FAST / quadtree / orientation / BRIEF stages (the CPU oracle's, tests/oracle_py.py)
run on both pyramids -- the keypoint-set overlap, orientation drift and descriptor Hamming drift.
Usage: opencv_float_study.py [out.json] [frames per workload]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "orb_slam3_v1.0_amd", "python"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402
import oracle_py as O  # noqa: E402
from orbfe import synth  # noqa: E402

f32 = np.float32
WORKLOADS = {
    "C1_752x480": (1000, 40000, 1.2, 8, 20, 7, 752, 480),
    "C4_1280x720": (2000, 100000, 1.2, 8, 20, 7, 1280, 720),
    "C5_1024x1024_L12": (1500, 100000, 1.2, 12, 20, 7, 1024, 1024),
}


def rn_u8(v):
    """saturate_cast<uchar>(float): round half to even, clamp"""
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def resize_linear_f32(src, dw, dh):
    sh, sw = src.shape
    fx = f32(1.0 / (float(dw) / float(sw)))
    fy = f32(1.0 / (float(dh) / float(sh)))
    sx = (np.arange(dw, dtype=f32) * fx).astype(f32)
    sy = (np.arange(dh, dtype=f32) * fy).astype(f32)
    x1 = np.floor(sx).astype(np.int32)
    y1 = np.floor(sy).astype(np.int32)
    x2, y2 = x1 + 1, y1 + 1
    x2r, y2r = np.minimum(x2, sw - 1), np.minimum(y2, sh - 1)
    s = src.astype(f32)
    wx2 = (x2.astype(f32) - sx).astype(f32)[None, :]   # weight of column x1
    wx1 = (sx - x1.astype(f32)).astype(f32)[None, :]   # weight of column x2
    wy2 = (y2.astype(f32) - sy).astype(f32)[:, None]
    wy1 = (sy - y1.astype(f32)).astype(f32)[:, None]
    out = np.zeros((dh, dw), f32)
    out = (out + (s[y1][:, x1] * (wx2 * wy2).astype(f32)).astype(f32)).astype(f32)
    out = (out + (s[y1][:, x2r] * (wx1 * wy2).astype(f32)).astype(f32)).astype(f32)
    out = (out + (s[y2r][:, x1] * (wx2 * wy1).astype(f32)).astype(f32)).astype(f32)
    out = (out + (s[y2r][:, x2r] * (wx1 * wy1).astype(f32)).astype(f32)).astype(f32)
    return rn_u8(out)


def gaussian_kernel_f32():
    k = np.exp(-np.square(np.arange(5, dtype=np.float64) - 2.0) / (2.0 * 1.2 * 1.2))
    return (k / k.sum()).astype(f32)


def gauss5_f32(src, fused):
    h, w = src.shape
    k = gaussian_kernel_f32()

    def pass1d(a, axis):
        p = np.pad(a, [(2, 2) if ax == axis else (0, 0) for ax in range(2)], mode="reflect")  # REFLECT_101
        acc = np.zeros(a.shape, f32)
        for t in range(5):
            sl = [slice(None)] * 2
            sl[axis] = slice(t, t + a.shape[axis])
            v = p[tuple(sl)].astype(f32)
            if fused:   # fmaf: exact product + sum, one rounding
                acc = (v.astype(np.float64) * np.float64(k[t]) + acc.astype(np.float64)).astype(f32)
            else:
                acc = (acc + (v * k[t]).astype(f32)).astype(f32)
        return acc

    row = pass1d(src.astype(f32), 1)
    return rn_u8(pass1d(row, 0))


def extract_on_pyramid(e, args, imgs, blurs):
    """ComputeKeyPointsOctTree + orientation + descriptors (src/ORBextractor.cc:433-585) on a GIVEN pyramid, assembled from
    the oracle's own stage functions (validated below against O.Extractor.extract on the S1 pyramid)."""
    nFeat, nFast, _, nL, iniTh, minTh, _, _ = args
    kps, descs = [], []
    for l in range(nL):
        img, blur = imgs[l], blurs[l]
        H, W = img.shape
        xyH, rH, _ = O.fast_detect(img, iniTh, nFast)
        xy, resp = xyH, rH
        if float(nFast - len(rH)) > 0.25 * nFast:
            xyL, rL, _ = O.fast_detect(img, minTh, nFast)
            keep = min(len(rL), nFast - len(rH))
            xy = np.concatenate([xyH, xyL[:keep]])
            resp = np.concatenate([rH, rL[:keep]])
        if len(resp) == 0:
            continue
        sel, n = O.distribute(xy, resp, W, H, int(e.featuresPerLevel[l]))
        for i in sel[:max(n, 0)]:
            x, y = int(xy[i, 0]), int(xy[i, 1])
            ang = O.ic_angle(img, x, y)
            kps.append((l, x, y, int(resp[i]), ang))
            descs.append(O.brief(blur, x, y, ang))
    return kps, (np.stack(descs) if descs else np.zeros((0, 32), np.uint8))


def hist(d):
    v, c = np.unique(d, return_counts=True)
    return {int(a): int(b) for a, b in zip(v, c)}


def study(name, args, n_frames):
    W, H, nL = args[6], args[7], args[3]
    e = O.Extractor(*args)
    res = {"levels": [[int(e.levelW[l]), int(e.levelH[l])] for l in range(nL)], "frames": n_frames}
    px = {"unblurred": [dict() for _ in range(nL)], "blurred_plain": [dict() for _ in range(nL)], "blurred_fma": [dict() for _ in range(nL)]}
    down = {k: {"kp_s1": 0, "kp_float": 0, "common": 0, "angle_diff_deg_max": 0.0, "angle_diff_gt_1deg": 0, "desc_bits_differing": 0,
                "desc_identical": 0, "response_diff": 0} for k in ("plain", "fma")}
    for fi in range(n_frames):
        img = synth.frame(W, H, 100 + fi)
        kp_ref, desc_ref, _ = e.extract(img)
        s1_img = [e.level_image(l, False) for l in range(nL)]
        s1_blur = [e.level_image(l, True) for l in range(nL)]
        if fi == 0:   # the Python orchestration reproduces the C oracle on its own pyramid
            kps, descs = extract_on_pyramid(e, args, s1_img, s1_blur)
            assert len(kps) == len(kp_ref) and np.array_equal(descs, desc_ref)
            assert all(k[1] == int(r["x"]) and k[2] == int(r["y"]) and k[0] == int(r["octave"]) for k, r in zip(kps, kp_ref))
        fl_img = [img]
        for l in range(1, nL):
            fl_img.append(resize_linear_f32(fl_img[l - 1], int(e.levelW[l]), int(e.levelH[l])))
        for l in range(nL):
            d = s1_img[l].astype(np.int32) - fl_img[l].astype(np.int32)
            for k, v in hist(d).items():
                px["unblurred"][l][k] = px["unblurred"][l].get(k, 0) + v
        kps_s1, d_s1 = extract_on_pyramid(e, args, s1_img, s1_blur)
        idx_s1 = {(k[0], k[1], k[2]): i for i, k in enumerate(kps_s1)}
        for mode, fused in (("plain", False), ("fma", True)):
            fl_blur = [gauss5_f32(fl_img[l], fused) for l in range(nL)]
            for l in range(nL):
                d = s1_blur[l].astype(np.int32) - fl_blur[l].astype(np.int32)
                for k, v in hist(d).items():
                    px["blurred_" + mode][l][k] = px["blurred_" + mode][l].get(k, 0) + v
            kps_f, d_f = extract_on_pyramid(e, args, fl_img, fl_blur)
            D = down[mode]
            D["kp_s1"] += len(kps_s1)
            D["kp_float"] += len(kps_f)
            for j, k in enumerate(kps_f):
                i = idx_s1.get((k[0], k[1], k[2]))
                if i is None:
                    continue
                D["common"] += 1
                da = abs(k[4] - kps_s1[i][4])
                da = min(da, 360.0 - da)
                D["angle_diff_deg_max"] = max(D["angle_diff_deg_max"], float(da))
                D["angle_diff_gt_1deg"] += int(da > 1.0)
                bits = int(np.unpackbits(np.bitwise_xor(d_f[j], d_s1[i])).sum())
                D["desc_bits_differing"] += bits
                D["desc_identical"] += int(bits == 0)
                D["response_diff"] += int(k[3] != kps_s1[i][3])
    res["pixel_diff_histograms_S1_minus_float"] = px
    for mode, D in down.items():
        D["overlap_of_s1_keypoints"] = D["common"] / max(1, D["kp_s1"])
        D["mean_desc_bits_differing_on_common"] = D["desc_bits_differing"] / max(1, D["common"])
    res["downstream"] = down
    for key in px:
        tot = sum(sum(h.values()) for h in px[key])
        nz = sum(v for h in px[key] for k, v in h.items() if k != 0)
        res["fraction_of_pixels_differing_" + key] = nz / max(1, tot)
        res["max_abs_pixel_diff_" + key] = max(abs(k) for h in px[key] for k in h)
    return res


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "r02_opencv_float_distance.json")
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 2
    report = {"what": __doc__.split("\n\n")[0], "spec": "S1 (DESIGN.md section 2) vs the published OpenCV-CUDA forms in binary32",
              "workloads": {}}
    for name, args in WORKLOADS.items():
        report["workloads"][name] = study(name, args, n)
        r = report["workloads"][name]
        print(name, "px differing: unblurred %.3f, blurred %.3f (max |d| %d / %d); keypoint overlap %.4f, mean descriptor bits %.2f" % (
            r["fraction_of_pixels_differing_unblurred"], r["fraction_of_pixels_differing_blurred_plain"],
            r["max_abs_pixel_diff_unblurred"], r["max_abs_pixel_diff_blurred_plain"],
            r["downstream"]["plain"]["overlap_of_s1_keypoints"], r["downstream"]["plain"]["mean_desc_bits_differing_on_common"]), flush=True)
    json.dump(report, open(out, "w"), indent=1)


if __name__ == "__main__":
    main()
