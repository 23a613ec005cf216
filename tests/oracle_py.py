"""ctypes loader for the CPU oracle (oracle/liborb_oracle.so).  Tests / bench cpu_baseline only."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "oracle")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<i4"), ("size", "<f4"),
                     ("octave", "<i4"), ("angle", "<f4")])
assert KP_DTYPE.itemsize == 24

MP_DTYPE = np.dtype([("projX", "<f4"), ("projY", "<f4"), ("viewCos", "<f4"), ("trackDepth", "<f4"),
                     ("level", "<i4"), ("inView", "<i4"), ("bad", "<i4"), ("observations", "<i4")])


WP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("minDistance", "<f4"), ("maxDistance", "<f4"),
                     ("bad", "<i4"), ("observations", "<i4"), ("skip", "<i4")])


class Frustum(C.Structure):
    _fields_ = [("rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("twc", C.c_float * 3), ("minX", C.c_float),
                ("maxX", C.c_float), ("minY", C.c_float), ("maxY", C.c_float), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("k1", C.c_float), ("k2", C.c_float), ("k3", C.c_float),
                ("k4", C.c_float), ("mbf", C.c_float), ("logScaleFactor", C.c_float),
                ("nLevels", C.c_int), ("cameraModel", C.c_int)]


class Sim3Dir(C.Structure):
    _fields_ = [("rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("sr", C.c_float * 9), ("t", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("minX", C.c_float),
                ("maxX", C.c_float), ("minY", C.c_float), ("maxY", C.c_float), ("logScaleFactor", C.c_float),
                ("nLevels", C.c_int)]


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int), ("kp", C.c_void_p), ("desc", C.c_void_p), ("gridCols", C.c_int),
                ("gridRows", C.c_int), ("minX", C.c_float), ("minY", C.c_float),
                ("gridInvW", C.c_float), ("gridInvH", C.c_float), ("nLevels", C.c_int),
                ("scaleFactors", C.c_void_p)]


def build(asan=False):
    target = "liborb_oracle_asan.so" if asan else "liborb_oracle.so"
    subprocess.check_call(["make", "-s", "-C", ODIR, target])
    return os.path.join(ODIR, target)


_lib = None


def lib():
    global _lib
    if _lib is None:
        path = os.path.join(ODIR, "liborb_oracle.so")
        srcs = ("orb_oracle.c", "match_oracle.c", "prep_oracle.c", "orb_oracle.h")
        if os.environ.get("ORB_ORACLE_ASAN"):  # tests/test_oracle_asan.py: the sanitizer build of the same sources
            path = build(asan=True)
        elif os.environ.get("ORB_ORACLE_VARIANT"):  # tools/s5_libm_study.py: liborb_oracle_libm.so (-DORC_LIBM), never the parity authority
            target = "liborb_oracle_%s.so" % os.environ["ORB_ORACLE_VARIANT"]
            subprocess.check_call(["make", "-s", "-C", ODIR, target])
            path = os.path.join(ODIR, target)
        elif not os.path.exists(path) or any(
                os.path.getmtime(os.path.join(ODIR, f)) > os.path.getmtime(path) for f in srcs):
            build()
        L = C.CDLL(path)
        vp, ci, cf = C.c_void_p, C.c_int, C.c_float
        L.orc_create.restype = vp
        L.orc_create.argtypes = [ci, ci, cf, ci, ci, ci, ci, ci]
        L.orc_destroy.argtypes = [vp]
        L.orc_get_tables.argtypes = [vp] + [vp] * 8
        L.orc_max_keypoints.argtypes = [vp]
        L.orc_extract.argtypes = [vp, vp, ci, vp, vp, vp]
        L.orc_level_image.restype = vp
        L.orc_level_image.argtypes = [vp, ci, ci]
        L.orc_level_candidates.argtypes = [vp, ci, vp, vp, ci, vp, vp, vp]
        L.orc_resize_bilinear.argtypes = [vp, ci, ci, ci, vp, ci, ci, ci]
        L.orc_gauss5.argtypes = [vp, ci, ci, ci, vp, ci]
        L.orc_fast_detect.argtypes = [vp, ci, ci, ci, ci, ci, vp, vp, vp]
        L.orc_fast_score.argtypes = [vp, ci, ci, ci, ci]
        L.orc_fast_arc9.argtypes = [ci]
        L.orc_distribute.argtypes = [ci, vp, vp, ci, ci, ci, vp, ci]
        L.orc_ic_angle.restype = cf
        L.orc_ic_angle.argtypes = [vp, ci, ci, ci, ci, ci]
        L.orc_brief.argtypes = [vp, ci, ci, ci, ci, ci, cf, vp]
        L.orc_atan2_deg.restype = cf
        L.orc_atan2_deg.argtypes = [cf, cf]
        L.orc_cos_sin_deg.argtypes = [cf, vp, vp]
        L.orc_hamming.argtypes = [vp, vp]
        L.orc_search_by_projection.argtypes = [vp, ci, vp, vp, vp, cf, ci, cf, cf, vp]
        L.orc_search_by_bow.argtypes = [ci, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, cf, ci, vp]
        L.orc_search_by_bow_rig.argtypes = [ci, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, ci, cf, ci, vp]
        L.orc_assign_grid.argtypes = [vp, vp]
        L.orc_search_for_initialization.argtypes = [vp, vp, ci, cf, ci, vp]
        L.orc_vocab_transform.argtypes = [ci, vp, vp, vp, vp, vp, ci, vp, ci, ci, vp, vp, vp]
        L.orc_vocab_transform.restype = None
        L.orc_spec_atan2f.argtypes = [cf, cf]
        L.orc_spec_atan2f.restype = cf
        L.orc_spec_logf.argtypes = [cf]
        L.orc_spec_logf.restype = cf
        L.orc_is_in_frustum.argtypes = [C.POINTER(Frustum), ci, vp, vp, vp]
        L.orc_is_in_frustum.restype = None
        L.orc_fuse_search.argtypes = [C.POINTER(FrameView), vp, vp, C.POINTER(Frustum), cf, ci, vp, vp, vp, vp]
        L.orc_fuse_search.restype = None
        L.orc_prepare_image.argtypes = [vp, ci, ci, ci, vp, vp, ci, ci, vp, ci, vp]
        L.orc_prepare_image.restype = None
        L.orc_prep_remap_pixel.argtypes = [vp, ci, ci, ci, cf, cf, vp]
        L.orc_prep_remap_pixel.restype = None
        L.orc_prep_scale.argtypes = [ci, ci]
        L.orc_prep_scale.restype = cf
        L.orc_fuse_search_sim3.argtypes = [C.POINTER(FrameView), C.POINTER(Frustum), cf, ci, vp, vp, vp, vp]
        L.orc_fuse_search_sim3.restype = None
        L.orc_search_by_sim3.argtypes = [C.POINTER(FrameView), C.POINTER(FrameView), C.POINTER(Sim3Dir),
                                         C.POINTER(Sim3Dir), vp, vp, vp, vp, cf, vp]
        L.orc_search_by_sim3.restype = ci
        L.orc_search_by_projection_kf.argtypes = [C.POINTER(FrameView), C.POINTER(Frustum), ci, vp, vp, vp, vp, cf, ci, vp]
        L.orc_search_by_projection_kf.restype = ci
        L.orc_distinctive_descriptors.argtypes = [ci, vp, vp, vp, vp]
        L.orc_distinctive_descriptors.restype = None
        L.orc_search_for_triangulation.argtypes = [ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, vp, cf, cf,
                                                   ci, ci, ci, vp]
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Extractor:
    """Mirror of ORB_SLAM3::ORBextractor (include/ORBextractor.h:52-130) on the CPU oracle."""

    def __init__(self, nFeatures, nFastFeatures, scaleFactor, nLevels, iniThFAST, minThFAST, W, H):
        self.L = lib()
        self.h = self.L.orc_create(nFeatures, nFastFeatures, scaleFactor, nLevels, iniThFAST, minThFAST, W, H)
        if not self.h:
            raise ValueError("orc_create failed")
        self.nLevels, self.W, self.H, self.nFast = nLevels, W, H, nFastFeatures
        self.cap = self.L.orc_max_keypoints(self.h)
        n = nLevels
        self.scaleFactors = np.zeros(n, np.float32)
        self.invScaleFactors = np.zeros(n, np.float32)
        self.levelSigma2 = np.zeros(n, np.float32)
        self.invLevelSigma2 = np.zeros(n, np.float32)
        self.featuresPerLevel = np.zeros(n, np.int32)
        self.umax = np.zeros(16, np.int32)
        self.levelW = np.zeros(n, np.int32)
        self.levelH = np.zeros(n, np.int32)
        self.L.orc_get_tables(self.h, _p(self.scaleFactors), _p(self.invScaleFactors), _p(self.levelSigma2),
                              _p(self.invLevelSigma2), _p(self.featuresPerLevel), _p(self.umax),
                              _p(self.levelW), _p(self.levelH))

    def __del__(self):
        if getattr(self, "h", None):
            self.L.orc_destroy(self.h)
            self.h = None

    def extract(self, img):
        img = np.ascontiguousarray(img, np.uint8)
        assert img.shape == (self.H, self.W)
        kp = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        per = np.zeros(self.nLevels, np.int32)
        n = self.L.orc_extract(self.h, _p(img), img.strides[0], _p(kp), _p(desc), _p(per))
        return kp[:n].copy(), desc[:n].copy(), per

    def level_image(self, level, blurred=False):
        ptr = self.L.orc_level_image(self.h, level, int(blurred))
        w, h = int(self.levelW[level]), int(self.levelH[level])
        buf = (C.c_uint8 * (w * h)).from_address(ptr)
        return np.frombuffer(buf, np.uint8).reshape(h, w).copy()

    def level_candidates(self, level):
        cap = max(1, self.nFast)
        xy = np.zeros((cap, 2), np.int16)
        resp = np.zeros(cap, np.int32)
        nh, ph, pl = C.c_int(), C.c_int(), C.c_int()
        n = self.L.orc_level_candidates(self.h, level, _p(xy), _p(resp), cap, C.byref(nh), C.byref(ph), C.byref(pl))
        return xy[:n].copy(), resp[:n].copy(), nh.value, ph.value, pl.value


def resize_bilinear(src, dw, dh):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros((dh, dw), np.uint8)
    lib().orc_resize_bilinear(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dw, dh, dw)
    return dst


def gauss5(src):
    src = np.ascontiguousarray(src, np.uint8)
    dst = np.zeros_like(src)
    lib().orc_gauss5(_p(src), src.shape[1], src.shape[0], src.strides[0], _p(dst), dst.strides[0])
    return dst


def fast_detect(img, th, maxKp):
    img = np.ascontiguousarray(img, np.uint8)
    xy = np.zeros((max(1, maxKp), 2), np.int16)
    resp = np.zeros(max(1, maxKp), np.int32)
    pre = C.c_int()
    n = lib().orc_fast_detect(_p(img), img.shape[1], img.shape[0], img.strides[0], th, maxKp, _p(xy), _p(resp), C.byref(pre))
    return xy[:n].copy(), resp[:n].copy(), pre.value


def fast_score(img, x, y, th):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_fast_score(_p(img), img.strides[0], x, y, th)


def distribute(xy, resp, W, H, maxFeatures):
    xy = np.ascontiguousarray(xy, np.int16)
    resp = np.ascontiguousarray(resp, np.int32)
    cap = max(maxFeatures + 16, 64) + 4 * max(1, int(round(W / H)))
    sel = np.zeros(cap, np.int32)
    n = lib().orc_distribute(len(resp), _p(xy), _p(resp), W, H, maxFeatures, _p(sel), cap)
    return sel[:max(n, 0)].copy(), n


def ic_angle(img, x, y):
    img = np.ascontiguousarray(img, np.uint8)
    return lib().orc_ic_angle(_p(img), img.shape[1], img.shape[0], img.strides[0], x, y)


def brief(img, x, y, angle):
    img = np.ascontiguousarray(img, np.uint8)
    d = np.zeros(32, np.uint8)
    lib().orc_brief(_p(img), img.shape[1], img.shape[0], img.strides[0], x, y, angle, _p(d))
    return d


def atan2_deg(m01, m10):
    return lib().orc_atan2_deg(float(m01), float(m10))


def cos_sin_deg(a):
    c, s = C.c_float(), C.c_float()
    lib().orc_cos_sin_deg(float(a), C.byref(c), C.byref(s))
    return c.value, s.value


def hamming(a, b):
    a = np.ascontiguousarray(a, np.uint8)
    b = np.ascontiguousarray(b, np.uint8)
    return lib().orc_hamming(_p(a), _p(b))


def make_frame_view(kp, desc, gridCols, gridRows, minX, minY, maxX, maxY, scaleFactors):
    """Frame grid statics as src/Frame.cc:101-105 computes them."""
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    sf = np.ascontiguousarray(scaleFactors, np.float32)
    invw = np.float32(gridCols) / np.float32(np.float32(maxX) - np.float32(minX))
    invh = np.float32(gridRows) / np.float32(np.float32(maxY) - np.float32(minY))
    fv = FrameView(len(kp), kp.ctypes.data, desc.ctypes.data, gridCols, gridRows, minX, minY,
                   float(invw), float(invh), len(sf), sf.ctypes.data)
    fv._keep = (kp, desc, sf)
    return fv


def search_by_projection(fv, mps, mpDesc, initObs, th, nnRatio, bFarPoints=False, thFarPoints=0.0):
    mps = np.ascontiguousarray(mps, MP_DTYPE)
    mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
    out = np.zeros(max(1, fv.n), np.int32)
    io = None if initObs is None else np.ascontiguousarray(initObs, np.int32)
    n = lib().orc_search_by_projection(C.byref(fv), len(mps), _p(mps), _p(mpDesc), _p(io), th,
                                       int(bFarPoints), thFarPoints, nnRatio, _p(out))
    return n, out[:fv.n].copy()


def assign_grid(fv):
    out = np.zeros(max(1, fv.n), np.int32)
    lib().orc_assign_grid(C.byref(fv), _p(out))
    return out[:fv.n].copy()


def search_by_bow(kfOff, kfIdx, fOff, fIdx, kfDesc, kfAngle, kfHasMP, fDesc, fAngle, nnRatio, checkOrientation=True, nLeft=-1):
    kfOff = np.ascontiguousarray(kfOff, np.int32)
    kfIdx = np.ascontiguousarray(kfIdx, np.int32)
    fOff = np.ascontiguousarray(fOff, np.int32)
    fIdx = np.ascontiguousarray(fIdx, np.int32)
    kfDesc = np.ascontiguousarray(kfDesc, np.uint8)
    fDesc = np.ascontiguousarray(fDesc, np.uint8)
    kfAngle = np.ascontiguousarray(kfAngle, np.float32)
    fAngle = np.ascontiguousarray(fAngle, np.float32)
    kfHasMP = np.ascontiguousarray(kfHasMP, np.uint8)
    out = np.zeros(max(1, len(fDesc)), np.int32)
    n = lib().orc_search_by_bow_rig(len(kfOff) - 1, _p(kfOff), _p(kfIdx), _p(fOff), _p(fIdx), len(kfDesc),
                                    _p(kfDesc), _p(kfAngle), _p(kfHasMP), len(fDesc), _p(fDesc), _p(fAngle),
                                    int(nLeft), nnRatio, int(checkOrientation), _p(out))
    return n, out[:len(fDesc)].copy()


def search_for_initialization(fv1, fv2, windowSize, nnRatio, checkOrientation=True):
    out = np.zeros(max(1, fv1.n), np.int32)
    n = lib().orc_search_for_initialization(C.byref(fv1), C.byref(fv2), int(windowSize), nnRatio, int(checkOrientation), _p(out))
    return n, out[:fv1.n].copy()


def vocab_transform(childOff, childIdx, nodeDesc, wordId, weight, L, desc, levelsup):
    a32 = lambda v: np.ascontiguousarray(v, np.int32)
    childOff, childIdx, wordId = a32(childOff), a32(childIdx), a32(wordId)
    nodeDesc = np.ascontiguousarray(nodeDesc, np.uint8)
    weight = np.ascontiguousarray(weight, np.float64)
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(desc)
    word = np.zeros(max(n, 1), np.int32)
    node = np.zeros(max(n, 1), np.int32)
    w = np.zeros(max(n, 1), np.float64)
    lib().orc_vocab_transform(len(wordId), _p(childOff), _p(childIdx), _p(nodeDesc), _p(wordId), _p(weight), int(L),
                              _p(desc), n, int(levelsup), _p(word), _p(node), _p(w))
    return word[:n], node[:n], w[:n]


def spec_logf(x):
    return float(lib().orc_spec_logf(float(np.float32(x))))


def is_in_frustum(frustum, points):
    points = np.ascontiguousarray(points, WP_DTYPE)
    n = len(points)
    out = np.zeros(max(n, 1), MP_DTYPE)
    xr = np.zeros(max(n, 1), np.float32)
    lib().orc_is_in_frustum(C.byref(frustum), n, _p(points), _p(out), _p(xr))
    return out[:n], xr[:n]


def fuse_search(kf_view, invLevelSigma2, uRight, frustum, th, points, mpDesc):
    points = np.ascontiguousarray(points, WP_DTYPE)
    mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
    is2 = np.ascontiguousarray(invLevelSigma2, np.float32)
    ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
    M = len(points)
    bi = np.zeros(max(M, 1), np.int32)
    bd = np.zeros(max(M, 1), np.int32)
    lib().orc_fuse_search(C.byref(kf_view), _p(is2), _p(ur), C.byref(frustum), th, M, _p(points), _p(mpDesc), _p(bi), _p(bd))
    return bi[:M], bd[:M]


def fuse_search_right(kf_left_view, nRight, invLevelSigma2, uRight, frustum, th, points, mpDesc):
    points = np.ascontiguousarray(points, WP_DTYPE)
    mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
    is2 = np.ascontiguousarray(invLevelSigma2, np.float32)
    ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
    M = len(points)
    bi = np.zeros(max(M, 1), np.int32)
    bd = np.zeros(max(M, 1), np.int32)
    L = lib()
    L.orc_fuse_search_right.restype = None
    L.orc_fuse_search_right.argtypes = [C.POINTER(FrameView), C.c_int, C.c_void_p, C.c_void_p, C.POINTER(Frustum), C.c_float, C.c_int,
                                        C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
    L.orc_fuse_search_right(C.byref(kf_left_view), int(nRight), _p(is2), _p(ur), C.byref(frustum), th, M, _p(points), _p(mpDesc),
                            _p(bi), _p(bd))
    return bi[:M], bd[:M]


def prepare_image(bgr, map1, map2, dst_w, dst_h, want_undistorted=False):
    """ImageGrabber::ConvertImageToGPU (image_grabber.hpp:96-110), SPEC DECISION S9."""
    bgr = np.ascontiguousarray(bgr, np.uint8)
    map1 = np.ascontiguousarray(map1, np.float32)
    map2 = np.ascontiguousarray(map2, np.float32)
    h, w = map1.shape
    assert bgr.shape == (h, w, 3)
    grey = np.zeros((dst_h, dst_w), np.uint8)
    und = np.zeros((h, w, 3), np.uint8) if want_undistorted else None
    lib().orc_prepare_image(bgr.ctypes.data, w * 3, w, h, _p(map1), _p(map2), dst_w, dst_h, _p(grey), dst_w, _p(und))
    return (grey, und) if want_undistorted else grey


def prep_scale(src_n, dst_n):
    return np.float32(lib().orc_prep_scale(src_n, dst_n))


def fuse_search_sim3(kf_view, frustum, th, points, mpDesc):
    points = np.ascontiguousarray(points, WP_DTYPE)
    mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
    M = len(points)
    bi = np.zeros(max(M, 1), np.int32)
    bd = np.zeros(max(M, 1), np.int32)
    lib().orc_fuse_search_sim3(C.byref(kf_view), C.byref(frustum), th, M, _p(points), _p(mpDesc), _p(bi), _p(bd))
    return bi[:M], bd[:M]


def search_by_sim3(kf1_view, kf2_view, d12, d21, mp1, mpDesc1, mp2, mpDesc2, th):
    mp1 = np.ascontiguousarray(mp1, WP_DTYPE)
    mp2 = np.ascontiguousarray(mp2, WP_DTYPE)
    mpDesc1 = np.ascontiguousarray(mpDesc1, np.uint8)
    mpDesc2 = np.ascontiguousarray(mpDesc2, np.uint8)
    out = np.full(max(1, kf1_view.n), -1, np.int32)
    n = lib().orc_search_by_sim3(C.byref(kf1_view), C.byref(kf2_view), C.byref(d12), C.byref(d21), _p(mp1), _p(mpDesc1),
                                 _p(mp2), _p(mpDesc2), th, _p(out))
    return n, out[:kf1_view.n]


def search_by_projection_kf(fv, frustum, points, mpDesc, kfAngle, frameHasMP, th, checkOrientation=True):
    points = np.ascontiguousarray(points, WP_DTYPE)
    mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
    ang = None if kfAngle is None else np.ascontiguousarray(kfAngle, np.float32)
    has = None if frameHasMP is None else np.ascontiguousarray(frameHasMP, np.uint8)
    out = np.full(max(1, fv.n), -1, np.int32)
    n = lib().orc_search_by_projection_kf(C.byref(fv), C.byref(frustum), len(points), _p(points), _p(mpDesc), _p(ang),
                                          _p(has), th, int(bool(checkOrientation)), _p(out))
    return n, out[:fv.n]


class TriCameras(C.Structure):
    """orc_tri_cameras"""
    _fields_ = [("model1", C.c_int), ("model2", C.c_int), ("cam1", C.c_float * 8), ("cam2", C.c_float * 8),
                ("precision", C.c_float), ("R12", C.c_float * 9), ("t12", C.c_float * 3), ("sigma2_1", C.c_float * 32),
                ("kf1HasCamera2", C.c_int)]


def tri_cameras(cameras):
    T = TriCameras()
    T.model1, T.model2 = int(cameras["model1"]), int(cameras["model2"])
    for i in range(8):
        T.cam1[i] = float(cameras["cam1"][i])
        T.cam2[i] = float(cameras["cam2"][i])
    T.precision = float(cameras.get("precision", 1e-6))
    for i, v in enumerate(np.asarray(cameras["R12"], np.float32).reshape(-1)):
        T.R12[i] = float(v)
    for i, v in enumerate(np.asarray(cameras["t12"], np.float32).reshape(-1)):
        T.t12[i] = float(v)
    for i, v in enumerate(np.asarray(cameras["levelSigma2_1"], np.float32).reshape(-1)):
        T.sigma2_1[i] = float(v)
    T.kf1HasCamera2 = int(cameras.get("kf1HasCamera2", 0))
    return T


def kb8_epipolar_constrain(cameras, u1, v1, u2, v2, sigmaLevel, unc=1.0):
    """(verdict, triangulated point in camera 1) of KannalaBrandt8::epipolarConstrain, SPEC DECISION S10"""
    L = lib()
    L.orc_kb8_epipolar_constrain.restype = C.c_int
    L.orc_kb8_epipolar_constrain.argtypes = [C.c_void_p] + [C.c_float] * 6 + [C.c_void_p]
    T = tri_cameras(cameras)
    xyz = np.zeros(3, np.float32)
    ok = L.orc_kb8_epipolar_constrain(C.addressof(T), float(np.float32(u1)), float(np.float32(v1)), float(np.float32(u2)),
                                      float(np.float32(v2)), float(np.float32(sigmaLevel)), float(np.float32(unc)), _p(xyz))
    return bool(ok), xyz


def kb8_unproject(cam, model, precision, u, v):
    L = lib()
    L.orc_kb8_unproject.restype = None
    L.orc_kb8_unproject.argtypes = [C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p]
    c = np.ascontiguousarray(cam, np.float32)
    rx, ry = C.c_float(), C.c_float()
    L.orc_kb8_unproject(_p(c), int(model), float(np.float32(precision)), float(np.float32(u)), float(np.float32(v)),
                        C.addressof(rx), C.addressof(ry))
    return rx.value, ry.value


def search_for_triangulation(off1, idx1, off2, idx2, kp1, desc1, hasMP1, stereo1, kp2, desc2, hasMP2, stereo2,
                             scaleFactors2, F12, ep, bOnlyStereo=False, bCoarse=False, checkOrientation=True, cameras=None):
    a32 = lambda v: np.ascontiguousarray(v, np.int32)
    u8 = lambda v: None if v is None else np.ascontiguousarray(v, np.uint8)
    off1, idx1, off2, idx2 = a32(off1), a32(idx1), a32(off2), a32(idx2)
    kp1, kp2 = np.ascontiguousarray(kp1, KP_DTYPE), np.ascontiguousarray(kp2, KP_DTYPE)
    desc1, desc2, h1, h2, s1, s2 = u8(desc1), u8(desc2), u8(hasMP1), u8(hasMP2), u8(stereo1), u8(stereo2)
    sf = np.ascontiguousarray(scaleFactors2, np.float32)
    F = np.ascontiguousarray(np.asarray(F12, np.float32).reshape(-1))
    out = np.full(max(len(kp1), 1), -1, np.int32)
    if cameras is not None:
        T = tri_cameras(cameras)
        L = lib()
        L.orc_search_for_triangulation_cam.restype = C.c_int
        L.orc_search_for_triangulation_cam.argtypes = list(L.orc_search_for_triangulation.argtypes[:-1]) + [C.c_void_p, C.c_void_p]
        n = L.orc_search_for_triangulation_cam(len(off1) - 1, _p(off1), _p(idx1), _p(off2), _p(idx2), len(kp1), _p(kp1),
                                               _p(desc1), _p(h1), _p(s1), len(kp2), _p(kp2), _p(desc2), _p(h2), _p(s2), _p(sf),
                                               _p(F), float(np.float32(ep[0])), float(np.float32(ep[1])), int(bOnlyStereo),
                                               int(bCoarse), int(checkOrientation), C.addressof(T), _p(out))
        return n, out[:len(kp1)].copy()
    n = lib().orc_search_for_triangulation(len(off1) - 1, _p(off1), _p(idx1), _p(off2), _p(idx2), len(kp1), _p(kp1),
                                           _p(desc1), _p(h1), _p(s1), len(kp2), _p(kp2), _p(desc2), _p(h2), _p(s2), _p(sf),
                                           _p(F), float(np.float32(ep[0])), float(np.float32(ep[1])), int(bOnlyStereo),
                                           int(bCoarse), int(checkOrientation), _p(out))
    return n, out[:len(kp1)].copy()


def spec_atan2f(y, x):
    return float(lib().orc_spec_atan2f(float(np.float32(y)), float(np.float32(x))))


def distinctive_descriptors(setOff, desc):
    setOff = np.ascontiguousarray(setOff, np.int32)
    desc = np.ascontiguousarray(desc, np.uint8)
    n = len(setOff) - 1
    bi = np.zeros(max(n, 1), np.int32)
    bm = np.zeros(max(n, 1), np.int32)
    lib().orc_distinctive_descriptors(n, _p(setOff), _p(desc), _p(bi), _p(bm))
    return bi[:n], bm[:n]
