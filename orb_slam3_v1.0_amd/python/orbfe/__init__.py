"""orbfe -- ctypes binding of liborbfe.so (include/orbfe.h) with the reference's class names.

`ORBextractor` / `ORBmatcher` mirror include/ORBextractor.h:52-130 and include/ORBmatcher.h:36-84 of
geoeo/ORB_SLAM3_V1.0 (same constructor arguments, same method names, same results) so the parity
tests read like calls into the reference.  This module is plumbing: every result comes from the
HIP kernels behind the C ABI.  There is NO CPU fallback -- if liborbfe.so is missing or no gfx950
device is present the calls raise.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.normpath(os.path.join(_HERE, "..", "..", "csrc"))
LIB_PATH = os.path.join(CSRC, "liborbfe.so")

KP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("response", "<i4"), ("size", "<f4"),
                     ("octave", "<i4"), ("angle", "<f4")])
MP_DTYPE = np.dtype([("proj_x", "<f4"), ("proj_y", "<f4"), ("view_cos", "<f4"), ("track_depth", "<f4"),
                     ("level", "<i4"), ("in_view", "<i4"), ("bad", "<i4"), ("observations", "<i4")])
WP_DTYPE = np.dtype([("x", "<f4"), ("y", "<f4"), ("z", "<f4"), ("min_distance", "<f4"), ("max_distance", "<f4"),
                     ("bad", "<i4"), ("observations", "<i4"), ("skip", "<i4")])
NUM_STAGES = 5

STATUS = {0: "ORBFE_OK", 1: "ORBFE_ERR_INVALID_ARG", 2: "ORBFE_ERR_UNSUPPORTED", 3: "ORBFE_ERR_NO_DEVICE",
          4: "ORBFE_ERR_HIP", 5: "ORBFE_ERR_OUT_OF_MEMORY", 6: "ORBFE_ERR_INTERNAL", 7: "ORBFE_ERR_BUSY"}
ERR_BUSY = 7


class OrbfeError(RuntimeError):
    def __init__(self, code, where, detail=""):
        self.code = code
        super().__init__("%s: %s (%d) %s" % (where, STATUS.get(code, "?"), code, detail))


class Params(C.Structure):
    _fields_ = [("n_features", C.c_int), ("n_fast_features", C.c_int), ("scale_factor", C.c_float),
                ("n_levels", C.c_int), ("ini_th_fast", C.c_int), ("min_th_fast", C.c_int),
                ("image_width", C.c_int), ("image_height", C.c_int), ("device_id", C.c_int),
                ("max_batch", C.c_int)]


class Frustum(C.Structure):
    """orbfe_frustum: what Frame::isInFrustum reads from the frame (src/Frame.cc:272-331)."""
    _fields_ = [("rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("twc", C.c_float * 3), ("min_x", C.c_float),
                ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float), ("fx", C.c_float), ("fy", C.c_float),
                ("cx", C.c_float), ("cy", C.c_float), ("k1", C.c_float), ("k2", C.c_float), ("k3", C.c_float),
                ("k4", C.c_float), ("mbf", C.c_float), ("log_scale_factor", C.c_float),
                ("n_levels", C.c_int), ("camera_model", C.c_int)]


class Sim3View(C.Structure):
    """orbfe_sim3_view: one search direction of ORBmatcher::SearchBySim3 (src/ORBmatcher.cc:977-1200)."""
    _fields_ = [("rcw", C.c_float * 9), ("tcw", C.c_float * 3), ("sr", C.c_float * 9), ("t", C.c_float * 3),
                ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float), ("min_x", C.c_float),
                ("max_x", C.c_float), ("min_y", C.c_float), ("max_y", C.c_float), ("log_scale_factor", C.c_float),
                ("n_levels", C.c_int)]


class TriParams(C.Structure):
    """orbfe_tri_params: F12, epipole and the three flags of SearchForTriangulation."""
    _fields_ = [("struct_size", C.c_int), ("f12", C.c_float * 9), ("ep_x", C.c_float), ("ep_y", C.c_float), ("only_stereo", C.c_int),
                ("coarse", C.c_int), ("check_orientation", C.c_int),
                ("camera_model1", C.c_int), ("camera_model2", C.c_int), ("cam1", C.c_float * 8), ("cam2", C.c_float * 8),
                ("kb_precision", C.c_float), ("r12", C.c_float * 9), ("t12", C.c_float * 3),
                ("level_sigma2_1", C.c_float * 32), ("kf1_has_camera2", C.c_int)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(TriParams)


def fill_tri_cameras(P, cameras):
    """copy a cameras dict (see ORBmatcher.SearchForTriangulation) into the tail of an orbfe_tri_params"""
    P.camera_model1, P.camera_model2 = int(cameras["model1"]), int(cameras["model2"])
    for i in range(8):
        P.cam1[i] = float(cameras["cam1"][i])
        P.cam2[i] = float(cameras["cam2"][i])
    P.kb_precision = float(cameras.get("precision", 1e-6))
    for i, v in enumerate(np.asarray(cameras["R12"], np.float32).reshape(-1)):
        P.r12[i] = float(v)
    for i, v in enumerate(np.asarray(cameras["t12"], np.float32).reshape(-1)):
        P.t12[i] = float(v)
    for i, v in enumerate(np.asarray(cameras["levelSigma2_1"], np.float32).reshape(-1)):
        P.level_sigma2_1[i] = float(v)
    P.kf1_has_camera2 = int(cameras.get("kf1HasCamera2", 0))


class TrackParams(C.Structure):
    """orbfe_track_params: frame grid statics (src/Frame.cc:101-105) + the call parameters of SearchByProjection."""
    _fields_ = [("struct_size", C.c_int), ("grid_cols", C.c_int), ("grid_rows", C.c_int), ("min_x", C.c_float),
                ("min_y", C.c_float), ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("th", C.c_float),
                ("nn_ratio", C.c_float), ("far_points", C.c_int), ("th_far_points", C.c_float)]

    def __init__(self, *a, **k):
        super().__init__(*a, **k)
        self.struct_size = C.sizeof(TrackParams)


class FrameView(C.Structure):
    _fields_ = [("n", C.c_int), ("kp", C.c_void_p), ("desc", C.c_void_p), ("grid_cols", C.c_int),
                ("grid_rows", C.c_int), ("min_x", C.c_float), ("min_y", C.c_float),
                ("grid_inv_w", C.c_float), ("grid_inv_h", C.c_float), ("n_levels", C.c_int),
                ("scale_factors", C.c_void_p)]


# every symbol include/orbfe.h declares (checked by tests/test_abi.py)
SYMBOLS = [
    "orbfe_create", "orbfe_destroy", "orbfe_get_levels", "orbfe_get_scale_factor", "orbfe_get_scale_tables",
    "orbfe_get_level_info", "orbfe_max_keypoints", "orbfe_extract", "orbfe_extract_batch",
    "orbfe_extract_batch_device", "orbfe_get_pyramid_level", "orbfe_debug_get_candidates",
    "orbfe_set_stage_timing", "orbfe_get_stage_ms", "orbfe_stage_name", "orbfe_hamming",
    "orbfe_match_projection", "orbfe_match_projection_batch_device", "orbfe_match_bow", "orbfe_match_bow_rig", "orbfe_match_initialization", "orbfe_vocab_create", "orbfe_vocab_destroy", "orbfe_bow_transform",
    "orbfe_prep_create", "orbfe_prep_destroy", "orbfe_prepare_image", "orbfe_prepare_image_device", "orbfe_prepare_and_extract",
    "orbfe_project_map_points", "orbfe_project_map_points_device", "orbfe_fuse_search", "orbfe_fuse_search_right", "orbfe_fuse_search_sim3", "orbfe_search_by_sim3", "orbfe_match_projection_keyframe", "orbfe_match_triangulation", "orbfe_distinctive_descriptors", "orbfe_status_string", "orbfe_last_error", "orbfe_version",
    "orbfe_get_device_status", "orbfe_stream_create", "orbfe_stream_destroy", "orbfe_stream_submit", "orbfe_stream_collect",
    "orbfe_stream_collect_view", "orbfe_stream_in_flight", "orbfe_track_frame",
    "orbfe_keyframe_create", "orbfe_keyframe_destroy", "orbfe_keyframe_size", "orbfe_match_triangulation_batch",
    "orbfe_triangulation_select", "orbfe_map_create", "orbfe_map_destroy", "orbfe_map_update", "orbfe_stream_enable_track",
    "orbfe_stream_submit_track", "orbfe_stream_collect_track", "orbfe_track_frame_map", "orbfe_track_reference_keyframe", "orbfe_debug_graph_stats", "orbfe_set_graph_capture",
    "orbfe_debug_clock_probe", "orbfe_keyframe_set_grid", "orbfe_fuse_search_keyframe",
    "orbfe_init_frame_create", "orbfe_init_frame_destroy", "orbfe_init_frame_size", "orbfe_track_initialization",
    "orbfe_set_stream_priority",
]

_lib = None


def _share_hip_runtime_with_torch():
    """One HIP runtime per process.  PyTorch-ROCm wheels bundle their own libamdhip64.so (soname
    libamdhip64.so.7, looked up by file name through $ORIGIN); if liborbfe.so pulled in the system
    copy first, a later `import torch` would load a second HIP/HSA runtime that sees no GPU and
    whose streams are foreign to ours.  Loading torch's copy first makes liborbfe.so bind to it."""
    try:
        import importlib.util
        spec = importlib.util.find_spec("torch")
        if spec is None or not spec.submodule_search_locations:
            return
        cand = os.path.join(list(spec.submodule_search_locations)[0], "lib", "libamdhip64.so")
        if os.path.exists(cand):
            C.CDLL(cand, mode=C.RTLD_GLOBAL)
    except OSError:
        pass


def lib():
    """Load liborbfe.so; fail loudly when the HIP extension has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise RuntimeError("liborbfe.so not found at %s -- run __graft_entry__.build() (hipcc, gfx950); "
                           "there is no CPU fallback" % LIB_PATH)
    _share_hip_runtime_with_torch()
    L = C.CDLL(LIB_PATH)
    vp, ci, cf, sz = C.c_void_p, C.c_int, C.c_float, C.c_size_t
    L.orbfe_create.argtypes = [C.POINTER(Params), C.POINTER(vp)]
    L.orbfe_destroy.argtypes = [vp]
    L.orbfe_destroy.restype = None
    L.orbfe_get_levels.argtypes = [vp]
    L.orbfe_get_scale_factor.argtypes = [vp]
    L.orbfe_get_scale_factor.restype = cf
    L.orbfe_get_scale_tables.argtypes = [vp, vp, vp, vp, vp]
    L.orbfe_get_level_info.argtypes = [vp, vp, vp, vp]
    L.orbfe_max_keypoints.argtypes = [vp]
    L.orbfe_extract.argtypes = [vp, vp, ci, vp, vp, vp, vp]
    L.orbfe_extract_batch.argtypes = [vp, vp, ci, ci, vp, vp, vp, vp]
    L.orbfe_get_device_status.argtypes = [vp, vp]
    L.orbfe_debug_graph_stats.argtypes = [vp, vp, vp]
    L.orbfe_set_graph_capture.argtypes = [vp, ci]
    L.orbfe_set_stream_priority.argtypes = [vp, ci]
    L.orbfe_debug_clock_probe.argtypes = [vp, ci, vp, vp]
    L.orbfe_stream_create.argtypes = [vp, ci, ci, C.POINTER(vp)]
    L.orbfe_stream_destroy.argtypes = [vp]
    L.orbfe_stream_destroy.restype = None
    L.orbfe_stream_submit.argtypes = [vp, vp, ci, ci]
    L.orbfe_stream_collect.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orbfe_stream_collect_view.argtypes = [vp, vp, vp, vp, vp, vp]
    L.orbfe_stream_in_flight.argtypes = [vp]
    L.orbfe_map_create.argtypes = [vp, ci, C.POINTER(vp)]
    L.orbfe_map_destroy.argtypes = [vp]
    L.orbfe_map_destroy.restype = None
    L.orbfe_map_update.argtypes = [vp, vp, ci, vp, vp, vp]
    L.orbfe_stream_enable_track.argtypes = [vp, vp, ci]
    L.orbfe_stream_submit_track.argtypes = [vp, vp, ci, ci, C.POINTER(TrackParams), vp, ci, vp]
    L.orbfe_stream_collect_track.argtypes = [vp, vp, vp, vp, vp, vp, vp, vp]
    L.orbfe_extract_batch_device.argtypes = [vp, vp, sz, ci, ci, vp, vp, vp, vp, vp]
    L.orbfe_get_pyramid_level.argtypes = [vp, ci, ci, ci, vp, ci]
    L.orbfe_debug_get_candidates.argtypes = [vp, ci, ci, vp, ci, vp, vp]
    L.orbfe_set_stage_timing.argtypes = [vp, ci]
    L.orbfe_get_stage_ms.argtypes = [vp, vp, vp]
    L.orbfe_stage_name.argtypes = [ci]
    L.orbfe_stage_name.restype = C.c_char_p
    L.orbfe_hamming.argtypes = [vp, vp]
    L.orbfe_match_projection.argtypes = [vp, C.POINTER(FrameView), ci, vp, vp, vp, cf, ci, cf, cf, vp, vp]
    L.orbfe_match_projection_batch_device.argtypes = [vp, ci, vp, vp, vp, ci, ci, ci, cf, cf, cf, cf, ci, vp, vp, vp,
                                                      cf, ci, cf, cf, vp, vp, vp]
    L.orbfe_match_bow.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, cf, ci, vp, vp]
    L.orbfe_match_bow_rig.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, ci, vp, vp, ci, cf, ci, vp, vp]
    L.orbfe_match_initialization.argtypes = [vp, C.POINTER(FrameView), C.POINTER(FrameView), ci, cf, ci, vp, vp]
    L.orbfe_track_frame.argtypes = [vp, vp, ci, C.POINTER(Frustum), C.POINTER(TrackParams), ci, vp, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orbfe_track_frame_map.argtypes = [vp, vp, ci, C.POINTER(Frustum), C.POINTER(TrackParams), vp, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orbfe_track_reference_keyframe.argtypes = [vp, vp, ci, vp, ci, vp, vp, C.c_float, ci, vp, vp, vp, vp, vp, vp, vp, vp, vp]
    L.orbfe_init_frame_create.argtypes = [vp, ci, vp, vp, C.POINTER(vp)]
    L.orbfe_init_frame_destroy.argtypes = [vp]
    L.orbfe_init_frame_destroy.restype = None
    L.orbfe_init_frame_size.argtypes = [vp]
    L.orbfe_track_initialization.argtypes = [vp, vp, ci, vp, C.POINTER(TrackParams), ci, cf, ci, vp, vp, vp, vp, vp, vp]
    L.orbfe_project_map_points.argtypes = [vp, C.POINTER(Frustum), ci, vp, vp, vp]
    L.orbfe_project_map_points_device.argtypes = [vp, C.POINTER(Frustum), ci, vp, vp, vp, vp]
    L.orbfe_fuse_search.argtypes = [vp, C.POINTER(FrameView), vp, vp, C.POINTER(Frustum), cf, ci, vp, vp, vp, vp]
    L.orbfe_fuse_search_right.argtypes = [vp, C.POINTER(FrameView), ci, vp, vp, C.POINTER(Frustum), cf, ci, vp, vp, vp, vp]
    L.orbfe_prep_create.argtypes = [vp, ci, ci, vp, vp, ci, ci, C.POINTER(vp)]
    L.orbfe_prep_destroy.argtypes = [vp]
    L.orbfe_prep_destroy.restype = None
    L.orbfe_prepare_image.argtypes = [vp, vp, vp, ci, vp, ci]
    L.orbfe_prepare_image_device.argtypes = [vp, vp, vp, ci, vp, ci, vp]
    L.orbfe_prepare_and_extract.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp, ci]
    L.orbfe_fuse_search_sim3.argtypes = [vp, C.POINTER(FrameView), C.POINTER(Frustum), cf, ci, vp, vp, vp, vp]
    L.orbfe_search_by_sim3.argtypes = [vp, C.POINTER(FrameView), C.POINTER(FrameView), C.POINTER(Sim3View),
                                       C.POINTER(Sim3View), vp, vp, vp, vp, cf, vp, C.POINTER(ci)]
    L.orbfe_match_projection_keyframe.argtypes = [vp, C.POINTER(FrameView), C.POINTER(Frustum), ci, vp, vp, vp, vp, cf,
                                                  ci, vp, C.POINTER(ci)]
    L.orbfe_match_triangulation.argtypes = [vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, ci, vp, vp, vp, vp, vp, ci,
                                            C.POINTER(TriParams), vp, vp]
    L.orbfe_keyframe_create.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, C.POINTER(vp)]
    L.orbfe_keyframe_destroy.argtypes = [vp]
    L.orbfe_keyframe_destroy.restype = None
    L.orbfe_keyframe_size.argtypes = [vp]
    L.orbfe_keyframe_set_grid.argtypes = [vp, vp, ci, ci, cf, cf, cf, cf, vp, vp]
    L.orbfe_fuse_search_keyframe.argtypes = [vp, vp, vp, ci, vp, C.POINTER(Frustum), cf, vp, vp]
    L.orbfe_match_triangulation_batch.argtypes = [vp, vp, vp, ci, vp, vp, vp, vp, vp]
    L.orbfe_triangulation_select.argtypes = [ci, vp, vp, vp, ci, vp, vp]
    L.orbfe_distinctive_descriptors.argtypes = [vp, ci, vp, vp, vp, vp]
    L.orbfe_vocab_create.argtypes = [vp, ci, vp, vp, vp, vp, vp, ci, C.POINTER(vp)]
    L.orbfe_vocab_destroy.argtypes = [vp]
    L.orbfe_vocab_destroy.restype = None
    L.orbfe_bow_transform.argtypes = [vp, vp, vp, ci, ci, vp, vp, vp]
    L.orbfe_status_string.argtypes = [ci]
    L.orbfe_status_string.restype = C.c_char_p
    L.orbfe_last_error.argtypes = [vp]
    L.orbfe_last_error.restype = C.c_char_p
    L.orbfe_version.restype = C.c_char_p
    _lib = L
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class ORBextractor:
    """ORB_SLAM3::ORBextractor (include/ORBextractor.h:52-130) on one MI355X."""

    def __init__(self, nFeatures, nFastFeatures, scaleFactor, nlevels, iniThFAST, minThFAST, imageWidth,
                 imageHeight, device=0, max_batch=1):
        self.L = lib()
        self.h = C.c_void_p()
        prm = Params(nFeatures, nFastFeatures, scaleFactor, nlevels, iniThFAST, minThFAST, imageWidth,
                     imageHeight, device, max_batch)
        rc = self.L.orbfe_create(C.byref(prm), C.byref(self.h))
        if rc != 0:
            self.h = None
            raise OrbfeError(rc, "orbfe_create")
        self.nlevels, self.W, self.H, self.max_batch = nlevels, imageWidth, imageHeight, max_batch
        self.cap = self.L.orbfe_max_keypoints(self.h)
        n = nlevels
        self.mvScaleFactor = np.zeros(n, np.float32)
        self.mvInvScaleFactor = np.zeros(n, np.float32)
        self.mvLevelSigma2 = np.zeros(n, np.float32)
        self.mvInvLevelSigma2 = np.zeros(n, np.float32)
        self._chk(self.L.orbfe_get_scale_tables(self.h, _p(self.mvScaleFactor), _p(self.mvInvScaleFactor),
                                                _p(self.mvLevelSigma2), _p(self.mvInvLevelSigma2)), "scale_tables")
        self.mnFeaturesPerLevel = np.zeros(n, np.int32)
        self.levelW = np.zeros(n, np.int32)
        self.levelH = np.zeros(n, np.int32)
        self._chk(self.L.orbfe_get_level_info(self.h, _p(self.mnFeaturesPerLevel), _p(self.levelW), _p(self.levelH)),
                  "level_info")

    def _chk(self, rc, where):
        if rc != 0:
            raise OrbfeError(rc, where, self.L.orbfe_last_error(self.h).decode())

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_destroy(self.h)
            self.h = None

    __del__ = close

    # getters, include/ORBextractor.h:64-92
    def GetLevels(self):
        return self.L.orbfe_get_levels(self.h)

    def GetScaleFactor(self):
        return self.L.orbfe_get_scale_factor(self.h)

    def GetScaleFactors(self):
        return self.mvScaleFactor.copy()

    def GetInverseScaleFactors(self):
        return self.mvInvScaleFactor.copy()

    def GetScaleSigmaSquares(self):
        return self.mvLevelSigma2.copy()

    def GetInverseScaleSigmaSquares(self):
        return self.mvInvLevelSigma2.copy()

    def extractFeatures(self, im):
        """extractFeatures (include/ORBextractor.h:62): returns (keypoints, descriptors) or None."""
        im = np.asarray(im)
        assert im.dtype == np.uint8 and im.shape == (self.H, self.W) and im.strides[1] == 1
        kp = np.zeros(self.cap, KP_DTYPE)
        desc = np.zeros((self.cap, 32), np.uint8)
        n = C.c_int()
        self.last_per_level = np.zeros(self.nlevels, np.int32)
        self._chk(self.L.orbfe_extract(self.h, _p(im), im.strides[0], _p(kp), _p(desc), C.byref(n),
                                       _p(self.last_per_level)), "orbfe_extract")
        if n.value == 0:
            return None
        return kp[:n.value].copy(), desc[:n.value].copy()

    def extract_batch(self, ims):
        """Host-pointer batched mode: list/array of frames -> list of (kp, desc, per_level)."""
        ims = [np.ascontiguousarray(im, np.uint8) for im in ims]
        B = len(ims)
        assert 1 <= B <= self.max_batch
        ptrs = (C.c_void_p * B)(*[im.ctypes.data for im in ims])
        kp = np.zeros((B, self.cap), KP_DTYPE)
        desc = np.zeros((B, self.cap, 32), np.uint8)
        n = np.zeros(B, np.int32)
        per = np.zeros((B, self.nlevels), np.int32)
        self._chk(self.L.orbfe_extract_batch(self.h, ptrs, self.W, B, _p(kp), _p(desc), _p(n), _p(per)),
                  "orbfe_extract_batch")
        return [(kp[b, :n[b]].copy(), desc[b, :n[b]].copy(), per[b].copy()) for b in range(B)]

    def extract_batch_device(self, d_gray_ptr, frame_stride, pitch, batch, d_kp_ptr, d_desc_ptr, d_n_ptr,
                             d_per_ptr=None, stream=None):
        """Everything resident in HBM (raw device pointers as ints); asynchronous on `stream`."""
        self._chk(self.L.orbfe_extract_batch_device(self.h, d_gray_ptr, frame_stride, pitch, batch, d_kp_ptr,
                                                    d_desc_ptr, d_n_ptr, d_per_ptr, stream),
                  "orbfe_extract_batch_device")

    def device_status(self):
        """Guard flags of the last extract call on any stream (waits for it); raises when one is set."""
        flags = C.c_uint()
        self._chk(self.L.orbfe_get_device_status(self.h, C.byref(flags)), "orbfe_get_device_status")
        return flags.value

    def stream(self, slots=3, slot_frames=None):
        """Pipelined host-pointer extraction (orbfe_stream_*): see ExtractStream."""
        return ExtractStream(self, slots, slot_frames or self.max_batch)

    # mvImagePyramid / mvBlurredImagePyramid (include/ORBextractor.h:94-95)
    def pyramid_level(self, level, blurred=False, frame=0):
        w, h = int(self.levelW[level]), int(self.levelH[level])
        out = np.zeros((h, w), np.uint8)
        self._chk(self.L.orbfe_get_pyramid_level(self.h, frame, level, int(blurred), _p(out), w), "pyramid_level")
        return out

    def debug_candidates(self, level, frame=0):
        cap = ((int(self.levelW[level]) + 1) // 2) * ((int(self.levelH[level]) + 1) // 2)
        packed = np.zeros(cap, np.uint32)
        n = C.c_int()
        cnt = np.zeros(4, np.int32)
        self._chk(self.L.orbfe_debug_get_candidates(self.h, frame, level, _p(packed), cap, C.byref(n), _p(cnt)),
                  "debug_candidates")
        return packed[:n.value].copy(), cnt

    def set_graph_capture(self, on):
        self._chk(self.L.orbfe_set_graph_capture(self.h, int(bool(on))), "orbfe_set_graph_capture")

    def set_stream_priority(self, high):
        """orbfe_set_stream_priority: the handle's own stream at the device's highest (True) / lowest (False) priority"""
        self._chk(self.L.orbfe_set_stream_priority(self.h, int(bool(high))), "orbfe_set_stream_priority")

    def graph_stats(self):
        """(graphs captured, captures that failed and fell back to plain launches) of this handle"""
        a, b = C.c_int(), C.c_int()
        self._chk(self.L.orbfe_debug_graph_stats(self.h, C.byref(a), C.byref(b)), "orbfe_debug_graph_stats")
        return a.value, b.value

    def clock_probe(self, d_out_ptr, spin_us=20, stream=None):
        """orbfe_debug_clock_probe: asynchronous; d_out_ptr -> two device uint64 {shader cycles, 100 MHz ticks}"""
        self._chk(self.L.orbfe_debug_clock_probe(self.h, int(spin_us), d_out_ptr, stream), "orbfe_debug_clock_probe")

    def set_stage_timing(self, on):
        self._chk(self.L.orbfe_set_stage_timing(self.h, int(on)), "set_stage_timing")

    def stage_ms(self):
        ms = np.zeros(NUM_STAGES, np.float32)
        n = C.c_int()
        self._chk(self.L.orbfe_get_stage_ms(self.h, _p(ms), C.byref(n)), "get_stage_ms")
        names = [self.L.orbfe_stage_name(i).decode() for i in range(NUM_STAGES)]
        return dict(zip(names, ms.tolist())), n.value


def make_frame_view(kp, desc, gridCols, gridRows, minX, minY, maxX, maxY, scaleFactors):
    """Frame grid statics exactly as src/Frame.cc:101-105 derives them."""
    kp = np.ascontiguousarray(kp, KP_DTYPE)
    desc = np.ascontiguousarray(desc, np.uint8)
    sf = np.ascontiguousarray(scaleFactors, np.float32)
    invw = np.float32(gridCols) / np.float32(np.float32(maxX) - np.float32(minX))
    invh = np.float32(gridRows) / np.float32(np.float32(maxY) - np.float32(minY))
    fv = FrameView(len(kp), kp.ctypes.data, desc.ctypes.data, gridCols, gridRows, minX, minY, float(invw),
                   float(invh), len(sf), sf.ctypes.data)
    fv._keep = (kp, desc, sf)
    return fv


class MapPoints:
    """orbfe_map: map points resident in HBM (world position, distance range, isBad, Observations, descriptor), indexed
    by a caller-chosen id in [0, capacity); frames name their local map points by id."""

    def __init__(self, extractor, capacity):
        self.e, self.L = extractor, extractor.L
        self.capacity = int(capacity)
        self.h = C.c_void_p()
        extractor._chk(self.L.orbfe_map_create(extractor.h, self.capacity, C.byref(self.h)), "orbfe_map_create")

    def update(self, ids, points, desc):
        ids = np.ascontiguousarray(ids, np.int32)
        points = np.ascontiguousarray(points, WP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        assert len(ids) == len(points) == len(desc)
        self.e._chk(self.L.orbfe_map_update(self.e.h, self.h, len(ids), _p(ids), _p(points), _p(desc)), "orbfe_map_update")

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_map_destroy(self.h)
            self.h = None

    __del__ = close


class ExtractStream:
    """orbfe_stream_*: a ring of pinned + device slots; submit() enqueues upload, kernels and download of up to
    `slot_frames` frames without waiting, collect() waits for the oldest submission only."""

    def __init__(self, ex, slots, slot_frames):
        self.ex, self.L = ex, ex.L
        self.slot_frames = slot_frames
        h = C.c_void_p()
        ex._chk(self.L.orbfe_stream_create(ex.h, slots, slot_frames, C.byref(h)), "orbfe_stream_create")
        self.h = h
        cap, nl = ex.cap, ex.nlevels
        self._kp = np.zeros((slot_frames, cap), KP_DTYPE)
        self._desc = np.zeros((slot_frames, cap, 32), np.uint8)
        self._n = np.zeros(slot_frames, np.int32)
        self._per = np.zeros((slot_frames, nl), np.int32)

    def close(self):
        if self.h:
            self.L.orbfe_stream_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def in_flight(self):
        return self.L.orbfe_stream_in_flight(self.h)

    def submit_ptrs(self, ptrs, pitch, n):
        """ptrs: ctypes array of `n` host frame addresses; returns False when every slot is in flight."""
        rc = self.L.orbfe_stream_submit(self.h, ptrs, pitch, n)
        if rc == ERR_BUSY:
            return False
        self.ex._chk(rc, "orbfe_stream_submit")
        return True

    def submit(self, frames, pitch=None):
        """frames: [n][H][W] uint8 array (numpy, or anything exposing ctypes data) in host memory."""
        n = len(frames)
        base = frames.ctypes.data
        stride = frames.strides[0]
        ptrs = (C.c_void_p * n)(*[base + b * stride for b in range(n)])
        return self.submit_ptrs(ptrs, pitch or frames.strides[1], n)

    def collect_raw(self):
        """Results of the oldest submission in the stream's own arrays (overwritten by the next collect)."""
        nf = C.c_int()
        self.ex._chk(self.L.orbfe_stream_collect(self.h, _p(self._kp), _p(self._desc), _p(self._n), _p(self._per), C.byref(nf)),
                     "orbfe_stream_collect")
        return nf.value, self._kp, self._desc, self._n, self._per

    def collect(self):
        nf, kp, desc, n, per = self.collect_raw()
        return [(kp[b, :n[b]].copy(), desc[b, :n[b]].copy(), per[b].copy()) for b in range(nf)]

    # ---- extract-and-match submissions (orbfe_stream_enable_track / submit_track / collect_track) ----
    def enable_track(self, map_points, max_points, gridCols, gridRows, minX, minY, maxX, maxY):
        self.ex._chk(self.L.orbfe_stream_enable_track(self.h, map_points.h, int(max_points)), "orbfe_stream_enable_track")
        self._map = map_points
        self._grid = (int(gridCols), int(gridRows), float(minX), float(minY),
                      float(np.float32(gridCols) / np.float32(np.float32(maxX) - np.float32(minX))),
                      float(np.float32(gridRows) / np.float32(np.float32(maxY) - np.float32(minY))))
        self._match = np.full((self.slot_frames, self.ex.cap), -1, np.int32)
        self._nmatch = np.zeros(self.slot_frames, np.int32)

    def submit_track(self, frames, frusta, ids, th, nnRatio, bFarPoints=False, thFarPoints=0.0, pitch=None):
        """frames as submit(); frusta: ctypes array (Frustum * n) or list of Frustum; ids: [n][n_points] int32 (id >= 0:
        entry of the resident map, ~id: the entry with mnLastFrameSeen == current frame, i.e. skipped)."""
        n = len(frames)
        if not isinstance(frusta, C.Array):
            frusta = (Frustum * n)(*frusta)
        ids = np.ascontiguousarray(ids, np.int32)
        assert ids.ndim == 2 and ids.shape[0] == n
        tp = TrackParams()
        (tp.grid_cols, tp.grid_rows, tp.min_x, tp.min_y, tp.grid_inv_w, tp.grid_inv_h) = self._grid
        tp.th, tp.nn_ratio, tp.far_points, tp.th_far_points = th, nnRatio, int(bFarPoints), thFarPoints
        base, stride = frames.ctypes.data, frames.strides[0]
        ptrs = (C.c_void_p * n)(*[base + b * stride for b in range(n)])
        rc = self.L.orbfe_stream_submit_track(self.h, ptrs, pitch or frames.strides[1], n, C.byref(tp), frusta, ids.shape[1], _p(ids))
        if rc == ERR_BUSY:
            return False
        self.ex._chk(rc, "orbfe_stream_submit_track")
        return True

    def collect_track_raw(self):
        nf = C.c_int()
        self.ex._chk(self.L.orbfe_stream_collect_track(self.h, _p(self._kp), _p(self._desc), _p(self._n), _p(self._per), _p(self._match),
                                                       _p(self._nmatch), C.byref(nf)), "orbfe_stream_collect_track")
        return nf.value, self._kp, self._desc, self._n, self._per, self._match, self._nmatch

    def collect_track(self):
        nf, kp, desc, n, per, match, nm = self.collect_track_raw()
        return [(kp[b, :n[b]].copy(), desc[b, :n[b]].copy(), per[b].copy(), match[b, :n[b]].copy(), int(nm[b])) for b in range(nf)]

    def collect_view(self):
        """orbfe_stream_collect_view: the oldest submission's results as numpy VIEWS of the slot's pinned block (no
        copy); they stay valid until the submission that reuses the slot, `slots` submissions later.
        Returns (n_frames, kp[slot_frames][cap], desc[slot_frames][cap][32], n[slot_frames], per[slot_frames][levels])."""
        kp, desc, n, per = C.c_void_p(), C.c_void_p(), C.c_void_p(), C.c_void_p()
        nf = C.c_int()
        self.ex._chk(self.L.orbfe_stream_collect_view(self.h, C.byref(kp), C.byref(desc), C.byref(n), C.byref(per), C.byref(nf)),
                     "orbfe_stream_collect_view")
        cap, nl, sf = self.ex.cap, self.ex.nlevels, self.slot_frames

        def view(ptr, nbytes, dtype, shape):
            return np.frombuffer((C.c_uint8 * nbytes).from_address(ptr.value), dtype=dtype).reshape(shape)

        return (nf.value, view(kp, sf * cap * 24, KP_DTYPE, (sf, cap)), view(desc, sf * cap * 32, np.uint8, (sf, cap, 32)),
                view(n, sf * 4, np.int32, (sf,)), view(per, sf * nl * 4, np.int32, (sf, nl)))


class ORBmatcher:
    """ORB_SLAM3::ORBmatcher (include/ORBmatcher.h:36-84); statics bound to one extractor handle."""
    TH_LOW = 30
    TH_HIGH = 100
    HISTO_LENGTH = 30

    def __init__(self, extractor):
        self.e = extractor
        self.L = extractor.L

    @staticmethod
    def DescriptorDistance(a, b):
        a = np.ascontiguousarray(a, np.uint8)
        b = np.ascontiguousarray(b, np.uint8)
        return lib().orbfe_hamming(_p(a), _p(b))

    def SearchByProjection(self, fv, mps, mpDesc, th, bFarPoints, thFarPoints, nnRatio, initObs=None):
        mps = np.ascontiguousarray(mps, MP_DTYPE)
        mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
        io = None if initObs is None else np.ascontiguousarray(initObs, np.int32)
        out = np.full(max(1, fv.n), -1, np.int32)
        n = C.c_int()
        self.e._chk(self.L.orbfe_match_projection(self.e.h, C.byref(fv), len(mps), _p(mps), _p(mpDesc), _p(io), th,
                                                  int(bFarPoints), thFarPoints, nnRatio, _p(out), C.byref(n)),
                    "orbfe_match_projection")
        return n.value, out[:fv.n].copy()

    def SearchByProjection_batch_device(self, batch, d_kp, d_desc, d_n, kp_stride, gridCols, gridRows, minX, minY,
                                        maxX, maxY, M, d_mps, d_mp_desc, d_init_obs, th, nnRatio, d_match_out,
                                        d_n_matches, bFarPoints=False, thFarPoints=0.0, stream=None):
        """Batched HBM-resident SearchByProjection; all d_* are raw device pointers (ints)."""
        invw = float(np.float32(gridCols) / np.float32(np.float32(maxX) - np.float32(minX)))
        invh = float(np.float32(gridRows) / np.float32(np.float32(maxY) - np.float32(minY)))
        self.e._chk(self.L.orbfe_match_projection_batch_device(
            self.e.h, batch, d_kp, d_desc, d_n, kp_stride, gridCols, gridRows, minX, minY, invw, invh, M, d_mps,
            d_mp_desc, d_init_obs, th, int(bFarPoints), thFarPoints, nnRatio, d_match_out, d_n_matches, stream),
            "orbfe_match_projection_batch_device")

    def isInFrustum_batch(self, frustum, points):
        """The isInFrustum loop of Tracking::SearchLocalPoints (src/Tracking.cc:1059-1077) for all points at once:
        returns (map point records for SearchByProjection, mTrackProjXR)."""
        points = np.ascontiguousarray(points, WP_DTYPE)
        n = len(points)
        out = np.zeros(max(n, 1), MP_DTYPE)
        xr = np.zeros(max(n, 1), np.float32)
        self.e._chk(self.L.orbfe_project_map_points(self.e.h, C.byref(frustum), n, _p(points), _p(out), _p(xr)),
                    "orbfe_project_map_points")
        return out[:n], xr[:n]

    def SearchForTriangulation(self, off1, idx1, off2, idx2, kp1, desc1, hasMP1, stereo1, kp2, desc2, hasMP2, stereo2,
                               scaleFactors2, F12, ep, bOnlyStereo=False, bCoarse=False, checkOrientation=True, cameras=None):
        """ORBmatcher::SearchForTriangulation (src/ORBmatcher.cc:441-676): returns (nmatches, vMatches12).
        cameras (optional) = dict(model1, model2, cam1[8], cam2[8], precision, R12[3x3], t12[3], levelSigma2_1,
        kf1HasCamera2): the camera models of the two key frames; model1 == CAMERA_KANNALA_BRANDT8 selects
        KannalaBrandt8::epipolarConstrain (src/CameraModels/KannalaBrandt8.cpp:216-220)."""
        a32 = lambda v: np.ascontiguousarray(v, np.int32)
        u8 = lambda v: None if v is None else np.ascontiguousarray(v, np.uint8)
        off1, idx1, off2, idx2 = a32(off1), a32(idx1), a32(off2), a32(idx2)
        kp1, kp2 = np.ascontiguousarray(kp1, KP_DTYPE), np.ascontiguousarray(kp2, KP_DTYPE)
        desc1, desc2 = u8(desc1), u8(desc2)
        h1, h2, s1, s2 = u8(hasMP1), u8(hasMP2), u8(stereo1), u8(stereo2)
        sf = np.ascontiguousarray(scaleFactors2, np.float32)
        P = TriParams()
        for i, v in enumerate(np.asarray(F12, np.float32).reshape(-1)):
            P.f12[i] = float(v)
        P.ep_x, P.ep_y = float(ep[0]), float(ep[1])
        P.only_stereo, P.coarse, P.check_orientation = int(bOnlyStereo), int(bCoarse), int(checkOrientation)
        if cameras is not None:
            fill_tri_cameras(P, cameras)
        out = np.full(max(len(kp1), 1), -1, np.int32)
        n = C.c_int()
        self.e._chk(self.L.orbfe_match_triangulation(self.e.h, len(off1) - 1, _p(off1), _p(idx1), _p(off2), _p(idx2), len(kp1),
                                                     _p(kp1), _p(desc1), _p(h1), _p(s1), len(kp2), _p(kp2), _p(desc2), _p(h2),
                                                     _p(s2), _p(sf), len(sf), C.byref(P), _p(out), C.byref(n)),
                    "orbfe_match_triangulation")
        return n.value, out[:len(kp1)].copy()

    def ComputeDistinctiveDescriptors(self, setOff, desc):
        """MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:343-416) for a batch of descriptor sets (CSR):
        returns (index of the representative inside each set, its median distance)."""
        setOff = np.ascontiguousarray(setOff, np.int32)
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(setOff) - 1
        bi = np.zeros(max(n, 1), np.int32)
        bm = np.zeros(max(n, 1), np.int32)
        self.e._chk(self.L.orbfe_distinctive_descriptors(self.e.h, n, _p(setOff), _p(desc), _p(bi), _p(bm)),
                    "orbfe_distinctive_descriptors")
        return bi[:n], bm[:n]

    def Fuse_search(self, kf_view, invLevelSigma2, uRight, frustum, th, points, mpDesc):
        """The search part of ORBmatcher::Fuse(pKF, vpMapPoints, th) (src/ORBmatcher.cc:678-836): (bestIdx, bestDist)
        per map point; the caller applies bestDist <= TH_LOW and the graph edits."""
        points = np.ascontiguousarray(points, WP_DTYPE)
        mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
        is2 = np.ascontiguousarray(invLevelSigma2, np.float32)
        ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        M = len(points)
        bi = np.zeros(max(M, 1), np.int32)
        bd = np.zeros(max(M, 1), np.int32)
        self.e._chk(self.L.orbfe_fuse_search(self.e.h, C.byref(kf_view), _p(is2), _p(ur), C.byref(frustum), th, M,
                                             _p(points), _p(mpDesc), _p(bi), _p(bd)), "orbfe_fuse_search")
        return bi[:M], bd[:M]

    def Fuse_search_keyframe(self, kf, map_points, ids, frustum, th):
        """orbfe_fuse_search_keyframe: the search part of Fuse against a RESIDENT KeyFrame (set_grid done) with the map points
        named by id out of a resident MapPoints table (id >= 0; ~id = "!pMP || pMP->IsInKeyFrame(pKF)" for this call; outside
        the map = no point) -> (bestIdx, bestDist) per id."""
        ids = np.ascontiguousarray(ids, np.int32)
        M = len(ids)
        bi = np.zeros(max(M, 1), np.int32)
        bd = np.zeros(max(M, 1), np.int32)
        self.e._chk(self.L.orbfe_fuse_search_keyframe(self.e.h, kf.h, map_points.h, M, _p(ids), C.byref(frustum), th, _p(bi), _p(bd)),
                    "orbfe_fuse_search_keyframe")
        return bi[:M], bd[:M]

    def Fuse_search_right(self, kf_left_view, nRight, invLevelSigma2, uRight, frustum, th, points, mpDesc):
        """Fuse(pKF, vpMapPoints, th, bRight = true) (src/ORBmatcher.cc:684-688,:820): kf_left_view describes the NLeft left
        features with desc = all NLeft + nRight rows of mDescriptors; frustum = right pose / mpCamera2; returned indices are
        idx + NLeft."""
        points = np.ascontiguousarray(points, WP_DTYPE)
        mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
        is2 = np.ascontiguousarray(invLevelSigma2, np.float32)
        ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        M = len(points)
        bi = np.zeros(max(M, 1), np.int32)
        bd = np.zeros(max(M, 1), np.int32)
        self.e._chk(self.L.orbfe_fuse_search_right(self.e.h, C.byref(kf_left_view), int(nRight), _p(is2), _p(ur), C.byref(frustum),
                                                   th, M, _p(points), _p(mpDesc), _p(bi), _p(bd)), "orbfe_fuse_search_right")
        return bi[:M], bd[:M]

    def Fuse_search_sim3(self, kf_view, frustum, th, points, mpDesc):
        """The search part of ORBmatcher::Fuse(pKF, Scw, vpPoints, th, vpReplacePoint) (src/ORBmatcher.cc:864-975)."""
        points = np.ascontiguousarray(points, WP_DTYPE)
        mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
        M = len(points)
        bi = np.zeros(max(M, 1), np.int32)
        bd = np.zeros(max(M, 1), np.int32)
        self.e._chk(self.L.orbfe_fuse_search_sim3(self.e.h, C.byref(kf_view), C.byref(frustum), th, M, _p(points),
                                                  _p(mpDesc), _p(bi), _p(bd)), "orbfe_fuse_search_sim3")
        return bi[:M], bd[:M]

    def SearchBySim3(self, kf1_view, kf2_view, dir12, dir21, mp1, mpDesc1, mp2, mpDesc2, th):
        """include/ORBmatcher.h:63 -> (nFound, match12) with match12[i1] = feature of key frame 2 or -1."""
        mp1 = np.ascontiguousarray(mp1, WP_DTYPE)
        mp2 = np.ascontiguousarray(mp2, WP_DTYPE)
        assert len(mp1) == kf1_view.n and len(mp2) == kf2_view.n
        mpDesc1 = np.ascontiguousarray(mpDesc1, np.uint8)
        mpDesc2 = np.ascontiguousarray(mpDesc2, np.uint8)
        out = np.full(max(1, kf1_view.n), -1, np.int32)
        n = C.c_int()
        self.e._chk(self.L.orbfe_search_by_sim3(self.e.h, C.byref(kf1_view), C.byref(kf2_view), C.byref(dir12),
                                                C.byref(dir21), _p(mp1), _p(mpDesc1), _p(mp2), _p(mpDesc2), th, _p(out),
                                                C.byref(n)), "orbfe_search_by_sim3")
        return n.value, out[:kf1_view.n]

    def SearchByProjection_keyframe(self, fv, frustum, points, mpDesc, kfAngle, frameHasMP, th, checkOrientation=True):
        """include/ORBmatcher.h:48 (relocalisation overload) -> (nmatches, match) with match[i2] = key-frame feature
        whose map point lands in CurrentFrame->mvpMapPoints[i2], or -1."""
        points = np.ascontiguousarray(points, WP_DTYPE)
        mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
        ang = None if kfAngle is None else np.ascontiguousarray(kfAngle, np.float32)
        has = None if frameHasMP is None else np.ascontiguousarray(frameHasMP, np.uint8)
        out = np.full(max(1, fv.n), -1, np.int32)
        n = C.c_int()
        self.e._chk(self.L.orbfe_match_projection_keyframe(self.e.h, C.byref(fv), C.byref(frustum), len(points),
                                                           _p(points), _p(mpDesc), _p(ang), _p(has), th,
                                                           int(bool(checkOrientation)), _p(out), C.byref(n)),
                    "orbfe_match_projection_keyframe")
        return n.value, out[:fv.n]

    def isInFrustum_batch_device(self, frustum, n, d_points, d_out, d_proj_xr=None, stream=None):
        self.e._chk(self.L.orbfe_project_map_points_device(self.e.h, C.byref(frustum), n, d_points, d_out, d_proj_xr,
                                                           stream), "orbfe_project_map_points_device")

    def SearchForInitialization(self, fv1, fv2, windowSize, nnRatio, checkOrientation=True):
        """include/ORBmatcher.h:58 -> (nmatches, vnMatches12)."""
        out = np.full(max(1, fv1.n), -1, np.int32)
        n = C.c_int()
        self.e._chk(self.L.orbfe_match_initialization(self.e.h, C.byref(fv1), C.byref(fv2), int(windowSize), nnRatio,
                                                      int(checkOrientation), _p(out), C.byref(n)),
                    "orbfe_match_initialization")
        return n.value, out[:fv1.n].copy()

    def SearchByBoW(self, kfOff, kfIdx, fOff, fIdx, kfDesc, kfAngle, kfHasMP, fDesc, fAngle, nnRatio,
                    checkOrientation=True, nLeft=-1):
        a32 = lambda v: np.ascontiguousarray(v, np.int32)
        kfOff, kfIdx, fOff, fIdx = a32(kfOff), a32(kfIdx), a32(fOff), a32(fIdx)
        kfDesc = np.ascontiguousarray(kfDesc, np.uint8)
        fDesc = np.ascontiguousarray(fDesc, np.uint8)
        kfAngle = np.ascontiguousarray(kfAngle, np.float32)
        fAngle = np.ascontiguousarray(fAngle, np.float32)
        kfHasMP = np.ascontiguousarray(kfHasMP, np.uint8)
        out = np.full(max(1, len(fDesc)), -1, np.int32)
        n = C.c_int()
        self.e._chk(self.L.orbfe_match_bow_rig(self.e.h, len(kfOff) - 1, _p(kfOff), _p(kfIdx), _p(fOff), _p(fIdx),
                                               len(kfDesc), _p(kfDesc), _p(kfAngle), _p(kfHasMP), len(fDesc), _p(fDesc),
                                               _p(fAngle), int(nLeft), nnRatio, int(checkOrientation), _p(out), C.byref(n)),
                    "orbfe_match_bow_rig")
        return n.value, out[:len(fDesc)].copy()


def tri_params(F12, ep, bOnlyStereo=False, bCoarse=False, checkOrientation=True, cameras=None):
    """orbfe_tri_params of one key-frame pair (see ORBmatcher.SearchForTriangulation)"""
    P = TriParams()
    for i, v in enumerate(np.asarray(F12, np.float32).reshape(-1)):
        P.f12[i] = float(v)
    P.ep_x, P.ep_y = float(ep[0]), float(ep[1])
    P.only_stereo, P.coarse, P.check_orientation = int(bOnlyStereo), int(bCoarse), int(checkOrientation)
    if cameras is not None:
        fill_tri_cameras(P, cameras)
    return P


class KeyFrame:
    """A key frame resident in HBM (orbfe_keyframe_*): mvKeysUn, mDescriptors, mFeatVec (as the node of every feature, -1 =
    none), mvuRight >= 0, mvScaleFactors -- what the key-frame matchers read and what never changes after construction."""

    def __init__(self, extractor, kp, desc, nodeId, scaleFactors, stereo=None):
        self.e, self.L = extractor, extractor.L
        kp = np.ascontiguousarray(kp, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        node = np.ascontiguousarray(nodeId, np.int32)
        sf = np.ascontiguousarray(scaleFactors, np.float32)
        st = None if stereo is None else np.ascontiguousarray(stereo, np.uint8)
        assert len(desc) == len(kp) == len(node)
        self.n = len(kp)
        self.h = C.c_void_p()
        extractor._chk(self.L.orbfe_keyframe_create(extractor.h, self.n, _p(kp), _p(desc), _p(node), _p(st), _p(sf), len(sf),
                                                    C.byref(self.h)), "orbfe_keyframe_create")

    def set_grid(self, gridCols, gridRows, minX, minY, maxX, maxY, invLevelSigma2, uRight=None):
        """orbfe_keyframe_set_grid: mGrid's geometry (as make_frame_view derives it, src/Frame.cc:101-105), mvInvLevelSigma2 and
        mvuRight -- the per-level cell tables are built once and stay with the key frame (Fuse_search_keyframe)."""
        invw = float(np.float32(gridCols) / np.float32(np.float32(maxX) - np.float32(minX)))
        invh = float(np.float32(gridRows) / np.float32(np.float32(maxY) - np.float32(minY)))
        is2 = np.ascontiguousarray(invLevelSigma2, np.float32)
        ur = None if uRight is None else np.ascontiguousarray(uRight, np.float32)
        assert ur is None or len(ur) == self.n
        self.e._chk(self.L.orbfe_keyframe_set_grid(self.e.h, self.h, int(gridCols), int(gridRows), float(minX), float(minY), invw, invh,
                                                   _p(is2), _p(ur)), "orbfe_keyframe_set_grid")

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_keyframe_destroy(self.h)
            self.h = None

    __del__ = close


class InitialFrame:
    """mInitialFrame of Tracking::MonocularInitialization (src/Tracking.cc:569-586) resident in HBM (orbfe_init_frame_*): the
    keypoints and descriptors every following frame is matched against while the map is being initialised."""

    def __init__(self, extractor, kp, desc):
        self.e, self.L = extractor, extractor.L
        kp = np.ascontiguousarray(kp, KP_DTYPE)
        desc = np.ascontiguousarray(desc, np.uint8)
        assert len(kp) == len(desc)
        self.n = len(kp)
        self.h = C.c_void_p()
        extractor._chk(self.L.orbfe_init_frame_create(extractor.h, self.n, _p(kp), _p(desc), C.byref(self.h)), "orbfe_init_frame_create")

    def close(self):
        if getattr(self, "h", None):
            self.L.orbfe_init_frame_destroy(self.h)
            self.h = None

    __del__ = close


def SearchForTriangulation_batch(extractor, kf1, hasMP1, kf2s, hasMP2s, params):
    """orbfe_match_triangulation_batch: key frame kf1 against the neighbours kf2s in ONE launch -> (raw_match12 [K][n1],
    raw_bin [K][n1]); walk the neighbours in order with triangulation_select and the flags as they stand then."""
    K = len(kf2s)
    h1 = np.ascontiguousarray(hasMP1, np.uint8)
    h2 = [np.ascontiguousarray(v, np.uint8) for v in hasMP2s]
    kfp = (C.c_void_p * max(K, 1))(*[k.h.value for k in kf2s])
    h2p = (C.c_void_p * max(K, 1))(*[v.ctypes.data for v in h2])
    P = (TriParams * max(K, 1))(*params)
    raw = np.full((max(K, 1), max(kf1.n, 1)), -1, np.int32)
    rbin = np.zeros((max(K, 1), max(kf1.n, 1)), np.uint8)
    extractor._chk(extractor.L.orbfe_match_triangulation_batch(extractor.h, kf1.h, _p(h1), K, kfp, h2p, P, _p(raw), _p(rbin)),
                   "orbfe_match_triangulation_batch")
    return raw[:K, :kf1.n], rbin[:K, :kf1.n]


def triangulation_select(raw_match12, raw_bin, hasMP1_now, checkOrientation=True):
    """orbfe_triangulation_select (host-only): one neighbour's (nmatches, vMatches12) from its raw batch results and the
    CURRENT has-map-point flags of key frame 1."""
    raw = np.ascontiguousarray(raw_match12, np.int32)
    rb = np.ascontiguousarray(raw_bin, np.uint8)
    now = np.ascontiguousarray(hasMP1_now, np.uint8)
    out = np.full(max(len(raw), 1), -1, np.int32)
    n = C.c_int()
    rc = lib().orbfe_triangulation_select(len(raw), _p(raw), _p(rb), _p(now), int(checkOrientation), _p(out), C.byref(n))
    if rc != 0:
        raise OrbfeError(rc, "orbfe_triangulation_select")
    return n.value, out[:len(raw)]


class FrameTracker:
    """The tracking thread's per-frame chain of this fork once the IMU is initialised -- Frame::Frame -> ExtractORB
    (src/Tracking.cc:152-173, src/Frame.cc:178-189), the isInFrustum loop of Tracking::SearchLocalPoints (:1059-1077) and
    ORBmatcher::SearchByProjection(mCurrentFrame, mvpLocalMapPoints, ...) (:1108-1115) -- as ONE call / one captured
    hipGraph (orbfe_track_frame).  The grid statics are the frame's (src/Frame.cc:101-105)."""

    def __init__(self, extractor, gridCols, gridRows, minX, minY, maxX, maxY):
        self.e, self.L = extractor, extractor.L
        self.grid = (int(gridCols), int(gridRows), float(minX), float(minY),
                     float(np.float32(gridCols) / np.float32(np.float32(maxX) - np.float32(minX))),
                     float(np.float32(gridRows) / np.float32(np.float32(maxY) - np.float32(minY))))

    def TrackFrame(self, im, frustum, points, mpDesc, th, nnRatio, bFarPoints=False, thFarPoints=0.0):
        """-> dict(kp, desc, per_level, mps, proj_xr, match, nmatches); kp is empty when the frame has no keypoints
        (the reference returns early, src/Tracking.cc:158-159)."""
        e = self.e
        im = np.asarray(im)
        assert im.dtype == np.uint8 and im.shape == (e.H, e.W) and im.strides[1] == 1
        points = np.ascontiguousarray(points, WP_DTYPE)
        mpDesc = np.ascontiguousarray(mpDesc, np.uint8)
        M = len(points)
        assert mpDesc.shape == (M, 32) or M == 0
        tp = TrackParams()
        (tp.grid_cols, tp.grid_rows, tp.min_x, tp.min_y, tp.grid_inv_w, tp.grid_inv_h) = self.grid
        tp.th, tp.nn_ratio, tp.far_points, tp.th_far_points = th, nnRatio, int(bFarPoints), thFarPoints
        kp = np.zeros(e.cap, KP_DTYPE)
        desc = np.zeros((e.cap, 32), np.uint8)
        per = np.zeros(e.nlevels, np.int32)
        mps = np.zeros(max(M, 1), MP_DTYPE)
        xr = np.zeros(max(M, 1), np.float32)
        match = np.full(e.cap, -1, np.int32)
        n, nm = C.c_int(), C.c_int()
        e._chk(self.L.orbfe_track_frame(e.h, _p(im), im.strides[0], C.byref(frustum), C.byref(tp), M, _p(points), _p(mpDesc),
                                        _p(kp), _p(desc), C.byref(n), _p(per), _p(mps), _p(xr), _p(match), C.byref(nm)),
               "orbfe_track_frame")
        k = n.value
        return dict(kp=kp[:k], desc=desc[:k], per_level=per, mps=mps[:M], proj_xr=xr[:M], match=match[:k], nmatches=nm.value)

    def TrackFrameMap(self, im, frustum, map_points, ids, th, nnRatio, bFarPoints=False, thFarPoints=0.0):
        """orbfe_track_frame_map: the same chain with the local map points named by id out of a resident MapPoints table
        (id >= 0, ~id = skipped for this frame, outside the map = no point); match / mps / proj_xr index the id list."""
        e = self.e
        im = np.asarray(im)
        assert im.dtype == np.uint8 and im.shape == (e.H, e.W) and im.strides[1] == 1
        ids = np.ascontiguousarray(ids, np.int32)
        M = len(ids)
        tp = TrackParams()
        (tp.grid_cols, tp.grid_rows, tp.min_x, tp.min_y, tp.grid_inv_w, tp.grid_inv_h) = self.grid
        tp.th, tp.nn_ratio, tp.far_points, tp.th_far_points = th, nnRatio, int(bFarPoints), thFarPoints
        kp = np.zeros(e.cap, KP_DTYPE)
        desc = np.zeros((e.cap, 32), np.uint8)
        per = np.zeros(e.nlevels, np.int32)
        mps = np.zeros(max(M, 1), MP_DTYPE)
        xr = np.zeros(max(M, 1), np.float32)
        match = np.full(e.cap, -1, np.int32)
        n, nm = C.c_int(), C.c_int()
        e._chk(self.L.orbfe_track_frame_map(e.h, _p(im), im.strides[0], C.byref(frustum), C.byref(tp), map_points.h, M, _p(ids), _p(kp),
                                            _p(desc), C.byref(n), _p(per), _p(mps), _p(xr), _p(match), C.byref(nm)), "orbfe_track_frame_map")
        k = n.value
        return dict(kp=kp[:k], desc=desc[:k], per_level=per, mps=mps[:M], proj_xr=xr[:M], match=match[:k], nmatches=nm.value)

    def TrackInitialization(self, im, initial_frame, windowSize=40, nnRatio=0.45, checkOrientation=True):
        """orbfe_track_initialization: ExtractORB -> SearchForInitialization(mInitialFrame, mCurrentFrame, 40, 0.45, true)
        (Tracking::MonocularInitialization, src/Tracking.cc:566-607) as one call against a resident InitialFrame ->
        dict(kp, desc, per_level, matches12, nmatches); matches12[i1] = index in the current frame or -1."""
        e = self.e
        im = np.asarray(im)
        assert im.dtype == np.uint8 and im.shape == (e.H, e.W) and im.strides[1] == 1
        tp = TrackParams()
        (tp.grid_cols, tp.grid_rows, tp.min_x, tp.min_y, tp.grid_inv_w, tp.grid_inv_h) = self.grid
        kp = np.zeros(e.cap, KP_DTYPE)
        desc = np.zeros((e.cap, 32), np.uint8)
        per = np.zeros(e.nlevels, np.int32)
        m12 = np.full(max(initial_frame.n, 1), -1, np.int32)
        n, nm = C.c_int(), C.c_int()
        e._chk(self.L.orbfe_track_initialization(e.h, _p(im), im.strides[0], initial_frame.h, C.byref(tp), int(windowSize), nnRatio,
                                                 int(bool(checkOrientation)), _p(kp), _p(desc), C.byref(n), _p(per), _p(m12), C.byref(nm)),
               "orbfe_track_initialization")
        k = n.value
        return dict(kp=kp[:k], desc=desc[:k], per_level=per, matches12=m12[:initial_frame.n], nmatches=nm.value)

    def TrackReferenceKeyFrame(self, im, vocab, levelsup, kf, kfHasMP, nnRatio=0.75, checkOrientation=True):
        """orbfe_track_reference_keyframe: ExtractORB -> the per-feature part of ComputeBoW -> SearchByBoW(reference key frame,
        frame) (Tracking::TrackReferenceKeyFrame, src/Tracking.cc:825-835) as one call against a resident KeyFrame ->
        dict(kp, desc, per_level, word, node, weight, match, nmatches); match[i] = key-frame feature or -1."""
        e = self.e
        im = np.asarray(im)
        assert im.dtype == np.uint8 and im.shape == (e.H, e.W) and im.strides[1] == 1
        has = np.ascontiguousarray(kfHasMP, np.uint8)
        assert len(has) == kf.n
        kp = np.zeros(e.cap, KP_DTYPE)
        desc = np.zeros((e.cap, 32), np.uint8)
        per = np.zeros(e.nlevels, np.int32)
        word = np.zeros(e.cap, np.int32)
        node = np.zeros(e.cap, np.int32)
        weight = np.zeros(e.cap, np.float64)
        match = np.full(e.cap, -1, np.int32)
        n, nm = C.c_int(), C.c_int()
        e._chk(self.L.orbfe_track_reference_keyframe(e.h, _p(im), im.strides[0], vocab.v, int(levelsup), kf.h, _p(has), nnRatio,
                                                     int(checkOrientation), _p(kp), _p(desc), C.byref(n), _p(per), _p(word), _p(node),
                                                     _p(weight), _p(match), C.byref(nm)), "orbfe_track_reference_keyframe")
        k = n.value
        return dict(kp=kp[:k], desc=desc[:k], per_level=per, word=word[:k], node=node[:k], weight=weight[:k], match=match[:k],
                    nmatches=nm.value)


class ImagePreparer:
    """ImageGrabber::ConvertImageToGPU (ros2_ws/src/mono-inertial/include/image_grabber.hpp:96-110): fisheye remap
    (INTER_CUBIC) + resize (INTER_LINEAR) + BGR2GRAY as one kernel; `extract` chains ORBextractor::extractFeatures
    without the grey frame leaving the device."""

    def __init__(self, extractor, map1, map2, dst_w, dst_h):
        self.e = extractor
        self.L = extractor.L
        map1 = np.ascontiguousarray(map1, np.float32)
        map2 = np.ascontiguousarray(map2, np.float32)
        assert map1.ndim == 2 and map1.shape == map2.shape
        self.src_h, self.src_w = map1.shape
        self.dst_w, self.dst_h = int(dst_w), int(dst_h)
        self.p = C.c_void_p()
        extractor._chk(self.L.orbfe_prep_create(extractor.h, self.src_w, self.src_h, _p(map1), _p(map2), self.dst_w,
                                                self.dst_h, C.byref(self.p)), "orbfe_prep_create")

    def close(self):
        if self.p:
            self.L.orbfe_prep_destroy(self.p)
            self.p = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _bgr(self, bgr):
        assert bgr.dtype == np.uint8 and bgr.ndim == 3 and bgr.shape == (self.src_h, self.src_w, 3) and bgr.strides[2] == 1 \
            and bgr.strides[1] == 3
        return bgr, bgr.strides[0]

    def prepare(self, bgr):
        bgr, pitch = self._bgr(bgr)
        out = np.zeros((self.dst_h, self.dst_w), np.uint8)
        self.e._chk(self.L.orbfe_prepare_image(self.e.h, self.p, bgr.ctypes.data, pitch, _p(out), self.dst_w),
                    "orbfe_prepare_image")
        return out

    def prepare_device(self, d_bgr, pitch, d_gray, gray_pitch, stream=None):
        self.e._chk(self.L.orbfe_prepare_image_device(self.e.h, self.p, d_bgr, pitch, d_gray, gray_pitch, stream),
                    "orbfe_prepare_image_device")

    def extract(self, bgr, want_gray=False):
        """-> (keypoints, descriptors[, grey]) or None when the frame has no keypoints (as extractFeatures)."""
        bgr, pitch = self._bgr(bgr)
        cap = self.e.cap
        kp = np.zeros(cap, KP_DTYPE)
        desc = np.zeros((cap, 32), np.uint8)
        n = C.c_int()
        per = np.zeros(self.e.nlevels, np.int32)
        gray = np.zeros((self.dst_h, self.dst_w), np.uint8) if want_gray else None
        self.e._chk(self.L.orbfe_prepare_and_extract(self.e.h, self.p, bgr.ctypes.data, pitch, _p(kp), _p(desc), C.byref(n),
                                                     _p(per), _p(gray), self.dst_w), "orbfe_prepare_and_extract")
        if n.value == 0:
            return None
        return (kp[:n.value], desc[:n.value], gray) if want_gray else (kp[:n.value], desc[:n.value])


class ORBVocabulary:
    """Device copy of a DBoW2 vocabulary tree + the per-feature descent of Frame::ComputeBoW
    (src/Frame.cc:483-495 -> TemplatedVocabulary::transform)."""

    def __init__(self, extractor, childOff, childIdx, nodeDesc, wordId, weight, L):
        self.e, self.L_ = extractor, extractor.L
        a32 = lambda v: np.ascontiguousarray(v, np.int32)
        self._keep = (a32(childOff), a32(childIdx), np.ascontiguousarray(nodeDesc, np.uint8), a32(wordId),
                      np.ascontiguousarray(weight, np.float64))
        self.v = C.c_void_p()
        co, ci_, nd, wi, we = self._keep
        extractor._chk(self.L_.orbfe_vocab_create(extractor.h, len(wi), _p(co), _p(ci_), _p(nd), _p(wi), _p(we), int(L),
                                                  C.byref(self.v)), "orbfe_vocab_create")

    def close(self):
        if getattr(self, "v", None):
            self.L_.orbfe_vocab_destroy(self.v)
            self.v = None

    __del__ = close

    def transform(self, desc, levelsup):
        desc = np.ascontiguousarray(desc, np.uint8)
        n = len(desc)
        word = np.zeros(max(n, 1), np.int32)
        node = np.zeros(max(n, 1), np.int32)
        weight = np.zeros(max(n, 1), np.float64)
        self.e._chk(self.L_.orbfe_bow_transform(self.e.h, self.v, _p(desc), n, levelsup, _p(word), _p(node), _p(weight)),
                    "orbfe_bow_transform")
        return word[:n], node[:n], weight[:n]

    # weighting / scoring enums of DBoW2 (TemplatedVocabulary.h:35-55)
    TF_IDF, TF, IDF, BINARY = 0, 1, 2, 3
    L1_NORM, L2_NORM, CHI_SQUARE, KL, BHATTACHARYYA, DOT_PRODUCT = 0, 1, 2, 3, 4, 5

    def transform_bow(self, desc, levelsup, weighting=0, scoring=0):
        """TemplatedVocabulary::transform(features, BowVector&, FeatureVector&, levelsup)
        (TemplatedVocabulary.h:1136-1204) as Frame::ComputeBoW calls it (src/Frame.cc:483-495, levelsup 4).
        The per-feature tree descent runs on the GPU; the map assembly (<= N inserts) stays on the host in the
        reference's order so the doubles come out bit-identical.  Returns (BowVector, FeatureVector) as dicts
        in ascending key order (std::map iteration order)."""
        word, node, w = self.transform(desc, levelsup)
        bow, fv = {}, {}
        tf = weighting in (self.TF_IDF, self.TF)
        for i in range(len(word)):
            wi = float(w[i])
            if wi > 0:
                k = int(word[i])
                if tf:
                    bow[k] = bow.get(k, 0.0) + wi      # BowVector::addWeight
                elif k not in bow:
                    bow[k] = wi                        # BowVector::addIfNotExist
                fv.setdefault(int(node[i]), []).append(i)  # FeatureVector::addFeature
        bow = dict(sorted(bow.items()))
        fv = dict(sorted(fv.items()))
        must = scoring != self.DOT_PRODUCT
        if tf and bow and not must:
            nd = float(len(bow))
            for k in bow:
                bow[k] /= nd
        if must:                                       # BowVector::normalize (src/DBoW2/BowVector.cpp:62-84)
            norm = 0.0
            if scoring == self.L2_NORM:
                for v in bow.values():
                    norm += v * v
                norm = float(np.sqrt(np.float64(norm)))
            else:
                for v in bow.values():
                    norm += abs(v)
            if norm > 0.0:
                for k in bow:
                    bow[k] /= norm
        return bow, fv


def load_vocabulary_text(path):
    """Parse the ORBvoc.txt format of TemplatedVocabulary::loadFromTextFile (TemplatedVocabulary.h:1349-1436)
    into the CSR tree orbfe_vocab_create takes.  Returns a dict with k, L, scoring, weighting, childOff,
    childIdx, nodeDesc, wordId, weight.  Blank lines are skipped (the reference would read a node from
    them with an indeterminate parent)."""
    with open(path) as f:
        k, L, scoring, weighting = (int(t) for t in f.readline().split()[:4])
        if k < 0 or k > 20 or L < 1 or L > 10 or scoring < 0 or scoring > 5 or weighting < 0 or weighting > 3:
            raise ValueError("not a DBoW2 text vocabulary: %s" % path)
        parent, desc, weight, leaf = [0], [np.zeros(32, np.uint8)], [0.0], [0]
        for line in f:
            t = line.split()
            if not t:
                continue
            parent.append(int(t[0]))
            leaf.append(int(t[1]))
            desc.append(np.array([int(v) for v in t[2:34]], np.uint8))
            weight.append(float(t[34]))
    n = len(parent)
    kids = [[] for _ in range(n)]
    for i in range(1, n):
        kids[parent[i]].append(i)
    childOff = np.zeros(n + 1, np.int32)
    childOff[1:] = np.cumsum([len(c) for c in kids])
    childIdx = np.array([c for cs in kids for c in cs], np.int32)
    wordId = np.zeros(n, np.int32)  # Node::word_id defaults to 0 (TemplatedVocabulary.h:275)
    nw = 0
    for i in range(1, n):
        if leaf[i] > 0:
            wordId[i] = nw
            nw += 1
    return dict(k=k, L=L, scoring=scoring, weighting=weighting, childOff=childOff, childIdx=childIdx,
                nodeDesc=np.stack(desc), wordId=wordId, weight=np.array(weight, np.float64))
