"""Edge cases on the GPU path: ragged / empty inputs, padded pitches and strides, argument validation."""
import ctypes as C

import numpy as np
import pytest

import oracle_py as O
from orbfe import synth

pytestmark = pytest.mark.gpu
ARGS = (300, 20000, 1.2, 4, 20, 7, 320, 240)


def test_batch_with_blank_frames_and_partial_batch(built):
    import orbfe
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=8)
    ref = O.Extractor(*ARGS)
    blank = np.full((240, 320), 60, np.uint8)
    ims = [synth.frame(320, 240, 1), blank, synth.frame(320, 240, 2), blank, blank]  # 5 of max 8
    res = ex.extract_batch(ims)
    assert len(res) == 5
    for im, (kp, desc, per) in zip(ims, res):
        kp_r, desc_r, per_r = ref.extract(im)
        assert len(kp) == len(kp_r) and kp.tobytes() == kp_r.tobytes() and np.array_equal(desc, desc_r)
        assert np.array_equal(per, per_r)
    assert len(res[1][0]) == 0 and len(res[3][0]) == 0


def test_padded_pitch_and_frame_stride(built):
    import torch
    import orbfe
    W, H, B = 320, 240, 3
    pitch, stride = 352, 352 * 250  # padded rows and padded frames
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=B)
    ref = O.Extractor(*ARGS)
    buf = np.random.default_rng(0).integers(0, 256, B * stride, dtype=np.uint8)  # garbage in the padding
    ims = [synth.frame(W, H, 30 + b) for b in range(B)]
    for b in range(B):
        v = buf[b * stride:b * stride + H * pitch].reshape(H, pitch)
        v[:, :W] = ims[b]
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(buf).to(dev)
    cap = ex.cap
    d_kp = torch.zeros(B * cap * 24, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_in.data_ptr(), stride, pitch, B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), None, None)
    torch.cuda.synchronize(dev)
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy().view(orbfe.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().reshape(B, cap, 32)
    for b in range(B):
        kp_r, desc_r, _ = ref.extract(ims[b])
        assert n[b] == len(kp_r) and kp[b, :n[b]].tobytes() == kp_r.tobytes() and np.array_equal(desc[b, :n[b]], desc_r)
    # host API with a padded pitch
    padded = np.zeros((H, pitch), np.uint8)
    padded[:, :W] = ims[0]
    got = ex.extractFeatures(padded[:, :W])
    kp_r, desc_r, _ = ref.extract(ims[0])
    assert got[0].tobytes() == kp_r.tobytes() and np.array_equal(got[1], desc_r)


@pytest.mark.parametrize("pitch_extra,base_off", [(3, 0), (0, 1), (5, 3), (1, 2)])
def test_unaligned_device_input(built, pitch_extra, base_off):
    """Level 0 handed over with a pitch or a base address that is not a multiple of 4 (a cropped view of a larger image):
    the kernels leave the dword paths (byte staging in the FAST kernel, the tile resize kernel for level 1, unstaged patch
    reads in orient_brief) and must give the same bytes."""
    import torch
    import orbfe
    W, H, B = 321, 243, 2
    args = (300, 20000, 1.2, 4, 20, 7, W, H)
    pitch = W + pitch_extra
    stride = pitch * H + 7
    ex = orbfe.ORBextractor(*args, device=0, max_batch=B)
    ref = O.Extractor(*args)
    buf = np.random.default_rng(1).integers(0, 256, base_off + B * stride + 16, dtype=np.uint8)  # garbage around the frames
    ims = [synth.frame(W, H, 50 + b) for b in range(B)]
    for b in range(B):
        v = buf[base_off + b * stride: base_off + b * stride + H * pitch].reshape(H, pitch)
        v[:, :W] = ims[b]
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(buf).to(dev)
    cap = ex.cap
    d_kp = torch.zeros(B * cap * 24, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_in.data_ptr() + base_off, stride, pitch, B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), None, None)
    torch.cuda.synchronize(dev)
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy().view(orbfe.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().reshape(B, cap, 32)
    for b in range(B):
        kp_r, desc_r, _ = ref.extract(ims[b])
        assert n[b] == len(kp_r) and kp[b, :n[b]].tobytes() == kp_r.tobytes() and np.array_equal(desc[b, :n[b]], desc_r), b
        for l in range(4):
            assert np.array_equal(ex.pyramid_level(l, True, frame=b), ref.level_image(l, True)), (b, l)


def test_square_frame_at_the_size_limit_through_the_chained_pyramid(built):
    """4095 x 4095, one frame per call: the single-frame path builds two pyramid levels per launch (launch_pyramid_chain), and
    levels 1 + 2 of this frame are 19.7 M pixels = 77 k blocks of 256 -- more than the 65535 a grid's y dimension takes (the
    pixel index travels on grid.x since round 5).  Every level image and the keypoints against the oracle."""
    import orbfe
    W = H = 4095
    args = (3000, 400000, 1.2, 4, 20, 7, W, H)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    ref = O.Extractor(*args)
    im = synth.frame(W, H, 5)
    kp, desc = ex.extractFeatures(im)
    kp_r, desc_r, _ = ref.extract(im)
    assert len(kp) == len(kp_r) > 2500 and kp.tobytes() == kp_r.tobytes() and np.array_equal(desc, desc_r)
    for l in range(1, 4):
        assert np.array_equal(ex.pyramid_level(l, False), ref.level_image(l, False)), l


@pytest.mark.parametrize("W,H", [(3840, 2160), (4095, 2303)])
def test_frames_at_the_size_limit(built, W, H):
    """4K frames and the largest level the packed candidate words allow (x, y < 4096): 130 x 72 FAST
    tiles in level 0, 32-bit byte offsets of ~9.4 MB per frame, 5000 features, one frame and a batch of two."""
    import orbfe
    args = (5000, 400000, 1.2, 8, 20, 7, W, H)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=2)
    ref = O.Extractor(*args)
    ims = [synth.frame(W, H, 3), synth.hostile("pink", W, H, 1)]
    res = ex.extract_batch(ims)
    for im, (kp, desc, per) in zip(ims, res):
        kp_r, desc_r, per_r = ref.extract(im)
        assert len(kp) == len(kp_r) > 4000 and np.array_equal(per, per_r)
        assert kp.tobytes() == kp_r.tobytes() and np.array_equal(desc, desc_r)
    # one past the limit is refused at create
    with pytest.raises(orbfe.OrbfeError) as e:
        orbfe.ORBextractor(5000, 400000, 1.2, 8, 20, 7, 4097, 2303, device=0, max_batch=1)
    assert e.value.code in (1, 2)


def test_matcher_empty_and_invalid_inputs(built):
    import orbfe
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=1)
    m = orbfe.ORBmatcher(ex)
    kp, desc = ex.extractFeatures(synth.frame(320, 240, 5))
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 320.0, 240.0, ex.mvScaleFactor)
    # no map points
    n, out = m.SearchByProjection(fv, np.zeros(0, orbfe.MP_DTYPE), np.zeros((0, 32), np.uint8), 20.0, False, 0.0, 0.85, None)
    assert n == 0 and (out == -1).all()
    # all map points filtered (not in view / bad)
    mps = np.zeros(10, orbfe.MP_DTYPE)
    mps["proj_x"], mps["proj_y"], mps["view_cos"] = 100, 100, 1.0
    mps["bad"] = 1
    mps["in_view"] = 1
    n, out = m.SearchByProjection(fv, mps, np.zeros((10, 32), np.uint8), 20.0, False, 0.0, 0.85, None)
    assert n == 0 and (out == -1).all()
    # projections far outside the image: empty cell range
    mps["bad"] = 0
    mps["proj_x"] = 5000
    n, out = m.SearchByProjection(fv, mps, np.zeros((10, 32), np.uint8), 20.0, False, 0.0, 0.85, None)
    assert n == 0
    # level out of range is rejected by the host API
    mps["proj_x"] = 100
    mps["level"] = 9
    with pytest.raises(orbfe.OrbfeError) as ei:
        m.SearchByProjection(fv, mps, np.zeros((10, 32), np.uint8), 20.0, False, 0.0, 0.85, None)
    assert ei.value.code == 1
    # empty frame
    fv0 = orbfe.make_frame_view(kp[:0], desc[:0], 64, 48, 0.0, 0.0, 320.0, 240.0, ex.mvScaleFactor)
    mps["level"] = 0
    n, out = m.SearchByProjection(fv0, mps, np.zeros((10, 32), np.uint8), 20.0, False, 0.0, 0.85, None)
    assert n == 0 and len(out) == 0
    # BoW with no shared node
    n, out = m.SearchByBoW([0], [], [0], [], desc, kp["angle"], np.ones(len(kp), np.uint8), desc, kp["angle"], 0.75, True)
    assert n == 0 and (out == -1).all()
    # initialization against an empty second frame
    n, out = m.SearchForInitialization(fv, fv0, 40, 0.9, True)
    assert n == 0 and (out == -1).all()


def test_extract_rejects_bad_arguments(built):
    import orbfe
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=2)
    L = ex.L
    img = synth.frame(320, 240, 1)
    kp = np.zeros(ex.cap, orbfe.KP_DTYPE)
    desc = np.zeros((ex.cap, 32), np.uint8)
    n = C.c_int()
    assert L.orbfe_extract(ex.h, None, 320, kp.ctypes.data, desc.ctypes.data, C.byref(n), None) == 1
    assert L.orbfe_extract(ex.h, img.ctypes.data, 100, kp.ctypes.data, desc.ctypes.data, C.byref(n), None) == 1  # pitch < width
    ptrs = (C.c_void_p * 3)(img.ctypes.data, img.ctypes.data, img.ctypes.data)
    assert L.orbfe_extract_batch(ex.h, ptrs, 320, 3, kp.ctypes.data, desc.ctypes.data, C.byref(n), None) == 1  # batch > max_batch
    assert L.orbfe_get_pyramid_level(ex.h, 0, 99, 0, kp.ctypes.data, 320) == 1


def test_last_frame_flush_with_the_end_of_its_buffer(built):
    """the pyramid kernel's row offset travels in the buffer instruction's scalar offset, which the range check
    need not see -- its lanes are bounded by the row width instead.  A tight [B][H][W] device buffer of EXACTLY the
    contract size (pitch * (H - 1) + round4(W) for the last frame, W * H = 225 pages at 1280x720) gives the oracle's
    results for the last frame, and a device pitch past the 32-bit frame extent is refused."""
    import torch
    import orbfe
    args = (500, 40000, 1.2, 8, 20, 7, 1280, 720)
    W, H, B = 1280, 720, 2
    ex = orbfe.ORBextractor(*args, device=0, max_batch=B)
    ref = O.Extractor(*args)
    ims = [synth.frame(W, H, 70 + b) for b in range(B)]
    dev = torch.device("cuda", 0)
    d_in = torch.from_numpy(np.stack(ims).reshape(-1)).to(dev)  # numel == B * W * H: nothing of ours behind the last row
    assert d_in.numel() == B * W * H
    cap = ex.cap
    d_kp = torch.zeros(B * cap * 24, dtype=torch.uint8, device=dev)
    d_desc = torch.zeros(B * cap * 32, dtype=torch.uint8, device=dev)
    d_n = torch.zeros(B, dtype=torch.int32, device=dev)
    ex.extract_batch_device(d_in.data_ptr(), W * H, W, B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), None, None)
    torch.cuda.synchronize(dev)
    n = d_n.cpu().numpy()
    kp = d_kp.cpu().numpy().view(orbfe.KP_DTYPE).reshape(B, cap)
    desc = d_desc.cpu().numpy().reshape(B, cap, 32)
    for b in range(B):
        kp_r, desc_r, _ = ref.extract(ims[b])
        assert n[b] == len(kp_r) and kp[b, :n[b]].tobytes() == kp_r.tobytes() and np.array_equal(desc[b, :n[b]], desc_r)
    for l in range(1, 8):
        assert np.array_equal(ex.pyramid_level(l, False, frame=B - 1), O_level(ref, ims[B - 1], l)), l
    # 32-bit frame extent: 720 rows at a 4 MiB pitch end past 0x7ffffff0 -> refused before any launch
    with pytest.raises(orbfe.OrbfeError) as e:
        ex.extract_batch_device(d_in.data_ptr(), 0, 4 << 20, 1, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), None, None)
    assert e.value.code == 1


def O_level(ref, img, l):
    ref.extract(img)
    return ref.level_image(l, False)


@pytest.mark.gpu
def test_graph_replay_equals_plain_launches(built):
    """The host-pointer extract call is replayed as a captured hipGraph; with stage timing on it takes the plain-launch
    path.  Both must give identical results, for alternating batch sizes and images, including the pyramid getter."""
    import orbfe
    from orbfe import synth
    W, H = 320, 240
    ex = orbfe.ORBextractor(600, 8000, 1.2, 6, 20, 7, W, H, device=0, max_batch=3)
    imgs = [synth.frame(W, H, i) for i in range(5)]

    def run_all():
        out = []
        for i, im in enumerate(imgs):
            out.append(ex.extractFeatures(im))
            lvl = ex.pyramid_level(2, True, 0)
            out.append((lvl.copy(),))
            if i % 2 == 0:
                out.append(ex.extract_batch([imgs[i], imgs[(i + 1) % 5], imgs[(i + 2) % 5]]))
        return out

    a = run_all()               # captures graphs for batch 1 and 3, then replays
    b = run_all()               # replays only
    ex.set_stage_timing(True)   # plain launches with stage events
    c = run_all()
    ms, n = ex.stage_ms()
    ex.set_stage_timing(False)
    assert n > 0 and ms["total"] > 0

    def same(u, v):
        if isinstance(u, (tuple, list)):
            return len(u) == len(v) and all(same(x, y) for x, y in zip(u, v))
        return np.array_equal(np.asarray(u), np.asarray(v))

    assert same(a, b) and same(a, c)


@pytest.mark.gpu
def test_pinned_input_fast_path(built):
    """Pinned host images (what the reference passes: cv::cuda::HostMem) are uploaded straight from the caller's buffer
    with a pitch-converting DMA copy; results must equal the pageable (staged) path, also for a padded pitch."""
    import torch
    import orbfe
    from orbfe import synth
    W, H = 320, 240
    ex = orbfe.ORBextractor(600, 8000, 1.2, 6, 20, 7, W, H, device=0, max_batch=2)
    img = synth.frame(W, H, 3)
    ref = ex.extractFeatures(img)
    pin = torch.empty((H, W + 40), dtype=torch.uint8).pin_memory()
    pin[:, :W] = torch.from_numpy(img)
    view = pin.numpy()[:, :W]  # pitch W + 40, pinned
    got = ex.extractFeatures(view)
    assert got[0].tobytes() == ref[0].tobytes() and np.array_equal(got[1], ref[1])
    got2 = ex.extractFeatures(view)  # graph replay after the pinned upload
    assert got2[0].tobytes() == ref[0].tobytes()
    # a pinned image whose pitch fits the device rows is sent without re-pitching and read with the caller's pitch
    ex2 = orbfe.ORBextractor(600, 8000, 1.2, 6, 20, 7, 300, H, device=0, max_batch=2)  # device pitch 320
    img2 = synth.frame(300, H, 4)
    ref2 = ex2.extractFeatures(img2)
    for pitch in (300, 304, 320):
        pin2 = torch.empty((H, pitch), dtype=torch.uint8).pin_memory()
        pin2[:, :300] = torch.from_numpy(img2)
        v2 = pin2.numpy()[:, :300]
        for _ in range(2):  # capture, then replay
            g = ex2.extractFeatures(v2)
            assert g[0].tobytes() == ref2[0].tobytes() and np.array_equal(g[1], ref2[1]), pitch
        b = ex2.extract_batch([v2, v2])
        assert b[0][0].tobytes() == ref2[0].tobytes() and b[1][0].tobytes() == ref2[0].tobytes()


def test_sim3_reloc_prep_reject_bad_arguments(built):
    """Status codes (never aborts) of the entry points added for the remaining ORBmatcher statics and the image
    preparation: pyramid depth mismatches, unknown camera model, size mismatches, empty inputs."""
    import frustum_scenarios as FS
    import orbfe
    from test_frustum import PN
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=1)
    m = orbfe.ORBmatcher(ex)
    kp, desc = ex.extractFeatures(synth.frame(320, 240, 5))
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 320.0, 240.0, ex.mvScaleFactor)
    fv0 = orbfe.make_frame_view(kp[:0], desc[:0], 64, 48, 0.0, 0.0, 320.0, 240.0, ex.mvScaleFactor)
    F = orbfe.Frustum()
    FS.fill_frustum(F, PN, W=320.0, H=240.0, n_levels=ex.nlevels, seed=1)
    pts = np.zeros(5, orbfe.WP_DTYPE)
    pts["z"], pts["max_distance"], pts["min_distance"] = 3.0, 10.0, 0.1
    d = np.zeros((5, 32), np.uint8)
    ang = np.zeros(5, np.float32)
    # more predicted levels than the frame has scale factors
    F.n_levels = ex.nlevels + 1
    for call in (lambda: m.SearchByProjection_keyframe(fv, F, pts, d, ang, None, 10.0, True),
                 lambda: m.Fuse_search_sim3(fv, F, 4.0, pts, d)):
        with pytest.raises(orbfe.OrbfeError) as ei:
            call()
        assert ei.value.code == 1
    F.n_levels = ex.nlevels
    F.camera_model = 7
    with pytest.raises(orbfe.OrbfeError) as ei:
        m.SearchByProjection_keyframe(fv, F, pts, d, ang, None, 10.0, True)
    assert ei.value.code in (1, 2)
    F.camera_model = 0
    # empty frame / no points: defined results, no launch
    n, out = m.SearchByProjection_keyframe(fv0, F, pts, d, ang, None, 10.0, True)
    assert n == 0 and len(out) == 0
    bi, bd = m.Fuse_search_sim3(fv0, F, 4.0, pts, d)
    assert (bi == -1).all() and (bd == 256).all()
    D = orbfe.Sim3View()
    D.rcw[0] = D.rcw[4] = D.rcw[8] = 1.0
    D.sr[0] = D.sr[4] = D.sr[8] = 1.0
    D.fx = D.fy = 300.0
    D.cx, D.cy, D.max_x, D.max_y = 160.0, 120.0, 320.0, 240.0
    D.log_scale_factor, D.n_levels = float(np.log(np.float32(1.2))), ex.nlevels
    wp = np.zeros(len(kp), orbfe.WP_DTYPE)
    wp["skip"] = 1
    n, out = m.SearchBySim3(fv, fv, D, D, wp, desc, wp, desc, 7.5)  # every feature without a map point
    assert n == 0 and (out == -1).all()
    n, out = m.SearchBySim3(fv0, fv, D, D, wp[:0], desc[:0], wp, desc, 7.5)
    assert n == 0 and len(out) == 0
    D.n_levels = 0
    with pytest.raises(orbfe.OrbfeError) as ei:
        m.SearchBySim3(fv, fv, D, D, wp, desc, wp, desc, 7.5)
    assert ei.value.code == 1
    # image preparation: map / image / extractor size mismatches
    m1 = np.zeros((48, 64), np.float32)
    with pytest.raises(orbfe.OrbfeError):
        orbfe.ImagePreparer(ex, m1, m1, 0, 10)
    prep = orbfe.ImagePreparer(ex, m1, m1, 32, 24)
    with pytest.raises(AssertionError):
        prep.prepare(np.zeros((48, 63, 3), np.uint8))
    with pytest.raises(orbfe.OrbfeError) as ei:  # the extractor was created for 320x240 frames
        prep.extract(np.zeros((48, 64, 3), np.uint8))
    assert ei.value.code == 1
    assert prep.prepare(np.full((48, 64, 3), 200, np.uint8)).shape == (24, 32)
