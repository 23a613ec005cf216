"""The device-pointer entry points of include/orbfe.h against the oracle, each alone and chained WITHOUT a host copy in
between: orbfe_prepare_image_device -> orbfe_extract_batch_device -> orbfe_project_map_points_device ->
orbfe_match_projection_batch_device == O.prepare_image -> O.Extractor.extract -> O.is_in_frustum -> O.search_by_projection
(image_grabber.hpp:96-110 -> src/Frame.cc:178-189 -> src/Tracking.cc:1059-1115); plus orbfe_stream_collect_view."""
import numpy as np
import pytest

import frustum_scenarios as FS
import oracle_py as O
from frustum_scenarios import world_points_on_keypoints as _world_points_on_keypoints
from test_frustum import ON, OP, PN

pytestmark = pytest.mark.gpu


def _dev(t, a):
    """numpy array -> uint8 device tensor holding its bytes"""
    return t.from_numpy(np.ascontiguousarray(a).view(np.uint8).reshape(-1).copy()).cuda()


def _host(x, dtype, shape=None):
    a = x.cpu().numpy().view(dtype)
    return a.reshape(shape) if shape is not None else a


@pytest.mark.parametrize("n,seed,kb8,own_stream", [(1, 0, False, False), (257, 1, False, True), (5000, 2, False, False),
                                                    (100000, 3, False, True), (20000, 4, True, True)])
def test_project_map_points_device_matches_oracle(built, n, seed, kb8, own_stream):
    import torch
    import orbfe
    e = orbfe.ORBextractor(500, 2000, 1.2, 8, 20, 7, 320, 240)
    m = orbfe.ORBmatcher(e)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    FS.fill_frustum(Fo, ON, seed=seed, kb8=kb8)
    FS.fill_frustum(Fp, PN, seed=seed, kb8=kb8)
    pts = FS.world_points(n, O.WP_DTYPE, OP, seed=seed + 10)
    ref, ref_xr = O.is_in_frustum(Fo, pts)
    d_pts = _dev(torch, pts)
    d_out = torch.full((n * orbfe.MP_DTYPE.itemsize,), 0xAB, dtype=torch.uint8, device="cuda")
    d_xr = torch.full((n,), -7.0, dtype=torch.float32, device="cuda")
    st = torch.cuda.Stream() if own_stream else None
    torch.cuda.synchronize()
    m.isInFrustum_batch_device(Fp, n, d_pts.data_ptr(), d_out.data_ptr(), d_xr.data_ptr(), st.cuda_stream if st else None)
    if st:
        st.synchronize()
    else:
        torch.cuda.synchronize()  # stream NULL == the handle's own (non-blocking) stream: wait for the device
    assert _host(d_out, orbfe.MP_DTYPE).tobytes() == ref.tobytes()
    assert d_xr.cpu().numpy().tobytes() == ref_xr.tobytes()
    # proj_xr is optional; n == 0 is a no-op; a bad camera model is refused before any launch
    d_out.fill_(0)
    torch.cuda.synchronize()  # the fill runs on torch's stream, the kernel on the handle's non-blocking one
    m.isInFrustum_batch_device(Fp, n, d_pts.data_ptr(), d_out.data_ptr(), None, None)
    torch.cuda.synchronize()
    assert _host(d_out, orbfe.MP_DTYPE).tobytes() == ref.tobytes()
    m.isInFrustum_batch_device(Fp, 0, None, None, None, None)
    Fp.camera_model = 7
    with pytest.raises(orbfe.OrbfeError):
        m.isInFrustum_batch_device(Fp, n, d_pts.data_ptr(), d_out.data_ptr(), None, None)


@pytest.mark.parametrize("w,h,dw,dh,seed,pad", [(160, 120, 48, 36, 1, 0), (97, 61, 97, 61, 2, 7), (640, 480, 192, 144, 4, 64),
                                                (2048, 1536, 614, 460, 5, 0)])
def test_prepare_image_device_matches_oracle(built, w, h, dw, dh, seed, pad):
    import torch
    import orbfe
    from orbfe.synth import colour_image, fisheye_maps
    img = colour_image(w, h, seed)
    m1, m2 = fisheye_maps(w, h, seed)
    ref = O.prepare_image(img, m1, m2, dw, dh)
    ex = orbfe.ORBextractor(300, 4000, 1.2, 3, 20, 7, max(dw, 64), max(dh, 64))
    prep = orbfe.ImagePreparer(ex, m1, m2, dw, dh)
    # pitched source (3 * w + pad bytes per row) and pitched destination (dw + 5): only the dw x dh pixels are written
    spitch, gpitch = 3 * w + pad, dw + 5
    src = np.full((h, spitch), 0x5A, np.uint8)
    src[:, :3 * w] = img.reshape(h, 3 * w)
    d_src = torch.from_numpy(src).cuda()
    d_gray = torch.full((dh, gpitch), 0xEE, dtype=torch.uint8, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    prep.prepare_device(d_src.data_ptr(), spitch, d_gray.data_ptr(), gpitch, st.cuda_stream)
    st.synchronize()
    got = d_gray.cpu().numpy()
    assert np.array_equal(got[:, :dw], ref)
    assert (got[:, dw:] == 0xEE).all()
    with pytest.raises(orbfe.OrbfeError):  # a pitch shorter than a row is refused
        prep.prepare_device(d_src.data_ptr(), 3 * w - 1, d_gray.data_ptr(), gpitch, None)
    prep.close()


def test_device_chain_prepare_extract_project_match(built):
    """Four device-pointer calls on ONE caller stream, nothing crosses PCIe in between; results are downloaded once at the
    end and every stage is compared with the oracle's chain."""
    import torch
    import orbfe
    from orbfe import synth
    from orbfe.synth import fisheye_maps
    w, h, dw, dh = 1024, 768, 614, 460
    args = (1000, 20000, 1.2, 8, 20, 7, dw, dh)
    base = synth.frame(w, h, 3)
    img = np.stack([base, np.roll(base, 1, 1), 255 - base // 2], 2).astype(np.uint8)
    m1, m2 = fisheye_maps(w, h, 9, strength=0.1)
    # ---- oracle chain ----
    grey_ref = O.prepare_image(img, m1, m2, dw, dh)
    eo = O.Extractor(*args)
    kp_r, desc_r, _ = eo.extract(grey_ref)
    Fo, Fp = O.Frustum(), orbfe.Frustum()
    v = FS.fill_frustum(Fo, ON, W=float(dw), H=float(dh), seed=7)
    FS.fill_frustum(Fp, PN, W=float(dw), H=float(dh), seed=7)
    M = 1800
    pts, mpd = _world_points_on_keypoints(kp_r, desc_r, v, M, np.random.default_rng(11), 8)
    mps_ref, _ = O.is_in_frustum(Fo, pts)
    fvo = O.make_frame_view(kp_r, desc_r, 64, 48, 0.0, 0.0, float(dw), float(dh), eo.scaleFactors)
    n_ref, match_ref = O.search_by_projection(fvo, mps_ref, mpd, None, 3.0, 0.8)
    assert len(kp_r) > 500 and n_ref > 100 and 0.3 < mps_ref["inView"].mean() < 1.0
    # ---- device chain ----
    ex = orbfe.ORBextractor(*args)
    m = orbfe.ORBmatcher(ex)
    prep = orbfe.ImagePreparer(ex, m1, m2, dw, dh)
    cap = ex.cap
    gpitch = (dw + 63) // 64 * 64
    d_bgr = torch.from_numpy(img.reshape(h, 3 * w)).cuda()
    d_gray = torch.zeros((dh, gpitch), dtype=torch.uint8, device="cuda")
    d_kp = torch.zeros(cap * 24, dtype=torch.uint8, device="cuda")
    d_desc = torch.zeros(cap * 32, dtype=torch.uint8, device="cuda")
    d_n = torch.zeros(1, dtype=torch.int32, device="cuda")
    d_per = torch.zeros(8, dtype=torch.int32, device="cuda")
    d_pts, d_mpd = _dev(torch, pts), _dev(torch, mpd)
    d_mps = torch.zeros(M * orbfe.MP_DTYPE.itemsize, dtype=torch.uint8, device="cuda")
    d_match = torch.full((cap,), -9, dtype=torch.int32, device="cuda")
    d_nm = torch.zeros(1, dtype=torch.int32, device="cuda")
    st = torch.cuda.Stream()
    torch.cuda.synchronize()
    s = st.cuda_stream
    prep.prepare_device(d_bgr.data_ptr(), 3 * w, d_gray.data_ptr(), gpitch, s)
    ex.extract_batch_device(d_gray.data_ptr(), gpitch * dh, gpitch, 1, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(),
                            d_per.data_ptr(), s)
    m.isInFrustum_batch_device(Fp, M, d_pts.data_ptr(), d_mps.data_ptr(), None, s)
    m.SearchByProjection_batch_device(1, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, 64, 48, 0.0, 0.0, float(dw),
                                      float(dh), M, d_mps.data_ptr(), d_mpd.data_ptr(), None, 3.0, 0.8, d_match.data_ptr(),
                                      d_nm.data_ptr(), stream=s)
    st.synchronize()
    assert ex.device_status() == 0
    n = int(d_n.item())
    assert np.array_equal(d_gray.cpu().numpy()[:, :dw], grey_ref), "stage 1 (prepare) differs"
    assert n == len(kp_r), "stage 2 (extract): keypoint count"
    assert _host(d_kp, O.KP_DTYPE)[:n].tobytes() == kp_r.tobytes(), "stage 2 (extract): keypoints"
    assert np.array_equal(d_desc.cpu().numpy().reshape(cap, 32)[:n], desc_r), "stage 2 (extract): descriptors"
    assert _host(d_mps, orbfe.MP_DTYPE).tobytes() == mps_ref.tobytes(), "stage 3 (isInFrustum) differs"
    assert int(d_nm.item()) == n_ref, "stage 4 (SearchByProjection): count"
    assert np.array_equal(d_match.cpu().numpy()[:n], match_ref), "stage 4 (SearchByProjection): indices"
    prep.close()


def test_stream_collect_view_matches_collect_and_oracle(built):
    """orbfe_stream_collect_view hands out pointers into the slot's pinned block: same bytes as orbfe_stream_collect and
    as the oracle, views stay valid until the slot is reused (`slots` submissions later), partial submissions report
    their frame count, and an empty ring is refused."""
    import orbfe
    from orbfe import synth
    args = (600, 24000, 1.2, 6, 20, 7, 376, 240)
    W, H = args[6], args[7]
    slot, n_frames = 8, 29  # 3 full submissions + one of 5
    frames = np.stack(list(synth.stream(W, H, n_frames, index0=310)))
    ex = orbfe.ORBextractor(*args, device=0, max_batch=slot)
    eo = O.Extractor(*args)
    ref = [eo.extract(f) for f in frames]
    st = ex.stream(slots=3, slot_frames=slot)
    with pytest.raises(orbfe.OrbfeError):
        st.collect_view()  # nothing in flight
    got, held = [], []
    pos = 0
    while pos < n_frames or st.in_flight():
        while pos < n_frames:
            k = min(slot, n_frames - pos)
            if not st.submit(frames[pos:pos + k]):
                break
            pos += k
        nf, kp, desc, n, per = st.collect_view()
        assert nf == min(slot, n_frames - len(got))
        held.append((nf, kp, desc, n, per))  # views, NOT copies
        for b in range(nf):
            got.append((kp[b, :n[b]].copy(), desc[b, :n[b]].copy(), per[b].copy()))
    assert len(got) == n_frames
    for i, ((kp_g, desc_g, per_g), (kp_r, desc_r, per_r)) in enumerate(zip(got, ref)):
        assert len(kp_r) > 50 and kp_g.tobytes() == kp_r.tobytes() and np.array_equal(desc_g, desc_r), i
        assert np.array_equal(per_g, per_r), i
    # 4 submissions through 3 slots: submission 3 reused slot 0, so the views of submissions 1..3 are still intact
    base = 0
    for k, (nf, kp, desc, n, per) in enumerate(held):
        if k >= 1:
            for b in range(nf):
                assert kp[b, :n[b]].tobytes() == ref[base + b][0].tobytes(), (k, b)
                assert np.array_equal(desc[b, :n[b]], ref[base + b][1]), (k, b)
        base += nf
    # collect() and collect_view() agree on a fresh submission
    st.submit(frames[:slot])
    a = st.collect()
    st.submit(frames[:slot])
    nf, kp, desc, n, per = st.collect_view()
    for b in range(slot):
        assert a[b][0].tobytes() == kp[b, :n[b]].tobytes() and np.array_equal(a[b][1], desc[b, :n[b]])
    st.close()
