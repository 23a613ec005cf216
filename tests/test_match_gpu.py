"""GPU parity for the matchers: HIP kernels through the C ABI vs the CPU oracle, exact indices."""
import numpy as np
import pytest

import match_scenarios as S
import oracle_py as O
from orbfe import synth

pytestmark = pytest.mark.gpu
NAMES_O = ("projX", "projY", "viewCos", "trackDepth", "level", "inView", "bad", "observations")


def _setup(W=752, H=480, nfeat=1000, levels=8, idx=2):
    import orbfe
    args = (nfeat, 40000, 1.2, levels, 20, 7, W, H)
    e = O.Extractor(*args)
    kp, desc, _ = e.extract(synth.frame(W, H, idx))
    ex = orbfe.ORBextractor(*args, device=0, max_batch=4)
    return orbfe, ex, e, kp, desc


@pytest.mark.parametrize("grid,th,nn,M,seed", [((64, 48), 20.0, 0.85, 2000, 1), ((64, 48), 40.0, 0.75, 2000, 2),
                                               ((512, 512), 20.0, 0.85, 500, 3), ((16, 12), 1.0, 0.9, 300, 4)])
def test_search_by_projection_exact(built, grid, th, nn, M, seed):
    orbfe, ex, e, kp, desc = _setup()
    mps, mpd, init_obs = S.projection_scenario(kp, desc, M, seed, O.MP_DTYPE, NAMES_O, e.nLevels)
    fvo = O.make_frame_view(kp, desc, grid[0], grid[1], 0.0, 0.0, 752.0, 480.0, e.scaleFactors)
    n_ref, out_ref = O.search_by_projection(fvo, mps, mpd, init_obs, th, nn)
    m = orbfe.ORBmatcher(ex)
    fv = orbfe.make_frame_view(kp, desc, grid[0], grid[1], 0.0, 0.0, 752.0, 480.0, ex.mvScaleFactor)
    n, out = m.SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, th, False, 0.0, nn, init_obs)
    assert n == n_ref and np.array_equal(out, out_ref)
    assert n_ref > M // 10
    # far-points filter + no initial claims
    n_ref2, out_ref2 = O.search_by_projection(fvo, mps, mpd, None, th, nn, True, 12.0)
    n2, out2 = m.SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, th, True, 12.0, nn, None)
    assert n2 == n_ref2 and np.array_equal(out2, out_ref2)


def test_projection_adversarial_claim_chain(built):
    """Many map points fighting for the same few keypoints: long greedy dependency chains."""
    orbfe, ex, e, kp, desc = _setup(320, 240, 300, 4, 7)
    rng = np.random.default_rng(0)
    M = 600
    mps = np.zeros(M, O.MP_DTYPE)
    src = rng.integers(0, min(len(kp), 12), M)  # only 12 distinct sources
    mpd = np.stack([S.flip_bits(desc[s], int(rng.integers(0, 6)), rng) for s in src])
    mps["projX"] = kp["x"][src] + rng.uniform(-2, 2, M).astype(np.float32)
    mps["projY"] = kp["y"][src] + rng.uniform(-2, 2, M).astype(np.float32)
    mps["viewCos"] = 1.0
    mps["level"] = kp["octave"][src]
    mps["inView"] = 1
    mps["observations"] = rng.integers(0, 3, M)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 320.0, 240.0, e.scaleFactors)
    n_ref, out_ref = O.search_by_projection(fvo, mps, mpd, None, 20.0, 0.85)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 320.0, 240.0, ex.mvScaleFactor)
    n, out = orbfe.ORBmatcher(ex).SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, 20.0, False, 0.0, 0.85, None)
    assert n == n_ref and np.array_equal(out, out_ref)


@pytest.mark.parametrize("B,M", [(3, 800), (24, 1500), (130, 700)])  # small launch: wave top-K + 1024-thread resolve; large: thread top-K; >= 128 frames: 256-thread resolve
def test_projection_batch_device_equals_host_api(built, B, M):
    import torch
    orbfe, ex, e, kp0, desc0 = _setup()
    if B > 4:
        ex = orbfe.ORBextractor(1000, 40000, 1.2, 8, 20, 7, 752, 480, device=0, max_batch=B)
    ims = [synth.frame(752, 480, 20 + b) for b in range(B)]
    res = ex.extract_batch(ims)
    cap = ex.cap
    dev = torch.device("cuda", 0)
    kp_all = np.zeros((B, cap), orbfe.KP_DTYPE)
    desc_all = np.zeros((B, cap, 32), np.uint8)
    n_all = np.zeros(B, np.int32)
    mps_all = np.zeros((B, M), orbfe.MP_DTYPE)
    mpd_all = np.zeros((B, M, 32), np.uint8)
    obs_all = np.full((B, cap), -1, np.int32)
    refs = []
    for b in range(B):
        kp, desc, _ = res[b]
        n_all[b] = len(kp)
        kp_all[b, :len(kp)] = kp
        desc_all[b, :len(kp)] = desc
        mps, mpd, init_obs = S.projection_scenario(kp, desc, M, 100 + b, O.MP_DTYPE, NAMES_O, e.nLevels)
        mps_all[b] = mps.view(orbfe.MP_DTYPE)
        mpd_all[b] = mpd
        obs_all[b, :len(kp)] = init_obs
        fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 752.0, 480.0, e.scaleFactors)
        refs.append(O.search_by_projection(fvo, mps, mpd, init_obs, 20.0, 0.85))
    t = lambda a: torch.from_numpy(a.view(np.uint8).reshape(-1)).to(dev)
    d_kp, d_desc, d_n, d_mps, d_mpd, d_obs = t(kp_all), t(desc_all), t(n_all), t(mps_all), t(mpd_all), t(obs_all)
    d_out = torch.zeros(B * cap, dtype=torch.int32, device=dev)
    d_nm = torch.zeros(B, dtype=torch.int32, device=dev)
    orbfe.ORBmatcher(ex).SearchByProjection_batch_device(
        B, d_kp.data_ptr(), d_desc.data_ptr(), d_n.data_ptr(), cap, 64, 48, 0.0, 0.0, 752.0, 480.0, M, d_mps.data_ptr(),
        d_mpd.data_ptr(), d_obs.data_ptr(), 20.0, 0.85, d_out.data_ptr(), d_nm.data_ptr(),
        stream=torch.cuda.current_stream(dev).cuda_stream)
    torch.cuda.synchronize(dev)
    out = d_out.cpu().numpy().reshape(B, cap)
    nm = d_nm.cpu().numpy()
    for b in range(B):
        assert nm[b] == refs[b][0]
        assert np.array_equal(out[b, :n_all[b]], refs[b][1])


@pytest.mark.parametrize("orient,nodes,seed", [(True, 120, 1), (False, 120, 2), (True, 5, 3)])
def test_search_by_bow_exact(built, orient, nodes, seed):
    import orbfe
    W, H = 752, 480
    args = (1000, 40000, 1.2, 8, 20, 7, W, H)
    e = O.Extractor(*args)
    frames = list(synth.stream(W, H, 2, index0=3))
    kpk, dk, _ = e.extract(frames[0])
    kpf, df, _ = e.extract(frames[1])
    kfOff, kfIdx, fOff, fIdx, has = S.bow_scenario(kpk, dk, kpf, df, nodes, seed)
    n_ref, out_ref = O.search_by_bow(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    n, out = orbfe.ORBmatcher(ex).SearchByBoW(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient)
    assert n == n_ref and np.array_equal(out, out_ref)
    assert n_ref > 20


@pytest.mark.parametrize("orient,nodes,seed,left_frac", [(True, 120, 7, 0.5), (False, 60, 8, 0.75), (True, 5, 9, 0.25)])
def test_search_by_bow_rig_exact(built, orient, nodes, seed, left_frac):
    """two-camera frame (F->Nleft != -1), src/ORBmatcher.cc:205-233, 263-286"""
    import orbfe
    W, H = 752, 480
    args = (1000, 40000, 1.2, 8, 20, 7, W, H)
    e = O.Extractor(*args)
    frames = list(synth.stream(W, H, 2, index0=13))
    kpk, dk, _ = e.extract(frames[0])
    kpf, df, _ = e.extract(frames[1])
    kfOff, kfIdx, fOff, fIdx, has = S.bow_scenario(kpk, dk, kpf, df, nodes, seed)
    nLeft = int(len(df) * left_frac)
    n_ref, out_ref = O.search_by_bow(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient, nLeft=nLeft)
    ex = orbfe.ORBextractor(*args, device=0, max_batch=1)
    n, out = orbfe.ORBmatcher(ex).SearchByBoW(kfOff, kfIdx, fOff, fIdx, dk, kpk["angle"], has, df, kpf["angle"], 0.75, orient,
                                             nLeft=nLeft)
    assert n == n_ref and np.array_equal(out, out_ref)
    assert n_ref > 10


def test_descriptor_distance_host(built):
    import orbfe
    rng = np.random.default_rng(5)
    for _ in range(50):
        a = rng.integers(0, 256, 32, dtype=np.uint8)
        b = rng.integers(0, 256, 32, dtype=np.uint8)
        assert orbfe.ORBmatcher.DescriptorDistance(a, b) == O.hamming(a, b)


@pytest.mark.gpu
@pytest.mark.parametrize("th,nn,M,seed", [(20.0, 0.85, 3000, 11), (6.0, 0.9, 1500, 12), (10.0, 0.85, 33000, 13)])  # the last: >= 128 top-K blocks
def test_projection_large_frame_paths(built, th, nn, M, seed):
    """Frames with more than 2048 keypoints take the other code paths: global-memory sort of the visit order,
    top-K segments that do not fit the LDS tile, claim table in global memory."""
    orbfe, ex, e, kp, desc = _setup(1280, 720, 5000, 8, 3)
    assert len(kp) > 2048
    mps, mpd, init_obs = S.projection_scenario(kp, desc, M, seed, O.MP_DTYPE, NAMES_O, e.nLevels)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 1280.0, 720.0, e.scaleFactors)
    n_ref, out_ref = O.search_by_projection(fvo, mps, mpd, init_obs, th, nn)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 1280.0, 720.0, ex.mvScaleFactor)
    n, out = orbfe.ORBmatcher(ex).SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, th, False, 0.0, nn, init_obs)
    assert n == n_ref and np.array_equal(out, out_ref)
    assert n_ref > M // 10


@pytest.mark.gpu
@pytest.mark.parametrize("th,M,reloc", [(40.0, 600, False), (14.0, 900, False), (60.0, 500, True), (25.0, 40000, True)])
def test_projection_many_survivors(built, th, M, reloc):
    """Every keypoint descriptor is a near copy of one pattern, so (almost) every keypoint inside a window survives the
    distance cut-off: hundreds of survivors per map point.  Exercises the list-overflow reduction of the
    wave-per-map-point top-K pass (> 64 and > 128 survivors), long claim chains and the exact rescans of the
    resolve pass; the last case takes the thread-per-map-point kernel (>= 128 blocks)."""
    import frustum_scenarios as FS
    from test_frustum import ON, PN
    orbfe, ex, e, kp, desc = _setup()
    rng = np.random.default_rng(5)
    base = rng.integers(0, 256, 32, dtype=np.uint8)
    desc = np.stack([S.flip_bits(base, int(rng.integers(0, 14)), rng) for _ in range(len(kp))])
    m = orbfe.ORBmatcher(ex)
    fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 752.0, 480.0, e.scaleFactors)
    fv = orbfe.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, 752.0, 480.0, ex.mvScaleFactor)
    if not reloc:
        mps, mpd, init_obs = S.projection_scenario(kp, desc, M, 3, O.MP_DTYPE, NAMES_O, e.nLevels, jitter=6.0, max_flip=10)
        n_ref, out_ref = O.search_by_projection(fvo, mps, mpd, init_obs, th, 0.95)
        n, out = m.SearchByProjection(fv, mps.view(orbfe.MP_DTYPE), mpd, th, False, 0.0, 0.95, init_obs)
        assert n == n_ref and np.array_equal(out, out_ref)
        assert n_ref > 50
    else:
        import test_sim3_reloc as T3
        Fo, Fp = O.Frustum(), orbfe.Frustum()
        v = FS.fill_frustum(Fo, ON, seed=77)
        FS.fill_frustum(Fp, PN, seed=77)
        pts, mpd, ang, has = T3.reloc_scenario(kp, desc, e.scaleFactors, v, M, 9)
        n_ref, out_ref = O.search_by_projection_kf(fvo, Fo, pts, mpd, ang, has, th, True)
        n, out = m.SearchByProjection_keyframe(fv, Fp, pts.view(orbfe.WP_DTYPE), mpd, ang, has, th, True)
        assert n == n_ref and np.array_equal(out, out_ref)
        assert n_ref > 50
