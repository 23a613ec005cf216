"""The pipelined host-pointer API (orbfe_stream_*) and the cross-stream ordering of the handle's scratch.

The ring must hand back, for every frame, exactly the bytes the batched device path produces (which
tests/test_extract_gpu.py / test_stream_gpu.py compare with the oracle), whatever the source memory is: pageable numpy
rows, pinned packed blocks, pinned frames with a padded pitch, partial last submissions, more submissions than slots."""
import numpy as np
import pytest

ARGS = (600, 24000, 1.2, 6, 20, 7, 376, 240)


def _reference(ex, frames):
    out = []
    B = ex.max_batch
    for i in range(0, len(frames), B):
        out += ex.extract_batch(list(frames[i:i + B]))
    return out


def _same(got, ref):
    assert len(got) == len(ref)
    for i, ((kp_g, desc_g, per_g), (kp_r, desc_r, per_r)) in enumerate(zip(got, ref)):
        assert len(kp_g) == len(kp_r) and len(kp_r) > 0, i
        assert kp_g.tobytes() == kp_r.tobytes(), i
        assert np.array_equal(desc_g, desc_r), i
        assert np.array_equal(per_g, per_r), i


@pytest.mark.gpu
def test_stream_ring_matches_batched_path(built):
    import torch
    import orbfe
    from orbfe import synth
    W, H = ARGS[6], ARGS[7]
    n_frames, slot = 70, 16
    frames = np.stack(list(synth.stream(W, H, n_frames, index0=77)))
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=slot)
    ref = _reference(ex, frames)

    # the oracle agrees on a few frames (the rest of the chain is covered by test_stream_gpu.py)
    import oracle_py as O
    e = O.Extractor(*ARGS)
    for i in (0, 33, 69):
        kp_r, desc_r, _ = e.extract(frames[i])
        assert ref[i][0].tobytes() == kp_r.tobytes() and np.array_equal(ref[i][1], desc_r)

    pinned = torch.from_numpy(frames).pin_memory()
    padded = torch.zeros((n_frames, H, W + 8), dtype=torch.uint8).pin_memory()  # pitch W + 8 (4-aligned), pinned
    padded[:, :, :W] = torch.from_numpy(frames)
    odd = np.zeros((n_frames, H, W + 2), np.uint8)   # pitch W + 2: not dword-aligned -> re-pitched row by row
    odd[:, :, :W] = frames
    sources = {
        "pageable": (frames, W),
        "pageable_odd_pitch": (odd, W + 2),
        "pinned_packed": (pinned.numpy(), W),
        "pinned_padded_pitch": (padded.numpy(), W + 8),
    }
    for name, (src, pitch) in sources.items():
        st = ex.stream(slots=3, slot_frames=slot)
        got, pos = [], 0
        # keep the ring full: submit until it refuses, then collect one
        while pos < n_frames or st.in_flight():
            while pos < n_frames:
                n = min(slot, n_frames - pos)
                if not st.submit(src[pos:pos + n], pitch=pitch):
                    break
                pos += n
            assert st.in_flight() <= 3
            got += st.collect()
        _same(got, ref)
        st.close()


@pytest.mark.gpu
def test_stream_refuses_when_full_and_empty(built):
    import orbfe
    from orbfe import synth
    W, H = ARGS[6], ARGS[7]
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=4)
    st = ex.stream(slots=2, slot_frames=4)
    fr = np.stack(list(synth.stream(W, H, 4)))
    with pytest.raises(orbfe.OrbfeError):
        st.collect()                 # nothing in flight
    assert st.submit(fr) and st.submit(fr)
    assert not st.submit(fr)         # ORBFE_ERR_BUSY
    a = st.collect()
    assert st.submit(fr[:2])         # partial submission
    b = st.collect()
    c = st.collect()
    _same(b, a)
    _same(c, a[:2])
    with pytest.raises(orbfe.OrbfeError):
        st.submit(fr[:0])
    st.close()


@pytest.mark.gpu
def test_scratch_is_ordered_across_streams(built):
    """a *_device call on stream A followed by calls on the handle's own stream (and back) must not
    corrupt each other although they share the extraction scratch and the matcher arena."""
    import torch
    import bench
    import orbfe
    from orbfe import synth
    W, H = ARGS[6], ARGS[7]
    B = 24
    frames = np.stack(list(synth.stream(W, H, B, index0=5)))
    other = np.stack(list(synth.stream(W, H, B, index0=500)))
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=B)
    m = orbfe.ORBmatcher(ex)
    ref = _reference(ex, frames)
    ref_other = _reference(ex, other)
    dev = torch.device("cuda", 0)
    cap = ex.cap
    sA, sB = torch.cuda.Stream(dev), torch.cuda.Stream(dev)

    def bufs():
        return (torch.zeros((B, cap, 24), dtype=torch.uint8, device=dev), torch.zeros((B, cap, 32), dtype=torch.uint8, device=dev),
                torch.zeros(B, dtype=torch.int32, device=dev), torch.zeros((B, ex.nlevels), dtype=torch.int32, device=dev))

    d1, d2 = torch.from_numpy(frames).to(dev), torch.from_numpy(other).to(dev)
    rng = np.random.default_rng(3)
    kp0, desc0, _ = ref[0]
    mps, mpd = bench.make_map_points(kp0.view(orbfe.KP_DTYPE), len(kp0), desc0, 800, rng, ex.nlevels, orbfe.MP_DTYPE)
    fv = orbfe.make_frame_view(kp0.view(orbfe.KP_DTYPE), desc0, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
    n_ref, out_ref = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
    d_mps = torch.from_numpy(np.tile(mps.view(np.uint8).reshape(1, -1), (B, 1))).to(dev)
    d_mpd = torch.from_numpy(np.tile(mpd.reshape(1, -1), (B, 1))).to(dev)
    torch.cuda.synchronize()
    o1, o2 = bufs(), bufs()
    match = torch.full((B, cap), -1, dtype=torch.int32, device=dev)
    nmatch = torch.zeros(B, dtype=torch.int32, device=dev)
    for rep in range(6):
        for t in (*o1, *o2, nmatch):
            t.zero_()
        match.fill_(-1)
        torch.cuda.synchronize()  # torch's fills run on ITS stream: finished before the side streams touch the buffers
        # extraction of two different batches on two caller streams back to back, no host sync in between
        ex.extract_batch_device(d1.data_ptr(), W * H, W, B, o1[0].data_ptr(), o1[1].data_ptr(), o1[2].data_ptr(), o1[3].data_ptr(), sA.cuda_stream)
        ex.extract_batch_device(d2.data_ptr(), W * H, W, B, o2[0].data_ptr(), o2[1].data_ptr(), o2[2].data_ptr(), o2[3].data_ptr(), sB.cuda_stream)
        # batched matcher on stream A (after A's extraction), then a host matcher call on the handle's stream
        m.SearchByProjection_batch_device(B, o1[0].data_ptr(), o1[1].data_ptr(), o1[2].data_ptr(), cap, 64, 48, 0.0, 0.0, float(W), float(H),
                                          800, d_mps.data_ptr(), d_mpd.data_ptr(), None, 20.0, 0.85, match.data_ptr(), nmatch.data_ptr(),
                                          stream=sA.cuda_stream)
        n_h, out_h = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
        host = ex.extractFeatures(frames[3])  # host-pointer call on the handle's stream while A / B may still run
        torch.cuda.synchronize()
        assert ex.device_status() == 0
        assert n_h == n_ref and np.array_equal(out_h, out_ref)
        assert host[0].tobytes() == ref[3][0].tobytes() and np.array_equal(host[1], ref[3][1])
        for o, r in ((o1, ref), (o2, ref_other)):
            n = o[2].cpu().numpy()
            kp = o[0].cpu().numpy().reshape(B, cap * 24).view(orbfe.KP_DTYPE).reshape(B, cap)
            desc = o[1].cpu().numpy()
            for b in range(B):
                assert n[b] == len(r[b][0]), (rep, b)
                assert kp[b, :n[b]].tobytes() == r[b][0].tobytes(), (rep, b)
                assert np.array_equal(desc[b, :n[b]], r[b][1]), (rep, b)
        # frame 0 of batch 1 against the same map points: the batched matcher must agree with the host call
        assert int(nmatch[0].item()) == n_ref and np.array_equal(match[0, :len(kp0)].cpu().numpy(), out_ref)
