"""BASELINE configs C2/C3, C4 and C5 at stream length: every frame of a synthetic stream (256 x 752x480, 64 x 1280x720,
64 x 1024x1024 with 12 levels) is extracted on the GPU (batched) and compared bit for bit with the CPU oracle; a quarter of
the frames also go through SearchByProjection.  The two extra streams `bench.py --texture-sweep` times -- 1/f ("pink")
noise, which doubles the FAST candidate population, and the low-texture scene -- run at the headline geometry too, so
every number the bench prints is on a verified workload."""
from concurrent.futures import ThreadPoolExecutor

import numpy as np
import pytest

import oracle_py as O

# case -> (bench.WORKLOADS name, synth stream generator, frames)
CASES = {"euroc_752x480": ("euroc_752x480", "stream", 256), "batched_1280x720": ("batched_1280x720", "stream", 64),
         "tumvi_1024x1024": ("tumvi_1024x1024", "stream", 64),
         "euroc_752x480_pink": ("euroc_752x480", "pink_stream", 128), "euroc_752x480_lowtex": ("euroc_752x480", "lowtex_stream", 128)}


@pytest.mark.gpu
@pytest.mark.parametrize("name", sorted(CASES))
def test_stream_bit_exact(built, name):
    import bench
    import orbfe
    from orbfe import synth
    workload, gen, n_frames = CASES[name]
    ARGS = tuple(bench.WORKLOADS[workload])
    W, H = ARGS[6], ARGS[7]
    frames = list(getattr(synth, gen)(W, H, n_frames, index0=1000))
    ex = orbfe.ORBextractor(*ARGS, device=0, max_batch=64)
    gpu = []
    for i in range(0, len(frames), 64):
        gpu += ex.extract_batch(frames[i:i + 64])

    def ref_chunk(chunk):
        e = O.Extractor(*ARGS)  # one oracle instance per thread
        return [e.extract(f) for f in chunk]

    n_thr = 8
    per = len(frames) // n_thr
    with ThreadPoolExecutor(n_thr) as pool:
        ref = [r for part in pool.map(ref_chunk, [frames[i * per:(i + 1) * per] for i in range(n_thr)]) for r in part]
    assert len(ref) == len(gpu) == n_frames
    total = 0
    for i, ((kp_g, desc_g, per_g), (kp_r, desc_r, per_r)) in enumerate(zip(gpu, ref)):
        assert len(kp_g) == len(kp_r), i
        assert kp_g.tobytes() == kp_r.tobytes(), i
        assert np.array_equal(desc_g, desc_r), i
        assert np.array_equal(per_g, per_r), i
        total += len(kp_r)
    assert total > 0.85 * ARGS[0] * n_frames

    m = orbfe.ORBmatcher(ex)
    e = O.Extractor(*ARGS)
    rng = np.random.default_rng(99)
    for i in range(0, n_frames, 4):
        kp, desc, _ = ref[i]
        mps, mpd = bench.make_map_points(kp.view(orbfe.KP_DTYPE), len(kp), desc, 2000, rng, e.nLevels, orbfe.MP_DTYPE)
        fvo = O.make_frame_view(kp, desc, 64, 48, 0.0, 0.0, float(W), float(H), e.scaleFactors)
        n_ref, out_ref = O.search_by_projection(fvo, mps.view(O.MP_DTYPE), mpd, None, 20.0, 0.85)
        fv = orbfe.make_frame_view(kp.view(orbfe.KP_DTYPE), desc, 64, 48, 0.0, 0.0, float(W), float(H), ex.mvScaleFactor)
        n, out = m.SearchByProjection(fv, mps, mpd, 20.0, False, 0.0, 0.85, None)
        assert n == n_ref and np.array_equal(out, out_ref), i
