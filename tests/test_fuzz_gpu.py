"""Randomised parity sweep (tests/tools/fuzz_extract.py): random image sizes, pyramid depths, scale factors, thresholds,
budgets and image classes (the default scene or a hostile class of orbfe.synth); the GPU extractor must equal the oracle bit for bit or refuse the configuration with the documented
ORBFE_ERR_UNSUPPORTED (portrait images with round(W/H) == 0: the reference divides by zero there)."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "tools"))


@pytest.mark.gpu
def test_random_configurations_bit_exact(built):
    import fuzz_extract as FZ
    import orbfe
    rng = np.random.default_rng(2)
    done = refused = 0
    kinds = set()
    for k in range(32):
        cfg = FZ.random_config(rng)
        kind = FZ.random_kind(rng)
        try:
            FZ.check(cfg, seed=100 + k, kind=kind)
            done += 1
            kinds.add(kind)
        except orbfe.OrbfeError as err:
            assert err.code == 2 and round(cfg[6] / cfg[7]) == 0, (cfg, str(err))
            refused += 1
    assert done >= 20 and len(kinds) >= 5


@pytest.mark.gpu
def test_random_projection_matching_exact(built):
    """tests/tools/fuzz_match.py: random frame sizes, grids (8x6 ... 257x130), radii, ratios, map-point counts (1 ... 4000),
    initial claims and far-point filters; match indices must equal the oracle's."""
    import fuzz_match as FM
    rng = np.random.default_rng(4)
    ran = sum(FM.one(rng, k) is not None for k in range(16))
    assert ran >= 12


@pytest.mark.gpu
def test_random_batched_projection_matching_exact(built):
    """tests/tools/fuzz_batch_match.py: the batched (throughput) kernels of SearchByProjection on random batches, image
    classes (incl. the hostile look-alike ones), grids, radii, ratios and initial claims; every frame against the oracle."""
    import fuzz_batch_match as FB
    rng = np.random.default_rng(6)
    frames = sum(FB.one(rng, k)[0] for k in range(4))
    assert frames >= 60


@pytest.mark.gpu
def test_random_reference_keyframe_chains_exact(built):
    """tests/tools/fuzz_ref_keyframe.py: orbfe_track_reference_keyframe on random geometries, image classes, vocabulary shapes
    (one node ... hundreds), key frames (same scene / same image / another scene, shuffled, bits flipped, features outside the
    FeatureVector), flags, ratios; every field against the oracle's chain."""
    import fuzz_ref_keyframe as FR
    rng = np.random.default_rng(8)
    tot = [FR.one(rng, k) for k in range(14)]
    assert sum(m for _, m in tot) > 500


@pytest.mark.gpu
def test_random_initialisation_chains_and_resident_fuse_searches_exact(built):
    """tests/tools/fuzz_new_chains.py: orbfe_track_initialization on random geometries, image classes, grids, initial frames
    (previous frame / same image / another scene; shuffled, thinned, upper levels only, empty), windows, ratios; and
    orbfe_fuse_search_keyframe on random key frames, grids, radii, mono / stereo / KannalaBrandt8, map sizes 1 ... 5000 with the
    skip pattern travelling in the ids; every result against the oracle."""
    import fuzz_new_chains as FN
    rng = np.random.default_rng(10)
    ini = [FN.one_init(rng, k) for k in range(14)]
    fus = [FN.one_fuse(rng, k) for k in range(10)]
    assert sum(t[1] for t in ini) > 300 and sum(t[1] for t in fus) > 500
