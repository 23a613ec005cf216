"""Synthetic matcher inputs: map points derived from a frame's own keypoints."""
import numpy as np


def flip_bits(desc, nflip, rng):
    d = desc.copy()
    if nflip:
        bits = rng.choice(256, size=nflip, replace=False)
        for b in bits:
            d[b >> 3] ^= np.uint8(1 << (b & 7))
    return d


def projection_scenario(kp, desc, M, seed, mp_dtype, names, n_levels, jitter=3.0, max_flip=20, claimed_frac=0.05,
                        dup_frac=0.3):
    """M map points: descriptor = a random keypoint's descriptor with 0..max_flip bit flips, projected
    position = that keypoint +- jitter px, level = its octave (or octave+1 -> searches [lvl-1, lvl]).
    dup_frac of the points reuse the SAME source keypoint as an earlier one (exercises the greedy claims)."""
    rng = np.random.default_rng(seed)
    n = len(kp)
    mps = np.zeros(M, mp_dtype)
    mpd = np.zeros((M, 32), np.uint8)
    src = rng.integers(0, n, M)
    ndup = int(dup_frac * M)
    if ndup and M > 1:
        tgt = rng.integers(1, M, ndup)
        src[tgt] = src[rng.integers(0, np.maximum(tgt, 1))]
    fx, fy, fcos, fdepth, flevel, fin, fbad, fobs = names
    for i in range(M):
        k = kp[src[i]]
        mpd[i] = flip_bits(desc[src[i]], int(rng.integers(0, max_flip + 1)), rng)
        mps[fx][i] = np.float32(k["x"] + rng.uniform(-jitter, jitter))
        mps[fy][i] = np.float32(k["y"] + rng.uniform(-jitter, jitter))
        mps[fcos][i] = 1.0 if rng.random() < 0.8 else 0.9
        mps[fdepth][i] = np.float32(rng.uniform(0.5, 30.0))
        lvl = int(k["octave"]) + int(rng.random() < 0.3)
        mps[flevel][i] = min(lvl, n_levels - 1)
        mps[fin][i] = int(rng.random() < 0.93)
        mps[fbad][i] = int(rng.random() < 0.03)
        mps[fobs][i] = int(rng.integers(0, 4))
    init_obs = np.full(n, -1, np.int32)
    m = rng.random(n) < claimed_frac
    init_obs[m] = rng.integers(0, 3, int(m.sum()))
    return mps, mpd, init_obs


def bow_scenario(kp_kf, desc_kf, kp_f, desc_f, n_nodes, seed):
    """Assign features to `n_nodes` vocabulary nodes by a descriptor hash (so similar descriptors often
    share a node), build the merge-walked CSR groups in ascending node id with ascending feature indices."""
    rng = np.random.default_rng(seed)

    def node_of(desc):
        return (desc[:, 0].astype(np.int64) * 7 + (desc[:, 5] >> 3)) % n_nodes

    nk, nf = node_of(desc_kf), node_of(desc_f)
    # drop ~5% of features (stopped words are omitted from a FeatureVector)
    keep_k = rng.random(len(nk)) > 0.05
    keep_f = rng.random(len(nf)) > 0.05
    kf_off, kf_idx, f_off, f_idx = [0], [], [0], []
    for node in range(n_nodes):
        a = np.nonzero((nk == node) & keep_k)[0]
        b = np.nonzero((nf == node) & keep_f)[0]
        if len(a) == 0 or len(b) == 0:
            continue  # merge-walk only visits nodes present in both maps
        kf_idx += list(a)
        f_idx += list(b)
        kf_off.append(len(kf_idx))
        f_off.append(len(f_idx))
    has_mp = (rng.random(len(desc_kf)) < 0.8).astype(np.uint8)
    return (np.array(kf_off, np.int32), np.array(kf_idx, np.int32), np.array(f_off, np.int32),
            np.array(f_idx, np.int32), has_mp)
