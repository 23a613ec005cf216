"""MapPoint::ComputeDistinctiveDescriptors (src/MapPoint.cc:343-416) for a batch of descriptor sets."""
import numpy as np
import pytest

import oracle_py as O


def make_sets(seed, sizes):
    rng = np.random.default_rng(seed)
    descs, off = [], [0]
    for n in sizes:
        centre = rng.integers(0, 256, 32, dtype=np.uint8)
        for _ in range(n):
            d = centre.copy()
            bits = np.unpackbits(d)
            bits[rng.integers(0, 256, int(rng.integers(0, 60)))] ^= 1
            d = np.packbits(bits)
            if descs and rng.random() < 0.15 and off[-1] < len(descs):
                d = descs[int(rng.integers(off[-1], len(descs)))].copy()  # exact duplicates -> tied medians
            descs.append(d)
        off.append(len(descs))
    return np.array(off, np.int32), (np.stack(descs) if descs else np.zeros((0, 32), np.uint8))


def py_distinct(off, desc):
    """Plain restatement of :380-409."""
    bi, bm = [], []
    for s in range(len(off) - 1):
        d = desc[off[s]:off[s + 1]]
        N = len(d)
        if N == 0:
            bi.append(-1)
            bm.append(0)
            continue
        D = np.unpackbits(d[:, None, :] ^ d[None, :, :], axis=2).sum(2)
        best, best_i = 2 ** 31 - 1, 0
        for i in range(N):
            med = int(np.sort(D[i])[int(0.5 * (N - 1))])
            if med < best:
                best, best_i = med, i
        bi.append(best_i)
        bm.append(best)
    return np.array(bi), np.array(bm)


SIZES = [0, 1, 2, 3, 4, 5, 7, 10, 33, 64, 65, 150, 0, 2, 2, 9]


def test_oracle_distinct_matches_restatement():
    off, desc = make_sets(1, SIZES)
    bi, bm = O.distinctive_descriptors(off, desc)
    pi, pm = py_distinct(off, desc)
    assert np.array_equal(bi, pi) and np.array_equal(bm, pm)


@pytest.mark.gpu
@pytest.mark.parametrize("seed,sizes", [(2, SIZES), (3, [12] * 3000), (4, [400, 1, 0, 700])])
def test_gpu_distinct_matches_oracle(built, seed, sizes):
    import orbfe
    off, desc = make_sets(seed, sizes)
    ex = orbfe.ORBextractor(500, 2000, 1.2, 4, 20, 7, 320, 240)
    bi, bm = orbfe.ORBmatcher(ex).ComputeDistinctiveDescriptors(off, desc)
    ri, rm = O.distinctive_descriptors(off, desc)
    assert np.array_equal(bi, ri) and np.array_equal(bm, rm)
    b0, _ = orbfe.ORBmatcher(ex).ComputeDistinctiveDescriptors(np.array([0], np.int32), np.zeros((0, 32), np.uint8))
    assert len(b0) == 0
