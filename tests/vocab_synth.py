"""Synthetic DBoW2 vocabulary trees for the f4 parity tests (no ORBvoc.txt ships with the reference checkout
that travels to the GPU box; the tree SHAPE -- k children per node, depth L, early leaves, duplicate cluster
centres that force first-minimum tie-breaks -- is what the descent has to get right)."""
import numpy as np


def make_tree(k=10, L=4, seed=0, early_leaf_p=0.08, dup_p=0.1, stop_p=0.05):
    """Breadth-first ids like loadFromTextFile produces are NOT assumed by the kernels, so ids are assigned
    depth-first here (children of one node are not contiguous)."""
    rng = np.random.default_rng(seed)
    parent, depth, desc = [0], [0], [np.zeros(32, np.uint8)]
    kids = [[]]

    def grow(pid, d, centre):
        nk = k if d > 0 else k
        prev = None
        for j in range(nk):
            nid = len(parent)
            parent.append(pid)
            depth.append(d + 1)
            if prev is not None and rng.random() < dup_p:
                c = prev.copy()                      # exact duplicate -> tie between siblings
            else:
                flips = rng.integers(0, 256, size=max(2, 96 >> d))
                c = centre.copy()
                bits = np.unpackbits(c)
                bits[flips] ^= 1
                c = np.packbits(bits)
            prev = c
            desc.append(c)
            kids.append([])
            kids[pid].append(nid)
            if d + 1 < L and not (d + 1 >= 1 and rng.random() < early_leaf_p):
                grow(nid, d + 1, c)

    grow(0, 0, rng.integers(0, 256, 32, dtype=np.uint8))
    n = len(parent)
    childOff = np.zeros(n + 1, np.int32)
    childOff[1:] = np.cumsum([len(c) for c in kids])
    childIdx = np.array([c for cs in kids for c in cs], np.int32)
    wordId = np.zeros(n, np.int32)
    weight = np.zeros(n, np.float64)
    nw = 0
    for i in range(1, n):
        if not kids[i]:
            wordId[i] = nw
            nw += 1
            weight[i] = 0.0 if rng.random() < stop_p else rng.uniform(0.1, 9.0)
    return dict(k=k, L=L, childOff=childOff, childIdx=childIdx, nodeDesc=np.stack(desc), wordId=wordId,
                weight=weight, parent=np.array(parent, np.int32), depth=np.array(depth, np.int32), nWords=nw)


def spread_first_level(t, seed):
    """moves the first-level subtrees far apart (each XORed with its own random pattern; deeper centres keep their offsets
    relative to their first-level ancestor), so that the descriptors of an IMAGE -- bits close to uniform -- fall into many
    nodes, as they do in ORBvoc.txt.  In place; returns the tree."""
    rng = np.random.default_rng(seed)
    d = t["nodeDesc"]
    for c in np.flatnonzero(t["depth"] == 1):
        delta = rng.integers(0, 256, 32, dtype=np.uint8)
        stack = [int(c)]
        while stack:
            i = stack.pop()
            d[i] ^= delta
            stack += [int(x) for x in t["childIdx"][t["childOff"][i]:t["childOff"][i + 1]]]
    return t


def features_near(tree, n, seed=1, noise_bits=20):
    """Descriptors near random tree nodes (so descents spread over the tree) plus pure noise."""
    rng = np.random.default_rng(seed)
    nd = tree["nodeDesc"]
    src = rng.integers(1, len(nd), n)
    out = nd[src].copy()
    bits = np.unpackbits(out, axis=1)
    for i in range(n):
        nb = rng.integers(0, noise_bits + 1)
        bits[i, rng.integers(0, 256, nb)] ^= 1
    out = np.packbits(bits, axis=1)
    out[:: 17] = rng.integers(0, 256, out[::17].shape, dtype=np.uint8)
    return out


def write_text(tree, path, scoring=0, weighting=0):
    """ORBvoc.txt layout (TemplatedVocabulary::saveToTextFile): nodes in id order need parents before children;
    make_tree's depth-first ids satisfy that."""
    with open(path, "w") as f:
        f.write("%d %d %d %d\n" % (tree["k"], tree["L"], scoring, weighting))
        n = len(tree["parent"])
        for i in range(1, n):
            leaf = tree["childOff"][i + 1] == tree["childOff"][i]
            f.write("%d %d %s %r\n" % (tree["parent"][i], 1 if leaf else 0,
                                       " ".join(str(int(v)) for v in tree["nodeDesc"][i]), float(tree["weight"][i])))
