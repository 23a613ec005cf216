import sys, time
sys.path.insert(0, "orb_slam3_v1.0_amd/python"); sys.path.insert(0, "tests"); sys.path.insert(0, ".")
import numpy as np, ctypes as C
import oracle_py as O, orbfe
import test_triangulation_batch as TB
from orbfe import synth
W, H = 752, 480
ARGS = (1000, 40000, 1.2, 8, 20, 7, W, H)
eo = O.Extractor(*ARGS); ex = orbfe.ORBextractor(*ARGS)
kp, desc, _ = eo.extract(next(iter(synth.stream(W, H, 1))))
K = 20
node1 = TB.nodes_of(kp)
nbs = [TB.neighbour(kp, desc, 500 + k, True, False) for k in range(K)]
h1 = (np.random.default_rng(1).random(len(kp)) < 0.3).astype(np.uint8)
kf1 = orbfe.KeyFrame(ex, kp.view(orbfe.KP_DTYPE), desc, node1, ex.mvScaleFactor)
kf2 = [orbfe.KeyFrame(ex, nb["kp"].view(orbfe.KP_DTYPE), nb["desc"], nb["node"], ex.mvScaleFactor) for nb in nbs]
prm = [orbfe.tri_params(nb["F12"], nb["ep"], False, False, True) for nb in nbs]
has2 = [nb["has"] for nb in nbs]
# pre-marshalled raw C call
kfp = (C.c_void_p * K)(*[k.h.value for k in kf2]); h2p = (C.c_void_p * K)(*[v.ctypes.data for v in has2]); P = (orbfe.TriParams * K)(*prm)
raw = np.full((K, kf1.n), -1, np.int32); rbin = np.zeros((K, kf1.n), np.uint8)
p = lambda a: a.ctypes.data_as(C.c_void_p)
L = ex.L
def t(fn, reps=300):
    fn(); t0 = time.perf_counter()
    for _ in range(reps): fn()
    return (time.perf_counter() - t0) / reps * 1e3
print("C call alone (pre-marshalled):   %.3f ms" % t(lambda: L.orbfe_match_triangulation_batch(ex.h, kf1.h, p(h1), K, kfp, h2p, P, p(raw), p(rbin))))
print("python wrapper batch call:       %.3f ms" % t(lambda: orbfe.SearchForTriangulation_batch(ex, kf1, h1, kf2, has2, prm)))
out = np.zeros(kf1.n, np.int32); n = C.c_int()
print("20 x select (raw C, pre-marsh.): %.3f ms" % t(lambda: [L.orbfe_triangulation_select(kf1.n, raw[k].ctypes.data, rbin[k].ctypes.data, p(h1), 1, p(out), C.byref(n)) for k in range(K)]))
print("20 x python triangulation_select: %.3f ms" % t(lambda: [orbfe.triangulation_select(raw[k], rbin[k], h1, True) for k in range(K)]))
for KK in (5, 10, 30):
    kk = (C.c_void_p * KK)(*[kf2[i % K].h.value for i in range(KK)]); hh = (C.c_void_p * KK)(*[has2[i % K].ctypes.data for i in range(KK)])
    PP = (orbfe.TriParams * KK)(*[prm[i % K] for i in range(KK)]); r2 = np.zeros((KK, kf1.n), np.int32); b2 = np.zeros((KK, kf1.n), np.uint8)
    print("C call alone K=%d: %.3f ms" % (KK, t(lambda: L.orbfe_match_triangulation_batch(ex.h, kf1.h, p(h1), KK, kk, hh, PP, p(r2), p(b2)))))
