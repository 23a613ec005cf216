"""The C-ABI library loads without a GPU and exports every symbol include/orbfe.h declares."""
import ctypes as C
import os
import re

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    txt = open(os.path.join(ROOT, "include", "orbfe.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(orbfe_[a-z_0-9]+)\s*\(", txt)))


def test_header_symbols_exported(built):
    import orbfe
    L = orbfe.lib()
    decl = _declared()
    assert len(decl) >= 20
    for name in decl:
        assert hasattr(L, name), "liborbfe.so does not export %s" % name
    assert sorted(orbfe.SYMBOLS) == decl, "python binding and header disagree"


def test_host_only_entry_points(built):
    import orbfe
    L = orbfe.lib()
    assert b"gfx950" in L.orbfe_version()
    assert L.orbfe_status_string(0) == b"ok" and L.orbfe_status_string(3).startswith(b"no usable")
    a = np.arange(32, dtype=np.uint8)
    b = a[::-1].copy()
    assert orbfe.ORBmatcher.DescriptorDistance(a, b) == int(np.unpackbits(a ^ b).sum())
    assert [L.orbfe_stage_name(i) for i in range(5)][-1] == b"total"


def test_create_fails_loudly_without_gpu(built):
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    import orbfe
    with pytest.raises(orbfe.OrbfeError) as ei:
        orbfe.ORBextractor(1000, 16000, 1.2, 8, 20, 7, 752, 480)
    assert ei.value.code == 3  # ORBFE_ERR_NO_DEVICE: no CPU fallback exists


def test_create_rejects_bad_arguments(built):
    import orbfe
    L = orbfe.lib()
    h = C.c_void_p()
    assert L.orbfe_create(None, C.byref(h)) == 1
    bad = orbfe.Params(1000, 16000, 1.2, 0, 20, 7, 752, 480, 0, 1)
    assert L.orbfe_create(C.byref(bad), C.byref(h)) == 1
    bad = orbfe.Params(1000, 16000, 1.2, 8, 5, 7, 752, 480, 0, 1)  # iniTh < minTh: unsupported by the fused pass
    assert L.orbfe_create(C.byref(bad), C.byref(h)) == 2
    assert L.orbfe_max_keypoints(None) == 0 and L.orbfe_get_levels(None) == 0
    # the round-5 entry points refuse a missing handle / object with a status code before anything touches the GPU
    assert L.orbfe_track_initialization(None, None, 752, None, None, 40, 0.45, 1, None, None, None, None, None, None) == 1
    assert L.orbfe_init_frame_create(None, 0, None, None, C.byref(h)) == 1 and L.orbfe_init_frame_size(None) == 0
    L.orbfe_init_frame_destroy(None)
    assert L.orbfe_keyframe_set_grid(None, None, 64, 48, 0.0, 0.0, 0.1, 0.1, None, None) == 1
    assert L.orbfe_fuse_search_keyframe(None, None, None, 0, None, None, 3.0, None, None) == 1
    assert L.orbfe_set_stream_priority(None, 1) == 1 and L.orbfe_debug_clock_probe(None, 20, None, None) == 1


def test_shipped_library_reads_no_environment_and_host_only_entries_work(built):
    """every tuning / diagnostics switch lives in the -DORBFE_DIAG / -DORBFE_ABLATION builds; the
    shipped library does not even import getenv.  And the host-only half of the batched triangulation
    (orbfe_triangulation_select) and the versioned parameter blocks work without a GPU."""
    import subprocess
    import orbfe
    syms = subprocess.check_output(["nm", "-D", "--undefined-only", orbfe.LIB_PATH]).decode()
    assert "getenv" not in syms
    assert orbfe.TriParams().struct_size == C.sizeof(orbfe.TriParams) and orbfe.TrackParams().struct_size == C.sizeof(orbfe.TrackParams)
    raw = np.array([3, -1, 5, 7, 2, 2], np.int32)
    rbin = np.array([1, 0, 1, 9, 1, 1], np.uint8)
    now = np.array([0, 0, 0, 0, 1, 0], np.uint8)
    n, m = orbfe.triangulation_select(raw, rbin, now, True)  # bin 1 holds 3 matches, bin 9 one (>= 0.1 * 3: kept), feature 4 skipped
    assert n == 4 and list(m) == [3, -1, 5, 7, -1, 2]
    n, m = orbfe.triangulation_select(raw, np.array([1, 0, 1, 9, 1, 1] + [], np.uint8), np.zeros(6, np.uint8), False)
    assert n == 5
    with pytest.raises(orbfe.OrbfeError):
        orbfe.triangulation_select(raw, np.full(6, 30, np.uint8), now, True)  # a bin outside the histogram
