"""CPU tests of the boundary: the shared library loads, exports every symbol that
include/sdr_amd.h declares, validates inputs before touching a device, and its synthetic-row
generator equals the independent numpy statement.  No kernel is launched here."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import sdr_amd
from sdr_amd import _lib
import util

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def declared_functions():
    text = open(os.path.join(ROOT, "include", "sdr_amd.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    return sorted(set(re.findall(r"\b(sdrb_[a-z0-9_]+)\s*\(", text)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = declared_functions()
    assert len(names) >= 15
    for name in names:
        assert hasattr(lib, name), f"{name} declared in include/sdr_amd.h but not exported"
    assert set(names) == set(_lib.PROTOTYPES), "python prototypes and header disagree"


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libsdr_amd.so")
    with pytest.raises(ImportError):
        _lib.load()


def test_synth_generator_matches_numpy_statement():
    for width, active, wd in ((2048, 40, False), (16384, 328, True), (64, 64, True), (10, 0, False)):
        off, ids, data = sdr_amd.synth_csr(1, 5, 33, width, active, wd)
        off2, ids2, data2 = util.synth_rows_numpy(1, 5, 33, width, active, wd)
        assert np.array_equal(off, off2) and np.array_equal(ids, ids2)
        if wd:
            assert np.array_equal(data.view(np.uint32), data2.view(np.uint32))
        rows = ids.reshape(33, active) if active else ids.reshape(33, 0)
        assert (np.diff(rows, axis=1) > 0).all() and (rows >= 0).all() and (rows < width).all()
    # windows of one dataset agree with each other
    a = sdr_amd.synth_csr(2, 0, 64, 2048, 40)[1].reshape(64, 40)
    b = sdr_amd.synth_csr(2, 32, 32, 2048, 40)[1].reshape(32, 40)
    assert np.array_equal(a[32:], b)


def _upload_rc(width, kind, off, ids, data=None):
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.sdrb_batch_create(0, width, kind, C.byref(h)) == 0
    off = np.asarray(off, dtype=np.uint64)
    ids = np.asarray(ids, dtype=np.int32)
    dp = None
    if data is not None:
        data = np.asarray(data, dtype=np.float32)
        dp = data.ctypes.data_as(C.c_void_p)
    rc = lib.sdrb_batch_upload_csr(h, off.size - 1, off.ctypes.data_as(C.c_void_p),
                                   ids.ctypes.data_as(C.c_void_p), dp, None)
    msg = lib.sdrb_last_error().decode()
    lib.sdrb_batch_destroy(h)
    return rc, msg


def test_upload_validation_happens_before_any_device_work():
    # unsorted and duplicate ids (the reference asserts ascending order, SDR.hpp:679,1621)
    assert _upload_rc(0, 0, [0, 3], [1, 3, 2])[0] == -2
    assert _upload_rc(0, 0, [0, 2], [5, 5])[0] == -2
    # declared width violated
    assert _upload_rc(8, 0, [0, 2], [1, 8])[0] == -3
    assert _upload_rc(8, 0, [0, 2], [-1, 3])[0] == -3
    # payload outside [0,1] (UnitData.hpp:15-17) and NaN
    assert _upload_rc(0, 1, [0, 1], [1], [1.5])[0] == -7
    assert _upload_rc(0, 1, [0, 1], [1], [float("nan")])[0] == -7
    # broken offsets
    assert _upload_rc(0, 0, [1, 2], [1, 2])[0] == -1
    assert _upload_rc(0, 0, [0, 2, 1], [1, 2])[0] == -1
    rc, msg = _upload_rc(0, 1, [0, 1], [1], None)
    assert rc == -1 and "payload" in msg


def test_create_rejects_bad_arguments():
    lib = _lib.load()
    h = C.c_void_p()
    assert lib.sdrb_batch_create(0, -1, 0, C.byref(h)) == -1
    assert lib.sdrb_batch_create(0, 0, 9, C.byref(h)) == -1
    assert lib.sdrb_batch_create(-1, 0, 0, C.byref(h)) == -1
    assert lib.sdrb_batch_destroy(None) == 0


def _check_rc(off, ids, data, width, kind):
    lib = _lib.load()
    rc = lib.sdrb_csr_check(off.size - 1, off.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                            data.ctypes.data_as(C.c_void_p) if data is not None else None, width, kind)
    return rc, lib.sdrb_last_error().decode()


def test_large_batches_are_checked_by_the_thread_team_like_small_ones(monkeypatch):
    """sdrb_csr_check runs the checks of an upload without a device.  Arrays of a few megabytes and more go
    through the team of host threads (pieces of the element arrays handed out in order); the first finding --
    offsets, then order, then width, then payloads -- must come out whatever the number of threads."""
    width, active, n = 2048, 40, 300_000      # 48 MB of ids: a dozen pieces
    off, ids, data = sdr_amd.synth_csr(7, 0, n, width, active, True)
    for threads in ("1", "3", "8"):
        monkeypatch.setenv("SDRB_HOST_THREADS", threads)
        assert _check_rc(off, ids, data, width, 1)[0] == 0
        assert _check_rc(off, ids, None, width, 0)[0] == 0
        # a descent inside one row, far into the arrays; an equal neighbour elsewhere, later
        bad = ids.copy()
        r = 217_123
        bad[r * active + 7] = bad[r * active + 6]
        bad[(r + 5000) * active + 1] = bad[(r + 5000) * active]
        rc, msg = _check_rc(off, bad, data, width, 1)
        assert rc == -2 and f"row {r} " in msg and "element 7" in msg, msg
        # the last id of a row above the first of the next is fine; above the width it is not
        bad = ids.copy()
        bad[(r + 1) * active - 1] = width + 5
        rc, msg = _check_rc(off, bad, data, width, 1)
        assert rc == -3 and str(width + 5) in msg, msg
        assert _check_rc(off, bad, data, 0, 1)[0] == 0
        bad = ids.copy()
        bad[0] = -3
        assert _check_rc(off, bad, None, width, 0)[0] == -3
        assert _check_rc(off, bad, None, 0, 0)[0] == 0
        # payloads: the first bad one is named; order findings come first
        pd = data.copy()
        pd[9_000_017] = 1.25
        pd[11_000_000] = np.nan
        rc, msg = _check_rc(off, ids, pd, width, 1)
        assert rc == -7 and "9000017" in msg, msg
        assert _check_rc(off, ids, pd, width, 2)[0] == -7          # ArithData: the NaN
        pd[11_000_000] = 3.0
        assert _check_rc(off, ids, pd, width, 2)[0] == 0           # ArithData takes any finite value
        bad = ids.copy()
        bad[11_500_001] = bad[11_500_000]                          # the first two ids of a row
        rc, msg = _check_rc(off, bad, pd, width, 1)
        assert rc == -2, msg
        # offsets: not monotone, beyond the arrays
        boff = off.copy()
        boff[123_456] = boff[123_457] + 1
        rc, msg = _check_rc(boff, ids, data, width, 1)
        assert rc == -1 and "row 123456" in msg, msg
        boff = off.copy()
        boff[n - 1] = off[n] + 40
        assert _check_rc(boff, ids, data, width, 1)[0] == -1
    # ragged rows with empty ones between: row starts are where descents are allowed
    rng = np.random.default_rng(3)
    lens = rng.integers(0, 90, size=200_000)
    lens[rng.integers(0, lens.size, size=20_000)] = 0
    roff = np.zeros(lens.size + 1, dtype=np.uint64)
    roff[1:] = np.cumsum(lens)
    steps = rng.integers(1, 40, size=int(roff[-1])).astype(np.int64)
    flat = np.cumsum(steps)
    starts = np.repeat(flat[np.minimum(roff[:-1], flat.size - 1).astype(np.int64)] - 1, lens)
    rids = (flat - starts).astype(np.int32)                         # every row restarts near 1 and ascends
    assert _check_rc(roff, rids, None, 0, 0)[0] == 0
    row = 150_000 + int(np.argmax(lens[150_000:] >= 2))             # a row with a second element to spoil
    first = int(roff[row])
    rids2 = rids.copy()
    rids2[first + 1] = rids2[first]
    rc, msg = _check_rc(roff, rids2, None, 0, 0)
    assert rc == -2 and f"row {row} " in msg, msg
