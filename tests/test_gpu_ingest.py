"""Large uploads: the rows cross the bus through page-locked staging (a team of host threads) and are checked in
HBM.  What arrives must equal what was sent, and every kind of refusal must come out with the code and message the
host-side checks (small batches, sdrb_csr_check) give."""
import ctypes as C

import numpy as np
import pytest
import torch

import oracle
import sdr_amd
from oracle import AND, Batch
from sdr_amd import SDRBatch, _lib

pytestmark = pytest.mark.gpu


def device_arrays(b: SDRBatch):
    wire = b.to_wire().cpu().numpy()
    n, nnz = b.n_rows, b.nnz
    off = wire[: n + 1].view(np.uint32)
    ids = wire[n + 1: n + 1 + nnz]
    data = wire[n + 1 + nnz:].view(np.float32) if b.data_kind else None
    return off, ids, data


@pytest.mark.parametrize("threads", ["1", "5", "16"])
@pytest.mark.parametrize("with_data", [False, True])
def test_large_upload_arrives_intact(monkeypatch, threads, with_data):
    monkeypatch.setenv("SDRB_HOST_THREADS", threads)
    rng = np.random.default_rng(5)
    n = 250_000
    lens = rng.integers(0, 70, size=n)
    lens[rng.integers(0, n, size=n // 10)] = 0
    off = np.zeros(n + 1, dtype=np.uint64)
    off[1:] = np.cumsum(lens)
    nnz = int(off[-1])
    steps = rng.integers(1, 50, size=nnz).astype(np.int64)
    flat = np.cumsum(steps)
    first = np.repeat(flat[np.minimum(off[:-1], nnz - 1).astype(np.int64)] - 1, lens)
    ids = (flat - first).astype(np.int32)                  # every row restarts near 1 and ascends; ids < 70 * 50
    data = rng.random(nnz, dtype=np.float32) if with_data else None
    b = SDRBatch.from_csr(off, ids, data, width=4096)
    d_off, d_ids, d_data = device_arrays(b)
    assert np.array_equal(d_off.astype(np.uint64), off) and np.array_equal(d_ids, ids)
    if with_data:
        assert np.array_equal(d_data.view(np.uint32), data.view(np.uint32))
    st = b.stats()
    assert st["max_row_len"] == int(lens.max()) and st["min_id"] == int(ids.min()) and st["max_id"] == int(ids.max())
    # and it computes: one query against the first rows equals the oracle
    q = Batch(off[:2], ids[: int(off[1])] if off[1] else ids[:0], None)
    if off[1]:
        head = Batch(off[:2001], ids[: int(off[2000])], None)
        want = oracle.port().size_op(AND, q, head)
        got = SDRBatch.from_csr(q.off, q.ids, None, width=4096).ands(b).cpu().numpy().view(np.uint32)[:, :2000]
        assert np.array_equal(got, want)


def _upload(b, off, ids, data):
    lib = _lib.load()
    rc = lib.sdrb_batch_upload_csr(b._h, off.size - 1, off.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                                   data.ctypes.data_as(C.c_void_p) if data is not None else None, None)
    return rc, lib.sdrb_last_error().decode()


def _check(off, ids, data, width, kind):
    lib = _lib.load()
    rc = lib.sdrb_csr_check(off.size - 1, off.ctypes.data_as(C.c_void_p), ids.ctypes.data_as(C.c_void_p),
                            data.ctypes.data_as(C.c_void_p) if data is not None else None, width, kind)
    return rc, lib.sdrb_last_error().decode()


def test_large_refused_uploads_report_like_the_host_checks_and_leave_no_rows():
    width, active, n = 2048, 40, 200_000
    off, ids, data = sdr_amd.synth_csr(7, 0, n, width, active, True)
    cases = []
    bad = ids.copy(); bad[150_001 * active + 7] = bad[150_001 * active + 6]; bad[160_000 * active + 1] = bad[160_000 * active]
    cases.append(("order", off, bad, data, width, 1, -2))
    bad = ids.copy(); bad[(150_001 + 1) * active - 1] = width + 5
    cases.append(("range", off, bad, data, width, 1, -3))
    bad = ids.copy(); bad[0] = -3
    cases.append(("negative", off, bad, None, width, 0, -3))
    pd = data.copy(); pd[6_000_017] = 1.25; pd[7_000_000] = np.nan
    cases.append(("payload", off, ids, pd, width, 1, -7))
    cases.append(("arith nan", off, ids, pd, width, 2, -7))
    boff = off.copy(); boff[123_456] = boff[123_457] + 1
    cases.append(("offsets", boff, ids, data, width, 1, -1))
    boff = off.copy(); boff[77_777] += np.uint64(1) << np.uint64(32)
    cases.append(("offset beyond 32 bits", boff, ids, data, width, 1, -1))
    bad = ids.copy(); bad[150_001 * active + 7] = bad[150_001 * active + 6]
    cases.append(("order before payload", off, bad, pd, width, 1, -2))
    for name, o, i, d, w, kind, code in cases:
        b = SDRBatch(0, w, kind)
        b.upload(off, ids, data if kind else None)          # good rows first
        assert b.n_rows == n
        rc, msg = _upload(b, o, i, d)
        want_rc, want_msg = _check(o, i, d, w, kind)
        assert rc == code == want_rc, (name, rc, want_rc, msg)
        assert msg == want_msg, (name, msg, want_msg)
        b._refresh_info()
        assert b.n_rows == 0 and b.nnz == 0, name           # a refused large batch leaves the handle without rows
        # ... and the handle takes good rows again
        b.upload(off, ids, data if kind else None)
        assert b.n_rows == n
    # ids beyond the width are fine for a batch without a declared width
    bad = ids.copy(); bad[301 * active - 1] = 1 << 30        # the last id of a row: the row still ascends
    b = SDRBatch.from_csr(off, bad, None, width=0)
    assert b.stats()["max_id"] == 1 << 30


def test_large_upload_of_64_bit_ids_and_the_narrowed_twin():
    width, active, n = 2048, 40, 120_000
    off, ids, data = sdr_amd.synth_csr(9, 0, n, width, active, True)
    wide = ids.astype(np.int64)
    b64 = SDRBatch.from_csr(off, wide, data)
    q_off, q_ids, q_data = sdr_amd.synth_csr(10, 0, 3, width, active, True)
    q64 = SDRBatch.from_csr(q_off, q_ids.astype(np.int64), q_data)
    want = oracle.port().size_op(AND, Batch(q_off, q_ids, q_data), Batch(off[:3001], ids[: 3000 * active], data[: 3000 * active]))
    got = q64.ands(b64).cpu().numpy().view(np.uint32)[:, :3000]
    assert np.array_equal(got, want)
    far = wide + (np.int64(1) << 40)                        # no twin: the merge kernels
    bfar = SDRBatch.from_csr(off, far, data)
    qfar = SDRBatch.from_csr(q_off, q_ids.astype(np.int64) + (np.int64(1) << 40), q_data)
    got = qfar.ands(bfar).cpu().numpy().view(np.uint32)[:, :3000]
    assert np.array_equal(got, want)
    bad = far.copy(); bad[99_999 * active + 3] = bad[99_999 * active + 2]
    b = SDRBatch(0, 0, "unit", 64)
    with pytest.raises(_lib.SdrbError) as e:
        b.upload(off, bad, data)
    assert e.value.code == -2 and "row 99999 " in str(e.value)


def test_stream_release_frees_the_scratch_of_a_stream():
    lib = _lib.load()
    s = torch.cuda.Stream()
    off, ids, _ = sdr_amd.synth_csr(1, 0, 2000, 2048, 40)
    q = SDRBatch.from_csr(off[:65], ids[: 64 * 40], None, width=2048)
    d = SDRBatch.from_csr(off, ids, None, width=2048)
    with torch.cuda.stream(s):
        first = q.ands(d, stream=s).cpu().numpy()
    assert lib.sdrb_stream_release(0, C.c_void_p(s.cuda_stream)) == 0
    with torch.cuda.stream(s):
        again = q.ands(d, stream=s).cpu().numpy()
    assert np.array_equal(first, again)
    assert lib.sdrb_stream_release(0, C.c_void_p(s.cuda_stream)) == 0


@pytest.mark.parametrize("with_data", [False, True])
def test_upload_from_page_locked_arrays(with_data):
    """Arrays the caller already holds in page-locked memory are copied without staging: the same rows must arrive, and a
    refusal must still leave the batch empty."""
    off, ids, data = sdr_amd.synth_csr(3, 0, 300_000, 2048, 40, with_data)
    ids_pin = torch.from_numpy(ids).pin_memory().numpy()
    data_pin = torch.from_numpy(data).pin_memory().numpy() if with_data else None
    b = SDRBatch.from_csr(off, ids_pin, data_pin, width=2048)
    d_off, d_ids, d_data = device_arrays(b)
    assert np.array_equal(d_off.astype(np.uint64), off) and np.array_equal(d_ids, ids)
    if with_data:
        assert np.array_equal(d_data.view(np.uint32), data.view(np.uint32))
    bad = ids_pin.copy()
    bad[5_000_001] = bad[5_000_000]              # a repeated id inside a row (rows hold 40 ids: both lie in row 125000)
    bad_pin = torch.from_numpy(bad).pin_memory().numpy()
    with pytest.raises(sdr_amd.SdrbError) as e:
        b.upload(off, bad_pin, data_pin)
    assert e.value.code == -2 and "125000" in str(e.value)
    b._refresh_info()
    assert b.n_rows == 0 and b.nnz == 0
