"""Scratch memory is private to a stream: size and element ops issued on two streams at once -- from one host
thread and from two -- give what the oracle gives; plus the single-pass offsets scan on long, odd-sized and
unaligned arrays."""
import ctypes as C
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _inputs():
    import sdr_amd
    width, active = 2048, 40
    d_csr = sdr_amd.synth_csr(41, 0, 40_000, width, active)
    qa_csr = sdr_amd.synth_csr(42, 0, 300, width, active)
    qb_csr = sdr_amd.synth_csr(43, 0, 77, width, active)
    return width, d_csr, qa_csr, qb_csr


def test_size_ops_on_two_streams_at_once():
    import torch

    import oracle
    from sdr_amd import SDRBatch

    width, d_csr, qa_csr, qb_csr = _inputs()
    d = SDRBatch.from_csr(*d_csr, width=width)
    qa, qb = SDRBatch.from_csr(*qa_csr, width=width), SDRBatch.from_csr(*qb_csr, width=width)
    od, oa, ob = oracle.Batch(*d_csr), oracle.Batch(*qa_csr), oracle.Batch(*qb_csr)
    want_a = oracle.port().size_op(0, oa, od)
    want_b = oracle.port().size_op(2, ob, od)
    s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
    out_a = torch.empty((qa.n_rows, d.n_rows), dtype=torch.int32, device="cuda:0")
    out_b = torch.empty((qb.n_rows, d.n_rows), dtype=torch.int32, device="cuda:0")
    torch.cuda.synchronize()
    # no synchronisation between the calls: each stream's table build and kernel overlap the other's
    for _ in range(20):
        qa.ands(d, out=out_a, stream=s1)
        qb.xors(d, out=out_b, stream=s2)
    torch.cuda.synchronize()
    assert np.array_equal(out_a.cpu().numpy().view(np.uint32), want_a)
    assert np.array_equal(out_b.cpu().numpy().view(np.uint32), want_b)
    # the popcount kernel builds a derived form of d on the stream of its first call; the other stream waits for it
    d2 = SDRBatch.from_csr(*d_csr, width=width)
    torch.cuda.synchronize()
    qa.ands(d2, out=out_a, stream=s1, algo="popcount")
    qb.xors(d2, out=out_b, stream=s2, algo="popcount")
    torch.cuda.synchronize()
    assert np.array_equal(out_a.cpu().numpy().view(np.uint32), want_a)
    assert np.array_equal(out_b.cpu().numpy().view(np.uint32), want_b)


def test_ops_from_two_host_threads():
    import torch

    import oracle
    from sdr_amd import SDRBatch

    width, d_csr, qa_csr, qb_csr = _inputs()
    d = SDRBatch.from_csr(*d_csr, width=width)
    od = oracle.Batch(*d_csr)
    results, errors = {}, []

    def work(name, q_csr, size_name, elem_name, op):
        try:
            torch.cuda.set_device(0)
            q = SDRBatch.from_csr(*q_csr, width=width)
            stream = torch.cuda.Stream()
            for _ in range(5):
                counts = getattr(q, size_name)(d, stream=stream)
                res = getattr(q, elem_name)(d, stream=stream)
            stream.synchronize()
            results[name] = (counts.cpu().numpy().view(np.uint32), res.to_host(), op, oracle.Batch(*q_csr))
        except Exception as e:      # noqa: BLE001 -- reported by the main thread
            errors.append((name, repr(e)))

    ta = threading.Thread(target=work, args=("a", qa_csr, "ors", "ande", 1))
    tb = threading.Thread(target=work, args=("b", qb_csr, "rms", "rme", 3))
    ta.start(); tb.start(); ta.join(); tb.join()
    assert not errors, errors
    for name, elem_op in (("a", 0), ("b", 3)):
        counts, (off, ids, _), op, oq = results[name]
        assert np.array_equal(counts, oracle.port().size_op(op, oq, od))
        w_off, w_ids, _ = oracle.port().elem_op(elem_op, oq, od)
        assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids)


def test_offsets_scan_long_odd_and_unaligned():
    """sdrb_range_elems_count writes counts and scans them in place: every count here is known in closed form."""
    import torch

    from sdr_amd import SDRBatch, _lib
    from sdr_amd._lib import check

    lib = _lib.load()
    rng = np.random.default_rng(3)
    n_rows = 200_003
    lens = rng.integers(0, 7, n_rows)
    off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    ids = np.concatenate([np.arange(k, dtype=np.int32) * 3 for k in lens]) if off[-1] else np.zeros(0, np.int32)
    b = SDRBatch.from_csr(off, ids, width=64)
    n_ranges = 13
    lo = np.arange(n_ranges, dtype=np.int32)
    hi = np.full(n_ranges, 64, dtype=np.int32)
    # ids of a row: 0, 3, 6, ...; in [lo, 64): those with 3 k >= lo
    sizes = np.maximum(lens[None, :] - ((lo[:, None] + 2) // 3), 0).astype(np.uint64).reshape(-1)
    want = np.concatenate([[0], np.cumsum(sizes)])
    pairs = n_ranges * n_rows
    buf = torch.zeros(pairs + 8, dtype=torch.int32, device="cuda:0")
    for shift in (0, 1):       # 16-byte aligned, and 4 bytes past it
        view = buf[shift: shift + pairs + 1]
        total = C.c_uint64()
        check(lib.sdrb_range_elems_count(b._h, n_ranges, lo.ctypes.data, hi.ctypes.data, view.data_ptr(),
                                         C.byref(total), None))
        got = view.cpu().numpy().view(np.uint32).astype(np.uint64)
        assert int(total.value) == int(want[-1])
        assert np.array_equal(got, want)


def test_first_use_of_a_derived_form_from_two_host_threads():
    """The popcount kernel and the matrix-core all-pairs kernel build a form of the batch inside the first call that
    needs it and cache it in the handle.  Two host threads whose FIRST such calls on one handle coincide must both get
    right answers (the build is published under a lock, after the event that marks its end)."""
    import torch

    import oracle
    from sdr_amd import SDRBatch

    width, d_csr, qa_csr, qb_csr = _inputs()
    od = oracle.Batch(*d_csr)
    want = {"a": oracle.port().size_op(0, oracle.Batch(*qa_csr), od), "b": oracle.port().size_op(0, oracle.Batch(*qb_csr), od)}
    for attempt in range(6):
        d = SDRBatch.from_csr(*d_csr, width=width)       # a fresh handle: nothing derived yet
        start = threading.Barrier(2)
        results, errors = {}, []

        def work(name, q_csr):
            try:
                torch.cuda.set_device(0)
                q = SDRBatch.from_csr(*q_csr, width=width)
                stream = torch.cuda.Stream()
                start.wait()
                counts = q.ands(d, stream=stream, algo="popcount")
                top = q.allpairs_topk(d, 4, stream=stream, algo="mfma")
                stream.synchronize()
                results[name] = (counts.cpu().numpy().view(np.uint32), top[0].cpu().numpy(), top[1].cpu().numpy())
            except Exception as e:      # noqa: BLE001 -- reported by the main thread
                errors.append((name, repr(e)))

        ta = threading.Thread(target=work, args=("a", qa_csr))
        tb = threading.Thread(target=work, args=("b", qb_csr))
        ta.start(); tb.start(); ta.join(); tb.join()
        assert not errors, errors
        for name in ("a", "b"):
            counts, top_c, top_r = results[name]
            assert np.array_equal(counts, want[name]), (attempt, name)
            # the matrix-core kernel's best overlap per row is the row maximum of ands
            assert np.array_equal(top_c[:, 0].astype(np.uint32), want[name].max(axis=1)), (attempt, name)
