"""The entry point that allocates nothing (``sdrb_size_op_static``) against ``sdrb_size_op`` -- both through
the C ABI -- on two calls in a row, the second with a smaller input, so that what the first call left in the
library's static scratch must not leak into the second; plus the oracle on the small call, the size limit
and the ops / flags it takes."""
import ctypes as C

import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _both(lib, a, b, op, flags, torch):
    """(sdrb_size_op_static, sdrb_size_op) results of one call each, as uint32 arrays [na][nb]."""
    from sdr_amd._lib import check
    dev = f"cuda:{b.device}"
    stream = int(torch.cuda.current_stream(b.device).cuda_stream)
    got = torch.full((a.n_rows, b.n_rows), -1, dtype=torch.int32, device=dev)
    want = torch.full((a.n_rows, b.n_rows), -2, dtype=torch.int32, device=dev)
    check(lib.sdrb_size_op_static(a._h, b._h, op, flags, got.data_ptr(), got.stride(0), stream))
    check(lib.sdrb_size_op(a._h, b._h, op, flags, want.data_ptr(), want.stride(0), stream))
    torch.cuda.synchronize()
    return got.cpu().numpy().view(np.uint32), want.cpu().numpy().view(np.uint32)


def test_static_entry_point_matches_size_op_on_two_calls_in_a_row():
    import torch

    import oracle
    import sdr_amd
    from sdr_amd import SDRBatch, _lib

    lib = _lib.load()
    width, active = 2048, 40
    # first call: 1500 rows of A (24 table tiles, the last one partial) x 70 000 rows of B
    a1_csr = sdr_amd.synth_csr(12, 0, 1500, width, active)
    b1_csr = sdr_amd.synth_csr(11, 0, 70_000, width, active)
    a1, b1 = SDRBatch.from_csr(*a1_csr, width=width), SDRBatch.from_csr(*b1_csr, width=width)
    # second call: smaller on both sides, other rows, ragged lengths -- its table covers only the start of
    # the scratch the first call filled
    rng = np.random.default_rng(5)
    from util import ragged_rows
    a2_csr = ragged_rows(rng, 70, width, 64)
    b2_csr = ragged_rows(rng, 3001, width, 48)
    a2, b2 = SDRBatch.from_csr(*a2_csr, width=width), SDRBatch.from_csr(*b2_csr, width=width)

    for op in (_lib.OP_AND, _lib.OP_OR, _lib.OP_XOR, _lib.OP_RM):
        for flags in (0, _lib.FLAG_SWAP):
            got1, want1 = _both(lib, a1, b1, op, flags, torch)
            assert np.array_equal(got1, want1), f"first call differs (op {op}, flags {flags})"
            got2, want2 = _both(lib, a2, b2, op, flags, torch)
            assert np.array_equal(got2, want2), f"second, smaller call differs (op {op}, flags {flags})"
            oa, ob = oracle.Batch(*a2_csr), oracle.Batch(*b2_csr)
            # swapped: result (i, j) = B_j.op(A_i), still laid out [i][j]
            ref = oracle.port().size_op(op, ob, oa).T if flags & _lib.FLAG_SWAP else oracle.port().size_op(op, oa, ob)
            assert np.array_equal(got2, ref), f"second call differs from the oracle (op {op}, flags {flags})"


def test_static_entry_point_few_rows_and_unit_ors():
    import torch

    import sdr_amd
    from sdr_amd import SDRBatch, _lib

    lib = _lib.load()
    width, active = 2048, 40
    b_csr = sdr_amd.synth_csr(21, 0, 5000, width, active, with_data=True)
    b = SDRBatch.from_csr(*b_csr, width=width)
    for na in (1, 8, 9, 200):
        a_csr = sdr_amd.synth_csr(22, 0, na, width, active, with_data=True)
        a = SDRBatch.from_csr(*a_csr, width=width)
        got, want = _both(lib, a, b, _lib.OP_OR, 0, torch)       # ors ignores payloads
        assert np.array_equal(got, want)
        # sizes that depend on payloads are not this entry point's
        out = torch.empty((na, b.n_rows), dtype=torch.int32, device="cuda:0")
        rc = lib.sdrb_size_op_static(a._h, b._h, _lib.OP_AND, 0, out.data_ptr(), out.stride(0), None)
        assert rc == -5


def test_static_entry_point_refuses_what_exceeds_its_scratch():
    import torch

    import sdr_amd
    from sdr_amd import SDRBatch, _lib

    lib = _lib.load()
    b = SDRBatch.from_csr(*sdr_amd.synth_csr(31, 0, 300, 4096, 40), width=4096)
    ok = SDRBatch.from_csr(*sdr_amd.synth_csr(32, 0, 4096, 4096, 10), width=4096)
    big = SDRBatch.from_csr(*sdr_amd.synth_csr(32, 0, 4097, 4096, 10), width=4096)
    got, want = _both(lib, ok, b, _lib.OP_AND, 0, torch)
    assert np.array_equal(got, want)
    out = torch.empty((big.n_rows, b.n_rows), dtype=torch.int32, device="cuda:0")
    rc = lib.sdrb_size_op_static(big._h, b._h, _lib.OP_AND, 0, out.data_ptr(), out.stride(0), None)
    assert rc == -5 and b"static scratch" in lib.sdrb_last_error()
    # flags other than SWAP, row-wise included, are refused
    rc = lib.sdrb_size_op_static(ok._h, b._h, _lib.OP_AND, _lib.FLAG_ROWWISE, out.data_ptr(), out.stride(0), None)
    assert rc == -1
    # and the next call still works
    got, want = _both(lib, ok, b, _lib.OP_XOR, 0, torch)
    assert np.array_equal(got, want)
