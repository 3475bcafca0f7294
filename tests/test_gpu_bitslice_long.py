"""The bit-sliced size kernel for long rows and wide tables (rows of B beyond 64 ids, widths beyond 4096): every count
against the oracle, with 32- and 64-row tiles, on both sides of its limits."""
import numpy as np
import pytest

import oracle
import sdr_amd
import util
from oracle import AND, OR, RM, XOR, Batch
from sdr_amd import SDRBatch, _lib

pytestmark = pytest.mark.gpu
NAMES = {AND: "ands", OR: "ors", XOR: "xors", RM: "rms"}


def sizes(op, t, a, **kw):
    return getattr(t, NAMES[op])(a, **kw).cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("mask_bytes", ["4", "8"])
@pytest.mark.parametrize("width,len_t,len_a,nt,na", [(2048, 40, 40, 70, 700), (2048, 300, 300, 33, 513), (16384, 328, 328, 64, 300),
                                                      (9000, 700, 90, 100, 257), (300, 280, 290, 65, 255), (5000, 3, 600, 1, 1000),
                                                      (20000, 50, 1200, 40, 130)])
def test_long_rows_and_wide_tables_vs_oracle(monkeypatch, mask_bytes, width, len_t, len_a, nt, na):
    monkeypatch.setenv("SDRB_LONG_MASK_BYTES", mask_bytes)
    rng = np.random.default_rng(width + len_a + nt)
    t = Batch(*util.ragged_rows(rng, nt, width + 40, min(len_t, width), lo=-10))      # some ids of A outside B's width
    a = Batch(*util.ragged_rows(rng, na, width, min(len_a, width)))
    dt, da = SDRBatch.from_csr(t.off, t.ids, None, width=0), SDRBatch.from_csr(a.off, a.ids, None, width=width)
    for op in (AND, OR, XOR, RM):
        for swap in (False, True):
            if swap:
                continue        # the streamed batch must be the one with the declared width
            want = oracle.port().size_op(op, t, a)
            assert np.array_equal(sizes(op, dt, da, algo="bitslice_long"), want), (NAMES[op], mask_bytes)
    # swapped roles: B_j.op(A_i) into the same layout
    want = oracle.port().size_op(RM, a, t).T
    assert np.array_equal(sizes(RM, dt, da, swap=True, algo="bitslice_long"), want)


def test_default_choice_for_long_id_only_rows_and_payload_batches():
    rng = np.random.default_rng(2)
    width = 16384
    t = Batch(*util.ragged_rows(rng, 64, width, 328, True))
    a = Batch(*util.ragged_rows(rng, 2000, width, 328, True))
    dt, da = SDRBatch.from_csr(t.off, t.ids, t.data, width=width), SDRBatch.from_csr(a.off, a.ids, a.data, width=width)
    # ors never looks at payloads: the default choice equals the forced kernel and the oracle
    want = oracle.port().size_op(OR, t, a)
    assert np.array_equal(sizes(OR, dt, da), want)
    assert np.array_equal(sizes(OR, dt, da, algo="bitslice_long"), want)
    assert np.array_equal(sizes(OR, dt, da, algo="masktable"), want)
    # payload-dependent sizes are refused by it
    with pytest.raises(_lib.SdrbError) as e:
        sizes(AND, dt, da, algo="bitslice_long")
    assert e.value.code == -5
    # widths beyond the LDS are refused when the ids spread over them
    wide = SDRBatch.from_csr(a.off, a.ids * 6, None, width=100_000)
    with pytest.raises(_lib.SdrbError) as e:
        sizes(OR, SDRBatch.from_csr(t.off, t.ids * 6, None, width=100_000), wide, algo="bitslice_long")
    assert e.value.code == -5
    # ... and served when the ids all lie in a window a table holds (they are moved to zero for the call)
    window = SDRBatch.from_csr(a.off, a.ids + 70_000, None, width=100_000)
    got = sizes(OR, SDRBatch.from_csr(t.off, t.ids + 70_000, None, width=100_000), window, algo="bitslice_long")
    assert np.array_equal(got, want)


def test_full_share_of_config_4_id_only():
    """64 x 128 Ki rows at 16 384 / 328: ors through the long-row kernel equals the mask-table kernel everywhere and the
    oracle on a sample of rows."""
    import torch
    w, act, nq, nd = 16384, 328, 64, 1 << 17
    q_csr, d_csr = sdr_amd.synth_csr(2, 0, nq, w, act), sdr_amd.synth_csr(1, 0, nd, w, act)
    q, d = SDRBatch.from_csr(*q_csr, width=w), SDRBatch.from_csr(*d_csr, width=w)
    for name, op in (("ors", OR), ("ands", AND)):
        got = getattr(q, name)(d, algo="bitslice_long")
        assert bool(torch.equal(got, getattr(q, name)(d, algo="masktable"))), name
        rows = np.sort(np.random.default_rng(1).choice(nd, 24, replace=False))
        sub = Batch(np.arange(25, dtype=np.uint64) * act, np.concatenate([d_csr[1][r * act:(r + 1) * act] for r in rows]), None)
        want = oracle.port().size_op(op, Batch(*q_csr), sub)
        assert np.array_equal(got[:, torch.as_tensor(rows, device="cuda")].cpu().numpy().view(np.uint32), want), name
