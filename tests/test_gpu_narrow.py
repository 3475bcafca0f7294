"""Size operations with narrow counts (uint8 / uint16) through sdrb_size_op_narrow: bit-equal to the oracle and to the
uint32 counts, refused -- never truncated -- when the longest rows allow a value that does not fit."""
import numpy as np
import pytest
import torch

import oracle
import sdr_amd
import util
from oracle import AND, OR, RM, XOR, Batch
from sdr_amd import SDRBatch, _lib

pytestmark = pytest.mark.gpu

NAMES = {AND: "ands", OR: "ors", XOR: "xors", RM: "rms"}


def dev(b: Batch, width):
    return SDRBatch.from_csr(b.off, b.ids, b.data, width=width)


def narrow(op, t, a, dtype, **kw):
    got = getattr(t, NAMES[op])(a, dtype=dtype, **kw).cpu().numpy()
    return got.astype(np.uint32) if dtype == torch.uint8 else got.view(np.uint16).astype(np.uint32)


@pytest.mark.parametrize("dtype", [torch.uint8, torch.int16])
def test_fuzz_grid_narrow_counts_vs_golden(golden, dtype):
    g = golden("fuzz_sizes.npz")
    off, ids, _ = util.fuzz_universe(256, None)
    b = Batch(off, ids, None)
    d = dev(b, 8)
    for op in (AND, OR, XOR, RM):       # the golden array is indexed by op in this order
        assert np.array_equal(narrow(op, d, d, dtype), g["EE"][op].astype(np.uint32)), NAMES[op]


@pytest.mark.parametrize("width,len_t,len_a,nt,na", [(2048, 40, 40, 130, 1111), (4096, 64, 64, 70, 513), (300, 20, 16, 1, 257),
                                                      (2048, 150, 33, 65, 900), (64, 64, 64, 9, 255), (2048, 40, 40, 64, 4)])
@pytest.mark.parametrize("dtype", [torch.uint8, torch.int16])
def test_ragged_rows_narrow_counts_vs_oracle(width, len_t, len_a, nt, na, dtype):
    rng = np.random.default_rng(width + len_t + na)
    t = Batch(*util.ragged_rows(rng, nt, width, len_t))
    a = Batch(*util.ragged_rows(rng, na, width, len_a))
    dt, da = dev(t, width), dev(a, width)
    for op in (AND, OR, XOR, RM):
        for swap in (False, True):
            bound = {AND: min(len_t, len_a), RM: len_a if swap else len_t}.get(op, len_t + len_a)
            if dtype == torch.uint8 and bound > 255:
                # the rows drawn may still be short enough: the library decides by the longest rows it holds
                longest = {AND: min(dt.max_row_len, da.max_row_len), RM: da.max_row_len if swap else dt.max_row_len}.get(
                    op, dt.max_row_len + da.max_row_len)
                if longest > 255:
                    with pytest.raises(_lib.SdrbError) as e:
                        narrow(op, dt, da, dtype, swap=swap)
                    assert e.value.code == -5
                    continue
            want = oracle.port().size_op(op, a, t).T if swap else oracle.port().size_op(op, t, a)
            assert np.array_equal(narrow(op, dt, da, dtype, swap=swap), want), (NAMES[op], swap)


def test_narrow_counts_with_payload_batches_follow_the_id_only_rules():
    rng = np.random.default_rng(3)
    width = 2048
    t = Batch(*util.ragged_rows(rng, 70, width, 40, True))
    a = Batch(*util.ragged_rows(rng, 600, width, 40, True))
    dt, da = dev(t, width), dev(a, width)
    # ors never looks at payloads (SDR.hpp:1298-1301)
    assert np.array_equal(narrow(OR, dt, da, torch.uint8), oracle.port().size_op(OR, t, a))
    # payload-dependent sizes have no narrow form
    with pytest.raises(_lib.SdrbError) as e:
        narrow(AND, dt, da, torch.uint8)
    assert e.value.code == -5
    # an EmptyData `this` ignores the argument's payloads (EmptyData.hpp:44-80)
    te = Batch(t.off, t.ids, None)
    assert np.array_equal(narrow(XOR, dev(te, width), da, torch.uint8), oracle.port().size_op(XOR, te, a))


def test_narrow_counts_into_a_caller_tensor_and_bad_leading_dimensions():
    off, ids, _ = sdr_amd.synth_csr(5, 0, 1001, 2048, 40)
    q_off, q_ids, _ = sdr_amd.synth_csr(6, 0, 67, 2048, 40)
    d, q = SDRBatch.from_csr(off, ids, None, width=2048), SDRBatch.from_csr(q_off, q_ids, None, width=2048)
    want = oracle.port().size_op(AND, Batch(q_off, q_ids, None), Batch(off, ids, None))
    out = torch.full((67, 1024), 77, dtype=torch.uint8, device="cuda")
    got = q.ands(d, dtype=torch.uint8, out=out)
    assert np.array_equal(got.cpu().numpy(), want.astype(np.uint8))
    assert bool((out[:, 1004:] == 77).all())                  # only the padding up to the next multiple of 4 may change
    bad = torch.empty((67, 1002), dtype=torch.uint8, device="cuda")
    with pytest.raises(_lib.SdrbError) as e:
        q.ands(d, dtype=torch.uint8, out=bad)
    assert e.value.code == -1


def test_full_size_narrow_counts_equal_the_uint32_counts():
    """4096 queries x 1 Mi rows: every byte count equals the uint32 count of the same pair (which the full-size tests
    pin against the oracle and closed forms), for ands and xors."""
    width, active, nq, nd = 2048, 40, 4096, 1 << 20
    q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, width, active)[:2], None, width=width)
    d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, width, active)[:2], None, width=width)
    wide = torch.empty((nq, nd), dtype=torch.int32, device="cuda")
    for name in ("ands", "xors"):
        getattr(q, name)(d, out=wide)
        small = getattr(q, name)(d, dtype=torch.uint8)
        for i0 in range(0, nq, 512):
            assert bool(torch.equal(small[i0:i0 + 512].to(torch.int32), wide[i0:i0 + 512])), name
        del small
    small16 = q.ors(d, dtype=torch.int16)
    q.ors(d, out=wide)
    for i0 in range(0, nq, 512):
        assert bool(torch.equal(small16[i0:i0 + 512].to(torch.int32), wide[i0:i0 + 512]))
