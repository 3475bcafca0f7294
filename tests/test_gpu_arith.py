"""The set algebra on ArithData<float> batches (ArithData.hpp:37-88: and = product, or = sum, rm = difference,
every payload relevant, no xor) against outputs of the reference (tests/golden/arith.npz) and the oracle,
through every kernel that serves it."""
import numpy as np
import pytest

import oracle
import sdr_amd
import util
from oracle import AND, OR, RM, Batch
from sdr_amd import SDRBatch

pytestmark = pytest.mark.gpu

OPS = {AND: ("ands", "ande"), OR: ("ors", "ore"), RM: ("rms", "rme")}
PAIRINGS = {"AA": (True, True), "AE": (True, False), "EA": (False, True)}


def dev(b: Batch, width=0) -> SDRBatch:
    return SDRBatch.from_csr(b.off, b.ids, b.data, width=width, kind="arith" if b.kind == oracle.ARITH else None)


def load(g, tag):
    ta, aa = PAIRINGS[tag]
    t = Batch(g[f"{tag}_t_off"], g[f"{tag}_t_ids"], g[f"{tag}_t_data"] if ta else None, arith=ta)
    a = Batch(g[f"{tag}_a_off"], g[f"{tag}_a_ids"], g[f"{tag}_a_data"] if aa else None, arith=aa)
    return t, a


def same_elems(got, want):
    assert np.array_equal(got[0], want[0]) and np.array_equal(got[1], want[1])
    assert (want[2] is None and got[2] is None) or np.array_equal(got[2].view(np.uint32), want[2].view(np.uint32))


@pytest.mark.parametrize("tag", list(PAIRINGS))
@pytest.mark.parametrize("width", [300, 0])
def test_arith_set_algebra_vs_golden(golden, tag, width):
    g = golden("arith.npz")
    t, a = load(g, tag)
    dt, da = dev(t, width), dev(a, width)
    for op, (sname, ename) in OPS.items():
        assert np.array_equal(getattr(dt, sname)(da).cpu().numpy().view(np.uint32), g[f"{tag}_size{op}"])
        for algo in (None, "merge"):
            off, ids, data = getattr(dt, ename)(da, algo=algo).to_host()
            assert np.array_equal(np.diff(off).astype(np.uint32), g[f"{tag}_len{op}"])
            assert np.array_equal(util.pair_digests(off, ids, data), g[f"{tag}_digest{op}"])
            same_elems((off, ids, data), oracle.port().elem_op(op, t, a))


@pytest.mark.parametrize("tag", list(PAIRINGS))
def test_arith_swap_rowwise_and_inplace(golden, tag):
    g = golden("arith.npz")
    t, a = load(g, tag)
    dt, da = dev(t, 300), dev(a, 300)
    for op, (sname, ename) in OPS.items():
        # swapped roles: result (i, j) = a_j.op(t_i)
        want = oracle.port().size_op(op, a, t).T
        assert np.array_equal(getattr(dt, sname)(da, swap=True).cpu().numpy().view(np.uint32), want)
        off, ids, data = getattr(dt, ename)(da, swap=True).to_host()
        w = oracle.port().elem_op(op, a, t)
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n),
                              util.pair_digests(*w).reshape(a.n, t.n).T)
    # row-wise / in-place on the first rows of both batches
    n = t.n
    a_head = Batch(a.off[: n + 1], a.ids[: int(a.off[n])], None if a.data is None else a.data[: int(a.off[n])],
                   arith=a.kind == oracle.ARITH)
    dh = dev(a_head, 300)
    for op, (sname, ename) in OPS.items():
        full = oracle.port().elem_op(op, t, a_head)
        diag = np.arange(n) * n + np.arange(n)
        lens = np.diff(full[0])[diag]
        assert np.array_equal(dt.rowwise_size(sname, dh).cpu().numpy().view(np.uint32), lens.astype(np.uint32))
        off, ids, data = dt.rowwise_elem(ename, dh).to_host()
        assert np.array_equal(np.diff(off), lens)
        want_ids = np.concatenate([full[1][int(full[0][p]):int(full[0][p + 1])] for p in diag])
        assert np.array_equal(ids, want_ids)
        if full[2] is not None:
            want_data = np.concatenate([full[2][int(full[0][p]):int(full[0][p + 1])] for p in diag])
            assert np.array_equal(data.view(np.uint32), want_data.view(np.uint32))
    c = dev(t, 300)
    c += dh          # ori row by row: sums where both rows hold the id
    full = oracle.port().elem_op(OR, t, a_head)
    diag = np.arange(n) * n + np.arange(n)
    new_off = c.to_wire().cpu().numpy()[: n + 1].view(np.uint32)
    assert np.array_equal(np.diff(new_off), np.diff(full[0])[diag])


def test_arith_refusals():
    off = np.array([0, 2], np.uint64)
    ids = np.array([1, 5], np.int32)
    ar = SDRBatch.from_csr(off, ids, np.array([2.0, -3.0], np.float32), width=8, kind="arith")
    un = SDRBatch.from_csr(off, ids, np.array([0.5, 0.25], np.float32), width=8)
    for bad in (lambda: ar.xore(ar), lambda: ar.xors(ar), lambda: ar.ande(un), lambda: un.ore(ar), lambda: ar.ors(un)):
        with pytest.raises(sdr_amd.SdrbError):
            bad()
    # known answers: product, sum, difference on the common ids; singles keep their payloads
    em = SDRBatch.from_csr(np.array([0, 2], np.uint64), np.array([5, 6], np.int32), None, width=8)
    b2 = SDRBatch.from_csr(np.array([0, 2], np.uint64), np.array([5, 6], np.int32), np.array([10.0, 7.0], np.float32),
                           width=8, kind="arith")
    assert ar.ande(b2).to_host()[2].tolist() == [-30.0]
    assert ar.ore(b2).to_host()[1].tolist() == [1, 5, 6] and ar.ore(b2).to_host()[2].tolist() == [2.0, 7.0, 7.0]
    assert ar.rme(b2).to_host()[2].tolist() == [2.0, -13.0]
    assert ar.ore(em).to_host()[2].tolist() == [2.0, -3.0, 0.0]      # ArithData(EmptyData) is 0
    assert ar.rme(em).to_host()[1].tolist() == [1, 5]                  # an ArithData element is never removed
    assert int(ar.rms(b2)[0, 0]) == 2 and int(ar.ands(b2)[0, 0]) == 1 and int(ar.ors(b2)[0, 0]) == 3
