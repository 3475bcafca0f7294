"""Result-type retargeting of the element ops and rmi with a caller's combiner, against a per-row restatement of
the reference's templates written out here (SDR.hpp:999-1025, 1183-1222, 1307-1349, 1447-1513 with
EmptyData.hpp:27-80 and UnitData.hpp:13-28).  The reference's own test retargets the id type and the container only
(/root/reference/src/test.cpp:204-229); payload retargeting has no golden vector there: parity unpinned beyond the
restatement."""
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import util  # noqa: E402

pytestmark = pytest.mark.gpu

AND, OR, XOR, RM = range(4)
F = np.float32


def batch_csr(b):
    wire = b.to_wire().cpu().numpy()
    n, nnz = b.n_rows, b.nnz
    data = wire[n + 1 + nnz:].view(np.float32) if wire.size > n + 1 + nnz else None
    return wire[: n + 1].view(np.uint32), wire[n + 1: n + 1 + nnz], data


def rows_of(off, ids, data):
    return [[(int(ids[k]), None if data is None else F(data[k])) for k in range(int(off[r]), int(off[r + 1]))]
            for r in range(len(off) - 1)]


def convert(v, ret):          # ret_t(elem): UnitData(EmptyData) = 1, EmptyData(anything) = empty
    if ret == "empty":
        return None
    return F(1.0) if v is None else v


def combine(op, t, a, ret):
    """(kept, data) of a common id: this_pos->data().op<ret data>(arg_pos->data()) and its relevance test."""
    if t is None:                                   # EmptyData::op<ret_t>() = ret_t()
        d = None if ret in (None, "empty") else F(1.0)
    else:                                           # UnitData op, then converted to the result's data type
        if a is None:
            d = t
        elif op == AND:
            d = F(t * a)
        elif op == OR:
            d = t if t > a else a
        elif op == XOR:
            d = F(abs(F(t - a)))
        else:
            d = F(t * F(F(1.0) - a))
        if ret == "empty":
            d = None
    relevant = True if d is None else bool(d >= F(0.1))
    rm_relevant = False if d is None else bool(d >= F(0.1))
    return (relevant if op in (AND, OR) else rm_relevant), d


def ref_op(op, this, arg, ret):
    out, i, j = [], 0, 0
    while i < len(this) or j < len(arg):
        if j == len(arg) or (i < len(this) and this[i][0] < arg[j][0]):
            if op != AND:
                out.append((this[i][0], convert(this[i][1], ret) if ret else this[i][1]))
            i += 1
        elif i == len(this) or arg[j][0] < this[i][0]:
            if op in (OR, XOR):
                this_has_data = (ret == "unit") or (ret is None and this and this[0][1] is not None)
                out.append((arg[j][0], (F(1.0) if arg[j][1] is None else arg[j][1]) if this_has_data else None))
            j += 1
        else:
            kept, d = combine(op, this[i][1], arg[j][1], ret)
            if kept or op == OR:
                out.append((this[i][0], d))
            i += 1
            j += 1
    return out


def check(res, na, nb, want_rows, with_data):
    off, ids, data = res.to_host()
    assert (data is not None) == with_data
    for i in range(na):
        for j in range(nb):
            lo, hi = int(off[i * nb + j]), int(off[i * nb + j + 1])
            want = want_rows[i][j]
            assert ids[lo:hi].tolist() == [w[0] for w in want], (i, j)
            if with_data:
                assert data[lo:hi].view(np.uint32).tolist() == [int(np.float32(w[1]).view(np.uint32)) for w in want], (i, j)


@pytest.mark.parametrize("this_unit,arg_unit", [(False, False), (False, True), (True, False), (True, True)])
def test_result_type_retargeting(this_unit, arg_unit):
    import torch
    from sdr_amd import SDRBatch
    rng = np.random.default_rng(17 + 2 * this_unit + arg_unit)
    t = util.ragged_rows(rng, 7, 120, 30, this_unit)
    a = util.ragged_rows(rng, 23, 120, 30, arg_unit)
    dt, da = SDRBatch.from_csr(*t, width=120), SDRBatch.from_csr(*a, width=120)
    rt, ra = rows_of(*t), rows_of(*a)
    for op, name in ((AND, "ande"), (OR, "ore"), (XOR, "xore"), (RM, "rme")):
        res = getattr(dt, name)(da, ret="empty")
        check(res, 7, 23, [[ref_op(op, x, y, "empty") for y in ra] for x in rt], False)
        if name == "rme" and not this_unit:
            with pytest.raises(ValueError):
                dt.rme(da, ret="unit")
            continue
        res = getattr(dt, name)(da, ret="unit")
        check(res, 7, 23, [[ref_op(op, x, y, "unit") for y in ra] for x in rt], True)
        assert res.ids_as(torch.int64).dtype == torch.int64 and torch.equal(res.ids_as(torch.int64).to(torch.int32), res.ids)


@pytest.mark.parametrize("arg_unit", [False, True])
def test_rmi_with_a_combiner(arg_unit):
    import torch
    from sdr_amd import SDRBatch
    rng = np.random.default_rng(29 + arg_unit)
    t = util.ragged_rows(rng, 60, 200, 50, True)
    a = util.ragged_rows(rng, 60, 200, 50, arg_unit)
    plain = SDRBatch.from_csr(*t, width=200).rmi(SDRBatch.from_csr(*a, width=200))
    # the payload type's own rule handed in as a combiner gives the same batch
    own = (lambda tv, av: tv * (1.0 - av)) if arg_unit else (lambda tv, av: tv)
    same = SDRBatch.from_csr(*t, width=200).rmi(SDRBatch.from_csr(*a, width=200), combine=own)
    assert all(np.array_equal(x, y) for x, y in zip(batch_csr(plain), batch_csr(same)))
    # a rule of the caller: halve the payload of every common id
    got = SDRBatch.from_csr(*t, width=200).rmi(SDRBatch.from_csr(*a, width=200), combine=lambda tv, av: tv * 0.5)
    g_off, g_ids, g_data = batch_csr(got)
    rt, ra = rows_of(*t), rows_of(*a)
    for r in range(60):
        arg_ids = {x[0] for x in ra[r]}
        want = []
        for i, v in rt[r]:
            if i in arg_ids:
                v = F(v * F(0.5))
                if not v >= F(0.1):
                    continue
            want.append((i, v))
        lo, hi = int(g_off[r]), int(g_off[r + 1])
        assert g_ids[lo:hi].tolist() == [w[0] for w in want], r
        assert g_data[lo:hi].view(np.uint32).tolist() == [int(np.float32(w[1]).view(np.uint32)) for w in want], r
