"""Batched visitors on the GPU (sdrb_visit_count / sdrb_visit_fill through the C ABI): the calls that
SDR::andv (SDR.hpp:186-194, 958-997) and SDR::orv (SDR.hpp:268-285, 1099-1166) make to their visitors, for
every pair, against the golden call sequences taken from the reference's own visitors and against the oracle."""
import numpy as np
import pytest

import oracle
import sdr_amd
import util
from oracle import Batch
from sdr_amd import POS_NONE, SDRBatch

pytestmark = pytest.mark.gpu

VISITORS = {0: "andv", 1: "orv"}


def dev(b: Batch, width=0) -> SDRBatch:
    return SDRBatch.from_csr(b.off, b.ids, b.data, width=width)


def assert_same(got, want):
    assert np.array_equal(got[0], want[0]), "offsets differ"
    assert np.array_equal(got[1], want[1]), "this positions differ"
    assert np.array_equal(got[2], want[2]), "arg positions differ"


@pytest.mark.parametrize("algo", [None, "merge"])
def test_fuzz_universe_vs_golden(golden, algo):
    g = golden("visit.npz")
    t = Batch(*util.fuzz_universe(64, None))
    for width in (6, 0):                      # declared width: rank tables; none: per-pair merge
        d = dev(t, width)
        for which, name in VISITORS.items():
            got = getattr(d, name)(d, algo=algo).to_host()
            assert_same(got, (g[f"fuzz_{name}_off"], g[f"fuzz_{name}_this"], g[f"fuzz_{name}_arg"]))


@pytest.mark.parametrize("algo", [None, "merge"])
def test_ragged_rows_vs_golden_and_oracle(golden, algo):
    g = golden("visit.npz")
    q, d = Batch(g["q_off"], g["q_ids"]), Batch(g["d_off"], g["d_ids"])
    dq, dd = dev(q, 3000), dev(d, 3000)
    for which, name in VISITORS.items():
        off, tp, ap = getattr(dq, name)(dd, algo=algo).to_host()
        assert np.array_equal(np.diff(off).astype(np.uint32), g[f"ragged_{name}_len"])
        assert np.array_equal(util.visit_digests(off, tp, ap), g[f"ragged_{name}_digest"])
        assert_same((off, tp, ap), oracle.port().visit(which, q, d))
        # swapped roles: this = the long side; pair (i, j) holds row j's view of row i
        s_off, s_tp, s_ap = getattr(dq, name)(dd, swap=True, algo=algo).to_host()
        w_off, w_tp, w_ap = oracle.port().visit(which, d, q)
        assert np.array_equal(np.diff(s_off).reshape(q.n, d.n), np.diff(w_off).reshape(d.n, q.n).T)
        assert np.array_equal(util.visit_digests(s_off, s_tp, s_ap).reshape(q.n, d.n),
                              util.visit_digests(w_off, w_tp, w_ap).reshape(d.n, q.n).T)


def test_payloads_play_no_part_and_positions_address_them():
    rng = np.random.default_rng(8)
    t = Batch(*util.ragged_rows(rng, 12, 500, 90, True))
    a = Batch(*util.ragged_rows(rng, 70, 500, 90, True))
    dt, da = dev(t, 500), dev(a, 500)
    plain = dev(Batch(t.off, t.ids), 500).andv(dev(Batch(a.off, a.ids), 500)).to_host()
    off, tp, ap = dt.andv(da).to_host()
    assert_same((off, tp, ap), plain)
    # a custom operation built from the visitor records: the sum of payload products of every pair (= inner)
    pair = np.repeat(np.arange(t.n * a.n), np.diff(off).astype(np.int64))
    i, j = pair // a.n, pair % a.n
    prod = t.data[t.off[i].astype(np.int64) + tp].astype(np.float64) * a.data[a.off[j].astype(np.int64) + ap]
    dots = np.zeros(t.n * a.n)
    np.add.at(dots, pair, prod)
    want = np.zeros((t.n, a.n))
    for x in range(t.n):
        ti, tv = t.row(x)
        for y in range(a.n):
            ai, av = a.row(y)
            _, ia, ib = np.intersect1d(ti, ai, return_indices=True)
            want[x, y] = float(np.sum(tv[ia].astype(np.float64) * av[ib]))
    assert np.allclose(dots.reshape(t.n, a.n), want, rtol=0, atol=1e-12)
    # orv: singles carry POS_NONE on the absent side, every element of both rows appears exactly once
    off, tp, ap = dt.orv(da).to_host()
    lens = np.diff(off).reshape(t.n, a.n)
    assert np.array_equal(lens, dt.ors(da).cpu().numpy())
    assert np.count_nonzero(tp != POS_NONE) == int(np.diff(t.off).sum()) * a.n
    assert np.count_nonzero(ap != POS_NONE) == int(np.diff(a.off).sum()) * t.n


def test_rowwise_and_64_bit_ids():
    rng = np.random.default_rng(9)
    t = Batch(*util.ragged_rows(rng, 50, 200, 40, False))
    a = Batch(*util.ragged_rows(rng, 50, 200, 40, False))
    dt, da = dev(t, 200), dev(a, 200)
    for which, name in VISITORS.items():
        off, tp, ap = getattr(dt, name)(da, rowwise=True).to_host()
        w_off, w_tp, w_ap = oracle.port().visit(which, t, a)
        diag = np.arange(t.n) * a.n + np.arange(t.n)
        lens = np.diff(w_off)[diag]
        assert np.array_equal(np.diff(off), lens)
        want_tp = np.concatenate([w_tp[int(w_off[p]):int(w_off[p + 1])] for p in diag])
        want_ap = np.concatenate([w_ap[int(w_off[p]):int(w_off[p + 1])] for p in diag])
        assert np.array_equal(tp, want_tp) and np.array_equal(ap, want_ap)
    # 64-bit ids: the same positions as for the order-preserving 32-bit originals
    scale = lambda ids: ids.astype(np.int64) * 0x100000003 - 0x4000000000   # noqa: E731
    lt = SDRBatch.from_csr(t.off, scale(t.ids), None)      # int64 ids make a 64-bit batch
    la = SDRBatch.from_csr(a.off, scale(a.ids), None)
    for which, name in VISITORS.items():
        assert_same(getattr(lt, name)(la).to_host(), oracle.port().visit(which, t, a))


def test_batch_against_itself_with_long_rows():
    # rows longer than a wave: the count runs through the dense-bitset kernel, whose derived bit planes are
    # built through payload-free views of the batch -- twice the same batch here, and again on a second call
    rng = np.random.default_rng(10)
    t = Batch(*util.ragged_rows(rng, 40, 512, 200, True))
    d = dev(t, 512)
    for _ in range(2):
        for which, name in VISITORS.items():
            assert_same(getattr(d, name)(d).to_host(), oracle.port().visit(which, t, t))
    assert np.array_equal(d.ands(d).cpu().numpy().view(np.uint32), oracle.port().size_op(oracle.AND, t, t))


def test_operator_equal_vs_golden_and_oracle(golden):
    # SDR::operator== for every pair and row-wise, every pairing the reference compiles, 32- and 64-bit ids
    from test_oracle_golden import equal_cases
    g = golden("equal.npz")
    for tag, (t, a) in equal_cases(g).items():
        kt = "arith" if t.kind == oracle.ARITH else None
        ka = "arith" if a.kind == oracle.ARITH else None
        for width in (6, 0):
            dt = SDRBatch.from_csr(t.off, t.ids, t.data, width=width, kind=kt)
            da = SDRBatch.from_csr(a.off, a.ids, a.data, width=width, kind=ka)
            got = dt.equals(da).cpu().numpy()
            assert np.array_equal(got, g[tag]), tag
            assert np.array_equal(dt.equals(da, rowwise=True).cpu().numpy(), np.diagonal(g[tag]))
    rng = np.random.default_rng(12)
    t = Batch(*util.ragged_rows(rng, 60, 50, 6, True))
    a = Batch(*util.ragged_rows(rng, 90, 50, 6, True))
    assert np.array_equal(dev(t, 50).equals(dev(a, 50)).cpu().numpy(), oracle.port().equal(t, a))
    wide = lambda b: SDRBatch.from_csr(b.off, b.ids.astype(np.int64) * 0x100000003, b.data)   # noqa: E731
    assert np.array_equal(wide(t).equals(wide(a)).cpu().numpy(), oracle.port().equal(t, a))
    with pytest.raises(sdr_amd.SdrbError):      # the reference does not compile EmptyData == payload either
        dev(Batch(t.off, t.ids), 50).equals(dev(a, 50))


def test_empty_rows_and_batches():
    e = SDRBatch.from_rows([[], [], []], width=64)
    some = SDRBatch.from_rows([[1, 5], [], [0, 1, 63]], width=64)
    off, tp, ap = e.andv(some).to_host()
    assert off.tolist() == [0] * 10 and tp.size == 0
    off, tp, ap = e.orv(some).to_host()
    assert np.diff(off).reshape(3, 3).tolist() == [[2, 0, 3]] * 3
    assert (tp == POS_NONE).all() and ap.tolist() == [0, 1, 0, 1, 2] * 3
    with pytest.raises(ValueError):
        some.andv(SDRBatch.from_rows([[1]], width=64), rowwise=True)
    # batches of different id widths visit each other like any two rows: id 1 of [1, 5] and of [0, 1, 63]
    off, tp, ap = some.andv(SDRBatch.from_csr(np.array([0, 1], np.uint64), np.array([1], np.int64), None)).to_host()
    assert off.tolist() == [0, 1, 1, 2] and tp.tolist() == [0, 1] and ap.tolist() == [0, 0]


def test_andv_over_sub_ranges():
    """andv on sub-ranges of both rows (SDR.hpp:196-202) = andv of the rows cut to those ranges, positions counted
    from the start of the whole rows; here every pair gets the same windows."""
    rng = np.random.default_rng(77)
    t_off, t_ids, _ = util.ragged_rows(rng, 9, 300, 60)
    a_off, a_ids, _ = util.ragged_rows(rng, 40, 300, 60)
    t, a = SDRBatch.from_csr(t_off, t_ids, width=300), SDRBatch.from_csr(a_off, a_ids, width=300)
    t_lo, t_hi, a_lo, a_hi = 5, 33, 2, 41

    def cut(off, ids, lo, hi):
        rows = [ids[int(off[r]):int(off[r + 1])][lo:hi] for r in range(len(off) - 1)]
        new_off = np.concatenate([[0], np.cumsum([len(r) for r in rows])]).astype(np.uint64)
        return new_off, (np.concatenate(rows) if rows else np.zeros(0, np.int32)).astype(np.int32)

    ct, ca = SDRBatch.from_csr(*cut(t_off, t_ids, t_lo, t_hi), width=300), SDRBatch.from_csr(*cut(a_off, a_ids, a_lo, a_hi), width=300)
    got = t.andv(a).window(t_lo, t_hi, a_lo, a_hi)
    want = ct.andv(ca)
    g_off, g_tp, g_ap = got.to_host()
    w_off, w_tp, w_ap = want.to_host()
    assert got.total == want.total and np.array_equal(g_off, w_off)
    assert np.array_equal(g_tp, w_tp + t_lo) and np.array_equal(g_ap, w_ap + a_lo)


@pytest.mark.gpu
def test_orv_five_callback_tails():
    """The five-callback orv (SDR.hpp:284-285, 1099-1150), restated here call by call: single-side visitors until one side
    is exhausted, then one position call for the rest of the other side.  VisitResult.orv_tails must name, for every
    pair, the same number of visitor calls, the same position call and the same position -- rows that end together,
    empty rows on either side and ordinary ragged rows."""
    import util
    from oracle import Batch
    from sdr_amd import SDRBatch
    rng = np.random.default_rng(5)
    t = Batch(*util.ragged_rows(rng, 23, 60, 30, False))
    a = Batch(*util.ragged_rows(rng, 41, 60, 30, False))
    # make some rows empty and some pairs identical
    t_rows = [t.row(r)[0].tolist() for r in range(t.n)]
    a_rows = [a.row(r)[0].tolist() for r in range(a.n)]
    t_rows[3], a_rows[5], a_rows[6] = [], [], list(t_rows[0])
    t, a = Batch.from_rows(t_rows), Batch.from_rows(a_rows)
    res = SDRBatch.from_csr(t.off, t.ids, None, width=60).orv(SDRBatch.from_csr(a.off, a.ids, None, width=60))
    off, _, _ = res.to_host()
    head_end, arg_only, position = res.orv_tails(np.diff(t.off), np.diff(a.off))

    def five_callbacks(x, y):
        i = j = calls = 0
        while True:
            if i == len(x):
                return calls, True, j          # arg_only_position(arg_pos)
            if j == len(y):
                return calls, False, i         # this_only_position(this_pos)
            if x[i] < y[j]:
                i += 1
            elif x[i] > y[j]:
                j += 1
            else:
                i += 1
                j += 1
            calls += 1

    for i in range(t.n):
        for j in range(a.n):
            p = i * a.n + j
            calls, which, pos = five_callbacks(t_rows[i], a_rows[j])
            assert head_end[p] - off[p] == calls and bool(arg_only[p]) == which and position[p] == pos, (i, j)
