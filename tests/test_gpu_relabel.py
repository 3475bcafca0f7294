"""Size operations of batches whose ids no table kernel takes as they are -- no declared width, negative ids, 64-bit ids
that do not fit 32 bits, widths beyond every table -- go through small labels (relabel.hip) and the table kernels.  Every
count must equal the oracle's and the per-pair merge kernel's."""
import numpy as np
import pytest

import oracle
import sdr_amd
import util
from oracle import AND, OR, RM, XOR, Batch
from sdr_amd import SDRBatch

pytestmark = pytest.mark.gpu

PAIRINGS = {"EE": (False, False), "UU": (True, True), "UE": (True, False), "EU": (False, True)}
NAMES = {AND: "ands", OR: "ors", XOR: "xors", RM: "rms"}


def sizes(op, t, a, **kw):
    return getattr(t, NAMES[op])(a, **kw).cpu().numpy().view(np.uint32)


def spread(ids, factor, shift):
    """The same rows with ids far apart and partly negative: order kept."""
    return ids.astype(np.int64) * factor - shift


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("apart", [70001, 1])
@pytest.mark.parametrize("nt,na,len_t,len_a", [(1, 3000, 40, 40), (70, 1500, 40, 40), (300, 700, 40, 40), (5, 600, 300, 500),
                                               (130, 400, 64, 64), (9, 300, 2000, 33), (3, 257, 0, 40)])
def test_widthless_sizes_vs_oracle_and_merge(name, nt, na, len_t, len_a, apart):
    """Negative ids, no declared width.  apart = 70001: ids spread over +-1.75e8, the labels; apart = 1: ids within 5000 of
    each other, which the library moves to start at zero instead (the table kernels as they are)."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(nt * 7 + na + len_t)
    width = 5000
    if len_t:
        t = Batch(*util.ragged_rows(rng, nt, width, len_t, tu, lo=-2500))
    else:
        t = Batch(np.zeros(nt + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32) if tu else None)
    a = Batch(*util.ragged_rows(rng, na, width, len_a, au, lo=-2500))
    dt, da = SDRBatch.from_csr(t.off, t.ids * apart, t.data), SDRBatch.from_csr(a.off, a.ids * apart, a.data)       # no declared width
    for op in (AND, OR, XOR, RM):
        for swap in (False, True):
            want = oracle.port().size_op(op, a, t).T if swap else oracle.port().size_op(op, t, a)
            assert np.array_equal(sizes(op, dt, da, swap=swap), want), (NAMES[op], swap)
            assert np.array_equal(sizes(op, dt, da, swap=swap, algo="merge"), want), (NAMES[op], swap, "merge")


@pytest.mark.parametrize("name", ["EE", "UU"])
def test_64_bit_ids_beyond_32_bits_and_huge_declared_widths(name):
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(17)
    t = Batch(*util.ragged_rows(rng, 150, 3000, 40, tu))
    a = Batch(*util.ragged_rows(rng, 2000, 3000, 40, au))
    far_t = SDRBatch.from_csr(t.off, spread(t.ids, 0x100000003, 0x4000000000), t.data)
    far_a = SDRBatch.from_csr(a.off, spread(a.ids, 0x100000003, 0x4000000000), a.data)
    wide_t = SDRBatch.from_csr(t.off, t.ids * 300, t.data, width=1_000_000)
    wide_a = SDRBatch.from_csr(a.off, a.ids * 300, a.data, width=1_000_000)
    for op in (AND, OR, XOR, RM):
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(sizes(op, far_t, far_a), want), NAMES[op]
        assert np.array_equal(sizes(op, wide_t, wide_a), want), NAMES[op]
        assert np.array_equal(sizes(op, far_t, far_a, algo="merge"), want), NAMES[op]


def test_config_shaped_widthless_unitdata_rows():
    """Config 4's shape (16 384 / 328, UnitData) without a declared width: two chunks of A, the mask-table kernel."""
    w, act = 16384, 328
    t_csr = sdr_amd.synth_csr(2, 0, 64, w, act, True)
    a_csr = sdr_amd.synth_csr(1, 0, 1500, w, act, True)
    t, a = Batch(*t_csr), Batch(*a_csr)
    dt, da = SDRBatch.from_csr(*t_csr), SDRBatch.from_csr(*a_csr)
    for op in (AND, XOR, RM):
        assert np.array_equal(sizes(op, dt, da), oracle.port().size_op(op, t, a)), NAMES[op]


def test_element_ops_of_widthless_batches_still_equal_the_oracle():
    rng = np.random.default_rng(4)
    t = Batch(*util.ragged_rows(rng, 40, 4000, 40, True, lo=-2000))
    a = Batch(*util.ragged_rows(rng, 900, 4000, 40, True, lo=-2000))
    dt, da = SDRBatch.from_csr(t.off, t.ids, t.data), SDRBatch.from_csr(a.off, a.ids, a.data)
    for op, name in ((AND, "ande"), (OR, "ore"), (XOR, "xore"), (RM, "rme")):
        off, ids, data = getattr(dt, name)(da).to_host()
        w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
        assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids), name
        assert np.array_equal(data.view(np.uint32), w_data.view(np.uint32)), name


@pytest.mark.parametrize("name,wide", [("EE", False), ("UU", False), ("EE", True)])
def test_large_database_takes_chunks_that_fit_the_table_in_lds(name, wide):
    """More labels than a table in LDS holds (300 rows of 40 ids) against more than 2^20 elements: the rows of A are cut
    into chunks whose table fits LDS; 64-bit ids beyond 32 bits take the narrower limit of their wider keys."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(23)
    t = Batch(*util.ragged_rows(rng, 300, 3000, 40, tu))
    a = Batch(*util.ragged_rows(rng, 60000, 3000, 40, au))
    if wide:
        dt = SDRBatch.from_csr(t.off, spread(t.ids, 0x100000003, 0x4000000000), t.data)
        da = SDRBatch.from_csr(a.off, spread(a.ids, 0x100000003, 0x4000000000), a.data)
    else:
        dt = SDRBatch.from_csr(t.off, spread(t.ids, 700001, 1000000000).astype(np.int32), t.data)
        da = SDRBatch.from_csr(a.off, spread(a.ids, 700001, 1000000000).astype(np.int32), a.data)
    assert a.ids.size > 1 << 20
    assert np.array_equal(sizes(XOR, dt, da), oracle.port().size_op(XOR, t, a))
    assert np.array_equal(sizes(RM, dt, da, swap=True), oracle.port().size_op(RM, a, t).T)


def test_undeclared_width_is_derived_from_the_ids_whatever_their_size():
    """No width declared, no negative id: the library serves the batch as if largest id + 1 had been declared.  Small ids
    reach the table kernels (sizes and elements); ids spread up to 1.5e9 must still come out right (a derived width far
    beyond every table)."""
    rng = np.random.default_rng(29)
    t = Batch(*util.ragged_rows(rng, 20, 3000, 40, True))
    a = Batch(*util.ragged_rows(rng, 700, 3000, 40, True))
    for factor in (1, 500000):
        ts, as_ = Batch(t.off, (t.ids.astype(np.int64) * factor).astype(np.int32), t.data), Batch(a.off, (a.ids.astype(np.int64) * factor).astype(np.int32), a.data)
        dt, da = SDRBatch.from_csr(ts.off, ts.ids, ts.data), SDRBatch.from_csr(as_.off, as_.ids, as_.data)
        assert dt.width == 0 and da.width == 0          # what callers see stays "undeclared"
        assert da.table_width() == int(as_.ids.max()) + 1
        for op in (AND, OR, XOR, RM):
            assert np.array_equal(sizes(op, dt, da), oracle.port().size_op(op, ts, as_)), (NAMES[op], factor)
        for name, op in (("ande", AND), ("ore", OR), ("rme", RM)):
            off, ids, data = getattr(dt, name)(da).to_host()
            w_off, w_ids, w_data = oracle.port().elem_op(op, ts, as_)
            assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids), (name, factor)
            assert np.array_equal(data.view(np.uint32), w_data.view(np.uint32)), (name, factor)
    # a handful of queries against such a database take the fused few-rows path (total still on the device)
    few = SDRBatch.from_csr(t.off[:4], t.ids[: int(t.off[3])], t.data[: int(t.off[3])])
    res = few.ande(SDRBatch.from_csr(a.off, a.ids, a.data))
    assert res._total_dev is not None
    w_off, w_ids, w_data = oracle.port().elem_op(AND, Batch(t.off[:4], t.ids[: int(t.off[3])], t.data[: int(t.off[3])]), a)
    off, ids, data = res.to_host()
    assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids) and np.array_equal(data.view(np.uint32), w_data.view(np.uint32))
    neg = SDRBatch.from_csr(a.off, a.ids - 1, a.data)
    assert neg.table_width() == 0                      # a negative id: no width
    # the table kernels can be asked for by name once a width is derived
    small_t, small_a = SDRBatch.from_csr(t.off, t.ids, None), SDRBatch.from_csr(a.off, a.ids, None)
    assert np.array_equal(sizes(AND, small_t, small_a, algo="bitslice"), oracle.port().size_op(AND, Batch(t.off, t.ids, None), Batch(a.off, a.ids, None)))


ELEMS = ((AND, "ande"), (OR, "ore"), (XOR, "xore"), (RM, "rme"))


def assert_same_elems(got, want, id_type, what):
    off, ids, data = got
    w_off, w_ids, w_data = want
    assert np.array_equal(off, w_off), what
    assert ids.dtype == id_type and np.array_equal(ids, w_ids.astype(id_type)), what
    assert (data is None) == (w_data is None), what
    if data is not None:
        assert np.array_equal(data.view(np.uint32), w_data.view(np.uint32)), what


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("base,id_type", [(-1700, np.int32), (-(1 << 31), np.int32), ((1 << 31) - 4000, np.int32),
                                          (-5, np.int64), (1 << 40, np.int64), (-(1 << 50), np.int64)])
@pytest.mark.parametrize("nt,na,len_t,len_a", [(40, 900, 40, 40), (3, 600, 200, 300), (100, 130, 64, 64)])
def test_ids_close_together_but_far_from_zero_take_the_table_kernels(name, base, id_type, nt, na, len_t, len_a):
    """Ids no table indexes as they are (negative, at either end of 32 bits, 64-bit ids far beyond 32 bits) but within a
    few thousand of each other: both batches are seen through ids less the smallest one, every kernel of the path
    applies, and the elements come back as the ids they were -- sizes, elements of all four ops in both orientations,
    kernels by name (which would refuse ids they cannot index), visitors' positions."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(nt + na + len_t + (base % 1000))
    width = 4000
    t = Batch(*util.ragged_rows(rng, nt, width, len_t, tu))
    a = Batch(*util.ragged_rows(rng, na, width, len_a, au))
    dt = SDRBatch.from_csr(t.off, t.ids.astype(np.int64).astype(id_type) + id_type(base), t.data)
    da = SDRBatch.from_csr(a.off, a.ids.astype(np.int64).astype(id_type) + id_type(base), a.data)
    back = lambda r: (r[0], r[1].astype(np.int64) + base, r[2])   # noqa: E731
    for op, ename in ELEMS:
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(sizes(op, dt, da), want), NAMES[op]
        assert np.array_equal(sizes(op, dt, da, swap=True), oracle.port().size_op(op, a, t).T), NAMES[op]
        assert_same_elems(getattr(dt, ename)(da).to_host(), back(oracle.port().elem_op(op, t, a)), id_type, ename)
        got = getattr(dt, ename)(da, swap=True).to_host()
        w = back(oracle.port().elem_op(op, a, t))
        assert np.array_equal(np.diff(got[0]).reshape(t.n, a.n), np.diff(w[0]).reshape(a.n, t.n).T), ename
        assert np.array_equal(util.pair_digests(*got).reshape(t.n, a.n), util.pair_digests(*w).reshape(a.n, t.n).T), ename
    # counts as bytes (the bit-sliced kernel's shapes: sizes that ids alone decide, rows of at most 64 ids)
    if (not tu) and len_a <= 64 and len_t <= 200:
        import torch
        got = dt.ands(da, dtype=torch.uint8).cpu().contiguous().numpy()
        assert np.array_equal(got.astype(np.uint32), oracle.port().size_op(AND, t, a))
    # kernels by name: they index tables by id, so they only run because the ids were moved
    assert np.array_equal(sizes(AND, dt, da, algo="masktable"), oracle.port().size_op(AND, t, a))
    assert_same_elems(dt.ore(da, algo="rank").to_host(), back(oracle.port().elem_op(OR, t, a)), id_type, "rank")
    # the visitors report positions, which do not move
    for which, vname in ((0, "andv"), (1, "orv")):
        off, tp, ap = getattr(dt, vname)(da).to_host()
        w_off, w_tp, w_ap = oracle.port().visit(which, t, a)
        assert np.array_equal(off, w_off) and np.array_equal(tp, w_tp) and np.array_equal(ap, w_ap), vname
    # the merges, asked for by name, take the ids as they are
    assert_same_elems(dt.xore(da, algo="merge").to_host(), back(oracle.port().elem_op(XOR, t, a)), id_type, "merge")


@pytest.mark.parametrize("end", ["top", "bottom"])
def test_ids_of_two_widths_close_together_beyond_32_bits(end):
    """A 32-bit `this` whose ids end at the top of 32 bits against 64-bit rows that go on beyond it (or start at the
    bottom, the 64-bit rows reaching below): moved together; intersections and differences come back as 32-bit ids, a
    union has no 32-bit result and is refused."""
    rng = np.random.default_rng(61)
    if end == "top":
        t = Batch(*util.ragged_rows(rng, 30, 3000, 40, True))
        a = Batch(*util.ragged_rows(rng, 500, 6000, 40, False))
        top = (1 << 31) - 3000
        assert int(a.ids.max()) + top > (1 << 31)
    else:
        t = Batch(*util.ragged_rows(rng, 30, 3000, 40, True))
        a = Batch(*util.ragged_rows(rng, 500, 6000, 40, False, lo=-3000))
        top = -(1 << 31)
        assert int(a.ids.min()) + top < -(1 << 31)
    dt = SDRBatch.from_csr(t.off, (t.ids.astype(np.int64) + top).astype(np.int32), t.data)
    da = SDRBatch.from_csr(a.off, a.ids.astype(np.int64) + top, None)
    for op, ename in ((AND, "ande"), (RM, "rme")):
        assert np.array_equal(sizes(op, dt, da), oracle.port().size_op(op, t, a)), ename
        w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
        assert_same_elems(getattr(dt, ename)(da).to_host(), (w_off, w_ids.astype(np.int64) + top, w_data), np.int32, ename)
        w_off, w_ids, w_data = oracle.port().elem_op(op, a, t)
        assert_same_elems(getattr(da, ename)(dt).to_host(), (w_off, w_ids.astype(np.int64) + top, w_data), np.int64, ename + " 64")
    assert np.array_equal(sizes(OR, dt, da), oracle.port().size_op(OR, t, a))
    with pytest.raises(sdr_amd.SdrbError):
        dt.ore(da).to_host()
    w_off, w_ids, w_data = oracle.port().elem_op(OR, a, t)
    assert_same_elems(da.ore(dt).to_host(), (w_off, w_ids.astype(np.int64) + top, w_data), np.int64, "ore 64")


def test_shift_of_64_bit_ids_beyond_32_bits():
    rng = np.random.default_rng(67)
    a = Batch(*util.ragged_rows(rng, 200, 3000, 40, True))
    far = a.ids.astype(np.int64) + (1 << 45)
    d = SDRBatch.from_csr(a.off, far, a.data)
    d.shift(-1000)
    other = SDRBatch.from_csr(a.off, far - 1000, a.data)
    assert np.array_equal(sizes(AND, d, other), oracle.port().size_op(AND, a, a))
    assert np.array_equal(d.equals(other).cpu().numpy().view(np.uint32), oracle.port().equal(a, a))
    w_off, w_ids, w_data = oracle.port().elem_op(AND, a, a)
    assert_same_elems(d.ande(other).to_host(), (w_off, w_ids.astype(np.int64) + (1 << 45) - 1000, w_data), np.int64, "shifted")
    near_top = SDRBatch.from_csr(a.off, a.ids.astype(np.int64) + ((1 << 63) - 5000), a.data)
    with pytest.raises(sdr_amd.SdrbError):
        near_top.shift(3000)                      # would leave 64 bits


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("id_type", [np.int32, np.int64])
@pytest.mark.parametrize("nt,na,len_t,len_a", [(40, 900, 40, 40), (3, 257, 200, 300), (17, 130, 64, 65), (1, 70, 0, 33),
                                               (5, 64, 700, 1)])
def test_wave_per_pair_search_kernel_takes_any_ids(name, id_type, nt, na, len_t, len_a):
    """Ids far apart and partly negative (32-bit: over +-1.7e9; 64-bit: far beyond 32 bits) have no table: the element
    ops run one wave per pair with binary searches over the other row.  By name (`algo="search"`) and as the default
    choice, all four ops, both orientations, ragged and empty rows, rows beyond 64 elements; against the oracle and
    the per-pair merge."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(nt * 11 + na + len_t)
    width = 3000
    empty = lambda n, unit: Batch(np.zeros(n + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32) if unit else None)   # noqa: E731
    t = Batch(*util.ragged_rows(rng, nt, width, len_t, tu, lo=-1500)) if len_t else empty(nt, tu)
    a = Batch(*util.ragged_rows(rng, na, width, len_a, au, lo=-1500))
    far = (lambda x: x.astype(np.int64) * 0x100000003 - 5) if id_type == np.int64 else (lambda x: (x.astype(np.int64) * 1100003 - 7).astype(np.int32))
    dt, da = SDRBatch.from_csr(t.off, far(t.ids), t.data), SDRBatch.from_csr(a.off, far(a.ids), a.data)
    back = lambda r: (r[0], far(r[1]), r[2])   # noqa: E731
    for op, ename in ELEMS:
        want = back(oracle.port().elem_op(op, t, a))
        assert_same_elems(getattr(dt, ename)(da, algo="search").to_host(), want, id_type, ename + " search")
        assert_same_elems(getattr(dt, ename)(da).to_host(), want, id_type, ename + " default")
        assert_same_elems(getattr(dt, ename)(da, algo="merge").to_host(), want, id_type, ename + " merge")
        got = getattr(dt, ename)(da, swap=True, algo="search").to_host()
        w = back(oracle.port().elem_op(op, a, t))
        assert np.array_equal(np.diff(got[0]).reshape(t.n, a.n), np.diff(w[0]).reshape(a.n, t.n).T), ename
        assert np.array_equal(util.pair_digests(*got).reshape(t.n, a.n), util.pair_digests(*w).reshape(a.n, t.n).T), ename
    for which, vname in ((0, "andv"), (1, "orv")):
        w_off, w_tp, w_ap = oracle.port().visit(which, t, a)
        for algo in ("search", None):
            off, tp, ap = getattr(dt, vname)(da, algo=algo).to_host()
            assert np.array_equal(off, w_off) and np.array_equal(tp, w_tp) and np.array_equal(ap, w_ap), (vname, algo)


def test_search_kernel_with_arithdata_and_rows_too_long_for_it():
    rng = np.random.default_rng(71)
    ta = util.arith_rows(rng, 20, 3000, 60)
    aa = util.arith_rows(rng, 300, 3000, 60)
    far = lambda x: (x.astype(np.int64) * 700001 - 1000000000).astype(np.int32)   # noqa: E731
    dt = SDRBatch.from_csr(ta[0], far(ta[1]), ta[2], kind="arith")
    da = SDRBatch.from_csr(aa[0], far(aa[1]), aa[2], kind="arith")
    for op, ename in ((AND, "ande"), (OR, "ore"), (RM, "rme")):
        w_off, w_ids, w_data = oracle.port().elem_op(op, Batch(*ta, arith=True), Batch(*aa, arith=True))
        for algo in ("search", "merge", None):
            off, ids, data = getattr(dt, ename)(da, algo=algo).to_host()
            assert np.array_equal(off, w_off) and np.array_equal(ids, far(w_ids)), (ename, algo)
            assert np.array_equal(data.view(np.uint32), w_data.view(np.uint32)), (ename, algo)
    # a row that does not fit LDS: the kernel is refused by name, the default falls back to the merge
    long_off = np.array([0, 30000], np.uint64)
    long_ids = (np.arange(30000, dtype=np.int64) * 70001 - 1000000000).astype(np.int32)
    one = SDRBatch.from_csr(long_off, long_ids, None)
    with pytest.raises(sdr_amd.SdrbError) as e:
        one.ande(one, algo="search")
    assert e.value.code == -5
    off, ids, data = one.ande(one).to_host()
    assert int(off[-1]) == 30000 and np.array_equal(ids, long_ids)


@pytest.mark.parametrize("base,id_type", [(-1700, np.int32), (1 << 40, np.int64)])
def test_dot_products_and_allpairs_topk_of_ids_moved_to_zero(base, id_type):
    """`inner` visits the matches in ascending id order and the all-pairs overlap only asks which ids are equal: both are
    served for ids close together but far from zero (64-bit ids beyond 32 bits included), with the results of the same
    rows at their plain ids."""
    rng = np.random.default_rng(83)
    ta = util.arith_rows(rng, 30, 3000, 40)
    aa = util.arith_rows(rng, 800, 3000, 40)
    mv = lambda x: x.astype(np.int64).astype(id_type) + id_type(base)   # noqa: E731
    if id_type == np.int32:      # (ArithData batches hold 32-bit ids)
        dt = SDRBatch.from_csr(ta[0], mv(ta[1]), ta[2], kind="arith")
        da = SDRBatch.from_csr(aa[0], mv(aa[1]), aa[2], kind="arith")
        want = oracle.port().inner(Batch(*ta), Batch(*aa)).view(np.uint32)
        assert np.array_equal(dt.inner(da).cpu().numpy().view(np.uint32), want)
        assert np.array_equal(dt.inner(da, algo="merge").cpu().numpy().view(np.uint32), want)
    plain_t, plain_a = SDRBatch.from_csr(ta[0], ta[1], None, width=3000), SDRBatch.from_csr(aa[0], aa[1], None, width=3000)
    moved_t, moved_a = SDRBatch.from_csr(ta[0], mv(ta[1]), None), SDRBatch.from_csr(aa[0], mv(aa[1]), None)
    want_c, want_r = plain_t.allpairs_topk(plain_a, 8)
    got_c, got_r = moved_t.allpairs_topk(moved_a, 8)
    assert np.array_equal(got_c.cpu().numpy(), want_c.cpu().numpy()) and np.array_equal(got_r.cpu().numpy(), want_r.cpu().numpy())
    for algo in ("bitslice", "mfma"):
        c, r = moved_t.allpairs_topk(moved_a, 8, algo=algo)
        assert np.array_equal(c.cpu().numpy(), want_c.cpu().numpy()) and np.array_equal(r.cpu().numpy(), want_r.cpu().numpy()), algo
