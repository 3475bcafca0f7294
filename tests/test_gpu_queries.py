"""Range and point queries for every row of a batch (SDR.hpp:860-956) through the C ABI, against
the fixture the reference's own ands(lo,hi) / ande(lo,hi) / ande(id) generated."""
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


@pytest.mark.parametrize("tag", ["e", "u"])
def test_range_and_point_queries_vs_golden(golden, tag):
    from sdr_amd import SDRBatch
    g = golden("queries.npz")
    data = g[f"{tag}_data"] if tag == "u" else None
    off, ids = g[f"{tag}_off"], g[f"{tag}_ids"]
    b = SDRBatch.from_csr(off, ids, data)
    lo, hi, qid = g[f"{tag}_lo"], g[f"{tag}_hi"], g[f"{tag}_qid"]
    n = b.n_rows
    counts = b.range_size(lo, hi).cpu().numpy()
    assert np.array_equal(counts.view(np.uint32), g[f"{tag}_counts"])
    # slices: lengths, and the reference's (order-sensitive) digest of the ids of every slice
    r_off, r_ids, r_data = b.range_elems(lo, hi).to_host()
    assert np.array_equal(np.diff(r_off).reshape(lo.size, n), g[f"{tag}_counts"].astype(np.uint64))
    dig = np.zeros((lo.size, n), np.int64)
    for r in range(lo.size):
        for j in range(n):
            sl = r_ids[int(r_off[r * n + j]):int(r_off[r * n + j + 1])].astype(np.int64)
            dig[r, j] = int((np.arange(1, sl.size + 1) * sl).sum()) + sl.size * 1000003
            if data is not None and sl.size:      # payloads travel with their ids
                first = int(np.searchsorted(ids[int(off[j]):int(off[j + 1])], sl[0]))
                want = data[int(off[j]) + first: int(off[j]) + first + sl.size]
                assert np.array_equal(r_data[int(r_off[r * n + j]):int(r_off[r * n + j + 1])], want)
    assert np.array_equal(dig, g[f"{tag}_digests"])
    # point queries: hit / miss as the reference's ande(id), and the payload found at the position
    pos = b.point_query(qid).cpu().numpy()
    assert np.array_equal(pos >= 0, g[f"{tag}_found"].astype(bool))
    if data is not None:
        for q in range(qid.size):
            for j in range(n):
                if pos[q, j] >= 0:
                    assert data[int(off[j]) + pos[q, j]] == g[f"{tag}_value"][q, j]


def test_queries_vs_oracle_large_and_rejections():
    import oracle
    from sdr_amd import SDRBatch, SdrbError
    rng = np.random.default_rng(5)
    csr = util.ragged_rows(rng, 5000, 2048, 40, False)
    b = SDRBatch.from_csr(*csr, width=2048)
    lo = rng.integers(0, 2048, 300).astype(np.int32)
    hi = np.minimum(lo + rng.integers(0, 600, 300), 5000).astype(np.int32)
    want, _ = oracle.port().range_size(oracle.Batch(*csr), lo, hi)
    assert np.array_equal(b.range_size(lo, hi).cpu().numpy().view(np.uint32), want)
    qid = rng.integers(-5, 2060, 200).astype(np.int32)
    assert np.array_equal(b.point_query(qid).cpu().numpy(), oracle.port().point(oracle.Batch(*csr), qid))
    with pytest.raises(SdrbError):
        b.range_size([5], [4])            # start > stop: the reference's precondition (SDR.hpp:884)
    assert b.range_size([], []).shape == (0, 5000)


def _wire(b):
    w = b.to_wire().cpu().numpy()
    off = w[: b.n_rows + 1].view(np.uint32).astype(np.uint64)
    ids = w[b.n_rows + 1: b.n_rows + 1 + b.nnz]
    data = w[b.n_rows + 1 + b.nnz:].view(np.float32) if w.size > b.n_rows + 1 + b.nnz else None
    return off, ids, data


@pytest.mark.parametrize("tag", ["small", "wide"])
def test_transpose_vs_reference_columns(golden, tag):
    """CSR -> CSC against the reference's transpose(): the same elements in every column; within a
    column ascending by row (the reference itself emits heap order there, so the fixture holds its
    columns sorted by row -- the order of its own 2x2 known answers)."""
    from sdr_amd import SDRBatch
    g = golden("transpose.npz")
    width = int(g[f"{tag}_width"])
    b = SDRBatch.from_csr(g[f"{tag}_off"], g[f"{tag}_ids"], g[f"{tag}_data"], width=width, kind="arith")
    t = b.transpose()
    assert (t.n_rows, t.nnz, t.width) == (width, b.nnz, b.n_rows)
    off, rows, vals = _wire(t)
    assert np.array_equal(off, g[f"{tag}_col_off"])
    assert np.array_equal(rows, g[f"{tag}_rows"])
    assert np.array_equal(vals.view(np.uint32), g[f"{tag}_vals"].view(np.uint32))
    # transposing back restores the batch (every row of the source is non-empty or simply absent)
    back = t.transpose()
    b_off, b_ids, b_vals = _wire(back)
    assert np.array_equal(b_off, g[f"{tag}_off"]) and np.array_equal(b_ids, g[f"{tag}_ids"])
    assert np.array_equal(b_vals.view(np.uint32), g[f"{tag}_data"].view(np.uint32))


def test_transpose_known_answer_and_inverted_index():
    """The reference's 2x2 case (src/test.cpp:517-527): [[1 2] [3 4]] -> [[1 3] [2 4]]; and an id-only
    database's transpose is its inverted index: row c = the rows containing id c."""
    from sdr_amd import SDRBatch
    m = SDRBatch.from_csr(np.array([0, 2, 4], np.uint64), np.array([0, 1, 0, 1], np.int32),
                          np.array([1, 2, 3, 4], np.float32), width=2, kind="arith")
    off, rows, vals = _wire(m.transpose())
    assert off.tolist() == [0, 2, 4] and rows.tolist() == [0, 1, 0, 1] and vals.tolist() == [1, 3, 2, 4]
    rng = np.random.default_rng(9)
    d_off, d_ids, _ = util.ragged_rows(rng, 4000, 2048, 40, False)
    inv = SDRBatch.from_csr(d_off, d_ids, None, width=2048).transpose()
    off, rows, _ = _wire(inv)
    for c in (0, 17, 2047):
        want = [r for r in range(4000) if c in d_ids[int(d_off[r]):int(d_off[r + 1])]]
        assert rows[int(off[c]):int(off[c + 1])].tolist() == want
    # the inverted index answers a point query for all rows at once
    hits = SDRBatch.from_csr(d_off, d_ids, None, width=2048).point_query([17]).cpu().numpy()[0] >= 0
    assert np.flatnonzero(hits).tolist() == rows[int(off[17]):int(off[18])].tolist()


@pytest.mark.gpu
@pytest.mark.parametrize("width,active,n,with_data", [(2048, 40, 300_000, True), (16384, 328, 9_000, False), (17, 5, 70_000, False),
                                                      (70_000, 3, 50_000, True)])
def test_transpose_of_large_batches_is_a_stable_regrouping(width, active, n, with_data):
    """Many tiles of the radix sort, several digit passes (widths of 5, 11, 14 and 17 bits): within an id the rows come
    out ascending, with the payloads they had -- compared with numpy's stable sort."""
    import sdr_amd
    from sdr_amd import SDRBatch
    off, ids, data = sdr_amd.synth_csr(3, 0, n, width, active, with_data)
    b = SDRBatch.from_csr(off, ids, data, width=width)
    t = b.transpose()
    wire = t.to_wire().cpu().numpy()
    t_off = wire[: width + 1].view(np.uint32).astype(np.int64)
    t_rows = wire[width + 1: width + 1 + t.nnz]
    order = np.argsort(ids, kind="stable")
    rows_of = np.repeat(np.arange(n, dtype=np.int32), active)
    assert np.array_equal(t_off, np.concatenate([[0], np.cumsum(np.bincount(ids, minlength=width))]))
    assert np.array_equal(t_rows, rows_of[order])
    if with_data:
        t_data = wire[width + 1 + t.nnz:].view(np.float32)
        assert np.array_equal(t_data.view(np.uint32), data[order].view(np.uint32))


@pytest.mark.gpu
def test_shift_moves_every_id_and_keeps_the_rows_usable():
    """``SDR::shift`` (SDR.hpp:1579-1590) and the operators built on it (SDR.hpp:461-468) for all rows at once: ids move
    by the amount, payloads stay; a batch without a declared width may go below zero (and loses its derived width), one
    with a width may not leave it; an overflow of 32 bits is refused, as the reference asserts."""
    import sdr_amd
    from sdr_amd import SDRBatch
    rng = np.random.default_rng(8)
    off, ids, data = util.ragged_rows(rng, 300, 2048, 40, True)
    b = SDRBatch.from_csr(off, ids, data)
    assert b.table_width() == int(ids.max()) + 1
    q = SDRBatch.from_csr(off[:6], ids[: int(off[5])], data[: int(off[5])])
    want = q.ands(b).cpu().numpy()
    up = b << 5000                     # a copy
    got = up.to_wire().cpu().numpy()
    n, nnz = b.n_rows, b.nnz
    assert np.array_equal(got[n + 1: n + 1 + nnz], ids + 5000) and np.array_equal(got[n + 1 + nnz:], data.view(np.int32))
    assert up.table_width() == int(ids.max()) + 5001
    assert np.array_equal(b.to_wire().cpu().numpy()[n + 1: n + 1 + nnz], ids)          # the original is untouched
    assert np.array_equal((q << 5000).ands(up).cpu().numpy(), want)                    # table kernels on the derived width
    down = b >> 3000                   # ids below zero: no width to derive, the label path
    assert down.table_width() == 0
    assert np.array_equal((q >> 3000).ands(down).cpu().numpy(), want)
    b <<= 17
    b >>= 17
    assert np.array_equal(b.to_wire().cpu().numpy()[n + 1: n + 1 + nnz], ids)
    assert np.array_equal(q.ands(b).cpu().numpy(), want)
    declared = SDRBatch.from_csr(off, ids, data, width=2048)
    with pytest.raises(sdr_amd.SdrbError) as e:
        declared.shift(2048)
    assert e.value.code == -3
    with pytest.raises(sdr_amd.SdrbError) as e:
        up.shift(2**31 - 6000)
    assert e.value.code == -3
    assert np.array_equal(q.ands(declared).cpu().numpy(), want)                        # a refused shift changes nothing
    # 64-bit ids that fit 32 bits move with their narrowed twin
    b64 = SDRBatch.from_csr(off, ids.astype(np.int64), data)
    q64 = SDRBatch.from_csr(off[:6], ids[: int(off[5])].astype(np.int64), data[: int(off[5])])
    b64.shift(-100)
    q64.shift(-100)
    assert np.array_equal(q64.ands(b64).cpu().numpy(), want)
