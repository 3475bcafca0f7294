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
