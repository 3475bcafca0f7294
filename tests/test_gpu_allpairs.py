"""All-pairs overlap + fused per-row top-k (int8 0/1 product on the matrix cores) against the
oracle.  Each count is the reference's `ands` on EmptyData elements (SDR.hpp:1083-1095), which
the oracle restates; the reference has no top-k, so the selection rule checked here -- overlap
descending, then the lower row -- is this engine's own ("parity unpinned" for the rule, pinned
for every count)."""
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


def expected_topk(counts, k, exclude_offset):
    """counts: [na, nb] overlaps from the oracle."""
    na, nb = counts.shape
    want_c = np.zeros((na, k), np.int64)
    want_r = np.full((na, k), -1, np.int64)
    for i in range(na):
        c = counts[i].astype(np.int64)
        cand = np.arange(nb)
        if exclude_offset >= 0:
            cand = cand[cand != i + exclude_offset]
        order = cand[np.lexsort((cand, -c[cand]))][:k]
        want_c[i, : len(order)] = c[order]
        want_r[i, : len(order)] = order
    return want_c, want_r


ALGOS = ["mfma", "bitslice"]


def run_case(a_csr, b_csr, width, k, exclude_offset, algos=ALGOS):
    import oracle
    from sdr_amd import SDRBatch
    a = SDRBatch.from_csr(*a_csr, width=width)
    b = SDRBatch.from_csr(*b_csr, width=width)
    counts = oracle.port().size_op(oracle.AND, oracle.Batch(a_csr[0], a_csr[1], None),
                                   oracle.Batch(b_csr[0], b_csr[1], None))
    want_c, want_r = expected_topk(counts, k, exclude_offset)
    for algo in algos:
        got_c, got_r = a.allpairs_topk(b, k, exclude_offset=exclude_offset, algo=algo)
        assert np.array_equal(got_r.cpu().numpy(), want_r), algo
        assert np.array_equal(got_c.cpu().numpy(), want_c), algo


@pytest.mark.parametrize("na,nb,width,active,k,excl", [
    (300, 1000, 2048, 40, 16, -1),      # partial panels on both sides
    (256, 2048, 2048, 40, 16, -1),      # whole panels, several j splits
    (700, 700, 2048, 40, 5, 0),         # self all-pairs, self match excluded
    (130, 900, 300, 60, 16, 200),       # a = rows 200.. of b, width not a multiple of 64
    (40, 9, 2048, 40, 16, -1),          # fewer candidates than k
    (64, 3000, 4096, 64, 1, -1),        # k = 1, width 4096, rows up to 64 ids
    (1000, 70, 100, 30, 3, -1),         # one candidate tile and a bit, short lists
])
def test_allpairs_topk_vs_oracle(na, nb, width, active, k, excl):
    rng = np.random.default_rng(na * 7919 + nb)
    b_csr = util.ragged_rows(rng, nb, width, active, False)
    if excl >= 0:
        off, ids, _ = b_csr
        lo, hi = excl, excl + na
        a_csr = ((off[lo:hi + 1] - off[lo]).astype(np.uint64), ids[int(off[lo]):int(off[hi])], None)
    else:
        a_csr = util.ragged_rows(rng, na, width, active, False)
    run_case(a_csr, b_csr, width, k, excl)


def test_allpairs_topk_long_rows_need_the_matrix_cores():
    """Rows longer than 64 ids do not fit the bit-sliced kernel's registers: the default falls
    back to the matrix-core kernel and forcing the bit-sliced one is refused."""
    from sdr_amd import SDRBatch, SdrbError
    rng = np.random.default_rng(8)
    a_csr = util.ragged_rows(rng, 70, 4096, 300, False)
    b_csr = util.ragged_rows(rng, 900, 4096, 300, False)
    run_case(a_csr, b_csr, 4096, 16, -1, algos=[None, "mfma"])
    a = SDRBatch.from_csr(*a_csr, width=4096)
    b = SDRBatch.from_csr(*b_csr, width=4096)
    with pytest.raises(SdrbError):
        a.allpairs_topk(b, 16, algo="bitslice")


def test_allpairs_topk_many_ties_and_duplicates():
    """Heavy ties: a small id universe makes many rows identical, so the order is decided by
    the row index almost everywhere."""
    rng = np.random.default_rng(3)
    b_csr = util.ragged_rows(rng, 1500, 12, 6, False)
    a_csr = util.ragged_rows(rng, 90, 12, 6, False)
    run_case(a_csr, b_csr, 12, 16, -1)


def test_allpairs_topk_ignores_payloads_and_reuses_dense_cache():
    import torch
    from sdr_amd import SDRBatch
    rng = np.random.default_rng(11)
    off, ids, data = util.ragged_rows(rng, 600, 2048, 40, True)
    unit = SDRBatch.from_csr(off, ids, data, width=2048)
    empty = SDRBatch.from_csr(off, ids, None, width=2048)
    c1, r1 = unit.allpairs_topk(unit, 8, exclude_offset=0)
    c2, r2 = empty.allpairs_topk(empty, 8, exclude_offset=0)
    c3, r3 = empty.allpairs_topk(empty, 8, exclude_offset=0)     # second call: cached byte matrix
    assert torch.equal(c1, c2) and torch.equal(r1, r2) and torch.equal(c2, c3) and torch.equal(r2, r3)
    # a new upload into the same handle must drop the cached matrix
    off2, ids2, _ = util.ragged_rows(rng, 600, 2048, 40, False)
    empty.upload(off2, ids2, None)
    fresh = SDRBatch.from_csr(off2, ids2, None, width=2048)
    c4, r4 = empty.allpairs_topk(empty, 8, exclude_offset=0)
    c5, r5 = fresh.allpairs_topk(fresh, 8, exclude_offset=0)
    assert torch.equal(c4, c5) and torch.equal(r4, r5)


def test_sharded_driver_allpairs_single_rank():
    """The shard driver's all-pairs path on one GPU: the shard is exported to wire form, adopted
    as the candidate batch, and every row's self match is excluded by the shard's row offset."""
    import oracle
    from sdr_amd.shard import HipEngine, ShardedDatabase
    rng = np.random.default_rng(21)
    off, ids, data = util.ragged_rows(rng, 900, 2048, 40, True)     # payloads travel too, and are ignored
    db = ShardedDatabase.from_global_csr(off, ids, data, width=2048, engine=HipEngine(0), rank=0, world=1)
    whole = db.gather_database()
    assert whole.stats() == db.shard.stats()
    got_c, got_r = db.allpairs_topk(16, candidates=whole)
    counts = oracle.port().size_op(oracle.AND, oracle.Batch(off, ids, None), oracle.Batch(off, ids, None))
    want_c, want_r = expected_topk(counts, 16, 0)
    assert np.array_equal(got_r.cpu().numpy(), want_r) and np.array_equal(got_c.cpu().numpy(), want_c)


def test_sharded_query_topk_full_size_one_rank_with_collectives():
    """4096 queries against a 1 Mi-row shard through ShardedDatabase.query_topk, with a one-rank RCCL group so
    that the candidate all-gather really runs: the [Q][N] count matrix (17 GB) is never built -- device memory
    taken beyond the inputs stays under 1 GB --, and sampled queries equal the oracle's order."""
    import socket
    import torch
    import torch.distributed as dist
    import oracle
    import sdr_amd
    from sdr_amd.shard import HipEngine, ShardedDatabase
    nq, nd, width, active, k = 4096, 1 << 20, 2048, 40, 8
    q_off, q_ids, _ = sdr_amd.synth_csr(2, 0, nq, width, active)
    d_off, d_ids, _ = sdr_amd.synth_csr(1, 0, nd, width, active)
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    dist.init_process_group("nccl", init_method=f"tcp://127.0.0.1:{port}", rank=0, world_size=1,
                            device_id=torch.device("cuda:0"))
    try:
        engine = HipEngine(0)
        db = ShardedDatabase.from_global_csr(d_off, d_ids, None, width=width, engine=engine, rank=0, world=1)
        queries = engine.pack_host(q_off, q_ids, None, width)
        warm = torch.ones(1, device="cuda:0")
        dist.all_reduce(warm)                      # RCCL's own buffers are not this path's memory
        torch.cuda.synchronize()
        torch.cuda.empty_cache()
        free_before, _ = torch.cuda.mem_get_info(0)
        counts, rows = db.query_topk(queries, k)
        torch.cuda.synchronize()
        free_after, _ = torch.cuda.mem_get_info(0)   # scratch and torch's cache are both still held: the peak
        assert free_before - free_after < (1 << 30), (free_before - free_after) / 2 ** 20
        assert counts.shape == (nq, k) and rows.dtype == torch.int64
        rng = np.random.default_rng(3)
        pick = np.sort(rng.choice(nq, 8, replace=False))
        lo = q_off[pick].astype(np.int64)
        sub_ids = np.concatenate([q_ids[a: a + active] for a in lo])
        sub_off = np.arange(len(pick) + 1, dtype=np.uint64) * active
        want = oracle.port().size_op(oracle.AND, oracle.Batch(sub_off, sub_ids, None), oracle.Batch(d_off, d_ids, None))
        want_c, want_r = expected_topk(want, k, -1)
        assert np.array_equal(counts.cpu().numpy()[pick], want_c) and np.array_equal(rows.cpu().numpy()[pick], want_r)
        # and the selection from existing counts (slabs of queries) agrees on a slice of the queries
        few = engine.pack_host(sub_off, sub_ids, None, width)
        c2, r2 = db.topk(db.size_op("ands", few), k)
        assert np.array_equal(c2.cpu().numpy(), want_c) and np.array_equal(r2.cpu().numpy(), want_r)
    finally:
        dist.destroy_process_group()


def test_candidate_row_limits_are_refused_before_any_work():
    """The merge of the partial top-k lists holds 512 keys a row: 8 splits of the matrix-core kernel (4 192 256
    candidate rows) or 32 of the bit-sliced one (16 775 168).  One candidate row more than the first limit: the
    matrix-core kernel refuses up front, the bit-sliced kernel still serves the call."""
    from sdr_amd import SDRBatch, SdrbError
    nb = 4_192_256 + 1
    ids = (np.arange(nb, dtype=np.int64) % 8).astype(np.int32)          # one id a row, ids 0..7 in turn
    off = np.arange(nb + 1, dtype=np.uint64)
    b = SDRBatch.from_csr(off, ids, None, width=8)
    a = SDRBatch.from_csr(np.array([0, 1, 3], np.uint64), np.array([5, 0, 7], np.int32), None, width=8)
    with pytest.raises(SdrbError) as e:
        a.allpairs_topk(b, 4, algo="mfma")
    assert e.value.code == -5 and "4 192 256" in str(e.value)
    counts, rows = a.allpairs_topk(b, 4)
    assert counts.cpu().numpy().tolist() == [[1, 1, 1, 1], [1, 1, 1, 1]]
    # ties go to the lower row: rows holding id 5, and rows holding id 0 or 7
    assert rows.cpu().numpy().tolist() == [[5, 13, 21, 29], [0, 7, 8, 15]]
