"""Multi-process CPU tests (gloo, world_size 2) of the row-sharded driver: partitioning, the
query broadcast, per-shard evaluation, count gathering and the top-k merge.  The per-shard
computation is done by a CPU engine that this test injects (the oracle); the HIP engine is
exercised by the GPU tests."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)

import util  # noqa: E402
from sdr_amd.shard import PackedQueries, ShardedDatabase, shard_range, slice_csr  # noqa: E402

OPS = {"ands": 0, "ors": 1, "xors": 2, "rms": 3}


class OracleEngine:
    """Test-only engine: batches are oracle.Batch objects, evaluation is the CPU oracle."""

    def torch_device(self):
        return "cpu"

    def pack_host(self, off, ids, data, width):
        import oracle
        return oracle.Batch(off, ids, data)

    def pack_file(self, path, row_lo, row_hi):
        import oracle
        from sdr_amd import csrfile
        return oracle.Batch(*csrfile.read_rows(path, row_lo, row_hi))

    def pack_device(self, off_t, ids_t, data_t, width):
        import oracle
        return oracle.Batch(off_t.numpy().astype(np.uint64), ids_t.numpy(),
                            None if data_t is None else data_t.numpy())

    def adopt(self, tensor, plan, width, into=None):
        nq, nnz = plan.n_rows, plan.nnz
        data_t = tensor[nq + 1 + nnz:].view(torch.float32) if plan.has_data else None
        return self.pack_device(tensor[: nq + 1], tensor[nq + 1: nq + 1 + nnz], data_t, width)

    def to_wire(self, batch):
        off, ids, data = batch.csr()
        wire = PackedQueries.from_host_csr(off, ids, data, width=0, engine=self)
        return wire.tensor, wire.plan

    def allpairs_topk(self, rows, candidates, k, exclude_offset):
        import oracle
        counts = oracle.port().size_op(0, oracle.Batch(rows.csr()[0], rows.csr()[1], None),
                                       oracle.Batch(candidates.csr()[0], candidates.csr()[1], None)).astype(np.int64)
        out_c = np.zeros((counts.shape[0], k), np.int64)
        out_r = np.full((counts.shape[0], k), -1, np.int64)
        for i in range(counts.shape[0]):
            cand = np.arange(counts.shape[1])
            if exclude_offset >= 0:
                cand = cand[cand != i + exclude_offset]
            order = cand[np.lexsort((cand, -counts[i][cand]))][:k]
            out_c[i, : len(order)], out_r[i, : len(order)] = counts[i][order], order
        return torch.from_numpy(out_c), torch.from_numpy(out_r)

    def elem_op(self, name, queries, shard):
        import oracle
        from types import SimpleNamespace
        off, ids, data = oracle.port().elem_op({"ande": 0, "ore": 1, "xore": 2, "rme": 3}[name], queries, shard)
        return SimpleNamespace(offsets=off, ids=ids, data=data, total=int(off[-1]))

    def size_op(self, name, queries, shard, out=None):
        import oracle
        return torch.from_numpy(oracle.port().size_op(OPS[name], queries, shard).astype(np.int32))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, with_data, n_rows, results, tmpdir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        rng = np.random.default_rng(5)  # same global database on every rank
        d_off, d_ids, d_data = util.ragged_rows(rng, n_rows, 300, 40, with_data)
        q_off, q_ids, q_data = util.ragged_rows(rng, 9, 300, 40, with_data)
        db = ShardedDatabase.from_global_csr(d_off, d_ids, d_data, width=300, engine=OracleEngine(),
                                             rank=rank, world=world)
        assert (db.row_lo, db.row_hi) == shard_range(n_rows, rank, world)
        # the same shard, packed from a database file that rank 0 wrote
        from sdr_amd import csrfile
        path = os.path.join(tmpdir, "db.sdrb")
        if rank == 0:
            csrfile.write(path, d_off, d_ids, d_data, width=300)
        dist.barrier()
        from_file = ShardedDatabase.from_file(path, engine=OracleEngine(), rank=rank, world=world)
        assert (from_file.row_lo, from_file.row_hi, from_file.n_rows_total, from_file.width) == \
            (db.row_lo, db.row_hi, n_rows, 300)
        for got, want in zip(from_file.shard.csr(), db.shard.csr()):
            assert (got is None and want is None) or np.array_equal(got, want)
        packed = None
        if rank == 0:
            packed = (torch.from_numpy(q_off.astype(np.int64).astype(np.int32)), torch.from_numpy(q_ids),
                      None if q_data is None else torch.from_numpy(q_data))
        queries = db.broadcast_queries(packed, src=0)
        out = {}
        for name in OPS:
            local = db.size_op(name, queries)
            assert local.shape == (9, db.row_hi - db.row_lo)
            out[name] = db.gather_counts(local).numpy()
        # the one-collective path: every rank knows the plan, only the wire tensor travels
        wire = PackedQueries.from_host_csr(q_off, q_ids, q_data, width=300, engine=OracleEngine())
        assert wire.plan.words == wire.tensor.numel()
        again = db.broadcast_packed(wire if rank == 0 else None, wire.plan, src=0)
        assert np.array_equal(db.gather_counts(db.size_op("xors", again)).numpy(), out["xors"])
        # element ops: results stay sharded, only the totals travel
        res, elem_lo, elem_total = db.elem_op("xore", queries)
        results[f"xore_{rank}"] = (db.row_lo, db.row_hi, res.offsets, res.ids, elem_lo, elem_total)
        # all-pairs: the shards are all-gathered into the whole database on every rank
        whole = db.gather_database()
        w_off, w_ids, w_data = whole.csr()
        assert np.array_equal(w_off, d_off) and np.array_equal(w_ids, d_ids)
        assert (w_data is None) == (d_data is None) and (w_data is None or np.array_equal(w_data, d_data))
        ap_c, ap_r = db.allpairs_topk(4, candidates=whole)
        out[f"allpairs_{rank}"] = (db.row_lo, ap_c.numpy(), ap_r.numpy())
        results[f"allpairs_{rank}"] = out[f"allpairs_{rank}"]
        counts, rows = db.topk(db.size_op("ands", queries), 5)
        out["topk_counts"], out["topk_rows"] = counts.numpy(), rows.numpy()
        # the same selection straight from the queries (fused all-pairs top-k per shard, no count matrix); it ranks
        # by id overlap, so with payloads the expectation comes from the id-only views of both sides
        import oracle
        id_only = lambda b: oracle.Batch(b.csr()[0], b.csr()[1], None)
        id_counts = db.engine.size_op("ands", id_only(queries), id_only(db.shard))
        want_counts, want_rows = db.topk(id_counts, 5)
        if not with_data:
            assert torch.equal(want_counts, counts) and torch.equal(want_rows, rows)
        f_counts, f_rows = db.query_topk(queries, 5)
        assert torch.equal(f_counts, want_counts) and torch.equal(f_rows, want_rows)
        w_counts, w_rows = db.query_topk(queries, 20)       # more than the fused kernel keeps: falls back to counts
        assert torch.equal(w_counts[:, :5], counts) and torch.equal(w_rows[:, :5], rows)
        if rank == 0:
            results.update(out)
        else:  # every rank must hold the same gathered result
            results["rank1_ands"] = out["ands"]
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("with_data,n_rows", [(False, 101), (True, 64)])
def test_two_rank_sharded_size_ops(with_data, n_rows, tmp_path):
    import oracle
    world = 2
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_worker, args=(world, _free_port(), with_data, n_rows, results, str(tmp_path)), nprocs=world, join=True)
    rng = np.random.default_rng(5)
    d = oracle.Batch(*util.ragged_rows(rng, n_rows, 300, 40, with_data))
    q = oracle.Batch(*util.ragged_rows(rng, 9, 300, 40, with_data))
    for name, op in OPS.items():
        want = oracle.port().size_op(op, q, d)
        assert np.array_equal(results[name].view(np.uint32), want), name
    assert np.array_equal(results["rank1_ands"], results["ands"])
    # sharded element op: every shard's CSR equals the oracle on its columns, and the element
    # offsets tile the global total
    x_off, x_ids, _ = oracle.port().elem_op(2, q, d)
    seen = 0
    for rank in range(world):
        lo, hi, off, ids, elem_lo, elem_total = results[f"xore_{rank}"]
        assert elem_lo == seen and elem_total == sum(len(results[f"xore_{r}"][3]) for r in range(world))
        seen += len(ids)
        for i in range(9):
            for j in range(lo, hi):
                want = x_ids[int(x_off[i * n_rows + j]):int(x_off[i * n_rows + j + 1])]
                k = i * (hi - lo) + (j - lo)
                assert np.array_equal(ids[int(off[k]):int(off[k + 1])], want)
    # all-pairs top-4 of every row over the whole database, self match excluded
    self_counts = oracle.port().size_op(0, oracle.Batch(d.csr()[0], d.csr()[1], None),
                                        oracle.Batch(d.csr()[0], d.csr()[1], None)).astype(np.int64)
    for rank in range(world):
        lo, got_c, got_r = results[f"allpairs_{rank}"]
        for n in range(got_r.shape[0]):
            i = lo + n
            cand = [r for r in range(n_rows) if r != i]
            order = sorted(cand, key=lambda r: (-self_counts[i, r], r))[:4]
            assert got_r[n].tolist() == order and got_c[n].tolist() == [int(self_counts[i, r]) for r in order]
    # top-k: count descending, ties to the lower row index
    want = oracle.port().size_op(0, q, d).astype(np.int64)
    for i in range(9):
        order = sorted(range(n_rows), key=lambda r: (-want[i, r], r))[:5]
        assert results["topk_rows"][i].tolist() == order
        assert results["topk_counts"][i].tolist() == [int(want[i, r]) for r in order]


def test_slice_csr_rebases_offsets():
    off = np.array([0, 2, 2, 5, 9], dtype=np.uint64)
    ids = np.arange(9, dtype=np.int32)
    s_off, s_ids, s_data = slice_csr(off, ids, None, 1, 3)
    assert s_off.tolist() == [0, 0, 3] and s_ids.tolist() == [2, 3, 4] and s_data is None
    parts = [shard_range(10, r, 3) for r in range(3)]
    assert parts == [(0, 3), (3, 6), (6, 10)]


def _subgroup_worker(rank, world, port, results):
    """Ranks 1 and 2 of a three-rank job form the database's group: a rank within the group is not its rank in
    the job, which is what the collectives' ``src`` must be given."""
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        group = dist.new_group([1, 2])
        if rank == 0:
            return
        g_rank = dist.get_rank(group)
        rng = np.random.default_rng(8)
        d_off, d_ids, _ = util.ragged_rows(rng, 33, 200, 20, False)      # 16 + 17 rows: the ragged gather
        q_off, q_ids, _ = util.ragged_rows(rng, 5, 200, 20, False)
        db = ShardedDatabase.from_global_csr(d_off, d_ids, None, width=200, engine=OracleEngine(),
                                             rank=g_rank, world=2, group=group)
        packed = None
        if g_rank == 1:      # the source is the group's second rank (job rank 2)
            packed = (torch.from_numpy(q_off.astype(np.int64).astype(np.int32)), torch.from_numpy(q_ids), None)
        queries = db.broadcast_queries(packed, src=1)
        full = db.gather_counts(db.size_op("ands", queries))
        wire = PackedQueries.from_host_csr(q_off, q_ids, None, width=200, engine=OracleEngine())
        again = db.broadcast_packed(wire if g_rank == 1 else None, wire.plan, src=1)
        assert np.array_equal(db.gather_counts(db.size_op("ands", again)).numpy(), full.numpy())
        c, r = db.query_topk(queries, 3)
        whole = db.gather_database()
        assert np.array_equal(whole.csr()[0], d_off) and np.array_equal(whole.csr()[1], d_ids)
        results[f"counts_{g_rank}"] = full.numpy()
        results[f"topk_{g_rank}"] = (c.numpy(), r.numpy())
    finally:
        dist.destroy_process_group()


def test_database_group_inside_a_larger_job():
    import oracle
    mgr = mp.Manager()
    results = mgr.dict()
    mp.spawn(_subgroup_worker, args=(3, _free_port(), results), nprocs=3, join=True)
    rng = np.random.default_rng(8)
    d = oracle.Batch(*util.ragged_rows(rng, 33, 200, 20, False))
    q = oracle.Batch(*util.ragged_rows(rng, 5, 200, 20, False))
    want = oracle.port().size_op(0, q, d)
    for g_rank in range(2):
        assert np.array_equal(results[f"counts_{g_rank}"].view(np.uint32), want)
        c, r = results[f"topk_{g_rank}"]
        for i in range(5):
            order = sorted(range(33), key=lambda n: (-int(want[i, n]), n))[:3]
            assert r[i].tolist() == order and c[i].tolist() == [int(want[i, n]) for n in order]


def test_gathered_offsets_are_read_as_uint32():
    """Wire offsets at or above 2^31 travel as negative int32; the stitch in gather_database must read them as
    unsigned before rebasing."""
    w = torch.tensor([0, -(2 ** 31), -1], dtype=torch.int32)
    assert ((w.to(torch.int64) & 0xFFFFFFFF) + 5).tolist() == [5, 2 ** 31 + 5, 2 ** 32 + 4]
