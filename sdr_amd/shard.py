"""Row-sharded database across the GPUs of one node, one process per GPU.

Every pair (query, database row) is independent, so the database is split into contiguous
row ranges, one per rank, each packed as its own CSR with rebased offsets.  The only
exchange on the data path is the query batch: the source rank broadcasts the packed query
CSR (header + one int32 buffer; 0.67 MB for 4096 queries of 40 ids) and every rank
evaluates it against its own shard.  Counts stay sharded -- rank r owns columns
[row_lo, row_hi) of the global [Q][N] result; all-gathering the full matrix is offered for
small problems only, since it moves Q*N*4 bytes per GPU over xGMI.

The collectives go through ``torch.distributed`` (backend "nccl" is RCCL on ROCm; "gloo"
on CPU for tests).  The per-shard computation is delegated to an *engine*; the default
engine is the HIP library.  Tests inject a CPU engine to exercise the sharding and the
collectives without a GPU.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional, Tuple

import numpy as np


@dataclass(frozen=True)
class QueryPlan:
    """Everything a receiving rank needs to size, receive and adopt a packed query batch without
    talking to the host of the sender: shape, payload kind and the validated batch's id statistics.
    A front end distributes it out of band (it is six integers); the bench derives it on every
    rank from the same synthetic generator."""
    n_rows: int
    nnz: int
    has_data: bool
    max_row_len: int
    min_id: int
    max_id: int

    @property
    def words(self) -> int:
        return (self.n_rows + 1) + self.nnz * (2 if self.has_data else 1)


@dataclass
class PackedQueries:
    """A validated query batch in its wire form: one int32 device tensor
    ``[offsets (n_rows+1) | ids (nnz) | payload bits (nnz, UnitData only)]`` plus its plan."""
    plan: QueryPlan
    tensor: "object"

    @classmethod
    def from_host_csr(cls, off, ids, data, *, width: int, engine):
        """Validates the batch by packing it once on this rank's device, then lays it out for
        the wire."""
        import torch
        probe = engine.pack_host(off, ids, data, width)
        st = probe.stats() if hasattr(probe, "stats") else _host_stats(off, ids, data)
        plan = QueryPlan(st["n_rows"], st["nnz"], bool(st["has_data"]), st["max_row_len"], st["min_id"], st["max_id"])
        parts = [np.asarray(off, dtype=np.uint64).astype(np.uint32).view(np.int32), np.asarray(ids, dtype=np.int32)]
        if data is not None:
            parts.append(np.asarray(data, dtype=np.float32).view(np.int32))
        tensor = torch.from_numpy(np.concatenate(parts)).to(engine.torch_device())
        return cls(plan, tensor)


def _host_stats(off, ids, data):
    off = np.asarray(off, dtype=np.uint64)
    lens = np.diff(off)
    return dict(n_rows=len(off) - 1, nnz=int(off[-1]), max_row_len=int(lens.max()) if len(lens) else 0,
                min_id=int(ids.min()) if len(ids) else 0, max_id=int(ids.max()) if len(ids) else -1,
                has_data=data is not None)


def shard_range(n_rows: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous, balanced row range of ``rank``: sizes differ by at most one row."""
    return n_rows * rank // world, n_rows * (rank + 1) // world


def slice_csr(off: np.ndarray, ids: np.ndarray, data: Optional[np.ndarray], lo: int, hi: int):
    """Rows [lo, hi) of a host CSR with offsets rebased to zero."""
    off = np.asarray(off, dtype=np.uint64)
    e0, e1 = int(off[lo]), int(off[hi])
    return (off[lo:hi + 1] - off[lo], ids[e0:e1], None if data is None else data[e0:e1])


class HipEngine:
    """Default engine: batches are :class:`sdr_amd.SDRBatch` objects on ``device``."""

    def __init__(self, device: int):
        self.device = device

    def torch_device(self):
        return f"cuda:{self.device}"

    def pack_host(self, off, ids, data, width):
        from .batch import SDRBatch
        return SDRBatch.from_csr(off, ids, data, width=width, device=self.device)

    def pack_file(self, path, row_lo, row_hi):
        from . import csrfile
        return csrfile.upload_rows(path, row_lo, row_hi, device=self.device)

    def pack_device(self, off_t, ids_t, data_t, width, into=None):
        from .batch import SDRBatch
        from ._lib import DATA_EMPTY, DATA_UNIT
        kind = DATA_UNIT if data_t is not None else DATA_EMPTY
        b = into if into is not None and into.data_kind == kind and into.width == width else None
        if b is None:
            b = SDRBatch(self.device, width, kind)
        b.upload_device(off_t, ids_t, data_t)   # reuses the batch's device storage when it fits
        return b

    def adopt(self, tensor, plan: "QueryPlan", width, into=None):
        """Zero-copy view of a received wire tensor as a query batch."""
        import torch
        from ._lib import DATA_EMPTY, DATA_UNIT
        from .batch import SDRBatch
        kind = DATA_UNIT if plan.has_data else DATA_EMPTY
        b = into if into is not None and into.data_kind == kind and into.width == width else None
        if b is None:
            b = SDRBatch(self.device, width, kind)
        nq, nnz = plan.n_rows, plan.nnz
        data_t = tensor[nq + 1 + nnz:].view(torch.float32) if plan.has_data else None
        b.adopt_device(tensor[: nq + 1], tensor[nq + 1: nq + 1 + nnz], data_t,
                       max_row_len=plan.max_row_len, min_id=plan.min_id, max_id=plan.max_id)
        return b

    def to_wire(self, batch):
        """(wire tensor, plan) of a batch that lives on this engine's device."""
        st = batch.stats()
        plan = QueryPlan(st["n_rows"], st["nnz"], bool(st["has_data"]), st["max_row_len"], st["min_id"], st["max_id"])
        return batch.to_wire(), plan

    def size_op(self, name: str, queries, shard, out=None):
        return getattr(queries, name)(shard, out=out)

    def elem_op(self, name: str, queries, shard):
        return getattr(queries, name)(shard)

    def allpairs_topk(self, rows, candidates, k, exclude_offset):
        return rows.allpairs_topk(candidates, k, exclude_offset=exclude_offset)


class ShardedDatabase:
    """One rank's view of a row-sharded SDR database."""

    def __init__(self, shard, row_lo: int, row_hi: int, n_rows_total: int, width: int, engine,
                 group=None):
        self.shard = shard
        self.row_lo, self.row_hi, self.n_rows_total = row_lo, row_hi, n_rows_total
        self.width = width
        self.engine = engine
        self.group = group

    # -- construction ---------------------------------------------------------------
    @classmethod
    def from_global_csr(cls, off, ids, data, *, width: int, engine, rank: int, world: int, group=None):
        """Every rank holds (or can generate) the whole host CSR and packs only its rows."""
        n = len(off) - 1
        lo, hi = shard_range(n, rank, world)
        s_off, s_ids, s_data = slice_csr(off, ids, data, lo, hi)
        return cls(engine.pack_host(s_off, s_ids, s_data, width), lo, hi, n, width, engine, group)

    @classmethod
    def from_file(cls, path, *, engine, rank: int, world: int, group=None):
        """Every rank maps the same CSR database file and packs only its row range, straight from
        the mapping (no rank ever holds the whole database in host memory)."""
        from . import csrfile
        meta = csrfile.info(path)
        lo, hi = shard_range(meta["n_rows"], rank, world)
        return cls(engine.pack_file(path, lo, hi), lo, hi, meta["n_rows"], meta["width"], engine, group)

    # -- the exchange step ------------------------------------------------------------
    def broadcast_queries(self, packed=None, src: int = 0, force_collective: bool = False, into=None):
        """Broadcasts a packed query batch from ``src`` and packs it on every rank.

        ``packed`` (needed on ``src`` only) is ``(offsets, ids, data_or_None)`` as torch tensors
        on the engine's device: int32 offsets [nq+1], int32 ids, float32 payloads.
        ``force_collective`` runs the collectives even in a one-rank group (used to rehearse the
        RCCL path on a single GPU).  ``into`` recycles an earlier query batch of the same payload
        kind (its device storage is reused when large enough).
        """
        import torch
        import torch.distributed as dist

        dev = self.engine.torch_device()
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        rank = dist.get_rank(self.group) if dist.is_initialized() else 0
        if world == 1 and not (force_collective and dist.is_initialized()):
            off_t, ids_t, data_t = packed
            return self._pack(off_t, ids_t, data_t, into)
        header = torch.zeros(3, dtype=torch.int64, device=dev)
        src_global = _global_rank(self.group, src)
        if rank == src:
            off_t, ids_t, data_t = packed
            header[0], header[1], header[2] = off_t.numel() - 1, ids_t.numel(), int(data_t is not None)
        dist.broadcast(header, src=src_global, group=self.group)
        nq, nnz, has_data = (int(x) for x in header.tolist())
        total = (nq + 1) + nnz * (2 if has_data else 1)
        if rank == src:
            parts = [off_t, ids_t] + ([data_t.view(torch.int32)] if has_data else [])
            buf = torch.cat(parts)
        else:
            buf = torch.empty(total, dtype=torch.int32, device=dev)
        dist.broadcast(buf, src=src_global, group=self.group)
        off_t = buf[: nq + 1]
        ids_t = buf[nq + 1: nq + 1 + nnz]
        data_t = buf[nq + 1 + nnz:].view(torch.float32) if has_data else None
        return self._pack(off_t.contiguous(), ids_t.contiguous(),
                          None if data_t is None else data_t.contiguous(), into)

    def broadcast_packed(self, packed: Optional[PackedQueries], plan: QueryPlan, src: int = 0,
                         force_collective: bool = False, recv=None, into=None):
        """Fast path of the exchange step when every rank knows the plan of the batch: ONE
        collective (the wire tensor), no host round trip, and the received tensor is adopted as
        the query batch without copying or re-validating it (the sender validated it).
        ``recv`` is an optional preallocated int32 tensor of ``plan.words`` elements."""
        import torch
        import torch.distributed as dist

        on = dist.is_initialized()
        world = dist.get_world_size(self.group) if on else 1
        rank = dist.get_rank(self.group) if on else 0
        if rank == src:
            tensor = packed.tensor
            if tensor.numel() != plan.words:
                raise ValueError("packed tensor does not match the plan")
        else:
            tensor = recv if recv is not None else torch.empty(plan.words, dtype=torch.int32,
                                                               device=self.engine.torch_device())
        if world > 1 or (force_collective and on):
            dist.broadcast(tensor, src=_global_rank(self.group, src), group=self.group)
        return self.engine.adopt(tensor, plan, self.width, into=into)

    # -- all-pairs: every rank needs every row as a candidate ---------------------------------
    def gather_database(self):
        """All-gathers the shards' CSR (wire form, padded to the largest shard; 164 B per 40-id
        row) and returns a batch that holds the WHOLE database in global row order on this rank.
        Two collectives: the shards' plans (six integers each), then the wire tensors."""
        import torch
        import torch.distributed as dist

        wire, plan = self.engine.to_wire(self.shard)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return self.engine.adopt(wire, plan, self.width)
        dev = self.engine.torch_device()
        mine = torch.tensor([plan.n_rows, plan.nnz, int(plan.has_data), plan.max_row_len, plan.min_id, plan.max_id],
                            dtype=torch.int64, device=dev)
        plans_t = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(plans_t, mine, group=self.group)
        plans = [QueryPlan(*(int(x) for x in t.tolist()[:2]), bool(int(t[2])), *(int(x) for x in t.tolist()[3:]))
                 for t in plans_t]
        if len({p.has_data for p in plans}) != 1:
            raise ValueError("shards disagree on the payload kind")
        words = max(p.words for p in plans)
        padded = torch.zeros(words, dtype=torch.int32, device=dev)
        padded[: wire.numel()] = wire
        wires = [torch.empty_like(padded) for _ in range(world)]
        dist.all_gather(wires, padded, group=self.group)
        # stitch: offsets rebased by the elements of the shards before, then ids, then payloads
        offs, ids, datas, base = [], [], [], 0
        for p, w in zip(plans, wires):
            o = (w[: p.n_rows + 1].to(torch.int64) & 0xFFFFFFFF) + base     # wire offsets are uint32 bit patterns
            offs.append(o if not offs else o[1:])
            ids.append(w[p.n_rows + 1: p.n_rows + 1 + p.nnz])
            if p.has_data:
                datas.append(w[p.n_rows + 1 + p.nnz: p.n_rows + 1 + 2 * p.nnz])
            base += p.nnz
        if base > 0xFFFFFFFF:
            raise OverflowError("the gathered database exceeds 2^32-1 elements")
        # (offsets travel as the low 32 bits; the batch reads them as uint32)
        off_all = torch.cat(offs)
        off_all = torch.where(off_all >= 2 ** 31, off_all - 2 ** 32, off_all).to(torch.int32)
        whole = torch.cat([off_all] + ids + datas)
        with_elems = [p for p in plans if p.nnz]
        total = QueryPlan(sum(p.n_rows for p in plans), base, plans[0].has_data,
                          max(p.max_row_len for p in plans),
                          min((p.min_id for p in with_elems), default=0),
                          max((p.max_id for p in with_elems), default=-1))
        return self.engine.adopt(whole, total, self.width)

    def allpairs_topk(self, k: int, exclude_self: bool = True, candidates=None):
        """For every row of this rank's shard: the k rows of the WHOLE database with the largest id
        overlap, as (counts, global rows), overlap descending then row ascending.  The shards'
        rows are disjoint, so results need no merge; ``candidates`` (from :meth:`gather_database`)
        can be reused across calls."""
        if candidates is None:
            candidates = self.gather_database()
        return self.engine.allpairs_topk(self.shard, candidates, k, self.row_lo if exclude_self else -1)

    def _pack(self, off_t, ids_t, data_t, into):
        if into is not None:
            try:
                return self.engine.pack_device(off_t, ids_t, data_t, self.width, into=into)
            except TypeError:   # engines without storage reuse
                pass
        return self.engine.pack_device(off_t, ids_t, data_t, self.width)

    # -- per-shard evaluation ----------------------------------------------------------
    def size_op(self, name: str, queries, out=None):
        """``name`` in {ands, ors, xors, rms}: counts of ``queries`` x this rank's rows, i.e.
        columns [row_lo, row_hi) of the global result.  No communication."""
        return self.engine.size_op(name, queries, self.shard, out=out)

    def elem_op(self, name: str, queries):
        """``name`` in {ande, ore, xore, rme}: the element results of ``queries`` x this rank's rows
        stay on this rank (CSR over the local pairs).  Only the shards' element totals travel
        (one int64 per rank), giving the position of this shard's elements in the global result:
        returns ``(local_result, global_element_offset, global_element_total)``."""
        import torch
        import torch.distributed as dist

        res = self.engine.elem_op(name, queries, self.shard)
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return res, 0, int(res.total)
        rank = dist.get_rank(self.group)
        mine = torch.tensor([int(res.total)], dtype=torch.int64, device=self.engine.torch_device())
        totals = [torch.empty_like(mine) for _ in range(world)]
        dist.all_gather(totals, mine, group=self.group)
        totals = [int(t) for t in totals]
        return res, sum(totals[:rank]), sum(totals)

    def gather_counts(self, local_counts):
        """All-gathers the per-shard count slabs into the full [Q][N] matrix on every rank.
        Moves Q*N*4 bytes per GPU: meant for small problems and tests, not the timed path."""
        import torch
        import torch.distributed as dist

        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return local_counts
        nq = local_counts.shape[0]
        slabs = []
        for r in range(world):
            lo, hi = shard_range(self.n_rows_total, r, world)
            slabs.append(torch.empty((nq, hi - lo), dtype=local_counts.dtype, device=local_counts.device))
        if _equal_shapes(slabs):
            dist.all_gather(slabs, local_counts.contiguous(), group=self.group)
        else:
            _all_gather_ragged(slabs, local_counts.contiguous(), self.group)
        return torch.cat(slabs, dim=1)

    def _merge_topk(self, counts, rows, k: int):
        """Per query, the best k of the candidates (count int64 [Q][m], global row int64 [Q][m], row -1 = empty
        slot) by count descending, then row ascending.  m is a few dozen: the sort key is a small tensor."""
        import torch
        span = self.n_rows_total + 1
        key = torch.where(rows >= 0, counts * span + (self.n_rows_total - rows), torch.full_like(counts, -1))
        kk = min(k, key.shape[1])
        top_key, _ = torch.topk(key, kk, dim=1)
        if kk < k:
            top_key = torch.cat([top_key, torch.full((key.shape[0], k - kk), -1, dtype=torch.int64, device=key.device)], dim=1)
        valid = top_key >= 0
        out_counts = torch.where(valid, top_key // span, torch.zeros_like(top_key))
        out_rows = torch.where(valid, self.n_rows_total - top_key % span, torch.full_like(top_key, -1))
        return out_counts, out_rows

    def _gather_candidates(self, counts, rows):
        """All ranks' per-query candidate lists side by side: [Q][world * k]."""
        import torch
        import torch.distributed as dist
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return counts, rows
        both = torch.stack([counts, rows]).contiguous()          # one collective for counts and rows
        gathered = [torch.empty_like(both) for _ in range(world)]
        dist.all_gather(gathered, both, group=self.group)
        return torch.cat([g[0] for g in gathered], dim=1), torch.cat([g[1] for g in gathered], dim=1)

    def query_topk(self, queries, k: int):
        """Per query, the k rows of the WHOLE database with the largest id overlap (``ands`` on id-only views;
        count descending, then lower row).  Each rank runs the fused all-pairs top-k kernel on its shard -- the
        [Q][rows] count matrix is never written --, rebases the rows it found, and the Q*k candidates of every
        rank are all-gathered (Q*k*16 bytes per rank) and merged.  Returns int64 ``(counts, rows)`` [Q][k],
        empty slots (0, -1).  The fused kernel keeps 16 candidates per query: a larger k goes through the count
        matrix of ``ands`` instead (same order for id-only batches).  The reference has no top-k; this rule is
        this engine's own."""
        import torch
        kk = min(k, 16)
        if kk < k:
            return self.topk(self.size_op("ands", queries), k)    # the fused kernel keeps at most 16 per query
        try:
            c, r = self.engine.allpairs_topk(queries, self.shard, kk, -1)
        except Exception as e:          # shapes the fused kernels do not take (widths beyond 16 384, very long rows)
            if getattr(e, "code", None) != -5:
                raise
            return self.topk(self.size_op("ands", queries.ids_only() if hasattr(queries, "ids_only") else queries), k)
        c, r = c.to(torch.int64), r.to(torch.int64)
        r = torch.where(r >= 0, r + self.row_lo, r)
        c, r = self._gather_candidates(c, r)
        return self._merge_topk(c, r, k)

    def topk(self, local_counts, k: int):
        """The same selection from counts that already exist (``size_op`` output of this rank's shard): local
        top-k per query -- in slabs of queries, so that the sort keys never take more than ~256 MB --, then the
        all-gather and merge of :meth:`query_topk`."""
        import torch
        n_local = local_counts.shape[1]
        kk = min(k, n_local)
        rows = torch.arange(self.row_lo, self.row_lo + n_local, device=local_counts.device, dtype=torch.int64)
        span = self.n_rows_total + 1
        slab = max(1, (32 << 20) // max(n_local, 1))
        best_c, best_r = [], []
        for q0 in range(0, local_counts.shape[0], slab):
            part = local_counts[q0: q0 + slab].to(torch.int64)
            key = part * span + (self.n_rows_total - rows)[None, :]      # count descending, then row ascending
            top_key, _ = torch.topk(key, kk, dim=1)
            best_c.append(top_key // span)
            best_r.append(self.n_rows_total - top_key % span)
        c, r = torch.cat(best_c), torch.cat(best_r)
        if kk < k:
            pad = k - kk
            c = torch.cat([c, torch.zeros((c.shape[0], pad), dtype=torch.int64, device=c.device)], dim=1)
            r = torch.cat([r, torch.full((r.shape[0], pad), -1, dtype=torch.int64, device=r.device)], dim=1)
        c, r = self._gather_candidates(c, r)
        return self._merge_topk(c, r, k)


def _global_rank(group, group_rank: int) -> int:
    """``src`` of the collectives is a rank of the whole job; callers here count within ``group``."""
    import torch.distributed as dist
    return group_rank if group is None else dist.get_global_rank(group, group_rank)


def _equal_shapes(tensors) -> bool:
    return all(t.shape == tensors[0].shape for t in tensors)


def _all_gather_ragged(slabs, mine, group):
    """all_gather for slabs whose widths differ by one row (unbalanced split)."""
    import torch.distributed as dist
    for r, slab in enumerate(slabs):
        if r == dist.get_rank(group):
            slab.copy_(mine)
        dist.broadcast(slab, src=_global_rank(group, r), group=group)
