"""BASELINE.json configs 4 and 5 at one GPU's full share, under pytest:
config 4 -- UnitData, width 16384, 328 active, 64 queries x 128 Ki rows (the 32-row mask table at full length, its
            fill and sort; the unions in chunks);
config 5 -- all-pairs overlap + top-16, 32 Ki rows against 256 Ki rows, self match excluded, both kernels (splits,
            merge of the partial lists, XCD mapping)."""
import numpy as np
import pytest

import oracle
import sdr_amd
from oracle import Batch
from sdr_amd import SDRBatch

pytestmark = pytest.mark.gpu


def sub_batch(csr, rows):
    off, ids, data = csr
    lens = (off[rows + 1] - off[rows]).astype(np.int64)
    new_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    pick = lambda arr: np.concatenate([arr[int(off[r]):int(off[r + 1])] for r in rows])   # noqa: E731
    return Batch(new_off, pick(ids), None if data is None else pick(data))


def test_config4_unitdata_full_share():
    import torch
    w, act, nq, nd = 16384, 328, 64, 1 << 17
    q_csr = sdr_amd.synth_csr(2, 0, nq, w, act, True)
    d_csr = sdr_amd.synth_csr(1, 0, nd, w, act, True)
    q = SDRBatch.from_csr(*q_csr, width=w)
    d = SDRBatch.from_csr(*d_csr, width=w)
    oq = Batch(*q_csr)
    rng = np.random.default_rng(4)
    rows = np.sort(rng.choice(nd, 200, replace=False))
    sub_d = sub_batch(d_csr, rows)
    cols = torch.as_tensor(rows, device="cuda:0")

    # sizes: closed form where one exists (ors never looks at payloads), the oracle on 200 sampled rows for all four
    hq = np.bincount(q_csr[1], minlength=w).astype(np.int64)
    hd = np.bincount(d_csr[1], minlength=w).astype(np.int64)
    matches = int((hq * hd).sum())
    sizes = {}
    for op, name in enumerate(("ands", "ors", "xors", "rms")):
        sizes[op] = getattr(q, name)(d)
        got = sizes[op][:, cols].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, oracle.port().size_op(op, oq, sub_d)), name
    assert int(sizes[1].sum(dtype=torch.int64)) == 2 * nq * nd * act - matches
    assert abs(matches / (nq * nd) - act * act / w) < 0.05      # the generator's sanity checksum

    # ande: lengths equal ands for EVERY pair; ids and payload bits equal the oracle's on the sampled rows
    res = q.ande(d)
    assert bool(torch.equal((res.offsets[1:] - res.offsets[:-1]).view(nq, nd), sizes[0]))
    assert res.total == int(sizes[0].sum(dtype=torch.int64))
    w_off, w_ids, w_data = oracle.port().elem_op(oracle.AND, oq, sub_d)
    off = res.offsets
    pair = torch.as_tensor(np.add.outer(np.arange(nq) * nd, rows).reshape(-1), device="cuda:0")
    lo = off[pair].cpu().numpy().view(np.uint32).astype(np.int64)
    hi = off[pair + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    assert np.array_equal(hi - lo, np.diff(w_off).astype(np.int64))
    ids, data = res.ids.cpu().numpy(), res.data.cpu().numpy()
    got_ids = np.concatenate([ids[a:b] for a, b in zip(lo, hi)])
    got_data = np.concatenate([data[a:b] for a, b in zip(lo, hi)])
    assert np.array_equal(got_ids, w_ids)
    assert np.array_equal(got_data.view(np.uint32), w_data.view(np.uint32))
    del res

    # rme (2.7e9 elements, one call) on a few sampled rows: lengths against rms everywhere
    res = q.rme(d)
    assert bool(torch.equal((res.offsets[1:] - res.offsets[:-1]).view(nq, nd), sizes[3]))
    w_off, w_ids, w_data = oracle.port().elem_op(oracle.RM, oq, sub_batch(d_csr, rows[:8]))
    pair = torch.as_tensor(np.add.outer(np.arange(nq) * nd, rows[:8]).reshape(-1), device="cuda:0")
    lo = res.offsets[pair].cpu().numpy().view(np.uint32).astype(np.int64)
    hi = res.offsets[pair + 1].cpu().numpy().view(np.uint32).astype(np.int64)
    got_ids = np.concatenate([res.ids_buffer[a:b].cpu().numpy() for a, b in zip(lo, hi)])
    got_data = np.concatenate([res.data_buffer[a:b].cpu().numpy() for a, b in zip(lo, hi)])
    assert np.array_equal(got_ids, w_ids)
    assert np.array_equal(got_data.view(np.uint32), w_data.view(np.uint32))


def test_config5_allpairs_full_share():
    na, nb, w, act, k = 1 << 15, 1 << 18, 2048, 40, 16
    off, ids, _ = sdr_amd.synth_csr(1, 0, nb, w, act)
    b = SDRBatch.from_csr(off, ids, None, width=w)
    a = SDRBatch.from_csr(off[: na + 1].copy(), ids[: int(off[na])], None, width=w)
    c_bs, r_bs = a.allpairs_topk(b, k, exclude_offset=0, algo="bitslice")
    c_mm, r_mm = a.allpairs_topk(b, k, exclude_offset=0, algo="mfma")
    c_bs, r_bs, c_mm, r_mm = (t.cpu().numpy() for t in (c_bs, r_bs, c_mm, r_mm))
    # the two kernels agree everywhere
    assert np.array_equal(c_bs, c_mm) and np.array_equal(r_bs, r_mm)
    # no row lists itself, lists are sorted by (count descending, row ascending), every slot is filled
    assert not (r_bs == np.arange(na)[:, None]).any()
    assert (r_bs >= 0).all() and (r_bs < nb).all()
    assert (np.diff(c_bs.astype(np.int64), axis=1) <= 0).all()
    ties = np.diff(c_bs.astype(np.int64), axis=1) == 0
    assert (np.diff(r_bs.astype(np.int64), axis=1)[ties] > 0).all()
    # the oracle's counts against ALL rows for 16 sampled rows of a, ranked by the engine's rule
    pick = np.linspace(0, na - 1, 16).astype(np.int64)
    want = oracle.port().size_op(oracle.AND, sub_batch((off, ids, None), pick), Batch(off, ids, None))
    for n, i in enumerate(pick):
        c = want[n].astype(np.int64)
        cand = np.arange(nb)
        cand = cand[cand != i]
        order = cand[np.lexsort((cand, -c[cand]))][:k]
        assert np.array_equal(r_bs[i], order), i
        assert np.array_equal(c_bs[i], c[order]), i


def test_config4_unions_in_chunks():
    """ore / xore of 64 queries x 128 Ki rows hold 5.4e9 elements, more than one CSR with 32-bit offsets addresses:
    the host layer cuts the rows of A into chunks.  Lengths against the size ops for every pair of a chunk, ids
    and payload bits against the oracle on sampled pairs of every chunk."""
    import torch
    w, act, nq, nd = 16384, 328, 64, 1 << 17
    q_csr = sdr_amd.synth_csr(2, 0, nq, w, act, True)
    d_csr = sdr_amd.synth_csr(1, 0, nd, w, act, True)
    q = SDRBatch.from_csr(*q_csr, width=w)
    d = SDRBatch.from_csr(*d_csr, width=w)
    rng = np.random.default_rng(44)
    rows = np.sort(rng.choice(nd, 6, replace=False))
    sub_d = sub_batch(d_csr, rows)
    for op, name, size_name in ((oracle.OR, "ore", "ors"), (oracle.XOR, "xore", "xors")):
        sizes = getattr(q, size_name)(d)
        res = getattr(q, name)(d)
        assert isinstance(res, sdr_amd.ChunkedElemResult) and len(res.chunks) >= 2
        assert res.starts[0] == 0 and res.starts[-1] == nq
        assert res.total == int(sizes.sum(dtype=torch.int64))
        w_off, w_ids, w_data = oracle.port().elem_op(op, Batch(*q_csr), sub_d)
        for c, chunk in enumerate(res.chunks):
            i0, i1 = res.starts[c], res.starts[c + 1]
            assert bool(torch.equal((chunk.offsets[1:] - chunk.offsets[:-1]).view(i1 - i0, nd), sizes[i0:i1]))
        for i in (0, 1, res.starts[1] - 1, res.starts[1], nq - 1):
            for k, j in enumerate(rows):
                ids, data = res.pair(i, int(j))
                p = i * len(rows) + k
                assert np.array_equal(ids.cpu().numpy(), w_ids[int(w_off[p]):int(w_off[p + 1])]), (name, i, j)
                assert np.array_equal(data.cpu().numpy().view(np.uint32),
                                      w_data[int(w_off[p]):int(w_off[p + 1])].view(np.uint32)), (name, i, j)
        del res, sizes
