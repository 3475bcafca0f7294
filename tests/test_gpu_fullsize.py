"""GPU tests at the benchmark's full size (4096 queries x 1 Mi rows, width 2048, 40 active):
size-independent properties plus oracle spot checks, since the oracle cannot visit 4.3e9 pairs."""
import numpy as np
import pytest

import oracle
import sdr_amd
from oracle import Batch
from sdr_amd import SDRBatch

pytestmark = pytest.mark.gpu

NQ, ND, WIDTH, ACTIVE = 4096, 1 << 20, 2048, 40


@pytest.fixture(scope="module")
def problem():
    import torch
    q_csr = sdr_amd.synth_csr(2, 0, NQ, WIDTH, ACTIVE)
    d_csr = sdr_amd.synth_csr(1, 0, ND, WIDTH, ACTIVE)
    q = SDRBatch.from_csr(*q_csr, width=WIDTH)
    d = SDRBatch.from_csr(*d_csr, width=WIDTH)
    out = torch.empty((NQ, ND), dtype=torch.int32, device="cuda:0")
    return q_csr, d_csr, q, d, out


def gather_rows(csr, rows):
    off, ids, _ = csr
    lens = (off[rows + 1] - off[rows]).astype(np.int64)
    new_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    return Batch(new_off, np.concatenate([ids[int(off[r]):int(off[r + 1])] for r in rows]))


def test_full_size_sums_and_samples(problem):
    import torch
    q_csr, d_csr, q, d, out = problem
    hq = np.bincount(q_csr[1], minlength=WIDTH).astype(np.int64)
    hd = np.bincount(d_csr[1], minlength=WIDTH).astype(np.int64)
    overlap_total = int((hq * hd).sum())       # sum over ids of (#queries with id) * (#rows with id)
    len_total = NQ * ND * ACTIVE               # sum over pairs of |q| (= of |d|)
    rng = np.random.default_rng(1)
    rows = np.sort(rng.choice(ND, 256, replace=False))
    sub_d = gather_rows(d_csr, rows)
    oq = Batch(q_csr[0], q_csr[1])
    want_sum = {0: overlap_total, 1: 2 * len_total - overlap_total,
                2: 2 * len_total - 2 * overlap_total, 3: len_total - overlap_total}
    for op, name in enumerate(("ands", "ors", "xors", "rms")):
        getattr(q, name)(d, out=out)
        assert int(out.sum(dtype=torch.int64)) == want_sum[op], name
        got = out[:, torch.from_numpy(rows).cuda()].cpu().numpy().view(np.uint32)
        assert np.array_equal(got, oracle.port().size_op(op, oq, sub_d)), name
    # mean overlap ~ A^2 / W: the generator's sanity checksum
    assert abs(overlap_total / (NQ * ND) - ACTIVE * ACTIVE / WIDTH) < 0.01


def test_full_size_merge_kernel_agrees_on_a_slab(problem):
    # the general merge kernel against the table kernel on 64 queries x 64 Ki rows
    q_csr, d_csr, q, d, out = problem
    n = 1 << 16
    qs = SDRBatch.from_csr(q_csr[0][:65], q_csr[1][: 64 * ACTIVE], width=WIDTH)
    ds = SDRBatch.from_csr(d_csr[0][: n + 1], d_csr[1][: n * ACTIVE], width=WIDTH)
    for name in ("ands", "ors", "xors", "rms"):
        a = getattr(qs, name)(ds, algo="bitslice")
        b = getattr(qs, name)(ds, algo="merge")
        assert bool((a == b).all()), name


def test_self_overlap_is_row_length(problem):
    # ands(x, x) on the diagonal is |x|; idempotence of the set algebra
    q_csr, d_csr, q, d, out = problem
    res = q.ands(q).cpu().numpy()
    assert (np.diag(res) == ACTIVE).all()
    assert (res == res.T).all()


def test_element_ops_one_query_against_full_database(problem):
    # config 3 shape: variable-length outputs over 1 Mi rows; lengths must equal the size ops,
    # totals must equal the closed forms, and sampled rows must equal the oracle
    import torch
    q_csr, d_csr, q, d, out = problem
    one = SDRBatch.from_csr(q_csr[0][:2], q_csr[1][:ACTIVE], width=WIDTH)
    oq = Batch(q_csr[0][:2], q_csr[1][:ACTIVE])
    rng = np.random.default_rng(2)
    rows = np.sort(rng.choice(ND, 200, replace=False))
    sub_d = gather_rows(d_csr, rows)
    for op, (sname, ename) in enumerate((("ands", "ande"), ("ors", "ore"), ("xors", "xore"), ("rms", "rme"))):
        res = getattr(one, ename)(d)
        sizes = getattr(one, sname)(d).to(torch.int64).flatten()
        lens = torch.diff(res.offsets.to(torch.int64))
        assert bool((lens == sizes).all()), ename
        assert res.total == int(sizes.sum())
        off, ids, _ = res.to_host()
        w_off, w_ids, _ = oracle.port().elem_op(op, oq, sub_d)
        for k, r in enumerate(rows):
            got = ids[int(off[r]):int(off[r + 1])]
            assert np.array_equal(got, w_ids[int(w_off[k]):int(w_off[k + 1])]), (ename, r)


def test_element_ops_sixteen_queries_against_full_database(problem):
    """Config 3 at 16 queries x 1 Mi rows (the shape the bench line times): result lengths equal the size ops for every
    pair, totals are their sums, and the elements of every query against 48 sampled rows equal the oracle's."""
    import torch
    q_csr, d_csr, q, d, out = problem
    nq = 16
    some = SDRBatch.from_csr(q_csr[0][: nq + 1], q_csr[1][: nq * ACTIVE], width=WIDTH)
    oq = Batch(q_csr[0][: nq + 1], q_csr[1][: nq * ACTIVE])
    rows = np.sort(np.random.default_rng(5).choice(ND, 48, replace=False))
    sub_d = gather_rows(d_csr, rows)
    for op, (sname, ename) in enumerate((("ands", "ande"), ("ors", "ore"), ("xors", "xore"), ("rms", "rme"))):
        res = getattr(some, ename)(d)
        sizes = getattr(some, sname)(d).to(torch.int64)
        lens = torch.diff(res.offsets.to(torch.int64) & 0xFFFFFFFF).view(nq, ND)
        assert bool((lens == sizes).all()), ename
        assert res.total == int(sizes.sum()), ename
        w_off, w_ids, _ = oracle.port().elem_op(op, oq, sub_d)
        pair = torch.as_tensor(np.add.outer(np.arange(nq) * ND, rows).reshape(-1), device="cuda")
        lo = (res.offsets[pair].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
        hi = (res.offsets[pair + 1].to(torch.int64) & 0xFFFFFFFF).cpu().numpy()
        assert np.array_equal(hi - lo, np.diff(w_off).astype(np.int64)), ename
        got = np.concatenate([res.ids_buffer[a:b].cpu().numpy() for a, b in zip(lo, hi)])
        assert np.array_equal(got, w_ids), ename
        del res


def test_ande_whose_worst_case_bound_exceeds_32_bits(problem):
    """256 queries x 1 Mi rows: pairs x shortest row = 1.07e10 could not be indexed with 32 bits, the
    actual intersections (0.78 ids a pair) can -- the sizes are summed exactly before the scan instead
    of refusing the call on the bound.  Total = closed form; sampled pairs equal the oracle."""
    q_csr, d_csr, q, d, out = problem
    nq = 256
    n_el = int(q_csr[0][nq])
    sub_q = SDRBatch.from_csr(q_csr[0][:nq + 1], q_csr[1][:n_el], width=WIDTH)
    res = sub_q.ande(d)
    hq = np.bincount(q_csr[1][:n_el], minlength=WIDTH).astype(np.int64)
    hd = np.bincount(d_csr[1], minlength=WIDTH).astype(np.int64)
    assert res.total == int((hq * hd).sum())
    rng = np.random.default_rng(5)
    qi = np.sort(rng.choice(nq, 8, replace=False))
    dj = np.sort(rng.choice(ND, 64, replace=False))
    w_off, w_ids, _ = oracle.port().elem_op(oracle.AND, gather_rows(q_csr, qi), gather_rows(d_csr, dj))
    off = res.offsets.cpu().numpy().view(np.uint32).astype(np.int64)
    ids = res.ids.cpu().numpy()
    for a, i in enumerate(qi):
        for b, j in enumerate(dj):
            p = int(i) * ND + int(j)
            k = a * len(dj) + b
            assert np.array_equal(ids[off[p]:off[p + 1]], w_ids[int(w_off[k]):int(w_off[k + 1])]), (i, j)


def test_few_unit_queries_against_full_database():
    """UnitData: 1 and 8 queries x 1 Mi rows -- the lane-per-row kernels for a handful of rows of A at full
    length of B (4096 blocks striding over the rows): sizes and ande on sampled rows equal the oracle,
    ande lengths equal ands everywhere."""
    import torch
    q_csr = sdr_amd.synth_csr(2, 0, 8, WIDTH, ACTIVE, True)
    d_csr = sdr_amd.synth_csr(1, 0, ND, WIDTH, ACTIVE, True)
    d = SDRBatch.from_csr(*d_csr, width=WIDTH)
    rng = np.random.default_rng(11)
    rows = np.sort(rng.choice(ND, 300, replace=False))
    off, ids, data = d_csr
    lens = (off[rows + 1] - off[rows]).astype(np.int64)
    sub_off = np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64)
    sub_ids = np.concatenate([ids[int(off[r]):int(off[r + 1])] for r in rows])
    sub_data = np.concatenate([data[int(off[r]):int(off[r + 1])] for r in rows])
    sub_d = Batch(sub_off, sub_ids, sub_data)
    for nq in (1, 8):
        n_el = int(q_csr[0][nq])
        q = SDRBatch.from_csr(q_csr[0][:nq + 1], q_csr[1][:n_el], q_csr[2][:n_el], width=WIDTH)
        oq = Batch(q_csr[0][:nq + 1], q_csr[1][:n_el], q_csr[2][:n_el])
        for op, name in enumerate(("ands", "ors", "xors", "rms")):
            got = getattr(q, name)(d).cpu().numpy().view(np.uint32)[:, rows]
            assert np.array_equal(got, oracle.port().size_op(op, oq, sub_d)), (nq, name)
        res = q.ande(d)
        sizes = q.ands(d).to(torch.int64).flatten()
        assert bool((torch.diff(res.offsets.to(torch.int64)) == sizes).all())
        r_off, r_ids, r_data = res.to_host()
        w_off, w_ids, w_data = oracle.port().elem_op(oracle.AND, oq, sub_d)
        for i in range(nq):
            for k, r in enumerate(rows):
                p, w = i * ND + int(r), i * len(rows) + k
                assert np.array_equal(r_ids[int(r_off[p]):int(r_off[p + 1])], w_ids[int(w_off[w]):int(w_off[w + 1])])
                assert np.array_equal(r_data[int(r_off[p]):int(r_off[p + 1])].view(np.uint32),
                                      w_data[int(w_off[w]):int(w_off[w + 1])].view(np.uint32))


def test_many_unit_queries_against_full_database(problem):
    """UnitData on both sides, 1024 queries x 1 Mi rows: the lane-per-row size kernel with its tiles shared over
    blocks, every tile and block boundary of the full length.  ands / xors / rms equal the oracle on 300 sampled
    rows of the database for all queries; everywhere, a payload can only remove a match from ands and can only
    keep one in xors and rms, so each result lies on its side of the id-only result of the same pair."""
    import torch
    _, _, _, _, out = problem
    nq = 1024
    q_csr = sdr_amd.synth_csr(2, 0, nq, WIDTH, ACTIVE, True)
    d_csr = sdr_amd.synth_csr(1, 0, ND, WIDTH, ACTIVE, True)
    q = SDRBatch.from_csr(*q_csr, width=WIDTH)
    d = SDRBatch.from_csr(*d_csr, width=WIDTH)
    q_ids = SDRBatch.from_csr(q_csr[0], q_csr[1], width=WIDTH)
    d_ids = SDRBatch.from_csr(d_csr[0], d_csr[1], width=WIDTH)
    rng = np.random.default_rng(23)
    rows = np.sort(rng.choice(ND, 300, replace=False))
    off, ids, data = d_csr
    lens = (off[rows + 1] - off[rows]).astype(np.int64)
    sub_d = Batch(np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64),
                  np.concatenate([ids[int(off[r]):int(off[r + 1])] for r in rows]),
                  np.concatenate([data[int(off[r]):int(off[r + 1])] for r in rows]))
    oq = Batch(*q_csr)
    cols = torch.from_numpy(rows).cuda()
    got, plain = out[:nq], torch.empty((nq, ND), dtype=torch.int32, device="cuda:0")
    for op, name in ((0, "ands"), (2, "xors"), (3, "rms")):
        getattr(q, name)(d, out=got)
        assert np.array_equal(got[:, cols].cpu().numpy().view(np.uint32), oracle.port().size_op(op, oq, sub_d)), name
        getattr(q_ids, name)(d_ids, out=plain)
        assert bool((got <= plain).all() if op == 0 else (got >= plain).all()), name
