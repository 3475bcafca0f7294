"""GPU parity tests: the HIP path (through the C ABI) against the CPU oracle and the committed
golden vectors, bit for bit.  Sizes are kept where the oracle finishes in seconds."""
import numpy as np
import pytest

import oracle
import sdr_amd
import util
from oracle import AND, OR, RM, XOR, Batch
from sdr_amd import SDRBatch

pytestmark = pytest.mark.gpu

PAIRINGS = {"EE": (False, False), "UU": (True, True), "UE": (True, False), "EU": (False, True)}
OPS = {AND: ("ands", "ande"), OR: ("ors", "ore"), XOR: ("xors", "xore"), RM: ("rms", "rme")}


def dev(b: Batch, width=0) -> SDRBatch:
    return SDRBatch.from_csr(b.off, b.ids, b.data, width=width)


def gpu_sizes(op, t: SDRBatch, a: SDRBatch, **kw) -> np.ndarray:
    return getattr(t, OPS[op][0])(a, **kw).cpu().numpy().view(np.uint32)


def gpu_elems(op, t: SDRBatch, a: SDRBatch, **kw):
    return getattr(t, OPS[op][1])(a, **kw).to_host()


def id_only(op, this_unit):
    return op == OR or not this_unit


def assert_elems_equal(got, want):
    assert np.array_equal(got[0], want[0]), "offsets differ"
    assert np.array_equal(got[1], want[1]), "ids differ"
    if want[2] is None:
        assert got[2] is None
    else:
        assert np.array_equal(got[2].view(np.uint32), want[2].view(np.uint32)), "payload bits differ"


def fuzz_pair(n, name, seed):
    tu, au = PAIRINGS[name]
    return (Batch(*util.fuzz_universe(n, seed if tu else None)),
            Batch(*util.fuzz_universe(n, seed + 1 if au else None)))


@pytest.mark.parametrize("name", list(PAIRINGS))
def test_fuzz_grid_sizes_vs_golden(golden, name):
    g = golden("fuzz_sizes.npz")
    t, a = fuzz_pair(256, name, int(g["seed"]))
    dt, da = dev(t, 8), dev(a, 8)
    for op in range(4):
        want = g[name][op].astype(np.uint32)
        assert np.array_equal(gpu_sizes(op, dt, da, algo="merge"), want)
        assert np.array_equal(gpu_sizes(op, dt, da), want)
        if id_only(op, PAIRINGS[name][0]):
            assert np.array_equal(gpu_sizes(op, dt, da, algo="bitslice"), want)
        else:
            with pytest.raises(sdr_amd.SdrbError):
                gpu_sizes(op, dt, da, algo="bitslice")


@pytest.mark.parametrize("name", list(PAIRINGS))
def test_fuzz_grid_elems_vs_golden(golden, name):
    g = golden("fuzz_elems.npz")
    seed = int(g["seed"])
    t, a = fuzz_pair(128, name, seed)
    dt, da = dev(t, 7), dev(a, 7)
    for op in range(4):
        off, ids, data = gpu_elems(op, dt, da)
        assert np.array_equal(np.diff(off).reshape(128, 128), g[name + "_len"][op])
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(128, 128), g[name + "_digest"][op])
    t, a = fuzz_pair(64, name, seed)
    dt, da = dev(t), dev(a)
    for op in range(4):
        off, ids, data = gpu_elems(op, dt, da)
        assert np.array_equal(off, g[f"{name}_full{op}_off"])
        assert np.array_equal(ids, g[f"{name}_full{op}_ids"])
        if data is not None:
            assert np.array_equal(data.view(np.uint32), g[f"{name}_full{op}_data"].view(np.uint32))


@pytest.mark.parametrize("name", list(PAIRINGS))
def test_swap_flag_matches_swapped_roles(name):
    # swap: result(i, j) = B_j.op(A_i)
    t, a = fuzz_pair(64, name, 99)
    dt, da = dev(t, 6), dev(a, 6)
    for op in range(4):
        want = oracle.port().size_op(op, a, t).T
        assert np.array_equal(gpu_sizes(op, dt, da, swap=True), want)
        assert np.array_equal(gpu_sizes(op, dt, da, swap=True, algo="merge"), want)
        off, ids, data = gpu_elems(op, dt, da, swap=True)
        w_off, w_ids, w_data = oracle.port().elem_op(op, a, t)
        # oracle order is [j][i]; compare pair by pair
        lens = np.diff(w_off).reshape(a.n, t.n).T
        assert np.array_equal(np.diff(off).reshape(t.n, a.n), lens)
        got_d = util.pair_digests(off, ids, data).reshape(t.n, a.n)
        want_d = util.pair_digests(w_off, w_ids, w_data).reshape(a.n, t.n).T
        assert np.array_equal(got_d, want_d)


@pytest.mark.parametrize("width,max_len_b,nq,nd", [
    (2048, 40, 70, 700), (2048, 10, 64, 256), (300, 30, 1, 513), (2048, 64, 129, 300),
    (4096, 40, 65, 257), (3001, 16, 5, 100), (2047, 33, 130, 1000)])
def test_bitslice_ragged_vs_oracle(width, max_len_b, nq, nd):
    rng = np.random.default_rng(width * 31 + max_len_b)
    t = Batch(*util.ragged_rows(rng, nq, width, 70))
    a = Batch(*util.ragged_rows(rng, nd, width, max_len_b))
    dt, da = dev(t, width), dev(a, width)
    for op in range(4):
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(gpu_sizes(op, dt, da, algo="bitslice"), want)
        assert np.array_equal(gpu_sizes(op, dt, da, algo="merge"), want)
    want = oracle.port().size_op(RM, a, t).T
    assert np.array_equal(gpu_sizes(RM, dt, da, swap=True, algo="bitslice"), want)


def test_query_ids_beyond_db_width_never_match():
    rng = np.random.default_rng(5)
    t = Batch(*util.ragged_rows(rng, 20, 400, 50, lo=-100))  # ids in [-100, 300)
    a = Batch(*util.ragged_rows(rng, 300, 200, 40))          # ids in [0, 200)
    dt, da = dev(t), dev(a, 200)
    for op in range(4):
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(gpu_sizes(op, dt, da, algo="bitslice"), want)
        assert np.array_equal(gpu_sizes(op, dt, da), want)


@pytest.mark.parametrize("tu,au", list(PAIRINGS.values()))
def test_merge_and_elems_ragged_signed_ids(tu, au):
    rng = np.random.default_rng(11 + 2 * tu + au)
    t = Batch(*util.ragged_rows(rng, 37, 500, 90, tu, lo=-250))
    a = Batch(*util.ragged_rows(rng, 333, 500, 120, au, lo=-250))
    dt, da = dev(t), dev(a)
    for op in range(4):
        assert np.array_equal(gpu_sizes(op, dt, da), oracle.port().size_op(op, t, a))
        assert_elems_equal(gpu_elems(op, dt, da), oracle.port().elem_op(op, t, a))


def test_rows_longer_than_table_kernel_fall_back_to_merge():
    rng = np.random.default_rng(3)
    t = Batch(*util.ragged_rows(rng, 9, 1000, 200))
    a = Batch(*util.ragged_rows(rng, 130, 1000, 200))
    dt, da = dev(t, 1000), dev(a, 1000)
    assert a.n and da.max_row_len > 64
    assert np.array_equal(gpu_sizes(AND, dt, da), oracle.port().size_op(AND, t, a))
    with pytest.raises(sdr_amd.SdrbError) as e:
        gpu_sizes(AND, dt, da, algo="bitslice")
    assert e.value.code == -5


def test_empty_and_degenerate_batches():
    empty = Batch([0], [])
    some = Batch.from_rows([[], [1, 2, 3], []])
    de, ds = dev(empty, 8), dev(some, 8)
    assert gpu_sizes(AND, de, ds).shape == (0, 3)
    assert gpu_sizes(AND, ds, de).shape == (3, 0)
    for op in range(4):
        for algo in (None, "merge", "bitslice"):
            assert np.array_equal(gpu_sizes(op, ds, ds, algo=algo), oracle.port().size_op(op, some, some))
        assert_elems_equal(gpu_elems(op, ds, ds), oracle.port().elem_op(op, some, some))
        off, ids, data = gpu_elems(op, de, ds)
        assert off.tolist() == [0] and ids.size == 0


def test_leading_dimension_and_partial_tiles():
    import torch
    rng = np.random.default_rng(21)
    t = Batch(*util.ragged_rows(rng, 67, 2048, 40))
    a = Batch(*util.ragged_rows(rng, 1000, 2048, 40))
    dt, da = dev(t, 2048), dev(a, 2048)
    out = torch.full((70, 1100), -7, dtype=torch.int32, device="cuda:0")
    dt.ands(da, out=out)
    got = out.cpu().numpy()
    assert np.array_equal(got[:67, :1000].view(np.uint32), oracle.port().size_op(AND, t, a))
    assert (got[67:, :] == -7).all() and (got[:, 1000:] == -7).all()


@pytest.mark.parametrize("fname", ["mini_2048_40.npz", "mini_16k_328.npz"])
def test_config_shaped_minis_vs_golden(golden, fname):
    g = golden(fname)
    nq, nd, width, active = int(g["nq"]), int(g["nd"]), int(g["width"]), int(g["active"])
    wd = bool(g["with_data"])
    q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, width, active, wd), width=width)
    d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, width, active, wd), width=width)
    for op in range(4):
        assert np.array_equal(gpu_sizes(op, q, d), g["sizes"][op])
        off, ids, data = gpu_elems(op, q, d)
        assert np.array_equal(np.diff(off).reshape(nq, nd), g[f"len{op}"])
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(nq, nd), g[f"digest{op}"])
        assert np.array_equal(ids[: int(off[1])], g[f"pair00_ids{op}"])
        if wd:
            assert np.array_equal(data[: int(off[1])].view(np.uint32), g[f"pair00_data{op}"].view(np.uint32))


def test_readme_examples_on_device():
    a = SDRBatch.from_rows([[1, 2]])
    b = SDRBatch.from_rows([[2, 3]])
    assert a.ande(b).to_host()[1].tolist() == [2]          # README.md:65-70
    assert a.ore(b).to_host()[1].tolist() == [1, 2, 3]
    assert a.xore(b).to_host()[1].tolist() == [1, 3]
    assert a.rme(b).to_host()[1].tolist() == [1]
    assert int(a.ands(b)[0, 0]) == 1                         # README.md:79-90
    u = SDRBatch.from_rows([([0], [0.5])], with_data=True)
    off, ids, data = u.ande(u).to_host()                     # README.md:113-116
    assert ids.tolist() == [0] and data.tolist() == [0.25]
    q = SDRBatch.from_rows([([0], [0.25])], with_data=True)
    assert q.ande(q).total == 0 and int(q.ands(q)[0, 0]) == 0  # README.md:120-123
    off, ids, data = u.ande(SDRBatch.from_rows([[0]])).to_host()  # README.md:127-128
    assert ids.tolist() == [0] and data.tolist() == [0.5]


def test_operator_sugar_matches_the_named_ops():
    # SDR.hpp:431-460: & * -> ande, | + -> ore, ^ -> xore, - -> rme; the augmented forms are the in-place ops
    rng = np.random.default_rng(5)
    t = Batch(*util.ragged_rows(rng, 30, 100, 30, True))
    a = Batch(*util.ragged_rows(rng, 30, 100, 30, True))
    dt, da = dev(t, 100), dev(a, 100)
    for sugar, op in ((dt & da, AND), (dt * da, AND), (dt | da, OR), (dt + da, OR), (dt ^ da, XOR), (dt - da, RM)):
        assert_elems_equal(sugar.to_host(), oracle.port().elem_op(op, t, a))
    c = dev(t, 100)
    c &= da
    c |= dt
    c ^= da
    c -= dt
    want = dev(t, 100).andi(da).ori(dt).xori(da).rmi(dt)
    assert c.n_rows == want.n_rows and c.nnz == want.nnz
    assert np.array_equal(c.to_wire().cpu().numpy(), want.to_wire().cpu().numpy())   # offsets, ids, payload bits


@pytest.mark.parametrize("width,max_len", [(2048, 40), (4096, 100), (100, 100), (33, 20)])
@pytest.mark.parametrize("au", [False, True])
def test_bitset_element_kernel_vs_oracle(width, max_len, au):
    # `this` EmptyData => results carry ids only => dense-bitset fill; the general merge fill must agree
    rng = np.random.default_rng(width + max_len + au)
    t = Batch(*util.ragged_rows(rng, 19, width, max_len))
    a = Batch(*util.ragged_rows(rng, 777, width, max_len, au))
    dt, da = dev(t, width), dev(a, width)
    for op in range(4):
        want = oracle.port().elem_op(op, t, a)
        assert_elems_equal(gpu_elems(op, dt, da), want)
        assert_elems_equal(gpu_elems(op, dt, da, algo="merge"), want)
        assert_elems_equal(gpu_elems(op, dt, da, algo="bitmap"), want)
        if op != AND:   # unions of short rows: a lane per pair (rows of 100 ids do not fit the LDS stage)
            for algo in ("lanemerge", "lanerank"):
                try:
                    got = gpu_elems(op, dt, da, algo=algo)
                except sdr_amd.SdrbError as e:
                    assert e.code == -5 and max_len > 64
                else:
                    assert max_len <= 64
                    assert_elems_equal(got, want)
    # swapped roles: this = rows of the second batch
    b2 = Batch(*util.ragged_rows(rng, 300, width, max_len))
    db2 = dev(b2, width)
    for op in range(4):
        w_off, w_ids, w_data = oracle.port().elem_op(op, b2, t)
        for algo in (None, "lanerank", "lanemerge"):
            if algo is not None and (op == AND or max_len > 64):
                continue
            off, ids, data = gpu_elems(op, dt, db2, swap=True, algo=algo)
            assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, b2.n),
                                  util.pair_digests(w_off, w_ids, w_data).reshape(b2.n, t.n).T), (op, algo)


def test_bitset_element_kernel_not_used_when_a_exceeds_width():
    rng = np.random.default_rng(77)
    t = Batch(*util.ragged_rows(rng, 5, 600, 30))          # ids up to 599
    a = Batch(*util.ragged_rows(rng, 100, 256, 30))        # declared width 256
    dt, da = dev(t), dev(a, 256)
    for op in range(4):
        assert_elems_equal(gpu_elems(op, dt, da), oracle.port().elem_op(op, t, a))


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("width,len_a,len_b", [(16384, 400, 400), (5000, 70, 130), (32768, 30, 500), (257, 200, 257)])
def test_table_kernels_with_declared_width_vs_oracle(name, width, len_a, len_b):
    # payload-dependent ops (and long rows / wide id spaces) go through the wave-per-row table
    # kernels; the general merge kernels and the oracle must agree bit for bit
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(width + len_a + 2 * tu + au)
    t = Batch(*util.ragged_rows(rng, 21, width, len_a, tu))
    a = Batch(*util.ragged_rows(rng, 150, width, len_b, au))
    dt, da = dev(t, width), dev(a, width)
    for op in range(4):
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(gpu_sizes(op, dt, da), want)
        assert np.array_equal(gpu_sizes(op, dt, da, algo="merge"), want)
        assert np.array_equal(gpu_sizes(op, dt, da, algo="rowtable"), want)
        assert np.array_equal(gpu_sizes(op, dt, da, swap=True), oracle.port().size_op(op, a, t).T)
        assert np.array_equal(gpu_sizes(op, dt, da, swap=True, algo="rowtable"), oracle.port().size_op(op, a, t).T)
        w = oracle.port().elem_op(op, t, a)
        assert_elems_equal(gpu_elems(op, dt, da), w)
        assert_elems_equal(gpu_elems(op, dt, da, algo="merge"), w)
        assert_elems_equal(gpu_elems(op, dt, da, algo="rowtable"), w)
        off, ids, data = gpu_elems(op, dt, da, swap=True)
        w_off, w_ids, w_data = oracle.port().elem_op(op, a, t)
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n),
                              util.pair_digests(w_off, w_ids, w_data).reshape(a.n, t.n).T)


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("width,len_a,len_b,nq,nd", [(2048, 40, 40, 37, 301), (16384, 400, 400, 19, 130), (63, 63, 63, 20, 90),
                                                   (32768, 90, 700, 5, 77), (20000, 300, 100, 33, 50), (640, 640, 5, 17, 64), (4100, 0, 50, 3, 40),
                                                   (28000, 40, 40, 33, 70)])   # short rows, tables past 64 KiB of LDS
def test_rank_table_element_kernel_vs_oracle(name, width, len_a, len_b, nq, nd):
    # wave-per-pair fill: positions come from rank tables; every op, payload pairing and orientation
    # must equal the oracle's merge bit for bit, rows longer than a wave and empty rows included
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(width + len_a + 2 * tu + au)
    t = Batch(*util.ragged_rows(rng, nq, width, len_a, tu))
    a = Batch(*util.ragged_rows(rng, nd, width, len_b, au))
    dt, da = dev(t, width), dev(a, width)
    for op in range(4):
        assert_elems_equal(gpu_elems(op, dt, da, algo="rank"), oracle.port().elem_op(op, t, a))
        off, ids, data = gpu_elems(op, dt, da, swap=True, algo="rank")
        w_off, w_ids, w_data = oracle.port().elem_op(op, a, t)
        assert np.array_equal(np.diff(off).reshape(t.n, a.n), np.diff(w_off).reshape(a.n, t.n).T)
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n),
                              util.pair_digests(w_off, w_ids, w_data).reshape(a.n, t.n).T)


def test_rank_table_kernel_refuses_what_it_cannot_hold():
    rng = np.random.default_rng(3)
    t = Batch(*util.ragged_rows(rng, 4, 300, 20, False))
    a = Batch(*util.ragged_rows(rng, 9, 200, 20, False))
    with pytest.raises(sdr_amd.SdrbError):          # ids of `this` beyond the declared width of the other side
        gpu_elems(OR, dev(t, 300), dev(a, 200), algo="rank")
    neg = Batch(a.off, (a.ids - 7) * 1000003, a.data)
    with pytest.raises(sdr_amd.SdrbError):          # no declared width and none to derive: a negative id, ids far apart
        gpu_elems(OR, dev(neg), dev(neg), algo="rank")
    # no declared width, no negative id: served as if the largest id + 1 had been declared
    assert_elems_equal(gpu_elems(OR, dev(a), dev(a), algo="rank"), oracle.port().elem_op(OR, a, a))


@pytest.mark.parametrize("width,max_len,nq,nd", [(2048, 40, 70, 700), (2048, 2048, 9, 300), (1000, 600, 33, 513),
                                                 (300, 300, 1, 257), (64, 64, 130, 100)])
@pytest.mark.parametrize("au", [False, True])
def test_popcount_kernel_vs_oracle(width, max_len, nq, nd, au):
    # dense-bitset popcount size ops: any row density, id-only results
    rng = np.random.default_rng(width + max_len + nq)
    t = Batch(*util.ragged_rows(rng, nq, width + 50, min(max_len, width), lo=-20))   # some ids outside B's width
    a = Batch(*util.ragged_rows(rng, nd, width, max_len, au))
    dt, da = dev(t), dev(a, width)
    for op in range(4):
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(gpu_sizes(op, dt, da, algo="popcount"), want)
        assert np.array_equal(gpu_sizes(op, dt, da), want)           # auto: bit-sliced or popcount
    want = oracle.port().size_op(RM, a, t).T
    if not au:
        assert np.array_equal(gpu_sizes(RM, dt, da, swap=True, algo="popcount"), want)
    else:   # `this` would be UnitData: payload-dependent, popcount must refuse
        with pytest.raises(sdr_amd.SdrbError):
            gpu_sizes(RM, dt, da, swap=True, algo="popcount")


@pytest.mark.gpu
@pytest.mark.parametrize("with_data", [False, True])
def test_adopted_view_equals_uploaded_batch(with_data):
    """A zero-copy view of caller tensors (wire form of the query exchange) must behave exactly
    like a validated upload of the same rows, for size ops and element ops, and re-adopting a
    smaller batch into the same handle must not leave stale state behind."""
    import torch
    from sdr_amd.shard import HipEngine, PackedQueries
    rng = np.random.default_rng(77)
    eng = HipEngine(0)
    d_off, d_ids, d_data = util.ragged_rows(rng, 500, 2048, 40, with_data)
    db = SDRBatch.from_csr(d_off, d_ids, d_data, width=2048)
    view = None
    for nq in (70, 9):
        q_off, q_ids, q_data = util.ragged_rows(rng, nq, 2048, 40, with_data)
        up = SDRBatch.from_csr(q_off, q_ids, q_data, width=2048)
        wire = PackedQueries.from_host_csr(q_off, q_ids, q_data, width=2048, engine=eng)
        assert wire.plan.words == wire.tensor.numel()
        view = eng.adopt(wire.tensor, wire.plan, 2048, into=view)
        assert view.stats() == up.stats()
        for name in ("ands", "ors", "xors", "rms"):
            assert torch.equal(getattr(view, name)(db), getattr(up, name)(db)), name
        a, b = view.xore(db).to_host(), up.xore(db).to_host()
        for x, y in zip(a, b):
            assert (x is None and y is None) or np.array_equal(x, y)


@pytest.mark.gpu
@pytest.mark.parametrize("with_data", [False, True])
def test_batch_from_database_file_equals_batch_from_arrays(tmp_path, with_data):
    """A row range uploaded straight from a CSR database file behaves like the same rows uploaded
    from arrays; payloads of a UnitData file can be dropped at load time."""
    import torch
    from sdr_amd import csrfile
    rng = np.random.default_rng(31)
    off, ids, data = util.ragged_rows(rng, 800, 2048, 40, with_data)
    path = tmp_path / "db.sdrb"
    csrfile.write(path, off, ids, data, width=2048)
    q = SDRBatch.from_csr(*util.ragged_rows(rng, 33, 2048, 40, with_data), width=2048)
    lo, hi = 123, 700
    s_off = off[lo:hi + 1] - off[lo]
    s_ids = ids[int(off[lo]):int(off[hi])]
    s_data = None if data is None else data[int(off[lo]):int(off[hi])]
    a = csrfile.upload_rows(path, lo, hi)
    b = SDRBatch.from_csr(s_off, s_ids, s_data, width=2048)
    assert a.stats() == b.stats()
    for name in ("ands", "ors", "xors", "rms"):
        assert torch.equal(getattr(q, name)(a), getattr(q, name)(b)), name
    if with_data:
        ids_only = csrfile.upload_rows(path, lo, hi, payloads=False)
        want = SDRBatch.from_csr(s_off, s_ids, None, width=2048)
        assert torch.equal(q.ands(ids_only), q.ands(want))


def _repeat_rows(b: Batch, rows):
    """The listed rows of a host batch, in that order, as a new host batch."""
    off, ids, data = [0], [], []
    for r in rows:
        r_ids, r_data = b.row(r)
        ids.append(r_ids)
        if r_data is not None:
            data.append(r_data)
        off.append(off[-1] + len(r_ids))
    return Batch(np.asarray(off, np.uint64), np.concatenate(ids) if ids else np.zeros(0, np.int32),
                 np.concatenate(data) if data else None)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PAIRINGS))
def test_rowwise_and_inplace_ops_vs_inplace_golden(golden, name):
    """Row-wise evaluation (result j = A_j.op(B_j)) and the in-place batch ops built on it against the
    fixture the reference's andi/ori/xori/rmi generated (lengths and content digests)."""
    g = golden("inplace.npz")
    n = 32
    t, a = fuzz_pair(n, name, int(g["seed"]))
    rows_i = list(range(0, n, 3))
    pairs = [(i, j) for i in rows_i for j in range(n)]
    tt = _repeat_rows(t, [i for i, _ in pairs])
    aa = _repeat_rows(a, [j for _, j in pairs])
    d_a = dev(aa)
    for op, (sname, ename) in OPS.items():
        want_len = np.array([g[f"{name}_len{op}"][i, j] for i, j in pairs], dtype=np.int64)
        want_dig = np.array([g[f"{name}_digest{op}"][i, j] for i, j in pairs], dtype=np.uint32)
        d_t = dev(tt)
        assert np.array_equal(d_t.rowwise_size(sname, d_a).cpu().numpy().astype(np.int64), want_len), sname
        off, ids, data = d_t.rowwise_elem(ename, d_a).to_host()
        assert np.array_equal(np.diff(off).astype(np.int64), want_len), ename
        assert np.array_equal(util.pair_digests(off, ids, data), want_dig), ename
        # in place: the batch itself now holds the results
        getattr(d_t, ename[:-1] + "i")(d_a)
        assert d_t.n_rows == len(pairs) and d_t.nnz == int(want_len.sum())
        w = d_t.to_wire().cpu().numpy()
        got_off = w[: len(pairs) + 1].view(np.uint32).astype(np.uint64)
        got_ids = w[len(pairs) + 1: len(pairs) + 1 + d_t.nnz]
        got_data = w[len(pairs) + 1 + d_t.nnz:].view(np.float32) if tt.data is not None else None
        assert np.array_equal(util.pair_digests(got_off, got_ids, got_data), want_dig), ename + " in place"


@pytest.mark.gpu
@pytest.mark.parametrize("kind", ["E", "U"])
def test_inplace_aliasing_a_op_a(golden, kind):
    """a.andi(a), a.ori(a), a.xori(a), a.rmi(a) are legal in the reference (src/test.cpp:241-272)."""
    g = golden("inplace.npz")
    n = 32
    u = Batch(*util.fuzz_universe(n, int(g["seed"]) if kind == "U" else None))
    for op, (_, ename) in OPS.items():
        d_u = dev(u)
        getattr(d_u, ename[:-1] + "i")(d_u)
        w = d_u.to_wire().cpu().numpy()
        got_off = w[: n + 1].view(np.uint32).astype(np.uint64)
        got_ids = w[n + 1: n + 1 + d_u.nnz]
        got_data = w[n + 1 + d_u.nnz:].view(np.float32) if u.data is not None else None
        assert np.array_equal(np.diff(got_off), g[f"alias_{kind}_len{op}"].astype(np.uint64)), ename
        assert np.array_equal(util.pair_digests(got_off, got_ids, got_data), g[f"alias_{kind}_digest{op}"]), ename


@pytest.mark.gpu
def test_rowwise_needs_equal_lengths():
    rng = np.random.default_rng(4)
    a = SDRBatch.from_csr(*util.ragged_rows(rng, 5, 100, 10), width=100)
    b = SDRBatch.from_csr(*util.ragged_rows(rng, 6, 100, 10), width=100)
    with pytest.raises(sdr_amd.SdrbError):
        a.rowwise_size("ands", b)
    with pytest.raises(ValueError):
        a.andi(b)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("width,max_len", [(16384, 328), (2048, 40), (9000, 200), (40000, 90), (700, 200), (500, 200)])
def test_mask_table_kernel_many_tiles_and_groups(name, width, max_len):
    """The mask-table kernel over several tiles of A (partial last tile), several 64-row groups of B
    (partial last group) and rows of several 64-element chunks; the shapes pick its variants: 32-row
    tiles with the payloads of A in LDS, 64-row tiles with them, and (width 40000: too wide for the
    table) the fallback.  The fill's pair sort: four pairs a thread in place (sparse results: 2048/40,
    9000/200), 256 pairs staged in LDS (16384/328, 700/200: about 14 elements a pair), and blocks whose 256
    pairs hold more than the stage (500/200: about 20 elements a pair), which sort in place."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(99 + 2 * tu + au + width)
    t = Batch(*util.ragged_rows(rng, 150, width, max_len, tu))
    a = Batch(*util.ragged_rows(rng, 333, width, max_len, au))
    dt, da = dev(t, width), dev(a, width)
    for op in range(4):
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(gpu_sizes(op, dt, da), want), op
        assert np.array_equal(gpu_sizes(op, dt, da, swap=True), oracle.port().size_op(op, a, t).T), op
        if width <= 20000:      # the table fits LDS: the kernel by name, whatever the default choice is
            assert np.array_equal(gpu_sizes(op, dt, da, algo="masktable"), want), op
            assert np.array_equal(gpu_sizes(op, dt, da, swap=True, algo="masktable"), oracle.port().size_op(op, a, t).T), op
    # ande through the mask-table fill (slots claimed concurrently, then sorted per pair)
    assert_elems_equal(gpu_elems(AND, dt, da), oracle.port().elem_op(AND, t, a))
    off, ids, data = gpu_elems(AND, dt, da, swap=True)
    w_off, w_ids, w_data = oracle.port().elem_op(AND, a, t)
    assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n),
                          util.pair_digests(w_off, w_ids, w_data).reshape(a.n, t.n).T)


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["UU", "UE", "EU"])
@pytest.mark.parametrize("width,len_t,len_a,nt,na", [(2048, 40, 40, 150, 700), (4096, 128, 64, 33, 300), (300, 100, 16, 70, 513),
                                                     (64, 64, 64, 40, 257), (2048, 0, 40, 20, 300), (2000, 40, 0, 64, 300),
                                                     (4000, 10, 33, 17, 100)])
def test_lane_rows_size_kernel_vs_oracle(name, width, len_t, len_a, nt, na):
    """Payload-dependent size ops by the lane-per-row match kernel: partial tiles of A and blocks of B, rows of A
    longer than a wave, widths where most ids of a tile are taken, empty sides; both orientations."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(7 + 2 * tu + au + width + len_t)
    t = Batch(*util.ragged_rows(rng, nt, width, len_t, tu)) if len_t else Batch(np.zeros(nt + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32) if tu else None)
    a = Batch(*util.ragged_rows(rng, na, width, len_a, au)) if len_a else Batch(np.zeros(na + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32) if au else None)
    dt, da = dev(t, width), dev(a, width)
    served = 0
    for op in range(4):
        for swap in (False, True):
            this_unit = au if swap else tu
            if op == OR or not this_unit:       # ids alone decide: not this kernel's case, it must refuse
                with pytest.raises(sdr_amd.SdrbError):
                    gpu_sizes(op, dt, da, swap=swap, algo="lanerows")
                continue
            want = oracle.port().size_op(op, a, t).T if swap else oracle.port().size_op(op, t, a)
            assert np.array_equal(gpu_sizes(op, dt, da, swap=swap, algo="lanerows"), want), (op, swap)
            assert np.array_equal(gpu_sizes(op, dt, da, swap=swap), want), (op, swap)
            served += 1
    assert served >= 3


@pytest.mark.gpu
@pytest.mark.parametrize("width", [40, 64])
def test_lane_rows_size_kernel_with_every_id_matching(width):
    """Rows that hold every id of the width: every pair matches in `width` ids, the largest count the kernel's
    bit-sliced counters take (64 matches, and 64 + 64 for xors when every match is dropped), and every id of a
    tile is held by all 32 rows (the bit loop's longest run)."""
    rng = np.random.default_rng(width)
    nt, na = 70, 300

    def full(n, low, high):
        off = np.arange(n + 1, dtype=np.uint64) * width
        ids = np.tile(np.arange(width, dtype=np.int32), n)
        return Batch(off, ids, rng.uniform(low, high, n * width).astype(np.float32))

    for low, high in ((0.0, 1.0), (0.5, 0.5), (0.0, 0.05)):     # mixed, all kept by ands / all dropped by xors, none kept
        t, a = full(nt, low, high), full(na, low, high)
        dt, da = dev(t, width), dev(a, width)
        for op in (AND, XOR, RM):
            for swap in (False, True):
                want = oracle.port().size_op(op, a, t).T if swap else oracle.port().size_op(op, t, a)
                assert np.array_equal(gpu_sizes(op, dt, da, swap=swap, algo="lanerows"), want), (low, op, swap)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("nq,width,len_a,len_b", [(1, 2048, 40, 40), (8, 2048, 40, 40), (3, 4099, 300, 64), (5, 33, 33, 20),
                                                  (8, 17000, 90, 7), (2, 1, 1, 1), (9, 2048, 40, 40), (16, 2049, 64, 64),
                                                  (13, 9001, 500, 33), (16, 31, 31, 31)])
def test_few_rows_of_a_size_kernels_vs_oracle(name, nq, width, len_a, len_b):
    """One query (or up to sixteen) against a database: the one-launch kernels with a byte (or two) per id
    saying which rows of A hold it; every op, payload pairing and orientation, ragged rows, a partial last
    block; ande through the same kernels (elements in id order without a sort)."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(7 * nq + width + 2 * tu + au)
    t = Batch(*util.ragged_rows(rng, nq, width, len_a, tu))
    a = Batch(*util.ragged_rows(rng, 777, width, len_b, au))
    dt, da = dev(t, width), dev(a, width)
    for op in range(4):
        assert np.array_equal(gpu_sizes(op, dt, da), oracle.port().size_op(op, t, a)), op
        assert np.array_equal(gpu_sizes(op, dt, da, swap=True), oracle.port().size_op(op, a, t).T), op
        # the mask-table family's own few-rows kernel, whatever the default choice is
        assert np.array_equal(gpu_sizes(op, dt, da, algo="masktable"), oracle.port().size_op(op, t, a)), op
        assert np.array_equal(gpu_sizes(op, dt, da, swap=True, algo="masktable"), oracle.port().size_op(op, a, t).T), op
    assert_elems_equal(gpu_elems(AND, dt, da), oracle.port().elem_op(AND, t, a))
    off, ids, data = gpu_elems(AND, dt, da, swap=True)
    w_off, w_ids, w_data = oracle.port().elem_op(AND, a, t)
    assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n),
                          util.pair_digests(w_off, w_ids, w_data).reshape(a.n, t.n).T)


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PAIRINGS))
def test_batches_with_64_bit_ids_vs_reference(golden, name):
    """Rows with `long` ids (SDRElem<long, ...>, src/fuzz.cpp:413-423), far outside the 32-bit range:
    sizes and elements (int64 ids, payload bits) against what the reference itself returned."""
    g = golden("long_ids.npz")

    def side(s):
        key = f"{name}_{s}_data"
        return g[f"{name}_{s}_off"], g[f"{name}_{s}_ids"], (g[key] if key in g.files else None)
    t, a = SDRBatch.from_csr(*side("t")), SDRBatch.from_csr(*side("a"))
    assert t.id_bits == 64 and a.id_bits == 64
    for op in range(4):
        assert np.array_equal(gpu_sizes(op, t, a), g[f"{name}_size{op}"]), op
        off, ids, data = gpu_elems(op, t, a)
        assert ids.dtype == np.int64
        assert np.array_equal(off, g[f"{name}_off{op}"]) and np.array_equal(ids, g[f"{name}_ids{op}"]), op
        if f"{name}_data{op}" in g.files:
            assert np.array_equal(data.view(np.uint32), g[f"{name}_data{op}"].view(np.uint32)), op
        else:
            assert data is None
        # the swap flag evaluates a.op(t) into the same layout
        assert np.array_equal(gpu_sizes(op, a, t, swap=True), g[f"{name}_size{op}"].T), op


@pytest.mark.gpu
def test_64_bit_batches_are_refused_where_not_supported():
    rng = np.random.default_rng(12)
    off, ids, _ = util.ragged_rows(rng, 20, 100, 10)
    wide = SDRBatch.from_csr(off, ids.astype(np.int64) * (1 << 35))
    narrow = SDRBatch.from_csr(off, ids)
    for call in (lambda: wide.allpairs_topk(wide, 4), lambda: wide.range_size([0], [5]), lambda: wide.to_wire()):
        with pytest.raises(sdr_amd.SdrbError):
            call()
    with pytest.raises(sdr_amd.SdrbError) as e:
        SDRBatch.from_csr(np.array([0, 2], np.uint64), np.array([5, 5], np.int64))
    assert e.value.code == -2


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PAIRINGS))
def test_batches_whose_ids_differ_in_width(name):
    """`long` ids against `int` ids, what the reference's fuzzer runs (src/fuzz.cpp:413-423: SDRElem<long, UnitData> rows
    against SDRElem<int, UnitData> rows).  (1) long ids that fit 32 bits: the twin serves the pair; (2) long ids beyond
    32 bits: the 32-bit batch is widened for the call.  A result holds ids of `this`'s width."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(41 + 2 * tu + au)
    t = Batch(*util.ragged_rows(rng, 33, 3000, 40, tu))
    a = Batch(*util.ragged_rows(rng, 400, 3000, 40, au))
    t64, t32 = SDRBatch.from_csr(t.off, t.ids.astype(np.int64), t.data), SDRBatch.from_csr(t.off, t.ids, t.data)
    a64, a32 = SDRBatch.from_csr(a.off, a.ids.astype(np.int64), a.data), SDRBatch.from_csr(a.off, a.ids, a.data)
    for dt, da, id_type in ((t64, a32, np.int64), (t32, a64, np.int32)):
        for op in range(4):
            assert np.array_equal(gpu_sizes(op, dt, da), oracle.port().size_op(op, t, a)), op
            assert np.array_equal(gpu_sizes(op, dt, da, swap=True), oracle.port().size_op(op, a, t).T), op
            off, ids, data = gpu_elems(op, dt, da)
            w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
            assert ids.dtype == id_type
            assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids.astype(id_type)), op
            assert (data is None) == (w_data is None) and (data is None or np.array_equal(data.view(np.uint32), w_data.view(np.uint32)))
        if tu or not au:    # (an EmptyData `this` compares with EmptyData rows only)
            assert np.array_equal(dt.equals(da).cpu().numpy().view(np.uint32), oracle.port().equal(t, a))
        for which, vname in ((0, "andv"), (1, "orv")):
            off, tp, ap = getattr(dt, vname)(da).to_host()
            w_off, w_tp, w_ap = oracle.port().visit(which, t, a)
            assert np.array_equal(off, w_off) and np.array_equal(tp, w_tp) and np.array_equal(ap, w_ap), vname

    # (2) every odd id of `this` moved beyond 32 bits (the rows stay ascending: the moved ids are larger than all
    # others); in the oracle's 32-bit model the same ids are moved by 100000, which keeps every comparison
    def moved(b, by):
        ids = b.ids.astype(np.int64)
        rows = [np.sort(np.where(r % 2 == 1, r + by, r)) for r in np.split(ids, b.off[1:-1].astype(np.int64))]
        order = [np.argsort(np.where(r % 2 == 1, r + by, r), kind="stable") for r in np.split(ids, b.off[1:-1].astype(np.int64))]
        data = None if b.data is None else np.concatenate([d[o] for d, o in zip(np.split(b.data, b.off[1:-1].astype(np.int64)), order)])
        return np.concatenate(rows), data
    far_ids, far_data = moved(t, np.int64(1) << 40)
    model_ids, model_data = moved(t, np.int64(100000))
    far = SDRBatch.from_csr(t.off, far_ids, far_data)
    model = Batch(t.off, model_ids.astype(np.int32), model_data)
    assert far.id_bits == 64
    for op in range(4):
        assert np.array_equal(gpu_sizes(op, far, a32), oracle.port().size_op(op, model, a)), op
        assert np.array_equal(gpu_sizes(op, a32, far), oracle.port().size_op(op, a, model)), op
        # `this` = the 64-bit batch: ids of the result are 64-bit
        off, ids, data = gpu_elems(op, far, a32)
        w_off, w_ids, w_data = oracle.port().elem_op(op, model, a)
        w_ids = w_ids.astype(np.int64)
        w_ids = np.where(w_ids >= 100000, w_ids - 100000 + (np.int64(1) << 40), w_ids)
        assert ids.dtype == np.int64 and np.array_equal(off, w_off) and np.array_equal(ids, w_ids), op
        assert (data is None) == (w_data is None) and (data is None or np.array_equal(data.view(np.uint32), w_data.view(np.uint32)))
        # `this` = the 32-bit batch: intersections and differences hold its ids; unions cannot
        if op in (AND, RM):
            off, ids, data = gpu_elems(op, a32, far)
            w_off, w_ids, w_data = oracle.port().elem_op(op, a, model)
            assert ids.dtype == np.int32 and np.array_equal(off, w_off) and np.array_equal(ids, w_ids), op
            assert (data is None) == (w_data is None) and (data is None or np.array_equal(data.view(np.uint32), w_data.view(np.uint32)))
        else:
            with pytest.raises(sdr_amd.SdrbError) as e:
                gpu_elems(op, a32, far)
            assert e.value.code == -5


def test_lanemerge_fill_signed_ids_chunks_of_a_and_sentinel():
    # ids of any sign without a declared width (32-bit staging), more rows of A than one block's chunk, a ragged
    # last wave of B; an id equal to INT32_MAX would collide with the row terminator, so such batches go elsewhere
    rng = np.random.default_rng(99)
    t = Batch(*util.ragged_rows(rng, 300, 5000, 40, lo=-2500))
    a = Batch(*util.ragged_rows(rng, 1111, 5000, 40, lo=-2500))
    dt, da = dev(t, 0), dev(a, 0)
    for op in (OR, XOR, RM):
        want = oracle.port().elem_op(op, t, a)
        assert_elems_equal(gpu_elems(op, dt, da, algo="lanemerge"), want)
        assert_elems_equal(gpu_elems(op, dt, da), want)
        off, ids, data = gpu_elems(op, dt, da, swap=True, algo="lanemerge")
        w_off, w_ids, w_data = oracle.port().elem_op(op, a, t)
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n),
                              util.pair_digests(w_off, w_ids, w_data).reshape(a.n, t.n).T)
    top = Batch(np.array([0, 2, 3], np.uint64), np.array([5, 2**31 - 1, 7], np.int32))
    dtop = dev(top, 0)
    with pytest.raises(sdr_amd.SdrbError):
        gpu_elems(OR, dtop, da, algo="lanemerge")
    assert_elems_equal(gpu_elems(OR, dtop, da), oracle.port().elem_op(OR, top, a))


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PAIRINGS))
def test_64_bit_batches_whose_ids_fit_32_bits_take_the_fast_kernels(name):
    """`long` ids that happen to be small (what the reference's own fuzzer feeds SDRElem<long>, src/fuzz.cpp:413-423):
    the batch keeps a narrowed twin, so every kernel of the 32-bit path serves it and the ids come back as int64;
    results equal the oracle's and those of the 64-bit merge kernels; a partner with huge ids keeps the pair there."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(31 + 2 * tu + au)
    t = Batch(*util.ragged_rows(rng, 40, 2048, 40, tu))
    a = Batch(*util.ragged_rows(rng, 300, 2048, 40, au))
    t64 = SDRBatch.from_csr(t.off, t.ids.astype(np.int64), t.data)
    a64 = SDRBatch.from_csr(a.off, a.ids.astype(np.int64), a.data)
    assert t64.id_bits == 64 and a64.id_bits == 64
    for op in range(4):
        want = oracle.port().size_op(op, t, a)
        assert np.array_equal(gpu_sizes(op, t64, a64), want), op
        assert np.array_equal(gpu_sizes(op, t64, a64, algo="merge"), want), op
        assert np.array_equal(gpu_sizes(op, t64, a64, swap=True), oracle.port().size_op(op, a, t).T), op
        w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
        for algo in (None, "merge"):
            off, ids, data = gpu_elems(op, t64, a64, algo=algo)
            assert ids.dtype == np.int64
            assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids.astype(np.int64)), (op, algo)
            assert (data is None) == (w_data is None) and (data is None or np.array_equal(data.view(np.uint32), w_data.view(np.uint32)))
    # all-pairs top-k, refused for 64-bit ids in general, runs on the twins
    c64, r64 = a64.allpairs_topk(a64, 4, exclude_offset=0)
    a32 = dev(Batch(a.off, a.ids, None), 2048)
    c32, r32 = a32.allpairs_topk(a32, 4, exclude_offset=0)
    assert np.array_equal(c64.cpu().numpy(), c32.cpu().numpy()) and np.array_equal(r64.cpu().numpy(), r32.cpu().numpy())
    # one side with ids beyond 32 bits: no twin there, the pair stays on the merge kernels and is still right
    shift = np.int64(1) << 40
    far = SDRBatch.from_csr(a.off, a.ids.astype(np.int64) + shift, a.data)
    assert np.array_equal(gpu_sizes(OR, t64, far), oracle.port().size_op(OR, t, Batch(a.off, a.ids + 5000, a.data)))
    # negative ids: a twin without a declared width
    neg_t = Batch(*util.ragged_rows(rng, 9, 300, 20, tu, lo=-150))
    neg_a = Batch(*util.ragged_rows(rng, 70, 300, 20, au, lo=-150))
    n_t = SDRBatch.from_csr(neg_t.off, neg_t.ids.astype(np.int64), neg_t.data)
    n_a = SDRBatch.from_csr(neg_a.off, neg_a.ids.astype(np.int64), neg_a.data)
    for op in range(4):
        assert np.array_equal(gpu_sizes(op, n_t, n_a), oracle.port().size_op(op, neg_t, neg_a)), op
        off, ids, data = gpu_elems(op, n_t, n_a)
        w_off, w_ids, w_data = oracle.port().elem_op(op, neg_t, neg_a)
        assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids.astype(np.int64)), op


@pytest.mark.gpu
@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("nq,width,len_t,len_a,na", [(1, 2048, 40, 40, 3000), (16, 2048, 40, 40, 1300), (32, 4096, 200, 64, 700),
                                                     (7, 64, 64, 64, 513), (3, 2048, 0, 40, 300), (5, 2000, 40, 0, 300),
                                                     (2, 2048, 1024, 33, 257)])
def test_fused_few_rows_ande_vs_oracle(name, nq, width, len_t, len_a, na):
    """ande of 1..32 rows against a database through the fused path (bytes per pair, a scan of per-group sums, offsets
    and elements in one enqueue): elements in id order, payload bits, both orientations, partial groups, and a second
    pass when the estimate-sized buffers turn out too small."""
    tu, au = PAIRINGS[name]
    rng = np.random.default_rng(11 + 2 * tu + au + width + nq)
    empty = lambda n, unit: Batch(np.zeros(n + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32) if unit else None)   # noqa: E731
    t = Batch(*util.ragged_rows(rng, nq, width, len_t, tu)) if len_t else empty(nq, tu)
    a = Batch(*util.ragged_rows(rng, na, width, len_a, au)) if len_a else empty(na, au)
    dt, da = dev(t, width), dev(a, width)
    res = dt.ande(da)
    assert res._total_dev is not None          # the fused path: total still on the device
    assert_elems_equal(res.to_host(), oracle.port().elem_op(AND, t, a))
    assert_elems_equal(gpu_elems(AND, dt, da, algo="merge"), oracle.port().elem_op(AND, t, a))
    off, ids, data = gpu_elems(AND, dt, da, swap=True)
    w_off, w_ids, w_data = oracle.port().elem_op(AND, a, t)
    assert np.array_equal(np.diff(off).reshape(t.n, a.n), np.diff(w_off).reshape(a.n, t.n).T)
    assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n),
                          util.pair_digests(w_off, w_ids, w_data).reshape(a.n, t.n).T)


@pytest.mark.gpu
def test_fused_few_rows_ande_refills_when_the_estimate_is_too_small():
    # every row of both batches holds the same 40 ids: 40 matches a pair where a uniform spread predicts 0.78
    ids = np.tile(np.arange(100, 140, dtype=np.int32), 600)
    off = np.arange(601, dtype=np.uint64) * 40
    t = Batch(off[:5], ids[: 4 * 40], None)
    a = Batch(off, ids, None)
    res = dev(t, 2048).ande(dev(a, 2048))
    assert res.total == 4 * 600 * 40 and res.ids.numel() == res.total
    assert_elems_equal(res.to_host(), oracle.port().elem_op(AND, t, a))


@pytest.mark.gpu
@pytest.mark.parametrize("nq,width,len_t,len_a,na", [(32, 64, 8, 4, 64), (17, 8, 8, 1, 64), (8, 2048, 40, 40, 5000)])
def test_fused_few_rows_ande_is_the_same_every_time(nq, width, len_t, len_a, na):
    """The scan between count and fill runs one workgroup a row of `this`, each adding up the sums of the rows before
    its own: those sums must still be there whenever a workgroup gets to run (a scan that wrote its starts over them
    was right almost always, and wrong when a workgroup started late).  Many rows, tiny work, many repeats."""
    import torch
    rng = np.random.default_rng(nq * 1000 + na)
    t = Batch(*util.ragged_rows(rng, nq, width, len_t, False))
    a = Batch(*util.ragged_rows(rng, na, width, len_a, True))
    dt, da = dev(t, width), dev(a, width)
    first = dt.ande(da)
    assert first._total_dev is not None
    assert_elems_equal(first.to_host(), oracle.port().elem_op(AND, t, a))
    for _ in range(400):
        again = dt.ande(da)
        assert torch.equal(again.offsets, first.offsets) and torch.equal(again.ids[: first.total], first.ids[: first.total])


@pytest.mark.gpu
@pytest.mark.parametrize("name", ["UU", "UE"])
@pytest.mark.parametrize("len_t,fused", [(310, True), (450, False)])
def test_few_rows_ande_with_more_payloads_than_16_bit_offsets_reach(name, len_t, fused):
    """The fused few-rows path finds a payload of A through a 16-bit byte offset into its tile.  32 UnitData rows of
    310 ids at width 4096 stay below 64 KiB (fused path); 32 rows of 450 ids do not: the entry point refuses them and
    ande takes another kernel -- equal to the oracle either way."""
    tu, au = PAIRINGS[name]
    width = 4096
    off_t, ids_t, data_t = sdr_amd.synth_csr(21, 0, 32, width, len_t, True)
    off_a, ids_a, data_a = sdr_amd.synth_csr(22, 0, 900, width, 64, au)
    t, a = Batch(off_t, ids_t, data_t), Batch(off_a, ids_a, data_a)
    dt, da = dev(t, width), dev(a, width)
    res = dt.ande(da)
    assert (res._total_dev is not None) == fused
    assert_elems_equal(res.to_host(), oracle.port().elem_op(AND, t, a))
    if not fused:
        import torch
        from sdr_amd import _lib
        total = torch.zeros(1, dtype=torch.int64, device="cuda")
        offs = torch.zeros(32 * 900 + 1, dtype=torch.int32, device="cuda")
        rc = _lib.load().sdrb_ande_few(dt._h, da._h, 0, offs.data_ptr(), None, None, 0, total.data_ptr(), None)
        assert rc == -5


@pytest.mark.gpu
def test_estimate_sized_ande_result_and_changed_inputs():
    """A few-rows ande result whose estimate was too small is computed again when its count is first read -- from
    the batches as they are THEN.  An upload in between must not yield the result of other data; with ret="unit" the
    ones are as long as the ids."""
    ids = np.tile(np.arange(100, 140, dtype=np.int32), 600)
    off = np.arange(601, dtype=np.uint64) * 40
    t, a = Batch(off[:5], ids[: 4 * 40], None), Batch(off, ids, None)
    dt, da = dev(t, 2048), dev(a, 2048)
    res = dt.ande(da)
    da.upload(off[:301], ids[: 300 * 40])
    with pytest.raises(RuntimeError):
        res.total
    da.upload(off, ids)
    res = dt.ande(da, ret="unit")
    assert res.total == 4 * 600 * 40 and res.ids.numel() == res.total and res.data.numel() == res.total
    assert bool((res.data == 1.0).all())
    assert_elems_equal((res.to_host()[0], res.to_host()[1], None), oracle.port().elem_op(AND, t, a))
