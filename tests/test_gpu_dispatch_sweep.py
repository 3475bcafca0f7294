"""The default kernel choice at its boundaries.  sdrb_size_op / sdrb_elem_op pick a kernel from the shapes -- rows of A
(1, 2, 8 / 9, 16 / 17, 32 / 33), the longest row of B (40 / 41, 64 / 65), the declared width (2048 / 2049, 4096 / 4097,
none) and the payload kinds; every such choice has to give the reference's results, so the shapes on either side of each
threshold are run through the defaults (no algo=...) and compared with the CPU oracle, all four ops, both orientations."""
import itertools

import numpy as np
import pytest

import oracle
import util
from oracle import AND, Batch
from sdr_amd import SDRBatch

pytestmark = pytest.mark.gpu

OPS = {0: ("ands", "ande"), 1: ("ors", "ore"), 2: ("xors", "xore"), 3: ("rms", "rme")}
PAIRINGS = {"EE": (False, False), "UU": (True, True), "UE": (True, False), "EU": (False, True)}


def dev(b: Batch, width) -> SDRBatch:
    return SDRBatch.from_csr(b.off, b.ids, b.data, width=width)


def check_sizes(dt, da, t, a, tag):
    for op in range(4):
        got = getattr(dt, OPS[op][0])(da).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, oracle.port().size_op(op, t, a)), (tag, OPS[op][0])
        got = getattr(dt, OPS[op][0])(da, swap=True).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, oracle.port().size_op(op, a, t).T), (tag, OPS[op][0], "swap")


def check_elems(dt, da, t, a, tag):
    for op in range(4):
        off, ids, data = getattr(dt, OPS[op][1])(da).to_host()
        w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
        assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids), (tag, OPS[op][1])
        if w_data is None:
            assert data is None, (tag, OPS[op][1])
        else:
            assert np.array_equal(data.view(np.uint32), w_data.view(np.uint32)), (tag, OPS[op][1])


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("na", [1, 2, 8, 9, 16, 17, 32, 33])
def test_rows_of_a_thresholds(name, na):
    tu, au = PAIRINGS[name]
    for width, len_b in ((2048, 40), (4096, 64)):
        rng = np.random.default_rng(1000 * na + width + 2 * tu + au)
        t = Batch(*util.ragged_rows(rng, na, width, 40, tu))
        a = Batch(*util.ragged_rows(rng, 333, width, len_b, au))
        dt, da = dev(t, width), dev(a, width)
        check_sizes(dt, da, t, a, (name, na, width))
        check_elems(dt, da, t, a, (name, na, width))


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("len_b", [40, 41, 64, 65])
def test_row_length_thresholds(name, len_b):
    tu, au = PAIRINGS[name]
    for na, width in itertools.product((3, 20, 70), (2048, 4096)):
        rng = np.random.default_rng(77 * len_b + na + width + 2 * tu + au)
        # rows of exactly the threshold length among shorter ones
        a_rows = util.ragged_rows(rng, 257, width, len_b, au)
        t = Batch(*util.ragged_rows(rng, na, width, min(len_b, 128), tu))
        a = Batch(*a_rows)
        dt, da = dev(t, width), dev(a, width)
        check_sizes(dt, da, t, a, (name, len_b, na, width))
        if na != 70:
            check_elems(dt, da, t, a, (name, len_b, na, width))


@pytest.mark.parametrize("name", list(PAIRINGS))
@pytest.mark.parametrize("width,declared", [(2048, 2048), (2049, 2049), (4096, 4096), (4097, 4097), (17000, 17000),
                                            (18000, 18000), (28000, 28000), (29000, 29000), (4096, 0)])
def test_width_thresholds(name, width, declared):
    tu, au = PAIRINGS[name]
    for na in (2, 24):
        rng = np.random.default_rng(width + declared + na + 2 * tu + au)
        t = Batch(*util.ragged_rows(rng, na, width, 40, tu))
        a = Batch(*util.ragged_rows(rng, 300, width, 40, au))
        dt, da = dev(t, declared), dev(a, declared)
        check_sizes(dt, da, t, a, (name, width, declared, na))
        check_elems(dt, da, t, a, (name, width, declared, na))


@pytest.mark.parametrize("na", [1, 8, 9, 40])
@pytest.mark.parametrize("len_b", [16, 17, 32, 33, 40, 41, 64, 65])
def test_dot_product_thresholds(na, len_b):
    """sdrb_inner's default choice (few rows / match-visiting / dense table / per-pair merge) around its thresholds:
    float bits equal the oracle's, with ArithData and with EmptyData arguments."""
    for width, declared in ((2048, 2048), (4096, 4096), (4097, 4097), (300, 300), (4096, 0)):
        rng = np.random.default_rng(31 * na + len_b + width + declared)
        t_csr = util.arith_rows(rng, na, width, 40)
        a_csr = util.arith_rows(rng, 270, width, len_b)
        t = SDRBatch.from_csr(*t_csr, width=declared, kind="arith")
        a = SDRBatch.from_csr(*a_csr, width=declared, kind="arith")
        a_ids = SDRBatch.from_csr(a_csr[0], a_csr[1], None, width=declared)
        want = oracle.port().inner(Batch(*t_csr), Batch(*a_csr)).view(np.uint32)
        want_e = oracle.port().inner(Batch(*t_csr), Batch(a_csr[0], a_csr[1], None)).view(np.uint32)
        assert np.array_equal(t.inner(a).cpu().numpy().view(np.uint32), want), (na, len_b, width, declared)
        assert np.array_equal(t.inner(a_ids).cpu().numpy().view(np.uint32), want_e), (na, len_b, width, declared, "empty arg")


@pytest.mark.parametrize("top", [2 ** 31 - 2, 2 ** 31 - 1])
def test_64_bit_ids_at_the_edge_of_32_bits(top):
    """Batches of 64-bit ids whose values just fit (a narrowed twin exists: largest id 2^31 - 2) or just do not
    (2^31 - 1: the pair stays with the 64-bit kernels), smallest id -2^31: same results either way."""
    rng = np.random.default_rng(top % 1000)
    pool = np.array([-2 ** 31, -2 ** 31 + 1, -7, 0, 3, 2 ** 20, 2 ** 31 - 3, top], dtype=np.int64)

    def rows(n):
        off, ids = [0], []
        for _ in range(n):
            pick = np.sort(rng.choice(pool, rng.integers(0, len(pool) + 1), replace=False))
            ids.extend(pick.tolist())
            off.append(len(ids))
        return np.array(off, np.uint64), np.array(ids, np.int64)

    t_off, t_ids = rows(21)
    a_off, a_ids = rows(130)
    # one row on each side that surely holds both extremes
    t_off, t_ids = np.append(t_off, t_off[-1] + 2), np.append(t_ids, [-2 ** 31, top])
    a_off, a_ids = np.append(a_off, a_off[-1] + 3), np.append(a_ids, [-2 ** 31, 3, top])
    t64, a64 = SDRBatch.from_csr(t_off, t_ids, None), SDRBatch.from_csr(a_off, a_ids, None)
    assert t64.id_bits == 64
    t, a = Batch(t_off, t_ids.astype(np.int32), None), Batch(a_off, a_ids.astype(np.int32), None)
    for op in range(4):
        got = getattr(t64, OPS[op][0])(a64).cpu().numpy().view(np.uint32)
        assert np.array_equal(got, oracle.port().size_op(op, t, a)), OPS[op][0]
        off, ids, data = getattr(t64, OPS[op][1])(a64).to_host()
        w_off, w_ids, _ = oracle.port().elem_op(op, t, a)
        assert ids.dtype == np.int64 and data is None
        assert np.array_equal(off, w_off) and np.array_equal(ids, w_ids.astype(np.int64)), OPS[op][1]


@pytest.mark.parametrize("len_rows,width", [(64, 4096), (65, 4096), (40, 2048), (40, 4097), (64, 8192)])
def test_allpairs_default_choice_thresholds(len_rows, width):
    """All-pairs top-k through the default choice on either side of the bit-sliced kernel's limits (rows of 64 ids,
    width 4096): overlaps and rows equal the oracle's ranking; k = 16 is the longest list, k = 17 is refused."""
    import sdr_amd
    from test_gpu_allpairs import expected_topk
    rng = np.random.default_rng(len_rows + width)
    a_csr = util.ragged_rows(rng, 150, width, len_rows, False)
    b_csr = util.ragged_rows(rng, 1100, width, len_rows, False)
    a, b = SDRBatch.from_csr(*a_csr, width=width), SDRBatch.from_csr(*b_csr, width=width)
    counts = oracle.port().size_op(AND, Batch(a_csr[0], a_csr[1], None), Batch(b_csr[0], b_csr[1], None))
    for k in (1, 16):
        want_c, want_r = expected_topk(counts, k, -1)
        got_c, got_r = a.allpairs_topk(b, k)
        assert np.array_equal(got_r.cpu().numpy(), want_r) and np.array_equal(got_c.cpu().numpy(), want_c), k
    with pytest.raises(sdr_amd.SdrbError):
        a.allpairs_topk(b, 17)
