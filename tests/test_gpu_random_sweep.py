"""Randomised sweep: every size and element op on random shapes (widths that are not multiples of a word,
rows of exactly one or two wave lengths, empty rows, every payload pairing, both orientations), through the
default kernel choice and every forced kernel family that accepts the shape, against the oracle bit for bit.
Shapes are small enough for the oracle to finish in seconds."""
import numpy as np
import pytest

import oracle
import sdr_amd
import util
from oracle import AND, OR, RM, XOR, Batch
from sdr_amd import SDRBatch

pytestmark = pytest.mark.gpu

SIZE = {AND: "ands", OR: "ors", XOR: "xors", RM: "rms"}
ELEM = {AND: "ande", OR: "ore", XOR: "xore", RM: "rme"}


def rows(rng, n, width, max_len, kind):
    if kind == "arith":
        off, ids, data = util.arith_rows(rng, n, width, max_len)
        return Batch(off, ids, data, arith=True)
    return Batch(*util.ragged_rows(rng, n, width, max_len, kind == "unit"))


def dev(b: Batch, width):
    return SDRBatch.from_csr(b.off, b.ids, b.data, width=width, kind="arith" if b.kind == oracle.ARITH else None)


def try_algo(fn, algo):
    """The result, or None when the forced kernel family does not take the shape."""
    try:
        return fn(algo=algo)
    except sdr_amd.SdrbError as e:
        if algo is not None and e.code == -5:
            return None
        raise


@pytest.mark.parametrize("seed", range(70))
def test_random_shapes_all_kernels_vs_oracle(seed):
    rng = np.random.default_rng(1000 + seed)
    width = int(rng.choice([1, 7, 31, 32, 33, 63, 64, 65, 127, 500, 2047, 2048, 2049, 4096, 4097, 9000, 16384, 30001]))
    lens = [0, 1, 5, 40, 63, 64, 65, 128, 129, 300]
    len_a = min(int(rng.choice(lens)), width)
    len_b = min(int(rng.choice(lens)), width)
    kinds = [("empty", "empty"), ("unit", "unit"), ("unit", "empty"), ("empty", "unit"),
             ("arith", "arith"), ("arith", "empty"), ("empty", "arith")][seed % 7]
    na, nb = int(rng.integers(1, 40)), int(rng.integers(1, 260))
    t, a = rows(rng, na, width, len_a, kinds[0]), rows(rng, nb, width, len_b, kinds[1])
    declared = width if seed % 5 else 0          # every fifth case: no declared width (merge kernels only)
    dt, da = dev(t, declared), dev(a, declared)
    arith = "arith" in kinds
    for op in (AND, OR, RM) if arith else (AND, OR, XOR, RM):
        for swap in (False, True):
            this, arg = (a, t) if swap else (t, a)
            want = oracle.port().size_op(op, this, arg)
            want = want.T if swap else want
            for algo in (None, "merge", "bitslice", "popcount", "rowtable"):
                got = try_algo(lambda algo: getattr(dt, SIZE[op])(da, swap=swap, algo=algo), algo)
                if got is not None:
                    assert np.array_equal(got.cpu().numpy().view(np.uint32), want), (SIZE[op], swap, algo)
            w_off, w_ids, w_data = oracle.port().elem_op(op, this, arg)
            lens_want = np.diff(w_off).reshape(this.n, arg.n)
            dig_want = util.pair_digests(w_off, w_ids, w_data).reshape(this.n, arg.n)
            if swap:
                lens_want, dig_want = lens_want.T, dig_want.T
            for algo in (None, "merge", "rank", "rowtable", "bitmap", "lanemerge", "lanerank"):
                res = try_algo(lambda algo: getattr(dt, ELEM[op])(da, swap=swap, algo=algo), algo)
                if res is None:
                    continue
                off, ids, data = res.to_host()
                assert np.array_equal(np.diff(off).reshape(t.n, a.n), lens_want), (ELEM[op], swap, algo)
                assert np.array_equal(util.pair_digests(off, ids, data).reshape(t.n, a.n), dig_want), (ELEM[op], swap, algo)
                assert (data is None) == (w_data is None)
