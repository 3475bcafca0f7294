"""Sparse dot products (SDR::inner / diff_mul / row_major_mul_vec on ArithData<float>,
SDR.hpp:1682-1692,1729-1750,1886-1923) through the C ABI: float BITS equal to the reference's,
from the golden fixture the reference generated and from the oracle on fresh inputs."""
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


def bits(t):
    return t.cpu().numpy().view(np.uint32)


@pytest.mark.parametrize("tag,width", [("small", 64), ("small", 0), ("c2like", 2048), ("c2like", 0), ("long", 16384)])
def test_inner_vs_golden(golden, tag, width):
    """width 0 = undeclared ids -> the per-pair merge kernel; a declared width -> the table kernel
    when rows are short enough ("long" rows exceed it and take the merge kernel too)."""
    from sdr_amd import SDRBatch
    g = golden("inner.npz")
    t = SDRBatch.from_csr(g[f"{tag}_t_off"], g[f"{tag}_t_ids"], g[f"{tag}_t_data"], width=width, kind="arith")
    a = SDRBatch.from_csr(g[f"{tag}_a_off"], g[f"{tag}_a_ids"], g[f"{tag}_a_data"], width=width, kind="arith")
    a_ids_only = SDRBatch.from_csr(g[f"{tag}_a_off"], g[f"{tag}_a_ids"], None, width=width)
    assert np.array_equal(bits(t.inner(a)), g[f"{tag}_aa_bits"])
    assert np.array_equal(bits(t.inner(a_ids_only)), g[f"{tag}_ae_bits"])


@pytest.mark.parametrize("width,max_len,nt,na", [(2048, 40, 37, 1000), (4096, 64, 5, 700), (300, 16, 130, 513),
                                                 (2048, 40, 1, 3000), (4099, 64, 8, 777), (33, 20, 3, 300),
                                                 (17001, 7, 2, 515), (2048, 64, 9, 300)])
def test_inner_table_kernel_vs_oracle(width, max_len, nt, na):
    """Ragged rows, partial 4-row tiles of A, partial blocks of B, one-vector (matrix x vector) shape; up to
    eight rows of A take the lane-per-row kernel (any declared width whose tables fit LDS), nine the table kernel."""
    import oracle
    from sdr_amd import SDRBatch
    rng = np.random.default_rng(width + nt)
    t_csr = util.arith_rows(rng, nt, width, max_len)
    a_csr = util.arith_rows(rng, na, width, max_len)
    t = SDRBatch.from_csr(*t_csr, width=width, kind="arith")
    a = SDRBatch.from_csr(*a_csr, width=width, kind="arith")
    want = oracle.port().inner(oracle.Batch(*t_csr), oracle.Batch(*a_csr))
    assert np.array_equal(bits(t.inner(a)), want.view(np.uint32))
    # an argument without payloads contributes the values of `this` alone
    a_ids_only = SDRBatch.from_csr(a_csr[0], a_csr[1], None, width=width)
    want_e = oracle.port().inner(oracle.Batch(*t_csr), oracle.Batch(a_csr[0], a_csr[1], None))
    assert np.array_equal(bits(t.inner(a_ids_only)), want_e.view(np.uint32))
    # both kernels agree: the same rows without a declared width go through the merge kernel
    t0 = SDRBatch.from_csr(*t_csr, kind="arith")
    a0 = SDRBatch.from_csr(*a_csr, kind="arith")
    assert np.array_equal(bits(t0.inner(a0)), want.view(np.uint32))


def test_inner_special_values_and_rejections():
    import oracle
    from sdr_amd import SDRBatch, SdrbError
    # signed zeros, denormals, cancellation, overflow to infinity inside a sum
    off = np.array([0, 4, 8], np.uint64)
    ids = np.array([1, 2, 3, 9, 1, 2, 3, 9], np.int32)
    t_data = np.array([-0.0, 1e-40, 3e38, 3e38, 1.0, -1.0, 1e20, 1e20], np.float32)
    a_data = np.array([5.0, 1e-5, 2.0, 2.0, -0.0, 0.0, 1e20, -1e20], np.float32)
    want = oracle.port().inner(oracle.Batch(off, ids, t_data), oracle.Batch(off, ids, a_data))
    for width in (0, 16):
        t = SDRBatch.from_csr(off, ids, t_data, width=width, kind="arith")
        a = SDRBatch.from_csr(off, ids, a_data, width=width, kind="arith")
        assert np.array_equal(bits(t.inner(a)), want.view(np.uint32)), width
    with pytest.raises(SdrbError) as e:
        SDRBatch.from_csr(off, ids, np.array([np.inf] * 8, np.float32), kind="arith")
    assert e.value.code == -7
    unit = SDRBatch.from_csr(off, ids, np.abs(np.clip(a_data, 0, 1)), width=16)
    arith = SDRBatch.from_csr(off, ids, t_data, width=16, kind="arith")
    with pytest.raises(SdrbError):
        arith.inner(unit)          # UnitData argument: not an ArithData product
    with pytest.raises(SdrbError):
        arith.xors(arith)          # ArithData has no xor (ArithData.hpp:110-114)
    with pytest.raises(SdrbError):
        arith.ands(unit)           # ArithData and UnitData do not combine


def test_inner_few_rows_against_a_long_database():
    """Matrix x vector at full length (1 Mi rows): the lane-per-row kernel's blocks stride over the rows;
    sampled entries equal the oracle bit for bit."""
    import oracle
    import sdr_amd
    from sdr_amd import SDRBatch
    nd, width, active = 1 << 20, 2048, 40
    t_csr = sdr_amd.synth_csr(2, 0, 2, width, active, True)
    a_csr = sdr_amd.synth_csr(1, 0, nd, width, active, True)
    t = SDRBatch.from_csr(*t_csr, width=width, kind="arith")
    a = SDRBatch.from_csr(*a_csr, width=width, kind="arith")
    got = bits(t.inner(a))
    rng = np.random.default_rng(3)
    rows = np.sort(rng.choice(nd, 500, replace=False))
    off, ids, data = a_csr
    lens = (off[rows + 1] - off[rows]).astype(np.int64)
    sub = oracle.Batch(np.concatenate([[0], np.cumsum(lens)]).astype(np.uint64),
                       np.concatenate([ids[int(off[r]):int(off[r + 1])] for r in rows]),
                       np.concatenate([data[int(off[r]):int(off[r + 1])] for r in rows]))
    want = oracle.port().inner(oracle.Batch(*t_csr), sub)
    assert np.array_equal(got[:, rows], want.view(np.uint32))


@pytest.mark.parametrize("width,len_t,len_a,nt,na", [(2048, 40, 40, 37, 1000), (4096, 64, 64, 70, 700), (300, 100, 16, 130, 513),
                                                     (2048, 128, 64, 33, 300), (64, 64, 64, 40, 257), (2048, 0, 40, 5, 300),
                                                     (2000, 40, 0, 64, 300)])
def test_inner_kernels_agree_with_the_oracle(width, len_t, len_a, nt, na):
    """The match-visiting kernel (32-row mask tiles), the dense value-table kernel and the per-pair merge on the
    same batches: partial tiles of A, partial blocks of B, rows of A longer than a wave, dense small widths where
    most ids of a tile are taken, empty rows and empty sides."""
    import oracle
    from sdr_amd import SDRBatch
    rng = np.random.default_rng(width + nt + len_t)
    t_csr = util.arith_rows(rng, nt, width, len_t) if len_t else (np.zeros(nt + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32))
    a_csr = util.arith_rows(rng, na, width, len_a) if len_a else (np.zeros(na + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32))
    t = SDRBatch.from_csr(*t_csr, width=width, kind="arith")
    a = SDRBatch.from_csr(*a_csr, width=width, kind="arith")
    a_ids_only = SDRBatch.from_csr(a_csr[0], a_csr[1], None, width=width)
    want = oracle.port().inner(oracle.Batch(*t_csr), oracle.Batch(*a_csr)).view(np.uint32)
    want_e = oracle.port().inner(oracle.Batch(*t_csr), oracle.Batch(a_csr[0], a_csr[1], None)).view(np.uint32)
    for algo in ("sparse", "table", "merge"):
        assert np.array_equal(bits(t.inner(a, algo=algo)), want), algo
        assert np.array_equal(bits(t.inner(a_ids_only, algo=algo)), want_e), algo


def test_inner_sparse_kernel_special_values_and_golden(golden):
    import oracle
    from sdr_amd import SDRBatch, SdrbError
    g = golden("inner.npz")
    for tag, width in (("small", 64), ("c2like", 2048)):
        t = SDRBatch.from_csr(g[f"{tag}_t_off"], g[f"{tag}_t_ids"], g[f"{tag}_t_data"], width=width, kind="arith")
        a = SDRBatch.from_csr(g[f"{tag}_a_off"], g[f"{tag}_a_ids"], g[f"{tag}_a_data"], width=width, kind="arith")
        assert np.array_equal(bits(t.inner(a, algo="sparse")), g[f"{tag}_aa_bits"])
    # signed zeros, denormals, cancellation, overflow to infinity inside a sum
    off = np.array([0, 4, 8], np.uint64)
    ids = np.array([1, 2, 3, 9, 1, 2, 3, 9], np.int32)
    t_data = np.array([-0.0, 1e-40, 3e38, 3e38, 1.0, -1.0, 1e20, 1e20], np.float32)
    a_data = np.array([5.0, 1e-5, 2.0, 2.0, -0.0, 0.0, 1e20, -1e20], np.float32)
    want = oracle.port().inner(oracle.Batch(off, ids, t_data), oracle.Batch(off, ids, a_data))
    t = SDRBatch.from_csr(off, ids, t_data, width=16, kind="arith")
    a = SDRBatch.from_csr(off, ids, a_data, width=16, kind="arith")
    assert np.array_equal(bits(t.inner(a, algo="sparse")), want.view(np.uint32))
    rng = np.random.default_rng(1)
    long_t = SDRBatch.from_csr(*util.arith_rows(rng, 3, 4096, 300), width=4096, kind="arith")
    some_a = SDRBatch.from_csr(*util.arith_rows(rng, 9, 4096, 40), width=4096, kind="arith")
    with pytest.raises(SdrbError):          # rows of `this` beyond what a tile's value store holds
        long_t.inner(some_a, algo="sparse")
