"""CPU tests: the plain-C oracle against the committed golden vectors (generated from the
real reference by tools/gen_golden.py), against the reference's own known answers, and --
where oracle/_ref has been built -- against the real reference on fresh random inputs."""
import os
import subprocess

import numpy as np
import pytest

import oracle
import util
from oracle import AND, OR, RM, XOR, Batch

PAIRINGS = {"EE": (False, False), "UU": (True, True), "UE": (True, False), "EU": (False, True)}


def fuzz_pair(n, name, seed):
    tu, au = PAIRINGS[name]
    return (Batch(*util.fuzz_universe(n, seed if tu else None)),
            Batch(*util.fuzz_universe(n, seed + 1 if au else None)))


@pytest.mark.parametrize("name", list(PAIRINGS))
def test_fuzz_grid_sizes(golden, name):
    g = golden("fuzz_sizes.npz")
    t, a = fuzz_pair(256, name, int(g["seed"]))
    for op in range(4):
        assert np.array_equal(oracle.port().size_op(op, t, a), g[name][op].astype(np.uint32))


@pytest.mark.parametrize("name", list(PAIRINGS))
def test_fuzz_grid_elems(golden, name):
    g = golden("fuzz_elems.npz")
    seed = int(g["seed"])
    t, a = fuzz_pair(128, name, seed)
    for op in range(4):
        off, ids, data = oracle.port().elem_op(op, t, a)
        assert np.array_equal(np.diff(off).reshape(128, 128), g[name + "_len"][op])
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(128, 128), g[name + "_digest"][op])
        # the reference's fuzzer checks size op == element op size (src/fuzz.cpp:281,309,338,367)
        assert np.array_equal(oracle.port().size_op(op, t, a), np.diff(off).reshape(128, 128))
    t, a = fuzz_pair(64, name, seed)
    for op in range(4):
        off, ids, data = oracle.port().elem_op(op, t, a)
        assert np.array_equal(off, g[f"{name}_full{op}_off"])
        assert np.array_equal(ids, g[f"{name}_full{op}_ids"])
        if data is not None:
            assert np.array_equal(data.view(np.uint32), g[f"{name}_full{op}_data"].view(np.uint32))


@pytest.mark.parametrize("fname", ["mini_2048_40.npz", "mini_16k_328.npz"])
def test_config_shaped_minis(golden, fname):
    g = golden(fname)
    nq, nd, width, active = int(g["nq"]), int(g["nd"]), int(g["width"]), int(g["active"])
    wd = bool(g["with_data"])
    q = Batch(*util.synth_rows_numpy(2, 0, nq, width, active, wd))
    d = Batch(*util.synth_rows_numpy(1, 0, nd, width, active, wd))
    assert np.array_equal(q.ids[:active], g["q_ids_head"])
    assert np.array_equal(d.ids[:active], g["d_ids_head"])
    for op in range(4):
        assert np.array_equal(oracle.port().size_op(op, q, d), g["sizes"][op])
        off, ids, data = oracle.port().elem_op(op, q, d)
        assert np.array_equal(np.diff(off).reshape(nq, nd), g[f"len{op}"])
        assert np.array_equal(util.pair_digests(off, ids, data).reshape(nq, nd), g[f"digest{op}"])
        assert np.array_equal(ids[: int(off[1])], g[f"pair00_ids{op}"])
    # mean overlap ~ active^2 / width is the generator's sanity checksum
    mean = oracle.port().size_op(AND, Batch(*util.synth_rows_numpy(2, 0, nq, width, active)),
                                 Batch(*util.synth_rows_numpy(1, 0, nd, width, active))).mean()
    assert abs(mean - active * active / width) < 0.15 * active * active / width


def test_inplace_golden(golden):
    g = golden("inplace.npz")
    seed = int(g["seed"])
    n = 32
    for name in PAIRINGS:
        t, a = fuzz_pair(n, name, seed)
        for op in range(4):
            for i in range(0, n, 3):
                ti, td = t.row(i)
                for j in range(n):
                    ai, ad = a.row(j)
                    r_ids, r_data = oracle.port_inplace(op, ti, td, ai, ad)
                    assert len(r_ids) == g[f"{name}_len{op}"][i, j]
                    dig = util.pair_digests(np.array([0, len(r_ids)], dtype=np.uint64), r_ids, r_data)[0]
                    assert dig == g[f"{name}_digest{op}"][i, j]
    for kind, s in (("E", None), ("U", seed)):
        u = Batch(*util.fuzz_universe(n, s))
        for op in range(4):
            for i in range(n):
                ids, data = u.row(i)
                r_ids, r_data = oracle.port_inplace(op, ids, data, None, None, alias=True)
                assert len(r_ids) == g[f"alias_{kind}_len{op}"][i]
                dig = util.pair_digests(np.array([0, len(r_ids)], dtype=np.uint64), r_ids, r_data)[0]
                assert dig == g[f"alias_{kind}_digest{op}"][i]


def _one(op, t_rows, a_rows, t_data=False, a_data=False):
    t = Batch.from_rows([t_rows], with_data=t_data)
    a = Batch.from_rows([a_rows], with_data=a_data)
    off, ids, data = oracle.port().elem_op(op, t, a)
    size = int(oracle.port().size_op(op, t, a)[0, 0])
    assert size == len(ids)
    return ids.tolist(), (None if data is None else data.tolist())


@pytest.mark.parametrize("tag", ["small", "c2like", "long"])
def test_inner_golden(golden, tag):
    """Sparse dot products (inner / diff_mul / row_major_mul_vec on ArithData<float>,
    SDR.hpp:1682-1692): the restatement reproduces the reference's float bits."""
    g = golden("inner.npz")
    t = oracle.Batch(g[f"{tag}_t_off"], g[f"{tag}_t_ids"], g[f"{tag}_t_data"])
    a = oracle.Batch(g[f"{tag}_a_off"], g[f"{tag}_a_ids"], g[f"{tag}_a_data"])
    assert np.array_equal(oracle.port().inner(t, a).view(np.uint32), g[f"{tag}_aa_bits"])
    assert np.array_equal(oracle.port().inner(t, oracle.Batch(a.off, a.ids, None)).view(np.uint32), g[f"{tag}_ae_bits"])
    # the order of the terms matters: a float64 dot product rounded once differs somewhere
    if tag != "small":
        return
    dense_t = np.zeros((t.n, 64)); dense_a = np.zeros((a.n, 64))
    for r in range(t.n):
        ids, data = t.row(r)
        dense_t[r, ids] = data
    for r in range(a.n):
        ids, data = a.row(r)
        dense_a[r, ids] = data
    approx = (dense_t @ dense_a.T).astype(np.float32)
    exact = g["small_aa_bits"].view(np.float32)
    assert np.allclose(approx, exact, rtol=1e-4, atol=1e-3)
    assert not np.array_equal(approx.view(np.uint32), g["small_aa_bits"])


@pytest.mark.parametrize("tag", ["e", "u"])
def test_queries_golden(golden, tag):
    """Range / point queries (SDR.hpp:860-956): the restatement against the reference's fixture."""
    g = golden("queries.npz")
    b = oracle.Batch(g[f"{tag}_off"], g[f"{tag}_ids"], g[f"{tag}_data"] if tag == "u" else None)
    counts, firsts = oracle.port().range_size(b, g[f"{tag}_lo"], g[f"{tag}_hi"])
    assert np.array_equal(counts, g[f"{tag}_counts"])
    pos = oracle.port().point(b, g[f"{tag}_qid"])
    assert np.array_equal(pos >= 0, g[f"{tag}_found"].astype(bool))


def test_readme_known_answers():
    # README.md:65-70
    assert _one(AND, [1, 2], [2, 3])[0] == [2]
    assert _one(OR, [1, 2], [2, 3])[0] == [1, 2, 3]
    assert _one(XOR, [1, 2], [2, 3])[0] == [1, 3]
    assert _one(RM, [1, 2], [2, 3])[0] == [1]
    # README.md:18-20
    assert _one(AND, [1, 2, 3, 4], [2, 3, 4, 5])[0] == [2, 3, 4]
    # README.md:113-128 -- UnitData multiplies on AND; < 0.1 is dropped; EmptyData acts as 1
    ids, data = _one(AND, ([0], [0.5]), ([0], [0.5]), True, True)
    assert ids == [0] and data == [0.25]
    assert _one(AND, ([0], [0.25]), ([0], [0.25]), True, True)[0] == []
    ids, data = _one(AND, ([0], [0.5]), [0], True, False)
    assert ids == [0] and data == [0.5]


def test_arith_set_algebra_vs_golden(golden):
    """ArithData<float> rows (ArithData.hpp:37-88): the restatement equals the reference's outputs."""
    g = golden("arith.npz")
    for tag, (ta, aa) in {"AA": (True, True), "AE": (True, False), "EA": (False, True)}.items():
        t = Batch(g[f"{tag}_t_off"], g[f"{tag}_t_ids"], g[f"{tag}_t_data"] if ta else None, arith=ta)
        a = Batch(g[f"{tag}_a_off"], g[f"{tag}_a_ids"], g[f"{tag}_a_data"] if aa else None, arith=aa)
        for op in (oracle.AND, oracle.OR, oracle.RM):
            assert np.array_equal(oracle.port().size_op(op, t, a), g[f"{tag}_size{op}"])
            off, ids, data = oracle.port().elem_op(op, t, a)
            assert np.array_equal(np.diff(off).astype(np.uint32), g[f"{tag}_len{op}"])
            assert np.array_equal(util.pair_digests(off, ids, data), g[f"{tag}_digest{op}"])
            assert np.array_equal(np.diff(off).reshape(t.n, a.n), g[f"{tag}_size{op}"])   # ands == ande.size() ...


def equal_cases(g):
    off, ids, data = util.fuzz_universe(64, int(g["seed"]))
    e, u, v = Batch(off, ids), Batch(off, ids, data), Batch(off, ids, g["changed"])
    x, y = Batch(off, ids, g["arith"], arith=True), Batch(off, ids, np.abs(g["arith"]), arith=True)
    return {"EE": (e, e), "UU": (u, v), "UE": (u, e), "AA": (x, y), "AE": (x, e)}


def test_operator_equal_vs_golden(golden):
    """SDR::operator== (SDR.hpp:470-483): ids position by position, payloads only where both sides have them."""
    g = golden("equal.npz")
    for tag, (t, a) in equal_cases(g).items():
        assert np.array_equal(oracle.port().equal(t, a), g[tag]), tag
    assert g["EE"].sum() == 64 and g["UE"].sum() == 64 and 0 < g["UU"].sum() < 64


def test_visitor_call_sequences_vs_golden(golden):
    """andv / orv (SDR.hpp:958-997, 1099-1166): the restatement calls its visitors with the positions the
    reference's own visitors received."""
    g = golden("visit.npz")
    t = Batch(*util.fuzz_universe(64, None))
    for which, name in ((0, "andv"), (1, "orv")):
        off, tp, ap = oracle.port().visit(which, t, t)
        assert np.array_equal(off, g[f"fuzz_{name}_off"])
        assert np.array_equal(tp, g[f"fuzz_{name}_this"]) and np.array_equal(ap, g[f"fuzz_{name}_arg"])
    q, d = Batch(g["q_off"], g["q_ids"]), Batch(g["d_off"], g["d_ids"])
    for which, name in ((0, "andv"), (1, "orv")):
        off, tp, ap = oracle.port().visit(which, q, d)
        assert np.array_equal(np.diff(off).astype(np.uint32), g[f"ragged_{name}_len"])
        assert np.array_equal(util.visit_digests(off, tp, ap), g[f"ragged_{name}_digest"])
    # andv reports each common id once, with positions that point at equal ids
    off, tp, ap = oracle.port().visit(0, q, d)
    pair = np.repeat(np.arange(q.n * d.n), np.diff(off).astype(np.int64))
    qi, dj = pair // d.n, pair % d.n
    assert np.array_equal(q.ids[q.off[qi].astype(np.int64) + tp], d.ids[d.off[dj].astype(np.int64) + ap])


def test_reference_unit_test_known_answers():
    # src/test.cpp:172-184 -- andv visits two pairs
    t = Batch.from_rows([[1, 2, 3]])
    a = Batch.from_rows([[2, 3, 4]])
    assert int(oracle.port().size_op(AND, t, a)[0, 0]) == 2
    # src/test.cpp:186-191 -- disjoint union
    assert _one(OR, [1, 2, 3], [4, 5, 6])[0] == [1, 2, 3, 4, 5, 6]
    # src/test.cpp:204-229 -- results for {1,2,3} op {2,3,4}
    assert _one(AND, [1, 2, 3], [2, 3, 4])[0] == [2, 3]
    assert _one(OR, [1, 2, 3], [2, 3, 4])[0] == [1, 2, 3, 4]
    assert _one(XOR, [1, 2, 3], [2, 3, 4])[0] == [1, 4]
    assert _one(RM, [1, 2, 3], [2, 3, 4])[0] == [1]
    # src/test.cpp:241-272 -- aliasing a.op(a)
    for op, want in ((RM, []), (XOR, []), (OR, [1, 2, 3]), (AND, [1, 2, 3])):
        ids, _ = oracle.port_inplace(op, [1, 2, 3], None, None, None, alias=True)
        assert ids.tolist() == want
    # src/test.cpp:397-433 -- {1,3} | {2,3} and {1,3} ^ {2,3}
    assert _one(OR, [1, 3], [2, 3])[0] == [1, 2, 3]
    assert _one(XOR, [1, 3], [2, 3])[0] == [1, 2]


def test_unit_vs_empty_rm_quirk():
    # UnitData `this` against an EmptyData arg: rmi is the identity (UnitData.hpp:112-115), so
    # matched elements stay when their own payload is relevant.
    ids, data = _one(RM, ([0, 1, 2, 5], [0.5, 0.25, 1.0, 0.3]), [1, 2, 3], True, False)
    assert ids == [0, 1, 2, 5]
    ids, data = _one(RM, ([0, 1, 2, 5], [0.5, 0.05, 1.0, 0.3]), [1, 2, 3], True, False)
    assert ids == [0, 2, 5]


def test_threshold_is_exactly_point_one_f32():
    below = float(np.nextafter(np.float32(0.1), np.float32(0)))
    for v, kept in ((0.1, True), (below, False), (float(np.nextafter(np.float32(0.1), np.float32(1))), True)):
        ids, _ = _one(AND, ([7], [v]), ([7], [1.0]), True, True)
        assert (ids == [7]) == kept


@pytest.mark.skipif(not oracle.have_reference(), reason="oracle/_ref not built (needs the reference checkout)")
def test_port_equals_reference_on_random_ragged():
    rng = np.random.default_rng(7)
    ref = oracle.reference(debug=True)
    for tu, au in ((False, False), (True, True), (True, False), (False, True)):
        t = Batch(*util.ragged_rows(rng, 40, 96, 48, tu, lo=-20))
        a = Batch(*util.ragged_rows(rng, 70, 96, 64, au, lo=-20))
        for op in range(4):
            assert np.array_equal(ref.size_op(op, t, a), oracle.port().size_op(op, t, a))
            r, p = ref.elem_op(op, t, a), oracle.port().elem_op(op, t, a)
            assert np.array_equal(r[0], p[0]) and np.array_equal(r[1], p[1])
            if r[2] is not None:
                assert np.array_equal(r[2].view(np.uint32), p[2].view(np.uint32))


@pytest.mark.skipif(not os.path.exists(oracle.REF_FUZZ), reason="reference fuzzer not built")
def test_reference_fuzzer_passes_here():
    # the reference's own exhaustive fuzzer (src/fuzz.cpp), the same invocation as its ctest
    r = subprocess.run([oracle.REF_FUZZ, "20"], capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "fail" not in r.stdout
