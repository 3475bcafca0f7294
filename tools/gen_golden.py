#!/usr/bin/env python3
"""Generates tests/golden/*.npz from the REAL reference (oracle/_ref/libsdr_ref_dbg.so, i.e.
jagprog5/SDR's own headers compiled with asserts on).  Run from the repo root in a
container that has /root/reference:

    make -C oracle && python tools/gen_golden.py

The fixtures hold inputs and expected outputs only.  While generating, the plain-C
restatement (oracle/sdr_oracle.c) is required to agree with the reference on every value
written -- that is what pins the oracle.

Fixtures
  fuzz_sizes.npz    exhaustive fuzz grid of src/fuzz.cpp:236-252 on an 8-id universe:
                    for all 256x256 pairs, the four size ops (uint8) for the payload pairings
                    EE, UU, UE, EU (this x arg; E = EmptyData, U = UnitData).
  fuzz_elems.npz    element ops on the 7-id universe (128x128 pairs): result lengths and
                    32-bit digests of every result (tests/util.py:pair_digests), plus the
                    complete results of the 6-id universe (64x64 pairs) as CSR.
  mini_2048_40.npz  config-shaped: 8 queries x 4096 rows, width 2048 / 40 active, EmptyData:
                    sizes + element-op lengths and digests.
  mini_16k_328.npz  4 queries x 512 rows, width 16384 / 328 active, UnitData x UnitData.
  inplace.npz       andi/ori/xori/rmi on a 5-id universe incl. the aliased a.op(a) cases.
  arith.npz         and / or / rm on ArithData<float> rows (sizes, element-op lengths and digests).
  equal.npz         operator== for every pair of the 6-id universe, with and without payloads.
  visit.npz         the calls of the reference's andv / orv visitors (positions in this and arg) on the
                    6-id fuzz universe (64x64 pairs, complete) and on ragged rows (lengths and digests).
"""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle  # noqa: E402
import util  # noqa: E402

OUT = os.path.join(ROOT, "tests", "golden")
FUZZ_SEED = 20240607
PAIRINGS = {"EE": (None, None), "UU": (FUZZ_SEED, FUZZ_SEED + 1), "UE": (FUZZ_SEED, None),
            "EU": (None, FUZZ_SEED + 1)}


def both(fn_name, *args):
    """Runs the entry point on the reference and on the restatement; they must agree."""
    ref = getattr(oracle.reference(debug=True), fn_name)(*args)
    mine = getattr(oracle.port(), fn_name)(*args)
    if isinstance(ref, tuple):
        for r, m in zip(ref, mine):
            assert (r is None and m is None) or np.array_equal(r, m), fn_name
    else:
        assert np.array_equal(ref, mine), fn_name
    return ref


def fuzz_batches(n, pairing):
    ts, as_ = PAIRINGS[pairing]
    return oracle.Batch(*util.fuzz_universe(n, ts)), oracle.Batch(*util.fuzz_universe(n, as_))


def gen_fuzz_sizes():
    out = {}
    for name in PAIRINGS:
        t, a = fuzz_batches(256, name)
        sizes = np.stack([both("size_op", op, t, a) for op in range(4)])
        assert sizes.max() < 256
        out[name] = sizes.astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "fuzz_sizes.npz"), seed=FUZZ_SEED, **out)


def gen_fuzz_elems():
    out = {}
    for name in PAIRINGS:
        t, a = fuzz_batches(128, name)
        lens, digs = [], []
        for op in range(4):
            off, ids, data = both("elem_op", op, t, a)
            lens.append(np.diff(off).astype(np.uint8).reshape(128, 128))
            digs.append(util.pair_digests(off, ids, data).reshape(128, 128))
        out[name + "_len"] = np.stack(lens)
        out[name + "_digest"] = np.stack(digs)
        t, a = fuzz_batches(64, name)
        for op in range(4):
            off, ids, data = both("elem_op", op, t, a)
            out[f"{name}_full{op}_off"] = off.astype(np.uint32)
            out[f"{name}_full{op}_ids"] = ids.astype(np.int8)
            if data is not None:
                out[f"{name}_full{op}_data"] = data
    np.savez_compressed(os.path.join(OUT, "fuzz_elems.npz"), seed=FUZZ_SEED, **out)


def gen_mini(fname, nq, nd, width, active, with_data):
    # db seed 1, query seed 2; independent numpy statement of the generator
    q = oracle.Batch(*util.synth_rows_numpy(2, 0, nq, width, active, with_data))
    d = oracle.Batch(*util.synth_rows_numpy(1, 0, nd, width, active, with_data))
    out = dict(nq=nq, nd=nd, width=width, active=active, with_data=with_data,
               q_ids_head=q.ids[:active].copy(), d_ids_head=d.ids[:active].copy())
    if with_data:
        out["q_data_head"] = q.data[:active].copy()
        out["d_data_head"] = d.data[:active].copy()
    out["sizes"] = np.stack([both("size_op", op, q, d) for op in range(4)]).astype(np.uint16)
    for op in range(4):
        off, ids, data = both("elem_op", op, q, d)
        out[f"len{op}"] = np.diff(off).astype(np.uint16).reshape(nq, nd)
        out[f"digest{op}"] = util.pair_digests(off, ids, data).reshape(nq, nd)
        # the complete result of pair (0, 0)
        out[f"pair00_ids{op}"] = ids[: int(off[1])].copy()
        if data is not None:
            out[f"pair00_data{op}"] = data[: int(off[1])].copy()
    np.savez_compressed(os.path.join(OUT, fname), **out)


def gen_inplace():
    ref = oracle.reference(debug=True)
    out = {}
    n = 32
    for name in PAIRINGS:
        t, a = fuzz_batches(n, name)
        for op in range(4):
            off, ids, data = ref.elem_op(op, t, a, inplace=True)
            # restatement, pair by pair
            k = 0
            for i in range(n):
                ti, td = t.row(i)
                for j in range(n):
                    ai, ad = a.row(j)
                    r_ids, r_data = oracle.port_inplace(op, ti, td, ai, ad)
                    lo, hi = int(off[k]), int(off[k + 1])
                    assert np.array_equal(r_ids, ids[lo:hi]), (name, op, i, j)
                    if data is not None:
                        assert np.array_equal(r_data, data[lo:hi]), (name, op, i, j)
                    k += 1
            out[f"{name}_len{op}"] = np.diff(off).astype(np.uint8).reshape(n, n)
            out[f"{name}_digest{op}"] = util.pair_digests(off, ids, data).reshape(n, n)
    # aliasing: a.op(a) with the same object on both sides (src/test.cpp:241-272)
    for kind, seed in (("E", None), ("U", FUZZ_SEED)):
        u = oracle.Batch(*util.fuzz_universe(n, seed))
        for op in range(4):
            lens, digs = [], []
            for i in range(n):
                ids, data = u.row(i)
                buf_ids = np.zeros(2 * len(ids) + 1, dtype=np.int32)
                buf_ids[: len(ids)] = ids
                buf_data = np.zeros(2 * len(ids) + 1, dtype=np.float32)
                if data is not None:
                    buf_data[: len(ids)] = data
                ln = ref.lib.rsdr_inplace_self(op, buf_ids.ctypes.data, buf_data.ctypes.data, len(ids),
                                               oracle.UNIT if data is not None else oracle.EMPTY)
                p_ids, p_data = oracle.port_inplace(op, ids, data, None, None, alias=True)
                assert np.array_equal(p_ids, buf_ids[:ln]), (kind, op, i)
                if data is not None:
                    assert np.array_equal(p_data, buf_data[:ln]), (kind, op, i)
                lens.append(ln)
                digs.append(util.pair_digests(np.array([0, ln], dtype=np.uint64), buf_ids[:ln],
                                              buf_data[:ln] if data is not None else None)[0])
            out[f"alias_{kind}_len{op}"] = np.asarray(lens, dtype=np.uint8)
            out[f"alias_{kind}_digest{op}"] = np.asarray(digs, dtype=np.uint32)
    np.savez_compressed(os.path.join(OUT, "inplace.npz"), seed=FUZZ_SEED, **out)


def gen_inner():
    """inner / diff_mul / row_major_mul_vec on ArithData<float> rows (SDR.hpp:1682-1692,1729-1750,
    1886-1923): float bits of every pair, reference == restatement."""
    rng = np.random.default_rng(FUZZ_SEED + 7)
    out = {}
    for tag, (nt, na, width, max_len) in {"small": (48, 96, 64, 40), "c2like": (16, 512, 2048, 40),
                                          "long": (6, 64, 16384, 328)}.items():
        t = oracle.Batch(*util.arith_rows(rng, nt, width, max_len))
        a = oracle.Batch(*util.arith_rows(rng, na, width, max_len))
        a_ids_only = oracle.Batch(a.off, a.ids, None)
        for kind, arg in (("aa", a), ("ae", a_ids_only)):
            ref = oracle.reference(debug=True).inner(t, arg)
            mine = oracle.port().inner(t, arg)
            assert np.array_equal(ref.view(np.uint32), mine.view(np.uint32)), (tag, kind)
            out[f"{tag}_{kind}_bits"] = ref.view(np.uint32)
        for name, b in (("t", t), ("a", a)):
            out[f"{tag}_{name}_off"], out[f"{tag}_{name}_ids"], out[f"{tag}_{name}_data"] = b.off, b.ids, b.data
    np.savez_compressed(os.path.join(OUT, "inner.npz"), **out)


def gen_queries():
    """Range and point queries (SDR.hpp:860-956): the reference's ands(lo,hi) / ande(lo,hi) / ande(id) on
    ragged rows with signed ids; the restatement must give the same counts, slices and hits."""
    rng = np.random.default_rng(FUZZ_SEED + 11)
    out = {}
    for tag, with_data in (("e", False), ("u", True)):
        b = oracle.Batch(*util.ragged_rows(rng, 300, 400, 60, with_data, lo=-150))
        lo = rng.integers(-170, 270, 40).astype(np.int32)
        hi = (lo + rng.integers(0, 200, 40)).astype(np.int32)
        lo[:3], hi[:3] = [-1000, 0, 249], [1000, 0, 250]          # everything, an empty range, one id
        qid = rng.integers(-160, 260, 50).astype(np.int32)
        ref_counts, ref_digests = oracle.reference(debug=True).ref_range(b, lo, hi)
        counts, firsts = oracle.port().range_size(b, lo, hi)
        assert np.array_equal(ref_counts, counts)
        # the slices the restatement implies reproduce the reference's ande(lo, hi)
        dig = np.zeros_like(ref_digests)
        for r in range(lo.size):
            for j in range(b.n):
                ids, _ = b.row(j)
                sl = ids[firsts[r, j]: firsts[r, j] + counts[r, j]].astype(np.int64)
                dig[r, j] = int((np.arange(1, sl.size + 1) * sl).sum()) + sl.size * 1000003
        assert np.array_equal(dig, ref_digests)
        found, value = oracle.reference(debug=True).ref_point(b, qid)
        pos = oracle.port().point(b, qid)
        assert np.array_equal(found.astype(bool), pos >= 0)
        if with_data:
            for q in range(qid.size):
                for j in range(b.n):
                    if pos[q, j] >= 0:
                        assert b.row(j)[1][pos[q, j]] == value[q, j]
        out.update({f"{tag}_off": b.off, f"{tag}_ids": b.ids, f"{tag}_lo": lo, f"{tag}_hi": hi, f"{tag}_qid": qid,
                    f"{tag}_counts": ref_counts, f"{tag}_digests": ref_digests, f"{tag}_found": found,
                    f"{tag}_value": value})
        if with_data:
            out[f"{tag}_data"] = b.data
    np.savez_compressed(os.path.join(OUT, "queries.npz"), **out)


def gen_long_ids():
    """Size and element ops on rows with `long` ids (SDRElem<long, ...>, src/fuzz.cpp:413-423): ids far
    outside the 32-bit range, all four payload pairings; outputs of the reference itself."""
    rng = np.random.default_rng(FUZZ_SEED + 17)
    out = {}
    universe = np.sort(rng.choice(np.arange(-2 ** 40, 2 ** 40, 2 ** 33 + 12345, dtype=np.int64), 90, replace=False))
    for name, (tu, au) in {"EE": (False, False), "UU": (True, True), "UE": (True, False), "EU": (False, True)}.items():
        t32 = util.ragged_rows(rng, 24, 90, 40, tu)
        a32 = util.ragged_rows(rng, 60, 90, 40, au)
        t = (t32[0], universe[t32[1]], t32[2])
        a = (a32[0], universe[a32[1]], a32[2])
        for side, csr in (("t", t), ("a", a)):
            out[f"{name}_{side}_off"], out[f"{name}_{side}_ids"] = csr[0], csr[1]
            if csr[2] is not None:
                out[f"{name}_{side}_data"] = csr[2]
        for op in range(4):
            out[f"{name}_size{op}"] = oracle.reference(debug=True).ref_op64(op, False, t, a)
            off, ids, data = oracle.reference(debug=True).ref_op64(op, True, t, a)
            out[f"{name}_off{op}"], out[f"{name}_ids{op}"] = off, ids
            if data is not None:
                out[f"{name}_data{op}"] = data
            # ids only matter through their order: the 32-bit restatement on the ranks agrees
            want = oracle.port().size_op(op, oracle.Batch(*t32), oracle.Batch(*a32))
            assert np.array_equal(want, out[f"{name}_size{op}"])
            w_off, w_ids, w_data = oracle.port().elem_op(op, oracle.Batch(*t32), oracle.Batch(*a32))
            assert np.array_equal(w_off, off) and np.array_equal(universe[w_ids], ids)
            assert (w_data is None and data is None) or np.array_equal(w_data.view(np.uint32), data.view(np.uint32))
    np.savez_compressed(os.path.join(OUT, "long_ids.npz"), **out)


def gen_transpose():
    """SDR::transpose (SDR.hpp:1807-1826) on a matrix stored as batch rows.  The reference emits every
    column's elements in the pop order of its priority queue (MatrixUtils.hpp:13-75), which for equal
    column ids is heap order, not row order; the fixture holds the reference's columns with their
    elements sorted by row (the order its own 2x2 known answers show, src/test.cpp:517-571)."""
    rng = np.random.default_rng(FUZZ_SEED + 13)
    out = {}
    for tag, (n, width, max_len) in {"small": (300, 64, 30), "wide": (500, 5000, 60)}.items():
        b = oracle.Batch(*util.arith_rows(rng, n, width, max_len))
        col_off, rows, vals = oracle.reference(debug=False).ref_transpose(b, width)
        assert int(col_off[-1]) == b.ids.size
        for c in range(width):
            lo, hi = int(col_off[c]), int(col_off[c + 1])
            order = np.argsort(rows[lo:hi], kind="stable")
            rows[lo:hi], vals[lo:hi] = rows[lo:hi][order], vals[lo:hi][order]
        out.update({f"{tag}_off": b.off, f"{tag}_ids": b.ids, f"{tag}_data": b.data, f"{tag}_width": width,
                    f"{tag}_col_off": col_off, f"{tag}_rows": rows, f"{tag}_vals": vals})
    np.savez_compressed(os.path.join(OUT, "transpose.npz"), **out)


def gen_arith():
    """The set algebra on ArithData<float> rows (ArithData.hpp:37-88): and = product, or = sum, rm = difference,
    every payload relevant, no xor; pairings arith x arith, arith x empty, empty x arith."""
    rng = np.random.default_rng(FUZZ_SEED + 31)
    out = {}
    for tag, (ta, aa) in {"AA": (True, True), "AE": (True, False), "EA": (False, True)}.items():
        t_off, t_ids, t_data = util.arith_rows(rng, 24, 300, 60)
        a_off, a_ids, a_data = util.arith_rows(rng, 90, 300, 60)
        t = oracle.Batch(t_off, t_ids, t_data if ta else None, arith=ta)
        a = oracle.Batch(a_off, a_ids, a_data if aa else None, arith=aa)
        out.update({f"{tag}_t_off": t.off, f"{tag}_t_ids": t.ids, f"{tag}_a_off": a.off, f"{tag}_a_ids": a.ids})
        if ta:
            out[f"{tag}_t_data"] = t.data
        if aa:
            out[f"{tag}_a_data"] = a.data
        for op in (oracle.AND, oracle.OR, oracle.RM):
            out[f"{tag}_size{op}"] = both("size_op", op, t, a)
            off, ids, data = both("elem_op", op, t, a)
            out[f"{tag}_len{op}"] = np.diff(off).astype(np.uint32)
            out[f"{tag}_digest{op}"] = util.pair_digests(off, ids, data)
    np.savez_compressed(os.path.join(OUT, "arith.npz"), **out)


def gen_equal():
    """SDR::operator== (SDR.hpp:470-483) on the 6-id fuzz universe: ids alone, payloads on both sides (a copy
    with some payloads changed, signed zeros and all), payloads against none, and ArithData rows."""
    out = {}
    off, ids, data = util.fuzz_universe(64, FUZZ_SEED)
    changed = data.copy()
    changed[::5] = np.float32(0.75)
    changed[3::11] = data[3::11]                       # keep some rows fully equal
    e = oracle.Batch(off, ids)
    u = oracle.Batch(off, ids, data)
    v = oracle.Batch(off, ids, changed)
    rng = np.random.default_rng(FUZZ_SEED + 41)
    arith = (rng.integers(-2, 3, ids.size)).astype(np.float32)
    arith[arith == 0] = np.float32(-0.0)               # -0 == +0 in the comparison
    x = oracle.Batch(off, ids, arith, arith=True)
    y = oracle.Batch(off, ids, np.abs(arith), arith=True)
    out.update(changed=changed, arith=arith)
    for tag, (t, a) in {"EE": (e, e), "UU": (u, v), "UE": (u, e), "AA": (x, y), "AE": (x, e)}.items():
        out[tag] = both("equal", t, a).astype(np.uint8)
    np.savez_compressed(os.path.join(OUT, "equal.npz"), seed=FUZZ_SEED, **out)


def gen_visit():
    """What andv / orv call their visitors with (SDR.hpp:186-194,268-269), taken from the reference's own
    visitors on std::vector rows; the restatement must produce the same call sequences."""
    out = {}
    t = oracle.Batch(*util.fuzz_universe(64, None))
    for which, name in ((0, "andv"), (1, "orv")):
        off, tp, ap = both("visit", which, t, t)
        out.update({f"fuzz_{name}_off": off, f"fuzz_{name}_this": tp, f"fuzz_{name}_arg": ap})
    rng = np.random.default_rng(FUZZ_SEED + 21)
    q = oracle.Batch(*util.ragged_rows(rng, 40, 3000, 150, False))
    d = oracle.Batch(*util.ragged_rows(rng, 300, 3000, 150, False))
    out.update({"q_off": q.off, "q_ids": q.ids, "d_off": d.off, "d_ids": d.ids})
    for which, name in ((0, "andv"), (1, "orv")):
        off, tp, ap = both("visit", which, q, d)
        out.update({f"ragged_{name}_len": np.diff(off).astype(np.uint32), f"ragged_{name}_digest": util.visit_digests(off, tp, ap)})
    np.savez_compressed(os.path.join(OUT, "visit.npz"), **out)


def main():
    os.makedirs(OUT, exist_ok=True)
    if not oracle.have_reference():
        sys.exit("oracle/_ref is not built: this script needs /root/reference (make -C oracle)")
    if sys.argv[1:] == ["long_ids"]:
        gen_long_ids()
        return
    if sys.argv[1:] == ["transpose"]:
        gen_transpose()
        return
    if sys.argv[1:] == ["queries"]:     # only the fixture of the range / point queries
        gen_queries()
        return
    if sys.argv[1:] == ["inner"]:       # only the fixture of the sparse dot products
        gen_inner()
        return
    if sys.argv[1:] == ["visit"]:
        gen_visit()
        return
    if sys.argv[1:] == ["arith"]:
        gen_arith()
        return
    if sys.argv[1:] == ["equal"]:
        gen_equal()
        return
    gen_equal()
    gen_arith()
    gen_visit()
    gen_inner()
    gen_queries()
    gen_transpose()
    gen_long_ids()
    gen_fuzz_sizes()
    gen_fuzz_elems()
    gen_mini("mini_2048_40.npz", 8, 4096, 2048, 40, False)
    gen_mini("mini_16k_328.npz", 4, 512, 16384, 328, True)
    gen_inplace()
    for f in sorted(os.listdir(OUT)):
        print(f, os.path.getsize(os.path.join(OUT, f)))


if __name__ == "__main__":
    main()
