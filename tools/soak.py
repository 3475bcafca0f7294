#!/usr/bin/env python3
"""Randomised soak run on the GPU: random shapes, payload kinds, ops and orientations through the default kernel
choice -- sizes, elements, and in turn the visitors' records, operator==, 64-bit ids, dot products, the row-wise forms,
range and point queries, narrow counts, ids without a usable width -- every
result compared with the CPU oracle bit for bit.  usage: soak.py [seconds] [seed] [big|algos|long|-] [first case]
Prints a line per 50 cases and a summary; exits non-zero on the first mismatch (the failing case is printed so that
it can be replayed from its seed)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402

import oracle  # noqa: E402
import util  # noqa: E402
from oracle import Batch  # noqa: E402
from sdr_amd import SDRBatch  # noqa: E402

SIZE = ("ands", "ors", "xors", "rms")
ELEM = ("ande", "ore", "xore", "rme")


def main():
    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    big = len(sys.argv) > 3 and sys.argv[3] == "big"
    algos = len(sys.argv) > 3 and sys.argv[3] in ("algos", "long")
    long_rows = len(sys.argv) > 3 and sys.argv[3] == "long"
    t_end = time.time() + budget
    case = int(sys.argv[4]) if len(sys.argv) > 4 else 0     # replay: the case a failing seed names (seed - seed0 * 1000003)
    checks = 0
    while time.time() < t_end:
        seed = seed0 * 1000003 + case
        rng = np.random.default_rng(seed)
        width = int(rng.choice([8, 33, 64, 100, 256, 300, 1000, 2048, 2049, 4096, 4097, 9000, 17000, 20000, 30000]))
        declared = width if rng.random() < 0.8 else 0
        na = int(rng.choice([1, 2, 3, 8, 9, 16, 17, 31, 32, 33, 64, 65, 100]))
        nb = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 300, 511, 700]))
        if big:   # several tiles of A, many blocks of B, the benchmark's row shapes
            width = int(rng.choice([256, 2048, 4096, 16384]))
            declared = width
            na = int(rng.choice([33, 64, 65, 130]))
            nb = int(rng.choice([1025, 2500, 4000]))
        if long_rows:   # rows beyond one id per lane slot: the streamed bit-sliced kernel and its neighbours
            width = int(rng.choice([2048, 4096, 16384, 20000, 32768, 39000]))
            declared = width
            na = int(rng.choice([1, 7, 64, 65, 100]))
            nb = int(rng.choice([63, 257, 700, 1500]))
        len_a = int(min(width, rng.choice([0, 1, 5, 20, 40, 41, 64, 65, 128, 129, 300])))
        len_b = int(min(width, rng.choice([0, 1, 5, 20, 40, 41, 64, 65, 128, 300])))
        if long_rows:
            len_a = int(rng.choice([40, 328, 1000]))
            len_b = int(rng.choice([65, 100, 328, 600, 1500]))
        tu, au = bool(rng.integers(2)), bool(rng.integers(2))
        t = Batch(*util.ragged_rows(rng, na, width, len_a, tu)) if len_a else Batch(
            np.zeros(na + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32) if tu else None)
        a = Batch(*util.ragged_rows(rng, nb, width, len_b, au)) if len_b else Batch(
            np.zeros(nb + 1, np.uint64), np.zeros(0, np.int32), np.zeros(0, np.float32) if au else None)
        if declared == 0 and rng.random() < 0.5 and not long_rows:
            # no width to derive either: some ids negative (labels for the sizes, the merges for the elements)
            t = Batch(t.off, t.ids - 3, t.data)
            a = Batch(a.off, a.ids - 3, a.data)
        dt = SDRBatch.from_csr(t.off, t.ids, t.data, width=declared)
        da = SDRBatch.from_csr(a.off, a.ids, a.data, width=declared)
        tag = dict(seed=seed, width=width, declared=declared, na=na, nb=nb, len_a=len_a, len_b=len_b, tu=tu, au=au)
        for op in range(4):
            got = getattr(dt, SIZE[op])(da).cpu().numpy().view(np.uint32)
            if not np.array_equal(got, oracle.port().size_op(op, t, a)):
                print("MISMATCH", SIZE[op], tag, flush=True)
                return 1
            got = getattr(dt, SIZE[op])(da, swap=True).cpu().numpy().view(np.uint32)
            if not np.array_equal(got, oracle.port().size_op(op, a, t).T):
                print("MISMATCH", SIZE[op], "swap", tag, flush=True)
                return 1
            off, ids, data = getattr(dt, ELEM[op])(da).to_host()
            w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
            ok = np.array_equal(off, w_off) and np.array_equal(ids, w_ids) and (
                (data is None and w_data is None) or
                (data is not None and w_data is not None and np.array_equal(data.view(np.uint32), w_data.view(np.uint32))))
            if not ok:
                print("MISMATCH", ELEM[op], tag, flush=True)
                return 1
            checks += 3
        if algos:
            # every kernel that accepts the shapes, not only the one the library would choose
            from sdr_amd import SdrbError
            for op in range(4):
                want_s, want_sw = oracle.port().size_op(op, t, a), oracle.port().size_op(op, a, t).T
                for algo in ("merge", "bitslice", "bitslice_long", "popcount", "rowtable", "masktable", "lanerows"):
                    for swap, want in ((False, want_s), (True, want_sw)):
                        try:
                            got = getattr(dt, SIZE[op])(da, swap=swap, algo=algo).cpu().numpy().view(np.uint32)
                        except SdrbError as e:
                            if e.code != -5:
                                raise
                            continue
                        if not np.array_equal(got, want):
                            print("MISMATCH", SIZE[op], "algo", algo, "swap", swap, tag, flush=True)
                            return 1
                        checks += 1
                w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
                for algo in ("merge", "rank", "bitmap", "lanemerge", "lanerank", "masktable", "search"):
                    try:
                        off, ids, data = getattr(dt, ELEM[op])(da, algo=algo).to_host()
                    except SdrbError as e:
                        if e.code != -5:
                            raise
                        continue
                    ok = np.array_equal(off, w_off) and np.array_equal(ids, w_ids) and (
                        (data is None and w_data is None) or
                        (data is not None and w_data is not None and np.array_equal(data.view(np.uint32), w_data.view(np.uint32))))
                    if not ok:
                        print("MISMATCH", ELEM[op], "algo", algo, tag, flush=True)
                        return 1
                    checks += 1
        # now and then: the visitors' records, operator==, the same batches as 64-bit ids, dot products
        extra = case % 4
        if extra == 0:
            for which, name in ((0, "andv"), (1, "orv")):
                v = getattr(dt, name)(da)
                off, tp, ap = v.to_host()
                w_off, w_tp, w_ap = oracle.port().visit(which, t, a)
                if not (np.array_equal(off, w_off) and np.array_equal(tp, w_tp) and np.array_equal(ap, w_ap)):
                    print("MISMATCH", name, tag, flush=True)
                    return 1
            # (an EmptyData `this` compares with EmptyData rows only, as in the reference)
            if (tu or not au) and not np.array_equal(dt.equals(da).cpu().numpy().view(np.uint32), oracle.port().equal(t, a)):
                print("MISMATCH equals", tag, flush=True)
                return 1
            checks += 3
        elif extra == 1:
            t64 = SDRBatch.from_csr(t.off, t.ids.astype(np.int64) * 3 - 5, t.data)
            a64 = SDRBatch.from_csr(a.off, a.ids.astype(np.int64) * 3 - 5, a.data)
            op = case // 4 % 4
            got = getattr(t64, SIZE[op])(a64).cpu().numpy().view(np.uint32)
            off, ids, data = getattr(t64, ELEM[op])(a64).to_host()
            w_off, w_ids, w_data = oracle.port().elem_op(op, t, a)
            ok = np.array_equal(got, oracle.port().size_op(op, t, a)) and np.array_equal(off, w_off) and \
                np.array_equal(ids, w_ids.astype(np.int64) * 3 - 5) and \
                ((data is None and w_data is None) or np.array_equal(data.view(np.uint32), w_data.view(np.uint32)))
            if not ok:
                print("MISMATCH 64-bit ids", SIZE[op], tag, flush=True)
                return 1
            checks += 2
            # ids of different widths in one call (`long` this, `int` arg and the other way round)
            a32 = SDRBatch.from_csr(a.off, a.ids, a.data)
            t32 = SDRBatch.from_csr(t.off, t.ids, t.data)
            for this_b, arg_b, id_type in ((t64, a32, np.int64), (t32, a64, np.int32)):
                got = getattr(this_b, SIZE[op])(arg_b).cpu().numpy().view(np.uint32)
                off, ids, data = getattr(this_b, ELEM[op])(arg_b).to_host()
                # (the 32-bit batch holds the plain ids: compare through the oracle on matching id spaces)
                if id_type == np.int64:
                    want_s = oracle.port().size_op(op, Batch(t.off, (t.ids.astype(np.int64) * 3 - 5).astype(np.int32), t.data), a)
                    w2 = oracle.port().elem_op(op, Batch(t.off, (t.ids.astype(np.int64) * 3 - 5).astype(np.int32), t.data), a)
                else:
                    want_s = oracle.port().size_op(op, t, Batch(a.off, (a.ids.astype(np.int64) * 3 - 5).astype(np.int32), a.data))
                    w2 = oracle.port().elem_op(op, t, Batch(a.off, (a.ids.astype(np.int64) * 3 - 5).astype(np.int32), a.data))
                ok = np.array_equal(got, want_s) and np.array_equal(off, w2[0]) and ids.dtype == id_type and \
                    np.array_equal(ids, w2[1].astype(id_type)) and \
                    ((data is None and w2[2] is None) or np.array_equal(data.view(np.uint32), w2[2].view(np.uint32)))
                if not ok:
                    print("MISMATCH mixed id widths", SIZE[op], id_type, tag, flush=True)
                    return 1
                checks += 2
            # ids far from zero but close together (64-bit ids beyond 32 bits, 32-bit ids at the low end of their range):
            # moved to zero for the call, every kernel open to them, elements back as the ids they were
            for base, id_type in ((1 << 40, np.int64), (-(1 << 31) + 3, np.int32), (-(1 << 44) - 7, np.int64)):
                mv = lambda x: (x.astype(np.int64) + base).astype(id_type)   # noqa: E731
                tm = SDRBatch.from_csr(t.off, mv(t.ids), t.data)
                am = SDRBatch.from_csr(a.off, mv(a.ids), a.data)
                got = getattr(tm, SIZE[op])(am).cpu().numpy().view(np.uint32)
                off, ids, data = getattr(tm, ELEM[op])(am).to_host()
                ok = np.array_equal(got, oracle.port().size_op(op, t, a)) and np.array_equal(off, w_off) and ids.dtype == id_type and \
                    np.array_equal(ids, mv(w_ids)) and \
                    ((data is None and w_data is None) or np.array_equal(data.view(np.uint32), w_data.view(np.uint32)))
                if not ok:
                    print("MISMATCH moved ids", SIZE[op], base, tag, flush=True)
                    return 1
                checks += 2
            # ids no table can index: 32-bit ids spread over the whole signed range, 64-bit ids beyond 32 bits
            # (both keep the rows' order); sizes only depend on which ids rows share
            if len_a and len_b and width <= 4096:
                spread = lambda x: (x.astype(np.int64) * 1000003 - 2000000000).astype(np.int32)   # noqa: E731
                far = lambda x: x.astype(np.int64) * (1 << 33) - (1 << 40)                        # noqa: E731
                for name, f in (("spread", spread), ("far", far)):
                    tw = SDRBatch.from_csr(t.off, f(t.ids), t.data)
                    aw = SDRBatch.from_csr(a.off, f(a.ids), a.data)
                    # (elements: a wave per pair, ranks by binary search)
                    off, ids, data = getattr(tw, ELEM[op])(aw).to_host()
                    if not (np.array_equal(off, w_off) and np.array_equal(ids, f(w_ids)) and
                            ((data is None and w_data is None) or np.array_equal(data.view(np.uint32), w_data.view(np.uint32)))):
                        print("MISMATCH widthless elements", name, ELEM[op], tag, flush=True)
                        return 1
                    checks += 1
                    for swap in (False, True):
                        got = getattr(tw, SIZE[op])(aw, swap=swap).cpu().numpy().view(np.uint32)
                        want = oracle.port().size_op(op, a, t).T if swap else oracle.port().size_op(op, t, a)
                        if not np.array_equal(got, want):
                            print("MISMATCH widthless", name, SIZE[op], "swap", swap, tag, flush=True)
                            return 1
                        checks += 1
            # counts as bytes and as 16-bit words: the same numbers, or the refusal when a count could overflow
            from sdr_amd import SdrbError
            want = oracle.port().size_op(op, t, a)
            import torch
            served = (not tu or op == 1) and 0 < declared <= 4096 and len_b <= 64   # the bit-sliced kernel's shapes
            for dtype, top in ((torch.uint8, 255), (torch.int16, 65535)):
                try:
                    got = getattr(dt, SIZE[op])(da, dtype=dtype).cpu().contiguous().numpy()
                    got = got.view(np.uint16) if dtype == torch.int16 else got
                except SdrbError as e:
                    bound = len_a + len_b if op in (1, 2) else (len_a if op == 3 else min(len_a, len_b))
                    if e.code != -5 or (served and bound <= top):
                        print("MISMATCH narrow refusal", SIZE[op], dtype, e.code, tag, flush=True)
                        return 1
                    continue
                if not np.array_equal(got.astype(np.uint32), want):
                    print("MISMATCH narrow", SIZE[op], dtype, tag, flush=True)
                    return 1
                checks += 1
        elif extra == 2:
            from sdr_amd import SdrbError
            if len_a and len_b:
                ta = util.arith_rows(rng, na, width, len_a)
                aa = util.arith_rows(rng, nb, width, len_b)
                dta = SDRBatch.from_csr(*ta, width=declared, kind="arith")
                daa = SDRBatch.from_csr(*aa, width=declared, kind="arith")
                want = oracle.port().inner(Batch(*ta), Batch(*aa)).view(np.uint32)
                if not np.array_equal(dta.inner(daa).cpu().numpy().view(np.uint32), want):
                    print("MISMATCH inner", tag, flush=True)
                    return 1
                checks += 1
            # all-pairs overlap with the fused top-k (overlap descending, then the lower row), every kernel that takes the shapes
            if declared and len_a and len_b:
                k = int(rng.choice([1, 5, 16]))
                counts = oracle.port().size_op(0, Batch(t.off, t.ids, None), Batch(a.off, a.ids, None)).astype(np.int64)
                want_c, want_r = np.zeros((na, k), np.int64), np.full((na, k), -1, np.int64)
                for i in range(na):
                    order = np.lexsort((np.arange(nb), -counts[i]))[:k]
                    want_c[i, : len(order)], want_r[i, : len(order)] = counts[i][order], order
                for algo in (None, "bitslice", "mfma"):
                    try:
                        got_c, got_r = dt.allpairs_topk(da, k, algo=algo)
                    except SdrbError as e:
                        if e.code != -5:
                            raise
                        continue
                    if not (np.array_equal(got_c.cpu().numpy(), want_c) and np.array_equal(got_r.cpu().numpy(), want_r)):
                        print("MISMATCH allpairs_topk", algo, k, tag, flush=True)
                        return 1
                    checks += 1
            # the transpose: row c of the result lists the rows that hold id c, ascending, with their payloads
            if declared and len_b:
                tr = da.transpose()
                wire = tr.to_wire().cpu().numpy()
                n, nnz = tr.n_rows, tr.nnz
                rows_of = np.repeat(np.arange(nb, dtype=np.int64), np.diff(a.off).astype(np.int64))
                order = np.lexsort((rows_of, a.ids))
                w_off = np.concatenate(([0], np.cumsum(np.bincount(a.ids, minlength=declared))))
                ok = n == declared and nnz == a.ids.size and np.array_equal(wire[: n + 1].view(np.uint32), w_off) and \
                    np.array_equal(wire[n + 1: n + 1 + nnz], rows_of[order]) and \
                    (a.data is None or np.array_equal(wire[n + 1 + nnz:].view(np.uint32), a.data[order].view(np.uint32)))
                if not ok:
                    print("MISMATCH transpose", tag, flush=True)
                    return 1
                checks += 1
        elif extra == 3:
            # row by row (the in-place forms): the diagonal of the all-pairs results; range and point queries
            n = min(na, nb)
            if n:
                ts = Batch(t.off[: n + 1], t.ids[: int(t.off[n])], None if t.data is None else t.data[: int(t.off[n])])
                as_ = Batch(a.off[: n + 1], a.ids[: int(a.off[n])], None if a.data is None else a.data[: int(a.off[n])])
                dts = SDRBatch.from_csr(ts.off, ts.ids, ts.data, width=declared)
                das = SDRBatch.from_csr(as_.off, as_.ids, as_.data, width=declared)
                op = case // 4 % 4
                want = np.diagonal(oracle.port().size_op(op, ts, as_))
                if not np.array_equal(dts.rowwise_size(SIZE[op], das).cpu().numpy().view(np.uint32), want):
                    print("MISMATCH row-wise", SIZE[op], tag, flush=True)
                    return 1
                off, ids, data = dts.rowwise_elem(ELEM[op], das).to_host()
                w_off, w_ids, w_data = oracle.port().elem_op(op, ts, as_)
                for r in range(n):
                    p = r * n + r
                    lo, hi = int(w_off[p]), int(w_off[p + 1])
                    same = np.array_equal(ids[int(off[r]):int(off[r + 1])], w_ids[lo:hi]) and (
                        w_data is None or np.array_equal(data[int(off[r]):int(off[r + 1])].view(np.uint32), w_data[lo:hi].view(np.uint32)))
                    if not same:
                        print("MISMATCH row-wise", ELEM[op], "row", r, tag, flush=True)
                        return 1
                checks += 2
            if nb and len_b:
                lo = rng.integers(-5, width, 7).astype(np.int32)
                hi = (lo + rng.integers(0, width, 7)).astype(np.int32)
                counts, _ = oracle.port().range_size(a, lo, hi)
                if not np.array_equal(da.range_size(lo, hi).cpu().numpy().view(np.uint32), counts):
                    print("MISMATCH range_size", tag, flush=True)
                    return 1
                qid = rng.integers(-2, width + 2, 9).astype(np.int32)
                if not np.array_equal(da.point_query(qid).cpu().numpy(), oracle.port().point(a, qid)):
                    print("MISMATCH point_query", tag, flush=True)
                    return 1
                checks += 2
        case += 1
        if case % (5 if big else 50) == 0:
            print(f"{case} cases, {checks} comparisons, all equal", flush=True)
    print(f"soak done: {case} cases, {checks} comparisons, all equal", flush=True)
    return 0


if __name__ == "__main__":
    sys.exit(main())
