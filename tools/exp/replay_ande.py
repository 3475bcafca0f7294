#!/usr/bin/env python3
"""Lab: replays one soak case (seed) and shows which fill kernel disagrees with the oracle for ande."""
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
for p in (ROOT, os.path.join(ROOT, "tests")):
    sys.path.insert(0, p)
import numpy as np  # noqa: E402
import oracle  # noqa: E402
import util  # noqa: E402
from oracle import Batch  # noqa: E402
from sdr_amd import SDRBatch, SdrbError  # noqa: E402

seed = int(sys.argv[1])
rng = np.random.default_rng(seed)
width = int(rng.choice([8, 33, 64, 100, 256, 300, 1000, 2048, 2049, 4096, 4097, 9000, 17000, 20000, 30000]))
declared = width if rng.random() < 0.8 else 0
na = int(rng.choice([1, 2, 3, 8, 9, 16, 17, 31, 32, 33, 64, 65, 100]))
nb = int(rng.choice([1, 63, 64, 65, 255, 256, 257, 300, 511, 700]))
len_a = int(min(width, rng.choice([0, 1, 5, 20, 40, 41, 64, 65, 128, 129, 300])))
len_b = int(min(width, rng.choice([0, 1, 5, 20, 40, 41, 64, 65, 128, 300])))
tu, au = bool(rng.integers(2)), bool(rng.integers(2))
t = Batch(*util.ragged_rows(rng, na, width, len_a, tu))
a = Batch(*util.ragged_rows(rng, nb, width, len_b, au))
shifted = False
if declared == 0 and rng.random() < 0.5:
    t, a, shifted = Batch(t.off, t.ids - 3, t.data), Batch(a.off, a.ids - 3, a.data), True
print(dict(width=width, declared=declared, na=na, nb=nb, len_a=len_a, len_b=len_b, tu=tu, au=au, shifted=shifted))
dt = SDRBatch.from_csr(t.off, t.ids, t.data, width=declared)
da = SDRBatch.from_csr(a.off, a.ids, a.data, width=declared)
w_off, w_ids, w_data = oracle.port().elem_op(0, t, a)
for algo in (None, "merge", "rank", "bitmap", "lanemerge", "lanerank", "masktable"):
    try:
        off, ids, data = dt.ande(da, algo=algo).to_host()
    except SdrbError as e:
        print(algo, "n/a", e.code)
        continue
    ok_off, ok_ids = np.array_equal(off, w_off), np.array_equal(ids, w_ids)
    print(algo, "offsets", ok_off, "ids", ok_ids, "total", int(off[-1]), "want", int(w_off[-1]))
    if not (ok_off and ok_ids):
        bad = np.nonzero(np.diff(off) != np.diff(w_off))[0][:5]
        print("   first pairs with other lengths:", bad, np.diff(off)[bad], np.diff(w_off)[bad])
        if ok_off:
            k = np.nonzero(ids != w_ids)[0][:5]
            print("   first ids that differ at", k, ids[k], w_ids[k])
sizes = dt.ands(da).cpu().numpy().view(np.uint32)
print("ands ok", np.array_equal(sizes, oracle.port().size_op(0, t, a)))
