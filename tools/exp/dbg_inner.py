import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np, torch
from sdr_amd import SDRBatch
np.set_printoptions(linewidth=250)
# A: 12 rows, row q = ids {q, q+1, 20}, values 100*(q+1) + position
rows = [[q, q + 1, 20] for q in range(12)]
off = np.arange(13, dtype=np.uint64) * 3
ids = np.array(rows, np.int32).ravel()
vals = np.array([[100 * (q + 1) + p for p in range(3)] for q in range(12)], np.float32).ravel()
t = SDRBatch.from_csr(off, ids, vals, width=64, kind="arith")
# B: 6 rows, row j = {j} and {20}: values 1 and 0.5
brow = [[j, 20] for j in range(6)]
boff = np.arange(7, dtype=np.uint64) * 2
bids = np.array(brow, np.int32).ravel()
bvals = np.array([[1.0, 0.5]] * 6, np.float32).ravel()
a = SDRBatch.from_csr(boff, bids, bvals, width=64, kind="arith")
print("merge\n", t.inner(a, algo="merge").cpu().numpy())
print("sparse\n", t.inner(a, algo="sparse").cpu().numpy())
