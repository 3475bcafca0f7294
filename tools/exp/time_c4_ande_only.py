import os, sys
sys.path.insert(0, "/root/repo")
import torch, sdr_amd
from sdr_amd import SDRBatch
w, act, nq, nd = 16384, 328, 64, 1 << 17
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, w, act, True), width=w)
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, w, act, True), width=w)
best = 1e9
for _ in range(6):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); r = q.ande(d); e1.record(); torch.cuda.synchronize()
    best = min(best, e0.elapsed_time(e1))
print(f"ande: {best:.3f} ms", flush=True)
