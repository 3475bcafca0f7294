#!/usr/bin/env python3
"""Perf lab (not shipped): config 4 ande fill through tools/exp/lane_queue_lab.hip against the product's fill.
usage: SDR_AMD_LIB=tools/exp/libsdr_amd_queue.so python tools/exp/time_queue_lab.py [rows_of_B [rows_of_A [empty]]]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch  # noqa: E402

import sdr_amd  # noqa: E402
from sdr_amd import SDRBatch, _lib  # noqa: E402

nd = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 17
nq = int(sys.argv[2]) if len(sys.argv) > 2 else 64
unit = not (len(sys.argv) > 3 and sys.argv[3] == "empty")
w, act = 16384, 328
d = SDRBatch.from_csr(*sdr_amd.synth_csr(1, 0, nd, w, act, unit), width=w)
q = SDRBatch.from_csr(*sdr_amd.synth_csr(2, 0, nq, w, act, unit), width=w)
lib = _lib.load()
fn = lib.sdrb_lab_lane_queue_fill
fn.restype = C.c_int
fn.argtypes = [C.c_void_p] * 6 + [C.c_int]


def timed(f, reps=5):
    f()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        r = f()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    return best, r


ms, ref = timed(lambda: q.ande(d))
total = ref.total
print(f"product ande (count + scan + fill + sort): {ms:.3f} ms, {total} elements", flush=True)
ms, _ = timed(lambda: q.ands(d))
print(f"product ands: {ms:.3f} ms", flush=True)
ids = torch.full((total,), -1, dtype=torch.int32, device="cuda")
data = torch.full((total,), -1.0, dtype=torch.float32, device="cuda") if unit else None
stream = torch.cuda.current_stream().cuda_stream


def lab(variant):
    _lib.check(fn(q._h, d._h, ref.offsets.data_ptr(), ids.data_ptr(), data.data_ptr() if unit else None, stream, variant))


def same():
    return bool(torch.equal(ids, ref.ids)) and (not unit or bool(torch.equal(data.view(torch.int32), ref.data.view(torch.int32))))


for variant, what in ((0, "rows where they lie"), (1, "rows dealt to the lanes, dealt in the call"), (2, "rows dealt to the lanes, walk alone")):
    ids.fill_(-1)
    ms, _ = timed(lambda: lab(variant))
    print(f"lab fill, {what}: {ms:.3f} ms  equal: {same()}", flush=True)
ok = bool(torch.equal(ids, ref.ids)) and (not unit or bool(torch.equal(data.view(torch.int32), ref.data.view(torch.int32))))
print("equal to the product's result:", ok, flush=True)
if not ok:
    bad = (ids != ref.ids).nonzero().flatten()
    print("first differing positions:", bad[:8].tolist(), "of", bad.numel())
    sys.exit(1)
