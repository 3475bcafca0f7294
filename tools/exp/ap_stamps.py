#!/usr/bin/env python3
"""Diagnostic: phase cycle counters of block 0 / wave 0 of the all-pairs kernel
(library built by tools/exp/build_ap_variant.sh stamps -DSDRB_AP_EXP_STAMPS)."""
import ctypes as C
import os
import sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
os.environ["SDR_AMD_LIB"] = os.path.join(ROOT, "tools/exp/libsdr_amd_ap_stamps.so")
import torch
import sdr_amd
from sdr_amd import SDRBatch, _lib

na, nb = 32768, 262144
off, ids, _ = sdr_amd.synth_csr(1, 0, nb, 2048, 40)
b = SDRBatch.from_csr(off, ids, None, width=2048)
a = SDRBatch.from_csr(off[: na + 1].copy(), ids[: int(off[na])], None, width=2048)
lib = _lib.load()
out = (C.c_ulonglong * 8)()
a.allpairs_topk(b, 16, exclude_offset=0)
torch.cuda.synchronize()
lib.sdrb_debug_read_stamps(out, 1)
a.allpairs_topk(b, 16, exclude_offset=0)
torch.cuda.synchronize()
lib.sdrb_debug_read_stamps(out, 1)
names = ["stage loop", "epilogue", "barrier after", "park", "prefilter", "detail", "-", "-"]
for n, v in zip(names, out):
    print(f"{n:14s} {int(v):>14d}")
