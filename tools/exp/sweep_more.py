#!/usr/bin/env python3
"""Lab (not part of the suite): the randomised all-kernel sweep of tests/test_gpu_random_sweep.py on further seeds.
usage: sweep_more.py [first_seed [last_seed]]   (default 70 .. 469)"""
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch  # noqa: E402,F401

import test_gpu_random_sweep as sweep  # noqa: E402

first = int(sys.argv[1]) if len(sys.argv) > 1 else 70
last = int(sys.argv[2]) if len(sys.argv) > 2 else 469
bad = 0
for seed in range(first, last + 1):
    try:
        sweep.test_random_shapes_all_kernels_vs_oracle(seed)
    except AssertionError as e:
        bad += 1
        print("FAIL seed", seed, e, flush=True)
    if seed % 50 == 0:
        print("seed", seed, flush=True)
print("done, failures:", bad)
sys.exit(1 if bad else 0)
