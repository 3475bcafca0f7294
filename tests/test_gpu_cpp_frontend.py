"""Runs the C++ SDRBatch front end (include/sdr_amd/SDRBatch.hpp) against the real reference:
oracle/_ref/test_sdrbatch builds SDR objects with the reference's own classes and containers,
evaluates all eight ops on the GPU and compares every pair with the reference's per-pair call.
The binary embeds the reference, so it exists only where oracle/_ref could be built."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "oracle", "_ref", "test_sdrbatch")


@pytest.mark.gpu
@pytest.mark.skipif(not os.path.exists(BIN), reason="oracle/_ref/test_sdrbatch not built (needs the reference checkout)")
def test_cpp_frontend_equals_reference_per_pair():
    r = subprocess.run([BIN], capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stdout + r.stderr
    assert "PASS" in r.stdout
