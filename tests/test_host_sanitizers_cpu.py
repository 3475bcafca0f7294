"""Host sanitizer preset: the C ABI's host code (argument checks, CSR database files, the synthetic
row generator) rebuilt with AddressSanitizer + UBSan and driven by tests/cpp/test_host_asan.cpp.
No GPU is needed: device code is not instrumented and no kernel is launched."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_host_code_is_clean_under_address_and_ub_sanitizers(tmp_path):
    env = dict(os.environ, TMPDIR=str(tmp_path))
    env.pop("ASAN_OPTIONS", None)
    run = subprocess.run(["make", "-C", os.path.join(ROOT, "sdr_amd", "csrc"), "asan", "-j8"],
                         env=env, capture_output=True, text=True, timeout=900)
    tail = (run.stdout + run.stderr)[-4000:]
    assert run.returncode == 0, tail
    assert "host checks passed" in run.stdout, tail
    assert "ERROR: AddressSanitizer" not in tail and "runtime error" not in tail, tail
