"""The randomized parity sweeps of tests/manual/ as seeded, bounded GPU tests: each script draws shapes from a fixed seed,
compares the HIP path with the oracle bit for bit (see its header) and prints `done: N cases, M mismatches`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("script,cases,seed", [
    ("fuzz_bruteforce.py", 24, 11),   # small / ragged shards: every screen, both metrics, k up to 112
    ("fuzz_compat.py", 16, 12),       # reference-compat insert on every layer + search, incl. tie-heavy grids
    ("fuzz_ef.py", 24, 7),            # ef-mode beam search on random kNN graphs vs the oracle's ef spec: fp32 / bf16 rows, register and LDS beam kernels
    ("fuzz_stream.py", 6, 14),        # large shards: stream kernels (bf16 / int8 / fp4), guessed + verified thresholds, wide k
    ("fuzz_exact_scan.py", 24, 5),    # the certificate's repair pass alone: ragged shapes, 1 .. 100 listed queries, k up to 240, ties, every metric, bf16 rows
])
def test_seeded_fuzz_sweep(script, cases, seed):
    import zvdb_amd
    assert zvdb_amd.device_count() > 0
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tests", "manual", script), str(cases), str(seed)], cwd=ROOT,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert f"done: {cases} cases, 0 mismatches" in r.stdout, r.stdout[-3000:]
