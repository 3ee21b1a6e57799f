"""Runs the C++ restatement of the reference's src/test_hnsw.zig (tests/cpp/test_hnsw_host.cpp)
against the C++ host mirror + C ABI on the GPU."""
import os
import subprocess

import pytest

pytestmark = pytest.mark.gpu


def test_cpp_host_mirror_passes_reference_tests():
    from zvdb_amd import _build
    exe = _build.build_cpp_tests()
    assert os.path.exists(exe)
    out = subprocess.run([exe, "20000"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "ALL OK" in out.stdout
    for name in ("basic_functionality", "empty_index", "single_point", "edge_cases", "large_dataset",
                 "consistency", "concurrent_access", "stress", "memory_leaks"):
        assert f"ok {name}" in out.stdout


def test_pure_c99_caller_of_the_abi(tmp_path):
    """tests/c/abi_smoke.c (gcc -std=c99 -pedantic -Werror): create -> insert_batch -> search_graph -> brute force on two certified
    screens -> options -> save / load -> destroy, linked against libzvdb_hip.so -- the stand-in for Zig's @cImport of the header."""
    from zvdb_amd import _build
    exe = _build.build_c_smoke()
    out = subprocess.run([exe, str(tmp_path / "abi_smoke.zvdb")], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "abi_smoke ok" in out.stdout
