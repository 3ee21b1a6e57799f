"""CPU checks of the drop-in boundary: libzvdb_hip.so loads and exports EVERY symbol that
include/zvdb_hip.h declares; no compute call succeeds without a GPU (there is no CPU fallback)."""
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared_symbols():
    src = open(os.path.join(ROOT, "include", "zvdb_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(zvdb_hip_[a-z0-9_]+)\s*\(", src)))


def test_every_declared_symbol_is_exported_and_bound():
    import ctypes
    from zvdb_amd import _build, _lib
    _build.build_all()
    lib = ctypes.CDLL(_build.LIB_HIP)
    declared = _declared_symbols()
    assert len(declared) >= 25
    for name in declared:
        assert hasattr(lib, name), f"{name} declared in include/zvdb_hip.h but not exported"
    bound = {n for n, _, _ in _lib.HIP_SYMBOLS}
    assert set(declared) == bound, set(declared) ^ bound
    assert _lib.hip().zvdb_hip_abi_version() == 1


def test_no_cpu_fallback_without_device():
    import zvdb_amd as zv
    if zv.device_count() > 0:
        pytest.skip("a GPU is visible; the no-device path is exercised on the CPU box")
    with pytest.raises(zv.ZvdbError) as e:
        zv.Index(dim=8)
    assert e.value.code == 7  # ZVDB_ERR_NO_DEVICE
    h = zv.HNSW(16, 200)      # init is infallible and does no device work (hnsw.zig:52-62)
    assert h.count() == 0 and h.search([1, 2, 3], 5) == []  # empty index needs no device
    with pytest.raises(zv.ZvdbError):
        h.insert([1.0, 2.0, 3.0])


def test_product_package_never_imports_the_oracle():
    """Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may use oracle/ (comments
    that cite the oracle's spec are fine; includes, imports and loads are not)."""
    pkg = os.path.join(ROOT, "zvdb_amd")
    bad = re.compile(r"^\s*(import oracle|from oracle)|#\s*include[^\n]*zvdb_oracle|libzvdb_oracle", re.M)
    for dp, _, fns in os.walk(pkg):
        for fn in fns:
            if fn.endswith((".py", ".hip", ".cpp", ".hpp", ".h")) or fn == "Makefile":
                txt = open(os.path.join(dp, fn), errors="replace").read()
                assert not bad.search(txt), os.path.join(dp, fn)


def test_header_is_c99_and_a_c_program_links_against_the_library(tmp_path):
    """include/zvdb_hip.h is what Zig's @cImport parses: it must be C (not C++), and a C translation unit must link against
    libzvdb_hip.so.  tests/c/abi_smoke.c is compiled with gcc -std=c99 -pedantic -Wall -Wextra -Werror; without a GPU it exits 77."""
    import subprocess
    import zvdb_amd as zv
    from zvdb_amd import _build
    exe = _build.build_c_smoke(force=True)
    out = subprocess.run([exe, str(tmp_path / "x.zvdb")], capture_output=True, text=True, timeout=600)
    assert out.returncode == (0 if zv.device_count() > 0 else 77), out.stdout + out.stderr
