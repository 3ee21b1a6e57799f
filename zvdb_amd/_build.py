"""In-tree build of the native libraries (gfx950 only).

  libzvdb_hip.so   hipcc, HIP kernels + C ABI (include/zvdb_hip.h)        -- the drop-in boundary
  libzvdb_host.so  g++,  C wrappers over host/hnsw.hpp (C++ HNSW(T) mirror) -- for ctypes callers
  tests/cpp/test_hnsw_host  g++, C++ restatement of the reference's test_hnsw.zig over hnsw.hpp
  tests/c/abi_smoke         gcc -std=c99 -pedantic, a pure C caller of include/zvdb_hip.h (what @cImport sees)
"""
from __future__ import annotations

import os
import subprocess

PKG = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(PKG)
CSRC = os.path.join(PKG, "csrc")
LIB_HIP = os.path.join(PKG, "libzvdb_hip.so")
LIB_HOST = os.path.join(PKG, "libzvdb_host.so")
CPP_TEST_SRC = os.path.join(ROOT, "tests", "cpp", "test_hnsw_host.cpp")
CPP_TEST_BIN = os.path.join(ROOT, "tests", "cpp", "test_hnsw_host")
C_TEST_SRC = os.path.join(ROOT, "tests", "c", "abi_smoke.c")
C_TEST_BIN = os.path.join(ROOT, "tests", "c", "abi_smoke")


def _newer(srcs, out) -> bool:
    if not os.path.exists(out):
        return True
    t = os.path.getmtime(out)
    return any(os.path.exists(s) and os.path.getmtime(s) > t for s in srcs)


def build_hip(force: bool = False) -> str:
    if force:
        subprocess.check_call(["make", "-C", CSRC, "clean"])
    subprocess.check_call(["make", "-C", CSRC, "-j4"])
    return LIB_HIP


def build_host(force: bool = False) -> str:
    srcs = [os.path.join(PKG, "host", "host_capi.cpp"), os.path.join(PKG, "host", "hnsw.hpp"),
            os.path.join(ROOT, "include", "zvdb_hip.h"), LIB_HIP]
    if force or _newer(srcs, LIB_HOST):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-o", LIB_HOST,
                               srcs[0], "-L" + PKG, "-lzvdb_hip", "-Wl,-rpath,$ORIGIN"])
    return LIB_HOST


def build_cpp_tests(force: bool = False) -> str:
    srcs = [CPP_TEST_SRC, os.path.join(PKG, "host", "hnsw.hpp"), LIB_HIP]
    if os.path.exists(CPP_TEST_SRC) and (force or _newer(srcs, CPP_TEST_BIN)):
        subprocess.check_call(["g++", "-O2", "-std=c++17", "-pthread", "-Wall", "-o", CPP_TEST_BIN,
                               CPP_TEST_SRC, "-L" + PKG, "-lzvdb_hip",
                               "-Wl,-rpath," + PKG])
    return CPP_TEST_BIN


def build_c_smoke(force: bool = False) -> str:
    """The pure-C99 caller of include/zvdb_hip.h (stand-in for Zig's @cImport: no zig toolchain in this image)."""
    srcs = [C_TEST_SRC, os.path.join(ROOT, "include", "zvdb_hip.h"), LIB_HIP]
    if os.path.exists(C_TEST_SRC) and (force or _newer(srcs, C_TEST_BIN)):
        subprocess.check_call(["gcc", "-std=c99", "-pedantic", "-Wall", "-Wextra", "-Werror", "-O1", "-I" + os.path.join(ROOT, "include"),
                               "-o", C_TEST_BIN, C_TEST_SRC, "-L" + PKG, "-lzvdb_hip", "-lm", "-Wl,-rpath," + PKG])
    return C_TEST_BIN


def build_all(force: bool = False) -> None:
    build_hip(force)
    build_host(force)
    build_cpp_tests(force)
    build_c_smoke(force)
