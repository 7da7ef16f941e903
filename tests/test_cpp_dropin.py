"""C++ drop-in surface: compiles tests/cpp/test_dropin.cpp (a mirror of the reference's gtest files) against
include/motcpp/ with plain g++ on CPU; runs the binary on the GPU box."""
import os
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
BIN = os.path.join(ROOT, "tests", "_build", "test_dropin")


def build():
    from motcpp_amd import _lib
    if not (os.path.exists(_lib.HIP_LIB) and os.path.exists(_lib.HOST_LIB)):
        _lib.build()
    os.makedirs(os.path.dirname(BIN), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_dropin.cpp")
    if not os.path.exists(BIN) or os.path.getmtime(src) > os.path.getmtime(BIN) or os.path.getmtime(_lib.HOST_LIB) > os.path.getmtime(BIN):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-I", os.path.join(ROOT, "include"), src, "-o", BIN,
                               "-L", _lib.LIBDIR, "-lmotcpp", "-lmotcpp_hip", "-pthread", "-Wl,-rpath," + _lib.LIBDIR])
    return BIN


def test_user_code_compiles_against_the_dropin_headers():
    assert os.path.exists(build())


@pytest.mark.gpu
def test_dropin_behaviour_on_gpu():
    out = subprocess.run([build()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "drop-in ok" in out.stdout


CBIN = os.path.join(ROOT, "tests", "_build", "test_capi_batched")


def build_c():
    from motcpp_amd import _lib
    if not os.path.exists(_lib.HIP_LIB):
        _lib.build()
    os.makedirs(os.path.dirname(CBIN), exist_ok=True)
    src = os.path.join(ROOT, "tests", "cpp", "test_capi_batched.c")
    if not os.path.exists(CBIN) or os.path.getmtime(src) > os.path.getmtime(CBIN) or os.path.getmtime(_lib.HIP_LIB) > os.path.getmtime(CBIN):
        subprocess.check_call(["gcc", "-std=c11", "-O1", "-Wall", "-I", os.path.join(ROOT, "include"), src, "-o", CBIN,
                               "-L", _lib.LIBDIR, "-lmotcpp_hip", "-Wl,-rpath," + _lib.LIBDIR])
    return CBIN


def test_c_consumer_of_the_batched_abi_compiles_as_c():
    assert os.path.exists(build_c())


@pytest.mark.gpu
def test_batched_c_abi_and_launch_flags_on_gpu():
    out = subprocess.run([build_c()], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "batched C ABI ok" in out.stdout
