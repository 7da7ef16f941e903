"""The C-ABI libraries load without a GPU and export every symbol their headers declare (no compute calls here)."""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def libs():
    from motcpp_amd import _lib
    if not (os.path.exists(_lib.HIP_LIB) and os.path.exists(_lib.HOST_LIB)):
        _lib.build()
    return ctypes.CDLL(_lib.HIP_LIB, mode=ctypes.RTLD_GLOBAL), ctypes.CDLL(_lib.HOST_LIB)


def declared(header, prefix):
    src = open(os.path.join(ROOT, "include", header)).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(%s[a-z0-9_]+)\s*\(" % prefix, src)))


def test_hip_library_exports_the_c_abi(libs):
    names = declared("motcpp_amd.h", "mot_")
    assert len(names) >= 30
    for n in names:
        assert hasattr(libs[0], n), n


def test_host_library_exports_the_handles(libs):
    names = declared("motcpp_c.h", "motcpp_")
    assert len(names) >= 15
    for n in names:
        assert hasattr(libs[1], n), n


def test_no_cpu_fallback_without_a_device(libs):
    # in a container without a GPU creating a context must fail loudly, never fall back
    import torch
    if torch.cuda.is_available():
        pytest.skip("GPU present")
    ctx = ctypes.c_void_p()
    rc = libs[0].mot_ctx_create(0, None, ctypes.byref(ctx))
    assert rc != 0 and not ctx.value
    from motcpp_amd import _lib
    with pytest.raises(_lib.MotError):
        _lib.Context(0)
    with pytest.raises(_lib.MotError):
        _lib.Tracker("bytetrack")


def test_the_tools_mirror_of_mot_kf_task_has_the_struct_s_size(tmp_path):
    """tools/kf_update_microbench.py builds arrays of mot_kf_task with ctypes: a field appended to the struct (round 6: mean_dense, cov_blocks,
    dense_flag, meas4) and not to the mirror makes the kernels read every task after the first from the wrong offset — the microbench aborted that way
    once. sizeof from the header, compiled by gcc, against the mirror."""
    import importlib.util
    import subprocess
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    src = tmp_path / "sz.c"
    src.write_text('#include <stdio.h>\n#include "motcpp_amd.h"\nint main(void) { printf("%zu\\n", sizeof(mot_kf_task)); return 0; }\n')
    exe = tmp_path / "sz"
    subprocess.check_call(["gcc", "-I", os.path.join(root, "include"), "-o", str(exe), str(src)])
    want = int(subprocess.check_output([str(exe)]).decode().strip())
    text = open(os.path.join(root, "tools", "kf_update_microbench.py")).read()
    ns = {}
    start = text.index("class KfTask")
    end = text.index("\n\n\n", start)
    exec("import ctypes as C\n" + text[start:end], ns)  # (the class only: importing the tool would load the library and look for a GPU)
    assert ctypes.sizeof(ns["KfTask"]) == want
