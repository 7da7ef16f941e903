"""Builds the host-emulation harness (test infrastructure, not product)."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
OUT = os.path.join(ROOT, "tests", "_build")


def build_lap_emu(tie_runs=False):
    """tie_runs: lower lap_core.hpp's thresholds so that problems of test size take the closed-form tie runs of the replay
    (True: held in registers; "mem": one record per lane in registers, so that longer runs take the memory-staged variant)."""
    os.makedirs(OUT, exist_ok=True)
    so = os.path.join(OUT, {False: "liblapemu.so", True: "liblapemu_tie.so", "mem": "liblapemu_tiemem.so"}[tie_runs])
    flags = {False: [], True: ["-DMOT_LAP_TIE_MIN=2", "-DMOT_LAP_TIE_PER=24"], "mem": ["-DMOT_LAP_TIE_MIN=2", "-DMOT_LAP_TIE_PER=1"]}[tie_runs]
    srcs = [os.path.join(ROOT, "tests", "emu", f) for f in ("lap_emu.cpp", "emu_group.hpp")] + \
           [os.path.join(ROOT, "motcpp_amd", "csrc", f) for f in ("lap_core.hpp", "lap_cost.hpp", "lap_sparse.hpp", "cost_math.hpp", "grp.hpp", "mem.hpp")]
    if not os.path.exists(so) or any(os.path.getmtime(s) > os.path.getmtime(so) for s in srcs):
        subprocess.check_call(["g++", "-std=c++17", "-O1", "-fPIC", "-shared", "-pthread", "-ffp-contract=off"] + flags +
                              ["-o", so, srcs[0]])
    return so
