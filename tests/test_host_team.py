"""Blocking host worker team (motcpp_amd/csrc/host/team.*): pure C++, no GPU and no HIP library involved."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_team_semantics():
    out = os.path.join(ROOT, "tests", "_build", "test_team")
    os.makedirs(os.path.dirname(out), exist_ok=True)
    host = os.path.join(ROOT, "motcpp_amd", "csrc", "host")
    subprocess.check_call(["g++", "-std=c++17", "-O1", "-pthread", "-I", host, os.path.join(ROOT, "tests", "cpp", "test_team.cpp"),
                           os.path.join(host, "team.cpp"), "-o", out])
    r = subprocess.run([out], capture_output=True, text=True, timeout=120)
    assert r.returncode == 0 and "team ok" in r.stdout, r.stdout + r.stderr
