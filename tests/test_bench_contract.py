"""bench.py's command-line contract, checked without a GPU: the documented flags exist, the defaults are the north-star workload,
a machine without an MI355X gets a loud refusal (no CPU fallback), and nothing but the JSON line can reach stdout."""
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_help_lists_the_contract_flags():
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--help"], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0
    for flag in ("--gpus", "--steps", "--warmup", "--workload", "--streams", "--gather", "--no-in-flight", "--no-cpu-baseline"):
        assert flag in out.stdout, flag
    assert "NS" in out.stdout and "north-star" in out.stdout


def test_refuses_loudly_without_a_gpu():
    import torch
    if torch.cuda.is_available():
        return  # (on the GPU box the benchmark itself is the test)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--steps", "1", "--warmup", "1"], capture_output=True, text=True, timeout=600)
    assert out.returncode != 0
    assert out.stdout.strip() == ""  # stdout is reserved for the JSON line
    assert "MI355X" in out.stderr or "gfx950" in out.stderr
