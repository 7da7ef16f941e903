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


def test_gpus_n_starts_n_ranks_itself():
    # `python bench.py --gpus 2` with no launcher around it: the command re-executes itself under torch.distributed.run, one process
    # per GPU, rank r on LOCAL_RANK r. MOT_BENCH_LAUNCH_PROBE=1 stops the ranks right after the rendezvous (gloo here: no GPU is
    # touched) and makes rank 0 report who is there.
    import json
    env = dict(os.environ, MOT_BENCH_LAUNCH_PROBE="1")
    for k in ("RANK", "WORLD_SIZE", "LOCAL_RANK", "MASTER_ADDR", "MASTER_PORT"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "1", "--warmup", "0"], capture_output=True,
                         text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.splitlines() if l.strip()]
    assert len(lines) == 1  # exactly one JSON line, from rank 0
    d = json.loads(lines[0])
    assert d["probe"] is True and d["world"] == 2
    assert sorted(r["rank"] for r in d["ranks"]) == [0, 1]
    assert sorted(r["local_rank"] for r in d["ranks"]) == [0, 1]
    assert len({r["pid"] for r in d["ranks"]}) == 2  # two processes
    # the per-rank rates of the real line travel the same way (one all-reduce over the job's process group): rank count and one rate per rank
    assert d["per_rank"]["ranks"] == 2 and d["per_rank"]["frames/s"] == [1000.0, 2000.0] and d["per_rank"]["backend"] == "gloo"
