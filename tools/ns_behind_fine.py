"""Fine-profile split of the exact kernel's serial row-reduction rounds on the problems the sparse solver declines in a north-star run
(a -DMOT_LAP_FINE_PROF build of lap_kernel_wide.hip linked as lib/libmotcpp_hip_fineprof.so). Runs bench.py's NS workload in-process and
prints mot_lap_behind_stats' twelve fine counters.  MOTCPP_HIP_LIB_DIAG=libmotcpp_hip_fineprof.so python tools/ns_behind_fine.py [streams] [long-run steps]"""
import json, os, subprocess, sys
os.environ.setdefault("MOTCPP_HIP_LIB_DIAG", "libmotcpp_hip_fineprof.so")
streams = sys.argv[1] if len(sys.argv) > 1 else "6144"
steps = sys.argv[2] if len(sys.argv) > 2 else "300"
env = dict(os.environ, MOT_BEHIND_RAW="1")
out = subprocess.run([sys.executable, "bench.py", "--streams", streams, "--no-cpu-baseline", "--sweep-streams=", "--host-input-steps", "0", "--parity-streams", "0",
                      "--long-run-steps", steps], env=env, capture_output=True, text=True)
line = out.stdout.strip().splitlines()[-1]
d = json.loads(line)
b = d["lap_behind_fast_path"]
raw = b.get("raw_fine")
n = max(1, b["problems"])
print("problems", b["problems"], "value", round(d["value"]), "avg Mcycles: p1", round(b["sum"]["cyc_phase1_columns"] / n / 1e6, 2), "transfer", round(b["sum"]["cyc_phase1_transfer"] / n / 1e6, 2),
      "row reduction", round(b["sum"]["cyc_row_reduction"] / n / 1e6, 2), "augmentation", round(b["sum"]["cyc_augmentation"] / n / 1e6, 2), "rounds", round(b["sum"]["serial_row_rounds"] / n))
if raw:
    names = ("evaluate", "row-min reduce", "top-2 reduce", "tail", "closed-form runs", "dummy rounds", "n real rounds", "n dummy rounds", "n runs")
    for k, nm in enumerate(names):
        print("  %-18s %12.1f per problem" % (nm, raw[k] / n))
    if raw[6]:
        print("  per real round: evaluate %.0f row-min %.0f top-2 %.0f tail %.0f cycles" % tuple(raw[k] / raw[6] for k in range(4)))
    if raw[7]: print("  per dummy round %.0f cycles" % (raw[5] / raw[7]))
    if raw[8]: print("  per closed-form run %.0f cycles" % (raw[4] / raw[8]))
