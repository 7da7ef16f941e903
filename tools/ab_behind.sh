# Runs ON the GPU box: the exact kernel behind the fast path, A/B of its launch options (MOT_LAP_BEHIND_FULL / _ALL / _PRIO) on the same
# box: cycles per declined problem by phase (mot_lap_behind_stats) and the long run's step-time tail.   tools/ab_behind.sh [tag]
TAG=${1:-r04w}
B="--steps 20 --warmup 5 --no-cpu-baseline --sweep-streams= --host-input-steps 0 --long-run-steps 600"
for rep in 1 2; do
MOT_LAP_BEHIND_FULL=0 python bench.py $B > gpurun_out/${TAG}_lean_$rep.json 2>/dev/null
MOT_LAP_BEHIND_ALL=0 python bench.py $B > gpurun_out/${TAG}_full_$rep.json 2>/dev/null
python bench.py $B > gpurun_out/${TAG}_all_$rep.json 2>/dev/null
done
MOT_LAP_BEHIND_ALL=0 python bench.py --workload C3 $B > gpurun_out/${TAG}_C3_lean.json 2>/dev/null
python bench.py --workload C3 $B > gpurun_out/${TAG}_C3_all.json 2>/dev/null
TAG=$TAG python - <<'P'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/" + os.environ["TAG"] + "_*.json")):
    d = json.load(open(f)); b = d["lap_behind_fast_path"]; l = d["long_run"]; n = max(1, b["problems"])
    s = b["sum"]
    print(os.path.basename(f), "problems", b["problems"], "avg Mcyc: p1", round(s["cyc_phase1_columns"]/n/1e6,2), "tr", round(s["cyc_phase1_transfer"]/n/1e6,2), "rr", round(s["cyc_row_reduction"]/n/1e6,2), "aug", round(s["cyc_augmentation"]/n/1e6,2),
          "slowest", round(b["slowest_cycles"]/1e6,1), "| p99", round(l["step_ms_p99"],2), "max", round(l["step_ms_max"],2), "median", round(l["step_ms_median"],2), "value", round(l["value"]), "mismatch-free" if d.get("parity") is None else d["parity"]["mismatching_stream_frames"])
P
