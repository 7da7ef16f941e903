timeout 900 python -m pytest tests -m gpu -x -q > gpurun_out/r04s_gputests.txt 2>&1; grep -E "passed|failed" gpurun_out/r04s_gputests.txt | tail -1
B="--steps 20 --warmup 5 --no-cpu-baseline --sweep-streams= --host-input-steps 0 --long-run-steps 600"
for rep in 1 2; do python bench.py $B > gpurun_out/r04s_all_$rep.json 2>/dev/null; done
python bench.py --workload C3 $B > gpurun_out/r04s_C3_all.json 2>/dev/null
python bench.py --workload C5 $B > gpurun_out/r04s_C5_all.json 2>/dev/null
TAG=r04s python - <<'P'
import json, glob, os
for f in sorted(glob.glob("gpurun_out/" + os.environ["TAG"] + "_*.json")):
    d = json.load(open(f)); b = d["lap_behind_fast_path"]; l = d["long_run"]; n = max(1, b["problems"])
    s = b["sum"]
    print(os.path.basename(f), "problems", b["problems"], "avg Mcyc: p1", round(s["cyc_phase1_columns"]/n/1e6,2), "tr", round(s["cyc_phase1_transfer"]/n/1e6,2), "rr", round(s["cyc_row_reduction"]/n/1e6,2), "aug", round(s["cyc_augmentation"]/n/1e6,2),
          "slowest", round(b["slowest_cycles"]/1e6,1), "| p99", round(l["step_ms_p99"],2), "max", round(l["step_ms_max"],2), "median", round(l["step_ms_median"],2), "value", round(l["value"]))
P
