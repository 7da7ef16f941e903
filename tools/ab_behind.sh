B="--steps 20 --warmup 5 --no-cpu-baseline --sweep-streams= --host-input-steps 0 --long-run-steps 600"
for rep in 1 2; do
MOT_LAP_BEHIND_FULL=0 python bench.py $B > gpurun_out/r04w_lean_$rep.json 2>/dev/null
MOT_LAP_BEHIND_FULL=1 python bench.py $B > gpurun_out/r04w_full_$rep.json 2>/dev/null
done
MOT_LAP_BEHIND_FULL=0 MOT_LAP_BEHIND_PRIO=0 python bench.py $B > gpurun_out/r04w_leannoprio_1.json 2>/dev/null
python - <<'P'
import json, glob
for f in sorted(glob.glob("gpurun_out/r04w_*.json")):
    d = json.load(open(f)); b = d["lap_behind_fast_path"]; l = d["long_run"]; n = max(1, b["problems"])
    s = b["sum"]
    print(f[11:], "problems", b["problems"], "avg Mcyc: p1", round(s["cyc_phase1_columns"]/n/1e6,2), "tr", round(s["cyc_phase1_transfer"]/n/1e6,2), "rr", round(s["cyc_row_reduction"]/n/1e6,2), "aug", round(s["cyc_augmentation"]/n/1e6,2),
          "slowest", round(b["slowest_cycles"]/1e6,1), "| p99", round(l["step_ms_p99"],2), "max", round(l["step_ms_max"],2), "median", round(l["step_ms_median"],2), "value", round(l["value"]))
P
