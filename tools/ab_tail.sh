# Runs ON the GPU box: A/B of the exact kernel behind the fast path (issue priority, full LDS state) on the north-star workload and C3,
# same box, one after the other.   tools/ab_tail.sh [tag]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-r04t}
timeout 900 python -m pytest tests -m gpu -x -q -k "lap or bytetrack or device_lifecycle or sort" > $OUT/${TAG}_gputests_subset.txt 2>&1; tail -3 $OUT/${TAG}_gputests_subset.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --sweep-streams= --host-input-steps 0 --long-run-steps 600"
for rep in 1 2; do
MOT_LAP_BEHIND_PRIO=0 MOT_LAP_BEHIND_FULL=0 timeout 400 python bench.py $B > $OUT/${TAG}_NS_old_$rep.json 2> $OUT/${TAG}_err.txt
MOT_LAP_BEHIND_PRIO=1 MOT_LAP_BEHIND_FULL=0 timeout 400 python bench.py $B > $OUT/${TAG}_NS_prio_$rep.json 2>> $OUT/${TAG}_err.txt
MOT_LAP_BEHIND_PRIO=0 MOT_LAP_BEHIND_FULL=1 timeout 400 python bench.py $B > $OUT/${TAG}_NS_full_$rep.json 2>> $OUT/${TAG}_err.txt
timeout 400 python bench.py $B > $OUT/${TAG}_NS_new_$rep.json 2>> $OUT/${TAG}_err.txt
done
for P in 4 6; do timeout 400 python bench.py $B --pipeline $P > $OUT/${TAG}_NS_new_pipe$P.json 2>> $OUT/${TAG}_err.txt; done
MOT_LAP_BEHIND_PRIO=0 timeout 400 python bench.py --workload C3 $B > $OUT/${TAG}_C3_old.json 2>> $OUT/${TAG}_err.txt
timeout 400 python bench.py --workload C3 $B > $OUT/${TAG}_C3_new.json 2>> $OUT/${TAG}_err.txt
python - <<'P'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ.get("TAG", "r04t") + "_*.json"))):
    try:
        d = json.load(open(f)); l = d["long_run"]
        print(os.path.basename(f), round(d["value"]), round(l["value"]), "median", round(l["step_ms_median"], 2), "p99", round(l["step_ms_p99"], 2), "max", round(l["step_ms_max"], 2), "over", l["steps_over_1.5x_median"], d["lap_fast_path"]["not_unique"])
    except Exception as e:
        print(f, "ERR", e)
P
