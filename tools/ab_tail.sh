# Runs ON the GPU box: the exact kernel behind the fast path on the north-star workload (long run: step-time tail), A/B of its
# launch options on the same box, kernel-duration tail from a rocprofv3 trace.   tools/ab_tail.sh [tag]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-r04t}
timeout 1500 python -m pytest tests -m gpu -x -q > $OUT/${TAG}_gputests.txt 2>&1; tail -3 $OUT/${TAG}_gputests.txt
B="--steps 20 --warmup 5 --no-cpu-baseline --sweep-streams= --host-input-steps 0 --long-run-steps 600"
for rep in 1 2; do
MOT_LAP_BEHIND_FULL=0 timeout 400 python bench.py $B > $OUT/${TAG}_NS_lean_$rep.json 2> $OUT/${TAG}_err.txt
timeout 400 python bench.py $B > $OUT/${TAG}_NS_new_$rep.json 2>> $OUT/${TAG}_err.txt
done
timeout 400 python bench.py --workload C3 $B > $OUT/${TAG}_C3_new.json 2>> $OUT/${TAG}_err.txt
timeout 400 python bench.py --workload C5 $B > $OUT/${TAG}_C5_new.json 2>> $OUT/${TAG}_err.txt
export TMPDIR=/tmp; rm -rf /tmp/kt_$TAG
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline --sweep-streams= --host-input-steps 0 --long-run-steps 300 > $OUT/${TAG}_bench_under_rocprof_NS.json 2> $OUT/${TAG}_kt.err )
python tools/kernel_duration_tail.py /tmp/kt_$TAG > $OUT/${TAG}_kernel_duration_tail_NS.txt 2>&1
python tools/rocpd_top_kernels.py /tmp/kt_$TAG $OUT/${TAG}_kernel_stats_NS.csv > $OUT/${TAG}_kernel_stats_NS.txt 2>&1
head -8 $OUT/${TAG}_kernel_duration_tail_NS.txt | cut -c1-160
TAG=$TAG python - <<'P'
import json, glob, os
for f in sorted(glob.glob(os.path.join("gpurun_out", os.environ["TAG"] + "_*.json"))):
    try:
        d = json.load(open(f)); l = d["long_run"]
        print(os.path.basename(f), round(d["value"]), round(l["value"]), "median", round(l["step_ms_median"], 2), "p99", round(l["step_ms_p99"], 2), "max", round(l["step_ms_max"], 2), "over", l["steps_over_1.5x_median"], d["lap_fast_path"]["not_unique"])
    except Exception as e:
        print(f, "ERR", e)
P
