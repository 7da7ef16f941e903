# Runs ON the GPU box (through gpurun): round 6's closing evidence in one call, everything into gpurun_out/<tag>_* (copied to profiles/ afterwards).
#   GPU suite; default bench of every workload; rocprofv3 kernel tables of NS (default command AND a command whose launches are all full sub-batches),
#   C2, C3, C4; PMC passes (each in its own run, nothing else traced): HBM traffic of the dominant kernel of NS / C2 / C3 / C4 / C5 / SORT, SQ counters of the
#   north-star assignment kernel, matrix-core busy cycles of embed_kernel; BaseTracker::update on T objects / T threads; the f3 trackers.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-r06}; export TMPDIR=/tmp MOT_EVIDENCE_TAG=$TAG
Q="--no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 0 --parity-streams 0"
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_gputests.txt 2>&1
# ---- counters first: the bench lines below then quote counter files collected on these very sources ----
bash tools/collect_profiles.sh $TAG NS "--steps 20 --warmup 5 --no-cpu-baseline" "--streams 2048 --pipeline 1 --steps 2 --warmup 5 $Q" > $OUT/${TAG}_collect_NS.log 2>&1
cp $OUT/${TAG}_pmc_fetch_NS.json $OUT/${TAG}_pmc_write_NS.json $OUT/${TAG}_pmc_sq1_NS.json $OUT/${TAG}_pmc_sq2_NS.json profiles/ 2>/dev/null
python tools/pmc_derive.py $TAG NS 2048 "bench.py --workload NS --streams 2048 --pipeline 1 --steps 2 --warmup 5 $Q" > $OUT/${TAG}_pmc_derive_NS.log 2>&1
cp profiles/pmc_NS.json profiles/${TAG}_pmc_sq_lap.json $OUT/ 2>/dev/null
for spec in "C2 lap1_sparse 4096" "C3 feat 512" "C4 lap 256" "C5 lap1_sparse 2048" "SORT lap 4096"; do
  set -- $spec
  bash tools/pmc_traffic.sh $TAG $1 $2 $3 > $OUT/${TAG}_pmc_traffic_$1.log 2>&1
  cp $OUT/pmc_$1.json profiles/pmc_$1.json 2>/dev/null
done
bash tools/pmc_mfma_embed.sh $TAG > $OUT/${TAG}_pmc_mfma.log 2>&1
cp $OUT/${TAG}_pmc_mfma_embed.json profiles/ 2>/dev/null
# ---- the default command of every workload ----
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default_NS.json 2> $OUT/${TAG}_ns.err
for WL in C2 C3 C4 C5 SORT; do timeout 900 python bench.py --workload $WL > $OUT/${TAG}_bench_default_$WL.json 2> $OUT/${TAG}_$WL.err; done
timeout 300 python tools/kf_update_microbench.py 1024 2048 > $OUT/${TAG}_kf_update_microbench.json 2>/dev/null
# ---- kernel tables ----
for WL in NS_timed_only C2 C3 C4; do
  rm -rf /tmp/kt_$WL
  ARGS="--workload $WL --no-cpu-baseline"; [ $WL = NS_timed_only ] && ARGS="--steps 20 --warmup 5 $Q"
  ( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$WL -- python $ROOT/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof_$WL.json 2> $OUT/${TAG}_kt_$WL.err )
  ( python tools/rocpd_top_kernels.py /tmp/kt_$WL $OUT/${TAG}_kernel_stats_$WL.csv > $OUT/${TAG}_kernel_stats_$WL.txt 2>&1 )
  rm -rf /tmp/kt_$WL
done
# events against rocprofv3 on the SAME launches: one sub-batch, no leg behind the timed region -> the trace's last 20 launches are the timed ones
rm -rf /tmp/kt_one
( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_one -- python $ROOT/bench.py --steps 20 --warmup 5 --streams 6144 --pipeline 1 --no-outputs-resident $Q > $OUT/${TAG}_bench_under_rocprof_NS_one_subbatch.json 2> $OUT/${TAG}_kt_one.err )
python tools/rocpd_top_kernels.py /tmp/kt_one $OUT/${TAG}_kernel_stats_NS_one_subbatch.csv > $OUT/${TAG}_kernel_stats_NS_one_subbatch.txt 2>&1
python tools/last_launches_avg.py /tmp/kt_one "lap_sparse_kernel<true, 3, 256>" 20 > $OUT/${TAG}_NS_one_subbatch_last20.txt 2>&1
python - <<PY >> $OUT/${TAG}_NS_one_subbatch_last20.txt
import json
d = json.loads(open("$OUT/${TAG}_bench_under_rocprof_NS_one_subbatch.json").read().strip().splitlines()[-1])
print("HIP events of the same run (roofline.avg_launch_ms, the timed 20 launches): %.1f us" % (d["roofline"]["avg_launch_ms"] * 1e3))
PY
rm -rf /tmp/kt_one
python tools/kernel_duration_tail.py /tmp/kt_$TAG > $OUT/${TAG}_kernel_duration_tail_NS.txt 2>&1   # (the default NS command's trace, left there by collect_profiles.sh)
# ---- the plugin surface ----
timeout 600 python tools/bench_pooled.py NS 1 16 64 256 1024 > $OUT/${TAG}_basetracker_update_NS.json 2> $OUT/${TAG}_pooled.err
timeout 600 python tools/bench_pooled.py C2 1 16 64 256 1024 > $OUT/${TAG}_basetracker_update_C2.json 2>> $OUT/${TAG}_pooled.err
timeout 600 python tools/bench_f3.py 64 > $OUT/${TAG}_f3_streams_and_threads.json 2> $OUT/${TAG}_f3.err
python tools/kernel_sources_hash.py > $OUT/${TAG}_kernel_sources_sha.txt
for f in $OUT/${TAG}_bench_default_*.json; do echo $f; cut -c100-175 $f; done
grep -a "passed\|failed" $OUT/${TAG}_gputests.txt | tail -2
