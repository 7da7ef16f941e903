# Runs ON the GPU box (through gpurun): the default bench of every workload, rocprofv3 kernel traces of NS, C2, C3 and C4, the pooled / threaded legs
# and the GPU test suite, into gpurun_out/<tag>_*.   tools/final_evidence.sh [tag]      (PMC passes: tools/collect_profiles.sh, tools/pmc_traffic.sh)
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-r05}
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default_NS.json 2> $OUT/${TAG}_ns.err
for WL in C2 C3 C4 C5 SORT; do timeout 900 python bench.py --workload $WL > $OUT/${TAG}_bench_default_$WL.json 2> $OUT/${TAG}_$WL.err; done
timeout 300 python tools/kf_update_microbench.py 1024 2048 > $OUT/${TAG}_kf_update_microbench.json 2>/dev/null
export TMPDIR=/tmp
for WL in NS C2 C3 C4; do
  rm -rf /tmp/kt_$WL
  ARGS="--workload $WL --no-cpu-baseline"; [ $WL = NS ] && ARGS="--steps 20 --warmup 5 --no-cpu-baseline"
  ( cd /tmp && timeout 1200 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$WL -- python $ROOT/bench.py $ARGS > $OUT/${TAG}_bench_under_rocprof_$WL.json 2> $OUT/${TAG}_kt_$WL.err )
  ( python tools/rocpd_top_kernels.py /tmp/kt_$WL $OUT/${TAG}_kernel_stats_$WL.csv > $OUT/${TAG}_kernel_stats_$WL.txt 2>&1 )
  [ $WL = NS ] && ( python tools/kernel_duration_tail.py /tmp/kt_NS > $OUT/${TAG}_kernel_duration_tail_NS.txt 2>&1 )
  rm -rf /tmp/kt_$WL
done
timeout 600 python tools/bench_pooled.py NS 1 16 64 256 1024 > $OUT/${TAG}_basetracker_update_NS.json 2> $OUT/${TAG}_pooled.err
timeout 600 python tools/bench_pooled.py C2 1 16 64 256 1024 > $OUT/${TAG}_basetracker_update_C2.json 2>> $OUT/${TAG}_pooled.err
timeout 600 python tools/bench_f3.py 64 > $OUT/${TAG}_f3_streams_and_threads.json 2> $OUT/${TAG}_f3.err
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_gputests.txt 2>&1
for f in $OUT/${TAG}_bench_default_*.json; do echo $f; cut -c100-175 $f; done
tail -2 $OUT/${TAG}_gputests.txt
