# Runs ON the GPU box (through gpurun): the default bench of every workload, rocprofv3 kernel traces of NS and C4 and the PMC passes of NS,
# into gpurun_out/<tag>_*.   tools/final_evidence.sh [tag]
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-r04z}
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default_NS.json 2> $OUT/${TAG}_ns.err
for WL in C2 C3 C4 C5 SORT; do timeout 700 python bench.py --workload $WL > $OUT/${TAG}_bench_default_$WL.json 2> $OUT/${TAG}_$WL.err; done
timeout 300 python tools/kf_update_microbench.py 1024 2048 > $OUT/${TAG}_kf_update_microbench.json 2>/dev/null
MOT_KFMB_LIKE_TRACKER=1 timeout 300 python tools/kf_update_microbench.py 1024 2048 > $OUT/${TAG}_kf_update_microbench_tracker_like.json 2>/dev/null
export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/kt_C4
( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_C4 -- python bench.py --workload C4 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof_C4.json 2> $OUT/${TAG}_kt_C4.err )
( cd $ROOT && python tools/rocpd_top_kernels.py /tmp/kt_C4 $OUT/${TAG}_kernel_stats_C4.csv > $OUT/${TAG}_kernel_stats_C4.txt 2>&1 )
cd $ROOT
bash tools/collect_profiles.sh $TAG NS "--steps 20 --warmup 5 --no-cpu-baseline" "--steps 20 --warmup 5 --no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 0" > $OUT/${TAG}_collect.log 2>&1
( python tools/kernel_duration_tail.py /tmp/kt_$TAG > $OUT/${TAG}_kernel_duration_tail_NS.txt 2>&1 )
for WL in C2 C3; do
  rm -rf /tmp/kt_$WL
  ( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$WL -- python $ROOT/bench.py --workload $WL --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof_$WL.json 2> $OUT/${TAG}_kt_$WL.err )
  ( python tools/rocpd_top_kernels.py /tmp/kt_$WL $OUT/${TAG}_kernel_stats_$WL.csv > $OUT/${TAG}_kernel_stats_$WL.txt 2>&1 )
done
timeout 300 python tools/bench_pooled.py NS 1 16 64 256 > $OUT/${TAG}_basetracker_update_NS.json 2> $OUT/${TAG}_pooled.err
timeout 300 python tools/bench_pooled.py C2 1 16 64 256 > $OUT/${TAG}_basetracker_update_C2.json 2>> $OUT/${TAG}_pooled.err
timeout 1500 python -m pytest tests -m gpu -q > $OUT/${TAG}_gputests.txt 2>&1
for f in $OUT/${TAG}_bench_default_*.json; do echo $f; cut -c100-175 $f; done
