# Runs ON the GPU box (through gpurun): the default bench of every workload plus rocprofv3 kernel traces of NS and C4, into gpurun_out/r02y_*.
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=r02y
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default_NS.json 2> $OUT/${TAG}_ns.err
for WL in C2 C3 C4 C5 SORT; do timeout 700 python bench.py --workload $WL > $OUT/${TAG}_bench_default_$WL.json 2> $OUT/${TAG}_$WL.err; done
export TMPDIR=/tmp; cd /tmp
for WL in NS C4; do
rm -rf /tmp/kt_$WL
ARGS="--no-cpu-baseline"; [ $WL = NS ] && ARGS="--steps 20 --warmup 5 --no-cpu-baseline"
( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$WL -- python bench.py --workload $WL $ARGS > $OUT/${TAG}_bench_under_rocprof_$WL.json 2> $OUT/${TAG}_kt_$WL.err )
( cd $ROOT && python tools/rocpd_top_kernels.py /tmp/kt_$WL $OUT/${TAG}_kernel_stats_$WL.csv > $OUT/${TAG}_kernel_stats_$WL.txt 2>&1 )
done
cd $ROOT; for f in $OUT/${TAG}_bench_default_*.json; do echo $f; cut -c100-175 $f; done
