# Runs ON the GPU box: the north-star default bench line and its rocprofv3 kernel trace on the build in the tree.   tools/final_ns.sh <tag>
set -u
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; TAG=${1:-r04g}
timeout 600 python bench.py --steps 20 --warmup 5 > $OUT/${TAG}_bench_default_NS.json 2> $OUT/${TAG}_ns.err
export TMPDIR=/tmp; rm -rf /tmp/kt_$TAG
( cd /tmp && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python $ROOT/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/${TAG}_bench_under_rocprof_NS.json 2> $OUT/${TAG}_kt_NS.err )
python tools/rocpd_top_kernels.py /tmp/kt_$TAG $OUT/${TAG}_kernel_stats_NS.csv > $OUT/${TAG}_kernel_stats_NS.txt 2>&1
python tools/kernel_duration_tail.py /tmp/kt_$TAG > $OUT/${TAG}_kernel_duration_tail_NS.txt 2>&1
cut -c100-175 $OUT/${TAG}_bench_default_NS.json; head -5 $OUT/${TAG}_kernel_stats_NS.txt | cut -c1-130
