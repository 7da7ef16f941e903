export TMPDIR=/tmp
mkdir -p gpurun_out/prof_C2 gpurun_out/prof_NS
R=$PWD
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_C2 -o c2 -- python $R/bench.py --workload C2 --steps 30 --warmup 40 --streams 2048 --no-cpu-baseline > $R/gpurun_out/prof_C2/bench.json 2> $R/gpurun_out/prof_C2/bench.err
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/prof_NS -o ns -- python $R/bench.py --workload NS --steps 15 --warmup 30 --streams 256 --no-cpu-baseline > $R/gpurun_out/prof_NS/bench.json 2> $R/gpurun_out/prof_NS/bench.err
cd $R
find gpurun_out/prof_C2 gpurun_out/prof_NS -type f | head -30
ls -la gpurun_out/prof_C2/* | head
