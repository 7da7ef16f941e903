mkdir -p gpurun_out
python -m pytest tests -x -q -m gpu 2>&1 | tail -5
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
for S in 64 256 1024; do
  python bench.py --steps 40 --warmup 30 --streams $S --no-cpu-baseline > gpurun_out/bench_C2_S$S.json 2> gpurun_out/bench_C2_S$S.err; tail -c 1800 gpurun_out/bench_C2_S$S.json; echo; tail -3 gpurun_out/bench_C2_S$S.err
done
python bench.py --workload NS --steps 30 --warmup 30 --streams 128 --no-cpu-baseline > gpurun_out/bench_NS_S128.json 2> gpurun_out/bench_NS.err; tail -c 1800 gpurun_out/bench_NS_S128.json; tail -3 gpurun_out/bench_NS.err
nproc; lscpu | grep "Model name"
