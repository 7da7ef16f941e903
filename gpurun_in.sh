export TMPDIR=/tmp
R=$PWD
mkdir -p $R/gpurun_out /tmp/pmcout
cd /tmp
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d /tmp/pmcout -o f1 -- python $R/bench.py --workload C2 --steps 5 --warmup 40 --streams 2048 --no-cpu-baseline > /tmp/pmcout/b1.json 2> /tmp/pmcout/b1.err
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d /tmp/pmcout -o w1 -- python $R/bench.py --workload C2 --steps 5 --warmup 40 --streams 2048 --no-cpu-baseline > /tmp/pmcout/b2.json 2> /tmp/pmcout/b2.err
cd $R; python gpurun_pmc_agg.py
