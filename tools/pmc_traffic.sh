#!/bin/bash
# Runs ON the GPU box (through gpurun): HBM traffic of the dominant kernel family of one workload from the PMC counters — FETCH_SIZE and WRITE_SIZE in
# SEPARATE rocprofv3 passes (nothing else traced), summed per kernel by tools/pmc_aggregate.py, turned into profiles-style pmc_<WL>.json by
# tools/pmc_traffic_derive.py.   tools/pmc_traffic.sh <tag> <workload> <family: cosine | lap | lap1_sparse | feat> <streams> "<extra bench args>"
set -u
TAG=$1; WL=$2; FAM=$3; S=$4; EXTRA=${5:-}
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
ARGS="--workload $WL --streams $S --pipeline 1 --no-cpu-baseline --long-run-steps 0 --sweep-streams= --host-input-steps 0 --isolated-steps 0 --parity-streams 0 $EXTRA"
for C in FETCH_SIZE WRITE_SIZE; do
  N=$( [ $C = FETCH_SIZE ] && echo fetch || echo write )
  rm -rf /tmp/pmc_$TAG
  ( cd /tmp && timeout 900 rocprofv3 --pmc $C --output-format csv -d /tmp/pmc_$TAG -- python $ROOT/bench.py $ARGS > $OUT/${TAG}_pmc_${N}_bench_$WL.json 2> $OUT/${TAG}_pmc_${N}_$WL.err )
  python tools/pmc_aggregate.py /tmp/pmc_$TAG $OUT/${TAG}_pmc_${N}_$WL.json > /dev/null 2>> $OUT/${TAG}_pmc_${N}_$WL.err
done
python tools/pmc_traffic_derive.py $OUT/${TAG}_pmc_fetch_$WL.json $OUT/${TAG}_pmc_write_$WL.json $WL $FAM $S "bench.py $ARGS" > $OUT/pmc_$WL.json
cat $OUT/pmc_$WL.json | head -c 600; echo
