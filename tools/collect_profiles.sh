#!/bin/bash
# Runs ON the GPU box (through gpurun): rocprofv3 evidence for one bench.py workload. Everything lands in gpurun_out/<tag>_*.
#   tools/collect_profiles.sh <tag> <workload> "<bench args of the timed command>" "<bench args of the PMC passes>"
# Pass 1 = kernel trace + stats of the bench command as given; passes 2-5 = PMC counters, each in its own run (HBM bytes and SQ
# counters never share a pass; no other tracing is combined with --pmc).
set -u
TAG=$1; WL=$2; ARGS=$3; PMCARGS=$4
ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT
export TMPDIR=/tmp
cd /tmp
rm -rf /tmp/kt_$TAG
( cd $ROOT && timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/kt_$TAG -- python bench.py --workload $WL $ARGS > $OUT/${TAG}_bench_under_rocprof_$WL.json 2> $OUT/${TAG}_kt_$WL.err )
( cd $ROOT && python tools/rocpd_top_kernels.py /tmp/kt_$TAG $OUT/${TAG}_kernel_stats_$WL.csv > $OUT/${TAG}_kernel_stats_$WL.txt 2>&1 )
i=0
for CTRS in "FETCH_SIZE" "WRITE_SIZE" "SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY GRBM_GUI_ACTIVE" "SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_ACTIVE_INST_VALU"; do
  NAME=(fetch write sq1 sq2); N=${NAME[$i]}; i=$((i+1))
  rm -rf /tmp/pmc_$TAG
  ( cd $ROOT && timeout 900 rocprofv3 --pmc $CTRS --output-format csv -d /tmp/pmc_$TAG -- python bench.py --workload $WL $PMCARGS > $OUT/${TAG}_pmc_${N}_bench_$WL.json 2> $OUT/${TAG}_pmc_${N}_$WL.err )
  ( cd $ROOT && python tools/pmc_aggregate.py /tmp/pmc_$TAG $OUT/${TAG}_pmc_${N}_$WL.json > /dev/null 2>> $OUT/${TAG}_pmc_${N}_$WL.err )
done
ls -la $OUT | grep ${TAG}_
