#!/bin/bash
# Runs ON the GPU box (through gpurun): matrix-core utilisation of embed_kernel<cosine> from the hardware counters — SQ_VALU_MFMA_BUSY_CYCLES and
# GRBM_GUI_ACTIVE in one rocprofv3 --pmc pass of tools/embed_microbench.py (nothing else traced), summed by tools/pmc_aggregate.py.
#   tools/pmc_mfma_embed.sh <tag>     -> gpurun_out/<tag>_pmc_mfma_embed.json  (copy to profiles/)
set -u
TAG=${1:-r06}; ROOT=$(pwd); OUT=$ROOT/gpurun_out; mkdir -p $OUT; export TMPDIR=/tmp
rm -rf /tmp/pmc_mfma
( cd /tmp && timeout 600 rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_mfma -- python $ROOT/tools/embed_microbench.py 512 > $OUT/${TAG}_pmc_mfma_embed_run.json 2> $OUT/${TAG}_pmc_mfma_embed.err )
python tools/pmc_aggregate.py /tmp/pmc_mfma $OUT/${TAG}_pmc_mfma_raw.json > /dev/null 2>> $OUT/${TAG}_pmc_mfma_embed.err
python - <<PY
import json, os, sys
sys.path.insert(0, "$ROOT/tools")
from kernel_sources_hash import kernel_sources_hash
raw = json.load(open("$OUT/${TAG}_pmc_mfma_raw.json"))
k = [n for n in raw if n.startswith("embed_kernel<")]
k = max(k, key=lambda n: raw[n]["SQ_VALU_MFMA_BUSY_CYCLES"]["sum"])
busy, act = raw[k]["SQ_VALU_MFMA_BUSY_CYCLES"], raw[k]["GRBM_GUI_ACTIVE"]
simds = 256 * 4
# GRBM_GUI_ACTIVE is reported once per XCD (8 instances summed by rocprofv3): cycles the kernel was active = sum / 8
frac = busy["sum"] / (act["sum"] / 8.0 * simds)
out = {"kernel_sources_sha": kernel_sources_hash(), "tag": "$TAG", "kernel": k, "dispatches": busy["dispatches"],
       "SQ_VALU_MFMA_BUSY_CYCLES_sum": busy["sum"], "GRBM_GUI_ACTIVE_sum": act["sum"], "simds": simds,
       "mfma_busy_frac": frac, "run": json.loads(open("$OUT/${TAG}_pmc_mfma_embed_run.json").read().strip().splitlines()[-1]),
       "_comment": "rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES GRBM_GUI_ACTIVE on tools/embed_microbench.py 512 (1024 x 512 x 256-d per task); busy cycles summed over the SIMDs / (active cycles x 1024 SIMDs); GRBM_GUI_ACTIVE summed over 8 XCDs"}
json.dump(out, open("$OUT/${TAG}_pmc_mfma_embed.json", "w"), indent=1)
print(json.dumps(out)[:600])
PY
