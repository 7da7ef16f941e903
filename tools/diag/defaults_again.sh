# the default command of every workload once more, on whatever box this lands on (a second sample next to the closing evidence run)
cd /root/repo; mkdir -p gpurun_out; TAG=${1:-r06f}
timeout 600 python bench.py > gpurun_out/${TAG}_bench_default_NS.json 2>/dev/null
for WL in C2 C3 C4 C5 SORT; do timeout 900 python bench.py --workload $WL > gpurun_out/${TAG}_bench_default_$WL.json 2>/dev/null; done
python - <<P
import json
for w in ["NS","C2","C3","C4","C5","SORT"]:
    j=json.loads(open(f"gpurun_out/${TAG}_bench_default_{w}.json").read().strip().splitlines()[-1])
    print(w, round(j["value"]), "steps", j["steps"], "roofline", round(j["roofline"]["frac"],4), "stale", (j["roofline"].get("traffic_source") or {}).get("stale"), "parity", j["parity"]["mismatching_stream_frames"], "/", j["parity"]["stream_frames_checked"])
P
