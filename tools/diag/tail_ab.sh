cd /root/repo
summ='import json,sys; j=json.loads(sys.stdin.read().strip().splitlines()[-1])["basetracker_update"]; print({k:(round(v["frames_per_s"]), round(v["latency_ms_p50"],2), round(v["latency_ms_p99"],2), round(v["latency_ms_max"],1)) for k,v in j.items()})'
for i in 1 2; do
echo "library first (no torch):"; timeout 300 python tools/bench_pooled.py NS 16 64 256 1024 2>/dev/null | python -c "$summ"
echo "torch first:"; timeout 300 python -c "
import torch, sys, runpy
torch.zeros(1).cuda()
sys.argv=['bench_pooled.py','NS','16','64','256','1024']
runpy.run_path('tools/bench_pooled.py', run_name='__main__')" 2>/dev/null | python -c "$summ"
done
