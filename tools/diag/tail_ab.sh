cd /root/repo
MOTCPP_BENCH_SPIKES=1.0 timeout 300 python tools/bench_pooled.py NS 64 256 2>&1 | grep "bench_threads\]\|assignments" | python -c "
import sys,re,json
for l in sys.stdin:
    if l.startswith('[bench'): print(l.strip()[:300])
    elif l.startswith('{'):
        j=json.loads(l); print({k:(v['assignments'], round(v['latency_ms_p99'],2)) for k,v in j['basetracker_update'].items()})"
