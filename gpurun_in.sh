mkdir -p gpurun_out
python gpurun_lapprof.py
python -m pytest tests -x -q -m gpu 2>&1 | tail -3
for cfg in "NS 2048 32 2 12 35" "C2 4096 16 1 12 40"; do set -- $cfg
  timeout 600 python bench.py --workload $1 --steps $5 --warmup $6 --streams $2 --threads $3 --pipeline $4 --no-cpu-baseline > gpurun_out/b.json 2> gpurun_out/b.err; python -c "
import json;d=json.load(open('gpurun_out/b.json'));print('$1 S=$2 thr=$3 pipe=$4 fps',round(d['value'],1),'ms/step',round(d['ms_per_step'],2),'busy',round(d['gpu_busy_frac'],3),'host',{k:round(v,2) for k,v in d['host_ms_per_step'].items()},{k:round(v['ms_total']/v['launches'],3) for k,v in d['kernels'].items()})" || tail -5 gpurun_out/b.err
done
