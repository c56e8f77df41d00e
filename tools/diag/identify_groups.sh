# 1-to-N identification: matches recorded per flush (--group) against throughput, 32 matches each
mkdir -p gpurun_out/identify_groups
for g in 2 4 8; do
  timeout -k 10 400 python bench.py --mode identify --matches 32 --group $g --steps 1 --warmup 0 --no-cpu-baseline --extras 0 > gpurun_out/identify_groups/g$g.json 2> gpurun_out/identify_groups/g$g.err || { tail -3 gpurun_out/identify_groups/g$g.err; exit 1; }
  python3 -c "
import json; j=json.loads(open('gpurun_out/identify_groups/g$g.json').read().strip().split('\n')[-1]); print('group $g: value', round(j['value']), 'gates/s, s per match', round(j['ms_per_step']/32/1e3,3), {k:v for k,v in j.items() if 'host' in k or 'record' in k})"
done
