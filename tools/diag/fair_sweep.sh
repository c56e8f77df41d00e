# priority time slice of co-resident blind-rotate workgroups (2^k shader cycles; 0 = off) against the time of a whole
# match -- the match runs at the warm clock, the microbenchmarks this was first tuned on do not
mkdir -p gpurun_out/fair
for k in 18 0 12 14 16 20 22 18; do
  TFHE_HIP_BR_FAIR=$k timeout -k 10 200 python bench.py --steps 2 --warmup 1 --extras 0 --no-cpu-baseline > gpurun_out/fair/k$k.json 2> gpurun_out/fair/k$k.err || { tail -3 gpurun_out/fair/k$k.err; exit 1; }
  python3 -c "
import json; j=json.loads(open('gpurun_out/fair/k$k.json').read().strip().split('\n')[-1]); print('br_fair $k: match_ms', round(j['match_ms'],1), 'avg 4-wave launch', round(j['roofline']['avg_launch_ms'],3))"
done
