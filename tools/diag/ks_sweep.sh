# key-switch tile and split settings against the time of a whole match (warm clock)
mkdir -p gpurun_out/ks
for v in "TFHE_HIP_KS_TILE=16" "TFHE_HIP_KS_TILE=32" "TFHE_HIP_KS_BLOCKS=16384" "TFHE_HIP_KS_BLOCKS=65536" "TFHE_HIP_KS_TILE=16"; do
  env $v timeout -k 10 200 python bench.py --steps 2 --warmup 1 --extras 0 --no-cpu-baseline > gpurun_out/ks/run.json 2> gpurun_out/ks/run.err || { tail -3 gpurun_out/ks/run.err; exit 1; }
  python3 -c "
import json; j=json.loads(open('gpurun_out/ks/run.json').read().strip().split('\n')[-1]); print('$v: match_ms', round(j['match_ms'],1), 'ks ms per match', round(j['roofline']['ms_keyswitch_per_step'],1))"
done
