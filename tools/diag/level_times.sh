# blind-rotate time of every level of one match by launch width: does the match run at the clock a warmed-up
# launch gets (tools/diag/launch_clock.py: 5.1 ms for 512 rotations) or at the cold one (5.9-6.2 ms)?
mkdir -p gpurun_out/trace; rm -f gpurun_out/trace/times.txt
TFHE_HIP_TRACE_TIMES=gpurun_out/trace/times.txt timeout -k 10 300 python bench.py --steps 2 --warmup 1 --extras 0 --no-cpu-baseline > gpurun_out/trace/bench_times.json 2> gpurun_out/trace/bench_times.err || { tail -5 gpurun_out/trace/bench_times.err; exit 1; }
python3 - <<'PY'
import collections
rows=[l.split() for l in open("gpurun_out/trace/times.txt") if not l.startswith("flush")]
by=collections.defaultdict(list)
for r in rows:
    n=int(r[0]); 
    if n: by[n].append(float(r[2]))
tot=sum(len(v) for v in by.values())
print("launches", tot)
for n in sorted(by):
    v=sorted(by[n])
    if len(v)>=8 or n in (256,512,1024): print(f"width {n:5d}: {len(v):4d} launches  br ms min {v[0]:.3f} p50 {v[len(v)//2]:.3f} max {v[-1]:.3f}")
# time course of the 512-wide launches in the last flush
last=[]; 
for l in open("gpurun_out/trace/times.txt"):
    if l.startswith("flush"): last=[]
    else: last.append(l.split())
w512=[(float(r[1]),float(r[2])) for r in last if int(r[0])==512]
print("512-wide launches of the last match, (start ms, br ms), every 20th:", [(round(a),round(b,2)) for a,b in w512[::20]])
gaps=[float(last[i+1][1])-(float(last[i][1])+float(last[i][2])+float(last[i][3])) for i in range(len(last)-1)]
print("gap between a level's key switch end and the next level's start: mean %.4f ms max %.4f total %.2f ms" % (sum(gaps)/len(gaps), max(gaps), sum(gaps)))
PY
