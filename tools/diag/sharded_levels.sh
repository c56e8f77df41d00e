# per-level widths and times of the slot-sharded match's per-rank phase (32 slots of 256 on each of 8 logical ranks):
# how far is a rank's 745 ms from what its level count and rotation count allow?
O=gpurun_out/shlev; mkdir -p $O; rm -f $O/times.txt
TFHE_HIP_TRACE_TIMES=$O/times.txt timeout -k 10 300 python bench.py --mode sharded --steps 1 --warmup 1 --extras 0 --no-cpu-baseline > $O/bench.json 2> $O/bench.err || { tail -5 $O/bench.err; exit 1; }
python3 - <<'PY'
import collections
flushes=[]; cur=None
for l in open("gpurun_out/shlev/times.txt"):
    if l.startswith("flush"):
        cur=[]; flushes.append(cur)
    elif cur is not None:
        r=l.split(); cur.append((int(r[0]),float(r[1]),float(r[2]),float(r[3])))
print("flushes", len(flushes), [len(f) for f in flushes][-12:])
# the last nine flushes: eight ranks' partials and rank 0's combine
for f in flushes[-9:-7]+flushes[-1:]:
    rot=sum(w for w,_,_,_ in f); br=sum(b for _,_,b,_ in f); ks=sum(k for _,_,_,k in f)
    wall=f[-1][1]+f[-1][2]+f[-1][3]-f[0][1]
    print(f"levels {len(f)} rotations {rot} br {br:.1f} ms ks {ks:.1f} ms span {wall:.1f} ms")
    cls=collections.OrderedDict()
    for lo,hi in ((1,64),(65,128),(129,192),(193,256),(257,384),(385,512),(513,768),(769,1024),(1025,10**9)):
        v=[(w,b,k) for w,_,b,k in f if lo<=w<=hi]
        if v: print(f"   width {lo:5d}..{hi if hi<10**9 else 'inf':>5}: {len(v):4d} levels, {sum(x[0] for x in v):6d} rotations, br {sum(x[1] for x in v):7.1f} ms (mean {sum(x[1] for x in v)/len(v):.2f}), ks {sum(x[2] for x in v):6.1f} ms")
    print("   widths in order:", [w for w,_,_,_ in f])
PY
