#!/usr/bin/env python3
"""Where the time of the reference's UNMODIFIED program goes on the drop-in library.

Input: the per-level trace libtfhe-hip writes when TFHE_HIP_TRACE_TIMES names a file (engine.cpp wait_flight):
    flush levels=L start_ms=S          one line per flush (S: host time since the first flush started)
    nrot start_ms br_ms ks_ms wide8    one line per level that ran (wide8 = 1: the 8-wave latency kernel)
    wall_ms=W                          host wall time of the flush, enqueue to completion
and the program's own wall time.  Output: flushes, levels, the histogram of rotations per level, the time in each
blind-rotate kernel, in the key switch, and what is left (host: recording between flushes, key generation, encrypt/decrypt).

    python tools/refmain_breakdown.py times.txt --wall-s 80.1 > profiles/r06_refmain_breakdown.txt
Reference: /root/reference/src/main.cpp:297-465 (the tests of the circuits), :533-586 (the protocol run)."""
import argparse
import collections


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("trace")
    ap.add_argument("--wall-s", type=float, default=None, help="wall time of the whole program (seconds)")
    ap.add_argument("--label", default="oracle/_ref/tfhe_protocol_hip")
    args = ap.parse_args()
    flushes, cur = [], None
    for line in open(args.trace):
        line = line.strip()
        if line.startswith("flush"):
            kv = dict(f.split("=") for f in line.split()[1:])
            cur = {"levels": int(kv["levels"]), "start": float(kv.get("start_ms", 0)), "rows": [], "wall": None}
            flushes.append(cur)
        elif line.startswith("wall_ms="):
            cur["wall"] = float(line.split("=")[1])
        elif line and cur is not None:
            n, a, br, ks, w8 = line.split()
            cur["rows"].append((int(n), float(a), float(br), float(ks), int(w8)))
    rows = [r for f in flushes for r in f["rows"]]
    rot = sum(r[0] for r in rows)
    br8 = sum(r[2] for r in rows if r[4])
    br4 = sum(r[2] for r in rows if not r[4])
    ks = sum(r[3] for r in rows)
    wall_fl = sum(f["wall"] or 0.0 for f in flushes)
    print(f"# {args.label}: per-level trace of every flush (TFHE_HIP_TRACE_TIMES), summarised by tools/refmain_breakdown.py")
    print(f"flushes                      {len(flushes)}")
    print(f"levels that ran a launch     {len(rows)}   (scheduled levels: {sum(f['levels'] for f in flushes)})")
    print(f"blind rotations              {rot}")
    if args.wall_s:
        print(f"program wall time            {args.wall_s:.1f} s   -> {rot / args.wall_s:,.0f} gates/s end to end")
    print(f"sum of flush wall times      {wall_fl / 1e3:.2f} s   (enqueue to completion; the rest of the program is the caller's own host time)")
    print(f"  8-wave latency kernel      {br8 / 1e3:.2f} s   in {sum(1 for r in rows if r[4])} launches, "
          f"{sum(r[0] for r in rows if r[4])} rotations")
    print(f"  4-wave streaming kernel    {br4 / 1e3:.2f} s   in {sum(1 for r in rows if not r[4] and r[0])} launches, "
          f"{sum(r[0] for r in rows if not r[4])} rotations   (a tail round on the 8-wave kernel is inside this figure)")
    print(f"  key switch                 {ks / 1e3:.2f} s")
    print(f"  flush wall - kernels       {(wall_fl - br8 - br4 - ks) / 1e3:.2f} s   (uploads, launch gaps, the host wait)")
    if args.wall_s:
        print(f"outside the flushes          {args.wall_s - wall_fl / 1e3:.2f} s   (key generation and upload, recording ~0.35 M API calls per "
              f"match-sized circuit, bootsSymEncrypt / bootsSymDecrypt, the program's own plaintext code)")
    print()
    print("rotations per level (bucket: levels, rotations, blind-rotate ms, ms per level)")
    edges = [1, 2, 4, 8, 16, 32, 64, 128, 256, 512, 1024, 4096, 1 << 30]
    hist = collections.OrderedDict()
    for r in rows:
        if r[0] == 0:
            continue
        lo = 0
        for e in edges:
            if r[0] <= e:
                key = f"{lo + 1}-{e}" if e < (1 << 30) else f">{lo}"
                break
            lo = e
        h = hist.setdefault(key, [0, 0, 0.0, e])
        h[0] += 1; h[1] += r[0]; h[2] += r[2]
    for key, (nl, nr, ms, e) in sorted(hist.items(), key=lambda kv: kv[1][3]):
        print(f"  {key:>10}  {nl:6d} levels  {nr:8d} rotations  {ms / 1e3:7.2f} s  {ms / nl:6.2f} ms/level  {nr / ms * 1e3 if ms else 0:9,.0f} rot/s")
    print()
    sizes = sorted(((len(f["rows"]), sum(r[0] for r in f["rows"]), f["wall"] or 0.0) for f in flushes), key=lambda t: -t[2])
    print("largest flushes (levels run, rotations, wall ms):")
    for s in sizes[:8]:
        print(f"  {s[0]:5d} {s[1]:8d} {s[2]:10.1f}")
    narrow = [f for f in flushes if sum(r[0] for r in f["rows"]) <= 64]
    print(f"flushes of at most 64 rotations: {len(narrow)} of {len(flushes)}, {sum(f['wall'] or 0 for f in narrow) / 1e3:.2f} s")


if __name__ == "__main__":
    main()
