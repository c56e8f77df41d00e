#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc passes (FETCH_SIZE and WRITE_SIZE collected in SEPARATE runs of
the same bench command) into per-kernel HBM traffic, applying the gfx950 corrections of
/opt/skills/guides/MI355X_MICROARCH.md (HBM section): counter unit is KiB; FETCH_SIZE reads
exactly half of a wide (16 B/lane) coalesced stream, so the read side of kernels whose
traffic is such a stream (blind rotate: key image, key switch: KSK rows) is doubled.

usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json>"""
import collections
import csv
import hashlib
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """bench.py quotes this summary only while what the kernels are built from is unchanged (peba1_amd/kernel_id.py)"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from peba1_amd.kernel_id import kernels_sha16
    return kernels_sha16()

WIDE_STREAM = ("blind_rotate2_kernel", "blind_rotate4_kernel", "blind_rotate8_kernel", "blind_rotate_split_kernel", "keyswitch_kernel",
               "keyswitch_strip_kernel", "keyswitch_index_kernel")


def short(name):
    base = name.replace("(anonymous namespace)::", "").split("(")[0]
    base = base.split("<")[0]                      # template arguments (kernel<10>)
    return base.split("::")[-1].split()[-1]        # "void ns::kernel" -> "kernel"


def collect(path, counter):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] != counter:
            continue
        k = short(r["Kernel_Name"])
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return agg


def main():
    fetch = collect(sys.argv[1], "FETCH_SIZE")
    write = collect(sys.argv[2], "WRITE_SIZE")
    out = {"source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes of: bench.py --steps 1 --warmup 0",
           "corrections": "KiB -> bytes; FETCH_SIZE x2 for 16 B/lane coalesced streams (gfx950)",
           "kernels_sha16": kernel_source_hash(), "kernels": {}}
    for k in sorted(set(fetch) | set(write)):
        n = fetch.get(k, write.get(k))[0]
        f_raw = fetch.get(k, [0, 0.0])[1] * 1024.0
        w = write.get(k, [0, 0.0])[1] * 1024.0
        f = f_raw * (2.0 if k in WIDE_STREAM else 1.0)
        out["kernels"][k] = {"launches": n, "fetch_bytes_raw": f_raw, "fetch_bytes_corrected": f, "write_bytes": w,
                             "hbm_bytes_per_launch": (f + w) / max(1, n)}
    br = [out["kernels"][k] for k in ("blind_rotate4_kernel",) if k in out["kernels"]]
    n = sum(b["launches"] for b in br)
    out["blind_rotate_launches"] = n
    out["hbm_bytes_per_launch"] = sum(b["fetch_bytes_corrected"] + b["write_bytes"] for b in br) / max(1, n)
    with open(sys.argv[3], "w") as fo:
        json.dump(out, fo, indent=1)
    print(json.dumps(out, indent=1))


if __name__ == "__main__":
    main()
