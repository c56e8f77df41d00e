#!/usr/bin/env python3
"""One JSON summary per parameter set from the rocprofv3 passes of `tools/gate_throughput.py [--p80|--p2048] G`
(tools/gpu_profile_sets.sh): kernel-trace time, HBM-side traffic (separate FETCH_SIZE / WRITE_SIZE passes, gfx950
corrections of MI355X_MICROARCH.md: KiB units, FETCH_SIZE x2 for 16 B/lane coalesced streams) and the SQ counters
(VALU instructions per wave and step, VALU-busy share, wave-cycle shares, LDS conflict share) of the blind-rotate
kernel the set runs.  bench.py quotes the result in `independent_gates_4096` while `kernels_sha16` matches.

usage: set_profile_summary.py <dir with stats/ pmc_FETCH_SIZE/ pmc_WRITE_SIZE/ sq1/ sq2/ sq3/> <set> <n> <N> <kpl> <G> <out.json>"""
import collections
import glob
import json
import os
import sqlite3
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from sq_summary import kernel_short_name, kernel_source_hash  # noqa: E402


def db_of(d):
    hits = sorted(glob.glob(os.path.join(d, "**", "*_results.db"), recursive=True))
    return hits[0] if hits else None


def counters(db, needle="blind_rotate"):
    """(kernel, counter) -> [dispatches, sum]"""
    agg = collections.defaultdict(lambda: [0, 0.0])
    if not db:
        return agg
    c = sqlite3.connect(db)
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = next((n for n in names if n == "counters_collection"), None) or next(n for n in names if n.startswith("counters_collection"))
    for name, counter, value in c.execute(f"select kernel_name, counter_name, value from {view}"):
        if needle in name:
            a = agg[(kernel_short_name(name), counter)]
            a[0] += 1
            a[1] += value
    return agg


def main():
    root, pset, n, N, kpl, G, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), sys.argv[7]
    a_br = n * kpl * 2 * N * 8                       # algorithmic bytes per blind rotation (k = 1)
    res = {"set": pset, "n": n, "N": N, "kpl": kpl, "rotations_per_launch": G, "A_br": a_br,
           "source": f"rocprofv3 passes of: python3 tools/gate_throughput.py {'--' + pset.lower() + ' ' if pset != 'P128' else ''}{G} "
                     "(--kernel-trace --stats; --pmc FETCH_SIZE; --pmc WRITE_SIZE; three SQ passes), tools/gpu_profile_sets.sh",
           "corrections": "FETCH_SIZE / WRITE_SIZE in KiB -> bytes; FETCH_SIZE x2 (16 B/lane coalesced key-image stream, gfx950)",
           "kernels_sha16": kernel_source_hash()}
    # kernel trace: the widest blind-rotate kernel of the run
    db = db_of(os.path.join(root, "stats"))
    kern = None
    if db:
        c = sqlite3.connect(db)
        per = collections.defaultdict(list)
        for name, dur in c.execute("select name, duration from kernels"):
            if "blind_rotate" in name:
                per[kernel_short_name(name)].append(dur)
        if per:
            kern = max(per, key=lambda k: sum(per[k]))
            v = per[kern]
            res["kernel"] = kern
            res["launches_traced"] = len(v)
            res["avg_launch_ms"] = sum(v) / len(v) / 1e6
            res["rotations_per_s"] = G / (sum(v) / len(v) / 1e9)
            res["roofline_frac_algorithmic"] = res["rotations_per_s"] * a_br / 8e12
    # HBM-side traffic
    for ctr in ("FETCH_SIZE", "WRITE_SIZE"):
        agg = counters(db_of(os.path.join(root, "pmc_" + ctr)))
        for (k, cn), (cnt, s) in agg.items():
            if cn == ctr and (kern is None or k == kern):
                kern = kern or k
                b = s * 1024.0 / cnt
                if ctr == "FETCH_SIZE":
                    res["fetch_bytes_raw_per_launch"] = b
                    res["fetch_bytes_corrected_per_launch"] = 2.0 * b
                else:
                    res["write_bytes_per_launch"] = b
    if "fetch_bytes_corrected_per_launch" in res and "avg_launch_ms" in res:
        hbm = res["fetch_bytes_corrected_per_launch"] + res.get("write_bytes_per_launch", 0.0)
        res["hbm_bytes_per_launch"] = hbm
        res["hbm_side_GBps"] = hbm / (res["avg_launch_ms"] * 1e-3) / 1e9
        res["hbm_side_frac_of_8TBps"] = res["hbm_side_GBps"] / 8000.0
        res["traffic_over_algorithmic"] = hbm / (a_br * G)
    # SQ counters
    sq = {}
    for p in ("sq1", "sq2", "sq3"):
        for (k, cn), (cnt, s) in counters(db_of(os.path.join(root, p))).items():
            if kern is None or k == kern:
                sq[cn] = s / cnt               # per dispatch
    if sq:
        res["sq_per_launch"] = sq
        steps = n * (1.0 - 1.0 / (2 * N))       # a step whose rotation amount is 0 is skipped
        if "SQ_INSTS_VALU" in sq and "SQ_WAVES" in sq:
            res["valu_insts_per_wave_step"] = sq["SQ_INSTS_VALU"] / sq["SQ_WAVES"] / steps
        if "SQ_ACTIVE_INST_VALU" in sq and "GRBM_GUI_ACTIVE" in sq:
            res["valu_busy_frac"] = 4.0 * sq["SQ_ACTIVE_INST_VALU"] / (sq["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4)
        if "SQ_WAVE_CYCLES" in sq:
            wc = sq["SQ_WAVE_CYCLES"]
            for name, ctr in (("issuing", "SQ_ACTIVE_INST_ANY"), ("issue_stalled", "SQ_WAIT_INST_ANY"), ("parked", "SQ_WAIT_ANY"),
                              ("lds_issue_stalled", "SQ_WAIT_INST_LDS")):
                if ctr in sq:
                    res["wave_cycles_" + name] = sq[ctr] / wc
        if "SQ_LDS_BANK_CONFLICT" in sq and sq.get("SQ_LDS_IDX_ACTIVE"):
            res["lds_conflict_share_of_active"] = sq["SQ_LDS_BANK_CONFLICT"] / sq["SQ_LDS_IDX_ACTIVE"]
    with open(out, "w") as f:
        json.dump(res, f, indent=1)
    print(json.dumps(res))


if __name__ == "__main__":
    main()
