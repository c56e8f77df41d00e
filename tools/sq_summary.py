#!/usr/bin/env python3
"""Per-kernel sums of the counters in a rocprofv3 --pmc rocpd database (SQ_* counters make the
database too large to carry around, so this runs next to it and prints one line per
kernel and counter).

usage: sq_summary.py <results.db> [kernel substring] [--json out.json --steps-per-wave S]

--json: also write the VALU-issue summary of the blind-rotate kernel that bench.py quotes
(profiles/valu_blind_rotate.json): VALU instructions per wave and blind-rotate step, and the share
of the launch's SIMD cycles spent issuing VALU instructions,
    valu_busy_frac = 4 * SQ_ACTIVE_INST_VALU / (GRBM_GUI_ACTIVE / XCDs * CUs * SIMDs per CU)
(SQ_ACTIVE_INST_* count quad-cycles, GRBM_GUI_ACTIVE is summed over the 8 XCDs;
/opt/skills/guides/MI355X_MICROARCH.md, counter units), stamped with the hash of the kernel sources."""
import collections
import hashlib
import json
import os
import re
import sqlite3
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def kernel_source_hash():
    """bench.py quotes this summary only while what the kernels are built from is unchanged (peba1_amd/kernel_id.py)"""
    import sys
    if ROOT not in sys.path:
        sys.path.insert(0, ROOT)
    from peba1_amd.kernel_id import kernels_sha16
    return kernels_sha16()


def kernel_short_name(name):
    """'void ns::(anonymous namespace)::kernel<10, 0, true>(args)' -> 'kernel<10, 0, true>'"""
    base = name.replace("(anonymous namespace)::", "").split("(")[0].strip()
    m = re.search(r"([A-Za-z_]\w*(?:<.*>)?)$", base.split("::")[-1])
    return m.group(1) if m else name


def main():
    args = sys.argv[1:]
    out_json, steps = None, 629.7          # n = 630 steps, one in 2N skipped (abar = 0)
    if "--json" in args:
        i = args.index("--json"); out_json = args[i + 1]; del args[i:i + 2]
    if "--steps-per-wave" in args:
        i = args.index("--steps-per-wave"); steps = float(args[i + 1]); del args[i:i + 2]
    db, needle = args[0], (args[1] if len(args) > 1 else "")
    c = sqlite3.connect(db)
    # rocprofv3 7.x: tables carry a per-run suffix; find the counters view
    names = [r[0] for r in c.execute("select name from sqlite_master where type in ('table','view')")]
    view = next((n for n in names if n == "counters_collection"), None) or next(n for n in names if n.startswith("counters_collection"))
    agg = collections.defaultdict(lambda: [0, 0.0])
    for name, counter, value in c.execute(f"select kernel_name, counter_name, value from {view}"):
        if needle not in name:
            continue
        short = kernel_short_name(name)
        a = agg[(short, counter)]
        a[0] += 1
        a[1] += value
    for (k, counter), (n, v) in sorted(agg.items()):
        print(f"{k:32s} {counter:24s} dispatches {n:5d}  sum {v:.6g}  per dispatch {v / n:.6g}")
    if out_json:
        br = collections.defaultdict(float)
        for (k, ctr), (n, v) in agg.items():
            if k.startswith("blind_rotate4_kernel"):
                br[ctr] += v
        need = ("SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVES", "GRBM_GUI_ACTIVE")
        if all(k in br for k in need):
            simd_cycles = br["GRBM_GUI_ACTIVE"] / 8.0 * 256 * 4
            out = {"source": "rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVES GRBM_GUI_ACTIVE ... -- python3 "
                             "tools/gate_throughput.py 4096 (blind_rotate4_kernel launches, tools/sq_summary.py)",
                   "kernels_sha16": kernel_source_hash(),
                   "valu_insts_per_wave_step": br["SQ_INSTS_VALU"] / br["SQ_WAVES"] / steps,
                   "valu_busy_frac": 4.0 * br["SQ_ACTIVE_INST_VALU"] / simd_cycles,
                   "counters": {k: v for k, v in br.items()}}
            with open(out_json, "w") as f:
                json.dump(out, f, indent=1)
            print(json.dumps(out))


if __name__ == "__main__":
    main()
