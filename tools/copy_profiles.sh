#!/bin/bash
# copies the summaries tools/gpu_profile.sh left under gpurun_out/prof_<tag>/ into profiles/ (round 3 names),
# adding the wave-cycle shares to the VALU summary
set -e
P=gpurun_out/prof_${1:?tag}
cp $P/kernel_stats.csv profiles/r03_kernel_stats.csv
cp $P/pmc_blind_rotate.json profiles/pmc_blind_rotate.json
cp $P/sq_counters_blind_rotate.txt profiles/r03_sq_counters_blind_rotate.txt
cp $P/bench_under_rocprof.json profiles/r03_bench_under_rocprof.json
python3 - "$P" <<'PY'
import json, re, sys
P = sys.argv[1] + "/"
j = json.load(open(P + "valu_blind_rotate.json"))
txt = open(P + "sq_counters_blind_rotate.txt").read()
g = lambda name: float(re.findall(name + r"\s+dispatches\s+\d+\s+sum ([0-9.e+]+)", txt)[-1])
wc = g("SQ_WAVE_CYCLES")
j["wave_cycle_shares"] = {"issuing": g("SQ_ACTIVE_INST_ANY") / wc, "issue_stalled": g("SQ_WAIT_INST_ANY") / wc,
                          "parked_waitcnt_barrier": g("SQ_WAIT_ANY") / wc}
json.dump(j, open("profiles/valu_blind_rotate.json", "w"), indent=1)
print(j["kernels_sha16"], j["valu_insts_per_wave_step"], j["wave_cycle_shares"])
PY
