#!/bin/bash
# copies the summaries tools/gpu_profile.sh left under gpurun_out/prof_<tag>/ into profiles/ (named per round: r04_*),
# adding the wave-cycle shares to the VALU summary; and, when tools/gpu_profile_sets.sh ran under the same tag, the
# per-parameter-set summaries (profiles/<round>_set_profile_<set>.json, <round>_<set>_kernel_stats.csv, ..._sq_counters.txt)
#   tools/copy_profiles.sh <tag> [round prefix, default r04]
set -e
P=gpurun_out/prof_${1:?tag}
R=${2:-r04}
if [ -d $P ]; then
cp $P/kernel_stats.csv profiles/${R}_kernel_stats.csv
cp $P/pmc_blind_rotate.json profiles/pmc_blind_rotate.json
cp $P/sq_counters_blind_rotate.txt profiles/${R}_sq_counters_blind_rotate.txt
cp $P/bench_under_rocprof.json profiles/${R}_bench_under_rocprof.json
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
fi
S=gpurun_out/sets_$1
if [ -d $S ]; then
  for n in P128 P80 P2048; do [ -f $S/set_profile_$n.json ] && cp $S/set_profile_$n.json profiles/${R}_set_profile_$n.json; done
  for s in p128 p80 p2048; do
    [ -f $S/${s}_kernel_stats.csv ] && cp $S/${s}_kernel_stats.csv profiles/${R}_${s}_kernel_stats.csv
    [ -f $S/${s}_sq_counters.txt ] && cp $S/${s}_sq_counters.txt profiles/${R}_${s}_sq_counters.txt
  done
  [ -f $S/lds_conflict_attribution.txt ] && cp $S/lds_conflict_attribution.txt profiles/${R}_lds_conflict_attribution.txt
fi
echo copied
