"""benchkit.roofline -- the two rooflines of the bench line: SURVEY 8(d)'s algorithmic bytes against the HBM peak (mandated;
does not bind) and the VALU-issue model that does (profiles/isa_mix.json priced with profiles/valu_issue_costs.json), plus
the committed counter summaries quoted beside them.  Nothing here runs on the GPU or inside the timed region."""
import json
import os

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HBM_PEAK_GBPS = 8000.0   # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E peak 8.0 TB/s (spec)
MI355X_CUS = 256         # same guide: 8 XCDs x 32 CUs, 4 SIMDs per CU


def algorithmic_bytes(pp):
    """bytes per blind rotate / key switch / ciphertext, no cross-gate reuse,
    bootstrapping key counted at 8 B per coefficient (the density this engine stores: two
    32-bit residues)."""
    kpl = (pp.k + 1) * pp.l
    a_br = pp.n * kpl * (pp.k + 1) * pp.N * 8
    a_ks = pp.N * pp.k * pp.ks_t * (1.0 - 2.0 ** (-pp.ks_basebit)) * (pp.n + 1) * 4
    ct = (pp.n + 1) * 4
    return a_br, a_ks, ct


def kernel_source_hash():
    """Identifies the kernels a committed counter summary was measured on: every file kernels.hip is built from, the
    generated key-switch statements and build.sh with its compile flags (peba1_amd/kernel_id.py)."""
    from peba1_amd.kernel_id import kernels_sha16
    return kernels_sha16()


def committed_counters():
    """HBM-side bytes per blind-rotate launch (separate --pmc FETCH_SIZE / WRITE_SIZE passes,
    tools/pmc_summary.py) and the VALU-issue share of the blind-rotate kernel (SQ counters,
    tools/sq_summary.py -> profiles/valu_blind_rotate.json).  Hardware counters cannot be read inside
    this process, so the committed summaries are quoted -- and only when they were measured on
    exactly the kernel sources that are running now; otherwise null."""
    now = kernel_source_hash()
    traffic, valu = None, None
    for name in ("pmc_blind_rotate.json", "valu_blind_rotate.json"):
        path = os.path.join(ROOT, "profiles", name)
        if not os.path.exists(path):
            continue
        with open(path) as f:
            j = json.load(f)
        if j.get("kernels_sha16") != now:
            continue
        if name.startswith("pmc"):
            traffic = j.get("hbm_bytes_per_launch")
        else:
            valu = {k: j[k] for k in ("valu_busy_frac", "valu_insts_per_wave_step", "source") if k in j}
    return traffic, valu, now


def committed_set_profile(name):
    """The rocprofv3 summary of 4,096 independent gates under parameter set `name` (tools/gpu_profile_sets.sh ->
    profiles/r05_set_profile_<name>.json): HBM-side traffic, VALU share, wave-cycle shares.  Quoted only while it was
    measured on exactly the kernel sources running now."""
    j = None
    for rnd in ("r06", "r05", "r04"):                       # the newest summary measured on the kernels running now
        path = os.path.join(ROOT, "profiles", f"{rnd}_set_profile_{name}.json")
        if os.path.exists(path):
            with open(path) as f:
                cand = json.load(f)
            if cand.get("kernels_sha16") == kernel_source_hash():
                j = dict(cand, source_file=f"profiles/{rnd}_set_profile_{name}.json")
                break
    if j is None:
        return None
    keep = ("kernel", "avg_launch_ms", "hbm_bytes_per_launch", "hbm_side_GBps", "hbm_side_frac_of_8TBps", "traffic_over_algorithmic",
            "valu_insts_per_wave_step", "valu_busy_frac", "wave_cycles_issuing", "wave_cycles_issue_stalled", "wave_cycles_parked",
            "wave_cycles_lds_issue_stalled", "lds_conflict_share_of_active", "kernels_sha16", "source_file")
    return {k: j[k] for k in keep if k in j}


MI355X_CUS = 256        # /opt/skills/guides/MI355X_MICROARCH.md: 8 XCDs x 32 CUs, 4 SIMDs per CU


def valu_issue_model():
    """The roofline that binds: cycles one SIMD needs just to ISSUE the vector instructions of one
    blind-rotate step, from two tracked files -- profiles/isa_mix.json (static instruction mix per wave and step of the
    kernels as built, tools/isa_mix.py at build() time; quoted only while its kernels_sha16 is the running one) priced
    with profiles/valu_issue_costs.json (measured issue cost per instruction class, tools/valu_rates*.hip).  Two prices
    (VERDICT r5 item 3): at the occupancy the kernel runs at (two waves per SIMD where two are resident: `model_cycles`),
    and at the chip's best measured issue rates (eight waves per SIMD: `chip_peak_cycles`) -- a ceiling that does not
    concede the kernel's own occupancy.  Returns ({kernel: {...}}, costs) or (None, None)."""
    try:
        with open(os.path.join(ROOT, "profiles", "isa_mix.json")) as f:
            mix = json.load(f)
        with open(os.path.join(ROOT, "profiles", "valu_issue_costs.json")) as f:
            costs = json.load(f)["classes"]
    except (OSError, ValueError, KeyError):
        return None, None
    if mix.get("kernels_sha16") != kernel_source_hash():
        return None, None
    classes = ("mul", "three_operand", "two_operand")
    out = {}
    for name, k in mix["kernels"].items():
        cycles, peak, per_role = 0.0, 0.0, []
        resident = sum(role["waves_per_simd"] for role in k["roles"])
        rate = "two_waves_per_simd" if resident >= 2 else "one_wave_per_simd"
        for role in k["roles"]:
            v = role["variants"][0]                         # the heaviest variant of the role (they differ in scalar code only)
            c = sum(v[cls] * costs[cls][rate] for cls in classes)
            cycles += c * role["waves_per_simd"]
            peak += sum(v[cls] * costs[cls]["eight_waves_per_simd"] for cls in classes) * role["waves_per_simd"]
            per_role.append({"role": role["role"], "waves_per_simd": role["waves_per_simd"], "gadget_rows": role["gadget_rows"],
                             "valu": v["valu"], "mul": v["mul"], "three_operand": v["three_operand"], "two_operand": v["two_operand"],
                             "lds": v["lds"], "barriers": v["barriers"], "issue_cycles_per_wave_step": c})
        out[name] = {"l": k["l"], "insts_per_wave_step": per_role, "model_cycles_per_simd_step": cycles,
                     "chip_peak_cycles_per_simd_step": peak}
    return out, costs


def valu_issue_block(gates4096, sweep, narrow=None):
    """`roofline.valu_issue`: the model above beside what a step takes -- launch time x shader clock / rounds / steps of
    the 4,096-gate launches (4-wave form: two workgroups per CU, 8 rounds; split form at N = 2048: one per CU, 16 rounds)
    and of the 256-gate launch of the batch sweep (8-wave form: one round).  `narrow` (N > 1, where the 8-wave kernel
    dominates and no sweep runs): that kernel's launches of the TIMED steps -- {"ms", "launches", "shader_clock_ghz"};
    every one of them is a single round.
    frac = model / measured <= 1: the share of a step's cycles that the SIMD's vector issue port is busy by the issue-cost
    model at the kernel's own occupancy; frac_vs_chip_peak = the same mix at the chip's best issue rates / measured -- what
    the SQ counters' valu_busy_frac shows; the rest is LDS issue, waits and barrier skew."""
    model, costs = valu_issue_model()
    if model is None:
        return None
    blk = {"files": ["profiles/isa_mix.json", "profiles/valu_issue_costs.json"], "kernels_sha16": kernel_source_hash(),
           "costs_cycles_per_wave_instruction_two_waves_per_simd": {k: costs[k]["two_waves_per_simd"] for k in ("mul", "three_operand", "two_operand")},
           "costs_cycles_per_wave_instruction_chip_peak_eight_waves_per_simd": {k: costs[k]["eight_waves_per_simd"] for k in ("mul", "three_operand", "two_operand")},
           "kernels": model, "frac": None}

    def measured(entry, name, n_steps, per_cu, gates):
        if not entry or name not in model or not entry.get("shader_clock_ghz"):
            return
        rounds = -(-gates // (per_cu * MI355X_CUS))
        cyc = entry["ms_blind_rotate"] * 1e-3 * entry["shader_clock_ghz"] * 1e9 / rounds / n_steps
        m = model[name]
        m.update({"measured_cycles_per_simd_step": cyc, "measured_on": f"{gates} independent gates, {rounds} round(s) of {per_cu} "
                  f"workgroup(s) per CU, {n_steps} steps, {entry['ms_blind_rotate']:.3f} ms at {entry['shader_clock_ghz']:.3f} GHz",
                  "frac": m["model_cycles_per_simd_step"] / cyc, "frac_vs_chip_peak": m["chip_peak_cycles_per_simd_step"] / cyc})
    if gates4096:
        measured(gates4096.get("P128"), "blind_rotate4_kernel<10,true>", 630, 2, 4096)
        measured(gates4096.get("P80"), "blind_rotate4_kernel<10,false>", 500, 2, 4096)
        measured(gates4096.get("P2048"), "blind_rotate_split_kernel<11,2>", 1024, 1, 4096)
    if sweep:
        measured(sweep.get("256"), "blind_rotate8_kernel<10,true>", 630, 1, 256)
    head_name = "blind_rotate4_kernel<10,true>"
    if narrow and narrow.get("launches") and narrow.get("shader_clock_ghz") and "blind_rotate8_kernel<10,true>" in model:
        head_name = "blind_rotate8_kernel<10,true>"
        cyc = narrow["ms"] / narrow["launches"] * 1e-3 * narrow["shader_clock_ghz"] * 1e9 / 630
        m = model[head_name]
        m.update({"measured_cycles_per_simd_step": cyc, "measured_on": f"the {narrow['launches']} launches of the timed steps (one round each), "
                  f"630 steps, {narrow['ms'] / narrow['launches']:.3f} ms per launch at {narrow['shader_clock_ghz']:.3f} GHz",
                  "frac": m["model_cycles_per_simd_step"] / cyc, "frac_vs_chip_peak": m["chip_peak_cycles_per_simd_step"] / cyc})
    head = model.get(head_name, {})
    blk["kernel"] = head_name
    for k in ("frac", "frac_vs_chip_peak", "insts_per_wave_step", "model_cycles_per_simd_step", "chip_peak_cycles_per_simd_step",
              "measured_cycles_per_simd_step"):
        blk[k] = head.get(k)
    return blk
